"""Renders a few passes of a named scene on cuda:0 (profiling target: small, no oracle, no torch)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import api  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="final_scene")
ap.add_argument("--spp", type=int, default=8)
ap.add_argument("--passes", type=int, default=1)
ap.add_argument("--n", type=int, default=0)
ap.add_argument("--width", type=int, default=0)
ap.add_argument("--height", type=int, default=0)
a = ap.parse_args()
b = api.Builder()
b.register_reference_textures(tuple(api.TEXTURE_FILES) if a.scene == "my_scene" else ("earthmap.jpg",))
info = b.named_scene(a.scene, n=a.n)
ctx = api.Context(0)
sc = ctx.upload(b.compile(info.world))
kw = {}
if a.width:
    kw = dict(width=a.width, height=a.height)
p = api.params_from_info(info, spp_per_pass=a.spp, profile=1, **kw)
st = api.rt_stats()
img = ctx.render(sc, info.camera, p, a.spp * a.passes, stats=st)
print("%s: %d samples, %d segments, %.3f s, %.1f Mrays/s; ms: raygen %.1f tree %.1f shade_main %.1f shade_textured %s accumulate %.1f" %
      (a.scene, st.camera_samples, st.ray_segments, st.seconds_render, st.ray_segments / st.seconds_render / 1e6, st.raygen_ms,
       st.extend_ms, st.shade_main_ms, [round(x, 1) for x in st.shade_ms], st.accumulate_ms))
