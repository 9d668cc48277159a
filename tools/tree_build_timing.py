"""Host-built (surface-area heuristic) against device-built (linear BVH) world tree: build time and the rays/s each tree
gives. One MI355X.  python tools/tree_build_timing.py [--scene spheres_grid --n 1000000 --spp 16]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import api  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="spheres_grid")
ap.add_argument("--n", type=int, default=1000000)
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--passes", type=int, default=2)
ap.add_argument("--width", type=int, default=0)
ap.add_argument("--height", type=int, default=0)
a = ap.parse_args()

ctx = api.Context(0)
out = {"scene": a.scene, "n": a.n}


def build(flags):
    b = api.Builder(flags=flags)
    b.register_reference_textures(tuple(api.TEXTURE_FILES) if a.scene == "my_scene" else ("earthmap.jpg",))
    t0 = time.time()
    info = b.named_scene(a.scene, n=a.n)
    t1 = time.time()
    blob = b.compile(info.world)
    t2 = time.time()
    sc = ctx.upload(blob)
    t3 = time.time()
    return info, sc, {"scene_function_s": t1 - t0, "compile_s": t2 - t1, "upload_s": t3 - t2, "blob_bytes": len(blob)}


def rate(info, sc):
    kw = dict(width=a.width, height=a.height) if a.width else {}
    p = api.params_from_info(info, spp_per_pass=a.spp, **kw)
    ctx.render(sc, info.camera, p, a.spp)  # warm-up
    st = api.rt_stats()
    img = ctx.render(sc, info.camera, p, a.spp * a.passes, stats=st)
    return {"rays_per_s": st.ray_segments / st.seconds_render, "samples_per_s": st.camera_samples / st.seconds_render,
            "image_mean": float(img.mean())}


info, sc, t = build(0)
out["host_sah"] = dict(t, **rate(info, sc))
print(json.dumps({"host_sah": out["host_sah"]}), flush=True)
sc.close()
info, sc, t = build(api.FLAG_DEVICE_TREE)
out["standin"] = dict(t, **rate(info, sc))
print(json.dumps({"standin": out["standin"]}), flush=True)
ts = ctx.build_tree(sc)
out["device_lbvh"] = dict({k: getattr(ts, k) for k, _ in ts._fields_}, **rate(info, sc))
print(json.dumps({"device_lbvh": out["device_lbvh"]}), flush=True)
print(json.dumps(out))
