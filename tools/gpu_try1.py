import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from raytracer_amd import api
import checkers as ck

ctx = api.Context(0)
for name in ["final_scene", "cornell_box", "cornell_smoke", "random_scene", "stress_volume"]:
    b = api.Builder(); b.register_reference_textures()
    info = b.named_scene(name)
    blob, desc = b.compile(info.world), b.describe(info.world)
    osc = ck.OracleScene(desc, share_bvh_axes=True)
    sc = ctx.upload(blob)
    p = api.params_from_info(info, width=96, height=64)
    rg = ctx.generate_rays(info.camera, p, 0)
    ro = ck.oracle_generate_rays(info.camera, p, 0)
    print(name, "rays equal", np.array_equal(rg.view(np.float64), ro.view(np.float64)))
    xi = np.random.default_rng(1).random((len(ro), max(osc.n_media, 1)))[:, :osc.n_media]
    hg = ctx.trace_first_hit(sc, ro, xi if osc.n_media else None)
    ho = osc.trace_first_hit(ro, xi if osc.n_media else None)
    print("  prim match", (hg["prim_id"] == ho["prim_id"]).mean(), "t equal", (hg["t"] == ho["t"]).mean(), "p equal", (hg["px"] == ho["px"]).mean(), "n equal", (hg["nx"] == ho["nx"]).mean())
    t = time.time(); got = ctx.render(sc, info.camera, p, 8); tg = time.time() - t
    want, st = osc.render(info.camera, p, 8, threads=8)
    rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-30); rel[got == want] = 0
    print("  render gpu %.3fs mean %.6f oracle %.6f exact %.4f maxrel %.3e bad %d" % (tg, got.mean(), want.mean(), (got == want).mean(), rel.max(), (rel > 1e-9).sum()))
    sc.close()
# throughput probe
b = api.Builder(); b.register_reference_textures()
info = b.named_scene("final_scene")
sc = ctx.upload(b.compile(info.world))
p = api.params_from_info(info, spp_per_pass=8, profile=1)
st = api.rt_stats()
t = time.time(); img = ctx.render(sc, info.camera, p, 16, stats=st); dt = time.time() - t
print("final_scene 800x800x16: %.3fs wall, %.3fs device, %d samples %d segs -> %.1f Mrays/s %.1f Msamples/s" % (dt, st.seconds_render, st.camera_samples, st.ray_segments, st.ray_segments / st.seconds_render / 1e6, st.camera_samples / st.seconds_render / 1e6))
print("ms: raygen %.2f extend %.2f shade %s accum %.2f; extend launches %d" % (st.raygen_ms, st.extend_ms, [round(x, 2) for x in st.shade_ms], st.accumulate_ms, st.extend_launches))
ctx.close()
