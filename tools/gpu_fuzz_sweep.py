"""One-off wider sweep of the random-scene-graph parity checks on the GPU (the suite runs 12 + 6 seeds):
python tools/gpu_fuzz_sweep.py <first seed> <last seed>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import checkers as ck  # noqa: E402
import util  # noqa: E402
from raytracer_amd import api  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
ctx = api.Context(0)
bad = n = built = 0
cam = api.make_camera((0, 2, -26), (0, 0, 0), (0, 1, 0), 40.0, 4 / 3, 0.1, 20.0, 0.0, 1.0)
for seed in range(lo, hi):
    for flags in (0, 4, 8, 12):  # 8: the world's tree is built on the device
        try:
            b, world = util.random_scene_graph(seed, flags)
            blob, desc = b.compile(world), b.describe(world)
        except api.RtError:
            continue
        osc = ck.OracleScene(desc, share_bvh_axes=True)
        scene = ctx.upload(blob)
        if flags & 8:
            try:
                ctx.build_tree(scene)
                built += 1
            except api.RtError as e:
                if e.code != -1:  # -1: this world keeps reference-order trees, nothing was left to the device
                    raise
        rng = np.random.default_rng(1000 + seed)
        rays = util.random_rays(rng, 4000, any_time=bool(flags & 4))
        xi = rng.random((len(rays), osc.n_media)) if osc.n_media else None
        lane = ctx.trace_first_hit(scene, rays, xi)
        rep = util.first_hit_report(lane, osc.trace_first_hit(rays, xi))
        n += 1
        ok = rep["prim_match"] >= 0.9995 and rep["t_rel_max"] <= 1e-5 and rep["t_bit_equal"] >= 0.95
        # the renderer's closest-hit kernels (the one the scene gets, and the general tree kernel): the same records
        for variant in (0, 12, 16, 32):
            if ctx.trace_first_hit(scene, rays, xi, wavefront=variant).tobytes() != lane.tobytes():
                ok = False
                print("KERNELS DISAGREE", seed, flags, variant, flush=True)
        if not flags & 4:
            p = api.make_params(40, 30, max_depth=12, background=(0.7, 0.8, 1.0), seed=77 + seed, spp_per_pass=3)
            got = ctx.render(scene, cam, p, 4)
            want, _ = osc.render(cam, p, 4, threads=8)
            ok = ok and float((util.rel_err(got, want) > 1e-9).mean()) <= 0.005 and abs(got.mean() / want.mean() - 1.0) <= 1e-3
        if not ok:
            bad += 1
            print("DIFF", seed, flags, rep["prim_match"], rep["t_rel_max"], rep["t_bit_equal"], flush=True)
        scene.close()
    if seed % 20 == 0:
        print("seed", seed, "checked", n, "bad", bad, flush=True)
print("done", n, "bad", bad, "trees built on the device", built)
