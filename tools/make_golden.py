"""Regenerates tests/golden/*.npz from the CPU oracle: per scene, a fixed set of rays with the oracle's first hits,
and a small render. The reference ships no vectors of its own; these pin the oracle (and, through the tests, the
GPU path) against regressions.

    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import checkers as ck  # noqa: E402
import util  # noqa: E402
from raytracer_amd import api  # noqa: E402

import json  # noqa: E402

out_dir = os.path.join(ROOT, "tests", "golden")
os.makedirs(out_dir, exist_ok=True)
visits = {}
for name in util.SCENES:
    _, info, blob, desc = util.build_scene(name)
    osc = ck.OracleScene(desc, share_bvh_axes=True)
    p = api.params_from_info(info, width=20, height=15, seed=util.RENDER_SEED)
    rng = np.random.default_rng(2025)
    cam = ck.oracle_generate_rays(info.camera, p, 0)
    first = osc.trace_first_hit(cam, rng.random((len(cam), osc.n_media)) if osc.n_media else None)
    rays = util.ray_suite(cam, first, rng, n_secondary=150)
    xi = rng.random((len(rays), max(osc.n_media, 1)))[:, :osc.n_media]
    hits = osc.trace_first_hit(rays, xi if osc.n_media else None)
    img, st = osc.render(info.camera, p, 4, threads=4)
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), rays=rays, xi=xi, hits=hits, image=img,
                        segments=np.uint64(st.ray_segments))
    print(name, len(rays), "rays,", int((hits["prim_id"] != 0xFFFFFFFF).sum()), "hits, image mean %.6f" % img.mean())
    # work per path segment under the reference's traversal (every BvhNode::hit, every primitive hit() call), which
    # is what the roofline's scene-traffic and flop figures are stated in
    visits[name] = util.oracle_visits(osc, info)
json.dump(visits, open(os.path.join(out_dir, "visits.json"), "w"), indent=1, sort_keys=True)
