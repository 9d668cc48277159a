"""Experiment: the same number of samples as two half-size renders running at once in two contexts (two streams) of one GPU."""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import api  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="final_scene")
ap.add_argument("--spp", type=int, default=512, help="per context and pass")
ap.add_argument("--passes", type=int, default=2)
ap.add_argument("--contexts", type=int, default=2)
a = ap.parse_args()
b = api.Builder()
b.register_reference_textures(("earthmap.jpg",))
info = b.named_scene(a.scene)
blob = b.compile(info.world)
ctxs = [api.Context(0) for _ in range(a.contexts)]
scs = [c.upload(blob) for c in ctxs]
p = api.params_from_info(info, spp_per_pass=a.spp, profile=0)
stats = [api.rt_stats() for _ in ctxs]


def work(k, n):
    ctxs[k].render(scs[k], info.camera, p, n, stats=stats[k])


for k in range(a.contexts):  # warm-up: allocates
    work(k, a.spp)
stats = [api.rt_stats() for _ in ctxs]
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(k, a.spp * a.passes)) for k in range(a.contexts)]
for t in th:
    t.start()
for t in th:
    t.join()
dt = time.perf_counter() - t0
seg = sum(s.ray_segments for s in stats)
print("%d contexts x %d spp x %d passes: %d segments in %.3f s = %.1f Mrays/s" % (a.contexts, a.spp, a.passes, seg, dt, seg / dt / 1e6))
