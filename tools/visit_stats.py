"""Work per path segment of the product's traversal, counted on the CPU by the host build of the device code (no GPU):
node visits (two box tests each) and object tests by kind, by bounce.   python tools/visit_stats.py [scene] [width] [spp]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import checkers as ck  # noqa: E402
import util  # noqa: E402
from raytracer_amd import _lib, api  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "final_scene"
width = int(sys.argv[2]) if len(sys.argv) > 2 else 64
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 8
_, info, blob, _ = util.build_scene(name)
p = api.params_from_info(info, width=width, height=max(2, width * info.height // info.width), seed=util.RENDER_SEED)
h = ck.hostsim()
h.hs_visit_stats.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_lib.rt_camera), C.POINTER(_lib.rt_params), C.c_uint32, C.c_void_p]
v = np.zeros((p.max_depth, 12), dtype=np.uint64)
assert h.hs_visit_stats(blob, len(blob), C.byref(info.camera), C.byref(p), spp, v.ctypes.data) == 0
v = v.astype(np.float64)
kinds = ["nodes", "-", "-", "sphere", "msphere", "rect", "box", "medium"]
def row(label, a):
    seg = a[11]
    print("%-10s %10d segments: %6.2f node visits, " % (label, seg, a[0] / seg) + ", ".join("%.2f %s" % (a[1 + k] / seg, kinds[k]) for k in (2 + 1, 3 + 1, 4 + 1, 5 + 1, 6 + 1)[0:0]) +
          ", ".join("%.2f %s" % (a[1 + t] / seg, n) for t, n in ((2, "sphere"), (3, "moving sphere"), (4, "rect"), (5, "box"), (6, "medium")) if a[1 + t]))
row("bounce 0", v[0]); row("bounce 1", v[1]); row("bounce 2-9", v[2:10].sum(0)); row("bounce 10+", v[10:].sum(0)); row("all", v.sum(0))
