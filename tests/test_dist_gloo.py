"""world_size-2 rehearsal of the multi-GPU path on the CPU (gloo): scene broadcast, tile ownership, gather."""
import os
import subprocess
import sys

import numpy as np

from raytracer_amd import dist as rdist

HERE = os.path.dirname(os.path.abspath(__file__))


def test_tile_ownership_partitions_the_image():
    for (w, h, tile, world) in [(800, 800, 32, 8), (100, 70, 32, 3), (33, 5, 8, 2), (16, 16, 0, 1), (7, 9, 4, 16)]:
        seen = np.zeros(w * h, dtype=np.int64)
        for r in range(world):
            idx = rdist.local_pixel_indices(w, h, tile, r, world)
            assert len(np.unique(idx)) == len(idx)
            seen[idx] += 1
        assert np.all(seen == 1)
    # balance at the bench size: 625 tiles of 32x32 over 8 ranks -> 78 or 79 tiles each
    counts = [len(rdist.local_pixel_indices(800, 800, 32, r, 8)) for r in range(8)]
    assert max(counts) - min(counts) <= 32 * 32


def test_two_ranks_broadcast_and_gather():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py")], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
        assert "rank %d ok" % r in o
