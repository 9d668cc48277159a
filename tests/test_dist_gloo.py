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


def test_spawn_ranks_runs_the_two_rank_rehearsal():
    """The launcher bench.py uses for --gpus N > 1 when nothing else has started its ranks."""
    rc, codes = rdist.spawn_ranks(os.path.join(HERE, "dist_worker.py"), [], 2, extra_env={"OMP_NUM_THREADS": "1"})
    assert rc == 0 and codes == [0, 0]


def test_spawn_ranks_reports_a_failing_rank_and_ends_the_others(tmp_path):
    script = tmp_path / "w.py"
    script.write_text("import os, sys, time\n"
                      "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and os.environ['WORLD_SIZE'] == '3'\n"
                      "r = int(os.environ['RANK'])\n"
                      "assert os.environ['LOCAL_RANK'] == str(r)\n"
                      "if r == 1: sys.exit(7)\n"
                      "if r == 2: time.sleep(600)\n")
    rc, codes = rdist.spawn_ranks(str(script), [], 3, grace=1.0)
    assert rc != 0 and codes[0] == 0 and codes[1] == 7 and codes[2] not in (0, None)


def test_bench_with_more_gpus_than_the_machine_has_says_so():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this machine has the GPUs")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=240, env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    assert r.returncode != 0
    assert "needs 2 GPUs" in r.stderr and "torch.distributed.run" not in r.stderr
