"""The multi-GPU plumbing on one GPU: the scene taken from a device buffer (what arrives by RCCL broadcast), and the
broadcast / gather helpers under backend "nccl" (= RCCL) with a world of one rank. Needs an MI355X. The checks run in
one child process (tests/gpu_dist_worker.py) that imports torch before the HIP library, the order bench.py uses."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_device_upload_and_rccl_helpers_on_one_gpu():
    r = subprocess.run([sys.executable, os.path.join(HERE, "gpu_dist_worker.py"), "upload_from_device", "rccl_one_rank"],
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "upload_from_device ok" in r.stdout and "rccl_one_rank ok" in r.stdout


def test_bench_refuses_more_gpus_than_there_are():
    """`python bench.py --gpus 2` without a launcher starts its own ranks - or, on a one-GPU box, says what it needs."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this machine has the GPUs")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=280, env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    assert r.returncode != 0 and "needs 2 GPUs" in r.stderr


def test_two_rank_bench_on_one_gpu_renders_the_one_rank_image():
    """bench.py --gpus 2 started without a launcher: it spawns its own ranks, broadcasts the scene, shards the tiles,
    gathers the image. Both ranks use this box's one GPU and gloo carries the collectives; the image must be the
    one-rank image bit for bit (same samples per pixel: weak scaling doubles the per-step spp with the rank count)."""
    import json
    bench = os.path.join(os.path.dirname(HERE), "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    lines = []
    for extra in (["--gpus", "1", "--spp", "8"], ["--gpus", "2", "--rehearse-on-one-gpu", "--spp", "4"]):
        r = subprocess.run([sys.executable, bench, "--steps", "2", "--warmup", "1", "--cpu-seconds", "0"] + extra,
                           capture_output=True, text=True, timeout=280, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        lines.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]))
    one, two = lines
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["rccl_ranks"] == 2
    assert one["config"]["spp_per_step"] == two["config"]["spp_per_step"] == 8
    assert one["image_sha256"] == two["image_sha256"]
    assert abs(one["rays_per_sample"] - two["rays_per_sample"]) < 1e-9
