"""Child process of tests/test_gpu_dist.py: torch is imported BEFORE the HIP library, as in bench.py, so that both use
the HIP runtime torch brings (loaded the other way round, torch finds no GPU). Runs the checks named on the command
line on cuda:0 and prints "<name> ok" for each."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import util  # noqa: E402
from raytracer_amd import api  # noqa: E402
from raytracer_amd import dist as rdist  # noqa: E402


def upload_from_device(ctx):
    _, info, blob, _ = util.build_scene("final_scene")
    a = ctx.upload(blob)
    buf = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to("cuda:0")
    torch.cuda.synchronize()
    b = ctx.upload_from_device(buf.data_ptr(), buf.numel())
    del buf                                                         # the library keeps its own copy
    torch.cuda.synchronize()
    p = api.params_from_info(info, width=160, height=120, seed=util.RENDER_SEED)
    rays = ctx.generate_rays(info.camera, p, 0)
    xi = np.random.default_rng(3).random((len(rays), 2))
    assert ctx.trace_first_hit(a, rays, xi).tobytes() == ctx.trace_first_hit(b, rays, xi).tobytes()
    p = api.params_from_info(info, width=96, height=64, seed=util.RENDER_SEED, spp_per_pass=4)
    assert ctx.render(a, info.camera, p, 8).tobytes() == ctx.render(b, info.camera, p, 8).tobytes()
    try:
        ctx.upload_from_device(0, len(blob))
        raise AssertionError("a null device pointer was accepted")
    except api.RtError:
        pass
    a.close()
    b.close()


def rccl_one_rank(ctx):
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from raytracer_amd.dist import free_port
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % free_port(), rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
        _, info, blob, _ = util.build_scene("cornell_box")
        buf = rdist.broadcast_blob(blob, dev)
        assert buf.is_cuda and bytes(buf.cpu().numpy().tobytes()) == blob
        scene = ctx.upload_from_device(buf.data_ptr(), buf.numel())
        W, H = 100, 70
        p = api.params_from_info(info, width=W, height=H, seed=util.RENDER_SEED, tile=rdist.DEFAULT_TILE, tile_rank=0, tile_world=1,
                                 spp_per_pass=4)
        d_sum = torch.zeros(W * H * 3, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        ctx.render_accumulate(scene, info.camera, p, 0, 4, d_sum.data_ptr())
        full = rdist.gather_image(d_sum, W, H, rdist.DEFAULT_TILE, 0, 1)
        again = rdist.gather_image(d_sum, W, H, rdist.DEFAULT_TILE, 0, 1)       # (index lists come from the cache now)
        host = ctx.render(ctx.upload(blob), info.camera, p, 4)
        assert np.array_equal(full.cpu().numpy(), host) and torch.equal(full, again)
        scene.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    context = api.Context(0)
    for name in sys.argv[1:]:
        {"upload_from_device": upload_from_device, "rccl_one_rank": rccl_one_rank}[name](context)
        print(name, "ok", flush=True)
    context.close()
