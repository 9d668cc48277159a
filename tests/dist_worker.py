"""One rank of the world_size-2 CPU rehearsal of the multi-GPU path (gloo). The render itself needs a GPU; here each
rank fills its own tiles from the host simulator's image, so what is exercised is the sharding rule, the scene
broadcast and the gather."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import checkers as ck  # noqa: E402
import util  # noqa: E402
from raytracer_amd import api  # noqa: E402
from raytracer_amd import dist as rdist  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, TILE, SPP = 50, 37, 8, 2
    blob = None
    if rank == 0:
        _, info, blob, _ = util.build_scene("cornell_smoke")
    buf = rdist.broadcast_blob(blob, torch.device("cpu"))
    blob_here = bytes(buf.numpy().tobytes())
    _, info, blob0, _ = util.build_scene("cornell_smoke")          # every rank can rebuild it: must equal the broadcast
    assert blob_here == blob0
    p = api.params_from_info(info, width=W, height=H, seed=util.RENDER_SEED)
    full, _ = ck.hostsim_render(blob_here, info.camera, p, SPP)
    mine = rdist.local_pixel_indices(W, H, TILE, rank, world)
    part = np.zeros_like(full).reshape(-1, 3)
    part[mine] = full.reshape(-1, 3)[mine]
    # every pixel has exactly one owner
    owners = np.zeros(W * H, dtype=np.int64)
    for r in range(world):
        owners[rdist.local_pixel_indices(W, H, TILE, r, world)] += 1
    assert np.all(owners == 1)
    got = rdist.gather_image(torch.from_numpy(part.reshape(-1)), W, H, TILE, rank, world).numpy()
    assert np.array_equal(got, full), "gathered image differs from the single-process image"
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
