"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).

The path shards by image tile exactly as the reference shards by row band (main.rs:143-152): the scene is read-only
and shared, every pixel is owned by one worker, results are gathered once (main.rs:242-253). There is no exchange
inside the render: rank r renders the tiles k with k % world == r. Because the Philox key is the pixel index, the
gathered image is bit-identical for every world size.

Collectives: one broadcast of the compiled scene blob from rank 0, one all_gather of the per-rank pixel sums.
"""
import numpy as np
import torch
import torch.distributed as dist

DEFAULT_TILE = 32


def local_pixel_indices(width, height, tile, rank, world):
    """Global pixel indices (y*width + x) of the tiles rank owns, in the order the device pixel map uses:
    tile k belongs to rank k % world; tiles in row-major order, pixels row-major inside a tile."""
    tile = tile or DEFAULT_TILE
    tx, ty = -(-width // tile), -(-height // tile)
    out = []
    for k in range(rank, tx * ty, world):
        x0, y0 = (k % tx) * tile, (k // tx) * tile
        ys = np.arange(y0, min(y0 + tile, height), dtype=np.int64)
        xs = np.arange(x0, min(x0 + tile, width), dtype=np.int64)
        out.append((ys[:, None] * width + xs[None, :]).ravel())
    return np.concatenate(out) if out else np.zeros(0, dtype=np.int64)


def broadcast_blob(blob, device, src=0):
    """rank `src` passes the compiled scene bytes, the others None; returns a uint8 tensor on `device` holding the
    blob on every rank (one size broadcast + one payload broadcast)."""
    rank = dist.get_rank()
    n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src)
    if rank == src:
        buf = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src)
    return buf


def gather_image(rgb_sum, width, height, tile, rank, world):
    """rgb_sum: this rank's (height*width*3) f64 tensor with its own tiles filled. Returns the full image on every
    rank: each rank contributes only its own pixels (padded to a common length), one all_gather."""
    flat = rgb_sum.view(-1, 3)
    counts = [len(local_pixel_indices(width, height, tile, r, world)) for r in range(world)]
    longest = max(counts)
    mine = torch.from_numpy(local_pixel_indices(width, height, tile, rank, world)).to(flat.device)
    send = torch.zeros((longest, 3), dtype=flat.dtype, device=flat.device)
    send[:counts[rank]] = flat.index_select(0, mine)
    recv = torch.empty((world, longest, 3), dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1))
    full = torch.zeros_like(flat)
    for r in range(world):
        idx = torch.from_numpy(local_pixel_indices(width, height, tile, r, world)).to(flat.device)
        full.index_copy_(0, idx, recv[r, :counts[r]])
    return full.view(height, width, 3)
