"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).

The path shards by image tile exactly as the reference shards by row band (main.rs:143-152): the scene is read-only
and shared, every pixel is owned by one worker, results are gathered once (main.rs:242-253). There is no exchange
inside the render: rank r renders the tiles k with k % world == r. Because the Philox key is the pixel index, the
gathered image is bit-identical for every world size.

Collectives: one broadcast of the compiled scene blob from rank 0, one all_gather of the per-rank pixel sums.
"""
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

DEFAULT_TILE = 32


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(script, argv, world, extra_env=None, grace=30.0, timeout=3300.0):
    """One fresh child process per rank running `script argv...` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT set the way torch.distributed.run sets them (rendezvous on 127.0.0.1, a free port). The children inherit
    stdout and stderr, so what rank 0 prints is what the caller's caller reads. Returns the worst exit code. When a
    rank fails the others may wait for it in a collective for ever: they get `grace` seconds, then are ended (by pid).
    When no rank fails but none finishes either - every rank stuck in the rendezvous or in a collective - the wait ends
    after `timeout` seconds all told: the children are ended by pid and the call returns 124, as timeout(1) would.
    The caller must not have initialised a GPU: the children are started with fork + exec."""
    port = free_port()
    started = time.monotonic()
    timed_out = False
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env))
    failed_at = None
    while any(p.poll() is None for p in procs):
        if failed_at is None and any(p.returncode not in (None, 0) for p in procs):
            failed_at = time.monotonic()
        if timeout is not None and not timed_out and time.monotonic() - started > timeout:
            timed_out = True
            print("spawn_ranks: no result after %.0f s; ending the %d rank processes" % (timeout, world), file=sys.stderr)
        if timed_out or (failed_at is not None and time.monotonic() - failed_at > grace):
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.05)
    codes = [p.returncode for p in procs]
    if timed_out:
        return 124, codes
    bad = [c for c in codes if c != 0]
    return (max(bad, key=abs) if bad else 0), codes


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
    index = _rank_indices(width, height, tile, world, flat.device)
    counts = [len(i) for i in index]
    longest = max(counts)
    send = torch.zeros((longest, 3), dtype=flat.dtype, device=flat.device)
    send[:counts[rank]] = flat.index_select(0, index[rank])
    recv = torch.empty((world, longest, 3), dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1))
    full = torch.zeros_like(flat)
    for r in range(world):
        full.index_copy_(0, index[r], recv[r, :counts[r]])
    return full.view(height, width, 3)


_index_cache = {}


def _rank_indices(width, height, tile, world, device):
    """Every rank's pixel index list as tensors on `device`, built once per (image, tile, world, device)."""
    key = (width, height, tile, world, str(device))
    if key not in _index_cache:
        _index_cache[key] = [torch.from_numpy(local_pixel_indices(width, height, tile, r, world)).to(device) for r in range(world)]
    return _index_cache[key]
