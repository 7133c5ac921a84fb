"""Multi-GPU host logic: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm).

The path shards by independent units: tiles are independent (Renderer.hpp:75-88) and every random draw depends only
on the global LaunchIndex, ID and accumulations (Renderer.hpp:107,117), so any partition of the tiles reproduces the
single-GPU result bit for bit.  Rank r renders the tile ROWS r, r+N, r+2N, ... (tile_rows; every rank sees the whole
image height — contiguous stripes differ by 1.5x in cost, sky above, spheres below) with the scene replicated, and the
only exchange is ONE gather of the accumulator slabs ([tile][bucket][rgb][256] f32, Renderer.hpp:43-46) to rank 0 at
the end, un-interleaved there on the device.  No all-reduce, no per-frame traffic.  tile_range / gather_accumulator
are the contiguous variant.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def tile_range(n_tiles: int, rank: int, world: int):
    """Contiguous, balanced split of [0, n_tiles): the first n_tiles % world ranks own one extra tile."""
    base, extra = divmod(n_tiles, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def gather_accumulator(local: torch.Tensor, n_tiles: int, rank: int, world: int, buckets: int):
    """Gather per-rank slabs [count_r, buckets, 3, 256] to rank 0 -> [n_tiles, buckets, 3, 256] (None elsewhere)."""
    if world == 1:
        return local
    counts = [tile_range(n_tiles, r, world)[1] for r in range(world)]
    pad_to = max(counts)
    send = local
    if local.shape[0] != pad_to:                      # dist.gather wants equal shapes: pad the short slabs by one tile
        send = torch.zeros((pad_to,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, gather_list=bufs, dst=0)
    if rank != 0:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def tile_rows(v_tiles: int, rank: int, world: int):
    """Interleaved split of the tile rows [0, v_tiles): rank r owns rows r, r + world, ...  Returns (first_row, row_stride, n_rows)."""
    n_rows = (v_tiles - rank + world - 1) // world if rank < v_tiles else 0
    return rank, world, n_rows


def gather_accumulator_rows(local: torch.Tensor, h_tiles: int, v_tiles: int, rank: int, world: int, buckets: int):
    """Gather per-rank slabs of interleaved tile rows ([n_rows_r * h_tiles, buckets, 3, 256]) to rank 0 and put every row
    back in its place -> [v_tiles * h_tiles, buckets, 3, 256] in LaunchIndex order (None on the other ranks)."""
    if world == 1:
        return local
    rows = [tile_rows(v_tiles, r, world)[2] for r in range(world)]
    pad_to = max(rows) * h_tiles
    send = local
    if local.shape[0] != pad_to:                      # dist.gather wants equal shapes: pad the short slabs by one tile row
        send = torch.zeros((pad_to,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, gather_list=bufs, dst=0)
    if rank != 0:
        return None
    full = torch.empty((v_tiles, h_tiles) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r, (b, n) in enumerate(zip(bufs, rows)):
        full[r::world] = b[: n * h_tiles].view((n, h_tiles) + tuple(local.shape[1:]))
    return full.view((v_tiles * h_tiles,) + tuple(local.shape[1:]))


class _DeviceMemory:
    """Exposes a raw device allocation (the context's accumulator slab) through __cuda_array_interface__ so torch can
    wrap it without a copy for the RCCL gather."""

    def __init__(self, ptr: int, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_tensor(ptr: int, nbytes: int, shape) -> torch.Tensor:
    n = 1
    for s in shape:
        n *= int(s)
    assert n * 4 == nbytes, (shape, nbytes)
    return torch.as_tensor(_DeviceMemory(ptr, shape), device="cuda")
