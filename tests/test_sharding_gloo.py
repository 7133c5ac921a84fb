"""world_size-2 gloo test of the multi-GPU host logic (tile partition + single gather + resolve on the root),
run on CPU with the oracle standing in for each rank's renderer."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, load_mirt


def _worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    mirt = load_mirt()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = mirt.scene.synthetic(64, ambient=0.5)
        w, h, spp = 64, 48, 5
        tiles = (w // 16) * (h // 16)
        first, count = mirt.distributed.tile_range(tiles, rank, world)
        # stand-in renderer: the oracle renders everything, the rank keeps only the tiles it owns
        o = ob.Oracle(sc, max_bounces=5, trav_mode=0, threads=1); o.Resize(w, h); o.Accumulate(spp)
        full = o.accumulator()
        local = torch.from_numpy(full[first:first + count].copy())
        gathered = mirt.distributed.gather_accumulator(local, tiles, rank, world, buckets=5)
        if rank == 0:
            np.save(out_path, gathered.numpy())
            assert np.array_equal(gathered.numpy().view(np.uint32), full.view(np.uint32))
        # interleaved tile rows (what bench.py uses): rank r keeps rows r, r+N, ...; the root un-interleaves after the gather
        h_tiles, v_tiles = w // 16, h // 16
        first_row, stride, n_rows = mirt.distributed.tile_rows(v_tiles, rank, world)
        rows = full.reshape(v_tiles, h_tiles, 5, 3, 256)[first_row::stride]
        assert rows.shape[0] == n_rows
        local = torch.from_numpy(rows.reshape(n_rows * h_tiles, 5, 3, 256).copy())
        gathered = mirt.distributed.gather_accumulator_rows(local, h_tiles, v_tiles, rank, world, buckets=5)
        if rank == 0:
            assert np.array_equal(gathered.numpy().view(np.uint32), full.view(np.uint32))
    finally:
        dist.destroy_process_group()


def test_tile_ranges_cover_everything():
    mirt = load_mirt()
    for tiles in (1, 7, 64, 4096, 65536):
        for world in (1, 2, 3, 8):
            spans = [mirt.distributed.tile_range(tiles, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == tiles
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_tile_rows_cover_everything():
    mirt = load_mirt()
    for v_tiles in (1, 3, 12, 64, 128):
        for world in (1, 2, 3, 8):
            owned = []
            for r in range(world):
                first, stride, n = mirt.distributed.tile_rows(v_tiles, r, world)
                owned += list(range(first, v_tiles, stride))
                assert n == len(range(first, v_tiles, stride))
            assert sorted(owned) == list(range(v_tiles))


@pytest.mark.timeout(300)
def test_two_rank_gather(tmp_path):
    out = str(tmp_path / "gathered.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert os.path.exists(out)
