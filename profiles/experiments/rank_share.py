#!/usr/bin/env python3
"""One rank's share of the strong-scaled cfg4 image (tile rows r, r+N, ... of 4096x4096, S(100000)) on this box's GPU:
Mray/s of that share for different batch sizes / streams — what each GPU of an N-GPU run does (no exchange while rendering).
    python profiles/experiments/rank_share.py [N=8]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mirt = importlib.import_module("cpu-raytracing-experiments_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = mirt.scene.CONFIGS["cfg4"]
sc = mirt.scene.synthetic(cfg["n"])
for streams, max_batch in ((3, 0), (1, 0), (3, 8), (3, 32), (3, 64), (2, 32), (1, 64)):
    r = mirt.Renderer(sc, max_bounces=cfg["max_bounces"], use_bvh=True, streams=streams, max_batch=max_batch)
    r.Resize(cfg["width"], cfg["height"]); r.SetTileRows(0, N)
    r.Accumulate(64)
    c0 = r.counters()["rays"]; t0 = time.perf_counter()
    r.AccumulateAsync(64); r.AccumulateAsync(64); r.Synchronize()
    dt = time.perf_counter() - t0
    rays = r.counters()["rays"] - c0
    print(f"1/{N} of cfg4, streams {streams}, batch {r.get_policy()['max_batch']:2d}: {rays / dt / 1e6:8.1f} Mray/s  ({dt / 2 * 1e3:.1f} ms per 64 accumulations)  -> x{N} = {rays / dt / 1e9 * N:.1f} Gray/s", flush=True)
    r.close()
