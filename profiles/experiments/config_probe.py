"""Throughput probe of the other BASELINE configs on one GPU (cfg4/cfg5: one GPU's 1/8 tile shard of the 4096x4096 image)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mirt = importlib.import_module("cpu-raytracing-experiments_amd")
which = sys.argv[1:] or ["cfg1", "cfg3", "cfg4", "cfg5"]
for name in which:
    cfg = mirt.scene.CONFIGS[name]
    t0 = time.perf_counter()
    sc = mirt.scene.synthetic(cfg["n"], ambient=cfg["ambient"])
    r = mirt.Renderer(sc, max_bounces=cfg["max_bounces"], buckets=cfg["buckets"], use_bvh=bool(cfg["use_bvh"]))
    r.Resize(cfg["width"], cfg["height"])
    tiles = (cfg["width"] // 16) * (cfg["height"] // 16)
    if name in ("cfg4", "cfg5"):
        first, count = mirt.distributed.tile_range(tiles, 3, 8)      # rank 3 of 8
        r.SetTileRange(first, count)
    setup = time.perf_counter() - t0
    spp = min(cfg["spp"], 64 if name != "cfg1" else 1)
    r.Accumulate(cfg["buckets"] if name != "cfg1" else 1)            # warm-up batch
    r.Synchronize(); c0 = r.counters()
    t0 = time.perf_counter(); r.Accumulate(spp); dt = time.perf_counter() - t0
    c1 = r.counters()
    print(f"{name}: {cfg['width']}x{cfg['height']} S({cfg['n']}) bounces {cfg['max_bounces']} buckets {cfg['buckets']} | setup {setup:.2f}s | {spp} acc in {dt*1e3:.1f} ms "
          f"-> {(c1['rays']-c0['rays'])/dt/1e6:.0f} Mray/s (+{(c1['shadow_rays']-c0['shadow_rays'])/dt/1e6:.0f} M shadow rays/s) | {r.debug_info()}", flush=True)
    r.close()
