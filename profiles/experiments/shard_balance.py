"""How even are the per-rank shards of the weak-scaling image?  Runs each rank's tile range of the N-GPU image on ONE GPU, one after the other."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mirt = importlib.import_module("cpu-raytracing-experiments_amd")
sys.path.insert(0, ROOT)
import bench
sc = mirt.scene.synthetic(1000, ambient=0.5)
for world, mode in ((2, "contiguous"), (2, "rows"), (8, "contiguous"), (8, "rows")):
    w, h = bench.shape_for(world)
    tiles = (w // 16) * (h // 16)
    res = []
    for rank in range(world):
        r = mirt.Renderer(sc, max_bounces=5, use_bvh=True); r.Resize(w, h)
        if mode == "rows":
            r.SetTileRows(rank, world)
        else:
            first, count = mirt.distributed.tile_range(tiles, rank, world)
            r.SetTileRange(first, count)
        r.Accumulate(32); c0 = r.counters()["rays"]
        t0 = time.perf_counter(); r.Accumulate(64); dt = time.perf_counter() - t0
        res.append((r.counters()["rays"] - c0, dt)); r.close()
    tmax = max(d for _, d in res); total = sum(n for n, _ in res)
    print(f"world {world} {mode:10s} ({w}x{h}): rays per rank (M) {[round(n/1e6,1) for n,_ in res]} | ms {[round(d*1e3,1) for _,d in res]} | "
          f"aggregate if run in parallel {total/tmax/1e6:.0f} Mray/s = {total/tmax/ (max(n/d for n,d in res)*world):.2f} of {world} x best rank", flush=True)
