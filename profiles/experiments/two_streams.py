"""Does overlapping independent batches on several HIP streams hide the per-launch tails?  K contexts (= K streams) render
the same workload concurrently (async enqueue from one host thread) vs one context doing K times the accumulations."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mirt = importlib.import_module("cpu-raytracing-experiments_amd")
sc = lambda: mirt.scene.synthetic(1000, ambient=0.5)
def make():
    r = mirt.Renderer(sc(), max_bounces=5, use_bvh=True); r.Resize(1024, 1024); r.Accumulate(5); return r
for K in (1, 2, 3, 4):
    rs = [make() for _ in range(K)]
    for r in rs: r.Synchronize()
    spp = 60
    t0 = time.perf_counter()
    for chunk in range(spp // 5):            # interleave enqueues so the streams' batches alternate
        for r in rs: r.AccumulateAsync(5)
    for r in rs: r.Synchronize()
    dt = time.perf_counter() - t0
    rays = sum(r.counters()["rays"] for r in rs) * spp / (spp + 5)
    print(f"K={K}: {dt*1e3:.1f} ms for {K}x{spp} accumulations -> {rays/dt/1e6:.0f} Mray/s", flush=True)
    for r in rs: r.close()
