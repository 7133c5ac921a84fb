"""Per-kernel times vs stream size: S(100000), 9 iterations; image heights 512..4096 at width 4096 (streams=1 for clean timings)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mirt = importlib.import_module("cpu-raytracing-experiments_amd")
sc = mirt.scene.synthetic(int(sys.argv[1]) if len(sys.argv) > 1 else 100000)
for h in (512, 1024, 2048, 4096):
    r = mirt.Renderer(sc, max_bounces=9, use_bvh=True, profile=True, streams=1)
    r.Resize(4096, h); r.Accumulate(5); r.kernel_times(reset=True); c0 = r.counters()
    t0 = time.perf_counter(); r.Accumulate(10); dt = time.perf_counter() - t0
    kt = r.kernel_times(); c = r.counters()
    rays = c["rays"] - c0["rays"]
    print(f"4096x{h}: {rays/dt/1e6:.0f} Mray/s | per launch ms:", {k: round(v["ms"] / max(v["launches"], 1), 3) for k, v in kt.items() if v["launches"]},
          "| ns per ray: trace %.2f shade %.2f" % (kt["trace"]["ms"] * 1e6 / (rays + c["shadow_rays"] - c0["shadow_rays"]), kt["shade"]["ms"] * 1e6 / rays), flush=True)
    r.close()
