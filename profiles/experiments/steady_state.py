"""One big batch (S(1000), 4096x4096, 5 accumulations = 84 M primary rays per launch, streams=1): the trace launches are long
enough that their tails do not matter, so PMC counters of this run show the steady state of k_trace.
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY -- python3 steady_state.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mirt = importlib.import_module("cpu-raytracing-experiments_amd")
sc = mirt.scene.synthetic(int(sys.argv[1]) if len(sys.argv) > 1 else 1000, ambient=0.5)
r = mirt.Renderer(sc, max_bounces=5, use_bvh=True, profile=True, streams=1)
r.Resize(4096, 4096)
r.Accumulate(5); r.kernel_times(reset=True); c0 = r.counters()
t0 = time.perf_counter(); r.Accumulate(5); dt = time.perf_counter() - t0
kt = r.kernel_times(); c = r.counters()
rays = c["rays"] - c0["rays"]; srays = c["shadow_rays"] - c0["shadow_rays"]
print(f"{rays/dt/1e6:.0f} Mray/s | per launch ms:", {k: round(v["ms"] / max(v["launches"], 1), 3) for k, v in kt.items() if v["launches"]},
      "| trace ns per (ray+shadow ray) %.3f" % (kt["trace"]["ms"] * 1e6 / (rays + srays)))
r.close()
