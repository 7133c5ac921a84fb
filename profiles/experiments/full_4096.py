"""4096x4096 on ONE GPU (2^24 pixels, the limit of the path id's pixel field): 100k spheres, 9 iterations, 10 accumulations."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
mirt = importlib.import_module("cpu-raytracing-experiments_amd")
t0 = time.perf_counter(); sc = mirt.scene.synthetic(100000); t1 = time.perf_counter()
r = mirt.Renderer(sc, max_bounces=9, use_bvh=True); t2 = time.perf_counter()
r.Resize(4096, 4096); t3 = time.perf_counter()
r.Accumulate(15); t4 = time.perf_counter(); c0 = r.counters(); r.Accumulate(20); t5 = time.perf_counter()
c = r.counters()
print(f"scene {t1-t0:.2f}s  renderer(BVH builds+upload) {t2-t1:.2f}s  resize {t3-t2:.2f}s  first 15 acc (touches all 3 slots) {t4-t3:.2f}s  next 20 acc {t5-t4:.3f}s")
print(c, "paths accounted:", c["terminated"] + c["dropped"] == 35 * 4096 * 4096, " Mray/s %.0f" % ((c["rays"] - c0["rays"]) / (t5 - t4) / 1e6))
assert r.Render()
fb = r.GetFrame(); print("frame", fb.shape, float(fb[..., :3].mean()), bool(np.isfinite(fb).all()))
acc = r.accumulator(); print("accumulator", acc.shape, float(acc.sum()))
r.close()
