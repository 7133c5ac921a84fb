"""Per-launch floor of k_trace_closest: run the kernel on N incoherent rays (origins on sphere surfaces of S(1000),
cosine-ish random directions) for several N under rocprofv3 --kernel-trace and fit T = a + b*N."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mirt = importlib.import_module("cpu-raytracing-experiments_amd")
sc = mirt.scene.synthetic(1000, ambient=0.5)
r = mirt.Renderer(sc, use_bvh=True)
rng = np.random.default_rng(1)
geo = sc.geometry
for n in (64, 1024, 1 << 14, 1 << 18):
    pick = rng.integers(1, len(geo), n)
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    p = (geo["position"][pick] + nrm * (np.sqrt(geo["radius_sq"][pick])[:, None] * 1.001)).astype(np.float32).T
    d = nrm + rng.normal(size=(n, 3)) * 0.7; d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32).T
    for _ in range(3):
        t, i = r.debug_trace_closest(np.ascontiguousarray(p), np.ascontiguousarray(d))
    print(n, "hit frac %.2f" % (i >= 0).mean(), flush=True)
r.close()
