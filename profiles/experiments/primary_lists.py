#!/usr/bin/env python3
"""Length histogram of the per-pixel candidate lists (k_primary_cand) for the BASELINE configs."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mirt = importlib.import_module("cpu-raytracing-experiments_amd")
for name in sys.argv[1:] or ["cfg2", "cfg3", "cfg4"]:
    cfg = mirt.scene.CONFIGS[name]
    r = mirt.Renderer(mirt.scene.synthetic(cfg["n"], ambient=cfg["ambient"]), max_bounces=cfg["max_bounces"], buckets=cfg["buckets"], use_bvh=True)
    r.Resize(cfg["width"], cfg["height"])
    h = r.debug_primary_lists(); n = sum(h)
    print(name, "pixels", n, "lists of 0..8:", " ".join(f"{v / n:.3f}" for v in h[:9]), "| no list (traced):", f"{h[9] / n:.3f}", flush=True)
    r.close()
