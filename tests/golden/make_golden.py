"""Generates tests/golden/*.npz from the CPU oracle (oracle/oracle.cpp).

The reference ships no tests, fixtures or golden vectors, and cannot be built or run here, so
these vectors are outputs of OUR restatement of the reference source ("parity unpinned" at the glm/VCL/MSVC
boundary).  They pin the oracle against regressions and give the GPU tests fixed expected values.  The RNG
known-answer values in test_oracle_cpu.py are independent of this script (re-derived from Random.hpp formulas).

    python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_binding as ob  # noqa: E402

mirt_scene = importlib.import_module("cpu-raytracing-experiments_amd.scene")


def render_case(scene, w, h, spp, max_bounces, buckets=5, mis=True, trav=ob.TRAV_BRUTE):
    o = ob.Oracle(scene, max_bounces=max_bounces, buckets=buckets, mis=mis, trav_mode=trav, threads=1)
    o.Resize(w, h)
    o.Accumulate(spp)
    img = o.Render()
    c = o.counters()
    return dict(accumulator=o.accumulator(), frame=img if img is not None else np.zeros(0, np.float32),
                rays=np.uint64(c["rays"]), shadow_rays=np.uint64(c["shadow_rays"]), terminated=np.uint64(c["terminated"]))


def main():
    rng = np.random.default_rng(20261004)
    lib = ob.load()
    import ctypes as C
    # per-function vectors
    x = np.concatenate([rng.uniform(-10, 10, 2000), rng.uniform(0, 6.2832, 2000), [0.0, 1e-30, 3.1415927, 6.2831855, -0.0]]).astype(np.float32)
    s = np.empty_like(x); c = np.empty_like(x)
    for i, v in enumerate(x):
        a, b = C.c_float(), C.c_float()
        lib.orc_fast_sincos(float(v), C.byref(a), C.byref(b)); s[i], c[i] = a.value, b.value
    ya = rng.uniform(-2, 2, 2000).astype(np.float32); xa = rng.uniform(-2, 2, 2000).astype(np.float32)
    at = np.array([lib.orc_fast_atan2(float(p), float(q)) for p, q in zip(ya, xa)], dtype=np.float32)
    xs = np.concatenate([rng.uniform(-1, 1, 2000), [1.0, -1.0, 0.0, 1.5]]).astype(np.float32)
    asn = np.array([lib.orc_fast_asin(float(v)) for v in xs], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "math_vectors.npz"), sincos_x=x, sincos_s=s, sincos_c=c, atan2_y=ya, atan2_x=xa, atan2=at, asin_x=xs, asin=asn)

    cases = {
        "default9_64x64_10spp_b16": render_case(mirt_scene.default9(), 64, 64, 10, 16),
        "default9_64x64_5spp_nomis": render_case(mirt_scene.default9(), 64, 64, 5, 16, mis=False),
        "furnace_32x32_5spp": render_case(mirt_scene.white_furnace(), 32, 32, 5, 16),
        "S8_cfg1_64x64_1spp_b2": render_case(mirt_scene.synthetic(8, ambient=0.5), 64, 64, 1, 2),
        "S1000_64x64_5spp_b5": render_case(mirt_scene.synthetic(1000, ambient=0.5), 64, 64, 5, 5),
        "S1000_48x32_16buckets_16spp_b3": render_case(mirt_scene.synthetic(1000, ambient=0.0), 48, 32, 16, 3, buckets=16),
    }
    for name, d in cases.items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, {k: (v.shape if hasattr(v, "shape") and v.shape else int(v)) for k, v in d.items()})
    # BVH of S(1000): node array hash + first nodes
    o = ob.Oracle(mirt_scene.synthetic(1000), trav_mode=0)
    nodes, prims = o.bvh()
    np.savez_compressed(os.path.join(HERE, "S1000_bvh.npz"), nodes=nodes.view(np.uint8).reshape(-1, 32), prims=prims.view(np.uint8).reshape(-1, 32), lights=o.lights())


if __name__ == "__main__":
    main()
