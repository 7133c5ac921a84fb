"""Randomised differential soak on the GPU: random scenes, image sizes, policies (tree builder, record precision, batch size, streams,
bounces, buckets, MIS) — the BVH pipeline must reproduce the brute-force pipeline bit for bit every time, and repeated runs must be
reproducible.  usage: soak.py [seconds]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mirt = importlib.import_module("cpu-raytracing-experiments_amd")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
t_end = time.time() + budget
it = 0; rays = 0
while time.time() < t_end:
    n = int(rng.choice([2, 5, 33, 200, 1000, 3000, 12000, 70000]))
    sc = mirt.scene.synthetic(n, ambient=float(rng.choice([0.0, 0.5])), scene_seed=int(rng.integers(1, 1 << 30)))
    if rng.random() < 0.3:                                   # squash the scene onto a plane / a line now and then
        sc.geometry["position"][1:, int(rng.integers(0, 3))] = 0.5
    if rng.random() < 0.6:                                   # a random camera: anywhere in or around the sphere field (also inside spheres), any direction, wide to narrow lens
        L = float(2.0 * np.cbrt(n))
        eye = rng.uniform(-1.5 * L, 1.5 * L, 3); eye[1] = rng.uniform(-0.5, 1.5 * L)
        if rng.random() < 0.2 and n > 2:
            eye = sc.geometry["position"][int(rng.integers(1, n))] + rng.normal(size=3) * 0.3      # next to / inside a sphere
        d = rng.normal(size=3)
        sc.camera = mirt.scene.Camera(eye=tuple(float(v) for v in eye), direction=tuple(float(v) for v in d / np.linalg.norm(d)),
                                      focal_length=float(rng.choice([12.0, 24.0, 40.0, 85.0, 200.0])), exposure=1.0)
    if rng.random() < 0.35:                                  # the same scene at another scale / far from the origin: f32 error margins of the cone and cover tests
        k = float(rng.choice([0.01, 0.1, 8.0, 50.0])); off = (rng.normal(size=3) * float(rng.choice([0.0, 100.0, 3000.0]))).astype(np.float32)
        sc.geometry["position"] = (sc.geometry["position"] * np.float32(k) + off).astype(np.float32)
        sc.geometry["radius_sq"] = (sc.geometry["radius_sq"] * np.float32(k * k)).astype(np.float32)
        sc.camera.pos = (sc.camera.pos * np.float32(k) + off).astype(np.float32)
    w, h = int(rng.choice([64, 96, 208, 512, 1024])), int(rng.choice([48, 64, 160, 256, 768]))
    mb, buckets, mis = int(rng.integers(1, 10)), int(rng.choice([1, 3, 5, 8, 16])), bool(rng.random() < 0.8)
    spp = int(rng.integers(1, 24))
    if w * h <= 96 * 64 and rng.random() < 0.3: spp = int(rng.integers(65, 200))      # a small image leaves room for many batch slots: one batch of up to 199 accumulations
    ref = mirt.Renderer(sc, max_bounces=mb, buckets=buckets, mis=mis, use_bvh=False, streams=1, max_batch=int(rng.choice([0, 1, 5])))
    ref.Resize(w, h); ref.Accumulate(spp); want = ref.accumulator(); cw = ref.counters(); ref.close()
    kw = dict(gpu_build=bool(rng.random() < 0.5), reference_tree=bool(rng.random() < 0.2), allow_half_boxes=bool(rng.random() < 0.7),
              streams=int(rng.choice([0, 1, 2, 3, 5])), max_batch=int(rng.choice([0, 0, 1, 3, 7, 32, 64, 200])), trace_primary_rays=bool(rng.random() < 0.15))
    r = mirt.Renderer(sc, max_bounces=mb, buckets=buckets, mis=mis, use_bvh=True, **kw); r.Resize(w, h)
    left = spp
    while left:                                              # split the calls randomly, mixing sync and async
        k = int(rng.integers(1, left + 1)); left -= k
        (r.AccumulateAsync if rng.random() < 0.5 else r.Accumulate)(k)
    got = r.accumulator(); cg = r.counters(); r.close()
    ok = np.array_equal(got.view(np.uint32), want.view(np.uint32)) and cg["rays"] == cw["rays"] and cg["terminated"] == cw["terminated"]
    it += 1; rays += cg["rays"]
    if not ok:
        print(f"MISMATCH at iteration {it}: n={n} {w}x{h} mb={mb} buckets={buckets} mis={mis} spp={spp} {kw}", flush=True)
        sys.exit(1)
    if it % 10 == 0:
        print(f"{it} scenes ok, {rays/1e6:.0f} M rays", flush=True)
print(f"soak passed: {it} random scene/policy combinations, {rays/1e6:.0f} M rays, BVH == brute force bit for bit")
