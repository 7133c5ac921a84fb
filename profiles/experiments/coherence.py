"""How much would ray ORDER buy k_trace?  cfg4 scene: the bounce-1 rays and the bounce-0 NEE rays of one accumulation (origins = primary hit
points, stream order = pixel order with misses removed, like the compacted stream), traced by the product's kernel through the stage-level
entry points in several orders: as emitted / sorted by direction octant inside 512-ray blocks (what k_shade's compaction could do for free) /
inside 4096-ray blocks / globally by (origin cell, octant).  Directions use numpy's RNG (timing only, no parity claim)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mirt = importlib.import_module("cpu-raytracing-experiments_amd")
cfg = mirt.scene.CONFIGS["cfg4"]
sc = mirt.scene.synthetic(cfg["n"], ambient=0.0)
r = mirt.Renderer(sc, max_bounces=9, buckets=5, use_bvh=True, profile=True)
W = int(os.environ.get("COH_SIZE", "4096"))
r.Resize(W, W)
rng = np.random.default_rng(7)
prims = r.prims; ppos = prims["position"].astype(np.float32); prad2 = prims["radius_sq"].astype(np.float32)
geo = r.geometry; lights = r.lights


def timed(fn, *a):
    r.kernel_times(reset=True)
    out = fn(*a)
    t = r.kernel_times(reset=True)["trace"]
    return out, t["ms"]


def cosine_dirs(N):
    m = len(N)
    u0, u1 = rng.random(m, dtype=np.float32), rng.random(m, dtype=np.float32)
    rr, ph = np.sqrt(u0), 2 * np.pi * u1
    a = np.where(np.abs(N[:, 0:1]) > 0.9, np.array([[0, 1, 0]], np.float32), np.array([[1, 0, 0]], np.float32))
    T = np.cross(N, a); T /= np.linalg.norm(T, axis=1, keepdims=True)
    B = np.cross(N, T)
    return (T * (rr * np.cos(ph))[:, None] + B * (rr * np.sin(ph))[:, None] + N * np.sqrt(np.maximum(0, 1 - u0))[:, None]).astype(np.float32)


def hits_to_frame(p, d, tf, prim):
    hit = prim >= 0
    P = (p[:, hit] + d[:, hit] * tf[hit]).T
    N = P - ppos[prim[hit]]; N /= np.linalg.norm(N, axis=1, keepdims=True)
    flip = (N * d[:, hit].T).sum(1) >= 0
    N[flip] = -N[flip]
    return (P + N * 1e-4).astype(np.float32), N.astype(np.float32)


def nee(P, N):
    m = len(P)
    li = rng.integers(0, len(lights), m)
    Lc = geo["position"][lights[li]]; Lr2 = geo["radius_sq"][lights[li]]
    Wc = Lc - P; dist = np.linalg.norm(Wc, axis=1)
    D = Wc / dist[:, None]
    ok = ((D * N).sum(1) > 0) & (dist * dist > Lr2)
    return P[ok], D[ok].astype(np.float32), (dist[ok] - np.sqrt(Lr2[ok])).astype(np.float32)


def octant(D):
    return ((D[:, 0] < 0).astype(np.uint32) | ((D[:, 1] < 0).astype(np.uint32) << 1) | ((D[:, 2] < 0).astype(np.uint32) << 2))


def block_sort(key, block):
    n = len(key)
    blk = np.arange(n, dtype=np.uint64) // block
    return np.argsort((blk << np.uint64(32)) | key.astype(np.uint64), kind="stable")


def cell_key(P, D, cell):
    q = np.clip(((P - P.min(0)) / cell).astype(np.uint64), 0, 1023)
    return (((q[:, 0] << np.uint64(20)) | (q[:, 2] << np.uint64(10)) | q[:, 1]) << np.uint64(3)) | octant(D).astype(np.uint64)


def run_orders(name, P, D, tf=None):
    orders = {"as emitted": np.arange(len(P)), "octant in 512-blocks": block_sort(octant(D), 512), "octant in 4096-blocks": block_sort(octant(D), 4096),
              "octant in 64k-blocks": block_sort(octant(D), 65536),
              "global (4-unit cell, octant)": np.argsort(cell_key(P, D, 4.0), kind="stable"), "random": rng.permutation(len(P))}
    for oname, o in orders.items():
        p_, d_ = np.ascontiguousarray(P[o].T), np.ascontiguousarray(D[o].T)
        best = 1e9
        for _ in range(2):
            if tf is None:
                _, ms = timed(r.debug_trace_closest, p_, d_)
            else:
                _, ms = timed(r.debug_trace_shadow, p_, d_, np.ascontiguousarray(tf[o]))
            best = min(best, ms)
        print(f"{name:28s} {oname:30s} {len(P) / 1e6:7.2f} M rays  {best:8.3f} ms  {len(P) / best / 1e3:8.1f} Mray/s", flush=True)


p, d = r.debug_raygen(1)
(tf, prim), ms = timed(r.debug_trace_closest, p, d)
print(f"primary: {p.shape[1] / 1e6:.1f} M rays, {ms:.2f} ms, hit frac {(prim >= 0).mean():.3f}", flush=True)
P1, N1 = hits_to_frame(p, d, tf, prim)
del p, d
D1 = cosine_dirs(N1)
run_orders("bounce-1 closest", P1, D1)
Ps, Ds, ts = nee(P1, N1)
run_orders("bounce-0 shadow", Ps, Ds, ts)
# bounce 2: trace bounce 1, survivors in emission order
p1, d1 = np.ascontiguousarray(P1.T), np.ascontiguousarray(D1.T)
(tf1, prim1), _ = timed(r.debug_trace_closest, p1, d1)
keep = rng.random(len(prim1)) < 0.75          # Russian roulette, roughly
sel = (prim1 >= 0) & keep
P2, N2 = hits_to_frame(p1[:, sel], d1[:, sel], tf1[sel], prim1[sel])
D2 = cosine_dirs(N2)
run_orders("bounce-2 closest", P2, D2)
Ps2, Ds2, ts2 = nee(P2, N2)
run_orders("bounce-1 shadow", Ps2, Ds2, ts2)
r.close()
