#!/usr/bin/env python3
"""bench.py — Mray/s (primary+bounce) of the Renderer::Accumulate hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1], "cfg2"): synthetic scene S(1000) (ground + 999 random spheres, 17
materials, 15 emissive spheres, ambient 0.5), SAH BVH, MIS on, Policy.max_bounces = 5 (primary + 4 bounces),
1024x1024 pixels, 64 accumulations.  One STEP = one such 64-accumulation pass (64 x Renderer::Accumulate()).
N > 1: weak scaling — the image grows to N x 1024^2 pixels and rank r renders the r-th contiguous range of 4096
tiles (no data-path collective; one RCCL gather of the accumulator slabs after the timed region, timed separately).

value = rays handed to closest-hit traversal by all ranks (Renderer.hpp:165: primary + extension rays; shadow rays
are reported separately) / max-over-ranks wall time of the K timed steps, inputs resident in HBM.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
LDS_PEAK_GBPS = 150000.0        # MI355X_MICROARCH.md §LDS: ~150 TB/s aggregate for ds_read_b64/b128 with every CU streaming


def algorithmic_bytes(c):
    """SURVEY.md §8d, for the k_trace launches (closest-hit rays of bounce b + shadow rays of bounce b-1 in one kernel):
    per closest-hit ray 24 B (p,dir) + 12 B (tfar,primID,matID) + 32 B per BVH child box tested + 32 B per sphere tested;
    per shadow ray 28 B (p,dir,tfar) + 1/8 B (occluded bit) + 32 B per box + 32 B per sphere."""
    return (36 * c["rays"] + 32 * (c["nodes"] + c["spheres"])
            + 28.125 * c["shadow_rays"] + 32 * (c["shadow_nodes"] + c["shadow_spheres"]))


def shape_for(n_gpus, base=1024):
    w = h = base
    k = n_gpus
    while k > 1:                      # double width, then height, ...: 1024x1024, 2048x1024, 2048x2048, 4096x2048
        if w <= h:
            w *= 2
        else:
            h *= 2
        k //= 2
    assert (w // 16) * (h // 16) == n_gpus * (base // 16) ** 2, "n_gpus must be a power of two"
    return w, h


def cpu_baseline(mirt, scene_fn, cfg, log):
    """The reference's own CPU path, as restated in oracle/ (the reference itself cannot be built here): stream-BVH
    traversal (BVH.hpp:320-358) with the AVX2 8-ray sphere kernel, tiles over all host threads — on a bounded sample
    (a few accumulations of the same 1024x1024 workload; rays/s does not depend on the accumulation count)."""
    import oracle_binding as ob
    # host threads this job may use: the GPU box gives a 1-GPU job a 16-CPU share, whatever the machine has
    threads = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    o = ob.Oracle(scene_fn(), max_bounces=cfg["max_bounces"], buckets=cfg["buckets"], trav_mode=ob.TRAV_STREAM_BVH, threads=threads)
    o.Resize(cfg["width"], cfg["height"])
    t0 = time.perf_counter(); o.Accumulate(1); t1 = time.perf_counter() - t0
    n = min(16, max(3, int(15.0 / max(t1, 1e-3))))
    o.ResetAccumulator()
    t0 = time.perf_counter(); o.Accumulate(n); dt = time.perf_counter() - t0
    rays = o.counters()["rays"]
    log(f"cpu baseline: {n} accumulations, {rays} rays in {dt:.2f}s on {threads} threads")
    out = {"value": rays / dt / 1e6, "unit": "Mray/s", "cores": threads, "kind": "port",
           "sample": f"{n} of {cfg['spp']} accumulations of the same {cfg['width']}x{cfg['height']} S({cfg['n']}) workload, "
                     f"oracle stream-BVH mode (reference BVH.hpp:320-358 restated; reference itself unbuildable here)"}
    # the reference AS SHIPPED traverses nothing (#define USEBVH false, BVH.hpp:307): brute force over all spheres (SURVEY.md §8d asks for both)
    try:
        b = ob.Oracle(scene_fn(), max_bounces=cfg["max_bounces"], buckets=cfg["buckets"], trav_mode=ob.TRAV_BRUTE, threads=threads)
        win = 256 if cfg["n"] <= 2000 else 128 if cfg["n"] <= 20000 else 48          # cost grows with the sphere count: keep the sample to seconds
        b.Resize(win, win)
        t0 = time.perf_counter(); b.Accumulate(2); dtb = time.perf_counter() - t0
        out["as_shipped_brute_force"] = {"value": b.counters()["rays"] / dtb / 1e6, "unit": "Mray/s", "cores": threads,
                                         "sample": f"2 accumulations of a {win}x{win} window of the same scene and camera (USEBVH false: every ray tests all {cfg['n']} spheres)"}
        b.close()
    except Exception as e:                                   # the headline baseline above is what the contract needs
        log(f"brute-force cpu baseline skipped: {e}")
    o.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=None, help="accumulations per step (default: cfg2's 64)")
    ap.add_argument("--streams", type=int, default=0, help="batches in flight on separate HIP streams (0 = library default, 3)")
    ap.add_argument("--max-batch", type=int, default=0, help="Accumulate() calls traced together as one batch (0 = library default: about 32 M primary rays)")
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"],
                    help="BASELINE config. cfg2 (default, the metric's config) scales weakly: 1024x1024 px per GPU. The others keep their own image "
                         "(cfg3 1920x1088, cfg4/5 4096x4096) and split its tile rows over the ranks (strong scaling); --spp bounds the accumulations per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-counts", action="store_true", help="skip the counting replay (roofline.achieved becomes null)")
    args = ap.parse_args()

    import torch
    mirt = importlib.import_module("cpu-raytracing-experiments_amd")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
        args.gpus = world
    # Rehearsal knobs (single-GPU box): MIRT_BENCH_SHARE_GPU=1 puts every rank on cuda:0, MIRT_BENCH_BACKEND=gloo replaces RCCL
    # (which refuses two ranks on one device); the gather then stages through host memory.  Never set by the driver.
    backend = os.environ.get("MIRT_BENCH_BACKEND", "nccl")
    if os.environ.get("MIRT_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    comm_device = "cuda" if backend == "nccl" else "cpu"

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    cfg = dict(mirt.scene.CONFIGS[args.config])
    weak = args.config == "cfg2"
    if args.spp:
        cfg["spp"] = args.spp
    elif not weak:
        cfg["spp"] = 64                                      # one step = 64 accumulations whatever the config's total
    scene_fn = lambda: mirt.scene.synthetic(cfg["n"], ambient=cfg["ambient"])   # noqa: E731
    width, height = shape_for(world, cfg["width"]) if weak else (cfg["width"], cfg["height"])
    tiles = (width // 16) * (height // 16)
    h_tiles, v_tiles = width // 16, height // 16
    first_row, row_stride, n_rows = mirt.distributed.tile_rows(v_tiles, rank, world)     # interleaved tile rows: every rank sees sky and ground alike
    count = n_rows * h_tiles

    r = mirt.Renderer(scene_fn(), device=local_rank, max_bounces=cfg["max_bounces"], buckets=cfg["buckets"], mis=True,
                      use_bvh=bool(cfg["use_bvh"]), profile=True, streams=args.streams, max_batch=args.max_batch)
    n_streams = args.streams or 3
    r.Resize(width, height)
    if world > 1:
        r.SetTileRows(first_row, row_stride)
    spp, K, W = cfg["spp"], args.steps, args.warmup

    def sync_all():
        r.Synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- timed pass: W warmup steps, then exactly K steps ----
    for _ in range(W):
        r.Accumulate(spp)
    r.kernel_times(reset=True)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(K):
        r.AccumulateAsync(spp)         # one step = one 64-accumulation call, enqueued like the reference's frame loop; the K steps are
    sync_all()                         # bracketed by device synchronisation on both sides, not separated by it
    elapsed = time.perf_counter() - t0
    ktimes = r.kernel_times(reset=True)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- one gather of the accumulated radiance to rank 0 (RCCL over xGMI), outside the timed region ----
    gather_ms = None
    if dist is not None:
        local = mirt.distributed.device_tensor(*r.accumulator_device(), shape=(count, cfg["buckets"], 3, 256))
        if comm_device == "cpu":
            local = local.cpu()
        sync_all()
        g0 = time.perf_counter()
        full = mirt.distributed.gather_accumulator_rows(local, h_tiles, v_tiles, rank, world, cfg["buckets"])
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        if rank == 0:
            assert tuple(full.shape) == (tiles, cfg["buckets"], 3, 256)

    # ---- roofline pass: the same W+K steps with ONE batch in flight, so every launch has the GPU to itself and its HIP-event
    #      duration is the kernel's own (in the pipelined pass above a launch's duration includes time shared with the
    #      other streams' kernels; those figures are reported as well, as "overlapped") ----
    ktimes_serial = None
    if n_streams != 1 and not args.no_counts:
        r.set_policy(streams=1, profile=1, count_traffic=0)
        r.ResetAccumulator()
        for _ in range(W):
            r.Accumulate(spp)
        r.kernel_times(reset=True)
        for _ in range(K):
            r.Accumulate(spp)
        ktimes_serial = r.kernel_times(reset=True)

    # ---- counting replay of the same steps: rays / nodes / spheres of exactly the timed accumulation indices ----
    counts = None
    if not args.no_counts:
        r.set_policy(count_traffic=1, profile=0)
        r.ResetAccumulator()
        for _ in range(W):
            r.Accumulate(spp)
        c0 = r.counters()
        for _ in range(K):
            r.Accumulate(spp)
        c1 = r.counters()
        counts = {k: c1[k] - c0[k] for k in c1}
    rays_local = counts["rays"] if counts else None
    if rays_local is None:                                  # rays are counted in every mode
        c = r.counters(); rays_local = c["rays"] * K // (K + W)
    rays_total = rays_local
    if dist is not None:
        t = torch.tensor([rays_local], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        rays_total = int(t.item())

    if rank == 0:
        value = rays_total / elapsed / 1e6
        roofline = None
        kt = ktimes_serial or ktimes
        tr = kt["trace"]
        if tr["launches"]:
            avg_ms = tr["ms"] / tr["launches"]
            roofline = {"bound": "hbm", "kernel": "k_trace", "achieved": None, "peak": HBM_PEAK_GBPS, "peak_measured_copy": 6290.0, "unit": "GB/s", "frac": None,
                        "traffic": None, "launches": tr["launches"], "avg_launch_ms": avg_ms,
                        "measured_with": ("second live pass of the same steps with streams=1 (one kernel on the GPU at a time)" if ktimes_serial
                                          else "the timed pass (streams=1)")}
            if counts:
                ab = algorithmic_bytes(counts)
                ach = ab / (tr["ms"] * 1e-3) / 1e9
                lds_bytes = 32.0 * (counts["nodes"] + counts["spheres"] + counts["shadow_nodes"] + counts["shadow_spheres"])
                roofline["note"] = ("SURVEY.md §8d prices every BVH child box and sphere fetched at 32 B against HBM; k_trace serves them from the tree it stages "
                                    "in LDS once per launch, so `achieved` can exceed the HBM peak: frac > 1 means HBM does not bound this kernel. Its real HBM "
                                    "bytes per launch are `traffic` (PMC); the box/sphere part of the algorithmic bytes against the LDS roof is `lds`; the kernel "
                                    "is bound by VALU issue and dependent LDS latency (DESIGN.md §4)")
                roofline["lds"] = {"achieved": lds_bytes / (tr["ms"] * 1e-3) / 1e9, "peak": LDS_PEAK_GBPS, "unit": "GB/s",
                                   "frac": lds_bytes / (tr["ms"] * 1e-3) / 1e9 / LDS_PEAK_GBPS}
                roofline.update(achieved=ach, frac=ach / HBM_PEAK_GBPS, algorithmic_bytes_per_launch=ab / tr["launches"],
                                nodes_per_ray=counts["nodes"] / counts["rays"], spheres_per_ray=counts["spheres"] / counts["rays"],
                                nodes_per_shadow_ray=counts["shadow_nodes"] / max(counts["shadow_rays"], 1))
                if ktimes_serial:
                    ov = ktimes["trace"]
                    roofline["overlapped"] = {"avg_launch_ms": ov["ms"] / ov["launches"], "achieved": ab / (ov["ms"] * 1e-3) / 1e9,
                                              "frac": ab / (ov["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                              "note": f"same launches during the value pass, {n_streams} batches in flight: durations include time shared with other kernels"}
            sh = kt.get("shade")
            if counts and sh and sh["launches"]:
                # second kernel: k_shade moves the ray streams and IS HBM-bound.  SURVEY.md §8d: 56 B read per ray shaded + 56 B written per
                # extension ray + 24 B of accumulator RMW per terminated path
                primary = K * spp * count * 256
                sb = 56.0 * counts["rays"] + 56.0 * (counts["rays"] - primary) + 24.0 * counts["terminated"]
                roofline["shade"] = {"bound": "hbm", "kernel": "k_shade", "achieved": sb / (sh["ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                     "frac": sb / (sh["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, "launches": sh["launches"], "avg_launch_ms": sh["ms"] / sh["launches"],
                                     "note": "stream I/O only; the shadow-ray records it also writes (up to 68 B per NEE ray) are not in the §8d figure"}
            if counts:
                # all algorithmic bytes of the job (traversal + 56 B written and 56 B read per extension ray + 24 B per terminated path)
                # over the job's wall time, against 8 TB/s per GPU
                ext = counts["rays"] - K * spp * count * 256
                job_bytes = algorithmic_bytes(counts) + 112.0 * ext + 24.0 * counts["terminated"]
                roofline["whole_job"] = {"achieved": job_bytes * world / elapsed / 1e9, "peak": HBM_PEAK_GBPS * world, "unit": "GB/s",
                                         "frac": job_bytes / elapsed / 1e9 / HBM_PEAK_GBPS,
                                         "note": "rank 0's algorithmic bytes x ranks over the timed pass (all kernels, batches overlapped)"}
            traffic_file = os.path.join(ROOT, "profiles", "r01", "pmc_hbm_traffic.json")
            if os.path.exists(traffic_file):
                try:
                    roofline["traffic"] = json.load(open(traffic_file))["kernels"]["mirt::k_trace<false>"]["hbm_bytes_per_launch"]
                    roofline["traffic_source"] = "profiles/r01/pmc_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950-corrected)"
                except Exception:
                    pass
        out = {
            "metric": "Mray/s (primary+bounce)", "value": value, "unit": "Mray/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{args.config}: S({cfg['n']}) spheres + SAH BVH, MIS, Policy.max_bounces={cfg['max_bounces']} "
                                    f"(primary+{cfg['max_bounces'] - 1} bounces), {spp} accumulations/step, "
                                    + ("1024x1024 px per GPU" if weak else f"{width}x{height} px over all GPUs")), "image": f"{width}x{height}",
                       "spp_per_step": spp, "spheres": cfg["n"], "max_bounces": cfg["max_bounces"], "buckets": cfg["buckets"],
                       "parallelism": f"tile rows interleaved over {world} GPUs, one RCCL gather" if world > 1 else "single GPU", "batches_in_flight": n_streams,
                       "accumulations_per_batch": min(r.get_policy()["max_batch"], spp)},
            "rays_per_step": rays_total / K,
            "shadow_rays_per_step": (counts["shadow_rays"] / K) if counts else None,
            "kernel_ms_per_step": {k: v["ms"] / K for k, v in (ktimes_serial or ktimes).items() if v["launches"]},
            "gather_ms": gather_ms,
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(mirt, scene_fn, cfg, log)
        elif world > 1:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    r.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
