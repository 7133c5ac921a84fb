#!/usr/bin/env python3
"""Copies gpurun_out/final/ (see collect_round.sh) into profiles/rNN/ as small text summaries."""
import csv, glob, json, os, shutil, sys
dst = sys.argv[1] if len(sys.argv) > 1 else "profiles/r03"
src = "gpurun_out/final"
os.makedirs(dst, exist_ok=True)
for name in ("bench.json", "bench_cfg2.json", "bench_cfg3.json", "bench_cfg5.json", "pmc_by_dispatch_cfg4.txt", "pmc_by_dispatch_cfg3.txt", "pmc_mix_cfg4.txt", "pmc_mix_cfg3.txt", "rank_share_8.txt", "bench_group_rehearsal.json", "bench_two_ranks_rehearsal.json", "gpu_build.txt"):
    if os.path.exists(f"{src}/{name}"):
        shutil.copy(f"{src}/{name}", f"{dst}/{name}")
for cfg in ("cfg4", "cfg3"):
    files = glob.glob(f"{src}/kt_{cfg}/*/*_kernel_stats.csv")
    if files:
        shutil.copy(max(files, key=os.path.getmtime), f"{dst}/kernel_stats_{cfg}.csv")
    if os.path.exists(f"{src}/bench_rocprof_{cfg}.json"):
        shutil.copy(f"{src}/bench_rocprof_{cfg}.json", f"{dst}/bench_under_rocprof_{cfg}.json")
b = json.load(open(f"{dst}/bench.json")); r = b["roofline"]
print("value %.1f Mray/s  ms/step %.2f  useful-lane frac %.3f  issue frac %.3f (at %.2f GHz: %.3f)  arithmetic frac %.3f  lane util %.2f  avg k_trace launch %.3f ms  cpu %.2f Mray/s" % (
    b["value"], b["ms_per_step"], r["frac"], r["issue_frac"], r.get("clock_GHz_during_k_trace") or 0, r.get("issue_frac_at_that_clock") or 0, r["arithmetic_frac"], r["lane_utilisation"], r["avg_launch_ms"], b["cpu_baseline"]["value"]))
for cfg in ("cfg4", "cfg3"):
    fn = f"{dst}/kernel_stats_{cfg}.csv"
    if not os.path.exists(fn): continue
    print(cfg, "rocprofv3 --kernel-trace --stats:")
    for row in csv.DictReader(open(fn)):
        if "mirt" in row["Name"]:
            print("   %-34s calls %5s avg %10.1f us  total %8.1f ms" % (row["Name"].split("(")[0].replace("void ", "")[:34], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 1e6))
    rb = json.load(open(f"{dst}/bench_under_rocprof_{cfg}.json"))
    print("   bench under rocprof: HIP-event avg k_trace launch %.1f us" % (rb["roofline"]["avg_launch_ms"] * 1e3))
