#!/usr/bin/env python3
"""Copies gpurun_out/final/ (see collect_round.sh) into profiles/rNN/ as small summaries."""
import collections, csv, glob, json, os, shutil, subprocess, sys
dst = sys.argv[1] if len(sys.argv) > 1 else "profiles/r01"
src = "gpurun_out/final"
os.makedirs(dst, exist_ok=True)
shutil.copy(f"{src}/bench.json", f"{dst}/bench.json")
shutil.copy(max(glob.glob(f"{src}/kt_serial/*/*_kernel_stats.csv"), key=os.path.getmtime), f"{dst}/kernel_stats_streams1.csv")
shutil.copy(max(glob.glob(f"{src}/kt_pipe/*/*_kernel_stats.csv"), key=os.path.getmtime), f"{dst}/kernel_stats_streams3.csv")
shutil.copy(f"{src}/bench_serial_rocprof.json", f"{dst}/bench_under_rocprof_streams1.json")
out = {"note": "rocprofv3 --pmc passes (separate runs, one counter each) of `python3 bench.py --steps 1 --warmup 0 --spp 32 --streams 1 --no-cpu-baseline --no-counts` "
               "(one batch of 32 accumulations, 1024x1024, S(1000): the launch sizes of the default bench run). sum_KB are KB summed over the launches as rocprofv3 reports them; "
               "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024/launches, the gfx950 correction of MI355X_MICROARCH.md §HBM (FETCH_SIZE reports half of wide coalesced reads).", "kernels": {}}
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for name in ("pmc_fetch", "pmc_write"):
    f = max(glob.glob(f"{src}/{name}/*/*_counter_collection.csv"), key=os.path.getmtime)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        a = agg[k][row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
for k, v in agg.items():
    if "mirt" not in k: continue
    e = {c: {"launches": a[0], "sum_KB": a[1]} for c, a in v.items()}
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        e["hbm_bytes_per_launch"] = (2 * v["FETCH_SIZE"][1] + v["WRITE_SIZE"][1]) * 1024 / v["FETCH_SIZE"][0]
    out["kernels"][k] = e
json.dump(out, open(f"{dst}/pmc_hbm_traffic.json", "w"), indent=1)
with open(f"{dst}/pmc_sq_by_dispatch.txt", "w") as f:
    f.write("# rocprofv3 --pmc SQ counters per k_trace dispatch (one batch, streams=1); lane-util = SQ_THREAD_CYCLES_VALU / (64*SQ_ACTIVE_INST_VALU)\n")
    f.write(subprocess.run([sys.executable, "profiles/pmc_by_dispatch.py", src], capture_output=True, text=True).stdout)
print(open(f"{dst}/pmc_sq_by_dispatch.txt").read())
b = json.load(open(f"{dst}/bench.json")); r = b["roofline"]
print("value %.1f Mray/s  ms/step %.2f  frac %.3f  avg_launch_ms %.4f  cpu %.2f" % (b["value"], b["ms_per_step"], r["frac"], r["avg_launch_ms"], b["cpu_baseline"]["value"]))
for fn in ("kernel_stats_streams1.csv",):
    for row in csv.DictReader(open(f"{dst}/{fn}")):
        if "mirt" in row["Name"]: print("   %-26s calls %5s avg %9.1f us" % (row["Name"].split("(")[0].replace("void ", "")[:26], row["Calls"], float(row["AverageNs"]) / 1e3))
