#!/usr/bin/env python3
"""Per-dispatch view of two rocprofv3 --pmc passes (p1: SQ_WAVES.. p2: LDS/VMEM..): lane utilisation, wait share, LDS conflicts."""
import collections, csv, glob, os, sys
root = sys.argv[1]
def load(f):
    rows = collections.defaultdict(dict)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "mirt" not in k: continue
        d = rows[(int(row["Dispatch_Id"]), k)]
        d[row["Counter_Name"]] = float(row["Counter_Value"]); d["us"] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    # dispatch ids of the two passes need not agree (other libraries launch kernels too): key by (n-th launch of this kernel, name)
    out, seen = {}, collections.Counter()
    for (_, k), d in sorted(rows.items()):
        out[(seen[k], k)] = d; seen[k] += 1
    return out
r1 = load(max(glob.glob(f"{root}/p1/*/*_counter_collection.csv"), key=os.path.getmtime)); r2 = load(max(glob.glob(f"{root}/p2/*/*_counter_collection.csv"), key=os.path.getmtime))
pat = sys.argv[2] if len(sys.argv) > 2 else "trace"
for (i, k), d in sorted(r1.items()):
    if pat not in k: continue
    e = r2.get((i, k), {})
    util = d["SQ_THREAD_CYCLES_VALU"] / max(d["SQ_ACTIVE_INST_VALU"] * 64, 1)
    print(i, k[:30].ljust(30), "us %7.1f" % d["us"], "VALU %.3g SALU %.3g LDS %.3g VMEM %.3g" % (d["SQ_INSTS_VALU"], d["SQ_INSTS_SALU"], e.get("SQ_INSTS_LDS", 0), e.get("SQ_INSTS_VMEM", 0)),
          "lane-util %.2f" % util, "wait %.2f" % (d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]), "issue-wait %.2f" % (e.get("SQ_WAIT_INST_ANY", 0) / max(d["SQ_WAVE_CYCLES"], 1)),
          "active %.2f" % (e.get("SQ_ACTIVE_INST_ANY", 0) / max(d["SQ_WAVE_CYCLES"], 1)), "lds-conflict %.2f" % (e.get("SQ_LDS_BANK_CONFLICT", 0) / max(e.get("SQ_LDS_IDX_ACTIVE", 1), 1)),
          "valu-busy %.2f" % (d["SQ_ACTIVE_INST_VALU"] * 4 / max(d["SQ_BUSY_CYCLES"], 1)))
