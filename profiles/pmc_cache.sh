#!/bin/bash
# Vector-memory counters of the trace kernels for one batch (bench.py --pmc-child): L1 (TCP) accesses / misses, L1 -> L2 requests, L2
# hits / misses, TA busy.   bash profiles/pmc_cache.sh cfg4   -> per-kernel sums on stdout
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CFG=${1:-cfg4}; OUT=gpurun_out/pmc_cache_$CFG; rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --pmc-child --config $CFG"
rocprofv3 -L > $OUT/avail.txt 2>&1
i=0
for SET in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM" ; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- $B > $OUT/p$i.out 2> $OUT/p$i.err || { echo "pass $i ($SET) failed"; tail -2 $OUT/p$i.err; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float)
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot):
    if "mirt" in k: print(k, {c: "%.4g" % v for c, v in sorted(tot[k].items())})
PY
