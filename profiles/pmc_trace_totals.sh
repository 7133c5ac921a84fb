cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/pmc_x; mkdir -p gpurun_out/pmc_x
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc_x/p1 -- python3 bench.py --steps 1 --warmup 0 --spp 32 --streams 1 --no-cpu-baseline --no-counts > /dev/null 2> gpurun_out/pmc_x/p1.err
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_x/p1/*/*_counter_collection.csv')[0]
agg = collections.defaultdict(float); us = 0.0; seen=set()
for r in csv.DictReader(open(f)):
    if 'k_trace<' not in r['Kernel_Name']: continue
    agg[r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen: seen.add(r['Dispatch_Id']); us += (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
print("k_trace total us %.0f" % us, {k: "%.4g" % v for k, v in agg.items()})
PY
