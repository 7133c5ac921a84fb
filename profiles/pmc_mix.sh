#!/bin/bash
# Dynamic VALU instruction mix of one batch (bench.py --pmc-child) by rocprofv3's per-type instruction counters.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CFG=${1:-cfg4}; OUT=gpurun_out/pmc_mix_$CFG; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU2 --output-format csv -d $OUT/p -- python3 bench.py --pmc-child --config $CFG > $OUT/p.out 2> $OUT/p.err || { echo "mix pass failed"; tail -5 $OUT/p.err; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob('$OUT/p/*/*_counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if 'mirt::' in k: agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in agg.items():
    n = d['SQ_INSTS_VALU']
    if n < 1e6: continue
    full = d['SQ_INSTS_VALU_ADD_F32'] + d['SQ_INSTS_VALU_MUL_F32'] + d['SQ_INSTS_VALU_FMA_F32']
    print(k[:34].ljust(34), 'VALU %.3g' % n, ' '.join('%s %.3f' % (c.replace('SQ_INSTS_VALU_', '').replace('SQ_ACTIVE_INST_', 'ACT_'), v / n) for c, v in sorted(d.items()) if c != 'SQ_INSTS_VALU'),
          '| add+mul+fma %.3f trans %.3f' % (full / n, d['SQ_INSTS_VALU_TRANS_F32'] / n))
PY
