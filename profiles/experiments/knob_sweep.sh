cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { env $1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-counts > gpurun_out/ab.json 2> gpurun_out/ab.err && python -c "
import json
d=json.load(open('gpurun_out/ab.json')); print('$1', 'Mray/s %.1f ms/step %.2f' % (d['value'], d['ms_per_step']))
"; }
for V in X=0 MIRT_TUNE_LEAF_BATCH=8 MIRT_TUNE_LEAF_BATCH=12 MIRT_TUNE_LEAF_BATCH=20 MIRT_TUNE_LEAF_BATCH=24 MIRT_TUNE_REFILL_IDLE=16 MIRT_TUNE_REFILL_IDLE=24 MIRT_TUNE_REFILL_IDLE=40 MIRT_TUNE_CHUNK=256 MIRT_TUNE_CHUNK=1024 MIRT_TUNE_SHADE_WGS=4 MIRT_TUNE_SHADE_WGS=6 X=1; do run $V; done
