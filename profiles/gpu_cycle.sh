cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cur.json 2> gpurun_out/bench_cur.err; echo "bench rc=$?"; tail -2 gpurun_out/bench_cur.err
python -c "
import json; d=json.load(open('gpurun_out/bench_cur.json')); print('Mray/s %.1f  ms/step %.2f'%(d['value'], d['ms_per_step']), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()}, 'frac %.3f'%d['roofline']['frac'])"
rm -rf gpurun_out/pmc_cur; mkdir -p gpurun_out/pmc_cur; B="python3 bench.py --steps 1 --warmup 0 --spp 32 --streams 1 --no-cpu-baseline --no-counts"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc_cur/p1 -- $B > /dev/null 2> gpurun_out/pmc_cur/p1.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM --output-format csv -d gpurun_out/pmc_cur/p2 -- $B > /dev/null 2> gpurun_out/pmc_cur/p2.err
python profiles/pmc_by_dispatch.py gpurun_out/pmc_cur
