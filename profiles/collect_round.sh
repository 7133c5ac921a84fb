#!/bin/bash
# Collects the round's judged artefacts on the GPU box into gpurun_out/final/ (copied to profiles/rNN/ afterwards).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/final; mkdir -p gpurun_out/final
timeout -k 10 400 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; echo "bench rc=$?"; tail -1 gpurun_out/final/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/kt_serial -- python3 bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline > gpurun_out/final/bench_serial_rocprof.json 2>/dev/null; echo "kt_serial $?"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/kt_pipe -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-counts > gpurun_out/final/bench_pipe_rocprof.json 2>/dev/null; echo "kt_pipe $?"
B="python3 bench.py --steps 1 --warmup 0 --spp 32 --streams 1 --no-cpu-baseline --no-counts"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/final/pmc_fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/final/pmc_write -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d gpurun_out/final/p1 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM --output-format csv -d gpurun_out/final/p2 -- $B > /dev/null 2>&1
echo "pmc done"; cat gpurun_out/final/bench.json
