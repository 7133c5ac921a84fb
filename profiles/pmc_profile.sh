#!/bin/bash
# SQ counters per dispatch of one batch (bench.py --pmc-child: one batch, one kernel on the GPU at a time) for a BASELINE config.
#   bash profiles/pmc_profile.sh cfg4 [pattern]     -> gpurun_out/pmc_<cfg>/{p1,p2}, summary on stdout
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CFG=${1:-cfg4}; PAT=${2:-k_}; OUT=gpurun_out/pmc_$CFG; rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --pmc-child --config $CFG"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d $OUT/p1 -- $B > $OUT/p1.out 2> $OUT/p1.err || { echo "p1 failed"; tail -3 $OUT/p1.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM --output-format csv -d $OUT/p2 -- $B > $OUT/p2.out 2> $OUT/p2.err || { echo "p2 failed"; tail -3 $OUT/p2.err; exit 1; }
python3 profiles/pmc_by_dispatch.py $OUT "$PAT"
