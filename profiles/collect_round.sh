#!/bin/bash
# Collects the round's judged artefacts on the GPU box into gpurun_out/final/ (copied to profiles/rNN/ by store_round.py).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/final; mkdir -p gpurun_out/final; F=gpurun_out/final
step() { name=$1; shift; timeout -k 10 "$@"; rc=$?; echo "[$name] rc=$rc" >&2; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] timed out: stopping" >&2; exit 1; fi; return 0; }
# 1. the default bench line (cfg4, with its in-run PMC child passes) and the other BASELINE configs
step bench 600 python bench.py > $F/bench.json 2> $F/bench.err; tail -2 $F/bench.err
for C in cfg2 cfg3 cfg5; do step bench_$C 400 python bench.py --config $C --no-cpu-baseline > $F/bench_$C.json 2> $F/bench_$C.err; done
# 2. rocprofv3 --kernel-trace --stats of the same command (per-kernel average durations; HIP-event figures must agree)
for C in cfg4 cfg3; do
  step kt_$C 600 rocprofv3 --kernel-trace --stats --output-format csv -d $F/kt_$C -- python3 bench.py --config $C --steps 2 --warmup 1 --no-pmc --no-cpu-baseline > $F/bench_rocprof_$C.json 2> $F/kt_$C.err
done
# 3. SQ counters per dispatch of one batch and the VALU instruction mix, cfg4 and cfg3
for C in cfg4 cfg3; do
  step pmc_$C 700 bash profiles/pmc_profile.sh $C > $F/pmc_by_dispatch_$C.txt 2> $F/pmc_$C.err; cp -r gpurun_out/pmc_$C $F/ 2>/dev/null
  step mix_$C 400 bash profiles/pmc_mix.sh $C > $F/pmc_mix_$C.txt 2> $F/mix_$C.err
done
# 4. one rank's share of the strong-scaled cfg4 image (what each GPU of an 8-GPU run does)
step share 400 python profiles/experiments/rank_share.py 8 > $F/rank_share_8.txt 2> $F/share.err
# 5. the N > 1 line rehearsed on this box's one GPU (two ranks over gloo; rank 0's PMC share, CPU baseline, group host child) and bench.py --group
step two_ranks 600 env MIRT_BENCH_SHARE_GPU=1 MIRT_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29731 bench.py --gpus 2 --steps 2 --warmup 1 > $F/bench_two_ranks_rehearsal.json 2> $F/two_ranks.err
step group 300 env MIRT_BENCH_DEVICES=0,0 python bench.py --group --gpus 2 --steps 2 --warmup 1 > $F/bench_group_rehearsal.json 2> $F/group.err
step gpu_build 300 python profiles/experiments/gpu_build.py > $F/gpu_build.txt 2>&1
echo "collected"; head -c 600 $F/bench.json
