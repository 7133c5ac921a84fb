#!/bin/bash
# Edit-measure loop on the GPU box: GPU tests, the default bench line (cfg4, with its rocprofv3 --pmc child passes), then short
# A/B runs of launch-shape knobs.  A step that times out ends the cycle (no further GPU work after a hang).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
step() { name=$1; shift; timeout -k 10 "$@"; rc=$?; echo "[$name] rc=$rc" >&2; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] timed out: stopping" >&2; exit 1; fi; return 0; }
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  step pytest 1100 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/pytest_gpu.log 2>&1; tail -15 gpurun_out/pytest_gpu.log
fi
step bench 500 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_cur.json 2> gpurun_out/bench_cur.err; tail -3 gpurun_out/bench_cur.err
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/bench_cur.json')); r = d['roofline'] or {}
    print('cfg4 Mray/s %.1f  ms/step %.2f' % (d['value'], d['ms_per_step']), {k: round(v, 2) for k, v in (d['kernel_ms_per_step'] or {}).items()})
    print('  valu frac', r.get('frac'), 'at clock', r.get('clock_GHz_during_k_trace'), r.get('issue_frac_at_that_clock'), 'lane util', r.get('lane_utilisation'), 'valu/ray', r.get('valu_wave_instructions_per_traced_ray'), 'cyc/inst', r.get('simd_cycles_per_valu_instruction'), 'dual', r.get('dual_issue_share'))
    print('  lanes', r.get('lane_instructions', {}).get('frac'), 'mix', r.get('instruction_mix'), 'boxes', r.get('boxes_per_ray'), r.get('boxes_per_shadow_ray'))
    print('  hbm', {k: v for k, v in (r.get('hbm') or {}).items() if k != 'note'})
    print('  shade', {k: v for k, v in (r.get('shade') or {}).items() if k != 'note'})
    print('  cpu', d.get('cpu_baseline'))
except Exception as e:
    print('bench parse failed', e)
PY
for V in "$@"; do
  # V = config:ENV1=a,ENV2=b[:extra bench.py arguments separated by commas]
  CFG=$(echo "$V" | cut -d: -f1); ENVS=$(echo "$V" | cut -d: -f2); EXTRA=$(echo "$V" | cut -s -d: -f3 | tr ',' ' ')
  step "ab $V" 300 env $(echo $ENVS | tr ',' ' ') python bench.py --config $CFG --steps 3 --warmup 1 --no-pmc --no-cpu-baseline $EXTRA > gpurun_out/ab.json 2> gpurun_out/ab.err
  python -c "
import json
try:
    d=json.load(open('gpurun_out/ab.json')); print('$V', 'Mray/s %.1f ms/step %.2f' % (d['value'], d['ms_per_step']), {k: round(v,2) for k,v in (d['kernel_ms_per_step'] or {}).items()})
except Exception as e: print('$V failed', e); print(open('gpurun_out/ab.err').read()[-500:])
"
done
