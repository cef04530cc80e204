#!/bin/bash
# One short GPU visit for the pixel-stationary 1 x 1 kernel (csrc/conv_px.hip): bit-equality tests, per-layer A/B, whole-step A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 400 python -m pytest tests/test_gpu_nn_ops.py -m gpu -q -x -k "conv_px" -p no:cacheprovider > gpurun_out/px_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/px_tests.log
tail -5 gpurun_out/px_tests.log
timeout 300 python tools/conv_px_ab.py 40 > gpurun_out/px_ab.log 2>&1
echo "ab exit $?" >> gpurun_out/px_ab.log
cat gpurun_out/px_ab.log | tail -8
for PX in 0 2; do
  SNIPER_CONV_PX=$PX timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-inference --no-fit-path > gpurun_out/px_bench_$PX.log 2>&1
  echo "bench px=$PX exit $?"
  tail -1 gpurun_out/px_bench_$PX.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = d.get('roofline') or {}
print('px=$PX value', d.get('value'), 'ms_per_step', d.get('ms_per_step'), 'frac', r.get('frac'), 'conv_ms', r.get('conv_ms_per_step'))
"
done
