#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
rm -f gpurun_out/parity_quantiles.jsonl
timeout 900 python -m pytest tests/test_gpu_nn_ops.py -m gpu -q -x -p no:cacheprovider -k "configurations and (18 or 19 or 20) or stats_epilogue or dgrad_bn_epilogue" 2>&1 | tail -6
timeout 900 python tools/conv_tune.py --cfgs 14,16,18,19,20 --cold 600 --insitu --only "s3" > gpurun_out/conv_tune_ps.txt 2>&1; cut -c1-200 gpurun_out/conv_tune_ps.txt | tail -12
timeout 900 python tools/conv_tune.py --cfgs 14,16,7,18,19,20 --cold 600 --insitu --only "s4" >> gpurun_out/conv_tune_ps.txt 2>&1
timeout 900 python tools/conv_tune.py --cfgs 14,16,7,18,19,20 --cold 600 --insitu --only "s2" >> gpurun_out/conv_tune_ps.txt 2>&1
for V in "14 16" "18 18" "18 16" "14 18"; do set -- $V
  SNIPER_CONV_BAL=$1 SNIPER_CONV_BAL_D=$2 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference > gpurun_out/bench_bal_$1_$2.log 2>&1
  echo "BAL $1 $2: $(tail -n 1 gpurun_out/bench_bal_$1_$2.log | cut -c1-160)"
done
timeout 900 python -m pytest tests/test_gpu_fp32_parity.py -m gpu -q -x -s -p no:cacheprovider 2>&1 | grep -v Warning | tail -25
