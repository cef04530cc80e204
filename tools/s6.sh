#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
for V in "18 16" "0 16" "18 12" "18 16"; do set -- $V
  SNIPER_CONV_PS=$1 SNIPER_CONV_PS_NK=$2 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-inference > gpurun_out/bench_ps_$1_$2.log 2>&1
  echo "PS $1 nk>=$2: $(tail -n 1 gpurun_out/bench_ps_$1_$2.log | cut -c1-160)"
done
rm -f gpurun_out/parity_quantiles.jsonl
timeout 900 python -m pytest tests/test_gpu_fp32_parity.py -m gpu -q -x -s -p no:cacheprovider 2>&1 | grep -v Warning | tail -25
