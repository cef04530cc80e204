#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
timeout 600 python tools/wgrad_batch_bench.py --trace > gpurun_out/wgrad_batch_na.txt 2>&1
cut -c1-400 gpurun_out/wgrad_batch_na.txt
timeout 600 python tools/wgrad_batch_bench.py --trace --job-steps 100000 --only "s3" > gpurun_out/wgrad_batch_na1.txt 2>&1
cut -c1-400 gpurun_out/wgrad_batch_na1.txt | grep -v amdgpu
timeout 900 python -m pytest tests/test_gpu_nn_ops.py tests/test_gpu_baseline_shapes.py -m gpu -q -x -p no:cacheprovider -k "wgrad or c2_launch or fc_new" 2>&1 | tail -4
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-inference > gpurun_out/bench_na.log 2>&1; tail -n 1 gpurun_out/bench_na.log | cut -c1-200
