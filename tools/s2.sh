#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
timeout 600 python tools/wgrad_batch_bench.py --trace > gpurun_out/wgrad_batch.txt 2>&1
cat gpurun_out/wgrad_batch.txt | cut -c1-420
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py tests/test_gpu_nn_ops.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
