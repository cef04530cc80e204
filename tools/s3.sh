#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8
SNIPER_WGRAD_DEFER=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference > gpurun_out/bench_defer1.log 2>&1; tail -n 1 gpurun_out/bench_defer1.log | cut -c1-600
SNIPER_WGRAD_DEFER=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference > gpurun_out/bench_defer0.log 2>&1; tail -n 1 gpurun_out/bench_defer0.log | cut -c1-300
SNIPER_WGRAD_DEFER=0 SNIPER_WGRAD_IMPL=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference > gpurun_out/bench_old.log 2>&1; tail -n 1 gpurun_out/bench_old.log | cut -c1-300
