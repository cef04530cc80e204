#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
timeout 600 python tools/infer_profile.py 40 > gpurun_out/infer_profile.txt 2>&1; grep -v Warning gpurun_out/infer_profile.txt | head -70 | cut -c1-180
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_infer" -o infer -- python "$GRAFT_REPO_ROOT/tools/infer_profile.py" 5 > "$GRAFT_REPO_ROOT/gpurun_out/rocprof_infer.log" 2>&1)
head -25 gpurun_out/prof_infer/infer_kernel_stats.csv | cut -c1-160
find gpurun_out/prof_infer -name "*kernel_trace.csv" -size +20M -delete
timeout 1200 python -m pytest tests/test_gpu_inference.py tests/test_gpu_mask.py tests/test_gpu_nn_ops.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
