#!/bin/bash
# One GPU-box visit: parity tests, micro-benchmarks, rocprof stats.  Everything lands in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
ROOT=$(pwd)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/device.txt
nproc >> gpurun_out/device.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -n 1 --timeout 600 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
timeout 600 python tools/microbench.py > gpurun_out/microbench.log 2>&1
echo "microbench exit $?" >> gpurun_out/microbench.log
cat gpurun_out/microbench.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/prof_micro" -o micro -- python "$ROOT/tools/microbench.py" --quick > "$ROOT/gpurun_out/rocprof_micro.log" 2>&1
echo "rocprof exit $?"
find "$ROOT/gpurun_out/prof_micro" -name "*stats*" | head
