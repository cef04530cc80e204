#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof stats.  Everything lands in gpurun_out/.
# usage: tools/gpu_session.sh [tests|bench|micro|all]   (default all)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
ROOT=$(pwd)
WHAT=${1:-all}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/device.txt
nproc >> gpurun_out/device.txt
if [[ $WHAT == all || $WHAT == tests ]]; then
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -n 1 --timeout 900 > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
  tail -60 gpurun_out/pytest_gpu.log
  timeout 900 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
  echo "smoke exit $?" >> gpurun_out/smoke.log
  tail -5 gpurun_out/smoke.log
fi
if [[ $WHAT == all || $WHAT == micro ]]; then
  timeout 600 python tools/microbench.py > gpurun_out/microbench.log 2>&1
  echo "microbench exit $?" >> gpurun_out/microbench.log
  cat gpurun_out/microbench.log
fi
if [[ $WHAT == all || $WHAT == bench ]]; then
  timeout 1200 python bench.py --steps 20 --warmup 5 --inference > gpurun_out/bench.log 2>&1
  echo "bench exit $?" >> gpurun_out/bench.log
  tail -5 gpurun_out/bench.log
  cd /tmp && export TMPDIR=/tmp
  timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_bench" -o bench -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$ROOT/gpurun_out/rocprof_bench.log" 2>&1
  echo "rocprof exit $?"
  find "$ROOT/gpurun_out/prof_bench" -name "*stats*" | head
  F=$(find "$ROOT/gpurun_out/prof_bench" -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && head -40 "$F"
  T=$(find "$ROOT/gpurun_out/prof_bench" -name "*kernel_trace.csv" | head -1)
  [ -n "$T" ] && python "$ROOT/tools/trace_gaps.py" "$T" "$ROOT/gpurun_out/trace_gaps.json" --lo 0.45 --hi 0.75
  # the raw kernel trace is large; keep only the stats
  find "$ROOT/gpurun_out/prof_bench" -name "*kernel_trace.csv" -size +20M -delete
fi
if [[ $WHAT == pmc ]]; then
  # HBM traffic of the conv family: FETCH_SIZE and WRITE_SIZE need separate passes (TCC has 4 slots: 3 + 2)
  cd /tmp && export TMPDIR=/tmp
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 1200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$ROOT/gpurun_out/pmc_$C" -o pmc -- python "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$ROOT/gpurun_out/rocprof_pmc_$C.log" 2>&1
    echo "pmc $C exit $?"
  done
  cd "$ROOT" && python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_traffic.json
  find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +8M -delete
fi
