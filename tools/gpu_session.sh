#!/bin/bash
# One GPU-box visit.  Everything lands in gpurun_out/ (copy what should be judged into profiles/).
# usage: tools/gpu_session.sh [tests|bench|prof|pmc|pmcc4|pmcinf|micro]...
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
ROOT=$(pwd)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/device.txt
nproc >> gpurun_out/device.txt
# which physical card this visit ran on (a misbehaving box shows up as the same id across visits)
{ hostname; rocm-smi --showuniqueid --showserial --showbus 2>/dev/null | grep -E "Unique|Serial|PCI"; } >> gpurun_out/device.txt
for WHAT in "${@:-bench}"; do
case $WHAT in
tests)
  timeout 3000 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=10 > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
  tail -30 gpurun_out/pytest_gpu.log
  timeout 900 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
  echo "smoke exit $?" >> gpurun_out/smoke.log
  tail -3 gpurun_out/smoke.log ;;
micro)
  timeout 600 python tools/microbench.py > gpurun_out/microbench.log 2>&1
  echo "microbench exit $?" >> gpurun_out/microbench.log
  tail -20 gpurun_out/microbench.log ;;
bench)
  timeout 1200 python bench.py > gpurun_out/bench.log 2>&1
  echo "bench exit $?" >> gpurun_out/bench.log
  tail -2 gpurun_out/bench.log ;;
prof)
  # kernel stats of the SAME command (minus the CPU / inference legs): 3 + 10 + 1 + 3 = 17 steps of conv launches
  (cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_bench" -o bench -- \
      python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-inference --no-fit-path --no-c4 > "$ROOT/gpurun_out/rocprof_bench.log" 2>&1; echo "rocprof exit $?")
  F=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && head -30 "$F" && python tools/roofline_check.py "$F" gpurun_out/rocprof_bench.log 17 gpurun_out/roofline_check.json
  T=$(find gpurun_out/prof_bench -name "*kernel_trace.csv" | head -1)
  [ -n "$T" ] && python tools/trace_gaps.py "$T" gpurun_out/trace_gaps.json --lo 0.45 --hi 0.75
  find gpurun_out/prof_bench -name "*kernel_trace.csv" -size +30M -delete ;;
pmc)
  # HBM traffic: FETCH_SIZE and WRITE_SIZE need separate passes (TCC has 4 slots: 3 + 2); counters only, no other trace domain
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$ROOT/gpurun_out/pmc_$C" -o pmc -- \
        python "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-inference --no-fit-path --no-c4 > "$ROOT/gpurun_out/rocprof_pmc_$C.log" 2>&1; echo "pmc $C exit $?")
  done
  python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_traffic.json > /dev/null
  # (rates also over the un-countered durations of the `prof` step's kernel statistics, when that step ran in this visit)
  python tools/pmc_report.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_kernels.json "" \
      "$(find gpurun_out/prof_bench -name '*kernel_stats.csv' 2>/dev/null | head -1)" | head -60
  # (copy gpurun_out/pmc_traffic.json and gpurun_out/pmc_kernels.json to profiles/: bench.py reads them there, keyed on the source hash)
  find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +8M -delete ;;
pmcc4)
  # HBM traffic of the position-sensitive R-FCN head's kernels (BASELINE C4, tools/c4_bench.py): durations from an un-countered --stats
  # run, FETCH / WRITE from two counter passes of the same command -> gpurun_out/pmc_c4_kernels.json (copy to profiles/: bench.py's c4 leg)
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_c4" -o c4 -- \
      python "$ROOT/tools/c4_bench.py" 6 2 16 > "$ROOT/gpurun_out/prof_c4.log" 2>&1; echo "prof c4 exit $?")
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$ROOT/gpurun_out/pmcc4_$C" -o pmc -- \
        python "$ROOT/tools/c4_bench.py" 2 1 16 > "$ROOT/gpurun_out/rocprof_pmcc4_$C.log" 2>&1; echo "pmcc4 $C exit $?")
  done
  python tools/pmc_report.py gpurun_out/pmcc4_FETCH_SIZE gpurun_out/pmcc4_WRITE_SIZE gpurun_out/pmc_c4_kernels.json "" \
      "$(find gpurun_out/prof_c4 -name '*kernel_stats.csv' 2>/dev/null | head -1)" | grep -i "psroi_ps\|avgpool\|kernel" | head -12
  find gpurun_out/pmcc4_FETCH_SIZE gpurun_out/pmcc4_WRITE_SIZE gpurun_out/prof_c4 -name "*.csv" -size +8M -delete ;;
pmcinf)
  # HBM traffic of the inference pass's kernels: the same two counter passes over a 16-image AutoFocus pass,
  # durations from an un-countered --stats run of the same command
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_inf16" -o inf16 -- \
      python "$ROOT/tools/infer_profile.py" 1 3 - 1 16 - distinct > "$ROOT/gpurun_out/prof_inf16.log" 2>&1; echo "prof inf16 exit $?")
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$ROOT/gpurun_out/pmcinf_$C" -o pmc -- \
        python "$ROOT/tools/infer_profile.py" 1 3 - 1 16 - distinct > "$ROOT/gpurun_out/rocprof_pmcinf_$C.log" 2>&1; echo "pmcinf $C exit $?")
  done
  python tools/pmc_report.py gpurun_out/pmcinf_FETCH_SIZE gpurun_out/pmcinf_WRITE_SIZE gpurun_out/pmc_infer_kernels.json "" \
      "$(find gpurun_out/prof_inf16 -name '*kernel_stats.csv' 2>/dev/null | head -1)" | head -40
  find gpurun_out/pmcinf_FETCH_SIZE gpurun_out/pmcinf_WRITE_SIZE gpurun_out/prof_inf16 -name "*.csv" -size +8M -delete ;;
esac
done
