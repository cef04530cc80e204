#!/bin/bash
# round-3 session 1: XCD-aware weight-gradient block mapping, A/B + fabric traffic
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
ROOT=$(pwd); mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
M="-1/-1/0,0/0/256,0/0/512,2/4/256,2/4/512,3/3/256"
SNIPER_WGRAD_XCD=1 timeout 600 python tools/wgrad_tune.py --modes="$M" > gpurun_out/wgrad_xcd1.txt 2>&1
SNIPER_WGRAD_XCD=0 timeout 600 python tools/wgrad_tune.py --modes="$M" > gpurun_out/wgrad_xcd0.txt 2>&1
SNIPER_WGRAD_XCD=1 timeout 600 python tools/wgrad_tune.py --modes="$M" --cold 600 > gpurun_out/wgrad_xcd1_cold.txt 2>&1
tail -n 3 gpurun_out/wgrad_xcd1.txt gpurun_out/wgrad_xcd0.txt gpurun_out/wgrad_xcd1_cold.txt
for X in 1 0; do
  (cd /tmp && export TMPDIR=/tmp && SNIPER_WGRAD_XCD=$X timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$ROOT/gpurun_out/pmc_wg_xcd$X" -o pmc -- \
     python "$ROOT/tools/wgrad_tune.py" --modes="-1/-1/0" --only "s3 " --iters 5 > "$ROOT/gpurun_out/pmc_wg_xcd$X.log" 2>&1; echo "pmc exit $?")
done

