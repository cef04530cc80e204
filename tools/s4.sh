#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
for V in "1 1" "0 1" "0 0"; do
  set -- $V
  echo "=== DEFER=$1 IMPL=$2"
  SNIPER_WGRAD_DEFER=$1 SNIPER_WGRAD_IMPL=$2 timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -x -p no:cacheprovider -k test_split_backward 2>&1 | grep -E "^E  |passed|failed" | head -8
done
