#!/bin/bash
# 1 / 2 / 4 / 8-GPU sweep of the BASELINE metric on ONE node (main_train.py:43,143-144 kvstore='device' is the
# reference's counterpart): one process per GPU over RCCL, the launch line the driver uses.  Every run prints bench.py's compact
# JSON line; N > 1 lines carry `dist` = {allreduce_ms, overlap_frac, rccl_ranks_seen}.  Needs an N-GPU lease (none existed in
# rounds 1-6: no N > 1 RCCL number has been measured); results land in gpurun_out/scale_<N>.json.
#   tools/scale_sweep.sh [steps (20)] [warmup (5)] [gpu counts ("1 2 4 8")]
set -u
cd "$(dirname "$0")/.."
STEPS=${1:-20}; WARM=${2:-5}; COUNTS=${3:-"1 2 4 8"}
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
have=$(python -c 'import torch; print(torch.cuda.device_count())')
for n in $COUNTS; do
  if [ "$n" -gt "$have" ]; then echo "scale_sweep: $n GPUs asked, $have visible: skipped"; continue; fi
  port=$((29500 + n))
  if [ "$n" -eq 1 ]; then
    python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARM" --no-inference > gpurun_out/scale_$n.log 2>&1
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "$port" \
      bench.py --gpus "$n" --steps "$STEPS" --warmup "$WARM" --no-inference > gpurun_out/scale_$n.log 2>&1
  fi
  tail -n 1 gpurun_out/scale_$n.log > gpurun_out/scale_$n.json
  python - "$n" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.load(open('gpurun_out/scale_%s.json' % n))
    print('N=%s: %.1f chips/s, %.2f ms/step, dist=%s' % (n, d['value'], d['ms_per_step'], d.get('dist')))
except Exception as e:  # noqa: BLE001
    print('N=%s: no bench line (%r); see gpurun_out/scale_%s.log' % (n, e, n))
PY
done
