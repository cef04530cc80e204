#!/bin/bash
# Per-kernel A/B from rocprofv3 kernel statistics of the SAME short bench run under different environment overrides:
#   tools/kernel_ab.sh "dpsroi|nms_lazy|topk|anchor_finish|copy" "" "SNIPER_DPSROI_NO_STAGE=1"
# prints, per variant, calls / average us of the kernels whose name matches the pattern.
PAT="$1"; shift
ROOT=$(pwd)
for v in "$@"; do
  D="$ROOT/gpurun_out/kab_$(echo "$v" | tr -c 'A-Za-z0-9' '_')"
  rm -rf "$D"
  (cd /tmp && export TMPDIR=/tmp && env $v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -o k -- \
      python "$ROOT/bench.py" --steps 6 --warmup 3 --no-cpu-baseline --no-inference --no-fit-path --no-c4 > "$D.log" 2>&1)
  F=$(find "$D" -name "*kernel_stats.csv" | head -1)
  echo "== ${v:-(defaults)}  $(grep -o '"ms_per_step": [0-9.]*' "$D.log" | head -1)"
  [ -n "$F" ] && python - "$F" "$PAT" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r['Name']):
        print('   %-70s calls %5s avg %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  find "$D" -name "*kernel_trace.csv" -delete
done
