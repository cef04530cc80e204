#!/bin/bash
# SQ counters of the conv kernels on the RPN-sized problem (rocprofv3 --pmc, kernel-trace only).
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
ROOT=$(pwd)
mkdir -p gpurun_out
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCC|TCP|TA)_[A-Z0-9_]+" | sort -u > gpurun_out/counters.txt
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$ROOT/gpurun_out/pmc_conv$i" -o c -- python "$ROOT/tools/microbench.py" --only rpn --quick > "$ROOT/gpurun_out/pmc_conv$i.log" 2>&1
  echo "set $i exit $?"
done
cd "$ROOT" && python - <<'PY'
import csv, glob, collections
for d in ('gpurun_out/pmc_conv1', 'gpurun_out/pmc_conv2'):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + '/**/*counter_collection*.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:60]
            if 'conv' not in k: continue
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k, v in acc.items():
        print(k)
        for c, x in sorted(v.items()):
            print('   %-28s %14.0f per launch' % (c, x / max(n[(k, c)], 1)))
PY
