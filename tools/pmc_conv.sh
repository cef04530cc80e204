#!/bin/bash
# SQ / TCC counters of the convolution kernels IN SITU (one profiled bench.py step sequence per counter set; rocprofv3 --pmc with
# --kernel-trace only).  Output: gpurun_out/pmc_conv_insitu.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
ROOT=$(pwd)
mkdir -p gpurun_out
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCC|TCP|TA|TD|GRBM)_[A-Za-z0-9_]+" | sort -u > gpurun_out/counters.txt
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$ROOT/gpurun_out/pmc_conv$i" -o c -- \
      python "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-inference --no-fit-path --no-c4 > "$ROOT/gpurun_out/pmc_conv$i.log" 2>&1; echo "set $i exit $?")
done
python - <<'PY' > gpurun_out/pmc_conv_insitu.txt
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(list)
for d in sorted(glob.glob('gpurun_out/pmc_conv[0-9]')):
    for f in glob.glob(d + '/**/*counter_collection*.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', ''))[:64]
            if 'conv' not in k and 'wgrad' not in k and 'bn_' not in k: continue
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for f in glob.glob(d + '/**/*kernel_trace*.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', ''))[:64]
            dur[k].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
for k, v in sorted(acc.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
    d_ = dur.get(k, [0])
    print('%s   launches %d  avg %.1f us' % (k, len(d_), sum(d_) / max(len(d_), 1) / 1e3))
    per = {c: x / max(n[(k, c)], 1) for c, x in v.items()}
    for c, x in sorted(per.items()):
        print('   %-28s %16.0f per launch' % (c, x))
    # derived (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, summed over the SIMDs; GRBM_GUI_ACTIVE is reported per
    # XCD and summed over the 8 of them; 256 CUs x 4 SIMDs): share of SIMD-cycles with the matrix pipe busy
    if per.get('GRBM_GUI_ACTIVE') and per.get('SQ_VALU_MFMA_BUSY_CYCLES'):
        print('   %-28s %16.3f  (MFMA busy cycles / (GUI_ACTIVE / 8 x 1024 SIMDs))' % (
            'derived mfma_busy_frac', per['SQ_VALU_MFMA_BUSY_CYCLES'] / (per['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0)))
    if per.get('SQ_LDS_IDX_ACTIVE'):
        print('   %-28s %16.4f  (bank-conflict cycles / LDS active cycles)' % (
            'derived lds_conflict_frac', per.get('SQ_LDS_BANK_CONFLICT', 0.0) / per['SQ_LDS_IDX_ACTIVE']))
    if per.get('TCC_REQ_sum'):
        print('   %-28s %16.3f  (L2 hits / requests)' % ('derived l2_hit_frac', per.get('TCC_HIT_sum', 0.0) / per['TCC_REQ_sum']))
PY
head -150 gpurun_out/pmc_conv_insitu.txt
find gpurun_out/pmc_conv[0-9] -name "*.csv" -size +6M -delete
