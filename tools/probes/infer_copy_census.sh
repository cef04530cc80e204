# which launches surround the __amd_rocclr_copyBuffer / fill kernels of an inference pass (rocprofv3 kernel trace of tools/infer_profile.py)
cd $GRAFT_REPO_ROOT; export PYTHONDONTWRITEBYTECODE=1; ROOT=$(pwd)
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$ROOT/gpurun_out/inf_trace" -o t -- python "$ROOT/tools/infer_profile.py" 1 3 - 1 16 - distinct > "$ROOT/gpurun_out/inf_trace.log" 2>&1)
python - <<'PY'
import csv, glob, re
from collections import Counter
f = glob.glob('gpurun_out/inf_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = re.sub(r'\(.*', '', n).replace('void ', '').replace('(anonymous namespace)::', '')
    return n[:44]
rows = rows[len(rows) * 4 // 7:]          # the later passes (replays)
c = Counter(); tot = Counter()
for i, r in enumerate(rows):
    k = r['Kernel_Name']
    if 'copyBuffer' in k or 'FillFunctor' in k or 'fillBuffer' in k:
        key = (short(k)[:28], r['Grid_Size_X'], short(rows[i - 1]['Kernel_Name']), short(rows[i + 1]['Kernel_Name']) if i + 1 < len(rows) else '-')
        c[key] += 1
        tot[short(k)[:28]] += 1
print('launches in window', len(rows), dict(tot))
for k, v in c.most_common(40):
    print(v, k)
PY
