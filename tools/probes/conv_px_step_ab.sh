cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
export SNIPER_HIP_LIB=sniper_amd/lib/libsniper_hip_px.so PYTHONDONTWRITEBYTECODE=1
for PX in 0 3 2; do
  SNIPER_CONV_PX=$PX timeout 100 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-inference --no-fit-path > gpurun_out/px2_bench_$PX.log 2>&1
  tail -1 gpurun_out/px2_bench_$PX.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d.get('roofline') or {}
print('experiment library, SNIPER_CONV_PX=$PX: value', d.get('value'), 'chips/s, ms_per_step', d.get('ms_per_step'), 'conv frac', r.get('frac'), 'conv_ms_per_step', r.get('conv_ms_per_step'))"
done
