# a new batch shape of a warm test-time Module: first forward = the capture (default) against eager first pass + capture (SNIPER_CAPTURE_FIRST=0)
#   gpurun -- 'bash tools/probes/capture_first_ab.sh'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_inference.py -m gpu -q -x -p no:cacheprovider -n 3 -W error::UserWarning 2>&1 | tail -4
for v in "" "SNIPER_CAPTURE_FIRST=0"; do echo "## $v"; env $v python tools/cold_shape_probe.py 2>&1 | grep "^shape\|Warn\|warn"; done
run() { env $1 python bench.py --steps 2 --warmup 1 --no-fit-path --no-cpu-baseline --no-c4 2>/tmp/iab.err | grep '^BENCH_DETAIL ' | tail -1 | cut -c14- | python -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d['inference']; u=i['unseen_shapes']
print('%-26s value %.1f steady %.1f cold_ms %.1f unseen %.1f  pass1 %.3f pass2 %.3f pass3 %.3f new %d' % (sys.argv[1] or '(defaults)', i['value'], i['value_steady'], i['cold_shape_ms'], i['value_unseen_shapes'], u['seconds_pass1_bind'], u['seconds_pass2_capture'], u['seconds_pass3_replay'], u['new_executors']))
" "$1" || tail -3 /tmp/iab.err; grep -i "warn" /tmp/iab.err | head -3; }
for r in 1 2 3; do for v in "" "SNIPER_CAPTURE_FIRST=0"; do run "$v"; done; done
