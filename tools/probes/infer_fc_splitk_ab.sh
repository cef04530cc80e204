# inference leg A/B of the test-time FullyConnected / deformable-GEMM split-K routing (profiles/r06_infer_fc_splitk_ab.txt)
#   gpurun -- 'bash tools/probes/infer_fc_splitk_ab.sh'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
run() { env $1 python bench.py --steps 2 --warmup 1 --no-fit-path --no-cpu-baseline --no-c4 2>/tmp/iab.err | grep '^BENCH_DETAIL ' | tail -1 | cut -c14- | python -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d['inference']
print('%-30s value %.1f steady %.1f cold_ms %.1f unseen %.1f convms %.1f' % (sys.argv[1] or '(defaults)', i['value'], i['value_steady'], i['cold_shape_ms'], i['value_unseen_shapes'], i['roofline']['conv_ms_per_pass']))
" "$1" || tail -3 /tmp/iab.err; }
for r in 1 2 3; do for v in "" "SNIPER_FC_SPLITK=0"; do run "$v"; done; done
