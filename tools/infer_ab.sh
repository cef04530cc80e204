#!/bin/bash
# inference leg only, one card: variants of the round-6 test-time changes
run() { env $1 python bench.py --steps 2 --warmup 1 --no-fit-path --no-cpu-baseline --no-c4 2>/tmp/iab.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d['inference']
print('%-50s value %.1f steady %.1f cold_ms %.1f unseen %.1f' % (sys.argv[1] or '(defaults)', i['value'], i['value_steady'], i['cold_shape_ms'], i['value_unseen_shapes']))" "$1" || tail -3 /tmp/iab.err; }
for r in 1 2; do for v in "" "SNIPER_INFER_LIGHT_CAPTURE=0" "SNIPER_ADOPT_DERIVED=0" "SNIPER_CONV_NO_PERSIST=1"; do run "$v"; done; done
