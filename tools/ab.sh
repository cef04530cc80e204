#!/bin/bash
# Whole-step A/B on ONE card in ONE session: every argument is a set of environment overrides in front of the same bench.py run
# (e.g. "SNIPER_HIP_LIB=sniper_amd/lib/libsniper_hip_nt.so" or "SNIPER_DGRAD_BY_CLASS=0"; "" = the defaults), repeated ROUNDS times
# interleaved so that clock drift hits every variant alike.
#   tools/ab.sh "" "SNIPER_DGRAD_BY_CLASS=0"
ROUNDS=${ROUNDS:-2}
run() { env $1 python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline --no-inference --no-fit-path --no-c4 2>/tmp/ab.err | grep '^BENCH_DETAIL ' | tail -1 | cut -c14- | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; e=r['by_entry']
g=lambda k: e.get(k, {'ms_per_step': 0.0})['ms_per_step']
print('%-60s %.3f ms/step  conv in situ %.2f ms (%.0f TF/s)  fwd %.2f fwd_stats %.2f dgrad %.2f dgrad_bn %.2f wgrad %.2f' % (sys.argv[1] or '(defaults)', d['ms_per_step'], r['conv_ms_per_step'], r['achieved'], g('sn_conv_fwd'), g('sn_conv_fwd_stats'), g('sn_conv_dgrad'), g('sn_conv_dgrad_bn'), g('sn_conv_wgrad_batch')))" "$1" || tail -5 /tmp/ab.err; }
for r in $(seq $ROUNDS); do for v in "$@"; do run "$v"; done; done
