# A/B of whole-step variants: each line is one bench.py run (environment overrides in front)
run() { env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-inference 2>/tmp/conv_ab.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; e=r['by_entry']; print('%-44s %.2f ms  insitu %.0f  fwd %.2f stats %.2f dgrad %.2f wgrad %.2f' % (' '.join(sys.argv[1:]), d['ms_per_step'], r['achieved'], e['sn_conv_fwd']['ms_per_step'], e['sn_conv_fwd_stats']['ms_per_step'], e['sn_conv_dgrad']['ms_per_step'] + e.get('sn_conv_dgrad_bn', {'ms_per_step': 0})['ms_per_step'], e['sn_conv_wgrad']['ms_per_step']))" "$@"; }
for v in "$@"; do run $v || tail -5 /tmp/conv_ab.err; done
