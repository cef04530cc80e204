import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('BENCH_DETAIL '):
        d=json.loads(ln[13:])
        print(sys.argv[1], d['ms_per_step'])
        for x in d['roofline']['by_shape']:
            if any(k in x['shape'] for k in sys.argv[2].split('|')):
                print('  %-18s %-34s n=%2d %7.1f us' % (x['entry'],x['shape'],x['launches_per_step'],x['us_per_launch']))
