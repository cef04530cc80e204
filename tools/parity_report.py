"""gpurun_out/parity_quantiles.jsonl (written by tests/gpu_util.assert_close on the GPU box) -> a table for profiles/:
per comparison the elementwise relative-error quantiles and the max-norm error behind the tests' `<= 1e-2` assertions.

    python tools/parity_report.py gpurun_out/parity_quantiles.jsonl > profiles/r03_parity_quantiles.txt
"""
import json
import sys


def main():
    rows = [json.loads(ln) for ln in open(sys.argv[1]) if ln.strip()]
    print('%-86s %10s %9s | %s | %s | %s' % ('comparison (test :: tensor)', 'elements', 'rel L2',
                                           '|err|/|want|, elements >= 5%% of max: p50 p99 max', 'elements >= 1e-3 of max: p50 p99 p99.9',
                                           '|err|/max|want|: p99 p99.9 max'))
    for r in rows:
        q, m, g = r['rel_elementwise'], r['err_over_max'], r.get('rel_significant', r['rel_elementwise'])
        name = (r.get('test', '').split('::')[-1] + ' :: ' + r['what'])[:86]
        print('%-86s %10d %9.2e | %.1e %.1e %.1e | %.1e %.1e %.1e | %.1e %.1e %.1e' % (
            name, r['n'], r['rel_l2'], g['p50'], g['p99'], g['p100'], q['p50'], q['p99'], q['p99.9'], m['p99'], m['p99.9'], m['p100']))
    worst = max(rows, key=lambda r: r['err_over_max']['p100'])
    print('\n%d comparisons; largest max-norm error %.2e (%s)' % (len(rows), worst['err_over_max']['p100'], worst['what']))


if __name__ == '__main__':
    main()
