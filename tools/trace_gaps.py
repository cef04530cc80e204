"""GPU busy / idle accounting from a rocprofv3 --kernel-trace CSV: union of the kernel intervals vs wall time over the
window [--lo, --hi) of the kernel sequence (fractions; default the middle of the run: graph-replayed steps), the largest idle gaps and the kernels that follow them.
Writes a small JSON summary (the raw trace is too large to keep).

    python tools/trace_gaps.py <kernel_trace.csv> <out.json> [--lo 0.45 --hi 0.75]
"""
import csv
import json
import sys


def main():
    path, out = sys.argv[1], sys.argv[2]
    lo = float(sys.argv[sys.argv.index('--lo') + 1]) if '--lo' in sys.argv else 0.45
    hi = float(sys.argv[sys.argv.index('--hi') + 1]) if '--hi' in sys.argv else 0.75
    rows = []
    with open(path, newline='') as fh:
        rd = csv.DictReader(fh)
        for r in rd:
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '')))
    rows.sort()
    n = len(rows)
    rows = rows[int(n * lo):int(n * hi)]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    gaps = []
    for s, e, name, q in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, name))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    total_kernel = sum(e - s for s, e, _, _ in rows)
    gaps.sort(reverse=True)
    by_kernel = {}
    for g, name in gaps:
        k = name.split('(')[0][-60:]
        d = by_kernel.setdefault(k, [0, 0])
        d[0] += 1
        d[1] += g
    hist = {'<2us': 0, '2-5us': 0, '5-20us': 0, '20-100us': 0, '>100us': 0}
    for g, _ in gaps:
        us = g / 1e3
        hist['<2us' if us < 2 else '2-5us' if us < 5 else '5-20us' if us < 20 else '20-100us' if us < 100 else '>100us'] += g
    res = {'kernels': len(rows), 'wall_ms': (t1 - t0) / 1e6, 'busy_ms': busy / 1e6, 'busy_frac': busy / (t1 - t0),
           'sum_kernel_ms': total_kernel / 1e6, 'overlap_factor': total_kernel / busy, 'n_gaps': len(gaps),
           'idle_ms_by_gap_size': {k: v / 1e6 for k, v in hist.items()},
           'top_gaps_us': [(round(g / 1e3, 1), name[:80]) for g, name in gaps[:15]],
           'idle_ms_before_kernel': sorted(((k, v[0], round(v[1] / 1e6, 3)) for k, v in by_kernel.items()), key=lambda x: -x[2])[:15]}
    with open(out, 'w') as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k: res[k] for k in ('kernels', 'wall_ms', 'busy_ms', 'busy_frac', 'sum_kernel_ms', 'overlap_factor',
                                          'idle_ms_by_gap_size')}))


if __name__ == '__main__':
    main()
