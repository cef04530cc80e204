"""Recompute the conv family's in-situ TFLOP/s from a `rocprofv3 --kernel-trace --stats` summary of bench.py and compare it with the
`roofline.achieved` the same command printed:

    python tools/roofline_check.py <kernel_stats.csv> <bench json line file> <steps profiled> [out.json]

steps profiled = warmup + steps + 1 (idle-device step) + 3 (the eager profiling pass: one settling step, two bracketed) of that
bench.py run: every one of them launches the same conv kernels."""
import csv
import json
import sys

FAMILY = ('conv_dma_kernel', 'conv_igemm_kernel', 'conv_wgrad_kernel', 'wgrad_reduce_kernel', 'wgrad_ps_kernel', 'wgrad_reduce2_kernel', 'wgrad_reduce_batch_kernel')


def main():
    stats, bench, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    tot_ns, rows = 0.0, {}
    with open(stats, newline='') as fh:
        for r in csv.DictReader(fh):
            fam = next((k for k in FAMILY if k in r['Name']), None)
            if fam:
                tot_ns += float(r['TotalDurationNs'])
                e = rows.setdefault(fam, [0, 0.0])
                e[0] += int(r['Calls'])
                e[1] += float(r['TotalDurationNs'])
    lines = list(open(bench))
    full = [ln[len('BENCH_DETAIL '):] for ln in lines if ln.startswith('BENCH_DETAIL {')]        # the full record (the final line is compact)
    line = full[-1] if full else [ln for ln in lines if ln.startswith('{"metric"')][-1]
    d = json.loads(line)
    gflop = d['roofline']['gflop_per_step']
    ms = tot_ns / steps / 1e6
    tf = gflop / ms
    res = {'conv_family_ms_per_step_rocprof': round(ms, 3), 'tflops_rocprof': round(tf, 1), 'frac_rocprof': round(tf / 2500.0, 4),
           'bench_achieved': d['roofline']['achieved'], 'bench_frac': d['roofline']['frac'],
           'ratio_bench_over_rocprof': round(d['roofline']['achieved'] / tf, 3), 'steps_profiled': steps,
           'device': d.get('device'), 'library_sources_hash': d.get('library_sources_hash'),
           'bench_ms_per_step_under_profiler': d.get('ms_per_step'),
           'by_kernel_ms_per_step': {k: round(v[1] / steps / 1e6, 3) for k, v in rows.items()},
           'launches_per_step': {k: round(v[0] / steps, 1) for k, v in rows.items()}}
    print(json.dumps(res))
    if len(sys.argv) > 4:
        json.dump(res, open(sys.argv[4], 'w'), indent=1)


if __name__ == '__main__':
    main()
