"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass on gfx950) of
`python bench.py ...` into profiles/pmc_traffic.json: HBM bytes per launch of the conv kernel family.

    python tools/pmc_traffic.py <dir with FETCH_SIZE csv> <dir with WRITE_SIZE csv> [out.json]

Corrections (MI355X_MICROARCH.md, section HBM): both counters are in KiB; on gfx950 FETCH_SIZE reports half the
bytes of a wide coalesced read, so it is doubled.  WRITE_SIZE is used as reported (uncalibrated)."""
import csv
import glob
import json
import os
import sys

FAMILY = ('conv_dma_kernel', 'conv_igemm_kernel', 'conv_wgrad_kernel', 'wgrad_reduce_kernel', 'wgrad_ps_kernel', 'wgrad_reduce2_kernel', 'wgrad_reduce_batch_kernel')


def collect(d, counter):
    """-> {kernel short name: [sum_value, launches]} for `counter`."""
    out = {}
    files = glob.glob(os.path.join(d, '**', '*counter_collection*.csv'), recursive=True)
    if not files:
        raise SystemExit('no *counter_collection*.csv under %s' % d)
    for f in files:
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') != counter:
                    continue
                name = row.get('Kernel_Name', '')
                fam = next((k for k in FAMILY if k in name), None)
                if fam is None:
                    continue
                e = out.setdefault(fam, [0.0, 0])
                e[0] += float(row['Counter_Value'])
                e[1] += 1
    return out


def main():
    fdir, wdir = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 7       # steps the profiled command ran: warmup 1 + 2 timed + 1 idle-device + 3 eager
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             'profiles', 'pmc_traffic.json')
    fetch, write = collect(fdir, 'FETCH_SIZE'), collect(wdir, 'WRITE_SIZE')
    per = {}
    tot_b, tot_n = 0.0, 0
    for k in sorted(set(fetch) | set(write)):
        fv, fn = fetch.get(k, [0.0, 0])
        wv, wn = write.get(k, [0.0, 0])
        n = max(fn, wn, 1)
        fb, wb = 2.0 * fv * 1024.0 / max(fn, 1), wv * 1024.0 / max(wn, 1)
        per[k] = {'launches': n, 'fetch_bytes_per_launch': round(fb), 'write_bytes_per_launch': round(wb)}
        tot_b += (fb + wb) * n
        tot_n += n
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import conv_sources_hash
    res = {'hbm_bytes_per_launch': round(tot_b / max(tot_n, 1)), 'launches': tot_n, 'steps': steps,
           'hbm_bytes_per_step': round(tot_b / steps), 'kernel_launches_per_step': round(tot_n / steps, 1), 'conv_sources_hash': conv_sources_hash(), 'by_kernel': per,
           'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 2 '
                     '--warmup 1 --no-cpu-baseline`; KiB -> bytes, FETCH_SIZE x2 (gfx950 correction)'}
    with open(out, 'w') as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
