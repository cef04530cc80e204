"""Per-kernel HBM traffic and achieved bandwidth from two rocprofv3 passes (--kernel-trace --pmc FETCH_SIZE, --pmc WRITE_SIZE;
they do not fit one pass on gfx950) of one command:

    python tools/pmc_report.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> out.json [name filter, comma separated] [kernel_stats.csv]

For every kernel: launches, average duration (kernel trace of the FETCH pass), HBM bytes fetched / written per launch and
(fetch + write) / duration in GB/s.  Counter collection slows the kernels (a streaming kernel of 19 us takes 29 us in the FETCH
pass): with the `--kernel-trace --stats` summary of the same command given as last argument the rate is also computed over the
UN-countered average duration (avg_us_stats, hbm_gb_per_s_stats) -- the bytes of a launch do not depend on the pass.  Corrections as MI355X_MICROARCH.md section HBM prescribes: both counters are in KiB; on
gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read, so it is doubled; WRITE_SIZE is used as reported
(uncalibrated).  Traffic served by the 256 MB Infinity Cache appears to be counted, so for small working sets the figure
is an upper bound of what reached HBM."""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '')
    name = re.sub(r'\(.*', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'_Z\d+([A-Za-z_0-9]+?)(I[A-Z].*|P.*|v)?$', name)
    if name.startswith('_Z'):
        m = re.match(r'_Z(\d+)', name)
        n = int(m.group(1))
        return name[2 + len(m.group(1)):][:n]
    return name.strip()


def counters(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection*.csv'), recursive=True):
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') != counter:
                    continue
                e = out.setdefault(short(row.get('Kernel_Name', '')), [0.0, 0])
                e[0] += float(row['Counter_Value'])
                e[1] += 1
    return out


def durations(d):
    out = {}
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace*.csv'), recursive=True):
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                e = out.setdefault(short(row.get('Kernel_Name', '')), [0.0, 0])
                e[0] += float(row['End_Timestamp']) - float(row['Start_Timestamp'])
                e[1] += 1
    return out


def _lib_hash():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import library_sources_hash
    return library_sources_hash()


def main():
    fdir, wdir, out = sys.argv[1], sys.argv[2], sys.argv[3]
    flt = [s for s in (sys.argv[4].split(',') if len(sys.argv) > 4 else []) if s]
    fetch, write, dur = counters(fdir, 'FETCH_SIZE'), counters(wdir, 'WRITE_SIZE'), durations(fdir)
    stats = {}
    if len(sys.argv) > 5 and os.path.exists(sys.argv[5]):
        with open(sys.argv[5], newline='') as fh:
            for row in csv.DictReader(fh):
                stats[short(row['Name'])] = float(row['AverageNs']) / 1e3
    res = {}
    for k in sorted(set(fetch) | set(write)):
        if flt and not any(s in k for s in flt):
            continue
        fv, fn = fetch.get(k, [0.0, 0])
        wv, wn = write.get(k, [0.0, 0])
        dv, dn = dur.get(k, [0.0, 0])
        fb, wb = 2.0 * fv * 1024.0 / max(fn, 1), wv * 1024.0 / max(wn, 1)
        us = dv / max(dn, 1) / 1e3
        res[k] = {'launches': max(fn, wn), 'avg_us': round(us, 2), 'fetch_bytes_per_launch': round(fb), 'write_bytes_per_launch': round(wb),
                  'hbm_gb_per_s': round((fb + wb) / (us * 1e-6) / 1e9, 1) if us > 0 else None}
        if k in stats and stats[k] > 0:
            res[k]['avg_us_stats'] = round(stats[k], 2)
            res[k]['hbm_gb_per_s_stats'] = round((fb + wb) / (stats[k] * 1e-6) / 1e9, 1)
    doc = {'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); KiB -> bytes, FETCH_SIZE x2 '
                     '(gfx950 correction, MI355X_MICROARCH.md); durations from the kernel trace of the FETCH pass (avg_us, hbm_gb_per_s) and, where given, from the --stats summary of the '
                     'un-countered run of the same command (avg_us_stats, hbm_gb_per_s_stats)',
           'peak_gb_per_s': 8000, 'library_sources_hash': _lib_hash(), 'kernels': res}
    with open(out, 'w') as fh:
        json.dump(doc, fh, indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -(kv[1]['avg_us'] * kv[1]['launches'])):
        print('%-44s n %5d  %9.1f us  fetch %11d  write %11d  %8s GB/s  (un-countered %s us, %s GB/s)' % (
            k[:44], v['launches'], v['avg_us'], v['fetch_bytes_per_launch'], v['write_bytes_per_launch'], v['hbm_gb_per_s'],
            v.get('avg_us_stats'), v.get('hbm_gb_per_s_stats')))


if __name__ == '__main__':
    main()
