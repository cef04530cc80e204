"""Instruction pattern of the MFMA loops of a device assembly file: one line per basic block that holds MFMAs, as a string of
M (v_mfma), r (ds_read), w (ds_write), D (LDS-DMA / buffer load), [L(n)] / [V(n)] (s_waitcnt), |BAR|, n (s_nop), . (anything else).
What the scheduler pins (sched_group_barrier) asked for and what came out differ per template instantiation: the data-gradient twin of
the specialised kernel had its fragment reads bunched behind the MFMAs (round 6), which the source does not show.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Isniper_amd/csrc --offload-device-only -S sniper_amd/csrc/conv_dma.hip -o /tmp/conv_dma.s
    python tools/isa_loop_pattern.py /tmp/conv_dma.s [name filter] [min MFMAs per block]
"""
import re
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', n], stdout=subprocess.PIPE, text=True).stdout.strip() or n
    except OSError:
        return n


def main():
    path = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ''
    min_m = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    name, label, cur = None, None, []
    out = []

    def flush():
        if name and cur:
            m = sum(1 for c in cur if c == 'M')
            if m >= min_m:
                out.append((name, label, ''.join(cur)))
    for ln in open(path):
        m = re.match(r'^(_Z\w+):', ln)
        if m:
            flush()
            name, label, cur = m.group(1), 'entry', []
            continue
        m = re.match(r'^(\.LBB\d+_\d+):', ln)
        if m:
            flush()
            label, cur = m.group(1), []
            continue
        if not ln.startswith('\t') or name is None:
            continue
        t = ln.strip().split(';')[0].strip()
        if not t or t.startswith('.'):
            continue
        op = t.split()[0]
        if op.startswith('v_mfma'):
            cur.append('M')
        elif op.startswith('ds_read'):
            cur.append('r')
        elif op.startswith('ds_write'):
            cur.append('w')
        elif op.startswith('buffer_load') or op.startswith('global_load'):
            cur.append('D')
        elif op == 's_waitcnt':
            cur.append('[' + t.split(None, 1)[1].replace('lgkmcnt', 'L').replace('vmcnt', 'V').replace(' ', '') + ']')
        elif op == 's_barrier':
            cur.append('|BAR|')
        elif op == 's_nop':
            cur.append('n')
        elif op == 's_endpgm':
            flush()
            name, cur = None, []
        else:
            cur.append('.')
    flush()
    last = None
    for n, lab, pat in out:
        d = demangle(n)
        if filt and filt not in d:
            continue
        if d != last:
            print(d[:150])
            last = d
        print('   %-12s %s' % (lab, pat))


if __name__ == '__main__':
    main()
