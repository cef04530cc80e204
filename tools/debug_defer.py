"""Deferred (batched) vs per-layer weight gradients on the same network and batch: gradients of every parameter after one
forward + backward, single-pass and split backward.  python tools/debug_defer.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(defer, split, steps=2):
    os.environ['SNIPER_WGRAD_DEFER'] = defer
    os.environ['SNIPER_OVERLAP_ALLREDUCE'] = split
    os.environ['SNIPER_HIP_GRAPHS'] = '0'
    from sniper_amd.train import Trainer
    tr = Trainer(batch_images=2, n_images=4, seed=3)
    ex = tr.mod.exe
    rec = []
    for _ in range(steps):
        tr.mod.forward_backward(tr.batch)
        torch.cuda.synchronize()
        rec.append(([t.double().cpu().numpy().copy() for t in ex.outputs],
                    {n: p.grad.double().cpu().numpy().copy() for n, p in ex.params.items() if p.trainable}))
        tr.mod.update()
    return rec


def main():
    base = run('0', '0')
    for defer, split in (('1', '0'), ('1', 'force'), ('0', 'force')):
        got = run(defer, split)
        for s in range(len(base)):
            o = max(float(np.abs(u - v).max() / (np.abs(u).max() + 1e-30)) for u, v in zip(base[s][0], got[s][0]))
            bad = []
            for n in base[s][1]:
                u, v = base[s][1][n], got[s][1][n]
                if not np.isfinite(v).all():
                    bad.append((float('inf'), n))
                    continue
                bad.append((float(np.abs(u - v).max() / (np.abs(u).max() + 1e-30)), n))
            bad.sort(reverse=True)
            print('defer %s split %s step %d: outputs rel %.2e; worst gradients: %s' % (
                defer, split, s, o, ', '.join('%s %.1e' % (n, e) for e, n in bad[:6])), flush=True)


if __name__ == '__main__':
    main()
