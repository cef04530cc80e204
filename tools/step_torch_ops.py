"""Which torch tensor operations (device copies, fills, casts: launches that are not ours) does one EAGER training step issue, and from
where?  rocprofv3 shows ~67 `__amd_rocclr_copyBuffer` and ~35 `FillFunctor` launches per step; this counts the Python call sites.

    python tools/step_torch_ops.py [batch]
"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['SNIPER_HIP_GRAPHS'] = '0'


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    from sniper_amd.train import Trainer
    tr = Trainer(batch_images=B, n_images=48, seed=0, rank_local=True)
    for _ in range(2):
        tr.step()
    torch.cuda.synchronize()
    counts = collections.Counter()
    names = ['copy_', 'clone', 'zero_', 'fill_', 'contiguous', 'to', 'half', 'float', 'permute', 'add_', 'mul_', '__setitem__',
             '__getitem__', 'sum', 'cpu', 'item']
    orig = {n: getattr(torch.Tensor, n) for n in names}
    fact = {n: getattr(torch, n) for n in ('zeros', 'ones', 'full', 'zeros_like', 'empty', 'cat', 'stack', 'as_tensor', 'from_numpy', 'tensor')}

    def site():
        for fr in reversed(traceback.extract_stack()[:-2]):
            if 'sniper_amd' in fr.filename or fr.filename.endswith('bench.py'):
                return '%s:%d %s' % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.name)
        return '?'

    def wrap(n, f, kind):
        def g(*a, **k):
            if kind == 'factory' or (len(a) and isinstance(a[0], torch.Tensor) and a[0].is_cuda) or n in ('to',):
                counts[(n, site())] += 1
            return f(*a, **k)
        return g
    for n, f in orig.items():
        setattr(torch.Tensor, n, wrap(n, f, 'method'))
    for n, f in fact.items():
        setattr(torch, n, wrap(n, f, 'factory'))
    try:
        tr.step()
    finally:
        for n, f in orig.items():
            setattr(torch.Tensor, n, f)
        for n, f in fact.items():
            setattr(torch, n, f)
    torch.cuda.synchronize()
    for (n, s), c in sorted(counts.items(), key=lambda kv: -kv[1]):
        print('%4d  %-12s %s' % (c, n, s))


if __name__ == '__main__':
    main()
