"""Run the small training graph of tests/test_gpu_engine.py many times (fresh executor every `--rebuild` iterations) and compare
every intermediate tensor and parameter gradient bit for bit with the first run: all kernels are deterministic, so any
difference is a race or a read of uninitialised memory.  Prints the first tensor (in step order) that differs.

    python tools/race_hunt.py [--iters 200] [--rebuild 10] [--save ref.pt | --check ref.pt]

--save / --check: the same comparison ACROSS processes (a race that only shows in a fresh process: first launches, cold caches).
Activations are snapshotted right after the forward pass (the backward pass recycles some of their buffers)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--rebuild', type=int, default=10)
    ap.add_argument('--save')
    ap.add_argument('--check')
    a = ap.parse_args()
    import sniper_amd.mx as mx
    from sniper_amd.engine.executor import Executor
    from test_gpu_engine import _mini_graph
    A, B, S = 3, 2, 64
    sym = _mini_graph(mx, A)
    F = S // 8
    shapes = dict(data=(B, 3, S, S), label=(B, A * F * F), bbox_target=(B, 4 * A, F, F), bbox_weight=(B, 4 * A, F, F))
    fixed = [n for n in sym.list_arguments() if any(p in n for p in ('conv0', 'bn0', 'bn_data'))]
    rs = np.random.RandomState(0)
    args, _, auxs = sym.infer_shape(**shapes)
    P, AUX = {}, {}
    for name, shp in zip(sym.list_arguments(), args):
        if name in shapes:
            continue
        if name.endswith('_gamma'):
            P[name] = rs.uniform(0.5, 1.5, shp).astype(np.float32)
        elif name.endswith('_beta') or name.endswith('_bias'):
            P[name] = (rs.standard_normal(shp) * 0.1).astype(np.float32)
        else:
            P[name] = (rs.standard_normal(shp) * np.sqrt(2.0 / np.prod(shp[1:]))).astype(np.float32)
    for name, shp in zip(sym.list_auxiliary_states(), auxs):
        AUX[name] = rs.uniform(0.5, 1.5, shp).astype(np.float32) if name.endswith('_var') else \
            (rs.standard_normal(shp) * 0.1).astype(np.float32)
    inp = dict(data=(rs.standard_normal((B, 3, S, S)) * 2).astype(np.float32),
               label=rs.choice([-1, 0, 1], size=(B, A * F * F), p=[0.5, 0.3, 0.2]).astype(np.float32),
               bbox_target=rs.standard_normal((B, 4 * A, F, F)).astype(np.float32),
               bbox_weight=(rs.uniform(size=(B, 4 * A, F, F)) < 0.2).astype(np.float32))

    def step_outputs(ex):
        for st in ex.steps:
            t = getattr(getattr(st, 'y', None), 't', None)
            if isinstance(t, torch.Tensor) and t.is_floating_point():
                yield 'act %s (%s)' % (st.node.name, type(st).__name__), t

    def poison(ex):
        # a step output nobody writes (a convolution whose result goes straight into the residual sum, ...) stays NaN and is
        # left out of the comparison
        for _, t in step_outputs(ex):
            t.fill_(float('nan'))

    def snap_forward(ex):
        out = [(n, t.detach().clone()) for n, t in step_outputs(ex) if not bool(torch.isnan(t).all())]
        for k, o in enumerate(ex.outputs):
            out.append(('out %d' % k, o.detach().clone()))
        return out

    def snap_backward(ex):
        return [('grad ' + name, p.grad.detach().clone()) for name, p in ex.params.items() if p.trainable]
    first, bad, ex, differ, total = None, 0, None, {}, 0
    if a.check:
        first = [(n, t.cuda()) for n, t in torch.load(a.check)]
    for it in range(a.iters):
        if ex is None or it % a.rebuild == 0:
            ex = Executor(sym, shapes, True, fixed)
            ex.use_graphs = False
        ex.set_params(P, AUX)               # (the moving statistics move with every training forward: reset them too)
        poison(ex)
        ex.forward(inp, is_train=True)
        torch.cuda.synchronize()
        snap = snap_forward(ex)
        ex.backward()
        torch.cuda.synchronize()
        snap += snap_backward(ex)
        if first is None:
            first = snap
            if a.save:
                torch.save([(n, t.cpu()) for n, t in snap], a.save)
            continue
        cur = dict(snap)
        for n0, t0 in first:
            t1 = cur.get(n0)
            if t1 is None or t0.shape != t1.shape or not torch.equal(t0, t1):
                d = (t0.float() - t1.float()).abs() if t1 is not None and t0.shape == t1.shape else torch.full((1,), float('inf'))
                rec = differ.setdefault(n0, [0, 0, 0.0, it])
                rec[0] += 1
                rec[1] = max(rec[1], int((d > 0).sum()))
                rec[2] = max(rec[2], float(d.max()))
        total += 1
    for n, (cnt, nel, mx, it0) in differ.items():
        bad += 1
        print('%-60s differs in %d of %d runs (first at %d), up to %d elements, max |diff| %.4g' % (n, cnt, total, it0, nel, mx))
    print('%d runs compared (%d tensors each), %d tensors differed' % (total, len(first), bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
