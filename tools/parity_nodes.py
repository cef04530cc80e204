"""Per-node forward mismatch of the teacher-forced R101 parity run (tests/test_gpu_engine.py::_forced_parity), largest first:
which operator disagrees with oracle/graph_cpu.py on identical inputs, and by how much.

    python tools/parity_nodes.py [batch (2)] [seed (12)]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    import test_gpu_engine as T
    from oracle import graph_cpu
    from sniper_amd import config as cfgmod
    from sniper_amd.engine.executor import Executor
    from sniper_amd.symbols.faster import resnet_mx_101_e2e as rn
    from sniper_amd.train import fixed_param_names
    A, F = 21, 32
    cfg = cfgmod.res101_e2e(batch_images=B)
    sym = rn.resnet_mx_101_e2e(momentum=0.995).get_symbol_rcnn(cfg)
    shapes = dict(data=(B, 3, 512, 512), valid_ranges=(B, 2), im_info=(B, 3), label=(B, A * F * F),
                  bbox_target=(B, 4 * A, F, F), bbox_weight=(B, 4 * A, F, F), gt_boxes=(B, 100, 5))
    os.environ['SNIPER_HIP_GRAPHS'] = '0'
    ex = Executor(sym, shapes, True, fixed_param_names(cfg, sym))
    rs = np.random.RandomState(seed)
    P, AUX = T._init_params(sym, shapes, rs, bn_gamma=(0.5, 1.0), bn_beta=(-0.2, 0.4))
    P['bn_data_gamma'][:] = 1.0
    AUX['bn_data_moving_mean'][:] = 0.0
    AUX['bn_data_moving_var'][:] = 1.0 - 2e-5
    P['bn_data_beta'][:] = 0.0
    inp = T._train_inputs(rs, B, A, F)
    graph_cpu.run.keep = set(sys.argv[3].split(',')) if len(sys.argv) > 3 else set()
    try:
        T._forced_parity(sym, ex, P, AUX, inp, tol_fwd=1.0, tol_grad=1.0)
    except AssertionError as e:
        print('assertion:', str(e)[:300])
    le = graph_cpu.run.local_err
    for k, v in sorted(le.items(), key=lambda kv: -kv[1])[:25]:
        print('%-40s %.5f' % (k, v))
    for name, (want, got) in graph_cpu.run.kept.items():
        err = np.abs(want - got)
        thr = 20 * np.median(err) + 1e-6
        bad = np.argwhere(err > thr)
        print('%s: shape %s, median |err| %.3g, max %.3g, %d elements beyond 20 x median' % (name, want.shape, np.median(err), err.max(), len(bad)))
        if len(bad):
            for ax in range(bad.shape[1]):
                vals, cnt = np.unique(bad[:, ax], return_counts=True)
                print('   axis %d: %d distinct indices, e.g. %s' % (ax, len(vals), list(zip(vals[:12].tolist(), cnt[:12].tolist()))))
            for b in bad[:8]:
                print('   at', tuple(int(v) for v in b), 'oracle %.5f device %.5f' % (want[tuple(b)], got[tuple(b)]))
    # magnitudes around the worst node
    worst = max(le, key=le.get)
    for st in ex.steps:
        if st.node.name.startswith(worst.rsplit('_', 1)[0]):
            y = getattr(st, 'y', None)
            if y is not None and y.t is not None:
                t = y.t.float()
                print('%-40s %-28s absmax %.4g rms %.4g' % (st.node.name, type(st).__name__, float(t.abs().max()), float(t.pow(2).mean().sqrt())))


if __name__ == '__main__':
    main()
