"""What a batch shape a warm test-time Module has NOT met costs, phase by phase: bind (lowering + allocation + derived buffers) and the
first, eager forward; the second call (hipGraph capture); replays.  Round 5 on one card: bind + eager 21 - 31 ms (Executor.__init__
13 - 17 ms, refresh_compute_copies 6 ms), capture 6 - 39 ms, replay 3 - 5 ms -- per (shape, lane); bench.py's `cold_shape_ms` is the same
quantity measured through a whole pass of unseen shapes.

    python tools/cold_shape_probe.py
"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import sniper_amd.mx as mx
from sniper_amd import config as cfgmod
from sniper_amd.symbols.faster import resnet_mx_101_e2e as rn
cfg = cfgmod.res101_e2e_autofocus()
sym = rn.resnet_mx_101_e2e(n_proposals=400, test_nbatch=2).get_symbol_rcnn(cfg, is_train=False)
names = ['data', 'im_info', 'im_ids', 'chip_ids']
def shapes(h, w): return [('data', (2, 3, h, w)), ('im_info', (2, 3)), ('im_ids', (2,)), ('chip_ids', (2,))]
mod = mx.mod.Module(symbol=sym, context=[mx.gpu(0)], data_names=names, label_names=None)
mod.slice_inputs = False
mod.bind(shapes(576, 768), None, for_training=False)
mod.init_params(arg_params=None, aux_params=None, allow_missing=True)
rs = np.random.RandomState(0)
def batch(h, w):
    return mx.io.DataBatch(data=[mx.nd.array(rs.standard_normal((2, 3, h, w)).astype(np.float32)), mx.nd.array(np.array([[h, w, 1.0]] * 2, np.float32)),
                                 mx.nd.array(np.zeros(2, np.float32)), mx.nd.array(np.zeros(2, np.float32))], label=None, pad=0, index=None,
                           provide_data=shapes(h, w), provide_label=None)
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
b0 = batch(576, 768)
for _ in range(4): mod.forward(b0, is_train=False)
import cProfile, pstats, io
for (h, w) in ((640, 832), (448, 704), (704, 960)):
    b = batch(h, w)
    pr = cProfile.Profile() if h == 704 else None
    if pr: pr.enable()
    t = [timed(lambda: mod.forward(b, is_train=False)) for _ in range(4)]
    if pr:
        pr.disable(); s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28); print(s.getvalue()[-3800:])
    print('shape %dx%d: bind+eager %.1f ms, capture %.1f ms, replay %.1f / %.1f ms' % (h, w, t[0], t[1], t[2], t[3]), flush=True)
