"""Test-time executor cache of the Module (sniper_amd/mx/module.py::_exe_for / _evict_stale): least recently used first out,
bounded by a count (and by HBM share on a GPU) -- the first-in-first-out cache of eight missed on every batch of a 64-image pass."""
import torch

from sniper_amd.mx.module import Module


def _module(n, monkeypatch, cap):
    monkeypatch.setenv('SNIPER_EXE_CACHE', str(cap))
    m = Module.__new__(Module)
    m.for_training = False
    m._device = torch.device('cpu')
    m._exes = {('s', i): object() for i in range(n)}
    m.exe = m._exes.get(('s', 0))
    return m


def test_eviction_drops_the_oldest_and_keeps_the_executor_in_use(monkeypatch):
    m = _module(6, monkeypatch, cap=4)
    assert m._evict_stale()
    # ('s', 0) is in use: rotated to the back, the next oldest go
    assert list(m._exes) == [('s', 3), ('s', 4), ('s', 5), ('s', 0)]
    assert m._exes[('s', 0)] is m.exe


def test_no_eviction_below_the_bound(monkeypatch):
    m = _module(6, monkeypatch, cap=256)
    assert not m._evict_stale()
    assert len(m._exes) == 6


def test_a_hit_moves_the_shape_to_the_back(monkeypatch):
    m = _module(3, monkeypatch, cap=256)
    shapes = {'data': (1, 3, 64, 64)}
    key = tuple(sorted((k, tuple(v)) for k, v in shapes.items()))
    m._exes = {key: 'first', ('s', 1): 'second'}
    assert m._exe_for(shapes) == 'first'
    assert list(m._exes) == [('s', 1), key]


def test_cyclic_walk_over_one_more_shape_than_the_old_bound_never_misses(monkeypatch):
    """nine shapes visited in the same order pass after pass: every visit a hit (the old FIFO of eight: every visit a miss)"""
    m = _module(0, monkeypatch, cap=256)
    built = []
    keys = [{'data': (2, 3, 64 * (i + 1), 64)} for i in range(9)]

    def key_of(s):
        return tuple(sorted((k, tuple(v)) for k, v in s.items()))
    for s in keys:                       # stands for the builds of pass 1
        m._exes[key_of(s)] = object()
        built.append(key_of(s))
    for _ in range(3):
        for s in keys:
            assert m._exe_for(s) is not None
    assert sorted(m._exes) == sorted(built)


def test_activation_pool_lays_executors_over_the_same_bytes():
    """every executor walks the Module's buffers from the start: same requests -> same addresses; a larger shape grows the pool
    by appending (earlier addresses stay valid: captured forwards hold them)"""
    from sniper_amd.engine.executor import ActivationPool
    pool = ActivationPool(torch.device('cpu'))
    pool.CHUNK = 1 << 16
    a, b = pool.cursor(), pool.cursor()
    ta = [pool.take(a, (3, 100), torch.float16), pool.take(a, (7,), torch.float32), pool.take(a, (1 << 15,), torch.uint8)]
    tb = [pool.take(b, (3, 100), torch.float16), pool.take(b, (7,), torch.float32), pool.take(b, (1 << 15,), torch.uint8)]
    assert [t.data_ptr() for t in ta] == [t.data_ptr() for t in tb]
    assert all(t.data_ptr() % 256 == 0 for t in ta)
    assert ta[0].shape == (3, 100) and ta[0].dtype == torch.float16 and ta[1].dtype == torch.float32
    assert ta[1].data_ptr() - ta[0].data_ptr() == 768          # 600 bytes rounded up to the alignment
    n0 = pool.nbytes()
    big = pool.take(a, (1 << 17,), torch.uint8)                # larger than a chunk: its own buffer
    assert big.numel() == 1 << 17 and len(pool.buffers) == 2 and pool.nbytes() == n0 + (1 << 17) + pool.ALIGN
    again = pool.take(b, (1 << 17,), torch.uint8)
    assert again.data_ptr() == big.data_ptr()
    ta[0].fill_(1.0)
    assert float(tb[0].sum()) == 300.0                         # the same memory


def test_test_time_executors_of_one_module_share_their_activations():
    """two bound shapes of the R101 test graph: the second executor's activations lie inside the first's pool (no growth when it
    is smaller), parameters stay per executor"""
    from sniper_amd import config as cfgmod
    from sniper_amd.engine.executor import ActivationPool, Executor
    from sniper_amd.symbols.faster import resnet_mx_101_e2e as ours
    cfg = cfgmod.res101_e2e_autofocus()
    sym = ours.resnet_mx_101_e2e(n_proposals=400, test_nbatch=1).get_symbol_rcnn(cfg, is_train=False)
    args = set(sym.list_arguments())
    pool = ActivationPool(torch.device('cpu'))

    def bind(h, w):
        shapes = {k: v for k, v in dict(data=(1, 3, h, w), im_info=(1, 3), im_ids=(1,), chip_ids=(1,)).items() if k in args}
        return Executor(sym, shapes, False, (), device=torch.device('cpu'), act_pool=pool)
    big = bind(192, 256)
    held = pool.nbytes()
    small = bind(128, 192)
    assert pool.nbytes() == held
    lo = min(b.data_ptr() for b in pool.buffers)
    inside = lambda t: any(b.data_ptr() <= t.data_ptr() and t.data_ptr() + t.numel() * t.element_size() <= b.data_ptr() + b.numel()
                           for b in pool.buffers)
    acts = [v.t for v in small.vals.values() if v.t is not None and getattr(v, 'producer', None) is not None]
    assert len(acts) > 100 and all(inside(t) for t in acts)
    first = lo + (-lo) % pool.ALIGN
    assert min(t.data_ptr() for t in acts) == first == min(v.t.data_ptr() for v in big.vals.values() if v.t is not None)
    assert not inside(small.params['stage3_unit1_conv2_weight'].w16)
    # a training executor never takes the pool
    assert Executor.__init__.__code__.co_varnames.count('act_pool') == 1


def test_device_image_cache_bounds_bytes_and_survives_id_reuse(monkeypatch):
    """data/im_worker.py::DeviceImageCache: one upload per image; oldest entries leave when the byte bound is reached; an in-memory
    array is matched by identity, not by a recycled id()"""
    import numpy as np
    from sniper_amd.data import im_worker as iw
    ups = []
    monkeypatch.setattr(iw, '_to_device', lambda a: (ups.append(a.shape), torch.from_numpy(a.copy()))[1])
    c = iw.DeviceImageCache(max_bytes=3 * 48)
    ims = [np.full((4, 4, 3), i, np.uint8) for i in range(4)]
    a0 = c.get(ims[0])
    assert c.get(ims[0]) is a0 and (c.hits, c.misses) == (1, 1)
    c.get(ims[1]); c.get(ims[2])
    assert len(c) == 3
    c.get(ims[3])                                   # 4 x 48 bytes > bound: the oldest goes
    assert len(c) == 3 and c.get(ims[0]) is not a0 and len(ups) == 5
    # a different array that happens to present the same key is a miss
    key = id(ims[3])
    c._d[key] = (ims[2], c._d[key][1], 48)
    assert int(c.get(ims[3])[0, 0, 0]) == 3


def test_test_time_executors_share_one_parameter_set():
    """share_params: masters, fp16 copies, moving statistics and the BatchNorm-folded weights of a second bound shape ARE the
    first shape's tensors; a training executor never shares"""
    from sniper_amd import config as cfgmod
    from sniper_amd.engine.executor import Executor
    from sniper_amd.symbols.faster import resnet_mx_101_e2e as ours
    cfg = cfgmod.res101_e2e_autofocus()
    sym = ours.resnet_mx_101_e2e(n_proposals=400, test_nbatch=1).get_symbol_rcnn(cfg, is_train=False)
    args = set(sym.list_arguments())

    def bind(h, w, share=None):
        shapes = {k: v for k, v in dict(data=(1, 3, h, w), im_info=(1, 3), im_ids=(1,), chip_ids=(1,)).items() if k in args}
        return Executor(sym, shapes, False, (), device=torch.device('cpu'), share_params=share)
    a = bind(192, 256)
    b = bind(128, 192, a)
    assert not a.shared_names and b.fold_store is a.fold_store
    assert set(b.shared_names) == set(b.params) | set(b.aux)
    for name in ('conv0_weight', 'stage3_unit1_conv2_weight', 'fc_new_1_weight', 'rpn_conv_3x3_weight'):
        assert b.params[name].master.data_ptr() == a.params[name].master.data_ptr()
        assert b.params[name].w16.data_ptr() == a.params[name].w16.data_ptr()
    assert b.aux['stage1_unit1_bn2_moving_var'].data_ptr() == a.aux['stage1_unit1_bn2_moving_var'].data_ptr()
    c = bind(128, 192)
    assert c.params['conv0_weight'].master.data_ptr() != a.params['conv0_weight'].master.data_ptr()
    # round 6: what is DERIVED from the parameters alone is the Module's too -- a BatchNorm's scale / shift (derived_buffer) -- and a
    # further shape adopts it instead of recomputing (Executor.adopt_derived), but only once an executor of the Module has derived
    # everything from the current parameters (the '__valid__' mark refresh_compute_copies leaves; set_params clears it first)
    bn_a = next(s for s in a.steps if type(s).__name__ == 'BatchNormStep' and s.node.name == 'stage3_unit1_bn1')
    bn_b = next(s for s in b.steps if type(s).__name__ == 'BatchNormStep' and s.node.name == 'stage3_unit1_bn1')
    assert bn_b.scale.data_ptr() == bn_a.scale.data_ptr() and bn_b.shift.data_ptr() == bn_a.shift.data_ptr()
    assert not b.adopt_derived()                      # nothing derived yet: no parameters were ever set
    a.fold_store['__valid__'] = True                  # (what a.set_params(...) leaves on a device; no kernels on the CPU)
    folded = next(s for s in b.steps if type(s).__name__ == 'BatchNormStep' and s.folded_into is not None)
    assert not b.adopt_derived()                      # ... and the folded weights must be there as well
    for s in a.steps:
        if type(s).__name__ == 'BatchNormStep' and s.folded_into is not None:
            w = s.folded_into.w
            a.fold_store[w.name] = (torch.zeros_like(w.w16), torch.zeros(w.w16.shape[0]))
    assert b.adopt_derived()
    assert folded.folded_into.wf is a.fold_store[folded.folded_into.w.name][0] and folded._global_ready
    assert not c.adopt_derived()                      # an executor with its own parameters never adopts


def test_closed_prefetching_iter_dies_by_reference_counting_even_with_a_frozen_heap():
    """a closure over self kept on self made a closed PrefetchingIter a reference cycle; frozen by settle_heap it (and
    the pass's device image cache behind it) stayed alive until the next thaw."""
    import gc
    import weakref
    from sniper_amd.iterators.PrefetchingIter import PrefetchingIter

    class Payload(object):
        pass

    class It(object):
        provide_data, provide_label = [('data', (1, 3, 4, 4))], []
        batch_size = 1

        def __init__(self):
            self.payload = Payload()

        def reset(self):
            pass

        def next(self):
            raise StopIteration

    inner = It()
    alive = weakref.ref(inner.payload)
    it = PrefetchingIter(inner)
    gc.collect()
    gc.freeze()
    try:
        gc.disable()
        it.close()
        del it, inner
        assert alive() is None, 'the closed iterator is still reachable: a cycle the frozen collector cannot see'
    finally:
        gc.enable()
        gc.unfreeze()
