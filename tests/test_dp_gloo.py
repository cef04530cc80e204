"""CPU, world_size 2 over gloo: the data-parallel pieces -- rendezvous from the torchrun environment, rank slicing of a
global batch, the bucketed gradient-arena all-reduce (sum, in place), and bench.py's max-over-ranks timing reduction.
The kernels need a GPU; what is exercised here is everything that differs between N = 1 and N > 1."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world_size, port, out):
    os.environ.update(WORLD_SIZE=str(world_size), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from sniper_amd import parallel
    dist = parallel.init(backend='gloo')
    assert dist is not None and dist.get_world_size() == world_size and parallel.world() == (world_size, rank, rank)
    # every rank owns an independent chip minibatch: its gradients differ
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    n_coll = parallel.allreduce_gradients(g, dist, bucket_bytes=1024)          # 256 floats per bucket -> 4 collectives
    want = torch.arange(1000, dtype=torch.float32) * sum(r + 1 for r in range(world_size))
    ok_sum = bool(torch.equal(g, want)) and n_coll == 4
    # fp16 bucket: the first 300 gradients travel in fp16 (the weights the reference holds in fp16), the rest in fp32
    g2 = torch.cat((torch.arange(300, dtype=torch.float32) * (rank + 1) / 8, torch.full((700,), 0.1 * (rank + 1))))
    buf = torch.empty(512, dtype=torch.float16)
    n2 = parallel.allreduce_gradients(g2, dist, bucket_bytes=1 << 20, half_elems=300, half_buf=buf)
    tot = sum(r + 1 for r in range(world_size))
    ok_sum = ok_sum and n2 == 2 and bool(torch.equal(g2[:300], torch.arange(300, dtype=torch.float32) * tot / 8)) and \
        bool(torch.allclose(g2[300:], torch.full((700,), 0.1 * tot), rtol=1e-6)) and \
        not bool(torch.equal(g2[300:].half().float(), g2[300:]))          # the fp32 part did NOT go through fp16
    # a global batch is split rank-major
    batch = np.arange(8 * 3).reshape(8, 3)
    mine = parallel.rank_slice(batch, rank, world_size)
    ok_slice = mine.shape == (4, 3) and mine[0, 0] == rank * 12
    # bench.py's reduction: the job time is the slowest rank's
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok_time = abs(float(t) - 0.1 * world_size) < 1e-12
    # Module's view of the shapes: per-rank batch = global / world
    import sniper_amd.mx as mx
    m = mx.mod.Module(mx.sym.Variable('x'), data_names=['x'], label_names=None)
    ok_local = m.world == world_size and m._local((8, 3, 4, 4)) == (4, 3, 4, 4)
    # epoch synchronisation: every rank builds its epoch from numpy's global RNG; unsynchronised ranks would slice different
    # global batches.  Module._sync_epoch re-seeds from rank 0 and resets the iterator: identical order everywhere.
    class FakeIter(object):
        def __init__(self, n):
            self.n, self.order = n, None

        def reset(self):
            self.order = np.random.permutation(self.n)

        def __len__(self):
            return self.n
    m._device = torch.device('cpu')
    np.random.seed(1234 + 77 * rank)                      # ranks start out of sync, as separately launched processes do
    it = FakeIter(16)
    assert m._sync_epoch(it, 0)
    o = torch.from_numpy(it.order.copy())
    lo, hi = o.clone(), o.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    ok_sync = bool(torch.equal(lo, hi))
    m._sync_epoch(it, 1)
    ok_sync = ok_sync and not np.array_equal(it.order, o.numpy())      # a new epoch is a new order
    try:
        m._sync_epoch(FakeIter(16 + 4 * rank), 2)                      # ranks disagree on the epoch length
        ok_len = False
    except RuntimeError:
        ok_len = True
    # rank-sliced global-batch iterator (sniper_amd/ext/rank_slice.py): an iterator with MNIteratorE2E's batch contract
    # (cur_i / size / batch_size / _get_batch, lib/iterators/MNIteratorE2E.py:105-117) prepares B chips per step on each rank, not
    # W * B; the ranks' slices are disjoint and together the global batch of the unwrapped iterator, the cursor still walks
    # global batches, and Module.fit's adoption makes the Module bind the per-rank shape without slicing again
    from sniper_amd.ext import rank_slice

    class RefLikeIter(object):
        def __init__(self, n, batch_size):
            self.size, self.batch_size, self.cur_i = n, batch_size, 0
            self.inds = np.random.permutation(n)
            self.prepared = []                     # chips this process cropped / labelled, per step

        def reset(self):
            self.cur_i = 0
            self.inds = np.random.permutation(self.size)

        def __len__(self):
            return self.size

        def get_batch(self):
            if self.cur_i >= self.size:
                return False
            self.batch = self._get_batch()
            self.cur_i += self.batch_size
            return True

        def _get_batch(self):
            ids = [int(self.inds[i % self.size]) for i in range(self.cur_i, self.cur_i + self.batch_size)]
            self.prepared.append(ids)
            return np.asarray(ids)

        @property
        def provide_data(self):
            return [('x', (len(self.batch), 3, 4, 4))]
    rank_slice.patch_iterator_class(RefLikeIter)
    B = 4
    it2 = RefLikeIter(6 * world_size * B, world_size * B)
    m2 = mx.mod.Module(mx.sym.Variable('x'), data_names=['x'], label_names=None)
    m2._device = torch.device('cpu')
    m2._adopt_iterator(it2)
    assert m2._sync_epoch(it2, 0)                  # same permutation on every rank
    steps, mine_all = 0, []
    while it2.get_batch():
        steps += 1
        mine_all.append(torch.from_numpy(it2.batch.copy()))
    ok_rs = steps == 6 and all(len(p) == B for p in it2.prepared) and it2.cur_i == it2.size and it2.batch_size == world_size * B
    ok_rs = ok_rs and m2.slice_inputs is False and m2._local(it2.provide_data[0][1]) == (B, 3, 4, 4)
    got = [torch.zeros(6 * B, dtype=torch.int64) for _ in range(world_size)]
    dist.all_gather(got, torch.cat(mine_all))
    per_step = torch.stack([g.view(6, B) for g in got], 1).reshape(6, world_size * B)       # rank-major inside a global batch
    ok_rs = ok_rs and bool(torch.equal(per_step.reshape(-1), torch.from_numpy(it2.inds.astype(np.int64))))
    dist.barrier()
    out[rank] = int(ok_sum) + 2 * int(ok_slice) + 4 * int(ok_time) + 8 * int(ok_local) + 16 * int(ok_sync) + 32 * int(ok_len) + \
        64 * int(ok_rs)
    dist.destroy_process_group()


def test_two_rank_gradient_exchange_gloo():
    ctx = mp.get_context('spawn')
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, 'rank exited with %s' % p.exitcode
    assert dict(out) == {0: 127, 1: 127}, dict(out)


def test_rank_sliced_iterator_walks_the_chips_of_the_unsliced_one(monkeypatch):
    """An iterator with MNIteratorE2E's per-image chip cursor (lib/iterators/MNIteratorE2E.py:118-129: chip
    `chip_order[crop_idx[image] % n]` for every entry, cursors read at the start of the batch and advanced for every entry):
    images occur several times per epoch (one entry per chip), so the chips of one image land in different ranks' slices.
    The ranks' slices together must be exactly the (image, chip) sequence of the unsliced iterator, over two epochs, and every
    rank's cursors must end equal to the unsliced iterator's (each rank used to advance only its own entries)."""
    from sniper_amd.ext import rank_slice

    class ChipCursorIter(object):
        def __init__(self, n_images, batch_size, seed):
            rs = np.random.RandomState(seed)
            self.n_chips = rs.randint(1, 6, n_images)
            self.roidb = [dict(chip_order=rs.permutation(int(k))) for k in self.n_chips]
            self.batch_size, self.seed, self.epoch = batch_size, seed, 0
            self.reset()

        def reset(self):
            rs = np.random.RandomState(self.seed + self.epoch)
            self.epoch += 1
            inds = np.repeat(np.arange(len(self.roidb)), self.n_chips)       # one entry per chip of every image
            self.inds = inds[rs.permutation(len(inds))]
            self.size = len(self.inds) // self.batch_size * self.batch_size
            self.crop_idx = [0] * len(self.roidb)
            self.cur_i = 0

        def get_batch(self):
            if self.cur_i >= self.size:
                return False
            self.batch = self._get_batch()
            self.cur_i += self.batch_size
            return True

        def _get_batch(self):
            rng = range(self.cur_i, self.cur_i + self.batch_size)
            cropids = [self.roidb[self.inds[i]]['chip_order'][self.crop_idx[self.inds[i]] % len(self.roidb[self.inds[i]]['chip_order'])]
                       for i in rng]
            for i in rng:
                self.crop_idx[self.inds[i]] = self.crop_idx[self.inds[i]] + 1
            return [(int(self.inds[i]), int(c)) for i, c in zip(rng, cropids)]

    class Sliced(ChipCursorIter):
        pass
    rank_slice.patch_iterator_class(Sliced)
    world, B = 4, 3
    monkeypatch.delenv('SNIPER_RANK_SLICE', raising=False)
    ref = ChipCursorIter(40, world * B, seed=3)
    monkeypatch.setenv('WORLD_SIZE', str(world))
    its = []
    for r in range(world):
        its.append(Sliced(40, world * B, seed=3))
    for epoch in range(2):
        steps = 0
        while ref.get_batch():
            steps += 1
            merged = []
            for r, it in enumerate(its):
                monkeypatch.setenv('RANK', str(r))
                assert it.get_batch() and len(it.batch) == B
                merged += it.batch
            assert merged == ref.batch, (epoch, steps)
            for it in its:
                assert it.crop_idx == ref.crop_idx and it.cur_i == ref.cur_i
        assert steps == ref.size // (world * B) and steps > 5
        for r, it in enumerate(its):
            monkeypatch.setenv('RANK', str(r))
            assert not it.get_batch()
            it.reset()
        ref.reset()
    # an image with several chips did cross slices (otherwise the test would not see the defect it is here for)
    seen = {}
    ref.reset()
    while ref.get_batch():
        for k, (im, _) in enumerate(ref.batch):
            seen.setdefault(im, set()).add(k // B)
    assert any(len(v) > 1 for v in seen.values())


def test_single_process_is_a_no_op():
    from sniper_amd import parallel
    g = torch.ones(10)
    assert parallel.allreduce_gradients(g, None) == 0 and torch.equal(g, torch.ones(10))


def test_ext_install_scopes_the_pool_to_the_reference_modules(tmp_path, monkeypatch):
    """sniper_amd.ext.pool: after install(), a module named like the reference's `iterators/MNIteratorE2E.py` or `inference.py`
    that does `from multiprocessing import Pool` sees the drop-in pool under that name (post-import hook, file untouched), while
    `multiprocessing.Pool` itself and every other module keep the real process pool (lib/dataset/imdb.py:81-118 forks numpy-only
    workers).  The drop-in has Pool's interface; unrecognised work items run on threads of THIS process, map() keeps order,
    initializer / initargs are honoured; the reference's chip_worker / nms_worker bound methods are recognised for batching."""
    import importlib
    import multiprocessing
    import threading
    from sniper_amd.ext import pool
    real = multiprocessing.Pool
    pkg = tmp_path / 'refpkg'
    (pkg / 'iterators').mkdir(parents=True)
    (pkg / 'iterators' / '__init__.py').write_text('')
    src = 'from multiprocessing import Pool\n\n\nclass MNIteratorE2E(object):\n    def get_batch(self):\n        return True\n\n    def _get_batch(self):\n        return None\n'
    (pkg / 'iterators' / 'MNIteratorE2E.py').write_text(src)
    (pkg / 'inference.py').write_text('from multiprocessing import Pool\n\n\nclass Tester(object):\n    pass\n')
    (pkg / 'imdb_like.py').write_text('from multiprocessing import Pool\n')
    monkeypatch.syspath_prepend(str(pkg))
    for name in ('iterators', 'iterators.MNIteratorE2E', 'inference', 'imdb_like'):
        monkeypatch.delitem(sys.modules, name, raising=False)
    pool.install()
    pool.install()                                                            # idempotent
    assert multiprocessing.Pool is real
    m_it, m_inf, m_other = (importlib.import_module(n) for n in ('iterators.MNIteratorE2E', 'inference', 'imdb_like'))
    assert m_it.Pool is pool.Pool and m_inf.Pool is pool.Pool and m_other.Pool is real
    seen = []
    p = m_it.Pool(4, initializer=seen.append, initargs=('init',))
    pid = os.getpid()
    import time
    res = p.map(lambda i: (time.sleep(0.02), i * i, os.getpid(), threading.current_thread().name)[1:], range(32))
    r2 = p.map_async(lambda i: i + 1, range(5)).get(10)
    p.close()
    p.join()
    assert [r[0] for r in res] == [i * i for i in range(32)] and r2 == [1, 2, 3, 4, 5]
    assert all(r[1] == pid for r in res) and len(set(r[2] for r in res)) > 1
    assert seen == ['init'] * 4 and p.routed_maps == 0

    # recognition of the reference's work items (the batched calls themselves need the GPU: tests/test_gpu_acceptance.py)
    class chip_worker(object):                      # lib/data_utils/data_workers.py:374-392
        valid_ranges, scales, chip_size, use_neg_chips, chip_stride = ((-1, 80), (32, 150), (120, -1)), ((1400, 2000), (800, 1280), (-1, 512)), 512, False, 57

        def chip_extractor(self, r):
            raise AssertionError('per-item path')

        def box_assigner(self, r):
            raise AssertionError('per-item path')

        def other(self, r):
            return r

    class nms_wrapper(object):
        thresh, sigma = -1, 0.55

    class nms_worker(object):                       # lib/data_utils/data_workers.py:124-129
        def __init__(self):
            self.nms_wrapper = nms_wrapper()

        def worker(self, data):
            raise AssertionError('per-item path')
    cw = chip_worker()
    f = pool.route(cw.chip_extractor)
    assert f is not None and f.__name__ == 'extract_batch' and f.__self__.chip_stride == 57 and f.__self__.res_based
    cw.chip_stride = 58                             # chip_worker.reset() draws a new stride every epoch
    assert pool.route(cw.box_assigner).__self__.chip_stride == 58 and pool.route(cw.box_assigner).__name__ == 'assign_batch'
    assert pool.route(cw.other) is None and pool.route(lambda r: r) is None and pool.route(len) is None
    assert pool.route(nms_worker().worker).__name__ == 'process_many'
    monkeypatch.setenv('SNIPER_POOL_ROUTE', '0')
    assert pool.route(cw.chip_extractor) is None
