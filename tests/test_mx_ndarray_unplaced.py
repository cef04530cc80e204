"""CPU: the shim's UNPLACED arrays (sniper_amd/mx/ndarray.py).  `mx.nd.zeros` / `ones` / `-ones` allocate nothing until first used; any
host use gives exactly the array MXNet would have made -- the reference iterator's assembly lines (lib/iterators/MNIteratorE2E.py:175-201)
run unchanged on them.  (Placement on the device by an HBM-resident write: tests/test_gpu_acceptance.py.)"""
import numpy as np

import sniper_amd.mx as mx


def test_unplaced_arrays_behave_like_host_arrays():
    a = mx.nd.zeros((3, 4), mx.cpu(0))
    assert a._store is None and a.shape == (3, 4) and a.dtype == np.float32 and len(a) == 3 and a.size == 12
    a[1] = np.arange(4)
    assert a._store is not None and a.asnumpy().tolist() == [[0, 0, 0, 0], [0, 1, 2, 3], [0, 0, 0, 0]]
    b = -mx.nd.ones((2, 3, 5))                                     # MNIteratorE2E.py:180: gt_boxes = -mx.nd.ones((n, 100, 5))
    assert b._store is None and b.shape == (2, 3, 5)
    b[0] = np.full((3, 5), 7.0)
    assert (b.asnumpy()[0] == 7).all() and (b.asnumpy()[1] == -1).all()
    c = mx.nd.zeros((2, 4, 3, 3))                                  # :191: bbox_targets[i][pids[0], pids[1], pids[2]] = values
    pid = mx.nd.array(np.array([[0, 1], [2, 2], [0, 1]]))
    c[1][pid[0], pid[1], pid[2]] = np.array([5.0, 6.0], np.float32)
    c[1][pid[0], pid[1], pid[2]] = 1.0 * c[1].asnumpy()[0, 2, 0] + np.array([0.0, 1.0])    # a second write through a fresh row view
    out = c.asnumpy()
    assert out[1, 0, 2, 0] == 5.0 and out[1, 1, 2, 1] == 6.0 and out.sum() == 11.0 and out[0].sum() == 0
    d = mx.nd.zeros((2, 3))
    assert ((d + 1).asnumpy() == 1).all() and d[0].shape == (3,) and d[0].asnumpy().tolist() == [0, 0, 0]
    e = mx.nd.zeros((4,), dtype='float16')
    e[:] = 3
    assert e.dtype == np.float16 and e.asnumpy().tolist() == [3, 3, 3, 3]
    f = mx.nd.ones((2, 2))
    f *= 4
    assert f.asnumpy().tolist() == [[4, 4], [4, 4]]
    assert mx.nd.zeros(5).shape == (5,) and mx.nd.array(mx.nd.zeros((2, 2))).asnumpy().shape == (2, 2)


def test_a_routed_workers_handles_materialise_the_reference_values_on_the_host():
    """The handles a routed `anchor_worker.worker` map returns (sniper_amd/ext/pool.py) written into a HOST-placed array: the sparse
    (values, pids) pair materialises as the reference worker's `bbox_targets[pids]` / `np.where(bbox_weights == 1)`."""
    from sniper_amd.ext import pool

    class FakeTensor(object):                 # stands for the device tensors of a batch (only .cpu().numpy() is used on this path)
        def __init__(self, a):
            self.a = a

        def cpu(self):
            return self

        def numpy(self):
            return self.a

        def __getitem__(self, i):
            return FakeTensor(self.a[i])
    rs = np.random.RandomState(0)
    w = (rs.rand(2, 8, 3, 3) < 0.2).astype(np.float32)
    t = (rs.standard_normal((2, 8, 3, 3)) * w).astype(np.float32)
    batch = pool._AnchorBatch({'bbox_target': FakeTensor(t), 'bbox_weight': FakeTensor(w)})
    tgt, wgt = mx.nd.zeros((2, 8, 3, 3), mx.cpu(0)), mx.nd.zeros((2, 8, 3, 3), mx.cpu(0))
    tgt[0] = np.zeros((8, 3, 3))              # a host write first: the arrays live on the host
    wgt[0] = np.zeros((8, 3, 3))
    for i in range(2):
        vals, pids = pool._SparseVals(batch, i), pool._SparsePids(batch, i)
        if len(pids[0]) > 0:
            tgt[i][pids[0], pids[1], pids[2]] = vals
            wgt[i][pids[0], pids[1], pids[2]] = 1.0
        assert np.array_equal(np.asarray(pids).astype(int), np.stack(np.where(w[i] == 1)))
        assert np.array_equal(np.asarray(vals), t[i][np.where(w[i] == 1)])
    assert np.array_equal(tgt.asnumpy(), t) and np.array_equal(wgt.asnumpy(), w)
