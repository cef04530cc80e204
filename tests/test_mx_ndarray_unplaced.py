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


def test_index_array_writes_into_device_arrays_land_in_the_array(monkeypatch):
    """Fancy-index writes into a device NDArray (`row[pids0, pids1, pids2] = values` with plain numpy pids; tuple-of-array and mask
    indices on the array itself): torch's advanced indexing returns a copy, so the write must be an in-place index assignment.
    Simulated on CPU tensors: `_is_device` patched to accept them (the GPU twin: tests/test_gpu_acceptance.py)."""
    import torch
    from sniper_amd.mx import ndarray as nd
    monkeypatch.setattr(nd, '_is_device', lambda v: getattr(v, '_device_resident', False) or
                        isinstance(v._store if isinstance(v, nd.NDArray) else v, torch.Tensor))
    a = nd.NDArray(torch.zeros((2, 4, 3, 3)))
    i0, i1, i2 = np.array([0, 1, 3]), np.array([2, 2, 0]), np.array([0, 1, 2])
    a[1][i0, i1, i2] = np.array([5.0, 6.0, 7.0], np.float32)
    out = a._store.numpy()
    assert out[1, 0, 2, 0] == 5 and out[1, 1, 2, 1] == 6 and out[1, 3, 0, 2] == 7 and out.sum() == 18 and out[0].sum() == 0
    a[1][i0, i1, i2] = 1.0                                   # the scalar form (`bbox_weights[i][pids...] = 1.0`)
    assert a._store.numpy().sum() == 3
    b = nd.NDArray(torch.zeros((4, 5)))
    b[(np.array([0, 3]), np.array([1, 4]))] = 7
    assert b._store.numpy()[0, 1] == 7 and b._store.numpy()[3, 4] == 7 and b._store.numpy().sum() == 14
    b[nd.array(np.array([1, 2]))] = torch.ones(5)            # an NDArray of row indices, a tensor value
    assert b._store.numpy()[1:3].sum() == 10
    b[1] = np.arange(5)                                       # the basic-index fast path still writes through a view
    b[2, 1:3] = np.array([9.0, 9.0])
    assert b._store.numpy()[1].tolist() == [0, 1, 2, 3, 4] and b._store.numpy()[2].tolist() == [1, 9, 9, 1, 1]


def test_integer_index_of_a_one_dimensional_unplaced_array():
    z = mx.nd.zeros(5)
    assert float(z[2].asnumpy()) == 0.0 and z[2].shape == ()
    z[3] = 4.0
    assert z.asnumpy().tolist() == [0, 0, 0, 4, 0]
