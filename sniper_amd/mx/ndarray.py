"""``mx.nd`` subset: host arrays are numpy-backed, device arrays wrap a torch
tensor (torch = memory/stream plumbing).  Only what the reference's iterators, metrics, callbacks and
weight initialisers call is provided."""
import numpy as np

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


class Context(object):
    def __init__(self, device_type, device_id=0):
        self.device_type, self.device_id = device_type, int(device_id)

    def __repr__(self):
        return '%s(%d)' % (self.device_type, self.device_id)

    def __eq__(self, o):
        return isinstance(o, Context) and (self.device_type, self.device_id) == (o.device_type, o.device_id)

    def __hash__(self):
        return hash((self.device_type, self.device_id))


def cpu(i=0):
    return Context('cpu', i)


def gpu(i=0):
    return Context('gpu', i)


_DT = {'float32': np.float32, 'float16': np.float16, 'float64': np.float64, 'int32': np.int32, 'uint8': np.uint8,
       'int64': np.int64}


def _dtype(d):
    if d is None:
        return np.float32
    if isinstance(d, str):
        return _DT[d]
    return np.dtype(d).type


def _is_device(v):
    """a value that lives in HBM: a CUDA tensor, an NDArray backed by one, or a routed worker's marker (ext/pool.py)"""
    if getattr(v, '_device_resident', False):
        return True
    t = v._store if isinstance(v, NDArray) else v
    return torch is not None and isinstance(t, torch.Tensor) and t.is_cuda


class NDArray(object):
    """Array handle.  ``_data`` is a numpy array (cpu context) or a torch tensor (gpu context).

    UNPLACED arrays: ``zeros`` / ``ones`` (and their negation) allocate nothing.  The first write decides where the array lives: a
    value that is resident in HBM (a batch tensor a routed worker produced, sniper_amd/ext/pool.py) places it on the device and the
    write is a device copy; anything else -- and any read -- materialises the host array MXNet would have made.  The reference's
    ``MNIteratorE2E._get_batch`` (lib/iterators/MNIteratorE2E.py:175-201) builds its batch as ``mx.nd.zeros(...)`` + per-chip
    writes: with GPU-resident worker results the 63 MB image tensor and the dense labels are then born in HBM instead of being
    assembled on the host and uploaded from pageable memory every step."""
    __array_priority__ = 100.0

    def __init__(self, data, ctx=None, lazy=None):
        self._store = data
        self._lazy = lazy                        # (shape, numpy dtype, fill) while unplaced
        self.context = ctx or (cpu() if data is None or isinstance(data, np.ndarray) else gpu(0))

    @property
    def _data(self):
        if self._store is None:
            shape, dt, fill = self._lazy
            self._store = np.full(shape, fill, dtype=dt) if fill else np.zeros(shape, dtype=dt)
            self._lazy = None
        return self._store

    @_data.setter
    def _data(self, v):
        self._store, self._lazy = v, None

    def _place_on(self, device):
        """an unplaced array becomes a device array (filled on the caller's current stream)"""
        shape, dt, fill = self._lazy
        tdt = getattr(torch, np.dtype(dt).name)
        self._store = torch.full(tuple(shape), float(fill), dtype=tdt, device=device) if fill else \
            torch.zeros(tuple(shape), dtype=tdt, device=device)
        self._lazy = None
        self.context = gpu(device.index or 0)

    # ---- basics
    @property
    def shape(self):
        if self._store is None:
            return tuple(self._lazy[0])
        return tuple(self._store.shape)

    @property
    def dtype(self):
        if self._store is None:
            return np.dtype(self._lazy[1]).type
        if isinstance(self._data, np.ndarray):
            return self._data.dtype.type
        return np.dtype(str(self._data.dtype).replace('torch.', '')).type

    @property
    def size(self):
        return int(np.prod(self.shape))

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def T(self):
        return NDArray(self.asnumpy().T.copy())

    def asnumpy(self):
        if isinstance(self._data, np.ndarray):
            return self._data
        ev = getattr(self, '_ready_event', None)       # assembled on a prefetch worker's stream (PrefetchingIter.stamp_ready)
        if ev is not None:
            ev.synchronize()
            self._ready_event = None
        return self._data.detach().float().cpu().numpy() if self._data.dtype in (torch.float16, torch.bfloat16) \
            else self._data.detach().cpu().numpy()

    def asscalar(self):
        return self.asnumpy().reshape(-1)[0]

    def astype(self, dtype):
        return NDArray(self.asnumpy().astype(_dtype(dtype)))

    def copy(self):
        return NDArray(self.asnumpy().copy())

    def copyto(self, other):
        if isinstance(other, NDArray):
            other[:] = self
            return other
        return self.as_in_context(other)

    def as_in_context(self, ctx):
        return self  # placement is decided by the executor; host copies stay numpy-backed

    def wait_to_read(self):
        if not isinstance(self._data, np.ndarray):
            torch.cuda.synchronize()

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = shape[0]
        return NDArray(self.asnumpy().reshape(shape))

    def __len__(self):
        return self.shape[0]

    def __repr__(self):
        return '<NDArray %s @%s>\n%r' % ('x'.join(map(str, self.shape)), self.context, self.asnumpy())

    # ---- indexing (MNIteratorE2E.py:186-194 uses int, slice and fancy-tuple indices)
    @staticmethod
    def _idx(i):
        if isinstance(i, NDArray):
            return i.asnumpy().astype(np.int64)
        if isinstance(i, tuple):
            return tuple(NDArray._idx(j) for j in i)
        return i

    def __getitem__(self, i):
        if isinstance(i, (int, np.integer)) and len(self.shape) > 1 and (self._store is None or _is_device(self)):
            return _RowView(self, int(i))          # row of an unplaced / device array: writes through (and may place the parent)
        a = self.asnumpy()[self._idx(i)]
        if isinstance(self._data, np.ndarray) and isinstance(a, np.ndarray) and a.base is not None and not isinstance(
                self._idx(i), tuple):
            return NDArray(a)  # a view: writes through, like mx.nd slices of the first axis
        return NDArray(np.asarray(a))

    def __setitem__(self, i, v):
        if self._store is None and _is_device(v):
            dev = v.device if hasattr(v, 'device') and not isinstance(v, NDArray) else (v._store.device if isinstance(v, NDArray) else None)
            self._place_on(dev if dev is not None else torch.device('cuda', torch.cuda.current_device()))
        if _is_device(self):
            idx = self._idx(i)
            if not _is_basic_index(idx):
                # index arrays / masks: torch's advanced indexing returns a COPY, so `view.copy_` would write into a temporary --
                # in-place index assignment instead (`row[pids0, pids1, pids2] = values` with plain numpy pids)
                if hasattr(v, 'write_dense_into'):
                    raise TypeError('a routed worker result cannot be written through an index array')
                src = v._store if isinstance(v, NDArray) else v
                if not isinstance(src, torch.Tensor):
                    src = torch.as_tensor(np.asarray(src.asnumpy() if isinstance(src, NDArray) else src))
                tidx = tuple(torch.as_tensor(np.asarray(j), device=self._store.device) if isinstance(j, (np.ndarray, list)) else j
                             for j in (idx if isinstance(idx, tuple) else (idx,)))
                self._store[tidx] = src.to(device=self._store.device, dtype=self._store.dtype)
                return
            dst = self._store[idx] if not (isinstance(i, slice) and i == slice(None)) else self._store
            if hasattr(v, 'write_dense_into'):                 # a routed worker's marker: the dense device tensor it stands for
                v.write_dense_into(dst)
                return
            src = v._store if isinstance(v, NDArray) else v
            if isinstance(src, torch.Tensor):
                dst.copy_(src.reshape(dst.shape) if src.numel() == dst.numel() else src)
            else:
                dst.copy_(torch.as_tensor(np.asarray(v.asnumpy() if isinstance(v, NDArray) else v)).to(dst.dtype))
            return
        if isinstance(v, NDArray):
            v = v.asnumpy()
        elif hasattr(v, '__array__') and not isinstance(v, np.ndarray):
            v = np.asarray(v)
        self._data[self._idx(i)] = v

    # ---- arithmetic
    def _bin(self, o, f):
        if isinstance(o, NDArray):
            o = o.asnumpy()
        return NDArray(np.asarray(f(self.asnumpy(), o)))

    def __add__(self, o): return self._bin(o, lambda a, b: a + b)
    def __radd__(self, o): return self._bin(o, lambda a, b: b + a)
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._bin(o, lambda a, b: b - a)
    def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._bin(o, lambda a, b: b * a)
    def __truediv__(self, o): return self._bin(o, lambda a, b: a / b)
    __div__ = __truediv__
    def __neg__(self):
        if self._store is None and not isinstance(self, _RowView):          # -mx.nd.ones(...) stays unplaced (MNIteratorE2E.py:180)
            shape, dt, fill = self._lazy
            return NDArray(None, None, lazy=(shape, dt, -fill))
        return NDArray(-self.asnumpy())

    def __iadd__(self, o):
        self[:] = self + o
        return self

    def __imul__(self, o):
        self[:] = self * o
        return self


class _RowView(NDArray):
    """``a[i]`` of an unplaced or device-resident array: a handle that writes through to row i (MNIteratorE2E.py:189-194 writes
    ``bbox_targets[i][pids[0], pids[1], pids[2]] = values`` and ``labels[i] = ...``).  A write of an HBM-resident value / marker
    into a row of an unplaced parent places the PARENT on the device."""

    def __init__(self, parent, index):
        self._parent, self._index = parent, index
        self._lazy = None
        self.context = parent.context

    @property
    def _store(self):
        p = self._parent
        return None if p._store is None else p._store[self._index]

    @_store.setter
    def _store(self, v):       # (never rebound)
        raise AttributeError('a row view has no storage of its own')

    @property
    def _data(self):
        return self._parent._data[self._index]

    @property
    def shape(self):
        return tuple(self._parent.shape[1:])

    @property
    def dtype(self):
        return self._parent.dtype

    def __setitem__(self, i, v):
        p = self._parent
        marker = v if hasattr(v, 'write_dense_into') else None
        if marker is None and isinstance(i, tuple) and i and all(hasattr(j, 'dense_source') for j in i):
            marker = i[0].dense_source(v)              # `row[pids[0], pids[1], pids[2]] = values | 1.0` of a routed anchor worker
        if p._store is None and (marker is not None or _is_device(v)):
            dev = marker.device if marker is not None else (v._store.device if isinstance(v, NDArray) else v.device)
            p._place_on(dev)
        if marker is not None and _is_device(p):
            marker.write_dense_into(p._store[self._index])
            return
        if marker is not None:                         # host parent: the marker's real sparse values
            i, v = tuple(np.asarray(j) for j in i), np.asarray(v) if hasattr(v, '__array__') else v
        if _is_device(p):
            NDArray(p._store[self._index]).__setitem__(i, v)
            return
        if isinstance(v, NDArray):
            v = v.asnumpy()
        p._data[self._index][self._idx(i)] = v


def _is_basic_index(idx):
    """ints, slices, Ellipsis / None (and tuples of them) select a VIEW; anything else (index arrays, lists, masks) is advanced"""
    basic = (int, np.integer, slice, type(Ellipsis), type(None))
    if isinstance(idx, tuple):
        return all(isinstance(j, basic) for j in idx)
    return isinstance(idx, basic)


def array(source, ctx=None, dtype=None):
    if isinstance(source, NDArray):
        source = source.asnumpy()
    a = np.array(source, dtype=_dtype(dtype) if dtype is not None else None)
    if dtype is None and a.dtype != np.float32:
        a = a.astype(np.float32)  # mx.nd.array defaults to float32
    return NDArray(a, ctx)


def _shape(shape):
    return (int(shape),) if isinstance(shape, (int, np.integer)) else tuple(int(x) for x in shape)


def zeros(shape, ctx=None, dtype=None, **kw):
    return NDArray(None, ctx, lazy=(_shape(shape), _dtype(dtype), 0.0))          # unplaced: see NDArray


def ones(shape, ctx=None, dtype=None, **kw):
    return NDArray(None, ctx, lazy=(_shape(shape), _dtype(dtype), 1.0))


def empty(shape, ctx=None, dtype=None):
    return zeros(shape, ctx, dtype)


def full(shape, val, ctx=None, dtype=None):
    return NDArray(np.full(shape, val, dtype=_dtype(dtype)), ctx)


def sum(a, axis=None, **kw):  # noqa: A001
    return NDArray(np.asarray(np.sum(a.asnumpy(), axis=axis)))


def argmax_channel(a):
    """mx.ndarray.argmax_channel: argmax over axis 1 (metric.py:59,114)."""
    return NDArray(np.argmax(a.asnumpy(), axis=1).astype(np.float32))


def smooth_l1(data, scalar=1.0):
    x = data.asnumpy()
    s2 = scalar * scalar
    return NDArray(np.where(np.abs(x) < 1.0 / s2, 0.5 * s2 * x * x, np.abs(x) - 0.5 / s2))


def SoftmaxActivation(data, mode='instance'):
    x = data.asnumpy()
    ax = 1 if mode == 'channel' else -1
    e = np.exp(x - x.max(axis=ax, keepdims=True))
    return NDArray(e / e.sum(axis=ax, keepdims=True))


def concatenate(arrays, axis=0):
    return NDArray(np.concatenate([a.asnumpy() for a in arrays], axis=axis))


def save(fname, data):
    """mx.nd.save: MXNet's binary NDArray-list container (sniper_amd/mx/params_io.py), so that checkpoints written by
    `mx.model.save_checkpoint` (resnet_mx_101_e2e.py:14) are readable by MXNet tooling and vice versa."""
    from . import params_io
    if isinstance(data, dict):
        arrs = {k: v.asnumpy() for k, v in data.items()}
    else:
        arrs = [v.asnumpy() for v in (data if isinstance(data, (list, tuple)) else [data])]
    with open(fname, 'wb') as fh:
        fh.write(params_io.dumps(arrs))


def load(fname):
    """mx.nd.load (lib/train_utils/utils.py:56): MXNet binary files of every record version; `.npz` containers written
    by earlier builds of this package are still accepted."""
    from . import params_io
    with open(fname, 'rb') as fh:
        buf = fh.read()
    if params_io.is_mxnet_file(buf[:8]):
        d = params_io.loads(buf)
        if isinstance(d, dict):
            return {k: NDArray(v) for k, v in d.items()}
        return [NDArray(v) for v in d]
    import io
    with np.load(io.BytesIO(buf), allow_pickle=False) as z:
        return {k: NDArray(z[k]) for k in z.files}
