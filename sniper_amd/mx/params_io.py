"""Reader / writer for MXNet's binary NDArray-list container (`*.params`, `mx.nd.save` / `mx.nd.load`).

SURVEY.md 8(f) item 4 (checkpoint interop): `load_checkpoint` (lib/train_utils/utils.py:44-66) calls
`mx.nd.load('%s-%04d.params')` on the ImageNet-pretrained backbone (main_train.py:98) and on the released SNIPER /
AutoFocus detectors (scripts/download_sniper_autofocus_detectors.sh), and `mx.model.save_checkpoint`
(resnet_mx_101_e2e.py:14) writes the same container.  The format belongs to the un-vendored MXNet runtime
(`NDArray::Save/Load`, src/ndarray/ndarray.cc of Apache MXNet 1.x) -- restated here from its published layout; the
reference tree ships no `.params` file, so this is **parity unpinned** against real files (tests pin the byte layout
against hand-assembled streams of every version).

All integers little-endian:

    file    := u64 0x112 | u64 0 | u64 n_arrays | array * n_arrays | u64 n_names | (u64 len | bytes) * n_names
    array   := V2/V3: u32 magic (0xF993FAC9 / 0xF993FACA) | i32 storage_type (0 = dense) | shape | rest
               V1   : u32 magic 0xF993FAC8 | shape | rest
               legacy (MXNet <= 0.8, e.g. the 2016 model-zoo ResNets): u32 ndim | u32 dims[ndim] | rest
    shape   := u32 ndim | i64 dims[ndim]            (V3: i32 ndim, -1 = none)
    rest    := (nothing if the shape is empty) | i32 dev_type | i32 dev_id | i32 type_flag | raw data, C order
    type_flag: 0 f32, 1 f64, 2 f16, 3 u8, 4 i32, 5 i8, 6 i64

n_names is either 0 (a list was saved) or n_arrays (a dict).  The writer emits V2 dense arrays on cpu(0).
"""
import struct

import numpy as np

LIST_MAGIC = 0x112
V1_MAGIC, V2_MAGIC, V3_MAGIC = 0xF993FAC8, 0xF993FAC9, 0xF993FACA
_DTYPES = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64}
_FLAGS = {np.dtype(v): k for k, v in _DTYPES.items()}


class _Reader(object):
    def __init__(self, buf):
        self.buf, self.pos = memoryview(buf), 0

    def take(self, fmt):
        n = struct.calcsize(fmt)
        if self.pos + n > len(self.buf):
            raise ValueError('truncated MXNet NDArray file (wanted %d bytes at offset %d of %d)' % (n, self.pos, len(self.buf)))
        v = struct.unpack_from(fmt, self.buf, self.pos)
        self.pos += n
        return v if len(v) > 1 else v[0]

    def raw(self, n):
        if self.pos + n > len(self.buf):
            raise ValueError('truncated MXNet NDArray file (wanted %d data bytes at offset %d of %d)' % (n, self.pos, len(self.buf)))
        v = self.buf[self.pos:self.pos + n]
        self.pos += n
        return v


def _read_array(r):
    magic = r.take('<I')
    if magic in (V2_MAGIC, V3_MAGIC):
        stype = r.take('<i')
        if stype != 0:
            raise NotImplementedError('sparse NDArray (storage type %d) in a .params file' % stype)
        ndim = r.take('<i' if magic == V3_MAGIC else '<I')
        if ndim < 0:
            return None
        shape = tuple(r.take('<%dq' % ndim)) if ndim > 1 else ((r.take('<q'),) if ndim == 1 else ())
        if magic == V2_MAGIC and ndim == 0:
            return None
    elif magic == V1_MAGIC:
        ndim = r.take('<I')
        shape = tuple(r.take('<%dq' % ndim)) if ndim > 1 else ((r.take('<q'),) if ndim == 1 else ())
        if ndim == 0:
            return None
    else:                                   # legacy: the word just read is ndim, dims are u32
        ndim = magic
        if ndim > 32:
            raise ValueError('not an MXNet NDArray record (leading word 0x%08x)' % magic)
        shape = tuple(r.take('<%dI' % ndim)) if ndim > 1 else ((r.take('<I'),) if ndim == 1 else ())
        if ndim == 0:
            return None
    r.take('<ii')                           # context (dev_type, dev_id): everything is loaded to the host
    flag = r.take('<i')
    if flag not in _DTYPES:
        raise ValueError('unknown MXNet type flag %d' % flag)
    dt = np.dtype(_DTYPES[flag]).newbyteorder('<')
    count = int(np.prod(shape, dtype=np.int64)) if shape else 1
    data = np.frombuffer(r.raw(count * dt.itemsize), dtype=dt, count=count).reshape(shape)
    return np.array(data, dtype=_DTYPES[flag])     # native-endian, owned copy


def loads(buf):
    """bytes -> dict name -> numpy array (dict files) or list of numpy arrays (list files)."""
    r = _Reader(buf)
    header, _reserved = r.take('<QQ')
    if header != LIST_MAGIC:
        raise ValueError('not an MXNet NDArray list file (header 0x%x)' % header)
    n = r.take('<Q')
    arrays = [_read_array(r) for _ in range(n)]
    k = r.take('<Q')
    names = []
    for _ in range(k):
        ln = r.take('<Q')
        names.append(bytes(r.raw(ln)).decode('utf-8'))
    if k == 0:
        return arrays
    if k != n:
        raise ValueError('MXNet NDArray file holds %d arrays but %d names' % (n, k))
    return dict(zip(names, arrays))


def dumps(data):
    """dict name -> array, or list of arrays -> bytes (V2 records, dense, cpu(0))."""
    if isinstance(data, dict):
        names, arrays = list(data.keys()), list(data.values())
    else:
        names, arrays = [], list(data)
    out = [struct.pack('<QQQ', LIST_MAGIC, 0, len(arrays))]
    for a in arrays:
        a = np.asarray(a)
        if a.dtype not in _FLAGS:
            raise TypeError('dtype %s has no MXNet type flag' % a.dtype)
        if a.ndim == 0:
            raise ValueError('MXNet (pre numpy-shape semantics) cannot store a 0-d array; reshape to (1,)')
        a = np.ascontiguousarray(a)
        out.append(struct.pack('<Ii', V2_MAGIC, 0))
        out.append(struct.pack('<I%dq' % a.ndim, a.ndim, *a.shape))
        out.append(struct.pack('<iii', 1, 0, _FLAGS[a.dtype]))
        out.append(a.astype(a.dtype.newbyteorder('<'), copy=False).tobytes())
    out.append(struct.pack('<Q', len(names)))
    for s in names:
        b = s.encode('utf-8')
        out.append(struct.pack('<Q', len(b)))
        out.append(b)
    return b''.join(out)


def is_mxnet_file(head):
    return len(head) >= 8 and struct.unpack_from('<Q', head, 0)[0] == LIST_MAGIC
