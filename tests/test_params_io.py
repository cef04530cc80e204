"""CPU: MXNet binary `.params` container (sniper_amd/mx/params_io.py).  The byte streams below are
assembled by hand from the published record layouts of NDArray::Save / LegacyLoad (Apache MXNet src/ndarray/ndarray.cc), one
per record version; the reference tree ships no .params file to pin against."""
import os
import struct

import numpy as np
import pytest

import sniper_amd.mx as mx
from sniper_amd.mx import params_io


def _file(records, names):
    out = struct.pack('<QQQ', 0x112, 0, len(records)) + b''.join(records) + struct.pack('<Q', len(names))
    for n in names:
        out += struct.pack('<Q', len(n)) + n.encode()
    return out


def _tail(a, flag, dev=(2, 3)):
    return struct.pack('<iii', dev[0], dev[1], flag) + a.tobytes()


def test_reads_every_record_version():
    w = np.arange(24, dtype=np.float32).reshape(2, 3, 4) / 7
    g = np.array([1.5, -2.25], np.float16)
    idx = np.array([[1, 2], [3, 4]], np.int64)
    legacy = struct.pack('<I3I', 3, 2, 3, 4) + _tail(w, 0)                          # pre-magic: u32 ndim, u32 dims
    v1 = struct.pack('<II1q', 0xF993FAC8, 1, 2) + _tail(g, 2)                       # V1: magic, u32 ndim, i64 dims
    v2 = struct.pack('<IiI2q', 0xF993FAC9, 0, 2, 2, 2) + _tail(idx, 6)              # V2: magic, stype, shape
    v3 = struct.pack('<Iii1q', 0xF993FACA, 0, 1, 2) + _tail(g, 2)                   # V3: i32 ndim
    d = params_io.loads(_file([legacy, v1, v2, v3], ['arg:conv0_weight', 'aux:bn0_moving_var', 'arg:idx', 'arg:g3']))
    assert list(d) == ['arg:conv0_weight', 'aux:bn0_moving_var', 'arg:idx', 'arg:g3']
    assert d['arg:conv0_weight'].dtype == np.float32 and np.array_equal(d['arg:conv0_weight'], w)
    assert d['aux:bn0_moving_var'].dtype == np.float16 and np.array_equal(d['aux:bn0_moving_var'], g)
    assert d['arg:idx'].dtype == np.int64 and np.array_equal(d['arg:idx'], idx)
    assert np.array_equal(d['arg:g3'], g)
    # a list file (no names) and an empty ("none") V2 record
    lst = params_io.loads(_file([legacy, struct.pack('<IiI', 0xF993FAC9, 0, 0)], []))
    assert isinstance(lst, list) and np.array_equal(lst[0], w) and lst[1] is None


def test_writer_layout_and_round_trip():
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    blob = params_io.dumps({'arg:w': a})
    want = (struct.pack('<QQQ', 0x112, 0, 1) + struct.pack('<IiI2q', 0xF993FAC9, 0, 2, 2, 3) + struct.pack('<iii', 1, 0, 0) +
            a.tobytes() + struct.pack('<QQ', 1, 5) + b'arg:w')
    assert blob == want
    rs = np.random.RandomState(0)
    d = {'arg:a': rs.standard_normal((3, 1, 7, 7)).astype(np.float32), 'aux:b': rs.standard_normal(5).astype(np.float64),
         'arg:c': rs.standard_normal((4, 4)).astype(np.float16), 'arg:d': rs.randint(0, 255, (9,)).astype(np.uint8),
         'arg:e': rs.randint(-5, 5, (2, 2)).astype(np.int32), 'arg:f': rs.randint(-5, 5, (2,)).astype(np.int8)}
    back = params_io.loads(params_io.dumps(d))
    assert list(back) == list(d)
    for k in d:
        assert back[k].dtype == d[k].dtype and np.array_equal(back[k], d[k])
    with pytest.raises(ValueError):
        params_io.loads(blob[:-3])
    with pytest.raises(ValueError):
        params_io.loads(b'\x00' * 32)
    with pytest.raises(NotImplementedError):
        params_io.loads(_file([struct.pack('<Ii', 0xF993FAC9, 1)], []))            # row-sparse storage


def test_checkpoint_round_trip_through_the_mx_api(tmp_path):
    """mx.model.save_checkpoint -> `prefix-0003.params` -> the reference's load_checkpoint pattern (utils.py:56-66)."""
    rs = np.random.RandomState(1)
    arg = {'conv0_weight': mx.nd.array(rs.standard_normal((4, 3, 7, 7))), 'fc_bias': mx.nd.zeros((5,))}
    aux = {'bn0_moving_mean': mx.nd.array(rs.standard_normal(4))}
    prefix = str(tmp_path / 'e2e')
    mx.model.save_checkpoint(prefix, 3, None, arg, aux)
    with open(prefix + '-0003.params', 'rb') as fh:
        assert struct.unpack('<Q', fh.read(8))[0] == 0x112
    save_dict = mx.nd.load('%s-%04d.params' % (prefix, 3))
    assert sorted(save_dict) == ['arg:conv0_weight', 'arg:fc_bias', 'aux:bn0_moving_mean']
    _, a2, x2 = mx.model.load_checkpoint(prefix, 3)
    assert np.array_equal(a2['conv0_weight'].asnumpy(), arg['conv0_weight'].asnumpy())
    assert np.array_equal(x2['bn0_moving_mean'].asnumpy(), aux['bn0_moving_mean'].asnumpy())
    # containers written by earlier builds (.npz under the .params name) still load
    old = str(tmp_path / 'old-0001.params')
    with open(old, 'wb') as fh:
        np.savez(fh, **{'arg:w': np.ones((2, 2), np.float32)})
    assert np.array_equal(mx.nd.load(old)['arg:w'].asnumpy(), np.ones((2, 2)))
    # list form
    lf = str(tmp_path / 'list.nd')
    mx.nd.save(lf, [mx.nd.ones((2,)), mx.nd.zeros((3,))])
    got = mx.nd.load(lf)
    assert isinstance(got, list) and got[0].shape == (2,) and got[1].shape == (3,)
