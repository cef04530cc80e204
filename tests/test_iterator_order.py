"""CPU: the epoch order of the three iterator mirrors (sniper_amd/iterators: `orientation_order` and the `reset` methods built on it)
against the reference's own `reset` bodies, run here from the lib2to3 translation oracle/build.py makes of lib/iterators
(oracle/_ref/py3; MNIteratorBase.py:59-82, MNIteratorTest.py:52-65, MNIteratorTestAutoFocus.py:96-134) on stub objects under the same
numpy seed.  Skipped where the translation is absent (a box without /root/reference)."""
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, 'oracle', '_ref', 'py3', 'lib')


def _ref_class(mod, cls):
    """The reference class's `reset`, compiled ALONE from the translated file (the module's import chain reaches mxnet, cv2 and the
    compiled extensions; `reset` itself is numpy only): a namespace object with `.reset(stub)`."""
    import ast
    path = os.path.join(REF, 'iterators', mod + '.py')
    if not os.path.isfile(path):
        pytest.skip('no translated reference (oracle/_ref/py3)')
    tree = ast.parse(open(path).read())
    klass = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    fn = next(n for n in klass.body if isinstance(n, ast.FunctionDef) and n.name == 'reset')
    ns = {'np': np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, 'exec'), ns)
    return types.SimpleNamespace(reset=ns['reset'])


def _roidb(rs, n, crops=False):
    out = []
    for _ in range(n):
        w, h = int(rs.randint(200, 700)), int(rs.randint(200, 700))
        r = {'width': w, 'height': h}
        if crops:
            k = int(rs.randint(0, 4))
            x1, y1 = rs.randint(0, 100, k), rs.randint(0, 100, k)
            r['inference_crops'] = np.stack([x1, y1, x1 + rs.randint(50, 300, k), y1 + rs.randint(50, 300, k)], 1).astype(np.float64)
        out.append(r)
    return out


@pytest.mark.parametrize('n,bs,single', [(37, 4, False), (37, 4, True), (16, 8, False), (5, 4, False), (64, 20, False)])
def test_training_order_equals_the_reference_reset(n, bs, single):
    from sniper_amd.iterators.MNIteratorBase import MNIteratorBase
    ref = _ref_class('MNIteratorBase', 'MNIteratorBase')
    roidb = _roidb(np.random.RandomState(n + bs), n)
    a = types.SimpleNamespace(roidb=roidb, batch_size=bs, single_size_change=single)
    b = types.SimpleNamespace(roidb=roidb, batch_size=bs, single_size_change=single)
    b._set_order = lambda order: MNIteratorBase._set_order(b, order)
    for seed in (0, 3):
        np.random.seed(seed)
        ref.reset(a)
        np.random.seed(seed)
        MNIteratorBase.reset(b)
        assert np.array_equal(a.inds, b.inds) and a.size == b.size and b.cur_i == 0
        assert np.random.randint(1 << 30) == (np.random.seed(seed), ref.reset(a), np.random.randint(1 << 30))[2]   # same number of draws


@pytest.mark.parametrize('n,bs', [(37, 4), (9, 8), (3, 4)])
def test_single_scale_test_order_equals_the_reference_reset(n, bs):
    from sniper_amd.iterators.MNIteratorBase import orientation_order
    ref = _ref_class('MNIteratorTest', 'MNIteratorTest')
    roidb = _roidb(np.random.RandomState(n), n)
    a = types.SimpleNamespace(roidb=roidb, batch_size=bs)
    ref.reset(a)
    ours = orientation_order([r['width'] for r in roidb], [r['height'] for r in roidb], bs, 'first')
    assert np.array_equal(a.inds, ours)


@pytest.mark.parametrize('n,bs', [(23, 4), (9, 8), (40, 2)])
def test_autofocus_order_and_chip_maps_equal_the_reference_reset(n, bs):
    from sniper_amd.iterators.MNIteratorBase import MNIteratorBase
    from sniper_amd.iterators.MNIteratorTestAutoFocus import MNIteratorTestAutoFocus
    ref = _ref_class('MNIteratorTestAutoFocus', 'MNIteratorTestAutoFocus')
    rs = np.random.RandomState(n)
    roidb_a, roidb_b = _roidb(rs, n, crops=True), None
    import copy
    roidb_b = copy.deepcopy(roidb_a)
    a = types.SimpleNamespace(roidb=roidb_a, batch_size=bs)
    b = types.SimpleNamespace(roidb=roidb_b, batch_size=bs)
    b._set_order = lambda order: MNIteratorBase._set_order(b, order)
    ref.reset(a)
    MNIteratorTestAutoFocus.reset(b)
    assert np.array_equal(a.inds, b.inds) and a.size == b.size
    assert a.crop2im == b.crop2im
    assert all(x['crop_mapping'] == y['crop_mapping'] for x, y in zip(roidb_a, roidb_b))
