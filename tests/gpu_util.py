"""Helpers shared by the -m gpu tests."""
import numpy as np
import torch


def dev():
    return torch.device('cuda', 0)


def to_nhwc_f16(x_nchw):
    """numpy (N,C,H,W) float -> device (N,H,W,C) fp16 contiguous."""
    return torch.from_numpy(np.ascontiguousarray(x_nchw.transpose(0, 2, 3, 1))).to(dev()).half().contiguous()


def from_nhwc(t):
    """device (N,H,W,C) -> numpy (N,C,H,W) float32."""
    return t.float().cpu().numpy().transpose(0, 3, 1, 2)


def w_to_otI(w_oihw):
    """(O,I,KH,KW) -> [O][KH*KW][I] float32 numpy."""
    O, I, KH, KW = w_oihw.shape
    return np.ascontiguousarray(w_oihw.transpose(0, 2, 3, 1).reshape(O, KH * KW, I))


def f16r(x):
    """round-trip through fp16 (what the device actually multiplies)."""
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def error_quantiles(got, want):
    """Elementwise error of got against want, two denominators: |want| of the element itself -- over the elements of at least
    1e-3 of the tensor's largest magnitude (`rel_elementwise`; below that a relative error is rounding noise over nothing) and
    over the SIGNIFICANT elements, at least 5 % of the largest magnitude (`rel_significant`: an fp16-stored tensor carries an
    absolute error of ~1e-4 of its scale in every element, so the relative error of an element grows as the element shrinks;
    for the elements that matter it must stay under the tolerance) -- and the tensor's largest magnitude (`err_over_max`, the
    max-norm the tolerances of assert_close are stated in)."""
    got = np.asarray(got, np.float64).ravel()
    want = np.asarray(want, np.float64).ravel()
    err = np.abs(got - want)
    top = max(float(np.abs(want).max()), 1e-30)
    big = np.abs(want) >= 1e-3 * top
    rel = err[big] / np.abs(want[big]) if big.any() else np.zeros(1)
    sig = np.abs(want) >= 5e-2 * top
    rel_sig = err[sig] / np.abs(want[sig]) if sig.any() else np.zeros(1)
    qs = (0.5, 0.9, 0.99, 0.999, 1.0)
    return {'n': int(want.size), 'max_abs_want': top,
            'rel_elementwise': {('p%g' % (100 * q)): float(np.quantile(rel, q)) for q in qs},
            'rel_significant': {('p%g' % (100 * q)): float(np.quantile(rel_sig, q)) for q in qs},
            'err_over_max': {('p%g' % (100 * q)): float(np.quantile(err, q) / top) for q in qs},
            'rel_l2': float(np.sqrt((err ** 2).sum() / max((want ** 2).sum(), 1e-300)))}


def _log_quantiles(what, q):
    """one JSON line per comparison into gpurun_out/parity_quantiles.jsonl (merged back from the GPU box; summarised into
    profiles/ by tools/parity_report.py) -- the measured distribution behind every `<= 1e-2` claim"""
    import json
    import os
    try:
        root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
        os.makedirs(root, exist_ok=True)
        rec = {'what': str(what), 'test': os.environ.get('PYTEST_CURRENT_TEST', '').split(' ')[0]}
        rec.update(q)
        with open(os.path.join(root, 'parity_quantiles.jsonl'), 'a') as fh:
            fh.write(json.dumps(rec) + '\n')
    except OSError:
        pass


def assert_close(got, want, rtol, atol, what='', sig_rtol=None):
    """Two bounds.  (1) every element: |err| <= atol + rtol |want| (the call sites state atol as a share of the tensor's largest
    magnitude: a max-norm bound).  (2) north_star's "1e-2 rel" taken literally where a relative error means something: over the
    SIGNIFICANT elements (|want| >= 5 % of the largest magnitude) the 99.9th percentile of |err| / |want| stays under `sig_rtol`
    (default: 1e-2, or the call's own rtol where the test declares a looser one) -- an element at 5 % of the scale that is 20 % off
    passes (1) and fails (2)."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    q = None
    if got.size >= 64:
        q = error_quantiles(got, want)
        _log_quantiles(what, q)
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    bad = err > tol
    if bad.any():
        idx = np.argwhere(bad)
        first = tuple(idx[0])
        worst = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(
            "%s: %d/%d elements off (%.2f%%); first %s got %.6g want %.6g; worst %s got %.6g want %.6g; max|want| %.4g"
            % (what, bad.sum(), bad.size, 100.0 * bad.mean(), first, got[first], want[first], worst, got[worst],
               want[worst], np.abs(want).max()))
    if q is not None and np.isfinite(want).all():
        lim = sig_rtol if sig_rtol is not None else (rtol if rtol > 1e-2 else 1e-2)
        p999 = q['rel_significant']['p99.9']
        assert p999 <= lim, ("%s: relative error of the significant elements (>= 5%% of max|want| = %.4g): p99.9 %.3g > %.3g "
                             "(p50 %.3g, p99 %.3g, max %.3g)" % (what, q['max_abs_want'], p999, lim, q['rel_significant']['p50'],
                                                                 q['rel_significant']['p99'], q['rel_significant']['p100']))
