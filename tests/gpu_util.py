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


def assert_close(got, want, rtol, atol, what=''):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
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
