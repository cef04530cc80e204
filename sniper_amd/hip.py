"""Glue between torch device tensors and the C ABI: torch owns memory and streams (plumbing), the
kernels are ours.  `call("sn_xxx", tensor_or_scalar, ...)` passes `tensor.data_ptr()` for tensors,
None -> NULL, and appends nothing implicitly -- the stream is passed explicitly via `stream()`."""
import ctypes

import numpy as np

import torch

from ._lib import SniperHipError, lib  # noqa: F401


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("sniper_amd needs a HIP device (MI355X); no CPU fallback exists")
    return torch.device("cuda", torch.cuda.current_device())


_GET_DEVICE = hasattr(torch._C, '_cuda_getDevice')      # the raw getter: ~0.1 us, no Python-side device object
_DEVICE = None      # this process's GPU (one process per GPU); set by use_device(), else looked up per call


def use_device(index):
    """Record the device this process computes on (callers of torch.cuda.set_device): stream() then skips the lookup."""
    global _DEVICE
    _DEVICE = int(index)


def stream():
    """hipStream_t of torch's current stream (so our kernels order with torch's allocator/events).  The raw getter costs well
    under a microsecond; torch.cuda.current_stream() builds a Stream object (~8 us, x 1500 launches of an eager inference pass)."""
    # torch's current device is per THREAD (a pool worker of sniper_amd.ext.pool binds its own; a second Module on another
    # device changes it): the recorded index is only a shortcut while it still IS the current device
    idx = torch._C._cuda_getDevice() if _GET_DEVICE else torch.cuda.current_device()
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(idx))


def _conv(a):
    if isinstance(a, torch.Tensor):
        return ctypes.c_void_p(a.data_ptr())
    if isinstance(a, np.ndarray):        # a HOST array argument (the entry point reads it before returning)
        if not a.flags['C_CONTIGUOUS']:
            raise ValueError('host array arguments must be C-contiguous')
        return ctypes.c_void_p(a.ctypes.data)
    return a


def call(name, *args):
    # a numpy array is a HOST pointer: only where the header says so (an entry point named *_host, a parameter named h_*) --
    # anywhere else it would hand a kernel a host address
    if any(isinstance(a, np.ndarray) for a in args) and not name.endswith('_host'):
        names = lib().protos[name][2]
        for a, pname in zip(args, names):
            if isinstance(a, np.ndarray) and not pname.startswith('h_'):
                raise TypeError('%s: parameter %s is a device pointer, got a numpy array' % (name, pname))
    return lib().call(name, *[_conv(a) for a in args])


def query(name, *args):
    """For non-status entry points (workspace sizes, counts)."""
    return lib().raw(name)(*[_conv(a) for a in args])


class WgradDesc(ctypes.Structure):
    """sn_wgrad_desc (include/sniper_hip.h): one layer of a batched weight-gradient launch."""
    _fields_ = [("dy", ctypes.c_void_p), ("x", ctypes.c_void_p), ("dw", ctypes.c_void_p)] + \
               [(k, ctypes.c_int) for k in ("N", "H", "W", "Cin", "x_pix_stride", "Cout", "dy_pix_stride", "KH", "KW", "stride", "pad", "dil")]


def wgrad_table(problems):
    """[(dy, x, dw, N, H, W, Cin, x_ps, Cout, dy_ps, KH, KW, stride, pad, dil)] -> ctypes array of sn_wgrad_desc (tensors or None)."""
    arr = (WgradDesc * len(problems))()
    for d, pr in zip(arr, problems):
        d.dy, d.x, d.dw = [t.data_ptr() if isinstance(t, torch.Tensor) else None for t in pr[:3]]
        (d.N, d.H, d.W, d.Cin, d.x_pix_stride, d.Cout, d.dy_pix_stride, d.KH, d.KW, d.stride, d.pad, d.dil) = [int(v) for v in pr[3:]]
    return arr


def dev(x, dtype=None, device=None):
    """numpy / tensor -> contiguous device tensor."""
    device = device or require_gpu()
    t = torch.as_tensor(x)
    if dtype is not None:
        t = t.to(dtype)
    return t.to(device).contiguous()


class Workspace(object):
    """Grow-only device scratch buffer (no allocation inside the hot calls)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes):
        nbytes = int(nbytes)
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=require_gpu())
        return self.buf
