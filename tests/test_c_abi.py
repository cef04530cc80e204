"""CPU: the C-ABI shared library loads without a GPU and exports every symbol include/sniper_hip.h declares; the
host-only entry points (sizes, plans, argument validation) behave.  No kernel is launched here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from sniper_amd import build as hipbuild
    hipbuild.build(verbose=False)          # no-op when the objects are current; hipcc cross-compiles without a GPU
    from sniper_amd._lib import lib as load
    return load()


def test_c_abi_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, 'include', 'sniper_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', ' ', hdr, flags=re.S)
    declared = set(re.findall(r'\b(sn_\w+)\s*\(', hdr))
    assert len(declared) >= 45
    assert declared == set(lib.protos), declared ^ set(lib.protos)
    dll = ctypes.CDLL(os.path.join(ROOT, 'sniper_amd', 'lib', 'libsniper_hip.so'))
    for name in declared:
        assert hasattr(dll, name), name
    # nothing but the declared C symbols is exported from the kernel library (-fvisibility=hidden)
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', os.path.join(ROOT, 'sniper_amd', 'lib', 'libsniper_hip.so')],
                         stdout=subprocess.PIPE, text=True).stdout
    exported = set(l.split()[-1] for l in out.splitlines() if ' T ' in l)
    assert declared <= exported
    assert not [e for e in exported - declared if e.startswith('sn_')]


def test_host_only_entry_points(lib):
    raw = lib.raw
    assert raw('sn_version')() >= 100
    # chips::cgenerate candidate count (cchips.cpp:62-108): 3 corner chips + grid + right column + bottom row
    assert raw('sn_chips_num_candidates')(512, 512, 512, 56) == 3
    nx = -(-(1920 - 512) // 56)
    ny = -(-(1440 - 512) // 56)
    assert raw('sn_chips_num_candidates')(1920, 1440, 512, 56) == 3 + nx * ny + nx + ny
    assert raw('sn_nms_workspace_bytes')(20, 6000) >= 20 * 6000 * ((6000 + 63) // 64) * 8
    assert raw('sn_bn_workspace_bytes')(81920, 512) > 0 and raw('sn_bn_workspace_bytes')(81920, 7) == 0
    # a 3x3 256->256 layer at 20x32x32 needs K-splits (36 tiles for 512 slots); the RPN conv does not
    assert raw('sn_conv_wgrad_workspace_bytes')(20, 32, 32, 256, 256, 256, 256, 3, 3, 1, 1, 1) > 0
    assert raw('sn_conv_wgrad_workspace_bytes')(20, 32, 32, 3072, 3072, 512, 512, 3, 3, 1, 1, 1) == 0
    assert raw('sn_dpsroi_bwd_workspace_bytes')(6000) >= 6000 * 16


def test_argument_errors_are_reported_before_any_launch(lib):
    from sniper_amd._lib import SniperHipError
    with pytest.raises(SniperHipError) as e:
        lib.call('sn_conv_fwd', None, None, None, None, None, 1, 8, 8, 8, 8, 8, 8, 8, 3, 3, 1, 1, 1, 0, 0, None)
    assert 'null pointer' in str(e.value)
    with pytest.raises(SniperHipError) as e:
        lib.call('sn_bn_stats', ctypes.c_void_p(16), 10, 12, 12, ctypes.c_void_p(16), None)
    assert 'multiple of 8' in str(e.value)
    with pytest.raises(SniperHipError) as e:
        lib.call('sn_dwconv_fwd', ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 8, 8, 16, 16, 16, 5, 5, 1, 2,
                 1, None)
    assert '3x3' in str(e.value)


def test_every_python_source_compiles():
    """Modules that only the -m gpu tests import must still be syntactically valid on the CPU tier."""
    bad = []
    files = [os.path.join(ROOT, f) for f in ('bench.py', '__graft_entry__.py')]
    for top in ('sniper_amd', 'oracle', 'tools', 'tests'):
        for dp, _, fs in os.walk(os.path.join(ROOT, top)):
            files += [os.path.join(dp, f) for f in fs if f.endswith('.py')]
    for f in files:
        try:
            compile(open(f).read(), f, 'exec')
        except SyntaxError as e:
            bad.append('%s: %s' % (f, e))
    assert len(files) > 40 and not bad, bad


def test_conv_tile_plan_divides_the_baseline_shapes_over_the_cus():
    """Host-side selection only (no launch): at BASELINE C2 (20 chips) the pipelined convolution takes 160-row tiles, so that
    20 480 stage-3 pixels are 128 row tiles (256 / 512 / 1024 workgroups for 256 / 512 / 1024 channels on 256 CUs) and 81 920
    stage-2 pixels 512 -- DESIGN.md section 8.4 (a silent change of this table costs 5 % of the step).  The row-tile count is
    what sn_conv_fwd_stats_blocks / sn_conv_dgrad_bn_blocks report (the statistics partials are per row tile)."""
    from sniper_amd import hip

    def fwd(N, H, W, Cin, Cout, K, stride=1, pad=0, res=0):
        return hip.query('sn_conv_fwd_stats_blocks', N, H, W, Cin, Cin, Cout, Cout, res, K, K, stride, pad, 1)

    def dgrad(N, H, W, Cin, Cout, K, pad=0):
        return hip.query('sn_conv_dgrad_bn_blocks', N, H, W, Cin, Cin, Cout, Cout, 0, K, K, 1, pad, 1)
    assert fwd(20, 32, 32, 1024, 256, 1) == 128            # stage-3 conv1
    assert fwd(20, 32, 32, 256, 256, 3, pad=1) == 128      # stage-3 conv2
    assert fwd(20, 32, 32, 256, 1024, 1, res=1024) == 128  # stage-3 conv3 + residual
    assert fwd(20, 64, 64, 128, 128, 3, pad=1) == 512      # stage-2 conv2
    assert fwd(20, 32, 32, 1024, 512, 1) == 128            # stage-4 conv1
    assert dgrad(20, 32, 32, 256, 256, 3, pad=1) == 128    # data gradient of stage-3 conv2, BatchNorm reduction fused
    assert fwd(20, 128, 128, 64, 256, 1) == 5120           # a single K-step: 64-row tiles (all epilogue)
    assert fwd(20, 128, 128, 64, 64, 1) == 0               # <= 64 output channels: register-staged kernel, no fused statistics


def test_numpy_arguments_are_host_pointers_only_where_the_header_says_so():
    """sniper_amd.hip.call hands a numpy array to the library as a HOST address: allowed for the *_host entry points and for
    parameters named h_*; a numpy array in a device-pointer slot is a TypeError, not a kernel reading host memory."""
    import numpy as np
    from sniper_amd import hip
    with pytest.raises(TypeError):
        hip.call('sn_bbox_decode', np.zeros((4, 5), np.float32), None, None, None, 1, 4, None)
    with pytest.raises(ValueError):            # host arrays must be contiguous
        hip.call('sn_focus_chips_host', np.zeros((8, 8), np.float32)[:, ::2], 8, 4, 1, 0.5, 1, 64.0, 128.0, 1.0,
                 np.zeros((4, 4), np.float64), 4, np.zeros(1, np.int32))
    out, n = np.zeros((4, 4), np.float64), np.zeros(1, np.int32)
    hip.call('sn_focus_chips_host', np.full((8, 8), 0.9, np.float32), 8, 8, 1, 0.5, 1, 128.0, 128.0, 1.0, out, 4, n)
    assert n[0] == 1 and out[0].tolist() == [0.0, 0.0, 128.0, 128.0]
