"""Run the reference's *own* Python data path in this container (TEST INFRASTRUCTURE ONLY).

The reference is Python 2 + MXNet + OpenCV; none of those exist here.  To execute it anyway
 we translate the needed files with ``lib2to3`` into a scratch
directory **outside the repository** (nothing from the reference is copied into the repo), apply
the two documented integer-division fixes, and import them over tiny stubs for ``mxnet.nd`` (numpy
pass-through), ``cv2`` and ``nms``.  The native pieces come from ``oracle/_ref`` (built by
``oracle/build.py`` from the reference sources where they lie).

Where ``/root/reference`` exists (the build container) it produces the golden vectors in ``tests/golden``
(``tests/golden/make_golden.py``) and pins the numpy/C restatements.  Where it does not (the GPU box), the same modules load from
the translated artefact ``oracle/_ref/py3`` and the compiled modules in ``oracle/_ref`` -- bench.py's ``cpu_baseline`` times the
reference's own workers that way (kind "reference").
"""
import ctypes
import importlib
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

from . import build

REF = build.REF
_STATE = {}


ARTEFACT_LIB = os.path.join(build.REF_OUT, "py3", "lib")      # lib2to3 output of build.build_reference_py3 (travels to the GPU box)


def available():
    """The reference's Python can be run here: from its checkout, or from the translated artefact + the compiled modules that
    oracle/build.py left under oracle/_ref/ (the GPU box has no checkout)."""
    return build.have_reference() or artefact_available()


def artefact_available():
    try:
        names = os.listdir(build.REF_OUT)
    except OSError:
        return False
    return (os.path.isfile(os.path.join(ARTEFACT_LIB, "data_utils", "data_workers.py")) and any(f.startswith("chips.") for f in names)
            and any(f.startswith("bbox.") for f in names))


def _stub_modules():
    mx = types.ModuleType("mxnet")
    nd = types.ModuleType("mxnet.nd")

    def _array(x, dtype=None, ctx=None):
        return np.array(x, dtype=dtype)

    nd.array = _array
    nd.zeros = lambda shape, ctx=None, dtype="float32": np.zeros(shape, dtype=dtype)
    nd.ones = lambda shape, ctx=None, dtype="float32": np.ones(shape, dtype=dtype)
    mx.nd = nd
    mx.cpu = lambda i=0: None
    mx.gpu = lambda i=0: None
    io = types.ModuleType("mxnet.io")

    class DataIter(object):
        def __init__(self, batch_size=0):
            self.batch_size = batch_size

        def getpad(self):
            return 0

        def getindex(self):
            return None

    class DataBatch(object):
        def __init__(self, data, label=None, pad=None, index=None, provide_data=None, provide_label=None):
            self.data, self.label, self.pad, self.index = data, label, pad, index
            self.provide_data, self.provide_label = provide_data, provide_label

    io.DataIter, io.DataBatch = DataIter, DataBatch
    mx.io = io
    cv2 = types.ModuleType("cv2")
    nms_pkg = types.ModuleType("nms")
    nms_mod = types.ModuleType("nms.nms")
    nms_mod.nms_wrapper = lambda *a, **k: None
    nms_pkg.nms = nms_mod
    return {"mxnet": mx, "mxnet.nd": nd, "mxnet.io": io, "cv2": cv2, "nms": nms_pkg, "nms.nms": nms_mod}


def load():
    """Returns a namespace with the reference's modules: chips (native), bbox (native),
    chip_generator, data_workers, bbox_transform, generate_anchor."""
    if "ns" in _STATE:
        return _STATE["ns"]
    if not available():
        raise RuntimeError("neither the reference checkout (%s) nor its translated artefact (%s) is present" % (REF, ARTEFACT_LIB))
    # py2-era numpy aliases used by the reference (bbox.pyx:14, data_workers.py)
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "int"):
        np.int = int
    if build.have_reference():
        build.build_reference()
        tmp = tempfile.mkdtemp(prefix="sniper_ref_py3_")
        _STATE["tmp"] = tmp
        lib3 = os.path.join(tmp, "lib3")
        os.makedirs(lib3)
        for sub in ("bbox", "chips", "data_utils"):
            os.makedirs(os.path.join(lib3, sub))
            src = os.path.join(REF, "lib", sub)
            for f in os.listdir(src):
                if f.endswith(".py"):
                    shutil.copy(os.path.join(src, f), os.path.join(lib3, sub, f))
        subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", lib3], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)

        def _sed(path, old, new):
            with open(path) as fh:
                s = fh.read()
            assert old in s, (path, old)
            with open(path, "w") as fh:
                fh.write(s.replace(old, new))

        dw = os.path.join(lib3, "data_utils", "data_workers.py")
        _sed(dw, "chip_size / cfg.network.RPN_FEAT_STRIDE", "chip_size // cfg.network.RPN_FEAT_STRIDE")
    else:
        lib3 = ARTEFACT_LIB          # the same translation (+ the same division fix), made where the checkout exists
    # py2-only pickling shim (copy_reg / im_func) is dead weight for single-process use
    for name, mod in _stub_modules().items():
        sys.modules.setdefault(name, mod)
    # native modules under their reference import names
    sys.path.insert(0, build.REF_OUT)
    sys.path.insert(0, lib3)
    ns = types.SimpleNamespace()
    # `chips` is both a package dir (lib3/chips) and the native module name: load the native one
    # explicitly and place it where chip_generator's `import chips` finds it.
    import importlib.util

    ext = [f for f in os.listdir(build.REF_OUT) if f.startswith("chips.")][0]
    # the module's init symbol is PyInit_chips, so it must be loaded under the name "chips"
    spec = importlib.util.spec_from_file_location("chips", os.path.join(build.REF_OUT, ext))
    native_chips = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(native_chips)
    ns.chips = native_chips
    extb = [f for f in os.listdir(build.REF_OUT) if f.startswith("bbox.")][0]
    specb = importlib.util.spec_from_file_location("bbox", os.path.join(build.REF_OUT, extb))
    native_bbox = importlib.util.module_from_spec(specb)
    specb.loader.exec_module(native_bbox)
    ns.bbox = native_bbox
    # python packages: lib3/bbox/bbox_transform.py does `from bbox import ...` (native),
    # lib3/chips/chip_generator.py does `import chips` (native) and `from bbox.bbox_transform import`
    bbox_pkg = types.ModuleType("bbox")
    bbox_pkg.__path__ = [os.path.join(lib3, "bbox")]
    bbox_pkg.bbox_overlaps_cython = native_bbox.bbox_overlaps_cython
    bbox_pkg.ignore_overlaps_cython = native_bbox.ignore_overlaps_cython
    sys.modules["bbox"] = bbox_pkg
    chips_pkg = types.ModuleType("chips")
    chips_pkg.__path__ = [os.path.join(lib3, "chips")]
    chips_pkg.generate = native_chips.generate
    sys.modules["chips"] = chips_pkg
    ns.bbox_transform = importlib.import_module("bbox.bbox_transform")
    ns.chip_generator = importlib.import_module("chips.chip_generator")
    ns.generate_anchor = importlib.import_module("data_utils.generate_anchor")
    ns.data_workers = importlib.import_module("data_utils.data_workers")
    ns.libc = ctypes.CDLL(None)
    _STATE["ns"] = ns
    return ns


def srand(seed):
    """Seed libc rand(), which drives std::random_shuffle inside the reference's cchips.cpp:117."""
    load().libc.srand(ctypes.c_uint(int(seed)))
