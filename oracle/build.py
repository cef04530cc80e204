"""Build the oracle (TEST INFRASTRUCTURE ONLY -- never imported by sniper_amd/).

Two artefacts:

* ``oracle/libsniper_oracle.so`` -- our plain-C restatement (``sniper_oracle.c``).
* ``oracle/_ref/{chips,bbox,cpu_nms}*.so`` -- the reference's *own* native code (``lib/chips/cchips.cpp``
  + ``lib/chips/chips.pyx``, ``lib/bbox/bbox.pyx``, ``lib/nms/cpu_nms.pyx``) compiled from the sources where they lie
  under ``/root/reference``.  ``cpu_nms.pyx`` does not compile under Cython 3 as written (``np.int_t`` /
  ``np.float`` are gone): it is read from the reference, four type spellings on lines 112-124 are
  replaced IN A SCRATCH COPY (see ``_CPU_NMS_PATCH``; ``cpu_soft_nms``, lines 17-110, is compiled untouched), and
  only the shared object is kept.  No reference source is copied into this repository: Cython's
  generated C/C++ goes to a scratch directory outside the repo, only the shared objects land in
  ``oracle/_ref/`` (git-ignored, but shipped to the GPU box).  Built only when the reference
  checkout is present; the GPU box uses the prebuilt files.

The reference's own build (``lib/chips/setup.py``, ``scripts/compile.sh``) is not used: it is
distutils/py2 only and ``std::random_shuffle`` needs ``-std=c++14``.
"""
import os
import subprocess
import sys
import sysconfig
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SNIPER_REFERENCE", "/root/reference")
REF_OUT = os.path.join(HERE, "_ref")


def _run(cmd):
    subprocess.check_call(cmd)


def _newer(target, *sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def build_restatement(force=False):
    src = os.path.join(HERE, "sniper_oracle.c")
    out = os.path.join(HERE, "libsniper_oracle.so")
    if force or not _newer(out, src):
        _run(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-o", out, src, "-lm"])
    return out


def have_reference():
    return os.path.isfile(os.path.join(REF, "lib", "chips", "cchips.cpp"))


def build_reference(force=False):
    """Compile the reference's chips/bbox extension modules into oracle/_ref/."""
    if not have_reference():
        return None
    import numpy

    os.makedirs(REF_OUT, exist_ok=True)
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    py_inc = sysconfig.get_paths()["include"]
    np_inc = numpy.get_include()
    out_chips = os.path.join(REF_OUT, "chips" + ext)
    out_bbox = os.path.join(REF_OUT, "bbox" + ext)
    if not force and os.path.exists(out_chips) and os.path.exists(out_bbox):
        return REF_OUT
    with tempfile.TemporaryDirectory(prefix="sniper_ref_build_") as tmp:
        chips_dir = os.path.join(REF, "lib", "chips")
        gen_cpp = os.path.join(tmp, "chips.cpp")
        _run([sys.executable, "-m", "cython", "-2", "--cplus", "-I", chips_dir, "-o", gen_cpp,
              os.path.join(chips_dir, "chips.pyx")])
        _run(["g++", "-O2", "-std=c++14", "-shared", "-fPIC", "-w", "-I", py_inc, "-I", np_inc, "-I", chips_dir,
              gen_cpp, os.path.join(chips_dir, "cchips.cpp"), "-o", out_chips])
        bbox_dir = os.path.join(REF, "lib", "bbox")
        gen_c = os.path.join(tmp, "bbox.c")
        _run([sys.executable, "-m", "cython", "-2", "-o", gen_c, os.path.join(bbox_dir, "bbox.pyx")])
        _run(["gcc", "-O2", "-shared", "-fPIC", "-w", "-I", py_inc, "-I", np_inc, gen_c, "-o", out_bbox])
    return REF_OUT


# (line, old, new): the hard-NMS function's buffer types.  `order` is produced by `.astype('i')` (int32), so its buffer
# is declared int32 (the original `np.int_t` = C long would refuse that array on LP64); `suppressed` keeps a
# pointer-sized integer.  Nothing inside cpu_soft_nms (lines 17-110) is touched.
_CPU_NMS_PATCH = ((112, 'np.float thresh', 'float thresh'), (120, 'np.int_t', 'np.int32_t'), (123, 'np.int_t', 'np.intp_t'),
                  (124, 'np.int)', 'np.intp)'))


def build_reference_cpu_nms(force=False):
    """oracle/_ref/cpu_nms*.so from lib/nms/cpu_nms.pyx (cpu_soft_nms verbatim, cpu_nms with _CPU_NMS_PATCH)."""
    out = os.path.join(REF_OUT, "cpu_nms" + sysconfig.get_config_var("EXT_SUFFIX"))
    if not force and os.path.exists(out):
        return out                      # (the GPU box has the prebuilt module and no reference checkout)
    if not have_reference():
        return None
    import numpy

    os.makedirs(REF_OUT, exist_ok=True)
    with open(os.path.join(REF, "lib", "nms", "cpu_nms.pyx")) as fh:
        lines = fh.read().split("\n")
    for ln, old, new in _CPU_NMS_PATCH:
        assert old in lines[ln - 1], (ln, lines[ln - 1])
        lines[ln - 1] = lines[ln - 1].replace(old, new)
    with tempfile.TemporaryDirectory(prefix="sniper_ref_build_") as tmp:
        pyx = os.path.join(tmp, "cpu_nms.pyx")
        with open(pyx, "w") as fh:
            fh.write("\n".join(lines))
        gen_c = os.path.join(tmp, "cpu_nms.c")
        _run([sys.executable, "-m", "cython", "-2", "-o", gen_c, pyx])
        _run(["gcc", "-O2", "-shared", "-fPIC", "-w", "-I", sysconfig.get_paths()["include"], "-I", numpy.get_include(), gen_c,
              "-o", out])
    return out


# (file, old, new): the only edits made to the lib2to3 output -- py2 integer divisions that py3 turns into floats
_PY3_FIXES = (("lib/data_utils/data_workers.py", "chip_size / cfg.network.RPN_FEAT_STRIDE", "chip_size // cfg.network.RPN_FEAT_STRIDE"),
              ("lib/iterators/MNIteratorBase.py", "self.n_per_gpu = batch_size / nGPUs", "self.n_per_gpu = batch_size // nGPUs"),
              ("lib/iterators/MNIteratorE2E.py", "self.crop_size[1] / self.cfg.network.RPN_FEAT_STRIDE", "self.crop_size[1] // self.cfg.network.RPN_FEAT_STRIDE"),
              ("lib/iterators/MNIteratorE2E.py", "self.crop_size[0] / self.cfg.network.RPN_FEAT_STRIDE", "self.crop_size[0] // self.cfg.network.RPN_FEAT_STRIDE"))
_PY3_TREES = ("lib", "symbols", "configs")
_PY3_FILES = ("main_train.py", "main_test.py", "init.py")


def build_reference_py3(force=False):
    """oracle/_ref/py3/: the reference's OWN Python (main_train.py, main_test.py, init.py, lib/, symbols/, configs/) run
    through ``lib2to3`` -- a derived artefact like the shared objects above (git-ignored, shipped to the GPU box), so that
    the acceptance test (tests/acceptance_main_train.py: main_train.py's __main__ block executed over sniper_amd's mxnet
    shim on the GPU) can run where /root/reference does not exist.  Only .py / .yml files are translated / carried; no
    reference file enters the git history."""
    if not have_reference():
        return None
    import shutil

    out = os.path.join(REF_OUT, "py3")
    stamp = os.path.join(out, ".built")
    if not force and os.path.exists(stamp):
        return out
    if os.path.isdir(out):
        shutil.rmtree(out)
    os.makedirs(out)
    for f in _PY3_FILES:
        shutil.copy(os.path.join(REF, f), os.path.join(out, f))
    for tree in _PY3_TREES:
        for root, dirs, files in os.walk(os.path.join(REF, tree)):
            dirs[:] = [d for d in dirs if d not in ("pycocotools", "__pycache__")]
            rel = os.path.relpath(root, REF)
            os.makedirs(os.path.join(out, rel), exist_ok=True)
            for f in files:
                if f.endswith((".py", ".yml")):
                    shutil.copy(os.path.join(root, f), os.path.join(out, rel, f))
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # lib/inference.py is imported as the TOP-LEVEL module `inference` (main_test.py:14, with lib/ on sys.path); lib2to3's import
    # fixer, seeing lib/__init__.py, turned its imports of the sibling packages into relative ones -- put them back
    import re
    inf = os.path.join(out, "lib", "inference.py")
    with open(inf) as fh:
        src = fh.read()
    with open(inf, "w") as fh:
        fh.write(re.sub(r"^from \.(\w)", r"from \1", src, flags=re.M))
    for rel, old, new in _PY3_FIXES:
        path = os.path.join(out, rel)
        with open(path) as fh:
            src = fh.read()
        assert old in src, (rel, old)
        with open(path, "w") as fh:
            fh.write(src.replace(old, new))
    for root, dirs, files in os.walk(out):
        os.chmod(root, 0o755)
        for f in files:
            os.chmod(os.path.join(root, f), 0o644)
    with open(stamp, "w") as fh:
        fh.write("lib2to3 of %s\n" % REF)
    return out


def build_all(force=False):
    build_restatement(force)
    build_reference(force)
    build_reference_cpu_nms(force)
    build_reference_py3(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print("oracle built; reference modules:", sorted(os.listdir(REF_OUT)) if os.path.isdir(REF_OUT) else None)
