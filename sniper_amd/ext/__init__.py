"""Drop-in replacements for the reference's compiled extension modules (SURVEY.md section 8(b),
"Extension-module contract"): same module names, function names, argument meaning and return types
as lib/chips/chips.pyx, lib/bbox/bbox.pyx, lib/nms/cpu_nms.pyx, lib/nms/gpu_nms.pyx -- computed by the
HIP kernels of libsniper_hip.so.  `install()` registers them in sys.modules under the names the
reference imports (`chips`, `bbox`, `cpu_nms`, `gpu_nms`), makes `from multiprocessing import Pool` safe for work items that
call them (ext/pool.py: the reference forks its pools; a forked child cannot use the parent's HIP context) and arranges that a
data-parallel rank assembles only its own slice of the reference iterator's global batch (ext/rank_slice.py)."""
import sys


def install():
    from . import bbox, chips, cpu_nms, gpu_nms, pool, rank_slice

    for name, mod in (("chips", chips), ("bbox", bbox), ("cpu_nms", cpu_nms), ("gpu_nms", gpu_nms)):
        sys.modules.setdefault(name, mod)
    pool.install()
    rank_slice.install_import_hook()
