"""Drop-in replacements for the reference's compiled extension modules (SURVEY.md section 8(b),
"Extension-module contract"): same module names, function names, argument meaning and return types
as lib/chips/chips.pyx, lib/bbox/bbox.pyx, lib/nms/cpu_nms.pyx, lib/nms/gpu_nms.pyx -- computed by the
HIP kernels of libsniper_hip.so.  `install()` registers them in sys.modules under the names the
reference imports (`chips`, `bbox`, `cpu_nms`, `gpu_nms`)."""
import sys


def install():
    from . import bbox, chips, cpu_nms, gpu_nms

    for name, mod in (("chips", chips), ("bbox", bbox), ("cpu_nms", cpu_nms), ("gpu_nms", gpu_nms)):
        sys.modules.setdefault(name, mod)
