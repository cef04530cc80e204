"""ctypes binding of libsniper_hip.so -- the only way Python reaches the HIP kernels.

The signatures are parsed from ``include/sniper_hip.h`` so that the header IS the binding: every
prototype declared there must be exported by the library (checked at load time), and argument
types come from the C declaration.  There is no CPU fallback: if the library is missing or does not
export a declared symbol, importing fails loudly.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(_HERE, "..", "include", "sniper_hip.h")
# SNIPER_HIP_LIB: another build of the same library (A/B runs of whole programs: tools/ab.sh)
LIB_PATH = os.environ.get("SNIPER_HIP_LIB") or os.path.join(_HERE, "lib", "libsniper_hip.so")

_SCALARS = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
    "size_t": ctypes.c_size_t, "uint64_t": ctypes.c_uint64, "int32_t": ctypes.c_int32, "sn_stream_t": ctypes.c_void_p,
    "unsigned": ctypes.c_uint,
}


def _ctype(decl):
    decl = decl.replace("const", " ").strip()
    if "*" in decl:
        return ctypes.c_char_p if decl.replace(" ", "").startswith("char*") else ctypes.c_void_p
    base = decl.split()[0] if decl.split() else decl
    if base not in _SCALARS:
        raise ValueError("unknown C type in header: %r" % decl)
    return _SCALARS[base]


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes], [argnames])} for every prototype in the header."""
    with open(path) as fh:
        src = fh.read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"#[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(sn_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret:
            continue
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                argtypes.append(_ctype(mm.group(1)))
                argnames.append(mm.group(2))
        protos[name] = (_ctype(ret), argtypes, argnames)
    return protos


class SniperHipError(RuntimeError):
    pass


class _Lib(object):
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError("libsniper_hip.so not built (%s); run `python -m sniper_amd.build` -- there is no "
                              "CPU fallback" % LIB_PATH)
        # torch bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1 (same sonames as /opt/rocm's).  Whichever is
        # mapped first serves the whole process; two different runtimes in one process leave the second without a
        # device ("no ROCm-capable device is detected").  torch owns device memory and streams here, so its runtime
        # must be the one: load it before the kernel library resolves its DT_NEEDED entries.
        import torch  # noqa: F401
        self._dll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        missing = []
        for name, (ret, argtypes, _) in self.protos.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.restype = ret
            fn.argtypes = argtypes
        if missing:
            raise ImportError("libsniper_hip.so does not export symbols declared in include/sniper_hip.h: %s" % missing)

    def raw(self, name):
        return getattr(self._dll, name)

    def call(self, name, *args):
        """Call an int-status entry point; raises SniperHipError with sn_last_error() on failure."""
        fn = getattr(self._dll, name)
        rc = fn(*args)
        if rc != 0:
            raise SniperHipError("%s failed (%d): %s" % (name, rc, self._dll.sn_last_error().decode()))
        return rc


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB
