"""Build libsniper_hip.so (the C-ABI kernel library) in-tree with hipcc for gfx950.

    python -m sniper_amd.build [--force]

hipcc cross-compiles without a GPU.  Every translation unit is compiled separately (cached by
mtime) and linked into sniper_amd/lib/libsniper_hip.so, which travels to the GPU box with the
repository snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.environ.get("SNIPER_BUILD_CSRC") or os.path.join(HERE, "csrc")   # (experiments: a patched COPY of csrc, tools/probes/conv_px_build.sh)
OUT_DIR = os.path.join(HERE, "lib")
# A/B builds of the whole library (tools/ab.sh loads them through SNIPER_HIP_LIB): SNIPER_BUILD_SUFFIX=_x SNIPER_BUILD_DEFS="-DSN_X=1"
# compiles into lib/obj_x and links lib/libsniper_hip_x.so; without them this is the one library the package loads.
_SUFFIX = os.environ.get("SNIPER_BUILD_SUFFIX", "")
_DEFS = os.environ.get("SNIPER_BUILD_DEFS", "").split()
OBJ_DIR = os.path.join(OUT_DIR, "obj" + _SUFFIX)
LIB = os.path.join(OUT_DIR, "libsniper_hip%s.so" % _SUFFIX)
ARCH = "gfx950"

# per-file extra flags: the bit-exact integer/box kernels must not contract a*b+c into FMA
EXTRA = {
    "data_path.hip": ["-ffp-contract=off"],
    "nms.hip": ["-ffp-contract=off"],
    "proposal.hip": ["-ffp-contract=off"],
    "infer.hip": ["-ffp-contract=off"],
    "host_inference.cpp": ["-ffp-contract=off"],
    "mask.hip": ["-ffp-contract=off"],
}
BASE = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
        "-Rpass-analysis=kernel-resource-usage"]
# the compiler's per-kernel resource remarks (VGPRs, scratch, occupancy, LDS) are kept beside the object: tests/test_kernel_resources.py
# reads them -- an accumulator tile that silently moved to scratch memory shows up there and nowhere else


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _is_remark_context(line):
    """the source excerpt clang prints under a remark: `  NN | code` and the caret line"""
    t = line.lstrip()
    return t.startswith("|") or (t[:1].isdigit() and "|" in t[:12])


def sources():
    """Kernel sources (.hip) and host-only sources (.cpp) of the one library."""
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [src] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    deps.append(os.path.join(HERE, "..", "include", "sniper_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    objs = []
    for f in sources():
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJ_DIR, os.path.splitext(f)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, src):
            jobs.append([hipcc] + BASE + _DEFS + EXTRA.get(f, []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[sniper_amd.build]", os.path.basename(cmd[-3]), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return cmd, r

    failed = False
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for cmd, r in ex.map(run, jobs):
            if r.returncode != 0:
                failed = True
                sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + "\n")
            else:
                with open(cmd[-1][:-2] + ".remarks", "w") as fh:
                    fh.write(r.stdout)
                rest = "\n".join(ln for ln in r.stdout.splitlines() if "remark:" not in ln and not _is_remark_context(ln))
                if verbose and rest.strip():
                    print(rest)
    if failed:
        raise RuntimeError("hipcc failed")
    if jobs or not os.path.exists(LIB):
        subprocess.check_call([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
