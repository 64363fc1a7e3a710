"""Builds the HIP shared library in-tree with hipcc for gfx950.

    python -m street_crafter_amd.build            # incremental
    python -m street_crafter_amd.build --force
    python -m street_crafter_amd.build --diag     # the DIAGNOSTIC build as well (tools/exp_*.py only)

Output: street_crafter_amd/lib/libstreet_crafter_hip.so (git-ignored, shipped to the GPU box by
gpurun).  hipcc cross-compiles without a GPU, so this also is the "does it build" check.

The diagnostic build (lib/libstreet_crafter_hip_diag.so, -DSC_DIAG, objects under build/diag/) is the same
sources with the `debug0..3` skip switches compiled in (csrc/sc_common.h).  Nothing in the package, the tests
or bench.py loads it; the measuring tools under tools/ ask for it with _lib.use_diagnostic_build().
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libstreet_crafter_hip.so")
LIB_DIAG = os.path.join(LIBDIR, "libstreet_crafter_hip_diag.so")
BINDING = os.path.join(LIBDIR, "_sc_fast.so")          # compiled Python <-> C-ABI binding layer (csrc/binding.cpp)
ARCH = "gfx950"

SOURCES = ["capi.hip", "projection.hip", "isect.hip", "isect_bin.hip", "radix_sort.hip", "sh.hip",
           "raster_fwd.hip", "raster_bwd.hip", "knn.hip", "fused_fwd.hip"]
COMMON_FLAGS = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
                "-fno-gpu-rdc"]


def abi_hash() -> str:
    """Digest of the C-ABI header: compiled into the library (sc_version) and into the binding layer (abi_version)."""
    import hashlib
    with open(os.path.join(HERE, "..", "include", "street_crafter_amd.h"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _deps(src):
    d = [os.path.join(CSRC, src)]
    d += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    d.append(os.path.join(HERE, "..", "include", "street_crafter_amd.h"))
    return d


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, force, verbose, diag=False):
    tag = os.environ.get("SC_DIAG_TAG", "") if diag else ""
    obj = os.path.join(OBJDIR, ("diag" + ("_" + tag if tag else "")) if diag else "", src.replace(".hip", ".o"))
    if not force and not _stale(obj, _deps(src)):
        return obj, False
    # (SC_EXP_DEFS: extra -D flags for the DIAGNOSTIC build only -- kernel experiments are A/B-ed as shipped library
    #  vs diagnostic library, tools/ab_lib.py; the shipped build never sees them)
    extra = os.environ.get("SC_EXP_DEFS", "").split() if diag else []
    cmd = [_hipcc(), *COMMON_FLAGS, f'-DSC_ABI_HASH="{abi_hash()}"', *(["-DSC_DIAG", *extra] if diag else []), "-c",
           os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip() and verbose:
        print(r.stderr, file=sys.stderr)
    return obj, True


def build(force=False, verbose=False, jobs=None, diag=False):
    """Builds the shipped library (diag=False) or the diagnostic one (diag=True); returns its path."""
    os.makedirs(LIBDIR, exist_ok=True)
    tag = os.environ.get("SC_DIAG_TAG", "") if diag else ""       # several experiment builds side by side (tools/ab_lib.py)
    os.makedirs(os.path.join(OBJDIR, "diag" + ("_" + tag if tag else "")) if diag else OBJDIR, exist_ok=True)
    lib = (LIB_DIAG.replace("_diag.so", f"_diag_{tag}.so") if tag else LIB_DIAG) if diag else LIB
    jobs = jobs or min(len(SOURCES), max(1, (os.cpu_count() or 2) - 1))
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        results = list(ex.map(lambda s: _compile(s, force, verbose, diag), SOURCES))
    objs = [o for o, _ in results]
    if force or any(ch for _, ch in results) or _stale(lib, objs):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib


def build_binding(force=False, verbose=False):
    """The compiled binding layer (csrc/binding.cpp -> lib/_sc_fast.so): host-only C++ (g++), built against torch's
    headers for tensor allocation and against libstreet_crafter_hip.so for the C ABI it calls.  ~1 minute."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    src = os.path.join(CSRC, "binding.cpp")
    deps = [src, os.path.join(HERE, "..", "include", "street_crafter_amd.h"), LIB]     # (a rebuilt library: rebuild the binding too)
    if not os.path.exists(LIB):
        raise RuntimeError("build the HIP library first")
    if not force and not _stale(BINDING, deps):
        return BINDING
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", BINDING,
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-DTORCH_EXTENSION_NAME=_sc_fast",
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-Wno-deprecated-declarations", f'-DSC_ABI_HASH="{abi_hash()}"',
           *[f"-I{i}" for i in ce.include_paths()], f"-I{sysconfig.get_paths()['include']}",
           f"-L{LIBDIR}", "-lstreet_crafter_hip", f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python",
           "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"binding build failed:\n{r.stdout}\n{r.stderr[-4000:]}")
    return BINDING


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print("built", path)
    print("built", build_binding(force="--force" in sys.argv, verbose=True))
    if "--diag" in sys.argv:
        print("built", build(force="--force" in sys.argv, verbose=True, diag=True))
