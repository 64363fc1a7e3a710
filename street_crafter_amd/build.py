"""Builds the HIP shared library in-tree with hipcc for gfx950.

    python -m street_crafter_amd.build            # incremental
    python -m street_crafter_amd.build --force

Output: street_crafter_amd/lib/libstreet_crafter_hip.so (git-ignored, shipped to the GPU box by
gpurun).  hipcc cross-compiles without a GPU, so this also is the "does it build" check.
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
ARCH = "gfx950"

SOURCES = ["capi.hip", "projection.hip", "isect.hip", "isect_bin.hip", "radix_sort.hip", "sh.hip",
           "raster_fwd.hip", "raster_bwd.hip", "knn.hip", "fused_fwd.hip"]
COMMON_FLAGS = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
                "-fno-gpu-rdc"]


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


def _compile(src, force, verbose):
    obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
    if not force and not _stale(obj, _deps(src)):
        return obj, False
    cmd = [_hipcc(), *COMMON_FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip() and verbose:
        print(r.stderr, file=sys.stderr)
    return obj, True


def build(force=False, verbose=False, jobs=None):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    jobs = jobs or min(len(SOURCES), max(1, (os.cpu_count() or 2) - 1))
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        results = list(ex.map(lambda s: _compile(s, force, verbose), SOURCES))
    objs = [o for o, _ in results]
    if force or any(ch for _, ch in results) or _stale(LIB, objs):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print("built", path)
