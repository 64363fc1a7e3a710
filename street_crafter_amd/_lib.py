"""ctypes binding of the C ABI declared in include/street_crafter_amd.h.

The HIP shared library is the product: there is NO CPU / torch fallback.  If the library is
missing, importing an operator raises ImportError naming the build command; if a tensor is not
on a HIP device the operator raises before launching anything.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libstreet_crafter_hip.so")
DIAG_LIB_PATH = os.path.join(_HERE, "lib", "libstreet_crafter_hip_diag.so")

_lib = None
_path = LIB_PATH
_lock = __import__("threading").Lock()      # two host threads may make the first call at the same time


def use_diagnostic_build(tag: str = ""):
    """tools/exp_*.py only: load the DIAGNOSTIC build (`python -m street_crafter_amd.build --diag`; same sources
    with the sc_set_option("debug0".."debug3") skip switches compiled in) instead of the shipped library.  Must be
    called before the first operator call; nothing in the package, tests/ or bench.py calls it.
    `tag`: an experiment build made with SC_DIAG_TAG=tag SC_EXP_DEFS="-D..." (tools/ab_lib.py)."""
    global _path
    want = DIAG_LIB_PATH.replace("_diag.so", f"_diag_{tag}.so") if tag else DIAG_LIB_PATH
    if _lib is not None and _path != want:
        raise RuntimeError("use_diagnostic_build() must be called before the library is first loaded")
    if not os.path.exists(want):
        raise ImportError(f"diagnostic library not built ({want} missing): "
                          "python -m street_crafter_amd.build --diag")
    _path = want

c_f32p = C.c_void_p   # device pointers travel as integers
c_i32p = C.c_void_p
c_i64p = C.c_void_p
c_u8p = C.c_void_p
c_u64p = C.c_void_p
c_stream = C.c_void_p

# name -> (restype, argtypes); must match include/street_crafter_amd.h exactly
SIGNATURES = {
    "sc_version": (C.c_char_p, []),
    "sc_error_string": (C.c_char_p, [C.c_int]),
    "sc_wait_i64": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64]),
    "sc_target_arch": (C.c_char_p, []),
    "sc_projection_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, c_i32p, c_f32p,
                                    c_f32p, c_f32p, c_f32p, c_stream]),
    "sc_projection_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_float, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                    c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "sc_isect_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "sc_isect_count": (C.c_int, [c_f32p, c_i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_i32p,
                                 c_i64p, C.c_void_p, C.c_size_t, c_stream]),
    "sc_isect_emit": (C.c_int, [c_f32p, c_i32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                c_i32p, C.c_int64, c_i64p, c_i32p, C.c_void_p, C.c_size_t, c_stream]),
    "sc_radix_sort_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "sc_radix_sort_pairs_u64_i32": (C.c_int, [c_u64p, c_i32p, c_u64p, c_i32p, C.c_int64, C.c_int,
                                              C.c_void_p, C.c_size_t, c_stream]),
    "sc_isect_bin_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64]),
    "sc_isect_bin_count": (C.c_int, [c_f32p, c_i32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_i32p,
                                     c_i32p, c_i64p, c_i64p, C.c_int64, C.c_void_p, C.c_size_t, c_i32p, c_f32p, c_i32p,
                                     c_i32p, c_stream]),
    "sc_view_slots": (C.c_int, []),
    "sc_view_registry_words": (C.c_int, []),
    "sc_isect_bin_bucket_capacity": (C.c_int, []),
    "sc_isect_bin_sort": (C.c_int, [c_f32p, c_i32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    c_i32p, c_i64p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, c_i64p,
                                    c_i32p, C.c_void_p, C.c_size_t, c_stream]),
    "sc_isect_ids_rebuild": (C.c_int, [c_i32p, c_i32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, c_i64p,
                                       c_stream]),
    "sc_isect_bin_reset_cursors": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, c_stream]),
    "sc_isect_offsets": (C.c_int, [c_i64p, C.c_int64, C.c_int, C.c_int, C.c_int, c_i32p, c_stream]),
    "sc_sh_fwd": (C.c_int, [C.c_int, c_f32p, c_f32p, c_u8p, C.c_int64, C.c_int, c_f32p, c_stream]),
    "sc_sh_bwd": (C.c_int, [C.c_int, c_f32p, c_f32p, c_u8p, C.c_int64, C.c_int, c_f32p, c_f32p, c_f32p,
                            c_stream]),
    "sc_rasterize_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_u8p, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_i32p, c_i32p, C.c_int64,
                                   c_f32p, c_f32p, c_i32p, c_i32p, c_i32p, c_stream]),
    "sc_rasterize_fwd_ed": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_u8p, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_i32p, c_i32p, C.c_int64,
                                      c_f32p, c_f32p, c_i32p, c_i32p, c_stream]),
    "sc_rasterize_fwd_planar": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_u8p, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_i32p, c_i32p, C.c_int64,
                                          c_f32p, c_f32p, c_i32p, c_i32p, c_stream]),
    "sc_camera_centers": (C.c_int, [c_f32p, C.c_int, c_f32p, c_stream]),
    "sc_projection_sh_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                       C.c_float, C.c_float, C.c_float, C.c_int, c_i32p, c_f32p, c_f32p,
                                       c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "sc_rasterize_fwd_packed": (C.c_int, [c_f32p, c_f32p, c_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          c_i32p, c_i32p, C.c_int64, c_f32p, c_f32p, c_i32p, c_i32p, C.c_int, c_stream]),
    "sc_records_unpack": (C.c_int, [c_f32p, C.c_int64, c_f32p, c_f32p, c_f32p, c_stream]),
    "sc_tile_order_len": (C.c_int, [C.c_int]),
    "sc_rasterize_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_u8p, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_i32p, c_i32p, C.c_int64,
                                   c_f32p, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                   c_i32p, c_stream]),
    "sc_knn_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "sc_knn3_mean_dist2": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "sc_frame_composite_u8": (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, C.c_int, C.c_int64, C.c_int, c_u8p,
                                        c_stream]),
    "sc_frame_composite_u8_strided": (C.c_int, [c_f32p, C.c_int64, C.c_int64, c_f32p, c_f32p, C.c_int64, C.c_int64, C.c_int64,
                                                C.c_int, c_u8p, c_stream]),
    "sc_test_wave_transpose_sum16": (C.c_int, [c_f32p, C.c_int, c_f32p, c_stream]),
    "sc_stream_create": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "sc_stream_destroy": (C.c_int, [c_stream]),
    "sc_stream_priority_range": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sc_set_option": (C.c_int, [C.c_char_p, C.c_int]),
}


def load():
    """Returns the ctypes handle, loading it on first use; never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        return _load_locked()


def _load_locked():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_path):
        raise ImportError(
            f"street_crafter_amd: HIP library not built ({_path} missing). "
            "Build it with `python -m street_crafter_amd.build` (needs hipcc, gfx950).")
    lib = C.CDLL(_path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # the upstream-version-dependent constants of fully_fused_projection (include/street_crafter_amd.h, sc_set_option
    # "proj_clamp" / "radius_floor"), for a user who holds the real gsplat fork and need not touch code to match it
    env_c, env_f = os.environ.get("SC_PROJ_CLAMP"), os.environ.get("SC_RADIUS_FLOOR")
    if env_c:
        if env_c not in ("symmetric", "asymmetric"):
            raise ValueError(f"SC_PROJ_CLAMP must be symmetric or asymmetric, got {env_c!r}")
        lib.sc_set_option(b"proj_clamp", 1 if env_c == "asymmetric" else 0)
    if env_f:
        if float(env_f) not in (0.01, 0.1):
            raise ValueError(f"SC_RADIUS_FLOOR must be 0.01 or 0.1, got {env_f!r}")
        lib.sc_set_option(b"radius_floor", 1 if float(env_f) == 0.1 else 0)
    return lib


FAST_PATH = os.path.join(_HERE, "lib", "_sc_fast.so")
_fast = None
_fast_enabled = True


def fast():
    """The compiled binding layer (csrc/binding.cpp: one call per operator validates the tensors, allocates every
    output / workspace through torch's caching allocator and calls the same C-ABI entry point as the ctypes table
    above), or None when it is switched off (set_fast_binding(False), or the diagnostic build is selected: the binding
    is linked against the shipped library).  A missing binding is an ImportError, never a silent fallback."""
    global _fast
    if not _fast_enabled or _path != LIB_PATH:
        return None
    if _fast is None:
        load()
        with _lock:
            return _fast_locked()
    return _fast


def _fast_locked():
    global _fast
    if _fast is None:
        if not os.path.exists(FAST_PATH):
            raise ImportError(f"street_crafter_amd: binding layer not built ({FAST_PATH} missing). "
                              "Build it with `python -m street_crafter_amd.build`.")
        import importlib.util
        import torch  # noqa: F401  (libtorch must be loaded before the extension)
        spec = importlib.util.spec_from_file_location("_sc_fast", FAST_PATH)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if mod.abi_version().encode() != load().sc_version():
            # (each binary carries the digest of the include/street_crafter_amd.h it was compiled against)
            raise ImportError("street_crafter_amd: _sc_fast.so and libstreet_crafter_hip.so were built against different "
                              f"editions of the C ABI ({mod.abi_version()!r} vs {load().sc_version().decode()!r}): "
                              "python -m street_crafter_amd.build")
        _fast = mod
    return _fast


def set_fast_binding(enabled: bool) -> bool:
    """A/B switch between the compiled binding layer (default) and the ctypes table; same C ABI, same kernels, same
    results.  Returns the previous setting."""
    global _fast_enabled
    prev, _fast_enabled = _fast_enabled, bool(enabled)
    return prev


def check(code: int, what: str):
    if code != 0:
        msg = load().sc_error_string(int(code)).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {code})")


def set_projection_variant(proj_clamp: str = None, radius_floor: float = None):
    """Selects the upstream-version-dependent constants of fully_fused_projection (the reference installs
    an unpinned gsplat fork, README.md:35): proj_clamp "symmetric" (1.3 tan(fov/2), gsplat v1.0-1.3, default) or "asymmetric"
    ((W - cx)/fx + 0.3 tan, cx/fx + 0.3 tan: v1.4+); radius_floor 0.01 (v1.x, default) or 0.1 (Inria / early forks).  None
    leaves a setting as it is.  Returns the previous (proj_clamp, radius_floor)."""
    lib = load()
    cur_c = lib.sc_set_option(b"proj_clamp", 0)
    lib.sc_set_option(b"proj_clamp", cur_c)
    cur_f = lib.sc_set_option(b"radius_floor", 0)
    lib.sc_set_option(b"radius_floor", cur_f)
    if proj_clamp is not None:
        if proj_clamp not in ("symmetric", "asymmetric"):
            raise ValueError(f"proj_clamp must be 'symmetric' or 'asymmetric', got {proj_clamp!r}")
        lib.sc_set_option(b"proj_clamp", 1 if proj_clamp == "asymmetric" else 0)
    if radius_floor is not None:
        if float(radius_floor) not in (0.01, 0.1):
            raise ValueError(f"radius_floor must be 0.01 or 0.1, got {radius_floor!r}")
        lib.sc_set_option(b"radius_floor", 1 if float(radius_floor) == 0.1 else 0)
    return ("asymmetric" if cur_c else "symmetric", 0.1 if cur_f else 0.01)


def set_option(key: str, value: int) -> int:
    prev = load().sc_set_option(key.encode(), int(value))
    if prev < 0:
        raise ValueError(f"unknown option {key}={value}")
    return prev
