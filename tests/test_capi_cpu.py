"""No-GPU checks of the drop-in boundary: the shared library builds/loads, exports every symbol
declared in include/*.h, the ctypes table matches the header, and the Python operators refuse
CPU tensors (there is no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from street_crafter_amd import build
    build.build()
    from street_crafter_amd import _lib
    return _lib.load()


def _declared():
    names = set()
    inc = os.path.join(ROOT, "include")
    for f in os.listdir(inc):
        if f.endswith(".h"):
            src = open(os.path.join(inc, f)).read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            names |= set(re.findall(r"\b(sc_[a-z0-9_]+)\s*\(", src))
    return names


def test_library_exports_every_declared_symbol(lib):
    from street_crafter_amd import _lib
    declared = _declared()
    assert len(declared) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} declared in include/ but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_library_info_and_errors(lib):
    assert b"gfx950" in lib.sc_target_arch()
    assert lib.sc_error_string(0) == b"ok"
    assert b"invalid" in lib.sc_error_string(-1)
    assert b"workspace" in lib.sc_error_string(-2)
    assert lib.sc_set_option(b"no_such_option", 1) == -1
    prev = lib.sc_set_option(b"raster_fwd", 0)
    assert lib.sc_set_option(b"raster_fwd", prev) == 0


def test_argument_validation_without_gpu(lib):
    # rejected before any launch: nothing here touches a device
    assert lib.sc_projection_fwd(None, None, None, None, None, 1, 8, 0, 0, 0.3, 0.01, 1e10, 0.0,
                                 None, None, None, None, None, None) == -1
    assert lib.sc_projection_fwd(None, None, None, None, None, 1, 8, 64, 64, 0.3, 0.01, 1e10, 0.0,
                                 None, None, None, None, None, None) == -1        # null pointers
    assert lib.sc_sh_fwd(5, None, None, None, 4, 36, None, None) == -1              # degree > 4
    assert lib.sc_sh_fwd(3, None, None, None, 4, 9, None, None) == -1               # K < 16
    assert lib.sc_rasterize_fwd(None, None, None, None, None, None, 1, 4, 33, 64, 64, 16, 4, 4,
                                None, None, 0, None, None, None, None, None, None) == -1        # D > 32
    assert lib.sc_rasterize_fwd(None, None, None, None, None, None, 1, 4, 3, 65, 64, 16, 4, 4,
                                None, None, 0, None, None, None, None, None, None) == -1        # tiles too few
    assert lib.sc_radix_sort_pairs_u64_i32(None, None, None, None, 10, 65, None, 0, None) == -1
    assert lib.sc_radix_sort_pairs_u64_i32(None, None, None, None, 1, 40, None, 0, None) == 0   # n<=1 no-op
    # round 4: planar rasterizer output, strided frame export, stream factory -- rejected before any launch / HIP call
    assert lib.sc_rasterize_fwd_planar(None, None, None, None, None, None, 1, 4, 33, 64, 64, 16, 4, 4,
                                       None, None, 0, None, None, None, None, None) == -1               # D > 32
    assert lib.sc_rasterize_fwd_planar(None, None, None, None, None, None, 0, 4, 4, 64, 64, 16, 4, 4,
                                       None, None, 0, None, None, None, None, None) == 0                # C == 0: nothing to do
    assert lib.sc_frame_composite_u8_strided(None, 0, 1, None, None, 1, 1, 10, 0, None, None) == -1     # pixel stride < 1
    assert lib.sc_frame_composite_u8_strided(None, 1, 1, None, None, 1, 1, 0, 0, None, None) == 0       # no pixels
    assert lib.sc_frame_composite_u8_strided(None, 1, 1, None, None, 1, 1, 10, 2, None, None) == -1     # rounding mode
    assert lib.sc_stream_create(0, None, 0, None) == -1                                                 # no place for the handle
    assert lib.sc_stream_create(0, None, 4, ctypes.byref(ctypes.c_void_p())) == -1                      # mask words without a mask
    zero_mask = (ctypes.c_uint32 * 8)()
    assert lib.sc_stream_create(0, ctypes.cast(zero_mask, ctypes.c_void_p), 8, ctypes.byref(ctypes.c_void_p())) == -1   # empty mask
    assert lib.sc_stream_destroy(None) == -1 and lib.sc_stream_priority_range(None, None) == -1
    for key, bad in ((b"proj_clamp", 2), (b"radius_floor", 2), (b"isect_pull", 2)):
        assert lib.sc_set_option(key, bad) == -1
        prev = lib.sc_set_option(key, 1)
        assert prev in (0, 1) and lib.sc_set_option(key, prev) == 1
    # fused forward entries
    assert lib.sc_camera_centers(None, -1, None, None) == -1
    assert lib.sc_camera_centers(None, 0, None, None) == 0
    assert lib.sc_projection_sh_fwd(None, None, None, None, None, None, None, None, 1, 8, 4, 5, 64, 64, 0.3, 0.01,
                                    1e10, 0.0, 1, None, None, None, None, None, None, None, None) == -1     # degree > 4
    assert lib.sc_projection_sh_fwd(None, None, None, None, None, None, None, None, 1, 8, 3, 1, 64, 64, 0.3, 0.01,
                                    1e10, 0.0, 1, None, None, None, None, None, None, None, None) == -1     # K < 4
    assert lib.sc_projection_sh_fwd(None, None, None, None, None, None, None, None, 1, 0, 4, 1, 64, 64, 0.3, 0.01,
                                    1e10, 0.0, 1, None, None, None, None, None, None, None, None) == 0      # N == 0
    assert lib.sc_rasterize_fwd_packed(None, None, None, 1, 4, 64, 64, 4, 4, None, None, 0, None, None, None, None, 1,
                                       None) == -1
    assert lib.sc_records_unpack(None, 8, None, None, None, None) == -1 and lib.sc_records_unpack(None, 0, None, None, None, None) == 0
    assert lib.sc_rasterize_fwd_ed(None, None, None, None, None, None, 1, 4, 3, 64, 64, 16, 4, 4,
                                   None, None, 0, None, None, None, None, None) in (-1, -3)   # needs D == 4
    assert lib.sc_isect_bin_count(None, None, None, 1, 8, 16, 4, 4, None, None, None, None, 0, None, 0, None, None,
                                  None, None, None) == -1
    assert lib.sc_view_slots() == 8 and lib.sc_view_registry_words() == 4 + 4 * 8
    assert lib.sc_knn3_mean_dist2(None, 0, None, None, 0, None) == 0
    assert lib.sc_knn_workspace_bytes(1000) >= 1000 * 36
    assert lib.sc_isect_workspace_bytes(1_000_000) >= (1_000_000 // 256) * 8


def test_host_side_waits_and_list_sizes(lib):
    """The two host-only entry points: the GIL-free wait for the intersection counts' sequence number (here on plain
    host memory: value already there, value never arrives -> timeout, a second thread delivers it) and the size of
    the dispatch-list buffer (forward list incl. room for half tiles + the whole-tile list)."""
    import ctypes
    import threading
    import time
    word = (ctypes.c_int64 * 1)(7)
    addr = ctypes.addressof(word)
    assert lib.sc_wait_i64(addr, 7, 0) == 0
    t0 = time.perf_counter()
    assert lib.sc_wait_i64(addr, 8, 20_000) == 1                    # 20 ms timeout
    assert 0.015 < time.perf_counter() - t0 < 2.0
    assert lib.sc_wait_i64(None, 0, 0) == -1

    def deliver():
        time.sleep(0.05)
        word[0] = 9

    th = threading.Thread(target=deliver)
    th.start()
    assert lib.sc_wait_i64(addr, 9, 5_000_000) == 0                 # the waiting call holds no GIL: the thread runs
    th.join()
    for tiles in (0, 1, 6, 425, 9600, 36864):
        assert lib.sc_tile_order_len(tiles) == (tiles + tiles // 8 + 8) + tiles + 2
    assert lib.sc_tile_order_len(-3) == 0
    for key in (b"raster_split", b"raster_hint_blend", b"raster_bwd_split", b"raster_map"):
        prev = lib.sc_set_option(key, 0)
        assert prev >= 0 and lib.sc_set_option(key, prev) == 0
    assert lib.sc_set_option(b"raster_split", 101) == -1 and lib.sc_set_option(b"raster_hint_blend", 5) == -1


def test_work_hint_buffers_are_keyed_and_bounded():
    """The rasterizer's per-tile work hints: one buffer per (device, cameras, Gaussian count, tile grid) -- the two
    passes of a novel-view frame must not share one -- and at most 8 of them (densification changes N)."""
    from street_crafter_amd import _lib, rendering
    rendering._STATE.tile_work.clear()
    dev = torch.device("cpu")
    a = rendering._tile_work(dev, 1, 1000, 4, 3)
    b = rendering._tile_work(dev, 1, 31, 4, 3)
    assert a.shape == b.shape == (12 * _lib.load().sc_view_slots(),) and a.data_ptr() != b.data_ptr() and a.dtype == torch.int32
    assert rendering._tile_work(dev, 1, 1000, 4, 3).data_ptr() == a.data_ptr()
    assert rendering._tile_work(dev, 2, 1000, 4, 3).shape == (24 * _lib.load().sc_view_slots(),)
    for n in range(2000, 2010):
        rendering._tile_work(dev, 1, n, 4, 3)
    assert len(rendering._STATE.tile_work) == 8
    assert rendering._tile_work(dev, 1, 2009, 4, 3) is rendering._STATE.tile_work[(None, 1, 2009, 4, 3)]
    assert (None, 1, 1000, 4, 3) not in rendering._STATE.tile_work          # the oldest were dropped
    rendering._STATE.tile_work.clear()


def test_reset_state_forgets_the_hints_between_calls_per_device():
    """rendering.reset_state: the tables the operators keep between calls (size predictions + history + last counts,
    work-hint buffers, view registries) are dropped for one device or for all; switches keep their values."""
    from street_crafter_amd import rendering
    tables = (rendering._STATE.prediction, rendering._STATE.history, rendering._STATE.last_meta, rendering._STATE.tile_work,
              rendering._STATE.view_registry)
    saved = [dict(t) for t in tables]
    try:
        for t in tables:
            t.clear()
        for dev in (0, 1):
            rendering._STATE.prediction[(dev, 1, 10, 16, 4, 3)] = (1, 1, 1)
            rendering._STATE.history[(dev, 1, 10, 16, 4, 3)] = [(1, 1, 1)]
            rendering._STATE.last_meta[(dev, 1, 10, 16, 4, 3)] = (1, 1, 1)
            rendering._STATE.tile_work[(dev, 1, 10, 4, 3)] = torch.zeros(4, dtype=torch.int32)
            rendering._STATE.view_registry[dev] = torch.zeros(4, dtype=torch.int32)
        prev = rendering.set_deferred_isect(False)
        assert rendering.reset_state(1) == {"predictions": 1, "history": 1, "last_meta": 1, "tile_work": 1, "view_registry": 1}
        assert all(len(t) == 1 for t in tables) and (0, 1, 10, 16, 4, 3) in rendering._STATE.prediction and 0 in rendering._STATE.view_registry
        assert rendering.reset_state(torch.device("cuda", 0))["predictions"] == 1 and not any(len(t) for t in tables)
        assert rendering.reset_state() == {"predictions": 0, "history": 0, "last_meta": 0, "tile_work": 0, "view_registry": 0}
        assert rendering.set_deferred_isect(prev) is False          # (a switch, not state: untouched)
    finally:
        for t, sv in zip(tables, saved):
            t.clear()
            t.update(sv)


def test_operators_refuse_cpu_tensors(lib):
    from gsplat.rendering import (fully_fused_projection, isect_offset_encode, isect_tiles,
                                  rasterize_to_pixels, spherical_harmonics, rasterization)  # noqa: F401
    from simple_knn._C import distCUDA2
    n = 8
    with pytest.raises(RuntimeError, match="HIP device"):
        fully_fused_projection(torch.zeros(n, 3), None, torch.zeros(n, 4), torch.zeros(n, 3),
                               torch.eye(4)[None], torch.eye(3)[None], 64, 64)
    with pytest.raises(RuntimeError, match="HIP device"):
        spherical_harmonics(1, torch.zeros(1, n, 3), torch.zeros(1, n, 4, 3))
    with pytest.raises(RuntimeError, match="HIP device"):
        isect_tiles(torch.zeros(1, n, 2), torch.zeros(1, n, dtype=torch.int32), torch.zeros(1, n), 16, 4, 4)
    with pytest.raises(RuntimeError, match="HIP device"):
        isect_offset_encode(torch.zeros(3, dtype=torch.int64), 1, 4, 4)
    with pytest.raises(RuntimeError, match="HIP device"):
        rasterize_to_pixels(torch.zeros(1, n, 2), torch.zeros(1, n, 3), torch.zeros(1, n, 3), torch.zeros(1, n),
                            64, 64, 16, torch.zeros(1, 4, 4, dtype=torch.int32), torch.zeros(0, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="HIP device"):
        distCUDA2(torch.zeros(n, 3))
    with pytest.raises(NotImplementedError):
        fully_fused_projection(torch.zeros(n, 3), torch.zeros(n, 3, 3), None, None, torch.eye(4)[None],
                               torch.eye(3)[None], 64, 64)


def test_product_path_never_imports_oracle():
    """The oracle is test infrastructure: no product module may import it."""
    bad = []
    for pkg in ("street_crafter_amd", "gsplat", "simple_knn"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, pkg)):
            for f in files:
                if f.endswith(".py"):
                    src = open(os.path.join(dirpath, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_shipped_library_has_no_diagnostic_switches(lib):
    """the `debug0..3` skip switches (two GPU memory faults in two rounds) exist only in the
    separate diagnostic build.  The shipped library does not know the keys and exports no knob array; nothing in the
    product packages, the harness, tests/ or bench.py asks for the diagnostic build; the transcribed caller sequence
    lives outside the product package."""
    import subprocess
    from street_crafter_amd import _lib
    for k in range(4):
        assert lib.sc_set_option(f"debug{k}".encode(), 0) == -1
        with pytest.raises(ValueError):
            _lib.set_option(f"debug{k}", 1)
    assert b"DIAGNOSTIC" not in lib.sc_version()
    syms = subprocess.run(["nm", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "g_sc_debug" not in syms
    assert _lib._path == _lib.LIB_PATH
    users = []
    for top in ("street_crafter_amd", "gsplat", "simple_knn", "harness", "tests"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith(".py") and f not in ("_lib.py", "build.py", "test_capi_cpu.py"):     # (build.py: its docstring)
                    if "use_diagnostic_build" in open(os.path.join(dirpath, f)).read():
                        users.append(os.path.join(dirpath, f))
    for f in ("bench.py", "__graft_entry__.py"):
        if "use_diagnostic_build" in open(os.path.join(ROOT, f)).read():
            users.append(f)
    assert not users, users
    assert not os.path.exists(os.path.join(ROOT, "street_crafter_amd", "pipeline.py"))
    # no source of the product reads a knob outside the SC_DIAG macros
    for f in os.listdir(os.path.join(ROOT, "street_crafter_amd", "csrc")):
        src = open(os.path.join(ROOT, "street_crafter_amd", "csrc", f)).read()
        for line in src.splitlines():
            if "g_sc_debug" in line:
                assert "SC_DIAG" in line or line.lstrip().startswith(("//", "extern int g_sc_debug", "int g_sc_debug",
                                                                       "const int prev = g_sc_debug", "g_sc_debug[key")), (f, line)


def test_densification_stats_mirror_cpu():
    """Host logic of the densification-statistics consumer (street_gaussian_model.py:487-521) on CPU
    tensors: per-model slicing, absgrad/grad columns, pixel scaling, visibility gating, NaN -> 0."""
    import torch
    from street_crafter_amd.densify_stats import DensificationStats
    n, W, H = 10, 200, 100
    st = DensificationStats({"background": (0, 6), "obj_001": (6, 10)}, device="cpu")
    vp = torch.zeros(1, n, 2)
    vp.grad = torch.arange(2 * n, dtype=torch.float32).reshape(1, n, 2) * 0.01
    vp.absgrad = vp.grad * 3.0
    vis = torch.tensor([True, False] * 5)
    radii = torch.arange(n, dtype=torch.int32)
    st.set_max_radii2D(radii / float(max(H, W)), vis)
    st.add_densification_stats(vp, vis, W, H)
    st.add_densification_stats(vp, vis, W, H)
    scale = torch.tensor([0.5 * W, 0.5 * H])
    want_grad = 2 * torch.norm(vp.grad[0] * scale, dim=-1)
    want_abs = 2 * torch.norm(vp.absgrad[0] * scale, dim=-1)
    acc = torch.cat([st.xyz_gradient_accum["background"], st.xyz_gradient_accum["obj_001"]])
    den = torch.cat([st.denom["background"], st.denom["obj_001"]])[:, 0]
    assert torch.allclose(acc[vis, 0], want_abs[vis]) and torch.allclose(acc[vis, 1], want_grad[vis])
    assert float(acc[~vis].abs().sum()) == 0.0
    assert torch.equal(den, vis.float() * 2)
    mr = torch.cat([st.max_radii2D["background"], st.max_radii2D["obj_001"]])
    assert torch.allclose(mr[vis], radii[vis].float() / 200.0) and float(mr[~vis].sum()) == 0.0
    g = st.mean_grads("obj_001", use_abs=True)
    assert g.shape == (4, 1) and not torch.isnan(g).any() and float(g[1]) == 0.0   # invisible row: 0/0 -> 0
    clone, split = st.clone_split_masks("background", 0.0, torch.tensor([0.001, 1, 0.001, 1, 0.001, 1.0]), 10.0)
    assert clone.tolist() == [True, False, True, False, True, False] and split.tolist() == [False, True] * 3
    # no absgrad attribute -> plain grad path (street_gaussian_model.py:510-511)
    vp2 = torch.zeros(1, n, 2)
    vp2.grad = torch.ones(1, n, 2)
    st.reset()
    st.add_densification_stats(vp2, vis, W, H)
    assert torch.allclose(st.xyz_gradient_accum["background"][0], torch.tensor([2 ** 0.5, 0.0]))


def test_bench_algorithmic_bytes_are_the_survey_formula():
    """B_alg = 197 N + 88 I + 24 P + 4 T for K = 4; the per-operator split bench.py uses for
    `roofline.achieved` must add up to it (341 N for K = 16)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from harness.caller import algorithmic_bytes
    N, I, W, H = 1_000_000, 19_753_547, 1920, 1280
    P, T = W * H, 120 * 80
    for K, per_gauss in ((4, 197), (16, 341)):
        total = sum(bench.stage_algorithmic_bytes(s, N, I, P, T, K) for s in
                    ("projection", "isect_tiles", "isect_offset_encode", "spherical_harmonics", "rasterize_to_pixels"))
        assert total == per_gauss * N + 88 * I + 24 * P + 4 * T
        assert algorithmic_bytes(N, I, W, H, 16, K) == total


def test_bench_refuses_stale_pmc_traffic_records():
    """bench.py's `roofline.traffic` comes from committed rocprofv3 --pmc passes; a record collected on another kernel
    symbol (or one that does not say which) must not be reported."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sym = bench.OPERATOR_KERNEL["rasterize_to_pixels"]
    good = {"rasterize_to_pixels": {"n1000000": 2.0e8}, "_kernels": {"n1000000": {"rasterize_to_pixels": [sym]}}}
    assert bench.pmc_traffic_entry(good, "rasterize_to_pixels", "n1000000", sym)[0] == 2.0e8
    assert bench.pmc_traffic_entry(good, "rasterize_to_pixels", "n1000000", "raster_fwd_wave_kernel<4, false, false>")[0] is None
    assert bench.pmc_traffic_entry(good, "rasterize_to_pixels", "n5", sym)[0] is None
    assert bench.pmc_traffic_entry({"rasterize_to_pixels": {"n1000000": 2.0e8}}, "rasterize_to_pixels", "n1000000", sym)[0] is None
    # the backward's byte model: 92 B per intersection + 28 B per pixel; knn: 104 B per point
    assert bench.raster_bwd_algorithmic_bytes(1000, 10) == 92 * 1000 + 280 and bench.KNN_BYTES_PER_POINT == 104
    # the committed record, if present, must be usable by THIS build's kernel symbols or be refused -- never misread
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tj):
        import json
        val, why = bench.pmc_traffic_entry(json.load(open(tj)), "rasterize_to_pixels", "n1000000", sym)
        assert val is None or val > 1e6, (val, why)


def test_lazy_tensor_fills_on_first_read_only():
    """street_crafter_amd/lazy.py: metadata never triggers the fill, any read does, exactly once."""
    import numpy as np
    import torch
    from street_crafter_amd.lazy import LazyTensor
    calls = []

    def fill(t):
        calls.append(t.numel())
        t.copy_(torch.arange(t.numel(), dtype=t.dtype) * 3)

    z = LazyTensor(torch.empty(7, dtype=torch.int64), fill)
    assert isinstance(z, torch.Tensor) and not z.is_materialized
    assert (z.shape, z.numel(), z.dtype, z.dim(), z.is_cuda, str(z.device), z.is_contiguous()) == \
           (torch.Size([7]), 7, torch.int64, 1, False, "cpu", True)
    z._sc_note = "attributes can be attached"
    assert calls == [] and z._version == 0
    assert (z + 1).tolist() == [1, 4, 7, 10, 13, 16, 19] and calls == [7] and z.is_materialized
    assert type(z + 1) is torch.Tensor and z[2:4].tolist() == [6, 9] and calls == [7]          # filled once
    for read in (lambda t: t.cpu(), lambda t: t.numpy(), lambda t: np.asarray(t), lambda t: t.tolist(),
                 lambda t: t.data_ptr(), lambda t: repr(t), lambda t: torch.equal(t, t.clone()),
                 lambda t: t.view(torch.int32), lambda t: t.contiguous(), lambda t: torch.cat([t, t]),
                 lambda t: t.sum().item()):
        n0 = len(calls)
        t = LazyTensor(torch.empty(4, dtype=torch.int64), fill)
        read(t)
        assert len(calls) == n0 + 1, read
        assert t.tolist() == [0, 3, 6, 9]
    assert LazyTensor(torch.empty(0, dtype=torch.int64), fill).materialize().numel() == 0


def test_lazy_tensor_copies_and_serialisation_are_plain_filled_tensors():
    """state that rides on tensor objects must survive what callers do to tensors.  A LazyTensor
    that is copied, deep-copied, pickled, torch.save-d, cloned, detached or converted yields the FILLED contents (the
    fill runs once), and whatever leaves by deepcopy / pickle is an ordinary torch.Tensor (the pending fill is a closure
    over device buffers: it must not travel)."""
    import copy
    import io
    import pickle
    import torch
    from street_crafter_amd.lazy import LazyTensor
    want = [0, 3, 6, 9, 12]

    def make(calls):
        def fill(t):
            calls.append(1)
            t.copy_(torch.arange(t.numel(), dtype=t.dtype) * 3)
        z = LazyTensor(torch.empty(5, dtype=torch.int64), fill)
        z._sc_offsets = ("cached offsets ride along on the object",)
        return z

    def save_load(t):
        b = io.BytesIO()
        torch.save(t, b)
        b.seek(0)
        return torch.load(b, weights_only=True)

    for name, fn, plain in (("copy", copy.copy, False), ("deepcopy", copy.deepcopy, True),
                            ("pickle", lambda t: pickle.loads(pickle.dumps(t)), True), ("torch.save", save_load, True),
                            ("clone", lambda t: t.clone(), True), ("detach", lambda t: t.detach(), True),
                            ("to", lambda t: t.to(torch.int32), True), ("contiguous", lambda t: t.contiguous(), None)):
        calls = []
        z = make(calls)
        r = fn(z)
        assert r.tolist() == want and z.tolist() == want and calls == [1], (name, r.tolist(), calls)
        if plain:
            assert type(r) is torch.Tensor, (name, type(r))
    # a list / dict holding one (what copy.deepcopy(meta) does)
    calls = []
    d = copy.deepcopy({"isect_ids": make(calls), "n": 3})
    assert type(d["isect_ids"]) is torch.Tensor and d["isect_ids"].tolist() == want and calls == [1]


def test_lazy_tensor_with_a_deferred_shape():
    """lazy.py, round 3: a LazyTensor whose LENGTH is settled on first observation (isect_tiles hands out flatten_ids /
    isect_ids before the frame's intersection count has reached the host).  What does not depend on the length is
    answered without resolving; shape, numel, len and every read resolve first (once), the same Python object then has
    the true length; copies and serialisation hand out plain tensors of the true length."""
    import copy
    import pickle
    import torch
    from street_crafter_amd.lazy import LazyTensor
    cap = torch.zeros(10, dtype=torch.int32)
    log = []

    def make(fill=True):
        def resolve(t):
            log.append("resolve")
            t.set_(cap.untyped_storage(), 0, (4,), (1,))

        def fill_(t):
            log.append("fill")
            t.copy_(torch.arange(4, dtype=torch.int32) * 5)
        return LazyTensor(cap[:0], fill_ if fill else None, resolve)

    z = make()
    assert isinstance(z, torch.Tensor) and not z.is_resolved and not z.is_materialized
    assert (z.dtype, str(z.device), z.ndim, z.dim(), z.is_cuda, z.requires_grad, z.is_contiguous()) == \
           (torch.int32, "cpu", 1, 1, False, False, True)
    z._sc_note = "attributes ride along"
    assert log == [] and z._version == 0
    assert z.shape == (4,) and log == ["resolve"] and z.is_resolved and not z.is_materialized       # shape: resolve only
    assert z.numel() == 4 and len(z) == 4 and log == ["resolve"]
    assert z.tolist() == [0, 5, 10, 15] and log == ["resolve", "fill"] and z.is_materialized and z._sc_note
    for observe in (lambda t: t.shape, lambda t: t.numel(), lambda t: len(t), lambda t: t.size(0), lambda t: t.cpu(),
                    lambda t: t.tolist(), lambda t: t.data_ptr(), lambda t: repr(t), lambda t: t + 1, lambda t: t[1:3],
                    lambda t: torch.cat([t, t]), lambda t: t.contiguous(), lambda t: t.plain(), lambda t: t.materialize()):
        log.clear()
        t = make()
        observe(t)
        assert log and log[0] == "resolve" and log.count("resolve") == 1, observe
        assert t.shape == (4,) and t.tolist() == [0, 5, 10, 15] and log.count("fill") == 1
    for fn in (copy.deepcopy, lambda t: pickle.loads(pickle.dumps(t)), lambda t: t.clone()):
        out = fn(make())
        assert type(out) is torch.Tensor and out.tolist() == [0, 5, 10, 15]
    # the DLPack PROTOCOL asks the object (LazyTensor.__dlpack__ settles first); the legacy capsule call unwraps the tensor in
    # C++ without any hook and exports the placeholder: the documented hazard (SC_DEFER_ISECT=0 SC_LAZY_IDS=0)
    from torch.utils import dlpack
    log.clear()
    assert torch.from_dlpack(make()).tolist() == [0, 5, 10, 15] and log == ["resolve", "fill"]
    assert dlpack.from_dlpack(dlpack.to_dlpack(make())).shape == (0,)
    assert dlpack.from_dlpack(dlpack.to_dlpack(make().materialize())).tolist() == [0, 5, 10, 15]
    t = make(fill=False)                  # no contents to produce (flatten_ids: the sort has written them)
    assert type(t.plain()) is torch.Tensor and t.plain().shape == (4,) and t.plain().data_ptr() == cap.data_ptr()


def test_lazy_tensor_whose_resolver_fails_keeps_failing():
    """a resolver (or fill) that raises -- more than 2^31 - 1 intersections, an allocation or HIP error in the
    exact-size relaunch -- must raise on EVERY observation; the zero-length placeholder must never be handed out as if it
    were the settled list (a caller that catches the error and rasterizes again would otherwise get a blank frame)."""
    import pytest
    import torch
    from street_crafter_amd.lazy import LazyTensor
    from street_crafter_amd.rendering import _PendingIsect
    calls = []

    def settle():
        calls.append(1)
        raise RuntimeError("isect_tiles: too many intersections")

    pend = _PendingIsect(settle)                       # what isect_tiles' deferred tensors carry
    z = LazyTensor(torch.zeros(0, dtype=torch.int32), None, pend.resolve)
    for observe in (lambda t: t.shape, lambda t: t.numel(), lambda t: t.plain(), lambda t: t + 1, lambda t: t.shape):
        with pytest.raises(RuntimeError, match="too many intersections"):
            observe(z)
        assert not z.is_resolved
    assert len(calls) == 1                             # the settle itself ran once; its error is what every observation raises
    assert z.dtype == torch.int32 and z.ndim == 1      # (shape-free questions still need no settle)
    # a fill that fails stays pending as well
    n = []

    def fill(t):
        n.append(1)
        raise ValueError("rebuild failed")

    y = LazyTensor(torch.zeros(3, dtype=torch.int64), fill)
    for _ in range(2):
        with pytest.raises(ValueError):
            y.tolist()
    assert len(n) == 2 and not y.is_materialized and y.shape == (3,)


def test_compiled_binding_layer_loads_and_refuses_cpu_tensors(lib):
    """csrc/binding.cpp -> lib/_sc_fast.so: the default host path.  It must be the same build as the library (version
    string), export one entry per hot operator call, and refuse tensors that are not on a HIP device before anything
    is launched; set_fast_binding() switches to the ctypes table and back; the diagnostic build never gets it."""
    import torch
    from street_crafter_amd import _lib
    fast = _lib.fast()
    assert fast is not None and os.path.exists(_lib.FAST_PATH)
    assert fast.abi_version().encode() == lib.sc_version()
    # both binaries carry the digest of the header they were compiled against (a stale binding next to a rebuilt library with
    # changed signatures must refuse to load)
    from street_crafter_amd import build as _b
    assert fast.abi_version().endswith("abi:" + _b.abi_hash()) and lib.sc_version().decode().endswith("abi:" + _b.abi_hash())
    for name in ("projection_fwd", "projection_bwd", "isect_bin_count", "isect_bin_sort", "wait_i64", "sh_fwd", "sh_bwd",
                 "rasterize_fwd", "rasterize_bwd", "projection_sh_fwd", "rasterize_fwd_packed", "frame_composite_u8"):
        assert callable(getattr(fast, name)), name
    n = 8
    z = torch.zeros
    with pytest.raises(RuntimeError, match="HIP device"):
        fast.projection_fwd(z(n, 3), z(n, 4), z(n, 3), torch.eye(4)[None], torch.eye(3)[None], 64, 64, 0.3, 0.01, 1e10,
                            0.0, True, 0)
    with pytest.raises(RuntimeError, match="HIP device"):
        fast.sh_fwd(1, z(1, n, 3), z(1, n, 4, 3), None, 0)
    with pytest.raises(RuntimeError, match="HIP device"):
        fast.isect_bin_count(z(1, n, 2), z(1, n, dtype=torch.int32), z(1, n), 16, 4, 4, None, None, None, False, 0, 1, 0)
    with pytest.raises(RuntimeError, match="HIP device"):
        fast.rasterize_fwd(z(1, n, 2), z(1, n, 3), z(1, n, 3), z(1, n), None, None, 64, 64, 16,
                           z(1, 4, 4, dtype=torch.int32), z(0, dtype=torch.int32), False, None, None, 0)
    # host-side wait through the binding: same contract as sc_wait_i64 (0 = arrived, 1 = timed out)
    word = torch.zeros(1, dtype=torch.int64)
    word[0] = 5
    assert fast.wait_i64(word.data_ptr(), 5, 1000) == 0 and fast.wait_i64(word.data_ptr(), 6, 2000) == 1
    prev = _lib.set_fast_binding(False)
    try:
        assert prev is True and _lib.fast() is None
    finally:
        _lib.set_fast_binding(prev)
    assert _lib.fast() is fast


def test_setup_py_names_the_three_packages():
    """setup.py (the editable install that replaces the reference's `pip install gsplat` / simple-knn steps) lists the
    package names the reference imports; nothing is installed or built by this test."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "setup.py"), "--name", "--version"], capture_output=True,
                         text=True, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-500:]
    assert out.stdout.split()[:1] == ["street_crafter_amd"]
    from setuptools import find_packages
    pk = set(find_packages(where=ROOT, include=["street_crafter_amd", "street_crafter_amd.*", "gsplat", "gsplat.*",
                                                "simple_knn", "simple_knn.*"]))
    assert {"gsplat", "simple_knn", "street_crafter_amd"} <= pk


def test_switches_come_from_the_environment_and_state_is_one_object():
    """the operators' A/B switches live in ONE object initialised from documented environment variables
    (a user who feeds flatten_ids to foreign C++ turns the placeholders off without touching code), and everything the
    operators remember between calls lives in ONE state object that reset_state() empties."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from street_crafter_amd import rendering as R; s = R._SWITCH; "
            "print(s.defer_isect, s.lazy_ids, s.planar_out, s.tile_order, s.view_slots, s.packed_records)" % ROOT)
    env = dict(os.environ)
    for k in ("SC_DEFER_ISECT", "SC_LAZY_IDS", "SC_PLANAR_OUTPUT", "SC_TILE_ORDER", "SC_VIEW_SLOTS", "SC_PACKED_RECORDS"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True, env=env).stdout.split()
    assert out == ["True"] * 6
    env.update(SC_DEFER_ISECT="0", SC_LAZY_IDS="0", SC_PLANAR_OUTPUT="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True, env=env).stdout.split()
    assert out == ["False", "False", "False", "True", "True", "True"]
    from street_crafter_amd import rendering
    assert rendering.set_deferred_isect(False) is True and rendering._SWITCH.defer_isect is False
    assert rendering.set_deferred_isect(True) is False
    st = rendering._STATE
    st.prediction[(5, 1, 10, 16, 4, 3)] = (1, 1, 1)
    st.last_meta[(6, 1, 10, 16, 4, 3)] = (1, 1, 1)
    assert rendering.reset_state(5)["predictions"] == 1 and (6, 1, 10, 16, 4, 3) in st.last_meta
    assert rendering.reset_state()["last_meta"] == 1 and not st.prediction and not st.last_meta
    # no module-level table is left behind the object
    for name in ("_BIN_PREDICTION", "_BIN_HISTORY", "_BIN_LAST_META", "_TILE_WORK", "_VIEW_REGISTRY", "_DEFER_ISECT"):
        assert not hasattr(rendering, name)

