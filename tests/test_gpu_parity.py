"""GPU parity tests: the HIP path (through the gsplat / simple_knn drop-in names, i.e. through the
C ABI) against the CPU oracle and the committed golden vectors.

Bars (BASELINE.json north_star): integer outputs -- radii, tiles_per_gauss, isect_ids (tile and
depth bits), flatten_ids, isect_offsets -- BIT-EXACT; projection / SH floats bit-exact too (the
kernels are compiled without FMA contraction in the oracle's op order); blended pixels within
1e-4 abs for RGB/alpha (the blend uses v_exp_f32 and FMA), depth channel within 1e-4 * far-range
relative.  A pixel whose alpha or transmittance lies within 2e-5 (relative) of a hard threshold
(1/255, 1e-4) may legitimately flip on a 1-ulp exp difference; the oracle flags those pixels
("unstable") and they are excluded, their count is asserted to stay < 0.5 %.
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gsplat_oracle as O          # noqa: E402  (checker only)
from oracle import gsplat_torch as OT          # noqa: E402
from oracle import knn_oracle as KO            # noqa: E402
from street_crafter_amd.scenes import make_camera, make_edge_case_scene, make_scene  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    # fails loudly (ImportError) if the HIP library is missing on a GPU box
    from street_crafter_amd import _lib
    _lib.load()
    import gsplat.rendering as R
    return R


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(DEV)


def _np(t):
    return t.detach().cpu().numpy()


def _project(R, means, quats, scales, viewmat, K, w, h, near, far):
    return R.fully_fused_projection(_t(means), None, _t(quats), _t(scales), _t(viewmat)[None], _t(K)[None],
                                    w, h, packed=False, near_plane=near, far_plane=far,
                                    calc_compensations=True)


# ---------------------------------------------------------------------------------------------
def test_native_library_is_loaded(ops):
    from street_crafter_amd import _lib
    lib = _lib.load()
    assert b"gfx950" in lib.sc_target_arch()
    maps = open("/proc/self/maps").read()
    assert "libstreet_crafter_hip.so" in maps
    assert "gfx95" in torch.cuda.get_device_properties(0).gcnArchName


def test_projection_bit_exact_golden(ops, golden_dir):
    g = _load(golden_dir, "proj_small.npz")
    radii, m2, d, con, comp = _project(ops, g["means"], g["quats"], g["scales"], g["viewmat"], g["K"],
                                       int(g["width"]), int(g["height"]), float(g["near"]), float(g["far"]))
    np.testing.assert_array_equal(_np(radii)[0], g["radii"])
    np.testing.assert_array_equal(_np(m2)[0].view(np.uint32), g["means2d"].view(np.uint32))
    np.testing.assert_array_equal(_np(d)[0].view(np.uint32), g["depths"].view(np.uint32))
    np.testing.assert_array_equal(_np(con)[0].view(np.uint32), g["conics"].view(np.uint32))
    np.testing.assert_array_equal(_np(comp)[0].view(np.uint32), g["compensations"].view(np.uint32))


def test_projection_bit_exact_multi_camera(ops):
    sc = make_scene(20000, seed=9)
    cams = [make_camera(640, 400, 600.0, 600.0, yaw=0.1 * i, shift=(0.3 * i, 0.0, -0.5 * i)) for i in range(3)]
    V = torch.stack([c.viewmat for c in cams])
    K = torch.stack([c.K for c in cams])
    out = ops.fully_fused_projection(sc.means.to(DEV), None, sc.quats.to(DEV), sc.scales.to(DEV), V.to(DEV),
                                     K.to(DEV), 640, 400, near_plane=0.001, far_plane=1000.0)
    assert out[4] is None
    for i, c in enumerate(cams):
        r, m2, d, con, _ = O.fully_fused_projection(sc.means.numpy(), sc.quats.numpy(), sc.scales.numpy(),
                                                    c.viewmat.numpy(), c.K.numpy(), 640, 400,
                                                    near_plane=0.001, far_plane=1000.0)
        np.testing.assert_array_equal(_np(out[0])[i], r)
        np.testing.assert_array_equal(_np(out[1])[i].view(np.uint32), m2.view(np.uint32))
        np.testing.assert_array_equal(_np(out[2])[i].view(np.uint32), d.view(np.uint32))
        np.testing.assert_array_equal(_np(out[3])[i].view(np.uint32), con.view(np.uint32))


@pytest.mark.parametrize("mode", ["radix", "bin"])
def test_isect_bit_exact_golden(ops, golden_dir, mode):
    from street_crafter_amd import rendering
    g = _load(golden_dir, "pipeline_small.npz")
    prev = rendering.set_isect_mode(mode)
    try:
        tpg, ids, fids = ops.isect_tiles(_t(g["means2d"])[None], _t(g["radii"], torch.int32)[None],
                                         _t(g["depths"])[None], 16, 8, 6, packed=False, n_cameras=1)
        off = ops.isect_offset_encode(ids, 1, 8, 6)
    finally:
        rendering.set_isect_mode(prev)
    np.testing.assert_array_equal(_np(tpg)[0], g["tiles_per_gauss"])
    np.testing.assert_array_equal(_np(ids), g["isect_ids"])
    np.testing.assert_array_equal(_np(fids), g["flatten_ids"])
    np.testing.assert_array_equal(_np(off), g["isect_offsets"])


def test_isect_ids_are_lazy_on_the_bucketed_route_and_exact_when_read(ops, golden_dir):
    """isect_tiles returns isect_ids as gsplat does, but on the reference's path nothing reads them
    (renderer.py:253 hands them to isect_offset_encode, whose result the bucketed route already has): the 8 B x I
    array is a LazyTensor, written by one kernel on first use.  Metadata and isect_offset_encode do not trigger
    the fill; any read does, and gives exactly the keys the sort used to emit (and the oracle's)."""
    from street_crafter_amd import rendering
    from street_crafter_amd.lazy import LazyTensor
    g = _load(golden_dir, "pipeline_small.npz")
    a = (_t(g["means2d"])[None], _t(g["radii"], torch.int32)[None], _t(g["depths"])[None], 16, 8, 6)
    tpg, ids, fids = ops.isect_tiles(*a, packed=False, n_cameras=1)
    assert isinstance(ids, LazyTensor) and not ids.is_materialized
    assert ids.shape == (g["isect_ids"].shape[0],) and ids.dtype == torch.int64 and ids.is_cuda and ids.numel() == fids.numel()
    off = ops.isect_offset_encode(ids, 1, 8, 6)
    assert not ids.is_materialized                                   # the caller's sequence never pays for the keys
    np.testing.assert_array_equal(_np(off), g["isect_offsets"])
    np.testing.assert_array_equal(_np(ids), g["isect_ids"])          # first read: filled
    assert ids.is_materialized
    assert int((ids >> 32).max()) == int(g["isect_ids"].max() >> 32) and bool(torch.equal(ids[3:9], _t(g["isect_ids"][3:9], torch.int64)))
    # a second, independent offsets computation from the (now real) keys gives the same answer
    ids2 = ids.clone()
    np.testing.assert_array_equal(_np(ops.isect_offset_encode(ids2, 1, 8, 6)), g["isect_offsets"])
    # filled from another stream: ordered after the producing stream
    _, ids3, _ = ops.isect_tiles(*a, packed=False, n_cameras=1)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        host = ids3.cpu()
    np.testing.assert_array_equal(host.numpy(), g["isect_ids"])
    # eager mode: the sort writes the keys itself
    prev = rendering.set_lazy_isect_ids(False)
    try:
        _, ids4, _ = ops.isect_tiles(*a, packed=False, n_cameras=1)
        assert not isinstance(ids4, LazyTensor)
        np.testing.assert_array_equal(_np(ids4), g["isect_ids"])
    finally:
        rendering.set_lazy_isect_ids(prev)


def test_isect_tiles_defers_its_host_wait_to_the_first_observation(ops):
    """Round 3 (street_crafter_amd/lazy.py, rendering._PendingIsect): from the second call of a frame shape on,
    isect_tiles enqueues scatter + sort with predicted sizes and returns WITHOUT waiting for the frame's counts;
    flatten_ids / isect_ids settle their length on first observation (rasterize_to_pixels on the reference's path).
    Same tensors as the eager form, bit for bit: in the good case, after a mis-prediction (exact relaunch inside the
    settle), when the outputs are never looked at, and when two calls follow each other unobserved."""
    from street_crafter_amd import rendering
    from street_crafter_amd.lazy import LazyTensor
    from street_crafter_amd.scenes import make_camera, make_scene
    sc = make_scene(30_000, seed=77).to(DEV)
    W, H, tw, th = 640, 416, 40, 26
    key = (torch.cuda.current_device(), 1, sc.n, 16, tw, th)

    def project(yaw):
        cam = make_camera(W, H, 680.0, 680.0, yaw=yaw).to(DEV)
        with torch.no_grad():
            r, m2, d, con, comp = ops.fully_fused_projection(sc.means, None, sc.quats, sc.scales, cam.viewmat[None], cam.K[None],
                                                              W, H, near_plane=0.001, far_plane=1000.0, calc_compensations=True)
        return r, m2, d, con, comp

    def eager(a):
        prev = rendering.set_deferred_isect(False)
        try:
            tpg, ids, fids = ops.isect_tiles(a[1], a[0], a[2], 16, tw, th, n_cameras=1)
            return tpg, ids.clone(), fids.clone(), ops.isect_offset_encode(ids, 1, tw, th).clone()
        finally:
            rendering.set_deferred_isect(prev)

    for d_ in (rendering._STATE.prediction, rendering._STATE.history, rendering._STATE.last_meta):
        d_.pop(key, None)
    a0, a1 = project(0.0), project(0.3)
    e0, e1 = eager(a0), eager(a1)                                        # (also teaches the prediction)
    assert e0[2].numel() != e1[2].numel()
    # 1. deferred, good prediction: nothing is settled by the call, by metadata that does not need the length, or by
    #    isect_offset_encode; shape / contents settle it
    stats = dict(rendering._STATE.stats)
    tpg, ids, fids = ops.isect_tiles(a0[1], a0[0], a0[2], 16, tw, th, n_cameras=1)
    assert type(fids) is LazyTensor and type(ids) is LazyTensor and not fids.is_resolved and not ids.is_resolved
    assert fids.dtype == torch.int32 and ids.dtype == torch.int64 and fids.is_cuda and fids.ndim == 1
    off = ops.isect_offset_encode(ids, 1, tw, th)
    assert not fids.is_resolved and not ids.is_resolved and rendering._STATE.stats["calls"] == stats["calls"]
    assert torch.equal(off, e0[3]) and torch.equal(tpg, e0[0])
    assert fids.shape == e0[2].shape and fids.is_resolved and ids.is_resolved and not ids.is_materialized
    assert rendering._STATE.stats["calls"] == stats["calls"] + 1 and rendering._STATE.stats["speculative_ok"] == stats["speculative_ok"] + 1
    assert ids.shape == e0[1].shape and torch.equal(fids, e0[2]) and torch.equal(ids, e0[1]) and ids.is_materialized
    assert torch.equal(ops.isect_offset_encode(ids, 1, tw, th), e0[3])
    # 2. the rasterizer settles it: image equal to the eager call's
    def image(a, off, fids):
        cols = torch.rand(1, sc.n, 3, device=DEV, generator=torch.Generator(DEV).manual_seed(5))
        return ops.rasterize_to_pixels(a[1], a[3], cols, sc.opacities[None, :, 0] * a[4], W, H, 16, off, fids)
    tpg, ids, fids = ops.isect_tiles(a1[1], a1[0], a1[2], 16, tw, th, n_cameras=1)
    off = ops.isect_offset_encode(ids, 1, tw, th)
    assert not fids.is_resolved
    img = image(a1, off, fids)
    assert fids.is_resolved and torch.equal(fids, e1[2])
    ref = image(a1, e1[3], e1[2])
    assert torch.equal(img[0], ref[0]) and torch.equal(img[1], ref[1])
    # 3. a prediction that is too small: the settle relaunches with exact sizes; one that is generous: no relaunch
    for pred, relaunch in (((16, 16, 16), 1), ((1 << 26, 1 << 26, 3000), 0)):
        rendering._STATE.prediction[key] = pred
        stats = dict(rendering._STATE.stats)
        tpg, ids, fids = ops.isect_tiles(a0[1], a0[0], a0[2], 16, tw, th, n_cameras=1)
        assert not fids.is_resolved
        assert torch.equal(fids, e0[2]) and torch.equal(ids, e0[1]) and torch.equal(tpg, e0[0])
        assert rendering._STATE.stats["exact_relaunch"] == stats["exact_relaunch"] + relaunch
    # 4. two calls in a row, the first never looked at: the second call settles it (the pinned slot is shared), and the
    #    first's tensors are still right afterwards; outputs dropped unobserved leave nothing behind
    t0 = ops.isect_tiles(a0[1], a0[0], a0[2], 16, tw, th, n_cameras=1)
    assert not t0[2].is_resolved
    t1 = ops.isect_tiles(a1[1], a1[0], a1[2], 16, tw, th, n_cameras=1)
    assert t0[2].is_resolved and not t1[2].is_resolved
    assert torch.equal(t1[2], e1[2]) and torch.equal(t0[2], e0[2]) and torch.equal(t0[1], e0[1]) and torch.equal(t1[1], e1[1])
    ops.isect_tiles(a0[1], a0[0], a0[2], 16, tw, th, n_cameras=1)          # dropped at once
    t2 = ops.isect_tiles(a1[1], a1[0], a1[2], 16, tw, th, n_cameras=1)
    assert torch.equal(t2[2], e1[2])
    # 5. only isect_ids kept and read
    ids_only = ops.isect_tiles(a0[1], a0[0], a0[2], 16, tw, th, n_cameras=1)[1]
    assert torch.equal(ids_only, e0[1])
    # 6. settled from another stream than the producing one
    t3 = ops.isect_tiles(a1[1], a1[0], a1[2], 16, tw, th, n_cameras=1)
    with torch.cuda.stream(torch.cuda.Stream()):
        host_f, host_i = t3[2].cpu(), t3[1].cpu()
    assert torch.equal(host_f, e1[2].cpu()) and torch.equal(host_i, e1[1].cpu())


def test_isect_unsorted_and_empty(ops, golden_dir):
    g = _load(golden_dir, "pipeline_small.npz")
    m2, r, d = _t(g["means2d"])[None], _t(g["radii"], torch.int32)[None], _t(g["depths"])[None]
    _, ids, fids = ops.isect_tiles(m2, r, d, 16, 8, 6, sort=False)
    _, e_ids, e_f = O.isect_tiles(g["means2d"][None], g["radii"][None], g["depths"][None], 16, 8, 6, sort=False)
    np.testing.assert_array_equal(_np(ids), e_ids)
    np.testing.assert_array_equal(_np(fids), e_f)
    # everything culled: I == 0 -> empty lists, all-zero offsets, black image
    tpg, ids0, f0 = ops.isect_tiles(m2, torch.zeros_like(r), d, 16, 8, 6)
    assert ids0.numel() == 0 and f0.numel() == 0 and int(tpg.sum()) == 0
    off0 = ops.isect_offset_encode(ids0, 1, 8, 6)
    assert int(off0.abs().sum()) == 0
    rc, ra = ops.rasterize_to_pixels(m2, _t(g["conics"])[None], _t(g["colors"])[None], _t(g["opacities"])[None],
                                     128, 96, 16, off0, f0)
    assert float(rc.abs().sum()) == 0.0 and float(ra.abs().sum()) == 0.0


@pytest.mark.parametrize("n_cams", [1, 3])
def test_isect_multi_camera_vs_oracle(ops, n_cams):
    sc = make_scene(6000, seed=21, z_range=(1.0, 40.0), scale_range=(0.01, 0.5))
    W, H = 200, 120                    # ragged: 13 x 8 tiles, last column/row partial
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    m2l, rl, dl = [], [], []
    for i in range(n_cams):
        c = make_camera(W, H, 220.0, 220.0, yaw=0.05 * i)
        r, m2, d, _, _ = O.fully_fused_projection(sc.means.numpy(), sc.quats.numpy(), sc.scales.numpy(),
                                                  c.viewmat.numpy(), c.K.numpy(), W, H, near_plane=0.001,
                                                  far_plane=1000.0)
        m2l.append(m2); rl.append(r); dl.append(d)
    m2, r, d = np.stack(m2l), np.stack(rl), np.stack(dl)
    e_tpg, e_ids, e_f = O.isect_tiles(m2, r, d, 16, tw, th, n_cameras=n_cams)
    e_off = O.isect_offset_encode(e_ids, n_cams, tw, th)
    tpg, ids, fids = ops.isect_tiles(_t(m2), _t(r, torch.int32), _t(d), 16, tw, th, n_cameras=n_cams)
    off = ops.isect_offset_encode(ids, n_cams, tw, th)
    np.testing.assert_array_equal(_np(tpg), e_tpg)
    np.testing.assert_array_equal(_np(ids), e_ids)
    np.testing.assert_array_equal(_np(fids), e_f)
    np.testing.assert_array_equal(_np(off), e_off)


@pytest.mark.parametrize("mode", ["radix", "bin"])
@pytest.mark.parametrize("n_depths", [1, 3, 4000])
def test_isect_depth_ties_keep_index_order(ops, mode, n_depths):
    """Equal depth keys inside a tile must stay in ascending flat-index order (stable sort).
    n_depths = 1 / 3: thousands of identical keys per tile (forces the bin path's radix fallback,
    and its tie re-sort); 4000: mixed.  Also a negative-depth block (bit-pattern order)."""
    from street_crafter_amd import rendering
    rng = np.random.default_rng(n_depths)
    N, W, H = 6000, 96, 64
    tw, th = 6, 4
    m2 = rng.uniform(-8, 104, size=(1, N, 2)).astype(np.float32)
    r = rng.integers(0, 40, size=(1, N)).astype(np.int32)
    levels = rng.uniform(1.0, 50.0, size=n_depths).astype(np.float32)
    d = levels[rng.integers(0, n_depths, size=(1, N))]
    if n_depths == 4000:
        d[0, :500] *= -1.0
    e_tpg, e_ids, e_f = O.isect_tiles(m2, r, d, 16, tw, th)
    prev = rendering.set_isect_mode(mode)
    try:
        tpg, ids, fids = ops.isect_tiles(_t(m2), _t(r, torch.int32), _t(d), 16, tw, th)
        off = ops.isect_offset_encode(ids, 1, tw, th)
    finally:
        rendering.set_isect_mode(prev)
    np.testing.assert_array_equal(_np(tpg), e_tpg)
    np.testing.assert_array_equal(_np(ids), e_ids)
    np.testing.assert_array_equal(_np(fids), e_f)
    np.testing.assert_array_equal(_np(off), O.isect_offset_encode(e_ids, 1, tw, th))


def _isect_both_routes(ops, m2, r, d, tw, th, C=1):
    """-> {"radix": ..., "bin": ... (isect_ids lazy, the default), "bin_eager": ... (keys written by the sort)}"""
    from street_crafter_amd import rendering
    out = {}
    for mode in ("radix", "bin", "bin_eager"):
        prev = rendering.set_isect_mode(mode.split("_")[0])
        prev_lazy = rendering.set_lazy_isect_ids(mode != "bin_eager")
        try:
            tpg, ids, fids = ops.isect_tiles(m2, r, d, 16, tw, th, n_cameras=C)
            out[mode] = (tpg, ids, fids, ops.isect_offset_encode(ids, C, tw, th))
        finally:
            rendering.set_isect_mode(prev)
            rendering.set_lazy_isect_ids(prev_lazy)
    return out


@pytest.mark.parametrize("n_small", [0, 300_000])
@pytest.mark.parametrize("edge", ["left", "right", "top", "bottom"])
def test_isect_bin_big_splats_clipped_to_one_column_or_row_of_super_tiles(ops, edge, n_small):
    """Big splats that the frame clips to ONE column (or row) of 2x2 super-tiles, more than 32 super-tiles long: the
    wave-walked rectangles of the record scatter with width 1 (found by the fuzz tool: the 32-bit reciprocal of a
    width of 1 wrapped to 0 and every cell went to the first row) -- through the small-input and the large-input
    scatter kernels, against the reference-shaped route, bit for bit."""
    tw, th = (44, 140) if edge in ("left", "right") else (140, 44)        # 70 super-tiles along the long side
    w, h = tw * 16, th * 16
    g = torch.Generator(device=DEV).manual_seed(7)
    n_big = 400
    r_big = 2000
    along = torch.rand(n_big, device=DEV, generator=g) * (h if edge in ("left", "right") else w)
    off = {"left": -r_big + 20.0, "right": w + r_big - 20.0, "top": -r_big + 20.0, "bottom": h + r_big - 20.0}[edge]
    jitter = torch.rand(n_big, device=DEV, generator=g) * 8.0 * (1 if edge in ("left", "top") else -1)
    if edge in ("left", "right"):
        m_big = torch.stack([off + jitter, along], dim=-1)
    else:
        m_big = torch.stack([along, off + jitter], dim=-1)
    m_small = torch.rand(n_small, 2, device=DEV, generator=g) * torch.tensor([w, h], device=DEV)
    m2 = torch.cat([m_big, m_small])[None].contiguous()
    r = torch.cat([torch.full((n_big,), r_big, dtype=torch.int32, device=DEV),
                   torch.randint(1, 12, (n_small,), device=DEV, generator=g, dtype=torch.int64).to(torch.int32)])[None].contiguous()
    d = (torch.rand(1, n_big + n_small, device=DEV, generator=g) * 50.0 + 1.0).contiguous()
    out = _isect_both_routes(ops, m2, r, d, tw, th)
    assert out["radix"][2].numel() > n_big * 2 * 100
    for mode in ("bin", "bin_eager"):
        for a, b in zip(out["radix"], out[mode]):
            assert torch.equal(a, b), mode


@pytest.mark.parametrize("shape", ["uniform", "cluster_and_outliers", "equal_depths", "heavy_bin", "two_heavy_bins"])
def test_isect_bin_oversized_super_tiles_are_split_not_abandoned(ops, shape):
    """A super-tile with more records than one workgroup sorts in LDS (3584) used to send the WHOLE frame down
    the radix route (11x slower).  Now only that bucket is cut into depth ranges (big_split_kernel)
    and the ranges are sorted like ordinary buckets; a range that cannot be cut by depth (thousands of records
    within 1/1024 of the key range: 'heavy') takes an exact quadratic path.  Every shape must reproduce the
    oracle bit for bit and stay on the bucketed route."""
    from street_crafter_amd import rendering
    rng = np.random.default_rng(5)
    N = 30000
    m2 = rng.uniform(0, 32, size=(1, N, 2)).astype(np.float32)          # everything inside one super-tile ...
    m2[0, :3000] = rng.uniform(0, 96, size=(3000, 2))                    # ... and some ordinary neighbours
    r = rng.integers(1, 6, size=(1, N)).astype(np.int32)
    d = rng.uniform(1.0, 50.0, size=(1, N)).astype(np.float32)
    if shape == "cluster_and_outliers":       # a facade (narrow cluster) + sky (far outliers)
        d[0, :24000] = rng.uniform(30.0, 30.1, size=24000)
        d[0, 24000:24100] = rng.uniform(200.0, 900.0, size=100)
    elif shape == "equal_depths":             # ties: only the flat id orders them
        d[0, :20000] = 7.5
    elif shape == "heavy_bin":                # 12000 records inside one histogram bin of the split
        d[0, :12000] = np.float32(10.0) + rng.integers(0, 3, size=12000).astype(np.float32) * np.float32(1e-6)
        d[0, 12000:12050] = rng.uniform(500.0, 1000.0, size=50)
    elif shape == "two_heavy_bins":
        d[0, :9000] = 10.0
        d[0, 9000:18000] = np.float32(10.000001)
        d[0, 18000:18010] = 900.0
    e_tpg, e_ids, e_f = O.isect_tiles(m2, r, d, 16, 6, 6)
    key = (torch.cuda.current_device(), 1, N, 16, 6, 6)
    rendering._STATE.last_meta.pop(key, None)
    rendering._STATE.prediction.pop(key, None)
    for _ in range(2):                        # second call: sizes predicted from the first
        tpg, ids, fids = ops.isect_tiles(_t(m2), _t(r, torch.int32), _t(d), 16, 6, 6)
        off = ops.isect_offset_encode(ids, 1, 6, 6)
        np.testing.assert_array_equal(_np(tpg), e_tpg)
        np.testing.assert_array_equal(_np(ids), e_ids)
        np.testing.assert_array_equal(_np(fids), e_f)
        np.testing.assert_array_equal(_np(off), O.isect_offset_encode(e_ids, 1, 6, 6))
    assert rendering._STATE.last_meta[key][2] > 20000           # the bucketed route ran, with a 20k+ bucket


@pytest.mark.parametrize("n_in_bucket", [120, 1000, 1024, 1030, 1290, 3500])
def test_isect_bin_small_and_large_bucket_sorts_agree_with_the_oracle(ops, n_in_bucket):
    """Frames whose largest super-tile bucket fits 1024 records are sorted by the 128-thread form of the bucket sort
    (super_sort_kernel<128, 8>), larger ones by the 512-thread form: both sides of the switch (the capacity is the
    largest bucket rounded up to 256: 1024 / 1280), depth ties included, first with exact and then with predicted
    sizes (predicted = 12.5 % over: 1000 records then provision 1280, the other form again)."""
    from street_crafter_amd import rendering
    rng = np.random.default_rng(n_in_bucket)
    N = n_in_bucket + 600
    m2 = rng.uniform(40, 56, size=(1, N, 2)).astype(np.float32)           # one super-tile (tiles 2..3) holds the bucket ...
    m2[0, n_in_bucket:] = rng.uniform(64, 192, size=(600, 2))             # ... the rest is spread over its neighbours
    r = np.ones((1, N), dtype=np.int32)
    r[0, n_in_bucket:] = rng.integers(1, 9, size=600)
    d = rng.uniform(1.0, 50.0, size=(1, N)).astype(np.float32)
    d[0, : n_in_bucket // 3] = 12.5                                       # a third of the bucket ties on depth
    e_tpg, e_ids, e_f = O.isect_tiles(m2, r, d, 16, 12, 12)
    key = (torch.cuda.current_device(), 1, N, 16, 12, 12)
    rendering._STATE.last_meta.pop(key, None)
    rendering._STATE.prediction.pop(key, None)
    rendering._STATE.history.pop(key, None)
    for _ in range(2):
        tpg, ids, fids = ops.isect_tiles(_t(m2), _t(r, torch.int32), _t(d), 16, 12, 12)
        np.testing.assert_array_equal(_np(tpg), e_tpg)
        np.testing.assert_array_equal(_np(ids), e_ids)
        np.testing.assert_array_equal(_np(fids), e_f)
    assert rendering._STATE.last_meta[key][2] >= n_in_bucket                # the bucketed route ran, bucket as built


def test_isect_bin_street_scene_matches_the_radix_route(ops):
    """The street-shaped scene (dense horizon band: hundreds of oversized super-tiles; sky splats hundreds of
    pixels wide: rectangles of hundreds of tiles) through both routes at full resolution: bit-identical."""
    from street_crafter_amd import rendering
    from street_crafter_amd.scenes import make_street_scene
    for scene in make_street_scene(400_000, n_sky=12_000, seed=11):
        cam = make_camera()
        with torch.no_grad():
            r, m2, d, _, _ = ops.fully_fused_projection(scene.means.to(DEV), None, scene.quats.to(DEV), scene.scales.to(DEV),
                                                        cam.viewmat.to(DEV)[None], cam.K.to(DEV)[None], 1920, 1280,
                                                        near_plane=0.001, far_plane=1000.0)
        key = (torch.cuda.current_device(), 1, scene.n, 16, 120, 80)
        rendering._STATE.last_meta.pop(key, None)
        out = _isect_both_routes(ops, m2, r, d, 120, 80)
        for a, b, c in zip(out["bin"], out["radix"], out["bin_eager"]):
            assert torch.equal(a, b) and torch.equal(c, b)
        assert key in rendering._STATE.last_meta                 # the bucketed route took it
    assert rendering._STATE.last_meta[(torch.cuda.current_device(), 1, 400_000, 16, 120, 80)][2] > 3584


def test_isect_bin_4k_frame_stays_on_the_bucketed_route(ops):
    """3840 x 2160 = 240 x 135 = 32400 tiles: above round 1's 16384-tile limit (whole frame to the radix route).
    The count pass's LDS grids no longer fit and it adds straight to the global grids; everything else is unchanged.
    Bit-identical to the radix route."""
    from street_crafter_amd import rendering
    W, H = 3840, 2160
    cam = make_camera(W, H, 4100.0, 4100.0)
    sc = make_scene(300_000, seed=41, z_range=(2.0, 60.0), scale_range=(0.005, 0.2))
    with torch.no_grad():
        r, m2, d, _, _ = ops.fully_fused_projection(sc.means.to(DEV), None, sc.quats.to(DEV), sc.scales.to(DEV),
                                                    cam.viewmat.to(DEV)[None], cam.K.to(DEV)[None], W, H,
                                                    near_plane=0.001, far_plane=1000.0)
    key = (torch.cuda.current_device(), 1, sc.n, 16, 240, 135)
    rendering._STATE.last_meta.pop(key, None)
    out = _isect_both_routes(ops, m2, r, d, 240, 135)
    for a, b in zip(out["bin"], out["radix"]):
        assert torch.equal(a, b)
    assert key in rendering._STATE.last_meta and int(out["bin"][1].numel()) > 1_000_000


@pytest.mark.parametrize("how", ["super_just_below", "super_rounding_window", "isects_too_small", "records_too_small",
                                 "all_too_small", "generous", "provisioned_for_split"])
def test_isect_bin_mispredicted_capacities_retry_exactly_once(ops, how):
    """The scatter + sort are enqueued with capacities predicted from the previous frame and verified on the
    device; a wrong prediction must end in ONE full run with exact sizes (a launch that passed the
    device check while the host decided to retry ran the scatter twice on advanced cursors).  Every
    mis-prediction shape, including the one inside the old 256-rounding window, must reproduce the radix
    route bit for bit."""
    from street_crafter_amd import rendering
    cam = make_camera(640, 416, 700.0, 700.0)
    sc = make_scene(60_000, seed=77, z_range=(1.0, 30.0), scale_range=(0.01, 0.2))
    with torch.no_grad():
        r, m2, d, _, _ = ops.fully_fused_projection(sc.means.to(DEV), None, sc.quats.to(DEV), sc.scales.to(DEV),
                                                    cam.viewmat.to(DEV)[None], cam.K.to(DEV)[None], 640, 416,
                                                    near_plane=0.001, far_plane=1000.0)
    tw, th = 40, 26
    prev = rendering.set_isect_mode("radix")
    try:
        e_tpg, e_ids, e_f = ops.isect_tiles(m2, r, d, 16, tw, th, n_cameras=1)
        e_off = ops.isect_offset_encode(e_ids, 1, tw, th)
    finally:
        rendering.set_isect_mode(prev)
    key = (torch.cuda.current_device(), 1, sc.n, 16, tw, th)
    rendering._STATE.prediction.pop(key, None)
    ops.isect_tiles(m2, r, d, 16, tw, th, n_cameras=1)                   # learns the true sizes
    n_is, n_rec, max_super = rendering._STATE.last_meta[key]
    assert n_is == int(e_ids.numel()) and max_super > 300
    big = 1 << 26
    window = (max_super + 255) // 256 * 256 - 255            # smallest value that rounds up to the same multiple
    seeds = {
        "super_just_below": (big, big, max_super - 1),
        # old bug: host compared with the unrounded number, device with the value rounded up to 256
        "super_rounding_window": (big, big, min(window, max_super - 1)),
        "isects_too_small": (n_is - 1, big, big),
        "records_too_small": (big, max(1, n_rec // 2), big),
        "all_too_small": (1, 1, 1),
        "generous": (big, big, max_super + 100),
        "provisioned_for_split": (big, big, 200_000),        # the split kernels launch and find nothing to do
    }
    pred = seeds[how]
    if how == "super_rounding_window":
        assert pred[2] < max_super <= (pred[2] + 255) // 256 * 256 or max_super % 256 == 1
    for _ in range(2):                                                    # second pass: prediction learnt from the retry
        rendering._STATE.prediction[key] = pred
        tpg, ids, fids = ops.isect_tiles(m2, r, d, 16, tw, th, n_cameras=1)
        off = ops.isect_offset_encode(ids, 1, tw, th)
        assert torch.equal(tpg, e_tpg) and torch.equal(ids, e_ids) and torch.equal(fids, e_f) and torch.equal(off, e_off)
        pred = rendering._STATE.prediction[key]
    assert rendering._bin_launch_ran((10, 10, 10), 10, 10, 10) and not rendering._bin_launch_ran((10, 10, 10), 10, 10, 11)


@pytest.mark.parametrize("mode", ["radix", "bin"])
def test_isect_refuses_more_than_int32_intersections(ops, mode):
    """isect_offsets and the positions in flatten_ids are int32 (as in gsplat, which wraps around silently):
    a call that would produce more than 2^31 - 1 intersections fails loudly after the count pass, before
    anything is allocated for them -- on both routes."""
    from street_crafter_amd import rendering
    tw, th = 160, 74                       # 11 840 tiles
    n = 200_000                            # every splat covers the whole frame: 2.37e9 intersections
    m2 = torch.full((1, n, 2), 600.0, device=DEV)
    r = torch.full((1, n), 100_000, dtype=torch.int32, device=DEV)
    d = torch.rand(1, n, device=DEV) + 1.0
    prev = rendering.set_isect_mode(mode)
    try:
        with pytest.raises(RuntimeError, match="exceed the int32 range"):
            ops.isect_tiles(m2, r, d, 16, tw, th, n_cameras=1)
    finally:
        rendering.set_isect_mode(prev)
    assert torch.cuda.memory_allocated() < 2 << 30


def test_radix_sort_large_stable(ops):
    """4.2 M pairs with heavy key duplication vs torch.sort(stable=True) (same device)."""
    from street_crafter_amd import _lib
    lib = _lib.load()
    n = 4_200_003
    g = torch.Generator().manual_seed(1)
    keys = torch.randint(0, 1 << 20, (n,), generator=g, dtype=torch.int64)
    keys = (keys << 27) | torch.randint(0, 8, (n,), generator=g, dtype=torch.int64)
    vals = torch.arange(n, dtype=torch.int32)
    k, v = keys.to(DEV), vals.to(DEV)
    tk, tv = torch.empty_like(k), torch.empty_like(v)
    ws = torch.empty(lib.sc_radix_sort_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    _lib.check(lib.sc_radix_sort_pairs_u64_i32(k.data_ptr(), v.data_ptr(), tk.data_ptr(), tv.data_ptr(), n, 47,
                                               ws.data_ptr(), ws.numel(),
                                               torch.cuda.current_stream().cuda_stream), "sort")
    ek, ei = torch.sort(keys, stable=True)
    assert torch.equal(k.cpu(), ek)
    assert torch.equal(v.cpu(), vals[ei])


@pytest.mark.parametrize("deg", [0, 1, 2, 3, 4])
def test_sh_forward_bit_exact_and_reference(ops, golden_dir, deg):
    g = _load(golden_dir, "sh_eval_ref.npz")
    K = (deg + 1) ** 2
    dirs = (g["dirs"] * 3.25).astype(np.float32)             # un-normalised, like renderer.py:256
    coeffs = g["coeffs"][:, :K].astype(np.float32)
    masks = np.arange(dirs.shape[0]) % 5 != 0
    got = ops.spherical_harmonics(deg, _t(dirs)[None], _t(coeffs)[None], masks=torch.from_numpy(masks)[None].to(DEV))
    exp = O.spherical_harmonics(deg, dirs, coeffs, masks=masks)
    np.testing.assert_array_equal(_np(got)[0].view(np.uint32), exp.view(np.uint32))
    # and against the reference's own eval_sh output (fixture), where not masked
    np.testing.assert_allclose(_np(got)[0][masks], g[f"deg{deg}"][masks], rtol=0, atol=5e-6)
    assert float(got[0][~torch.from_numpy(masks).to(DEV)].abs().sum()) == 0.0


@pytest.mark.parametrize("variant", [0, 3])
def test_rasterize_golden(ops, golden_dir, variant):
    from street_crafter_amd import _lib
    g = _load(golden_dir, "pipeline_small.npz")
    prev = _lib.set_option("raster_fwd", variant)
    try:
        rc, ra = ops.rasterize_to_pixels(_t(g["means2d"])[None], _t(g["conics"])[None], _t(g["colors"])[None],
                                         _t(g["opacities"])[None], 128, 96, 16,
                                         _t(g["isect_offsets"], torch.int32), _t(g["flatten_ids"], torch.int32))
    finally:
        _lib.set_option("raster_fwd", prev)
    ok = ~g["unstable"]
    assert g["unstable"].mean() < 0.005
    rc, ra = _np(rc), _np(ra)
    np.testing.assert_allclose(rc[..., :3][ok], g["render_colors"][..., :3][ok], rtol=0, atol=1e-4)
    np.testing.assert_allclose(ra[ok], g["render_alphas"][ok], rtol=0, atol=1e-4)
    np.testing.assert_allclose(rc[..., 3][ok], g["render_colors"][..., 3][ok], rtol=1e-5, atol=1e-3)


def _pipeline_inputs(n, cam, seed, **kw):
    sc = make_scene(n, seed=seed, **kw)
    exp = O.render_frame(sc.means.numpy(), sc.quats.numpy(), sc.scales.numpy(), sc.opacities.numpy(),
                         sc.sh.numpy(), cam.viewmat.numpy(), cam.K.numpy(), cam.width, cam.height, sc.sh_degree,
                         near_plane=cam.znear, far_plane=cam.zfar, return_unstable=True)
    return sc, exp


@pytest.mark.parametrize("variant", [0, 3])
@pytest.mark.parametrize("w,h", [(256, 160), (200, 120)])
def test_full_pipeline_vs_oracle(ops, variant, w, h):
    """The caller's whole sequence (harness.caller.render_gaussians == render_kernel_gsplat) vs the oracle:
    ints bit-exact, pixels within tolerance, last_ids-dependent outputs consistent."""
    from street_crafter_amd import _lib
    from harness.caller import render_gaussians
    cam = make_camera(w, h, 280.0, 280.0)
    sc, exp = _pipeline_inputs(8000, cam, 31, z_range=(1.0, 40.0), scale_range=(0.01, 0.3))
    prev = _lib.set_option("raster_fwd", variant)
    try:
        with torch.no_grad():
            out = render_gaussians(sc.to(DEV), cam.to(DEV), return_intermediates=True)
    finally:
        _lib.set_option("raster_fwd", prev)
    np.testing.assert_array_equal(_np(out["_radii"])[0], exp["radii"])
    np.testing.assert_array_equal(_np(out["_tiles_per_gauss"])[0], exp["tiles_per_gauss"])
    np.testing.assert_array_equal(_np(out["_isect_ids"]), exp["isect_ids"])
    np.testing.assert_array_equal(_np(out["_flatten_ids"]), exp["flatten_ids"])
    np.testing.assert_array_equal(_np(out["_isect_offsets"]), exp["isect_offsets"])
    np.testing.assert_array_equal(_np(out["_colors"])[0].view(np.uint32), exp["colors"].view(np.uint32))
    np.testing.assert_array_equal(_np(out["_opacities"])[0].view(np.uint32), exp["opacities"].view(np.uint32))
    ok = ~exp["unstable"]
    assert exp["unstable"].mean() < 0.005
    rc = _np(out["_render_colors"])
    np.testing.assert_allclose(rc[..., :3][ok], exp["render_colors"][..., :3][ok], rtol=0, atol=1e-4)
    np.testing.assert_allclose(_np(out["_render_alphas"])[ok], exp["render_alphas"][ok], rtol=0, atol=1e-4)
    np.testing.assert_allclose(rc[..., 3][ok], exp["render_colors"][..., 3][ok], rtol=1e-5, atol=1e-3)
    # PSNR of the clamped RGB image vs the oracle's (target: within 0.01 dB of identical => > 80 dB)
    rgb_hip = _np(out["rgb"]).transpose(1, 2, 0)
    rgb_ref = np.clip(exp["render_colors"][0, ..., :3], 0.0, 1.0)
    assert O.psnr(rgb_hip[ok[0]], rgb_ref[ok[0]]) > 80.0
    assert out["rgb"].shape == (3, h, w) and out["acc"].shape == (1, h, w) and out["depth"].shape == (1, h, w)


@pytest.mark.parametrize("D", [1, 3, 4, 7, 32])
def test_rasterize_channel_counts_and_backgrounds(ops, golden_dir, D):
    g = _load(golden_dir, "pipeline_small.npz")
    rng = np.random.default_rng(D)
    N = g["means2d"].shape[0]
    colors = rng.uniform(0, 1, size=(1, N, D)).astype(np.float32)
    bg = rng.uniform(0, 1, size=(1, D)).astype(np.float32)
    tile_masks = rng.uniform(size=(1, 6, 8)) > 0.2
    exp = O.rasterize_to_pixels(g["means2d"][None], g["conics"][None], colors, g["opacities"][None], 128, 96, 16,
                                g["isect_offsets"], g["flatten_ids"], backgrounds=bg, masks=tile_masks,
                                return_unstable=True)
    rc, ra = ops.rasterize_to_pixels(_t(g["means2d"])[None], _t(g["conics"])[None], _t(colors), _t(g["opacities"])[None],
                                     128, 96, 16, _t(g["isect_offsets"], torch.int32),
                                     _t(g["flatten_ids"], torch.int32), backgrounds=_t(bg),
                                     masks=torch.from_numpy(tile_masks).to(DEV))
    ok = ~exp[3]
    np.testing.assert_allclose(_np(rc)[ok], exp[0][ok], rtol=0, atol=1e-4)
    np.testing.assert_allclose(_np(ra)[ok], exp[1][ok], rtol=0, atol=1e-4)


def test_rasterize_other_tile_sizes(ops):
    cam = make_camera(96, 64, 110.0, 110.0)
    sc = make_scene(1500, seed=4, z_range=(1.0, 20.0), scale_range=(0.02, 0.3))
    for ts in (8, 32):
        exp = O.render_frame(sc.means.numpy(), sc.quats.numpy(), sc.scales.numpy(), sc.opacities.numpy(), sc.sh.numpy(),
                             cam.viewmat.numpy(), cam.K.numpy(), 96, 64, 1, tile_size=ts, near_plane=0.001,
                             far_plane=1000.0, return_unstable=True)
        tw, th = math.ceil(96 / ts), math.ceil(64 / ts)
        tpg, ids, fids = ops.isect_tiles(_t(exp["means2d"])[None], _t(exp["radii"], torch.int32)[None],
                                         _t(exp["depths"])[None], ts, tw, th)
        np.testing.assert_array_equal(_np(ids), exp["isect_ids"])
        off = ops.isect_offset_encode(ids, 1, tw, th)
        rc, ra = ops.rasterize_to_pixels(_t(exp["means2d"])[None], _t(exp["conics"])[None], _t(exp["colors"])[None],
                                         _t(exp["opacities"])[None], 96, 64, ts, off, fids)
        ok = ~exp["unstable"]
        np.testing.assert_allclose(_np(rc)[..., :3][ok], exp["render_colors"][..., :3][ok], rtol=0, atol=1e-4)
        np.testing.assert_allclose(_np(ra)[ok], exp["render_alphas"][ok], rtol=0, atol=1e-4)


@pytest.mark.parametrize("variant", [0, 3])
def test_rasterize_survives_corrupt_ids_and_offsets(ops, golden_dir, variant):
    """Regression for the round-1 GPU memory fault: flatten_ids / isect_offsets
    are caller-supplied.  Out-of-range ids must be skipped and offsets outside [0, n_isects] (negative,
    past the end, decreasing) clamped -- a finite image and finite gradients, no out-of-bounds access.
    Entries that are still valid must render exactly as before."""
    from street_crafter_amd import _lib
    g = _load(golden_dir, "pipeline_small.npz")
    N, I = g["means2d"].shape[0], g["flatten_ids"].shape[0]
    rng = np.random.default_rng(11)
    fids = g["flatten_ids"].astype(np.int32).copy()
    bad = rng.choice(I, size=I // 10, replace=False)
    fids[bad] = rng.choice(np.array([-1, -2**31, N, N + 12345, 2**31 - 1], dtype=np.int64), size=bad.size).astype(np.int32)
    offs = g["isect_offsets"].astype(np.int32).copy()
    offs_bad = offs.copy()
    flat = offs_bad.reshape(-1)
    flat[3] = -5                      # negative start
    flat[7] = I + 1000                # start (and the previous tile's end) past the end
    flat[11] = 2**31 - 1
    flat[20] = flat[19] - 50 if flat[19] > 50 else 0     # decreasing
    args = dict(means2d=_t(g["means2d"])[None], conics=_t(g["conics"])[None], colors=_t(g["colors"])[None],
                opacities=_t(g["opacities"])[None])
    prev = _lib.set_option("raster_fwd", variant)
    prev_b = _lib.set_option("raster_bwd", 1 if variant == 3 else 0)
    try:
        for o, f in ((offs, fids), (offs_bad, g["flatten_ids"].astype(np.int32)), (offs_bad, fids)):
            leaves = {k: v.clone().requires_grad_(True) for k, v in args.items()}
            rc, ra = ops.rasterize_to_pixels(leaves["means2d"], leaves["conics"], leaves["colors"], leaves["opacities"],
                                             128, 96, 16, _t(o, torch.int32), _t(f, torch.int32), absgrad=True)
            (rc.sum() + ra.sum()).backward()
            torch.cuda.synchronize()
            assert torch.isfinite(rc).all() and torch.isfinite(ra).all()
            assert float(ra.detach().min()) >= 0.0 and float(ra.detach().max()) < 1.0
            for k, v in leaves.items():
                assert torch.isfinite(v.grad).all(), k
        # with only ids corrupted, the result equals rendering with those entries' opacity contribution removed:
        # the oracle skips them the same way when their id is replaced by a zero-opacity splat
        rc, ra = ops.rasterize_to_pixels(args["means2d"], args["conics"], args["colors"], args["opacities"],
                                         128, 96, 16, _t(offs, torch.int32), _t(fids, torch.int32))
        pad = lambda a, v: np.concatenate([a, np.full((1,) + a.shape[1:], v, a.dtype)], axis=0)
        f2 = np.where((fids < 0) | (fids >= N), N, fids).astype(np.int32)
        exp = O.rasterize_to_pixels(pad(g["means2d"], 0.0)[None], pad(g["conics"], 1.0)[None], pad(g["colors"], 0.0)[None],
                                    pad(g["opacities"], 0.0)[None], 128, 96, 16, offs, f2, return_unstable=True)
        ok = ~exp[3]
        np.testing.assert_allclose(_np(rc)[..., :3][ok], exp[0][..., :3][ok], rtol=0, atol=1e-4)
        np.testing.assert_allclose(_np(ra)[ok], exp[1][ok], rtol=0, atol=1e-4)
    finally:
        _lib.set_option("raster_fwd", prev)
        _lib.set_option("raster_bwd", prev_b)


# ---- backward ---------------------------------------------------------------------------------
def test_transposing_wave_reduction_routes_every_lane_once():
    """wave_transpose_sum16 (v_permlane32/16_swap + DPP mirrors): lane l must end with the 64-lane total
    of value l >> 2.  One-hot inputs check the routing exactly (every (value, lane) cell reaches its
    owners once and nobody else), random inputs check the sums."""
    from street_crafter_amd import _lib
    lib = _lib.load()
    n_hot = 16 * 64
    x = torch.zeros(n_hot + 64, 16, 64)
    for v in range(16):
        for l in range(64):
            x[v * 64 + l, v, l] = 1.0 + v + 100.0 * l        # distinct, exactly representable
    x[n_hot:] = torch.randn(64, 16, 64, generator=torch.Generator().manual_seed(1))
    xd = x.to(DEV)
    out = torch.empty(n_hot + 64, 64, device=DEV)
    _lib.check(lib.sc_test_wave_transpose_sum16(xd.data_ptr(), n_hot + 64, out.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream),
               "sc_test_wave_transpose_sum16")
    got = out.cpu()
    want = x.double().sum(dim=2).float()[:, torch.arange(64) // 4]      # [waves, 64]: value l >> 2
    assert torch.equal(got[:n_hot], want[:n_hot])
    torch.testing.assert_close(got[n_hot:], want[n_hot:], rtol=1e-5, atol=1e-5)



def _rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-12)


def _make_golden():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(root, "tools", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("which", ["plain", "clamped"])
def test_projection_backward_one_output_at_a_time(ops, golden_dir, which):
    """sc_projection_bwd against the committed float64 autograd gradients (tests/golden/bwd_small.npz), with a UNIT
    upstream gradient on ONE output at a time (means2d x / y, depth, the three conic entries, the compensation),
    so that a wrong term in one branch cannot hide behind the others (round 1 summed all outputs under random
    weights and accepted 2e-2 on the worst row).  Tolerance per row, relative to the row's own gradient scale:
      means2d, depth      2e-5 on every row (no cancellation anywhere);
      conic entries       1e-4 + 2e-6 kappa, kappa = (a1 + c1)^2 / det1 the conditioning of the blurred 2-D
                          covariance: fp32 forms det1 = a1 c1 - b^2 and every conic gradient divides by it;
      compensation        1e-4 + 2e-6 max(kappa, kappa0) + 2e-6 / (1 - comp^2): comp = sqrt(det0 / det1), so the
                          unblurred determinant's own cancellation (kappa0, needles) enters, and for a splat much
                          larger than the blur comp -> 1 and its gradient is the small difference of two nearly
                          equal terms (fp32 eps / (1 - comp^2)); those rows' gradients are tiny in absolute terms,
                          hence the additional bound of 1e-3 of the block's LARGEST gradient on every row.
    'clamped': Gaussians far off to the side, whose EWA Jacobian is evaluated at the clamp limit (the clamp's
    derivative is zero: a dedicated branch of the VJP)."""
    mg = _make_golden()
    fix = _load(golden_dir, "bwd_small.npz")
    means, quats, scales, cam = mg.projection_bwd_case(which)
    V, K = cam.viewmat.to(DEV)[None], cam.K.to(DEV)[None]
    vis = fix[f"proj_{which}_radii"] > 0
    kappa = fix[f"proj_{which}_kappa"].astype(np.float64)
    clamped = fix[f"proj_{which}_is_clamped"] & vis
    assert clamped.sum() > (100 if which == "clamped" else 0)
    worst = {}
    for gi, name in enumerate(mg.PROJ_GROUPS):
        leaves = [t.clone().to(DEV).requires_grad_(True) for t in (means, quats, scales)]
        radii, m2, d, con, comp = ops.fully_fused_projection(leaves[0], None, leaves[1], leaves[2], V, K, 320, 200,
                                                             near_plane=0.001, far_plane=1000.0, calc_compensations=True)
        np.testing.assert_array_equal(_np(radii)[0] > 0, vis)
        (m2[0, :, 0], m2[0, :, 1], d[0], con[0, :, 0], con[0, :, 1], con[0, :, 2], comp[0])[gi].sum().backward()
        got = np.concatenate([_np(t.grad) for t in leaves], axis=1).astype(np.float64)
        ref = fix[f"proj_{which}_{name}"].astype(np.float64)
        assert np.isfinite(got).all(), name
        # conditioning of the compensation from the forward outputs: a1, b, c1 back from the conic
        cn = _np(con)[0].astype(np.float64)
        with np.errstate(all="ignore"):
            di = cn[:, 0] * cn[:, 2] - cn[:, 1] ** 2
            a1, c1, bb = cn[:, 2] / di, cn[:, 0] / di, -cn[:, 1] / di
            det0 = np.maximum((a1 - 0.3) * (c1 - 0.3) - bb * bb, 1e-300)
            kappa0 = np.where(vis, (a1 + c1 - 0.6) ** 2 / det0, 0.0)
            comp64 = _np(comp)[0].astype(np.float64)
            kcomp = np.where(vis, 1.0 / np.maximum(1.0 - comp64 ** 2, 1e-12), 0.0)
        assert (got[~vis] == 0).all(), name                      # culled rows get exactly zero
        # the three blocks (means 3, quats 4, scales 3) have different units: each against its own scale
        for lo, hi, blk in ((0, 3, "means"), (3, 7, "quats"), (7, 10, "scales")):
            gr, gh = ref[vis, lo:hi], got[vis, lo:hi]
            scale = np.abs(gr).max(axis=1, keepdims=True)
            live = scale[:, 0] > 1e-12 * max(1e-300, np.abs(gr).max())
            if not live.any():                                   # e.g. depth does not depend on quats / scales
                assert np.abs(gh).max() <= 1e-6 * max(1.0, np.abs(got).max()), (name, blk)
                continue
            rel = (np.abs(gh - gr) / np.where(live[:, None], scale, 1.0))[live].max(axis=1)
            if gi < 3:
                tol = 2e-5
            elif gi < 6:
                tol = 1e-4 + 2e-6 * kappa[vis][live]
            else:
                tol = 1e-4 + 2e-6 * np.maximum(kappa, kappa0)[vis][live] + 2e-6 * kcomp[vis][live]
                assert np.abs(gh - gr).max() <= 1e-3 * np.abs(gr).max(), (name, blk, "absolute")
            ratio = rel / tol
            worst[(name, blk)] = float(ratio.max())
            assert ratio.max() < 1.0, (name, blk, float(rel.max()), float(kappa[vis][live][ratio.argmax()]))
    print("worst error / tolerance:", {k: round(v, 3) for k, v in worst.items()})


@pytest.mark.parametrize("clamp,floor", [("symmetric", 0.01), ("asymmetric", 0.01), ("symmetric", 0.1), ("asymmetric", 0.1)])
def test_projection_variants_selected_at_run_time(ops, golden_dir, clamp, floor):
    """sc_set_option("proj_clamp" / "radius_floor") (_lib.set_projection_variant; SURVEY A.1 U1 / U2: the reference installs
    an unpinned gsplat fork, README.md:35): under every pair the forward is bit-identical to tests/golden/proj_variants.npz
    (numpy oracle, off-centre principal point), the fused rasterization() forward projects the same way, and the backward
    agrees with float64 autograd of the torch oracle under the same pair."""
    from street_crafter_amd import _lib
    mg = _make_golden()
    g = _load(golden_dir, "proj_variants.npz")
    tag = f"{clamp}_{floor}"
    W, H, near, far = int(g["width"]), int(g["height"]), float(g["near"]), float(g["far"])
    means, quats, scales = _t(g["means"]), _t(g["quats"]), _t(g["scales"])
    V, K = _t(g["viewmat"])[None], _t(g["K"])[None]
    sh = torch.rand(means.shape[0], 4, 3, generator=torch.Generator().manual_seed(1)).to(DEV)
    opac = torch.rand(means.shape[0], generator=torch.Generator().manual_seed(2)).to(DEV)
    prev = _lib.set_projection_variant(clamp, floor)
    try:
        assert _lib.set_projection_variant() == (clamp, floor)
        with torch.no_grad():
            out = ops.fully_fused_projection(means, None, quats, scales, V, K, W, H, near_plane=near, far_plane=far,
                                             calc_compensations=True)
            for t, name in zip(out, ("radii", "means2d", "depths", "conics", "compensations")):
                np.testing.assert_array_equal(_np(t)[0].view(np.uint32), g[f"{tag}_{name}"].view(np.uint32), err_msg=name)
            rc, ra, meta = ops.rasterization(means, quats, scales, opac, sh, V, K, W, H, near_plane=near, far_plane=far,
                                             sh_degree=1, render_mode="RGB+ED", rasterize_mode="antialiased")
            assert meta["fused"]
            np.testing.assert_array_equal(_np(meta["radii"])[0], g[f"{tag}_radii"])
            np.testing.assert_array_equal(_np(meta["means2d"])[0].view(np.uint32), g[f"{tag}_means2d"].view(np.uint32))
            np.testing.assert_array_equal(_np(meta["conics"])[0].view(np.uint32), g[f"{tag}_conics"].view(np.uint32))
        # backward on a well-conditioned case with clamped Jacobians, seen through the same off-centre camera
        means, quats, scales, bcam = mg.projection_bwd_case("clamped")      # (float64 autograd on the box: inputs need not be portable)
        Kb = bcam.K.clone()
        Kb[0, 2], Kb[1, 2] = 0.4 * 320, 0.58 * 200
        rng = np.random.default_rng(5)
        wts = [rng.normal(size=s_).astype(np.float32) for s_ in ((360, 2), (360,), (360, 3), (360,))]
        leaves = [t.clone().to(DEV).requires_grad_(True) for t in (means, quats, scales)]
        radii, m2, d, con, comp = ops.fully_fused_projection(leaves[0], None, leaves[1], leaves[2], bcam.viewmat.to(DEV)[None],
                                                             Kb.to(DEV)[None], 320, 200, near_plane=0.001, far_plane=1000.0,
                                                             calc_compensations=True)
        ((m2[0] * _t(wts[0])).sum() + (d[0] * _t(wts[1])).sum() + (con[0] * _t(wts[2])).sum()
         + (comp[0] * _t(wts[3])).sum()).backward()
        ref = [t.clone().double().requires_grad_(True) for t in (means, quats, scales)]
        rr, m2r, dr, conr, compr = OT.fully_fused_projection(ref[0], ref[1], ref[2], bcam.viewmat.double(), Kb.double(), 320, 200,
                                                             near_plane=0.001, far_plane=1000.0, proj_clamp=clamp,
                                                             radius_floor=floor)
        ((m2r * torch.from_numpy(wts[0]).double()).sum() + (dr * torch.from_numpy(wts[1]).double()).sum()
         + (conr * torch.from_numpy(wts[2]).double()).sum() + (compr * torch.from_numpy(wts[3]).double()).sum()).backward()
        np.testing.assert_array_equal(_np(radii)[0] > 0, rr.numpy() > 0)
        assert int((_np(radii)[0] > 0).sum()) > 100
        for h_, r_, name in zip(leaves, ref, ("means", "quats", "scales")):
            assert _rel_err(_np(h_.grad), r_.grad.numpy()) < 2e-3, name
    finally:
        _lib.set_projection_variant(*prev)
    assert _lib.set_projection_variant() == prev


def test_backward_kernels_reproduce_the_committed_gradients(ops, golden_dir):
    """Rasterize and SH backward against tests/golden/bwd_small.npz (float64 autograd of the oracle, committed):
    the same bars as the recomputing tests below, without running the oracle on the box."""
    mg = _make_golden()
    fix = _load(golden_dir, "bwd_small.npz")
    g = _load(golden_dir, "pipeline_small.npz")
    w_c, w_a = mg.raster_bwd_weights(g["unstable"], int(fix["raster_seed"]))
    src = (g["means2d"][None], g["conics"][None], g["colors"][None], g["opacities"][None])
    hip = [_t(a).requires_grad_(True) for a in src]
    rc, ra = ops.rasterize_to_pixels(hip[0], hip[1], hip[2], hip[3], 128, 96, 16, _t(g["isect_offsets"], torch.int32),
                                     _t(g["flatten_ids"], torch.int32), absgrad=True)
    ((rc * _t(w_c)).sum() + (ra * _t(w_a)).sum()).backward()
    for h_, name in zip(hip, ("means2d", "conics", "colors", "opacities")):
        assert _rel_err(_np(h_.grad), fix[f"raster_v_{name}"]) < 2e-3, name
    assert _rel_err(_np(hip[0].absgrad)[0], fix["raster_absgrad"]) < 2e-3
    for deg in range(5):
        Kb = (deg + 1) ** 2
        d = _t(fix["sh_dirs"]).requires_grad_(True)
        c = _t(fix["sh_coeffs"][:, :Kb]).requires_grad_(True)
        (ops.spherical_harmonics(deg, d, c) * _t(fix["sh_v_colors"])).sum().backward()
        np.testing.assert_allclose(_np(c.grad), fix[f"sh_v_coeffs_deg{deg}"], rtol=1e-5, atol=1e-5)
        if deg:
            ref = fix[f"sh_v_dirs_deg{deg}"]
            np.testing.assert_allclose(_np(d.grad), ref, rtol=2e-4, atol=2e-4 * np.abs(ref).max())


def test_train_step_at_config2_full_size(ops):
    """BASELINE config 2's shape at full size: 1 M Gaussians, 1600 x 1066 (the reference trains Waymo frames at
    1600 px width, camera_utils.py:150-152), forward + backward through projection, SH and rasterize with absgrad
    (train.py:236; read at street_gaussian_model.py:505-519).  Too large for the float64 oracle: checked through
    properties -- the shipped backward (one wave per tile) against the reference-shaped kernel, which the small
    tests pin to autograd; every gradient finite; absgrad >= |grad| and zero exactly where nothing was rendered."""
    from street_crafter_amd import _lib
    from harness.caller import render_gaussians
    W, H = 1600, 1066
    cam = make_camera(W, H, 2050.0 * W / 1920.0, 2050.0 * W / 1920.0).to(DEV)
    base = make_scene(1_000_000)
    target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)).to(DEV)
    res = {}
    for variant in (1, 0):
        sc = base.to(DEV)
        leaves = (sc.means, sc.quats, sc.scales, sc.opacities, sc.sh)
        for t in leaves:
            t.requires_grad_(True)
        prev = _lib.set_option("raster_bwd", variant)
        try:
            out = render_gaussians(sc, cam, mode="train")
            ((out["rgb"] - target).abs().mean() + 0.01 * out["acc"].mean()).backward()
            torch.cuda.synchronize()
        finally:
            _lib.set_option("raster_bwd", prev)
        vp = out["viewspace_points"]
        assert vp.grad is not None and vp.absgrad.shape == vp.shape           # retain_grad contract + absgrad
        res[variant] = dict(grads=[t.grad.clone() for t in leaves], vp=vp.grad.clone(), absgrad=vp.absgrad.clone(),
                            vis=out["visibility_filter"].clone(), rgb=out["rgb"].detach().clone())
    a, b = res[1], res[0]
    assert torch.equal(a["rgb"].view(torch.int32), b["rgb"].view(torch.int32))           # same forward
    for ga, gb, name in zip(a["grads"], b["grads"], ("means", "quats", "scales", "opacities", "sh")):
        assert torch.isfinite(ga).all(), name
        assert _rel_err(_np(ga), _np(gb)) < 2e-4, name         # two summation orders of float atomics
    assert _rel_err(_np(a["vp"]), _np(b["vp"])) < 2e-4 and _rel_err(_np(a["absgrad"]), _np(b["absgrad"])) < 2e-4
    ab, gr = a["absgrad"], a["vp"]
    assert bool((ab + 1e-6 * ab.max() >= gr.abs()).all())
    assert float(ab[0][~a["vis"]].abs().max()) == 0.0 and float(gr[0][~a["vis"]].abs().max()) == 0.0
    # (S-1M is deep: tiles saturate after the nearest ~12 % of their lists, so only the front layer receives gradient)
    reached = float((ab[0][a["vis"]].sum(dim=1) > 0).float().mean())
    assert 0.002 < reached < 0.5, reached


@pytest.mark.parametrize("deg,k_total", [(0, 3), (1, 6), (2, 11), (3, 18), (4, 27),       # rows of odd length: scalar stores
                                          (0, 4), (1, 4), (1, 16), (2, 16), (3, 16), (4, 28)])  # rows of n x 16 B: float4 rows
def test_sh_backward_vs_autograd(ops, deg, k_total):
    g = torch.Generator().manual_seed(deg)
    n, K = 2000, (deg + 1) ** 2
    dirs = torch.randn(1, n, 3, generator=g) * 2.0
    coeffs = torch.randn(1, n, k_total, 3, generator=g)         # bases beyond the degree's must get zero grads
    masks = torch.rand(1, n, generator=g) > 0.2
    w = torch.randn(1, n, 3, generator=g)
    dh = dirs.clone().to(DEV).requires_grad_(True)
    ch = coeffs.clone().to(DEV).requires_grad_(True)
    (ops.spherical_harmonics(deg, dh, ch, masks=masks.to(DEV)) * w.to(DEV)).sum().backward()
    dr = dirs.clone().double().requires_grad_(True)
    cr = coeffs.clone().double().requires_grad_(True)
    (OT.spherical_harmonics(deg, dr, cr, masks=masks) * w.double()).sum().backward()
    assert _rel_err(_np(ch.grad), cr.grad.numpy()) < 1e-5
    if deg == 0:                       # colour does not depend on the direction at degree 0
        assert dr.grad is None and float(dh.grad.abs().sum()) == 0.0
    else:
        assert _rel_err(_np(dh.grad), dr.grad.numpy()) < 2e-4
    assert float(ch.grad[..., K:, :].abs().sum()) == 0.0


@pytest.mark.parametrize("bwd_variant", [0, 1])
@pytest.mark.parametrize("D,use_bg", [(4, False), (3, True), (6, False)])
def test_rasterize_backward_vs_autograd(ops, golden_dir, D, use_bg, bwd_variant):
    """bwd_variant 0 = reference-shaped kernel, 1 = one wave per tile (DPP reductions); D = 6 always
    takes the reference-shaped kernel."""
    from street_crafter_amd import _lib
    g = _load(golden_dir, "pipeline_small.npz")
    rng = np.random.default_rng(17 + D)
    N = g["means2d"].shape[0]
    colors = rng.uniform(0, 1, size=(1, N, D)).astype(np.float32)
    bg = rng.uniform(0, 1, size=(1, D)).astype(np.float32) if use_bg else None
    W, H = 128, 96
    w_c = rng.normal(size=(1, H, W, D)).astype(np.float32)
    w_a = rng.normal(size=(1, H, W, 1)).astype(np.float32)
    # pixels the oracle flags as threshold-unstable do not take part in the loss
    _, _, _, unstable = O.rasterize_to_pixels(g["means2d"][None], g["conics"][None], colors, g["opacities"][None], W, H,
                                              16, g["isect_offsets"], g["flatten_ids"], return_unstable=True)
    w_c[unstable] = 0.0
    w_a[unstable] = 0.0

    names = ("means2d", "conics", "colors", "opacities")
    src = (g["means2d"][None], g["conics"][None], colors, g["opacities"][None])
    hip = [_t(a).requires_grad_(True) for a in src]
    rc, ra = ops.rasterize_to_pixels(hip[0], hip[1], hip[2], hip[3], W, H, 16, _t(g["isect_offsets"], torch.int32),
                                     _t(g["flatten_ids"], torch.int32), backgrounds=None if bg is None else _t(bg),
                                     absgrad=True)
    prev = _lib.set_option("raster_bwd", bwd_variant)
    try:
        ((rc * _t(w_c)).sum() + (ra * _t(w_a)).sum()).backward()
    finally:
        _lib.set_option("raster_bwd", prev)

    ref = [torch.from_numpy(a).double().requires_grad_(True) for a in src]
    pix = []
    rcr, rar = OT.rasterize_to_pixels(ref[0], ref[1], ref[2], ref[3], W, H, 16, torch.from_numpy(g["isect_offsets"]),
                                      torch.from_numpy(g["flatten_ids"]),
                                      backgrounds=None if bg is None else torch.from_numpy(bg).double(),
                                      pixel_grads=pix)
    ((rcr * torch.from_numpy(w_c).double()).sum() + (rar * torch.from_numpy(w_a).double()).sum()).backward()
    for h_, r_, name in zip(hip, ref, names):
        assert _rel_err(_np(h_.grad), r_.grad.numpy()) < 2e-3, name
    # absgrad: attribute on the caller's tensor; value = sum over pixels of |per-pixel mean gradient|
    assert hasattr(hip[0], "absgrad") and hip[0].absgrad.shape == hip[0].shape
    ab, gr = _np(hip[0].absgrad), _np(hip[0].grad)
    assert (ab + 1e-6 * ab.max() >= np.abs(gr)).all()
    assert _rel_err(ab[0], OT.absgrad_from_pixel_grads(pix, N).numpy()) < 2e-3


@pytest.mark.parametrize("n,w,h,seed", [(3000, 128, 96, 1), (40_000, 400, 272, 2), (200_000, 800, 528, 3)])
def test_rasterize_backward_wave_matches_reference(ops, n, w, h, seed):
    """The wave-per-tile backward (variant 1, shipped default) against the reference-shaped kernel
    (variant 0, itself checked against autograd above) through the whole train-mode render, on
    scenes with several staging batches per tile and both staging sub-batches populated."""
    from street_crafter_amd import _lib
    from harness.caller import render_gaussians
    cam = make_camera(w, h, 2050.0 * w / 1920.0, 2050.0 * w / 1920.0).to(DEV)
    target = torch.rand(3, h, w, device=DEV, generator=torch.Generator(device=DEV).manual_seed(seed))
    grads = []
    for variant in (0, 1):
        sc = make_scene(n, seed=seed).to(DEV)
        params = (sc.means, sc.quats, sc.scales, sc.opacities, sc.sh)
        for t in params:
            t.requires_grad_(True)
        prev = _lib.set_option("raster_bwd", variant)
        try:
            out = render_gaussians(sc, cam, mode="train")
            ((out["rgb"] - target).abs().mean() + 0.05 * out["acc"].mean() + 0.01 * out["depth"].mean()).backward()
        finally:
            _lib.set_option("raster_bwd", prev)
        vp = out["viewspace_points"]
        grads.append([_np(t.grad) for t in params] + [_np(vp.grad), _np(vp.absgrad)])
    for name, a, b in zip(("means", "quats", "scales", "opacities", "sh", "means2d", "absgrad"), *grads):
        assert np.isfinite(b).all(), name
        assert _rel_err(b, a) < 2e-4, name      # fp32 sums in a different order, nothing more


def test_train_mode_contract_retain_grad_and_absgrad(ops):
    """What train.py:236 + street_gaussian_model.py:505-508 rely on: viewspace_points (a non-leaf
    output of the projection) keeps .grad after backward and gains .absgrad."""
    from harness.caller import render_gaussians
    cam = make_camera(160, 96, 180.0, 180.0).to(DEV)
    sc = make_scene(2500, seed=2, z_range=(1.0, 30.0), scale_range=(0.02, 0.3)).to(DEV)
    for t in (sc.means, sc.quats, sc.scales, sc.opacities, sc.sh):
        t.requires_grad_(True)
    out = render_gaussians(sc, cam, mode="train")
    loss = out["rgb"].mean() + 0.1 * out["acc"].mean() + 0.01 * out["depth"].mean()
    loss.backward()
    vp = out["viewspace_points"]
    assert vp.grad is not None and vp.grad.shape == (1, sc.n, 2)
    assert hasattr(vp, "absgrad") and vp.absgrad.shape == (1, sc.n, 2)
    vis = out["visibility_filter"]
    assert float(vp.absgrad[0][~vis].abs().sum()) == 0.0
    for t in (sc.means, sc.quats, sc.scales, sc.opacities, sc.sh):
        assert t.grad is not None and torch.isfinite(t.grad).all()
    assert float(sc.means.grad.abs().sum()) > 0


def test_native_autograd_functions_equal_the_python_ones(ops):
    """csrc/binding.cpp: the three differentiable operators as C++ autograd functions (the default when the compiled
    binding layer is loaded) against the Python torch.autograd.Functions of rendering.py: same kernels, so the images, every
    parameter gradient, viewspace_points.grad and .absgrad must be IDENTICAL -- with a background (its gradient too), with
    some inputs not requiring grad, and the backward atomics' order apart (float atomics: allclose at 1e-6 of the largest
    entry).  Also: the contract pieces the native path must keep (retain_grad on the projection's output, .absgrad on the
    caller's tensor object, only after backward)."""
    from harness.caller import render_gaussians
    from street_crafter_amd import _lib, rendering
    if _lib.fast() is None:
        pytest.skip("compiled binding layer not loaded")
    cam = make_camera(192, 112, 210.0, 210.0).to(DEV)

    def step(native, freeze=()):
        sc = make_scene(3000, seed=4, z_range=(1.0, 30.0), scale_range=(0.02, 0.3)).to(DEV)
        ps = {"means": sc.means, "quats": sc.quats, "scales": sc.scales, "opacities": sc.opacities, "sh": sc.sh}
        for k, t in ps.items():
            t.requires_grad_(k not in freeze)
        prev = rendering.set_native_autograd(native)
        try:
            out = render_gaussians(sc, cam, mode="train")
            vp = out["viewspace_points"]
            assert not hasattr(vp, "absgrad")                       # gsplat sets it in backward, not before
            (out["rgb"].square().mean() + 0.1 * out["acc"].mean() + 0.01 * out["depth"].mean()).backward()
        finally:
            rendering.set_native_autograd(prev)
        return out, {k: (None if t.grad is None else t.grad.clone()) for k, t in ps.items()}, vp

    def close(a, b, what):
        assert (a is None) == (b is None), what
        if a is not None:
            scale = float(b.abs().max()) + 1e-30
            assert float((a - b).abs().max()) <= 1e-6 * scale + 1e-12, (what, float((a - b).abs().max()), scale)

    for freeze in ((), ("quats", "sh")):
        o_n, g_n, vp_n = step(True, freeze)
        o_p, g_p, vp_p = step(False, freeze)
        assert torch.equal(o_n["rgb"], o_p["rgb"]) and torch.equal(o_n["acc"], o_p["acc"]) and torch.equal(o_n["depth"], o_p["depth"])
        for k in g_n:
            close(g_n[k], g_p[k], k)
            assert (g_n[k] is None) == (k in freeze)
        assert vp_n.grad is not None and vp_n.grad.shape == vp_p.grad.shape
        close(vp_n.grad, vp_p.grad, "viewspace_points.grad")
        close(vp_n.absgrad, vp_p.absgrad, "absgrad")
        assert float(vp_n.absgrad.sum()) > 0 and bool((vp_n.absgrad >= vp_n.grad.abs() - 1e-6).all())
    # rasterize_to_pixels with a background that requires grad, and spherical_harmonics with masks, operator by operator
    sc = make_scene(2000, seed=9, z_range=(1.0, 20.0), scale_range=(0.03, 0.3)).to(DEV)
    res = []
    for native in (True, False):
        prev = rendering.set_native_autograd(native)
        try:
            with torch.no_grad():
                radii, m2, d, con, comp = ops.fully_fused_projection(sc.means, None, sc.quats, sc.scales, cam.viewmat[None], cam.K[None],
                                                                      192, 112, near_plane=0.001, far_plane=1000.0, calc_compensations=True)
                tpg, ids, fids = ops.isect_tiles(m2, radii, d, 16, 12, 7, n_cameras=1)
                off = ops.isect_offset_encode(ids, 1, 12, 7)
            g = torch.Generator(DEV).manual_seed(3)
            cols = torch.rand(1, sc.n, 3, device=DEV, generator=g).requires_grad_(True)
            bg = torch.rand(1, 3, device=DEV, generator=g).requires_grad_(True)
            m2g = m2.clone().requires_grad_(True)
            rc, ra = ops.rasterize_to_pixels(m2g, con, cols, sc.opacities[None, :, 0] * comp, 192, 112, 16, off, fids, backgrounds=bg,
                                             absgrad=True)
            (rc.sum() + ra.square().sum()).backward()
            dirs = (sc.means[None] - cam.camera_center).detach().requires_grad_(True)
            coeffs = sc.sh[None].detach().clone().requires_grad_(True)
            c2 = ops.spherical_harmonics(1, dirs, coeffs, masks=radii > 0)
            c2.square().sum().backward()
            res.append((rc.detach(), ra.detach(), cols.grad, bg.grad, m2g.grad, m2g.absgrad, c2.detach(), dirs.grad, coeffs.grad))
        finally:
            rendering.set_native_autograd(prev)
    for a, b, what in zip(res[0], res[1], ("rc", "ra", "v_colors", "v_backgrounds", "v_means2d", "absgrad", "sh", "v_dirs", "v_coeffs")):
        close(a, b, what)


@pytest.mark.parametrize("render_mode,rasterize_mode,deg,use_bg", [
    ("RGB+ED", "antialiased", 1, False),     # what render_kernel_gsplat spells out by hand
    ("RGB+ED", "classic", 3, True),
    ("RGB+D", "antialiased", 0, False),
])
def test_rasterization_fused_matches_composition(ops, render_mode, rasterize_mode, deg, use_bg):
    """gsplat's one-call `rasterization()` runs a fused forward (projection + opacity
    compensation + SH + clamp + depth channel in one kernel, depth normalisation in the rasterizer's
    epilogue).  It must be BIT-identical to the composition of the separate operators + torch glue."""
    from street_crafter_amd import rendering
    sc = make_scene(30_000, sh_degree=deg, seed=11, z_range=(1.0, 60.0)).to(DEV)
    cams = [make_camera(640, 400, 600.0, 600.0, yaw=0.1 * i, shift=(0.3 * i, 0.0, -0.5 * i)) for i in range(2)]
    V = torch.stack([c.viewmat for c in cams]).to(DEV)
    K = torch.stack([c.K for c in cams]).to(DEV)
    bg = torch.rand(2, 3, device=DEV) if use_bg else None
    kw = dict(near_plane=0.001, far_plane=1000.0, sh_degree=deg, render_mode=render_mode,
              rasterize_mode=rasterize_mode, backgrounds=bg)
    outs = []
    for fused in (True, False):
        prev = rendering.set_fused_rasterization(fused)
        try:
            with torch.no_grad():
                outs.append(ops.rasterization(sc.means, sc.quats, sc.scales, sc.opacities[:, 0], sc.sh, V, K,
                                              640, 400, **kw))
        finally:
            rendering.set_fused_rasterization(prev)
    (rc_f, ra_f, m_f), (rc_c, ra_c, m_c) = outs
    assert m_f["fused"] and not m_c["fused"]
    # the fused rasterizer gathered from packed 48-B records: conics / opacities / colors of its meta are rebuilt
    # from them on first access (compared below), and the frame is the same without the records
    assert not dict.__contains__(m_f, "conics") and "conics" in m_f and m_f.get("opacities") is not None
    prev = rendering.set_packed_records(False)
    try:
        with torch.no_grad():
            rc_u, ra_u, m_u = ops.rasterization(sc.means, sc.quats, sc.scales, sc.opacities[:, 0], sc.sh, V, K, 640, 400, **kw)
    finally:
        rendering.set_packed_records(prev)
    assert m_u["fused"] and dict.__contains__(m_u, "conics")
    assert torch.equal(rc_u, rc_f) and torch.equal(ra_u, ra_f) and torch.equal(m_u["colors"], m_f["colors"])
    assert float(ra_f.sum()) > 0
    np.testing.assert_array_equal(_np(rc_f).view(np.uint32), _np(rc_c).view(np.uint32))
    np.testing.assert_array_equal(_np(ra_f).view(np.uint32), _np(ra_c).view(np.uint32))
    for key in ("radii", "means2d", "depths", "conics", "opacities", "colors", "isect_ids", "flatten_ids",
                "isect_offsets", "tiles_per_gauss"):
        a, b = _np(m_f[key]), _np(m_c[key])
        assert a.shape == b.shape, key
        np.testing.assert_array_equal(a.view(np.uint8), b.view(np.uint8), err_msg=key)


@pytest.mark.parametrize("render_mode,rasterize_mode,direct_colors", [
    ("RGB", "classic", False), ("D", "antialiased", False), ("ED", "classic", False), ("RGB+D", "antialiased", True),
    ("RGB", "antialiased", True),
])
def test_rasterization_other_modes_vs_oracle(ops, render_mode, rasterize_mode, direct_colors):
    """gsplat's one-call API in the modes the reference's renderer does not use (they take the composition of the
    operators, not the fused forward): colour only, depth only, expected depth, colours given directly instead of
    SH coefficients -- against the numpy oracle's operators chained by hand.  Ints bit-exact, pixels 1e-4."""
    w, h, n = 208, 120, 1500
    cam = make_camera(w, h, 200.0, 200.0)
    sc = make_scene(n, sh_degree=2, seed=17, z_range=(1.0, 25.0), scale_range=(0.03, 0.5))
    aa = rasterize_mode == "antialiased"
    cols_direct = np.random.default_rng(3).random((n, 3), dtype=np.float32)
    with torch.no_grad():
        rc, ra, meta = ops.rasterization(
            sc.means.to(DEV), sc.quats.to(DEV), sc.scales.to(DEV), sc.opacities[:, 0].to(DEV),
            _t(cols_direct).to(DEV) if direct_colors else sc.sh.to(DEV), cam.viewmat[None].to(DEV), cam.K[None].to(DEV),
            w, h, near_plane=0.01, far_plane=100.0, sh_degree=None if direct_colors else 2, render_mode=render_mode,
            rasterize_mode=rasterize_mode)
    assert not meta["fused"] or render_mode in ("RGB+D", "RGB+ED")
    radii, m2, d, con, comp = O.fully_fused_projection(_np(sc.means), _np(sc.quats), _np(sc.scales), _np(cam.viewmat),
                                                      _np(cam.K), w, h, near_plane=0.01, far_plane=100.0,
                                                      calc_compensations=aa)
    op = _np(sc.opacities[:, 0]) * comp if aa else _np(sc.opacities[:, 0])
    tw, th = math.ceil(w / 16), math.ceil(h / 16)
    tpg, ids, fids = O.isect_tiles(m2[None], radii[None], d[None], 16, tw, th, n_cameras=1)
    offs = O.isect_offset_encode(ids, 1, tw, th)
    if direct_colors:
        cols = cols_direct
    else:
        V = _np(cam.viewmat).astype(np.float64)
        center = (-V[:3, :3].T @ V[:3, 3]).astype(np.float32)
        cols = O.spherical_harmonics(2, _np(sc.means) - center[None], _np(sc.sh), masks=radii > 0)
        cols = np.maximum(cols + np.float32(0.5), np.float32(0.0))
    if render_mode in ("RGB+D", "RGB+ED"):
        cols = np.concatenate([cols, d[:, None]], axis=-1)
    elif render_mode in ("D", "ED"):
        cols = d[:, None]
    ref_c, ref_a, _, unstable = O.rasterize_to_pixels(m2[None], con[None], cols[None], op[None].astype(np.float32), w, h,
                                                     16, offs, fids, return_unstable=True)
    if render_mode in ("ED", "RGB+ED"):
        ref_c = np.concatenate([ref_c[..., :-1], ref_c[..., -1:] / np.maximum(ref_a, np.float32(1e-10))], axis=-1)
    np.testing.assert_array_equal(_np(meta["radii"])[0], radii)
    np.testing.assert_array_equal(_np(meta["flatten_ids"]), fids)
    np.testing.assert_array_equal(_np(meta["isect_offsets"]), offs)
    assert rc.shape == ref_c.shape and ra.shape == ref_a.shape
    stable = ~unstable[0]
    scale = max(1.0, float(np.abs(ref_c).max()))          # depth channels are in metres
    assert np.abs(_np(rc)[0][stable] - ref_c[0][stable]).max() <= 1e-4 * scale
    assert np.abs(_np(ra)[0][stable] - ref_a[0][stable]).max() <= 1e-4
    assert float(ref_a.sum()) > 0


def test_rasterization_fused_equals_reference_caller_sequence(ops):
    """The fused one-call path against the caller's hand-written sequence (harness.caller.render_gaussians =
    renderer.py:186-302) on the same camera, camera centre taken from the Camera as the reference does."""
    from harness.caller import render_gaussians
    sc = make_scene(50_000, seed=5).to(DEV)
    cam = make_camera(800, 528, 2050.0 * 800 / 1920, 2050.0 * 800 / 1920).to(DEV)
    with torch.no_grad():
        ref = render_gaussians(sc, cam, mode="eval", return_intermediates=True)
        rc, ra, meta = ops.rasterization(sc.means, sc.quats, sc.scales, sc.opacities[:, 0], sc.sh, cam.viewmat[None],
                                         cam.K[None], cam.width, cam.height, near_plane=cam.znear,
                                         far_plane=cam.zfar, sh_degree=sc.sh_degree, render_mode="RGB+ED",
                                         rasterize_mode="antialiased", camera_centers_=cam.camera_center[None])
    assert meta["fused"]
    np.testing.assert_array_equal(_np(rc[0, ..., :3].clamp(0.0, 1.0).permute(2, 0, 1)).view(np.uint32),
                                  _np(ref["rgb"]).view(np.uint32))
    np.testing.assert_array_equal(_np(rc[0, ..., 3]).view(np.uint32), _np(ref["depth"][0]).view(np.uint32))
    np.testing.assert_array_equal(_np(ra[0, ..., 0]).view(np.uint32), _np(ref["acc"][0]).view(np.uint32))
    np.testing.assert_array_equal(_np(meta["radii"]), _np(ref["_radii"]))


def test_rasterization_fused_all_culled_and_tiny(ops):
    """Edge cases through the fused forward: nothing visible (I == 0) gives a black image with alpha 0,
    and a handful of Gaussians works like the composition."""
    from street_crafter_amd import rendering
    cam = make_camera(200, 120, 220.0, 220.0).to(DEV)
    sc = make_scene(500, seed=3).to(DEV)
    kw = dict(sh_degree=1, render_mode="RGB+ED", rasterize_mode="antialiased")
    with torch.no_grad():
        rc, ra, meta = ops.rasterization(sc.means, sc.quats, sc.scales, sc.opacities[:, 0], sc.sh, cam.viewmat[None],
                                         cam.K[None], 200, 120, near_plane=500.0, far_plane=1000.0, **kw)
    assert meta["fused"] and int(meta["radii"].abs().sum()) == 0 and meta["flatten_ids"].numel() == 0
    assert float(rc.abs().sum()) == 0.0 and float(ra.abs().sum()) == 0.0
    assert meta["isect_ids"].numel() == 0                       # built on demand
    few = make_scene(3, seed=9, z_range=(3.0, 5.0), scale_range=(0.2, 0.4)).to(DEV)
    outs = []
    for fused in (True, False):
        prev = rendering.set_fused_rasterization(fused)
        try:
            with torch.no_grad():
                outs.append(ops.rasterization(few.means, few.quats, few.scales, few.opacities[:, 0], few.sh,
                                              cam.viewmat[None], cam.K[None], 200, 120, near_plane=0.001,
                                              far_plane=1000.0, **kw))
        finally:
            rendering.set_fused_rasterization(prev)
    assert float(outs[0][1].sum()) > 0
    np.testing.assert_array_equal(_np(outs[0][0]).view(np.uint32), _np(outs[1][0]).view(np.uint32))
    np.testing.assert_array_equal(_np(outs[0][1]).view(np.uint32), _np(outs[1][1]).view(np.uint32))


@pytest.mark.parametrize("n_cams", [1, 2])
def test_tile_dispatch_order_is_a_permutation_and_changes_no_pixel(ops, n_cams):
    """The rasterizer renders tiles longest-running first: the order is built on the device from the
    work every tile reported the previous time.  Whatever the hint holds -- nothing (first frame), this scene
    (second frame) or a different scene of the same frame shape (stale) -- the order must be a permutation of
    the tiles and the image must be BIT-identical to the one rendered without it."""
    from street_crafter_amd import _lib, rendering
    w, h = 640, 400
    cams = [make_camera(w, h, 600.0, 600.0, yaw=0.15 * i, shift=(0.4 * i, 0.0, 0.0)) for i in range(n_cams)]
    V = torch.stack([c.viewmat for c in cams]).to(DEV)
    K = torch.stack([c.K for c in cams]).to(DEV)
    from street_crafter_amd.scenes import make_street_scene
    scenes = [make_street_scene(60_000, seed=21)[0].to(DEV),             # a few tiles far above the mean: halves
              make_scene(60_000, seed=22, z_range=(0.5, 8.0), scale_range=(0.02, 0.25)).to(DEV)]    # even: none
    assert scenes[0].n == scenes[1].n        # same Gaussian count and frame shape: the two scenes share one hint buffer
    kw = dict(near_plane=0.001, far_plane=1000.0, render_mode="RGB+ED", rasterize_mode="antialiased")

    def render(sc):
        with torch.no_grad():
            return ops.rasterization(sc.means, sc.quats, sc.scales, sc.opacities[:, 0], sc.sh, V, K, w, h,
                                     sh_degree=sc.sh_degree, **kw)

    prev = rendering.set_tile_order(False)
    try:
        plain = [render(sc) for sc in scenes]
    finally:
        rendering.set_tile_order(prev)
    assert all(getattr(p[2]["isect_offsets"], "_sc_sched", None) is None for p in plain)
    n_tiles = n_cams * (w // 16) * (h // 16)
    rendering._STATE.tile_work.clear()
    from scipy.ndimage import maximum_filter
    split_counts, slots_seen = [], []
    for which in (0, 0, 1, 0):          # cold, warm, stale hint from scene 0, stale hint from scene 1
        torch.cuda.synchronize()
        banks = _np(rendering._tile_work(torch.device(DEV, torch.cuda.current_device()), n_cams, scenes[which].n, w // 16,
                                         h // 16)).copy().reshape(-1, n_tiles)
        rc, ra, meta = render(scenes[which])
        order, work = meta["isect_offsets"]._sc_sched
        torch.cuda.synchronize()
        o = _np(order)
        assert o.shape == (_lib.load().sc_tile_order_len(n_tiles),)
        assert o[-2] == 0        # the whole-tile list behind the forward's is built under raster_bwd_split 0 only
        slot = int(o[-1])        # the view slot of the call: which bank of the hint buffer this view reads and writes
        assert 0 <= slot < banks.shape[0] == _lib.load().sc_view_slots()
        slots_seen.append(slot)
        hint = banks[slot]
        o = o[: n_tiles + n_tiles // 8 + 8]                                            # the forward's list
        n_items = int((o >= 0).sum())
        assert (o[:n_items] >= 0).all() and (o[n_items:] == -1).all()          # padding at the end only
        tiles, kinds = o[:n_items] >> 2, o[:n_items] & 3
        halves = tiles[kinds == 1]
        # every tile once: whole, or as an upper and a lower half next to each other
        np.testing.assert_array_equal(np.sort(np.concatenate([tiles[kinds == 0], halves])), np.arange(n_tiles))
        first = np.nonzero(kinds == 1)[0]
        assert (kinds[first + 1] == 2).all() and (tiles[first + 1] == halves).all() and (kinds == 3).sum() == 0
        assert (kinds == 2).sum() == halves.size <= n_tiles // 8
        np.testing.assert_array_equal(_np(rc).view(np.uint32), _np(plain[which][0]).view(np.uint32))
        np.testing.assert_array_equal(_np(ra).view(np.uint32), _np(plain[which][1]).view(np.uint32))
        # heaviest first, by the class the order job files a tile under: the larger of its own work the previous
        # time and 3/4 of the largest work within 2 tiles of it, scaled so that the heaviest tile lands in the top
        # classes of 1024 (isect_bin.hip, center_scatter_kernel block 2)
        hint = np.minimum(hint, 65535)
        sm = maximum_filter(hint.reshape(n_cams, h // 16, w // 16), size=(1, 5, 5), mode="nearest").reshape(-1)
        sm = np.maximum(hint, (sm * 3) >> 2)
        shift = 0
        while (int(hint.max()) >> shift) > 1023:
            shift += 1
        cls = 1023 - np.minimum(1023, sm >> shift)
        assert (np.diff(cls[tiles]) >= 0).all()
        # halved: on a skewed frame (heaviest tile >= 3x the mean) the tiles within a factor 2 of the heaviest,
        # heaviest classes first; on an even frame the LAST tiles of the list (the lightest classes)
        if halves.size:
            assert hint.max() >= 32
            whole = tiles[kinds == 0]
            if int(hint.max()) * n_tiles >= 3 * int(hint.sum()):
                assert (cls[halves] <= 1023 - ((int(hint.max()) * 50 // 100) >> shift)).all()
                assert cls[halves].max() <= cls[whole].min()
            else:
                assert cls[halves].min() >= cls[whole].max()
                assert (kinds[-2 * halves.size:] != 0).all()
        split_counts.append(halves.size)
        # what the kernel reported: entries walked (+8 per staged batch), zero exactly where the tile list is empty
        offs = _np(meta["isect_offsets"]).reshape(-1).astype(np.int64)
        counts = np.diff(np.concatenate([offs, [meta["flatten_ids"].numel()]]))
        wk = _np(work).reshape(-1, n_tiles)
        others = np.arange(wk.shape[0]) != slot
        np.testing.assert_array_equal(wk[others], banks[others])          # the other views' banks are left alone
        wk = wk[slot]
        assert ((wk == 0) == (counts == 0)).all()
        assert (wk <= counts + 8 * ((counts + 63) // 64 + 1)).all()
        assert wk.max() > 50
    assert len(set(slots_seen)) == 1                                   # one view: one slot
    assert split_counts[0] == 0 and max(split_counts[1:]) > 0          # no hint: nothing is split
    n_fwd = n_tiles + n_tiles // 8 + 8
    # raster_bwd_split 0: the whole-tile list for the backward is built too -- the same tiles in the same order
    prev_b = _lib.set_option("raster_bwd_split", 0)
    try:
        rc, ra, meta = render(scenes[0])
        o = _np(meta["isect_offsets"]._sc_sched[0])
    finally:
        _lib.set_option("raster_bwd_split", prev_b)
    assert o[-2] == 1
    fwd = o[:n_fwd][o[:n_fwd] >= 0]
    np.testing.assert_array_equal(o[n_fwd:n_fwd + n_tiles], (fwd[(fwd & 3) != 2] >> 2) << 2)
    prev = _lib.set_option("raster_split", 0)
    try:
        rc, ra, meta = render(scenes[0])
        o = _np(meta["isect_offsets"]._sc_sched[0])
    finally:
        _lib.set_option("raster_split", prev)
    assert ((o[:n_tiles] & 3) == 0).all() and (o[n_tiles:n_fwd] == -1).all()
    np.testing.assert_array_equal(np.sort(o[:n_tiles] >> 2), np.arange(n_tiles))
    np.testing.assert_array_equal(_np(rc).view(np.uint32), _np(plain[0][0]).view(np.uint32))


@pytest.mark.parametrize("w,h,n", [(3840, 2160, 150_000), (40, 24, 300), (1000, 16, 5_000)])
def test_tile_dispatch_order_at_odd_frame_shapes(ops, w, h, n):
    """The dispatch list on frames the order job treats differently: 32 400 tiles (too many for the smoothed hint's
    LDS tables: plain per-tile hint), 6 tiles (no room for halves), a single row of tiles.  Three frames each
    (cold, warm, warm): the list stays a valid cover of the tiles and the pixels do not change."""
    from street_crafter_amd import _lib, rendering
    from street_crafter_amd.scenes import make_street_scene
    cam = make_camera(w, h, 0.6 * w, 0.6 * w)
    V, K = cam.viewmat[None].to(DEV), cam.K[None].to(DEV)
    sc = make_street_scene(n, seed=5)[0].to(DEV)
    kw = dict(near_plane=0.001, far_plane=1000.0, sh_degree=sc.sh_degree, render_mode="RGB+ED", rasterize_mode="antialiased")

    def render():
        with torch.no_grad():
            return ops.rasterization(sc.means, sc.quats, sc.scales, sc.opacities[:, 0], sc.sh, V, K, w, h, **kw)

    prev = rendering.set_tile_order(False)
    try:
        plain = render()
    finally:
        rendering.set_tile_order(prev)
    n_tiles = ((w + 15) // 16) * ((h + 15) // 16)
    rendering._STATE.tile_work.clear()
    for _ in range(3):
        rc, ra, meta = render()
        o = _np(meta["isect_offsets"]._sc_sched[0])
        assert o.shape == (_lib.load().sc_tile_order_len(n_tiles),)
        assert 0 <= o[-1] < _lib.load().sc_view_slots()
        o = o[: n_tiles + n_tiles // 8 + 8]
        items = o[o >= 0]
        assert (o[: items.size] >= 0).all()
        tiles, kinds = items >> 2, items & 3
        np.testing.assert_array_equal(np.sort(np.concatenate([tiles[kinds == 0], tiles[kinds == 1]])), np.arange(n_tiles))
        np.testing.assert_array_equal(np.sort(tiles[kinds == 1]), np.sort(tiles[kinds == 2]))
        assert (kinds == 1).sum() <= n_tiles // 8
        np.testing.assert_array_equal(_np(rc).view(np.uint32), _np(plain[0]).view(np.uint32))
        np.testing.assert_array_equal(_np(ra).view(np.uint32), _np(plain[1]).view(np.uint32))
    assert float(ra.sum()) > 0


@pytest.mark.parametrize("fill", ["random", "negative", "int_max", "one_hot", "ramp"])
def test_tile_dispatch_list_survives_any_hint_values(ops, fill):
    """The work hint is a scheduling hint only: whatever the buffer holds (it may be half written by another stream's
    rasterizer) the dispatch list must name every tile exactly once -- whole, or as two halves -- and the pixels must
    not change."""
    from street_crafter_amd import _lib, rendering
    w, h, n = 640, 400, 30_000
    cam = make_camera(w, h, 600.0, 600.0)
    V, K = cam.viewmat[None].to(DEV), cam.K[None].to(DEV)
    sc = make_scene(n, seed=41, z_range=(1.0, 40.0)).to(DEV)
    kw = dict(near_plane=0.001, far_plane=1000.0, sh_degree=sc.sh_degree, render_mode="RGB+ED", rasterize_mode="antialiased")

    def render():
        with torch.no_grad():
            return ops.rasterization(sc.means, sc.quats, sc.scales, sc.opacities[:, 0], sc.sh, V, K, w, h, **kw)

    prev = rendering.set_tile_order(False)
    try:
        plain = render()
    finally:
        rendering.set_tile_order(prev)
    n_tiles = (w // 16) * (h // 16)
    # [view slots, tiles]: every bank gets the pattern, whichever slot this view is filed under
    hint = rendering._tile_work(torch.device(DEV, torch.cuda.current_device()), 1, n, w // 16, h // 16).view(-1, n_tiles)
    g = torch.Generator(device=DEV).manual_seed(3)
    if fill == "random":
        hint.copy_(torch.randint(-2**31, 2**31 - 1, (n_tiles,), device=DEV, generator=g, dtype=torch.int64).to(torch.int32))
    elif fill == "negative":
        hint.fill_(-7)
    elif fill == "int_max":
        hint.fill_(2**31 - 1)
    elif fill == "one_hot":
        hint.zero_()
        hint[:, n_tiles // 2] = 60_000
    else:
        hint.copy_(torch.arange(n_tiles, device=DEV, dtype=torch.int32) * 7)
    rc, ra, meta = render()
    torch.cuda.synchronize()
    o = _np(meta["isect_offsets"]._sc_sched[0])[: n_tiles + n_tiles // 8 + 8]
    items = o[o >= 0]
    assert (o[: items.size] >= 0).all() and items.size <= n_tiles + n_tiles // 8
    tiles, kinds = items >> 2, items & 3
    np.testing.assert_array_equal(np.sort(np.concatenate([tiles[kinds == 0], tiles[kinds == 1]])), np.arange(n_tiles))
    np.testing.assert_array_equal(np.sort(tiles[kinds == 1]), np.sort(tiles[kinds == 2]))
    np.testing.assert_array_equal(_np(rc).view(np.uint32), _np(plain[0]).view(np.uint32))
    np.testing.assert_array_equal(_np(ra).view(np.uint32), _np(plain[1]).view(np.uint32))


def test_view_slots_keep_one_work_hint_per_camera_of_a_rig(ops):
    """A rig's cameras are rendered in turn: each must find the hint ITS last frame left, not the previous call's.
    The count launch files a call under a slot by camera 0's forward axis, on the device; the slot selects the bank
    of the hint buffer that isect_tiles reads and the rasterizer writes.  Checked: three cameras get three slots and
    keep them, a bank only changes when its own camera renders, the image never depends on any of it, a camera that
    turns slowly keeps its slot, and a ninth view takes over the least recently used slot."""
    from street_crafter_amd import _lib, rendering
    from harness.caller import render_gaussians
    from street_crafter_amd.scenes import make_street_scene
    lib = _lib.load()
    K_SLOTS = lib.sc_view_slots()
    w, h, n = 640, 400, 60_000
    T = (w // 16) * (h // 16)
    sc = make_street_scene(n, seed=21)[0].to(DEV)
    yaws = (0.0, 0.5, -0.5)
    cams = [make_camera(w, h, 600.0, 600.0, yaw=y).to(DEV) for y in yaws]
    dev = torch.device(DEV, torch.cuda.current_device())

    def frame(cam):
        with torch.no_grad():
            o = render_gaussians(sc, cam, return_intermediates=True)
        torch.cuda.synchronize()
        order, work = o["_isect_offsets"]._sc_sched
        return _np(o["_render_colors"]), int(_np(order)[-1]), _np(work).reshape(K_SLOTS, T).copy()

    prev = rendering.set_view_slots(False)
    try:
        plain = [frame(c)[0] for c in cams]
    finally:
        rendering.set_view_slots(prev)
    rendering._STATE.tile_work.clear()
    rendering._STATE.view_registry.pop(dev.index, None)          # a fresh registry: slots are handed out in order
    slots, banks = {}, np.zeros((K_SLOTS, T), np.int32)
    for rnd in range(3):
        for i, cam in enumerate(cams):
            img, slot, after = frame(cam)
            np.testing.assert_array_equal(img.view(np.uint32), plain[i].view(np.uint32))
            assert slots.setdefault(i, slot) == slot
            others = np.arange(K_SLOTS) != slot
            np.testing.assert_array_equal(after[others], banks[others])
            assert after[slot].max() > 50
            if rnd:          # same scene, same camera: the bank already held THIS view's work (the counts of split
                             # tiles vary with which half reports last; the empty tiles are exactly the view's)
                np.testing.assert_array_equal(after[slot] == 0, banks[slot] == 0)
                for j, sj in slots.items():
                    assert j == i or ((after[slot] == 0) != (banks[sj] == 0)).sum() > 20
            banks = after
    assert sorted(slots.values()) == [0, 1, 2]
    # a camera that turns by 0.05 rad per frame (far beyond the 7 degree window in total) keeps its slot
    s0 = None
    for k in range(12):
        _, slot, _ = frame(make_camera(w, h, 600.0, 600.0, yaw=1.2 + 0.05 * k).to(DEV))
        s0 = slot if s0 is None else s0
        assert slot == s0 and slot not in slots.values()
    # the registry: 8 far-apart views fill the slots (tiny frames: only the slot matters here), views within the
    # window reuse theirs, and a ninth view takes over the least recently used slot and no other
    rendering._STATE.view_registry.pop(dev.index, None)
    small = make_scene(500, seed=3).to(DEV)

    def pick(yaw):
        cam = make_camera(64, 48, 60.0, 60.0, yaw=yaw).to(DEV)
        with torch.no_grad():
            o = render_gaussians(small, cam, return_intermediates=True)
        torch.cuda.synchronize()
        return int(_np(o["_isect_offsets"]._sc_sched[0])[-1])

    first = [pick(0.7 * k) for k in range(K_SLOTS)]
    assert sorted(first) == list(range(K_SLOTS))
    assert [pick(0.7 * k + 0.05) for k in range(K_SLOTS)] == first          # within the window: same slots
    assert pick(0.7 * 1 + 0.05) == first[1]
    assert pick(0.7 * K_SLOTS + 0.3) == first[0]                            # new view: the LRU view's slot (k = 0)
    assert [pick(0.7 * k + 0.05) for k in range(1, K_SLOTS)] == first[1:]   # the others are untouched


def test_speculative_sort_is_sized_for_the_fullest_of_the_recent_views(ops):
    """Two cameras of one rig in turn, one of which sees several times more of the scene: the scatter + sort launch
    that is enqueued before the counts are known must be sized for the fuller view after it has been seen once
    (sized by the previous call alone it missed on every switch back to it); the frames do not change."""
    from street_crafter_amd import rendering
    from harness.caller import render_gaussians
    w, h, n = 640, 400, 40_000
    sc = make_scene(n, seed=17, z_range=(1.0, 30.0)).to(DEV)
    cams = [make_camera(w, h, 500.0, 500.0, yaw=0.0).to(DEV), make_camera(w, h, 500.0, 500.0, yaw=0.75).to(DEV)]
    key = (torch.cuda.current_device(), 1, n, 16, w // 16, h // 16)
    for d in (rendering._STATE.prediction, rendering._STATE.history, rendering._STATE.last_meta):
        d.pop(key, None)
    sizes, frames = [], []
    with torch.no_grad():
        for f in range(8):
            if f == 4:
                before = dict(rendering._STATE.stats)
            o = render_gaussians(sc, cams[f % 2], return_intermediates=True)
            sizes.append(int(o["_flatten_ids"].numel()))
            frames.append(_np(o["_render_colors"]))
    assert sizes[0] > 2 * sizes[1] > 0 and sizes[:2] * 3 == sizes[2:]
    assert rendering._STATE.stats["exact_relaunch"] == before["exact_relaunch"]
    assert rendering._STATE.stats["speculative_ok"] == before["speculative_ok"] + 4
    for f in range(2, 8):
        np.testing.assert_array_equal(frames[f].view(np.uint32), frames[f - 2].view(np.uint32))


def test_compiled_binding_layer_equals_the_ctypes_table(ops):
    """street_crafter_amd/csrc/binding.cpp (the default host path: one compiled call per operator allocates the outputs
    and calls the C ABI) against the ctypes table of _lib.py (the same C ABI called from Python): every forward tensor
    of a train-mode frame bit-identical, gradients equal up to the backward's atomic summation order; the binding is
    what the rest of this suite runs through (`_lib.fast()` is not None)."""
    from street_crafter_amd import _lib
    from harness.caller import render_gaussians
    assert _lib.fast() is not None and _lib.fast().abi_version().encode() == _lib.load().sc_version()
    cam = make_camera(320, 208, 300.0, 300.0).to(DEV)
    base = make_scene(20_000, seed=9, z_range=(1.0, 30.0), scale_range=(0.01, 0.25))
    outs = {}
    for fast_on in (True, False):
        prev = _lib.set_fast_binding(fast_on)
        try:
            assert (_lib.fast() is not None) == fast_on
            sc = base.to(DEV)
            ps = (sc.means, sc.quats, sc.scales, sc.opacities, sc.sh)
            for t in ps:
                t.requires_grad_(True)
            for _ in range(2):                      # second frame: warm dispatch list, speculative sort
                for t in ps:
                    t.grad = None
                o = render_gaussians(sc, cam, mode="train", return_intermediates=True)
                (o["rgb"].mean() + o["acc"].mean() + 0.01 * o["depth"].mean()).backward()
            with torch.no_grad():
                inf = render_gaussians(sc, cam, return_intermediates=True)
            torch.cuda.synchronize()
            outs[fast_on] = (o, inf, [t.grad.clone() for t in ps], o["viewspace_points"].absgrad.clone())
        finally:
            _lib.set_fast_binding(prev)
    keys = ("_radii", "_means2d", "_depths", "_conics", "_compensations", "_opacities", "_tiles_per_gauss", "_isect_ids",
            "_flatten_ids", "_isect_offsets", "_colors", "_render_colors", "_render_alphas", "rgb", "acc", "depth")
    for which in (0, 1):
        for k in keys:
            assert torch.equal(outs[True][which][k], outs[False][which][k]), (which, k)
    for ga, gb in zip(outs[True][2], outs[False][2]):
        assert torch.allclose(ga, gb, rtol=1e-4, atol=1e-7 + 1e-5 * float(gb.abs().max()))
    assert torch.allclose(outs[True][3], outs[False][3], rtol=1e-4, atol=1e-5 * float(outs[False][3].abs().max()))


def test_hints_on_carrier_tensors_survive_what_callers_do_to_tensors(ops):
    """VERDICT r2 weak 12.  Three correctness-neutral hints ride as attributes on tensors the operators hand out
    (`means2d._sc_viewmats`: the frame's cameras for the view slot; `isect_offsets._sc_sched`: the rasterizer's dispatch
    list; `isect_ids._sc_offsets`: the cached bucket scan).  A caller may clone / detach / copy / move / re-create any
    of these tensors between the operator calls: the attribute is then gone and the operators must fall back (view
    slot 0, plain dispatch, offsets recomputed from the keys) -- with the SAME integers and the same pixels, bit for bit."""
    import copy
    cam = make_camera(400, 272, 380.0, 380.0, yaw=0.2).to(DEV)
    sc = make_scene(30_000, seed=21, z_range=(1.0, 40.0), scale_range=(0.01, 0.3)).to(DEV)
    w2c, K = cam.viewmat[None], cam.K[None]
    W, H, tw, th = 400, 272, 25, 17

    def frame(carry):
        with torch.no_grad():
            radii, means2d, depths, conics, comps = ops.fully_fused_projection(
                sc.means, None, sc.quats, sc.scales, w2c, K, W, H, near_plane=cam.znear, far_plane=cam.zfar,
                calc_compensations=True)
            means2d = carry(means2d)
            tpg, ids, fids = ops.isect_tiles(means2d, radii, depths, 16, tw, th, n_cameras=1)
            ids = carry(ids)
            off = ops.isect_offset_encode(ids, 1, tw, th)
            off = carry(off)
            cols = ops.spherical_harmonics(sc.sh_degree, sc.means[None] - cam.camera_center, sc.sh[None], masks=radii > 0)
            cols = torch.cat([torch.clamp_min(cols + 0.5, 0.0), depths[..., None]], -1)
            rc, ra = ops.rasterize_to_pixels(means2d, conics, cols, sc.opacities[None, :, 0] * comps, W, H, 16, off, fids)
        return tpg, ids, fids, off, rc, ra

    base = None
    for _ in range(2):                       # second frame: warm hint, speculative sort
        base = frame(lambda t: t)
    assert getattr(base[3], "_sc_sched", None) is not None          # the plain path does carry the list
    carriers = {"clone": lambda t: t.clone(), "detach": lambda t: t.detach(), "copy.copy": copy.copy,
                "deepcopy": copy.deepcopy, "cpu round trip": lambda t: t.cpu().to(DEV),
                "new storage": lambda t: torch.empty_like(t).copy_(t), "view": lambda t: t.view(t.shape)[...]}
    for name, carry in carriers.items():
        got = frame(carry)
        torch.cuda.synchronize()
        for a, b, what in zip(base, got, ("tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets", "colors", "alphas")):
            assert torch.equal(torch.as_tensor(a), torch.as_tensor(b)), (name, what)


def test_per_shape_tables_are_pruned_together(ops):
    """isect_tiles keeps three per-shape tables (sizes history, prediction, last meta), keyed by (device, C, N, tile
    grid).  Densification changes N every 100 training iterations (train.py:292-310): all three are bounded and pruned
    together, oldest shape first."""
    from street_crafter_amd import rendering
    cam = make_camera(128, 96, 120.0, 120.0).to(DEV)
    sc = make_scene(400, seed=3, z_range=(1.0, 10.0)).to(DEV)
    w2c, K = cam.viewmat[None], cam.K[None]
    with torch.no_grad():
        for n in range(300, 300 + rendering._STATE.KEYS_MAX + 12):
            r, m2, d, _, _ = ops.fully_fused_projection(sc.means[:n], None, sc.quats[:n], sc.scales[:n], w2c, K, 128, 96)
            ops.isect_tiles(m2, r, d, 16, 8, 6, n_cameras=1)
    torch.cuda.synchronize()
    keys = set(rendering._STATE.history)
    assert len(keys) == rendering._STATE.KEYS_MAX
    assert set(rendering._STATE.prediction) <= keys and set(rendering._STATE.last_meta) <= keys
    newest = (torch.device(DEV).index or 0, 1, 300 + rendering._STATE.KEYS_MAX + 11, 16, 8, 6)
    assert any(k[2] == newest[2] for k in keys) and not any(k[2] == 300 and k[4:] == (8, 6) for k in keys)


def test_frame_without_gaussians_is_rendered_everywhere(ops):
    """N = 0: the intersection stage's short path still hands the rasterizer a dispatch list that names every tile
    (in the list's item format), so the whole frame is written: background colour, alpha 0 -- with the list on, and
    the same with it off."""
    from street_crafter_amd import rendering
    w, h = 200, 72
    cam = make_camera(w, h, 180.0, 180.0)
    V, K = cam.viewmat[None].to(DEV), cam.K[None].to(DEV)
    e = lambda *shape: torch.empty(*shape, device=DEV)         # noqa: E731
    bg = torch.tensor([[0.25, 0.5, 0.75]], device=DEV)
    outs = []
    for on in (True, True, False):
        prev = rendering.set_tile_order(on)
        try:
            with torch.no_grad():
                rc, ra, meta = ops.rasterization(e(0, 3), e(0, 4), e(0, 3), e(0), e(0, 3), V, K, w, h, sh_degree=None,
                                                 render_mode="RGB", backgrounds=bg)
        finally:
            rendering.set_tile_order(prev)
        torch.cuda.synchronize()
        assert meta["flatten_ids"].numel() == 0
        outs.append((_np(rc), _np(ra)))
    for rc, ra in outs:
        assert rc.shape == (1, h, w, 3) and (ra == 0).all()
        np.testing.assert_array_equal(rc, np.broadcast_to(_np(bg)[0], rc.shape))


@pytest.mark.parametrize("shape", ["even", "street"])
@pytest.mark.parametrize("bwd_split", [1, 0])
def test_tile_dispatch_order_in_training(ops, shape, bwd_split):
    """Forward and backward share the dispatch list -- the backward with the forward's half tiles (two waves add into
    the same gradients) or, `raster_bwd_split` 0, whole tiles in the same order; gradients match the plain dispatch
    up to fp32 summation order.  "even": the halves are the last tiles of the list; "street": the heaviest."""
    from street_crafter_amd import _lib, rendering
    from harness.caller import render_gaussians
    from street_crafter_amd.scenes import make_street_scene
    w, h = 400, 272
    cam = make_camera(w, h, 2050.0 * w / 1920.0, 2050.0 * w / 1920.0).to(DEV)
    target = torch.rand(3, h, w, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    grads, images, n_halves = [], [], []
    prev_split = _lib.set_option("raster_bwd_split", bwd_split)
    try:
        for on in (False, True, True):
            sc = (make_scene(40_000, seed=8) if shape == "even" else make_street_scene(40_000, seed=8)[0]).to(DEV)
            params = (sc.means, sc.quats, sc.scales, sc.opacities, sc.sh)
            for t in params:
                t.requires_grad_(True)
            prev = rendering.set_tile_order(on)
            try:
                out = render_gaussians(sc, cam, mode="train", return_intermediates=True)
                ((out["rgb"] - target).abs().mean() + 0.05 * out["acc"].mean() + 0.01 * out["depth"].mean()).backward()
            finally:
                rendering.set_tile_order(prev)
            sched = getattr(out["_isect_offsets"], "_sc_sched", None)
            n_halves.append(0 if sched is None else int(((_np(sched[0])[: 425 + 425 // 8 + 8] & 3) == 1).sum()))
            images.append(_np(out["rgb"].detach()))
            grads.append([_np(t.grad) for t in params] + [_np(out["viewspace_points"].absgrad)])
    finally:
        _lib.set_option("raster_bwd_split", prev_split)
    assert n_halves[0] == 0 and n_halves[2] > 0            # 25 x 17 = 425 tiles; the warm list has halves
    np.testing.assert_array_equal(images[0].view(np.uint32), images[1].view(np.uint32))
    np.testing.assert_array_equal(images[0].view(np.uint32), images[2].view(np.uint32))
    for k in (1, 2):
        for name, a, b in zip(("means", "quats", "scales", "opacities", "sh", "absgrad"), grads[0], grads[k]):
            assert np.isfinite(b).all(), name
            assert _rel_err(b, a) < 2e-4, name


def test_rasterization_falls_back_to_autograd_operators_when_training(ops):
    sc = make_scene(3000, seed=6, z_range=(1.0, 30.0), scale_range=(0.02, 0.3)).to(DEV)
    cam = make_camera(160, 96, 180.0, 180.0).to(DEV)
    sc.means.requires_grad_(True)
    rc, ra, meta = ops.rasterization(sc.means, sc.quats, sc.scales, sc.opacities[:, 0], sc.sh, cam.viewmat[None],
                                     cam.K[None], 160, 96, near_plane=0.001, far_plane=1000.0, sh_degree=1,
                                     render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True)
    assert not meta["fused"]
    (rc.sum() + ra.sum()).backward()
    assert sc.means.grad is not None and torch.isfinite(sc.means.grad).all() and float(sc.means.grad.abs().sum()) > 0


def test_densification_statistics_consumer_vs_oracle(ops):
    """what train.py:283-284 reads after backward -- means2d.grad (non-leaf, retain_grad),
    means2d.absgrad (attribute set by the rasterizer's backward), visibility_filter and radii -- fed to
    the reference's consumer (densify_stats.py mirrors street_gaussian_model.py:487-521) over two
    cameras and two sub-model ranges, against the same consumer fed by the float64 oracle."""
    from street_crafter_amd.densify_stats import DensificationStats, accumulate_from_render
    from harness.caller import render_gaussians
    W, H = 128, 96
    sc = make_scene(2500, seed=4, z_range=(1.0, 30.0), scale_range=(0.02, 0.3))
    ranges = {"background": (0, 1800), "obj_001": (1800, 2500)}
    st_hip = DensificationStats(ranges, device=DEV)
    st_ref = DensificationStats(ranges, device="cpu")
    rng = np.random.default_rng(3)
    scd = sc.to(DEV)
    params = (scd.means, scd.quats, scd.scales, scd.opacities, scd.sh)
    for t in params:
        t.requires_grad_(True)
    for i in range(2):
        cam = make_camera(W, H, 150.0, 150.0, yaw=0.06 * i, shift=(0.2 * i, 0.0, 0.0))
        for t in params:
            t.grad = None
        out = render_gaussians(scd, cam.to(DEV), mode="train", return_intermediates=True)
        m2, con = _np(out["_means2d"]), _np(out["_conics"])
        col, opa = _np(out["_colors"]), _np(out["_opacities"])
        offs, fids = _np(out["_isect_offsets"]), _np(out["_flatten_ids"])
        # loss weights; threshold-unstable pixels do not take part (see module docstring)
        _, _, _, unstable = O.rasterize_to_pixels(m2, con, col, opa, W, H, 16, offs, fids, return_unstable=True)
        w_c = rng.normal(size=(1, H, W, 4)).astype(np.float32)
        w_a = rng.normal(size=(1, H, W, 1)).astype(np.float32)
        w_c[unstable] = 0.0
        w_a[unstable] = 0.0
        ((out["_render_colors"] * _t(w_c)).sum() + (out["_render_alphas"] * _t(w_a)).sum()).backward()
        accumulate_from_render(st_hip, out, W, H)

        m2r = torch.from_numpy(m2).double().requires_grad_(True)
        pix = []
        rc, ra = OT.rasterize_to_pixels(m2r, torch.from_numpy(con).double(), torch.from_numpy(col).double(),
                                        torch.from_numpy(opa).double(), W, H, 16, torch.from_numpy(offs),
                                        torch.from_numpy(fids), pixel_grads=pix)
        ((rc * torch.from_numpy(w_c).double()).sum() + (ra * torch.from_numpy(w_a).double()).sum()).backward()
        vp = torch.zeros(1, sc.n, 2)
        vp.grad = m2r.grad.float()
        vp.absgrad = OT.absgrad_from_pixel_grads(pix, sc.n).float()[None]
        radii_o = O.fully_fused_projection(sc.means.numpy(), sc.quats.numpy(), sc.scales.numpy(),
                                           cam.viewmat.numpy(), cam.K.numpy(), W, H, near_plane=cam.znear,
                                           far_plane=cam.zfar)[0]
        ref_out = {"radii": torch.from_numpy(radii_o) / float(max(H, W)),
                   "visibility_filter": torch.from_numpy(radii_o > 0), "viewspace_points": vp}
        accumulate_from_render(st_ref, ref_out, W, H)
    for name in ranges:
        a, b = _np(st_hip.xyz_gradient_accum[name]), st_ref.xyz_gradient_accum[name].numpy()
        assert _rel_err(a[:, 0], b[:, 0]) < 2e-3 and _rel_err(a[:, 1], b[:, 1]) < 2e-3, name
        np.testing.assert_array_equal(_np(st_hip.denom[name]), st_ref.denom[name].numpy())
        np.testing.assert_array_equal(_np(st_hip.max_radii2D[name]), st_ref.max_radii2D[name].numpy())
        assert float(st_hip.denom[name].sum()) > 0
        g_abs = _np(st_hip.mean_grads(name, use_abs=False))
        assert np.isfinite(g_abs).all()


def test_render_sharded_two_frames_in_flight_is_identical(ops):
    """dist.render_sharded with frames alternating over two HIP streams returns the same uint8 frames as
    the sequential loop (single process: the gather is the identity)."""
    from street_crafter_amd.dist import render_sharded, to_uint8_frame
    from harness.caller import render_gaussians
    sc = make_scene(60_000, seed=12).to(DEV)
    cams = [make_camera(640, 400, 600.0, 600.0, yaw=0.02 * i, shift=(0.1 * i, 0.0, 0.0)).to(DEV) for i in range(6)]

    def frame(f):
        with torch.no_grad():
            return to_uint8_frame(render_gaussians(sc, cams[f])["rgb"])

    seq = render_sharded(6, frame)
    ovl = render_sharded(6, frame, frames_in_flight=2)
    torch.cuda.synchronize()
    assert len(seq) == len(ovl) == 6
    for a, b in zip(seq, ovl):
        assert a.dtype == torch.uint8 and a.shape == (400, 640, 3) and torch.equal(a, b)
    assert not torch.equal(seq[0], seq[5])


@pytest.mark.parametrize("w,h,D,use_bg,use_mask", [(256, 160, 4, False, False), (333, 211, 4, True, True),
                                                     (402, 130, 3, True, False), (128, 96, 3, False, True)])
def test_planar_output_is_the_same_tensor_by_value(ops, w, h, D, use_bg, use_mask):
    """rendering.set_planar_output: under no_grad rasterize_to_pixels stores render_colors as one plane per channel and
    hands out the permuted [C,H,W,D] view (renderer.py:282-300 slices it channel-wise).  Values, shape and the frame the
    caller makes of it are identical to the interleaved form; widths that are not multiples of 4 / 2 take the scalar
    stores; a call that requires grad takes the interleaved form."""
    from street_crafter_amd import rendering
    from street_crafter_amd.dist import to_uint8_frame
    sc = make_scene(30_000, seed=21, z_range=(1.0, 30.0), scale_range=(0.01, 0.4))
    cam = make_camera(w, h, 300.0, 300.0)
    R = ops
    with torch.no_grad():
        radii, m2, d, con, comp = _project(R, sc.means.numpy(), sc.quats.numpy(), sc.scales.numpy(), cam.viewmat.numpy(),
                                           cam.K.numpy(), w, h, 0.001, 1000.0)
        tw, th = (w + 15) // 16, (h + 15) // 16
        _, ids, fids = R.isect_tiles(m2, radii, d, 16, tw, th, n_cameras=1)
        off = R.isect_offset_encode(ids, 1, tw, th)
        g = torch.Generator().manual_seed(3)
        colors = torch.rand(1, sc.n, D, generator=g).to(DEV)
        opac = (sc.opacities[None, :, 0].to(DEV) * comp).contiguous()
        bg = torch.rand(1, D, generator=g).to(DEV) if use_bg else None
        masks = (torch.rand(1, th, tw, generator=g) > 0.2).to(DEV) if use_mask else None
        outs = {}
        for planar in (False, True):
            prev = rendering.set_planar_output(planar)
            try:
                outs[planar] = R.rasterize_to_pixels(m2, con, colors, opac, w, h, 16, off, fids, backgrounds=bg, masks=masks)
            finally:
                rendering.set_planar_output(prev)
        (c0, a0), (c1, a1) = outs[False], outs[True]
        assert c0.shape == c1.shape == (1, h, w, D) and c0.is_contiguous()
        assert c1.stride() == (D * h * w, w, 1, h * w)             # planes behind the same indexing
        assert torch.equal(c0, c1) and torch.equal(a0, a1)
        # what the caller does with it (renderer.py:282-300), and the uint8 frame of it
        f0 = to_uint8_frame(torch.clamp(c0[..., :3], 0.0, 1.0)[0].permute(2, 0, 1))
        f1 = to_uint8_frame(torch.clamp(c1[..., :3], 0.0, 1.0)[0].permute(2, 0, 1))
        assert torch.equal(f0, f1) and int(f0.max()) > 0
        if D == 4:
            assert torch.equal(c0[..., -1:] / a0.clamp(min=1e-10), c1[..., -1:] / a1.clamp(min=1e-10))
    colors.requires_grad_(True)
    c2, _ = R.rasterize_to_pixels(m2, con, colors, opac, w, h, 16, off, fids, backgrounds=bg, masks=masks)
    assert c2.is_contiguous() and torch.equal(c2.detach(), c0)


def test_planar_output_with_two_cameras_and_the_two_pass_frame(ops):
    """Planes per camera ([C][D][H][W]) behind the [C,H,W,D] indexing, and the two-pass novel-view frame (foreground planar or not,
    sky planar or not: dist.to_uint8_frame reads either layout off the strides) -- identical values in every combination."""
    from street_crafter_amd import rendering
    from street_crafter_amd.dist import to_uint8_frame
    w, h, C = 200, 120, 2
    sc = make_scene(12_000, seed=41, z_range=(1.0, 30.0), scale_range=(0.02, 0.5))
    cams = [make_camera(w, h, 220.0, 220.0, yaw=0.15 * c) for c in range(C)]
    with torch.no_grad():
        radii, m2, d, con, comp = ops.fully_fused_projection(
            sc.means.to(DEV), None, sc.quats.to(DEV), sc.scales.to(DEV), torch.stack([c.viewmat for c in cams]).to(DEV),
            torch.stack([c.K for c in cams]).to(DEV), w, h, near_plane=0.001, far_plane=1000.0, calc_compensations=True)
        tw, th = (w + 15) // 16, (h + 15) // 16
        _, ids, fids = ops.isect_tiles(m2, radii, d, 16, tw, th, n_cameras=C)
        off = ops.isect_offset_encode(ids, C, tw, th)
        colors = torch.rand(C, sc.n, 4, generator=torch.Generator().manual_seed(4)).to(DEV)
        opac = (sc.opacities[None, :, 0].to(DEV) * comp).contiguous()
        outs = {}
        for planar in (False, True):
            prev = rendering.set_planar_output(planar)
            try:
                outs[planar] = ops.rasterize_to_pixels(m2, con, colors, opac, w, h, 16, off, fids)
            finally:
                rendering.set_planar_output(prev)
        (c0, a0), (c1, a1) = outs[False], outs[True]
        assert c1.stride() == (4 * h * w, w, 1, h * w) and torch.equal(c0, c1) and torch.equal(a0, a1)
        assert float(a0[1].sum()) > 0 and not torch.equal(c0[0], c0[1])
        # camera 0 as the foreground pass, camera 1 standing in for the sky pass: all four layout combinations
        ref = None
        for fg, sky in ((c0, c0), (c1, c0), (c0, c1), (c1, c1)):
            f = to_uint8_frame(fg[0, ..., :3].permute(2, 0, 1), acc=a0[0, ..., 0], sky_rgb_chw=sky[1, ..., :3].permute(2, 0, 1))
            ref = f if ref is None else ref
            assert torch.equal(f, ref)
        assert int(ref.max()) > 0


def test_rasterizer_on_a_cu_masked_or_prioritised_side_stream_renders_the_same_frames(ops):
    """rendering.set_raster_side_stream + dist.make_stream (sc_stream_create): the inference rasterizer handed to a stream
    confined to 64 of the CUs, or of the lowest priority, fenced by events -- frames identical to the plain loop, with two
    frames in flight as well."""
    from street_crafter_amd import rendering
    from street_crafter_amd.dist import destroy_stream, make_stream, to_uint8_frame
    from harness.caller import render_gaussians
    sc = make_scene(80_000, seed=13).to(DEV)
    cams = [make_camera(640, 400, 600.0, 600.0, yaw=0.02 * i, shift=(0.1 * i, 0.0, 0.0)).to(DEV) for i in range(6)]
    dev = torch.device(DEV, torch.cuda.current_device())

    def frames(mains=None):
        out = []
        home = torch.cuda.current_stream(dev)
        with torch.no_grad():
            for f, cam in enumerate(cams):
                if mains:
                    torch.cuda.set_stream(mains[f % len(mains)])
                out.append(to_uint8_frame(render_gaussians(sc, cam)["rgb"]))
        torch.cuda.set_stream(home)
        torch.cuda.synchronize()
        return out

    ref = frames()
    for kw in (dict(cus=64), dict(priority=100)):
        side = make_stream(dev, **kw)
        mains = [make_stream(dev), make_stream(dev)]
        try:
            assert rendering.set_raster_side_stream(dev, side) is None
            got = frames()
            got2 = frames(mains)
        finally:
            rendering.set_raster_side_stream(dev, None)
            torch.cuda.synchronize()
            for st in [side] + mains:
                destroy_stream(st)
        for a, b, c in zip(ref, got, got2):
            assert torch.equal(a, b) and torch.equal(a, c)
    assert not rendering._RASTER_SIDE


@pytest.mark.parametrize("placeholders", [True, False])
def test_flatten_ids_handed_to_a_dlpack_consumer_right_after_isect_tiles(ops, golden_dir, placeholders):
    """what `flatten_ids` / `isect_ids` are until first use.  Default (placeholders on): LazyTensors whose
    length / contents are settled on first observation -- the DLPack PROTOCOL (torch.from_dlpack, i.e. `__dlpack__`) settles
    them, and after `.materialize()` so does the legacy capsule call; with SC_DEFER_ISECT=0 SC_LAZY_IDS=0 (here: the set_*
    switches) they are ordinary, fully written tensors the moment isect_tiles returns, and the legacy
    torch.utils.dlpack.to_dlpack -- which unwraps the tensor in C++ without any hook -- exports the true lists."""
    from torch.utils import dlpack
    from street_crafter_amd import rendering
    g = _load(golden_dir, "pipeline_small.npz")
    a = (_t(g["means2d"])[None], _t(g["radii"], torch.int32)[None], _t(g["depths"])[None], 16, 8, 6)
    prev = (rendering.set_deferred_isect(placeholders), rendering.set_lazy_isect_ids(placeholders))
    try:
        rendering.reset_state()
        for rep in range(3):             # (the deferral starts with the second call of a frame shape)
            tpg, ids, fids = ops.isect_tiles(*a, n_cameras=1)
            if placeholders:
                got_f = torch.from_dlpack(fids)                  # the protocol: settles first
                got_i = torch.from_dlpack(ids)
                if rep:                      # (the first call of a frame shape has no prediction: plain tensors)
                    assert type(fids) is not torch.Tensor and fids.is_materialized
                    again = dlpack.from_dlpack(dlpack.to_dlpack(fids.materialize()))
                    assert torch.equal(again, got_f)
            else:
                assert type(fids) is torch.Tensor and type(ids) is torch.Tensor
                got_f = dlpack.from_dlpack(dlpack.to_dlpack(fids))          # legacy capsule, no torch call in between
                got_i = dlpack.from_dlpack(dlpack.to_dlpack(ids))
            np.testing.assert_array_equal(_np(got_f), g["flatten_ids"])
            np.testing.assert_array_equal(_np(got_i), g["isect_ids"])
            np.testing.assert_array_equal(_np(ops.isect_offset_encode(ids, 1, 8, 6)), g["isect_offsets"])
    finally:
        rendering.set_deferred_isect(prev[0])
        rendering.set_lazy_isect_ids(prev[1])
        rendering.reset_state()


@pytest.mark.parametrize("which", ["golden", "iid", "big_splats", "two_cameras", "odd_frame"])
def test_isect_pull_route_is_bit_exact(ops, golden_dir, which):
    """sc_set_option("isect_pull", 1): every super-tile bucket's sort workgroup gathers its records itself from a (size class,
    anchor)-sorted payload (no scatter launch, no records buffer).  Same tensors as the scatter route and the oracle, bit for
    bit, incl. oversized buckets (big_pull + big_split), two cameras, and frames that are not whole super-tiles."""
    from street_crafter_amd import _lib, rendering
    if which == "golden":
        g = _load(golden_dir, "pipeline_small.npz")
        m2, radii, d, tw, th, C = _t(g["means2d"])[None], _t(g["radii"], torch.int32)[None], _t(g["depths"])[None], 8, 6, 1
    else:
        cfg = {"iid": (120_000, 1, 960, 640, (0.005, 0.15)), "big_splats": (25_000, 1, 1920, 1280, (0.5, 4.0)),
               "two_cameras": (60_000, 2, 640, 400, (0.01, 0.3)), "odd_frame": (20_000, 1, 333, 211, (0.01, 0.6))}[which]
        n, C, w, h, sr = cfg
        sc = make_scene(n, seed=31, scale_range=sr)
        cams = [make_camera(w, h, 2050.0 * w / 1920.0, 2050.0 * w / 1920.0, yaw=0.2 * c) for c in range(C)]
        with torch.no_grad():
            radii, m2, d, _, _ = ops.fully_fused_projection(
                sc.means.to(DEV), None, sc.quats.to(DEV), sc.scales.to(DEV), torch.stack([c.viewmat for c in cams]).to(DEV),
                torch.stack([c.K for c in cams]).to(DEV), w, h, near_plane=0.001, far_plane=1000.0)
        tw, th = (w + 15) // 16, (h + 15) // 16
    e_tpg, e_ids, e_f = O.isect_tiles(_np(m2), _np(radii), _np(d), 16, tw, th, n_cameras=C)
    e_off = O.isect_offset_encode(e_ids, C, tw, th)
    for pull in (1, 0):
        prev = _lib.set_option("isect_pull", pull)
        rendering.reset_state()
        try:
            for rep in range(2):           # (second call: predicted sizes, deferred settle)
                tpg, ids, fids = ops.isect_tiles(m2, radii, d, 16, tw, th, n_cameras=C)
                if which == "big_splats" and rep == 0:      # the case is there for the oversized-bucket path
                    assert list(rendering._STATE.last_meta.values())[-1][2] > _lib.load().sc_isect_bin_bucket_capacity()
                off = ops.isect_offset_encode(ids, C, tw, th)
                np.testing.assert_array_equal(_np(tpg), e_tpg)
                np.testing.assert_array_equal(_np(off), e_off)
                np.testing.assert_array_equal(_np(torch.as_tensor(fids)), e_f)
                np.testing.assert_array_equal(_np(torch.as_tensor(ids)), e_ids)
        finally:
            _lib.set_option("isect_pull", prev)
            rendering.reset_state()


def test_rccl_gather_ring_at_world_one(ops):
    """The RCCL transport of the frame gather, executed on the one GPU a box has (VERDICT r2 missing 1): a child
    process runs init_process_group("nccl", world_size=1, device_id=...) and pushes frames through the REAL
    FrameGatherer ring (force_collective=True: async dist.gather, event waits on two compute streams, ring reuse,
    partial tail batch, drain); gathered uint8 frames must equal the local renders and librccl must be mapped.
    The 1 -> 8 GPU curve itself is the driver's to measure: no scaling number exists until it does."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NCCL_DEBUG="VERSION", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_world1.py")], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    tail = "\n".join((p.stdout + p.stderr).splitlines()[-25:])
    assert p.returncode == 0 and "RCCL_WORLD1_OK" in p.stdout, tail
    assert "gathers=3" in p.stdout and "rccl" in p.stdout.lower(), p.stdout


def test_two_host_threads_render_on_one_device(ops):
    """Two host threads, each with its own HIP stream, render different frames of different scenes on one device
    at the same time (the intersection stage's size read-back is a pinned slot + sequence number per
    HOST THREAD, its kernel attributes are set once per device under a mutex, and the wait for the counts holds
    no GIL): every frame equals the one rendered alone."""
    import threading
    from street_crafter_amd.dist import to_uint8_frame
    from harness.caller import render_gaussians
    scenes = [make_scene(50_000, seed=31).to(DEV), make_scene(80_000, seed=32, z_range=(1.0, 40.0)).to(DEV)]
    # (cameras 0.03 rad apart at first, then 0.3: the second half lands in different VIEW SLOTS, so the two threads'
    # count launches look up, take over and update the device-side view registry at the same time)
    cams = [make_camera(640, 400, 600.0, 600.0, yaw=0.03 * i if i < 4 else 0.3 * (i - 3) * (-1) ** i,
                        shift=(0.1 * i, 0.0, 0.0)).to(DEV) for i in range(8)]

    def frame(w, f):
        with torch.no_grad():
            return to_uint8_frame(render_gaussians(scenes[w], cams[f])["rgb"])

    alone = [[frame(w, f).clone() for f in range(8)] for w in range(2)]
    torch.cuda.synchronize()
    got = [[None] * 8, [None] * 8]
    errs = []
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def worker(w):
        try:
            with torch.cuda.stream(streams[w]):
                for rep in range(3):
                    for f in range(8):
                        got[w][f] = frame(w, f)
            streams[w].synchronize()
        except BaseException as e:      # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(w,)) for w in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for w in range(2):
        for f in range(8):
            assert torch.equal(got[w][f], alone[w][f]), (w, f)


def test_scene_files_drive_the_renderer(ops, tmp_path):
    """a scene written in the reference's PLY layout (background + a posed actor with Fourier
    colour), read back and composed, renders bit-identically to the in-memory composition; and an actor
    under pose (q, t) renders exactly like the same Gaussians moved by hand into the background."""
    from street_crafter_amd import scene_io as sio
    from harness.caller import render_gaussians
    bk = sio.scene_to_submodel(make_scene(8000, seed=8, z_range=(4.0, 40.0)))
    g = torch.Generator().manual_seed(2)
    n = 1500
    act = sio.SubModel(name="obj_007", xyz=torch.randn(n, 3, generator=g) * torch.tensor([1.0, 0.6, 2.0]),
                       features_dc=torch.randn(n, 5, 3, generator=g), features_rest=torch.randn(n, 3, 3, generator=g) * 0.2,
                       scaling=torch.randn(n, 3, generator=g) * 0.3 - 2.5, rotation=torch.randn(n, 4, generator=g),
                       opacity=torch.randn(n, 1, generator=g) + 1.0, start_frame=0, end_frame=40)
    q = torch.tensor([math.cos(0.3), 0.0, math.sin(0.3), 0.0])
    t = torch.tensor([0.5, 0.2, 12.0])
    path = str(tmp_path / "point_cloud.ply")
    sio.write_ply(path, [bk, act])
    cam = make_camera(640, 400, 600.0, 600.0).to(DEV)
    mem = sio.compose_scene({"background": bk, "obj_007": act}, {"obj_007": (q, t)}, frame=10.0, device=DEV)
    fil = sio.compose_scene(sio.read_ply(path), {"obj_007": (q, t)}, frame=10.0, device=DEV)
    # the file does not carry the actor's frame range: set it as the scene metadata would
    assert fil.graph_gaussian_range == mem.graph_gaussian_range == {"background": (0, 8000), "obj_007": (8000, 9500)}
    with torch.no_grad():
        a = render_gaussians(mem.scene, cam)
        fil_models = sio.read_ply(path)
        fil_models["obj_007"].start_frame, fil_models["obj_007"].end_frame = 0, 40
        b = render_gaussians(sio.compose_scene(fil_models, {"obj_007": (q, t)}, frame=10.0, device=DEV).scene, cam)
    assert float(a["acc"].sum()) > 0
    for k in ("rgb", "acc", "depth"):
        np.testing.assert_array_equal(_np(a[k]).view(np.uint32), _np(b[k]).view(np.uint32))
    # the posed actor against the same Gaussians transformed by hand
    R = sio.quaternion_to_matrix(q[None])[0]
    moved = sio.SubModel(name="background", xyz=torch.cat([bk.xyz, act.xyz @ R.T + t]),
                         features_dc=torch.cat([bk.features_dc, sio.actor_features(act, 10.0)[:, :1]]),
                         features_rest=torch.cat([bk.features_rest, act.features_rest]),
                         scaling=torch.cat([bk.scaling, act.scaling]),
                         rotation=torch.cat([bk.rotation, sio.quaternion_raw_multiply(q[None], torch.nn.functional.normalize(act.rotation))]),
                         opacity=torch.cat([bk.opacity, act.opacity]))
    with torch.no_grad():
        c = render_gaussians(sio.compose_scene({"background": moved}, device=DEV).scene, cam)
    assert _rel_err(_np(c["rgb"]), _np(a["rgb"])) < 1e-5


# ---- knn ------------------------------------------------------------------------------------------
def test_novel_view_two_pass_frame_vs_oracle(ops):
    """render_novel_view (street_gaussian_renderer.py:136-163): foreground pass, sky pass (render_sky :80-93),
    rgb + rgb_sky * (1 - acc), clamp.  Per pass: integer outputs bit-exact vs the oracle; composite <= 1e-4 on the
    pixels stable in both passes; the fused composite + clamp + uint8 kernel equals the torch composition bit for
    bit in both of the reference's rounding modes; the rasterization()-based fast path gives the same frame."""
    from street_crafter_amd.dist import to_uint8_frame
    from harness.caller import render_novel_view, render_novel_view_u8
    from street_crafter_amd.scenes import make_street_scene
    W, H = 320, 208
    cam = make_camera(W, H, 340.0, 340.0)
    fg, sky = make_street_scene(12000, n_sky=1000, seed=3)
    as_dict = lambda sc: dict(means=sc.means.numpy(), quats=sc.quats.numpy(), scales=sc.scales.numpy(),
                              opacities=sc.opacities.numpy(), sh=sc.sh.numpy(), sh_degree=sc.sh_degree)
    exp = O.render_novel_view(as_dict(fg), as_dict(sky), cam.viewmat.numpy(), cam.K.numpy(), W, H,
                              near_plane=cam.znear, far_plane=cam.zfar, return_unstable=True)
    assert int(exp["sky"]["radii"].max()) > 60 and exp["sky"]["isect_ids"].shape[0] > 3000       # big sky splats
    fgd, skyd, camd = fg.to(DEV), sky.to(DEV), cam.to(DEV)
    with torch.no_grad():
        out = render_novel_view(fgd, skyd, camd, return_intermediates=True)
    for o, e in ((out, exp["fg"]), (out["_sky"], exp["sky"])):
        np.testing.assert_array_equal(_np(o["_radii"])[0], e["radii"])
        np.testing.assert_array_equal(_np(o["_isect_ids"]), e["isect_ids"])
        np.testing.assert_array_equal(_np(o["_flatten_ids"]), e["flatten_ids"])
        np.testing.assert_array_equal(_np(o["_isect_offsets"]), e["isect_offsets"])
    ok = ~(exp["fg"]["unstable"][0] | exp["sky"]["unstable"][0])
    assert (~ok).mean() < 0.01
    rgb = _np(out["rgb"]).transpose(1, 2, 0)
    np.testing.assert_allclose(rgb[ok], exp["rgb"][ok], rtol=0, atol=1e-4)
    assert float(np.abs(rgb - np.clip(exp["fg"]["render_colors"][0, ..., :3], 0, 1))[ok].max()) > 0.05   # the sky shows
    # one kernel instead of the torch expression: bit-identical, both rounding modes, straight from the raw images
    fg_raw = out["_render_colors"][0, ..., :3].permute(2, 0, 1)
    sky_raw = out["_sky"]["_render_colors"][0, ..., :3].permute(2, 0, 1)
    acc = out["_render_alphas"][0, ..., 0]
    for mode in ("video", "save_image"):
        got = to_uint8_frame(fg_raw, acc=acc, sky_rgb_chw=sky_raw, rounding=mode)
        np.testing.assert_array_equal(_np(got), O.quantise_u8(rgb, mode))
        single = to_uint8_frame(fg_raw, rounding=mode)
        np.testing.assert_array_equal(_np(single), O.quantise_u8(np.clip(_np(fg_raw).transpose(1, 2, 0), 0, 1), mode))
    slot = torch.empty((H, W, 3), dtype=torch.uint8, device=DEV)
    for fused in (False, True):
        u8 = render_novel_view_u8(fgd, skyd, camd, out=slot, fused=fused)
        assert u8.data_ptr() == slot.data_ptr()
        np.testing.assert_array_equal(_np(u8), O.quantise_u8(rgb, "video"))
    np.testing.assert_array_equal(_np(render_novel_view_u8(fgd, None, camd)),
                                  O.quantise_u8(np.clip(_np(fg_raw).transpose(1, 2, 0), 0, 1), "video"))


def test_knn_golden_bit_exact(golden_dir):
    from simple_knn._C import distCUDA2
    g = _load(golden_dir, "knn_small.npz")
    for k in ("pts", "dup", "line", "tiny"):
        key = "out" if k == "pts" else f"out_{k}"
        got = _np(distCUDA2(_t(g[k])))
        np.testing.assert_array_equal(got.view(np.uint32), g[key].view(np.uint32))
    assert distCUDA2(torch.zeros(0, 3, device=DEV)).shape == (0,)


def test_knn_large_vs_ckdtree():
    from scipy.spatial import cKDTree
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(5)
    # clustered + uniform mix, 300 k points: beyond what the brute-force oracle does quickly
    a = rng.normal(size=(200_000, 3)) * np.array([5.0, 0.2, 3.0])
    b = rng.uniform(-30, 30, size=(100_000, 3))
    pts = np.concatenate([a, b]).astype(np.float32)
    got = _np(distCUDA2(_t(pts)))
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4, workers=-1)
    ref = (d[:, 1:] ** 2).mean(axis=1)
    np.testing.assert_allclose(got, ref, rtol=3e-5, atol=1e-9)
    # and bit-exact against the oracle on a 20 k subset run as its own cloud
    sub = pts[:20000]
    np.testing.assert_array_equal(_np(distCUDA2(_t(sub))).view(np.uint32), KO.dist_cuda2(sub).view(np.uint32))


def test_lidar_condition_knn_scale_uses_the_hip_knn():
    """SURVEY f-1 / a14: the `use_knn_scale` branch of the LiDAR render (render_utils.py:123-127) is the LiDAR
    path's one consumer of simple_knn.distCUDA2.  A 200 k-point synthetic sweep (ground rings + walls, density
    falling with range like a spinning LiDAR): the radii computed through the product's HIP k-NN equal the ones
    from an independent exact k-NN (scipy cKDTree) and the rendered condition image is the same."""
    from scipy.spatial import cKDTree
    from simple_knn._C import distCUDA2
    from street_crafter_amd.lidar_condition import knn_point_radii, render_points
    rng = np.random.default_rng(12)
    n_ring, n_wall = 150_000, 50_000
    rng_m = 2.0 + 58.0 * rng.random(n_ring) ** 2                       # dense near the sensor, sparse far away
    az = rng.uniform(-np.pi, np.pi, n_ring)
    ground = np.stack([rng_m * np.sin(az), np.full(n_ring, 1.8) + rng.normal(scale=0.02, size=n_ring), rng_m * np.cos(az)], -1)
    wall = np.stack([np.where(rng.random(n_wall) < 0.5, -9.0, 9.0) + rng.normal(scale=0.03, size=n_wall),
                     rng.uniform(-6.0, 1.8, n_wall), rng.uniform(2.0, 60.0, n_wall)], -1)
    pts = np.concatenate([ground, wall]).astype(np.float32)
    d2_hip = distCUDA2(torch.from_numpy(pts).to(DEV)).cpu().numpy()
    dist, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    d2_ref = (dist[:, 1:] ** 2).mean(axis=1)
    np.testing.assert_allclose(d2_hip, d2_ref, rtol=2e-5, atol=1e-9)
    r_hip = knn_point_radii(pts, scale=0.05, knn_scale_down=0.8)                    # product path: HIP distCUDA2
    r_ref = np.minimum(np.sqrt(np.maximum(d2_ref, 1e-7)) * 0.8, 0.05)
    np.testing.assert_allclose(r_hip, r_ref, rtol=2e-5)
    assert 0.05 < (r_hip < 0.05).mean() < 0.95                                     # both regimes occur
    c2w = np.eye(4)
    ixt = np.array([[400.0, 0, 320.0], [0, 400.0, 200.0], [0, 0, 1.0]])
    front = pts[(pts[:, 2] > 1.0) & (np.abs(pts[:, 0] / pts[:, 2]) < 0.8) & (np.abs(pts[:, 1] / pts[:, 2]) < 0.5)]
    feat = np.concatenate([rng.random((front.shape[0], 3)), np.ones((front.shape[0], 2))], axis=-1).astype(np.float32)
    img_hip = render_points(c2w, ixt, front, feat, 400, 640, scale=0.05, use_knn_scale=True, knn_scale_down=0.8)
    d, _ = cKDTree(front.astype(np.float64)).query(front.astype(np.float64), k=4)
    img_ref = render_points(c2w, ixt, front, feat, 400, 640, scale=0.05, use_knn_scale=True, knn_scale_down=0.8,
                            knn_dist2=(d[:, 1:] ** 2).mean(axis=1))
    assert img_hip.shape == (1, 400, 640, 4) and (img_hip[..., 3] > 0).mean() > 0.02
    assert (img_hip != img_ref).any(axis=-1).mean() < 1e-3          # a disc edge may move by a pixel on a 1e-5 radius change


def test_full_resolution_s100k_against_committed_digest(ops, golden_dir):
    """S-100k at 1920x1280 against the digest the oracle produced in the build container
    (tests/golden/s100k_fullres_digest.json, tools/make_golden.py): CRC32 of every integer tensor and of
    the bit-exact float tensors, 8x8 block means of the image.  No oracle run on the box."""
    import json
    import zlib
    from harness.caller import render_gaussians
    from street_crafter_amd.scenes import make_scene_portable
    dg = json.load(open(os.path.join(golden_dir, "s100k_fullres_digest.json")))
    crc_np = lambda a: int(zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff)
    sc_cpu = make_scene_portable(100_000)
    for k, want in dg["scene"]["inputs_crc32"].items():          # the scene generator is host-independent
        assert crc_np(getattr(sc_cpu, k).numpy()) == want, f"input {k} differs on this host"
    sc = sc_cpu.to(DEV)
    cam = make_camera().to(DEV)
    with torch.no_grad():
        o = render_gaussians(sc, cam, return_intermediates=True)
    crc = lambda t: int(zlib.crc32(np.ascontiguousarray(_np(t)).tobytes()) & 0xffffffff)
    got = {"radii": o["_radii"][0], "means2d": o["_means2d"][0], "depths": o["_depths"][0], "conics": o["_conics"][0],
           "compensations": o["_compensations"][0], "opacities": o["_opacities"][0],
           "tiles_per_gauss": o["_tiles_per_gauss"][0], "isect_ids": o["_isect_ids"], "flatten_ids": o["_flatten_ids"],
           "isect_offsets": o["_isect_offsets"], "colors": o["_colors"][0]}
    assert o["_isect_ids"].numel() == dg["n_isects"]
    for k, t in got.items():
        assert crc(t) == dg["crc32"][k], k
    img = np.concatenate([_np(o["_render_colors"])[0], _np(o["_render_alphas"])[0]], axis=-1).astype(np.float64)
    H, W = img.shape[:2]
    stable = np.ones((H, W), bool)
    yx = np.asarray(dg["unstable_yx"], dtype=np.int64).reshape(-1, 2)
    stable[yx[:, 0], yx[:, 1]] = False           # threshold-unstable pixels (listed by the oracle) are left out
    assert (~stable).mean() < 0.005
    img[~stable] = 0.0
    cnt = stable.reshape(H // 160, 160, W // 240, 240).sum(axis=(1, 3))
    blocks = img.reshape(H // 160, 160, W // 240, 240, 5).sum(axis=(1, 3)) / cnt[..., None]
    want = np.asarray(dg["image_block_means_over_stable_pixels"])
    # colour / alpha block means within 2e-6; the depth channel (values up to 80) relative
    np.testing.assert_allclose(blocks[..., [0, 1, 2, 4]], want[..., [0, 1, 2, 4]], atol=2e-6, rtol=0)
    np.testing.assert_allclose(blocks[..., 3], want[..., 3], rtol=2e-6, atol=1e-5)


# ---- full-size, size-independent properties ------------------------------------------------------
def test_full_size_properties_1m(ops):
    """BASELINE config: 1 M Gaussians, 1920x1280.  Oracle for the streaming/integer stages
    (vectorised numpy is fast enough); size-independent properties for the blend."""
    from street_crafter_amd import _lib, rendering
    from harness.caller import render_gaussians
    sc = make_scene(1_000_000)
    cam = make_camera()
    scd, camd = sc.to(DEV), cam.to(DEV)
    with torch.no_grad():
        out = render_gaussians(scd, camd, return_intermediates=True)
    r, m2, d, con, comp = O.fully_fused_projection(sc.means.numpy(), sc.quats.numpy(), sc.scales.numpy(),
                                                   cam.viewmat.numpy(), cam.K.numpy(), cam.width, cam.height,
                                                   near_plane=cam.znear, far_plane=cam.zfar)
    np.testing.assert_array_equal(_np(out["_radii"])[0], r)
    np.testing.assert_array_equal(_np(out["_means2d"])[0].view(np.uint32), m2.view(np.uint32))
    np.testing.assert_array_equal(_np(out["_depths"])[0].view(np.uint32), d.view(np.uint32))
    tpg, e_ids, e_f = O.isect_tiles(m2[None], r[None], d[None], 16, 120, 80)
    np.testing.assert_array_equal(_np(out["_tiles_per_gauss"])[0], tpg[0])
    ids, fids = _np(out["_isect_ids"]), _np(out["_flatten_ids"])
    np.testing.assert_array_equal(ids, e_ids)
    np.testing.assert_array_equal(fids, e_f)
    np.testing.assert_array_equal(_np(out["_isect_offsets"]), O.isect_offset_encode(e_ids, 1, 120, 80))
    # blend: the two kernel variants agree exactly (culling never changes a pixel) ...
    prev = _lib.set_option("raster_fwd", 0)
    try:
        with torch.no_grad():
            out0 = render_gaussians(scd, camd, return_intermediates=True)
    finally:
        _lib.set_option("raster_fwd", prev)
    assert torch.equal(out0["_render_colors"], out["_render_colors"])
    assert torch.equal(out0["_render_alphas"], out["_render_alphas"])
    assert torch.equal(out0["depth"], out["depth"])
    # ... alpha in [0, 1 - 1e-4), image finite, and 4 random tiles match the oracle
    ra = out["_render_alphas"]
    assert torch.isfinite(out["_render_colors"]).all() and float(ra.min()) >= 0.0 and float(ra.max()) < 1.0
    rng = np.random.default_rng(0)
    offs = _np(out["_isect_offsets"])
    cols, opac = _np(out["_colors"]), _np(out["_opacities"])
    for _ in range(4):
        ty, tx = int(rng.integers(80)), int(rng.integers(120))
        sub_off = np.zeros((1, 1, 1), np.int32)
        s = offs[0, ty, tx]
        e = offs.reshape(-1)[ty * 120 + tx + 1] if ty * 120 + tx + 1 < 9600 else len(fids)
        sh_m2 = m2.copy()
        sh_m2[:, 0] -= tx * 16
        sh_m2[:, 1] -= ty * 16
        exp = O.rasterize_to_pixels(sh_m2[None], con[None], cols, opac, 16, 16, 16, sub_off, fids[s:e],
                                    return_unstable=True)
        got_c = _np(out["_render_colors"])[0, ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
        got_a = _np(out["_render_alphas"])[0, ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
        ok = ~exp[3][0]
        # (north_star's bar: 1e-4 abs on the pixels the oracle does not flag threshold-unstable)
        np.testing.assert_allclose(got_c[..., :3][ok], exp[0][0][..., :3][ok], rtol=0, atol=1e-4)
        np.testing.assert_allclose(got_a[ok], exp[1][0][ok], rtol=0, atol=1e-4)
