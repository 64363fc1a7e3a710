"""gsplat.rendering operator mirror for MI355X.

Same names, argument meaning, return shapes and error behaviour as the six gsplat (v1.0-v1.4
API family) operators StreetCrafter imports at
``street_gaussian/models/street_gaussian_renderer.py:204`` and calls at ``:219-280``.  Every op
launches hand-written HIP kernels through the C ABI (``include/street_crafter_amd.h``) on torch's
current stream; torch is used only for device memory, the stream handle and autograd wiring.
There is no CPU path: non-HIP tensors raise.
"""
from __future__ import annotations

import collections
import math
import threading
import weakref
from typing import Optional, Tuple

import os

import torch
from torch import Tensor

from . import _lib
from .lazy import LazyTensor as _LazyTensor

__all__ = ["fully_fused_projection", "isect_tiles", "isect_offset_encode", "spherical_harmonics",
           "rasterize_to_pixels", "rasterization"]

def _env_on(name: str, default: bool = True) -> bool:
    v = os.environ.get(name)
    return default if v is None or v == "" else v not in ("0", "false", "False", "off", "no")


class _Switches:
    """The A/B switches of the operators: ONE object, every attribute settable through its set_* function below and
    initialised from the environment, so that a user who cannot touch code can still turn a behaviour off.  None of them
    changes a result.
        SC_DEFER_ISECT=0   isect_tiles waits for the frame's intersection count inside the call (set_deferred_isect)
        SC_LAZY_IDS=0      isect_tiles' isect_ids are written by the sort instead of on first read (set_lazy_isect_ids)
      -> with BOTH off, flatten_ids / isect_ids are ordinary, fully written tensors the moment isect_tiles returns: what a
         caller needs who hands them to foreign C++ / DLPack consumers without a torch call in between
        SC_PLANAR_OUTPUT=0 render_colors interleaved under no_grad too (set_planar_output)
        SC_TILE_ORDER=0    no dispatch list for the rasterizer (set_tile_order)
        SC_VIEW_SLOTS=0    one work hint for all views (set_view_slots)
        SC_PACKED_RECORDS=0  the fused forward writes the four per-splat arrays instead of packed records (set_packed_records)"""
    __slots__ = ("tile_order", "lazy_ids", "view_slots", "packed_records", "defer_isect", "planar_out")

    def __init__(self):
        self.tile_order = _env_on("SC_TILE_ORDER")
        self.lazy_ids = _env_on("SC_LAZY_IDS")
        self.view_slots = _env_on("SC_VIEW_SLOTS")
        self.packed_records = _env_on("SC_PACKED_RECORDS")
        self.defer_isect = _env_on("SC_DEFER_ISECT")
        self.planar_out = _env_on("SC_PLANAR_OUTPUT")

    def flip(self, name: str, value: bool) -> bool:
        prev = getattr(self, name)
        setattr(self, name, bool(value))
        return prev


class _OperatorState:
    """Everything the operators remember BETWEEN calls, in one place (round 3 had eleven module-level tables).  Every
    entry is a hint the kernels verify on the device or that only orders work: dropping any of it (reset_state) never
    changes a result.  All tables are keyed by the device index first; `reset(device)` forgets one device's entries.
      prediction  (device, C, N, tile_size, tile_width, tile_height) -> (capacity, rec_capacity, super_capacity) the next
                  isect_tiles of this frame shape launches its scatter + sort with, before the host has read the counts
      history     same key -> the sizes of the last `history_len` calls (the prediction covers the largest of them)
      last_meta   same key -> (n_isects, n_records, largest super-tile) of the last call (diagnostics, tests)
      tile_work   (device, C, N, tile_width, tile_height) -> int32 [view slots, C * tiles]: what every tile walked the last
                  time a frame of this shape was rasterized (the rasterizer's scheduling hint)
      view_registry  device -> int32 [sc_view_registry_words()]: the device-side table forward axis -> view slot
      stats       how often the predicted sizes held (bench.py reports it)"""
    KEYS_MAX = 64        # frame shapes remembered in prediction / history / last_meta (pruned together)

    def __init__(self):
        self.prediction, self.history, self.last_meta = {}, {}, {}
        self.history_len = 8     # (tools/exp_camera_rig.py sets 1 for its A/B: round 2's first form)
        self.tile_work, self.view_registry = {}, {}
        self.stats = {"calls": 0, "speculative_ok": 0, "exact_relaunch": 0}
        self.bucket_cap = {}     # "v": sc_isect_bin_bucket_capacity()
        self.sched_sizes = {}    # tiles -> (words of the work-hint buffer, items of the dispatch list)

    def tables(self):
        return (("predictions", self.prediction), ("history", self.history), ("last_meta", self.last_meta),
                ("tile_work", self.tile_work), ("view_registry", self.view_registry))

    def reset(self, idx=None) -> dict:
        dropped = {}
        for name, table in self.tables():
            keys = [k for k in table if idx is None or (k[0] if isinstance(k, tuple) else k) == idx]
            for k in keys:
                table.pop(k, None)
            dropped[name] = len(keys)
        return dropped


_SWITCH = _Switches()
_STATE = _OperatorState()

# "bin"  : tile-bucketed count + in-LDS per-tile sort (default once available)
# "radix": reference-shaped count -> emit -> device-wide radix sort
_ISECT_MODE = {"mode": "bin"}


def set_isect_mode(mode: str) -> str:
    assert mode in ("bin", "radix")
    prev = _ISECT_MODE["mode"]
    _ISECT_MODE["mode"] = mode
    return prev


# ------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------
def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t: Tensor):
    """The current HIP stream of the tensor's device as a raw handle.  (torch.cuda.current_stream builds a Stream
    object: ~4 us a call, seven calls a frame; small scenes are bound by this wrapper's host time.)"""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(t.device.index if t.device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(t.device).cuda_stream


def _req(t: Tensor, name: str, dtype=torch.float32):
    # (the common case first: small frames are bound by the host time of these wrappers)
    if type(t) is Tensor and t.dtype is dtype and t.is_cuda and t.is_contiguous():
        return t
    if not isinstance(t, Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on a HIP device (got {t.device}); "
                           "street_crafter_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    return t.contiguous()


class _NoGradCtx:
    """Stands in for the autograd context when no gradient can be required: the forward bodies run
    directly, without torch.autograd.Function.apply's bookkeeping (~10 us of host time per operator,
    which is what bounds small scenes)."""
    needs_input_grad = (False,) * 16

    def save_for_backward(self, *tensors):
        pass

    def mark_non_differentiable(self, *tensors):
        pass

    def __setattr__(self, name, value):      # ctx.meta = ... etc.: nothing is kept
        pass


_NO_GRAD_CTX = _NoGradCtx()
_EMPTY = {}            # device -> a zero-element tensor (placeholder for None in save_for_backward)


def _needs_grad(*args) -> bool:
    return torch.is_grad_enabled() and any(isinstance(a, Tensor) and a.requires_grad for a in args)


def _call(fn, *args):
    """fn.apply(*args), or fn.forward on a dummy context when autograd has nothing to record."""
    if _needs_grad(*args):
        return fn.apply(*args)
    return fn.forward(_NO_GRAD_CTX, *args)


_NATIVE_AUTOGRAD = {"on": True}


def set_native_autograd(enabled: bool) -> bool:
    """The three differentiable operators' autograd plumbing: True (default) = the C++ autograd functions of the compiled
    binding layer (csrc/binding.cpp: forward and backward without a Python frame; the training step is host-bound on a busy
    box), False = the Python torch.autograd.Functions of this module (also what the ctypes path uses and what runs while a
    backward probe is set).  Same kernels, same gradients.  Returns the previous setting."""
    prev, _NATIVE_AUTOGRAD["on"] = _NATIVE_AUTOGRAD["on"], bool(enabled)
    return prev


def _native():
    """The compiled binding layer if ITS autograd functions are to be used for this call, else None."""
    if not _NATIVE_AUTOGRAD["on"] or _BACKWARD_PROBE["events"] is not None or _RAW_STREAM is None:
        return None
    return _lib.fast()


def _ws(nbytes: int, device) -> Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# Measurement hook (bench.py's train-step roofline): autograd calls the backward operators itself, so a harness
# cannot bracket them.  With a dict set here, every backward operator appends a (start, end) pair of HIP events,
# recorded on the current stream around its kernel launch, under its own name.  None (the default): nothing.
_BACKWARD_PROBE = {"events": None}


def set_backward_probe(events):
    """events: dict name -> list to append (torch.cuda.Event, torch.cuda.Event) pairs to, or None.  Returns the old one."""
    prev, _BACKWARD_PROBE["events"] = _BACKWARD_PROBE["events"], events
    return prev


class _probe:
    __slots__ = ("name", "ev")

    def __init__(self, name):
        self.name, self.ev = name, None

    def __enter__(self):
        if _BACKWARD_PROBE["events"] is not None:
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record()
        return self

    def __exit__(self, *exc):
        if self.ev is not None:
            self.ev[1].record()
            _BACKWARD_PROBE["events"].setdefault(self.name, []).append(self.ev)
        return False


# ------------------------------------------------------------------------------------------
# a1 fully_fused_projection  (renderer.py:219-232)
# ------------------------------------------------------------------------------------------
class _Projection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane,
                far_plane, radius_clip, calc_compensations):
        lib = _lib.load()
        C, N = viewmats.shape[0], means.shape[0]
        dev = means.device
        fast = _lib.fast()
        if fast is not None:
            rc, radii, means2d, depths, conics, comps = fast.projection_fwd(
                means, quats, scales, viewmats, Ks, int(width), int(height), float(eps2d), float(near_plane),
                float(far_plane), float(radius_clip), bool(calc_compensations), _stream(means))
            if rc:
                _lib.check(rc, "sc_projection_fwd")
        else:
            radii = torch.empty((C, N), dtype=torch.int32, device=dev)
            means2d = torch.empty((C, N, 2), dtype=torch.float32, device=dev)
            depths = torch.empty((C, N), dtype=torch.float32, device=dev)
            conics = torch.empty((C, N, 3), dtype=torch.float32, device=dev)
            comps = torch.empty((C, N), dtype=torch.float32, device=dev) if calc_compensations else None
            _lib.check(lib.sc_projection_fwd(_p(means), _p(quats), _p(scales), _p(viewmats), _p(Ks), C, N,
                                             int(width), int(height), float(eps2d), float(near_plane),
                                             float(far_plane), float(radius_clip), _p(radii), _p(means2d),
                                             _p(depths), _p(conics), _p(comps), _stream(means)),
                       "sc_projection_fwd")
        ctx.save_for_backward(means, quats, scales, viewmats, Ks, radii, conics,
                              comps if comps is not None else torch.empty(0, device=dev))
        ctx.dims = (int(width), int(height), float(eps2d), bool(calc_compensations))
        ctx.mark_non_differentiable(radii)
        if comps is None:
            return radii, means2d, depths, conics
        return radii, means2d, depths, conics, comps

    @staticmethod
    def backward(ctx, v_radii, v_means2d, v_depths, v_conics, v_comps=None):
        lib = _lib.load()
        means, quats, scales, viewmats, Ks, radii, conics, comps = ctx.saved_tensors
        width, height, eps2d, has_comp = ctx.dims
        C, N = viewmats.shape[0], means.shape[0]
        dev = means.device

        def z(t, shape):
            return torch.zeros(shape, dtype=torch.float32, device=dev) if t is None else t.contiguous()

        v_means2d = z(v_means2d, (C, N, 2))
        v_depths = z(v_depths, (C, N))
        v_conics = z(v_conics, (C, N, 3))
        if has_comp:
            v_comps = z(v_comps, (C, N))
        fast = _lib.fast()
        if fast is not None:
            with _probe("projection_bwd"):
                rc, v_means, v_quats, v_scales = fast.projection_bwd(
                    means, quats, scales, viewmats, Ks, width, height, eps2d, radii, conics, comps if has_comp else None,
                    v_means2d, v_depths, v_conics, v_comps if has_comp else None, _stream(means))
            if rc:
                _lib.check(rc, "sc_projection_bwd")
            return (v_means if ctx.needs_input_grad[0] else None,
                    v_quats if ctx.needs_input_grad[1] else None,
                    v_scales if ctx.needs_input_grad[2] else None,
                    None, None, None, None, None, None, None, None, None)
        v_means = torch.empty_like(means)
        v_quats = torch.empty_like(quats)
        v_scales = torch.empty_like(scales)
        with _probe("projection_bwd"):
            _lib.check(lib.sc_projection_bwd(_p(means), _p(quats), _p(scales), _p(viewmats), _p(Ks), C, N, width,
                                             height, eps2d, _p(radii), _p(conics),
                                             _p(comps) if has_comp else None, _p(v_means2d), _p(v_depths),
                                             _p(v_conics), _p(v_comps) if has_comp else None, _p(v_means),
                                             _p(v_quats), _p(v_scales), _stream(means)),
                       "sc_projection_bwd")
        return (v_means if ctx.needs_input_grad[0] else None,
                v_quats if ctx.needs_input_grad[1] else None,
                v_scales if ctx.needs_input_grad[2] else None,
                None, None, None, None, None, None, None, None, None)


def fully_fused_projection(means: Tensor, covars: Optional[Tensor], quats: Optional[Tensor],
                           scales: Optional[Tensor], viewmats: Tensor, Ks: Tensor, width: int,
                           height: int, eps2d: float = 0.3, near_plane: float = 0.01,
                           far_plane: float = 1e10, radius_clip: float = 0.0, packed: bool = False,
                           sparse_grad: bool = False, calc_compensations: bool = False):
    """means [N,3], quats [N,4] (wxyz), scales [N,3], viewmats [C,4,4], Ks [C,3,3] ->
    (radii i32[C,N], means2d [C,N,2], depths [C,N], conics [C,N,3], compensations [C,N] | None)."""
    if covars is not None:
        raise NotImplementedError("covars= is not supported: the reference passes quats/scales "
                                  "(street_gaussian_renderer.py:219-224)")
    if packed:
        raise NotImplementedError("packed=True is not supported: the reference passes packed=False "
                                  "(street_gaussian_renderer.py:228)")
    assert quats is not None and scales is not None, "quats and scales are required"
    means = _req(means, "means")
    quats = _req(quats, "quats")
    scales = _req(scales, "scales")
    viewmats = _req(viewmats, "viewmats")
    Ks = _req(Ks, "Ks")
    N = means.shape[0]
    C = viewmats.shape[0]
    assert means.shape == (N, 3), means.shape
    assert quats.shape == (N, 4), quats.shape
    assert scales.shape == (N, 3), scales.shape
    assert viewmats.shape == (C, 4, 4), viewmats.shape
    assert Ks.shape == (C, 3, 3), Ks.shape
    nat = _native() if _needs_grad(means, quats, scales) else None
    if nat is not None:
        out = nat.projection_autograd(means, quats, scales, viewmats, Ks, int(width), int(height), float(eps2d),
                                      float(near_plane), float(far_plane), float(radius_clip), bool(calc_compensations),
                                      _RAW_STREAM)
    else:
        out = _call(_Projection, means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane,
                                far_plane, radius_clip, calc_compensations)
    if N and C:
        out[1]._sc_viewmats = viewmats           # means2d carries the cameras to isect_tiles (view slots)
    if calc_compensations:
        return out
    return (*out, None)


# ------------------------------------------------------------------------------------------
# a3 isect_tiles  (renderer.py:243-252)
# ------------------------------------------------------------------------------------------
@torch.no_grad()
def isect_tiles(means2d: Tensor, radii: Tensor, depths: Tensor, tile_size: int, tile_width: int,
                tile_height: int, sort: bool = True, packed: bool = False,
                n_cameras: Optional[int] = None, camera_ids: Optional[Tensor] = None,
                gaussian_ids: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (tiles_per_gauss i32[C,N], isect_ids i64[I] (sorted), flatten_ids i32[I])."""
    if packed:
        raise NotImplementedError("packed=True is not supported (reference passes packed=False)")
    lib = _lib.load()
    view_cams = getattr(means2d, "_sc_viewmats", None)
    # (detach only what autograd tracks: a detach is ~2 us of host time, and small frames are bound by this wrapper)
    means2d = _req(means2d.detach() if means2d.requires_grad else means2d, "means2d")
    radii = _req(radii, "radii", torch.int32)
    depths = _req(depths.detach() if depths.requires_grad else depths, "depths")
    C, N = radii.shape
    assert means2d.shape == (C, N, 2), means2d.shape
    assert depths.shape == (C, N), depths.shape
    if n_cameras is not None:
        assert int(n_cameras) == C, (n_cameras, C)
    dev = means2d.device
    st = _stream(means2d)
    mode = _ISECT_MODE["mode"] if sort else "radix"
    if mode == "bin":
        res = _isect_tiles_bin(lib, means2d, radii, depths, C, N, tile_size, tile_width, tile_height,
                               None, None, st, viewmats=view_cams, defer=True)
        if res is not None:
            return res[:3]
    return _isect_tiles_radix(lib, means2d, radii, depths, C, N, tile_size, tile_width, tile_height, sort, st)


def _isect_tiles_radix(lib, means2d, radii, depths, C, N, tile_size, tile_width, tile_height, sort, st):
    """The reference-shaped route: count -> emit -> device-wide radix sort (sort=False, frames outside the bucketed
    path's limits, and the on-GPU cross-check of the bucketed route)."""
    dev = means2d.device
    tiles_per_gauss = torch.empty((C, N), dtype=torch.int32, device=dev)
    total_dev = torch.empty(1, dtype=torch.int64, device=dev)
    wsb = lib.sc_isect_workspace_bytes(C * N)
    ws = _ws(wsb, dev)
    _lib.check(lib.sc_isect_count(_p(means2d), _p(radii), C, N, int(tile_size), int(tile_width),
                                  int(tile_height), _p(tiles_per_gauss), _p(total_dev), _p(ws), ws.numel(), st),
               "sc_isect_count")
    n_isects = int(total_dev.item())    # the one unavoidable D2H read (sizes the outputs)
    _check_isect_count(n_isects, C, N, tile_width, tile_height)
    isect_ids = torch.empty(n_isects, dtype=torch.int64, device=dev)
    flatten_ids = torch.empty(n_isects, dtype=torch.int32, device=dev)
    if n_isects:
        _lib.check(lib.sc_isect_emit(_p(means2d), _p(radii), _p(depths), C, N, int(tile_size), int(tile_width),
                                     int(tile_height), _p(tiles_per_gauss), n_isects, _p(isect_ids),
                                     _p(flatten_ids), _p(ws), ws.numel(), st), "sc_isect_emit")
        if sort:
            n_tiles = tile_width * tile_height
            tile_bits = int(math.floor(math.log2(n_tiles))) + 1
            cam_bits = int(math.floor(math.log2(C))) + 1
            tmp_k = torch.empty_like(isect_ids)
            tmp_v = torch.empty_like(flatten_ids)
            sws = _ws(lib.sc_radix_sort_workspace_bytes(n_isects), dev)
            _lib.check(lib.sc_radix_sort_pairs_u64_i32(_p(isect_ids), _p(flatten_ids), _p(tmp_k), _p(tmp_v),
                                                       n_isects, 32 + tile_bits + cam_bits, _p(sws),
                                                       sws.numel(), st), "sc_radix_sort_pairs_u64_i32")
    return tiles_per_gauss, isect_ids, flatten_ids


_WARNED = set()


def _warn_once(key, msg, *args):
    """One WARNING per process and key (logging, logger "street_crafter_amd")."""
    if key not in _WARNED:
        _WARNED.add(key)
        import logging
        logging.getLogger("street_crafter_amd").warning(msg, *args)


def _check_isect_count(n_isects, C, N, tile_width, tile_height):
    """isect_offsets / positions in flatten_ids are int32, as in gsplat (which wraps around silently here)."""
    if n_isects > 0x7fffffff:
        raise RuntimeError(f"isect_tiles: {n_isects} tile intersections exceed the int32 range of isect_offsets "
                           f"({C} cameras x {N} Gaussians on {tile_width}x{tile_height} tiles); render fewer "
                           "cameras per call or cull the scene")


# Sizes seen on the previous call with the same shape: lets the bucket scatter + per-tile sort be
# enqueued with predicted buffer sizes BEFORE the host has read the counts back, so the GPU never
# idles on the host round-trip.  The kernels verify the prediction on the device (see
# sc_isect_bin_sort) and the wrapper retries with exact sizes when it was too small.


def set_tile_order(enabled: bool) -> bool:
    """Longest-running-tile-first dispatch of the rasterizer (A/B switch; results are identical either way).
    Returns the previous setting."""
    prev, _SWITCH.tile_order = _SWITCH.tile_order, bool(enabled)
    return prev


def set_packed_records(enabled: bool) -> bool:
    """A/B switch of the fused forward's packed rasterizer records (results are identical either way).  Returns the
    previous setting."""
    prev, _SWITCH.packed_records = _SWITCH.packed_records, bool(enabled)
    return prev


def set_view_slots(enabled: bool) -> bool:
    """The rasterizer's work hint per VIEW (A/B switch; results are identical either way): a rig's cameras rendered
    in turn each find the hint their own last frame left.  Returns the previous setting."""
    prev, _SWITCH.view_slots = _SWITCH.view_slots, bool(enabled)
    return prev


def _view_registry(dev) -> Optional[Tensor]:
    """The device-side table forward axis -> view slot of `dev` (sc_isect_bin_count looks the frame's camera up in
    it, no host round trip), or None when the dispatch list / the slots are off."""
    if not (_SWITCH.view_slots and _SWITCH.tile_order):
        return None
    reg = _STATE.view_registry.get(dev.index)
    if reg is None:
        reg = _STATE.view_registry[dev.index] = torch.zeros(_lib.load().sc_view_registry_words(), dtype=torch.int32, device=dev)
    return reg


def _tile_work(dev, C, N, tile_width, tile_height) -> Tensor:
    """The rasterizer's per-tile work hint of this frame shape AND Gaussian count: the foreground and the sky pass
    of a novel-view frame have the same frame shape and must not feed each other's dispatch list.  At most 8
    buffers are kept (densification changes N every few hundred training steps)."""
    key = (dev.index, int(C), int(N), int(tile_width), int(tile_height))
    t = _STATE.tile_work.pop(key, None)
    if t is None:
        # one bank of C * tiles words per view slot (the kernels pick the bank: sc_common.h)
        t = torch.zeros(_lib.load().sc_view_slots() * int(C) * int(tile_width) * int(tile_height), dtype=torch.int32,
                        device=dev)
        while len(_STATE.tile_work) >= 8:
            _STATE.tile_work.pop(next(iter(_STATE.tile_work)))
    _STATE.tile_work[key] = t                  # (re-inserted: dicts keep insertion order, the first key is the oldest)
    return t


def set_lazy_isect_ids(enabled: bool) -> bool:
    """isect_tiles' `isect_ids` on the tile-bucketed path: True (default) = a LazyTensor filled on first use
    (street_crafter_amd/lazy.py: the reference's path never reads it), False = written by the sort as before.
    Returns the previous setting."""
    prev, _SWITCH.lazy_ids = _SWITCH.lazy_ids, bool(enabled)
    return prev

_PINNED_META = threading.local()   # .slots: device index -> [pinned int64[8] the device publishes meta into, its
                                   # numpy view, seq] of THIS host thread




def set_deferred_isect(enabled: bool) -> bool:
    """isect_tiles' host wait for the frame's intersection count: True (default) = deferred to the first observation of
    `flatten_ids` / `isect_ids` (normally inside rasterize_to_pixels, by when the count has long arrived: lazy.py), False =
    inside isect_tiles as before.  Applies when a prediction of the buffer sizes exists (from the second call of a frame
    shape on).  Results are identical either way.  Returns the previous setting."""
    prev, _SWITCH.defer_isect = _SWITCH.defer_isect, bool(enabled)
    return prev


class _PendingIsect:
    """The part of one isect_tiles call that needs the frame's counts on the host: wait, check of the predicted launch
    (exact relaunch when it was too small), bookkeeping for the next prediction, and the final length of the two id
    tensors.  Shared by the `flatten_ids` and `isect_ids` LazyTensors of the call (each holds it; it holds them weakly)
    and by the pinned meta slot of the host thread (weakly): the next isect_tiles call of the thread settles it first,
    because the slot's words are overwritten by every count phase."""
    __slots__ = ("settle", "flat_ref", "ids_ref", "done", "lock", "flat_plain", "error", "__weakref__")

    def __init__(self, settle):
        self.settle, self.flat_ref, self.ids_ref = settle, None, None
        self.done, self.lock, self.flat_plain, self.error = False, threading.Lock(), None, None

    def resolve(self, _tensor=None):
        with self.lock:
            if self.done:
                return
            if self.error is not None:         # (e.g. more than 2^31 - 1 intersections: every observation says so)
                raise self.error
            settle, self.settle = self.settle, None
            try:
                fids, ids_buf, after = settle()
            except BaseException as e:
                self.error = e
                raise
            self.flat_plain = fids
            with torch._C.DisableTorchFunctionSubclass():
                for ref, src in ((self.flat_ref, fids), (self.ids_ref, ids_buf)):
                    t = ref() if ref is not None else None
                    if t is not None and src is not None:
                        t.set_(src)
                        t.__dict__["_sc_resolve"] = None      # (settled through the other tensor / the next call)
            self.done = True
            if after is not None:
                after()


def _bin_launch_ran(capacities, n_isects, n_records, max_super) -> bool:
    """True iff a bucket scatter + sort launched with `capacities` = (capacity, rec_capacity, super_capacity)
    passed the device-side size check, i.e. ran in full.  Must mirror the kernels' test
    (`meta[0] > capacity || meta[2] > rec_capacity || meta[3] > super_capacity`, isect_bin.hip) exactly:
    sc_isect_bin_sort hands the kernels the SAME unrounded numbers it is given."""
    return n_isects <= capacities[0] and n_records <= capacities[1] and max_super <= capacities[2]


def _isect_tiles_bin(lib, means2d, radii, depths, C, N, tile_size, tile_width, tile_height,
                     tiles_per_gauss, total_dev, st, want_ids=True, viewmats=None, defer=False):
    """-> (tiles_per_gauss, isect_ids | None, flatten_ids, isect_offsets), or None when the shape is outside
    the tile-bucketed path's limits.  want_ids=False skips the 8 B x I key array altogether (the fused
    rasterization() forward never reads it).  Host threads do not serialise each other: the count phase reports
    its sizes through a pinned slot + sequence number per (host thread, device), and the wait for it releases
    the GIL (sc_wait_i64) -- a host that renders two frames in flight from two threads keeps launching one frame
    while it waits for the other's counts (tests/test_gpu_parity.py::test_two_host_threads_render_on_one_device).
    defer: see set_deferred_isect (the wait, the check of the predicted launch and the outputs' lengths move to the first
    observation of the outputs; only the public isect_tiles asks for it).
    `tiles_per_gauss`: None = allocated here (the compiled binding layer allocates every output of the count phase
    in its one call)."""
    dev = means2d.device
    fast = _lib.fast()
    # meta (output sizes) comes back through host-mapped pinned memory that the device writes
    # directly, followed by a sequence number: no D2H copy and no event on the stream (include/*.h)
    slots = getattr(_PINNED_META, "slots", None)
    if slots is None:
        slots = _PINNED_META.slots = {}
    slot = slots.get(dev.index)
    if slot is None:
        host = torch.zeros(8, dtype=torch.int64, pin_memory=True)
        slot = slots[dev.index] = [host, host.numpy(), 0, host.data_ptr(), None]
    if len(slot) > 4 and slot[4] is not None:      # an earlier call of this thread whose counts nobody has looked at yet
        prev_pending = slot[4]()
        slot[4] = None
        if prev_pending is not None:
            try:
                prev_pending.resolve()
            except Exception as e:      # noqa: BLE001  (kept on the pending call: its own tensors raise it when looked at)
                _warn_once("settle_prev", "isect_tiles: settling the previous, never observed call failed (%s: %s); its own "
                           "tensors raise the error when they are looked at", type(e).__name__, e)
    meta_np, meta_ptr = slot[1], slot[3]
    slot[2] += 1
    seq = slot[2]
    # the rasterizer's dispatch order (longest-running tiles first) is built here, beside the count kernels, from
    # what every tile walked the last time (tile_size 16: the wave-per-tile rasterizer)
    want_order = _SWITCH.tile_order and int(tile_size) == 16
    work = _tile_work(dev, C, N, tile_width, tile_height) if want_order else None
    registry = None
    if (want_order and viewmats is not None and viewmats.device == dev and viewmats.dtype == torch.float32
            and viewmats.is_contiguous() and viewmats.shape == (C, 4, 4)):
        registry = _view_registry(dev)
    if registry is None:
        viewmats = None
    if fast is not None:
        rc, tpg, offsets, meta_dev, ws0, order = fast.isect_bin_count(
            means2d, radii, depths, int(tile_size), int(tile_width), int(tile_height), work, viewmats, registry,
            bool(want_order), meta_ptr, seq, st)
        if tiles_per_gauss is not None and rc == 0:
            tiles_per_gauss.copy_(tpg)            # (a caller that brought its own buffer; none in this package)
        else:
            tiles_per_gauss = tpg
    else:
        if tiles_per_gauss is None:
            tiles_per_gauss = torch.empty((C, N), dtype=torch.int32, device=dev)
        offsets = torch.empty((C, tile_height, tile_width), dtype=torch.int32, device=dev)
        meta_dev = torch.empty(4, dtype=torch.int64, device=dev)
        ws0 = _ws(lib.sc_isect_bin_workspace_bytes(C * N, C, tile_width, tile_height, -1), dev)
        order = (torch.empty(lib.sc_tile_order_len(C * tile_width * tile_height), dtype=torch.int32, device=dev)
                 if want_order else None)
        rc = lib.sc_isect_bin_count(_p(means2d), _p(radii), _p(depths), C, N, int(tile_size), int(tile_width),
                                    int(tile_height), _p(tiles_per_gauss), _p(offsets), _p(meta_dev),
                                    meta_ptr, seq, _p(ws0), ws0.numel(),
                                    _p(work) if want_order else None, _p(viewmats), _p(registry), _p(order), st)
    if want_order:
        offsets._sc_sched = (order, work)          # travels with isect_offsets to rasterize_to_pixels
    if rc == -3:     # SC_EUNSUPPORTED -> reference-shaped route
        return None
    if rc:
        _lib.check(rc, "sc_isect_bin_count")

    def read_meta():
        # wait for the sequence number (a spin in C that holds no GIL); if the GPU is far behind (or something went
        # wrong) fall back to a plain synchronising copy after 2 s
        waited = (fast.wait_i64(meta_ptr + 32, seq, 2_000_000) if fast is not None
                  else lib.sc_wait_i64(meta_ptr + 32, seq, 2_000_000))
        if waited != 0:
            # (the settle may run under another current stream than `st` -- the thread's next isect_tiles, reset_state -- and
            #  a copy on THAT stream would not be ordered behind the count kernels: wait for the producer's stream first)
            torch.cuda.ExternalStream(st, device=dev).synchronize()
            return tuple(int(v) for v in meta_dev.cpu().tolist())
        return int(meta_np[0]), int(meta_np[1]), int(meta_np[2]), int(meta_np[3])

    eager_ids = want_ids and not _SWITCH.lazy_ids

    def launch(capacity, rec_capacity, super_capacity):
        if fast is not None:
            return fast.isect_bin_sort(means2d, radii, depths, int(tile_size), int(tile_width), int(tile_height), offsets,
                                       meta_dev, ws0, int(capacity), int(rec_capacity), int(super_capacity),
                                       bool(eager_ids), st)
        ids = torch.empty(capacity, dtype=torch.int64, device=dev) if eager_ids else None
        fids = torch.empty(capacity, dtype=torch.int32, device=dev)
        ws = _ws(lib.sc_isect_bin_workspace_bytes(C * N, C, tile_width, tile_height, rec_capacity), dev)
        r = lib.sc_isect_bin_sort(_p(means2d), _p(radii), _p(depths), C, N, int(tile_size), int(tile_width),
                                  int(tile_height), _p(offsets), _p(meta_dev), _p(ws0), capacity, rec_capacity,
                                  super_capacity, _p(ids), _p(fids), _p(ws), ws.numel(), st)
        return r, ids, fids

    key = (dev.index, C, N, int(tile_size), int(tile_width), int(tile_height))
    pred = _STATE.prediction.get(key)
    rc, ids, fids = (None, None, None)
    if pred is not None:
        rc, ids, fids = launch(*pred)
        if rc == -3:
            rc = None
        elif rc != 0:
            _lib.check(rc, "sc_isect_bin_sort")

    def settle(rc=rc, ids=ids, fids=fids):
        """Everything that needs the frame's counts on the host.  -> (n_isects, ids, fids), or None when the frame is
        outside the tile-bucketed path's limits (the caller takes the reference-shaped route)."""
        n_isects, _, n_records, max_super = read_meta()     # the one host wait of a frame; GPU already has work
        _check_isect_count(n_isects, C, N, tile_width, tile_height)
        # the device ran the predicted launch iff ALL THREE of its checks passed; `_bin_launch_ran` restates those
        # checks exactly (a launch that ran in full has consumed the bucket cursors: it must never be repeated)
        _STATE.stats["calls"] += 1
        if rc is not None and _bin_launch_ran(pred, n_isects, n_records, max_super):
            _STATE.stats["speculative_ok"] += 1
        elif rc is not None:
            _STATE.stats["exact_relaunch"] += 1
        if rc is None or not _bin_launch_ran(pred, n_isects, n_records, max_super):
            if rc is not None:     # a predicted launch was enqueued and (by the device's own check) did nothing:
                # belt and braces, the cursors are re-zeroed before the exact-size launch all the same
                _lib.check(lib.sc_isect_bin_reset_cursors(_p(ws0), C * N, C, int(tile_width), int(tile_height), st),
                           "sc_isect_bin_reset_cursors")
            if _stream(means2d) != st:      # (settled late, under another current stream: allocate where the kernels run)
                with torch.cuda.stream(torch.cuda.ExternalStream(st, device=dev)):
                    rc, ids, fids = launch(n_isects, n_records, max_super)
            else:
                rc, ids, fids = launch(n_isects, n_records, max_super)
            if rc == -3:
                return None
            _lib.check(rc, "sc_isect_bin_sort")
        _STATE.last_meta[key] = (n_isects, n_records, max_super)
        # next call: 12.5 % head-room over the largest of the last 8 calls of this shape (a rig's cameras are rendered
        # in turn and see different amounts of the scene: sized by the previous call alone, every switch to a fuller view
        # missed the prediction and paid the host round trip + a second launch)
        hist = _STATE.history.get(key)
        if hist is None:
            hist = _STATE.history[key] = collections.deque(maxlen=_STATE.history_len)
            while len(_STATE.history) > _STATE.KEYS_MAX:
                # one key per distinct (device, C, N, tile grid): densification changes N every 100 training iterations
                # (train.py:292-310), so all three per-shape tables are pruned together, oldest shape first
                old = next(iter(_STATE.history))
                _STATE.history.pop(old)
                _STATE.prediction.pop(old, None)
                _STATE.last_meta.pop(old, None)
        hist.append((n_isects, n_records, max_super))
        mi, mr, ms = map(max, zip(*hist))
        ms_pred = ms + ms // 8 + 64
        # (head-room alone must not cross the capacity of the one-workgroup bucket sort: provisioned above it, every
        #  frame also launches the split kernel and ~10 k idle segment workgroups -- 8-10 us at the training resolution,
        #  whose largest bucket sits just below it.  A frame that does exceed it fails the device-side check and is
        #  relaunched with exact sizes, and the history then provisions for it.)
        cap1 = _STATE.bucket_cap.get("v")
        if cap1 is None:
            cap1 = _STATE.bucket_cap["v"] = int(lib.sc_isect_bin_bucket_capacity())
        if ms <= cap1 < ms_pred:
            ms_pred = cap1
        _STATE.prediction[key] = (mi + mi // 8 + 4096, mr + mr // 8 + 4096, ms_pred)
        return n_isects, ids, fids

    def make_fill(get_flat, n_isects_of):
        # isect_ids: allocated now, written on first use (one kernel, from flatten_ids / offsets / depths)
        def fill(buf, off=offsets, dep=depths, producer=st):
            fl, n_isects = get_flat(), n_isects_of()
            cur = _stream(buf)
            if cur != producer:            # filled from another stream than the one that sorted: order them
                ev = torch.cuda.Event()
                ev.record(torch.cuda.ExternalStream(producer, device=dev))
                torch.cuda.current_stream(dev).wait_event(ev)
            _lib.check(lib.sc_isect_ids_rebuild(fl.data_ptr(), off.data_ptr(), dep.data_ptr(), C, N,
                                                int(tile_width), int(tile_height), n_isects, buf.data_ptr(),
                                                cur), "sc_isect_ids_rebuild")
        return fill

    if defer and rc == 0 and want_ids and not eager_ids and _SWITCH.defer_isect:
        # The predicted scatter + sort are enqueued; the wait for the counts, the check of the prediction and the true
        # length of the two id tensors are settled on their first observation (lazy.py) -- on the reference's path
        # inside rasterize_to_pixels, three torch calls later, when the counts have arrived.
        from .lazy import LazyTensor
        state = {}

        def after():
            t = pending.ids_ref() if pending.ids_ref is not None else None
            if t is not None:       # set_ has bumped the version the cached offsets were filed under
                with torch._C.DisableTorchFunctionSubclass():
                    t._sc_offsets = (offsets, C, int(tile_width), int(tile_height), t._version)

        def settle_deferred():
            res = settle()
            if res is None:       # outside the bucketed path's limits after all: the reference-shaped route, now
                # (called directly: switching the process-wide mode around a nested isect_tiles would send a concurrent
                #  call of another host thread down the radix route too -- ADVICE r3)
                with torch.cuda.stream(torch.cuda.ExternalStream(st, device=dev)):
                    _, ids2, fids2 = _isect_tiles_radix(lib, means2d, radii, depths, C, N, tile_size, tile_width,
                                                        tile_height, True, st)
                state["n"] = fids2.numel()
                pend_ids = pending.ids_ref() if pending.ids_ref is not None else None
                if pend_ids is not None:
                    pend_ids.__dict__["_sc_fill"] = None          # (the radix route has written the keys)
                return fids2, ids2, after
            n_isects, _, fids_ = res
            state["n"] = n_isects
            return fids_[:n_isects], torch.empty(n_isects, dtype=torch.int64, device=dev), after

        pending = _PendingIsect(settle_deferred)
        flatten_ids = LazyTensor(fids[:0], None, pending.resolve)
        isect_ids = LazyTensor(torch.empty(0, dtype=torch.int64, device=dev),
                               make_fill(lambda: pending.flat_plain, lambda: state["n"]), pending.resolve)
        pending.flat_ref, pending.ids_ref = weakref.ref(flatten_ids), weakref.ref(isect_ids)
        with torch._C.DisableTorchFunctionSubclass():
            isect_ids._sc_offsets = (offsets, C, int(tile_width), int(tile_height), isect_ids._version)
        slot[4] = weakref.ref(pending)
        return tiles_per_gauss, isect_ids, flatten_ids, offsets

    res = settle()
    if res is None:
        return None
    n_isects, ids, fids = res
    flatten_ids = fids[:n_isects]
    isect_ids = None
    if want_ids:
        if eager_ids:
            isect_ids = ids[:n_isects]
        else:
            from .lazy import LazyTensor
            isect_ids = LazyTensor(torch.empty(n_isects, dtype=torch.int64, device=dev),
                                   make_fill(lambda: flatten_ids, lambda: n_isects))
        # the bucket scan already IS isect_offset_encode's result: remember it on the tensor object so
        # the caller's next call (renderer.py:253) does not re-read the 8 B x I key array
        isect_ids._sc_offsets = (offsets, C, int(tile_width), int(tile_height), isect_ids._version)
    return tiles_per_gauss, isect_ids, flatten_ids, offsets


# ------------------------------------------------------------------------------------------
# a4 isect_offset_encode  (renderer.py:253)
# ------------------------------------------------------------------------------------------
@torch.no_grad()
def isect_offset_encode(isect_ids: Tensor, n_cameras: int, tile_width: int, tile_height: int) -> Tensor:
    lib = _lib.load()
    cached = getattr(isect_ids, "_sc_offsets", None)
    if cached is not None and cached[1:4] == (int(n_cameras), int(tile_width), int(tile_height)):
        # still the keys isect_tiles produced?  A LazyTensor knows whether anything has seen its contents (no torch
        # call needed to ask: an attribute read through a tensor subclass costs ~4 us of host time); an ordinary
        # tensor (eager keys) is checked by its version counter
        if (isect_ids.__dict__.get("_sc_touched") is None if type(isect_ids) is _LazyTensor
                else isect_ids._version == cached[4]):
            return cached[0]
    isect_ids = _req(isect_ids, "isect_ids", torch.int64)
    dev = isect_ids.device
    offsets = torch.empty((n_cameras, tile_height, tile_width), dtype=torch.int32, device=dev)
    _lib.check(lib.sc_isect_offsets(_p(isect_ids), isect_ids.numel(), int(n_cameras), int(tile_width),
                                    int(tile_height), _p(offsets), _stream(isect_ids)), "sc_isect_offsets")
    return offsets


# ------------------------------------------------------------------------------------------
# a6 spherical_harmonics  (renderer.py:259)
# ------------------------------------------------------------------------------------------
class _SphericalHarmonics(torch.autograd.Function):
    @staticmethod
    def forward(ctx, degree, dirs, coeffs, masks):
        lib = _lib.load()
        M = dirs.numel() // 3
        K = coeffs.shape[-2]
        fast = _lib.fast()
        if fast is not None:
            rc, colors = fast.sh_fwd(int(degree), dirs, coeffs, masks, _stream(dirs))
            if rc:
                _lib.check(rc, "sc_sh_fwd")
        else:
            colors = torch.empty(dirs.shape, dtype=torch.float32, device=dirs.device)
            _lib.check(lib.sc_sh_fwd(int(degree), _p(dirs), _p(coeffs), _p(masks), M, K, _p(colors),
                                     _stream(dirs)), "sc_sh_fwd")
        ctx.save_for_backward(dirs, coeffs, masks if masks is not None else torch.empty(0, device=dirs.device))
        ctx.meta = (int(degree), M, K, masks is not None)
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        lib = _lib.load()
        dirs, coeffs, masks = ctx.saved_tensors
        degree, M, K, has_mask = ctx.meta
        v_colors = v_colors.contiguous()
        need_dirs = ctx.needs_input_grad[1]
        fast = _lib.fast()
        if fast is not None:
            with _probe("spherical_harmonics_bwd"):
                rc, v_coeffs, v_dirs = fast.sh_bwd(degree, dirs, coeffs, masks if has_mask else None, v_colors,
                                                   bool(need_dirs), _stream(dirs))
            if rc:
                _lib.check(rc, "sc_sh_bwd")
            return None, v_dirs, (v_coeffs if ctx.needs_input_grad[2] else None), None
        v_coeffs = torch.empty_like(coeffs)
        v_dirs = torch.empty_like(dirs) if need_dirs else None
        with _probe("spherical_harmonics_bwd"):
            _lib.check(lib.sc_sh_bwd(degree, _p(dirs), _p(coeffs), _p(masks) if has_mask else None, M, K,
                                     _p(v_colors), _p(v_coeffs), _p(v_dirs), _stream(dirs)), "sc_sh_bwd")
        return None, v_dirs, (v_coeffs if ctx.needs_input_grad[2] else None), None


def spherical_harmonics(degrees_to_use: int, dirs: Tensor, coeffs: Tensor,
                        masks: Optional[Tensor] = None) -> Tensor:
    """dirs [...,3] (not normalised by the caller, renderer.py:256), coeffs [...,K,3], masks [...]."""
    assert 0 <= degrees_to_use <= 4, degrees_to_use
    assert (degrees_to_use + 1) ** 2 <= coeffs.shape[-2], coeffs.shape
    assert dirs.shape[:-1] == coeffs.shape[:-2], (dirs.shape, coeffs.shape)
    assert dirs.shape[-1] == 3 and coeffs.shape[-1] == 3, (dirs.shape, coeffs.shape)
    dirs = _req(dirs, "dirs")
    coeffs = _req(coeffs, "coeffs")
    if masks is not None:
        assert masks.shape == dirs.shape[:-1], masks.shape
        if not masks.is_cuda:
            raise RuntimeError("masks must live on a HIP device")
        if masks.dtype == torch.bool:
            masks = masks.contiguous().view(torch.uint8)      # same bytes (0/1), no conversion kernel
        elif masks.dtype != torch.uint8:
            masks = masks.to(torch.uint8).contiguous()
        else:
            masks = masks.contiguous()
    nat = _native() if _needs_grad(dirs, coeffs) else None
    if nat is not None:
        return nat.sh_autograd(int(degrees_to_use), dirs, coeffs, masks, _RAW_STREAM)
    return _call(_SphericalHarmonics, int(degrees_to_use), dirs, coeffs, masks)


# ------------------------------------------------------------------------------------------
# a9 rasterize_to_pixels  (renderer.py:267-280)
# ------------------------------------------------------------------------------------------


def set_planar_output(enabled: bool) -> bool:
    """Storage of rasterize_to_pixels' `render_colors` under no_grad (tile_size 16, 3 or 4 channels): True (default) = one plane
    per channel, [C][D][H][W], returned as the permuted [C,H,W,D] view; False = interleaved as before.  Values, shapes and every
    indexing expression are the same; what the reference's caller does with the result right behind the operator
    (renderer.py:282-300: `[..., :-1]`, `[..., -1:] / alpha`, clamp, `.permute(2, 0, 1)`) becomes dense kernels instead of
    strided ones (-10 us per 1920x1280 frame, tools/exp_glue_layout.py).  The tensor is not `is_contiguous()` in this form
    (`.view(-1)` on it needs `.reshape`).  Training (anything requires grad) always takes the interleaved form.  Returns the
    previous setting."""
    prev, _SWITCH.planar_out = _SWITCH.planar_out, bool(enabled)
    return prev


_RASTER_SIDE = {}      # device index -> {raw handle of the caller's stream (None = any): torch stream the rasterizer runs on}


def set_raster_side_stream(device, side, main=None):
    """Frame-loop option (dist.make_stream; bench.py --raster-cus / --raster-priority): the INFERENCE rasterizer of `device`
    is launched on the stream `side` -- typically one confined to a subset of the CUs or of lower priority -- instead of on
    the caller's current stream, fenced by events on both sides, so that the results and their ordering are unchanged.
    With several frames in flight the one-wave workgroups of the VALU-bound rasterizer otherwise take every free wave slot
    and the next frame's latency-bound intersection kernels (8..16-wave workgroups) wait for whole CUs to drain.
    `main`: only rasterizer calls issued under this stream (a torch stream) use `side`; None = calls under any stream.
    side=None removes the entry.  Returns the previous side stream of that (device, main)."""
    idx = device if isinstance(device, int) else torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    tab = _RASTER_SIDE.setdefault(idx, {})
    key = None if main is None else int(main.cuda_stream)
    prev = tab.get(key)
    if side is None:
        tab.pop(key, None)
        if not tab:
            _RASTER_SIDE.pop(idx, None)
    else:
        tab[key] = side
    return prev


def reset_state(device=None) -> dict:
    """Forgets everything the operators remember BETWEEN calls -- the size predictions and their history, the
    rasterizer's per-view work hints and view registries, the last counts -- for `device` (an index or torch.device;
    None = all devices).  None of it is ever needed for a correct result (every piece is a hint that the kernels verify
    or that only orders work); a long-running process that is done with a scene may call this to hand the hint buffers
    (32 B x tiles per frame shape, at most 8 shapes) back, and tests use it to start from a cold state.  The A/B
    switches (set_tile_order, set_deferred_isect, ...) are not state and keep their values.  Call it between frames:
    an isect_tiles call whose outputs have not been looked at yet is settled first.
    -> how many entries of each table were dropped."""
    idx = None if device is None else (device if isinstance(device, int) else torch.device(device).index)
    slots = getattr(_PINNED_META, "slots", None) or {}
    for d, slot in slots.items():
        if (idx is None or d == idx) and len(slot) > 4 and slot[4] is not None:
            pend, slot[4] = slot[4](), None
            if pend is not None:
                try:
                    pend.resolve()
                except Exception as e:      # noqa: BLE001  (stays on the pending call's own tensors)
                    _warn_once("settle_reset", "reset_state: settling a never observed isect_tiles call failed (%s: %s)",
                               type(e).__name__, e)
    return _STATE.reset(idx)


def _sched_of(isect_offsets, n_tiles):
    """(tile_order, tile_work) the intersection stage left on this isect_offsets tensor, or (None, None)."""
    sched = getattr(isect_offsets, "_sc_sched", None) if _SWITCH.tile_order else None
    if sched is None:
        return None, None
    want = _STATE.sched_sizes.get(n_tiles)
    if want is None:       # (two foreign calls per frame otherwise)
        if len(_STATE.sched_sizes) > 64:
            _STATE.sched_sizes.clear()
        want = _STATE.sched_sizes[n_tiles] = (_lib.load().sc_view_slots() * n_tiles, _lib.load().sc_tile_order_len(n_tiles))
    if sched[1].numel() != want[0] or sched[0].device != isect_offsets.device or sched[0].numel() != want[1]:
        return None, None
    return sched


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities, backgrounds, masks, width, height, tile_size,
                isect_offsets, flatten_ids, absgrad, means2d_obj):
        lib = _lib.load()
        C, N = opacities.shape
        D = colors.shape[-1]
        th, tw = isect_offsets.shape[1], isect_offsets.shape[2]
        dev = means2d.device
        # last_ids only feeds the backward replay: inference (no input requires grad) skips it
        needs_bwd = any(ctx.needs_input_grad[:5])
        order, work = _sched_of(isect_offsets, C * tw * th)
        fast = _lib.fast()
        st = _stream(means2d)
        side = cur = None
        if _RASTER_SIDE and not needs_bwd:         # set_raster_side_stream: the kernel runs on another stream, fenced by events
            tab = _RASTER_SIDE.get(dev.index)
            side = (tab.get(st) or tab.get(None)) if tab else None
            if side is not None:
                # (the outputs are allocated under the caller's stream as always: every later use of them on that stream is
                #  behind the second event, and the inputs outlive the kernel for the same reason)
                cur = torch.cuda.current_stream(dev)
                ev = torch.cuda.Event()
                ev.record(cur)
                side.wait_event(ev)
                st = side.cuda_stream
        rc = -3
        last_ids = None
        if _SWITCH.planar_out and not needs_bwd and int(tile_size) == 16 and D in (3, 4):
            # inference: one plane per channel behind the same [C,H,W,D] indexing (set_planar_output); -3 = not this kernel
            if fast is not None:
                rc, render_colors, render_alphas = fast.rasterize_fwd_planar(
                    means2d, conics, colors, opacities, backgrounds, masks, int(width), int(height), int(tile_size),
                    isect_offsets, flatten_ids, order, work, st)
            else:
                planes = torch.empty((C, D, height, width), dtype=torch.float32, device=dev)
                render_alphas = torch.empty((C, height, width, 1), dtype=torch.float32, device=dev)
                rc = lib.sc_rasterize_fwd_planar(_p(means2d), _p(conics), _p(colors), _p(opacities), _p(backgrounds),
                                                 _p(masks), C, N, D, int(width), int(height), int(tile_size), tw, th,
                                                 _p(isect_offsets), _p(flatten_ids), flatten_ids.numel(), _p(planes),
                                                 _p(render_alphas), _p(order), _p(work), st)
                render_colors = planes.permute(0, 2, 3, 1)
            if rc not in (0, -3):
                _lib.check(rc, "sc_rasterize_fwd_planar")
        if rc == 0:
            pass
        elif fast is not None:
            rc, render_colors, render_alphas, last_ids = fast.rasterize_fwd(
                means2d, conics, colors, opacities, backgrounds, masks, int(width), int(height), int(tile_size),
                isect_offsets, flatten_ids, bool(needs_bwd), order, work, st)
            if rc:
                _lib.check(rc, "sc_rasterize_fwd")
        else:
            render_colors = torch.empty((C, height, width, D), dtype=torch.float32, device=dev)
            render_alphas = torch.empty((C, height, width, 1), dtype=torch.float32, device=dev)
            last_ids = torch.empty((C, height, width), dtype=torch.int32, device=dev) if needs_bwd else None
            _lib.check(lib.sc_rasterize_fwd(_p(means2d), _p(conics), _p(colors), _p(opacities), _p(backgrounds),
                                            _p(masks), C, N, D, int(width), int(height), int(tile_size), tw, th,
                                            _p(isect_offsets), _p(flatten_ids), flatten_ids.numel(),
                                            _p(render_colors), _p(render_alphas), _p(last_ids), _p(order), _p(work),
                                            st),
                       "sc_rasterize_fwd")
        if side is not None:
            ev = torch.cuda.Event()
            ev.record(side)
            cur.wait_event(ev)
        e = _EMPTY.get(dev)
        if e is None:
            e = _EMPTY[dev] = torch.empty(0, device=dev)
        ctx.save_for_backward(means2d, conics, colors, opacities, backgrounds if backgrounds is not None else e,
                              masks if masks is not None else e, isect_offsets, flatten_ids, render_alphas,
                              last_ids if last_ids is not None else e)
        ctx.meta = (int(width), int(height), int(tile_size), bool(absgrad), backgrounds is not None,
                    masks is not None)
        ctx.means2d_obj = means2d_obj
        ctx.tile_order = order
        return render_colors, render_alphas

    @staticmethod
    def backward(ctx, v_render_colors, v_render_alphas):
        lib = _lib.load()
        (means2d, conics, colors, opacities, backgrounds, masks, isect_offsets, flatten_ids, render_alphas,
         last_ids) = ctx.saved_tensors
        width, height, tile_size, absgrad, has_bg, has_mask = ctx.meta
        C, N = opacities.shape
        D = colors.shape[-1]
        th, tw = isect_offsets.shape[1], isect_offsets.shape[2]
        dev = means2d.device
        v_render_colors = v_render_colors.contiguous()
        v_render_alphas = v_render_alphas.contiguous()
        fast = _lib.fast()
        if fast is not None:
            with _probe("rasterize_to_pixels_bwd"):
                rc, v_means2d, v_conics, v_colors, v_opacities, v_abs = fast.rasterize_bwd(
                    means2d, conics, colors, opacities, backgrounds if has_bg else None, masks if has_mask else None,
                    width, height, tile_size, isect_offsets, flatten_ids, render_alphas, last_ids, v_render_colors,
                    v_render_alphas, bool(absgrad), ctx.tile_order, _stream(means2d))
            if rc:
                _lib.check(rc, "sc_rasterize_bwd")
            if absgrad:
                ctx.means2d_obj.tensor.absgrad = v_abs          # gsplat contract (street_gaussian_model.py:505-506)
            v_bg = None
            if has_bg and ctx.needs_input_grad[4]:
                v_bg = (v_render_colors * (1.0 - render_alphas)).sum(dim=(1, 2))
            return (v_means2d, v_conics, v_colors, v_opacities, v_bg, None, None, None, None, None, None, None, None)
        # the kernel accumulates with float atomics: ONE zero-fill for all five gradient buffers
        sizes = (2 * C * N, 3 * C * N, D * C * N, C * N, 2 * C * N if absgrad else 0)
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        parts = torch.split(flat, sizes)
        v_means2d = parts[0].view(C, N, 2)
        v_conics = parts[1].view(C, N, 3)
        v_colors = parts[2].view(C, N, D)
        v_opacities = parts[3].view(C, N)
        v_abs = parts[4].view(C, N, 2) if absgrad else None
        with _probe("rasterize_to_pixels_bwd"):
            _lib.check(lib.sc_rasterize_bwd(_p(means2d), _p(conics), _p(colors), _p(opacities),
                                            _p(backgrounds) if has_bg else None, _p(masks) if has_mask else None,
                                            C, N, D, width, height, tile_size, tw, th, _p(isect_offsets),
                                            _p(flatten_ids), flatten_ids.numel(), _p(render_alphas), _p(last_ids),
                                            _p(v_render_colors), _p(v_render_alphas), _p(v_abs), _p(v_means2d),
                                            _p(v_conics), _p(v_colors), _p(v_opacities), _p(ctx.tile_order),
                                            _stream(means2d)),
                       "sc_rasterize_bwd")
        if absgrad:
            # gsplat contract: the tensor object the CALLER passed gets an `.absgrad` attribute
            # (read at street_gaussian/models/street_gaussian_model.py:505-506)
            ctx.means2d_obj.tensor.absgrad = v_abs
        v_bg = None
        if has_bg and ctx.needs_input_grad[4]:
            v_bg = (v_render_colors * (1.0 - render_alphas)).sum(dim=(1, 2))
        return (v_means2d, v_conics, v_colors, v_opacities, v_bg, None, None, None, None, None, None, None, None)


class _AbsgradTarget:
    """Carries the caller's means2d tensor OBJECT through autograd untouched, so that backward can
    attach `.absgrad` to it (gsplat's contract, relied on at street_gaussian_model.py:505-506)."""
    __slots__ = ("tensor",)

    def __init__(self, tensor):
        self.tensor = tensor


def rasterize_to_pixels(means2d: Tensor, conics: Tensor, colors: Tensor, opacities: Tensor,
                        image_width: int, image_height: int, tile_size: int, isect_offsets: Tensor,
                        flatten_ids: Tensor, backgrounds: Optional[Tensor] = None,
                        masks: Optional[Tensor] = None, packed: bool = False,
                        absgrad: bool = False) -> Tuple[Tensor, Tensor]:
    """-> (render_colors [C,H,W,D], render_alphas [C,H,W,1])."""
    if packed:
        raise NotImplementedError("packed=True is not supported (reference passes packed=False)")
    caller_means2d = means2d
    means2d_c = _req(means2d, "means2d")
    conics = _req(conics, "conics")
    colors = _req(colors, "colors")
    opacities = _req(opacities, "opacities")
    isect_offsets = _req(isect_offsets, "isect_offsets", torch.int32)
    if type(flatten_ids) is _LazyTensor:      # isect_tiles' deferred list: the frame's counts are settled here (lazy.py)
        flatten_ids = flatten_ids.plain()
    flatten_ids = _req(flatten_ids, "flatten_ids", torch.int32)
    C, N = opacities.shape
    D = colors.shape[-1]
    assert means2d_c.shape == (C, N, 2), means2d_c.shape
    assert conics.shape == (C, N, 3), conics.shape
    assert colors.shape == (C, N, D), colors.shape
    assert isect_offsets.ndim == 3 and isect_offsets.shape[0] == C, isect_offsets.shape
    th, tw = isect_offsets.shape[1], isect_offsets.shape[2]
    assert tw * tile_size >= image_width, "image_width must fit in tile_width * tile_size"
    assert th * tile_size >= image_height, "image_height must fit in tile_height * tile_size"
    if not 1 <= D <= 32:
        raise NotImplementedError(f"colour channels must be in 1..32, got {D}")
    if backgrounds is not None:
        backgrounds = _req(backgrounds, "backgrounds")
        assert backgrounds.shape == (C, D), backgrounds.shape
    if masks is not None:
        assert masks.shape == isect_offsets.shape, masks.shape
        if not masks.is_cuda:
            raise RuntimeError("masks must live on a HIP device")
        masks = masks.to(torch.uint8).contiguous()

    nat = _native() if _needs_grad(means2d_c, conics, colors, opacities, backgrounds) else None
    if nat is not None:
        order, work = _sched_of(isect_offsets, C * tw * th)
        return nat.rasterize_autograd(means2d_c, conics, colors, opacities, backgrounds, masks, int(image_width),
                                      int(image_height), int(tile_size), isect_offsets, flatten_ids, bool(absgrad), order,
                                      work, caller_means2d, _RAW_STREAM)
    return _call(_Rasterize, means2d_c, conics, colors, opacities, backgrounds, masks, int(image_width),
                            int(image_height), int(tile_size), isect_offsets, flatten_ids, bool(absgrad),
                            _AbsgradTarget(caller_means2d))


# ------------------------------------------------------------------------------------------
# a13 rasterization  (imported at renderer.py:204, never called there): thin composition
# ------------------------------------------------------------------------------------------
def camera_centers(viewmats: Tensor) -> Tensor:
    """[C,3] camera positions -R^T t of rigid world-to-camera matrices [C,4,4] (what the reference keeps
    as Camera.camera_center, camera_utils.py:51, and gsplat takes from inverse(viewmats))."""
    lib = _lib.load()
    viewmats = _req(viewmats, "viewmats")
    out = torch.empty((viewmats.shape[0], 3), dtype=torch.float32, device=viewmats.device)
    _lib.check(lib.sc_camera_centers(_p(viewmats), viewmats.shape[0], _p(out), _stream(viewmats)),
               "sc_camera_centers")
    return out


_FUSED_RASTERIZATION = True


def set_fused_rasterization(enabled: bool) -> bool:
    """A/B switch for `rasterization()`'s fused forward (tests compare both); returns the old value."""
    global _FUSED_RASTERIZATION
    prev, _FUSED_RASTERIZATION = _FUSED_RASTERIZATION, bool(enabled)
    return prev


def _fused_forward_ok(tensors, sh_degree, render_mode, tile_size, colors) -> bool:
    if not _FUSED_RASTERIZATION or sh_degree is None or render_mode not in ("RGB+D", "RGB+ED"):
        return False
    if tile_size != 16 or colors.dim() != 3 or colors.shape[-1] != 3:
        return False
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        return False         # the fused path is forward-only; training goes through the autograd operators
    return True


class _FusedMeta(dict):
    """gsplat's meta dict.  The fused forward does not materialise `isect_ids` (8 B x I of keys nothing on
    this path reads); `meta["isect_ids"]` builds them on first access from the tensors already here.  Likewise
    `conics` / `opacities` / `colors` when the rasterizer gathered from packed records: first access unpacks them."""

    _FROM_RECORDS = ("conics", "opacities", "colors")

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def __contains__(self, key):
        return (dict.__contains__(self, key) or key == "isect_ids"
                or (key in self._FROM_RECORDS and getattr(self, "_sc_records", None) is not None))

    def __missing__(self, key):
        if key in self._FROM_RECORDS and getattr(self, "_sc_records", None) is not None:
            # the fused frame gathered these from the rasterizer's packed records and never wrote the arrays:
            # one kernel rebuilds all three on first access, on the current stream
            rec, (C, N), producer = self._sc_records          # (kept as an attribute: not one of gsplat's keys)
            dev = rec.device
            out = {"conics": torch.empty((C, N, 3), dtype=torch.float32, device=dev),
                   "opacities": torch.empty((C, N), dtype=torch.float32, device=dev),
                   "colors": torch.empty((C, N, 4), dtype=torch.float32, device=dev)}
            cur = torch.cuda.current_stream(dev)
            if cur != producer:
                cur.wait_stream(producer)
            _lib.check(_lib.load().sc_records_unpack(rec.data_ptr(), C * N, out["conics"].data_ptr(),
                                                     out["opacities"].data_ptr(), out["colors"].data_ptr(),
                                                     cur.cuda_stream), "sc_records_unpack")
            for k, v in out.items():
                self[k] = v
            return out[key]
        if key != "isect_ids":
            raise KeyError(key)
        with torch.no_grad():
            _, ids, _ = isect_tiles(self["means2d"], self["radii"], self["depths"], self["tile_size"],
                                    self["tile_width"], self["tile_height"], packed=False,
                                    n_cameras=self["n_cameras"])
        self[key] = ids
        return ids


@torch.no_grad()
def _rasterization_fused(means, quats, scales, opacities, colors, viewmats, Ks, width, height, near_plane,
                         far_plane, radius_clip, eps2d, sh_degree, tile_size, backgrounds, render_mode,
                         antialiased, centers):
    """a1 + a2 + a5 + a6 + a7 in one kernel, a3/a4 unchanged, a9 with the depth-normalising
    epilogue of renderer.py:284 -- three operator-level launches and no torch elementwise kernels."""
    lib = _lib.load()
    means, quats, scales = _req(means, "means"), _req(quats, "quats"), _req(scales, "scales")
    viewmats, Ks = _req(viewmats, "viewmats"), _req(Ks, "Ks")
    opacities = _req(opacities, "opacities").reshape(-1)
    colors = _req(colors, "colors")
    C, N, K = viewmats.shape[0], means.shape[0], colors.shape[1]
    assert opacities.shape[0] == N and colors.shape[0] == N, (opacities.shape, colors.shape)
    dev = means.device
    centers = camera_centers(viewmats) if centers is None else _req(centers, "camera_centers").reshape(C, 3)
    st = _stream(means)
    fast = _lib.fast()
    # the rasterizer's 48-B record per (camera, Gaussian): one gather line per splat instead of four.  With it the
    # conics / opacities / colours arrays of `meta` are not written at all (32 B per Gaussian the frame never reads):
    # _FusedMeta rebuilds them from the records on first access
    use_records = _SWITCH.packed_records and int(tile_size) == 16 and N > 0 and C > 0
    records = conics = opac = cols = None
    if fast is not None:
        rc, radii, means2d, depths, records, conics, opac, cols = fast.projection_sh_fwd(
            means, quats, scales, opacities, colors, viewmats, Ks, centers, int(sh_degree), int(width), int(height),
            float(eps2d), float(near_plane), float(far_plane), float(radius_clip), bool(antialiased), bool(use_records), st)
        if rc:
            _lib.check(rc, "sc_projection_sh_fwd")
    else:
        radii = torch.empty((C, N), dtype=torch.int32, device=dev)
        means2d = torch.empty((C, N, 2), dtype=torch.float32, device=dev)
        depths = torch.empty((C, N), dtype=torch.float32, device=dev)
        if use_records:
            records = torch.empty((C, N, 12), dtype=torch.float32, device=dev)
        else:
            conics = torch.empty((C, N, 3), dtype=torch.float32, device=dev)
            opac = torch.empty((C, N), dtype=torch.float32, device=dev)
            cols = torch.empty((C, N, 4), dtype=torch.float32, device=dev)
        _lib.check(lib.sc_projection_sh_fwd(_p(means), _p(quats), _p(scales), _p(opacities), _p(colors), _p(viewmats),
                                            _p(Ks), _p(centers), C, N, K, int(sh_degree), int(width), int(height),
                                            float(eps2d), float(near_plane), float(far_plane), float(radius_clip),
                                            int(antialiased), _p(radii), _p(means2d), _p(depths), _p(conics),
                                            _p(opac), _p(cols), _p(records), st), "sc_projection_sh_fwd")
    tile_width = math.ceil(width / float(tile_size))
    tile_height = math.ceil(height / float(tile_size))
    res = None
    if _ISECT_MODE["mode"] == "bin":
        res = _isect_tiles_bin(lib, means2d, radii, depths, C, N, tile_size, tile_width, tile_height,
                               None, None, st, want_ids=False, viewmats=viewmats)
    if res is not None:
        tiles_per_gauss, isect_ids, flatten_ids, isect_offsets = res
    else:
        tiles_per_gauss, isect_ids, flatten_ids = isect_tiles(means2d, radii, depths, tile_size, tile_width,
                                                              tile_height, packed=False, n_cameras=C)
        isect_offsets = isect_offset_encode(isect_ids, C, tile_width, tile_height)
    if backgrounds is not None:
        backgrounds = torch.cat([_req(backgrounds, "backgrounds"),
                                 torch.zeros(C, 1, device=dev)], dim=-1).contiguous()
    order, work = _sched_of(isect_offsets, C * tile_width * tile_height)
    sch = (_p(order), _p(work))
    render_colors = render_alphas = None
    if records is not None and fast is not None:
        rc, render_colors, render_alphas = fast.rasterize_fwd_packed(records, backgrounds, int(width), int(height),
                                                                     isect_offsets, flatten_ids, order, work,
                                                                     render_mode == "RGB+ED", st)
    else:
        render_colors = torch.empty((C, height, width, 4), dtype=torch.float32, device=dev)
        render_alphas = torch.empty((C, height, width, 1), dtype=torch.float32, device=dev)
    if records is not None:
        if fast is None:
            rc = lib.sc_rasterize_fwd_packed(_p(records), _p(backgrounds), None, C, N, int(width), int(height), tile_width,
                                             tile_height, _p(isect_offsets), _p(flatten_ids), flatten_ids.numel(),
                                             _p(render_colors), _p(render_alphas), *sch, int(render_mode == "RGB+ED"), st)
        if rc == -3:          # the reference-shaped raster kernel is selected: it reads the separate arrays
            conics = torch.empty((C, N, 3), dtype=torch.float32, device=dev)
            opac = torch.empty((C, N), dtype=torch.float32, device=dev)
            cols = torch.empty((C, N, 4), dtype=torch.float32, device=dev)
            _lib.check(lib.sc_records_unpack(_p(records), C * N, _p(conics), _p(opac), _p(cols), st), "sc_records_unpack")
            records = None
        else:
            _lib.check(rc, "sc_rasterize_fwd_packed")
    if records is None:
        args = (_p(means2d), _p(conics), _p(cols), _p(opac), _p(backgrounds), None, C, N, 4, int(width), int(height),
                int(tile_size), tile_width, tile_height, _p(isect_offsets), _p(flatten_ids), flatten_ids.numel(),
                _p(render_colors), _p(render_alphas))
        if render_mode == "RGB+ED":
            rc = lib.sc_rasterize_fwd_ed(*args, *sch, st)
            if rc == -3:          # the reference-shaped raster kernel is selected: plain launch + the torch post-step
                _lib.check(lib.sc_rasterize_fwd(*args, None, *sch, st), "sc_rasterize_fwd")
                render_colors = torch.cat([render_colors[..., :-1],
                                           render_colors[..., -1:] / render_alphas.clamp(min=1e-10)], dim=-1)
            else:
                _lib.check(rc, "sc_rasterize_fwd_ed")
        else:
            _lib.check(lib.sc_rasterize_fwd(*args, None, *sch, st), "sc_rasterize_fwd")
    meta = _FusedMeta({"radii": radii, "means2d": means2d, "depths": depths,
                       "tile_width": tile_width, "tile_height": tile_height, "tiles_per_gauss": tiles_per_gauss,
                       "flatten_ids": flatten_ids, "isect_offsets": isect_offsets,
                       "width": width, "height": height, "tile_size": tile_size, "n_cameras": C, "fused": True})
    if records is not None:          # conics / opacities / colors: rebuilt from the records on first access
        meta._sc_records = (records, (C, N), torch.cuda.current_stream(dev))
    else:
        meta.update({"conics": conics, "opacities": opac, "colors": cols})
    if isect_ids is not None:
        meta["isect_ids"] = isect_ids
    return render_colors, render_alphas, meta


def rasterization(means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, colors: Tensor,
                  viewmats: Tensor, Ks: Tensor, width: int, height: int, near_plane: float = 0.01,
                  far_plane: float = 1e10, radius_clip: float = 0.0, eps2d: float = 0.3,
                  sh_degree: Optional[int] = None, packed: bool = False, tile_size: int = 16,
                  backgrounds: Optional[Tensor] = None, render_mode: str = "RGB",
                  sparse_grad: bool = False, absgrad: bool = False, rasterize_mode: str = "classic",
                  channel_chunk: int = 32, distributed: bool = False, camera_model: str = "pinhole",
                  covars: Optional[Tensor] = None, camera_centers_: Optional[Tensor] = None):
    """gsplat's one-call API (imported at renderer.py:204).  means [N,3], quats [N,4], scales [N,3],
    opacities [N], colors [N,D] | [N,K,3] (with sh_degree), viewmats [C,4,4], Ks [C,3,3].  Returns
    (render_colors [C,H,W,*], render_alphas [C,H,W,1], meta).

    With `sh_degree` set, `render_mode` "RGB+D"/"RGB+ED" and no gradient required -- i.e. exactly what
    render_kernel_gsplat (renderer.py:186-302) spells out by hand: sh_degree=max_sh_degree,
    rasterize_mode="antialiased", render_mode="RGB+ED" -- the forward runs FUSED (see
    _rasterization_fused); results are bit-identical to the composition a1 -> a3 -> a4 -> a6 -> a9 below,
    which remains the path for training and for every other mode.
    `camera_centers_` (not in gsplat): optional precomputed [C,3] camera positions, e.g. the reference's
    Camera.camera_center; by default they are derived from `viewmats` (rigid inverse)."""
    assert render_mode in ("RGB", "D", "ED", "RGB+D", "RGB+ED"), render_mode
    assert rasterize_mode in ("classic", "antialiased"), rasterize_mode
    if camera_model != "pinhole" or distributed or covars is not None:
        raise NotImplementedError("only pinhole, single-process, quats/scales input is supported")
    if packed:
        raise NotImplementedError("packed=True is not supported (reference passes packed=False)")
    C = viewmats.shape[0]
    N = means.shape[0]
    aa = rasterize_mode == "antialiased"
    if _fused_forward_ok((means, quats, scales, opacities, colors, viewmats, Ks, backgrounds), sh_degree,
                         render_mode, tile_size, colors):
        return _rasterization_fused(means, quats, scales, opacities, colors, viewmats, Ks, width, height,
                                    near_plane, far_plane, radius_clip, eps2d, sh_degree, tile_size,
                                    backgrounds, render_mode, aa, camera_centers_)
    radii, means2d, depths, conics, comps = fully_fused_projection(
        means, None, quats, scales, viewmats, Ks, width, height, eps2d=eps2d, near_plane=near_plane,
        far_plane=far_plane, radius_clip=radius_clip, calc_compensations=aa)
    opac = opacities.reshape(1, N).expand(C, N)
    if comps is not None:
        opac = opac * comps
    tile_width = math.ceil(width / float(tile_size))
    tile_height = math.ceil(height / float(tile_size))
    tiles_per_gauss, isect_ids, flatten_ids = isect_tiles(means2d, radii, depths, tile_size, tile_width,
                                                          tile_height, packed=False, n_cameras=C)
    isect_offsets = isect_offset_encode(isect_ids, C, tile_width, tile_height)
    if sh_degree is None:
        cols = colors.reshape(1, N, colors.shape[-1]).expand(C, N, -1) if colors.dim() == 2 else colors
    else:
        campos = camera_centers(viewmats) if camera_centers_ is None else camera_centers_.reshape(C, 3)
        dirs = means[None, :, :] - campos[:, None, :]
        shs = colors.reshape(1, N, colors.shape[-2], 3).expand(C, N, -1, 3)
        cols = spherical_harmonics(sh_degree, dirs, shs, masks=radii > 0)
        cols = torch.clamp_min(cols + 0.5, 0.0)
    if render_mode in ("RGB+D", "RGB+ED"):
        cols = torch.cat([cols, depths[..., None]], dim=-1)
        if backgrounds is not None:
            backgrounds = torch.cat([backgrounds, torch.zeros(C, 1, device=backgrounds.device)], dim=-1)
    elif render_mode in ("D", "ED"):
        cols = depths[..., None]
        if backgrounds is not None:
            backgrounds = torch.zeros(C, 1, device=backgrounds.device)
    render_colors, render_alphas = rasterize_to_pixels(means2d, conics, cols.contiguous(), opac.contiguous(),
                                                       width, height, tile_size, isect_offsets, flatten_ids,
                                                       backgrounds=backgrounds, packed=False, absgrad=absgrad)
    if render_mode in ("ED", "RGB+ED"):
        render_colors = torch.cat([render_colors[..., :-1],
                                   render_colors[..., -1:] / render_alphas.clamp(min=1e-10)], dim=-1)
    meta = {"radii": radii, "means2d": means2d, "depths": depths, "conics": conics, "opacities": opac,
            "tile_width": tile_width, "tile_height": tile_height, "tiles_per_gauss": tiles_per_gauss,
            "isect_ids": isect_ids, "flatten_ids": flatten_ids, "isect_offsets": isect_offsets,
            "width": width, "height": height, "tile_size": tile_size, "n_cameras": C, "colors": cols,
            "fused": False}
    return render_colors, render_alphas, meta
