"""One-frame-per-GPU sharding of the novel-view loop with a gather of rendered frames.

The reference renders frames in a plain sequential loop (render.py:64-70: no cross-frame state,
model read-only under no_grad), single process, single GPU; it has no collective to mirror
(SURVEY.md 2.2, 8e).  Here: one process per GPU (torch.distributed, backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for tests), frame f is rendered by rank f % world, and finished
frames are gathered to rank 0 as uint8 [H,W,3] (7.4 MB at 1920x1280 -- tiny against 7 x 153 GB/s
of xGMI into the root), asynchronously, so the next frame's raster overlaps the transfer.
No data-path collective exists besides this gather: the shards are independent (weak scaling).
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch
import torch.distributed as dist


def frames_for_rank(n_frames: int, rank: int, world: int) -> List[int]:
    """frame f -> rank f % world (round-robin keeps neighbouring cameras on different GPUs)."""
    return list(range(rank, n_frames, world))


def to_uint8_frame(rgb_chw: torch.Tensor) -> torch.Tensor:
    """[3,H,W] float -> [H,W,3] uint8 = clamp(.,0,1)*255 rounded half up (what the reference's
    visualizer writes to disk).  On a HIP device this is ONE fused kernel (sc_frame_to_u8) when
    the tensor is the permuted view of an [H,W,C>=3] image, as the renderer returns it."""
    x = rgb_chw.detach()
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[0] == 3:
        hwc = x.permute(1, 2, 0)                      # [H,W,3] view
        H, W = hwc.shape[0], hwc.shape[1]
        if hwc.stride(2) == 1 and hwc.stride(1) >= 3 and hwc.stride(0) == W * hwc.stride(1):
            from . import _lib
            out = torch.empty((H, W, 3), dtype=torch.uint8, device=x.device)
            _lib.check(_lib.load().sc_frame_to_u8(hwc.data_ptr(), H * W, hwc.stride(1), out.data_ptr(),
                                                  torch.cuda.current_stream(x.device).cuda_stream),
                       "sc_frame_to_u8")
            return out
    return (x.clamp(0.0, 1.0) * 255.0 + 0.5).to(torch.uint8).permute(1, 2, 0).contiguous()


class FrameGatherer:
    """Gathers one frame per rank per round to `dst`.  `submit()` starts an async gather and
    returns immediately; `drain()` waits for everything outstanding and returns, on dst, the
    frames in global frame order."""

    def __init__(self, dst: int = 0, group=None):
        self.dst = dst
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._pending = []          # (work, round_index, gather_list | None, frame)
        self._done = {}             # global frame index -> tensor (dst only)

    def submit(self, round_index: int, frame_u8: torch.Tensor):
        if self.world == 1:
            self._done[round_index] = frame_u8
            return
        glist = None
        if self.rank == self.dst:
            glist = [torch.empty_like(frame_u8) for _ in range(self.world)]
        work = dist.gather(frame_u8, gather_list=glist, dst=self.dst, group=self.group, async_op=True)
        self._pending.append((work, round_index, glist, frame_u8))

    def drain(self):
        for work, r, glist, _ in self._pending:
            work.wait()
            if glist is not None:
                for k, f in enumerate(glist):
                    self._done[r * self.world + k] = f
        self._pending.clear()
        if self.rank != self.dst:
            return []
        return [self._done[k] for k in sorted(self._done)]


def render_sharded(n_frames: int, render_frame: Callable[[int], torch.Tensor], dst: int = 0,
                   group=None, frames_in_flight: int = 1) -> Optional[List[torch.Tensor]]:
    """Renders frames 0..n_frames-1 across the ranks of `group` and returns them in order on `dst`
    (None elsewhere).  n_frames must be a multiple of the world size (every round is a full
    gather); `render_frame(f)` returns the uint8 [H,W,3] frame f on this rank's device.

    frames_in_flight > 1 (HIP devices only): this rank's frames alternate over that many HIP streams, so
    the latency-bound intersection kernels of one frame run under the rasterizer of another (+20 % frames/s
    with 2 on S-1M, identical images; bench.py `two_frames_in_flight`).  Frames are independent, every
    operator launches on torch's current stream, and the gather of a frame is enqueued on its stream."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if n_frames % world:
        raise ValueError(f"n_frames={n_frames} must be a multiple of world size {world}")
    g = FrameGatherer(dst, group)
    streams = None
    if frames_in_flight > 1 and torch.cuda.is_available():
        streams = [torch.cuda.Stream() for _ in range(int(frames_in_flight))]
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
    for r, f in enumerate(frames_for_rank(n_frames, rank, world)):
        if streams is None:
            g.submit(r, render_frame(f))
        else:
            with torch.cuda.stream(streams[r % len(streams)]):
                g.submit(r, render_frame(f))
    if streams is not None:
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
    frames = g.drain()
    return frames if rank == dst else None
