"""One-frame-per-GPU sharding of the novel-view loop with a batched gather of rendered frames.

The reference renders frames in a plain sequential loop (render.py:64-70: no cross-frame state,
model read-only under no_grad), single process, single GPU; it has no collective to mirror
(SURVEY.md 2.2, 8e).  Here: one process per GPU (torch.distributed, backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for tests), frame f is rendered by rank f % world, and finished
frames travel to rank 0 as uint8 [H,W,3] (7.4 MB at 1920x1280), `batch` frames per collective
(one `gather` of uint8[K,H,W,3] instead of K: the root otherwise enqueues world-1 receives every
half millisecond), asynchronously, from a ring of staging buffers, so rendering never waits for
a transfer.  No data-path collective exists besides this gather: the shards are independent
(weak scaling).  `launch_ranks` starts the ranks of a single-node job from a parent process that
never touches the GPU.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import _lib

# uint8 quantisation of a [0,1] image, as the reference's visualizer does it:
#   "video"      (x * 255).astype(np.uint8): the frames render_novel_view collects for the video
#                (street_gaussian_visualizer.py:97, base_visualizer.py:37; render.py:48-49 turns
#                save_video on and save_image off for mode=novel_view) -- truncation
#   "save_image" torchvision.utils.save_image's PNGs (street_gaussian_visualizer.py:92): x * 255 + 0.5
ROUNDING = {"video": 0, "save_image": 1}


def frames_for_rank(n_frames: int, rank: int, world: int) -> List[int]:
    """frame f -> rank f % world (round-robin keeps neighbouring cameras on different GPUs)."""
    return list(range(rank, n_frames, world))


def _hwc_view(x: torch.Tensor):
    """[3,H,W] tensor that is a permuted view of an [H,W,C>=3] image, or three planes of a [C,H,W] one -> (x itself: its first
    element IS pixel (0,0) channel 0 of that image, pixel stride, channel stride) or None.  (Read off the strides: building the permuted view costs ~2 us of
    host time per frame.)"""
    if x.dim() != 3 or x.shape[0] != 3:
        return None
    s0, s1, s2 = x.stride()
    if s0 == 1 and s2 >= 3 and s1 == x.shape[2] * s2:
        return x, s2, 1
    if s2 == 1 and s1 == x.shape[2] and s0 >= x.shape[1] * x.shape[2]:
        return x, 1, s0            # planes (rendering.set_planar_output): pixel stride 1, channel stride s0
    return None


_STREAM_FN = []


def _raw_stream(t: torch.Tensor):
    if not _STREAM_FN:
        from .rendering import _stream
        _STREAM_FN.append(_stream)
    return _STREAM_FN[0](t)


def to_uint8_frame(rgb_chw: torch.Tensor, acc: Optional[torch.Tensor] = None,
                   sky_rgb_chw: Optional[torch.Tensor] = None, rounding: str = "video",
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[3,H,W] float -> [H,W,3] uint8, the tail of render_novel_view + the visualizer's conversion:
    `clamp(clamp(rgb) + clamp(sky) * (1 - acc), 0, 1)` (street_gaussian_renderer.py:151-163; sky / acc
    optional: both or neither), then `rounding` "video" (truncate, default: what mode=novel_view keeps)
    or "save_image" (+0.5).  On a HIP device this is ONE fused kernel (sc_frame_composite_u8) when the
    images are permuted views of the rasterizer's [H,W,C>=3] outputs, as the renderer returns them.
    `out`: optional preallocated uint8 [H,W,3] (e.g. a slot of a gather batch)."""
    if rounding not in ROUNDING:
        raise ValueError(f"rounding must be one of {sorted(ROUNDING)}, got {rounding!r}")
    if (acc is None) != (sky_rgb_chw is None):
        raise ValueError("acc and sky_rgb_chw go together (two-pass frame) or are both None")
    x = rgb_chw.detach() if rgb_chw.requires_grad else rgb_chw
    H, W = x.shape[1], x.shape[2]
    if out is None:
        out = torch.empty((H, W, 3), dtype=torch.uint8, device=x.device)
    else:
        assert out.shape == (H, W, 3) and out.dtype == torch.uint8 and out.is_contiguous() and out.device == x.device
    if x.is_cuda and x.dtype == torch.float32:
        fg = _hwc_view(x)
        sky = _hwc_view(sky_rgb_chw.detach()) if sky_rgb_chw is not None else None
        a = acc.detach().reshape(H, W) if acc is not None else None
        if fg is not None and (sky_rgb_chw is None or (sky is not None and a.is_contiguous()
                                                      and a.dtype == torch.float32)):
            fast = _lib.fast()
            if fg[2] != 1 or (sky is not None and sky[2] != 1):
                if fast is not None:
                    rc = fast.frame_composite_u8_strided(fg[0].data_ptr(), fg[1], fg[2], 0 if a is None else a.data_ptr(),
                                                         0 if sky is None else sky[0].data_ptr(), 1 if sky is None else sky[1],
                                                         1 if sky is None else sky[2], H * W, ROUNDING[rounding], out,
                                                         _raw_stream(x))
                else:
                    rc = _lib.load().sc_frame_composite_u8_strided(
                        fg[0].data_ptr(), fg[1], fg[2], None if a is None else a.data_ptr(),
                        None if sky is None else sky[0].data_ptr(), 1 if sky is None else sky[1],
                        1 if sky is None else sky[2], H * W, ROUNDING[rounding], out.data_ptr(), _raw_stream(x))
            elif fast is not None:
                rc = fast.frame_composite_u8(fg[0].data_ptr(), fg[1], 0 if a is None else a.data_ptr(),
                                             0 if sky is None else sky[0].data_ptr(), 0 if sky is None else sky[1],
                                             H * W, ROUNDING[rounding], out, _raw_stream(x))
            else:
                rc = _lib.load().sc_frame_composite_u8(
                    fg[0].data_ptr(), fg[1], None if a is None else a.data_ptr(),
                    None if sky is None else sky[0].data_ptr(), 0 if sky is None else sky[1], H * W,
                    ROUNDING[rounding], out.data_ptr(), _raw_stream(x))
            if rc:
                _lib.check(rc, "sc_frame_composite_u8")
            return out
    v = x.float().clamp(0.0, 1.0)
    if sky_rgb_chw is not None:
        v = (v + sky_rgb_chw.detach().float().clamp(0.0, 1.0) * (1.0 - acc.detach().float().reshape(1, H, W))).clamp(0.0, 1.0)
    v = v * 255.0
    if ROUNDING[rounding]:
        v = v + 0.5
    out.copy_(v.to(torch.uint8).permute(1, 2, 0))
    return out


class FrameGatherer:
    """Gathers rendered frames to `dst`, `batch` rounds per collective.

    Round r = one frame from every rank (global frame index r * world + rank).  `slot(r)` hands out the
    uint8 [H,W,3] view the renderer writes round r's frame into (no copy); `submit(r)` marks it written
    (ordered after everything enqueued on the CURRENT stream) and, when its batch is complete, starts ONE
    async `gather` of uint8 [batch,H,W,3] per rank.  Staging buffers form a ring of `ring` batches, so
    rendering only waits for a gather when it is `ring` batches behind.  `drain()` flushes a partial
    batch, waits for everything and returns, on dst, the frames in global frame order.

    Frames may be produced on several HIP streams (frames in flight): the gather of a batch waits for
    the recorded event of each of its frames.

    With ONE rank there is nothing to gather and the frames are handed out as plain tensors -- unless
    `force_collective=True`, which sends a world of one through the very same staging ring, stream events and
    async `dist.gather` calls as a world of eight (needs an initialised process group): the only way to execute
    the RCCL transport path on a one-GPU box (tests/test_gpu_parity.py::test_rccl_gather_ring_at_world_one,
    `bench.py --force-gather`).

    `reserve_rounds` = R > 0: the caller knows how many rounds it will render (a video of R frames per rank): the
    memory the delivered frames live in -- uint8 [R,H,W,3] at one rank, [R/batch, world, batch, H,W,3] on dst -- is
    allocated ONCE here instead of one 7.4 MB tensor per frame (one rank) or `world` 59 MB tensors per collective
    (dst) in the render loop: a first pass over fresh memory is a hipMalloc per allocation, ~0.1 ms each on the
    host.  Rounds beyond R fall back to per-frame allocation."""

    def __init__(self, frame_shape: Sequence[int], device, dst: int = 0, group=None, batch: int = 8,
                 ring: int = 3, keep: bool = True, force_collective: bool = False, reserve_rounds: int = 0):
        self.dst, self.group = dst, group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.batch, self.ring = max(1, int(batch)), max(2, int(ring))
        self.shape = tuple(int(v) for v in frame_shape)
        self.device = torch.device(device)
        self.keep = keep
        if force_collective and not dist.is_initialized():
            raise RuntimeError("FrameGatherer(force_collective=True) needs torch.distributed.init_process_group first")
        # world 1: nothing to gather, frames are handed out as fresh tensors and kept as they are
        self._local = self.world == 1 and not force_collective
        self._staging = [] if self._local else [
            torch.empty((self.batch, *self.shape), dtype=torch.uint8, device=self.device) for _ in range(self.ring)]
        self._reserved = max(0, int(reserve_rounds))
        self._block = None                 # one rank: the delivered frames' storage, uint8 [R, *shape]
        self._recv_block = None            # dst of a real gather: uint8 [ceil(R / batch), world, batch, *shape]
        if self._reserved and keep:
            if self._local:
                self._block = torch.empty((self._reserved, *self.shape), dtype=torch.uint8, device=self.device)
            elif self.rank == self.dst:
                nb = (self._reserved + self.batch - 1) // self.batch
                self._recv_block = torch.empty((nb, self.world, self.batch, *self.shape), dtype=torch.uint8,
                                               device=self.device)
        self._single: Dict[int, torch.Tensor] = {}
        self._busy: List[Optional[object]] = [None] * self.ring    # outstanding work per ring entry
        self._events: Dict[int, object] = {}                      # round -> event on its stream (HIP only)
        self._written: Dict[int, set] = {}                        # batch index -> slots written
        self._pending = []                                        # (work, batch index, n rounds, recv list)
        self._done: Dict[int, torch.Tensor] = {}                  # global frame -> tensor (dst only)
        self.stats = {"gathers": 0, "bytes_per_gather": 0, "host_s_in_gather_calls": 0.0,
                      "host_s_waiting_for_ring": 0.0}

    def slot(self, round_index: int) -> torch.Tensor:
        if self._local:
            t = self._single.get(int(round_index))
            if t is None:
                if self._block is not None and 0 <= int(round_index) < self._reserved:
                    t = self._block[int(round_index)]
                else:
                    t = torch.empty(self.shape, dtype=torch.uint8, device=self.device)
                self._single[int(round_index)] = t
            return t
        b, j = divmod(int(round_index), self.batch)
        e = b % self.ring
        w = self._busy[e]
        if w is not None:
            # ring wrapped: the gather that last read this buffer must be done before the CURRENT stream
            # writes into it (every slot of the batch waits: its frames may be on different streams).
            # nccl: orders the stream, does not block the host; gloo: blocks until the transfer is done.
            t0 = time.perf_counter()
            w.wait()
            self.stats["host_s_waiting_for_ring"] += time.perf_counter() - t0
        return self._staging[e][j]

    def submit(self, round_index: int, frame_u8: Optional[torch.Tensor] = None):
        """frame_u8: given only when the frame was NOT written into slot(round_index) (it is copied in)."""
        r = int(round_index)
        if self._local:
            t = self._single.pop(r, None) if frame_u8 is None else frame_u8
            self._single.pop(r, None)
            if t is None:
                raise RuntimeError(f"round {r}: no frame was written (call slot({r}) or pass the frame)")
            if self.keep:
                self._done[r] = t
            return
        if frame_u8 is not None:
            s = self.slot(r)
            if frame_u8.data_ptr() != s.data_ptr():
                s.copy_(frame_u8)
        b = r // self.batch
        if self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record()
            self._events[r] = ev
        self._written.setdefault(b, set()).add(r % self.batch)
        if len(self._written[b]) == self.batch:
            self._flush(b, self.batch)

    def _flush(self, b: int, n_rounds: int):
        e = b % self.ring
        if self._written.pop(b) != set(range(n_rounds)):
            raise RuntimeError("FrameGatherer: rounds must be submitted 0, 1, 2, ... without gaps "
                               "(a partial batch is the LEADING part of its batch)")
        send = self._staging[e][:n_rounds]
        t0 = time.perf_counter()
        if self.device.type == "cuda":
            cur = torch.cuda.current_stream(self.device)
            for j in range(n_rounds):
                ev = self._events.pop(b * self.batch + j, None)
                if ev is not None:
                    cur.wait_event(ev)
        recv = None
        if self.rank == self.dst:
            if self._recv_block is not None and b < self._recv_block.shape[0]:
                recv = [self._recv_block[b, k, :n_rounds] for k in range(self.world)]
            else:
                recv = [torch.empty_like(send) for _ in range(self.world)]
        work = dist.gather(send, gather_list=recv, dst=self.dst, group=self.group, async_op=True)
        self._busy[e] = work
        self._pending.append((work, b, n_rounds, recv))
        self.stats["gathers"] += 1
        self.stats["bytes_per_gather"] = send.numel() * (self.world - 1)
        self.stats["host_s_in_gather_calls"] += time.perf_counter() - t0

    def drain(self) -> List[torch.Tensor]:
        for b in sorted(self._written):
            self._flush(b, len(self._written[b]))
        for work, b, n_rounds, recv in self._pending:
            work.wait()
            if recv is not None and self.keep:
                for k, batch_k in enumerate(recv):
                    for j in range(n_rounds):
                        self._done[(b * self.batch + j) * self.world + k] = batch_k[j]
        self._pending.clear()
        self._busy = [None] * self.ring
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
        if self.rank != self.dst:
            return []
        return [self._done[k] for k in sorted(self._done)]

    def reset(self):
        """Forget the frames delivered so far; round numbering restarts at 0 (call after drain())."""
        assert not self._pending and not self._written, "reset() needs a drained gatherer"
        self._done.clear()
        self._single.clear()
        self._events.clear()
        for k in self.stats:
            self.stats[k] = 0 if isinstance(self.stats[k], int) else 0.0


def make_stream(device, cus: Optional[int] = None, priority: Optional[int] = None):
    """A HIP stream of `device` for the frame loop, wrapped for torch: confined to the first `cus` CUs of the chip's numbering
    (the driver deals the mask bits round-robin over the XCDs: `cus` / 8 CUs of every XCD; include/street_crafter_amd.h,
    sc_stream_create), or of the given `priority` (lower = more urgent; sc_stream_priority_range), else a plain
    non-blocking stream.  The returned torch.cuda.ExternalStream keeps the raw handle alive until `destroy_stream`."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    dev = torch.device(device)
    out = ctypes.c_void_p()
    with torch.cuda.device(dev):
        if cus is not None:
            total = torch.cuda.get_device_properties(dev).multi_processor_count
            cus = max(1, min(int(cus), total))
            words = (total + 31) // 32
            mask = (ctypes.c_uint32 * words)()
            for i in range(cus):
                mask[i // 32] |= 1 << (i % 32)
            _lib.check(lib.sc_stream_create(0, ctypes.cast(mask, ctypes.c_void_p), words, ctypes.byref(out)), "sc_stream_create")
        else:
            lo, hi = ctypes.c_int(), ctypes.c_int()
            _lib.check(lib.sc_stream_priority_range(ctypes.byref(lo), ctypes.byref(hi)), "sc_stream_priority_range")
            pr = 0 if priority is None else max(min(int(priority), lo.value), hi.value)
            _lib.check(lib.sc_stream_create(pr, None, 0, ctypes.byref(out)), "sc_stream_create")
    return torch.cuda.ExternalStream(out.value, device=dev)


def destroy_stream(stream) -> None:
    """Synchronises and destroys a stream made by make_stream."""
    from . import _lib
    stream.synchronize()
    _lib.check(_lib.load().sc_stream_destroy(stream.cuda_stream), "sc_stream_destroy")


def render_sharded(n_frames: int, render_frame: Callable[..., torch.Tensor], dst: int = 0,
                   group=None, frames_in_flight: int = 3, batch: int = 8, ring: int = 3,
                   force_collective: bool = False) -> Optional[List[torch.Tensor]]:
    """Renders frames 0..n_frames-1 across the ranks of `group` and returns them in order on `dst`
    (None elsewhere).  n_frames must be a multiple of the world size (every round is one frame per
    rank); `render_frame(f)` returns the uint8 [H,W,3] frame f on this rank's device (if it accepts a
    keyword `out`, it is handed the staging slot to write into and the copy is skipped).

    frames_in_flight > 1 (HIP devices only): this rank's frames alternate over that many HIP streams, so
    the latency-bound intersection kernels of one frame run under the rasterizer of another (+12..18 % frames/s
    with 2 on S-1M, +2..3 % more with 3, identical images).  Frames are independent, every operator launches on torch's current
    stream, and the gather of a batch waits for the events of its frames.
    force_collective: see FrameGatherer (a world of one goes through the real gather ring)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if n_frames % world:
        raise ValueError(f"n_frames={n_frames} must be a multiple of world size {world}")
    mine = frames_for_rank(n_frames, rank, world)
    if not mine:
        return [] if rank == dst else None
    import inspect
    takes_out = "out" in inspect.signature(render_frame).parameters
    g = None
    streams = None
    home = None                           # the caller's stream (restored at the end; set_stream per frame instead of the
    try:                                  # `with torch.cuda.stream(...)` context manager: ~10 us of host time per frame)
        for r, f in enumerate(mine):
            if g is not None and streams is not None:
                torch.cuda.set_stream(streams[r % len(streams)])
            if g is None:                 # the first frame tells the frame shape / device
                first = render_frame(f)
                g = FrameGatherer(first.shape, first.device, dst, group, batch=min(batch, len(mine)), ring=ring,
                                  force_collective=force_collective, reserve_rounds=len(mine))
                render_sharded.last_stats = g.stats          # (for tests / the bench line)
                if frames_in_flight > 1 and first.is_cuda:
                    home = torch.cuda.current_stream(first.device)
                    streams = [torch.cuda.Stream(device=first.device) for _ in range(int(frames_in_flight))]
                    for st in streams:
                        st.wait_stream(home)
                g.submit(r, first)
            elif takes_out:
                render_frame(f, out=g.slot(r))
                g.submit(r)
            else:
                g.submit(r, render_frame(f))
    finally:
        if home is not None:
            torch.cuda.set_stream(home)
    if streams is not None:
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
    frames = g.drain()
    return frames if rank == dst else None


# ---- per-rank host placement (one process per GPU; render.py:64-70's frames are sharded one per rank) -------------------
def _parse_cpulist(text: str) -> List[int]:
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_host_topology(sysfs: str = "/sys") -> List[Dict]:
    """The GPUs of this machine in HIP's default enumeration order (= the order of the KFD topology's GPU nodes) with the
    PCI address, NUMA node and local CPUs of each, read from sysfs WITHOUT touching the GPU (a rank calls this before its
    first HIP call): [{"index", "pci", "numa_node", "cpus"}].  Empty when the machine has no KFD topology (a CPU box)."""
    nodes_dir = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    out = []
    try:
        names = sorted(os.listdir(nodes_dir), key=lambda n: int(n))
    except (OSError, ValueError):
        return out
    for n in names:
        props = {}
        try:
            with open(os.path.join(nodes_dir, n, "properties")) as f:
                for line in f:
                    k, _, v = line.strip().partition(" ")
                    if v.strip().lstrip("-").isdigit():
                        props[k] = int(v)
        except OSError:
            continue
        if props.get("simd_count", 0) <= 0:          # a CPU node
            continue
        loc, dom = props.get("location_id", 0), props.get("domain", 0)
        pci = "%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 0x7)
        numa, cpus = -1, []
        try:
            with open(os.path.join(sysfs, "bus", "pci", "devices", pci, "numa_node")) as f:
                numa = int(f.read().strip())
            with open(os.path.join(sysfs, "bus", "pci", "devices", pci, "local_cpulist")) as f:
                cpus = _parse_cpulist(f.read())
        except (OSError, ValueError):
            pass
        out.append({"index": len(out), "pci": pci, "numa_node": numa, "cpus": cpus})
    return out


def cpus_for_local_rank(topology: List[Dict], local_rank: int, n_local: int, allowed: Optional[Sequence[int]] = None) -> List[int]:
    """The host CPUs rank `local_rank` of `n_local` on this node should run on: the CPUs local to ITS GPU's NUMA node, cut
    into equal contiguous shares among the local ranks whose GPUs sit on the same node (8 GPUs on 2 sockets: 4 ranks per
    socket, a quarter of the socket's CPUs each), intersected with what the process may use at all (`allowed`: a cgroup /
    taskset).  [] = no recommendation (no topology, GPU without a NUMA node, or nothing left after the intersection)."""
    if not topology or not (0 <= local_rank < len(topology)):
        return []
    me = topology[local_rank]
    cpus = sorted(me["cpus"])
    if allowed is not None:
        ok = set(allowed)
        cpus = [c for c in cpus if c in ok]
    if not cpus:
        return []
    same = [g["index"] for g in topology[:max(n_local, local_rank + 1)] if g["numa_node"] == me["numa_node"] and g["cpus"] == me["cpus"]]
    k, n = same.index(local_rank), len(same)
    per = len(cpus) // n
    if per == 0:
        return cpus
    return cpus[k * per:(k + 1) * per]


def bind_rank(local_rank: int, n_local: int, sysfs: str = "/sys", num_threads: int = 4) -> Dict:
    """Called by every rank of a multi-GPU job BEFORE its first GPU call: pins the process (the calling thread; the threads
    HIP / RCCL start later inherit it) to the CPUs of its GPU's NUMA node (cpus_for_local_rank) and caps torch's intra-op
    CPU threads (the render loop is one Python thread issuing ~25 launches per 0.36 ms frame: eight such hosts left to the
    scheduler land on one socket as easily as not).  SC_BIND_CPUS=0 switches it off.  All GPUs stay VISIBLE to every rank
    (torch.cuda.set_device(local_rank) selects one): RCCL's xGMI transport maps its peers' buffers through hipIpc /
    peer access, which needs the peer devices enumerated in the process; per-rank HIP_VISIBLE_DEVICES is the configuration
    RCCL falls back to host shared memory on.  Returns what was done (for the bench line)."""
    info = {"local_rank": int(local_rank), "bound": False, "cpus": None, "numa_node": None, "pci": None}
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return info
    info["cpus_before"] = len(allowed)
    if os.environ.get("SC_BIND_CPUS", "1") == "0":
        info["why"] = "SC_BIND_CPUS=0"
        return info
    topo = gpu_host_topology(sysfs)
    if topo and 0 <= local_rank < len(topo):
        info["numa_node"], info["pci"] = topo[local_rank]["numa_node"], topo[local_rank]["pci"]
    cpus = cpus_for_local_rank(topo, local_rank, n_local, allowed)
    if not cpus:
        info["why"] = "no KFD topology / NUMA information for this GPU, or no local CPU is allowed to this process"
        return info
    try:
        os.sched_setaffinity(0, cpus)
    except OSError as e:
        info["why"] = f"sched_setaffinity failed: {e}"
        return info
    torch.set_num_threads(max(1, min(int(num_threads), len(cpus))))
    info.update(bound=True, cpus=f"{cpus[0]}-{cpus[-1]} ({len(cpus)})" if cpus == list(range(cpus[0], cpus[-1] + 1))
                else ",".join(map(str, cpus)), torch_threads=torch.get_num_threads())
    return info


def verify_rank_binding(info: Dict, local_rank: int, n_local: int, pci_bus: Optional[int], sysfs: str = "/sys") -> Dict:
    """After the first GPU call: bind_rank chose this rank's CPUs by the GPU's position in the KFD topology, which is HIP's
    default device order; `pci_bus` (torch.cuda.get_device_properties(dev).pci_bus_id) says which GPU the rank really got.  If
    that is another entry of the topology (a reordered HIP_VISIBLE_DEVICES), every thread of the process is moved to that
    GPU's share of CPUs instead.  Returns the updated record."""
    info = dict(info or {})
    info["pci_bus_of_device"] = pci_bus
    if not info.get("bound") or pci_bus is None:
        return info
    topo = gpu_host_topology(sysfs)
    mine = [g for g in topo if int(g["pci"].split(":")[1], 16) == int(pci_bus)]
    if not mine or mine[0]["index"] == local_rank:
        info["binding_verified"] = bool(mine)
        return info
    try:
        # (no intersection with the current affinity: the first binding has already narrowed this thread to the wrong share)
        cpus = cpus_for_local_rank(topo, mine[0]["index"], max(n_local, mine[0]["index"] + 1), None)
        if cpus:
            for tid in os.listdir("/proc/self/task"):
                try:
                    os.sched_setaffinity(int(tid), cpus)
                except OSError:
                    pass
            info.update(cpus=f"{cpus[0]}-{cpus[-1]} ({len(cpus)})", numa_node=mine[0]["numa_node"], pci=mine[0]["pci"],
                        rebound_after_init=True)
    except (AttributeError, OSError) as e:
        info["rebind_error"] = str(e)
    return info


# ---- single-node launcher -----------------------------------------------------------------------
def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def launch_ranks(argv: Sequence[str], world: int, env: Optional[Dict[str, str]] = None,
                 timeout: Optional[float] = None) -> int:
    """Starts `world` copies of `argv` (one per rank: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT set, rendezvous on 127.0.0.1) as child processes, waits for them, and returns the worst
    exit code.  If a rank fails, the others are terminated (the exact PIDs started here).  The CALLER
    must not have initialised the GPU: children are started with subprocess (fork + exec of a fresh
    interpreter), never by exec-ing over a process that holds a HIP context."""
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base.setdefault("MASTER_PORT", str(free_port()))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base["WORLD_SIZE"] = str(int(world))
    procs = []
    for r in range(int(world)):
        e = dict(base)
        e["RANK"] = e["LOCAL_RANK"] = str(r)
        procs.append(subprocess.Popen(list(argv), env=e))
    deadline = None if timeout is None else time.monotonic() + timeout
    worst = 0
    alive = set(range(len(procs)))
    while alive:
        for i in sorted(alive):
            rc = procs[i].poll()
            if rc is None:
                continue
            alive.discard(i)
            if rc != 0:
                worst = worst or rc
                for j in alive:                # one rank failed: the job cannot finish
                    procs[j].terminate()
        if deadline is not None and time.monotonic() > deadline:
            for j in alive:
                procs[j].terminate()
            worst = worst or 124
            deadline = None
        time.sleep(0.05)
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    if worst < 0:
        worst = 128 - worst
    return worst


def visible_gpus() -> int:
    """Number of HIP devices WITHOUT creating a context (safe in a launcher parent)."""
    return int(torch.cuda.device_count())


if __name__ == "__main__":          # python -m street_crafter_amd.dist N prog args...
    sys.exit(launch_ranks(sys.argv[2:], int(sys.argv[1])))
