"""Is the frame loop host-bound?  Host time spent inside the per-frame call (no synchronisation by the caller)
against the wall time per frame, one and two frames in flight, plus a cProfile of the host side.
Usage: python tools/exp_host.py [frames] [profile]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from street_crafter_amd.dist import to_uint8_frame  # noqa: E402
from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_scene  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
if os.environ.get("SC_DEFER") == "0":      # A/B: isect_tiles waits for the frame's counts itself (round 2's form)
    from street_crafter_amd import rendering as _r
    _r.set_deferred_isect(False)
dev = "cuda"
W, H = 1920, 1280
cams = [bench.frame_camera(s, W, H).to(dev) for s in range(frames + 10)]
sc = make_scene(int(os.environ.get("SC_N", 1_000_000))).to(dev)
outs = [torch.empty(H, W, 3, dtype=torch.uint8, device=dev) for _ in range(4)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def frame(s):
    with torch.no_grad():
        to_uint8_frame(render_gaussians(sc, cams[s])["rgb"], out=outs[s % 4])


def loop(n_streams, prof=None):
    for s in range(10):
        frame(s)
    torch.cuda.synchronize()
    host = 0.0
    if prof:
        prof.enable()
    t0 = time.perf_counter()
    for s in range(10, 10 + frames):
        h0 = time.perf_counter()
        if n_streams == 1:
            frame(s)
        else:
            with torch.cuda.stream(streams[s % 2]):
                frame(s)
        host += time.perf_counter() - h0
    t_host_done = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if prof:
        prof.disable()
    print(f"{n_streams} in flight: wall {wall / frames * 1e3:.3f} ms/frame, host inside the call {host / frames * 1e3:.3f} ms/frame, "
          f"host loop finished {100 * t_host_done / wall:.0f} % into the run", flush=True)


def loop_threads(n_thr=2):
    """one host thread per frame in flight, each with its own HIP stream"""
    import threading
    for s in range(10):
        frame(s)
    torch.cuda.synchronize()
    errs = []

    def worker(w):
        try:
            with torch.cuda.stream(streams[w]):
                for s in range(10 + w, 10 + frames, n_thr):
                    frame(s)
        except BaseException as e:      # noqa: BLE001
            errs.append(e)

    t0 = time.perf_counter()
    ths = [threading.Thread(target=worker, args=(w,)) for w in range(n_thr)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    assert not errs, errs
    print(f"{n_thr} host threads, one stream each: wall {wall / frames * 1e3:.3f} ms/frame", flush=True)


loop(1)
loop(2)
loop(2)
loop_threads(2)
loop_threads(2)
streams.append(torch.cuda.Stream())
loop_threads(3)
if len(sys.argv) > 2:
    p = cProfile.Profile()
    loop(2, p)
    pstats.Stats(p).sort_stats("tottime").print_stats(22)

# ---- host time per operator call (two frames in flight, no profiler): wrap the operators with perf_counter
import gsplat.rendering as G  # noqa: E402
from street_crafter_amd import dist as D  # noqa: E402
acc = {}


def wrap(mod, name):
    fn = getattr(mod, name)

    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return r
    setattr(mod, name, w)


for n in ("fully_fused_projection", "isect_tiles", "isect_offset_encode", "spherical_harmonics", "rasterize_to_pixels"):
    wrap(G, n)
_tu = to_uint8_frame


def to_uint8_frame(*a, **k):      # noqa: F811
    t = time.perf_counter()
    r = _tu(*a, **k)
    acc["to_uint8_frame"] = acc.get("to_uint8_frame", 0.0) + time.perf_counter() - t
    return r


acc.clear()
loop(2)
tot = sum(acc.values())
print("host time per frame inside the operators (us):", {k: round(v / (frames + 10) * 1e6, 1) for k, v in acc.items()},
      "sum", round(tot / (frames + 10) * 1e6, 1))
