#!/usr/bin/env python3
"""Throughput / roofline benchmark of the MI355X Gaussian-splat hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = every rank renders ONE frame of the synthetic S-1M scene (1 000 000 Gaussians,
1920x1280, SURVEY.md 8d / BASELINE.md 2) through the drop-in gsplat operators (the caller's
sequence of street_gaussian_renderer.py:186-302, forward only) with a per-frame camera and turns it
into the uint8 frame the novel-view loop keeps; finished frames are gathered to rank 0 (RCCL), K
frames per collective.  Scene tensors are resident in HBM before the timed region.  Every rank keeps `--frames-in-flight`
(default 3) independent frames in flight, frame f on HIP stream f % 3: the latency-bound
intersection kernels of one frame run under the VALU-bound rasterizer of another.  Every 8th timed
frame (every 4th when fewer than 40 steps are timed) is a PROBE frame with HIP events around every operator, which is where `roofline` and
`stage_ms` come from; with several frames in flight a probed kernel shares the GPU with the other
streams' frames and its time says so (rocprofv3 of the same command sees the same).  The N = 1 run
therefore also carries `single_stream`: the same K frames one at a time, with the per-kernel times
and the roofline of kernels running ALONE.  Rank 0 prints ONE JSON line.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process (which never touches the
GPU) starts the N ranks itself and forwards rank 0's line.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12   # B/s, MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
METRIC = "frames/sec @1M Gaussians 1920x1280 + PSNR vs ref; 1/2/4/8 MI355X"
# kernel launched by each operator of the caller's sequence (rocprofv3 names, profiles/)
OPERATOR_KERNEL = {
    "projection": "projection_fwd_kernel",
    "isect_tiles": "bin_count + center_scatter + bin_scatter_flat + super_sort (4 kernels)",
    "isect_offset_encode": "(none: cached bucket scan)",
    "spherical_harmonics": "sh_fwd_kernel",
    "rasterize_to_pixels": "raster_fwd_wave_kernel<4, false, false, false, true>",
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n-gauss", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1280)
    ap.add_argument("--sh-degree", type=int, default=1)
    ap.add_argument("--frames-in-flight", type=int, default=None,
                    help="independent frames each rank keeps in flight, one HIP stream each (default 3 at every "
                         "--gpus N; 1 = one frame at a time: per-kernel HIP-event times then describe a kernel "
                         "running alone)")
    ap.add_argument("--gather-batch", type=int, default=8, help="frames per gather collective")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager-isect-wait", action="store_true",
                    help="A/B: isect_tiles waits for the frame's intersection count itself (round 2's form) instead of "
                         "deferring the wait to the first observation of its outputs (rendering.set_deferred_isect)")
    ap.add_argument("--no-reserve", action="store_true",
                    help="A/B: allocate every delivered frame / every collective's receive buffers inside the render loop "
                         "instead of one block before it")
    ap.add_argument("--isect-mode", choices=["bin", "radix"], default=None)
    ap.add_argument("--raster-variant", type=int, default=None)
    ap.add_argument("--stage-times", action="store_true", help="print per-operator times to stderr")
    ap.add_argument("--interleaved-output", action="store_true",
                    help="A/B: rasterize_to_pixels stores render_colors interleaved [C,H,W,D] as gsplat does instead of one plane "
                         "per channel behind the same indexing (rendering.set_planar_output; profiles/r04_planar_output_ab.txt)")
    ap.add_argument("--raster-cus", type=int, default=None,
                    help="A/B: every frame stream hands its rasterizer to a side stream confined to this many CUs "
                         "(dist.make_stream / rendering.set_raster_side_stream; measured slower: profiles/r04_cu_mask_ab.txt)")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip every secondary measurement (keeps rocprof profiles of the headline clean)")
    ap.add_argument("--skip", default="", help="comma list of secondary lines to skip: single_stream,two_in_flight,fused,"
                                               "two_pass,street,train,variants,knn,cpu_python,forced_gather")
    ap.add_argument("--force-gather", action="store_true",
                    help="N = 1 only: initialise RCCL (init_process_group('nccl', world_size=1)) and send the headline's "
                         "frames through the REAL gather ring (FrameGatherer(force_collective=True)) instead of the "
                         "world == 1 short cut")
    ap.add_argument("--cpu-python-frames", type=int, default=1,
                    help="timed frames of the pure-numpy oracle at S-100k for `cpu_baseline_python` (27 s each; "
                         "BASELINE.md section 2's protocol is 1 warm-up + 3: pass 3 and a warm-up frame is added)")
    ap.add_argument("--scene-ply", default=None,
                    help="render a scene file in the reference's point_cloud.ply layout instead of S-<n>")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="debug: every rank uses GPU 0 (exercises the RCCL gather path on a 1-GPU box if RCCL accepts "
                         "two ranks on one device; numbers are meaningless)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the frame gather (nccl = RCCL; gloo only for debugging)")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="launcher / sharding / gather / JSON plumbing on CPU tensors over gloo with a stand-in "
                         "frame source (no renderer, no GPU): what tests/test_dist_cpu.py drives")
    return ap.parse_args(argv)


def frame_camera(f, width, height):
    from street_crafter_amd.scenes import make_camera
    # small per-frame jitter (a lane-shift style novel view): yaw +-0.01 rad, x shift
    j = ((f * 37) % 11 - 5) / 5.0
    return make_camera(width, height, 2050.0 * width / 1920.0, 2050.0 * width / 1920.0,
                       yaw=0.01 * j, shift=(0.05 * j, 0.0, 0.0))


def stage_algorithmic_bytes(stage, N, I, P, T, K):
    """SURVEY.md 8(d) split by operator (read every input once, write every output once)."""
    return {
        "projection": 72 * N,
        "isect_tiles": 52 * N + (12 + 24) * I,
        "isect_offset_encode": 8 * I + 4 * T,
        "spherical_harmonics": (25 + 12 * K) * N,
        "rasterize_to_pixels": 44 * I + 24 * P,
    }[stage]


def percentiles(values):
    v = sorted(values)
    if not v:
        return None
    pick = lambda q: v[min(len(v) - 1, int(round(q * (len(v) - 1))))]
    return {"p10": pick(0.1), "p50": pick(0.5), "p90": pick(0.9), "samples": len(v)}


def host_threads(cap=32):
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, cap))


def cpu_baseline(scene, cam, W, H, hip_frame_fn, budget_s=30.0):
    """The C/OpenMP restatement of the oracle (oracle/gsplat_oracle_c.c, kind "port") on frame 0 of the SAME
    workload (all N Gaussians, full resolution): 1 warm-up + up to 3 timed frames within `budget_s`, on the
    host threads stated in `cores`.  The warm-up frame doubles as the parity reference for the HIP frame."""
    import numpy as np
    from oracle import gsplat_oracle as O          # checker / baseline only
    from oracle import gsplat_oracle_c as OC
    cores = host_threads()
    OC.set_num_threads(cores)
    args = (scene.means.cpu().numpy(), scene.quats.cpu().numpy(), scene.scales.cpu().numpy(),
            scene.opacities.cpu().numpy(), scene.sh.cpu().numpy(), cam.viewmat.cpu().numpy(), cam.K.cpu().numpy(),
            W, H, scene.sh_degree)
    kw = dict(near_plane=cam.znear, far_plane=cam.zfar)
    t0 = time.perf_counter()
    exp = OC.render_frame(*args, return_unstable=True, **kw)
    t_warm = time.perf_counter() - t0
    n_timed = int(max(1, min(3, (budget_s - t_warm) // max(t_warm, 1e-3))))
    times = []
    for _ in range(n_timed):
        t0 = time.perf_counter()
        OC.render_frame(*args, **kw)
        times.append(time.perf_counter() - t0)
    got = hip_frame_fn()
    ref_rgb = np.clip(exp["render_colors"][0, ..., :3], 0.0, 1.0)
    err = np.abs(got["rgb"] - ref_rgb).max(axis=-1)
    # EVERY pixel beyond the flat 1e-4 bar (threshold-unstable or not) gets a verdict from the float64 blend of the same
    # fp32 inputs (oracle/blend_f64.py: a pixel is right iff it is within 1e-4 of one of the float64 outcomes, or at least
    # as close to it as the fp32 oracle's own pixel): "all pixels" has a judge, not an exclusion window
    from oracle import blend_f64 as B64
    ys, xs = np.nonzero(err > 1e-4)
    order = np.argsort(-err[ys, xs])[:256]
    f64 = {"f64_judged": 0, "f64_failed": 0, "f64_not_judged": int(max(0, len(ys) - 256)), "worst_hip_vs_f64": 0.0,
           "worst_oracle_fp32_vs_f64": 0.0}
    inputs_same = bool(np.array_equal(got["colors"].view(np.uint32), exp["colors"].view(np.uint32)) and
                       np.array_equal(got["opacities"].view(np.uint32), exp["opacities"].view(np.uint32)) and
                       np.array_equal(got["means2d"].view(np.uint32), exp["means2d"].view(np.uint32)) and
                       np.array_equal(got["conics"].view(np.uint32), exp["conics"].view(np.uint32)))
    if len(order):
        scale = np.maximum(1.0, np.abs(exp["render_colors"][0]).max(axis=(0, 1)))      # (depth channel: metres)
        rows = B64.judge_pixels(list(zip(xs[order], ys[order])), W, 16, exp["isect_offsets"], exp["flatten_ids"],
                                exp["means2d"], exp["conics"], exp["colors"], exp["opacities"],
                                {"hip": (got["render_colors"], got["render_alphas"]),
                                 "oracle32": (exp["render_colors"][0], exp["render_alphas"][0])}, scale=scale)
        f64.update(f64_judged=len(rows),
                   f64_failed=sum(not (r["hip"]["err"] <= max(1e-4, r["oracle32"]["err"])) for r in rows),
                   worst_hip_vs_f64=max(r["hip"]["err"] for r in rows),
                   worst_oracle_fp32_vs_f64=max(r["oracle32"]["err"] for r in rows))
    # pixels where some alpha / transmittance sits within 2e-5 (relative) of a hard threshold can legitimately
    # flip on a 1-ulp exp difference (the oracle flags them); all-pixel figures are reported beside the stable ones
    stable = ~exp["unstable"][0]
    parity = {
        "psnr_db_vs_oracle": O.psnr(got["rgb"], ref_rgb),
        "max_abs_rgb_stable_pixels": float(err[stable].max()),
        "max_abs_rgb_all_pixels": float(err.max()),
        "threshold_unstable_pixels": int((~stable).sum()),
        "pixels_over_1e-4": int((err > 1e-4).sum()),
        "pixels_over_1e-4_among_stable": int((err[stable] > 1e-4).sum()),
        "n_pixels": int(err.size),
        **f64,
        "f64_what": "every pixel over 1e-4 vs the fp32 oracle (stable or not, the 256 worst at most) against oracle/blend_f64.py: "
                    "float64 blend of the SAME fp32 rasterizer inputs (bit-identical on both sides: "
                    f"{inputs_same}), all outcomes of decisions no fp32 evaluation can resolve enumerated; failed = farther than "
                    "1e-4 from every outcome AND farther than the fp32 oracle's own pixel",
        "isect_ids_and_flatten_ids_bit_exact": bool(np.array_equal(got["isect_ids"], exp["isect_ids"]) and
                                                    np.array_equal(got["flatten_ids"], exp["flatten_ids"])),
        "radii_bit_exact": bool(np.array_equal(got["radii"], exp["radii"])),
        "oracle": "oracle/gsplat_oracle_c.c (pinned bit-for-bit to oracle/gsplat_oracle.py on ints and "
                  "projection / SH floats by tests/test_oracle_cpu.py)",
    }
    med = sorted(times)[len(times) // 2]
    base = {"value": 1.0 / med, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle/gsplat_oracle_c.c (C + OpenMP, {cores} threads) on frame 0 of the SAME workload "
                      f"({scene.n} Gaussians, {W}x{H}, I={int(exp['isect_ids'].shape[0])}): 1 warm-up "
                      f"({t_warm:.2f} s) + {n_timed} timed frames, value = 1 / median ({med:.2f} s)",
            "frame_seconds": times, "host_cpus": os.cpu_count()}
    return base, parity


def cpu_baseline_python(W, H, frames=1, n=100_000):
    """BASELINE.md section 2's stated baseline: the pure-Python (numpy) oracle, oracle/gsplat_oracle.py, on S-100k at
    full resolution on the GPU box's host (numpy's elementwise kernels run on ONE core).  `frames` timed frames
    (+ 1 warm-up frame when frames >= 3: the BASELINE.md protocol; the default single frame keeps the bench run
    short: a frame takes ~27 s)."""
    from oracle import gsplat_oracle as O          # baseline only
    from street_crafter_amd.scenes import make_camera, make_scene
    sc = make_scene(n)
    cam = make_camera(W, H, 2050.0 * W / 1920.0, 2050.0 * W / 1920.0)
    a = (sc.means.numpy(), sc.quats.numpy(), sc.scales.numpy(), sc.opacities.numpy(), sc.sh.numpy(),
         cam.viewmat.numpy(), cam.K.numpy(), W, H, sc.sh_degree)
    kw = dict(near_plane=cam.znear, far_plane=cam.zfar)
    warm = 1 if frames >= 3 else 0
    times, n_is = [], 0
    for k in range(warm + frames):
        t0 = time.perf_counter()
        r = O.render_frame(*a, **kw)
        dt = time.perf_counter() - t0
        n_is = int(r["flatten_ids"].size)
        if k >= warm:
            times.append(dt)
    med = sorted(times)[len(times) // 2]
    return {"value": 1.0 / med, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"oracle/gsplat_oracle.py (pure numpy, vectorised per tile; numpy's elementwise loops use one "
                      f"core) on S-{n // 1000}k {W}x{H} (I={n_is}): {warm} warm-up + {frames} timed frame(s), value = "
                      f"1 / median ({med:.1f} s)",
            "frame_seconds": times, "host_cpus": os.cpu_count(),
            "what": "the pure-Python/CPU rasterizer BASELINE.md section 2 and north_star name; `cpu_baseline` beside it "
                    "is the stronger C + OpenMP port on the headline's own S-1M frame"}


def pmc_traffic_entry(traffic_json, op, key, kernel_symbol):
    """HBM bytes per launch of `op` from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json), or
    (None, reason) when the file has no such entry or was recorded for ANOTHER kernel symbol (it goes stale silently
    when a kernel changes: the symbol the counters were collected on is stored beside them and must match)."""
    val = traffic_json.get(op, {}).get(key)
    if val is None:
        return None, f"profiles/pmc_traffic.json has no entry {op}/{key}"
    recorded = traffic_json.get("_kernels", {}).get(key, {}).get(op)
    if not recorded:
        return None, "profiles/pmc_traffic.json does not say which kernel symbol its counters were collected on (stale file)"
    want = kernel_symbol.replace(" ", "")
    if not any(want in r.replace(" ", "") for r in recorded):
        return None, (f"profiles/pmc_traffic.json was recorded on {recorded}, this build launches {kernel_symbol}: "
                      "re-run tools/pmc_run.sh")
    return val, ("profiles/pmc_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE of this kernel symbol from separate rocprofv3 "
                 "--pmc passes (committed; not re-measured in this run)")


KNN_BYTES_PER_POINT = 104
KNN_BYTES_WHAT = ("12 (bbox pass reads the points) + 24 (Morton pass: 12 in, 8-B key + 4-B id out) + 24 (sort counted as "
                  "ONE read + one write of the 12-B pairs, as SURVEY 8d counts the tile sort) + 28 (gather into curve "
                  "order: id 4 + point 12 in, point 12 out) + 16 (search: own point 12 in, result 4 out; box AABBs are "
                  "24 B per 1024 points) = 104 B per point")


def knn_line(dev, n=1_000_000, reps=7):
    """a14 (gaussian_model.py:65-66): simple_knn._C.distCUDA2 on n uniform points and on n street-LiDAR-shaped points
    (the foreground of scenes.make_street_scene: ground plane, facades, clutter), HIP events around the call on
    the current stream, median of `reps` after 2 warm-ups."""
    import torch
    from simple_knn._C import distCUDA2
    from street_crafter_amd.scenes import make_street_scene
    g = torch.Generator().manual_seed(20250404)
    clouds = {"uniform": (torch.rand(n, 3, generator=g) * 100.0).to(dev),
              "street_lidar_shaped": make_street_scene(n)[0].means.to(dev).contiguous()}
    out = {"n_points": n, "algorithmic_bytes": KNN_BYTES_PER_POINT * n, "algorithmic_bytes_is": KNN_BYTES_WHAT,
           "kernels": "knn_box_minmax x2, knn_reduce_boxes, knn_morton, radix sort (4 passes over the 30-bit code), "
                      "knn_mean_dist", "call_site": "street_gaussian/models/gaussian_model.py:65-66"}
    for name, pts in clouds.items():
        for _ in range(2):
            distCUDA2(pts)
        ms = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            distCUDA2(pts)
            b.record()
            b.synchronize()
            ms.append(a.elapsed_time(b))
        med = sorted(ms)[len(ms) // 2]
        out[name] = {"ms": med, "points_per_s": n / (med * 1e-3), "gb_per_s": KNN_BYTES_PER_POINT * n / (med * 1e-3) / 1e9,
                     "frac_of_hbm_peak": KNN_BYTES_PER_POINT * n / (med * 1e-3) / HBM_PEAK}
    out["what"] = ("init-time operator (once per sub-model); bound by the pruned candidate scan (distance evaluations "
                   "per point), not by bandwidth: the clustered cloud walks more boxes per point")
    return out


def raster_bwd_algorithmic_bytes(I, P):
    """Backward of rasterize_to_pixels at the operator boundary, by analogy with SURVEY 8(d)'s forward figure: the
    replayed gather (44 B per intersection: id 4 + xy 8 + opacity 4 + conic 12 + colour 16), ONE set of gradient
    atomics per (Gaussian, tile) (v_means2d 8 + absgrad 8 + v_conics 12 + v_colors 16 + v_opacities 4 = 48 B), and
    28 B per pixel in (v_render_colors 16, v_render_alphas 4, render_alphas 4, last_ids 4)."""
    return (44 + 48) * I + 28 * P


# ---------------------------------------------------------------------------------------------
def launch_self(args) -> int:
    """--gpus N > 1 without a launcher: start the N ranks from here.  This parent never initialises the
    GPU (device_count() does not create a context on this image) and never exec()s: children are fresh
    interpreters; rank 0's JSON line goes straight to the inherited stdout."""
    from street_crafter_amd.dist import launch_ranks, visible_gpus
    if not args.selftest_cpu and not args.oversubscribe:
        have = visible_gpus()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} requested but this machine exposes {have} GPU(s); "
                  f"run with --gpus {max(have, 1)} (the 8-GPU curve comes from the driver's 8-GPU node)",
                  file=sys.stderr)
            return 2
    return launch_ranks([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], args.gpus)


def main():
    args = parse()
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        sys.exit(2)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_self(args))
    run_rank(args)


def _rccl_version(torch):
    """RCCL's version as torch reports it (never worth failing a multi-GPU run over)."""
    try:
        v = torch.cuda.nccl.version()
        return ".".join(map(str, v)) if isinstance(v, (tuple, list)) else str(v)
    except Exception as e:      # noqa: BLE001
        return f"unavailable ({type(e).__name__})"


def init_world1(dev):
    """init_process_group("nccl", world_size=1): the same call a rank of an N > 1 job makes, on the one GPU here."""
    import torch.distributed as dist
    from street_crafter_amd.dist import free_port
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(free_port()))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "1"
    if not dist.is_initialized():
        dist.init_process_group("nccl", device_id=dev)


_REAL_STDOUT = None


def emit(line: dict):
    """The ONE JSON line of rank 0, written to the process's original stdout."""
    data = (json.dumps(line) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def run_rank(args):
    import torch
    import torch.distributed as dist

    # RCCL prints a version banner ("RCCL version : ...", five lines) to STDOUT when its first communicator is
    # created (seen on the one-GPU box as soon as the forced gather initialised it).  Rank 0's stdout must carry ONE
    # JSON line: from here on file descriptor 1 points at stderr, and the line is written to the saved descriptor.
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
              f"(or without a launcher: bench.py starts its own ranks)", file=sys.stderr)
        sys.exit(2)
    selftest = args.selftest_cpu
    placement = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("NCCL_DEBUG", "VERSION")      # (RCCL states its version once, on stderr; the line carries it too)
        if not selftest and not args.oversubscribe:
            # before the first GPU call: this rank's host threads onto the CPUs of its GPU's NUMA node (dist.bind_rank)
            try:
                from street_crafter_amd.dist import bind_rank
                placement = bind_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
            except Exception as e:      # noqa: BLE001  (placement is an optimisation: never a reason to lose the run)
                placement = {"bound": False, "why": f"{type(e).__name__}: {e}"}
    if selftest:
        dev = torch.device("cpu")
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        if not torch.cuda.is_available():
            print("bench.py needs a GPU (HIP); there is no CPU path", file=sys.stderr)
            sys.exit(2)
        if args.oversubscribe:
            local_rank = 0
        if local_rank >= torch.cuda.device_count():
            print(f"bench.py: rank {rank} has no GPU (LOCAL_RANK={local_rank}, "
                  f"{torch.cuda.device_count()} visible)", file=sys.stderr)
            sys.exit(2)
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if placement is not None:          # which GPU did LOCAL_RANK really select?  (dist.verify_rank_binding)
            try:
                from street_crafter_amd.dist import verify_rank_binding
                placement = verify_rank_binding(placement, local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)),
                                                getattr(torch.cuda.get_device_properties(dev), "pci_bus_id", None))
            except Exception as e:      # noqa: BLE001
                placement = dict(placement, verify_error=f"{type(e).__name__}: {e}")
        if world > 1:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
        elif args.force_gather:
            init_world1(dev)

    from street_crafter_amd.dist import FrameGatherer, to_uint8_frame
    W, H = args.width, args.height
    total_steps = args.warmup + args.steps
    if args.frames_in_flight is None:
        # three since the second session of round 3 (2 -> 3: +2.2..3.4 % in four paired runs, 4: -3..-15 %;
        # profiles/r03_frames_in_flight.txt): with the host wait of isect_tiles deferred the host runs a frame ahead
        args.frames_in_flight = 3
    n_streams = 1 if selftest else max(1, args.frames_in_flight)
    skip = set(x for x in args.skip.split(",") if x)
    if args.headline_only:
        skip |= {"single_stream", "two_in_flight", "fused", "two_pass", "street", "train", "variants", "knn", "cpu_python",
                 "forced_gather"}

    if selftest:
        W, H = 64, 48
        scene = cams = None
        K = 4

        def render_into(s, out, events=None, intermediates=False):
            out.fill_((7 * (rank + s * world)) % 251)       # stand-in frame: a function of the GLOBAL frame index
            return None
    else:
        from street_crafter_amd import _lib, rendering
        from harness.caller import algorithmic_bytes, render_gaussians
        from street_crafter_amd.scenes import make_scene
        _lib.load()
        if args.isect_mode:
            rendering.set_isect_mode(args.isect_mode)
        if args.eager_isect_wait:
            rendering.set_deferred_isect(False)
        if args.raster_variant is not None:
            _lib.set_option("raster_fwd", args.raster_variant)
        if args.interleaved_output:
            rendering.set_planar_output(False)
        # (the forward's kernel symbol carries the output layout: template arguments CDIM, TRACK, PACKED, ED, PLANAR)
        OPERATOR_KERNEL["rasterize_to_pixels"] = ("raster_fwd_wave_kernel<4, false, false, false, true>"
                                                  if rendering._SWITCH.planar_out else
                                                  "raster_fwd_wave_kernel<4, false, false, false, false>")
        if args.scene_ply:
            # a scene in the reference's point_cloud.ply layout (street_crafter_amd/scene_io.py); actors, if
            # any, are placed with identity poses.  Not the headline workload: the metric string stays S-1M's.
            from street_crafter_amd import scene_io
            models = scene_io.read_ply(args.scene_ply)
            ident = (torch.tensor([1.0, 0.0, 0.0, 0.0]), torch.zeros(3))
            composed = scene_io.compose_scene(models, {n: ident for n in models if n not in ("background", "sky")})
            scene = composed.scene.to(dev)
            args.n_gauss, args.sh_degree = scene.n, scene.sh_degree
        else:
            scene = make_scene(args.n_gauss, sh_degree=args.sh_degree).to(dev)     # resident before timing
        K = (args.sh_degree + 1) ** 2
        cams = [frame_camera(rank + s * world, W, H).to(dev) for s in range(total_steps)]

        def render_into(s, out, events=None, intermediates=False):
            with torch.no_grad():
                o = render_gaussians(scene, cams[s], stage_events=events, return_intermediates=intermediates)
                to_uint8_frame(o["rgb"], out=out)
            return o

    # the delivered frames' memory (the "video") is allocated before the timed region, like the scene: one block for
    # the K rounds of a run (--no-reserve: one allocation per frame / per collective inside the loop, the earlier form)
    reserve = 0 if args.no_reserve else max(args.steps, args.warmup)
    gatherer = FrameGatherer((H, W, 3), dev, dst=0, batch=args.gather_batch, reserve_rounds=reserve,
                             force_collective=bool(args.force_gather and world == 1 and not selftest))
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)] if (not selftest and n_streams > 1) else None
    side_streams = []
    if args.raster_cus and not selftest:
        from street_crafter_amd.dist import make_stream
        for m in (streams or [torch.cuda.current_stream(dev)]):
            side_streams.append(make_stream(dev, cus=args.raster_cus))
            rendering.set_raster_side_stream(dev, side_streams[-1], main=m)

    def recorder():
        return {"events": {}, "n_isects": [], "frame_ev": [], "probe_ev": []}

    main_rec = recorder()

    def sync_streams():
        if not selftest:
            torch.cuda.synchronize(dev)

    probe_every = 8 if args.steps >= 40 else 4       # (the driver's --steps 20: five probe frames instead of three)

    def run_steps(first, last, timed, g, step_fn, n_str=n_streams, rec=None):
        """steps [first, last): frame s on stream s % n_str; with a recorder, every 8th timed step is a probe frame
        with per-operator HIP events (no draining: with n_str > 1 a probed kernel shares the GPU with the other
        stream's frame and its time says so) and every timed frame gets an event pair."""
        strs = streams[:n_str] if (streams is not None and n_str > 1) else None
        # (the frame's stream is made current with set_stream, once per frame, and the caller's stream is restored at the
        #  end of the run: the `with torch.cuda.stream(...)` context manager costs ~10 us of host time per frame, 5 % of
        #  an S-100k frame)
        home = torch.cuda.current_stream(dev) if strs is not None else None
        try:
            for s in range(first, last):
                r = s - first          # round number of this run (the gatherer is reset between runs)
                probe = rec is not None and timed and ((s - first) % probe_every == 0) and not selftest
                if strs is not None:
                    torch.cuda.set_stream(strs[s % n_str])
                ev = None
                if rec is not None and timed and not selftest:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                o = step_fn(s, g.slot(r), rec["events"] if probe else None, probe)
                if ev is not None:
                    ev[1].record()
                    rec["probe_ev" if probe else "frame_ev"].append(ev)
                if probe and o is not None:
                    rec["n_isects"].append(int(o["_isect_ids"].numel()))
                g.submit(r)
        finally:
            if home is not None:
                torch.cuda.set_stream(home)

    def barrier():
        if world > 1:
            dist.barrier()

    def timed_run(step_fn, g, n_str=n_streams, rec=None):
        run_steps(0, args.warmup, False, g, step_fn, n_str, rec)
        g.drain()
        g.reset()
        barrier()
        sync_streams()
        t0 = time.perf_counter()
        run_steps(args.warmup, total_steps, True, g, step_fn, n_str, rec)
        t_submitted = time.perf_counter()
        frames = g.drain()
        sync_streams()
        barrier()
        sync_streams()
        t1 = time.perf_counter()
        return t1 - t0, t1 - t_submitted, frames

    spec0 = dict(rendering._STATE.stats) if not selftest else None
    elapsed_local, drain_s, frames = timed_run(render_into, gatherer, rec=main_rec)
    spec = None if selftest else {k: rendering._STATE.stats[k] - spec0[k] for k in spec0}
    elapsed = elapsed_local
    per_rank_fps = [args.steps / elapsed_local]
    if world > 1:
        t = torch.tensor([elapsed_local], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank_fps = [args.steps / float(x.item()) for x in allt]
        elapsed = max(float(x.item()) for x in allt)
    if rank == 0:
        assert len(frames) == args.steps * world, (len(frames), args.steps, world)
        if selftest:
            for k, f in enumerate(frames):
                r_, w_ = divmod(k, world)
                expect = (7 * (w_ + (r_ + args.warmup) * world)) % 251
                assert int(f[0, 0, 0]) == expect and int(f.min()) == int(f.max()), (k, int(f[0, 0, 0]), expect)

    placements = None
    if world > 1 and not selftest:
        placements = [None] * world
        try:
            dist.all_gather_object(placements, dict(placement or {}, rank=rank, device=str(dev)))
        except Exception as e:      # noqa: BLE001  (diagnostics only: the measured value stands without them)
            placements = [f"not gathered ({type(e).__name__}: {e})"]
    gathered_ok = None
    if rank == 0 and world > 1 and not selftest:
        # every rank's first timed frame, as it arrived through the gather, against a local re-render of that frame
        gathered_ok = True
        with torch.no_grad():
            for k in range(world):
                cam_k = frame_camera(k + args.warmup * world, W, H).to(dev)
                local = to_uint8_frame(render_gaussians(scene, cam_k)["rgb"])
                gathered_ok = gathered_ok and bool(torch.equal(local, frames[k]))
    line = None
    if rank == 0:
        fps = args.steps * world / elapsed
        line = {
            "metric": METRIC, "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "per_rank_frames_per_s": per_rank_fps, "gathered_frames_match_local_render": gathered_ok,
            "ranks": None if placements is None else {
                "placement": placements,
                "rccl_version": _rccl_version(torch) if args.backend == "nccl" else None,
                "devices_visible_per_rank": "all (torch.cuda.set_device(LOCAL_RANK)): RCCL's xGMI transport maps peer buffers "
                                            "through hipIpc / peer access, which needs the peers enumerated in the process",
                "what": "dist.bind_rank before the first GPU call: host threads of every rank on the CPUs local to its GPU's "
                        "NUMA node (an equal share per rank on that node), torch intra-op threads capped; SC_BIND_CPUS=0 = off"},
            "gather": {"frames_per_collective": gatherer.batch, "collectives": gatherer.stats["gathers"],
                       "bytes_into_root_per_collective": gatherer.stats["bytes_per_gather"],
                       "host_ms_issuing_per_collective": (gatherer.stats["host_s_in_gather_calls"] * 1e3 /
                                                          max(1, gatherer.stats["gathers"])),
                       "host_ms_waiting_for_staging_ring": gatherer.stats["host_s_waiting_for_ring"] * 1e3,
                       "tail_ms_after_last_frame_was_submitted": drain_s * 1e3},
        }
    if selftest:
        if rank == 0:
            line.update({"data": "selftest-cpu (stand-in frames over gloo: launcher / sharding / gather plumbing only)",
                         "config": {"workload": "selftest", "parallelism": f"frames x{world}"},
                         "roofline": None, "selftest": True})
            emit(line)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- secondary measurements (N = 1 only; every one is K timed steps of its own) -----------------------
    secondary = {}
    if world == 1:
        from gsplat.rendering import rasterization
        from harness.caller import render_novel_view_u8
        op1 = scene.opacities[:, 0].contiguous()

        def fresh(shape=None):
            return FrameGatherer(shape or (H, W, 3), dev, dst=0, batch=args.gather_batch, reserve_rounds=reserve)

        def fused_into(s, out, events=None, intermediates=False):
            cam = cams[s]
            with torch.no_grad():
                rc, _, _ = rasterization(scene.means, scene.quats, scene.scales, op1, scene.sh, cam.viewmat[None],
                                         cam.K[None], W, H, near_plane=cam.znear, far_plane=cam.zfar,
                                         sh_degree=scene.sh_degree, render_mode="RGB+ED",
                                         rasterize_mode="antialiased", camera_centers_=cam.camera_center[None])
                to_uint8_frame(rc[0, ..., :3].permute(2, 0, 1), out=out)

        def measure(step_fn, n_str, what, compare=None, rec=None, shape=None):
            # SECONDARY lines only (the headline above is one timed region of exactly K steps, as the contract says): the
            # K steps are timed twice and the better run stands.  A 20-step region is 5-15 ms, and on the shared boxes
            # a single host hiccup of tens of ms has turned such a line into a fraction of itself (two_pass 692 instead of
            # 1470 frames/s, S-100k 269 instead of 3850, in two of ~20 runs of round 3); both runs are reported.
            el, _, fr = timed_run(step_fn, fresh(shape), n_str, rec)
            el2, _, fr2 = timed_run(step_fn, fresh(shape), n_str, None)
            runs = [el, el2]
            if el2 < el:
                el, fr = el2, fr2
            del fr2
            d = {"value": args.steps / el, "unit": "frames/s", "ms_per_step": el / args.steps * 1e3,
                 "frames_in_flight": n_str, "what": what, "ms_per_step_both_runs": [r / args.steps * 1e3 for r in runs]}
            if compare is not None:
                d["frames_identical_to_headline"] = bool(all(torch.equal(a, b) for a, b in zip(compare, fr)))
            return d, fr

        single_rec = None
        if streams is None and "two_in_flight" not in skip:
            streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        if "single_stream" not in skip and n_streams > 1:
            # one frame at a time (round 1's headline configuration): the probed kernels run ALONE
            single_rec = recorder()
            secondary["single_stream"], _ = measure(render_into, 1, "reference caller sequence, ONE frame in flight: "
                                                    "kernels run alone (round 1's headline configuration)", frames,
                                                    rec=single_rec)
        elif "two_in_flight" not in skip and n_streams == 1:
            secondary["two_frames_in_flight"], _ = measure(render_into, 2, "reference caller sequence, two frames in "
                                                           "flight on two HIP streams", frames)
        if "fused" not in skip:
            secondary["fused_rasterization"], _ = measure(
                fused_into, 1, "gsplat.rendering.rasterization(sh_degree, render_mode='RGB+ED', "
                "rasterize_mode='antialiased') -> uint8 frame; fused forward (DESIGN.md section 7)", frames)
            if "two_in_flight" not in skip:
                secondary["fused_rasterization"]["two_frames_in_flight"], _ = measure(fused_into, 2, "same, two frames in flight")
        if "two_pass" not in skip and not args.scene_ply:
            # render_novel_view's frame (renderer.py:136-163): foreground pass + sky pass + composite
            from street_crafter_amd.scenes import make_street_scene
            _, sky = make_street_scene(64, n_sky=max(1000, args.n_gauss // 32), sh_degree=args.sh_degree)
            sky = sky.to(dev)

            def two_pass_into(s, out, events=None, intermediates=False):
                render_novel_view_u8(scene, sky, cams[s], out=out, fused=False)

            def two_pass_fused_into(s, out, events=None, intermediates=False):
                render_novel_view_u8(scene, sky, cams[s], out=out, fused=True)

            d, fr = measure(two_pass_into, 1, f"render_novel_view frame: S-{args.n_gauss // 1000}k foreground pass + "
                            f"{sky.n}-Gaussian sky pass (radii of hundreds of px) + fused composite/clamp/uint8; "
                            "caller sequence per pass")
            d2, fr2 = measure(two_pass_fused_into, 1, "same, each pass through the fused rasterization()")
            d["fused_rasterization"] = d2
            d["fused_frames_identical"] = bool(all(torch.equal(a, b) for a, b in zip(fr, fr2)))
            if "two_in_flight" not in skip:
                # two frames in flight on two HIP streams for render.py's real frame
                d["two_frames_in_flight"], _ = measure(two_pass_into, 2, "same frames, two in flight", fr)
                d2["two_frames_in_flight"], _ = measure(two_pass_fused_into, 2, "same frames, two in flight", fr2)
            secondary["two_pass_frame"] = d
            del fr, fr2, sky
        if "street" not in skip and not args.scene_ply:
            # street-SHAPED scenes (ground plane, facades, horizon band, big soft splats): super-tiles far above
            # the in-LDS sort capacity and splat radii of hundreds of pixels, at 1x and 3x the headline size
            from street_crafter_amd.scenes import make_street_scene
            secondary["street_scene"] = {}
            for n_st in (args.n_gauss, 3 * args.n_gauss):
                fg, sky = make_street_scene(n_st, sh_degree=args.sh_degree)
                fg, sky = fg.to(dev), sky.to(dev)
                ev_st = {}
                key = (dev.index, 1, n_st, 16, math.ceil(W / 16), math.ceil(H / 16))
                rendering._STATE.last_meta.pop(key, None)          # (keyed by shape: S-1M's entry has the same key)

                def street_into(s, out, events=None, intermediates=False, fg=fg, ev_st=ev_st):
                    with torch.no_grad():
                        o = render_gaussians(fg, cams[s], stage_events=ev_st if s % 8 == 0 else None,
                                             return_intermediates=True)
                        to_uint8_frame(o["rgb"], out=out)
                    street_into.I = int(o["_isect_ids"].numel())

                d, _ = measure(street_into, 1, f"street-shaped scene, {n_st} foreground Gaussians, caller sequence, one frame in flight")
                torch.cuda.synchronize(dev)
                d["n_isects"] = street_into.I
                # median over the probe frames (every 8th, warm-up included: its first frame runs with exact sizes
                # after a host round trip, which a mean would carry)
                d["stage_ms"] = {k: sorted(a.elapsed_time(b) for a, b in v)[len(v) // 2] for k, v in ev_st.items()}
                d["stage_ms_is"] = "median over the probe frames"
                meta = rendering._STATE.last_meta.get(key)
                if meta:
                    d["largest_super_tile_records"] = meta[2]
                    d["isect_route"] = "bucketed"
                else:
                    d["isect_route"] = "radix (bucketed path returned SC_EUNSUPPORTED)"
                secondary["street_scene"][f"n{n_st}"] = d
                del fg, sky
        if "variants" not in skip and not args.scene_ply:
            # BASELINE config 1 (100 k Gaussians, configs/waymo_val_121.yaml:18 sh_degree 1) and SURVEY 8(d)'s "variants
            # to also report": K = 16 (sh_degree 3, the library default: street_gaussian/config/config.py:100) and the
            # reference's train / render resolution 1600 x 1066 (utils/camera_utils.py:150-152) -- same caller sequence,
            # the headline's timing (its number of frames in flight) and one frame at a time beside it
            secondary["variants"] = {}
            for vname, vn, vw, vh, vdeg, vwhat in (
                    ("s100k", 100_000, W, H, args.sh_degree, "BASELINE config 1: S-100k static forward raster"),
                    ("k16", args.n_gauss, W, H, 3, "SURVEY 8d variant: sh_degree 3 (K = 16 SH bases)"),
                    ("w1600", args.n_gauss, 1600, (1600 * H) // W, args.sh_degree,
                     "SURVEY 8d variant: the reference's render resolution (camera_utils.py:150-152: 1600 px wide, "
                     "int() of the scaled height)")):
                sc_v = scene if (vn == args.n_gauss and vdeg == args.sh_degree) else make_scene(vn, sh_degree=vdeg).to(dev)
                cams_v = [frame_camera(s, vw, vh).to(dev) for s in range(total_steps)]
                rec_v = recorder()

                def variant_into(s, out, events=None, intermediates=False, sc_v=sc_v, cams_v=cams_v):
                    with torch.no_grad():
                        o = render_gaussians(sc_v, cams_v[s], stage_events=events, return_intermediates=intermediates)
                        to_uint8_frame(o["rgb"], out=out)
                    return o

                vnf = n_streams if n_streams > 1 else 2      # the headline's number of frames in flight
                d2, _ = measure(variant_into, vnf, f"{vwhat}, {vn} Gaussians, {vw}x{vh}, sh_degree {vdeg}; caller "
                                f"sequence -> uint8 frame, {vnf} frames in flight (the headline's timing)", rec=rec_v,
                                shape=(vh, vw, 3))
                d1, _ = measure(variant_into, 1, "same, one frame in flight", shape=(vh, vw, 3))
                op_v = sc_v.opacities[:, 0].contiguous()

                def variant_fused_into(s, out, events=None, intermediates=False, sc_v=sc_v, cams_v=cams_v, op_v=op_v):
                    cam = cams_v[s]
                    with torch.no_grad():
                        rc, _, _ = rasterization(sc_v.means, sc_v.quats, sc_v.scales, op_v, sc_v.sh, cam.viewmat[None],
                                                 cam.K[None], cam.width, cam.height, near_plane=cam.znear, far_plane=cam.zfar,
                                                 sh_degree=sc_v.sh_degree, render_mode="RGB+ED",
                                                 rasterize_mode="antialiased", camera_centers_=cam.camera_center[None])
                        to_uint8_frame(rc[0, ..., :3].permute(2, 0, 1), out=out)

                df2, _ = measure(variant_fused_into, vnf, f"the same frames through the fused rasterization(), {vnf} in flight",
                                 shape=(vh, vw, 3))
                df1, _ = measure(variant_fused_into, 1, "same, one in flight", shape=(vh, vw, 3))
                torch.cuda.synchronize(dev)
                Kv, Pv, Tv = (vdeg + 1) ** 2, vw * vh, math.ceil(vw / 16) * math.ceil(vh / 16)
                I_v = sum(rec_v["n_isects"]) / max(len(rec_v["n_isects"]), 1)
                b_v = algorithmic_bytes(vn, int(I_v), vw, vh, 16, Kv)
                d2["single_stream"] = {"value": d1["value"], "ms_per_step": d1["ms_per_step"], "frames_in_flight": 1,
                                       "frac_of_hbm_roofline_wall": d1["value"] / (HBM_PEAK / b_v)}
                d2["fused_rasterization"] = {"value": df2["value"], "unit": "frames/s", "frames_in_flight": vnf,
                                             "single_stream": df1["value"], "what": df2["what"]}
                d2["n_isects_mean"] = I_v
                d2["frame_roofline"] = {"algorithmic_bytes_per_frame": b_v, "hbm_bound_fps_per_gpu": HBM_PEAK / b_v,
                                        "frac_of_hbm_roofline_wall": d2["value"] / (HBM_PEAK / b_v),
                                        "per_gaussian_bytes": 72 + 25 + 12 * Kv + 52}
                d2["stage_ms"] = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in rec_v["events"].items()}
                d2["stage_ms_is"] = f"mean over the probe frames (every 8th timed frame), {vnf} frames in flight"
                secondary["variants"][vname] = d2
                del sc_v, cams_v
        if "knn" not in skip:
            secondary["knn"] = knn_line(dev)
        if "train" not in skip and not args.scene_ply:
            # SURVEY 8(d) secondary: forward + backward steps/s at the reference's training resolution
            # (camera_utils.py:150-152: Waymo frames are trained at 1600 px width), L1 loss, all five parameter
            # groups requiring grad, absgrad on -- the shape of BASELINE config 2.
            from street_crafter_amd.scenes import make_camera
            tw_, th_ = 1600, (1600 * H) // W          # camera_utils.py:150-152: int(orig_h * 1600 / orig_w) = 1066
            tcam = make_camera(tw_, th_, 2050.0 * tw_ / 1920.0, 2050.0 * tw_ / 1920.0).to(dev)
            tscene = make_scene(args.n_gauss, sh_degree=args.sh_degree).to(dev)
            tparams = (tscene.means, tscene.quats, tscene.scales, tscene.opacities, tscene.sh)
            for t in tparams:
                t.requires_grad_(True)
            target = torch.rand(3, th_, tw_, device=dev)

            def train_step(fwd_ev=None, bwd_ev=None):
                for t in tparams:
                    t.grad = None
                # (the probe is set BEFORE the forward: a probed step runs the Python autograd.Functions, whose backward
                #  records the events; the timed steps run the C++ ones of the binding layer)
                prev = rendering.set_backward_probe(bwd_ev)
                try:
                    out = render_gaussians(tscene, tcam, mode="train", stage_events=fwd_ev, return_intermediates=fwd_ev is not None)
                    ((out["rgb"] - target).abs().mean() + 0.01 * out["acc"].mean()).backward()
                finally:
                    rendering.set_backward_probe(prev)
                return out

            for _ in range(3):
                train_step()
            torch.cuda.synchronize()
            ms0 = torch.cuda.memory_stats(dev)
            n_train = min(args.steps, 20)
            train_runs = []
            for _ in range(2):                     # timed twice, the better run stands (see measure())
                t1 = time.perf_counter()
                host_t = []
                for _ in range(n_train):
                    h0 = time.perf_counter()
                    train_step()
                    host_t.append(time.perf_counter() - h0)
                t_enq = time.perf_counter() - t1
                torch.cuda.synchronize()
                train_runs.append(time.perf_counter() - t1)
            el = min(train_runs)
            if args.stage_times:
                ms1 = torch.cuda.memory_stats(dev)
                import gc
                print("train: host enqueue per step (ms)", [round(x * 1e3, 2) for x in host_t], "enqueue total", round(t_enq * 1e3, 2),
                      "wall", round(el * 1e3, 2), "device mallocs during the timed steps",
                      ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0), "alloc retries",
                      ms1.get("num_alloc_retries", 0) - ms0.get("num_alloc_retries", 0), "reserved GB",
                      ms1.get("reserved_bytes.all.current", 0) / 1e9, "gc counts", gc.get_count(), "gc objects", len(gc.get_objects()),
                      file=sys.stderr)
            # probe steps AFTER the timed ones: HIP events around every forward operator and every backward operator
            fwd_ev, bwd_ev = {}, {}
            for _ in range(5):
                o_t = train_step(fwd_ev, bwd_ev)
            torch.cuda.synchronize()
            I_t = int(o_t["_isect_ids"].numel())
            P_t = tw_ * th_
            med = lambda v: sorted(a.elapsed_time(b) for a, b in v)[len(v) // 2]      # noqa: E731
            st_ms = {k: med(v) for k, v in fwd_ev.items()}
            st_ms.update({k: med(v) for k, v in bwd_ev.items()})
            bwd_ms = st_ms.get("rasterize_to_pixels_bwd")
            bwd_bytes = raster_bwd_algorithmic_bytes(I_t, P_t)
            tkey = f"train_n{args.n_gauss}_{tw_}x{th_}"
            secondary["train_fwd_bwd"] = {
                "value": n_train / el, "unit": "steps/s", "ms_per_step": el / n_train * 1e3,
                "ms_per_step_both_runs": [r / n_train * 1e3 for r in train_runs],
                "what": f"render (train mode) + L1 loss + backward, {args.n_gauss} Gaussians, {tw_}x{th_}, absgrad "
                        "(train.py:236; BASELINE config 2's shape); the better of two timed runs of the same steps",
                "n_isects": I_t, "stage_ms": st_ms,
                "stage_ms_is": "median of 5 probe steps after the timed ones: HIP events around every forward operator "
                               "(harness) and every backward operator (rendering.set_backward_probe), kernels running alone",
                "device_ms_operators_sum": sum(st_ms.values()),
                "_traffic_key": tkey, "_bwd_bytes": bwd_bytes, "_bwd_ms": bwd_ms, "_pairs": 256 * I_t}
            del tscene, tparams, target, o_t

    if world == 1 and not selftest and "forced_gather" not in skip and not args.force_gather:
        # (last of the secondary measurements: RCCL's threads exist from here on)
        # the RCCL transport on the one GPU of this box: a world of ONE through the real staging ring + async
        # dist.gather (VERDICT r2 missing 1).  Not a scaling number -- there is none until an 8-GPU node runs this.
        try:
            init_world1(dev)
            gf = FrameGatherer((H, W, 3), dev, dst=0, batch=args.gather_batch, force_collective=True, reserve_rounds=reserve)
            el, _, fr = timed_run(render_into, gf, n_streams)
            secondary["forced_gather_world1"] = {
                "value": args.steps / el, "unit": "frames/s", "ms_per_step": el / args.steps * 1e3,
                "frames_in_flight": n_streams, "backend": dist.get_backend(), "collectives": gf.stats["gathers"],
                "frames_per_collective": gf.batch,
                "frames_identical_to_headline": bool(all(torch.equal(a, b) for a, b in zip(frames, fr))),
                "what": "the headline loop with init_process_group('nccl', world_size=1) and the world == 1 short cut of "
                        "the gatherer disabled: every frame goes through the staging ring and an async dist.gather on "
                        "RCCL's stream (at one rank the collective is a device copy; no bytes cross xGMI)"}
            del fr, gf
        except Exception as e:          # noqa: BLE001  (a box whose RCCL cannot initialise must not cost the bench line)
            secondary["forced_gather_world1"] = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- per-operator device time from the probe frames' HIP events ----------------------------------------
    torch.cuda.synchronize(dev)
    P, T = W * H, math.ceil(W / 16) * math.ceil(H / 16)
    traffic_json = {}
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tj):
        try:
            traffic_json = json.load(open(tj))
        except Exception:
            traffic_json = {}

    def pmc_traffic(op, key, kernel_symbol):
        return pmc_traffic_entry(traffic_json, op, key, kernel_symbol)

    def kernel_report(rec, in_flight):
        """stage_ms / operators / roofline of one run's probe frames."""
        ev = rec["events"]
        stage = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in ev.items()}
        I_m = sum(rec["n_isects"]) / max(len(rec["n_isects"]), 1)
        ops = {}
        for k, ms in stage.items():
            by = stage_algorithmic_bytes(k, args.n_gauss, I_m, P, T, K)
            ops[k] = {"ms": ms, "algorithmic_bytes": by, "gb_per_s": by / (ms * 1e-3) / 1e9,
                      "frac_of_hbm_peak": by / (ms * 1e-3) / HBM_PEAK, "kernels": OPERATOR_KERNEL.get(k)}
        # the dominant KERNEL (not operator): the only operators that are one launch each are projection, SH and
        # rasterize, and the intersection operator's largest kernel is shorter than the rasterizer (profiles/)
        dom = "rasterize_to_pixels"
        dom_ms = stage[dom]
        dom_bytes = stage_algorithmic_bytes(dom, args.n_gauss, I_m, P, T, K)
        traffic, tsrc = pmc_traffic(dom, f"n{args.n_gauss}", OPERATOR_KERNEL[dom])
        ratio = None if traffic is None else traffic / dom_bytes
        roof = {"bound": "hbm", "kernel": OPERATOR_KERNEL[dom],
                "achieved": dom_bytes / (dom_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": dom_bytes / (dom_ms * 1e-3) / HBM_PEAK, "traffic": traffic,
                "traffic_source": tsrc,
                "traffic_over_algorithmic": ratio,
                "real_hbm_gb_per_s": None if traffic is None else traffic / (dom_ms * 1e-3) / 1e9,
                "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_ms,
                "launches_timed": len(ev.get(dom, [])), "frames_in_flight": in_flight,
                "how": "HIP events around the rasterize_to_pixels operator (= this one kernel launch) on the probe "
                       "frames of the timed region" + ("; the kernel runs alone" if in_flight == 1 else
                       f"; {in_flight} frames in flight: the kernel shares the GPU with the other stream's frame while "
                       "it is timed (rocprofv3 of the same command sees the same)") +
                       "; algorithmic bytes = 44 B x I (id + xy + opacity + conic + colour gathered per intersection) "
                       "+ 24 B x P (SURVEY 8d)",
                "note": "an HBM-EQUIVALENT rate: the kernel is VALU-bound, not bandwidth-bound.  Tiles terminate after "
                        "~12 % of their lists, so the bytes it really moves are " +
                        ("unknown here (no valid PMC record)" if ratio is None else
                         f"{ratio:.3f} of the algorithmic figure (`traffic` / `algorithmic_bytes_per_launch`), i.e. "
                         f"{traffic / (dom_ms * 1e-3) / HBM_PEAK:.2f} of the HBM peak") +
                        "; its floor is VALU issue (DESIGN.md section 4)"}
        return stage, ops, roof, I_m

    stage_ms, operators, roofline, I_mean = kernel_report(main_rec, n_streams)
    b_alg = algorithmic_bytes(args.n_gauss, int(I_mean), W, H, 16, K)

    if rank == 0:
        fps = line["value"]
        if world == 1 and secondary.get("single_stream") is not None and single_rec is not None:
            st1, ops1, roof1, _ = kernel_report(single_rec, 1)
            secondary["single_stream"].update({"stage_ms": st1, "operators": ops1, "roofline": roof1,
                                               "frame_ms_device": percentiles([a.elapsed_time(b) for a, b in
                                                                               single_rec["frame_ev"]])})
        ss = secondary.get("single_stream")
        if world == 1 and ss is not None:
            # SURVEY 8(d)'s literal timing (one frame at a time on the current stream), kept NEXT TO the headline keys so that a
            # truncated tail of the line cannot lose it; the full record stays under "single_stream"
            line["single_stream_summary"] = {
                "value": ss["value"], "unit": "frames/s", "ms_per_step": ss["ms_per_step"], "frames_in_flight": 1,
                "frac_of_hbm_roofline_wall": ss["value"] / (HBM_PEAK / b_alg),
                "what": "the same K frames ONE at a time (SURVEY 8d's timing method); `value` above keeps "
                        f"{n_streams} frames in flight"}
        line.update({
            "config": {"workload": (f"scene file {os.path.basename(args.scene_ply)} ({args.n_gauss} Gaussians)"
                                    if args.scene_ply else f"S-{args.n_gauss // 1000}k") +
                                   f" static gsplat forward raster, 1 cam {W}x{H}, sh_degree {args.sh_degree}, "
                                   f"tile 16, RGB+depth, antialiased; the reference caller's operator sequence -> "
                                   f"uint8 frame; one frame per GPU per step, {n_streams} frames in flight per GPU, "
                                   f"uint8 frames gathered to rank 0 {gatherer.batch} per collective",
                       "n_gaussians": args.n_gauss, "n_isects_mean": I_mean, "rho": I_mean / args.n_gauss,
                       "isect_mode": rendering._ISECT_MODE["mode"], "frames_in_flight": n_streams,
                       "render_colors_storage": "planar (one plane per channel behind the [C,H,W,D] indexing)"
                                                if rendering._SWITCH.planar_out else "interleaved",
                       "raster_side_stream_cus": args.raster_cus,
                       "isect_speculation": dict(spec, what="isect_tiles calls of the headline run incl. warm-up: "
                                                 "speculative_ok = the scatter + sort launched with sizes predicted "
                                                 "from the previous frame ran (no host wait in front of them); "
                                                 "exact_relaunch = the prediction was too small and they were "
                                                 "launched again with exact sizes; the rest had no prediction yet"),
                       "isect_ids": "lazy (written on first read; nothing on this path reads them)"
                                    if rendering._SWITCH.lazy_ids else "written by the sort",
                       "parallelism": f"frames x{world}"},
            "roofline": roofline,
            "operators": operators,
            "frame_roofline": {"algorithmic_bytes_per_frame": b_alg,
                               "hbm_bound_fps_per_gpu": HBM_PEAK / b_alg,
                               "frac_of_hbm_roofline_wall": (fps / world) / (HBM_PEAK / b_alg),
                               "valu_pair_bound": 256 * I_mean},
            "stage_ms": stage_ms,
            "frame_ms_device": {**(percentiles([a.elapsed_time(b) for a, b in main_rec["frame_ev"]]) or {}),
                                "what": f"HIP events around every timed frame that is not a probe frame (operators + "
                                        f"torch glue + uint8 conversion), {n_streams} frame(s) in flight: with more "
                                        f"than one, the latency of a frame while it shares the GPU"},
            "probe_frame_ms_device": {**(percentiles([a.elapsed_time(b) for a, b in main_rec["probe_ev"]]) or {}),
                                      "what": "the probe frames (10 more event pairs and the intermediates kept)"},
        })
        tr = secondary.get("train_fwd_bwd")
        if tr is not None and "_bwd_ms" in tr:
            bwd_ms, bwd_bytes, tkey, pairs = tr.pop("_bwd_ms"), tr.pop("_bwd_bytes"), tr.pop("_traffic_key"), tr.pop("_pairs")
            ksym = "raster_bwd_wave_kernel<4>"
            if bwd_ms:
                traffic, tsrc = pmc_traffic("rasterize_to_pixels_bwd", tkey, ksym)
                tr["roofline"] = {
                    "bound": "hbm", "kernel": ksym, "achieved": bwd_bytes / (bwd_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9,
                    "unit": "GB/s", "frac": bwd_bytes / (bwd_ms * 1e-3) / HBM_PEAK, "traffic": traffic,
                    "traffic_source": tsrc, "traffic_over_algorithmic": None if traffic is None else traffic / bwd_bytes,
                    "algorithmic_bytes_per_launch": bwd_bytes, "avg_launch_ms": bwd_ms, "launches_timed": 5,
                    "valu_pair_bound": pairs,
                    "how": "HIP events around sc_rasterize_bwd (= this one kernel launch) inside autograd's backward, on "
                           "5 probe steps after the timed ones (rendering.set_backward_probe); algorithmic bytes = 92 B x I "
                           "(replayed 44-B gather + one 48-B set of gradient atomics per (Gaussian, tile)) + 28 B x P "
                           "(DESIGN.md section 4)",
                    "note": "HBM-EQUIVALENT like the forward's: the replay walks only the part of each list the forward "
                            "blended, and its cost is VALU (the same pairs with ~2.5x the instructions) plus one 12-lane "
                            "atomic per kept splat and tile"}
        line.update(secondary)
        if world == 1 and "cpu_python" not in skip and not args.no_cpu_baseline:
            line["cpu_baseline_python"] = cpu_baseline_python(W, H, frames=max(1, args.cpu_python_frames))
        if world == 1 and not args.no_cpu_baseline and not args.headline_only:
            def hip_frame():
                with torch.no_grad():
                    o = render_gaussians(scene, cams[0], return_intermediates=True)
                return {"rgb": o["rgb"].permute(1, 2, 0).cpu().numpy(), "isect_ids": o["_isect_ids"].cpu().numpy(),
                        "flatten_ids": o["_flatten_ids"].cpu().numpy(), "radii": o["_radii"][0].cpu().numpy(),
                        "render_colors": o["_render_colors"][0].cpu().numpy(),
                        "render_alphas": o["_render_alphas"][0].cpu().numpy(), "colors": o["_colors"][0].cpu().numpy(),
                        "opacities": o["_opacities"][0].cpu().numpy(), "means2d": o["_means2d"][0].cpu().numpy(),
                        "conics": o["_conics"][0].cpu().numpy()}

            line["cpu_baseline"], line["parity"] = cpu_baseline(scene, cams[0], W, H, hip_frame)
        if args.stage_times:
            print(json.dumps(stage_ms, indent=1), file=sys.stderr)
        emit(line)
    if dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
