#!/usr/bin/env python3
"""Throughput / roofline benchmark of the MI355X Gaussian-splat hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = every rank renders ONE frame of the synthetic S-1M scene (1 000 000 Gaussians,
1920x1280, SURVEY.md 8d / BASELINE.md 2) through the drop-in gsplat operators (the caller's
sequence of street_gaussian_renderer.py:186-302, forward only) with a per-frame camera, and the
finished uint8 frames are gathered to rank 0 (RCCL).  Scene tensors are resident in HBM before
the timed region.  Rank 0 prints ONE JSON line.
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n-gauss", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1280)
    ap.add_argument("--sh-degree", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--isect-mode", choices=["bin", "radix"], default=None)
    ap.add_argument("--raster-variant", type=int, default=None)
    ap.add_argument("--stage-times", action="store_true", help="print per-operator times to stderr")
    ap.add_argument("--no-overlap", action="store_true",
                    help="skip the secondary measurement with two frames in flight on two HIP streams")
    ap.add_argument("--no-fused", action="store_true",
                    help="skip the secondary fused rasterization() measurement (keeps profiles of the headline clean)")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary fwd+bwd (training-step) measurement")
    ap.add_argument("--scene-ply", default=None,
                    help="render a scene file in the reference's point_cloud.ply layout instead of S-<n>")
    return ap.parse_args()


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


def cpu_baseline(width, height, hip_frame_fn):
    """Oracle (numpy, 1 thread) on S-100k at full resolution: one frame, ~10-30 s of CPU work."""
    import numpy as np
    from oracle import gsplat_oracle as O        # checker / baseline only
    from street_crafter_amd.scenes import make_camera, make_scene
    sc = make_scene(100_000)
    cam = make_camera(width, height, 2050.0 * width / 1920.0, 2050.0 * width / 1920.0)
    t0 = time.perf_counter()
    exp = O.render_frame(sc.means.numpy(), sc.quats.numpy(), sc.scales.numpy(), sc.opacities.numpy(),
                         sc.sh.numpy(), cam.viewmat.numpy(), cam.K.numpy(), width, height, sc.sh_degree,
                         near_plane=cam.znear, far_plane=cam.zfar, return_unstable=True)
    dt = time.perf_counter() - t0
    got = hip_frame_fn(sc, cam)
    ref_rgb = np.clip(exp["render_colors"][0, ..., :3], 0.0, 1.0)
    err = np.abs(got["rgb"] - ref_rgb).max(axis=-1)
    # pixels where some alpha / transmittance sits within 2e-5 (relative) of a hard threshold can
    # legitimately flip on a 1-ulp exp difference (oracle flags them); reported separately
    stable = ~exp["unstable"][0]
    parity = {
        "psnr_db_vs_oracle": O.psnr(got["rgb"], ref_rgb),
        "max_abs_rgb_stable_pixels": float(err[stable].max()),
        "max_abs_rgb_all_pixels": float(err.max()),
        "threshold_unstable_pixels": int((~stable).sum()),
        "pixels_over_1e-4": int((err > 1e-4).sum()),
        "n_pixels": int(err.size),
        "isect_ids_and_flatten_ids_bit_exact": bool(np.array_equal(got["isect_ids"], exp["isect_ids"]) and
                                                    np.array_equal(got["flatten_ids"], exp["flatten_ids"])),
    }
    return dt, int(exp["isect_ids"].shape[0]), parity


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a GPU (HIP); there is no CPU path"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from street_crafter_amd import _lib, rendering
    from street_crafter_amd.dist import FrameGatherer, to_uint8_frame
    from street_crafter_amd.pipeline import algorithmic_bytes, render_gaussians
    from street_crafter_amd.scenes import make_scene
    _lib.load()
    if args.isect_mode:
        rendering.set_isect_mode(args.isect_mode)
    if args.raster_variant is not None:
        _lib.set_option("raster_fwd", args.raster_variant)

    W, H = args.width, args.height
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
    total_steps = args.warmup + args.steps
    cams = [frame_camera(rank + s * world, W, H).to(dev) for s in range(total_steps)]
    gatherer = FrameGatherer(dst=0)
    events = {}
    n_isects = []
    frame_events = []

    def step(s, timed):
        # per-operator HIP events cost ~200 us of host time per frame (10 event pairs), so they are
        # recorded on every 8th timed step only; the kernels and the stream are the same either way
        probe = timed and ((s - args.warmup) % 8 == 0)
        with torch.no_grad():
            if probe:
                f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                f0.record()
            out = render_gaussians(scene, cams[s], stage_events=events if probe else None,
                                   return_intermediates=probe)
            if probe:
                n_isects.append(int(out["_isect_ids"].numel()))
            gatherer.submit(s, to_uint8_frame(out["rgb"]))
            if probe:
                f1.record()
                frame_events.append((f0, f1))

    for s in range(args.warmup):
        step(s, False)
    gatherer.drain()
    gatherer._done.clear()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.warmup, total_steps):
        step(s, True)
    frames = gatherer.drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        assert len(frames) == args.steps * world, (len(frames), args.steps, world)

    # SURVEY 8f-2 (secondary, N = 1 only): the same frames through gsplat's one-call `rasterization()`,
    # whose forward is fused (projection + SH + glue in one kernel, depth normalisation in the raster
    # epilogue).  `value` above stays the reference caller's own operator sequence.
    fused_line = None
    if world == 1 and not args.no_fused:
        from gsplat.rendering import rasterization
        op1 = scene.opacities[:, 0].contiguous()

        def fused_step(s):
            cam = cams[s]
            with torch.no_grad():
                rc, _, _ = rasterization(scene.means, scene.quats, scene.scales, op1, scene.sh, cam.viewmat[None],
                                         cam.K[None], W, H, near_plane=cam.znear, far_plane=cam.zfar,
                                         sh_degree=scene.sh_degree, render_mode="RGB+ED",
                                         rasterize_mode="antialiased", camera_centers_=cam.camera_center[None])
                return to_uint8_frame(rc[0, ..., :3].permute(2, 0, 1))

        for s in range(args.warmup):
            fused_step(s)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fused_frames = [fused_step(s) for s in range(args.warmup, total_steps)]
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        same = all(torch.equal(a, b) for a, b in zip(frames, fused_frames))
        fused_line = {"value": args.steps / el, "unit": "frames/s", "ms_per_step": el / args.steps * 1e3,
                      "frames_identical_to_caller_sequence": bool(same),
                      "what": "gsplat.rendering.rasterization(sh_degree, render_mode='RGB+ED', "
                              "rasterize_mode='antialiased') -> uint8 frame; fused forward (DESIGN.md section 4)"}
        del fused_frames

    # Secondary (N = 1 only): the same K frames with TWO frames in flight, frame f on HIP stream f % 2.
    # Frames are independent (that is what the multi-GPU sharding relies on); the second stream lets the
    # latency-bound intersection kernels of one frame run under the VALU-bound rasterizer of the other.
    # Not the headline: with overlapped frames per-operator durations (and so `roofline`) lose their meaning.
    overlap_line = None
    if world == 1 and not args.no_overlap:
        two = [torch.cuda.Stream(device=dev) for _ in range(2)]

        def timed_overlapped(fn):
            for s in range(args.warmup):
                with torch.cuda.stream(two[s % 2]):
                    fn(s)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            outs = []
            for s in range(args.warmup, total_steps):
                with torch.cuda.stream(two[s % 2]):
                    outs.append(fn(s))
            torch.cuda.synchronize()
            return time.perf_counter() - t1, outs

        def caller_step(s):
            with torch.no_grad():
                return to_uint8_frame(render_gaussians(scene, cams[s])["rgb"])

        el, outs = timed_overlapped(caller_step)
        overlap_line = {"value": args.steps / el, "unit": "frames/s", "ms_per_step": el / args.steps * 1e3,
                        "frames_identical_to_single_stream": bool(all(torch.equal(a, b) for a, b in zip(frames, outs))),
                        "what": "reference caller sequence, two frames in flight on two HIP streams"}
        del outs
        if fused_line is not None:
            el, outs = timed_overlapped(fused_step)
            overlap_line["fused_rasterization"] = {"value": args.steps / el, "ms_per_step": el / args.steps * 1e3}
            del outs

    # SURVEY 8(d) secondary (N = 1 only): forward + backward steps/s at the reference's training resolution
    # (camera_utils.py:150-152: Waymo frames are trained at 1600 px width), L1 loss, all five parameter
    # groups requiring grad, absgrad on -- the shape of BASELINE config 3.
    train_line = None
    if world == 1 and not args.no_train and not args.scene_ply:
        from street_crafter_amd.scenes import make_camera
        tw_, th_ = 1600, int(round(1600 * H / W))
        tcam = make_camera(tw_, th_, 2050.0 * tw_ / 1920.0, 2050.0 * tw_ / 1920.0).to(dev)
        tscene = make_scene(args.n_gauss, sh_degree=args.sh_degree).to(dev)
        tparams = (tscene.means, tscene.quats, tscene.scales, tscene.opacities, tscene.sh)
        for t in tparams:
            t.requires_grad_(True)
        target = torch.rand(3, th_, tw_, device=dev)

        def train_step():
            for t in tparams:
                t.grad = None
            out = render_gaussians(tscene, tcam, mode="train")
            ((out["rgb"] - target).abs().mean() + 0.01 * out["acc"].mean()).backward()

        for _ in range(3):
            train_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_train = min(args.steps, 20)
        for _ in range(n_train):
            train_step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        train_line = {"value": n_train / el, "unit": "steps/s", "ms_per_step": el / n_train * 1e3,
                      "what": f"render (train mode) + L1 loss + backward, {args.n_gauss} Gaussians, {tw_}x{th_}, absgrad"}
        del tscene, tparams, target

    # per-operator device time from the HIP events recorded inside the timed region
    stage_ms = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in events.items()}
    I_mean = sum(n_isects) / max(len(n_isects), 1)
    P, T = W * H, math.ceil(W / 16) * math.ceil(H / 16)
    dominant = max(stage_ms, key=stage_ms.get)
    dom_bytes = stage_algorithmic_bytes(dominant, args.n_gauss, I_mean, P, T, K)
    dom_gbs = dom_bytes / (stage_ms[dominant] * 1e-3) / 1e9
    b_alg = algorithmic_bytes(args.n_gauss, int(I_mean), W, H, 16, K)
    device_ms = sum(stage_ms.values())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        fps = args.steps * world / elapsed
        traffic = None
        tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tj):
            try:
                traffic = json.load(open(tj)).get(dominant, {}).get(f"n{args.n_gauss}")
            except Exception:
                traffic = None
        line = {
            "metric": "frames/sec @1M Gaussians 1920x1280 + PSNR vs ref; 1/2/4/8 MI355X",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"scene file {os.path.basename(args.scene_ply)} ({args.n_gauss} Gaussians)"
                                    if args.scene_ply else f"S-{args.n_gauss // 1000}k") +
                                   f" static gsplat forward raster, 1 cam {W}x{H}, "
                                   f"sh_degree {args.sh_degree}, tile 16, RGB+depth, antialiased; one frame per "
                                   f"GPU per step, uint8 frames gathered to rank 0",
                       "n_gaussians": args.n_gauss, "n_isects_mean": I_mean, "rho": I_mean / args.n_gauss,
                       "isect_mode": rendering._ISECT_MODE["mode"], "parallelism": f"frames x{world}"},
            "roofline": {"bound": "hbm", "kernel": dominant + " (operator; its kernels: DESIGN.md section 4)", "achieved": dom_gbs, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": dom_gbs / (HBM_PEAK / 1e9), "traffic": traffic,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": stage_ms[dominant]},
            "frame_roofline": {"algorithmic_bytes_per_frame": b_alg,
                               "hbm_bound_fps_per_gpu": HBM_PEAK / b_alg,
                               "frac_of_hbm_roofline_wall": (fps / world) / (HBM_PEAK / b_alg),
                               "frac_of_hbm_roofline_device": (1e3 / device_ms) / (HBM_PEAK / b_alg),
                               "valu_pair_bound": 256 * I_mean},
            "stage_ms": stage_ms,
        }
        if frame_events:
            fm = sorted(a.elapsed_time(b) for a, b in frame_events)
            pick = lambda q: fm[min(len(fm) - 1, int(round(q * (len(fm) - 1))))]
            line["frame_ms_device"] = {"p10": pick(0.1), "p50": pick(0.5), "p90": pick(0.9), "samples": len(fm),
                                       "what": "HIP events around whole frames (every 8th timed step, the ones "
                                               "that also carry the per-operator events)"}
        if fused_line is not None:
            line["fused_rasterization"] = fused_line
        if overlap_line is not None:
            line["two_frames_in_flight"] = overlap_line
        if train_line is not None:
            line["train_fwd_bwd"] = train_line
        if world == 1 and not args.no_cpu_baseline:
            import numpy as np

            def hip_frame(sc, cam):
                with torch.no_grad():
                    o = render_gaussians(sc.to(dev), cam.to(dev), return_intermediates=True)
                return {"rgb": o["rgb"].permute(1, 2, 0).cpu().numpy(), "isect_ids": o["_isect_ids"].cpu().numpy(),
                        "flatten_ids": o["_flatten_ids"].cpu().numpy()}

            dt, I100k, parity = cpu_baseline(W, H, hip_frame)
            scale = I100k / max(I_mean, 1.0)
            line["cpu_baseline"] = {
                "value": (1.0 / dt) * scale, "unit": "frames/s", "cores": 1, "kind": "port",
                "sample": f"oracle/gsplat_oracle.py (numpy, 1 thread) on ONE frame of S-100k {W}x{H} "
                          f"(I={I100k}) took {dt:.2f} s; value = 1/that, scaled by I_100k/I_1M={scale:.4f} "
                          f"to the S-1M unit (work ~ intersections)",
                "measured_fps_at_100k": 1.0 / dt, "host_cpus": os.cpu_count()}
            line["parity_100k"] = parity
        if args.stage_times:
            print(json.dumps(stage_ms, indent=1), file=sys.stderr)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
