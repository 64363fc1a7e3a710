"""Where the waves of the forward rasterizer spend their lives (diagnostic build, "debug1" bit 4: shader-clock cycles
summed over all waves of raster_fwd_wave_kernel, csrc/raster_fwd.hip g_sc_phase_cycles).

    python -m street_crafter_amd.build --diag && python tools/exp_raster_phases.py [s1m|s100k|street1m|sky] [frames]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from street_crafter_amd import _lib  # noqa: E402

_lib.use_diagnostic_build()
from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_scene, make_street_scene  # noqa: E402
import bench  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "s1m"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
W, H = 1920, 1280
sc = {"s1m": lambda: make_scene(1_000_000), "s100k": lambda: make_scene(100_000),
      "street1m": lambda: make_street_scene(1_000_000)[0], "sky": lambda: make_street_scene(1_000_000)[1]}[which]().to("cuda")
cams = [bench.frame_camera(s, W, H).to("cuda") for s in range(frames + 4)]
lib = _lib.load()
fn = lib.sc_diag_phase_cycles
fn.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
fn.restype = ctypes.c_int
out = (ctypes.c_uint64 * 10)()
ev = {}
with torch.no_grad():
    for s in range(4):
        render_gaussians(sc, cams[s])
    _lib.set_option("debug1", 4)
    assert fn(out, 1) == 0
    for s in range(4, frames + 4):
        render_gaussians(sc, cams[s], stage_events=ev)
assert fn(out, 1) == 0
_lib.set_option("debug1", 0)
total, issue, cull, blend, waves, batches, splats, first, ticks, _ = [int(v) / frames for v in out]
us = sorted(a.elapsed_time(b) for a, b in ev["rasterize_to_pixels"])[frames // 2] * 1e3
per = total / max(waves, 1)
print(f"# {which}: {frames} frames, per frame: {waves:.0f} waves, {batches:.0f} batches ({batches / max(waves, 1):.2f} per wave), "
      f"{splats:.0f} records kept by the cull ({splats / max(batches, 1):.1f} per batch); operator {us:.1f} us with the clock reads in")
print(f"# mean wave lifetime {per:.0f} cycles; of it: entry -> first batch's parameters there {first / total:.3f}, "
      f"cull + compaction {cull / total:.3f}, issuing the next gathers {issue / total:.3f}, blend loop {blend / total:.3f}, "
      f"epilogue and clock reads {(total - first - issue - cull - blend) / total:.3f}")
print(f"# per batch: cull + compaction {cull / max(batches, 1):.0f}, issue {issue / max(batches, 1):.0f}, blend {blend / max(batches, 1):.0f} cycles; "
      f"blend per kept record {blend / max(splats, 1):.0f} cycles of wave time")
print(f"# shader clock while the kernel ran: {100.0 * total / max(ticks, 1):.0f} MHz (shader cycles / 100 MHz ticks over the waves' lifetimes)")
print(f"# wave-cycles per frame {total:.3e} = {total / (256 * 4):.0f} per SIMD if spread evenly")
