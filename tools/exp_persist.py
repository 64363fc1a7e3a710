"""(Needs tools/patches/r04_raster_persist.diff applied: the shipped library has no "raster_persist" option.)
A/B of the PERSISTENT forward rasterizer (sc_set_option "raster_persist" = one-wave workgroups per CU that pop the dispatch
list) through the reference caller sequence -> uint8 frame: frames/s with 1 and 3 frames in flight; frames compared bit for bit.
    python tools/exp_persist.py [values,comma] [n_gauss|street] [frames] [rounds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from harness.caller import render_gaussians
from street_crafter_amd import _lib
from street_crafter_amd.dist import to_uint8_frame
from street_crafter_amd.scenes import make_scene, make_street_scene
VALUES = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,8,12,16,20").split(",")]
WHICH = sys.argv[2] if len(sys.argv) > 2 else "1000000"
FRAMES = int(sys.argv[3]) if len(sys.argv) > 3 else 60
ROUNDS = int(sys.argv[4]) if len(sys.argv) > 4 else 2
W, H, WARM = 1920, 1280, 6
dev = torch.device("cuda", 0)
scene = (make_street_scene(1_000_000)[0] if WHICH == "street" else make_scene(int(WHICH))).to(dev)
cams = [bench.frame_camera(s, W, H).to(dev) for s in range(FRAMES + WARM)]
out = torch.empty((FRAMES + WARM, H, W, 3), dtype=torch.uint8, device=dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]


def run(val, nstr):
    prev = _lib.set_option("raster_persist", val)
    home = torch.cuda.current_stream(dev)
    with torch.no_grad():
        for s in range(FRAMES + WARM):
            if s == WARM:
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
            if nstr > 1:
                torch.cuda.set_stream(streams[s % nstr])
            to_uint8_frame(render_gaussians(scene, cams[s])["rgb"], out=out[s])
    torch.cuda.set_stream(home)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    _lib.set_option("raster_persist", prev)
    return FRAMES / el, out[WARM:].clone()


res, ref = {}, None
for r in range(ROUNDS):
    for nstr in (1, 3):
        for v in VALUES:
            fps, fr = run(v, nstr)
            res.setdefault((nstr, v), []).append(fps)
            if ref is None:
                ref = fr
            elif not torch.equal(ref, fr):
                print("FRAMES DIFFER", nstr, v, flush=True)
for (nstr, v), x in sorted(res.items()):
    x = sorted(x)
    print(f"{nstr} in flight, raster_persist {v:2d}: best {x[-1]:7.1f} frames/s  (worst {x[0]:7.1f})", flush=True)
