"""A/B of rasterize_to_pixels' planar output (rendering.set_planar_output) through the reference caller sequence -> uint8
frame: frames/s one frame at a time and three in flight, arms alternating in one process; frames compared bit for bit.
    python tools/exp_planar.py [n_gauss] [frames] [rounds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from harness.caller import render_gaussians
from street_crafter_amd import rendering
from street_crafter_amd.dist import to_uint8_frame
from street_crafter_amd.scenes import make_scene
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
FRAMES = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
W, H, WARM = 1920, 1280, 6
dev = torch.device("cuda", 0)
scene = make_scene(N).to(dev)
cams = [bench.frame_camera(s, W, H).to(dev) for s in range(FRAMES + WARM)]
out = torch.empty((FRAMES + WARM, H, W, 3), dtype=torch.uint8, device=dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]


def run(planar, nstr):
    prev = rendering.set_planar_output(planar)
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
    rendering.set_planar_output(prev)
    return FRAMES / el, out[WARM:].clone()


res = {}
ref = None
for r in range(ROUNDS):
    for nstr in (1, 3):
        for planar in (False, True):
            fps, fr = run(planar, nstr)
            res.setdefault((nstr, planar), []).append(fps)
            if ref is None:
                ref = fr
            elif not torch.equal(ref, fr):
                print("FRAMES DIFFER", nstr, planar)
for (nstr, planar), v in sorted(res.items()):
    v = sorted(v)
    print(f"{nstr} in flight, planar {planar!s:5}: median {v[len(v) // 2]:7.1f} frames/s  (min {v[0]:7.1f}, max {v[-1]:7.1f})")
