"""Does dispatching the tiles in descending order of their TRUE walked length shorten the rasterizer?
Walked length per tile = max over its pixels of last_ids - first index (from a tracking run)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from street_crafter_amd import _lib  # noqa: E402
from street_crafter_amd.pipeline import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene, make_street_scene  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "s1m"
sc = (make_scene(1_000_000) if which == "s1m" else make_street_scene(1_000_000)[0]).to("cuda")
cam = make_camera().to("cuda")
lib = _lib.load()
W, H, tw, th = 1920, 1280, 120, 80
with torch.no_grad():
    o = render_gaussians(sc, cam, return_intermediates=True)
m2, con, col, op = o["_means2d"], o["_conics"], o["_colors"].contiguous(), o["_opacities"].contiguous()
off, fids = o["_isect_offsets"], o["_flatten_ids"]
N = op.shape[1]
rc = torch.empty(1, H, W, 4, device="cuda"); ra = torch.empty(1, H, W, 1, device="cuda")
last = torch.empty(1, H, W, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
work = torch.zeros(tw * th, dtype=torch.int32, device="cuda")       # what the kernel reports per tile


def launch(order=None, track=False):
    _lib.check(lib.sc_rasterize_fwd(m2.data_ptr(), con.data_ptr(), col.data_ptr(), op.data_ptr(), None, None, 1, N, 4, W, H,
                                    16, tw, th, off.data_ptr(), fids.data_ptr(), fids.numel(), rc.data_ptr(), ra.data_ptr(),
                                    last.data_ptr() if track else None, order.data_ptr() if order is not None else None,
                                    work.data_ptr(), None, 0, st), "fwd")


def timeit(order=None):
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); launch(order); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2] * 1e3


launch(track=True)
torch.cuda.synchronize()
ref = rc.clone()
walked = (last.view(th, 16, tw, 16).permute(0, 2, 1, 3).reshape(th * tw, 256).max(dim=1).values - off.view(-1)).clamp(min=0)
print(which, "walked length per tile: mean %.0f  p50 %.0f  p99 %.0f  max %d; list length mean %.0f" % (
    walked.float().mean(), walked.float().median(), walked.float().quantile(0.99), int(walked.max()),
    fids.numel() / (tw * th)))
reported = work.clone()
print("  kernel-reported work vs true walked length: corr %.4f" % float(torch.corrcoef(torch.stack([reported.float(), walked.float()]))[0, 1]))
by_report = torch.argsort(reported, descending=True).to(torch.int32)
ident = torch.arange(tw * th, dtype=torch.int32, device="cuda")
lpt = torch.argsort(walked, descending=True).to(torch.int32)
rnd = torch.randperm(tw * th, device="cuda").to(torch.int32)
# heavy first, but otherwise in tile order: only the top 5 % leave their place
k = tw * th // 20
top = lpt[:k]
rest_mask = torch.ones(tw * th, dtype=torch.bool, device="cuda"); rest_mask[top.long()] = False
top_first = torch.cat([top, ident[rest_mask]])
for name, order in (("no order array (tile = block)", None), ("identity order array", ident), ("random order", rnd),
                    ("descending true walked length (LPT)", lpt), ("descending kernel-reported work (128-entry steps)", by_report),
                    ("top 5 % first, rest in tile order", top_first)):
    t = timeit(order)
    launch(order); torch.cuda.synchronize()
    print(f"  {name:45s} {t:7.1f} us  identical={bool(torch.equal(rc, ref))}", flush=True)
