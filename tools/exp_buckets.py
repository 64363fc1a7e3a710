"""Distribution of the 2x2 super-tile bucket sizes (records) of a scene, and what big_split has to cut.
Usage: python tools/exp_buckets.py [street1m|street3m|s1m]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene, make_street_scene  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "street1m"
sc = {"s1m": lambda: make_scene(1_000_000), "street1m": lambda: make_street_scene(1_000_000)[0],
      "street3m": lambda: make_street_scene(3_000_000)[0]}[which]().to("cuda")
cam = make_camera().to("cuda")
with torch.no_grad():
    o = render_gaussians(sc, cam, return_intermediates=True)
m2, r = o["_means2d"][0], o["_radii"][0]
vis = r > 0
m2, r = m2[vis], r[vis].float()
tw, th = 120, 80
x0 = ((m2[:, 0] - r) / 16).floor().clamp(0, tw).long(); x1 = ((m2[:, 0] + r) / 16).ceil().clamp(0, tw).long()
y0 = ((m2[:, 1] - r) / 16).floor().clamp(0, th).long(); y1 = ((m2[:, 1] + r) / 16).ceil().clamp(0, th).long()
ok = (x1 > x0) & (y1 > y0)
x0, x1, y0, y1 = x0[ok], x1[ok], y0[ok], y1[ok]
sx0, sx1, sy0, sy1 = x0 // 2, (x1 - 1) // 2, y0 // 2, (y1 - 1) // 2
gw, gh = tw // 2, th // 2
grid = torch.zeros(gh + 1, gw + 1, dtype=torch.long, device="cuda")
grid.index_put_((sy0, sx0), torch.ones_like(sx0), accumulate=True)
grid.index_put_((sy0, sx1 + 1), -torch.ones_like(sx0), accumulate=True)
grid.index_put_((sy1 + 1, sx0), -torch.ones_like(sx0), accumulate=True)
grid.index_put_((sy1 + 1, sx1 + 1), torch.ones_like(sx0), accumulate=True)
cnt = grid.cumsum(0).cumsum(1)[:gh, :gw].reshape(-1)
print(which, "records", int(cnt.sum()), "buckets", cnt.numel(), "max", int(cnt.max()))
for lo, hi in ((0, 3584), (3584, 8192), (8192, 16384), (16384, 32768), (32768, 65536), (65536, 1 << 30)):
    m = (cnt > lo) & (cnt <= hi)
    print(f"  ({lo:6d}, {hi:10d}]: {int(m.sum()):5d} buckets, {int(cnt[m].sum()):10d} records")
