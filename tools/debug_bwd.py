"""Compares the two rasterize-backward kernels on the golden small scene and prints the worst rows."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from street_crafter_amd import _lib  # noqa: E402
import gsplat.rendering as R  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "pipeline_small.npz"))
dev = "cuda"
rng = np.random.default_rng(21)
N = g["means2d"].shape[0]
W, H = 128, 96
w_c = torch.from_numpy(rng.normal(size=(1, H, W, 4)).astype(np.float32)).to(dev)
w_a = torch.from_numpy(rng.normal(size=(1, H, W, 1)).astype(np.float32)).to(dev)


def run(variant):
    src = [torch.from_numpy(np.ascontiguousarray(a)).to(dev).requires_grad_(True)
           for a in (g["means2d"][None], g["conics"][None], g["colors"][None], g["opacities"][None])]
    rc, ra = R.rasterize_to_pixels(src[0], src[1], src[2], src[3], W, H, 16,
                                   torch.from_numpy(g["isect_offsets"]).to(dev),
                                   torch.from_numpy(g["flatten_ids"]).to(dev), absgrad=True)
    prev = _lib.set_option("raster_bwd", variant)
    try:
        ((rc * w_c).sum() + (ra * w_a).sum()).backward()
    finally:
        _lib.set_option("raster_bwd", prev)
    return [s.grad.cpu().numpy()[0] for s in src] + [src[0].absgrad.cpu().numpy()[0]]


a = run(0)
b = run(1)
names = ["means2d", "conics", "colors", "opacities", "absgrad"]
for n_, x, y in zip(names, a, b):
    d = np.abs(x - y).reshape(N, -1).max(axis=1)
    print(n_, "max abs diff", d.max(), "ref max", np.abs(x).max())
    for i in np.argsort(-d)[:2]:
        print("   row", i, "v0", x[i], "v1", y[i], "radius", g["radii"][i], "m2d", g["means2d"][i], "tpg", g["tiles_per_gauss"][i],
              "depth", g["depths"][i], "op", g["opacities"][i])
