"""How much would a per-8x8-quadrant cull save in the forward rasterizer?  For the walked head of every
tile list of one S-1M frame: fraction of (tile, splat) pairs that touch the tile at all (some pixel with
alpha >= 1/255), and the mean / max-over-quadrants number of splats per quadrant."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene  # noqa: E402

HEAD = int(sys.argv[1]) if len(sys.argv) > 1 else 256
scene = make_scene(1_000_000).to("cuda")
cam = make_camera().to("cuda")
with torch.no_grad():
    o = render_gaussians(scene, cam, return_intermediates=True)
m2, con, op = o["_means2d"][0], o["_conics"][0], o["_opacities"][0]
offs = o["_isect_offsets"].reshape(-1).long()
fids = o["_flatten_ids"].long()
T = offs.numel()
ends = torch.cat([offs[1:], torch.tensor([fids.numel()], device="cuda")])
tw = o["_isect_offsets"].shape[2]
tot_pairs = tot_tile = 0
sum_q = torch.zeros(4, device="cuda")
sum_max = 0.0
sum_tile_kept = 0.0
ys, xs = torch.meshgrid(torch.arange(16, device="cuda"), torch.arange(16, device="cuda"), indexing="ij")
quad = ((ys // 8) * 2 + (xs // 8)).reshape(-1)               # [256]
strip = (ys // 4).reshape(-1)
sum_strip_max = 0.0
CH = 64
for t0 in range(0, T, CH):
    tiles = torch.arange(t0, min(t0 + CH, T), device="cuda")
    n = (ends[tiles] - offs[tiles]).clamp(max=HEAD)                        # [B]
    idx = offs[tiles][:, None] + torch.arange(HEAD, device="cuda")[None, :]
    valid = torch.arange(HEAD, device="cuda")[None, :] < n[:, None]
    g = fids[idx.clamp(max=fids.numel() - 1)]
    px = ((tiles % tw) * 16)[:, None, None] + xs.reshape(-1)[None, None, :] + 0.5
    py = ((tiles // tw) * 16)[:, None, None] + ys.reshape(-1)[None, None, :] + 0.5
    dx = m2[g][..., 0:1] - px
    dy = m2[g][..., 1:2] - py
    c = con[g]
    sigma = 0.5 * (c[..., 0:1] * dx * dx + c[..., 2:3] * dy * dy) + c[..., 1:2] * dx * dy
    alpha = torch.clamp(op[g][..., None] * torch.exp(-sigma), max=0.999)
    hit = (sigma >= 0) & (alpha >= 1.0 / 255.0) & valid[..., None]          # [B, HEAD, 256]
    any_tile = hit.any(-1)
    tot_pairs += int(valid.sum())
    tot_tile += int(any_tile.sum())
    per_q = torch.stack([hit[..., quad == q].any(-1).sum(1) for q in range(4)], 1).float()   # [B,4]
    per_s = torch.stack([hit[..., strip == q].any(-1).sum(1) for q in range(4)], 1).float()
    sum_q += per_q.sum(0)
    sum_max += float(per_q.max(1).values.sum())
    sum_strip_max += float(per_s.max(1).values.sum())
    sum_tile_kept += float(any_tile.sum(1).float().sum())
print(f"walked head {HEAD}: pairs {tot_pairs}, touching the tile {tot_tile} ({tot_tile / tot_pairs:.3f})")
print(f"per tile: kept at tile level {sum_tile_kept / T:.1f}; mean per quadrant {float(sum_q.sum()) / 4 / T:.1f}; "
      f"max over the 4 quadrants {sum_max / T:.1f}; max over 4 strips (16x4) {sum_strip_max / T:.1f}")
print(f"blend iterations would shrink by {sum_tile_kept / sum_max:.2f}x (8x8 quadrants), {sum_tile_kept / sum_strip_max:.2f}x (16x4 strips)")
