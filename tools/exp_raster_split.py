"""Heavy tiles shared by two waves (16 x 8 halves): dispatch lists built on the host from the work the kernel
reported for THIS frame, swept over the split threshold (fraction of the heaviest tile's work).
Usage: python tools/exp_raster_split.py [s1m|street|sky]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from street_crafter_amd import _lib  # noqa: E402
from street_crafter_amd import rendering  # noqa: E402
from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene, make_street_scene  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "s1m"
if which == "s1m":
    sc = make_scene(1_000_000)
else:
    fg, sky = make_street_scene(1_000_000)
    sc = fg if which == "street" else sky
sc = sc.to("cuda")
cam = make_camera().to("cuda")
lib = _lib.load()
W, H, tw, th = 1920, 1280, 120, 80
T = tw * th
rendering.set_tile_order(False)
with torch.no_grad():
    o = render_gaussians(sc, cam, return_intermediates=True)
m2, con, col, op = o["_means2d"], o["_conics"], o["_colors"].contiguous(), o["_opacities"].contiguous()
off, fids = o["_isect_offsets"], o["_flatten_ids"]
N = op.shape[1]
rc = torch.empty(1, H, W, 4, device="cuda"); ra = torch.empty(1, H, W, 1, device="cuda")
st = torch.cuda.current_stream().cuda_stream
work = torch.zeros(T, dtype=torch.int32, device="cuda")
scratch_work = torch.zeros(T, dtype=torch.int32, device="cuda")
L = lib.sc_tile_order_len(T)          # the whole buffer; the forward's list is its first L_FWD items
L_FWD = T + T // 8 + 8


def launch(order=None, wk=scratch_work):
    _lib.check(lib.sc_rasterize_fwd(m2.data_ptr(), con.data_ptr(), col.data_ptr(), op.data_ptr(), None, None, 1, N, 4, W, H,
                                    16, tw, th, off.data_ptr(), fids.data_ptr(), fids.numel(), rc.data_ptr(), ra.data_ptr(),
                                    None, order.data_ptr() if order is not None else None, wk.data_ptr(), st), "fwd")


def timeit(order=None):
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); launch(order); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2] * 1e3


def items(wk, thr, cap):
    """tiles with work >= thr * max are listed as two halves, at most `cap` of them; the halves are adjacent and keep
    their tile's place in the heaviest-first order (what the order job of isect_bin.hip builds)"""
    wk = wk.to(torch.float32)
    order = torch.argsort(wk, descending=True, stable=True)
    ws = wk[order]
    parts = torch.ones(T, dtype=torch.int64, device="cuda")
    parts[ws >= thr * wk.max()] = 2
    extra = torch.cumsum(parts - 1, 0)
    parts[extra > cap] = 1                                  # out of room: the lighter ones stay whole
    rep = torch.repeat_interleave(order, parts)
    first = torch.cumsum(parts, 0) - parts
    sub = torch.arange(rep.numel(), device="cuda") - torch.repeat_interleave(first, parts)
    np_ = torch.repeat_interleave(parts, parts)
    kind = torch.where(np_ == 1, 0, 1 + sub)
    it = ((rep << 2) | kind).to(torch.int32)
    out = torch.full((L,), -1, dtype=torch.int32, device="cuda")
    out[: it.numel()] = it
    return out, int((parts == 2).sum())


def items_sorted(wk, thr, cap, half_weight):
    """round-2 first try: whole tiles and halves sorted by weight; the two halves of a tile are NOT adjacent"""
    wk = wk.to(torch.float32)
    split = wk >= thr * wk.max()
    if int(split.sum()) > cap:
        idx = torch.argsort(wk, descending=True)[:cap]
        split = torch.zeros_like(split); split[idx] = True
    t = torch.arange(T, device="cuda", dtype=torch.int32)
    it = torch.cat([t[~split] << 2, (t[split] << 2) | 1, (t[split] << 2) | 2])
    wt = torch.cat([wk[~split], wk[split] * half_weight, wk[split] * half_weight])
    it = it[torch.argsort(wt, descending=True, stable=True)]
    out = torch.full((L,), -1, dtype=torch.int32, device="cuda")
    out[: it.numel()] = it
    return out


launch(None, work)
torch.cuda.synchronize()
ref = rc.clone()
w = work.clone()
print(which, "work per tile: mean %.0f p50 %.0f p99 %.0f max %d; I=%d" % (w.float().mean(), w.float().median(),
      w.float().quantile(0.99), int(w.max()), fids.numel()))
cap = L_FWD - T
nlist = torch.diff(torch.cat([off.view(-1), torch.tensor([fids.numel()], dtype=torch.int32, device="cuda")]))
print("  list length vs work: corr %.3f" % float(torch.corrcoef(torch.stack([nlist.float(), w.float()]))[0, 1]))
cands = [("no list (tile = block)", None), ("heaviest first, whole tiles", items(w, 2.0, cap)[0]),
         ("longest list first, whole tiles", items(nlist, 2.0, cap)[0])]
for thr in (0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3):
    o, n = items(w, thr, cap)
    cands.append((f"halves >= {thr:.1f} x max ({n} tiles), adjacent", o))
cands.append(("halves >= 0.5 x max, sorted apart, half weight 1.0", items_sorted(w, 0.5, cap, 1.0)))
cands.append(("halves >= 0.5 x max, sorted apart, half weight 0.65", items_sorted(w, 0.5, cap, 0.65)))
def tail_halves(k):
    """tile order, the LAST k tiles as halves (no hint needed: smaller items at the end of the launch)"""
    t = torch.arange(T, device="cuda", dtype=torch.int32)
    head, tl = t[: T - k] << 2, t[T - k:]
    it = torch.cat([head, torch.stack([(tl << 2) | 1, (tl << 2) | 2], 1).reshape(-1)])
    out = torch.full((L,), -1, dtype=torch.int32, device="cuda")
    out[: it.numel()] = it
    return out


for k in (300, 600, 1200):
    cands.append((f"tile order, last {k} tiles as halves", tail_halves(k)))


def lpt_tail_halves(wk, k, noise):
    """heaviest first by a NOISY hint (what a moving camera leaves of it), the last k items' tiles as halves"""
    g = torch.Generator(device="cuda").manual_seed(1)
    wn = wk.float() * (1.0 + noise * torch.randn(T, device="cuda", generator=g))
    order = torch.argsort(wn, descending=True, stable=True).to(torch.int32)
    head, tl = order[: T - k] << 2, order[T - k:]
    it = torch.cat([head, torch.stack([(tl << 2) | 1, (tl << 2) | 2], 1).reshape(-1)]) if k else order << 2
    out = torch.full((L,), -1, dtype=torch.int32, device="cuda")
    out[: it.numel()] = it
    return out


for noise in (0.0, 0.3, 1.0):
    for k in (0, 1200):
        cands.append((f"heaviest first, hint noise {noise:.1f}, last {k} as halves", lpt_tail_halves(w, k, noise)))
res = {n: [] for n, _ in cands}
for rnd in range(4):                      # round 0 = warm-up (clocks), not shown
    for n, o in cands:
        res[n].append(timeit(o))
for n, o in cands:
    rc.zero_(); launch(o); torch.cuda.synchronize()
    print(f"  {n:55s} " + " ".join(f"{t:7.1f}" for t in res[n][1:]) + f" us  identical={bool(torch.equal(rc, ref))}")
