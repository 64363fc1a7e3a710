"""cProfile of the host side of the training step (render in train mode + L1 loss + backward), S-1M at 1600x1066:
which part of the ~0.9 ms of host time per step is this library's (operator wrappers, autograd.Function plumbing) and
which is torch autograd over the reference caller's glue and the loss."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene  # noqa: E402

if os.environ.get("SC_NATIVE_AUTOGRAD") == "0":      # A/B: the Python torch.autograd.Functions instead of the C++ ones
    from street_crafter_amd import rendering as _r
    _r.set_native_autograd(False)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W, H = 1600, 1066
sc = make_scene(N).to("cuda")
cam = make_camera(W, H, 2050.0 * W / 1920.0, 2050.0 * W / 1920.0).to("cuda")
ps = (sc.means, sc.quats, sc.scales, sc.opacities, sc.sh)
for t in ps:
    t.requires_grad_(True)
target = torch.rand(3, H, W, device="cuda")
acc = {"fwd": 0.0, "loss": 0.0, "bwd": 0.0}


def step():
    for t in ps:
        t.grad = None
    t0 = time.perf_counter()
    out = render_gaussians(sc, cam, mode="train")
    t1 = time.perf_counter()
    loss = (out["rgb"] - target).abs().mean() + 0.01 * out["acc"].mean()
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    acc["fwd"] += t1 - t0; acc["loss"] += t2 - t1; acc["bwd"] += t3 - t2


for _ in range(5):
    step()
torch.cuda.synchronize()
for k in acc:
    acc[k] = 0.0
t0 = time.perf_counter()
for _ in range(50):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / 50:.3f} ms/step, wall {1e3 * (t2 - t0) / 50:.3f} ms/step; host time by phase (us/step): "
      + ", ".join(f"{k} {v / 50 * 1e6:.0f}" for k, v in acc.items()), flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("cumtime").print_stats(28)
