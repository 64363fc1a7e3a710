"""Which autograd nodes of the train step launch fill / copy / strided kernels (the step has 7 FillFunctor launches, 42 us)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from harness.caller import render_gaussians
from street_crafter_amd.scenes import make_camera, make_scene
dev = "cuda"
W, H = 1600, 1066
scene = make_scene(1_000_000).to(dev)
cam = make_camera(W, H, 2050.0 * W / 1920.0, 2050.0 * W / 1920.0).to(dev)
ps = (scene.means, scene.quats, scene.scales, scene.opacities, scene.sh)
for t in ps:
    t.requires_grad_(True)
target = torch.rand(3, H, W, device=dev)
def step():
    for t in ps:
        t.grad = None
    out = render_gaussians(scene, cam, mode="train")
    loss = (out["rgb"] - target).abs().mean() + 0.01 * out["acc"].mean()
    loss.backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=70, max_name_column_width=48,
                                                          max_shapes_column_width=60))
