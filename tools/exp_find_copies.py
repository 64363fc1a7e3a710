"""Which torch calls of a caller-sequence frame end in a device copy (rocprof lists 3 __amd_rocclr_copyBuffer per frame)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from harness.caller import render_gaussians
from street_crafter_amd.dist import to_uint8_frame
from street_crafter_amd.scenes import make_scene
dev = torch.device("cuda", 0)
scene = make_scene(1_000_000).to(dev)
cams = [bench.frame_camera(s, 1920, 1280).to(dev) for s in range(8)]
out = torch.empty((1280, 1920, 3), dtype=torch.uint8, device=dev)
with torch.no_grad():
    for s in range(4):
        to_uint8_frame(render_gaussians(scene, cams[s])["rgb"], out=out)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        for s in range(4, 6):
            to_uint8_frame(render_gaussians(scene, cams[s])["rgb"], out=out)
        torch.cuda.synchronize()
for e in prof.events():
    if "copy" in e.name.lower() or "Memcpy" in e.name or "contiguous" in e.name or "clone" in e.name:
        st = [f for f in (e.stack or []) if "site-packages" not in f][:3]
        print(e.name, e.input_shapes, "cuda_us=%.1f" % (e.device_time if hasattr(e, "device_time") else 0), st)
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
