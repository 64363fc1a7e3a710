import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torch.profiler import profile, ProfilerActivity
from harness.caller import render_gaussians
from street_crafter_amd.dist import to_uint8_frame
from street_crafter_amd.scenes import make_scene
import bench
dev = "cuda"
sc = make_scene(1_000_000).to(dev)
cams = [bench.frame_camera(s, 1920, 1280).to(dev) for s in range(8)]
out = torch.empty(1280, 1920, 3, dtype=torch.uint8, device=dev)
def frame(s):
    with torch.no_grad():
        to_uint8_frame(render_gaussians(sc, cams[s])["rgb"], out=out)
for s in range(5): frame(s)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    frame(6)
    torch.cuda.synchronize()
evs = [e for e in prof.events()]
for e in evs:
    n = e.name
    if "Memcpy" in n or "copy" in n.lower() or "hipMemcpy" in n:
        print(n, getattr(e, "device_time_total", None), [s for s in (e.stack or [])][:6])
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40))
