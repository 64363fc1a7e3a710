import torch
H, W = 1280, 1920
dev = "cuda"
alpha = torch.rand(1, H, W, 1, device=dev)
inter = torch.rand(1, H, W, 4, device=dev)
planar = torch.rand(1, 4, H, W, device=dev).permute(0, 2, 3, 1)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, x in (("interleaved", inter), ("planar", planar)):
    d = t(lambda: x[..., -1:] / alpha.clamp(min=1e-10))
    c = t(lambda: torch.clamp(x[..., :-1], 0.0, 1.0))
    y = torch.clamp(x[..., :-1], 0.0, 1.0)
    print(name, "depth div + clamp(alpha) %.1f us, colour clamp %.1f us" % (d, c), "clamp out strides", y.stride(), "rgb chw strides", y[0].permute(2, 0, 1).stride())
