"""distCUDA2 timing on uniform and clustered clouds."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from simple_knn._C import distCUDA2  # noqa: E402

for n in (100_000, 1_000_000, 4_000_000):
    g = torch.Generator().manual_seed(0)
    uni = (torch.rand(n, 3, generator=g) * 100).cuda()
    clu = (torch.randn(n, 3, generator=g) * torch.tensor([20.0, 1.0, 8.0])).cuda()
    for name, p in (("uniform", uni), ("clustered", clu)):
        distCUDA2(p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            distCUDA2(p)
        torch.cuda.synchronize()
        print(f"N={n} {name}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms", flush=True)
