"""Randomised check of simple_knn.distCUDA2 (mean squared distance to the 3 nearest neighbours): bit-exact against
the oracle (oracle/knn_oracle.py) for clouds up to 20 k points, against scipy's exact k-NN (float64) beyond, over
degenerate shapes: duplicates, a plane, a line, tight clusters far apart, 1..5 points, huge coordinates.
Test infrastructure (it uses oracle/): lives under tests/.  Usage: python tests/fuzz/fuzz_knn.py [seed] [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from scipy.spatial import cKDTree  # noqa: E402

from oracle import knn_oracle as KO  # noqa: E402
from simple_knn._C import distCUDA2  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bad = 0
for it in range(ROUNDS):
    n = int(rng.choice([1, 2, 3, 4, 5, 37, 1000, 20_000, 150_000]))
    shape = str(rng.choice(["uniform", "plane", "line", "duplicates", "clusters", "huge", "anisotropic"]))
    p = rng.uniform(-10, 10, size=(n, 3))
    if shape == "plane":
        p[:, 2] = 1.5
    elif shape == "line":
        p[:, 1:] = p[:, :1] * np.array([0.5, -2.0])
    elif shape == "duplicates":
        p = np.round(p)                                   # many coincident points
    elif shape == "clusters":
        c = rng.uniform(-1e3, 1e3, size=(8, 3))
        p = c[rng.integers(0, 8, size=n)] + rng.normal(size=(n, 3)) * 1e-3
    elif shape == "huge":
        p *= 1e5
    elif shape == "anisotropic":
        p *= np.array([50.0, 0.01, 1.0])
    pts = p.astype(np.float32)
    got = distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()
    if n <= 20_000:
        ref = KO.dist_cuda2(pts)
        ok = np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        how = "bit-exact vs oracle"
    else:
        d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4, workers=-1)
        ref = (d[:, 1:] ** 2).mean(axis=1)
        ok = bool(np.allclose(got, ref, rtol=3e-5, atol=1e-9 * max(1.0, float(np.abs(pts).max()) ** 2)))
        how = "vs cKDTree"
    print(f"[{it}] n={n} {shape}: {how}: {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
print("FAILED" if bad else "distCUDA2 agrees")
