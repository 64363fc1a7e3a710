"""CPU oracle for ``simple_knn._C.distCUDA2`` -- TEST INFRASTRUCTURE ONLY.

Reference call sites: ``street_gaussian/models/gaussian_model.py:65``,
``gaussian_model_actor.py:139``, ``data_processor/utils/render_utils.py:125``.
The algorithm lives in the un-vendored submodule ``submodules/simple-knn``
(https://gitlab.inria.fr/bkerbl/simple-knn.git, ``.gitmodules:4-6``; the directory is
empty and no SHA is recorded), so this restates its published behaviour
(SURVEY.md A.7): out[i] = mean of the squared Euclidean distances from point i to
its 3 nearest OTHER points (by position in the array: exact duplicates count as
distance 0); with fewer than 4 points the missing neighbours contribute FLT_MAX.

PARITY STATUS: the reference holds no fixture for this op ("parity unpinned" w.r.t.
reference outputs); the restatement is pinned against ``scipy.spatial.cKDTree``
(an independent exact k-NN) in ``tests/test_oracle_cpu.py``.

Normative fp32 op order: d2 = (dx*dx + dy*dy) + dz*dz, one rounding per op, no FMA;
out = ((b0 + b1) + b2) / 3 with b0 <= b1 <= b2.
Only tests/, smoke() and bench.py's cpu_baseline may import this.
"""
from __future__ import annotations

import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)


def dist_cuda2(points, block=2048):
    p = np.ascontiguousarray(points, dtype=np.float32)
    n = p.shape[0]
    out = np.empty(n, dtype=np.float32)
    if n == 0:
        return out
    for s in range(0, n, block):
        q = p[s:s + block]
        dx = q[:, None, 0] - p[None, :, 0]
        dy = q[:, None, 1] - p[None, :, 1]
        dz = q[:, None, 2] - p[None, :, 2]
        d2 = (dx * dx + dy * dy) + dz * dz
        d2[np.arange(q.shape[0]), np.arange(s, s + q.shape[0])] = np.inf  # self by index
        k = min(3, n - 1)
        if k > 0:
            part = np.partition(d2, k - 1, axis=1)[:, :k]
            part.sort(axis=1)
        else:
            part = np.empty((q.shape[0], 0), dtype=np.float32)
        best = np.full((q.shape[0], 3), FLT_MAX, dtype=np.float32)
        best[:, :k] = part
        with np.errstate(over="ignore"):
            out[s:s + block] = ((best[:, 0] + best[:, 1]) + best[:, 2]) / np.float32(3.0)
    return out
