"""Seeded synthetic scenes and cameras for benchmarks and parity tests.

"S-100k" / "S-1M" of SURVEY.md section 8(d) / BASELINE.md section 2: a pin-hole camera at the
origin looking down +z (OpenCV convention, like the w2c the reference passes at
``street_gaussian/models/street_gaussian_renderer.py:215``) over N random Gaussians.
Tensors are produced on the CPU from a ``torch.Generator`` so that the GPU box, this
container and the oracle all see bit-identical inputs.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

SEED = 20250404


@dataclass
class Scene:
    means: torch.Tensor       # f32[N,3]  world xyz            (pc.get_xyz)
    quats: torch.Tensor       # f32[N,4]  wxyz                 (pc.get_rotation)
    scales: torch.Tensor      # f32[N,3]  already exp()-ed     (pc.get_scaling)
    opacities: torch.Tensor   # f32[N,1]  already sigmoid()-ed (pc.get_opacity)
    sh: torch.Tensor          # f32[N,K,3]                     (pc.get_features)
    sh_degree: int

    def to(self, device):
        return Scene(self.means.to(device), self.quats.to(device), self.scales.to(device),
                     self.opacities.to(device), self.sh.to(device), self.sh_degree)

    @property
    def n(self):
        return self.means.shape[0]


@dataclass
class Camera:
    viewmat: torch.Tensor     # f32[4,4] row-major world->camera
    K: torch.Tensor           # f32[3,3]
    width: int
    height: int
    znear: float = 0.001      # street_gaussian/utils/camera_utils.py:47-48
    zfar: float = 1000.0
    camera_center: torch.Tensor = None   # f32[3], precomputed like the reference's Camera.camera_center

    def __post_init__(self):
        if self.camera_center is None:
            R = self.viewmat[:3, :3].double().cpu()
            t = self.viewmat[:3, 3].double().cpu()
            self.camera_center = (-(R.T @ t)).float().to(self.viewmat.device)

    def to(self, device):
        return Camera(self.viewmat.to(device), self.K.to(device), self.width, self.height,
                      self.znear, self.zfar, self.camera_center.to(device))


def make_camera(width=1920, height=1280, fx=2050.0, fy=2050.0, yaw=0.0, shift=(0.0, 0.0, 0.0)):
    """Identity view by default; `yaw` (radians, about +y) and `shift` give per-frame jitter."""
    c, s = math.cos(yaw), math.sin(yaw)
    R = torch.tensor([[c, 0.0, -s], [0.0, 1.0, 0.0], [s, 0.0, c]], dtype=torch.float64)
    t = -R @ torch.tensor(shift, dtype=torch.float64)
    V = torch.eye(4, dtype=torch.float64)
    V[:3, :3] = R
    V[:3, 3] = t
    K = torch.tensor([[fx, 0.0, width / 2.0], [0.0, fy, height / 2.0], [0.0, 0.0, 1.0]])
    return Camera(V.float(), K.float(), int(width), int(height))


def make_scene(n, sh_degree=1, seed=SEED, x_span=0.55, y_span=0.37, z_range=(2.0, 80.0),
               scale_range=(0.005, 0.15)):
    g = torch.Generator().manual_seed(seed)

    def U(lo, hi, *shape):
        return torch.rand(*shape, generator=g, dtype=torch.float32) * (hi - lo) + lo

    z = U(z_range[0], z_range[1], n)
    x = z * U(-x_span, x_span, n)
    y = z * U(-y_span, y_span, n)
    means = torch.stack([x, y, z], dim=-1)
    scales = torch.exp(U(math.log(scale_range[0]), math.log(scale_range[1]), n, 3))
    q = torch.randn(n, 4, generator=g, dtype=torch.float32)
    quats = q / q.norm(dim=-1, keepdim=True)
    opac = torch.sigmoid(torch.randn(n, 1, generator=g, dtype=torch.float32) * 1.5)
    K = (sh_degree + 1) ** 2
    sh = torch.randn(n, K, 3, generator=g, dtype=torch.float32)
    sh[:, 1:, :] *= 0.3
    return Scene(means.contiguous(), quats.contiguous(), scales.contiguous(),
                 opac.contiguous(), sh.contiguous(), sh_degree)


def make_scene_portable(n, sh_degree=1, seed=SEED, x_span=0.55, y_span=0.37, z_range=(2.0, 80.0),
                        scale_range=(0.005, 0.15)):
    """Same distributions as make_scene, but bit-identical on every host: torch's CPU exp / sigmoid /
    randn are vectorised differently per instruction set (AVX2 vs AVX-512 builds differ by an ulp, found
    when a digest made in one container met a scene regenerated on another CPU).  Here every random
    number is a PCG64 uniform double and the only transcendental, exp, is the scalar libm one; values
    are rounded to fp32 once at the end.  Used by fixtures that store digests instead of inputs."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    U = lambda lo, hi, *shape: rng.random(shape) * (hi - lo) + lo
    z = U(z_range[0], z_range[1], n)
    x = z * U(-x_span, x_span, n)
    y = z * U(-y_span, y_span, n)
    lo, hi = math.log(scale_range[0]), math.log(scale_range[1])
    scales = np.array([math.exp(v) for v in U(lo, hi, n * 3)]).reshape(n, 3)
    quats = U(-1.0, 1.0, n, 4)                                   # normalised by the kernels themselves
    quats[np.abs(quats).sum(axis=1) < 1e-3] = [1.0, 0.0, 0.0, 0.0]
    t = U(-4.5, 4.5, n)                                          # logit-uniform opacity
    opac = np.array([1.0 / (1.0 + math.exp(-v)) for v in t]).reshape(n, 1)
    K = (sh_degree + 1) ** 2
    sh = U(-1.7, 1.7, n, K, 3)
    sh[:, 1:, :] *= 0.3
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.float32)))
    return Scene(f(np.stack([x, y, z], axis=-1)), f(quats), f(scales), f(opac), f(sh), sh_degree)


def make_edge_case_scene(n=4096, seed=7):
    """Projection edge cases for the golden fixtures: behind the camera,
    nearer than znear, farther than zfar, enormous scales, needle-thin scales, off-screen,
    un-normalised and tiny quaternions, duplicates."""
    sc = make_scene(n, sh_degree=1, seed=seed, x_span=0.6, y_span=0.42, z_range=(0.5, 60.0))
    m = sc.means.clone()
    s = sc.scales.clone()
    q = sc.quats.clone()
    k = n // 16
    m[0 * k:1 * k, 2] *= -1.0                      # behind
    m[1 * k:2 * k, 2] = 0.0005                     # < znear
    m[2 * k:3 * k, 2] = 1500.0                     # > zfar
    s[3 * k:4 * k] *= 200.0                        # enormous
    s[4 * k:5 * k, 0] = 1e-7                       # needle
    s[5 * k:6 * k] = 1e-9                          # sub-pixel everything
    q[6 * k:7 * k] *= 37.5                         # un-normalised
    q[7 * k:8 * k] *= 1e-3
    m[8 * k:9 * k, 0] = m[8 * k:9 * k, 2] * 3.0    # far off-screen
    m[9 * k:10 * k] = m[9 * k]                     # duplicates (equal depth keys)
    s[9 * k:10 * k] = s[9 * k]
    q[9 * k:10 * k] = q[9 * k]
    m[10 * k:11 * k, 2] = 5.0                      # identical depths, different xy
    return Scene(m.contiguous(), q.contiguous(), s.contiguous(), sc.opacities, sc.sh, 1)


def make_street_scene(n, n_sky=None, sh_degree=1, seed=SEED, road_half_width=9.0, depth=160.0,
                      cam_height=1.6):
    """A street-SHAPED synthetic scene (not i.i.d. in the frustum like S-1M): what the reference renders
    (render.py:64-70 over Waymo scenes) has a ground plane, facades converging on a vanishing point, clutter
    along the kerbs, a dense far field squeezed into a few pixel rows at the horizon, and -- as a separate
    sub-model composited in a second pass (street_gaussian_renderer.py:80-93,136-163) -- sky Gaussians far
    outside the LiDAR sphere (gaussian_model_sky.py:69-76) whose screen radii reach hundreds of pixels.
    Camera convention as make_camera (OpenCV: +x right, +y DOWN, +z forward, camera at the origin).

    Returns (foreground Scene with n Gaussians, sky Scene with n_sky Gaussians; n_sky defaults to n // 32).
    Composition of the foreground: 34 % ground, 34 % facades, 14 % kerb clutter (cars / trees as blobs),
    12 % far field (150-450 m), 6 % large soft splats (canopies, haze).  Splat sizes grow with distance
    (~z^0.75: finer detail near the trajectory), as in a trained scene."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    n_sky = n // 32 if n_sky is None else int(n_sky)
    K = (sh_degree + 1) ** 2

    def U(lo, hi, *shape):
        return rng.random(shape) * (hi - lo) + lo

    def logU(lo, hi, *shape):
        return np.exp(U(math.log(lo), math.log(hi), *shape))

    parts = [int(n * f) for f in (0.34, 0.34, 0.14, 0.12)]
    parts.append(n - sum(parts))
    n_ground, n_facade, n_clutter, n_far, n_soft = parts
    means, scales = [], []
    # ground: uniform over the road plane -> projected density piles up at the horizon line
    x = U(-3.5 * road_half_width, 3.5 * road_half_width, n_ground)
    z = U(1.5, depth, n_ground)
    means.append(np.stack([x, np.full(n_ground, cam_height) + U(-0.03, 0.03, n_ground), z], -1))
    lod = lambda zz: (zz / 20.0) ** 0.75      # trained scenes carry finer splats near the camera trajectory
    scales.append(np.stack([logU(0.01, 0.12, n_ground), logU(0.002, 0.01, n_ground), logU(0.01, 0.12, n_ground)], -1)
                  * lod(z)[:, None])
    # facades: two walls, uniform over the wall area, thin across the wall
    side = np.where(rng.random(n_facade) < 0.5, -1.0, 1.0)
    x = side * (road_half_width + U(0.0, 1.5, n_facade))
    y = cam_height - U(0.0, 18.0, n_facade) ** 1.0
    z = U(2.0, depth, n_facade)
    means.append(np.stack([x, y, z], -1))
    scales.append(np.stack([logU(0.002, 0.01, n_facade), logU(0.01, 0.12, n_facade), logU(0.01, 0.12, n_facade)], -1)
                  * lod(z)[:, None])
    # kerb clutter: blobs (cars ~ 2 x 1.5 x 4.5 m, trees ~ 3 m balls at 4-8 m height)
    n_blobs = max(8, n_clutter // 4000)
    centre = np.stack([np.where(rng.random(n_blobs) < 0.5, -1.0, 1.0) * U(2.5, road_half_width - 0.5, n_blobs),
                       cam_height - U(0.4, 6.0, n_blobs), U(4.0, depth * 0.8, n_blobs)], -1)
    size = np.stack([U(0.8, 1.6, n_blobs), U(0.6, 1.6, n_blobs), U(1.0, 2.4, n_blobs)], -1)
    which = rng.integers(0, n_blobs, n_clutter)
    means.append(centre[which] + rng.standard_normal((n_clutter, 3)) * size[which] * 0.5)
    scales.append(logU(0.008, 0.1, n_clutter, 3) * lod(means[-1][:, 2].clip(2.0))[:, None])
    # far field: everything beyond the mapped street, squeezed into the rows around the horizon
    z = U(depth * 0.95, 450.0, n_far)
    x = z * U(-0.5, 0.5, n_far)
    y = cam_height - z * np.abs(rng.standard_normal(n_far)) * 0.03
    means.append(np.stack([x, y, z], -1))
    scales.append(logU(0.08, 1.2, n_far, 3))
    # large soft splats (canopies, haze)
    z = U(4.0, depth, n_soft)
    means.append(np.stack([z * U(-0.5, 0.5, n_soft), cam_height - U(2.0, 14.0, n_soft), z], -1))
    scales.append(logU(0.15, 1.2, n_soft, 3) * lod(z)[:, None])

    def finish(m, s, opacity_sigma, opacity_mean):
        cnt = m.shape[0]
        perm = rng.permutation(cnt)            # sub-models are not spatially sorted in a trained scene
        m, s = m[perm], s[perm]
        q = rng.standard_normal((cnt, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        opac = 1.0 / (1.0 + np.exp(-(rng.standard_normal((cnt, 1)) * opacity_sigma + opacity_mean)))
        sh = rng.standard_normal((cnt, K, 3))
        sh[:, 1:, :] *= 0.3
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.float32)))
        return Scene(f(m), f(q), f(s), f(opac), f(sh), sh_degree)

    fg = finish(np.concatenate(means), np.concatenate(scales), 1.5, 0.0)
    # sky shell: directions over the upper hemisphere in front of the camera, 2-5 LiDAR-sphere radii away
    R_lidar = 80.0
    d = rng.standard_normal((n_sky, 3))
    d[:, 0] *= 0.6
    d[:, 1] = -np.abs(d[:, 1]) * 0.35         # above the horizon (y is down), flattened towards it
    d[:, 2] = np.abs(d[:, 2]) * 0.3 + 1.0     # mostly in front of the camera
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dist = U(2.0 * R_lidar, 5.0 * R_lidar, n_sky, 1)
    sky = finish(d * dist, np.minimum(logU(1.0, 12.0, n_sky, 3), R_lidar), 1.0, 0.5)
    return fg, sky
