"""BASELINE config 0: one Waymo-sized frame (1920x1280, +-10 frames of synthetic LiDAR, a few tracked
actors) through the CPU / numpy condition render (street_crafter_amd/lidar_condition.py).  Prints
timings; no GPU.

    python tools/lidar_condition_demo.py [points_per_frame] [out.png]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from street_crafter_amd import lidar_condition as lc  # noqa: E402

PPF = int(sys.argv[1]) if len(sys.argv) > 1 else 150_000
rng = np.random.default_rng(20250404)
F, H, W = 21, 1280, 1920
ego = []
for f in range(F):
    p = np.eye(4)
    p[:3, 3] = [1.5 * f, 0.0, 0.0]
    ego.append(p)
# LiDAR-like background: ground plane + facades on both sides, in world coordinates
def cloud(f):
    n_g = PPF // 2
    g = np.stack([rng.uniform(-20, 75, n_g) + ego[f][0, 3], rng.uniform(-15, 15, n_g), rng.normal(-1.8, 0.02, n_g)], 1)
    n_w = PPF - n_g
    side = rng.choice([-12.0, 12.0], n_w)
    wl = np.stack([rng.uniform(-20, 75, n_w) + ego[f][0, 3], side + rng.normal(0, 0.05, n_w), rng.uniform(-1.8, 8, n_w)], 1)
    xyz = np.concatenate([g, wl])
    rgb = np.clip(0.5 + 0.1 * rng.normal(size=(PPF, 3)) + 0.3 * np.sin(xyz[:, :1] * 0.3), 0, 1)
    return np.concatenate([xyz, rgb], 1)
ply = {"background": {f: cloud(f) for f in range(F)}}
for k in range(4):
    ply[f"veh_{k}"] = {f: np.concatenate([rng.uniform(-1, 1, (800, 3)) * [2.3, 1.0, 0.8], np.tile(rng.uniform(0, 1, 3), (800, 1))], 1)
                       for f in range(F)}
track = {f"veh_{k}": {"camera_box": None, "lidar_box": {"heading": 0.1 * k, "center_x": 12.0 + 9 * k,
                                                         "center_y": -3.0 + 2.0 * k, "center_z": -1.0}} for k in range(4)}
ext = np.eye(4)
ext[:3, :3] = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], float)
ext[:3, 3] = [1.5, 0.0, 0.3]
ixt = np.array([[2050.0, 0, 960.0], [0, 2050.0, 640.0], [0, 0, 1.0]])
frame = 10
t0 = time.perf_counter()
cloud_w = lc.assemble_frame(ply, track, ego[frame], frame, F, delta_frames=10)
c2w = lc.shifted_camera(ego[frame], ego, frame, ext)
t1 = time.perf_counter()
xyz, feat = lc.filter_visible(cloud_w[:, :3], cloud_w[:, 3:], c2w, ixt, H, W)
t2 = time.perf_counter()
img = lc.render_points(c2w, ixt, xyz, feat, H, W, use_ndc_scale=True, scale=0.01)
t3 = time.perf_counter()
print(f"aggregated points {cloud_w.shape[0]}, visible {xyz.shape[0]}, covered pixels {(img[0, ..., 3] > 0).mean():.3f}")
print(f"assemble {t1 - t0:.2f} s, filter {t2 - t1:.2f} s, point render {t3 - t2:.2f} s, total {t3 - t0:.2f} s "
      f"(numpy, 1 thread, host cpus {os.cpu_count()})")
if len(sys.argv) > 2:
    import struct, zlib
    rgb8 = (img[0, ..., :3] * 255).astype(np.uint8)
    raw = b"".join(b"\x00" + rgb8[y].tobytes() for y in range(H))
    def chunk(t, d): return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    open(sys.argv[2], "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0)) +
                                  chunk(b"IDAT", zlib.compress(raw, 3)) + chunk(b"IEND", b""))
