"""LiDAR -> image condition render, CPU / numpy restatement (BASELINE config 0).

PARITY UNPINNED.  This is the plumbing of data_processor/waymo_processor/waymo_render_lidar_pcd.py
(:104-148 aggregation, :199-277 frame assembly) and of the call-site contract of
data_processor/utils/render_utils.py:83-183 (`render_pointcloud_diff_point_rasterization`), restated
in numpy.  The rasterizer underneath the reference's call, `diff_point_rasterization.PointRasterizer`,
is a third-party CUDA extension (requirements.txt:39) that is not in /root/reference and is not
installed; the reference holds no output image or test for this path.  What IS contractual and is
reproduced here: the frame assembly (which points, in which frame, under which poses), the visibility
filter, the point radius rule (`use_ndc_scale`, render_utils.py:116-122: a constant SCREEN-space radius
of scale * 0.5 * min(H, W) pixels; `use_knn_scale`, :123-127: world radius from the local point density
through simple_knn.distCUDA2 -- the LiDAR path's one consumer of SURVEY row a14 -- capped by `scale`),
`max_hit = 10` front-most hits per pixel (:160), opacity `occ`,
black background, and the output layout `[1, H, W, 4]` = rgb + accumulated alpha (:179-183).  The
splat footprint is taken as a hard disc with alpha = occ; the extension's exact footprint is unknown.
No GPU is involved (BASELINE config 0: "CPU/numpy (plumbing, no GPU)") except for `use_knn_scale` without a
precomputed `knn_dist2`, which calls the HIP distCUDA2.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np


# ---- frame assembly (waymo_render_lidar_pcd.py) ---------------------------------------------------
def make_lidar_ply(ply_dict: Dict[str, Dict[int, np.ndarray]], start_frame: int, end_frame: int
                   ) -> Dict[str, np.ndarray]:
    """Concatenates the per-frame [n, xyz+rgb] clouds of frames start..end (inclusive) per track
    (:104-129).  Background has every frame; an actor only the frames it was seen in."""
    out = {"background": np.concatenate([ply_dict["background"][f] for f in range(start_frame, end_frame + 1)], axis=0)}
    for track_id, frames in ply_dict.items():
        if track_id == "background":
            continue
        parts = [frames[f] for f in range(start_frame, end_frame + 1) if f in frames]
        if parts:
            out[track_id] = np.concatenate(parts, axis=0)
    return out


def transform_lidar_ply(lidar_ply: np.ndarray, pose: np.ndarray) -> np.ndarray:
    """(:131-136) xyz through a 4x4 pose, colours untouched."""
    xyz, rgb = lidar_ply[..., :3], lidar_ply[..., 3:]
    xyz_homo = np.concatenate([xyz, np.ones_like(xyz[..., :1])], axis=-1)
    return np.concatenate([(xyz_homo @ pose.T)[..., :3], rgb], axis=-1)


def box_pose(box: dict) -> np.ndarray:
    """Object-to-vehicle pose of a tracked box: yaw `heading` about +z, then the box centre (:226-232)."""
    c, s = np.cos(box["heading"]), np.sin(box["heading"])
    pose = np.eye(4)
    pose[:3, :3] = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    pose[:3, 3] = np.array([box["center_x"], box["center_y"], box["center_z"]])
    return pose


def lane_shift_direction(ego_frame_poses, frame: int) -> np.ndarray:
    """Unit vector to the right of the direction of travel in the ground plane
    (waymo_helpers.py:272-282)."""
    assert 0 <= frame < len(ego_frame_poses)
    a, b = (0, 1) if frame == 0 else (frame - 1, frame)
    d = (ego_frame_poses[b][:3, 3] - ego_frame_poses[a][:3, 3])[:2].astype(np.float64)
    d = d / np.linalg.norm(d)
    return np.array([d[1], -d[0], 0.0])


def assemble_frame(ply_dict, track_info_frame: Dict[str, dict], ego_pose: np.ndarray, frame: int,
                   num_frames: int, delta_frames: int = 10, shift: float = 0.0) -> np.ndarray:
    """Step 1 of render_one (:207-236): background of frames frame +- delta, plus every actor tracked in
    THIS frame, its aggregated points moved by ego_pose @ box pose (camera box when present and the view
    is not shifted, LiDAR box otherwise).  -> [N, 6] xyz + rgb in world coordinates."""
    start, end = max(0, frame - delta_frames), min(num_frames - 1, frame + delta_frames)
    per_track = make_lidar_ply(ply_dict, start, end)
    parts = [per_track.pop("background")]
    for track_id, cloud in per_track.items():
        if track_id not in track_info_frame:
            continue
        info = track_info_frame[track_id]
        if shift == 0:
            box = info["camera_box"] if info.get("camera_box") is not None else info["lidar_box"]
        else:
            box = info["lidar_box"]
        parts.append(transform_lidar_ply(cloud, ego_pose @ box_pose(box)))
    return np.concatenate(parts, axis=0)


def shifted_camera(ego_pose: np.ndarray, ego_frame_poses, frame: int, extrinsic: np.ndarray,
                   shift: float = 0.0, lane_shift_sign: float = 1.0) -> np.ndarray:
    """Step 2 (:239-245): the ego pose pushed sideways by `shift` metres, times the camera extrinsic -> c2w."""
    pose = ego_pose.copy()
    pose[:3, 3] += lane_shift_sign * lane_shift_direction(ego_frame_poses, frame) * shift
    return pose @ extrinsic


def filter_visible(ply_xyz: np.ndarray, ply_rgb: np.ndarray, c2w: np.ndarray, ixt: np.ndarray, h: int, w: int
                   ) -> Tuple[np.ndarray, np.ndarray]:
    """Step 3 (:250-267): keeps points in front of the camera whose projection falls inside the image;
    features = rgb, depth, 1."""
    w2c = np.linalg.inv(c2w)
    cam = np.dot(ply_xyz, w2c[:3, :3].T) + w2c[:3, 3:].T
    depth = cam[:, 2]
    pix = np.dot(cam, ixt.T)
    with np.errstate(divide="ignore", invalid="ignore"):
        pix = pix[:, :2] / pix[:, 2:]
    valid = (depth > 1e-3) & (pix[:, 0] >= 0) & (pix[:, 0] < w) & (pix[:, 1] >= 0) & (pix[:, 1] < h)
    feat = np.concatenate([ply_rgb[valid], depth[valid, None], np.ones((int(valid.sum()), 1))], axis=-1)
    return ply_xyz[valid], feat


# ---- the point render (render_utils.py:83-183 call-site contract) ---------------------------------
def knn_point_radii(points: np.ndarray, scale: float, knn_scale_down: float = 1.0,
                    knn_dist2: Optional[np.ndarray] = None) -> np.ndarray:
    """The `use_knn_scale` branch of render_pointcloud_diff_point_rasterization (render_utils.py:123-127), the
    LiDAR path's one consumer of simple_knn:  `dist2 = clamp_min(distCUDA2(xyz), 1e-7)`;
    `radius = minimum(sqrt(dist2) * knn_scale_down, scale)` -- a point's world radius follows the local point
    density (mean squared distance to its 3 nearest neighbours) and is capped by `scale`.
    `knn_dist2`: that mean squared distance per point, when the caller already has it; otherwise it is computed
    by the product's own `simple_knn._C.distCUDA2` (HIP: needs a GPU -- there is no CPU k-NN in the product)."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    if knn_dist2 is None:
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("use_knn_scale needs simple_knn._C.distCUDA2 (a HIP device), or pass knn_dist2=")
        from simple_knn._C import distCUDA2
        knn_dist2 = distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()
    d2 = np.maximum(np.asarray(knn_dist2, dtype=np.float32).reshape(-1), np.float32(0.0000001))
    if d2.shape[0] != pts.shape[0]:
        raise ValueError(f"knn_dist2 has {d2.shape[0]} entries for {pts.shape[0]} points")
    return np.minimum(np.sqrt(d2) * np.float32(knn_scale_down), np.float32(scale)).astype(np.float64)


def render_points(c2w: np.ndarray, ixt: np.ndarray, points: np.ndarray, features: np.ndarray, H: int, W: int,
                  occ: float = 1.0, scale: float = 0.035, use_ndc_scale: bool = False, max_hit: int = 10,
                  near: float = 1.0, far: float = 100.0, use_knn_scale: bool = False,
                  knn_scale_down: float = 1.0, knn_dist2: Optional[np.ndarray] = None) -> np.ndarray:
    """-> float32 [1, H, W, 4]: rgb composited front to back over a black background, and alpha.

    Radius rule: world radius `scale`; with `use_ndc_scale` the world radius is scale * z / fx * 0.5 *
    min(H, W) (render_utils.py:116-122), i.e. a constant scale * 0.5 * min(H, W) pixels on screen; else with
    `use_knn_scale` it follows the local point density, capped by `scale` (:123-127, see knn_point_radii;
    `use_ndc_scale` wins when both are set, as in the reference's if / elif).
    Per pixel the `max_hit` nearest covering points are blended with alpha = occ (:160).  Points outside
    [near, far] in depth are dropped (the camera is built with znear = 1, zfar = 100, :133-134)."""
    img = np.zeros((1, H, W, 4), np.float32)
    if points.shape[0] == 0:
        return img
    knn_r = None
    if use_knn_scale and not use_ndc_scale:          # over ALL points handed in, before the depth filter (:125)
        knn_r = knn_point_radii(points, scale, knn_scale_down, knn_dist2)
    w2c = np.linalg.inv(np.asarray(c2w, np.float64))
    cam = points.astype(np.float64) @ w2c[:3, :3].T + w2c[:3, 3]
    z = cam[:, 2]
    keep = (z > near) & (z < far)
    cam, z, rgb = cam[keep], z[keep], np.asarray(features, np.float64)[keep, :3]
    if z.size == 0:
        return img
    if knn_r is not None:
        knn_r = knn_r[keep]
    fx, fy, cx, cy = ixt[0, 0], ixt[1, 1], ixt[0, 2], ixt[1, 2]
    u = fx * cam[:, 0] / z + cx
    v = fy * cam[:, 1] / z + cy
    if use_ndc_scale:
        world_r = scale * z / fx * (0.5 * H if H <= W else 0.5 * W)
    elif knn_r is not None:
        world_r = knn_r
    else:
        world_r = np.full_like(z, scale)
    rad = world_r * fx / z                                    # pixels
    R = int(np.ceil(rad.max()))
    order = np.argsort(z, kind="stable")                      # front to back
    u, v, z, rad, rgb = u[order], v[order], z[order], rad[order], rgb[order]
    n = z.size
    T = np.ones(H * W)
    hits = np.zeros(H * W, np.int32)
    out = np.zeros((H * W, 3))
    if occ >= 1.0:
        # opaque points: the nearest covering point owns the pixel (hits 2..max_hit see T = 0)
        best = np.full(H * W, n, np.int64)                    # index in depth order
        ui, vi = np.floor(u).astype(np.int64), np.floor(v).astype(np.int64)
        for dy in range(-R, R + 1):
            for dx in range(-R, R + 1):
                px, py = ui + dx, vi + dy
                inside = (px >= 0) & (px < W) & (py >= 0) & (py < H)
                ddx, ddy = (px + 0.5) - u, (py + 0.5) - v
                inside &= ddx * ddx + ddy * ddy <= rad * rad
                idx = np.nonzero(inside)[0]
                np.minimum.at(best, py[idx] * W + px[idx], idx)
        hit = best < n
        out[hit] = rgb[best[hit]] * occ
        T[hit] = 1.0 - occ if occ < 1.0 else 0.0
    else:
        # translucent points: true front-to-back compositing of the max_hit nearest hits, in depth order
        ui, vi = np.floor(u).astype(np.int64), np.floor(v).astype(np.int64)
        for i in range(n):                                    # small inputs only (tests, demos)
            x0, x1 = max(ui[i] - R, 0), min(ui[i] + R, W - 1)
            y0, y1 = max(vi[i] - R, 0), min(vi[i] + R, H - 1)
            if x0 > x1 or y0 > y1:
                continue
            xs, ys = np.meshgrid(np.arange(x0, x1 + 1), np.arange(y0, y1 + 1))
            m = ((xs + 0.5 - u[i]) ** 2 + (ys + 0.5 - v[i]) ** 2) <= rad[i] ** 2
            pix = (ys[m] * W + xs[m])
            pix = pix[hits[pix] < max_hit]
            out[pix] += (T[pix] * occ)[:, None] * rgb[i]
            T[pix] *= 1.0 - occ
            hits[pix] += 1
    img[0, ..., :3] = out.reshape(H, W, 3)
    img[0, ..., 3] = (1.0 - T).reshape(H, W)
    return img


def render_condition_frame(ply_dict, track_info_frame, ego_frame_poses, ego_cam_pose: np.ndarray, frame: int,
                           extrinsic: np.ndarray, ixt: np.ndarray, h: int, w: int, delta_frames: int = 10,
                           shift: float = 0.0, lane_shift_sign: float = 1.0) -> Tuple[np.ndarray, np.ndarray]:
    """One iteration of render_one's frame loop (:207-283): -> (uint8 rgb [h,w,3], uint8 mask [h,w])
    exactly as they are written to `{frame:06d}_{cam}.png` / `..._mask.png`."""
    cloud = assemble_frame(ply_dict, track_info_frame, ego_cam_pose, frame, len(ego_frame_poses), delta_frames, shift)
    c2w = shifted_camera(ego_cam_pose, ego_frame_poses, frame, extrinsic, shift, lane_shift_sign)
    xyz, feat = filter_visible(cloud[:, :3], cloud[:, 3:], c2w, ixt, h, w)
    r = render_points(c2w, ixt, xyz, feat, h, w, use_ndc_scale=True, scale=0.01)
    return (r[0, ..., :3] * 255).astype(np.uint8), (r[0, ..., 3] * 255).astype(np.uint8)
