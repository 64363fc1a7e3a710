"""Scene I/O and scene-graph composition: the data formats either side of the hot path.

Reads and writes the two files a StreetCrafter training run leaves behind, and composes their
sub-models into the flat Gaussian arrays the rasterizer operators consume -- so that a trained scene
can drive the renderer / bench without any of the reference's model classes:

  * `point_cloud/iteration_*/point_cloud.ply`: one PLY element `vertex_{model}` per sub-model
    (street_gaussian/models/street_gaussian_model.py:88-110), float32 properties in the order of
    GaussianModel.construct_list_of_attributes (gaussian_model.py:328-342): x y z nx ny nz f_dc_* f_rest_*
    opacity scale_* rot_* semantic_*, with the feature blocks stored channel-major
    (`features.transpose(1, 2).flatten(1)`, gaussian_model.py:85-86, read back :132-133,151-152);
  * `trained_model/iteration_*.pth`: `torch.save` of {model_name: {xyz, feature_dc, feature_rest, scaling,
    rotation, opacity, semantic, ...}} (gaussian_model.py:159-206, street_gaussian_model.py:112-153).
    Loaded with `weights_only=True` only.

Composition (street_gaussian_model.py:273-407 with the activations of gaussian_model.py:215-241):
background as is; every actor's local Gaussians moved by its rigid pose (xyz = R(q_obj) xyz_local + t,
rotation = normalize(q_obj (x) q_local)), its time-dependent colour collapsed from the Fourier
coefficients (gaussian_model_actor.py:67-76, sh_utils.py:120-130); scaling = exp, opacity = sigmoid,
rotation = normalize.  Plain numpy / torch: host-side plumbing, no kernels.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .scenes import Scene


@dataclass
class SubModel:
    """Raw (pre-activation) parameters of one sub-model, as the reference stores them."""
    name: str
    xyz: torch.Tensor               # [n,3]
    features_dc: torch.Tensor       # [n,F,3]  F = 1 (background) or fourier_dim (actors)
    features_rest: torch.Tensor     # [n,K-1,3]
    scaling: torch.Tensor           # [n,3]   log scale
    rotation: torch.Tensor          # [n,4]   wxyz, not normalised
    opacity: torch.Tensor           # [n,1]   logit
    semantic: torch.Tensor = None   # [n,S]   (S may be 0)
    # actors only: the time axis of the Fourier colour model (gaussian_model_actor.py:68-69)
    start_frame: int = 0
    end_frame: int = 1
    fourier_scale: float = 1.0

    def __post_init__(self):
        if self.semantic is None:
            self.semantic = torch.zeros(self.xyz.shape[0], 0)

    @property
    def n(self) -> int:
        return int(self.xyz.shape[0])


# ------------------------------------------------------------------------------------------------
# PLY (binary little endian, float properties) -- the subset plyfile writes for these files
# ------------------------------------------------------------------------------------------------
def _attribute_names(m: SubModel) -> List[str]:
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(m.features_dc.shape[1] * m.features_dc.shape[2])]
    names += [f"f_rest_{i}" for i in range(m.features_rest.shape[1] * m.features_rest.shape[2])]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(m.scaling.shape[1])]
    names += [f"rot_{i}" for i in range(m.rotation.shape[1])]
    names += [f"semantic_{i}" for i in range(m.semantic.shape[1])]
    return names


def _rows(m: SubModel) -> np.ndarray:
    f32 = lambda t: t.detach().cpu().float().numpy()
    xyz = f32(m.xyz)
    f_dc = f32(m.features_dc.transpose(1, 2).flatten(start_dim=1).contiguous())
    f_rest = f32(m.features_rest.transpose(1, 2).flatten(start_dim=1).contiguous())
    return np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, f32(m.opacity), f32(m.scaling),
                           f32(m.rotation), f32(m.semantic)), axis=1).astype("<f4")


def write_ply(path: str, models: Sequence[SubModel]):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    header = ["ply", "format binary_little_endian 1.0"]
    blocks = []
    for m in models:
        names = _attribute_names(m)
        rows = _rows(m)
        assert rows.shape == (m.n, len(names)), (rows.shape, len(names))
        header.append(f"element vertex_{m.name} {m.n}")
        header += [f"property float {a}" for a in names]
        blocks.append(rows.tobytes())
    header.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        for b in blocks:
            f.write(b)


_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "int": "<i4", "int32": "<i4",
              "uint": "<u4", "uint32": "<u4", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
              "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1"}


def _read_ply_elements(path: str) -> List[Tuple[str, np.ndarray]]:
    with open(path, "rb") as f:
        data = f.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    lines = data[:end].decode("ascii").splitlines()
    if lines[0] != "ply" or not lines[1].startswith("format binary_little_endian"):
        raise ValueError(f"{path}: only binary_little_endian PLY files are supported")
    elements = []
    for ln in lines[2:]:
        tok = ln.split()
        if not tok or tok[0] == "comment":
            continue
        if tok[0] == "element":
            elements.append([tok[1], int(tok[2]), []])
        elif tok[0] == "property":
            if tok[1] == "list":
                raise ValueError(f"{path}: list properties are not supported")
            elements[-1][2].append((tok[2], _PLY_TYPES[tok[1]]))
    out, off = [], end
    for name, count, props in elements:
        dt = np.dtype(props)
        arr = np.frombuffer(data, dtype=dt, count=count, offset=off)
        off += dt.itemsize * count
        out.append((name, arr))
    return out


def _block(arr: np.ndarray, prefix: str) -> np.ndarray:
    names = [n for n in arr.dtype.names if n.startswith(prefix)]
    names.sort(key=lambda x: int(x.split("_")[-1]))
    if not names:
        return np.zeros((arr.shape[0], 0), np.float32)
    return np.stack([np.asarray(arr[n], np.float32) for n in names], axis=1)


def read_ply(path: str) -> Dict[str, SubModel]:
    """-> {model_name: SubModel} in file order.  A plain single-model 3DGS file (element `vertex`)
    comes back under the name "background"."""
    models: Dict[str, SubModel] = {}
    for ename, arr in _read_ply_elements(path):
        if not ename.startswith("vertex"):
            continue
        name = ename[7:] if ename.startswith("vertex_") else "background"
        n = arr.shape[0]
        xyz = np.stack([np.asarray(arr[a], np.float32) for a in ("x", "y", "z")], axis=1)
        f_dc = _block(arr, "f_dc_").reshape(n, 3, -1)
        f_rest = _block(arr, "f_rest_").reshape(n, 3, -1)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        models[name] = SubModel(
            name=name, xyz=t(xyz), features_dc=t(f_dc).transpose(1, 2).contiguous(),
            features_rest=t(f_rest).transpose(1, 2).contiguous(),
            scaling=t(_block(arr, "scale_")), rotation=t(_block(arr, "rot_")),
            opacity=t(np.asarray(arr["opacity"], np.float32)[:, None]), semantic=t(_block(arr, "semantic_")))
    return models


# ------------------------------------------------------------------------------------------------
# checkpoints
# ------------------------------------------------------------------------------------------------
_CKPT_KEYS = (("xyz", "xyz"), ("feature_dc", "features_dc"), ("feature_rest", "features_rest"),
              ("scaling", "scaling"), ("rotation", "rotation"), ("opacity", "opacity"), ("semantic", "semantic"))


def save_checkpoint(path: str, models: Sequence[SubModel], extra: Optional[dict] = None):
    """The `is_final=True` layout of StreetGaussianModel.save_state_dict (no optimizer state)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    sd = {m.name: {k: getattr(m, attr).detach().cpu() for k, attr in _CKPT_KEYS} for m in models}
    if extra:
        sd.update(extra)
    torch.save(sd, path)


def load_checkpoint(path: str) -> Dict[str, SubModel]:
    """Sub-models of an `iteration_*.pth`; entries without an `xyz` tensor (actor_pose, sky_cubemap,
    colour / pose correction, the iteration counter) are skipped.  Never unpickles code."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    models: Dict[str, SubModel] = {}
    for name, d in sd.items():
        if not isinstance(d, dict) or "xyz" not in d:
            continue
        kw = {attr: d[k].detach().float() for k, attr in _CKPT_KEYS if k in d}
        models[name] = SubModel(name=name, **kw)
    return models


# ------------------------------------------------------------------------------------------------
# composition
# ------------------------------------------------------------------------------------------------
def idft(time, dim: int) -> torch.Tensor:
    """[T, dim] inverse-DFT basis at normalised time(s): cos(pi t k) on even k, sin(pi t (k+1)) on odd k
    (street_gaussian/utils/sh_utils.py:120-130)."""
    t = torch.as_tensor(time, dtype=torch.float32).reshape(-1, 1)
    k = torch.arange(dim)
    out = torch.zeros(t.shape[0], dim)
    out[:, 0::2] = torch.cos(math.pi * t * k[0::2])
    out[:, 1::2] = torch.sin(math.pi * t * (k[1::2] + 1))
    return out


def quaternion_raw_multiply(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Hamilton product, real part first (street_gaussian/utils/general_utils.py:245-267)."""
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz,
                        aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw), -1)


def quaternion_to_matrix(q: torch.Tensor) -> torch.Tensor:
    """wxyz -> 3x3 of the normalised quaternion (street_gaussian/utils/general_utils.py:125-146)."""
    r, i, j, k = torch.unbind(q, -1)
    s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - s * (j * j + k * k), s * (i * j - k * r), s * (i * k + j * r),
                     s * (i * j + k * r), 1 - s * (i * i + k * k), s * (j * k - i * r),
                     s * (i * k - j * r), s * (j * k + i * r), 1 - s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


@dataclass
class ComposedScene:
    scene: Scene
    graph_gaussian_range: Dict[str, Tuple[int, int]] = field(default_factory=dict)


def actor_features(m: SubModel, frame: float) -> torch.Tensor:
    """[n,K,3] SH coefficients of an actor at `frame`: DC collapsed over the Fourier axis
    (gaussian_model_actor.py:67-76)."""
    fdim = m.features_dc.shape[1]
    if fdim == 1:
        return torch.cat((m.features_dc, m.features_rest), dim=1)
    t = m.fourier_scale * (frame - m.start_frame) / (m.end_frame - m.start_frame)
    base = idft(float(t), fdim)[0].to(m.features_dc)
    dc = torch.sum(m.features_dc * base[..., None], dim=1, keepdim=True)
    return torch.cat((dc, m.features_rest), dim=1)


def compose_scene(models: Dict[str, SubModel], actor_poses: Optional[Dict[str, Tuple[torch.Tensor, torch.Tensor]]] = None,
                  frame: float = 0.0, background: str = "background", sky: str = "sky",
                  include: Optional[Sequence[str]] = None, device=None) -> ComposedScene:
    """Flat arrays in the reference's order: background, actors (in dict order), sky
    (street_gaussian_model.py:273-407).  `actor_poses[name] = (quat wxyz [4], translation [3])` is the
    object-to-world pose of that actor at `frame`; actors without a pose are not visible in this frame
    (the reference's get_visibility) and are left out.  `include` restricts the sub-models."""
    actor_poses = actor_poses or {}
    names = [n for n in models if include is None or n in include]
    order = ([background] if background in names else []) + \
            [n for n in names if n not in (background, sky) and n in actor_poses] + \
            ([sky] if sky in names else [])
    xyz, rot, scl, opa, feat = [], [], [], [], []
    ranges, at = {}, 0
    for name in order:
        m = models[name]
        if name in (background, sky):
            x = m.xyz
            q = torch.nn.functional.normalize(m.rotation)
            f = torch.cat((m.features_dc[:, :1], m.features_rest), dim=1)
        else:
            q_obj, t_obj = actor_poses[name]
            q_obj = torch.as_tensor(q_obj, dtype=torch.float32).reshape(1, 4)
            t_obj = torch.as_tensor(t_obj, dtype=torch.float32).reshape(1, 3)
            R = quaternion_to_matrix(q_obj)[0]
            x = m.xyz @ R.T + t_obj
            q = torch.nn.functional.normalize(quaternion_raw_multiply(q_obj, torch.nn.functional.normalize(m.rotation)))
            f = actor_features(m, frame)
        xyz.append(x); rot.append(q); feat.append(f)
        scl.append(torch.exp(m.scaling)); opa.append(torch.sigmoid(m.opacity))
        ranges[name] = (at, at + m.n)
        at += m.n
    if not order:
        raise ValueError("no sub-model to compose")
    K = max(f.shape[1] for f in feat)
    feat = [torch.cat((f, torch.zeros(f.shape[0], K - f.shape[1], 3)), dim=1) if f.shape[1] < K else f for f in feat]
    sh = torch.cat(feat, dim=0).contiguous()
    deg = int(round(math.sqrt(K))) - 1
    scene = Scene(means=torch.cat(xyz).contiguous(), quats=torch.cat(rot).contiguous(),
                  scales=torch.cat(scl).contiguous(), opacities=torch.cat(opa).contiguous(), sh=sh, sh_degree=deg)
    if device is not None:
        scene = scene.to(device)
    return ComposedScene(scene, ranges)


def scene_to_submodel(scene: Scene, name: str = "background") -> SubModel:
    """Inverse activations: wraps flat arrays (e.g. scenes.make_scene) as a raw sub-model, for writing
    test / benchmark scenes in the reference's formats."""
    op = scene.opacities.clamp(1e-6, 1 - 1e-6)
    return SubModel(name=name, xyz=scene.means.clone(), features_dc=scene.sh[:, :1].clone(),
                    features_rest=scene.sh[:, 1:].clone(), scaling=torch.log(scene.scales),
                    rotation=scene.quats.clone(), opacity=torch.log(op / (1 - op)))
