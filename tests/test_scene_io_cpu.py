"""the reference's scene files (multi-element PLY, iteration_*.pth) and the scene-graph
composition, on CPU.  The IDFT basis is pinned against vectors produced by the reference's own
street_gaussian/utils/sh_utils.py (tests/golden/idft_ref.npz, tools/make_golden.py)."""
import math
import os

import numpy as np
import torch
from scipy.spatial.transform import Rotation

from street_crafter_amd import scene_io as sio
from street_crafter_amd.scenes import make_scene


def _actor(n, fourier_dim, K, seed):
    g = torch.Generator().manual_seed(seed)
    return sio.SubModel(name=f"obj_{seed:03d}", xyz=torch.randn(n, 3, generator=g),
                        features_dc=torch.randn(n, fourier_dim, 3, generator=g),
                        features_rest=torch.randn(n, K - 1, 3, generator=g) * 0.3,
                        scaling=torch.randn(n, 3, generator=g) - 3.0, rotation=torch.randn(n, 4, generator=g),
                        opacity=torch.randn(n, 1, generator=g), semantic=torch.randn(n, 2, generator=g),
                        start_frame=10, end_frame=50, fourier_scale=1.0)


def _same(a: sio.SubModel, b: sio.SubModel):
    for f in ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity", "semantic"):
        x, y = getattr(a, f), getattr(b, f)
        assert x.shape == y.shape, (f, x.shape, y.shape)
        assert torch.equal(x.float(), y.float()), f


def test_ply_round_trip_multi_element(tmp_path):
    bkgd = sio.scene_to_submodel(make_scene(500, sh_degree=2, seed=3))
    act = _actor(77, fourier_dim=5, K=9, seed=1)
    p = str(tmp_path / "point_cloud" / "iteration_100" / "point_cloud.ply")
    sio.write_ply(p, [bkgd, act])
    head = open(p, "rb").read(400).decode("ascii", "replace")
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex_background 500\nproperty float x\n")
    got = sio.read_ply(p)
    assert list(got) == ["background", "obj_001"]
    _same(got["background"], bkgd)
    _same(got["obj_001"], act)
    # the channel-major feature layout of gaussian_model.py:85-86: f_dc_{c * F + f}
    raw = dict(sio._read_ply_elements(p))["vertex_obj_001"]
    assert np.float32(raw["f_dc_7"][3]) == act.features_dc[3, 7 % 5, 7 // 5].numpy()


def test_checkpoint_round_trip_safe_load(tmp_path):
    bkgd = sio.scene_to_submodel(make_scene(300, seed=4))
    act = _actor(40, fourier_dim=1, K=4, seed=2)
    p = str(tmp_path / "trained_model" / "iteration_30000.pth")
    sio.save_checkpoint(p, [bkgd, act], extra={"iter": 30000, "actor_pose": {"opt_trans": torch.zeros(3, 3)}})
    got = sio.load_checkpoint(p)            # torch.load(weights_only=True) inside
    assert list(got) == ["background", "obj_002"]
    _same(got["background"], bkgd)
    _same(got["obj_002"], act)


def test_idft_matches_reference_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "idft_ref.npz"))
    for dim in (1, 5, 8):
        np.testing.assert_array_equal(sio.idft(torch.from_numpy(g["times"]), dim).numpy(), g[f"dim{dim}"])


def test_quaternion_product_and_matrix_against_scipy():
    rng = np.random.default_rng(0)
    a, b = rng.normal(size=(50, 4)), rng.normal(size=(50, 4))
    wxyz = lambda q: Rotation.from_quat(np.concatenate([q[:, 1:], q[:, :1]], axis=1))
    prod = sio.quaternion_raw_multiply(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    Ra, Rb = wxyz(a).as_matrix(), wxyz(b).as_matrix()
    np.testing.assert_allclose(sio.quaternion_to_matrix(torch.from_numpy(a)).numpy(), Ra, atol=1e-12)
    np.testing.assert_allclose(sio.quaternion_to_matrix(torch.from_numpy(prod)).numpy(), Ra @ Rb, atol=1e-12)


def test_compose_scene_graph_semantics():
    bkgd = sio.scene_to_submodel(make_scene(200, sh_degree=1, seed=5))
    a1, a2 = _actor(30, 5, 4, seed=1), _actor(20, 5, 4, seed=2)
    q = torch.tensor([math.cos(0.4), 0.0, math.sin(0.4), 0.0])            # 0.8 rad about +y
    t = torch.tensor([1.0, -2.0, 10.0])
    models = {"background": bkgd, a1.name: a1, a2.name: a2}
    cs = sio.compose_scene(models, {a1.name: (q, t)}, frame=30.0)          # a2 has no pose: not visible
    assert cs.graph_gaussian_range == {"background": (0, 200), "obj_001": (200, 230)}
    sc = cs.scene
    assert sc.n == 230 and sc.sh.shape == (230, 4, 3) and sc.sh_degree == 1
    # activations (gaussian_model.py:215-231)
    assert torch.allclose(sc.scales[:200], torch.exp(bkgd.scaling))
    assert torch.allclose(sc.opacities[200:], torch.sigmoid(a1.opacity))
    assert torch.allclose(sc.quats.norm(dim=-1), torch.ones(230), atol=1e-6)
    # rigid transform of the actor (street_gaussian_model.py:351-352) and quaternion composition (:319-320)
    R = torch.from_numpy(Rotation.from_rotvec([0.0, 0.8, 0.0]).as_matrix()).float()
    assert torch.allclose(sc.means[200:], a1.xyz @ R.T + t, atol=1e-5)
    Rl = sio.quaternion_to_matrix(torch.nn.functional.normalize(a1.rotation))
    assert torch.allclose(sio.quaternion_to_matrix(sc.quats[200:]), R[None] @ Rl, atol=1e-5)
    # Fourier colour at the normalised time (frame - start) / (end - start) = 0.5
    base = sio.idft(0.5, 5)[0]
    assert torch.allclose(sc.sh[200:, 0], (a1.features_dc * base[None, :, None]).sum(1), atol=1e-6)
    assert torch.equal(sc.sh[200:, 1:], a1.features_rest)
    assert torch.equal(sc.sh[:200], torch.cat((bkgd.features_dc, bkgd.features_rest), 1))
    # identity pose + inverse activations reproduce the flat scene it was made from
    flat = make_scene(64, seed=6)
    back = sio.compose_scene({"background": sio.scene_to_submodel(flat)}).scene
    assert torch.allclose(back.means, flat.means) and torch.allclose(back.scales, flat.scales, rtol=1e-6)
    assert torch.allclose(back.opacities, flat.opacities, atol=1e-6)
