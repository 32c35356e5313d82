"""The CPU oracle (oracle/vm_render_np.py) against the golden vectors recorded from the
real reference (tests/golden/make_golden.py).  This is what pins parity."""
import numpy as np
import pytest

from oracle import vm_render_np as oracle
from util import golden_field_dict, load_golden, rel_err

FIELD_CASES = ["field_small_eval", "field_small_floater", "field_small_mask", "field_small_default_ns"]


def _nsamples(g):
    n = int(g["N_samples"])
    return n if n > 0 else int(g["nSamples"])


@pytest.mark.parametrize("name", FIELD_CASES)
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_field_forward_matches_reference(name, dt):
    g = load_golden(name)
    fld = golden_field_dict(g)
    z = oracle.z_schedule(_nsamples(g), dt)
    assert np.abs(z - g["z"]).max() < 2e-6
    rgb, depth = oracle.render_field(fld, g["rays"].astype(dt), z, True, float(g["floater"]))
    assert rel_err(rgb, g["rgb"]) < 2e-6
    assert rel_err(depth, g["depth"]) < 2e-6


@pytest.mark.parametrize("name", FIELD_CASES)
def test_feature_lookups_match_reference(name):
    g = load_golden(name)
    fld = golden_field_dict(g)
    u = oracle.normalize_coord(g["xyz0"].reshape(-1, 3).astype(np.float32), fld["aabb"])
    assert np.abs(oracle.density_feature(fld, u) - g["sig_feat"]).max() < 1e-6
    assert np.abs(oracle.app_feature(fld, u)[0] - g["app_feat"]).max() < 1e-6
    # contracted sample positions
    d = g["rays"][:4, 3:6]
    dh = d / np.linalg.norm(d, axis=-1, keepdims=True)
    x = oracle.sample_ray_contracted(g["rays"][:4, :3], dh.astype(np.float32), g["z"])
    assert np.abs(x - g["xyz0"]).max() < 1e-6


def test_mask_actually_masks():
    g = load_golden("field_small_mask")
    fld = golden_field_dict(g)
    z = oracle.z_schedule(_nsamples(g))
    _, dep_m = oracle.render_field(fld, g["rays"], z, True, 0.0)
    fld2 = {k: v for k, v in fld.items() if not k.startswith("alphaMask")}
    _, dep_n = oracle.render_field(fld2, g["rays"], z, True, 0.0)
    assert np.abs(dep_m - dep_n).max() > 1.0       # the fixture exercises the mask branch


def test_train_mode_jitter_forward():
    g = load_golden("field_small_train_grad")
    fld = golden_field_dict(g)
    z = oracle.z_schedule(int(g["N_samples"]), np.float32, jitter=(g["U"], g["U2"]))
    rgb, depth = oracle.render_field(fld, g["rays"], z, True, 0.0)
    assert rel_err(rgb, g["rgb"]) < 2e-6
    assert rel_err(depth, g["depth"]) < 2e-6


def test_gradients_by_finite_differences():
    """Reference autograd gradients (golden) vs central differences of the fp64 oracle on a
    handful of parameter entries and one ray: pins the backward spec the HIP kernels follow."""
    g = load_golden("field_small_train_grad")
    fld = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in golden_field_dict(g).items()}
    z = oracle.z_schedule(int(g["N_samples"]), np.float64, jitter=(g["U"], g["U2"]))
    rays = g["rays"].astype(np.float64)

    def loss():
        rgb, depth = oracle.render_field(fld, rays, z, True, 0.0)
        return float((rgb * g["g_rgb"]).sum() + (depth * g["g_depth"]).sum())

    rng = np.random.default_rng(0)
    checks = []
    for key in ["density_plane.1", "density_line.2", "app_plane.0", "app_line.1", "basis_mat.weight",
                "renderModule.mlp.0.weight", "renderModule.mlp.2.bias", "renderModule.mlp_view.0.weight"]:
        ref = g["grad." + key]
        # choose the entry with the largest reference gradient among a random subset
        idxs = [tuple(rng.integers(0, s) for s in ref.shape) for _ in range(40)]
        idx = max(idxs, key=lambda i: abs(ref[i]))
        h = 1e-4
        old = fld[key][idx]
        fld[key][idx] = old + h
        lp = loss()
        fld[key][idx] = old - h
        lm = loss()
        fld[key][idx] = old
        checks.append((key, (lp - lm) / (2 * h), float(ref[idx])))
    # one ray component
    h = 1e-5
    rays[3, 1] += h
    lp = loss()
    rays[3, 1] -= 2 * h
    lm = loss()
    rays[3, 1] += h
    checks.append(("rays", (lp - lm) / (2 * h), float(g["grad.rays"][3, 1])))
    for key, fd, ref in checks:
        assert abs(fd - ref) <= 2e-3 * max(abs(ref), 1e-3) + 1e-5, (key, fd, ref)


def test_config1_golden_regenerates_from_seed():
    """BASELINE.json configs[0] (64^3, 256 rays x 64 samples): the field is regenerated from
    torch.manual_seed(0) by localrf_amd.TensorVMSplit (same parameter creation order as the
    reference), checked by checksum, then rendered by the oracle."""
    import torch
    from util import make_field, quiet
    g = load_golden("config1_64cube")
    f = quiet(make_field, [64, 64, 64], "cpu", seed=int(g["seed"]))
    s = float(sum(v.double().abs().sum() for v in f.state_dict().values()))
    assert abs(s - float(g["field_sum"][0])) < 1e-6 * float(g["field_sum"][0])
    fld = {k: v.detach().numpy() for k, v in f.state_dict().items()}
    z = oracle.z_schedule(int(g["N_samples"]))
    assert z.shape[0] == 64
    rgb, depth = oracle.render_field(fld, g["rays"], z, True, 0.0)
    assert rel_err(rgb, g["rgb"]) < 2e-6
    assert rel_err(depth, g["depth"]) < 2e-6


def test_local_blend_matches_reference():
    g = load_golden("local_4fields")
    n_fields = int(g["n_fields"])
    lt = {k[3:]: v for k, v in g.items() if k.startswith("lt.")}
    fields = [{k[len(f"tensorfs.{i}."):]: v for k, v in lt.items() if k.startswith(f"tensorfs.{i}.")}
              for i in range(n_fields)]
    n_frames = lt["blending_weights"].shape[0]
    r = np.stack([lt[f"r_c2w.{i}"] for i in range(n_frames)])
    t = np.stack([lt[f"t_c2w.{i}"] for i in range(n_frames)])
    e = np.stack([lt[f"exposure.{i}"] for i in range(n_frames)])
    w2rf = np.stack([lt[f"world2rf.{i}"] for i in range(n_fields)])
    W, H = int(g["W"]), int(g["H"])
    focal = float(lt["init_focal"][0] * lt["focal_offset"][0])
    center = np.array([W, H], np.float32) * lt["center_rel"]
    zs = [oracle.z_schedule(int(n)) for n in g["nSamples"]]
    rgbs, depths, dirs, ij = oracle.render_local(
        fields, w2rf, g["ray_ids"], g["view_ids"], W, H, r, t, focal, center, g["bw"], e, zs)
    assert np.abs(dirs - g["dirs"]).max() < 1e-6
    assert (ij == g["ij"]).all()
    assert rel_err(rgbs, g["rgbs"]) < 5e-6
    assert rel_err(depths, g["depths"]) < 5e-6


def test_sample_ray_aabb_shapes_and_mask():
    rng = np.random.default_rng(3)
    o = rng.normal(size=(16, 3)).astype(np.float32) * 0.3
    d = rng.normal(size=(16, 3)).astype(np.float32)
    d[0, 1] = 0.0
    aabb = np.array([[-2, -2, -2], [2, 2, 2]], np.float32)
    pts, t, inside = oracle.sample_ray_aabb(o, d, aabb, 0.05, 40, (0.1, 1e3))
    assert pts.shape == (16, 40, 3) and t.shape == (16, 40) and inside.shape == (16, 40)
    assert inside[:, 0].all()
    assert ((np.abs(pts) <= 2).all(-1) == inside).all()


@pytest.mark.parametrize("name", FIELD_CASES)
def test_torch_port_matches_reference(name):
    """oracle/vm_render_torch.py (the timed CPU baseline) against the same goldens."""
    import torch
    from oracle import vm_render_torch as ot
    g = load_golden(name)
    fld = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in golden_field_dict(g).items()}
    z = ot.z_schedule(_nsamples(g))
    assert np.abs(z[0].numpy() - g["z"]).max() == 0.0
    with torch.no_grad():
        rgb, depth = ot.render_field(fld, torch.from_numpy(g["rays"]), z, True, float(g["floater"]))
    assert rel_err(rgb.numpy(), g["rgb"]) < 2e-6
    assert rel_err(depth.numpy(), g["depth"]) < 2e-6


def test_torch_port_gradients_match_reference():
    """The ATen-op port under autograd reproduces the reference's recorded gradients
    (incl. the detached view direction, tensorBase.py:628)."""
    import torch
    from oracle import vm_render_torch as ot
    g = load_golden("field_small_train_grad")
    fld = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in golden_field_dict(g).items()}
    names = [k for k in fld if ("plane" in k or "line" in k or "basis" in k or "renderModule" in k)]
    for k in names:
        fld[k].requires_grad_(True)
    rays = torch.from_numpy(g["rays"]).requires_grad_(True)
    z = ot.z_schedule(int(g["N_samples"]), jitter=(torch.from_numpy(g["U"]), torch.from_numpy(g["U2"])))
    rgb, depth = ot.render_field(fld, rays, z, True, 0.0)
    loss = (rgb * torch.from_numpy(g["g_rgb"])).sum() + (depth * torch.from_numpy(g["g_depth"])).sum()
    grads = torch.autograd.grad(loss, [fld[k] for k in names] + [rays])
    for k, gr in zip(names + ["rays"], grads):
        ref = g["grad." + k]
        assert np.abs(gr.numpy() - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-6), k


def test_scene_chain_and_port_match_reference_train_golden():
    """The plain-torch scene chain (tests/util.py::torch_scene_chain, used as the checker for the
    lrf_scene_* kernels) + the ATen-op field port reproduce the reference's LocalTensorfs train
    forward and its pose / intrinsic / exposure gradients."""
    import torch
    from oracle import vm_render_torch as ot
    from localrf_amd.rays import sixD_to_mtx
    from util import torch_scene_chain
    g = load_golden("local_train_grad")
    sd = {k[3:]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in g.items() if k.startswith("lt.")}
    W, H = int(g["W"]), int(g["H"])
    ray_ids, view_ids = torch.from_numpy(g["ray_ids"]), [int(v) for v in g["view_ids"]]
    leaves = {k: sd[k].clone().requires_grad_(True) for k in sd
              if k.split(".")[0] in ("r_c2w", "t_c2w", "exposure", "focal_offset", "center_rel")}
    focal = sd["init_focal"] * leaves["focal_offset"]
    center = torch.tensor([float(W), float(H)]) * leaves["center_rel"]
    r = torch.stack([leaves[f"r_c2w.{v}"] for v in view_ids])
    t = torch.stack([leaves[f"t_c2w.{v}"] for v in view_ids])
    c2w = torch.cat([sixD_to_mtx(r), t[..., None]], -1)
    per = ray_ids.numel() // len(view_ids)
    rays, dirs, ij = torch_scene_chain(ray_ids, c2w, sd["world2rf.0"][None], focal, center, per, W, H, False)
    fld = {k[len("tensorfs.0."):]: v for k, v in sd.items() if k.startswith("tensorfs.0.")}
    z = ot.z_schedule(int(g["nSamples"]), jitter=(torch.from_numpy(g["U"]), torch.from_numpy(g["U2"])))
    rgb, depth = ot.render_field(fld, rays[0], z, True, 0.0)
    ex = torch.stack([leaves[f"exposure.{v}"] for v in view_ids]).repeat_interleave(per, 0)
    rgbs = torch.bmm(ex, rgb[..., None])[..., 0].clamp(0, 1)
    assert (ij.numpy() == g["ij"]).all() and np.abs(dirs.detach().numpy() - g["dirs"]).max() < 1e-6
    assert rel_err(rgbs.detach().numpy(), g["rgbs"]) < 2e-6 and rel_err(depth.detach().numpy(), g["depths"]) < 2e-6
    loss = ((rgbs * torch.from_numpy(g["g_rgb"])).sum() + (depth * torch.from_numpy(g["g_depth"])).sum()
            + (dirs * torch.from_numpy(g["g_dirs"])).sum())
    loss.backward()
    checked = 0
    for k, leaf in leaves.items():
        if leaf.grad is None:
            continue
        ref = g["grad." + k]
        assert np.abs(leaf.grad.numpy() - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-6), k
        checked += 1
    assert checked >= 4 * 3 + 2
