"""The CPU oracle (oracle/vm_render_np.py) against the golden vectors recorded from the
real reference (tests/golden/make_golden.py).  This is what pins parity."""
import numpy as np
import pytest
import torch

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


@pytest.mark.parametrize("name", ["field_pe_2_3_64", "field_pe_0_2_128", "field_pe_6_6_200"])
def test_nondefault_colour_network_oracle_matches_reference(name):
    """The oracle's positional encodings / late-view network for view_pe, fea_pe, featureC other than opt.py's defaults
    (tensorBase.py:14-21, 97-135), with and without the feature encodings (refine)."""
    g = load_golden(name)
    fld = golden_field_dict(g)
    fld["fea_pe"], fld["view_pe"] = int(g["fea_pe"]), int(g["view_pe"])
    assert fld["renderModule.mlp.0.weight"].shape == (int(g["featureC"]), 27 * (1 + 2 * int(g["fea_pe"])))
    z = oracle.z_schedule(int(g["N_samples"]), np.float64)
    for refine, key in ((True, "eval"), (False, "eval_norefine")):
        rgb, depth = oracle.render_field(fld, g["rays"].astype(np.float64), z, True, 0.0, refine=refine)
        assert rel_err(rgb, g["rgb_" + key]) < 5e-6 and rel_err(depth, g["depth_" + key]) < 5e-6, (refine, rel_err(rgb, g["rgb_" + key]))
    zt = oracle.z_schedule(int(g["N_samples"]), np.float64, jitter=(g["U"], g["U2"]))
    rgb, depth = oracle.render_field(fld, g["rays"].astype(np.float64), zt, True, 0.0)
    assert rel_err(rgb, g["rgb"]) < 5e-6 and rel_err(depth, g["depth"]) < 5e-6


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



# ----------------------------------------------------------------------------- round-2 goldens
def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a))


def test_sample_ray_aabb_matches_reference():
    """TensorBase.sample_ray (tensorBase.py:396-417) recorded from the reference, eval and jittered."""
    import torch
    from oracle import vm_render_torch as ot
    g = load_golden("sample_ray")
    o, d = g["rays"][:, :3], g["rays"][:, 3:]
    nf = [float(v) for v in g["near_far"]]
    for jit, sfx in ((None, ""), (g["U"], "_j")):
        pts, t, inside = oracle.sample_ray_aabb(o, d, g["aabb"], float(g["stepSize"]), int(g["N_samples"]), nf, jitter=jit)
        assert np.abs(t - g["t" + sfx]).max() < 1e-6 and np.abs(pts - g["pts" + sfx]).max() < 2e-5
        assert (inside != g["inside" + sfx]).mean() < 1e-3          # points within fp32 rounding of a box face
        tp, tt, ti = ot.sample_ray_aabb(_t(o), _t(d), _t(g["aabb"]), float(g["stepSize"]), int(g["N_samples"]), nf,
                                        jitter=None if jit is None else _t(jit)[:, None])
        assert torch.equal(tt, _t(g["t" + sfx])) and torch.equal(tp, _t(g["pts" + sfx]))
        assert torch.equal(ti, _t(g["inside" + sfx]))


def test_sixd_to_mtx_matches_reference_including_the_three_view_case():
    import torch
    from oracle import vm_render_torch as ot
    from localrf_amd.rays import sixD_to_mtx
    g = load_golden("sixd_to_mtx")
    for V in (1, 2, 3, 4, 7):
        assert np.abs(oracle.sixd_to_mtx(g[f"r{V}"]) - g[f"m{V}"]).max() < 1e-6, V
        for fn in (ot.sixd_to_mtx, sixD_to_mtx):
            r = _t(g[f"r{V}"]).requires_grad_(True)
            m = fn(r)
            assert np.abs(m.detach().numpy() - g[f"m{V}"]).max() < 1e-6, V
            (m * _t(g[f"ct{V}"])).sum().backward()
            assert np.abs(r.grad.numpy() - g[f"g{V}"]).max() < 1e-5, V
    # the quirk is real: for 3 views the result is not a rotation
    m3 = g["m3"]
    assert np.abs(np.einsum("vij,vkj->vik", m3, m3) - np.eye(3)).max() > 1e-2
    assert np.abs(sixD_to_mtx(_t(g["r3"]), reference_cross=False).numpy() - m3).max() > 1e-2


def test_regularisers_match_reference():
    """density_L1 / TV losses: values and autograd gradients recorded from the reference."""
    import torch
    from oracle import vm_render_torch as ot
    from util import field_from_seed
    g = load_golden("reg_losses")
    f = field_from_seed(g)
    fld = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and ("plane" in k or "line" in k))
           for k, v in f.state_dict().items()}
    for key, fn in (("l1", lambda: ot.density_l1(fld, g["grid"])), ("tv_density", lambda: ot.tv_loss(fld, "density")),
                    ("tv_app", lambda: ot.tv_loss(fld, "app"))):
        for v in fld.values():
            v.grad = None
        out = fn()
        out.backward()
        ref = float(g[key + ".value"])
        assert abs(float(out) - ref) <= 2e-6 * abs(ref), key
        n = 0
        for k, v in fld.items():
            gk = f"{key}.grad.{k}"
            if gk in g:
                assert np.abs(v.grad.numpy() - g[gk]).max() <= 1e-5 * max(np.abs(g[gk]).max(), 1e-12), gk
                n += 1
        assert n == 6, (key, n)


def test_alpha_mask_rebuild_matches_reference():
    """updateAlphaMask (tensorBase.py:501-536): identical binary volume, also through an existing mask."""
    import torch
    from oracle import vm_render_torch as ot
    from util import field_from_seed
    g = load_golden("alpha_mask_rebuild")
    f = field_from_seed(g)
    fld = dict(f.state_dict())
    step = float(f.stepSize)
    m1 = ot.update_alpha_mask(fld, g["g1"], step, 1e-4, float(g["density_shift"]))
    ref1 = np.unpackbits(g["m1"])[:int(np.prod(g["m1_shape"]))].reshape(g["m1_shape"])
    assert (m1.numpy() == ref1).all() and 0.05 < ref1.mean() < 0.95
    fld["alphaMask.alpha_volume"] = m1[None, None]
    fld["alphaMask.aabb"] = fld["aabb"]
    m2 = ot.update_alpha_mask(fld, g["g2"], step, 1e-4, float(g["density_shift"]))
    ref2 = np.unpackbits(g["m2"])[:int(np.prod(g["m2_shape"]))].reshape(g["m2_shape"])
    assert (m2.numpy() == ref2).all()
    fld["alphaMask.alpha_volume"] = m2[None, None]
    with torch.no_grad():
        rgb, depth = ot.render_field(fld, _t(g["rays"]), ot.z_schedule(int(g["N_samples"])), True, 0.0,
                                     density_shift=float(g["density_shift"]))
    assert rel_err(rgb.numpy(), g["rgb"]) < 2e-6 and rel_err(depth.numpy(), g["depth"]) < 2e-6


def test_config2_full_size_port_vs_reference_golden():
    """BASELINE.json configs[1] (300^3, 512 samples): seed-regenerated field, a 256-ray subset of the
    recorded 4096 through the ATen-op port.  (The HIP path is checked on all 4096 in -m gpu.)"""
    import torch
    from oracle import vm_render_torch as ot
    from util import field_from_seed
    g = load_golden("config2_300cube")
    f = field_from_seed(g)
    fld = dict(f.state_dict())
    idx = np.arange(0, 4096, 16)
    with torch.no_grad():
        rgb, depth = ot.render_field(fld, _t(g["rays"][idx]), ot.z_schedule(1536), True, 0.0)
    e = np.abs(rgb.numpy() - g["rgb"][idx]).max(-1)
    flips = e > 1e-4                                   # a sample sitting on the weight > 1e-3 threshold
    assert flips.sum() <= 1 and (g["near_thres"][idx][flips] < 1e-6).all()
    assert rel_err(rgb.numpy()[~flips], g["rgb"][idx][~flips]) < 5e-6
    assert rel_err(depth.numpy(), g["depth"][idx]) < 5e-6
    assert 0.3 < int(g["n_shaded"]) / (4096 * 512) < 0.4


def test_train_grad_128_port_vs_reference_golden():
    import torch
    from oracle import vm_render_torch as ot
    from util import field_from_seed, packed_grad_check
    g = load_golden("field_128_train_grad")
    f = field_from_seed(g)
    fld = {k: v.detach().clone() for k, v in f.state_dict().items()}
    names = [k for k in fld if ("plane" in k or "line" in k or "basis" in k or "renderModule" in k)]
    for k in names:
        fld[k].requires_grad_(True)
    rays = _t(g["rays"]).requires_grad_(True)
    z = ot.z_schedule(int(g["nSamples"]), jitter=(_t(g["U"]), _t(g["U2"])))
    rgb, depth = ot.render_field(fld, rays, z, True, 0.0)
    assert rel_err(rgb.detach().numpy(), g["rgb"]) < 5e-6
    ((rgb * _t(g["g_rgb"])).sum() + (depth * _t(g["g_depth"])).sum()).backward()
    for k in names:
        packed_grad_check(g, k, fld[k].grad.numpy(), 2e-5)
    packed_grad_check(g, "rays", rays.grad.numpy(), 2e-5)


def test_train_grad_500_port_vs_reference_golden():
    """The ATen port (the checker of the GPU test at this size) against the reference's own 500^3 recording: colours,
    depths and the gradients of every tensor on the packed subset, 128 of the 512 rays' worth of loss being enough to
    keep this in the CPU suite's budget would change the gradients -- so all 512 rays are differentiated (25 s)."""
    import torch
    from oracle import vm_render_torch as ot
    from util import field_from_seed, packed_grad_check
    g = load_golden("field_500_train_grad")
    f = field_from_seed(g)
    fld = {k: v.detach().clone() for k, v in f.state_dict().items()}
    names = [k for k in fld if ("plane" in k or "line" in k or "basis" in k or "renderModule" in k)]
    for k in names:
        fld[k].requires_grad_(True)
    rays = _t(g["rays"]).requires_grad_(True)
    z = ot.z_schedule(int(g["nSamples"]), jitter=(_t(g["U"]), _t(g["U2"])))
    rgb, depth = ot.render_field(fld, rays, z, True, 0.0)
    assert rel_err(rgb.detach().numpy(), g["rgb"]) < 5e-6 and rel_err(depth.detach().numpy(), g["depth"]) < 5e-6
    ((rgb * _t(g["g_rgb"])).sum() + (depth * _t(g["g_depth"])).sum()).backward()
    for k in names:
        packed_grad_check(g, k, fld[k].grad.numpy(), 2e-5)
    packed_grad_check(g, "rays", rays.grad.numpy(), 2e-5)


def test_ladder_golden_is_the_train_py_schedule():
    """tests/golden/ladder_64_to_640.npz walks the resolutions train.py computes (train.py:275-288 with the opt.py:61-69
    defaults) and this package's N_to_reso / update_stepSize agree with the recorded resolutions and sample counts."""
    import torch
    from localrf_amd.rays import N_to_reso
    from util import make_field, quiet
    g = load_golden("ladder_64_to_640")
    sides = [round(float(n) ** (1 / 3)) for n in g["n_voxels"]]
    assert sides == [101, 161, 255, 404, 640]
    f = quiet(make_field, [8, 8, 8], "cpu", seed=1)
    for i, n in enumerate(g["n_voxels"].tolist()):
        reso = N_to_reso(int(n), f.aabb)
        assert list(reso) == g[f"reso{i}"].tolist()
        f.update_stepSize(list(reso))
        assert f.nSamples == int(g[f"nSamples{i}"])


@pytest.mark.parametrize("name,prior", [("local_train_3views", False), ("local_train_prior", True)])
def test_local_seeded_goldens_regenerate_and_match_port(name, prior):
    """LocalTensorfs built by THIS package from the golden's seed has the reference's parameters
    (checksum), and scene chain + ATen port reproduce the recorded train forward / pose gradients --
    for exactly three views (torch.cross quirk) and with camera priors ([3,3] r_c2w parameters)."""
    import torch
    from oracle import vm_render_torch as ot
    from util import local_from_golden_seed, torch_scene_chain
    g = load_golden(name)
    cp = {"transforms": {"fl_x": 30.0, "w": 44.0}, "rel_poses": _t(g["rel_poses"])} if prior else None
    lt = local_from_golden_seed(g, camera_prior=cp, scale_density_last=3.0)
    assert tuple(lt.r_c2w[0].shape) == ((3, 3) if prior else (3, 2))
    sd = lt.state_dict()
    W, H = int(g["W"]), int(g["H"])
    ray_ids, view_ids = _t(g["ray_ids"]), [int(v) for v in g["view_ids"]]
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in sd
              if k.split(".")[0] in ("r_c2w", "t_c2w", "exposure", "focal_offset", "center_rel")}
    focal = sd["init_focal"] * leaves["focal_offset"]
    center = torch.tensor([float(W), float(H)]) * leaves["center_rel"]
    r = torch.stack([leaves[f"r_c2w.{v}"] for v in view_ids])
    t = torch.stack([leaves[f"t_c2w.{v}"] for v in view_ids])
    c2w = torch.cat([ot.sixd_to_mtx(r), t[..., None]], -1)
    per = ray_ids.numel() // len(view_ids)
    rays, dirs, ij = torch_scene_chain(ray_ids, c2w, sd["world2rf.0"][None], focal, center, per, W, H, False)
    fld = {k[len("tensorfs.0."):]: v for k, v in sd.items() if k.startswith("tensorfs.0.")}
    z = ot.z_schedule(int(lt.tensorfs[0].nSamples), jitter=(_t(g["U"]), _t(g["U2"])))
    rgb, depth = ot.render_field(fld, rays[0], z, True, 0.0)
    ex = torch.stack([leaves[f"exposure.{v}"] for v in view_ids]).repeat_interleave(per, 0)
    rgbs = torch.bmm(ex, rgb[..., None])[..., 0].clamp(0, 1)
    assert rel_err(rgbs.detach().numpy(), g["rgbs"]) < 5e-6 and rel_err(depth.detach().numpy(), g["depths"]) < 5e-6
    ((rgbs * _t(g["g_rgb"])).sum() + (depth * _t(g["g_depth"])).sum()).backward()
    n = 0
    for k, leaf in leaves.items():
        if ("grad." + k) in g:
            ref = g["grad." + k]
            got = np.zeros_like(ref) if leaf.grad is None else leaf.grad.numpy()   # the reference stacks every
            assert got.shape == ref.shape, k                                       # exposure: unused ones get zeros
            assert np.abs(got - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-6), k
            n += 1
    assert n >= 3 * len(view_ids)


def test_config3_seeded_scene_regenerates():
    """BASELINE.json configs[2] at full size (4 x 300^3): the scene regrown from the seed by this
    package has the reference's fields (checksum) and blending weights; rendering is checked in -m gpu."""
    from util import local_from_golden_seed
    g = load_golden("config3_4x300")
    lt = local_from_golden_seed(g, lr_i=0, n_grow=3)
    assert len(lt.tensorfs) == int(g["n_fields"]) == 4
    assert [f.nSamples for f in lt.tensorfs] == [int(v) for v in g["nSamples"]]


def _geo_inputs(g, device="cpu", grad=True):
    import torch
    t = lambda k: torch.from_numpy(g[k]).to(device)
    leaf = lambda k: t(k).clone().requires_grad_(grad)
    return dict(depth_map=leaf("depth"), directions=leaf("directions"), ij=t("ij"), cam2world=leaf("cam2world"),
                view_ids=t("view_ids"), starting_frame_id=int(g["starting_frame_id"]), fwd_flow=t("fwd_flow"),
                fwd_mask=t("fwd_mask"), bwd_flow=t("bwd_flow"), bwd_mask=t("bwd_mask"), focal=leaf("focal"), center=leaf("center"))


def test_geometric_losses_match_reference():
    """Optical-flow and monocular-depth losses (train.py:385-423 on utils/utils.py:15-59): values, the clipped
    per-ray arrays and every gradient recorded from the reference's own functions."""
    import torch
    from oracle import vm_render_torch as ot
    g = load_golden("geo_losses")
    a = _geo_inputs(g)
    mean, arr = ot.flow_loss(**a)
    mean.backward()
    assert abs(float(mean) - float(g["flow.mean"])) <= 2e-6 * abs(float(g["flow.mean"]))
    assert np.abs(arr.detach().numpy() - g["flow.arr"]).max() <= 1e-5 * np.abs(g["flow.arr"]).max()
    assert ((arr.detach().numpy() == 0) == (g["flow.arr"] == 0)).all()
    for key, v in (("g_depth", a["depth_map"]), ("g_dirs", a["directions"]), ("g_cam2world", a["cam2world"]),
                   ("g_focal", a["focal"]), ("g_center", a["center"])):
        ref = g["flow." + key]
        assert np.abs(v.grad.numpy() - ref).max() <= 1e-5 * np.abs(ref).max(), key
    d = torch.from_numpy(g["depth"]).clone().requires_grad_(True)
    mean, arr = ot.depth_loss(d, torch.from_numpy(g["invdepths"]))
    mean.backward()
    assert abs(float(mean) - float(g["depth.mean"])) <= 2e-6 * abs(float(g["depth.mean"]))
    assert np.abs(arr.detach().numpy() - g["depth.arr"]).max() <= 1e-5 * np.abs(g["depth.arr"]).max()
    assert np.abs(d.grad.numpy() - g["depth.g_depth"]).max() <= 1e-5 * np.abs(g["depth.g_depth"]).max()


def test_upsample_matches_reference():
    """upsample_volume_grid (tensoRF.py:198-233): the 12 resized tensors recorded from the reference."""
    import torch
    from oracle import vm_render_torch as ot
    from util import field_from_seed
    g = load_golden("upsample_grid")
    f = field_from_seed(g)
    up = ot.upsample_vm({k: v.detach() for k, v in f.state_dict().items()}, g["target"])
    n = 0
    for k, v in up.items():
        assert np.abs(v.numpy() - g["up." + k]).max() <= 1e-6, k
        n += 1
    assert n == 12


def test_torch_port_forced_relu_masks():
    """oracle/vm_render_torch.py with the colour network's ReLU masks forced (the gradient tests of the GPU suite feed the
    kernel's masks in): forcing the port's OWN masks changes nothing; flipping one (sample, unit) is counted, moves the
    output by that unit's pre-activation only, and moves the gradients."""
    from oracle import vm_render_torch as ot
    g = load_golden("field_small_train_grad")
    fld = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in golden_field_dict(g).items()}
    rays = torch.from_numpy(g["rays"])
    z = torch.from_numpy(oracle.z_schedule(int(g["N_samples"]), np.float32, jitter=(g["U"], g["U2"])))[None]
    names = ["basis_mat.weight", "renderModule.mlp.0.weight", "renderModule.mlp.2.weight", "app_plane.1", "density_line.0"]

    def grads(masks, info=None):
        leaves = {k: fld[k].clone().requires_grad_(True) for k in names}
        rgb, depth = ot.render_field({**fld, **leaves}, rays, z, True, 0.0, relu_masks=masks, info=info)
        (rgb.sum() + depth.sum()).backward()
        return rgb.detach(), {k: leaves[k].grad for k in names}
    info = {"want_masks": True}
    rgb0, g0 = grads(None, info)
    lin, m1, m2 = info["own_lin"], info["own_m1"], info["own_m2"]
    assert lin.numel() > 1000 and bool((lin[1:] > lin[:-1]).all())
    info1 = {}
    rgb1, g1 = grads((lin, m1, m2), info1)
    assert info1["n_flips"] == 0 and info1["n_forced"] == lin.numel()
    assert torch.equal(rgb0, rgb1)
    for k in names:
        assert float((g0[k] - g1[k]).abs().max()) <= 1e-6 * float(g0[k].abs().max()), k
    m2f = m2.clone()
    m2f[7, 5] = ~m2f[7, 5]
    info2 = {}
    rgb2, g2 = grads((lin, m1, m2f), info2)
    assert info2["n_flips"] == 1
    assert float((g2["renderModule.mlp.2.weight"] - g0["renderModule.mlp.2.weight"]).abs().max()) > 0
    # a sub-list of forced samples leaves the others on their own masks
    info3 = {}
    grads((lin[::2], m1[::2], m2[::2]), info3)
    assert info3["n_forced"] == lin[::2].numel() and info3["n_flips"] == 0


def test_torch_port_relu_density_matches_numpy_oracle():
    """fea2denseAct = "relu" (tensorBase.py:498-499) in the torch port against the numpy oracle."""
    from oracle import vm_render_torch as ot
    g = load_golden("field_small_eval")
    fldn = golden_field_dict(g)
    fldn["fea2denseAct"] = "relu"
    for k in list(fldn):
        if k.startswith("density_plane"):
            fldn[k] = fldn[k] * 3.0
    rays = g["rays"].astype(np.float32)
    zn = oracle.z_schedule(int(g["N_samples"]))
    rn, dn = oracle.render_field(fldn, rays, zn, True, 0.0)
    fldt = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in fldn.items() if isinstance(v, np.ndarray)}
    rt, dt = ot.render_field(fldt, torch.from_numpy(rays), torch.from_numpy(zn)[None], True, 0.0, fea2dense_act="relu")
    assert rel_err(rt.numpy(), rn) < 2e-5 and rel_err(dt.numpy(), dn) < 2e-5


# ----------------------------------------------------------------------------- round-3 golden: the 30-iteration trajectory
def test_trajectory_golden_first_iteration_and_events():
    """tests/golden/trajectory_30it.npz (the REAL LocalTensorfs stepped 30 times by tests/trajectory.py): the plain-torch
    scene chain + field port reproduce iteration 0 from the recorded initial state and sample distances, and the file
    holds every event the GPU replay is meant to cross."""
    import torch
    from oracle import vm_render_torch as ot
    from localrf_amd.rays import sixD_to_mtx
    from util import torch_scene_chain
    import trajectory as tj
    g = load_golden("trajectory_30it")
    sd = {k[5:]: _t(v) for k, v in g.items() if k.startswith("init.")}
    views = [int(v) for v in g["views"][0]]
    ray_ids = _t(g["ray_ids"][0].astype(np.int64))
    focal = sd["init_focal"] * sd["focal_offset"]
    center = torch.tensor([float(tj.W), float(tj.H)]) * sd["center_rel"]
    c2w = torch.cat([sixD_to_mtx(torch.stack([sd[f"r_c2w.{v}"] for v in views])),
                     torch.stack([sd[f"t_c2w.{v}"] for v in views])[..., None]], -1)
    rays, dirs, ij = torch_scene_chain(ray_ids, c2w, sd["world2rf.0"][None], focal, center, tj.PER_VIEW, tj.W, tj.H, False)
    fld = {k[len("tensorfs.0."):]: v for k, v in sd.items() if k.startswith("tensorfs.0.")}
    rgb, depth = ot.render_field(fld, rays[0], _t(g["z.0"])[None], True, 0.0)
    ex = torch.stack([sd[f"exposure.{v}"] for v in views]).repeat_interleave(tj.PER_VIEW, 0)
    rgbs = torch.bmm(ex, rgb[..., None])[..., 0].clamp(0, 1)
    assert rel_err(rgbs.numpy(), g["rgb"][0]) < 2e-6 and rel_err(depth.numpy(), g["depth"][0]) < 2e-6
    want = torch.cat([_t(tj.targets())[v][ray_ids[k * tj.PER_VIEW:(k + 1) * tj.PER_VIEW]] for k, v in enumerate(views)], 0)
    assert abs(float((0.25 * (rgbs - want).abs()).mean()) - float(g["photo"][0])) < 1e-7
    vu, ri = tj.batches(int(g["seed"]) + 3)
    assert np.array_equal(vu, g["view_u"]) and np.array_equal(ri, g["ray_ids"])
    # the events
    assert g["rf_iter"].tolist() == [0] * 6 + list(range(1, 17)) + [0] * 6 + [1, 2]
    assert g["n_fields"].tolist() == [1] * 22 + [2] * 8 and g["can_add_rf"].tolist() == [False] * 21 + [True] + [False] * 8
    assert g["grid_it"][11].tolist() == [20, 20, 20] and g["grid_it"][12].tolist() == [26, 26, 26]
    assert g["nSamples_it"][12] > g["nSamples_it"][11] and len(g["z.12"]) == 2 * (int(g["nSamples_it"][12]) // 6)
    assert g["has_mask"].tolist() == [False] * 15 + [True] * 7 + [False] * 8 and 0.2 < float(g["mask_kept"]) < 0.8
    assert g["regularize"].tolist() == [True] * 18 + [False] * 5 + [True] * 7
    assert "init.r_c2w.3" in g and "init.r_c2w.4" not in g and "final.r_c2w.6" in g and "final.r_c2w.7" not in g
    assert g["photo"][21] < 0.85 * g["photo"][0] and g["photo"][22] > g["photo"][21]      # learns; the new field starts over
