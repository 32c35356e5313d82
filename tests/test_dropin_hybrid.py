"""The literal drop-in (north_star: "so train.py drops it in unchanged"), as far as it can be exercised without a GPU:
the REFERENCE's own `local_tensorfs.LocalTensorfs` (imported read-only from /root/reference, as tests/golden/make_golden.py
does) with `TensorVMSplit` / `AlphaGridMask` swapped for this package's classes -- INTEGRATION.md "Level 1".  Everything
the reference's scene class touches on a field must exist and behave: construction from `tensorf_args`, `append_frame`,
`append_rf` (which parks the finished field with `.to("cpu")`, local_tensorfs.py:132), the device shuffles of forward
(:432-434,476-479), `get_optparam_groups` (group order is read by index, train.py:480,485), `get_kwargs`, `save` ->
`load`, and `forward` failing loudly (NativeError) where there is no GPU.  Skipped where /root/reference is absent (the
GPU box); nothing under /root/reference is modified."""
import os
import sys

import pytest
import torch

REF = "/root/reference/localTensoRF"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")

FIELD_KW = dict(density_n_comp=[8, 8, 8], appearance_n_comp=[24, 24, 24], app_dim=27, shadingMode="MLP_Fea_late_view",
                near_far=[0.1, 1e3], density_shift=-5, alphaMask_thres=1e-4, distance_scale=25,
                rayMarch_weight_thres=1e-3, pos_pe=0, view_pe=0, fea_pe=0, featureC=128, step_ratio=0.5,
                fea2denseAct="softplus")


@pytest.fixture(scope="module")
def hybrid():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    ref_field, ref_mask, ref_scene = make_golden.import_reference()
    import local_tensorfs as ref_lt
    import localrf_amd
    saved = (ref_lt.TensorVMSplit, ref_lt.AlphaGridMask)
    ref_lt.TensorVMSplit, ref_lt.AlphaGridMask = localrf_amd.TensorVMSplit, localrf_amd.AlphaGridMask   # the class swap
    try:
        yield ref_lt, ref_field, make_golden.quiet
    finally:
        ref_lt.TensorVMSplit, ref_lt.AlphaGridMask = saved


def _scene(ref_lt, quiet, grid=(12, 12, 10)):     # x == y: the reference's load() swaps them (local_tensorfs.py:341)
    torch.manual_seed(7)
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    return quiet(ref_lt.LocalTensorfs, fov=85.6, n_init_frames=5, n_overlap=3, WH=(32, 24), n_iters_per_frame=600,
                 n_iters_reg=100, lr_R_init=5e-3, lr_t_init=5e-4, lr_i_init=1e-3, lr_exposure_init=1e-3, rf_lr_init=0.02,
                 rf_lr_basis=1e-3, lr_decay_target_ratio=0.1, N_voxel_list={}, update_AlphaMask_list=[],
                 camera_prior=None, device="cpu", lr_upsample_reset=True, aabb=aabb, gridSize=list(grid), **FIELD_KW)


def test_reference_scene_runs_its_lifecycle_over_native_fields(hybrid, tmp_path):
    ref_lt, ref_field, quiet = hybrid
    import localrf_amd
    from localrf_amd._native import NativeError
    lt = _scene(ref_lt, quiet)
    assert type(lt).__module__ == "local_tensorfs" and isinstance(lt.tensorfs[0], localrf_amd.TensorVMSplit)
    for _ in range(3):
        quiet(lt.append_frame)
    quiet(lt.append_rf, 3)                                  # parks field 0 with .to(torch.device("cpu")) (:132)
    assert len(lt.tensorfs) == 2 and lt.blending_weights.shape == (len(lt.r_c2w), 2)
    assert lt.tensorfs[0].device == torch.device("cpu")
    # the device shuffles of forward (:432-434, :476-479): .to() returns the module, keeps every attribute a forward reads
    f = lt.tensorfs[0]
    assert f.to(torch.device("cpu")) is f and f.to("cpu") is f
    for attr in ("aabb", "gridSize", "device", "alphaMask", "nSamples", "stepSize", "density_plane", "density_line",
                 "app_plane", "app_line", "basis_mat", "renderModule"):
        assert hasattr(f, attr), attr
    # optimiser groups: the order train.py:480,485 indexes, same tensors as the reference class yields
    torch.manual_seed(3)
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    mine = quiet(localrf_amd.TensorVMSplit, "cpu", aabb, [12, 14, 10], **FIELD_KW)
    torch.manual_seed(3)
    theirs = quiet(ref_field, "cpu", aabb, [12, 14, 10], **FIELD_KW)
    ga, gb = mine.get_optparam_groups(0.02, 1e-3), theirs.get_optparam_groups(0.02, 1e-3)
    assert len(ga) == len(gb) == 6
    for a, b in zip(ga, gb):
        pa, pb = list(a["params"]), list(b["params"])
        assert a["lr"] == b["lr"] and [tuple(p.shape) for p in pa] == [tuple(p.shape) for p in pb]
        assert all(torch.equal(x, y) for x, y in zip(pa, pb))          # same seed -> same parameters, group by group
    assert list(mine.state_dict().keys()) == list(theirs.state_dict().keys())
    ka, kb = mine.get_kwargs(), theirs.get_kwargs()
    assert set(ka) == set(kb) and all(str(ka[k]) == str(kb[k]) for k in ka if k != "aabb")
    # save -> load: the reference's own checkpoint round trip over native fields
    path = str(tmp_path / "ckpt.th")
    lt.save(path)
    ckpt = torch.load(path, weights_only=False)
    kw = dict(ckpt["kwargs"])
    kw["device"] = "cpu"                                    # as train.py:182-191 rebuilds a scene from a checkpoint
    lt2 = quiet(ref_lt.LocalTensorfs, **kw)
    quiet(lt2.load, ckpt["state_dict"])
    sd, sd2 = lt.state_dict(), lt2.state_dict()
    assert list(sd) == list(sd2) and all(torch.equal(sd[k], sd2[k]) for k in sd)
    # the render path itself has no CPU fallback: loud, typed failure
    ray_ids = torch.randint(0, 32 * 24, (2 * 16,))
    with pytest.raises(NativeError):
        lt(ray_ids, torch.tensor([0, 1]), 32, 24, is_train=True)
    with pytest.raises(NativeError):
        lt.tensorfs[-1](torch.randn(8, 6))


def test_members_of_the_reference_class_exist_on_the_native_one(hybrid):
    """Every public attribute of the reference's TensorVMSplit (methods included) exists on the native class: live ones
    are implemented, dead ones (no caller in train.py / renderer.py / local_tensorfs.py) are real wrappers where that is
    two lines (compute_features, get_arange, save, init_render_func) and loud NotImplementedError stubs otherwise."""
    ref_lt, ref_field, quiet = hybrid
    import localrf_amd
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    torch.manual_seed(5)
    theirs = quiet(ref_field, "cpu", aabb, [8, 8, 8], **FIELD_KW)
    torch.manual_seed(5)
    mine = quiet(localrf_amd.TensorVMSplit, "cpu", aabb, [8, 8, 8], **FIELD_KW)
    missing = [n for n in dir(theirs) if not n.startswith("_") and not hasattr(mine, n)]
    assert missing == [], missing
    assert mine.get_arange(1).shape == theirs.get_arange(1).shape and torch.allclose(mine.get_arange(1), theirs.get_arange(1), atol=1e-5)
    for call in (lambda: mine.sample_ray_ndc(torch.zeros(1, 3), torch.ones(1, 3)), lambda: mine.shrink(aabb),
                 lambda: mine.filtering_rays(torch.zeros(4, 6), torch.zeros(4, 3))):
        with pytest.raises(NotImplementedError):
            call()


@pytest.mark.parametrize("cfg", [dict(view_pe=2, fea_pe=2, featureC=64), dict(view_pe=6, fea_pe=0, featureC=128), dict(view_pe=0, fea_pe=6, featureC=256)])
def test_nondefault_colour_network_has_the_reference_state_dict(hybrid, cfg):
    """opt.py:148-157 off their defaults: the native class builds the same parameter tensors (names, shapes, values under
    the same seed) as the reference's, so checkpoints interchange; the generic engine (csrc/lrf_generic.inl) renders them."""
    ref_lt, ref_field, quiet = hybrid
    import localrf_amd
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    kw = dict(FIELD_KW, **cfg)
    torch.manual_seed(9)
    theirs = quiet(ref_field, "cpu", aabb, [10, 12, 8], **kw)
    torch.manual_seed(9)
    mine = quiet(localrf_amd.TensorVMSplit, "cpu", aabb, [10, 12, 8], **kw)
    sd_t, sd_m = theirs.state_dict(), mine.state_dict()
    assert list(sd_t) == list(sd_m)
    for k in sd_t:
        assert sd_t[k].shape == sd_m[k].shape and torch.equal(sd_t[k], sd_m[k]), k
    mine.load_state_dict(sd_t)
    assert mine.get_kwargs()["view_pe"] == cfg["view_pe"] and mine.get_kwargs()["fea_pe"] == cfg["fea_pe"] and mine.get_kwargs()["featureC"] == cfg["featureC"]
