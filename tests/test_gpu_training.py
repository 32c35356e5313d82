"""End-to-end training checks on the GPU: LocalTensorfs.forward (is_train=True) + backward
through poses / exposure / field parameters against the ATen-op port of the reference, and a
short optimisation run through optimizer_step (Adam, upsample, layout-cache invalidation)."""
import numpy as np
import pytest
import torch

from oracle import vm_render_torch as ot
from util import FIELD_KW, quiet, torch_scene_chain

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scene(grid=(20, 24, 28), n_voxel_list=None, seed=5):
    from localrf_amd import LocalTensorfs
    torch.manual_seed(seed)
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]]).to(DEV)
    lt = quiet(LocalTensorfs, fov=85.6, n_init_frames=4, n_overlap=3, WH=(40, 30),
               n_iters_per_frame=600, n_iters_reg=100, lr_R_init=5e-3, lr_t_init=5e-4,
               lr_i_init=0, lr_exposure_init=1e-3, rf_lr_init=0.02, rf_lr_basis=1e-3,
               lr_decay_target_ratio=0.1, N_voxel_list=n_voxel_list or {}, update_AlphaMask_list=[],
               camera_prior=None, device=DEV, lr_upsample_reset=True,
               aabb=aabb, gridSize=list(grid), **FIELD_KW)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for i in range(len(lt.r_c2w)):
            lt.t_c2w[i].add_(0.05 * torch.randn(3, generator=g).to(DEV))
            lt.r_c2w[i].add_(0.05 * torch.randn(3, 2, generator=g).to(DEV))
            lt.exposure[i].add_(0.05 * torch.randn(3, 3, generator=g).to(DEV))
        for p in lt.tensorfs[-1].density_plane:
            p.mul_(3.0)
    return lt


def _batch(lt, n_views=4, per=64, seed=9):
    g = torch.Generator().manual_seed(seed)
    view_ids = torch.arange(n_views)
    ray_ids = torch.randint(0, lt.W * lt.H, (n_views * per,), generator=g)
    return ray_ids.to(DEV), view_ids.to(DEV)


def test_scene_gradients_match_aten_port():
    lt = _scene()
    field = lt.tensorfs[-1]
    z = field.z_schedule(False, -1, torch.device(DEV)).clone()        # fixed schedule for both paths
    field.z_override = z
    ray_ids, view_ids = _batch(lt)
    _g = torch.Generator().manual_seed(77)
    gr = torch.randn(ray_ids.shape[0], 3, generator=_g).to(DEV)
    gd = torch.randn(ray_ids.shape[0], generator=_g).to(DEV)

    def run():
        for p in lt.parameters():
            p.grad = None
        rgb, depth, _, _ = lt(ray_ids, view_ids, lt.W, lt.H, is_train=True)
        ((rgb * gr).sum() + (depth * gd).sum()).backward()
        return (rgb.detach().clone(), depth.detach().clone(),
                {n: p.grad.detach().clone() for n, p in lt.named_parameters() if p.grad is not None})

    rgb_n, depth_n, g_native = run()

    def port_forward(rays, white_bg=True, is_train=False, N_samples=-1, refine=True, floater_thresh=0):
        fld = {k: v for k, v in field.named_parameters()}
        fld = {**{k: v for k, v in field.state_dict(keep_vars=True).items()}, **fld}
        return ot.render_field(fld, rays, z[None], white_bg, floater_thresh)
    field.forward = port_forward                                        # same scene, ATen op chain
    try:
        rgb_p, depth_p, g_port = run()
    finally:
        del field.forward
    assert torch.allclose(rgb_n, rgb_p, rtol=1e-4, atol=1e-5)
    assert torch.allclose(depth_n, depth_p, rtol=1e-4, atol=1e-4)
    assert set(g_native) == set(g_port)
    worst = {}
    for k in g_port:
        denom = float(g_port[k].abs().max())
        if denom == 0.0:
            assert float(g_native[k].abs().max()) == 0.0, k
            continue
        worst[k] = float((g_native[k] - g_port[k]).abs().max()) / denom
    bad = {k: v for k, v in worst.items() if v > 2e-3}
    assert not bad, bad
    # poses, exposure and the field all received gradients
    assert any(k.startswith("r_c2w") for k in worst) and any(k.startswith("t_c2w") for k in worst)
    assert any(k.startswith("exposure") for k in worst) and any("app_plane" in k for k in worst)


def test_short_optimisation_run_with_upsample():
    lt = _scene(grid=(16, 16, 16), n_voxel_list={2: 24 ** 3})
    lt.is_refining = True                                              # rf_iter advances, lr decays
    ray_ids, view_ids = _batch(lt, per=128)
    target = torch.rand(ray_ids.shape[0], 3, device=DEV) * 0.5 + 0.25
    losses = []
    for it in range(14):
        rgb, depth, _, _ = lt(ray_ids, view_ids, lt.W, lt.H, is_train=True)
        loss = (rgb - target).abs().mean()
        losses.append(float(loss.detach()))
        lt.optimizer_step(loss, optimize_poses=True)
    assert all(np.isfinite(losses))
    assert lt.tensorfs[-1].gridSize.tolist() == [24, 24, 24]           # upsample_volume_grid happened
    assert losses[-1] < 0.98 * losses[0] and losses[7] < losses[0], losses    # keeps improving across the upsample
    with torch.no_grad():                                              # eval path still consistent
        rgb_e, _, _, _ = lt(ray_ids, view_ids, lt.W, lt.H, is_train=False)
    assert torch.isfinite(rgb_e).all()


def test_scene_train_gradients_vs_reference_golden():
    """LocalTensorfs.forward(is_train=True) + backward against gradients recorded from the REAL
    reference (tests/golden/make_golden.py::case_local_train): poses, intrinsics, exposure (with
    part of one view clamped), world2rf and the active field."""
    from localrf_amd import LocalTensorfs
    from oracle import vm_render_np as onp
    from util import load_golden
    g = load_golden("local_train_grad")
    assert float(g["clamped"]) > 0.05                     # the clamp mask is exercised
    W, H = int(g["W"]), int(g["H"])
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]]).to(DEV)
    lt = quiet(LocalTensorfs, fov=85.6, n_init_frames=5, n_overlap=3, WH=(W, H),
               n_iters_per_frame=600, n_iters_reg=100, lr_R_init=5e-3, lr_t_init=5e-4,
               lr_i_init=1e-3, lr_exposure_init=1e-3, rf_lr_init=0.02, rf_lr_basis=1e-3,
               lr_decay_target_ratio=0.1, N_voxel_list={}, update_AlphaMask_list=[],
               camera_prior=None, device=DEV, lr_upsample_reset=True,
               aabb=aabb, gridSize=[int(v) for v in g["grid"]], **FIELD_KW)
    ref = {k[3:]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in g.items() if k.startswith("lt.")}
    quiet(lt.load, ref)
    lt = lt.to(DEV)
    field = lt.tensorfs[-1].to(DEV)
    assert field.nSamples == int(g["nSamples"])
    z = onp.z_schedule(field.nSamples, np.float32, jitter=(g["U"], g["U2"]))
    field.z_override = torch.from_numpy(z)
    ray_ids = torch.from_numpy(g["ray_ids"]).to(DEV)
    view_ids = torch.from_numpy(g["view_ids"]).to(DEV)
    rgbs, depths, dirs, ij = lt(ray_ids, view_ids, W, H, is_train=True, white_bg=True)
    assert (ij.cpu().numpy() == g["ij"]).all()
    assert np.abs(dirs.detach().cpu().numpy() - g["dirs"]).max() < 1e-6
    assert np.abs(rgbs.detach().cpu().numpy() - g["rgbs"]).max() < 1e-4
    assert (np.abs(depths.detach().cpu().numpy() - g["depths"]) / np.abs(g["depths"])).max() < 1e-4
    loss = ((rgbs * torch.from_numpy(g["g_rgb"]).to(DEV)).sum()
            + (depths * torch.from_numpy(g["g_depth"]).to(DEV)).sum()
            + (dirs * torch.from_numpy(g["g_dirs"]).to(DEV)).sum())
    loss.backward()
    got = {k: p.grad.detach().cpu().numpy() for k, p in lt.named_parameters() if p.grad is not None}
    want = {k[5:]: v for k, v in g.items() if k.startswith("grad.")}
    assert set(got) == set(want), set(got) ^ set(want)
    bad = {}
    for k, w in want.items():
        denom = float(np.abs(w).max())
        if denom == 0.0:
            assert float(np.abs(got[k]).max()) == 0.0, k
            continue
        # scene-level parameters: tight.  Field MLP rows may see one ReLU / shading-threshold flip
        # among the ~10^3 shaded samples of this small batch, which moves a few elements by ~0.3 %.
        err = float(np.abs(got[k] - w).max()) / denom
        l2 = float(np.linalg.norm(got[k] - w) / np.linalg.norm(w))
        tol = 5e-3 if k.startswith("tensorfs.") else 5e-4
        if err > tol or l2 > 2e-3:
            bad[k] = (err, l2)
    assert not bad, bad


@pytest.mark.parametrize("fov360", [False, True])
def test_scene_rays_kernel_vs_torch_chain(fov360):
    from localrf_amd.scene_ops import scene_rays
    g = torch.Generator().manual_seed(3)
    V, per, n_rf, W, H = 5, 300, 3, 64, 48                # per > block size: strided loop + tail
    ray_ids = torch.randint(0, 7 * W * H, (V * per,), generator=g).to(DEV)
    leaves = [torch.randn(V, 3, 4, generator=g), torch.randn(n_rf, 3, generator=g),
              torch.tensor([41.0]), torch.tensor([30.5, 25.25])]
    g_rays = torch.randn(n_rf, V * per, 6, generator=g).to(DEV)
    g_dirs = torch.randn(V * per, 3, generator=g).to(DEV)
    res = []
    for fn in (scene_rays, torch_scene_chain):
        c2w, sh, fo, ce = [t.clone().to(DEV).requires_grad_(True) for t in leaves]
        rays, dirs, ij = fn(ray_ids, c2w, sh, None if fov360 else fo, None if fov360 else ce, per, W, H, fov360)
        ((rays * g_rays).sum() + (dirs * g_dirs).sum()).backward()
        res.append((rays.detach(), dirs.detach(), ij, c2w.grad, sh.grad, fo.grad, ce.grad))
    a, b = res
    assert (a[2] == b[2]).all()
    assert torch.allclose(a[0], b[0], rtol=1e-5, atol=1e-6) and torch.allclose(a[1], b[1], rtol=1e-5, atol=1e-6)
    for x, y in zip(a[3:5], b[3:5]):
        assert (x - y).abs().max() <= 2e-5 * y.abs().max(), ((x - y).abs().max(), y.abs().max())
    if fov360:
        assert a[5] is None and a[6] is None
    else:
        for x, y in zip(a[5:], b[5:]):
            assert (x - y).abs().max() <= 2e-5 * y.abs().max()


@pytest.mark.parametrize("with_exposure", [True, False])
def test_scene_blend_kernel_vs_torch_chain(with_exposure):
    from localrf_amd.scene_ops import scene_blend
    g = torch.Generator().manual_seed(4)
    V, per, n_rf = 4, 333, 3
    R = V * per
    leaves = [torch.rand(n_rf, R, 3, generator=g), torch.rand(n_rf, R, generator=g) * 5,
              torch.eye(3)[None] * 1.3 + 0.2 * torch.randn(V, 3, 3, generator=g)]
    bw = torch.rand(V, n_rf, generator=g).to(DEV)
    g_rgb, g_dep = torch.randn(R, 3, generator=g).to(DEV), torch.randn(R, generator=g).to(DEV)

    def chain(rgb_f, dep_f, bw, ex, per_view):            # local_tensorfs.py:468-474,481-499
        w = bw.repeat_interleave(per_view, dim=0)
        rgb = torch.zeros_like(rgb_f[0])
        dep = torch.zeros_like(dep_f[0])
        for k in range(rgb_f.shape[0]):
            rgb = rgb + rgb_f[k] * w[:, k][..., None]
            dep = dep + dep_f[k] * w[:, k]
        if ex is not None:
            rgb = torch.bmm(ex.repeat_interleave(per_view, dim=0), rgb[..., None])[..., 0]
        return rgb.clamp(0, 1), dep

    res = []
    for fn in (scene_blend, chain):
        rgb_f, dep_f, ex = [t.clone().to(DEV).requires_grad_(True) for t in leaves]
        rgb, dep = fn(rgb_f, dep_f, bw, ex if with_exposure else None, per)
        ((rgb * g_rgb).sum() + (dep * g_dep).sum()).backward()
        res.append((rgb.detach(), dep.detach(), rgb_f.grad, dep_f.grad, ex.grad))
    a, b = res
    frac = float(((b[0] <= 0) | (b[0] >= 1)).float().mean())
    assert 0.02 < frac < 0.9 or not with_exposure        # the clamp mask is exercised
    assert torch.allclose(a[0], b[0], rtol=1e-5, atol=1e-6) and torch.allclose(a[1], b[1], rtol=1e-5, atol=1e-6)
    assert torch.allclose(a[2], b[2], rtol=1e-5, atol=1e-6) and torch.allclose(a[3], b[3], rtol=1e-5, atol=1e-6)
    if with_exposure:
        assert (a[4] - b[4]).abs().max() <= 2e-5 * b[4].abs().max()
    else:
        assert a[4] is None and b[4] is None


def test_scene_forward_host_ids_and_no_cpu_fallback():
    """Ids may be handed over on the host (staged through pinned memory, no blocking copy) and give
    the same render; a scene that lives on the CPU has no render path at all."""
    from localrf_amd import LocalTensorfs, NativeError
    lt = _scene()
    ray_ids, view_ids = _batch(lt)
    with torch.no_grad():
        a = lt(ray_ids, view_ids, lt.W, lt.H, is_train=False)
        b = lt(ray_ids.cpu(), view_ids.cpu().numpy(), lt.W, lt.H, is_train=False)
        c = lt(ray_ids.cpu(), view_ids.tolist(), lt.W, lt.H, is_train=False, test_id=True)
        d = lt(ray_ids, view_ids, lt.W, lt.H, is_train=False, test_id=True)
    for x, y in zip(a + c, b + d):
        assert torch.equal(x, y)
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    cpu = quiet(LocalTensorfs, fov=85.6, n_init_frames=4, n_overlap=3, WH=(40, 30),
                n_iters_per_frame=600, n_iters_reg=100, lr_R_init=5e-3, lr_t_init=5e-4,
                lr_i_init=0, lr_exposure_init=1e-3, rf_lr_init=0.02, rf_lr_basis=1e-3,
                lr_decay_target_ratio=0.1, N_voxel_list={}, update_AlphaMask_list=[],
                camera_prior=None, device="cpu", lr_upsample_reset=True,
                aabb=aabb, gridSize=[16, 16, 16], **FIELD_KW)
    with pytest.raises(NativeError):
        cpu(ray_ids.cpu(), view_ids.cpu(), 40, 30, is_train=False)


def test_fused_adam_matches_torch_adam():
    """lrf_adam_step against torch.optim.Adam (the reference's optimiser, local_tensorfs.py:88-97,146):
    several groups with different lr, odd sizes (vector tail + unaligned views), 25 steps with lr
    decay, a parameter that gets no gradient on some steps, and step_many across optimisers."""
    from localrf_amd import FusedAdam
    g = torch.Generator().manual_seed(12)
    shapes = [(1, 8, 37, 41), (1, 24, 19, 1), (27, 72), (128,), (3, 131), (5,), (3, 2), (3,)]
    base = [torch.randn(*s, generator=g) for s in shapes]
    pa = [torch.nn.Parameter(b.clone().to(DEV)) for b in base]
    pb = [torch.nn.Parameter(b.clone().to(DEV)) for b in base]

    def make(cls, ps):
        main = cls([{"params": ps[:2], "lr": 0.02}, {"params": ps[2:6], "lr": 1e-3}], betas=(0.9, 0.99))
        pose = [cls([ps[6]], betas=(0.9, 0.99), lr=5e-3), cls([ps[7]], betas=(0.9, 0.99), lr=5e-4)]
        return main, pose
    ma, posea = make(torch.optim.Adam, pa)
    mb, poseb = make(FusedAdam, pb)
    for it in range(25):
        grads = [torch.randn(*s, generator=g).to(DEV) * (10.0 ** (it % 3 - 1)) for s in shapes]
        for ps in (pa, pb):
            for i, (p, gr) in enumerate(zip(ps, grads)):
                p.grad = None if (i == 3 and it % 4 == 1) else gr.clone()
        ma.step(); [o.step() for o in posea]
        mb.step(); FusedAdam.step_many(poseb)
        for opts in ((ma, *posea), (mb, *poseb)):
            for o in opts:
                for grp in o.param_groups:
                    grp["lr"] *= 0.97
    for a, b, s in zip(pa, pb, shapes):
        assert (a - b).abs().max() <= 2e-6 * max(1.0, float(a.abs().max())), s
    sa, sb = ma.state[pa[3]], mb.state[pb[3]]
    assert int(sa["step"]) == int(sb["step"]) == 25 - 6
    assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-5, atol=1e-12)
    # state dicts interchange with torch.optim.Adam
    fresh = torch.optim.Adam([{"params": pb[:2], "lr": 0.02}, {"params": pb[2:6], "lr": 1e-3}], betas=(0.9, 0.99))
    fresh.load_state_dict(mb.state_dict())
    v0 = pb[0]._version
    for p in pb[:6]:
        p.grad = torch.ones_like(p)
    mb.step()
    assert pb[0]._version > v0                      # layout caches keyed on _version see the update
    with pytest.raises(Exception):
        FusedAdam([torch.nn.Parameter(torch.zeros(3))], lr=1e-3).step_many(
            [FusedAdam([_cpu_param_with_grad()], lr=1e-3)])


def _cpu_param_with_grad():
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.ones(3)
    return p


@pytest.mark.parametrize("grid,act", [((20, 24, 28), "softplus"), ((33, 17, 9), "relu"), ((64, 64, 64), "softplus")])
def test_density_l1_kernel_vs_reference_formula(grid, act):
    """lrf_density_l1_fwd/_bwd against the reference's materialising formula (tensoRF.py:83-92),
    non-cubic grids included: its three planes flatten the lattice in three different orders."""
    from util import make_field
    f = quiet(make_field, list(grid), "cpu", seed=8, fea2denseAct=act).to(DEV)
    with torch.no_grad():
        for p in f.density_plane:
            p.mul_(6.0)                           # spread the features over both sides of the clamp / relu

    def reference(fld):                           # verbatim arithmetic of the reference method
        n = int(torch.prod(fld.gridSize))
        feat = torch.zeros((n,), device=DEV)
        for i in range(3):
            pl = fld.density_plane[i].view(-1, int(torch.prod(fld.gridSize[fld.matMode[i]])))
            ln = fld.density_line[i].view(-1, int(fld.gridSize[fld.vecMode[i]]))
            feat = feat + torch.sum(torch.bmm(pl[..., None], ln[:, None]).view(-1, n), dim=0)
        return torch.sqrt(fld.feature2density(feat).clamp(1e-5)).mean()
    res = []
    for fn in (f.density_L1, lambda: reference(f)):
        for p in f.parameters():
            p.grad = None
        out = fn()
        (out * 0.37).backward()
        res.append((out.detach().clone(), [p.grad.clone() for p in list(f.density_plane) + list(f.density_line)]))
    (a, ga), (b, gb) = res
    assert abs(float(a) - float(b)) <= 2e-6 * abs(float(b))
    for x, y in zip(ga, gb):
        assert float(y.abs().max()) > 0
        # sqrt' = 0.5/sqrt(sig) is steep just above the 1e-5 clamp: summation-order rounding of feat
        # there moves single lattice terms by ~1e-4 relative (seen with relu, whose sig = feat)
        assert float((x - y).abs().max()) <= 2e-4 * float(y.abs().max())
    assert all(p.grad is None for n, p in f.named_parameters() if "density" not in n)


def test_density_l1_riding_on_the_render_backward():
    """TensorVMSplit.fuse_density_L1: the regulariser as a third output of the render's autograd node.  Same loss value and
    the same gradients as the two separate nodes (the density tensors: render + regulariser, each summed in the same order,
    added in one more place -- 1e-6 of the tensor's maximum), .grad of every parameter still a view of the ONE flat gradient
    buffer (what the data-parallel exchange reduces in place: no rebucket_grads copy), the flag off or the value unused: the
    plain paths."""
    from util import make_field, make_rays
    f = quiet(make_field, [40, 36, 44], "cpu", seed=5).to(DEV)
    rays = make_rays(512, 3).to(DEV)
    gen = torch.Generator().manual_seed(2)
    gr, gd = torch.randn(512, 3, generator=gen).to(DEV), torch.randn(512, generator=gen).to(DEV)
    z = f.z_schedule(False, -1, DEV)
    f.z_override = z.clone()

    def run(fuse, use_l1=True):
        f.fuse_density_L1 = fuse
        for p in f.parameters():
            p.grad = None
        rgb, depth = f(rays, white_bg=True, is_train=False, N_samples=-1)
        total = (rgb * gr).sum() + (depth * gd).sum()
        l1 = f.density_L1() if use_l1 else None
        if use_l1:
            total = total + 0.37 * l1
        total.backward()
        return (None if l1 is None else float(l1.detach())), {n: p.grad.clone() for n, p in f.named_parameters() if p.grad is not None}, f.grad_bucket()
    try:
        v0, g0, b0 = run(False)
        v1, g1, b1 = run(True)
        _, g2, b2 = run(True, use_l1=False)            # fused forward, value never used: the render's gradients alone
        _, g3, _ = run(False, use_l1=False)
    finally:
        f.fuse_density_L1 = False
        f.z_override = None
    assert v0 == v1
    assert b0 is None and b1 is not None and b2 is not None      # separate nodes: autograd's sums live outside the flat buffer
    assert set(g0) == set(g1) == set(g2)
    for n in g0:
        den = float(g0[n].abs().max())
        assert float((g0[n] - g1[n]).abs().max()) <= 1e-6 * den, (n, float((g0[n] - g1[n]).abs().max()) / den)
        assert float((g2[n] - g3[n]).abs().max()) <= 1e-6 * float(g3[n].abs().max()), n
    assert any(float((g1[n] - g2[n]).abs().max()) > 0 for n in g1 if "density" in n)   # the regulariser did contribute


def test_pose_assemble_kernel_vs_torch_chain():
    """lrf_pose_assemble/_bwd against stack + sixD_to_mtx + cat (local_tensorfs.py:292-299), more
    frames than one launch holds, one frame used twice."""
    from localrf_amd.rays import sixD_to_mtx
    from localrf_amd.scene_ops import pose_assemble
    g = torch.Generator().manual_seed(6)
    V = 70
    base_r = [torch.eye(3, 2) + 0.3 * torch.randn(3, 2, generator=g) for _ in range(V)]
    base_t = [torch.randn(3, generator=g) for _ in range(V)]
    gout = torch.randn(V + 1, 3, 4, generator=g).to(DEV)
    res = []
    for native in (True, False):
        rs = [b.clone().to(DEV).requires_grad_(True) for b in base_r]
        ts = [b.clone().to(DEV).requires_grad_(True) for b in base_t]
        rl, tl = rs + [rs[3]], ts + [ts[3]]
        if native:
            c2w = pose_assemble(rl, tl)
        else:
            c2w = torch.cat([sixD_to_mtx(torch.stack(rl, 0)), torch.stack(tl, 0)[..., None]], -1)
        (c2w * gout).sum().backward()
        res.append((c2w.detach(), torch.stack([r.grad for r in rs]), torch.stack([t.grad for t in ts])))
    a, b = res
    assert torch.allclose(a[0], b[0], rtol=1e-6, atol=1e-6)
    assert (a[1] - b[1]).abs().max() <= 1e-5 * b[1].abs().max()
    assert (a[2] - b[2]).abs().max() <= 1e-6 * b[2].abs().max()


def test_tv_loss_kernel_vs_reference_module():
    """lrf_tv_loss_fwd/_bwd against the reference's TVLoss module arithmetic (utils/utils.py:293-309)
    applied as tensoRF.py:94-110, non-cubic grid, non-default weight."""
    from util import make_field

    class TVLoss(torch.nn.Module):                      # the reference module, verbatim arithmetic
        def __init__(self, TVLoss_weight=1):
            super().__init__()
            self.TVLoss_weight = TVLoss_weight

        def forward(self, x):
            h_x, w_x = x.size()[2], x.size()[3]
            tv = 0
            if h_x > 1:
                tv += torch.pow((x[:, :, 1:, :] - x[:, :, :h_x - 1, :]), 2).mean()
            if w_x > 1:
                tv += torch.pow((x[:, :, :, 1:] - x[:, :, :, :w_x - 1]), 2).mean()
            return self.TVLoss_weight * 2 * tv
    f = quiet(make_field, [37, 41, 29], "cpu", seed=9).to(DEV)
    reg = TVLoss(0.7)

    def reference(planes, lines):
        total = 0
        for i in range(3):
            total = total + reg(planes[i].transpose(0, 1)) * 1e-2 + reg(lines[i].transpose(0, 1)) * 1e-3
        return total
    for native, ref_args in ((f.TV_loss_density, (f.density_plane, f.density_line)),
                             (f.TV_loss_app, (f.app_plane, f.app_line))):
        res = []
        for fn in (lambda: native(reg), lambda: reference(*ref_args)):
            for p in f.parameters():
                p.grad = None
            out = fn()
            (out * 1.3).backward()
            res.append((out.detach().clone(), [p.grad.clone() for p in list(ref_args[0]) + list(ref_args[1])]))
        (a, ga), (b, gb) = res
        assert abs(float(a) - float(b)) <= 1e-5 * abs(float(b))
        for x, y in zip(ga, gb):
            assert float((x - y).abs().max()) <= 1e-5 * float(y.abs().max())
    # any other callable is applied tensor by tensor, as in the reference
    plain = lambda x: (x ** 2).mean()
    want = sum(plain(f.density_plane[i].transpose(0, 1)) * 1e-2 + plain(f.density_line[i].transpose(0, 1)) * 1e-3
               for i in range(3))
    assert abs(float(f.TV_loss_density(plain).detach()) - float(want.detach())) <= 1e-6 * abs(float(want.detach()))


def test_scene_360_forward_backward():
    """fov = 360 (equirectangular directions, utils/ray_utils.py:26-37): the scene renders, its
    directions / rays equal the torch chain, and pose gradients flow (intrinsics get none)."""
    from localrf_amd import LocalTensorfs
    from localrf_amd.rays import get_ray_directions_360, ids2pixel
    torch.manual_seed(2)
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]]).to(DEV)
    lt = quiet(LocalTensorfs, fov=360, n_init_frames=4, n_overlap=3, WH=(64, 32),
               n_iters_per_frame=600, n_iters_reg=100, lr_R_init=5e-3, lr_t_init=5e-4,
               lr_i_init=0, lr_exposure_init=1e-3, rf_lr_init=0.02, rf_lr_basis=1e-3,
               lr_decay_target_ratio=0.1, N_voxel_list={}, update_AlphaMask_list=[],
               camera_prior=None, device=DEV, lr_upsample_reset=True,
               aabb=aabb, gridSize=[16, 20, 24], **FIELD_KW)
    with torch.no_grad():
        for p in lt.tensorfs[-1].density_plane:
            p.mul_(3.0)
    ray_ids, view_ids = _batch(lt, n_views=4, per=32)
    rgb, depth, dirs, ij = lt(ray_ids, view_ids, lt.W, lt.H, is_train=True)
    col, row = ids2pixel(lt.W, lt.H, ray_ids)
    assert torch.allclose(dirs, get_ray_directions_360(col, row, lt.W, lt.H), atol=1e-6)
    assert (ij == torch.stack([col, row], -1)).all()
    assert torch.isfinite(rgb).all() and torch.isfinite(depth).all()
    (rgb.sum() + depth.sum()).backward()
    assert all(lt.r_c2w[i].grad is not None and torch.isfinite(lt.r_c2w[i].grad).all() for i in range(4))
    assert lt.focal_offset.grad is None and lt.center_rel.grad is None


# ----------------------------------------------------------------------------- round-2 goldens
def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_pose_assemble_vs_reference_golden_including_three_views():
    """k_pose_assemble / _bwd against sixD_to_mtx outputs and gradients RECORDED from the reference
    (tests/golden/sixd_to_mtx.npz).  For exactly three views the reference's dim-less torch.cross runs
    over the view axis (utils/utils.py:386): reproduced with cross_over_views."""
    from localrf_amd.scene_ops import pose_assemble
    from util import load_golden
    g = load_golden("sixd_to_mtx")
    for V in (1, 2, 3, 4, 7):
        rs = [_t(g[f"r{V}"][v]).to(DEV).requires_grad_(True) for v in range(V)]
        ts = [torch.zeros(3, device=DEV, requires_grad=True) for _ in range(V)]
        c2w = pose_assemble(rs, ts, cross_over_views=(V == 3))
        assert np.abs(c2w[:, :, :3].detach().cpu().numpy() - g[f"m{V}"]).max() < 1e-6, V
        (c2w[:, :, :3] * _t(g[f"ct{V}"]).to(DEV)).sum().backward()
        got = torch.stack([r.grad for r in rs]).cpu().numpy()
        assert np.abs(got - g[f"g{V}"]).max() <= 1e-5 * max(np.abs(g[f"g{V}"]).max(), 1.0), V
    # without the quirk, three views give proper rotations -- and not the reference's matrices
    rs = [_t(g["r3"][v]).to(DEV) for v in range(3)]
    m = pose_assemble(rs, [torch.zeros(3, device=DEV)] * 3, cross_over_views=False)[:, :, :3]
    assert float((m @ m.transpose(1, 2) - torch.eye(3, device=DEV)).abs().max()) < 1e-5
    assert np.abs(m.cpu().numpy() - g["m3"]).max() > 1e-2


@pytest.mark.parametrize("name,prior", [("local_train_3views", False), ("local_train_prior", True)])
def test_scene_train_gradients_vs_reference_golden_three_views_and_camera_priors(name, prior):
    """LocalTensorfs train forward + backward through the HIP kernels against reference-recorded
    outputs and pose / exposure / intrinsic gradients: a batch of exactly three views (torch.cross
    quirk), and a scene built with camera priors, whose r_c2w parameters are [3,3]
    (local_tensorfs.py:171-176; only columns 0,1 enter sixD_to_mtx)."""
    from oracle import vm_render_np as onp
    from util import load_golden, local_from_golden_seed
    g = load_golden(name)
    cp = {"transforms": {"fl_x": 30.0, "w": 44.0}, "rel_poses": _t(g["rel_poses"])} if prior else None
    lt = local_from_golden_seed(g, DEV, camera_prior=cp, scale_density_last=3.0)
    assert tuple(lt.r_c2w[0].shape) == ((3, 3) if prior else (3, 2))
    field = lt.tensorfs[-1]
    field.z_override = _t(onp.z_schedule(field.nSamples, np.float32, jitter=(g["U"], g["U2"])))
    W, H = int(g["W"]), int(g["H"])
    ray_ids, view_ids = _t(g["ray_ids"]).to(DEV), _t(g["view_ids"]).to(DEV)
    rgbs, depths, dirs, ij = lt(ray_ids, view_ids, W, H, is_train=True, white_bg=True)
    assert np.abs(rgbs.detach().cpu().numpy() - g["rgbs"]).max() < 1e-4
    assert (np.abs(depths.detach().cpu().numpy() - g["depths"]) / np.abs(g["depths"])).max() < 1e-4
    ((rgbs * _t(g["g_rgb"]).to(DEV)).sum() + (depths * _t(g["g_depth"]).to(DEV)).sum()).backward()
    checked, bad = 0, {}
    for k, p in lt.named_parameters():
        gk = "grad." + k
        if gk not in g or k.startswith("tensorfs."):
            continue
        want = g[gk]
        got = np.zeros_like(want) if p.grad is None else p.grad.detach().cpu().numpy()
        assert got.shape == want.shape, (k, got.shape, want.shape)
        denom = float(np.abs(want).max())
        if denom == 0.0:
            assert float(np.abs(got).max()) == 0.0, k
            continue
        err = float(np.abs(got - want).max()) / denom
        if err > 5e-4:
            bad[k] = err
        checked += 1
    assert not bad, bad
    assert checked >= 3 * int(view_ids.numel())
    if prior:                                         # the third column of a [3,3] rotation never enters the path
        v = int(view_ids[0])
        assert float(lt.r_c2w[v].grad[:, 2].abs().max()) == 0.0


def test_config3_full_size_vs_reference_golden():
    """BASELINE.json configs[2] at FULL size: 4 overlapping 300^3 fields regrown from the golden's seed,
    4096 rays (4 views x 1024), explicit blending weights, exposure on -- against the reference's output
    for every ray (tests/golden/config3_4x300.npz)."""
    from util import load_golden, local_from_golden_seed
    g = load_golden("config3_4x300")
    lt = local_from_golden_seed(g, DEV, lr_i=0, n_grow=3)
    ray_ids, view_ids = _t(g["ray_ids"]).to(DEV), _t(g["view_ids"]).to(DEV)
    with torch.no_grad():                                # (one native call: lrf_scene_fwd)
        rgbs, depths, _, _ = lt(ray_ids, view_ids, int(g["W"]), int(g["H"]), is_train=False,
                                blending_weights=_t(g["bw"]).to(DEV), chunk=4096)
        # the rays every field rendered, for the per-ray threshold check below (the same kernel the call used)
        from localrf_amd.scene_ops import scene_rays
        W, H = int(g["W"]), int(g["H"])
        per_field_rays, _, _ = scene_rays(ray_ids, lt.get_cam2world(view_ids.tolist()), torch.stack(list(lt.world2rf), 0),
                                          lt.focal(W), lt.center(W, H), ray_ids.shape[0] // view_ids.shape[0], W, H, False)
    seen = {i: [per_field_rays[i]] for i in range(len(lt.tensorfs))}
    e_rgb = np.abs(rgbs.cpu().numpy() - g["rgbs"]).max(-1)
    e_dep = np.abs(depths.cpu().numpy() - g["depths"]) / np.maximum(np.abs(g["depths"]), 1e-3)
    assert e_dep.max() < 1e-4, e_dep.max()
    # Blended colours are in [0,1]: absolute = relative to 1.  A ray may miss the 1e-4 bar only because one of the FOUR
    # field renders has a sample on the shading threshold weight > 1e-3 (tensorBase.py:622) -- checked per ray, as
    # test_config2_all_rays_vs_reference_golden does: at most 0.2 % of the rays per field, each with a sample whose
    # weight is within 1e-6 of the threshold in this path's own weights, and then (weights <= 0.4) by less than 1e-3.
    bad = e_rgb > 1e-4
    assert bad.sum() <= 32 and e_rgb.max() < 1e-3, (int(bad.sum()), float(e_rgb.max()))
    if bad.any():
        near = np.full(e_rgb.shape[0], np.inf)
        for i, f in enumerate(lt.tensorfs):
            if i not in seen:
                continue
            rays_i = torch.cat(seen[i])
            assert rays_i.shape[0] == e_rgb.shape[0], (i, rays_i.shape)
            with torch.no_grad():
                _, _, w, _, _ = f.render_weights(rays_i, N_samples=-1)
            near = np.minimum(near, (w - f.rayMarch_weight_thres).abs().amin(-1).cpu().numpy())
        for r in np.nonzero(bad)[0]:
            assert near[r] < 1e-6, (int(r), float(near[r]), float(e_rgb[r]))


def test_regularisers_vs_reference_golden():
    """lrf_density_l1_* and lrf_tv_loss_* against values and gradients RECORDED from the reference
    (density_L1 tensoRF.py:83-92, TV_loss_density/app :94-110 with utils.TVLoss)."""
    from util import field_from_seed, load_golden

    class TVLoss(torch.nn.Module):                      # recognised by its TVLoss_weight (utils/utils.py:293-309)
        TVLoss_weight = 1
    g = load_golden("reg_losses")
    f = field_from_seed(g, DEV)
    for key, fn in (("l1", lambda: f.density_L1()), ("tv_density", lambda: f.TV_loss_density(TVLoss())),
                    ("tv_app", lambda: f.TV_loss_app(TVLoss()))):
        for p in f.parameters():
            p.grad = None
        out = fn()
        out = out.mean() if out.dim() else out
        out.backward()
        ref = float(g[key + ".value"])
        assert abs(float(out) - ref) <= 5e-6 * abs(ref), (key, float(out), ref)
        n = 0
        for name, p in f.named_parameters():
            gk = f"{key}.grad.{name}"
            if gk in g:
                want = g[gk]
                assert np.abs(p.grad.cpu().numpy() - want).max() <= 2e-4 * max(np.abs(want).max(), 1e-12), gk
                n += 1
        assert n == 6, (key, n)


# ----------------------------------------------------------------------------- data parallel
def _dp_batch():
    g = torch.Generator().manual_seed(21)
    view_ids = torch.arange(4)
    ray_ids = torch.randint(0, 40 * 30, (4 * 48,), generator=g)
    return ray_ids, view_ids, torch.randn(4 * 48, 3, generator=g), torch.randn(4 * 48, generator=g)


def _dp_loss_backward(lt, ray_ids, view_ids, gr, gd):
    field = lt.tensorfs[-1]
    field.z_override = field.z_schedule(False, -1, torch.device(DEV)).clone()     # same samples on every rank
    for p in lt.parameters():
        p.grad = None
    rgb, depth, _, _ = lt(ray_ids.to(DEV), view_ids.to(DEV), lt.W, lt.H, is_train=True, white_bg=True)
    loss = (rgb * gr.to(DEV)).sum() + (depth * gd.to(DEV)).sum()
    loss.backward()
    return loss.detach()


def _dp_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from localrf_amd.dist import allreduce_grads, shard_views
    torch.cuda.set_device(0)
    lt = _scene()
    ray_ids, view_ids, gr, gd = _dp_batch()
    r_ids, v_ids = shard_views(ray_ids, view_ids)
    per = ray_ids.shape[0] // view_ids.shape[0]
    lo, n = rank * v_ids.shape[0] * per, v_ids.shape[0] * per
    _dp_loss_backward(lt, r_ids, v_ids, gr[lo:lo + n], gd[lo:lo + n])
    bucket = lt.tensorfs[-1].grad_bucket()
    nbytes = allreduce_grads(lt)
    torch.cuda.synchronize()
    # one optimiser step on the reduced gradients: replicas must stay bit-identical
    lt.rf_optimizer.step()
    torch.cuda.synchronize()
    csum = float(sum(p.double().sum() for p in lt.tensorfs[-1].parameters()))
    torch.save({"grads": {n_: p.grad.cpu() for n_, p in lt.named_parameters() if p.grad is not None},
                "bucket": bucket is not None, "bytes": nbytes, "csum": csum}, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_step_on_one_gpu(tmp_path):
    """The data-parallel design of DESIGN.md s6 end to end through the HIP kernels: two processes
    (both on this GPU; gloo, because RCCL refuses two ranks on one device) each render half of a
    4-view batch through LocalTensorfs, all-reduce with localrf_amd.dist.allreduce_grads -- the field
    gradients in place in the flat buffer lrf_render_bwd wrote -- and must end with the gradients of
    the single-process run on the whole batch, and with bit-identical replicas after an Adam step."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    out = str(tmp_path / "dp")
    mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
    got = [torch.load(f"{out}.{r}") for r in range(2)]
    assert got[0]["bucket"] and got[1]["bucket"], "field gradients were not reduced in place"
    assert got[0]["csum"] == got[1]["csum"]
    lt = _scene()
    ray_ids, view_ids, gr, gd = _dp_batch()
    _dp_loss_backward(lt, ray_ids, view_ids, gr, gd)
    n = 0
    for name, p in lt.named_parameters():
        if p.grad is None:
            continue
        for r in range(2):
            a = got[r]["grads"][name]
            den = max(float(p.grad.abs().max()), 1e-12)
            assert float((a - p.grad.cpu()).abs().max()) <= 2e-5 * den, (name, r)
        n += 1
    assert n >= 19 + 3 * 4
    assert got[0]["bytes"] >= 4 * sum(p.numel() for p in lt.tensorfs[-1].parameters() if p.requires_grad)


def _rccl_one_rank_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LRF_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))      # RCCL
    from localrf_amd import dist as ldist
    assert ldist.active()
    lt = _scene()
    field = lt.tensorfs[-1]
    ray_ids, view_ids, gr, gd = _dp_batch()
    res = {}

    def snapshot():
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in lt.named_parameters() if p.grad is not None}

    # (a) the in-place path: five pieces, each behind its lrf_render_bwd_wait event on the side stream
    _dp_loss_backward(lt, ray_ids, view_ids, gr, gd)
    res["a_bucket"] = field.grad_bucket() is not None
    res["a_events"] = field.grad_events_valid()
    res["a_chunks"] = [c[0] for c in field.grad_chunks()]
    before = snapshot()
    st = {}
    res["a_bytes"] = ldist.allreduce_grads(lt, has_grad=ldist.scene_has_grad(lt, view_ids), stats=st, average=True)
    after = snapshot()
    res["a_stats"] = st
    res["a_same"] = sorted(before) == sorted(after) and all(torch.equal(before[k], after[k]) for k in before)
    res["a_views"] = field.grad_bucket() is not None
    # the same backward without a process group's collectives in between, as the reference run: per-plane passes give the
    # gradients of the single pass (LDS / global adds in another order: not bit-identical, within fp32 accumulation noise)
    os.environ["LRF_DIST_FORCE"] = "0"
    _dp_loss_backward(lt, ray_ids, view_ids, gr, gd)
    res["single_pass_chunks"] = [c[0] for c in field.grad_chunks()]
    one = snapshot()
    os.environ["LRF_DIST_FORCE"] = "1"
    res["plane_vs_single"] = max(float((one[k] - before[k]).abs().max()) / max(float(one[k].abs().max()), 1e-12) for k in one)
    # (b) density_L1 in the loss (local_tensorfs.py:361-375): autograd sums the regulariser's gradient first, .grad of the
    # density tensors is no view of the bucket -> rebucket_grads, then the same in-place pieces in plain stream order
    field.z_override = field.z_schedule(False, -1, torch.device(DEV)).clone()
    for p in lt.parameters():
        p.grad = None
    rgb, depth, _, _ = lt(ray_ids.to(DEV), view_ids.to(DEV), lt.W, lt.H, is_train=True, white_bg=True)
    _, l1 = lt.get_reg_loss(None, 0, 0, 1e-4)
    ((rgb * gr.to(DEV)).sum() + (depth * gd.to(DEV)).sum() + 50.0 * l1).backward()
    res["b_bucket_before"] = field.grad_bucket() is not None
    before = snapshot()
    st = {}
    res["b_bytes"] = ldist.allreduce_grads(lt, stats=st)                 # flag path: one host read-back of a few flags
    after = snapshot()
    res["b_stats"] = st
    res["b_same"] = sorted(before) == sorted(after) and all(torch.equal(before[k], after[k]) for k in before)
    res["b_bucket_after"] = field.grad_bucket() is not None
    res["b_events"] = field.grad_events_valid()
    # (c) an empty shard on this rank: no lrf_render_bwd ran, the events are not this backward's -> plain stream order, no hang
    for p in lt.parameters():
        p.grad = None
    rays0 = torch.zeros(0, 6, device=DEV)
    rgb0, depth0 = field(rays0, white_bg=True, is_train=True, N_samples=-1)
    (rgb0.sum() + depth0.sum()).backward()
    res["c_events"] = field.grad_events_valid()
    st = {}
    res["c_bytes"] = ldist.allreduce_grads(field, stats=st)
    torch.cuda.synchronize()
    res["c_stats"] = st
    res["c_zero"] = all(float(p.grad.abs().max()) == 0.0 for p in field.parameters() if p.grad is not None)
    # an optimiser step behind the reduced gradients
    lt.rf_optimizer.step()
    torch.cuda.synchronize()
    res["finite"] = all(bool(torch.isfinite(p).all()) for p in field.parameters())
    torch.save(res, out)
    dist.barrier(device_ids=[0])
    dist.destroy_process_group()


def test_gradient_exchange_on_rccl_with_one_rank_runs_every_collective(tmp_path):
    """the data-parallel exchange had only ever run on gloo.  Here it runs on the nccl (= RCCL)
    backend on this GPU with LRF_DIST_FORCE=1 -- a one-rank group that still issues every collective of the N-rank step:
    the backward with its per-plane appearance passes and five bucket events, each piece handed to RCCL from the side stream
    behind lrf_render_bwd_wait, async works waited for on the caller's stream, in-place average; with a regulariser
    (gradients brought back by rebucket_grads) and with an empty shard (no events: plain stream order).  One rank's sum
    is the identity: gradients must come back bit-identical, nothing may hang."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    out = str(tmp_path / "rccl.pt")
    mp.spawn(_rccl_one_rank_worker, args=(1, port, out), nprocs=1, join=True)
    r = torch.load(out)
    n_field = 4 * sum(p.numel() for p in _scene().tensorfs[-1].parameters() if p.requires_grad)
    assert r["a_bucket"] and r["a_events"] and r["a_chunks"] == [0, 1, 3, 4, 2], r
    assert r["a_same"] and r["a_views"]
    assert r["a_stats"]["collectives"] == 6 and r["a_stats"]["field_bytes"] == n_field and len(r["a_stats"]["chunks"]) == 5
    assert r["a_bytes"] > n_field                              # + poses / exposure of the four sampled views
    assert r["single_pass_chunks"] == [0, 1, 2] and r["plane_vs_single"] < 1e-5, r["plane_vs_single"]
    assert not r["b_bucket_before"], "density_L1 no longer strays: the rebucket path is not exercised"
    assert r["b_same"] and r["b_bucket_after"] and not r["b_events"]
    assert r["b_stats"]["collectives"] == 6 and r["b_stats"]["field_bytes"] == n_field
    assert not r["c_events"] and r["c_stats"]["collectives"] == 5 and r["c_zero"]
    assert r["finite"]


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher must become the launcher (it used to exit
    with a usage message).  Two ranks on this one GPU need --backend gloo (RCCL refuses two ranks per device); everything
    else is the driver's N = 2 run: process group, barriers, max-over-ranks timing, per-rank ray shard, the train step with
    its piecewise gradient exchange, one JSON line from rank 0."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1",
                        "--no-baselines", "--no-pmc"], capture_output=True, text=True, timeout=420, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 4096 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    ar = d["train_step"]["allreduce"]
    assert ar["n_ranks_seen"] == 2 and ar["collectives_per_step"] == 5 and len(ar["bytes_per_piece"]) == 5
    assert d["train_step"]["ms_per_step_without_allreduce"] is not None


def test_bench_under_torch_distributed_run_one_rank(tmp_path):
    """The driver launches the multi-GPU bench as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`.
    The same plumbing with N = 1: RCCL process group on this GPU, barrier + max-over-ranks timing, the train step with its
    gradient all-reduce call, one JSON line from rank 0 -- so that the first real 8-GPU run cannot fail on it."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
           "--no-baselines", "--no-pmc"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=root,
                       env={**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["value"] > 1e6 and abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    assert "RCCL" in d["train_step"]["what"] or "allreduce" in d["train_step"]["what"], d["train_step"]["what"]
    ar = d["train_step"]["allreduce"]                       # one rank, yet every collective of the N-rank step was issued (LRF_DIST_FORCE)
    assert ar["backend"] == "nccl" and ar["forced_at_one_rank"] and ar["collectives_per_step"] == 5 and d["n_ranks_seen"] == 1
    assert sum(ar["bytes_per_piece"]) >= ar["field_bytes"] > 30e6
    assert d["roofline"]["kernel"] in ("k_shade3", "k_march") and d["roofline"]["bound"] in ("hbm", "mfma")


def test_progressive_training_driver_on_synthetic_frames():
    """scripts/train_synth.py: the loop of the reference's train.py:349-474 (sample -> forward -> loss + density_L1 ->
    optimizer_step -> progressive append_frame / append_rf, upsample schedule, alpha-mask rebuilds) around
    LocalTensorfs on synthetic frames -- a short run must visit several resolutions and fields, keep the loss
    finite and falling, and round-trip its checkpoint through the reference's key set."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import train_synth
    out = train_synth.run(frames=16, final=96, iters_per_frame=12, n_max_frames=6, dev=DEV, geo_every=5)
    assert out["finite"] and out["iterations"] > 100, out
    assert out["loss_last"] < 0.7 * out["loss_first"], (out["loss_first"], out["loss_last"])
    assert out["fields"] >= 2 and out["frames"] == 16, (out["fields"], out["frames"], out["events"])
    assert len(out["ms_per_iteration_by_resolution"]) >= 3 and out["final_resolution"] >= 90, out["ms_per_iteration_by_resolution"]
    assert out["checkpoint_roundtrip"] and out["checkpoint_keys_follow_reference"]
    assert out["target_image_stats"]["std"] > 0.05, out["target_image_stats"]       # the teacher scene is not blank
    # the optical-flow / monocular-depth terms (train.py:385-423), field by field while they are in the loss: the camera
    # stays in free space, so the target flow keeps its magnitude over the run (round 2's straight path walked into the
    # teacher's wall: target flow x10, which is what "flow loss 2.0 -> 13" in profiles/r02_train_synth_500.json was --
    # profiles/r09a_geo_curve.md), and the optimisation brings the flow error of the first field down relative to it
    by_field = out["geo_by_field"]
    assert by_field and by_field[0]["field"] == 0 and by_field[0]["records"] >= 4, by_field
    for f in by_field:
        lo, hi = f["target_flow_first_last"]
        assert 0.4 < hi / lo < 2.5 and 0.3 < lo < 4.0, f
        assert f["flow_rel_min"] < 1.3 and f["depth_last"] < 0.2, f
    assert by_field[0]["flow_rel_first"] > 0.9                                      # all poses equal at the start: predicted flow 0
    assert by_field[0]["flow_rel_last"] < 0.8 * by_field[0]["flow_rel_first"], by_field[0]


def test_upsample_vs_reference_golden(built_lib):
    """upsample_volume_grid through lrf_upsample_bilinear against the 12 tensors the reference's
    F.interpolate(bilinear, align_corners=True) produced (tensoRF.py:198-233), then a render on the new grid."""
    from util import field_from_seed, load_golden, make_rays
    g = load_golden("upsample_grid")
    f = field_from_seed(g, DEV)
    f.upsample_volume_grid([int(v) for v in g["target"]])
    n = 0
    for k, v in f.state_dict().items():
        if "plane" in k or "line" in k:
            ref = g["up." + k]
            assert tuple(v.shape) == ref.shape, k
            assert np.abs(v.cpu().numpy() - ref).max() <= 1e-6, k
            n += 1
    assert n == 12
    assert int(f.nSamples) == int(g["nSamples"]) and abs(float(f.stepSize) - float(g["stepSize"])) < 1e-6
    with torch.no_grad():
        rgb, depth = f(make_rays(64, 3).to(DEV), white_bg=True, is_train=False, N_samples=96)
    assert torch.isfinite(rgb).all() and torch.isfinite(depth).all()


def _geo_inputs(g, grad=True):
    t = lambda k: torch.from_numpy(g[k]).to(DEV)
    leaf = lambda k: t(k).clone().requires_grad_(grad)
    return dict(depth_map=leaf("depth"), directions=leaf("directions"), ij=t("ij"), cam2world=leaf("cam2world"),
                view_ids=t("view_ids"), starting_frame_id=int(g["starting_frame_id"]), fwd_flow=t("fwd_flow"),
                fwd_mask=t("fwd_mask"), bwd_flow=t("bwd_flow"), bwd_mask=t("bwd_mask"), focal=leaf("focal"), center=leaf("center"))


def test_geometric_losses_vs_reference_golden(built_lib):
    """lrf_flow_loss_* / lrf_depth_loss_* against values, clipped per-ray arrays and gradients recorded from the
    reference's utils/utils.py functions combined as train.py:385-423 does (first frame, last frame, a frame turned
    far enough that some rays reproject behind it)."""
    from localrf_amd import losses
    from util import load_golden
    g = load_golden("geo_losses")
    a = _geo_inputs(g)
    mean, arr = losses.flow_loss(return_arr=True, **a)
    mean.backward()
    ref = g["flow.arr"]
    assert abs(float(mean) - float(g["flow.mean"])) <= 1e-5 * abs(float(g["flow.mean"]))
    assert ((arr.cpu().numpy() == 0) == (ref == 0)).all()
    assert np.abs(arr.cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max()
    for key, v in (("g_depth", a["depth_map"]), ("g_dirs", a["directions"]), ("g_cam2world", a["cam2world"]),
                   ("g_focal", a["focal"]), ("g_center", a["center"])):
        r = g["flow." + key]
        err = np.abs(v.grad.cpu().numpy().reshape(r.shape) - r).max() / np.abs(r).max()
        print("flow", key, err)
        assert err <= 1e-4, key
    d = torch.from_numpy(g["depth"]).to(DEV).requires_grad_(True)
    mean, arr = losses.depth_loss(d, torch.from_numpy(g["invdepths"]).to(DEV), d.shape[0], return_arr=True)
    mean.backward()
    ref = g["depth.arr"]
    assert abs(float(mean) - float(g["depth.mean"])) <= 1e-5 * abs(float(g["depth.mean"]))
    assert ((arr.cpu().numpy() == 0) == (ref == 0)).all()
    assert np.abs(arr.cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max()
    err = np.abs(d.grad.cpu().numpy() - g["depth.g_depth"]).max() / np.abs(g["depth.g_depth"]).max()
    print("depth g_depth", err)
    assert err <= 1e-4


@pytest.mark.parametrize("V,n", [(16, 256), (3, 1000), (1, 4096), (5, 7)])
def test_geometric_losses_vs_aten_chain_at_batch_sizes(built_lib, V, n):
    """The same losses at train.py's batch shapes (4096 rays over V views; ragged and tiny cases) against the ATen
    restatement (oracle/vm_render_torch.py, pinned to the reference golden above) on the same GPU."""
    from localrf_amd import losses
    from oracle import vm_render_torch as ot
    gen = torch.Generator().manual_seed(100 + V)
    F_ = V + 3
    r6 = torch.eye(3)[:, :2][None].repeat(F_, 1, 1) + 0.05 * torch.randn(F_, 3, 2, generator=gen)
    b1 = torch.nn.functional.normalize(r6[..., 0], dim=-1)
    b2 = torch.nn.functional.normalize(r6[..., 1] - (b1 * r6[..., 1]).sum(-1, keepdim=True) * b1, dim=-1)
    rot = torch.stack([b1, b2, torch.cross(b1, b2, dim=-1)], -1)
    c2w = torch.cat([rot, 0.2 * torch.randn(F_, 3, 1, generator=gen)], -1)
    W, H = 640, 480
    col, row = torch.randint(0, W, (V, n), generator=gen), torch.randint(0, H, (V, n), generator=gen)
    start = 2
    frames = torch.randperm(F_, generator=gen)[:V]
    frames[0] = 0                                                                    # first frame: its backward neighbour is itself
    if V > 1:
        frames[1] = F_ - 1                                                           # last frame: forward mask off
    view_ids = frames + start
    base = dict(ij=torch.stack([col, row], -1), view_ids=view_ids, starting_frame_id=start,
                fwd_flow=4 * torch.randn(V, n, 2, generator=gen), bwd_flow=4 * torch.randn(V, n, 2, generator=gen),
                fwd_mask=(torch.rand(V, n, generator=gen) > 0.2).float(), bwd_mask=(torch.rand(V, n, generator=gen) > 0.2).float())
    focal0, center0 = torch.tensor([500.0]), torch.tensor([W * 0.5, H * 0.5])
    dirs0 = torch.stack([(col + 0.5 - center0[0]) / focal0, -(row + 0.5 - center0[1]) / focal0, -torch.ones(V, n)], -1)
    depth0 = 0.5 + 5 * torch.rand(V, n, generator=gen)
    inv0 = 0.1 + torch.rand(V, n, generator=gen)
    res = {}
    for impl in ("hip", "aten"):
        leaves = dict(depth_map=depth0.to(DEV).requires_grad_(True), directions=dirs0.to(DEV).requires_grad_(True),
                      cam2world=c2w.to(DEV).requires_grad_(True), focal=focal0.to(DEV).requires_grad_(True),
                      center=center0.to(DEV).requires_grad_(True))
        kw = {k: v.to(DEV) if torch.is_tensor(v) else v for k, v in base.items()}
        if impl == "hip":
            fl, farr = losses.flow_loss(return_arr=True, **leaves, **kw)
            dl, darr = losses.depth_loss(leaves["depth_map"], inv0.to(DEV), V, return_arr=True)
        else:
            fl, farr = ot.flow_loss(**leaves, **kw)
            dl, darr = ot.depth_loss(leaves["depth_map"], inv0.to(DEV))
        gf = torch.autograd.grad(fl, list(leaves.values()), retain_graph=True)
        gd = torch.autograd.grad(dl, [leaves["depth_map"]])
        res[impl] = (float(fl), farr.detach(), float(dl), darr.detach(), [x.detach() for x in gf], gd[0].detach())
    h, a = res["hip"], res["aten"]
    assert abs(h[0] - a[0]) <= 1e-5 * abs(a[0]) and abs(h[2] - a[2]) <= 1e-5 * abs(a[2]), (h[0], a[0], h[2], a[2])
    flips = int(((h[1] == 0) != (a[1] == 0)).sum()) + int(((h[3] == 0) != (a[3] == 0)).sum())
    assert flips == 0, flips
    for name, x, y in zip(("depth", "dirs", "cam2world", "focal", "center"), h[4], a[4]):
        err = float((x.reshape(y.shape) - y).abs().max() / y.abs().max().clamp(min=1e-20))
        assert err <= 1e-4, (name, err)
    err = float((h[5] - a[5]).abs().max() / a[5].abs().max())
    assert err <= 1e-4, ("depth loss g_depth", err)


def test_trajectory_replay_vs_reference_golden():
    """30 optimisation iterations recorded from the REAL reference on CPU (tests/golden/make_golden.py::case_trajectory)
    replayed through localrf_amd by the SAME loop (tests/trajectory.py): two append_frame, the switch to refining, the
    schedule rescale, one upsample with a fresh Adam, one alpha-mask rebuild, the end of regularisation, append_rf and the
    first iterations of the second field.  Compared per iteration: control state (rf_iter, grid, nSamples, mask,
    regularize, can_add_rf, views), the photometric and density-L1 losses, every rendered colour and depth; at the end:
    every parameter of the scene and an eval-mode render through both fields.  The golden also holds what the reference
    does against ITSELF when its initial parameters move by 1e-6 (drift.*): the bars below sit above that."""
    from localrf_amd import LocalTensorfs
    from util import load_golden
    import trajectory as tj
    g = load_golden("trajectory_30it")
    kw = dict(FIELD_KW)
    kw.update(tj.FIELD_OVER)
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]]).to(DEV)
    scene_kw = {k: (dict(v) if isinstance(v, dict) else v) for k, v in tj.SCENE_KW.items()}
    lt = quiet(LocalTensorfs, device=DEV, aabb=aabb, gridSize=list(tj.GRID), **scene_kw, **kw)
    quiet(lt.load, {k[5:]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in g.items() if k.startswith("init.")})
    lt = lt.to(DEV)

    def before_forward(scene, it):                       # the reference's train-mode jitter, iteration by iteration
        scene.tensorfs[-1].z_override = torch.from_numpy(g[f"z.{it}"]).to(DEV)

    def after_append_rf(scene):                          # the second field starts from the reference's random draw
        sd = {k[4:]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in g.items() if k.startswith("rf1.")}
        scene.tensorfs[-1].load_state_dict(sd)
    log = quiet(tj.run, lt, g["view_u"], g["ray_ids"], tj.targets(), DEV, before_forward=before_forward,
                after_append_rf=after_append_rf)

    for key, name in (("rf_iter", "rf_iter"), ("n_fields", "n_fields"), ("nSamples", "nSamples_it"),
                      ("has_mask", "has_mask"), ("regularize", "regularize"), ("can_add_rf", "can_add_rf")):
        assert [r[key] for r in log] == g[name].tolist(), key
    assert [r["grid"] for r in log] == g["grid_it"].tolist()
    assert [r["views"] for r in log] == g["views"].tolist()
    assert g["grid_it"][12].tolist() == [26, 26, 26] and g["has_mask"][15] and 0.2 < float(g["mask_kept"]) < 0.8
    assert g["n_fields"][-1] == 2 and not g["regularize"][18] and g["photo"][21] < 0.85 * g["photo"][0]

    photo = np.array([r["photo"] for r in log])
    l1 = np.array([r["l1"] for r in log])
    rel_photo = np.abs(photo - g["photo"]) / g["photo"]
    rel_l1 = np.abs(l1 - g["l1"]) / np.maximum(g["l1"], 1e-12)
    rgb_err = np.array([np.abs(r["rgb"] - g["rgb"][i]).max() for i, r in enumerate(log)])
    dep_err = np.array([(np.abs(r["depth"] - g["depth"][i]) / np.abs(g["depth"][i])).max() for i, r in enumerate(log)])
    print("trajectory: worst relative photometric loss error %.2e (iteration %d), density L1 %.2e, colour %.2e, "
          "relative depth %.2e" % (rel_photo.max(), rel_photo.argmax(), rel_l1.max(), rgb_err.max(), dep_err.max()))
    assert rel_photo.max() < 1e-4, rel_photo
    assert rel_l1.max() < 1e-4, rel_l1
    assert rgb_err.max() < 1e-3 and np.median(rgb_err) < 1e-4, rgb_err
    assert dep_err.max() < 2e-3 and np.median(dep_err) < 1e-4, dep_err

    final = {k: v.detach().cpu().numpy() for k, v in lt.state_dict().items()}
    want = {k[6:]: v for k, v in g.items() if k.startswith("final.")}
    assert set(final) == set(want), set(final) ^ set(want)
    worst = []
    for k, w in want.items():
        assert final[k].shape == w.shape, k
        if w.dtype.kind != "f" or not np.abs(w).max() > 0:
            assert np.array_equal(final[k], w), k
            continue
        d = np.abs(final[k] - w)
        err = float(d.max()) / float(np.abs(w).max())
        l2 = float(np.linalg.norm(d) / np.linalg.norm(w))
        n_over = int((d > 1e-3 * np.abs(w).max()).sum())
        worst.append((err, l2, n_over, d.size, k))
    worst.sort(reverse=True)
    print("trajectory: final parameters, worst (max error / max|p|, relative L2, elements over 1e-3 max|p|, of):",
          [("%.1e" % e, "%.1e" % l, n, sz, k) for e, l, n, sz, k in worst[:4]])
    flips = int((final["tensorfs.0.alphaMask.alpha_volume"] != want["tensorfs.0.alphaMask.alpha_volume"]).sum())
    print("trajectory: alpha-mask cells that differ:", flips, "of", want["tensorfs.0.alphaMask.alpha_volume"].size)
    # Element-wise 1e-3 of max is not a bar an Adam trajectory can hold in general: the update is g / sqrt(E[g^2]), so an
    # element whose own gradient is 1e-4 of the tensor's largest turns a 1e-5-of-max gradient error into a tenth of the
    # learning rate per step.  The golden records what the REFERENCE does with every non-zero gradient element moved by
    # 1e-5 of its tensor's maximum (gdrift.*: up to 24 % of max|p|, hundreds of elements, photometric loss off by 1e-3).
    # The replay has to stay an order of magnitude inside that, hold 3e-3 in relative L2 for every tensor, and have at
    # most 1 % of any tensor's elements beyond 1e-3 of its maximum.  (Run to run the replay itself spreads -- the tile sums
    # are flushed into the gradients with fp32 atomics: over 16 runs the worst tensor, the 160-element density_line.2 of the
    # second field, four iterations old at the end, came out between 1.2e-4 and 1.1e-3 in relative L2; every other tensor
    # stays below 5e-4.  The bar was 1e-3 through round 5 and failed one run in twelve.)
    gd = {k[7:]: float(v) for k, v in g.items() if k.startswith("gdrift.")}
    gd_worst = max(v / float(np.abs(want[k]).max()) for k, v in gd.items() if np.abs(want[k]).max() > 0)
    assert gd_worst > 0.1, gd_worst
    for err, l2, n_over, size, k in worst:
        if k.endswith("alpha_volume"):
            assert l2 < 0.05, (k, l2)                     # a handful of cells at the threshold may flip
            continue
        assert l2 < 3e-3 and err < 1e-2 and n_over <= max(6, size // 100), (k, err, l2, n_over, size)     # (that tensor: 0-4 elements over, 16 runs)
        assert err <= 0.1 * gd_worst, (k, err, gd_worst)
    rel_g = float((np.abs(g["photo"] - g["photo_gnoise"]) / g["photo"]).max())
    assert rel_photo.max() < 0.01 * rel_g, (rel_photo.max(), rel_g)
    for f in lt.tensorfs:
        f.z_override = None
    rgb, depth = tj.final_render(lt, DEV)
    e_rgb = float(np.abs(rgb - g["final_rgb"]).max())
    e_dep = float((np.abs(depth - g["final_depth"]) / np.abs(g["final_depth"])).max())
    print("trajectory: final eval render through both fields: colour %.2e, relative depth %.2e" % (e_rgb, e_dep))
    assert e_rgb < 5e-4 and e_dep < 1e-3


def test_full_frame_scene_forward_in_chunks_over_two_streams(built_lib):
    """A whole view through LocalTensorfs.forward without a tape (renderer.py:65-77 hands the scene an image at a time; one
    active field: lrf_scene_fwd's per-field path): chunks of 32768 rays and more are rendered in pieces of 16384 alternating
    over two streams -- bit-identical to the one-pass form (lrf_debug_set_pipe_chunk(0)), ragged last piece included."""
    from localrf_amd import LocalTensorfs
    W, H = 256, 164                                              # 41984 rays: 16384 + 16384 + 9216
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]]).to(DEV)
    torch.manual_seed(3)
    lt = quiet(LocalTensorfs, fov=85.6, n_init_frames=3, n_overlap=1, WH=(W, H), n_iters_per_frame=600, n_iters_reg=100,
               lr_R_init=5e-3, lr_t_init=5e-4, lr_i_init=0, lr_exposure_init=1e-3, rf_lr_init=0.02, rf_lr_basis=1e-3,
               lr_decay_target_ratio=0.1, N_voxel_list={}, update_AlphaMask_list=[], camera_prior=None, device=DEV,
               lr_upsample_reset=True, aabb=aabb, gridSize=[40, 44, 36], **FIELD_KW).to(DEV)
    ray_ids = torch.arange(W * H, device=DEV)
    view_ids = torch.tensor([1], device=DEV)
    out = {}
    try:
        for chunk in (0, 16384):
            built_lib.lrf_debug_set_pipe_chunk(chunk)
            for f in lt.tensorfs:
                f._ws = None
            with torch.no_grad():
                out[chunk] = [t.clone() for t in lt(ray_ids, view_ids, W, H, is_train=False, white_bg=True, chunk=65536)]
    finally:
        built_lib.lrf_debug_set_pipe_chunk(16384)
        for f in lt.tensorfs:
            f._ws = None
    assert float(out[0][0].std()) > 1e-3
    for a, b in zip(out[0], out[16384]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("chunk,min_chunk,test_id", [(16384, 65536, False), (192, 1, False), (4096, 65536, True)])
def test_scene_forward_single_native_call_equals_the_per_field_path(chunk, min_chunk, test_id):
    """lrf_scene_fwd (what LocalTensorfs.forward calls when no gradient is recorded: rays of every active field, the
    per-field renders in the reference's chunk / field order, blend, exposure) against the same scene put together by
    hand (lrf_scene_rays -> TensorVMSplit.forward per field on the whole batch -> lrf_scene_blend): bit-identical, including a
    chunk that splits the batch unevenly and the held-out-view exposure of local_tensorfs.py:483-492."""
    from localrf_amd import LocalTensorfs
    from util import load_golden
    g = load_golden("local_4fields")
    W, H = int(g["W"]), int(g["H"])
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]]).to(DEV)
    lt = quiet(LocalTensorfs, fov=85.6, n_init_frames=5, n_overlap=3, WH=(W, H),
               n_iters_per_frame=600, n_iters_reg=100, lr_R_init=5e-3, lr_t_init=5e-4,
               lr_i_init=0, lr_exposure_init=1e-3, rf_lr_init=0.02, rf_lr_basis=1e-3,
               lr_decay_target_ratio=0.1, N_voxel_list={}, update_AlphaMask_list=[],
               camera_prior=None, device=DEV, lr_upsample_reset=True,
               aabb=aabb, gridSize=[int(v) for v in g["grid"]], **FIELD_KW)
    quiet(lt.load, {k[3:]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in g.items() if k.startswith("lt.")})
    lt = lt.to(DEV)
    lt.min_chunk = min_chunk
    ray_ids = torch.from_numpy(g["ray_ids"]).to(DEV)
    view_ids = torch.from_numpy(g["view_ids"]).to(DEV)
    assert len(lt.tensorfs) >= 3
    with torch.no_grad():
        native = lt(ray_ids, view_ids, W, H, is_train=False, white_bg=True, chunk=chunk, test_id=test_id, floater_thresh=0.0)
        # the same scene by hand: one launch group per step, every field on the whole batch
        from localrf_amd.scene_ops import scene_blend, scene_rays
        views = view_ids.tolist()
        host = lt._blending_host()[views]
        active = torch.nonzero(host.sum(0))[:, 0].tolist()
        assert len(active) >= 2
        rays, dirs, ij = scene_rays(ray_ids, lt.get_cam2world(views), torch.stack([lt.world2rf[rf] for rf in active], 0),
                                    lt.focal(W), lt.center(W, H), ray_ids.shape[0] // len(views), W, H, False)
        cols = [lt.tensorfs[rf](rays[k], is_train=False, white_bg=True, N_samples=-1) for k, rf in enumerate(active)]
        bw = lt.blending_weights[view_ids][:, active]
        rgbs, depth = scene_blend(torch.stack([c[0] for c in cols]), torch.stack([c[1] for c in cols]), bw,
                                  lt._exposure_for(view_ids, test_id), ray_ids.shape[0] // len(views))
    for a, b, name in zip((rgbs, depth, dirs, ij), native, ("rgbs", "depth", "directions", "ij")):
        assert torch.equal(a, b), name
    if test_id:                                                 # and the reference's own values for this call
        assert np.abs(native[0].cpu().numpy() - g["rgbs_testid"]).max() < 1e-4


def test_scene_with_a_nondefault_colour_network_trains_and_renders():
    """LocalTensorfs over fields with positional encodings and another hidden width (the generic engine, csrc/lrf_generic.inl):
    the no-grad scene forward (one native call) equals the taped path's outputs bit for bit, a backward fills every field
    and pose gradient with finite values, an optimiser step moves the loss, and the state dict keeps the reference's shapes."""
    from localrf_amd import LocalTensorfs
    W, H = 32, 24
    torch.manual_seed(7)
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]]).to(DEV)
    kw = dict(FIELD_KW, fea_pe=1, view_pe=2, featureC=48)
    lt = quiet(LocalTensorfs, fov=85.6, n_init_frames=5, n_overlap=3, WH=(W, H),
               n_iters_per_frame=600, n_iters_reg=100, lr_R_init=5e-3, lr_t_init=5e-4,
               lr_i_init=0, lr_exposure_init=1e-3, rf_lr_init=0.02, rf_lr_basis=1e-3,
               lr_decay_target_ratio=0.1, N_voxel_list={}, update_AlphaMask_list=[],
               camera_prior=None, device=DEV, lr_upsample_reset=True, aabb=aabb, gridSize=[20, 24, 28], **kw).to(DEV)
    sd = lt.state_dict()
    assert tuple(sd["tensorfs.0.renderModule.mlp.0.weight"].shape) == (48, 27 * 3)
    assert tuple(sd["tensorfs.0.renderModule.mlp_view.0.weight"].shape) == (3, 48 + 15)
    g = torch.Generator().manual_seed(3)
    view_ids = torch.tensor([0, 2, 4])
    ray_ids = torch.randint(0, W * H, (3 * 64,), generator=g)
    target = torch.rand(3 * 64, 3, generator=g).to(DEV)
    with torch.no_grad():
        e_off = lt(ray_ids, view_ids.tolist(), W, H, is_train=False, white_bg=True)
        lt.is_refining = True                                       # local_tensorfs.py:446: refine = self.is_refining switches the feature encodings on
        e = lt(ray_ids, view_ids.tolist(), W, H, is_train=False, white_bg=True)
    assert float((e[0] - e_off[0]).abs().max()) > 1e-4
    lt.tensorfs[-1].z_override = lt.tensorfs[-1].z_schedule(False, -1, torch.device(DEV)).clone()      # the same samples in both paths
    losses = []
    for it in range(3):
        rgb, depth, _, _ = lt(ray_ids, view_ids.tolist(), W, H, is_train=True, white_bg=True)
        if it == 0:
            with torch.no_grad():
                e2 = lt(ray_ids, view_ids.tolist(), W, H, is_train=False, white_bg=True)
            assert torch.equal(e[0], e2[0])
            assert float((rgb.detach() - e2[0]).abs().max()) < 1e-6 and float((depth.detach() - e2[1]).abs().max()) < 1e-5
        loss = ((rgb - target) ** 2).mean()
        losses.append(float(loss.detach()))
        lt.optimizer_step(loss, True)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_render_calls_can_be_captured_in_a_hip_graph():
    """The C calls enqueue kernels, memsets and event waits only (no allocation, no synchronisation, no host read-back after
    their first use on a device), so a caller can capture them in a HIP graph (torch.cuda.CUDAGraph) and replay it: the eval
    forward bit-identically, forward + backward of a training step -- with the library's side stream and events forked
    from and joined back into the capturing stream -- with the eager gradients up to the order of the scatter adds."""
    from util import make_field, make_rays
    f = quiet(make_field, [48, 40, 44], "cpu", seed=5).to(DEV)
    with torch.no_grad():
        for p in f.density_plane:
            p.mul_(3.0)
    R = 600
    rays = make_rays(R, 3, pinhole=True).to(DEV)
    f.z_override = f.z_schedule(False, 96, torch.device(DEV)).clone()
    with torch.no_grad():
        rgb0, d0 = f(rays, white_bg=True, is_train=False, N_samples=96)
        out = (torch.empty_like(rgb0), torch.empty_like(d0))
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            f(rays, white_bg=True, is_train=False, N_samples=96, out=out)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            f(rays, white_bg=True, is_train=False, N_samples=96, out=out)
        out[0].zero_(); out[1].zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out[0], rgb0) and torch.equal(out[1], d0)
    gen = torch.Generator().manual_seed(3)
    gr, gd = torch.randn(R, 3, generator=gen).to(DEV), torch.randn(R, generator=gen).to(DEV)
    r = rays.clone().requires_grad_(True)

    def fb():
        for p in f.parameters():
            p.grad = None
        r.grad = None
        rgb, depth = f(r, white_bg=True, is_train=False, N_samples=96)
        ((rgb * gr).sum() + (depth * gd).sum()).backward()
    with torch.cuda.stream(side):
        for _ in range(2):
            fb()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ref = [p.grad.clone() for p in f.parameters() if p.grad is not None] + [r.grad.clone()]
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        fb()
    for _ in range(2):
        g2.replay()
    torch.cuda.synchronize()
    got = [p.grad for p in f.parameters() if p.grad is not None] + [r.grad]
    for a, b in zip(got, ref):
        assert torch.isfinite(a).all() and float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-12)


def test_gradient_buckets_become_final_in_order_and_can_be_awaited_separately(built_lib):
    """lrf_render_bwd_wait (the hand-off localrf_amd.dist uses to start the density all-reduce while the rest of the
    backward runs): after a backward, a side stream that waits for bucket k only must see that bucket's gradients final --
    equal to what the caller's stream sees at the end -- for k = 0 (density), 1 (colour network), 2 (appearance); the
    segments of grad_bucket() tile the flat buffer in that order; the chunked single-process 'reduction' of
    localrf_amd.dist._reduce_field_chunks touches every byte once.  Bad bucket ids fail loudly."""
    import ctypes as C
    from localrf_amd import _native as N
    from util import make_field, make_rays
    f = quiet(make_field, [48, 40, 44], "cpu", seed=7).to(DEV)
    with torch.no_grad():
        for p in f.density_plane:
            p.mul_(3.0)
    rays = make_rays(700, 9, pinhole=True).to(DEV)
    g = torch.Generator().manual_seed(10)
    gr, gd = torch.randn(700, 3, generator=g).to(DEV), torch.randn(700, generator=g).to(DEV)
    rgb, depth = f(rays, white_bg=True, is_train=False, N_samples=120)
    ((rgb * gr).sum() + (depth * gd).sum()).backward()
    flat, held = f.grad_bucket()
    segs = f.grad_segments()
    assert len(segs) == 3 and sorted(segs)[0][0] == 0 and sorted(segs)[-1][1] == flat.numel()
    covered = sorted(segs)
    assert all(covered[i][1] == covered[i + 1][0] for i in range(2))          # the three buckets tile the parameter part
    side = torch.cuda.Stream(DEV)
    snaps = []
    for which, (a, b) in enumerate(segs):
        f._wait_bwd_bucket(which, side)
        with torch.cuda.stream(side):
            snaps.append(flat[a:b].clone())
    torch.cuda.synchronize()
    for (a, b), snap in zip(segs, snaps):
        assert torch.equal(snap, flat[a:b])
        assert float(snap.abs().max()) > 0
    for which in (3, 4):                                     # appearance planes 0 / 1 alone: without per-plane passes the same point as 2
        f._wait_bwd_bucket(which, side)
    assert [c[0] for c in f.grad_chunks()] == [0, 1, 2] and f.grad_events_valid()
    assert N.lib().lrf_render_bwd_wait(5, C.c_void_p(side.cuda_stream)) != 0
    assert b"bucket" in N.lib().lrf_last_error()


def test_captured_iteration_matches_the_eager_loop():
    """the training iteration without the host in it.  scripts/train_synth.py with graph=True runs
    every iteration as one replayed hipGraph (localrf_amd/graph_step.py: pose assembly, rays, field forward, losses, backward,
    all Adam launches, layout refresh), re-captured at every lifecycle event; the same seeds through the eager loop must
    give the same trajectory -- the photometric loss of every iteration (the scatter kernels' atomics make even two eager
    runs differ in the last bits, and a training trajectory amplifies that: early iterations tight, the end loose), the same
    lifecycle events, and nearly all iterations must have been replays."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import train_synth
    kw = dict(frames=9, final=80, iters_per_frame=30, n_max_frames=5, dev=DEV, geo_every=5, record_all=True)
    # ... and with the intrinsics tuned (lr_i_init > 0: focal_offset / center_rel join the captured Adam launches while the first
    # field refines, local_tensorfs.py:218-222): the first iterations, where the two trajectories have not drifted apart yet
    e_i = train_synth.run(graph=False, lr_i_init=1e-3, max_iters=150, **kw)
    g_i = train_synth.run(graph=True, lr_i_init=1e-3, max_iters=150, **kw)
    ai, bi = np.array(e_i["all_losses"]), np.array(g_i["all_losses"])
    assert ai.shape == bi.shape and np.abs(ai[:25] - bi[:25]).max() <= 5e-3 * np.abs(ai[:25]).max(), np.abs(ai[:25] - bi[:25]).max()
    assert np.abs(ai[:60] - bi[:60]).max() <= 5e-2 * np.abs(ai[:60]).max(), np.abs(ai[:60] - bi[:60]).max()
    assert abs(ai[-20:].mean() - bi[-20:].mean()) <= 0.25 * ai[-20:].mean()
    assert g_i["graph"]["plans_with_intrinsics"] >= 1, g_i["graph"]          # the refining phase of the first field was reached
    eager = train_synth.run(graph=False, **kw)
    graph = train_synth.run(graph=True, **kw)
    assert graph["iterations"] == eager["iterations"] and graph["events"] == eager["events"], (graph["events"], eager["events"])
    assert graph["fields"] == eager["fields"] and graph["frames"] == eager["frames"] and graph["final_resolution"] == eager["final_resolution"]
    a, b = np.array(eager["all_losses"]), np.array(graph["all_losses"])
    assert a.shape == b.shape and np.isfinite(b).all()
    assert np.abs(a[:10] - b[:10]).max() <= 2e-5 * np.abs(a[:10]).max(), (a[:10], b[:10])
    assert np.abs(a[:25] - b[:25]).max() <= 5e-3 * np.abs(a[:25]).max(), np.abs(a[:25] - b[:25]).max()
    assert np.abs(a[:60] - b[:60]).max() <= 5e-2 * np.abs(a[:60]).max(), np.abs(a[:60] - b[:60]).max()       # (seen once in ~15 runs: 5e-3 exceeded by iteration 60)
    assert abs(a[-20:].mean() - b[-20:].mean()) <= 0.25 * a[-20:].mean(), (a[-20:].mean(), b[-20:].mean())
    st = graph["graph"]
    assert st["replays"] + st["eager"] == graph["iterations"] and st["captures"] >= 3, st
    assert st["replays"] >= 0.6 * graph["iterations"], st
    assert graph["checkpoint_roundtrip"] and graph["geometric_losses"]["iterations_with_them"] > 0


def test_photometric_loss_and_row_gather_vs_torch_expressions():
    """lrf_photo_loss_* against train.py:369-371 written in torch (value and d/d rgb, with and without weights, with a
    supplied batch-global mean); lrf_rows_gather* against torch.stack(params)[view_ids] with repeated and negative ids
    (local_tensorfs.py:292-299,496) -- values bit-identical, gradients to fp32 summation order."""
    from localrf_amd.losses import photometric_loss
    from localrf_amd.scene_ops import rows_gather
    g = torch.Generator().manual_seed(5)
    R = 4096
    rgb = torch.rand(R, 3, generator=g).to(DEV).requires_grad_(True)
    tgt = torch.rand(R, 3, generator=g).to(DEV)
    with torch.no_grad():
        tgt[:7] = rgb[:7]                                           # exact zeros of |.|: sign(0) = 0 in torch's backward
    w = (0.1 + 3 * torch.rand(R, 1, generator=g)).to(DEV)
    wm = torch.tensor(1.7, device=DEV)
    for weights, mean in ((None, None), (w, None), (w, wm)):
        ref_w = torch.ones(R, 1, device=DEV) if weights is None else weights
        ref = (0.25 * torch.abs(rgb - tgt) * ref_w / (ref_w.mean() if mean is None else mean)).mean()
        (g_ref,) = torch.autograd.grad(ref * 3.0, rgb)
        out = photometric_loss(rgb, tgt, weights, mean)
        (g_out,) = torch.autograd.grad(out * 3.0, rgb)
        assert abs(float(out) - float(ref)) <= 2e-6 * abs(float(ref)), (float(out), float(ref))
        assert float((g_out - g_ref).abs().max()) <= 2e-6 * float(g_ref.abs().max())
        assert float(g_out[:7].abs().max()) == 0.0
    src = torch.randn(23, 3, 4, generator=g).to(DEV).requires_grad_(True)
    idx = torch.tensor([0, 5, 5, 22, -1, 7, 5, 0], device=DEV)
    up = torch.randn(8, 3, 4, generator=g).to(DEV)
    a = rows_gather(src, idx)
    b = src[idx]
    assert torch.equal(a, b)
    (ga,) = torch.autograd.grad((a * up).sum(), src)
    (gb,) = torch.autograd.grad((b * up).sum(), src)
    assert float((ga - gb).abs().max()) <= 1e-6 * float(gb.abs().max())
    assert torch.equal(ga[1], torch.zeros(3, 4, device=DEV))


def test_two_rank_progressive_loop_with_captured_iterations(tmp_path):
    """configs[4]'s ray shard with the captured iteration: two ranks (gloo: both on this GPU) run scripts/train_synth.py
    --graph under torch.distributed.run -- per iteration a forward + backward graph, the gradient exchange
    (localrf_amd.dist.allreduce_grads through LocalTensorfs.grad_sync) on the same stream, and the Adam graph.  The loss must
    fall as in the one-rank run, nearly every iteration must be a replay, and the replicas must hold IDENTICAL parameters at
    the end (max |p - p of rank 0| over ranks = 0 for the field, the poses and everything else: every rank applied the same
    reduced gradients -- with the regulariser in the loss (strayed density gradients) and with views only the other rank
    sampled, the two ways the captured path let replicas drift apart before round 6)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "scripts", "train_synth.py"), "--graph", "--backend", "gloo",
           "--frames", "7", "--final", "90", "--iters-per-frame", "40", "--n-max-frames", "5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["world"] == 2 and d["finite"] and d["iterations"] > 150
    assert d["loss_last"] < 0.5 * d["loss_first"], (d["loss_first"], d["loss_last"])
    st = d["graph"]
    assert st["replays"] >= 0.8 * d["iterations"] and st["captures"] >= 3, st
    assert d["checkpoint_roundtrip"]
    assert d["replica_divergence"] == {"field": 0.0, "poses": 0.0, "other": 0.0}, d["replica_divergence"]


@pytest.mark.gpu
def test_batch_and_loss_assembly_kernels_vs_torch_expressions(built_lib):
    """lrf_batch_gather against the reference's tensor indexing and mask expressions (train.py:352-358,385-420: values
    bit-identical, negative view ids included); lrf_loss_combine_* and the per-view forms of flow_loss / depth_loss against
    the scalar chain of train.py:425-437 written in torch: same total (fp32 summation order), same gradients in every leaf."""
    from localrf_amd import losses
    g = torch.Generator().manual_seed(11)
    F_, HW, V, n = 9, 640, 5, 37
    images, fwd, bwd = torch.rand(F_, HW, 3, generator=g).to(DEV), torch.randn(F_, HW, 2, generator=g).to(DEV), torch.randn(F_, HW, 2, generator=g).to(DEV)
    inv = torch.rand(F_, HW, generator=g).to(DEV)
    views = torch.tensor([0, 3, 8, -1, 4], device=DEV)
    pix = torch.randint(0, HW, (V, n), generator=g).to(DEV)
    rows = losses.batch_gather(views, pix, images=images, fwd_flow=fwd, bwd_flow=bwd, invdepths=inv)
    va = views % F_
    assert torch.equal(rows["target"], images[va[:, None], pix].reshape(-1, 3))
    assert torch.equal(rows["fwd_flow"], fwd[va[:, None], pix].reshape(-1, 2)) and torch.equal(rows["bwd_flow"], bwd[va[:, None], pix].reshape(-1, 2))
    assert torch.equal(rows["invdepths"], inv[va[:, None], pix].reshape(-1))
    assert torch.equal(rows["fwd_mask"], (va < F_ - 1).float()[:, None].expand(V, n).reshape(-1))
    assert torch.equal(rows["bwd_mask"], (va > 0).float()[:, None].expand(V, n).reshape(-1))
    only = losses.batch_gather(views, pix, images=images)
    assert set(only) == {"target"} and torch.equal(only["target"], rows["target"])
    # combine: scalars and vectors of partial sums, weights a + b s
    s = torch.tensor(0.37, device=DEV)
    base = [torch.randn((), generator=g), torch.randn(V, generator=g), torch.randn(V, generator=g), torch.rand((), generator=g)]
    coef = [(1.0, 0.0), (0.0, 1.0 / 56 / (V * n)), (0.0, 0.1 / (V * n)), (1e-2, 0.0)]
    xs = [b.to(DEV).requires_grad_(True) for b in base]
    total = losses.combine([(x, a, b) for x, (a, b) in zip(xs, coef)], s)
    gs = torch.autograd.grad(total * 2.5, xs)
    ys = [b.to(DEV).requires_grad_(True) for b in base]
    ref = sum(y.sum() * (a + b * s) for y, (a, b) in zip(ys, coef))
    gr = torch.autograd.grad(ref * 2.5, ys)
    assert abs(float(total) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    for a_, b_ in zip(gs, gr):
        assert a_.shape == b_.shape and float((a_ - b_).abs().max()) <= 1e-6 * max(1e-6, float(b_.abs().max()))
    # per-view sums through combine == the means through the scalar chain, values and gradients
    from util import load_golden
    gg = load_golden("geo_losses")
    res = {}
    for form in ("mean", "per_view"):
        a = _geo_inputs(gg)
        d = torch.from_numpy(gg["depth"]).to(DEV).requires_grad_(True)
        Vg = int(d.shape[0]); ng = int(d.numel() // Vg)
        if form == "mean":
            fl = losses.flow_loss(**a)
            dl = losses.depth_loss(d, torch.from_numpy(gg["invdepths"]).to(DEV), Vg)
            tot = fl * (s / 56) + dl * (0.1 * s)
        else:
            fl = losses.flow_loss(per_view=True, **a)
            dl = losses.depth_loss(d, torch.from_numpy(gg["invdepths"]).to(DEV), Vg, per_view=True)
            assert fl.shape == (Vg,) and dl.shape == (Vg,)
            tot = losses.combine([(fl, 0.0, 1.0 / 56 / (Vg * ng)), (dl, 0.0, 0.1 / (Vg * ng))], s)
        tot.backward()
        res[form] = (float(tot), a["depth_map"].grad.clone(), a["cam2world"].grad.clone(), d.grad.clone())
    m, p = res["mean"], res["per_view"]
    assert abs(m[0] - p[0]) <= 2e-6 * abs(m[0])
    for x, y in zip(m[1:], p[1:]):
        assert float((x - y).abs().max()) <= 2e-6 * float(x.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("grid", [(24, 20, 28), (150, 131, 140)])
def test_adam_step_fused_with_the_layout_refresh(built_lib, grid):
    """lrf_adam_step_pack (FusedAdam(pack_field=field)): parameters and Adam state bit-identical to the plain fused step, the
    layout cache it leaves bit-identical to what lrf_pack_field builds from the stepped parameters (ragged widths: 131 = one
    full 128-texel block + 3), a tensor without gradient repacked but not stepped, the next forward takes the cache as it is
    (no repack) and renders what the two-pass path renders; a later in-place edit still invalidates the cache."""
    import ctypes as C
    from localrf_amd import FusedAdam
    from localrf_amd import _native as N
    from util import make_field, make_rays, quiet
    fa = quiet(make_field, list(grid), "cpu", seed=5).to(DEV)
    fb = quiet(make_field, list(grid), "cpu", seed=5).to(DEV)
    oa = FusedAdam(fa.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    ob = FusedAdam(fb.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), pack_field=fb)
    rays = make_rays(96, 3).to(DEV)
    g = torch.Generator().manual_seed(9)
    for it in range(3):
        grads = {n: torch.randn(p.shape, generator=g).to(DEV) * 0.1 for n, p in fa.named_parameters()}
        for f in (fa, fb):
            with torch.no_grad():
                out = f(rays, N_samples=64)                     # builds / takes the cache
            for n, p in f.named_parameters():
                p.grad = None if (n == "density_line.1" and it == 1) else grads[n].clone()
        key_before = fb._cache_key
        oa.step()
        ob.step()
        assert fb._cache_key is not None and fb._cache_key != key_before          # marked fresh for the NEW parameter versions
        for (n, p), (_, q) in zip(fa.named_parameters(), fb.named_parameters()):
            assert torch.equal(p, q), n
            if p in oa.state and "exp_avg" in oa.state[p]:
                assert torch.equal(oa.state[p]["exp_avg"], ob.state[q]["exp_avg"]) and torch.equal(oa.state[p]["exp_avg_sq"], ob.state[q]["exp_avg_sq"]), n
        # the cache the fused step left == a fresh pack of the same parameters
        cp, keep = fb._c_params()
        fresh = fb._cache.clone()                                # (the alignment gaps between the cache's sections are nobody's: same bytes on both sides)
        N.check(built_lib.lrf_pack_field(C.byref(cp), fresh.data_ptr(), torch.cuda.current_stream().cuda_stream), "lrf_pack_field")
        torch.cuda.synchronize()
        assert torch.equal(fresh.view(torch.int32), fb._cache.view(torch.int32)), it
        key = fb._cache_key
        with torch.no_grad():
            ra, _ = fa(rays, N_samples=64)
            rb, _ = fb(rays, N_samples=64)
        assert fb._cache_key == key                              # the forward took the cache as the step left it
        assert torch.equal(ra, rb)
    with torch.no_grad():
        fb.app_plane[0].mul_(1.5)
        rc, _ = fb(rays, N_samples=64)
    assert fb._cache_key != key and not torch.equal(rb, rc)
