"""CPU-side checks: the C-ABI library loads and exports every symbol include/lrf.h
declares; host logic (constructors, state-dict surface, z schedule, loud failure without a
GPU).  No compute call is issued here."""
import os
import re

import numpy as np
import pytest
import torch

from util import FIELD_KW, load_golden, make_field, quiet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built_lib):
    from localrf_amd import _native
    hdr = open(os.path.join(ROOT, "include", "lrf.h")).read() + open(os.path.join(ROOT, "include", "lrf_debug.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(lrf_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_native.SYMBOLS), declared ^ set(_native.SYMBOLS)
    for name in declared:
        assert getattr(built_lib, name) is not None
    assert built_lib.lrf_abi_version() == 7


def test_cache_and_workspace_sizes(built_lib):
    import ctypes as C
    grid = (C.c_int32 * 3)(300, 300, 300)
    n = built_lib.lrf_cache_bytes(grid)
    # 3 planes x (8 + 32 + 24: appearance texels once padded to 128 B for the 16-sample / training kernels and once dense
    # for k_shade3) ch x 300^2 + lines + MLP images, fp32
    assert 3 * 64 * 300 * 300 * 4 <= n <= 3 * 64 * 300 * 300 * 4 + 800_000
    # lists + partials (12.6 MB) + the k_app -> k_mlp fragment buffer at its worst case (every sample
    # shaded: 4096 x 32 tiles x 2 KB = 268 MB; the benchmark touches 99 MB of it)
    assert built_lib.lrf_workspace_bytes(4096, 512) < 300 << 20


def test_state_dict_surface_matches_reference():
    g = load_golden("field_small_eval")
    ref_keys = {k[2:]: v.shape for k, v in g.items() if k.startswith("f.")}
    f = quiet(make_field, [int(v) for v in g["grid"]])
    mine = {k: tuple(v.shape) for k, v in f.state_dict().items()}
    assert mine == {k: tuple(s) for k, s in ref_keys.items()}
    # optimiser groups: order is read by index in train.py:480,485
    groups = f.get_optparam_groups(0.02, 1e-3)
    assert [g_["lr"] for g_ in groups] == [0.02] * 4 + [1e-3] * 2
    assert len(list(groups[5]["params"])) == 6
    kw = f.get_kwargs()
    for k in FIELD_KW:
        if k not in ("alphaMask_thres",):
            assert k in kw


def test_nsamples_follows_grid():
    # 64^3 -> 72 samples/ray, 300^3 -> 344
    assert 2 * (quiet(make_field, [64] * 3).nSamples // 6) == 72
    f = quiet(make_field, [300] * 3)
    assert 2 * (f.nSamples // 6) == 344
    g = load_golden("field_small_default_ns")
    assert quiet(make_field, [int(v) for v in g["grid"]]).nSamples == int(g["nSamples"])


def test_z_schedule_eval_and_train():
    g = load_golden("field_small_eval")
    f = quiet(make_field, [int(v) for v in g["grid"]])
    z = f.z_schedule(False, int(g["N_samples"]), torch.device("cpu"))
    assert np.abs(z.numpy() - g["z"]).max() == 0.0
    gt = load_golden("field_small_train_grad")
    n = int(gt["N_samples"])
    torch.manual_seed(23)                    # make_golden: seed + 2 with seed = 21
    zt = f.z_schedule(True, n, torch.device("cpu")).numpy()
    h = n // 6
    t = np.arange(h, dtype=np.float32) / np.float32(h)
    a = t + gt["U"] / np.float32(h)
    b = 1.0 / ((1.0 - (t + gt["U2"] / np.float32(h))) + (t + gt["U2"] / np.float32(h)) / 1000.0)
    assert np.abs(zt - (np.concatenate([a, b]) + 0.1)).max() < 1e-6


def test_unsupported_configs_fail_loudly():
    with pytest.raises(NotImplementedError):
        quiet(make_field, [16] * 3, shadingMode="MLP_PE")
    with pytest.raises(NotImplementedError):
        quiet(make_field, [16] * 3, fea_pe=7)
    with pytest.raises(NotImplementedError):
        quiet(make_field, [16] * 3, featureC=512)
    f = quiet(make_field, [16] * 3, fea_pe=2, view_pe=3, featureC=64)          # the generic engine's shapes (tensorBase.py:97-113)
    assert tuple(f.renderModule.mlp[0].weight.shape) == (64, 27 * 5)
    assert tuple(f.renderModule.mlp_view[0].weight.shape) == (3, 64 + 3 * 7)
    with pytest.raises(NotImplementedError):
        quiet(make_field, [16] * 3, density_n_comp=[16, 16, 16])


def test_forward_without_gpu_raises_not_falls_back():
    from localrf_amd._native import NativeError
    f = quiet(make_field, [16] * 3)
    rays = torch.randn(8, 6)
    with pytest.raises(NativeError):
        f(rays)
    with pytest.raises(NativeError):
        f.compute_densityfeature(torch.zeros(4, 3))


def test_local_tensorfs_surface_and_checkpoint_roundtrip(tmp_path):
    from localrf_amd import LocalTensorfs
    g = load_golden("local_4fields")
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]])

    def build():
        return quiet(LocalTensorfs, fov=85.6, n_init_frames=5, n_overlap=3, WH=(32, 24),
                     n_iters_per_frame=600, n_iters_reg=100, lr_R_init=5e-3, lr_t_init=5e-4,
                     lr_i_init=0, lr_exposure_init=1e-3, rf_lr_init=0.02, rf_lr_basis=1e-3,
                     lr_decay_target_ratio=0.1, N_voxel_list={}, update_AlphaMask_list=[],
                     camera_prior=None, device="cpu", lr_upsample_reset=True,
                     aabb=aabb, gridSize=[16, 16, 16], **FIELD_KW)
    lt = build()
    for _ in range(3):
        for _ in range(3):
            lt.append_frame()
        quiet(lt.append_rf, 3)
    ref = {k[3:]: v for k, v in g.items() if k.startswith("lt.")}
    mine = lt.state_dict()
    assert set(mine) == set(ref)
    for k in ref:
        assert tuple(mine[k].shape) == tuple(ref[k].shape), k
    # blending weights produced by append_frame/append_rf are data-independent: must match
    assert np.abs(mine["blending_weights"].numpy() - ref["blending_weights"]).max() < 1e-6
    # reference checkpoint loads through .load(): regrows fields and frames
    lt2 = build()
    quiet(lt2.load, {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in ref.items()})
    assert len(lt2.tensorfs) == 4 and len(lt2.r_c2w) == 14
    p = tmp_path / "ckpt.th"
    lt2.save(str(p))
    ck = torch.load(str(p), weights_only=False)
    assert set(ck["state_dict"]) == set(ref)
    # host-side ray setup matches the reference (dirs, ij)
    from localrf_amd.rays import get_ray_directions_lean, ids2pixel
    col, row = ids2pixel(32, 24, torch.from_numpy(g["ray_ids"]))
    d = get_ray_directions_lean(col, row, lt2.focal(32), lt2.center(32, 24)).detach().numpy()
    assert np.abs(d - g["dirs"]).max() < 1e-6
    assert (torch.stack([col, row], -1).numpy() == g["ij"]).all()


def test_fused_adam_host_logic_and_no_cpu_fallback():
    """Constructor checks, param_groups / state layout of torch.optim.Adam, cheap zero_grad, and no
    CPU arithmetic: stepping host tensors raises."""
    from localrf_amd import FusedAdam, LocalTensorfs, NativeError
    p = torch.nn.Parameter(torch.zeros(4))
    q = torch.nn.Parameter(torch.ones(2, 3))
    opt = FusedAdam([{"params": [p], "lr": 0.02}, {"params": [q]}], lr=1e-3, betas=(0.9, 0.99))
    assert [g["lr"] for g in opt.param_groups] == [0.02, 1e-3]
    assert all(g["betas"] == (0.9, 0.99) and g["eps"] == 1e-8 for g in opt.param_groups)
    for bad in (dict(weight_decay=0.1), dict(amsgrad=True), dict(lr=-1.0), dict(betas=(1.0, 0.9))):
        with pytest.raises(ValueError):
            FusedAdam([torch.nn.Parameter(torch.zeros(1))], **bad)
    opt.step()                                   # nothing has a gradient: no launch, no error
    p.grad = torch.ones(4)
    with pytest.raises(NativeError):
        opt.step()
    with pytest.raises(NativeError):
        FusedAdam.step_many([opt])
    opt.zero_grad()
    assert p.grad is None
    p.grad = torch.ones(4)
    opt.zero_grad(set_to_none=False)
    assert p.grad is not None and float(p.grad.abs().sum()) == 0.0
    # a torch.optim.Adam state dict (tensor step counters) loads, and ours loads back into torch's
    ref = torch.optim.Adam([{"params": [p], "lr": 0.02}, {"params": [q]}], lr=1e-3, betas=(0.9, 0.99))
    p.grad, q.grad = torch.ones(4), torch.ones(2, 3)
    ref.step()
    opt.load_state_dict(ref.state_dict())
    assert int(opt.state[p]["step"]) == 1 and torch.equal(opt.state[p]["exp_avg"], ref.state[p]["exp_avg"])
    ref.load_state_dict(opt.state_dict())
    # the scene creates FusedAdam objects everywhere the reference creates torch.optim.Adam
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    lt = quiet(LocalTensorfs, fov=85.6, n_init_frames=3, n_overlap=3, WH=(32, 24),
               n_iters_per_frame=600, n_iters_reg=100, lr_R_init=5e-3, lr_t_init=5e-4,
               lr_i_init=1e-3, lr_exposure_init=1e-3, rf_lr_init=0.02, rf_lr_basis=1e-3,
               lr_decay_target_ratio=0.1, N_voxel_list={}, update_AlphaMask_list=[],
               camera_prior=None, device="cpu", lr_upsample_reset=True,
               aabb=aabb, gridSize=[16, 16, 16], **FIELD_KW)
    opts = [lt.rf_optimizer, lt.intrinsic_optimizer] + lt.r_optimizers + lt.t_optimizers + lt.exp_optimizers
    assert len(opts) == 2 + 3 * 3 and all(isinstance(o, FusedAdam) for o in opts)
    assert [g["lr"] for g in lt.rf_optimizer.param_groups][:2] == [0.02, 0.02]


def test_c_abi_argument_validation_without_a_gpu():
    """Entry points reject bad arguments with a non-zero code and a message (no kernel is launched
    on these paths, so this runs on the CPU-only container)."""
    import ctypes as C
    from localrf_amd import _native as N
    lib = N.lib()
    st = None
    assert lib.lrf_adam_step(None, 0, 0.9, 0.99, 1e-8, st) == 0                  # empty table: no-op
    assert lib.lrf_adam_step(None, 3, 0.9, 0.99, 1e-8, st) != 0
    assert b"lrf_adam_step" in lib.lrf_last_error()
    assert lib.lrf_adam_step(None, N.LRF_ADAM_MAX + 1, 0.9, 0.99, 1e-8, st) != 0
    assert lib.lrf_render_fwd(None, None, None, 4, 8, 0, 0.0, None, None, None, None, None, st) != 0
    assert b"null argument" in lib.lrf_last_error()
    assert lib.lrf_scene_rays(None, 4, 2, None, None, 1, None, None, 8, 8, 0, None, None, None, st) != 0
    assert lib.lrf_scene_blend(None, None, None, None, 4, 2, 1, None, None, None, st) != 0
    assert lib.lrf_pose_assemble(None, None, 1, 0, None, st) != 0
    hw = (C.c_int32 * 3)(64, 64, 64)
    ll = (C.c_int32 * 3)(8, 8, 9)                                              # planes x lines spanning different lattices
    assert lib.lrf_density_l1_fwd(None, None, hw, ll, -5.0, 0, None, None, st) != 0
    assert lib.lrf_tv_loss_fwd(None, 0, 1.0, None, None, st) != 0
    assert lib.lrf_tv_workspace(None, 0) == 0
    grid = (C.c_int32 * 3)(300, 300, 300)
    assert lib.lrf_workspace_bytes_bwd(4096, 512, grid) > lib.lrf_workspace_bytes(4096, 512) > 0
    assert lib.lrf_cache_bytes(grid) >= 34_800_000
    # the generic engine's operand rows: only for a non-default colour network, or the exact-fp32 training path (LRF_FLAG_MLP_VALU = 4)
    base = lib.lrf_workspace_bytes_bwd(4096, 512, grid)
    assert lib.lrf_workspace_bytes_bwd_cfg(4096, 512, grid, 0, 0, 128, 0) == base == lib.lrf_workspace_bytes_bwd_cfg(4096, 512, grid, 0, 0, 0, 1)
    pe = lib.lrf_workspace_bytes_bwd_cfg(4096, 512, grid, 2, 2, 128, 0)
    assert pe > base and lib.lrf_workspace_bytes_bwd_cfg(4096, 512, grid, 0, 0, 128, 4) > base
    rows = 4096 * 512
    ld = 2 * 128 + (27 * 5 + 1) + 129 + (128 + 15 + 1) + 4
    assert 0 <= pe - base - rows * ld * 4 < 4096


def test_saved_row_layout_is_a_bijection_and_matches_the_test_reader(built_lib):
    """The training workspace keeps activation / gradient rows in MFMA-fragment order (csrc/lrf_common.h frag_off).
    Host side of the same functions the kernels index with: every (row, column) of a few tiles maps to a distinct
    float inside its tile, a producer lane's four columns are adjacent, the 64 lanes of a block are one contiguous
    1 KB, the dX block is row-major -- and the
    permute that tests/util.py::relu_flip_report uses to read the rows agrees with it."""
    off = built_lib.lrf_debug_saved_row_offset
    for buf, ld in ((0, 32), (1, 128)):
        assert off(buf, 0, ld) == -1 and off(buf, 0, -1) == -1
        rows = list(range(0, 48)) + [16 * 1000 + 5]
        seen = set()
        for r in rows:
            for c in range(ld):
                o = off(buf, r, c)
                assert (r // 16) * 16 * ld <= o < (r // 16 + 1) * 16 * ld          # inside the row's tile
                assert o not in seen
                seen.add(o)
        nfrag = ld if buf == 0 else 48                                             # GRD: columns 48.. are the dX block
        for c in range(0, nfrag, 4):                                               # a lane's float4
            assert [off(buf, 7, c + k) - off(buf, 7, c) for k in range(4)] == [0, 1, 2, 3]
        for blk in range(nfrag // 16):                                             # a wave's store: lanes (s, g) -> 1 KB
            base = off(buf, 0, 16 * blk)
            got = sorted(off(buf, s, 16 * blk + 4 * g) - base for s in range(16) for g in range(4))
            assert got == list(range(0, 256, 4))
    assert [off(1, 3, 48 + k) - off(1, 3, 48) for k in range(80)] == list(range(80))     # dX: row-major inside the tile
    assert off(1, 4, 48) - off(1, 3, 48) == 80
    assert off(2, 0, 0) == -1                                                      # (the X block is gone: round 4)
    # the reader in tests/util.py: view(tiles, LD/16, 4, 16, 4).permute(0, 3, 1, 2, 4).reshape(rows, LD)
    ld, tiles = 32, 3
    flat = np.arange(tiles * 16 * ld)
    rowsv = flat.reshape(tiles, ld // 16, 4, 16, 4).transpose(0, 3, 1, 2, 4).reshape(tiles * 16, ld)
    for r in (0, 5, 17, 47):
        for c in (0, 3, 4, 15, 16, 27, 28, 31):
            assert rowsv[r, c] == off(0, r, c)


def test_geometric_loss_wrappers_fail_loudly_without_a_gpu_and_check_their_arguments():
    """localrf_amd.losses (train.py:385-423): no CPU fallback; the per-view rays limit and the view-id range are checked
    on the host before anything is launched."""
    from localrf_amd import _native as N
    from localrf_amd import losses
    V, n = 2, 8
    depth = torch.rand(V * n) + 0.5
    dirs = torch.randn(V * n, 3)
    ij = torch.zeros(V * n, 2, dtype=torch.int64)
    c2w = torch.eye(3, 4)[None].repeat(4, 1, 1)
    flow, mask = torch.zeros(V * n, 2), torch.ones(V * n)
    with pytest.raises(N.NativeError, match="no CPU fallback"):
        losses.flow_loss(depth, dirs, ij, c2w, [1, 2], 0, flow, mask, flow, mask, 50.0, torch.tensor([32.0, 24.0]))
    with pytest.raises(N.NativeError, match="no CPU fallback"):
        losses.depth_loss(depth, torch.rand(V * n), V)
    assert N.LRF_LOSS_MAX_PER_VIEW == 4096                       # LOSS_NMAX of csrc/lrf_losses.inl: one workgroup sorts a view in LDS
    hdr = open(os.path.join(ROOT, "localrf_amd", "csrc", "lrf_losses.inl")).read()
    assert re.search(r"LOSS_NMAX\s*=\s*4096", hdr)
    meta = torch.empty(V * (N.LRF_LOSS_MAX_PER_VIEW + 1), device="meta")
    for call in (lambda: losses.depth_loss(_FakeCuda(meta), meta, V),
                 lambda: losses.flow_loss(_FakeCuda(meta), dirs, ij, c2w, [0, 1], 0, flow, mask, flow, mask, 50.0, torch.zeros(2))):
        with pytest.raises(ValueError, match="rays per view"):
            call()
    ok = _FakeCuda(torch.empty(V * n, device="meta"))
    with pytest.raises(IndexError, match="outside cam2world"):   # ids are absolute, cam2world starts at starting_frame_id
        losses.flow_loss(ok, dirs, ij, c2w, [3, 9], 3, flow, mask, flow, mask, 50.0, torch.zeros(2))
    with pytest.raises(IndexError, match="outside cam2world"):
        losses.flow_loss(ok, dirs, ij, c2w, [2, 3], 3, flow, mask, flow, mask, 50.0, torch.zeros(2))


class _FakeCuda:
    """A tensor stand-in whose device says cuda: lets the host-side argument checks of the wrappers run where there is
    no GPU (they raise before any pointer is taken)."""

    def __init__(self, t):
        self._t = t
        self.device = torch.device("cuda:0")

    def reshape(self, *shape):
        return _FakeCuda(self._t.reshape(*shape))

    @property
    def shape(self):
        return self._t.shape


def test_untaped_chunk_honours_the_caller_beyond_the_workspace_bound(built_lib):
    """LocalTensorfs renders up to min_chunk rays per field call when no gradient is recorded, whatever the caller's chunk --
    but only while that call's workspace stays under max_untaped_workspace; a caller that bounds memory with `chunk` gets its
    bound back once the raised chunk would exceed it."""
    from localrf_amd.scene import LocalTensorfs

    class F:                                                  # what _untaped_chunk reads of a field
        def __init__(self, n):
            self.nSamples = n
    s = LocalTensorfs.__new__(LocalTensorfs)
    s.min_chunk, s.max_untaped_workspace = 65536, 1 << 30
    assert s._untaped_chunk(4096, [F(2214)]) == 65536          # 640^3: 0.31 GiB per call, under the bound
    s.max_untaped_workspace = 100 << 20
    got = s._untaped_chunk(4096, [F(2214)])
    assert 4096 <= got < 65536 and built_lib.lrf_workspace_bytes(got, 738) <= 100 << 20
    s.max_untaped_workspace = 1 << 20
    assert s._untaped_chunk(4096, [F(2214)]) == 4096            # the caller's chunk, exactly
    s.min_chunk = 1
    assert s._untaped_chunk(4096, [F(2214)]) == 4096


def test_no_runtime_fill_or_copy_in_the_library_sources():
    """profiles/r18_memset_fault.md: a hipMemsetAsync in the backward filled the counting sort's histogram with a stale
    pattern -- another dispatch's kernel arguments -- instead of zeros (memory access fault in the captured progressive loop).  The
    library clears and moves its buffers with its own kernels; nothing in csrc may call the runtime's fill / copy."""
    src_dir = os.path.join(ROOT, "localrf_amd", "csrc")
    bad = []
    for name in sorted(os.listdir(src_dir)):
        if not name.endswith((".hip", ".inl", ".h")):
            continue
        text = re.sub(r"//[^\n]*", "", open(os.path.join(src_dir, name)).read())
        bad += [(name, m.group(0)) for m in re.finditer(r"\bhipMem(set\w*|cpy\w*Async)\s*\(", text)]   # (a blocking debug read-back is fine)
    assert not bad, bad
