"""A short progressive-optimisation run of a LocalTensorfs scene, written against the reference's public interface only
(local_tensorfs.py: forward / get_reg_loss / optimizer_step / append_frame / append_rf; the loop of train.py:349-470
reduced to its control flow).  The SAME function drives
  * the real reference on CPU in tests/golden/make_golden.py::case_trajectory (recording every iteration), and
  * localrf_amd.LocalTensorfs on the GPU in tests/test_gpu_training.py (replaying the recorded sample distances),
so that what is compared is the two implementations, not two copies of a loop.

Timeline of the 30 iterations (events happen after the optimiser step of the iteration named):
  it  1, 3   append_frame                        4 -> 6 frames, n_added_frames = 2
  it  4      is_refining = True                  rf_iter starts to advance at it 5
  it  6      rf_iter == 1: n_iters = 3 x 6 = 18, n_iters_reg = 2 x 6 = 12, lr decay 0.1^(1/18), schedules scaled by 6
  it 11      rf_iter == 6: upsample_volume_grid to N_to_reso(18600) = 26^3 (18600 sits mid-way between 26^3 and 27^3:
             a cube number would put the reference's float -> long truncation on a knife edge) and a fresh Adam (lr_upsample_reset)
  it 14      rf_iter == 9: updateAlphaMask (alphaMask_thres chosen so that the mask really culls)
  it 17      rf_iter == 12: regularize turns off, the density L1 term leaves the loss
  it 21      rf_iter reaches 17 = n_iters - 1: can_add_rf -> append_rf(2); a new field trains the last two frames
  it 24      append_frame (linked to the new field)
  it 26      is_refining = True again; it 28 rescales the schedules for the 3 training frames of the new field
"""
import numpy as np
import torch

N_ITERS = 30
W, H = 24, 18
PER_VIEW = 48
N_VIEWS = 2
L1_WEIGHT = 0.01
APPEND_FRAME_AFTER = (1, 3, 24)
REFINE_AFTER = (4, 26)
GRID = (20, 20, 20)

SCENE_KW = dict(fov=85.6, n_init_frames=4, n_overlap=3, WH=(W, H), n_iters_per_frame=3, n_iters_reg=2,
                lr_R_init=5e-3, lr_t_init=5e-4, lr_i_init=1e-3, lr_exposure_init=1e-3, rf_lr_init=0.02,
                rf_lr_basis=5e-3, lr_decay_target_ratio=0.1, N_voxel_list={1: 18600}, update_AlphaMask_list=[1.5],
                camera_prior=None, lr_upsample_reset=True)
FIELD_OVER = dict(alphaMask_thres=7.9e-4)


def targets(n_frames=8):
    """Smooth synthetic frames [F, H*W, 3] in (0.1, 0.6): what the photometric loss pulls towards."""
    y, x = np.meshgrid(np.arange(H, dtype=np.float32) / H, np.arange(W, dtype=np.float32) / W, indexing="ij")
    out = np.empty((n_frames, H * W, 3), np.float32)
    for f in range(n_frames):
        for c in range(3):
            out[f, :, c] = (0.35 + 0.25 * np.sin(5.0 * x + 3.0 * y * (c + 1) + 0.7 * f + c)).reshape(-1)
    return out


def batches(seed):
    """Per iteration: N_VIEWS views (indices into the frames whose weight on the newest field is positive at that time,
    resolved by run()) and PER_VIEW pixel ids each.  Pure numpy so that both sides draw the same numbers."""
    g = np.random.default_rng(seed)
    return g.random((N_ITERS, N_VIEWS)).astype(np.float32), g.integers(0, W * H, (N_ITERS, N_VIEWS * PER_VIEW))


def run(lt, view_u, ray_ids, target, device, before_forward=None, after_append_rf=None, record=None):
    """Returns a list of per-iteration dicts.  `before_forward(lt, it)` / `after_append_rf(lt)` are the hooks the GPU
    replay uses to inject the recorded sample distances and the recorded initial state of the second field."""
    target = torch.from_numpy(target).to(device)
    n_added = 0
    log = []
    for it in range(N_ITERS):
        active = torch.nonzero(lt.blending_weights[:, -1] > 0)[:, 0].tolist()
        views = []
        for u in view_u[it]:                                       # distinct frames; a repeated draw moves on to the next one
            k = min(int(u * len(active)), len(active) - 1)
            while active[k] in views and len(views) < len(active):
                k = (k + 1) % len(active)
            if active[k] not in views:
                views.append(active[k])
        views.sort()
        view_ids = torch.tensor(views, device=device)
        ids = torch.from_numpy(ray_ids[it][:len(views) * PER_VIEW].astype(np.int64)).to(device)
        if before_forward is not None:
            before_forward(lt, it)
        rgb, depth, dirs, ij = lt(ids, view_ids, W, H, is_train=True, white_bg=True)
        want = torch.cat([target[v][ids[k * PER_VIEW:(k + 1) * PER_VIEW]] for k, v in enumerate(views)], 0)
        photo = (0.25 * torch.abs(rgb - want)).mean()              # train.py:369-371 with unit loss weights
        total = photo
        l1 = torch.zeros((), device=device)
        regularize = bool(lt.regularize)
        if regularize:                                             # train.py:425-427, TV weights 0
            _, l1 = lt.get_reg_loss(None, 0, 0, L1_WEIGHT)
            total = total + l1
        rec = dict(it=it, views=views, photo=float(photo.detach()), l1=float(torch.as_tensor(l1).detach()),
                   total=float(total.detach()), regularize=regularize, rf_iter=int(lt.rf_iter[-1]),
                   n_fields=len(lt.tensorfs), grid=[int(v) for v in lt.tensorfs[-1].gridSize],
                   nSamples=int(lt.tensorfs[-1].nSamples), has_mask=lt.tensorfs[-1].alphaMask is not None,
                   rgb=rgb.detach().cpu().numpy().copy(), depth=depth.detach().cpu().numpy().copy())
        can_add_rf = lt.optimizer_step(total, optimize_poses=True)
        rec["can_add_rf"] = bool(can_add_rf)
        if record is not None:
            record(lt, it, rec)
        log.append(rec)
        if it in REFINE_AFTER:
            lt.is_refining = True
        if it in APPEND_FRAME_AFTER:
            lt.append_frame()
            n_added += 1
        if can_add_rf and len(lt.tensorfs) == 1:                   # train.py:463-466
            lt.append_rf(n_added)
            n_added = 0
            if after_append_rf is not None:
                after_append_rf(lt)
    return log


def final_render(lt, device):
    """Eval-mode render of fixed pixels of frames 1, 4 and 6 through the blended fields: the function-space check of the
    final state (element-wise parameter parity is not a meaningful bar for an Adam trajectory, see case_trajectory)."""
    g = np.random.default_rng(7)
    ids = torch.from_numpy(g.integers(0, W * H, 3 * 64)).to(device)
    views = torch.tensor([1, 4, 6], device=device)
    with torch.no_grad():
        rgb, depth, _, _ = lt(ids, views, W, H, is_train=False, white_bg=True)
    return rgb.cpu().numpy(), depth.cpu().numpy()
