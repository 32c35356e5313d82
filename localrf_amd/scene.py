"""LocalTensorfs -- self-calibrating scene of blended local radiance fields.

Drop-in for the reference's `local_tensorfs.LocalTensorfs` (same constructor, attribute and
state-dict names, `forward` signature/returns, `optimizer_step`, `append_frame`,
`append_rf`, `save`/`load`).  `forward` (local_tensorfs.py:382-499) is three kinds of HIP launch:
`lrf_scene_rays` (pixel ids -> per-field rays), the native field renderer
(`TensorVMSplit.forward`) per active field, and `lrf_scene_blend` (weighted sum over fields,
per-view exposure, clamp).  Poses, intrinsics and exposure get the gradients autograd derives
in the reference (scene_ops.py); the [V,3,2]->[V,3,4] pose assembly is `lrf_pose_assemble`.

MI355X-first differences (results identical):
  * finished fields stay resident in HBM (288 GB) instead of being parked on the host and
    shuttled back for every evaluation call (local_tensorfs.py:132,432-434,476-479);
  * no torch.cuda.empty_cache() between chunks (local_tensorfs.py:444-445).

Cited lines are relative to /root/reference/localTensoRF.
"""
import math
import re

import torch

from ._native import NativeError
from .field import AlphaGridMask, TensorVMSplit
from .rays import N_to_reso, mtx_to_sixD, sixD_to_mtx
from .optim import FusedAdam
from .scene_ops import pose_assemble, scene_blend, scene_forward, scene_rays

_ADAM_BETAS = (0.9, 0.99)


def _upload_ids(ids, dev):
    """Host ids -> device int64 without a blocking copy: staged in pinned memory from torch's
    caching host allocator (a block is not reused before the copy that reads it has run)."""
    src = torch.as_tensor(ids, dtype=torch.int64)
    stage = torch.empty(src.shape, dtype=torch.int64, pin_memory=True)
    stage.copy_(src)
    return stage.to(dev, non_blocking=True)


class LocalTensorfs(torch.nn.Module):
    def __init__(self, fov, n_init_frames, n_overlap, WH, n_iters_per_frame, n_iters_reg,
                 lr_R_init, lr_t_init, lr_i_init, lr_exposure_init, rf_lr_init, rf_lr_basis,
                 lr_decay_target_ratio, N_voxel_list, update_AlphaMask_list, camera_prior,
                 device, lr_upsample_reset, **tensorf_args):
        super().__init__()
        self.fov = fov
        self.n_init_frames = n_init_frames
        self.n_overlap = n_overlap
        self.W, self.H = WH
        self.n_iters_per_frame = n_iters_per_frame
        self.n_iters_reg_per_frame = n_iters_reg
        self.lr_R_init, self.lr_t_init = lr_R_init, lr_t_init
        self.lr_i_init, self.lr_exposure_init = lr_i_init, lr_exposure_init
        self.rf_lr_init, self.rf_lr_basis = rf_lr_init, rf_lr_basis
        self.lr_decay_target_ratio = lr_decay_target_ratio
        self.N_voxel_per_frame_list = N_voxel_list
        self.update_AlphaMask_per_frame_list = update_AlphaMask_list
        self.device = torch.device(device)
        self.camera_prior = camera_prior
        self.tensorf_args = tensorf_args
        self.is_refining = False
        self.lr_upsample_reset = lr_upsample_reset
        # utils/utils.py:386 calls torch.cross without `dim`, which picks the FIRST axis of size 3:
        # for a batch of exactly 3 views that is the view axis, not xyz.  True reproduces the
        # reference result (checked against a reference-recorded golden); False = the cross product
        # the reference meant.  Only batches of exactly 3 views differ.
        self.reference_cross = True
        # data-parallel hook (localrf_amd/dist.py): called between backward and the optimiser steps of
        # optimizer_step with this module; None = single process, as the reference
        self.grad_sync = None
        # lower bound of the rays per field call in forward (see there); 1 = chunk exactly as the reference
        self.min_chunk = 65536

        self.lr_factor = 1
        self.regularize = True
        self.n_iters_reg = self.n_iters_reg_per_frame
        self.n_iters = self.n_iters_per_frame
        self.update_AlphaMask_list = update_AlphaMask_list
        self.N_voxel_list = N_voxel_list

        # per-frame pose / exposure parameters, each with its own Adam (local_tensorfs.py:88-97)
        self.r_c2w = torch.nn.ParameterList()
        self.t_c2w = torch.nn.ParameterList()
        self.exposure = torch.nn.ParameterList()
        self.r_optimizers, self.t_optimizers, self.exp_optimizers = [], [], []
        self.pose_linked_rf = []
        self.blending_weights = torch.nn.Parameter(
            torch.ones([1, 1], device=self.device), requires_grad=False)
        for _ in range(n_init_frames):
            self.append_frame()

        if self.camera_prior is not None:                       # :99-104
            focal = self.camera_prior["transforms"]["fl_x"]
            focal *= self.W / self.camera_prior["transforms"]["w"]
        else:
            focal = self.W / math.tan(fov * math.pi / 180 / 2) / 2
        self.init_focal = torch.nn.Parameter(torch.Tensor([focal]).to(self.device))
        self.focal_offset = torch.nn.Parameter(torch.ones(1, device=device))
        self.center_rel = torch.nn.Parameter(0.5 * torch.ones(2, device=device))
        if lr_i_init > 0:
            self.intrinsic_optimizer = FusedAdam(
                [self.focal_offset, self.center_rel], betas=_ADAM_BETAS, lr=self.lr_i_init)

        self.tensorfs = torch.nn.ParameterList()
        self.rf_iter = []
        self.world2rf = torch.nn.ParameterList()
        self.append_rf()

    # ----------------------------------------------------------------- field / frame lifecycle
    def append_rf(self, n_added_frames=1):
        """Start a new local field centred on the last camera (local_tensorfs.py:116-146)."""
        self.is_refining = False
        if len(self.tensorfs) > 0:
            n_ov = min(n_added_frames, self.n_overlap, self.blending_weights.shape[0] - 1)
            ramp = 1 / n_ov + torch.arange(0, 1, 1 / n_ov)
            self.blending_weights.requires_grad = False
            self.blending_weights[-n_ov:, -1] = 1 - ramp
            fresh = torch.zeros_like(self.blending_weights[:, 0:1])
            fresh[-n_ov:, 0] = ramp
            self.blending_weights = torch.nn.Parameter(
                torch.cat([self.blending_weights, fresh], dim=1), requires_grad=False)
            world2rf = -self.t_c2w[-1].clone().detach()
            # reference parks the finished field on the CPU here (:132); with 288 GB of HBM it
            # stays resident.  Its optimiser is gone: drop its gradients (35-96 MB) with it.
            for p in self.tensorfs[-1].parameters():
                p.grad = None
            self.tensorfs[-1]._grad_flat = None
            self.tensorfs[-1]._grad_fresh = False
        else:
            world2rf = torch.zeros(3, device=self.device)
        self.tensorfs.append(TensorVMSplit(device=self.device, **self.tensorf_args))
        self.world2rf.append(world2rf.clone().detach())
        self.rf_iter.append(0)
        self.rf_optimizer = FusedAdam(
            self.tensorfs[-1].get_optparam_groups(self.rf_lr_init, self.rf_lr_basis), betas=_ADAM_BETAS)

    def append_frame(self):
        """New frame initialised from the previous pose (local_tensorfs.py:148-177)."""
        if len(self.r_c2w) == 0:
            self.r_c2w.append(torch.eye(3, 2, device=self.device))
            self.t_c2w.append(torch.zeros(3, device=self.device))
            self.pose_linked_rf.append(0)
        else:
            self.r_c2w.append(mtx_to_sixD(sixD_to_mtx(self.r_c2w[-1].clone().detach()[None]))[0])
            self.t_c2w.append(self.t_c2w[-1].clone().detach())
            self.blending_weights = torch.nn.Parameter(
                torch.cat([self.blending_weights, self.blending_weights[-1:, :]], dim=0),
                requires_grad=False)
            self.pose_linked_rf.append(int(torch.nonzero(self.blending_weights[-1, :])[0]))
        self.exposure.append(torch.eye(3, 3, device=self.device))
        if self.camera_prior is not None:
            idx = len(self.r_c2w) - 1
            rel = self.camera_prior["rel_poses"][idx]
            last_r = sixD_to_mtx(self.r_c2w[-1].clone().detach()[None])[0]
            self.r_c2w[-1] = last_r @ rel[:3, :3]
            self.t_c2w[-1].data += last_r @ rel[:3, 3]
        self.r_optimizers.append(FusedAdam([self.r_c2w[-1]], betas=_ADAM_BETAS, lr=self.lr_R_init))
        self.t_optimizers.append(FusedAdam([self.t_c2w[-1]], betas=_ADAM_BETAS, lr=self.lr_t_init))
        self.exp_optimizers.append(
            FusedAdam([self.exposure[-1]], betas=_ADAM_BETAS, lr=self.lr_exposure_init))

    # ----------------------------------------------------------------- optimisation
    def _active_pose_ids(self):
        last = len(self.rf_iter) - 1
        if self.rf_iter[-1] >= self.n_iters:
            return []
        return [i for i, rf in enumerate(self.pose_linked_rf) if rf == last]

    def optimizer_step_poses_only(self, loss):
        """local_tensorfs.py:179-191."""
        ids = self._active_pose_ids()
        for i in ids:
            self.r_optimizers[i].zero_grad()
            self.t_optimizers[i].zero_grad()
        loss.backward()
        if ids:
            FusedAdam.step_many([o for i in ids for o in (self.r_optimizers[i], self.t_optimizers[i])])

    def optimizer_step(self, loss, optimize_poses):
        """One optimisation step of the current field, its linked poses/exposures and the
        intrinsics, plus the scheduled upsample / alpha-mask rebuild (local_tensorfs.py:193-290)."""
        it = self.rf_iter[-1]
        if it == 0:
            self.lr_factor = 1
            self.n_iters = self.n_iters_per_frame
            self.n_iters_reg = self.n_iters_reg_per_frame
        elif it == 1:
            n_train = (self.blending_weights[:, -1] > 0).sum()
            self.n_iters = int(self.n_iters_per_frame * n_train)
            self.n_iters_reg = int(self.n_iters_reg_per_frame * n_train)
            self.lr_factor = self.lr_decay_target_ratio ** (1 / self.n_iters)
            self.N_voxel_list = {int(k * n_train): v for k, v in self.N_voxel_per_frame_list.items()}
            self.update_AlphaMask_list = [int(u * n_train) for u in self.update_AlphaMask_per_frame_list]
        self.regularize = self.rf_iter[-1] < self.n_iters_reg

        def decay(opt):
            for grp in opt.param_groups:
                grp["lr"] *= self.lr_factor

        pose_ids = self._active_pose_ids()
        for i in pose_ids:
            if optimize_poses:
                decay(self.r_optimizers[i]); decay(self.t_optimizers[i])
                self.r_optimizers[i].zero_grad(); self.t_optimizers[i].zero_grad()
            if self.lr_exposure_init > 0:
                decay(self.exp_optimizers[i])
                self.exp_optimizers[i].zero_grad()
        tune_intrinsics = (self.lr_i_init > 0 and self.blending_weights.shape[1] == 1
                           and self.is_refining)
        if tune_intrinsics:
            decay(self.intrinsic_optimizer)
            self.intrinsic_optimizer.zero_grad()
        self.rf_optimizer.zero_grad()

        loss.backward()
        if self.grad_sync is not None:          # data parallel: localrf_amd.dist.allreduce_grads(self)
            self.grad_sync(self)

        self.rf_optimizer.step()
        if self.is_refining:
            decay(self.rf_optimizer)

        if self.rf_iter[-1] in self.N_voxel_list:               # :251-261
            reso = N_to_reso(self.N_voxel_list[self.rf_iter[-1]], self.tensorfs[-1].aabb)
            self.tensorfs[-1].upsample_volume_grid(reso)
            if self.lr_upsample_reset:
                self.rf_optimizer = FusedAdam(
                    self.tensorfs[-1].get_optparam_groups(self.rf_lr_init, self.rf_lr_basis),
                    betas=_ADAM_BETAS)
        if self.rf_iter[-1] in self.update_AlphaMask_list:      # :264-266
            self.tensorfs[-1].updateAlphaMask(tuple((self.tensorfs[-1].gridSize / 2).int().tolist()))

        small = []                                              # :229-249, one launch for all of them
        for i in pose_ids:
            if optimize_poses:
                small += [self.r_optimizers[i], self.t_optimizers[i]]
            if self.lr_exposure_init > 0:
                small.append(self.exp_optimizers[i])
        if tune_intrinsics:
            small.append(self.intrinsic_optimizer)
        if small:
            FusedAdam.step_many(small)
        if self.is_refining:
            self.rf_iter[-1] += 1
        return self.rf_iter[-1] >= self.n_iters - 1             # can_add_rf

    # ----------------------------------------------------------------- accessors / io
    def get_cam2world(self, view_ids=None, starting_id=0):
        """[V,3,4] camera-to-world from the 6D rotation + translation params (:292-299)."""
        if view_ids is not None:
            ids = view_ids.tolist() if torch.is_tensor(view_ids) else list(view_ids)   # one sync, not one per view
            rp, tp = self.r_c2w._parameters, self.t_c2w._parameters      # ParameterList.__getitem__ costs ~3 us a piece
            n = len(rp)
            keys = [str(v if v >= 0 else v + n) for v in ids]
            r = [rp[k] for k in keys]
            t = [tp[k] for k in keys]
        else:
            r = list(self.r_c2w[starting_id:])
            t = list(self.t_c2w[starting_id:])
        # with camera priors append_frame stores full [3,3] rotations (local_tensorfs.py:171-176);
        # sixD_to_mtx reads columns 0 and 1 only (utils/utils.py:381-384): slice before the kernel so
        # autograd maps the [3,2] gradient back into the [3,3] parameter
        r = [x if x.shape[-1] == 2 else x[:, :2] for x in r]
        if r[0].is_cuda:                                        # one launch (per 64 frames) each way
            return pose_assemble(r, t, cross_over_views=self.reference_cross and len(r) == 3)
        return torch.cat([sixD_to_mtx(torch.stack(r, 0), self.reference_cross), torch.stack(t, 0)[..., None]], dim=-1)

    def get_kwargs(self):
        """local_tensorfs.py:301-324."""
        kw = {
            "camera_prior": None, "fov": self.fov, "n_init_frames": self.n_init_frames,
            "n_overlap": self.n_overlap, "WH": (self.W, self.H),
            "n_iters_per_frame": self.n_iters_per_frame, "n_iters_reg": self.n_iters_reg_per_frame,
            "lr_R_init": self.lr_R_init, "lr_t_init": self.lr_t_init, "lr_i_init": self.lr_i_init,
            "lr_exposure_init": self.lr_exposure_init, "rf_lr_init": self.rf_lr_init,
            "rf_lr_basis": self.rf_lr_basis, "lr_decay_target_ratio": self.lr_decay_target_ratio,
            "N_voxel_list": self.N_voxel_per_frame_list,
            "update_AlphaMask_list": self.update_AlphaMask_per_frame_list,
            "lr_upsample_reset": self.lr_upsample_reset,
        }
        kw.update(self.tensorfs[0].get_kwargs())
        return kw

    def save(self, path):
        torch.save({"kwargs": self.get_kwargs(), "state_dict": self.state_dict()}, path)

    def load(self, state_dict):
        """Re-grow fields/frames to match a checkpoint, then load it (local_tensorfs.py:331-356)."""
        n_frames = 0
        for key in state_dict:
            if re.fullmatch(r"r_c2w.[0-9]*", key):
                n_frames += 1
            if re.fullmatch(r"tensorfs.[1-9][0-9]*.density_plane.0", key):
                pl = state_dict[key]
                ln = state_dict[f"{key[:-15]}density_line.0"]
                self.tensorf_args["gridSize"] = [pl.shape[2], pl.shape[3], ln.shape[2]]
                self.append_rf()
        for i in range(len(self.tensorfs)):
            if f"tensorfs.{i}.alphaMask.aabb" in state_dict:
                vol = state_dict[f"tensorfs.{i}.alphaMask.alpha_volume"].to(self.device)
                aabb = state_dict[f"tensorfs.{i}.alphaMask.aabb"].to(self.device)
                self.tensorfs[i].alphaMask = AlphaGridMask(self.device, aabb, vol)
        for _ in range(n_frames - len(self.r_c2w)):
            self.append_frame()
        self.blending_weights = torch.nn.Parameter(
            torch.ones_like(state_dict["blending_weights"]), requires_grad=False)
        self.load_state_dict(state_dict)

    def get_dist_to_last_rf(self):
        return torch.norm(self.t_c2w[-1] + self.world2rf[-1])

    def get_reg_loss(self, tvreg, TV_weight_density, TV_weight_app, L1_weight_inital):
        """local_tensorfs.py:361-375."""
        tv_loss, l1_loss = 0, 0
        if self.rf_iter[-1] < self.n_iters:
            sched = self.lr_factor ** self.rf_iter[-1]
            if TV_weight_density > 0:
                tv_loss += self.tensorfs[-1].TV_loss_density(tvreg).mean() * TV_weight_density * sched
            if TV_weight_app > 0:
                tv_loss += self.tensorfs[-1].TV_loss_app(tvreg).mean() * TV_weight_app * sched
            if L1_weight_inital > 0:
                l1_loss += self.tensorfs[-1].density_L1() * L1_weight_inital
        return tv_loss, l1_loss

    def focal(self, W):
        return self.init_focal * self.focal_offset * W / self.W

    def center(self, W, H):
        key = (W, H, self.center_rel.device)
        wh = getattr(self, "_wh_cache", {}).get(key)
        if wh is None:                          # the reference uploads [W, H] on every call (:380)
            wh = torch.tensor([float(W), float(H)], device=self.center_rel.device)
            self._wh_cache = {key: wh}
        return wh * self.center_rel

    # ----------------------------------------------------------------- the render entry point
    def forward(self, ray_ids, view_ids, W, H, white_bg=True, is_train=True, cam2world=None,
                world2rf=None, blending_weights=None, chunk=16384, test_id=False, floater_thresh=0):
        """Pixel ids -> rays -> per-field native render -> blend -> exposure -> clamp
        (local_tensorfs.py:382-499).  Returns (rgbs [R,3], depth [R], directions [R,3], ij [R,2])."""
        dev = self.blending_weights.device
        if dev.type != "cuda":
            raise NativeError("localrf_amd: LocalTensorfs.forward runs only on an AMD GPU (HIP kernels); "
                              f"the scene lives on {dev}. There is no CPU fallback.")
        # Ids may arrive on the host (extension): they are staged through pinned memory without
        # blocking.  Device-resident view_ids (what train.py:352 passes) cost one host sync here,
        # as in the reference, which reads them one .item() at a time (local_tensorfs.py:294-295).
        # Any pageable host->device copy blocks the host until the stream drains (measured,
        # scripts/ubench/h2d_sync.py), so a caller that keeps ids on the host overlaps its next
        # iteration's host work with this one's kernels.
        if torch.is_tensor(view_ids) and view_ids.is_cuda:
            view_list = view_ids.tolist()
        else:
            view_list = [int(v) for v in (view_ids.tolist() if hasattr(view_ids, "tolist") else view_ids)]
            view_ids = _upload_ids(view_list, dev)
        if not (torch.is_tensor(ray_ids) and ray_ids.is_cuda):
            ray_ids = _upload_ids(ray_ids, dev)
        n_rays, n_views = ray_ids.shape[0], len(view_list)
        if cam2world is None:
            cam2world = self.get_cam2world(view_list)
        if world2rf is None:
            world2rf = self.world2rf

        # which fields render, and with what per-view weight (:405-422)
        if is_train:                                            # one field trains at a time (:411-416)
            active = [len(self.tensorfs) - 1]
            bw = self._ones(n_views, dev)
        else:
            if blending_weights is None:
                host = self._blending_host()[view_list]         # host mirror: no device round trip
                active = torch.nonzero(host.sum(0))[:, 0].tolist()
                blending_weights = self.blending_weights[view_ids]
            elif blending_weights.is_cuda:                      # :405-410; reading the active set back is a device sync
                active = torch.nonzero(torch.sum(blending_weights, dim=0))[:, 0].tolist()
            else:                                               # host weights (extension): no sync, pinned upload
                active = torch.nonzero(torch.sum(blending_weights, dim=0))[:, 0].tolist()
                stage = torch.empty(blending_weights.shape, dtype=torch.float32, pin_memory=True)
                stage.copy_(blending_weights)
                blending_weights = stage.to(dev, non_blocking=True)
            # blending_weights[:, active] with a Python list uploads the index with a pageable copy, which blocks the
            # host until the stream drains (DESIGN.md finding 7): slice when the active fields are contiguous
            if active == list(range(active[0], active[-1] + 1)) if active else False:
                bw = blending_weights[:, active[0]:active[-1] + 1]
            else:
                bw = blending_weights.index_select(1, _upload_ids(active, dev)) if active else blending_weights[:, :0]

        pinhole = self.fov != 360
        focal = self.focal(W) if pinhole else None
        center = self.center(W, H) if pinhole else None
        if len(active) == 0:                                    # degenerate 5-tuple (:420-422)
            print("****** No valid RF")
            _, directions, ij = scene_rays(ray_ids, cam2world, torch.zeros(1, 3, device=dev), focal, center,
                                           max(1, n_rays // max(1, n_views)), W, H, not pinhole)
            return (torch.ones([n_rays, 3]), torch.ones_like(ray_ids).float(),
                    torch.ones_like(ray_ids).float(), directions, ij)
        if n_rays % n_views:
            raise ValueError("ray_ids must hold the same number of rays for every view")
        per_view = n_rays // n_views

        shifts = torch.stack([world2rf[rf] for rf in active], dim=0)
        for rf in active:
            if self.tensorfs[rf].device != dev:
                self.tensorfs[rf].to(dev)                       # stays there (no shuttle back)

        # `chunk` bounds the reference's peak memory (:440: chunk // n_active rays per field call).  Rays are
        # independent, so the result does not depend on it; with 288 GB of HBM a field renders up to
        # self.min_chunk rays per call whatever the caller's chunk
        chunk = max(1, chunk // len(active), self.min_chunk)
        taped = torch.is_grad_enabled() and (cam2world.requires_grad or shifts.requires_grad or any(
            p.requires_grad for rf in active for p in self.tensorfs[rf].parameters()) or any(
            t is not None and t.requires_grad for t in (focal, center)))
        if not taped and not is_train:                          # one native call for the whole scene forward (lrf_scene_fwd)
            return scene_forward(ray_ids, cam2world, shifts, focal, center, per_view, W, H, not pinhole,
                                 [self.tensorfs[rf] for rf in active], white_bg, floater_thresh, chunk, bw,
                                 self._exposure_for(view_ids, test_id))
        rays, directions, ij = scene_rays(ray_ids, cam2world, shifts, focal, center, per_view, W, H, not pinhole)
        cols_rgb, cols_dep = [], []
        for k, rf in enumerate(active):
            parts = [self.tensorfs[rf](rays[k, lo:lo + chunk], is_train=is_train, white_bg=white_bg,
                                       N_samples=-1, refine=self.is_refining, floater_thresh=floater_thresh)
                     for lo in range(0, n_rays, chunk)]
            cols_rgb.append(parts[0][0] if len(parts) == 1 else torch.cat([p[0] for p in parts], 0))
            cols_dep.append(parts[0][1] if len(parts) == 1 else torch.cat([p[1] for p in parts], 0))
        rgb_f = cols_rgb[0][None] if len(active) == 1 else torch.stack(cols_rgb, 0)
        dep_f = cols_dep[0][None] if len(active) == 1 else torch.stack(cols_dep, 0)

        exposure = self._exposure_for(view_ids, test_id)
        rgbs, depth_maps = scene_blend(rgb_f, dep_f, bw, exposure, per_view)
        return rgbs, depth_maps, directions, ij

    def _exposure_for(self, view_ids, test_id):
        """Per-view 3x3 colour transform (local_tensorfs.py:481-496); None when exposure is not optimised."""
        if self.lr_exposure_init <= 0:
            return None
        if test_id:                                             # a held-out view borrows its neighbours' (:483-492)
            prev = torch.clamp(view_ids - 1, min=0)             # scalar bounds: no blocking upload
            prev[prev == view_ids] = 1
            nxt = torch.clamp(view_ids + 1, max=len(self.exposure) - 1)
            nxt[prev == view_ids] = len(self.exposure) - 2
            stacked = torch.stack(list(self.exposure), dim=0).clone().detach()
            return (stacked[prev] + stacked[nxt]) / 2
        return torch.stack(list(self.exposure), dim=0)[view_ids]

    def _ones(self, n, dev):
        key = (n, str(dev))
        if getattr(self, "_ones_cache", (None, None))[0] != key:
            self._ones_cache = (key, torch.ones(n, 1, device=dev))
        return self._ones_cache[1]

    def _blending_host(self):
        """CPU mirror of the (non-trainable) blending weights, refreshed when they change."""
        bwp = self.blending_weights
        key = (bwp.data_ptr(), bwp._version, tuple(bwp.shape))
        if getattr(self, "_bw_host", (None, None))[0] != key:
            self._bw_host = (key, bwp.detach().cpu())
        return self._bw_host[1]
