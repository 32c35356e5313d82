"""LocalTensorfs -- self-calibrating scene of blended local radiance fields.

Drop-in for the reference's `local_tensorfs.LocalTensorfs` (same constructor, attribute and
state-dict names, `forward` signature/returns; the lifecycle -- `optimizer_step`, `append_frame`, `append_rf`,
`save`/`load` -- lives in compat_lifecycle.py, with its attribution).  `forward` (local_tensorfs.py:382-499) is three kinds of HIP launch:
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
import torch

from ._native import NativeError
from .compat_lifecycle import SceneLifecycle
from .rays import sixD_to_mtx
from .scene_ops import pose_assemble, rows_gather, scene_blend, scene_forward, scene_rays


def _upload_ids(ids, dev):
    """Host ids -> device int64 without a blocking copy: staged in pinned memory from torch's
    caching host allocator (a block is not reused before the copy that reads it has run)."""
    src = torch.as_tensor(ids, dtype=torch.int64)
    stage = torch.empty(src.shape, dtype=torch.int64, pin_memory=True)
    stage.copy_(src)
    return stage.to(dev, non_blocking=True)


class LocalTensorfs(SceneLifecycle):
    """The scene: lifecycle from compat_lifecycle.SceneLifecycle (the reference's control flow, attributed there), the render
    entry point and its helpers here."""

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

    def focal(self, W):
        if self.freeze_intrinsics:                              # no optimiser reads their gradients this iteration: no backward chain,
            return self._frozen("focal", (W,), (self.init_focal, self.focal_offset),   # and the value is kept until they change
                                lambda: self.init_focal * self.focal_offset * W / self.W)
        return self.init_focal * self.focal_offset * W / self.W

    def center(self, W, H):
        if self.freeze_intrinsics:
            return self._frozen("center", (W, H), (self.center_rel,), lambda: self._center(W, H))
        return self._center(W, H)

    def _frozen(self, name, args, params, fn):
        key = (args,) + tuple((p.data_ptr(), p._version) for p in params)
        cache = self.__dict__.setdefault("_frozen_cache", {})
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = cache[name] = (key, fn())
        return hit[1]

    def _shifts(self, world2rf, active):
        """[n_active, 3] world -> field translations (local_tensorfs.py:427-431), stacked once per set of tensors: they are
        constants of the scene between lifecycle events, and a tape through them only when one of them wants a gradient."""
        ws = [world2rf[rf] for rf in active]
        if any(w.requires_grad for w in ws) and torch.is_grad_enabled():
            return torch.stack(ws, dim=0)
        key = tuple((w.data_ptr(), w._version) for w in ws)
        hit = self.__dict__.get("_shift_cache")
        if hit is None or hit[0] != key:
            hit = (key, torch.stack([w.detach() for w in ws], dim=0))
            self.__dict__["_shift_cache"] = hit
        return hit[1]

    def _center(self, W, H):
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
        if torch.is_tensor(view_ids) and view_ids.is_cuda and cam2world is not None and is_train:
            view_list = None                                    # poses supplied, one field trains: nothing on the host needs the ids
            n_views = int(view_ids.shape[0])                    # (no read-back: this call can sit in a captured graph)
        elif torch.is_tensor(view_ids) and view_ids.is_cuda:
            view_list = view_ids.tolist()
        else:
            view_list = [int(v) for v in (view_ids.tolist() if hasattr(view_ids, "tolist") else view_ids)]
            view_ids = _upload_ids(view_list, dev)
        if not (torch.is_tensor(ray_ids) and ray_ids.is_cuda):
            ray_ids = _upload_ids(ray_ids, dev)
        n_rays = ray_ids.shape[0]
        if view_list is not None:
            n_views = len(view_list)
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
            # host until the stream drains (docs/GFX950_FINDINGS.md finding 7): slice when the active fields are contiguous
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

        shifts = self._shifts(world2rf, active)
        for rf in active:
            if self.tensorfs[rf].device != dev:
                self.tensorfs[rf].to(dev)                       # stays there (no shuttle back)

        # `chunk` bounds the reference's peak memory (:440: chunk // n_active rays per field call).  Rays are
        # independent, so the result does not depend on it; with 288 GB of HBM a field renders up to
        # self.min_chunk rays per call whatever the caller's chunk when no gradient is recorded
        per_field = max(1, chunk // len(active))
        taped = torch.is_grad_enabled() and (cam2world.requires_grad or shifts.requires_grad or any(
            p.requires_grad for rf in active for p in self.tensorfs[rf].parameters()) or any(
            t is not None and t.requires_grad for t in (focal, center)) or (
            self.lr_exposure_init > 0 and not test_id and any(e.requires_grad for e in self.exposure)))
        if not taped and not is_train:                          # one native call for the whole scene forward (lrf_scene_fwd)
            return scene_forward(ray_ids, cam2world, shifts, focal, center, per_view, W, H, not pinhole,
                                 [self.tensorfs[rf] for rf in active], white_bg, floater_thresh,
                                 self._untaped_chunk(per_field, [self.tensorfs[rf] for rf in active]), bw,
                                 self._exposure_for(view_ids, test_id), refine=self.is_refining)
        # with a tape the caller's chunk is honoured as is: the row-saving workspace is ~0.37 MB per ray at S = 512
        chunk = per_field
        single = len(active) == 1 and chunk >= n_rays          # one field, one call: its rays without indexing the [1,R,6] result
        rays, directions, ij = scene_rays(ray_ids, cam2world, shifts, focal, center, per_view, W, H, not pinhole, squeeze=single)
        cols_rgb, cols_dep = [], []
        for k, rf in enumerate(active):
            parts = [self.tensorfs[rf](rays if single else rays[k, lo:lo + chunk], is_train=is_train, white_bg=white_bg,
                                       N_samples=-1, refine=self.is_refining, floater_thresh=floater_thresh)
                     for lo in range(0, n_rays, chunk)]
            cols_rgb.append(parts[0][0] if len(parts) == 1 else torch.cat([p[0] for p in parts], 0))
            cols_dep.append(parts[0][1] if len(parts) == 1 else torch.cat([p[1] for p in parts], 0))
        rgb_f = cols_rgb[0][None] if len(active) == 1 else torch.stack(cols_rgb, 0)
        dep_f = cols_dep[0][None] if len(active) == 1 else torch.stack(cols_dep, 0)

        exposure = self._exposure_for(view_ids, test_id)
        rgbs, depth_maps = scene_blend(rgb_f, dep_f, bw, exposure, per_view)
        return rgbs, depth_maps, directions, ij

    def _untaped_chunk(self, per_field, fields):
        """Rays per field call of a forward without a tape: the caller's chunk // n_active, raised to self.min_chunk (an
        eval image rendered 4096 rays at a time, renderer.py:75, is launch-bound on this GPU) -- but only while the
        per-call workspace of the largest active field stays below self.max_untaped_workspace bytes (default 1 GiB: 65536
        rays are 0.31 GiB at 640^3's 738 samples); beyond that the caller's bound is what it asked for and is honoured."""
        want = max(per_field, self.min_chunk)
        if want > per_field:
            from . import _native as N
            S = max(2 * (int(f.nSamples) // 6) for f in fields)
            while want > per_field and N.lib().lrf_workspace_bytes(want, S) > self.max_untaped_workspace:
                want = max(per_field, want // 2)
        return want

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
        stacked = torch.stack(list(self.exposure), dim=0)
        if stacked.is_cuda and view_ids.is_cuda and view_ids.dtype == torch.int64:
            return rows_gather(stacked, view_ids)               # [view_ids] and its index_put backward in one launch each way
        return stacked[view_ids]

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
