#!/usr/bin/env python
"""Config-5 driver: the progressive optimisation loop of the reference's train.py:349-474 around
localrf_amd.LocalTensorfs, on SYNTHETIC data (no dataset, no network): sample -> forward -> losses ->
optimizer_step -> progressive append_frame / append_rf, the upsample schedule of opt.py:61-70
(64^3 -> ... -> N_voxel_final), alpha-mask rebuilds, density_L1, FusedAdam.

The targets come from a hidden teacher scene (a TensorVMSplit with solid walls and random appearance)
rendered by the same HIP path from a camera that moves along a straight line, so the loss really
can fall and the poses really drift as the camera moves.  The optical-flow and monocular-depth
losses (train.py:385-423, weights 1 and 0.1 as opt.py sets them) are on while the field regularises:
their targets -- what RAFT / DPT provide to the reference -- are the teacher's exact flow between
neighbouring frames and its inverse depth (localrf_amd.losses kernels).  `--no-geo` switches them off.

  python scripts/train_synth.py [--frames 24] [--final 300] [--iters-per-frame 60] [--json out.json]
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 scripts/train_synth.py ...
      (data parallel: every rank draws the same batch and renders its share of the views; one
       gradient all-reduce per iteration through LocalTensorfs.grad_sync)

Prints one JSON object: loss curve summary, ms / iteration at every grid resolution, peak memory,
checkpoint round trip.  `run()` is also called (tiny settings) by tests/test_gpu_training.py.
"""
import argparse
import contextlib
import io
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FIELD_KW = dict(density_n_comp=[8, 8, 8], appearance_n_comp=[24, 24, 24], app_dim=27,
                shadingMode="MLP_Fea_late_view", near_far=[0.1, 1e3], density_shift=-5,
                alphaMask_thres=1e-4, distance_scale=25, rayMarch_weight_thres=1e-3,
                pos_pe=0, view_pe=0, fea_pe=0, featureC=128, step_ratio=0.5, fea2denseAct="softplus")


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


class SyntheticFrames:
    """What train.py needs of LocalRFDataset (dataLoader/localrf_dataset.py): a growing window of
    active frames, `sample()` -> 16 views x batch/16 rays with their target colours, activate /
    deactivate.  Targets: the teacher field rendered from the true camera of each frame."""

    def __init__(self, n_frames, W, H, n_init, dev, seed=0):
        from localrf_amd import TensorVMSplit
        self.W, self.H, self.num_images, self.dev = W, H, n_frames, dev
        self.active_frames_bounds = [0, n_init]
        self.rng = np.random.default_rng(seed)
        torch.manual_seed(1234)
        aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
        teacher = quiet(TensorVMSplit, torch.device("cpu"), aabb, [96, 96, 96], **FIELD_KW)
        with torch.no_grad():
            for p in teacher.density_plane:
                p.mul_(0.1)
            c = torch.linspace(-2, 2, 96)
            for p in range(3):
                for comp, centre in ((0, 1.2), (1, -1.2)):
                    teacher.density_plane[p][0, comp].fill_(1.0)
                    teacher.density_line[p][0, comp, :, 0] = 40.0 * torch.exp(-((c - centre) / 0.1) ** 2)
            for p in teacher.app_plane:
                p.mul_(6.0)
            teacher.renderModule.mlp_view[0].weight.mul_(25.0)          # saturated, textured colours (image std ~0.2)
        teacher = teacher.to(dev)
        focal = W / math.tan(85.6 * math.pi / 180 / 2) / 2
        jj, ii = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        dirs = torch.stack([(ii.float() + 0.5 - W / 2) / focal, -(jj.float() + 0.5 - H / 2) / focal,
                            -torch.ones(H, W)], -1).reshape(-1, 3)
        self.images, depths = [], []
        # true camera path: 0.04 per frame along an arc that stays inside |x|, |z| <= 0.6, well clear of the teacher's walls
        # at +-1.2 (round 2 walked +x in a straight line and reached the wall at frame 30: the target flow grew from 1.4
        # to 15 pixels and the teacher's median depth fell from 0.97 to 0.09 over the run, which is what the "rising"
        # geometric losses of profiles/r02_train_synth_500.json were: see profiles/r09a_geo_curve.md)
        cam_t = torch.tensor([[0.6 * math.sin(f / 15.0), 0.01 * math.sin(0.5 * f), 0.6 * (1 - math.cos(f / 15.0)) - 0.3]
                              for f in range(n_frames)])
        with torch.no_grad():
            for f in range(n_frames):
                rays = torch.cat([cam_t[f].expand_as(dirs), dirs], -1).to(dev)
                rgb, dep = teacher(rays, white_bg=True, is_train=False, N_samples=300)
                self.images.append(rgb.clamp(0, 1))
                depths.append(dep)
        self.images = torch.stack(self.images)                        # [F, H*W, 3] on the device
        self.cam_t = cam_t
        # what RAFT / DPT give the reference (dataLoader/localrf_dataset.py): flow to the next / previous frame and an
        # inverse depth per pixel -- here exact, from the teacher's depth and the true (rotation-free) camera path
        depths = torch.stack(depths)                                  # [F, H*W]
        pix = torch.stack([ii.reshape(-1), jj.reshape(-1)], -1).float().to(dev)
        dirs_d = dirs.to(dev)

        def flow_to(f, g):
            q = dirs_d * depths[f][:, None] + (cam_t[f] - cam_t[g]).to(dev)        # point in frame g's camera (utils.py:43-46)
            zc = (-q[:, 2]).clamp(min=1e-6)
            px = torch.stack([q[:, 0] / zc * focal + W / 2 - 0.5, -q[:, 1] / zc * focal + H / 2 - 0.5], -1)
            return px - pix
        self.fwd_flow = torch.stack([flow_to(f, min(f + 1, n_frames - 1)) for f in range(n_frames)])
        self.bwd_flow = torch.stack([flow_to(f, max(f - 1, 0)) for f in range(n_frames)])
        self.invdepths = 1.0 / depths.clamp(min=1e-6)
        self.stats = {"mean": float(self.images.mean()), "std": float(self.images.std()),
                      "frame_to_frame": float((self.images[1:] - self.images[:-1]).abs().mean())}
        del teacher

    def has_left_frames(self):
        return self.active_frames_bounds[1] < self.num_images

    def activate_frames(self, n=1):
        self.active_frames_bounds[1] = min(self.active_frames_bounds[1] + n, self.num_images)

    def deactivate_frames(self, first_kept):
        self.active_frames_bounds[0] = int(first_kept)

    def sample(self, batch_size, n_views=16):
        lo, hi = self.active_frames_bounds
        views = np.sort(self.rng.integers(lo, hi, n_views))
        per = batch_size // n_views
        pix = self.rng.integers(0, self.W * self.H, (n_views, per))
        view_t = torch.from_numpy(views)
        pix_t = torch.from_numpy(pix)
        return view_t, pix_t.reshape(-1), (view_t[:, None], pix_t)


def geo_by_field(curve):
    """Per field: the forward-flow error relative to the magnitude of the target flow, and the depth loss, at the first
    and the last recorded regularising iteration of that field (the schedule weight reg_w decays in between)."""
    out = []
    for fld in sorted({c["field"] for c in curve}):
        cs = [c for c in curve if c["field"] == fld and c["target_flow_mag"] > 0]
        if len(cs) >= 2:
            out.append({"field": fld, "records": len(cs), "frames_first": cs[0]["frames"], "frames_last": cs[-1]["frames"],
                        "flow_rel_first": cs[0]["fwd_err"] / cs[0]["target_flow_mag"], "flow_rel_last": cs[-1]["fwd_err"] / cs[-1]["target_flow_mag"],
                        "flow_rel_min": min(c["fwd_err"] / c["target_flow_mag"] for c in cs),
                        "target_flow_first_last": [cs[0]["target_flow_mag"], cs[-1]["target_flow_mag"]],
                        "depth_first": cs[0]["depth"], "depth_last": cs[-1]["depth"], "depth_max": max(c["depth"] for c in cs)})
    return out


class LateScalar:
    """A device scalar the host reads one iteration late: request() enqueues a copy into pinned memory behind the work issued so
    far, value() waits for THAT copy only -- the host stays one iteration ahead of the GPU instead of draining the stream on
    every iteration (train.py:441 reads get_dist_to_last_rf().cpu().item() synchronously)."""

    def __init__(self, initial=0.0):
        self.pin = torch.zeros(1, pin_memory=True)
        self.pin[0] = initial
        self.ev = None

    def request(self, t):
        self.pin.copy_(t.detach().reshape(1), non_blocking=True)
        if self.ev is None:
            self.ev = torch.cuda.Event()
        self.ev.record()

    waited = 0.0                                                      # (class-wide: seconds the host spent waiting in value())

    def value(self):
        if self.ev is not None:
            t0 = time.perf_counter()
            self.ev.synchronize()
            LateScalar.waited += time.perf_counter() - t0
        return float(self.pin[0])


class LateLog:
    """Logged device values read one logging period late (the captured loop): put(tensors, sink) enqueues their copy into
    pinned memory behind the replay that produced them; the values reach `sink(floats)` at the next put() / flush().  A
    float(tensor) per logged value drains the stream -- the GPU then idles for the host's share of an iteration, every 25
    iterations, twice in the regularised phase."""

    def __init__(self, width=4):
        self.pin = torch.zeros(width, pin_memory=True)
        self.ev = None
        self.pending = None

    def flush(self):
        if self.pending is not None:
            t0 = time.perf_counter()
            self.ev.synchronize()
            LateScalar.waited += time.perf_counter() - t0
            n, sink = self.pending
            sink([float(x) for x in self.pin[:n]])
            self.pending = None

    def put(self, tensors, sink):
        self.flush()
        for i, t in enumerate(tensors):
            self.pin[i:i + 1].copy_(t.detach().reshape(-1)[:1] if t.numel() == 1 else t.detach().sum().reshape(1), non_blocking=True)
        if self.ev is None:
            self.ev = torch.cuda.Event()
        self.ev.record()
        self.pending = (len(tensors), sink)


def geometric_terms(lt, data, depth_map, directions, ij, cam2world_all, view_ids, start, vsel, psel, W, H):
    """train.py:385-423: the optical-flow and monocular-depth losses of one batch (localrf_amd.losses kernels).  view_ids:
    host or device ids; vsel [V] / psel [V, n]: device indices of the batch's views / pixels into the dataset tensors."""
    from localrf_amd import losses as geo_losses
    last = data.num_images - 1
    fl = geo_losses.flow_loss(depth_map, directions, ij, cam2world_all, view_ids, start,
                              data.fwd_flow[vsel[:, None], psel], (vsel < last).float()[:, None].expand(psel.shape),
                              data.bwd_flow[vsel[:, None], psel], (vsel > 0).float()[:, None].expand(psel.shape),
                              lt.focal(W), lt.center(W, H))
    dl = geo_losses.depth_loss(depth_map, data.invdepths[vsel[:, None], psel], int(vsel.shape[0]))
    return fl, dl


def run(frames=24, W=64, H=48, n_init=5, final=300, iters_per_frame=60, batch=4096, max_iters=None, seed=0,
        dev="cuda:0", ddp=False, log=None, max_drift=0.25, n_max_frames=12, geo=True, geo_every=25, graph=False, record_all=False, live=None, lr_i_init=0,
        fuse_l1=True):
    from localrf_amd import LocalTensorfs, losses as geo_losses
    from localrf_amd.dist import allreduce_grads, shard_views
    from localrf_amd.rays import N_to_reso
    import torch.distributed as dist
    dev = torch.device(dev)
    rank = dist.get_rank() if ddp else 0
    world = dist.get_world_size() if ddp else 1
    data = SyntheticFrames(frames, W, H, n_init, dev, seed)
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]]).to(dev)
    # opt.py:61-70 scaled by iters_per_frame / 600
    sc = iters_per_frame / 600.0
    upsamp = [max(1, round(u * sc)) for u in (100, 150, 200, 250, 300)]
    n_init_vox, n_final_vox = 64 ** 3, final ** 3
    nvox = torch.round(torch.exp(torch.linspace(math.log(n_init_vox), math.log(n_final_vox), len(upsamp) + 1))).long().tolist()[1:]
    N_voxel_list = {u: round(n ** (1 / 3)) ** 3 for u, n in zip(upsamp, nvox)}
    mask_list = [max(1, round(u * sc)) for u in (100, 200, 300)]
    torch.manual_seed(seed)                                           # identical replicas under DDP
    lt = quiet(LocalTensorfs, camera_prior=None, fov=85.6, n_init_frames=min(n_init, frames), n_overlap=3, WH=(W, H),
               n_iters_per_frame=iters_per_frame, n_iters_reg=max(1, round(100 * sc)), lr_R_init=5e-3, lr_t_init=5e-4,
               lr_i_init=lr_i_init, lr_exposure_init=1e-3, rf_lr_init=0.02, rf_lr_basis=1e-3, lr_decay_target_ratio=0.1,
               N_voxel_list=N_voxel_list, update_AlphaMask_list=mask_list, lr_upsample_reset=True, device=dev,
               aabb=aabb, gridSize=N_to_reso(n_init_vox, aabb), **FIELD_KW).to(dev)
    if ddp:
        lt.grad_sync = lambda m: allreduce_grads(m, average=True)
    # train.py defaults: add_frames_every 100, n_max_frames 100, max_drift 1 (opt.py); the synthetic camera moves 0.04 per
    # frame and short runs do not recover that motion, so the frame-count criterion (n_max_frames) is what starts a
    # refinement / a new field here
    L1_weight, add_frames_every, n_overlap = 1e-2, max(1, round(100 * sc)), 3
    # graph=True: the iteration as one replayed hipGraph (localrf_amd/graph_step.py) -- the same loss written against static
    # device inputs; what this iteration's loss contains (decided by the PREVIOUS optimizer_step, as in the eager branch below)
    # is part of the captured state
    phase = {"reg": False}

    def graph_loss(rgb_map, depth_map, directions, ij, inp):
        # the same loss as the eager branch below, written with the two assembly kernels (losses.batch_gather / combine): the
        # dozen (view, pixel) gathers, mask expressions and scalar products of train.py:352-437 are 45 of the 94 launches of a
        # captured iteration of the regularised phase when each goes through ATen
        V = inp.n_views
        psel = inp.ray_ids.reshape(V, -1)
        n = int(psel.shape[1])
        want_geo = phase["reg"] and geo
        rows = geo_losses.batch_gather(inp.view_ids, psel, images=data.images, fwd_flow=data.fwd_flow if want_geo else None,
                                       bwd_flow=data.bwd_flow if want_geo else None, invdepths=data.invdepths if want_geo else None)
        loss = geo_losses.photometric_loss(rgb_map, rows["target"])
        terms, kept = [(loss, 1.0, 0.0)], {"photo": loss}
        if want_geo:
            fl = geo_losses.flow_loss(depth_map, directions, ij, inp.cam2world_all, inp.view_ids, inp.start, rows["fwd_flow"], rows["fwd_mask"],
                                      rows["bwd_flow"], rows["bwd_mask"], lt.focal(W), lt.center(W, H), per_view=True, frame_ids=inp.frame32)
            dl = geo_losses.depth_loss(depth_map, rows["invdepths"], V, per_view=True)
            terms += [(fl, 0.0, 1.0 / ((W + H) / 2) / (V * n)), (dl, 0.0, 0.1 / (V * n))]      # opt.py: loss weights 1 and 0.1, times reg_w
            kept.update(flow=fl, depth=dl, geo_norm=float(V * n))
        if phase["reg"] and lt.rf_iter[-1] < lt.n_iters and L1_weight > 0:                      # local_tensorfs.py:361-375
            terms.append((lt.tensorfs[-1].density_L1(), L1_weight, 0.0))
        total = loss if len(terms) == 1 else geo_losses.combine(terms, inp.scalars["reg_w"])
        return total, kept

    gs = None
    if graph:
        from localrf_amd.graph_step import CapturedIteration
        gs = CapturedIteration(lt, W, H, batch // world, 16 // world, graph_loss, scalar_names=("reg_w",), optimize_poses=True)
        gs.extra_signature = lambda: (phase["reg"], bool(getattr(lt.tensorfs[-1], "fuse_density_L1", False)))
    n_added, last_add, it = 0, 0, 0
    drift, drift_prev = LateScalar(), 0.0
    all_losses = []
    losses, per_res, events, geo_vals, geo_curve = [], {}, [], [], []
    late_geo, late_loss = LateLog(), LateLog()
    LateScalar.waited = 0.0
    torch.cuda.reset_peak_memory_stats(dev)
    mem_marks = []
    training = True
    t_mark, it_mark = time.perf_counter(), 0
    res = int(lt.tensorfs[-1]._grid_host[0])                          # (the host copy: gridSize itself is a device tensor, as in the reference -- reading it drains the stream)
    pace = []
    host_t = {}                                                        # res -> seconds of host time by section of the loop body (captured loop)
    while training and (max_iters is None or it < max_iters):
        tq0 = time.perf_counter()
        view_ids, ray_idx, (vv, pp) = data.sample(batch)
        tq1 = time.perf_counter()
        # the regulariser's gradient rides on the render's backward (TensorVMSplit.fuse_density_L1) whenever this iteration's
        # loss will contain it: local_tensorfs.py:361-375
        lt.tensorfs[-1].fuse_density_L1 = bool(fuse_l1 and lt.regularize and lt.rf_iter[-1] < lt.n_iters and L1_weight > 0)
        if gs is not None:                                             # the captured iteration: the same schedule calls around one replay
            all_views = view_ids.tolist()                              # every rank steps the poses of the whole batch's views
            if ddp:
                ray_idx, view_ids = shard_views(ray_idx, view_ids, rank, world)
            phase["reg"] = bool(lt.regularize)
            reg_w = lt.lr_factor ** lt.rf_iter[-1]
            start = max(data.active_frames_bounds[0] - 1, 0)
            pose_ids, tune = lt.step_begin(True, zero_grad=False)
            tq2 = time.perf_counter()
            kept = gs.step(view_ids.tolist(), ray_idx.numpy(), {"reg_w": reg_w}, all_poses_active=phase["reg"] and geo,
                           pose_ids=pose_ids, tune_intrinsics=tune, start=start, global_views=all_views)
            tq3 = time.perf_counter()
            lt.step_schedule()
            can_add_rf = lt.step_finish()
            tq4 = time.perf_counter()
            ht = host_t.setdefault(res, [0.0, 0.0, 0.0, 0.0, 0.0, 0])
            ht[0] += tq1 - tq0; ht[1] += tq2 - tq1; ht[2] += tq3 - tq2; ht[3] += tq4 - tq3; ht[5] += 1
            loss = kept["photo"]
            if phase["reg"] and geo:
                geo_vals.append(None)
                if it % geo_every == 0:                                # (read one period late: no drain)
                    slot, norm = len(geo_vals) - 1, kept["geo_norm"]
                    late_geo.put([kept["flow"], kept["depth"]], lambda v, slot=slot, norm=norm: geo_vals.__setitem__(slot, (v[0] / norm, v[1] / norm)))
        else:
            # indices go up through pinned memory: indexing a device tensor with host indices (or any pageable
            # host->device copy, as train.py:352-358 does) blocks the host until the stream has drained
            vv_d = vv.pin_memory().to(dev, non_blocking=True)
            pp_d = pp.pin_memory().to(dev, non_blocking=True)
            target = data.images[vv_d, pp_d].reshape(-1, 3)
            if ddp:                                                        # this rank's views of the common batch
                per = ray_idx.shape[0] // view_ids.shape[0]
                ray_idx, v_sh = shard_views(ray_idx, view_ids, rank, world)
                v0 = rank * v_sh.shape[0]
                target = target[v0 * per:(v0 + v_sh.shape[0]) * per]
                view_ids = v_sh
            rgb_map, depth_map, directions, ij = lt(ray_idx, view_ids.tolist(), W, H, is_train=True, test_id=False)
            loss = geo_losses.photometric_loss(rgb_map, target)            # train.py:369-371, unit loss weights (one launch each way)
            total = loss
            if lt.regularize and geo:                                      # train.py:357,385-423; opt.py: weights 1 and 0.1
                reg_w = lt.lr_factor ** lt.rf_iter[-1]
                start = max(data.active_frames_bounds[0] - 1, 0)
                if ddp:
                    vsel = view_ids.pin_memory().to(dev, non_blocking=True)
                    psel = ray_idx.reshape(view_ids.shape[0], -1).pin_memory().to(dev, non_blocking=True)
                else:
                    vsel, psel = vv_d[:, 0], pp_d
                fl, dl = geometric_terms(lt, data, depth_map, directions, ij, lt.get_cam2world(starting_id=start), view_ids, start, vsel, psel, W, H)
                total = total + fl * 1.0 * reg_w / ((W + H) / 2) + dl * 0.1 * reg_w
                geo_vals.append((float(fl.detach()), float(dl.detach())) if it % geo_every == 0 else None)
                if it % geo_every == 0:                                    # the curve, phase by phase (profiles/r09*_geo_curve)
                    lo, hi = data.active_frames_bounds
                    with torch.no_grad():
                        t_est = torch.stack([lt.t_c2w[f].detach() for f in range(lo, hi)]).cpu()
                        step_est = (t_est[1:] - t_est[:-1]).norm(dim=-1).mean() if hi - lo > 1 else torch.zeros(())
                        rel = (t_est - t_est[:1]) - (data.cam_t[lo:hi] - data.cam_t[lo:lo + 1])
                        # diagnostics in plain torch (train.py:388-404 restated; not the loss that is optimised)
                        c2w = lt.get_cam2world(starting_id=start).detach()
                        fr = (vsel - start).long()
                        nxt = (fr + 1).clamp(max=c2w.shape[0] - 1)
                        Rn, tn = c2w[nxt, :3, :3], c2w[nxt, :3, 3]
                        Rc, tc = c2w[fr, :3, :3], c2w[fr, :3, 3]
                        dm = depth_map.detach().reshape(fr.shape[0], -1)
                        pts = directions.detach().reshape(fr.shape[0], -1, 3) * dm[..., None]
                        p_w = torch.einsum("vij,vnj->vni", Rc, pts) + tc[:, None]
                        q = torch.einsum("vji,vnj->vni", Rn, p_w - tn[:, None])
                        f_, c_ = lt.focal(W).detach(), lt.center(W, H).detach()
                        # pts2px (utils/utils.py:15-21): x / z * f + cx - 0.5 with the camera looking down -z, y up
                        px = torch.stack([q[..., 0] / -q[..., 2] * f_ + c_[0] - 0.5, -q[..., 1] / -q[..., 2] * f_ + c_[1] - 0.5], -1)
                        pred = px - ij.detach().reshape(fr.shape[0], -1, 2).float()
                        tgt = data.fwd_flow[vsel[:, None], psel]
                        ok = (fr < c2w.shape[0] - 1)
                        cosang = ((torch.einsum("vii->v", torch.einsum("vji,vjk->vik", Rc, Rn)) - 1) / 2).clamp(-1, 1)
                        diag = {"pred_flow_mag": float(pred[ok].norm(dim=-1).mean()) if ok.any() else 0.0,
                                "target_flow_mag": float(tgt[ok].norm(dim=-1).mean()) if ok.any() else 0.0,
                                "fwd_err": float((pred - tgt)[ok].abs().sum(-1).mean()) if ok.any() else 0.0,
                                "depth_median": float(dm.median()), "teacher_depth_median": float((1.0 / data.invdepths[vsel[:, None], psel]).median()),
                                "rot_step_deg": float(torch.rad2deg(torch.acos(cosang))[ok].mean()) if ok.any() else 0.0}
                    geo_curve.append({**diag, "it": it, "field": len(lt.tensorfs) - 1, "rf_iter": int(lt.rf_iter[-1]), "refining": bool(lt.is_refining),
                                      "reg_w": float(reg_w), "flow": geo_vals[-1][0], "depth": geo_vals[-1][1], "photo": float(loss.detach()),
                                      "frames": [lo, hi], "res": int(lt.tensorfs[-1]._grid_host[0]),
                                      "pose_err": float(rel.norm(dim=-1).mean()), "est_step": float(step_est), "true_step": 0.04})
            if lt.regularize:
                tv, l1 = lt.get_reg_loss(None, 0.0, 0.0, L1_weight)         # train.py:425-429, opt.py:111-113
                total = total + tv + l1
            can_add_rf = lt.optimizer_step(total, True)
        training |= data.active_frames_bounds[1] != data.num_images
        if not lt.is_refining:                                         # train.py:438-460
            if n_added > n_overlap:                                    # the drift as of the previous iteration (LateScalar)
                drift_prev = drift.value()
                drift.request(lt.get_dist_to_last_rf())
            should_refine = (not data.has_left_frames()) or (n_added > n_overlap and (
                drift_prev > max_drift
                or data.active_frames_bounds[1] - data.active_frames_bounds[0] >= n_max_frames))
            if should_refine and (it - last_add) >= add_frames_every:
                lt.is_refining = True
                events.append((it, "refine"))
            add = data.has_left_frames() and (it - last_add + 1) % add_frames_every == 0
            if add and not should_refine and not lt.is_refining:
                lt.append_frame()
                lt.to(dev)
                data.activate_frames()
                n_added += 1
                last_add = it
        if can_add_rf:                                                 # train.py:462-474
            if data.has_left_frames():
                lt.append_rf(n_added)
                lt.to(dev)
                n_added = 0
                keep = (lt.blending_weights[:, -1] > 0)
                data.deactivate_frames(int(np.argmax(keep.cpu().numpy(), axis=0)))
                events.append((it, "append_rf"))
            else:
                training = False
        if record_all:                                                 # (tests: every iteration's photometric loss; synchronises)
            all_losses.append(float(loss.detach()))
        if gs is not None:
            host_t[res][4] += time.perf_counter() - tq4                # (drift read, frame / field appends of this iteration)
        it += 1
        if it % 250 == 0:
            pace.append((it, time.perf_counter()))                     # (the host's pace: at most four iterations ahead of the GPU)
        new_res = int(lt.tensorfs[-1]._grid_host[0])                   # (host copy: int(gridSize[0]) here drained the stream on EVERY iteration -- host and GPU took turns, 0.15 ms of every captured iteration)
        if new_res != res or not training or (max_iters is not None and it == max_iters):
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t_mark
            if it - it_mark > 0:
                e = per_res.setdefault(res, {"iters": 0, "s": 0.0})
                e["iters"] += it - it_mark
                e["s"] += dt
            mem_marks.append((it, torch.cuda.memory_allocated(dev)))
            t_mark, it_mark, res = time.perf_counter(), it, new_res
        if it % 25 == 0 and gs is not None:                               # the captured loop logs one period late
            def sink(v, it=it, res=res, nf=len(lt.tensorfs), nfr=len(lt.r_c2w)):
                losses.append(v[0])
                if log and rank == 0:
                    log(f"it {it} loss {v[0]:.4f} res {res} fields {nf} frames {nfr}")
            late_loss.put([loss], sink)
        elif it % 25 == 0:
            losses.append(float(loss.detach()))
            if log and rank == 0:
                log(f"it {it} loss {losses[-1]:.4f} res {res} fields {len(lt.tensorfs)} frames {len(lt.r_c2w)}")
    torch.cuda.synchronize(dev)
    late_geo.flush(); late_loss.flush()
    if live is not None:                                               # (probes: the live objects of the run)
        live.update(scene=lt, captured=gs, data=data)
    # data parallel: replicas must hold the SAME parameters -- every rank applied the same reduced gradients.  max |p - p of
    # rank 0| over ranks, by group (field tensors, poses, exposures / intrinsics / the rest)
    divergence = None
    if ddp:
        divergence = {"field": 0.0, "poses": 0.0, "other": 0.0}
        for name, p in lt.named_parameters():
            mine = p.detach().float().cpu() if dist.get_backend() == "gloo" else p.detach().float().clone()
            ref = mine.clone()
            dist.broadcast(ref, src=0)
            d = (mine - ref).abs().max().reshape(1) if mine.numel() else torch.zeros(1, device=mine.device)
            dist.all_reduce(d, op=dist.ReduceOp.MAX)
            key = "field" if name.startswith("tensorfs.") else ("poses" if name.startswith(("r_c2w.", "t_c2w.")) else "other")
            divergence[key] = max(divergence[key], float(d))
    # checkpoint round trip into the reference's key set (local_tensorfs.py:326-356)
    sd = {k: v.detach().clone() for k, v in lt.state_dict().items()}
    lt2 = quiet(LocalTensorfs, **{**lt.get_kwargs(), "device": dev})
    quiet(lt2.load, sd)
    sd2 = lt2.state_dict()
    same = set(sd) == set(sd2) and all(torch.equal(sd[k].to(dev), sd2[k].to(dev)) for k in sd)
    import re
    pat = re.compile(r"^(blending_weights|init_focal|focal_offset|center_rel|(r_c2w|t_c2w|exposure|world2rf)\.\d+|"
                     r"tensorfs\.\d+\.(aabb|invaabbSize|(density|app)_(plane|line)\.[012]|basis_mat\.weight|"
                     r"renderModule\.(mlp\.[02]|mlp_view\.0)\.(weight|bias)|alphaMask\.(aabb|invgridSize|alpha_volume)))$")
    keys_ok = all(pat.match(k) for k in sd)
    return {"target_image_stats": data.stats, "iterations": it, "fields": len(lt.tensorfs), "frames": len(lt.r_c2w), "events": events[:20],
            "loss_first": float(np.mean(losses[:2])) if losses else None, "loss_last": float(np.mean(losses[-2:])) if losses else None,
            "loss_curve": losses[:: max(1, len(losses) // 20)], "finite": bool(all(math.isfinite(x) for x in losses)),
            "ms_per_iteration_by_resolution": {str(r): 1e3 * e["s"] / e["iters"] for r, e in per_res.items()},
            "iterations_by_resolution": {str(r): e["iters"] for r, e in per_res.items()},
            "peak_memory_GB": torch.cuda.max_memory_allocated(dev) / 2 ** 30, "memory_marks_GB": [(i, m / 2 ** 30) for i, m in mem_marks],
            "geometric_losses": {"iterations_with_them": len(geo_vals), "flow_first_last": [v[0] for v in geo_vals if v][:1] + [v[0] for v in geo_vals if v][-1:],
                                 "depth_first_last": [v[1] for v in geo_vals if v][:1] + [v[1] for v in geo_vals if v][-1:]},
            "geo_curve": geo_curve, "geo_by_field": geo_by_field(geo_curve),
            "late_reads_wait_s": LateScalar.waited,
            "ms_per_iteration_by_250": [[b[0], round(1e3 * (b[1] - a[1]) / 250, 4)] for a, b in zip(pace, pace[1:])],
            "host_ms_per_iteration_by_resolution": {str(r): {"sample": 1e3 * v[0] / max(1, v[5]), "prepare": 1e3 * v[1] / max(1, v[5]), "step": 1e3 * v[2] / max(1, v[5]),
                                                             "schedule": 1e3 * v[3] / max(1, v[5]), "rest": 1e3 * v[4] / max(1, v[5])} for r, v in host_t.items()}, "graph": (dict(gs.stats, stage_wait_s=gs.stats.get("stage_wait_s", 0.0) + (gs.inputs.wait_s if getattr(gs, "inputs", None) is not None else 0.0)) if gs is not None else None), "all_losses": all_losses,
            "param_checksum": float(sum(p.detach().double().abs().sum() for p in lt.parameters())),
            "checkpoint_roundtrip": bool(same), "checkpoint_keys_follow_reference": bool(keys_ok), "world": world,
            "final_resolution": res, "replica_divergence": divergence}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--final", type=int, default=300, help="N_voxel_final^(1/3) (the reference: 640; BASELINE configs[4]: 500)")
    ap.add_argument("--iters-per-frame", type=int, default=60, help="the reference: 600 (opt.py:31)")
    ap.add_argument("--max-iters", type=int, default=None)
    ap.add_argument("--n-max-frames", type=int, default=12, help="frames per field before refinement (the reference: 100)")
    ap.add_argument("--json", default=None)
    ap.add_argument("--graph", action="store_true", help="the iteration as one replayed hipGraph (localrf_amd/graph_step.py)")
    ap.add_argument("--no-fuse-l1", action="store_true", help="density_L1 as an autograd node of its own (A/B of TensorVMSplit.fuse_density_L1)")
    ap.add_argument("--no-geo", action="store_true", help="without the optical-flow / monocular-depth losses")
    ap.add_argument("--backend", default="nccl", help="under torchrun: nccl (= RCCL, one rank per GPU) or gloo (ranks may share a GPU)")
    args = ap.parse_args()
    import __graft_entry__ as ge
    ddp = "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    if ddp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        first = int(os.environ.get("LOCAL_RANK", "0")) == 0
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            bar = lambda: dist.barrier(device_ids=[local])
        else:
            dist.init_process_group(args.backend)
            bar = dist.barrier
        if not first:
            bar()
    ge.build()
    if os.environ.get("LRF_TRAIN_ENG"):                                 # A/B of backward engines (lrf_debug_set_train_fwd_engine bits, include/lrf_debug.h)
        from localrf_amd import _native
        _native.lib().lrf_debug_set_train_fwd_engine(int(os.environ["LRF_TRAIN_ENG"], 0))
    if ddp and first:
        bar()
    out = run(frames=args.frames, final=args.final, iters_per_frame=args.iters_per_frame, max_iters=args.max_iters, n_max_frames=args.n_max_frames,
              dev=f"cuda:{local}", ddp=ddp, geo=not args.no_geo, graph=args.graph, fuse_l1=not args.no_fuse_l1, log=lambda m: print(m, file=sys.stderr, flush=True))
    if not ddp or int(os.environ["RANK"]) == 0:
        print(json.dumps(out, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o)))
        if args.json:
            json.dump(out, open(args.json, "w"), indent=1, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))
    if ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
