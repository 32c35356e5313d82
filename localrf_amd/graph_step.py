"""A training iteration of a LocalTensorfs scene as ONE replayed hipGraph (BASELINE.json configs[4]: the progressive loop).

The loop of train.py:349-474 is a few hundred microseconds of GPU work at the early grid sizes and ~2 ms of host work
(~ 80 launches through autograd, a dozen tiny optimisers, the lifecycle bookkeeping): driven from Python it is bound by the
host at every resolution (profiles/r13_train_synth_500.json: 1.9-2.2 ms per iteration from 97^3 to 500^3).
Nothing in an iteration depends on values the host reads back, so the whole of it -- pose assembly, ray generation, the
field's forward, the losses, autograd's backward through all of them, the Adam launches of the field and of every
per-frame pose / exposure optimiser, the layout refresh -- is captured once per lifecycle state and replayed:

    host, per iteration:  sample ids -> one pinned staging block -> ONE async copy -> two jitter draws -> graph launch

What changes from iteration to iteration travels through a static device block the captured kernels read at execution
time: pixel ids, view ids, their index into the assembled poses, the scalars of the loss (schedule weights), and one
(step_size, bc2_sqrt) row per optimised tensor (lrf_adam_step_dev; a zero row skips the tensor, as torch.optim.Adam skips a
parameter without gradient -- the poses of views nobody sampled, local_tensorfs.py:229-243).  What changes the SET of
tensors or shapes -- append_frame, append_rf, a grid upsample, an alpha-mask rebuild, the end of the regularised phase -- is a
lifecycle event: the next iteration runs eagerly (the same Python function, not captured) and the one after is captured
again.  The two jitter draws of the sample schedule stay outside the graph (torch.rand into static buffers, in the
reference's order) so that the random stream is the eager loop's.

Data parallel (scene.grad_sync set): two graphs per iteration -- forward + backward, then the Adam launches -- with the
gradient exchange (localrf_amd.dist.allreduce_grads, RCCL) between them on the same stream.

No reference counterpart (train.py drives the same work from Python); results equal the eager loop's
(tests/test_gpu_training.py::test_captured_iteration_matches_the_eager_loop)."""
import time

import numpy as np
import torch

from .optim import StaticAdamPlan
from .scene_ops import rows_gather

_NP = {torch.int64: np.int64, torch.int32: np.int32, torch.float32: np.float32}


class StaticInputs:
    """Named device tensors inside ONE device block, fed from a ring of pinned host blocks with one asynchronous copy per
    iteration.  `dev[name]`: the static device view captured kernels read; stage() -> (slot, {name: numpy view}) to fill;
    push(slot): enqueue the copy on the current stream.  A pinned block is reused only after its copy has run."""

    def __init__(self, device, fields, ring=4):
        self.device = torch.device(device)
        offs, off = {}, 0
        for name, (shape, dtype) in fields.items():
            nbytes = int(np.prod(shape, dtype=np.int64)) * torch.empty((), dtype=dtype).element_size()
            offs[name] = (off, nbytes, tuple(shape), dtype)
            off += (nbytes + 255) // 256 * 256
        self.nbytes = max(off, 256)
        self.blob = torch.zeros(self.nbytes, dtype=torch.uint8, device=self.device)
        self.dev = {n: self.blob[o:o + b].view(dt).view(sh) for n, (o, b, sh, dt) in offs.items()}
        self._host = [torch.zeros(self.nbytes, dtype=torch.uint8, pin_memory=True) for _ in range(ring)]
        self._np = [{n: h.numpy()[o:o + b].view(_NP[dt]).reshape(sh) for n, (o, b, sh, dt) in offs.items()} for h in self._host]
        self._ev = [None] * ring
        self._next = 0
        self.wait_s = 0.0

    def stage(self):
        k = self._next
        self._next = (k + 1) % len(self._host)
        if self._ev[k] is not None:
            t0 = time.perf_counter()
            self._ev[k].synchronize()                             # (a copy issued `ring` iterations ago: long done)
            self.wait_s += time.perf_counter() - t0               # ~0 over a run = the host never waited: the loop is host-bound
        return k, self._np[k]

    def push(self, k):
        self.blob.copy_(self._host[k], non_blocking=True)
        ev = self._ev[k]
        if ev is None:
            ev = self._ev[k] = torch.cuda.Event()
        ev.record()


class StepInputs:
    """What the loss function of a captured iteration sees besides the render's outputs (all static device tensors)."""

    def __init__(self, ray_ids, view_ids, frame, scalars, cam2world_all, start, n_views, frame32=None):
        self.ray_ids, self.view_ids, self.frame = ray_ids, view_ids, frame    # int64 [B], int64 [V], int64 [V] = view - start
        self.scalars = scalars                                    # {name: 0-dim float32 tensor}
        self.cam2world_all, self.start, self.n_views = cam2world_all, start, n_views   # get_cam2world(starting_id=start)
        self.frame32 = frame32                                    # int32 [2, V]: (view - start, view == frames assembled - 1): losses.flow_loss(frame_ids=)


class CapturedIteration:
    """scene: localrf_amd.LocalTensorfs.  loss_fn(rgb, depth, directions, ij, inputs: StepInputs) -> (total loss, {name:
    tensor to keep readable after the step}).  step(...) runs one iteration: forward, loss, backward, [grad_sync,] every Adam
    launch -- eagerly right after a lifecycle event, as a graph replay otherwise.  The caller keeps driving the lifecycle
    (scene.step_begin / step_schedule / step_finish around step(), see scripts/train_synth.py)."""

    def __init__(self, scene, W, H, batch, n_views, loss_fn, scalar_names=(), optimize_poses=True, enabled=True):
        self.scene, self.W, self.H, self.batch, self.n_views = scene, int(W), int(H), int(batch), int(n_views)
        self.loss_fn, self.scalar_names, self.optimize_poses = loss_fn, tuple(scalar_names), bool(optimize_poses)
        self.enabled = bool(enabled)
        self._sig = None
        self._graphs = None
        self._pool = None
        self._side = None
        self._eager_done = False
        self.kept = {}
        self.stats = {"eager": 0, "captures": 0, "replays": 0, "capture_host_s": 0.0, "eager_host_s": 0.0, "plans_with_intrinsics": 0}
        self.extra_signature = lambda: ()

    # ------------------------------------------------------------------ lifecycle state
    def _signature(self, start, pose_ids, tune_intrinsics):
        lt = self.scene
        f = lt.tensorfs[-1]
        return (len(lt.tensorfs), len(lt.r_c2w), tuple(int(g) for g in f._grid_host), int(f.nSamples), id(f.alphaMask),
                None if f.alphaMask is None else f.alphaMask.alpha_volume.data_ptr(), id(lt.rf_optimizer), bool(lt.is_refining),
                int(start), tuple(pose_ids), bool(tune_intrinsics), lt.grad_sync is not None,
                tuple(p.data_ptr() for p in f._param_list())) + tuple(self.extra_signature())

    def _build(self, start, pose_ids, tune_intrinsics):
        """Static buffers and the Adam plan for the current lifecycle state."""
        lt = self.scene
        dev = lt.blending_weights.device
        field = lt.tensorfs[-1]
        pairs = [(lt.rf_optimizer, p) for g in lt.rf_optimizer.param_groups for p in g["params"]]
        self._pose_params = {}
        for i in pose_ids:
            if self.optimize_poses:
                pairs += [(lt.r_optimizers[i], lt.r_c2w[i]), (lt.t_optimizers[i], lt.t_c2w[i])]
                self._pose_params[i] = (id(lt.r_c2w[i]), id(lt.t_c2w[i]))
            if lt.lr_exposure_init > 0:
                pairs.append((lt.exp_optimizers[i], lt.exposure[i]))
        if tune_intrinsics:
            pairs += [(lt.intrinsic_optimizer, p) for g in lt.intrinsic_optimizer.param_groups for p in g["params"]]
        self.plan = StaticAdamPlan(pairs)
        self._always = {id(p) for _, p in pairs} - {q for pr in self._pose_params.values() for q in pr}
        h = int(field.nSamples) // 6
        fields = {"ray_ids": ((self.batch,), torch.int64), "view_ids": ((self.n_views,), torch.int64),
                  "frame": ((self.n_views,), torch.int64), "frame32": ((2, self.n_views), torch.int32),
                  "adam": ((len(self.plan), 2), torch.float32),
                  "scalars": ((max(1, len(self.scalar_names)),), torch.float32)}
        if getattr(self, "inputs", None) is not None:
            self.stats["stage_wait_s"] = self.stats.get("stage_wait_s", 0.0) + self.inputs.wait_s
        self.inputs = StaticInputs(dev, fields)
        self.u1 = torch.empty(1, h, dtype=torch.float32, device=dev)
        self.u2 = torch.empty(1, h, dtype=torch.float32, device=dev)
        self.start = int(start)
        self._tune_intrinsics = bool(tune_intrinsics)
        self.stats["plans_with_intrinsics"] += int(bool(tune_intrinsics))
        self._graphs = None
        self._eager_done = False

    # ------------------------------------------------------------------ the iteration (eager or under capture)
    def _forward_backward(self):
        lt, d = self.scene, self.inputs.dev
        field = lt.tensorfs[-1]
        c2w_all = lt.get_cam2world(starting_id=self.start)        # one assembly: the render's poses and the flow loss's
        c2w_r = c2w_all
        if lt.reference_cross and len(lt.r_c2w) - self.start == 3:
            # the reference's dim-less torch.cross (utils/utils.py:386) fires for exactly three assembled frames: that is the
            # flow loss's assembly (train.py:388), never the render's (16 sampled views) -- assemble those without it
            lt.reference_cross = False
            try:
                c2w_r = lt.get_cam2world(starting_id=self.start)
            finally:
                lt.reference_cross = True
        c2w = rows_gather(c2w_r, d["frame"])
        field.jitter_override = (self.u1, self.u2)
        frozen = lt.freeze_intrinsics
        lt.freeze_intrinsics = frozen or not self._tune_intrinsics   # nobody reads d/d(focal, centre) unless they are tuned
        try:
            # world2rf is a constant of the scene (never in an optimiser): no gradient is formed for it here
            rgb, depth, directions, ij = lt(d["ray_ids"], d["view_ids"], self.W, self.H, is_train=True, cam2world=c2w, test_id=False,
                                            world2rf=[w.detach() for w in lt.world2rf])
            sc = {n: d["scalars"][i] for i, n in enumerate(self.scalar_names)}
            total, kept = self.loss_fn(rgb, depth, directions, ij,
                                       StepInputs(d["ray_ids"], d["view_ids"], d["frame"], sc, c2w_all, self.start, self.n_views, d["frame32"]))
        finally:
            field.jitter_override = None
            lt.freeze_intrinsics = frozen
        # (the seed of the backward: autograd's own ones_like is a fill launch per iteration -- 4.8 us of a captured replay)
        one = getattr(self, "_one", None)
        if one is None or one.device != total.device or one.dtype != total.dtype or one.shape != total.shape:
            one = self._one = torch.ones_like(total)
        total.backward(gradient=one)
        if lt.grad_sync is not None:
            # Data parallel: with a regulariser in the loss autograd leaves the density tensors' .grad in a tensor of its own
            # (regulariser + render), not in the flat buffer the exchange reduces in place.  Bringing them back HERE -- under
            # capture too -- makes the copy a node of every replay and makes the Adam graph captured next bake the flat
            # buffer's addresses: what the ranks reduce is what Adam reads (outside the graph the copy ran,
            # but the captured Adam kept reading the rank-local stray and the replicas drifted apart).
            field.rebucket_grads()
        self.kept = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in kept.items()}

    def _adam(self):
        self.plan.launch(self.inputs.dev["adam"])

    def _zero_grads(self):
        for _, p in self.plan.pairs:
            p.grad = None
        lt = self.scene
        for p in list(lt.r_c2w[self.start:]) + list(lt.t_c2w[self.start:]) + list(lt.exposure):
            p.grad = None                                         # (frames outside the plan are differentiated too; their .grad is never read)
        for p in (lt.focal_offset, lt.center_rel):
            if isinstance(p, torch.Tensor):
                p.grad = None

    def _capture(self, fn):
        # one private memory pool and one capture stream for every capture of this object: the blocks of the graph just
        # destroyed (workspace, gradient buffer: up to GBs at the late grid sizes) are taken again instead of going through
        # hipFree / hipMalloc, which synchronise the device (23 ms per capture at 500^3 without this)
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
            self._side = torch.cuda.Stream()
            self._keeper = torch.cuda.CUDAGraph()                 # a one-node graph that lives as long as this object: the allocator
            self._side.wait_stream(torch.cuda.current_stream())  # drops a private pool with its last graph
            with torch.cuda.stream(self._side):
                self._keeper.capture_begin(pool=self._pool)
                self._keeper_buf = torch.zeros(64, device=self.inputs.blob.device)
                self._keeper.capture_end()
        g = torch.cuda.CUDAGraph()
        cur = torch.cuda.current_stream()
        side = self._side
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            g.capture_begin(pool=self._pool)
            try:
                fn()
            finally:
                g.capture_end()
        cur.wait_stream(side)
        return g

    # ------------------------------------------------------------------ one iteration
    def step(self, view_list, ray_ids, scalars=None, all_poses_active=False, pose_ids=(), tune_intrinsics=False, start=0,
             global_views=None):
        """view_list: the batch's view ids (host ints, n_views of them); ray_ids: host int64 [batch] pixel ids, view-major;
        scalars: {name: float}; all_poses_active: every pose of `pose_ids` receives a gradient this iteration (the flow loss
        differentiates through all assembled frames) -- otherwise only the sampled views' poses are stepped; pose_ids /
        tune_intrinsics: what scene.step_begin returned; start: first frame of the pose assembly (the flow loss's
        starting_frame_id; every sampled view must be >= start); global_views (data parallel): the view ids of the WHOLE
        batch before localrf_amd.dist.shard_views -- the gradient exchange hands every rank the summed pose / exposure
        gradients of all of them, so all of them are stepped on every rank, as the eager path does (has-gradient flags,
        MAX over ranks); default: view_list."""
        lt = self.scene
        sig = self._signature(start, pose_ids, tune_intrinsics)
        if sig != self._sig:
            self._build(start, pose_ids, tune_intrinsics)
            self._sig = sig
        k, host = self.inputs.stage()
        host["ray_ids"][:] = np.asarray(ray_ids, dtype=np.int64).reshape(-1)
        views = np.asarray(view_list, dtype=np.int64).reshape(-1)
        host["view_ids"][:] = views
        host["frame"][:] = views - self.start
        host["frame32"][0] = views - self.start
        host["frame32"][1] = views == (len(lt.r_c2w) - self.start) - 1   # (train.py:396 compares the ABSOLUTE id with the slice's length: losses.flow_loss)
        if views.min() < self.start:
            raise IndexError("a sampled view lies before the first assembled frame")
        for i, n in enumerate(self.scalar_names):
            host["scalars"][i] = float(scalars[n])
        if all_poses_active:
            active = None
        else:
            active = set(self._always)
            stepped = views if global_views is None else np.asarray(global_views, dtype=np.int64).reshape(-1)
            for v in set(int(x) for x in stepped.tolist()):
                active.update(self._pose_params.get(v, ()))
        self.plan.host_scalars(host["adam"], active)
        self.inputs.push(k)
        torch.rand(self.u1.shape, out=self.u1)                    # the reference's two rand_like draws, in its order
        torch.rand(self.u2.shape, out=self.u2)
        sync = lt.grad_sync
        field = lt.tensorfs[-1]
        if not self.enabled or not self._eager_done:              # first iteration of this lifecycle state: plain launches
            t0 = time.perf_counter()
            self._zero_grads()
            self._forward_backward()
            if sync is not None:
                sync(lt)
            self._adam()
            self._eager_done = True
            self.stats["eager"] += 1
            self.stats["eager_host_s"] += time.perf_counter() - t0
        else:
            if self._graphs is None:
                t0 = time.perf_counter()
                self._zero_grads()                                # gradients are created inside the capture: static addresses in its pool
                # the layout refresh: written by the Adam launch itself when the field's optimiser steps it fused
                # (lrf_adam_step_pack: the cache is fresh from the eager iteration just run, and every replay leaves it fresh
                # for the next) -- otherwise lrf_pack_field is the graph's first node
                if not self.plan.packs(field):
                    field._cache_key = None
                if sync is None:
                    self._graphs = (self._capture(lambda: (self._forward_backward(), self._adam())),)
                else:
                    self._graphs = (self._capture(self._forward_backward), self._capture(self._adam))
                self.stats["captures"] += 1
                self.stats["capture_host_s"] += time.perf_counter() - t0
            t0 = time.perf_counter()
            self._graphs[0].replay()
            self.stats["replay_host_s"] = self.stats.get("replay_host_s", 0.0) + time.perf_counter() - t0    # (launch + any back-pressure of the queue)
            if sync is not None:
                field._grad_fresh = True                          # (set by the backward's Python, which a replay does not run)
                sync(lt)
                self._graphs[1].replay()
            self.stats["replays"] += 1
        self.plan.bump_versions()                                 # the parameters changed behind autograd's back
        return self.kept
