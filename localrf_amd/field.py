"""TensorVMSplit -- one VM-decomposed radiance field, MI355X-native.

Drop-in for the reference's `models.tensoRF.TensorVMSplit` (+ `models.tensorBase.TensorBase`,
`AlphaGridMask`, `MLPRender_Fea_late_view`): same constructor arguments and defaults, same
parameter names / shapes / creation order (so `torch.manual_seed(s)` yields the same field
and reference checkpoints load with `load_state_dict`), same `forward` signature and
return tuple.  Behind `forward` the work of tensorBase.py:567-636 + tensoRF.py:112-196 is
done by hand-written HIP kernels (csrc/lrf_render.hip and the .inl files it includes) called
through the C ABI in include/lrf.h.  There is no PyTorch/CPU fallback for that path: a
missing library or a non-GPU tensor raises.

Cited lines are relative to /root/reference/localTensoRF.
"""
import ctypes as C

import torch
import torch.nn.functional as F

from . import _native as N

EARLY_TERM_T_FAST = 1e-9                # TensorVMSplit.early_term_T opt-in value (north_star: "early termination")
MAT_MODE = [[0, 1], [0, 2], [1, 2]]     # tensorBase.py:274
VEC_MODE = [2, 1, 0]                    # tensorBase.py:275


class AlphaGridMask(torch.nn.Module):
    """Binary occupancy volume used to skip empty samples (tensorBase.py:38-62).  The
    lookup on the render path is fused into the march kernel; `sample_alpha` here serves
    the mask-rebuild row."""

    def __init__(self, device, aabb, alpha_volume):
        super().__init__()
        self.device = device
        self.aabb = torch.nn.Parameter(aabb.to(device), requires_grad=False)
        self.aabbSize = self.aabb[1] - self.aabb[0]
        self.invgridSize = torch.nn.Parameter(1.0 / self.aabbSize * 2, requires_grad=False)
        self.alpha_volume = torch.nn.Parameter(
            alpha_volume.view(1, 1, *alpha_volume.shape[-3:]), requires_grad=False)
        self.gridSize = torch.LongTensor(
            [alpha_volume.shape[-1], alpha_volume.shape[-2], alpha_volume.shape[-3]]).to(device)

    def normalize_coord(self, xyz):
        return (xyz - self.aabb[0]) * self.invgridSize - 1

    def sample_alpha(self, xyz):
        g = self.normalize_coord(xyz).view(1, -1, 1, 1, 3)
        return F.grid_sample(self.alpha_volume, g, align_corners=True).view(-1)

    def to(self, device):
        self.device = torch.device(device)
        return super().to(device)


class MLPRender_Fea_late_view(torch.nn.Module):
    """Parameter container of the late-view colour network (tensorBase.py:97-113):
    27 -> 128 -> 128 (+3 view) -> 3.  Its arithmetic runs inside k_shade as an MFMA chain."""

    def __init__(self, inChanel, viewpe=0, feape=0, featureC=128):
        super().__init__()
        self.in_mlpC = 2 * feape * inChanel + inChanel
        self.in_view = 2 * viewpe * 3 + 3
        self.viewpe, self.feape = viewpe, feape
        l1 = torch.nn.Linear(self.in_mlpC, featureC)
        l2 = torch.nn.Linear(featureC, featureC)
        l3 = torch.nn.Linear(featureC + self.in_view, 3)
        self.mlp = torch.nn.Sequential(l1, torch.nn.ReLU(inplace=True), l2, torch.nn.ReLU(inplace=True))
        self.mlp_view = torch.nn.Sequential(l3)
        torch.nn.init.constant_(self.mlp_view[-1].bias, 0)

    def forward(self, *a, **k):
        raise RuntimeError("MLPRender_Fea_late_view is evaluated inside the fused HIP render "
                           "kernel (lrf_render_fwd); call TensorVMSplit.forward")


class _RenderFn(torch.autograd.Function):
    """Seam between autograd and the C ABI: lrf_render_fwd / lrf_render_bwd."""

    @staticmethod
    def forward(ctx, field, rays, z, flags, floater, *params):
        # Default engine: the forward already leaves what the backward needs (density features,
        # shaded-sample lists, per-sample colours, activation rows) in a workspace owned by this
        # graph node -- a 1.5 GB worst-case reservation at 4096 x 512 (every sample shaded), 288 GB of HBM -- instead of recomputing it.
        if not flags & (N.LRF_FLAG_MLP_VALU | N.LRF_FLAG_MLP_F32):
            rgb, depth, ctx.ws, ctx.versions = field._native_forward_train(rays, z, flags)
        else:
            rgb, depth = field._native_forward(rays, z, flags, floater)
            ctx.ws, ctx.versions = None, field._param_versions()
        ctx.field, ctx.flags = field, flags
        ctx.save_for_backward(rays, z)
        # TensorVMSplit.fuse_density_L1: the regulariser's value is a third output of THIS node, so that its gradient is added
        # by this node's backward into the buffers the render gradient was scattered into (lrf_density_l1_bwd_acc) -- with a
        # node of its own autograd sums the two contributions to each of the six density tensors in six more passes, and
        # .grad ends up outside the flat gradient buffer (rebucket_grads)
        ctx.l1 = None
        l1 = None
        if getattr(field, "fuse_density_L1", False):
            ctx.set_materialize_grads(False)
            dens = tuple(p.detach() for p in params[:6])
            l1, l1_ws = _DensityL1Fn.run_forward(field, dens)
            ctx.l1 = (l1_ws, dens)
        return rgb, depth, l1

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_l1=None):
        rays, z = ctx.saved_tensors
        ws = ctx.ws
        if g_rgb is None or g_depth is None:                 # (fuse_density_L1: gradients are not materialised)
            R = rays.shape[0]
            g_rgb = torch.zeros(R, 3, device=rays.device) if g_rgb is None else g_rgb
            g_depth = torch.zeros(R, device=rays.device) if g_depth is None else g_depth
        if ctx.versions is not None and ctx.versions != ctx.field._param_versions():
            # the reference's autograd raises here too ("modified by an inplace operation"): the
            # gradients of the recorded forward cannot be formed from updated parameters
            raise RuntimeError(
                "localrf_amd: a field parameter was modified (optimizer step, upsample, load_state_dict) "
                "between forward and backward of the same render graph")
        g_rays, g_params = ctx.field._native_backward(rays, z, ctx.flags, g_rgb, g_depth, saved_ws=ws)
        ctx.ws = None                            # a second backward through this graph recomputes
        if ctx.l1 is not None and g_l1 is not None:          # the regulariser's gradient on top (same stream: behind the scatters)
            l1_ws, dens = ctx.l1
            _DensityL1Fn.run_backward(l1_ws, dens, g_l1, g_params[:6], accumulate=True)
        return (None, g_rays, None, None, None) + tuple(g_params)


class _DensityL1Fn(torch.autograd.Function):
    """Seam between autograd and lrf_density_l1_fwd / lrf_density_l1_bwd."""

    @staticmethod
    def _args(tensors):
        planes, lines = tensors[:3], tensors[3:]
        for t in tensors:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.shape[1] != 8:
                raise N.NativeError("localrf_amd: density_L1 needs contiguous fp32 planes/lines with 8 components")
        hw = (C.c_int32 * 3)(*[int(p.shape[2] * p.shape[3]) for p in planes])
        ll = (C.c_int32 * 3)(*[int(l.shape[2]) for l in lines])
        pp = (C.c_void_p * 3)(*[p.data_ptr() for p in planes])
        lp = (C.c_void_p * 3)(*[l.data_ptr() for l in lines])
        return pp, lp, hw, ll

    @staticmethod
    def run_forward(field, tensors):
        """(value [] , workspace) of the six detached density tensors."""
        lib = N.lib()
        pp, lp, hw, ll = _DensityL1Fn._args(tensors)
        dev = tensors[0].device
        ws = torch.empty(lib.lrf_density_l1_workspace(hw, ll), dtype=torch.uint8, device=dev)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        N.check(lib.lrf_density_l1_fwd(pp, lp, hw, ll, float(field.density_shift),
                                       1 if field.fea2denseAct == "relu" else 0, ws.data_ptr(), N.ptr(out), st),
                "lrf_density_l1_fwd")
        return out[0], ws

    @staticmethod
    def run_backward(ws, tensors, g_out, grads, accumulate=False):
        """d value / d tensors times g_out into `grads` (stored, or added to what they hold)."""
        pp, lp, hw, ll = _DensityL1Fn._args(tensors)
        gp = (C.c_void_p * 3)(*[g.data_ptr() for g in grads[:3]])
        gl = (C.c_void_p * 3)(*[g.data_ptr() for g in grads[3:]])
        g = g_out.detach().reshape(1).contiguous().float()
        st = torch.cuda.current_stream(g.device).cuda_stream
        fn = N.lib().lrf_density_l1_bwd_acc if accumulate else N.lib().lrf_density_l1_bwd
        N.check(fn(pp, lp, hw, ll, ws.data_ptr(), N.ptr(g), gp, gl, st), "lrf_density_l1_bwd")

    @staticmethod
    def forward(ctx, field, *tensors):
        tensors = tuple(t.detach() for t in tensors)
        out, ws = _DensityL1Fn.run_forward(field, tensors)
        ctx.save_for_backward(ws, *tensors)
        return out

    @staticmethod
    def backward(ctx, g_out):
        ws, *tensors = ctx.saved_tensors
        grads = [torch.empty_like(t) for t in tensors]
        _DensityL1Fn.run_backward(ws, tensors, g_out, grads)
        return (None, *grads)


class _TVLossFn(torch.autograd.Function):
    """Seam between autograd and lrf_tv_loss_fwd / lrf_tv_loss_bwd (3 planes then 3 lines)."""

    @staticmethod
    def _table(tensors, grads=None):
        tab = (N.LrfTvSeg * len(tensors))()
        for i, (sg, t) in enumerate(zip(tab, tensors)):
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise N.NativeError("localrf_amd: TV loss needs contiguous fp32 planes/lines")
            sg.x = t.data_ptr()
            sg.g = None if grads is None else grads[i].data_ptr()
            sg.C, sg.H, sg.W = int(t.shape[1]), int(t.shape[2]), int(t.shape[3])
            sg.scale = 1e-2 if i < 3 else 1e-3
        return tab

    @staticmethod
    def forward(ctx, weight, *tensors):
        lib = N.lib()
        tensors = tuple(t.detach() for t in tensors)
        tab = _TVLossFn._table(tensors)
        dev = tensors[0].device
        ws = torch.empty(lib.lrf_tv_workspace(tab, len(tensors)), dtype=torch.uint8, device=dev)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        N.check(lib.lrf_tv_loss_fwd(tab, len(tensors), weight, ws.data_ptr(), N.ptr(out), st), "lrf_tv_loss_fwd")
        ctx.weight = weight
        ctx.save_for_backward(*tensors)
        return out[0]

    @staticmethod
    def backward(ctx, g_out):
        tensors = ctx.saved_tensors
        grads = [torch.empty_like(t) for t in tensors]
        tab = _TVLossFn._table(tensors, grads)
        g = g_out.detach().reshape(1).contiguous().float()
        st = torch.cuda.current_stream(g.device).cuda_stream
        N.check(N.lib().lrf_tv_loss_bwd(tab, len(tensors), ctx.weight, N.ptr(g), st), "lrf_tv_loss_bwd")
        return (None, *grads)


class TensorVMSplit(torch.nn.Module):
    # constructor signature and defaults: tensorBase.py:236-257, tensoRF.py:10-12
    def __init__(self, device, aabb, gridSize, density_n_comp=8, appearance_n_comp=24, app_dim=27,
                 shadingMode="MLP_PE", alphaMask=None, near_far=[2.0, 6.0], density_shift=-10,
                 alphaMask_thres=0.001, distance_scale=25, rayMarch_weight_thres=0.001,
                 pos_pe=6, view_pe=6, fea_pe=6, featureC=128, step_ratio=2.0,
                 fea2denseAct="softplus"):
        super().__init__()
        if isinstance(density_n_comp, int):
            density_n_comp = [density_n_comp] * 3
        if isinstance(appearance_n_comp, int):
            appearance_n_comp = [appearance_n_comp] * 3
        self.density_n_comp = list(density_n_comp)
        self.app_n_comp = list(appearance_n_comp)
        self.app_dim = app_dim
        self.aabb = torch.nn.Parameter(aabb, requires_grad=False)
        self.alphaMask = alphaMask
        self.device = device
        self.density_shift = density_shift
        self.alphaMask_thres = alphaMask_thres
        self.distance_scale = distance_scale
        self.rayMarch_weight_thres = rayMarch_weight_thres
        self.fea2denseAct = fea2denseAct
        self.near_far = list(near_far)
        self.step_ratio = step_ratio
        self.matMode, self.vecMode = MAT_MODE, VEC_MODE
        self.comp_w = [1, 1, 1]
        self._check_supported(shadingMode, pos_pe, view_pe, fea_pe, featureC)
        self.update_stepSize(list(gridSize))
        self.init_svd_volume(list(gridSize), device)
        self.shadingMode, self.pos_pe, self.view_pe, self.fea_pe, self.featureC = (
            shadingMode, pos_pe, view_pe, fea_pe, featureC)
        self.renderModule = MLPRender_Fea_late_view(app_dim, view_pe, fea_pe, featureC).to(device)
        # native-side state (derived; never stored in checkpoints)
        self._cache = None
        self._cache_key = None
        self._ws = None
        # colour-MLP engine: "bf16x3" split-bf16 (hi + lo, 3-term) MFMA chain, 32 samples per wave, k_shade3 (default) |
        # "f32" exact fp32 MFMA chain (LRF_FLAG_MLP_F32) | "valu" the generic fp32 engine on the vector ALU (LRF_FLAG_MLP_VALU); a non-default
        # view_pe / fea_pe / featureC always runs the generic engine, whatever this says
        self.mlp_engine = "bf16x3"
        self.z_override = None          # tests: inject a recorded z schedule
        self.jitter_override = None     # (u1, u2): the two jitter draws of a training forward, supplied by the caller
        # early termination of the march (LrfField.term_T in include/lrf.h).  Default 0 = every sample is evaluated, the
        # reference's arithmetic (tensorBase.py:600-610).  Opt-in: EARLY_TERM_T_FAST (1e-9) skips the density gathers of a
        # ray once no later sample can pass rayMarch_weight_thres -- colours / acc identical, depth within 1e-6 absolute
        self.early_term_T = 0.0
        # opt-in: render each batch in direction-sorted order (LRF_FLAG_SORT_RAYS: one small launch; per-ray results are
        # bit-identical whatever the order), so that rays gathering the same texels run on the same XCD at the same time.
        # Measured (profiles/r11b): it cuts HBM-side traffic but not time -- k_shade3 124 vs 125 us, k_march unchanged, at
        # 300^3, 500^3 and 640^3; the sort launch costs 38 us -- so it is off by default
        self.sort_rays = False

    # ------------------------------------------------------------------ construction
    def _check_supported(self, shadingMode, pos_pe, view_pe, fea_pe, featureC):
        """This build specialises the kernels to the configuration train.py runs
        (opt.py:117-119,148-157).  Anything else fails loudly instead of falling back."""
        bad = []
        if shadingMode != "MLP_Fea_late_view":
            bad.append(f"shadingMode={shadingMode!r} (only 'MLP_Fea_late_view' is live in the "
                       "reference: tensorBase.py:627-629 passes 4 arguments)")
        if self.density_n_comp != [8, 8, 8] or self.app_n_comp != [24, 24, 24]:
            bad.append(f"n_comp {self.density_n_comp}/{self.app_n_comp} (built for [8,8,8]/[24,24,24])")
        if self.app_dim != 27:
            bad.append(f"app_dim={self.app_dim} (built for 27)")
        # view_pe / fea_pe / featureC other than opt.py's 0 / 0 / 128 render and train through the generic fp32 engine
        # (csrc/lrf_generic.inl: plain loops, 10-30 x slower than the default kernels); its limits:
        if not (0 <= view_pe <= 6 and 0 <= fea_pe <= 6 and 1 <= featureC <= 256):
            bad.append(f"view_pe={view_pe}, fea_pe={fea_pe}, featureC={featureC} (0..6, 0..6, 1..256)")
        if self.fea2denseAct not in ("softplus", "relu"):
            bad.append(f"fea2denseAct={self.fea2denseAct!r}")
        if bad:
            raise NotImplementedError("localrf_amd.TensorVMSplit: unsupported " + "; ".join(bad))

    def update_stepSize(self, gridSize):
        """tensorBase.py:317-330 (without the prints)."""
        self.aabbSize = self.aabb[1] - self.aabb[0]
        self.invaabbSize = torch.nn.Parameter(2.0 / self.aabbSize, requires_grad=False)
        self.gridSize = torch.LongTensor(gridSize).to(self.device)
        self.units = self.aabbSize / (self.gridSize - 1)
        self.stepSize = torch.mean(self.units) * self.step_ratio
        self.aabbDiag = torch.sqrt(torch.sum(torch.square(self.aabbSize)))
        self.nSamples = int((self.aabbDiag / self.stepSize).item()) + 1
        # host copies, so the render call never synchronises to read them back
        self._grid_host = [int(g) for g in gridSize]
        self._aabb_host = [float(v) for v in self.aabb.detach().reshape(-1).tolist()]
        self._aabb_key = (self.aabb.data_ptr(), self.aabb._version)
        self._z_cache = {}

    def init_svd_volume(self, res, device):
        """tensoRF.py:18-50: 0.1*randn planes [1,C,g[m1],g[m0]] and lines [1,C,g[v],1],
        drawn on the CPU generator in the reference's order, then the 72->27 basis."""
        self.density_plane, self.density_line = self.init_one_svd(self.density_n_comp, res, 0.1, device)
        self.app_plane, self.app_line = self.init_one_svd(self.app_n_comp, res, 0.1, device)
        self.basis_mat = torch.nn.Linear(sum(self.app_n_comp), self.app_dim, bias=False).to(device)

    def init_one_svd(self, n_component, gridSize, scale, device):
        planes, lines = [], []
        for i in range(3):
            m0, m1 = self.matMode[i]
            planes.append(torch.nn.Parameter(
                scale * torch.randn((1, n_component[i], gridSize[m1], gridSize[m0]))))
            lines.append(torch.nn.Parameter(
                scale * torch.randn((1, n_component[i], gridSize[self.vecMode[i]], 1))))
        return torch.nn.ParameterList(planes).to(device), torch.nn.ParameterList(lines).to(device)

    # ------------------------------------------------------------------ bookkeeping
    def get_optparam_groups(self, lr_init_spatialxyz=0.02, lr_init_network=0.001):
        """tensoRF.py:52-64 (group order is read by index at train.py:480,485)."""
        return [
            {"params": self.density_line, "lr": lr_init_spatialxyz},
            {"params": self.density_plane, "lr": lr_init_spatialxyz},
            {"params": self.app_line, "lr": lr_init_spatialxyz},
            {"params": self.app_plane, "lr": lr_init_spatialxyz},
            {"params": self.basis_mat.parameters(), "lr": lr_init_network},
            {"params": self.renderModule.parameters(), "lr": lr_init_network},
        ]

    def get_kwargs(self):
        """tensorBase.py:350-369."""
        return {
            "aabb": self.aabb, "gridSize": self.gridSize.tolist(),
            "density_n_comp": self.density_n_comp, "appearance_n_comp": self.app_n_comp,
            "app_dim": self.app_dim, "density_shift": self.density_shift,
            "alphaMask_thres": self.alphaMask_thres, "distance_scale": self.distance_scale,
            "rayMarch_weight_thres": self.rayMarch_weight_thres,
            "fea2denseAct": self.fea2denseAct, "near_far": self.near_far,
            "step_ratio": self.step_ratio, "shadingMode": self.shadingMode,
            "pos_pe": self.pos_pe, "view_pe": self.view_pe, "fea_pe": self.fea_pe,
            "featureC": self.featureC,
        }

    def to(self, device):
        """tensorBase.py:560-565."""
        self.device = torch.device(device)
        self.stepSize = self.stepSize.to(device)
        self.gridSize = self.gridSize.to(device)
        if self.alphaMask is not None:
            self.alphaMask = self.alphaMask.to(device)
        self._cache = self._cache_key = self._ws = self._ws_bwd = None
        self._cfield_key = None
        self._z_cache = {}
        return super().to(device)

    def normalize_coord(self, xyz):
        """tensorBase.py:342-345."""
        return (xyz - self.aabb[0]) * self.invaabbSize - 1

    # ------------------------------------------------------------------ native plumbing
    def _param_list(self):
        """The 19 parameter tensors in the order of LrfParams.  Read through the modules' parameter dicts:
        ParameterList.__getitem__ / Module.__getattr__ cost ~4 us per tensor, which at 16 field calls per
        scene forward (4 fields x 4 chunks, BASELINE configs[2]) was half of the host time."""
        out = []
        for pl in (self.density_plane, self.density_line, self.app_plane, self.app_line):
            d = pl._parameters
            out += [d["0"], d["1"], d["2"]]
        mods = self._modules
        rm = mods["renderModule"]._modules
        mlp, view = rm["mlp"]._modules, rm["mlp_view"]._modules
        out.append(mods["basis_mat"]._parameters["weight"])
        for lin in (mlp["0"], mlp["2"], view["0"]):
            out += [lin._parameters["weight"], lin._parameters["bias"]]
        return out

    def _require_gpu(self, t):
        if not t.is_cuda:
            raise N.NativeError(
                "localrf_amd: the render path runs only on an AMD GPU (HIP kernels); got a "
                f"{t.device} tensor. There is no CPU fallback.")

    def _c_params(self):
        ps = self._param_list()                              # only data_ptr() is taken: no detach()
        for p in ps:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise N.NativeError("localrf_amd: parameters must be contiguous fp32")
        cp = N.LrfParams()
        for i in range(3):
            cp.density_plane[i] = ps[i].data_ptr()
            cp.density_line[i] = ps[3 + i].data_ptr()
            cp.app_plane[i] = ps[6 + i].data_ptr()
            cp.app_line[i] = ps[9 + i].data_ptr()
        (cp.basis, cp.w1, cp.b1, cp.w2, cp.b2, cp.w3, cp.b3) = [p.data_ptr() for p in ps[12:]]
        cp.grid[:] = self._grid_host
        cp.fea_pe, cp.view_pe, cp.feature_c = int(self.fea_pe), int(self.view_pe), int(self.featureC)
        return cp, ps

    def _ensure_cache(self):
        """(Re)build the channel-last / fragment-ordered layout cache when any parameter
        changed (optimizer step, upsample tensoRF.py:224-233, load_state_dict, .to())."""
        lib = N.lib()
        ps = self._param_list()
        akey = (self.aabb.data_ptr(), self.aabb._version)
        if akey != getattr(self, "_aabb_key", None):          # load_state_dict / .to() replaced the bbox:
            self._aabb_host = [float(v) for v in self.aabb.detach().reshape(-1).tolist()]   # kernels read it live,
            self._aabb_key = akey                             # as normalize_coord does (tensorBase.py:342-345)
        key = tuple((p.data_ptr(), p._version) for p in ps) + tuple(self._grid_host) + tuple(self._aabb_host)
        if self._cache is not None and key == self._cache_key:
            return
        cp, keep = self._c_params()
        nbytes = lib.lrf_cache_bytes(cp.grid)
        dev = ps[0].device
        if self._cache is None or self._cache.numel() * 4 != nbytes or self._cache.device != dev:
            self._cache = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        N.check(lib.lrf_pack_field(C.byref(cp), self._cache.data_ptr(), st), "lrf_pack_field")
        self._cache_key = key

    def _fused_step_target(self):
        """(LrfParams, parameter list, cache) for lrf_adam_step_pack -- the optimiser step that leaves the layout cache holding
        the stepped values (localrf_amd.optim.FusedAdam(pack_field=...)) -- or None while no cache of the current grid's size
        exists (the first forward, after an upsample / a device move): then the step runs alone and the next forward packs."""
        if self._cache is None or self._cache_key is None:
            return None
        cp, ps = self._c_params()
        if self._cache.device != ps[0].device or self._cache.numel() * 4 != N.lib().lrf_cache_bytes(cp.grid):
            return None
        return cp, ps, self._cache

    def _mark_cache_fresh(self):
        """The cache was just rewritten from the parameters as they are now (lrf_adam_step_pack): _ensure_cache's key."""
        ps = self._param_list()
        self._cache_key = tuple((p.data_ptr(), p._version) for p in ps) + tuple(self._grid_host) + tuple(self._aabb_host)

    def _c_field(self):
        """LrfField struct for the current cache / alpha mask (rebuilt only when they change)."""
        mask = self.alphaMask
        key = (self._cache.data_ptr(), self._cache_key, id(mask),
               None if mask is None else mask.alpha_volume.data_ptr(),
               float(self.density_shift), float(self.distance_scale), float(self.rayMarch_weight_thres),
               float(self.early_term_T), tuple(self._aabb_host))
        if getattr(self, "_cfield_key", None) == key:
            return self._cfield
        f = N.LrfField()
        f.cache = self._cache.data_ptr()
        f.aabb[:] = self._aabb_host
        f.grid[:] = self._grid_host
        if mask is not None:
            vol = mask.alpha_volume.detach()
            f.alpha_vol = vol.data_ptr()
            f.alpha_dim[:] = [vol.shape[-1], vol.shape[-2], vol.shape[-3]]
            mk = (mask.aabb.data_ptr(), mask.aabb._version)
            if getattr(mask, "_aabb_host_key", None) != mk:       # read back once per mask (a rebuild makes a new mask): _c_field can
                mask._aabb_host = [float(v) for v in mask.aabb.detach().reshape(-1).tolist()]   # then run inside a stream capture
                mask._aabb_host_key = mk
            f.alpha_aabb[:] = mask._aabb_host
        else:
            f.alpha_vol = None
            f.alpha_dim[:] = [0, 0, 0]
            f.alpha_aabb[:] = self._aabb_host
        f.density_shift = float(self.density_shift)
        f.distance_scale = float(self.distance_scale)
        f.weight_thres = float(self.rayMarch_weight_thres)
        f.term_T = float(self.early_term_T)
        ps = self._param_list()[12:]
        (f.basis, f.w1, f.b1, f.w2, f.b2, f.w3, f.b3) = [p.data_ptr() for p in ps]
        f.fea_pe, f.view_pe, f.feature_c = int(self.fea_pe), int(self.view_pe), int(self.featureC)
        self._cfield, self._cfield_key = f, key
        return f

    def _workspace(self, R, S, dev):
        nbytes = N.lib().lrf_workspace_bytes(R, S)
        # one workspace per stream that renders through this field: two streams may run eval forwards of the same field side by
        # side (k_march of one batch beside k_shade3 of another: 4096-ray batches alternating over two streams 0.164 -> 0.145 ms
        # per batch, scripts/two_stream_fwd_probe.py); the layout cache they read is shared
        key = torch.cuda.current_stream(dev).cuda_stream
        if not isinstance(self._ws, dict):
            self._ws = {}
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes or ws.device != dev:
            ws = self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return ws

    def _flags(self, white_bg):
        fl = 0
        if white_bg:
            fl |= N.LRF_FLAG_WHITE_BG
        if self.fea2denseAct == "relu":
            fl |= N.LRF_FLAG_RELU_DENS
        if self.mlp_engine == "valu":
            fl |= N.LRF_FLAG_MLP_VALU
        elif self.mlp_engine == "f32":
            if self.fea_pe == 0 and self.view_pe == 0 and self.featureC == 128:    # a non-default network has one engine (generic fp32): the flag would be rejected
                fl |= N.LRF_FLAG_MLP_F32
        elif self.mlp_engine != "bf16x3":
            raise ValueError(f"unknown mlp_engine {self.mlp_engine!r}")
        if self.sort_rays:
            fl |= N.LRF_FLAG_SORT_RAYS
        return fl

    def _native_forward(self, rays, z, flags, floater, want_weights=False, out=None):
        self._require_gpu(rays)
        lib = N.lib()
        self._ensure_cache()
        rays = rays.detach().contiguous().float()
        z = z.detach().contiguous().float().view(-1)
        R, S = rays.shape[0], z.shape[0]
        dev = rays.device
        if out is not None:                 # caller-owned outputs (LocalTensorfs blends them in place)
            rgb, depth = out
            if (rgb.shape != (R, 3) or depth.shape != (R,) or rgb.dtype != torch.float32
                    or depth.dtype != torch.float32 or rgb.device != dev or depth.device != dev):
                raise ValueError("out must be (float32 [R,3], float32 [R]) on the rays' device")
        else:
            rgb = torch.empty(R, 3, dtype=torch.float32, device=dev)
            depth = torch.empty(R, dtype=torch.float32, device=dev)
        w_out = torch.empty(R, S, dtype=torch.float32, device=dev) if want_weights else None
        acc = torch.empty(R, dtype=torch.float32, device=dev) if want_weights else None
        if R == 0:
            return (rgb, depth, w_out, acc) if want_weights else (rgb, depth)
        ws = self._workspace(R, S, dev)
        f = self._c_field()
        st = torch.cuda.current_stream(dev).cuda_stream
        N.check(lib.lrf_render_fwd(C.byref(f), N.ptr(rays), N.ptr(z), R, S, flags, float(floater),
                                   N.ptr(rgb), N.ptr(depth), N.ptr(w_out), N.ptr(acc),
                                   ws.data_ptr(), st), "lrf_render_fwd")
        return (rgb, depth, w_out, acc) if want_weights else (rgb, depth)

    def _param_versions(self):
        return tuple((p.data_ptr(), p._version) for p in self._param_list())

    def _native_forward_train(self, rays, z, flags):
        """lrf_render_fwd_train: forward that keeps the backward's per-sample state in a workspace."""
        self._require_gpu(rays)
        lib = N.lib()
        self._ensure_cache()
        rays = rays.detach().contiguous().float()
        z = z.detach().contiguous().float().view(-1)
        R, S = rays.shape[0], z.shape[0]
        dev = rays.device
        rgb = torch.empty(R, 3, dtype=torch.float32, device=dev)
        depth = torch.empty(R, dtype=torch.float32, device=dev)
        if R == 0:
            return rgb, depth, None, None
        grid = (C.c_int32 * 3)(*self._grid_host)
        ws = torch.empty(lib.lrf_workspace_bytes_bwd_cfg(R, S, grid, int(self.fea_pe), int(self.view_pe), int(self.featureC), flags),
                         dtype=torch.uint8, device=dev)
        f = self._c_field()
        st = torch.cuda.current_stream(dev).cuda_stream
        N.check(lib.lrf_render_fwd_train(C.byref(f), N.ptr(rays), N.ptr(z), R, S, flags, N.ptr(rgb), N.ptr(depth),
                                         ws.data_ptr(), st), "lrf_render_fwd_train")
        return rgb, depth, ws, self._param_versions()

    def _new_grad_bucket(self, keep, R, dev, plane_events=False, events=False):
        """One zero-filled buffer, one launch: the 19 gradients (and d/d rays [R,6] behind them) are views into it, 256-byte
        aligned.  grad_bucket() / grad_chunks(): the data-parallel all-reduce runs in place on it, without copies.
        `events`: the bucket events of lrf_render_bwd_wait belong to the backward that fills this buffer -- not for an empty
        batch, and not inside a stream capture (a captured event cannot be waited for from outside the graph)."""
        sizes = [p.numel() for p in keep] + [R * 6]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + (n + 63) // 64 * 64)
        # lrf_render_bwd clears the buffer itself (LrfGrads.zero_base, with the launch that clears its bins) -- except for an
        # empty batch, where it is not called
        flat = torch.empty(offs[-1], dtype=torch.float32, device=dev) if R > 0 else torch.zeros(offs[-1], dtype=torch.float32, device=dev)
        grads = [flat[offs[i]:offs[i] + p.numel()].view(p.shape) for i, p in enumerate(keep)]
        g_rays = flat[offs[-2]:offs[-2] + R * 6].view(R, 6)
        # (the view tensors themselves are NOT kept: autograd adopts an incoming gradient as .grad without a copy only while
        # nobody else holds a reference to it)
        self._grad_flat = {"flat": flat, "params": keep, "offs": offs[:len(keep)], "n_param": offs[-2],
                           "dens": (offs[0], offs[6]), "app": (offs[6], offs[12]), "net": (offs[12], offs[-2]),
                           "app_planes": (offs[6], offs[7], offs[8]),
                           "events": bool(events), "plane_events": bool(plane_events)}
        self._grad_fresh = True                    # written by THIS backward (localrf_amd.dist reduces fresh buckets only)
        return grads, g_rays

    def _native_backward(self, rays, z, flags, g_rgb, g_depth, saved_ws=None):
        lib = N.lib()
        self._ensure_cache()
        rays = rays.detach().contiguous().float()
        z = z.detach().contiguous().float().view(-1)
        R, S = rays.shape[0], z.shape[0]
        dev = rays.device
        cp, keep = self._c_params()
        from . import dist as _dist
        plane_events = _dist.active()              # ranks exchange gradients: per-plane passes + events in the appearance scatter
        grads, g_rays = self._new_grad_bucket(keep, R, dev, plane_events,
                                              events=R > 0 and not torch.cuda.is_current_stream_capturing())
        if R == 0:
            return g_rays, grads
        cg = N.LrfGrads()
        for i in range(3):
            cg.density_plane[i] = grads[i].data_ptr()
            cg.density_line[i] = grads[3 + i].data_ptr()
            cg.app_plane[i] = grads[6 + i].data_ptr()
            cg.app_line[i] = grads[9 + i].data_ptr()
        (cg.basis, cg.w1, cg.b1, cg.w2, cg.b2, cg.w3, cg.b3) = [g.data_ptr() for g in grads[12:]]
        flat = self._grad_flat["flat"]
        cg.zero_base, cg.zero_floats = flat.data_ptr(), flat.numel()          # (offsets are multiples of 64 floats: so is the total)
        nbytes = lib.lrf_workspace_bytes_bwd_cfg(R, S, cp.grid, int(self.fea_pe), int(self.view_pe), int(self.featureC), flags)
        if saved_ws is not None:                 # filled by lrf_render_fwd_train for exactly this call
            ws = saved_ws
            flags = flags | N.LRF_FLAG_ROWS_SAVED
        else:
            if getattr(self, "_ws_bwd", None) is None or self._ws_bwd.numel() < nbytes or self._ws_bwd.device != dev:
                self._ws_bwd = None
                self._ws_bwd = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            ws = self._ws_bwd
        if plane_events:
            flags = flags | N.LRF_FLAG_PLANE_EVENTS
        f = self._c_field()
        st = torch.cuda.current_stream(dev).cuda_stream
        g_rgb_c = g_rgb.contiguous().float()         # named: must outlive the launch below
        g_depth_c = g_depth.contiguous().float()
        N.check(lib.lrf_render_bwd(C.byref(f), C.byref(cp), N.ptr(rays), N.ptr(z), R, S, flags,
                                   N.ptr(g_rgb_c), N.ptr(g_depth_c),
                                   C.byref(cg), N.ptr(g_rays), ws.data_ptr(), st), "lrf_render_bwd")
        return g_rays, grads

    def grad_bucket(self):
        """(flat, params): the parameter part of the ONE flat fp32 buffer holding the gradients of all 19 parameter tensors
        after a backward, if .grad of every parameter is still a view of it (autograd adopts the views
        when .grad was None, i.e. after zero_grad(set_to_none=True) -- what the optimisers here do).
        localrf_amd.dist.allreduce_grads reduces it in place: one collective, no copies.  None when the
        gradients were accumulated elsewhere (rebucket_grads() brings them back)."""
        gf = getattr(self, "_grad_flat", None)
        if gf is None:
            return None
        flat = gf["flat"]
        base = flat.untyped_storage().data_ptr()
        ps = [p for p in self._param_list() if p.requires_grad]
        for p in ps:
            if p.grad is None or p.grad.untyped_storage().data_ptr() != base:
                return None
        return flat[:gf["n_param"]], ps              # the parameter part: the d/d rays tail behind it is rank-local

    def rebucket_grads(self):
        """Bring the gradients back into the flat buffer of the last backward when autograd put (some of) them elsewhere: it
        sums the contributions to a parameter BEFORE it writes .grad, so with a regulariser in the loss (density_L1 / TV,
        local_tensorfs.py:316-330: their node runs first) .grad of the density tensors is the regulariser's tensor with the
        render gradient added to it, not the view lrf_render_bwd wrote.  One multi-tensor copy (device to device, the size of
        the strays) and .grad re-pointed to the views -- instead of a concatenation of the whole field and a host read-back
        on the data-parallel path.  Returns grad_bucket()."""
        gf = getattr(self, "_grad_flat", None)
        if gf is None:
            return None
        base = gf["flat"].untyped_storage().data_ptr()
        src, dst, who = [], [], []
        flat = gf["flat"]
        for p, o in zip(gf["params"], gf["offs"]):
            v = flat[o:o + p.numel()].view(p.shape)
            if not p.requires_grad:
                continue
            if p.grad is None:                       # (no gradient reached it: nothing to bring back, nothing is invented)
                return None
            if p.grad.untyped_storage().data_ptr() != base:
                if p.grad.shape != v.shape or p.grad.dtype != v.dtype or p.grad.device != v.device:
                    return None
                src.append(p.grad)
                dst.append(v)
                who.append(p)
        if src:
            torch._foreach_copy_(dst, src)
            for p, v in zip(who, dst):
                p.grad = v
            gf["events"] = False                     # the bucket events of lrf_render_bwd are behind these copies
        return self.grad_bucket()

    def grad_segments(self):
        """[(start, end)] float offsets into grad_bucket()'s flat buffer of the three branches of lrf_render_bwd, in the order
        lrf_render_bwd_wait numbers them: density planes + lines, colour network (basis, three layers), appearance planes +
        lines.  (grad_chunks() is what localrf_amd.dist reduces.)"""
        gf = getattr(self, "_grad_flat", None)
        if gf is None:
            return None
        return [gf["dens"], gf["net"], gf["app"]]

    def grad_chunks(self):
        """[(bucket, start, end)] in the order the backward finishes them: the pieces localrf_amd.dist all-reduces one by one,
        each behind lrf_render_bwd_wait(bucket).  Density planes + lines (bucket 0: the per-ray branch ends early), colour
        network (1), then the appearance tensors -- as ONE piece (2), or, when the backward ran its appearance scatter per
        plane (LRF_FLAG_PLANE_EVENTS: a process group with more than one rank exists), plane 0 (3), plane 1 (4) and plane 2
        with the three lines (2), so that only the last ~ third of the 26 MB (300^3) is exposed behind the backward."""
        gf = getattr(self, "_grad_flat", None)
        if gf is None:
            return None
        out = [(0,) + gf["dens"], (1,) + gf["net"]]
        a0, a1, a2 = gf["app_planes"]
        if gf["plane_events"]:
            out += [(3, a0, a1), (4, a1, a2), (2, a2, gf["app"][1])]
        else:
            out.append((2,) + gf["app"])
        return out

    def grad_events_valid(self):
        """True when the bucket events of lrf_render_bwd_wait belong to the backward that filled grad_bucket()."""
        gf = getattr(self, "_grad_flat", None)
        return bool(gf is not None and gf["events"])

    def _wait_bwd_bucket(self, which, stream):
        """Make `stream` wait until bucket `which` (grad_chunks numbering) of the last lrf_render_bwd on this device is final."""
        N.check(N.lib().lrf_render_bwd_wait(int(which), stream.cuda_stream), "lrf_render_bwd_wait")

    # ------------------------------------------------------------------ sampling
    def z_schedule(self, is_train, N_samples, device):
        """Ray-independent sample distances of sample_ray_contracted (tensorBase.py:419-437):
        half linear in [0,1), half inverse-depth out to 1e3, +0.1; two independent jitters in
        train mode drawn with the reference's RNG calls (rand_like on the device generator)."""
        if self.z_override is not None:
            return self.z_override.to(device).view(-1)
        n = N_samples if N_samples > 0 else self.nSamples
        h = n // 6
        if not is_train:                    # deterministic: build once per (count, device)
            zc = self._z_cache.get((h, str(device)))
            if zc is not None:
                return zc
        if torch.device(device).type == "cuda":   # one launch (lrf_z_schedule) instead of sixteen elementwise ones per training iteration
            dev = torch.device(device)
            z = torch.empty(2 * h, dtype=torch.float32, device=dev)
            if is_train and self.jitter_override is not None:   # (u1, u2) device buffers of >= h floats the caller fills (a captured
                u1, u2 = self.jitter_override                   # iteration draws them outside its graph: localrf_amd/graph_step.py)
                if u1.numel() < h or u2.numel() < h:
                    raise ValueError("jitter_override buffers are smaller than the sample schedule")
            elif is_train:                      # the reference's two rand_like draws, in its order
                u1 = torch.rand(1, h, dtype=torch.float32, device=dev)
                u2 = torch.rand(1, h, dtype=torch.float32, device=dev)
            N.check(N.lib().lrf_z_schedule(h, N.ptr(u1) if is_train else None, N.ptr(u2) if is_train else None, N.ptr(z),
                                           torch.cuda.current_stream(dev).cuda_stream), "lrf_z_schedule")
            if not is_train:
                self._z_cache[(h, str(device))] = z
            return z
        # host tensors (schedule arithmetic only; rendering needs the GPU): the reference's expression
        t = torch.linspace(0.0, h - 1, h, device=device)[None] / h
        a = t.clone()
        if is_train:
            a = a + torch.rand_like(t) / h
            t = t + torch.rand_like(t) / h
        near, far = 1.0, 1e3
        b = 1.0 / (1.0 / near * (1.0 - t) + 1.0 / far * t)
        z = (torch.cat([a, b], dim=1) + 1e-1).view(-1).contiguous()
        if not is_train:
            self._z_cache[(h, str(device))] = z
        return z

    def sample_ray_contracted(self, rays_o, rays_d, is_train=True, N_samples=-1):
        """tensorBase.py:419-443 with its signature and return values: (rays_pts [R,S,3] contracted sample positions,
        interpx [1,S] sample distances, ~mask_outbbox [R,S] all True).  forward() does this per sample inside k_march /
        k_shade3; the method is here for callers of the reference's API."""
        self._require_gpu(rays_o)
        dev = rays_o.device
        z = self.z_schedule(is_train, N_samples, dev).detach().contiguous().float().view(-1)
        ro, rd = rays_o.detach().contiguous().float(), rays_d.detach().contiguous().float()
        R, S = ro.shape[0], z.shape[0]
        pts = torch.empty(R, S, 3, dtype=torch.float32, device=dev)
        N.check(N.lib().lrf_sample_ray_contracted(N.ptr(ro), N.ptr(rd), N.ptr(z), R, S, N.ptr(pts),
                                                  torch.cuda.current_stream(dev).cuda_stream), "lrf_sample_ray_contracted")
        return pts, z[None], torch.ones(R, S, dtype=torch.bool, device=dev)

    def sample_ray(self, rays_o, rays_d, is_train=True, N_samples=-1, jitter=None):
        """AABB march (tensorBase.py:396-417), via lrf_sample_ray_aabb.  `jitter` [R] (extension)
        replaces the per-ray torch.rand draw of train mode (:408-409) so a recorded draw can be replayed."""
        self._require_gpu(rays_o)
        lib = N.lib()
        n = N_samples if N_samples > 0 else self.nSamples
        rays = torch.cat([rays_o, rays_d], -1).contiguous().float()
        R, dev = rays.shape[0], rays.device
        if jitter is not None:
            jit = jitter.to(dev).reshape(R).contiguous().float()
        else:
            jit = torch.rand(R, 1, device=dev)[:, 0].contiguous() if is_train else None
        pts = torch.empty(R, n, 3, device=dev)
        t = torch.empty(R, n, device=dev)
        inside = torch.empty(R, n, dtype=torch.uint8, device=dev)
        aabb = (C.c_float * 6)(*self._aabb_host)
        st = torch.cuda.current_stream(dev).cuda_stream
        N.check(lib.lrf_sample_ray_aabb(N.ptr(rays), aabb, float(self.stepSize), float(self.near_far[0]),
                                        float(self.near_far[1]), N.ptr(jit), R, n, N.ptr(pts), N.ptr(t),
                                        inside.data_ptr(), st), "lrf_sample_ray_aabb")
        return pts, t, inside.bool()

    # ------------------------------------------------------------------ features
    def compute_densityfeature(self, xyz_sampled):
        """tensoRF.py:112-151 on normalised coordinates [P,3] (no autograd; the training
        gradient flows through forward())."""
        self._require_gpu(xyz_sampled)
        self._ensure_cache()
        u = xyz_sampled.detach().reshape(-1, 3).contiguous().float()
        out = torch.empty(u.shape[0], device=u.device)
        f = self._c_field()
        st = torch.cuda.current_stream(u.device).cuda_stream
        N.check(N.lib().lrf_density_feature(C.byref(f), N.ptr(u), u.shape[0], N.ptr(out), st),
                "lrf_density_feature")
        return out

    def compute_appfeature(self, xyz_sampled):
        """tensoRF.py:153-196 on normalised coordinates [P,3] -> [P,27]."""
        self._require_gpu(xyz_sampled)
        self._ensure_cache()
        u = xyz_sampled.detach().reshape(-1, 3).contiguous().float()
        out = torch.empty(u.shape[0], self.app_dim, device=u.device)
        f = self._c_field()
        st = torch.cuda.current_stream(u.device).cuda_stream
        N.check(N.lib().lrf_app_feature(C.byref(f), N.ptr(u), u.shape[0], N.ptr(out), st),
                "lrf_app_feature")
        return out

    def compute_features(self, xyz_sampled):
        """tensorBase.py:333-334 declares it (no body, no caller): here the pair (density feature [P], appearance feature
        [P,27]) of normalised coordinates [P,3]."""
        return self.compute_densityfeature(xyz_sampled), self.compute_appfeature(xyz_sampled)

    def init_render_func(self, shadingMode, pos_pe, view_pe, fea_pe, featureC, device):
        """tensorBase.py:289-315.  Only the mode train.py runs exists in this build (see _check_supported)."""
        self._check_supported(shadingMode, pos_pe, view_pe, fea_pe, featureC)
        self.renderModule = MLPRender_Fea_late_view(self.app_dim, view_pe, fea_pe, featureC).to(device)
        # the kernels, the workspace size and get_kwargs() read the configuration from these attributes (ADVICE round 4:
        # stale values made the kernels index the new weights with the old shapes)
        self.shadingMode, self.pos_pe, self.view_pe, self.fea_pe, self.featureC = shadingMode, pos_pe, view_pe, fea_pe, featureC
        self._cache_key = None
        self._cfield_key = None

    def get_arange(self, idx):
        """tensoRF.py:14-16 (unused by the reference): the lattice coordinates along axis idx, pulled 1e-6 inside the box."""
        lo, hi = self.aabb[0, idx] + 1e-6, self.aabb[1, idx] - 1e-6
        step = (hi - lo) / (self.gridSize[idx] - 1)
        return torch.arange(lo, hi + step, step, device=self.device)

    def save(self, se3_poses, path):
        """tensorBase.py:371-380 (unused by train.py, which saves through LocalTensorfs.save): kwargs + state dict, the
        alpha mask as packed bits."""
        import numpy as np
        kw = self.get_kwargs()
        kw["se3_poses"] = se3_poses
        ckpt = {"kwargs": kw, "state_dict": self.state_dict()}
        if self.alphaMask is not None:
            vol = self.alphaMask.alpha_volume.bool().cpu().numpy()
            ckpt.update({"alphaMask.shape": vol.shape, "alphaMask.mask": np.packbits(vol.reshape(-1)),
                         "alphaMask.aabb": self.alphaMask.aabb.cpu()})
        torch.save(ckpt, path)

    def _not_on_the_path(self, what, where):
        raise NotImplementedError(
            f"localrf_amd.TensorVMSplit.{what}: {where} in the reference is not called by train.py / renderer.py / "
            "local_tensorfs.py (SURVEY.md s0.4) and is not part of the MI355X render path; there is no fallback.")

    def sample_ray_ndc(self, rays_o, rays_d, is_train=True, N_samples=-1):
        self._not_on_the_path("sample_ray_ndc", "tensorBase.py:382-394")

    def shrink(self, new_aabb, voxel_size=None):
        self._not_on_the_path("shrink", "tensoRF.py:236 / tensorBase.py:445")

    def filtering_rays(self, all_rays, all_rgbs, N_samples=256, chunk=10240 * 5, bbox_only=False):
        self._not_on_the_path("filtering_rays", "tensorBase.py:449-493")

    def feature2density(self, density_features):
        """tensorBase.py:495-499."""
        if self.fea2denseAct == "softplus":
            return F.softplus(density_features + self.density_shift)
        return F.relu(density_features)

    # ------------------------------------------------------------------ the hot path
    def forward(self, rays_chunk, white_bg=True, is_train=False, N_samples=-1, refine=True,
                floater_thresh=0, out=None):
        """tensorBase.py:567-636.  rays_chunk [R,6] -> (rgb_map [R,3], depth_map [R]).
        `refine` only matters when fea_pe > 0 (tensorBase.py:117-126): False feeds zeros in place of the feature encodings.
        `out=(rgb, depth)` (extension, no-grad calls only) renders into caller-owned tensors."""
        self._require_gpu(rays_chunk)
        z = self.z_schedule(is_train, N_samples, rays_chunk.device)
        use_white = bool(white_bg) or bool(is_train and torch.rand((1,)) < 0.5)   # :633
        flags = self._flags(use_white)
        if self.fea_pe > 0 and not refine:
            flags |= N.LRF_FLAG_PE_OFF
        needs_grad = torch.is_grad_enabled() and (
            rays_chunk.requires_grad or any(p.requires_grad for p in self._param_list()))
        if needs_grad:
            if floater_thresh > 0:
                raise N.NativeError("floater_thresh > 0 is an eval-only filter (train.py:107,139)")
            if out is not None:
                raise ValueError("out= is only valid when no gradient is recorded")
            rgb, depth, l1 = _RenderFn.apply(self, rays_chunk, z, flags, 0.0, *self._param_list())
            self._fused_l1 = None if l1 is None else (l1, self._param_versions())
            return rgb, depth
        return self._native_forward(rays_chunk, z, flags, float(floater_thresh), out=out)

    def render_weights(self, rays_chunk, N_samples=-1, floater_thresh=0, white_bg=True):
        """Debug/test hook: forward plus the per-sample weights and acc map."""
        z = self.z_schedule(False, N_samples, rays_chunk.device)
        return self._native_forward(rays_chunk, z, self._flags(white_bg), float(floater_thresh),
                                    want_weights=True) + (z,)

    # ------------------------------------------------------------------ "next" rows
    # Regularisers, upsample and alpha-mask rebuild sit either side of the hot path; they are
    # host-side torch code for now, same arithmetic as the reference.
    def vectorDiffs(self, vector_comps):
        """tensoRF.py:66-78."""
        total = 0
        for v in vector_comps:
            n_comp, n_size = v.shape[1:-1]
            m = v.view(n_comp, n_size)
            dotp = m @ m.transpose(-1, -2)
            off = dotp.view(-1)[1:].view(n_comp - 1, n_comp + 1)[..., :-1]
            total = total + torch.mean(torch.abs(off))
        return total

    def vector_comp_diffs(self):
        return self.vectorDiffs(self.density_line) + self.vectorDiffs(self.app_line)

    def density_L1(self):
        """tensoRF.py:83-92 through lrf_density_l1_fwd/_bwd: the lattice values are formed in
        registers instead of materialising 8 x g^3 floats per plane (same arithmetic and the
        reference's per-plane flattening orders)."""
        self._require_gpu(self.density_plane[0])
        fused = getattr(self, "_fused_l1", None)
        if fused is not None:                    # fuse_density_L1: the value the last taped forward of this field computed,
            self._fused_l1 = None                # once, and only for the parameters it was computed from
            if fused[1] == self._param_versions() and torch.is_grad_enabled():
                return fused[0]
        return _DensityL1Fn.apply(self, *self.density_plane, *self.density_line)

    def _tv_loss(self, reg, planes, lines):
        """tensoRF.py:94-110.  With the reference's TVLoss module (utils/utils.py:293-309, recognised
        by its `TVLoss_weight`) on the GPU all six tensors go through lrf_tv_loss_fwd/_bwd in one
        launch each way; any other `reg` callable is applied tensor by tensor as the reference does."""
        w = getattr(reg, "TVLoss_weight", None)
        if w is not None and planes[0].is_cuda:
            return _TVLossFn.apply(float(w), *planes, *lines)
        total = 0
        for i in range(3):
            total = total + reg(planes[i].transpose(0, 1)) * 1e-2 + reg(lines[i].transpose(0, 1)) * 1e-3
        return total

    def TV_loss_density(self, reg):
        """tensoRF.py:94-101."""
        return self._tv_loss(reg, list(self.density_plane), list(self.density_line))

    def TV_loss_app(self, reg):
        """tensoRF.py:103-110."""
        return self._tv_loss(reg, list(self.app_plane), list(self.app_line))

    @torch.no_grad()
    def up_sampling_VM(self, plane_coef, line_coef, res_target):
        """tensoRF.py:198-221: bilinear, align_corners=True (lrf_upsample_bilinear); new Parameter objects."""
        def resize(t, h2, w2):
            src = t.data.detach().contiguous().float()
            if src.device.type != "cuda":
                raise N.NativeError("localrf_amd: upsample_volume_grid needs the field on the GPU (no CPU fallback)")
            _, c, h, w = src.shape
            dst = torch.empty(1, c, int(h2), int(w2), dtype=torch.float32, device=src.device)
            N.check(N.lib().lrf_upsample_bilinear(N.ptr(src), c, h, w, N.ptr(dst), int(h2), int(w2),
                                                  torch.cuda.current_stream(src.device).cuda_stream), "lrf_upsample_bilinear")
            return torch.nn.Parameter(dst)
        for i in range(3):
            m0, m1 = self.matMode[i]
            plane_coef[i] = resize(plane_coef[i], res_target[m1], res_target[m0])
            line_coef[i] = resize(line_coef[i], res_target[self.vecMode[i]], 1)
        return plane_coef, line_coef

    @torch.no_grad()
    def upsample_volume_grid(self, res_target):
        """tensoRF.py:223-233.  The layout cache is keyed on parameter identity, so the new
        Parameters invalidate it automatically."""
        self.app_plane, self.app_line = self.up_sampling_VM(self.app_plane, self.app_line, res_target)
        self.density_plane, self.density_line = self.up_sampling_VM(
            self.density_plane, self.density_line, res_target)
        self.update_stepSize(list(res_target))

    @torch.no_grad()
    def compute_alpha(self, xyz_locs, length=1):
        """tensorBase.py:538-558, density features from the HIP gather kernel."""
        if self.alphaMask is not None:
            mask = self.alphaMask.sample_alpha(xyz_locs) > 0
        else:
            mask = torch.ones_like(xyz_locs[:, 0], dtype=bool)
        sigma = torch.zeros(xyz_locs.shape[:-1], device=xyz_locs.device)
        if mask.any():
            feat = self.compute_densityfeature(self.normalize_coord(xyz_locs[mask]))
            sigma[mask] = self.feature2density(feat)
        return 1 - torch.exp(-sigma * length).view(xyz_locs.shape[:-1])

    @torch.no_grad()
    def getDenseAlpha(self, gridSize=None):
        """tensorBase.py:501-516 as ONE launch (lrf_dense_alpha): alpha at every lattice point, through
        the current mask if there is one.  Returned in the reference's [X][Y][Z] indexing (a view of the
        [Z][Y][X] buffer the kernel writes, which is the order updateAlphaMask wants)."""
        gridSize = self.gridSize if gridSize is None else gridSize
        gx, gy, gz = (int(g) for g in gridSize)
        dev = self.aabb.device
        self._require_gpu(self.aabb)
        self._ensure_cache()
        # torch.linspace on the host, as the reference builds its lattice (:504-508), then uploaded
        lin = [torch.linspace(0, 1, g).to(dev) for g in (gx, gy, gz)]
        alpha = torch.empty(gz, gy, gx, dtype=torch.float32, device=dev)
        f = self._c_field()
        st = torch.cuda.current_stream(dev).cuda_stream
        N.check(N.lib().lrf_dense_alpha(C.byref(f), N.ptr(lin[0]), N.ptr(lin[1]), N.ptr(lin[2]), gx, gy, gz,
                                        float(self.stepSize), self._flags(False), N.ptr(alpha), st), "lrf_dense_alpha")
        return alpha.permute(2, 1, 0)

    @torch.no_grad()
    def updateAlphaMask(self, gridSize=(200, 200, 200)):
        """tensorBase.py:518-536 on the device: lrf_dense_alpha + lrf_alpha_pool_threshold, two launches,
        no host round trip and no per-slab synchronisation."""
        gx, gy, gz = (int(g) for g in gridSize)
        alpha = self.getDenseAlpha((gx, gy, gz)).permute(2, 1, 0)          # back to the kernel's [Z][Y][X]
        assert alpha.is_contiguous()
        out = torch.empty_like(alpha)
        st = torch.cuda.current_stream(alpha.device).cuda_stream
        N.check(N.lib().lrf_alpha_pool_threshold(N.ptr(alpha), gx, gy, gz, float(self.alphaMask_thres), N.ptr(out), st),
                "lrf_alpha_pool_threshold")
        self.alphaMask = AlphaGridMask(self.aabb.device, self.aabb.detach(), out)
