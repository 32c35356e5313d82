// Optimiser step of the path's parameters: torch.optim.Adam as the reference
// configures it (local_tensorfs.py:88-97,146,245: betas (0.9, 0.99), no weight decay, no amsgrad)
// for up to LRF_ADAM_MAX tensors in ONE launch -- planes, lines, basis, MLP and, batched with
// them, the per-frame pose / exposure tensors that the reference steps one tiny optimiser at a
// time.  Same update as torch/optim/adam.py::_single_tensor_adam:
//   m <- m + (g - m)(1 - b1);  v <- v b2 + (1 - b2) g g;
//   p <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// (the two bias corrections are evaluated on the host in double, as torch does).
#pragma once

namespace lrf {

constexpr int ADAM_TPB = 256, ADAM_VEC = 4, ADAM_CHUNK = ADAM_TPB * ADAM_VEC * 4;   // 4096 elements per block

template <int CAP>
struct AdamTableT {
  LrfAdamTensor t[CAP];
  int first_block[CAP + 1];            // prefix sum of per-tensor block counts
  short row[CAP];                      // the tensor's row of dev_scalars (its index in the caller's table)
  int count;
};
typedef AdamTableT<LRF_ADAM_MAX> AdamTable;
constexpr int ADAM_SMALL_CAP = 44;      // small tensors k_adam_pack takes along (kernel arguments stay under 4 KB beside its own table)
typedef AdamTableT<ADAM_SMALL_CAP> AdamTableS;

// (every operation spelt out with its rounding: the compiler contracts a * b + c into an fma where it sees fit, and it saw fit
// differently in k_adam_multi and in k_adam_pack -- the two kernels must produce the same bits)
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float b1, float b2, float eps,
                                      float step_size, float bc2_sqrt) {
  m = __fmaf_rn(__fsub_rn(g, m), __fsub_rn(1.0f, b1), m);
  v = __fmaf_rn(v, b2, __fmul_rn(__fmul_rn(__fsub_rn(1.0f, b2), g), g));
  const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bc2_sqrt), eps);
  p = __fmaf_rn(-step_size, __fdiv_rn(m, denom), p);
}

// dev_scalars (lrf_adam_step_dev): [count][2] = {step_size, bc2_sqrt} per tensor in DEVICE memory, read at execution time -- the
// launch can sit in a captured hipGraph whose replays step with the learning rates / bias corrections (and skip the tensors,
// bc2_sqrt <= 0) the host wrote before each replay.
template <int TPB, class TAB>
__device__ __forceinline__ void adam_table_block(const TAB& tab, int blk, float b1, float b2, float eps, const float* __restrict__ dev_scalars) {
  constexpr int CHUNK = TPB * ADAM_VEC * 4;
  int ti = 0;
  while (ti + 1 < tab.count && blk >= tab.first_block[ti + 1]) ++ti;   // <= 64 uniform steps
  LrfAdamTensor T = tab.t[ti];
  if (dev_scalars) {
    T.step_size = dev_scalars[2 * tab.row[ti]];
    T.bc2_sqrt = dev_scalars[2 * tab.row[ti] + 1];
    if (!(T.bc2_sqrt > 0.0f)) return;                // not stepped this iteration (a view nobody sampled: torch skips .grad None)
  }
  const long long base = (long long)(blk - tab.first_block[ti]) * CHUNK;
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(T.p) | reinterpret_cast<uintptr_t>(T.g) |
                        reinterpret_cast<uintptr_t>(T.m) | reinterpret_cast<uintptr_t>(T.v)) & 15) == 0;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const long long i = base + ((long long)it * TPB + threadIdx.x) * ADAM_VEC;
    if (i >= T.n) break;
    if (vec_ok && i + ADAM_VEC <= T.n) {
      float4 p = *reinterpret_cast<float4*>(T.p + i), m = *reinterpret_cast<float4*>(T.m + i),
             v = *reinterpret_cast<float4*>(T.v + i);
      const float4 g = *reinterpret_cast<const float4*>(T.g + i);
      adam1(p.x, g.x, m.x, v.x, b1, b2, eps, T.step_size, T.bc2_sqrt);
      adam1(p.y, g.y, m.y, v.y, b1, b2, eps, T.step_size, T.bc2_sqrt);
      adam1(p.z, g.z, m.z, v.z, b1, b2, eps, T.step_size, T.bc2_sqrt);
      adam1(p.w, g.w, m.w, v.w, b1, b2, eps, T.step_size, T.bc2_sqrt);
      *reinterpret_cast<float4*>(T.p + i) = p; *reinterpret_cast<float4*>(T.m + i) = m;
      *reinterpret_cast<float4*>(T.v + i) = v;
    } else {
      for (long long j = i; j < i + ADAM_VEC && j < T.n; ++j) {
        float p = T.p[j], m = T.m[j], v = T.v[j];
        adam1(p, T.g[j], m, v, b1, b2, eps, T.step_size, T.bc2_sqrt);
        T.p[j] = p; T.m[j] = m; T.v[j] = v;
      }
    }
  }
}
__global__ __launch_bounds__(ADAM_TPB) void k_adam_multi(AdamTable tab, float b1, float b2, float eps, const float* __restrict__ dev_scalars) {
  adam_table_block<ADAM_TPB>(tab, (int)blockIdx.x, b1, b2, eps, dev_scalars);
}

}  // namespace lrf

static int adam_step_impl(const LrfAdamTensor* tensors, int32_t count, const float* dev_scalars, float beta1, float beta2, float eps,
                          void* stream) {
  using namespace lrf;
  if (count < 0 || count > LRF_ADAM_MAX) return set_err("lrf_adam_step: count must be in [0, LRF_ADAM_MAX]");
  if (!count) return 0;
  if (!tensors) return set_err("lrf_adam_step: null argument");
  AdamTable tab;
  tab.count = count;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const LrfAdamTensor& t = tensors[i];
    if (!t.p || !t.g || !t.m || !t.v || t.n < 0) return set_err("lrf_adam_step: null tensor pointer or negative size");
    if (t.n > (int64_t)2000000000) return set_err("lrf_adam_step: tensor too large");
    tab.t[i] = t;
    tab.row[i] = (short)i;
    tab.first_block[i] = blocks;
    blocks += (int)((t.n + ADAM_CHUNK - 1) / ADAM_CHUNK);
  }
  tab.first_block[count] = blocks;
  if (!blocks) return 0;
  hipLaunchKernelGGL(k_adam_multi, dim3(blocks), dim3(ADAM_TPB), 0, reinterpret_cast<hipStream_t>(stream), tab,
                     beta1, beta2, eps, dev_scalars);
  LRF_HIP(hipGetLastError());
  return 0;
}

extern "C" int lrf_adam_step(const LrfAdamTensor* tensors, int32_t count, float beta1, float beta2, float eps, void* stream) {
  return adam_step_impl(tensors, count, nullptr, beta1, beta2, eps, stream);
}

extern "C" int lrf_adam_step_dev(const LrfAdamTensor* tensors, int32_t count, const float* dev_scalars, float beta1, float beta2,
                                 float eps, void* stream) {
  if (!dev_scalars) return lrf::set_err("lrf_adam_step_dev: null scalar table");
  return adam_step_impl(tensors, count, dev_scalars, beta1, beta2, eps, stream);
}

// ---------------------------------------------------------------- Adam fused with the layout refresh (round 6)
// The twelve plane / line tensors of a field are 99.9 % of its parameters, and every one of their texels is rewritten by
// Adam at the end of an iteration and re-read by the layout refresh (k_pack_planes) at the start of the next: 106 + 84 us and
// 674 + 400 MB of counter traffic per step at 500^3.  Here the thread that steps element (c, y, x) keeps the new value and the
// workgroup -- 128 consecutive texels of one row, all channels -- writes the channel-last records itself: the parameter is
// read once instead of three times (Adam, and the padded + dense appearance caches each re-read it), and written as before.
namespace lrf {

struct AdamPackSeg {
  LrfAdamTensor t;            // p: the parameter [C][H][W]; g == null: not stepped this launch (only packed)
  float* dst0; float* dst1;   // density: dst0 [H][W][8]; appearance: dst0 padded [H][W][32] (app_pc), dst1 dense [H][W][24]
  int C, H, W, row;           // row of dev_scalars, -1 = the host fields of t
};
struct AdamPackTab { AdamPackSeg s[12]; int first_block[13]; };

template <int C>
__device__ __forceinline__ void adam_pack_block(const AdamPackSeg& sg, int blk, float b1, float b2, float eps,
                                                const float* __restrict__ dev_scalars, float* s_t) {
  constexpr bool APPC = C == LRF_CA;
  constexpr int CS = APPC ? LRF_CAS : LRF_CD, ld = CS + 4;   // records stay 16-byte aligned in LDS
  const int nbx = (sg.W + 127) / 128;
  const int y = blk / nbx, x0 = (blk % nbx) * 128, t = threadIdx.x;
  const int nx = min(128, sg.W - x0);
  float step_size = sg.t.step_size, bc2 = sg.t.bc2_sqrt;
  if (dev_scalars && sg.row >= 0) { step_size = dev_scalars[2 * sg.row]; bc2 = dev_scalars[2 * sg.row + 1]; }
  const bool adam = sg.t.g != nullptr && bc2 > 0.0f;
  const size_t cs = (size_t)sg.H * sg.W;
  if (APPC) {                                                // the pad slots of the padded record
    float* d = s_t + t * ld;
#pragma unroll
    for (int q = 0; q < 4; ++q) { d[8 * q + 6] = 0.0f; d[8 * q + 7] = 0.0f; }
  }
  // Rows whose start is 16-byte aligned in all four arrays (W a multiple of 4: 64, 220, 300, 500, 640 ...): a thread takes
  // FOUR consecutive texels of C / 4 channels (c = slice, slice + 4, ...) with 16-byte loads and stores -- 24 of each per array
  // for the appearance tensors instead of 96 of four bytes; otherwise (97^3, 331^3: odd widths) one texel, all channels.
  const bool vec = (sg.W & 3) == 0 && ((reinterpret_cast<uintptr_t>(sg.t.p) | reinterpret_cast<uintptr_t>(sg.t.g) |
                                        reinterpret_cast<uintptr_t>(sg.t.m) | reinterpret_cast<uintptr_t>(sg.t.v)) & 15) == 0;
  if (vec) {
    const int q = t & 31, slice = t >> 5;                    // texels x0 + 4 q .. + 3, channels slice, slice + 4, ...
    if (4 * q < nx) {
      const size_t i0 = (size_t)y * sg.W + x0 + 4 * q;
#pragma unroll
      for (int k = 0; k < C / 4; ++k) {
        const int c = slice + 4 * k;
        const size_t i = i0 + c * cs;
        float4 p = *reinterpret_cast<const float4*>(sg.t.p + i);
        if (adam) {
          float4 m = *reinterpret_cast<const float4*>(sg.t.m + i), v = *reinterpret_cast<const float4*>(sg.t.v + i);
          const float4 g = *reinterpret_cast<const float4*>(sg.t.g + i);
          adam1(p.x, g.x, m.x, v.x, b1, b2, eps, step_size, bc2);
          adam1(p.y, g.y, m.y, v.y, b1, b2, eps, step_size, bc2);
          adam1(p.z, g.z, m.z, v.z, b1, b2, eps, step_size, bc2);
          adam1(p.w, g.w, m.w, v.w, b1, b2, eps, step_size, bc2);
          *reinterpret_cast<float4*>(sg.t.p + i) = p; *reinterpret_cast<float4*>(sg.t.m + i) = m; *reinterpret_cast<float4*>(sg.t.v + i) = v;
        }
        const int slot = APPC ? app_pc(c) : c;
        float* d = s_t + (4 * q) * ld + slot;
        d[0] = p.x; d[ld] = p.y; d[2 * ld] = p.z; d[3 * ld] = p.w;
      }
    }
  } else if (t < nx) {
    const size_t i0 = (size_t)y * sg.W + x0 + t;
    float pv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) pv[c] = sg.t.p[i0 + c * cs];
    if (adam) {
      float gv[C], mv[C], vv[C];
#pragma unroll
      for (int c = 0; c < C; ++c) { gv[c] = sg.t.g[i0 + c * cs]; mv[c] = sg.t.m[i0 + c * cs]; vv[c] = sg.t.v[i0 + c * cs]; }
#pragma unroll
      for (int c = 0; c < C; ++c) {
        adam1(pv[c], gv[c], mv[c], vv[c], b1, b2, eps, step_size, bc2);
        sg.t.p[i0 + c * cs] = pv[c]; sg.t.m[i0 + c * cs] = mv[c]; sg.t.v[i0 + c * cs] = vv[c];
      }
    }
    float* d = s_t + t * ld;
#pragma unroll
    for (int c = 0; c < C; ++c) d[APPC ? app_pc(c) : c] = pv[c];
  }
  __syncthreads();
  {
    float4* out = reinterpret_cast<float4*>(sg.dst0 + ((size_t)y * sg.W + x0) * CS);
    constexpr int q4 = CS / 4;
    for (int i = t; i < nx * q4; i += 128) out[i] = *reinterpret_cast<const float4*>(&s_t[(i / q4) * ld + 4 * (i % q4)]);
  }
  if (APPC) {                                                // the dense 24-channel record: channel c sits at app_pc(c) of the padded one
    float4* out = reinterpret_cast<float4*>(sg.dst1 + ((size_t)y * sg.W + x0) * LRF_CA);
    constexpr int q4 = LRF_CA / 4;
    for (int i = t; i < nx * q4; i += 128) {
      const float* r = s_t + (i / q4) * ld;
      const int c0 = 4 * (i % q4);
      out[i] = make_float4(r[app_pc(c0)], r[app_pc(c0 + 1)], r[app_pc(c0 + 2)], r[app_pc(c0 + 3)]);
    }
  }
}

// blocks [0, first_block[12]): the field's twelve tensors; behind them the table kernel's blocks for the small tensors
// (2048 elements each at 128 threads)
constexpr int ADAM_PACK_CHUNK = 128 * ADAM_VEC * 4;
__global__ __launch_bounds__(128) void k_adam_pack(AdamPackTab tab, AdamTableS small, float b1, float b2, float eps, const float* __restrict__ dev_scalars) {
  __shared__ __attribute__((aligned(16))) float s_t[128 * (LRF_CAS + 4)];
  if ((int)blockIdx.x >= tab.first_block[12]) {
    adam_table_block<128>(small, (int)blockIdx.x - tab.first_block[12], b1, b2, eps, dev_scalars);
    return;
  }
  int k = 0;
  while (k + 1 < 12 && (int)blockIdx.x >= tab.first_block[k + 1]) ++k;
  const AdamPackSeg& sg = tab.s[k];
  const int blk = (int)blockIdx.x - tab.first_block[k];
  if (sg.C == LRF_CA) adam_pack_block<LRF_CA>(sg, blk, b1, b2, eps, dev_scalars, s_t);
  else                adam_pack_block<LRF_CD>(sg, blk, b1, b2, eps, dev_scalars, s_t);
}

}  // namespace lrf

extern "C" int lrf_adam_step_pack(const LrfAdamTensor* tensors, int32_t count, const float* dev_scalars, float beta1, float beta2,
                                  float eps, const LrfParams* p, void* cache, void* stream) {
  using namespace lrf;
  if (count < 0 || count > LRF_ADAM_MAX) return set_err("lrf_adam_step_pack: count must be in [0, LRF_ADAM_MAX]");
  if ((count && !tensors) || !p || !cache) return set_err("lrf_adam_step_pack: null argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const Layout L = make_layout(p->grid);
  float* base = reinterpret_cast<float*>(cache);
  AdamPackTab pt;
  bool taken[LRF_ADAM_MAX] = {};
  int blocks = 0;
  for (int q = 0; q < 3; ++q) {
    const float* prm[4] = {p->density_plane[q], p->density_line[q], p->app_plane[q], p->app_line[q]};
    float* d0[4] = {base + L.dplane[q], base + L.dline[q], base + L.aplane[q], base + L.aline[q]};
    float* d1[4] = {nullptr, nullptr, base + L.aplane2[q], base + L.aline2[q]};
    const int Cs[4] = {LRF_CD, LRF_CD, LRF_CA, LRF_CA}, Hs[4] = {L.ph[q], 1, L.ph[q], 1}, Ws[4] = {L.pw[q], L.ll[q], L.pw[q], L.ll[q]};
    for (int k = 0; k < 4; ++k) {
      AdamPackSeg& sg = pt.s[4 * q + k];
      if (!prm[k]) return set_err("lrf_adam_step_pack: null parameter pointer");
      sg.t = LrfAdamTensor{const_cast<float*>(prm[k]), nullptr, nullptr, nullptr, (int64_t)Cs[k] * Hs[k] * Ws[k], 0.0f, 0.0f};
      sg.row = -1;
      for (int j = 0; j < count; ++j)
        if (tensors[j].p == prm[k]) {
          if (!tensors[j].g || !tensors[j].m || !tensors[j].v || tensors[j].n != sg.t.n) return set_err("lrf_adam_step_pack: a field tensor's Adam entry has null state or another size");
          sg.t = tensors[j]; sg.row = dev_scalars ? j : -1; taken[j] = true;
          break;
        }
      sg.dst0 = d0[k]; sg.dst1 = d1[k]; sg.C = Cs[k]; sg.H = Hs[k]; sg.W = Ws[k];
      pt.first_block[4 * q + k] = blocks;
      blocks += Hs[k] * ((Ws[k] + 127) / 128);
    }
  }
  pt.first_block[12] = blocks;
  // the small tensors (basis, colour network, per-frame poses / exposures ...): the table kernel's blocks behind the field's in
  // the same launch (up to ADAM_SMALL_CAP of them; more -- dozens of per-frame optimisers in one call -- go through k_adam_multi)
  int n_small = 0;
  for (int j = 0; j < count; ++j) {
    if (taken[j]) continue;
    const LrfAdamTensor& t = tensors[j];
    if (!t.p || !t.g || !t.m || !t.v || t.n < 0) return set_err("lrf_adam_step_pack: null tensor pointer or negative size");
    if (t.n > (int64_t)2000000000) return set_err("lrf_adam_step_pack: tensor too large");
    ++n_small;
  }
  AdamTableS small;
  small.count = 0;
  small.first_block[0] = 0;
  int tb = 0;
  if (n_small <= ADAM_SMALL_CAP) {
    for (int j = 0; j < count; ++j) {
      if (taken[j]) continue;
      small.t[small.count] = tensors[j]; small.row[small.count] = (short)j; small.first_block[small.count] = tb;
      tb += (int)((tensors[j].n + ADAM_PACK_CHUNK - 1) / ADAM_PACK_CHUNK);
      ++small.count;
    }
    small.first_block[small.count] = tb;
  } else {
    AdamTable tab;
    tab.count = 0;
    int nb = 0;
    for (int j = 0; j < count; ++j) {
      if (taken[j]) continue;
      tab.t[tab.count] = tensors[j]; tab.row[tab.count] = (short)j; tab.first_block[tab.count] = nb;
      nb += (int)((tensors[j].n + ADAM_CHUNK - 1) / ADAM_CHUNK);
      ++tab.count;
    }
    tab.first_block[tab.count] = nb;
    if (nb) hipLaunchKernelGGL(k_adam_multi, dim3(nb), dim3(ADAM_TPB), 0, st, tab, beta1, beta2, eps, dev_scalars);
  }
  hipLaunchKernelGGL(k_adam_pack, dim3(blocks + tb), dim3(128), 0, st, pt, small, beta1, beta2, eps, dev_scalars);
  // the fragment-ordered images of the colour network, from the weights the table kernel just stepped (slice 18 of k_pack_planes)
  PackTab none = {};
  PackMlp pm;
  pm.mlp = base + L.mlp; pm.mlpb = reinterpret_cast<uint32_t*>(base + L.mlpb); pm.mlpw = reinterpret_cast<uint32_t*>(base + L.mlpw);
  pm.mlpwt = reinterpret_cast<uint32_t*>(base + L.mlpwt);
  pm.mode = gen_is_default(p->fea_pe, p->view_pe, p->feature_c ? p->feature_c : LRF_FEATC) ? 0 : 1;
  hipLaunchKernelGGL(k_pack_planes, dim3(8, 8, 1), dim3(128), 0, st, none, *p, pm, 18);
  LRF_HIP(hipGetLastError());
  return 0;
}
