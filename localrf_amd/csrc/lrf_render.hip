// lrf_render.hip -- gfx950 (MI355X / CDNA4) forward kernels of the localrf render path and the C ABI declared in
// include/lrf.h.  One translation unit (this file + the .inl files it includes), built without the SLP vectoriser
// (lrf_tu.h says why).
//
// Pipeline of lrf_render_fwd (one field, R rays x S samples), two launches, no atomics on floats:
//   k_march      one wavefront per ray: contracted sampling, density VM gather (channel-last planes, 2 x float4 per tap,
//                lines staged in LDS), softplus, alpha, wave-level prefix product for transmittance, acc / depth,
//                floater filter, and in-wave compaction of the samples that pass weight > thres into the ray's list
//                (u16 index + weight); also leaves d / |d| per ray for the colour kernel.
//   k_shade3     (lrf_shade3.inl) the colour stage, 32 samples per wave on v_mfma_f32_32x32x16_bf16: tile scan,
//                appearance gather, basis -> 128 -> 128 as a register-resident split-bf16 MFMA chain, VALU head,
//                per-ray ordered sum of the tile partials + white background.
// Other colour engines behind the same ABI: k_shade (exact fp32 on v_mfma_f32_16x16x4_f32, LRF_FLAG_MLP_F32) and
// k_shade_gen (lrf_generic.inl: fp32 tile GEMMs on the vector ALU -- LRF_FLAG_MLP_VALU, and every colour-network configuration
// other than opt.py's defaults), both behind k_finalize.
// The split-bf16 helpers of the 16-sample chain (gemm_step ...) serve the training kernels of lrf_backward.inl.
//
// Reference lines (relative to /root/reference/localTensoRF) are cited at each step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <mutex>
#include "lrf_common.h"

namespace lrf {

constexpr int ITEM = 16;          // compact samples per shade work item = one 16-column MFMA tile

thread_local char g_err[512] = "";       // one error text per thread for the whole library (lrf_error_slot)
static int set_err(const char* msg, hipError_t e = hipSuccess) {
  char* slot = lrf_error_slot();
  if (e != hipSuccess) snprintf(slot, 512, "%s: %s", msg, hipGetErrorString(e));
  else snprintf(slot, 512, "%s", msg);
  return 1;
}
#define LRF_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return set_err(#call, e_); } while (0)

// ---------------------------------------------------------------------------- pack
// [C,H,W] -> [H,W,CS] channel-last; app=1: padded appearance layout (slot app_pc(c), zero pads)
struct PackSeg { const float* src; float* dst; int C, H, W, CS, app; };
struct PackTab { PackSeg s[18]; };
struct PackMlp { float* mlp; uint32_t* mlpb; uint32_t* mlpw; uint32_t* mlpwt; int mode; };   // mode 0: all four network images; 1: generic engine (only the basis^T fragments of mlpwt)
__device__ void pack_mlp_elem(const LrfParams& p, float* __restrict__ img, int idx);
__device__ void pack_mlp_bf16_elem(const LrfParams& p, uint32_t* __restrict__ img, int idx);
__device__ void pack_mlp_w32_elem(const LrfParams& p, uint32_t* __restrict__ img, int idx);
__device__ void pack_mlp_w32_t_elem(const LrfParams& p, uint32_t* __restrict__ img, int idx, int basis_only);
// The whole layout cache in ONE launch: all plane / line tensors of a field (the appearance ones twice: padded and dense;
// blockIdx.z < 18 selects; a line [C,L,1] is a plane with H = 1) and, in slice 18, the four fragment-ordered images of the
// colour network (they were four launches of ~5 us each, three of them in front of every training forward).
__global__ __launch_bounds__(128) void k_pack_planes(PackTab tab, LrfParams p, PackMlp pm, int z0 /* first slice of this launch: 0, or 18 = the network images only (lrf_adam_step_pack) */) {
  const int zsl = (int)blockIdx.z + z0;
  if (zsl == 18) {
    const int nthr = gridDim.x * gridDim.y * 128;
    for (int idx = (blockIdx.y * gridDim.x + blockIdx.x) * 128 + threadIdx.x; idx < 26880; idx += nthr) {   // >= the largest image (IMGB_ALL * 4 = 26624)
      if (pm.mode == 0) {
        pack_mlp_elem(p, pm.mlp, idx);
        pack_mlp_bf16_elem(p, pm.mlpb, idx);
        pack_mlp_w32_elem(p, pm.mlpw, idx);
      }
      pack_mlp_w32_t_elem(p, pm.mlpwt, idx, pm.mode);
    }
    return;
  }
  const PackSeg sg = tab.s[zsl];
  const float* __restrict__ src = sg.src;
  float* __restrict__ dst = sg.dst;
  const int C = sg.C, H = sg.H, W = sg.W, CS = sg.CS, app = sg.app;
  if ((int)blockIdx.y >= H) return;
  // a block packs 128 consecutive texels of one row: channel rows are read 128 floats at a time into LDS, the
  // CS-float records of those texels are one contiguous span of the destination and leave as float4 (a thread writing
  // its own record dword by dword stored 4 bytes at a 32- or 128-byte stride)
  __shared__ __attribute__((aligned(16))) float s_t[128 * (LRF_CAS + 4)];
  const int x0 = blockIdx.x * 128, y = blockIdx.y, t = threadIdx.x;
  if (x0 >= W) return;
  const int nx = min(128, W - x0);
  const int ld = CS + 4;                                    // records stay 16-byte aligned in LDS
  if (t < nx) {
    float* d = &s_t[t * ld];
    if (app) for (int q = 0; q < 4; ++q) { d[8 * q + 6] = 0.0f; d[8 * q + 7] = 0.0f; }
    for (int c = 0; c < C; ++c) d[app ? app_pc(c) : c] = src[((size_t)c * H + y) * W + x0 + t];
  }
  __syncthreads();
  float4* out = reinterpret_cast<float4*>(dst + ((size_t)y * W + x0) * CS);
  const int q4 = CS / 4, n4 = nx * q4;
  for (int i = t; i < n4; i += 128) out[i] = *reinterpret_cast<const float4*>(&s_t[(i / q4) * ld + 4 * (i % q4)]);
}
// colour network -> MFMA-fragment-ordered image (see lrf_common.h IMG_*).
// Fragment lane l = (i = l & 15, g = l >> 4): A operand row 16t'+i, K-slot g.
__device__ void pack_mlp_elem(const LrfParams& p, float* __restrict__ img, int idx) {
  if (idx >= IMG_FLOATS) return;
  float v = 0.0f;
  if (idx < IMG_W1) {                       // basis_mat.weight [27,72]   (tensoRF.py:25-27,196)
    const int e = idx - IMG_BAS;
    const int j = e & 7, lane = (e >> 3) & 63, tp = e >> 9;          // tp = t'*3 + p
    const int t1 = tp / 3, pl = tp % 3;
    const int row = 16 * t1 + (lane & 15), col = pl * LRF_CA + 6 * (lane >> 4) + j;
    if (row < LRF_APP_DIM && j < 6) v = p.basis[row * 72 + col];
  } else if (idx < IMG_W2) {                // mlp.0.weight [128,27]      (tensorBase.py:105)
    const int e = idx - IMG_W1;
    const int r = e & 3, lane = (e >> 2) & 63, tt = e >> 8;            // tt = t'*2 + t
    const int t1 = tt >> 1, t0 = tt & 1;
    const int row = 16 * t1 + (lane & 15), col = 16 * t0 + 4 * (lane >> 4) + r;
    if (col < LRF_APP_DIM) v = p.w1[row * LRF_APP_DIM + col];
  } else if (idx < IMG_W3H) {               // mlp.2.weight [128,128]     (tensorBase.py:106)
    const int e = idx - IMG_W2;
    const int r = e & 3, lane = (e >> 2) & 63, tt = e >> 8;            // tt = t'*8 + t
    const int t1 = tt >> 3, t0 = tt & 7;
    const int row = 16 * t1 + (lane & 15), col = 16 * t0 + 4 * (lane >> 4) + r;
    v = p.w2[row * LRF_FEATC + col];
  } else if (idx < IMG_B1) {                // mlp_view.0.weight[:, :128] (tensorBase.py:107)
    const int e = idx - IMG_W3H;
    const int o = e & 3, fidx = (e >> 2) & 31, g = e >> 7;
    const int feat = 16 * (fidx >> 2) + 4 * g + (fidx & 3);
    if (o < 3) v = p.w3[o * (LRF_FEATC + 3) + feat];
  } else if (idx < IMG_B2) {
    v = p.b1[idx - IMG_B1];
  } else if (idx < IMG_W3V) {
    v = p.b2[idx - IMG_B2];
  } else {                                  // view columns + bias of mlp_view.0
    const int e = idx - IMG_W3V;
    const int o = e >> 2, c = e & 3;
    if (o < 3) v = (c < 3) ? p.w3[o * (LRF_FEATC + 3) + LRF_FEATC + c] : p.b3[o];
  }
  img[idx] = v;
}

__device__ __forceinline__ unsigned short bf16_bits(float v) {
  return __builtin_bit_cast(unsigned short, (__bf16)v);       // round-to-nearest-even
}
__device__ __forceinline__ float bf16_val(unsigned short b) {
  return __uint_as_float(((unsigned)b) << 16);
}

// colour network -> split-bf16 fragment image (lrf_common.h IMGB_*).  One thread per
// 32-bit word (two bf16) of the fragment area, then the fp32 tail.
__device__ void pack_mlp_bf16_elem(const LrfParams& p, uint32_t* __restrict__ img, int idx) {
  if (idx >= IMGB_ALL * 4) return;
  if (idx >= IMGB_W3F * 4) {                                   // head fragments (k_mlp): frag = ks
    const int e = idx - IMGB_W3F * 4;
    const int u4 = e >> 2, wj = e & 3;
    const int lane = u4 & 63, part = (u4 >> 6) & 1, ks = u4 >> 7;
    const int i = lane & 15, g = lane >> 4;
    unsigned short out[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = 2 * wj + h;
      const int col = 16 * (2 * ks + (j >> 2)) + 4 * g + (j & 3);
      const float v = i < 3 ? p.w3[i * (LRF_FEATC + 3) + col] : 0.0f;
      const unsigned short hi = bf16_bits(v);
      out[h] = part ? bf16_bits(v - bf16_val(hi)) : hi;
    }
    img[idx] = (uint32_t)out[0] | ((uint32_t)out[1] << 16);
    return;
  }
  if (idx >= IMGB_TAIL * 4) {                                  // fp32 tail
    const int e = idx - IMGB_TAIL * 4;
    float v = 0.0f;
    if (e < TAIL_B1) {
      const int g = e / TAIL_W3H_GS, rem = e % TAIL_W3H_GS;
      const int o = rem & 3, fidx = rem >> 2;
      const int feat = 16 * (fidx >> 2) + 4 * g + (fidx & 3);
      if (o < 3 && fidx < 32) v = p.w3[o * (LRF_FEATC + 3) + feat];
    } else if (e < TAIL_B2) v = p.b1[e - TAIL_B1];
    else if (e < TAIL_W3V) v = p.b2[e - TAIL_B2];
    else {
      const int o = (e - TAIL_W3V) >> 2, c = (e - TAIL_W3V) & 3;
      if (o < 3) v = (c < 3) ? p.w3[o * (LRF_FEATC + 3) + LRF_FEATC + c] : p.b3[o];
    }
    img[idx] = __float_as_uint(v);
    return;
  }
  // word -> (frag, part, lane, j pair)
  const int u4 = idx >> 2, wj = idx & 3;
  const int lane = u4 & 63, part = (u4 >> 6) & 1, frag = u4 >> 7;
  const int i = lane & 15, g = lane >> 4;
  unsigned short out[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int j = 2 * wj + h;
    float v = 0.0f;
    if (frag < 6) {                                            // basis: frag = t'*3 + ks
      const int t1 = frag / 3, pl = frag % 3, row = 16 * t1 + i;
      if (j < 6 && row < LRF_APP_DIM) v = p.basis[row * 72 + pl * LRF_CA + 6 * g + j];
    } else if (frag < 14) {                                    // layer 1: frag-6 = t'
      const int t1 = frag - 6;
      const int col = 16 * (j >> 2) + 4 * g + (j & 3);
      if (col < LRF_APP_DIM) v = p.w1[(16 * t1 + i) * LRF_APP_DIM + col];
    } else {                                                   // layer 2: frag-14 = t'*4 + ks
      const int t1 = (frag - 14) >> 2, ks = (frag - 14) & 3;
      const int col = 16 * (2 * ks + (j >> 2)) + 4 * g + (j & 3);
      v = p.w2[(16 * t1 + i) * LRF_FEATC + col];
    }
    const unsigned short hi = bf16_bits(v);
    out[h] = part ? bf16_bits(v - bf16_val(hi)) : hi;
  }
  img[idx] = (uint32_t)out[0] | ((uint32_t)out[1] << 16);
}

// colour network -> w32 fragment image (lrf_common.h W32_*), for k_shade3.  One thread per 32-bit word.
__device__ void pack_mlp_w32_elem(const LrfParams& p, uint32_t* __restrict__ img, int idx) {
  if (idx >= W32_ALL_U4 * 4) return;
  if (idx >= W32_U4 * 4) {                                     // fp32 tail
    const int e = idx - W32_U4 * 4;
    float v = 0.0f;
    if (e < W32_T_B2) v = p.b1[e - W32_T_B1];
    else if (e < W32_T_W3) v = p.b2[e - W32_T_B2];
    else if (e < W32_T_B3) {
      const int c = (e - W32_T_W3) / W32_T_W3_LD, u = (e - W32_T_W3) % W32_T_W3_LD;
      if (u < LRF_FEATC + 3) v = p.w3[c * (LRF_FEATC + 3) + u];
    } else if (e < W32_T_B3 + 3) v = p.b3[e - W32_T_B3];
    img[idx] = __float_as_uint(v);
    return;
  }
  const int u4 = idx >> 2, wj = idx & 3;
  const int lane = u4 & 63, part = (u4 >> 6) & 1, frag = u4 >> 7;
  const int n = lane & 31, h = lane >> 5;
  unsigned short out[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int j = 2 * wj + hh;
    float v = 0.0f;
    if (frag < W32_W1) {                                       // basis_mat.weight [27,72] (tensoRF.py:25-27,196)
      const int c = w32_chan(h, 8 * frag + j);
      if (c >= 0 && n < LRF_APP_DIM) v = p.basis[n * 72 + c];
    } else if (frag < W32_W2) {                                // mlp.0.weight [128,27] (tensorBase.py:105)
      const int e = frag - W32_W1, m = e >> 1, q = e & 1;
      const int u = w32_unit(0, q, h, j);
      if (u < LRF_APP_DIM) v = p.w1[(32 * m + n) * LRF_APP_DIM + u];
    } else {                                                   // mlp.2.weight [128,128] (tensorBase.py:106)
      const int e = frag - W32_W2, m = e >> 3, m0 = (e >> 1) & 3, q = e & 1;
      v = p.w2[(32 * m + n) * LRF_FEATC + w32_unit(m0, q, h, j)];
    }
    const unsigned short hi = bf16_bits(v);
    out[hh] = part ? bf16_bits(v - bf16_val(hi)) : hi;
  }
  img[idx] = (uint32_t)out[0] | ((uint32_t)out[1] << 16);
}

// --------------------------------------------------------------------------- march
// One wavefront per ray.  tensorBase.py:576-622 (sampling, density, alpha2weights,
// acc/depth, floater filter, shading mask).
// LDSL: the three density lines ([L][8] floats each, 9.6 KB at 300) are staged in LDS behind the per-wave alpha
// slices, so a sample costs 24 texture-path loads instead of 36 (the march is bound by that path: TA 57 % busy)
// Several fields in one launch (LocalTensorfs' blended evaluation, local_tensorfs.py:440-474: every active field renders
// every ray): the rays of the fields are one list of nf * Rf "virtual" rays, field-major -- exactly the layout of
// lrf_scene_rays' [n_rf, R, 6] output -- and a workgroup / a tile range picks its field's DField.  All per-ray arrays
// (counts, compaction lists, partials, unit directions, outputs) are indexed by the virtual ray.
constexpr int LRF_MULTI_MAX = 4;
// Rs / lo: a chunk of a larger batch -- the rays the caller sees (inputs `rays`, outputs `rgb`, `depth`) of field k's r-th ray of
// the chunk sit at index k * Rs + lo + r of arrays laid out [n_rf][Rs]; Rs = Rf, lo = 0 when the chunk is the batch.
struct MultiF { DField f[LRF_MULTI_MAX]; int nf, Rf, Rs, lo; };
__device__ __forceinline__ int multi_io(const MultiF& m, int k, int vray) { return k * m.Rs + m.lo + (vray - k * m.Rf); }
template <bool MULTI> struct FieldArg { typedef DField type; };
template <> struct FieldArg<true> { typedef MultiF type; };
__device__ __forceinline__ const DField& field_of(const DField& f, int) { return f; }
__device__ __forceinline__ const DField& field_of(const MultiF& m, int k) { return m.f[k]; }

template <bool LDSL, bool MULTI = false>
__global__ __launch_bounds__(1024) void k_march(
    typename FieldArg<MULTI>::type fin, const float* __restrict__ rays, const float* __restrict__ z, int R, int S,
    uint32_t flags, float floater,
    float* __restrict__ depth, float* __restrict__ acc_ws, float* __restrict__ w_all,
    int* __restrict__ ncomp, uint16_t* __restrict__ cidx, float* __restrict__ cw,
    float* __restrict__ feat_out /* [R,S] density feature, -inf where not evaluated; or null */) {
  extern __shared__ float s_alpha_all[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nb = gridDim.x;                              // XCD-aware block order, see tile_walk_begin
  const int lb = (nb % 8 == 0) ? (int)(blockIdx.x % 8) * (nb / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int nw = blockDim.x >> 6;                          // rays per workgroup: 4, 8 or 16 (launch_march)
  const int ray = lb * nw + wave;
  int fk_ = 0;
  if constexpr (MULTI) fk_ = min((lb * nw) / fin.Rf, fin.nf - 1);   // the workgroup's field (Rf is a multiple of nw: one field per workgroup)
  const DField& f = field_of(fin, fk_);
  int io_ray = ray;                                        // where the caller keeps this ray (rays in, depth out)
  if constexpr (MULTI) io_ray = multi_io(fin, fk_, ray);
  const float* s_line[3] = {nullptr, nullptr, nullptr};
  if (LDSL) {                                              // lines behind the alpha slices (whole block: before any return)
    float* base = s_alpha_all + (size_t)nw * S;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int n = f.ll[p] * LRF_CD / 4;
      float4* dst = reinterpret_cast<float4*>(base);
      const float4* src = reinterpret_cast<const float4*>(f.dline[p]);
      for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
      s_line[p] = base;
      base += f.ll[p] * LRF_CD;
    }
    __syncthreads();
  }
  if (ray >= R) return;
  float* s_alpha = s_alpha_all + (size_t)wave * S;
  const int oray = f.perm ? f.perm[ray] : io_ray;          // where the caller sees this ray (ray sorting / a chunk of a multi-field batch)

  const float* rp = rays + (size_t)io_ray * 6;
  const float o[3] = {rp[0], rp[1], rp[2]};
  const float dn = sqrtf(rp[3] * rp[3] + rp[4] * rp[4] + rp[5] * rp[5]);      // :578-580
  const float dh[3] = {rp[3] / dn, rp[4] / dn, rp[5] / dn};
  const bool relu = flags & LRF_FLAG_RELU_DENS;
  const int nchunk = (S + 63) >> 6;

  // pass A: alpha per sample -> LDS.
  //
  // Early termination: `carry` is the transmittance entering the next 64-sample chunk (a wave-level
  // product of this chunk's 1 - alpha + 1e-10, the same value in every lane).  Once it is below
  // f.term_T no later sample can pass weight > weight_thres (w_k = alpha_k T_k <= T_k <= carry), so the
  // remaining density gathers are skipped and those samples count as empty (alpha = 0, feature "not
  // evaluated"): exactly what the reference computes for a field that is empty behind that point --
  // the forced last sample (:24) picks up the remaining transmittance.  acc is unchanged; sum(w z)
  // moves by at most carry * z[S-1] (term_T = 1e-9, z <= 1000.1: 1e-6 absolute against depth * |d| >= 0.1).
  // term_T = 0 disables it.
  {
    float carry = 1.0f;
    bool live = true;                                      // wave-uniform
    for (int c = 0; c < nchunk; ++c) {
      const int k = (c << 6) + lane;
      float alpha = 0.0f, fk = -INFINITY;
      if (live && k < S - 1) {                             // last sample is never valid (:600)
        const float zk = z[k];
        float x[3], u[3];
        sample_point(f, o, dh, zk, x, u);
        bool valid = true;
        if (f.alpha_vol) valid = alpha_mask_sample(f, x[0], x[1], x[2]) > 0.0f;   // :593-598
        if (valid) {
          fk = density_feature_m<LDSL>(f, u, s_line);
          const float sigma = feature2density(fk, f.density_shift, relu);            // :603-608
          const float dist = z[k + 1] - zk;                                          // :584-587
          alpha = 1.0f - expf(-sigma * dist * f.distance_scale);                     // :610
        }
      }
      if (k < S) {
        s_alpha[k] = alpha;
        if (feat_out) feat_out[(size_t)ray * S + k] = fk;
      }
      if (live && f.term_T > 0.0f) {
        float v = (k < S - 1) ? (1.0f - alpha + 1e-10f) : 1.0f;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v *= __shfl_xor(v, d, 64);
        carry *= v;
        if (__builtin_amdgcn_readfirstlane(__float_as_int(carry)) < __float_as_int(f.term_T)) live = false;
      }
    }
  }
  // (each wave only touches its own LDS slice: no barrier needed, LDS ops are in order per wave)

  float acc = 0.0f, dsum = 0.0f, kbar = 0.0f;
  int nsh = 0;
  const int npass = floater > 0.0f ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
    const bool emit = (pass == npass - 1);
    float carry = 1.0f, a_acc = 0.0f, a_d = 0.0f, a_k = 0.0f;
    for (int c = 0; c < nchunk; ++c) {
      const int k = (c << 6) + lane;
      float alpha = 0.0f;
      if (k < S) {
        alpha = s_alpha[k];
        if (pass == 1 && (float)k < kbar * floater) alpha = 0.0f;                  // :617-619
        if (k == S - 1) alpha = 1.0f;                                               // :24
      }
      const float v = (k < S) ? (1.0f - alpha + 1e-10f) : 1.0f;                    // :25-29
      float excl, total;
      wave_scan_prod(v, lane, excl, total);
      const float T = carry * excl;
      carry *= total;
      const float w = alpha * T;                                                    // :31
      if (pass == 0) {
        a_acc += w;
        a_d += (k < S) ? w * z[k] : 0.0f;
        a_k += w * (float)k;
      }
      if (emit) {
        if (w_all && k < S) w_all[(size_t)oray * S + k] = w;
        const bool sh = (k < S) && (w > f.weight_thres);                            // :622
        const unsigned long long m = __ballot(sh);
        if (sh) {
          const int pos = nsh + __popcll(m & ((1ull << lane) - 1ull));
          cidx[(size_t)ray * S + pos] = (uint16_t)k;
          cw[(size_t)ray * S + pos] = w;
        }
        nsh += __popcll(m);
      }
    }
    if (pass == 0) {
      acc = wave_sum(a_acc);                                                        // :614
      dsum = wave_sum(a_d);
      kbar = wave_sum(a_k);
    }
  }
  if (lane == 0) {
    depth[oray] = dsum / dn;                                                        // :615
    acc_ws[ray] = acc;
    ncomp[ray] = nsh;
    if (f.rdir) *reinterpret_cast<float4*>(f.rdir + (size_t)ray * 4) = make_float4(dh[0], dh[1], dh[2], dn);
  }
}

// Exclusive prefix sum of per-ray tile counts: toff[r] = sum_{q<r} ceil(ncomp[q]/16),
// toff[R] = total.  One 1024-thread block; replaces a device-wide atomic work queue (one
// atomic word saturates at ~88 dequeues/us on this chip -- 52K pulls cost 0.59 ms).
__global__ __launch_bounds__(1024) void k_scan_tiles(const int* __restrict__ ncomp, int R, int* __restrict__ toff) {
  __shared__ int s_wave[16];
  __shared__ int s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < R; base += 1024) {
    const int r = base + tid;
    const int v = r < R ? (ncomp[r] + ITEM - 1) / ITEM : 0;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int q = 0; q < wave; ++q) woff += s_wave[q];
    const int carry = s_carry;
    if (r < R) toff[r] = carry + woff + incl - v;
    __syncthreads();
    if (tid == 1023) s_carry = carry + woff + incl;
    __syncthreads();
  }
  if (tid == 0) toff[R] = s_carry;
}

// --------------------------------------------------------------------------- shade
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// appearance products for this lane's 6 channels of each plane (tensoRF.py:153-195):
// lane (s, g) of a tile owns channels 6g..6g+5 of plane p -> K-slot g of MFMA k-step (p, j).
// appearance products of lane group g for plane p: eight slots (six channels + two zero pads)
// per tap, read as two aligned float4 of the padded 128-byte texel (tensoRF.py:153-195)
template <int p>
__device__ __forceinline__ void gather_app6_plane(const DField& f, const float u[3], int g, float X[8]) {
  int x0, x1, y0, y1, l0, l1; float tx, ty, tl;
  tap1d(u[MAT0[p]], f.pw[p], x0, x1, tx);
  tap1d(u[MAT1[p]], f.ph[p], y0, y1, ty);
  tap1d(u[VEC[p]],  f.ll[p], l0, l1, tl);
  const float* pl = f.aplane[p] + 8 * g;
  const float* q00 = pl + ((size_t)y0 * f.pw[p] + x0) * LRF_CAS;
  const float* q10 = pl + ((size_t)y0 * f.pw[p] + x1) * LRF_CAS;
  const float* q01 = pl + ((size_t)y1 * f.pw[p] + x0) * LRF_CAS;
  const float* q11 = pl + ((size_t)y1 * f.pw[p] + x1) * LRF_CAS;
  const float* r0 = f.aline[p] + (size_t)l0 * LRF_CAS + 8 * g;
  const float* r1 = f.aline[p] + (size_t)l1 * LRF_CAS + 8 * g;
  const float w00 = (1.0f - tx) * (1.0f - ty), w10 = tx * (1.0f - ty);
  const float w01 = (1.0f - tx) * ty,          w11 = tx * ty;
  const float wl0 = 1.0f - tl, wl1 = tl;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float4 a = ld4(q00 + 4 * h), b = ld4(q10 + 4 * h), c = ld4(q01 + 4 * h), d = ld4(q11 + 4 * h);
    const float4 e = ld4(r0 + 4 * h), q = ld4(r1 + 4 * h);
    X[4 * h]     = (a.x * w00 + b.x * w10 + c.x * w01 + d.x * w11) * (e.x * wl0 + q.x * wl1);
    X[4 * h + 1] = (a.y * w00 + b.y * w10 + c.y * w01 + d.y * w11) * (e.y * wl0 + q.y * wl1);
    X[4 * h + 2] = (a.z * w00 + b.z * w10 + c.z * w01 + d.z * w11) * (e.z * wl0 + q.z * wl1);
    X[4 * h + 3] = (a.w * w00 + b.w * w10 + c.w * w01 + d.w * w11) * (e.w * wl0 + q.w * wl1);
  }
}
__device__ __forceinline__ void gather_app6(const DField& f, const float u[3], int g, float X[3][6]) {
  float v[8];
  gather_app6_plane<0>(f, u, g, v);
#pragma unroll
  for (int j = 0; j < 6; ++j) X[0][j] = v[j];
  gather_app6_plane<1>(f, u, g, v);
#pragma unroll
  for (int j = 0; j < 6; ++j) X[1][j] = v[j];
  gather_app6_plane<2>(f, u, g, v);
#pragma unroll
  for (int j = 0; j < 6; ++j) X[2][j] = v[j];
}

// Static, contiguous split of the T = toff[R] tiles over all waves of the grid (tiles cost
// the same, so this is balanced to one tile; consecutive tiles of a wave belong to the same
// ray -> L1/L2 locality; no atomics).  TileWalk yields (ray, j0) for tiles [t, t_end).
struct TileWalk {
  int t, t_end, ray, next_off;
};
__device__ __forceinline__ TileWalk tile_walk_begin(const int* __restrict__ toff, int R) {
  TileWalk tw;
  const int T = toff[R];
  const long long waves = (long long)gridDim.x * (blockDim.x >> 6);
  // XCD-aware order: hardware places block b on XCD b % 8 (speed only, never correctness), so the
  // logical block id below gives every XCD one contiguous eighth of the tile list -- rays that are
  // neighbours in the caller's order (pixels of one image, or direction-sorted rays) share an L2.
  const int nb = gridDim.x;
  const int lb = (nb % 8 == 0) ? (int)(blockIdx.x % 8) * (nb / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const long long wid = (long long)lb * (blockDim.x >> 6) + (threadIdx.x >> 6);
  tw.t = __builtin_amdgcn_readfirstlane((int)(wid * T / waves));
  tw.t_end = __builtin_amdgcn_readfirstlane((int)((wid + 1) * T / waves));
  // largest ray with toff[ray] <= t
  int lo = 0, hi = R;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (toff[mid] <= tw.t) lo = mid; else hi = mid;
  }
  tw.ray = __builtin_amdgcn_readfirstlane(lo);            // (wave-uniform: kept in scalar registers across the tile loop)
  tw.next_off = __builtin_amdgcn_readfirstlane(toff[lo + 1]);
  return tw;
}
// advance to the ray owning tile tw.t (skips rays without shaded samples)
__device__ __forceinline__ void tile_walk_seek(TileWalk& tw, const int* __restrict__ toff) {
  while (tw.next_off <= tw.t) { ++tw.ray; tw.next_off = __builtin_amdgcn_readfirstlane(toff[tw.ray + 1]); }
}

__global__ __launch_bounds__(1024) void k_shade(
    DField f, const float* __restrict__ rays, const float* __restrict__ z, int S,
    const int* __restrict__ toff, int R,
    const int* __restrict__ ncomp, const uint16_t* __restrict__ cidx, const float* __restrict__ cw,
    float* __restrict__ part, int pmax) {
  __shared__ __attribute__((aligned(16))) float img[IMG_FLOATS];
  {
    const float4* src = reinterpret_cast<const float4*>(f.mlp);
    float4* dst = reinterpret_cast<float4*>(img);
    for (int i = threadIdx.x; i < IMG_FLOATS / 4; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, s = lane & 15, g = lane >> 4;
  TileWalk tw = tile_walk_begin(toff, R);
  for (; tw.t < tw.t_end; ++tw.t) {
    asm volatile("" ::: "memory");     // keep LDS weight reads inside the loop (no LICM -> no spills)
    tile_walk_seek(tw, toff);
    const int ray = __builtin_amdgcn_readfirstlane(tw.ray);
    const int j0 = (tw.t - (tw.next_off - (ncomp[ray] + ITEM - 1) / ITEM)) * ITEM;
    const int cnt = min(ITEM, ncomp[ray] - j0);

    const float* rp = rays + (size_t)ray * 6;
    const float o[3] = {rp[0], rp[1], rp[2]};
    const float dn = sqrtf(rp[3] * rp[3] + rp[4] * rp[4] + rp[5] * rp[5]);
    const float dh[3] = {rp[3] / dn, rp[4] / dn, rp[5] / dn};
    // view-direction part of mlp_view.0 + bias: constant per ray (tensorBase.py:131-132;
    // viewdirs detached :628)
    float vb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float4 wv = *reinterpret_cast<const float4*>(&img[IMG_W3V + 4 * c]);
      vb[c] = wv.w + wv.x * dh[0] + wv.y * dh[1] + wv.z * dh[2];
    }
    {
      const bool valid = s < cnt;
      const size_t ci = (size_t)ray * S + j0 + (valid ? s : 0);
      const int k = cidx[ci];
      const float w = valid ? cw[ci] : 0.0f;
      float x[3], u[3];
      sample_point(f, o, dh, z[k], x, u);
      float X[3][6];
      gather_app6(f, u, g, X);

      // basis: feat = basis_mat.weight @ (plane*line)            (tensoRF.py:196)
      f32x4 fe[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int t1 = 0; t1 < 2; ++t1) {
          const float* ap = &img[IMG_BAS + ((t1 * 3 + p) * 64 + lane) * 8];
          const float4 a0 = *reinterpret_cast<const float4*>(ap);
          const float2 a1 = *reinterpret_cast<const float2*>(ap + 4);
          fe[t1] = mfma4(a0.x, X[p][0], fe[t1]);
          fe[t1] = mfma4(a0.y, X[p][1], fe[t1]);
          fe[t1] = mfma4(a0.z, X[p][2], fe[t1]);
          fe[t1] = mfma4(a0.w, X[p][3], fe[t1]);
          fe[t1] = mfma4(a1.x, X[p][4], fe[t1]);
          fe[t1] = mfma4(a1.y, X[p][5], fe[t1]);
        }
      }
      // layer 1: relu(W1 feat + b1)                               (tensorBase.py:129-130)
      f32x4 h1[8];
#pragma unroll
      for (int t1 = 0; t1 < 8; ++t1) h1[t1] = *reinterpret_cast<const f32x4*>(&img[IMG_B1 + 16 * t1 + 4 * g]);
#pragma unroll
      for (int t0 = 0; t0 < 2; ++t0) {
#pragma unroll
        for (int t1 = 0; t1 < 8; ++t1) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(&img[IMG_W1 + ((t1 * 2 + t0) * 64 + lane) * 4]);
#pragma unroll
          for (int r = 0; r < 4; ++r) h1[t1] = mfma4(a[r], fe[t0][r], h1[t1]);
        }
      }
#pragma unroll
      for (int t1 = 0; t1 < 8; ++t1)
#pragma unroll
        for (int r = 0; r < 4; ++r) h1[t1][r] = fmaxf(h1[t1][r], 0.0f);
      // layer 2: relu(W2 h1 + b2)
      f32x4 h2[8];
#pragma unroll
      for (int t1 = 0; t1 < 8; ++t1) h2[t1] = *reinterpret_cast<const f32x4*>(&img[IMG_B2 + 16 * t1 + 4 * g]);
#pragma unroll
      for (int t0 = 0; t0 < 8; ++t0) {
#pragma unroll
        for (int t1 = 0; t1 < 8; ++t1) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(&img[IMG_W2 + ((t1 * 8 + t0) * 64 + lane) * 4]);
#pragma unroll
          for (int r = 0; r < 4; ++r) h2[t1] = mfma4(a[r], h1[t0][r], h2[t1]);
        }
      }
      // head: sigmoid(W3 [h2 ; dhat] + b3) on the VALU            (tensorBase.py:131-133)
      float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
#pragma unroll
      for (int t1 = 0; t1 < 8; ++t1)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float hv = fmaxf(h2[t1][r], 0.0f);
          const float4 wv = *reinterpret_cast<const float4*>(&img[IMG_W3H + (g * 32 + t1 * 4 + r) * 4]);
          o0 += hv * wv.x; o1 += hv * wv.y; o2 += hv * wv.z;
        }
      o0 += __shfl_xor(o0, 16, 64); o1 += __shfl_xor(o1, 16, 64); o2 += __shfl_xor(o2, 16, 64);
      o0 += __shfl_xor(o0, 32, 64); o1 += __shfl_xor(o1, 32, 64); o2 += __shfl_xor(o2, 32, 64);
      float cr = w / (1.0f + expf(-(o0 + vb[0])));                 // :133, :632
      float cg = w / (1.0f + expf(-(o1 + vb[1])));
      float cb = w / (1.0f + expf(-(o2 + vb[2])));
#pragma unroll
      for (int dd = 1; dd < 16; dd <<= 1) {
        cr += __shfl_xor(cr, dd, 64); cg += __shfl_xor(cg, dd, 64); cb += __shfl_xor(cb, dd, 64);
      }
      if (lane == 0) {
        float* pp = part + ((size_t)ray * pmax + j0 / ITEM) * 3;
        pp[0] = cr; pp[1] = cg; pp[2] = cb;
      }
    }
  }
}

// ------------------------------------------------------------------ split-bf16 products, 16-sample tiles
// The training kernels (lrf_backward.inl) run every GEMM of the colour network on v_mfma_f32_16x16x32_bf16 with both
// operands split into hi + lo bf16 (x ~ hi + lo to 2^-16): acc += Al*Bh + Ah*Bl + Ah*Bh, fp32 accumulate -- ~1e-5
// relative error per product, inside the 1e-4 parity budget (tests/test_gpu_parity.py).
// Rounds 1-2 issued these MFMAs by hand (tied accumulators, four wait states behind each, operands held for 48 more)
// because renders differed from run to run and a late operand read was suspected.  It was not that
// (scripts/ubench/mfma_war.hip: the pipe reads its sources at issue; profiles/r08b: the cause was packed fp32 VALU
// arithmetic): the chain is plain builtins now, the three terms term-major over the NT accumulators of a step so that
// consecutive MFMAs never depend on each other.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split8(const float v[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];             // v_cvt_pk_bf16_f32, round-to-nearest-even
    hi[j] = h;
    lo[j] = (__bf16)(v[j] - (float)h);
  }
}
template <int NT>
__device__ __forceinline__ void settle(f32x4*) {}          // (was: wait states behind a hand-issued chain)
__device__ __forceinline__ void mfma_bf16_acc(bf16x8 a, bf16x8 b, f32x4& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 lds_frag(const uint4* img, int frag, int part, int lane) {
  return __builtin_bit_cast(bf16x8, img[(frag * 2 + part) * 64 + lane]);
}
// acc[t1] += A(frag0 + t1*stride) x B for t1 in [0, NT), three-term split product; the accumulators are taken G at a
// time (the training kernels run 1024-thread workgroups: 128 registers per lane, a fragment pair costs eight)
template <int NT, int G = (NT < 2 ? NT : 2)>
__device__ __forceinline__ void gemm_step(const uint4* img, int frag0, int stride, int lane,
                                          bf16x8 bh, bf16x8 bl, f32x4* acc) {
#pragma unroll
  for (int t0 = 0; t0 < NT; t0 += G) {
    bf16x8 ah[G], al[G];
#pragma unroll
    for (int t1 = 0; t1 < G; ++t1)
      if (t0 + t1 < NT) { ah[t1] = lds_frag(img, frag0 + (t0 + t1) * stride, 0, lane); al[t1] = lds_frag(img, frag0 + (t0 + t1) * stride, 1, lane); }
#pragma unroll
    for (int t1 = 0; t1 < G; ++t1) if (t0 + t1 < NT) mfma_bf16_acc(al[t1], bh, acc[t0 + t1]);
#pragma unroll
    for (int t1 = 0; t1 < G; ++t1) if (t0 + t1 < NT) mfma_bf16_acc(ah[t1], bl, acc[t0 + t1]);
#pragma unroll
    for (int t1 = 0; t1 < G; ++t1) if (t0 + t1 < NT) mfma_bf16_acc(ah[t1], bh, acc[t0 + t1]);
  }
}

}  // namespace lrf
#include "lrf_tiles.inl"
#include "lrf_shade3.inl"
namespace lrf {

// (the plain-loop engine of LRF_FLAG_MLP_VALU is k_shade_gen of lrf_generic.inl: the same kernel serves every non-default
// network configuration)

// rgb_map = sum_k w_k rgb_k (+ 1 - acc)                        (tensorBase.py:632-634)
__global__ void k_finalize(int R, int pmax, uint32_t flags, const int* __restrict__ ncomp,
                           const float* __restrict__ acc, const float* __restrict__ part,
                           float* __restrict__ rgb, float* __restrict__ acc_out, const int* __restrict__ perm) {
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= R) return;
  const int oray = perm ? perm[ray] : ray;
  const int nit = (ncomp[ray] + ITEM - 1) / ITEM;
  float r = 0.0f, g = 0.0f, b = 0.0f;
  for (int i = 0; i < nit; ++i) {
    const float* pp = part + ((size_t)ray * pmax + i) * 3;
    r += pp[0]; g += pp[1]; b += pp[2];
  }
  if (flags & LRF_FLAG_WHITE_BG) {
    const float bg = 1.0f - acc[ray];
    r += bg; g += bg; b += bg;
  }
  rgb[(size_t)oray * 3 + 0] = r; rgb[(size_t)oray * 3 + 1] = g; rgb[(size_t)oray * 3 + 2] = b;
  if (acc_out) acc_out[oray] = acc[ray];
}

// ---------------------------------------------------------------------------- ray sorting (LRF_FLAG_SORT_RAYS)
// Rays are independent, so the batch may be rendered in any order; the caller's order is usually the worst one (random
// pixels of a few views; random directions in the benchmark): the 256 CUs then gather from everywhere at once and an
// XCD's 4 MB L2 sees the whole 35-160 MB field.  One 1024-thread workgroup sorts the batch by a 15-bit direction key
// (cube face of d / |d|, then the Morton code of the two in-face coordinates at 6 bits each) with a bitonic network on
// (key << 16 | index) words in LDS, writes the sorted copy of the rays and the slot -> ray permutation.  Everything after
// it indexes rays by slot; only what the caller sees (rgb, depth, weights, g_rgb, g_depth, g_rays) goes through `perm`.
// With the XCD-aware block order of the kernels, an XCD then renders one contiguous eighth of the direction sphere.
__device__ __forceinline__ uint32_t ray_dir_key(const float* __restrict__ rp) {
  const float x = rp[3], y = rp[4], z = rp[5];
  const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
  int face; float m, u, v;
  if (ax >= ay && ax >= az) { face = x < 0.0f ? 1 : 0; m = ax; u = y; v = z; }
  else if (ay >= az)        { face = y < 0.0f ? 3 : 2; m = ay; u = x; v = z; }
  else                      { face = z < 0.0f ? 5 : 4; m = az; u = x; v = y; }
  m = fmaxf(m, 1e-30f);
  const int iu = min(63, max(0, (int)((u / m * 0.5f + 0.5f) * 64.0f)));
  const int iv = min(63, max(0, (int)((v / m * 0.5f + 0.5f) * 64.0f)));
  uint32_t mort = 0;
#pragma unroll
  for (int b = 0; b < 6; ++b) mort |= (uint32_t)((iu >> b) & 1) << (2 * b) | (uint32_t)((iv >> b) & 1) << (2 * b + 1);
  return ((uint32_t)face << 12) | mort;
}
__global__ __launch_bounds__(1024) void k_sort_rays(const float* __restrict__ rays, int R, int N /* pow2 >= R */,
                                                    float* __restrict__ rays_s, int* __restrict__ perm) {
  extern __shared__ uint32_t s_key[];
  const int tid = threadIdx.x;
  for (int i = tid; i < N; i += 1024) s_key[i] = i < R ? (ray_dir_key(rays + (size_t)i * 6) << 16) | (uint32_t)i : 0xffffffffu;
  __syncthreads();
  for (int k = 2; k <= N; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < N / 2; t += 1024) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;        // the pair (lo, lo + j)
        const uint32_t a = s_key[lo], b = s_key[hi];
        const bool up = (lo & k) == 0;
        if ((a > b) == up) { s_key[lo] = b; s_key[hi] = a; }
      }
      __syncthreads();
    }
  for (int i = tid; i < R; i += 1024) {
    const int src = (int)(s_key[i] & 0xffffu);
    perm[i] = src;
    const float2* sp = reinterpret_cast<const float2*>(rays + (size_t)src * 6);
    float2* dp = reinterpret_cast<float2*>(rays_s + (size_t)i * 6);
    dp[0] = sp[0]; dp[1] = sp[1]; dp[2] = sp[2];
  }
}

// ----------------------------------------------------------------- stand-alone pieces
__global__ void k_density_feature(DField f, const float* __restrict__ u, int P, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float uu[3] = {u[(size_t)i * 3], u[(size_t)i * 3 + 1], u[(size_t)i * 3 + 2]};
  out[i] = density_feature(f, uu);
}

__global__ void k_app_feature(DField f, const float* __restrict__ u, int P, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float uu[3] = {u[(size_t)i * 3], u[(size_t)i * 3 + 1], u[(size_t)i * 3 + 2]};
  float acc[LRF_APP_DIM];
  for (int a = 0; a < LRF_APP_DIM; ++a) acc[a] = 0.0f;
  for (int p = 0; p < 3; ++p) {
    int x0, x1, y0, y1, l0, l1; float tx, ty, tl;
    tap1d(uu[MAT0[p]], f.pw[p], x0, x1, tx);
    tap1d(uu[MAT1[p]], f.ph[p], y0, y1, ty);
    tap1d(uu[VEC[p]],  f.ll[p], l0, l1, tl);
    const float* pl = f.aplane[p];
    for (int c = 0; c < LRF_CA; ++c) {
      const int pc = app_pc(c);
      const float v = pl[((size_t)y0 * f.pw[p] + x0) * LRF_CAS + pc] * ((1.0f - tx) * (1.0f - ty))
                    + pl[((size_t)y0 * f.pw[p] + x1) * LRF_CAS + pc] * (tx * (1.0f - ty))
                    + pl[((size_t)y1 * f.pw[p] + x0) * LRF_CAS + pc] * ((1.0f - tx) * ty)
                    + pl[((size_t)y1 * f.pw[p] + x1) * LRF_CAS + pc] * (tx * ty);
      const float l = f.aline[p][(size_t)l0 * LRF_CAS + pc] * (1.0f - tl) + f.aline[p][(size_t)l1 * LRF_CAS + pc] * tl;
      const float xv = v * l;
      for (int a = 0; a < LRF_APP_DIM; ++a) acc[a] += f.basis[a * 72 + p * LRF_CA + c] * xv;
    }
  }
  for (int a = 0; a < LRF_APP_DIM; ++a) out[(size_t)i * LRF_APP_DIM + a] = acc[a];
}

// tensorBase.py:396-417
// TensorBase.sample_ray_contracted (tensorBase.py:419-443) as a public call: pts[r][k] = contract(o_r + d_r z_k).  (The render
// kernels do this per sample in registers; this entry exists for callers of the reference's method.)
__global__ __launch_bounds__(256) void k_sample_contracted(const float* __restrict__ ro, const float* __restrict__ rd,
                                                           const float* __restrict__ z, int R, int S, float* __restrict__ pts) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)R * S) return;
  const int r = (int)(i / S), k = (int)(i % S);
  const float zk = z[k];
  float x = ro[3 * r] + rd[3 * r] * zk, y = ro[3 * r + 1] + rd[3 * r + 1] * zk, w = ro[3 * r + 2] + rd[3 * r + 2] * zk;
  contract3(x, y, w);
  pts[3 * i] = x; pts[3 * i + 1] = y; pts[3 * i + 2] = w;
}

// The ray-independent sample distances of sample_ray_contracted (tensorBase.py:419-437): z[i] = t_i (+ u1_i / h) + 0.1 and
// z[h + i] = 1 / ((1 - s_i) + s_i / 1000) + 0.1 with s_i = t_i (+ u2_i / h), t_i = i / h -- the reference's sixteen
// elementwise launches per training iteration as one; every operation rounded separately, in the reference's order, with
// IEEE divisions: bit-identical to the reference evaluated on the CPU (what the goldens record).  A reference run on a GPU
// evaluates tensor / python_scalar as tensor * (1 / scalar) (ATen's scalar fast path): the linear half then differs by <= 1 ulp
// for h that is not a power of two, the inverse-depth half by up to ~1e-5 relative at its far end (the reciprocal amplifies an ulp of
// s near 1) -- the reference's own GPU / CPU difference, below the tolerances of the path, but not bit-identical
// (test_z_schedule_kernel_vs_reference_expression measures both).
__global__ __launch_bounds__(256) void k_z_schedule(int h, const float* __restrict__ u1, const float* __restrict__ u2, float* __restrict__ z) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= h) return;
  const float fh = (float)h;
  const float t = __fdiv_rn((float)i, fh);
  const float a = u1 ? __fadd_rn(t, __fdiv_rn(u1[i], fh)) : t;
  const float s = u2 ? __fadd_rn(t, __fdiv_rn(u2[i], fh)) : t;
  const float den = __fadd_rn(__fmul_rn(1.0f, __fsub_rn(1.0f, s)), __fmul_rn(1.0f / 1e3f, s));   // 1 / near * (1 - t) + 1 / far * t, near = 1
  z[i] = __fadd_rn(a, 1e-1f);
  z[h + i] = __fadd_rn(__fdiv_rn(1.0f, den), 1e-1f);
}

__global__ void k_sample_ray_aabb(const float* __restrict__ rays, float lo0, float lo1, float lo2,
                                  float hi0, float hi1, float hi2, float step, float near_, float far_,
                                  const float* __restrict__ jitter, int R, int N,
                                  float* __restrict__ pts, float* __restrict__ tt, uint8_t* __restrict__ inside) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)R * N) return;
  const int ray = (int)(i / N), k = (int)(i % N);
  const float* rp = rays + (size_t)ray * 6;
  const float lo[3] = {lo0, lo1, lo2}, hi[3] = {hi0, hi1, hi2};
  float tmin = -INFINITY;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float v = rp[3 + a] == 0.0f ? 1e-6f : rp[3 + a];
    const float ra = (hi[a] - rp[a]) / v, rb = (lo[a] - rp[a]) / v;
    tmin = fmaxf(tmin, fminf(ra, rb));
  }
  tmin = fminf(fmaxf(tmin, near_), far_);
  float rng = (float)k;
  if (jitter) rng += jitter[ray];
  const float t = tmin + step * rng;
  bool in = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float pv = rp[a] + rp[3 + a] * t;
    pts[i * 3 + a] = pv;
    in = in && !(lo[a] > pv) && !(pv > hi[a]);
  }
  tt[i] = t;
  inside[i] = in ? 1 : 0;
}

// ----------------------------------------------------------------------- host side
static DField make_dfield(const LrfField* f) {
  DField d;
  const Layout L = make_layout(f->grid);
  const float* base = reinterpret_cast<const float*>(f->cache);
  for (int p = 0; p < 3; ++p) {
    d.dplane[p] = base + L.dplane[p]; d.dline[p] = base + L.dline[p];
    d.aplane[p] = base + L.aplane[p]; d.aline[p] = base + L.aline[p];
    d.pw[p] = L.pw[p]; d.ph[p] = L.ph[p]; d.ll[p] = L.ll[p];
  }
  d.mlp = base + L.mlp;
  d.mlpb = reinterpret_cast<const uint4*>(base + L.mlpb);
  for (int p = 0; p < 3; ++p) { d.aplane2[p] = base + L.aplane2[p]; d.aline2[p] = base + L.aline2[p]; }
  d.mlpw = reinterpret_cast<const uint4*>(base + L.mlpw);
  d.mlpwt = reinterpret_cast<const uint4*>(base + L.mlpwt);
  d.alpha_vol = f->alpha_vol;
  d.ax = f->alpha_dim[0]; d.ay = f->alpha_dim[1]; d.az = f->alpha_dim[2];
  for (int a = 0; a < 3; ++a) {
    d.lo[a] = f->aabb[a];
    d.hi[a] = f->aabb[3 + a];
    d.inv[a] = 2.0f / (f->aabb[3 + a] - f->aabb[a]);                      // tensorBase.py:321
    d.m_lo[a] = f->alpha_aabb[a];
    d.m_inv[a] = 1.0f / (f->alpha_aabb[3 + a] - f->alpha_aabb[a]) * 2.0f;  // tensorBase.py:44
  }
  d.density_shift = f->density_shift; d.distance_scale = f->distance_scale; d.weight_thres = f->weight_thres;
  d.term_T = f->term_T > 0.0f ? f->term_T : 0.0f;
  d.dump = nullptr;
  d.rdir = nullptr;
  d.perm = nullptr;
  d.basis = f->basis; d.w1 = f->w1; d.b1 = f->b1; d.w2 = f->w2; d.b2 = f->b2; d.w3 = f->w3; d.b3 = f->b3;
  d.fea_pe = f->fea_pe; d.view_pe = f->view_pe; d.fc = f->feature_c ? f->feature_c : LRF_FEATC;
  return d;
}

struct Workspace {
  int* toff; int* ncomp; float* acc; uint16_t* cidx; float* cw; float* part;
  float* rdir;             // k_march -> k_shade3: unit direction and length per ray
  int* perm; float* rays_s;  // ray sorting: slot -> ray, sorted copy of the rays
  int pmax; size_t bytes;
};
static size_t up256(size_t x) { return (x + 255) & ~size_t(255); }
static Workspace carve(void* ws, int R, int S) {
  Workspace w;
  char* p = reinterpret_cast<char*>(ws);
  size_t off = 0;
  w.pmax = 2 * ((S + ITEM3 - 1) / ITEM3);             // partial slots per ray: enough for 16-sample tiles, 32-sample tiles and the training rows (two 16-row tiles per 32-sample tile)
  w.toff  = reinterpret_cast<int*>(p + off);       off += up256((size_t)(R + 1) * 4);
  w.ncomp = reinterpret_cast<int*>(p + off);       off += up256((size_t)R * 4);
  w.acc   = reinterpret_cast<float*>(p + off);     off += up256((size_t)R * 4);
  w.cidx  = reinterpret_cast<uint16_t*>(p + off);  off += up256((size_t)R * S * 2);
  w.cw    = reinterpret_cast<float*>(p + off);     off += up256((size_t)R * S * 4);
  w.part  = reinterpret_cast<float*>(p + off);     off += up256((size_t)R * w.pmax * 12);
  w.rdir  = reinterpret_cast<float*>(p + off);     off += up256((size_t)R * 16);
  w.perm  = reinterpret_cast<int*>(p + off);       off += up256((size_t)R * 4);
  w.rays_s = reinterpret_cast<float*>(p + off);    off += up256((size_t)R * 24);
  w.bytes = off;
  return w;
}

// Test hooks (include/lrf_debug.h): process-wide, not part of the re-entrant ABI.
static float* g_dump = nullptr;    // lrf_debug_set_dump: device buffer for the s_memtime totals of k_shade3<TIMED>
static int g_no_lds_lines = 0;     // lrf_debug_set_lds_lines(0): k_march reads its lines from global memory
static int g_pipe_chunk = 16384;    // lrf_debug_set_pipe_chunk: rays per chunk of lrf_render_fwd's large-batch mode (0: one pass over the whole batch, as rounds 1-5)
static int g_no_scene_fuse = 0;    // lrf_debug_set_scene_fuse(0): lrf_scene_fwd renders field by field (the tests compare the two forms)

static int device_cus() {                 // of the current device (one process may drive several)
  static int cache[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  int& cus = cache[dev & 63];
  if (!cus) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

static int g_bwd_overlap = 1;      // lrf_debug_set_bwd_overlap: weight-gradient GEMMs on a side stream, beside the scatter kernels
struct SideStream { hipStream_t s; hipEvent_t fork, join, app[2], bucket[5]; bool ok, bucket_set; std::mutex mu; };   // bucket[]: lrf_render_bwd_wait
static SideStream* side_stream() {
  static SideStream tab[64];
  static std::mutex init_mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  SideStream& x = tab[dev & 63];
  std::lock_guard<std::mutex> lk(init_mu);
  if (!x.ok) {
    if (hipStreamCreateWithFlags(&x.s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&x.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&x.join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&x.app[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&x.app[1], hipEventDisableTiming) != hipSuccess) return nullptr;
    for (int q = 0; q < 5; ++q)
      if (hipEventCreateWithFlags(&x.bucket[q], hipEventDisableTiming) != hipSuccess) return nullptr;
    x.bucket_set = false;
    x.ok = true;
  }
  return &x;
}
// LRF_FLAG_SORT_RAYS: sort the batch by direction (k_sort_rays) into the workspace; returns the rays the kernels should read
// and sets d.perm.  Batches beyond the 16-bit index of the sort words are rendered in the caller's order.
constexpr int LRF_SORT_MAX_R = 32768;
static const float* sort_rays_if_asked(DField& d, const float* rays, int R, uint32_t flags, const Workspace& w, hipStream_t st) {
  d.perm = nullptr;
  if (!(flags & LRF_FLAG_SORT_RAYS) || R > LRF_SORT_MAX_R || R < 2) return rays;
  int N = 2;
  while (N < R) N <<= 1;
  if ((size_t)N * 4 > 64 * 1024) {
    static std::once_flag once[64];
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess)
      std::call_once(once[dev & 63], [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rays), hipFuncAttributeMaxDynamicSharedMemorySize, LRF_SORT_MAX_R * 4);
      });
  }
  hipLaunchKernelGGL(k_sort_rays, dim3(1), dim3(1024), (size_t)N * 4, st, rays, R, N, w.rays_s, w.perm);
  d.perm = w.perm;
  return w.rays_s;
}

// rays per k_march workgroup with the density lines in LDS (4, 8 or 16), 0 when alpha slices + lines do not fit (grids of
// about 900^3 and above) or lrf_debug_set_lds_lines(0): k_march<false> then reads the lines from global memory
static int march_lds_rays(const int32_t ll[3], int S) {
  if (g_no_lds_lines) return 0;
  const size_t lds_l = (size_t)(ll[0] + ll[1] + ll[2]) * LRF_CD * sizeof(float);
  for (int cand = 4; cand <= 16; cand *= 2)
    if (((size_t)cand * S * sizeof(float) + lds_l) * (16 / cand) <= 156 * 1024) return cand;
  return 0;
}
// k_march launch: lines in LDS when the three of them (+ the alpha slices) leave four workgroups per CU.  mf (the fused
// multi-field form) exists with the lines in LDS only: lrf_scene_fwd does not fuse when march_lds_rays() is 0.
static void launch_march(const DField& d, const float* rays, const float* z, int R, int S, uint32_t flags, float floater,
                         float* depth, float* acc, float* w_all, int* ncomp, uint16_t* cidx, float* cw, float* feat,
                         hipStream_t st, const MultiF* mf = nullptr) {
  // 4 waves per SIMD either way (123 VGPRs): 4 / 2 / 1 workgroups of 4 / 8 / 16 rays per CU, whichever keeps
  // alpha slices + lines within the CU's 160 KB (300^3: 8 + 29 KB x 4; 500^3: 18 + 48 KB x 2; 640^3: 47 + 61 KB x 1)
  const size_t lds_l = (size_t)(d.ll[0] + d.ll[1] + d.ll[2]) * LRF_CD * sizeof(float);
  const int nw = march_lds_rays(d.ll, S);
  if (nw) {
    const size_t lds = (size_t)nw * S * sizeof(float) + lds_l;
    if (lds > 64 * 1024) {
      static std::once_flag attr_once[64];                  // per device; host threads may render concurrently
      int dev = 0;
      if (hipGetDevice(&dev) == hipSuccess)
        std::call_once(attr_once[dev & 63], [] {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_march<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
    }
    if (mf) {                                              // (the caller made sure that Rf is a multiple of 16 >= nw)
      static std::once_flag attr_once_m[64];
      int dev = 0;
      if (lds > 64 * 1024 && hipGetDevice(&dev) == hipSuccess)
        std::call_once(attr_once_m[dev & 63], [] {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_march<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
      hipLaunchKernelGGL((k_march<true, true>), dim3((R + nw - 1) / nw), dim3(64 * nw), lds, st,
                         *mf, rays, z, R, S, flags, floater, depth, acc, w_all, ncomp, cidx, cw, feat);
      return;
    }
    hipLaunchKernelGGL(k_march<true>, dim3((R + nw - 1) / nw), dim3(64 * nw), lds, st,
                       d, rays, z, R, S, flags, floater, depth, acc, w_all, ncomp, cidx, cw, feat);
  } else {
    hipLaunchKernelGGL(k_march<false>, dim3((R + 3) / 4), dim3(256), (size_t)4 * S * sizeof(float), st,
                       d, rays, z, R, S, flags, floater, depth, acc, w_all, ncomp, cidx, cw, feat);
  }
}

// k_march's output -> colours: k_shade3 behind an in-kernel scan of the tile offsets when they fit in LDS beside the image,
// behind k_scan_tiles_n otherwise (toff32: R + 1 ints).  sv == nullptr: the eval forward; else the training forward, which
// also leaves the rows of SaveOut3 (and sv->toff16 must not be toff32).  dump: lrf_debug_set_dump's buffer (eval only).
static hipError_t launch_shade3(DField d, const float* rays, const float* z, int R, int S, uint32_t flags, const Workspace& w,
                                int* toff32, float* rgb, float* acc_out, const SaveOut3* sv, float* dump, hipStream_t st) {
  const size_t lds_base = (size_t)W32_ALL_U4 * sizeof(uint4) + (size_t)S * sizeof(float);
  const size_t lds_toff = (size_t)(R + 1) * sizeof(int) + (size_t)R * sizeof(unsigned short) + 16;
  const bool in_lds = lds_base + lds_toff + 64 <= 160 * 1024 - 256;
  static std::once_flag attr3_once[64];                  // per device; the launch below must not overtake the opt-in on another host thread
  static hipError_t attr3_err[64];
  int dev = 0;
  hipError_t e0 = hipGetDevice(&dev);
  if (e0 != hipSuccess) return e0;
  std::call_once(attr3_once[dev & 63], [dev] {
    const int lim = 160 * 1024 - 256;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_shade3<8, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_shade3<8, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_shade3<8, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_shade3<8, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_shade3<8, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    attr3_err[dev & 63] = e;
  });
  if (attr3_err[dev & 63] != hipSuccess) return attr3_err[dev & 63];
  const dim3 grid(device_cus()), block(512);
  if (sv) {
    if (in_lds) {
      hipLaunchKernelGGL((k_shade3<8, true, false, true>), grid, block, lds_base + lds_toff, st,
                         d, rays, z, S, toff32, R, w.ncomp, w.cidx, w.cw, w.part, w.pmax, flags, w.acc, rgb, acc_out, *sv);
    } else {
      hipLaunchKernelGGL(k_scan_tiles_n<ITEM3>, dim3(1), dim3(1024), 0, st, w.ncomp, R, toff32);
      hipLaunchKernelGGL((k_shade3<8, false, false, true>), grid, block, lds_base, st,
                         d, rays, z, S, toff32, R, w.ncomp, w.cidx, w.cw, w.part, w.pmax, flags, w.acc, rgb, acc_out, *sv);
    }
  } else if (in_lds && dump) {                    // test hook: phase timing, s_memtime totals -> the dump buffer
    d.dump = dump;
    hipLaunchKernelGGL((k_shade3<8, true, true>), grid, block, lds_base + lds_toff, st,
                       d, rays, z, S, toff32, R, w.ncomp, w.cidx, w.cw, w.part, w.pmax, flags, w.acc, rgb, acc_out, SaveOut3{});
  } else if (in_lds) {
    hipLaunchKernelGGL((k_shade3<8, true, false>), grid, block, lds_base + lds_toff, st,
                       d, rays, z, S, toff32, R, w.ncomp, w.cidx, w.cw, w.part, w.pmax, flags, w.acc, rgb, acc_out, SaveOut3{});
  } else {
    hipLaunchKernelGGL(k_scan_tiles_n<ITEM3>, dim3(1), dim3(1024), 0, st, w.ncomp, R, toff32);
    hipLaunchKernelGGL((k_shade3<8, false, false>), grid, block, lds_base, st,
                       d, rays, z, S, toff32, R, w.ncomp, w.cidx, w.cw, w.part, w.pmax, flags, w.acc, rgb, acc_out, SaveOut3{});
  }
  return hipSuccess;
}

// Several fields, one launch (see MultiF): global tile offsets over all virtual rays (k_scan_tiles_n), then the colour kernel.
static hipError_t launch_shade3_multi(const MultiF& mf, const float* rays, const float* z, int Rv, int S, uint32_t flags,
                                      const Workspace& w, float* rgb, hipStream_t st) {
  const size_t lds_base = (size_t)W32_ALL_U4 * sizeof(uint4) + (size_t)S * sizeof(float);
  static std::once_flag attrm_once[64];
  static hipError_t attrm_err[64];
  int dev = 0;
  hipError_t e0 = hipGetDevice(&dev);
  if (e0 != hipSuccess) return e0;
  std::call_once(attrm_once[dev & 63], [dev] {
    attrm_err[dev & 63] = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_shade3m<8, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  });
  if (attrm_err[dev & 63] != hipSuccess) return attrm_err[dev & 63];
  hipLaunchKernelGGL(k_scan_tiles_n<ITEM3>, dim3(1), dim3(1024), 0, st, w.ncomp, Rv, w.toff);
  hipLaunchKernelGGL((k_shade3m<8, false>), dim3(device_cus()), dim3(512), lds_base, st,
                     mf, rays, z, S, w.toff, Rv, w.ncomp, w.cidx, w.cw, w.part, w.pmax, flags, w.acc, rgb, (float*)nullptr, SaveOut3{});
  return hipSuccess;
}

}  // namespace lrf

#include "lrf_generic.inl"
namespace lrf {
// network configuration of a field: null, or why it cannot be rendered
static const char* gen_check(const LrfField* f) {
  const int fc = f->feature_c ? f->feature_c : LRF_FEATC;
  if (f->fea_pe < 0 || f->fea_pe > GEN_MAX_PE || f->view_pe < 0 || f->view_pe > GEN_MAX_PE || fc < 1 || fc > GEN_MAX_FC)
    return "localrf: unsupported colour-network configuration (need 0 <= fea_pe, view_pe <= 6 and 1 <= featureC <= 256)";
  if (!gen_is_default(f->fea_pe, f->view_pe, fc) && (!f->basis || !f->w1 || !f->b1 || !f->w2 || !f->b2 || !f->w3 || !f->b3))
    return "localrf: a non-default colour-network configuration needs the natural-layout weights in LrfField (basis, w1 .. b3)";
  return nullptr;
}
// dynamic LDS above 64 KB has to be opted into once per device
static hipError_t gen_opt_in() {
  static std::once_flag once[64];
  static hipError_t err[64];
  int dev = 0;
  hipError_t e0 = hipGetDevice(&dev);
  if (e0 != hipSuccess) return e0;
  std::call_once(once[dev & 63], [dev] {
    const void* ks[4] = {reinterpret_cast<const void*>(&k_shade_gen<32, false>), reinterpret_cast<const void*>(&k_shade_gen<32, true>),
                         reinterpret_cast<const void*>(&k_gen_dgrad<32>), reinterpret_cast<const void*>(&k_gen_dgrad<16>)};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    err[dev & 63] = e;
  });
  return err[dev & 63];
}
// the generic colour kernel behind k_march; toff32 != null: the training forward (also leaves colours, feat rows, tile records)
static hipError_t launch_shade_gen(const DField& d, const GenCfg& gc, const float* rays, const float* z, int R, int S, const Workspace& w,
                                   const int* toff32, float* crgb, float* act, int4* tileinfo, hipStream_t st) {
  hipError_t e = gen_opt_in();
  if (e != hipSuccess) return e;
  constexpr int ls = 32;                                      // (the forward's LDS image is at most ~116 KB (fea_pe = view_pe = 6, feature_c = 256), under the 159 KB opt-in)
  const int nt = gen_block_threads(gc);
  const size_t lds = (size_t)gen_lds(gc, ls, false).total * 4;
  const dim3 grid(R * ((w.pmax * 16 + ls - 1) / ls));
  if (toff32) hipLaunchKernelGGL((k_shade_gen<32, true>), grid, dim3(nt), lds, st, d, gc, rays, z, S, w.ncomp, w.cidx, w.cw, w.part, w.pmax, toff32, crgb, act, tileinfo);
  else        hipLaunchKernelGGL((k_shade_gen<32, false>), grid, dim3(nt), lds, st, d, gc, rays, z, S, w.ncomp, w.cidx, w.cw, w.part, w.pmax, toff32, crgb, act, tileinfo);
  return hipGetLastError();
}
}  // namespace lrf
#include "lrf_backward.inl"
#include "lrf_scene.inl"
#include "lrf_adam.inl"
#include "lrf_losses.inl"
#include "lrf_reg.inl"
#include "lrf_mask.inl"

using namespace lrf;

extern "C" {

int lrf_abi_version(void) { return LRF_ABI_VERSION; }
void lrf_debug_set_dump(float* buf) { g_dump = buf; }
void lrf_debug_set_bwd_overlap(int on) { g_bwd_overlap = (on & 1) ? 1 : 0; if (on > 1) g_wgrad_split = (on >> 1) - 1; }   // on = 1 + 2 * (n + 1): k_wgrad_w2w3 on the caller's stream (n > 0) or on the side stream (n = 0)
void lrf_debug_set_lds_lines(int on) { g_no_lds_lines = on ? 0 : 1; }
void lrf_debug_set_pipe_chunk(int rays) { g_pipe_chunk = rays > 0 ? rays : 0; }
#ifdef LRF_SCATTER_PROF
int lrf_debug_scatter_prof(unsigned long long* host_out /* [2][2048][12] */) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lrf::g_scat_prof), sizeof(unsigned long long) * 2 * 2048 * 12);
}
#endif
void lrf_debug_set_scene_fuse(int on) { g_no_scene_fuse = on ? 0 : 1; }
// Where column `col` of saved row `row` lives, in floats from the start of the ACT (buffer 0) / GRD (buffer 1) region
// of a training workspace (lrf_workspace_layout_bwd gives the regions): the fragment order of lrf_common.h, host side.
// -1 for bad arguments.
int64_t lrf_debug_saved_row_offset(int buffer, uint64_t row, int col) {
  using namespace lrf;
  if (buffer == 0) return (col < 0 || col >= ACT_LD) ? -1 : (int64_t)frag_off((size_t)row, col, ACT_LD);
  if (buffer == 1) {
    if (col < 0 || col >= GRD_LD) return -1;
    if (col >= GRD_DX) return (int64_t)((row >> 4) * (uint64_t)(16 * GRD_LD) + GRD_DX * 16 + (row & 15) * (GRD_LD - GRD_DX) + (col - GRD_DX));   // (the row-major order of the dX block; the other one: lrf_common.h)
    return (int64_t)frag_off((size_t)row, col, GRD_LD);
  }
  return -1;
}
const char* lrf_last_error(void) { return lrf_error_slot(); }
char* lrf_error_slot(void) { return g_err; }

size_t lrf_cache_bytes(const int32_t grid[3]) { return make_layout(grid).total * sizeof(float); }

int lrf_pack_field(const LrfParams* p, void* cache, void* stream) {
  if (!p || !cache) return set_err("lrf_pack_field: null argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const Layout L = make_layout(p->grid);
  float* base = reinterpret_cast<float*>(cache);
  PackTab tab;
  int wmax = 1, hmax = 1;
  for (int q = 0; q < 3; ++q) {
    tab.s[4 * q + 0] = PackSeg{p->density_plane[q], base + L.dplane[q], LRF_CD, L.ph[q], L.pw[q], LRF_CD, 0};
    tab.s[4 * q + 1] = PackSeg{p->app_plane[q], base + L.aplane[q], LRF_CA, L.ph[q], L.pw[q], LRF_CAS, 1};
    tab.s[4 * q + 2] = PackSeg{p->density_line[q], base + L.dline[q], LRF_CD, 1, L.ll[q], LRF_CD, 0};
    tab.s[4 * q + 3] = PackSeg{p->app_line[q], base + L.aline[q], LRF_CA, 1, L.ll[q], LRF_CAS, 1};
    tab.s[12 + 2 * q + 0] = PackSeg{p->app_plane[q], base + L.aplane2[q], LRF_CA, L.ph[q], L.pw[q], LRF_CA, 0};
    tab.s[12 + 2 * q + 1] = PackSeg{p->app_line[q], base + L.aline2[q], LRF_CA, 1, L.ll[q], LRF_CA, 0};
    wmax = max(wmax, max(L.pw[q], L.ll[q]));
    hmax = max(hmax, L.ph[q]);
  }
  static_assert(IMG_FLOATS <= 26880 && IMGB_ALL * 4 <= 26880 && W32_ALL_U4 * 4 <= 26880 && W32T_ALL_U4 * 4 <= 26880, "slice 18 of k_pack_planes covers every network image");
  PackMlp pm;
  pm.mlp = base + L.mlp; pm.mlpb = reinterpret_cast<uint32_t*>(base + L.mlpb); pm.mlpw = reinterpret_cast<uint32_t*>(base + L.mlpw);
  pm.mlpwt = reinterpret_cast<uint32_t*>(base + L.mlpwt);
  pm.mode = gen_is_default(p->fea_pe, p->view_pe, p->feature_c ? p->feature_c : LRF_FEATC) ? 0 : 1;   // the generic engine reads the parameter tensors themselves
  hipLaunchKernelGGL(k_pack_planes, dim3((wmax + 127) / 128, hmax, 19), dim3(128), 0, st, tab, *p, pm, 0);
  LRF_HIP(hipGetLastError());
  return 0;
}

// Large batches (full-frame evaluation: renderer.py:65-77 renders an image 4096 rays at a time) go through the two kernels
// in chunks of g_pipe_chunk rays that alternate over the caller's stream and the side stream: a chunk's k_march and the
// tail of its k_shade3 run beside the other stream's colour kernel (k_shade3 is one persistent workgroup per CU bound by
// instruction issue at 46 %; k_march is latency / L2 bound).  65536 rays at configs[1]: 2.38 ms in one pass, 2.49 / 2.30 /
// 2.24 / 2.26 / 2.30 ms in chunks of 4096 / 8192 / 16384 / 24576 / 32768 (scripts/pipe_chunk_probe.py; 262144 rays: 9.55 ->
// 8.69 ms = 30.2 M rays/s); below two chunks of 16384 one pass is as fast or faster.  Every chunk has a workspace of its own
// inside the caller's.
static int pipe_chunks(int R) { return (g_pipe_chunk > 0 && R >= 2 * g_pipe_chunk) ? (R + g_pipe_chunk - 1) / g_pipe_chunk : 1; }
size_t lrf_workspace_bytes(int32_t R, int32_t S) {
  const size_t whole = carve(nullptr, R, S).bytes;
  const int nc = pipe_chunks(R);
  return nc > 1 ? std::max(whole, (size_t)nc * carve(nullptr, g_pipe_chunk, S).bytes) : whole;
}

// One batch of rays through k_march and the colour stage on `st`; ev (optional, lrf_render_fwd_profile): 4 events =
// start, after k_march, after the colour kernel(s), end.
static int render_fwd_impl(const LrfField* f, const float* rays, const float* z, int32_t R, int32_t S,
                           uint32_t flags, float floater_thresh, float* rgb, float* depth,
                           float* weight_out, float* acc_out, void* workspace, hipStream_t st, hipEvent_t* ev) {
  if (!f || !f->cache || !rays || !z || !rgb || !depth || !workspace) return set_err("lrf_render_fwd: null argument");
  if (R <= 0 || S < 2 || S > 4096) return set_err("lrf_render_fwd: need R > 0 and 2 <= S <= 4096");
  if (flags & ~(LRF_FLAG_ALL & ~(LRF_FLAG_ROWS_SAVED | LRF_FLAG_PLANE_EVENTS))) return set_err("lrf_render_fwd: unknown flag bits (caller built against another ABI version?)");
  DField d = make_dfield(f);
  const Workspace w = carve(workspace, R, S);
  if (ev) LRF_HIP(hipEventRecord(ev[0], st));
  rays = sort_rays_if_asked(d, rays, R, flags, w, st);
  if (const char* bad = gen_check(f)) return set_err(bad);
  const bool generic = !gen_is_default(d.fea_pe, d.view_pe, d.fc);
  if (generic && (flags & LRF_FLAG_MLP_F32)) return set_err("lrf_render_fwd: the exact-fp32 MFMA engine is built for fea_pe = view_pe = 0, featureC = 128 only");
  if (!generic && !(flags & (LRF_FLAG_MLP_VALU | LRF_FLAG_MLP_F32))) {
    // Default engine: k_march -> k_shade3 (32 samples per wave on v_mfma_f32_32x32x16_bf16, lrf_shade3.inl); the tile
    // offsets are scanned inside the colour kernel when they fit in LDS beside the image, by k_scan_tiles_n otherwise.
    d.rdir = w.rdir;
    launch_march(d, rays, z, R, S, flags, floater_thresh, depth, w.acc, weight_out, w.ncomp, w.cidx, w.cw, nullptr, st);
    if (ev) LRF_HIP(hipEventRecord(ev[1], st));
    LRF_HIP(launch_shade3(d, rays, z, R, S, flags, w, w.toff, rgb, acc_out, nullptr, g_dump, st));
    if (ev) { LRF_HIP(hipEventRecord(ev[2], st)); LRF_HIP(hipEventRecord(ev[3], st)); }
    LRF_HIP(hipGetLastError());
    return 2;            // (internal) done, two launches: no finalize interval
  }
  // exact-fp32 / plain-loop engines: 16-sample tiles, k_march -> [k_scan_tiles ->] colour kernel -> k_finalize
  launch_march(d, rays, z, R, S, flags, floater_thresh, depth, w.acc, weight_out, w.ncomp, w.cidx, w.cw, nullptr, st);
  if (ev) LRF_HIP(hipEventRecord(ev[1], st));
  if (generic || (flags & LRF_FLAG_MLP_VALU)) {
    const GenCfg gc = gen_cfg(d.fea_pe, d.view_pe, d.fc, !(flags & LRF_FLAG_PE_OFF));
    LRF_HIP(launch_shade_gen(d, gc, rays, z, R, S, w, nullptr, nullptr, nullptr, nullptr, st));
  } else {
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st, w.ncomp, R, w.toff);
    hipLaunchKernelGGL(k_shade, dim3(device_cus()), dim3(1024), 0, st, d, rays, z, S, w.toff, R, w.ncomp, w.cidx, w.cw, w.part, w.pmax);
  }
  if (ev) LRF_HIP(hipEventRecord(ev[2], st));
  hipLaunchKernelGGL(k_finalize, dim3((R + 255) / 256), dim3(256), 0, st, R, w.pmax, flags, w.ncomp, w.acc, w.part, rgb, acc_out, d.perm);
  if (ev) LRF_HIP(hipEventRecord(ev[3], st));
  LRF_HIP(hipGetLastError());
  return 0;
}

// render_fwd_impl, large batches in chunks over two streams (see lrf_workspace_bytes); 0 or an error code
static int render_fwd_pipelined(const LrfField* f, const float* rays, const float* z, int32_t R, int32_t S,
                                uint32_t flags, float floater_thresh, float* rgb, float* depth,
                                float* weight_out, float* acc_out, void* workspace, hipStream_t st) {
  const int nc = (f && rays && rgb && depth && workspace && R > 0) ? pipe_chunks(R) : 1;
  SideStream* ss = nc > 1 ? side_stream() : nullptr;
  if (!ss) {
    const int rc = render_fwd_impl(f, rays, z, R, S, flags, floater_thresh, rgb, depth, weight_out, acc_out, workspace, st, nullptr);
    return rc == 2 ? 0 : rc;
  }
  std::lock_guard<std::mutex> lk(ss->mu);
  LRF_HIP(hipEventRecord(ss->fork, st));
  LRF_HIP(hipStreamWaitEvent(ss->s, ss->fork, 0));
  const size_t wb = carve(nullptr, g_pipe_chunk, S).bytes;
  int rc = 0;
  for (int c = 0; c < nc && (rc == 0 || rc == 2); ++c) {
    const int r0 = c * g_pipe_chunk, rn = std::min(g_pipe_chunk, R - r0);
    rc = render_fwd_impl(f, rays + (size_t)r0 * 6, z, rn, S, flags, floater_thresh, rgb + (size_t)r0 * 3, depth + r0,
                         weight_out ? weight_out + (size_t)r0 * S : nullptr, acc_out ? acc_out + r0 : nullptr,
                         static_cast<char*>(workspace) + (size_t)c * wb, (c & 1) ? ss->s : st, nullptr);
  }
  LRF_HIP(hipEventRecord(ss->join, ss->s));                  // (also after an error: the side stream is joined again)
  LRF_HIP(hipStreamWaitEvent(st, ss->join, 0));
  return rc == 2 ? 0 : rc;
}

int lrf_render_fwd(const LrfField* f, const float* rays, const float* z, int32_t R, int32_t S,
                   uint32_t flags, float floater_thresh, float* rgb, float* depth,
                   float* weight_out, float* acc_out, void* workspace, void* stream) {
  return render_fwd_pipelined(f, rays, z, R, S, flags, floater_thresh, rgb, depth, weight_out, acc_out, workspace,
                              reinterpret_cast<hipStream_t>(stream));
}

// LocalTensorfs.forward without a tape (local_tensorfs.py:397-499) as ONE call: the rays of every active field, the
// per-field renders chunk by chunk in the reference's order (:440-474: for each chunk, for each field), the blend.  Same
// launches as lrf_scene_rays + n_rf x lrf_render_fwd + lrf_scene_blend, enqueued from C: the host side of a 4-field scene
// forward drops from 0.36 ms of Python per call to one ctypes call.
int lrf_scene_fwd(const int64_t* ray_ids, int32_t R, int32_t per_view, const float* cam2world, const float* world2rf,
                  int32_t n_rf, const float* focal, const float* center, int32_t W, int32_t H, int32_t fov360,
                  const LrfSceneField* fields, float floater_thresh, int32_t chunk,
                  const float* blend_w, const float* exposure,
                  float* rays, float* rgb_f, float* depth_f, float* directions, int64_t* ij,
                  float* rgbs, float* depth, void* scene_workspace, size_t scene_workspace_bytes, void* stream) {
  if (!fields || !rays || !rgb_f || !depth_f || !rgbs || !depth) return set_err("lrf_scene_fwd: null argument");
  if (n_rf <= 0 || n_rf > LRF_SCENE_MAX_FIELDS) return set_err("lrf_scene_fwd: 1 <= n_rf <= LRF_SCENE_MAX_FIELDS");
  if (R < 0 || per_view <= 0 || R % per_view) return set_err("lrf_scene_fwd: R must be a multiple of per_view");
  if (chunk <= 0) chunk = R;
  for (int k = 0; k < n_rf; ++k)
    if (!fields[k].field || !fields[k].z || !fields[k].workspace) return set_err("lrf_scene_fwd: null field / z / workspace");
  int rc = lrf_scene_rays(ray_ids, R, per_view, cam2world, world2rf, n_rf, focal, center, W, H, fov360, rays, directions, ij, stream);
  if (rc) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // Fused form: groups of up to LRF_MULTI_MAX fields in ONE march and ONE colour launch over their field-major "virtual" rays
  // (3 launches per group and chunk instead of 2 per field; one prologue, one tail).  Needs: chunks of a multiple of 16 rays
  // (a march workgroup stays inside one field; a ragged last chunk goes field by field), the default engine, and fields of one shape: same grid, sample count,
  // flags, thresholds (their schedules z are then the same numbers).  `scene_workspace` (lrf_workspace_bytes(group rays, S))
  // holds the group's per-ray state.  Same arithmetic per ray (depths bit-identical, colours to an ulp: a second compilation of the colour kernel).
  bool fuse = scene_workspace && n_rf >= 2 && R > 0 && chunk % 16 == 0 && !g_no_scene_fuse;
  for (int k = 0; fuse && k < n_rf; ++k) {
    const LrfSceneField& a = fields[k], &b = fields[0];
    if (!a.field->cache || gen_check(a.field) || !gen_is_default(a.field->fea_pe, a.field->view_pe, a.field->feature_c ? a.field->feature_c : LRF_FEATC)) fuse = false;
    else if (a.flags & (LRF_FLAG_MLP_VALU | LRF_FLAG_MLP_F32 | LRF_FLAG_SORT_RAYS)) fuse = false;
    else if (a.S != b.S || a.flags != b.flags || memcmp(a.field->grid, b.field->grid, sizeof(b.field->grid))) fuse = false;
    else if (a.field->weight_thres != b.field->weight_thres || a.field->term_T != b.field->term_T) fuse = false;
  }
  if (fuse && floater_thresh > 0.0f) fuse = false;             // (the floater filter's second pass: per-field path)
  if (fuse) {                                                  // k_march<MULTI> keeps the density lines in LDS: no such form without them
    const int32_t* g = fields[0].field->grid;                  // (line p runs along axis VEC[p]: the three lengths are the three grid sizes)
    const int32_t ll[3] = {g[0], g[1], g[2]};
    if (!march_lds_rays(ll, fields[0].S)) fuse = false;
  }
  if (fuse) {
    const int32_t S = fields[0].S;
    if (S < 2 || S > 4096) return set_err("lrf_scene_fwd: need 2 <= S <= 4096");
    for (int32_t lo = 0; lo < R; lo += chunk) {                // chunk by chunk, as the field-by-field form (:440)
      const int32_t n = R - lo < chunk ? R - lo : chunk;
      if (n % 16) {                                            // a ragged last chunk: field by field
        for (int k = 0; k < n_rf; ++k) {
          const LrfSceneField& sf = fields[k];
          rc = render_fwd_pipelined(sf.field, rays + ((size_t)k * R + lo) * 6, sf.z, n, sf.S, sf.flags, floater_thresh,
                                    rgb_f + ((size_t)k * R + lo) * 3, depth_f + (size_t)k * R + lo, nullptr, nullptr, sf.workspace, st);
          if (rc) return rc;
        }
        continue;
      }
      for (int k0 = 0; k0 < n_rf; k0 += LRF_MULTI_MAX) {
        const int nf = n_rf - k0 < LRF_MULTI_MAX ? n_rf - k0 : LRF_MULTI_MAX;
        const int Rv = nf * n;
        if (lrf_workspace_bytes(Rv, S) > scene_workspace_bytes) return set_err("lrf_scene_fwd: scene workspace too small");
        const Workspace w = carve(scene_workspace, Rv, S);
        MultiF mf;
        mf.nf = nf; mf.Rf = n; mf.Rs = R; mf.lo = lo;
        for (int k = 0; k < nf; ++k) { mf.f[k] = make_dfield(fields[k0 + k].field); mf.f[k].rdir = w.rdir; }
        for (int k = nf; k < LRF_MULTI_MAX; ++k) mf.f[k] = mf.f[0];
        const float* rv = rays + (size_t)k0 * R * 6;           // field k of the group: rays [k0 + k][lo + r], outputs likewise (multi_io)
        launch_march(mf.f[0], rv, fields[k0].z, Rv, S, fields[k0].flags, 0.0f, depth_f + (size_t)k0 * R, w.acc, nullptr,
                     w.ncomp, w.cidx, w.cw, nullptr, st, &mf);
        LRF_HIP(launch_shade3_multi(mf, rv, fields[k0].z, Rv, S, fields[k0].flags, w, rgb_f + (size_t)k0 * R * 3, st));
      }
    }
    LRF_HIP(hipGetLastError());
    return lrf_scene_blend(rgb_f, depth_f, blend_w, exposure, R, per_view, n_rf, rgbs, depth, nullptr, stream);
  }
  for (int32_t lo = 0; lo < R; lo += chunk) {
    const int32_t n = R - lo < chunk ? R - lo : chunk;
    for (int k = 0; k < n_rf; ++k) {
      const LrfSceneField& sf = fields[k];
      rc = render_fwd_pipelined(sf.field, rays + ((size_t)k * R + lo) * 6, sf.z, n, sf.S, sf.flags, floater_thresh,   // (chunks of 32768 rays and more: over two streams)
                                rgb_f + ((size_t)k * R + lo) * 3, depth_f + (size_t)k * R + lo, nullptr, nullptr, sf.workspace, st);
      if (rc) return rc;
    }
  }
  return lrf_scene_blend(rgb_f, depth_f, blend_w, exposure, R, per_view, n_rf, rgbs, depth, nullptr, stream);
}

int lrf_render_fwd_profile(const LrfField* f, const float* rays, const float* z, int32_t R, int32_t S,
                           uint32_t flags, float floater_thresh, float* rgb, float* depth,
                           void* workspace, void* stream, float* ms_out, int32_t* n_shaded_out) {
  if (!ms_out) return set_err("lrf_render_fwd_profile: null ms_out");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipEvent_t ev[4];
  for (int i = 0; i < 4; ++i) LRF_HIP(hipEventCreate(&ev[i]));
  int rc = render_fwd_impl(f, rays, z, R, S, flags, floater_thresh, rgb, depth, nullptr, nullptr, workspace, st, ev);
  const bool two_launches = rc == 2;
  if (two_launches) rc = 0;
  if (rc == 0) {
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) rc = set_err("hipStreamSynchronize", e);
  }
  if (rc == 0) {
    for (int i = 0; i < 3; ++i) (void)hipEventElapsedTime(&ms_out[i], ev[i], ev[i + 1]);      // march, colour stage, finalize (0 for the default engine)
    (void)hipEventElapsedTime(&ms_out[3], ev[0], ev[3]);
    if (two_launches) { ms_out[2] = 0.0f; (void)hipEventElapsedTime(&ms_out[3], ev[0], ev[2]); }
    ms_out[4] = ms_out[5] = 0.0f;
    if (n_shaded_out) {
      // shaded-sample count of this batch = sum of ncomp (host copy; measurement only)
      const Workspace w = carve(workspace, R, S);
      int* h = (int*)malloc((size_t)R * sizeof(int));
      if (h && hipMemcpy(h, w.ncomp, (size_t)R * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess) {
        long long tot = 0; for (int i = 0; i < R; ++i) tot += h[i];
        *n_shaded_out = (int32_t)tot;
      }
      free(h);
    }
  }
  for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ev[i]);
  return rc;
}

int lrf_density_feature(const LrfField* f, const float* u, int32_t P, float* out, void* stream) {
  if (!f || !f->cache || !u || !out) return set_err("lrf_density_feature: null argument");
  if (P <= 0) return 0;
  hipLaunchKernelGGL(k_density_feature, dim3((P + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     make_dfield(f), u, P, out);
  LRF_HIP(hipGetLastError());
  return 0;
}

int lrf_app_feature(const LrfField* f, const float* u, int32_t P, float* out, void* stream) {
  if (!f || !f->cache || !u || !out || !f->basis) return set_err("lrf_app_feature: null argument");
  if (P <= 0) return 0;
  hipLaunchKernelGGL(k_app_feature, dim3((P + 127) / 128), dim3(128), 0, reinterpret_cast<hipStream_t>(stream),
                     make_dfield(f), u, P, out);
  LRF_HIP(hipGetLastError());
  return 0;
}

int lrf_sample_ray_aabb(const float* rays, const float aabb[6], float step_size, float near_, float far_,
                        const float* jitter, int32_t R, int32_t N, float* pts, float* t, uint8_t* inside,
                        void* stream) {
  if (!rays || !aabb || !pts || !t || !inside) return set_err("lrf_sample_ray_aabb: null argument");
  const size_t n = (size_t)R * N;
  if (!n) return 0;
  hipLaunchKernelGGL(k_sample_ray_aabb, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     rays, aabb[0], aabb[1], aabb[2], aabb[3], aabb[4], aabb[5], step_size, near_, far_,
                     jitter, R, N, pts, t, inside);
  LRF_HIP(hipGetLastError());
  return 0;
}

int lrf_z_schedule(int32_t h, const float* u1, const float* u2, float* z, void* stream) {
  if (h <= 0 || !z) return set_err("lrf_z_schedule: need h > 0 and an output of 2 h floats");
  if ((u1 == nullptr) != (u2 == nullptr)) return set_err("lrf_z_schedule: both jitters or none");
  hipLaunchKernelGGL(k_z_schedule, dim3((unsigned)((h + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), h, u1, u2, z);
  LRF_HIP(hipGetLastError());
  return 0;
}

int lrf_sample_ray_contracted(const float* rays_o, const float* rays_d, const float* z, int32_t R, int32_t S,
                              float* pts, void* stream) {
  if (!rays_o || !rays_d || !z || !pts) return set_err("lrf_sample_ray_contracted: null argument");
  const size_t n = (size_t)R * S;
  if (!n) return 0;
  hipLaunchKernelGGL(k_sample_contracted, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     rays_o, rays_d, z, R, S, pts);
  LRF_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
