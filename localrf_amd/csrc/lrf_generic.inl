// lrf_generic.inl -- the colour network for ANY MLPRender_Fea_late_view configuration (tensorBase.py:97-135): positional
// encodings of the appearance features (fea_pe) and of the view direction (view_pe), any hidden width featureC <= 256.
// The fast kernels (k_shade3, k_train_dgrad3, k_wgrad_w2w3) are specialised to opt.py's defaults (0 / 0 / 128), which is what
// train.py runs; every other configuration takes this engine: plain fp32 loops on the vector ALU over the parameter
// tensors in their natural layout.  A block owns 32 samples of one ray, keeps their activation vectors in LDS as
// [element][sample] and gives every OUTPUT unit of a layer to a thread (weight read once, 32 broadcast multiply-adds): a plain
// tile GEMM -- on the matrix pipe since round 5 (v_mfma_f32_16x16x4_f32, exact fp32: gen_tile_dense_mfma below; the vector-ALU form
// stays as gen_tile_dense_valu behind LRF_GEN_MFMA=0).  (First version: one lane per sample with private arrays -- 3-7 KB of scratch
// per lane, gigabytes per launch, every multiply-add waiting for HBM: 155 ms per 4096 x 512 batch.)  Correct and differentiable;
// 4.1 ms forward / 18.8 ms forward + backward per batch at view_pe = fea_pe = 2 (round 4: 6.5 / 36): the tile products are no longer
// what it waits for -- a block per 32 samples re-stages everything per tile and synchronises ten times (profiles/r15_generic_engine.md).
// It is also the LRF_FLAG_MLP_VALU debug engine of the default configuration.
//
// Forward (k_shade_gen): a block renders 32 (16) consecutive samples of one ray; with SAVE it leaves what the backward needs in the
// same places as k_shade3<SAVE> (colours, feat rows, tile records of the 16-row tiles) -- no mask bits: the backward
// recomputes the network from the saved feat row with the same arithmetic, so its ReLU signs are the forward's.
// Backward: k_gen_dgrad, a block per 32 (16) saved rows: recompute, d(loss)/d(pre-sigmoid) -> dz2 -> dz1 -> d(input) -> dfeat (through
// the encodings), written as the gradient row's dfeat block -- from there k_train_app3 and the scatter kernels run unchanged --
// and the operands of the weight gradients as rows [dz1 | dz2 | x, 1 | relu(h1), 1 | relu(h2), venc, 1 | go]; k_gen_gemm forms
// dW = A^T B over those rows (three launches) and adds into the reference-layout gradients.
#pragma once

namespace lrf {

constexpr int GEN_MAX_PE = 6, GEN_MAX_FC = 256;
constexpr int GEN_MAX_IN1 = LRF_APP_DIM * (1 + 2 * GEN_MAX_PE), GEN_MAX_INV = 3 * (1 + 2 * GEN_MAX_PE);

struct GenCfg { int fea_pe, view_pe, fc, in1, inv, pe_on; };
__host__ __device__ inline GenCfg gen_cfg(int fea_pe, int view_pe, int fc, bool pe_on) {
  GenCfg g;
  g.fea_pe = fea_pe; g.view_pe = view_pe; g.fc = fc;
  g.in1 = LRF_APP_DIM * (1 + 2 * fea_pe);           // tensorBase.py:101
  g.inv = 3 * (1 + 2 * view_pe);                    // :102
  g.pe_on = pe_on ? 1 : 0;                          // refine == False feeds zeros in place of the feature encodings (:118-126)
  return g;
}
// floats per saved row of the weight-gradient operands
__host__ __device__ inline int gen_row_ld(const GenCfg& g) { return 2 * g.fc + (g.in1 + 1) + (g.fc + 1) + (g.fc + g.inv + 1) + 4; }
struct GenRowOff { int dz1, dz2, x1, h1, h2v, go; };
__host__ __device__ inline GenRowOff gen_row_off(const GenCfg& g) {
  GenRowOff o;
  o.dz1 = 0; o.dz2 = g.fc; o.x1 = 2 * g.fc; o.h1 = o.x1 + g.in1 + 1; o.h2v = o.h1 + g.fc + 1; o.go = o.h2v + g.fc + g.inv + 1;
  return o;
}
__host__ __device__ inline bool gen_is_default(int fea_pe, int view_pe, int fc) { return fea_pe == 0 && view_pe == 0 && fc == LRF_FEATC; }

// ---- block-cooperative pieces.  A block owns LS samples (32, or 16 where the LDS budget asks for it); every vector of a
// sample lives in LDS as [element][sample], so a thread that owns an OUTPUT unit reads its weight once and multiplies it
// with LS broadcast activations: 4 LS multiply-adds per (weight float4, four ds_read_b128 per sample quad).

// out[i][s] = (relu) b[i] + sum_c W[i * nin + c] in[c][s]
template <int LS>
__device__ __forceinline__ void gen_tile_dense_valu(const float* __restrict__ W, const float* __restrict__ b, int nout, int nin,
                                               const float* in, float* out, bool relu) {
  for (int i = threadIdx.x; i < nout; i += blockDim.x) {
    float acc[LS];
    const float bi = b ? b[i] : 0.0f;
#pragma unroll
    for (int s = 0; s < LS; ++s) acc[s] = bi;
    const float* wr = W + (size_t)i * nin;
    int c = 0;
    for (; c + 4 <= nin; c += 4) {
      const float w0 = wr[c], w1 = wr[c + 1], w2 = wr[c + 2], w3 = wr[c + 3];
#pragma unroll
      for (int q = 0; q < LS / 4; ++q) {
        const float4 v0 = *reinterpret_cast<const float4*>(in + (c + 0) * LS + 4 * q), v1 = *reinterpret_cast<const float4*>(in + (c + 1) * LS + 4 * q);
        const float4 v2 = *reinterpret_cast<const float4*>(in + (c + 2) * LS + 4 * q), v3 = *reinterpret_cast<const float4*>(in + (c + 3) * LS + 4 * q);
        acc[4 * q]     += w0 * v0.x; acc[4 * q + 1] += w0 * v0.y; acc[4 * q + 2] += w0 * v0.z; acc[4 * q + 3] += w0 * v0.w;
        acc[4 * q]     += w1 * v1.x; acc[4 * q + 1] += w1 * v1.y; acc[4 * q + 2] += w1 * v1.z; acc[4 * q + 3] += w1 * v1.w;
        acc[4 * q]     += w2 * v2.x; acc[4 * q + 1] += w2 * v2.y; acc[4 * q + 2] += w2 * v2.z; acc[4 * q + 3] += w2 * v2.w;
        acc[4 * q]     += w3 * v3.x; acc[4 * q + 1] += w3 * v3.y; acc[4 * q + 2] += w3 * v3.z; acc[4 * q + 3] += w3 * v3.w;
      }
    }
    for (; c < nin; ++c) {
      const float w0 = wr[c];
#pragma unroll
      for (int q = 0; q < LS / 4; ++q) {
        const float4 v0 = *reinterpret_cast<const float4*>(in + c * LS + 4 * q);
        acc[4 * q] += w0 * v0.x; acc[4 * q + 1] += w0 * v0.y; acc[4 * q + 2] += w0 * v0.z; acc[4 * q + 3] += w0 * v0.w;
      }
    }
#pragma unroll
    for (int q = 0; q < LS / 4; ++q) {
      float4 r = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
      if (relu) { r.x = fmaxf(r.x, 0.0f); r.y = fmaxf(r.y, 0.0f); r.z = fmaxf(r.z, 0.0f); r.w = fmaxf(r.w, 0.0f); }
      *reinterpret_cast<float4*>(out + i * LS + 4 * q) = r;
    }
  }
}
// out[c][s] = sum_r W[r * ld + c] in[r][s]  (the transposed products of the backward; consecutive threads read consecutive weights)
template <int LS>
__device__ __forceinline__ void gen_tile_dense_t_valu(const float* __restrict__ W, int nrow, int ncol, int ld, const float* in, float* out) {
  for (int c = threadIdx.x; c < ncol; c += blockDim.x) {
    float acc[LS];
#pragma unroll
    for (int s = 0; s < LS; ++s) acc[s] = 0.0f;
    for (int r = 0; r < nrow; ++r) {
      const float w0 = W[(size_t)r * ld + c];
#pragma unroll
      for (int q = 0; q < LS / 4; ++q) {
        const float4 v0 = *reinterpret_cast<const float4*>(in + r * LS + 4 * q);
        acc[4 * q] += w0 * v0.x; acc[4 * q + 1] += w0 * v0.y; acc[4 * q + 2] += w0 * v0.z; acc[4 * q + 3] += w0 * v0.w;
      }
    }
#pragma unroll
    for (int q = 0; q < LS / 4; ++q)
      *reinterpret_cast<float4*>(out + c * LS + 4 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
  }
}

// ---- the same two tile products on the matrix pipe (round 5): v_mfma_f32_16x16x4_f32, exact fp32 products and accumulation.
// A wave owns 16 output units x LS samples (LS / 16 accumulator tiles) and walks K four inputs at a time: lane (i = lane & 15,
// g = lane >> 4) supplies A[i][g] = the weight of unit i and input k0 + g (one dword load, 16 consecutive inputs of a row /
// 16 consecutive units of a transposed row per four lanes) and B[g][i] = activation k0 + g of sample i from the block's
// [element][sample] LDS image (conflict-free ds_read_b32); D[4 g + j][i] comes back as four registers per tile.  The vector-ALU
// form above read every weight once per unit and 32-sample block and issued one FMA per product and lane (155 -> 6.5 ms
// per 4096 x 512 batch at view_pe = fea_pe = 2 after two rounds of tuning); this one issues a 1024-product instruction per
// 16 x 16 x 4 block.  Inputs beyond nin / rows beyond nrow enter as zeros (never as whatever the LDS holds there).
#ifndef LRF_GEN_MFMA
#define LRF_GEN_MFMA 1
#endif
#ifndef LRF_GEN_KU
#define LRF_GEN_KU 8
#endif
constexpr int GEN_KU = LRF_GEN_KU;
template <int LS>
__device__ __forceinline__ void gen_tile_dense_mfma(const float* __restrict__ W, const float* __restrict__ b, int nout, int nin,
                                                    const float* in, float* out, bool relu) {
  constexpr int NTL = LS / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  for (int mt = wave; mt * 16 < nout; mt += nwave) {
    const int u = mt * 16 + i;
    const bool uok = u < nout;
    const float* wr = W + (size_t)(uok ? u : 0) * nin;
    f32x4 acc[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int k0 = 0; k0 < nin; k0 += 4 * GEN_KU) {            // GEN_KU K-steps per iteration: their weight loads are in flight together
      float a[GEN_KU], bv[GEN_KU][NTL];
#pragma unroll
      for (int q = 0; q < GEN_KU; ++q) {
        const int k = k0 + 4 * q + g;
        const bool kok = k < nin;
        const int kc = kok ? k : nin - 1;                          // branch-free: every load is issued (clamped), zeros are selected afterwards
        const float av = wr[kc];
        a[q] = (uok && kok) ? av : 0.0f;
#pragma unroll
        for (int t = 0; t < NTL; ++t) { const float xv = in[kc * LS + 16 * t + i]; bv[q][t] = kok ? xv : 0.0f; }
      }
#pragma unroll
      for (int q = 0; q < GEN_KU; ++q)
#pragma unroll
        for (int t = 0; t < NTL; ++t) acc[t] = mfma4(a[q], bv[q][t], acc[t]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int unit = mt * 16 + 4 * g + j;
      if (unit < nout) {
        const float bi = b ? b[unit] : 0.0f;
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
          float v = acc[t][j] + bi;
          if (relu) v = fmaxf(v, 0.0f);
          out[unit * LS + 16 * t + i] = v;
        }
      }
    }
  }
}
template <int LS>
__device__ __forceinline__ void gen_tile_dense_t_mfma(const float* __restrict__ W, int nrow, int ncol, int ld, const float* in, float* out) {
  constexpr int NTL = LS / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  for (int mt = wave; mt * 16 < ncol; mt += nwave) {
    const int c = mt * 16 + i;
    const bool cok = c < ncol;
    f32x4 acc[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int r0 = 0; r0 < nrow; r0 += 4 * GEN_KU) {
      float a[GEN_KU], bv[GEN_KU][NTL];
#pragma unroll
      for (int q = 0; q < GEN_KU; ++q) {
        const int r = r0 + 4 * q + g;
        const bool rok = r < nrow;
        const int rc = rok ? r : nrow - 1;
        const float av = W[(size_t)rc * ld + (cok ? c : 0)];
        a[q] = (cok && rok) ? av : 0.0f;
#pragma unroll
        for (int t = 0; t < NTL; ++t) { const float xv = in[rc * LS + 16 * t + i]; bv[q][t] = rok ? xv : 0.0f; }
      }
#pragma unroll
      for (int q = 0; q < GEN_KU; ++q)
#pragma unroll
        for (int t = 0; t < NTL; ++t) acc[t] = mfma4(a[q], bv[q][t], acc[t]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = mt * 16 + 4 * g + j;
      if (col < ncol) {
#pragma unroll
        for (int t = 0; t < NTL; ++t) out[col * LS + 16 * t + i] = acc[t][j];
      }
    }
  }
}
template <int LS>
__device__ __forceinline__ void gen_tile_dense(const float* __restrict__ W, const float* __restrict__ b, int nout, int nin,
                                               const float* in, float* out, bool relu) {
  if constexpr (LRF_GEN_MFMA) gen_tile_dense_mfma<LS>(W, b, nout, nin, in, out, relu);
  else gen_tile_dense_valu<LS>(W, b, nout, nin, in, out, relu);
}
template <int LS>
__device__ __forceinline__ void gen_tile_dense_t(const float* __restrict__ W, int nrow, int ncol, int ld, const float* in, float* out) {
  if constexpr (LRF_GEN_MFMA) gen_tile_dense_t_mfma<LS>(W, nrow, ncol, ld, in, out);
  else gen_tile_dense_t_valu<LS>(W, nrow, ncol, ld, in, out);
}
// positional_encoding (tensorBase.py:14-21) of D rows v[d][s] -> out[2 D F][s]: [sin(v_d 2^f)] then [cos(v_d 2^f)], row d * F + f inside each half
template <int LS>
__device__ __forceinline__ void gen_tile_encode(const float* v, int D, int F, bool on, float* out) {
  for (int e = threadIdx.x; e < D * F * LS; e += blockDim.x) {
    const int s = e % LS, df = e / LS, d = df / F, q = df % F;
    const float a = v[d * LS + s] * (float)(1 << q);
    out[df * LS + s] = on ? sinf(a) : 0.0f;
    out[(D * F + df) * LS + s] = on ? cosf(a) : 0.0f;
  }
}
// LDS layout of a block (floats): x [in1][LS] | h1 [max(fc, 72)][LS] (first the 72 plane x line products) | h2v [fc + inv][LS]
// | o [4][LS] | (backward) dz1 [fc][LS] | dx [in1][LS] | go [4][LS]
struct GenLds { int x, h1, h2v, o, dz1, dx, go, total; };
__host__ __device__ inline GenLds gen_lds(const GenCfg& g, int LS, bool bwd) {
  GenLds l;
  l.x = 0; l.h1 = l.x + g.in1 * LS; l.h2v = l.h1 + (g.fc > 72 ? g.fc : 72) * LS; l.o = l.h2v + (g.fc + g.inv) * LS;
  l.dz1 = l.o + 4 * LS; l.dx = l.dz1 + (bwd ? g.fc * LS : 0); l.go = l.dx + (bwd ? g.in1 * LS : 0); l.total = l.go + (bwd ? 4 * LS : 0);
  return l;
}
// samples per block: 32 unless the block's LDS image would pass 150 KB
__host__ __device__ inline int gen_tile_samples(const GenCfg& g, bool bwd) { return gen_lds(g, 32, bwd).total * 4 <= 150 * 1024 ? 32 : 16; }
__host__ __device__ inline int gen_block_threads(const GenCfg& g) {
#if LRF_GEN_MFMA
  (void)g; return 256;                   // four waves share the 16-unit tiles of a layer (a wave per SIMD and block; 3 blocks per CU)
#else
  const int t = ((g.fc > 64 ? g.fc : 64) + 63) / 64 * 64; return t > 256 ? 256 : t;
#endif
}

// the network on a block's LS samples: x rows 0..26 hold feat, dirs[s] the unit view direction of sample s (all the same ray).
// Leaves x (with encodings), h1 = relu(..), h2v = [relu(h2) | venc], o = pre-sigmoid colours.  MLPRender_Fea_late_view.forward
// (tensorBase.py:115-135).
template <int LS>
__device__ __forceinline__ void gen_tile_network(const DField& f, const GenCfg& g, const float dh[3], float* sm, const GenLds& l) {
  float* x = sm + l.x; float* h1 = sm + l.h1; float* h2v = sm + l.h2v; float* o = sm + l.o;
  if (g.fea_pe > 0) gen_tile_encode<LS>(x, LRF_APP_DIM, g.fea_pe, g.pe_on != 0, x + LRF_APP_DIM * LS);
  for (int e = threadIdx.x; e < 3 * LS; e += blockDim.x) h2v[(g.fc + e / LS) * LS + e % LS] = dh[e / LS];
  __syncthreads();
  gen_tile_dense<LS>(f.w1, f.b1, g.fc, g.in1, x, h1, true);
  if (g.view_pe > 0) gen_tile_encode<LS>(h2v + g.fc * LS, 3, g.view_pe, true, h2v + (g.fc + 3) * LS);
  __syncthreads();
  gen_tile_dense<LS>(f.w2, f.b2, g.fc, g.fc, h1, h2v, true);
  __syncthreads();
  gen_tile_dense<LS>(f.w3, f.b3, 3, g.fc + g.inv, h2v, o, false);   // the three colours (one 16-unit tile, three of its rows used)
  __syncthreads();
}

// toff16[r] = 2 toff32[r]: a ray owns two 16-row tiles per 32-sample tile (as k_shade3<SAVE>)
__global__ void k_toff16(const int* __restrict__ toff32, int R, int* __restrict__ toff16) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r <= R) toff16[r] = 2 * toff32[r];
}

// block = (ray, chunk q of LS consecutive compact samples).  Partial colours per 16 samples -> part[ray][16-sample slot]
// (k_finalize sums ceil(n / 16) of them).
template <int LS, bool SAVE>
__global__ __launch_bounds__(256) void k_shade_gen(
    DField f, GenCfg g, const float* __restrict__ rays, const float* __restrict__ z, int S,
    const int* __restrict__ ncomp, const uint16_t* __restrict__ cidx, const float* __restrict__ cw,
    float* __restrict__ part, int pmax, const int* __restrict__ toff32, float* __restrict__ crgb, float* __restrict__ act,
    int4* __restrict__ tileinfo) {
  extern __shared__ __attribute__((aligned(16))) float s_gen[];
  __shared__ float s_u[3][LS];                               // normalised sample positions
  const GenLds l = gen_lds(g, LS, false);
  const int nq = (pmax * 16 + LS - 1) / LS;                  // chunks per ray
  const int ray = blockIdx.x / nq, q = blockIdx.x % nq;
  const int nc = ncomp[ray];
  if (q * LS >= nc) return;                                   // the chunk is behind the ray's samples
  const int tid = threadIdx.x, j0 = q * LS, cnt = min(LS, nc - j0);
  const float* rp = rays + (size_t)ray * 6;
  const float o3[3] = {rp[0], rp[1], rp[2]};
  const float dn = sqrtf(rp[3] * rp[3] + rp[4] * rp[4] + rp[5] * rp[5]);
  const float dh[3] = {rp[3] / dn, rp[4] / dn, rp[5] / dn};
  float* x = s_gen + l.x; float* X = s_gen + l.h1; float* o = s_gen + l.o;
  if (tid < LS) {                                             // samples beyond the ray's count sit on its last one (results unused)
    const size_t ci = (size_t)ray * S + j0 + min(tid, cnt - 1);
    float xp[3], u[3];
    sample_point(f, o3, dh, z[cidx[ci]], xp, u);
    s_u[0][tid] = u[0]; s_u[1][tid] = u[1]; s_u[2][tid] = u[2];
  }
  __syncthreads();
  for (int e = tid; e < 18 * LS; e += blockDim.x) {           // the 72 plane x line products of every sample (tensoRF.py:153-195):
    const int s = e % LS, pq = e / LS, p = pq / 6, c4 = 4 * (pq % 6);   // a thread forms four channels of one plane from the dense
    const float u[3] = {s_u[0][s], s_u[1][s], s_u[2][s]};             // 24-channel texels (six 16-byte loads; the padded layout cost
    int x0, x1, y0, y1, l0, l1; float tx, ty, tl;                       // one 4-byte load per channel and tap: 1.2 of the 4.9 ms)
    tap1d(u[MAT0[p]], f.pw[p], x0, x1, tx);
    tap1d(u[MAT1[p]], f.ph[p], y0, y1, ty);
    tap1d(u[VEC[p]],  f.ll[p], l0, l1, tl);
    const float* pl = f.aplane2[p] + c4;
    const float* ln = f.aline2[p] + c4;
    const size_t r0 = (size_t)y0 * f.pw[p], r1 = (size_t)y1 * f.pw[p];
    const float4 a = *reinterpret_cast<const float4*>(pl + (r0 + x0) * LRF_CA), b = *reinterpret_cast<const float4*>(pl + (r0 + x1) * LRF_CA);
    const float4 c = *reinterpret_cast<const float4*>(pl + (r1 + x0) * LRF_CA), d = *reinterpret_cast<const float4*>(pl + (r1 + x1) * LRF_CA);
    const float4 e0 = *reinterpret_cast<const float4*>(ln + (size_t)l0 * LRF_CA), e1 = *reinterpret_cast<const float4*>(ln + (size_t)l1 * LRF_CA);
    const float w00 = (1.0f - tx) * (1.0f - ty), w10 = tx * (1.0f - ty), w01 = (1.0f - tx) * ty, w11 = tx * ty;
    const int row = p * LRF_CA + c4;
    X[(row + 0) * LS + s] = (a.x * w00 + b.x * w10 + c.x * w01 + d.x * w11) * (e0.x * (1.0f - tl) + e1.x * tl);
    X[(row + 1) * LS + s] = (a.y * w00 + b.y * w10 + c.y * w01 + d.y * w11) * (e0.y * (1.0f - tl) + e1.y * tl);
    X[(row + 2) * LS + s] = (a.z * w00 + b.z * w10 + c.z * w01 + d.z * w11) * (e0.z * (1.0f - tl) + e1.z * tl);
    X[(row + 3) * LS + s] = (a.w * w00 + b.w * w10 + c.w * w01 + d.w * w11) * (e0.w * (1.0f - tl) + e1.w * tl);
  }
  __syncthreads();
  gen_tile_dense<LS>(f.basis, nullptr, LRF_APP_DIM, 72, X, x, false);     // feat = basis_mat(X) (tensoRF.py:196)
  __syncthreads();
  gen_tile_network<LS>(f, g, dh, s_gen, l);
  if (tid < LS) {
    const int j = j0 + tid;
    const bool valid = tid < cnt;
    float cr = 0.0f, cg = 0.0f, cb = 0.0f;
    if (valid) {
      const size_t ci = (size_t)ray * S + j;
      const float w = cw[ci];
      const float s0 = 1.0f / (1.0f + expf(-o[tid])), s1 = 1.0f / (1.0f + expf(-o[LS + tid])), s2 = 1.0f / (1.0f + expf(-o[2 * LS + tid]));
      cr = w * s0; cg = w * s1; cb = w * s2;
      if (SAVE) {
        float* cp = crgb + ci * 3;
        cp[0] = s0; cp[1] = s1; cp[2] = s2;
        const size_t row = ((size_t)2 * (toff32[ray] + (j >> 5)) + ((j >> 4) & 1)) * 16 + (j & 15);
        for (int c = 0; c < LRF_APP_DIM; ++c) act[frag_off(row, ACT_FEAT + c, ACT_LD)] = x[c * LS + tid];
        act[frag_off(row, ACT_FEAT + LRF_APP_DIM, ACT_LD)] = 1.0f;
        if ((j & 31) == 0) {                                  // the first sample of a 32-sample tile writes its two 16-row tile records
          const int tir = j >> 5, c32 = min(32, nc - j);
          const size_t t16 = (size_t)2 * (toff32[ray] + tir);
          tileinfo[t16] = make_int4(ray, j, min(16, c32), 2 * tir);
          tileinfo[t16 + 1] = make_int4(ray, c32 > 16 ? j + 16 : j, max(0, c32 - 16), 2 * tir + 1);
        }
      }
    }
#pragma unroll
    for (int dd = 1; dd < 16; dd <<= 1) { cr += __shfl_xor(cr, dd, 64); cg += __shfl_xor(cg, dd, 64); cb += __shfl_xor(cb, dd, 64); }
    if ((j & 15) == 0 && valid) {                              // this 16-sample slot holds samples
      float* pp = part + ((size_t)ray * pmax + (j >> 4)) * 3;
      pp[0] = cr; pp[1] = cg; pp[2] = cb;
    }
  }
}

// A block per LS saved rows (LS / 16 of the 16-row tiles of tileinfo; LS = 32: the two tiles of one 32-sample tile, one ray).
// See the file header.
template <int LS>
__global__ __launch_bounds__(256) void k_gen_dgrad(
    DField f, GenCfg g, const float* __restrict__ rays, int S, const int* __restrict__ toff16, int R,
    const int4* __restrict__ tileinfo, const uint16_t* __restrict__ cidx, const float* __restrict__ cw,
    const float* __restrict__ crgb, const float* __restrict__ g_rgb, const float* __restrict__ act,
    float* __restrict__ grd, uint32_t* __restrict__ rowinfo, float* __restrict__ gen, int ld) {
  extern __shared__ __attribute__((aligned(16))) float s_gen[];
  const GenLds l = gen_lds(g, LS, true);
  const int tid = threadIdx.x;
  const size_t row0 = (size_t)blockIdx.x * LS;
  const int T = toff16[R];
  if (row0 >= (size_t)T * 16) return;
  const int4 ti0 = tileinfo[row0 >> 4];
  const int ray = ti0.x;                                       // (LS = 32: both tiles belong to this ray; the second may be missing when T is odd -- it is not: T is even)
  float* x = s_gen + l.x; float* h1 = s_gen + l.h1; float* h2v = s_gen + l.h2v; float* o = s_gen + l.o;
  float* dz1 = s_gen + l.dz1; float* dx = s_gen + l.dx; float* go = s_gen + l.go;
  const float* rp = rays + (size_t)ray * 6;
  const float dn = sqrtf(rp[3] * rp[3] + rp[4] * rp[4] + rp[5] * rp[5]);
  const float dh[3] = {rp[3] / dn, rp[4] / dn, rp[5] / dn};
  const GenRowOff ro = gen_row_off(g);
  // feat rows; rows beyond a tile's count were never written by the forward: zeros (finite activations; their gradients are zero)
  for (int e = tid; e < LRF_APP_DIM * LS; e += blockDim.x) {
    const int s = e % LS, c = e / LS;
    const int4 ti = tileinfo[(row0 + s) >> 4];
    x[c * LS + s] = (s & 15) < ti.z ? act[frag_off(row0 + s, ACT_FEAT + c, ACT_LD)] : 0.0f;
  }
  __syncthreads();
  gen_tile_network<LS>(f, g, dh, s_gen, l);
  if (tid < LS) {                                              // d(loss)/d(pre-sigmoid colour): rgb_map = sum_k w_k rgb_k (tensorBase.py:632-633)
    const size_t row = row0 + tid;
    const int4 ti = tileinfo[row >> 4];
    const int s16 = tid & 15;
    const bool valid = s16 < ti.z;
    float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
    if (valid) {
      const size_t ci = (size_t)ray * S + ti.y + s16;
      rowinfo[row] = (uint32_t)((size_t)ray * S + cidx[ci]);
      const float w = cw[ci];
      const int oray = f.perm ? f.perm[ray] : ray;
      const float r0 = crgb[ci * 3], r1 = crgb[ci * 3 + 1], r2 = crgb[ci * 3 + 2];
      g0 = g_rgb[(size_t)oray * 3] * w * r0 * (1.0f - r0);
      g1 = g_rgb[(size_t)oray * 3 + 1] * w * r1 * (1.0f - r1);
      g2 = g_rgb[(size_t)oray * 3 + 2] * w * r2 * (1.0f - r2);
    } else {
      rowinfo[row] = 0xffffffffu;
    }
    go[tid] = g0; go[LS + tid] = g1; go[2 * LS + tid] = g2; go[3 * LS + tid] = 0.0f;
  }
  __syncthreads();
  // operand rows of the weight gradients, part 1 (before the activations are overwritten): [x, 1], [relu(h1), 1], [relu(h2), venc, 1], go
  for (int s = 0; s < LS; ++s) {
    float* gr = gen + (row0 + s) * (size_t)ld;
    const bool live = go[s] != 0.0f || go[LS + s] != 0.0f || go[2 * LS + s] != 0.0f;     // rows without gradient contribute nothing: zero rows
    for (int e = tid; e < ld; e += blockDim.x) {
      float v = 0.0f;
      if (live) {
        if (e >= ro.go) v = e - ro.go < 3 ? go[(e - ro.go) * LS + s] : 0.0f;
        else if (e >= ro.h2v) v = e - ro.h2v < g.fc + g.inv ? h2v[(e - ro.h2v) * LS + s] : 1.0f;
        else if (e >= ro.h1) v = e - ro.h1 < g.fc ? h1[(e - ro.h1) * LS + s] : 1.0f;
        else if (e >= ro.x1) v = e - ro.x1 < g.in1 ? x[(e - ro.x1) * LS + s] : 1.0f;
      }
      if (e >= ro.x1) gr[e] = v;                               // (dz1, dz2 columns follow below)
    }
  }
  __syncthreads();
  const int ld3 = g.fc + g.inv;
  for (int u = tid; u < g.fc; u += blockDim.x) {               // dz2 = (W3[:, :fc]^T go) * [h2 > 0], in place of relu(h2)
    const float w0 = f.w3[u], w1 = f.w3[ld3 + u], w2 = f.w3[2 * ld3 + u];
    for (int s = 0; s < LS; ++s) {
      const float v = w0 * go[s] + w1 * go[LS + s] + w2 * go[2 * LS + s];
      const float d2 = h2v[u * LS + s] > 0.0f ? v : 0.0f;
      h2v[u * LS + s] = d2;
      gen[(row0 + s) * (size_t)ld + ro.dz2 + u] = d2;
    }
  }
  __syncthreads();
  gen_tile_dense_t<LS>(f.w2, g.fc, g.fc, g.fc, h2v, dz1);
  __syncthreads();
  for (int v = tid; v < g.fc; v += blockDim.x)
    for (int s = 0; s < LS; ++s) {
      const float d1 = h1[v * LS + s] > 0.0f ? dz1[v * LS + s] : 0.0f;
      dz1[v * LS + s] = d1;
      gen[(row0 + s) * (size_t)ld + ro.dz1 + v] = d1;
    }
  __syncthreads();
  gen_tile_dense_t<LS>(f.w1, g.fc, g.in1, g.in1, dz1, dx);
  __syncthreads();
  const int F = g.fea_pe, DF = LRF_APP_DIM * F;
  for (int e = tid; e < 32 * LS; e += blockDim.x) {            // dfeat block of the gradient row (32 columns; 27.. are zero)
    const int s = e % LS, d = e / LS;
    float a = 0.0f;
    if (d < LRF_APP_DIM) {
      a = dx[d * LS + s];
      if (F > 0 && g.pe_on)
        for (int qf = 0; qf < F; ++qf) {                       // d sin(v 2^q) = 2^q cos(..), d cos(v 2^q) = -2^q sin(..): the encodings are in x
          const float sc = (float)(1 << qf);
          a += sc * (x[(LRF_APP_DIM + DF + d * F + qf) * LS + s] * dx[(LRF_APP_DIM + d * F + qf) * LS + s]
                     - x[(LRF_APP_DIM + d * F + qf) * LS + s] * dx[(LRF_APP_DIM + DF + d * F + qf) * LS + s]);
        }
    }
    grd[frag_off(row0 + s, GRD_DFEAT + d, GRD_LD)] = a;
  }
}

// C[m][n] += sum over the K-chunk's rows of A[row][m] B[row][n], A = gen + offA (M columns), B = gen + offB (N columns; the
// last one is the constant 1 of the bias).  64 x 64 output tile per block, 4 x 4 per thread, rows staged 16 at a time.
// Columns n < N - 1 go to dW[m * ldw + n], column N - 1 to db[m].
constexpr int GEN_GEMM_CHUNK = 8192;
__global__ __launch_bounds__(256) void k_gen_gemm(const float* __restrict__ gen, int ld, int offA, int M, int offB, int N,
                                                  const int* __restrict__ toff16, int R, float* __restrict__ dW, int ldw, float* __restrict__ db) {
  __shared__ float sA[16][64], sB[16][64];
  const int rows = toff16[R] * 16;
  const int r0 = blockIdx.y * GEN_GEMM_CHUNK;
  if (r0 >= rows) return;
  const int r1 = min(rows, r0 + GEN_GEMM_CHUNK);
  const int ntn = (N + 63) / 64;
  const int m0 = (blockIdx.x / ntn) * 64, n0 = (blockIdx.x % ntn) * 64;
  const int tm = threadIdx.x >> 4, tn = threadIdx.x & 15;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
  for (int rb = r0; rb < r1; rb += 16) {
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {
      const int rr = e >> 6, cc = e & 63, row = rb + rr;
      const bool ok = row < r1;
      sA[rr][cc] = (ok && m0 + cc < M) ? gen[(size_t)row * ld + offA + m0 + cc] : 0.0f;
      sB[rr][cc] = (ok && n0 + cc < N) ? gen[(size_t)row * ld + offB + n0 + cc] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = sA[rr][4 * tm + i]; b[i] = sB[rr][4 * tn + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + 4 * tm + i, n = n0 + 4 * tn + j;
      if (m < M && n < N && acc[i][j] != 0.0f) {
        if (n < N - 1) unsafeAtomicAdd(dW + (size_t)m * ldw + n, acc[i][j]);
        else unsafeAtomicAdd(db + m, acc[i][j]);
      }
    }
}

}  // namespace lrf
