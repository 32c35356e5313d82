// lrf_generic.inl -- the colour network for ANY MLPRender_Fea_late_view configuration (tensorBase.py:97-135): positional
// encodings of the appearance features (fea_pe) and of the view direction (view_pe), any hidden width featureC <= 256.
// The fast kernels (k_shade3, k_train_dgrad3, k_wgrad_w2w3) are specialised to opt.py's defaults (0 / 0 / 128), which is what
// train.py runs; every other configuration takes this engine: plain fp32 loops on the vector ALU over the parameter
// tensors in their natural layout, one lane per sample -- correct, differentiable, not tuned (expect 10-30 x the default
// engine's time).  It is also the LRF_FLAG_MLP_VALU debug engine of the default configuration.
//
// Forward (k_shade_gen): a block renders two 32-sample tiles of one ray; with SAVE it leaves what the backward needs in the
// same places as k_shade3<SAVE> (colours, feat rows, tile records of the 16-row tiles) -- no mask bits: the backward
// recomputes the network from the saved feat row with the same arithmetic, so its ReLU signs are the forward's.
// Backward: k_gen_dgrad, one lane per saved row: recompute, d(loss)/d(pre-sigmoid) -> dz2 -> dz1 -> d(input) -> dfeat (through
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

// positional_encoding (tensorBase.py:14-21): [sin(v_d 2^f)] then [cos(v_d 2^f)], index d * F + f inside each half
__device__ __forceinline__ void gen_encode(const float* v, int D, int F, bool on, float* out /* [2 D F] */) {
  for (int d = 0; d < D; ++d)
    for (int q = 0; q < F; ++q) {
      const float a = v[d] * (float)(1 << q);
      out[d * F + q] = on ? sinf(a) : 0.0f;
      out[D * F + d * F + q] = on ? cosf(a) : 0.0f;
    }
}

// the 27 appearance features of one sample (tensoRF.py:153-196) from the padded 32-channel texels
__device__ __forceinline__ void gen_app_features(const DField& f, const float u[3], float fe[LRF_APP_DIM]) {
  float X[72];
  for (int p = 0; p < 3; ++p) {
    int x0, x1, y0, y1, l0, l1; float tx, ty, tl;
    tap1d(u[MAT0[p]], f.pw[p], x0, x1, tx);
    tap1d(u[MAT1[p]], f.ph[p], y0, y1, ty);
    tap1d(u[VEC[p]],  f.ll[p], l0, l1, tl);
    const float* pl = f.aplane[p];
    for (int c = 0; c < LRF_CA; ++c) {
      const int pc = app_pc(c);
      const float v = pl[((size_t)y0 * f.pw[p] + x0) * LRF_CAS + pc] * ((1.0f - tx) * (1.0f - ty))
                    + pl[((size_t)y0 * f.pw[p] + x1) * LRF_CAS + pc] * (tx * (1.0f - ty))
                    + pl[((size_t)y1 * f.pw[p] + x0) * LRF_CAS + pc] * ((1.0f - tx) * ty)
                    + pl[((size_t)y1 * f.pw[p] + x1) * LRF_CAS + pc] * (tx * ty);
      const float l = f.aline[p][(size_t)l0 * LRF_CAS + pc] * (1.0f - tl) + f.aline[p][(size_t)l1 * LRF_CAS + pc] * tl;
      X[p * LRF_CA + c] = v * l;
    }
  }
  for (int i = 0; i < LRF_APP_DIM; ++i) {
    float a = 0.0f;
    for (int c = 0; c < 72; ++c) a += f.basis[i * 72 + c] * X[c];
    fe[i] = a;
  }
}

// MLPRender_Fea_late_view.forward (tensorBase.py:115-135) for one sample: x = [feat, PE(feat)], h1 = relu(W1 x + b1),
// h2 = relu(W2 h1 + b2), o = W3 [h2, d, PE(d)] + b3 (pre-sigmoid).  Arrays are the caller's (private memory).
__device__ __forceinline__ void gen_network(const DField& f, const GenCfg& g, const float* feat, const float dh[3],
                                            float* x, float* h1, float* h2, float* venc, float o[3]) {
  for (int c = 0; c < LRF_APP_DIM; ++c) x[c] = feat[c];
  if (g.fea_pe > 0) gen_encode(feat, LRF_APP_DIM, g.fea_pe, g.pe_on != 0, x + LRF_APP_DIM);
  for (int i = 0; i < g.fc; ++i) {
    float a = f.b1[i];
    const float* wr = f.w1 + (size_t)i * g.in1;
    for (int c = 0; c < g.in1; ++c) a += wr[c] * x[c];
    h1[i] = fmaxf(a, 0.0f);
  }
  for (int i = 0; i < g.fc; ++i) {
    float a = f.b2[i];
    const float* wr = f.w2 + (size_t)i * g.fc;
    for (int c = 0; c < g.fc; ++c) a += wr[c] * h1[c];
    h2[i] = fmaxf(a, 0.0f);
  }
  venc[0] = dh[0]; venc[1] = dh[1]; venc[2] = dh[2];
  if (g.view_pe > 0) gen_encode(dh, 3, g.view_pe, true, venc + 3);
  const int ld3 = g.fc + g.inv;
  for (int c = 0; c < 3; ++c) {
    float a = f.b3[c];
    const float* wr = f.w3 + (size_t)c * ld3;
    for (int u = 0; u < g.fc; ++u) a += wr[u] * h2[u];
    for (int j = 0; j < g.inv; ++j) a += wr[g.fc + j] * venc[j];
    o[c] = a;
  }
}

// toff16[r] = 2 toff32[r]: a ray owns two 16-row tiles per 32-sample tile (as k_shade3<SAVE>)
__global__ void k_toff16(const int* __restrict__ toff32, int R, int* __restrict__ toff16) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r <= R) toff16[r] = 2 * toff32[r];
}

// block = (ray, pair q of 32-sample tiles): lanes 0..31 tile 2 q, lanes 32..63 tile 2 q + 1; lane & 31 = sample of the tile.
// Partial colours per 16 samples -> part[ray][2 t32 + half] (k_finalize sums ceil(n / 16) of them).
template <bool SAVE>
__global__ __launch_bounds__(64) void k_shade_gen(
    DField f, GenCfg g, const float* __restrict__ rays, const float* __restrict__ z, int S,
    const int* __restrict__ ncomp, const uint16_t* __restrict__ cidx, const float* __restrict__ cw,
    float* __restrict__ part, int pmax, const int* __restrict__ toff32, float* __restrict__ crgb, float* __restrict__ act,
    int4* __restrict__ tileinfo) {
  const int npair = (pmax + 3) / 4;                          // pairs of 32-sample tiles per ray (pmax 16-sample slots)
  const int ray = blockIdx.x / npair, q = blockIdx.x % npair;
  const int lane = threadIdx.x, n = lane & 31, tir = 2 * q + (lane >> 5);
  const int nc = ncomp[ray];
  const int j0 = tir * 32, cnt = min(32, nc - j0);
  if (nc - 2 * q * 32 <= 0) return;                          // both tiles of the block are behind the ray's samples
  const bool valid = cnt > 0 && n < cnt;
  const float* rp = rays + (size_t)ray * 6;
  const float o3[3] = {rp[0], rp[1], rp[2]};
  const float dn = sqrtf(rp[3] * rp[3] + rp[4] * rp[4] + rp[5] * rp[5]);
  const float dh[3] = {rp[3] / dn, rp[4] / dn, rp[5] / dn};
  float cr = 0.0f, cg = 0.0f, cb = 0.0f;
  if (valid) {
    const size_t ci = (size_t)ray * S + j0 + n;
    const int k = cidx[ci];
    const float w = cw[ci];
    float xp[3], u[3];
    sample_point(f, o3, dh, z[k], xp, u);
    float fe[LRF_APP_DIM];
    gen_app_features(f, u, fe);
    float x[GEN_MAX_IN1], h1[GEN_MAX_FC], h2[GEN_MAX_FC], venc[GEN_MAX_INV], o[3];
    gen_network(f, g, fe, dh, x, h1, h2, venc, o);
    const float s0 = 1.0f / (1.0f + expf(-o[0])), s1 = 1.0f / (1.0f + expf(-o[1])), s2 = 1.0f / (1.0f + expf(-o[2]));
    cr = w * s0; cg = w * s1; cb = w * s2;
    if (SAVE) {
      float* cp = crgb + ci * 3;
      cp[0] = s0; cp[1] = s1; cp[2] = s2;
      const size_t row = ((size_t)2 * (toff32[ray] + tir) + (n >> 4)) * 16 + (n & 15);
      for (int c = 0; c < LRF_APP_DIM; ++c) act[frag_off(row, ACT_FEAT + c, ACT_LD)] = fe[c];
      act[frag_off(row, ACT_FEAT + LRF_APP_DIM, ACT_LD)] = 1.0f;
    }
  }
  if (SAVE && cnt > 0 && n == 0) {
    const size_t t16 = (size_t)2 * (toff32[ray] + tir);
    tileinfo[t16] = make_int4(ray, j0, min(16, cnt), 2 * tir);
    tileinfo[t16 + 1] = make_int4(ray, cnt > 16 ? j0 + 16 : j0, max(0, cnt - 16), 2 * tir + 1);
  }
#pragma unroll
  for (int dd = 1; dd < 16; dd <<= 1) { cr += __shfl_xor(cr, dd, 64); cg += __shfl_xor(cg, dd, 64); cb += __shfl_xor(cb, dd, 64); }
  if ((n & 15) == 0 && cnt > 16 * ((n >> 4))) {                // this 16-sample slot holds samples
    float* pp = part + ((size_t)ray * pmax + 2 * tir + (n >> 4)) * 3;
    pp[0] = cr; pp[1] = cg; pp[2] = cb;
  }
}

// One lane per saved row (16-row tiles of tileinfo).  See the file header.
__global__ __launch_bounds__(64) void k_gen_dgrad(
    DField f, GenCfg g, const float* __restrict__ rays, int S, const int* __restrict__ toff16, int R,
    const int4* __restrict__ tileinfo, const uint16_t* __restrict__ cidx, const float* __restrict__ cw,
    const float* __restrict__ crgb, const float* __restrict__ g_rgb, const float* __restrict__ act,
    float* __restrict__ grd, uint32_t* __restrict__ rowinfo, float* __restrict__ gen, int ld) {
  const size_t row = (size_t)blockIdx.x * 64 + threadIdx.x;
  const int T = toff16[R];
  if (row >= (size_t)T * 16) return;
  const int tile = (int)(row >> 4), s = (int)(row & 15);
  const int4 ti = tileinfo[tile];
  const int ray = ti.x;
  const bool valid = s < ti.z;
  const GenRowOff ro = gen_row_off(g);
  float* gr = gen + row * (size_t)ld;
  for (int c = 0; c < LRF_APP_DIM + 5; ++c) grd[frag_off(row, GRD_DFEAT + c, GRD_LD)] = 0.0f;   // dfeat block (32 columns)
  if (!valid) {
    rowinfo[row] = 0xffffffffu;
    for (int c = 0; c < ld; ++c) gr[c] = 0.0f;
    return;
  }
  const size_t ci = (size_t)ray * S + ti.y + s;
  rowinfo[row] = (uint32_t)((size_t)ray * S + cidx[ci]);
  const float* rp = rays + (size_t)ray * 6;
  const float dn = sqrtf(rp[3] * rp[3] + rp[4] * rp[4] + rp[5] * rp[5]);
  const float dh[3] = {rp[3] / dn, rp[4] / dn, rp[5] / dn};
  float fe[LRF_APP_DIM];
  for (int c = 0; c < LRF_APP_DIM; ++c) fe[c] = act[frag_off(row, ACT_FEAT + c, ACT_LD)];
  float x[GEN_MAX_IN1], h1[GEN_MAX_FC], h2[GEN_MAX_FC], venc[GEN_MAX_INV], o[3];
  gen_network(f, g, fe, dh, x, h1, h2, venc, o);
  // d(loss)/d(pre-sigmoid colour): rgb_map = sum_k w_k rgb_k (tensorBase.py:632-633), the saved sigmoid values
  const float w = cw[ci];
  const int oray = f.perm ? f.perm[ray] : ray;
  float go[3];
  for (int c = 0; c < 3; ++c) {
    const float r = crgb[ci * 3 + c];
    go[c] = g_rgb[(size_t)oray * 3 + c] * w * r * (1.0f - r);
  }
  const int ld3 = g.fc + g.inv;
  float dz2[GEN_MAX_FC], dz1[GEN_MAX_FC], dx[GEN_MAX_IN1];
  for (int u = 0; u < g.fc; ++u) {
    const float v = f.w3[u] * go[0] + f.w3[ld3 + u] * go[1] + f.w3[2 * ld3 + u] * go[2];
    dz2[u] = h2[u] > 0.0f ? v : 0.0f;
  }
  for (int v = 0; v < g.fc; ++v) {
    float a = 0.0f;
    for (int u = 0; u < g.fc; ++u) a += f.w2[(size_t)u * g.fc + v] * dz2[u];
    dz1[v] = h1[v] > 0.0f ? a : 0.0f;
  }
  for (int c = 0; c < g.in1; ++c) {
    float a = 0.0f;
    for (int v = 0; v < g.fc; ++v) a += f.w1[(size_t)v * g.in1 + c] * dz1[v];
    dx[c] = a;
  }
  const int F = g.fea_pe, DF = LRF_APP_DIM * F;
  for (int d = 0; d < LRF_APP_DIM; ++d) {
    float a = dx[d];
    if (F > 0 && g.pe_on)
      for (int qf = 0; qf < F; ++qf) {                         // d sin(v 2^q) = 2^q cos(..), d cos(v 2^q) = -2^q sin(..): the encodings are in x
        const float sc = (float)(1 << qf);
        a += sc * (x[LRF_APP_DIM + DF + d * F + qf] * dx[LRF_APP_DIM + d * F + qf] - x[LRF_APP_DIM + d * F + qf] * dx[LRF_APP_DIM + DF + d * F + qf]);
      }
    grd[frag_off(row, GRD_DFEAT + d, GRD_LD)] = a;
  }
  for (int v = 0; v < g.fc; ++v) { gr[ro.dz1 + v] = dz1[v]; gr[ro.dz2 + v] = dz2[v]; gr[ro.h1 + v] = h1[v]; gr[ro.h2v + v] = h2[v]; }
  for (int c = 0; c < g.in1; ++c) gr[ro.x1 + c] = x[c];
  gr[ro.x1 + g.in1] = 1.0f; gr[ro.h1 + g.fc] = 1.0f;
  for (int j = 0; j < g.inv; ++j) gr[ro.h2v + g.fc + j] = venc[j];
  gr[ro.h2v + g.fc + g.inv] = 1.0f;
  gr[ro.go] = go[0]; gr[ro.go + 1] = go[1]; gr[ro.go + 2] = go[2]; gr[ro.go + 3] = 0.0f;
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
