// lrf_train32.inl -- the colour network's backward on the k_shade3 skeleton (round 4): 32 samples per wave on
// v_mfma_f32_32x32x16_bf16, 512-thread workgroups, compiler-scheduled term-major chain.  Included by lrf_backward.inl.
//
// The round-3 kernel it replaces (16 samples per wave on v_mfma_f32_16x16x32_bf16, 1024-thread workgroups, 128 registers) issued ~1650
// VALU + 143 MFMA + 179 LDS instructions per 16-sample tile and spent half its wave-cycles in issue stalls
// (profiles/r11a: 320-330 us).  Here a wave takes two consecutive 16-row tiles of the saved rows (lane n = lane & 31 is
// row 16 (n >> 4) + (n & 15) of the pair, h = lane >> 5 the K half), so every A fragment feeds 32 columns and the
// D registers of a product are again the B operand of the next one.  Two kernels:
//   k_train_dgrad3  go -> dz2 (VALU: rank 3, gated by the saved layer-2 mask bits) -> dz1 = W2^T dz2 (4 x 8 x 3 MFMAs, gated by
//                   the layer-1 bits) -> dfeat = W1^T dz1 (8 x 3); writes the go block and dfeat; accumulates dW1.
//   k_train_app3    dfeat -> dX = basis^T dfeat (3 x 2 x 3), delivered per lane for exactly the channels its half gathers
//                   (12 h .. 12 h + 11 of each plane, the dense 24-channel texels of k_shade3); re-gathers the taps for the
//                   position gradient of the appearance lookups, which also yields X = plane x line again -- so
//                   dbasis = dfeat^T X is accumulated here and X is never a row (it was 320 B per shaded sample, written
//                   by the forward and read once by a GEMM kernel).
// The transposed network is packed with the layout cache (slice 18 of k_pack_planes) into the same fragment format as the
// forward image (pack_mlp_w32_t_elem).
// Rows are written in the 16-row fragment order the weight-gradient kernel and the scatter kernels read (lrf_common.h);
// dX goes out as three 48-byte pieces per lane (round 3: nine 8-byte stores).
//
// dW1 (+ db1) = dz1^T [feat | 1] and dbasis = dfeat^T X are accumulated in registers, so dz1 and X are never rows (with the
// dz1 rows the data-gradient kernel ran 310-430 us against 160 us without any row store -- it was bound by its own stores).
// These products contract over samples, which sit in the lane dimension of the chain's registers; both operands are therefore
// TRANSPOSED ON THE MATRIX PIPE, by multiplying with a 0/1 selector: with the sample-major fragment as A and a selector
// as B, D[sample][unit] arrives with lane = unit and registers = samples -- exactly an A / B operand over K = samples
// (the K order is again folded into w32_unit).  dz1's split halves exist anyway (B operand of the dfeat product): the
// transpose costs 2 x 8 MFMAs, feat's 2 x 2, the product 4 x 2 x 3; selectors are exact in bf16, so the transposed value
// is hi + lo of the original (what a split product sees anyway).  64 (dW1) / 48 (dbasis) accumulator registers per wave,
// reduced over the workgroup's waves once at the end -> one partial block per workgroup (WP_W1 / WP_BAS), summed by
// k_wgrad_reduce.
// (included inside namespace lrf)
#pragma once

// ---- transposed w32 image: fragments [part hi, lo][lane 64][8 bf16], slot j of lane (n, h) = A[n][8 h + j]
//   frag 8 m + ks        (0..31)  W2^T:    A[n][slot] = W2[u = 16 ks + 8 h + j][v = 32 m + n]
//   frag 32 + 2 m + q    (32..39) W1^T:    A[n][slot] = W1[v = w32_unit(m, q, h, j)][f = n]          (rows n >= 27 zero)
//   frag 40 + 2 mt + q   (40..45) basis^T: A[n][slot] = basis[f = w32_unit(0, q, h, j)][channel of row n of tile mt]
//                                          row n <-> (half hr = (n >> 2) & 1, value vv = 16 mt + 4 (n >> 3) + (n & 3)):
//                                          channel w32_chan(hr, vv), zero rows for vv >= 36
// then an fp32 tail: mlp_view.0.weight[c][0..127] padded to 132 per colour.
// (W32T_* constants: lrf_common.h)

__device__ void pack_mlp_w32_t_elem(const LrfParams& p, uint32_t* __restrict__ img, int idx, int basis_only /* generic engine: only k_train_app3's basis^T fragments exist in these shapes */) {
  if (idx >= W32T_ALL_U4 * 4) return;
  if (basis_only && (idx >= W32T_U4 * 4 || (idx >> 9) < W32T_BAS)) { img[idx] = 0u; return; }
  if (idx >= W32T_U4 * 4) {
    const int e = idx - W32T_U4 * 4, c = e / W32T_T_W3_LD, u = e % W32T_T_W3_LD;
    img[idx] = __float_as_uint(c < 3 && u < LRF_FEATC ? p.w3[c * (LRF_FEATC + 3) + u] : 0.0f);
    return;
  }
  const int u4 = idx >> 2, wj = idx & 3;
  const int lane = u4 & 63, part = (u4 >> 6) & 1, frag = u4 >> 7;
  const int n = lane & 31, h = lane >> 5;
  unsigned short out[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int j = 2 * wj + e;
    float v = 0.0f;
    if (frag < W32T_W1) {
      const int m = frag >> 3, ks = frag & 7;
      v = p.w2[(16 * ks + 8 * h + j) * LRF_FEATC + 32 * m + n];
    } else if (frag < W32T_BAS) {
      const int m = (frag - W32T_W1) >> 1, q = (frag - W32T_W1) & 1;
      if (n < LRF_APP_DIM) v = p.w1[w32_unit(m, q, h, j) * LRF_APP_DIM + n];
    } else {
      const int mt = (frag - W32T_BAS) >> 1, q = (frag - W32T_BAS) & 1;
      const int hr = (n >> 2) & 1, vv = 16 * mt + 4 * (n >> 3) + (n & 3);
      const int ch = vv < 36 ? w32_chan(hr, vv) : -1;
      const int f = w32_unit(0, q, h, j);
      if (ch >= 0 && f < LRF_APP_DIM) v = p.basis[f * 72 + ch];
    }
    const unsigned short hi = bf16_bits(v);
    out[e] = part ? bf16_bits(v - bf16_val(hi)) : hi;
  }
  img[idx] = (uint32_t)out[0] | ((uint32_t)out[1] << 16);
}

// position-gradient terms of plane p for lane half h: channels 12 h .. 12 h + 11 of the dense texels (gather_app12's taps),
// dX[12] = d(loss)/d(X) of those channels.  Adds to gu[] the derivative with respect to the three normalised coordinates
// and returns the products X[12] = plane x line of those channels (tensoRF.py:153-195).
// Two channels per instruction: the float4 of a tap is two aligned register pairs, the arithmetic below is written on
// float2 vectors with straight halves (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32; finding 17 concerns crossed low selects
// only, which this form cannot produce) -- 16 packed operations per pair of channels instead of ~50 scalar ones.
// vm: running max of |dX line| and |dX plane| -- the bound every contribution of the fixed-point appearance scatter obeys
// (k_scatter_fix<24, true>: what it adds to a plane tap is dX line times a bilinear weight, to a line cell dX plane times a
// linear weight)
template <int p>
__device__ __forceinline__ void app12_position_grad(const DField& f, const int i0[3], const int i1[3], const float t[3],
                                                    const float gm[3], int h, const float dX[12], float gu[3], float X[12], float& vm) {
  const int x0 = i0[MAT0[p]], x1 = i1[MAT0[p]], y0 = i0[MAT1[p]], y1 = i1[MAT1[p]];
  const int l0 = i0[VEC[p]], l1 = i1[VEC[p]];
  const f32x2v tx = {t[MAT0[p]], t[MAT0[p]]}, ty = {t[MAT1[p]], t[MAT1[p]]}, tl = {t[VEC[p]], t[VEC[p]]};
  const unsigned hb = 48u * (unsigned)h;
  const unsigned row0 = (unsigned)y0 * (unsigned)f.pw[p], row1 = (unsigned)y1 * (unsigned)f.pw[p];
  const unsigned o00 = (row0 + x0) * (LRF_CA * 4u) + hb, o10 = (row0 + x1) * (LRF_CA * 4u) + hb;
  const unsigned o01 = (row1 + x0) * (LRF_CA * 4u) + hb, o11 = (row1 + x1) * (LRF_CA * 4u) + hb;
  const unsigned q0 = (unsigned)l0 * (LRF_CA * 4u) + hb, q1 = (unsigned)l1 * (LRF_CA * 4u) + hb;
  f32x2v gix = {0.0f, 0.0f}, giy = {0.0f, 0.0f}, gil = {0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float4 a4 = ld4b(f.aplane2[p], o00 + 16 * i), b4 = ld4b(f.aplane2[p], o10 + 16 * i);
    const float4 c4 = ld4b(f.aplane2[p], o01 + 16 * i), d4 = ld4b(f.aplane2[p], o11 + 16 * i);
    const float4 e4 = ld4b(f.aline2[p], q0 + 16 * i), g4 = ld4b(f.aline2[p], q1 + 16 * i);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const f32x2v v00 = c ? f32x2v{a4.z, a4.w} : f32x2v{a4.x, a4.y}, v10 = c ? f32x2v{b4.z, b4.w} : f32x2v{b4.x, b4.y};
      const f32x2v v01 = c ? f32x2v{c4.z, c4.w} : f32x2v{c4.x, c4.y}, v11 = c ? f32x2v{d4.z, d4.w} : f32x2v{d4.x, d4.y};
      const f32x2v e0 = c ? f32x2v{e4.z, e4.w} : f32x2v{e4.x, e4.y}, e1 = c ? f32x2v{g4.z, g4.w} : f32x2v{g4.x, g4.y};
      const f32x2v d0 = v10 - v00, d1 = v11 - v01;             // d/dx along the two rows of the cell
      const f32x2v r0 = tx * d0 + v00, r1 = tx * d1 + v01;     // the rows at tx
      const f32x2v dy = r1 - r0;                               // d(plane)/d(ty)
      const f32x2v P = ty * dy + r0;
      const f32x2v dxp = ty * (d1 - d0) + d0;                  // d(plane)/d(tx)
      const f32x2v de = e1 - e0;                               // d(line)/d(tl)
      const f32x2v Lv = tl * de + e0;
      const f32x2v d = {dX[4 * i + 2 * c], dX[4 * i + 2 * c + 1]};
      const f32x2v x2 = P * Lv;
      X[4 * i + 2 * c] = x2[0]; X[4 * i + 2 * c + 1] = x2[1];
      const f32x2v dP = d * Lv, dL = d * P;
      vm = fmaxf(fmaxf(vm, fabsf(dP[0])), fabsf(dP[1]));
      vm = fmaxf(fmaxf(vm, fabsf(dL[0])), fabsf(dL[1]));
      gix = dP * dxp + gix;
      giy = dP * dy + giy;
      gil = dL * de + gil;
    }
  }
  gu[MAT0[p]] += (gix[0] + gix[1]) * gm[MAT0[p]]; gu[MAT1[p]] += (giy[0] + giy[1]) * gm[MAT1[p]]; gu[VEC[p]] += (gil[0] + gil[1]) * gm[VEC[p]];
}

// plain (temporal) 16-byte row store: for this kernel's 512-byte segments the `nt` hint of the 16-sample kernels costs time
// (1 KB rows: 411 us with nt, 312 us without, profiles/r11c)
__device__ __forceinline__ void row_store_plain(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// tileinfo[t] = (ray, first compact sample j0, samples in the tile, tile number inside the ray), written by k_shade3<SAVE>
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_train_dgrad3(
    DField f, const uint4* __restrict__ imt, const float* __restrict__ rays, int S,
    const int* __restrict__ toff, int R, const int4* __restrict__ tileinfo,
    const uint16_t* __restrict__ cidx, const float* __restrict__ cw, const float* __restrict__ crgb,
    const float* __restrict__ g_rgb, float* __restrict__ grd, uint32_t* __restrict__ rowinfo,
    const uint32_t* __restrict__ relu_bits, const float* __restrict__ act /* saved feat rows */, float* __restrict__ wpart,
    int dbg /* timing experiments: 1 no row stores, 4 no products */) {
  constexpr int NT = NW * 64;
  extern __shared__ uint4 s_dyn3[];
  uint4* img = s_dyn3;
  float* tail = reinterpret_cast<float*>(s_dyn3 + W32T_U4);
  uint4* s_sel = s_dyn3 + W32T_ALL_U4;                          // the four 0 / 1 selectors, [selector][lane] (built once per workgroup)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, h = lane >> 5, s = n & 15;
  for (int i = tid; i < 4 * 64; i += NT) {
    // selectors (B operands of the transposing products): slot j of lane (n, h) is 1 where the K index of the slot equals the
    // lane's column.  0, 1: K index = feature 16 ks + 8 h + j (the order the feat row is loaded in); 2, 3: K index = unit
    // 16 q + 8 (j >> 2) + 4 h + (j & 3) of a 32-unit M-tile (dz1's D-register order)
    const int id = i >> 6, ln = i & 63, nn = ln & 31, hh = ln >> 5;
    uint32_t wds[4];
#pragma unroll
    for (int q2 = 0; q2 < 4; ++q2) {
      uint32_t wd = 0;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = 2 * q2 + e;
        const int key = id < 2 ? 16 * id + 8 * hh + j : 16 * (id - 2) + 8 * (j >> 2) + 4 * hh + (j & 3);
        if (key == nn) wd |= e ? 0x3f800000u : 0x3f80u;
      }
      wds[q2] = wd;
    }
    s_sel[i] = make_uint4(wds[0], wds[1], wds[2], wds[3]);
  }
  {
    const int rot = (int)((blockIdx.x * 37u) % 93u) * 64;     // every workgroup starts its copy somewhere else (finding 18)
    for (int i = tid; i < W32T_ALL_U4; i += NT) { int j = i + rot; if (j >= W32T_ALL_U4) j -= W32T_ALL_U4; img[j] = imt[j]; }
  }
  __syncthreads();
  const int T = toff[R];
  const int P = (T + 1) >> 1;                                 // pairs of 16-row tiles
  const int nb = gridDim.x;
  const int lb = (nb % 8 == 0) ? (int)(blockIdx.x % 8) * (nb / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;   // XCD-aware order
  const long long wid = (long long)lb * NW + wave, waves = (long long)nb * NW;
  const int p_beg = (int)(wid * P / waves), p_end = (int)((wid + 1) * P / waves);
  f32x16 w1acc[4];                                             // dW1 partial of this wave: [v = 32 m + 8 (r >> 2) + 4 h + (r & 3)][f = n]
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) w1acc[m][r] = 0.0f;
  auto selector = [&](int id) { return __builtin_bit_cast(bf16x8, s_sel[id * 64 + lane]); };
  int4 ti_c = make_int4(0, 0, 0, 0);
  if (p_beg < p_end) { const int t0 = 2 * p_beg + (n >> 4); ti_c = tileinfo[t0 < T ? t0 : 2 * p_beg]; }
  for (int pr = p_beg; pr < p_end; ++pr) {
    asm volatile("" ::: "memory");                             // keep the LDS fragment reads inside the loop
    const int tile = 2 * pr + (n >> 4);
    const bool have_t = tile < T;
    const bool have = have_t && !(dbg & 1);                    // (`have` guards the row stores)
    const int4 ti = ti_c;                                      // (fetched one pair ahead: the loads below depend on it)
    if (pr + 1 < p_end) { const int tn = 2 * (pr + 1) + (n >> 4); ti_c = tileinfo[tn < T ? tn : 2 * (pr + 1)]; }
    const int ray = ti.x, j0 = ti.y, cnt = ti.z;
    const bool valid = have_t && s < cnt;
    const size_t ci = (size_t)ray * S + j0 + (valid ? s : 0);
    const int k = cidx[ci];
    const size_t trow = (size_t)tile * (size_t)(16 * GRD_LD);   // this lane's 16-row tile of the gradient rows
    const uint32_t* rb = relu_bits + (size_t)tile * 128 + s;
    uint32_t m2a = 0, m2b = 0, m1a = 0, m1b = 0;
    float go[3] = {0.0f, 0.0f, 0.0f};
    if (have_t) {
      m1a = rb[16 * h]; m1b = rb[16 * (2 + h)];               // layer 1: lane groups h, 2 + h of the 16-sample layout
      m2a = rb[64 + 16 * (2 * h)]; m2b = rb[64 + 16 * (2 * h + 1)];   // layer 2: lane groups 2 h, 2 h + 1
    }
    const float* rp = rays + (size_t)ray * 6;
    const float dn = sqrtf(rp[3] * rp[3] + rp[4] * rp[4] + rp[5] * rp[5]);
    const float dh[3] = {rp[3] / dn, rp[4] / dn, rp[5] / dn};
    if (valid) {                                               // d(loss)/d(pre-sigmoid colour): rgb_map = sum_k w_k rgb_k (tensorBase.py:632-633)
      const float w = cw[ci];
      const int oray = f.perm ? f.perm[ray] : ray;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float r = crgb[ci * 3 + c];
        go[c] = g_rgb[(size_t)oray * 3 + c] * w * r * (1.0f - r);
      }
    }
    if (have) {                                                // go block: lane group 0 = go, lane group 1 = (dhat, 1) (k_wgrad_w2w3)
      float* gp = grd + trow + 16 * GRD_GO + ((16 * h + s) << 2);
      row_store_plain(gp, h == 0 ? make_float4(go[0], go[1], go[2], 0.0f) : make_float4(dh[0], dh[1], dh[2], 1.0f));
      if (h == 0) rowinfo[(size_t)tile * 16 + s] = valid ? (uint32_t)((size_t)ray * S + k) : 0xffffffffu;
    }
    __builtin_amdgcn_iglp_opt(0);
    // ---- dz1 = (W2^T dz2) * [h1 > 0], dz2 = (W3[:, :128]^T go) * [h2 > 0] formed K-step by K-step on the VALU
    f32x16 d1[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) d1[m][r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {                            // K slots j = 4 q .. 4 q + 3: units 16 ks + 8 h + j
        const int u = 16 * ks + 8 * h + 4 * q;
        const float4 w0 = *reinterpret_cast<const float4*>(&tail[W32T_T_W3 + u]);
        const float4 w1 = *reinterpret_cast<const float4*>(&tail[W32T_T_W3 + W32T_T_W3_LD + u]);
        const float4 w2 = *reinterpret_cast<const float4*>(&tail[W32T_T_W3 + 2 * W32T_T_W3_LD + u]);
        const uint32_t mb = q ? m2b : m2a;
        v[4 * q]     = relu_gate(w0.x * go[0] + w1.x * go[1] + w2.x * go[2], mb, 4 * ks);
        v[4 * q + 1] = relu_gate(w0.y * go[0] + w1.y * go[1] + w2.y * go[2], mb, 4 * ks + 1);
        v[4 * q + 2] = relu_gate(w0.z * go[0] + w1.z * go[1] + w2.z * go[2], mb, 4 * ks + 2);
        v[4 * q + 3] = relu_gate(w0.w * go[0] + w1.w * go[1] + w2.w * go[2], mb, 4 * ks + 3);
      }
      bf16x8 bh, bl;
      split8c(v, bh, bl);
      if (!(dbg & 4)) {                                        // two M-tiles at a time: 16 fragment registers live instead of 32
        mma3_step<2>(img, W32T_W2 + ks, 8, lane, bh, bl, d1);
        mma3_step<2>(img, W32T_W2 + 16 + ks, 8, lane, bh, bl, d1 + 2);
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {                            // registers 4 q .. 4 q + 3: units 32 m + 8 q + 4 h + r
        const uint32_t mb = (q & 1) ? m1b : m1a;
        const int b0 = 4 * (2 * m + (q >> 1));
#pragma unroll
        for (int r = 0; r < 4; ++r) d1[m][4 * q + r] = relu_gate(d1[m][4 * q + r], mb, b0 + r);
      }
    // ---- feat^T: this lane's sample as a row of A (features 16 ks + 8 h + j), selector as B -> lane = feature, registers = samples
    bf16x8 Fh[2], Fl[2];
    {
      const float* fp = act + (size_t)(have_t ? tile : 2 * pr) * (size_t)(16 * ACT_LD) + 16 * ACT_FEAT + (s << 2);
      f32x16 ft;
#pragma unroll
      for (int r = 0; r < 16; ++r) ft[r] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const float4 a = row_load4(fp + ks * 256 + ((2 * h) * 16 << 2)), b = row_load4(fp + ks * 256 + ((2 * h + 1) * 16 << 2));
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        bf16x8 fh, fl;
        split8c(v, fh, fl);
        const bf16x8 sel = selector(ks);
        ft = mfma32(fh, sel, ft);
        ft = mfma32(fl, sel, ft);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ft[8 * q + j];
        split8c(v, Fh[q], Fl[q]);
      }
    }
    // ---- dfeat = W1^T dz1
    f32x16 df;
#pragma unroll
    for (int r = 0; r < 16; ++r) df[r] = 0.0f;
    {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        f32x16 dzt;                                            // dz1^T of M-tile m: lane = unit 32 m + n, registers = samples
#pragma unroll
        for (int r = 0; r < 16; ++r) dzt[r] = 0.0f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = d1[m][8 * q + j];
          bf16x8 bh, bl;
          split8c(v, bh, bl);
          const bf16x8 ah = w32_frag(img, W32T_W1 + 2 * m + q, 0, lane), al = w32_frag(img, W32T_W1 + 2 * m + q, 1, lane);
          const bf16x8 sel = selector(2 + q);
          df = mfma32(al, bh, df);
          dzt = mfma32(bh, sel, dzt);                          // the same split halves as A: transposed by the selector
          df = mfma32(ah, bl, df);
          dzt = mfma32(bl, sel, dzt);
          df = mfma32(ah, bh, df);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {                          // dW1[32 m + ..][f] += sum over samples 8 (2 q + (j >> 2)) + 4 h + (j & 3)
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = dzt[8 * q + j];
          bf16x8 th, tl;
          split8c(v, th, tl);
          w1acc[m] = mfma32(tl, Fh[q], w1acc[m]);
          w1acc[m] = mfma32(th, Fl[q], w1acc[m]);
          w1acc[m] = mfma32(th, Fh[q], w1acc[m]);
        }
      }
    }
    if (have) {                                                // dfeat block: register r = feature 8 (r >> 2) + 4 h + (r & 3) of sample n
#pragma unroll
      for (int q = 0; q < 4; ++q)
        row_store_plain(grd + trow + (GRD_DFEAT / 16 + (q >> 1)) * 256 + (((2 * (q & 1) + h) * 16 + s) << 2),
                        make_float4(df[4 * q], df[4 * q + 1], df[4 * q + 2], df[4 * q + 3]));
    }
  }
  // ---- dW1 partial of the workgroup: the waves' accumulators meet in LDS (the image is no longer needed), upper half of the
  // remaining waves into the lower half, in a fixed order; wave 0 writes the block
  f32x4* s_red = reinterpret_cast<f32x4*>(s_dyn3);           // [wave slot][16 float4 of the 64 accumulator registers][lane]
  for (int half = NW / 2; half >= 1; half >>= 1) {
    __syncthreads();
    if (wave >= half && wave < 2 * half) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          s_red[((wave - half) * 16 + 4 * m + q) * 64 + lane] = f32x4{w1acc[m][4 * q], w1acc[m][4 * q + 1], w1acc[m][4 * q + 2], w1acc[m][4 * q + 3]};
    }
    __syncthreads();
    if (wave < half) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 o4 = s_red[(wave * 16 + 4 * m + q) * 64 + lane];
#pragma unroll
          for (int r = 0; r < 4; ++r) w1acc[m][4 * q + r] += o4[r];
        }
    }
  }
  if (wave == 0) {
    float* out = wpart + (size_t)blockIdx.x * WP_FLOATS + WP_W1;           // [128 units][32 features (27 = the bias column)]
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[(32 * m + 8 * (r >> 2) + 4 * h + (r & 3)) * 32 + n] = w1acc[m][r];
  }
}

// ---- the appearance half of the colour network's backward: dfeat -> dX, d/d(position), dbasis.
// Per pair of 16-row tiles (lane (n, h) as above): the dfeat block comes back exactly as k_train_dgrad3's lanes stored it
// (D registers of the dfeat product), dX = basis^T dfeat lands per lane on the 36 channels its half gathers, the taps are
// gathered once for both the position gradient and X = plane x line, and
//   dbasis[f][channel] += sum over the 32 samples of dfeat[sample][f] X[sample][channel]
// runs on the matrix pipe with both operands transposed by selector products: dfeat^T (2 K-steps x hi / lo = 4 MFMAs;
// lane = feature, registers = samples) and, per plane, X^T of its 24 channels (column c < 24 of N-tile p = channel 24 p + c;
// the lane's values 12 p .. 12 p + 11 sit in two of its five K-steps: 4 MFMAs), then 2 K-steps x 3 terms per plane.
// 48 accumulator registers per wave, one [32][96] block per workgroup at WP_BAS (k_wgrad_reduce).
constexpr int APP3_IMG_U4 = (W32T_NFRAG - W32T_BAS) * 128;       // the six basis^T fragments
constexpr int APP3_STG_FLOATS = 32 * (GRD_LD - GRD_DX);          // dX rows of a pair of tiles, per wave
template <int NW, bool DXG = false /* order of the dX block (lrf_common.h): true = [plane][channel group][row][8] for k_scatter_fix<24>, false = row-major */>
__global__ __launch_bounds__(NW * 64) void k_train_app3(
    DField f, const uint4* __restrict__ imt, const float* __restrict__ rays, const float* __restrict__ z, int S,
    const int* __restrict__ toff, int R, const int4* __restrict__ tileinfo, const uint16_t* __restrict__ cidx,
    float* __restrict__ grd /* in: dfeat blocks, out: dX blocks */, float* __restrict__ rpart, int pmax, float* __restrict__ wpart,
    BinGeom bg, uint16_t* __restrict__ tile_id /* [3][nmax] plane-tile id of every row, 0xffff = none */, int* __restrict__ hist, uint32_t nmax,
    int dbg /* timing experiments: 1 no row stores, 2 no position gradient / X */,
    unsigned* __restrict__ vmax_bits /* max |dX line|, |dX plane| over the batch as float bits (atomicMax), or null */) {
  constexpr bool dx_groups = DXG;
  constexpr int NT = NW * 64;
  float vmax = 0.0f;
  extern __shared__ uint4 s_dyn4[];
  uint4* img = s_dyn4;                                        // fragment q (0..5) = W32T_BAS + q of the transposed image
  uint4* s_sel = s_dyn4 + APP3_IMG_U4;                        // the eight 0 / 1 selectors, [selector][lane]
  float* s_stg = reinterpret_cast<float*>(s_sel + 8 * 64) + (size_t)(threadIdx.x >> 6) * APP3_STG_FLOATS;   // this wave's dX staging tile pair
  float* s_z = reinterpret_cast<float*>(s_sel + 8 * 64) + (size_t)NW * APP3_STG_FLOATS;
  int* s_h = reinterpret_cast<int*>(s_z + S);                  // histogram of this workgroup's rows over the plane tiles (first pass of the appearance scatter's counting sort)
  for (int i = threadIdx.x; i < bg.total; i += NT) s_h[i] = 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, h = lane >> 5, s = n & 15;
  for (int i = tid; i < APP3_IMG_U4; i += NT) img[i] = imt[W32T_BAS * 128 + i];
  for (int i = tid; i < S; i += NT) s_z[i] = z[i];
  // 0 / 1 selectors (B operands of the transposing products, see k_train_dgrad3): slot j of lane (n, h) is 1 where the K index
  // of the slot is the lane's column.  0, 1: dfeat^T, K index = feature 16 q + 8 (j >> 2) + 4 h + (j & 3) (dfeat's D-register
  // order); 2 + 2 p + e: X^T of plane p, K-step KS0[p] + e of the lane's 40 values: value 8 ks + j is channel 12 h + (8 ks + j - 12 p)
  // of the plane when that is in 0..11.  Built once per workgroup, read back with one ds_read_b128 each per pair of tiles.
  for (int i = tid; i < 8 * 64; i += NT) {
    const int id = i >> 6, ln = i & 63, nn = ln & 31, hh = ln >> 5;
    uint32_t wds[4];
#pragma unroll
    for (int q2 = 0; q2 < 4; ++q2) {
      uint32_t wd = 0;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = 2 * q2 + e;
        int key;
        if (id < 2) key = 16 * id + 8 * (j >> 2) + 4 * hh + (j & 3);
        else {
          const int pp = (id - 2) >> 1, ks = (pp == 0 ? 0 : pp == 1 ? 1 : 3) + ((id - 2) & 1), c = 8 * ks + j - 12 * pp;
          key = (c >= 0 && c < 12) ? 12 * hh + c : -1;
        }
        if (key == nn) wd |= e ? 0x3f800000u : 0x3f80u;
      }
      wds[q2] = wd;
    }
    s_sel[i] = make_uint4(wds[0], wds[1], wds[2], wds[3]);
  }
  __syncthreads();
  const int T = toff[R];
  const int P = (T + 1) >> 1;                                 // pairs of 16-row tiles
  const int nb = gridDim.x;
  const int lb = (nb % 8 == 0) ? (int)(blockIdx.x % 8) * (nb / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;   // XCD-aware order
  const long long wid = (long long)lb * NW + wave, waves = (long long)nb * NW;
  const int p_beg = (int)(wid * P / waves), p_end = (int)((wid + 1) * P / waves);
  f32x16 bacc[3];                                              // dbasis partial: [f = 8 (r >> 2) + 4 h + (r & 3)][channel 24 p + n], n < 24
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) bacc[p][r] = 0.0f;
  auto selector = [&](int id) { return __builtin_bit_cast(bf16x8, s_sel[id * 64 + lane]); };
  // The header of a pair (tile -> ray, first sample, count -> this lane's sample index) is a chain of dependent loads in
  // front of the gathers: it is fetched one pair ahead (the tile record at the top of the previous pair, the index behind
  // its matrix products), as k_shade3 does with its tile header.
  auto tile_of = [&](int pr) { const int t = 2 * pr + (n >> 4); return t < T ? t : 2 * pr; };
  auto index_of = [&](int pr, const int4& ti) {
    const bool ok = 2 * pr + (n >> 4) < T && s < ti.z;
    return cidx[(size_t)ti.x * S + ti.y + (ok ? s : 0)];
  };
  int4 ti_c = make_int4(0, 0, 0, 0);
  int k_c = 0;
  if (p_beg < p_end) { ti_c = tileinfo[tile_of(p_beg)]; k_c = index_of(p_beg, ti_c); }
  for (int pr = p_beg; pr < p_end; ++pr) {
    asm volatile("" ::: "memory");                             // keep the LDS fragment reads inside the loop
    const int tile = 2 * pr + (n >> 4);
    const bool have_t = tile < T;
    const bool have = have_t && !(dbg & 1);
    const int4 ti = ti_c;
    const int ray = ti.x, cnt = ti.z;
    const bool valid = have_t && s < cnt;
    const int k = k_c;
    const bool more = pr + 1 < p_end;
    int4 ti_n = ti;
    if (more) ti_n = tileinfo[tile_of(pr + 1)];
    const size_t trow = (size_t)tile * (size_t)(16 * GRD_LD);
    // ---- dfeat of this lane's sample: register r = feature 8 (r >> 2) + 4 h + (r & 3) (rows beyond the tile's count are zero: go = 0)
    f32x16 df;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (have_t) v = *reinterpret_cast<const float4*>(grd + trow + (GRD_DFEAT / 16 + (q >> 1)) * 256 + (((2 * (q & 1) + h) * 16 + s) << 2));
      df[4 * q] = v.x; df[4 * q + 1] = v.y; df[4 * q + 2] = v.z; df[4 * q + 3] = v.w;
    }
    const float* rp = rays + (size_t)ray * 6;
    const float o[3] = {rp[0], rp[1], rp[2]};
    const float dn = sqrtf(rp[3] * rp[3] + rp[4] * rp[4] + rp[5] * rp[5]);
    const float dh[3] = {rp[3] / dn, rp[4] / dn, rp[5] / dn};
    __builtin_amdgcn_iglp_opt(0);
    // ---- dX = basis^T dfeat: register r of tile mt = value 16 mt + r of this lane half = channel w32_chan(h, 16 mt + r);
    //      dfeat^T through the selector (the same split halves as A)
    bf16x8 Ah[2], Al[2];
    float dX[36];
    {
      f32x16 dx[3], dft;
#pragma unroll
      for (int r = 0; r < 16; ++r) { dx[0][r] = 0.0f; dx[1][r] = 0.0f; dx[2][r] = 0.0f; dft[r] = 0.0f; }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = df[8 * q + j];
        bf16x8 bh, bl;
        split8c(v, bh, bl);
        mma3_step<3>(img, q, 2, lane, bh, bl, dx);
        const bf16x8 sel = selector(q);
        dft = mfma32(bh, sel, dft);
        dft = mfma32(bl, sel, dft);
      }
#pragma unroll
      for (int vv = 0; vv < 36; ++vv) dX[vv] = dx[vv >> 4][vv & 15];
#pragma unroll
      for (int q = 0; q < 2; ++q) {                            // A operand of the dbasis product: lane = feature, slots = samples
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = dft[8 * q + j];
        split8c(v, Ah[q], Al[q]);
      }
    }
    int k_n = k;
    if (more) k_n = index_of(pr + 1, ti_n);
    if (!(dbg & 1)) {
      // dX block of the two tiles, in the order this pass's appearance scatter reads it (lrf_common.h: rows, or [plane][channel
      // group of 8][row][8]).  A lane holds 3 x 48 B of its row: stored directly that is nine instructions of 64 scattered
      // 16-byte pieces (65 of the kernel's 160 us); staged through this wave's LDS tile in the global order instead, every
      // store instruction writes 1 KB contiguous.
      float* tile_s = s_stg + (n >> 4) * (16 * (GRD_LD - GRD_DX));
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int c = 12 * h + 4 * i;                        // the float4's first channel: 0, 4, .. 20 -- inside one group of 8
          float* d = dx_groups ? tile_s + ((((p * 3 + (c >> 3)) * 16 + (n & 15)) << 3) + (c & 7)) : tile_s + (n & 15) * (GRD_LD - GRD_DX) + p * LRF_CA + c;
          *reinterpret_cast<float4*>(d) = make_float4(dX[12 * p + 4 * i], dX[12 * p + 4 * i + 1], dX[12 * p + 4 * i + 2], dX[12 * p + 4 * i + 3]);
        }
      *reinterpret_cast<float4*>(dx_groups ? tile_s + 9 * 128 + ((n & 15) << 3) + 4 * h : tile_s + (n & 15) * (GRD_LD - GRD_DX) + 72 + 4 * h) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // the pad columns
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int th = 0; th < 2; ++th) {
        if (2 * pr + th < T) {
          float* gdx = grd + (size_t)(2 * pr + th) * (size_t)(16 * GRD_LD) + GRD_DX * 16;
#pragma unroll
          for (int i = 0; i < 5; ++i)
            row_store_plain(gdx + (i * 64 + lane) * 4, *reinterpret_cast<const float4*>(s_stg + th * (16 * (GRD_LD - GRD_DX)) + (i * 64 + lane) * 4));
        }
      }
      __builtin_amdgcn_wave_barrier();                         // (the next pair's staging writes stay behind these reads)
    }
    // ---- d/d(position) from the appearance lookups, and X again
    const float zk = s_z[k];
    float xr[3] = {o[0] + dh[0] * zk, o[1] + dh[1] * zk, o[2] + dh[2] * zk};
    float xc[3] = {xr[0], xr[1], xr[2]};
    contract3(xc[0], xc[1], xc[2]);
    float u[3], gu[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int a = 0; a < 3; ++a) u[a] = (xc[a] - f.lo[a]) * f.inv[a] - 1.0f;
    float X[40];
#pragma unroll
    for (int i = 0; i < 40; ++i) X[i] = 0.0f;
    int i0[3], i1[3]; float t[3], gm[3];
    tap1d_g(u[0], f.pw[0], i0[0], i1[0], t[0], gm[0]);         // grid[a] = pw[0], ph[0], ll[0] for a = 0, 1, 2 (axis_taps)
    tap1d_g(u[1], f.ph[0], i0[1], i1[1], t[1], gm[1]);
    tap1d_g(u[2], f.ll[0], i0[2], i1[2], t[2], gm[2]);
    if (have_t && h == 0) {                                    // the row's tile in each plane, counted for the binned scatter
      const uint32_t row = (uint32_t)tile * 16u + (uint32_t)s;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const int tt = ((unsigned)i0[MAT1[p]] / BTILE) * bg.tx[p] + (unsigned)i0[MAT0[p]] / BTILE;
        tile_id[(uint32_t)p * nmax + row] = valid ? (uint16_t)tt : (uint16_t)0xffff;
        if (valid) atomicAdd(&s_h[bg.base[p] + tt], 1);
      }
    }
    if (!(dbg & 2)) {                                          // every lane gathers (rows beyond the tile's count sit on the tile's first sample and carry dX = dfeat = 0):
      app12_position_grad<0>(f, i0, i1, t, gm, h, dX, gu, X, vmax);  // no branch between the matrix products and the gathers, the compiler interleaves them
      app12_position_grad<1>(f, i0, i1, t, gm, h, dX + 12, gu, X + 12, vmax);
      app12_position_grad<2>(f, i0, i1, t, gm, h, dX + 24, gu, X + 24, vmax);
    }
    // ---- dbasis += dfeat^T X, plane by plane
    {
      bf16x8 xh[5], xl[5];
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) split8c(X + 8 * ks, xh[ks], xl[ks]);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        constexpr int KS0[3] = {0, 1, 3};                      // values 12 p .. 12 p + 11 of a lane half live in K-steps KS0[p], KS0[p] + 1
        f32x16 xt;                                             // X^T of plane p: lane = channel n (< 24), registers = samples
#pragma unroll
        for (int r = 0; r < 16; ++r) xt[r] = 0.0f;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int ks = KS0[p] + e;
          const bf16x8 sel = selector(2 + 2 * p + e);
          xt = mfma32(xh[ks], sel, xt);
          xt = mfma32(xl[ks], sel, xt);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = xt[8 * q + j];
          bf16x8 th, tl;
          split8c(v, th, tl);
          bacc[p] = mfma32(Al[q], th, bacc[p]);
          bacc[p] = mfma32(Ah[q], tl, bacc[p]);
          bacc[p] = mfma32(Ah[q], th, bacc[p]);
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) gu[a] += __shfl_xor(gu[a], 32, 64);      // the two channel halves of a sample
    float gx3[3] = {gu[0] * f.inv[0], gu[1] * f.inv[1], gu[2] * f.inv[2]};
    contract3_bwd(xr, gx3);
    float prt[6] = {gx3[0], gx3[1], gx3[2], gx3[0] * zk, gx3[1] * zk, gx3[2] * zk};
    if (!valid) { prt[0] = prt[1] = prt[2] = prt[3] = prt[4] = prt[5] = 0.0f; }
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int dd = 1; dd < 16; dd <<= 1) prt[q] += __shfl_xor(prt[q], dd, 64);       // over the 16 samples of the lane's tile
    if (have_t && s == 0 && h == 0) {
      float* rpp = rpart + ((size_t)ray * pmax + ti.w) * 8;
#pragma unroll
      for (int q = 0; q < 6; ++q) rpp[q] = prt[q];
    }
    ti_c = ti_n; k_c = k_n;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < bg.total; i += NT) {
    const int v = s_h[i];
    if (v) atomicAdd(&hist[i], v);
  }
  if (vmax_bits) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, d, 64));
    if ((threadIdx.x & 63) == 0 && __float_as_uint(vmax) > __hip_atomic_load(vmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))   // (the maximum only grows: k_bwd_ray)
      atomicMax(vmax_bits, __float_as_uint(vmax));
  }
  // ---- dbasis partial of the workgroup (as k_train_dgrad3's dW1 block)
  f32x4* s_red = reinterpret_cast<f32x4*>(s_dyn4);           // [wave slot][12 float4 of the 48 accumulator registers][lane]
  for (int half = NW / 2; half >= 1; half >>= 1) {
    __syncthreads();
    if (wave >= half && wave < 2 * half) {
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          s_red[((wave - half) * 12 + 4 * p + q) * 64 + lane] = f32x4{bacc[p][4 * q], bacc[p][4 * q + 1], bacc[p][4 * q + 2], bacc[p][4 * q + 3]};
    }
    __syncthreads();
    if (wave < half) {
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 o4 = s_red[(wave * 12 + 4 * p + q) * 64 + lane];
#pragma unroll
          for (int r = 0; r < 4; ++r) bacc[p][4 * q + r] += o4[r];
        }
    }
  }
  if (wave == 0) {
    float* out = wpart + (size_t)blockIdx.x * WP_FLOATS + WP_BAS;          // [32 features][3 planes x 32 columns (24 channels)]
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[(8 * (r >> 2) + 4 * h + (r & 3)) * 96 + 32 * p + n] = bacc[p][r];
  }
}
constexpr size_t app3_lds_bytes(int S, int NW, int nbins) {
  const size_t a = (size_t)(APP3_IMG_U4 + 8 * 64) * 16 + (size_t)NW * APP3_STG_FLOATS * 4 + (size_t)S * 4 + (size_t)nbins * 4, b = (size_t)(NW / 2) * 12 * 64 * 16;
  return a > b ? a : b;
}

// (Round 4 first built the ROW-SAVING FORWARD on this skeleton -- k_shade3's gather and chain on a static split of tile pairs --
// and measured it: 318-338 us against 175-179 us for the 16-sample kernel of rounds 1-3.  Without k_shade3's tile queue and
// one-tile-ahead header prefetch the dependent chain toff -> ncomp -> cidx -> z -> gathers is exposed at two waves per SIMD,
// and the extra per-lane state pushed the chain into 95 spilled registers; profiles/r11 s7.  What ships is the eval kernel
// itself with a SAVE switch, k_shade3<SAVE>: 125 us.)
