// lrf_shade3_kernel.inl -- the colour kernel of lrf_shade3.inl, included twice (k_shade3: one field; k_shade3m: several fields)
// NW waves per workgroup (one workgroup per CU).  LDSTOFF: the tile offsets are scanned by every workgroup itself into
// LDS (R + 1 ints beside the image: two launches per render, k_march -> k_shade3); otherwise k_scan_tiles_n<32> ran before
// and toff_g holds them.
// TIMED (debug, lrf_debug_set_dump): s_memtime totals per wave -> dump[block][wave][8] =
// {rest of the prologue, header + position, gather + split, image copy, scan, chain, tiles, finalize}
// SAVE: the training forward -- the same tile loop additionally writes what SaveOut3 lists (128 B of feat + 32 B of mask bits
// + 12 B of colour per shaded sample): the eval kernel IS the row-saving forward, its rgb is bit-identical to the eval's.
// LRF_SHADE3_MULTI (this file is included twice, lrf_shade3.inl): the second kernel, k_shade3m, renders several fields in one
// launch (lrf_scene_fwd; MultiF; !LDSTOFF, !SAVE): R counts the virtual rays of all fields, field-major; a workgroup's ray range is
// cut at the field boundaries into segments and each segment runs the tile loop with its field's weight image in LDS (three of
// the 256 ranges of a 4-field batch span two fields).  The preprocessor, not a template parameter, keeps k_shade3 itself
// token for token what it was: its register allocation is part of the measured headline.
template <int NW, bool LDSTOFF, bool TIMED = false, bool SAVE = false>
__global__ __launch_bounds__(NW * 64) void LRF_SHADE3_NAME(
#if LRF_SHADE3_MULTI
    MultiF mf,
#else
    DField f,
#endif
    const float* __restrict__ rays, const float* __restrict__ z, int S,
    const int* __restrict__ toff_g, int R, const int* __restrict__ ncomp, const uint16_t* __restrict__ cidx,
    const float* __restrict__ cw, float* __restrict__ part, int pmax,
    uint32_t flags, const float* __restrict__ acc, float* __restrict__ rgb_out, float* __restrict__ acc_out, SaveOut3 sv) {
  constexpr int NT = NW * 64;
  extern __shared__ uint4 s_dyn[];                             // image, tail, z[S][, toff[R + 1]]
  unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#define LRF_TICK(i) do { if (TIMED) { const unsigned long long now_ = __builtin_readcyclecounter(); tk[i] += now_ - tlast; tlast = now_; } } while (0)
  if (TIMED) tlast = __builtin_readcyclecounter();
  uint4* img = s_dyn;
  float* tail = reinterpret_cast<float*>(s_dyn + W32_U4);
  float* s_z = tail + W32_T_FLOATS;
  lds_int* s_toff = (lds_int*)(s_z + S);
  typedef __attribute__((address_space(3))) unsigned short lds_u16;
  lds_u16* s_nc = (lds_u16*)(s_toff + (LDSTOFF ? R + 1 : 0));  // per-ray shaded-sample counts beside the offsets (LDSTOFF)
  __shared__ int s_wave[NW];
  __shared__ int s_next;                                       // tile queue of this workgroup

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, h = lane >> 5;
#if LRF_SHADE3_MULTI
  static_assert(!LDSTOFF && !SAVE && !TIMED, "the multi-field kernel reads global tile offsets and saves nothing");
  const DField& f = mf.f[0];                                   // (the tile loop below names its segment's field `f` again)
#else
  {
    // Every workgroup starts its copy of the 95 KB image at a different place.  Started at the same place, the CUs of an
    // XCD ask the same L2 channel for the same line at the same time and the copy runs at ~11 B / cycle / CU (8.2 K
    // cycles); rotated it takes 5.0 K (colour stage 120.6 -> 119.4 us, scripts/ab_shade.sh).  Rotating the reads of the
    // per-ray counts and of k_march's line staging the same way gains nothing measurable.
    const int rot = (int)((blockIdx.x * 37u) % 93u) * 64;
    for (int i = tid; i < W32_ALL_U4; i += NT) { int j = i + rot; if (j >= W32_ALL_U4) j -= W32_ALL_U4; img[j] = f.mlpw[j]; }
  }
#endif
  for (int i = tid; i < S; i += NT) s_z[i] = z[i];
  if (TIMED) { __syncthreads(); LRF_TICK(3); }                 // (TIMED only: image + z in LDS)
  if (LDSTOFF) {                                               // exclusive scan of ceil(ncomp / 32): eight rays per thread and round
    int carry = 0;
    for (int base = 0; base < R; base += NT * 8) {
      const int r0 = base + tid * 8;
      int v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int nc = r0 + i < R ? ncomp[r0 + i] : 0;
        if (r0 + i < R) s_nc[r0 + i] = (unsigned short)nc;
        v[i] = (nc + ITEM3 - 1) / ITEM3;
      }
#pragma unroll
      for (int i = 1; i < 8; ++i) v[i] += v[i - 1];
      int incl = v[7];
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
      }
      __syncthreads();                                         // s_wave of the previous round has been read
      if (lane == 63) s_wave[wave] = incl;
      __syncthreads();
      int woff = 0, tot = 0;
#pragma unroll
      for (int q = 0; q < NW; ++q) { const int x = s_wave[q]; woff += q < wave ? x : 0; tot += x; }
      const int excl = carry + woff + incl - v[7];
      if (r0 < R) s_toff[r0] = excl;
#pragma unroll
      for (int i = 1; i < 8; ++i) if (r0 + i < R) s_toff[r0 + i] = excl + v[i - 1];
      carry += tot;
    }
    if (tid == 0) s_toff[R] = carry;
  }
  __syncthreads();
  LRF_TICK(4);                                                 // (TIMED only: scan done)
  typedef typename std::conditional<LDSTOFF, const lds_int*, const int*>::type ToffP;
  ToffP toff;
  if constexpr (LDSTOFF) toff = s_toff; else toff = toff_g;
  if constexpr (SAVE) {
    if (blockIdx.x == 0) for (int r = tid; r <= R; r += NT) sv.toff16[r] = 2 * (int)toff[r];
  }

  // this workgroup's rays [ra, rb) and tiles [T0, T1): the even split of the tile list, moved to ray boundaries
  const int nb = gridDim.x;
  const int lb = (nb % 8 == 0) ? (int)(blockIdx.x % 8) * (nb / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;   // XCD-aware order
  const int T = toff[R];
  const int ra = __builtin_amdgcn_readfirstlane(toff_lower_bound_p(toff, R, (int)((long long)lb * T / nb)));
  const int rb = __builtin_amdgcn_readfirstlane(lb == nb - 1 ? R : toff_lower_bound_p(toff, R, (int)((long long)(lb + 1) * T / nb)));
  const int T0 = __builtin_amdgcn_readfirstlane(toff[ra]), T1 = __builtin_amdgcn_readfirstlane(toff[rb]);
#if LRF_SHADE3_MULTI
  for (int fk = ra / mf.Rf; fk < mf.nf; ++fk) {                // the segments of this workgroup's range, one field each
    const int ra_k = max(ra, fk * mf.Rf), rb_k = min(rb, (fk + 1) * mf.Rf);
    if (ra_k >= rb_k) break;
    const int T0_k = __builtin_amdgcn_readfirstlane(toff[ra_k]), T1_k = __builtin_amdgcn_readfirstlane(toff[rb_k]);
    __syncthreads();                                           // every wave is through with the previous segment's image
    {
      const int rot = (int)((blockIdx.x * 37u) % 93u) * 64;
      const uint4* src = mf.f[fk].mlpw;
      for (int i = tid; i < W32_ALL_U4; i += NT) { int j = i + rot; if (j >= W32_ALL_U4) j -= W32_ALL_U4; img[j] = src[j]; }
    }
    {                                                          // the names the tile loop uses, for this segment
    const DField& f = mf.f[fk];
    const int ra = ra_k, T0 = T0_k, T1 = T1_k;
#endif
  if (tid == 0) s_next = T0 + 2 * NW;                          // tiles T0 .. T0 + 2 NW - 1 are handed out statically below
  __syncthreads();

  // Tiles are pulled from the workgroup's queue (an LDS counter): a wave that gathers from warm lines moves on instead
  // of waiting for a slower neighbour.  Every wave sees its tiles in increasing order, so the ray of a tile is found by
  // walking forward from the previous one (state cached per ray).
  int w_ray = ra, w_next = ra < R ? (int)toff[ra + 1] : T1, w_tile0 = T0, w_nc = 0;
  bool w_fresh = true;
  auto issue_header = [&](int t) {
    while (w_next <= t) { ++w_ray; w_tile0 = w_next; w_next = toff[w_ray + 1]; w_fresh = true; }
    w_ray = __builtin_amdgcn_readfirstlane(w_ray);
    if (w_fresh) { w_nc = __builtin_amdgcn_readfirstlane(LDSTOFF ? (int)s_nc[w_ray] : ncomp[w_ray]); w_fresh = false; }
    Hdr3 hd;
    hd.ray = w_ray; hd.tile_in_ray = t - w_tile0;
    const int j0 = hd.tile_in_ray * ITEM3;
    const int cnt = min(ITEM3, w_nc - j0);
    hd.cnt = cnt;
    const size_t ci = (size_t)w_ray * S + j0 + (n < cnt ? n : 0);
    hd.k = cidx[ci];
    hd.wgt = (n < cnt && h == 0) ? cw[ci] : 0.0f;              // the two K halves of a sample hold the same colour: count it once
#if LRF_SHADE3_MULTI
    const float* rp = rays + (size_t)multi_io(mf, fk, w_ray) * 6;       // (a chunk of a larger batch: the caller's ray index)
#else
    const float* rp = rays + (size_t)w_ray * 6;
#endif
    const float4 dq = *reinterpret_cast<const float4*>(f.rdir + (size_t)w_ray * 4);     // d / |d| as k_march formed it (tensorBase.py:578-580)
    hd.o[0] = rp[0]; hd.o[1] = rp[1]; hd.o[2] = rp[2]; hd.d[0] = dq.x; hd.d[1] = dq.y; hd.d[2] = dq.z;
    return hd;
  };
  LRF_TICK(0);
  int t_cur = T0 + wave, t_nxt = T0 + NW + wave;               // the first two tiles of a wave are fixed: no queue latency at start
  Hdr3 cur;
  if (t_cur < T1) cur = issue_header(t_cur);
  while (t_cur < T1) {
    asm volatile("" ::: "memory");                             // keep the LDS fragment reads inside the loop
    int t_after = 0;                                           // the tile after next: its number is back long before it is needed
    if (lane == 0) t_after = atomicAdd(&s_next, 1);
    Hdr3 nxt = cur;
    if (t_nxt < T1) nxt = issue_header(t_nxt);
    // ------------------------------------------------------------------ gather
    bf16x8 xh[5], xl[5];
    float vb[3];
    {
      const float dh[3] = {cur.d[0], cur.d[1], cur.d[2]};
#pragma unroll
      for (int c = 0; c < 3; ++c) {                            // view-direction part of mlp_view.0 + bias (tensorBase.py:131-132; viewdirs detached :628)
        const float4 wv = *reinterpret_cast<const float4*>(&tail[W32_T_W3 + W32_T_W3_LD * c + LRF_FEATC]);
        vb[c] = tail[W32_T_B3 + c] + wv.x * dh[0] + wv.y * dh[1] + wv.z * dh[2];
      }
      float x[3], u[3];
      sample_point(f, cur.o, dh, s_z[cur.k], x, u);
      const AxisTaps at = axis_taps(f.pw[0], f.ph[0], f.ll[0], u);
      if (TIMED) { asm volatile("" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2])); LRF_TICK(1); }
      float X[40];
      gather_app12<0>(f, at, h, X);
      gather_app12<1>(f, at, h, X + 12);
      gather_app12<2>(f, at, h, X + 24);
      X[36] = X[37] = X[38] = X[39] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) split8c(X + 8 * ks, xh[ks], xl[ks]);
    }
    if (TIMED) { asm volatile("" : "+v"(xh[0]), "+v"(xl[4])); LRF_TICK(2); }
    // ------------------------------------------------------------------ chain
    // While it multiplies, a wave outranks its SIMD partner in the issue arbitration (the partner mostly waits for
    // gathers): 123.3 -> 120.1 us (interleaved A/B, profiles/r08c).  Requesting every A fragment one K-step ahead by
    // hand (sched_barrier regions) was measured too: 137 vs 133 us, the compiler's own order is better.  Starting the
    // second wave of every SIMD 8 K / 16 K / 32 K cycles late changes nothing (123.4 / 123.1 / 124.7 / 127.7 us): the waves
    // are not phase-locked; a wave issues one instruction per 4-cycle slot and the loop body is 1882 of them (1265 VALU,
    // 135 MFMA, 196 LDS, 58 VMEM, 228 SALU): instruction count is what is left to cut.  Also measured and dropped: layer 2
    // with double-buffered A fragments and sched_group_barrier(DS_READ, MFMA) pinning (125.0 vs 122.2 us), the scheduler
    // strategies max-ilp (131.5) and max-memory-clause (133.8); the weight image copied by direct global -> LDS loads
    // under the first tile's gathers (prologue 20.8 K -> 15.8 K cycles, but every arrangement of the loop that allows it
    // costs the chain 0.9-3.3 K cycles per tile in the compiler's schedule: 122.6-127.5 us against 121.3-123.9).
    __builtin_amdgcn_iglp_opt(0);                             // DS-read / MFMA interleave of the small-GEMM heuristic: 122.7 -> 121.2 us (scripts/ab_shade.sh)
    __builtin_amdgcn_s_setprio(2);
    // basis 72 -> 27 (tensoRF.py:196): five K-steps; the three terms in three accumulators (one output tile only)
    f32x16 fa, fb, fc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { fa[r] = 0.0f; fb[r] = 0.0f; fc[r] = 0.0f; }
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      const bf16x8 ah = w32_frag(img, W32_BAS + ks, 0, lane), al = w32_frag(img, W32_BAS + ks, 1, lane);
      fa = mfma32(al, xh[ks], fa);
      fb = mfma32(ah, xl[ks], fb);
      fc = mfma32(ah, xh[ks], fc);
    }
    const f32x16 fe = (fa + fb) + fc;
    const size_t t16 = 2 * (size_t)t_cur + (size_t)(n >> 4);   // (SAVE) this lane's 16-row tile
    if constexpr (SAVE) {                                      // feat row: register r = feature 8 (r >> 2) + 4 h + (r & 3); column 27 = 1 (bias column of dW1), 28.. = 0
      float* ap = sv.act + t16 * (size_t)(16 * ACT_LD) + 16 * ACT_FEAT + ((n & 15) << 2);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 v4 = make_float4(fe[4 * q], fe[4 * q + 1], fe[4 * q + 2], fe[4 * q + 3]);
        if (q == 3) { if (h == 0) v4.w = 1.0f; else v4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
        *reinterpret_cast<float4*>(ap + (q >> 1) * 256 + (((2 * (q & 1) + h) * 16) << 2)) = v4;
      }
      if (lane == 0) {
        const int j0 = cur.tile_in_ray * ITEM3;
        sv.tileinfo[2 * (size_t)t_cur] = make_int4(cur.ray, j0, min(16, cur.cnt), 2 * cur.tile_in_ray);
        sv.tileinfo[2 * (size_t)t_cur + 1] = make_int4(cur.ray, cur.cnt > 16 ? j0 + 16 : j0, max(0, cur.cnt - 16), 2 * cur.tile_in_ray + 1);
      }
    }
    f32x16 h1[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bq = *reinterpret_cast<const float4*>(&tail[W32_T_B1 + 32 * m + 8 * q + 4 * h]);
        h1[m][4 * q] = bq.x; h1[m][4 * q + 1] = bq.y; h1[m][4 * q + 2] = bq.z; h1[m][4 * q + 3] = bq.w;
      }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fe[8 * q + j];
      bf16x8 bh, bl;
      split8c(v, bh, bl);
      mma3_step<4>(img, W32_W1 + q, 2, lane, bh, bl, h1);
    }
    bf16x8 b2h[8], b2l[8];
    // (SAVE) mask bits: register 4 q4 + r of M-tile m is unit 32 m + 8 q4 + 4 h + r = unit 16 t1 + 4 g + r of the 16-row
    // layout with t1 = 2 m + (q4 >> 1), g = 2 (q4 & 1) + h: dword q4 & 1 of this lane, bit 4 t1 + r
    uint32_t mk[2] = {0u, 0u};
#pragma unroll
    for (int m0 = 0; m0 < 4; ++m0)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[j] = relu_i(h1[m0][8 * q + j]);
          if constexpr (SAVE) mk[j >> 2] |= min(__float_as_uint(v[j]), 1u) << (4 * (2 * m0 + q) + (j & 3));
        }
        split8c(v, b2h[2 * m0 + q], b2l[2 * m0 + q]);
      }
    if constexpr (SAVE) {
      uint32_t* bp = sv.relu_bits + t16 * 128 + (n & 15) + 16 * h;
      bp[0] = mk[0]; bp[32] = mk[1];
      mk[0] = 0u; mk[1] = 0u;
    }
    float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x16 h2[2];
#pragma unroll
      for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 bq = *reinterpret_cast<const float4*>(&tail[W32_T_B2 + 32 * (2 * half + mm) + 8 * q + 4 * h]);
          h2[mm][4 * q] = bq.x; h2[mm][4 * q + 1] = bq.y; h2[mm][4 * q + 2] = bq.z; h2[mm][4 * q + 3] = bq.w;
        }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) mma3_step<2>(img, W32_W2 + 16 * half + ks, 8, lane, b2h[ks], b2l[ks], h2);
#pragma unroll
      for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int u = 32 * (2 * half + mm) + 8 * q + 4 * h;
          const float4 w0 = *reinterpret_cast<const float4*>(&tail[W32_T_W3 + u]);
          const float4 w1 = *reinterpret_cast<const float4*>(&tail[W32_T_W3 + W32_T_W3_LD + u]);
          const float4 w2 = *reinterpret_cast<const float4*>(&tail[W32_T_W3 + 2 * W32_T_W3_LD + u]);
          const float a0 = relu_i(h2[mm][4 * q]), a1 = relu_i(h2[mm][4 * q + 1]), a2 = relu_i(h2[mm][4 * q + 2]), a3 = relu_i(h2[mm][4 * q + 3]);
          if constexpr (SAVE) {
            const int b0 = 4 * (2 * (2 * half + mm) + (q >> 1));
            mk[q & 1] |= (min(__float_as_uint(a0), 1u) << b0) | (min(__float_as_uint(a1), 1u) << (b0 + 1))
                       | (min(__float_as_uint(a2), 1u) << (b0 + 2)) | (min(__float_as_uint(a3), 1u) << (b0 + 3));
          }
          o0 += a0 * w0.x; o0 += a1 * w0.y; o0 += a2 * w0.z; o0 += a3 * w0.w;
          o1 += a0 * w1.x; o1 += a1 * w1.y; o1 += a2 * w1.z; o1 += a3 * w1.w;
          o2 += a0 * w2.x; o2 += a1 * w2.y; o2 += a2 * w2.z; o2 += a3 * w2.w;
        }
    }
    __builtin_amdgcn_s_setprio(0);
    o0 += __shfl_xor(o0, 32, 64); o1 += __shfl_xor(o1, 32, 64); o2 += __shfl_xor(o2, 32, 64);
    // w * sigmoid(x) (:133, :632), hardware exp2 / reciprocal; partial colour of the tile = sum over its samples
    const float s0 = __frcp_rn(1.0f + __expf(-(o0 + vb[0]))), s1 = __frcp_rn(1.0f + __expf(-(o1 + vb[1]))), s2 = __frcp_rn(1.0f + __expf(-(o2 + vb[2])));
    if constexpr (SAVE) {
      uint32_t* bp = sv.relu_bits + t16 * 128 + 64 + (n & 15) + 16 * h;
      bp[0] = mk[0]; bp[32] = mk[1];
      if (h == 0 && n < cur.cnt) {
        float* cp = sv.crgb + ((size_t)cur.ray * S + cur.tile_in_ray * ITEM3 + n) * 3;
        cp[0] = s0; cp[1] = s1; cp[2] = s2;
      }
    }
    float cr = cur.wgt * s0;
    float cg = cur.wgt * s1;
    float cb = cur.wgt * s2;
#pragma unroll
    for (int dd = 1; dd < 32; dd <<= 1) {
      cr += __shfl_xor(cr, dd, 64); cg += __shfl_xor(cg, dd, 64); cb += __shfl_xor(cb, dd, 64);
    }
    if (lane == 0) {
      float* pp = part + ((size_t)cur.ray * pmax + cur.tile_in_ray) * 3;
      pp[0] = cr; pp[1] = cg; pp[2] = cb;
    }
    cur = nxt; t_cur = t_nxt; t_nxt = __builtin_amdgcn_readfirstlane(t_after);
    LRF_TICK(5);
    tk[6] += 1;
  }
#if LRF_SHADE3_MULTI
    }
  }
#endif
  // rgb_map = sum_k w_k rgb_k (+ 1 - acc) (tensorBase.py:632-634): this workgroup wrote every partial of its rays.  Its
  // waves share one L1 and these lines were never read before in this launch; the stores only have to be acknowledged.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#if LRF_SHADE3_MULTI
  for (int r = ra + tid; r < rb; r += NT) finalize_ray<false>(r, (int)toff[r + 1] - (int)toff[r], pmax, flags, acc, part, rgb_out, acc_out, multi_io(mf, min(r / mf.Rf, mf.nf - 1), r));
#else
  for (int r = ra + tid; r < rb; r += NT) finalize_ray<false>(r, (int)toff[r + 1] - (int)toff[r], pmax, flags, acc, part, rgb_out, acc_out, f.perm ? f.perm[r] : r);
#endif
  LRF_TICK(7);
  if (TIMED && f.dump && lane == 0) {
    unsigned long long* dp = reinterpret_cast<unsigned long long*>(f.dump) + ((size_t)blockIdx.x * NW + wave) * 8;
    for (int i = 0; i < 8; ++i) dp[i] = tk[i];
  }
#undef LRF_TICK
}

