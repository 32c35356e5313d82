// lrf_shade3.inl -- the colour stage on v_mfma_f32_32x32x16_bf16, 32 samples per wave (default engine since round 3).
//
// What k_shade2 (16 samples per wave on v_mfma_f32_16x16x32_bf16) taught (profiles/r07c_round_end.md): every 16-sample
// tile re-reads the 92 KB weight image from LDS, the three-term products ran as dependent MFMA triples with hand-placed
// wait states, and 16 waves per CU drifted into lockstep (all gathering, then all multiplying).  This kernel:
//   * 32 samples per wave: lane (n = lane & 31, h = lane >> 5) = (sample, K half).  One A fragment now feeds 32 columns:
//     half the LDS fragment traffic per sample, and the D registers of a layer are again the B operand of the next
//     (K permutation folded into the packed weights, lrf_common.h W32_*): no LDS round trip, no lane movement.
//   * compiler-scheduled builtins.  scripts/ubench/mfma_war.hip (profiles/r08a) shows the matrix pipe reads its sources at
//     issue: hipcc's hazard table is sufficient, the hand-issued chain of k_shade2 and its 299 s_nop per tile were not
//     needed.  The three terms of a split product go term-major over independent accumulators, so no MFMA waits for the
//     one issued just before it.
//   * 512-thread workgroups, one per CU: 8 waves, two per SIMD, up to 256 registers each; tiles come from a per-workgroup
//     queue, and each wave fetches the header of its next tile one tile ahead.  (Forcing the two waves of a SIMD into
//     opposite phases with s_barrier -- one gathers while the other multiplies -- was measured: 153 vs 139 us; 12 waves
//     with 168 registers: 156 us, the chain spills.  profiles/r08c.)
//   * dense 24-channel appearance texels (96 B): lane half h reads channels 12 h .. 12 h + 11 of a tap as three aligned
//     float4: 54 wave-level loads per 32 samples (k_shade2: 72), five basis K-steps of 16 instead of six.
//   * a workgroup owns WHOLE rays (its tile range is cut at ray boundaries): every ray's tile partials are summed, in
//     tile order, by the workgroup that wrote them -- no hand-off between workgroups, no counter, no cross-XCD visibility
//     protocol (the boundary rays of k_shade2<FUSE> needed one).
// Arithmetic per sample as k_shade2: split-bf16 three-term products, fp32 accumulate, VALU head, hardware exp2 / rcp.
#pragma once

namespace lrf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ITEM3 = 32;          // compact samples per work item of k_shade3 = one 32-column MFMA tile

__device__ __forceinline__ bf16x8 w32_frag(const uint4* img, int frag, int part, int lane) {
  return __builtin_bit_cast(bf16x8, img[(frag * 2 + part) * 64 + lane]);
}
// hi = bf16(v) (round to nearest even), lo = bf16(v - hi), two values at a time: one packed conversion gives both hi
// halves, a shift and a mask turn them back into floats, a packed subtract forms both remainders (2.5 VALU instructions per
// value; the element-wise form costs 4)
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8c(const float v[8], bf16x8& hi, bf16x8& lo) {
  uint32_t H[4], L[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2v ab = {v[2 * j], v[2 * j + 1]};
    const uint32_t p = __builtin_bit_cast(uint32_t, __builtin_convertvector(ab, bf16x2v));
    const f32x2v hv = {__uint_as_float(p << 16), __uint_as_float(p & 0xffff0000u)};
    const f32x2v r = ab - hv;                                  // one packed subtract (straight halves: finding 17 does not apply): k_shade3 119.2 -> 116.8 us
    H[j] = p;
    L[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2v));
  }
  hi = __builtin_bit_cast(bf16x8, make_uint4(H[0], H[1], H[2], H[3]));
  lo = __builtin_bit_cast(bf16x8, make_uint4(L[0], L[1], L[2], L[3]));
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// acc[m] += A(frag0 + m * stride) x B for NM output tiles, three-term split product, term-major: the MFMAs that follow
// each other write different accumulators
template <int NM>
__device__ __forceinline__ void mma3_step(const uint4* img, int frag0, int stride, int lane, bf16x8 bh, bf16x8 bl, f32x16* acc) {
  bf16x8 ah[NM], al[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) { ah[m] = w32_frag(img, frag0 + m * stride, 0, lane); al[m] = w32_frag(img, frag0 + m * stride, 1, lane); }
#pragma unroll
  for (int m = 0; m < NM; ++m) acc[m] = mfma32(al[m], bh, acc[m]);
#pragma unroll
  for (int m = 0; m < NM; ++m) acc[m] = mfma32(ah[m], bl, acc[m]);
#pragma unroll
  for (int m = 0; m < NM; ++m) acc[m] = mfma32(ah[m], bh, acc[m]);
}


// the 12 appearance products of lane half h for plane p: channels 12 h .. 12 h + 11 of the dense texel, three aligned
// float4 per tap (tensoRF.py:153-195); same per-channel arithmetic, in the same order, as gather_app6_plane32
// the 12 appearance products of lane half h for plane p: channels 12 h .. 12 h + 11 of the dense texel, three aligned
// float4 per tap (tensoRF.py:153-195); same per-channel arithmetic, in the same order, as gather_app6_plane32
template <int p>
__device__ __forceinline__ void gather_app12(const DField& f, const AxisTaps& at, int h, float X[12]) {
  const int x0 = at.i0[MAT0[p]], x1 = at.i1[MAT0[p]], y0 = at.i0[MAT1[p]], y1 = at.i1[MAT1[p]];
  const int l0 = at.i0[VEC[p]], l1 = at.i1[VEC[p]];
  const float tx = at.t[MAT0[p]], ty = at.t[MAT1[p]], tl = at.t[VEC[p]];
  const unsigned hb = 48u * (unsigned)h;
  const unsigned row0 = (unsigned)y0 * (unsigned)f.pw[p], row1 = (unsigned)y1 * (unsigned)f.pw[p];
  const unsigned o00 = (row0 + x0) * (LRF_CA * 4u) + hb, o10 = (row0 + x1) * (LRF_CA * 4u) + hb;
  const unsigned o01 = (row1 + x0) * (LRF_CA * 4u) + hb, o11 = (row1 + x1) * (LRF_CA * 4u) + hb;
  const unsigned q0 = (unsigned)l0 * (LRF_CA * 4u) + hb, q1 = (unsigned)l1 * (LRF_CA * 4u) + hb;
  const float w00 = (1.0f - tx) * (1.0f - ty), w10 = tx * (1.0f - ty);
  const float w01 = (1.0f - tx) * ty,          w11 = tx * ty;
  const float wl0 = 1.0f - tl, wl1 = tl;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float4 a = ld4b(f.aplane2[p], o00 + 16 * i), b = ld4b(f.aplane2[p], o10 + 16 * i);
    const float4 c = ld4b(f.aplane2[p], o01 + 16 * i), d = ld4b(f.aplane2[p], o11 + 16 * i);
    const float4 e = ld4b(f.aline2[p], q0 + 16 * i), q = ld4b(f.aline2[p], q1 + 16 * i);
    X[4 * i]     = (a.x * w00 + b.x * w10 + c.x * w01 + d.x * w11) * (e.x * wl0 + q.x * wl1);
    X[4 * i + 1] = (a.y * w00 + b.y * w10 + c.y * w01 + d.y * w11) * (e.y * wl0 + q.y * wl1);
    X[4 * i + 2] = (a.z * w00 + b.z * w10 + c.z * w01 + d.z * w11) * (e.z * wl0 + q.z * wl1);
    X[4 * i + 3] = (a.w * w00 + b.w * w10 + c.w * w01 + d.w * w11) * (e.w * wl0 + q.w * wl1);
  }
}

// per-ray tile counts -> exclusive prefix sum, one 1024-thread block (the caller's R does not fit the LDS copy)
template <int ITEMSZ>
__global__ __launch_bounds__(1024) void k_scan_tiles_n(const int* __restrict__ ncomp, int R, int* __restrict__ toff) {
  __shared__ int s_wave[16];
  __shared__ int s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < R; base += 1024) {
    const int r = base + tid;
    const int v = r < R ? (ncomp[r] + ITEMSZ - 1) / ITEMSZ : 0;
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

// first index r in [0, n] with toff[r] >= v (toff non-decreasing, n + 1 entries)
template <class P>
__device__ __forceinline__ int toff_lower_bound_p(P toff, int n, int v) {
  int lo = 0, hi = n + 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (toff[mid] >= v) hi = mid; else lo = mid + 1;
  }
  return lo < n ? lo : n;
}

// What one wave needs to know about a tile before it can gather.  The loads are issued one tile ahead and first used at
// the top of the next tile, so the dependent chain tile -> ray -> sample index -> distance is off the critical path.
struct Hdr3 {
  int ray, tile_in_ray, k;       // ray, tile number inside the ray, this lane's sample index into z
  int cnt;                       // samples in the tile (SAVE)
  float wgt;                     // this lane's compositing weight (0 for lanes beyond the tile's count and for K half 1)
  float o[3], d[3];              // ray origin, unit direction
};

// What the training forward keeps for the backward (SAVE; lrf_backward.inl, lrf_common.h): a 32-sample tile t of this kernel
// is the pair 2 t, 2 t + 1 of the backward's 16-row tiles (lane n -> row n & 15 of tile 2 t + (n >> 4)), so a ray owns
// 2 ceil(n / 32) of them and the second of a pair may be empty.
struct SaveOut3 {
  float* crgb;            // [ray * S + j][3] sigmoid colour of compact sample j
  float* act;             // feat rows, 16-row fragment order (ACT_LD)
  uint32_t* relu_bits;    // [16-row tile][layer][lane s + 16 g]: bit 4 t1 + r = unit 16 t1 + 4 g + r is active
  int4* tileinfo;         // [16-row tile] (ray, first compact sample, count, tile number inside the ray)
  int* toff16;            // [R + 1] offsets of the rays' 16-row tiles (= 2 x this kernel's tile offsets)
};

// NW waves per workgroup (one workgroup per CU).  LDSTOFF: the tile offsets are scanned by every workgroup itself into
// LDS (R + 1 ints beside the image: two launches per render, k_march -> k_shade3); otherwise k_scan_tiles_n<32> ran before
// and toff_g holds them.
// TIMED (debug, lrf_debug_set_dump): s_memtime totals per wave -> dump[block][wave][8] =
// {rest of the prologue, header + position, gather + split, image copy, scan, chain, tiles, finalize}
// SAVE: the training forward -- the same tile loop additionally writes what SaveOut3 lists (128 B of feat + 32 B of mask bits
// + 12 B of colour per shaded sample): the eval kernel IS the row-saving forward, its rgb is bit-identical to the eval's.
template <int NW, bool LDSTOFF, bool TIMED = false, bool SAVE = false>
__global__ __launch_bounds__(NW * 64) void k_shade3(
    DField f, const float* __restrict__ rays, const float* __restrict__ z, int S,
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
  {
    // Every workgroup starts its copy of the 95 KB image at a different place.  Started at the same place, the CUs of an
    // XCD ask the same L2 channel for the same line at the same time and the copy runs at ~11 B / cycle / CU (8.2 K
    // cycles); rotated it takes 5.0 K (colour stage 120.6 -> 119.4 us, scripts/ab_shade.sh).  Rotating the reads of the
    // per-ray counts and of k_march's line staging the same way gains nothing measurable.
    const int rot = (int)((blockIdx.x * 37u) % 93u) * 64;
    for (int i = tid; i < W32_ALL_U4; i += NT) { int j = i + rot; if (j >= W32_ALL_U4) j -= W32_ALL_U4; img[j] = f.mlpw[j]; }
  }
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
    const float* rp = rays + (size_t)w_ray * 6;
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
  // rgb_map = sum_k w_k rgb_k (+ 1 - acc) (tensorBase.py:632-634): this workgroup wrote every partial of its rays.  Its
  // waves share one L1 and these lines were never read before in this launch; the stores only have to be acknowledged.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int r = ra + tid; r < rb; r += NT) finalize_ray<false>(r, (int)toff[r + 1] - (int)toff[r], pmax, flags, acc, part, rgb_out, acc_out, f.perm ? f.perm[r] : r);
  LRF_TICK(7);
  if (TIMED && f.dump && lane == 0) {
    unsigned long long* dp = reinterpret_cast<unsigned long long*>(f.dump) + ((size_t)blockIdx.x * NW + wave) * 8;
    for (int i = 0; i < 8; ++i) dp[i] = tk[i];
  }
#undef LRF_TICK
}

}  // namespace lrf
