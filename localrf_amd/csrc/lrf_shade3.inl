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

#define LRF_SHADE3_NAME k_shade3
#define LRF_SHADE3_MULTI 0
#include "lrf_shade3_kernel.inl"
#undef LRF_SHADE3_NAME
#undef LRF_SHADE3_MULTI
#define LRF_SHADE3_NAME k_shade3m
#define LRF_SHADE3_MULTI 1
#include "lrf_shade3_kernel.inl"
#undef LRF_SHADE3_NAME
#undef LRF_SHADE3_MULTI

}  // namespace lrf
