// lrf_shade2.inl -- the colour stage as TWO kernels (default engine), included by lrf_render.hip.
//
// Round 1's k_shade_bf16 does gather -> basis -> 128 -> 128 -> head per 16-sample tile in one
// persistent kernel.  Its counters (profiles/r01v_round_end.md) say the waves sit parked on
// s_waitcnt 63 % of the time: each tile walks ~7 dependent memory round trips (tile header chain,
// then three gather rounds, each consumed by its MFMAs before the next is issued) with only four
// waves per SIMD to hide them, because the 97 KB weight image and the 128-register budget of a
// 1024-thread workgroup cap the occupancy.  The two halves want different machines:
//
//   k_app   gather + basis(72 -> 27).  Memory-latency / texture-path bound, 18 MFMAs per tile.
//           Needs only the 12 KB basis fragments in LDS -> 256-thread workgroups, several per CU;
//           the tile header is prefetched one tile ahead and the sample distances sit in LDS, so a
//           tile costs three gather rounds and nothing else.  Writes the 27 features of every shaded
//           sample ALREADY as the split-bf16 B fragment of layer 1 (hi + lo, 2 x 16 B per lane,
//           2 KB per tile, coalesced): 128 B per sample against the 1728 B it gathered.
//   k_mlp   27 -> 128 -> 128 -> 3 + sigmoid + weighting.  Matrix-pipe / VALU bound, no gathers: the
//           only global reads are the next tile's fragment (prefetched) and its weights.  Persistent,
//           one 1024-thread workgroup per CU with the W1/W2 image (83 KB) resident in LDS.
//
// Arithmetic is identical to k_shade_bf16 (same fragments, same MFMA order, same head): the two
// engines agree bit for bit, which tests/test_gpu_parity.py checks.
#pragma once
#include <type_traits>

namespace lrf {

// Tile walk with the per-ray state cached: consecutive tiles of a wave mostly belong to the same ray,
// so the header of a tile is ONE dependent load (the sample index) unless the ray changes.
struct TileWalk2 {
  int ray, tile0, next_off, nc;     // current ray, its first tile, first tile of the next ray, ncomp[ray]
};
typedef __attribute__((address_space(3))) int lds_int;       // tile offsets held in LDS by the fused k_shade2
template <class P>
__device__ __forceinline__ void tile_range(P toff, int R, int& t0, int& t1) {
  const int T = toff[R];
  const long long waves = (long long)gridDim.x * (blockDim.x >> 6);
  const int nb = gridDim.x;                                    // XCD-aware block order, see tile_walk_begin
  const int lb = (nb % 8 == 0) ? (int)(blockIdx.x % 8) * (nb / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const long long wid = (long long)lb * (blockDim.x >> 6) + (threadIdx.x >> 6);
  t0 = __builtin_amdgcn_readfirstlane((int)(wid * T / waves));
  t1 = __builtin_amdgcn_readfirstlane((int)((wid + 1) * T / waves));
}
template <class P>
__device__ __forceinline__ TileWalk2 tile_walk2_begin(P toff, const int* __restrict__ ncomp,
                                                      int R, int t) {
  int lo = 0, hi = R;                                          // largest ray with toff[ray] <= t
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (toff[mid] <= t) lo = mid; else hi = mid;
  }
  TileWalk2 tw;
  tw.ray = lo; tw.tile0 = toff[lo]; tw.next_off = toff[lo + 1]; tw.nc = ncomp[lo];
  return tw;
}
// advance to the ray owning tile t (skips rays without shaded samples); true if the ray changed
template <class P>
__device__ __forceinline__ bool tile_walk2_seek(TileWalk2& tw, P toff,
                                                const int* __restrict__ ncomp, int t) {
  bool moved = false;
  while (tw.next_off <= t) { ++tw.ray; tw.tile0 = tw.next_off; tw.next_off = toff[tw.ray + 1]; moved = true; }
  if (moved) tw.nc = ncomp[tw.ray];
  return moved;
}

// ReLU on the integer pipe: for x >= +0 the bit pattern is a non-negative int, for x < 0 (and -0) a negative
// one, so max_i32(bits, 0) is relu(x) in ONE instruction (fmaxf(x, 0) costs two: IEEE maxNum first quiets its
// operand with v_max x, x, x).
__device__ __forceinline__ float relu_i(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

struct RayGeo { float o[3], dh[3]; };
__device__ __forceinline__ RayGeo load_ray(const float* __restrict__ rays, int ray) {
  const float* rp = rays + (size_t)ray * 6;
  RayGeo g;
  g.o[0] = rp[0]; g.o[1] = rp[1]; g.o[2] = rp[2];
  const float dn = sqrtf(rp[3] * rp[3] + rp[4] * rp[4] + rp[5] * rp[5]);      // tensorBase.py:578-580
  g.dh[0] = rp[3] / dn; g.dh[1] = rp[4] / dn; g.dh[2] = rp[5] / dn;
  return g;
}

// gather_app6_plane with 32-bit byte offsets (ld4b): same taps, same arithmetic order
template <int p>
__device__ __forceinline__ void gather_app6_plane32(const DField& f, const AxisTaps& at, int g, float X[8]) {
  const int x0 = at.i0[MAT0[p]], x1 = at.i1[MAT0[p]], y0 = at.i0[MAT1[p]], y1 = at.i1[MAT1[p]];
  const int l0 = at.i0[VEC[p]], l1 = at.i1[VEC[p]];
  const float tx = at.t[MAT0[p]], ty = at.t[MAT1[p]], tl = at.t[VEC[p]];
  const unsigned gb = 32u * (unsigned)g;                                   // this lane group's 8 slots of the 128-byte texel
  const unsigned row0 = (unsigned)y0 * (unsigned)f.pw[p], row1 = (unsigned)y1 * (unsigned)f.pw[p];
  const unsigned o00 = (row0 + x0) * (LRF_CAS * 4u) + gb, o10 = (row0 + x1) * (LRF_CAS * 4u) + gb;
  const unsigned o01 = (row1 + x0) * (LRF_CAS * 4u) + gb, o11 = (row1 + x1) * (LRF_CAS * 4u) + gb;
  const unsigned q0 = (unsigned)l0 * (LRF_CAS * 4u) + gb, q1 = (unsigned)l1 * (LRF_CAS * 4u) + gb;
  const float w00 = (1.0f - tx) * (1.0f - ty), w10 = tx * (1.0f - ty);
  const float w01 = (1.0f - tx) * ty,          w11 = tx * ty;
  const float wl0 = 1.0f - tl, wl1 = tl;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float4 a = ld4b(f.aplane[p], o00 + 16 * h), b = ld4b(f.aplane[p], o10 + 16 * h);
    const float4 c = ld4b(f.aplane[p], o01 + 16 * h), d = ld4b(f.aplane[p], o11 + 16 * h);
    const float4 e = ld4b(f.aline[p], q0 + 16 * h), q = ld4b(f.aline[p], q1 + 16 * h);
    X[4 * h]     = (a.x * w00 + b.x * w10 + c.x * w01 + d.x * w11) * (e.x * wl0 + q.x * wl1);
    X[4 * h + 1] = (a.y * w00 + b.y * w10 + c.y * w01 + d.y * w11) * (e.y * wl0 + q.y * wl1);
    if (h == 0) {
      X[2] = (a.z * w00 + b.z * w10 + c.z * w01 + d.z * w11) * (e.z * wl0 + q.z * wl1);
      X[3] = (a.w * w00 + b.w * w10 + c.w * w01 + d.w * w11) * (e.w * wl0 + q.w * wl1);
    }
  }
  X[6] = 0.0f; X[7] = 0.0f;            // slots 6, 7 of a lane group are the texel's zero pads (app_pc): +0 x weights = +0
}

// ------------------------------------------------------------------------------- k_app
__global__ __launch_bounds__(256) void k_app(
    DField f, const float* __restrict__ rays, const float* __restrict__ z, int S,
    const int* __restrict__ toff, int R, const int* __restrict__ ncomp, const uint16_t* __restrict__ cidx,
    uint4* __restrict__ ffrag /* [tile][hi, lo][lane] */, int2* __restrict__ tinfo /* [tile] (ray, j0 * 32 + count) */) {
  extern __shared__ uint4 s_dyn[];                             // basis fragments, then z[S]
  uint4* bas = s_dyn;
  float* s_z = reinterpret_cast<float*>(s_dyn + IMGB_W1);
  for (int i = threadIdx.x; i < IMGB_W1; i += blockDim.x) bas[i] = f.mlpb[i];
  for (int i = threadIdx.x; i < S; i += blockDim.x) s_z[i] = z[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, s = lane & 15, g = lane >> 4;
  int t0, t1;
  tile_range(toff, R, t0, t1);
  if (t0 >= t1) return;
  TileWalk2 tw = tile_walk2_begin(toff, ncomp, R, t0);
  tile_walk2_seek(tw, toff, ncomp, t0);
  RayGeo rg = load_ray(rays, tw.ray);
  int j0 = (t0 - tw.tile0) * ITEM;
  int cnt = min(ITEM, tw.nc - j0);
  int k = cidx[(size_t)tw.ray * S + j0 + (s < cnt ? s : 0)];
  int ray_c = tw.ray;
  for (int t = t0; t < t1; ++t) {
    asm volatile("" ::: "memory");     // keep the LDS fragment reads inside the loop
    float x[3], u[3];
    sample_point(f, rg.o, rg.dh, s_z[k], x, u);
    // header of the next tile: one dependent load, issued before this tile's gathers and consumed
    // after them
    int k_n = 0, j0_n = 0, cnt_n = 0, ray_n = ray_c;
    RayGeo rg_n = rg;
    if (t + 1 < t1) {
      if (tile_walk2_seek(tw, toff, ncomp, t + 1)) rg_n = load_ray(rays, tw.ray);
      ray_n = tw.ray;
      j0_n = (t + 1 - tw.tile0) * ITEM;
      cnt_n = min(ITEM, tw.nc - j0_n);
      k_n = cidx[(size_t)tw.ray * S + j0_n + (s < cnt_n ? s : 0)];
    }
    // basis 72 -> 27 (tensoRF.py:196), one k-step per plane, exactly as in k_shade_bf16
    f32x4 fe[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    asm volatile("" : "+v"(fe[0]), "+v"(fe[1]));
    {
      float v[8];
      bf16x8 bh, bl;
      const AxisTaps at = axis_taps(f.pw[0], f.ph[0], f.ll[0], u);
      gather_app6_plane32<0>(f, at, g, v);
      split8(v, bh, bl);
      gemm_step<2>(bas, IMGB_BAS / 128 + 0, 3, lane, bh, bl, fe);
      gather_app6_plane32<1>(f, at, g, v);
      split8(v, bh, bl);
      gemm_step<2>(bas, IMGB_BAS / 128 + 1, 3, lane, bh, bl, fe);
      gather_app6_plane32<2>(f, at, g, v);
      split8(v, bh, bl);
      gemm_step<2>(bas, IMGB_BAS / 128 + 2, 3, lane, bh, bl, fe);
      settle<2>(fe);
    }
    {
      const float v[8] = {fe[0][0], fe[0][1], fe[0][2], fe[0][3], fe[1][0], fe[1][1], fe[1][2], fe[1][3]};
      bf16x8 bh, bl;
      split8(v, bh, bl);
      ffrag[((size_t)t * 2 + 0) * 64 + lane] = __builtin_bit_cast(uint4, bh);
      ffrag[((size_t)t * 2 + 1) * 64 + lane] = __builtin_bit_cast(uint4, bl);
      if (lane == 0) tinfo[t] = make_int2(ray_c, j0 * 32 + cnt);
    }
    k = k_n; j0 = j0_n; cnt = cnt_n; rg = rg_n; ray_c = ray_n;
  }
}

// ------------------------------------------------------------------------------- k_mlp
// MFMA issue policy of the layer-1 / layer-2 chains (runtime choice for the hazard experiments of
// DESIGN.md "gfx950 / hipcc findings"; the shipped default is POLICY 4, the hand-issued fallback POLICY 0):
//   0  hand-issued in-place MFMA, 4 wait states behind each, operands held 48 wait states (= k_shade_bf16)
//   1  hand-issued in-place MFMA, 2 wait states behind each, A fragments held for two further
//      fragments (>= 6 MFMAs) by register rotation instead of wait states, no tail pad
//   2  compiler-scheduled builtin, operands kept live the same way (hold), no wait states
//   3  compiler-scheduled builtin, nothing else (the build that showed run-to-run differences in round 1)
//   4  as 3, with the A fragments fetched from LDS two fragments ahead of their MFMAs (explicit software
//      pipeline: the 16 waves of a workgroup run in lockstep phases, so without it every wave sits on the
//      LDS queue at the same time -- 57 % of wave time parked on s_waitcnt, profiles/r02c)
//   5  as 1 (hand-issued), with the same two-fragment prefetch
template <int POLICY>
__device__ __forceinline__ void mfma_p(bf16x8 a, bf16x8 b, f32x4& acc) {
  if (POLICY >= 2 && POLICY <= 4) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  } else {
    const i32x4 ai = __builtin_bit_cast(i32x4, a), bi = __builtin_bit_cast(i32x4, b);
    if (POLICY == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 3" : "+v"(acc) : "v"(ai), "v"(bi));
    else             asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 1" : "+v"(acc) : "v"(ai), "v"(bi));
  }
}
template <int POLICY>
__device__ __forceinline__ void split8_p(const float v[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];
    hi[j] = h;
    lo[j] = (__bf16)(v[j] - (float)h);
  }
  if (POLICY == 0) {
    uint4 H = __builtin_bit_cast(uint4, hi), L = __builtin_bit_cast(uint4, lo);
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(H.x), "+v"(H.y), "+v"(H.z), "+v"(H.w),
                                          "+v"(L.x), "+v"(L.y), "+v"(L.z), "+v"(L.w));
    hi = __builtin_bit_cast(bf16x8, H);
    lo = __builtin_bit_cast(bf16x8, L);
  }
}
template <int POLICY, int NT>
__device__ __forceinline__ void settle_p(f32x4* acc) {
  if (POLICY <= 1 || POLICY == 5) {
#pragma unroll
    for (int t = 0; t < NT; ++t) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc[t]));
  }
}
// acc[t1] += A(frag0 + t1*stride) x B, three-term split product (see gemm_step)
template <int POLICY, int NT>
__device__ __forceinline__ void gemm_step_q(const uint4* img, int frag0, int stride, int lane,
                                            bf16x8 bh, bf16x8 bl, f32x4* acc) {
  if (POLICY == 0) { gemm_step<NT>(img, frag0, stride, lane, bh, bl, acc); return; }
  if (POLICY >= 4) {
    bf16x8 ah[NT], al[NT];                                      // fully unrolled: only ~3 fragments are live at a time
    ah[0] = lds_frag(img, frag0, 0, lane); al[0] = lds_frag(img, frag0, 1, lane);
    if constexpr (NT > 1) { ah[1] = lds_frag(img, frag0 + stride, 0, lane); al[1] = lds_frag(img, frag0 + stride, 1, lane); }
#pragma unroll
    for (int t1 = 0; t1 < NT; ++t1) {
      if (t1 + 2 < NT) {
        ah[(t1 + 2) % NT] = lds_frag(img, frag0 + (t1 + 2) * stride, 0, lane);
        al[(t1 + 2) % NT] = lds_frag(img, frag0 + (t1 + 2) * stride, 1, lane);
      }
      if (POLICY == 4) __builtin_amdgcn_sched_barrier(0);       // keep the fetch ahead of this fragment's MFMAs
      mfma_p<POLICY>(al[t1], bh, acc[t1]);
      mfma_p<POLICY>(ah[t1], bl, acc[t1]);
      mfma_p<POLICY>(ah[t1], bh, acc[t1]);
      if (POLICY == 5 && t1 >= 2) { hold(ah[(t1 - 2) % NT]); hold(al[(t1 - 2) % NT]); }
    }
    if (POLICY == 5) { hold(ah[NT - 1]); hold(al[NT - 1]); if constexpr (NT > 1) { hold(ah[NT - 2]); hold(al[NT - 2]); } }
    return;
  }
  bf16x8 p1h = bh, p1l = bl, p2h = bh, p2l = bl;               // the two previous A fragments (dummies at first)
#pragma unroll
  for (int t1 = 0; t1 < NT; ++t1) {
    const bf16x8 ah = lds_frag(img, frag0 + t1 * stride, 0, lane);
    const bf16x8 al = lds_frag(img, frag0 + t1 * stride, 1, lane);
    mfma_p<POLICY>(al, bh, acc[t1]);
    mfma_p<POLICY>(ah, bl, acc[t1]);
    mfma_p<POLICY>(ah, bh, acc[t1]);
    if (POLICY <= 2) { hold(p2h); hold(p2l); }                 // >= 6 MFMAs behind their last use
    p2h = p1h; p2l = p1l; p1h = ah; p1l = al;
  }
  if (POLICY <= 2) { hold(p1h); hold(p1l); hold(p2h); hold(p2l); }
}

// TIMED (debug, lrf_debug_set_dump): per-wave s_memtime totals of the phases of a tile -> dump[wave][8]
// {header+prefetch issue, layer 1, layer 2, head+store, tiles}
// HEADM: the 128 -> 3 head as a fourth split-bf16 MFMA layer (12 MFMAs per tile) instead of 96 fp32 FMAs,
// 32 LDS weight reads and six cross-row shuffles per lane on the VALU / LDS pipes.
template <int POLICY, bool TIMED = false, bool HEADM = true>
__global__ __launch_bounds__(1024) void k_mlp(
    DField f, const float* __restrict__ rays, int S, const int* __restrict__ toff, int R,
    const int* __restrict__ ncomp, const float* __restrict__ cw, const uint4* __restrict__ ffrag,
    const int2* __restrict__ tinfo, float* __restrict__ part, int pmax) {
  unsigned long long tk[5] = {0, 0, 0, 0, 0}, tlast = 0;
#define LRF_TICK(i) do { if (TIMED) { const unsigned long long now_ = __builtin_readcyclecounter(); tk[i] += now_ - tlast; tlast = now_; } } while (0)
  __shared__ uint4 img[IMGB_ALL - IMGB_W1];                    // W1, W2 fragments, fp32 tail, head fragments (91 KB)
  for (int i = threadIdx.x; i < IMGB_ALL - IMGB_W1; i += blockDim.x) img[i] = f.mlpb[IMGB_W1 + i];
  __syncthreads();
  constexpr int F_W1 = 0, F_W2 = (IMGB_W2 - IMGB_W1) / 128, F_W3 = (IMGB_W3F - IMGB_W1) / 128;   // fragment indices inside img
  const float* tail = reinterpret_cast<const float*>(img + (IMGB_TAIL - IMGB_W1));
  const int lane = threadIdx.x & 63, s = lane & 15, g = lane >> 4;
  // Tiles: this workgroup owns one contiguous range (XCD-aware block order); its waves pull tiles from
  // an LDS counter.  With a static split per wave the waves of a SIMD finished up to 2x apart (the oldest
  // wave wins every arbitration; s_memtime totals 85K..175K cycles per wave, scripts/gpu_diag.py mlp_phases),
  // and the kernel lasts as long as its slowest wave.
  __shared__ int s_next;
  const int T = toff[R];
  const int nb = gridDim.x, nw = blockDim.x >> 6;
  const int lb = (nb % 8 == 0) ? (int)(blockIdx.x % 8) * (nb / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int T0 = (int)((long long)lb * T / nb), T1 = (int)((long long)(lb + 1) * T / nb);
  if (threadIdx.x == 0) s_next = T0 + nw;
  __syncthreads();
  int t = __builtin_amdgcn_readfirstlane(T0 + (int)(threadIdx.x >> 6));
  if (t >= T1) return;
  // state of the current tile: its header (k_app wrote (ray, j0 * 32 + count) per tile) and its fragment are
  // fetched one tile ahead; weight and view direction are only needed at the end of a tile and are loaded at
  // its top
  int2 hdr = tinfo[t];
  uint4 fh = ffrag[((size_t)t * 2 + 0) * 64 + lane], fl = ffrag[((size_t)t * 2 + 1) * 64 + lane];
  if (TIMED) tlast = __builtin_readcyclecounter();
  while (t < T1) {
    asm volatile("" ::: "memory");     // keep LDS weight reads inside the loop (no LICM -> no spills)
    const int ray = __builtin_amdgcn_readfirstlane(hdr.x);
    const int j0 = __builtin_amdgcn_readfirstlane(hdr.y) >> 5, cnt = __builtin_amdgcn_readfirstlane(hdr.y) & 31;
    const float w = s < cnt ? cw[(size_t)ray * S + j0 + s] : 0.0f;
    const float* rp = rays + (size_t)ray * 6;
    const float d0 = rp[3], d1 = rp[4], d2 = rp[5];
    int t_n = 0;
    if (lane == 0) t_n = atomicAdd(&s_next, 1);
    t_n = __builtin_amdgcn_readfirstlane(t_n);
    int2 hdr_n = hdr;
    uint4 fh_n = fh, fl_n = fl;
    if (t_n < T1) {
      hdr_n = tinfo[t_n];
      fh_n = ffrag[((size_t)t_n * 2 + 0) * 64 + lane];
      fl_n = ffrag[((size_t)t_n * 2 + 1) * 64 + lane];
    }
    LRF_TICK(0);
    // layer 1 (tensorBase.py:129-130): one k-step, B = the fragment k_app wrote
    f32x4 h1[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) h1[q] = *reinterpret_cast<const f32x4*>(&tail[TAIL_B1 + 16 * q + 4 * g]);
    {
      bf16x8 bh = __builtin_bit_cast(bf16x8, fh), bl = __builtin_bit_cast(bf16x8, fl);
      if (POLICY <= 1 || POLICY == 5) {                        // loaded operands: settle before the asm MFMAs
        uint4 H = __builtin_bit_cast(uint4, bh), L = __builtin_bit_cast(uint4, bl);
        asm volatile("s_nop 1" : "+v"(H.x), "+v"(H.y), "+v"(H.z), "+v"(H.w), "+v"(L.x), "+v"(L.y), "+v"(L.z), "+v"(L.w),
                                 "+v"(h1[0]), "+v"(h1[1]), "+v"(h1[2]), "+v"(h1[3]), "+v"(h1[4]), "+v"(h1[5]), "+v"(h1[6]), "+v"(h1[7]));
        bh = __builtin_bit_cast(bf16x8, H);
        bl = __builtin_bit_cast(bf16x8, L);
      }
      gemm_step_q<POLICY, 8>(img, F_W1, 1, lane, bh, bl, h1);
      settle_p<POLICY, 8>(h1);
    }
    LRF_TICK(1);
    // layer 2: 4 k-steps, k-step ks consumes relu(h1) tiles 2ks and 2ks+1
    f32x4 h2[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) h2[q] = *reinterpret_cast<const f32x4*>(&tail[TAIL_B2 + 16 * q + 4 * g]);
    if (POLICY <= 1 || POLICY == 5) {
      asm volatile("s_nop 1" : "+v"(h2[0]), "+v"(h2[1]), "+v"(h2[2]), "+v"(h2[3]), "+v"(h2[4]), "+v"(h2[5]), "+v"(h2[6]), "+v"(h2[7]));
    }
    bf16x8 pbh = {}, pbl = {};                                  // previous k-step's B operands
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = relu_i(h1[2 * ks + (j >> 2)][j & 3]);
      bf16x8 bh, bl;
      split8_p<POLICY>(v, bh, bl);
      gemm_step_q<POLICY, 8>(img, F_W2 + ks, 4, lane, bh, bl, h2);
      if (POLICY == 1 || POLICY == 2 || POLICY == 5) { if (ks) { hold(pbh); hold(pbl); } pbh = bh; pbl = bl; }
    }
    if (POLICY == 1 || POLICY == 2 || POLICY == 5) { hold(pbh); hold(pbl); }
    settle_p<POLICY, 8>(h2);
    LRF_TICK(2);
    // head (tensorBase.py:131-133)
    float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
    if (HEADM) {
      // as a fourth layer: rows 0..2 of D are r,g,b -- lanes of group g = 0 hold them for their sample
      f32x4 oc = {0, 0, 0, 0};
      if (POLICY <= 1 || POLICY == 5) asm volatile("" : "+v"(oc));
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = relu_i(h2[2 * ks + (j >> 2)][j & 3]);
        bf16x8 bh, bl;
        split8_p<POLICY>(v, bh, bl);
        gemm_step_q<POLICY, 1>(img, F_W3 + ks, 1, lane, bh, bl, &oc);
      }
      settle_p<POLICY, 1>(&oc);
      o0 = oc[0]; o1 = oc[1]; o2 = oc[2];
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float hv = relu_i(h2[q][r]);
          const float4 wv = *reinterpret_cast<const float4*>(&tail[TAIL_W3H + g * TAIL_W3H_GS + (q * 4 + r) * 4]);
          o0 += hv * wv.x; o1 += hv * wv.y; o2 += hv * wv.z;
        }
      o0 += __shfl_xor(o0, 16, 64); o1 += __shfl_xor(o1, 16, 64); o2 += __shfl_xor(o2, 16, 64);
      o0 += __shfl_xor(o0, 32, 64); o1 += __shfl_xor(o1, 32, 64); o2 += __shfl_xor(o2, 32, 64);
    }
    // view-direction part of mlp_view.0 + bias: constant per ray (tensorBase.py:131-132; viewdirs
    // detached :628)
    float vb[3];
    {
      // d / |d| (tensorBase.py:578-580) with the hardware reciprocal square root (1 ulp) instead of sqrt + 3 divides
      const float inv = __frsqrt_rn(d0 * d0 + d1 * d1 + d2 * d2);
      const float dh[3] = {d0 * inv, d1 * inv, d2 * inv};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float4 wv = *reinterpret_cast<const float4*>(&tail[TAIL_W3V + 4 * c]);
        vb[c] = wv.w + wv.x * dh[0] + wv.y * dh[1] + wv.z * dh[2];
      }
    }
    // w * sigmoid(x) (:133, :632): hardware exp2 / reciprocal (each ~1 ulp; a 1e-7 relative change of a colour)
    const float wq = (HEADM && g != 0) ? 0.0f : w;              // MFMA head: only the g = 0 lanes hold a colour
    float cr = wq * __frcp_rn(1.0f + __expf(-(o0 + vb[0])));
    float cg = wq * __frcp_rn(1.0f + __expf(-(o1 + vb[1])));
    float cb = wq * __frcp_rn(1.0f + __expf(-(o2 + vb[2])));
#pragma unroll
    for (int dd = 1; dd < 16; dd <<= 1) {
      cr += __shfl_xor(cr, dd, 64); cg += __shfl_xor(cg, dd, 64); cb += __shfl_xor(cb, dd, 64);
    }
    if (lane == 0) {
      float* pp = part + ((size_t)ray * pmax + j0 / ITEM) * 3;
      pp[0] = cr; pp[1] = cg; pp[2] = cb;
    }
    t = t_n; hdr = hdr_n; fh = fh_n; fl = fl_n;
    LRF_TICK(3);
    tk[4] += 1;
  }
  if (TIMED && f.dump && lane == 0) {
    unsigned long long* dp = reinterpret_cast<unsigned long long*>(f.dump) + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8;
    for (int i = 0; i < 5; ++i) dp[i] = tk[i];
  }
#undef LRF_TICK
}

// gather_app6_plane32 in two halves: issue the twelve 16-byte loads of a plane (kept in registers), combine later
struct PlaneRaw { float4 a[2], b[2], c[2], d[2], e[2], q[2]; float tx, ty, tl; };
template <int p>
__device__ __forceinline__ PlaneRaw plane_issue(const DField& f, const float u[3], int g) {
  PlaneRaw r;
  int x0, x1, y0, y1, l0, l1;
  tap1d(u[MAT0[p]], f.pw[p], x0, x1, r.tx);
  tap1d(u[MAT1[p]], f.ph[p], y0, y1, r.ty);
  tap1d(u[VEC[p]],  f.ll[p], l0, l1, r.tl);
  const unsigned gb = 32u * (unsigned)g;
  const unsigned row0 = (unsigned)y0 * (unsigned)f.pw[p], row1 = (unsigned)y1 * (unsigned)f.pw[p];
  const unsigned o00 = (row0 + x0) * (LRF_CAS * 4u) + gb, o10 = (row0 + x1) * (LRF_CAS * 4u) + gb;
  const unsigned o01 = (row1 + x0) * (LRF_CAS * 4u) + gb, o11 = (row1 + x1) * (LRF_CAS * 4u) + gb;
  const unsigned q0 = (unsigned)l0 * (LRF_CAS * 4u) + gb, q1 = (unsigned)l1 * (LRF_CAS * 4u) + gb;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    r.a[h] = ld4b(f.aplane[p], o00 + 16 * h); r.b[h] = ld4b(f.aplane[p], o10 + 16 * h);
    r.c[h] = ld4b(f.aplane[p], o01 + 16 * h); r.d[h] = ld4b(f.aplane[p], o11 + 16 * h);
    r.e[h] = ld4b(f.aline[p], q0 + 16 * h);   r.q[h] = ld4b(f.aline[p], q1 + 16 * h);
  }
  return r;
}
__device__ __forceinline__ void plane_combine(const PlaneRaw& r, float X[8]) {
  const float w00 = (1.0f - r.tx) * (1.0f - r.ty), w10 = r.tx * (1.0f - r.ty);
  const float w01 = (1.0f - r.tx) * r.ty,          w11 = r.tx * r.ty;
  const float wl0 = 1.0f - r.tl, wl1 = r.tl;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float4 a = r.a[h], b = r.b[h], c = r.c[h], d = r.d[h], e = r.e[h], q = r.q[h];
    X[4 * h]     = (a.x * w00 + b.x * w10 + c.x * w01 + d.x * w11) * (e.x * wl0 + q.x * wl1);
    X[4 * h + 1] = (a.y * w00 + b.y * w10 + c.y * w01 + d.y * w11) * (e.y * wl0 + q.y * wl1);
    X[4 * h + 2] = (a.z * w00 + b.z * w10 + c.z * w01 + d.z * w11) * (e.z * wl0 + q.z * wl1);
    X[4 * h + 3] = (a.w * w00 + b.w * w10 + c.w * w01 + d.w * w11) * (e.w * wl0 + q.w * wl1);
  }
}

// Cache prefetch of a plane's six taps: one dword per lane and tap (the four lanes of a sample touch the four
// 32-byte pieces of the 128-byte texel, so every line the real 16-byte loads will read is requested), results
// summed into `sink` by the caller one iteration later.  A quarter of the texture-path cycles of a real load.
template <int p>
__device__ __forceinline__ void plane_touch(const DField& f, const float u[3], int g, float t[6]) {
  int x0, x1, y0, y1, l0, l1; float tx, ty, tl;
  tap1d(u[MAT0[p]], f.pw[p], x0, x1, tx);
  tap1d(u[MAT1[p]], f.ph[p], y0, y1, ty);
  tap1d(u[VEC[p]],  f.ll[p], l0, l1, tl);
  const unsigned gb = 32u * (unsigned)g;
  const unsigned row0 = (unsigned)y0 * (unsigned)f.pw[p], row1 = (unsigned)y1 * (unsigned)f.pw[p];
  const char* pl = reinterpret_cast<const char*>(f.aplane[p]);
  const char* ln = reinterpret_cast<const char*>(f.aline[p]);
  t[0] = *reinterpret_cast<const float*>(pl + ((row0 + x0) * (LRF_CAS * 4u) + gb));
  t[1] = *reinterpret_cast<const float*>(pl + ((row0 + x1) * (LRF_CAS * 4u) + gb));
  t[2] = *reinterpret_cast<const float*>(pl + ((row1 + x0) * (LRF_CAS * 4u) + gb));
  t[3] = *reinterpret_cast<const float*>(pl + ((row1 + x1) * (LRF_CAS * 4u) + gb));
  t[4] = *reinterpret_cast<const float*>(ln + ((unsigned)l0 * (LRF_CAS * 4u) + gb));
  t[5] = *reinterpret_cast<const float*>(ln + ((unsigned)l1 * (LRF_CAS * 4u) + gb));
}

// ------------------------------------------------------------------------------- k_shade2
// The fused colour kernel again (gather -> basis -> 128 -> 128 -> head per tile, one persistent 1024-thread
// workgroup per CU), rebuilt from what the split taught: the tile header is prefetched one tile ahead and the
// sample distances sit in LDS (k_shade_bf16 walks four dependent loads before its first gather), 32-bit saddr
// gathers, integer ReLU, hardware exp2 / rcp.  Gathers and an MFMA chain share this kernel, so every MFMA is
// hand-issued (policy 0: mfma_bf16_acc / hold / settle) and -- shipped variant VAR = 3 -- the prefetched header
// loads are drained before the first MFMA of a tile, so that no global load is in flight under the chain (free:
// 158.5 vs 158.4 us), and the head runs on the VALU (155.5 us; the 12 hand-padded head MFMAs cost more than the
// 96 FMAs they replace here).  The variant with the MFMA head and loads in flight showed 3 renders with 1e-6-scale
// differences in ~5000 on two of six boxes, none in 24000 on a third (scripts/gpu_diag.py flake / coldstart).
// PIPE: the next tile's plane-0 gather is issued before the head phase of the current tile (h1 is dead by then) and
// consumed at the top of the next iteration -- one of the three gather round trips per tile leaves the critical path.
// TOUCH = n: the lines of the next tile's first n planes are requested (plane_touch) before the head phase.
// VAR bit 2 (experiment): layers 1 and 2 on the compiler-scheduled builtin with LDS prefetch (policy 4 of k_mlp), fenced
// by scheduling barriers from the hand-issued basis phase -- no global load of this wave is in flight then.
// VAR (experiments on the rare run-to-run difference): bit 0 = wait for the prefetched header loads before the
// first MFMA (no global load in flight under the MFMA chain), bit 1 = head on the VALU as in k_shade_bf16.
// FUSE: the launch sequence of the default engine is k_march -> k_shade2<FUSE> (two launches instead of four).
//  * k_scan_tiles is gone: every workgroup scans the R per-ray tile counts itself into LDS (1024 threads, ~1 us,
//    under the image load) -- and the tile walk then reads its offsets from LDS instead of from L2;
//  * k_finalize is gone: a workgroup's waves own one contiguous tile range, so after its tile loop the workgroup
//    sums (in tile order, as k_finalize does: same bits) the partials of every ray whose tiles all lie inside
//    that range; the <= gridDim.x - 1 rays that straddle a workgroup boundary are summed by whichever workgroup
//    finishes last (release fence + counter f.ctr, zeroed by the k_march of the same call; agent-scope loads).
// No float atomics, no spinning: results do not depend on the order in which workgroups finish.
// plain 16-byte store for save_x_plane_with (k_shade2<SAVE>)
struct St16 { __device__ __forceinline__ void operator()(float* p, float4 q) const { *reinterpret_cast<float4*>(p) = q; } };
template <bool COHERENT>
__device__ __forceinline__ void finalize_ray(int ray, int nit, int pmax, uint32_t flags, const float* __restrict__ acc,
                                             const float* part, float* __restrict__ rgb, float* __restrict__ acc_out) {
  const float* pp = part + (size_t)ray * pmax * 3;
  float r = 0.0f, g = 0.0f, b = 0.0f;
  for (int i0 = 0; i0 < nit; i0 += 8) {                         // 8 tiles' partials in flight, summed in tile order
    float v[24];
#pragma unroll
    for (int j = 0; j < 24; ++j) {
      v[j] = 0.0f;
      if (i0 + j / 3 < nit)
        v[j] = COHERENT ? __hip_atomic_load(pp + i0 * 3 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : pp[i0 * 3 + j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (i0 + j < nit) { r += v[3 * j]; g += v[3 * j + 1]; b += v[3 * j + 2]; }
  }
  const float a = acc[ray];
  if (flags & LRF_FLAG_WHITE_BG) {
    const float bg = 1.0f - a;
    r += bg; g += bg; b += bg;
  }
  rgb[(size_t)ray * 3 + 0] = r; rgb[(size_t)ray * 3 + 1] = g; rgb[(size_t)ray * 3 + 2] = b;
  if (acc_out) acc_out[ray] = a;
}
// first index r in [0, n] with toff[r] >= v (toff non-decreasing, n + 1 entries)
__device__ __forceinline__ int toff_lower_bound(const lds_int* toff, int n, int v) {
  int lo = 0, hi = n + 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (toff[mid] >= v) hi = mid; else lo = mid + 1;
  }
  return lo < n ? lo : n;
}

// SAVE: the row-saving forward of training (lrf_render_fwd_train, and the recompute path of lrf_render_bwd): besides
// the tile partials it leaves, per shaded sample, the colour (crgb), the ACT row [X | feat, 1 | relu(h1), 1 |
// relu(h2), dhat, 1] (the 1 columns turn bias gradients into GEMM columns of k_wgrad) and the two layers' ReLU
// masks as one dword per lane and layer (bit 4 q + r = unit 16 q + 4 g + r of sample s) for k_bwd_shade_dgrad.
// With FUSE the tile offsets (needed by the backward kernels) are also written to global memory by workgroup 0.
template <bool TIMED, bool PIPE = false, int TOUCH = 0, int VAR = 0, int FUSE = 0, bool SAVE = false>
__global__ __launch_bounds__(1024) void k_shade2(
    DField f, const float* __restrict__ rays, const float* __restrict__ z, int S,
    const int* __restrict__ toff_g, int R, const int* __restrict__ ncomp, const uint16_t* __restrict__ cidx,
    const float* __restrict__ cw, float* __restrict__ part, int pmax, int skew,
    uint32_t flags, const float* __restrict__ acc, float* __restrict__ rgb_out, float* __restrict__ acc_out,
    float* __restrict__ crgb = nullptr, float* __restrict__ act = nullptr, uint32_t* __restrict__ relu_bits = nullptr,
    int* __restrict__ toff_out = nullptr) {
  static_assert(!FUSE || (VAR & 2), "the fused variant stages the image without the head fragments (VALU head)");
  extern __shared__ uint4 s_dyn[];                             // image (basis, W1, W2, tail[, head]), z[S][, toff[R + 1]]
  unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#define LRF_TICK(i) do { if (TIMED) { const unsigned long long now_ = __builtin_readcyclecounter(); tk[i] += now_ - tlast; tlast = now_; } } while (0)
  constexpr int NIMG = FUSE ? IMGB_U4 : IMGB_ALL;
  uint4* img = s_dyn;
  float* s_z = reinterpret_cast<float*>(s_dyn + NIMG);
  for (int i = threadIdx.x; i < NIMG; i += blockDim.x) img[i] = f.mlpb[i];
  for (int i = threadIdx.x; i < S; i += blockDim.x) s_z[i] = z[i];
  lds_int* s_toff = (lds_int*)(s_z + S);
  if (FUSE) {                                                  // k_scan_tiles, per workgroup, into LDS
    __shared__ int s_wave[16];
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    int carry = 0;
    for (int base = 0; base < R; base += 1024) {
      const int r = base + tid;
      const int v = r < R ? (ncomp[r] + ITEM - 1) / ITEM : 0;
      int incl = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (ln >= d) incl += t;
      }
      __syncthreads();                                         // s_wave of the previous round has been read
      if (ln == 63) s_wave[wv] = incl;
      __syncthreads();
      int woff = 0, tot = 0;
#pragma unroll
      for (int q = 0; q < 16; ++q) { const int x = s_wave[q]; woff += q < wv ? x : 0; tot += x; }
      if (r < R) s_toff[r] = carry + woff + incl - v;
      carry += tot;
    }
    if (tid == 0) s_toff[R] = carry;
  }
  __syncthreads();
  if (FUSE && SAVE && blockIdx.x == 0)
    for (int i = threadIdx.x; i <= R; i += blockDim.x) toff_out[i] = s_toff[i];
  const float* tail = reinterpret_cast<const float*>(img + IMGB_TAIL);
  const int lane = threadIdx.x & 63, s = lane & 15, g = lane >> 4;
  // Phase skew: the four waves of a SIMD (waves w, w+4, w+8, w+12 of the workgroup) start `skew` x 6400 cycles
  // apart.  All waves run the same code on same-sized tiles, so without it they stay in lockstep -- all in the
  // gather phases (texture path) at once, then all in the MFMA phases at once -- and the kernel's time is the SUM
  // of the phases (s_memtime: 9.0K gather + 14.6K MLP cycles per tile and wave) instead of their overlap.
  for (int i = 0, n = (int)(threadIdx.x >> 8) * skew; i < n; ++i) __builtin_amdgcn_s_sleep(100);
  int t0, t1;
  typedef typename std::conditional<FUSE != 0, const lds_int*, const int*>::type ToffP;
  ToffP toff;
  if constexpr (FUSE) toff = s_toff; else toff = toff_g;
  tile_range(toff, R, t0, t1);
  if (!FUSE && t0 >= t1) return;
  TileWalk2 tw = {0, 0, 0, 0};
  RayGeo rg = {{0, 0, 0}, {0, 0, 0}};
  int ray_c = 0, j0 = 0, cnt = 0, k = 0;
  if (t0 < t1) {
    tw = tile_walk2_begin(toff, ncomp, R, t0);
    tile_walk2_seek(tw, toff, ncomp, t0);
    rg = load_ray(rays, tw.ray);
    ray_c = tw.ray;
    j0 = (t0 - tw.tile0) * ITEM;
    cnt = min(ITEM, tw.nc - j0);
    k = cidx[(size_t)tw.ray * S + j0 + (s < cnt ? s : 0)];
  }
  float x[3], u[3];
  PlaneRaw raw0;
  if (PIPE && t0 < t1) {
    sample_point(f, rg.o, rg.dh, s_z[k], x, u);
    raw0 = plane_issue<0>(f, u, g);
  }
  float touch[TOUCH > 0 ? 6 * TOUCH : 1] = {};
  float sink = 0.0f;
  if (TIMED) tlast = __builtin_readcyclecounter();
  for (int t = t0; t < t1; ++t) {
    asm volatile("" ::: "memory");     // keep the LDS fragment reads inside the loop
    if (TOUCH > 0) {
#pragma unroll
      for (int i = 0; i < 6 * TOUCH; ++i) sink += touch[i];
    }
    if (!PIPE) sample_point(f, rg.o, rg.dh, s_z[k], x, u);
    float* afr = SAVE ? frag_lane_base(act, (size_t)t, ACT_LD, s, g) : nullptr;       // fragment-order ACT tile (lrf_common.h)
    float xc[2] = {0.0f, 0.0f};
    const float w = s < cnt ? cw[(size_t)ray_c * S + j0 + s] : 0.0f;          // needed at the end of the tile
    int k_n = 0, j0_n = 0, cnt_n = 0, ray_n = ray_c;
    RayGeo rg_n = rg;
    if (t + 1 < t1) {                  // header of the next tile: consumed after this tile's gathers
      if (tile_walk2_seek(tw, toff, ncomp, t + 1)) rg_n = load_ray(rays, tw.ray);
      ray_n = tw.ray;
      j0_n = (t + 1 - tw.tile0) * ITEM;
      cnt_n = min(ITEM, tw.nc - j0_n);
      k_n = cidx[(size_t)tw.ray * S + j0_n + (s < cnt_n ? s : 0)];
    }
    if (VAR & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LRF_TICK(0);
    f32x4 fe[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    asm volatile("" : "+v"(fe[0]), "+v"(fe[1]));
    {
      float v[8];
      bf16x8 bh, bl;
      const AxisTaps at = axis_taps(f.pw[0], f.ph[0], f.ll[0], u);
      if (PIPE) plane_combine(raw0, v); else gather_app6_plane32<0>(f, at, g, v);
      if (SAVE) save_x_plane_with<0>(afr, v, xc, St16());
      split8(v, bh, bl);
      gemm_step<2>(img, IMGB_BAS / 128 + 0, 3, lane, bh, bl, fe);
      LRF_TICK(1);
      gather_app6_plane32<1>(f, at, g, v);
      if (SAVE) save_x_plane_with<1>(afr, v, xc, St16());
      split8(v, bh, bl);
      gemm_step<2>(img, IMGB_BAS / 128 + 1, 3, lane, bh, bl, fe);
      LRF_TICK(2);
      gather_app6_plane32<2>(f, at, g, v);
      if (SAVE) save_x_plane_with<2>(afr, v, xc, St16());
      split8(v, bh, bl);
      gemm_step<2>(img, IMGB_BAS / 128 + 2, 3, lane, bh, bl, fe);
      settle<2>(fe);
      LRF_TICK(3);
    }
    if (SAVE) {                        // feat (27) | 1 | 0 0 0 0
      float4 a4 = make_float4(fe[0][0], fe[0][1], fe[0][2], fe[0][3]);
      float4 b4 = make_float4(fe[1][0], fe[1][1], fe[1][2], fe[1][3]);
      if (g == 2) b4.w = 1.0f;                       // column 27 = bias column of the dW1 GEMM
      if (g == 3) b4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      *reinterpret_cast<float4*>(afr + 16 * ACT_FEAT) = a4;
      *reinterpret_cast<float4*>(afr + 16 * (ACT_FEAT + 16)) = b4;
    }
    f32x4 h1[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) h1[q] = *reinterpret_cast<const f32x4*>(&tail[TAIL_B1 + 16 * q + 4 * g]);
    if (VAR & 4) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      const float v[8] = {fe[0][0], fe[0][1], fe[0][2], fe[0][3], fe[1][0], fe[1][1], fe[1][2], fe[1][3]};
      bf16x8 bh, bl;
      if (VAR & 4) {
        split8_p<4>(v, bh, bl);
        gemm_step_q<4, 8>(img, IMGB_W1 / 128, 1, lane, bh, bl, h1);
      } else {
        split8(v, bh, bl);
        gemm_step<8>(img, IMGB_W1 / 128, 1, lane, bh, bl, h1);
        settle<8>(h1);
      }
    }
    if (SAVE) {
      uint32_t m1 = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          h1[q][r] = relu_i(h1[q][r]);
          m1 |= min(__float_as_uint(h1[q][r]), 1u) << (4 * q + r);         // relu output: +0 or positive
        }
        *reinterpret_cast<f32x4*>(afr + 16 * (ACT_H1 + 16 * q)) = h1[q];
      }
      *reinterpret_cast<float4*>(afr + 16 * (ACT_H1 + 128)) = make_float4(g == 0 ? 1.0f : 0.0f, 0.0f, 0.0f, 0.0f);
      relu_bits[((size_t)t * 2 + 0) * 64 + lane] = m1;
    }
    LRF_TICK(4);
    f32x4 h2[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) h2[q] = *reinterpret_cast<const f32x4*>(&tail[TAIL_B2 + 16 * q + 4 * g]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = relu_i(h1[2 * ks + (j >> 2)][j & 3]);
      bf16x8 bh, bl;
      if (VAR & 4) {
        split8_p<4>(v, bh, bl);
        gemm_step_q<4, 8>(img, IMGB_W2 / 128 + ks, 4, lane, bh, bl, h2);
      } else {
        split8(v, bh, bl);
        gemm_step<8>(img, IMGB_W2 / 128 + ks, 4, lane, bh, bl, h2);
      }
    }
    if (VAR & 4) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
    } else {
      settle<8>(h2);
    }
    if (SAVE) {
      uint32_t m2 = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          h2[q][r] = relu_i(h2[q][r]);
          m2 |= min(__float_as_uint(h2[q][r]), 1u) << (4 * q + r);
        }
        *reinterpret_cast<f32x4*>(afr + 16 * (ACT_H2 + 16 * q)) = h2[q];
      }
      *reinterpret_cast<float4*>(afr + 16 * (ACT_H2 + 128)) =
          g == 0 ? make_float4(rg.dh[0], rg.dh[1], rg.dh[2], 1.0f) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      relu_bits[((size_t)t * 2 + 1) * 64 + lane] = m2;
    }
    LRF_TICK(5);
    if (PIPE && t + 1 < t1) {          // next tile: position, then its plane-0 loads go out under the head phase
      sample_point(f, rg_n.o, rg_n.dh, s_z[k_n], x, u);
      raw0 = plane_issue<0>(f, u, g);
    }
    if (TOUCH > 0 && t + 1 < t1) {     // next tile: request its lines now, read them after the head phase
      float xn[3], un[3];
      sample_point(f, rg_n.o, rg_n.dh, s_z[k_n], xn, un);
      plane_touch<0>(f, un, g, touch);
      if (TOUCH > 1) plane_touch<1>(f, un, g, touch + 6 * (TOUCH > 1 ? 1 : 0));
      if (TOUCH > 2) plane_touch<2>(f, un, g, touch + 6 * (TOUCH > 2 ? 2 : 0));
    }
    f32x4 oc = {0, 0, 0, 0};
    if (VAR & 2) {
      float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float hv = relu_i(h2[q][r]);
          const float4 wv = *reinterpret_cast<const float4*>(&tail[TAIL_W3H + g * TAIL_W3H_GS + (q * 4 + r) * 4]);
          o0 += hv * wv.x; o1 += hv * wv.y; o2 += hv * wv.z;
        }
      o0 += __shfl_xor(o0, 16, 64); o1 += __shfl_xor(o1, 16, 64); o2 += __shfl_xor(o2, 16, 64);
      o0 += __shfl_xor(o0, 32, 64); o1 += __shfl_xor(o1, 32, 64); o2 += __shfl_xor(o2, 32, 64);
      oc[0] = o0; oc[1] = o1; oc[2] = o2;
    } else {
      asm volatile("" : "+v"(oc));
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = relu_i(h2[2 * ks + (j >> 2)][j & 3]);
        bf16x8 bh, bl;
        split8(v, bh, bl);
        gemm_step<1>(img, IMGB_W3F / 128 + ks, 1, lane, bh, bl, &oc);
      }
      settle<1>(&oc);
    }
    float vb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float4 wv = *reinterpret_cast<const float4*>(&tail[TAIL_W3V + 4 * c]);
      vb[c] = wv.w + wv.x * rg.dh[0] + wv.y * rg.dh[1] + wv.z * rg.dh[2];
    }
    const float wq = g != 0 ? 0.0f : w;                        // only the g = 0 lanes hold a colour
    const float sr = __frcp_rn(1.0f + __expf(-(oc[0] + vb[0])));
    const float sg = __frcp_rn(1.0f + __expf(-(oc[1] + vb[1])));
    const float sb = __frcp_rn(1.0f + __expf(-(oc[2] + vb[2])));
    if (SAVE && g == 0 && s < cnt) {                           // the sample's colour: d sigmoid = c (1 - c) in the backward
      float* cp = crgb + ((size_t)ray_c * S + j0 + s) * 3;
      cp[0] = sr; cp[1] = sg; cp[2] = sb;
    }
    float cr = wq * sr, cg = wq * sg, cb = wq * sb;
#pragma unroll
    for (int dd = 1; dd < 16; dd <<= 1) {
      cr += __shfl_xor(cr, dd, 64); cg += __shfl_xor(cg, dd, 64); cb += __shfl_xor(cb, dd, 64);
    }
    if (lane == 0) {
      float* pp = part + ((size_t)ray_c * pmax + j0 / ITEM) * 3;
      if (FUSE == 1) {                 // written through to the agent's coherence point: a boundary ray's partials are read from another XCD
        __hip_atomic_store(pp + 0, cr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 1, cg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 2, cb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        pp[0] = cr; pp[1] = cg; pp[2] = cb;
      }
    }
    k = k_n; j0 = j0_n; cnt = cnt_n; rg = rg_n; ray_c = ray_n;
    LRF_TICK(6);
    tk[7] += 1;
  }
  if (TIMED && f.dump && lane == 0) {
    unsigned long long* dp = reinterpret_cast<unsigned long long*>(f.dump) + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8;
    for (int i = 0; i < 8; ++i) dp[i] = tk[i];
  }
  if (TOUCH > 0 && sink == 1.2345678e30f) part[0] = sink;      // keeps the touch loads alive; never true
  if (FUSE) {                                                  // k_finalize, see above
    __shared__ int s_last;
    // Release of this workgroup's partials to the workgroup that finishes last (possibly on another XCD, whose L2 is
    // not coherent with this one's).  FUSE 1: the partials were stored write-through (agent scope) and the barrier
    // waits for their acknowledgements (s_waitcnt vmcnt(0)) -- no cache-wide operation.  FUSE 2: release fence
    // (L2 write-back).  FUSE 3: __threadfence(), which also INVALIDATES this XCD's L2 under the workgroups still
    // rendering (measured: colour kernel 160 -> 235 us).
    if (FUSE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (FUSE == 3) __threadfence();
    // every wave's partial stores acknowledged BEFORE the barrier: __syncthreads() alone emits s_waitcnt lgkmcnt(0)
    // only (a workgroup-scope release does not wait for vector stores when the workgroup shares one L1), and a
    // partial still in flight when the counter is bumped -- or when another wave of this workgroup sums the ray --
    // is a stale read: seen as 1-2 differing renders in 6000 on one (faster) box of the pool, none on six others
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int nb = gridDim.x, tid = threadIdx.x;
    const int lb = (nb % 8 == 0) ? (int)(blockIdx.x % 8) * (nb / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int T = s_toff[R];
    const long long waves = (long long)nb * (blockDim.x >> 6);
    const int wpb = blockDim.x >> 6;
    const int B0 = (int)((long long)lb * wpb * T / waves), B1 = (int)((long long)(lb + 1) * wpb * T / waves);
    const int ra = toff_lower_bound(s_toff, R, B0);
    const int rb = lb == nb - 1 ? R : toff_lower_bound(s_toff, R, B1);
    for (int r = ra + tid; r < rb; r += blockDim.x) {
      const int a = s_toff[r], b = s_toff[r + 1];
      if (b <= B1) finalize_ray<false>(r, b - a, pmax, flags, acc, part, rgb_out, acc_out);
    }
    if (tid == 0) s_last = __hip_atomic_fetch_add(f.ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nb - 1;
    __syncthreads();
    if (s_last) {                                              // every other workgroup has released its partials
      if (FUSE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (FUSE == 3) __threadfence();
      for (int ob = tid; ob < nb - 1; ob += blockDim.x) {      // the ray owned by logical block ob that crosses its upper boundary
        const int C0 = (int)((long long)ob * wpb * T / waves), C1 = (int)((long long)(ob + 1) * wpb * T / waves);
        const int r = toff_lower_bound(s_toff, R, C1) - 1;
        if (r >= 0) {
          const int a = s_toff[r], b = s_toff[r + 1];
          if (a >= C0 && b > C1) finalize_ray<true>(r, b - a, pmax, flags, acc, part, rgb_out, acc_out);
        }
      }
    }
  }
#undef LRF_TICK
}

}  // namespace lrf
