// lrf_tiles.inl -- tile walk, per-ray sum and 16-sample gather helpers shared by the colour kernels (included by
// lrf_render.hip).  A tile is a group of consecutive compact (shaded) samples of one ray: 16 for the exact-fp32 engine
// and the training kernels, 32 for k_shade3.
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

template <bool COHERENT>
__device__ __forceinline__ void finalize_ray(int ray, int nit, int pmax, uint32_t flags, const float* __restrict__ acc,
                                             const float* part, float* __restrict__ rgb, float* __restrict__ acc_out, int oray) {
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
  rgb[(size_t)oray * 3 + 0] = r; rgb[(size_t)oray * 3 + 1] = g; rgb[(size_t)oray * 3 + 2] = b;     // oray: the caller's index of slot `ray`
  if (acc_out) acc_out[oray] = a;
}

}  // namespace lrf
