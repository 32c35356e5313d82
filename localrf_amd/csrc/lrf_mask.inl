// lrf_mask.inl -- alpha-mask rebuild on the device, included by lrf_render.hip.
//
// Replaces TensorBase.getDenseAlpha + updateAlphaMask (models/tensorBase.py:501-536, which move the
// model to the CPU and loop over lattice slabs) with two launches and no host synchronisation:
//   k_dense_alpha   one thread per lattice point: position on the aabb lattice (:504-509), optional
//                   lookup in the CURRENT mask (compute_alpha :540-545), density feature through the
//                   same gather as the march, feature2density, alpha = 1 - exp(-sigma * length) (:556);
//                   written in the [Z][Y][X] order the reference reaches with transpose(0, 2) (:523)
//   k_alpha_pool    3x3x3 max-pool with implicit -inf padding (F.max_pool3d, :527) and the
//                   ">= alphaMask_thres" binarisation (:528-529)
#pragma once

namespace lrf {

__global__ __launch_bounds__(256) void k_dense_alpha(DField f, const float* __restrict__ lin_x,
                                                     const float* __restrict__ lin_y, const float* __restrict__ lin_z,
                                                     int gx, int gy, int gz, float length, uint32_t flags,
                                                     float* __restrict__ alpha /* [gz][gy][gx] */) {
  const size_t n = (size_t)gx * gy * gz;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int ix = (int)(i % gx), iy = (int)((i / gx) % gy), iz = (int)(i / ((size_t)gx * gy));
  const float t[3] = {lin_x[ix], lin_y[iy], lin_z[iz]};
  float x[3], u[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    // aabb[0] * (1 - t) + aabb[1] * t with the reference's roundings (no contraction into an fma)
    x[a] = __fadd_rn(__fmul_rn(f.lo[a], __fsub_rn(1.0f, t[a])), __fmul_rn(f.hi[a], t[a]));
  }
  bool valid = true;
  if (f.alpha_vol) valid = alpha_mask_sample(f, x[0], x[1], x[2]) > 0.0f;          // :540-542
#pragma unroll
  for (int a = 0; a < 3; ++a) u[a] = (x[a] - f.lo[a]) * f.inv[a] - 1.0f;           // normalize_coord :342-345
  float sigma = 0.0f;
  if (valid) sigma = feature2density(density_feature(f, u), f.density_shift, flags & LRF_FLAG_RELU_DENS);
  alpha[i] = 1.0f - expf(-sigma * length);                                          // :556
}

__global__ __launch_bounds__(256) void k_alpha_pool(const float* __restrict__ alpha, int gx, int gy, int gz,
                                                    float thres, float* __restrict__ out) {
  const size_t n = (size_t)gx * gy * gz;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int ix = (int)(i % gx), iy = (int)((i / gx) % gy), iz = (int)(i / ((size_t)gx * gy));
  float m = -INFINITY;
  for (int dz = -1; dz <= 1; ++dz)
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int x = ix + dx, y = iy + dy, z = iz + dz;
        if (x < 0 || x >= gx || y < 0 || y >= gy || z < 0 || z >= gz) continue;
        const float v = fminf(fmaxf(alpha[((size_t)z * gy + y) * gx + x], 0.0f), 1.0f);   // clamp(0, 1) :523
        m = fmaxf(m, v);
      }
  out[i] = m >= thres ? 1.0f : 0.0f;
}

}  // namespace lrf

extern "C" int lrf_dense_alpha(const LrfField* f, const float* lin_x, const float* lin_y, const float* lin_z,
                               int32_t gx, int32_t gy, int32_t gz, float length, uint32_t flags, float* alpha,
                               void* stream) {
  using namespace lrf;
  if (!f || !f->cache || !lin_x || !lin_y || !lin_z || !alpha) return set_err("lrf_dense_alpha: null argument");
  if (gx <= 0 || gy <= 0 || gz <= 0) return set_err("lrf_dense_alpha: bad lattice size");
  const size_t n = (size_t)gx * gy * gz;
  hipLaunchKernelGGL(k_dense_alpha, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     make_dfield(f), lin_x, lin_y, lin_z, gx, gy, gz, length, flags, alpha);
  LRF_HIP(hipGetLastError());
  return 0;
}

extern "C" int lrf_alpha_pool_threshold(const float* alpha, int32_t gx, int32_t gy, int32_t gz, float thres, float* out,
                                        void* stream) {
  using namespace lrf;
  if (!alpha || !out) return set_err("lrf_alpha_pool_threshold: null argument");
  if (gx <= 0 || gy <= 0 || gz <= 0) return set_err("lrf_alpha_pool_threshold: bad lattice size");
  const size_t n = (size_t)gx * gy * gz;
  hipLaunchKernelGGL(k_alpha_pool, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     alpha, gx, gy, gz, thres, out);
  LRF_HIP(hipGetLastError());
  return 0;
}
