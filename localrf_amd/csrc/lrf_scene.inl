// Scene-level ends of the path (LocalTensorfs.forward, local_tensorfs.py:382-499):
//   k_scene_rays      pixel ids -> per-field rays            (local_tensorfs.py:23-29,397-431;
//                                                             utils/ray_utils.py:14-54)
//   k_scene_rays_bwd  gradients of poses / intrinsics        (what autograd derives for the above)
//   k_scene_blend     sum_rf w*rgb, sum_rf w*depth, per-view 3x3 exposure, clamp
//                                                            (local_tensorfs.py:468-474,481-499)
//   k_scene_blend_bwd gradients of exposure and of the per-field colours / depths
// All O(R) work: one fused launch each instead of ~40 small framework launches per call.
// View of ray r is r / per_view (the reference's repeat_interleave, local_tensorfs.py:437).
// Cited lines are relative to /root/reference/localTensoRF.
#pragma once

namespace lrf {

constexpr int SCENE_TPB = 256;

struct PixDir { float x, y, z; long long col, row; };

// ids2pixel + get_ray_directions_lean / _360 (same operation order as the torch expressions)
__device__ __forceinline__ PixDir pixel_dir(long long id, int W, int H, int fov360, float f, float cx, float cy) {
  PixDir p;
  p.col = id % W;
  p.row = (id / W) % H;
  if (fov360) {
    const float pi = 3.14159265358979323846f;
    const float phi = ((float)p.row + 0.5f) * pi / (float)H - pi / 2.0f;
    const float th = ((float)p.col + 0.5f) * 2.0f * pi / (float)W + pi;
    p.x = cosf(phi) * sinf(th);
    p.y = sinf(phi);
    p.z = cosf(phi) * cosf(th);
  } else {
    p.x = ((float)p.col + 0.5f - cx) / f;
    p.y = -((float)p.row + 0.5f - cy) / f;
    p.z = -1.0f;
  }
  return p;
}

__global__ __launch_bounds__(SCENE_TPB) void k_scene_rays(
    const long long* __restrict__ ray_ids, int R, int per_view, const float* __restrict__ c2w,
    const float* __restrict__ world2rf, int n_rf, const float* __restrict__ focal,
    const float* __restrict__ center, int W, int H, int fov360,
    float* __restrict__ rays, float* __restrict__ directions, long long* __restrict__ ij) {
  const int r = blockIdx.x * SCENE_TPB + threadIdx.x;
  if (r >= R) return;
  const float f = fov360 ? 1.0f : focal[0];
  const float cx = fov360 ? 0.0f : center[0], cy = fov360 ? 0.0f : center[1];
  const PixDir p = pixel_dir(ray_ids[r], W, H, fov360, f, cx, cy);
  const float* M = c2w + (size_t)(r / per_view) * 12;          // [3,4] row-major: R | t
  const float dx = M[0] * p.x + M[1] * p.y + M[2] * p.z;
  const float dy = M[4] * p.x + M[5] * p.y + M[6] * p.z;
  const float dz = M[8] * p.x + M[9] * p.y + M[10] * p.z;
  directions[3 * r + 0] = p.x; directions[3 * r + 1] = p.y; directions[3 * r + 2] = p.z;
  ij[2 * r + 0] = p.col; ij[2 * r + 1] = p.row;
  for (int k = 0; k < n_rf; ++k) {                              // fields differ by the origin shift only
    float* o = rays + ((size_t)k * R + r) * 6;
    o[0] = M[3] + world2rf[3 * k + 0];
    o[1] = M[7] + world2rf[3 * k + 1];
    o[2] = M[11] + world2rf[3 * k + 2];
    o[3] = dx; o[4] = dy; o[5] = dz;
  }
}

// block-wide sum of NV per-thread values; result valid in thread 0.  red: [SCENE_TPB/64][NV]
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float s = wave_sum(v[i]);
    if (lane == 0) red[wv * NV + i] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float s = 0.f;
      for (int w = 0; w < SCENE_TPB / 64; ++w) s += red[w * NV + i];
      v[i] = s;
    }
  }
  __syncthreads();
}

// One block per view.  g_c2w [V,3,4]; g_intr [V,3] = per-view partial (d focal, d cx, d cy);
// g_w2rf [V,n_rf,3] = per-view partial of the origin-shift gradient.
__global__ __launch_bounds__(SCENE_TPB) void k_scene_rays_bwd(
    const long long* __restrict__ ray_ids, int per_view, const float* __restrict__ c2w, int n_rf, int R,
    const float* __restrict__ focal, const float* __restrict__ center, int W, int H, int fov360,
    const float* __restrict__ g_rays, const float* __restrict__ g_dirs,
    float* __restrict__ g_c2w, float* __restrict__ g_intr, float* __restrict__ g_w2rf) {
  __shared__ float red[(SCENE_TPB / 64) * 15];
  const int v = blockIdx.x;
  const float f = fov360 ? 1.0f : focal[0];
  const float cx = fov360 ? 0.0f : center[0], cy = fov360 ? 0.0f : center[1];
  const float* M = c2w + (size_t)v * 12;
  float acc[15];                       // 0..8 dR (row-major), 9..11 dt, 12 dfocal, 13 dcx, 14 dcy
#pragma unroll
  for (int i = 0; i < 15; ++i) acc[i] = 0.f;
  for (int k = 0; k < n_rf; ++k) {     // per-field origin-shift partials first (they also sum into dt)
    float go[3] = {0.f, 0.f, 0.f};
    for (int q = threadIdx.x; q < per_view; q += SCENE_TPB) {
      const float* g = g_rays + ((size_t)k * R + (size_t)v * per_view + q) * 6;
      go[0] += g[0]; go[1] += g[1]; go[2] += g[2];
    }
    block_sum<3>(go, red);
    if (threadIdx.x == 0) {
      float* o = g_w2rf + ((size_t)v * n_rf + k) * 3;
      o[0] = go[0]; o[1] = go[1]; o[2] = go[2];
      acc[9] += go[0]; acc[10] += go[1]; acc[11] += go[2];
    }
  }
  for (int q = threadIdx.x; q < per_view; q += SCENE_TPB) {
    const size_t r = (size_t)v * per_view + q;
    const PixDir p = pixel_dir(ray_ids[r], W, H, fov360, f, cx, cy);
    float gd[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < n_rf; ++k) {
      const float* g = g_rays + ((size_t)k * R + r) * 6;
      gd[0] += g[3]; gd[1] += g[4]; gd[2] += g[5];
    }
    acc[0] += gd[0] * p.x; acc[1] += gd[0] * p.y; acc[2] += gd[0] * p.z;
    acc[3] += gd[1] * p.x; acc[4] += gd[1] * p.y; acc[5] += gd[1] * p.z;
    acc[6] += gd[2] * p.x; acc[7] += gd[2] * p.y; acc[8] += gd[2] * p.z;
    if (!fov360) {
      float gx = M[0] * gd[0] + M[4] * gd[1] + M[8] * gd[2];     // R^T g_d
      float gy = M[1] * gd[0] + M[5] * gd[1] + M[9] * gd[2];
      if (g_dirs) { gx += g_dirs[3 * r + 0]; gy += g_dirs[3 * r + 1]; }
      // x = (i+.5-cx)/f, y = -(j+.5-cy)/f
      acc[12] += -(gx * p.x + gy * p.y) / f;
      acc[13] += -gx / f;
      acc[14] += gy / f;
    }
  }
  float t3[3] = {acc[9], acc[10], acc[11]};      // thread 0 holds dt already reduced
  acc[9] = acc[10] = acc[11] = 0.f;
  block_sum<15>(acc, red);
  if (threadIdx.x == 0) {
    float* o = g_c2w + (size_t)v * 12;
    o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];  o[3] = t3[0];
    o[4] = acc[3]; o[5] = acc[4]; o[6] = acc[5];  o[7] = t3[1];
    o[8] = acc[6]; o[9] = acc[7]; o[10] = acc[8]; o[11] = t3[2];
    g_intr[3 * v + 0] = acc[12]; g_intr[3 * v + 1] = acc[13]; g_intr[3 * v + 2] = acc[14];
  }
}

// rgbs = clamp(E_v * sum_k bw[v,k] rgb_k, 0, 1); depth = sum_k bw[v,k] depth_k; `pre` keeps the blended
// colour before exposure for the backward pass (may be NULL).
__global__ __launch_bounds__(SCENE_TPB) void k_scene_blend(
    const float* __restrict__ rgb_f, const float* __restrict__ dep_f, const float* __restrict__ bw,
    const float* __restrict__ expo, int R, int per_view, int n_rf,
    float* __restrict__ rgbs, float* __restrict__ depth, float* __restrict__ pre) {
  const int r = blockIdx.x * SCENE_TPB + threadIdx.x;
  if (r >= R) return;
  const int v = r / per_view;
  float c0 = 0.f, c1 = 0.f, c2 = 0.f, d = 0.f;
  for (int k = 0; k < n_rf; ++k) {
    const float w = bw[(size_t)v * n_rf + k];
    const float* c = rgb_f + ((size_t)k * R + r) * 3;
    c0 = c0 + c[0] * w; c1 = c1 + c[1] * w; c2 = c2 + c[2] * w;
    d = d + dep_f[(size_t)k * R + r] * w;
  }
  if (pre) { pre[3 * r + 0] = c0; pre[3 * r + 1] = c1; pre[3 * r + 2] = c2; }
  float y0 = c0, y1 = c1, y2 = c2;
  if (expo) {
    const float* E = expo + (size_t)v * 9;
    y0 = E[0] * c0 + E[1] * c1 + E[2] * c2;
    y1 = E[3] * c0 + E[4] * c1 + E[5] * c2;
    y2 = E[6] * c0 + E[7] * c1 + E[8] * c2;
  }
  rgbs[3 * r + 0] = fminf(fmaxf(y0, 0.f), 1.f);
  rgbs[3 * r + 1] = fminf(fmaxf(y1, 0.f), 1.f);
  rgbs[3 * r + 2] = fminf(fmaxf(y2, 0.f), 1.f);
  depth[r] = d;
}

// One block per view.  clamp passes gradient where 0 <= y <= 1 (ATen clamp_backward).
__global__ __launch_bounds__(SCENE_TPB) void k_scene_blend_bwd(
    const float* __restrict__ g_rgbs, const float* __restrict__ g_depth, const float* __restrict__ pre,
    const float* __restrict__ bw, const float* __restrict__ expo, int R, int per_view, int n_rf,
    float* __restrict__ g_rgb_f, float* __restrict__ g_dep_f, float* __restrict__ g_expo) {
  __shared__ float red[(SCENE_TPB / 64) * 9];
  const int v = blockIdx.x;
  float E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (expo) for (int i = 0; i < 9; ++i) E[i] = expo[(size_t)v * 9 + i];
  float acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) acc[i] = 0.f;
  for (int q = threadIdx.x; q < per_view; q += SCENE_TPB) {
    const size_t r = (size_t)v * per_view + q;
    const float c0 = pre[3 * r + 0], c1 = pre[3 * r + 1], c2 = pre[3 * r + 2];
    const float y0 = E[0] * c0 + E[1] * c1 + E[2] * c2;
    const float y1 = E[3] * c0 + E[4] * c1 + E[5] * c2;
    const float y2 = E[6] * c0 + E[7] * c1 + E[8] * c2;
    const float g0 = (y0 >= 0.f && y0 <= 1.f) ? g_rgbs[3 * r + 0] : 0.f;
    const float g1 = (y1 >= 0.f && y1 <= 1.f) ? g_rgbs[3 * r + 1] : 0.f;
    const float g2 = (y2 >= 0.f && y2 <= 1.f) ? g_rgbs[3 * r + 2] : 0.f;
    acc[0] += g0 * c0; acc[1] += g0 * c1; acc[2] += g0 * c2;
    acc[3] += g1 * c0; acc[4] += g1 * c1; acc[5] += g1 * c2;
    acc[6] += g2 * c0; acc[7] += g2 * c1; acc[8] += g2 * c2;
    const float h0 = E[0] * g0 + E[3] * g1 + E[6] * g2;        // E^T g
    const float h1 = E[1] * g0 + E[4] * g1 + E[7] * g2;
    const float h2 = E[2] * g0 + E[5] * g1 + E[8] * g2;
    const float gd = g_depth ? g_depth[r] : 0.f;
    for (int k = 0; k < n_rf; ++k) {
      const float w = bw[(size_t)v * n_rf + k];
      float* o = g_rgb_f + ((size_t)k * R + r) * 3;
      o[0] = h0 * w; o[1] = h1 * w; o[2] = h2 * w;
      g_dep_f[(size_t)k * R + r] = gd * w;
    }
  }
  if (g_expo) {
    block_sum<9>(acc, red);
    if (threadIdx.x == 0)
      for (int i = 0; i < 9; ++i) g_expo[(size_t)v * 9 + i] = acc[i];
  }
}

}  // namespace lrf

extern "C" int lrf_scene_rays(const int64_t* ray_ids, int32_t R, int32_t per_view, const float* cam2world,
                              const float* world2rf, int32_t n_rf, const float* focal, const float* center,
                              int32_t W, int32_t H, int32_t fov360, float* rays, float* directions,
                              int64_t* ij, void* stream) {
  using namespace lrf;
  if (!ray_ids || !cam2world || !world2rf || !rays || !directions || !ij) return set_err("lrf_scene_rays: null argument");
  if (!fov360 && (!focal || !center)) return set_err("lrf_scene_rays: pinhole rays need focal and center");
  if (R < 0 || per_view <= 0 || n_rf <= 0 || W <= 0 || H <= 0) return set_err("lrf_scene_rays: bad sizes");
  if (R % per_view) return set_err("lrf_scene_rays: R must be a multiple of rays-per-view");
  if (!R) return 0;
  hipLaunchKernelGGL(k_scene_rays, dim3((R + SCENE_TPB - 1) / SCENE_TPB), dim3(SCENE_TPB), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const long long*>(ray_ids), R, per_view,
                     cam2world, world2rf, n_rf, focal, center, W, H, fov360, rays, directions,
                     reinterpret_cast<long long*>(ij));
  LRF_HIP(hipGetLastError());
  return 0;
}

extern "C" int lrf_scene_rays_bwd(const int64_t* ray_ids, int32_t R, int32_t per_view, const float* cam2world,
                                  int32_t n_rf, const float* focal, const float* center, int32_t W, int32_t H,
                                  int32_t fov360, const float* g_rays, const float* g_directions,
                                  float* g_cam2world, float* g_intr, float* g_world2rf, void* stream) {
  using namespace lrf;
  if (!ray_ids || !cam2world || !g_rays || !g_cam2world || !g_intr || !g_world2rf)
    return set_err("lrf_scene_rays_bwd: null argument");
  if (!fov360 && (!focal || !center)) return set_err("lrf_scene_rays_bwd: pinhole rays need focal and center");
  if (R < 0 || per_view <= 0 || n_rf <= 0 || R % per_view) return set_err("lrf_scene_rays_bwd: bad sizes");
  if (!R) return 0;
  hipLaunchKernelGGL(k_scene_rays_bwd, dim3(R / per_view), dim3(SCENE_TPB), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const long long*>(ray_ids), per_view, cam2world, n_rf, R, focal, center,
                     W, H, fov360, g_rays, g_directions, g_cam2world, g_intr, g_world2rf);
  LRF_HIP(hipGetLastError());
  return 0;
}

extern "C" int lrf_scene_blend(const float* rgb_f, const float* depth_f, const float* blend_w, const float* exposure,
                               int32_t R, int32_t per_view, int32_t n_rf, float* rgbs, float* depth, float* pre,
                               void* stream) {
  using namespace lrf;
  if (!rgb_f || !depth_f || !blend_w || !rgbs || !depth) return set_err("lrf_scene_blend: null argument");
  if (R < 0 || per_view <= 0 || n_rf <= 0 || R % per_view) return set_err("lrf_scene_blend: bad sizes");
  if (!R) return 0;
  hipLaunchKernelGGL(k_scene_blend, dim3((R + SCENE_TPB - 1) / SCENE_TPB), dim3(SCENE_TPB), 0,
                     reinterpret_cast<hipStream_t>(stream), rgb_f, depth_f, blend_w, exposure, R, per_view, n_rf,
                     rgbs, depth, pre);
  LRF_HIP(hipGetLastError());
  return 0;
}

extern "C" int lrf_scene_blend_bwd(const float* g_rgbs, const float* g_depth, const float* pre, const float* blend_w,
                                   const float* exposure, int32_t R, int32_t per_view, int32_t n_rf,
                                   float* g_rgb_f, float* g_depth_f, float* g_exposure, void* stream) {
  using namespace lrf;
  if (!g_rgbs || !pre || !blend_w || !g_rgb_f || !g_depth_f) return set_err("lrf_scene_blend_bwd: null argument");
  if (R < 0 || per_view <= 0 || n_rf <= 0 || R % per_view) return set_err("lrf_scene_blend_bwd: bad sizes");
  if (!R) return 0;
  hipLaunchKernelGGL(k_scene_blend_bwd, dim3(R / per_view), dim3(SCENE_TPB), 0, reinterpret_cast<hipStream_t>(stream),
                     g_rgbs, g_depth, pre, blend_w, exposure, R, per_view, n_rf, g_rgb_f, g_depth_f, g_exposure);
  LRF_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Pose assembly: LocalTensorfs.get_cam2world (local_tensorfs.py:292-299) + sixD_to_mtx
// (utils/utils.py:381-388) for up to LRF_POSE_MAX frames per launch.  The per-frame parameters are
// separate tensors (ParameterList), so their device pointers travel in the kernel-argument table.
namespace lrf {

struct PoseTable { const float* r[LRF_POSE_MAX]; const float* t[LRF_POSE_MAX]; };

struct Frame6D { float a1[3], a2[3], n1, b1[3], s, u[3], n2, b2[3], b3[3]; };
__device__ __forceinline__ Frame6D gram_schmidt(const float* r /*[3,2] row-major*/) {
  Frame6D F;
  for (int i = 0; i < 3; ++i) { F.a1[i] = r[2 * i]; F.a2[i] = r[2 * i + 1]; }
  F.n1 = sqrtf(F.a1[0] * F.a1[0] + F.a1[1] * F.a1[1] + F.a1[2] * F.a1[2]);
  for (int i = 0; i < 3; ++i) F.b1[i] = F.a1[i] / F.n1;
  F.s = F.b1[0] * F.a2[0] + F.b1[1] * F.a2[1] + F.b1[2] * F.a2[2];
  for (int i = 0; i < 3; ++i) F.u[i] = F.a2[i] - F.s * F.b1[i];
  F.n2 = sqrtf(F.u[0] * F.u[0] + F.u[1] * F.u[1] + F.u[2] * F.u[2]);
  for (int i = 0; i < 3; ++i) F.b2[i] = F.u[i] / F.n2;
  F.b3[0] = F.b1[1] * F.b2[2] - F.b1[2] * F.b2[1];
  F.b3[1] = F.b1[2] * F.b2[0] - F.b1[0] * F.b2[2];
  F.b3[2] = F.b1[0] * F.b2[1] - F.b1[1] * F.b2[0];
  return F;
}

// cross_views (V == 3 only): the reference calls torch.cross(b1, b2) without `dim`
// (utils/utils.py:386), which runs over the FIRST axis of size 3 -- for a stack of exactly three
// views that is the view axis: b3[v][c] = b1[v+1][c] b2[v+2][c] - b1[v+2][c] b2[v+1][c] (indices mod 3).
__global__ void k_pose_assemble(PoseTable tab, int V, int cross_views, float* __restrict__ c2w) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  Frame6D F = gram_schmidt(tab.r[v]);
  if (cross_views) {
    const Frame6D P = gram_schmidt(tab.r[(v + 1) % 3]), Q = gram_schmidt(tab.r[(v + 2) % 3]);
    for (int c = 0; c < 3; ++c) F.b3[c] = P.b1[c] * Q.b2[c] - Q.b1[c] * P.b2[c];
  }
  float* M = c2w + (size_t)v * 12;                       // columns (b1, b2, b3, t)
  for (int i = 0; i < 3; ++i) {
    M[4 * i + 0] = F.b1[i]; M[4 * i + 1] = F.b2[i]; M[4 * i + 2] = F.b3[i]; M[4 * i + 3] = tab.t[v][i];
  }
}

__global__ void k_pose_assemble_bwd(PoseTable tab, int V, int cross_views, const float* __restrict__ g_c2w,
                                    float* __restrict__ g_r, float* __restrict__ g_t) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const Frame6D F = gram_schmidt(tab.r[v]);
  const float* G = g_c2w + (size_t)v * 12;
  float g1[3], g2[3], g3[3];
  for (int i = 0; i < 3; ++i) { g1[i] = G[4 * i]; g2[i] = G[4 * i + 1]; g3[i] = G[4 * i + 2]; g_t[3 * v + i] = G[4 * i + 3]; }
  // b3 = b1 x b2
  float gb1[3] = {g1[0] + F.b2[1] * g3[2] - F.b2[2] * g3[1], g1[1] + F.b2[2] * g3[0] - F.b2[0] * g3[2],
                  g1[2] + F.b2[0] * g3[1] - F.b2[1] * g3[0]};
  float gb2[3] = {g2[0] + g3[1] * F.b1[2] - g3[2] * F.b1[1], g2[1] + g3[2] * F.b1[0] - g3[0] * F.b1[2],
                  g2[2] + g3[0] * F.b1[1] - g3[1] * F.b1[0]};
  if (cross_views) {                   // b3[w] = b1[w+1] * b2[w+2] - b1[w+2] * b2[w+1], element-wise over xyz
    const int vp = (v + 1) % 3, vm = (v + 2) % 3;
    const Frame6D P = gram_schmidt(tab.r[vp]), Q = gram_schmidt(tab.r[vm]);
    const float* Gp = g_c2w + (size_t)vp * 12;
    const float* Gm = g_c2w + (size_t)vm * 12;
    for (int c = 0; c < 3; ++c) {
      const float g3p = Gp[4 * c + 2], g3m = Gm[4 * c + 2];
      gb1[c] = g1[c] + g3m * P.b2[c] - g3p * Q.b2[c];
      gb2[c] = g2[c] + g3p * Q.b1[c] - g3m * P.b1[c];
    }
  }
  // b2 = u / |u|
  const float d2 = F.b2[0] * gb2[0] + F.b2[1] * gb2[1] + F.b2[2] * gb2[2];
  float gu[3];
  for (int i = 0; i < 3; ++i) gu[i] = (gb2[i] - F.b2[i] * d2) / F.n2;
  // u = a2 - s b1, s = b1 . a2
  const float gs = -(gu[0] * F.b1[0] + gu[1] * F.b1[1] + gu[2] * F.b1[2]);
  float ga2[3];
  for (int i = 0; i < 3; ++i) { ga2[i] = gu[i] + gs * F.b1[i]; gb1[i] += -F.s * gu[i] + gs * F.a2[i]; }
  // b1 = a1 / |a1|
  const float d1 = F.b1[0] * gb1[0] + F.b1[1] * gb1[1] + F.b1[2] * gb1[2];
  for (int i = 0; i < 3; ++i) {
    g_r[(size_t)v * 6 + 2 * i] = (gb1[i] - F.b1[i] * d1) / F.n1;
    g_r[(size_t)v * 6 + 2 * i + 1] = ga2[i];
  }
}

static int pose_table(const float* const* r, const float* const* t, int V, PoseTable& tab) {
  if (!r || V <= 0 || V > LRF_POSE_MAX) return 1;
  for (int i = 0; i < V; ++i) {
    if (!r[i] || (t && !t[i])) return 1;
    tab.r[i] = r[i]; tab.t[i] = t ? t[i] : nullptr;
  }
  return 0;
}

}  // namespace lrf

extern "C" int lrf_pose_assemble(const float* const* r6d, const float* const* trans, int32_t V, int32_t cross_views,
                                 float* cam2world, void* stream) {
  using namespace lrf;
  PoseTable tab;
  if (!trans || !cam2world || pose_table(r6d, trans, V, tab))
    return set_err("lrf_pose_assemble: null argument or V outside [1, LRF_POSE_MAX]");
  if (cross_views && V != 3) return set_err("lrf_pose_assemble: cross_views needs exactly 3 views");
  hipLaunchKernelGGL(k_pose_assemble, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), tab, V, cross_views, cam2world);
  LRF_HIP(hipGetLastError());
  return 0;
}

extern "C" int lrf_pose_assemble_bwd(const float* const* r6d, int32_t V, int32_t cross_views, const float* g_cam2world,
                                     float* g_r6d, float* g_trans, void* stream) {
  using namespace lrf;
  PoseTable tab;
  if (!g_cam2world || !g_r6d || !g_trans || pose_table(r6d, nullptr, V, tab))
    return set_err("lrf_pose_assemble_bwd: null argument or V outside [1, LRF_POSE_MAX]");
  if (cross_views && V != 3) return set_err("lrf_pose_assemble_bwd: cross_views needs exactly 3 views");
  hipLaunchKernelGGL(k_pose_assemble_bwd, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), tab, V, cross_views,
                     g_cam2world, g_r6d, g_trans);
  LRF_HIP(hipGetLastError());
  return 0;
}
