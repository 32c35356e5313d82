// Geometric losses around the render path and the grid upsample.
//
//   lrf_upsample_bilinear   TensorVMSplit.up_sampling_VM (tensoRF.py:198-221): F.interpolate(mode="bilinear",
//                           align_corners=True) of a plane [C,H,W] -> [C,H2,W2] or a line [C,L,1] -> [C,L2,1]
//   lrf_flow_loss_fwd/_bwd  the optical-flow loss of train.py:385-412 with utils/utils.py:15-48 (pts2px,
//                           inverse_pose, get_cam2cams, get_pred_flow): per view the forward / backward cam2cam
//                           transforms, reprojection of dir * depth, |pred - flow| * mask, values above the
//                           view's 0.9-quantile zeroed, mean
//   lrf_depth_loss_fwd/_bwd the monocular-depth loss of train.py:414-423 with compute_depth_loss
//                           (utils/utils.py:50-59): median / mean-abs-deviation normalisation of 1/depth and of the
//                           target per view, squared difference, values above the 0.8-quantile zeroed, mean
//
// One workgroup per view: the per-view statistics (torch.median, torch.quantile) come from a bitonic sort of the
// view's values in LDS.  The reference runs these as ~60 framework launches forward and ~100 backward per
// iteration on [V, n] tensors; here it is two launches each way.
#pragma once

namespace lrf {

constexpr int LOSS_NMAX = 4096;      // rays per view (train.py: batch_size 4096 over >= 1 view)
constexpr int LOSS_NT = 1024;

// ------------------------------------------------------------------------------------------- upsample
// ATen upsample_bilinear2d, align_corners=True: src = dst * (in - 1) / (out - 1), lambda1 = src - floor(src),
// value = l0h * (l0w * v00 + l1w * v01) + l1h * (l0w * v10 + l1w * v11)
__global__ void k_upsample_bilinear(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int W, int H2, int W2) {
  const long long n = (long long)C * H2 * W2;
  const float sh = H2 > 1 ? (float)(H - 1) / (float)(H2 - 1) : 0.0f;
  const float sw = W2 > 1 ? (float)(W - 1) / (float)(W2 - 1) : 0.0f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int x2 = (int)(i % W2), y2 = (int)((i / W2) % H2), c = (int)(i / ((long long)W2 * H2));
    const float fy = sh * (float)y2, fx = sw * (float)x2;
    const int y0 = (int)fy, x0 = (int)fx;
    const int yp = y0 < H - 1 ? 1 : 0, xp = x0 < W - 1 ? 1 : 0;
    const float ly1 = fminf(fmaxf(fy - (float)y0, 0.0f), 1.0f), lx1 = fminf(fmaxf(fx - (float)x0, 0.0f), 1.0f);
    const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    const float* p = src + ((size_t)c * H + y0) * W + x0;
    const float v00 = p[0], v01 = p[xp], v10 = p[(size_t)yp * W], v11 = p[(size_t)yp * W + xp];
    dst[i] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
  }
}

// ------------------------------------------------------------------------------------------- LDS sort
// ascending bitonic sort of N = pow2 keys (and payloads) in LDS, by the whole workgroup
template <bool PAYLOAD>
__device__ __forceinline__ void lds_bitonic_sort(float* key, int* val, int N) {
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const float a = key[i], b = key[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) {
            key[i] = b; key[l] = a;
            if (PAYLOAD) { const int t = val[i]; val[i] = val[l]; val[l] = t; }
          }
        }
      }
    }
  }
  __syncthreads();
}
__device__ __forceinline__ int pow2_ceil(int n) { int p = 1; while (p < n) p <<= 1; return p; }
// torch.quantile(x, q, interpolation="linear") of n sorted values: rank = q * (n - 1) in fp32,
// below.lerp(above, rank - floor(rank)) with ATen's two-sided lerp
__device__ __forceinline__ float sorted_quantile(const float* sorted, int n, float q) {
  const float rank = q * (float)(n - 1);
  const float lo = floorf(rank);
  const float w = rank - lo;
  const float a = sorted[(int)lo], b = sorted[(int)ceilf(rank)];
  return w < 0.5f ? a + w * (b - a) : b - (b - a) * (1.0f - w);
}
__device__ __forceinline__ float block_sum(float v, float* red /* [16] */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.0f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}

// ------------------------------------------------------------------------------------------- flow loss
struct Cam2Cam { float R[9], t[3]; };
// utils.py:22-35: world2cam = inverse_pose(cam2world[idx]) = (Ra^T, -(Ra^T ta)); cam2cam = (W Rb, W tb + tw)
__device__ __forceinline__ Cam2Cam make_cam2cam(const float* A /* cam2world[idx] [3,4] */, const float* B /* cam2world[i] */) {
  Cam2Cam c;
  float W[9], tw[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k) W[3 * r + k] = A[4 * k + r];
#pragma unroll
  for (int r = 0; r < 3; ++r) tw[r] = -(W[3 * r] * A[3] + W[3 * r + 1] * A[7] + W[3 * r + 2] * A[11]);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int k = 0; k < 3; ++k) c.R[3 * r + k] = W[3 * r] * B[k] + W[3 * r + 1] * B[4 + k] + W[3 * r + 2] * B[8 + k];
    c.t[r] = (W[3 * r] * B[3] + W[3 * r + 1] * B[7] + W[3 * r + 2] * B[11]) + tw[r];
  }
  return c;
}
struct FlowArgs {
  const float* c2w;        // [F,3,4]
  const int* frame;        // [V]: view id - starting frame id
  const int* fwd_off;      // [V]: forward mask zeroed for this view (train.py:396)
  const float* dirs;       // [V*n,3]  camera-space directions (LocalTensorfs.forward's third output)
  const float* depth;      // [V*n]
  const long long* ij;     // [V*n,2]  (col, row)
  const float* fwd_flow; const float* fwd_mask; const float* bwd_flow; const float* bwd_mask;
  const float* focal;      // device [1]
  const float* center;     // device [2]
  int F, V, n;
  float q;
};
// reprojected flow of one ray through one cam2cam (utils.py:15-21,43-48); optionally its pieces for the backward
struct Reproj { float fx, fy, qx, yq, zc; bool zpass; };
__device__ __forceinline__ Reproj reproject(const Cam2Cam& c, const float p[3], float f, float cx, float cy, float col, float row) {
  Reproj r;
  const float q0 = (c.R[0] * p[0] + c.R[1] * p[1] + c.R[2] * p[2]) + c.t[0];
  const float q1 = (c.R[3] * p[0] + c.R[4] * p[1] + c.R[5] * p[2]) + c.t[1];
  const float q2 = (c.R[6] * p[0] + c.R[7] * p[1] + c.R[8] * p[2]) + c.t[2];
  r.qx = q0; r.yq = -q1;
  const float zn = -q2;
  r.zpass = zn >= 1e-6f;                                    // clamp(min) passes the gradient where input >= min
  r.zc = fmaxf(zn, 1e-6f);
  r.fx = (r.qx / r.zc * f + cx - 0.5f) - col;
  r.fy = (r.yq / r.zc * f + cy - 0.5f) - row;
  return r;
}
__device__ __forceinline__ float sgn(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// arr_out[v*n + j] = flow_loss_arr with the entries above the view's q-quantile zeroed; vsum[v] = their sum
__global__ __launch_bounds__(LOSS_NT) void k_flow_loss_fwd(FlowArgs a, float* __restrict__ arr_out, float* __restrict__ vsum) {
  __shared__ float s_key[LOSS_NMAX];
  __shared__ float red[16];
  __shared__ Cam2Cam cc[2];
  __shared__ float s_thr;
  const int v = blockIdx.x, n = a.n;
  const int fi = a.frame[v];
  if (threadIdx.x < 2) {
    const int idx = min(max(fi + (threadIdx.x == 0 ? 1 : -1), 0), a.F - 1);
    cc[threadIdx.x] = make_cam2cam(a.c2w + (size_t)idx * 12, a.c2w + (size_t)fi * 12);
  }
  __syncthreads();
  const float f = a.focal[0], cx = a.center[0], cy = a.center[1];
  const bool last = a.fwd_off[v] != 0;                      // train.py:396: fwd_mask[view_ids == len(cam2world) - 1] = 0
  const int N = pow2_ceil(n);
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    float val = __builtin_inff();
    if (j < n) {
      const size_t r = (size_t)v * n + j;
      const float d = a.depth[r];
      const float p[3] = {a.dirs[r * 3] * d, a.dirs[r * 3 + 1] * d, a.dirs[r * 3 + 2] * d};
      const float col = (float)a.ij[r * 2], row = (float)a.ij[r * 2 + 1];
      const Reproj rf = reproject(cc[0], p, f, cx, cy, col, row), rb = reproject(cc[1], p, f, cx, cy, col, row);
      const float mf = last ? 0.0f : a.fwd_mask[r], mb = a.bwd_mask[r];
      val = (fabsf(rb.fx - a.bwd_flow[r * 2]) + fabsf(rb.fy - a.bwd_flow[r * 2 + 1])) * mb;
      val += (fabsf(rf.fx - a.fwd_flow[r * 2]) + fabsf(rf.fy - a.fwd_flow[r * 2 + 1])) * mf;
      arr_out[r] = val;
    }
    s_key[j] = val;
  }
  lds_bitonic_sort<false>(s_key, nullptr, N);
  if (threadIdx.x == 0) s_thr = sorted_quantile(s_key, n, a.q);
  __syncthreads();
  const float thr = s_thr;
  float acc = 0.0f;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const size_t r = (size_t)v * n + j;
    float val = arr_out[r];
    if (val > thr) { val = 0.0f; arr_out[r] = 0.0f; }
    acc += val;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) vsum[v] = acc;
}

// gradients of scale * sum(arr_out): depth, directions, per-view parts of d/d cam2world and d/d(focal, cx, cy)
__global__ __launch_bounds__(LOSS_NT) void k_flow_loss_bwd(FlowArgs a, const float* __restrict__ arr_out, const float* __restrict__ g_loss, float scale,
                                                          float* __restrict__ g_depth, float* __restrict__ g_dirs,
                                                          float* __restrict__ g_parts /* [V][3 roles: frame, fwd idx, bwd idx][12] */,
                                                          float* __restrict__ g_intr /* [V][3] */) {
  __shared__ float red[16];
  __shared__ Cam2Cam cc[2];
  __shared__ float s_tot[27];
  __shared__ float s_gb[2][12];
  const int v = blockIdx.x, n = a.n;
  const int fi = a.frame[v];
  const int idx_f = min(fi + 1, a.F - 1), idx_b = max(fi - 1, 0);
  if (threadIdx.x < 2) cc[threadIdx.x] = make_cam2cam(a.c2w + (size_t)(threadIdx.x == 0 ? idx_f : idx_b) * 12, a.c2w + (size_t)fi * 12);
  __syncthreads();
  const float f = a.focal[0], cx = a.center[0], cy = a.center[1];
  const bool last = a.fwd_off[v] != 0;
  const float gs = g_loss[0] * scale;
  float acc[27];                                            // fwd: dR[9] dt[3], bwd: dR[9] dt[3], df, dcx, dcy
#pragma unroll
  for (int i = 0; i < 27; ++i) acc[i] = 0.0f;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const size_t r = (size_t)v * n + j;
    float gp[3] = {0.0f, 0.0f, 0.0f};
    const float d = a.depth[r];
    const float dir[3] = {a.dirs[r * 3], a.dirs[r * 3 + 1], a.dirs[r * 3 + 2]};
    if (arr_out[r] > 0.0f) {                                // zeroed (or exactly zero) entries carry no gradient
      const float p[3] = {dir[0] * d, dir[1] * d, dir[2] * d};
      const float col = (float)a.ij[r * 2], row = (float)a.ij[r * 2 + 1];
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const float m = w == 0 ? (last ? 0.0f : a.fwd_mask[r]) : a.bwd_mask[r];
        const float* fl = (w == 0 ? a.fwd_flow : a.bwd_flow) + r * 2;
        const Reproj rp = reproject(cc[w], p, f, cx, cy, col, row);
        const float gpx = gs * m * sgn(rp.fx - fl[0]), gpy = gs * m * sgn(rp.fy - fl[1]);
        const float iz = 1.0f / rp.zc;
        const float gq0 = gpx * f * iz;
        const float gyq = gpy * f * iz;
        const float gz = -(gpx * rp.qx + gpy * rp.yq) * f * iz * iz;
        const float gq[3] = {gq0, -gyq, rp.zpass ? -gz : 0.0f};
        acc[24] += (gpx * rp.qx + gpy * rp.yq) * iz; acc[25] += gpx; acc[26] += gpy;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
          for (int k = 0; k < 3; ++k) acc[12 * w + 3 * i + k] += gq[i] * p[k];
          acc[12 * w + 9 + i] += gq[i];
          gp[0] += cc[w].R[3 * i] * gq[i]; gp[1] += cc[w].R[3 * i + 1] * gq[i]; gp[2] += cc[w].R[3 * i + 2] * gq[i];
        }
      }
    }
    g_depth[r] = gp[0] * dir[0] + gp[1] * dir[1] + gp[2] * dir[2];
    g_dirs[r * 3] = gp[0] * d; g_dirs[r * 3 + 1] = gp[1] * d; g_dirs[r * 3 + 2] = gp[2] * d;
  }
  for (int i = 0; i < 27; ++i) {
    const float s = block_sum(acc[i], red);
    if (threadIdx.x == 0) s_tot[i] = s;
  }
  __syncthreads();
  if (threadIdx.x < 2) {                                    // cam2cam -> cam2world: roles (frame fi = B, idx = A)
    const int w = threadIdx.x;
    const float* G = s_tot + 12 * w;                        // dR[9], dt[3]
    const float* A = a.c2w + (size_t)(w == 0 ? idx_f : idx_b) * 12;
    const float* B = a.c2w + (size_t)fi * 12;
    float W[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) W[3 * r + k] = A[4 * k + r];
    const float ta[3] = {A[3], A[7], A[11]}, tb[3] = {B[3], B[7], B[11]};
    const float* gt = G + 9;
    float gW[9], gB[12], gA[12];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k)                           // dW = G_R Rb^T + g_t tb^T - g_t ta^T
        gW[3 * r + k] = G[3 * r] * B[4 * k] + G[3 * r + 1] * B[4 * k + 1] + G[3 * r + 2] * B[4 * k + 2] + gt[r] * (tb[k] - ta[k]);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        gB[4 * r + k] = W[r] * G[k] + W[3 + r] * G[3 + k] + W[6 + r] * G[6 + k];     // dRb = W^T G_R
        gA[4 * r + k] = gW[3 * k + r];                                               // dRa = dW^T
      }
      const float wtg = W[r] * gt[0] + W[3 + r] * gt[1] + W[6 + r] * gt[2];         // (W^T g_t)_r
      gB[4 * r + 3] = wtg;
      gA[4 * r + 3] = -wtg;
    }
    float* out = g_parts + (size_t)v * 36;
    for (int i = 0; i < 12; ++i) { out[12 * (1 + w) + i] = gA[i]; s_gb[w][i] = gB[i]; }
  }
  __syncthreads();
  if (threadIdx.x < 12) g_parts[(size_t)v * 36 + threadIdx.x] = s_gb[0][threadIdx.x] + s_gb[1][threadIdx.x];   // role 0: frame fi, both cam2cams
  if (threadIdx.x == 0) { g_intr[v * 3] = s_tot[24]; g_intr[v * 3 + 1] = s_tot[25]; g_intr[v * 3 + 2] = s_tot[26]; }
}

// g_c2w[frame] = sum of the parts whose role names that frame, in a fixed order
__global__ void k_flow_pose_reduce(const float* __restrict__ g_parts, const int* __restrict__ frame, int V, int F, float* __restrict__ g_c2w) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F * 12) return;
  const int fr = i / 12, e = i % 12;
  float s = 0.0f;
  for (int v = 0; v < V; ++v) {
    const int fi = frame[v];
    const int ids[3] = {fi, min(fi + 1, F - 1), max(fi - 1, 0)};
#pragma unroll
    for (int r = 0; r < 3; ++r)
      if (ids[r] == fr) s += g_parts[(size_t)v * 36 + 12 * r + e];
  }
  g_c2w[i] = s;
}

// ------------------------------------------------------------------------------------------- depth loss
struct DepthArgs { const float* depth; const float* gt; int V, n; float q; };
// stats[v] = {t_d, s_d, t_gt, s_gt, thr, median index}
__global__ __launch_bounds__(LOSS_NT) void k_depth_loss_fwd(DepthArgs a, float* __restrict__ arr_out, float* __restrict__ stats, float* __restrict__ vsum) {
  __shared__ float s_key[LOSS_NMAX];
  __shared__ int s_val[LOSS_NMAX];
  __shared__ float red[16];
  __shared__ float s_st[6];
  const int v = blockIdx.x, n = a.n, N = pow2_ceil(n);
  const float* dp = a.depth + (size_t)v * n;
  const float* gp = a.gt + (size_t)v * n;
  // 1/clamp(depth, 1e-6): lower median (torch.median) and its index
  for (int j = threadIdx.x; j < N; j += blockDim.x) { s_key[j] = j < n ? 1.0f / fmaxf(dp[j], 1e-6f) : __builtin_inff(); s_val[j] = j; }
  lds_bitonic_sort<true>(s_key, s_val, N);
  if (threadIdx.x == 0) { s_st[0] = s_key[(n - 1) / 2]; s_st[5] = __int_as_float(s_val[(n - 1) / 2]); }
  __syncthreads();
  const float td = s_st[0];
  float acc = 0.0f;
  for (int j = threadIdx.x; j < n; j += blockDim.x) acc += fabsf(1.0f / fmaxf(dp[j], 1e-6f) - td);
  const float sd = block_sum(acc, red) / (float)n;
  __syncthreads();
  for (int j = threadIdx.x; j < N; j += blockDim.x) s_key[j] = j < n ? gp[j] : __builtin_inff();
  lds_bitonic_sort<false>(s_key, nullptr, N);
  if (threadIdx.x == 0) s_st[2] = s_key[(n - 1) / 2];
  __syncthreads();
  const float tg = s_st[2];
  acc = 0.0f;
  for (int j = threadIdx.x; j < n; j += blockDim.x) acc += fabsf(gp[j] - tg);
  const float sg = block_sum(acc, red) / (float)n;
  __syncthreads();
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    float val = __builtin_inff();
    if (j < n) {
      const float xn = (1.0f / fmaxf(dp[j], 1e-6f) - td) / sd, gn = (gp[j] - tg) / sg;
      val = (xn - gn) * (xn - gn);
      arr_out[(size_t)v * n + j] = val;
    }
    s_key[j] = val;
  }
  lds_bitonic_sort<false>(s_key, nullptr, N);
  if (threadIdx.x == 0) s_st[4] = sorted_quantile(s_key, n, a.q);
  __syncthreads();
  const float thr = s_st[4];
  acc = 0.0f;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const size_t r = (size_t)v * n + j;
    float val = arr_out[r];
    if (val > thr) { val = 0.0f; arr_out[r] = 0.0f; }
    acc += val;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) {
    vsum[v] = acc;
    float* st = stats + (size_t)v * 6;
    st[0] = td; st[1] = sd; st[2] = tg; st[3] = sg; st[4] = thr; st[5] = s_st[5];
  }
}
// xn_i = (x_i - x_m) / s, s = mean |x_k - x_m|, x = 1 / clamp(depth, 1e-6), m = median element:
//   dL/dx_j = g_j / s - [j == m] A / s - (B / n) (sign(x_j - x_m) - [j == m] sum_k sign(x_k - x_m)),
//   g_i = 2 (xn_i - gn_i) kept_i scale, A = sum g_i, B = sum g_i xn_i / s
__global__ __launch_bounds__(LOSS_NT) void k_depth_loss_bwd(DepthArgs a, const float* __restrict__ arr_out, const float* __restrict__ stats,
                                                           const float* __restrict__ g_loss, float scale, float* __restrict__ g_depth) {
  __shared__ float red[16];
  const int v = blockIdx.x, n = a.n;
  const float* dp = a.depth + (size_t)v * n;
  const float* gp = a.gt + (size_t)v * n;
  const float* st = stats + (size_t)v * 6;
  const float td = st[0], sd = st[1], tg = st[2], sg = st[3];
  const int m = __float_as_int(st[5]);
  const float gs = g_loss[0] * scale;
  float A = 0.0f, B = 0.0f, Sg = 0.0f;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const float x = 1.0f / fmaxf(dp[j], 1e-6f);
    const float xn = (x - td) / sd, gn = (gp[j] - tg) / sg;
    const float g = arr_out[(size_t)v * n + j] > 0.0f ? 2.0f * (xn - gn) * gs : 0.0f;
    A += g; B += g * xn; Sg += sgn(x - td);
  }
  A = block_sum(A, red); B = block_sum(B, red) / sd; Sg = block_sum(Sg, red);
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const float c = fmaxf(dp[j], 1e-6f);
    const float x = 1.0f / c;
    const float xn = (x - td) / sd, gn = (gp[j] - tg) / sg;
    const float g = arr_out[(size_t)v * n + j] > 0.0f ? 2.0f * (xn - gn) * gs : 0.0f;
    float dx = g / sd - (B / (float)n) * sgn(x - td);
    if (j == m) dx += -A / sd + (B / (float)n) * Sg;
    g_depth[(size_t)v * n + j] = dp[j] >= 1e-6f ? -dx / (c * c) : 0.0f;
  }
}

}  // namespace lrf

extern "C" int lrf_upsample_bilinear(const float* src, int32_t C, int32_t H, int32_t W, float* dst, int32_t H2, int32_t W2, void* stream) {
  using namespace lrf;
  if (!src || !dst || C <= 0 || H <= 0 || W <= 0 || H2 <= 0 || W2 <= 0) return set_err("lrf_upsample_bilinear: bad argument");
  const long long n = (long long)C * H2 * W2;
  const int blocks = (int)min((n + 255) / 256, (long long)65535 * 4);
  hipLaunchKernelGGL(k_upsample_bilinear, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, C, H, W, H2, W2);
  LRF_HIP(hipGetLastError());
  return 0;
}

static int flow_args_ok(const LrfFlowLoss* a) {
  return a && a->cam2world && a->frame && a->fwd_off && a->dirs && a->depth && a->ij && a->fwd_flow && a->fwd_mask && a->bwd_flow &&
         a->bwd_mask && a->focal && a->center && a->F > 0 && a->V > 0 && a->n > 0 && a->n <= LRF_LOSS_MAX_PER_VIEW;
}
static lrf::FlowArgs flow_args(const LrfFlowLoss* a) {
  lrf::FlowArgs f;
  f.c2w = a->cam2world; f.frame = a->frame; f.fwd_off = a->fwd_off; f.dirs = a->dirs; f.depth = a->depth;
  f.ij = reinterpret_cast<const long long*>(a->ij);
  f.fwd_flow = a->fwd_flow; f.fwd_mask = a->fwd_mask; f.bwd_flow = a->bwd_flow; f.bwd_mask = a->bwd_mask;
  f.focal = a->focal; f.center = a->center; f.F = a->F; f.V = a->V; f.n = a->n; f.q = a->quantile;
  return f;
}
extern "C" int lrf_flow_loss_fwd(const LrfFlowLoss* a, float* arr_out, float* view_sum, void* stream) {
  using namespace lrf;
  if (!flow_args_ok(a) || !arr_out || !view_sum) return set_err("lrf_flow_loss_fwd: bad argument (1 <= rays per view <= LRF_LOSS_MAX_PER_VIEW)");
  hipLaunchKernelGGL(k_flow_loss_fwd, dim3(a->V), dim3(LOSS_NT), 0, reinterpret_cast<hipStream_t>(stream), flow_args(a), arr_out, view_sum);
  LRF_HIP(hipGetLastError());
  return 0;
}
extern "C" int lrf_flow_loss_bwd(const LrfFlowLoss* a, const float* arr_out, const float* g_loss, float scale, float* g_depth,
                                 float* g_dirs, float* g_cam2world, float* g_intr, float* workspace, void* stream) {
  using namespace lrf;
  if (!flow_args_ok(a) || !arr_out || !g_loss || !g_depth || !g_dirs || !g_cam2world || !g_intr || !workspace)
    return set_err("lrf_flow_loss_bwd: bad argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(k_flow_loss_bwd, dim3(a->V), dim3(LOSS_NT), 0, st, flow_args(a), arr_out, g_loss, scale, g_depth, g_dirs, workspace, g_intr);
  hipLaunchKernelGGL(k_flow_pose_reduce, dim3((a->F * 12 + 255) / 256), dim3(256), 0, st, workspace, a->frame, a->V, a->F, g_cam2world);
  LRF_HIP(hipGetLastError());
  return 0;
}
extern "C" int lrf_depth_loss_fwd(const float* depth, const float* gt, int32_t V, int32_t n, float quantile, float* arr_out,
                                  float* stats, float* view_sum, void* stream) {
  using namespace lrf;
  if (!depth || !gt || !arr_out || !stats || !view_sum || V <= 0 || n <= 0 || n > LRF_LOSS_MAX_PER_VIEW)
    return set_err("lrf_depth_loss_fwd: bad argument (1 <= rays per view <= LRF_LOSS_MAX_PER_VIEW)");
  const DepthArgs a{depth, gt, V, n, quantile};
  hipLaunchKernelGGL(k_depth_loss_fwd, dim3(V), dim3(LOSS_NT), 0, reinterpret_cast<hipStream_t>(stream), a, arr_out, stats, view_sum);
  LRF_HIP(hipGetLastError());
  return 0;
}
extern "C" int lrf_depth_loss_bwd(const float* depth, const float* gt, int32_t V, int32_t n, const float* arr_out, const float* stats,
                                  const float* g_loss, float scale, float* g_depth, void* stream) {
  using namespace lrf;
  if (!depth || !gt || !arr_out || !stats || !g_loss || !g_depth || V <= 0 || n <= 0 || n > LRF_LOSS_MAX_PER_VIEW)
    return set_err("lrf_depth_loss_bwd: bad argument");
  const DepthArgs a{depth, gt, V, n, 0.0f};
  hipLaunchKernelGGL(k_depth_loss_bwd, dim3(V), dim3(LOSS_NT), 0, reinterpret_cast<hipStream_t>(stream), a, arr_out, stats, g_loss, scale, g_depth);
  LRF_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------- photometric loss, row gather
// The photometric loss of train.py:369-371, loss = mean_{i,c}(0.25 |rgb - target| w_i / mean(w)), as ONE launch each way
// (the reference -- and autograd through it -- is a dozen elementwise / reduction launches of 5 us each on a [4096, 3] batch:
// at the early grid sizes of the progressive schedule that is a tenth of the iteration).  w [R] or NULL (ones); w_mean
// device [1] or NULL (the mean of w over THIS batch; under ray sharding the caller passes the batch-global mean,
// localrf_amd.dist.global_mean).  One workgroup, fixed summation order: the value does not depend on anything but the inputs.
// aux[0] = 0.25 / (mean(w) 3 R) is what the backward multiplies sign(rgb - target) w_i with.
namespace lrf {
__global__ __launch_bounds__(1024) void k_photo_loss_fwd(const float* __restrict__ rgb, const float* __restrict__ target, const float* __restrict__ w,
                                                         const float* __restrict__ w_mean, int R, float* __restrict__ loss, float* __restrict__ aux) {
  __shared__ float s_red[1024];
  __shared__ float s_wm;
  const int t = threadIdx.x;
  if (w && !w_mean) {
    float a = 0.0f;
    for (int i = t; i < R; i += 1024) a += w[i];
    s_red[t] = a;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) { if (t < s) s_red[t] += s_red[t + s]; __syncthreads(); }
    if (t == 0) s_wm = s_red[0] / (float)R;
    __syncthreads();
  } else if (t == 0) {
    s_wm = w_mean ? w_mean[0] : 1.0f;
  }
  __syncthreads();
  float a = 0.0f;
  for (int i = t; i < 3 * R; i += 1024) a += fabsf(rgb[i] - target[i]) * (w ? w[i / 3] : 1.0f);
  __syncthreads();
  s_red[t] = a;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) { if (t < s) s_red[t] += s_red[t + s]; __syncthreads(); }
  if (t == 0) {
    const float sc = 0.25f / (s_wm * (float)(3 * R));
    aux[0] = sc;
    loss[0] = s_red[0] * sc;
  }
}
__global__ void k_photo_loss_bwd(const float* __restrict__ rgb, const float* __restrict__ target, const float* __restrict__ w,
                                 const float* __restrict__ aux, const float* __restrict__ g_loss, int R, float* __restrict__ g_rgb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * R) return;
  const float d = rgb[i] - target[i];
  const float sg = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);       // torch's abs backward: sign(x), 0 at 0
  g_rgb[i] = g_loss[0] * aux[0] * (w ? w[i / 3] : 1.0f) * sg;
}
// out[v, :] = src[idx[v], :] (K floats per row) and its backward g_src[f, :] = sum over the v with idx[v] == f of g_out[v, :],
// summed in v order by one thread per output element: no atomics, no zero fill, the same bits every time.  For the per-view rows a
// training batch picks out of per-frame tables (poses [F,12], exposures [N,9]): torch's index_select / index backward are an
// index_add / a sort-based index_put of 8 launches and 35 us for sixteen rows.
__global__ void k_rows_gather(const float* __restrict__ src, const long long* __restrict__ idx, int V, int K, int F, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V * K) return;
  long long f = idx[i / K];
  if (f < 0) f += F;
  out[i] = (f >= 0 && f < F) ? src[f * K + i % K] : 0.0f;
}
__global__ void k_rows_gather_bwd(const float* __restrict__ g_out, const long long* __restrict__ idx, int V, int K, int F, float* __restrict__ g_src) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F * K) return;
  const int f = i / K, k = i % K;
  float a = 0.0f;
  for (int v = 0; v < V; ++v) {
    long long q = idx[v];
    if (q < 0) q += F;
    if (q == f) a += g_out[v * K + k];
  }
  g_src[i] = a;
}
}  // namespace lrf

extern "C" int lrf_photo_loss_fwd(const float* rgb, const float* target, const float* w, const float* w_mean, int32_t R, float* loss, float* aux, void* stream) {
  using namespace lrf;
  if (!rgb || !target || !loss || !aux || R <= 0) return set_err("lrf_photo_loss_fwd: bad argument");
  hipLaunchKernelGGL(k_photo_loss_fwd, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), rgb, target, w, w_mean, R, loss, aux);
  LRF_HIP(hipGetLastError());
  return 0;
}
extern "C" int lrf_photo_loss_bwd(const float* rgb, const float* target, const float* w, const float* aux, const float* g_loss, int32_t R, float* g_rgb, void* stream) {
  using namespace lrf;
  if (!rgb || !target || !aux || !g_loss || !g_rgb || R <= 0) return set_err("lrf_photo_loss_bwd: bad argument");
  hipLaunchKernelGGL(k_photo_loss_bwd, dim3((3 * R + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), rgb, target, w, aux, g_loss, R, g_rgb);
  LRF_HIP(hipGetLastError());
  return 0;
}
extern "C" int lrf_rows_gather(const float* src, const int64_t* idx, int32_t V, int32_t K, int32_t F, float* out, void* stream) {
  using namespace lrf;
  if (!src || !idx || !out || V <= 0 || K <= 0 || F <= 0) return set_err("lrf_rows_gather: bad argument");
  hipLaunchKernelGGL(k_rows_gather, dim3((V * K + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, reinterpret_cast<const long long*>(idx), V, K, F, out);
  LRF_HIP(hipGetLastError());
  return 0;
}
extern "C" int lrf_rows_gather_bwd(const float* g_out, const int64_t* idx, int32_t V, int32_t K, int32_t F, float* g_src, void* stream) {
  using namespace lrf;
  if (!g_out || !idx || !g_src || V <= 0 || K <= 0 || F <= 0) return set_err("lrf_rows_gather_bwd: bad argument");
  hipLaunchKernelGGL(k_rows_gather_bwd, dim3((F * K + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g_out, reinterpret_cast<const long long*>(idx), V, K, F, g_src);
  LRF_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- batch assembly, loss assembly (round 6)
// What surrounds the three loss kernels in an iteration of train.py:352-437 is a few dozen one-line tensor expressions: the
// (view, pixel) indexing of the images, flows and inverse depths, the flow masks, the schedule weights, the sums.  Through
// ATen each is a launch (or three, with its backward): 45 of the 94 kernels of a captured iteration of the regularised
// phase, ~0.2 ms of its 1.35 (profiles/r17_graph_iteration_timeline_64.md).  Two kernels take their place.
namespace lrf {
__global__ __launch_bounds__(256) void k_batch_gather(LrfBatchGather a, float* __restrict__ target, float* __restrict__ fwd_flow,
                                                      float* __restrict__ fwd_mask, float* __restrict__ bwd_flow,
                                                      float* __restrict__ bwd_mask, float* __restrict__ invdepth) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.V * a.n) return;
  long long v = reinterpret_cast<const long long*>(a.view_ids)[i / a.n];
  if (v < 0) v += a.n_images;                                 // (negative ids count from the end, as tensor indexing does)
  const long long px = reinterpret_cast<const long long*>(a.pix)[i];
  const size_t q = (size_t)v * a.HW + (size_t)px;
  if (a.images && target) { target[3 * (size_t)i] = a.images[3 * q]; target[3 * (size_t)i + 1] = a.images[3 * q + 1]; target[3 * (size_t)i + 2] = a.images[3 * q + 2]; }
  if (a.fwd_flow && fwd_flow) { fwd_flow[2 * (size_t)i] = a.fwd_flow[2 * q]; fwd_flow[2 * (size_t)i + 1] = a.fwd_flow[2 * q + 1]; }
  if (a.bwd_flow && bwd_flow) { bwd_flow[2 * (size_t)i] = a.bwd_flow[2 * q]; bwd_flow[2 * (size_t)i + 1] = a.bwd_flow[2 * q + 1]; }
  if (a.invdepths && invdepth) invdepth[i] = a.invdepths[q];
  if (fwd_mask) fwd_mask[i] = v < a.n_images - 1 ? 1.0f : 0.0f;
  if (bwd_mask) bwd_mask[i] = v > 0 ? 1.0f : 0.0f;
}
// total = sum_k w_k sum_j x_k[j],  w_k = a_k + b_k s: one wave, term k's values summed in index order by lane k
__global__ __launch_bounds__(64) void k_loss_combine_fwd(LrfLossTerms t, float* __restrict__ total, float* __restrict__ w_out) {
  const int k = threadIdx.x;
  float v = 0.0f, w = 0.0f;
  if (k < t.count) {
    w = t.a[k] + t.b[k] * (t.s ? t.s[0] : 0.0f);
    float acc = 0.0f;
    for (int j = 0; j < t.n[k]; ++j) acc += t.x[k][j];
    v = acc * w;
    w_out[k] = w;
  }
  float tot = 0.0f;                                           // ordered: term 0 first
  for (int j = 0; j < t.count; ++j) tot += __shfl(v, j, 64);
  if (k == 0) total[0] = tot;
}
__global__ __launch_bounds__(64) void k_loss_combine_bwd(const float* __restrict__ w, const float* __restrict__ g_total, int count, float* __restrict__ g) {
  if ((int)threadIdx.x < count) g[threadIdx.x] = g_total[0] * w[threadIdx.x];
}
}  // namespace lrf

extern "C" int lrf_batch_gather(const LrfBatchGather* a, float* target, float* fwd_flow, float* fwd_mask, float* bwd_flow,
                                float* bwd_mask, float* invdepth, void* stream) {
  using namespace lrf;
  if (!a || !a->view_ids || !a->pix || a->V <= 0 || a->n <= 0 || a->HW <= 0 || a->n_images <= 0) return set_err("lrf_batch_gather: bad argument");
  hipLaunchKernelGGL(k_batch_gather, dim3((a->V * a->n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a,
                     target, fwd_flow, fwd_mask, bwd_flow, bwd_mask, invdepth);
  LRF_HIP(hipGetLastError());
  return 0;
}
extern "C" int lrf_loss_combine_fwd(const LrfLossTerms* t, float* total, float* w_out, void* stream) {
  using namespace lrf;
  if (!t || !total || !w_out || t->count <= 0 || t->count > LRF_LOSS_TERMS_MAX) return set_err("lrf_loss_combine_fwd: bad argument");
  for (int k = 0; k < t->count; ++k) if (!t->x[k] || t->n[k] <= 0) return set_err("lrf_loss_combine_fwd: null or empty term");
  hipLaunchKernelGGL(k_loss_combine_fwd, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), *t, total, w_out);
  LRF_HIP(hipGetLastError());
  return 0;
}
extern "C" int lrf_loss_combine_bwd(const float* w, const float* g_total, int32_t count, float* g, void* stream) {
  using namespace lrf;
  if (!w || !g_total || !g || count <= 0 || count > LRF_LOSS_TERMS_MAX) return set_err("lrf_loss_combine_bwd: bad argument");
  hipLaunchKernelGGL(k_loss_combine_bwd, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), w, g_total, count, g);
  LRF_HIP(hipGetLastError());
  return 0;
}
