// Optimiser step of the path's parameters: torch.optim.Adam as the reference
// configures it (local_tensorfs.py:88-97,146,245: betas (0.9, 0.99), no weight decay, no amsgrad)
// for up to LRF_ADAM_MAX tensors in ONE launch -- planes, lines, basis, MLP and, batched with
// them, the per-frame pose / exposure tensors that the reference steps one tiny optimiser at a
// time.  Same update as torch/optim/adam.py::_single_tensor_adam:
//   m <- m + (g - m)(1 - b1);  v <- v b2 + (1 - b2) g g;
//   p <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// (the two bias corrections are evaluated on the host in double, as torch does).
#pragma once

namespace lrf {

constexpr int ADAM_TPB = 256, ADAM_VEC = 4, ADAM_CHUNK = ADAM_TPB * ADAM_VEC * 4;   // 4096 elements per block

struct AdamTable {
  LrfAdamTensor t[LRF_ADAM_MAX];
  int first_block[LRF_ADAM_MAX + 1];   // prefix sum of per-tensor block counts
  int count;
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float b1, float b2, float eps,
                                      float step_size, float bc2_sqrt) {
  m = m + (g - m) * (1.0f - b1);
  v = v * b2 + (1.0f - b2) * g * g;
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = p - step_size * (m / denom);
}

// dev_scalars (lrf_adam_step_dev): [count][2] = {step_size, bc2_sqrt} per tensor in DEVICE memory, read at execution time -- the
// launch can sit in a captured hipGraph whose replays step with the learning rates / bias corrections (and skip the tensors,
// bc2_sqrt <= 0) the host wrote before each replay.
__global__ __launch_bounds__(ADAM_TPB) void k_adam_multi(AdamTable tab, float b1, float b2, float eps, const float* __restrict__ dev_scalars) {
  int ti = 0;
  while (ti + 1 < tab.count && (int)blockIdx.x >= tab.first_block[ti + 1]) ++ti;   // <= 64 uniform steps
  LrfAdamTensor T = tab.t[ti];
  if (dev_scalars) {
    T.step_size = dev_scalars[2 * ti];
    T.bc2_sqrt = dev_scalars[2 * ti + 1];
    if (!(T.bc2_sqrt > 0.0f)) return;                // not stepped this iteration (a view nobody sampled: torch skips .grad None)
  }
  const long long base = (long long)((int)blockIdx.x - tab.first_block[ti]) * ADAM_CHUNK;
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(T.p) | reinterpret_cast<uintptr_t>(T.g) |
                        reinterpret_cast<uintptr_t>(T.m) | reinterpret_cast<uintptr_t>(T.v)) & 15) == 0;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const long long i = base + ((long long)it * ADAM_TPB + threadIdx.x) * ADAM_VEC;
    if (i >= T.n) break;
    if (vec_ok && i + ADAM_VEC <= T.n) {
      float4 p = *reinterpret_cast<float4*>(T.p + i), m = *reinterpret_cast<float4*>(T.m + i),
             v = *reinterpret_cast<float4*>(T.v + i);
      const float4 g = *reinterpret_cast<const float4*>(T.g + i);
      adam1(p.x, g.x, m.x, v.x, b1, b2, eps, T.step_size, T.bc2_sqrt);
      adam1(p.y, g.y, m.y, v.y, b1, b2, eps, T.step_size, T.bc2_sqrt);
      adam1(p.z, g.z, m.z, v.z, b1, b2, eps, T.step_size, T.bc2_sqrt);
      adam1(p.w, g.w, m.w, v.w, b1, b2, eps, T.step_size, T.bc2_sqrt);
      *reinterpret_cast<float4*>(T.p + i) = p; *reinterpret_cast<float4*>(T.m + i) = m;
      *reinterpret_cast<float4*>(T.v + i) = v;
    } else {
      for (long long j = i; j < i + ADAM_VEC && j < T.n; ++j) {
        float p = T.p[j], m = T.m[j], v = T.v[j];
        adam1(p, T.g[j], m, v, b1, b2, eps, T.step_size, T.bc2_sqrt);
        T.p[j] = p; T.m[j] = m; T.v[j] = v;
      }
    }
  }
}

}  // namespace lrf

static int adam_step_impl(const LrfAdamTensor* tensors, int32_t count, const float* dev_scalars, float beta1, float beta2, float eps,
                          void* stream) {
  using namespace lrf;
  if (count < 0 || count > LRF_ADAM_MAX) return set_err("lrf_adam_step: count must be in [0, LRF_ADAM_MAX]");
  if (!count) return 0;
  if (!tensors) return set_err("lrf_adam_step: null argument");
  AdamTable tab;
  tab.count = count;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const LrfAdamTensor& t = tensors[i];
    if (!t.p || !t.g || !t.m || !t.v || t.n < 0) return set_err("lrf_adam_step: null tensor pointer or negative size");
    if (t.n > (int64_t)2000000000) return set_err("lrf_adam_step: tensor too large");
    tab.t[i] = t;
    tab.first_block[i] = blocks;
    blocks += (int)((t.n + ADAM_CHUNK - 1) / ADAM_CHUNK);
  }
  tab.first_block[count] = blocks;
  if (!blocks) return 0;
  hipLaunchKernelGGL(k_adam_multi, dim3(blocks), dim3(ADAM_TPB), 0, reinterpret_cast<hipStream_t>(stream), tab,
                     beta1, beta2, eps, dev_scalars);
  LRF_HIP(hipGetLastError());
  return 0;
}

extern "C" int lrf_adam_step(const LrfAdamTensor* tensors, int32_t count, float beta1, float beta2, float eps, void* stream) {
  return adam_step_impl(tensors, count, nullptr, beta1, beta2, eps, stream);
}

extern "C" int lrf_adam_step_dev(const LrfAdamTensor* tensors, int32_t count, const float* dev_scalars, float beta1, float beta2,
                                 float eps, void* stream) {
  if (!dev_scalars) return lrf::set_err("lrf_adam_step_dev: null scalar table");
  return adam_step_impl(tensors, count, dev_scalars, beta1, beta2, eps, stream);
}
