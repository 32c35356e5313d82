// Unit check of the run sums of k_scatter_fix (lrf_backward.inl: seg_run / seg_sum): random keys with runs, against a host loop.
// hipcc --offload-arch=gfx950 -O3 -o seg_sum seg_sum.hip && ./seg_sum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct SegRun { float m1, m2, m4, m8; bool tail; };
template <int D>
__device__ __forceinline__ int row_shr_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x110 + D, 0xf, 0xf, true); }
__device__ __forceinline__ SegRun seg_run(int key, bool valid, int lane) {
  if (!valid) key = -1 - lane;
  const int prev = row_shr_i<1>(key);
  const unsigned long long starts = __ballot((lane & 15) == 0 || prev != key);
  const int leader = 63 - __builtin_clzll(starts & (~0ull >> (63 - lane)));
  const int pos = lane - leader;
  SegRun r;
  r.m1 = pos >= 1 ? 1.0f : 0.0f; r.m2 = pos >= 2 ? 1.0f : 0.0f; r.m4 = pos >= 4 ? 1.0f : 0.0f; r.m8 = pos >= 8 ? 1.0f : 0.0f;
  r.tail = valid && (lane == 63 || ((starts >> (lane + 1)) & 1ull));
  return r;
}
__device__ __forceinline__ float seg_sum(float v, const SegRun& r) {
  // (a select around the DPP read lets the compiler predicate it: a DPP read from a lane that EXEC disables returns 0)
  v = fmaf(__int_as_float(row_shr_i<1>(__float_as_int(v))), r.m1, v);
  v = fmaf(__int_as_float(row_shr_i<2>(__float_as_int(v))), r.m2, v);
  v = fmaf(__int_as_float(row_shr_i<4>(__float_as_int(v))), r.m4, v);
  v = fmaf(__int_as_float(row_shr_i<8>(__float_as_int(v))), r.m8, v);
  return v;
}
// six values at once, one v_fmac_f32 with a DPP source per value and step (the compiler's form is v_mov_b32_dpp + v_fmac + hazard nops:
// 2.6 issue slots per value and step); the steps of one value are six instructions apart: no DPP read follows its VALU write closer
__device__ __forceinline__ void seg_sum6(float (&v)[6], const SegRun& r) {
#define LRF_STEP(N, M) \
  "v_fmac_f32_dpp %0, %0, " M " row_shr:" N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_fmac_f32_dpp %1, %1, " M " row_shr:" N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_fmac_f32_dpp %2, %2, " M " row_shr:" N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_fmac_f32_dpp %3, %3, " M " row_shr:" N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_fmac_f32_dpp %4, %4, " M " row_shr:" N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_fmac_f32_dpp %5, %5, " M " row_shr:" N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
  asm volatile("s_nop 1\n\t" LRF_STEP("1", "%6") LRF_STEP("2", "%7") LRF_STEP("4", "%8") LRF_STEP("8", "%9")
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5])
               : "v"(r.m1), "v"(r.m2), "v"(r.m4), "v"(r.m8));
#undef LRF_STEP
}
__global__ void k(const int* key, const float* val, float* out, int* tail, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
  const bool valid = i < n;
  const SegRun r = seg_run(valid ? key[i] : 0, valid, lane);
  float s = seg_sum(valid ? val[i] : 0.0f, r);
  {
    const float x = valid ? val[i] : 0.0f;
    float w[6] = {x, 2.0f * x, x + 1.0f, -x, 3.0f * x, x * x};
    seg_sum6(w, r);
    const float w0[6] = {seg_sum(x, r), seg_sum(2.0f * x, r), seg_sum(x + 1.0f, r), seg_sum(-x, r), seg_sum(3.0f * x, r), seg_sum(x * x, r)};
    for (int j = 0; j < 6; ++j) if (w[j] != w0[j]) s = 1e30f;        // (reported as a mismatch)
  }
  if (valid) { out[i] = s; tail[i] = r.tail; }
}
int main() {
  const int n = 64 * 50 - 7;
  std::vector<int> key(n); std::vector<float> val(n);
  srand(1);
  int cur = 5;
  for (int i = 0; i < n; ++i) { if (rand() % 4 == 0) cur = rand() % 1000; key[i] = cur; val[i] = (float)(rand() % 17 - 8); }
  int *dk, *dt; float *dv, *dout;
  hipMalloc(&dk, n * 4); hipMalloc(&dv, n * 4); hipMalloc(&dout, n * 4); hipMalloc(&dt, n * 4);
  hipMemcpy(dk, key.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dv, val.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dk, dv, dout, dt, n);
  std::vector<float> out(n); std::vector<int> tail(n);
  hipMemcpy(out.data(), dout, n * 4, hipMemcpyDeviceToHost); hipMemcpy(tail.data(), dt, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    float s = 0; int j = i;
    while (true) { s += val[j]; if (j % 16 == 0 || key[j - 1] != key[j]) break; --j; }
    const int t = (i == n - 1) || ((i + 1) % 16 == 0) || key[i + 1] != key[i];
    if (s != out[i] || t != tail[i]) { if (bad < 10) printf("i %d key %d want %g tail %d got %g tail %d\n", i, key[i], s, t, out[i], tail[i]); ++bad; }
  }
  printf("%s: %d mismatches of %d\n", bad ? "FAIL" : "ok", bad, n);
  return bad != 0;
}
