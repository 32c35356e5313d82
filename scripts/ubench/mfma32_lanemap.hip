// Lane maps of v_mfma_f32_32x32x16_bf16 on gfx950, read off the hardware: A and B are filled with values that encode
// (row, k) / (k, col) under the ASSUMED maps of tests/test_mfma_chain_model.py::mfma_32x32x16
//   A: lane (n = l & 31, h = l >> 5) holds A[n][8 h + j];  B: lane holds B[8 h + j][n];
//   D: register r of lane (n, h) = D[8 (r >> 2) + 4 h + (r & 3)][n]
// and the result is compared with a host matmul.  Prints OK or the first mismatches; also times the instruction
// against v_mfma_f32_16x16x32_bf16 (issue rate per SIMD).  First thing to run before building a 32-sample colour kernel.
//   hipcc --offload-arch=gfx950 -O3 -o mfma32_lanemap mfma32_lanemap.hip && ./mfma32_lanemap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_map(const float* A /* [32][16] */, const float* B /* [16][32] */, float* D /* [32][32] */) {
  const int l = threadIdx.x, n = l & 31, h = l >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[n * 16 + 8 * h + j]; b[j] = (__bf16)B[(8 * h + j) * 32 + n]; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[(8 * (r >> 2) + 4 * h + (r & 3)) * 32 + n] = c[r];
}

template <int WHICH>
__global__ __launch_bounds__(256) void k_rate(int iters, float* out) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(1.0f + j * 0.01f); }
  f32x16 c0 = {0}, c1 = {0};
  f32x4 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
  for (int it = 0; it < iters; ++it) {
    if (WHICH == 0) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    } else {
      d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d2, 0, 0, 0);
      d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d3, 0, 0, 0);
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
  for (int r = 0; r < 4; ++r) s += d0[r] + d1[r] + d2[r] + d3[r];
  if (s == 12345.678f) out[0] = s;
}

int main() {
  float hA[32 * 16], hB[16 * 32], hD[32 * 32], ref[32 * 32];
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) hA[i * 16 + k] = (float)(i + 1) + (k % 4) * 0.25f;      // exact in bf16
  for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) hB[k * 32 + j] = (k == (j % 16)) ? 1.0f : ((k == 3) ? 0.5f : 0.0f);
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += (double)hA[i * 16 + k] * hB[k * 32 + j]; ref[i * 32 + j] = (float)s; }
  float *dA, *dB, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_map, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32 * 32; ++i) if (fabsf(hD[i] - ref[i]) > 1e-3f) { if (bad++ < 8) printf("mismatch D[%d][%d] = %g, expected %g\n", i / 32, i % 32, hD[i], ref[i]); }
  printf("lane maps of v_mfma_f32_32x32x16_bf16: %s (%d mismatches)\n", bad ? "NOT as assumed" : "OK", bad);
  for (int which = 0; which < 2; ++which) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 4;
    if (which == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, 10, dD); else hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, 10, dD);
    hipEventRecord(e0);
    if (which == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, iters, dD); else hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, iters, dD);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double macs = (double)blocks * 4 * iters * (which == 0 ? 2.0 * 32 * 32 * 16 : 4.0 * 16 * 16 * 32);
    printf("%s: %.3f ms, %.1f TFLOP/s dense\n", which == 0 ? "32x32x16 bf16" : "16x16x32 bf16", ms, 2 * macs / ms * 1e-9);
  }
  return bad != 0;
}
