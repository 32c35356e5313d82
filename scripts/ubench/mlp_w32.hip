// The colour network on v_mfma_f32_32x32x16_bf16 with 32 samples per wave, split-bf16 (hi + lo, three terms), as a
// stand-alone kernel without gathers: does the layout of tests/test_mfma_chain_model.py::chain_w32 run on the hardware,
// and what does the MFMA + LDS-fragment part of a 32-sample colour kernel cost next to the 16-sample one (library:
// k_mlp, 86-88 us for the 725 K shaded samples of the benchmark batch)?  Compiler-scheduled builtins only: a kernel
// that issues no gathers was deterministic with them (docs/GFX950_FINDINGS.md finding 9a).
//   hipcc --offload-arch=gfx950 -O3 -o mlp_w32 mlp_w32.hip && ./mlp_w32 [rows]
// Lane l = (n = l & 31: sample, h = l >> 5: K half).  Fragment f, half (hi, lo), lane: 8 bf16 = A[n][8 h + j].
//   f = ks            (0..4)   basis: A[n][slot] = basis[n][chan(h, 8 ks + j)]
//   f = 5 + 2 m + q            W1:    A[n][slot] = W1[32 m + n][unit(0, q, h, j)]
//   f = 13 + 8 m + 2 m0 + q    W2:    A[n][slot] = W2[32 m + n][unit(m0, q, h, j)]
// unit(m0, q, h, j) = 32 m0 + 16 q + 8 (j >> 2) + 4 h + (j & 3) = the unit D register 8 q + j of tile m0 holds in lane half h.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NFRAG = 45, IMG_U4 = NFRAG * 2 * 64;            // uint4: 92,160 B
constexpr int T_B1 = 0, T_B2 = 128, T_W3 = 256, T_B3 = 256 + 3 * 132, T_FLOATS = 672;   // fp32 tail: w3 rows padded to 132

__host__ __device__ inline int unit_of(int m0, int q, int h, int j) { return 32 * m0 + 16 * q + 8 * (j >> 2) + 4 * h + (j & 3); }
__host__ __device__ inline int chan_of(int h, int v) {          // gathered value v (0..39) of lane half h -> channel or -1
  if (v >= 36) return -1;
  const int pl = v / 12, w = v % 12;
  return pl * 24 + 6 * (2 * h + w / 6) + w % 6;
}

__device__ __forceinline__ void split8(const float v[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) { const __bf16 hh = (__bf16)v[j]; hi[j] = hh; lo[j] = (__bf16)(v[j] - (float)hh); }
}
__device__ __forceinline__ f32x16 mma3(const uint4* img, int f, int lane, bf16x8 bh, bf16x8 bl, f32x16 acc) {
  const bf16x8 ah = __builtin_bit_cast(bf16x8, img[(f * 2 + 0) * 64 + lane]);
  const bf16x8 al = __builtin_bit_cast(bf16x8, img[(f * 2 + 1) * 64 + lane]);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  return acc;
}

// MODE 0: from the 36 gathered products per lane (basis step included); MODE 1: from feat (16 registers per lane), the
// work k_mlp does.  in: [group][lane][40] floats (MODE 0) or [group][lane][16] (MODE 1); out: [row][4]
template <int MODE>
__global__ __launch_bounds__(512) void k_mlp_w32(const uint4* __restrict__ gimg, const float* __restrict__ gtail,
                                                 const float* __restrict__ in, int groups, float dhx, float dhy, float dhz,
                                                 float* __restrict__ out) {
  extern __shared__ uint4 s_img[];
  float* tail = reinterpret_cast<float*>(s_img + IMG_U4);
  for (int i = threadIdx.x; i < IMG_U4; i += blockDim.x) s_img[i] = gimg[i];
  for (int i = threadIdx.x; i < T_FLOATS; i += blockDim.x) tail[i] = gtail[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, h = lane >> 5;
  const int nw = gridDim.x * (blockDim.x >> 6);
  for (int grp = blockIdx.x * (blockDim.x >> 6) + wave; grp < groups; grp += nw) {
    asm volatile("" ::: "memory");                               // keep the fragment reads inside the loop
    f32x16 fe;
    if (MODE == 0) {
      float v[40];
      const float4* src = reinterpret_cast<const float4*>(in + ((size_t)grp * 64 + lane) * 40);
#pragma unroll
      for (int i = 0; i < 10; ++i) { const float4 t = src[i]; v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w; }
#pragma unroll
      for (int r = 0; r < 16; ++r) fe[r] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) {
        bf16x8 bh, bl;
        split8(v + 8 * ks, bh, bl);
        fe = mma3(s_img, ks, lane, bh, bl, fe);
      }
    } else {
      const float4* src = reinterpret_cast<const float4*>(in + ((size_t)grp * 64 + lane) * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float4 t = src[i]; fe[4 * i] = t.x; fe[4 * i + 1] = t.y; fe[4 * i + 2] = t.z; fe[4 * i + 3] = t.w; }
    }
    f32x16 h1[4], h2[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) h1[m][r] = tail[T_B1 + 32 * m + 8 * (r >> 2) + 4 * h + (r & 3)];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fe[8 * q + j];
      bf16x8 bh, bl;
      split8(v, bh, bl);
#pragma unroll
      for (int m = 0; m < 4; ++m) h1[m] = mma3(s_img, 5 + 2 * m + q, lane, bh, bl, h1[m]);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) h2[m][r] = tail[T_B2 + 32 * m + 8 * (r >> 2) + 4 * h + (r & 3)];
    if (MODE == 2) {
      // W2 with the next K-step's eight fragment halves requested from LDS before the current step's twelve MFMAs
      // (two register buffers), and the three terms interleaved over the four accumulators (no MFMA waits on the one
      // issued just before it)
      bf16x8 Ah[2][4], Al[2][4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        Ah[0][m] = __builtin_bit_cast(bf16x8, s_img[((13 + 8 * m) * 2 + 0) * 64 + lane]);
        Al[0][m] = __builtin_bit_cast(bf16x8, s_img[((13 + 8 * m) * 2 + 1) * 64 + lane]);
      }
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int cur = st & 1, nxt = cur ^ 1;
        if (st + 1 < 8) {
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            Ah[nxt][m] = __builtin_bit_cast(bf16x8, s_img[((13 + 8 * m + st + 1) * 2 + 0) * 64 + lane]);
            Al[nxt][m] = __builtin_bit_cast(bf16x8, s_img[((13 + 8 * m + st + 1) * 2 + 1) * 64 + lane]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(h1[st >> 1][8 * (st & 1) + j], 0.0f);
        bf16x8 bh, bl;
        split8(v, bh, bl);
#pragma unroll
        for (int m = 0; m < 4; ++m) h2[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al[cur][m], bh, h2[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m) h2[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[cur][m], bl, h2[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m) h2[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[cur][m], bh, h2[m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int m0 = 0; m0 < 4; ++m0)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(h1[m0][8 * q + j], 0.0f);
        bf16x8 bh, bl;
        split8(v, bh, bl);
#pragma unroll
        for (int m = 0; m < 4; ++m) h2[m] = mma3(s_img, 13 + 8 * m + 2 * m0 + q, lane, bh, bl, h2[m]);
      }
    }
    float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float a = fmaxf(h2[m][r], 0.0f);
        const int u = 32 * m + 8 * (r >> 2) + 4 * h + (r & 3);
        o0 += a * tail[T_W3 + u]; o1 += a * tail[T_W3 + 132 + u]; o2 += a * tail[T_W3 + 264 + u];
      }
    o0 += __shfl_xor(o0, 32, 64); o1 += __shfl_xor(o1, 32, 64); o2 += __shfl_xor(o2, 32, 64);
    if (h == 0) {
      const float v0 = o0 + tail[T_W3 + 128] * dhx + tail[T_W3 + 129] * dhy + tail[T_W3 + 130] * dhz + tail[T_B3 + 0];
      const float v1 = o1 + tail[T_W3 + 132 + 128] * dhx + tail[T_W3 + 132 + 129] * dhy + tail[T_W3 + 132 + 130] * dhz + tail[T_B3 + 1];
      const float v2 = o2 + tail[T_W3 + 264 + 128] * dhx + tail[T_W3 + 264 + 129] * dhy + tail[T_W3 + 264 + 130] * dhz + tail[T_B3 + 2];
      float4 r4 = make_float4(1.0f / (1.0f + expf(-v0)), 1.0f / (1.0f + expf(-v1)), 1.0f / (1.0f + expf(-v2)), 0.0f);
      reinterpret_cast<float4*>(out)[(size_t)grp * 32 + n] = r4;
    }
  }
}

static uint16_t bf16_rne(float x) { uint32_t b; memcpy(&b, &x, 4); b += 0x7FFF + ((b >> 16) & 1); return (uint16_t)(b >> 16); }
static float bf16_f(uint16_t v) { uint32_t b = (uint32_t)v << 16; float x; memcpy(&x, &b, 4); return x; }
static double urand(double a) { return (2.0 * rand() / RAND_MAX - 1.0) * a; }

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 725 * 1024, groups = rows / 32;
  srand(7);
  std::vector<double> basis(27 * 72), w1(128 * 27), b1(128), w2(128 * 128), b2(128), w3(3 * 131), b3(3, 0.0);
  for (auto& x : basis) x = urand(1 / sqrt(72.0));
  for (auto& x : w1) x = urand(1 / sqrt(27.0));
  for (auto& x : b1) x = urand(1 / sqrt(27.0));
  for (auto& x : w2) x = urand(1 / sqrt(128.0));
  for (auto& x : b2) x = urand(1 / sqrt(128.0));
  for (auto& x : w3) x = urand(1 / sqrt(131.0));
  const double dh[3] = {0.48, -0.6, 0.64};
  // image
  std::vector<uint16_t> img((size_t)IMG_U4 * 8, 0);
  auto put = [&](int f, int lane, int j, double v) {
    const float x = (float)v; const uint16_t hi = bf16_rne(x), lo = bf16_rne(x - bf16_f(hi));
    img[(((size_t)f * 2 + 0) * 64 + lane) * 8 + j] = hi; img[(((size_t)f * 2 + 1) * 64 + lane) * 8 + j] = lo;
  };
  for (int lane = 0; lane < 64; ++lane) {
    const int n = lane & 31, h = lane >> 5;
    for (int j = 0; j < 8; ++j) {
      for (int ks = 0; ks < 5; ++ks) { const int c = chan_of(h, 8 * ks + j); put(ks, lane, j, (c >= 0 && n < 27) ? basis[n * 72 + c] : 0.0); }
      for (int m = 0; m < 4; ++m) for (int q = 0; q < 2; ++q) { const int u = unit_of(0, q, h, j); put(5 + 2 * m + q, lane, j, u < 27 ? w1[(32 * m + n) * 27 + u] : 0.0); }
      for (int m = 0; m < 4; ++m) for (int m0 = 0; m0 < 4; ++m0) for (int q = 0; q < 2; ++q) put(13 + 8 * m + 2 * m0 + q, lane, j, w2[(32 * m + n) * 128 + unit_of(m0, q, h, j)]);
    }
  }
  std::vector<float> tail(T_FLOATS, 0.0f);
  for (int i = 0; i < 128; ++i) { tail[T_B1 + i] = (float)b1[i]; tail[T_B2 + i] = (float)b2[i]; }
  for (int c = 0; c < 3; ++c) { for (int u = 0; u < 131; ++u) tail[T_W3 + 132 * c + u] = (float)w3[c * 131 + u]; tail[T_B3 + c] = (float)b3[c]; }
  // inputs: X [rows][72] (only the first 4096 rows are distinct; the rest repeat them), in both lane layouts
  const int distinct = 4096;
  std::vector<double> X((size_t)distinct * 72);
  for (auto& x : X) x = urand(0.05);
  std::vector<float> in0((size_t)groups * 64 * 40, 0.0f), in1((size_t)groups * 64 * 16, 0.0f);
  std::vector<double> feat((size_t)distinct * 32, 0.0), ref((size_t)distinct * 3);
  for (int r = 0; r < distinct; ++r) {
    double hh1[128], hh2[128];
    for (int f = 0; f < 27; ++f) { double s = 0; for (int c = 0; c < 72; ++c) s += basis[f * 72 + c] * X[(size_t)r * 72 + c]; feat[(size_t)r * 32 + f] = s; }
    for (int u = 0; u < 128; ++u) { double s = b1[u]; for (int f = 0; f < 27; ++f) s += w1[u * 27 + f] * feat[(size_t)r * 32 + f]; hh1[u] = s > 0 ? s : 0; }
    for (int u = 0; u < 128; ++u) { double s = b2[u]; for (int v = 0; v < 128; ++v) s += w2[u * 128 + v] * hh1[v]; hh2[u] = s > 0 ? s : 0; }
    for (int c = 0; c < 3; ++c) { double s = b3[c]; for (int u = 0; u < 128; ++u) s += w3[c * 131 + u] * hh2[u]; for (int k = 0; k < 3; ++k) s += w3[c * 131 + 128 + k] * dh[k]; ref[(size_t)r * 3 + c] = 1.0 / (1.0 + exp(-s)); }
  }
  for (int g = 0; g < groups; ++g)
    for (int lane = 0; lane < 64; ++lane) {
      const int n = lane & 31, h = lane >> 5, r = (g * 32 + n) % distinct;
      for (int v = 0; v < 36; ++v) in0[((size_t)g * 64 + lane) * 40 + v] = (float)X[(size_t)r * 72 + chan_of(h, v)];
      for (int q = 0; q < 16; ++q) in1[((size_t)g * 64 + lane) * 16 + q] = (float)feat[(size_t)r * 32 + 8 * (q >> 2) + 4 * h + (q & 3)];   // D layout of tile 0
    }
  uint4* d_img; float *d_tail, *d_in0, *d_in1, *d_out;
  hipMalloc(&d_img, img.size() * 2); hipMalloc(&d_tail, tail.size() * 4); hipMalloc(&d_in0, in0.size() * 4); hipMalloc(&d_in1, in1.size() * 4);
  hipMalloc(&d_out, (size_t)groups * 32 * 16);
  hipMemcpy(d_img, img.data(), img.size() * 2, hipMemcpyHostToDevice); hipMemcpy(d_tail, tail.data(), tail.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_in0, in0.data(), in0.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_in1, in1.data(), in1.size() * 4, hipMemcpyHostToDevice);
  const int lds = IMG_U4 * 16 + T_FLOATS * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_w32<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_w32<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_w32<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  std::vector<float> out((size_t)groups * 32 * 4);
  for (int mode = 0; mode < 3; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k_mlp_w32<0>, dim3(256), dim3(512), lds, 0, d_img, d_tail, d_in0, groups, (float)dh[0], (float)dh[1], (float)dh[2], d_out);
      else if (mode == 1) hipLaunchKernelGGL(k_mlp_w32<1>, dim3(256), dim3(512), lds, 0, d_img, d_tail, d_in1, groups, (float)dh[0], (float)dh[1], (float)dh[2], d_out);
      else           hipLaunchKernelGGL(k_mlp_w32<2>, dim3(256), dim3(512), lds, 0, d_img, d_tail, d_in1, groups, (float)dh[0], (float)dh[1], (float)dh[2], d_out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 2; }
    hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0; int nan = 0;
    for (int r = 0; r < groups * 32; ++r) for (int c = 0; c < 3; ++c) {
      const float v = out[(size_t)r * 4 + c];
      if (!(v == v)) { ++nan; continue; }
      const double d = fabs((double)v - ref[(size_t)(r % distinct) * 3 + c]);
      if (d > worst) worst = d;
    }
    printf("mode %d (%s): %d rows in %.1f us, max |rgb - fp64 reference| %.2e, NaN %d -> %s\n", mode, mode == 0 ? "basis + W1 + W2 + head" : (mode == 1 ? "W1 + W2 + head (k_mlp's work)" : "as mode 1, W2 fragments prefetched one K-step ahead"),
           groups * 32, best * 1e3, worst, nan, (worst < 2e-5 && !nan) ? "OK" : "MISMATCH");
  }
  return 0;
}
