// Micro-benchmark: LDS atomic throughput on gfx950 (ds_add_f32 / ds_add_u32 / ds_add_rtn_u32 /
// plain ds_write), random vs conflict-free addresses.  hipcc --offload-arch=gfx950 -O3 -o lds_atomic lds_atomic.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
constexpr int N = 8192;     // LDS words
template <int MODE, int PATTERN>
__global__ __launch_bounds__(256) void k(int iters, float* out) {
  __shared__ float acc[N];
  for (int i = threadIdx.x; i < N; i += 256) acc[i] = 0.f;
  __syncthreads();
  uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1;
  float r = 0.f;
  for (int it = 0; it < iters; ++it) {
    int a;
    if (PATTERN == 0) a = (threadIdx.x + it * 256) & (N - 1);              // conflict-free, distinct
    else if (PATTERN == 1) { s = s * 1664525u + 1013904223u; a = (s >> 8) & (N - 1); }  // random
    else if (PATTERN == 2) { s = s * 1664525u + 1013904223u; a = (((s >> 8) & 1023) * 8 + (threadIdx.x & 7)) & (N - 1); } // 8-lane groups on random texels
    else { uint32_t g = (threadIdx.x >> 2) * 2654435761u + blockIdx.x * 40503u + it * 97u; g = g * 1664525u + 1013904223u;     // 4-lane groups, lane owns 64-bit slot j of a random 24-float texel
           a = (((g >> 8) % 340) * 24 + (threadIdx.x & 3) * 6 + 2 * (it % 3)) & (N - 1); }
    if (MODE == 0) atomicAdd(&acc[a], 1.0f);                                // ds_add_f32
    else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(&acc[a]), 1u);  // ds_add_u32
    else if (MODE == 2) r += (float)atomicAdd(reinterpret_cast<unsigned*>(&acc[a]), 1u);  // ds_add_rtn_u32
    else if (MODE == 3) acc[a] = (float)it;                                 // ds_write_b32
    else if (MODE == 4) r += atomicAdd(&acc[a], 1.0f);                      // ds_add_rtn_f32
    else if (MODE == 6) {                                                    // CAS-loop float add
      unsigned* p = reinterpret_cast<unsigned*>(&acc[a]);
      unsigned old = *p, assumed;
      do { assumed = old; old = atomicCAS(p, assumed, __float_as_uint(__uint_as_float(assumed) + 1.0f)); } while (old != assumed);
    } else if (MODE == 7) atomicAdd(reinterpret_cast<double*>(&acc[(a >> 1) << 1]), 1.0);          // ds_add_f64
    else if (MODE == 8) atomicAdd(reinterpret_cast<unsigned long long*>(&acc[(a >> 1) << 1]), 1ull);  // ds_add_u64
    else if (MODE == 9) r += (float)atomicAdd(reinterpret_cast<unsigned long long*>(&acc[(a >> 1) << 1]), 1ull);  // ds_add_rtn_u64
    else if (MODE == 10) {                                                   // CAS-loop on 64 bits: two fp32 adds (the scatter kernels' form up to round 5)
      unsigned long long* p = reinterpret_cast<unsigned long long*>(&acc[(a >> 1) << 1]);
      unsigned long long old = *p, assumed;
      do {
        assumed = old;
        const unsigned long long nv = (unsigned long long)__float_as_uint(__uint_as_float((unsigned)assumed) + 1.0f) |
                                      ((unsigned long long)__float_as_uint(__uint_as_float((unsigned)(assumed >> 32)) + 1.0f) << 32);
        old = atomicCAS(p, assumed, nv);
      } while (old != assumed);
    }
    else { float v = acc[a]; acc[a] = v + 1.0f; }                           // non-atomic rmw
  }
  __syncthreads();
  float t = r;
  for (int i = threadIdx.x; i < N; i += 256) t += acc[i];
  if (t == 12345.678f) out[0] = t;
}
template <int MODE, int PATTERN>
void run(const char* name) {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4096, blocks = 256 * 8;
  k<MODE, PATTERN><<<blocks, 256>>>(16, out);
  hipEventRecord(e0);
  k<MODE, PATTERN><<<blocks, 256>>>(iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ops = (double)blocks * 256 * iters;
  printf("%-28s %8.3f ms  %8.1f G lane-ops/s  (%.2f lanes/clk/CU at 2.4 GHz, 256 CUs)\n", name, ms, ops / ms / 1e6,
         ops / (ms * 1e-3) / 256 / 2.4e9);
}
int main() {
  run<0, 0>("ds_add_f32 distinct"); run<0, 1>("ds_add_f32 random"); run<0, 2>("ds_add_f32 8-lane texel");
  run<1, 0>("ds_add_u32 distinct"); run<1, 1>("ds_add_u32 random");
  run<2, 0>("ds_add_rtn_u32 distinct"); run<2, 1>("ds_add_rtn_u32 random");
  run<4, 1>("ds_add_rtn_f32 random");
  run<6, 0>("CAS-loop f32 distinct"); run<6, 1>("CAS-loop f32 random"); run<6, 2>("CAS-loop f32 8-lane texel");
  run<7, 1>("ds_add_f64 random"); run<8, 1>("ds_add_u64 random"); run<8, 0>("ds_add_u64 distinct"); run<8, 3>("ds_add_u64 4-lane texel");
  run<9, 1>("ds_add_rtn_u64 random"); run<10, 1>("CAS-loop 2xf32 (b64) random"); run<10, 3>("CAS-loop 2xf32 (b64) 4-lane texel");
  run<1, 3>("ds_add_u32 4-lane texel"); run<6, 3>("CAS-loop f32 4-lane texel");
  run<3, 0>("ds_write distinct"); run<3, 1>("ds_write random");
  run<5, 0>("rmw non-atomic distinct"); run<5, 1>("rmw non-atomic random");
  return 0;
}
