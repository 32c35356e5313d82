// valu_vmem_raw.hip -- does a multi-pass integer VALU result reach the NEXT instruction of the wave in time?
//
// The colour kernel's gather addresses are (row + x) * 96 + 48 h: hipcc computes them with v_mad_u64_u32 (a quarter-rate,
// multi-pass VALU instruction) and issues global_load_dwordx4 with that register as its address 1-3 issue slots later.
// Captures of k_shade3 (scripts/gpu_diag.py capture3) show whole groups of lanes 48..63 -- the last quarter of the
// wave -- gathering from a wrong address now and then, only when the wave gathers while the texture path is otherwise idle.
// This program tests that in isolation: table[i] = i, every lane computes an index with the instruction under test and
// loads table[index] right behind it; the loaded value must equal the index.  Exact, no floating point.
//   mode 0  v_mad_u64_u32 -> 8 x s_nop 7 -> global_load_dword          (reference)
//   mode 1  v_mad_u64_u32 -> global_load_dword in the next issue slot
//   mode 2  v_mul_lo_u32  -> global_load_dword in the next issue slot
//   mode 3  v_mad_u64_u32 -> v_add_u32 (VALU consumer) in the next issue slot -> ... -> load
//   mode 4  v_mad_u64_u32 -> global_load_dwordx4 (16-byte) in the next issue slot
//   mode 5  v_lshl_add_u32 (full rate) -> global_load_dword in the next issue slot
//   hipcc --offload-arch=gfx950 -O3 -o valu_vmem_raw valu_vmem_raw.hip && ./valu_vmem_raw [launches] [iters] [waves per block]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int TABLE = 1 << 22;        // 4 M dwords = 16 MB: L2 / MALL resident, L1 misses are common (latency varies)

template <int MODE>
__global__ void k_raw(const uint32_t* __restrict__ table, uint32_t* __restrict__ bad, int iters, uint32_t mulc) {
  const uint32_t lane = threadIdx.x & 63, gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t x = gw * 2654435761u + lane * 40503u + 12345u;
  uint32_t nbad = 0, lanebits = 0;
  for (int it = 0; it < iters; ++it) {
    x = x * 1664525u + 1013904223u;
    const uint32_t a = (x >> 9) % 43000u;                 // a * 96 + c < 4 M
    const uint32_t c = (x >> 3) & 63u;
    uint32_t idx, got, got4[4];
    uint32_t lo_in = c, hi_in = 0;
    if (MODE == 0)
      asm volatile("v_mad_u64_u32 v[100:101], s[40:41], %2, %3, v[102:103]\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n"
                   "v_lshlrev_b32 v104, 2, v100\n s_nop 7\n global_load_dword %0, v104, %4\n v_mov_b32 %1, v100\n s_waitcnt vmcnt(0)\n"
                   : "=v"(got), "=v"(idx) : "v"(a), "s"(mulc), "s"(table), "{v102}"(lo_in), "{v103}"(hi_in) : "v100", "v101", "v104", "s40", "s41", "memory");
    else if (MODE == 1)
      // byte offset = (a * mulc + c) * 4 computed by ONE mad: a * (4 mulc) + 4 c, load in the next slot
      asm volatile("v_mad_u64_u32 v[100:101], s[40:41], %2, %3, v[102:103]\n global_load_dword %0, v100, %4\n v_lshrrev_b32 %1, 2, v100\n s_waitcnt vmcnt(0)\n"
                   : "=v"(got), "=v"(idx) : "v"(a), "s"(mulc * 4u), "s"(table), "{v102}"(lo_in * 4u), "{v103}"(hi_in) : "v100", "v101", "s40", "s41", "memory");
    else if (MODE == 2)
      asm volatile("v_mul_lo_u32 v100, %2, %3\n global_load_dword %0, v100, %4\n v_lshrrev_b32 %1, 2, v100\n s_waitcnt vmcnt(0)\n"
                   : "=v"(got), "=v"(idx) : "v"(a), "s"(mulc * 4u), "s"(table) : "v100", "memory");
    else if (MODE == 3)
      asm volatile("v_mad_u64_u32 v[100:101], s[40:41], %2, %3, v[102:103]\n v_add_u32 v104, 0, v100\n s_nop 7\n s_nop 7\n global_load_dword %0, v104, %4\n v_lshrrev_b32 %1, 2, v100\n s_waitcnt vmcnt(0)\n"
                   : "=v"(got), "=v"(idx) : "v"(a), "s"(mulc * 4u), "s"(table), "{v102}"(lo_in * 4u), "{v103}"(hi_in) : "v100", "v101", "v104", "s40", "s41", "memory");
    else if (MODE == 4) {
      // 16-byte load: index forced to a multiple of 4
      asm volatile("v_mad_u64_u32 v[100:101], s[40:41], %5, %6, v[102:103]\n global_load_dwordx4 v[104:107], v100, %7\n v_lshrrev_b32 %4, 2, v100\n s_waitcnt vmcnt(0)\n"
                   "v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n v_mov_b32 %2, v106\n v_mov_b32 %3, v107\n"
                   : "=v"(got4[0]), "=v"(got4[1]), "=v"(got4[2]), "=v"(got4[3]), "=v"(idx)
                   : "v"(a), "s"(mulc * 4u), "s"(table), "{v102}"((lo_in & ~3u) * 4u), "{v103}"(hi_in) : "v100", "v101", "v104", "v105", "v106", "v107", "s40", "s41", "memory");
      got = got4[0];
      if (got4[1] != idx + 1 || got4[2] != idx + 2 || got4[3] != idx + 3) got = ~idx;
    } else
      asm volatile("v_lshl_add_u32 v100, %2, 8, %5\n global_load_dword %0, v100, %4\n v_lshrrev_b32 %1, 2, v100\n s_waitcnt vmcnt(0)\n"
                   : "=v"(got), "=v"(idx) : "v"(a), "s"(mulc), "s"(table), "v"(lo_in * 4u) : "v100", "memory");
    if (got != idx) { ++nbad; lanebits = 1; }
    x += got;                                              // (keeps the load in the dependency chain)
  }
  if (nbad) { atomicAdd(&bad[0], nbad); atomicAdd(&bad[1 + (lane >> 4)], lanebits); }
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 20, iters = argc > 2 ? atoi(argv[2]) : 2000;
  const int wpb = argc > 3 ? atoi(argv[3]) : 4;
  std::vector<uint32_t> h(TABLE);
  for (int i = 0; i < TABLE; ++i) h[i] = (uint32_t)i;
  uint32_t *d_t, *d_bad;
  hipMalloc(&d_t, TABLE * 4); hipMalloc(&d_bad, 64);
  hipMemcpy(d_t, h.data(), TABLE * 4, hipMemcpyHostToDevice);
  const char* names[6] = {"v_mad_u64_u32 -> 64 wait states -> load (reference)", "v_mad_u64_u32 -> global_load_dword next slot", "v_mul_lo_u32 -> global_load_dword next slot",
                          "v_mad_u64_u32 -> v_add_u32 next slot", "v_mad_u64_u32 -> global_load_dwordx4 next slot", "v_lshl_add_u32 -> global_load_dword next slot"};
  for (int blocks : {256, 1024}) {
    for (int mode = 0; mode < 6; ++mode) {
      hipMemset(d_bad, 0, 64);
      for (int l = 0; l < launches; ++l) {
        switch (mode) {
          case 0: hipLaunchKernelGGL(k_raw<0>, dim3(blocks), dim3(64 * wpb), 0, 0, d_t, d_bad, iters, 96u); break;
          case 1: hipLaunchKernelGGL(k_raw<1>, dim3(blocks), dim3(64 * wpb), 0, 0, d_t, d_bad, iters, 96u); break;
          case 2: hipLaunchKernelGGL(k_raw<2>, dim3(blocks), dim3(64 * wpb), 0, 0, d_t, d_bad, iters, 96u); break;
          case 3: hipLaunchKernelGGL(k_raw<3>, dim3(blocks), dim3(64 * wpb), 0, 0, d_t, d_bad, iters, 96u); break;
          case 4: hipLaunchKernelGGL(k_raw<4>, dim3(blocks), dim3(64 * wpb), 0, 0, d_t, d_bad, iters, 96u); break;
          default: hipLaunchKernelGGL(k_raw<5>, dim3(blocks), dim3(64 * wpb), 0, 0, d_t, d_bad, iters, 96u); break;
        }
      }
      uint32_t hb[16];
      hipMemcpy(hb, d_bad, 64, hipMemcpyDeviceToHost);
      printf("blocks %4d x %d waves  mode %d  %-52s wrong loads %u of %.3g (lanes 0-15: %u, 16-31: %u, 32-47: %u, 48-63: %u lane-launch hits)\n", blocks, wpb, mode, names[mode], hb[0],
             (double)launches * blocks * wpb * 64 * iters, hb[1], hb[2], hb[3], hb[4]);
    }
  }
  return 0;
}
