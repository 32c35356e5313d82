// pk_mfma.hip -- a VALU instruction of one wave, disturbed by the MFMAs of ANOTHER wave on the same SIMD.
//
// Captures of k_shade3 (scripts/gpu_diag.py capture3) show, in renders that differ from run to run, whole groups of lanes
// 48..63 whose gathered products are wrong in exactly the LOW halves of hipcc's v_pk_* register pairs -- only when waves
// interpolate (packed fp32 arithmetic) while other waves of the workgroup run their MFMA chain, never when a workgroup
// barrier keeps the phases apart.  This program isolates it: waves 0-3 of a 512-thread workgroup (one per SIMD) execute
// the instruction under test on small-integer operands (every result is exact, and the expected value comes from the
// integer ALU); waves 4-7 (the other wave of each SIMD) run MFMAs, or something else, or nothing.
//   hipcc --offload-arch=gfx950 -O3 -o pk_mfma pk_mfma.hip && ./pk_mfma [launches] [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// OP (victim wave): 0 v_pk_mul_f32 | 1 v_pk_fma_f32 | 2 v_pk_add_f32 | 3 v_cvt_pk_bf16_f32 | 4 v_mul_f32 | 5 v_fma_f32 |
//                   6 v_pk_mov_b32 | 7 v_mad_u64_u32 | 8 v_cvt_f32_i32 + v_add_f32 (scalar pair) | 9 v_lshl_add_u64
// PARTNER: 0 idle | 1 32x32x16 bf16, 2 accumulators | 2 16x16x32 bf16, 4 accumulators | 3 ds_read_b128 | 4 VALU |
//          5 32x32x16 bf16, 4 accumulators | 6 16x16x32 bf16, one accumulator (dependent) | 7 16x16x4 f32 | 8 16x16x32 bf16 + ds_read_b128
template <int OP, int PARTNER>
__global__ __launch_bounds__(512) void k_pk(uint32_t* __restrict__ bad, int iters) {
  __shared__ uint4 s_f[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) s_f[i] = make_uint4(i, i * 3, i * 5, i * 7);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < 4) {
    uint32_t ia = 3 + lane % 7, ib = 1 + lane % 5;                    // small integers: a * b + a < 2^24, exact in fp32
    uint32_t nlo = 0, nhi = 0;
    for (int it = 0; it < iters; ++it) {
      const float a = (float)ia, b = (float)ib;
      uint32_t lo = 0, hi = 0, want_lo = 0, want_hi = 0;
      if (OP == 0) {
        asm volatile("v_pk_mul_f32 v[104:105], %2, %3 op_sel:[0,1] op_sel_hi:[0,1]\n s_nop 1\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n"
                     : "=v"(lo), "=v"(hi) : "v"(((float __attribute__((ext_vector_type(2)))){a, b})), "v"(((float __attribute__((ext_vector_type(2)))){b, a})) : "v104", "v105");
        want_lo = want_hi = __float_as_uint((float)(ia * ia));
      } else if (OP == 1) {
        asm volatile("v_pk_fma_f32 v[104:105], %2, %3, %2 op_sel_hi:[0,0,0]\n s_nop 1\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n"
                     : "=v"(lo), "=v"(hi) : "v"(((float __attribute__((ext_vector_type(2)))){a, b})), "v"(((float __attribute__((ext_vector_type(2)))){b, a})) : "v104", "v105");
        want_lo = want_hi = __float_as_uint((float)(ia * ib + ia));
      } else if (OP == 2) {
        asm volatile("v_pk_add_f32 v[104:105], %2, %3 op_sel_hi:[0,0]\n s_nop 1\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n"
                     : "=v"(lo), "=v"(hi) : "v"(((float __attribute__((ext_vector_type(2)))){a, b})), "v"(((float __attribute__((ext_vector_type(2)))){b, a})) : "v104", "v105");
        want_lo = want_hi = __float_as_uint((float)(ia + ib));
      } else if (OP == 3) {
        uint32_t r;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n" : "=v"(r) : "v"(a), "v"(b));
        lo = r & 0xffffu; hi = r >> 16;
        const uint32_t ba = __float_as_uint(a), bb = __float_as_uint(b);                      // round to nearest even
        want_lo = (ba + 0x7fffu + ((ba >> 16) & 1u)) >> 16; want_hi = (bb + 0x7fffu + ((bb >> 16) & 1u)) >> 16;
      } else if (OP == 4) {
        float r, r2;
        asm volatile("v_mul_f32 %0, %2, %3\n v_mul_f32 %1, %3, %3\n" : "=v"(r), "=v"(r2) : "v"(a), "v"(b));
        lo = __float_as_uint(r); hi = __float_as_uint(r2);
        want_lo = __float_as_uint((float)(ia * ib)); want_hi = __float_as_uint((float)(ib * ib));
      } else if (OP == 5) {
        float r, r2;
        asm volatile("v_fma_f32 %0, %2, %3, %2\n v_fma_f32 %1, %3, %3, %2\n" : "=v"(r), "=v"(r2) : "v"(a), "v"(b));
        lo = __float_as_uint(r); hi = __float_as_uint(r2);
        want_lo = __float_as_uint((float)(ia * ib + ia)); want_hi = __float_as_uint((float)(ib * ib + ia));
      } else if (OP == 6) {
        asm volatile("v_pk_mov_b32 v[104:105], %2, %3 op_sel:[0,1]\n s_nop 1\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n"
                     : "=v"(lo), "=v"(hi) : "v"(((float __attribute__((ext_vector_type(2)))){a, b})), "v"(((float __attribute__((ext_vector_type(2)))){b, a})) : "v104", "v105");
        want_lo = __float_as_uint(a); want_hi = __float_as_uint(a);                          // lo = src0.lo, hi = src1.hi
      } else if (OP == 7) {
        asm volatile("v_mov_b32 v102, %3\n v_mov_b32 v103, 0\n v_mad_u64_u32 v[104:105], s[40:41], %2, %3, v[102:103]\n s_nop 1\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n"
                     : "=v"(lo), "=v"(hi) : "v"(ia * 65537u), "v"(ib * 257u) : "v102", "v103", "v104", "v105", "s40", "s41");
        const unsigned long long w = (unsigned long long)(ia * 65537u) * (ib * 257u) + (ib * 257u);
        want_lo = (uint32_t)w; want_hi = (uint32_t)(w >> 32);
      } else if (OP == 8) {
        float r, r2;
        asm volatile("v_cvt_f32_u32 %0, %2\n v_add_f32 %1, %3, %3\n" : "=v"(r), "=v"(r2) : "v"(ia), "v"(b));
        lo = __float_as_uint(r); hi = __float_as_uint(r2);
        want_lo = __float_as_uint(a); want_hi = __float_as_uint((float)(2 * ib));
      } else if (OP == 10) {                                  // v_pk_fma_f32, low result reads the HIGH half of src1
        asm volatile("v_pk_fma_f32 v[104:105], %2, %3, %2 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n s_nop 1\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n"
                     : "=v"(lo), "=v"(hi) : "v"(((float __attribute__((ext_vector_type(2)))){a, b})), "v"(((float __attribute__((ext_vector_type(2)))){b, a})) : "v104", "v105");
        want_lo = want_hi = __float_as_uint((float)(ia * ia + ia));
      } else if (OP == 11) {                                  // v_pk_mul_f32, straight halves (lo x lo, hi x hi)
        asm volatile("v_pk_mul_f32 v[104:105], %2, %3\n s_nop 1\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n"
                     : "=v"(lo), "=v"(hi) : "v"(((float __attribute__((ext_vector_type(2)))){a, b})), "v"(((float __attribute__((ext_vector_type(2)))){b, a})) : "v104", "v105");
        want_lo = want_hi = __float_as_uint((float)(ia * ib));
      } else if (OP == 12) {                                  // v_pk_add_f32, low result reads the HIGH half of src1
        asm volatile("v_pk_add_f32 v[104:105], %2, %3 op_sel:[0,1] op_sel_hi:[0,1]\n s_nop 1\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n"
                     : "=v"(lo), "=v"(hi) : "v"(((float __attribute__((ext_vector_type(2)))){a, b})), "v"(((float __attribute__((ext_vector_type(2)))){b, a})) : "v104", "v105");
        want_lo = want_hi = __float_as_uint((float)(ia + ia));
      } else if (OP == 13) {                                  // v_pk_mul_f32, low result reads the HIGH half of src0
        asm volatile("v_pk_mul_f32 v[104:105], %2, %3 op_sel:[1,0] op_sel_hi:[1,0]\n s_nop 1\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n"
                     : "=v"(lo), "=v"(hi) : "v"(((float __attribute__((ext_vector_type(2)))){a, b})), "v"(((float __attribute__((ext_vector_type(2)))){b, a})) : "v104", "v105");
        want_lo = want_hi = __float_as_uint((float)(ib * ib));
      } else if (OP == 14) {                                  // v_pk_mul_f32, HIGH result reads the LOW halves (op_sel_hi 0,0), low straight
        asm volatile("v_pk_mul_f32 v[104:105], %2, %3 op_sel_hi:[0,0]\n s_nop 1\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n"
                     : "=v"(lo), "=v"(hi) : "v"(((float __attribute__((ext_vector_type(2)))){a, b})), "v"(((float __attribute__((ext_vector_type(2)))){b, a})) : "v104", "v105");
        want_lo = want_hi = __float_as_uint((float)(ia * ib));
      } else {
        asm volatile("v_mov_b32 v102, %3\n v_mov_b32 v103, %2\n v_mov_b32 v100, %2\n v_mov_b32 v101, %3\n v_lshl_add_u64 v[104:105], v[100:101], 3, v[102:103]\n s_nop 1\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n"
                     : "=v"(lo), "=v"(hi) : "v"(ia), "v"(ib) : "v100", "v101", "v102", "v103", "v104", "v105");
        want_lo = ia * 8 + ib; want_hi = ib * 8 + ia;
      }
      nlo += lo != want_lo;
      nhi += hi != want_hi;
      ia += 3; ib += 1;
      if (ia > 2000) ia = 3 + lane % 7;
      if (ib > 2000) ib = 1 + lane % 5;
    }
    if (nlo | nhi) { atomicAdd(&bad[0], nlo); atomicAdd(&bad[1], nhi); atomicAdd(&bad[2 + (lane >> 4)], 1u); }
  } else {
    bf16x8 A, B;
    for (int j = 0; j < 8; ++j) { A[j] = (__bf16)(float)((lane + j) % 3); B[j] = (__bf16)(float)((lane * 2 + j) % 3); }
    float s = 0;
    if (PARTNER == 1) {
      f32x16 c0 = {0}, c1 = {0};
      for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, c1, 0, 0, 0);
      }
      for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    } else if (PARTNER == 5) {
      f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
      for (int it = 0; it < iters / 2; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, c3, 0, 0, 0);
      }
      for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    } else if (PARTNER == 2 || PARTNER == 8) {
      f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
      uint32_t q = 0;
      for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, c3, 0, 0, 0);
        if (PARTNER == 8) { const uint4 v = s_f[(lane + it * 64) & 4095]; q += v.x ^ v.w; }
      }
      for (int r = 0; r < 4; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
      s += (float)q;
    } else if (PARTNER == 6) {
      f32x4 c0 = {0};
      for (int it = 0; it < iters * 2; ++it) c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, c0, 0, 0, 0);
      for (int r = 0; r < 4; ++r) s += c0[r];
    } else if (PARTNER == 7) {
      f32x4 c0 = {0}, c1 = {0};
      for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32((float)lane, 1.0f, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32((float)lane, 2.0f, c1, 0, 0, 0);
      }
      for (int r = 0; r < 4; ++r) s += c0[r] + c1[r];
    } else if (PARTNER == 3) {
      uint32_t q = 0;
      for (int it = 0; it < iters * 2; ++it) { const uint4 v = s_f[(lane + it * 64) & 4095]; q += v.x ^ v.y ^ v.z ^ v.w; }
      s = (float)q;
    } else if (PARTNER == 4) {
      s = (float)lane;
      for (int it = 0; it < iters * 8; ++it) s = s * 1.0001f + 0.5f;
    }
    if (s == 1.2345f) bad[8] = 1;
  }
}

static const char* OPN[15] = {"v_pk_mul_f32 lo<-s1.hi", "v_pk_fma_f32 straight", "v_pk_add_f32 straight", "v_cvt_pk_bf16_f32", "v_mul_f32 x2", "(test artefact)", "v_pk_mov_b32", "v_mad_u64_u32", "(test artefact)", "v_lshl_add_u64", "v_pk_fma_f32 lo<-s1.hi", "v_pk_mul_f32 straight", "v_pk_add_f32 lo<-s1.hi", "v_pk_mul_f32 lo<-s0.hi", "v_pk_mul_f32 hi<-lo halves"};
static const char* PN[9] = {"idle", "mfma 32x32x16 bf16 (2 acc)", "mfma 16x16x32 bf16 (4 acc)", "ds_read_b128", "VALU", "mfma 32x32x16 bf16 (4 acc)", "mfma 16x16x32 bf16 (1 acc, dependent)", "mfma 16x16x4 f32", "mfma 16x16x32 bf16 + ds_read_b128"};

template <int OP, int P>
static void run(uint32_t* d_bad, int launches, int iters) {
  hipMemset(d_bad, 0, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((k_pk<OP, P>), dim3(256), dim3(512), 0, 0, d_bad, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  uint32_t hb[16];
  hipMemcpy(hb, d_bad, 64, hipMemcpyDeviceToHost);
  printf("victim %-26s partner %-40s wrong lo %9u hi %9u of %.3g | lanes 0-15: %u, 16-31: %u, 32-47: %u, 48-63: %u | %.2f ms/launch\n", OPN[OP], PN[P], hb[0], hb[1],
         (double)launches * 256 * 4 * 64 * iters, hb[2], hb[3], hb[4], hb[5], ms / launches);
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 10, iters = argc > 2 ? atoi(argv[2]) : 20000;
  uint32_t* d_bad; hipMalloc(&d_bad, 64);
  run<0, 0>(d_bad, launches, iters);
  run<0, 2>(d_bad, launches, iters); run<0, 1>(d_bad, launches, iters); run<0, 5>(d_bad, launches, iters); run<0, 6>(d_bad, launches, iters);
  run<0, 7>(d_bad, launches, iters); run<0, 8>(d_bad, launches, iters); run<0, 3>(d_bad, launches, iters); run<0, 4>(d_bad, launches, iters);
  run<11, 2>(d_bad, launches, iters); run<13, 2>(d_bad, launches, iters); run<14, 2>(d_bad, launches, iters);
  run<1, 2>(d_bad, launches, iters); run<10, 2>(d_bad, launches, iters); run<2, 2>(d_bad, launches, iters); run<12, 2>(d_bad, launches, iters);
  run<3, 2>(d_bad, launches, iters); run<4, 2>(d_bad, launches, iters); run<6, 2>(d_bad, launches, iters); run<7, 2>(d_bad, launches, iters); run<9, 2>(d_bad, launches, iters);
  run<10, 5>(d_bad, launches, iters); run<12, 5>(d_bad, launches, iters); run<13, 5>(d_bad, launches, iters); run<3, 5>(d_bad, launches, iters);
  run<0, 8>(d_bad, launches, iters); run<10, 8>(d_bad, launches, iters);
  return 0;
}
