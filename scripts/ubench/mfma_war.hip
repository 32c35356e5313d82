// mfma_war.hip -- when does v_mfma_f32_16x16x32_bf16 read its sources?
//
// docs/GFX950_FINDINGS.md findings 9 / 17 blamed rare run-to-run differences of the colour kernel on "an issued MFMA reads SrcA /
// SrcB later than the compiler assumes": hipcc reloads an A-fragment register with ds_read_b128 in the very next issue
// slot behind the MFMA that read it, and its hazard table asks for three wait states before an LDS / VALU write of a
// 4-pass MFMA's SrcC and for none on SrcA / SrcB.  That claim was never tested in isolation.  This program does:
// 16 waves per CU (4 per SIMD, the colour kernel's occupancy) run nothing but MFMA groups whose source registers are
// overwritten as early as hipcc would do it.  Operands are small integers, every sum stays below 2^24, so the
// arithmetic is EXACT: any late source read shows up as a wrong integer -- no rounding, no summation order.
//
//   mode 0  reference: 64 wait states between the last MFMA that reads a register and the load that overwrites it
//   mode 1  SrcA reloaded by ds_read_b128 in the issue slot right behind its last MFMA, 4 independent accumulators
//   mode 2  the same, the four MFMAs of a group chained on ONE accumulator (issued MFMAs queue on their dependency)
//   mode 3  SrcA overwritten by VALU (4 x v_mov_b32) right behind its last MFMA
//   mode 4  SrcA reloaded by global_load_dwordx4 right behind its last MFMA
//   mode 5  SrcC (vDst != SrcC) overwritten by ds_read_b128 three wait states behind the MFMA (hipcc's WAR rule)
//   mode 6  as 1, and SrcB overwritten with garbage by VALU right behind the MFMAs, restored before the next group
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_war mfma_war.hip && ./mfma_war [launches] [pairs]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define NFRAG 64
// v[64:79] acc0..3 | v[80:83] A0 | v[84:87] A1 | v[88:91] B | v[92:95] T or copy of B | v96 fragment address |
// v97 junk address | v[100:103] X | v[104:107] Y | s40 loop counter
#define CLOBBERS "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
                 "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95", \
                 "v96","v97","v100","v101","v102","v103","v104","v105","v106","v107","s40","scc","memory"
#define MF(d, a, c) "v_mfma_f32_16x16x32_bf16 " d ", " a ", v[88:91], " c "\n"
#define ACC0 "v[64:67]"
#define ACC1 "v[68:71]"
#define ACC2 "v[72:75]"
#define ACC3 "v[76:79]"
#define RA0 "v[80:83]"
#define RA1 "v[84:87]"
#define RX "v[100:103]"
#define RY "v[104:107]"
#define ADV "v_add_u32 v96, 0x1400, v96\n v_and_b32 v96, 0xffff, v96\n"          /* next fragment: +5 KB mod 64 KB */
#define PROLOG \
  "v_mov_b32 v88, %16\n v_mov_b32 v89, %17\n v_mov_b32 v90, %18\n v_mov_b32 v91, %19\n" \
  "v_mov_b32 v96, %20\n s_mov_b32 s40, %21\n v_mov_b32 v97, %23\n" \
  "v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n" \
  "v_mov_b32 v68, 0\n v_mov_b32 v69, 0\n v_mov_b32 v70, 0\n v_mov_b32 v71, 0\n" \
  "v_mov_b32 v72, 0\n v_mov_b32 v73, 0\n v_mov_b32 v74, 0\n v_mov_b32 v75, 0\n" \
  "v_mov_b32 v76, 0\n v_mov_b32 v77, 0\n v_mov_b32 v78, 0\n v_mov_b32 v79, 0\n" \
  "v_mov_b32 v100, 0\n v_mov_b32 v101, 0\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0\n" \
  "v_mov_b32 v104, 0\n v_mov_b32 v105, 0\n v_mov_b32 v106, 0\n v_mov_b32 v107, 0\n" \
  "v_mov_b32 v92, v88\n v_mov_b32 v93, v89\n v_mov_b32 v94, v90\n v_mov_b32 v95, v91\n s_nop 7\n"
#define EPILOG_(l0, l1, l2, l3) \
  "s_nop 15\n s_nop 15\n s_waitcnt vmcnt(0) lgkmcnt(0)\n" \
  "v_mov_b32 %0, v64\n v_mov_b32 %1, v65\n v_mov_b32 %2, v66\n v_mov_b32 %3, v67\n" \
  "v_mov_b32 %4, v68\n v_mov_b32 %5, v69\n v_mov_b32 %6, v70\n v_mov_b32 %7, v71\n" \
  "v_mov_b32 %8, v72\n v_mov_b32 %9, v73\n v_mov_b32 %10, v74\n v_mov_b32 %11, v75\n" \
  "v_mov_b32 %12, " l0 "\n v_mov_b32 %13, " l1 "\n v_mov_b32 %14, " l2 "\n v_mov_b32 %15, " l3 "\n"
#define EPILOG EPILOG_("v76", "v77", "v78", "v79")
#define EPILOG_Y EPILOG_("v104", "v105", "v106", "v107")
#define OPERANDS \
  : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), \
    "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15]) \
  : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(va), "s"(pairs), "s"(gbase), "v"(vjunk) \
  : CLOBBERS
#define LDS_(reg) "ds_read_b128 " reg ", v96\n"
#define GLB_(reg) "global_load_dwordx4 " reg ", v96, %22\n"
#define HOLD64 "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
#define FOUR_IND(a) MF(ACC0, a, ACC0) MF(ACC1, a, ACC1) MF(ACC2, a, ACC2) MF(ACC3, a, ACC3)
#define FOUR_DEP(a) MF(ACC0, a, ACC0) MF(ACC0, a, ACC0) MF(ACC0, a, ACC0) MF(ACC0, a, ACC0)
#define THREE_IND(a) MF(ACC0, a, ACC0) MF(ACC1, a, ACC1) MF(ACC2, a, ACC2)
#define LOOP_END "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 L_top%=\n"
#define TRASH_B "v_mov_b32 v88, 0x40404040\n v_mov_b32 v89, 0x40404040\n v_mov_b32 v90, 0x40404040\n v_mov_b32 v91, 0x40404040\n" \
                "v_mov_b32 v88, v92\n v_mov_b32 v89, v93\n v_mov_b32 v90, v94\n v_mov_b32 v91, v95\n s_nop 3\n"
// two-register-set body: WAIT1 = counter wait for "all but the newest load", LOAD = reload instruction, FOUR = MFMA group,
// GAP = what sits between a group's last MFMA and the reload of its A registers
#define BODY2(WAIT1, LOAD, FOUR, GAP) \
  PROLOG LOAD(RA0) ADV LOAD(RA1) \
  "L_top%=:\n" \
  WAIT1 ADV FOUR(RA0) GAP LOAD(RA0) \
  WAIT1 ADV FOUR(RA1) GAP LOAD(RA1) \
  LOOP_END EPILOG

// G = 2 * pairs groups; group i multiplies fragment f_i = (7 wave + 5 i) mod 64 by B.  Expected: every independent
// accumulator = sum_i A_{f_i} B  (mode 2: acc0 = 4 x that, the others 0; mode 5: acc0..2 and Y = that).
template <int MODE>
__global__ __launch_bounds__(1024) void k_war(const uint4* __restrict__ frags, float* __restrict__ out, int pairs) {
  __shared__ uint4 s_f[(NFRAG + 1) * 64];
  for (int i = threadIdx.x; i < (NFRAG + 1) * 64; i += blockDim.x) s_f[i] = frags[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t b[4];
  for (int q = 0; q < 4; ++q) {                     // B: small integers as bf16 pairs
    const int e0 = (lane * 3 + 2 * q) % 5 - 2, e1 = (lane * 3 + 2 * q + 1) % 5 - 2;
    b[q] = (__float_as_uint((float)e0) >> 16) | (__float_as_uint((float)e1) & 0xffff0000u);
  }
  const uint32_t lds0 = (uint32_t)(size_t)s_f;       // LDS byte offset of the table (0: the only shared object)
  const uint32_t va = (uint32_t)(wave * 7 % NFRAG) * 1024u + (uint32_t)lane * 16u;   // fragment f_0
  const uint32_t vjunk = NFRAG * 1024u + (uint32_t)lane * 16u;                         // 1.0f everywhere
  const uint4* gbase = frags;
  float r[16];
  if (lds0 != 0) { if (threadIdx.x == 0) out[0] = -12345.0f; return; }
  if constexpr (MODE == 0)      asm volatile(BODY2("s_waitcnt lgkmcnt(1)\n", LDS_, FOUR_IND, HOLD64) OPERANDS);
  else if constexpr (MODE == 1) asm volatile(BODY2("s_waitcnt lgkmcnt(1)\n", LDS_, FOUR_IND, "") OPERANDS);
  else if constexpr (MODE == 2) asm volatile(BODY2("s_waitcnt lgkmcnt(1)\n", LDS_, FOUR_DEP, "") OPERANDS);
  else if constexpr (MODE == 4) asm volatile(BODY2("s_waitcnt vmcnt(1)\n", GLB_, FOUR_IND, "") OPERANDS);
  else if constexpr (MODE == 6) asm volatile(BODY2("s_waitcnt lgkmcnt(1)\n", LDS_, FOUR_IND, TRASH_B) OPERANDS);
  else if constexpr (MODE == 3)
    asm volatile(PROLOG LDS_(RA0) "s_lshl_b32 s40, s40, 1\n"
                 "L_top%=:\n"
                 ADV "ds_read_b128 v[92:95], v96\n s_waitcnt lgkmcnt(0)\n"
                 FOUR_IND(RA0)
                 "v_mov_b32 v80, v92\n v_mov_b32 v81, v93\n v_mov_b32 v82, v94\n v_mov_b32 v83, v95\n s_nop 3\n"
                 LOOP_END EPILOG OPERANDS);
  else if constexpr (MODE == 5)
    asm volatile(PROLOG LDS_(RA0) ADV LDS_(RA1)
                 "L_top%=:\n"
                 "s_waitcnt lgkmcnt(0)\n" ADV THREE_IND(RA0) MF(RX, RA0, RY)
                 "s_nop 2\n ds_read_b128 v[104:107], v97\n" LDS_(RA0)
                 "s_waitcnt lgkmcnt(0)\n" ADV THREE_IND(RA1) MF(RY, RA1, RX)
                 "s_nop 2\n ds_read_b128 v[100:103], v97\n" LDS_(RA1)
                 LOOP_END EPILOG_Y OPERANDS);
  float* o = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  for (int i = 0; i < 16; ++i) o[i] = r[i];
}

static uint16_t bf16_of_int(int v) { float f = (float)v; uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }

template <int MODE>
static void launch(const uint4* frags, float* out, int pairs, int blocks) {
  hipLaunchKernelGGL(k_war<MODE>, dim3(blocks), dim3(1024), 0, 0, frags, out, pairs);
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 200;
  const int pairs = argc > 2 ? atoi(argv[2]) : 1500;           // 2 * pairs fragments of |entries| <= 2: sums < 2^24 (mode 2: x 4)
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount;
  std::vector<uint16_t> h((NFRAG + 1) * 64 * 8);
  std::vector<int> hv(NFRAG * 64 * 8);
  uint32_t seed = 12345;
  for (int i = 0; i < NFRAG * 64 * 8; ++i) { seed = seed * 1664525u + 1013904223u; hv[i] = (int)((seed >> 16) % 5) - 2; h[i] = bf16_of_int(hv[i]); }
  { const float one = 1.0f; for (int i = 0; i < 64 * 4; ++i) memcpy(&h[NFRAG * 64 * 8 + 2 * i], &one, 4); }
  uint4* d_frags; float* d_out;
  const size_t out_floats = (size_t)blocks * 1024 * 16;
  hipMalloc(&d_frags, h.size() * 2); hipMalloc(&d_out, out_floats * 4);
  hipMemcpy(d_frags, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  std::vector<float> ref(out_floats), got(out_floats);
  // reference on the device (mode 0), checked against the host for the 16 waves of block 0
  hipMemset(d_out, 0, out_floats * 4);
  launch<0>(d_frags, d_out, pairs, blocks);
  hipMemcpy(ref.data(), d_out, out_floats * 4, hipMemcpyDeviceToHost);
  if (ref[0] == -12345.0f) { printf("LDS table not at offset 0\n"); return 2; }
  long long host_bad = 0;
  for (int wave = 0; wave < 16; ++wave) {
    std::vector<long long> D(16 * 16, 0);                          // [row][col]
    for (int i = 0; i < 2 * pairs; ++i) {
      const int f = (wave * 7 + 5 * i) % NFRAG;
      for (int row = 0; row < 16; ++row)
        for (int col = 0; col < 16; ++col) {
          long long s = 0;
          for (int k = 0; k < 32; ++k) {
            const int a = hv[(f * 64 + (k / 8) * 16 + row) * 8 + k % 8];              // A: lane (row, k / 8), slot k % 8
            const int lb = (k / 8) * 16 + col, j = k % 8;                             // B: lane (col, k / 8), slot j
            const int bv = (lb * 3 + j) % 5 - 2;
            s += (long long)a * bv;
          }
          D[row * 16 + col] += s;
        }
    }
    for (int lane = 0; lane < 64; ++lane)
      for (int rr = 0; rr < 4; ++rr) {
        const float want = (float)D[(4 * (lane >> 4) + rr) * 16 + (lane & 15)];
        for (int a = 0; a < 4; ++a) if (ref[((size_t)wave * 64 + lane) * 16 + 4 * a + rr] != want) ++host_bad;
      }
  }
  printf("mode 0 (reference) vs host integer matmul, block 0: %lld mismatching values of %d\n", host_bad, 16 * 64 * 16);
  long long self_bad = 0;
  for (size_t i = 0; i < out_floats; ++i) if (ref[i] != ref[i % (1024 * 16)]) ++self_bad;
  printf("mode 0 across %d blocks: %lld values differ from block 0\n", blocks, self_bad);
  const char* names[7] = {"reference, 64 wait states", "SrcA <- ds_read_b128 right behind the MFMA", "same, dependent accumulator chain",
                          "SrcA <- v_mov right behind the MFMA", "SrcA <- global_load right behind the MFMA",
                          "SrcC <- ds_read_b128 3 wait states behind (vDst != SrcC)", "SrcB trashed + restored by v_mov behind the MFMAs"};
  for (int mode = 0; mode < 7; ++mode) {
    long long bad_vals = 0, bad_launches = 0; float worst = 0.0f;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms_total = 0.0f;
    for (int it = 0; it < launches; ++it) {
      hipMemsetAsync(d_out, 0xff, out_floats * 4, 0);
      hipEventRecord(e0);
      switch (mode) {
        case 0: launch<0>(d_frags, d_out, pairs, blocks); break;
        case 1: launch<1>(d_frags, d_out, pairs, blocks); break;
        case 2: launch<2>(d_frags, d_out, pairs, blocks); break;
        case 3: launch<3>(d_frags, d_out, pairs, blocks); break;
        case 4: launch<4>(d_frags, d_out, pairs, blocks); break;
        case 5: launch<5>(d_frags, d_out, pairs, blocks); break;
        default: launch<6>(d_frags, d_out, pairs, blocks); break;
      }
      hipEventRecord(e1);
      hipMemcpy(got.data(), d_out, out_floats * 4, hipMemcpyDeviceToHost);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms_total += ms;
      long long bad = 0;
      for (size_t t = 0; t < out_floats / 16; ++t) {
        const float* g = &got[t * 16]; const float* w = &ref[(t % 1024) * 16];
        for (int i = 0; i < 16; ++i) {
          float want = w[i & 3];                                     // the reference's acc0
          if (mode == 2) want = i < 4 ? 4.0f * want : 0.0f;
          if (g[i] != want) { ++bad; const float d = g[i] - want; if (d * d > worst * worst) worst = d; }
        }
      }
      bad_vals += bad; bad_launches += bad != 0;
    }
    printf("mode %d  %-58s  %lld wrong values in %lld of %d launches (largest error %g)  %.3f ms/launch  %.0f TFLOP/s\n", mode, names[mode],
           bad_vals, bad_launches, launches, worst, ms_total / launches,
           2.0 * 16 * 16 * 32 * 8.0 * pairs * 16 * blocks / (ms_total / launches) * 1e-9);
  }
  return 0;
}
