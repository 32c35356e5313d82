// lrf_backward.inl -- backward of the render path for gfx950 (included by lrf_render.hip).
//
// lrf_render_fwd_train + lrf_render_bwd replace autograd through tensorBase.py:567-636 +
// tensoRF.py:112-196 (paths relative to /root/reference/localTensoRF).
//
// forward with a graph (lrf_render_fwd_train; the same two kernels run at the head of lrf_render_bwd when the caller has
// no saved workspace):
//   k_march (+feat)      density features of every sample, compaction lists (as in the forward)
//   k_shade3<SAVE>       (lrf_shade3.inl) the eval forward's colour kernel itself; additionally saves per shaded sample rgb,
//                        the ReLU masks of both hidden layers as bits, and the activation row ACT = [feat, 1] in MFMA-
//                        fragment order (lrf_common.h).  Hidden activations and plane x line products are not stored.
// backward (two branches on two streams, lrf_render_bwd):
//   k_train_dgrad3       (lrf_train32.inl) per pair of 16-row tiles: d(loss)/d(pre-sigmoid) -> dz2 -> dz1 -> dfeat on TRANSPOSED
//                        weight fragments, split-bf16 on v_mfma_f32_32x32x16_bf16, 32 samples per wave (same register-
//                        resident trick as k_shade3: D layout of one product = B operand of the next), ReLU masks
//                        from the saved bits; dW1 / db1 accumulated in-kernel (transposes on the matrix pipe);
//                        gradient row GRD = [go, dhat | dfeat | .]
//   k_train_app3         (lrf_train32.inl) dfeat -> dX = basis^T dfeat (GRD's dX block, staged through LDS: 1 KB stores);
//                        re-gathers the appearance taps: d/d(position) -> per-tile ray partials, X = plane x line ->
//                        dbasis accumulated in-kernel; tile ids + histogram of the appearance scatter's counting sort
//   k_wgrad_w2w3         dW2 = dz2^T [relu(h1) | 1] and dW3 = go^T [relu(h2) | dhat | 1] from three masked products over
//                        relu(h1) recomputed from the saved feat rows; no h1 / h2 / dz2 rows
//   k_wgrad_reduce       ordered sum of the per-chunk / per-workgroup partials into the reference's layouts (1 launch)
//   k_bwd_ray            one wavefront per ray: weights, d(loss)/d(w), suffix sums ->
//                        d/d(alpha) -> d/d(density feature); position gradients through
//                        normalise / contraction to d(loss)/d(rays); tile ids + histogram of the density scatter's
//                        counting sort
//   k_bin_fill           the rest of the counting sort of the (sample, plane) entries by 32x32-texel tile (scans the histogram itself)
//   k_scatter_plane      plane AND line gradients of one pass over the binned entries, accumulated per workgroup in
//                        LDS (CAS-loop fp32 adds, two channels per 64-bit CAS; ds_add_f32 is 30x slower on this chip),
//                        runs of consecutive same-cell entries merged in registers first, tiles added straight into the
//                        reference layout (k_scatter_line: the fallback where tile + line accumulators exceed LDS)
#pragma once

namespace lrf {

// weight-gradient partial block per K-chunk of rows (W2, W3: k_wgrad_w2w3) / per workgroup (W1: k_train_dgrad3, BAS: k_train_app3), floats
constexpr int WP_W2 = 0;                            // [128][144]  dz2^T [h1r | 1]
constexpr int WP_W1 = WP_W2 + 128 * 144;            // [128][32]   dz1^T [feat | 1]
constexpr int WP_BAS = WP_W1 + 128 * 32;            // [32][96]    dfeat^T X, plane p's 24 channels in columns 32 p .. 32 p + 23
constexpr int WP_W3 = WP_BAS + 32 * 96;             // [16][144]   go^T [h2r | dhat | 1]
constexpr int WP_FLOATS = WP_W3 + 16 * 144;
// rows per K-chunk of the weight-gradient GEMMs: about one chunk per CU (a multiple of 256 rows, at least 512), from
// the number of rows the forward actually produced -- at configs[1] 768 K rows -> 3072: fixed chunk sizes measured
// 2.38 (1024) / 2.35 (2048) / 2.30 (3072) / 2.41 ms (4096) forward+backward.  At most WGRAD_MAXCH chunks exist.
constexpr int WGRAD_MAXCH = 264;
__host__ __device__ inline int wgrad_chunk_rows(int rows) {
  const int per = (rows + 255) / 256;                      // rows / 256 chunks
  return max(512, (per + 255) / 256 * 256);
}

// ---- plane tiles of the binned scatter kernels (see there)
constexpr int BTILE = 32, BCELL = BTILE + 1;
constexpr int BIN_CHUNK = 4096;          // entries per block in the hist/fill passes
constexpr int BIN_MAX = 2048;            // max tiles over the three planes (640^3 -> 1200)
#ifndef LRF_LINE_WGS
#define LRF_LINE_WGS 256
#endif
#ifndef LRF_DPLANE_MULT
#define LRF_DPLANE_MULT 4
#endif
constexpr int LINE_WGS = LRF_LINE_WGS;   // workgroups per line
#ifndef LRF_SCATTER_CAS64
#define LRF_SCATTER_CAS64 1
#endif
#ifndef LRF_APP_NT
#define LRF_APP_NT 1024                  // threads of the appearance scatter's workgroup (one per CU: its tile + line accumulators fill the LDS)
#endif
#ifndef LRF_DENS_LPE
#define LRF_DENS_LPE 4                   // lanes per entry of the density scatter (8 channels): 4 lanes x one pair each (64-bit CAS), 259 -> 224 us against 8 lanes x one channel
#endif

// A workgroup of a scatter kernel pays ~13-24 K cycles per tile it VISITS (zero the accumulators, flush them, three barriers)
// and ~8-10 K per 1024 entries.  Shares of equal ENTRIES leave the workgroups whose share crosses a dozen sparse tiles 3-5
// times behind the rest once the field is trained (profiles/r17_scatter_trained_phases.md: appearance, 621 entries per
// workgroup on average, 1.4 visits -- and one workgroup with 12 visits = the kernel's 187 us).  Shares are therefore cut at
// equal COST, a bin counting SCATTER_VISIT_COST entries more than it holds: k_bin_fill leaves the prefix sums of that cost
// behind the entry offsets, share_entry maps a position on the cost axis back to an entry.
#ifndef LRF_SCATTER_VISIT_COST
#define LRF_SCATTER_VISIT_COST 2048
#endif
constexpr int SCATTER_VISIT_COST = LRF_SCATTER_VISIT_COST;
struct BinGeom { int tx[3], ty[3], base[3], total; };
__host__ __device__ inline BinGeom make_bins(const Layout& L) {
  BinGeom b; int off = 0;
  for (int p = 0; p < 3; ++p) {
    b.tx[p] = (L.pw[p] + BTILE - 1) / BTILE; b.ty[p] = (L.ph[p] + BTILE - 1) / BTILE;
    b.base[p] = off; off += b.tx[p] * b.ty[p];
  }
  b.total = off;
  return b;
}

// tap1d + d(ix)/d(u): (size-1)/2 inside, 0 where ATen's clip_coordinates_set_grad zeroes it
__device__ __forceinline__ void tap1d_g(float u, int size, int& i0, int& i1, float& t, float& gmul) {
  const float raw = ((u + 1.0f) * 0.5f) * (float)(size - 1);
  const float hi = (float)(size - 1);
  gmul = (raw <= 0.0f || raw >= hi) ? 0.0f : 0.5f * hi;
  const float ix = fminf(fmaxf(raw, 0.0f), hi);
  const float f0 = floorf(ix);
  t = ix - f0;
  i0 = (int)f0;
  i1 = min(i0 + 1, size - 1);
}

// backward of contract3 (utils/ray_utils.py:9-12): xr = uncontracted position, g = grad wrt
// the contracted position (in/out: becomes grad wrt xr)
__device__ __forceinline__ void contract3_bwd(const float xr[3], float g[3]) {
  const float a0 = fabsf(xr[0]), a1 = fabsf(xr[1]), a2 = fabsf(xr[2]);
  const float m = fmaxf(fmaxf(fmaxf(a0, a1), a2), 1e-6f);
  if (m > 1.0f) {
    const float s = (2.0f * m - 1.0f) / (m * m);
    const float sp = (2.0f - 2.0f * m) / (m * m * m);
    const float dot = xr[0] * g[0] + xr[1] * g[1] + xr[2] * g[2];
    const int am = (a0 >= a1 && a0 >= a2) ? 0 : (a1 >= a2 ? 1 : 2);
    g[0] *= s; g[1] *= s; g[2] *= s;
    g[am] += (xr[am] >= 0.0f ? 1.0f : -1.0f) * sp * dot;
  }
}

__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }
// fp32 add into LDS.  The native ds_add_f32 retires ~0.33 lanes/clk/CU on MI355X (203 G lane-ops/s
// chip-wide, scripts/ubench/lds_atomic.hip) -- 30x below ds_add_u32 -- while a compare-and-swap
// loop on ds_cmpst_rtn_b32 sustains 1.9-2.7 T lane-ops/s at the collision rates seen here.
__device__ __forceinline__ void lds_add_f32(float* p, float v) {
  unsigned* q = reinterpret_cast<unsigned*>(p);
  unsigned old = *q, assumed;
  do {
    assumed = old;
    old = atomicCAS(q, assumed, __float_as_uint(__uint_as_float(assumed) + v));
  } while (old != assumed);
}

// Four independent adds with their LDS round trips overlapped; a failed compare (another lane, or
// two of the four addresses coinciding at a clamped border) falls back to the loop.
__device__ __forceinline__ void lds_add_retry(unsigned* q, unsigned seen, float v) {
  unsigned assumed;
  do {
    assumed = seen;
    seen = atomicCAS(q, assumed, __float_as_uint(__uint_as_float(assumed) + v));
  } while (seen != assumed);
}
__device__ __forceinline__ void lds_add4_f32(float* p0, float v0, float* p1, float v1, float* p2, float v2,
                                             float* p3, float v3) {
  unsigned* q0 = reinterpret_cast<unsigned*>(p0); unsigned* q1 = reinterpret_cast<unsigned*>(p1);
  unsigned* q2 = reinterpret_cast<unsigned*>(p2); unsigned* q3 = reinterpret_cast<unsigned*>(p3);
  const unsigned o0 = *q0, o1 = *q1, o2 = *q2, o3 = *q3;
  const unsigned r0 = atomicCAS(q0, o0, __float_as_uint(__uint_as_float(o0) + v0));
  const unsigned r1 = atomicCAS(q1, o1, __float_as_uint(__uint_as_float(o1) + v1));
  const unsigned r2 = atomicCAS(q2, o2, __float_as_uint(__uint_as_float(o2) + v2));
  const unsigned r3 = atomicCAS(q3, o3, __float_as_uint(__uint_as_float(o3) + v3));
  if (r0 != o0) lds_add_retry(q0, r0, v0);
  if (r1 != o1) lds_add_retry(q1, r1, v1);
  if (r2 != o2) lds_add_retry(q2, r2, v2);
  if (r3 != o3) lds_add_retry(q3, r3, v3);
}

// Loads from the saved rows.  Rows are written once (plain stores: with rows of 128 + 512 B the `nt` store hint of rounds
// 2-3 no longer pays anywhere, profiles/r11 s3) and read by one or two later kernels with the `nt` hint, so that they do not
// displace the field's texels in L2 under the gathers running beside them.  LRF_ROW_NT bits: 4 = the feat / go row loads of
// k_train_dgrad3 and k_wgrad_w2w3, 16 = the scatter kernels' dX loads.  Default: both.
#ifndef LRF_ROW_NT
#define LRF_ROW_NT 63
#endif
template <int BIT>
__device__ __forceinline__ float2 row_load2_b(const float2* p) {
  if constexpr ((LRF_ROW_NT & BIT) != 0) {
    typedef float f32x2s __attribute__((ext_vector_type(2)));
    const f32x2s v = __builtin_nontemporal_load(reinterpret_cast<const f32x2s*>(p));
    return make_float2(v[0], v[1]);
  } else {
    return *p;
  }
}
__device__ __forceinline__ float4 row_load4(const float* p) {
#if LRF_ROW_NT & 4
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
#else
  return *reinterpret_cast<const float4*>(p);
#endif
}
// dX of (row, plane p, lane group sub): six channels 24 p + 6 sub .. + 5 from the GRD tile's row-major dX block.
// (Round 2, 16-sample data-gradient kernel: slot order like the X block -- five coalesced float4 stores per lane instead of nine 8-byte
// ones -- was measured: dgrad 282 -> 262 us, but k_scatter_line<24> 178 -> 256 us, whose lanes walk the rows in
// order and then read 24 B out of every 1 KB block; forward+backward 1.99 vs 1.94 ms.  Not adopted.)
// (Also measured in round 2: the data-gradient kernel storing dX * L and dX * P -- it holds both factors when it forms the position
// gradient -- so that the appearance scatter kernels need not re-gather the other factor's taps: 288 B more per row,
// gradients unchanged (74 tests), forward+backward 1.94 vs 1.79 ms.  The scatters are bound by their LDS adds, not by
// those gathers.  Not adopted.)
__device__ __forceinline__ void load_dx6(const float* __restrict__ grd, size_t row, int p, int sub, float dv[6]) {
  const float2* dx2 = reinterpret_cast<const float2*>(grd_dx_row(grd, row) + p * LRF_CA + 6 * sub);       // (rows order: lrf_common.h)
#pragma unroll
  for (int h = 0; h < 3; ++h) { const float2 t2 = row_load2_b<16>(dx2 + h); dv[2 * h] = t2.x; dv[2 * h + 1] = t2.y; }
}

// the same for PAIRS of adjacent floats (8-byte aligned): one 64-bit compare-and-swap adds two channels -- half the LDS
// instructions of the appearance scatter's flushes
__device__ __forceinline__ unsigned long long pack_f2(float a, float b) {
  return (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
}
__device__ __forceinline__ unsigned long long add_f2(unsigned long long o, float a, float b) {
  return pack_f2(__uint_as_float((unsigned)o) + a, __uint_as_float((unsigned)(o >> 32)) + b);
}
__device__ __forceinline__ void lds_add_retry2(unsigned long long* q, unsigned long long seen, float a, float b) {
  unsigned long long assumed;
  do {
    assumed = seen;
    seen = atomicCAS(q, assumed, add_f2(assumed, a, b));
  } while (seen != assumed);
}
__device__ __forceinline__ void lds_add4_f2(float* p0, float a0, float b0, float* p1, float a1, float b1,
                                            float* p2, float a2, float b2, float* p3, float a3, float b3) {
  unsigned long long* q0 = reinterpret_cast<unsigned long long*>(p0); unsigned long long* q1 = reinterpret_cast<unsigned long long*>(p1);
  unsigned long long* q2 = reinterpret_cast<unsigned long long*>(p2); unsigned long long* q3 = reinterpret_cast<unsigned long long*>(p3);
  const unsigned long long o0 = *q0, o1 = *q1, o2 = *q2, o3 = *q3;
  const unsigned long long r0 = atomicCAS(q0, o0, add_f2(o0, a0, b0));
  const unsigned long long r1 = atomicCAS(q1, o1, add_f2(o1, a1, b1));
  const unsigned long long r2 = atomicCAS(q2, o2, add_f2(o2, a2, b2));
  const unsigned long long r3 = atomicCAS(q3, o3, add_f2(o3, a3, b3));
  if (r0 != o0) lds_add_retry2(q0, r0, a0, b0);
  if (r1 != o1) lds_add_retry2(q1, r1, a1, b1);
  if (r2 != o2) lds_add_retry2(q2, r2, a2, b2);
  if (r3 != o3) lds_add_retry2(q3, r3, a3, b3);
}
__device__ __forceinline__ void lds_add2_f2(float* p0, float a0, float b0, float* p1, float a1, float b1) {
  unsigned long long* q0 = reinterpret_cast<unsigned long long*>(p0); unsigned long long* q1 = reinterpret_cast<unsigned long long*>(p1);
  const unsigned long long o0 = *q0, o1 = *q1;
  const unsigned long long r0 = atomicCAS(q0, o0, add_f2(o0, a0, b0));
  const unsigned long long r1 = atomicCAS(q1, o1, add_f2(o1, a1, b1));
  if (r0 != o0) lds_add_retry2(q0, r0, a0, b0);
  if (r1 != o1) lds_add_retry2(q1, r1, a1, b1);
}

// ---------------------------------------------------------------- helpers of the data-gradient / scatter kernels
// aligned 16-byte load from a pointer known to be global memory
__device__ __forceinline__ void ld4g(const float* p, float* out) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef f32x4 __attribute__((address_space(1))) gf32x4;
  const f32x4 v = *(const gf32x4*)p;
#else
  const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#endif
  out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
}
// x where bit k of the saved ReLU mask is set, else +0: sign-extended one-bit field (0 / ~0) ANDed into the value
__device__ __forceinline__ float relu_gate(float x, uint32_t bits, int k) {
  return __uint_as_float(__float_as_uint(x) & (uint32_t)__builtin_amdgcn_sbfe((int)bits, k, 1));
}
// ---------------------------------------------------------------- weight gradients
// dW2 (+ db2) = dz2^T [relu(h1) | 1] and dW3 (+ db3) = go^T [relu(h2) | dhat | 1] with NEITHER hidden activation stored
// and without recomputing layer 2.  Round 3 read 576 B of h1 and 576 B of h2 per shaded sample here (and the forward
// wrote them: 1.8 GB per step at BASELINE configs[1]).  With m2 = [h2 > 0] (the saved mask bits), B = [relu(h1) | 1]:
//     T_c[u][v]  = sum_rows go[row][c] m2[row][u] B[row][v]                       (three 128 x 129 products, K = rows)
//     dW2[u][v]  = sum_c W3[c][u] T_c[u][v],   db2[u] = sum_c W3[c][u] T_c[u][128]          (dz2 = m2 * W3^T go)
//     dW3[c][u]  = sum_v W2[u][v] T_c[u][v] + b2[u] T_c[u][128]                   (relu(h2) = m2 * (W2 relu(h1) + b2))
// so one GEMM family over the rows yields both gradients; relu(h1) is recomputed from the 108 B `feat` row with split-bf16
// layer-1 fragments (the forward's products, summed in the 16-sample kernels' order), and h2 never exists here.  The first version of this
// round recomputed layer 2 as well and contracted dW3 on the VALU at one wave per SIMD: 392 us, issue-bound (1450 VALU +
// 230 MFMA + 305 LDS instructions per 64-row step and wave, profiles/r11a).  This one:
//   * 512 threads, two waves per SIMD; a step is 128 rows = 8 tiles, one per wave for layer 1; relu(h1) goes to LDS
//     TRANSPOSED and already split, s_bh / s_bl [column][128 rows (+8 pad)] bf16, so a lane's B operand of one
//     v_mfma_f32_16x16x32_bf16 (rows 8 g .. 8 g + 7 of column 16 n + i) is one ds_read_b128 per half;
//   * wave w owns M-tile w (units 16 w .. 16 w + 15) of all three T_c: 27 accumulator tiles; its A operand
//     go[row][c] m2[row][u] is go's pre-split halves (s_ga, one ds_read_b128 per colour and half) ANDed with a mask
//     expanded from one byte of row bits -- the staging waves transpose the mask dwords with ballots (s_mt[unit][row]);
//   * 81 independent MFMAs per 32 rows and wave against ~60 VALU: the matrix pipe is what the kernel waits for;
//   * the W3 / W2 contractions of the epilogue run once per chunk, in registers; the view / bias columns of dW3 come
//     from the (dhat, 1) quadruple the data-gradient kernel left in the go block.
// Rows behind the chunk's end read tile r0 again; their go and mask bits are zeroed in LDS, so they contribute nothing.
// (Measured and dropped: 64-row steps double-buffered in LDS, four alternating waves staging step i + 1 while all eight
// multiply step i, one barrier per step -- 228 us against 200 us: twice the barriers, and the staging waves hold the others up.)
// KT rows per step.  128: the form above -- 27 accumulator tiles + operands are ~240 registers, two waves per SIMD is all the
// register file holds, and all eight wait at both barriers of a step while nobody multiplies (the matrix pipe is busy 53 % of
// the kernel, profiles/r18_pmc_train.md).  64: the step's operands are DOUBLE-BUFFERED in LDS (2 x 43.6 KB) and every wave
// stages its share of step i + 1 itself -- a 16-row tile's layer 1 is split over two waves, four unit tiles each -- but waves
// 0-3 stage first and multiply second while waves 4-7 (the other wave of each SIMD) multiply first and stage second: one
// barrier per step, and on every SIMD one wave's VALU / LDS staging runs under the other's MFMAs.
// (Measured and dropped, round 6: four-wave workgroups owning half of the units each, two per CU, single-buffered 64-row
// steps -- the rows staged twice per CU: forward + backward 1.19 -> 1.30 ms.)
constexpr int W23_NT = 9;
__host__ __device__ constexpr size_t w23_lds(int KT) {
  return (size_t)(8 * 128) * 16 + 128 * 4
       + (KT == 64 ? 2 : 1) * (2 * (size_t)(W23_NT * 16) * (KT + 8) * 2 + (size_t)6 * KT * 2 + (size_t)LRF_FEATC * (KT / 32) * 4);
}
template <int KT>
__global__ __launch_bounds__(512) void k_wgrad_w2w3(const uint4* __restrict__ mlpb, const float* __restrict__ feat /* act + 16 * ACT_FEAT */, int lda,
                                                    const float* __restrict__ go /* grd + 16 * GRD_GO */, int ldg,
                                                    const uint32_t* __restrict__ relu_bits, const float* __restrict__ w3 /* [3][131] */,
                                                    const float* __restrict__ w2 /* [128][128] */, const float* __restrict__ b2,
                                                    const int* __restrict__ toff, int R, float* __restrict__ wpart) {
  constexpr int NT = W23_NT, WB = NT * 16, CS = KT + 8;
  constexpr bool DB = KT == 64;                                   // double-buffered step operands
  constexpr int TPS = KT / 16, WPT = 8 / TPS, MT1 = 8 / WPT;     // tiles per step, waves per tile, layer-1 unit tiles per wave
  constexpr int WPG = KT / 64;                                    // waves per lane group of mask dwords
  constexpr int BUF_BYTES = 2 * WB * CS * 2 + 6 * KT * 2 + LRF_FEATC * (KT / 32) * 4;
  extern __shared__ uint4 s_w23[];
  uint4* s_w1 = s_w23;                                              // W1 fragments [t' 8][hi, lo][lane 64]
  float* s_b1 = reinterpret_cast<float*>(s_w1 + 8 * 128);
  char* s_buf0 = reinterpret_cast<char*>(s_b1 + 128);
  __bf16* s_bh = reinterpret_cast<__bf16*>(s_buf0);                // [144 columns][CS rows]
  __bf16* s_bl = s_bh + WB * CS;
  __bf16* s_ga = s_bl + WB * CS;                                    // [colour 3][hi, lo][KT rows]: go, pre-split
  uint32_t* s_mt = reinterpret_cast<uint32_t*>(s_ga + 6 * KT);     // [unit 128][KT / 32]: bit (row % 32) of dword row / 32 = m2[row][unit]
  auto use_buf = [&](int b) {                                       // (DB) point the four arrays at buffer b
    s_bh = reinterpret_cast<__bf16*>(s_buf0 + (size_t)b * BUF_BYTES);
    s_bl = s_bh + WB * CS;
    s_ga = s_bl + WB * CS;
    s_mt = reinterpret_cast<uint32_t*>(s_ga + 6 * KT);
  };
  const int rows = toff[R] * 16;
  const int WGRAD_CH = wgrad_chunk_rows(rows);
  const int chunk = (int)blockIdx.x;
  const int r0 = chunk * WGRAD_CH;
  if (r0 >= rows) return;
  const int r1 = min(r0 + WGRAD_CH, rows);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, g = lane >> 4;
  const int mt = wave;                                              // this wave's M-tile: units 16 mt .. 16 mt + 15
  for (int q = tid; q < 8 * 128; q += 512) s_w1[q] = mlpb[IMGB_W1 + q];
  if (tid < 128) s_b1[tid] = reinterpret_cast<const float*>(mlpb + IMGB_TAIL)[TAIL_B1 + tid];
  for (int b = 0; b < (DB ? 2 : 1); ++b) {
    use_buf(b);
    for (int q = tid; q < 16 * CS; q += 512) {                      // the bias block of B: column 128 = 1, 129 .. 143 = 0
      s_bh[128 * CS + q] = (__bf16)(q < CS ? 1.0f : 0.0f);
      s_bl[128 * CS + q] = (__bf16)0.0f;
    }
  }
  use_buf(0);
  f32x4 acc[3][NT];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[c][n] = f32x4{0, 0, 0, 0};
  float xv[3][4];                                                  // threads < KT: sum over their rows of go[c] * (dhat, 1)[a]
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int a = 0; a < 4; ++a) xv[c][a] = 0.0f;
  struct Pre { float4 f0, f1, go, dh; uint32_t m; };
  Pre pre;
  auto fetch = [&](Pre& p, int rb) {
    {                                                      // feat of this wave's tile, as the forward's D registers: blocks 0, 1 of the row
      int tile = (rb >> 4) + (wave % TPS);
      if (tile * 16 >= r1) tile = r0 >> 4;
      const float* fp = feat + (size_t)tile * (size_t)(16 * lda) + (lane << 2);
      p.f0 = row_load4(fp);
      p.f1 = row_load4(fp + 256);
    }
    {                                                      // go / (dhat, 1) of row rb + tid % KT: lane groups 0 and 1 of the go block (KT = 64: waves 4-7 repeat 0-3's, branch-free)
      const int row = rb + (tid & (KT - 1));
      const int rowc = row < r1 ? row : r0;
      const float* gp = go + (size_t)(rowc >> 4) * (size_t)(16 * ldg) + ((rowc & 15) << 2);
      p.go = *reinterpret_cast<const float4*>(gp);
      p.dh = *reinterpret_cast<const float4*>(gp + 64);
      // layer-2 mask dword (row, lane group gg = tid / 128): tile row / 16, lane (row % 16) + 16 gg
      p.m = relu_bits[((size_t)(rowc >> 4) * 2 + 1) * 64 + (rowc & 15) + 16 * ((tid / KT) & 3)];
    }
  };
  auto stage = [&](const Pre& p, int rb) {
    if (!DB) __syncthreads();                              // previous step's products have read LDS
    // ---- layer 1 of the forward on this wave's 16 rows (the 16-sample fragments of the exact-order engines; k_shade3 sums the same products in another order: relu(h1) differs from the forward's in the last bit at most)
    const int tl = wave % TPS, mb = (wave / TPS) * MT1;      // this wave's tile of the step, its first unit tile
    f32x4 h1[MT1];
#pragma unroll
    for (int t1 = 0; t1 < MT1; ++t1) h1[t1] = *reinterpret_cast<const f32x4*>(&s_b1[16 * (mb + t1) + 4 * g]);
    {
      const float v[8] = {p.f0.x, p.f0.y, p.f0.z, p.f0.w, p.f1.x, p.f1.y, p.f1.z, p.f1.w};
      bf16x8 bh, bl;
      split8(v, bh, bl);
      gemm_step<MT1>(s_w1, mb, 1, lane, bh, bl, h1);
    }
#pragma unroll
    for (int t1 = 0; t1 < MT1; ++t1)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = fmaxf(h1[t1][r], 0.0f);
        const __bf16 h = (__bf16)v;
        const int at = (16 * (mb + t1) + 4 * g + r) * CS + 16 * tl + i;
        s_bh[at] = h;
        s_bl[at] = (__bf16)(v - (float)h);
      }
    const bool ok = rb + (tid & (KT - 1)) < r1;
    if (tid < KT) {                                        // go of row tid, split once for all eight product waves
      const float gv[3] = {ok ? p.go.x : 0.0f, ok ? p.go.y : 0.0f, ok ? p.go.z : 0.0f};
      const float dv[4] = {p.dh.x, p.dh.y, p.dh.z, p.dh.w};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const __bf16 h = (__bf16)gv[c];
        s_ga[(2 * c) * KT + tid] = h;
        s_ga[(2 * c + 1) * KT + tid] = (__bf16)(gv[c] - (float)h);
#pragma unroll
        for (int a = 0; a < 4; ++a) xv[c][a] += gv[c] * dv[a];
      }
    }
    if (wave < 4 * WPG) {                                  // mask bits, transposed: wave w holds lane group gg = w / WPG of rows 64 (w % WPG) + lane
      const uint32_t m = ok ? p.m : 0u;
      const int gg = wave / WPG, half = wave % WPG;
#pragma unroll
      for (int b = 0; b < 32; ++b) {                       // bit b = unit 16 (b / 4) + 4 gg + b % 4
        const unsigned long long bal = __ballot((m >> b) & 1u);
        if (lane == b) {
          const int u = 16 * (b >> 2) + 4 * gg + (b & 3);
          *reinterpret_cast<uint2*>(&s_mt[u * (KT / 32) + 2 * half]) = make_uint2((uint32_t)bal, (uint32_t)(bal >> 32));
        }
      }
    }
    if (!DB) __syncthreads();
  };
  auto compute = [&]() {
#pragma unroll (DB ? 2 : 1)
    for (int kk = 0; kk < KT / 32; ++kk) {
      uint4 ar[3][2];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        ar[c][0] = *reinterpret_cast<const uint4*>(&s_ga[(2 * c) * KT + 32 * kk + 8 * g]);
        ar[c][1] = *reinterpret_cast<const uint4*>(&s_ga[(2 * c + 1) * KT + 32 * kk + 8 * g]);
      }
      // rows 32 kk + 8 g + j, j = 0 .. 7: bit j of this byte says whether unit 16 wave + i was active in row j
      const uint32_t byte = (s_mt[(16 * mt + i) * (KT / 32) + kk] >> (8 * g)) & 0xffu;
      uint32_t pm[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {                        // K slots 2 q, 2 q + 1 share a dword of the operand
        const uint32_t lo = (uint32_t)((int32_t)(byte << (31 - 2 * q)) >> 31), hi = (uint32_t)((int32_t)(byte << (30 - 2 * q)) >> 31);
        pm[q] = (lo & 0xffffu) | (hi & 0xffff0000u);
      }
      bf16x8 ah[3], al[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        ah[c] = __builtin_bit_cast(bf16x8, make_uint4(ar[c][0].x & pm[0], ar[c][0].y & pm[1], ar[c][0].z & pm[2], ar[c][0].w & pm[3]));
        al[c] = __builtin_bit_cast(bf16x8, make_uint4(ar[c][1].x & pm[0], ar[c][1].y & pm[1], ar[c][1].z & pm[2], ar[c][1].w & pm[3]));
      }
      // three N-tiles at a time (24 operand registers live instead of 72), term-major over their nine accumulators:
      // consecutive MFMAs never depend on each other
#pragma unroll
      for (int n0 = 0; n0 < NT; n0 += 3) {
        bf16x8 bh[3], bl[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) {
          bh[n] = *reinterpret_cast<const bf16x8*>(&s_bh[(16 * (n0 + n) + i) * CS + 32 * kk + 8 * g]);
          bl[n] = *reinterpret_cast<const bf16x8*>(&s_bl[(16 * (n0 + n) + i) * CS + 32 * kk + 8 * g]);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int n = 0; n < 3; ++n) acc[c][n0 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[c], bh[n], acc[c][n0 + n], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int n = 0; n < 3; ++n) acc[c][n0 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[c], bl[n], acc[c][n0 + n], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int n = 0; n < 3; ++n) acc[c][n0 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[c], bh[n], acc[c][n0 + n], 0, 0, 0);
      }
    }
  };
  fetch(pre, r0);
  if constexpr (!DB) {
    for (int rb = r0; rb < r1; rb += KT) {
      stage(pre, rb);
      fetch(pre, rb + KT);
      compute();
    }
  } else {
    __syncthreads();                                       // the W1 fragments and b1 are in LDS
    stage(pre, r0);                                        // step 0 into buffer 0
    fetch(pre, r0 + KT);
    __syncthreads();
    int cur = 0;
    for (int rb = r0; rb < r1; rb += KT, cur ^= 1) {
      const bool more = rb + KT < r1;
      if (wave < 4) {                                      // stage step i + 1, then multiply step i ...
        if (more) { use_buf(cur ^ 1); stage(pre, rb + KT); fetch(pre, rb + 2 * KT); }
        use_buf(cur); compute();
      } else {                                             // ... while the SIMD's other wave does it the other way round
        use_buf(cur); compute();
        if (more) { use_buf(cur ^ 1); stage(pre, rb + KT); fetch(pre, rb + 2 * KT); }
      }
      __syncthreads();
    }
    use_buf(0);
  }
  // ---- epilogue, once per chunk: acc[c][n][r] = T_c[u = 16 wave + 4 g + r][v = 16 n + i]
  float* o2 = wpart + (size_t)chunk * WP_FLOATS + WP_W2;
  float* o3 = wpart + (size_t)chunk * WP_FLOATS + WP_W3;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int u = 16 * mt + 4 * g + r;
    const float w3r = w3[u], w3g = w3[131 + u], w3b = w3[262 + u];
    float d3[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      o2[(size_t)u * (NT * 16) + 16 * n + i] = w3r * acc[0][n][r] + w3g * acc[1][n][r] + w3b * acc[2][n][r];
      const float wv = n < 8 ? w2[u * LRF_FEATC + 16 * n + i] : (i == 0 ? b2[u] : 0.0f);
#pragma unroll
      for (int c = 0; c < 3; ++c) d3[c] += wv * acc[c][n][r];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int dd = 1; dd < 16; dd <<= 1) d3[c] += __shfl_xor(d3[c], dd, 64);
      if (i == 0) o3[c * 144 + u] = d3[c];
    }
  }
  __syncthreads();                                         // the view / bias columns: 128 row threads -> 12 sums (LDS reused)
  float* s_x = reinterpret_cast<float*>(s_bh);
  if (tid < KT) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float v = wave_sum(xv[c][a]);
        if (lane == 0) s_x[wave * 12 + 4 * c + a] = v;
      }
  }
  __syncthreads();
  if (tid < 12) o3[(tid >> 2) * 144 + LRF_FEATC + (tid & 3)] = s_x[tid] + (KT > 64 ? s_x[12 + tid] : 0.0f);
}

// dst[m*dst_ld + n] += sum_chunks part[chunk][off + m*ld + n_off + n]   (chunks in a fixed order) for
// the seven weight / bias tensors in one launch: 16 lanes per output element.
struct WgradSeg { int off, ld, n_off, m_count, n_count, dst_ld, first_elem, x_slots, nch; float* dst; };   // x_slots: source column of channel n is 32 (n / 24) + n % 24 (WP_BAS); nch: partial blocks (0: one per K-chunk of rows)
struct WgradSegs { WgradSeg s[7]; int total_elems; };
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ wpart, const int* __restrict__ toff, int R,
                                                      WgradSegs segs) {
  const int idx = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, sub = threadIdx.x & 15;
  const bool ok = idx < segs.total_elems;
  int k = 0;
#pragma unroll
  for (int q = 1; q < 7; ++q) k += (ok && idx >= segs.s[q].first_elem) ? 1 : 0;
  const WgradSeg sg = segs.s[k];
  const int e = ok ? idx - sg.first_elem : 0;
  const int m = e / sg.n_count, n = e % sg.n_count;
  const int WGRAD_CH = wgrad_chunk_rows(toff[R] * 16);
  const int nch = sg.nch ? sg.nch : (toff[R] * 16 + WGRAD_CH - 1) / WGRAD_CH;
  float acc = 0.0f;
  const int src = sg.off + m * sg.ld + sg.n_off + (sg.x_slots ? 32 * (n / 24) + n % 24 : n);
  for (int c = sub; c < nch; c += 16) acc += wpart[(size_t)c * WP_FLOATS + src];
#pragma unroll
  for (int d = 8; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);          // fixed tree: deterministic
  if (ok && sub == 0) sg.dst[m * sg.dst_ld + n] += acc;
}

// ---------------------------------------------------------------- per-ray backward
// tensorBase.py:584-615,632-634 backwards: d(loss)/d(w) -> d/d(alpha) -> d/d(sigma feature),
// density plane/line gradients, d/d(rays).
// (the plane loop is rolled: one plane's taps live at a time, 238 -> 137 VGPRs = 3 instead of 2 waves per SIMD; capping it
// at 128 VGPRs / 4 waves changed nothing measurable in the two-branch backward: 2.94-2.99 ms either way)
__global__ __launch_bounds__(256) void k_bwd_ray(
    DField f, const float* __restrict__ rays, const float* __restrict__ z, int R, int S, uint32_t flags,
    float* __restrict__ feat /* in: density feature; out: d(loss)/d(feature) */,
    const int* __restrict__ ncomp, const uint16_t* __restrict__ cidx,
    const float* __restrict__ crgb, const float* __restrict__ g_rgb, const float* __restrict__ g_depth,
    const float* __restrict__ rpart, int pmax, float* __restrict__ g_rays,
    BinGeom bg, uint16_t* __restrict__ tid /* [3][nmax] plane-tile id of every density entry, 0xffff = none */,
    int* __restrict__ hist /* += entries per tile */, uint32_t nmax,
    unsigned* __restrict__ vmax_bits /* max over samples, planes, channels of |g line| and |g plane| as float bits (atomicMax): every
                                        contribution the fixed-point density scatter adds is at most this (k_scatter_fix) */) {
  extern __shared__ float s_all[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + wave;
  // first pass of the density scatter's counting sort (it was a kernel of its own that re-derived every sample's position):
  // the workgroup's histogram over the plane tiles, behind the four per-ray arrays (one per wave was measured: four times the
  // global atomics at the end, 80 -> 120 us)
  int* s_h = reinterpret_cast<int*>(s_all + (size_t)14 * S);      // (per wave: alpha, w, gw [S] floats + idx [S] shorts = 14 S bytes x 4 waves / 4)
  for (int i = threadIdx.x; i < bg.total; i += 256) s_h[i] = 0;
  __syncthreads();
  auto flush_hist = [&]() {                                    // every wave of the workgroup, behind the one barrier at the end
    __syncthreads();
    for (int i = threadIdx.x; i < bg.total; i += 256) {
      const int v = s_h[i];
      if (v) atomicAdd(&hist[i], v);
    }
  };
  if (ray >= R) { flush_hist(); return; }                      // (a wave without a ray still meets the others at that barrier)
  float* s_alpha = s_all + (size_t)wave * 3 * S;
  float* s_w = s_alpha + S;
  float* s_gw = s_w + S;
  // compact index of a sample among the ray's shaded ones (< S <= 2048) or -1, as shorts behind the three float arrays of all
  // four waves: 29.9 KB per workgroup at S = 512 instead of 34 KB -- five workgroups per CU instead of four
  short* s_idx = reinterpret_cast<short*>(s_all + (size_t)12 * S) + (size_t)wave * S;
  const float* rp = rays + (size_t)ray * 6;
  const float o[3] = {rp[0], rp[1], rp[2]};
  const float dn = sqrtf(rp[3] * rp[3] + rp[4] * rp[4] + rp[5] * rp[5]);
  const float dh[3] = {rp[3] / dn, rp[4] / dn, rp[5] / dn};
  const bool relu = flags & LRF_FLAG_RELU_DENS;
  const float white = (flags & LRF_FLAG_WHITE_BG) ? 1.0f : 0.0f;
  const int oray = f.perm ? f.perm[ray] : ray;              // the caller's index of this slot (ray sorting)
  const float gr[3] = {g_rgb[(size_t)oray * 3], g_rgb[(size_t)oray * 3 + 1], g_rgb[(size_t)oray * 3 + 2]};
  const float gsum = (gr[0] + gr[1] + gr[2]) * white;
  const float gd = g_depth[oray];
  const int nchunk = (S + 63) >> 6;
  const int nsh = ncomp[ray];

  for (int k = lane; k < S; k += 64) s_idx[k] = -1;
  for (int j = lane; j < nsh; j += 64) s_idx[cidx[(size_t)ray * S + j]] = (short)j;
  // alpha per sample (same arithmetic as k_march).  This kernel is one latency chain per wave (every ray's wave is resident at
  // once: its duration is the time ONE wave needs): the loops over the ray's 64-sample chunks therefore issue the loads of
  // four chunks together instead of paying a round trip per chunk (k_bwd_ray 63 -> us in a captured 300^3 iteration).
  const float* frow = feat + (size_t)ray * S;
  for (int c0 = 0; c0 < nchunk; c0 += 4) {
    float fk4[4], z04[4], z14[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = ((c0 + u) << 6) + lane, kc = min(k, S - 1);
      fk4[u] = frow[kc]; z04[u] = z[kc]; z14[u] = z[min(k + 1, S - 1)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = ((c0 + u) << 6) + lane;
      if (k < S) {
        float alpha = 0.0f;
        if (k < S - 1 && fk4[u] > -INFINITY) {
          const float sigma = feature2density(fk4[u], f.density_shift, relu);
          alpha = 1.0f - expf(-sigma * (z14[u] - z04[u]) * f.distance_scale);
        }
        if (k == S - 1) alpha = 1.0f;
        s_alpha[k] = alpha;
      }
    }
  }
  // weights and d(loss)/d(w_k);  total = sum_k gw_k w_k
  float carry = 1.0f, tot = 0.0f, dsum = 0.0f;
  for (int c0 = 0; c0 < nchunk; c0 += 4) {
    float zk4[4], col4[4];                                     // z and g_rgb . colour of four chunks' samples, requested together
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = ((c0 + u) << 6) + lane;
      zk4[u] = z[min(k, S - 1)];
      col4[u] = 0.0f;
      const int j = k < S ? (int)s_idx[k] : -1;
      if (j >= 0) {
        const float* cp = crgb + ((size_t)ray * S + j) * 3;
        col4[u] = gr[0] * cp[0] + gr[1] * cp[1] + gr[2] * cp[2];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + u;
      if (c >= nchunk) break;
      const int k = (c << 6) + lane;
      const float alpha = k < S ? s_alpha[k] : 0.0f;
      const float v = k < S ? (1.0f - alpha + 1e-10f) : 1.0f;
      float excl, total;
      wave_scan_prod(v, lane, excl, total);
      const float T = carry * excl;
      carry *= total;
      const float w = alpha * T;
      if (k < S) {
        const float gw = (gd * zk4[u] / dn - gsum) + col4[u];
        s_w[k] = w;
        s_gw[k] = gw;
        tot += gw * w;
        dsum += w * zk4[u];
      }
    }
  }
  tot = wave_sum(tot);
  dsum = wave_sum(dsum);
  // d/d(alpha_k) = gw_k T_k - (sum_{j>k} gw_j w_j) / (1 - alpha_k + 1e-10)   (alpha2weights,
  // tensorBase.py:23-32), then alpha -> sigma -> feature (:610, :495-499).  T_k comes from a
  // second product scan, the suffix sum from a running inclusive sum of gw*w.
  float go3[3] = {0.0f, 0.0f, 0.0f}, gdh[3] = {0.0f, 0.0f, 0.0f};
  float run = 0.0f;
  carry = 1.0f;
  for (int c0 = 0; c0 < nchunk; c0 += 4) {
    float fk4[4], z04[4], z14[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = ((c0 + u) << 6) + lane, kc = min(k, S - 1);
      fk4[u] = frow[kc]; z04[u] = z[kc]; z14[u] = z[min(k + 1, S - 1)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
    const int c = c0 + u;
    if (c >= nchunk) break;
    const int k = (c << 6) + lane;
    const float alpha = k < S ? s_alpha[k] : 0.0f;
    const float vk = k < S ? (1.0f - alpha + 1e-10f) : 1.0f;
    float excl, total;
    wave_scan_prod(vk, lane, excl, total);
    const float Tk = carry * excl;
    carry *= total;
    const float gww = k < S ? s_gw[k] * s_w[k] : 0.0f;
    float incl = gww;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    const float chunk_total = __shfl(incl, 63, 64);
    float gf = 0.0f;
    if (k < S - 1) {
      const float fk = fk4[u];
      if (fk > -INFINITY) {
        const float suffix = tot - (run + incl);
        const float dalpha = s_gw[k] * Tk - suffix / vk;
        const float y = fk + f.density_shift;
        const float dsig_df = relu ? (fk > 0.0f ? 1.0f : 0.0f) : (y > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-y)));
        gf = dalpha * (z14[u] - z04[u]) * f.distance_scale * (1.0f - alpha) * dsig_df;
      }
    }
    run += chunk_total;
    if (k < S) {
      s_alpha[k] = gf;                     // alpha of sample k is not needed any more
      feat[(size_t)ray * S + k] = gf;      // consumed by the binned scatter kernels
    }
    }
  }
  // density scatter (tile ids for the binned scatter kernel) + position gradient
  float vmax = 0.0f;
  for (int k = lane; k < S; k += 64) {
    const float gf = s_alpha[k];                               // (0 for the last sample)
    const size_t ei = (size_t)ray * S + k;                     // entry index = sample id
    if (gf == 0.0f) {
      tid[ei] = 0xffff; tid[(size_t)nmax + ei] = 0xffff; tid[2 * (size_t)nmax + ei] = 0xffff;
      continue;
    }
    const float zk = z[k];
    float xr[3] = {o[0] + dh[0] * zk, o[1] + dh[1] * zk, o[2] + dh[2] * zk};
    float xc[3] = {xr[0], xr[1], xr[2]};
    contract3(xc[0], xc[1], xc[2]);
    float u[3], gu[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int a = 0; a < 3; ++a) u[a] = (xc[a] - f.lo[a]) * f.inv[a] - 1.0f;
#pragma unroll 1
    for (int p = 0; p < 3; ++p) {          // rolled: one plane's taps live at a time (238 -> 137 VGPRs)
      int x0, x1, y0, y1, l0, l1; float tx, ty, tl, gx, gy, gl;
      tap1d_g(u[MAT0[p]], f.pw[p], x0, x1, tx, gx);
      tap1d_g(u[MAT1[p]], f.ph[p], y0, y1, ty, gy);
      tap1d_g(u[VEC[p]],  f.ll[p], l0, l1, tl, gl);
      {
        const int t = (y0 / BTILE) * bg.tx[p] + x0 / BTILE;
        tid[(size_t)p * nmax + ei] = (uint16_t)t;
        atomicAdd(&s_h[bg.base[p] + t], 1);
      }
      // 32-bit byte offsets, two aligned float4 per tap (as density_feature32 in the forward)
      const unsigned row0 = (unsigned)y0 * (unsigned)f.pw[p], row1 = (unsigned)y1 * (unsigned)f.pw[p];
      const unsigned o00 = (row0 + x0) * (LRF_CD * 4u), o10 = (row0 + x1) * (LRF_CD * 4u);
      const unsigned o01 = (row1 + x0) * (LRF_CD * 4u), o11 = (row1 + x1) * (LRF_CD * 4u);
      const unsigned q0 = (unsigned)l0 * (LRF_CD * 4u), q1 = (unsigned)l1 * (LRF_CD * 4u);
      const float* pl = f.dplane[p];
      const float* ln = f.dline[p];
      float gix = 0.0f, giy = 0.0f, gil = 0.0f;
#pragma unroll
      for (int h = 0; h < LRF_CD / 4; ++h) {
        const float4 a4 = ld4b(pl, o00 + 16 * h), b4 = ld4b(pl, o10 + 16 * h);
        const float4 c4 = ld4b(pl, o01 + 16 * h), d4 = ld4b(pl, o11 + 16 * h);
        const float4 e4 = ld4b(ln, q0 + 16 * h), f4 = ld4b(ln, q1 + 16 * h);
        const float va[4] = {a4.x, a4.y, a4.z, a4.w}, vb[4] = {b4.x, b4.y, b4.z, b4.w};
        const float vc[4] = {c4.x, c4.y, c4.z, c4.w}, vd[4] = {d4.x, d4.y, d4.z, d4.w};
        const float ve[4] = {e4.x, e4.y, e4.z, e4.w}, vf[4] = {f4.x, f4.y, f4.z, f4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float v00 = va[c], v10 = vb[c], v01 = vc[c], v11 = vd[c];
          const float e0 = ve[c], e1 = vf[c];
          const float P = (v00 * (1.0f - tx) + v10 * tx) * (1.0f - ty) + (v01 * (1.0f - tx) + v11 * tx) * ty;
          const float Lv = e0 * (1.0f - tl) + e1 * tl;
          const float dP = gf * Lv, dL = gf * P;
          vmax = fmaxf(vmax, fmaxf(fabsf(dP), fabsf(dL)));
          gix += dP * ((v10 - v00) * (1.0f - ty) + (v11 - v01) * ty);
          giy += dP * ((v01 - v00) * (1.0f - tx) + (v11 - v10) * tx);
          gil += dL * (e1 - e0);
        }
      }
      gu[MAT0[p]] += gix * gx; gu[MAT1[p]] += giy * gy; gu[VEC[p]] += gil * gl;
    }
    float gx3[3] = {gu[0] * f.inv[0], gu[1] * f.inv[1], gu[2] * f.inv[2]};
    contract3_bwd(xr, gx3);
#pragma unroll
    for (int a = 0; a < 3; ++a) { go3[a] += gx3[a]; gdh[a] += gx3[a] * zk; }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, d, 64));
  // one atomic per wave on ONE word is 4096 memory-side atomics in a row (~12 ns each: 32 us of this kernel when first
  // measured): the maximum only grows, so a wave whose value does not exceed what the word already holds has nothing to add
  if (lane == 0 && __float_as_uint(vmax) > __hip_atomic_load(vmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(vmax_bits, __float_as_uint(vmax));
  // appearance partials of this ray's tiles (written by k_train_dgrad3); with rpart == null they are added
  // afterwards by k_rays_add_rpart, so that this kernel does not have to wait for the data-gradient kernel
  const int nt = rpart ? 2 * ((nsh + ITEM3 - 1) / ITEM3) : 0;          // 16-row tiles of the ray (k_shade3<SAVE>)
  for (int t = lane; t < nt; t += 64) {
    const float* rpp = rpart + ((size_t)ray * pmax + t) * 8;
#pragma unroll
    for (int a = 0; a < 3; ++a) { go3[a] += rpp[a]; gdh[a] += rpp[3 + a]; }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) { go3[a] = wave_sum(go3[a]); gdh[a] = wave_sum(gdh[a]); }
  if (lane == 0) {
    // dhat = d / n (tensorBase.py:578-580); depth = sum(w z) / n (:615)
    const float dot = dh[0] * gdh[0] + dh[1] * gdh[1] + dh[2] * gdh[2];
    const float depth = dsum / dn;
    float* gp = g_rays + (size_t)oray * 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      gp[a] = go3[a];
      gp[3 + a] = (gdh[a] - dh[a] * dot) / dn - gd * depth / dn * dh[a];
    }
  }
  flush_hist();
}

// d(loss)/d(rays) += the appearance lookups' position gradients (per-tile partials of k_train_dgrad3): the tail of
// k_bwd_ray as its own launch, linear in the partial sums: g_o += sum go, g_d += (I - dhat dhat^T) sum gdh / |d|.
__global__ __launch_bounds__(256) void k_rays_add_rpart(const float* __restrict__ rays, int R, const int* __restrict__ ncomp,
                                                        const float* __restrict__ rpart, int pmax, float* __restrict__ g_rays,
                                                        const int* __restrict__ perm) {
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= R) return;
  const int nt = 2 * ((ncomp[ray] + ITEM3 - 1) / ITEM3);    // 16-row tiles of the ray (k_shade3<SAVE>)
  float go3[3] = {0.0f, 0.0f, 0.0f}, gdh[3] = {0.0f, 0.0f, 0.0f};
  for (int t = 0; t < nt; ++t) {
    const float4* rpp = reinterpret_cast<const float4*>(rpart + ((size_t)ray * pmax + t) * 8);
    const float4 a = rpp[0], b = rpp[1];
    go3[0] += a.x; go3[1] += a.y; go3[2] += a.z;
    gdh[0] += a.w; gdh[1] += b.x; gdh[2] += b.y;
  }
  const float* rp = rays + (size_t)ray * 6;
  const float dn = sqrtf(rp[3] * rp[3] + rp[4] * rp[4] + rp[5] * rp[5]);
  const float dh[3] = {rp[3] / dn, rp[4] / dn, rp[5] / dn};
  const float dot = dh[0] * gdh[0] + dh[1] * gdh[1] + dh[2] * gdh[2];
  float* gp = g_rays + (size_t)(perm ? perm[ray] : ray) * 6;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    gp[a] += go3[a];
    gp[3 + a] += (gdh[a] - dh[a] * dot) / dn;
  }
}

// ---------------------------------------------------------------- binned gradient scatter
// Plane/line gradients are sums over samples that land on the same texels.  Doing them with
// global fp32 atomics measured 40 ms per 4096x512 batch (12-20 G lane-atomics/s: every ray
// starts near the field centre, so the same texels are hit by thousands of rays at once) --
// no better than the stock PyTorch-ROCm grid_sample backward.  Instead samples are binned by
// 32x32-texel plane tile (counting sort, LDS histograms), each tile's contributions are
// accumulated in LDS (ds_add_f32) by the workgroups that own it and flushed once; lines are
// accumulated per workgroup in LDS as well.  Global atomics remain only in the flushes.
// entry i -> sample id (ray*S + k) or ~0u.  Density: every sample with a non-zero feature
// gradient; appearance: the saved rows of the shaded samples.
template <bool APP>
__device__ __forceinline__ uint32_t entry_cid(uint32_t i, const float* __restrict__ gf, const uint32_t* __restrict__ rowinfo) {
  if (APP) return rowinfo[i];
  return gf[i] != 0.0f ? i : 0xffffffffu;
}
template <bool APP>
__device__ __forceinline__ uint32_t entry_count(int R, int S, const int* __restrict__ toff) {
  return APP ? (uint32_t)toff[R] * 16u : (uint32_t)R * (uint32_t)S;
}
__device__ __forceinline__ void cid_point(const DField& f, const float* __restrict__ rays, const float* __restrict__ z,
                                          int S, uint32_t cid, float u[3]) {
  const int ray = cid / S, k = cid % S;
  const float* rp = rays + (size_t)ray * 6;
  const float o[3] = {rp[0], rp[1], rp[2]};
  const float dn = sqrtf(rp[3] * rp[3] + rp[4] * rp[4] + rp[5] * rp[5]);
  const float dh[3] = {rp[3] / dn, rp[4] / dn, rp[5] / dn};
  float x[3];
  sample_point(f, o, dh, z[k], x, u);
}

// The same through what k_march left per ray (f.rdir: d / |d| as k_march divided it -- the very floats -- and |d|) and a
// multiply-high in place of the division by S: the position costs ~50 instructions instead of ~120 (a square root, three
// IEEE divisions and a 32-bit division) in kernels that are bound by their instruction count (k_scatter_fix).
// inv_s = floor(2^32 / S): umulhi(cid, inv_s) is floor(cid / S) or one less for every 32-bit cid.
__device__ __forceinline__ void cid_point_r(const DField& f, const float* __restrict__ rays, const float* __restrict__ z,
                                            int S, uint32_t inv_s, uint32_t cid, float u[3]) {
  uint32_t ray = __umulhi(cid, inv_s), k = cid - ray * (uint32_t)S;
  if (k >= (uint32_t)S) { k -= (uint32_t)S; ++ray; }
  const float* rp = rays + (size_t)ray * 6;
  const float4 dq = *reinterpret_cast<const float4*>(f.rdir + (size_t)ray * 4);
  const float o[3] = {rp[0], rp[1], rp[2]}, dh[3] = {dq.x, dq.y, dq.z};
  float x[3];
  sample_point(f, o, dh, z[k], x, u);
}

// entry index of position c on the cost axis of bins [bin_lo, bin_hi) (offs: entry offsets, coffs = offs + BIN_MAX + 1: cost
// offsets; the first SCATTER_VISIT_COST units of a non-empty bin are its visit, the rest its entries)
__device__ __forceinline__ int share_entry(const int* __restrict__ offs, int bin_lo, int bin_hi, long long c) {
  const int* coffs = offs + BIN_MAX + 1;
  if (c >= coffs[bin_hi]) return offs[bin_hi];
  int lo = bin_lo, hi = bin_hi;                        // largest bin with coffs[bin] <= c
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (coffs[mid] <= c) lo = mid; else hi = mid; }
  const int cnt = offs[lo + 1] - offs[lo];
  const long long within = c - coffs[lo] - (cnt ? SCATTER_VISIT_COST : 0);
  return offs[lo] + (int)max(0ll, min((long long)cnt, within));
}

// pass 1: tile id of every entry in every plane + global histogram
// (The histogram pass of the counting sort is not a kernel: k_bwd_ray (density) and k_train_app3 (appearance) hold every
// entry's taps anyway, write its three tile ids and count them in LDS histograms.)
// Second pass of the counting sort.  Every block scans the (<= 2048-entry) histogram itself -- there is no scan kernel --
// and reserves its entries' places with one atomic per tile on `cursor` (zero when the pass starts: cleared with the
// histogram); block 0 leaves the tile offsets in `offs` for the scatter kernel.
__global__ __launch_bounds__(256) void k_bin_fill(BinGeom bg, uint32_t nmax, int R, int S, const int* __restrict__ toff, int app,
                                                  const uint16_t* __restrict__ tid, const int* __restrict__ hist, int* __restrict__ cursor,
                                                  int* __restrict__ offs, uint32_t* __restrict__ list) {
  __shared__ int s_h[BIN_MAX];
  __shared__ int s_off[BIN_MAX];
  __shared__ int s_wsum[4];
  __shared__ int s_wnz[4];
  const uint32_t n = app ? (uint32_t)toff[R] * 16u : (uint32_t)R * (uint32_t)S;
  const uint32_t b0 = blockIdx.x * (uint32_t)BIN_CHUNK;
  if (b0 >= n && blockIdx.x != 0) return;                      // (block 0 always writes the offsets)
  {                                                            // exclusive scan of hist[0 .. total): 8 consecutive tiles per thread
    constexpr int PT = BIN_MAX / 256;
    const int t0 = threadIdx.x * PT, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int v[PT], sum = 0, nz = 0;
#pragma unroll
    for (int i = 0; i < PT; ++i) { v[i] = t0 + i < bg.total ? hist[t0 + i] : 0; sum += v[i]; nz += v[i] ? 1 : 0; }
    int incl = sum, incz = nz;                                   // (the second scan: non-empty bins in front -- the cost offsets of block 0)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64), tz = __shfl_up(incz, d, 64);
      if (lane >= d) { incl += t; incz += tz; }
    }
    if (lane == 63) { s_wsum[wave] = incl; s_wnz[wave] = incz; }
    __syncthreads();
    int base = incl - sum, basez = incz - nz;
    for (int q = 0; q < wave; ++q) { base += s_wsum[q]; basez += s_wnz[q]; }
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      if (t0 + i < BIN_MAX) s_off[t0 + i] = base;
      if (blockIdx.x == 0 && t0 + i < bg.total) { offs[t0 + i] = base; offs[BIN_MAX + 1 + t0 + i] = base + basez * SCATTER_VISIT_COST; }
      base += v[i]; basez += v[i] ? 1 : 0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 255) { offs[bg.total] = base; offs[BIN_MAX + 1 + bg.total] = base + basez * SCATTER_VISIT_COST; }   // = the number of entries / the total cost
  }
  if (b0 >= n) return;
  for (int i = threadIdx.x; i < bg.total; i += 256) s_h[i] = 0;
  __syncthreads();
  constexpr int PER = BIN_CHUNK / 256;
  // the 48 tile ids of this thread first, all loads in flight together: read inside the loop below -- behind a branch and in
  // front of an LDS atomic each -- they were 48 round trips one after the other, ~25 us of latency for a few us of work
  unsigned short tv[PER][3];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const uint32_t i = b0 + threadIdx.x + 256u * q;
#pragma unroll
    for (int p = 0; p < 3; ++p) tv[q][p] = i < n ? tid[(size_t)p * nmax + i] : (unsigned short)0xffff;
  }
  // rank of every entry among its block's entries of the same tile.  A wave's 64 entries are consecutive samples (or rows) of
  // a ray and mostly share their tile: one returning LDS atomic per lane put up to 64 lanes on one address -- serialised,
  // ~100 us of the density pass on a field without empty space.  The lanes of a RUN of equal ids take their ranks from ONE
  // atomic of the run's first lane (its length), the others add their position in the run.
  int rank[PER][3];
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < PER; ++q)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int key = tv[q][p];
      const int prev = __shfl_up(key, 1, 64);
      const unsigned long long starts = __ballot(lane == 0 || prev != key);
      const int leader = 63 - __builtin_clzll(starts & (~0ull >> (63 - lane)));
      const unsigned long long above = lane == 63 ? 0ull : (starts >> (lane + 1));
      const int len = (above ? lane + 1 + __builtin_ctzll(above) : 64) - leader;      // (read by the run's first lane only)
      int base = 0;
      if (lane == leader && key != 0xffff) base = atomicAdd(&s_h[bg.base[p] + key], len);
      base = __shfl(base, leader, 64);
      rank[q][p] = key != 0xffff ? base + (lane - leader) : -1;
    }
  __syncthreads();
  for (int i = threadIdx.x; i < bg.total; i += 256)          // (all of a thread's atomics in flight together: measured, slower -- 16.7 against 13.6 us)
    if (s_h[i]) s_h[i] = s_off[i] + atomicAdd(&cursor[i], s_h[i]);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const uint32_t i = b0 + threadIdx.x + 256u * q;
#pragma unroll
    for (int p = 0; p < 3; ++p)
      if (rank[q][p] >= 0) list[s_h[bg.base[p] + tv[q][p]] + rank[q][p]] = i;
  }
}

// Histogram + cursors + max|contribution| word of both counting sorts, cleared by ONE launch on the caller's stream in front
// of the fork.  NOT hipMemsetAsync: the runtime's fill reads its pattern from a staging slot, and in the captured progressive
// loop the density histogram came back filled with a stale 16-byte pattern -- another dispatch's kernel arguments -- instead
// of zeros (in the first graph behind a lifecycle event, profiles/r18_memset_fault.md); the fill pass then indexed the entry
// list 4 GB out of bounds: "Memory access fault by GPU node".
// The same launch clears the caller's gradient buffer when LrfGrads names it (zero_base / zero_floats: one range that holds
// every gradient the backward adds into): blocks behind the bins' take 2048 float4 each.
constexpr int BIN_CLEAR_WORDS = 2 * BIN_MAX + 8;
constexpr int BIN_CLEAR_BLOCKS = (BIN_CLEAR_WORDS + 255) / 256, ZERO_F4_PER_BLOCK = 2048;
__global__ __launch_bounds__(256) void k_clear_bins(int* __restrict__ h0, int* __restrict__ h1, float4* __restrict__ zero_base, long long zero_f4) {
  if (blockIdx.x < BIN_CLEAR_BLOCKS) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < BIN_CLEAR_WORDS) { h0[i] = 0; h1[i] = 0; }
    return;
  }
  const long long b0 = (long long)(blockIdx.x - BIN_CLEAR_BLOCKS) * ZERO_F4_PER_BLOCK;
#pragma unroll
  for (int k = 0; k < ZERO_F4_PER_BLOCK / 256; ++k) {
    const long long i = b0 + k * 256 + threadIdx.x;
    if (i < zero_f4) zero_base[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
}

// Plane gradients.  The entry lists are sorted by tile; workgroup w owns the w-th equal share
// of the concatenated list (a few tiles hold 12 % of all samples each, so one workgroup per
// tile is badly unbalanced), accumulates tile by tile in LDS and flushes each tile once.
// LINES: the gradient of line p is accumulated by the same pass over plane p's entries, in a second LDS array behind the
// tile ([L_p][C], flushed when the workgroup's share leaves the plane): an entry's position, taps and dX row are then
// read once instead of once per kernel, and the separate line kernel (which re-derives all of that to gather the four
// plane taps) disappears.  Round 3 measured this fusion as a no-op in a step bound by 7.3 GB of row traffic; with 1.8 GB
// of rows gone (round 4) the 0.15 ms of kernel time it removes show.  Falls back to the two-kernel form (lrf_render_bwd)
// when tile + line accumulators exceed the CU's LDS (appearance at 640^3).
// The accumulated tiles / lines are added straight into the reference's gradient tensors ([C][H][W] planes, [C][L] lines:
// ScatterDst), x fastest; round 3 flushed into a channel-last gradient image that one more kernel unpacked (a 35-70 MB
// memset + read + transpose per step).
struct ScatterDst { float* plane[3]; float* line[3]; };
#ifdef LRF_SCATTER_PROF          // experiment build: s_memtime totals of wave 0 of every workgroup -> g_scat_prof[APP][block][12]
__device__ unsigned long long g_scat_prof[2][2048][12];
#define SP_TICK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); sp_tk[i] += now_ - sp_last; sp_last = now_; } while (0)
#define SP_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define SP_WAIT_LGKM() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define SP_TICK(i) do {} while (0)
#define SP_WAIT_VM() do {} while (0)
#define SP_WAIT_LGKM() do {} while (0)
#endif
template <int C, bool APP, int NT, bool LINES>
__global__ __launch_bounds__(NT) void k_scatter_plane(DField f, BinGeom bg, ScatterDst dst, const float* __restrict__ rays,
                                                       const float* __restrict__ z, int S, const int* __restrict__ offs,
                                                       const uint32_t* __restrict__ list, const float* __restrict__ gf,
                                                       const uint32_t* __restrict__ rowinfo, const float* __restrict__ grd,
                                                       int bin_lo, int bin_hi) {
  extern __shared__ float s_acc[];                     // [BCELL*BCELL][C], then (LINES) [L_p][C]
  float* s_lacc = s_acc + BCELL * BCELL * C;
  // this launch's share of the list: the entries of bins [bin_lo, bin_hi) -- all of them (0, bg.total), or one plane's
  // when lrf_render_bwd runs the pass per plane (LRF_FLAG_PLANE_EVENTS: plane p's gradient is final behind its launch)
  const long long C0 = offs[BIN_MAX + 1 + bin_lo], Cn = (long long)offs[BIN_MAX + 1 + bin_hi] - C0;      // equal COST per workgroup (SCATTER_VISIT_COST)
  int a = share_entry(offs, bin_lo, bin_hi, C0 + Cn * blockIdx.x / gridDim.x);
  const int b = share_entry(offs, bin_lo, bin_hi, C0 + Cn * (blockIdx.x + 1) / gridDim.x);
  if (a >= b) return;
#ifdef LRF_SCATTER_PROF
  unsigned long long sp_tk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, sp_last = __builtin_readcyclecounter();
  const unsigned long long sp_t0 = sp_last;
  sp_tk[9] = (unsigned long long)(b - a);
#endif
  int lo = bin_lo, hi = bin_hi;                        // largest bin with offs[bin] <= a
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (offs[mid] <= a) lo = mid; else hi = mid; }
  int bin = lo;
  int lplane = -1;                                     // plane whose line gradient s_lacc currently holds (LINES)
  auto flush_line = [&]() {                            // whole workgroup; s_lacc -> += the gradient of line `lplane` ([C][L])
    __syncthreads();
    float* gln = dst.line[lplane];
    const int ll = f.ll[lplane], nl = ll * C;
    for (int i = threadIdx.x; i < nl; i += NT) {
      const int c = i / ll, l = i % ll;
      const float v = s_lacc[l * C + c];
      if (v != 0.0f) atomic_add_f32(gln + i, v);
    }
    __syncthreads();
  };
  while (a < b) {
    while (offs[bin + 1] <= a) ++bin;
    const int seg_end = min(b, offs[bin + 1]);
    const int p = bin >= bg.base[2] ? 2 : (bin >= bg.base[1] ? 1 : 0);
    const int t = bin - bg.base[p];
    const int tx0 = (t % bg.tx[p]) * BTILE, ty0 = (t / bg.tx[p]) * BTILE;
    if (LINES && p != lplane) {
      if (lplane >= 0) flush_line();
      lplane = p;
      for (int i = threadIdx.x; i < f.ll[p] * C; i += NT) s_lacc[i] = 0.0f;
    }
    for (int i = threadIdx.x; i < BCELL * BCELL * C; i += NT) s_acc[i] = 0.0f;
    __syncthreads();
    SP_TICK(0);
#ifdef LRF_SCATTER_PROF
    sp_tk[8] += 1;
#endif
    const float* lnp = APP ? f.aline[p] : f.dline[p];
    const float* plp = APP ? f.aplane[p] : f.dplane[p];
    // Two phases per 64 entries of a wave.  A: lane = entry -- list / row lookup, sample position,
    // taps (one dependent-load chain per 64 entries instead of one per 8).  B: the 8-lane groups
    // take the entries 8 at a time, fetch the packed taps of theirs with a cross-lane read, and each
    // lane adds its channels.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int e0 = a + wv * 64; e0 < seg_end; e0 += NT) {
      const int e = e0 + lane;
      int i_row = 0, c00 = 0, lpk = 0, ppk = 0;
      float tx = 0.0f, ty = 0.0f, tl = 0.0f;
      if (e < seg_end) {
        const uint32_t i = list[e];
        const uint32_t cid = APP ? rowinfo[i] : i;
        float u[3];
        cid_point(f, rays, z, S, cid, u);
        int x0, x1, y0, y1, l0, l1;
        tap1d(u[MAT0[p]], f.pw[p], x0, x1, tx);
        tap1d(u[MAT1[p]], f.ph[p], y0, y1, ty);
        tap1d(u[VEC[p]],  f.ll[p], l0, l1, tl);
        i_row = (int)i;
        c00 = (((y0 - ty0) * BCELL + (x0 - tx0)) << 2) | ((y1 - y0) << 1) | (x1 - x0);   // +1 taps are clamped: step 0 or 1
        lpk = (l0 << 1) | (l1 - l0);
        ppk = y0 * f.pw[p] + x0;                       // (LINES) texel index of the base tap in the plane itself
      }
      SP_WAIT_VM(); SP_TICK(1);
#ifdef LRF_SCATTER_PROF
      sp_tk[10] += 1;
#endif
      const int n_here = min(64, seg_end - e0);
      // A group of LPE lanes takes LPE consecutive entries IN LIST ORDER: the list keeps consecutive
      // samples of a ray adjacent, and those mostly share their base texel, so their contributions
      // are summed in registers and reach LDS once per run (same-address CAS adds from neighbouring
      // lanes would serialise instead).  Appearance: 4 lanes per entry, lane owns the aligned group of
      // 6 channels 6s..6s+5 (two float4 of the padded texel per tap -- the texture path retires one
      // wave-level load per ~16 cycles whatever its width, so wide loads are what counts);
      // density: 4 lanes per entry, two channels each.  Pairs of channels are added with ONE 64-bit compare-and-swap
      // (lds_add4_f2): appearance 320 -> 290 us, density (8 lanes x 1 channel before) 259 -> 224 us.
      // (density taps fetched as float4 by the group and redistributed through 192 bytes of LDS per group -- 2 gather
      // instructions + 8 LDS operations instead of 7 dword gathers per entry -- was measured too: 258 -> 306 us; the kernel is
      // bound by its LDS adds, not by the texture path.)
      // (density with 2 lanes per entry and float4 taps -- 3.5 x fewer gather instructions -- was measured in round 4: 244 ->
      // 640-800 us.  Thirty-two lane pairs then work on 64 CONSECUTIVE samples of a ray at once, which share their cells, and
      // their same-address CAS adds serialise; with 8 lanes per entry a group merges 8 consecutive samples in registers.)
      constexpr int LPE = APP ? 4 : LRF_DENS_LPE, CPL = C / LPE;
      const int sub = lane % LPE, grp = lane / LPE;
      int cur = -1, curl = -1;
      float acc[4][CPL], lac[2][CPL];
#pragma unroll
      for (int j = 0; j < CPL; ++j) { acc[0][j] = 0.0f; acc[1][j] = 0.0f; acc[2][j] = 0.0f; acc[3][j] = 0.0f; lac[0][j] = 0.0f; lac[1][j] = 0.0f; }
      auto flush = [&](int cp) {
        const int b00 = (cp >> 2) * C + CPL * sub, b10 = b00 + (cp & 1) * C, b01 = b00 + ((cp >> 1) & 1) * BCELL * C, b11 = b01 + (cp & 1) * C;
        if constexpr (LRF_SCATTER_CAS64 && CPL % 2 == 0) {
#pragma unroll
          for (int j = 0; j < CPL; j += 2) {
            lds_add4_f2(&s_acc[b00 + j], acc[0][j], acc[0][j + 1], &s_acc[b10 + j], acc[1][j], acc[1][j + 1],
                        &s_acc[b01 + j], acc[2][j], acc[2][j + 1], &s_acc[b11 + j], acc[3][j], acc[3][j + 1]);
            acc[0][j] = 0.0f; acc[1][j] = 0.0f; acc[2][j] = 0.0f; acc[3][j] = 0.0f;
            acc[0][j + 1] = 0.0f; acc[1][j + 1] = 0.0f; acc[2][j + 1] = 0.0f; acc[3][j + 1] = 0.0f;
          }
        } else {
#pragma unroll
          for (int j = 0; j < CPL; ++j) {
            lds_add4_f32(&s_acc[b00 + j], acc[0][j], &s_acc[b10 + j], acc[1][j],
                         &s_acc[b01 + j], acc[2][j], &s_acc[b11 + j], acc[3][j]);
            acc[0][j] = 0.0f; acc[1][j] = 0.0f; acc[2][j] = 0.0f; acc[3][j] = 0.0f;
          }
        }
      };
      auto flushl = [&](int lp) {                      // two channels' pairs of cells per call: four CAS round trips overlapped
        const int c0 = (lp >> 1) * C + CPL * sub, c1 = c0 + (lp & 1) * C;
        if constexpr (LRF_SCATTER_CAS64 && CPL % 2 == 0) {
#pragma unroll
          for (int j = 0; j < CPL; j += 2) {
            lds_add2_f2(&s_lacc[c0 + j], lac[0][j], lac[0][j + 1], &s_lacc[c1 + j], lac[1][j], lac[1][j + 1]);
            lac[0][j] = 0.0f; lac[1][j] = 0.0f; lac[0][j + 1] = 0.0f; lac[1][j + 1] = 0.0f;
          }
          return;
        }
#pragma unroll
        for (int j = 0; j + 1 < CPL; j += 2) {
          lds_add4_f32(&s_lacc[c0 + j], lac[0][j], &s_lacc[c1 + j], lac[1][j], &s_lacc[c0 + j + 1], lac[0][j + 1], &s_lacc[c1 + j + 1], lac[1][j + 1]);
          lac[0][j] = 0.0f; lac[1][j] = 0.0f; lac[0][j + 1] = 0.0f; lac[1][j + 1] = 0.0f;
        }
        if (CPL & 1) {
          lds_add_f32(&s_lacc[c0 + CPL - 1], lac[0][CPL - 1]); lds_add_f32(&s_lacc[c1 + CPL - 1], lac[1][CPL - 1]);
          lac[0][CPL - 1] = 0.0f; lac[1][CPL - 1] = 0.0f;
        }
      };
#pragma unroll 1
      for (int q = 0; q < LPE; ++q) {
        const int src = LPE * grp + q;
        const int ir = __shfl(i_row, src, 64), cp = __shfl(c00, src, 64), lp = __shfl(lpk, src, 64);
        const int pp = LINES ? __shfl(ppk, src, 64) : 0;
        const float sx = __shfl(tx, src, 64), sy = __shfl(ty, src, 64), sl = __shfl(tl, src, 64);
        if (src >= n_here) continue;
        if (cp != cur) {
          if (cur >= 0) flush(cur);
          cur = cp;
        }
        if (LINES && lp != curl) {
          if (curl >= 0) flushl(curl);
          curl = lp;
        }
        SP_WAIT_LGKM(); SP_TICK(4);
        const float w00 = (1.0f - sx) * (1.0f - sy), w10 = sx * (1.0f - sy), w01 = (1.0f - sx) * sy, w11 = sx * sy;
        constexpr int CS = APP ? LRF_CAS : C;               // channel stride of the cache / gradient image
        const float* r0 = lnp + (size_t)(lp >> 1) * CS;
        const float* r1 = r0 + (size_t)(lp & 1) * CS;
        float e0v[8], e1v[8], dv[6];
        float v00[8], v10[8], v01[8], v11[8];
        if (APP) {
          ld4g(r0 + 8 * sub, e0v); ld4g(r0 + 8 * sub + 4, e0v + 4);
          ld4g(r1 + 8 * sub, e1v); ld4g(r1 + 8 * sub + 4, e1v + 4);
          load_dx6(grd, (size_t)ir, p, sub, dv);
        } else {
#pragma unroll
          for (int j = 0; j < CPL; ++j) { e0v[j] = r0[CPL * sub + j]; e1v[j] = r1[CPL * sub + j]; }
          dv[0] = gf[ir];
#pragma unroll
          for (int j = 1; j < CPL; ++j) dv[j] = dv[0];
        }
        if (LINES) {                                   // the plane's own four taps: what the line gradient multiplies dX with
          const float* q00 = plp + (size_t)pp * CS;
          const float* q10 = q00 + (size_t)(cp & 1) * CS;
          const float* q01 = q00 + (size_t)((cp >> 1) & 1) * f.pw[p] * CS;
          const float* q11 = q01 + (size_t)(cp & 1) * CS;
          if (APP) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              ld4g(q00 + 8 * sub + 4 * h, v00 + 4 * h); ld4g(q10 + 8 * sub + 4 * h, v10 + 4 * h);
              ld4g(q01 + 8 * sub + 4 * h, v01 + 4 * h); ld4g(q11 + 8 * sub + 4 * h, v11 + 4 * h);
            }
          } else {
#pragma unroll
            for (int j = 0; j < CPL; ++j) { v00[j] = q00[CPL * sub + j]; v10[j] = q10[CPL * sub + j]; v01[j] = q01[CPL * sub + j]; v11[j] = q11[CPL * sub + j]; }
          }
        }
        SP_WAIT_VM(); SP_TICK(2);
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          const float Lv = e0v[j] * (1.0f - sl) + e1v[j] * sl;
          const float dP = dv[j] * Lv;
          acc[0][j] += dP * w00; acc[1][j] += dP * w10; acc[2][j] += dP * w01; acc[3][j] += dP * w11;
          if (LINES) {
            const float P = v00[j] * w00 + v10[j] * w10 + v01[j] * w01 + v11[j] * w11;
            const float dL = dv[j] * P;
            lac[0][j] += dL * (1.0f - sl);
            lac[1][j] += dL * sl;
          }
        }
        SP_TICK(3);
      }
      if (cur >= 0) flush(cur);
      if (LINES && curl >= 0) flushl(curl);
      SP_WAIT_LGKM(); SP_TICK(4);
    }
    __syncthreads();
    SP_TICK(5);
    float* gpl = dst.plane[p];
    for (int i = threadIdx.x; i < BCELL * BCELL * C; i += NT) {       // x fastest: 33 consecutive floats of one channel row
      const int c = i / (BCELL * BCELL), cell = i % (BCELL * BCELL);
      const float v = s_acc[cell * C + c];
      if (v == 0.0f) continue;
      const int x = tx0 + cell % BCELL, y = ty0 + cell / BCELL;
      if (x < f.pw[p] && y < f.ph[p])
        atomic_add_f32(gpl + ((size_t)c * f.ph[p] + y) * f.pw[p] + x, v);
    }
    __syncthreads();
    SP_TICK(6);
    a = seg_end;
  }
  if (LINES && lplane >= 0) flush_line();
  SP_TICK(7);
#ifdef LRF_SCATTER_PROF
  if (threadIdx.x == 0 && blockIdx.x < 2048) {
    sp_tk[11] = __builtin_readcyclecounter() - sp_t0;
    for (int i = 0; i < 12; ++i) g_scat_prof[APP ? 1 : 0][blockIdx.x][i] = sp_tk[i];
  }
#endif
}

// line gradients: LINE_WGS workgroups per line, each accumulates its slice of the entries
template <int C, bool APP, int NT>
__global__ __launch_bounds__(NT) void k_scatter_line(DField f, ScatterDst dst, const float* __restrict__ rays, const float* __restrict__ z,
                                                      int R, int S, const int* __restrict__ toff, const float* __restrict__ gf,
                                                      const uint32_t* __restrict__ rowinfo, const float* __restrict__ grd) {
  extern __shared__ float s_acc[];                     // [L_p][C]
  const int p = blockIdx.x / LINE_WGS, wg = blockIdx.x % LINE_WGS;
  const int nl = f.ll[p] * C;
  for (int i = threadIdx.x; i < nl; i += NT) s_acc[i] = 0.0f;
  __syncthreads();
  const uint32_t n = entry_count<APP>(R, S, toff);
  const uint32_t a = (uint32_t)((unsigned long long)n * wg / LINE_WGS), b = (uint32_t)((unsigned long long)n * (wg + 1) / LINE_WGS);
  const float* plp = APP ? f.aplane[p] : f.dplane[p];
  // Same two phases as k_scatter_plane.  A: lane = entry (row lookup, sample position, taps) for 64
  // consecutive entries.  B: 8-lane group g (lane `sub` owns channels sub, sub+8, ...) takes entries
  // 8g..8g+7 in order -- consecutive samples of a ray -- and keeps the sums for the current line cell
  // pair in registers, touching LDS only when the cell changes.
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr int NW = NT / 64;
  const uint32_t len = b - a;
  const uint32_t wa = a + (uint32_t)((unsigned long long)len * wv / NW), wb = a + (uint32_t)((unsigned long long)len * (wv + 1) / NW);
  for (uint32_t i0 = wa; i0 < wb; i0 += 64) {
    const uint32_t i = i0 + lane;
    int ppk = 0, lpk = -1;
    float tx = 0.0f, ty = 0.0f, tl = 0.0f;
    if (i < wb) {
      const uint32_t cid = entry_cid<APP>(i, gf, rowinfo);
      if (cid != 0xffffffffu) {
        float u[3];
        cid_point(f, rays, z, S, cid, u);
        int x0, x1, y0, y1, l0, l1;
        tap1d(u[MAT0[p]], f.pw[p], x0, x1, tx);
        tap1d(u[MAT1[p]], f.ph[p], y0, y1, ty);
        tap1d(u[VEC[p]],  f.ll[p], l0, l1, tl);
        ppk = ((y0 * f.pw[p] + x0) << 2) | ((y1 - y0) << 1) | (x1 - x0);
        lpk = (l0 << 1) | (l1 - l0);
      }
    }
    const int n_here = (int)min(64u, wb - i0);
    constexpr int LPE = APP ? 4 : 8, CPL = C / LPE;     // lanes per entry / channels per lane, see k_scatter_plane
    const int sub = lane % LPE, grp = lane / LPE;
    int cur = -1;
    float acc0[CPL], acc1[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) { acc0[j] = 0.0f; acc1[j] = 0.0f; }
    auto flush = [&](int lp) {
      const int c0 = (lp >> 1) * C + CPL * sub, c1 = c0 + (lp & 1) * C;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        lds_add_f32(&s_acc[c0 + j], acc0[j]);
        lds_add_f32(&s_acc[c1 + j], acc1[j]);
        acc0[j] = 0.0f; acc1[j] = 0.0f;
      }
    };
#pragma unroll 1
    for (int q = 0; q < LPE; ++q) {
      const int src = LPE * grp + q;
      const int pp = __shfl(ppk, src, 64), lp = __shfl(lpk, src, 64);
      const float sx = __shfl(tx, src, 64), sy = __shfl(ty, src, 64), sl = __shfl(tl, src, 64);
      if (src >= n_here || lp < 0) continue;
      if (lp != cur) {
        if (cur >= 0) flush(cur);
        cur = lp;
      }
      constexpr int CS = APP ? LRF_CAS : C;
      const float* q00 = plp + (size_t)(pp >> 2) * CS;
      const float* q10 = q00 + (size_t)(pp & 1) * CS;
      const float* q01 = q00 + (size_t)((pp >> 1) & 1) * f.pw[p] * CS;
      const float* q11 = q01 + (size_t)(pp & 1) * CS;
      const float w00 = (1.0f - sx) * (1.0f - sy), w10 = sx * (1.0f - sy), w01 = (1.0f - sx) * sy, w11 = sx * sy;
      const uint32_t ie = i0 + src;
      float v00[8], v10[8], v01[8], v11[8], dv[6];
      if (APP) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          ld4g(q00 + 8 * sub + 4 * h, v00 + 4 * h); ld4g(q10 + 8 * sub + 4 * h, v10 + 4 * h);
          ld4g(q01 + 8 * sub + 4 * h, v01 + 4 * h); ld4g(q11 + 8 * sub + 4 * h, v11 + 4 * h);
        }
        load_dx6(grd, (size_t)ie, p, sub, dv);
      } else {
        v00[0] = q00[sub]; v10[0] = q10[sub]; v01[0] = q01[sub]; v11[0] = q11[sub]; dv[0] = gf[ie];
      }
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const float P = v00[j] * w00 + v10[j] * w10 + v01[j] * w01 + v11[j] * w11;
        const float dL = dv[j] * P;
        acc0[j] += dL * (1.0f - sl);
        acc1[j] += dL * sl;
      }
    }
    if (cur >= 0) flush(cur);
  }
  __syncthreads();
  float* gln = dst.line[p];
  const int ll = f.ll[p];
  for (int i = threadIdx.x; i < nl; i += NT) {
    const int c = i / ll, l = i % ll;
    const float v = s_acc[l * C + c];
    if (v != 0.0f) atomic_add_f32(gln + i, v);
  }
}

// ---------------------------------------------------------------- density scatter on 64-bit fixed point (round 6)
// What the phase profile of k_scatter_plane said (profiles/r17_scatter_phases.md): per 64 entries a wave spends 2.3 K cycles on
// positions and taps, 2.7 K waiting for gathers and 10.3 K in "shuffles + LDS adds" -- and replacing the compare-and-swap pairs
// by fire-and-forget integer adds (wrong sums, timing only) took just 14 % off the kernel: what costs is the machinery around
// the adds -- seven cross-lane reads per entry to hand phase A's lane-per-entry results to phase B's lane groups, run merging,
// per-group flush branches.  This kernel has none of it: lane = entry from the list read to the last add, all eight channels
// in the lane (taps as two float4 each: the widest loads the texture path offers), and every contribution is ONE
// ds_add_u64 of a fixed-point value -- no return value, no retry loop, same-cell adds of neighbouring lanes (consecutive
// samples of a ray) serialised by the LDS itself.  Accumulators are channel-major ([C][cell]: lanes of one instruction hit
// different banks unless they share the cell).
// Fixed point: every contribution is at most vmax = max |g line|, |g plane| (reduced by k_bwd_ray, which forms those products
// for the position gradient), a cell receives at most 4 n of them from the n entries of the segment, so with
// scale = 2^(min(46, 62 - bits(4 n)) - exponent(vmax)) no sum leaves 63 bits and no addend (the sum of a run of up to 16 lanes)
// leaves 51 (the conversion below is exact there).  The quantum is vmax 2^-46 .. 2^-38: sums are exact to far below fp32's own rounding of each product, and --
// integer adds being associative -- independent of the order the entries arrive in: a workgroup's tile and line sums are
// bit-reproducible (what still varies from run to run is the order of the fp32 atomics that add them into the gradient).
constexpr int FIX_NT = 1024;
__device__ __forceinline__ unsigned long long fix64(float v, double scale) {
  // v * scale + 1.5 * 2^52: the low mantissa bits of the sum hold rint(v * scale) in two's complement (|v * scale| < 2^51)
  const double x = fma((double)v, scale, 6755399441055744.0);
  return (unsigned long long)(__double_as_longlong(x) - 0x4338000000000000ll);
}
__device__ __forceinline__ int fix_shift(unsigned n_contrib, int vex) { return min(46, 62 - (32 - __builtin_clz(n_contrib))) - vex; }   // (46: one add carries the sum of up to 16 lanes)
// runs of equal keys among consecutive lanes of a 16-lane row: position of the lane in its run and whether it is the run's
// last lane (which then holds the inclusive sum of seg_sum).  Lanes that are not `valid` form runs of their own.
struct SegRun { float m1, m2, m4, m8; bool tail; };
template <int D>
__device__ __forceinline__ int row_shr_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x110 + D, 0xf, 0xf, true); }   // lane i <- lane i - D of its row, 0 where there is none
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
// Inclusive sums over the lanes of the run up to this one, four values at once: v += mask * (v of the lane 1, 2, 4, 8 below).
// The shifted value is MULTIPLIED by the lane's 0 / 1 mask, not selected: with `cond ? dpp(v) : 0` the compiler predicates
// the DPP move on cond, and a DPP read from a lane that EXEC disables returns 0 -- the run's first lane (cond false) then
// contributes nothing (scripts/ubench/seg_sum.hip checks both forms against a host loop).  Written as v_fmac_f32 with a DPP
// source: the compiler's form of fmaf(dpp(v), m, v) is v_mov_b32_dpp + v_fmac_f32 + hazard nops, 2.6 issue slots per value
// and step (500 of the kernel's 1000 per 64 entries); here it is one, and the steps of one value are four instructions
// apart, so no DPP read follows the VALU write of its register closer than the two wait states it needs (s_nop 1 covers
// whatever the compiler placed in front).
__device__ __forceinline__ void seg_sum4(float& v0, float& v1, float& v2, float& v3, const SegRun& r) {
#define LRF_STEP(N, M) \
  "v_fmac_f32_dpp %0, %0, " M " row_shr:" N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_fmac_f32_dpp %1, %1, " M " row_shr:" N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_fmac_f32_dpp %2, %2, " M " row_shr:" N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_fmac_f32_dpp %3, %3, " M " row_shr:" N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
  asm volatile("s_nop 1\n\t" LRF_STEP("1", "%4") LRF_STEP("2", "%5") LRF_STEP("4", "%6") LRF_STEP("8", "%7")
               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)
               : "v"(r.m1), "v"(r.m2), "v"(r.m4), "v"(r.m8));
#undef LRF_STEP
}
template <int C, bool APP, int NT, int CH = LRF_CD>
__global__ __launch_bounds__(NT) void k_scatter_fix(DField f, BinGeom bg, ScatterDst dst, const float* __restrict__ rays,
                                                     const float* __restrict__ z, int S, const int* __restrict__ offs,
                                                     const uint32_t* __restrict__ list, const float* __restrict__ gf,
                                                     const uint32_t* __restrict__ rowinfo, const float* __restrict__ grd,
                                                     const unsigned* __restrict__ vmax_bits, int bin_lo, int bin_hi) {
  // APP (C = 24): the appearance tensors, EIGHT channels per sweep over a tile's entries (three sweeps per tile, back to back:
  // a 24-channel tile of 64-bit cells would be 209 KB; the line accumulators hold all 24 channels).  The sample's
  // d(loss)/d(feature) is then per channel (the dX row k_train_app3 left), the taps come from the dense 24-channel caches
  // (aplane2 / aline2: 32-byte pieces at 32 * sweep), vmax is k_train_app3's max |dX line|, |dX plane|.  Positions and
  // runs are formed again in every sweep (~a quarter of a sweep's instructions).  (Sweeping the workgroup's whole SHARE
  // three times instead -- 8-channel line accumulators -- fetched every 128-byte line of taps and dX rows three times from
  // beyond L2: 1.57 GB of counter traffic for the two scatter kernels against 0.88 GB with the compare-and-swap kernels.)
  // CH: channels per sweep.  8, or 4 for the appearance tensors of grids whose lines do not fit beside an 8-channel tile (lines of
  // 480 .. 640 cells: six sweeps per tile, the tile 34.8 KB; the dX block's 8-channel groups are read in halves)
  static_assert(C % CH == 0 && (CH == 8 || (APP && CH == 4)) && (APP ? C == LRF_CA : C == LRF_CD), "8 (or 4) channels per sweep");
  constexpr int CELLS = BCELL * BCELL;
  extern __shared__ unsigned long long s_fx[];          // [CH][CELLS] tile (of the sweep), then [C][L_p] line
  unsigned long long* s_fl = s_fx + CH * CELLS;
  const long long C0 = offs[BIN_MAX + 1 + bin_lo], Cn = (long long)offs[BIN_MAX + 1 + bin_hi] - C0;      // equal COST per workgroup (SCATTER_VISIT_COST)
  int a = share_entry(offs, bin_lo, bin_hi, C0 + Cn * blockIdx.x / gridDim.x);
  const int b = share_entry(offs, bin_lo, bin_hi, C0 + Cn * (blockIdx.x + 1) / gridDim.x);
  if (a >= b) return;
#ifdef LRF_SCATTER_PROF
  unsigned long long sp_tk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, sp_last = __builtin_readcyclecounter();
  const unsigned long long sp_t0 = sp_last;
  sp_tk[9] = (unsigned long long)(b - a);
#endif
  const uint32_t inv_s = (uint32_t)(0x100000000ull / (unsigned long long)S);
  int vex = 0;
  (void)frexpf(__uint_as_float(*vmax_bits), &vex);      // vmax < 2^vex
  vex = max(-120, min(127, vex));
  int lo = bin_lo, hi = bin_hi;                         // largest bin with offs[bin] <= a
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (offs[mid] <= a) lo = mid; else hi = mid; }
  int bin = lo;
  int lplane = -1, shL = 0;                             // plane whose line gradient s_fl holds, its scale
  auto flush_line = [&]() {                             // whole workgroup; s_fl -> += the gradient of line `lplane` ([C][L])
    __syncthreads();
    float* gln = dst.line[lplane];
    const int nl = f.ll[lplane] * C;
    const double inv = ldexp(1.0, -shL);
    for (int i = threadIdx.x; i < nl; i += NT) {
      const long long q = (long long)s_fl[i];
      if (q != 0) atomic_add_f32(gln + i, (float)((double)q * inv));
    }
    __syncthreads();
  };
  while (a < b) {
    while (offs[bin + 1] <= a) ++bin;
    const int seg_end = min(b, offs[bin + 1]);
    const int p = bin >= bg.base[2] ? 2 : (bin >= bg.base[1] ? 1 : 0);
    const int t = bin - bg.base[p];
    const int tx0 = (t % bg.tx[p]) * BTILE, ty0 = (t / bg.tx[p]) * BTILE;
    const int pw = f.pw[p], ph = f.ph[p], ll = f.ll[p];
    if (p != lplane) {
      if (lplane >= 0) flush_line();
      lplane = p;
      // this workgroup's entries of plane p: at most two contributions each to a line cell
      const int pend = min(b, offs[min(bin_hi, p == 2 ? bg.total : bg.base[p + 1])]);
      shL = fix_shift(2u * (unsigned)(pend - a), vex);
      for (int i = threadIdx.x; i < ll * C; i += NT) s_fl[i] = 0ull;
    }
#ifdef LRF_SCATTER_PROF
    sp_tk[8] += 1;
#endif
    for (int sweep = 0; sweep < C / CH; ++sweep) {
    for (int i = threadIdx.x; i < CH * CELLS; i += NT) s_fx[i] = 0ull;
    __syncthreads();
    SP_TICK(0);
    const int shT = fix_shift(4u * (unsigned)(seg_end - a), vex);
    const double scT = ldexp(1.0, shT), scL = ldexp(1.0, shL);
    const float* lnp = APP ? f.aline2[p] + CH * sweep : f.dline[p];      // the sweep's 8 channels of a texel: 32 bytes, 16-byte aligned
    const float* plp = APP ? f.aplane2[p] + CH * sweep : f.dplane[p];
    const int am0 = MAT0[p], am1 = MAT1[p], av = VEC[p];
    const int lane = threadIdx.x & 63;
    for (int e0 = a + (int)(threadIdx.x & ~63u); e0 < seg_end; e0 += NT) {      // (whole waves: the run sums below are wave operations)
      const int e = e0 + lane;
      const bool valid = e < seg_end;
      const uint32_t row = list[valid ? e : seg_end - 1];
      const uint32_t cid = APP ? rowinfo[row] : row;
      float dx[CH];                                     // d(loss)/d(feature of channel c): one scalar for the density, the row's dX for the appearance
      if (APP) {
        if constexpr (CH == 8) {
          const float* dxp = grd_dx8(grd, row, p, sweep);
          ld4g(dxp, dx); ld4g(dxp + 4, dx + 4);
        } else {
          ld4g(grd_dx8(grd, row, p, sweep >> 1) + 4 * (sweep & 1), dx);
        }
      } else {
        const float g = gf[cid];
#pragma unroll
        for (int c = 0; c < CH; ++c) dx[c] = g;
      }
      if (!valid) {
#pragma unroll
        for (int c = 0; c < CH; ++c) dx[c] = 0.0f;
      }
      float u[3];
      cid_point_r(f, rays, z, S, inv_s, cid, u);
      int x0, x1, y0, y1, l0, l1; float tx, ty, tl;
      tap1d(u[am0], pw, x0, x1, tx);
      tap1d(u[am1], ph, y0, y1, ty);
      tap1d(u[av],  ll, l0, l1, tl);
      const int c00 = (y0 - ty0) * BCELL + (x0 - tx0), cx = x1 - x0, cy = (y1 - y0) * BCELL;
      const float w00 = (1.0f - tx) * (1.0f - ty), w10 = tx * (1.0f - ty), w01 = (1.0f - tx) * ty, w11 = tx * ty;
      float e0v[CH], e1v[CH], v00[CH], v10[CH], v01[CH], v11[CH];
      const float* r0 = lnp + (size_t)l0 * C;                 // (C = the texel's channel stride in its cache)
      const float* r1 = lnp + (size_t)l1 * C;
      const float* q00 = plp + ((size_t)y0 * pw + x0) * C;
      const float* q10 = q00 + (size_t)cx * C;
      const float* q01 = q00 + (size_t)(y1 - y0) * pw * C;
      const float* q11 = q01 + (size_t)cx * C;
#pragma unroll
      for (int h = 0; h < CH / 4; ++h) {
        ld4g(r0 + 4 * h, e0v + 4 * h);  ld4g(r1 + 4 * h, e1v + 4 * h);
        ld4g(q00 + 4 * h, v00 + 4 * h); ld4g(q10 + 4 * h, v10 + 4 * h);
        ld4g(q01 + 4 * h, v01 + 4 * h); ld4g(q11 + 4 * h, v11 + 4 * h);
      }
      SP_WAIT_VM(); SP_TICK(1);
#ifdef LRF_SCATTER_PROF
      sp_tk[10] += 1;
#endif
      // Consecutive entries are mostly consecutive samples of one ray, and in the contracted far field dozens of them share
      // their cell: left alone, the adds of one instruction pile up on a handful of addresses and the LDS serialises them
      // (measured: 52 cycles per ds_add_u64 instruction against 11 for distinct addresses).  So the lanes of a RUN -- equal
      // cell and tap steps, inside one 16-lane row -- are summed across lanes first (inclusive segmented scan, four DPP
      // row shifts per value: seg_sum4) and only the run's last lane adds.
      const SegRun rt = seg_run((c00 << 2) | (cx << 1) | (cy ? 1 : 0), valid, lane);
      const SegRun rl = seg_run((l0 << 1) | (l1 - l0), valid, lane);
      unsigned long long* tc = s_fx + c00;
      unsigned long long* lc0 = s_fl + (size_t)sweep * CH * ll + l0;
      unsigned long long* lc1 = s_fl + (size_t)sweep * CH * ll + l1;
#pragma unroll
      for (int c = 0; c < CH; c += 2) {
        float t[2][4], sl[4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float Lv = e0v[c + j] * (1.0f - tl) + e1v[c + j] * tl;
          const float dP = dx[c + j] * Lv;
          t[j][0] = dP * w00; t[j][1] = dP * w10; t[j][2] = dP * w01; t[j][3] = dP * w11;
          seg_sum4(t[j][0], t[j][1], t[j][2], t[j][3], rt);
          const float P = v00[c + j] * w00 + v10[c + j] * w10 + v01[c + j] * w01 + v11[c + j] * w11;
          const float dL = dx[c + j] * P;
          sl[2 * j] = dL * (1.0f - tl); sl[2 * j + 1] = dL * tl;
        }
        seg_sum4(sl[0], sl[1], sl[2], sl[3], rl);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (rt.tail) {
            atomicAdd(tc + (c + j) * CELLS, fix64(t[j][0], scT));
            atomicAdd(tc + (c + j) * CELLS + cx, fix64(t[j][1], scT));
            atomicAdd(tc + (c + j) * CELLS + cy, fix64(t[j][2], scT));
            atomicAdd(tc + (c + j) * CELLS + cy + cx, fix64(t[j][3], scT));
          }
          if (rl.tail) {
            atomicAdd(lc0 + (c + j) * ll, fix64(sl[2 * j], scL));
            atomicAdd(lc1 + (c + j) * ll, fix64(sl[2 * j + 1], scL));
          }
        }
      }
      SP_WAIT_LGKM(); SP_TICK(4);
    }
    __syncthreads();
    SP_TICK(5);
    float* gpl = dst.plane[p] + (size_t)sweep * CH * ph * pw;
    const double invT = ldexp(1.0, -shT);
    for (int i = threadIdx.x; i < CH * CELLS; i += NT) {         // x fastest: 33 consecutive floats of one channel row
      const long long q = (long long)s_fx[i];
      if (q == 0) continue;
      const int c = i / CELLS, cell = i % CELLS;
      const int x = tx0 + cell % BCELL, y = ty0 + cell / BCELL;
      if (x < pw && y < ph) atomic_add_f32(gpl + ((size_t)c * ph + y) * pw + x, (float)((double)q * invT));
    }
    __syncthreads();
    SP_TICK(6);
    }                                                   // sweep
    a = seg_end;
  }
  if (lplane >= 0) flush_line();
  SP_TICK(7);
#ifdef LRF_SCATTER_PROF
  if (threadIdx.x == 0 && blockIdx.x < 2048) {
    sp_tk[11] = __builtin_readcyclecounter() - sp_t0;
    for (int i = 0; i < 12; ++i) g_scat_prof[APP ? 1 : 0][blockIdx.x][i] = sp_tk[i];
  }
#endif
}

#include "lrf_train32.inl"

struct BwdWorkspace {
  Workspace fw;
  float* feat; float* crgb; float* act; float* grd; float* rpart; float* wpart;
  float* depth; float* rgb;
  uint32_t* rowinfo; uint16_t* tid; int* hist; int* offs; int* cursor; uint32_t* list;
  uint32_t* relu_bits;       // [tile][layer 1, 2][lane]: ReLU masks, k_shade3<SAVE> -> k_train_dgrad3, k_wgrad_w2w3
  int4* tileinfo;            // [tile] (ray, j0, count, tile in ray), k_shade3<SAVE> -> k_train_dgrad3, k_train_app3
  float* gen;                // generic engine (lrf_generic.inl): [row][gen_row_ld] operands of the weight gradients, or null
  int* toff32;               // [R + 1] k_shade3's own tile offsets when they do not fit in its LDS (fw.toff holds the 16-row tiles')
  uint16_t* tid2; int* hist2; int* offs2; int* cursor2; uint32_t* list2;   // bins of the appearance scatter (runs beside the density scatter)
  uint32_t nmax;
  size_t bytes;
};
static BwdWorkspace carve_bwd(void* ws, int R, int S, const int32_t grid[3], int gen_ld = 0 /* generic engine: floats per row of weight-gradient operands */) {
  BwdWorkspace b;
  b.fw = carve(ws, R, S);
  char* p = reinterpret_cast<char*>(ws);
  size_t off = b.fw.bytes;
  const Layout L = make_layout(grid);
  const size_t rows = (size_t)R * b.fw.pmax * 16;
  const size_t nch = WGRAD_MAXCH;
  auto take = [&](size_t nfloat) { float* q = reinterpret_cast<float*>(p + off); off += up256(nfloat * 4); return q; };
  b.feat = take((size_t)R * S);
  b.crgb = take((size_t)R * S * 3);
  b.act = take(rows * ACT_LD);
  b.grd = take(rows * GRD_LD);
  b.rpart = take((size_t)R * b.fw.pmax * 8);
  b.wpart = take(nch * WP_FLOATS);
  b.depth = take(R);
  b.rgb = take((size_t)R * 3);
  b.nmax = (uint32_t)rows;                                     // rows >= R*S
  b.rowinfo = reinterpret_cast<uint32_t*>(take(rows));
  b.tid = reinterpret_cast<uint16_t*>(take((3 * rows + 1) / 2 + 2));
  b.hist = reinterpret_cast<int*>(take(BIN_CLEAR_WORDS));       // histogram, then the fill pass's cursors, then max|contribution| (k_bwd_ray): cleared by k_clear_bins
  b.cursor = b.hist + BIN_MAX;
  b.offs = reinterpret_cast<int*>(take(2 * (BIN_MAX + 1)));     // entry offsets | cost offsets (k_bin_fill)
  b.list = reinterpret_cast<uint32_t*>(take(3 * rows));
  b.relu_bits = reinterpret_cast<uint32_t*>(take(rows / 16 * 128));
  b.tileinfo = reinterpret_cast<int4*>(take(rows / 16 * 4));
  b.toff32 = reinterpret_cast<int*>(take((size_t)R + 1));
  b.tid2 = reinterpret_cast<uint16_t*>(take((3 * rows + 1) / 2 + 2));
  b.hist2 = reinterpret_cast<int*>(take(BIN_CLEAR_WORDS));      // (+ max|contribution| of the appearance scatter, k_train_app3)
  b.cursor2 = b.hist2 + BIN_MAX;
  b.offs2 = reinterpret_cast<int*>(take(2 * (BIN_MAX + 1)));
  b.list2 = reinterpret_cast<uint32_t*>(take(3 * rows));
  b.gen = gen_ld ? take(rows * (size_t)gen_ld) : nullptr;
  b.bytes = off;
  return b;
}

// The row-saving colour kernel of the training forward is the eval kernel itself: k_shade3<SAVE> (lrf_shade3.inl).
// (Rounds 1-3 ran a separate 16-sample kernel, k_bwd_shade_fwd, behind k_scan_tiles: 166-179 us at BASELINE configs[1] with
// the rows down to 160 B per sample; round 4 first tried the 32-sample chain on a static split of tile pairs: 318 us, see
// lrf_train32.inl.)
static int g_dgrad_dbg = 0;         // lrf_debug_set_train_fwd_engine bits 32 / 64 / 128: k_train_dgrad3 + k_train_app3 without row stores / position gradient and X / dz1 products (timing only)
static int g_scatter_fused = 1;     // lrf_debug_set_train_fwd_engine(8 | ...): separate plane / line scatter kernels (measurement)
static int g_scatter_fix = 3;       // bit 0: density, bit 1: appearance (where its accumulators fit in LDS) through k_scatter_fix.  lrf_debug_set_train_fwd_engine(16 | ...): both through the compare-and-swap kernels of rounds 2-5 (the tests compare the two); 256: the density alone
static int g_wgrad_kt = 64;          // rows per step of k_wgrad_w2w3 (lrf_debug_set_train_fwd_engine(512 | ...): 128, one workgroup per CU)
static int g_wgrad_split = 1;       // lrf_debug_set_bwd_overlap(1 + 2 * (n + 1)): n > 0 = k_wgrad_w2w3 on the caller's stream, 0 = on the side stream
static hipError_t launch_shade_save(DField d, const float* rays, const float* z, int S, int R, uint32_t flags, const Workspace& w,
                                    const BwdWorkspace& b, float* rgb, hipStream_t st) {
  if (!gen_is_default(d.fea_pe, d.view_pe, d.fc) || (flags & LRF_FLAG_MLP_VALU)) {          // generic engine (lrf_generic.inl): same saved state, no mask bits
    const GenCfg gc = gen_cfg(d.fea_pe, d.view_pe, d.fc, !(flags & LRF_FLAG_PE_OFF));
    hipLaunchKernelGGL(k_scan_tiles_n<ITEM3>, dim3(1), dim3(1024), 0, st, w.ncomp, R, b.toff32);
    hipLaunchKernelGGL(k_toff16, dim3((R + 256) / 256), dim3(256), 0, st, b.toff32, R, w.toff);
    hipError_t ge = launch_shade_gen(d, gc, rays, z, R, S, w, b.toff32, b.crgb, b.act, b.tileinfo, st);
    if (ge != hipSuccess) return ge;
    hipLaunchKernelGGL(k_finalize, dim3((R + 255) / 256), dim3(256), 0, st, R, w.pmax, flags, w.ncomp, w.acc, w.part, rgb, (float*)nullptr, d.perm);
    return hipGetLastError();
  }
  const SaveOut3 sv{b.crgb, b.act, b.relu_bits, b.tileinfo, w.toff};
  return launch_shade3(d, rays, z, R, S, flags, w, b.toff32, rgb, nullptr, &sv, nullptr, st);
}

}  // namespace lrf

extern "C" void lrf_debug_set_train_fwd_engine(int e) { lrf::g_scatter_fused = (e & 8) ? 0 : 1; lrf::g_scatter_fix = (e & 16) ? 0 : ((e & 256) ? 1 : ((e & 1024) ? 7 : 3)); lrf::g_dgrad_dbg = (e >> 5) & 7; lrf::g_wgrad_kt = (e & 512) ? 128 : 64; }

namespace lrf {
// floats per row of weight-gradient operands when the backward runs the generic engine (any non-default network, or
// LRF_FLAG_MLP_VALU on the default one: the exact-fp32 training path), else 0
static int field_gen_ld(int fea_pe, int view_pe, int fc, uint32_t flags) {
  fc = fc ? fc : LRF_FEATC;
  return (gen_is_default(fea_pe, view_pe, fc) && !(flags & LRF_FLAG_MLP_VALU)) ? 0 : gen_row_ld(gen_cfg(fea_pe, view_pe, fc, true));
}
}  // namespace lrf
extern "C" size_t lrf_workspace_bytes_bwd(int32_t R, int32_t S, const int32_t grid[3]) {
  return lrf::carve_bwd(nullptr, R, S, grid).bytes;
}
extern "C" size_t lrf_workspace_bytes_bwd_cfg(int32_t R, int32_t S, const int32_t grid[3], int32_t fea_pe, int32_t view_pe, int32_t feature_c, uint32_t flags) {
  return lrf::carve_bwd(nullptr, R, S, grid, lrf::field_gen_ld(fea_pe, view_pe, feature_c, flags)).bytes;
}

// Byte offsets of the pieces of the training workspace a test may want to look at (debug / parity
// diagnostics only: e.g. comparing the ReLU masks of the saved activation rows with the reference's).
extern "C" void lrf_workspace_layout_bwd(int32_t R, int32_t S, const int32_t grid[3], uint64_t out[9]) {
  using namespace lrf;
  const BwdWorkspace b = carve_bwd(nullptr, R, S, grid);
  auto off = [](const void* p) { return (uint64_t)reinterpret_cast<uintptr_t>(p); };
  out[0] = off(b.act); out[1] = off(b.grd); out[2] = off(b.rowinfo); out[3] = off(b.fw.toff);
  out[4] = (uint64_t)ACT_LD; out[5] = (uint64_t)GRD_LD; out[6] = off(b.relu_bits); out[7] = off(b.fw.perm);
  out[8] = off(b.feat);
}

extern "C" int lrf_render_fwd_train(const LrfField* f, const float* rays, const float* z, int32_t R, int32_t S,
                                    uint32_t flags, float* rgb, float* depth, void* workspace, void* stream) {
  using namespace lrf;
  if (!f || !f->cache || !rays || !z || !rgb || !depth || !workspace) return set_err("lrf_render_fwd_train: null argument");
  if (R <= 0 || S < 2 || S > LRF_MAX_S_TRAIN)
    return set_err("lrf_render_fwd_train: need R > 0 and 2 <= S <= LRF_MAX_S_TRAIN (2048: the per-ray backward keeps 16 B per sample in LDS)");
  if (flags & (LRF_FLAG_MLP_VALU | LRF_FLAG_MLP_F32))
    return set_err("lrf_render_fwd_train: the row-saving forward runs the split-bf16 engine only");
  if (flags & ~LRF_FLAG_ALL) return set_err("lrf_render_fwd_train: unknown flag bits");
  if (const char* bad = gen_check(f)) return set_err(bad);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  DField d = make_dfield(f);
  const BwdWorkspace b = carve_bwd(workspace, R, S, f->grid, field_gen_ld(f->fea_pe, f->view_pe, f->feature_c, flags));
  const Workspace& w = b.fw;
  rays = sort_rays_if_asked(d, rays, R, flags, w, st);
  d.rdir = w.rdir;
  launch_march(d, rays, z, R, S, flags, 0.0f, depth, w.acc, nullptr, w.ncomp, w.cidx, w.cw, b.feat, st);
  LRF_HIP(launch_shade_save(d, rays, z, S, R, flags, w, b, rgb, st));
  LRF_HIP(hipGetLastError());
  return 0;
}

extern "C" int lrf_render_bwd(const LrfField* f, const LrfParams* p, const float* rays, const float* z,
                              int32_t R, int32_t S, uint32_t flags, const float* g_rgb, const float* g_depth,
                              const LrfGrads* g, float* g_rays, void* workspace, void* stream) {
  using namespace lrf;
  if (!f || !f->cache || !p || !rays || !z || !g_rgb || !g_depth || !g || !g_rays || !workspace)
    return set_err("lrf_render_bwd: null argument");
  if (R <= 0 || S < 2 || S > LRF_MAX_S_TRAIN)
    return set_err("lrf_render_bwd: need R > 0 and 2 <= S <= LRF_MAX_S_TRAIN (2048: the per-ray backward keeps 16 B per sample in LDS)");
  if (flags & ~LRF_FLAG_ALL) return set_err("lrf_render_bwd: unknown flag bits");
  if (const char* bad = gen_check(f)) return set_err(bad);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  DField d = make_dfield(f);
  const Layout L = make_layout(f->grid);
  const int gen_ld = field_gen_ld(f->fea_pe, f->view_pe, f->feature_c, flags);
  const bool generic = gen_ld != 0;
  const BwdWorkspace b = carve_bwd(workspace, R, S, f->grid, gen_ld);
  const Workspace& w = b.fw;
  const int cus = device_cus();
  d.rdir = w.rdir;                                         // per-ray unit directions: written by k_march (the saved forward's, or the one below), read by k_scatter_fix
  if (flags & LRF_FLAG_ROWS_SAVED) {                       // lrf_render_fwd_train sorted (or not) with the same flags: same workspace
    if ((flags & LRF_FLAG_SORT_RAYS) && R <= LRF_SORT_MAX_R && R >= 2) { d.perm = w.perm; rays = w.rays_s; }
  } else {
    rays = sort_rays_if_asked(d, rays, R, flags, w, st);
  }
  {
    static std::once_flag lds_attr_once[64];   // dynamic LDS above 64 KB has to be opted into once per device
    static hipError_t lds_attr_err[64];
    int dev_id = 0;
    LRF_HIP(hipGetDevice(&dev_id));
    std::call_once(lds_attr_once[dev_id & 63], [dev_id] {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_plane<LRF_CA, true, LRF_APP_NT, false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_plane<LRF_CA, true, LRF_APP_NT, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_plane<LRF_CD, false, 512, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_line<LRF_CA, true, 1024>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_line<LRF_CD, false, 1024>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bwd_ray),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 14 * LRF_MAX_S_TRAIN * 4 + BIN_MAX * 4);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_train_dgrad3<8>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_train_app3<8, false>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_train_app3<8, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_fix<LRF_CD, false, FIX_NT>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_fix<LRF_CA, true, FIX_NT>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_fix<LRF_CA, true, FIX_NT, 4>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_w2w3<128>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)w23_lds(128));
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_w2w3<64>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)w23_lds(64));
      lds_attr_err[dev_id & 63] = e;
    });
    LRF_HIP(lds_attr_err[dev_id & 63]);
  }
  if (!(flags & LRF_FLAG_ROWS_SAVED)) {            // otherwise lrf_render_fwd_train left all of this in place
    d.rdir = w.rdir;
    launch_march(d, rays, z, R, S, flags, 0.0f, b.depth, w.acc, nullptr, w.ncomp, w.cidx, w.cw, b.feat, st);
    LRF_HIP(launch_shade_save(d, rays, z, S, R, flags, w, b, b.rgb, st));
  }
  // The backward runs as two branches that share no outputs (g_bwd_overlap, default on):
  //   caller's stream: k_train_dgrad3 -> k_train_app3 [-> k_wgrad_w2w3] -> appearance bins + scatter   [-> join] -> ray partials
  //   side stream:     k_bwd_ray -> density bins + scatter [-> (go / dfeat rows there) k_wgrad_w2w3] -> (partials there) reduce
  // (which stream runs the weight-gradient kernel is g_wgrad_split; rounds 2 / 3 had four GEMMs to place: 2.53 / 2.45 / 2.53 /
  // 2.57 / 2.59 ms and 2.00 / 1.90 / 1.86 / 1.92 / 2.01 ms for 0..4 of them on the caller's stream, 2.20 ms on one stream)
  // k_bwd_ray and the density scatter need nothing from the data-gradient kernel (the appearance lookups' position
  // gradients it produces are added to d/d(rays) afterwards by k_rays_add_rpart), so the texture / LDS-atomic bound
  // per-ray work runs under the row-traffic bound colour-network backward instead of behind it.
  SideStream* sx = side_stream();                          // also holds the bucket events of lrf_render_bwd_wait
  // Under stream capture the two branches go on ONE stream: ROCm runs the branches of a graph one after the other anyway
  // (DESIGN.md s4e), and every cross-stream edge of the captured fork / join costs ~10 us of idle time in the replay (four of
  // them per backward: profiles/r18_graph_iteration_timeline.md).
  hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cap_status);
  const bool capturing = cap_status == hipStreamCaptureStatusActive;
  SideStream* ss = (g_bwd_overlap && !capturing) ? sx : nullptr;
  // the side stream and its fork / join events are per device: host threads that enqueue backward passes on the same
  // device take turns (enqueueing is ~0.3 ms of host time; the kernels themselves still overlap on the GPU)
  std::unique_lock<std::mutex> side_lock;
  if (sx) side_lock = std::unique_lock<std::mutex>(sx->mu);
  // the scatter kernels add straight into the reference-layout gradients: the density tensors are final as soon as the
  // per-ray branch is through (bucket 0), the appearance tensors at the very end (bucket 2)
  ScatterDst dst_d, dst_a;
  for (int q = 0; q < 3; ++q) {
    dst_d.plane[q] = g->density_plane[q]; dst_d.line[q] = g->density_line[q];
    dst_a.plane[q] = g->app_plane[q]; dst_a.line[q] = g->app_line[q];
  }
  if (g->zero_floats < 0 || (g->zero_floats & 3) || (g->zero_floats && (!g->zero_base || (reinterpret_cast<uintptr_t>(g->zero_base) & 15))))
    return set_err("lrf_render_bwd: LrfGrads.zero_base / zero_floats must name a 16-byte aligned range of a multiple of 4 floats (or NULL / 0)");
  {
    const long long zf4 = g->zero_floats / 4;
    const unsigned nblk_clear = (unsigned)(BIN_CLEAR_BLOCKS + (zf4 + ZERO_F4_PER_BLOCK - 1) / ZERO_F4_PER_BLOCK);
    hipLaunchKernelGGL(k_clear_bins, dim3(nblk_clear), dim3(256), 0, st, b.hist, b.hist2, reinterpret_cast<float4*>(g->zero_base), zf4);     // (in front of the fork: both branches count into these)
  }
  hipStream_t sb = st;
  if (ss) {
    LRF_HIP(hipEventRecord(ss->fork, st));
    LRF_HIP(hipStreamWaitEvent(ss->s, ss->fork, 0));
    sb = ss->s;
  }
  const BinGeom bg = make_bins(L);
  if (bg.total > BIN_MAX) return set_err("lrf_render_bwd: grid too large for the tile binning (BIN_MAX)");
  const int nblk = (int)((b.nmax + BIN_CHUNK - 1) / BIN_CHUNK);
  for (int q = 0; q < 3; ++q) if ((size_t)L.ll[q] * LRF_CA * 4 > 150 * 1024) return set_err("lrf_render_bwd: line too long for LDS accumulation");

  // ---- caller's stream: data gradient of the colour network, then its appearance half (dX, position gradient, dbasis)
  const int n_dgrad_wg = min(cus, WGRAD_MAXCH);            // one dW1 / dbasis partial block per workgroup (k_wgrad_reduce: fixed count)
  const GenCfg gc = gen_cfg(d.fea_pe, d.view_pe, d.fc, !(flags & LRF_FLAG_PE_OFF));
  if (generic) {       // lrf_generic.inl: one lane per row; worst-case grid, rows behind the batch's last tile return at once
    LRF_HIP(gen_opt_in());
    const int ls = gen_tile_samples(gc, true), nt = gen_block_threads(gc);
    const size_t lds = (size_t)gen_lds(gc, ls, true).total * 4;
    const dim3 grid((unsigned)((b.nmax + ls - 1) / ls));
    if (ls == 32) hipLaunchKernelGGL(k_gen_dgrad<32>, grid, dim3(nt), lds, st, d, gc, rays, S, w.toff, R, b.tileinfo, w.cidx, w.cw, b.crgb, g_rgb, b.act, b.grd, b.rowinfo, b.gen, gen_ld);
    else          hipLaunchKernelGGL(k_gen_dgrad<16>, grid, dim3(nt), lds, st, d, gc, rays, S, w.toff, R, b.tileinfo, w.cidx, w.cw, b.crgb, g_rgb, b.act, b.grd, b.rowinfo, b.gen, gen_ld);
  } else {
    hipLaunchKernelGGL((k_train_dgrad3<8>), dim3(n_dgrad_wg), dim3(512), (size_t)W32T_ALL_U4 * 16 + 4 * 64 * 16, st, d,
                       d.mlpwt, rays, S, w.toff, R, b.tileinfo, w.cidx, w.cw, b.crgb, g_rgb,
                       b.grd, b.rowinfo, b.relu_bits, b.act, b.wpart, g_dgrad_dbg & 5);
  }
  if (ss) LRF_HIP(hipEventRecord(ss->app[0], st));         // go / dfeat rows: the weight-gradient kernel may start
  const size_t ll_max = (size_t)max(L.ll[0], max(L.ll[1], L.ll[2]));
  const size_t lds_ap = sizeof(float) * BCELL * BCELL * LRF_CA, lds_al = sizeof(float) * LRF_CA * ll_max;
  const bool fuse_a = g_scatter_fused && lds_ap + lds_al <= 158 * 1024;
  // line gradients ride on the plane pass when tile + line accumulators fit in LDS (g_scatter_fused; appearance at 640^3 does not)
  const size_t lds_dp = sizeof(float) * BCELL * BCELL * LRF_CD, lds_dl = sizeof(float) * LRF_CD * ll_max;
  const bool fuse_d = g_scatter_fused && lds_dp + lds_dl <= 64 * 1024;
  const bool fix_d = g_scatter_fix && g_scatter_fused && 2 * (lds_dp + lds_dl) <= 158 * 1024;   // 64-bit fixed-point accumulators: twice the bytes
  // the appearance scatter runs on fixed point too where an 8-channel tile + 24-channel line accumulators of 64-bit cells fit
  // in LDS (lines up to 479 cells): forward + backward 1.26 -> 1.12 ms at 64^3, 1.36 -> 1.29 at 300^3, on par with the
  // compare-and-swap kernel at 400^3-460^3 (profiles/r17_fixed_point_scatter.md); above that the compare-and-swap kernel stays
  const size_t lds_fa = 2 * lds_dp + sizeof(unsigned long long) * LRF_CA * ll_max;
  const bool fix_a8 = fix_d && (g_scatter_fix & 2) && lds_fa <= 158 * 1024;
  // ... and with FOUR channels per sweep (six sweeps, a 34.8 KB tile) where only that leaves room for the lines: 480 .. 640 cells
  const size_t lds_fa4 = lds_dp + sizeof(unsigned long long) * LRF_CA * ll_max;
  const bool fix_a4 = fix_d && (g_scatter_fix & 2) && !fix_a8 && !(g_scatter_fix & 4) && lds_fa4 <= 158 * 1024;
  const bool fix_a = fix_a8 || fix_a4;
  unsigned* vmax_a = reinterpret_cast<unsigned*>(b.hist2 + 2 * BIN_MAX);
  if (fix_a)
    hipLaunchKernelGGL((k_train_app3<8, true>), dim3(n_dgrad_wg), dim3(512), app3_lds_bytes(S, 8, bg.total), st, d,
                       d.mlpwt, rays, z, S, w.toff, R, b.tileinfo, w.cidx,
                       b.grd, b.rpart, w.pmax, b.wpart, bg, b.tid2, b.hist2, b.nmax, g_dgrad_dbg & 3, vmax_a);
  else
    hipLaunchKernelGGL((k_train_app3<8, false>), dim3(n_dgrad_wg), dim3(512), app3_lds_bytes(S, 8, bg.total), st, d,
                       d.mlpwt, rays, z, S, w.toff, R, b.tileinfo, w.cidx,
                       b.grd, b.rpart, w.pmax, b.wpart, bg, b.tid2, b.hist2, b.nmax, g_dgrad_dbg & 3, vmax_a);

  // ---- side stream: per-ray backward, density scatter
  unsigned* vmax_d = reinterpret_cast<unsigned*>(b.hist + 2 * BIN_MAX);
  hipLaunchKernelGGL(k_bwd_ray, dim3((R + 3) / 4), dim3(256), (size_t)14 * S * sizeof(float) + (size_t)bg.total * sizeof(int), sb,
                     d, rays, z, R, S, flags, b.feat, w.ncomp, w.cidx, b.crgb, g_rgb, g_depth,
                     (const float*)nullptr, w.pmax, g_rays, bg, b.tid, b.hist, b.nmax, vmax_d);
  if (fix_d) {
    hipLaunchKernelGGL(k_bin_fill, dim3(nblk), dim3(256), 0, sb, bg, b.nmax, R, S, w.toff, 0, b.tid, b.hist, b.cursor, b.offs, b.list);
    hipLaunchKernelGGL((k_scatter_fix<LRF_CD, false, FIX_NT>), dim3(cus), dim3(FIX_NT), 2 * (lds_dp + lds_dl), sb,
                       d, bg, dst_d, rays, z, S, b.offs, b.list, b.feat, b.rowinfo, b.grd, vmax_d, 0, bg.total);
  } else if (fuse_d) {
    hipLaunchKernelGGL(k_bin_fill, dim3(nblk), dim3(256), 0, sb, bg, b.nmax, R, S, w.toff, 0, b.tid, b.hist, b.cursor, b.offs, b.list);
    hipLaunchKernelGGL((k_scatter_plane<LRF_CD, false, 512, true>), dim3(cus * LRF_DPLANE_MULT), dim3(512), lds_dp + lds_dl, sb,
                       d, bg, dst_d, rays, z, S, b.offs, b.list, b.feat, b.rowinfo, b.grd, 0, bg.total);
  } else {
    hipLaunchKernelGGL(k_bin_fill, dim3(nblk), dim3(256), 0, sb, bg, b.nmax, R, S, w.toff, 0, b.tid, b.hist, b.cursor, b.offs, b.list);
    hipLaunchKernelGGL((k_scatter_plane<LRF_CD, false, 512, false>), dim3(cus * LRF_DPLANE_MULT), dim3(512), lds_dp, sb,
                       d, bg, dst_d, rays, z, S, b.offs, b.list, b.feat, b.rowinfo, b.grd, 0, bg.total);
    hipLaunchKernelGGL((k_scatter_line<LRF_CD, false, 1024>), dim3(3 * LINE_WGS), dim3(1024), lds_dl, sb,
                       d, dst_d, rays, z, R, S, w.toff, b.feat, b.rowinfo, b.grd);
  }
  if (sx) LRF_HIP(hipEventRecord(sx->bucket[0], sb));

  // ---- dW2 / dW3 (row reads, matrix pipe): on the caller's stream behind the appearance kernel (g_wgrad_split > 0), or on the
  // side stream behind the density scatter once the go / dfeat rows are there (g_wgrad_split == 0)
  const int nch_max = WGRAD_MAXCH;      // (blocks behind the last chunk of the actual row count return at once)
  const bool w23_on_st = !ss || g_wgrad_split > 0;
  if (!w23_on_st) LRF_HIP(hipStreamWaitEvent(sb, ss->app[0], 0));
  if (generic) {       // dW = A^T B over the operand rows k_gen_dgrad left (added straight into the reference-layout gradients)
    const GenRowOff ro = gen_row_off(gc);
    const int nchunk = (int)((b.nmax + GEN_GEMM_CHUNK - 1) / GEN_GEMM_CHUNK);
    auto gemm = [&](int offA, int M, int offB, int N, float* dW, int ldw, float* db) {
      hipLaunchKernelGGL(k_gen_gemm, dim3(((M + 63) / 64) * ((N + 63) / 64), nchunk), dim3(256), 0, w23_on_st ? st : sb,
                         b.gen, gen_ld, offA, M, offB, N, w.toff, R, dW, ldw, db);
    };
    gemm(ro.dz1, gc.fc, ro.x1, gc.in1 + 1, g->w1, gc.in1, g->b1);
    gemm(ro.dz2, gc.fc, ro.h1, gc.fc + 1, g->w2, gc.fc, g->b2);
    gemm(ro.go, 3, ro.h2v, gc.fc + gc.inv + 1, g->w3, gc.fc + gc.inv, g->b3);
  } else {
    if (g_wgrad_kt == 64)
      hipLaunchKernelGGL(k_wgrad_w2w3<64>, dim3(nch_max), dim3(512), w23_lds(64), w23_on_st ? st : sb, d.mlpb, b.act + 16 * ACT_FEAT, ACT_LD, b.grd + 16 * GRD_GO, GRD_LD,
                         b.relu_bits, p->w3, p->w2, p->b2, w.toff, R, b.wpart);
    else
      hipLaunchKernelGGL(k_wgrad_w2w3<128>, dim3(nch_max), dim3(512), w23_lds(128), w23_on_st ? st : sb, d.mlpb, b.act + 16 * ACT_FEAT, ACT_LD, b.grd + 16 * GRD_GO, GRD_LD,
                         b.relu_bits, p->w3, p->w2, p->b2, w.toff, R, b.wpart);
  }
  if (ss) LRF_HIP(hipEventRecord(ss->app[1], st));          // the caller's-stream partials (dW1, dbasis[, dW2, dW3]) are complete behind this
  {
    WgradSegs segs;
    for (int q = 0; q < 7; ++q) segs.s[q] = WgradSeg{0, 1, 0, 0, 1, 1, 0x7fffffff, 0, 0, nullptr};    // unused slots: behind every element
    int nseg = 0, elems = 0;
    auto seg = [&](int off, int ld, int n_off, int m, int n, float* dst, int dst_ld, int x_slots = 0, int nch = 0) {
      segs.s[nseg++] = WgradSeg{off, ld, n_off, m, n, dst_ld, elems, x_slots, nch, dst};
      elems += m * n;
    };
    if (!generic) {
      seg(WP_W2, 144, 0, 128, 128, g->w2, 128);
      seg(WP_W2, 144, 128, 128, 1, g->b2, 1);
      seg(WP_W1, 32, 0, 128, LRF_APP_DIM, g->w1, LRF_APP_DIM, 0, n_dgrad_wg);     // accumulated by k_train_dgrad3: one block per workgroup
      seg(WP_W1, 32, LRF_APP_DIM, 128, 1, g->b1, 1, 0, n_dgrad_wg);
    }
    seg(WP_BAS, 96, 0, LRF_APP_DIM, 72, g->basis, 72, 1, n_dgrad_wg);   // accumulated by k_train_app3: one block per workgroup
    if (!generic) {
      seg(WP_W3, 144, 0, 3, LRF_FEATC + 3, g->w3, LRF_FEATC + 3);
      seg(WP_W3, 144, LRF_FEATC + 3, 3, 1, g->b3, 1);
    }
    segs.total_elems = elems;
    if (ss) LRF_HIP(hipStreamWaitEvent(sb, ss->app[1], 0));     // partials written on the caller's stream
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((elems * 16 + 255) / 256), dim3(256), 0, sb, b.wpart, w.toff, R, segs);
  }
  if (sx) LRF_HIP(hipEventRecord(sx->bucket[1], sb));
  if (ss) LRF_HIP(hipEventRecord(ss->join, sb));

  // ---- caller's stream: appearance scatter (its own bin buffers: the density scatter may still be running)
  hipLaunchKernelGGL(k_bin_fill, dim3(nblk), dim3(256), 0, st, bg, b.nmax, R, S, w.toff, 1, b.tid2, b.hist2, b.cursor2, b.offs2, b.list2);
  // LRF_FLAG_PLANE_EVENTS (data parallel): one pass per plane, an event behind planes 0 and 1 -- a collective over plane p's
  // gradient (8.6 MB each at 300^3) starts while the later planes are still being scattered; otherwise one pass over all bins
  const int npass = (flags & LRF_FLAG_PLANE_EVENTS) ? 3 : 1;
  for (int q = 0; q < npass; ++q) {
    const int blo = npass == 1 ? 0 : bg.base[q], bhi = (npass == 1 || q == 2) ? bg.total : bg.base[q + 1];
    if (fix_a8) {
      hipLaunchKernelGGL((k_scatter_fix<LRF_CA, true, FIX_NT>), dim3(cus), dim3(FIX_NT), lds_fa, st,
                         d, bg, dst_a, rays, z, S, b.offs2, b.list2, b.feat, b.rowinfo, b.grd, vmax_a, blo, bhi);
    } else if (fix_a4) {
      hipLaunchKernelGGL((k_scatter_fix<LRF_CA, true, FIX_NT, 4>), dim3(cus), dim3(FIX_NT), lds_fa4, st,
                         d, bg, dst_a, rays, z, S, b.offs2, b.list2, b.feat, b.rowinfo, b.grd, vmax_a, blo, bhi);
    } else if (fuse_a) {
      hipLaunchKernelGGL((k_scatter_plane<LRF_CA, true, LRF_APP_NT, true>), dim3(cus), dim3(LRF_APP_NT), lds_ap + lds_al, st,
                         d, bg, dst_a, rays, z, S, b.offs2, b.list2, b.feat, b.rowinfo, b.grd, blo, bhi);
    } else {
      hipLaunchKernelGGL((k_scatter_plane<LRF_CA, true, LRF_APP_NT, false>), dim3(cus), dim3(LRF_APP_NT), lds_ap, st,
                         d, bg, dst_a, rays, z, S, b.offs2, b.list2, b.feat, b.rowinfo, b.grd, blo, bhi);
    }
    if (npass == 3 && q < 2 && sx) LRF_HIP(hipEventRecord(sx->bucket[3 + q], st));    // app_plane[q] is final (its line only if fused: bucket 2)
  }
  if (!fuse_a && !fix_a)
    hipLaunchKernelGGL((k_scatter_line<LRF_CA, true, 1024>), dim3(3 * LINE_WGS), dim3(1024), lds_al, st,
                       d, dst_a, rays, z, R, S, w.toff, b.feat, b.rowinfo, b.grd);

  // ---- join: both branches done
  if (ss) LRF_HIP(hipStreamWaitEvent(st, ss->join, 0));
  hipLaunchKernelGGL(k_rays_add_rpart, dim3((R + 255) / 256), dim3(256), 0, st, rays, R, w.ncomp, b.rpart, w.pmax, g_rays, d.perm);
  if (sx) {
    LRF_HIP(hipEventRecord(sx->bucket[2], st));
    if (npass == 1) { LRF_HIP(hipEventRecord(sx->bucket[3], st)); LRF_HIP(hipEventRecord(sx->bucket[4], st)); }   // no per-plane passes: the planes are final with everything else
    sx->bucket_set = true;
  }
  LRF_HIP(hipGetLastError());
  return 0;
}

extern "C" int lrf_render_bwd_wait(int32_t bucket, void* stream) {
  using namespace lrf;
  if (bucket < 0 || bucket > 4) return set_err("lrf_render_bwd_wait: bucket must be 0 (density), 1 (colour network), 2 (appearance = all), 3 or 4 (appearance plane 0 / 1)");
  SideStream* sx = side_stream();
  if (!sx) return set_err("lrf_render_bwd_wait: no event resources on this device");
  std::lock_guard<std::mutex> lk(sx->mu);
  if (!sx->bucket_set) return set_err("lrf_render_bwd_wait: no lrf_render_bwd has run on this device");
  LRF_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), sx->bucket[bucket], 0));
  return 0;
}
