// lrf_common.h -- layout of the derived field cache and device helpers shared by the
// gfx950 kernels.  Geometry helpers restate, per sample, what the reference does with
// whole-tensor ATen ops (file:line relative to /root/reference/localTensoRF).
#pragma once
#include "lrf_tu.h"          // first: which translation unit this is (renames, see there)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lrf.h"

namespace lrf {

// ------------------------------------------------------------------ cache layout
// Planes are stored channel-last [H][W][C] so one bilinear tap is one contiguous
// 32 B (density) / 128 B (appearance, padded) read; lines are [L][C].  Offsets are in floats and
// 64-float aligned.  The MLP image is in MFMA-fragment order (see lrf_mlp_image.h).
constexpr int MAT0[3] = {0, 0, 1};   // matMode[p][0]  (tensorBase.py:274) -> plane W axis
constexpr int MAT1[3] = {1, 2, 2};   // matMode[p][1]                      -> plane H axis
constexpr int VEC[3]  = {2, 1, 0};   // vecMode[p]     (tensorBase.py:275)

// Appearance planes/lines are stored with a channel stride of 32 floats (128 B = one cache
// line per texel): the 24 channels sit in four groups of 6 + 2 zero pads, channel c at slot
// app_pc(c) = 8*(c/6) + c%6, so lane group g of a shading tile reads its six channels as two
// ALIGNED float4 at slot 8g.  (The dense 24-channel layout made the compiler fuse the three
// 8-byte loads of a lane into a dwordx4 that is only 8-byte aligned for odd g; on MI355X such
// loads occasionally delivered a stale half -- the root cause of run-to-run differences in a
// few tiles per launch, see DESIGN.md.)
constexpr int LRF_CAS = 32;
__host__ __device__ constexpr int app_pc(int c) { return 8 * (c / 6) + c % 6; }

// MLP image (floats)
constexpr int IMG_BAS  = 0;                       // [t'2][p3][lane64][8]  basis_mat, 6 of 8 used
constexpr int IMG_W1   = IMG_BAS + 2 * 3 * 64 * 8;   // [t'8][t2][lane64][4]
constexpr int IMG_W2   = IMG_W1 + 8 * 2 * 64 * 4;    // [t'8][t8][lane64][4]
constexpr int IMG_W3H  = IMG_W2 + 8 * 8 * 64 * 4;    // [g4][f32][4]   mlp_view weight, hidden part
constexpr int IMG_B1   = IMG_W3H + 4 * 32 * 4;       // [128]
constexpr int IMG_B2   = IMG_B1 + 128;               // [128]
constexpr int IMG_W3V  = IMG_B2 + 128;               // [o4][4] = (wx,wy,wz,bias) per colour
constexpr int IMG_FLOATS = IMG_W3V + 16;             // 24336 floats = 97,344 B

// Split-bf16 MLP image (16-byte units, then a float tail).  Every weight w is stored as
// hi = bf16(w), lo = bf16(w - hi); a fragment is [part hi/lo][lane64][8 bf16] and feeds the
// A operand of v_mfma_f32_16x16x32_bf16.  K-slot (g = lane>>4, j) of k-step ks maps to input
// feature: basis k-step p (a plane): j<6 -> channel 6g+j, j>=6 -> zero; layers 1/2: tile 2ks+(j>>2), feature
// 16*tile+4g+(j&3) -- i.e. exactly the D registers the lane already holds.
constexpr int IMGB_BAS = 0;                        // frags [t'2][ks3]
constexpr int IMGB_W1  = IMGB_BAS + 2 * 3 * 128;   // frags [t'8][ks1]
constexpr int IMGB_W2  = IMGB_W1 + 8 * 1 * 128;    // frags [t'8][ks4]
constexpr int IMGB_TAIL = IMGB_W2 + 8 * 4 * 128;   // uint4 index of the float tail
constexpr int TAIL_W3H = 0;                        // floats [g4][33][4]: 32 features + one pad slot per lane
constexpr int TAIL_W3H_GS = 132;                   // group, so the groups' float4 reads fall on different banks
constexpr int TAIL_B1 = 528, TAIL_B2 = 656, TAIL_W3V = 784, TAIL_FLOATS = 800;
constexpr int IMGB_U4 = IMGB_TAIL + TAIL_FLOATS / 4;   // 6088 uint4 = 97,408 B: what the fused kernels keep in LDS
// (round 2's k_mlp kept mlp_view.0.weight[:, :128] as four more A fragments behind the tail; the slot stays reserved)
constexpr int IMGB_W3F = IMGB_TAIL + 256;              // fragment-aligned (multiples of 128 uint4), behind the tail
constexpr int IMGB_ALL = IMGB_W3F + 4 * 128;           // 6656 uint4 = 106,496 B
static_assert(IMGB_W3F % 128 == 0 && IMGB_W3F >= IMGB_U4 && IMGB_W1 % 128 == 0, "fragments are addressed in units of 128 uint4");

// ---- image of the colour network for the 32-sample kernel (k_shade3, lrf_shade3.inl): v_mfma_f32_32x32x16_bf16.
// Lane l = (n = l & 31: output row of the A operand / sample column of B and D, h = l >> 5: K half).  A fragment is
// [part hi, lo][lane64][8 bf16] (2 KB): slot j of lane (n, h) = A[n][8 h + j].  D register r of lane (n, h) is row
// 8 (r >> 2) + 4 h + (r & 3), column n -- so registers 8 q .. 8 q + 7 of a 32-row output tile m0 are the B operand of
// the next layer's K-step (m0, q) once the weights are packed with the K permutation w32_unit(): no lane movement.
//   frag ks            (0..4)  basis: A[n][slot j] = basis[n][w32_chan(h, 8 ks + j)]   (rows 27..31 zero)
//   frag 5 + 2 m + q           W1:    A[n][slot j] = W1[32 m + n][w32_unit(0, q, h, j)] (units >= 27 zero)
//   frag 13 + 8 m + 2 m0 + q   W2:    A[n][slot j] = W2[32 m + n][w32_unit(m0, q, h, j)]
// then an fp32 tail: b1[128] | b2[128] | mlp_view.0.weight rows padded to 132 | b3.
constexpr int W32_BAS = 0, W32_W1 = 5, W32_W2 = 13, W32_NFRAG = 45;
constexpr int W32_U4 = W32_NFRAG * 2 * 64;                      // 5760 uint4 = 92,160 B
constexpr int W32_T_B1 = 0, W32_T_B2 = 128, W32_T_W3 = 256, W32_T_W3_LD = 132, W32_T_B3 = 652, W32_T_FLOATS = 672;
constexpr int W32_ALL_U4 = W32_U4 + W32_T_FLOATS / 4;           // 5928 uint4 = 94,848 B
__host__ __device__ constexpr int w32_unit(int m0, int q, int h, int j) { return 32 * m0 + 16 * q + 8 * (j >> 2) + 4 * h + (j & 3); }
// gathered value v (0..39) of lane half h -> appearance channel 0..71 (plane v / 12, channels 12 h .. 12 h + 11 of the
// dense 24-channel texel), -1 for the four pad slots of the fifth K-step
__host__ __device__ constexpr int w32_chan(int h, int v) { return v >= 36 ? -1 : 24 * (v / 12) + 12 * h + v % 12; }

// transposed w32 image of the backward (fragment order: lrf_train32.inl)
constexpr int W32T_W2 = 0, W32T_W1 = 32, W32T_BAS = 40, W32T_NFRAG = 46;
constexpr int W32T_U4 = W32T_NFRAG * 128;                        // 5888 uint4 = 94,208 B
constexpr int W32T_T_W3 = 0, W32T_T_W3_LD = 132, W32T_T_FLOATS = 400;
constexpr int W32T_ALL_U4 = W32T_U4 + W32T_T_FLOATS / 4;         // 5988 uint4 = 95,808 B

struct Layout {
  size_t dplane[3], dline[3], aplane[3], aline[3], mlp, mlpb, total;   // float offsets
  size_t aplane2[3], aline2[3], mlpw;     // dense 24-channel appearance planes / lines and the w32 image (k_shade3)
  size_t mlpwt;                           // transposed w32 image (k_train_dgrad3, k_train_app3)
  int pw[3], ph[3], ll[3];
};

__host__ __device__ inline size_t align64(size_t x) { return (x + 63) & ~size_t(63); }

__host__ __device__ inline Layout make_layout(const int32_t grid[3]) {
  Layout L;
  size_t off = 0;
  for (int p = 0; p < 3; ++p) {
    L.pw[p] = grid[MAT0[p]]; L.ph[p] = grid[MAT1[p]]; L.ll[p] = grid[VEC[p]];
  }
  for (int p = 0; p < 3; ++p) { L.dplane[p] = off; off = align64(off + (size_t)L.pw[p] * L.ph[p] * LRF_CD); }
  for (int p = 0; p < 3; ++p) { L.dline[p]  = off; off = align64(off + (size_t)L.ll[p] * LRF_CD); }
  for (int p = 0; p < 3; ++p) { L.aplane[p] = off; off = align64(off + (size_t)L.pw[p] * L.ph[p] * LRF_CAS); }
  for (int p = 0; p < 3; ++p) { L.aline[p]  = off; off = align64(off + (size_t)L.ll[p] * LRF_CAS); }
  L.mlp = off; off = align64(off + IMG_FLOATS);
  L.mlpb = off; off = align64(off + (size_t)IMGB_ALL * 4);
  for (int p = 0; p < 3; ++p) { L.aplane2[p] = off; off = align64(off + (size_t)L.pw[p] * L.ph[p] * LRF_CA); }
  for (int p = 0; p < 3; ++p) { L.aline2[p]  = off; off = align64(off + (size_t)L.ll[p] * LRF_CA); }
  L.mlpw = off; off = align64(off + (size_t)W32_ALL_U4 * 4);
  L.mlpwt = off; off = align64(off + (size_t)W32T_ALL_U4 * 4);
  L.total = off;
  return L;
}

// Device-side view of a field, passed by value to kernels.
// ---- rows the training forward saves per shaded sample for the backward (floats; lrf_backward.inl) --------------
//   ACT row: feat[27], 1 (+pad)                                                               128 B
//   GRD row: go[3], 0, dhat[3], 1 (+pad) | dfeat | dX                                         512 B
// The hidden activations are NOT rows (round 4): relu(h1) and relu(h2) are recomputed from `feat` by the one kernel that
// needs them as GEMM operands (k_wgrad_w2w3: dW2 = dz2^T relu(h1), dW3 = go^T relu(h2)); the data-gradient kernel needs
// only their signs (relu_bits, 32 B per sample).  That removed 1.15 KB written + 1.15 KB read per shaded sample.  dz1 is not
// a row either: dW1 = dz1^T [feat | 1] is accumulated inside the data-gradient kernel (lrf_train32.inl), 0.5 + 0.5 KB more.
// The plane x line products X are not a row either (they were 320 B, read once for dbasis = dfeat^T X): the appearance
// kernel of the backward (k_train_app3) gathers the taps anyway for the position gradient, so it re-forms X and
// accumulates dbasis in registers.
// Layout in memory: MFMA-fragment order, not row-major.  The 16 rows of a tile are stored together; inside a tile every
// 16-column block is the 1 KB a wave holds for it, [lane group g][sample s][4 columns]: column c of row s of tile t is at
//   t * 16 * LD + (c / 16) * 256 + ((c / 4 % 4) * 16 + s) * 4 + c % 4            (frag_off below)
// so that a producer's store instruction (lane (s, g) holds columns 16 b + 4 g .. + 3 of row s) writes 1 KB contiguous
// instead of sixteen 64-byte pieces apart, and the weight-gradient kernel stages whole blocks with contiguous loads.
// Exception: the dX block of a GRD tile (16 x 80 floats behind the 3 fragment blocks), in one of two orders chosen per
// backward pass (lrf_render_bwd: the order its appearance scatter wants):
//   rows    row-major inside the tile, natural channel order (rounds 2-5): k_scatter_plane<24> / k_scatter_line<24> read the 96
//           bytes of a (row, plane) pair at once (grd_dx_row);
//   groups  [plane 3][channel group 3][row 16][8 channels]: the 32 bytes a (row, plane) pair contributes to one 8-channel sweep
//           of k_scatter_fix<24> lie beside those of the tile's other rows -- consecutive entries of the scatter are consecutive
//           samples of a ray, i.e. consecutive rows -- so a 128-byte line serves four entries of a sweep.  With row-major
//           rows each of the three sweeps fetched the row's line again: 1.4 GB of counter traffic for that kernel, 273 us
//           against 229 (grd_dx8).
constexpr int ACT_FEAT = 0, ACT_LD = 32;
constexpr int GRD_GO = 0, GRD_DFEAT = 16, GRD_DX = 48, GRD_LD = 128;
static_assert(ACT_FEAT % 16 == 0 && ACT_LD % 16 == 0, "fragment blocks are 16 columns");
static_assert(GRD_DFEAT % 16 == 0 && GRD_DX % 16 == 0 && GRD_LD % 16 == 0 && GRD_LD - GRD_DX == 80, "fragment blocks are 16 columns");
__host__ __device__ inline size_t frag_off(size_t row, int col, int ld) {
  return (row >> 4) * (size_t)(16 * ld) + (size_t)((col >> 4) * 256 + ((((col >> 2) & 3) * 16 + (int)(row & 15)) << 2) + (col & 3));
}
// where lane (s, g) of the wave that owns tile `tile` puts its float4 of block column COL (a multiple of 16): base + COL * 16
__device__ __forceinline__ float* frag_lane_base(float* buf, size_t tile, int ld, int s, int g) {
  return buf + tile * (size_t)(16 * ld) + ((g * 16 + s) << 2);
}
__device__ __forceinline__ const float* grd_dx_row(const float* grd, size_t row) {
  return grd + (row >> 4) * (size_t)(16 * GRD_LD) + GRD_DX * 16 + (row & 15) * (GRD_LD - GRD_DX);
}
// (groups order) the eight channels 8 grp .. 8 grp + 7 of plane p of row `row`: 32 bytes, 16-byte aligned
__device__ __forceinline__ const float* grd_dx8(const float* grd, size_t row, int p, int grp) {
  return grd + (row >> 4) * (size_t)(16 * GRD_LD) + GRD_DX * 16 + (((p * 3 + grp) * 16 + (int)(row & 15)) << 3);
}

struct DField {
  const float* dplane[3]; const float* dline[3];
  const float* aplane[3]; const float* aline[3];
  const float* mlp;
  const uint4* mlpb;
  const float* aplane2[3]; const float* aline2[3];     // dense [H][W][24] / [L][24] (k_shade3)
  const uint4* mlpw;                                   // w32 image
  const uint4* mlpwt;                                  // transposed w32 image (backward)
  int pw[3], ph[3], ll[3];
  const float* alpha_vol; int ax, ay, az;
  float m_lo[3], m_inv[3];     // alpha-mask aabb: lo and 2/size   (tensorBase.py:57-58)
  float lo[3], inv[3];         // field aabb: lo and 2/size        (tensorBase.py:342-345)
  float hi[3];                 // field aabb: hi (lattice of the alpha-mask rebuild)
  float density_shift, distance_scale, weight_thres;
  float term_T;                // early termination: skip density gathers once transmittance < term_T (0 = off)
  const float* basis; const float* w1; const float* b1; const float* w2; const float* b2;
  const float* w3; const float* b3;
  int fea_pe, view_pe, fc;     // MLPRender_Fea_late_view configuration (0 / 0 / 128: the fast kernels; else lrf_generic.inl)
  float* dump;                 // test hook: s_memtime totals of k_shade3<TIMED> (lrf_debug_set_dump), else null
  float* rdir;                 // k_march -> k_shade3: per ray (d / |d|, |d|), or null
  const int* perm;             // ray sorting (LRF_FLAG_SORT_RAYS): slot -> the caller's ray index for everything indexed by the
};                             // caller (rgb, depth, weights, g_rgb, g_depth, g_rays); null = identity.  `rays` is then the sorted copy

// ------------------------------------------------------------------ geometry
// utils/ray_utils.py:9-12
__device__ __forceinline__ void contract3(float& x, float& y, float& z) {
  float m = fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z));
  m = fmaxf(m, 1e-6f);
  if (m > 1.0f) {
    const float s = (2.0f * m - 1.0f) / (m * m);
    x *= s; y *= s; z *= s;
  }
}

// ATen grid_sampler_unnormalize(align_corners=True) + clip_coordinates (border padding):
// returns the integer base tap and the fractional weight; the +1 tap is index-clamped
// (its weight is 0 whenever it would fall outside).
__device__ __forceinline__ void tap1d(float u, int size, int& i0, int& i1, float& t) {
  float ix = ((u + 1.0f) * 0.5f) * (float)(size - 1);
  ix = fminf(fmaxf(ix, 0.0f), (float)(size - 1));
  const float f0 = floorf(ix);
  t = ix - f0;
  i0 = (int)f0;
  i1 = min(i0 + 1, size - 1);
}

// The interpolation taps of a sample along the three axes, computed once: plane p samples axes MAT0[p] / MAT1[p] and its
// line axis VEC[p], and make_layout gives every plane / line the grid size of the axis it lies along
// (pw[p] = grid[MAT0[p]], ph[p] = grid[MAT1[p]], ll[p] = grid[VEC[p]]), so the nine tap1d calls of a three-plane lookup
// are three distinct ones.  Same arithmetic, same values.
struct AxisTaps { int i0[3], i1[3]; float t[3]; };
__device__ __forceinline__ AxisTaps axis_taps(const int pw0, const int ph0, const int ll0, const float u[3]) {
  AxisTaps a;
  tap1d(u[0], pw0, a.i0[0], a.i1[0], a.t[0]);        // grid[0] = pw[0]
  tap1d(u[1], ph0, a.i0[1], a.i1[1], a.t[1]);        // grid[1] = ph[0]
  tap1d(u[2], ll0, a.i0[2], a.i1[2], a.t[2]);        // grid[2] = ll[0]
  return a;
}

// tensorBase.py:495-499 (torch softplus: beta=1, threshold=20)
__device__ __forceinline__ float feature2density(float f, float shift, bool relu) {
  if (relu) return fmaxf(f, 0.0f);
  const float y = f + shift;
  return y > 20.0f ? y : log1pf(expf(y));
}

// F.grid_sample 3-D, zeros padding, align_corners=True (tensorBase.py:51-55)
__device__ __forceinline__ float alpha_mask_sample(const DField& f, float x, float y, float z) {
  const float ux = (x - f.m_lo[0]) * f.m_inv[0] - 1.0f;
  const float uy = (y - f.m_lo[1]) * f.m_inv[1] - 1.0f;
  const float uz = (z - f.m_lo[2]) * f.m_inv[2] - 1.0f;
  const float ix = ((ux + 1.0f) * 0.5f) * (float)(f.ax - 1);
  const float iy = ((uy + 1.0f) * 0.5f) * (float)(f.ay - 1);
  const float iz = ((uz + 1.0f) * 0.5f) * (float)(f.az - 1);
  const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
  const float tx = ix - fx, ty = iy - fy, tz = iz - fz;
  const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
  float out = 0.0f;
#pragma unroll
  for (int dz = 0; dz < 2; ++dz)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int xi = x0 + dx, yi = y0 + dy, zi = z0 + dz;
        if (xi >= 0 && xi < f.ax && yi >= 0 && yi < f.ay && zi >= 0 && zi < f.az) {
          const float w = (dx ? tx : 1.0f - tx) * (dy ? ty : 1.0f - ty) * (dz ? tz : 1.0f - tz);
          out += f.alpha_vol[((size_t)zi * f.ay + yi) * f.ax + xi] * w;
        }
      }
  return out;
}

// Position of sample with distance zk on ray (o, dhat): tensorBase.py:438-440, then
// normalize_coord tensorBase.py:342-345.  Returns contracted position in (x,y,z) and
// the normalised coordinate in u[3].
__device__ __forceinline__ void sample_point(const DField& f, const float o[3], const float dh[3],
                                             float zk, float x[3], float u[3]) {
  x[0] = o[0] + dh[0] * zk; x[1] = o[1] + dh[1] * zk; x[2] = o[2] + dh[2] * zk;
  contract3(x[0], x[1], x[2]);
#pragma unroll
  for (int a = 0; a < 3; ++a) u[a] = (x[a] - f.lo[a]) * f.inv[a] - 1.0f;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }

// sum_p sum_c bilerp(density_plane_p,c) * lerp(density_line_p,c)   (tensoRF.py:112-151)
__device__ __forceinline__ float density_feature(const DField& f, const float u[3]) {
  float feat = 0.0f;
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    int x0, x1, y0, y1, l0, l1; float tx, ty, tl;
    tap1d(u[MAT0[p]], f.pw[p], x0, x1, tx);
    tap1d(u[MAT1[p]], f.ph[p], y0, y1, ty);
    tap1d(u[VEC[p]],  f.ll[p], l0, l1, tl);
    const float* pl = f.dplane[p];
    const float* q00 = pl + ((size_t)y0 * f.pw[p] + x0) * LRF_CD;
    const float* q10 = pl + ((size_t)y0 * f.pw[p] + x1) * LRF_CD;
    const float* q01 = pl + ((size_t)y1 * f.pw[p] + x0) * LRF_CD;
    const float* q11 = pl + ((size_t)y1 * f.pw[p] + x1) * LRF_CD;
    const float* r0 = f.dline[p] + (size_t)l0 * LRF_CD;
    const float* r1 = f.dline[p] + (size_t)l1 * LRF_CD;
    const float w00 = (1.0f - tx) * (1.0f - ty), w10 = tx * (1.0f - ty);
    const float w01 = (1.0f - tx) * ty,          w11 = tx * ty;
    const float wl0 = 1.0f - tl, wl1 = tl;
    float sp = 0.0f;
#pragma unroll
    for (int h = 0; h < LRF_CD / 4; ++h) {
      const float4 a = ld4(q00 + 4 * h), b = ld4(q10 + 4 * h), c = ld4(q01 + 4 * h), d = ld4(q11 + 4 * h);
      const float4 e = ld4(r0 + 4 * h), g = ld4(r1 + 4 * h);
      sp += (a.x * w00 + b.x * w10 + c.x * w01 + d.x * w11) * (e.x * wl0 + g.x * wl1);
      sp += (a.y * w00 + b.y * w10 + c.y * w01 + d.y * w11) * (e.y * wl0 + g.y * wl1);
      sp += (a.z * w00 + b.z * w10 + c.z * w01 + d.z * w11) * (e.z * wl0 + g.z * wl1);
      sp += (a.w * w00 + b.w * w10 + c.w * w01 + d.w * w11) * (e.w * wl0 + g.w * wl1);
    }
    feat += sp;
  }
  return feat;
}

// ---- gathers with 32-bit byte offsets ---------------------------------------------------------
// `base` is wave-uniform (an SGPR pair), the byte offset a 32-bit VGPR: the load becomes
// global_load_dwordx4 v, v_off, s[base:base+1] and the address arithmetic stays 32-bit (the
// size_t indexing of density_feature / gather_app6_plane costs ~100 64-bit VALU ops per 16-sample
// tile).  Planes are < 4 GB: 640 x 640 texels x 128 B = 52 MB.
__device__ __forceinline__ float4 ld4b(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}

// density_feature with 32-bit offsets (same arithmetic, same order)
__device__ __forceinline__ float density_feature32(const DField& f, const float u[3]) {
  float feat = 0.0f;
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    int x0, x1, y0, y1, l0, l1; float tx, ty, tl;
    tap1d(u[MAT0[p]], f.pw[p], x0, x1, tx);
    tap1d(u[MAT1[p]], f.ph[p], y0, y1, ty);
    tap1d(u[VEC[p]],  f.ll[p], l0, l1, tl);
    const unsigned row0 = (unsigned)y0 * (unsigned)f.pw[p], row1 = (unsigned)y1 * (unsigned)f.pw[p];
    const unsigned o00 = (row0 + x0) * (LRF_CD * 4u), o10 = (row0 + x1) * (LRF_CD * 4u);
    const unsigned o01 = (row1 + x0) * (LRF_CD * 4u), o11 = (row1 + x1) * (LRF_CD * 4u);
    const unsigned q0 = (unsigned)l0 * (LRF_CD * 4u), q1 = (unsigned)l1 * (LRF_CD * 4u);
    const float w00 = (1.0f - tx) * (1.0f - ty), w10 = tx * (1.0f - ty);
    const float w01 = (1.0f - tx) * ty,          w11 = tx * ty;
    const float wl0 = 1.0f - tl, wl1 = tl;
    float sp = 0.0f;
#pragma unroll
    for (int h = 0; h < LRF_CD / 4; ++h) {
      const float4 a = ld4b(f.dplane[p], o00 + 16 * h), b = ld4b(f.dplane[p], o10 + 16 * h);
      const float4 c = ld4b(f.dplane[p], o01 + 16 * h), d = ld4b(f.dplane[p], o11 + 16 * h);
      const float4 e = ld4b(f.dline[p], q0 + 16 * h), g = ld4b(f.dline[p], q1 + 16 * h);
      sp += (a.x * w00 + b.x * w10 + c.x * w01 + d.x * w11) * (e.x * wl0 + g.x * wl1);
      sp += (a.y * w00 + b.y * w10 + c.y * w01 + d.y * w11) * (e.y * wl0 + g.y * wl1);
      sp += (a.z * w00 + b.z * w10 + c.z * w01 + d.z * w11) * (e.z * wl0 + g.z * wl1);
      sp += (a.w * w00 + b.w * w10 + c.w * w01 + d.w * w11) * (e.w * wl0 + g.w * wl1);
    }
    feat += sp;
  }
  return feat;
}

// density_feature for the march: 32-bit saddr gathers of the four plane taps, the two line taps from LDS when
// the caller staged the lines there (LDSL: s_line[p] = the [L][8] line of plane p; 24 texture-path loads per sample
// instead of 36), and the 8-channel contraction on packed fp32 pairs (v_pk_fma_f32: even / odd channel sums,
// added at the end -- the summation order differs from density_feature in the last bits only).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool LDSL>
__device__ __forceinline__ float density_feature_m(const DField& f, const float u[3], const float* const s_line[3]) {
  f32x2 acc = {0.0f, 0.0f};
  const AxisTaps at = axis_taps(f.pw[0], f.ph[0], f.ll[0], u);
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const int x0 = at.i0[MAT0[p]], x1 = at.i1[MAT0[p]], y0 = at.i0[MAT1[p]], y1 = at.i1[MAT1[p]];
    const int l0 = at.i0[VEC[p]], l1 = at.i1[VEC[p]];
    const float tx = at.t[MAT0[p]], ty = at.t[MAT1[p]], tl = at.t[VEC[p]];
    const unsigned row0 = (unsigned)y0 * (unsigned)f.pw[p], row1 = (unsigned)y1 * (unsigned)f.pw[p];
    const unsigned o00 = (row0 + x0) * (LRF_CD * 4u), o10 = (row0 + x1) * (LRF_CD * 4u);
    const unsigned o01 = (row1 + x0) * (LRF_CD * 4u), o11 = (row1 + x1) * (LRF_CD * 4u);
    const f32x2 w00 = {(1.0f - tx) * (1.0f - ty), (1.0f - tx) * (1.0f - ty)}, w10 = {tx * (1.0f - ty), tx * (1.0f - ty)};
    const f32x2 w01 = {(1.0f - tx) * ty, (1.0f - tx) * ty}, w11 = {tx * ty, tx * ty};
    const f32x2 wl0 = {1.0f - tl, 1.0f - tl}, wl1 = {tl, tl};
#pragma unroll
    for (int h = 0; h < LRF_CD / 4; ++h) {
      const float4 a = ld4b(f.dplane[p], o00 + 16 * h), b = ld4b(f.dplane[p], o10 + 16 * h);
      const float4 c = ld4b(f.dplane[p], o01 + 16 * h), d = ld4b(f.dplane[p], o11 + 16 * h);
      float4 e, g;
      if (LDSL) {
        e = *reinterpret_cast<const float4*>(s_line[p] + l0 * LRF_CD + 4 * h);
        g = *reinterpret_cast<const float4*>(s_line[p] + l1 * LRF_CD + 4 * h);
      } else {
        e = ld4b(f.dline[p], (unsigned)l0 * (LRF_CD * 4u) + 16 * h);
        g = ld4b(f.dline[p], (unsigned)l1 * (LRF_CD * 4u) + 16 * h);
      }
      const f32x2 a0 = {a.x, a.y}, a1 = {a.z, a.w}, b0 = {b.x, b.y}, b1 = {b.z, b.w};
      const f32x2 c0 = {c.x, c.y}, c1 = {c.z, c.w}, d0 = {d.x, d.y}, d1 = {d.z, d.w};
      const f32x2 e0 = {e.x, e.y}, e1 = {e.z, e.w}, g0 = {g.x, g.y}, g1 = {g.z, g.w};
      acc += (a0 * w00 + b0 * w10 + c0 * w01 + d0 * w11) * (e0 * wl0 + g0 * wl1);
      acc += (a1 * w00 + b1 * w10 + c1 * w01 + d1 * w11) * (e1 * wl0 + g1 * wl1);
    }
  }
  return acc.x + acc.y;
}

// Inclusive product scan across the 64 lanes of a wave; returns the exclusive product in
// `excl` and the wave total in `total`.
__device__ __forceinline__ void wave_scan_prod(float v, int lane, float& excl, float& total) {
  float p = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float t = __shfl_up(p, d, 64);
    if (lane >= d) p *= t;
  }
  excl = __shfl_up(p, 1, 64);
  if (lane == 0) excl = 1.0f;
  total = __shfl(p, 63, 64);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

}  // namespace lrf
