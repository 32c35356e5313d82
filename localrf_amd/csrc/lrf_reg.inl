// density_L1 regulariser (tensoRF.py:83-92), the one regulariser that is on by
// default (opt.py:111: L1_weight 1e-2) while rf_iter < n_iters_reg.  The reference materialises
// the 8-channel outer product of every plane with its line (8 x g^3 floats per plane: 864 MB at
// 300^3) before summing; here each lattice value is formed in registers.
//
//   feat[i] = sum_p sum_c plane_p[c, i / L_p] * line_p[c, i % L_p]          i in [0, g0 g1 g2)
//   out     = mean_i sqrt(max(feature2density(feat[i]), 1e-5))
//
// NOTE the index arithmetic is the reference's: the three planes flatten the lattice in three
// different orders (plane-major, its own line fastest) and are added element by element in those
// orders -- reproduced as is, not "fixed".
#pragma once

namespace lrf {

constexpr int L1_TPB = 256;
constexpr int L1_QCHUNK = 256;       // plane texels per workgroup in the line-gradient pass, at most
constexpr int L1_QCHUNK_MIN = 8;
// plane texels per workgroup of the line-gradient pass: about 512 workgroups per plane whatever the grid (a fixed 256 left a
// 64^3 field -- 4096 texels per plane -- with 16 workgroups of 64 busy lanes: 30-60 us per plane for 0.26 MFLOP, a tenth of
// the captured iteration of the regularised phase)
__host__ __device__ inline int l1_qchunk(int hw) { const int q = (hw + 511) / 512; return q < L1_QCHUNK_MIN ? L1_QCHUNK_MIN : (q > L1_QCHUNK ? L1_QCHUNK : q); }

struct L1Geo {
  const float* plane[3]; const float* line[3];
  int hw[3], ll[3];
  long long n;
};

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float s = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return s;
}

// dfeat[i] = d sqrt(max(sig,1e-5)) / d feat[i]; partial[block] = sum of the block's values
__global__ __launch_bounds__(L1_TPB) void k_l1_fwd(L1Geo G, float shift, int relu, float* __restrict__ dfeat,
                                                   float* __restrict__ partial) {
  __shared__ float red[4];
  float acc = 0.0f;
  for (long long i = (long long)blockIdx.x * L1_TPB + threadIdx.x; i < G.n; i += (long long)gridDim.x * L1_TPB) {
    float feat = 0.0f;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int q = (int)(i / G.ll[p]), r = (int)(i - (long long)q * G.ll[p]);
      float s = 0.0f;
#pragma unroll
      for (int c = 0; c < LRF_CD; ++c) s += G.plane[p][(size_t)c * G.hw[p] + q] * G.line[p][c * G.ll[p] + r];
      feat += s;
    }
    float sig, dsig;
    if (relu) { sig = fmaxf(feat, 0.0f); dsig = feat > 0.0f ? 1.0f : 0.0f; }
    else {
      const float x = feat + shift;
      sig = x > 20.0f ? x : log1pf(expf(x));                 // F.softplus, threshold 20
      dsig = x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x));
    }
    const float cl = fmaxf(sig, 1e-5f);
    const float y = sqrtf(cl);
    acc += y;
    dfeat[i] = sig >= 1e-5f ? 0.5f / y * dsig : 0.0f;        // clamp(min) passes the gradient where sig >= min
  }
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(L1_TPB) void k_l1_mean(const float* __restrict__ partial, int nb, long long n,
                                                    float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.0f;
  for (int i = threadIdx.x; i < nb; i += L1_TPB) acc += partial[i];
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) out[0] = s / (float)n;
}

// The three planes' passes are one launch each (blockIdx -> plane through the block offsets in L1Blk).
struct L1Blk { int first[4]; int nchunk[3]; int qc[3]; size_t lpart_off[3]; };
__device__ __forceinline__ int l1_plane_of(const int first[4], int& blk) {
  const int p = (int)blockIdx.x >= first[2] ? 2 : ((int)blockIdx.x >= first[1] ? 1 : 0);
  blk = (int)blockIdx.x - first[p];
  return p;
}
struct L1Out { float* plane[3]; float* line[3]; };

// g_plane_p[c, q] = scale * sum_r dfeat[q L + r] line_p[c, r]: one wavefront per texel q
// (acc: add to what `out` holds -- lrf_density_l1_bwd_acc, every element has one writer -- instead of storing)
__global__ __launch_bounds__(L1_TPB) void k_l1_bwd_plane(L1Geo G, L1Blk B, const float* __restrict__ dfeat,
                                                         const float* __restrict__ g_out, L1Out out, int acc_out) {
  int blk;
  const int p = l1_plane_of(B.first, blk);
  const int lane = threadIdx.x & 63;
  const int q = blk * (L1_TPB / 64) + (threadIdx.x >> 6);
  if (q >= G.hw[p]) return;
  const int L = G.ll[p];
  float acc[LRF_CD];
#pragma unroll
  for (int c = 0; c < LRF_CD; ++c) acc[c] = 0.0f;
  for (int r = lane; r < L; r += 64) {
    const float v = dfeat[(size_t)q * L + r];
#pragma unroll
    for (int c = 0; c < LRF_CD; ++c) acc[c] += v * G.line[p][c * L + r];
  }
  const float scale = g_out[0] / (float)G.n;
#pragma unroll
  for (int c = 0; c < LRF_CD; ++c) {
    const float s = wave_sum(acc[c]);
    if (lane == 0) {
      float* o = out.plane[p] + (size_t)c * G.hw[p] + q;
      *o = acc_out ? *o + s * scale : s * scale;
    }
  }
}

// lpart_p[chunk][c][r] = sum over the chunk's texels q of dfeat[q L + r] plane_p[c, q]
__global__ __launch_bounds__(L1_TPB) void k_l1_bwd_line(L1Geo G, L1Blk B, const float* __restrict__ dfeat,
                                                        float* __restrict__ lpart) {
  int blk;
  const int p = l1_plane_of(B.first, blk);
  const int L = G.ll[p];
  const int q0 = blk * B.qc[p], q1 = min(q0 + B.qc[p], G.hw[p]);
  float* lp = lpart + B.lpart_off[p];
  for (int r = threadIdx.x; r < L; r += L1_TPB) {
    float acc[LRF_CD];
#pragma unroll
    for (int c = 0; c < LRF_CD; ++c) acc[c] = 0.0f;
    for (int q = q0; q < q1; ++q) {
      const float v = dfeat[(size_t)q * L + r];
#pragma unroll
      for (int c = 0; c < LRF_CD; ++c) acc[c] += v * G.plane[p][(size_t)c * G.hw[p] + q];   // wave-uniform operand
    }
#pragma unroll
    for (int c = 0; c < LRF_CD; ++c) lp[((size_t)blk * LRF_CD + c) * L + r] = acc[c];
  }
}

// g_line_p[c, r] = scale * sum over the chunks of lpart_p: 16 outputs per workgroup, 16 threads per output, thread j adds
// the chunks k = j (mod 16) in order and the 16 partial sums are added in order -- fixed order: deterministic.  (One thread per
// output walking all ~512 chunks: 119 us at 64^3, where the whole launch has 512 outputs.)
constexpr int L1_RED_OUT = 16, L1_RED_GRP = L1_TPB / L1_RED_OUT;
__global__ __launch_bounds__(L1_TPB) void k_l1_bwd_line_reduce(L1Geo G, L1Blk B, const float* __restrict__ lpart,
                                                               const float* __restrict__ g_out, L1Out out, int acc_out) {
  __shared__ float s_part[L1_RED_GRP][L1_RED_OUT];
  int blk;
  const int p = l1_plane_of(B.first, blk);
  const int L = G.ll[p], n_out = LRF_CD * L;
  const int o = threadIdx.x % L1_RED_OUT, j = threadIdx.x / L1_RED_OUT;
  const int i = blk * L1_RED_OUT + o;                        // (c, r)
  const float* lp = lpart + B.lpart_off[p];
  float acc = 0.0f;
  if (i < n_out)
    for (int k = j; k < B.nchunk[p]; k += L1_RED_GRP) acc += lp[(size_t)k * n_out + i];
  s_part[j][o] = acc;
  __syncthreads();
  if (j == 0 && i < n_out) {
    float t = 0.0f;
#pragma unroll
    for (int q = 0; q < L1_RED_GRP; ++q) t += s_part[q][o];
    const float r = t * (g_out[0] / (float)G.n);
    out.line[p][i] = acc_out ? out.line[p][i] + r : r;
  }
}

static int l1_geo(const float* const plane[3], const float* const line[3], const int32_t hw[3], const int32_t ll[3],
                  L1Geo& G) {
  long long n = -1;
  for (int p = 0; p < 3; ++p) {
    if (!plane[p] || !line[p] || hw[p] <= 0 || ll[p] <= 0) return 1;
    G.plane[p] = plane[p]; G.line[p] = line[p]; G.hw[p] = hw[p]; G.ll[p] = ll[p];
    const long long np = (long long)hw[p] * ll[p];
    if (n >= 0 && np != n) return 1;                         // every plane x line spans the same lattice
    n = np;
  }
  G.n = n;
  return 0;
}
static int l1_blocks(long long n) {
  const long long want = (n + L1_TPB - 1) / L1_TPB, cap = (long long)device_cus() * 16;
  return (int)(want < cap ? want : cap);
}
static int max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
static size_t l1_lpart_floats(const int32_t hw[3], const int32_t ll[3]) {          // the three planes' partial blocks
  size_t n = 0;
  for (int p = 0; p < 3; ++p) n += (size_t)((hw[p] + l1_qchunk(hw[p]) - 1) / l1_qchunk(hw[p])) * LRF_CD * (size_t)ll[p];
  return n;
}

}  // namespace lrf

extern "C" size_t lrf_density_l1_workspace(const int32_t hw[3], const int32_t ll[3]) {
  using namespace lrf;
  const long long n = (long long)hw[0] * ll[0];
  return sizeof(float) * ((size_t)n + (size_t)device_cus() * 16 + l1_lpart_floats(hw, ll)) + 1024;
}

extern "C" int lrf_density_l1_fwd(const float* const plane[3], const float* const line[3], const int32_t hw[3],
                                  const int32_t ll[3], float density_shift, int32_t relu, void* workspace,
                                  float* out, void* stream) {
  using namespace lrf;
  L1Geo G;
  if (!plane || !line || !hw || !ll || !workspace || !out || l1_geo(plane, line, hw, ll, G))
    return set_err("lrf_density_l1_fwd: null argument or planes/lines that do not span one lattice");
  float* dfeat = static_cast<float*>(workspace);
  float* partial = dfeat + G.n;
  const int nb = l1_blocks(G.n);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(k_l1_fwd, dim3(nb), dim3(L1_TPB), 0, st, G, density_shift, relu, dfeat, partial);
  hipLaunchKernelGGL(k_l1_mean, dim3(1), dim3(L1_TPB), 0, st, partial, nb, G.n, out);
  LRF_HIP(hipGetLastError());
  return 0;
}

static int density_l1_bwd_impl(const float* const plane[3], const float* const line[3], const int32_t hw[3],
                               const int32_t ll[3], const void* workspace, const float* g_out,
                               float* const g_plane[3], float* const g_line[3], void* stream, int acc_out) {
  using namespace lrf;
  L1Geo G;
  if (!plane || !line || !hw || !ll || !workspace || !g_out || !g_plane || !g_line || l1_geo(plane, line, hw, ll, G))
    return set_err("lrf_density_l1_bwd: null argument or planes/lines that do not span one lattice");
  const float* dfeat = static_cast<const float*>(workspace);
  float* lpart = const_cast<float*>(dfeat) + G.n + (size_t)device_cus() * 16;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  L1Out out;
  L1Blk Bp, Bl, Br;                                          // block offsets of the plane / line / reduce launches
  Bp.first[0] = Bl.first[0] = Br.first[0] = 0;
  size_t off = 0;
  for (int p = 0; p < 3; ++p) {
    if (!g_plane[p] || !g_line[p]) return set_err("lrf_density_l1_bwd: null gradient pointer");
    out.plane[p] = g_plane[p]; out.line[p] = g_line[p];
    const int qc = l1_qchunk(G.hw[p]), nchunk = (G.hw[p] + qc - 1) / qc;
    Bp.first[p + 1] = Bp.first[p] + (G.hw[p] + L1_TPB / 64 - 1) / (L1_TPB / 64);
    Bl.first[p + 1] = Bl.first[p] + nchunk;
    Br.first[p + 1] = Br.first[p] + (LRF_CD * G.ll[p] + L1_RED_OUT - 1) / L1_RED_OUT;
    Bp.qc[p] = Bl.qc[p] = Br.qc[p] = qc;
    Bp.nchunk[p] = Bl.nchunk[p] = Br.nchunk[p] = nchunk;
    Bp.lpart_off[p] = Bl.lpart_off[p] = Br.lpart_off[p] = off;
    off += (size_t)nchunk * LRF_CD * (size_t)G.ll[p];
  }
  hipLaunchKernelGGL(k_l1_bwd_plane, dim3(Bp.first[3]), dim3(L1_TPB), 0, st, G, Bp, dfeat, g_out, out, acc_out);
  hipLaunchKernelGGL(k_l1_bwd_line, dim3(Bl.first[3]), dim3(L1_TPB), 0, st, G, Bl, dfeat, lpart);
  hipLaunchKernelGGL(k_l1_bwd_line_reduce, dim3(Br.first[3]), dim3(L1_TPB), 0, st, G, Br, lpart, g_out, out, acc_out);
  LRF_HIP(hipGetLastError());
  return 0;
}
extern "C" int lrf_density_l1_bwd(const float* const plane[3], const float* const line[3], const int32_t hw[3],
                                  const int32_t ll[3], const void* workspace, const float* g_out,
                                  float* const g_plane[3], float* const g_line[3], void* stream) {
  return density_l1_bwd_impl(plane, line, hw, ll, workspace, g_out, g_plane, g_line, stream, 0);
}
extern "C" int lrf_density_l1_bwd_acc(const float* const plane[3], const float* const line[3], const int32_t hw[3],
                                      const int32_t ll[3], const void* workspace, const float* g_out,
                                      float* const g_plane[3], float* const g_line[3], void* stream) {
  return density_l1_bwd_impl(plane, line, hw, ll, workspace, g_out, g_plane, g_line, stream, 1);
}

// ---------------------------------------------------------------------------------------------
// TV regulariser (utils/utils.py:293-309 applied as tensoRF.py:94-110): for every tensor x [C,H,W]
// (lines: W = 1)  tv = 2 w (sum (x[y]-x[y-1])^2 / (C (H-1) W) + sum (x[,x]-x[,x-1])^2 / (C H (W-1))),
// loss = sum over tensors of scale * tv (1e-2 for planes, 1e-3 for lines).  Off by default in the
// reference (opt.py:112-113); one launch each way for up to LRF_TV_MAX tensors.
namespace lrf {

constexpr int TV_CHUNK = 4096;
struct TvTable { LrfTvSeg s[LRF_TV_MAX]; int first_block[LRF_TV_MAX + 1]; int count; };

__device__ __forceinline__ int tv_seg_of(const TvTable& tab) {
  int k = 0;
  while (k + 1 < tab.count && (int)blockIdx.x >= tab.first_block[k + 1]) ++k;
  return k;
}

// partial[block] = (sum of squared differences along H, along W) of this block's elements
__global__ __launch_bounds__(L1_TPB) void k_tv_fwd(TvTable tab, float2* __restrict__ partial) {
  __shared__ float red[4];
  const int k = tv_seg_of(tab);
  const LrfTvSeg sg = tab.s[k];
  const long long n = (long long)sg.C * sg.H * sg.W;
  const long long base = (long long)((int)blockIdx.x - tab.first_block[k]) * TV_CHUNK;
  float sh = 0.0f, sw = 0.0f;
  for (int it = 0; it < TV_CHUNK / L1_TPB; ++it) {
    const long long i = base + it * L1_TPB + threadIdx.x;
    if (i < n) {
      const int x = (int)(i % sg.W), y = (int)((i / sg.W) % sg.H);
      const float v = sg.x[i];
      if (y > 0) { const float d = v - sg.x[i - sg.W]; sh += d * d; }
      if (x > 0) { const float d = v - sg.x[i - 1]; sw += d * d; }
    }
  }
  const float a = block_sum_256(sh, red), b = block_sum_256(sw, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = make_float2(a, b);
}

__global__ void k_tv_final(TvTable tab, const float2* __restrict__ partial, float weight, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float total = 0.0f;
  for (int k = 0; k < tab.count; ++k) {                         // fixed order: deterministic
    const LrfTvSeg sg = tab.s[k];
    float sh = 0.0f, sw = 0.0f;
    for (int b = tab.first_block[k]; b < tab.first_block[k + 1]; ++b) { sh += partial[b].x; sw += partial[b].y; }
    float tv = 0.0f;
    if (sg.H > 1) tv += sh / ((float)sg.C * (float)(sg.H - 1) * (float)sg.W);
    if (sg.W > 1) tv += sw / ((float)sg.C * (float)sg.H * (float)(sg.W - 1));
    total += weight * 2.0f * tv * sg.scale;
  }
  out[0] = total;
}

__global__ __launch_bounds__(L1_TPB) void k_tv_bwd(TvTable tab, float weight, const float* __restrict__ g_out) {
  const int k = tv_seg_of(tab);
  const LrfTvSeg sg = tab.s[k];
  const long long n = (long long)sg.C * sg.H * sg.W;
  const long long base = (long long)((int)blockIdx.x - tab.first_block[k]) * TV_CHUNK;
  const float ch = sg.H > 1 ? 2.0f / ((float)sg.C * (float)(sg.H - 1) * (float)sg.W) : 0.0f;
  const float cw = sg.W > 1 ? 2.0f / ((float)sg.C * (float)sg.H * (float)(sg.W - 1)) : 0.0f;
  const float s = g_out[0] * weight * 2.0f * sg.scale;
  for (int it = 0; it < TV_CHUNK / L1_TPB; ++it) {
    const long long i = base + it * L1_TPB + threadIdx.x;
    if (i >= n) break;
    const int x = (int)(i % sg.W), y = (int)((i / sg.W) % sg.H);
    const float v = sg.x[i];
    float gh = 0.0f, gw = 0.0f;
    if (y > 0) gh += v - sg.x[i - sg.W];
    if (y < sg.H - 1) gh -= sg.x[i + sg.W] - v;
    if (x > 0) gw += v - sg.x[i - 1];
    if (x < sg.W - 1) gw -= sg.x[i + 1] - v;
    sg.g[i] = s * (ch * gh + cw * gw);
  }
}

static int tv_table(const LrfTvSeg* segs, int count, TvTable& tab) {
  if (!segs || count <= 0 || count > LRF_TV_MAX) return -1;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    if (!segs[i].x || segs[i].C <= 0 || segs[i].H <= 0 || segs[i].W <= 0) return -1;
    tab.s[i] = segs[i];
    tab.first_block[i] = blocks;
    blocks += (int)(((long long)segs[i].C * segs[i].H * segs[i].W + TV_CHUNK - 1) / TV_CHUNK);
  }
  tab.first_block[count] = blocks;
  tab.count = count;
  return blocks;
}

}  // namespace lrf

extern "C" size_t lrf_tv_workspace(const LrfTvSeg* segs, int32_t count) {
  lrf::TvTable tab;
  const int blocks = lrf::tv_table(segs, count, tab);
  return blocks < 0 ? 0 : sizeof(float) * 2 * (size_t)blocks + 256;
}

extern "C" int lrf_tv_loss_fwd(const LrfTvSeg* segs, int32_t count, float weight, void* workspace, float* out,
                               void* stream) {
  using namespace lrf;
  TvTable tab;
  const int blocks = tv_table(segs, count, tab);
  if (blocks < 0 || !workspace || !out) return set_err("lrf_tv_loss_fwd: null argument or bad tensor table");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(k_tv_fwd, dim3(blocks), dim3(L1_TPB), 0, st, tab, static_cast<float2*>(workspace));
  hipLaunchKernelGGL(k_tv_final, dim3(1), dim3(64), 0, st, tab, static_cast<const float2*>(workspace), weight, out);
  LRF_HIP(hipGetLastError());
  return 0;
}

extern "C" int lrf_tv_loss_bwd(const LrfTvSeg* segs, int32_t count, float weight, const float* g_out, void* stream) {
  using namespace lrf;
  TvTable tab;
  const int blocks = tv_table(segs, count, tab);
  if (blocks < 0 || !g_out) return set_err("lrf_tv_loss_bwd: null argument or bad tensor table");
  for (int i = 0; i < count; ++i) if (!segs[i].g) return set_err("lrf_tv_loss_bwd: null gradient pointer");
  hipLaunchKernelGGL(k_tv_bwd, dim3(blocks), dim3(L1_TPB), 0, reinterpret_cast<hipStream_t>(stream), tab, weight, g_out);
  LRF_HIP(hipGetLastError());
  return 0;
}
