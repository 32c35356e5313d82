/* lrf.h -- C ABI of the MI355X-native localrf render path (liblrf_hip.so).
 *
 * The reference (facebookresearch/localrf) has no FFI layer: its hot path is a chain of
 * ATen ops behind two Python call signatures.  This header is the boundary a maintainer
 * binds instead (ctypes stub in INTEGRATION.md).  Each entry point names the reference
 * lines (relative to /root/reference/localTensoRF) whose work it replaces.
 *
 * Conventions: every pointer is a DEVICE pointer to contiguous fp32 (unless typed
 * otherwise) on the current HIP device; `stream` is a hipStream_t passed as void*;
 * no allocation, no host synchronisation and no ownership transfer inside the library;
 * all calls are asynchronous on `stream` and re-entrant per stream (the test hooks of
 * include/lrf_debug.h are process-wide switches and are NOT part of this contract).
 * Return 0 on success, non-zero on error with a message available from lrf_last_error().
 */
#ifndef LRF_H_
#define LRF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LRF_ABI_VERSION 7      /* 7: lrf_density_l1_bwd_acc, LrfGrads.zero_base / zero_floats; 6: lrf_batch_gather, lrf_loss_combine_*, lrf_adam_step_pack; 5: lrf_scene_fwd takes a scene workspace (fused multi-field launches), LRF_FLAG_PLANE_EVENTS, lrf_render_bwd_wait buckets 3 / 4, lrf_adam_step_dev, lrf_photo_loss_*, lrf_rows_gather*; 4: lrf_z_schedule, training rows = feat + gradient row only (ACT_LD 32), network configuration in LrfParams / LrfField, lrf_workspace_bytes_bwd_cfg; 3: lrf_render_bwd_wait, unknown flag bits rejected */
#define LRF_MAX_S 4096         /* samples per ray accepted by lrf_render_fwd */
#define LRF_MAX_S_TRAIN 2048   /* ... by lrf_render_fwd_train / lrf_render_bwd (16 B of LDS per sample and ray) */

/* Fixed shape of the VM field this build is specialised for (opt.py:117-119,155-157:
 * n_lamb_sigma=[8,8,8], n_lamb_sh=[24,24,24], data_dim_color=27, featureC=128). */
#define LRF_CD 8      /* density components per plane  */
#define LRF_CA 24     /* appearance components per plane */
#define LRF_APP_DIM 27
#define LRF_FEATC 128

/* flags for lrf_render_fwd / lrf_render_bwd */
#define LRF_FLAG_WHITE_BG   1u   /* tensorBase.py:633-634 */
#define LRF_FLAG_RELU_DENS  2u   /* fea2denseAct == "relu" (tensorBase.py:498-499) */
#define LRF_FLAG_MLP_VALU   4u   /* debug engine: colour MLP on the vector ALU, natural-layout weights */
#define LRF_FLAG_MLP_F32    8u   /* colour MLP on exact-fp32 MFMA (16x16x4 f32) instead of the default
                                    split-bf16 (hi+lo, 3-term) chain on v_mfma_f32_32x32x16_bf16 (k_shade3) */
#define LRF_FLAG_ROWS_SAVED 16u  /* lrf_render_bwd only: the workspace was filled by lrf_render_fwd_train */
#define LRF_FLAG_SORT_RAYS  32u  /* render the batch in direction-sorted order (cube face + Morton key of d / |d|, one small launch):
                                    rays that march through the same texels run on the same XCD at the same time.  Results are
                                    per-ray and do not depend on the order (bit-identical); ignored for R > 32768.  The same value
                                    must be passed to lrf_render_fwd_train and the lrf_render_bwd that follows it. */
#define LRF_FLAG_PE_OFF     64u  /* fea_pe > 0 only: zeros in place of the feature encodings (MLPRender_Fea_late_view.forward with
                                  * refine == False, tensorBase.py:118-126) */
#define LRF_FLAG_PLANE_EVENTS 128u /* lrf_render_bwd only (data parallel): the appearance scatter runs as one pass per plane and records an
                                  * event behind planes 0 and 1 (lrf_render_bwd_wait buckets 3 / 4), so that the all-reduce of a plane's
                                  * gradient overlaps the scatter of the next one.  Same gradients. */
#define LRF_FLAG_ALL        255u /* any other bit is an error (a caller built against another ABI version) */

/* Parameters of one TensorVMSplit field as the reference stores them (state-dict layout,
 * models/tensoRF.py:18-50, models/tensorBase.py:97-113).  Plane p is [1,C,H_p,W_p] with
 * W_p = gridSize[matMode[p][0]], H_p = gridSize[matMode[p][1]]; line p is [1,C,L_p,1]
 * with L_p = gridSize[vecMode[p]]. */
typedef struct LrfParams {
  const float* density_plane[3];
  const float* density_line[3];
  const float* app_plane[3];
  const float* app_line[3];
  const float* basis;      /* basis_mat.weight          [27,72]   */
  const float* w1;         /* renderModule.mlp.0.weight [128,27]  */
  const float* b1;         /* renderModule.mlp.0.bias   [128]     */
  const float* w2;         /* renderModule.mlp.2.weight [128,128] */
  const float* b2;         /* renderModule.mlp.2.bias   [128]     */
  const float* w3;         /* renderModule.mlp_view.0.weight [3,131] */
  const float* b3;         /* renderModule.mlp_view.0.bias   [3]     */
  int32_t grid[3];         /* gridSize (x,y,z) */
  /* MLPRender_Fea_late_view configuration (tensorBase.py:97-113).  0 / 0 / 128 (opt.py:148-157, what train.py runs) takes the
   * fast kernels and the shapes above; anything else the generic fp32 engine (csrc/lrf_generic.inl), with
   * w1 [feature_c, 27 (1 + 2 fea_pe)], w2 [feature_c, feature_c], w3 [3, feature_c + 3 (1 + 2 view_pe)].
   * fea_pe, view_pe <= 6, feature_c <= 256; feature_c == 0 is read as 128. */
  int32_t fea_pe, view_pe, feature_c;
} LrfParams;

/* Derived, kernel-friendly image of a field ("layout cache").  Built by lrf_pack_field
 * into caller-owned memory of lrf_cache_bytes() bytes; must be rebuilt whenever a
 * parameter changes (optimizer step, upsample_volume_grid models/tensoRF.py:224-233). */
typedef struct LrfField {
  const void*  cache;      /* device buffer written by lrf_pack_field */
  const float* alpha_vol;  /* AlphaGridMask.alpha_volume [Z,Y,X] or NULL (tensorBase.py:36-58) */
  int32_t alpha_dim[3];    /* X,Y,Z of alpha_vol */
  float   alpha_aabb[6];   /* AlphaGridMask.aabb */
  float   aabb[6];         /* field aabb (min xyz, max xyz) */
  int32_t grid[3];
  float   density_shift;   /* tensorBase.py:497 */
  float   distance_scale;  /* tensorBase.py:610 */
  float   weight_thres;    /* rayMarch_weight_thres, tensorBase.py:622 */
  float   term_T;          /* early termination of the march: once the transmittance entering a 64-sample
                            * chunk is below term_T the remaining density lookups are skipped and those
                            * samples count as empty (the forced last sample takes the rest).  No sample
                            * behind that point can pass weight_thres; sum(w z) moves by <= term_T * z_max.
                            * 0 = off (the reference evaluates every sample, tensorBase.py:600-610). */
  /* natural-layout MLP weights (the parameter tensors themselves): read by the generic engine -- any configuration other
   * than fea_pe = view_pe = 0, feature_c = 128, and the LRF_FLAG_MLP_VALU debug engine of that one */
  const float* basis; const float* w1; const float* b1;
  const float* w2; const float* b2; const float* w3; const float* b3;
  int32_t fea_pe, view_pe, feature_c;   /* as in LrfParams */
} LrfField;

/* Gradients wrt the reference's parameters, in the reference's (state-dict) layout.
 * Accumulated into (+=); the caller zeroes them -- or (ABI 7) names ONE range that holds all of them (the flat gradient
 * buffer of localrf_amd: parameters' gradients and d/d rays) and lrf_render_bwd clears it with the launch that clears its own
 * bins: zero_base (16-byte aligned) / zero_floats (a multiple of 4), NULL / 0 = the caller cleared. */
typedef struct LrfGrads {
  float* density_plane[3];
  float* density_line[3];
  float* app_plane[3];
  float* app_line[3];
  float* basis; float* w1; float* b1; float* w2; float* b2; float* w3; float* b3;
  float* zero_base; int64_t zero_floats;
} LrfGrads;

int         lrf_abi_version(void);
const char* lrf_last_error(void);
char*       lrf_error_slot(void);                   /* internal: the calling thread's 512-byte error text */

/* Bytes of the layout cache for a grid (x,y,z). */
size_t lrf_cache_bytes(const int32_t grid[3]);

/* NCHW params -> channel-last planes/lines + MFMA-fragment-ordered MLP image. */
int lrf_pack_field(const LrfParams* p, void* cache, void* stream);

/* Bytes of scratch lrf_render_fwd / lrf_render_bwd need for R rays x S samples. */
size_t lrf_workspace_bytes(int32_t R, int32_t S);

/* TensorBase.forward (tensorBase.py:567-636) for one field, z schedule supplied
 * (tensorBase.py:419-437 is ray-independent, so the host draws the jitter and keeps RNG
 * parity).  rays [R,6] = (origin, un-normalised direction); z [S].
 * Outputs rgb [R,3], depth [R].  Optional debug outputs (may be NULL): weight_out [R,S]
 * (post floater filter, tensorBase.py:612/620), acc_out [R]. */
int lrf_render_fwd(const LrfField* f, const float* rays, const float* z,
                   int32_t R, int32_t S, uint32_t flags, float floater_thresh,
                   float* rgb, float* depth, float* weight_out, float* acc_out,
                   void* workspace, void* stream);

/* Measurement variant of lrf_render_fwd (bench.py only): brackets each kernel with HIP
 * events on `stream`, SYNCHRONISES, and returns ms_out[6] (host) = {march, shade, finalize,
 * total, 0, 0} plus the number of shaded samples (sum over rays of weight > thres).  The default engine has
 * no separate finalize launch (0). */
int lrf_render_fwd_profile(const LrfField* f, const float* rays, const float* z,
                           int32_t R, int32_t S, uint32_t flags, float floater_thresh,
                           float* rgb, float* depth, void* workspace, void* stream,
                           float* ms_out, int32_t* n_shaded_out);

/* Bytes of scratch lrf_render_bwd needs (worst case: every sample shaded).  _cfg: for a network configuration other than
 * the default one (the generic engine keeps the weight-gradient operands as rows: ~ (4 feature_c + in1 + in_view) floats
 * per shaded sample more -- also for the default configuration when `flags` carries LRF_FLAG_MLP_VALU: the backward then runs the
 * generic fp32 engine too, an exact-fp32 training path); lrf_workspace_bytes_bwd = _cfg(..., 0, 0, 128, 0). */
size_t lrf_workspace_bytes_bwd(int32_t R, int32_t S, const int32_t grid[3]);
size_t lrf_workspace_bytes_bwd_cfg(int32_t R, int32_t S, const int32_t grid[3], int32_t fea_pe, int32_t view_pe, int32_t feature_c, uint32_t flags);

/* Training forward: same outputs as lrf_render_fwd (split-bf16 engine, floater_thresh 0), but the
 * per-sample state the backward needs (density features, shaded-sample lists, per-sample colours,
 * activation rows) is left in `workspace` (lrf_workspace_bytes_bwd bytes) instead of being
 * recomputed by lrf_render_bwd: pass the SAME workspace, field, rays and z to lrf_render_bwd with
 * LRF_FLAG_ROWS_SAVED set.  Memory: a 1.5 GB worst-case reservation at 4096 x 512 (0.6 GB touched when 35 % of the samples are shaded) against 288 GB of HBM. */
int lrf_render_fwd_train(const LrfField* f, const float* rays, const float* z, int32_t R, int32_t S,
                         uint32_t flags, float* rgb, float* depth, void* workspace, void* stream);

/* Backward of lrf_render_fwd (replaces autograd through tensorBase.py:567-636,
 * tensoRF.py:112-196): recomputes the forward (unless LRF_FLAG_ROWS_SAVED), scatters parameter gradients into `g`
 * (reference layout, +=) and writes d(loss)/d(rays) [R,6]. The floater filter is eval-only
 * (train.py:107,139) and not differentiated. `workspace`: lrf_workspace_bytes_bwd bytes. */
int lrf_render_bwd(const LrfField* f, const LrfParams* p, const float* rays, const float* z,
                   int32_t R, int32_t S, uint32_t flags,
                   const float* g_rgb, const float* g_depth,
                   const LrfGrads* g, float* g_rays,
                   void* workspace, void* stream);

/* Data-parallel hand-off (no reference counterpart): makes `stream` wait until one bucket of the gradients
 * of the most recent lrf_render_bwd enqueued on the current device is final, so that a collective over that bucket can
 * start while the rest of the backward still runs.  bucket 0: density planes + lines (the per-ray branch finishes
 * early), 1: colour network (basis, mlp, mlp_view), 2: appearance planes + lines (= everything), 3 / 4: appearance plane 0 / 1
 * alone (behind their own scatter pass when the backward ran with LRF_FLAG_PLANE_EVENTS; otherwise the same point as 2).
 * Error if no lrf_render_bwd ran on this device. */
int lrf_render_bwd_wait(int32_t bucket, void* stream);

/* Debug / parity diagnostics: byte offsets inside the training workspace of {activation rows, gradient rows,
 * rowinfo (row -> ray*S+k or ~0), toff}, the row strides {ACT_LD, GRD_LD} in floats, the byte offset of the ReLU mask
 * bits the training forward saved ([tile][layer 1, 2][lane s + 16 g] dwords: bit 4 t + r = unit 16 t + 4 g + r of the
 * tile's sample s), the byte offset of the slot -> ray permutation of LRF_FLAG_SORT_RAYS (int32 [R]; rowinfo and the feature
 * buffer are indexed by SLOT when the batch was sorted), then the byte offset of the density-feature buffer [R,S] (-inf = sample not evaluated: masked,
 * last, or behind an early termination; overwritten with d(loss)/d(feature) by lrf_render_bwd).  Rows are
 * indexed tile*16 + lane; rowinfo is valid after lrf_render_bwd of the same workspace. */
void lrf_workspace_layout_bwd(int32_t R, int32_t S, const int32_t grid[3], uint64_t out[9]);

/* Pieces of the path exposed on their own (unit parity tests; also used by
 * TensorVMSplit.compute_densityfeature / compute_appfeature / compute_alpha):
 * u [P,3] are normalised coordinates in [-1,1] (tensorBase.py:342-345). */
int lrf_density_feature(const LrfField* f, const float* u, int32_t P, float* sigma_feature, void* stream); /* tensoRF.py:112-151 */
int lrf_app_feature(const LrfField* f, const float* u, int32_t P, float* app_features /*[P,27]*/, void* stream); /* tensoRF.py:153-196 */

/* TensorBase.sample_ray (tensorBase.py:396-417): AABB march named by BASELINE.json's
 * north_star (not on train.py's path).  jitter [R] or NULL.  Outputs pts [R,N,3],
 * t [R,N], inside [R,N] (uint8). */
int lrf_sample_ray_aabb(const float* rays, const float aabb[6], float step_size, float near_, float far_,
                        const float* jitter, int32_t R, int32_t N,
                        float* pts, float* t, uint8_t* inside, void* stream);

/* The sample distances of TensorBase.sample_ray_contracted (tensorBase.py:419-437) for N_samples = 6 h: z [2 h] on the
 * device, the first half linear in [0, 1), the second inverse-depth out to 1e3, + 0.1.  u1, u2 [h]: the two jitter draws
 * of train mode (torch.rand_like, in the reference's order), both NULL in eval mode. */
int lrf_z_schedule(int32_t h, const float* u1, const float* u2, float* z, void* stream);

/* TensorBase.sample_ray_contracted (tensorBase.py:419-443): pts [R,S,3] = contract(o + d z) for a caller-supplied
 * schedule z [S] (the second return value of the reference's method; its third is all-true).  rays_o, rays_d [R,3]. */
int lrf_sample_ray_contracted(const float* rays_o, const float* rays_d, const float* z, int32_t R, int32_t S,
                              float* pts, void* stream);

/* Scene-level ends of the path: LocalTensorfs.forward (local_tensorfs.py:382-499).
 * View of ray r is r / per_view (repeat_interleave at local_tensorfs.py:437); R % per_view == 0.
 *
 * lrf_scene_rays: ids2pixel (local_tensorfs.py:23-29) + get_ray_directions_lean / _360
 * (utils/ray_utils.py:14-37) + cam2rf = cam2world (+) world2rf (local_tensorfs.py:427-431) +
 * get_rays_lean (utils/ray_utils.py:39-54), for n_rf fields in one launch.
 *   ray_ids [R] int64; cam2world [V,3,4]; world2rf [n_rf,3]; focal [1], center [2] device
 *   scalars (NULL when fov360); outputs rays [n_rf,R,6], directions [R,3], ij [R,2] int64. */
int lrf_scene_rays(const int64_t* ray_ids, int32_t R, int32_t per_view, const float* cam2world,
                   const float* world2rf, int32_t n_rf, const float* focal, const float* center,
                   int32_t W, int32_t H, int32_t fov360, float* rays, float* directions, int64_t* ij,
                   void* stream);
/* Gradients autograd derives for the above: g_cam2world [V,3,4]; g_intr [V,3] per-view partial
 * sums of (d focal, d center_x, d center_y); g_world2rf [V,n_rf,3] per-view partial sums.
 * g_directions [R,3] may be NULL. */
int lrf_scene_rays_bwd(const int64_t* ray_ids, int32_t R, int32_t per_view, const float* cam2world,
                       int32_t n_rf, const float* focal, const float* center, int32_t W, int32_t H,
                       int32_t fov360, const float* g_rays, const float* g_directions,
                       float* g_cam2world, float* g_intr, float* g_world2rf, void* stream);
/* lrf_scene_blend: rgbs = clamp(E_v (sum_k w[v,k] rgb_k), 0, 1), depth = sum_k w[v,k] depth_k
 * (local_tensorfs.py:468-474,481-499).  rgb_f [n_rf,R,3], depth_f [n_rf,R], blend_w [V,n_rf],
 * exposure [V,3,3] or NULL; pre [R,3] (blended colour before exposure, kept for the backward
 * pass) may be NULL. */
int lrf_scene_blend(const float* rgb_f, const float* depth_f, const float* blend_w, const float* exposure,
                    int32_t R, int32_t per_view, int32_t n_rf, float* rgbs, float* depth, float* pre,
                    void* stream);
/* g_depth and g_exposure [V,3,3] may be NULL. */
int lrf_scene_blend_bwd(const float* g_rgbs, const float* g_depth, const float* pre, const float* blend_w,
                        const float* exposure, int32_t R, int32_t per_view, int32_t n_rf,
                        float* g_rgb_f, float* g_depth_f, float* g_exposure, void* stream);

/* LocalTensorfs.forward without a tape (local_tensorfs.py:397-499) in one call: lrf_scene_rays, then for every chunk of
 * `chunk` rays (<= 0: all at once) lrf_render_fwd of every active field in the reference's order (:440-474), then
 * lrf_scene_blend.  `fields` is a HOST array of n_rf entries; each field brings its own z schedule (S depends on the
 * field's grid, tensorBase.py:252-262), engine flags and workspace (lrf_workspace_bytes(min(chunk, R), S) bytes; fields
 * may share one, the launches are serialised on `stream`).  Scratch supplied by the caller: rays [n_rf,R,6],
 * rgb_f [n_rf,R,3], depth_f [n_rf,R].  Outputs as lrf_scene_rays / lrf_scene_blend.
 * scene_workspace (may be NULL; lrf_workspace_bytes(min(n_rf, 4) * min(chunk, R), S) bytes): with it, groups of up to four fields
 * of one shape (same grid, S, flags, thresholds; default colour engine) render each chunk in ONE march and ONE colour launch over
 * their field-major rays when chunk % 16 == 0 and floater_thresh == 0 (a ragged last chunk goes field by field) -- the same
 * per-ray arithmetic (depths bit-identical, colours to an ulp); otherwise field by field. */
#define LRF_SCENE_MAX_FIELDS 64
typedef struct LrfSceneField {
  const LrfField* field;
  const float*    z;          /* [S] device */
  int32_t         S;
  uint32_t        flags;      /* LRF_FLAG_* of lrf_render_fwd */
  void*           workspace;
} LrfSceneField;
int lrf_scene_fwd(const int64_t* ray_ids, int32_t R, int32_t per_view, const float* cam2world, const float* world2rf,
                  int32_t n_rf, const float* focal, const float* center, int32_t W, int32_t H, int32_t fov360,
                  const LrfSceneField* fields, float floater_thresh, int32_t chunk,
                  const float* blend_w, const float* exposure,
                  float* rays, float* rgb_f, float* depth_f, float* directions, int64_t* ij,
                  float* rgbs, float* depth, void* scene_workspace, size_t scene_workspace_bytes, void* stream);

/* Optimiser step after the path: torch.optim.Adam with the reference's settings
 * (local_tensorfs.py:88-97,146,245; no weight decay, no amsgrad) over up to LRF_ADAM_MAX tensors in
 * one launch.  p, m (exp_avg), v (exp_avg_sq) are updated in place; step_size = lr / (1 - beta1^t)
 * and bc2_sqrt = sqrt(1 - beta2^t) are evaluated by the caller in double, as torch does. */
#define LRF_ADAM_MAX 64
typedef struct LrfAdamTensor {
  float* p; const float* g; float* m; float* v;
  int64_t n;
  float step_size, bc2_sqrt;
} LrfAdamTensor;
int lrf_adam_step(const LrfAdamTensor* tensors /* host array */, int32_t count, float beta1, float beta2,
                  float eps, void* stream);
/* The same launch with step_size / bc2_sqrt read from DEVICE memory when the kernel runs (dev_scalars [count][2]; the two
 * fields of the host structs are ignored): the launch can be captured in a hipGraph and replayed with the learning rates and
 * bias corrections the host wrote before each replay.  bc2_sqrt <= 0 skips a tensor (torch.optim.Adam skips a parameter
 * whose .grad is None: local_tensorfs.py:229-243 steps the poses of sampled views only). */
int lrf_adam_step_dev(const LrfAdamTensor* tensors /* host array */, int32_t count, const float* dev_scalars, float beta1, float beta2,
                      float eps, void* stream);

/* The optimiser step FUSED with the layout refresh (local_tensorfs.py:146,245 step the field's optimiser, and
 * the next forward re-reads every parameter to rebuild the channel-last cache): the same update as lrf_adam_step /
 * lrf_adam_step_dev (dev_scalars NULL: the host fields of the structs) for the tensors of the table, and behind it `cache` --
 * the layout cache of the field whose parameters `p` names (lrf_cache_bytes, as lrf_pack_field fills it) -- holds the NEW
 * values: the field's twelve plane / line tensors are stepped and written channel-last by one kernel (tensors of `p` that
 * are not in the table are only repacked), the other tensors of the table by the table kernel, then the colour network's
 * images are rebuilt.  Equivalent to lrf_adam_step[_dev] followed by lrf_pack_field. */
int lrf_adam_step_pack(const LrfAdamTensor* tensors /* host array */, int32_t count, const float* dev_scalars /* or NULL */,
                       float beta1, float beta2, float eps, const LrfParams* p, void* cache, void* stream);

/* density_L1 regulariser (tensoRF.py:83-92), on by default while
 * rf_iter < n_iters_reg (opt.py:111, local_tensorfs.py:361-375):
 *   out = mean_i sqrt(max(feature2density(sum_p sum_c plane_p[c, i / L_p] line_p[c, i % L_p]), 1e-5))
 * over the g0*g1*g2 lattice, with the reference's per-plane flattening orders.  plane[p]: the
 * density plane [8, hw[p]] (the [1,8,H,W] parameter), line[p]: [8, ll[p]]; hw[p]*ll[p] is the
 * same for all p.  The forward leaves d out_i / d feat_i in the workspace for the backward. */
size_t lrf_density_l1_workspace(const int32_t hw[3], const int32_t ll[3]);
int lrf_density_l1_fwd(const float* const plane[3], const float* const line[3], const int32_t hw[3],
                       const int32_t ll[3], float density_shift, int32_t relu, void* workspace,
                       float* out /* device [1] */, void* stream);
int lrf_density_l1_bwd(const float* const plane[3], const float* const line[3], const int32_t hw[3],
                       const int32_t ll[3], const void* workspace, const float* g_out /* device [1] */,
                       float* const g_plane[3], float* const g_line[3], void* stream);
/* The same, ADDED to what g_plane / g_line hold (every element has one writer; stream order behind whoever filled them): the
 * regulariser's gradient lands in the buffers lrf_render_bwd scattered into -- one autograd node for render + regulariser,
 * no gradient-accumulation passes over the six density tensors between them (TensorVMSplit.fuse_density_L1). */
int lrf_density_l1_bwd_acc(const float* const plane[3], const float* const line[3], const int32_t hw[3],
                           const int32_t ll[3], const void* workspace, const float* g_out /* device [1] */,
                           float* const g_plane[3], float* const g_line[3], void* stream);

/* Pose assembly: LocalTensorfs.get_cam2world (local_tensorfs.py:292-299) with sixD_to_mtx
 * (utils/utils.py:381-388): per frame a 6D rotation [3,2] (Gram-Schmidt -> columns b1, b2, b1 x b2)
 * and a translation [3] -> cam2world [V,3,4].  r6d / trans are HOST arrays of V device pointers (the
 * per-frame parameters are separate tensors), 1 <= V <= LRF_POSE_MAX per call. */
#define LRF_POSE_MAX 64
/* cross_views != 0 (V == 3 only): b3 is the cross product over the VIEW axis, which is what the
 * reference's dim-less torch.cross (utils/utils.py:386) computes for a stack of exactly 3 views. */
int lrf_pose_assemble(const float* const* r6d, const float* const* trans, int32_t V, int32_t cross_views,
                      float* cam2world, void* stream);
/* g_cam2world [V,3,4] -> g_r6d [V,3,2], g_trans [V,3] */
int lrf_pose_assemble_bwd(const float* const* r6d, int32_t V, int32_t cross_views, const float* g_cam2world,
                          float* g_r6d, float* g_trans, void* stream);

/* Alpha-mask rebuild on the device: TensorBase.getDenseAlpha + updateAlphaMask
 * (tensorBase.py:501-536) as two launches, no host synchronisation.
 * lrf_dense_alpha: alpha = 1 - exp(-sigma * length) at every point of the gx x gy x gz lattice spanning the
 * field aabb (lin_* = device arrays torch.linspace(0, 1, g), as :504-508), through the field's CURRENT mask
 * when it has one (compute_alpha, :538-558); output [gz][gy][gx], the order :523 transposes to.
 * lrf_alpha_pool_threshold: clamp(0,1), 3x3x3 max-pool (stride 1, padding 1), out = pooled >= thres ? 1 : 0. */
int lrf_dense_alpha(const LrfField* f, const float* lin_x, const float* lin_y, const float* lin_z,
                    int32_t gx, int32_t gy, int32_t gz, float length, uint32_t flags, float* alpha, void* stream);
int lrf_alpha_pool_threshold(const float* alpha, int32_t gx, int32_t gy, int32_t gz, float thres, float* out,
                             void* stream);

/* TV regulariser (utils/utils.py:293-309 as applied by tensoRF.py:94-110; weights 0 by default,
 * opt.py:112-113): loss = sum_t scale_t * 2 w (sum_h (dx)^2 / (C (H-1) W) + sum_w (dx)^2 / (C H (W-1)))
 * over up to LRF_TV_MAX tensors x_t [C,H,W] (a line is [C,L,1]); scale 1e-2 for planes, 1e-3 for
 * lines.  `segs` is a host array; g (gradient, same shape, written not accumulated) is used by _bwd. */
#define LRF_TV_MAX 16
typedef struct LrfTvSeg { const float* x; float* g; int32_t C, H, W; float scale; } LrfTvSeg;
size_t lrf_tv_workspace(const LrfTvSeg* segs, int32_t count);
int lrf_tv_loss_fwd(const LrfTvSeg* segs, int32_t count, float weight, void* workspace, float* out /* device [1] */,
                    void* stream);
int lrf_tv_loss_bwd(const LrfTvSeg* segs, int32_t count, float weight, const float* g_out /* device [1] */,
                    void* stream);

/* ---- grid upsample and the geometric losses around the path ------------------------------
 * lrf_upsample_bilinear: TensorVMSplit.up_sampling_VM (models/tensoRF.py:198-221) = F.interpolate(mode="bilinear",
 * align_corners=True) of one plane [C,H,W] -> [C,H2,W2] (a line is [C,L,1] -> [C,L2,1]). */
int lrf_upsample_bilinear(const float* src, int32_t C, int32_t H, int32_t W, float* dst, int32_t H2, int32_t W2, void* stream);

/* Optical-flow loss of train.py:385-412 with utils/utils.py:15-48 (pts2px, inverse_pose, get_cam2cams,
 * get_pred_flow).  V views of n rays each (rays of a view contiguous, as LocalTensorfs.forward returns them):
 * arr[v][j] = sum|pred_bwd - bwd_flow| * bwd_mask + sum|pred_fwd - fwd_flow| * fwd_mask (fwd_mask counts as 0 for the
 * views flagged in fwd_off: train.py:396 flags `view_ids == len(cam2world) - 1`, absolute id against slice length),
 * entries above the view's `quantile` (0.9) zeroed (train.py:408).
 * fwd: arr_out [V*n] (after zeroing), view_sum [V] (sum of a view's kept entries; flow_loss_arr.mean() =
 * sum(view_sum) / (V n)).  bwd: gradients of scale * g_loss[0] * sum(arr) with respect to depth [V*n], dirs [V*n,3],
 * cam2world [F,3,4] and (focal, cx, cy) per view [V,3]; workspace = V * 36 floats. */
#define LRF_LOSS_MAX_PER_VIEW 4096
typedef struct LrfFlowLoss {
  const float* cam2world;   /* [F,3,4]: LocalTensorfs.get_cam2world(starting_id) */
  const int32_t* frame;     /* [V]: view id - starting frame id */
  const int32_t* fwd_off;   /* [V]: != 0 -> this view's forward mask is zeroed (train.py:396) */
  const float* dirs;        /* [V*n,3] */
  const float* depth;       /* [V*n] */
  const int64_t* ij;        /* [V*n,2] (col, row) */
  const float* fwd_flow; const float* fwd_mask; const float* bwd_flow; const float* bwd_mask;   /* [V*n,2], [V*n] */
  const float* focal;       /* device [1] */
  const float* center;      /* device [2] */
  int32_t F, V, n;
  float quantile;
} LrfFlowLoss;
int lrf_flow_loss_fwd(const LrfFlowLoss* a, float* arr_out, float* view_sum, void* stream);
int lrf_flow_loss_bwd(const LrfFlowLoss* a, const float* arr_out, const float* g_loss /* device [1] */, float scale,
                      float* g_depth, float* g_dirs, float* g_cam2world, float* g_intr /* [V,3] */, float* workspace,
                      void* stream);
/* Monocular-depth loss of train.py:414-423 with compute_depth_loss (utils/utils.py:50-59) on x = 1 / clamp(depth, 1e-6)
 * and the target inverse depths gt: per view median / mean-abs-deviation normalisation of both, squared difference,
 * entries above the view's `quantile` (0.8) zeroed.  stats [V,6] carries the per-view statistics to the backward. */
int lrf_depth_loss_fwd(const float* depth, const float* gt, int32_t V, int32_t n, float quantile, float* arr_out,
                       float* stats, float* view_sum, void* stream);
int lrf_depth_loss_bwd(const float* depth, const float* gt, int32_t V, int32_t n, const float* arr_out, const float* stats,
                       const float* g_loss /* device [1] */, float scale, float* g_depth, void* stream);

/* Photometric loss of train.py:369-371: loss = mean over [R,3] of 0.25 |rgb - target| w_i / mean(w), one launch each way.
 * w [R] or NULL (unit weights); w_mean: device [1] or NULL (= the mean of w over this batch; under ray sharding the batch-global
 * mean).  loss, aux: device [1] each; aux carries 0.25 / (mean(w) 3 R) to the backward.  g_rgb [R,3] = g_loss[0] d loss / d rgb. */
int lrf_photo_loss_fwd(const float* rgb, const float* target, const float* w, const float* w_mean, int32_t R, float* loss, float* aux, void* stream);
int lrf_photo_loss_bwd(const float* rgb, const float* target, const float* w, const float* aux, const float* g_loss /* device [1] */,
                       int32_t R, float* g_rgb, void* stream);

/* Batch assembly (train.py:352-358, 385-420 index the dataset tensors with the batch's (view, pixel) ids, one ATen gather
 * and one mask expression per tensor): target colours, both optical flows with their masks (forward: the view is not the
 * last image; backward: not the first) and inverse depths of V x n pixels in one launch.  Dataset tensors [n_images, HW, 3 | 2 |
 * 2 | 1] float32, a NULL tensor (or output) is skipped; view_ids int64 [V] (negative ids count from the end); pix int64 [V, n]:
 * pixel ids inside the view. */
typedef struct {
  const float* images; const float* fwd_flow; const float* bwd_flow; const float* invdepths;
  const int64_t* view_ids; const int64_t* pix;
  int32_t V, n, HW, n_images;
} LrfBatchGather;
int lrf_batch_gather(const LrfBatchGather* a, float* target /* [V n, 3] */, float* fwd_flow /* [V n, 2] */, float* fwd_mask /* [V n] */,
                     float* bwd_flow, float* bwd_mask, float* invdepth /* [V n] */, void* stream);

/* Loss assembly (train.py:425-437: loss + flow * w_flow * reg / ((W + H) / 2) + depth * w_depth * reg + L1 ...):
 * total = sum_k w_k sum_j x_k[j] with w_k = a_k + b_k s[0] -- s a device scalar (the schedule weight of the iteration), x_k a
 * device scalar or a vector of n_k partial sums (the per-view sums of lrf_flow_loss_fwd / lrf_depth_loss_fwd, added in index
 * order).  One launch each way instead of a dozen scalar kernels; w_out [count] carries the weights to the backward,
 * g[k] = g_total[0] w_k = d total / d x_k[j]. */
#define LRF_LOSS_TERMS_MAX 8
typedef struct {
  const float* x[LRF_LOSS_TERMS_MAX]; int32_t n[LRF_LOSS_TERMS_MAX];
  float a[LRF_LOSS_TERMS_MAX], b[LRF_LOSS_TERMS_MAX];
  int32_t count; const float* s;
} LrfLossTerms;
int lrf_loss_combine_fwd(const LrfLossTerms* t, float* total /* device [1] */, float* w_out /* device [count] */, void* stream);
int lrf_loss_combine_bwd(const float* w, const float* g_total /* device [1] */, int32_t count, float* g /* device [count] */, void* stream);

/* Row gather out[v,:] = src[idx[v],:] (src [F,K], idx int64 [V], negative ids count from the end) and its backward
 * g_src[f,:] = sum_{v: idx[v]=f} g_out[v,:] in v order (no atomics): the per-view poses / exposures a batch picks out of the per-frame
 * tables (local_tensorfs.py:292-299 stacks the sampled frames' parameters; :496 indexes the stacked exposures). */
int lrf_rows_gather(const float* src, const int64_t* idx, int32_t V, int32_t K, int32_t F, float* out, void* stream);
int lrf_rows_gather_bwd(const float* g_out, const int64_t* idx, int32_t V, int32_t K, int32_t F, float* g_src, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LRF_H_ */
