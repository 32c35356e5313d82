// One source, two translation units (see __graft_entry__.build):
//   LRF_TU 1  default flags                       -> every C-ABI entry point except the training ones
//   LRF_TU 2  the same source, -fno-slp-vectorize  -> lrf_render_fwd_train, lrf_render_bwd and what belongs to them
// Why: hipcc's SLP vectoriser packs the scalar fp32 chains of the backward kernels into v_pk_* pairs and raises their
// register pressure (k_bwd_shade_dgrad: 228 -> 28 B of scratch per lane without it): forward+backward 2.92 -> 2.76 ms at
// configs[1]; the eval forward is 2 % FASTER with it (0.2027 vs 0.2066 ms) -- measured on one box, scripts/gpu_diag.py
// fuse / bwd_overlap with LRF_LIB pointing at either build.  Each unit compiles the whole source inside its own
// namespace; the entry points it does not provide get private names here, so the library exports every symbol of
// include/lrf.h exactly once.  LRF_TU undefined: a plain single-unit build (ISA dumps, tools).
#pragma once
#ifndef LRF_TU
#define LRF_TU 0
#endif

#if LRF_TU == 1
#define lrf_render_fwd_train          lrf_tu1_render_fwd_train
#define lrf_render_bwd                lrf_tu1_render_bwd
#define lrf_workspace_bytes_bwd       lrf_tu1_workspace_bytes_bwd
#define lrf_workspace_layout_bwd      lrf_tu1_workspace_layout_bwd
#define lrf_debug_set_bwd_overlap     lrf_tu1_debug_set_bwd_overlap
#define lrf_debug_set_train_fwd_engine lrf_tu1_debug_set_train_fwd_engine
#elif LRF_TU == 2
#define lrf lrf_tu2                    /* the namespace of this unit */
#define lrf_abi_version               lrf_tu2_abi_version
#define lrf_adam_step                 lrf_tu2_adam_step
#define lrf_alpha_pool_threshold      lrf_tu2_alpha_pool_threshold
#define lrf_app_feature               lrf_tu2_app_feature
#define lrf_cache_bytes               lrf_tu2_cache_bytes
#define lrf_debug_set_app_oversubscribe lrf_tu2_debug_set_app_oversubscribe
#define lrf_debug_set_dump            lrf_tu2_debug_set_dump
#define lrf_debug_set_lds_lines       lrf_tu2_debug_set_lds_lines
#define lrf_debug_set_mlp_policy      lrf_tu2_debug_set_mlp_policy
#define lrf_debug_set_mlp_threads     lrf_tu2_debug_set_mlp_threads
#define lrf_debug_poison_cu_state     lrf_tu2_debug_poison_cu_state
#define lrf_debug_saved_row_offset    lrf_tu2_debug_saved_row_offset
#define lrf_debug_set_shade_pipe      lrf_tu2_debug_set_shade_pipe
#define lrf_debug_set_skew            lrf_tu2_debug_set_skew
#define lrf_debug_set_subbatches      lrf_tu2_debug_set_subbatches
#define lrf_dense_alpha               lrf_tu2_dense_alpha
#define lrf_density_feature           lrf_tu2_density_feature
#define lrf_density_l1_bwd            lrf_tu2_density_l1_bwd
#define lrf_density_l1_fwd            lrf_tu2_density_l1_fwd
#define lrf_density_l1_workspace      lrf_tu2_density_l1_workspace
#define lrf_depth_loss_bwd            lrf_tu2_depth_loss_bwd
#define lrf_depth_loss_fwd            lrf_tu2_depth_loss_fwd
#define lrf_flow_loss_bwd             lrf_tu2_flow_loss_bwd
#define lrf_flow_loss_fwd             lrf_tu2_flow_loss_fwd
#define lrf_last_error                lrf_tu2_last_error
#define lrf_pack_field                lrf_tu2_pack_field
#define lrf_pose_assemble             lrf_tu2_pose_assemble
#define lrf_pose_assemble_bwd         lrf_tu2_pose_assemble_bwd
#define lrf_render_fwd                lrf_tu2_render_fwd
#define lrf_render_fwd_profile        lrf_tu2_render_fwd_profile
#define lrf_sample_ray_aabb           lrf_tu2_sample_ray_aabb
#define lrf_scene_blend               lrf_tu2_scene_blend
#define lrf_scene_blend_bwd           lrf_tu2_scene_blend_bwd
#define lrf_scene_rays                lrf_tu2_scene_rays
#define lrf_scene_rays_bwd            lrf_tu2_scene_rays_bwd
#define lrf_tv_loss_bwd               lrf_tu2_tv_loss_bwd
#define lrf_tv_loss_fwd               lrf_tu2_tv_loss_fwd
#define lrf_tv_workspace              lrf_tu2_tv_workspace
#define lrf_upsample_bilinear         lrf_tu2_upsample_bilinear
#define lrf_workspace_bytes           lrf_tu2_workspace_bytes
#endif
