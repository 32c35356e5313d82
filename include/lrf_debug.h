/* lrf_debug.h -- test hooks of liblrf_hip.so.  NOT part of the drop-in ABI (include/lrf.h): these are process-wide
 * switches for the diagnostics in scripts/gpu_diag.py and for a few tests; a host thread that flips one while another
 * renders changes that render too.  Production callers never include this header. */
#ifndef LRF_DEBUG_H_
#define LRF_DEBUG_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* When set (device buffer of n_CUs * 8 waves * 8 uint64), lrf_render_fwd runs k_shade3<TIMED> and leaves per-wave
 * s_memtime totals there: {prologue, header + position, gather + split, -, -, chain, tiles, finalize}; NULL = off. */
void    lrf_debug_set_dump(float* buf);
void    lrf_debug_set_lds_lines(int on);          /* k_march: density lines staged in LDS (default on when they fit) */
void    lrf_debug_set_pipe_chunk(int rays);       /* rays per chunk of lrf_render_fwd's two-stream large-batch mode (default 16384; batches of at least two chunks); 0: one pass over the whole batch.  Changes lrf_workspace_bytes: set it before sizing a workspace */
void    lrf_debug_set_scene_fuse(int on);         /* lrf_scene_fwd: several fields per march / colour launch (default on); 0 = field by field */
void    lrf_debug_set_bwd_overlap(int on);        /* lrf_render_bwd: two branches on two streams (default on); 1 + 2 * (n + 1): k_wgrad_w2w3 on the caller's stream (n > 0, default) or on the side stream behind the density scatter (n = 0) */
void    lrf_debug_set_train_fwd_engine(int bits); /* 8: plane and line gradients by separate scatter kernels; 16: the gradient scatters with fp32 compare-and-swap adds (k_scatter_plane, rounds 2-5) instead of 64-bit fixed point (k_scatter_fix); 256: fixed point for the density tensors only (default: the appearance tensors too where their accumulators fit in LDS, lines up to 479 cells); 1024: no four-channel sweeps (lines of 480 .. 640 cells: the appearance tensors through the compare-and-swap kernel, as before); 512: k_wgrad_w2w3 in its single-buffered 128-row form (rounds 4-5) instead of the double-buffered 64-row one; 32 / 64 / 128: k_train_dgrad3 + k_train_app3 without their row stores / position gradient and X / dz1 products (timing experiments, wrong results) */
/* float offset of (row, col) inside the ACT (0) / GRD (1) region of a training workspace (MFMA-fragment order,
 * csrc/lrf_common.h); buffer 2: X-block column of appearance channel col */
int64_t lrf_debug_saved_row_offset(int buffer, uint64_t row, int col);

#ifdef __cplusplus
}
#endif
#endif
