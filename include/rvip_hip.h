/*
 * rvip_hip.h -- C ABI of librvip_hip.so: the MI355X (gfx950) kernels behind the heatmap-regression
 * U-Net training step of Cardio-AI/cmr-landmark-detection.
 *
 * The reference has no FFI for this path: every operation below is executed for it by
 * tensorflow==2.3.0 when Keras runs the graph that src/models/Unets.py:61-133 (create_unet),
 * :755-869 (unet) and src/models/KerasLayers.py:660-777 (conv_layer_fn / downsampling_block_fn /
 * upsampling_block_fn) compose, under Model.fit (src/models/train_model.py:105-112).  Each entry
 * point names the Keras call site whose arithmetic it replaces.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - returns 0 (RVIP_OK) or a negative RVIP_E* code; never throws, never allocates, never syncs;
 *   - the CALLER owns every buffer (device memory) and passes the HIP stream (hipStream_t as void*);
 *     all work is enqueued on that stream; re-entrant across streams;
 *   - activations NHWC, dtype tag per call (RVIP_F32 / RVIP_BF16 / RVIP_F16), accumulation always fp32;
 *   - parameters (bias, BN gamma/beta/stats), gradients and optimiser state are fp32;
 *   - master conv kernels are Keras HWIO fp32 [kh][kw][Cin][Cout]; the MFMA kernels read "packed"
 *     copies in the activation dtype (rvip_pack_conv3x3_weights);
 *   - channel counts of bf16 tensors must be multiples of 8, of f32 tensors multiples of 4
 *     (16-byte channel vectors), except the network input (Cin = 1) and the 1x1 head (<= 4 classes);
 *   - "state" is a caller-owned device block of RVIP_STATE_WORDS 32-bit words (see below) so that
 *     step-dependent values (Adam t, learning rate, dropout stream) survive hipGraph replay.
 */
#ifndef RVIP_HIP_H
#define RVIP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RVIP_OK            0
#define RVIP_EINVAL      (-1)   /* bad shape / alignment / null pointer */
#define RVIP_EUNSUPPORTED (-2)  /* combination not built */
#define RVIP_EWORKSPACE  (-3)   /* workspace too small */
#define RVIP_ELAUNCH     (-4)   /* hipLaunch error (see rvip_last_hip_error) */

#define RVIP_F32  0
#define RVIP_BF16 1
#define RVIP_F16  2      /* IEEE binary16 storage, fp32 accumulate; the caller scales the loss gradient (rvip_adam_step grad_scale) */

#define RVIP_ACT_NONE    0
#define RVIP_ACT_RELU    1
#define RVIP_ACT_ELU     2      /* alpha = 1 (Unets.py:82 default 'elu') */
#define RVIP_ACT_SIGMOID 3

#define RVIP_LOSS_MSE      0    /* tf.keras.losses.MSE, train_model.py:184 */
#define RVIP_LOSS_BCE_DICE 1    /* Loss_and_metrics.py:229-245, w_bce=0.5 w_dice=1 */

/* device-resident step state (uint32 words; floats stored by bit pattern) */
#define RVIP_STATE_STEP   0     /* optimizer.iterations (completed steps) */
#define RVIP_STATE_LR     1     /* float: learning rate (model.optimizer.lr) */
#define RVIP_STATE_SEED   2     /* dropout seed */
#define RVIP_STATE_WORDS  8

#define RVIP_BIT_OF_CHANNEL(c) (8 * (((c) & 15) >> 2) + 4 * (((c) & 31) >> 4) + ((c) & 3))      /* bit planes of rvip_conv3x3_desc / rvip_apply_desc */
#define RVIP_ABI_VERSION 8   /* what rvip_abi_version() of a matching library returns; _native.py checks it at every load */
int         rvip_abi_version(void);
const char* rvip_build_info(void);          /* "gfx950 ..." */
int         rvip_last_hip_error(void);      /* last hipError_t seen by a launcher (0 = none) */
/* Diagnostic (the ONLY entry point that synchronises): hipDeviceSynchronize, then the runtime's sticky error; returns the first
 * non-zero hipError_t of the two, 0 when the device is idle and clean.  tests/conftest.py calls it after every GPU test so that an
 * asynchronous kernel fault is reported by the test that launched it. */
int         rvip_device_check(void);

/* ------------------------------------------------------------------------------------------------
 * 3x3 "same" convolution as implicit GEMM on MFMA.  Replaces Conv2D(filters, 3, padding='same',
 * activation=act) -- KerasLayers.py:683,689 (conv_layer_fn) and :758 (up-conv) -- including the
 * UpSampling2D (:756-757) and Concatenate (:767) in front of it, which are addressing modes here:
 *   source 0: x0 [N, H>>(up0!=0), W>>(up0!=0), C0]
 *             up0 = 1: nearest-neighbour x2 read, v[h,w] = x[h/2,w/2]          (UpSampling2D)
 *             up0 = 2: zero-stuffed x2 read, v[2i+1,2j+1] = x[i,j], 0 elsewhere (Conv2DTranspose(3, strides=2,
 *                      'same'), KerasLayers.py:761-765: out[j] = sum_{2i+k=j} in[i] W[k] is this conv with taps
 *                      reversed; the host keeps the layer's master kernel in that equivalent HWIO form)
 *   source 1: x1 [N, H, W, C1] or NULL     (channels C0.. of the virtual concat [x0, x1])
 * Output y [N,H,W,Cout] = act(conv + bias).  With y1 != NULL the output channels are split:
 * [0,csplit) -> y (row stride csplit), [csplit,Cout) -> y1 (row stride Cout-csplit); used by the
 * data-gradient of a conv that read a concat.
 * The same entry point computes the data gradient: pass dy as x0, the dgrad-packed weights, no
 * bias, act NONE (autodiff of the Keras layer).
 * ------------------------------------------------------------------------------------------------ */
typedef struct rvip_conv3x3_desc {
    const void*  x0;   int32_t c0;   int32_t up0;
    const void*  x1;   int32_t c1;
    const void*  w_packed;            /* [9][Cout][C0+C1] in the activation dtype */
    const float* bias;                /* [Cout] or NULL */
    void*        y;    void* y1;      int32_t csplit;
    int32_t      n, h, w, cout;
    int32_t      act;                 /* RVIP_ACT_* applied after bias */
    int32_t      dtype;               /* RVIP_F32 / RVIP_BF16 (x0, x1, w_packed, y, y1) */
    /* Conv3D(3x3x3, 'same') on NDHWC (KerasLayers.py:679 f_size[:ndims]; cfg 5): the N images are volumes of `depth`
     * consecutive slices and the contraction also runs over kd = 3 depth taps (slice d-1, d, d+1 of the same
     * volume, zeros outside); w_packed is then [27][Cout][C0+C1], tap = (kd, kh, kw) row-major.  0 / 0 (or 1 / 1)
     * = the plain 2-D convolution. */
    int32_t      depth, kd;
    /* down2 != 0: y is [N, H/2, W/2, Cout] and receives the sum of every 2x2 block of the result: the data gradient
     * of an UpSampling2D -> conv pair in one pass (no bias / activation / y1 in this mode; H, W even). */
    int32_t      down2;
    /* subpix != 0 (with up0 = 1, no x1 / y1): the UpSampling2D -> conv pair in its sub-pixel form - four 2x2-tap phase
     * convolutions on the low-resolution x0, 16 instead of 36 multiply-adds per low-resolution pixel; same result up to
     * the summation order of the taps.  w_packed is then the [4][4][Cout][C0] block of rvip_pack_subpixel_weights.
     * subpix = 2 (ABI 6; 16-bit types, rvip_conv3x3_fwd / rvip_conv3x3_fwd_sums without mask_bits; up0 = 0, no x1 / y1 / bias /
     * activation / down2): the DATA GRADIENT of that pair in the same form.  x0 is the gradient at the up-sampled size [N, h, w, C0],
     * y the gradient of the low-resolution tensor [N, h/2, w/2, Cout]; the contraction runs over the four source phases
     * x0[2y + al][2x + be] (an addressing mode, nothing is re-laid out) x 2x2 summed taps x C0 -- 16 instead of 36 multiply-adds per
     * low-resolution pixel and no 2x2-sum epilogue (down2 is the nine-tap form of the same launch).  w_packed is the
     * [4][4][Cout][C0] block of rvip_pack_subpixel_dgrad_weights (summed taps rounded once, like the forward form's).
     * RVIP_EUNSUPPORTED for shapes the LDS-DMA kernel does not serve: fall back to down2. */
    int32_t      subpix;
    /* stream_in != 0: hint that this launch is the last reader of x0 for a while (e.g. the data gradient reading dz):
     * its input is fetched with the non-temporal cache policy. */
    int32_t      stream_in;
    /* Bit planes (ABI 5): one 32-bit word per pixel and 32-channel block, laid out [ceil(C/32)][N*H*W]; channel c of a pixel is bit
     * RVIP_BIT_OF_CHANNEL(c) of word [c / 32][pixel] -- the order in which the four lanes of a pixel hold its channels in the MFMA
     * epilogue (byte k of the word = channels 4k..4k+3 in its low and 16+4k..16+4k+3 in its high nibble), so that a lane writes and
     * reads one byte / one nibble without a lane exchange.
     * mask_bits (rvip_conv3x3_fwd_sums only): the first mask_channels (a multiple of 32) channels of the result are gated element-wise,
     * v = bit ? v * mask_scale : 0, before they are stored and summed.  Two uses: the Dropout backward (KerasLayers.py:718,772) when
     * the result is the gradient reaching a Dropout layer's output -- bits = the keep bits rvip_bn_apply wrote in the forward pass
     * (rvip_apply_desc.keep_bits), scale = 1 / (1 - rate) -- and the ReLU backward of a stage without BatchNormalization (the
     * up-conv, KerasLayers.py:758): bits = sign_bits of its forward launch, scale = 1; the stage then needs no rvip_bn_bwd_apply
     * pass at all.  The loader waves stage the words of every pixel tile through LDS; not with down2.
     * sign_bits (rvip_conv3x3_fwd only; Cout % 32 == 0, no y1 / down2): the launch also writes, per stored value, whether it is
     * > 0, in that layout ([Cout/32][N*H*W]; H, W of the full-resolution result for subpix).  rvip_conv3x3_sign_bits_ok(d) tells
     * whether the launch is served by a kernel that does (the register-staged fallback does not). */
    const uint32_t* mask_bits; int32_t mask_channels; float mask_scale;
    uint32_t*    sign_bits;
    /* rvip_conv3x3_fwd_sums only: the caller reads columns >= sums_from of the partial rows only (a multiple of 32; e.g. csplit when
     * the first half of a split result belongs to a stage without BatchNormalization); the columns below are unspecified. */
    int32_t      sums_from;
    /* ABI 8 -- cu_limit in 1..255: the persistent grid of the LDS-DMA kernels is sized for that many compute units instead of all 256
     * (one workgroup per CU), so that ANOTHER launch can run beside this one on the rest of the chip: the weight gradient of the same
     * layer on a second stream (both read the same gradient tensor), or RCCL's kernels while a gradient bucket is in flight
     * (RVIP_RCCL_CU_RESERVE).  The result tensor does not depend on it; the partial rows of the _stats / _sums forms do
     * (rvip_conv3x3_fwd_*_rows answers for the limit in the descriptor), and with them the last bits of what is folded from them.
     * 0 (or >= 256): the whole chip. */
    int32_t      cu_limit;
} rvip_conv3x3_desc;

int rvip_conv3x3_fwd(const rvip_conv3x3_desc* d, void* stream);
int rvip_conv3x3_sign_bits_ok(const rvip_conv3x3_desc* d);

/* The same convolution with the BatchNormalization statistics of its (stored) output fused into the epilogue:
 * writes rvip_conv3x3_fwd_stats_rows(d) partial rows [rows][2][cout] (per-channel sum, sum of squares) to stats_ws;
 * finish with rvip_bn_stats_finalize.  rows == 0 means this shape runs on the register-staged fallback kernel,
 * which does not fuse statistics (use rvip_conv3x3_fwd + rvip_bn_train_stats).  Not with y1 / down2 / subpix. */
int rvip_conv3x3_fwd_stats_rows(const rvip_conv3x3_desc* d);
int rvip_conv3x3_fwd_stats(const rvip_conv3x3_desc* d, float* stats_ws, size_t stats_ws_bytes, void* stream);
/* ABI 5, for data-gradient launches: the convolution with the per-channel SUMS of its stored result as rvip_conv3x3_fwd_sums_rows(d)
 * partial rows [rows][cout] (of the fp32 values in front of the storage rounding) -- also with y1 (channel c of the virtual [y, y1] row is column c), with down2 (sums of the stored 2x2
 * block sums) and with mask_bits.  Added over the rows, column c is the `T1 = sum g` term of the BatchNormalization backward of
 * the stage that produced channel c of this launch's result tensor (rvip_bn_bwd_coef).  rows == 0: fallback kernel, no sums. */
int rvip_conv3x3_fwd_sums_rows(const rvip_conv3x3_desc* d);
int rvip_conv3x3_fwd_sums(const rvip_conv3x3_desc* d, float* sums_ws, size_t sums_ws_bytes, void* stream);

/* Re-layout the fp32 HWIO master kernel [3][3][Cin][Cout] into the two packed operands:
 *   w_fwd [9][Cout][Cin]  (w_fwd[t][o][i] = W[t][i][o])        -- forward
 *   w_dgrad [9][Cin][Cout] (w_dgrad[t][i][o] = W[8-t][i][o])   -- data gradient (taps rotated 180)
 * either may be NULL. */
int rvip_pack_conv3x3_weights(const float* w_hwio, int cin, int cout, int dtype,
                              void* w_fwd, void* w_dgrad, void* stream);

/* Phase kernels of the sub-pixel form of UpSampling2D -> conv (rvip_conv3x3_desc.subpix): w_phase[2a+b][2u+v][Cout][Cin];
 * rvip_pack_subpixel_dgrad_weights: those of its data gradient (subpix = 2), w_dphase[2al+be][2u+v][Cin][Cout] (Cin of the LAYER
 * = the channels of that launch's result). */
int rvip_pack_subpixel_weights(const float* w_hwio, int cin, int cout, int dtype, void* w_phase, void* stream);
int rvip_pack_subpixel_dgrad_weights(const float* w_hwio, int cin, int cout, int dtype, void* w_dphase, void* stream);

/* The same re-layout for ALL 3x3 kernels of a model in one launch.  `theta` is the flat fp32 parameter block;
 * `table` is a DEVICE array of `entries` records {int64 w_off (floats into theta), int64 f_off, int64 d_off
 * (elements into wf_base / wd_base), int32 cin, int32 cout, int32 taps (9, or 27 for a 3x3x3 kernel; 0 = 9),
 * int32 mode (0: the two operands above; 1: the [4][4][Cout][Cin] phase kernels of rvip_pack_subpixel_weights at
 * f_off and the [4][4][Cin][Cout] ones of rvip_pack_subpixel_dgrad_weights at d_off)}; max_elems = max over entries of taps*cin*cout. */
typedef struct rvip_pack_entry { long long w_off, f_off, d_off; int32_t cin, cout; int32_t taps, mode; } rvip_pack_entry;
int rvip_pack_all_conv3x3_weights(const float* theta, const void* table, int entries, int max_elems, int dtype,
                                  void* wf_base, void* wd_base, void* stream);
/* The launch sees only the DEVICE copy of the table: call this on the HOST array before uploading it.  RVIP_EINVAL for an entry
 * no kernel of this library takes: cin / cout <= 0, cout % 4 != 0 (16-byte rows of four output channels), offsets that are
 * negative or not multiples of 4 elements, taps other than 9 / 27 (mode 0) or 9 (mode 1: phase kernels exist for 3x3 only), an
 * unknown mode.  (The kernel itself takes the element path for cout % 4 != 0 and stays inside its rows whatever the table says.) */
int rvip_pack_table_check(const rvip_pack_entry* host_table, int entries, int dtype);
/* The same launch as the closing one of an optimiser step: also increments state[RVIP_STATE_STEP] (= rvip_state_tick,
 * without its launch).  Must follow rvip_adam_step on the same stream. */
int rvip_pack_all_conv3x3_weights_tick(const float* theta, const void* table, int entries, int max_elems, int dtype,
                                       void* wf_base, void* wd_base, uint32_t* state, void* stream);

/* Weight gradient of the same conv (autodiff of KerasLayers.py:683,689,758):
 *   dw[t][i][o] = sum_{n,h,w} X[n,h+t/3-1,w+t%3-1,i] * dy[n,h,w,o],   X = virtual [up(x0), x1]
 * fp32 HWIO output.  Deterministic two-stage split-K: workspace holds nsplit partial slabs.
 * rvip_conv3x3_wgrad_workspace returns the bytes needed for the shape. */
typedef struct rvip_wgrad3x3_desc {
    const void*  x0;   int32_t c0;   int32_t up0;
    const void*  x1;   int32_t c1;
    const void*  dy;                  /* [N,H,W,Cout] */
    float*       dw;                  /* [9][C0+C1][Cout] fp32, overwritten */
    int32_t      n, h, w, cout;
    int32_t      dtype;
    void*        workspace;  size_t workspace_bytes;
    int32_t      depth, kd;           /* Conv3D: as in rvip_conv3x3_desc; dw is then [27][C0+C1][Cout] */
    /* defer_fold != 0 (2-D only): stop after stage 1 - the rvip_conv3x3_wgrad_splits(d) slabs [splits][9*(C0+C1)*Cout]
     * stay in `workspace` (which the caller must then keep private to this layer) and dw is not written; sum them later,
     * together with other layers', with rvip_fold_rows_batch(..., wide = 1). */
    int32_t      defer_fold;
    /* dot_rows != NULL (ABI 5; needs w_master, excludes defer_fold): the fold of the slabs also writes
     * rvip_conv3x3_wgrad_dot_rows(d) partial rows [rows][C0+C1] of DOUBLES whose column sums are  T2[i] = sum_{t,o} Wr[t][i][o] * dw[t][i][o],
     * Wr = w_master rounded to `dtype` (what the data gradient multiplies with).  Because the conv is linear in its input X,
     * T2[i] = sum_pixels X[.,i] * dX[.,i]: the `sum g*y` term of the BatchNormalization backward of whichever stage produced
     * channel i of X, without a pass over g and y (rvip_bn_bwd_coef). */
    const float* w_master;            /* [taps][C0+C1][Cout] fp32, the layer's HWIO kernel */
    double*      dot_rows; size_t dot_rows_bytes;     /* double: the sum cancels heavily when the gradient is mostly common-mode */
    /* ABI 7 -- w_phase != NULL (with dot_rows, UpSampling2D -> conv layers whose rvip_conv3x3_wgrad_form() is 1): the layer's DATA gradient
     * runs in sub-pixel form (rvip_conv3x3_desc.subpix = 2), i.e. it multiplies with the phase kernels of
     * rvip_pack_subpixel_dgrad_weights -- sums of two / four taps rounded ONCE to `dtype`, not the rounded taps themselves.  The rows
     * are then dotted against THOSE kernels, phase by phase (the four-phase form's slabs still hold the phase-resolved gradients:
     * dw[kh][kw] of a phase's slab is that phase's summed-tap block):  T2'[i] = sum_{phase, u, v, o} w_phase[phase][u][v][i][o] *
     * dW_phase[phase][u][v][i][o] = sum_pixels X[., i] * dX'[., i]  for the dX' the sub-pixel data gradient really writes, so that
     * T2' and the column sums of dX' describe the same gradient (rvip_bn_bwd_coef).  RVIP_EUNSUPPORTED for every other form. */
    const void*  w_phase;             /* [4][4][C0][Cout] in `dtype` */
    /* ABI 8 -- as rvip_conv3x3_desc.cu_limit: the pixel-split count (rvip_conv3x3_wgrad_splits, _dot_rows) is chosen for that many
     * compute units; fewer splits = fewer slabs to write and fold, and a different (still fixed) summation order of dw. */
    int32_t      cu_limit;
} rvip_wgrad3x3_desc;

size_t rvip_conv3x3_wgrad_workspace(int n, int h, int w, int cin, int cout);
int    rvip_conv3x3_wgrad(const rvip_wgrad3x3_desc* d, void* stream);
int    rvip_conv3x3_wgrad_splits(const rvip_wgrad3x3_desc* d);
int    rvip_conv3x3_wgrad_dot_rows(const rvip_wgrad3x3_desc* d);
/* Which kernel form rvip_conv3x3_wgrad takes for this descriptor (a query; the tests assert the form the real layer shapes get):
 * 0 nine-tap LDS-DMA kernel, 1 sub-pixel form of UpSampling2D -> conv (four phase workgroups), 2 the same with both column phases
 * per workgroup (64 x 32 blocks), 3 register-staged fallback (ragged channel counts), -1 invalid descriptor. */
int    rvip_conv3x3_wgrad_form(const rvip_wgrad3x3_desc* d);

/* ABI 8 -- the weight gradient and the data gradient of one layer as ONE launch: rvip_conv3x3_wgrad(wd) and rvip_conv3x3_fwd_sums(dd,
 * sums_ws, ...) on the two parts of one grid, wd->cu_limit and dd->cu_limit compute units each (both set, their sum <= 256), followed by
 * the weight gradient's slab fold.  dd must be the data gradient of the same layer (dd->x0 == wd->dy, no bias / activation / sign_bits;
 * mask_bits, y1 / csplit and down2 as rvip_conv3x3_fwd_sums takes them).  Every result is bit-identical to the two calls with the
 * same cu_limit values.  16-bit types, 2-D, nine-tap forms of both kernels; rvip_conv3x3_wgrad_dgrad_ok tells whether a pair is
 * served (0: launch the two entry points instead).  Autodiff of Conv2D, KerasLayers.py:683,689. */
int    rvip_conv3x3_wgrad_dgrad_ok(const rvip_wgrad3x3_desc* wd, const rvip_conv3x3_desc* dd);
int    rvip_conv3x3_wgrad_dgrad(const rvip_wgrad3x3_desc* wd, const rvip_conv3x3_desc* dd, float* sums_ws, size_t sums_ws_bytes, void* stream);

/* Batched stage 2 for reductions whose result only the optimiser reads (bias gradients, weight gradients):
 * dst[i] = sum_r src[r*width + i], r < nrows, for `entries` records of a DEVICE table in ONE launch, fixed order.
 * wide = 0: rows of C floats (double accumulation; rvip_bn_bwd_apply with bias_rows); wide = 1: rows of 9*Cin*Cout
 * floats, width % 4 == 0, 16-byte aligned (rvip_conv3x3_wgrad with defer_fold).  max_width = max over the entries. */
typedef struct rvip_fold_entry { const float* src; float* dst; int32_t nrows; int32_t stride; long long width; } rvip_fold_entry;   /* stride (narrow folds): floats between rows, 0 = width */
int rvip_fold_rows_batch(const void* table, int entries, long long max_width, int wide, void* stream);

/* First layer, Cin = 1 (bandwidth-bound; the forward pass without MFMA): y = act(conv3x3(x[N,H,W,1]) + bias); weights are
 * the fp32 HWIO master [9][1][Cout].  wgrad: dw[9][Cout] and nothing else (the input has no grad); with a 16-bit dtype and
 * Cout = 32 the contraction over the pixels runs on the matrix cores (im2col(x) x dy; exact 16-bit products, fp32 sums).
 * workspace for wgrad: rvip_reduce_workspace(n*h*w, 16*cout) bytes. */
int rvip_conv3x3_c1_fwd(const void* x, const float* w, const float* bias, void* y,
                        int n, int h, int w_, int cout, int act, int dtype, void* stream);

/* The same first layer with the BatchNormalization statistics of its stored output fused in (as rvip_conv3x3_fwd_stats):
 * writes rvip_conv3x3_c1_fwd_stats_rows(...) partial rows [rows][2][cout]; finish with rvip_bn_stats_finalize.
 * rows == 0: Cout / VE does not divide 256 (untiled kernel) -- use rvip_conv3x3_c1_fwd + rvip_bn_train_stats. */
int rvip_conv3x3_c1_fwd_stats_rows(int n, int h, int w, int cout, int dtype);
int rvip_conv3x3_c1_fwd_stats(const void* x, const float* w, const float* bias, void* y, int n, int h, int w_, int cout,
                              int act, int dtype, float* stats_ws, size_t stats_ws_bytes, void* stream);
int rvip_conv3x3_c1_wgrad(const void* x, const void* dy, float* dw, int n, int h, int w_, int cout,
                          int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* ABI 7: dw == NULL leaves the fold to the caller -- the rvip_conv3x3_c1_wgrad_rows() partial rows [rows][9][cout] stay in `workspace`
 * (which must then be private to the layer until they are folded), e.g. as one entry {nrows = rows, width = 9 * cout} of
 * rvip_fold_rows_batch(..., wide = 0) together with the step's other deferred folds. */
int rvip_conv3x3_c1_wgrad_rows(int n, int h, int w_, int cout, int dtype);
/* The same first layer of the 3-D graph: Conv3D(3x3x3, 'same') with Cin = 1 on n = N*depth slices (volumes of `depth`
 * consecutive slices), weights fp32 DHWIO [27][1][Cout]; wgrad writes dw[27][Cout].  Cout/VE must divide 256. */
int rvip_conv3d_c1_fwd(const void* x, const float* w, const float* bias, void* y,
                       int n, int depth, int h, int w_, int cout, int act, int dtype, void* stream);
int rvip_conv3d_c1_wgrad(const void* x, const void* dy, float* dw, int n, int depth, int h, int w_, int cout,
                         int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* The first layer with IMG_CHANNELS = 2..4 (Unets.py:77 `Input((*dim, IMG_CHANNELS))`; every config of the reference uses 1):
 * x NHWC [n][h][w][cin] in the activation dtype, weights fp32 HWIO [9][cin][Cout] (9 * cin * Cout * 4 <= 48 KiB), no fused
 * statistics (follow with rvip_bn_train_stats); wgrad writes dw[9][cin][Cout], Cout/VE must divide 256. */
int rvip_conv3x3_cn_fwd(const void* x, const float* w, const float* bias, void* y,
                        int n, int h, int w_, int cin, int cout, int act, int dtype, void* stream);
int rvip_conv3x3_cn_wgrad(const void* x, const void* dy, float* dw, int n, int h, int w_, int cin, int cout,
                          int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Per-channel reductions use a two-stage deterministic scheme: stage 1 writes one partial row per
 * workgroup into the workspace, stage 2 folds them.  rvip_reduce_workspace(rows, width) = bytes for
 * a reduction of `rows` pixels producing `width` floats.
 * ------------------------------------------------------------------------------------------------ */
size_t rvip_reduce_workspace(long long rows, int width);

/* BatchNormalization(axis=-1), training (KerasLayers.py:684,691; Keras defaults momentum 0.99,
 * eps 1e-3).  Step 1 (stats): batch mean / biased variance of z[rows][C]; writes
 *   mean[C], invstd[C], scale[C] = gamma*invstd, shift[C] = beta - mean*scale
 * and updates moving_mean/moving_var in place (unbiased variance when unbiased_moving != 0: the
 * fused 4-D TF kernel; 0 for 5-D inputs). */
int rvip_bn_train_stats(const void* z, long long rows, int c, int dtype,
                        const float* gamma, const float* beta,
                        float* moving_mean, float* moving_var, float momentum, float eps, int unbiased_moving,
                        float* mean, float* invstd, float* scale, float* shift,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Stage 2 of the statistics alone: fold `rows` partial rows [rows][2][c] into mean/invstd/scale/shift and update the
 * moving statistics; count = N*H*W. */
int rvip_bn_stats_finalize(const float* partial, int rows, long long count, int c, const float* gamma, const float* beta,
                           float* moving_mean, float* moving_var, float momentum, float eps, int unbiased_moving,
                           float* mean, float* invstd, float* scale, float* shift, void* stream);

/* Inference coefficients from the moving statistics (Model.predict: BN in inference mode). */
int rvip_bn_infer_coeffs(const float* gamma, const float* beta, const float* moving_mean,
                         const float* moving_var, float eps, int c, float* scale, float* shift, void* stream);

/* y = dropout(act(scale[c]*z + shift[c])), optionally with the 2x2/2 max-pool of y written to
 * `pooled` in the same pass (MaxPooling2D, KerasLayers.py:714,721).  scale/shift NULL = identity.
 * Dropout (KerasLayers.py:718,772; Unets.py:813): inverted dropout, keep = 1-rate; the keep-mask is
 * either `mask` (uint8 [rows][C], parity runs) or the counter-based stream
 * hash(seed, state[STEP], layer_id, element) shared with the backward kernels; rate 0 = none. */
typedef struct rvip_apply_desc {
    const void*  z;  void* y;  void* pooled;      /* pooled NULL = no pool */
    const float* scale; const float* shift;
    int32_t      act;
    float        drop_rate; const uint8_t* mask; const uint32_t* state; int32_t layer_id;
    int32_t      n, h, w, c;
    int32_t      dtype;
    /* pooled != NULL and rvip_bn_apply_argmax_ok(c, dtype): optional [n][h/2][w/2][C/VE] words, 2 bits per channel = position
     * (2*row + col) of the FIRST maximum of the 2x2 window of y, i.e. what rvip_maxpool2x2_bwd would find; lets the BN-backward
     * passes of the stage take (dpooled, argmax, skip gradient) instead of a materialised gradient */
    uint16_t*    argmax;
    /* un-pooled pass with dropout, C % 8 == 0 (ABI 5): also write the keep bits as bit planes [ceil(C/32)][n*h*w] of 32-bit words
     * (bit RVIP_BIT_OF_CHANNEL(c) of word [c / 32][pixel] = element kept), for rvip_conv3x3_desc.mask_bits of the consumer's data gradient */
    uint8_t*     keep_bits;
} rvip_apply_desc;
int rvip_bn_apply(const rvip_apply_desc* d, void* stream);
int rvip_bn_apply_argmax_ok(int c, int dtype);

/* Backward of conv -> [act] -> BN -> [act] -> dropout (autodiff of conv_layer_fn + Dropout).
 * Inputs: dy (grad w.r.t. the dropped-out BN output), z (BN input as stored by the forward) and, only
 * when the activation sits after BN (BN_FIRST, act_after_bn != 0), the forward scale/shift so that
 * act'(act(scale*z+shift)) can be recomputed.
 * Stage 1 (rvip_bn_bwd_reduce): dgamma, dbeta (+ coefficient vectors for stage 2).
 * Stage 2 (rvip_bn_bwd_apply): dconv = (c1*g + c2*z + c3) * act'(.)  and  dbias = sum dconv.
 * With gamma == NULL (no BatchNormalization: BATCH_NORMALISATION False, or the up-conv) stage 1 is
 * skipped by the caller and stage 2 computes dconv = g * act'(z), dbias. */
typedef struct rvip_bnbwd_desc {
    const void*  dy; const void* z;
    void*        dz;                               /* stage 2 output (same dtype) */
    const float* gamma; const float* mean; const float* invstd;
    const float* scale; const float* shift;        /* forward affine; only read when act_after_bn */
    float*       dgamma; float* dbeta; float* dbias;
    float*       coef;                             /* [3][C] scratch written by stage 1, read by stage 2 */
    int32_t      act; int32_t act_after_bn;
    float        drop_rate; const uint8_t* mask; const uint32_t* state; int32_t layer_id;
    long long    rows; int32_t c;
    int32_t      dtype;
    void*        workspace; size_t workspace_bytes;
    /* bias_rows != NULL: rvip_bn_bwd_apply leaves the rvip_bn_bwd_rows(rows, c, dtype) partial rows [rows][C] of the
     * bias gradient there instead of folding them into dbias (fold later with rvip_fold_rows_batch, wide = 0). */
    float*       bias_rows; size_t bias_rows_bytes;
    /* dpooled != NULL: MaxPooling2D backward folded into both stages -- the gradient of pixel (y, x) is
     * round(dy + (argmax of its window == 2*(y&1) + (x&1) ? dpooled : 0)), dy = the skip-connection gradient (may be NULL),
     * exactly the tensor rvip_maxpool2x2_bwd would have stored; rows = n*h*w. */
    const void*  dpooled; const uint16_t* argmax; int32_t h, w;
} rvip_bnbwd_desc;
int rvip_bn_bwd_reduce(const rvip_bnbwd_desc* d, void* stream);
int rvip_bn_bwd_apply(const rvip_bnbwd_desc* d, void* stream);
int rvip_bn_bwd_rows(long long rows, int c, int dtype);

/* Stage 1 of the BatchNormalization backward WITHOUT a pass over (g, z) (ABI 5).  For a stage  z -> BN -> [Dropout] -> y  whose
 * consumers are 3x3 convolutions (through MaxPooling2D / UpSampling2D / Concatenate or directly):
 *   T1[c] = sum g[c]       = column sums of the consumers' data-gradient outputs        (rvip_conv3x3_fwd_sums rows)
 *   T2[c] = sum g[c]*y[c]  = sum_{t,o} W[t][c][o] * dW[t][c][o] over the consumers      (rvip_conv3x3_wgrad dot_rows)
 * and with y = gamma * xhat + beta:  dbeta = T1,  dgamma = sum g*xhat = (T2 - beta*T1) / gamma.  Up to two sources per term
 * (a pooled stage with a skip connection has two consumers); source = `nrows` rows of `stride` floats, this stage's channels
 * start at column `offset`.  Writes dgamma, dbeta and the coef[3][C] vectors of rvip_bn_bwd_apply exactly like
 * rvip_bn_bwd_reduce.  The division needs |gamma| >= min_gamma and |beta| <= max_beta_ratio * |gamma|: a workgroup (32 channels)
 * that holds a channel failing the test recomputes ITS channels exactly -- sum g and sum g*xhat over every row of (g, z) as
 * described by `fallback`, the stage's rvip_bn_bwd_reduce descriptor (dy = the gradient the apply pass will read; drop_rate 0
 * when the consumer's data gradient already applied the Dropout backward) -- slowly (one workgroup per 32 channels), inside the
 * same launch; flags[ceil(C/32)] reports which blocks did (1).  Training from the Keras initialisation never takes that route. */
typedef struct rvip_bncoef_src { const void* rows; int32_t nrows; int32_t stride; int32_t offset; int32_t reserved; } rvip_bncoef_src;   /* t1: float rows, t2: double rows; stride / offset in elements */
typedef struct rvip_bncoef_desc {
    rvip_bncoef_src t1[2]; rvip_bncoef_src t2[2];      /* rows == NULL: unused */
    const float* gamma; const float* beta; const float* mean; const float* invstd;
    float*       dgamma; float* dbeta; float* coef;
    int32_t*     flags;                                /* [ceil(c / 32)] */
    const struct rvip_bnbwd_desc* fallback;            /* required; c, rows == count, act before BN (act_after_bn == 0) */
    long long    count;                                /* N*H*W of the stage */
    int32_t      c;
    float        min_gamma, max_beta_ratio;
} rvip_bncoef_desc;
int rvip_bn_bwd_coef(const rvip_bncoef_desc* d, void* stream);
/* the same count for rvip_bn_bwd_apply_head (its grid is one resident round of workgroups) */
int rvip_bn_bwd_apply_head_rows(long long rows, int c, int dtype, int k);

/* MaxPooling2D backward: dx = route(dpooled -> first max of each 2x2 window of y) + add (add may be
 * NULL; it carries the skip-connection gradient that reaches the same tensor). */
int rvip_maxpool2x2_bwd(const void* y, const void* dpooled, const void* add, void* dx,
                        int n, int h, int w, int c, int dtype, void* stream);

/* Data gradient of the zero-stuffed read: dst[n,i,j] = src[n,2i+1,2j+1]  (h, w = low-resolution extents). */
int rvip_subsample_odd(const void* src, void* dst, int n, int h, int w, int c, int dtype, void* stream);

/* UpSampling2D(2) forward (materialised; the conv reads it virtually) and backward (2x2 sum). */
int rvip_upsample2x_fwd(const void* x, void* y, int n, int h, int w, int c, int dtype, void* stream);
int rvip_upsample2x_bwd(const void* dy, void* dx, int n, int h, int w, int c, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Head + loss.  Replaces Conv2D(mask_classes, 1, activation='sigmoid', name='unet') (Unets.py:128)
 * and the compiled loss (Unets.py:130): Keras MSE or bce_dice_loss, plus the dice metrics
 * (Loss_and_metrics.py:134-171).
 *   rvip_head_fwd:   pred[rows][K] (fp32) = sigmoid(x[rows][Cin] . W[Cin][K] + b); if y_true != NULL
 *                    also folds the loss sums; `sums` (device, 16 floats) receives
 *                    [0] sum (p-t)^2  [1] sum bce  [2] sum t*p  [3] sum t  [4] sum p   -- [1]..[4] over the LAST THREE
 *                    channels only (all of them for K <= 3): bce_dice_loss / BceDiceLoss and dice_coef_labels slice
 *                    [..., -3:] (Loss_and_metrics.py:158-159, :222-224, :240-242), [0] over every channel;
 *                    [5+2k],[6+2k]... per-class sum t*p is not needed by the reference metrics beyond
 *                    the last two channels: [5] sum t*p (ch K-2) [6] sum t (K-2) [7] sum p (K-2)
 *                    [8] sum t*p (ch K-1) [9] sum t (K-1) [10] sum p (K-1)
 *   rvip_head_grad:  dlogit[rows][K] (fp32) from pred, y_true, sums; loss scalar -> loss_out[0]
 *                    (Keras mean reduction with the GLOBAL batch: inv_count = 1/(global_rows*K) for MSE,
 *                    1/(global_rows*min(K,3)) for BCE-Dice, whose dlogit of a sliced-off background channel is 0;
 *                    local_over_global = local batch / global batch scales the replica-local dice term;
 *                    w_bce/w_dice: 0.5/1 for bce_dice_loss, 1/1 for BceDiceLoss; ignored for MSE).
 *   rvip_head_bwd:   dx[rows][Cin] = dlogit . W^T ; dW[Cin][K], db[K].
 * ------------------------------------------------------------------------------------------------ */
int rvip_head_fwd(const void* x, const float* w, const float* b, float* pred, const float* y_true,
                  float* sums, long long rows, int cin, int k, int dtype,
                  void* workspace, size_t workspace_bytes, void* stream);
int rvip_head_grad(const float* pred, const float* y_true, const float* sums, float* dlogit, float* loss_out,
                   long long rows, int k, int loss_kind, float inv_count, float local_over_global,
                   float w_bce, float w_dice, void* stream);
int rvip_head_bwd(const void* x, const float* w, const float* dlogit, void* dx, float* dw, float* db,
                  long long rows, int cin, int k, int dtype,
                  void* workspace, size_t workspace_bytes, void* stream);

/* The LAST conv stage fused with the 1x1 sigmoid head (Unets.py:128): nothing but the head reads that stage's BN output
 * y = act(scale*z + shift), so it is built in registers instead of being stored and read back three times.
 *   rvip_bn_apply_head      = rvip_bn_apply (no dropout, no pool; d->y ignored) + rvip_head_fwd in one pass over z
 *   rvip_bn_bwd_reduce_head = rvip_head_bwd's weight / bias gradients + rvip_bn_bwd_reduce, the gradient reaching y being
 *                             rebuilt per pixel as sum_k head_w[c][k] * dlogit[p][k] (d->dy ignored);
 *                             workspace: rvip_reduce_workspace(rows, 16 * c) bytes
 *   rvip_bn_bwd_apply_head  = rvip_bn_bwd_apply on the same rebuilt gradient (d->dy ignored).
 * C / VE must be a power of two <= 64 (RVIP_EUNSUPPORTED otherwise: use the separate entry points). */
int rvip_bn_apply_head(const rvip_apply_desc* d, const float* head_w, const float* head_b, int k, float* pred,
                       const float* y_true, float* sums, void* workspace, size_t workspace_bytes, void* stream);
int rvip_bn_bwd_reduce_head(const rvip_bnbwd_desc* d, const float* head_w, const float* dlogit, int k,
                            float* head_dw, float* head_db, void* stream);
int rvip_bn_bwd_apply_head(const rvip_bnbwd_desc* d, const float* head_w, const float* dlogit, int k, void* stream);

/* ABI 6 -- the MSE form of the fused last stage (the north-star loss, Unets.py:131 'mse'): the logit gradient
 *   dlogit[p][k] = 2 (pred - y_true) * inv_count * pred (1 - pred) [* dscale]
 * depends on nothing outside the pixel, so the FORWARD pass writes it (the same values, bit for bit, as rvip_head_grad followed
 * by rvip_scale_f32) and accumulates, while y is in registers, rvip_bn_apply_head_mse_rows() partial rows [rows][k_cap + 1][C]
 * (k_cap = 2):   row kk < k_cap: S_kk[c] = sum_p (y[p][c] - beta[c]) * dlogit[p][kk];   row k_cap, column kk: Q_kk = sum_p dlogit[p][kk]
 * (`beta` = the stage's BatchNormalization beta: y - beta = gamma * xhat, so no `sum g*y - beta * sum g` difference is ever formed).
 * Their column sums give the head's gradients (dW_h[c][kk] = S_kk[c] + beta[c] Q_kk, db_h[kk] = Q_kk) and, the head being linear,
 *   sum_p g[p][c] = sum_kk W_h[c][kk] Q_kk,      gamma[c] * sum_p g[p][c] xhat[p][c] = sum_kk W_h[c][kk] S_kk[c]
 * of the stage's BatchNormalization backward -- rvip_head_mse_coef folds the rows and writes dW_h, db_h, the loss value
 * (sums[0] * inv_count), dgamma, dbeta and coef[3][C] as rvip_bn_bwd_coef does, with the same ill-conditioned-block exact route
 * (over z and dlogit, as rvip_bn_bwd_reduce_head computes them): rvip_head_grad, rvip_bn_bwd_reduce_head and their finalisers
 * drop out of the step, and with them two passes over z.  Needs k <= 2, d->act == NONE (activation in front of the BN, ReLU),
 * y_true; RVIP_EUNSUPPORTED otherwise (use the entry points above). */
int rvip_bn_apply_head_mse_rows(long long rows, int c, int dtype, int k);
int rvip_bn_apply_head_mse(const rvip_apply_desc* d, const float* head_w, const float* head_b, const float* beta, int k, float* pred,
                           const float* y_true, float* sums, float* dlogit, float inv_count, float dscale,
                           float* mse_rows, size_t mse_rows_bytes, void* workspace, size_t workspace_bytes, void* stream);
typedef struct rvip_headcoef_desc {
    const struct rvip_bnbwd_desc* bn;     /* the stage's rvip_bn_bwd_apply_head descriptor (z, mean, invstd, gamma, dgamma, dbeta, coef; act RELU in front of the BN) */
    const float* beta;                    /* the same beta the forward launch was given */
    const float* head_w; const float* dlogit; int32_t k; int32_t nrows;
    const float* mse_rows;                /* [nrows][3][C] from rvip_bn_apply_head_mse */
    float*       head_dw; float* head_db;
    const float* sums; float* loss_out; float inv_count;     /* loss_out[0] = sums[0] * inv_count (NULL: not written) */
    int32_t*     flags;                   /* [ceil(c / 32)] */
    float        min_gamma, max_beta_ratio;
    /* ABI 7 -- loss_kind RVIP_LOSS_BCE_DICE: the rows are rvip_bn_apply_head_bcedice's [nrows][7][C]; the logit gradient is
     *   d = ca (p - t) + (cb t + cc) p (1 - p),   ca = w_bce inv_count dscale,  cb = -w_dice local_over_global (2 / den) dscale,
     *   cc = w_dice local_over_global ((2 I + 1) / den^2) dscale,   I = sums[2], den = sums[3] + sums[4] + 1
     * (rvip_head_grad's expression times dscale); the three coefficients are written to dcoef for rvip_bn_bwd_apply_head_lazy,
     * loss_out[0] = w_bce sums[1] inv_count - w_dice local_over_global (2 I + 1) / den; dlogit is not read (none exists: the exact
     * route rebuilds the gradient from pred / y_true).  loss_kind RVIP_LOSS_MSE (0): the fields below are ignored. */
    int32_t      loss_kind;
    float        w_bce, w_dice, local_over_global, dscale;
    const float* pred; const float* y_true;
    float*       dcoef;                   /* [3] */
} rvip_headcoef_desc;
int rvip_head_mse_coef(const rvip_headcoef_desc* d, void* stream);
/* ABI 7 -- the BCE-Dice form of the fused last stage (Loss_and_metrics.py:229-245 `bce_dice_loss`, the Train notebook's loss; also
 * the class form): the logit gradient's coefficients need the batch's Dice sums, so the forward pass keeps THREE row sets, one per
 * pixel term t0 = p - t, t1 = t p (1 - p), t2 = p (1 - p):  rows_out [rvip_bn_apply_head_mse_rows()][3 k_cap + 1][C], row
 * j k_cap + kk = sum_p (y[p][c] - beta[c]) t_j[p][kk], row 3 k_cap column j k_cap + kk = sum_p t_j[p][kk]  (k_cap = 2).  Nothing is written
 * per pixel beyond the heat-map; rvip_head_mse_coef (loss_kind BCE_DICE) combines the sets, rvip_bn_bwd_apply_head_lazy rebuilds
 * the gradient per pixel.  rvip_head_grad, rvip_scale_f32, rvip_bn_bwd_reduce_head and their finalisers drop out of the BCE-Dice
 * step as they did from the MSE step.  Same limits as rvip_bn_apply_head_mse (k <= 2: a 4-class head keeps the classic launches). */
int rvip_bn_apply_head_bcedice(const rvip_apply_desc* d, const float* head_w, const float* head_b, const float* beta, int k, float* pred,
                               const float* y_true, float* sums, float* rows_out, size_t rows_bytes,
                               void* workspace, size_t workspace_bytes, void* stream);
int rvip_bn_bwd_apply_head_lazy(const rvip_bnbwd_desc* d, const float* head_w, const float* pred, const float* y_true,
                                const float* dcoef, int k, void* stream);

/* Per-(slice, class) argmax of the heat-map, row-major first-max (north_star landmark index), and the
 * >0.5 label mask of predict_model.py:149-156.  idx_out[n][k] int64; mask_out uint8 [rows][K] or NULL. */
int rvip_landmarks(const float* pred, long long* idx_out, uint8_t* mask_out, int n, int hw, int k, float thr,
                   void* stream);

/* The reference's post-threshold of a predicted batch, on the device (the step after the hot path):
 *   flat[n][h][w] uint8   = 0, then c+1 where pred[..., c] > thr, later channels overriding earlier ones
 *                           (predict_model.py:149-156, evaluate_cv.py preds_flat)
 *   cc_filter != 0        : per slice and label only the largest 4-connected component survives, ties -> the one met
 *                           first in raster order (Postprocess.py:108-120 clean_3d_prediction_2d_cc, cv2 connectivity 4)
 *   points[n][k][2] float = mean (y, x) of the surviving pixels of label c+1, NaN when there are none
 *                           (evaluate_cv.py:418-442 get_mean_rvip_2d);  sizes[n][k] int32 = their number.
 * workspace: rvip_postprocess_workspace(n, h, w, k) bytes. */
size_t rvip_postprocess_workspace(int n, int h, int w, int k);
int rvip_postprocess(const float* pred, uint8_t* flat, float* points, int* sizes, int n, int h, int w, int k, float thr,
                     int cc_filter, void* workspace, size_t workspace_bytes, void* stream);

/* Keras Adam (ModelUtils.py:106-107; beta1 .9, beta2 .999, eps 1e-7 outside the bias correction) over a
 * flat fp32 parameter block; t = state[STEP]+1, lr = state[LR].  rvip_state_tick increments STEP. */
int rvip_adam_step(float* theta, const float* grad, float* m, float* v, long long count,
                   float beta1, float beta2, float eps, float grad_scale, const uint32_t* state, void* stream);
int rvip_state_tick(uint32_t* state, void* stream);

/* x[i] *= scale over a flat fp32 buffer.  The RVIP_F16 path uses it for static loss scaling: dlogit (rvip_head_grad) is
 * multiplied by a power of two before the backward pass and rvip_adam_step's grad_scale removes the factor again.  The
 * reference has no half-precision path (TF 2.3 float32 throughout); this is the standard mixed-precision recipe. */
int rvip_scale_f32(float* x, long long count, float scale, void* stream);

/* dtype conversion of a flat buffer (host-side plumbing: generator batches are float32). */
int rvip_convert(const void* src, int src_dtype, void* dst, int dst_dtype, long long count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RVIP_HIP_H */
