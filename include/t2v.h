/*
 * t2v.h -- C ABI of libt2v_hip.so: the MI355X (gfx950) frame-synthesis hot path of Text2Video.
 *
 * This is the drop-in boundary "B2" of SURVEY.md section 8(b).  The reference reaches its device
 * kernels through torch-0.4.1's THCUNN C ABI; each entry point below names the reference
 * interface it replaces ($SP = /root/reference/venv_vid2vid/lib/python3.7/site-packages).
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch types.  All tensor pointers are DEVICE pointers to
 *     fp32 NHWC data (channels innermost) unless stated otherwise.  "cs" arguments are the
 *     channel *storage* stride (a multiple of 4 >= the logical channel count; extra channels
 *     must hold finite values and are ignored / written as zero).
 *   - ownership: the caller allocates every device buffer (tensors, packed weights, workspace)
 *     and keeps it alive until the stream has drained.  The compute entry points never allocate device memory; a
 *     host that has no device allocator of its own gets buffers, streams and events from the "host plumbing" section
 *     at the end (ABI 14).
 *   - asynchronous: work is enqueued on the hipStream_t passed as `void* stream`
 *     (torch.cuda.current_stream().cuda_stream on the Python side); nothing synchronises.
 *   - errors: every call returns T2V_OK (0) or a negative t2v_status; t2v_last_error() returns a
 *     thread-local message.  Nothing throws across the ABI.  (THCUNN raised THError -> Python
 *     RuntimeError; the Python binding re-raises RuntimeError from the status code.)
 *   - threading: a t2v_ctx is bound to one device; use one ctx per host thread.
 */
#ifndef T2V_H_
#define T2V_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T2V_ABI_VERSION 18

typedef enum {
    T2V_OK = 0,
    T2V_ERR_INVALID = -1,   /* bad argument / unsupported shape */
    T2V_ERR_HIP = -2,       /* a HIP runtime call failed */
    T2V_ERR_WORKSPACE = -3, /* workspace too small */
    T2V_ERR_HANDOVER = -4   /* a fixed-grid kernel's accumulator hand-over timed out in an EARLIER launch: that launch's
                             * output holds NaNs; reported once, the process then runs one block per tile */
} t2v_status;

enum { T2V_PAD_ZERO = 0, T2V_PAD_REFLECT = 1 };
/* epilogue activations fused into the conv (Tanh_updateOutput THCUNN.h:1177,
 * Sigmoid_updateOutput :1098, flow x20 [SURVEY App. A.1]) */
enum { T2V_ACT_NONE = 0, T2V_ACT_TANH = 1, T2V_ACT_FLOW_W = 2 /* ch0,1: x*20 ; ch2: sigmoid */,
       T2V_ACT_LRELU = 3 /* x>0 ? x : act_scale*x  (LeakyReLU_updateOutput THCUNN.h:220) */ };

/* Convolution algorithm, all fp32.  WINOGRAD = F(2x2,3x3): input transform -> 16 batched GEMMs on the
 * implicit-GEMM kernel -> output transform fused with bias and the norm statistics (2.25x fewer MFMA
 * FLOPs than the direct conv).  WINOGRAD_F4 = F(4x4,3x3), 36 batched GEMMs, 4x fewer MFMA FLOPs,
 * interpolation points {0, +-3/4, +-3/2, inf} (rounding error ~4x the direct kernel's, 1.5e-6 of the
 * output's std per conv).  Only where t2v_conv_winograd_supported() says so; packed weights differ
 * per algorithm.
 * POLYPHASE (ABI 15) = the stride-2 3x3 convs (zero pad 1) and ConvTranspose2d(3, stride 2, pad 1, output_padding 1) as
 * polyphase Winograd F(4,2): per dimension the 2-tap sub-correlation on one sub-pixel phase as F(4,2) (5 products per 4
 * outputs), the 1-tap one as it is -- 81 batched GEMMs [tiles x Cin] x [Cin x Cout] per 4x4 (8x8) output tile instead of
 * 144 multiply-adds per channel pair: 0.5625x the MFMA FLOPs of SpatialConvolutionMM / SpatialFullDilatedConvolution
 * (THCUNN.h:664,794), points {0, +-3/4, 3/2, inf}.  Only where t2v_conv_polyphase_supported() says so; it goes through
 * the *_winograd entry points (pack_weight, workspace_floats, forward_winograd), its statistics partials through the
 * finalize entries like any producer's. */
enum { T2V_ALGO_DIRECT = 0, T2V_ALGO_WINOGRAD = 1, T2V_ALGO_WINOGRAD_F4 = 2, T2V_ALGO_POLYPHASE = 3 };

typedef struct t2v_ctx t2v_ctx;

int t2v_abi_version(void);
const char* t2v_last_error(void);
int t2v_create(t2v_ctx** out, int device);
int t2v_destroy(t2v_ctx* ctx);
/* The T2V_* environment switches are read once, at the first t2v_create; this reads them again.
 * (THCUNN's counterpart are process-global flags such as torch.backends.cudnn.benchmark,
 * $SP/torch/backends/cudnn/__init__.py:454-455.) */
void t2v_reload_env(void);
/* Errors a kernel can only report after the fact (THCUNN: THError out of a later synchronising call).  T2V_OK, or
 * T2V_ERR_HANDOVER once after a fixed-grid hand-over timed out; every entry point that can launch such a kernel makes the
 * same check first.  Callers run it after they synchronise a stream. */
int t2v_check_async_errors(void);
/* 1 while the fixed-grid ("stream-K") forms of the Winograd GEMM stage / weight-gradient reduction may be used: the
 * dispatch-order self-test of t2v_create passed and no hand-over has timed out; 0: one block per tile everywhere. */
int t2v_fixed_grid_enabled(void);
/* Overlap hint (ABI 13): on != 0 tells the library that the caller runs a SECOND stream beside the one it passes in, so that
 * kernels which leave wave slots to that stream are preferred where they exist (today: the 512x512 ResnetBlock GEMM stage on
 * 256x128 tiles with one block per CU instead of 128x128 with two -- slower alone, faster in a two-stream frame).  Per calling
 * thread; returns the previous value.  t2v_generator_forward[_batch] sets it for its own launches; a caller that overlaps
 * single-op calls itself (or wants t2v_conv_winograd_gemm_form to answer for the generator's frames) sets it explicitly.
 * on == 2 (ABI 18): the second stream carries fixed-grid GEMMs of its OWN (the train step's weight gradients beside its data
 * gradients): the whole-tile fixed-grid GEMM and the Winograd-domain weight gradient then launch ONE 128x128 block per CU
 * (half a CU's LDS and registers each) instead of two, so that the two streams' launches -- and the bandwidth-bound kernels
 * between two GEMMs -- are resident side by side instead of queueing behind a grid that keeps every CU full until its last
 * block leaves.  The tile forms chosen are those of on == 0; the results carry the same bits.  T2V_SK_BLOCKS_PER_CU=1 / 2
 * forces one / two per CU whatever the hint.
 * (THCUNN has no counterpart: its ops run on the one current stream of THCState.) */
int t2v_set_overlap_hint(int on);
/* Test hook: raise != 0 sets the sticky error word exactly as a timed-out consumer wave would; raise == 0 clears it and
 * switches the fixed-grid kernels back on. */
void t2v_debug_async_error(int raise);

/* ------------------------------------------------------------------------------------------
 * Convolution.  Replaces SpatialReflectionPadding_updateOutput (THCUNN.h:952) +
 * SpatialConvolutionMM_updateOutput (THCUNN.h:664) [Conv2d.forward $SP/torch/nn/modules/conv.py:
 * 299-300] and SpatialFullDilatedConvolution_updateOutput (THCUNN.h:794) [ConvTranspose2d.forward
 * conv.py:687-691].  Reflection padding is resolved inside the kernel's tile loader.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int H, W;        /* input spatial size */
    int Cin, Cout;   /* logical channels */
    int kH, kW;      /* kernel */
    int stride;      /* 1 or 2 */
    int pad;         /* padding on each side (reflect: the ReflectionPad2d amount) */
    int pad_mode;    /* T2V_PAD_ZERO | T2V_PAD_REFLECT */
    int transposed;  /* 1: ConvTranspose2d(k in {3,4}, stride=2, padding, output_padding) */
    int act;         /* T2V_ACT_* applied after bias */
    float act_scale; /* T2V_ACT_FLOW_W: flow multiplier (20 * 2^scale); T2V_ACT_LRELU: negative slope */
    int output_padding; /* transposed only (0 or 1); the generator's up-convs use 1 */
    int algo;           /* T2V_ALGO_DIRECT | T2V_ALGO_WINOGRAD (3x3 stride-1 reflect-pad-1 convs only) */
} t2v_conv_desc;

/* output spatial size */
int t2v_conv_out_dims(const t2v_conv_desc* d, int* Hout, int* Wout);
/* number of floats of the packed weight buffer (includes all zero padding) */
size_t t2v_conv_packed_weight_floats(const t2v_conv_desc* d, int x_cs);
/* repack a torch-layout weight ([Cout,Cin,kH,kW], or [Cin,Cout,3,3] when transposed -- conv.py:
 * 28-33) that already lives on the device into the kernel's K-contiguous layout. */
int t2v_conv_pack_weight(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int x_cs,
                         const float* w_torch_dev, float* packed_dev);
/* The same for the DATA-GRADIENT conv `d` of a stride-1 forward layer, given that layer's own torch weight
 * [Cout_f = d->Cin][Cin_f = d->Cout][k][k]: the 180-degree flip and the in/out transpose of updateGradInput's filter are
 * folded into the gather (no flipped / transposed copy of the weight is ever made). */
int t2v_conv_pack_weight_adjoint(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int x_cs,
                                 const float* w_forward_torch_dev, float* packed_dev);
/* number of floats of the per-tile instance-norm partial-statistics buffer for this conv */
size_t t2v_conv_stats_floats(const t2v_conv_desc* d);
/* y = act(conv(x) + bias).  If stats_partial != NULL the epilogue also emits per-(tile,channel)
 * (mean, M2) partials over the block's pixels for the fused instance norm. */
int t2v_conv2d_forward(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, const float* x, int x_cs,
                       const float* w_packed, const float* bias, float* y, int y_cs,
                       float* stats_partial);
/* The same for `batch` images in ONE launch (ABI 14; direct algorithm): x = [batch][H][W][x_cs], y = [batch][Hout][Wout][y_cs],
 * image b's statistics partials at stats_partial + b * t2v_conv_stats_floats(d).  Every image's result is the single-image
 * call's, bit for bit.  (SpatialConvolutionMM_updateOutput loops over the batch with one GEMM per image, THCUNN.h:664; here the
 * images share a launch because the discriminators' layers of the train step fill a fraction of the chip each.) */
int t2v_conv2d_forward_batch(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int batch, const float* x, int x_cs,
                             const float* w_packed, const float* bias, float* y, int y_cs, float* stats_partial);

/* Bit mask of the Winograd variants `d` (algo ignored) can run as: 1 = F(2x2,3x3), 2 = F(4x4,3x3).
 * Needs 3x3, stride 1, ReflectionPad 1 or zero padding 0..2 (pad 2: the data gradient of the pad-1 conv),
 * Cin % 32 == 0 == x_cs, Cout % 4 == 0, no activation.  Any H, W >= 2:
 * the ceil(H/m) x ceil(W/m) tile grid (m = 2 | 4) is ragged at the bottom / right edge and padded with
 * empty tiles to a multiple of 128 (extra GEMM rows, masked in the output transform). */
int t2v_conv_winograd_supported(const t2v_conv_desc* d, int x_cs);
/* bit 0: T2V_ALGO_POLYPHASE applies to `d` (d->algo ignored); bit 1: ... and is the form the library itself selects (both
 * channel counts >= 256, >= 128 tiles: where it measured faster than the implicit-GEMM kernel).  Applies to: 3x3, stride 2; a conv with zero padding 1 and even H, W, or a
 * transposed conv with pad 1 / output_padding 1; x_cs == Cin, Cin % 32 == 0, Cout % 128 == 0, no activation.  The tile grid
 * (4x4 outputs | 4x4 inputs) is ragged at the bottom / right edge and padded with empty tiles, as for the Winograd forms.  t2v_generator_layer_desc() selects it where it measured faster. */
int t2v_conv_polyphase_supported(const t2v_conv_desc* d, int x_cs);
/* The algorithm the library itself would pick for `d` (d->algo ignored): the one with the fewest GEMM rows among
 * direct (9 per output pixel), F(2x2,3x3) and F(4x4,3x3) (16 | 36 per tile, tile count padded to 128).
 * cap: 0 = any, 1 = direct only, 2 = at most F(2x2,3x3).  Returns a T2V_ALGO_* value. */
int t2v_conv_best_algo(const t2v_conv_desc* d, int x_cs, int cap);
/* floats of scratch a Winograd forward needs: transformed input V + transformed output M, and for F(4x4,3x3) the
 * hand-over area of the fixed-grid GEMM stage (blocks that share a 128x128 tile pass accumulators through it; any
 * content on entry, one workspace per conv in flight) */
size_t t2v_conv_winograd_workspace_floats(const t2v_conv_desc* d, int x_cs);
/* GEMM rows one image contributes per F(4x4,3x3) transform position: its ceil(H/4) x ceil(W/4) tiles padded to the
 * 64 / 128-row granule (the slot pitch of the batch-wide tile lists in the weight-gradient workspace) */
int t2v_conv_winograd_tile_rows(const t2v_conv_desc* d);
/* Which form the batched GEMM stage of an F(4x4,3x3) conv over `nimg` images takes under the current switches (reporting
 * only: bench.py's roofline label): one block per tile, or a fixed grid of resident blocks that hand accumulators over where
 * a tile is cut -- on 128x128 tiles, 192x64 tiles, 160x128 tiles with one block per CU (129..160 tile rows: the reference's
 * 512x320 frames), or ragged M tiles of 4,..,4,r 32-row fragments.  -1: not an F(4x4,3x3) conv. */
enum { T2V_GEMM_TILE_PER_BLOCK_64x64 = 0, T2V_GEMM_TILE_PER_BLOCK_128x128 = 1, T2V_GEMM_FIXED_GRID_128x128 = 2,
       T2V_GEMM_FIXED_GRID_192x64 = 3, T2V_GEMM_FIXED_GRID_160x128 = 4, T2V_GEMM_FIXED_GRID_RAGGED = 5,
       T2V_GEMM_FIXED_GRID_256x128 = 6 /* one block per CU, like 160x128 */,
       T2V_GEMM_FIXED_GRID_RAGGED_TALL = 7 /* ragged, balanced tiles of 3..6 fragments, one block per CU */ };
int t2v_conv_winograd_gemm_form(const t2v_conv_desc* d, int nimg);
/* forward with d->algo == T2V_ALGO_WINOGRAD | T2V_ALGO_WINOGRAD_F4; same contract as t2v_conv2d_forward plus the workspace */
int t2v_conv2d_forward_winograd(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, const float* x, int x_cs,
                                const float* w_packed, const float* bias, float* y, int y_cs, float* stats_partial,
                                float* workspace);
/* the same, one stage at a time (measurement and tests): `stages` is a bit mask of
 * 1 = input transform x -> V, 2 = the 16 | 36 batched GEMMs V,U -> M, 4 = output transform M -> y (+bias, stats). */
int t2v_conv2d_forward_winograd_stages(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, const float* x, int x_cs,
                                       const float* w_packed, const float* bias, float* y, int y_cs,
                                       float* stats_partial, float* workspace, int stages);

/* ------------------------------------------------------------------------------------------
 * Instance norm (+affine) + ReLU + residual.  Replaces BatchNormalization_updateOutput(train)
 * (THCUNN.h:33) as reached from F.instance_norm ($SP/torch/nn/functional.py:1258-1301) and
 * Threshold_updateOutput (THCUNN.h:1328).  Biased variance, eps inside the sqrt.
 *   finalize: combines the conv epilogue's partials (Chan et al.) -> mean_rstd[C][2]
 *   apply:    y = act((x-mean)*rstd*gamma+beta) + res1 + res2   (gamma/beta/res* nullable,
 *             y may alias x); relu: 0 none, 1 ReLU, 2 LeakyReLU(0.2) (the discriminators)
 *   Batch statistics (BatchNorm2d in train mode over a batch of B images, the discriminators):
 *   run the conv per image with stats_partial + b*t2v_conv_stats_floats(d), then finalize_batch.
 * ------------------------------------------------------------------------------------------ */
int t2v_instance_norm_finalize(t2v_ctx* ctx, void* stream, const t2v_conv_desc* producer,
                               const float* stats_partial, float eps, float* mean_rstd);
int t2v_batch_norm_finalize(t2v_ctx* ctx, void* stream, const t2v_conv_desc* producer, int batch,
                            const float* stats_partial, float eps, float* mean_rstd);
/* BatchNorm2d's running statistics (training mode; $SP/torch/nn/modules/batchnorm.py:27-29,57-64; the momentum
 * argument of BatchNormalization_updateOutput, THCUNN.h:33-45): running_mean / running_var [C] are updated in place
 * from the (mean, rstd) table a finalize call produced over n values per channel:
 *   running_mean = (1-momentum)*running_mean + momentum*mean
 *   running_var  = (1-momentum)*running_var  + momentum*var*n/(n-1)      (unbiased; var = 1/rstd^2 - eps) */
int t2v_batch_norm_update_running(t2v_ctx* ctx, void* stream, const float* mean_rstd, float* running_mean,
                                  float* running_var, long n, int C, float momentum, float eps);
/* t2v_batch_norm_finalize + `times` running-statistics updates (n = batch * output pixels of `producer`) in the same
 * launch; batch = 1: an instance norm standing for BatchNorm2d(train) on a batch of one */
int t2v_batch_norm_finalize_running(t2v_ctx* ctx, void* stream, const t2v_conv_desc* producer, int batch,
                                    const float* stats_partial, float eps, float* mean_rstd, float* running_mean,
                                    float* running_var, float momentum, int times);
int t2v_instance_norm_apply(t2v_ctx* ctx, void* stream, const float* x, const float* mean_rstd,
                            const float* gamma, const float* beta, const float* res1,
                            const float* res2, float* y, long npix, int C, int relu);

/* ------------------------------------------------------------------------------------------
 * Flow-warp compositor.  Replaces SpatialGridSamplerBilinear_updateOutput (THCUNN.h:1048; F.
 * grid_sample functional.py:2046-2093, corner aligned, border padding) plus the blend
 * out = raw*w + warp(prev, flow)*(1-w) [SURVEY App. A.1].  fw = [H,W,3] (flow_x, flow_y in
 * pixels, weight).  prev: NHWC with storage stride prev_cs, the 3 channels starting at prev_c0.
 * warp_out (nullable) receives the warped image.  raw == NULL: plain `resample(prev, flow)` (the
 * train step's warp losses); the warped image goes to warp_out and, if given, out.
 *
 * _backward replaces SpatialGridSamplerBilinear_updateGradInput (THCUNN.h:1055: gradInput AND
 * gradGrid) fused with the blend's adjoint.  d_out: gradient of `out`, d_warp: gradient of
 * `warp_out` (either may be NULL, not both; d_out needs raw).  Writes d_raw [H,W,4] (nullable) =
 * d_out*w, d_fw [H,W,4] = (d flow_x, d flow_y, d weight, 0), and -- only if d_prev != NULL --
 * ACCUMULATES the image gradient into d_prev (same layout / channels as prev; the caller zero-fills;
 * atomic adds, so its summation order is not fixed).  Border rule as torch 0.4.1's kernel: bilinear
 * weights from the unclipped position, corner indices clipped into the image => the gradient with
 * respect to a coordinate is zero outside the image and exactly on its last row / column.
 * ------------------------------------------------------------------------------------------ */
int t2v_flow_warp_composite(t2v_ctx* ctx, void* stream, const float* raw, const float* fw,
                            const float* prev, int prev_cs, int prev_c0, float* out, float* warp_out,
                            int H, int W);
int t2v_flow_warp_composite_backward(t2v_ctx* ctx, void* stream, const float* d_out, const float* d_warp,
                                     const float* raw, const float* fw, const float* prev, int prev_cs,
                                     int prev_c0, float* d_raw, float* d_fw, float* d_prev, int H, int W);

/* AvgPool2d(3, stride 2, padding 1, count_include_pad=False) -- SpatialAveragePooling_
 * updateOutput (THCUNN.h:579; pooling.py:536-543).  NHWC, C % 4 == 0 not required. */
int t2v_avgpool3x3s2(t2v_ctx* ctx, void* stream, const float* x, float* y, int H, int W, int C);

/* ------------------------------------------------------------------------------------------
 * Convolution backward (accGradParameters of SpatialConvolutionMM / SpatialFullDilatedConvolution,
 * THCUNN.h:664,794).  `d` is the FORWARD conv descriptor.
 *   backward_weight: dW (packed layout, same as the packed forward weight) = sum over the batch and
 *       all pixels of dy (x) x through every filter tap; accumulate != 0 adds to dw_packed.
 *       Narrow high-resolution layers cut the pixel reduction into ranges (deterministic partials in
 *       `workspace`, summed in a fixed order: up to 4 partials by the last-arriving block of each (tap, n, c)
 *       tile inside the launch -- arrival tickets, write-through partials -- more by a reduce pass; the same
 *       bits either way).
 *       x: [batch][H][W][x_cs], dy: [batch][Hout][Wout][dy_cs].
 *   unpack_weight:   packed -> torch layout (inverse of t2v_conv_pack_weight).
 *   channel_sum:     out[c] = sum over pixels of x[.][c]  (bias gradient).
 * The data gradient needs no entry point of its own: it is a forward convolution with the adjoint
 * geometry (stride-2 conv <-> ConvTranspose with the same torch-layout weight; stride-1: flipped /
 * transposed weight, padding k-1-p) -- text2video_amd/backward.py builds those descriptors.
 * ------------------------------------------------------------------------------------------ */
size_t t2v_conv_backward_weight_workspace_floats(const t2v_conv_desc* d, int x_cs, int batch);
int t2v_conv2d_backward_weight(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int batch, const float* x,
                               int x_cs, const float* dy, int dy_cs, float* dw_packed, int accumulate,
                               float* workspace /* NULL when ..._workspace_floats() == 0 */);
/* The same with the images of the batch `x_img_stride` / `dy_img_stride` floats apart instead of contiguous (ABI 14): the two
 * frames of a training clip live in buffers of their own, and one launch over both runs its blocks twice as long as two
 * launches (the kernel's fixed cost per block is ~8 of a 512<->1024 layer's 36 stage times per frame: 0.64 -> 0.73 of peak).
 * Any non-zero multiple of 4 bytes, negative included.  Only where t2v_conv_backward_weight_strided_supported() says so
 * (zero padding, no folded taps, GEMM rows that are whole 16-pixel stages: the stride-2 and transposed layers); workspace as
 * for t2v_conv2d_backward_weight at the same batch.  The result is that call's on the contiguous batch, bit for bit. */
int t2v_conv_backward_weight_strided_supported(const t2v_conv_desc* d, int x_cs, int dy_cs);
int t2v_conv2d_backward_weight_strided(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int batch, const float* x,
                                       int x_cs, long x_img_stride, const float* dy, int dy_cs, long dy_img_stride,
                                       float* dw_packed, int accumulate, float* workspace);
/* Weight gradient of a 3x3 stride-1 conv in the Winograd domain, F(4x4,3x3): dU[xi] = sum over tiles of
 * (A dy A^T)[xi] x (B^T x B)[xi] -- 36 pixel-reduction GEMMs of a quarter of the direct gradient's FLOPs -- then
 * dW = G^T dU G, written in TORCH layout [Cout][Cin][3][3] (accumulate: += ).  Where
 * t2v_conv_backward_weight_winograd_supported() says so (the forward conditions of F(4x4,3x3) plus dy_cs == Cout). */
int t2v_conv_backward_weight_winograd_supported(const t2v_conv_desc* d, int x_cs, int dy_cs);
size_t t2v_conv_backward_weight_winograd_workspace_floats(const t2v_conv_desc* d, int x_cs, int batch);
int t2v_conv2d_backward_weight_winograd(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int batch, const float* x,
                                        int x_cs, const float* dy, int dy_cs, float* dw_torch, int accumulate,
                                        float* workspace);
/* The same in two stages, for callers that meet the images of one reduction at different times (the frames of a
 * training clip run through the same layer one after the other, and so do their backward passes): stages & 1
 * transforms images [b0, b0+nb) (x, dy point at image b0) into their slots of a workspace sized for `batch`
 * images; stages & 2 reduces over all `batch` slots and writes dW.  One reduction over K = batch * tiles instead
 * of `batch` short ones: the per-block fixed work of the 36 GEMMs and the filter transform are paid once. */
int t2v_conv2d_backward_weight_winograd_stages(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int batch, int b0,
                                               int nb, const float* x, int x_cs, const float* dy, int dy_cs,
                                               float* dw_torch, int accumulate, float* workspace, int stages);
/* The forward pass of such a layer in a train step (ABI 16): t2v_conv2d_forward_winograd (algo T2V_ALGO_WINOGRAD_F4, one image)
 * with V = B^T d B written into slot `slot` of the weight gradient's batch-wide workspace (`wgrad_workspace` sized by
 * t2v_conv_backward_weight_winograd_workspace_floats(d, x_cs, batch)) and read by the forward GEMM from there.  The backward
 * pass then calls _stages(stages & 1) with x == NULL for that image: only A dy A^T is transformed, the input transform of the
 * forward pass is not repeated (updateOutput and accGradParameters of THCUNN.h:664 share their im2col-equivalent). */
/* A dy A^T of one image into slot `slot`, for a layer followed by a norm (+ ReLU): `dy` is the gradient BEHIND the norm,
 * `conv_out` the layer's raw output, `sums` the (sum g, sum g*xhat) of t2v_instance_norm_backward[_affine] called with
 * dx == NULL.  The gradient in front of the norm (BatchNormalization_backward's gradInput, THCUNN.h:47) is formed per
 * loaded element -- inorm backward's arithmetic in its order, bit for bit -- and never stored: where the data gradient reads
 * A dy A^T from this workspace (t2v_conv2d_backward_data_winograd) nothing else needs it. */
int t2v_conv2d_backward_weight_winograd_dy_norm(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int batch, int slot,
                                                int x_cs, const float* conv_out, const float* dy, const float* mean_rstd,
                                                const float* gamma, const float* beta, int relu, const float* sums,
                                                float* workspace);
int t2v_conv2d_forward_winograd_keep_v(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, const float* x, int x_cs,
                                       const float* w_packed, const float* bias, float* y, int y_cs, float* stats_partial,
                                       float* workspace, float* wgrad_workspace, int batch, int slot);

/* Data gradient of a 3x3 stride-1 ReflectionPad(1) conv by the TRANSPOSED Winograd algorithm (updateGradInput of
 * SpatialConvolutionMM + SpatialReflectionPadding_updateGradInput, THCUNN.h:664,952): forward is V = B^T d B, M = U V,
 * Y = A^T M A; its transpose is dM = A dy A^T -- the tensor the Winograd-domain weight gradient builds anyway, read here
 * out of ITS workspace (slot `slot` of `batch`, after t2v_conv2d_backward_weight_winograd_stages(stages & 1)) -- then
 * dV = U^T dM (36 GEMMs over the layer's own T tiles, against (H+2)(W+2)/16 for the full-correlation form), the 6x6 patches
 * B dV B^T overlap-added into the padded gradient and folded by the reflect-pad adjoint.  `d` is the FORWARD descriptor;
 * H, W multiples of 4; ut_packed from t2v_conv_pack_weight_transposed (the forward weight's transform, transposed, no flip). */
int t2v_conv_backward_data_winograd_supported(const t2v_conv_desc* d, int x_cs, int dy_cs);
size_t t2v_conv_backward_data_winograd_weight_floats(const t2v_conv_desc* d, int x_cs);
size_t t2v_conv_backward_data_winograd_scratch_floats(const t2v_conv_desc* d, int x_cs);
int t2v_conv_pack_weight_transposed(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int x_cs, const float* w_forward_dev,
                                    float* packed_dev);
int t2v_conv2d_backward_data_winograd(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int batch, int slot,
                                      const float* wgrad_workspace, int x_cs, const float* ut_packed, float* scratch, float* dx);
/* The same with the FORWARD layer's own packed weights (t2v_conv_pack_weight of `d` with algo T2V_ALGO_WINOGRAD_F4 at the
 * same x_cs): U [36][Cout][x_cs] is the [K][N] operand of dV = dM U as it lies; the fixed-grid GEMM reads it in place (two
 * k rows of 128 columns per LDS-DMA instruction, ds_read_b64 per k), the same MFMA chains as with the transposed copy and so
 * the same bits -- no t2v_conv_pack_weight_transposed per optimiser step and no second copy of the transformed weights.
 * _takes_forward_weights: 1 where that form exists (whole 128 x 128 tiles on the fixed grid, Cout % 128 == 0, x_cs % 128 == 0). */
int t2v_conv_backward_data_winograd_takes_forward_weights(const t2v_conv_desc* d, int x_cs, int dy_cs);
int t2v_conv2d_backward_data_winograd_fw(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int batch, int slot,
                                         const float* wgrad_workspace, int x_cs, const float* u_forward_packed, float* scratch,
                                         float* dx);

int t2v_conv_unpack_weight(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int x_cs, const float* packed_dev,
                           float* w_torch_dev);
/* unpack_weight with a destination that may already hold a gradient: accumulate != 0 adds (w_torch += unpacked).  The
 * backward nodes of a layer used several times per step deliver straight into the parameter's slice of a flat
 * exchange bucket: the first writes, the later ones add -- what autograd's AccumulateGrad would do with an extra pass. */
int t2v_conv_unpack_weight_into(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int x_cs, const float* packed_dev,
                                float* w_torch_dev, int accumulate);
/* dst = src (overwrite != 0) or dst += src; x *= s; zero-fill: the small-tensor ends of the gradient path (bias / affine
 * gradients into their bucket slice, averaging a summed bucket where the collective cannot, clearing the slice of a
 * parameter no backward node reached).  Replace ATen's add / mul / fill on the product path. */
int t2v_accumulate(t2v_ctx* ctx, void* stream, float* dst, const float* src, long n, int overwrite);
int t2v_scale(t2v_ctx* ctx, void* stream, float* x, long n, float s);
int t2v_zero(t2v_ctx* ctx, void* stream, void* ptr, size_t bytes);
/* src [C][2] (t2v_instance_norm_backward's dbeta_dgamma) -> dst0[c] (+)= src[c][0], dst1[c] (+)= src[c][1] */
int t2v_unzip2(t2v_ctx* ctx, void* stream, const float* src, float* dst0, float* dst1, int C, int overwrite);
int t2v_channel_sum(t2v_ctx* ctx, void* stream, const float* x, long npix, int C, int cs,
                    float* scratch /* >= 256*C floats */, float* out);
/* adjoint of ReflectionPad2d(pad): dxp [H+2pad][W+2pad][C] -> dx [H][W][C]  (SpatialReflectionPadding_updateGradInput) */
int t2v_reflect_pad_backward(t2v_ctx* ctx, void* stream, const float* dxp, float* dx, int H, int W, int C, int pad);
/* BatchNormalization_backward(train) / instance norm backward fused with the activation derivative:
 *   g = dy * act'(gamma*xhat+beta); dx = rstd*gamma*(g - mean(g) - xhat*mean(g*xhat)); dbeta_dgamma[c] = (sum g, sum g*xhat)
 * x = the conv output that was normalised, npix = pixels in the statistics (all images of a batch-norm batch),
 * relu as in t2v_instance_norm_apply, C % 4 == 0, scratch >= 128*C*2 floats.  dx may be NULL (ABI 16): the sums only. */
int t2v_instance_norm_backward(t2v_ctx* ctx, void* stream, const float* x, const float* dy, const float* mean_rstd,
                               const float* gamma, const float* beta, int relu, long npix, int C, float* scratch,
                               float* dx, float* dbeta_dgamma);
/* ... the same, with the two sums also delivered into the affine parameters' gradient tensors: d_beta[c] (+)= sum g,
 * d_gamma[c] (+)= sum g*xhat (overwrite != 0: written; else added) -- BatchNormalization_backward's gradBias / gradWeight
 * accumulation (THCUNN.h:47-62) without a pass of its own */
int t2v_instance_norm_backward_affine(t2v_ctx* ctx, void* stream, const float* x, const float* dy, const float* mean_rstd,
                                      const float* gamma, const float* beta, int relu, long npix, int C, float* scratch,
                                      float* dx, float* dbeta_dgamma, float* d_beta, float* d_gamma, int overwrite);
/* dpre = dy * act'(.) from the activation OUTPUT y; act: T2V_ACT_TANH, 2 = sigmoid, T2V_ACT_LRELU (slope), 0 = scale by slope,
 * 4 = the fused flow / weight head (T2V_ACT_FLOW_W) on [.,4] storage: ch 0,1 scale by slope, ch 2 sigmoid, ch 3 zero */
int t2v_act_backward(t2v_ctx* ctx, void* stream, const float* dy, const float* y, int act, float slope, long n,
                     float* dpre);
int t2v_avgpool3x3s2_backward(t2v_ctx* ctx, void* stream, const float* dy, float* dx, int H, int W, int C);
/* MaxPool2d(2,2), floor mode (torchvision vgg19 `features`, $SP/torchvision/models/vgg.py:82 cfg 'E' -- the VGG
 * perceptual loss): x [H][W][C] -> y [H/2][W/2][C]; backward routes dy to the first maximum of
 * each window in row-major order and writes zeros elsewhere. */
int t2v_maxpool2x2(t2v_ctx* ctx, void* stream, const float* x, float* y, int H, int W, int C);
int t2v_maxpool2x2_backward(t2v_ctx* ctx, void* stream, const float* x, const float* dy, float* dx, int H, int W, int C);
/* gradients of scale*sum((x-c)^2) and scale*sum|a-b| (wrt x / a) */
int t2v_sum_sq_diff_const_backward(t2v_ctx* ctx, void* stream, const float* x, float c, float scale, long n, float* dx);
int t2v_sum_abs_diff_backward(t2v_ctx* ctx, void* stream, const float* a, const float* b, float scale, long n, float* da);

/* ------------------------------------------------------------------------------------------
 * Train-step scalars.
 *   sum_sq_diff_const : sum (x - c)^2      -> out[0]   (MSECriterion_updateOutput THCUNN.h:356,
 *                       LSGAN target 1/0; the caller divides by n for 'elementwise_mean')
 *   sum_abs_diff      : sum |a - b|        -> out[0]   (AbsCriterion_updateOutput THCUNN.h:18)
 *   adam_step         : torch.optim.Adam.step ($SP/torch/optim/adam.py:48-98), one fused pass:
 *                       m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ;
 *                       p -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)
 * `scratch` must hold >= 2048 floats.  Deterministic two-level reductions (no atomics).
 * ------------------------------------------------------------------------------------------ */
int t2v_sum_sq_diff_const(t2v_ctx* ctx, void* stream, const float* x, float c, long n, float* scratch, float* out);
int t2v_sum_abs_diff(t2v_ctx* ctx, void* stream, const float* a, const float* b, long n, float* scratch, float* out);
/* sum over pixels and the channels [c0, c0+C) of mask[pix] * |a - b| on [npix][cs] tensors (b NULL: zero target; mask
 * NULL: all ones) -- the numerator of vid2vid's MaskedL1Loss (flow, warp and weight losses; AbsCriterion THCUNN.h:18
 * on the masked operands); _backward: da = scale * mask * sign(a - b), zero in all other channels. */
int t2v_sum_abs_diff_masked(t2v_ctx* ctx, void* stream, const float* a, const float* b, const float* mask, long npix,
                            int c0, int C, int cs, float* scratch, float* out);
int t2v_sum_abs_diff_masked_backward(t2v_ctx* ctx, void* stream, const float* a, const float* b, const float* mask,
                                     float scale, long npix, int c0, int C, int cs, float* da);
/* All scalar loss terms of a train step in one launch (MSECriterion / AbsCriterion _updateOutput AND _updateGradInput,
 * THCUNN.h:356,365,18,27, for every term at once).  Term t is described by three device tables:
 *   term_ptrs[4t..]   = a, b, seed (device addresses; b / seed may be 0), n (op 0: pixels; op 1, 2: floats)
 *   term_ints[2t..]   = op, cs     op 0: sum (a[i*cs] - c)^2 over the pixels, seed[i*cs] = 2 s (a - c), pad channels 0
 *                                  op 1: sum |a - b|, seed = s sign(a - b);   op 2: no value, seed = 0
 *   term_floats[3t..] = c, s (seed scale), v (value scale: out[t] = v * sum)
 * Block b of the first kernel reduces `chunk` (a multiple of 4) elements of term chunk_term[b] starting at chunk_off[b] into
 * partials[b]; the chunks of a term are consecutive, term_chunk0[t] .. term_chunk0[t+1].  Deterministic (no atomics). */
int t2v_loss_terms(t2v_ctx* ctx, void* stream, const int64_t* term_ptrs, const int32_t* term_ints, const float* term_floats,
                   const int32_t* chunk_term, const int64_t* chunk_off, const int32_t* term_chunk0, int nterms, int nchunks,
                   int chunk, float* partials, float* out);
/* lr, betas and eps are doubles, as adam.py holds them: 1-beta, the bias corrections and the step size are evaluated
 * in double and rounded to fp32 once (adam.py:86-96 does the same through Python floats). */
int t2v_adam_step(t2v_ctx* ctx, void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                  long n, double lr, double beta1, double beta2, double eps, int step);

/* The same update for many tensors in ONE launch (all tables are DEVICE arrays): ptrs[t] = {param, grad, exp_avg,
 * exp_avg_sq} addresses of tensor t (grad 0: the tensor is skipped, as Adam skips parameters without gradient),
 * nelem[t] its length, step_size[t] = lr*sqrt(1-beta2^k)/(1-beta1^k) with that tensor's own step count k; the work is
 * cut into `nchunks` chunks of `chunk` elements: chunk c covers [chunk_off[c], chunk_off[c]+chunk) of tensor chunk_tensor[c]. */
int t2v_adam_step_multi(t2v_ctx* ctx, void* stream, const int64_t* ptrs, const int64_t* nelem, const float* step_size,
                        const int32_t* chunk_tensor, const int64_t* chunk_off, int nchunks, int chunk, double beta1,
                        double beta2, double eps);

/* layout / dtype plumbing on the device */
int t2v_nchw_to_nhwc(t2v_ctx* ctx, void* stream, const float* src, float* dst, int C, int H, int W, int dst_cs);
int t2v_nhwc_to_nchw(t2v_ctx* ctx, void* stream, const float* src, float* dst, int C, int H, int W, int src_cs);
/* ToTensor + Normalize(.5,.5) ($SP/torchvision/transforms/functional.py:38-60,206-208) of a
 * uint8 HWC(3) pose map into channels [c0,c0+3) of an NHWC fp32 buffer */
int t2v_pose_u8_to_f32(t2v_ctx* ctx, void* stream, const uint8_t* src_hwc3, float* dst, long npix, int dst_cs, int c0);
/* util.tensor2im [SURVEY 3.2]: uint8((x+1)/2*255 clipped), same layout */
int t2v_tensor2im_u8(t2v_ctx* ctx, void* stream, const float* x, uint8_t* y, long n);
int t2v_copy_channels(t2v_ctx* ctx, void* stream, const float* src, int src_cs, int src_c0, float* dst,
                      int dst_cs, int dst_c0, int nc, long npix);

/* ------------------------------------------------------------------------------------------
 * Whole-frame generator.  Replaces CompositeGenerator.forward / CompositeLocalGenerator.forward
 * [SURVEY App. A.1/A.2] as called by Vid2VidModelG.generate_frame_infer.
 *
 * Layer order of `layers` (conv layers only; gamma/beta are the affine params of the norm that
 * follows the conv, NULL for InstanceNorm(affine=False) and for the heads):
 *   global (is_local=0):
 *     down_seg: c7, d x n_downsample, RB x (n_blocks - n_blocks/2) [2 convs each]
 *     down_img: same
 *     res_img : RB x (n_blocks/2)        up_img: u x n_downsample       final_img: c7
 *     if !no_flow: res_flow RB x (n_blocks/2), up_flow u x n_downsample,
 *                  final_flow_w: ONE c7 with 3 outputs = cat(model_final_flow, model_final_w)
 *   local (is_local=1, ngf = ngf of this scale):
 *     down_seg: c7, d      down_img: c7, d
 *     up_img: RB x n_blocks, u         final_img: c7
 *     if !no_flow: up_flow: RB x n_blocks, u ; final_flow_w
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int H, W;           /* multiples of 2^n_downsample */
    int input_nc;       /* pose channels x n_frames_G (9)   -> storage round_up4 (12) */
    int prev_nc;        /* (n_frames_G-1) x output_nc (6)   -> storage round_up4 (8)  */
    int output_nc;      /* 3 */
    int ngf;
    int n_downsample;
    int n_blocks;
    int no_flow;
    int norm_affine;    /* 1: BatchNorm2d(affine) in train mode, N=1  ==  IN + gamma/beta */
    int is_local;       /* 0: CompositeGenerator, 1: CompositeLocalGenerator */
    float flow_multiplier; /* 20 * 2^scale */
    float eps;          /* 1e-5 */
    int conv_algo;      /* ResnetBlock convs: 0 = best supported of F(4x4,3x3) > F(2x2,3x3) > direct;
                         * 1 = direct only; 2 = F(2x2,3x3) or direct */
} t2v_gen_desc;

typedef struct {
    const float* w;      /* packed by t2v_conv_pack_weight */
    const float* bias;   /* [Cout] */
    const float* gamma;  /* [Cout] or NULL */
    const float* beta;   /* [Cout] or NULL */
} t2v_layer;

typedef struct {
    const float* pose;              /* [H,W,round_up4(input_nc)] */
    const float* prev;              /* [H,W,round_up4(prev_nc)]  */
    const float* coarse_img_feat;   /* local only: [H/2,W/2,2*ngf] */
    const float* coarse_flow_feat;  /* local only, !no_flow */
    int use_raw_only;
    float* out;        /* [H,W,4] final frame, channels 0..2 (channel 3 = 0) */
    float* raw;        /* nullable [H,W,4] img_raw */
    float* flow_w;     /* nullable [H,W,4] flow_x, flow_y, weight */
    float* img_feat;   /* nullable [H,W,ngf] */
    float* flow_feat;  /* nullable [H,W,ngf] */
} t2v_gen_io;

int t2v_generator_num_layers(const t2v_gen_desc* d);
/* conv descriptor of layer i (what to pack weights with -- its `algo` depends on d->H, d->W and
 * d->conv_algo) and its input storage stride */
int t2v_generator_layer_desc(const t2v_gen_desc* d, int i, t2v_conv_desc* out, int* x_cs);
size_t t2v_generator_workspace_bytes(const t2v_gen_desc* d);
int t2v_generator_forward(t2v_ctx* ctx, void* stream, const t2v_gen_desc* d, const t2v_layer* layers,
                          int n_layers, const t2v_gen_io* io, void* workspace, size_t ws_bytes);

/* `batch` INDEPENDENT recurrences advanced in lock-step on one GPU: the frames of `batch` sequences (the reference
 * always generates two per utterance -- tmp and tmp_smooth, text2video_audio.sh:24-31 -- and BASELINE configs[2] has
 * more chunks than GPUs on 1-4 GPUs) through ONE pass of the layer list.  ios[i] is image i's io block (every image
 * its own pose window, previous frames, use_raw_only and outputs).  The ResnetBlock chains (Winograd F(4x4,3x3)) run
 * batched: the transforms carry the image index in their grid, and the 36 GEMMs see batch x T tile rows per weight
 * matrix instead of T -- the skinny M = 256 of one 64x64 bottleneck becomes 512 at batch 2; norm statistics stay
 * per image (instance norm), so each image's frame is the frame t2v_generator_forward computes for it.  The
 * remaining layers are launched image by image.  Workspace from ..._workspace_bytes_batch(d, batch). */
#define T2V_MAX_BATCH 8
size_t t2v_generator_workspace_bytes_batch(const t2v_gen_desc* d, int batch);
int t2v_generator_forward_batch(t2v_ctx* ctx, void* stream, const t2v_gen_desc* d, const t2v_layer* layers,
                                int n_layers, const t2v_gen_io* ios, int batch, void* workspace, size_t ws_bytes);

/* ------------------------------------------------------------------------------------------
 * Host plumbing (ABI 14): device buffers, pinned host buffers, copies, streams and events for a host that has no HIP
 * binding of its own -- the reference's hosts got these from THC (THCudaMalloc / THCudaFree
 * $SP/torch/lib/include/THC/THCGeneral.h:143-144, THCudaHostAlloc :147, THCState_getCurrentStream :105,
 * THCCachingHostAllocator.h).  `vid2vid/test.py` runs its frame loop on these alone (text2video_amd/leantorch.py:
 * the one-shot command no longer pays for `import torch`); a host that does have an allocator (PyTorch in the tests,
 * the trainer and the resident server) passes its own pointers and never calls them.  All of them make ctx's device
 * current.  Nothing here caches: t2v_device_malloc is hipMalloc.
 * ------------------------------------------------------------------------------------------ */
int t2v_device_malloc(t2v_ctx* ctx, size_t bytes, void** out);
int t2v_device_free(t2v_ctx* ctx, void* ptr);
int t2v_host_malloc(t2v_ctx* ctx, size_t bytes, void** out);      /* page-locked, portable */
int t2v_host_free(t2v_ctx* ctx, void* ptr);
enum { T2V_COPY_H2D = 1, T2V_COPY_D2H = 2, T2V_COPY_D2D = 3 };
/* enqueued on `stream`; asynchronous to the host when the host side is page-locked (a pageable source is staged by the
 * runtime and the call returns when the host buffer may be reused) */
int t2v_memcpy(t2v_ctx* ctx, void* stream, void* dst, const void* src, size_t bytes, int kind);
int t2v_stream_create(t2v_ctx* ctx, void** out);                  /* non-blocking with respect to the null stream */
int t2v_stream_destroy(t2v_ctx* ctx, void* stream);
int t2v_stream_synchronize(t2v_ctx* ctx, void* stream);
int t2v_event_create(t2v_ctx* ctx, void** out);                   /* no timing */
int t2v_event_record(t2v_ctx* ctx, void* event, void* stream);
int t2v_event_synchronize(t2v_ctx* ctx, void* event);
int t2v_event_destroy(t2v_ctx* ctx, void* event);
int t2v_device_synchronize(t2v_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* T2V_H_ */
