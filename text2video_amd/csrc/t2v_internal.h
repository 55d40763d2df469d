// Internal declarations shared by the HIP translation units of libt2v_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/t2v.h"

namespace t2v {

// ---------------------------------------------------------------------------------------------
// error plumbing: nothing throws across the C ABI; every entry point returns t2v_status and
// leaves a thread-local message for t2v_last_error().
// ---------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define T2V_HIP_CHECK(expr)                                                            \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            ::t2v::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,             \
                             hipGetErrorString(_e));                                   \
            return T2V_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)
#define T2V_REQUIRE(cond, ...)                                                         \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            ::t2v::set_error(__VA_ARGS__);                                             \
            return T2V_ERR_INVALID;                                                    \
        }                                                                              \
    } while (0)
#define T2V_TRY(expr)                                                                  \
    do {                                                                               \
        int _s = (expr);                                                               \
        if (_s != T2V_OK) return _s;                                                   \
    } while (0)

// ---------------------------------------------------------------------------------------------
// Switches read from the environment ONCE (the first t2v_create, or the first call that needs them) -- never on a launch
// path.  t2v_reload_env() reads them again: tests and A/B runs that flip a variable inside one process call it.
// ---------------------------------------------------------------------------------------------
struct Options {
    int wino_gemm_sk;        // T2V_WINO_GEMM_SK: 0 off, 1 where it pays (default), 2 wherever the shape allows
    int overlap_hint;        // T2V_OVERLAP_HINT: 0 ignores the overlap hint (t2v_set_overlap_hint / the generator's two-stream frames)
    int overlap_hint_single; // T2V_OVERLAP_HINT_SINGLE: 1 = under the hint ONE 512x512 image's GEMM stage takes the 256 x 128 form as well (rounds 4-5)
    int wino_gemm_sk_ragged; // T2V_WINO_GEMM_SK_RAGGED: ragged M tiles for tile rows that are no whole 128s: 1 = 4,..,4,r fragments on
                             // two blocks per CU, 2 = balanced 3..6-fragment tiles on one block per CU where that applies
    int wgrad_sk;            // T2V_WGRAD_SK: as wino_gemm_sk, for the Winograd-domain weight gradient
    int wgrad_combine;       // T2V_WGRAD_COMBINE: in-kernel combine of split partials (default 1)
    int wgrad_combine_max;   // T2V_WGRAD_COMBINE_MAX: ... up to this many partials (default 4)
    int chain_lazy;          // T2V_CHAIN_LAZY: ResnetBlock chains apply their norms in the next input transform (default 1)
    int conv_tile;           // T2V_CONV_TILE (measurement): 1 = keep the 128x128 tile where the fill rule would take 64x64, 2 = always 64x64
    int streams;             // T2V_STREAMS: 1 = the generator on the caller's stream only, 2 = always two streams; 0 (default):
                             // two, except the global generator on a bottleneck of >= 1024 Winograd tiles (1024x1024 frames)
    int sk_blocks_per_cu;    // T2V_SK_BLOCKS_PER_CU: 0 (default) = by the overlap hint, 1 / 2 = always one / two blocks per CU
    int xcd_slices;          // T2V_XCD_SLICES: overlap-reading transforms give every XCD its own channel slices (default 1; 0 = the
                             // flat thread index: the A/B twin, same bits)
};
const Options& options();
// true while the calling thread has announced a second stream beside its launches (t2v_set_overlap_hint; the generator's
// two-stream frames set it for their own launches) and T2V_OVERLAP_HINT is not 0
bool overlap_hint();
int set_overlap_hint(int on);      // returns the previous value
struct OverlapScope {
    int prev;
    explicit OverlapScope(bool on) : prev(set_overlap_hint(on ? 1 : 0)) {}
    ~OverlapScope() { set_overlap_hint(prev); }
};
void options_reload();

// The fixed-grid kernels hand accumulators from block b - 8 (b - 8 * half) to block b inside one launch.  That cannot
// deadlock as long as workgroups are dispatched in index order (the producer publishes before anything it waits for).  Two
// guards: (i) a self-test at the first t2v_create per process launches an over-subscribed grid whose blocks draw a ticket as
// they start and checks, per XCD, that no block starts far from its place in index order; (ii) a consumer whose producer's tag does not show up within the poll
// bound poisons its tile with NaNs AND raises a sticky error word in pinned host memory.  check_async_errors() -- called by
// every entry point that may launch a fixed-grid kernel, and exported as t2v_check_async_errors -- then returns
// T2V_ERR_HANDOVER once and switches the process to one block per tile (fixed_grid_enabled() == false).
unsigned* async_error_word();      // device-visible; nullptr before the first t2v_create
bool fixed_grid_enabled();
void fixed_grid_disable(const char* why);
int check_async_errors();

// ---------------------------------------------------------------------------------------------
// implicit-GEMM convolution kernel parameters (conv_igemm.hip)
//   GEMM view:  D[m][n] = sum_k A[m][k] * B[n][k]
//     m = pixel of the "GEMM pixel grid" (Hm x Wm), n = output channel,
//     k = tap * Cin_s + c   (tap-major, channel-minor; NHWC makes c contiguous)
//   A is gathered on the fly from the NHWC input (reflect / zero padding resolved in the
//   loader's address computation -- SpatialReflectionPadding is never materialised),
//   B is the pre-packed weight [Cout_p][Kp] (K contiguous, zero padded).
//   A transposed convolution (k3 s2 p1 op1) is 4 sub-pixel "phases", each a small-tap conv on
//   the input grid whose output is scattered with stride 2 (no zero insertion).
// ---------------------------------------------------------------------------------------------
constexpr int kMaxTaps = 52;
constexpr int kMaxPhases = 4;
constexpr int kBK = 32;  // K-depth of one LDS stage (floats): 128-byte rows

struct ConvPhase {
    int ntaps;     // taps of this phase
    int tap0;      // first entry in tdy/tdx
    int nk;        // number of BK stages (= Kp / 32)
    int Kp;        // padded K of this phase
    long w_off;    // float offset of this phase's packed weights
    int oy0, ox0;  // output offset (sub-pixel phase)
};

struct ConvKParams {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    float* stats;  // [nparts][Cout][2] (mean_b, M2_b) or nullptr
    int Hin, Win, Cin_s;
    int Wm, M;
    int Cout, Cout_s, Wout, Hout;
    int stride, ostride;
    int pad_mode, act;
    float act_scale;  // flow multiplier of T2V_ACT_FLOW_W
    int mtiles, ntiles, nphases;
    int group_mtiles;     // M tiles per weight group (Winograd: one weight matrix per transform position)
    long group_w_stride;  // float offset between the groups' weight matrices (0: a single matrix)
    int KW, pad;  // MODE 1 (Cin_s % 32 != 0, regular conv): tap -> (kh,kw) by arithmetic
    // images per launch (blockIdx.y; the discriminators' batches of the train step): image b reads x + b * x_img_stride,
    // writes y + b * y_img_stride and its statistics partials at stats + b * stats_img_stride (floats; batch <= 1: unused)
    int batch;
    long x_img_stride, y_img_stride, stats_img_stride;
    ConvPhase ph[kMaxPhases];
    int tdy[kMaxTaps];  // ints: read with scalar loads (uniform index), never a vector load
    int tdx[kMaxTaps];
};

// weight-gradient kernel (conv_wgrad.hip): one block = 128 Cout x 128 Cin of one tap, reduction over pixels
struct WgradParams {
    const float* x;    // forward input  [batch][Hin][Win][Cin_s]
    const float* dy;   // output gradient [batch][Hout][Wout][Cout_s]
    float* dw;         // packed weight gradient (same layout as the packed forward weight)
    int batch, Hin, Win, Cin_s;
    // floats between consecutive images of x / dy.  The batch's own pitch by default; anything else (two frames of a clip in
    // buffers of their own: t2v_conv2d_backward_weight_strided) only where conv_wgrad_strided_ok() says so
    long x_img_stride, dy_img_stride;
    int Wm, M;         // GEMM pixel grid of the forward conv, per image
    int Hout, Wout, Cout, Cout_s;
    int stride, ostride;
    int ntaps, ntiles, ctiles;
    int reflect, accumulate;
    // fold: narrow inputs (Cin_s < 64, regular conv).  The 128-wide input-channel side of the block tile then
    // runs over K = tap*Cin_s + c of ALL taps (tap per lane chunk), instead of one tap with 128 - Cin_s idle columns
    int fold, fold_taps, KW, pad;
    // fold == 2: few output channels (the 3-channel heads).  x is a padded copy (no tap shifts on the X side), the
    // taps are folded into the 128-wide dY side instead: row n' = tap*Cout_s + n reads dY at pixel q - (kh, kw)
    int splits;        // the pixel reduction is cut into `splits` ranges (blockIdx major), each writing its own
    long dw_floats;    // partial gradient at dw + split*dw_floats (deterministic; summed by wgrad_reduce)
    // in-kernel combine (tickets != nullptr, splits > 1): the partials are published write-through, every (tap, n, c)
    // tile has an arrival counter, and the block that draws the last ticket of a tile adds the `splits` partials up in
    // split order -- the order wgrad_reduce uses -- and writes (accumulates into) dw_final: no zero-fill of the
    // partial slabs, no reduce launch
    float* dw_final;
    int* tickets;      // ntaps * ntiles * ctiles zeroed ints (left zeroed)
    int tdy[kMaxTaps], tdx[kMaxTaps];   // input offset of every tap (all phases concatenated)
    int toy[kMaxTaps], tox[kMaxTaps];   // output offset (sub-pixel phase) of the tap's phase
    long tap_woff[kMaxTaps];            // float offset of the tap's phase matrix in dw
    int tap_Kp[kMaxTaps];               // padded K of that phase matrix
    int tap_kidx[kMaxTaps];             // index of the tap inside its phase
};
int launch_conv_wgrad(hipStream_t s, const WgradParams& p);
// true when the launch takes the scalar-stepped loader path, the one that addresses every image through its own buffer resource
bool conv_wgrad_strided_ok(const WgradParams& p);
int launch_wgrad_reduce(hipStream_t s, const float* partial, int splits, long n, float* dw, int accumulate);
int launch_unpack_conv_weight(hipStream_t s, const float* packed, float* w, int Cout, int Cin, int KH, int KW, int Cin_s,
                              int Kp, int accumulate = 0);
int launch_unpack_convT_weight(hipStream_t s, const float* packed, float* w, int Cin, int Cout, int Cin_s, int Cout_p,
                               int K, int pad, int accumulate = 0);
int launch_accumulate(hipStream_t s, float* dst, const float* src, long n, int overwrite);
int launch_scale(hipStream_t s, float* x, long n, float sc);
int launch_unzip2(hipStream_t s, const float* src, float* dst0, float* dst1, int C, int overwrite);
int launch_pad_copy(hipStream_t s, const float* x, float* xp, int batch, int H, int W, int C, int pad, int reflect);
int launch_reflect_pad_backward(hipStream_t s, const float* dxp, float* dx, int H, int W, int C, int p);
int launch_inorm_backward(hipStream_t s, const float* x, const float* dy, const float* mean_rstd, const float* gamma,
                          const float* beta, int relu, long npix, int C, float* scratch, float* dx, float* sums,
                          float* d_beta = nullptr, float* d_gamma = nullptr, int overwrite = 0);
int launch_act_backward(hipStream_t s, const float* dy, const float* y, int mode, float slope, long n, float* dpre);
int launch_avgpool3s2_backward(hipStream_t s, const float* dy, float* dx, int H, int W, int C);
int launch_maxpool2x2(hipStream_t s, const float* x, float* y, int H, int W, int C);
int launch_maxpool2x2_backward(hipStream_t s, const float* x, const float* dy, float* dx, int H, int W, int C);
int launch_loss_backward(hipStream_t s, int op, const float* a, const float* b, float c, float scale, long n, float* da);
// Winograd tile count padded so that every transform position owns whole GEMM tiles: a multiple of 128 (128x128
// tiles), or of 64 (forces the 64x64 tile config) when that drops at least an eighth of the rows -- e.g. the
// 66x66 data gradient of a 64x64 map: 289 tiles -> 320 instead of 384
inline int wino_pad_tiles(int T) {
    const int a = (T + 63) / 64 * 64, b = (T + 127) / 128 * 128;
    return (b - a) * 8 >= b ? a : b;
}
int launch_winograd_weight(hipStream_t s, const float* w, float* U, int Cout, int Cin, int Cout_p, int Cin_s, int adjoint = 0);
int launch_winograd_input(hipStream_t s, const float* x, float* V, int H, int W, int C, int pad, int reflect);
int launch_winograd_output(hipStream_t s, const float* Mm, const float* bias, float* y, float* stats, int H, int W, int N);
int launch_winograd4_weight(hipStream_t s, const float* w, float* U, int Cout, int Cin, int Cout_p, int Cin_s, int adjoint = 0);
// nimg == 0: slot layout (image -> slot `image` of `batch` slots of Tp rows); nimg > 0: packed batch of nimg images
int launch_winograd4_input(hipStream_t s, const float* x, float* V, int H, int W, int C, int pad, int reflect, int batch = 1,
                           int image = 0, int nimg = 0, long img_stride = 0);
// input transform of a conv output that still has to go through its norm layer (winograd4_input_kernel<1|2>)
int launch_winograd4_input_lazy(hipStream_t s, const float* x, float* V, int H, int W, int C, int pad, int reflect,
                                const float* mean_rstd, const float* gamma, const float* beta, int relu_only,
                                const float* res, float* xout, int nimg = 1, long img_stride = 0);
int launch_winograd4_dgrad_output(hipStream_t s, const float* dV, float* dxp, int H, int W, int C);
// polyphase.hip
int launch_polyphase_weight(hipStream_t s, const float* w, float* U, int Cout, int Cin, int Cout_p, int Cin_s, int up);
int launch_polyphase_input(hipStream_t s, const float* x, float* V, int H, int W, int C, int up, int Tt,
                           const float* mean_rstd = nullptr, const float* gamma = nullptr, const float* beta = nullptr, int relu = 0);
int launch_polyphase_output(hipStream_t s, const float* Mm, const float* bias, float* y, float* stats, int Ho, int Wo, int N,
                            int up, int Tt);
int launch_winograd4_dy(hipStream_t s, const float* dy, float* Md, int Ho, int Wo, int N, int dy_cs, int batch, int image);
int launch_winograd4_dy_norm(hipStream_t s, const float* dy, const float* x, const float* mean_rstd, const float* gamma,
                             const float* beta, int relu, const float* sums, float* Md, int Ho, int Wo, int N, int batch,
                             int image);
int launch_winograd4_dw(hipStream_t s, const float* dU, float* dw, int Cout, int Cin, int Cout_p, int Kp, int accumulate);
int launch_winograd4_output(hipStream_t s, const float* Mm, const float* bias, float* y, float* stats, int H, int W, int N,
                            int lrelu, float slope, int nimg = 1);
int launch_channel_sum(hipStream_t s, const float* x, long npix, int C, int cs, float* scratch, float* out);

// 7x7 reflect-padded head convolution with <= 3 output channels (conv_head.hip)
struct HeadParams {
    const float* x;
    const float* w;      // the packed [Cout_p][Kp] weight of the implicit-GEMM kernels (rows 0..2 used)
    const float* bias;
    float* y;
    int H, W, Cin_s, Kp, Cout, Cout_s, act;
    float act_scale;
};
int launch_conv_head7x7(hipStream_t s, const HeadParams& p);

// single-output-channel convolution (the PatchGAN discriminators' last layer; conv_head.hip)
struct Cout1Params {
    const float* x;
    const float* w;      // row 0 of the packed [Cout_p][Kp] weight: K = tap * Cin_s + c
    const float* bias;
    float* y;
    int H, W, Cin_s, ksize, stride, pad, Hout, Wout, Cout_s, act;
    float act_scale;
};
bool conv_cout1_supported(int ksize, int Cin_s);
int launch_conv_cout1(hipStream_t s, const Cout1Params& p);

struct StemParams {
    const float* x;
    const float* w;      // the packed [Cout_p][Kp] weight of the implicit-GEMM kernels
    const float* bias;
    float* y;
    float* stats;        // [tiles][Cout] (mean, M2) per 16x16 pixel tile
    int H, W, Cin_s, Kp, Cout, Cout_s;
    int Cin;             // real input channels (<= Cin_s; the weights of the storage padding are zero)
};
bool conv_stem7x7_supported(int H, int W, int Cin_s, int Cout);
int launch_conv_stem7x7(hipStream_t s, const StemParams& p);

enum ConvTile { kTileL = 0 /*128x128, 32x32x2 MFMA*/, kTileS = 1 /*256x16, 16x16x4 MFMA*/, kTileQ = 2 /*64x64*/,
                kTileStem = 3 /*conv_stem.hip: 16x16 pixel tile x all channels*/ };
int conv_tile_for(int Cout);
void conv_tile_dims(int tile, int* BM, int* BN);
int launch_conv_igemm(hipStream_t s, const ConvKParams& p, int tile);

// The batched Winograd GEMM (36 groups x [T x K] x [K x N]) on a fixed grid of two blocks per CU, each block taking an
// equal run of K stages out of the tile-major list of (128x128 tile, stage) units: 576 tiles of a 512x512 frame are
// 2.25 rounds of whole tiles on 256 CUs, but exactly 36 stages per block.  A tile cut between two blocks is finished
// by the second one STARTING from the first one's accumulators (handed over through `scratch`), so every output is
// the same K-ordered sum as in conv_igemm_kernel: bit-identical results.
struct SkGemm {
    const float* a;        // [groups][a_group_rows][K], the first T rows of each group are used
    const float* b;        // [groups][N][K]
    float* c;              // [groups][T][c_cs]
    float* scratch;        // wino_gemm_sk_scratch_floats() floats, any content
    unsigned* err;         // sticky error word a timed-out hand-over raises (async_error_word())
    int rows = 0;          // real tile rows per position (<= T) when the caller knows them: selects the 160 x 128 tiles
    long a_group_stride;   // floats between the groups of a
    int groups, T, K, N, c_cs;
    bool b_kn = false;     // b is [groups][K][N] (wino_gemm_sk_bkn_ok): a data gradient reading the forward layer's U in place
};
size_t wino_gemm_sk_scratch_floats();
bool wino_gemm_sk_ok(int groups, int T, int K, int N, int c_cs, int rows = 0);   // rows: real tile rows per position, if known
bool wino_gemm_sk_bkn_ok(int groups, int T, int K, int N, int c_cs);
int wino_gemm_sk_tall_rows(int groups, int rows, int T, int N);      // 160 | 256: the one-block-per-CU form of that kernel is taken; 0: not
int launch_wino_gemm_sk(hipStream_t s, const SkGemm& g);
// the ragged form (conv_igemm.hip: wino_gemm_skr_kernel): `rows` real tile rows per position (<= g.T, the padded pitch), cut
// into 32-row fragments and M tiles of 4, ..., 4, r fragments -- no MFMA work on padding rows
bool wino_gemm_skr_ok(int groups, int rows, int Tp, int K, int N, int c_cs);
int launch_wino_gemm_skr(hipStream_t s, const SkGemm& g, int rows);
// ... and its one-block-per-CU form on balanced tiles of 3..6 fragments (wino_gemm_skt_kernel; T2V_WINO_GEMM_SK_RAGGED=2)
bool wino_gemm_skt_ok(int groups, int rows, int Tp, int K, int N, int c_cs);
int launch_wino_gemm_skt(hipStream_t s, const SkGemm& g, int rows);
int wino_gemm_sk_grid_blocks();                 // two blocks per CU, a multiple of 8
// blocks the whole-tile fixed-grid GEMM and the Winograd-domain weight gradient LAUNCH: wino_gemm_sk_grid_blocks(), or half of
// it (one block per CU) while the calling thread's overlap hint is 2 (t2v_set_overlap_hint) / T2V_SK_BLOCKS_PER_CU=1.  A
// 128 x 128 block holds half a CU (64 KiB of LDS, 8 waves of <= 128 VGPRs): with one per CU the launch of the caller's OTHER
// stream -- another fixed-grid GEMM, or the bandwidth-bound kernels between two of them -- is resident beside it instead of
// queueing behind a grid that keeps every CU full until its last block leaves.  Which block owns a tile changes, an output's
// K-ordered accumulation chain does not: the same bits.  The shape rules (*_ok) keep counting with two per CU.
int sk_launch_blocks();
constexpr int kSkExclusiveLds = 96 * 1024;      // dynamic LDS a one-per-CU launch asks for: no second GEMM block on its CU
int sk_exclusive_lds(int own_bytes);            // ... clamped to what the device lets a block opt in to (never below the kernel's own need)
unsigned long long wino_gemm_sk_next_tag();     // hand-over tags: unique per launch, process-wide
// the same scheme for the Winograd-domain weight gradient dU[xi] = M_dy[xi]^T V[xi] (conv_wgrad.hip); scratch as above
bool wino_wgrad_sk_ok(int Tt, int Cin, int Cout);
int launch_wino_wgrad_sk(hipStream_t s, const float* V, const float* Md, float* dU, float* scratch, int Tt, int Cin, int Cout,
                         int Cout_p, int Kp);

// elementwise.hip
// scratch (optional): kFinalizeMaxGroups * C * 4 doubles -- lets big layers pool their partials on many CUs
constexpr int kFinalizeMaxGroups = 64;
// ru (optional): BatchNorm2d running statistics moved by the statistics being finalized, in the same launch
struct RunningUpdate {
    float* mean;
    float* var;
    float n;          // values per channel behind the statistics (unbiased-variance factor n / (n - 1))
    float momentum;
    int times;        // updates to apply (a forward that stands for two upstream forwards: 2)
};
int launch_inorm_finalize_winograd(hipStream_t s, const float* stats, int wm, int H, int W, int C, float eps,
                                   float* mean_rstd, int batch, double* scratch = nullptr, const RunningUpdate* ru = nullptr);
int launch_inorm_finalize_tiles(hipStream_t s, const float* stats, int edge, int H, int W, int C, float eps,
                                float* mean_rstd, int batch, double* scratch = nullptr, const RunningUpdate* ru = nullptr);
int launch_inorm_finalize(hipStream_t s, const float* stats, int nparts, int mtiles, int BM, int M, int C,
                          float eps, float* mean_rstd, double* scratch = nullptr, const RunningUpdate* ru = nullptr);
int launch_inorm_apply(hipStream_t s, const float* x, const float* mean_rstd, const float* gamma,
                       const float* beta, const float* res1, const float* res2, float* y, long npix, int C,
                       int relu, int nimg = 1);
int launch_bn_running_update(hipStream_t s, const float* mean_rstd, float* running_mean, float* running_var, long n,
                             float momentum, float eps, int C);
int launch_pack_conv_weight(hipStream_t s, const float* w, float* packed, int Cout, int Cin, int KH, int KW,
                            int Cin_s, int Kp, int Cout_p, int adjoint = 0);
int launch_pack_convT_weight(hipStream_t s, const float* w, float* packed, int Cin, int Cout, int Cin_s, int Cout_p,
                             int K, int pad);
// taps of sub-pixel phase `phase` of ConvTranspose2d(k3,s2,p1,op1); shared by packer and launcher
void convT_phase_taps_host(int phase, int k, int pad, int* ntaps, int kh[4], int kw[4], int dy[4], int dx[4], int* a,
                           int* b);
int launch_nchw_to_nhwc(hipStream_t s, const float* src, float* dst, int C, int H, int W, int Cs);
int launch_nhwc_to_nchw(hipStream_t s, const float* src, float* dst, int C, int H, int W, int Cs);
int launch_reduce(hipStream_t s, int op, const float* a, const float* b, float c, long n, float* scratch, float* out);
int launch_loss_terms(hipStream_t s, const long long* tp, const int* ti, const float* tf, const int* chunk_term,
                      const long long* chunk_off, const int* term_chunk0, int nterms, int nchunks, int chunk, float* part,
                      float* out);
int launch_adam(hipStream_t s, float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2,
                double eps, int step);
int launch_adam_multi(hipStream_t s, const long long* ptrs, const long long* nelem, const float* step_size,
                      const int* chunk_tensor, const long long* chunk_off, int nchunks, int chunk, double b1, double b2,
                      double eps);
int launch_masked_l1(hipStream_t s, const float* a, const float* b, const float* mask, long npix, int c0, int C, int cs,
                     float* scratch, float* out);
int launch_masked_l1_backward(hipStream_t s, const float* a, const float* b, const float* mask, float scale, long npix,
                              int c0, int C, int cs, float* da);
int launch_add(hipStream_t s, const float* a, const float* b, float* y, long n);
int launch_warp_composite(hipStream_t s, const float* raw, const float* fw, const float* prev, int prev_cs,
                          int prev_c0, float* out, float* warp_out, int H, int W);
int launch_warp_composite_backward(hipStream_t s, const float* d_out, const float* d_warp, const float* raw,
                                   const float* fw, const float* prev, int prev_cs, int prev_c0, float* d_raw,
                                   float* d_fw, float* d_prev, int H, int W);
int launch_avgpool3s2(hipStream_t s, const float* x, float* y, int H, int W, int C);
int launch_to_u8(hipStream_t s, const float* x, uint8_t* y, long n);
int launch_u8_pose_to_f32(hipStream_t s, const uint8_t* src, float* dst, long npix, int Cs, int c0);
int launch_copy_channels(hipStream_t s, const float* src, int src_cs, int src_c0, float* dst, int dst_cs,
                         int dst_c0, int nc, long npix);

}  // namespace t2v

struct t2v_ctx {
    int device;
    // side stream + fork/join events: the generator runs its two independent encoders (and, with the flow
    // branch, its two decoder branches) concurrently; everything is joined back into the caller's stream
    // before the entry point returns
    hipStream_t side;
    hipEvent_t ev_fork, ev_join;
};
