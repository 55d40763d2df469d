// conv_plan.h -- host-side planning of one convolution launch (shared by capi.hip/generator.hip).
#pragma once
#include "t2v_internal.h"

namespace t2v {

struct ConvPlan {
    ConvKParams kp;   // everything except the pointers / Cout_s
    int tile;         // ConvTile
    int BM, BN;
    int Hout, Wout;
    int Cout_p;       // Cout rounded up to the N tile (rows of the packed weight)
    int Cin;          // real input channels (<= kp.Cin_s)
    int nparts;       // instance-norm partial-statistics rows (= nphases * mtiles)
    size_t wfloats;   // packed weight size
};

int build_conv_plan(const t2v_conv_desc* d, int x_cs, bool need_stats, ConvPlan* out);
inline bool is_winograd(int algo) { return algo == T2V_ALGO_WINOGRAD || algo == T2V_ALGO_WINOGRAD_F4; }
inline int wino_m(int algo) { return algo == T2V_ALGO_WINOGRAD_F4 ? 4 : 2; }          // output tile edge
inline int wino_pos(int algo) { return (wino_m(algo) + 2) * (wino_m(algo) + 2); }    // transform positions: 16 | 36
// tiles of the ceil(H/m) x ceil(W/m) grid, padded to whole 128-row GEMM tiles per transform position
inline int wino_out_h(const t2v_conv_desc* d) { return d->H + 2 * d->pad - 2; }   // 3x3, stride 1
inline int wino_out_w(const t2v_conv_desc* d) { return d->W + 2 * d->pad - 2; }
inline int wino_tiles_padded(const t2v_conv_desc* d, int algo) {
    const int m = wino_m(algo);
    const int T = ((wino_out_h(d) + m - 1) / m) * ((wino_out_w(d) + m - 1) / m);
    return wino_pad_tiles(T);
}
// GEMM rows per transform position for a batch of nimg images: F(4x4) packs the images' tiles and pads the total
inline int wino_tiles_real(const t2v_conv_desc* d, int algo) {
    const int m = wino_m(algo);
    return ((wino_out_h(d) + m - 1) / m) * ((wino_out_w(d) + m - 1) / m);
}
inline int wino_rows_batch(const t2v_conv_desc* d, int algo, int nimg) {
    return (nimg > 1 && algo == T2V_ALGO_WINOGRAD_F4) ? wino_pad_tiles(nimg * wino_tiles_real(d, algo)) : wino_tiles_padded(d, algo);
}
inline size_t winograd_vm_floats(const t2v_conv_desc* d, int nimg = 1) {             // V + M of `nimg` images
    return (size_t)wino_pos(d->algo) * wino_rows_batch(d, d->algo, nimg) * ((size_t)d->Cin + d->Cout);
}
// ... followed, for F(4x4,3x3), by the hand-over scratch of the fixed-grid GEMM (conv_igemm.hip: wino_gemm_sk_kernel)
inline size_t winograd_workspace_floats(const t2v_conv_desc* d, int nimg = 1) {
    return winograd_vm_floats(d, nimg) + (d->algo == T2V_ALGO_WINOGRAD_F4 ? wino_gemm_sk_scratch_floats() : 0);
}
// GEMM rows of the whole conv (all positions): what the algorithm choice compares
inline long wino_gemm_rows(const t2v_conv_desc* d, int algo) { return (long)wino_pos(algo) * wino_tiles_padded(d, algo); }
bool winograd_supported(const t2v_conv_desc* d, int x_cs, int algo);
// ---- polyphase Winograd F(4,2) of the stride-2 / transposed 3x3 layers (polyphase.hip): 81 positions, tiles of 4x4 outputs
// (down) | 4x4 inputs = 8x8 outputs (up)
bool polyphase_supported(const t2v_conv_desc* d, int x_cs);
bool polyphase_pays(const t2v_conv_desc* d, int x_cs);      // ... and is the faster form (the generator's selection rule)
inline int poly_tiles_real(const t2v_conv_desc* d) {
    return d->transposed ? ((d->H + 3) / 4) * ((d->W + 3) / 4) : ((d->H / 2 + 3) / 4) * ((d->W / 2 + 3) / 4);
}
inline int poly_tiles_padded(const t2v_conv_desc* d) { return wino_pad_tiles(poly_tiles_real(d)); }
inline int poly_out_h(const t2v_conv_desc* d) { return d->transposed ? 2 * d->H : d->H / 2; }
inline int poly_out_w(const t2v_conv_desc* d) { return d->transposed ? 2 * d->W : d->W / 2; }
inline int poly_m(const t2v_conv_desc* d) { return d->transposed ? 8 : 4; }       // output tile edge (statistics partial geometry)
inline size_t polyphase_workspace_floats(const t2v_conv_desc* d) {                // V + M + the fixed-grid GEMM's hand-over scratch
    return (size_t)81 * poly_tiles_padded(d) * ((size_t)d->Cin + d->Cout) + wino_gemm_sk_scratch_floats();
}
// lazy != null: x is the previous layer's raw conv output; its norm (+ ReLU) is applied inside the input transform
struct PolyLazyNorm {
    const float* mean_rstd;
    const float* gamma;
    const float* beta;
    int relu;
};
int polyphase_forward(t2v_ctx* ctx, hipStream_t s, const t2v_conv_desc* d, const float* x, const float* w_packed,
                      const float* bias, float* y, float* stats_partial, float* workspace, int stages,
                      const PolyLazyNorm* lazy = nullptr);
int best_conv_algo(const t2v_conv_desc* d, int x_cs, int cap);
int build_winograd_gemm_plan(const t2v_conv_desc* d, ConvPlan* pl, int nimg = 1);
// A batch of images through one Winograd conv (F(4x4,3x3) only when nimg > 1): the images' maps x / y are
// `img_stride_x` / H*W*Cout floats apart, V and M hold nimg*Tp tile rows per transform position (one GEMM with
// M = nimg*Tp rows per position), the statistics partials follow each other image by image.
struct WinoBatch {
    int nimg = 1;
    long img_stride_x = 0;     // floats between the input maps
    // keep_v != null (nimg == 1): V goes into slot `keep_slot` of a batch-wide tile list [36][keep_total * Tp][Cin] -- the
    // Winograd-domain weight gradient's workspace -- instead of the call's own workspace, and the GEMM reads it from there
    float* keep_v = nullptr;
    int keep_total = 1, keep_slot = 0;
};
int winograd_forward(t2v_ctx* ctx, hipStream_t s, const t2v_conv_desc* d, const float* x, const float* w_packed,
                     const float* bias, float* y, float* stats_partial, float* workspace, int stages,
                     const WinoBatch* batch = nullptr);
int run_conv(t2v_ctx* ctx, hipStream_t s, const ConvPlan& pl, const float* x, const float* w, const float* bias,
             float* y, int y_cs, float* stats);
// `batch` images a constant stride apart in one implicit-GEMM launch (blockIdx.y); strides in floats
int run_conv_batch(t2v_ctx* ctx, hipStream_t s, const ConvPlan& pl, int batch, const float* x, long x_stride, const float* w,
                   const float* bias, float* y, int y_cs, long y_stride, float* stats, long stats_stride);

}  // namespace t2v
