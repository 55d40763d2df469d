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
    int nparts;       // instance-norm partial-statistics rows (= nphases * mtiles)
    size_t wfloats;   // packed weight size
};

int build_conv_plan(const t2v_conv_desc* d, int x_cs, bool need_stats, ConvPlan* out);
bool winograd_supported(const t2v_conv_desc* d, int x_cs);
int build_winograd_gemm_plan(const t2v_conv_desc* d, ConvPlan* pl);
int run_conv(t2v_ctx* ctx, hipStream_t s, const ConvPlan& pl, const float* x, const float* w, const float* bias,
             float* y, int y_cs, float* stats);

}  // namespace t2v
