// generator.hip -- whole-frame orchestration of the vid2vid generators on one HIP stream.
//
// Replaces CompositeGenerator.forward / CompositeLocalGenerator.forward (SURVEY.md App. A.1/A.2;
// section 8a rows a4, a13).  Pure launch sequencing: every tensor lives in the caller's
// workspace (bump-allocated here, identically by t2v_generator_workspace_bytes), nothing
// synchronises, so a frame is ~150 back-to-back launches on the caller's stream.
#include <vector>

#include "conv_plan.h"

namespace t2v {
namespace {

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct Arena {
    char* base;
    size_t cap;
    size_t off = 0;
    bool overflow = false;
    float* alloc(size_t floats) {
        const size_t bytes = (floats * sizeof(float) + 255) / 256 * 256;
        float* p = reinterpret_cast<float*>(base + off);
        off += bytes;
        if (base && off > cap) overflow = true;
        return p;
    }
};

t2v_conv_desc mk_conv(int H, int W, int Cin, int Cout, int k, int stride, int pad, int pad_mode, int transposed,
                      int act = T2V_ACT_NONE, float act_scale = 1.f) {
    t2v_conv_desc d;
    d.H = H; d.W = W; d.Cin = Cin; d.Cout = Cout; d.kH = k; d.kW = k; d.stride = stride; d.pad = pad;
    d.pad_mode = pad_mode; d.transposed = transposed; d.act = act; d.act_scale = act_scale;
    d.output_padding = transposed ? 1 : 0;
    d.algo = T2V_ALGO_DIRECT;
    return d;
}

struct LayerSpec {
    t2v_conv_desc cd;
    int x_cs;
    bool has_norm;
};

// canonical layer list (the order documented in t2v.h)
void enumerate_layers(const t2v_gen_desc& g, std::vector<LayerSpec>& out) {
    const int G = g.ngf, n = g.is_local ? 1 : g.n_downsample;
    const int H = g.H, W = g.W;
    auto enc = [&](int in_nc) {
        out.push_back({mk_conv(H, W, in_nc, G, 7, 1, 3, T2V_PAD_REFLECT, 0), round_up(in_nc, 4), true});
        for (int i = 0; i < n; ++i)
            out.push_back({mk_conv(H >> i, W >> i, G << i, G << (i + 1), 3, 2, 1, T2V_PAD_ZERO, 0), G << i, true});
    };
    auto rbs = [&](int count) {
        const int C = G << n;
        t2v_conv_desc cd = mk_conv(H >> n, W >> n, C, C, 3, 1, 1, T2V_PAD_REFLECT, 0);
        // the ResnetBlock convs (84 % of the FLOPs) run as Winograd F(2x2,3x3) wherever the geometry allows
        cd.algo = best_conv_algo(&cd, C, g.conv_algo);
        for (int i = 0; i < 2 * count; ++i) out.push_back({cd, C, true});
    };
    auto ups = [&]() {
        for (int i = 0; i < n; ++i) {
            const int l = n - i;
            out.push_back({mk_conv(H >> l, W >> l, G << l, G << (l - 1), 3, 2, 1, T2V_PAD_ZERO, 1), G << l, true});
        }
    };
    const int nb_enc = g.is_local ? 0 : g.n_blocks - g.n_blocks / 2;
    const int nb_res = g.is_local ? g.n_blocks : g.n_blocks / 2;
    enc(g.input_nc); rbs(nb_enc);
    enc(g.prev_nc);  rbs(nb_enc);
    rbs(nb_res); ups();
    out.push_back({mk_conv(H, W, G, g.output_nc, 7, 1, 3, T2V_PAD_REFLECT, 0, T2V_ACT_TANH), G, false});
    if (!g.no_flow) {
        rbs(nb_res); ups();
        out.push_back({mk_conv(H, W, G, 3, 7, 1, 3, T2V_PAD_REFLECT, 0, T2V_ACT_FLOW_W, g.flow_multiplier), G, false});
    }
}

int check_desc(const t2v_gen_desc* g) {
    T2V_REQUIRE(g, "null generator descriptor");
    const int n = g->is_local ? 1 : g->n_downsample;
    T2V_REQUIRE(g->ngf > 0 && g->ngf % 4 == 0, "ngf=%d must be a positive multiple of 4", g->ngf);
    T2V_REQUIRE(n >= 1 && n <= 6, "n_downsample=%d out of range", n);
    T2V_REQUIRE(g->H > 0 && g->W > 0 && g->H % (1 << n) == 0 && g->W % (1 << n) == 0,
                "H=%d W=%d must be positive multiples of %d", g->H, g->W, 1 << n);
    T2V_REQUIRE((g->H >> n) >= 2 && (g->W >> n) >= 2, "bottleneck %dx%d too small for reflection pad 1", g->H >> n,
                g->W >> n);
    T2V_REQUIRE(g->output_nc == 3, "output_nc=%d: only RGB output is on the path", g->output_nc);
    T2V_REQUIRE(g->input_nc > 0 && g->prev_nc >= 3, "bad input_nc/prev_nc");
    T2V_REQUIRE(g->n_blocks >= 0, "bad n_blocks");
    return T2V_OK;
}

struct Buffers {
    float *encA[8], *encB[8];  // encoder activations per level (A: pose/seg, B: prev-image)
    float* bt[4];              // bottleneck temporaries (resnet chains)
    float* bt2[4];             // ... of the branch that runs on the side stream
    float* d;                  // encoder sum
    float *dimg, *dflow;       // local generator: d + coarse features
    float *decI[8], *decF[8];  // decoder activations per level
    float *raw, *fw;
    // per-stream scratch [0]: caller's stream, [1]: side stream
    float* stats[2];
    float* mean_rstd[2];
    float* wino[2];   // Winograd scratch: transformed input V + transformed output M
    double* fin[2];   // norm finalize scratch (pooled moments per group of partials)
};

void plan_buffers(const t2v_gen_desc& g, const std::vector<LayerSpec>& layers, Arena& a, Buffers& b) {
    const int G = g.ngf, n = g.is_local ? 1 : g.n_downsample;
    auto lvl = [&](int l) { return (size_t)(g.H >> l) * (g.W >> l) * (G << l); };
    for (int l = 0; l <= n; ++l) {
        b.encA[l] = a.alloc(lvl(l));
        b.encB[l] = a.alloc(lvl(l));
    }
    for (int i = 0; i < 4; ++i) b.bt[i] = a.alloc(lvl(n));
    for (int i = 0; i < 4; ++i) b.bt2[i] = a.alloc(lvl(n));
    b.d = a.alloc(lvl(n));
    b.dimg = a.alloc(lvl(n));
    b.dflow = a.alloc(lvl(n));
    for (int l = 0; l < n; ++l) {
        b.decI[l] = a.alloc(lvl(l));
        b.decF[l] = g.no_flow ? nullptr : a.alloc(lvl(l));
    }
    b.raw = a.alloc((size_t)g.H * g.W * 4);
    b.fw = a.alloc((size_t)g.H * g.W * 4);
    size_t max_stats = 0;
    int max_c = 4;
    for (const LayerSpec& L : layers) {
        ConvPlan pl;
        if (L.has_norm && is_winograd(L.cd.algo)) {
            const int m = wino_m(L.cd.algo);
            const size_t s = (size_t)(wino_tiles_padded(&L.cd, L.cd.algo) * m * m / 128) * L.cd.Cout * 2;
            if (s > max_stats) max_stats = s;
        } else if (L.has_norm && build_conv_plan(&L.cd, L.x_cs, true, &pl) == T2V_OK) {
            const size_t s = (size_t)pl.nparts * L.cd.Cout * 2;
            if (s > max_stats) max_stats = s;
        }
        if (L.cd.Cout > max_c) max_c = L.cd.Cout;
    }
    for (int k = 0; k < 2; ++k) {
        b.stats[k] = a.alloc(max_stats);
        b.mean_rstd[k] = a.alloc((size_t)max_c * 2);
        b.fin[k] = reinterpret_cast<double*>(a.alloc((size_t)kFinalizeMaxGroups * max_c * 4 * 2));
    }
    size_t max_wino = 0;
    for (const LayerSpec& L : layers)
        if (is_winograd(L.cd.algo)) {
            const size_t w = winograd_workspace_floats(&L.cd);
            if (w > max_wino) max_wino = w;
        }
    for (int k = 0; k < 2; ++k) b.wino[k] = max_wino ? a.alloc(max_wino) : nullptr;
}

struct Runner {
    t2v_ctx* ctx;
    hipStream_t s;
    const t2v_gen_desc& g;
    const std::vector<LayerSpec>& specs;
    const t2v_layer* layers;
    Buffers& b;
    int li = 0;
    int sc = 0;   // which per-stream scratch set this runner uses

    // conv (+ fused stats) -> finalize -> apply.  y receives the conv output and is normalised in
    // place: y = [relu](norm(conv(x))) + res1 + res2
    int conv_norm(const float* x, float* y, int relu, const float* res1, const float* res2) {
        const LayerSpec& L = specs[li];
        const t2v_layer& w = layers[li];
        ++li;
        ConvPlan pl;
        const int Cout = L.cd.Cout;
        if (is_winograd(L.cd.algo)) {
            const int M = L.cd.H * L.cd.W;
            T2V_TRY(winograd_forward(ctx, s, &L.cd, x, w.w, w.bias, y, b.stats[sc], b.wino[sc], 7));
            const float* gam = g.norm_affine ? w.gamma : nullptr;
            const float* bet = g.norm_affine ? w.beta : nullptr;
            if (g.norm_affine) T2V_REQUIRE(gam && bet, "layer %d: norm_affine=1 but gamma/beta missing", li - 1);
            const int wm = wino_m(L.cd.algo);
            const int nparts = wino_tiles_padded(&L.cd, L.cd.algo) / (128 / (wm * wm));
            if (inorm_fused_ok(nparts, Cout))   // the apply pass pools the (few) partials itself: no finalize launch
                return launch_inorm_apply_partials(s, y, b.stats[sc], nparts, nparts, 0, M, wm, L.cd.H, L.cd.W, Cout, g.eps,
                                                   gam, bet, res1, res2, y, (long)M, relu);
            T2V_TRY(launch_inorm_finalize_winograd(s, b.stats[sc], wm, L.cd.H, L.cd.W, Cout, g.eps, b.mean_rstd[sc], 1,
                                                   b.fin[sc]));
            return launch_inorm_apply(s, y, b.mean_rstd[sc], gam, bet, res1, res2, y, (long)M, Cout, relu);
        }
        T2V_TRY(build_conv_plan(&L.cd, L.x_cs, true, &pl));
        T2V_TRY(run_conv(ctx, s, pl, x, w.w, w.bias, y, Cout, b.stats[sc]));
        if (pl.tile != kTileStem && inorm_fused_ok(pl.nparts, Cout)) {
            const float* gam = g.norm_affine ? w.gamma : nullptr;
            const float* bet = g.norm_affine ? w.beta : nullptr;
            if (g.norm_affine) T2V_REQUIRE(gam && bet, "layer %d: norm_affine=1 but gamma/beta missing", li - 1);
            return launch_inorm_apply_partials(s, y, b.stats[sc], pl.nparts, pl.kp.mtiles, pl.BM, pl.kp.M, 0, 0, 0, Cout, g.eps,
                                               gam, bet, res1, res2, y, (long)pl.Hout * pl.Wout, relu);
        }
        if (pl.tile == kTileStem)
            T2V_TRY(launch_inorm_finalize_tiles(s, b.stats[sc], 16, L.cd.H, L.cd.W, Cout, g.eps, b.mean_rstd[sc], 1, b.fin[sc]));
        else
            T2V_TRY(launch_inorm_finalize(s, b.stats[sc], pl.nparts, pl.kp.mtiles, pl.BM, pl.kp.M, Cout, g.eps, b.mean_rstd[sc],
                                          b.fin[sc]));
        const float* gamma = g.norm_affine ? w.gamma : nullptr;
        const float* beta = g.norm_affine ? w.beta : nullptr;
        if (g.norm_affine) T2V_REQUIRE(gamma && beta, "layer %d: norm_affine=1 but gamma/beta missing", li - 1);
        return launch_inorm_apply(s, y, b.mean_rstd[sc], gamma, beta, res1, res2, y, (long)pl.Hout * pl.Wout, Cout, relu);
    }

    int head(const float* x, float* y) {
        const LayerSpec& L = specs[li];
        const t2v_layer& w = layers[li];
        ++li;
        ConvPlan pl;
        T2V_TRY(build_conv_plan(&L.cd, L.x_cs, false, &pl));
        return run_conv(ctx, s, pl, x, w.w, w.bias, y, 4, nullptr);
    }

    // x + [pad1,conv3,N,ReLU,pad1,conv3,N](x) (+ extra)
    int resblock(const float* x, float* t, float* y, const float* extra) {
        T2V_TRY(conv_norm(x, t, 1, nullptr, nullptr));
        return conv_norm(t, y, 0, x, extra);
    }

    // One F(4x4) Winograd conv of a chain, up to and including the finalize of its norm statistics; the norm itself
    // is left to the consumer.  `lz` (or null for a plain input map) describes the norm layer the INPUT still has to
    // go through: the previous conv's, whose (mean, rstd) are in mean_rstd[sc] until this conv's finalize replaces them.
    struct LazyIn {
        const t2v_layer* norm;
        int relu;
        const float* res;
        float* xout;
    };
    int wino4_conv_stats(const float* x, const LazyIn* lz, float* y_raw) {
        const LayerSpec& L = specs[li];
        const t2v_layer& w = layers[li];
        ++li;
        const t2v_conv_desc& cd = L.cd;
        if (g.norm_affine) T2V_REQUIRE(w.gamma && w.beta, "layer %d: norm_affine=1 but gamma/beta missing", li - 1);
        if (lz)
            T2V_TRY(launch_winograd4_input_lazy(s, x, b.wino[sc], cd.H, cd.W, cd.Cin, cd.pad, cd.pad_mode == T2V_PAD_REFLECT,
                                                b.mean_rstd[sc], g.norm_affine ? lz->norm->gamma : nullptr,
                                                g.norm_affine ? lz->norm->beta : nullptr, lz->relu, lz->res, lz->xout));
        T2V_TRY(winograd_forward(ctx, s, &cd, x, w.w, w.bias, y_raw, b.stats[sc], b.wino[sc], lz ? 6 : 7));
        return launch_inorm_finalize_winograd(s, b.stats[sc], 4, cd.H, cd.W, cd.Cout, g.eps, b.mean_rstd[sc], 1, b.fin[sc]);
    }

    // The same chain with every norm applied by its consumer: the next conv's input transform normalises (and adds
    // the residual) on the fly, and writes the block output the following block needs as ITS residual on the side.
    // Only the last norm of the chain runs as an apply pass.  Per block: 8 launches instead of 10, and one read +
    // one write of the map less per conv.  tmp: raw conv1 / conv2 outputs, block outputs (alternating).
    int res_chain_lazy(const float* x, int count, float* tmp[4], const float* extra, const float** out) {
        const float* cur = x;
        const t2v_layer* pend = nullptr;   // norm layer of the conv output waiting in tmp[1]
        for (int i = 0; i < count; ++i) {
            if (i == 0) {
                T2V_TRY(wino4_conv_stats(x, nullptr, tmp[0]));
            } else {
                float* xi = tmp[2 + (i & 1)];
                const LazyIn in{pend, 0, cur, xi};
                T2V_TRY(wino4_conv_stats(tmp[1], &in, tmp[0]));
                cur = xi;
            }
            const LazyIn mid{&layers[li - 1], 1, nullptr, nullptr};
            T2V_TRY(wino4_conv_stats(tmp[0], &mid, tmp[1]));
            pend = &layers[li - 1];
        }
        const LayerSpec& L = specs[li - 1];
        T2V_TRY(launch_inorm_apply(s, tmp[1], b.mean_rstd[sc], g.norm_affine ? pend->gamma : nullptr,
                                   g.norm_affine ? pend->beta : nullptr, cur, extra, tmp[1], (long)L.cd.H * L.cd.W,
                                   L.cd.Cout, 0));
        *out = tmp[1];
        return T2V_OK;
    }

    // chain of `count` resblocks starting from x (never written); result pointer in *out.
    // `extra` is added to the output of the LAST block.  tmp: 4 distinct buffers != x.
    int res_chain(const float* x, int count, float* tmp[4], const float* extra, const float** out) {
        static const bool lazy = !(getenv("T2V_CHAIN_LAZY") && atoi(getenv("T2V_CHAIN_LAZY")) == 0);
        if (lazy && count > 0 && specs[li].cd.algo == T2V_ALGO_WINOGRAD_F4) return res_chain_lazy(x, count, tmp, extra, out);
        const float* cur = x;
        for (int i = 0; i < count; ++i) {
            float* t = tmp[0];
            float* y = (cur == tmp[1]) ? tmp[2] : tmp[1];
            T2V_TRY(resblock(cur, t, y, i == count - 1 ? extra : nullptr));
            cur = y;
        }
        *out = cur;
        return T2V_OK;
    }

    // c7,N,R, (d,N,R) x n, RB x nb ; `extra` added to the final output
    int encoder(const float* x, float** act, int nb, float* tmp[4], const float* extra, const float** out) {
        const int n = g.is_local ? 1 : g.n_downsample;
        T2V_TRY(conv_norm(x, act[0], 1, nullptr, nullptr));
        for (int i = 0; i < n; ++i) {
            const bool last = (i == n - 1) && nb == 0;
            T2V_TRY(conv_norm(act[i], act[i + 1], 1, last ? extra : nullptr, nullptr));
        }
        if (nb == 0) {
            *out = act[n];
            return T2V_OK;
        }
        return res_chain(act[n], nb, tmp, extra, out);
    }

    int decoder(const float* x, float** dec, const float** out) {
        const int n = g.is_local ? 1 : g.n_downsample;
        const float* cur = x;
        for (int i = 0; i < n; ++i) {
            float* y = dec[n - 1 - i];
            T2V_TRY(conv_norm(cur, y, 1, nullptr, nullptr));
            cur = y;
        }
        *out = cur;
        return T2V_OK;
    }
};

}  // namespace
}  // namespace t2v

using namespace t2v;

extern "C" {

int t2v_generator_num_layers(const t2v_gen_desc* d) {
    if (check_desc(d) != T2V_OK) return -1;
    std::vector<LayerSpec> L;
    enumerate_layers(*d, L);
    return (int)L.size();
}

int t2v_generator_layer_desc(const t2v_gen_desc* d, int i, t2v_conv_desc* out, int* x_cs) {
    T2V_TRY(check_desc(d));
    std::vector<LayerSpec> L;
    enumerate_layers(*d, L);
    T2V_REQUIRE(i >= 0 && i < (int)L.size() && out, "layer index %d out of range", i);
    *out = L[i].cd;
    if (x_cs) *x_cs = L[i].x_cs;
    return T2V_OK;
}

size_t t2v_generator_workspace_bytes(const t2v_gen_desc* d) {
    if (check_desc(d) != T2V_OK) return 0;
    std::vector<LayerSpec> L;
    enumerate_layers(*d, L);
    Arena a{nullptr, 0};
    Buffers b;
    plan_buffers(*d, L, a, b);
    return a.off;
}

int t2v_generator_forward(t2v_ctx* ctx, void* stream, const t2v_gen_desc* d, const t2v_layer* layers, int n_layers,
                          const t2v_gen_io* io, void* workspace, size_t ws_bytes) {
    T2V_REQUIRE(ctx && layers && io && workspace, "generator_forward: null pointer");
    T2V_TRY(check_desc(d));
    std::vector<LayerSpec> specs;
    enumerate_layers(*d, specs);
    T2V_REQUIRE(n_layers == (int)specs.size(), "generator_forward: expected %d layers, got %d", (int)specs.size(),
                n_layers);
    T2V_REQUIRE(io->pose && io->prev && io->out, "generator_forward: pose/prev/out must be set");
    T2V_REQUIRE(((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
    if (d->is_local) {
        T2V_REQUIRE(io->coarse_img_feat, "local generator needs coarse_img_feat");
        T2V_REQUIRE(d->no_flow || io->coarse_flow_feat, "local generator with flow needs coarse_flow_feat");
    }
    for (int i = 0; i < n_layers; ++i)
        T2V_REQUIRE(layers[i].w && layers[i].bias, "layer %d: weight/bias pointer missing", i);
    Arena a{reinterpret_cast<char*>(workspace), ws_bytes};
    Buffers b;
    plan_buffers(*d, specs, a, b);
    if (a.overflow) {
        set_error("generator_forward: workspace %zu bytes < required %zu", ws_bytes, a.off);
        return T2V_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    // Two streams: the pose encoder and the previous-frame encoder are independent until their sum, and so
    // are the image and flow branches after it.  Each branch is a chain of ~50-100 kernels, a third of them
    // small (norm finalize / apply, Winograd transforms: 5-13 us, launch- and tail-bound); run side by side
    // the other branch's GEMM blocks fill those gaps.  T2V_STREAMS=1 runs everything on the caller's stream.
    static const bool two_streams = !(getenv("T2V_STREAMS") && atoi(getenv("T2V_STREAMS")) == 1);
    hipStream_t s2 = two_streams ? ctx->side : s;
    auto fork = [&]() -> int {
        if (!two_streams) return T2V_OK;
        T2V_HIP_CHECK(hipEventRecord(ctx->ev_fork, s));
        T2V_HIP_CHECK(hipStreamWaitEvent(s2, ctx->ev_fork, 0));
        return T2V_OK;
    };
    auto join = [&]() -> int {
        if (!two_streams) return T2V_OK;
        T2V_HIP_CHECK(hipEventRecord(ctx->ev_join, s2));
        T2V_HIP_CHECK(hipStreamWaitEvent(s, ctx->ev_join, 0));
        return T2V_OK;
    };
    const int n = d->is_local ? 1 : d->n_downsample;
    const int G = d->ngf;
    const size_t bott = (size_t)(d->H >> n) * (d->W >> n) * (G << n);
    const int nb_enc = d->is_local ? 0 : d->n_blocks - d->n_blocks / 2;
    const int nb_res = d->is_local ? d->n_blocks : d->n_blocks / 2;
    const int enc_layers = 1 + n + 2 * nb_enc;                 // layers of one encoder
    const int branch_layers = 2 * nb_res + n + 1;              // res trunk + decoder + head of one branch
    Runner r{ctx, s, *d, specs, layers, b};
    Runner r2{ctx, s2, *d, specs, layers, b};
    r2.sc = two_streams ? 1 : 0;

    // d = model_down_seg(x) + model_down_img(prev)
    float* tmpA[4] = {b.bt[0], b.bt[1], b.bt[2], b.bt[3]};
    float* tmpB[4] = {b.bt2[0], b.bt2[1], b.bt2[2], b.bt2[3]};
    const float *segout, *imgout;
    T2V_TRY(fork());
    r2.li = enc_layers;
    T2V_TRY(r2.encoder(io->prev, b.encB, nb_enc, tmpB, nullptr, &imgout));
    T2V_TRY(r.encoder(io->pose, b.encA, nb_enc, tmpA, nullptr, &segout));
    T2V_TRY(join());
    T2V_TRY(launch_add(s, imgout, segout, b.d, (long)bott));   // (norm + x) + seg: the order the fused form summed in
    const float* dsum = b.d;
    r.li = 2 * enc_layers;

    const float* img_in = dsum;
    const float* flow_in = dsum;
    if (d->is_local) {
        T2V_TRY(launch_add(s, dsum, io->coarse_img_feat, b.dimg, (long)bott));
        img_in = b.dimg;
        if (!d->no_flow) {
            T2V_TRY(launch_add(s, dsum, io->coarse_flow_feat, b.dflow, (long)bott));
            flow_in = b.dflow;
        }
    }
    const bool blend = !(d->no_flow || io->use_raw_only);
    float* raw = io->raw ? io->raw : (blend ? b.raw : io->out);
    float* fw = io->flow_w ? io->flow_w : b.fw;
    if (!d->no_flow) {
        // flow branch on the side stream (temporaries bt2: the encoders are done with them)
        T2V_TRY(fork());
        r2.li = 2 * enc_layers + branch_layers;
        const float *res_flow, *flow_feat;
        T2V_TRY(r2.res_chain(flow_in, nb_res, tmpB, nullptr, &res_flow));
        T2V_TRY(r2.decoder(res_flow, b.decF, &flow_feat));
        T2V_TRY(r2.head(flow_feat, fw));
        if (io->flow_feat)
            T2V_HIP_CHECK(hipMemcpyAsync(io->flow_feat, flow_feat, (size_t)d->H * d->W * G * sizeof(float),
                                         hipMemcpyDeviceToDevice, s2));
    }
    const float *res_img, *img_feat;
    T2V_TRY(r.res_chain(img_in, nb_res, tmpA, nullptr, &res_img));
    T2V_TRY(r.decoder(res_img, b.decI, &img_feat));
    T2V_TRY(r.head(img_feat, raw));
    if (io->img_feat)
        T2V_HIP_CHECK(hipMemcpyAsync(io->img_feat, img_feat, (size_t)d->H * d->W * G * sizeof(float),
                                     hipMemcpyDeviceToDevice, s));
    if (!d->no_flow) {
        T2V_TRY(join());
        r.li += branch_layers;
        T2V_REQUIRE(r2.li == n_layers, "internal: flow branch consumed up to layer %d of %d", r2.li, n_layers);
        if (blend) {
            const int prev_cs = round_up(d->prev_nc, 4);
            T2V_TRY(launch_warp_composite(s, raw, fw, io->prev, prev_cs, d->prev_nc - 3, io->out, nullptr, d->H,
                                          d->W));
        }
    }
    if (!blend && raw != io->out)
        T2V_HIP_CHECK(hipMemcpyAsync(io->out, raw, (size_t)d->H * d->W * 4 * sizeof(float), hipMemcpyDeviceToDevice,
                                     s));
    T2V_REQUIRE(r.li == n_layers, "internal: consumed %d of %d layers", r.li, n_layers);
    return T2V_OK;
}

}  // extern "C"
