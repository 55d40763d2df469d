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
        for (int i = 0; i < n; ++i) {
            t2v_conv_desc cd = mk_conv(H >> i, W >> i, G << i, G << (i + 1), 3, 2, 1, T2V_PAD_ZERO, 0);
            // the deep stride-2 layers as polyphase Winograd F(4,2) (polyphase.hip) where that is the faster form
            if (g.conv_algo == 0 && polyphase_pays(&cd, G << i)) cd.algo = T2V_ALGO_POLYPHASE;
            out.push_back({cd, G << i, true});
        }
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
            t2v_conv_desc cd = mk_conv(H >> l, W >> l, G << l, G << (l - 1), 3, 2, 1, T2V_PAD_ZERO, 1);
            if (g.conv_algo == 0 && polyphase_pays(&cd, G << l)) cd.algo = T2V_ALGO_POLYPHASE;
            out.push_back({cd, G << l, true});
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

constexpr int kMaxBatch = T2V_MAX_BATCH;

// Every buffer holds the `nimg` images of a batch back to back (image stride = the single-image size), so the
// batched kernels of the ResnetBlock chains address image i at base + i*stride and the per-image launches of the
// other layers do the same.
struct Buffers {
    float *encA[8], *encB[8];  // encoder activations per level (A: pose/seg, B: prev-image)
    float* bt[4];              // bottleneck temporaries (resnet chains)
    float* bt2[4];             // ... of the branch that runs on the side stream
    float* d;                  // encoder sum
    float *dimg, *dflow;       // local generator: d + coarse features
    float *decI[8], *decF[8];  // decoder activations per level
    float *raw, *fw;
    size_t lvl[8];             // floats of one image at level l
    size_t bott;               // = lvl[n]
    // per-stream scratch [0]: caller's stream, [1]: side stream; one slot per image
    float* stats[2];
    float* mean_rstd[2];
    float* wino[2];   // Winograd scratch: transformed input V + transformed output M of the whole batch
    double* fin[2];   // norm finalize scratch (pooled moments per group of partials)
    size_t stats_stride, mr_stride, fin_stride;   // floats (doubles for fin) between the images' slots
};

void plan_buffers(const t2v_gen_desc& g, const std::vector<LayerSpec>& layers, int nimg, Arena& a, Buffers& b) {
    const int G = g.ngf, n = g.is_local ? 1 : g.n_downsample;
    auto lvl = [&](int l) { return (size_t)(g.H >> l) * (g.W >> l) * (G << l); };
    const size_t N = (size_t)nimg;
    for (int l = 0; l <= n; ++l) {
        b.lvl[l] = lvl(l);
        b.encA[l] = a.alloc(N * lvl(l));
        b.encB[l] = a.alloc(N * lvl(l));
    }
    b.bott = lvl(n);
    for (int i = 0; i < 4; ++i) b.bt[i] = a.alloc(N * lvl(n));
    for (int i = 0; i < 4; ++i) b.bt2[i] = a.alloc(N * lvl(n));
    b.d = a.alloc(N * lvl(n));
    b.dimg = a.alloc(N * lvl(n));
    b.dflow = a.alloc(N * lvl(n));
    for (int l = 0; l < n; ++l) {
        b.decI[l] = a.alloc(N * lvl(l));
        b.decF[l] = g.no_flow ? nullptr : a.alloc(N * lvl(l));
    }
    b.raw = a.alloc(N * g.H * g.W * 4);
    b.fw = a.alloc(N * g.H * g.W * 4);
    size_t max_stats = 0;
    int max_c = 4;
    for (const LayerSpec& L : layers) {
        ConvPlan pl;
        if (L.has_norm && is_winograd(L.cd.algo)) {
            const int m = wino_m(L.cd.algo);
            const size_t s = (size_t)(wino_tiles_padded(&L.cd, L.cd.algo) * m * m / 128) * L.cd.Cout * 2;
            if (s > max_stats) max_stats = s;
        } else if (L.has_norm && L.cd.algo == T2V_ALGO_POLYPHASE) {
            const size_t s = (size_t)(poly_tiles_padded(&L.cd) * poly_m(&L.cd) * poly_m(&L.cd) / 128) * L.cd.Cout * 2;
            if (s > max_stats) max_stats = s;
        } else if (L.has_norm && build_conv_plan(&L.cd, L.x_cs, true, &pl) == T2V_OK) {
            const size_t s = (size_t)pl.nparts * L.cd.Cout * 2;
            if (s > max_stats) max_stats = s;
        }
        if (L.cd.Cout > max_c) max_c = L.cd.Cout;
    }
    b.stats_stride = (max_stats + 63) / 64 * 64;
    b.mr_stride = (size_t)max_c * 2;
    b.fin_stride = (size_t)kFinalizeMaxGroups * max_c * 4;
    for (int k = 0; k < 2; ++k) {
        b.stats[k] = a.alloc(N * b.stats_stride);
        b.mean_rstd[k] = a.alloc(N * b.mr_stride);
        b.fin[k] = reinterpret_cast<double*>(a.alloc(N * b.fin_stride * 2));
    }
    size_t max_wino = 0;
    for (const LayerSpec& L : layers)
        if (is_winograd(L.cd.algo)) {
            const size_t w = winograd_workspace_floats(&L.cd, L.cd.algo == T2V_ALGO_WINOGRAD_F4 ? nimg : 1);
            if (w > max_wino) max_wino = w;
        } else if (L.cd.algo == T2V_ALGO_POLYPHASE) {
            const size_t w = polyphase_workspace_floats(&L.cd);
            if (w > max_wino) max_wino = w;
        }
    for (int k = 0; k < 2; ++k) b.wino[k] = max_wino ? a.alloc(max_wino) : nullptr;
}

// the images of a batch as pointers (maps that are not laid out back to back: the caller's inputs and outputs)
struct Ptrs {
    const float* p[kMaxBatch];
};
struct MutPtrs {
    float* p[kMaxBatch];
    operator Ptrs() const {
        Ptrs q;
        for (int i = 0; i < kMaxBatch; ++i) q.p[i] = p[i];
        return q;
    }
};

struct Runner {
    t2v_ctx* ctx;
    hipStream_t s;
    const t2v_gen_desc& g;
    const std::vector<LayerSpec>& specs;
    const t2v_layer* layers;
    Buffers& b;
    int nimg;
    int li = 0;
    int sc = 0;   // which per-stream scratch set this runner uses

    Ptrs at(const float* base, size_t stride) const {
        Ptrs q{};
        for (int i = 0; i < nimg; ++i) q.p[i] = base ? base + (size_t)i * stride : nullptr;
        return q;
    }
    MutPtrs at(float* base, size_t stride) const {
        MutPtrs q{};
        for (int i = 0; i < nimg; ++i) q.p[i] = base ? base + (size_t)i * stride : nullptr;
        return q;
    }
    float* stats_of(int im) const { return b.stats[sc] + (size_t)im * b.stats_stride; }
    float* mr_of(int im) const { return b.mean_rstd[sc] + (size_t)im * b.mr_stride; }
    double* fin_of(int im) const { return b.fin[sc] + (size_t)im * b.fin_stride; }
    // conv (+ fused stats) -> finalize -> apply of layer `l` for ONE image.  y receives the conv output and is
    // normalised in place: y = [relu](norm(conv(x))) + res1 + res2
    // lazy_in: x is layer l-1's raw conv output, whose (mean, rstd) still sit in mr_of(im) -- this (polyphase) layer's input
    // transform normalises it; lazy_out: leave y raw for the next layer to do the same (no apply pass: a read and a write of
    // the map less).  Same arithmetic in the same order: the frames are bit-identical to the apply form (T2V_CHAIN_LAZY=0).
    int conv_norm_one(int l, int im, const float* x, float* y, int relu, const float* res1, const float* res2,
                      bool lazy_in = false, bool lazy_out = false) {
        const LayerSpec& L = specs[l];
        const t2v_layer& w = layers[l];
        ConvPlan pl;
        const int Cout = L.cd.Cout;
        float* stats = stats_of(im);
        float* mr = mr_of(im);
        const float* gam = g.norm_affine ? w.gamma : nullptr;
        const float* bet = g.norm_affine ? w.beta : nullptr;
        if (g.norm_affine) T2V_REQUIRE(gam && bet, "layer %d: norm_affine=1 but gamma/beta missing", l);
        if (is_winograd(L.cd.algo)) {
            const int M = L.cd.H * L.cd.W;
            const int wm = wino_m(L.cd.algo);
            WinoBatch wb;
            T2V_TRY(winograd_forward(ctx, s, &L.cd, x, w.w, w.bias, y, stats, b.wino[sc], 7, &wb));
            T2V_TRY(launch_inorm_finalize_winograd(s, stats, wm, L.cd.H, L.cd.W, Cout, g.eps, mr, 1, fin_of(im)));
            return launch_inorm_apply(s, y, mr, gam, bet, res1, res2, y, (long)M, Cout, relu);
        }
        T2V_REQUIRE(!lazy_out || (relu == 1 && !res1 && !res2), "internal: a lazy output carries a plain norm + ReLU");
        if (L.cd.algo == T2V_ALGO_POLYPHASE) {
            const int Ho = poly_out_h(&L.cd), Wo = poly_out_w(&L.cd);
            const PolyLazyNorm ln{mr, g.norm_affine ? layers[l - 1].gamma : nullptr, g.norm_affine ? layers[l - 1].beta : nullptr, 1};
            T2V_TRY(polyphase_forward(ctx, s, &L.cd, x, w.w, w.bias, y, stats, b.wino[sc], 7, lazy_in ? &ln : nullptr));
            T2V_TRY(launch_inorm_finalize_winograd(s, stats, poly_m(&L.cd), Ho, Wo, Cout, g.eps, mr, 1, fin_of(im)));
            if (lazy_out) return T2V_OK;
            return launch_inorm_apply(s, y, mr, gam, bet, res1, res2, y, (long)Ho * Wo, Cout, relu);
        }
        T2V_REQUIRE(!lazy_in, "internal: only the polyphase input transform applies a pending norm");
        T2V_TRY(build_conv_plan(&L.cd, L.x_cs, true, &pl));
        T2V_TRY(run_conv(ctx, s, pl, x, w.w, w.bias, y, Cout, stats));
        if (pl.tile == kTileStem)
            T2V_TRY(launch_inorm_finalize_tiles(s, stats, 16, L.cd.H, L.cd.W, Cout, g.eps, mr, 1, fin_of(im)));
        else
            T2V_TRY(launch_inorm_finalize(s, stats, pl.nparts, pl.kp.mtiles, pl.BM, pl.kp.M, Cout, g.eps, mr, fin_of(im)));
        if (lazy_out) return T2V_OK;
        return launch_inorm_apply(s, y, mr, gam, bet, res1, res2, y, (long)pl.Hout * pl.Wout, Cout, relu);
    }
    // the next layer for every image of the batch (one launch sequence per image).  (The images of a lock-step batch as
    // blockIdx.y of ONE launch of the stride-2 / transposed convs -- run_conv_batch, what the train step's discriminators
    // use -- was measured here and dropped: 142.1 / 141.9 vs 142.1 / 141.7 fps for two 512x320 sequences, 67.0 / 66.7 vs
    // 66.8 / 66.7 at 512x680, 92.0 / 92.0 vs 91.8 / 91.7 at 512x512, alternating runs: inside two-stream frames the other
    // stream already fills what a 2.5-blocks-per-CU launch leaves idle.)
    int conv_norm(const Ptrs& x, const MutPtrs& y, int relu, const Ptrs& res1, bool lazy_in = false, bool lazy_out = false) {
        for (int im = 0; im < nimg; ++im)
            T2V_TRY(conv_norm_one(li, im, x.p[im], y.p[im], relu, res1.p[im], nullptr, lazy_in, lazy_out));
        ++li;
        return T2V_OK;
    }
    // the layer after the current one is a polyphase layer that consumes this one's output and nothing else does
    bool next_takes_raw() const { return options().chain_lazy && specs[li + 1].cd.algo == T2V_ALGO_POLYPHASE; }

    int head(const Ptrs& x, const MutPtrs& y) {
        const LayerSpec& L = specs[li];
        const t2v_layer& w = layers[li];
        ++li;
        ConvPlan pl;
        T2V_TRY(build_conv_plan(&L.cd, L.x_cs, false, &pl));
        for (int im = 0; im < nimg; ++im) T2V_TRY(run_conv(ctx, s, pl, x.p[im], w.w, w.bias, y.p[im], 4, nullptr));
        return T2V_OK;
    }

    // x + [pad1,conv3,N,ReLU,pad1,conv3,N](x), image by image
    int resblock(const Ptrs& x, const MutPtrs& t, const MutPtrs& y) {
        T2V_TRY(conv_norm(x, t, 1, Ptrs{}));
        return conv_norm(t, y, 0, x);
    }

    // One F(4x4) Winograd conv of a chain over the whole batch (maps `bott` floats apart), up to and including the
    // finalize of its norm statistics; the norm itself is left to the consumer.  `lz` (or null for a plain input map)
    // describes the norm layer the INPUT still has to go through: the previous conv's, whose (mean, rstd) tables are in
    // mean_rstd[sc] until this conv's finalize replaces them.
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
        T2V_REQUIRE(b.mr_stride == (size_t)2 * cd.Cout, "internal: chain scratch layout");
        if (lz)
            T2V_TRY(launch_winograd4_input_lazy(s, x, b.wino[sc], cd.H, cd.W, cd.Cin, cd.pad, cd.pad_mode == T2V_PAD_REFLECT,
                                                b.mean_rstd[sc], g.norm_affine ? lz->norm->gamma : nullptr,
                                                g.norm_affine ? lz->norm->beta : nullptr, lz->relu, lz->res, lz->xout, nimg,
                                                (long)b.bott));
        WinoBatch wb;
        wb.nimg = nimg;
        wb.img_stride_x = (long)b.bott;
        // the images' partials are packed back to back by the batched output transform: nparts*Cout*2 floats each
        T2V_TRY(winograd_forward(ctx, s, &cd, x, w.w, w.bias, y_raw, b.stats[sc], b.wino[sc], lz ? 6 : 7, &wb));
        const size_t per_img = (size_t)(wino_tiles_padded(&cd, cd.algo) / 8) * cd.Cout * 2;
        for (int im = 0; im < nimg; ++im)
            T2V_TRY(launch_inorm_finalize_winograd(s, b.stats[sc] + im * per_img, 4, cd.H, cd.W, cd.Cout, g.eps, mr_of(im), 1,
                                                   fin_of(im)));
        return T2V_OK;
    }

    // The same chain with every norm applied by its consumer: the next conv's input transform normalises (and adds
    // the residual) on the fly, and writes the block output the following block needs as ITS residual on the side.
    // Only the last norm of the chain runs as an apply pass.  Per block: 8 launches instead of 10 for the WHOLE batch, and one read + one write of the map less
    // per conv.  tmp: raw conv1 / conv2 outputs, block outputs (alternating); all hold the batch back to back.
    int res_chain_lazy(const float* x, int count, float* tmp[4], const float** out) {
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
                                   g.norm_affine ? pend->beta : nullptr, cur, nullptr, tmp[1], (long)L.cd.H * L.cd.W,
                                   L.cd.Cout, 0, nimg));
        *out = tmp[1];
        return T2V_OK;
    }

    // chain of `count` resblocks starting from x (never written; the batch back to back, `bott` floats apart); result
    // pointer in *out.  tmp: 4 distinct buffers != x.
    int res_chain(const float* x, int count, float* tmp[4], const float** out) {
        if (options().chain_lazy && count > 0 && specs[li].cd.algo == T2V_ALGO_WINOGRAD_F4 && b.mr_stride == (size_t)2 * specs[li].cd.Cout)
            return res_chain_lazy(x, count, tmp, out);
        const float* cur = x;
        for (int i = 0; i < count; ++i) {
            float* t = tmp[0];
            float* y = (cur == tmp[1]) ? tmp[2] : tmp[1];
            T2V_TRY(resblock(at(cur, b.bott), at(t, b.bott), at(y, b.bott)));
            cur = y;
        }
        *out = cur;
        return T2V_OK;
    }

    // c7,N,R, (d,N,R) x n, RB x nb
    int encoder(const Ptrs& x, float** act, int nb, float* tmp[4], const float** out) {
        const int n = g.is_local ? 1 : g.n_downsample;
        bool raw = n > 0 && next_takes_raw();      // (the stem's output feeds the first stride-2 layer only)
        T2V_TRY(conv_norm(x, at(act[0], b.lvl[0]), 1, Ptrs{}, false, raw));
        for (int i = 0; i < n; ++i) {
            const bool raw_out = i + 1 < n && next_takes_raw();
            T2V_TRY(conv_norm(at(act[i], b.lvl[i]), at(act[i + 1], b.lvl[i + 1]), 1, Ptrs{}, raw, raw_out));
            raw = raw_out;
        }
        if (nb == 0) {
            *out = act[n];
            return T2V_OK;
        }
        return res_chain(act[n], nb, tmp, out);
    }

    int decoder(const float* x, float** dec, const float** out) {
        const int n = g.is_local ? 1 : g.n_downsample;
        const float* cur = x;
        size_t cur_stride = b.bott;
        bool raw = false;
        for (int i = 0; i < n; ++i) {
            const int l = n - 1 - i;
            float* y = dec[l];
            const bool raw_out = i + 1 < n && next_takes_raw();
            T2V_TRY(conv_norm(at(cur, cur_stride), at(y, b.lvl[l]), 1, Ptrs{}, raw, raw_out));
            raw = raw_out;
            cur = y;
            cur_stride = b.lvl[l];
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

size_t t2v_generator_workspace_bytes_batch(const t2v_gen_desc* d, int batch) {
    if (check_desc(d) != T2V_OK || batch < 1 || batch > kMaxBatch) return 0;
    std::vector<LayerSpec> L;
    enumerate_layers(*d, L);
    Arena a{nullptr, 0};
    Buffers b;
    plan_buffers(*d, L, batch, a, b);
    return a.off;
}
size_t t2v_generator_workspace_bytes(const t2v_gen_desc* d) { return t2v_generator_workspace_bytes_batch(d, 1); }

int t2v_generator_forward_batch(t2v_ctx* ctx, void* stream, const t2v_gen_desc* d, const t2v_layer* layers, int n_layers,
                                const t2v_gen_io* ios, int batch, void* workspace, size_t ws_bytes) {
    T2V_REQUIRE(ctx && layers && ios && workspace, "generator_forward: null pointer");
    T2V_REQUIRE(batch >= 1 && batch <= kMaxBatch, "generator_forward: batch %d out of range [1,%d]", batch, kMaxBatch);
    T2V_TRY(check_desc(d));
    T2V_TRY(check_async_errors());      // a hand-over that timed out in an earlier frame is reported here
    std::vector<LayerSpec> specs;
    enumerate_layers(*d, specs);
    T2V_REQUIRE(n_layers == (int)specs.size(), "generator_forward: expected %d layers, got %d", (int)specs.size(),
                n_layers);
    T2V_REQUIRE(((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
    for (int im = 0; im < batch; ++im) {
        const t2v_gen_io* io = ios + im;
        T2V_REQUIRE(io->pose && io->prev && io->out, "generator_forward: pose/prev/out must be set (image %d)", im);
        if (d->is_local) {
            T2V_REQUIRE(io->coarse_img_feat, "local generator needs coarse_img_feat");
            T2V_REQUIRE(d->no_flow || io->coarse_flow_feat, "local generator with flow needs coarse_flow_feat");
        }
    }
    for (int i = 0; i < n_layers; ++i)
        T2V_REQUIRE(layers[i].w && layers[i].bias, "layer %d: weight/bias pointer missing", i);
    Arena a{reinterpret_cast<char*>(workspace), ws_bytes};
    Buffers b;
    plan_buffers(*d, specs, batch, a, b);
    if (a.overflow) {
        set_error("generator_forward: workspace %zu bytes < required %zu", ws_bytes, a.off);
        return T2V_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    // Two streams: the pose encoder and the previous-frame encoder are independent until their sum, and so
    // are the image and flow branches after it.  Each branch is a chain of ~50-100 kernels, a third of them
    // small (norm finalize / apply, Winograd transforms: 5-13 us, launch- and tail-bound); run side by side
    // the other branch's GEMM blocks fill those gaps.  T2V_STREAMS=1 runs everything on the caller's stream.
    // Not where every kernel fills the chip by itself: a single-scale 1024x1024 frame (128x128 bottleneck, 1024 Winograd
    // tiles per position; whole-chip launches of 0.5-1.1 ms) is 1.6-2.2 % FASTER on one stream (41.5 vs 42.2 ms, no flow
    // 31.8 vs 32.5), 768x768 and everything below it 1-5 % slower -- so the default (T2V_STREAMS unset / 0) decides by size.
    const int bott_tiles = ((d->H >> d->n_downsample) + 3) / 4 * (((d->W >> d->n_downsample) + 3) / 4);
    const bool two_streams = options().streams == 2 || (options().streams == 0 && (d->is_local || bott_tiles < 1024));
    const OverlapScope overlap(two_streams);     // (kernels that leave the other stream wave slots, where they exist)
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
    const size_t bott = b.bott;
    const int nb_enc = d->is_local ? 0 : d->n_blocks - d->n_blocks / 2;
    const int nb_res = d->is_local ? d->n_blocks : d->n_blocks / 2;
    const int enc_layers = 1 + n + 2 * nb_enc;                 // layers of one encoder
    const int branch_layers = 2 * nb_res + n + 1;              // res trunk + decoder + head of one branch
    Runner r{ctx, s, *d, specs, layers, b, batch};
    Runner r2{ctx, s2, *d, specs, layers, b, batch};
    r2.sc = two_streams ? 1 : 0;

    Ptrs pose{}, prevp{};
    for (int im = 0; im < batch; ++im) {
        pose.p[im] = ios[im].pose;
        prevp.p[im] = ios[im].prev;
    }
    // d = model_down_seg(x) + model_down_img(prev)
    float* tmpA[4] = {b.bt[0], b.bt[1], b.bt[2], b.bt[3]};
    float* tmpB[4] = {b.bt2[0], b.bt2[1], b.bt2[2], b.bt2[3]};
    const float *segout, *imgout;
    T2V_TRY(fork());
    r2.li = enc_layers;
    T2V_TRY(r2.encoder(prevp, b.encB, nb_enc, tmpB, &imgout));
    T2V_TRY(r.encoder(pose, b.encA, nb_enc, tmpA, &segout));
    T2V_TRY(join());
    T2V_TRY(launch_add(s, imgout, segout, b.d, (long)(batch * bott)));   // (norm + x) + seg: the order the fused form summed in
    const float* dsum = b.d;
    r.li = 2 * enc_layers;

    const float* img_in = dsum;
    const float* flow_in = dsum;
    if (d->is_local) {
        for (int im = 0; im < batch; ++im) {
            T2V_TRY(launch_add(s, dsum + im * bott, ios[im].coarse_img_feat, b.dimg + im * bott, (long)bott));
            if (!d->no_flow)
                T2V_TRY(launch_add(s, dsum + im * bott, ios[im].coarse_flow_feat, b.dflow + im * bott, (long)bott));
        }
        img_in = b.dimg;
        if (!d->no_flow) flow_in = b.dflow;
    }
    const size_t px4 = (size_t)d->H * d->W * 4, feat = (size_t)d->H * d->W * G;
    MutPtrs raw{}, fw{};
    bool blend[kMaxBatch];
    for (int im = 0; im < batch; ++im) {
        blend[im] = !(d->no_flow || ios[im].use_raw_only);
        raw.p[im] = ios[im].raw ? ios[im].raw : (blend[im] ? b.raw + im * px4 : ios[im].out);
        fw.p[im] = ios[im].flow_w ? ios[im].flow_w : b.fw + im * px4;
    }
    if (!d->no_flow) {
        // flow branch on the side stream (temporaries bt2: the encoders are done with them)
        T2V_TRY(fork());
        r2.li = 2 * enc_layers + branch_layers;
        const float *res_flow, *flow_feat;
        T2V_TRY(r2.res_chain(flow_in, nb_res, tmpB, &res_flow));
        T2V_TRY(r2.decoder(res_flow, b.decF, &flow_feat));
        T2V_TRY(r2.head(r2.at(flow_feat, feat), fw));
        for (int im = 0; im < batch; ++im)
            if (ios[im].flow_feat)
                T2V_HIP_CHECK(hipMemcpyAsync(ios[im].flow_feat, flow_feat + im * feat, feat * sizeof(float),
                                             hipMemcpyDeviceToDevice, s2));
    }
    const float *res_img, *img_feat;
    T2V_TRY(r.res_chain(img_in, nb_res, tmpA, &res_img));
    T2V_TRY(r.decoder(res_img, b.decI, &img_feat));
    T2V_TRY(r.head(r.at(img_feat, feat), raw));
    for (int im = 0; im < batch; ++im)
        if (ios[im].img_feat)
            T2V_HIP_CHECK(hipMemcpyAsync(ios[im].img_feat, img_feat + im * feat, feat * sizeof(float), hipMemcpyDeviceToDevice,
                                         s));
    if (!d->no_flow) {
        T2V_TRY(join());
        r.li += branch_layers;
        T2V_REQUIRE(r2.li == n_layers, "internal: flow branch consumed up to layer %d of %d", r2.li, n_layers);
        const int prev_cs = round_up(d->prev_nc, 4);
        for (int im = 0; im < batch; ++im)
            if (blend[im])
                T2V_TRY(launch_warp_composite(s, raw.p[im], fw.p[im], ios[im].prev, prev_cs, d->prev_nc - 3, ios[im].out,
                                              nullptr, d->H, d->W));
    }
    for (int im = 0; im < batch; ++im)
        if (!blend[im] && raw.p[im] != ios[im].out)
            T2V_HIP_CHECK(hipMemcpyAsync(ios[im].out, raw.p[im], px4 * sizeof(float), hipMemcpyDeviceToDevice, s));
    T2V_REQUIRE(r.li == n_layers, "internal: consumed %d of %d layers", r.li, n_layers);
    return T2V_OK;
}

int t2v_generator_forward(t2v_ctx* ctx, void* stream, const t2v_gen_desc* d, const t2v_layer* layers, int n_layers,
                          const t2v_gen_io* io, void* workspace, size_t ws_bytes) {
    return t2v_generator_forward_batch(ctx, stream, d, layers, n_layers, io, 1, workspace, ws_bytes);
}

}  // extern "C"
