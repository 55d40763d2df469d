// capi.hip -- extern "C" surface of libt2v_hip.so (declared in include/t2v.h) and the conv planner.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <utility>
#include <vector>

#include "conv_plan.h"

namespace t2v {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ---- environment switches, read once ----
static Options g_opts;
static std::once_flag g_opts_once;
static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}
static thread_local int g_overlap = 0;
int set_overlap_hint(int on) {
    const int prev = g_overlap;
    g_overlap = on == 2 ? 2 : (on ? 1 : 0);
    return prev;
}
bool overlap_hint() { return g_overlap == 1 && options().overlap_hint != 0; }
int sk_launch_blocks() {
    const int two = wino_gemm_sk_grid_blocks(), mode = options().sk_blocks_per_cu;
    const bool one = mode == 1 || (mode == 0 && g_overlap == 2 && options().overlap_hint != 0);
    return one && two % 16 == 0 ? two / 2 : two;
}
void options_reload() {
    Options o;
    o.wino_gemm_sk = env_int("T2V_WINO_GEMM_SK", 1);
    o.wino_gemm_sk_ragged = env_int("T2V_WINO_GEMM_SK_RAGGED", 2);
    o.overlap_hint = env_int("T2V_OVERLAP_HINT", 1);
    o.overlap_hint_single = env_int("T2V_OVERLAP_HINT_SINGLE", 0);
    o.wgrad_sk = env_int("T2V_WGRAD_SK", 1);
    o.wgrad_combine = env_int("T2V_WGRAD_COMBINE", 1);
    o.wgrad_combine_max = env_int("T2V_WGRAD_COMBINE_MAX", 4);
    o.chain_lazy = env_int("T2V_CHAIN_LAZY", 1);
    o.streams = env_int("T2V_STREAMS", 0);
    o.conv_tile = env_int("T2V_CONV_TILE", 0);
    o.xcd_slices = env_int("T2V_XCD_SLICES", 1);
    o.sk_blocks_per_cu = env_int("T2V_SK_BLOCKS_PER_CU", 0);
    g_opts = o;
}
const Options& options() {
    std::call_once(g_opts_once, options_reload);
    return g_opts;
}

// ---- fixed-grid hand-over: sticky error word + dispatch-order self-test ----
static unsigned* g_err_word = nullptr;            // pinned, mapped host memory: kernels store to it, the host reads it
static std::atomic<int> g_fixed_grid_on{1};
static std::mutex g_fg_mutex;
static char g_fg_why[160] = "";
unsigned* async_error_word() { return g_err_word; }
bool fixed_grid_enabled() { return g_fixed_grid_on.load(std::memory_order_relaxed) != 0; }
void fixed_grid_disable(const char* why) {
    std::lock_guard<std::mutex> lk(g_fg_mutex);
    if (g_fixed_grid_on.exchange(0)) {
        snprintf(g_fg_why, sizeof(g_fg_why), "%s", why);
        fprintf(stderr, "libt2v_hip: fixed-grid kernels switched off (%s); running one block per tile\n", why);
    }
}
int check_async_errors() {
    if (!g_err_word || *reinterpret_cast<volatile unsigned*>(g_err_word) == 0) return T2V_OK;
    *reinterpret_cast<volatile unsigned*>(g_err_word) = 0;
    fixed_grid_disable("an accumulator hand-over timed out");
    set_error("a fixed-grid kernel's accumulator hand-over timed out in an earlier launch: the output of that launch holds "
              "NaNs (re-run it); the fixed-grid kernels are now off for this process");
    return T2V_ERR_HANDOVER;
}

// blocks shaped like the fixed-grid kernels' (512 threads, 64 KiB of LDS: two per CU), four rounds of them: each draws a
// ticket as it starts and stays resident for some microseconds
__global__ __launch_bounds__(512) void dispatch_order_kernel(unsigned* counter, unsigned* order) {
    extern __shared__ char smem[];
    if (threadIdx.x == 0) {
        order[blockIdx.x] = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        smem[0] = 0;
    }
    for (int i = 0; i < 64; ++i) __builtin_amdgcn_s_sleep(32);
}
static int fixed_grid_selftest() {
    const int grid = 4 * wino_gemm_sk_grid_blocks();
    unsigned* buf = nullptr;
    T2V_HIP_CHECK(hipMalloc(&buf, (size_t)(grid + 1) * sizeof(unsigned)));
    int st = T2V_OK;
    std::vector<unsigned> host(grid + 1);
    if (hipMemset(buf, 0, (size_t)(grid + 1) * sizeof(unsigned)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(dispatch_order_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            65536) != hipSuccess)
        st = T2V_ERR_HIP;
    if (st == T2V_OK) {
        hipLaunchKernelGGL(dispatch_order_kernel, dim3(grid), dim3(512), 65536, 0, buf, buf + 1);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(host.data(), buf, (size_t)(grid + 1) * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess)
            st = T2V_ERR_HIP;
    }
    (void)hipFree(buf);
    if (st != T2V_OK) {
        fixed_grid_disable("the dispatch-order self-test could not run");
        return T2V_OK;      // not fatal: the tile-per-block kernels need no such guarantee
    }
    // Start order per XCD (blocks b = x, x + 8, x + 16, ... run on XCD x under round-robin dispatch).  Blocks dispatched
    // within a microsecond of each other draw their tickets in any order, so the check is on DISPLACEMENT: with in-order
    // dispatch a block's rank among its XCD's tickets differs from its index there by less than the blocks the XCD holds
    // at once (a quarter of this launch's blocks per XCD; the limit is twice that); a queue served out of order (LIFO,
    // random, reversed) displaces by the whole over-subscribed length.
    const int per_xcd = grid / 8, limit = per_xcd / 2;
    for (int x = 0; x < 8 && fixed_grid_enabled(); ++x) {
        std::vector<std::pair<unsigned, int>> t(per_xcd);
        for (int q = 0; q < per_xcd; ++q) t[q] = {host[1 + q * 8 + x], q};
        std::sort(t.begin(), t.end());
        for (int r = 0; r < per_xcd; ++r)
            if (abs(t[r].second - r) > limit) {
                char why[160];
                snprintf(why, sizeof(why), "dispatch-order self-test: block %d of XCD %d started %d places from its index",
                         t[r].second * 8 + x, x, abs(t[r].second - r));
                fixed_grid_disable(why);
                break;
            }
    }
    return T2V_OK;
}

int build_conv_plan(const t2v_conv_desc* d, int x_cs, bool need_stats, ConvPlan* out) {
    T2V_REQUIRE(d && out, "null conv descriptor");
    T2V_REQUIRE(d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "conv: bad dims H=%d W=%d Cin=%d Cout=%d", d->H,
                d->W, d->Cin, d->Cout);
    T2V_REQUIRE(x_cs % 4 == 0 && x_cs >= d->Cin, "conv: input channel storage %d must be a multiple of 4 >= Cin=%d",
                x_cs, d->Cin);
    ConvPlan& pl = *out;
    memset(&pl, 0, sizeof(pl));
    pl.Cin = d->Cin;
    ConvKParams& k = pl.kp;
    // the small-Cout tile has no statistics epilogue; weights are always packed to the 128-row
    // granule so either tile can read them
    // narrow outputs (<= 64 channels: the 1024x1024 local enhancer's stems) use the 64x64 tile: half of a 128-wide
    // tile would be padding
    pl.tile = need_stats ? (d->Cout <= 64 ? kTileQ : kTileL) : (d->Cout > 16 && d->Cout <= 64 ? kTileQ : conv_tile_for(d->Cout));
    conv_tile_dims(pl.tile, &pl.BM, &pl.BN);
    k.Hin = d->H;
    k.Win = d->W;
    k.Cin_s = x_cs;
    k.Cout = d->Cout;
    k.act = d->act;
    k.act_scale = d->act_scale;
    k.ntiles = (d->Cout + pl.BN - 1) / pl.BN;
    pl.Cout_p = round_up(d->Cout, 128);
    int Hm, Wm;
    if (!d->transposed) {
        T2V_REQUIRE(d->kH * d->kW <= kMaxTaps, "conv: kernel %dx%d too large", d->kH, d->kW);
        T2V_REQUIRE(d->stride >= 1 && d->pad >= 0, "conv: bad stride/pad");
        pl.Hout = (d->H + 2 * d->pad - d->kH) / d->stride + 1;
        pl.Wout = (d->W + 2 * d->pad - d->kW) / d->stride + 1;
        T2V_REQUIRE(pl.Hout > 0 && pl.Wout > 0, "conv: empty output");
        if (d->pad_mode == T2V_PAD_REFLECT)
            T2V_REQUIRE(d->pad < d->H && d->pad < d->W, "reflection pad %d needs H,W > pad (H=%d W=%d)", d->pad, d->H,
                        d->W);
        Hm = pl.Hout;
        Wm = pl.Wout;
        k.stride = d->stride;
        k.ostride = 1;
        k.pad_mode = d->pad_mode;
        k.nphases = 1;
        k.KW = d->kW;
        k.pad = d->pad;
        ConvPhase& ph = k.ph[0];
        ph.ntaps = d->kH * d->kW;
        ph.tap0 = 0;
        ph.Kp = round_up(ph.ntaps * x_cs, kBK);
        ph.nk = ph.Kp / kBK;
        ph.w_off = 0;
        ph.oy0 = ph.ox0 = 0;
        for (int kh = 0; kh < d->kH; ++kh)
            for (int kw = 0; kw < d->kW; ++kw) {
                k.tdy[kh * d->kW + kw] = kh - d->pad;
                k.tdx[kh * d->kW + kw] = kw - d->pad;
            }
        pl.wfloats = (size_t)pl.Cout_p * ph.Kp;
    } else {
        T2V_REQUIRE(d->kH == d->kW && (d->kH == 3 || d->kH == 4) && d->stride == 2 && d->pad >= 0 && d->pad <= 2 &&
                        d->output_padding >= 0 && d->output_padding <= 1,
                    "transposed conv: k in {3,4}, stride 2, padding <= 2, output_padding <= 1 are on the path");
        pl.Hout = (d->H - 1) * 2 - 2 * d->pad + d->kH + d->output_padding;
        pl.Wout = (d->W - 1) * 2 - 2 * d->pad + d->kW + d->output_padding;
        T2V_REQUIRE(pl.Hout > 0 && pl.Wout > 0, "transposed conv: empty output");
        T2V_REQUIRE(!need_stats || (pl.Hout % 2 == 0 && pl.Wout % 2 == 0),
                    "transposed conv with fused norm statistics needs an even output (got %dx%d)", pl.Hout, pl.Wout);
        Hm = (pl.Hout + 1) / 2;   // phase grid; odd outputs: the odd phase's last row/col is masked at the store
        Wm = (pl.Wout + 1) / 2;
        k.stride = 1;
        k.ostride = 2;
        k.pad_mode = T2V_PAD_ZERO;
        k.nphases = 4;
        k.KW = 3;
        k.pad = 0;
        int tap0 = 0;
        long woff = 0;
        for (int p = 0; p < 4; ++p) {
            int nt, kh[4], kw[4], dy[4], dx[4], a, b;
            convT_phase_taps_host(p, d->kH, d->pad, &nt, kh, kw, dy, dx, &a, &b);
            ConvPhase& ph = k.ph[p];
            ph.ntaps = nt;
            ph.tap0 = tap0;
            ph.Kp = round_up(nt * x_cs, kBK);
            ph.nk = ph.Kp / kBK;
            ph.w_off = woff;
            ph.oy0 = a;
            ph.ox0 = b;
            for (int t = 0; t < nt; ++t) {
                k.tdy[tap0 + t] = dy[t];
                k.tdx[tap0 + t] = dx[t];
            }
            tap0 += nt;
            woff += (long)pl.Cout_p * ph.Kp;
        }
        pl.wfloats = (size_t)woff;
    }
    k.group_mtiles = 1 << 30;
    k.group_w_stride = 0;
    k.Wm = Wm;
    k.M = Hm * Wm;
    k.Wout = pl.Wout;
    k.Hout = pl.Hout;
    k.mtiles = (k.M + pl.BM - 1) / pl.BM;
    if (pl.tile == kTileL) {
        // tile quantisation: with 128x128 tiles a grid that fills the last wave of 256 CUs poorly
        // (e.g. 160 or 344 tiles at the real fadg0 geometries 512x320 / 512x680) idles a third of
        // the chip; 64x64 tiles (several co-resident blocks per CU) even that out
        const long nb = (long)k.mtiles * k.ntiles * k.nphases;
        const double fill = (double)nb / (double)(((nb + 255) / 256) * 256);
        // transposed convs with few blocks: the four phases have 1/2/2/4 taps, and one 128x128 block per CU cannot
        // balance such unequal blocks (1024->512 up-sampling: 0.317 -> 0.286 ms with 64x64 tiles)
        // ... or when four times as many 64x64 blocks fill the chip's rounds clearly better (864 tiles of the 64x85
        // Winograd GEMM: 0.84 -> 0.96, 0.259 -> 0.244 ms)
        const long nbq = 4 * nb;
        const double fillq = (double)nbq / (double)(((nbq + 255) / 256) * 256);
        // (rule against forced tiles, per layer, at 512x320 / 512x680 / 512x512: within 1.3 / 2.8 / 1.0 % of the best forced
        // choice -- profiles/r04_ab_s2_tiles.txt; the forcing switch is gone)
        const int force = options().conv_tile;
        if (force == 2 || (force != 1 && ((fill < 0.8 && nb < 1024) || (k.nphases > 1 && nb <= 512) || (fillq - fill >= 0.1 && nb < 2048)))) {
            pl.tile = kTileQ;
            conv_tile_dims(pl.tile, &pl.BM, &pl.BN);
            k.ntiles = (d->Cout + pl.BN - 1) / pl.BN;
            k.mtiles = (k.M + pl.BM - 1) / pl.BM;
        }
    }
    if (need_stats && !d->transposed && d->kH == 7 && d->kW == 7 && d->stride == 1 && d->pad == 3 &&
        d->pad_mode == T2V_PAD_REFLECT && d->act == T2V_ACT_NONE && conv_stem7x7_supported(d->H, d->W, x_cs, d->Cout)) {
        // the 7x7 stems: halo-in-LDS MFMA kernel (conv_stem.hip); statistics per 16x16 pixel tile
        pl.tile = kTileStem;
        pl.BM = 256;
        pl.BN = d->Cout;
        k.mtiles = ((d->H + 15) / 16) * ((d->W + 15) / 16);
        k.ntiles = 1;
    }
    pl.nparts = k.nphases * k.mtiles;
    T2V_REQUIRE((long)k.mtiles * k.ntiles * k.nphases < (1L << 31), "conv: grid too large");
    // buffer addressing: 32-bit byte offsets below the out-of-range marker 0x7fff0000
    T2V_REQUIRE((long)d->H * d->W * x_cs * 4 < 0x7fff0000L, "conv: input tensor too large for 32-bit buffer offsets");
    for (int ph = 0; ph < k.nphases; ++ph)
        T2V_REQUIRE((long)pl.Cout_p * k.ph[ph].Kp * 4 < 0x7fff0000L, "conv: weight too large for 32-bit buffer offsets");
    return T2V_OK;
}

bool winograd_supported(const t2v_conv_desc* d, int x_cs, int algo) {
    if (!d || !is_winograd(algo)) return false;
    if (d->transposed || d->kH != 3 || d->kW != 3 || d->stride != 1) return false;
    // ReflectionPad2d(1) (forward ResnetBlock conv) or zero padding 0..2 (pad 2: that conv's data gradient)
    if (d->pad_mode == T2V_PAD_REFLECT ? d->pad != 1 : (d->pad < 0 || d->pad > 2)) return false;
    if (wino_out_h(d) < 1 || wino_out_w(d) < 1) return false;
    // any H, W >= 2 (reflection needs 2): ragged tiles are masked, the tile count is padded to 128
    if (d->Cin % 32 != 0 || x_cs != d->Cin || d->Cout % 4 != 0 || d->H < 2 || d->W < 2) return false;
    // the F(4x4) output transform can apply a (Leaky)ReLU (convs without a norm: the VGG19 loss network)
    return d->act == T2V_ACT_NONE || (algo == T2V_ALGO_WINOGRAD_F4 && d->act == T2V_ACT_LRELU);
}

// F(4x4) unless the zero tiles that pad its coarser grid to 128 make F(2x2) the smaller GEMM (tiny maps); direct
// when neither beats 9 rows per output pixel
int best_conv_algo(const t2v_conv_desc* d, int x_cs, int cap) {
    const bool f4 = cap == 0 && winograd_supported(d, x_cs, T2V_ALGO_WINOGRAD_F4);
    const bool f2 = (cap == 0 || cap == 2) && winograd_supported(d, x_cs, T2V_ALGO_WINOGRAD);
    const long direct_rows = 9L * wino_out_h(d) * wino_out_w(d);
    if (f4 && (!f2 || wino_gemm_rows(d, T2V_ALGO_WINOGRAD_F4) <= wino_gemm_rows(d, T2V_ALGO_WINOGRAD)) &&
        wino_gemm_rows(d, T2V_ALGO_WINOGRAD_F4) < direct_rows)
        return T2V_ALGO_WINOGRAD_F4;
    if (f2 && wino_gemm_rows(d, T2V_ALGO_WINOGRAD) < direct_rows) return T2V_ALGO_WINOGRAD;
    return T2V_ALGO_DIRECT;
}

// the batched GEMM of a Winograd conv as a plan of the implicit-GEMM kernel: a 1x1 conv over a 16|36 x T image
int build_winograd_gemm_plan(const t2v_conv_desc* d, ConvPlan* pl, int nimg) {
    const int T = wino_rows_batch(d, d->algo, nimg);   // GEMM rows per transform position: all images' tiles, packed
    t2v_conv_desc g;
    memset(&g, 0, sizeof(g));
    g.H = wino_pos(d->algo); g.W = T; g.Cin = d->Cin; g.Cout = d->Cout; g.kH = g.kW = 1; g.stride = 1; g.pad = 0;
    g.pad_mode = T2V_PAD_ZERO; g.act = T2V_ACT_NONE; g.act_scale = 1.f;
    T2V_TRY(build_conv_plan(&g, d->Cin, /*need_stats: 128- or 64-row tiles only*/ true, pl));
    // A batch (nimg >= 2: T = 512, 1024, ... rows per position) keeps the 128x128 tiles build_conv_plan's fill rule
    // leaves it with: alone the 64x64 tiles are faster at T = 512 (315 vs 335 us per batch-2 launch), inside two-stream
    // frames they lose (11.80 vs 11.41 ms per frame) -- the big tiles leave wave slots to the other stream's kernels.
    if (T % pl->BM != 0 && pl->tile == kTileL) {   // tile count padded to 64 only: the 64x64 tile config
        pl->tile = kTileQ;
        conv_tile_dims(pl->tile, &pl->BM, &pl->BN);
        pl->kp.ntiles = (g.Cout + pl->BN - 1) / pl->BN;
        pl->kp.mtiles = (pl->kp.M + pl->BM - 1) / pl->BM;
        pl->nparts = pl->kp.nphases * pl->kp.mtiles;
    }
    pl->kp.group_mtiles = T / pl->BM;               // T % 128 == 0 and BM in {128, 64, 256?}: checked below
    T2V_REQUIRE(T % pl->BM == 0, "winograd gemm: tile rows %d do not divide %d", pl->BM, T);
    pl->kp.group_w_stride = (long)pl->Cout_p * d->Cin;
    return T2V_OK;
}

// `batch` images in one go (image b at x + b * x_stride, y + b * y_stride, statistics at stats + b * stats_stride, floats):
// the implicit-GEMM kernel takes them as blockIdx.y of ONE launch -- the discriminators' layers of the train step are 67-552
// blocks per image, a fraction of the 1024+ block slots of the chip; the dedicated stem / head / one-channel kernels are
// launched image by image.
int run_conv_batch(t2v_ctx* ctx, hipStream_t s, const ConvPlan& pl, int batch, const float* x, long x_stride, const float* w,
                   const float* bias, float* y, int y_cs, long y_stride, float* stats, long stats_stride) {
    T2V_REQUIRE(ctx && x && w && y, "conv: null pointer");
    T2V_REQUIRE(batch >= 1 && batch <= 65535, "conv: batch %d", batch);
    T2V_REQUIRE(y_cs >= pl.kp.Cout && y_cs <= pl.Cout_p, "conv: output channel storage %d out of range [%d,%d]", y_cs,
                pl.kp.Cout, pl.Cout_p);
    T2V_REQUIRE((long)pl.Hout * pl.Wout * y_cs * 4 < 0x7fff0000L, "conv: output tensor too large for 32-bit buffer offsets");
    {
        // the generator heads: dedicated halo-tile kernel (conv_head.hip)
        const ConvKParams& q = pl.kp;
        if (!stats && q.nphases == 1 && q.ph[0].ntaps == 49 && q.KW == 7 && q.pad == 3 && q.stride == 1 &&
            q.pad_mode == T2V_PAD_REFLECT && q.Cout <= 3 && q.Cin_s % 16 == 0 && q.Hin >= 4 && q.Win >= 4) {
            for (int b = 0; b < batch; ++b) {
                HeadParams h;
                h.x = x + b * x_stride; h.w = w; h.bias = bias; h.y = y + b * y_stride;
                h.H = q.Hin; h.W = q.Win; h.Cin_s = q.Cin_s; h.Kp = q.ph[0].Kp; h.Cout = q.Cout; h.Cout_s = y_cs;
                h.act = q.act; h.act_scale = q.act_scale;
                T2V_TRY(launch_conv_head7x7(s, h));
            }
            return T2V_OK;
        }
    }
    {
        // one output channel, long K (the discriminators' last layer): a wave per output pixel (conv_head.hip)
        const ConvKParams& q = pl.kp;
        if (!stats && q.nphases == 1 && q.Cout == 1 && q.pad_mode == T2V_PAD_ZERO && q.ostride == 1 &&
            q.ph[0].ntaps == q.KW * q.KW && conv_cout1_supported(q.KW, q.Cin_s) &&
            (q.act == T2V_ACT_NONE || q.act == T2V_ACT_LRELU)) {
            for (int b = 0; b < batch; ++b) {
                Cout1Params c;
                c.x = x + b * x_stride; c.w = w; c.bias = bias; c.y = y + b * y_stride;
                c.H = q.Hin; c.W = q.Win; c.Cin_s = q.Cin_s; c.ksize = q.KW; c.stride = q.stride; c.pad = q.pad;
                c.Hout = pl.Hout; c.Wout = pl.Wout; c.Cout_s = y_cs; c.act = q.act; c.act_scale = q.act_scale;
                T2V_TRY(launch_conv_cout1(s, c));
            }
            return T2V_OK;
        }
    }
    if (pl.tile == kTileStem && stats) {
        for (int b = 0; b < batch; ++b) {
            StemParams sp;
            sp.x = x + b * x_stride; sp.w = w; sp.bias = bias; sp.y = y + b * y_stride; sp.stats = stats + b * stats_stride;
            sp.H = pl.kp.Hin; sp.W = pl.kp.Win; sp.Cin_s = pl.kp.Cin_s; sp.Kp = pl.kp.ph[0].Kp; sp.Cout = pl.kp.Cout; sp.Cout_s = y_cs;
            sp.Cin = pl.Cin;
            T2V_TRY(launch_conv_stem7x7(s, sp));
        }
        return T2V_OK;
    }
    T2V_REQUIRE(pl.tile != kTileStem, "conv: the stem plan needs a statistics buffer");
    ConvKParams k = pl.kp;
    k.x = x;
    k.w = w;
    k.bias = bias;
    k.y = y;
    k.Cout_s = y_cs;
    k.stats = stats;
    k.batch = batch;
    k.x_img_stride = batch > 1 ? x_stride : 0;
    k.y_img_stride = batch > 1 ? y_stride : 0;
    k.stats_img_stride = (batch > 1 && stats) ? stats_stride : 0;
    return launch_conv_igemm(s, k, pl.tile);
}

int run_conv(t2v_ctx* ctx, hipStream_t s, const ConvPlan& pl, const float* x, const float* w, const float* bias,
             float* y, int y_cs, float* stats) {
    return run_conv_batch(ctx, s, pl, 1, x, 0, w, bias, y, y_cs, 0, stats, 0);
}

// all of a Winograd conv (shared with the generator orchestrator): stages bit 1 = input transform,
// 2 = batched GEMM, 4 = output transform
int winograd_forward(t2v_ctx* ctx, hipStream_t s, const t2v_conv_desc* d, const float* x, const float* w_packed,
                     const float* bias, float* y, float* stats_partial, float* workspace, int stages,
                     const WinoBatch* batch) {
    const WinoBatch one;
    const WinoBatch& wb = batch ? *batch : one;
    const int nimg = wb.nimg;
    const bool f4 = d->algo == T2V_ALGO_WINOGRAD_F4;
    if (f4) T2V_TRY(check_async_errors());
    T2V_REQUIRE(nimg >= 1 && (f4 || nimg == 1), "winograd: batches are F(4x4,3x3) only");
    const size_t T = (size_t)wino_rows_batch(d, d->algo, nimg);
    const bool keep = wb.keep_v != nullptr;
    T2V_REQUIRE(!keep || (f4 && nimg == 1 && wb.keep_slot >= 0 && wb.keep_slot < wb.keep_total),
                "winograd: V is kept for the weight gradient of single F(4x4,3x3) images only");
    // (kept: slot keep_slot of [36][keep_total * Tp][Cin]; the positions are keep_total * Tp rows apart)
    float* V = keep ? wb.keep_v + (size_t)wb.keep_slot * T * d->Cin : workspace;
    const long v_group_stride = (long)(keep ? wb.keep_total : 1) * (long)T * d->Cin;
    float* Mm = workspace + wino_pos(d->algo) * T * d->Cin;
    if (stages & 1) {
        const int reflect = d->pad_mode == T2V_PAD_REFLECT;
        if (keep) T2V_TRY(launch_winograd4_input(s, x, wb.keep_v, d->H, d->W, d->Cin, d->pad, reflect, wb.keep_total, wb.keep_slot));
        else
        T2V_TRY(f4 ? launch_winograd4_input(s, x, V, d->H, d->W, d->Cin, d->pad, reflect, nimg, 0, nimg, wb.img_stride_x)
                   : launch_winograd_input(s, x, V, d->H, d->W, d->Cin, d->pad, reflect));
    }
    if (stages & 2) {
        const int rows = nimg * wino_tiles_real(d, d->algo);      // real tile rows per position (the rest of T is padding)
        const bool tall_ragged = f4 && wino_gemm_skt_ok(36, rows, (int)T, d->Cin, d->Cout, d->Cout);
        if (tall_ragged || (f4 && wino_gemm_skr_ok(36, rows, (int)T, d->Cin, d->Cout, d->Cout))) {
            SkGemm g;
            g.a = V; g.b = w_packed; g.c = Mm; g.scratch = workspace + winograd_vm_floats(d, nimg);
            g.err = async_error_word();
            g.a_group_stride = v_group_stride;
            g.groups = 36; g.T = (int)T; g.K = d->Cin; g.N = d->Cout; g.c_cs = d->Cout;
            T2V_TRY(tall_ragged ? launch_wino_gemm_skt(s, g, rows) : launch_wino_gemm_skr(s, g, rows));
        } else if (f4 && wino_gemm_sk_ok(36, (int)T, d->Cin, d->Cout, d->Cout, rows)) {
            SkGemm g;
            g.a = V; g.b = w_packed; g.c = Mm; g.scratch = workspace + winograd_vm_floats(d, nimg);
            g.err = async_error_word();
            g.rows = rows;
            g.a_group_stride = v_group_stride;
            g.groups = 36; g.T = (int)T; g.K = d->Cin; g.N = d->Cout; g.c_cs = d->Cout;
            T2V_TRY(launch_wino_gemm_sk(s, g));
        } else {
            ConvPlan pl;
            T2V_TRY(build_winograd_gemm_plan(d, &pl, nimg));
            if (keep) {   // the [36][T] "image" of the 1x1 plan is a window of the batch-wide one: row pitch keep_total * T
                T2V_REQUIRE(v_group_stride * 36 * 4 < 0x7fff0000L, "winograd: kept V too large for 32-bit buffer offsets");
                pl.kp.Win = wb.keep_total * (int)T;
            }
            T2V_TRY(run_conv(ctx, s, pl, V, w_packed, nullptr, Mm, d->Cout, nullptr));
        }
    }
    if (stages & 4) {
        T2V_REQUIRE(d->act == T2V_ACT_NONE || !stats_partial, "winograd: an activation and norm statistics do not combine");
        if (f4)
            T2V_TRY(launch_winograd4_output(s, Mm, bias, y, stats_partial, wino_out_h(d), wino_out_w(d), d->Cout,
                                            d->act == T2V_ACT_LRELU, d->act_scale, nimg));
        else
            T2V_TRY(launch_winograd_output(s, Mm, bias, y, stats_partial, wino_out_h(d), wino_out_w(d), d->Cout));
    }
    return T2V_OK;
}

bool polyphase_supported(const t2v_conv_desc* d, int x_cs) {
    if (!d || d->kH != 3 || d->kW != 3 || d->stride != 2 || d->pad != 1 || d->pad_mode != T2V_PAD_ZERO) return false;
    if (d->act != T2V_ACT_NONE || d->Cin % 32 != 0 || x_cs != d->Cin || d->Cout % 128 != 0) return false;
    // any map: the tile grid is ragged at the bottom / right edge (outputs masked, inputs past the map read as the zeros they
    // are) and padded with empty tiles to the GEMM's row granule; a down conv's input must be even-sized
    if (d->transposed) return d->output_padding == 1 && d->H >= 2 && d->W >= 2;
    return d->H >= 4 && d->W >= 4 && d->H % 2 == 0 && d->W % 2 == 0;
}

// where it measured faster than the implicit-GEMM kernel (scripts/poly_check.py, MI355X): both channel counts >= 256 -- the
// transforms move 81/64 + 81/16 of a layer's input + output, against a GEMM of 0.5625x the direct FLOPs -- and enough tiles to
// fill the fixed grid.  512->1024 @128x128 301 -> 229 us, 1024->512 (transposed) @64x64 325 -> 214, 256->512 @256x256 272 ->
// 245, 512->256 (transposed) @128x128 303 -> 248; 128->256 @512x512 284 -> 362 (no)
bool polyphase_pays(const t2v_conv_desc* d, int x_cs) {
    return polyphase_supported(d, x_cs) && d->Cin >= 256 && d->Cout >= 256 && poly_tiles_real(d) >= 128;
}

// stages bit 1 = input transform, 2 = the 81 batched GEMMs, 4 = output transform (bias, statistics partials)
int polyphase_forward(t2v_ctx* ctx, hipStream_t s, const t2v_conv_desc* d, const float* x, const float* w_packed,
                      const float* bias, float* y, float* stats_partial, float* workspace, int stages,
                      const PolyLazyNorm* lazy) {
    T2V_TRY(check_async_errors());
    const int up = d->transposed ? 1 : 0;
    const int T = poly_tiles_padded(d), rows = poly_tiles_real(d);
    float* V = workspace;
    float* Mm = workspace + (size_t)81 * T * d->Cin;
    if (stages & 1)
        T2V_TRY(lazy ? launch_polyphase_input(s, x, V, d->H, d->W, d->Cin, up, T, lazy->mean_rstd, lazy->gamma, lazy->beta, lazy->relu)
                     : launch_polyphase_input(s, x, V, d->H, d->W, d->Cin, up, T));
    if (stages & 2) {
        SkGemm g;
        g.a = V; g.b = w_packed; g.c = Mm; g.scratch = workspace + (size_t)81 * T * ((size_t)d->Cin + d->Cout);
        g.err = async_error_word();
        g.a_group_stride = (long)T * d->Cin;
        g.groups = 81; g.T = T; g.K = d->Cin; g.N = d->Cout; g.c_cs = d->Cout;
        if (wino_gemm_skt_ok(81, rows, T, d->Cin, d->Cout, d->Cout)) {
            T2V_TRY(launch_wino_gemm_skt(s, g, rows));
        } else if (wino_gemm_skr_ok(81, rows, T, d->Cin, d->Cout, d->Cout)) {
            T2V_TRY(launch_wino_gemm_skr(s, g, rows));
        } else if (wino_gemm_sk_ok(81, T, d->Cin, d->Cout, d->Cout, rows)) {
            g.rows = rows;
            T2V_TRY(launch_wino_gemm_sk(s, g));
        } else {      // one block per tile: the batched GEMM as a 1x1 conv over an 81 x T image
            t2v_conv_desc gd;
            memset(&gd, 0, sizeof(gd));
            gd.H = 81; gd.W = T; gd.Cin = d->Cin; gd.Cout = d->Cout; gd.kH = gd.kW = 1; gd.stride = 1; gd.pad = 0;
            gd.pad_mode = T2V_PAD_ZERO; gd.act = T2V_ACT_NONE; gd.act_scale = 1.f;
            ConvPlan pl;
            T2V_TRY(build_conv_plan(&gd, d->Cin, true, &pl));
            if (T % pl.BM != 0 && pl.tile == kTileL) {
                pl.tile = kTileQ;
                conv_tile_dims(pl.tile, &pl.BM, &pl.BN);
                pl.kp.ntiles = (gd.Cout + pl.BN - 1) / pl.BN;
                pl.kp.mtiles = (pl.kp.M + pl.BM - 1) / pl.BM;
                pl.nparts = pl.kp.nphases * pl.kp.mtiles;
            }
            T2V_REQUIRE(T % pl.BM == 0, "polyphase gemm: tile rows %d do not divide %d", pl.BM, T);
            pl.kp.group_mtiles = T / pl.BM;
            pl.kp.group_w_stride = (long)pl.Cout_p * d->Cin;
            T2V_TRY(run_conv(ctx, s, pl, V, w_packed, nullptr, Mm, d->Cout, nullptr));
        }
    }
    if (stages & 4)
        T2V_TRY(launch_polyphase_output(s, Mm, bias, y, stats_partial, poly_out_h(d), poly_out_w(d), d->Cout, up, T));
    return T2V_OK;
}

}  // namespace t2v

using namespace t2v;

extern "C" {

int t2v_abi_version(void) { return T2V_ABI_VERSION; }
const char* t2v_last_error(void) { return g_err; }

int t2v_create(t2v_ctx** out, int device) {
    T2V_REQUIRE(out, "t2v_create: null out");
    int n = 0;
    T2V_HIP_CHECK(hipGetDeviceCount(&n));
    T2V_REQUIRE(device >= 0 && device < n, "t2v_create: device %d out of range (%d visible)", device, n);
    T2V_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    T2V_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    T2V_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
                "libt2v_hip is built for gfx950 (MI355X) only; device %d is %s", device, prop.gcnArchName);
    t2v_ctx* c = new t2v_ctx();
    c->device = device;
    c->side = nullptr;
    c->ev_fork = c->ev_join = nullptr;
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        t2v_destroy(c);
        set_error("t2v_create: cannot create the side stream / events");
        return T2V_ERR_HIP;
    }
    (void)options();
    {   // once per process: the sticky error word and the dispatch-order self-test of the fixed-grid kernels
        static std::once_flag once;
        int st = T2V_OK;
        std::call_once(once, [&]() {
            void* w = nullptr;
            if (hipHostMalloc(&w, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
                fixed_grid_disable("no pinned error word");
                return;
            }
            memset(w, 0, 64);
            g_err_word = static_cast<unsigned*>(w);
            st = fixed_grid_selftest();
        });
        if (st != T2V_OK) {
            t2v_destroy(c);
            return st;
        }
    }
    *out = c;
    return T2V_OK;
}

void t2v_reload_env(void) { options_reload(); }
int t2v_set_overlap_hint(int on) { return set_overlap_hint(on); }
int t2v_check_async_errors(void) { return check_async_errors(); }
int t2v_fixed_grid_enabled(void) { return fixed_grid_enabled() ? 1 : 0; }
void t2v_debug_async_error(int raise) {
    if (!g_err_word) return;
    if (raise) {
        *reinterpret_cast<volatile unsigned*>(g_err_word) = 1u;      // what a timed-out consumer wave stores
    } else {
        *reinterpret_cast<volatile unsigned*>(g_err_word) = 0u;
        g_fixed_grid_on.store(1);
    }
}

int t2v_destroy(t2v_ctx* ctx) {
    if (!ctx) return T2V_OK;
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    delete ctx;
    return T2V_OK;
}

int t2v_conv_out_dims(const t2v_conv_desc* d, int* Hout, int* Wout) {
    ConvPlan pl;
    T2V_TRY(build_conv_plan(d, round_up(d ? d->Cin : 0, 4), false, &pl));
    if (Hout) *Hout = pl.Hout;
    if (Wout) *Wout = pl.Wout;
    return T2V_OK;
}

size_t t2v_conv_packed_weight_floats(const t2v_conv_desc* d, int x_cs) {
    ConvPlan pl;
    if (build_conv_plan(d, x_cs, false, &pl) != T2V_OK) return 0;
    if (is_winograd(d->algo)) return winograd_supported(d, x_cs, d->algo) ? (size_t)wino_pos(d->algo) * pl.Cout_p * x_cs : 0;
    if (d->algo == T2V_ALGO_POLYPHASE) return polyphase_supported(d, x_cs) ? (size_t)81 * pl.Cout_p * x_cs : 0;
    return pl.wfloats;
}
int t2v_conv_polyphase_supported(const t2v_conv_desc* d, int x_cs) {
    return (polyphase_supported(d, x_cs) ? 1 : 0) | (polyphase_pays(d, x_cs) ? 2 : 0);
}

int t2v_conv_winograd_supported(const t2v_conv_desc* d, int x_cs) {
    return (winograd_supported(d, x_cs, T2V_ALGO_WINOGRAD) ? 1 : 0) | (winograd_supported(d, x_cs, T2V_ALGO_WINOGRAD_F4) ? 2 : 0);
}

int t2v_conv_best_algo(const t2v_conv_desc* d, int x_cs, int cap) { return d ? best_conv_algo(d, x_cs, cap) : T2V_ALGO_DIRECT; }

size_t t2v_conv_winograd_workspace_floats(const t2v_conv_desc* d, int x_cs) {
    if (d && d->algo == T2V_ALGO_POLYPHASE) return polyphase_supported(d, x_cs) ? polyphase_workspace_floats(d) : 0;
    if (!d || !winograd_supported(d, x_cs, d->algo)) return 0;
    return winograd_workspace_floats(d);
}

int t2v_conv2d_forward_winograd_stages(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, const float* x, int x_cs,
                                       const float* w_packed, const float* bias, float* y, int y_cs,
                                       float* stats_partial, float* workspace, int stages) {
    T2V_REQUIRE(ctx && d && x && w_packed && y && workspace, "winograd forward: null pointer");
    if (d->algo == T2V_ALGO_POLYPHASE) {
        T2V_REQUIRE(polyphase_supported(d, x_cs), "polyphase forward: shape not supported (t2v_conv_polyphase_supported)");
        T2V_REQUIRE(y_cs == d->Cout, "polyphase forward: output channel storage must equal Cout");
        return polyphase_forward(ctx, (hipStream_t)stream, d, x, w_packed, bias, y, stats_partial, workspace, stages);
    }
    T2V_REQUIRE(winograd_supported(d, x_cs, d->algo),
                "winograd forward: shape/algo not supported (t2v_conv_winograd_supported)");
    T2V_REQUIRE(y_cs == d->Cout, "winograd forward: output channel storage must equal Cout");
    return winograd_forward(ctx, (hipStream_t)stream, d, x, w_packed, bias, y, stats_partial, workspace, stages);
}

static bool wgrad_winograd_ok(const t2v_conv_desc* d, int x_cs, int dy_cs);
int t2v_conv2d_forward_winograd_keep_v(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, const float* x, int x_cs,
                                       const float* w_packed, const float* bias, float* y, int y_cs, float* stats_partial,
                                       float* workspace, float* wgrad_workspace, int batch, int slot) {
    T2V_REQUIRE(ctx && d && x && w_packed && y && workspace && wgrad_workspace, "winograd forward (V kept): null pointer");
    T2V_REQUIRE(d->algo == T2V_ALGO_WINOGRAD_F4 && winograd_supported(d, x_cs, d->algo) && x_cs == d->Cin &&
                    wgrad_winograd_ok(d, x_cs, d->Cout),
                "winograd forward (V kept): F(4x4,3x3) layers with a Winograd-domain weight gradient only "
                "(t2v_conv_backward_weight_winograd_supported)");
    T2V_REQUIRE(y_cs == d->Cout, "winograd forward: output channel storage must equal Cout");
    T2V_REQUIRE(batch >= 1 && slot >= 0 && slot < batch, "winograd forward (V kept): slot %d of %d", slot, batch);
    T2V_REQUIRE((long)36 * batch * wino_tiles_padded(d, d->algo) * x_cs * 4 < 0x7fff0000L,
                "winograd forward (V kept): batch-wide V too large for 32-bit buffer offsets");
    WinoBatch wb;
    wb.keep_v = wgrad_workspace;     // V is the first tensor of the weight gradient's workspace
    wb.keep_total = batch;
    wb.keep_slot = slot;
    return winograd_forward(ctx, (hipStream_t)stream, d, x, w_packed, bias, y, stats_partial, workspace, 7, &wb);
}

int t2v_conv2d_forward_winograd(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, const float* x, int x_cs,
                                const float* w_packed, const float* bias, float* y, int y_cs, float* stats_partial,
                                float* workspace) {
    return t2v_conv2d_forward_winograd_stages(ctx, stream, d, x, x_cs, w_packed, bias, y, y_cs, stats_partial, workspace, 7);
}

static int pack_weight(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int x_cs, const float* w_torch_dev,
                       float* packed_dev, int adjoint) {
    T2V_REQUIRE(ctx && w_torch_dev && packed_dev, "pack_weight: null pointer");
    ConvPlan pl;
    T2V_TRY(build_conv_plan(d, x_cs, false, &pl));
    hipStream_t s = (hipStream_t)stream;
    if (d->algo == T2V_ALGO_POLYPHASE) {
        T2V_REQUIRE(polyphase_supported(d, x_cs) && !adjoint, "pack_weight: polyphase form not supported for this shape");
        return launch_polyphase_weight(s, w_torch_dev, packed_dev, d->Cout, d->Cin, pl.Cout_p, x_cs, d->transposed ? 1 : 0);
    }
    if (is_winograd(d->algo)) {
        T2V_REQUIRE(winograd_supported(d, x_cs, d->algo), "pack_weight: Winograd not supported for this shape");
        return (d->algo == T2V_ALGO_WINOGRAD_F4 ? launch_winograd4_weight : launch_winograd_weight)(
            s, w_torch_dev, packed_dev, d->Cout, d->Cin, pl.Cout_p, x_cs, adjoint);
    }
    if (!d->transposed)
        return launch_pack_conv_weight(s, w_torch_dev, packed_dev, d->Cout, d->Cin, d->kH, d->kW, x_cs, pl.kp.ph[0].Kp,
                                       pl.Cout_p, adjoint);
    T2V_REQUIRE(!adjoint, "pack_weight_adjoint: stride-1 convolutions only (a transposed conv's data gradient reuses its weight)");
    return launch_pack_convT_weight(s, w_torch_dev, packed_dev, d->Cin, d->Cout, x_cs, pl.Cout_p, d->kH, d->pad);
}
int t2v_conv_pack_weight(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int x_cs, const float* w_torch_dev,
                         float* packed_dev) {
    return pack_weight(ctx, stream, d, x_cs, w_torch_dev, packed_dev, 0);
}
int t2v_conv_pack_weight_adjoint(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int x_cs, const float* w_forward_dev,
                                 float* packed_dev) {
    T2V_REQUIRE(d && !d->transposed && d->stride == 1, "pack_weight_adjoint: `d` must be a stride-1 (data-gradient) conv");
    return pack_weight(ctx, stream, d, x_cs, w_forward_dev, packed_dev, 1);
}

size_t t2v_conv_stats_floats(const t2v_conv_desc* d) {
    if (d && is_winograd(d->algo))   // one partial per 128 output-pixel slots of the padded tile grid
        return (size_t)(wino_tiles_padded(d, d->algo) * wino_m(d->algo) * wino_m(d->algo) / 128) * d->Cout * 2;
    if (d && d->algo == T2V_ALGO_POLYPHASE) return (size_t)(poly_tiles_padded(d) * poly_m(d) * poly_m(d) / 128) * d->Cout * 2;
    ConvPlan pl;
    if (!d || build_conv_plan(d, round_up(d->Cin, 4), true, &pl) != T2V_OK) return 0;
    return (size_t)pl.nparts * d->Cout * 2;
}

int t2v_conv2d_forward(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, const float* x, int x_cs,
                       const float* w_packed, const float* bias, float* y, int y_cs, float* stats_partial) {
    T2V_REQUIRE(d && d->algo == T2V_ALGO_DIRECT, "conv2d_forward: use t2v_conv2d_forward_winograd for algo=WINOGRAD");
    ConvPlan pl;
    T2V_TRY(build_conv_plan(d, x_cs, stats_partial != nullptr, &pl));
    return run_conv(ctx, (hipStream_t)stream, pl, x, w_packed, bias, y, y_cs, stats_partial);
}

int t2v_conv2d_forward_batch(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int batch, const float* x, int x_cs,
                             const float* w_packed, const float* bias, float* y, int y_cs, float* stats_partial) {
    T2V_REQUIRE(d && d->algo == T2V_ALGO_DIRECT, "conv2d_forward_batch: direct convolutions only");
    T2V_REQUIRE(batch >= 1, "conv2d_forward_batch: batch %d", batch);
    ConvPlan pl;
    T2V_TRY(build_conv_plan(d, x_cs, stats_partial != nullptr, &pl));
    return run_conv_batch(ctx, (hipStream_t)stream, pl, batch, x, (long)d->H * d->W * x_cs, w_packed, bias, y, y_cs,
                          (long)pl.Hout * pl.Wout * y_cs, stats_partial, (long)t2v_conv_stats_floats(d));
}

int t2v_instance_norm_finalize(t2v_ctx* ctx, void* stream, const t2v_conv_desc* producer, const float* stats_partial,
                               float eps, float* mean_rstd) {
    T2V_REQUIRE(ctx && stats_partial && mean_rstd, "inorm_finalize: null pointer");
    if (producer && is_winograd(producer->algo)) {   // the output transform emits one partial per 128 pixels
        return launch_inorm_finalize_winograd((hipStream_t)stream, stats_partial, wino_m(producer->algo),
                                              wino_out_h(producer), wino_out_w(producer), producer->Cout, eps, mean_rstd, 1);
    }
    if (producer && producer->algo == T2V_ALGO_POLYPHASE)
        return launch_inorm_finalize_winograd((hipStream_t)stream, stats_partial, poly_m(producer), poly_out_h(producer),
                                              poly_out_w(producer), producer->Cout, eps, mean_rstd, 1);
    ConvPlan pl;
    T2V_TRY(build_conv_plan(producer, round_up(producer ? producer->Cin : 0, 4), true, &pl));
    if (pl.tile == kTileStem)
        return launch_inorm_finalize_tiles((hipStream_t)stream, stats_partial, 16, producer->H, producer->W, producer->Cout, eps,
                                           mean_rstd, 1);
    return launch_inorm_finalize((hipStream_t)stream, stats_partial, pl.nparts, pl.kp.mtiles, pl.BM, pl.kp.M,
                                 producer->Cout, eps, mean_rstd);
}

static int batch_norm_finalize(hipStream_t s, const t2v_conv_desc* producer, int batch, const float* stats_partial, float eps,
                               float* mean_rstd, const RunningUpdate* ru) {
    if (producer && is_winograd(producer->algo))
        return launch_inorm_finalize_winograd(s, stats_partial, wino_m(producer->algo), wino_out_h(producer),
                                              wino_out_w(producer), producer->Cout, eps, mean_rstd, batch, nullptr, ru);
    if (producer && producer->algo == T2V_ALGO_POLYPHASE)
        return launch_inorm_finalize_winograd(s, stats_partial, poly_m(producer), poly_out_h(producer), poly_out_w(producer),
                                              producer->Cout, eps, mean_rstd, batch, nullptr, ru);
    ConvPlan pl;
    T2V_TRY(build_conv_plan(producer, round_up(producer ? producer->Cin : 0, 4), true, &pl));
    if (pl.tile == kTileStem)
        return launch_inorm_finalize_tiles(s, stats_partial, 16, producer->H, producer->W, producer->Cout, eps, mean_rstd, batch,
                                           nullptr, ru);
    // the per-image partial blocks are contiguous: a batch is just `batch` times more partial rows
    return launch_inorm_finalize(s, stats_partial, batch * pl.nparts, pl.kp.mtiles, pl.BM, pl.kp.M, producer->Cout, eps,
                                 mean_rstd, nullptr, ru);
}
int t2v_batch_norm_finalize(t2v_ctx* ctx, void* stream, const t2v_conv_desc* producer, int batch,
                            const float* stats_partial, float eps, float* mean_rstd) {
    T2V_REQUIRE(ctx && stats_partial && mean_rstd && batch >= 1, "batch_norm_finalize: bad arguments");
    return batch_norm_finalize((hipStream_t)stream, producer, batch, stats_partial, eps, mean_rstd, nullptr);
}
int t2v_batch_norm_finalize_running(t2v_ctx* ctx, void* stream, const t2v_conv_desc* producer, int batch,
                                    const float* stats_partial, float eps, float* mean_rstd, float* running_mean,
                                    float* running_var, float momentum, int times) {
    T2V_REQUIRE(ctx && producer && stats_partial && mean_rstd && running_mean && running_var && batch >= 1 && times >= 1,
                "batch_norm_finalize_running: bad arguments");
    int ho = 0, wo = 0;
    T2V_TRY(t2v_conv_out_dims(producer, &ho, &wo));
    const long n = (long)batch * ho * wo;
    T2V_REQUIRE(n >= 2, "batch_norm_finalize_running: the unbiased variance needs at least two values per channel");
    const RunningUpdate ru{running_mean, running_var, (float)n, momentum, times};
    return batch_norm_finalize((hipStream_t)stream, producer, batch, stats_partial, eps, mean_rstd, &ru);
}

// narrow-input regular convs (7x7 stems: 12 / 8 channels; the discriminators' 4x4 first layers: 8): fold the taps
// into the 128-wide channel side of the block tile
static bool wgrad_fold(const t2v_conv_desc* d, int x_cs) {
    return !d->transposed && x_cs < 64 && d->kH * d->kW > 1;
}

// few output channels (the 7x7 heads: 3 -> dy_cs 4) on a wide input: fold the taps onto the dY side
static bool wgrad_fold_n(const t2v_conv_desc* d, int x_cs, int dy_cs) {
    return !d->transposed && d->stride == 1 && dy_cs <= 16 && x_cs >= 64 && d->kH * d->kW > 1 &&
           d->kH * d->kW <= kMaxTaps;
}
static size_t wgrad_padded_floats(const t2v_conv_desc* d, int x_cs, int batch) {
    return (size_t)batch * (d->H + 2 * d->pad) * (d->W + 2 * d->pad) * x_cs;
}

static int wgrad_splits(const t2v_conv_desc* d, int x_cs, int batch, const ConvPlan& pl) {
    // few (tap, channel-tile) blocks but a long pixel reduction (high-resolution, narrow layers): cut the
    // reduction into ranges; partial gradients are summed in a fixed order afterwards
    int ntaps = 0;
    for (int ph = 0; ph < pl.kp.nphases; ++ph) ntaps += pl.kp.ph[ph].ntaps;
    const long blocks = wgrad_fold(d, x_cs) ? (long)((d->Cout + 127) / 128) * ((ntaps * x_cs + 127) / 128)
                                            : (long)ntaps * ((d->Cout + 127) / 128) * ((x_cs + 127) / 128);
    const long nk = ((long)batch * pl.kp.M + 31) / 32;
    // a launch costs rounds x (stages per block + ~3 stages of prologue / epilogue) stage times, a round being
    // one set of resident blocks: a block count just above a whole number of rounds idles most of the chip for
    // the last one.  Fewest stage times wins, ties to the smaller split (less partial-gradient traffic).
    long smax = nk / 8;              // at least 8 stages per block
    if (smax > 256) smax = 256;
    if (smax < 1) smax = 1;
    long s = 1, best = -1;
    const long slots = 2 * 256;      // two blocks (2 x 64 KiB of LDS, 104 VGPRs) are resident per CU
    for (long c = 1; c <= smax; ++c) {
        const long rounds = (blocks * c + slots - 1) / slots;
        // + zeroing, writing and re-reading c partial gradients at ~15 MB per stage time
        const long cost = rounds * ((nk + c - 1) / c + 3) + (c > 1 ? c * (long)pl.wfloats * 12 / 15000000 : 0);
        if (best < 0 || cost < best) { best = cost; s = c; }
        if (blocks * c >= 4096) break;
    }
    return s < 1 ? 1 : (int)s;
}

// arrival counters of the in-kernel combine: one int per (tap, n tile, c tile), rounded up to 256 floats
static size_t wgrad_ticket_floats(const t2v_conv_desc* d, int x_cs, const ConvPlan& pl) {
    int ntaps = 0;
    for (int ph = 0; ph < pl.kp.nphases; ++ph) ntaps += pl.kp.ph[ph].ntaps;
    const size_t tiles = wgrad_fold(d, x_cs) ? (size_t)((d->Cout + 127) / 128) * ((ntaps * x_cs + 127) / 128)
                                            : (size_t)ntaps * ((d->Cout + 127) / 128) * ((x_cs + 127) / 128);
    return (tiles + 255) / 256 * 256;
}
static int wgrad_combine_max_splits() { return options().wgrad_combine_max; }
// T2V_WGRAD_COMBINE=0: split partials zero-filled, written and summed by a reduce launch (the round-1 form)
static bool wgrad_combine_on() { return options().wgrad_combine != 0; }

size_t t2v_conv_backward_weight_workspace_floats(const t2v_conv_desc* d, int x_cs, int batch) {
    ConvPlan pl;
    if (!d || build_conv_plan(d, x_cs, true, &pl) != T2V_OK) return 0;
    const int dy_cs = round_up(d->Cout, 4);
    if (wgrad_fold_n(d, x_cs, dy_cs)) {   // padded input copy + the partials of its own split rule
        const long nk = ((long)batch * (d->H + 2 * d->pad) * (d->W + 2 * d->pad) + 31) / 32;
        long s = nk / 16;
        if (s > 256) s = 256;
        if (s < 1) s = 1;
        return wgrad_padded_floats(d, x_cs, batch) + (size_t)s * d->Cout * pl.kp.ph[0].Kp;   // partials: the Cout real rows
    }
    const int s = wgrad_splits(d, x_cs, batch, pl);
    return s > 1 ? (size_t)s * pl.wfloats + wgrad_ticket_floats(d, x_cs, pl) : 0;
}

// x_stride / dy_stride: floats between the images (0: the batch's own pitch)
static int backward_weight_impl(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int batch, const float* x,
                                int x_cs, long x_stride, const float* dy, int dy_cs, long dy_stride, float* dw_packed,
                                int accumulate, float* workspace) {
    T2V_REQUIRE(ctx && x && dy && dw_packed && batch >= 1, "backward_weight: bad arguments");
    ConvPlan pl;
    T2V_TRY(build_conv_plan(d, x_cs, true, &pl));   // 128-row weight granule, plain 128x128 bookkeeping
    T2V_REQUIRE(dy_cs >= d->Cout && dy_cs % 4 == 0, "backward_weight: dy channel storage %d", dy_cs);
    const ConvKParams& k = pl.kp;
    WgradParams w;
    memset(&w, 0, sizeof(w));
    w.x = x; w.dy = dy; w.dw = dw_packed;
    w.batch = batch; w.Hin = d->H; w.Win = d->W; w.Cin_s = x_cs;
    w.Wm = k.Wm; w.M = k.M;
    w.Hout = pl.Hout; w.Wout = pl.Wout; w.Cout = d->Cout; w.Cout_s = dy_cs;
    w.stride = k.stride; w.ostride = k.ostride;
    w.reflect = k.pad_mode == T2V_PAD_REFLECT; w.accumulate = accumulate;
    w.ntiles = (d->Cout + 127) / 128; w.ctiles = (x_cs + 127) / 128;
    int nt = 0;
    for (int ph = 0; ph < k.nphases; ++ph)
        for (int t = 0; t < k.ph[ph].ntaps; ++t, ++nt) {
            w.tdy[nt] = k.tdy[k.ph[ph].tap0 + t]; w.tdx[nt] = k.tdx[k.ph[ph].tap0 + t];
            w.toy[nt] = k.ph[ph].oy0; w.tox[nt] = k.ph[ph].ox0;
            w.tap_woff[nt] = k.ph[ph].w_off; w.tap_Kp[nt] = k.ph[ph].Kp; w.tap_kidx[nt] = t;
        }
    w.ntaps = nt;
    w.x_img_stride = (long)d->H * d->W * x_cs;
    w.dy_img_stride = (long)pl.Hout * pl.Wout * dy_cs;
    if (x_stride || dy_stride) {
        T2V_REQUIRE(!(wgrad_fold_n(d, x_cs, dy_cs) && dy_cs == round_up(d->Cout, 4)) && !wgrad_fold(d, x_cs) && conv_wgrad_strided_ok(w),
                    "backward_weight_strided: this shape keeps its images contiguous (t2v_conv_backward_weight_strided_supported)");
        if (x_stride) w.x_img_stride = x_stride;
        if (dy_stride) w.dy_img_stride = dy_stride;
    }
    if (wgrad_fold_n(d, x_cs, dy_cs) && dy_cs == round_up(d->Cout, 4)) {
        // x -> padded copy (reflection / zeros resolved once), taps folded onto the dY side of the tile
        T2V_REQUIRE(workspace, "backward_weight: this shape needs a workspace of "
                               "t2v_conv_backward_weight_workspace_floats() floats");
        hipStream_t st = (hipStream_t)stream;
        const int Hp = d->H + 2 * d->pad, Wp = d->W + 2 * d->pad;
        float* xp = workspace;
        float* partial = workspace + wgrad_padded_floats(d, x_cs, batch);
        T2V_TRY(launch_pad_copy(st, x, xp, batch, d->H, d->W, x_cs, d->pad, k.pad_mode == T2V_PAD_REFLECT));
        T2V_REQUIRE((long)batch * Hp * Wp * x_cs * 4 < 0x7fff0000L, "backward_weight: padded input too large for 32-bit offsets");
        w.x = xp; w.Hin = Hp; w.Win = Wp; w.Wm = Wp; w.M = Hp * Wp;
        w.reflect = 0;
        w.fold = 2; w.fold_taps = nt; w.KW = d->kW; w.pad = 0;
        w.ntiles = (nt * dy_cs + 127) / 128;
        w.ntaps = 1;
        for (int t = 0; t < kMaxTaps; ++t) w.tdy[t] = w.tdx[t] = 0;
        const long nk = ((long)batch * Hp * Wp + 31) / 32;
        long sp = nk / 16;
        if (sp > 256) sp = 256;
        if (sp < 1) sp = 1;
        w.splits = (int)sp;
        // only the first Cout rows of the packed [Cout_p][Kp] matrix are ever written: partials hold just those
        const long part = (long)d->Cout * k.ph[0].Kp;
        w.dw_floats = part;
        w.dw = partial;
        // (taps folded onto the dY side: tile rows are (tap, channel) pairs -- this shape keeps the reduce launch.  No
        // zero-fill of the slabs: every real element of every partial is stored by exactly one block -- a block with an
        // empty pixel range stores zeros -- and the padding columns are summed and never read)
        w.accumulate = 0;
        T2V_TRY(launch_conv_wgrad(st, w));
        return launch_wgrad_reduce(st, partial, w.splits, part, dw_packed, accumulate);
    }
    if (wgrad_fold(d, x_cs)) {
        w.fold = 1; w.fold_taps = nt; w.KW = d->kW; w.pad = d->pad;
        w.ctiles = (nt * x_cs + 127) / 128;
        w.ntaps = 1;
    }
    T2V_REQUIRE((long)batch * d->H * d->W * x_cs * 4 < 0x7fff0000L && (long)batch * pl.Hout * pl.Wout * dy_cs * 4 < 0x7fff0000L,
                "backward_weight: tensors too large for 32-bit buffer offsets (split the batch)");
    w.splits = wgrad_splits(d, x_cs, batch, pl);
    w.dw_floats = (long)pl.wfloats;
    if (w.splits == 1) return launch_conv_wgrad((hipStream_t)stream, w);
    T2V_REQUIRE(workspace, "backward_weight: this shape needs a workspace of "
                           "t2v_conv_backward_weight_workspace_floats() floats");
    w.dw = workspace;
    if (wgrad_combine_on() && w.splits <= wgrad_combine_max_splits()) {
        // in-kernel combine: no zero-fill of the partial slabs, no reduce launch -- only the arrival counters are cleared.
        // For few partials only: the last arriver of a tile reads them one after the other (T2V_WGRAD_COMBINE_MAX).
        int* tickets = reinterpret_cast<int*>(workspace + (size_t)w.splits * pl.wfloats);
        T2V_HIP_CHECK(hipMemsetAsync(tickets, 0, (size_t)w.ntaps * w.ntiles * w.ctiles * sizeof(int), (hipStream_t)stream));
        w.tickets = tickets;
        w.dw_final = dw_packed;
        w.accumulate = accumulate;
        return launch_conv_wgrad((hipStream_t)stream, w);
    }
    // (no zero-fill of the partial slabs either: every real (n, tap, c) element of every slab is stored by exactly one
    // block, zeros where a split's pixel range is empty; the padding rows / columns of the packed layout are summed as they
    // are and never read -- tests/ run with the workspaces NaN-filled (scripts/run_poisoned.py))
    w.accumulate = 0;
    T2V_TRY(launch_conv_wgrad((hipStream_t)stream, w));
    return launch_wgrad_reduce((hipStream_t)stream, workspace, w.splits, (long)pl.wfloats, dw_packed, accumulate);
}

int t2v_conv2d_backward_weight(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int batch, const float* x,
                               int x_cs, const float* dy, int dy_cs, float* dw_packed, int accumulate,
                               float* workspace) {
    return backward_weight_impl(ctx, stream, d, batch, x, x_cs, 0, dy, dy_cs, 0, dw_packed, accumulate, workspace);
}
int t2v_conv_backward_weight_strided_supported(const t2v_conv_desc* d, int x_cs, int dy_cs) {
    ConvPlan pl;
    if (!d || build_conv_plan(d, x_cs, true, &pl) != T2V_OK) return 0;
    if ((wgrad_fold_n(d, x_cs, dy_cs) && dy_cs == round_up(d->Cout, 4)) || wgrad_fold(d, x_cs)) return 0;
    WgradParams w;
    memset(&w, 0, sizeof(w));
    w.reflect = pl.kp.pad_mode == T2V_PAD_REFLECT;
    w.Wm = pl.kp.Wm;
    w.M = pl.kp.M;
    return conv_wgrad_strided_ok(w) ? 1 : 0;
}
int t2v_conv2d_backward_weight_strided(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int batch, const float* x,
                                       int x_cs, long x_img_stride, const float* dy, int dy_cs, long dy_img_stride,
                                       float* dw_packed, int accumulate, float* workspace) {
    T2V_REQUIRE(x_img_stride != 0 && dy_img_stride != 0, "backward_weight_strided: zero image stride");
    return backward_weight_impl(ctx, stream, d, batch, x, x_cs, x_img_stride, dy, dy_cs, dy_img_stride, dw_packed, accumulate,
                                workspace);
}

// ---- weight gradient in the Winograd domain (F(4x4,3x3)) -------------------------------------------------
static bool wgrad_winograd_ok(const t2v_conv_desc* d, int x_cs, int dy_cs) {
    return winograd_supported(d, x_cs, T2V_ALGO_WINOGRAD_F4) && dy_cs == d->Cout && d->Cout % 4 == 0;
}
int t2v_conv_backward_weight_winograd_supported(const t2v_conv_desc* d, int x_cs, int dy_cs) {
    return d && wgrad_winograd_ok(d, x_cs, dy_cs) ? 1 : 0;
}
size_t t2v_conv_backward_weight_winograd_workspace_floats(const t2v_conv_desc* d, int x_cs, int batch) {
    if (!d || batch < 1 || !winograd_supported(d, x_cs, T2V_ALGO_WINOGRAD_F4)) return 0;
    const size_t Tp = (size_t)wino_tiles_padded(d, T2V_ALGO_WINOGRAD_F4);
    const size_t Cout_p = (size_t)round_up(d->Cout, 128), Kp = (size_t)round_up(x_cs, kBK);
    // V, M_dy, dU, then the hand-over area of the fixed-grid reduction (conv_wgrad.hip: wino_wgrad_sk_kernel)
    return 36 * batch * Tp * ((size_t)x_cs + d->Cout) + 36 * Cout_p * Kp + wino_gemm_sk_scratch_floats();
}
int t2v_conv_winograd_tile_rows(const t2v_conv_desc* d) {
    return d ? wino_tiles_padded(d, T2V_ALGO_WINOGRAD_F4) : 0;
}
int t2v_conv_winograd_gemm_form(const t2v_conv_desc* d, int nimg) {
    if (!d || d->algo != T2V_ALGO_WINOGRAD_F4 || nimg < 1) return -1;
    const int T = wino_rows_batch(d, d->algo, nimg), rows = nimg * wino_tiles_real(d, d->algo);
    if (wino_gemm_skt_ok(36, rows, T, d->Cin, d->Cout, d->Cout)) return T2V_GEMM_FIXED_GRID_RAGGED_TALL;
    if (wino_gemm_skr_ok(36, rows, T, d->Cin, d->Cout, d->Cout)) return T2V_GEMM_FIXED_GRID_RAGGED;
    if (wino_gemm_sk_ok(36, T, d->Cin, d->Cout, d->Cout, rows))
        switch (wino_gemm_sk_tall_rows(36, rows, T, d->Cout)) {
            case 160: return T2V_GEMM_FIXED_GRID_160x128;
            case 256: return T2V_GEMM_FIXED_GRID_256x128;
            default: return T % 128 == 0 ? T2V_GEMM_FIXED_GRID_128x128 : T2V_GEMM_FIXED_GRID_192x64;
        }
    ConvPlan pl;
    if (build_winograd_gemm_plan(d, &pl, nimg) != T2V_OK) return -1;
    return pl.tile == kTileL ? T2V_GEMM_TILE_PER_BLOCK_128x128 : T2V_GEMM_TILE_PER_BLOCK_64x64;
}
int t2v_conv2d_backward_weight_winograd_stages(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int batch, int b0,
                                               int nb, const float* x, int x_cs, const float* dy, int dy_cs,
                                               float* dw_torch, int accumulate, float* workspace, int stages) {
    T2V_REQUIRE(ctx && d && workspace && batch >= 1 && b0 >= 0 && nb >= 0 && b0 + nb <= batch,
                "backward_weight_winograd: bad arguments");
    T2V_REQUIRE(wgrad_winograd_ok(d, x_cs, dy_cs), "backward_weight_winograd: shape not supported "
                                                  "(t2v_conv_backward_weight_winograd_supported)");
    hipStream_t s = (hipStream_t)stream;
    const int Ho = wino_out_h(d), Wo = wino_out_w(d);
    const int Tp = wino_tiles_padded(d, T2V_ALGO_WINOGRAD_F4), Tt = batch * Tp;
    const int Cout_p = round_up(d->Cout, 128), Kp = round_up(x_cs, kBK);
    float* V = workspace;
    float* Md = V + (size_t)36 * Tt * x_cs;
    float* dU = Md + (size_t)36 * Tt * d->Cout;
    T2V_REQUIRE((long)36 * Tt * x_cs * 4 < 0x7fff0000L && (long)36 * Tt * d->Cout * 4 < 0x7fff0000L,
                "backward_weight_winograd: transformed tensors too large for 32-bit buffer offsets (split the batch)");
    if (stages & 1) {   // transforms of images [b0, b0 + nb) into their slots of the batch-wide tile lists
        // (x == null: the forward pass has put V of these images there -- t2v_conv2d_forward_winograd_keep_v)
        T2V_REQUIRE(dy, "backward_weight_winograd: null image pointers");
        for (int b = 0; b < nb; ++b) {
            if (x)
            T2V_TRY(launch_winograd4_input(s, x + (size_t)b * d->H * d->W * x_cs, V, d->H, d->W, x_cs, d->pad,
                                           d->pad_mode == T2V_PAD_REFLECT, batch, b0 + b));
            T2V_TRY(launch_winograd4_dy(s, dy + (size_t)b * Ho * Wo * dy_cs, Md, Ho, Wo, d->Cout, dy_cs, batch, b0 + b));
        }
    }
    if (!(stages & 2)) return T2V_OK;
    T2V_REQUIRE(dw_torch, "backward_weight_winograd: null gradient pointer");
    // 36 reductions over the tiles on the pixel-reduction GEMM: position xi = row xi of a [36][Tt] image, taken
    // by "tap" xi (input row offset xi, output row offset xi, its own output matrix)
    WgradParams w;
    memset(&w, 0, sizeof(w));
    w.x = V; w.dy = Md; w.dw = dU;
    w.batch = 1; w.Hin = 36; w.Win = Tt; w.Cin_s = x_cs;
    w.x_img_stride = (long)36 * Tt * x_cs; w.dy_img_stride = (long)36 * Tt * d->Cout;
    w.Wm = Tt; w.M = Tt;
    w.Hout = 36; w.Wout = Tt; w.Cout = d->Cout; w.Cout_s = d->Cout;
    w.stride = 1; w.ostride = 1;
    w.reflect = 0; w.accumulate = 0;
    w.ntiles = (d->Cout + 127) / 128; w.ctiles = (x_cs + 127) / 128;
    static_assert(kMaxTaps >= 36, "one tap slot per Winograd transform position");
    for (int xi = 0; xi < 36; ++xi) {
        w.tdy[xi] = xi; w.tdx[xi] = 0;
        w.toy[xi] = xi; w.tox[xi] = 0;
        w.tap_woff[xi] = (long)xi * Cout_p * Kp; w.tap_Kp[xi] = Kp; w.tap_kidx[xi] = 0;
    }
    w.ntaps = 36;
    w.splits = 1;
    w.dw_floats = (long)36 * Cout_p * Kp;
    T2V_TRY(check_async_errors());
    if (wino_wgrad_sk_ok(Tt, x_cs, d->Cout))
        T2V_TRY(launch_wino_wgrad_sk(s, V, Md, dU, dU + (size_t)36 * Cout_p * Kp, Tt, x_cs, d->Cout, Cout_p, Kp));
    else
        T2V_TRY(launch_conv_wgrad(s, w));
    return launch_winograd4_dw(s, dU, dw_torch, d->Cout, d->Cin, Cout_p, Kp, accumulate);
}

int t2v_conv2d_backward_weight_winograd_dy_norm(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int batch, int slot,
                                                int x_cs, const float* conv_out, const float* dy, const float* mean_rstd,
                                                const float* gamma, const float* beta, int relu, const float* sums,
                                                float* workspace) {
    T2V_REQUIRE(ctx && d && conv_out && dy && mean_rstd && sums && workspace && batch >= 1 && slot >= 0 && slot < batch,
                "backward_weight_winograd_dy_norm: bad arguments");
    T2V_REQUIRE(wgrad_winograd_ok(d, x_cs, d->Cout), "backward_weight_winograd_dy_norm: shape not supported "
                                                     "(t2v_conv_backward_weight_winograd_supported)");
    const int Tp = wino_tiles_padded(d, T2V_ALGO_WINOGRAD_F4), Tt = batch * Tp;
    T2V_REQUIRE((long)36 * Tt * d->Cout * 4 < 0x7fff0000L, "backward_weight_winograd_dy_norm: A dy A^T too large for 32-bit offsets");
    float* Md = workspace + (size_t)36 * Tt * x_cs;
    return launch_winograd4_dy_norm((hipStream_t)stream, dy, conv_out, mean_rstd, gamma, beta, relu, sums, Md, wino_out_h(d),
                                    wino_out_w(d), d->Cout, batch, slot);
}

int t2v_conv2d_backward_weight_winograd(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int batch, const float* x,
                                        int x_cs, const float* dy, int dy_cs, float* dw_torch, int accumulate,
                                        float* workspace) {
    T2V_REQUIRE(x && dy && dw_torch, "backward_weight_winograd: bad arguments");
    return t2v_conv2d_backward_weight_winograd_stages(ctx, stream, d, batch, 0, batch, x, x_cs, dy, dy_cs, dw_torch,
                                                      accumulate, workspace, 3);
}

// ---- data gradient by the transposed Winograd algorithm (winograd.hip: winograd4_dgrad_output_kernel) -------------------
static bool dgrad_winograd_ok(const t2v_conv_desc* d, int x_cs, int dy_cs) {
    return wgrad_winograd_ok(d, x_cs, dy_cs) && d->pad_mode == T2V_PAD_REFLECT && d->pad == 1 && d->H % 4 == 0 && d->W % 4 == 0 &&
           d->Cout % kBK == 0;
}
int t2v_conv_backward_data_winograd_supported(const t2v_conv_desc* d, int x_cs, int dy_cs) {
    return d && dgrad_winograd_ok(d, x_cs, dy_cs) ? 1 : 0;
}
size_t t2v_conv_backward_data_winograd_weight_floats(const t2v_conv_desc* d, int x_cs) {
    if (!d) return 0;
    return (size_t)36 * round_up(x_cs, 128) * round_up(d->Cout, kBK);
}
size_t t2v_conv_backward_data_winograd_scratch_floats(const t2v_conv_desc* d, int x_cs) {
    if (!d) return 0;
    return (size_t)36 * wino_tiles_padded(d, T2V_ALGO_WINOGRAD_F4) * x_cs + (size_t)(d->H + 2) * (d->W + 2) * x_cs +
           wino_gemm_sk_scratch_floats();
}
int t2v_conv_pack_weight_transposed(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int x_cs, const float* w_forward_dev,
                                    float* packed_dev) {
    T2V_REQUIRE(ctx && d && w_forward_dev && packed_dev, "pack_weight_transposed: null pointer");
    T2V_REQUIRE(dgrad_winograd_ok(d, x_cs, d->Cout), "pack_weight_transposed: shape not supported "
                                                     "(t2v_conv_backward_data_winograd_supported)");
    // U^T[xi][c][n]: the rows are the forward layer's INPUT channels, K runs over its OUTPUT channels
    return launch_winograd4_weight((hipStream_t)stream, w_forward_dev, packed_dev, /*rows*/ d->Cin, /*K*/ d->Cout,
                                   round_up(x_cs, 128), round_up(d->Cout, kBK), /*transpose, no flip*/ 2);
}
// the forward layer's own packed U [36][Cout_p][x_cs] serves as the B matrix [K = Cout][N = x_cs] of dV = dM U when the
// fixed-grid GEMM has its [K][N] form for the shape (conv_igemm.hip: wino_gemm_sk_kernel<.., BKN>) and no padding rows sit
// between the positions
static bool dgrad_winograd_forward_weights_ok(const t2v_conv_desc* d, int x_cs, int dy_cs) {
    if (!dgrad_winograd_ok(d, x_cs, dy_cs)) return false;
    t2v_conv_desc f = *d;
    f.algo = T2V_ALGO_WINOGRAD_F4;
    ConvPlan pl;
    if (!winograd_supported(&f, x_cs, f.algo) || build_conv_plan(&f, x_cs, false, &pl) != T2V_OK || pl.Cout_p != d->Cout) return false;
    return wino_gemm_sk_bkn_ok(36, wino_tiles_padded(d, T2V_ALGO_WINOGRAD_F4), d->Cout, x_cs, x_cs);
}
int t2v_conv_backward_data_winograd_takes_forward_weights(const t2v_conv_desc* d, int x_cs, int dy_cs) {
    return d && dgrad_winograd_forward_weights_ok(d, x_cs, dy_cs) ? 1 : 0;
}
static int backward_data_winograd(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int batch, int slot,
                                  const float* wgrad_workspace, int x_cs, const float* ut_packed, bool forward_weights,
                                  float* scratch, float* dx);
int t2v_conv2d_backward_data_winograd(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int batch, int slot,
                                      const float* wgrad_workspace, int x_cs, const float* ut_packed, float* scratch, float* dx) {
    return backward_data_winograd(ctx, stream, d, batch, slot, wgrad_workspace, x_cs, ut_packed, false, scratch, dx);
}
int t2v_conv2d_backward_data_winograd_fw(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int batch, int slot,
                                         const float* wgrad_workspace, int x_cs, const float* u_forward_packed, float* scratch,
                                         float* dx) {
    T2V_REQUIRE(d && dgrad_winograd_forward_weights_ok(d, x_cs, d->Cout),
                "backward_data_winograd_fw: shape not supported (t2v_conv_backward_data_winograd_takes_forward_weights)");
    return backward_data_winograd(ctx, stream, d, batch, slot, wgrad_workspace, x_cs, u_forward_packed, true, scratch, dx);
}
static int backward_data_winograd(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int batch, int slot,
                                  const float* wgrad_workspace, int x_cs, const float* ut_packed, bool forward_weights,
                                  float* scratch, float* dx) {
    T2V_REQUIRE(ctx && d && wgrad_workspace && ut_packed && scratch && dx && batch >= 1 && slot >= 0 && slot < batch,
                "backward_data_winograd: bad arguments");
    T2V_REQUIRE(dgrad_winograd_ok(d, x_cs, d->Cout), "backward_data_winograd: shape not supported");
    T2V_TRY(check_async_errors());
    hipStream_t s = (hipStream_t)stream;
    const int Tp = wino_tiles_padded(d, T2V_ALGO_WINOGRAD_F4), Tt = batch * Tp;
    // M_dy = A dy A^T of all `batch` images sits behind V in the weight gradient's workspace ([36][Tt][Cout] each)
    const float* Md = wgrad_workspace + (size_t)36 * Tt * x_cs;
    float* dV = scratch;
    float* dxp = scratch + (size_t)36 * Tp * x_cs;
    // dV[xi][t][c] = sum_n M_dy[xi][slot*Tp + t][n] * U[xi][n][c]: the batched GEMM of a conv with the channel roles swapped,
    // reading its rows out of the batch-wide matrix (input row pitch Tt, Tp rows per position)
    if (forward_weights || (wino_gemm_sk_ok(36, Tp, d->Cout, x_cs, x_cs) && round_up(x_cs, 128) == x_cs)) {
        SkGemm g;
        g.b_kn = forward_weights;
        g.a = Md + (size_t)slot * Tp * d->Cout; g.b = ut_packed; g.c = dV;
        g.scratch = dxp + (size_t)(d->H + 2) * (d->W + 2) * x_cs;
        g.err = async_error_word();
        g.a_group_stride = (long)Tt * d->Cout;
        g.groups = 36; g.T = Tp; g.K = d->Cout; g.N = x_cs; g.c_cs = x_cs;
        T2V_TRY(launch_wino_gemm_sk(s, g));
    } else {
        t2v_conv_desc da = *d;
        da.Cin = d->Cout;
        da.Cout = d->Cin;
        da.algo = T2V_ALGO_WINOGRAD_F4;
        ConvPlan pl;
        T2V_TRY(build_winograd_gemm_plan(&da, &pl));
        T2V_REQUIRE((long)36 * Tt * d->Cout * 4 < 0x7fff0000L, "backward_data_winograd: M_dy too large for 32-bit buffer offsets");
        pl.kp.Win = Tt;
        T2V_TRY(run_conv(ctx, s, pl, Md + (size_t)slot * Tp * d->Cout, ut_packed, nullptr, dV, x_cs, nullptr));
    }
    T2V_TRY(launch_winograd4_dgrad_output(s, dV, dxp, d->H, d->W, x_cs));
    return launch_reflect_pad_backward(s, dxp, dx, d->H, d->W, x_cs, 1);
}

int t2v_conv_unpack_weight_into(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int x_cs, const float* packed_dev,
                                float* w_torch_dev, int accumulate) {
    T2V_REQUIRE(ctx && packed_dev && w_torch_dev, "unpack_weight: null pointer");
    ConvPlan pl;
    T2V_TRY(build_conv_plan(d, x_cs, false, &pl));
    if (!d->transposed)
        return launch_unpack_conv_weight((hipStream_t)stream, packed_dev, w_torch_dev, d->Cout, d->Cin, d->kH, d->kW, x_cs,
                                         pl.kp.ph[0].Kp, accumulate);
    return launch_unpack_convT_weight((hipStream_t)stream, packed_dev, w_torch_dev, d->Cin, d->Cout, x_cs, pl.Cout_p,
                                      d->kH, d->pad, accumulate);
}
int t2v_conv_unpack_weight(t2v_ctx* ctx, void* stream, const t2v_conv_desc* d, int x_cs, const float* packed_dev,
                           float* w_torch_dev) {
    return t2v_conv_unpack_weight_into(ctx, stream, d, x_cs, packed_dev, w_torch_dev, 0);
}
int t2v_accumulate(t2v_ctx* ctx, void* stream, float* dst, const float* src, long n, int overwrite) {
    T2V_REQUIRE(ctx && dst && src && n > 0, "accumulate: bad arguments");
    return launch_accumulate((hipStream_t)stream, dst, src, n, overwrite);
}
int t2v_scale(t2v_ctx* ctx, void* stream, float* x, long n, float s) {
    T2V_REQUIRE(ctx && x && n > 0, "scale: bad arguments");
    return launch_scale((hipStream_t)stream, x, n, s);
}
int t2v_unzip2(t2v_ctx* ctx, void* stream, const float* src, float* dst0, float* dst1, int C, int overwrite) {
    T2V_REQUIRE(ctx && src && dst0 && dst1 && C > 0, "unzip2: bad arguments");
    return launch_unzip2((hipStream_t)stream, src, dst0, dst1, C, overwrite);
}
int t2v_zero(t2v_ctx* ctx, void* stream, void* ptr, size_t bytes) {
    T2V_REQUIRE(ctx && ptr, "zero: null pointer");
    T2V_HIP_CHECK(hipMemsetAsync(ptr, 0, bytes, (hipStream_t)stream));
    return T2V_OK;
}

int t2v_channel_sum(t2v_ctx* ctx, void* stream, const float* x, long npix, int C, int cs, float* scratch, float* out) {
    T2V_REQUIRE(ctx && x && out && scratch && C > 0 && cs >= C, "channel_sum: bad arguments");
    return launch_channel_sum((hipStream_t)stream, x, npix, C, cs, scratch, out);
}

int t2v_reflect_pad_backward(t2v_ctx* ctx, void* stream, const float* dxp, float* dx, int H, int W, int C, int pad) {
    T2V_REQUIRE(ctx && dxp && dx && pad >= 0 && H > 2 * pad && W > 2 * pad, "reflect_pad_backward: bad arguments");
    return launch_reflect_pad_backward((hipStream_t)stream, dxp, dx, H, W, C, pad);
}
int t2v_instance_norm_backward(t2v_ctx* ctx, void* stream, const float* x, const float* dy, const float* mean_rstd,
                               const float* gamma, const float* beta, int relu, long npix, int C, float* scratch,
                               float* dx, float* dbeta_dgamma) {
    // (dx == null: the two sums only -- the gradient itself is formed by its consumer, t2v_conv2d_backward_weight_winograd_dy_norm)
    T2V_REQUIRE(ctx && x && dy && mean_rstd && scratch && dbeta_dgamma && npix > 0 && C > 0,
                "instance_norm_backward: bad arguments");
    return launch_inorm_backward((hipStream_t)stream, x, dy, mean_rstd, gamma, beta, relu, npix, C, scratch, dx,
                                 dbeta_dgamma);
}
int t2v_instance_norm_backward_affine(t2v_ctx* ctx, void* stream, const float* x, const float* dy, const float* mean_rstd,
                                      const float* gamma, const float* beta, int relu, long npix, int C, float* scratch,
                                      float* dx, float* dbeta_dgamma, float* d_beta, float* d_gamma, int overwrite) {
    T2V_REQUIRE(ctx && x && dy && mean_rstd && scratch && dbeta_dgamma && d_beta && d_gamma && npix > 0 && C > 0,
                "instance_norm_backward_affine: bad arguments");
    return launch_inorm_backward((hipStream_t)stream, x, dy, mean_rstd, gamma, beta, relu, npix, C, scratch, dx,
                                 dbeta_dgamma, d_beta, d_gamma, overwrite);
}
int t2v_act_backward(t2v_ctx* ctx, void* stream, const float* dy, const float* y, int act, float slope, long n,
                     float* dpre) {
    T2V_REQUIRE(ctx && dy && y && dpre && n > 0, "act_backward: bad arguments");
    return launch_act_backward((hipStream_t)stream, dy, y, act, slope, n, dpre);
}
int t2v_avgpool3x3s2_backward(t2v_ctx* ctx, void* stream, const float* dy, float* dx, int H, int W, int C) {
    T2V_REQUIRE(ctx && dy && dx, "avgpool_backward: null pointer");
    return launch_avgpool3s2_backward((hipStream_t)stream, dy, dx, H, W, C);
}
int t2v_maxpool2x2(t2v_ctx* ctx, void* stream, const float* x, float* y, int H, int W, int C) {
    T2V_REQUIRE(ctx && x && y && H >= 2 && W >= 2 && C >= 1, "maxpool2x2: bad arguments");
    return launch_maxpool2x2((hipStream_t)stream, x, y, H, W, C);
}
int t2v_maxpool2x2_backward(t2v_ctx* ctx, void* stream, const float* x, const float* dy, float* dx, int H, int W, int C) {
    T2V_REQUIRE(ctx && x && dy && dx && H >= 2 && W >= 2 && C >= 1, "maxpool2x2_backward: bad arguments");
    return launch_maxpool2x2_backward((hipStream_t)stream, x, dy, dx, H, W, C);
}
int t2v_sum_sq_diff_const_backward(t2v_ctx* ctx, void* stream, const float* x, float c, float scale, long n, float* dx) {
    T2V_REQUIRE(ctx && x && dx && n > 0, "mse backward: bad arguments");
    return launch_loss_backward((hipStream_t)stream, 0, x, nullptr, c, scale, n, dx);
}
int t2v_sum_abs_diff_backward(t2v_ctx* ctx, void* stream, const float* a, const float* b, float scale, long n, float* da) {
    T2V_REQUIRE(ctx && a && b && da && n > 0, "l1 backward: bad arguments");
    return launch_loss_backward((hipStream_t)stream, 1, a, b, 0.f, scale, n, da);
}

int t2v_sum_sq_diff_const(t2v_ctx* ctx, void* stream, const float* x, float c, long n, float* scratch, float* out) {
    T2V_REQUIRE(ctx && x && scratch && out && n > 0, "sum_sq_diff_const: bad arguments");
    return launch_reduce((hipStream_t)stream, 0, x, nullptr, c, n, scratch, out);
}
int t2v_sum_abs_diff(t2v_ctx* ctx, void* stream, const float* a, const float* b, long n, float* scratch, float* out) {
    T2V_REQUIRE(ctx && a && b && scratch && out && n > 0, "sum_abs_diff: bad arguments");
    return launch_reduce((hipStream_t)stream, 1, a, b, 0.f, n, scratch, out);
}
int t2v_sum_abs_diff_masked(t2v_ctx* ctx, void* stream, const float* a, const float* b, const float* mask, long npix,
                            int c0, int C, int cs, float* scratch, float* out) {
    T2V_REQUIRE(ctx && a && scratch && out && npix > 0 && c0 >= 0 && C > 0 && cs >= c0 + C,
                "sum_abs_diff_masked: bad arguments");
    return launch_masked_l1((hipStream_t)stream, a, b, mask, npix, c0, C, cs, scratch, out);
}
int t2v_sum_abs_diff_masked_backward(t2v_ctx* ctx, void* stream, const float* a, const float* b, const float* mask,
                                     float scale, long npix, int c0, int C, int cs, float* da) {
    T2V_REQUIRE(ctx && a && da && npix > 0 && c0 >= 0 && C > 0 && cs >= c0 + C,
                "sum_abs_diff_masked_backward: bad arguments");
    return launch_masked_l1_backward((hipStream_t)stream, a, b, mask, scale, npix, c0, C, cs, da);
}
int t2v_loss_terms(t2v_ctx* ctx, void* stream, const int64_t* term_ptrs, const int32_t* term_ints, const float* term_floats,
                   const int32_t* chunk_term, const int64_t* chunk_off, const int32_t* term_chunk0, int nterms, int nchunks,
                   int chunk, float* partials, float* out) {
    T2V_REQUIRE(ctx && term_ptrs && term_ints && term_floats && chunk_term && chunk_off && term_chunk0 && partials && out,
                "loss_terms: null pointer");
    T2V_REQUIRE(nterms > 0 && nchunks > 0 && chunk > 0 && chunk % 4 == 0, "loss_terms: nterms %d nchunks %d chunk %d", nterms,
                nchunks, chunk);
    return launch_loss_terms((hipStream_t)stream, reinterpret_cast<const long long*>(term_ptrs), term_ints, term_floats,
                             chunk_term, reinterpret_cast<const long long*>(chunk_off), term_chunk0, nterms, nchunks, chunk,
                             partials, out);
}
int t2v_adam_step(t2v_ctx* ctx, void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                  long n, double lr, double beta1, double beta2, double eps, int step) {
    T2V_REQUIRE(ctx && param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adam_step: bad arguments");
    return launch_adam((hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step);
}

int t2v_adam_step_multi(t2v_ctx* ctx, void* stream, const int64_t* ptrs, const int64_t* nelem, const float* step_size,
                        const int32_t* chunk_tensor, const int64_t* chunk_off, int nchunks, int chunk, double beta1,
                        double beta2, double eps) {
    T2V_REQUIRE(ctx && ptrs && nelem && step_size && chunk_tensor && chunk_off && nchunks > 0 && chunk > 0,
                "adam_step_multi: bad arguments");
    return launch_adam_multi((hipStream_t)stream, reinterpret_cast<const long long*>(ptrs),
                             reinterpret_cast<const long long*>(nelem), step_size, chunk_tensor,
                             reinterpret_cast<const long long*>(chunk_off), nchunks, chunk, beta1, beta2, eps);
}

int t2v_batch_norm_update_running(t2v_ctx* ctx, void* stream, const float* mean_rstd, float* running_mean,
                                  float* running_var, long n, int C, float momentum, float eps) {
    T2V_REQUIRE(ctx && mean_rstd && running_mean && running_var && C > 0 && n > 1,
                "batch_norm_update_running: bad arguments (needs more than one value per channel)");
    return launch_bn_running_update((hipStream_t)stream, mean_rstd, running_mean, running_var, n, momentum, eps, C);
}

int t2v_instance_norm_apply(t2v_ctx* ctx, void* stream, const float* x, const float* mean_rstd, const float* gamma,
                            const float* beta, const float* res1, const float* res2, float* y, long npix, int C,
                            int relu) {
    T2V_REQUIRE(ctx && x && mean_rstd && y, "inorm_apply: null pointer");
    T2V_REQUIRE((gamma == nullptr) == (beta == nullptr), "inorm_apply: gamma and beta must both be given or both NULL");
    return launch_inorm_apply((hipStream_t)stream, x, mean_rstd, gamma, beta, res1, res2, y, npix, C, relu);
}

int t2v_flow_warp_composite(t2v_ctx* ctx, void* stream, const float* raw, const float* fw, const float* prev,
                            int prev_cs, int prev_c0, float* out, float* warp_out, int H, int W) {
    T2V_REQUIRE(ctx && fw && prev && (out || warp_out), "flow_warp_composite: null pointer");
    T2V_REQUIRE(raw == nullptr || out != nullptr, "flow_warp_composite: raw without out");
    T2V_REQUIRE(H > 1 && W > 1, "flow_warp_composite: H,W must be > 1");
    T2V_REQUIRE(prev_c0 >= 0 && prev_c0 + 3 <= prev_cs, "flow_warp_composite: prev_c0 + 3 > prev_cs");
    return launch_warp_composite((hipStream_t)stream, raw, fw, prev, prev_cs, prev_c0, out, warp_out, H, W);
}
int t2v_flow_warp_composite_backward(t2v_ctx* ctx, void* stream, const float* d_out, const float* d_warp,
                                     const float* raw, const float* fw, const float* prev, int prev_cs, int prev_c0,
                                     float* d_raw, float* d_fw, float* d_prev, int H, int W) {
    T2V_REQUIRE(ctx && fw && prev && d_fw && (d_out || d_warp), "flow_warp_composite_backward: null pointer");
    T2V_REQUIRE(d_out == nullptr || raw != nullptr, "flow_warp_composite_backward: d_out needs raw (the blend's operands)");
    T2V_REQUIRE(H > 1 && W > 1, "flow_warp_composite_backward: H,W must be > 1");
    T2V_REQUIRE(prev_c0 >= 0 && prev_c0 + 3 <= prev_cs, "flow_warp_composite_backward: prev_c0 + 3 > prev_cs");
    return launch_warp_composite_backward((hipStream_t)stream, d_out, d_warp, raw, fw, prev, prev_cs, prev_c0, d_raw,
                                          d_fw, d_prev, H, W);
}

int t2v_avgpool3x3s2(t2v_ctx* ctx, void* stream, const float* x, float* y, int H, int W, int C) {
    T2V_REQUIRE(ctx && x && y, "avgpool: null pointer");
    return launch_avgpool3s2((hipStream_t)stream, x, y, H, W, C);
}

int t2v_nchw_to_nhwc(t2v_ctx* ctx, void* stream, const float* src, float* dst, int C, int H, int W, int dst_cs) {
    T2V_REQUIRE(ctx && src && dst && dst_cs >= C, "nchw_to_nhwc: bad arguments");
    return launch_nchw_to_nhwc((hipStream_t)stream, src, dst, C, H, W, dst_cs);
}
int t2v_nhwc_to_nchw(t2v_ctx* ctx, void* stream, const float* src, float* dst, int C, int H, int W, int src_cs) {
    T2V_REQUIRE(ctx && src && dst && src_cs >= C, "nhwc_to_nchw: bad arguments");
    return launch_nhwc_to_nchw((hipStream_t)stream, src, dst, C, H, W, src_cs);
}
int t2v_pose_u8_to_f32(t2v_ctx* ctx, void* stream, const uint8_t* src_hwc3, float* dst, long npix, int dst_cs,
                       int c0) {
    T2V_REQUIRE(ctx && src_hwc3 && dst && c0 + 3 <= dst_cs, "pose_u8_to_f32: bad arguments");
    return launch_u8_pose_to_f32((hipStream_t)stream, src_hwc3, dst, npix, dst_cs, c0);
}
int t2v_tensor2im_u8(t2v_ctx* ctx, void* stream, const float* x, uint8_t* y, long n) {
    T2V_REQUIRE(ctx && x && y, "tensor2im: null pointer");
    return launch_to_u8((hipStream_t)stream, x, y, n);
}
int t2v_copy_channels(t2v_ctx* ctx, void* stream, const float* src, int src_cs, int src_c0, float* dst, int dst_cs,
                      int dst_c0, int nc, long npix) {
    T2V_REQUIRE(ctx && src && dst && src_c0 + nc <= src_cs && dst_c0 + nc <= dst_cs, "copy_channels: bad arguments");
    return launch_copy_channels((hipStream_t)stream, src, src_cs, src_c0, dst, dst_cs, dst_c0, nc, npix);
}

// ---- host plumbing (ABI 14): buffers, copies, streams, events for a host without a HIP binding of its own ----
int t2v_device_malloc(t2v_ctx* ctx, size_t bytes, void** out) {
    T2V_REQUIRE(ctx && out, "device_malloc: null pointer");
    T2V_HIP_CHECK(hipSetDevice(ctx->device));
    *out = nullptr;
    if (bytes == 0) return T2V_OK;
    T2V_HIP_CHECK(hipMalloc(out, bytes));
    return T2V_OK;
}
int t2v_device_free(t2v_ctx* ctx, void* ptr) {
    T2V_REQUIRE(ctx, "device_free: null context");
    if (!ptr) return T2V_OK;
    T2V_HIP_CHECK(hipSetDevice(ctx->device));
    T2V_HIP_CHECK(hipFree(ptr));
    return T2V_OK;
}
int t2v_host_malloc(t2v_ctx* ctx, size_t bytes, void** out) {
    T2V_REQUIRE(ctx && out && bytes > 0, "host_malloc: bad arguments");
    T2V_HIP_CHECK(hipSetDevice(ctx->device));
    T2V_HIP_CHECK(hipHostMalloc(out, bytes, hipHostMallocPortable));
    return T2V_OK;
}
int t2v_host_free(t2v_ctx* ctx, void* ptr) {
    T2V_REQUIRE(ctx, "host_free: null context");
    if (!ptr) return T2V_OK;
    T2V_HIP_CHECK(hipHostFree(ptr));
    return T2V_OK;
}
int t2v_memcpy(t2v_ctx* ctx, void* stream, void* dst, const void* src, size_t bytes, int kind) {
    T2V_REQUIRE(ctx && (bytes == 0 || (dst && src)), "memcpy: null pointer");
    T2V_REQUIRE(kind == T2V_COPY_H2D || kind == T2V_COPY_D2H || kind == T2V_COPY_D2D, "memcpy: kind %d", kind);
    if (bytes == 0) return T2V_OK;
    T2V_HIP_CHECK(hipSetDevice(ctx->device));
    hipMemcpyKind k = kind == T2V_COPY_H2D ? hipMemcpyHostToDevice
                      : (kind == T2V_COPY_D2H ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
    T2V_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, k, (hipStream_t)stream));
    return T2V_OK;
}
int t2v_stream_create(t2v_ctx* ctx, void** out) {
    T2V_REQUIRE(ctx && out, "stream_create: null pointer");
    T2V_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = nullptr;
    T2V_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = s;
    return T2V_OK;
}
int t2v_stream_destroy(t2v_ctx* ctx, void* stream) {
    T2V_REQUIRE(ctx, "stream_destroy: null context");
    if (stream) T2V_HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
    return T2V_OK;
}
int t2v_stream_synchronize(t2v_ctx* ctx, void* stream) {
    T2V_REQUIRE(ctx, "stream_synchronize: null context");
    T2V_HIP_CHECK(hipSetDevice(ctx->device));
    T2V_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return T2V_OK;
}
int t2v_event_create(t2v_ctx* ctx, void** out) {
    T2V_REQUIRE(ctx && out, "event_create: null pointer");
    T2V_HIP_CHECK(hipSetDevice(ctx->device));
    hipEvent_t e = nullptr;
    T2V_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *out = e;
    return T2V_OK;
}
int t2v_event_record(t2v_ctx* ctx, void* event, void* stream) {
    T2V_REQUIRE(ctx && event, "event_record: null pointer");
    T2V_HIP_CHECK(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return T2V_OK;
}
int t2v_event_synchronize(t2v_ctx* ctx, void* event) {
    T2V_REQUIRE(ctx && event, "event_synchronize: null pointer");
    T2V_HIP_CHECK(hipEventSynchronize((hipEvent_t)event));
    return T2V_OK;
}
int t2v_event_destroy(t2v_ctx* ctx, void* event) {
    T2V_REQUIRE(ctx, "event_destroy: null context");
    if (event) T2V_HIP_CHECK(hipEventDestroy((hipEvent_t)event));
    return T2V_OK;
}
int t2v_device_synchronize(t2v_ctx* ctx) {
    T2V_REQUIRE(ctx, "device_synchronize: null context");
    T2V_HIP_CHECK(hipSetDevice(ctx->device));
    T2V_HIP_CHECK(hipDeviceSynchronize());
    return T2V_OK;
}

}  // extern "C"
