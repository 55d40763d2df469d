// conv_wgrad.hip -- fp32 weight-gradient of the generator / discriminator convolutions on gfx950.
//
// Backward-by-weight of SpatialConvolutionMM / SpatialFullDilatedConvolution (THCUNN.h:664,794
// `accGradParameters`): for a (forward) convolution y[m][n] = sum_{tap,c} x[pix(m,tap)][c] * w[n][tap][c]
//   dW[n][tap][c] = sum_{b, m} dY_b[m][n] * X_b[pix(m, tap)][c]
// i.e. one GEMM per tap that REDUCES OVER PIXELS.  With NHWC both operands are already stored
// pixel-major ([pixel][channel], channel contiguous), which is exactly the layout the fp32 MFMA
// wants when the reduction index is the pixel: lane i reads element [pixel k][channel i] with a
// conflict-free ds_read_b32 -- no transposes, no swizzle.
//
//   block tile : 128 output channels (n) x 128 input channels (c) of ONE tap, 64 accumulator VGPRs/wave
//   K loop     : 16 pixels per LDS stage (2 x 8 KiB), 4-slot ring = 64 KiB (two blocks per CU), loader waves with
//                buffer-addressed LDS-DMA (same wave-specialised structure as conv_igemm.hip); padding / tails = OOB lanes.
//                (Round 1 ran 32-pixel stages on a 2-slot ring; halving the stage is worth +3-4.5 % on every layer
//                shape -- rb1024 0.735 -> 0.706 ms, stride-2 / transposed layers 106 -> 110 TF -- whatever the ring
//                depth, 3 / 4 / 5 slots measure alike: T2V_WGRAD_PIX / T2V_WGRAD_RING select the variants)
//   grid       : (Cout/128) x (Cin_s/128) x taps  [x phases for transposed convs]
//   output     : written straight into the PACKED weight layout [Cout_p][Kp] (K = tap*Cin_s + c), so the
//                optimiser can run on packed parameters; t2v_conv_unpack_weight converts back.
// Transposed convolutions are the same reduction per sub-pixel phase: dY is sampled at the phase's
// strided output positions (ostride / toy / tox), X at the phase taps (t2v_conv2d_backward_weight).
#include <stdlib.h>

#include "t2v_internal.h"
#include "norm_pool.h"
#include "k_loops.h"

namespace t2v {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void wg_dma16(const float* base, int nbytes, char* lds_dst, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, nbytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
#endif
}

// ---- the MFMA waves' pixel loop, shared by conv_wgrad_kernel and wino_wgrad_sk_kernel ---------------------------------------
// 2 x 2 waves, each 64 (n) x 64 (c) = 2 x 2 tiles of 32x32.  Software pipeline (one MFMA wave per SIMD has nobody to hide
// its LDS latency): a stage's PIX/2 pixel pairs run as PIX/8 groups of 4 pairs = 16 MFMAs; the operands of group q+1 are
// read in the shadow of group q's MFMAs (issue order MFMA, ds_read, MFMA, ds_read, ...); the stage barrier sits before
// the last group's reads, which are the first of the next stage.  Begins with the B0 barrier, runs `nstages` stages out of
// ring slots 0, 1, ...; the caller closes.
template <int PIX, int RING>
__device__ __forceinline__ void wgrad_mfma_loop(const char* smem, int nstages, int wn, int wc, int fi, int kk, f32x16 (&acc)[2][2]) {
    constexpr int kStage = 2 * PIX * 128 * 4;
    constexpr int NQ = PIX / 8;
    constexpr int U = (NQ % 2) ? 2 : 1;   // stages per unrolled iteration: keeps the operand register set of a group static
    float av[2][4][2], bv[2][4][2];
    auto load_group = [&](int buf, int q, int set) {
        const float* sY = reinterpret_cast<const float*>(smem + buf * kStage);
        const float* sX = sY + PIX * 128;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int px = 8 * q + 2 * e + kk;
#pragma unroll
            for (int i = 0; i < 2; ++i) av[set][e][i] = sY[px * 128 + wn * 64 + i * 32 + fi];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[set][e][j] = sX[px * 128 + wc * 64 + j * 32 + fi];
        }
    };
    __syncthreads();  // B0
    load_group(0, 0, 0);
    int buf = 0;
    for (int kt0u = 0; kt0u < nstages; kt0u += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (kt0u + u >= nstages) break;   // wave-uniform
            const int nbuf = buf == RING - 1 ? 0 : buf + 1;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int cur = (u * NQ + q) & 1;
                __builtin_amdgcn_sched_barrier(0);
                if (q + 1 == NQ) {
                    __syncthreads();  // barrier(kt): slot `buf` fully read, stage kt+1 visible
                    __builtin_amdgcn_sched_barrier(0);
                    load_group(nbuf, 0, cur ^ 1);
                } else {
                    load_group(buf, q + 1, cur ^ 1);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][e][i], bv[cur][e][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
            buf = nbuf;
        }
    }
}

// PIX pixels per stage (dY tile + X tile, 128 channels each = PIX KiB), RING slots: 32 x 2 (a stage = 4096 MFMA cycles
// per wave, its DMA has < 1 stage time to land) or 16 x 4 (2048-cycle stages, DMA three stages ahead); both 64 KiB,
// i.e. two blocks per CU
// REFLECT: the forward conv used reflection padding (taps never fall outside; indices mirror)
template <bool REFLECT, int PIX, int RING>
__global__ __launch_bounds__(512) void conv_wgrad_kernel(const WgradParams p) {
    constexpr int kWgPix = PIX, kWgRing = RING;
    constexpr int kWgStage = 2 * kWgPix * 128 * 4;
    constexpr int RW = PIX / 4;    // pixel rows of a stage per loader wave
    constexpr int NI = PIX / 8;    // DMA instructions per loader wave, stage and operand (2 pixel rows each)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kOOB = 0x7fff0000;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = wave >= 4;
    const int wid = wave & 3;

    // block -> (tap, n tile, c tile); consecutive blocks share the dY tile (n) across c tiles
    int t = blockIdx.x;
    const int ct = t % p.ctiles; t /= p.ctiles;
    const int nt = t % p.ntiles; t /= p.ntiles;
    const int tap = t % p.ntaps; t /= p.ntaps;   // global tap index (all phases)
    const int split = t;
    const int n0 = nt * 128, c0 = ct * 128;
    const int dy = p.fold ? 0 : p.tdy[tap], dx = p.fold ? 0 : p.tdx[tap];
    const bool fold_n = p.fold == 2;
    const int P = p.batch * p.M;              // pixels to reduce over
    const int nk_all = (P + kWgPix - 1) / kWgPix;
    const int per = (nk_all + p.splits - 1) / p.splits;
    const int kt0 = split * per;              // this block's stage range [kt0, kt0 + nk)
    const int nk = max(0, min(per, nk_all - kt0));

    if (is_loader) {
        // each loader wave: NI dY instructions + NI X instructions per stage; one instruction = 2 pixel rows
        const int prow = lane >> 5;           // pixel row inside the instruction's pair
        const int chunk = lane & 31;          // 16-byte chunk inside the 512-byte channel row
        bool n_ok = n0 + chunk * 4 < p.Cout_s;
        // folded taps: this lane's chunk is 4 channels of tap (c0 + 4*chunk) / Cin_s, fixed for the whole block
        int ldy = dy, ldx = dx, lc = c0 + chunk * 4;
        bool c_ok = lc < p.Cin_s;
        int ny = 0, nx = 0, ln = n0 + chunk * 4;     // fold on the dY side: this lane's tap shift and channel
        if (fold_n) {
            const int ltap = ln / p.Cout_s;
            ln -= ltap * p.Cout_s;
            n_ok = ltap < p.fold_taps;
            ny = ltap / p.KW;
            nx = ltap - ny * p.KW;
        } else if (p.fold) {
            const int kidx = c0 + chunk * 4;
            const int ltap = kidx / p.Cin_s;
            lc = kidx - ltap * p.Cin_s;
            c_ok = ltap < p.fold_taps;
            const int kh = ltap / p.KW;
            ldy = kh - p.pad;
            ldx = ltap - kh * p.KW - p.pad;
        }
        const int dy_bytes = p.batch * p.Hout * p.Wout * p.Cout_s * 4;
        const int x_bytes = p.batch * p.Hin * p.Win * p.Cin_s * 4;
        // (image, row, column) of this lane's four pixels in the stage being issued: set up once with divisions,
        // then stepped by kWgPix pixels per stage (stages are issued in increasing order)
        const int Hm = p.M / p.Wm;
        int pb[NI], py[NI], px[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int pidx = kt0 * kWgPix + wid * RW + i * 2 + prow;   // global pixel (image-major)
            pb[i] = pidx / p.M;
            const int m = pidx - pb[i] * p.M;
            py[i] = m / p.Wm;
            px[i] = m - py[i] * p.Wm;
        }
        int cur = 0;
        auto issue_stage = [&](int kt, int slot) {
            char* sY = smem + slot * kWgStage;
            char* sX = sY + kWgPix * 512;
            if (kt > cur) {   // kt == cur + 1
                cur = kt;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    px[i] += kWgPix;
                    while (px[i] >= p.Wm) { px[i] -= p.Wm; ++py[i]; }
                    while (py[i] >= Hm) { py[i] -= Hm; ++pb[i]; }
                }
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int b = pb[i], my = py[i], mx = px[i];
                const bool ok = b < p.batch;
                // dY: output pixel of GEMM pixel m (strided for the sub-pixel phases of a transposed conv)
                int opix = (b * p.Hout + my * p.ostride + p.toy[tap]) * p.Wout + mx * p.ostride + p.tox[tap];
                bool oky = ok && n_ok;
                if (fold_n) {   // GEMM pixel = pixel q of the padded input; y pixel q - (kh, kw), zero outside
                    const int oy = my - ny, ox = mx - nx;
                    oky = oky && (unsigned)oy < (unsigned)p.Hout && (unsigned)ox < (unsigned)p.Wout;
                    opix = (b * p.Hout + oy) * p.Wout + ox;
                }
                const int vy = oky ? (opix * p.Cout_s + (fold_n ? ln : n0 + chunk * 4)) * 4 : kOOB;
                wg_dma16(p.dy, dy_bytes, sY + (wid * RW + i * 2) * 512, vy, 0);
                // X: gathered through the tap
                int iy = my * p.stride + ldy, ix = mx * p.stride + ldx;
                bool okx = ok && c_ok;
                if constexpr (REFLECT) {
                    iy = iy < 0 ? -iy : iy;
                    ix = ix < 0 ? -ix : ix;
                    iy = min(iy, 2 * p.Hin - 2 - iy);
                    ix = min(ix, 2 * p.Win - 2 - ix);
                } else {
                    okx = okx && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
                }
                const int vx = okx ? (((b * p.Hin + iy) * p.Win + ix) * p.Cin_s + lc) * 4 : kOOB;
                wg_dma16(p.x, x_bytes, sX + (wid * RW + i * 2) * 512, vx, 0);
            }
        };
        // Fast path (every layer of the generator but the folded stems / heads): the GEMM rows are whole stages long, so the
        // 16 pixels of a stage are one run of one output row -- the stage's (image, row, first column) live in SGPRs, stepped
        // with scalar compares, dY needs a scalar offset only and X one add, one multiply-add, one compare and one select per
        // DMA instruction.  The general path above spends ~55 vector instructions per instruction on stepping and index
        // arithmetic (481 VALU per 128 MFMA in the PMC passes of round 4: profiles/pmc/r04_wgrad_s2_*), which the MFMA waves
        // of the same SIMDs pay for in issue slots.
        const bool fast = !REFLECT && p.fold == 0 && (p.Wm % kWgPix) == 0 && (p.M % kWgPix) == 0;
        if (fast) {
            int sb, sy, sx;
            {
                const int p0 = kt0 * kWgPix;
                sb = p0 / p.M;
                const int m = p0 - sb * p.M;
                sy = m / p.Wm;
                sx = m - sy * p.Wm;
            }
            sb = __builtin_amdgcn_readfirstlane(sb);
            sy = __builtin_amdgcn_readfirstlane(sy);
            sx = __builtin_amdgcn_readfirstlane(sx);
            int vy_c[NI], lx0[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int lp = wid * RW + i * 2 + prow;           // this lane's pixel inside the stage
                vy_c[i] = n_ok ? ((lp * p.ostride) * p.Cout_s + n0 + chunk * 4) * 4 : kOOB;
                lx0[i] = lp * p.stride + ldx;
            }
            const int xrow_bytes = p.Win * p.Cin_s * 4;
            const int x_img_bytes = p.Hin * xrow_bytes;
            const int dy_img_bytes = p.Hout * p.Wout * p.Cout_s * 4;
            int curf = 0;
            auto issue_fast = [&](int kt, int slot) {
                char* sY = smem + slot * kWgStage;
                char* sX = sY + kWgPix * 512;
                if (kt > curf) {   // kt == curf + 1
                    curf = kt;
                    sx += kWgPix;
                    if (sx >= p.Wm) {
                        sx = 0;
                        if (++sy >= Hm) {
                            sy = 0;
                            ++sb;
                        }
                    }
                }
                const bool inb = sb < p.batch;
                // the stage's image has its own buffer resource: images sit x_img_stride / dy_img_stride floats apart -- the
                // batch's own pitch, or whatever separates the buffers of two frames (t2v_conv2d_backward_weight_strided)
                const int sbc = inb ? sb : 0;
                const float* xb = p.x + (long)sbc * p.x_img_stride;
                const float* yb = p.dy + (long)sbc * p.dy_img_stride;
                const int soff_y = (((sy * p.ostride + p.toy[tap]) * p.Wout) + sx * p.ostride + p.tox[tap]) * p.Cout_s * 4;
                const int iy = sy * p.stride + ldy;
                const bool rowok = inb && (unsigned)iy < (unsigned)p.Hin;
                const int soff_x = rowok ? iy * xrow_bytes : 0;
                const int ix0 = sx * p.stride;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    wg_dma16(yb, dy_img_bytes, sY + (wid * RW + i * 2) * 512, inb ? vy_c[i] : kOOB, inb ? soff_y : 0);
                    const int ix = ix0 + lx0[i];
                    const int vx = (rowok && c_ok && (unsigned)ix < (unsigned)p.Win) ? (ix * p.Cin_s + lc) * 4 : kOOB;
                    wg_dma16(xb, x_img_bytes, sX + (wid * RW + i * 2) * 512, vx, soff_x);
                }
            };
            loader_k_loop<kWgRing, 2 * NI>(0, nk, issue_fast);
            return;
        }
        loader_k_loop<kWgRing, 2 * NI>(0, nk, issue_stage);   // (nk == 0, an empty split range: stage 0 once, B0 only)
        return;
    }

    // ---- MFMA waves: 2x2 waves, each 64 (n) x 64 (c) = 2x2 tiles of 32x32 ----
    const int wn = wid >> 1, wc = wid & 1;
    const int fi = lane & 31, kk = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    wgrad_mfma_loop<PIX, RING>(smem, nk, wn, wc, fi, kk, acc);

    // ---- epilogue: D[row n][col c] -> packed dW[n][koff + c]  (koff: position of this tap in K) ----
    const bool combine = p.tickets != nullptr;   // splits > 1: this block's tile is a partial to be summed in-kernel
    float* out = p.dw + (size_t)split * p.dw_floats + p.tap_woff[tap];
    const int Kp = p.tap_Kp[tap];
    const int kbase = p.fold == 1 ? c0 : p.tap_kidx[tap] * p.Cin_s + c0;
    const int klimit = p.fold == 1 ? p.fold_taps * p.Cin_s - c0 : p.Cin_s - c0;   // valid columns of this tile
    // element (i, r, j) of this lane -> offset in the packed matrix, or -1 outside the valid rows / columns
    auto elem_off = [&](int i, int r, int j) -> long {
        int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        int koff = kbase;
        bool row_ok = n < p.Cout;
        if (fold_n) {   // row n' = tap*Cout_s + n
            const int ltap = n / p.Cout_s;
            n -= ltap * p.Cout_s;
            row_ok = ltap < p.fold_taps && n < p.Cout;
            koff = ltap * p.Cin_s + c0;
        }
        const int c = wc * 64 + j * 32 + fi;
        return (row_ok && c < klimit) ? (long)n * Kp + koff + c : -1L;
    };
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const long off = elem_off(i, r, j);
                if (off < 0) continue;
                float* dst = out + off;
                if (combine)   // published for the block that will sum this tile's partials inside this launch
                    __hip_atomic_store(dst, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else
                    *dst = p.accumulate ? *dst + acc[i][j][r] : acc[i][j][r];
            }
    if (!combine) return;
    // Arrival ticket of the (tap, n tile, c tile): the loader waves have left (the hardware barrier counts the live
    // waves only); the flag lives behind the LDS ring, which their last DMA may still be filling.
    const int tiles = p.ntaps * p.ntiles * p.ctiles;
    int* flag = reinterpret_cast<int*>(smem + kWgRing * kWgStage);
    if (!last_arriver(p.tickets + (blockIdx.x - split * tiles), p.splits, flag)) return;
    // The last arriver sums the tile's partials in split order -- the order launch_wgrad_reduce uses -- its own included
    // (read back like the others: the accumulators are dead after the publish).  The 128 x 128 tile is walked as 4096
    // float4 pieces, 16 per thread of the four live waves; sc1 (L1-bypassing) 16-byte buffer loads.
    float* fin = p.dw_final + p.tap_woff[tap];
    const int tw = threadIdx.x;         // 0..255: the MFMA waves
    int off[16];
    float4 sum[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int idx = q * 256 + tw, rr = idx >> 5, c4 = (idx & 31) * 4;
        const int n = n0 + rr;
        off[q] = (n < p.Cout && c4 < klimit) ? n * Kp + kbase + c4 : -1;
        sum[q] = (p.accumulate && off[q] >= 0) ? *reinterpret_cast<const float4*>(fin + off[q]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int k = 0; k < p.splits; ++k) {
        const float* pk = p.dw + (size_t)k * p.dw_floats + p.tap_woff[tap];
        const __amdgpu_buffer_rsrc_t srd =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pk), 0, (int)((p.dw_floats - p.tap_woff[tap]) * 4), 0x00020000);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (off[q] < 0) continue;
            const auto raw = __builtin_amdgcn_raw_buffer_load_b128(srd, off[q] * 4, 0, /*sc1*/ 16);
            float4 v;
            __builtin_memcpy(&v, &raw, 16);
            sum[q].x += v.x;
            sum[q].y += v.y;
            sum[q].z += v.z;
            sum[q].w += v.w;
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q)
        if (off[q] >= 0) *reinterpret_cast<float4*>(fin + off[q]) = sum[q];
}

// ---- the Winograd-domain weight gradient on a fixed grid --------------------------------------------------------------------
// dU[xi][n][c] = sum_t M_dy[xi][t][n] * V[xi][t][c] for the 36 transform positions: 36 x (Cout/128) x (Cin/128) output tiles,
// each a reduction over the Tt tile rows of all frames (512 for two 512x512 frames = 32 stages of 16 rows).  One block per
// tile (conv_wgrad_kernel with 36 "taps") is 2304 short blocks = 4.5 rounds of the 512 resident ones, each paying its own
// pipeline fill and 64 KiB epilogue.  Here, as in conv_igemm.hip's wino_gemm_sk_kernel: a grid of the resident blocks, every
// block an equal run of the tile-major (tile, stage) list; a tile cut between two blocks is FINISHED by the second one from
// the first one's accumulators (head first, tail last: nobody waits for a block dispatched after it), so every dU element is
// the same t-ordered MFMA chain as with one block per tile -- bit-identical.  Both operands are plain row-major matrices:
// the loaders' per-lane offsets are fixed for the whole block and a stage is one scalar offset (no address VALU per stage),
// and they fetch the next tile's first stages while the MFMA waves store the previous tile.
struct WinoWgradSkParams {
    const float* v;               // [36][Tt][Cin]
    const float* m;               // [36][Tt][Cout]
    float* du;                    // [36][Cout_p][Kp]
    float* partial;               // [grid][4 waves][64 regs][64 lanes]
    unsigned long long* flags;    // [grid][4 waves]
    unsigned long long tag;
    unsigned* err;                // sticky error word raised by a hand-over that timed out
    int Tt, Cin, Cout, Cout_p, Kp;
    int ntiles, ctiles, nk, tiles, tiles_per_xcd, blocks_per_xcd;
    int rounds;   // > 0: whole rounds of one tile per block + half a round cut in two (wino_gemm_sk_kernel's second schedule)
    int half_round;   // ... 0: whole rounds only (tiles a multiple of the grid: 2304 tiles on one block per CU = 9 rounds)
};
typedef unsigned int wg_v4u __attribute__((__vector_size__(16)));

template <int PIX, int RING>
__global__ __launch_bounds__(512) void wino_wgrad_sk_kernel(const WinoWgradSkParams p) {
    constexpr int kStage = 2 * PIX * 128 * 4;
    constexpr int RW = PIX / 4, NI = PIX / 8, LD = 2 * NI;
    constexpr int kOOB = 0x7fff0000;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = wave >= 4;
    const int wid = wave & 3;

    // ---- this block's run of (tile, stage) units, relative to its XCD's first tile (see wino_gemm_sk_kernel) ----
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int nk = p.nk;
    // second schedule (p.rounds > 0; 2304 tiles of a 1024 -> 1024 layer on 512 blocks = 4.5 rounds): every whole round one
    // tile per block -- an XCD's 64 blocks on the 8 x 8 tiles of ONE position, whose 2 + 2 MB of operands fit its L2 -- and
    // the half round's tiles cut in two K halves, first half FIRST by the lower half of the XCD's blocks, second half LAST
    const bool hybrid = p.rounds > 0;
    const int half = p.blocks_per_xcd >> 1;
    int t0 = 0, tf = 0, tl = 0, kf = 0, kl = nk, has_tail = 0, has_head = 0, nf, nmid;
    if (hybrid) {
        has_head = p.half_round && jb < half;
        has_tail = p.half_round && !has_head;
        nmid = p.rounds;
        nf = p.rounds + (p.half_round ? 1 : 0);
    } else {
        t0 = xcd * p.tiles_per_xcd;
        const int t1 = min(t0 + p.tiles_per_xcd, p.tiles);
        if (t0 >= t1) return;
        const int ux = (t1 - t0) * nk;
        const int S = max(nk, (ux + p.blocks_per_xcd - 1) / p.blocks_per_xcd);
        const int u0 = jb * S, u1 = min(u0 + S, ux);
        if (u0 >= u1) return;
        tf = u0 / nk;
        tl = (u1 - 1) / nk;
        kf = u0 - tf * nk;
        kl = u1 - tl * nk;
        has_tail = kf > 0;
        has_head = kl < nk;
        nf = tl - tf + 1;
        nmid = nf - has_head - has_tail;
    }
    const int grid = p.blocks_per_xcd * 8;

    // loader lanes: an instruction moves 2 tile rows x 512 B; lane -> (row of the pair, 16-byte chunk of the channel slice)
    const int prow = lane >> 5, chunk = lane & 31;
    int vm[NI], vx[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = wid * RW + i * 2 + prow;
        vm[i] = (r * p.Cout + chunk * 4) * 4;
        vx[i] = (r * p.Cin + chunk * 4) * 4;
    }
    // MFMA waves: 2 x 2 waves, each 64 (n) x 64 (c)
    const int wn = wid >> 1, wc = wid & 1;
    const int fi = lane & 31, kk = lane >> 5;
    bool flag_due = false;
    for (int f = 0; f < nf; ++f) {
        int tile, kb, ke;
        if (hybrid) {
            if ((has_head && f == 0) || (has_tail && f == nf - 1)) {
                tile = p.rounds * grid + xcd * half + (has_head ? jb : jb - half);
                kb = has_head ? 0 : nk >> 1;
                ke = has_head ? nk >> 1 : nk;
            } else {
                tile = (f - has_head) * grid + xcd * p.blocks_per_xcd + jb;
                kb = 0;
                ke = nk;
            }
        } else {
            int tr;
            if (has_head && f == 0) tr = tl;
            else if (f - has_head < nmid) tr = tf + has_tail + (f - has_head);
            else tr = tf;
            kb = tr == tf ? kf : 0;
            ke = tr == tl ? kl : nk;
            tile = t0 + tr;
        }
        const bool init = kb > 0, publish = ke < nk;
        const int tpp = p.ntiles * p.ctiles;
        const int xi = tile / tpp, rem = tile - xi * tpp;
        const int nt = rem / p.ctiles, ct = rem - nt * p.ctiles;   // c tile fastest: neighbours share the M_dy tile
        const int n0 = nt * 128, c0 = ct * 128;

        if (is_loader) {
            const float* mbase = p.m + ((long)xi * p.Tt + (long)kb * PIX) * p.Cout + n0;
            const float* xbase = p.v + ((long)xi * p.Tt + (long)kb * PIX) * p.Cin + c0;
            const bool n_ok = n0 + chunk * 4 < p.Cout, c_ok = c0 + chunk * 4 < p.Cin;
            auto issue_stage = [&](int kt, int slot) {
                char* sY = smem + slot * kStage + wid * RW * 512;
                char* sX = sY + PIX * 512;
                const int so_m = (kt - kb) * PIX * p.Cout * 4, so_x = (kt - kb) * PIX * p.Cin * 4;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    wg_dma16(mbase, kOOB, sY + i * 2 * 512, n_ok ? vm[i] : kOOB, so_m);
                    wg_dma16(xbase, kOOB, sX + i * 2 * 512, c_ok ? vx[i] : kOOB, so_x);
                }
            };
            loader_k_loop<RING, LD>(kb, ke, issue_stage);
            __syncthreads();   // the MFMA waves have read their last operands: the next tile's prologue may refill the ring
            continue;
        }

        // ---- MFMA waves ----
        f32x16 acc[2][2];
        if (init) {
            const int src = blockIdx.x - 8 * (hybrid ? half : 1);
            const unsigned long long* fl = p.flags + src * 4 + wid;
            // (bounded, 1 s; on a time-out the tile is poisoned with NaNs and p.err is raised: see wino_gemm_sk_kernel)
            const bool timed_out = handover_wait(fl, p.tag, p.err, lane);
            // taken: clear it, so that a replay of this very launch (a captured graph re-issues the same tag) starts clean
            if (lane == 0) __hip_atomic_store(const_cast<unsigned long long*>(fl), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.partial) + (size_t)(src * 4 + wid) * 4096, 0, 16384, 0x00020000);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        auto raw = __builtin_amdgcn_raw_buffer_load_b128(srd, lane * 16, ((i * 2 + j) * 4 + q) * 1024, /*sc1*/ 16);
                        float v[4];
                        __builtin_memcpy(v, &raw, 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][q * 4 + e] = timed_out ? __builtin_nanf("") : v[e];
                    }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        if (flag_due) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0)
                __hip_atomic_store(p.flags + blockIdx.x * 4 + wid, p.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            flag_due = false;
        }
        wgrad_mfma_loop<PIX, RING>(smem, ke - kb, wn, wc, fi, kk, acc);
        __syncthreads();   // pairs with the loaders' closing barrier

        if (publish) {
            const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
                p.partial + (size_t)(blockIdx.x * 4 + wid) * 4096, 0, 16384, 0x00020000);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][q * 4 + e];
                        wg_v4u raw;
                        __builtin_memcpy(&raw, v, 16);
                        __builtin_amdgcn_raw_buffer_store_b128(raw, srd, lane * 16, ((i * 2 + j) * 4 + q) * 1024, /*sc1*/ 16);
                    }
            flag_due = true;
        } else {
            // D[row n][col c] -> dU[xi][n][c]: one SRD on the tile's first element, one per-lane offset, scalar row steps
            const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
                p.du + ((size_t)xi * p.Cout_p + n0) * p.Kp + c0, 0, 0x7ffffffc, 0x00020000);
            const int voff = ((wn * 64 + 4 * kk) * p.Kp + wc * 64 + fi) * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i * 32 + (r & 3) + 8 * (r >> 2);
                    const bool row_ok = n0 + wn * 64 + 4 * kk + row < p.Cout;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float v = acc[i][j][r];
                        if (row_ok && c0 + wc * 64 + j * 32 + fi < p.Cin)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), srd, voff + j * 32 * 4, row * p.Kp * 4, 0);
                    }
                }
        }
    }
    if (flag_due) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(p.flags + blockIdx.x * 4 + wid, p.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

bool wino_wgrad_sk_ok(int Tt, int Cin, int Cout) {
    const int mode = fixed_grid_enabled() ? options().wgrad_sk : 0;
    if (mode == 0 || Tt % 16 || Cin % 4 || Cout % 4) return false;
    if (mode == 2) return true;   // T2V_WGRAD_SK=2: also with fewer tiles than blocks (the small-shape tests)
    const long tiles = 36L * ((Cout + 127) / 128) * ((Cin + 127) / 128);
    return tiles >= wino_gemm_sk_grid_blocks();
}

int launch_wino_wgrad_sk(hipStream_t s, const float* V, const float* Md, float* dU, float* scratch, int Tt, int Cin, int Cout,
                         int Cout_p, int Kp) {
    T2V_REQUIRE(wino_wgrad_sk_ok(Tt, Cin, Cout), "fixed-grid weight gradient: shape not supported");
    WinoWgradSkParams k;
    k.v = V; k.m = Md; k.du = dU;
    k.partial = scratch;
    k.flags = reinterpret_cast<unsigned long long*>(scratch + wino_gemm_sk_scratch_floats() - 1024 * 4 * 2);
    k.tag = wino_gemm_sk_next_tag();
    k.err = async_error_word();
    k.Tt = Tt; k.Cin = Cin; k.Cout = Cout; k.Cout_p = Cout_p; k.Kp = Kp;
    k.ntiles = (Cout + 127) / 128; k.ctiles = (Cin + 127) / 128; k.nk = Tt / 16;
    k.tiles = 36 * k.ntiles * k.ctiles;
    const int grid = sk_launch_blocks();
    k.blocks_per_xcd = grid / 8;
    k.tiles_per_xcd = (k.tiles + 7) / 8;
    k.rounds = (k.tiles >= grid && 2 * (k.tiles % grid) == grid && k.nk % 2 == 0 && grid % 16 == 0) ? k.tiles / grid : 0;
    k.half_round = 1;
    // whole rounds only (the 2304 tiles of a 1024 -> 1024 layer on ONE block per CU: 9 rounds): the same round-by-round walk --
    // an XCD's blocks on neighbouring tiles of one position at a time.  On contiguous runs of 9 tiles per block the 32 resident
    // blocks of an XCD sat on 32 tiles of five positions that share no operand: 1.18 GB of fabric reads per launch against
    // 0.24 GB (PMC, profiles/pmc/r06_wino_wgrad_sk_one_per_cu.txt), 3.8 TB/s taken from the kernels the form exists to
    // let in beside it
    if (!k.rounds && k.tiles >= grid && k.tiles % grid == 0) k.rounds = k.tiles / grid, k.half_round = 0;
    auto kern = wino_wgrad_sk_kernel<16, 4>;
    constexpr int lds = 4 * 2 * 16 * 128 * 4;
    static bool attr_done = false;
    if (!attr_done) {
        T2V_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, sk_exclusive_lds(lds)));
        attr_done = true;
    }
    // one block per CU (the caller's two streams both carry GEMMs): the block also asks for more than half a CU's LDS, so that
    // the OTHER stream's GEMM block cannot move in beside it -- two GEMMs on a CU share the MFMA pipe without gain and leave
    // the bandwidth-bound kernels no room (today the kernel's 131 registers have the same effect)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), grid < wino_gemm_sk_grid_blocks() ? sk_exclusive_lds(lds) : lds, s, k);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, int splits, long n, float* __restrict__ dw,
                                    int accumulate) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float s = accumulate ? dw[i] : 0.f;
        for (int k = 0; k < splits; ++k) s += partial[(size_t)k * n + i];   // fixed order: deterministic
        dw[i] = s;
    }
}
int launch_wgrad_reduce(hipStream_t s, const float* partial, int splits, long n, float* dw, int accumulate) {
    long g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)g), dim3(256), 0, s, partial, splits, n, dw, accumulate);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

template <bool REFLECT, int PIX, int RING>
static int launch_wgrad_variant(hipStream_t s, const WgradParams& p) {
    auto kern = conv_wgrad_kernel<REFLECT, PIX, RING>;
    constexpr int lds = RING * 2 * PIX * 128 * 4 + 16;   // + the arrival flag of the in-kernel combine
    static bool attr_done = false;   // per instantiation
    if (!attr_done) {
        T2V_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done = true;
    }
    const int nblocks = p.ntaps * p.ntiles * p.ctiles * p.splits;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(512), lds, s, p);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// 16-pixel stages on a 4-slot ring (64 KiB: two blocks per CU).  Swept in round 2: 32-pixel stages on 2 / 3
// slots, 8-pixel stages and ring depths 3 / 5 all measured slower by 3-8 %.
int launch_conv_wgrad(hipStream_t s, const WgradParams& p) {
    return p.reflect ? launch_wgrad_variant<true, 16, 4>(s, p) : launch_wgrad_variant<false, 16, 4>(s, p);
}
// (the kernel's `fast` condition at 16-pixel stages)
bool conv_wgrad_strided_ok(const WgradParams& p) { return !p.reflect && p.fold == 0 && p.Wm % 16 == 0 && p.M % 16 == 0; }

}  // namespace t2v
