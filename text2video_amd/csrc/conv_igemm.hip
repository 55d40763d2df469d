// conv_igemm.hip -- fp32 implicit-GEMM convolution for gfx950 (MI355X), hand-written for CDNA4.
//
// Replaces, for the vid2vid generator:
//   SpatialReflectionPadding_updateOutput + SpatialConvolutionMM_updateOutput  (THCUNN.h:952,664)
//   SpatialFullDilatedConvolution_updateOutput                                  (THCUNN.h:794)
//
// Design (MI355X-first, not a port of im2col+SGEMM):
//   * NHWC activations: the GEMM K index is tap*Cin + c, so an A-row segment of one LDS stage
//     (32 floats = 128 B) is one contiguous, coalesced 128-byte run of the input.
//   * Operands go HBM/L2 -> LDS with the gfx950 LDS-DMA (global_load_lds_dwordx4): no VGPR
//     round trip, no ds_write pass.  Reflection / zero padding and ragged edges are resolved in
//     the per-lane SOURCE offset of a buffer load (out-of-range lanes are zero-filled by the
//     hardware); the im2col matrix never exists.
//   * LDS image is lane-linear (DMA constraint); the 16-byte slot a lane fetches is XOR-swizzled
//     on the source side, (row>>1)&7, and un-swizzled on the ds_read_b128 side, so fragment reads
//     are bank-conflict free (cdna_hip_programming.md rule 21 / T2).
//   * Exact fp32 on the matrix cores: v_mfma_f32_32x32x2_f32 (big layers) / 16x16x4 (3-channel
//     heads).  One ds_read_b128 feeds 4 MFMAs: the k-order inside a stage is permuted
//     identically for A and B, which only reorders the fp32 summation.
//   * 128x128 block tile; 4 MFMA waves (one per SIMD, 64x64 each => 64 accumulator VGPRs) + 4 loader
//     waves; 3-slot LDS ring with counted vmcnt, one barrier per 32-deep K stage (4096 MFMA cycles).
//   * Epilogue fuses bias, the head activations (tanh | flow*20 + sigmoid) and the instance-norm
//     partial statistics (per block: mean and M2 over its pixels, two-pass in registers) so the
//     norm never re-reads the activation to compute statistics.
//   * blockIdx -> tile mapping is XCD-aware: each XCD's L2 sees a contiguous band of tiles that
//     share A rows / all B columns.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>

#include "t2v_internal.h"
#include "k_loops.h"

namespace t2v {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int v4u __attribute__((__vector_size__(16)));

template <int MF> struct Mfma;
template <> struct Mfma<32> {
    using acc_t = f32x16;
    static constexpr int NREG = 16;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    // C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    static __device__ __forceinline__ int row(int reg, int g) { return (reg & 3) + 8 * (reg >> 2) + 4 * g; }
};
template <> struct Mfma<16> {
    using acc_t = f32x4;
    static constexpr int NREG = 4;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // C/D layout: col = lane&15, row = 4*(lane>>4) + reg
    static __device__ __forceinline__ int row(int reg, int g) { return 4 * g + reg; }
};

// MF: MFMA tile edge; WAVES_M x WAVES_N waves; each wave TM x TN MFMA tiles.
template <int MF_, int WAVES_M_, int WAVES_N_, int TM_, int TN_>
struct TileCfg {
    static constexpr int MF = MF_, WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, TM = TM_, TN = TN_;
    static constexpr int BM = WAVES_M * TM * MF;
    static constexpr int BN = WAVES_N * TN * MF;
    static constexpr int NG = 64 / MF;           // lane groups along k inside one MFMA (2 or 4)
    static constexpr int RQ = 8 / NG;            // ds_read_b128 per fragment per stage (4 or 2)
    static constexpr int STAGE_BYTES = (BM + BN) * 128;

    static_assert(WAVES_M * WAVES_N == 4, "4 MFMA waves (+ 4 loader waves) per block");
    static_assert(BM % 32 == 0, "A loader: 8 rows per wave instruction, 4 waves");
};
using CfgL = TileCfg<32, 2, 2, 2, 2>;  // 128 x 128
using CfgS = TileCfg<16, 4, 1, 4, 1>;  // 256 x 16 (3-channel heads)
using CfgQ = TileCfg<32, 2, 2, 1, 1>;  // 64 x 64: finer granularity when 128x128 tiles fill the 256 CUs poorly
using CfgW = TileCfg<32, 2, 2, 3, 1>;  // 192 x 64: all tile rows of a 512x320 frame's transform position in one tile (fixed grid)
using CfgT = TileCfg<32, 1, 4, 5, 1>;  // 160 x 128: the same rows without the padding, 80 accumulator registers: one block per CU
using CfgT8 = TileCfg<32, 1, 4, 8, 1>; // 256 x 128: a 512x512 frame's 256 tile rows in one tile per position and column, 128
                                       // accumulator registers, 48 KiB stages: one block per CU as well

// One LDS-DMA instruction through buffer addressing: 64 lanes x 16 B -> 1 KiB at the wave-uniform LDS
// address `lds_dst`; source = base + voff (per lane, bytes) + soff (scalar, bytes).  Lanes with
// voff >= nbytes are out of range and write zeros.  (Device-only builtins: the host pass of hipcc
// must not see them, or it silently drops the kernel stubs.)
__device__ __forceinline__ void dma16(const float* base, int nbytes, char* lds_dst, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, nbytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
#endif
}

// ---- the MFMA waves' K loop, shared by conv_igemm_kernel and wino_gemm_sk_kernel -------------------------------------------
// One K stage = RQ groups of 4*TM*TN MFMAs, software-pipelined by hand (one wave per SIMD has nobody else to hide its
// latencies):
//   * fragments are double-buffered in registers: group q's MFMAs run while the ds_read_b128s of group q+1 are in flight
//     (LDS latency always covered by >= 1000 MFMA cycles);
//   * the DMA of stage kt+1 is issued by the loader waves right after barrier(kt-1);
//   * ONE barrier per stage, placed before the last group's MFMAs: by then every wave has read its last fragments of the
//     current slot (slot reusable) and the DMA issued >= 1 group ago has landed (vmcnt(0) folded into the barrier), so
//     group 0 of the next stage can be prefetched from the other slot under the last group's MFMAs.
// Runs `nstages` stages out of ring slots 0, 1, ... (the loaders start every run at slot 0) and begins with the B0 barrier
// (stage 0 has landed); the caller closes with its own barrier.
//
// BKN: the B stage lies in LDS as [32 k][BN columns] (B is stored [K][N] in memory: wino_gemm_sk_kernel<.., BKN = true>, the
// data gradient reading the FORWARD layer's U).  A lane then holds the two neighbouring columns b_row0, b_row0 + 1 of its
// wave's 64 (b_row0 = 64 wn + 2 (lane & 31)) and reads them with one ds_read_b64 per k: the same k of the same stage goes
// into the same MFMA as in the [N][K] form, so the accumulation chains -- and the bits -- are the same.
template <class Cfg, int RING, bool BKN = false>
__device__ __forceinline__ void mfma_k_loop(const char* smem, int nstages, int a_row0, int b_row0, int g, int fsw,
                                            typename Mfma<Cfg::MF>::acc_t (&acc)[Cfg::TM][Cfg::TN]) {
    using MM = Mfma<Cfg::MF>;
    constexpr int MF = Cfg::MF, BM = Cfg::BM;
    static_assert(Cfg::RQ % 2 == 0, "fragment register sets alternate per group");
    static_assert(!BKN || (Cfg::TN == 2 && Cfg::MF == 32), "[K][N] B stages: 64 columns per wave, two per lane");
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr int B_READS = BKN ? 4 : Cfg::TN;
    f32x4 af[2][Cfg::TM], bf[2][BKN ? 1 : Cfg::TN];
    f32x2 bk[2][BKN ? 4 : 1];
    auto load_frags = [&](int buf, int q, int set) {
        const char* sA = smem + buf * Cfg::STAGE_BYTES;
        const char* sB = sA + BM * 128;
        const int slot = ((q * Cfg::NG + g) ^ fsw) * 16;
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
            af[set][i] = *reinterpret_cast<const f32x4*>(sA + (a_row0 + i * MF) * 128 + slot);
        if constexpr (BKN) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                bk[set][e] = *reinterpret_cast<const f32x2*>(sB + ((q * Cfg::NG + g) * 4 + e) * (Cfg::BN * 4) + b_row0 * 4);
        } else {
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j)
                bf[set][j] = *reinterpret_cast<const f32x4*>(sB + (b_row0 + j * MF) * 128 + slot);
        }
    };
    __syncthreads();  // B0
    load_frags(0, 0, 0);
    int buf = 0;
    for (int kt = 0; kt < nstages; ++kt) {
        const int nbuf = buf == RING - 1 ? 0 : buf + 1;
#pragma unroll
        for (int q = 0; q < Cfg::RQ; ++q) {
            const int cur = q & 1;
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 == Cfg::RQ) {
                // barrier(kt): every wave has read its last fragments of slot `buf` (issued during the previous group)
                // -> slot released to the loaders; stage kt+1 is visible
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                load_frags(nbuf, 0, cur ^ 1);
            } else {
                load_frags(buf, q + 1, cur ^ 1);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                    for (int j = 0; j < Cfg::TN; ++j) {
                        float bv;
                        if constexpr (BKN) bv = bk[cur][e][j];
                        else bv = bf[cur][j][e];
                        acc[i][j] = MM::run(af[cur][i][e], bv, acc[i][j]);
                    }
            // issue order inside the group: MFMA, ds_read, MFMA, ds_read, ... so that every fragment read of the NEXT
            // group issues in the shadow of an executing MFMA
#pragma unroll
            for (int r = 0; r < Cfg::TM + B_READS; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * Cfg::TM * Cfg::TN - (Cfg::TM + B_READS), 0);
        }
        buf = nbuf;
    }
}

// MODE 0: Cin_s % 32 == 0 (a stage lies inside one tap; tap offsets read with scalar loads)
// MODE 1: any Cin_s % 4 == 0, regular conv (tap -> (kh,kw) by arithmetic): the 7x7 stems
// MODE 2: any Cin_s % 4 == 0, tap table looked up per lane: transposed convs of narrow test nets
// REFLECT: reflection padding (ResnetBlock 3x3, 7x7 stems/heads) vs zero padding (stride-2 and
// transposed convs) -- a template parameter so the loader has no runtime branch and each use has
// its own kernel symbol in profiles (conv_igemm_kernel<CfgL,0,true,true> == the 1024-ch ResnetBlock
// conv that is 84 % of the FLOPs).
//
// Wave specialisation: a block is 8 waves = 4 MFMA waves (one per SIMD) + 4 loader waves (one per
// SIMD).  With fp32 MFMA a single in-order wave per SIMD cannot hide its own LDS-DMA issue cost
// (measured: 71 % MFMA-busy when the MFMA waves issued their own global_load_lds); as separate
// waves the CU interleaves the loaders' address arithmetic + DMA issue with the MFMA stream, and the
// MFMA waves execute nothing but ds_read_b128 / v_mfma / one s_barrier per stage.
// RING: LDS ring slots.  3 = every DMA gets two stage times to land (one resident block per CU:
// launches of <= 256 blocks, the heads); 2 = 64 KiB so that two blocks share a CU and hide each
// other's prologue / epilogue / DMA latency (launches of many short-K blocks).
template <class Cfg, int MODE, bool STATS, bool REFLECT, int RING>
__global__ __launch_bounds__(512) void conv_igemm_kernel(const ConvKParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using MM = Mfma<Cfg::MF>;
    using acc_t = typename MM::acc_t;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, MF = Cfg::MF;
    constexpr int A_ITERS = BM / 32;                      // wave-instructions per wave for A
    constexpr int B_INSTR = BN / 8;                       // wave-instructions for B in total

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7
    const bool is_loader = wave >= 4;
    const int wid = wave & 3;  // MFMA wave id / loader share id
    // image of a batched launch (blockIdx.y; 0 and zero strides otherwise)
    const float* __restrict__ xg = p.x + (size_t)blockIdx.y * p.x_img_stride;
    float* __restrict__ yg = p.y + (size_t)blockIdx.y * p.y_img_stride;
    float* __restrict__ sg = p.stats + (size_t)blockIdx.y * p.stats_img_stride;

    // ---- block -> (phase, tile).  Blocks dispatch in blockIdx order and block b runs on XCD b%8.
    // Phases (sub-pixel phases of a transposed conv, heaviest first) follow blockIdx order, which is
    // an LPT schedule balanced over all XCDs; inside a phase the tiles are banded per XCD so that
    // each XCD's L2 sees a contiguous run of tiles sharing A rows / all B columns.
    const int tiles_per_phase = p.mtiles * p.ntiles;
    const int phase = blockIdx.x / tiles_per_phase;
    int rem;
    {
        const int nb = tiles_per_phase, b = blockIdx.x - phase * tiles_per_phase;
        const int xcd = b & 7, idx = b >> 3;
        const int q = nb >> 3, r = nb & 7;
        rem = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = rem / p.ntiles;
    const int nt = rem - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const ConvPhase ph = p.ph[phase];
    const float* __restrict__ wbase = p.w + ph.w_off + (long)(mt / p.group_mtiles) * p.group_w_stride;

    // ---- loader lane geometry: a wave instruction moves 8 rows x 128 B; lane -> (row, slot) ----
    // Buffer addressing (buffer_load_dwordx4 ... lds): address = SRD base + per-lane voffset + scalar
    // soffset.  The per-lane part (pixel, swizzled 16-B chunk) only changes when the filter tap
    // changes; the per-stage part (channel block / K position) is an SGPR.  So a K stage costs the
    // loader waves no VALU work at all in MODE 0, and lanes whose tap falls outside the image (zero
    // padding), past K or past M use an out-of-range voffset: the hardware writes zeros to LDS.
    constexpr int kOOB = 0x7fff0000;  // >= num_records of every SRD below (checked on the host)
    const int lrow = lane >> 3;   // 0..7
    const int lslot = lane & 7;   // physical 16-B slot in the row
    const int x_bytes = p.Hin * p.Win * p.Cin_s * 4;
    const int w_bytes = p.ntiles * BN * ph.Kp * 4;
    int a_by[A_ITERS], a_bx[A_ITERS];
    int a_coff[A_ITERS];  // float offset of the data chunk this lane fetches = 4*(slot ^ swizzle(row))
    bool a_ok[A_ITERS];
    int a_voff[A_ITERS];  // MODE 0: byte offset for the current tap
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
        const int r = wid * (BM / 4) + i * 8 + lrow;
        const int m = m0 + r;
        a_ok[i] = m < p.M;
        const int my = m / p.Wm, mx = m - my * p.Wm;
        a_by[i] = my * p.stride;
        a_bx[i] = mx * p.stride;
        a_coff[i] = (lslot ^ ((r >> 1) & 7)) * 4;
        a_voff[i] = kOOB;
    }
    // byte offset of pixel (by+dy, bx+dx), channel c, or kOOB
    auto pix_off = [&](int i, int dy, int dx, int c, bool ok) {
        int iy = a_by[i] + dy, ix = a_bx[i] + dx;
        if constexpr (REFLECT) {
            iy = iy < 0 ? -iy : iy;
            ix = ix < 0 ? -ix : ix;
            iy = min(iy, 2 * p.Hin - 2 - iy);
            ix = min(ix, 2 * p.Win - 2 - ix);
        } else {
            ok = ok && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
        }
        return ok ? ((iy * p.Win + ix) * p.Cin_s + c) * 4 : kOOB;
    };
    int cur_tap = -1;
    // per-tap refresh of the A voffsets (MODE 0): once every Cin_s/32 stages
    auto set_tap = [&](int tap) {
        const int dy = p.tdy[ph.tap0 + tap], dx = p.tdx[ph.tap0 + tap];
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) a_voff[i] = pix_off(i, dy, dx, a_coff[i], a_ok[i]);
        cur_tap = tap;
    };
    // A-operand DMA instruction i (of A_ITERS per wave) of K-stage kt into ring slot buf
    auto issue_a = [&](int i, int kt, int buf) {
        char* dst = smem + buf * Cfg::STAGE_BYTES + (wid * (BM / 4) + i * 8) * 128;
        if constexpr (MODE == 0) {
            const int cpt = p.Cin_s >> 5;
            const int c0 = (kt - cur_tap * cpt) << 5;  // set_tap(kt / cpt) was called for this stage
            dma16(xg, x_bytes, dst, a_voff[i], c0 * 4);
        } else {
            // general path (Cin_s = 8 / 12 stems, narrow test nets): each 16-B chunk has its own tap
            const int k = kt * kBK + a_coff[i];
            const bool kin = k < ph.ntaps * p.Cin_s;
            const int tap = kin ? k / p.Cin_s : 0;
            const int c = k - tap * p.Cin_s;
            int dy, dx;
            if constexpr (MODE == 1) {
                const int kh = tap / p.KW;
                dy = kh - p.pad;
                dx = tap - kh * p.KW - p.pad;
            } else {
                dy = p.tdy[ph.tap0 + tap];
                dx = p.tdx[ph.tap0 + tap];
            }
            dma16(xg, x_bytes, dst, pix_off(i, dy, dx, c, kin && a_ok[i]), 0);
        }
    };
    // B-operand (packed weights [Cout_p][Kp]) DMA instruction `instr` (of BN/8 per block)
    auto issue_b = [&](int instr, int kt, int buf) {
        char* dst = smem + buf * Cfg::STAGE_BYTES + BM * 128 + instr * 8 * 128;
        const int r = instr * 8 + lrow;
        const int chunk = lslot ^ ((r >> 1) & 7);
        const int voff = ((n0 + r) * ph.Kp + chunk * 4) * 4;  // loop invariant per instr
        dma16(wbase, w_bytes, dst, voff, kt * (kBK * 4));
    };
    // this wave's DMA instruction number `n` of a stage: first its A rows, then its B rows
    constexpr int B_PER_WAVE = (B_INSTR >= 4) ? B_INSTR / 4 : 1;
    constexpr int LD_PER_WAVE = A_ITERS + B_PER_WAVE;
    auto issue = [&](int n, int kt, int buf) {
        if (n < A_ITERS) {
            issue_a(n, kt, buf);
        } else if (n < LD_PER_WAVE) {
            if constexpr (B_INSTR >= 4) {
                issue_b(wid * B_PER_WAVE + (n - A_ITERS), kt, buf);
            } else {
                // fewer B instructions than loader waves (small-Cout tile): the spare waves re-issue
                // one of them (same bytes to the same LDS rows).  Every wave MUST issue exactly
                // LD_PER_WAVE loads per stage -- the counted s_waitcnt vmcnt below relies on it
                // (a wave issuing fewer would pass barrier(kt) with a load of stage kt+1 in flight).
                issue_b(wid % B_INSTR, kt, buf);
            }
        }
    };
    // all DMA instructions of stage kt (refreshing the tap offsets when the stage enters a new tap)
    auto issue_stage = [&](int kt, int buf) {
        if constexpr (MODE == 0) {
            const int tap = kt / (p.Cin_s >> 5);
            if (tap != cur_tap) set_tap(tap);
        }
#pragma unroll
        for (int n = 0; n < LD_PER_WAVE; ++n) issue(n, kt, buf);
    };

    // ---- MFMA fragment geometry ----
    const int wm = wid / Cfg::WAVES_N, wn = wid - wm * Cfg::WAVES_N;
    const int fr = lane & (MF - 1);  // row (A) / col (B) inside the MFMA tile
    const int g = lane / MF;         // k group
    const int fsw = (fr >> 1) & 7;   // tile bases are multiples of 16 rows, so swizzle(row) = swizzle(fr)
    const int a_row0 = wm * (Cfg::TM * MF) + fr;
    const int b_row0 = wn * (Cfg::TN * MF) + fr;

    acc_t acc[Cfg::TM][Cfg::TN];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
            for (int r = 0; r < MM::NREG; ++r) acc[i][j][r] = 0.f;

    // ---- main loop (mfma_k_loop / loader_k_loop above) ----
    const int nk = ph.nk;
    if (is_loader)
        loader_k_loop<RING, LD_PER_WAVE>(0, nk, issue_stage);
    else
        mfma_k_loop<Cfg, RING>(smem, nk, a_row0, b_row0, g, fsw, acc);
    __syncthreads();  // the epilogue reuses the LDS ring
    if (is_loader) {
        // loader waves only keep the barrier count of the statistics reduction below in step
        // (s_barrier counts arrivals of all 8 waves, wherever in the code they arrive from)
        if constexpr (STATS) {
            __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads();
        }
        return;
    }

    // ---- epilogue ----
    float bias_v[Cfg::TN];
    int col[Cfg::TN];
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
        col[j] = n0 + wn * (Cfg::TN * MF) + j * MF + fr;
        bias_v[j] = (col[j] < p.Cout && p.bias) ? p.bias[col[j]] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
            for (int r = 0; r < MM::NREG; ++r) acc[i][j][r] += bias_v[j];

    if constexpr (STATS) {
        // per-block, per-channel mean and M2 over the block's valid pixels (two passes over the
        // accumulators held in registers); combined across blocks by inorm_finalize (Chan).
        static_assert(!STATS || Cfg::WAVES_M == 2, "stats reduction written for 2 waves along M");
        float* red = reinterpret_cast<float*>(smem);  // [2][BN]; main loop ended with a barrier
        const int cnt = min(BM, p.M - m0);
        const float inv_cnt = 1.f / (float)cnt;
        float mean_b[Cfg::TN];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) {
                // pairwise (tree) summation: log-depth rounding error, and exact for a constant
                // map (power-of-two counts of equal values) -- the all-zero previous-frame input
                // of a sequence's first frame makes every channel of the image encoder constant
                float tsum[Cfg::TM * MM::NREG];
#pragma unroll
                for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                    for (int r = 0; r < MM::NREG; ++r) {
                        const int row = wm * (Cfg::TM * MF) + i * MF + MM::row(r, g);
                        float v = acc[i][j][r];
                        if (pass == 1) {
                            v -= mean_b[j];
                            v *= v;
                        }
                        tsum[i * MM::NREG + r] = (row < cnt) ? v : 0.f;
                    }
#pragma unroll
                for (int w = Cfg::TM * MM::NREG / 2; w >= 1; w >>= 1)
#pragma unroll
                    for (int i = 0; i < w; ++i) tsum[i] += tsum[i + w];
                float s = tsum[0];
                s += __shfl_xor(s, 32);
                if (g == 0) red[wm * BN + wn * (Cfg::TN * MF) + j * MF + fr] = s;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) {
                const int c = wn * (Cfg::TN * MF) + j * MF + fr;
                const float tot = red[c] + red[BN + c];
                if (pass == 0) {
                    mean_b[j] = tot * inv_cnt;
                } else if (wm == 0 && g == 0 && col[j] < p.Cout) {
                    const size_t part = (size_t)phase * p.mtiles + mt;
                    float2* dst = reinterpret_cast<float2*>(sg) + part * p.Cout + col[j];
                    *dst = make_float2(mean_b[j], tot);
                }
            }
            __syncthreads();
        }
    }

    // ---- store.  Round 4: the PMC passes of the stride-2 / transposed layers showed 2.4-3.4 vector instructions per MFMA,
    // most of them HERE -- a 32-bit integer division per accumulator row and lane (m -> (row, column) of the GEMM pixel
    // grid), 64-bit address arithmetic and three activation compares per element: ~3000 instructions per wave and tile,
    // 8 % of a 36-stage block and up to a third of the transposed convs' 8-stage phase blocks.  Now: m / Wm by one
    // multiply-high (exact for m < 2^20 and Wm < 2^20), buffer stores off one SRD (out-of-range lanes are dropped by the
    // hardware), the activation chosen once per tile.
    const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(yg, 0, p.Hout * p.Wout * p.Cout_s * 4, 0x00020000);
    // >= the record count of any output map (checked on the host: < 2 GiB); offsets are unsigned, so a pixel offset plus
    // kDrop (a padding column of a live pixel) stays a defined value >= kDrop and is dropped as well
    constexpr unsigned kDrop = 0x7fffff00u;
    const bool fastdiv = p.M <= (1 << 20);
    const unsigned long long magic = (1ull << 40) / (unsigned)p.Wm + 1;
    unsigned coff[Cfg::TN];
    bool creal[Cfg::TN];
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
        coff[j] = col[j] < p.Cout_s ? (unsigned)col[j] * 4u : kDrop;
        creal[j] = col[j] < p.Cout;
    }
    auto store_tile = [&](auto act) {
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
            for (int r = 0; r < MM::NREG; ++r) {
                const int row = wm * (Cfg::TM * MF) + i * MF + MM::row(r, g);
                const int m = m0 + row;
                const int my = fastdiv ? (int)(((unsigned long long)(unsigned)m * magic) >> 40) : m / p.Wm;
                const int mx = m - my * p.Wm;
                const int oy = my * p.ostride + ph.oy0, ox = mx * p.ostride + ph.ox0;
                // (oy / ox past the output: the ragged phase grid of an odd-sized transposed output)
                const bool ok = m < p.M && oy < p.Hout && ox < p.Wout;
                const unsigned poff = (unsigned)((oy * p.Wout + ox) * p.Cout_s) * 4u;
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j) {
                    const float v = creal[j] ? act(acc[i][j][r], j) : 0.f;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ysrd, (int)(ok ? poff + coff[j] : kDrop), 0, 0);
                }
            }
    };
    if (p.act == T2V_ACT_NONE) {
        store_tile([](float v, int) { return v; });
    } else if (p.act == T2V_ACT_TANH) {
        store_tile([](float v, int) { return tanhf(v); });
    } else if (p.act == T2V_ACT_FLOW_W) {
        const float sc = p.act_scale;
        store_tile([&](float v, int j) { return col[j] < 2 ? v * sc : 1.f / (1.f + expf(-v)); });
    } else {
        const float sc = p.act_scale;
        store_tile([&](float v, int) { return v > 0.f ? v : v * sc; });
    }
}

// ---- the batched Winograd GEMM on a fixed grid ("stream-K") ------------------------------------------------------------
// conv_igemm_kernel gives every 128x128 tile its own block: the 576 tiles of a 512x512 frame's F(4x4) GEMM stage
// (36 positions x 2 x 8) are 1.125 rounds of the 512 resident blocks (two per CU), and the 2304 64x64 tiles used instead
// pay twice the LDS-DMA traffic per FLOP.  Here the grid IS the 512 resident blocks: the (tile, K stage) units are laid
// out tile-major and every block takes an equal run of them -- 36 of 18432 stages for that frame, i.e. the tail of one
// tile, (whole tiles,) and the head of the next.  The head goes FIRST: its accumulators are written (write-through) to
// the block's slot of `partial` and a per-wave tag is raised; the block that owns the rest of that tile gets to it LAST
// and starts its K loop from those accumulators instead of zeros.  Every output is therefore the same K-ordered chain
// of MFMAs as in conv_igemm_kernel (bit-identical), nobody waits on a block that started after it (the producer of a
// block's hand-over is block b - 8, dispatched earlier), and whole tiles never touch the scratch.
// Blocks b, b + 8, ... (one XCD under round-robin dispatch) share a run of whole tiles, m tile fastest: the two blocks
// on one U column tile run side by side in one L2.
struct SkKParams {
    const float* a;
    const float* b;
    float* c;
    float* partial;               // [grid][4 waves][64 regs][64 lanes]
    unsigned long long* flags;    // [grid][4 waves]
    unsigned long long tag;       // unique per launch: stale flags of earlier launches never match
    unsigned* err;                // sticky error word raised by a hand-over that timed out
    long a_group_stride;
    int T, K, N, c_cs;
    int mtiles_g, ntiles, nk, tiles, tiles_per_xcd, blocks_per_xcd;
    int rounds;   // > 0: `rounds` whole rounds of one tile per block + half a round of tiles cut in two (see the kernel)
};

// BKN: B is [groups][K][N] (N contiguous) instead of [groups][N][K]: the data gradient of a Winograd layer contracts over the
// forward layer's OUTPUT channels, i.e. over the rows of the forward U [36][Cout][Cin] -- read in place, no transposed copy
// of the weights (36 x 45 us of winograd4_weight_adjoint_kernel and 5.4 GB per train step of a 1024-channel generator).
// A loader instruction fetches two k rows of 128 columns (1 KiB); see mfma_k_loop for the MFMA side.
template <class Cfg, int RING, bool BKN = false>
__global__ __launch_bounds__(512) void wino_gemm_sk_kernel(const SkKParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using MM = Mfma<32>;
    using acc_t = typename MM::acc_t;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, MF = 32;
    constexpr int A_ITERS = BM / 32, B_PER_WAVE = BN / 32, LD_PER_WAVE = A_ITERS + B_PER_WAVE;
    constexpr int PW = Cfg::TM * Cfg::TN * 1024 > 4096 ? Cfg::TM * Cfg::TN * 1024 : 4096;   // floats of one wave's hand-over slot

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = wave >= 4;
    const int wid = wave & 3;

    // ---- this block's run of units (relative to its XCD's first tile) ----
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int nk = p.nk;
    // Second schedule (p.rounds > 0), for tile counts of R whole rounds plus exactly half a round (2304 tiles on 512 blocks:
    // four 512x512 images, or a 1024x1024 frame): the contiguous runs above would put an XCD's 64 blocks into 64 tiles of
    // 4-5 different positions at once (L2 hit rate 26 %); here every round is one tile per block -- an XCD's blocks on 64
    // neighbouring tiles, as with one block per tile -- and the half round's tiles are cut in two: the lower half of an XCD's
    // blocks computes the first K half of one such tile FIRST and hands it over, the upper half finishes it LAST.
    const bool hybrid = p.rounds > 0;
    const int half = p.blocks_per_xcd >> 1;
    int t0 = 0, tf = 0, tl = 0, kf = 0, kl = nk, has_tail = 0, has_head = 0, nf, nmid;
    if (hybrid) {
        has_head = j < half;
        has_tail = !has_head;
        nmid = p.rounds;
        nf = p.rounds + 1;
    } else {
        t0 = xcd * p.tiles_per_xcd;
        const int t1 = min(t0 + p.tiles_per_xcd, p.tiles);
        if (t0 >= t1) return;
        const int ux = (t1 - t0) * nk;
        const int S = max(nk, (ux + p.blocks_per_xcd - 1) / p.blocks_per_xcd);   // >= one tile: a run never lies inside a tile
        const int u0 = j * S, u1 = min(u0 + S, ux);
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

    // ---- loader lane geometry (the same for every tile: only the SRD bases move) ----
    static_assert(!BKN || BN == 128, "[K][N] B: 128-column tiles (a 512-byte LDS row per k)");
    const int lrow = lane >> 3, lslot = lane & 7;
    const int a_bytes = BM * p.K * 4, b_bytes = BKN ? ((p.K - 1) * p.N + BN) * 4 : BN * p.K * 4;
    const int b_stage_step = BKN ? kBK * p.N * 4 : kBK * 4;   // bytes from one K stage to the next
    int a_voff[A_ITERS], b_voff[B_PER_WAVE];
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
        const int r = wid * (BM / 4) + i * 8 + lrow;
        a_voff[i] = (r * p.K + (lslot ^ ((r >> 1) & 7)) * 4) * 4;
    }
#pragma unroll
    for (int i = 0; i < B_PER_WAVE; ++i) {
        if constexpr (BKN) {
            const int kr = (wid * B_PER_WAVE + i) * 2 + (lane >> 5);   // k row of the stage; LDS row kr = [128 columns]
            b_voff[i] = (kr * p.N + (lane & 31) * 4) * 4;
        } else {
            const int r = (wid * B_PER_WAVE + i) * 8 + lrow;
            b_voff[i] = (r * p.K + (lslot ^ ((r >> 1) & 7)) * 4) * 4;
        }
    }

    // ---- MFMA fragment geometry ----
    const int wm = wid / Cfg::WAVES_N, wn = wid - wm * Cfg::WAVES_N;
    const int fr = lane & (MF - 1);
    const int g = lane / MF;
    const int fsw = (fr >> 1) & 7;
    const int a_row0 = wm * (Cfg::TM * MF) + fr;
    const int b_row0 = wn * (Cfg::TN * MF) + (BKN ? 2 * fr : fr);   // BKN: the first of this lane's two columns
    bool flag_due = false;   // this wave's hand-over stores are in flight; its tag is raised at the next wait point
    for (int f = 0; f < nf; ++f) {
        // processing order: the head handed on to block j+1, the whole tiles, the tail begun by block j-1
        int tile, kb, ke;
        if (hybrid) {
            if ((has_head && f == 0) || (has_tail && f == nf - 1)) {   // half of a tile of the half round
                tile = p.rounds * grid + xcd * half + (has_head ? j : j - half);
                kb = has_head ? 0 : nk >> 1;
                ke = has_head ? nk >> 1 : nk;
            } else {                                                   // round f - has_head: the XCD's band of that round
                tile = (f - has_head) * grid + xcd * p.blocks_per_xcd + j;
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
        const int tpg = p.mtiles_g * p.ntiles;
        const int pg = tile / tpg, rem = tile - pg * tpg;
        const int nt = rem / p.mtiles_g, mt = rem - nt * p.mtiles_g;

        if (is_loader) {
            const float* abase = p.a + pg * p.a_group_stride + (long)mt * BM * p.K;
            const float* bbase = BKN ? p.b + (long)pg * p.K * p.N + nt * BN : p.b + ((long)pg * p.N + nt * BN) * p.K;
            auto issue_stage = [&](int kt, int buf) {
                char* dstA = smem + buf * Cfg::STAGE_BYTES + wid * (BM / 4) * 128;
                char* dstB = smem + buf * Cfg::STAGE_BYTES + BM * 128 + wid * B_PER_WAVE * 8 * 128;
#pragma unroll
                for (int i = 0; i < A_ITERS; ++i) dma16(abase, a_bytes, dstA + i * 8 * 128, a_voff[i], kt * (kBK * 4));
#pragma unroll
                for (int i = 0; i < B_PER_WAVE; ++i) dma16(bbase, b_bytes, dstB + i * 8 * 128, b_voff[i], kt * b_stage_step);
            };
            loader_k_loop<RING, LD_PER_WAVE>(kb, ke, issue_stage);
            __syncthreads();   // MFMA waves have read their last fragments: the next tile's prologue may overwrite the ring
            continue;
        }

        // ---- MFMA waves ----
        acc_t acc[Cfg::TM][Cfg::TN];
        if (init) {
            // the accumulators an earlier block of this XCD left for this tile (block j-1 = blockIdx - 8; block j - half in
            // the second schedule); its wave `wid` wrote what this wave reads
            const int src = blockIdx.x - 8 * (hybrid ? half : 1);
            const unsigned long long* fl = p.flags + src * 4 + wid;
            // (bounded: 1 s.  The producer ran before this block was even dispatched; if its tag is still missing something
            // is broken: the tile is then poisoned with NaNs and the host is told through p.err -- no hung GPU, no silent frame)
            const bool timed_out = handover_wait(fl, p.tag, p.err, lane);
            // taken: clear it, so that a replay of this very launch (a captured graph re-issues the same tag) starts clean
            if (lane == 0) __hip_atomic_store(const_cast<unsigned long long*>(fl), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // write-through (sc1) stores on the producer's side, sc1 loads here: the pair that is coherent across the
            // XCDs' L2s inside one launch; 16 x 16 bytes per lane at scalar offsets off one SRD
            const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.partial) + (size_t)(src * 4 + wid) * PW, 0, PW * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int jj = 0; jj < Cfg::TN; ++jj)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        auto raw = __builtin_amdgcn_raw_buffer_load_b128(srd, lane * 16, ((i * Cfg::TN + jj) * 4 + q) * 1024, 16);
                        float v[4];
                        __builtin_memcpy(v, &raw, 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][jj][q * 4 + e] = timed_out ? __builtin_nanf("") : v[e];
                    }
        } else {
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int jj = 0; jj < Cfg::TN; ++jj)
#pragma unroll
                    for (int r = 0; r < MM::NREG; ++r) acc[i][jj][r] = 0.f;
        }
        if (flag_due) {
            // (the head went first: its stores drain while the loaders fetch this tile's first stage)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the write-through stores have left for memory
            if (lane == 0)
                __hip_atomic_store(p.flags + blockIdx.x * 4 + wid, p.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            flag_due = false;
        }
        mfma_k_loop<Cfg, RING, BKN>(smem, ke - kb, a_row0, b_row0, g, fsw, acc);
        __syncthreads();   // pairs with the loaders' closing barrier

        if (publish) {
            const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
                p.partial + (size_t)(blockIdx.x * 4 + wid) * PW, 0, PW * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int jj = 0; jj < Cfg::TN; ++jj)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][jj][q * 4 + e];
                        v4u raw;
                        __builtin_memcpy(&raw, v, 16);
                        __builtin_amdgcn_raw_buffer_store_b128(raw, srd, lane * 16, ((i * Cfg::TN + jj) * 4 + q) * 1024, 16);
                    }
            flag_due = true;
        } else {
            // one SRD on the tile's first output row, one per-lane offset (its row group and column), the rest scalar
            const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
                p.c + ((size_t)pg * p.T + (size_t)mt * BM) * p.c_cs + nt * BN, 0, 0x7ffffffc, 0x00020000);
            // (BKN: MFMA column tile jj of this lane is the tile's column b_row0 + jj)
            const int voff = ((wm * (Cfg::TM * MF) + 4 * g) * p.c_cs + (BKN ? b_row0 : wn * (Cfg::TN * MF) + fr)) * 4;
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int r = 0; r < MM::NREG; ++r) {
                    const int soff = (i * MF + (r & 3) + 8 * (r >> 2)) * p.c_cs * 4;
#pragma unroll
                    for (int jj = 0; jj < Cfg::TN; ++jj) {
                        const float v = acc[i][jj][r];
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), srd, voff + jj * (BKN ? 4 : MF * 4), soff, 0);
                    }
                }
        }
    }
    if (flag_due) {   // a run that was nothing but a head
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(p.flags + blockIdx.x * 4 + wid, p.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- the same GEMM for tile rows that are no whole 128s: ragged M tiles --------------------------------------------------
// The reference's own frames (512x680: 352 Winograd tiles per transform position; two 512x320 sequences in lock-step: 320;
// the 16:9 speakers: 464, 224) pad badly to 128-row tiles (384, 384, 512, 256: 8-17 % of the MFMA work on zero rows).  Here
// the rows of a position are cut into 32-row MFMA fragments, F = ceil(rows / 32), and every (position, 128-column) group into
// M tiles of 4, 4, ..., 4, r fragments (r = F mod 4): the full tiles run exactly as in wino_gemm_sk_kernel (2 x 2 waves of
// 64 x 64), the ragged one as 1 x 4 waves of (32 r) x 32 -- every wave busy on every k step, no zero rows.  A unit is one K
// stage of one fragment row; the unit list (group-major, M tile fastest, as above) is cut into equal WEIGHT runs per block, a
// run boundary rounded to the nearest stage of the tile it falls into (both neighbours compute the same boundary).  A run is
// at least one full tile long, so a tile is cut at most once: head first / tail last, the accumulator hand-over and its
// guards are those of wino_gemm_sk_kernel, and every output is still the same K-ordered MFMA chain (bit-identical to the
// other forms of the stage: which wave owns an element changes, its sum does not).
struct SkrKParams {
    const float* a;
    const float* b;
    float* c;
    float* partial;
    unsigned long long* flags;
    unsigned long long tag;
    unsigned* err;
    long a_group_stride;
    int Tp;                       // rows per position of a / c (the padded pitch)
    int K, N, c_cs;
    int F;                        // 32-row fragments per position that hold real rows
    int ntiles, nk, cgs;          // 128-column tiles, K stages, column groups (positions x ntiles)
    int cg_per_xcd, blocks_per_xcd;
    int S;                        // run length per block, in units (>= one full tile)
    int mt_count, mt_base, mt_extra;   // wino_gemm_skt_kernel: M tiles per column group, the first mt_extra of them mt_base + 1
                                       // fragments tall, the rest mt_base
};
using CfgR1 = TileCfg<32, 1, 4, 1, 1>;   //  32 x 128
using CfgR2 = TileCfg<32, 1, 4, 2, 1>;   //  64 x 128
using CfgR3 = TileCfg<32, 1, 4, 3, 1>;   //  96 x 128

// one piece [kb, ke) of one tile: the loader waves stream its stages, the MFMA waves start from zeros or from the handed-over
// accumulators (init), and end by publishing them (ke < nk) or by storing the finished tile
template <class Cfg, int RING, int PW = 4096>      // PW: floats of one wave's hand-over slot (>= Cfg::TM * Cfg::TN * 1024)
__device__ __forceinline__ void skr_piece(const SkrKParams& p, char* smem, int wid, int lane, bool is_loader, int pg, int nt,
                                          int row0, int kb, int ke, bool& flag_due) {
    static_assert(PW >= Cfg::TM * Cfg::TN * 1024, "hand-over slot too small for this tile");
    using MM = Mfma<32>;
    using acc_t = typename MM::acc_t;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, MF = 32;
    constexpr int A_ITERS = BM / 32, B_PER_WAVE = BN / 32, LD_PER_WAVE = A_ITERS + B_PER_WAVE;
    const bool init = kb > 0, publish = ke < p.nk;
    const int lrow = lane >> 3, lslot = lane & 7;
    if (is_loader) {
        const float* abase = p.a + pg * p.a_group_stride + (long)row0 * p.K;
        const float* bbase = p.b + ((long)pg * p.N + nt * BN) * p.K;
        const int a_bytes = BM * p.K * 4, b_bytes = BN * p.K * 4;
        int a_voff[A_ITERS], b_voff[B_PER_WAVE];
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            const int r = wid * (BM / 4) + i * 8 + lrow;
            a_voff[i] = (r * p.K + (lslot ^ ((r >> 1) & 7)) * 4) * 4;
        }
#pragma unroll
        for (int i = 0; i < B_PER_WAVE; ++i) {
            const int r = (wid * B_PER_WAVE + i) * 8 + lrow;
            b_voff[i] = (r * p.K + (lslot ^ ((r >> 1) & 7)) * 4) * 4;
        }
        auto issue_stage = [&](int kt, int buf) {
            char* dstA = smem + buf * Cfg::STAGE_BYTES + wid * (BM / 4) * 128;
            char* dstB = smem + buf * Cfg::STAGE_BYTES + BM * 128 + wid * B_PER_WAVE * 8 * 128;
#pragma unroll
            for (int i = 0; i < A_ITERS; ++i) dma16(abase, a_bytes, dstA + i * 8 * 128, a_voff[i], kt * (kBK * 4));
#pragma unroll
            for (int i = 0; i < B_PER_WAVE; ++i) dma16(bbase, b_bytes, dstB + i * 8 * 128, b_voff[i], kt * (kBK * 4));
        };
        loader_k_loop<RING, LD_PER_WAVE>(kb, ke, issue_stage);
        __syncthreads();   // the MFMA waves have read their last fragments: the next piece's prologue may overwrite the ring
        return;
    }
    const int wm = wid / Cfg::WAVES_N, wn = wid - wm * Cfg::WAVES_N;
    const int fr = lane & (MF - 1);
    const int g = lane / MF;
    const int fsw = (fr >> 1) & 7;
    const int a_row0 = wm * (Cfg::TM * MF) + fr;
    const int b_row0 = wn * (Cfg::TN * MF) + fr;
    acc_t acc[Cfg::TM][Cfg::TN];
    if (init) {
        const int src = blockIdx.x - 8;
        const unsigned long long* fl = p.flags + src * 4 + wid;
        const bool timed_out = handover_wait(fl, p.tag, p.err, lane);
        if (lane == 0) __hip_atomic_store(const_cast<unsigned long long*>(fl), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.partial) + (size_t)(src * 4 + wid) * PW, 0, PW * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
            for (int jj = 0; jj < Cfg::TN; ++jj)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    auto raw = __builtin_amdgcn_raw_buffer_load_b128(srd, lane * 16, ((i * Cfg::TN + jj) * 4 + q) * 1024, 16);
                    float v[4];
                    __builtin_memcpy(v, &raw, 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][jj][q * 4 + e] = timed_out ? __builtin_nanf("") : v[e];
                }
    } else {
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
            for (int jj = 0; jj < Cfg::TN; ++jj)
#pragma unroll
                for (int r = 0; r < MM::NREG; ++r) acc[i][jj][r] = 0.f;
    }
    if (flag_due) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(p.flags + blockIdx.x * 4 + wid, p.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag_due = false;
    }
    mfma_k_loop<Cfg, RING>(smem, ke - kb, a_row0, b_row0, g, fsw, acc);
    __syncthreads();   // pairs with the loaders' closing barrier
    if (publish) {
        // (the wave -> element mapping of the hand-over is the tile's own: producer and consumer run the same Cfg)
        const __amdgpu_buffer_rsrc_t srd =
            __builtin_amdgcn_make_buffer_rsrc(p.partial + (size_t)(blockIdx.x * 4 + wid) * PW, 0, PW * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
            for (int jj = 0; jj < Cfg::TN; ++jj)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][jj][q * 4 + e];
                    v4u raw;
                    __builtin_memcpy(&raw, v, 16);
                    __builtin_amdgcn_raw_buffer_store_b128(raw, srd, lane * 16, ((i * Cfg::TN + jj) * 4 + q) * 1024, 16);
                }
        flag_due = true;
    } else {
        const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
            p.c + ((size_t)pg * p.Tp + row0) * p.c_cs + nt * BN, 0, 0x7ffffffc, 0x00020000);
        const int voff = ((wm * (Cfg::TM * MF) + 4 * g) * p.c_cs + wn * (Cfg::TN * MF) + fr) * 4;
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
            for (int r = 0; r < MM::NREG; ++r) {
                const int soff = (i * MF + (r & 3) + 8 * (r >> 2)) * p.c_cs * 4;
#pragma unroll
                for (int jj = 0; jj < Cfg::TN; ++jj)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][jj][r]), srd, voff + jj * MF * 4, soff, 0);
            }
    }
}

template <int RING>
__global__ __launch_bounds__(512, 4) void wino_gemm_skr_kernel(const SkrKParams p) {   // (4 waves per SIMD: two blocks per CU)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = wave >= 4;
    const int wid = wave & 3;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int nk = p.nk, F = p.F, nfull = F >> 2, rem = F & 3, MT = nfull + (rem ? 1 : 0);
    const int cg0 = xcd * p.cg_per_xcd, cg1 = min(cg0 + p.cg_per_xcd, p.cgs);
    if (cg0 >= cg1) return;
    const int cgw = F * nk, Wx = (cg1 - cg0) * cgw;
    const int u0 = min(j * p.S, Wx), u1 = min(u0 + p.S, Wx);
    if (u0 >= u1) return;
    // unit offset inside the XCD's range -> (linear tile index = group * MT + M tile, stage), the stage rounded to the nearest
    // stage boundary of that tile; a boundary at a tile's end is the start of the next tile
    auto locate = [&](int u, int& lt, int& st) {
        const int c = u / cgw, r = u - c * cgw;
        int mt;
        if (r < nfull * 4 * nk) {
            mt = r / (4 * nk);
            st = (r - mt * 4 * nk + 2) >> 2;
        } else {
            mt = nfull;
            st = rem ? (r - nfull * 4 * nk + (rem >> 1)) / rem : 0;
        }
        lt = c * MT + mt;
        if (st >= nk) {
            st = 0;
            ++lt;
        }
    };
    int lt0, k0, lt1, k1;
    locate(u0, lt0, k0);
    locate(u1, lt1, k1);
    const bool has_tail = k0 > 0, has_head = k1 > 0;
    const int first_whole = lt0 + (has_tail ? 1 : 0);
    const int nwhole = max(0, lt1 - first_whole);
    const int nf = nwhole + (has_head ? 1 : 0) + (has_tail ? 1 : 0);
    bool flag_due = false;
    for (int f = 0; f < nf; ++f) {
        int lt, kb = 0, ke = nk;
        if (has_head && f == 0) {
            lt = lt1;
            ke = k1;
        } else if (f - (has_head ? 1 : 0) < nwhole) {
            lt = first_whole + f - (has_head ? 1 : 0);
        } else {
            lt = lt0;
            kb = k0;
        }
        const int c = lt / MT, mt = lt - c * MT;
        const int cg = cg0 + c;
        const int pg = cg / p.ntiles, nt = cg - pg * p.ntiles;
        const int row0 = mt * 128;
        const int frs = mt < nfull ? 4 : rem;
        if (frs == 4) skr_piece<CfgL, RING>(p, smem, wid, lane, is_loader, pg, nt, row0, kb, ke, flag_due);
        else if (frs == 3) skr_piece<CfgR3, RING>(p, smem, wid, lane, is_loader, pg, nt, row0, kb, ke, flag_due);
        else if (frs == 2) skr_piece<CfgR2, RING>(p, smem, wid, lane, is_loader, pg, nt, row0, kb, ke, flag_due);
        else skr_piece<CfgR1, RING>(p, smem, wid, lane, is_loader, pg, nt, row0, kb, ke, flag_due);
    }
    if (flag_due && !is_loader) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(p.flags + blockIdx.x * 4 + wid, p.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- the ragged form with ONE block per CU: balanced tall tiles ---------------------------------------------------------
// §4.6's second lesson applied to the first: with one resident block per CU (three-slot ring, up to 256 VGPRs) a tile may be
// up to 6 fragments = 192 rows tall, so the rows of a position are cut into ceil(F / 6) tiles of nearly equal height
// (352 rows = 11 fragments: 6 + 5; 464 -> 15: 5 + 5 + 5; 224 -> 7: 4 + 3; 688 -> 22: 6 + 6 + 5 + 5), every one of them run
// as 1 x 4 waves of (32 tm) x 32.  Less LDS-DMA per MAC than 128-row tiles, and the second stream of a frame finds free
// wave slots on every CU.  Units, weights, run boundaries and the hand-over are wino_gemm_skr_kernel's.
using CfgR4 = TileCfg<32, 1, 4, 4, 1>;   // 128 x 128 as 1 x 4 waves
using CfgR6 = TileCfg<32, 1, 4, 6, 1>;   // 192 x 128
constexpr int kSktPW = 6144;              // hand-over slot of a wave: 6 fragments

template <int RING>
__global__ __launch_bounds__(512) void wino_gemm_skt_kernel(const SkrKParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = wave >= 4;
    const int wid = wave & 3;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int nk = p.nk, F = p.F, MT = p.mt_count, base = p.mt_base, extra = p.mt_extra;
    const int cg0 = xcd * p.cg_per_xcd, cg1 = min(cg0 + p.cg_per_xcd, p.cgs);
    if (cg0 >= cg1) return;
    const int cgw = F * nk, Wx = (cg1 - cg0) * cgw;
    const int u0 = min(j * p.S, Wx), u1 = min(u0 + p.S, Wx);
    if (u0 >= u1) return;
    const int big_w = (base + 1) * nk, small_w = base * nk, big_all = extra * big_w;
    auto locate = [&](int u, int& lt, int& st) {
        const int c = u / cgw, r = u - c * cgw;
        int mt, tm, w0;
        if (r < big_all) {
            mt = r / big_w;
            tm = base + 1;
            w0 = mt * big_w;
        } else {
            mt = extra + (r - big_all) / small_w;
            tm = base;
            w0 = big_all + (mt - extra) * small_w;
        }
        st = (r - w0 + (tm >> 1)) / tm;
        lt = c * MT + mt;
        if (st >= nk) {
            st = 0;
            ++lt;
        }
    };
    int lt0, k0, lt1, k1;
    locate(u0, lt0, k0);
    locate(u1, lt1, k1);
    const bool has_tail = k0 > 0, has_head = k1 > 0;
    const int first_whole = lt0 + (has_tail ? 1 : 0);
    const int nwhole = max(0, lt1 - first_whole);
    const int nf = nwhole + (has_head ? 1 : 0) + (has_tail ? 1 : 0);
    bool flag_due = false;
    for (int f = 0; f < nf; ++f) {
        int lt, kb = 0, ke = nk;
        if (has_head && f == 0) {
            lt = lt1;
            ke = k1;
        } else if (f - (has_head ? 1 : 0) < nwhole) {
            lt = first_whole + f - (has_head ? 1 : 0);
        } else {
            lt = lt0;
            kb = k0;
        }
        const int c = lt / MT, mt = lt - c * MT;
        const int cg = cg0 + c;
        const int pg = cg / p.ntiles, nt = cg - pg * p.ntiles;
        const int row0 = 32 * (mt * base + min(mt, extra));
        const int frs = mt < extra ? base + 1 : base;
        if (frs == 6) skr_piece<CfgR6, RING, kSktPW>(p, smem, wid, lane, is_loader, pg, nt, row0, kb, ke, flag_due);
        else if (frs == 5) skr_piece<CfgT, RING, kSktPW>(p, smem, wid, lane, is_loader, pg, nt, row0, kb, ke, flag_due);
        else if (frs == 4) skr_piece<CfgR4, RING, kSktPW>(p, smem, wid, lane, is_loader, pg, nt, row0, kb, ke, flag_due);
        else skr_piece<CfgR3, RING, kSktPW>(p, smem, wid, lane, is_loader, pg, nt, row0, kb, ke, flag_due);
    }
    if (flag_due && !is_loader) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(p.flags + blockIdx.x * 4 + wid, p.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int wino_gemm_sk_grid_blocks() {
    static int n = 0;
    if (!n) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        n = max(8, min(1024, 2 * cus) / 8 * 8);   // two 64-KiB-ring blocks per CU, whole XCD groups
    }
    return n;
}
// the one-block-per-CU forms ask for 108-120 KiB of dynamic LDS (3-deep rings of 160/192/256-row stages): a device whose
// opt-in limit is below that (64 KiB parts) takes the tile-per-block / two-per-CU forms instead of failing the frame
static int lds_optin_bytes() {
    static int n = 0;
    if (!n) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0)
            v = 64 * 1024;
        n = v;
    }
    return n;
}
int sk_exclusive_lds(int own_bytes) { return std::max(own_bytes, std::min(kSkExclusiveLds, lds_optin_bytes())); }
constexpr int kSkMaxGrid = 1024;
size_t wino_gemm_sk_scratch_floats() { return (size_t)kSkMaxGrid * (4 * 64 * 64 + 4 * 2); }

// tile of the fixed-grid launch: 128 x 128 where the tile rows are whole 128s; 192 x 64 where they are whole 192s (a 512x320
// frame: 160 tiles padded to 192 -- all rows of a transform position in one tile); 0: no fixed-grid form
static int sk_tile_rows(int T, int N) {
    if (N % 128) return 0;            // (the packed weights hold round_up(N, 128) rows per position: N itself, then)
    if (T % 128 == 0) return 128;
    return T % 192 == 0 ? 192 : 0;
}
// R whole rounds + exactly half a round of tiles, an even number of K stages, whole 16-block halves per XCD
static bool sk_half_round(long tiles, long grid, int nk) {
    return tiles >= grid && 2 * (tiles % grid) == grid && nk % 2 == 0 && grid % 16 == 0;
}
// 129 .. 160 real rows per position (the 64 x 40 bottleneck of the reference's 512x320 frames: 160 Winograd tiles) -- and,
// under the overlap hint (a caller with a second stream: the generator's two-stream frames; t2v_set_overlap_hint),
// the 256 rows of a 512x512 frame (512 of two) on 256 x 128 tiles -- : ONE
// 160 x 128 tile per position and column tile -- no padding rows, 0.0141 B of LDS-DMA per MAC (192 x 64: 0.0208) -- whose 80
// accumulator registers and 108 KiB ring allow one block per CU: a fixed grid of one block per CU, 288 tiles on 256 blocks
static int sk_tall_rows(int groups, int rows, int T, int N) {     // tile rows of the one-block-per-CU form (160 | 256), or 0
    if (rows <= 0 || N % 128) return 0;
    int bm = 0, mt = 1;
    if (rows > 128 && rows <= 160 && T >= 160) bm = 160;
    // 512 rows (two 512x512 images in lock-step).  One image's 256 rows took this form too until round 6: with the transforms'
    // XCD slices (winograd.hip) the frame is no faster for it any more (10.25 vs 10.23 ms with the flow branch, 8.05 vs 7.96
    // without; two sequences still gain 2.5 %: 10.03 vs 10.28 ms per step) -- T2V_OVERLAP_HINT_SINGLE=1 brings it back
    else if (overlap_hint() && T % 256 == 0 && rows > T - 32 && T <= 512 && (T == 512 || options().overlap_hint_single))
        bm = 256, mt = T / 256;
    if (!bm || lds_optin_bytes() < 3 * (bm == 160 ? CfgT::STAGE_BYTES : CfgT8::STAGE_BYTES)) return 0;
    const long tiles = (long)groups * mt * (N / 128), grid = wino_gemm_sk_grid_blocks() / 2;
    return (tiles >= grid && tiles * 100 <= ((tiles + grid - 1) / grid) * grid * 90) ? bm : 0;
}
static bool sk_tall(int groups, int rows, int T, int N) { return sk_tall_rows(groups, rows, T, N) != 0; }
int wino_gemm_sk_tall_rows(int groups, int rows, int T, int N) { return rows > 0 ? sk_tall_rows(groups, rows, T, N) : 0; }
bool wino_gemm_sk_ok(int groups, int T, int K, int N, int c_cs, int rows) {
    const int mode = fixed_grid_enabled() ? options().wino_gemm_sk : 0;
    if (mode != 0 && rows > 0 && K % kBK == 0 && c_cs == N && sk_tall(groups, rows, T, N)) return true;
    const int bm = sk_tile_rows(T, N);
    if (mode == 0 || bm == 0 || K % kBK || c_cs != N || (long)192 * K * 4 >= 0x7fff0000L) return false;
    // fewer tiles than resident blocks: one tile per block is already less than one round.  Beyond that the fixed grid
    // pays where whole tiles fill their last round badly -- measured on MI355X (scripts/sk_probe.py, K = N = 1024):
    // 1.125 rounds 189 -> 144 us, 1.69 rounds 241 -> 210 us, 2.25 rounds 352 -> 294 us, 4.5 rounds 583 -> 611 us
    const long tiles = (long)groups * (T / bm) * (N / (bm == 128 ? 128 : 64)), grid = wino_gemm_sk_grid_blocks();
    if (mode == 2) return true;   // T2V_WINO_GEMM_SK=2: wherever the shape allows (fewer tiles than blocks: the
                                          // spare blocks leave at once -- what the small-shape tests run)
    if (tiles < grid) return false;
    if (sk_half_round(tiles, grid, K / kBK)) return true;     // R.5 rounds: the second schedule (whole rounds + a cut half round)
    const long rounds = (tiles + grid - 1) / grid;
    return tiles * 100 <= rounds * grid * 85;
}

// B given as [groups][K][N]: whole 128 x 128 tiles on two blocks per CU only (what a 512x512 train frame's 256 tile rows run)
bool wino_gemm_sk_bkn_ok(int groups, int T, int K, int N, int c_cs) {
    return wino_gemm_sk_ok(groups, T, K, N, c_cs, 0) && sk_tile_rows(T, N) == 128 && (long)K * N * 4 < 0x7fff0000L;
}

unsigned long long wino_gemm_sk_next_tag() {
    static std::atomic<unsigned long long> tag_counter{0};
    if (tag_counter.load() == 0) {
        unsigned long long seed = 0;
        FILE* f = fopen("/dev/urandom", "rb");
        if (!f || fread(&seed, sizeof(seed), 1, f) != 1) seed = 0x9e3779b97f4a7c15ull * (unsigned long long)(uintptr_t)&seed;
        if (f) fclose(f);
        unsigned long long zero = 0;
        tag_counter.compare_exchange_strong(zero, (seed >> 1) | 1ull);
    }
    return ++tag_counter;
}

template <class Cfg, int RING = 2, bool BKN = false>
static int launch_sk(hipStream_t s, const SkKParams& k, int grid) {
    auto kern = wino_gemm_sk_kernel<Cfg, RING, BKN>;
    constexpr int LDS_BYTES = RING * Cfg::STAGE_BYTES;
    static const int LDS_MAX = sk_exclusive_lds(LDS_BYTES);
    static bool attr_done = false;   // per instantiation
    if (!attr_done) {
        T2V_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
        attr_done = true;
    }
    // a two-per-CU form launched on ONE block per CU (sk_launch_blocks under overlap hint 2) keeps its CU's second GEMM slot
    // shut as well (conv_wgrad.hip: launch_wino_wgrad_sk)
    const bool exclusive = RING == 2 && grid < wino_gemm_sk_grid_blocks();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), exclusive ? LDS_MAX : LDS_BYTES, s, k);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

int launch_wino_gemm_sk(hipStream_t s, const SkGemm& g) {
    T2V_REQUIRE(wino_gemm_sk_ok(g.groups, g.T, g.K, g.N, g.c_cs, g.rows), "stream-K gemm: shape not supported");
    SkKParams k;
    k.a = g.a; k.b = g.b; k.c = g.c;
    k.partial = g.scratch;
    k.flags = reinterpret_cast<unsigned long long*>(g.scratch + (size_t)kSkMaxGrid * 4 * 64 * 64);
    k.tag = wino_gemm_sk_next_tag();
    k.err = g.err;
    k.a_group_stride = g.a_group_stride;
    k.T = g.T; k.K = g.K; k.N = g.N; k.c_cs = g.c_cs;
    if (g.b_kn) {
        T2V_REQUIRE(wino_gemm_sk_bkn_ok(g.groups, g.T, g.K, g.N, g.c_cs), "stream-K gemm, B as [K][N]: shape not supported");
    } else if (const int tall = g.rows > 0 ? sk_tall_rows(g.groups, g.rows, g.T, g.N) : 0) {
        k.mtiles_g = tall == 160 ? 1 : g.T / 256; k.ntiles = g.N / 128; k.nk = g.K / kBK;
        k.tiles = g.groups * k.mtiles_g * k.ntiles;
        const int grid = wino_gemm_sk_grid_blocks() / 2;       // one block per CU
        k.blocks_per_xcd = grid / 8;
        k.tiles_per_xcd = (k.tiles + 7) / 8;
        k.rounds = 0;
        return tall == 160 ? launch_sk<CfgT, 3>(s, k, grid) : launch_sk<CfgT8, 3>(s, k, grid);
    }
    const int bm = sk_tile_rows(g.T, g.N), bn = bm == 128 ? 128 : 64;
    k.mtiles_g = g.T / bm; k.ntiles = g.N / bn; k.nk = g.K / kBK;
    k.tiles = g.groups * k.mtiles_g * k.ntiles;
    const int grid = sk_launch_blocks();      // (one block per CU while the caller's second stream runs fixed-grid GEMMs of its own)
    k.blocks_per_xcd = grid / 8;
    k.tiles_per_xcd = (k.tiles + 7) / 8;
    k.rounds = sk_half_round(k.tiles, grid, k.nk) ? (int)(k.tiles / grid) : 0;
    if (g.b_kn) return launch_sk<CfgL, 2, true>(s, k, grid);
    return bm == 128 ? launch_sk<CfgL>(s, k, grid) : launch_sk<CfgW>(s, k, grid);
}

// ---- ragged form: when, and the launch ----
// `rows` real tile rows per position (all images of a batch, packed), `Tp` the padded pitch.  Taken when the rows are no whole
// 128s and the run per block -- in (fragment row x stage) units, at least one full tile -- is shorter than what the padded
// alternatives give a block: the fixed grid on whole 128 / 192-row tiles where that applies, else one block per tile.
static long skr_run_units(int groups, int rows, int K, int N, int* cg_per_xcd) {
    const int F = (rows + 31) / 32, nk = K / kBK, cgs = groups * (N / 128), bpx = wino_gemm_sk_grid_blocks() / 8;
    const int cgx = (cgs + 7) / 8;
    if (cg_per_xcd) *cg_per_xcd = cgx;
    const long Wx = (long)cgx * F * nk;
    return std::max<long>(4L * nk, (Wx + bpx - 1) / bpx);
}
static bool skr_shape_ok(int groups, int rows, int Tp, int K, int N, int c_cs) {
    const int mode = fixed_grid_enabled() ? options().wino_gemm_sk : 0;
    return !(mode == 0 || !options().wino_gemm_sk_ragged || rows < 1 || N % 128 || K % kBK || c_cs != N ||
             (long)128 * K * 4 >= 0x7fff0000L || (rows + 31) / 32 * 32 > Tp ||
             (long)groups * (N / 128) * ((rows + 31) / 32) * (K / kBK) >= 0x7fffffffL / 2);
}
bool wino_gemm_skr_ok(int groups, int rows, int Tp, int K, int N, int c_cs) {
    if (!skr_shape_ok(groups, rows, Tp, K, N, c_cs)) return false;
    const int mode = options().wino_gemm_sk;
    const int F = (rows + 31) / 32, nk = K / kBK;
    if (F % 4 == 0) return false;                       // whole 128-row tiles: wino_gemm_sk_kernel's case
    if (mode == 2) return true;
    const long grid = wino_gemm_sk_grid_blocks();
    const long run = skr_run_units(groups, rows, K, N, nullptr);        // fragment rows x stages per block, 2 blocks per CU
    // the alternatives, in the same units per CU-resident pair of blocks
    // The alternatives, in the same units per CU-resident pair of blocks, each with its measured cost per unit relative to the
    // whole-tile fixed grid (MI355X, scripts/sk_probe.py): a ragged run costs ~8 % more per unit (the 1 x 4 tail tiles read 4/3
    // fragments per MFMA and the per-stage barrier does not shrink with the tile), one block per 128x128 tile ~3 %, one block
    // per 64x64 tile ~12 % (bound by the LDS-DMA stream).  464 rows = 15 fragments lose against 16 padded ones;
    // 352 / 224 / 688 rows win 3-5 % against their padded fixed grid, 320 rows 9 % against 64x64 tiles.
    long alt, alt_cost;
    if (wino_gemm_sk_ok(groups, Tp, K, N, c_cs)) {
        const int bm = Tp % 128 == 0 ? 128 : 192, bn = bm == 128 ? 128 : 64;
        const long tiles = (long)groups * (Tp / bm) * (N / bn);
        const long per_tile = (long)(bm / 32) * nk * bn / 128;           // in units of a 128-column fragment row x stage
        alt = std::max(per_tile, (tiles * per_tile + grid - 1) / grid);
        alt_cost = 100;
    } else {
        // one block per 64x64 or 128x128 tile, whole rounds of the resident blocks
        const bool big = Tp % 128 == 0;
        const long tiles = big ? (long)groups * (Tp / 128) * (N / 128) : (long)groups * (Tp / 64) * (N / 64);
        const long resident = big ? grid : 2 * grid, per_tile = big ? 4L * nk : nk;
        alt = (tiles + resident - 1) / resident * per_tile * (big ? 1 : 2);     // (64x64: four blocks per CU = two per "slot")
        alt_cost = big ? 103 : 112;
    }
    return run * 108 <= alt * alt_cost;
}

// the one-block-per-CU form of the ragged kernel (T2V_WINO_GEMM_SK_RAGGED=2): balanced tiles of 3..6 fragments
static bool skt_split(int rows, int* mt_count, int* base, int* extra) {
    const int F = (rows + 31) / 32, MT = (F + 5) / 6;
    if (F < 6 || F / MT < 3) return false;
    *mt_count = MT; *base = F / MT; *extra = F % MT;
    return true;
}
bool wino_gemm_skt_ok(int groups, int rows, int Tp, int K, int N, int c_cs) {
    int mt, base, extra;
    if (options().wino_gemm_sk_ragged < 2 || !skt_split(rows, &mt, &base, &extra)) return false;
    if (lds_optin_bytes() < 3 * CfgR6::STAGE_BYTES) return false;
    const long cgs = (long)groups * (N / 128), grid = wino_gemm_sk_grid_blocks() / 2;
    if (cgs * mt < grid) return false;                                              // (at least one tile per block)
    return wino_gemm_skr_ok(groups, rows, Tp, K, N, c_cs);
}
int launch_wino_gemm_skt(hipStream_t s, const SkGemm& g, int rows) {
    T2V_REQUIRE(wino_gemm_skt_ok(g.groups, rows, g.T, g.K, g.N, g.c_cs), "tall ragged fixed-grid gemm: shape not supported");
    SkrKParams k;
    k.a = g.a; k.b = g.b; k.c = g.c;
    k.partial = g.scratch;
    k.flags = reinterpret_cast<unsigned long long*>(g.scratch + (size_t)kSkMaxGrid * 4 * 64 * 64);
    k.tag = wino_gemm_sk_next_tag();
    k.err = g.err;
    k.a_group_stride = g.a_group_stride;
    k.Tp = g.T; k.K = g.K; k.N = g.N; k.c_cs = g.c_cs;
    k.F = (rows + 31) / 32;
    k.ntiles = g.N / 128; k.nk = g.K / kBK; k.cgs = g.groups * k.ntiles;
    skt_split(rows, &k.mt_count, &k.mt_base, &k.mt_extra);
    const int grid = wino_gemm_sk_grid_blocks() / 2;            // one block per CU
    k.blocks_per_xcd = grid / 8;
    k.cg_per_xcd = (k.cgs + 7) / 8;
    const long Wx = (long)k.cg_per_xcd * k.F * k.nk;
    k.S = (int)std::max<long>((long)(k.mt_base + (k.mt_extra ? 1 : 0)) * k.nk, (Wx + k.blocks_per_xcd - 1) / k.blocks_per_xcd);
    auto kern = wino_gemm_skt_kernel<3>;
    constexpr int LDS_BYTES = 3 * CfgR6::STAGE_BYTES;
    static bool attr_done = false;
    if (!attr_done) {
        T2V_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_BYTES, s, k);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

int launch_wino_gemm_skr(hipStream_t s, const SkGemm& g, int rows) {
    T2V_REQUIRE(wino_gemm_skr_ok(g.groups, rows, g.T, g.K, g.N, g.c_cs), "ragged fixed-grid gemm: shape not supported");
    SkrKParams k;
    k.a = g.a; k.b = g.b; k.c = g.c;
    k.partial = g.scratch;
    k.flags = reinterpret_cast<unsigned long long*>(g.scratch + (size_t)kSkMaxGrid * 4 * 64 * 64);
    k.tag = wino_gemm_sk_next_tag();
    k.err = g.err;
    k.a_group_stride = g.a_group_stride;
    k.Tp = g.T; k.K = g.K; k.N = g.N; k.c_cs = g.c_cs;
    k.F = (rows + 31) / 32;
    k.ntiles = g.N / 128; k.nk = g.K / kBK; k.cgs = g.groups * k.ntiles;
    const int grid = wino_gemm_sk_grid_blocks();
    k.blocks_per_xcd = grid / 8;
    k.S = (int)skr_run_units(g.groups, rows, g.K, g.N, &k.cg_per_xcd);
    auto kern = wino_gemm_skr_kernel<2>;
    constexpr int LDS_BYTES = 2 * CfgL::STAGE_BYTES;
    static bool attr_done = false;
    if (!attr_done) {
        T2V_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_BYTES, s, k);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

int conv_tile_for(int Cout) { return Cout <= 16 ? kTileS : kTileL; }
void conv_tile_dims(int tile, int* BM, int* BN) {
    if (tile == kTileS) {
        *BM = CfgS::BM;
        *BN = CfgS::BN;
    } else if (tile == kTileQ) {
        *BM = CfgQ::BM;
        *BN = CfgQ::BN;
    } else {
        *BM = CfgL::BM;
        *BN = CfgL::BN;
    }
}

template <class Cfg, int MODE, bool STATS, bool REFLECT, int RING>
static int launch_ring(hipStream_t s, const ConvKParams& p) {
    auto kern = conv_igemm_kernel<Cfg, MODE, STATS, REFLECT, RING>;
    constexpr int LDS_BYTES = RING * Cfg::STAGE_BYTES;
    static bool attr_done = false;  // per instantiation
    if (!attr_done) {
        T2V_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_done = true;
    }
    const int nblocks = p.mtiles * p.ntiles * p.nphases;
    hipLaunchKernelGGL(kern, dim3(nblocks, p.batch > 1 ? p.batch : 1), dim3(512), LDS_BYTES, s, p);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

template <class Cfg, int MODE, bool STATS, bool REFLECT>
static int launch_pad(hipStream_t s, const ConvKParams& p) {
    // >= 4 blocks per CU queued: let two blocks share the CU (64 KiB ring each); measured on MI355X:
    // 2048-block stem 0.40 vs 0.47 ms, 1024-block layers 0.33 vs 0.345 ms, <= 512 blocks favour RING 3
    const long nblocks = (long)p.mtiles * p.ntiles * p.nphases * (p.batch > 1 ? p.batch : 1);
    // (swept per layer shape with scripts/kernel_bench.py: single-phase launches like two co-resident blocks from 512
    // blocks on; 64x64 tiles of a multi-phase launch prefer the deeper ring.  Round 4: also between 257 and 511 blocks -- one
    // block per CU would need a second, mostly empty round there: the 344-block stride-2 layers of a 512x680 frame, 456 at
    // 512x912; frames 16.07 -> 15.88 ms)
    const bool two = Cfg::MF == 32 && (Cfg::BM == 64 ? (p.nphases == 1 || nblocks >= 4096)
                                                     : (nblocks >= 1024 || (p.nphases == 1 && nblocks > 256)));
    return two ? launch_ring<Cfg, MODE, STATS, REFLECT, 2>(s, p) : launch_ring<Cfg, MODE, STATS, REFLECT, 3>(s, p);
}

template <class Cfg, int MODE, bool STATS>
static int launch_one(hipStream_t s, const ConvKParams& p) {
    return p.pad_mode == T2V_PAD_REFLECT ? launch_pad<Cfg, MODE, STATS, true>(s, p)
                                         : launch_pad<Cfg, MODE, STATS, false>(s, p);
}

int launch_conv_igemm(hipStream_t s, const ConvKParams& p, int tile) {
    const int mode = (p.Cin_s % kBK) == 0 ? 0 : (p.nphases == 1 ? 1 : 2);
    const bool stats = p.stats != nullptr;
    if (tile == kTileL) {
        switch (mode) {
            case 0: return stats ? launch_one<CfgL, 0, true>(s, p) : launch_one<CfgL, 0, false>(s, p);
            case 1: return stats ? launch_one<CfgL, 1, true>(s, p) : launch_one<CfgL, 1, false>(s, p);
            default: return stats ? launch_one<CfgL, 2, true>(s, p) : launch_one<CfgL, 2, false>(s, p);
        }
    }
    if (tile == kTileQ) {
        switch (mode) {
            case 0: return stats ? launch_one<CfgQ, 0, true>(s, p) : launch_one<CfgQ, 0, false>(s, p);
            case 1: return stats ? launch_one<CfgQ, 1, true>(s, p) : launch_one<CfgQ, 1, false>(s, p);
            default: return stats ? launch_one<CfgQ, 2, true>(s, p) : launch_one<CfgQ, 2, false>(s, p);
        }
    }
    T2V_REQUIRE(!stats, "small-Cout tile has no instance-norm statistics epilogue");
    switch (mode) {
        case 0: return launch_one<CfgS, 0, false>(s, p);
        case 1: return launch_one<CfgS, 1, false>(s, p);
        default: return launch_one<CfgS, 2, false>(s, p);
    }
}

}  // namespace t2v
