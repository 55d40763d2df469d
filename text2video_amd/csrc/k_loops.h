// k_loops.h -- the loader waves' side of the wave-specialised K loops (conv_igemm.hip, conv_wgrad.hip): 4 MFMA waves + 4
// loader waves per block, an LDS ring of RING stages, one s_barrier per stage.
#pragma once
#include <hip/hip_runtime.h>

namespace t2v {

// Stage kt+AHEAD is issued into the slot that barrier(kt-1) released; before barrier(kt) the loader only waits (counted
// vmcnt) for stage kt+1, so every DMA has AHEAD stage times to land and is never on the critical path.  Raw s_barrier +
// asm waits: __syncthreads() would drain vmcnt to 0.  Stages past the end re-fetch the last stage (no branches; an empty
// range [kb, kb) fetches stage kb once and meets the MFMA waves at B0 only).  issue_stage(stage, slot) must issue exactly
// LD_PER_WAVE DMA instructions per wave and is called with non-decreasing stages.
template <int RING, int LD_PER_WAVE, class Issue>
__device__ __forceinline__ void loader_k_loop(int kb, int ke, Issue&& issue_stage) {
    constexpr int AHEAD = RING - 1;  // stages in flight beyond the one being computed
#pragma unroll
    for (int st = 0; st < AHEAD; ++st) issue_stage(max(kb, min(kb + st, ke - 1)), st);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * LD_PER_WAVE) : "memory");
    __builtin_amdgcn_s_barrier();  // B0: the first stage has landed
    int slot = AHEAD;              // slot of stage kt+AHEAD (== slot released by barrier(kt-1))
    for (int kt = kb; kt < ke; ++kt) {
        issue_stage(max(kb, min(kt + AHEAD, ke - 1)), slot);
        // stage kt+1 landed; the younger stages (RING >= 3) stay in flight across the barrier
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * LD_PER_WAVE) : "memory");
        __builtin_amdgcn_s_barrier();  // barrier(kt)
        slot = slot >= RING - 1 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// The consumer side of the fixed-grid kernels' accumulator hand-over (conv_igemm.hip: wino_gemm_sk_kernel, conv_wgrad.hip:
// wino_wgrad_sk_kernel): poll the producer wave's tag with agent-scope loads until it shows this launch's value.  The
// producer (block b - 8, or b - 8 * half) was dispatched before this block and published before anything it waits for, so
// the wait is normally over before it starts.  It is bounded by TIME -- one second of the 100 MHz constant clock -- not by
// a poll count: after that the caller poisons its tile with NaNs and the sticky error word `err` (pinned host memory,
// t2v_internal.h: async_error_word) tells the host, which reports T2V_ERR_HANDOVER from its next entry point and falls
// back to one block per tile.  Returns true on a time-out.
constexpr unsigned long long kHandoverTimeoutTicks = 100000000ull;
__device__ __forceinline__ bool handover_wait(const unsigned long long* flag, unsigned long long tag, unsigned* err, int lane) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag) return false;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        __builtin_amdgcn_s_sleep(4);
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag) return false;
        if (wall_clock64() - t0 > kHandoverTimeoutTicks) break;
    }
    if (err && lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return true;
}

}  // namespace t2v
