// Tile hand-off between workgroups INSIDE a launch (conv_chain.hip, conv_sweep.hip): per-tile progress counters, deferred publish.
//
// Protocol (tools/probes/flag_sync.hip): producer = system-coherent stores (sc0 sc1), s_waitcnt vmcnt(0), workgroup barrier, one relaxed
// agent-scope store of the counter; consumer = relaxed agent-scope loads of the 8 neighbouring tiles' counters, workgroup barrier,
// system-coherent loads.  No cache write-back / invalidate fences (they cost 4x a whole tile on this part).
#pragma once
#include "common.h"

namespace {

// Tail of the shared workspace (tnr_conv_chain_workspace_bytes), counted from its end: [sweep dispensers: CH_SWEEP_WORDS]
// [chain per-CU arrival counters: CH_CU_KEYS] [chain tile dispensers: 4] [error word: 1].
constexpr int CH_CU_KEYS = 16 * 128;   // (xcc, se, sh, cu) keys of chain_kernel's per-CU arrival counters
constexpr int CH_SWEEP_WORDS = 16;     // tnr_conv_sweep: 8 tile dispensers + the finished-workgroup count (zero between launches)

struct ChainWait {
    unsigned *progress;
    unsigned need;
    int n, ty, tx, tiles_x, tiles_y;
    unsigned *err;
    int *pend_tile;          // tile whose previous-stage output still has to be published (-1: none)
    unsigned pend_value;
    // Deferred publish: the stores of the previous stage were issued a whole MFMA phase ago; waiting for
    // them here costs nothing, whereas waiting right after the epilogue would expose the full write burst.
    __device__ __forceinline__ void drain() const {
        if (*pend_tile >= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __device__ __forceinline__ void publish() const {
        if (*pend_tile >= 0) {
            if (threadIdx.x == 0) __hip_atomic_store(progress + *pend_tile, pend_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *pend_tile = -1;
        }
    }
    __device__ __forceinline__ void operator()() const {
        if (*pend_tile >= 0) {   // a wait before chunk 2: our own previous stage must be visible first (no circular wait)
            drain();
            __syncthreads();
            publish();
        }
        const int t = threadIdx.x;
#ifdef TNR_ABL_NOWAIT          /* (ablation build: no neighbour polling; results invalid) */
        if (t < 0) {
#else
        if (t < 9 && t != 4) {
#endif
            const int yy = ty + t / 3 - 1, xx = tx + t % 3 - 1;
            if (yy >= 0 && yy < tiles_y && xx >= 0 && xx < tiles_x) {
                const unsigned *p = progress + ((size_t)n * tiles_y + yy) * tiles_x + xx;
                // Bounded by the NUMBER OF POLLS, not by a clock: a poll is an agent-scope load round trip plus the sleep (>= ~0.5 us),
                // so 2^22 of them are seconds of waiting.  The bound used to be s_memtime() - t0 > 2^32; with several queues
                // oversubscribing the chip (two sweep launches and a third kernel on three streams, tools/probes/overlap_probe.py) it
                // fired within milliseconds in 2 runs of 5 -- a wave that is context-saved and restored meanwhile does not see a
                // continuous clock -- and a spurious timeout is worse than none: the tile reads unpublished data and the latch stops
                // the run.  A poll count only advances while the wave runs.
                // Round 6: the poll count alone can still run out falsely -- when the PRODUCER workgroup is the one that is
                // context-saved for a long time, the consumer polls at full rate meanwhile -- so a fault needs BOTH bounds: 2^22 polls AND
                // >= 2^28 ticks of the constant-rate 100 MHz counter (s_memrealtime: continuous across a context save, unlike s_memtime;
                // 2.7 s of wall time, whatever the shader clock).  Minimum wall time before a fault is declared: 2.7 s.
                unsigned polls = 0;
                const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#ifdef TNR_HANDOFF_CLOCK_BOUND     /* (probe build: the clock-based bound of rounds 2-4, kept to show the test next to other queues catches it) */
                const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
                // (int) difference: robust to the counter base wrapping around
                while ((int)(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - need) < 0) {
                    __builtin_amdgcn_s_sleep(1);
#ifdef TNR_HANDOFF_CLOCK_BOUND
                    if (__builtin_amdgcn_s_memtime() - t0 > (4ull << 30)) {
#else
                    // (the real-time counter is read once per 1024 polls beyond the count bound)
                    if (++polls > (1u << 22) && (polls & 1023u) == 0u && __builtin_amdgcn_s_memrealtime() - rt0 > (1ull << 28)) {
#endif
                        *err = 1u;
                        break;
                    }
                }
            }
        }
        __syncthreads();
    }
};

}  // namespace
