// Several dependent 3x3 convolutions over the same pixel grid in ONE launch ("chain"): the five growing-K
// convolutions of a residual dense block (RRDBNet_arch.py:150-163) and the five of its gradient mirror.
//
// Why: a trunk layer is a single residency round (512 tiles = 2 workgroups x 256 CUs at batch 16).  As a
// launch of its own, every workgroup runs its prologue (first-tile fetch: a 37 MB HBM read burst) and its
// epilogue (a 33 MB write burst) at the same time as all others, with the matrix pipes idle -- 17 % of the
// layer (tools/probes/conv_timeline.hip).  In a chain a workgroup keeps its pixel tile through all stages:
// the stores of stage s drain while stage s+1 already runs on the channels that existed before, and only
// the chunk that first touches stage-s output waits -- for the 8 neighbouring tiles (3x3 halo), through a
// per-tile progress counter.
//
// Hand-off protocol (tools/probes/flag_sync.hip): producer = system-coherent stores (sc0 sc1), s_waitcnt
// vmcnt(0), workgroup barrier, one relaxed agent-scope store of the counter; consumer = relaxed
// agent-scope loads of the 8 counters, workgroup barrier, system-coherent loads.  No cache write-back /
// invalidate fences (they cost 4x a whole tile on this part).
// Deadlock freedom: the grid never exceeds the co-resident capacity and every workgroup walks its tiles
// stage-major, so a waited-for tile always belongs to a running workgroup at an earlier program point.
#include <stddef.h>
#include "conv_body.h"
#include "conv_epilogue.h"
#include "conv_handoff.h"
#ifdef TNR_CONV_DL_EXPERIMENT     /* tools/build_variant.py dl -DTNR_CONV_DL_EXPERIMENT: LDS-DMA staging experiment, conv_body_dl.h */
#include "conv_body_dl.h"
#include <cstdlib>
#endif

namespace {

struct ChainK {
    int nstages;
    int tiles_x, tiles_y, tiles;       // 16 x 32 pixel tiles over (N, H, W)
    unsigned *progress;                // [tiles]: base + (stages of that tile whose output is visible)
    unsigned base;
    unsigned *err;                     // set to 1 when a dependency wait gives up (never in a healthy run)
    unsigned *cu_ctr;                  // [CH_CU_KEYS] arrivals per CU, never reset: parity = which of the CU's two slots
    unsigned *tile_ctr;                // [2 sets][2 populations] tile dispensers; set (epoch & 1) is live, the other gets zeroed
    int dyn;                           // 1: one tile per workgroup, dealt at run time by CU slot (phase-shifted populations)
    int set;
    int wait_chunk[TNR_CHAIN_MAX];
    ConvK st[TNR_CHAIN_MAX];
};

// Stage s of the kernel-argument table, fetched with scalar loads: indexing the by-value struct with a
// run-time s would make the compiler copy all of it to scratch and turn every field into a VGPR.
__device__ __forceinline__ ConvK chain_stage(int s, int *wait_chunk) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) ChainK karg_chain;
    karg_chain *ka = (karg_chain *)__builtin_amdgcn_kernarg_segment_ptr();   // constant address space: s_load
    *wait_chunk = ka->wait_chunk[s];
    return ka->st[s];
#else
    *wait_chunk = -1;
    return ConvK();
#endif
}

template <int BF, int DL = 0>
__global__ void __launch_bounds__(256, 2) conv_chain_kernel(const ChainK c) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int pend_tile = -1;          // wave-uniform
    unsigned pend_value = 0;
    int calls = 0;               // (timeline probe only)
    (void)calls;
    // Phase-shifted populations (c.dyn: tiles == grid == 2 workgroups per CU, i.e. the benchmark's trunk layers): all
    // workgroups reaching a stage boundary together leave the matrix pipes idle while the tile stores drain.  The two
    // workgroups of a CU are therefore put into different halves of the batch (dependencies never cross images) and the
    // second half starts TNR_CHAIN_STAGGER cycles late, so that one workgroup's boundary overlaps the other's MFMA phases.
    // Which slot of its CU a workgroup got is the parity of a per-CU arrival counter; tiles are dealt by two dispensers
    // (a workgroup whose half is exhausted takes from the other one, so the mapping stays a bijection whatever the parity).
    int first_tile = blockIdx.x, tile_step = gridDim.x;
#ifndef TNR_CHAIN_STAGGER
#define TNR_CHAIN_STAGGER 0
#endif
    if (TNR_CHAIN_STAGGER > 0 && c.dyn) {
        __shared__ int s_tile, s_slot;
        if (threadIdx.x == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
            const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) & 15u;   // HW_REG_XCC_ID
            const unsigned key = (xcc << 7) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
            const unsigned slot = atomicAdd(c.cu_ctr + key, 1u) & 1u;
            const unsigned half = (unsigned)c.tiles >> 1;
            unsigned *disp = c.tile_ctr + 2 * c.set;
            unsigned pop = slot, idx = atomicAdd(disp + pop, 1u);
            if (idx >= half) {
                pop ^= 1u;
                idx = atomicAdd(disp + pop, 1u);
            }
            s_tile = (int)(pop * half + idx);
            s_slot = (int)pop;
            if (blockIdx.x == 0) {                    // the other set serves the next launch on this stream
                c.tile_ctr[2 * (c.set ^ 1)] = 0u;
                c.tile_ctr[2 * (c.set ^ 1) + 1] = 0u;
            }
        }
        __syncthreads();
        first_tile = s_tile;
        tile_step = c.tiles;
        if (s_slot) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)(TNR_CHAIN_STAGGER)) __builtin_amdgcn_s_sleep(64);
        }
    }
    for (int s = 0; s < c.nstages; ++s) {
        int wait_chunk;
        const ConvK st = chain_stage(s, &wait_chunk);
        const int ncb = st.KoutP >> 5;
        for (int tile = first_tile; tile < c.tiles; tile += tile_step) {
            int q = tile;
            const int tx = q % c.tiles_x;
            q /= c.tiles_x;
            const int ty = q % c.tiles_y;
            const int n = q / c.tiles_y;
            const ChainWait w{c.progress, c.base + (unsigned)s, n, ty, tx, c.tiles_x, c.tiles_y, c.err, &pend_tile, pend_value};
            for (int cb = 0; cb < ncb; ++cb) {
                // opaque copies: keep the per-tile index arithmetic of the body INSIDE the loops (hoisted out of
                // them it stays live across the whole kernel and spills)
                int txo = tx, tyo = ty, no = n;
                asm volatile("" : "+s"(txo), "+s"(tyo), "+s"(no));
                TNR_STAMP_CALL(calls++);
#ifdef TNR_CONV_DL_EXPERIMENT
                if constexpr (DL != 0)      // (TNR_CONV_DL=2) LDS-DMA staging, conv_body_dl.h
                    conv_tile_body_dk8<1, 4, true>(st, cb, txo, tyo, no, smem, (tnr_lds_float *)smem, cb == 0 ? wait_chunk : -1, w);
                else
#endif
                    conv_tile_body<TNR_CONV_3x3, 32, 1, 4, true, BF>(st, cb, txo, tyo, no, 0, smem, cb == 0 ? wait_chunk : -1, w);
            }
            // this tile's stage-s output is on its way to memory: published from inside the next tile body
            // (every stage has >= 2 input chunks, so the previous pending tile has been published by now)
            pend_tile = tile;
            pend_value = c.base + (unsigned)s + 1u;
        }
    }
    // nothing in this launch waits for the last stage; publish it anyway so the counters stay consistent
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (pend_tile >= 0 && threadIdx.x == 0)
        __hip_atomic_store(c.progress + pend_tile, pend_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- TNR_MMA_BF16X3, 8-wave form (see conv_x3w8.h for the idea): ONE workgroup of 8 waves per CU, tile 32 x 32 pixels x 32 output
// channels per pass, input tile 34 x 34 pixels and weight slab 9 x 32 rows both pre-split in LDS (96-byte rows of three bf16 planes:
// 138 624 B), wave w owns tile rows 4 w .. 4 w + 3 (four M-tiles).  Same hand-off protocol as conv_chain_kernel (system-coherent input
// loads and output stores, per-tile progress counters, publish deferred to chunk 1, neighbour wait in front of the first chunk that
// reads the previous stage's channels); the tile grid is 32 x 32, so a 128 x 128 x 16 batch is exactly one tile per CU.  Arithmetic
// (split, kept partial products, chunk and tap order) is that of conv_tile_body<.., BF = 2>: bit-identical results.
struct CX3 {
    static constexpr int TW = 32, TH = 32, MT = 4, NT = 1, HT = TH + 2, WT = TW + 2, NC = 32, ROW = TNR_X3_ROW;
    static constexpr int IN_ROWS = HT * WT, W_ROWS = 9 * NC;
    static constexpr size_t LDS_BYTES = (size_t)(IN_ROWS + W_ROWS) * ROW * sizeof(float);
};

__device__ __forceinline__ void chain_x3w8_body(const ConvK a, const int cb, const int tx, const int ty, const int n, float *smem,
                                                const int wait_chunk, const ChainWait &wait) {
    using G = CX3;
    constexpr int MT = G::MT, NT = G::NT, WT = G::WT, NC = G::NC, ROW = G::ROW;
    float *s_in = smem, *s_w = smem + G::IN_ROWS * ROW;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, half = lane >> 5;
    const __amdgpu_buffer_rsrc_t x_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)((unsigned)a.N * a.H * a.W * a.x_ct * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rs =
        __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((unsigned)a.N * a.Ho * a.Wo * a.y_ct * 4u), 0x00020000);
    const int ty0 = ty * G::TH, tx0 = tx * G::TW;
    const int nchunks = a.KinP / TNR_CK;

    constexpr int IN_IT = (G::IN_ROWS * 4 + 511) / 512, W_IT = (G::W_ROWS * 4 + 511) / 512;
    int in_off[IN_IT], w_off[W_IT];      // element offsets without the chunk's channel offset, -1: zero fill
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
        const int i = tid + it * 512, row = i >> 2, q = i & 3;
        const int hr = row / WT, hc = row - hr * WT;
        const int Y = ty0 + hr - 1, X = tx0 + hc - 1;
        const bool ok = (row < G::IN_ROWS) & (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
        in_off[it] = ok ? (((n * a.H + Y) * a.W + X) * a.x_ct + a.x_co + q * 4) : -1;
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
        const int i = tid + it * 512, row = i >> 2, q = i & 3;
        const int t = row / NC, co = row - t * NC;
        const int cog = cb * NC + co;
        w_off[it] = (row < G::W_ROWS && cog < a.KoutP) ? ((t * a.KoutP + cog) * a.KinP + q * 4) : -1;
    }
    f32x4 rin[IN_IT], rw[W_IT];
    auto load_chunk = [&](int chunk) {
        const int c0 = chunk * TNR_CK;
#pragma unroll
        for (int it = 0; it < IN_IT; ++it) {       // system-coherent: the channels may have been written by another CU in this launch
            const unsigned bo = in_off[it] >= 0 ? (unsigned)(in_off[it] + c0) * 4u : 0xfffffff0u;      // (past the end: the range check returns 0)
            rin[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (int)bo, 0, TNR_COH_LOAD_AUX));
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (w_off[it] >= 0) v = *reinterpret_cast<const f32x4 *>(a.wp + (size_t)w_off[it] + c0);
            rw[it] = v;
        }
    };
    auto store_item = [&](float *base, int i, int rows, const f32x4 v) {
        const int row = i >> 2, q = i & 3;
        if (row < rows) {
            tnr_f32x2 pc[3];
            tnr_split4_bf16x3(v, pc);
            float *dst = base + row * ROW + 4 * ((q >> 1) ^ ((row >> TNR_X3_SWZ) & 1)) + 2 * (q & 1);
            *reinterpret_cast<tnr_f32x2 *>(dst) = pc[0];
            *reinterpret_cast<tnr_f32x2 *>(dst + 8) = pc[1];
            *reinterpret_cast<tnr_f32x2 *>(dst + 16) = pc[2];
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int it = 0; it < IN_IT; ++it) store_item(s_in, tid + it * 512, G::IN_ROWS, rin[it]);
#pragma unroll
        for (int it = 0; it < W_IT; ++it) store_item(s_w, tid + it * 512, G::W_ROWS, rw[it]);
    };

    const int apix0 = (wave * MT) * WT + li;                        // M-tile mi: + mi * WT
    const int boff = li * ROW + 4 * (half ^ ((li >> TNR_X3_SWZ) & 1));       // row t * 32 + li: the swizzle bit is that of li
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][0][r] = 0.f;
    tnr_bf16x8 ca[MT][3], cb_[2][3];
    auto load_a = [&](int t, int m0, int m1) {
#pragma unroll
        for (int mi = m0; mi < m1; ++mi) {
            const int pp = apix0 + mi * WT + (t / 3) * WT + (t % 3);
            const float *src = s_in + pp * ROW + 4 * (half ^ ((pp >> TNR_X3_SWZ) & 1));
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) ca[mi][sp] = *reinterpret_cast<const tnr_bf16x8 *>(src + 8 * sp);
        }
    };
    auto load_b = [&](int t, int set) {
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) cb_[set][sp] = *reinterpret_cast<const tnr_bf16x8 *>(s_w + boff + t * NC * ROW + 8 * sp);
    };
    auto mma = [&](int m0, int m1, int set) {
        constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int mi = m0; mi < m1; ++mi)
                acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[mi][TA[p]], cb_[set][TB[p]], acc[mi][0], 0, 0, 0);
    };

    if (wait_chunk == 0) wait();
    load_chunk(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        __syncthreads();
        store_chunk();
        if (chunk == 1) wait.drain();
        __syncthreads();
        if (chunk == 1) wait.publish();
        if (chunk + 1 < nchunks) {
            if (chunk + 1 == wait_chunk) wait();
            load_chunk(chunk + 1);
        }
        load_b(0, 0);
        load_a(0, 0, MT);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) load_b(t + 1, (t + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(0, 2, t & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < 9) load_a(t + 1, 0, 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(2, 4, t & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < 9) load_a(t + 1, 2, 4);
        }
    }
    conv_epilogue_dpp<TNR_CONV_3x3, G::TW, NT, MT, true>(a, acc, cb, n, ty0, tx0, 0, wave, li, half, y_rs);
}

__global__ void __launch_bounds__(512, 1) conv_chain_x3w8_kernel(const ChainK c) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int pend_tile = -1;
    unsigned pend_value = 0;
    for (int s = 0; s < c.nstages; ++s) {
        int wait_chunk;
        const ConvK st = chain_stage(s, &wait_chunk);
        const int ncb = st.KoutP >> 5;
        for (int tile = blockIdx.x; tile < c.tiles; tile += gridDim.x) {
            int q = tile;
            const int tx = q % c.tiles_x;
            q /= c.tiles_x;
            const int ty = q % c.tiles_y;
            const int n = q / c.tiles_y;
            const ChainWait w{c.progress, c.base + (unsigned)s, n, ty, tx, c.tiles_x, c.tiles_y, c.err, &pend_tile, pend_value};
            for (int cb = 0; cb < ncb; ++cb) {
                int txo = tx, tyo = ty, no = n;
                asm volatile("" : "+s"(txo), "+s"(tyo), "+s"(no));
                chain_x3w8_body(st, cb, txo, tyo, no, smem, cb == 0 ? wait_chunk : -1, w);
            }
            pend_tile = tile;
            pend_value = c.base + (unsigned)s + 1u;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (pend_tile >= 0 && threadIdx.x == 0)
        __hip_atomic_store(c.progress + pend_tile, pend_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int CH_TH = 16, CH_TW = 32;
constexpr size_t chain_lds(int bf = 0) {
    // (the split-operand form keeps the input tile as three bf16 planes: 96-byte rows, conv_body.h)
    const size_t lds_main = (size_t)((CH_TH + 2) * (CH_TW + 2) * ((bf == 2 && TNR_X3_REFILL != 0) ? TNR_X3_ROW : TNR_PST) + 9 * 32 * TNR_PST) * sizeof(float);
    const size_t lds_epi = (size_t)4 * 4 * 32 * 32 * sizeof(float);
    return lds_main > lds_epi ? lds_main : lds_epi;
}
static_assert(chain_lds(2) <= 80 * 1024, "two chain workgroups per CU");

int chain_capacity(int *out) {
    static int cap = 0;
    if (cap == 0) {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
            tnr_set_error("conv_chain: cannot query the device");
            return TNR_ELAUNCH;
        }
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv_chain_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)chain_lds()) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(conv_chain_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)chain_lds()) != hipSuccess ||
#ifdef TNR_CONV_DL_EXPERIMENT
            hipFuncSetAttribute(reinterpret_cast<const void *>(conv_chain_kernel<0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)chain_lds()) != hipSuccess ||
#endif
            hipFuncSetAttribute(reinterpret_cast<const void *>(conv_chain_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)chain_lds(2)) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_chain_kernel<0>, 256, chain_lds()) != hipSuccess) {
            tnr_set_error("conv_chain: cannot size the grid");
            return TNR_ELAUNCH;
        }
        // LDS allows two workgroups per CU; never trust a larger answer (the progress waits need every
        // workgroup of the grid to be resident at once)
        if (per_cu > 2) per_cu = 2;
        int per_cu_x3 = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_x3, conv_chain_kernel<2>, 256, chain_lds(2)) != hipSuccess || per_cu_x3 < per_cu) {
            tnr_set_error("conv_chain: the split-operand form does not reach %d workgroups per CU (%d)", per_cu, per_cu_x3);
            return TNR_ELAUNCH;
        }
        if (per_cu < 1 || cus < 1) {
            tnr_set_error("conv_chain: kernel does not fit a CU");
            return TNR_ELAUNCH;
        }
        cap = per_cu * cus;
    }
    *out = cap;
    return TNR_OK;
}

}  // namespace

extern "C" int64_t tnr_conv_chain_workspace_bytes(const tnr_conv_desc *d) {
    if (d == nullptr) return 0;
    // progress counters (sized for the 8 x 32 tiles of tnr_conv_sweep, which shares the buffer: twice the 16 x 32 tiles of this
    // kernel), the sweep's tile dispensers (words of their own: conv_handoff.h), per-CU arrival counters, 2 x 2 tile dispensers, the
    // error word (ALWAYS the last word of the buffer)
    const int64_t tiles = (int64_t)tnr_cdiv(d->Wo, CH_TW) * tnr_cdiv(d->Ho, 8) * d->N;
    return (tiles + CH_SWEEP_WORDS + CH_CU_KEYS + 4 + 1) * (int64_t)sizeof(uint32_t);
}

extern "C" int tnr_conv_chain(const tnr_conv_desc *stages, const int32_t *fresh_from, int32_t n, uint32_t *ws, int64_t ws_bytes,
                              uint32_t epoch, void *stream) {
    TNR_REQUIRE(stages != nullptr && fresh_from != nullptr && ws != nullptr && n >= 1 && n <= TNR_CHAIN_MAX,
                "conv_chain: 1..%d stages", TNR_CHAIN_MAX);
    const tnr_conv_desc &d0 = stages[0];
    TNR_REQUIRE(tnr_conv_chain_workspace_bytes(&d0) <= ws_bytes, "conv_chain: workspace too small");
    ChainK c;
    c.nstages = n;
    c.tiles_x = tnr_cdiv(d0.Wo, CH_TW);
    c.tiles_y = tnr_cdiv(d0.Ho, CH_TH);
    c.tiles = c.tiles_x * c.tiles_y * d0.N;
    c.progress = ws;
    c.cu_ctr = ws + (ws_bytes / 4 - 1 - 4 - CH_CU_KEYS);      // (behind the progress counters of either tile grid)
    c.tile_ctr = c.cu_ctr + CH_CU_KEYS;
    c.err = tnr_fault_word_or(ws + ws_bytes / 4 - 1);
    c.set = (int)(epoch & 1u);
    c.base = epoch * (uint32_t)(TNR_CHAIN_MAX + 2);
    for (int i = 0; i < n; ++i) {
        const tnr_conv_desc *d = &stages[i];
        TNR_REQUIRE(d->x.ptr && d->y.ptr && d->wp, "conv_chain: null pointer in stage %d", i);
        TNR_REQUIRE(d->mode == TNR_CONV_3x3 && d->N == d0.N && d->H == d0.H && d->W == d0.W && d->Ho == d0.H && d->Wo == d0.W,
                    "conv_chain: stage %d is not a 3x3 convolution over the pixel grid of stage 0", i);
        TNR_REQUIRE(d->KinP >= 2 * TNR_CK, "conv_chain: stage %d needs at least 32 (padded) input channels", i);
        TNR_REQUIRE((d->Cout % 32) == 0 && d->KoutP == d->Cout && (d->Cin % 4) == 0 && (d->KinP % TNR_CK) == 0 && d->Cin <= d->KinP,
                    "conv_chain: stage %d: Cout must be a multiple of 32, Cin of 4", i);
        TNR_REQUIRE((d->x.ctot % 4) == 0 && (d->x.coff % 4) == 0 && (d->y.ctot % 4) == 0 && (d->y.coff % 4) == 0,
                    "conv_chain: stage %d: views must be 4-channel aligned", i);
        TNR_REQUIRE(d->r1.ptr == nullptr || ((d->r1.ctot % 4) == 0 && (d->r1.coff % 4) == 0 && (d->r1_ch % 4) == 0),
                    "conv_chain: stage %d: r1 view must be 4-channel aligned", i);
        TNR_REQUIRE(d->r2.ptr == nullptr || ((d->r2.ctot % 4) == 0 && (d->r2.coff % 4) == 0), "conv_chain: stage %d: r2 view", i);
        TNR_REQUIRE(d->noise_pos >= 0 && d->noise_pos <= 2, "conv_chain: stage %d: bad noise_pos %d", i, d->noise_pos);
        TNR_REQUIRE(d->m.ptr == nullptr || ((d->m.ctot % 4) == 0 && (d->m.coff % 4) == 0 && (d->m_lo % 4) == 0 && (d->m_hi % 4) == 0),
                    "conv_chain: stage %d: mask view", i);
        TNR_REQUIRE((int64_t)d->N * d->H * d->W * d->x.ctot < (1LL << 30) && (int64_t)d->N * d->H * d->W * d->y.ctot < (1LL << 30),
                    "conv_chain: stage %d: buffers above 4 GiB are not addressable through a buffer descriptor", i);
        TNR_REQUIRE(fresh_from[i] < d->Cin && (i > 0 || fresh_from[i] < 0), "conv_chain: stage %d: bad fresh_from %d", i, fresh_from[i]);
        ConvK &k = c.st[i];
        k.x = d->x.ptr; k.x_ct = d->x.ctot; k.x_co = d->x.coff;
        k.N = d->N; k.H = d->H; k.W = d->W; k.Cin = d->Cin;
        k.wp = d->wp; k.KinP = d->KinP; k.KoutP = d->KoutP;
        k.y = d->y.ptr; k.y_ct = d->y.ctot; k.y_co = d->y.coff; k.Ho = d->Ho; k.Wo = d->Wo; k.Cout = d->Cout;
        k.bias = d->bias; k.act = d->act; k.slope = d->slope; k.alpha = d->alpha;
        k.r1 = d->r1.ptr; k.r1_ct = d->r1.ctot; k.r1_co = d->r1.coff; k.r1_ch = d->r1_ch; k.beta1 = d->beta1;
        k.r2 = d->r2.ptr; k.r2_ct = d->r2.ctot; k.r2_co = d->r2.coff; k.alpha2 = d->alpha2;
        k.m = d->m.ptr; k.m_ct = d->m.ctot; k.m_co = d->m.coff; k.m_lo = d->m_lo; k.m_hi = d->m_hi; k.m_slope = d->m_slope;
        k.noise_pos = d->noise_pos; k.noise_sigma = d->noise_sigma; k.noise_k0 = d->noise_key0; k.noise_k1 = d->noise_key1; k.noise_pix0 = d->noise_pix0;
        k.tiles_x = c.tiles_x; k.tiles_y = c.tiles_y; k.ncb = d->KoutP / 32;
        k.th_space = d->Ho; k.tw_space = d->Wo;
        k.ksplit = 1; k.split_stride = 0; k.bf = d->mma; k.reflect = d->pad_mode == 1;
        TNR_REQUIRE(d->mma == d0.mma, "conv_chain: stage %d: all stages share one matrix-core precision", i);
        c.wait_chunk[i] = fresh_from[i] < 0 ? -1 : fresh_from[i] / TNR_CK;
    }
    for (int i = n; i < TNR_CHAIN_MAX; ++i) { c.st[i] = c.st[0]; c.wait_chunk[i] = -1; }
    int cap = 0;
    const int rc = chain_capacity(&cap);
    if (rc != TNR_OK) return rc;
    const int grid = c.tiles < cap ? c.tiles : cap;
    c.dyn = (c.tiles == cap && (c.tiles & 1) == 0 && d0.N >= 2 && (d0.N & 1) == 0) ? 1 : 0;   // whole images per population
#ifdef TNR_CONV_DL_EXPERIMENT
    const char *dl_env = std::getenv("TNR_CONV_DL");
    int dl = (dl_env != nullptr && d0.mma == TNR_MMA_F32) ? dl_env[0] - '0' : 0;
    for (int i = 0; i < n; ++i)
        if (c.st[i].Cin != c.st[i].KinP || c.st[i].reflect) dl = 0;
    static_assert(Dk8Geom<1, 4>::LDS_BYTES <= chain_lds(), "the LDS-DMA form fits the chain's LDS allocation");
    if (dl == 2)
        hipLaunchKernelGGL((conv_chain_kernel<0, 2>), dim3(grid), dim3(256), chain_lds(), (hipStream_t)stream, c);
    else
#endif
    if (d0.mma == TNR_MMA_BF16X3) {
        // the 8-wave form (TNR_CHAIN_X3W8=1): 32 x 32 tiles, one workgroup per CU
        static const bool w8 = [] { const char *e = std::getenv("TNR_CHAIN_X3W8"); return e != nullptr && e[0] == '1'; }();
        bool ok8 = w8;
        for (int i = 0; i < n; ++i) ok8 = ok8 && c.st[i].Cin == c.st[i].KinP && !c.st[i].reflect;
        if (ok8) {
            static int cus = 0;
            if (cus == 0) {
                int dev = 0;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void *>(conv_chain_x3w8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)CX3::LDS_BYTES) != hipSuccess || cus < 1) {
                    cus = 0;
                    tnr_set_error("conv_chain: cannot set up the 8-wave form");
                    return TNR_ELAUNCH;
                }
            }
            ChainK c8 = c;
            c8.tiles_x = tnr_cdiv(d0.Wo, CX3::TW);
            c8.tiles_y = tnr_cdiv(d0.Ho, CX3::TH);
            c8.tiles = c8.tiles_x * c8.tiles_y * d0.N;
            c8.dyn = 0;
            const int grid8 = c8.tiles < cus ? c8.tiles : cus;
            constexpr size_t lds8 = CX3::LDS_BYTES;
            hipLaunchKernelGGL(conv_chain_x3w8_kernel, dim3(grid8), dim3(512), lds8, (hipStream_t)stream, c8);
            return tnr_check_launch("conv_chain_x3w8");
        }
        hipLaunchKernelGGL(conv_chain_kernel<2>, dim3(grid), dim3(256), chain_lds(2), (hipStream_t)stream, c);
    }
    else if (d0.mma == TNR_MMA_BF16)
        hipLaunchKernelGGL(conv_chain_kernel<1>, dim3(grid), dim3(256), chain_lds(), (hipStream_t)stream, c);
    else
        hipLaunchKernelGGL(conv_chain_kernel<0>, dim3(grid), dim3(256), chain_lds(), (hipStream_t)stream, c);
    return tnr_check_launch("conv_chain");
}
