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
                const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                // (int) difference: robust to the counter base wrapping around
                while ((int)(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - need) < 0) {
                    __builtin_amdgcn_s_sleep(1);
                    if (__builtin_amdgcn_s_memtime() - t0 > (4ull << 30)) {   // ~2 s: report instead of hanging the GPU
                        *err = 1u;
                        break;
                    }
                }
            }
        }
        __syncthreads();
    }
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

constexpr int CH_TH = 16, CH_TW = 32;
constexpr int CH_CU_KEYS = 16 * 128;   // (xcc, se, sh, cu) keys of chain_kernel's per-CU arrival counters
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
    const int64_t tiles = (int64_t)tnr_cdiv(d->Wo, CH_TW) * tnr_cdiv(d->Ho, CH_TH) * d->N;
    // progress counters, per-CU arrival counters, 2 x 2 tile dispensers, the error word (last)
    return (tiles + CH_CU_KEYS + 4 + 1) * (int64_t)sizeof(uint32_t);
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
    c.cu_ctr = ws + c.tiles;
    c.tile_ctr = c.cu_ctr + CH_CU_KEYS;
    c.err = c.tile_ctr + 4;
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
    if (d0.mma == TNR_MMA_BF16X3)
        hipLaunchKernelGGL(conv_chain_kernel<2>, dim3(grid), dim3(256), chain_lds(2), (hipStream_t)stream, c);
    else if (d0.mma == TNR_MMA_BF16)
        hipLaunchKernelGGL(conv_chain_kernel<1>, dim3(grid), dim3(256), chain_lds(), (hipStream_t)stream, c);
    else
        hipLaunchKernelGGL(conv_chain_kernel<0>, dim3(grid), dim3(256), chain_lds(), (hipStream_t)stream, c);
    return tnr_check_launch("conv_chain");
}
