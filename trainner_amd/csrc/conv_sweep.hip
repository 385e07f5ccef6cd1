// A residual dense block's five growing-K 3x3 convolutions (RRDBNet_arch.py:150-163) -- or the five of its gradient mirror -- in ONE
// launch, TNR_MMA_BF16X3 only, with every input channel chunk feeding ALL the stages that consume it ("sweep").
//
// Why (profiles/r03a_chain_timeline.txt): on the bf16x3 path conv_chain_kernel is bound by its memory traffic, not by the matrix core.
// Stage k re-reads the 64 + 32 (k - 1) channels the stages before it already read (832 channel reads per pixel for 192 distinct
// channels), a chunk's MFMA phase shrank to 6.9 k cycles per wave and the burst of next-chunk loads alone takes 6 k cycles to issue
// (120 KB per CU per chunk round against ~10 B / cycle / CU of streaming bandwidth).  Here a workgroup keeps the accumulators of all
// six 32-channel output groups of its tile (x1, x2, x3, x4 and the two halves of x5: 192 fp32 per pixel) in registers and sweeps the
// input channels once per phase:
//     phase 0 = channels of x      -> accumulate into x1 .. x5        phase 1 = channels of x1 -> x2 .. x5      ...
//     phase 4 = channels of x4 -> x5
// so an input element is read (and split into its three bf16 planes) at most twice instead of up to five times, an A fragment read
// from LDS feeds up to 5 x 6 MFMAs instead of 6, and the weights arrive PRE-SPLIT from HBM (tnr_conv_sweep_pack builds the LDS image
// once per optimiser step): no vector-ALU work in the MFMA phase, a pure copy on the weight path.
//
// Each phase runs as two passes.  Pass a computes ONLY the stage that completes in this phase (one output group), so that its epilogue
// (bias / LeakyReLU / residuals / mask, system-coherent stores) is issued as early as possible; pass b sweeps the same channels again
// for the remaining groups while those stores drain and the neighbouring tiles catch up -- the next phase's first chunk (this tile's
// and its 8 neighbours' fresh output: 3x3 halo) is fetched during the last chunk of pass b.  Same hand-off protocol as conv_chain.hip
// (conv_handoff.h).
//
// Geometry: tile 8 x 32 pixels, one workgroup per CU, in two forms.  Four-wave form (conv_sweep4_kernel, the default): one wave per
// SIMD, wave w owns tile rows 2 w, 2 w + 1 (two 32-pixel M-tiles) x up to six N-tiles (192 accumulator registers); see the comment in
// front of it.  Eight-wave form (conv_sweep_kernel, TNR_SWEEP_WAVES=8): wave w owns tile row w x up to six N-tiles (96 accumulator
// VGPRs).  Both: input chunk = 16 channels of the 10 x 34 halo tile as 96-byte rows of three bf16 planes (split once, by the stager,
// double-buffered: 2 x 36.9 KB with a row of its own per staging slot); weights stream through a three-slot ring of 9 units (unit =
// one tap x one N-tile x 16 channels = 32 rows x 96 B = 3 KB; 3 x 27 KB) in exactly the order the MFMA loop consumes them -- one
// workgroup barrier per slot.
// Deadlock freedom: the grid is at most one workgroup per CU and covers WHOLE images per round (dependencies never cross images).
// Eight-wave form: tiles dealt statically, so every workgroup must be resident (a waited-for tile always belongs to a running workgroup
// at an earlier program point).  Four-wave form: tiles DISPENSED in order from an atomic counter -- a waited-for tile was taken by a
// workgroup that is running, or is the next to be dispensed.  Stage s of tile T reads stage s - 1 of tile T + tiles_x + 1, so T
// completes once tiles up to T + 4 (tiles_x + 1) are held: progress needs 4 (tiles_x + 1) + 1 RESIDENT workgroups (21 on a 128-wide
// image; never more than the tiles of one image, which the grid always covers), not the whole grid.  Waits are bounded either way (fault latch, tnr_set_fault_word).
// Arithmetic (split, kept partial products and their order, channel and tap order, epilogue) is that of conv_tile_body<.., BF = 2>:
// results are bit-identical to five tnr_conv_forward launches in TNR_MMA_BF16X3.
#include <stddef.h>
#include <string.h>
#include <type_traits>
#include <utility>
#include "conv_body.h"
#include "conv_epilogue.h"
#include "conv_handoff.h"

namespace {

constexpr int SW_TH = 8, SW_TW = 32, SW_HT = SW_TH + 2, SW_WT = SW_TW + 2;
constexpr int SW_ROW = TNR_X3_ROW;                                  // floats per LDS row: three planes of 16 bf16
constexpr int SW_A_ROWS = SW_HT * SW_WT;                            // 340 halo pixels
constexpr int SW_A_ALLOC_ROWS = 384;                                // rows per LDS buffer: every staging item (512 x 3 or 256 x 6 float4) has a row of its own,
constexpr int SW_A_FLOATS = SW_A_ALLOC_ROWS * SW_ROW;               // so the four-wave form stores without lane masks (rows >= 340 are never read)
constexpr int SW_UNIT_FLOATS = 32 * SW_ROW;                         // one tap x one N-tile x 16 channels
constexpr int SW_SLOT_UNITS = 9;
constexpr int SW_SLOT_FLOATS = SW_SLOT_UNITS * SW_UNIT_FLOATS;
constexpr int SW_RING = 3;                                          // weight slots in LDS: slot k + 2 is fetched while slot k is consumed
constexpr size_t SW_LDS_BYTES = (size_t)(2 * SW_A_FLOATS + SW_RING * SW_SLOT_FLOATS) * sizeof(float);      // 156 672 B
constexpr int SW_NPASS = 9, SW_NSTAGE = 5, SW_NTILE = 6;
constexpr int SW_A_IT = (SW_A_ROWS * 4 + 511) / 512;                // staging items (float4) per thread and chunk: 3
constexpr int SW_B_PIECES = (SW_SLOT_UNITS * 3 + 7) / 8;            // 1 KB LDS-DMA pieces per wave and slot: <= 4
static_assert(SW_LDS_BYTES <= 160 * 1024, "one sweep workgroup per CU");

// The plan.  N-tile j = output channels [32 j, 32 j + 32) of the block's 192 (x1, x2, x3, x4, x5 lo, x5 hi); phase f sweeps the input
// channels that stage f - 1 produced (phase 0: the block input) into the N-tiles of stages >= f.  Pass p = 2 f (pass a: the N-tile
// of the stage that completes in this phase first -- plus one more, so that a chunk carries enough MFMAs to cover the next chunk's
// fetch) or 2 f + 1 (pass b: the remaining N-tiles, while the stores of pass a's epilogue drain and the neighbours catch up).
//                                       pass:  0  1  2  3  4  5  6  7  8
constexpr int SW_J0[SW_NPASS] =               { 0, 3, 1, 3, 2, 4, 3, 5, 4 };      // first N-tile
constexpr int SW_NJ[SW_NPASS] =               { 3, 3, 2, 3, 2, 2, 2, 1, 2 };      // N-tiles
constexpr int SW_TT[SW_NPASS] =               { 3, 3, 3, 3, 3, 3, 3, 9, 3 };      // taps per weight slot (T * NJ <= SW_SLOT_UNITS)
__host__ __device__ constexpr int sw_phase(int p) { return p >> 1; }
__host__ __device__ constexpr bool sw_apass(int p) { return (p & 1) == 0; }
__host__ __device__ constexpr int sw_j0(int p) { return SW_J0[p]; }
__host__ __device__ constexpr int sw_nj(int p) { return SW_NJ[p]; }
__host__ __device__ constexpr int sw_T(int p) { return SW_TT[p]; }
// (run-time p: arithmetic instead of a table in memory)
__host__ __device__ constexpr int sw_nj_rt(int p) { return p == 7 ? 1 : ((p == 0 || p == 1 || p == 3) ? 3 : 2); }
__host__ __device__ constexpr int sw_j0_rt(int p) { return p == 0 ? 0 : (p == 2 ? 1 : (p == 4 ? 2 : (p == 5 || p == 8 ? 4 : (p == 7 ? 5 : 3)))); }
__host__ __device__ constexpr int sw_T_rt(int p) { return p == 7 ? 9 : 3; }
constexpr bool sw_plan_ok() {
    for (int p = 0; p < SW_NPASS; ++p) {
        if (SW_TT[p] * SW_NJ[p] > SW_SLOT_UNITS || 9 % SW_TT[p]) return false;
        if (sw_nj_rt(p) != SW_NJ[p] || sw_j0_rt(p) != SW_J0[p] || sw_T_rt(p) != SW_TT[p]) return false;
        if (sw_apass(p) && SW_J0[p] != (sw_phase(p) < 4 ? sw_phase(p) : 4)) return false;           // pass a starts at the completing stage's N-tile
        if (!sw_apass(p) && SW_J0[p] != SW_J0[p - 1] + SW_NJ[p - 1]) return false;                  // pass b continues where pass a stopped
        if ((!sw_apass(p) || p == 8) && SW_J0[p] + SW_NJ[p] != SW_NTILE) return false;              // ... up to the last N-tile
    }
    return true;
}

static_assert(sw_plan_ok(), "sweep plan");
static_assert(SW_J0[1] == SW_J0[3] && SW_NJ[1] == SW_NJ[3] && SW_TT[1] == SW_TT[3] && SW_J0[5] == SW_J0[8] && SW_NJ[5] == SW_NJ[8] && SW_TT[5] == SW_TT[8], "passes 1 / 3 and 5 / 8 share a body");

struct SweepK {
    int nck0;                          // chunks of phase 0 (channels of the block input / 16); the other phases have 2
    int tiles_x, tiles_y, tpi, tiles;  // 8 x 32 pixel tiles; tpi = tiles per image
    unsigned *progress;                // [tiles]: base + (stages of that tile whose output is visible)
    unsigned base;
    unsigned *err;
    int nxcd;                          // 8: one dispenser per XCD (contiguous shares), 1: one for the whole grid
    unsigned *disp;                    // [9] tile dispensers of the four-wave form (one per XCD) + finished-workgroup count; zero between launches
    const float *wq;                   // pre-split weight stream (tnr_conv_sweep_pack)
    int wq_bytes;
    ConvK st[SW_NSTAGE];
};

__host__ __device__ inline int sw_nchunks(int p, int nck0) { return sw_phase(p) == 0 ? nck0 : 2; }
__host__ __device__ inline int sw_ch_lo(int p, int nck0) { return sw_phase(p) == 0 ? 0 : 16 * nck0 + 32 * (sw_phase(p) - 1); }
__host__ __device__ inline int sw_total_units(int nck0) {
    int u = 0;
    for (int p = 0; p < SW_NPASS; ++p) u += sw_nchunks(p, nck0) * 9 * sw_nj_rt(p);
    return u;
}

// ---- weight stream: for every pass, chunk, tap and N-tile of the pass (in that order) one unit of 32 rows (output channels) x 96 B:
// row r holds the three bf16 planes (hi, mid, lo) of input channels 16 c .. 16 c + 15, the 16-byte slot of channels 8 h .. 8 h + 7 of a
// plane at slot h ^ ((r >> TNR_X3_SWZ) & 1) -- byte for byte the LDS image the MFMA loop reads.
struct SweepPackK {
    int nck0, units;
    int direct;                        // 1: the register image of the direct form (conv_sweep4_kernel<true>), 0: the LDS image
    const float *wp[SW_NSTAGE];
    int KinP[SW_NSTAGE], KoutP[SW_NSTAGE];
    float *out;
};

__device__ __forceinline__ void sweep_pack_one(const SweepPackK &a, const int g) {
    const int unit = g >> 6, r = (g >> 1) & 31, sp = g & 1;
    if (unit >= a.units) return;
    int p = 0, u0 = 0;
    for (; p < SW_NPASS; ++p) {
        const int n = sw_nchunks(p, a.nck0) * 9 * sw_nj_rt(p);
        if (unit < u0 + n) break;
        u0 += n;
    }
    const int lu = unit - u0, nj = sw_nj_rt(p);
    const int jj = lu % nj, ct = lu / nj, tap = ct % 9, ck = ct / 9;
    const int j = sw_j0_rt(p) + jj;
    const int s = j < 4 ? j : 4, cb = j == 5 ? 1 : 0;
    const int h = sp ^ ((r >> TNR_X3_SWZ) & 1);
    const int ch = sw_ch_lo(p, a.nck0) + 16 * ck + 8 * h;
    const float *src = a.wp[s] + ((size_t)tap * a.KoutP[s] + cb * 32 + r) * a.KinP[s] + ch;
    const f32x4 q0 = *reinterpret_cast<const f32x4 *>(src), q1 = *reinterpret_cast<const f32x4 *>(src + 4);
    tnr_bf16x8 pl[3];
    tnr_split_bf16x3(q0, q1, pl);
    if (a.direct) {
        // direct form: plane k of a unit is ONE 1 KB block in the order the lanes hold it -- lane (half = channel octet h, li = row r)
        // owns 16 bytes -- so a fragment load is a fully coalesced 1 KB read straight into the MFMA operand registers
        float *dst = a.out + (size_t)unit * SW_UNIT_FLOATS + (h * 32 + r) * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<tnr_bf16x8 *>(dst + k * 256) = pl[k];
        return;
    }
    float *dst = a.out + (size_t)unit * SW_UNIT_FLOATS + r * SW_ROW + 4 * sp;
#pragma unroll
    for (int k = 0; k < 3; ++k) *reinterpret_cast<tnr_bf16x8 *>(dst + 8 * k) = pl[k];
}

__global__ void __launch_bounds__(256) sweep_pack_kernel(const SweepPackK a) { sweep_pack_one(a, blockIdx.x * 256 + threadIdx.x); }

// the same for MANY blocks in one launch (tnr_conv_sweep_pack_batch): blockIdx.y = entry of a device table of the kernel arguments above
__global__ void __launch_bounds__(256) sweep_pack_batch_strided_kernel(const tnr_sweep_pack_item *items) {
    const SweepPackK a = *reinterpret_cast<const SweepPackK *>(items[blockIdx.y].opaque);
    sweep_pack_one(a, blockIdx.x * 256 + threadIdx.x);
}

// The N-tiles [J0, J0 + NJ) x T taps of one ring slot: (tap, N-tile) units in stream order, 6 MFMAs each.
template <int J0, int NJ, int T>
__device__ __forceinline__ void sweep_slot(f32x16 (&acc)[SW_NTILE], const float *s_a, const float *s_b_lane, const int apix0,
                                           const int half, const int tap0) {
    constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};      // the six kept partial products, smallest first
    tnr_bf16x8 fa[2][3], fb[2][3];
    auto load_a = [&](int tap, int set) {
        const int dy = tap / 3, dx = tap - 3 * dy;
        const int pp = apix0 + dy * SW_WT + dx;
        const float *src = s_a + pp * SW_ROW + 4 * (half ^ ((pp >> TNR_X3_SWZ) & 1));
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) fa[set][sp] = *reinterpret_cast<const tnr_bf16x8 *>(src + 8 * sp);
    };
    auto load_b = [&](int u, int set) {
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) fb[set][sp] = *reinterpret_cast<const tnr_bf16x8 *>(s_b_lane + u * SW_UNIT_FLOATS + 8 * sp);
    };
    load_a(tap0, 0);
    load_b(0, 0);
#pragma unroll
    for (int tt = 0; tt < T; ++tt) {
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int u = tt * NJ + jj;
            // the fragments of the next unit are read before this unit's MFMAs issue (one register set each way)
#ifndef SW_ABL_NOFA          /* (ablation builds: fragments read once per slot -- timing only, results invalid) */
            if (jj == 0 && tt + 1 < T) load_a(tap0 + tt + 1, (tt + 1) & 1);
#endif
#ifndef SW_ABL_NOFB
            if (u + 1 < T * NJ) load_b(u + 1, (u + 1) & 1);
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 6; ++p)
                acc[J0 + jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tt & 1][TA[p]], fb[u & 1][TB[p]], acc[J0 + jj], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Stage s of the kernel-argument table, fetched with scalar loads (indexing the by-value struct with a run-time s would make the
// compiler copy all of it to scratch)
__device__ __forceinline__ ConvK sweep_stage(int s) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) SweepK karg_sweep;
    karg_sweep *ka = (karg_sweep *)__builtin_amdgcn_kernarg_segment_ptr();   // constant address space: s_load
    return ka->st[s];
#else
    return ConvK();
#endif
}

// cursor over the weight stream's slots (wave-uniform)
struct SweepCur {
    int p, ck, sl, unit;
    __device__ __forceinline__ bool valid() const { return p < SW_NPASS; }
    __device__ __forceinline__ int pieces() const { return sw_T_rt(p) * sw_nj_rt(p) * 3; }      // 1 KB pieces of the slot
    __device__ __forceinline__ void advance(int nck0) {
        unit += sw_T_rt(p) * sw_nj_rt(p);
        if (++sl * sw_T_rt(p) == 9) {
            sl = 0;
            if (++ck == sw_nchunks(p, nck0)) {
                ck = 0;
                ++p;
            }
        }
    }
};

typedef __attribute__((address_space(3))) float sw_lds_float;

#ifdef SW_TIMELINE   /* probe build (tools/build_variant.py sw_tl -DSW_TIMELINE): cycles wave 0 of every workgroup spends per part of the loop */
__device__ unsigned long long sw_tl[16];
__device__ unsigned long long sw_tl2[16];      // four-wave form, chunks with 3 N-tiles: cycles / count per unit class
#define SW_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define SW_ADD(i, d) do { tl[i] += (d); } while (0)
#else
#define SW_T(v) do { } while (0)
#define SW_ADD(i, d) do { } while (0)
#endif

// at most n vector-memory operations of this wave still in flight (n wave-uniform, 0 .. SW_B_PIECES)
__device__ __forceinline__ void sw_wait_vm(int n) {
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    }
}
static_assert(SW_B_PIECES <= 4, "sw_wait_vm covers 0 .. 4");

__global__ void __launch_bounds__(512, 1) conv_sweep_kernel(const SweepK c) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_a = smem, *s_b = smem + 2 * SW_A_FLOATS;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, half = lane >> 5;
    // blocks b, b + 8, ... share an XCD (observed, speed only): give every XCD a contiguous run of tiles, i.e. whole image halves,
    // so that the halo re-reads of neighbouring tiles meet in one L2
    int b = blockIdx.x;
    const int g = gridDim.x;
    if ((g & 7) == 0) b = (b & 7) * (g >> 3) + (b >> 3);
    const ConvK &x0 = c.st[0];
    const __amdgpu_buffer_rsrc_t x_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x0.x), 0, (int)((unsigned)x0.N * x0.H * x0.W * x0.x_ct * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(c.wq), 0, c.wq_bytes, 0x00020000);
    sw_lds_float *ring = (sw_lds_float *)s_b;
    const int apix0 = wave * SW_WT + li;
    const float *s_b_lane = s_b + li * SW_ROW + 4 * (half ^ ((li >> TNR_X3_SWZ) & 1));
    int pend_tile = -1;          // tile whose newest stage output still has to be published (wave-uniform)
    unsigned pend_value = 0;
    int pend_age = 0;
#ifdef SW_TIMELINE
    unsigned long long tl[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tl_begin = __builtin_amdgcn_s_memtime();
#endif

    for (int tile = b; tile < c.tiles; tile += g) {
        SW_T(t_tile0);
        const int n = tile / c.tpi, rem = tile - n * c.tpi;
        const int ty = rem / c.tiles_x, tx = rem - ty * c.tiles_x;
        const int ty0 = ty * SW_TH, tx0 = tx * SW_TW;
        // ---- staging plan of the input tile: item i = tid + 512 it -> halo pixel i / 4, channel quad i % 4
        int in_off[SW_A_IT], a_dst[SW_A_IT];
#pragma unroll
        for (int it = 0; it < SW_A_IT; ++it) {
            const int i = tid + it * 512, row = i >> 2, q = i & 3;
            const int hr = row / SW_WT, hc = row - hr * SW_WT;
            const int Y = ty0 + hr - 1, X = tx0 + hc - 1;
            const bool in = (row < SW_A_ROWS) & (Y >= 0) & (Y < x0.H) & (X >= 0) & (X < x0.W);
            in_off[it] = in ? (((n * x0.H + Y) * x0.W + X) * x0.x_ct + x0.x_co + q * 4) : -1;
            a_dst[it] = row < SW_A_ROWS ? row * SW_ROW + 4 * ((q >> 1) ^ ((row >> TNR_X3_SWZ) & 1)) + 2 * (q & 1) : -1;
        }
        f32x4 rin[SW_A_IT];
        auto a_load = [&](int ch) {          // system-coherent: the channels may have been written by another CU in this launch
#pragma unroll
            for (int it = 0; it < SW_A_IT; ++it) {
                const unsigned bo = in_off[it] >= 0 ? (unsigned)(in_off[it] + ch) * 4u : 0xfffffff0u;      // (past the end: the range check returns 0)
                rin[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (int)bo, 0, TNR_COH_LOAD_AUX));
            }
        };
        auto a_store = [&](int buf) {
            float *base = s_a + buf * SW_A_FLOATS;
#pragma unroll
            for (int it = 0; it < SW_A_IT; ++it) {
                if (a_dst[it] >= 0) {
                    tnr_f32x2 pc[3];
                    tnr_split4_bf16x3(rin[it], pc);
                    float *dst = base + a_dst[it];
                    *reinterpret_cast<tnr_f32x2 *>(dst) = pc[0];
                    *reinterpret_cast<tnr_f32x2 *>(dst + 8) = pc[1];
                    *reinterpret_cast<tnr_f32x2 *>(dst + 16) = pc[2];
                }
            }
        };
        // weights: LDS-DMA, no registers.  This wave's 1 KB pieces of a slot (piece q = bytes [1024 q, 1024 q + 1024) of the slot, dealt
        // round-robin to the 8 waves); returns how many it issued.  The stream in HBM is the LDS image: a pure copy.
        auto b_issue = [&](const SweepCur &cu, int slot) -> int {
            const int pieces = cu.pieces();
            const unsigned src0 = (unsigned)cu.unit * (unsigned)(SW_UNIT_FLOATS * 4) + (unsigned)lane * 16u;
            sw_lds_float *dst = ring + slot * SW_SLOT_FLOATS;
            int issued = 0;
#pragma unroll
            for (int i = 0; i < SW_B_PIECES; ++i) {
                const int q = wave + 8 * i;
                if (q < pieces) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, dst + q * 256, 16, (int)(src0 + (unsigned)q * 1024u), 0, 0, 0);
                    ++issued;
                }
            }
            return issued;
        };

        f32x16 acc[SW_NTILE];
#pragma unroll
        for (int j = 0; j < SW_NTILE; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

        // ---- prologue: input chunk 0 into LDS, weight slots 0 and 1 on their way
        SweepCur ld{0, 0, 0, 0};         // the next slot to fetch
        a_load(0);
        __syncthreads();                 // the previous tile's last fragments are consumed
        b_issue(ld, 0);
        ld.advance(c.nck0);
        int n1 = b_issue(ld, 1);         // pieces of slot k + 1 this wave has in flight
        ld.advance(c.nck0);
        a_store(0);
        { SW_T(t_pro); SW_ADD(0, t_pro - t_tile0); }
        int k = 0, e = 0;                // slots / input chunks consumed so far in this tile

        for (int p = 0; p < SW_NPASS; ++p) {
            const int T = sw_T_rt(p);
            const int nchunks = sw_nchunks(p, c.nck0), ch_lo = sw_ch_lo(p, c.nck0);
            for (int ck = 0; ck < nchunks; ++ck) {
                // the next input chunk goes into registers now and into the other LDS buffer after this chunk's last slot
                const bool last_ck = ck + 1 == nchunks;
                const bool has_next = !(last_ck && p == SW_NPASS - 1);
                if (has_next) {
                    int ch_next = ch_lo + 16 * (ck + 1);
                    if (last_ck) {
                        ch_next = sw_ch_lo(p + 1, c.nck0);
                        if (sw_apass(p + 1)) {
                            // the first chunk of the next phase is the output of the stage this tile and its 8 neighbours finished
                            // in pass a of the current phase
                            const ChainWait w{c.progress, c.base + (unsigned)sw_phase(p + 1), n, ty, tx, c.tiles_x, c.tiles_y, c.err,
                                              &pend_tile, pend_value};
#ifndef SW_ABL_NOWAIT
                            SW_T(t_w0);
                            w();
                            { SW_T(t_w1); SW_ADD(1, t_w1 - t_w0); }
#endif
                        }
                    }
#ifndef SW_ABL_NOA
                    SW_T(t_al0);
                    a_load(ch_next);
                    { SW_T(t_al1); SW_ADD(2, t_al1 - t_al0); }
#endif
                }
                for (int sl = 0; sl * T < 9; ++sl) {
                    // slot k must have landed: everything this wave issued before the pieces of slot k + 1 (vector-memory operations
                    // complete in order; the deferred publish -- the newest stage's stores were issued at least two slots ago --
                    // waits for all of them)
                    const bool pub = pend_tile >= 0 && ++pend_age >= 3;
                    SW_T(t_s0);
                    sw_wait_vm(pub ? 0 : n1);
                    SW_T(t_s1);
                    __syncthreads();
                    SW_T(t_s2);
                    SW_ADD(3, t_s1 - t_s0);
                    SW_ADD(4, t_s2 - t_s1);     // slot k is in LDS for every wave, slot k - 1 (= the buffer refilled below) is consumed
                    if (pub) {
                        if (tid == 0) __hip_atomic_store(c.progress + pend_tile, pend_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pend_tile = -1;
                    }
                    n1 = 0;
#ifndef SW_ABL_NOB          /* (ablation builds: timing only, results invalid) */
                    if (ld.valid()) {
                        n1 = b_issue(ld, (k + 2) % SW_RING);
                        ld.advance(c.nck0);
                    }
#endif
                    { SW_T(t_s3); SW_ADD(5, t_s3 - t_s2); }
                    SW_T(t_c0);
                    const float *sa = s_a + (e & 1) * SW_A_FLOATS, *sb = s_b_lane + (k % SW_RING) * SW_SLOT_FLOATS;
                    const int tap0 = sl * T;
                    switch (p) {         // (wave-uniform; one unrolled body per pass shape)
                    case 0: sweep_slot<sw_j0(0), sw_nj(0), sw_T(0)>(acc, sa, sb, apix0, half, tap0); break;
                    case 1: case 3: sweep_slot<sw_j0(1), sw_nj(1), sw_T(1)>(acc, sa, sb, apix0, half, tap0); break;
                    case 2: sweep_slot<sw_j0(2), sw_nj(2), sw_T(2)>(acc, sa, sb, apix0, half, tap0); break;
                    case 4: sweep_slot<sw_j0(4), sw_nj(4), sw_T(4)>(acc, sa, sb, apix0, half, tap0); break;
                    case 6: sweep_slot<sw_j0(6), sw_nj(6), sw_T(6)>(acc, sa, sb, apix0, half, tap0); break;
                    case 7: sweep_slot<sw_j0(7), sw_nj(7), sw_T(7)>(acc, sa, sb, apix0, half, tap0); break;
                    default: sweep_slot<sw_j0(5), sw_nj(5), sw_T(5)>(acc, sa, sb, apix0, half, tap0); break;     // passes 5 and 8
                    }
                    { SW_T(t_c1); SW_ADD(6, t_c1 - t_c0); SW_ADD(15, 1ull); }
                    ++k;
                }
#ifndef SW_ABL_NOA
                SW_T(t_as0);
                if (has_next) a_store((e + 1) & 1);
                { SW_T(t_as1); SW_ADD(7, t_as1 - t_as0); }
#endif
                ++e;
            }
            SW_T(t_e0);
            if (sw_apass(p)) {
                // ---- this phase's stage is complete: epilogue straight from the accumulators (coherent stores), publish deferred
                const int S = sw_phase(p);
                const ConvK a = sweep_stage(S);
                const __amdgpu_buffer_rsrc_t y_rs =
                    __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((unsigned)a.N * a.Ho * a.Wo * a.y_ct * 4u), 0x00020000);
                if (pend_tile >= 0) {          // (only when a pass was too short for the deferred publish: keep the counters in order)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (tid == 0) __hip_atomic_store(c.progress + pend_tile, pend_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pend_tile = -1;
                }
#ifdef SW_ABL_NOEPI
                if (a.alpha == 1.2345e30f)
#endif
                if (S < 4) {
                    f32x16 t[1][1];
                    switch (S) {
                    case 0: t[0][0] = acc[0]; break;
                    case 1: t[0][0] = acc[1]; break;
                    case 2: t[0][0] = acc[2]; break;
                    default: t[0][0] = acc[3]; break;
                    }
                    conv_epilogue_dpp<TNR_CONV_3x3, SW_TW, 1, 1, true, 1, false>(a, t, 0, n, ty0, tx0, 0, wave, li, half, y_rs);
                } else {
                    f32x16 t[1][2];
                    t[0][0] = acc[4];
                    t[0][1] = acc[5];
                    conv_epilogue_dpp<TNR_CONV_3x3, SW_TW, 2, 1, true>(a, t, 0, n, ty0, tx0, 0, wave, li, half, y_rs);
                }
                pend_tile = tile;
                pend_value = c.base + (unsigned)S + 1u;
                pend_age = 0;
                { SW_T(t_e1); SW_ADD(8, t_e1 - t_e0); }
            }
        }
    }
    // nothing in this launch waits for the last stage; publish it anyway so the counters stay consistent
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (pend_tile >= 0 && tid == 0) __hip_atomic_store(c.progress + pend_tile, pend_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef SW_TIMELINE
    if (tid == 0) {
        tl[14] = __builtin_amdgcn_s_memtime() - tl_begin;
        for (int i = 0; i < 16; ++i) atomicAdd(&sw_tl[i], tl[i]);
    }
#endif
}

// =====================================================================================================================================
// The four-wave form (default).  Same tile, same plan, same weight stream, same arithmetic as conv_sweep_kernel -- another division of
// the work among the waves.  Measured on the eight-wave form (profiles/r03q_sweep_lds_ablation.txt, r03r_mfma_slots.txt): with every
// memory operation removed it still ran at 2/3 of the matrix-core rate, because after each slot barrier BOTH waves of a SIMD execute
// the slot's scalar prologue (cursor, addresses, DMA issue, fragment reads) at the same time while the matrix core idles, and each
// wave's 32-pixel M-tile re-reads every weight fragment (LDS reads = 2/3 .. 1 of the MFMA time).  Here:
//   * one wave per SIMD (256 threads, the whole 512-entry register file): a wave owns TWO tile rows (2 M-tiles x 6 N-tiles = 192
//     accumulator registers), so a weight fragment read from LDS feeds 12 MFMAs instead of 6;
//   * a chunk (9 taps x NJ N-tiles = three ring slots) is ONE straight-line block: tap offsets, LDS addresses of the units and the
//     places of the side work are compile-time, the fragments of unit u + 1 are read in front of the MFMAs of unit u, and nothing
//     but MFMAs and LDS reads is issued between the slot synchronisations;
//   * the barrier that opens slot s + 1 stands in front of the LAST unit of slot s, so that the first fragments of the new slot are
//     read under that unit's 12 MFMAs; the weight pieces of slot s + 2 are issued right behind it (their ring slot was consumed a
//     whole slot ago);
//   * the input chunk's operand split + LDS store (6 items per thread) is handed out over the last units of slots 1 and 2, and the
//     epilogue of a completed stage runs at the top of the next chunk, behind that chunk's barrier.
constexpr int S4_A_IT = (SW_A_ROWS * 4 + 255) / 256;                // staging items (float4) per thread and chunk: 6
constexpr int S4_PIECES = (SW_SLOT_UNITS * 3 + 3) / 4;              // 1 KB LDS-DMA pieces per wave and slot: <= 7
static_assert(S4_PIECES <= 8 && S4_A_IT == 6, "the DMA / side-work plan: 8 DMA steps behind a synchronisation, 6 items per chunk");

template <int B, class F, int... I>
__device__ __forceinline__ void sw_static_for_seq(F &f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, B + I>{}), ...);
}
template <int B, int E, class F>
__device__ __forceinline__ void sw_static_for(F &&f) {
    sw_static_for_seq<B>(f, std::make_integer_sequence<int, (E > B ? E - B : 0)>{});
}

// One input chunk: N-tiles [J0, J0 + NJ) x 9 taps x 16 channels, three ring slots of 3 NJ units.  sync(s) opens slot s (1, 2).
// With one wave per SIMD an instruction costs nothing only while an MFMA the wave has already issued is still executing (32 cycles
// each), so everything else is handed out over the 12 MFMAs of a unit, a few instructions behind each:
//     MFMA 0 .. 2   the three planes of the next unit's weight fragment (ds_read_b128)
//     MFMA 3 .. 8   the six reads of the next tap's input fragments (units in front of a new tap)
//     MFMA 2, 5, 8, 11   one 1 KB piece of the weight DMA (the two units behind a slot synchronisation)
//     MFMA 4 .. 8   one step of an input-chunk item (hi plane, mid plane, lo plane + store, store, store: see the kernel)
// (profiles/r03t_sweep4_timeline.txt: issued in blocks between the units these cost 50 .. 100 cycles per plain unit, 150 per DMA
// unit and 520 per item against the unit's 384 cycles of matrix core.)
// DIRECT (the default since round 4): the weight fragments do not pass through LDS at all.  The stream is consumed strictly in
// order, one 3 KB unit per 12 MFMAs, and in the direct layout (sweep_pack_kernel) a plane of a unit is one coalesced 1 KB read in
// register order: every wave fetches the fragments of unit g + 2 straight from L2 / L1 into a three-deep register ring behind the first
// three MFMAs of unit g (`b_fetch`; the four waves of a CU ask for the same lines within a few hundred cycles: one L2 read, three L1
// hits).  No weight ring in LDS, no LDS-DMA pieces (60 cycles of issue each: +230 cycles on 2 units of every slot), no slot
// synchronisations (two per chunk at ~550 cycles) -- profiles/r03ap_sweep4_units.txt priced those at 14 % of the kernel.
template <int J0, int NJ, bool DIRECT, class Sync, class Item, class Dma, class Tick, class BFetch>
__device__ __forceinline__ void sweep4_chunk(f32x16 (&acc)[2][SW_NTILE], const float *sa, const float *s_b_lane, const int (&apix0)[2],
                                             const int half, Sync &&sync, Item &&item_step, Dma &&dma_step, Tick &&tick,
                                             tnr_bf16x8 (&fbr)[3][3], BFetch &&b_fetch) {
    static_assert((9 * NJ) % 3 == 0, "a chunk starts at ring position 0");
    constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};      // the six kept partial products, smallest first
    constexpr int NU = 9 * NJ, SU = 3 * NJ;
    tnr_bf16x8 fa[2][2][3], fb[2][3];
    const float *a_src[2][2];          // [tap parity][M-tile]: LDS address of the tap's fragment rows
    auto addr_a = [&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value, dy = t / 3, dx = t % 3;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int pp = apix0[m] + dy * SW_WT + dx;
            a_src[t & 1][m] = sa + pp * SW_ROW + 4 * (half ^ ((pp >> TNR_X3_SWZ) & 1));
        }
    };
    auto read_a = [&](auto tc, int m, int sp) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        fa[t & 1][m][sp] = *reinterpret_cast<const tnr_bf16x8 *>(a_src[t & 1][m] + 8 * sp);
    };
    auto read_b = [&](auto uc, int sp) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value, sl = u / SU, i = u % SU;
        fb[u & 1][sp] = *reinterpret_cast<const tnr_bf16x8 *>(s_b_lane + sl * SW_SLOT_FLOATS + i * SW_UNIT_FLOATS + 8 * sp);
    };
    addr_a(std::integral_constant<int, 0>{});
#pragma unroll
    for (int k = 0; k < 6; ++k) read_a(std::integral_constant<int, 0>{}, k / 3, k % 3);
    if constexpr (!DIRECT) {
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) read_b(std::integral_constant<int, 0>{}, sp);
    }
    sw_static_for<0, NU>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value, t = u / NJ, jj = u % NJ, sl = u / SU;
#ifdef SW_TIMELINE
        const unsigned long long tu0 = __builtin_amdgcn_s_memtime();
#endif
        constexpr bool SYNC_UNIT = !DIRECT && (u % SU == SU - 1 && sl < 2);
#ifndef S4_DMA_UNITS
#define S4_DMA_UNITS 2      /* units behind a synchronisation that carry DMA pieces (with the synchronisation unit itself) */
#endif
#ifndef S4_DMA_EVERY
#define S4_DMA_EVERY 3      /* one piece behind every n-th MFMA of such a unit */
#endif
        constexpr bool DMA_UNIT = !DIRECT && ((u % SU) < S4_DMA_UNITS || SYNC_UNIT);
        constexpr bool NEXT_TAP = u + 1 < NU && (u + 1) % NJ == 0;
        // input-chunk item i: behind the last units of slots 1 and 2 (NJ = 1: one per unit from unit 3)
        constexpr int ITEM = NJ == 1 ? u - 3 : (u >= 2 * SU - 4 && u <= 2 * SU - 2 ? u - (2 * SU - 4) : (u >= 3 * SU - 3 ? u - (3 * SU - 3) + 3 : -1));
        if constexpr (SYNC_UNIT) sync(std::integral_constant<int, sl + 1>{});
        if constexpr (NEXT_TAP) addr_a(std::integral_constant<int, (t + 1 < 9 ? t + 1 : 8)>{});
        __builtin_amdgcn_sched_barrier(0);
        sw_static_for<0, 12>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value, p = i / 2, m = i % 2;
            if constexpr (DIRECT) {
                acc[m][J0 + jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t & 1][m][TA[p]], fbr[u % 3][TB[p]], acc[m][J0 + jj], 0, 0, 0);
                if constexpr (i < 3) b_fetch(std::integral_constant<int, (u + 2) % 3>{}, i);       // unit g + 2 into the slot unit g - 1 left
            } else {
                acc[m][J0 + jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t & 1][m][TA[p]], fb[u & 1][TB[p]], acc[m][J0 + jj], 0, 0, 0);
                if constexpr (u + 1 < NU && i < 3) read_b(std::integral_constant<int, (u + 1 < NU ? u + 1 : u)>{}, i);
            }
            if constexpr (NEXT_TAP && i >= 3 && i < 9) read_a(std::integral_constant<int, (t + 1 < 9 ? t + 1 : 8)>{}, (i - 3) / 3, (i - 3) % 3);
            if constexpr (DMA_UNIT && i % S4_DMA_EVERY == S4_DMA_EVERY - 1) dma_step();
            if constexpr (ITEM >= 0 && ITEM < S4_A_IT && i >= 4 && i < 9) item_step(std::integral_constant<int, (ITEM >= 0 && ITEM < S4_A_IT ? ITEM : 0)>{}, std::integral_constant<int, (i >= 4 && i < 9 ? i - 4 : 0)>{});
            __builtin_amdgcn_sched_barrier(0);
        });
#ifdef SW_TIMELINE
        if constexpr (NJ == 3 || NJ == 2) {
            // class: 0 unit with the slot sync, 1 first two units of a slot (DMA), 2 units with an input-chunk item, 3 plain (NJ = 2: classes 4 .. 7)
            constexpr int cls = SYNC_UNIT ? 0 : ((u % SU) <= 1 ? 1 : (ITEM >= 0 && ITEM < S4_A_IT ? 2 : 3));
            tick(cls + (NJ == 2 ? 4 : 0), __builtin_amdgcn_s_memtime() - tu0);
        }
#endif
    });
}

// The same chunk in TNR_MMA_BF16 (`use_amp`: operands ROUNDED to bf16, base_model.py:736-744): one product per (tap, N-tile, M-tile), so a
// unit is 2 MFMAs (64 cycles of matrix core) and only plane 0 of either operand is used -- the weight stream's hi plane IS the
// round-to-nearest bf16 weight, the input tile's hi plane the rounded activation (the stager skips the other two).  With units this
// short the fragment ring is nine deep (unit g + 8 is fetched behind the first MFMA of unit g: ~500 cycles of look-ahead, an L2 hit),
// one whole input-chunk item (convert + one 8-byte LDS store) rides behind the second MFMA of the chunk's last units.
template <int J0, int NJ, class Item, class BFetch>
__device__ __forceinline__ void amp4_chunk(f32x16 (&acc)[2][SW_NTILE], const float *sa, const int (&apix0)[2], const int half,
                                           Item &&item, tnr_bf16x8 (&fbq)[9], BFetch &&b_fetch) {
    constexpr int NU = 9 * NJ;
    static_assert(NU % 9 == 0, "a chunk starts at ring position 0");
    tnr_bf16x8 fa[2][2];               // [tap parity][M-tile]
    const float *a_src[2][2];
    auto addr_a = [&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value, dy = t / 3, dx = t % 3;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int pp = apix0[m] + dy * SW_WT + dx;
            a_src[t & 1][m] = sa + pp * SW_ROW + 4 * (half ^ ((pp >> TNR_X3_SWZ) & 1));
        }
    };
    addr_a(std::integral_constant<int, 0>{});
    fa[0][0] = *reinterpret_cast<const tnr_bf16x8 *>(a_src[0][0]);
    fa[0][1] = *reinterpret_cast<const tnr_bf16x8 *>(a_src[0][1]);
    sw_static_for<0, NU>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value, t = u / NJ, jj = u % NJ;
        constexpr bool NEXT_TAP = u + 1 < NU && (u + 1) % NJ == 0;
        constexpr int ITEM = u - (NU - S4_A_IT);               // the chunk's last six units carry the next chunk's six items
        if constexpr (NEXT_TAP) addr_a(std::integral_constant<int, (t + 1 < 9 ? t + 1 : 8)>{});
        __builtin_amdgcn_sched_barrier(0);
        acc[0][J0 + jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t & 1][0], fbq[u % 9], acc[0][J0 + jj], 0, 0, 0);
        b_fetch(std::integral_constant<int, (u + 8) % 9>{});
        if constexpr (NEXT_TAP) fa[(t + 1) & 1][0] = *reinterpret_cast<const tnr_bf16x8 *>(a_src[(t + 1) & 1][0]);
        __builtin_amdgcn_sched_barrier(0);
        acc[1][J0 + jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t & 1][1], fbq[u % 9], acc[1][J0 + jj], 0, 0, 0);
        if constexpr (NEXT_TAP) fa[(t + 1) & 1][1] = *reinterpret_cast<const tnr_bf16x8 *>(a_src[(t + 1) & 1][1]);
        if constexpr (ITEM >= 0 && ITEM < S4_A_IT) item(std::integral_constant<int, (ITEM >= 0 && ITEM < S4_A_IT ? ITEM : 0)>{});
        __builtin_amdgcn_sched_barrier(0);
    });
}

#ifndef S4D_EPI_AFTER
#define S4D_EPI_AFTER 1       /* direct form: a stage's epilogue behind its pass's chunk loop (0: under `ck == 0` at the top of the next pass) */
#endif
#ifndef S4D_ALOAD_ALWAYS
#define S4D_ALOAD_ALWAYS 1    /* direct form: the next chunk's input loads on every path (0: only when there is a next chunk) */
#endif
template <bool DIRECT, bool AMP = false>
__global__ void __launch_bounds__(256, 1) conv_sweep4_kernel(const SweepK c) {
    static_assert(DIRECT || !AMP, "the bf16-operand form exists as a direct form only");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_a = smem, *s_b = smem + 2 * SW_A_FLOATS;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, half = lane >> 5;
    // Tiles are DISPENSED, not dealt: a workgroup takes the next tile from an atomic counter (one for the whole grid by default;
    // TNR_SWEEP_DISPENSERS=8: one per XCD over contiguous shares, so that the halo re-reads of neighbouring tiles meet in one L2 -- no
    // faster in the measurements).  With the static deal (tile = b + grid * k) the hand-off needs EVERY workgroup of the grid resident:
    // with one CU held by another kernel (tools/probes/sweep_hog.py, profiles/r03ad_sweep_hog*.txt) the eight-wave form ran 4.0x
    // slower and this form 1.5x with per-XCD shares, 1.10x with one dispenser -- the tiles of the missing workgroup stall all their
    // neighbours until a CU comes free.  Dispensed in order, a tile's neighbours are always taken by workgroups that ARE running; a
    // handful of resident workgroups is enough for progress, and a CU (or an XCD) that runs slower simply takes fewer tiles.
    const int g = gridDim.x;
    const bool xcd_split = (g & 7) == 0 && c.nxcd == 8;
    const int xcd = xcd_split ? ((int)blockIdx.x & 7) : 0, per_x = xcd_split ? (g >> 3) : g;
    __shared__ int s_next_tile;
    auto next_tile = [&]() __attribute__((always_inline)) {
        if (tid == 0) {
            const int slot = (int)atomicAdd(c.disp + xcd, 1u);
            const int kk = slot / per_x, ii = slot - kk * per_x;
            s_next_tile = xcd * per_x + ii + g * kk;
        }
        __syncthreads();
        return __builtin_amdgcn_readfirstlane(s_next_tile);
    };
    const ConvK &x0 = c.st[0];
    const __amdgpu_buffer_rsrc_t x_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x0.x), 0, (int)((unsigned)x0.N * x0.H * x0.W * x0.x_ct * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(c.wq), 0, c.wq_bytes, 0x00020000);
    sw_lds_float *ring = (sw_lds_float *)s_b;
    const int apix0[2] = {(2 * wave) * SW_WT + li, (2 * wave + 1) * SW_WT + li};
    const float *s_b_lane = s_b + li * SW_ROW + 4 * (half ^ ((li >> TNR_X3_SWZ) & 1));
    int pend_tile = -1;          // tile whose newest stage output still has to be published (wave-uniform)
    unsigned pend_value = 0;
    int pend_age = 0;
#ifdef SW_TIMELINE
    unsigned long long tl[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tl2[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tl_begin = __builtin_amdgcn_s_memtime();
#endif

    for (int tile = next_tile(); tile < c.tiles; tile = next_tile()) {
        SW_T(t_tile0);
        const int n = tile / c.tpi, rem = tile - n * c.tpi;
        const int ty = rem / c.tiles_x, tx = rem - ty * c.tiles_x;
        const int ty0 = ty * SW_TH, tx0 = tx * SW_TW;
        // ---- staging plan of the input tile: item i = tid + 256 it -> halo pixel i / 4, channel quad i % 4
        int in_off[S4_A_IT], a_dst[S4_A_IT];
#pragma unroll
        for (int it = 0; it < S4_A_IT; ++it) {
            const int i = tid + it * 256, row = i >> 2, q = i & 3;
            const int hr = row / SW_WT, hc = row - hr * SW_WT;
            const int Y = ty0 + hr - 1, X = tx0 + hc - 1;
            const bool in = (row < SW_A_ROWS) & (Y >= 0) & (Y < x0.H) & (X >= 0) & (X < x0.W);
            in_off[it] = in ? (((n * x0.H + Y) * x0.W + X) * x0.x_ct + x0.x_co + q * 4) : -1;
            a_dst[it] = row * SW_ROW + 4 * ((q >> 1) ^ ((row >> TNR_X3_SWZ) & 1)) + 2 * (q & 1);        // (row < SW_A_ALLOC_ROWS: always a valid address)
        }
        static_assert(S4_A_IT * 256 <= SW_A_ALLOC_ROWS * 4, "a row per staging item");
        f32x4 rin[S4_A_IT];
        auto a_load = [&](int ch, bool valid = true) __attribute__((always_inline)) {          // system-coherent: the channels may have been written by another CU in this launch
#pragma unroll
            for (int it = 0; it < S4_A_IT; ++it) {
                const unsigned bo = (valid && in_off[it] >= 0) ? (unsigned)(in_off[it] + ch) * 4u : 0xfffffff0u;      // (past the end: the range check returns 0)
                rin[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (int)bo, 0, TNR_COH_LOAD_AUX));
            }
        };
        // item `it` in five steps (see sweep4_chunk), each over the item's four channels at once so that no instruction waits for the one
        // in front of it: 0 hi plane and first residual, 1 mid plane and second residual, 2 lo plane + store hi, 3 store mid, 4 store lo
        // (tnr_split4_bf16x3's arithmetic)
        tnr_bf16x4 ih, im, il;
        f32x4 ir;
#ifndef S4_PK_SPLIT
#define S4_PK_SPLIT 1
#endif
        auto item_step_buf = [&](auto ic, auto kc, int buf) __attribute__((always_inline)) {
            constexpr int it = decltype(ic)::value, k = decltype(kc)::value;
#ifdef S4_ABL_NOSPLIT       /* (ablation build, results invalid: the upper bound of what PRE-SPLIT block buffers could save -- no operand split in the stager, the three LDS stores stay) */
            if constexpr (k >= 2) {
                float *dst = s_a + buf * SW_A_FLOATS + a_dst[it] + 8 * (k - 2);
                *reinterpret_cast<tnr_f32x2 *>(dst) = tnr_f32x2{rin[it][k - 2], rin[it][3]};
            }
#elif S4_PK_SPLIT           /* the same split with ONE packed conversion per channel pair and level (tnr_split4_bf16x3_pk's form, conv_body.h): 22 instead of 30 vector instructions per item, bit-identical; -0.3 % forward / -0.8 % gradient launch (profiles/r09o_sweep_upper_bounds.txt) */
            if constexpr (k == 0) {
                tnr_pk_level(rin[it], ih, ir);
            } else if constexpr (k == 1) {
                const f32x4 r1 = ir;
                tnr_pk_level(r1, im, ir);
            } else {
                if constexpr (k == 2) {
                    const tnr_f32x2 a = {ir[0], ir[1]}, b = {ir[2], ir[3]};
                    il = __builtin_bit_cast(tnr_bf16x4, tnr_f32x2{__builtin_bit_cast(float, __builtin_convertvector(a, tnr_bf16x2)),
                                                                __builtin_bit_cast(float, __builtin_convertvector(b, tnr_bf16x2))});
                }
                float *dst = s_a + buf * SW_A_FLOATS + a_dst[it] + 8 * (k - 2);
                *reinterpret_cast<tnr_f32x2 *>(dst) = __builtin_bit_cast(tnr_f32x2, k == 2 ? ih : (k == 3 ? im : il));
            }
#else
            if constexpr (k == 0) {
                ih = __builtin_convertvector(rin[it], tnr_bf16x4);
                ir = rin[it] - __builtin_convertvector(ih, f32x4);
            } else if constexpr (k == 1) {
                im = __builtin_convertvector(ir, tnr_bf16x4);
                ir = ir - __builtin_convertvector(im, f32x4);
            } else {
                if constexpr (k == 2) il = __builtin_convertvector(ir, tnr_bf16x4);
                float *dst = s_a + buf * SW_A_FLOATS + a_dst[it] + 8 * (k - 2);
                *reinterpret_cast<tnr_f32x2 *>(dst) = __builtin_bit_cast(tnr_f32x2, k == 2 ? ih : (k == 3 ? im : il));
            }
#endif
        };
        auto a_store_item = [&](auto ic, int buf) __attribute__((always_inline)) {
            sw_static_for<0, 5>([&](auto kc) __attribute__((always_inline)) { item_step_buf(ic, kc, buf); });
        };
        // weights: LDS-DMA, no registers; this wave's 1 KB pieces of a slot (piece q = bytes [1024 q, 1024 q + 1024), round-robin)
        auto b_issue = [&](const SweepCur &cu, int slot) __attribute__((always_inline)) {
            const int pieces = sw_nj_rt(cu.p) * 9;
            const int src0 = cu.unit * (SW_UNIT_FLOATS * 4);          // wave-uniform: the load's scalar offset
            sw_lds_float *dst = ring + slot * SW_SLOT_FLOATS;
#pragma unroll
            for (int i = 0; i < S4_PIECES; ++i) {
                const int q = wave + 4 * i;
                if (q < pieces) __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, dst + q * 256, 16, lane * 16, src0 + q * 1024, 0, 0);
            }
        };
        auto cur_advance = [&](SweepCur &cu) __attribute__((always_inline)) {                          // one slot = three taps of the pass's N-tiles
            cu.unit += 3 * sw_nj_rt(cu.p);
            if (++cu.sl == 3) {
                cu.sl = 0;
                if (++cu.ck == sw_nchunks(cu.p, c.nck0)) {
                    cu.ck = 0;
                    ++cu.p;
                }
            }
        };

        f32x16 acc[2][SW_NTILE];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < SW_NTILE; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;

        // ---- prologue: input chunk 0 into LDS, weight slot 0 on its way
        SweepCur ld{0, 0, 0, 0};         // the next slot to fetch
        a_load(0);
        __syncthreads();                 // the previous tile's last fragments are consumed
        // direct form: register ring of weight fragments, unit `bq` is the next to fetch (wave-uniform); units 0 and 1 go out now
        tnr_bf16x8 fbr[3][3];
        int bq = 0;
        auto b_fetch = [&](auto rc, int sp) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            fbr[r][sp] = __builtin_bit_cast(tnr_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rs, lane * 16, (bq * 3 + sp) * 1024, 0));      // (past the end of the stream: zeros)
            if (sp == 2) ++bq;
        };
        // bf16-operand form: nine single-plane fragments (plane 0 of unit `bq`), units 0 .. 7 go out now
        tnr_bf16x8 fbq[9];
        auto b_fetch1 = [&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            fbq[r] = __builtin_bit_cast(tnr_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rs, lane * 16, bq * 3072, 0));
            ++bq;
        };
        auto a_item1 = [&](auto ic, int buf) __attribute__((always_inline)) {      // a whole item: round to bf16, store plane 0
            constexpr int it = decltype(ic)::value;
            const tnr_bf16x4 h = __builtin_convertvector(rin[it], tnr_bf16x4);
            *reinterpret_cast<tnr_f32x2 *>(s_a + buf * SW_A_FLOATS + a_dst[it]) = __builtin_bit_cast(tnr_f32x2, h);
        };
        if constexpr (AMP) {
            sw_static_for<0, 8>([&](auto rc) __attribute__((always_inline)) { b_fetch1(rc); });
            sw_static_for<0, S4_A_IT>([&](auto ic) __attribute__((always_inline)) { a_item1(ic, 0); });
        } else if constexpr (DIRECT) {
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) b_fetch(std::integral_constant<int, 0>{}, sp);
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) b_fetch(std::integral_constant<int, 1>{}, sp);
        } else {
            b_issue(ld, 0);
            cur_advance(ld);
        }
        if constexpr (!AMP) sw_static_for<0, S4_A_IT>([&](auto ic) __attribute__((always_inline)) { a_store_item(ic, 0); });
        { SW_T(t_pro); SW_ADD(0, t_pro - t_tile0); }
        int e = 0;                       // input chunks consumed so far in this tile
        bool has_next = false;
        auto tick = [&](int cls, unsigned long long dt) __attribute__((always_inline)) {
#ifdef SW_TIMELINE
            tl2[cls] += dt;
            tl2[8 + cls] += 1;
#endif
        };
        // weight DMA in progress (wave-uniform): pieces dq, dq + 4, .. < dpieces of the slot at stream offset dsrc go to LDS at dbase
        int dq = 0, dpieces = 0, dsrc = 0;
        sw_lds_float *dbase = ring;
        auto dma_step = [&]() __attribute__((always_inline)) {
            if (dq < dpieces) {
#ifndef S4_ABL_NODMA        /* (ablation builds: timing only, results invalid) */
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, dbase + dq * 256, 16, lane * 16, dsrc + dq * 1024, 0, 0);
#endif
                dq += 4;
            }
        };

        // slot synchronisation: the pieces of the slot about to be read have landed for every wave (every piece was issued at least
        // 2/3 of a slot ago; vector-memory operations complete in order, so waiting for all of them also covers the input-chunk loads
        // issued a slot ago); the ring slot `fill` is free and its DMA is set up -- the pieces are issued from inside the next two units
        auto slot_sync = [&](int fill) __attribute__((always_inline)) {
            if constexpr (DIRECT) {
                // chunk top of the direct form: only the input tile's double buffer is synchronised (LDS stores of every wave done,
                // the other buffer consumed) -- the weight prefetch stays in flight across the barrier.  A pending stage output is
                // published at the second chunk top after its epilogue (a whole chunk later: its stores have drained, vmcnt(0) costs the
                // two prefetched units).
                const bool pub = pend_tile >= 0 && ++pend_age >= (S4D_EPI_AFTER ? 2 : 1);
                SW_T(t_s0);
                if (pub) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                SW_T(t_s2);
                SW_ADD(4, t_s2 - t_s0);
                if (pub) {
                    if (tid == 0) __hip_atomic_store(c.progress + pend_tile, pend_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pend_tile = -1;
                }
                return;
            }
            const bool pub = pend_tile >= 0 && ++pend_age >= 3;
            SW_T(t_s0);
            while (dq < dpieces) dma_step();         // (never taken: every slot has room for its pieces)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            SW_T(t_s1);
            __syncthreads();
            SW_T(t_s2);
            SW_ADD(3, t_s1 - t_s0);
            SW_ADD(4, t_s2 - t_s1);
            if (pub) {
                if (tid == 0) __hip_atomic_store(c.progress + pend_tile, pend_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pend_tile = -1;
            }
            if (ld.valid()) {
                dq = wave;
                dpieces = sw_nj_rt(ld.p) * 9;
                dsrc = ld.unit * (SW_UNIT_FLOATS * 4);
                dbase = ring + fill * SW_SLOT_FLOATS;
                cur_advance(ld);
            } else {
                dpieces = 0;
            }
        };
        // (the stage index is a compile-time constant at every accumulator access: a run-time switch or select over the array elements
        //  is rewritten into one load at a computed address and the whole accumulator array then lives in scratch memory)
        auto epilogue_of = [&](auto sc) __attribute__((always_inline)) {
            constexpr int S = decltype(sc)::value;
            SW_T(t_e0);
            const ConvK a = sweep_stage(S);
            const __amdgpu_buffer_rsrc_t y_rs =
                __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((unsigned)a.N * a.Ho * a.Wo * a.y_ct * 4u), 0x00020000);
            if constexpr (S < 4) {
                f32x16 t[2][1];
                t[0][0] = acc[0][S];
                t[1][0] = acc[1][S];
                conv_epilogue_dpp<TNR_CONV_3x3, SW_TW, 1, 2, true, 8, false>(a, t, 0, n, ty0, tx0, 0, wave, li, half, y_rs);
            } else {
                f32x16 t[2][2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    t[m][0] = acc[m][4];
                    t[m][1] = acc[m][5];
                }
                conv_epilogue_dpp<TNR_CONV_3x3, SW_TW, 2, 2, true, 8>(a, t, 0, n, ty0, tx0, 0, wave, li, half, y_rs);
            }
            pend_tile = tile;
            pend_value = c.base + (unsigned)S + 1u;
            pend_age = 0;
            { SW_T(t_e1); SW_ADD(8, t_e1 - t_e0); }
        };
        auto sync = [&](auto sc) __attribute__((always_inline)) {        // in front of the last unit of slot s - 1: opens slot s, refills slot s + 1
            constexpr int sl = decltype(sc)::value;
            slot_sync((sl + 1) % SW_RING);
        };
        auto item_step = [&](auto ic, auto kc) __attribute__((always_inline)) {
            if (has_next) item_step_buf(ic, kc, (e + 1) & 1);
        };

        // The nine passes are unrolled in the source: every accumulator access and every stage index is a compile-time constant, and
        // the control flow the register allocator sees is a plain sequence of chunk loops.
        sw_static_for<0, SW_NPASS>([&](auto pc) __attribute__((always_inline)) {
            constexpr int p = decltype(pc)::value;
            const int nchunks = sw_nchunks(p, c.nck0), ch_lo = sw_ch_lo(p, c.nck0);
#pragma unroll 1
            for (int ck = 0; ck < nchunks; ++ck) {
                // ---- chunk top: slot 0 of this chunk has landed, the chunk's input tile is in LDS
                slot_sync(1);
                if constexpr (!sw_apass(p) && !(DIRECT && S4D_EPI_AFTER)) {        // the stage completed by pass a of this phase (this pass never touches its accumulators)
                    if (ck == 0) epilogue_of(std::integral_constant<int, sw_phase(p)>{});
                }
                const bool last_ck = ck + 1 == nchunks;
                has_next = !(last_ck && p == SW_NPASS - 1);
                int ch_keep = 0;
                if (has_next) {
                    int ch_next = ch_lo + 16 * (ck + 1);
                    if (last_ck) {
                        ch_next = sw_ch_lo(p + 1 < SW_NPASS ? p + 1 : p, c.nck0);
                        if constexpr (p + 1 < SW_NPASS && sw_apass(p + 1)) {
                            // the first chunk of the next phase is the output of the stage this tile and its 8 neighbours finished
                            // in pass a of the current phase
                            const ChainWait w{c.progress, c.base + (unsigned)sw_phase(p + 1), n, ty, tx, c.tiles_x, c.tiles_y, c.err,
                                              &pend_tile, pend_value};
                            SW_T(t_w0);
                            w();
                            { SW_T(t_w1); SW_ADD(1, t_w1 - t_w0); }
                        }
                    }
                    if constexpr (!(DIRECT && S4D_ALOAD_ALWAYS)) {
                        SW_T(t_al0);
                        a_load(ch_next);
                        { SW_T(t_al1); SW_ADD(2, t_al1 - t_al0); }
                    } else {
                        ch_keep = ch_next;
                    }
                }
                if constexpr (DIRECT && S4D_ALOAD_ALWAYS) {
                    // the loads are issued on EVERY path (past-the-end addresses when there is no next chunk): with a branch around
                    // them the compiler has to assume the shorter queue in front of the first fragment use and waits for these loads too
                    SW_T(t_al0);
                    a_load(ch_keep, has_next);
                    { SW_T(t_al1); SW_ADD(2, t_al1 - t_al0); }
                }
                const float *sa = s_a + (e & 1) * SW_A_FLOATS;
                SW_T(t_c0);
                if constexpr (AMP) {
                    auto item1 = [&](auto ic) __attribute__((always_inline)) {
                        if (has_next) a_item1(ic, (e + 1) & 1);
                    };
                    amp4_chunk<sw_j0(p), sw_nj(p)>(acc, sa, apix0, half, item1, fbq, b_fetch1);
                } else {
                    sweep4_chunk<sw_j0(p), sw_nj(p), DIRECT>(acc, sa, s_b_lane, apix0, half, sync, item_step, dma_step, tick, fbr, b_fetch);
                }
                { SW_T(t_c1); SW_ADD(6, t_c1 - t_c0); SW_ADD(15, 1ull); SW_ADD(9 + (sw_nj(p) == 3 ? 0 : (sw_nj(p) == 2 ? 1 : 2)), t_c1 - t_c0); }
                ++e;
            }
            if constexpr (DIRECT && S4D_EPI_AFTER && sw_apass(p) && p + 1 < SW_NPASS) {
                // direct form: the epilogue of the stage this pass completed stands BEHIND the pass's chunk loop, on every path -- under
                // `if (ck == 0)` at the top of the next pass the compiler must assume, at the join, that the fragment ring is the
                // youngest thing in the memory queue and makes the first unit wait for the epilogue's stores.  Published two chunk tops
                // later (slot_sync), as before: a whole chunk after the stores were issued.
                epilogue_of(std::integral_constant<int, sw_phase(p)>{});
            }
        });
        epilogue_of(std::integral_constant<int, 4>{});      // the last stage (pass 8 is a pass a)
    }
    // nothing in this launch waits for the last stage; publish it anyway so the counters stay consistent
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (pend_tile >= 0 && tid == 0) __hip_atomic_store(c.progress + pend_tile, pend_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0 && atomicAdd(c.disp + 8, 1u) == (unsigned)g - 1u) {      // the last workgroup out leaves the dispensers at zero for the next launch
#pragma unroll
        for (int i = 0; i < 9; ++i) __hip_atomic_store(c.disp + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef SW_TIMELINE
    if (tid == 0) {
        tl[14] = __builtin_amdgcn_s_memtime() - tl_begin;
        for (int i = 0; i < 16; ++i) atomicAdd(&sw_tl[i], tl[i]);
        for (int i = 0; i < 16; ++i) atomicAdd(&sw_tl2[i], tl2[i]);
    }
#endif
}

// =====================================================================================================================================
// A plain 3x3 convolution / data-gradient on the machinery of the direct sweep ("d4"; TNR_MMA_BF16X3, Cout % 64 == 0, Cin % 16 == 0).
// The 8-wave kernel of conv_x3w8.h splits and stages the 9 x 64 x 16 weight slab through LDS in every workgroup at every chunk, in a
// phase of its own between two barriers (the matrix core idles meanwhile: MFMA busy 0.62).  Here the weights come as a pre-split stream
// (tnr_conv_wq_pack: units of one tap x one 32-cout N-tile x 16 channels, 3 KB, plane by plane in register order) straight from L2 into a
// register ring, the input tile is double-buffered in LDS and its split + store is handed out behind the MFMAs -- sweep4_chunk<0, 2, true>,
// the chunk body of the dense-block sweep with two N-tiles: one barrier per chunk, nothing but MFMAs and side work between them.
// Tile 8 x 32 pixels x 64 output channels, four waves (wave w: tile rows 2 w, 2 w + 1), OCC workgroups per CU (73.7 KB of LDS each):
// with two, one workgroup's epilogue / prologue / barrier runs under the other's MFMAs.  Arithmetic (split, kept products and their
// order, chunk and tap order, epilogue) is that of conv_tile_body<.., BF = 2> / conv3x3_x3w8_kernel: bit-identical results.
struct D4K {
    ConvK a;
    const float *wq;                   // [cb][chunk][tap][N-tile] units (sweep direct layout)
    int wq_bytes;
    int nck;                           // input chunks (Cin / 16)
    int tiles_x, tiles_y, ncb, tiles;
};

struct D4PackK {
    const float *wp;
    int KinP, KoutP, nck, ncb, units;
    int shuffle;                       // 2: the stream's cout r' = sub * (Cout / 4) + c holds conv channel 4 c + sub (tnr_conv_desc.shuffle)
    float *out;
};

__global__ void __launch_bounds__(256) d4_pack_kernel(const D4PackK a) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int unit = g >> 6, r = (g >> 1) & 31, h = g & 1;
    if (unit >= a.units) return;
    const int per_cb = a.nck * 18;
    const int cb = unit / per_cb, rem = unit - cb * per_cb;
    const int ck = rem / 18, tj = rem - ck * 18, tap = tj >> 1, j = tj & 1;
    int co = cb * 64 + j * 32 + r;
    if (a.shuffle == 2) {
        const int nfc = a.KoutP >> 2;
        co = (co % nfc) * 4 + co / nfc;
    }
    const float *src = a.wp + ((size_t)tap * a.KoutP + co) * a.KinP + 16 * ck + 8 * h;
    const f32x4 q0 = *reinterpret_cast<const f32x4 *>(src), q1 = *reinterpret_cast<const f32x4 *>(src + 4);
    tnr_bf16x8 pl[3];
    tnr_split_bf16x3(q0, q1, pl);
    float *dst = a.out + (size_t)unit * SW_UNIT_FLOATS + (h * 32 + r) * 4;
#pragma unroll
    for (int k = 0; k < 3; ++k) *reinterpret_cast<tnr_bf16x8 *>(dst + k * 256) = pl[k];
}

template <int OCC, bool AMP = false, bool PS = false>
__global__ void __launch_bounds__(256, OCC) conv3x3_d4_kernel(const D4K c) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_a = smem;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, half = lane >> 5;
    const ConvK &a = c.a;
    const __amdgpu_buffer_rsrc_t x_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)((unsigned)a.N * a.H * a.W * a.x_ct * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(c.wq), 0, c.wq_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rs =
        __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((unsigned)a.N * a.Ho * a.Wo * a.y_ct * 4u), 0x00020000);
    const int apix0[2] = {(2 * wave) * SW_WT + li, (2 * wave + 1) * SW_WT + li};

    for (int tile = blockIdx.x; tile < c.tiles; tile += gridDim.x) {
        // (cb innermost: the ncb workgroups of a pixel tile run side by side and share its input in L2)
        const int cb = tile % c.ncb;
        int rest = tile / c.ncb;
        const int tx = rest % c.tiles_x;
        rest /= c.tiles_x;
        const int ty = rest % c.tiles_y, n = rest / c.tiles_y;
        const int ty0 = ty * SW_TH, tx0 = tx * SW_TW;
        int in_off[S4_A_IT], a_dst[S4_A_IT];
#pragma unroll
        for (int it = 0; it < S4_A_IT; ++it) {
            const int i = tid + it * 256, row = i >> 2, q = i & 3;
            const int hr = row / SW_WT, hc = row - hr * SW_WT;
            int Y = ty0 + hr - 1, X = tx0 + hc - 1;
            if (a.reflect) {         // nn.ReflectionPad2d(1) in front of the layer: row / column -1 is 1, H / W is H - 2 / W - 2
                Y = Y == -1 ? 1 : (Y == a.H ? a.H - 2 : Y);
                X = X == -1 ? 1 : (X == a.W ? a.W - 2 : X);
            }
            const bool in = (row < SW_A_ROWS) & (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
            in_off[it] = in ? (((n * a.H + Y) * a.W + X) * a.x_ct + a.x_co + q * 4) : -1;
            a_dst[it] = row * SW_ROW + 4 * ((q >> 1) ^ ((row >> TNR_X3_SWZ) & 1)) + 2 * (q & 1);
        }
        f32x4 rin[S4_A_IT];
        auto a_load = [&](int ch, bool valid) __attribute__((always_inline)) {     // on every path (past-the-end addresses read zeros)
#pragma unroll
            for (int it = 0; it < S4_A_IT; ++it) {
                const unsigned bo = (valid && in_off[it] >= 0) ? (unsigned)(in_off[it] + ch) * 4u : 0xfffffff0u;
                rin[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (int)bo, 0, 0));
            }
        };
        tnr_bf16x4 ih, im, il;
        f32x4 ir;
        auto item_step_buf = [&](auto ic, auto kc, int buf) __attribute__((always_inline)) {      // (conv_sweep4_kernel: five steps per item)
            constexpr int it = decltype(ic)::value, k = decltype(kc)::value;
            if constexpr (k == 0) {
                tnr_pk_level(rin[it], ih, ir);           // (the packed-conversion form of the split: conv_sweep4_kernel's stager, bit-identical)
            } else if constexpr (k == 1) {
                const f32x4 r1 = ir;
                tnr_pk_level(r1, im, ir);
            } else {
                if constexpr (k == 2) {
                    const tnr_f32x2 a2 = {ir[0], ir[1]}, b2 = {ir[2], ir[3]};
                    il = __builtin_bit_cast(tnr_bf16x4, tnr_f32x2{__builtin_bit_cast(float, __builtin_convertvector(a2, tnr_bf16x2)),
                                                                __builtin_bit_cast(float, __builtin_convertvector(b2, tnr_bf16x2))});
                }
                float *dst = s_a + buf * SW_A_FLOATS + a_dst[it] + 8 * (k - 2);
                *reinterpret_cast<tnr_f32x2 *>(dst) = __builtin_bit_cast(tnr_f32x2, k == 2 ? ih : (k == 3 ? im : il));
            }
        };
        f32x16 acc[2][SW_NTILE];         // (only N-tiles 0 and 1 are ever touched: the rest is never materialised)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;

        a_load(0, true);
        __syncthreads();                 // the previous tile's last fragments are consumed
        tnr_bf16x8 fbr[3][3];
        int bq = cb * c.nck * 18;        // this channel block's part of the stream
        auto b_fetch = [&](auto rc, int sp) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            fbr[r][sp] = __builtin_bit_cast(tnr_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rs, lane * 16, (bq * 3 + sp) * 1024, 0));
            if (sp == 2) ++bq;
        };
        tnr_bf16x8 fbq[9];               // bf16-operand form (TNR_MMA_BF16): nine single-plane fragments, see amp4_chunk
        auto b_fetch1 = [&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            fbq[r] = __builtin_bit_cast(tnr_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rs, lane * 16, bq * 3072, 0));
            ++bq;
        };
        auto a_item1 = [&](auto ic, int buf) __attribute__((always_inline)) {
            constexpr int it = decltype(ic)::value;
            const tnr_bf16x4 h = __builtin_convertvector(rin[it], tnr_bf16x4);
            *reinterpret_cast<tnr_f32x2 *>(s_a + buf * SW_A_FLOATS + a_dst[it]) = __builtin_bit_cast(tnr_f32x2, h);
        };
        if constexpr (AMP) {
            sw_static_for<0, 8>([&](auto rc) __attribute__((always_inline)) { b_fetch1(rc); });
            sw_static_for<0, S4_A_IT>([&](auto ic) __attribute__((always_inline)) { a_item1(ic, 0); });
        } else {
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) b_fetch(std::integral_constant<int, 0>{}, sp);
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) b_fetch(std::integral_constant<int, 1>{}, sp);
            sw_static_for<0, S4_A_IT>([&](auto ic) __attribute__((always_inline)) {
                sw_static_for<0, 5>([&](auto kc) __attribute__((always_inline)) { item_step_buf(ic, kc, 0); });
            });
        }
        bool has_next = false;
        int e = 0;
        auto item_step = [&](auto ic, auto kc) __attribute__((always_inline)) {
            if (has_next) item_step_buf(ic, kc, (e + 1) & 1);
        };
        auto nop1 = [&](auto) __attribute__((always_inline)) {};
        auto nop0 = [&]() __attribute__((always_inline)) {};
        auto tick = [&](int, unsigned long long) __attribute__((always_inline)) {};
#pragma unroll 1
        for (int ck = 0; ck < c.nck; ++ck) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the chunk's input tile is in LDS; the other buffer is consumed
            has_next = ck + 1 < c.nck;
            a_load(16 * (ck + 1), has_next);
            if constexpr (AMP) {
                auto item1 = [&](auto ic) __attribute__((always_inline)) {
                    if (has_next) a_item1(ic, (e + 1) & 1);
                };
                amp4_chunk<0, 2>(acc, s_a + (e & 1) * SW_A_FLOATS, apix0, half, item1, fbq, b_fetch1);
            } else {
                sweep4_chunk<0, 2, true>(acc, s_a + (e & 1) * SW_A_FLOATS, nullptr, apix0, half, nop1, item_step, nop0, tick, fbr, b_fetch);
            }
            ++e;
        }
        f32x16 t[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            t[m][0] = acc[m][0];
            t[m][1] = acc[m][1];
        }
        // (two workgroups per CU: the other one's MFMAs cover this epilogue's load latency -- one unit of look-ahead keeps it in 256 registers)
        if constexpr (PS) {
            // nn.PixelShuffle(2) folded into the store (tnr_conv_desc.shuffle): the stream's 64-cout block cb holds the channels
            // c of ONE sub-pixel (dy, dx) = cb / (blocks per sub-pixel): 256-byte rows of the shuffled tensor, written where they belong
            ConvK ap = a;
            const int nfb = c.ncb >> 2, sub = cb / nfb;
            ap.Cout = a.Cout >> 2;
            ap.Ho = 2 * a.Ho;
            ap.Wo = 2 * a.Wo;
            ap.bias = a.bias != nullptr ? a.bias + sub : nullptr;
            conv_epilogue_dpp<TNR_DGRAD_4x4_S2, SW_TW, 2, 2, false, (OCC == 1 ? 8 : 1), false, 4>(ap, t, cb - sub * nfb, n, ty0, tx0, sub, wave, li, half, y_rs);
        } else {
            conv_epilogue_dpp<TNR_CONV_3x3, SW_TW, 2, 2, false, (OCC == 1 ? 8 : 1)>(a, t, cb, n, ty0, tx0, 0, wave, li, half, y_rs);
        }
    }
}

// =====================================================================================================================================
// The four-tap convolutions of the discriminators on the same machinery ("s2"; TNR_MMA_BF16X3 / TNR_MMA_BF16, Cout % 64 == 0, Cin % 16 == 0):
//   TNR_CONV_4x4_S2   k4 s2 p1 (discriminators.py:24-34) as 2 x 2 taps over the four parity planes of the input: chunk = (parity plane pp,
//                     16 channels), halo tile 9 x 33 positions of that plane;
//   TNR_DGRAD_4x4_S2  its data-gradient, one output parity class (py, px) per tile: 2 x 2 taps at positions (1 + py - ty, 1 + px - tx)
//                     of the 10 x 34 halo tile of the incoming gradient, stored to pixels (2 y + py, 2 x + px).
// conv_tile_kernel splits and stages a 4 x 64 x 16 weight slab through LDS per workgroup and chunk between two barriers; here the weights
// are a pre-split stream (tnr_conv_wq_pack: units of one tap x one 32-cout N-tile x 16 channels in register order) read straight from L2
// into a four-deep register ring, the input tile is double-buffered and its split + store rides behind the MFMAs: one barrier per chunk.
// Chunk, tap and product order are conv_tile_body's: bit-identical results.
constexpr int S2_TAPS = 4, S2_NU = S2_TAPS * 2, S2_RING = 4;
struct S2K {
    ConvK a;
    const float *wq;                   // forward: [cb][pp][chunk][tap][N-tile] units; gradient: [par][cb][chunk][tap][N-tile]
    int wq_bytes;
    int nck;                           // 16-channel chunks per parity plane (forward) / of the gradient's channels
    int tiles_x, tiles_y, ncb, tiles;
};
struct S2PackK {
    const float *wp;
    int KinP, KoutP, nck, ncb, units, dgrad;
    float *out;
};

__global__ void __launch_bounds__(256) s2_pack_kernel(const S2PackK a) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int unit = g >> 6, r = (g >> 1) & 31, h = g & 1;
    if (unit >= a.units) return;
    const int j = unit & 1, tap = (unit >> 1) & 3;
    int rest = unit >> 3;
    const float *src;
    if (a.dgrad) {       // [par][cb][ck]: packed slab [par][tap][KoutP][KinP]
        const int ck = rest % a.nck;
        rest /= a.nck;
        const int cb = rest % a.ncb, par = rest / a.ncb;
        src = a.wp + (((size_t)par * 4 + tap) * a.KoutP + cb * 64 + j * 32 + r) * a.KinP + 16 * ck + 8 * h;
    } else {             // [cb][pp][ck]: packed slab [tap][KoutP][4 KinP], parity plane pp = columns pp KinP ..
        const int ck = rest % a.nck;
        rest /= a.nck;
        const int pp = rest & 3, cb = rest >> 2;
        src = a.wp + ((size_t)tap * a.KoutP + cb * 64 + j * 32 + r) * (4 * a.KinP) + pp * a.KinP + 16 * ck + 8 * h;
    }
    const f32x4 q0 = *reinterpret_cast<const f32x4 *>(src), q1 = *reinterpret_cast<const f32x4 *>(src + 4);
    tnr_bf16x8 pl[3];
    tnr_split_bf16x3(q0, q1, pl);
    float *dst = a.out + (size_t)unit * SW_UNIT_FLOATS + (h * 32 + r) * 4;
#pragma unroll
    for (int k = 0; k < 3; ++k) *reinterpret_cast<tnr_bf16x8 *>(dst + k * 256) = pl[k];
}

// one input chunk: 4 taps x 2 N-tiles = 8 units of 12 MFMAs; tap t of M-tile m reads its fragment rows at LDS offset aaddr[m][t] (floats:
// row and slot swizzle of the SHIFTED row); everything else as in sweep4_chunk<.., DIRECT = true> with a four-deep fragment ring
// (8 units = two turns of it)
template <bool AMP, class Item, class BFetch>
__device__ __forceinline__ void tap4_chunk(f32x16 (&acc)[2][2], const float *sa, const int (&aaddr)[2][4], Item &&item_step,
                                           tnr_bf16x8 (&fbr)[S2_RING][3], BFetch &&b_fetch) {
    constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};      // the six kept partial products, smallest first
    tnr_bf16x8 fa[2][2][3];
    auto read_a = [&](auto tc, int m, int sp) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        fa[t & 1][m][sp] = *reinterpret_cast<const tnr_bf16x8 *>(sa + aaddr[m][t] + 8 * sp);
    };
#pragma unroll
    for (int k = 0; k < (AMP ? 2 : 6); ++k) read_a(std::integral_constant<int, 0>{}, AMP ? k : k / 3, AMP ? 0 : k % 3);
    sw_static_for<0, S2_NU>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value, t = u >> 1, jj = u & 1;
        constexpr bool NEXT_TAP = jj == 1 && t + 1 < S2_TAPS;
        constexpr int ITEM = u - (S2_NU - 6);           // the chunk's last six units carry the next chunk's (up to) six items
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (AMP) {
            sw_static_for<0, 2>([&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value;
                acc[m][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t & 1][m][0], fbr[u % S2_RING][0], acc[m][jj], 0, 0, 0);
                if constexpr (m == 0) b_fetch(std::integral_constant<int, (u + 3) % S2_RING>{}, 0);
                if constexpr (NEXT_TAP) read_a(std::integral_constant<int, (t + 1 < S2_TAPS ? t + 1 : t)>{}, m, 0);
                if constexpr (m == 1 && ITEM >= 0) item_step(std::integral_constant<int, (ITEM >= 0 ? ITEM : 0)>{}, std::integral_constant<int, -1>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        } else {
            sw_static_for<0, 12>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value, p = i / 2, m = i % 2;
                acc[m][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t & 1][m][TA[p]], fbr[u % S2_RING][TB[p]], acc[m][jj], 0, 0, 0);
                if constexpr (i < 3) b_fetch(std::integral_constant<int, (u + 3) % S2_RING>{}, i);       // unit g + 3 into the slot unit g - 1 left
                if constexpr (NEXT_TAP && i >= 3 && i < 9) read_a(std::integral_constant<int, (t + 1 < S2_TAPS ? t + 1 : t)>{}, (i - 3) / 3, (i - 3) % 3);
                if constexpr (ITEM >= 0 && i >= 4 && i < 9) item_step(std::integral_constant<int, (ITEM >= 0 ? ITEM : 0)>{}, std::integral_constant<int, (i >= 4 && i < 9 ? i - 4 : 0)>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    });
}

template <int MODE, bool AMP>
__global__ void __launch_bounds__(256, 2) conv_s2_d4_kernel(const S2K c) {
    constexpr bool DG = MODE == TNR_DGRAD_4x4_S2;
    constexpr int HT = DG ? SW_HT : SW_TH + 1, WT = DG ? SW_WT : SW_TW + 1, ROWS = HT * WT;      // halo tile of the gradient / of one parity plane
    constexpr int A_IT = (ROWS * 4 + 255) / 256;                                                 // 6 / 5 staging items per thread and chunk
    static_assert(A_IT <= 6 && A_IT * 256 <= SW_A_ALLOC_ROWS * 4, "staging plan");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_a = smem;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, half = lane >> 5;
    const ConvK &a = c.a;
    const __amdgpu_buffer_rsrc_t x_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)((unsigned)a.N * a.H * a.W * a.x_ct * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(c.wq), 0, c.wq_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rs =
        __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((unsigned)a.N * a.Ho * a.Wo * a.y_ct * 4u), 0x00020000);
    const int nchunks = DG ? c.nck : 4 * c.nck;

    for (int tile = blockIdx.x; tile < c.tiles; tile += gridDim.x) {
        // (cb -- and the gradient's parity class -- innermost: the workgroups of a pixel tile run side by side and share its input in L2)
        int rest = tile;
        const int cb = rest % c.ncb;
        rest /= c.ncb;
        int par = 0;
        if constexpr (DG) {
            par = rest & 3;
            rest >>= 2;
        }
        const int tx = rest % c.tiles_x;
        rest /= c.tiles_x;
        const int ty = rest % c.tiles_y, n = rest / c.tiles_y;
        const int ty0 = ty * SW_TH, tx0 = tx * SW_TW;
        const int py = par >> 1, px = par & 1;
        // tap t = (t >> 1, t & 1) sits at position (pos_y, pos_x) of the halo tile; fragment rows of this wave's two M-tiles (tile rows
        // 2 w, 2 w + 1): row (2 w + m + pos_y) WT + li + pos_x, slot swizzle of THAT row
        int aaddr[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int pos_y = DG ? 1 + py - (t >> 1) : (t >> 1), pos_x = DG ? 1 + px - (t & 1) : (t & 1);
                const int pp = (2 * wave + m + pos_y) * WT + li + pos_x;
                aaddr[m][t] = pp * SW_ROW + 4 * (half ^ ((pp >> TNR_X3_SWZ) & 1));
            }
        int in_off[A_IT], a_dst[A_IT];
        auto plan = [&](int pp) __attribute__((always_inline)) {        // forward: the parity plane pp of the input; gradient: pp = 0
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const int i = tid + it * 256, row = i >> 2, q = i & 3;
                const int hr = row / WT, hc = row - hr * WT;
                const int Y = DG ? ty0 + hr - 1 : 2 * (ty0 + hr) - 1 + (pp >> 1), X = DG ? tx0 + hc - 1 : 2 * (tx0 + hc) - 1 + (pp & 1);
                const bool in = (row < ROWS) & (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
                in_off[it] = in ? (((n * a.H + Y) * a.W + X) * a.x_ct + a.x_co + q * 4) : -1;
                a_dst[it] = row * SW_ROW + 4 * ((q >> 1) ^ ((row >> TNR_X3_SWZ) & 1)) + 2 * (q & 1);
            }
        };
        plan(0);
        f32x4 rin[A_IT];
        auto a_load = [&](int ch, bool valid) __attribute__((always_inline)) {     // on every path (past-the-end addresses read zeros)
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const unsigned bo = (valid && in_off[it] >= 0) ? (unsigned)(in_off[it] + ch) * 4u : 0xfffffff0u;
                rin[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (int)bo, 0, 0));
            }
        };
        tnr_bf16x4 ih, im, il;
        f32x4 ir;
        auto item_step_buf = [&](auto ic, auto kc, int buf) __attribute__((always_inline)) {      // (conv_sweep4_kernel: five steps per item; k = -1: a whole bf16-operand item)
            constexpr int it = decltype(ic)::value, k = decltype(kc)::value;
            if constexpr (it < A_IT) {
                if constexpr (k < 0) {
                    const tnr_bf16x4 hh = __builtin_convertvector(rin[it], tnr_bf16x4);
                    *reinterpret_cast<tnr_f32x2 *>(s_a + buf * SW_A_FLOATS + a_dst[it]) = __builtin_bit_cast(tnr_f32x2, hh);
                } else if constexpr (k == 0) {
                    tnr_pk_level(rin[it], ih, ir);       // (the packed-conversion form of the split, bit-identical)
                } else if constexpr (k == 1) {
                    const f32x4 r1 = ir;
                    tnr_pk_level(r1, im, ir);
                } else {
                    if constexpr (k == 2) {
                        const tnr_f32x2 a2 = {ir[0], ir[1]}, b2 = {ir[2], ir[3]};
                        il = __builtin_bit_cast(tnr_bf16x4, tnr_f32x2{__builtin_bit_cast(float, __builtin_convertvector(a2, tnr_bf16x2)),
                                                                    __builtin_bit_cast(float, __builtin_convertvector(b2, tnr_bf16x2))});
                    }
                    float *dst = s_a + buf * SW_A_FLOATS + a_dst[it] + 8 * (k - 2);
                    *reinterpret_cast<tnr_f32x2 *>(dst) = __builtin_bit_cast(tnr_f32x2, k == 2 ? ih : (k == 3 ? im : il));
                }
            }
        };
        f32x16 acc[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;

        a_load(0, true);
        __syncthreads();                 // the previous tile's last fragments are consumed
        tnr_bf16x8 fbr[S2_RING][3];
        int bq = (DG ? (par * c.ncb + cb) * c.nck : cb * 4 * c.nck) * S2_NU;        // this tile's part of the stream
        auto b_fetch = [&](auto rc, int sp) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            if constexpr (AMP) {
                fbr[r][0] = __builtin_bit_cast(tnr_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rs, lane * 16, bq * 3072, 0));
                ++bq;
            } else {
                fbr[r][sp] = __builtin_bit_cast(tnr_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rs, lane * 16, (bq * 3 + sp) * 1024, 0));
                if (sp == 2) ++bq;
            }
        };
        sw_static_for<0, 3>([&](auto rc) __attribute__((always_inline)) {
#pragma unroll
            for (int sp = 0; sp < (AMP ? 1 : 3); ++sp) b_fetch(rc, sp);
        });
        sw_static_for<0, 6>([&](auto ic) __attribute__((always_inline)) {
            if constexpr (AMP) item_step_buf(ic, std::integral_constant<int, -1>{}, 0);
            else sw_static_for<0, 5>([&](auto kc) __attribute__((always_inline)) { item_step_buf(ic, kc, 0); });
        });
        bool has_next = false;
        int e = 0;
        auto item_step = [&](auto ic, auto kc) __attribute__((always_inline)) {
            if (has_next) item_step_buf(ic, kc, (e + 1) & 1);
        };
#pragma unroll 1
        for (int ck = 0; ck < nchunks; ++ck) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the chunk's input tile is in LDS; the other buffer is consumed
            has_next = ck + 1 < nchunks;
            int ch_next = 16 * (ck + 1);
            if constexpr (!DG) {                     // forward: chunk = (parity plane, 16 channels); a new plane has its own addresses
                const int nx = ck + 1, ppn = nx / c.nck;
                ch_next = 16 * (nx - ppn * c.nck);
                if (has_next && ppn != ck / c.nck) plan(ppn);
            }
            a_load(ch_next, has_next);
            tap4_chunk<AMP>(acc, s_a + (e & 1) * SW_A_FLOATS, aaddr, item_step, fbr, b_fetch);
            ++e;
        }
        f32x16 t2[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            t2[m][0] = acc[m][0];
            t2[m][1] = acc[m][1];
        }
        conv_epilogue_dpp<MODE, SW_TW, 2, 2, false, 1>(a, t2, cb, n, ty0, tx0, par, wave, li, half, y_rs);
    }
}

bool s2_ok(const tnr_conv_desc *d) {
    if (d->mode != TNR_CONV_4x4_S2 && d->mode != TNR_DGRAD_4x4_S2) return false;
    const bool dg = d->mode == TNR_DGRAD_4x4_S2;
    const int tw = dg ? d->W : d->Wo, th = dg ? d->H : d->Ho;            // the tile space
    return (d->mma == TNR_MMA_BF16X3 || d->mma == TNR_MMA_BF16) && d->pad_mode == 0 && d->shuffle == 0 && (d->Cout % 64) == 0 && d->KoutP == d->Cout &&
           d->Cin == d->KinP && (d->Cin % 16) == 0 && d->Cin >= 32 && tw >= 32 && th >= 8 &&
           (dg ? (d->Ho == 2 * d->H && d->Wo == 2 * d->W) : (d->H == 2 * d->Ho && d->W == 2 * d->Wo)) &&
           (int64_t)d->N * d->H * d->W * d->x.ctot < (1LL << 30) && (int64_t)d->N * d->Ho * d->Wo * d->y.ctot < (1LL << 30) &&
           (int64_t)(d->Cout / 64) * (d->Cin / 16) * 4 * S2_NU * SW_UNIT_FLOATS * (int64_t)sizeof(float) < (1LL << 31);
}

bool d4_ok(const tnr_conv_desc *d) {
    if (d->shuffle != 0 && !(d->shuffle == 2 && (d->Cout % 256) == 0 && d->r1.ptr == nullptr && d->r2.ptr == nullptr && d->m.ptr == nullptr &&
                             d->noise_pos == 0 && d->pad_mode == 0 && (d->y.ctot % 4) == 0 &&
                             (int64_t)d->N * d->H * d->W * 4 * d->y.ctot < (1LL << 30)))
        return false;
    return d->mode == TNR_CONV_3x3 && (d->mma == TNR_MMA_BF16X3 || d->mma == TNR_MMA_BF16) && (d->pad_mode == 0 || (d->pad_mode == 1 && d->H >= 2 && d->W >= 2)) &&
           (d->Cout % 64) == 0 && d->KoutP == d->Cout &&
           d->Cin == d->KinP && (d->Cin % 16) == 0 && d->Cin >= 32 && d->Ho == d->H && d->Wo == d->W && d->W >= 32 && d->H >= 8 &&
           (int64_t)d->N * d->H * d->W * d->x.ctot < (1LL << 30) && (int64_t)d->N * d->H * d->W * d->y.ctot < (1LL << 30) &&
           (int64_t)(d->Cout / 64) * (d->Cin / 16) * 18 * SW_UNIT_FLOATS * (int64_t)sizeof(float) < (1LL << 31);
}

// Do the stages form a dense block the sweep kernel covers?  (5 stages over ONE input buffer, stage k reading channels [0, nf + 32 k)
// and -- except the last -- writing the next 32 channels of that buffer; 32, 32, 32, 32, 64 output channels)
bool sweep_pattern(const tnr_conv_desc *st, int n, const char **why) {
    auto no = [&](const char *w) { if (why) *why = w; return false; };
    if (n != SW_NSTAGE) return no("not 5 stages");
    const tnr_conv_desc &d0 = st[0];
    if (d0.Cin < 32 || (d0.Cin % 16) != 0) return no("block input channels must be a multiple of 16, >= 32");
    for (int i = 0; i < n; ++i) {
        const tnr_conv_desc &d = st[i];
        if (d.mode != TNR_CONV_3x3 || (d.mma != TNR_MMA_BF16X3 && d.mma != TNR_MMA_BF16) || d.mma != d0.mma || d.pad_mode != 0)
            return no("stage is not a zero-padded 3x3 in bf16x3 (or all stages with bf16 operands)");
        if (d.N != d0.N || d.H != d0.H || d.W != d0.W || d.Ho != d0.H || d.Wo != d0.W) return no("pixel grids differ");
        if (d.x.ptr != d0.x.ptr || d.x.ctot != d0.x.ctot || d.x.coff != d0.x.coff) return no("stages read different buffers");
        if (d.Cin != d0.Cin + 32 * i || d.KinP != d.Cin) return no("input channels do not grow by 32 per stage");
        if (d.Cout != (i < 4 ? 32 : 64) || d.KoutP != d.Cout) return no("output channels are not 32, 32, 32, 32, 64");
        if (i < 4 && (d.y.ptr != d0.x.ptr || d.y.ctot != d0.x.ctot || d.y.coff != d0.x.coff + d.Cin)) return no("stage output is not the next channel group");
        if ((d.x.ctot % 4) || (d.x.coff % 4) || (d.y.ctot % 4) || (d.y.coff % 4)) return no("views must be 4-channel aligned");
        if (d.r1.ptr && ((d.r1.ctot % 4) || (d.r1.coff % 4) || (d.r1_ch % 4))) return no("r1 view");
        if (d.r2.ptr && ((d.r2.ctot % 4) || (d.r2.coff % 4))) return no("r2 view");
        if (d.m.ptr && ((d.m.ctot % 4) || (d.m.coff % 4) || (d.m_lo % 4) || (d.m_hi % 4))) return no("mask view");
        if (d.noise_pos < 0 || d.noise_pos > 2 || (i < 4 && d.noise_pos != 0)) return no("the noise multiplier belongs to the last stage");
        if ((int64_t)d.N * d.H * d.W * d.x.ctot >= (1LL << 30) || (int64_t)d.N * d.H * d.W * d.y.ctot >= (1LL << 30)) return no("buffer above 4 GiB");
        if (!d.x.ptr || !d.y.ptr || !d.wp) return no("null pointer");
    }
    const tnr_conv_desc &dl = st[n - 1];
    if (dl.y.ptr == d0.x.ptr && dl.y.coff < d0.x.coff + dl.Cin && dl.y.coff + dl.Cout > d0.x.coff) return no("last stage overwrites its own input");
    return true;
}

// which form runs (one per process: the weight image is laid out for it).  TNR_SWEEP_WAVES=8: the eight-wave form; TNR_SWEEP_FORM=dma:
// the four-wave form with the LDS weight ring (round 3's default); otherwise the four-wave direct form.
int sweep_form() {
    static const int form = [] {
        const char *w = getenv("TNR_SWEEP_WAVES"), *f = getenv("TNR_SWEEP_FORM");
        if (w && atoi(w) == 8) return 8;
        return (f && f[0] == 'd' && f[1] == 'm') ? 4 : 5;
    }();
    return form;
}
constexpr size_t S4D_LDS_BYTES = (size_t)(2 * SW_A_FLOATS) * sizeof(float);      // direct form: the input tile's double buffer only

int sweep_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(conv_sweep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)SW_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(conv_sweep4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)SW_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(conv_sweep4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)S4D_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(conv_sweep4_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)S4D_LDS_BYTES) != hipSuccess || cus < 1)
            cus = -1;
    }
    return cus;
}

}  // namespace

#ifdef SW_TIMELINE
extern "C" int tnr_debug_sweep_units(unsigned long long *out16, int reset) {
    if (out16 != nullptr && hipMemcpyFromSymbol(out16, HIP_SYMBOL(sw_tl2), sizeof(sw_tl2)) != hipSuccess) return TNR_ELAUNCH;
    if (reset) {
        unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(sw_tl2), z, sizeof(z)) != hipSuccess) return TNR_ELAUNCH;
    }
    return TNR_OK;
}
extern "C" int tnr_debug_sweep_timeline(unsigned long long *out16, int reset) {
    if (out16 != nullptr && hipMemcpyFromSymbol(out16, HIP_SYMBOL(sw_tl), sizeof(sw_tl)) != hipSuccess) return TNR_ELAUNCH;
    if (reset) {
        unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(sw_tl), z, sizeof(z)) != hipSuccess) return TNR_ELAUNCH;
    }
    return TNR_OK;
}
#endif

extern "C" int64_t tnr_conv_sweep_image_bytes(const tnr_conv_desc *stages, int32_t n) {
    if (stages == nullptr || !sweep_pattern(stages, n, nullptr)) return 0;
    if (stages[0].mma == TNR_MMA_BF16 && sweep_form() != 5) return 0;      // bf16 operands: the direct four-wave form only
    const int cus = sweep_cus();
    const int tpi = tnr_cdiv(stages[0].W, SW_TW) * tnr_cdiv(stages[0].H, SW_TH);
    if (cus < 1 || tpi > cus) return 0;          // an image's tiles must be co-resident (one workgroup per CU)
    return (int64_t)sw_total_units(stages[0].Cin / 16) * SW_UNIT_FLOATS * (int64_t)sizeof(float);
}

extern "C" int tnr_conv_sweep_pack(const tnr_conv_desc *stages, int32_t n, void *image, int64_t image_bytes, void *stream) {
    const char *why = "";
    TNR_REQUIRE(stages != nullptr && image != nullptr && sweep_pattern(stages, n, &why), "conv_sweep_pack: not a sweepable dense block (%s)", why);
    SweepPackK a;
    a.nck0 = stages[0].Cin / 16;
    a.units = sw_total_units(a.nck0);
    a.direct = sweep_form() == 5;
    TNR_REQUIRE((int64_t)a.units * SW_UNIT_FLOATS * (int64_t)sizeof(float) <= image_bytes, "conv_sweep_pack: image buffer too small");
    for (int i = 0; i < SW_NSTAGE; ++i) {
        a.wp[i] = stages[i].wp;
        a.KinP[i] = stages[i].KinP;
        a.KoutP[i] = stages[i].KoutP;
    }
    a.out = static_cast<float *>(image);
    hipLaunchKernelGGL(sweep_pack_kernel, dim3((unsigned)tnr_cdiv(a.units * 64, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return tnr_check_launch("conv_sweep_pack");
}

static_assert(sizeof(SweepPackK) <= sizeof(((tnr_sweep_pack_item *)nullptr)->opaque), "tnr_sweep_pack_item holds the pack kernel's arguments");

extern "C" int tnr_conv_sweep_pack_item(const tnr_conv_desc *stages, int32_t n, void *image, int64_t image_bytes, tnr_sweep_pack_item *item) {
    const char *why = "";
    TNR_REQUIRE(stages != nullptr && image != nullptr && item != nullptr && sweep_pattern(stages, n, &why),
                "conv_sweep_pack_item: not a sweepable dense block (%s)", why);
    SweepPackK a;
    a.nck0 = stages[0].Cin / 16;
    a.units = sw_total_units(a.nck0);
    a.direct = sweep_form() == 5;
    TNR_REQUIRE((int64_t)a.units * SW_UNIT_FLOATS * (int64_t)sizeof(float) <= image_bytes, "conv_sweep_pack_item: image buffer too small");
    for (int i = 0; i < SW_NSTAGE; ++i) {
        a.wp[i] = stages[i].wp;
        a.KinP[i] = stages[i].KinP;
        a.KoutP[i] = stages[i].KoutP;
    }
    a.out = static_cast<float *>(image);
    memset(item, 0, sizeof(*item));
    memcpy(item->opaque, &a, sizeof(a));
    item->units = a.units;
    return TNR_OK;
}

extern "C" int tnr_conv_sweep_pack_batch(const tnr_sweep_pack_item *items_dev, int32_t n, int32_t max_units, void *stream) {
    TNR_REQUIRE(items_dev != nullptr && n >= 1 && n <= 65535 && max_units >= 1, "conv_sweep_pack_batch: bad arguments");
    static_assert(sizeof(tnr_sweep_pack_item) % 8 == 0, "table stride");
    hipLaunchKernelGGL(sweep_pack_batch_strided_kernel, dim3((unsigned)tnr_cdiv(max_units * 64, 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, items_dev);
    return tnr_check_launch("conv_sweep_pack_batch");
}

extern "C" int tnr_conv_sweep(const tnr_conv_desc *stages, int32_t n, const void *image, uint32_t *ws, int64_t ws_bytes, uint32_t epoch,
                              void *stream) {
    const char *why = "";
    TNR_REQUIRE(stages != nullptr && image != nullptr && ws != nullptr && sweep_pattern(stages, n, &why),
                "conv_sweep: not a sweepable dense block (%s)", why);
    const tnr_conv_desc &d0 = stages[0];
    TNR_REQUIRE(tnr_conv_chain_workspace_bytes(&d0) <= ws_bytes, "conv_sweep: workspace too small");
    const int cus = sweep_cus();
    TNR_REQUIRE(cus >= 1, "conv_sweep: cannot set up the kernel");
    SweepK c;
    c.nck0 = d0.Cin / 16;
    c.tiles_x = tnr_cdiv(d0.W, SW_TW);
    c.tiles_y = tnr_cdiv(d0.H, SW_TH);
    c.tpi = c.tiles_x * c.tiles_y;
    c.tiles = c.tpi * d0.N;
    TNR_REQUIRE(c.tpi <= cus, "conv_sweep: the %d tiles of an image exceed the %d co-resident workgroups", c.tpi, cus);
    c.progress = ws;
    c.err = tnr_fault_word_or(ws + ws_bytes / 4 - 1);
    static const int nxcd = [] { const char *e = getenv("TNR_SWEEP_DISPENSERS"); return e ? atoi(e) : 1; }();
    c.nxcd = nxcd == 1 ? 1 : 8;
    c.disp = ws + (ws_bytes / 4 - 1 - 4 - CH_CU_KEYS - CH_SWEEP_WORDS);      // (words of their own, zero-initialised; every launch leaves them at zero)
    c.base = epoch * (uint32_t)(TNR_CHAIN_MAX + 2);
    c.wq = static_cast<const float *>(image);
    c.wq_bytes = sw_total_units(c.nck0) * SW_UNIT_FLOATS * (int)sizeof(float);
    for (int i = 0; i < SW_NSTAGE; ++i) {
        const tnr_conv_desc *d = &stages[i];
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
        k.ksplit = 1; k.split_stride = 0; k.bf = d->mma; k.reflect = 0;
    }
    // whole images per round of the grid, one tile per workgroup and round
    const int per_round = (cus / c.tpi) * c.tpi;
    const int grid = c.tiles < per_round ? c.tiles : per_round;
    const int form = sweep_form();
    if (stages[0].mma == TNR_MMA_BF16) {
        TNR_REQUIRE(form == 5, "conv_sweep: bf16 operands (use_amp) run in the direct four-wave form only");
        hipLaunchKernelGGL((conv_sweep4_kernel<true, true>), dim3((unsigned)grid), dim3(256), S4D_LDS_BYTES, (hipStream_t)stream, c);
        return tnr_check_launch("conv_sweep");
    }
    if (form == 8) hipLaunchKernelGGL(conv_sweep_kernel, dim3((unsigned)grid), dim3(512), SW_LDS_BYTES, (hipStream_t)stream, c);
    else if (form == 4) hipLaunchKernelGGL(conv_sweep4_kernel<false>, dim3((unsigned)grid), dim3(256), SW_LDS_BYTES, (hipStream_t)stream, c);
    else hipLaunchKernelGGL(conv_sweep4_kernel<true>, dim3((unsigned)grid), dim3(256), S4D_LDS_BYTES, (hipStream_t)stream, c);
    return tnr_check_launch("conv_sweep");
}

extern "C" int64_t tnr_conv_wq_bytes(const tnr_conv_desc *d) {
    if (d != nullptr && s2_ok(d)) return (int64_t)(d->Cout / 64) * (d->Cin / 16) * 4 * S2_NU * SW_UNIT_FLOATS * (int64_t)sizeof(float);
    if (d == nullptr || !d4_ok(d)) return 0;
    return (int64_t)(d->Cout / 64) * (d->Cin / 16) * 18 * SW_UNIT_FLOATS * (int64_t)sizeof(float);
}

extern "C" int tnr_conv_wq_pack(const tnr_conv_desc *d, void *image, int64_t image_bytes, void *stream) {
    if (d != nullptr && image != nullptr && d->wp != nullptr && s2_ok(d)) {      // the four-tap forms: 4 (parity planes | parity classes) x chunks x 4 taps x 2 N-tiles
        S2PackK a;
        a.wp = d->wp; a.KinP = d->KinP; a.KoutP = d->KoutP; a.nck = d->Cin / 16; a.ncb = d->Cout / 64;
        a.dgrad = d->mode == TNR_DGRAD_4x4_S2;
        a.units = a.ncb * 4 * a.nck * S2_NU;
        TNR_REQUIRE((int64_t)a.units * SW_UNIT_FLOATS * (int64_t)sizeof(float) <= image_bytes, "conv_wq_pack: image buffer too small");
        a.out = static_cast<float *>(image);
        hipLaunchKernelGGL(s2_pack_kernel, dim3((unsigned)tnr_cdiv(a.units * 64, 256)), dim3(256), 0, (hipStream_t)stream, a);
        return tnr_check_launch("conv_wq_pack (four-tap form)");
    }
    TNR_REQUIRE(d != nullptr && image != nullptr && d->wp != nullptr && d4_ok(d), "conv_wq_pack: the launch cannot use a pre-split weight stream");
    D4PackK a;
    a.wp = d->wp; a.KinP = d->KinP; a.KoutP = d->KoutP; a.nck = d->Cin / 16; a.ncb = d->Cout / 64;
    a.units = a.ncb * a.nck * 18;
    a.shuffle = d->shuffle;
    TNR_REQUIRE((int64_t)a.units * SW_UNIT_FLOATS * (int64_t)sizeof(float) <= image_bytes, "conv_wq_pack: image buffer too small");
    a.out = static_cast<float *>(image);
    hipLaunchKernelGGL(d4_pack_kernel, dim3((unsigned)tnr_cdiv(a.units * 64, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return tnr_check_launch("conv_wq_pack");
}

// the four-tap forms (4x4 stride 2 forward, its data-gradient) with a pre-split weight stream; 1: not for this kernel
int tnr_launch_conv_s2_d4(const tnr_conv_desc *d, void *stream) {
    if (!s2_ok(d) || d->wq == nullptr || d->wq_bytes < tnr_conv_wq_bytes(d) || d->noise_pos != 0) return 1;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(conv_s2_d4_kernel<TNR_CONV_4x4_S2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S4D_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(conv_s2_d4_kernel<TNR_CONV_4x4_S2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S4D_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(conv_s2_d4_kernel<TNR_DGRAD_4x4_S2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S4D_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(conv_s2_d4_kernel<TNR_DGRAD_4x4_S2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S4D_LDS_BYTES) != hipSuccess || cus < 1) {
            cus = 0;
            tnr_set_error("conv_s2_d4: cannot set up the kernel");
            return TNR_ELAUNCH;
        }
    }
    const bool dg = d->mode == TNR_DGRAD_4x4_S2;
    S2K c;
    ConvK &k = c.a;
    k.x = d->x.ptr; k.x_ct = d->x.ctot; k.x_co = d->x.coff;
    k.N = d->N; k.H = d->H; k.W = d->W; k.Cin = d->Cin;
    k.wp = d->wp; k.KinP = d->KinP; k.KoutP = d->KoutP;
    k.y = d->y.ptr; k.y_ct = d->y.ctot; k.y_co = d->y.coff; k.Ho = d->Ho; k.Wo = d->Wo; k.Cout = d->Cout;
    k.bias = d->bias; k.act = d->act; k.slope = d->slope; k.alpha = d->alpha;
    k.r1 = d->r1.ptr; k.r1_ct = d->r1.ctot; k.r1_co = d->r1.coff; k.r1_ch = d->r1_ch; k.beta1 = d->beta1;
    k.r2 = d->r2.ptr; k.r2_ct = d->r2.ctot; k.r2_co = d->r2.coff; k.alpha2 = d->alpha2;
    k.m = d->m.ptr; k.m_ct = d->m.ctot; k.m_co = d->m.coff; k.m_lo = d->m_lo; k.m_hi = d->m_hi; k.m_slope = d->m_slope;
    k.noise_pos = 0; k.noise_sigma = 0.f; k.noise_k0 = 0; k.noise_k1 = 0; k.noise_pix0 = 0;
    k.th_space = dg ? d->H : d->Ho; k.tw_space = dg ? d->W : d->Wo;
    k.ksplit = 1; k.split_stride = 0; k.bf = d->mma; k.reflect = 0;
    c.wq = static_cast<const float *>(d->wq);
    c.wq_bytes = (int)tnr_conv_wq_bytes(d);
    c.nck = d->Cin / 16;
    c.tiles_x = tnr_cdiv(k.tw_space, SW_TW);
    c.tiles_y = tnr_cdiv(k.th_space, SW_TH);
    c.ncb = d->Cout / 64;
    const int64_t tiles = (int64_t)c.tiles_x * c.tiles_y * c.ncb * d->N * (dg ? 4 : 1);
    if (tiles >= (1LL << 31)) return 1;
    c.tiles = (int)tiles;
    k.tiles_x = c.tiles_x; k.tiles_y = c.tiles_y; k.ncb = c.ncb;
    const unsigned grid = (unsigned)(c.tiles < 2 * cus ? c.tiles : 2 * cus);
    const bool amp = d->mma == TNR_MMA_BF16;
    if (dg) {
        if (amp) hipLaunchKernelGGL((conv_s2_d4_kernel<TNR_DGRAD_4x4_S2, true>), dim3(grid), dim3(256), S4D_LDS_BYTES, (hipStream_t)stream, c);
        else hipLaunchKernelGGL((conv_s2_d4_kernel<TNR_DGRAD_4x4_S2, false>), dim3(grid), dim3(256), S4D_LDS_BYTES, (hipStream_t)stream, c);
    } else {
        if (amp) hipLaunchKernelGGL((conv_s2_d4_kernel<TNR_CONV_4x4_S2, true>), dim3(grid), dim3(256), S4D_LDS_BYTES, (hipStream_t)stream, c);
        else hipLaunchKernelGGL((conv_s2_d4_kernel<TNR_CONV_4x4_S2, false>), dim3(grid), dim3(256), S4D_LDS_BYTES, (hipStream_t)stream, c);
    }
    return tnr_check_launch("conv_s2_d4");
}

// called by tnr_conv_forward (conv_tile.hip) for a launch that carries a pre-split weight stream; 1: not for this kernel
int tnr_launch_conv3x3_d4(const tnr_conv_desc *d, void *stream) {
    if (!d4_ok(d) || d->wq == nullptr || d->wq_bytes < tnr_conv_wq_bytes(d) || d->noise_pos < 0 || d->noise_pos > 2) return 1;
    static int occ = [] { const char *e = getenv("TNR_D4_OCC"); return e ? atoi(e) : 2; }();
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_d4_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S4D_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_d4_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S4D_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_d4_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S4D_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_d4_kernel<2, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S4D_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_d4_kernel<2, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S4D_LDS_BYTES) != hipSuccess || cus < 1) {
            cus = 0;
            tnr_set_error("conv3x3_d4: cannot set up the kernel");
            return TNR_ELAUNCH;
        }
    }
    D4K c;
    ConvK &k = c.a;
    k.x = d->x.ptr; k.x_ct = d->x.ctot; k.x_co = d->x.coff;
    k.N = d->N; k.H = d->H; k.W = d->W; k.Cin = d->Cin;
    k.wp = d->wp; k.KinP = d->KinP; k.KoutP = d->KoutP;
    k.y = d->y.ptr; k.y_ct = d->y.ctot; k.y_co = d->y.coff; k.Ho = d->Ho; k.Wo = d->Wo; k.Cout = d->Cout;
    k.bias = d->bias; k.act = d->act; k.slope = d->slope; k.alpha = d->alpha;
    k.r1 = d->r1.ptr; k.r1_ct = d->r1.ctot; k.r1_co = d->r1.coff; k.r1_ch = d->r1_ch; k.beta1 = d->beta1;
    k.r2 = d->r2.ptr; k.r2_ct = d->r2.ctot; k.r2_co = d->r2.coff; k.alpha2 = d->alpha2;
    k.m = d->m.ptr; k.m_ct = d->m.ctot; k.m_co = d->m.coff; k.m_lo = d->m_lo; k.m_hi = d->m_hi; k.m_slope = d->m_slope;
    k.noise_pos = d->noise_pos; k.noise_sigma = d->noise_sigma; k.noise_k0 = d->noise_key0; k.noise_k1 = d->noise_key1; k.noise_pix0 = d->noise_pix0;
    k.th_space = d->Ho; k.tw_space = d->Wo;
    k.ksplit = 1; k.split_stride = 0; k.bf = d->mma; k.reflect = d->pad_mode == 1;
    c.wq = static_cast<const float *>(d->wq);
    c.wq_bytes = (int)tnr_conv_wq_bytes(d);
    c.nck = d->Cin / 16;
    c.tiles_x = tnr_cdiv(d->Wo, SW_TW);
    c.tiles_y = tnr_cdiv(d->Ho, SW_TH);
    c.ncb = d->Cout / 64;
    const int64_t tiles = (int64_t)c.tiles_x * c.tiles_y * c.ncb * d->N;
    if (tiles >= (1LL << 31)) return 1;
    c.tiles = (int)tiles;
    k.tiles_x = c.tiles_x; k.tiles_y = c.tiles_y; k.ncb = c.ncb;
    const int slots = cus * (occ == 1 ? 1 : 2);
    const int grid = c.tiles < slots ? c.tiles : slots;
    if (d->shuffle == 2) {
        // the buffer the store addresses is the shuffled one: [N, 2 Ho, 2 Wo, Cout / 4] (same element count)
        const unsigned g2 = (unsigned)(c.tiles < 2 * cus ? c.tiles : 2 * cus);
        if (d->mma == TNR_MMA_BF16) hipLaunchKernelGGL((conv3x3_d4_kernel<2, true, true>), dim3(g2), dim3(256), S4D_LDS_BYTES, (hipStream_t)stream, c);
        else hipLaunchKernelGGL((conv3x3_d4_kernel<2, false, true>), dim3(g2), dim3(256), S4D_LDS_BYTES, (hipStream_t)stream, c);
        return tnr_check_launch("conv3x3_d4 (pixel-shuffle store)");
    }
    if (d->mma == TNR_MMA_BF16) hipLaunchKernelGGL((conv3x3_d4_kernel<2, true>), dim3((unsigned)(c.tiles < 2 * cus ? c.tiles : 2 * cus)), dim3(256), S4D_LDS_BYTES, (hipStream_t)stream, c);
    else if (occ == 1) hipLaunchKernelGGL(conv3x3_d4_kernel<1>, dim3((unsigned)grid), dim3(256), S4D_LDS_BYTES, (hipStream_t)stream, c);
    else hipLaunchKernelGGL(conv3x3_d4_kernel<2>, dim3((unsigned)grid), dim3(256), S4D_LDS_BYTES, (hipStream_t)stream, c);
    return tnr_check_launch("conv3x3_d4");
}
