// LDS-DMA variant of the tiled 3x3 convolution (EXPERIMENT, selected by TNR_CONV_DL=1 in conv_tile.hip): the operand tiles go
// from global memory straight into LDS with `buffer_load_dwordx4 ... lds` -- no staging registers, no ds_write pass, no
// bank-conflict-prone refill -- into the idle one of TWO operand buffers, so a chunk costs ONE workgroup barrier and the matrix
// pipe never waits for a refill (ablation of the register-staged body: loads + ds_writes cost 8 % per layer, 12 % in the chain).
//   workgroup   8 waves (one per CU: 2 x 75 KB of LDS), tile 16 x 32 pixels x 64 output channels, wave w owns tile rows 2w, 2w+1
//   LDS image   unpadded 64-byte rows (a pixel's / a [tap][cout] row's 16 channels).  An LDS-DMA instruction writes
//               base + lane * 16 bytes, i.e. a run of 16 whole rows, conflict-free by construction.  The fragment reads
//               (ds_read_b128, 8 lanes = 8 consecutive rows per pass) would hit 4-way conflicts on unpadded rows, so the four
//               16-byte slots of row p are permuted: channel quad g sits in slot g ^ ((p >> 1) & 3).  The permutation is applied
//               on the SOURCE side (lane (row, slot) fetches quad slot ^ f(row)) and on the read side -- the destination
//               stays linear, as the DMA requires.
//   pipeline    wait own DMA pieces (vmcnt 0) -> barrier (everyone's pieces landed; everyone left the other buffer) -> issue
//               the DMA of chunk k+1 into the other buffer -> 18 MFMA steps of chunk k.
// Arithmetic (chunk, tap, k order) is that of conv_tile_body: results are bit-identical.
#pragma once
#include "conv_body.h"
#include "conv_epilogue.h"

namespace {

template <int NT>
struct DlGeom {
    static constexpr int TW = 32, TH = 16, MT = 2, NW = 8, HT = TH + 2, WT = TW + 2, NC = NT * 32;
    static constexpr int IN_ROWS = HT * WT, IN_RUNS = (IN_ROWS + 15) / 16;      // a run = 16 rows = one wave-wide DMA instruction
    static constexpr int W_ROWS = 9 * NC, W_RUNS = W_ROWS / 16;
    static constexpr int IN_FLOATS = IN_RUNS * 256, BUF_FLOATS = (IN_RUNS + W_RUNS) * 256;
    static constexpr size_t LDS_BYTES = (size_t)2 * BUF_FLOATS * sizeof(float);
};

typedef __attribute__((address_space(3))) float tnr_lds_float;

// NL > 0: waves 8 .. 8 + NL - 1 of the workgroup are LOADERS -- they issue every DMA piece and never touch the matrix core; the
// eight MFMA waves issue nothing but fragment reads and MFMAs (a wave issues in order: a vector-memory instruction in front of an
// MFMA costs the issuing wave 60 - 190 cycles whether its data go to registers or to LDS).
template <int NT, bool COH, int NL>
__device__ __forceinline__ void conv3x3_dl_body(const ConvK a, const int cb, const int tx, const int ty, const int n, float *smem,
                                                tnr_lds_float *lds) {
    using G = DlGeom<NT>;
    constexpr int MT = G::MT, WT = G::WT, NC = G::NC;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, half = lane >> 5;
    const __amdgpu_buffer_rsrc_t x_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)((unsigned)a.N * a.H * a.W * a.x_ct * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rs =
        __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((unsigned)a.N * a.Ho * a.Wo * a.y_ct * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wp), 0, (int)(9u * a.KoutP * a.KinP * 4u), 0x00020000);
    const int ty0 = ty * G::TH, tx0 = tx * G::TW;
    const int nchunks = a.KinP / TNR_CK;

    // ---- DMA plan: run r = it * 8 + wave; lane (row = 16 r + lane / 4, slot = lane % 4) fetches channel quad slot ^ f(row).
    // Byte offsets without the chunk's channel offset; rows outside the image / beyond the tile / beyond KoutP point past
    // the end of the buffer (the hardware range check delivers zeros).
    constexpr unsigned OOB = 0xf0000000u;
    constexpr int IN_IT = (G::IN_RUNS + 7) / 8, W_IT = (G::W_RUNS + 7) / 8;
    unsigned in_vo[IN_IT], w_vo[W_IT];
    {
        const int srow = lane >> 2, sslot = lane & 3;
#pragma unroll
        for (int it = 0; it < IN_IT; ++it) {
            const int row = (it * 8 + wave) * 16 + srow;
            const int cg = sslot ^ ((row >> 1) & 3);
            const int hr = row / WT, hc = row - hr * WT;
            const int Y = ty0 + hr - 1, X = tx0 + hc - 1;
            const bool ok = (row < G::IN_ROWS) & (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
            in_vo[it] = ok ? (unsigned)(((n * a.H + Y) * a.W + X) * a.x_ct + a.x_co + cg * 4) * 4u : OOB;
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int row = (it * 8 + wave) * 16 + srow;
            const int cg = sslot ^ ((row >> 1) & 3);
            const int t = row / NC, co = row - t * NC;
            const int cog = cb * NC + co;
            const bool ok = (row < G::W_ROWS) & (cog < a.KoutP);
            w_vo[it] = ok ? (unsigned)((t * a.KoutP + cog) * a.KinP + cg * 4) * 4u : OOB;
        }
    }
    auto issue = [&](const int chunk, const int buf) {
        const unsigned cb4 = (unsigned)chunk * (TNR_CK * 4u);
        tnr_lds_float *base = lds + buf * G::BUF_FLOATS + wave * 256;
#pragma unroll
        for (int it = 0; it < IN_IT; ++it)
            if (it * 8 + wave < G::IN_RUNS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, base + it * 8 * 256, 16, (int)(in_vo[it] + cb4), 0, 0, COH ? TNR_COH_LOAD_AUX : 0);
#pragma unroll
        for (int it = 0; it < W_IT; ++it)
            if (it * 8 + wave < G::W_RUNS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, base + G::IN_FLOATS + it * 8 * 256, 16, (int)(w_vo[it] + cb4), 0, 0, 0);
    };

    if constexpr (NL > 0) {
        if (wave >= G::NW) {
            constexpr int RUNS = G::IN_RUNS + G::W_RUNS, L_IT = (RUNS + NL - 1) / NL;
            __builtin_amdgcn_s_setprio(3);     // a loader's few instructions go ahead of the MFMA waves of its SIMD
            const int lw = wave - G::NW, srow = lane >> 2, sslot = lane & 3;
            unsigned vo[L_IT];
#pragma unroll
            for (int it = 0; it < L_IT; ++it) {
                const int r = it * NL + lw;
                if (r < G::IN_RUNS) {
                    const int row = r * 16 + srow;
                    const int cg = sslot ^ ((row >> 1) & 3);
                    const int hr = row / WT, hc = row - hr * WT;
                    const int Y = ty0 + hr - 1, X = tx0 + hc - 1;
                    const bool ok = (row < G::IN_ROWS) & (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
                    vo[it] = ok ? (unsigned)(((n * a.H + Y) * a.W + X) * a.x_ct + a.x_co + cg * 4) * 4u : OOB;
                } else {
                    const int row = (r - G::IN_RUNS) * 16 + srow;
                    const int cg = sslot ^ ((row >> 1) & 3);
                    const int t = row / NC, co = row - t * NC;
                    const int cog = cb * NC + co;
                    const bool ok = (row < G::W_ROWS) & (cog < a.KoutP);
                    vo[it] = ok ? (unsigned)((t * a.KoutP + cog) * a.KinP + cg * 4) * 4u : OOB;
                }
            }
            auto lissue = [&](const int chunk, const int buf) {
                const unsigned cb4 = (unsigned)chunk * (TNR_CK * 4u);
                tnr_lds_float *base = lds + buf * G::BUF_FLOATS + lw * 256;
#pragma unroll
                for (int it = 0; it < L_IT; ++it) {
                    const int r = it * NL + lw;
                    if (r < G::IN_RUNS)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, base + it * NL * 256, 16, (int)(vo[it] + cb4), 0, 0, COH ? TNR_COH_LOAD_AUX : 0);
                    else if (r < RUNS)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, base + it * NL * 256, 16, (int)(vo[it] + cb4), 0, 0, 0);
                }
            };
            lissue(0, 0);
            for (int chunk = 0; chunk < nchunks; ++chunk) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (chunk + 1 < nchunks) lissue(chunk + 1, (chunk + 1) & 1);
            }
            return;
        }
    }

    // ---- fragment addresses (floats, buffer 0).  A: pixel p = P0 + s * WT + tx_ (s = tap row + M-tile, tx_ = tap column) as a
    // compile-time offset; its slot permutation f(p) = ((P0 + tx_) / 2 + s) & 3 (WT / 2 = 17 = 1 mod 4) goes into per-lane bases.
    const int P0 = (wave * MT) * WT + li;
    int addrA[3][4][2];
#pragma unroll
    for (int tx_ = 0; tx_ < 3; ++tx_)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) addrA[tx_][s][kk] = P0 * 16 + 4 * ((2 * kk + half) ^ ((((P0 + tx_) >> 1) + s) & 3));
    int addrB[2];     // row t * NC + nn * 32 + li: (row / 2) & 3 = (li / 2) & 3
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) addrB[kk] = G::IN_FLOATS + li * 16 + 4 * ((2 * kk + half) ^ ((li >> 1) & 3));

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int nn = 0; nn < NT; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][nn][r] = 0.f;

    constexpr int NSTEP = 18;
    auto phase = [&](const float *sb) {      // the 18 MFMA steps of one chunk over the operand buffer at sb
        f32x4 fa[2][MT], fb[2][NT];
        auto fetch = [&](int s_, int set) {
            const int t = s_ >> 1, kk = s_ & 1;
            const int ty_ = t / 3, tx_ = t - ty_ * 3;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
                fa[set][mi] = *reinterpret_cast<const f32x4 *>(sb + addrA[tx_][ty_ + mi][kk] + ((ty_ + mi) * WT + tx_) * 16);
#pragma unroll
            for (int nn = 0; nn < NT; ++nn)
                fb[set][nn] = *reinterpret_cast<const f32x4 *>(sb + addrB[kk] + (t * NC + nn * 32) * 16);
        };
        fetch(0, 0);
#pragma unroll
        for (int s_ = 0; s_ < NSTEP; ++s_) {
            if (s_ + 1 < NSTEP) fetch(s_ + 1, (s_ + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int nn = 0; nn < NT; ++nn)
                        acc[mi][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s_ & 1][mi][j], fb[s_ & 1][nn][j], acc[mi][nn], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if constexpr (NL == 0) {
        issue(0, 0);
        for (int chunk = 0; chunk < nchunks; chunk += 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (chunk + 1 < nchunks) issue(chunk + 1, 1);
            phase(smem);
            if (chunk + 1 < nchunks) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (chunk + 2 < nchunks) issue(chunk + 2, 0);
                phase(smem + G::BUF_FLOATS);
            }
        }
    } else {
        // barrier k: the loaders arrive when chunk k has landed, the MFMA waves when they have left the buffer of chunk k - 1
        for (int chunk = 0; chunk < nchunks; chunk += 2) {
            __builtin_amdgcn_s_barrier();
            phase(smem);
            if (chunk + 1 < nchunks) {
                __builtin_amdgcn_s_barrier();
                phase(smem + G::BUF_FLOATS);
            }
        }
    }
    conv_epilogue_dpp<TNR_CONV_3x3, G::TW, NT, MT, COH>(a, acc, cb, n, ty0, tx0, 0, wave, li, half, y_rs);
}

// ---- second form: the 4-wave workgroup and tile shapes of conv_tile_body (two independent workgroups per CU fill each other's
// barrier / epilogue gaps; the chain kernel's tile walk and hand-off protocol stay as they are), 8-channel chunks so that TWO
// operand buffers fit the 80 KB per-workgroup budget.  LDS rows are 32 bytes (8 channels: the lane-half h reads quad h); a DMA
// instruction fills a run of 32 rows.  Rows p and p + 4 share their banks, so quad q of row p sits in slot q ^ ((p >> 2) & 1).
template <int NT, int MT>
struct Dk8Geom {
    static constexpr int TW = 32, TH = 4 * MT, HT = TH + 2, WT = TW + 2, NC = NT * 32, CK8 = 8;
    static constexpr int IN_ROWS = HT * WT, IN_RUNS = (IN_ROWS + 31) / 32;
    static constexpr int W_ROWS = 9 * NC, W_RUNS = W_ROWS / 32;
    static constexpr int IN_FLOATS = IN_RUNS * 256, BUF_FLOATS = (IN_RUNS + W_RUNS) * 256;
    static constexpr size_t LDS_BYTES = (size_t)2 * BUF_FLOATS * sizeof(float);
};

// Same contract as conv_tile_body<TNR_CONV_3x3, 32, NT, MT, COH, false> (zero padding, no split-K); wait_chunk counts
// 16-channel chunks as there.  Requires Cin == KinP (whole chunks of real input channels).
template <int NT, int MT, bool COH, class WaitFn>
__device__ __forceinline__ void conv_tile_body_dk8(const ConvK a, const int cb, const int tx, const int ty, const int n, float *smem,
                                                   tnr_lds_float *lds, const int wait_chunk, WaitFn &&wait) {
    using G = Dk8Geom<NT, MT>;
    constexpr int WT = G::WT, NC = G::NC;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, half = lane >> 5;
    const __amdgpu_buffer_rsrc_t x_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)((unsigned)a.N * a.H * a.W * a.x_ct * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rs =
        __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((unsigned)a.N * a.Ho * a.Wo * a.y_ct * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wp), 0, (int)(9u * a.KoutP * a.KinP * 4u), 0x00020000);
    const int ty0 = ty * G::TH, tx0 = tx * G::TW;
    const int n8 = a.KinP / G::CK8, wc8 = wait_chunk < 0 ? -1 : 2 * wait_chunk;

    // DMA plan: run r = it * 4 + wave; lane (row = 32 r + lane / 2, slot = lane % 2) fetches channel quad slot ^ f(row)
    constexpr unsigned OOB = 0xf0000000u;
    constexpr int IN_IT = (G::IN_RUNS + 3) / 4, W_IT = (G::W_RUNS + 3) / 4;
    unsigned in_vo[IN_IT], w_vo[W_IT];
    {
        const int srow = lane >> 1, sslot = lane & 1;
#pragma unroll
        for (int it = 0; it < IN_IT; ++it) {
            const int row = (it * 4 + wave) * 32 + srow;
            const int cq = sslot ^ ((row >> 2) & 1);
            const int hr = row / WT, hc = row - hr * WT;
            const int Y = ty0 + hr - 1, X = tx0 + hc - 1;
            const bool ok = (row < G::IN_ROWS) & (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
            in_vo[it] = ok ? (unsigned)(((n * a.H + Y) * a.W + X) * a.x_ct + a.x_co + cq * 4) * 4u : OOB;
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int row = (it * 4 + wave) * 32 + srow;
            const int cq = sslot ^ ((row >> 2) & 1);
            const int t = row / NC, co = row - t * NC;
            const int cog = cb * NC + co;
            const bool ok = (row < G::W_ROWS) & (cog < a.KoutP);
            w_vo[it] = ok ? (unsigned)((t * a.KoutP + cog) * a.KinP + cq * 4) * 4u : OOB;
        }
    }
    auto issue = [&](const int k, const int buf) {
        const unsigned cb4 = (unsigned)k * (G::CK8 * 4u);
        tnr_lds_float *base = lds + buf * G::BUF_FLOATS + wave * 256;
#pragma unroll
        for (int it = 0; it < IN_IT; ++it)
            if (it * 4 + wave < G::IN_RUNS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, base + it * 4 * 256, 16, (int)(in_vo[it] + cb4), 0, 0, COH ? TNR_COH_LOAD_AUX : 0);
#pragma unroll
        for (int it = 0; it < W_IT; ++it)
            if (it * 4 + wave < G::W_RUNS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, base + G::IN_FLOATS + it * 4 * 256, 16, (int)(w_vo[it] + cb4), 0, 0, 0);
    };

    // fragment addresses (floats, buffer 0).  A: pixel p = P0 + (ty_ + mi) * WT + tx_; WT = 34 = 32 + 2, so bit 2 of p is bit 2
    // of P0 + d with d = 2 (ty_ + mi) + tx_ (0 .. 2 MT + 4): one per-lane base per d, the rest is a compile-time offset
    constexpr int ND = 2 * (MT + 1) + 3;
    const int P0 = (wave * MT) * WT + li;
    int addrA[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) addrA[d] = P0 * 8 + 4 * (half ^ (((P0 + d) >> 2) & 1));
    const int addrB = G::IN_FLOATS + li * 8 + 4 * (half ^ ((li >> 2) & 1));    // row t * NC + nn * 32 + li: bit 2 is that of li

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int nn = 0; nn < NT; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][nn][r] = 0.f;

    auto phase = [&](const float *sb) {      // the 9 MFMA steps (one per tap) of an 8-channel chunk over the operand buffer at sb
        f32x4 fa[2][MT], fb[2][NT];
        auto fetch = [&](int t, int set) {
            const int ty_ = t / 3, tx_ = t - ty_ * 3;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
                fa[set][mi] = *reinterpret_cast<const f32x4 *>(sb + addrA[2 * (ty_ + mi) + tx_] + ((ty_ + mi) * WT + tx_) * 8);
#pragma unroll
            for (int nn = 0; nn < NT; ++nn) fb[set][nn] = *reinterpret_cast<const f32x4 *>(sb + addrB + (t * NC + nn * 32) * 8);
        };
        fetch(0, 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) fetch(t + 1, (t + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int nn = 0; nn < NT; ++nn)
                        acc[mi][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t & 1][mi][j], fb[t & 1][nn][j], acc[mi][nn], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    {
        // one chunk: own DMA pieces landed (vmcnt 0 also covers the previous pass's stores: the chain's drain) -> barrier (all
        // pieces landed, everyone left the other buffer) -> publish the previous stage's tile -> neighbour wait if the NEXT
        // chunk is the first to read the previous stage's output -> DMA of the next chunk into the other buffer -> MFMA
        auto step = [&](const int k, const int buf, const float *sb) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (k == 0) wait.publish();
            if (k + 1 < n8) {
                if (k + 1 == wc8) wait();
                issue(k + 1, buf ^ 1);
            }
            phase(sb);
        };
        if (wc8 == 0) wait();
        issue(0, 0);
        for (int k = 0; k < n8; k += 2) {
            step(k, 0, smem);
            if (k + 1 < n8) step(k + 1, 1, smem + G::BUF_FLOATS);
        }
    }
    conv_epilogue_dpp<TNR_CONV_3x3, G::TW, NT, MT, COH>(a, acc, cb, n, ty0, tx0, 0, wave, li, half, y_rs);
}

template <int NT, int MT>
__global__ void __launch_bounds__(256, 2) conv3x3_dk8_kernel(const ConvK a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int bid = blockIdx.x;
    const int cb = bid % a.ncb;
    bid /= a.ncb;
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    bid /= a.tiles_y;
    conv_tile_body_dk8<NT, MT, false>(a, cb, tx, ty, bid, smem, (tnr_lds_float *)smem, -1, NoWait());
}

template <int NT, int MT>
int launch_conv3x3_dk8(const ConvK &k, int64_t tiles, hipStream_t s) {
    using G = Dk8Geom<NT, MT>;
    static_assert(G::LDS_BYTES <= 80 * 1024, "two workgroups per CU");
    static bool attr_done = false;
    auto fn = conv3x3_dk8_kernel<NT, MT>;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES) != hipSuccess) {
            tnr_set_error("conv3x3_dk8: cannot raise dynamic LDS to %zu bytes", G::LDS_BYTES);
            return TNR_ELAUNCH;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)tiles), dim3(256), G::LDS_BYTES, s, k);
    return tnr_check_launch("conv3x3_dk8");
}

template <int NT, int NL>
__global__ void __launch_bounds__(512 + 64 * NL, 1) conv3x3_dl_kernel(const ConvK a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int bid = blockIdx.x;
    const int cb = bid % a.ncb;
    bid /= a.ncb;
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    bid /= a.tiles_y;
    conv3x3_dl_body<NT, false, NL>(a, cb, tx, ty, bid, smem, (tnr_lds_float *)smem);
}

// the launch is eligible when the register-staged dispatcher would have taken the 64-cout 8 x 32 tile: plain zero padding,
// fp32 operands, whole 16-channel chunks of real input (no guard on the chunk tail), offsets below the range-check sentinel
inline bool conv3x3_dl_ok(const ConvK &k) {
    return !k.bf && !k.reflect && k.ksplit == 1 && (k.Cin % TNR_CK) == 0 && k.Cin == k.KinP &&
           (uint64_t)k.N * k.H * k.W * k.x_ct * 4u < 0xe0000000ull && (uint64_t)9 * k.KoutP * k.KinP * 4u < 0xe0000000ull;
}

template <int NL>
int launch_conv3x3_dl(ConvK k, hipStream_t s) {
    using G = DlGeom<2>;
    k.tiles_x = tnr_cdiv(k.tw_space, G::TW);
    k.tiles_y = tnr_cdiv(k.th_space, G::TH);
    k.ncb = tnr_cdiv(k.Cout, G::NC);
    const int64_t tiles = (int64_t)k.tiles_x * k.tiles_y * k.ncb * k.N;
    static bool attr_done = false;
    auto fn = conv3x3_dl_kernel<2, NL>;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES) != hipSuccess) {
            tnr_set_error("conv3x3_dl: cannot raise dynamic LDS to %zu bytes", G::LDS_BYTES);
            return TNR_ELAUNCH;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)tiles), dim3(512 + 64 * NL), G::LDS_BYTES, s, k);
    return tnr_check_launch("conv3x3_dl");
}

}  // namespace
