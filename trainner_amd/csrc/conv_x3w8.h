// TNR_MMA_BF16X3, 64-cout 3x3 convolution / data-gradient with BOTH operands pre-split in LDS: one 8-wave workgroup per CU.
//
// conv_tile_body's split-operand form keeps two 4-wave workgroups per CU and therefore only 80 KB each: the input tile fits as three
// bf16 planes, the weight slab does not, so every wave splits its B fragments again at every tap of every tile (the vector ALU, not
// the matrix core, sets the pace there).  Here ONE workgroup of 8 waves owns the CU's LDS: input tile 18 x 34 pixels and weight slab
// 9 taps x 64 couts, both as 96-byte rows of three bf16 planes (hi, mid, lo; 114 KB).  Every operand element is split once, by the
// stager; the MFMA phase is ds_read_b128 + v_mfma_f32_32x32x16_bf16 only, and a weight slab serves a 16 x 32 pixel tile instead of
// 8 x 32.  Tile 16 x 32 pixels x 64 output channels; wave w owns tile rows 2w, 2w + 1 (two M-tiles) and both N-tiles.
// Arithmetic (split, kept partial products and their order, chunk and tap order) is that of conv_tile_body<.., BF = 2>: results are
// bit-identical to it.
#pragma once
#include "conv_body.h"
#include "conv_epilogue.h"

namespace {

struct X3W8 {
    static constexpr int TW = 32, TH = 16, MT = 2, NT = 2, NW = 8, HT = TH + 2, WT = TW + 2, NC = 64;
    static constexpr int ROW = TNR_X3_ROW;                       // floats per LDS row: 3 planes x 16 bf16
    static constexpr int IN_ROWS = HT * WT, W_ROWS = 9 * NC;
    static constexpr size_t LDS_BYTES = (size_t)(IN_ROWS + W_ROWS) * ROW * sizeof(float);
};

__global__ void __launch_bounds__(512, 1) conv3x3_x3w8_kernel(const ConvK a) {
    using G = X3W8;
    constexpr int MT = G::MT, NT = G::NT, WT = G::WT, NC = G::NC, ROW = G::ROW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_in = smem, *s_w = smem + G::IN_ROWS * ROW;
    int bid = blockIdx.x;
    const int cb = bid % a.ncb;
    bid /= a.ncb;
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int n = bid / a.tiles_y;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, half = lane >> 5;
    const __amdgpu_buffer_rsrc_t y_rs =
        __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((unsigned)a.N * a.Ho * a.Wo * a.y_ct * 4u), 0x00020000);
    const int ty0 = ty * G::TH, tx0 = tx * G::TW;
    const int nchunks = a.KinP / TNR_CK;

    // ---- staging plan: item i = tid + 512 it -> row i / 4, channel quad i % 4 (element offsets without the chunk's channel offset)
    constexpr int IN_IT = (G::IN_ROWS * 4 + 511) / 512, W_IT = (G::W_ROWS * 4 + 511) / 512;
    int in_off[IN_IT], w_off[W_IT];
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
        for (int it = 0; it < IN_IT; ++it) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (in_off[it] >= 0) v = *reinterpret_cast<const f32x4 *>(a.x + (size_t)in_off[it] + c0);
            rin[it] = v;
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (w_off[it] >= 0) v = *reinterpret_cast<const f32x4 *>(a.wp + (size_t)w_off[it] + c0);
            rw[it] = v;
        }
    };
    // row r of either operand: plane P at float 8 P, the 16-byte slot of channels 8 h .. 8 h + 7 at slot h ^ ((r >> TNR_X3_SWZ) & 1)
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

    // ---- fragment addresses
    int apix[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) apix[mi] = (wave * MT + mi) * WT + li;
    const int boff = li * ROW + 4 * (half ^ ((li >> TNR_X3_SWZ) & 1));      // row t * 64 + nn * 32 + li: the swizzle bit is that of li

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int nn = 0; nn < NT; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][nn][r] = 0.f;

    tnr_bf16x8 ca[MT][3], cb_[2][NT][3];
    auto load_a = [&](int t, int mi) {
        const int pp = apix[mi] + (t / 3) * WT + (t % 3);
        const float *src = s_in + pp * ROW + 4 * (half ^ ((pp >> TNR_X3_SWZ) & 1));
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) ca[mi][sp] = *reinterpret_cast<const tnr_bf16x8 *>(src + 8 * sp);
    };
    auto load_b = [&](int t, int set) {
#pragma unroll
        for (int nn = 0; nn < NT; ++nn)
#pragma unroll
            for (int sp = 0; sp < 3; ++sp)
                cb_[set][nn][sp] = *reinterpret_cast<const tnr_bf16x8 *>(s_w + boff + (t * NC + nn * 32) * ROW + 8 * sp);
    };
    auto mma = [&](int mi, int set) {
        constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};      // the six kept partial products, smallest first
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int nn = 0; nn < NT; ++nn)
                acc[mi][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[mi][TA[p]], cb_[set][nn][TB[p]], acc[mi][nn], 0, 0, 0);
    };

    load_chunk(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        __syncthreads();          // the previous chunk's fragments are consumed
        store_chunk();
        __syncthreads();
        if (chunk + 1 < nchunks) load_chunk(chunk + 1);      // in flight during the MFMA phase
        load_b(0, 0);
        load_a(0, 0);
        load_a(0, 1);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // M-tile 0's MFMAs cover the loads of M-tile 1's next fragments and vice versa; the B fragments of tap t+1 go
            // into the other register set at the start of tap t
            if (t + 1 < 9) load_b(t + 1, (t + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(0, t & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < 9) load_a(t + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma(1, t & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < 9) load_a(t + 1, 1);
        }
    }
    conv_epilogue_dpp<TNR_CONV_3x3, G::TW, NT, MT, false>(a, acc, cb, n, ty0, tx0, 0, wave, li, half, y_rs);
}

inline bool conv3x3_x3w8_ok(const ConvK &k) {
    return k.bf == 2 && !k.reflect && k.ksplit == 1 && k.Cin == k.KinP && (k.Cin % TNR_CK) == 0;
}

inline int launch_conv3x3_x3w8(ConvK k, hipStream_t s) {
    using G = X3W8;
    k.tiles_x = tnr_cdiv(k.tw_space, G::TW);
    k.tiles_y = tnr_cdiv(k.th_space, G::TH);
    k.ncb = tnr_cdiv(k.Cout, G::NC);
    const int64_t tiles = (int64_t)k.tiles_x * k.tiles_y * k.ncb * k.N;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_x3w8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)G::LDS_BYTES) != hipSuccess) {
            tnr_set_error("conv3x3_x3w8: cannot raise dynamic LDS to %zu bytes", G::LDS_BYTES);
            return TNR_ELAUNCH;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(conv3x3_x3w8_kernel, dim3((unsigned)tiles), dim3(512), G::LDS_BYTES, s, k);
    return tnr_check_launch("conv3x3_x3w8");
}

}  // namespace
