// The quad-transpose epilogue of conv_body.h as a function (same arithmetic, same order), for the kernels that do not go through
// conv_tile_body: conv_x3w8.h and the LDS-DMA experiments (conv_body_dl.h).
#pragma once
#include "conv_body.h"

namespace {

// Epilogue of a tile (the quad-transpose epilogue of conv_body.h as a function: same arithmetic, same order): `acc` holds the MFMA
// results of wave `wave` (M-tile mi = pixels (wave * MT + mi) * 32 .. + 31 of the TW-wide tile at (ty0, tx0), channel block cb).
// PixFn: (mi, row 0 .. 31 of that M-tile) -> (tile row, tile column) of the pixel that accumulator row holds.  The default is the row-major
// walk of a TW-wide tile; the Winograd kernel (conv_wino.hip) maps rows to the pixels of its 2 x 2 output patches instead.
// BIAS_STEP: distance between the bias entries of consecutive stored channels (4 when the store folds a pixel shuffle: stored channel c
// of sub-pixel s is conv channel 4 c + s; `a.bias` then points at entry s).
template <int MODE, int TW, int NT, int MT, bool COH, int DEPTH = 1, bool NOISE = true, int BIAS_STEP = 1, class PixFn>
__device__ __forceinline__ void conv_epilogue_dpp_map(const ConvK a, f32x16 (&acc)[MT][NT], const int cb, const int n, const int ty0,
                                                      const int tx0, const int par, const int li, const int half,
                                                      const __amdgpu_buffer_rsrc_t y_rs, PixFn &&pixfn) {
    constexpr bool DG2 = (MODE == TNR_DGRAD_4x4_S2);
    constexpr int NC = NT * 32;
    const int py = par >> 1, px = par & 1;
    // ---- epilogue.  The MFMA result layout gives a lane ONE channel (j = lane & 31) of 16 scattered pixels
    // (i = (r & 3) + 8 (r >> 2) + 4 half).  Stored as is that would be 16*MT*NT four-byte stores per lane; the
    // stores should be 16 bytes per lane and cover whole 128-byte pixel rows.  The four registers r = 4q .. 4q+3
    // of the four lanes of a quad (channels 4a .. 4a+3) form a 4x4 block {pixel 8q + 4half + k} x {channel 4a + b}:
    // transposing it INSIDE the quad with DPP quad_perm moves (two butterfly stages, 16 VALU ops per block) gives
    // lane b the float4 {channels 4a .. 4a+3} of pixel 8q + 4half + b.  A store instruction then has the 8 lanes
    // with the same b cover one pixel's 32 channels = one 128-byte row, 8 pixels per instruction.  No LDS, no
    // workgroup barrier: the first version transposed through the operand LDS (64 ds_write_b32 + 16 ds_read_b128
    // per lane between two __syncthreads()); at a stage boundary of the chain kernel that was 14 % of the time.
    TNR_STAMP(2);
#ifdef TNR_PRIO_EPI
    __builtin_amdgcn_s_setprio(TNR_PRIO_EPI);          // (experiment) the tile's stores / the next pass's first fetch ahead of the co-resident wave's MFMAs
#endif
    const int qa = li >> 2, qb = li & 3;               // quad index a (channel quad), position b inside the quad
    const bool b0 = (qb & 1) != 0, b1 = (qb & 2) != 0;
    auto quad_transpose = [&](float &v0, float &v1, float &v2, float &v3) {
        // stage 1: exchange with lane ^ 1 inside pairs (v0,v1), (v2,v3); stage 2: with lane ^ 2 inside (v0,v2), (v1,v3)
        auto xchg = [&](float &x, float &y, bool sel, int which) {
            const float send = sel ? x : y;
            const int r_ = which == 1 ? __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true)    // quad_perm:[1,0,3,2]
                                      : __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0x4E, 0xF, 0xF, true);   // quad_perm:[2,3,0,1]
            const float recv = __builtin_bit_cast(float, r_);
            x = sel ? recv : x;
            y = sel ? y : recv;
        };
        xchg(v0, v1, b0, 1);
        xchg(v2, v3, b0, 1);
        xchg(v0, v2, b1, 2);
        xchg(v1, v3, b1, 2);
    };
    TNR_STAMP(5);
    TNR_STAMP(6);
    TNR_STAMP(7);
    // Everything below is written branch-light (uniform switches hoisted, lane conditions as selects): a per-element
    // activation switch and per-unit residual / mask / partial-store branches once cost ~1000 cycles per float4 unit.
#ifdef TNR_ABL_NOMASK          /* (ablation build, results invalid: the upper bound of what a 1-bit activation mask could save -- no mask tensor read at all) */
    const bool has_r1 = a.r1 != nullptr, has_r2 = a.r2 != nullptr, has_m = false;
#else
    const bool has_r1 = a.r1 != nullptr, has_r2 = a.r2 != nullptr, has_m = a.m != nullptr;     // wave-uniform
#endif
    const bool has_noise = NOISE && a.noise_pos != 0;     // wave-uniform (NOISE = false: an instance for stages that never carry the ESRGAN+ noise)
    const bool all_full = (a.Cout & 3) == 0;                                                  // wave-uniform
    const float ns = a.act == TNR_ACT_LRELU ? a.slope : (a.act == TNR_ACT_RELU ? 0.f : 1.f);  // act(v) = max(v,0) + ns*min(v,0)
    int co_n[NT];
    bool co_ok[NT];
    f32x4 bv[NT];
    float b1f[NT], msf[NT];
    bool use_r1[NT], use_m[NT];
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
        const int co = cb * NC + nn * 32 + qa * 4;
        co_n[nn] = co;
        co_ok[nn] = co < a.Cout;
        bv[nn] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.bias != nullptr && co_ok[nn]) {
            if constexpr (BIAS_STEP != 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (co + k < a.Cout) bv[nn][k] = a.bias[(co + k) * BIAS_STEP];
            } else if (co + 4 <= a.Cout) {
                bv[nn] = *reinterpret_cast<const f32x4 *>(a.bias + co);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (co + k < a.Cout) bv[nn][k] = a.bias[co + k];
            }
        }
        use_r1[nn] = has_r1 && co < a.r1_ch;
        use_m[nn] = has_m && co >= a.m_lo && co < a.m_hi;
        b1f[nn] = use_r1[nn] ? a.beta1 : 0.f;
        msf[nn] = use_m[nn] ? a.m_slope : 1.f;        // factor for masked-off elements (1 outside the mask range)
    }
    // A unit = (mi, q): the NT float4s of pixel mi*32 + 8q + 4half + b.  Residual / mask loads of unit u+1 are issued
    // BEFORE the stores of unit u (two register sets): on gfx9 stores count in vmcnt like loads, so a load placed
    // after a store in program order makes its consumer wait for that store's acknowledgement.
    // DEPTH: how many units ahead the loads run (register sets = DEPTH + 1).  1 suits kernels whose co-resident waves cover the latency;
    // the one-wave-per-SIMD sweep kernel asks for all of them up front.
    constexpr int UNITS = MT * 4;
    constexpr int AHEAD = DEPTH < UNITS ? DEPTH : UNITS, SETS = AHEAD + 1 < UNITS ? AHEAD + 1 : UNITS;
    bool ok[SETS];
    size_t pixi[SETS];
    f32x4 q1[SETS][NT], q2[SETS][NT], qm[SETS][NT];
    auto prep = [&](int u, int set) {
        const int mi = u >> 2, q = u & 3;
        int rr, cc;
        pixfn(mi, 8 * q + 4 * half + qb, rr, cc);
        const int sy = ty0 + rr, sx = tx0 + cc;
        ok[set] = sy < a.th_space && sx < a.tw_space;
        const int oy = DG2 ? 2 * sy + py : sy;
        const int ox = DG2 ? 2 * sx + px : sx;
        const size_t pix = ((size_t)n * a.Ho + oy) * a.Wo + ox;
        pixi[set] = pix;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f}, one = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) {
            const bool okc = ok[set] && co_ok[nn];
            if (has_r1) {
                q1[set][nn] = zero;
                if (okc && use_r1[nn]) q1[set][nn] = *reinterpret_cast<const f32x4 *>(a.r1 + pix * a.r1_ct + a.r1_co + co_n[nn]);
            }
            if (has_r2) {
                q2[set][nn] = zero;
                if (okc) q2[set][nn] = *reinterpret_cast<const f32x4 *>(a.r2 + pix * a.r2_ct + a.r2_co + co_n[nn]);
            }
            if (has_m) {
                qm[set][nn] = one;
                if (okc && use_m[nn]) qm[set][nn] = *reinterpret_cast<const f32x4 *>(a.m + pix * a.m_ct + a.m_co + co_n[nn]);
            }
        }
    };
    auto finish = [&](int u, int set) {
        const int mi = u >> 2, q = u & 3;
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) {
            float v0 = acc[mi][nn][4 * q + 0], v1 = acc[mi][nn][4 * q + 1], v2 = acc[mi][nn][4 * q + 2], v3 = acc[mi][nn][4 * q + 3];
            quad_transpose(v0, v1, v2, v3);
            f32x4 v = f32x4{v0, v1, v2, v3} + bv[nn];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(ns, __builtin_fminf(v[e], 0.f), __builtin_fmaxf(v[e], 0.f)) * a.alpha;
            if (has_r1) v += b1f[nn] * q1[set][nn];
            if (has_noise) {           // (its own branch: the paths without noise keep their instruction stream)
                const f32x4 nm = tnr_gauss_mult4(((unsigned)pixi[set] + a.noise_pix0) * (unsigned)(a.Cout >> 2) + (unsigned)(co_n[nn] >> 2),
                                                 a.noise_k0, a.noise_k1, a.noise_sigma);
                if (a.noise_pos == 1) v *= nm;
                if (has_r2) v = v * a.alpha2 + q2[set][nn];
                if (a.noise_pos != 1) v *= nm;
            } else if (has_r2) {
                v = v * a.alpha2 + q2[set][nn];
            }
            if (has_m) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= (qm[set][nn][e] > 0.f ? 1.f : msf[nn]);
            }
            if (!(ok[set] && co_ok[nn])) continue;
            const int co = co_n[nn];
#ifdef TNR_ABL_NOEPISTORE      /* (ablation build: keeps the value alive without the store traffic; results invalid) */
            if (v[0] == 1.2345e30f) a.y[0] = v[1] + v[2] + v[3];
            continue;
#endif
            if (COH) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tnr_u32x4, v), y_rs,
                                                       (int)((unsigned)(pixi[set] * a.y_ct + a.y_co + co) * 4u), 0, TNR_COH_STORE_AUX);
            } else {
                float *yp = a.y + pixi[set] * a.y_ct + a.y_co + co;
                if (all_full) {
                    *reinterpret_cast<f32x4 *>(yp) = v;
                } else {                               // 3-channel image outputs only
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < a.Cout) yp[e] = v[e];
                }
            }
        }
    };
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) prep(u, u % SETS);
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
        if (u + AHEAD < UNITS) prep(u + AHEAD, (u + AHEAD) % SETS);
        finish(u, u % SETS);
    }
}

template <int MODE, int TW, int NT, int MT, bool COH, int DEPTH = 1, bool NOISE = true, int BIAS_STEP = 1>
__device__ __forceinline__ void conv_epilogue_dpp(const ConvK a, f32x16 (&acc)[MT][NT], const int cb, const int n, const int ty0,
                                                  const int tx0, const int par, const int wave, const int li, const int half,
                                                  const __amdgpu_buffer_rsrc_t y_rs) {
    conv_epilogue_dpp_map<MODE, TW, NT, MT, COH, DEPTH, NOISE, BIAS_STEP>(a, acc, cb, n, ty0, tx0, par, li, half, y_rs,
                                                                [&](int mi, int row, int &rr, int &cc) __attribute__((always_inline)) {
                                                                    const int p = (wave * MT + mi) * 32 + row;
                                                                    rr = p / TW;
                                                                    cc = p - rr * TW;
                                                                });
}

}  // namespace
