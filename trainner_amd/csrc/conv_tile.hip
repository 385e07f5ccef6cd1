// Tiled implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32,
// exact f32 == an fmaf chain), NHWC views, fused bias / activation / residual / mask epilogue.
//
// One workgroup = 256 threads = 4 waves (one per SIMD), two workgroups per CU (LDS <= 80 KiB).
//   M  = 256 output pixels (TH x TW spatial tile; TW in {32,16,8}); wave w owns pixels
//        [64w, 64w+64) = two 32-row MFMA tiles
//   N  = NT*32 output channels (NT in {1,2})
//   K  = taps x Cin, streamed in chunks of 16 channels: the (TH+KH-1) x (TW+KW-1) halo tile of the
//        chunk and the chunk's [tap][cout][16] weight slab are staged in LDS with a 20-dword pixel
//        stride, so every A / B fragment fetch is one conflict-free ds_read_b128 that feeds four
//        MFMA k-steps (lane-half h supplies channels 4h..4h+3 of an 8-channel group).
// Geometries (all reduce to "small dense tap set over a halo tile"):
//   3x3 s1 p1            9 taps, halo +1            (also data-gradient, with flipped packed weights)
//   3x3 s1 p1 of up2(x)  same, stager reads x[Y>>1][X>>1]          (block.py upconv_block)
//   4x4 s2 p1            as a 2x2 s1 conv over the space-to-depth view: chunk = (parity, 16 ch)
//   dgrad of 4x4 s2 p1   per output parity (py,px): 4 of the 9 halo taps, output scattered at stride 2
#include "common.h"

namespace {

struct ConvK {
    const float *x; int x_ct, x_co;
    int N, H, W, Cin;
    const float *wp; int KinP, KoutP;
    float *y; int y_ct, y_co; int Ho, Wo, Cout;
    const float *bias; int act; float slope; float alpha;
    const float *r1; int r1_ct, r1_co, r1_ch; float beta1;
    const float *r2; int r2_ct, r2_co; float alpha2;
    const float *m; int m_ct, m_co, m_lo, m_hi; float m_slope;
    int tiles_x, tiles_y, ncb;
    int th_space, tw_space;  // extent of the tile space (output dims, or gout dims for DGRAD_S2)
};

#ifdef TNR_TIMELINE   /* tools/probes/conv_timeline.hip: per-workgroup s_memtime stamps */
__device__ unsigned long long tnr_timeline[8 * 8192];
#define TNR_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) tnr_timeline[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TNR_STAMP(i) do { } while (0)
#endif

template <int MODE, int TW, int NT, int MT>
__global__ void __launch_bounds__(256, 2) conv_tile_kernel(const ConvK a) {
    TNR_STAMP(0);
    constexpr int TH = 128 * MT / TW;   // 4 waves x MT M-tiles of 32 pixels
    constexpr bool S2D = (MODE == TNR_CONV_4x4_S2);
    constexpr bool DG2 = (MODE == TNR_DGRAD_4x4_S2);
    constexpr bool UP = (MODE == TNR_CONV_3x3_UP2);
    constexpr int KH = S2D ? 2 : 3;
    constexpr int NTAPS = (S2D || DG2) ? 4 : 9;
    constexpr int HT = TH + KH - 1, WT = TW + KH - 1;
    constexpr int NC = NT * 32;
    constexpr int PST = TNR_PST, CK = TNR_CK;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_in = smem;                 // HT*WT*PST
    float *s_w = smem + HT * WT * PST;  // NTAPS*NC*PST

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, li = lane & 31, half = lane >> 5;

    int bid = blockIdx.x;
    const int cb = bid % a.ncb;
    bid /= a.ncb;
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    bid /= a.tiles_y;
    int par = 0;
    if (DG2) {
        par = bid & 3;
        bid >>= 2;
    }
    const int n = bid;
    const int ty0 = ty * TH, tx0 = tx * TW;
    const int py = par >> 1, px = par & 1;

    const float *wbase = a.wp + (DG2 ? (size_t)par * 4 * a.KoutP * a.KinP : (size_t)0);
    const int nck = a.KinP / CK;
    const int nchunks = S2D ? 4 * nck : nck;

    // per-lane A offsets (dwords) of the MT M-tiles this wave owns
    int aoff[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int p = (wave * MT + mi) * 32 + li;
        const int r = p / TW, c = p - r * TW;
        aoff[mi] = (r * WT + c) * PST + half * 4;
    }
    const int boff = li * PST + half * 4;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int nn = 0; nn < NT; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][nn][r] = 0.f;

    // ---- staging plan.  Every item is one float4 (4 channels); the per-item global offsets do not
    // depend on the chunk (except the parity of the space-to-depth view), so they are computed once.
    // Loads of chunk k+1 are issued into registers BEFORE the MFMA phase of chunk k and written to
    // LDS after it (issue-early / write-late): HBM/L2 latency hides under ~10-20k cycles of MFMA.
    constexpr int IN_ITEMS = HT * WT * 4, IN_IT = (IN_ITEMS + 255) / 256;
    constexpr int W_ITEMS = NTAPS * NC * 4, W_IT = (W_ITEMS + 255) / 256;
    int in_off[IN_IT];   // element offset into x (without the chunk's channel offset), -1 = zero fill
    int w_off[W_IT];     // element offset into the packed weights (without chunk offset), -1 = zero fill
    if (!S2D) {
#pragma unroll
        for (int it = 0; it < IN_IT; ++it) {
            const int i = tid + it * 256;
            const int pix = i >> 2, q = i & 3;
            const int hr = pix / WT, hc = pix - hr * WT;
            int Y = ty0 + hr - 1, X = tx0 + hc - 1;
            bool ok = i < IN_ITEMS;
            if (UP) {
                ok = ok & (Y >= 0) & (Y < 2 * a.H) & (X >= 0) & (X < 2 * a.W);
                Y >>= 1;
                X >>= 1;
            } else {
                ok = ok & (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
            }
            in_off[it] = ok ? (((n * a.H + Y) * a.W + X) * a.x_ct + a.x_co + q * 4) : -1;
        }
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
        const int i = tid + it * 256;
        const int row = i >> 2, q = i & 3;
        const int t = row / NC, co = row - t * NC;
        const int cog = cb * NC + co;
        const bool ok = (i < W_ITEMS) & (cog < a.KoutP);
        w_off[it] = ok ? ((t * a.KoutP + cog) * (S2D ? 4 * a.KinP : a.KinP) + q * 4) : -1;
    }
    f32x4 rin[IN_IT], rw[W_IT];

    auto load_chunk = [&](int chunk) {
        int c0, pp = 0;
        if (S2D) {
            pp = chunk / nck;
            c0 = (chunk - pp * nck) * CK;
        } else {
            c0 = chunk * CK;
        }
#pragma unroll
        for (int it = 0; it < IN_IT; ++it) {
            const int i = tid + it * 256;
            const int q = i & 3;
            int off;
            if (S2D) {
                const int pix = i >> 2;
                const int hr = pix / WT, hc = pix - hr * WT;
                const int Y = 2 * (ty0 + hr) - 1 + (pp >> 1), X = 2 * (tx0 + hc) - 1 + (pp & 1);
                const bool ok = (i < IN_ITEMS) & (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
                off = ok ? (((n * a.H + Y) * a.W + X) * a.x_ct + a.x_co + q * 4) : -1;
            } else {
                off = in_off[it];
            }
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (off >= 0 && c0 + q * 4 < a.Cin) v = *reinterpret_cast<const f32x4 *>(a.x + (size_t)off + c0);
            rin[it] = v;
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (w_off[it] >= 0) v = *reinterpret_cast<const f32x4 *>(wbase + (size_t)w_off[it] + (S2D ? pp * a.KinP : 0) + c0);
            rw[it] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int it = 0; it < IN_IT; ++it) {
            const int i = tid + it * 256;
            if (i < IN_ITEMS) *reinterpret_cast<f32x4 *>(s_in + (i >> 2) * PST + (i & 3) * 4) = rin[it];
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int i = tid + it * 256;
            if (i < W_ITEMS) *reinterpret_cast<f32x4 *>(s_w + (i >> 2) * PST + (i & 3) * 4) = rw[it];
        }
    };

    load_chunk(0);
    TNR_STAMP(4);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        __syncthreads();  // previous chunk's fragments are consumed
        store_chunk();
        __syncthreads();
        if (chunk == 0) TNR_STAMP(1);
        if (chunk + 1 < nchunks) load_chunk(chunk + 1);  // in flight during the MFMA phase below
        // ---- MFMA over taps x 16 channels, software-pipelined one step deep.  A step is one tap x one
        // 8-channel group: MT + NT ds_read_b128 feeding 4*MT*NT MFMAs (>= 1024 matrix-core cycles).  The
        // fragments of step s+1 are read into the other register set BEFORE the MFMAs of step s issue, so
        // a wave keeps the matrix pipe busy on its own; sched_barrier(0) pins that order (the scheduler
        // would otherwise sink the reads next to their first use).  All steps are unrolled: every LDS
        // address is a per-lane base + compile-time offset.
        constexpr int KG = CK / 8, NSTEP = NTAPS * KG;
        f32x4 fa[2][MT], fb[2][NT];
        auto fetch = [&](int s_, int set) {
            const int t = s_ / KG, kk = s_ - t * KG;
            int pos_y, pos_x;
            if (DG2) {
                pos_y = 1 + py - (t >> 1);
                pos_x = 1 + px - (t & 1);
            } else if (S2D) {
                pos_y = t >> 1;
                pos_x = t & 1;
            } else {
                pos_y = t / 3;
                pos_x = t - pos_y * 3;
            }
            const int tapoff = (pos_y * WT + pos_x) * PST + kk * 8;
            const int woff = t * NC * PST + boff + kk * 8;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) fa[set][mi] = *reinterpret_cast<const f32x4 *>(s_in + aoff[mi] + tapoff);
#pragma unroll
            for (int nn = 0; nn < NT; ++nn) fb[set][nn] = *reinterpret_cast<const f32x4 *>(s_w + woff + nn * 32 * PST);
        };
        auto mma = [&](int set) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int nn = 0; nn < NT; ++nn)
                        acc[mi][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][mi][j], fb[set][nn][j], acc[mi][nn], 0, 0, 0);
        };
        fetch(0, 0);
#pragma unroll
        for (int s_ = 0; s_ < NSTEP; ++s_) {
            if (s_ + 1 < NSTEP) fetch(s_ + 1, (s_ + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(s_ & 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue.  The MFMA result layout gives a lane ONE channel of 16 scattered pixels, which would
    // mean 16*MT*NT four-byte global stores per lane (store-issue bound).  Instead each wave transposes its
    // (MT*32 pixels) x (NT*32 channels) tile through LDS and every lane then owns float4s of 4 consecutive
    // channels of one pixel: 16-byte loads of residuals / masks and 16-byte stores, 4x fewer memory
    // instructions, whole 128/256-byte pixel rows per wave instruction.
    TNR_STAMP(2);
    __syncthreads();                                   // every wave is done with the operand tiles
    TNR_STAMP(5);
    float *s_o = smem + wave * (MT * 32 * NC);         // this wave's [MT*32][NC] tile (no padding needed)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int nn = 0; nn < NT; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * half;   // D[i][j]: j = lane&31 (channel), i = pixel
                s_o[(mi * 32 + i) * NC + nn * 32 + li] = acc[mi][nn][r];
            }
    TNR_STAMP(6);
    __syncthreads();
    TNR_STAMP(7);
    constexpr int C4 = NC / 4;                         // float4 per pixel row
    constexpr int UNITS = MT * 32 * C4 / 64;           // float4 per lane
    constexpr int G = 4, NG = UNITS / G;               // units are handled in groups of 4
    static_assert(UNITS % G == 0 && 64 % C4 == 0, "epilogue grouping");
    // 64 % C4 == 0: a lane keeps the same channel quad for all its units
    const int c4 = lane % C4, co = cb * NC + c4 * 4;
    const bool co_ok = co < a.Cout;
    const bool full = co + 4 <= a.Cout;                // false only for the 3-channel image output
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (a.bias != nullptr && co_ok) {
        if (full) {
            bv = *reinterpret_cast<const f32x4 *>(a.bias + co);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (co + k < a.Cout) bv[k] = a.bias[co + k];
        }
    }
    const bool use_r1 = a.r1 != nullptr && co < a.r1_ch;
    const bool use_r2 = a.r2 != nullptr;
    const bool use_m = a.m != nullptr && co >= a.m_lo && co < a.m_hi;
    // Residual / mask loads of group g+1 are issued BEFORE the stores of group g (two register sets): on
    // gfx9 stores count in vmcnt like loads, so a load placed after a store in program order makes its
    // consumer wait for that store's acknowledgement -- serialising the whole tail on write latency.
    bool ok[2][G];
    size_t pixi[2][G];
    f32x4 q1[2][G], q2[2][G], qm[2][G];
    auto prep = [&](int g, int set) {
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const int pl = (g * G + k) * (64 / C4) + lane / C4;
            const int p = wave * (MT * 32) + pl;
            const int rr = p / TW, cc = p - rr * TW;
            const int sy = ty0 + rr, sx = tx0 + cc;
            ok[set][k] = co_ok && sy < a.th_space && sx < a.tw_space;
            const int oy = DG2 ? 2 * sy + py : sy;
            const int ox = DG2 ? 2 * sx + px : sx;
            const size_t pix = ((size_t)n * a.Ho + oy) * a.Wo + ox;
            pixi[set][k] = pix;
            if (ok[set][k]) {
                if (use_r1) q1[set][k] = *reinterpret_cast<const f32x4 *>(a.r1 + pix * a.r1_ct + a.r1_co + co);
                if (use_r2) q2[set][k] = *reinterpret_cast<const f32x4 *>(a.r2 + pix * a.r2_ct + a.r2_co + co);
                if (use_m) qm[set][k] = *reinterpret_cast<const f32x4 *>(a.m + pix * a.m_ct + a.m_co + co);
            }
        }
    };
    auto finish = [&](int g, int set) {
#pragma unroll
        for (int k = 0; k < G; ++k) {
            if (!ok[set][k]) continue;
            const int pl = (g * G + k) * (64 / C4) + lane / C4;
            f32x4 v = *reinterpret_cast<const f32x4 *>(s_o + pl * NC + c4 * 4) + bv;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = tnr_act(v[e], a.act, a.slope) * a.alpha;
            if (use_r1) v += a.beta1 * q1[set][k];
            if (use_r2) v = v * a.alpha2 + q2[set][k];
            if (use_m) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= (qm[set][k][e] > 0.f ? 1.f : a.m_slope);
            }
            float *yp = a.y + pixi[set][k] * a.y_ct + a.y_co + co;
            if (full) {
                *reinterpret_cast<f32x4 *>(yp) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (co + e < a.Cout) yp[e] = v[e];
            }
        }
    };
    prep(0, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) prep(g + 1, (g + 1) & 1);
        finish(g, g & 1);
    }
    TNR_STAMP(3);
}

template <int MODE, int TW, int NT, int MT = 2>
int launch_conv(const ConvK &k, int tiles, hipStream_t s) {
    constexpr int TH = 128 * MT / TW;
    constexpr int KH = (MODE == TNR_CONV_4x4_S2) ? 2 : 3;
    constexpr int NTAPS = (MODE == TNR_CONV_4x4_S2 || MODE == TNR_DGRAD_4x4_S2) ? 4 : 9;
    constexpr size_t lds_main = (size_t)((TH + KH - 1) * (TW + KH - 1) + NTAPS * NT * 32) * TNR_PST * sizeof(float);
    constexpr size_t lds_epi = (size_t)4 * MT * 32 * NT * 32 * sizeof(float);   // output transpose tiles of the 4 waves
#ifdef TNR_DEBUG_LDS_PAD   /* experiment knob: force one workgroup per CU */
    constexpr size_t lds = (lds_main > lds_epi ? lds_main : lds_epi) + TNR_DEBUG_LDS_PAD;
#else
    constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
    static_assert(lds <= 80 * 1024, "conv tile exceeds the 2-workgroups-per-CU LDS budget");
#endif
    static bool attr_done = false;
    auto fn = conv_tile_kernel<MODE, TW, NT, MT>;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
            tnr_set_error("conv_tile: cannot raise dynamic LDS to %zu bytes", lds);
            return TNR_ELAUNCH;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(fn, dim3(tiles), dim3(256), lds, s, k);
    return tnr_check_launch("conv_tile");
}

template <int MODE>
int dispatch_conv(const ConvK &k, int tw, int nt, int tiles, hipStream_t s) {
    if (tw == 32) return nt == 2 ? launch_conv<MODE, 32, 2>(k, tiles, s) : launch_conv<MODE, 32, 1>(k, tiles, s);
    if (tw == 16) return nt == 2 ? launch_conv<MODE, 16, 2>(k, tiles, s) : launch_conv<MODE, 16, 1>(k, tiles, s);
    return nt == 2 ? launch_conv<MODE, 8, 2>(k, tiles, s) : launch_conv<MODE, 8, 1>(k, tiles, s);
}

}  // namespace

extern "C" int tnr_conv_forward(const tnr_conv_desc *d, void *stream) {
    TNR_REQUIRE(d != nullptr && d->x.ptr && d->y.ptr && d->wp, "conv: null pointer");
    TNR_REQUIRE(d->mode >= TNR_CONV_3x3 && d->mode <= TNR_DGRAD_4x4_S2, "conv: bad mode %d", d->mode);
    TNR_REQUIRE((d->x.ctot % 4) == 0 && (d->x.coff % 4) == 0 && (d->Cin % 4) == 0,
                "conv: input view must be 4-channel aligned (ctot %d coff %d Cin %d)", d->x.ctot, d->x.coff, d->Cin);
    TNR_REQUIRE((d->KinP % TNR_CK) == 0 && (d->KoutP % 32) == 0, "conv: bad packed dims %d %d", d->KinP, d->KoutP);
    TNR_REQUIRE(d->Cin <= d->KinP && d->Cout <= d->KoutP, "conv: Cin/Cout exceed the packing");
    TNR_REQUIRE((int64_t)d->N * d->H * d->W * d->x.ctot < (1LL << 31), "conv: input buffer above 2^31 elements needs 64-bit offsets");
    TNR_REQUIRE((d->y.ctot % 4) == 0 && (d->y.coff % 4) == 0, "conv: output view must be 4-channel aligned");
    TNR_REQUIRE(d->r1.ptr == nullptr || ((d->r1.ctot % 4) == 0 && (d->r1.coff % 4) == 0 && (d->r1_ch % 4) == 0),
                "conv: r1 view / r1_ch must be 4-channel aligned");
    TNR_REQUIRE(d->r2.ptr == nullptr || ((d->r2.ctot % 4) == 0 && (d->r2.coff % 4) == 0), "conv: r2 view must be 4-channel aligned");
    TNR_REQUIRE(d->m.ptr == nullptr || ((d->m.ctot % 4) == 0 && (d->m.coff % 4) == 0 && (d->m_lo % 4) == 0 && (d->m_hi % 4) == 0),
                "conv: mask view / range must be 4-channel aligned");
    TNR_REQUIRE(d->Cout % 4 == 0 || (d->r1.ptr == nullptr && d->r2.ptr == nullptr && d->m.ptr == nullptr),
                "conv: residual / mask epilogues need Cout %% 4 == 0");
    ConvK k;
    k.x = d->x.ptr; k.x_ct = d->x.ctot; k.x_co = d->x.coff;
    k.N = d->N; k.H = d->H; k.W = d->W; k.Cin = d->Cin;
    k.wp = d->wp; k.KinP = d->KinP; k.KoutP = d->KoutP;
    k.y = d->y.ptr; k.y_ct = d->y.ctot; k.y_co = d->y.coff; k.Ho = d->Ho; k.Wo = d->Wo; k.Cout = d->Cout;
    k.bias = d->bias; k.act = d->act; k.slope = d->slope; k.alpha = d->alpha;
    k.r1 = d->r1.ptr; k.r1_ct = d->r1.ctot; k.r1_co = d->r1.coff; k.r1_ch = d->r1_ch; k.beta1 = d->beta1;
    k.r2 = d->r2.ptr; k.r2_ct = d->r2.ctot; k.r2_co = d->r2.coff; k.alpha2 = d->alpha2;
    k.m = d->m.ptr; k.m_ct = d->m.ctot; k.m_co = d->m.coff; k.m_lo = d->m_lo; k.m_hi = d->m_hi; k.m_slope = d->m_slope;

    int sh, sw;  // tile space
    switch (d->mode) {
        case TNR_CONV_3x3:
            TNR_REQUIRE(d->Ho == d->H && d->Wo == d->W, "conv3x3: output must match input size");
            sh = d->Ho; sw = d->Wo;
            break;
        case TNR_CONV_3x3_UP2:
            TNR_REQUIRE(d->Ho == 2 * d->H && d->Wo == 2 * d->W, "conv3x3_up2: output must be 2x input");
            sh = d->Ho; sw = d->Wo;
            break;
        case TNR_CONV_4x4_S2:
            TNR_REQUIRE(2 * d->Ho == d->H && 2 * d->Wo == d->W, "conv4x4s2: output must be input/2");
            sh = d->Ho; sw = d->Wo;
            break;
        default:  // TNR_DGRAD_4x4_S2
            TNR_REQUIRE(d->Ho == 2 * d->H && d->Wo == 2 * d->W, "dgrad4x4s2: output must be 2x gout size");
            sh = d->H; sw = d->W;
            break;
    }
    k.th_space = sh; k.tw_space = sw;
    const int tw = sw >= 32 ? 32 : (sw >= 16 ? 16 : 8);
    const int nt = d->Cout > 32 ? 2 : 1;
    // 32-cout 3x3 layers (the dense-block convs and their gradients): 4 M-tiles per wave (16x32 pixel
    // tile) so the weight slab and the fixed per-workgroup costs are amortised like in the 64-cout kernel
    const bool big_m = (d->mode == TNR_CONV_3x3) && nt == 1 && tw == 32 && sh >= 16;
    const int th = (big_m ? 512 : 256) / tw;
    k.tiles_x = tnr_cdiv(sw, tw);
    k.tiles_y = tnr_cdiv(sh, th);
    k.ncb = tnr_cdiv(d->Cout, nt * 32);
    const int64_t tiles = (int64_t)k.tiles_x * k.tiles_y * k.ncb * d->N * (d->mode == TNR_DGRAD_4x4_S2 ? 4 : 1);
    TNR_REQUIRE(tiles > 0 && tiles < (1LL << 31), "conv: grid too large");
    hipStream_t s = (hipStream_t)stream;
    if (big_m) return launch_conv<TNR_CONV_3x3, 32, 1, 4>(k, (int)tiles, s);
    switch (d->mode) {
        case TNR_CONV_3x3: return dispatch_conv<TNR_CONV_3x3>(k, tw, nt, (int)tiles, s);
        case TNR_CONV_3x3_UP2: return dispatch_conv<TNR_CONV_3x3_UP2>(k, tw, nt, (int)tiles, s);
        case TNR_CONV_4x4_S2: return dispatch_conv<TNR_CONV_4x4_S2>(k, tw, nt, (int)tiles, s);
        default: return dispatch_conv<TNR_DGRAD_4x4_S2>(k, tw, nt, (int)tiles, s);
    }
}
