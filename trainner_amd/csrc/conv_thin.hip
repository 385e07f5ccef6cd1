// 3x3 s1 p1 convolution with <= 4 OUTPUT channels (G's last conv 64 -> 3, and the data-gradients that end in the
// RGB image: VGG conv1_1 and D conv0, 64 -> 3): one thread = one output pixel x 4 channels on the vector ALUs.
// The matrix-core kernel pads 3 output channels to a 32-wide MFMA tile (8x the work: 1.3 ms per launch at
// 16 x 512 x 512); here the launch is bounded by reading the 64-channel input once (1.07 GB, ~0.25 ms).
//   tile 16 x 16 pixels per workgroup, input halo tile (18 x 18) x 16 channels per chunk in LDS (20-dword pixel
//   stride: conflict-free ds_read_b128), the chunk's [tap][16 ch][4 cout] weights in LDS read as broadcasts;
//   fp32 FMA chain per output in (chunk, tap, channel) order.
#include "common.h"
#include <cstdlib>

namespace {

struct ThinK {
    const float *x; int x_ct, x_co; int N, H, W, Cin;
    const float *wp;            // [9][KinP][4]
    int KinP;
    float *y; int y_ct, y_co, Cout;
    const float *bias; float alpha;
    int tiles_x, tiles_y;
};

constexpr int TT = 16, THW = TT + 2, TPST = TNR_PST;

#ifndef THIN_SCALAR_W
#define THIN_SCALAR_W 1        /* the chunk's weights through the scalar cache into SGPR operands (0: staged in LDS, read as broadcasts) */
#endif
__global__ void __launch_bounds__(256) conv_thin_kernel(const ThinK a) {
    __shared__ __attribute__((aligned(16))) float s_in[THW * THW * TPST];
#if !THIN_SCALAR_W
    __shared__ __attribute__((aligned(16))) float s_w[9 * TNR_CK * 4];
#else
    // every lane multiplies by the SAME weight: the [tap][channel][4 cout] table is read with wave-uniform addresses from constant
    // address space (s_load: scalar cache -> SGPR operand of the FMA) instead of one LDS broadcast read per four FMAs, which kept the
    // LDS pipe as busy as the vector ALUs (the launch ran at 2 TB/s of its 1.07 GB input, a quarter of what HBM delivers)
    typedef const __attribute__((address_space(4))) f32x4 cf32x4;
    cf32x4 *wtab = (cf32x4 *)(a.wp);
#endif
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int n = bid / a.tiles_y;
    const int y0 = ty * TT, x0 = tx * TT;
    const int py = tid / TT, px = tid - py * TT;

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int IN_ITEMS = THW * THW * 4, IN_IT = (IN_ITEMS + 255) / 256;
    int in_off[IN_IT];
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
        const int i = tid + it * 256;
        const int pix = i >> 2, q = i & 3;
        const int hr = pix / THW, hc = pix - hr * THW;
        const int Y = y0 + hr - 1, X = x0 + hc - 1;
        const bool ok = (i < IN_ITEMS) & (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
        in_off[it] = ok ? (((n * a.H + Y) * a.W + X) * a.x_ct + a.x_co + q * 4) : -1;
    }
    const int nchunks = a.KinP / TNR_CK;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int c0 = chunk * TNR_CK;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < IN_IT; ++it) {
            const int i = tid + it * 256;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (in_off[it] >= 0 && c0 + (i & 3) * 4 < a.Cin) v = *reinterpret_cast<const f32x4 *>(a.x + (size_t)in_off[it] + c0);
            if (i < IN_ITEMS) *reinterpret_cast<f32x4 *>(s_in + (i >> 2) * TPST + (i & 3) * 4) = v;
        }
#if !THIN_SCALAR_W
        for (int i = tid; i < 9 * TNR_CK; i += 256) {   // one float4 (4 couts) per (tap, channel)
            const int t = i / TNR_CK, c = i - t * TNR_CK;
            *reinterpret_cast<f32x4 *>(s_w + i * 4) = *reinterpret_cast<const f32x4 *>(a.wp + ((size_t)t * a.KinP + c0 + c) * 4);
        }
#endif
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float *xp = s_in + ((py + t / 3) * THW + px + t % 3) * TPST;
#pragma unroll
            for (int c4 = 0; c4 < TNR_CK / 4; ++c4) {
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(xp + c4 * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#if THIN_SCALAR_W
                    const f32x4 w = wtab[t * a.KinP + c0 + c4 * 4 + k];                                       // wave-uniform: s_load
#else
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(s_w + (t * TNR_CK + c4 * 4 + k) * 4);   // broadcast
#endif
#pragma unroll
                    for (int o = 0; o < 4; ++o) acc[o] = __builtin_fmaf(xv[k], w[o], acc[o]);
                }
            }
        }
    }
    const int oy = y0 + py, ox = x0 + px;
    if (oy < a.H && ox < a.W) {
        float *yp = a.y + (((size_t)n * a.H + oy) * a.W + ox) * a.y_ct + a.y_co;
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (o < a.Cout) yp[o] = (acc[o] + (a.bias != nullptr ? a.bias[o] : 0.f)) * a.alpha;
    }
}

// [9][KinP][4] <- OIHW: kind 0: forward of a Cout <= 4 layer, wp[t][ci][co] = W[co][ci][t];
//                       kind 1: data-gradient of a Cin <= 4 layer, wp[t][co][ci] = W[co][ci][8 - t]
__global__ void pack_thin_kernel(const float *w, float *wp, int Cout, int Cin, int KinP, int kind) {
    const int total = 9 * KinP * 4;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int o = e & 3, r = (e >> 2) % KinP, t = (e >> 2) / KinP;
        float v = 0.f;
        if (kind == 0) {
            if (o < Cout && r < Cin) v = w[((size_t)o * Cin + r) * 9 + t];
        } else {
            if (o < Cin && r < Cout) v = w[((size_t)r * Cin + o) * 9 + (8 - t)];
        }
        wp[e] = v;
    }
}

// ---- 7x7 form: the image-side layers of ResnetGenerator (ResNet_arch.py:52-55, :86-88) in one launch for the 49 taps.
// Same tiling (16 x 16 pixels, one thread = one output pixel x 4 channels, 16-channel chunks), a (16 + 6)^2 halo tile; the border rule
// (reflection / zeros) is applied where the halo is gathered, so no padded copy of the wide tensor exists; output grid Ho x Wo and the
// offset `pad` are free (the data-gradient with respect to a padded input is the pad = 6 zero-border case).
struct Thin7K {
    const float *x; int x_ct, x_co; int N, H, W, Cin;
    const float *wp;            // [49][KinP][4]
    int KinP;
    float *y; int y_ct, y_co, Cout, Ho, Wo;
    int pad, reflect;
    const float *bias; float alpha;
    int tiles_x, tiles_y;
};

constexpr int T7 = 16, T7W = T7 + 6;
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) conv_thin7_kernel(const Thin7K a) {
    __shared__ __attribute__((aligned(16))) float s_in[T7W * T7W * TPST];
    typedef const __attribute__((address_space(4))) f32x4 cf32x4;
    cf32x4 *wtab = (cf32x4 *)(a.wp);          // wave-uniform addresses: scalar loads (see conv_thin_kernel)
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int n = bid / a.tiles_y;
    const int y0 = ty * T7, x0 = tx * T7;
    const int py = tid / T7, px = tid - py * T7;

    f32x2_t acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
    constexpr int IN_ITEMS = T7W * T7W * 4, IN_IT = (IN_ITEMS + 255) / 256;
    int in_off[IN_IT];
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
        const int i = tid + it * 256;
        const int pix = i >> 2, q = i & 3;
        const int hr = pix / T7W, hc = pix - hr * T7W;
        int Y = y0 + hr - a.pad, X = x0 + hc - a.pad;
        if (a.reflect) {
            Y = Y < 0 ? -Y : (Y >= a.H ? 2 * a.H - 2 - Y : Y);
            X = X < 0 ? -X : (X >= a.W ? 2 * a.W - 2 - X : X);
        }
        const bool ok = (i < IN_ITEMS) & (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
        in_off[it] = ok ? (((n * a.H + Y) * a.W + X) * a.x_ct + a.x_co + q * 4) : -1;
    }
    const int nchunks = a.KinP / TNR_CK;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int c0 = chunk * TNR_CK;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < IN_IT; ++it) {
            const int i = tid + it * 256;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (in_off[it] >= 0 && c0 + (i & 3) * 4 < a.Cin) v = *reinterpret_cast<const f32x4 *>(a.x + (size_t)in_off[it] + c0);
            if (i < IN_ITEMS) *reinterpret_cast<f32x4 *>(s_in + (i >> 2) * TPST + (i & 3) * 4) = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int tr = 0; tr < 7; ++tr) {                    // (a tap row per trip: 7 x 16 x 4 FMAs unrolled)
            const float *xrow = s_in + ((py + tr) * T7W + px) * TPST;
            cf32x4 *wrow = wtab + (size_t)(tr * 7) * a.KinP + c0;
#pragma unroll
            for (int tc = 0; tc < 7; ++tc) {
#pragma unroll
                for (int c4 = 0; c4 < TNR_CK / 4; ++c4) {
                    const f32x4 xv = *reinterpret_cast<const f32x4 *>(xrow + tc * TPST + c4 * 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f32x4 w = wrow[tc * a.KinP + c4 * 4 + k];
                        const f32x2_t xx = {xv[k], xv[k]};
                        acc01 = __builtin_elementwise_fma(xx, (f32x2_t){w[0], w[1]}, acc01);
                        acc23 = __builtin_elementwise_fma(xx, (f32x2_t){w[2], w[3]}, acc23);
                    }
                }
            }
        }
    }
    const int oy = y0 + py, ox = x0 + px;
    if (oy < a.Ho && ox < a.Wo) {
        float *yp = a.y + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.y_ct + a.y_co;
        const float acc[4] = {acc01[0], acc01[1], acc23[0], acc23[1]};
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (o < a.Cout) yp[o] = (acc[o] + (a.bias != nullptr ? a.bias[o] : 0.f)) * a.alpha;
    }
}

// ---- the same launch with LANES = INPUT CHANNELS (Cin <= 64).  In conv_thin7_kernel every lane multiplies by the same weight: 784
// scalar-cache loads per tap row and chunk, whose miss latency (12.5 KB of weights per chunk against a 16 KB cache shared by the CU's
// workgroups) set the pace -- 0.79 ms per launch at 16 x 256 x 256 x 64 where its packed FMAs need 0.19 (profiles/r12f).  Here a lane
// owns ONE input channel: its 49 x 3 weights stay in registers for a whole pair of output rows, the input arrives as coalesced
// 256-byte rows (no LDS, no scalar loads in the loop) into a sliding 8 x 7 register window (one new column per output column), and
// the 64 per-channel partial sums of a pixel are added through LDS, 9 columns at a time.  One wave (= one workgroup) = two output
// rows x up to 160 columns; the window is a ring of 9 column slots so that two columns are always in flight (one wave per SIMD at
// ~290 registers: nothing else hides the L2 latency).
constexpr int T7C_D = 2;                 // columns in flight ahead of the one being multiplied
constexpr int T7C_S = 7 + T7C_D;         // window slots = phases per trip = columns per LDS reduction
constexpr int T7C_SEG = 160;             // at most this many output columns per wave (more, smaller tasks: no ragged last round)

__global__ void __launch_bounds__(64) conv_thin7c_kernel(const Thin7K a, const int nseg, const int seg) {
    constexpr int RS = 65;                                 // lanes of a quantity + 1: conflict-free in both directions
    constexpr int NQ = T7C_S * 6;                          // quantities per trip: column x 2 rows x 3 channels
    static_assert(NQ <= 64, "one lane per quantity");
    __shared__ float red[NQ * RS];
    const int lane = threadIdx.x;
    int bid = blockIdx.x;
    const int sg = bid % nseg;
    bid /= nseg;
    const int rp2 = (a.Ho + 1) >> 1;
    const int n = bid / rp2;
    const int oyb = (bid - n * rp2) * 2;                   // this wave: output rows oyb, oyb + 1, columns [xb, xe)
    const int xb = sg * seg, xe = (xb + seg < a.Wo) ? xb + seg : a.Wo;
    const bool cok = lane < a.Cin;
    float w0[49], w1[49], w2[49];     // (scalar FMAs: a packed one would want every window value in a register PAIR of its own)
#pragma unroll
    for (int t = 0; t < 49; ++t) {
        f32x4 w = {0.f, 0.f, 0.f, 0.f};
        if (cok) w = *reinterpret_cast<const f32x4 *>(a.wp + ((size_t)t * a.KinP + lane) * 4);
        w0[t] = w[0]; w1[t] = w[1]; w2[t] = w[2];
    }
    int rowoff[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int Y = oyb - a.pad + r;
        if (a.reflect) Y = Y < 0 ? -Y : (Y >= a.H ? 2 * a.H - 2 - Y : Y);
        rowoff[r] = (cok && Y >= 0 && Y < a.H) ? ((n * a.H + Y) * a.W) * a.x_ct + a.x_co + lane : -1;
    }
    auto load_col = [&](int col, float (&dst)[8]) {        // input column `col` (before the border rule) of the 8 rows
        int X = col;
        if (a.reflect) X = X < 0 ? -X : (X >= a.W ? 2 * a.W - 2 - X : X);
        const bool xok = X >= 0 && X < a.W;
#pragma unroll
        for (int r = 0; r < 8; ++r) dst[r] = (xok && rowoff[r] >= 0) ? a.x[(size_t)rowoff[r] + (size_t)X * a.x_ct] : 0.f;
    };
    // ring of T7C_S window slots: slot k holds input column xb - pad + j with j mod T7C_S == k; output column xb + g (phase g mod T7C_S)
    // multiplies slots g .. g + 6 while column g + 7 + (T7C_D - 1) is loaded into the slot that column g - 1 has just left
    float win[T7C_S][8];
#pragma unroll
    for (int k = 0; k < T7C_S - 1; ++k) load_col(xb - a.pad + k, win[k]);
    const float b0 = a.bias != nullptr ? a.bias[0] : 0.f, b1 = (a.bias != nullptr && a.Cout > 1) ? a.bias[1] : 0.f,
                b2 = (a.bias != nullptr && a.Cout > 2) ? a.bias[2] : 0.f;
    for (int ox0 = xb; ox0 < xe; ox0 += T7C_S) {
#pragma unroll
        for (int ph = 0; ph < T7C_S; ++ph) {
            load_col(ox0 + ph - a.pad + T7C_S - 1, win[(ph + T7C_S - 1) % T7C_S]);
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int tr = 0; tr < 7; ++tr) {
#pragma unroll
                for (int tc = 0; tc < 7; ++tc) {
                    const int t = tr * 7 + tc;
                    const float xa = win[(ph + tc) % T7C_S][tr], xc = win[(ph + tc) % T7C_S][tr + 1];
                    a0 = __builtin_fmaf(xa, w0[t], a0);
                    a1 = __builtin_fmaf(xa, w1[t], a1);
                    a2 = __builtin_fmaf(xa, w2[t], a2);
                    c0 = __builtin_fmaf(xc, w0[t], c0);
                    c1 = __builtin_fmaf(xc, w1[t], c1);
                    c2 = __builtin_fmaf(xc, w2[t], c2);
                }
            }
            float *d = red + (ph * 6) * RS + lane;
            d[0] = a0; d[RS] = a1; d[2 * RS] = a2;
            d[3 * RS] = c0; d[4 * RS] = c1; d[5 * RS] = c2;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < NQ) {                                   // quantity (column ph, row, channel o) = lane: the 64 channels, in lane order
            const float *src = red + lane * RS;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int l = 0; l < 64; l += 4) {
                s0 += src[l]; s1 += src[l + 1]; s2 += src[l + 2]; s3 += src[l + 3];
            }
            const float sum = (s0 + s1) + (s2 + s3);
            const int ph = lane / 6, rr = (lane - ph * 6) / 3, o = lane % 3;
            const int ox = ox0 + ph, oy = oyb + rr;
            if (ox < xe && oy < a.Ho && o < a.Cout) {
                float *yp = a.y + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.y_ct + a.y_co;
                yp[o] = (sum + (o == 0 ? b0 : (o == 1 ? b1 : b2))) * a.alpha;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// [49][KinP][4] <- OIHW (7x7): kind 0: forward of a Cout <= 4 layer, wp[t][ci][co] = W[co][ci][t];
//                              kind 1: data-gradient of a Cin <= 4 layer, wp[t][co][ci] = W[co][ci][48 - t]
__global__ void pack_thin7_kernel(const float *w, float *wp, int Cout, int Cin, int KinP, int kind) {
    const int total = 49 * KinP * 4;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int o = e & 3, r = (e >> 2) % KinP, t = (e >> 2) / KinP;
        float v = 0.f;
        if (kind == 0) {
            if (o < Cout && r < Cin) v = w[((size_t)o * Cin + r) * 49 + t];
        } else {
            if (o < Cin && r < Cout) v = w[((size_t)r * Cin + o) * 49 + (48 - t)];
        }
        wp[e] = v;
    }
}

}  // namespace

extern "C" int64_t tnr_conv_thin7_pack_floats(int32_t reduce_channels) {
    return (int64_t)49 * tnr_round_up(reduce_channels, TNR_CK) * 4;
}

extern "C" int tnr_conv_thin7_pack(const float *w, float *wp, int32_t Cout, int32_t Cin, int32_t dgrad, void *stream) {
    TNR_REQUIRE(w != nullptr && wp != nullptr && Cout > 0 && Cin > 0, "conv_thin7_pack: bad arguments");
    TNR_REQUIRE(dgrad ? Cin <= 4 : Cout <= 4, "conv_thin7_pack: the thin side must have <= 4 channels");
    const int KinP = tnr_round_up(dgrad ? Cout : Cin, TNR_CK);
    hipLaunchKernelGGL(pack_thin7_kernel, dim3(tnr_cdiv(49 * KinP * 4, 256)), dim3(256), 0, (hipStream_t)stream, w, wp, Cout, Cin, KinP,
                       dgrad ? 1 : 0);
    return tnr_check_launch("conv_thin7_pack");
}

extern "C" int tnr_conv_thin7(tnr_view x, int32_t N, int32_t H, int32_t W, int32_t Cin, const float *wp, tnr_view y, int32_t Ho, int32_t Wo,
                              int32_t Cout, int32_t pad, int32_t reflect, const float *bias, float alpha, void *stream) {
    TNR_REQUIRE(x.ptr && y.ptr && wp && N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "conv_thin7: bad arguments");
    TNR_REQUIRE(Cout >= 1 && Cout <= 4 && (Cin % 4) == 0 && (x.ctot % 4) == 0 && (x.coff % 4) == 0, "conv_thin7: Cout <= 4, Cin %% 4 == 0");
    TNR_REQUIRE(pad >= 0 && pad <= 6 && (!reflect || (pad <= 3 && Ho == H && Wo == W && H > pad && W > pad)),
                "conv_thin7: pad in [0, 6]; reflection borders need pad <= 3 < H, W and an H x W output");
    TNR_REQUIRE(Ho <= H + 2 * pad - 6 + 12 && Wo <= W + 2 * pad - 6 + 12, "conv_thin7: output grid beyond the reach of the input");
    TNR_REQUIRE((int64_t)N * H * W * x.ctot < (1LL << 31), "conv_thin7: input buffer above 2^31 elements");
    Thin7K k;
    k.x = x.ptr; k.x_ct = x.ctot; k.x_co = x.coff; k.N = N; k.H = H; k.W = W; k.Cin = Cin;
    k.wp = wp; k.KinP = tnr_round_up(Cin, TNR_CK);
    k.y = y.ptr; k.y_ct = y.ctot; k.y_co = y.coff; k.Cout = Cout; k.Ho = Ho; k.Wo = Wo;
    k.pad = pad; k.reflect = reflect ? 1 : 0;
    k.bias = bias; k.alpha = alpha;
    k.tiles_x = tnr_cdiv(Wo, T7); k.tiles_y = tnr_cdiv(Ho, T7);
    const int64_t tiles = (int64_t)k.tiles_x * k.tiles_y * N;
    TNR_REQUIRE(tiles < (1LL << 31), "conv_thin7: grid too large");
    static const bool lanes_c = [] { const char *e = std::getenv("TNR_THIN7_LANES"); return e == nullptr || e[0] != 'p'; }();      // p: lanes = pixels (A/B switch)
    if (lanes_c && Cin <= 64 && Cout <= 3) {      // (a lane keeps 49 x 3 weights: a 4-channel output takes the lanes = pixels kernel below)
        const int nseg = tnr_cdiv(Wo, T7C_SEG), seg = tnr_cdiv(Wo, nseg);
        const int64_t waves = (int64_t)N * tnr_cdiv(Ho, 2) * nseg;
        TNR_REQUIRE(waves < (1LL << 31), "conv_thin7: grid too large");
        hipLaunchKernelGGL(conv_thin7c_kernel, dim3((unsigned)waves), dim3(64), 0, (hipStream_t)stream, k, nseg, seg);
        return tnr_check_launch("conv_thin7");
    }
    hipLaunchKernelGGL(conv_thin7_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, k);
    return tnr_check_launch("conv_thin7");
}

extern "C" int64_t tnr_conv_thin_pack_floats(int32_t reduce_channels) {
    return (int64_t)9 * tnr_round_up(reduce_channels, TNR_CK) * 4;
}

extern "C" int tnr_conv_thin_pack(const float *w, float *wp, int32_t Cout, int32_t Cin, int32_t dgrad, void *stream) {
    TNR_REQUIRE(w != nullptr && wp != nullptr && Cout > 0 && Cin > 0, "conv_thin_pack: bad arguments");
    TNR_REQUIRE(dgrad ? Cin <= 4 : Cout <= 4, "conv_thin_pack: the thin side must have <= 4 channels");
    const int KinP = tnr_round_up(dgrad ? Cout : Cin, TNR_CK);
    hipLaunchKernelGGL(pack_thin_kernel, dim3(tnr_cdiv(9 * KinP * 4, 256)), dim3(256), 0, (hipStream_t)stream, w, wp, Cout, Cin, KinP,
                       dgrad ? 1 : 0);
    return tnr_check_launch("conv_thin_pack");
}

extern "C" int tnr_conv_thin(tnr_view x, int32_t N, int32_t H, int32_t W, int32_t Cin, const float *wp, tnr_view y, int32_t Cout,
                             const float *bias, float alpha, void *stream) {
    TNR_REQUIRE(x.ptr && y.ptr && wp && N > 0 && H > 0 && W > 0, "conv_thin: bad arguments");
    TNR_REQUIRE(Cout >= 1 && Cout <= 4 && (Cin % 4) == 0 && (x.ctot % 4) == 0 && (x.coff % 4) == 0, "conv_thin: Cout <= 4, Cin %% 4 == 0");
    TNR_REQUIRE((int64_t)N * H * W * x.ctot < (1LL << 31), "conv_thin: input buffer above 2^31 elements");
    ThinK k;
    k.x = x.ptr; k.x_ct = x.ctot; k.x_co = x.coff; k.N = N; k.H = H; k.W = W; k.Cin = Cin;
    k.wp = wp; k.KinP = tnr_round_up(Cin, TNR_CK);
    k.y = y.ptr; k.y_ct = y.ctot; k.y_co = y.coff; k.Cout = Cout;
    k.bias = bias; k.alpha = alpha;
    k.tiles_x = tnr_cdiv(W, TT); k.tiles_y = tnr_cdiv(H, TT);
    const int64_t tiles = (int64_t)k.tiles_x * k.tiles_y * N;
    TNR_REQUIRE(tiles < (1LL << 31), "conv_thin: grid too large");
    hipLaunchKernelGGL(conv_thin_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, k);
    return tnr_check_launch("conv_thin");
}
