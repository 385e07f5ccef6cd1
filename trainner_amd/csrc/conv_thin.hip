// 3x3 s1 p1 convolution with <= 4 OUTPUT channels (G's last conv 64 -> 3, and the data-gradients that end in the
// RGB image: VGG conv1_1 and D conv0, 64 -> 3): one thread = one output pixel x 4 channels on the vector ALUs.
// The matrix-core kernel pads 3 output channels to a 32-wide MFMA tile (8x the work: 1.3 ms per launch at
// 16 x 512 x 512); here the launch is bounded by reading the 64-channel input once (1.07 GB, ~0.25 ms).
//   tile 16 x 16 pixels per workgroup, input halo tile (18 x 18) x 16 channels per chunk in LDS (20-dword pixel
//   stride: conflict-free ds_read_b128), the chunk's [tap][16 ch][4 cout] weights in LDS read as broadcasts;
//   fp32 FMA chain per output in (chunk, tap, channel) order.
#include "common.h"

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

}  // namespace

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
