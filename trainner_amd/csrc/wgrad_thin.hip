// Weight gradient of a 3x3 s1 p1 layer that has <= 4 channels on one side (the RGB-image layers: D conv0 and
// G's first conv, 3 -> 64; G's last conv, 64 -> 3) on the vector ALUs.  The matrix-core kernel pads the 3 channels
// to a 32-wide block and runs 1.3 ms per launch at 16 x 512 x 512; the real work is 27 FMAs per element of the
// 64-channel operand, so the launch is bounded by reading that operand once (1.07 GB).
//   big   : the >= 16-channel operand (g for a 3 -> C layer, x for a C -> 3 layer), read as float4 of 4 channels
//   small : the 4-channel NHWC4 operand (3 valid), three rows at a time in LDS
//   dW[big channel][small channel][tap] = sum_pixels big[p] * small[p + tap]      (flip = 0: big = g, small = x)
//                                       = sum_pixels big[p] * small[p - tap]      (flip = 1: big = x, small = g)
// One workgroup walks a contiguous range of image rows; thread = (pixel lane, 4 big channels) keeps 4 x 27
// accumulators; pixel lanes are combined by shuffles + LDS in a fixed order, workgroups by a reduce launch over
// [workgroup] partial slabs (deterministic).
#include "common.h"

namespace {

constexpr int WT_MAXW = 512;     // pixels of a row staged per pass
constexpr int WT_GRID = 1024;

struct WThinK {
    const float *big; int b_ct, b_co, Cb;
    const float *small;              // NHWC4
    int N, H, W, flip;
    float *ws;                       // [grid][Cb][28]: 27 weight partials + 1 big-side bias partial
    float *ws_small;                 // [grid][4]: small-side bias partials
    int rows_total, rows_per_wg;
};

template <int TPP>   // threads per pixel = Cb / 4
__global__ void __launch_bounds__(256) wgrad_thin_kernel(const WThinK a) {
    constexpr int PL = 256 / TPP;                              // pixel lanes
    __shared__ __attribute__((aligned(16))) float s_small[3 * (WT_MAXW + 2) * 4];
    __shared__ float s_red[4 * TPP * 112];
    const int tid = threadIdx.x;
    const int c4 = tid % TPP, pl = tid / TPP;
    float acc[4][27];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 27; ++j) acc[k][j] = 0.f;
    f32x4 accb = {0.f, 0.f, 0.f, 0.f};                         // sum of the big operand (bias of a 3 -> C layer)
    f32x4 accs = {0.f, 0.f, 0.f, 0.f};                         // sum of the small operand (bias of a C -> 3 layer)
    const int r_begin = blockIdx.x * a.rows_per_wg;
    int r_end = r_begin + a.rows_per_wg;
    if (r_end > a.rows_total) r_end = a.rows_total;
    for (int row = r_begin; row < r_end; ++row) {
        const int n = row / a.H, y = row - n * a.H;
        for (int x0 = 0; x0 < a.W; x0 += WT_MAXW) {
            const int wc = (a.W - x0 < WT_MAXW) ? a.W - x0 : WT_MAXW;
            __syncthreads();
            for (int i = tid; i < 3 * (wc + 2); i += 256) {     // small rows y-1 .. y+1, columns x0-1 .. x0+wc
                const int rr = i / (wc + 2), cc = i - rr * (wc + 2);
                const int Y = y + rr - 1, X = x0 + cc - 1;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (Y >= 0 && Y < a.H && X >= 0 && X < a.W) v = *reinterpret_cast<const f32x4 *>(a.small + (((size_t)n * a.H + Y) * a.W + X) * 4);
                *reinterpret_cast<f32x4 *>(s_small + (rr * (WT_MAXW + 2) + cc) * 4) = v;
            }
            __syncthreads();
            for (int x = pl; x < wc; x += PL) {
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(a.big + (((size_t)n * a.H + y) * a.W + x0 + x) * a.b_ct + a.b_co + c4 * 4);
                accb += bv;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int ty = t / 3, tx = t - ty * 3;
                    const int rr = a.flip ? 2 - ty : ty, cc = a.flip ? 2 - tx : tx;
                    const f32x4 sv = *reinterpret_cast<const f32x4 *>(s_small + (rr * (WT_MAXW + 2) + x + cc) * 4);
                    if (t == 4 && c4 == 0) accs += sv;          // centre tap: every pixel exactly once
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int ci = 0; ci < 3; ++ci) acc[k][t * 3 + ci] = __builtin_fmaf(bv[k], sv[ci], acc[k][t * 3 + ci]);
                }
            }
        }
    }
    // ---- combine the pixel lanes: inside a wave by xor-shuffles (lanes with equal c4 are TPP apart), then the
    // 4 waves through LDS in wave order
#pragma unroll
    for (int off = TPP; off < 64; off <<= 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int j = 0; j < 27; ++j) acc[k][j] += __shfl_xor(acc[k][j], off);
            accb[k] += __shfl_xor(accb[k], off);
            accs[k] += __shfl_xor(accs[k], off);
        }
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    if (lane < TPP) {
        float *dst = s_red + (wave * TPP + lane) * 112;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int j = 0; j < 27; ++j) dst[k * 27 + j] = acc[k][j];
            dst[108 + k] = accb[k];
        }
    }
    __shared__ float s_accs[4][4];
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s_accs[wave][k] = accs[k];
    }
    __syncthreads();
    // threads 0 .. TPP*112-1: one output each
    for (int i = tid; i < TPP * 112; i += 256) {
        const float v = ((s_red[i] + s_red[TPP * 112 + i]) + s_red[2 * TPP * 112 + i]) + s_red[3 * TPP * 112 + i];
        const int c4o = i / 112, j = i - c4o * 112;
        // slab layout [Cb][28]: channel = 4*c4o + k
        if (j < 108) a.ws[((size_t)blockIdx.x * a.Cb + c4o * 4 + j / 27) * 28 + j % 27] = v;
        else a.ws[((size_t)blockIdx.x * a.Cb + c4o * 4 + (j - 108)) * 28 + 27] = v;
    }
    if (tid < 4) a.ws_small[(size_t)blockIdx.x * 4 + tid] = ((s_accs[0][tid] + s_accs[1][tid]) + s_accs[2][tid]) + s_accs[3][tid];
}

struct WThinRedK {
    const float *ws; const float *ws_small; int grid, Cb, Cs, flip;
    float *dw; float *db; float alpha, beta;
};

// one thread per (big channel, j): j < 27 weight element, j == 27 bias of the big side; plus Cs threads for the
// small-side bias.  Workgroup partials are added in index order, 8 interleaved lanes would not pay here (<= 1024 terms).
__global__ void __launch_bounds__(256) wgrad_thin_reduce_kernel(const WThinRedK a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nmain = a.Cb * 28;
    if (i < nmain) {
        const int cb = i / 28, j = i - cb * 28;
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
        int g = 0;
        for (; g + 3 < a.grid; g += 4) {
            p0 += a.ws[((size_t)g * a.Cb + cb) * 28 + j];
            p1 += a.ws[((size_t)(g + 1) * a.Cb + cb) * 28 + j];
            p2 += a.ws[((size_t)(g + 2) * a.Cb + cb) * 28 + j];
            p3 += a.ws[((size_t)(g + 3) * a.Cb + cb) * 28 + j];
        }
        for (; g < a.grid; ++g) p0 += a.ws[((size_t)g * a.Cb + cb) * 28 + j];
        const float sum = (p0 + p1) + (p2 + p3);
        if (j < 27) {
            const int t = j / 3, cs = j - t * 3;
            if (cs < a.Cs) {
                // flip = 0: dW[cb][cs][t] of a Cs -> Cb layer;  flip = 1: dW[cs][cb][t] of a Cb -> Cs layer
                const size_t o = a.flip ? ((size_t)cs * a.Cb + cb) * 9 + t : ((size_t)cb * a.Cs + cs) * 9 + t;
                a.dw[o] = (a.beta != 0.f ? a.beta * a.dw[o] : 0.f) + a.alpha * sum;
            }
        } else if (!a.flip && a.db != nullptr) {
            a.db[cb] = (a.beta != 0.f ? a.beta * a.db[cb] : 0.f) + a.alpha * sum;
        }
    } else if (i < nmain + a.Cs && a.flip && a.db != nullptr) {
        const int cs = i - nmain;
        float sum = 0.f;
        for (int g = 0; g < a.grid; ++g) sum += a.ws_small[(size_t)g * 4 + cs];
        a.db[cs] = (a.beta != 0.f ? a.beta * a.db[cs] : 0.f) + a.alpha * sum;
    }
}

int thin_grid(int rows) { return rows < WT_GRID ? rows : WT_GRID; }

}  // namespace

extern "C" int64_t tnr_wgrad_thin_workspace_bytes(int32_t N, int32_t H, int32_t Cbig) {
    const int64_t g = thin_grid(N * H);
    return (g * Cbig * 28 + g * 4) * (int64_t)sizeof(float);
}

extern "C" int tnr_wgrad_thin(tnr_view big, tnr_view small, int32_t N, int32_t H, int32_t W, int32_t Cbig, int32_t Csmall, int32_t flip,
                              float *dw, float *db, float alpha, float beta, float *ws, int64_t ws_bytes, void *stream) {
    TNR_REQUIRE(big.ptr && small.ptr && dw && ws && N > 0 && H > 0 && W > 0, "wgrad_thin: bad arguments");
    TNR_REQUIRE(Cbig == 16 || Cbig == 32 || Cbig == 64, "wgrad_thin: the wide side must have 16, 32 or 64 channels (got %d)", Cbig);
    TNR_REQUIRE(Csmall >= 1 && Csmall <= 3 && small.ctot == 4 && small.coff == 0, "wgrad_thin: the thin side is an NHWC4 image with <= 3 channels");
    TNR_REQUIRE((big.ctot % 4) == 0 && (big.coff % 4) == 0, "wgrad_thin: wide view must be 4-channel aligned");
    TNR_REQUIRE(tnr_wgrad_thin_workspace_bytes(N, H, Cbig) <= ws_bytes, "wgrad_thin: workspace too small");
    TNR_REQUIRE((int64_t)N * H * W * big.ctot < (1LL << 40), "wgrad_thin: buffer too large");
    WThinK k;
    k.big = big.ptr; k.b_ct = big.ctot; k.b_co = big.coff; k.Cb = Cbig;
    k.small = small.ptr; k.N = N; k.H = H; k.W = W; k.flip = flip ? 1 : 0;
    const int grid = thin_grid(N * H);
    k.ws = ws; k.ws_small = ws + (size_t)grid * Cbig * 28;
    k.rows_total = N * H; k.rows_per_wg = tnr_cdiv(N * H, grid);
    const int used = tnr_cdiv(N * H, k.rows_per_wg);        // workgroups that actually own rows
    hipStream_t s = (hipStream_t)stream;
    switch (Cbig) {
        case 16: hipLaunchKernelGGL(wgrad_thin_kernel<4>, dim3(used), dim3(256), 0, s, k); break;
        case 32: hipLaunchKernelGGL(wgrad_thin_kernel<8>, dim3(used), dim3(256), 0, s, k); break;
        default: hipLaunchKernelGGL(wgrad_thin_kernel<16>, dim3(used), dim3(256), 0, s, k); break;
    }
    int rc = tnr_check_launch("wgrad_thin");
    if (rc != TNR_OK) return rc;
    WThinRedK r;
    r.ws = ws; r.ws_small = k.ws_small; r.grid = used; r.Cb = Cbig; r.Cs = Csmall; r.flip = k.flip;
    r.dw = dw; r.db = db; r.alpha = alpha; r.beta = beta;
    hipLaunchKernelGGL(wgrad_thin_reduce_kernel, dim3(tnr_cdiv(Cbig * 28 + 4, 256)), dim3(256), 0, s, r);
    return tnr_check_launch("wgrad_thin_reduce");
}
