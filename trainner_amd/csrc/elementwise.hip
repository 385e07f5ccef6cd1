// HBM-bound data movement around the convolutions: NCHW<->NHWC boundary transposes, nearest-x2
// backward, PixelShuffle (depth_to_space, PyTorch order) forward/backward, 2x2 max-pool,
// masked scaling and fills.  All kernels are grid-stride over float4 (16 B/lane) where the layout allows.
#include "common.h"
#include "gauss_noise.h"

namespace {

constexpr int EW_BLOCK = 256;
inline unsigned ew_grid(int64_t n) {
    int64_t b = tnr_cdiv64(n, EW_BLOCK);
    if (b > 256 * 8) b = 256 * 8;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// dst[n,y,x, co + c] = src[n,c,y,x]*scale[c] + shift[c] for c < C, 0 for C <= c < Cpad.
// One thread per (pixel, 4-channel group); reads are coalesced along x for each channel plane.
__global__ void nchw_to_nhwc_kernel(const float *src, int N, int C, int H, int W, float *dst, int ct, int co, int Cpad,
                                    const float *scale, const float *shift) {
    const int64_t HW = (int64_t)H * W;
    const int groups = Cpad / 4;
    const int64_t total = (int64_t)N * HW * groups;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = e % ((int64_t)N * HW);
        const int gidx = (int)(e / ((int64_t)N * HW));
        const int64_t n = pix / HW, hw = pix - n * HW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = gidx * 4 + k;
            if (c < C) {
                float t = src[(n * C + c) * HW + hw];
                if (scale) t = t * scale[c] + shift[c];
                v[k] = t;
            }
        }
        *reinterpret_cast<f32x4 *>(dst + pix * ct + co + gidx * 4) = v;
    }
}

// dst[n,c,y,x] (+)= src[n,y,x,co+c] * scale[c]
__global__ void nhwc_to_nchw_kernel(const float *src, int ct, int co, int N, int C, int H, int W, float *dst,
                                    const float *scale, int accumulate) {
    const int64_t HW = (int64_t)H * W;
    const int64_t total = (int64_t)N * C * HW;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t hw = e % HW;
        const int64_t q = e / HW;
        const int c = (int)(q % C);
        const int64_t n = q / C;
        float v = src[(n * HW + hw) * ct + co + c];
        if (scale) v *= scale[c];
        if (accumulate) v += dst[e];
        dst[e] = v;
    }
}

// gx[n,y,x,c] = sum of the 2x2 block of gup, optionally * (mask > 0 ? 1 : slope)
__global__ void upsample2x_bwd_kernel(const float *gup, int u_ct, int u_co, float *gx, int x_ct, int x_co, int N, int H,
                                      int W, int C, const float *mask, int m_ct, int m_co, float mslope) {
    const int c4n = C / 4;
    const int64_t total = (int64_t)N * H * W * c4n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % c4n);
        const int64_t pix = e / c4n;
        const int x = (int)(pix % W);
        const int64_t q = pix / W;
        const int y = (int)(q % H);
        const int64_t n = q / H;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int64_t up = (n * 2 * H + 2 * y + dy) * (2 * W) + 2 * x + dx;
                s += *reinterpret_cast<const f32x4 *>(gup + up * u_ct + u_co + c4 * 4);
            }
        if (mask) {
            const f32x4 mv = *reinterpret_cast<const f32x4 *>(mask + pix * m_ct + m_co + c4 * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] *= (mv[k] > 0.f ? 1.f : mslope);
        }
        *reinterpret_cast<f32x4 *>(gx + pix * x_ct + x_co + c4 * 4) = s;
    }
}

// PixelShuffle(2), PyTorch order (block.py:434-460): y[n, 2h+dy, 2w+dx, c] = x[n, h, w, c*4 + dy*2 + dx]
__global__ void depth_to_space_kernel(const float *x, int x_ct, int x_co, float *y, int y_ct, int y_co, int N, int H,
                                      int W, int Cout) {
    const int64_t total = (int64_t)N * H * W * Cout;  // one thread per (input pixel, out channel): reads a float4
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % Cout);
        const int64_t pix = e / Cout;
        const int w = (int)(pix % W);
        const int64_t q = pix / W;
        const int h = (int)(q % H);
        const int64_t n = q / H;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(x + pix * x_ct + x_co + c * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t op = (n * 2 * H + 2 * h + (k >> 1)) * (2 * W) + 2 * w + (k & 1);
            y[op * y_ct + y_co + c] = v[k];
        }
    }
}

// gx[n,h,w,c*4+k] = gy[n,2h+dy,2w+dx,c] * (mask_at_gy_position > 0 ? 1 : slope)
__global__ void space_to_depth_bwd_kernel(const float *gy, int g_ct, int g_co, float *gx, int x_ct, int x_co, int N, int H,
                                          int W, int Cout, const float *mask, int m_ct, int m_co, float mslope) {
    const int64_t total = (int64_t)N * H * W * Cout;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % Cout);
        const int64_t pix = e / Cout;
        const int w = (int)(pix % W);
        const int64_t q = pix / W;
        const int h = (int)(q % H);
        const int64_t n = q / H;
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t op = (n * 2 * H + 2 * h + (k >> 1)) * (2 * W) + 2 * w + (k & 1);
            float g = gy[op * g_ct + g_co + c];
            if (mask) g *= (mask[op * m_ct + m_co + c] > 0.f ? 1.f : mslope);
            v[k] = g;
        }
        *reinterpret_cast<f32x4 *>(gx + pix * x_ct + x_co + c * 4) = v;
    }
}

// patch matrix of a k x k convolution: out[pixel][(ky*k + kx)*C + c] (float4 granularity)
__global__ void im2col_kernel(const float *x, int x_ct, int x_co, int N, int H, int W, int C, int k, int stride, int pad, int Ho,
                              int Wo, float *out) {
    const int c4n = C / 4, kk = k * k;
    const int64_t total = (int64_t)N * Ho * Wo * kk * c4n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % c4n);
        int64_t q = e / c4n;
        const int t = (int)(q % kk);
        const int64_t pix = q / kk;
        const int ox = (int)(pix % Wo);
        q = pix / Wo;
        const int oy = (int)(q % Ho);
        const int64_t n = q / Ho;
        const int Y = oy * stride - pad + t / k, X = ox * stride - pad + t % k;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (Y >= 0 && Y < H && X >= 0 && X < W) v = *reinterpret_cast<const f32x4 *>(x + ((n * H + Y) * W + X) * x_ct + x_co + c4 * 4);
        *reinterpret_cast<f32x4 *>(out + (pix * kk + t) * C + c4 * 4) = v;
    }
}

__global__ void maxpool2_fwd_kernel(const float *x, int x_ct, int x_co, float *y, int y_ct, int y_co, int N, int H, int W,
                                    int C) {
    const int Ho = H / 2, Wo = W / 2, c4n = C / 4;
    const int64_t total = (int64_t)N * Ho * Wo * c4n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % c4n);
        const int64_t pix = e / c4n;
        const int ox = (int)(pix % Wo);
        const int64_t q = pix / Wo;
        const int oy = (int)(q % Ho);
        const int64_t n = q / Ho;
        f32x4 best;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t ip = (n * H + 2 * oy + (k >> 1)) * W + 2 * ox + (k & 1);
            const f32x4 v = *reinterpret_cast<const f32x4 *>(x + ip * x_ct + x_co + c4 * 4);
            if (k == 0) best = v;
            else
#pragma unroll
                for (int j = 0; j < 4; ++j) best[j] = v[j] > best[j] ? v[j] : best[j];
        }
        *reinterpret_cast<f32x4 *>(y + pix * y_ct + y_co + c4 * 4) = best;
    }
}

// x is the pool input (post-ReLU activation): the first maximum in window scan order takes the
// gradient (aten max_pool2d_with_indices_backward), then ReLU's backward zeroes it where x <= 0.
__global__ void maxpool2_bwd_kernel(const float *gy, int g_ct, int g_co, const float *x, int x_ct, int x_co, float *gx,
                                    int gx_ct, int gx_co, int N, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2, c4n = C / 4;
    const int64_t total = (int64_t)N * Ho * Wo * c4n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % c4n);
        const int64_t pix = e / c4n;
        const int ox = (int)(pix % Wo);
        const int64_t q = pix / Wo;
        const int oy = (int)(q % Ho);
        const int64_t n = q / Ho;
        const f32x4 g = *reinterpret_cast<const f32x4 *>(gy + pix * g_ct + g_co + c4 * 4);
        f32x4 v[4];
        int64_t ip[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ip[k] = (n * H + 2 * oy + (k >> 1)) * W + 2 * ox + (k & 1);
            v[k] = *reinterpret_cast<const f32x4 *>(x + ip[k] * x_ct + x_co + c4 * 4);
        }
        f32x4 o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int arg = 0;
            float best = v[0][j];
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (v[k][j] > best) {
                    best = v[k][j];
                    arg = k;
                }
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k][j] = (k == arg && best > 0.f) ? g[j] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4 *>(gx + ip[k] * gx_ct + gx_co + c4 * 4) = o[k];
    }
}

__global__ void axpby_kernel(float *dst, int d_ct, int d_co, const float *src, int s_ct, int s_co, int64_t pixels, int C,
                             float a, float b) {
    const int c4n = C / 4;
    const int64_t total = pixels * c4n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % c4n);
        const int64_t pix = e / c4n;
        f32x4 s = *reinterpret_cast<const f32x4 *>(src + pix * s_ct + s_co + c4 * 4);
        float *dp = dst + pix * d_ct + d_co + c4 * 4;
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        if (b != 0.f) d = *reinterpret_cast<const f32x4 *>(dp);
        *reinterpret_cast<f32x4 *>(dp) = a * s + b * d;
    }
}

// dst = (src or 1) * ESRGAN+ noise multiplier (gauss_noise.h): same counter as the convolution epilogues
__global__ void gauss_mult_kernel(float *dst, int d_ct, int d_co, const float *src, int s_ct, int s_co, int64_t pixels, int C, float sigma,
                                  unsigned k0, unsigned k1, unsigned pix0) {
    const int c4n = C / 4;
    const int64_t total = pixels * c4n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % c4n);
        const int64_t pix = e / c4n;
        f32x4 v = {1.f, 1.f, 1.f, 1.f};
        if (src != nullptr) v = *reinterpret_cast<const f32x4 *>(src + pix * s_ct + s_co + c4 * 4);
        v *= tnr_gauss_mult4(((unsigned)pix + pix0) * (unsigned)c4n + (unsigned)c4, k0, k1, sigma);
        *reinterpret_cast<f32x4 *>(dst + pix * d_ct + d_co + c4 * 4) = v;
    }
}

__global__ void mask_mul_kernel(float *g, int g_ct, int g_co, const float *y, int y_ct, int y_co, int64_t pixels, int C,
                                float mslope) {
    const int c4n = C / 4;
    const int64_t total = pixels * c4n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % c4n);
        const int64_t pix = e / c4n;
        float *gp = g + pix * g_ct + g_co + c4 * 4;
        f32x4 gv = *reinterpret_cast<const f32x4 *>(gp);
        const f32x4 yv = *reinterpret_cast<const f32x4 *>(y + pix * y_ct + y_co + c4 * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) gv[k] *= (yv[k] > 0.f ? 1.f : mslope);
        *reinterpret_cast<f32x4 *>(gp) = gv;
    }
}

// ---- bilinear x2, align_corners = False (F.interpolate(scale_factor=2, mode='bilinear'): UNetDiscriminator,
// discriminators.py:745-769).  Source coordinate of output index o: max((o + 0.5) / 2 - 0.5, 0): even o = 2i blends
// 0.25 x[i-1] + 0.75 x[i], odd o = 2i+1 blends 0.75 x[i] + 0.25 x[i+1], indices clamped at the border (where the
// clamped source makes the blend collapse onto the border sample).
__device__ __forceinline__ void bil_taps(int o, int n, int &i0, int &i1, float &w0, float &w1) {
    const int i = o >> 1;
    if (o & 1) {
        i0 = i;
        i1 = i + 1 < n ? i + 1 : n - 1;
        w0 = 0.75f;
        w1 = 0.25f;
    } else {
        i0 = i > 0 ? i - 1 : 0;
        i1 = i;
        w0 = 0.25f;
        w1 = 0.75f;
    }
}

__global__ void bilinear2x_fwd_kernel(const float *x, int x_ct, int x_co, float *y, int y_ct, int y_co, int N, int H, int W, int C) {
    const int c4n = C / 4;
    const int H2 = 2 * H, W2 = 2 * W;
    const int64_t total = (int64_t)N * H2 * W2 * c4n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % c4n);
        const int64_t pix = e / c4n;
        const int ox = (int)(pix % W2);
        const int64_t q = pix / W2;
        const int oy = (int)(q % H2);
        const int64_t n = q / H2;
        int y0, y1, x0, x1;
        float wy0, wy1, wx0, wx1;
        bil_taps(oy, H, y0, y1, wy0, wy1);
        bil_taps(ox, W, x0, x1, wx0, wx1);
        const float *b = x + x_co + c4 * 4;
        const f32x4 v00 = *reinterpret_cast<const f32x4 *>(b + ((n * H + y0) * W + x0) * x_ct);
        const f32x4 v01 = *reinterpret_cast<const f32x4 *>(b + ((n * H + y0) * W + x1) * x_ct);
        const f32x4 v10 = *reinterpret_cast<const f32x4 *>(b + ((n * H + y1) * W + x0) * x_ct);
        const f32x4 v11 = *reinterpret_cast<const f32x4 *>(b + ((n * H + y1) * W + x1) * x_ct);
        // PyTorch's order: horizontal blend inside each row, then the vertical blend of the two rows
        const f32x4 r = wy0 * (wx0 * v00 + wx1 * v01) + wy1 * (wx0 * v10 + wx1 * v11);
        *reinterpret_cast<f32x4 *>(y + pix * y_ct + y_co + c4 * 4) = r;
    }
}

// adjoint of bilinear2x_fwd: gx[i] gathers, per axis, 0.75 (g[2i] + g[2i+1]) + 0.25 (g[2i-1] + g[2i+2]) with the
// out-of-range neighbours folded back onto the border sample (g[-1] -> the clamped tap of g[0], i.e. + 0.25 g[0]).
// Writes the plain gradient (gx, may be null) and/or the gradient times LeakyReLU'(mask) (gz, may be null).
__device__ __forceinline__ void bil_adj(int i, int n, int o[4], float w[4]) {
    o[0] = 2 * i - 1; o[1] = 2 * i; o[2] = 2 * i + 1; o[3] = 2 * i + 2;
    w[0] = 0.25f; w[1] = 0.75f; w[2] = 0.75f; w[3] = 0.25f;
    if (i == 0) { o[0] = 0; }                       // output 0 puts its clamped 0.25 tap on x[0] as well
    if (i == n - 1) { o[3] = 2 * n - 1; }           // output 2n-1 likewise on x[n-1]
}

__global__ void bilinear2x_bwd_kernel(const float *gy, int g_ct, int g_co, float *gx, int x_ct, int x_co, float *gz, int z_ct,
                                      int z_co, const float *mask, int m_ct, int m_co, float mslope, int N, int H, int W, int C) {
    const int c4n = C / 4;
    const int W2 = 2 * W;
    const int64_t total = (int64_t)N * H * W * c4n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % c4n);
        const int64_t pix = e / c4n;
        const int x = (int)(pix % W);
        const int64_t q = pix / W;
        const int y = (int)(q % H);
        const int64_t n = q / H;
        int oy[4], ox[4];
        float wy[4], wx[4];
        bil_adj(y, H, oy, wy);
        bil_adj(x, W, ox, wx);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 4; ++b)
                r += wx[b] * *reinterpret_cast<const f32x4 *>(gy + ((n * 2 * H + oy[a]) * W2 + ox[b]) * g_ct + g_co + c4 * 4);
            s += wy[a] * r;
        }
        if (gx) *reinterpret_cast<f32x4 *>(gx + pix * x_ct + x_co + c4 * 4) = s;
        if (gz) {
            const f32x4 mv = *reinterpret_cast<const f32x4 *>(mask + pix * m_ct + m_co + c4 * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] *= (mv[k] > 0.f ? 1.f : mslope);
            *reinterpret_cast<f32x4 *>(gz + pix * z_ct + z_co + c4 * 4) = s;
        }
    }
}

// dst = a + b   (skip connections of the U-Net discriminator; dst may alias neither)
__global__ void add2_kernel(float *dst, int d_ct, int d_co, const float *a, int a_ct, int a_co, const float *b, int b_ct, int b_co,
                            int64_t pixels, int C) {
    const int c4n = C / 4;
    const int64_t total = pixels * c4n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % c4n);
        const int64_t pix = e / c4n;
        const f32x4 av = *reinterpret_cast<const f32x4 *>(a + pix * a_ct + a_co + c4 * 4);
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(b + pix * b_ct + b_co + c4 * 4);
        *reinterpret_cast<f32x4 *>(dst + pix * d_ct + d_co + c4 * 4) = av + bv;
    }
}

// dst = src * (y > 0 ? 1 : slope)   (out-of-place LeakyReLU backward: src stays intact for a skip path)
__global__ void mask_copy_kernel(float *dst, int d_ct, int d_co, const float *src, int s_ct, int s_co, const float *y, int y_ct,
                                 int y_co, int64_t pixels, int C, float mslope) {
    const int c4n = C / 4;
    const int64_t total = pixels * c4n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % c4n);
        const int64_t pix = e / c4n;
        f32x4 gv = *reinterpret_cast<const f32x4 *>(src + pix * s_ct + s_co + c4 * 4);
        const f32x4 yv = *reinterpret_cast<const f32x4 *>(y + pix * y_ct + y_co + c4 * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) gv[k] *= (yv[k] > 0.f ? 1.f : mslope);
        *reinterpret_cast<f32x4 *>(dst + pix * d_ct + d_co + c4 * 4) = gv;
    }
}

__global__ void fill_kernel(float *p, int64_t n, float v) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) p[e] = v;
}

inline bool view_ok(const tnr_view &v) { return v.ptr != nullptr && (v.ctot % 4) == 0 && (v.coff % 4) == 0; }

}  // namespace

extern "C" int tnr_nchw_to_nhwc(const float *src, int32_t N, int32_t C, int32_t H, int32_t W, tnr_view dst, int32_t Cpad,
                                const float *scale, const float *shift, void *stream) {
    TNR_REQUIRE(src && view_ok(dst) && (Cpad % 4) == 0 && Cpad >= C, "nchw_to_nhwc: bad arguments");
    TNR_REQUIRE((scale == nullptr) == (shift == nullptr), "nchw_to_nhwc: scale and shift go together");
    const int64_t total = (int64_t)N * H * W * (Cpad / 4);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream, src, N, C, H, W,
                       dst.ptr, dst.ctot, dst.coff, Cpad, scale, shift);
    return tnr_check_launch("nchw_to_nhwc");
}

extern "C" int tnr_nhwc_to_nchw(tnr_view src, int32_t N, int32_t C, int32_t H, int32_t W, float *dst, const float *scale,
                                int32_t accumulate, void *stream) {
    TNR_REQUIRE(dst && view_ok(src), "nhwc_to_nchw: bad arguments");
    const int64_t total = (int64_t)N * C * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream, src.ptr, src.ctot,
                       src.coff, N, C, H, W, dst, scale, accumulate);
    return tnr_check_launch("nhwc_to_nchw");
}

extern "C" int tnr_upsample2x_bwd(tnr_view gup, tnr_view gx, int32_t N, int32_t H, int32_t W, int32_t C, tnr_view mask,
                                  float mslope, void *stream) {
    TNR_REQUIRE(view_ok(gup) && view_ok(gx) && (C % 4) == 0, "upsample2x_bwd: bad arguments");
    const int64_t total = (int64_t)N * H * W * (C / 4);
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream, gup.ptr, gup.ctot,
                       gup.coff, gx.ptr, gx.ctot, gx.coff, N, H, W, C, mask.ptr, mask.ctot, mask.coff, mslope);
    return tnr_check_launch("upsample2x_bwd");
}

extern "C" int tnr_depth_to_space(tnr_view x, tnr_view y, int32_t N, int32_t H, int32_t W, int32_t Cout, void *stream) {
    TNR_REQUIRE(view_ok(x) && view_ok(y), "depth_to_space: bad arguments");
    const int64_t total = (int64_t)N * H * W * Cout;
    hipLaunchKernelGGL(depth_to_space_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream, x.ptr, x.ctot,
                       x.coff, y.ptr, y.ctot, y.coff, N, H, W, Cout);
    return tnr_check_launch("depth_to_space");
}

extern "C" int tnr_space_to_depth_bwd(tnr_view gy, tnr_view gx, int32_t N, int32_t H, int32_t W, int32_t Cout, tnr_view mask,
                                      float mslope, void *stream) {
    TNR_REQUIRE(view_ok(gy) && view_ok(gx), "space_to_depth_bwd: bad arguments");
    const int64_t total = (int64_t)N * H * W * Cout;
    hipLaunchKernelGGL(space_to_depth_bwd_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream, gy.ptr, gy.ctot,
                       gy.coff, gx.ptr, gx.ctot, gx.coff, N, H, W, Cout, mask.ptr, mask.ctot, mask.coff, mslope);
    return tnr_check_launch("space_to_depth_bwd");
}

extern "C" int tnr_maxpool2_fwd(tnr_view x, tnr_view y, int32_t N, int32_t H, int32_t W, int32_t C, void *stream) {
    TNR_REQUIRE(view_ok(x) && view_ok(y) && (C % 4) == 0 && (H % 2) == 0 && (W % 2) == 0, "maxpool2_fwd: bad arguments");
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream, x.ptr, x.ctot, x.coff,
                       y.ptr, y.ctot, y.coff, N, H, W, C);
    return tnr_check_launch("maxpool2_fwd");
}

extern "C" int tnr_maxpool2_bwd(tnr_view gy, tnr_view x, tnr_view gx, int32_t N, int32_t H, int32_t W, int32_t C,
                                void *stream) {
    TNR_REQUIRE(view_ok(gy) && view_ok(x) && view_ok(gx) && (C % 4) == 0, "maxpool2_bwd: bad arguments");
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream, gy.ptr, gy.ctot,
                       gy.coff, x.ptr, x.ctot, x.coff, gx.ptr, gx.ctot, gx.coff, N, H, W, C);
    return tnr_check_launch("maxpool2_bwd");
}

extern "C" int tnr_im2col(tnr_view x, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, int32_t pad,
                          int32_t Ho, int32_t Wo, float *out, void *stream) {
    TNR_REQUIRE(view_ok(x) && out != nullptr && (C % 4) == 0 && k >= 1 && stride >= 1 && pad >= 0, "im2col: bad arguments");
    TNR_REQUIRE(Ho == (H + 2 * pad - k) / stride + 1 && Wo == (W + 2 * pad - k) / stride + 1, "im2col: output size mismatch");
    const int64_t total = (int64_t)N * Ho * Wo * k * k * (C / 4);
    hipLaunchKernelGGL(im2col_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream, x.ptr, x.ctot, x.coff, N, H, W,
                       C, k, stride, pad, Ho, Wo, out);
    return tnr_check_launch("im2col");
}

extern "C" int tnr_axpby(tnr_view dst, tnr_view src, int64_t pixels, int32_t C, float a, float b, void *stream) {
    TNR_REQUIRE(view_ok(dst) && view_ok(src) && (C % 4) == 0, "axpby: bad arguments");
    hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid(pixels * (C / 4))), dim3(EW_BLOCK), 0, (hipStream_t)stream, dst.ptr, dst.ctot,
                       dst.coff, src.ptr, src.ctot, src.coff, pixels, C, a, b);
    return tnr_check_launch("axpby");
}

extern "C" int tnr_gauss_mult(tnr_view dst, tnr_view src, int64_t pixels, int32_t C, float sigma, uint32_t key0, uint32_t key1,
                              uint32_t pix0, void *stream) {
    TNR_REQUIRE(view_ok(dst) && (src.ptr == nullptr || view_ok(src)) && (C % 4) == 0 && pixels >= 0, "gauss_mult: bad arguments");
    if (pixels == 0) return TNR_OK;
    hipLaunchKernelGGL(gauss_mult_kernel, dim3(ew_grid(pixels * (C / 4))), dim3(EW_BLOCK), 0, (hipStream_t)stream, dst.ptr, dst.ctot,
                       dst.coff, src.ptr, src.ctot, src.coff, pixels, C, sigma, key0, key1, pix0);
    return tnr_check_launch("gauss_mult");
}

extern "C" int tnr_mask_mul(tnr_view g, tnr_view y, int64_t pixels, int32_t C, float mslope, void *stream) {
    TNR_REQUIRE(view_ok(g) && view_ok(y) && (C % 4) == 0, "mask_mul: bad arguments");
    hipLaunchKernelGGL(mask_mul_kernel, dim3(ew_grid(pixels * (C / 4))), dim3(EW_BLOCK), 0, (hipStream_t)stream, g.ptr, g.ctot,
                       g.coff, y.ptr, y.ctot, y.coff, pixels, C, mslope);
    return tnr_check_launch("mask_mul");
}

extern "C" int tnr_bilinear2x_fwd(tnr_view x, tnr_view y, int32_t N, int32_t H, int32_t W, int32_t C, void *stream) {
    TNR_REQUIRE(view_ok(x) && view_ok(y) && (C % 4) == 0 && H >= 1 && W >= 1, "bilinear2x_fwd: bad arguments");
    const int64_t total = (int64_t)N * 4 * H * W * (C / 4);
    hipLaunchKernelGGL(bilinear2x_fwd_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream, x.ptr, x.ctot, x.coff,
                       y.ptr, y.ctot, y.coff, N, H, W, C);
    return tnr_check_launch("bilinear2x_fwd");
}

extern "C" int tnr_bilinear2x_bwd(tnr_view gy, tnr_view gx, tnr_view gz, tnr_view mask, float mslope, int32_t N, int32_t H,
                                  int32_t W, int32_t C, void *stream) {
    TNR_REQUIRE(view_ok(gy) && (gx.ptr == nullptr || view_ok(gx)) && (gz.ptr == nullptr || (view_ok(gz) && view_ok(mask))) &&
                    (gx.ptr != nullptr || gz.ptr != nullptr) && (C % 4) == 0,
                "bilinear2x_bwd: bad arguments");
    const int64_t total = (int64_t)N * H * W * (C / 4);
    hipLaunchKernelGGL(bilinear2x_bwd_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream, gy.ptr, gy.ctot, gy.coff,
                       gx.ptr, gx.ctot, gx.coff, gz.ptr, gz.ctot, gz.coff, mask.ptr, mask.ctot, mask.coff, mslope, N, H, W, C);
    return tnr_check_launch("bilinear2x_bwd");
}

extern "C" int tnr_add2(tnr_view dst, tnr_view a, tnr_view b, int64_t pixels, int32_t C, void *stream) {
    TNR_REQUIRE(view_ok(dst) && view_ok(a) && view_ok(b) && (C % 4) == 0, "add2: bad arguments");
    hipLaunchKernelGGL(add2_kernel, dim3(ew_grid(pixels * (C / 4))), dim3(EW_BLOCK), 0, (hipStream_t)stream, dst.ptr, dst.ctot,
                       dst.coff, a.ptr, a.ctot, a.coff, b.ptr, b.ctot, b.coff, pixels, C);
    return tnr_check_launch("add2");
}

extern "C" int tnr_mask_copy(tnr_view dst, tnr_view src, tnr_view y, int64_t pixels, int32_t C, float mslope, void *stream) {
    TNR_REQUIRE(view_ok(dst) && view_ok(src) && view_ok(y) && (C % 4) == 0, "mask_copy: bad arguments");
    hipLaunchKernelGGL(mask_copy_kernel, dim3(ew_grid(pixels * (C / 4))), dim3(EW_BLOCK), 0, (hipStream_t)stream, dst.ptr, dst.ctot,
                       dst.coff, src.ptr, src.ctot, src.coff, y.ptr, y.ctot, y.coff, pixels, C, mslope);
    return tnr_check_launch("mask_copy");
}

extern "C" int tnr_fill(float *p, int64_t n, float v, void *stream) {
    TNR_REQUIRE(p != nullptr && n >= 0, "fill: bad arguments");
    if (n == 0) return TNR_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, (hipStream_t)stream, p, n, v);
    return tnr_check_launch("fill");
}
