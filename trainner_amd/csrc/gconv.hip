// Generic convolution family on the vector ALUs -- any square kernel, stride, zero or reflection padding -- for the few
// layers of the image-to-image networks (ResnetGenerator ResNet_arch.py:11-90, NLayerDiscriminator
// discriminators.py:472-579) that fall outside the matrix-core kernels' geometries: the 7x7 reflection-padded first /
// last convolutions (3 or 64 channels on one side) and the PatchGAN's two 4x4 stride-1 layers.  Together < 5 % of those
// networks' FLOP; the 3x3 / 4x4-s2 / transposed layers run on the MFMA tiles (conv_tile.hip), reflection padding for the
// residual blocks is materialised by tnr_pad2d below.
//   forward        y[n,oy,ox,co] = act(b[co] + sum_{ky,kx,ci} x[n, R(oy s - p + ky), R(ox s - p + kx), ci] w[co,ci,ky,kx])
//   data-gradient  gx[n,y,x,ci]  = sum over every (oy, ky) with R(oy s - p + ky) == y (same in x) of g[n,oy,ox,co] w[co,ci,ky,kx]
//   weight-grad    dw[co,ci,ky,kx] = beta dw + sum_{n,oy,ox} g[n,oy,ox,co] x[n, R(..), R(..), ci]   (two-stage, fixed order)
// R = identity with zeros outside (zero padding) or the reflection of nn.ReflectionPad2d (no edge repeat).
#include "common.h"

namespace {

__device__ __forceinline__ int refl(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

struct GcK {
    const float *x; int x_ct, x_co;
    const float *w;            // OIHW
    const float *bias;
    float *y; int y_ct, y_co;
    int N, H, W, Cin, Ho, Wo, Cout, k, stride, pad, reflect, act;
    float slope;
};

// one thread = one output pixel x 4 consecutive output channels
__global__ void gconv_fwd_kernel(const GcK a) {
    const int cg = (a.Cout + 3) / 4;
    const int64_t total = (int64_t)a.N * a.Ho * a.Wo * cg;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % cg);
        int64_t q = e / cg;
        const int ox = (int)(q % a.Wo);
        q /= a.Wo;
        const int oy = (int)(q % a.Ho);
        const int n = (int)(q / a.Ho);
        const int co = c4 * 4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const size_t wco = (size_t)a.Cin * a.k * a.k;
        for (int ky = 0; ky < a.k; ++ky) {
            int Y = oy * a.stride - a.pad + ky;
            if (a.reflect) Y = refl(Y, a.H);
            else if (Y < 0 || Y >= a.H) continue;
            for (int kx = 0; kx < a.k; ++kx) {
                int X = ox * a.stride - a.pad + kx;
                if (a.reflect) X = refl(X, a.W);
                else if (X < 0 || X >= a.W) continue;
                const float *xp = a.x + (((size_t)n * a.H + Y) * a.W + X) * a.x_ct + a.x_co;
                const float *wp = a.w + (size_t)ky * a.k + kx;
                for (int ci = 0; ci < a.Cin; ++ci) {
                    const float xv = xp[ci];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (co + j < a.Cout) acc[j] = fmaf(xv, wp[((size_t)(co + j) * a.Cin + ci) * a.k * a.k], acc[j]);
                }
            }
        }
        (void)wco;
        float *yp = a.y + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.y_ct + a.y_co + co;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (co + j < a.Cout) {
                float v = acc[j] + (a.bias ? a.bias[co + j] : 0.f);
                yp[j] = tnr_act(v, a.act, a.slope);
            }
    }
}

// one thread = one input pixel x 4 consecutive input channels; x / y of GcK are the gradient of the output (g) and gx
__global__ void gconv_dgrad_kernel(const GcK a) {   // a.x = g [N,Ho,Wo,Cout], a.y = gx [N,H,W,Cin]
    const int cg = (a.Cin + 3) / 4;
    const int64_t total = (int64_t)a.N * a.H * a.W * cg;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % cg);
        int64_t q = e / cg;
        const int x = (int)(q % a.W);
        q /= a.W;
        const int y = (int)(q % a.H);
        const int n = (int)(q / a.H);
        const int ci = c4 * 4;
        // padded-grid coordinates that read input row y (reflection: the row itself and its mirror images)
        int ys[3], xs[3], nys = 0, nxs = 0;
        ys[nys++] = y;
        xs[nxs++] = x;
        if (a.reflect) {
            if (y >= 1 && y <= a.pad) ys[nys++] = -y;
            if (y <= a.H - 2 && y >= a.H - 1 - a.pad) ys[nys++] = 2 * a.H - 2 - y;
            if (x >= 1 && x <= a.pad) xs[nxs++] = -x;
            if (x <= a.W - 2 && x >= a.W - 1 - a.pad) xs[nxs++] = 2 * a.W - 2 - x;
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int iy = 0; iy < nys; ++iy)
            for (int ky = 0; ky < a.k; ++ky) {
                const int ty = ys[iy] + a.pad - ky;
                if (ty < 0 || ty % a.stride) continue;
                const int oy = ty / a.stride;
                if (oy >= a.Ho) continue;
                for (int ix = 0; ix < nxs; ++ix)
                    for (int kx = 0; kx < a.k; ++kx) {
                        const int tx = xs[ix] + a.pad - kx;
                        if (tx < 0 || tx % a.stride) continue;
                        const int ox = tx / a.stride;
                        if (ox >= a.Wo) continue;
                        const float *gp = a.x + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.x_ct + a.x_co;
                        const float *wp = a.w + (size_t)ky * a.k + kx;
                        for (int co = 0; co < a.Cout; ++co) {
                            const float gv = gp[co];
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (ci + j < a.Cin) acc[j] = fmaf(gv, wp[((size_t)co * a.Cin + ci + j) * a.k * a.k], acc[j]);
                        }
                    }
            }
        float *op = a.y + (((size_t)n * a.H + y) * a.W + x) * a.y_ct + a.y_co + ci;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (ci + j < a.Cin) op[j] = acc[j];
    }
}

constexpr int GW_SPLITS = 64;

// partial[split][co][ci][ky][kx] over the pixels of split `blockIdx.y` (contiguous pixel ranges: fixed summation order)
__global__ void gconv_wgrad_partial_kernel(const GcK a, double *partial, double *bpartial) {   // a.y = g
    const int kk = a.k * a.k;
    const int64_t outs = (int64_t)a.Cout * a.Cin * kk;
    const int64_t pixels = (int64_t)a.N * a.Ho * a.Wo;
    const int64_t per = (pixels + GW_SPLITS - 1) / GW_SPLITS;
    const int64_t p0 = (int64_t)blockIdx.y * per, p1 = p0 + per < pixels ? p0 + per : pixels;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < outs + a.Cout; e += (int64_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        if (e >= outs) {               // bias gradient
            const int co = (int)(e - outs);
            for (int64_t p = p0; p < p1; ++p) s += (double)a.y[p * a.y_ct + a.y_co + co];
            bpartial[(size_t)blockIdx.y * a.Cout + co] = s;
            continue;
        }
        const int t = (int)(e % kk);
        int64_t q = e / kk;
        const int ci = (int)(q % a.Cin);
        const int co = (int)(q / a.Cin);
        const int ky = t / a.k, kx = t % a.k;
        for (int64_t p = p0; p < p1; ++p) {
            const int ox = (int)(p % a.Wo);
            const int64_t r = p / a.Wo;
            const int oy = (int)(r % a.Ho);
            const int n = (int)(r / a.Ho);
            int Y = oy * a.stride - a.pad + ky, X = ox * a.stride - a.pad + kx;
            if (a.reflect) {
                Y = refl(Y, a.H);
                X = refl(X, a.W);
            } else if (Y < 0 || Y >= a.H || X < 0 || X >= a.W) {
                continue;
            }
            s += (double)a.y[p * a.y_ct + a.y_co + co] * (double)a.x[(((size_t)n * a.H + Y) * a.W + X) * a.x_ct + a.x_co + ci];
        }
        partial[(size_t)blockIdx.y * outs + e] = s;
    }
}

__global__ void gconv_wgrad_final_kernel(const double *partial, const double *bpartial, int64_t outs, int Cout, float *dw, float *db,
                                         float alpha, float beta) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < outs + Cout; e += (int64_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        if (e >= outs) {
            if (db == nullptr) continue;
            for (int sp = 0; sp < GW_SPLITS; ++sp) s += bpartial[(size_t)sp * Cout + (e - outs)];
            float *o = db + (e - outs);
            *o = (beta != 0.f ? beta * *o : 0.f) + alpha * (float)s;
        } else {
            for (int sp = 0; sp < GW_SPLITS; ++sp) s += partial[(size_t)sp * outs + e];
            dw[e] = (beta != 0.f ? beta * dw[e] : 0.f) + alpha * (float)s;
        }
    }
}

// ---- padding helpers for the reflection-padded 3x3 convolutions of the residual blocks (run on the MFMA 3x3 kernel over
// the materialised padded tensor) -----------------------------------------------------------------------------------
// mode 0: zero border (embed), 1: reflection.  y [N, H + 2p, W + 2p, C]
__global__ void pad2d_kernel(const float *x, int x_ct, int x_co, float *y, int y_ct, int y_co, int N, int H, int W, int C, int p, int mode) {
    const int Hp = H + 2 * p, Wp = W + 2 * p, c4n = C / 4;
    const int64_t total = (int64_t)N * Hp * Wp * c4n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % c4n);
        const int64_t pix = e / c4n;
        const int X = (int)(pix % Wp) - p;
        const int64_t q = pix / Wp;
        const int Y = (int)(q % Hp) - p;
        const int64_t n = q / Hp;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (mode == 1 || (Y >= 0 && Y < H && X >= 0 && X < W))
            v = *reinterpret_cast<const f32x4 *>(x + ((n * H + refl(Y, H)) * W + refl(X, W)) * x_ct + x_co + c4 * 4);
        *reinterpret_cast<f32x4 *>(y + pix * y_ct + y_co + c4 * 4) = v;
    }
}

// mode 0: crop the centre of xp [N, H + 2p, W + 2p, C] into y [N,H,W,C]; mode 1: adjoint of the reflection padding (the
// border rows / columns are folded back onto their source pixels); optional y = a * result + b * y0 not needed here
__global__ void unpad2d_kernel(const float *xp, int x_ct, int x_co, float *y, int y_ct, int y_co, int N, int H, int W, int C, int p, int mode) {
    const int Wp = W + 2 * p, Hp = H + 2 * p, c4n = C / 4;
    const int64_t total = (int64_t)N * H * W * c4n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % c4n);
        const int64_t pix = e / c4n;
        const int x = (int)(pix % W);
        const int64_t q = pix / W;
        const int yy = (int)(q % H);
        const int64_t n = q / H;
        int ys[3], xs[3], nys = 0, nxs = 0;
        ys[nys++] = yy;
        xs[nxs++] = x;
        if (mode == 1) {
            if (yy >= 1 && yy <= p) ys[nys++] = -yy;
            if (yy <= H - 2 && yy >= H - 1 - p) ys[nys++] = 2 * H - 2 - yy;
            if (x >= 1 && x <= p) xs[nxs++] = -x;
            if (x <= W - 2 && x >= W - 1 - p) xs[nxs++] = 2 * W - 2 - x;
        }
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < nys; ++a)
            for (int b = 0; b < nxs; ++b)
                s += *reinterpret_cast<const f32x4 *>(xp + ((n * Hp + ys[a] + p) * Wp + xs[b] + p) * x_ct + x_co + c4 * 4);
        *reinterpret_cast<f32x4 *>(y + pix * y_ct + y_co + c4 * 4) = s;
    }
}

// tanh forward (y = tanh(x)) and backward (gx = g * (1 - y^2)) on flat fp32 buffers
__global__ void tanh_fwd_kernel(const float *x, float *y, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) y[e] = tanhf(x[e]);
}
__global__ void tanh_bwd_kernel(const float *g, const float *y, float *gx, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        gx[e] = g[e] * (1.f - y[e] * y[e]);
}

// GANLoss against a constant label (modules/loss.py:61-137): type 0 vanilla = BCEWithLogits(mean), 1 lsgan = MSE(mean),
// 2 = mean(pred) itself (log entries).
// out[0] = loss; grad[i] = d loss / d pred[i] (unit upstream gradient).  One block: the logit maps are small (30 x 30 per image).
__global__ void __launch_bounds__(256) gan_loss_kernel(const float *pred, int64_t n, int type, float target, float *out, float *grad) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float x = pred[i];
        float l, g;
        if (type == 0) {       // BCE with logits: max(x, 0) - x t + log(1 + exp(-|x|))
            l = fmaxf(x, 0.f) - x * target + log1pf(expf(-fabsf(x)));
            g = 1.f / (1.f + expf(-x)) - target;
        } else if (type == 1) {
            const float d = x - target;
            l = d * d;
            g = 2.f * d;
        } else {               // plain mean of the logits (the D_real / D_fake log entries, losses.py:519-520)
            l = x;
            g = 1.f;
        }
        s += (double)l;
        if (grad) grad[i] = g / (float)n;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(sh[0] / (double)n);
}

// dst[n, y, x, c] (+)= src[n, y + oy, x + ox, c] where that lies inside the source, else 0 (float4 per thread; acc: added to dst): zero-embedding of a
// gradient at an offset (oy, ox < 0) and offset crops, for the convolutions that run as shifted windows of another geometry.
__global__ void window2d_kernel(const float *src, int s_ct, int s_co, int Hs, int Ws, float *dst, int d_ct, int d_co, int N, int Hd,
                                int Wd, int C, int oy, int ox, int acc) {
    const int c4n = C / 4;
    const int64_t total = (int64_t)N * Hd * Wd * c4n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % c4n);
        int64_t q = e / c4n;
        const int x = (int)(q % Wd);
        q /= Wd;
        const int y = (int)(q % Hd);
        const int n = (int)(q / Hd);
        const int sy = y + oy, sx = x + ox;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (sy >= 0 && sy < Hs && sx >= 0 && sx < Ws)
            v = *reinterpret_cast<const f32x4 *>(src + (((size_t)n * Hs + sy) * Ws + sx) * s_ct + s_co + c4 * 4);
        float *dp = dst + (((size_t)n * Hd + y) * Wd + x) * d_ct + d_co + c4 * 4;
        if (acc) v += *reinterpret_cast<const f32x4 *>(dp);
        *reinterpret_cast<f32x4 *>(dp) = v;
    }
}

// ---- bias gradient: column sums of an NHWC view.  partial[b][c] = sum over the block's contiguous pixel range (thread =
// (pixel lane, 4 channels), double accumulation, lanes combined through LDS in lane order); colsum_final adds the blocks in
// index order: fixed summation order, bit-reproducible.
constexpr int CS_BLOCKS = 512;

__global__ void __launch_bounds__(256) colsum_partial_kernel(const float *g, int ct, int co, int64_t pixels, int C, int64_t ppb, double *partial) {
    extern __shared__ double cs_sh[];      // [lanes][C]
    const int G = C / 4, lanes = 256 / G;
    const int tid = threadIdx.x, cg = tid % G, pl = tid / G;
    const int64_t p0 = (int64_t)blockIdx.x * ppb;
    const int64_t p1 = p0 + ppb < pixels ? p0 + ppb : pixels;
    if (pl < lanes) {
        double s[4] = {0, 0, 0, 0};
        for (int64_t p = p0 + pl; p < p1; p += lanes) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(g + p * ct + co + cg * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] += (double)v[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) cs_sh[(size_t)pl * C + cg * 4 + k] = s[k];
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        double a = 0;
        for (int l = 0; l < lanes; ++l) a += cs_sh[(size_t)l * C + c];
        partial[(size_t)blockIdx.x * C + c] = a;
    }
}

__global__ void colsum_final_kernel(const double *partial, int nblocks, int C, float *db, float alpha, float beta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int b = 0;
    for (; b + 3 < nblocks; b += 4) {
        s0 += partial[(size_t)b * C + c];
        s1 += partial[(size_t)(b + 1) * C + c];
        s2 += partial[(size_t)(b + 2) * C + c];
        s3 += partial[(size_t)(b + 3) * C + c];
    }
    for (; b < nblocks; ++b) s0 += partial[(size_t)b * C + c];
    db[c] = (beta != 0.f ? beta * db[c] : 0.f) + alpha * (float)((s0 + s1) + (s2 + s3));
}

inline unsigned g1d(int64_t n) {
    int64_t b = tnr_cdiv64(n, 256);
    return (unsigned)(b > 65535 ? 65535 : (b < 1 ? 1 : b));
}
inline bool v_ok(const tnr_view &v) { return v.ptr != nullptr; }

int fill_gck(GcK &a, tnr_view x, int32_t N, int32_t H, int32_t W, int32_t Cin, const float *w, const float *bias, tnr_view y, int32_t Ho,
             int32_t Wo, int32_t Cout, int32_t k, int32_t stride, int32_t pad, int32_t reflect, int32_t act, float slope) {
    TNR_REQUIRE(v_ok(x) && v_ok(y) && w && N >= 1 && k >= 1 && stride >= 1 && pad >= 0, "gconv: bad arguments");
    TNR_REQUIRE(Ho == (H + 2 * pad - k) / stride + 1 && Wo == (W + 2 * pad - k) / stride + 1, "gconv: output size mismatch");
    TNR_REQUIRE(!reflect || (stride == 1 && pad < H && pad < W), "gconv: reflection padding needs stride 1 and pad < size");
    a.x = x.ptr; a.x_ct = x.ctot; a.x_co = x.coff; a.w = w; a.bias = bias; a.y = y.ptr; a.y_ct = y.ctot; a.y_co = y.coff;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.k = k; a.stride = stride; a.pad = pad;
    a.reflect = reflect; a.act = act; a.slope = slope;
    return TNR_OK;
}

}  // namespace

extern "C" int tnr_gconv_fwd(tnr_view x, int32_t N, int32_t H, int32_t W, int32_t Cin, const float *w, const float *bias, tnr_view y,
                             int32_t Ho, int32_t Wo, int32_t Cout, int32_t k, int32_t stride, int32_t pad, int32_t reflect, int32_t act,
                             float slope, void *stream) {
    GcK a;
    const int rc = fill_gck(a, x, N, H, W, Cin, w, bias, y, Ho, Wo, Cout, k, stride, pad, reflect, act, slope);
    if (rc != TNR_OK) return rc;
    hipLaunchKernelGGL(gconv_fwd_kernel, dim3(g1d((int64_t)N * Ho * Wo * ((Cout + 3) / 4))), dim3(256), 0, (hipStream_t)stream, a);
    return tnr_check_launch("gconv_fwd");
}

extern "C" int tnr_gconv_dgrad(tnr_view g, int32_t N, int32_t H, int32_t W, int32_t Cin, const float *w, tnr_view gx, int32_t Ho,
                               int32_t Wo, int32_t Cout, int32_t k, int32_t stride, int32_t pad, int32_t reflect, void *stream) {
    GcK a;
    const int rc = fill_gck(a, g, N, H, W, Cin, w, nullptr, gx, Ho, Wo, Cout, k, stride, pad, reflect, 0, 0.f);
    if (rc != TNR_OK) return rc;
    hipLaunchKernelGGL(gconv_dgrad_kernel, dim3(g1d((int64_t)N * H * W * ((Cin + 3) / 4))), dim3(256), 0, (hipStream_t)stream, a);
    return tnr_check_launch("gconv_dgrad");
}

extern "C" int64_t tnr_gconv_wgrad_workspace_bytes(int32_t Cout, int32_t Cin, int32_t k) {
    return (int64_t)GW_SPLITS * ((int64_t)Cout * Cin * k * k + Cout) * (int64_t)sizeof(double);
}

extern "C" int tnr_gconv_wgrad(tnr_view x, int32_t N, int32_t H, int32_t W, int32_t Cin, tnr_view g, int32_t Ho, int32_t Wo, int32_t Cout,
                               int32_t k, int32_t stride, int32_t pad, int32_t reflect, float *dw, float *db, float alpha, float beta,
                               double *ws, int64_t ws_bytes, void *stream) {
    GcK a;
    const int rc = fill_gck(a, x, N, H, W, Cin, dw, nullptr, g, Ho, Wo, Cout, k, stride, pad, reflect, 0, 0.f);
    if (rc != TNR_OK) return rc;
    TNR_REQUIRE(ws && ws_bytes >= tnr_gconv_wgrad_workspace_bytes(Cout, Cin, k), "gconv_wgrad: workspace too small");
    const int64_t outs = (int64_t)Cout * Cin * k * k;
    double *bpart = ws + (size_t)GW_SPLITS * outs;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gconv_wgrad_partial_kernel, dim3(g1d(outs + Cout), GW_SPLITS), dim3(256), 0, s, a, ws, bpart);
    hipLaunchKernelGGL(gconv_wgrad_final_kernel, dim3(g1d(outs + Cout)), dim3(256), 0, s, ws, bpart, outs, Cout, dw, db, alpha, beta);
    return tnr_check_launch("gconv_wgrad");
}

// db = beta db + alpha * sum over pixels of g[p][c]: the bias gradient of layers whose weight gradient runs elsewhere
// (transposed convolutions on the MFMA kernels); same two-stage fixed-order reduction (ws: >= 64 * C doubles)
extern "C" int tnr_bias_grad(tnr_view g, int64_t pixels, int32_t C, float *db, float alpha, float beta, double *ws, int64_t ws_bytes,
                             void *stream) {
    TNR_REQUIRE(v_ok(g) && db && ws && pixels >= 1 && C >= 1 && ws_bytes >= (int64_t)GW_SPLITS * C * (int64_t)sizeof(double),
                "bias_grad: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if ((C % 4) == 0 && C / 4 <= 256 && (g.ctot % 4) == 0 && (g.coff % 4) == 0 && ws_bytes >= (int64_t)CS_BLOCKS * C * (int64_t)sizeof(double)) {
        const int lanes = 256 / (C / 4);
        int64_t want = tnr_cdiv64(pixels, (int64_t)lanes * 16);
        if (want > CS_BLOCKS) want = CS_BLOCKS;
        const int64_t ppb = tnr_cdiv64(pixels, want);
        const int nblocks = (int)tnr_cdiv64(pixels, ppb);
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblocks), dim3(256), (size_t)lanes * C * sizeof(double), s, g.ptr, g.ctot, g.coff,
                           pixels, C, ppb, ws);
        hipLaunchKernelGGL(colsum_final_kernel, dim3(tnr_cdiv(C, 256)), dim3(256), 0, s, ws, nblocks, C, db, alpha, beta);
        return tnr_check_launch("bias_grad");
    }
    GcK a = {};
    a.y = g.ptr; a.y_ct = g.ctot; a.y_co = g.coff;
    a.N = 1; a.Ho = 1; a.Wo = (int)pixels; a.H = 1; a.W = (int)pixels; a.Cin = 0; a.Cout = C; a.k = 1; a.stride = 1;
    TNR_REQUIRE(pixels < (1LL << 31), "bias_grad: too many pixels");
    hipLaunchKernelGGL(gconv_wgrad_partial_kernel, dim3(g1d(C), GW_SPLITS), dim3(256), 0, s, a, ws, ws);
    hipLaunchKernelGGL(gconv_wgrad_final_kernel, dim3(g1d(C)), dim3(256), 0, s, ws, ws, (int64_t)0, C, db, db, alpha, beta);
    return tnr_check_launch("bias_grad");
}

extern "C" int tnr_pad2d(tnr_view x, tnr_view y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t pad, int32_t mode, void *stream) {
    TNR_REQUIRE(v_ok(x) && v_ok(y) && (C % 4) == 0 && (x.ctot % 4) == 0 && (x.coff % 4) == 0 && (y.ctot % 4) == 0 && (y.coff % 4) == 0 &&
                    pad >= 0 && (mode == 0 || (pad < H && pad < W)),
                "pad2d: bad arguments");
    hipLaunchKernelGGL(pad2d_kernel, dim3(g1d((int64_t)N * (H + 2 * pad) * (W + 2 * pad) * (C / 4))), dim3(256), 0, (hipStream_t)stream, x.ptr,
                       x.ctot, x.coff, y.ptr, y.ctot, y.coff, N, H, W, C, pad, mode);
    return tnr_check_launch("pad2d");
}

extern "C" int tnr_unpad2d(tnr_view xp, tnr_view y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t pad, int32_t mode, void *stream) {
    TNR_REQUIRE(v_ok(xp) && v_ok(y) && (C % 4) == 0 && (xp.ctot % 4) == 0 && (xp.coff % 4) == 0 && (y.ctot % 4) == 0 && (y.coff % 4) == 0 &&
                    pad >= 0 && (mode == 0 || (pad < H && pad < W)),
                "unpad2d: bad arguments");
    hipLaunchKernelGGL(unpad2d_kernel, dim3(g1d((int64_t)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, xp.ptr, xp.ctot, xp.coff,
                       y.ptr, y.ctot, y.coff, N, H, W, C, pad, mode);
    return tnr_check_launch("unpad2d");
}

extern "C" int tnr_window2d(tnr_view src, int32_t Hs, int32_t Ws, tnr_view dst, int32_t N, int32_t Hd, int32_t Wd, int32_t C, int32_t oy,
                            int32_t ox, int32_t acc, void *stream) {
    TNR_REQUIRE(v_ok(src) && v_ok(dst) && N > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && (C % 4) == 0 && (src.ctot % 4) == 0 &&
                    (src.coff % 4) == 0 && (dst.ctot % 4) == 0 && (dst.coff % 4) == 0,
                "window2d: bad arguments");
    hipLaunchKernelGGL(window2d_kernel, dim3(g1d((int64_t)N * Hd * Wd * (C / 4))), dim3(256), 0, (hipStream_t)stream, src.ptr, src.ctot,
                       src.coff, Hs, Ws, dst.ptr, dst.ctot, dst.coff, N, Hd, Wd, C, oy, ox, acc);
    return tnr_check_launch("window2d");
}

extern "C" int tnr_tanh_fwd(const float *x, float *y, int64_t n, void *stream) {
    TNR_REQUIRE(x && y && n >= 0, "tanh_fwd: bad arguments");
    if (n) hipLaunchKernelGGL(tanh_fwd_kernel, dim3(g1d(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return tnr_check_launch("tanh_fwd");
}

extern "C" int tnr_tanh_bwd(const float *g, const float *y, float *gx, int64_t n, void *stream) {
    TNR_REQUIRE(g && y && gx && n >= 0, "tanh_bwd: bad arguments");
    if (n) hipLaunchKernelGGL(tanh_bwd_kernel, dim3(g1d(n)), dim3(256), 0, (hipStream_t)stream, g, y, gx, n);
    return tnr_check_launch("tanh_bwd");
}

extern "C" int tnr_gan_loss(const float *pred, int64_t n, int32_t type, float target, float *out, float *grad, void *stream) {
    TNR_REQUIRE(pred && out && n >= 1 && (type >= 0 && type <= 2), "gan_loss: bad arguments");
    hipLaunchKernelGGL(gan_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, n, type, target, out, grad);
    return tnr_check_launch("gan_loss");
}
