// Validation metrics on the device: the reference converts the network output to uint8 images on the host
// (tensor2np, dataops/common.py:502-566) and computes PSNR / SSIM with numpy + cv2 (utils/metrics.py:110-126,
// 180-223, called from train.py:335-372 on border-cropped images).  Here the quantisation and both metrics run on
// the GPU over the whole batch; only one (sum, count) pair per image and metric crosses PCIe.
//   tensor2np : NCHW float -> NHWC uint8, (x + 1) / 2 if denormalize, round_half_even(clip(255 x, 0, 255)), RGB -> BGR
//   PSNR      : sum of squared uint8 differences over the cropped image (exact integer arithmetic)
//   SSIM      : 11 x 11 Gaussian window (sigma 1.5, cv2.getGaussianKernel's formula), 'valid' region, fp64 moments,
//               C1 = (0.01 * 255)^2, C2 = (0.03 * 255)^2, mean over pixels and channels
// Reductions are two-stage with a fixed summation order (deterministic).
#include "common.h"

namespace {

constexpr int MT_BLOCKS = 256;   // partial sums per image

__global__ void tensor2np_u8_kernel(const float *src, int N, int C, int H, int W, uint8_t *dst, int rgb2bgr, int denorm) {
    const int64_t total = (int64_t)N * H * W * C;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const int64_t pix = e / C;                       // n * H * W + y * W + x
        const int64_t hw = (int64_t)H * W;
        const int64_t n = pix / hw, r = pix - n * hw;
        const int cs = (rgb2bgr && C >= 3 && c < 3) ? 2 - c : c;      // RGB(A) -> BGR(A)
        float v = src[(n * C + cs) * hw + r];
        if (denorm) v = (v + 1.0f) / 2.0f;               // denorm(): [-1, 1] -> [0, 1]
        v = 255.0f * v;
        v = fminf(fmaxf(v, 0.0f), 255.0f);
        dst[e] = (uint8_t)rintf(v);                      // np.round: half to even
    }
}

__device__ __forceinline__ double block_sum(double v, double *sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// blockIdx.y = image; partial[(n * MT_BLOCKS + blockIdx.x) * 2 + {0: squared error, 1: ssim sum}]
__global__ void __launch_bounds__(256) psnr_ssim_partial_kernel(const uint8_t *a, const uint8_t *b, int H, int W, int C, int crop,
                                                                 int want_ssim, double *partial) {
    __shared__ double sh[256];
    __shared__ double win[121];
    if (threadIdx.x < 121) {
        // cv2.getGaussianKernel(11, 1.5): exp(-(i - 5)^2 / (2 sigma^2)) normalised to sum 1; window = outer product
        double sum = 0;
        for (int i = 0; i < 11; ++i) sum += exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5));
        const int i = threadIdx.x / 11, j = threadIdx.x % 11;
        win[threadIdx.x] = (exp(-(double)((i - 5) * (i - 5)) / 4.5) / sum) * (exp(-(double)((j - 5) * (j - 5)) / 4.5) / sum);
    }
    __syncthreads();
    const int n = blockIdx.y;
    const uint8_t *pa = a + (size_t)n * H * W * C, *pb = b + (size_t)n * H * W * C;
    const int h = H - 2 * crop, w = W - 2 * crop;
    double se = 0.0;
    const int64_t tot = (int64_t)h * w * C;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const int64_t p = e / C;
        const int x = (int)(p % w), y = (int)(p / w);
        const size_t o = ((size_t)(y + crop) * W + (x + crop)) * C + c;
        const int d = (int)pa[o] - (int)pb[o];
        se += (double)(d * d);
    }
    se = block_sum(se, sh);
    double ss = 0.0;
    if (want_ssim && h >= 11 && w >= 11) {
        const double C1 = (0.01 * 255) * (0.01 * 255), C2 = (0.03 * 255) * (0.03 * 255);
        const int vh = h - 10, vw = w - 10;
        const int64_t vt = (int64_t)vh * vw * C;
        for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < vt; e += (int64_t)gridDim.x * blockDim.x) {
            const int c = (int)(e % C);
            const int64_t p = e / C;
            const int x = (int)(p % vw), y = (int)(p / vw);
            double m1 = 0, m2 = 0, s11 = 0, s22 = 0, s12 = 0;
            for (int dy = 0; dy < 11; ++dy)
                for (int dx = 0; dx < 11; ++dx) {
                    const size_t o = ((size_t)(y + dy + crop) * W + (x + dx + crop)) * C + c;
                    const double wv = win[dy * 11 + dx], va = (double)pa[o], vb = (double)pb[o];
                    m1 += wv * va;
                    m2 += wv * vb;
                    s11 += wv * va * va;
                    s22 += wv * vb * vb;
                    s12 += wv * va * vb;
                }
            const double m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
            ss += ((2 * m12 + C1) * (2 * (s12 - m12) + C2)) / ((m11 + m22 + C1) * ((s11 - m11) + (s22 - m22) + C2));
        }
        ss = block_sum(ss, sh);
    }
    if (threadIdx.x == 0) {
        partial[((size_t)n * MT_BLOCKS + blockIdx.x) * 2 + 0] = se;
        partial[((size_t)n * MT_BLOCKS + blockIdx.x) * 2 + 1] = ss;
    }
}

// out[n] = {sum squared error, count, ssim sum, ssim count}
__global__ void psnr_ssim_finalize_kernel(const double *partial, int nblocks, int H, int W, int C, int crop, int want_ssim, double *out) {
    const int n = blockIdx.x;
    if (threadIdx.x != 0) return;
    double se = 0, ss = 0;
    for (int b = 0; b < nblocks; ++b) {
        se += partial[((size_t)n * MT_BLOCKS + b) * 2 + 0];
        ss += partial[((size_t)n * MT_BLOCKS + b) * 2 + 1];
    }
    const int h = H - 2 * crop, w = W - 2 * crop;
    out[n * 4 + 0] = se;
    out[n * 4 + 1] = (double)h * w * C;
    out[n * 4 + 2] = ss;
    out[n * 4 + 3] = (want_ssim && h >= 11 && w >= 11) ? (double)(h - 10) * (w - 10) * C : 0.0;
}

}  // namespace

extern "C" int tnr_tensor2np_u8(const float *src, int32_t N, int32_t C, int32_t H, int32_t W, uint8_t *dst, int32_t rgb2bgr,
                                int32_t denormalize, void *stream) {
    TNR_REQUIRE(src && dst && N >= 1 && C >= 1 && C <= 4 && H >= 1 && W >= 1, "tensor2np_u8: bad arguments");
    const int64_t total = (int64_t)N * C * H * W;
    int64_t blocks = tnr_cdiv64(total, 256);
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(tensor2np_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, N, C, H, W, dst, rgb2bgr,
                       denormalize);
    return tnr_check_launch("tensor2np_u8");
}

extern "C" int64_t tnr_metrics_workspace_bytes(int32_t N) { return (int64_t)N * MT_BLOCKS * 2 * (int64_t)sizeof(double); }

extern "C" int tnr_psnr_ssim_u8(const uint8_t *a, const uint8_t *b, int32_t N, int32_t H, int32_t W, int32_t C, int32_t crop,
                                int32_t want_ssim, double *out, double *ws, int64_t ws_bytes, void *stream) {
    TNR_REQUIRE(a && b && out && ws && N >= 1 && C >= 1 && crop >= 0 && H > 2 * crop && W > 2 * crop, "psnr_ssim_u8: bad arguments");
    TNR_REQUIRE(ws_bytes >= tnr_metrics_workspace_bytes(N), "psnr_ssim_u8: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int64_t work = (int64_t)(H - 2 * crop) * (W - 2 * crop) * C;
    int nb = (int)tnr_cdiv64(work, 256);
    if (nb > MT_BLOCKS) nb = MT_BLOCKS;
    hipLaunchKernelGGL(psnr_ssim_partial_kernel, dim3(nb, N), dim3(256), 0, s, a, b, H, W, C, crop, want_ssim, ws);
    int rc = tnr_check_launch("psnr_ssim_partial");
    if (rc != TNR_OK) return rc;
    hipLaunchKernelGGL(psnr_ssim_finalize_kernel, dim3(N), dim3(64), 0, s, ws, nb, H, W, C, crop, want_ssim, out);
    return tnr_check_launch("psnr_ssim_finalize");
}
