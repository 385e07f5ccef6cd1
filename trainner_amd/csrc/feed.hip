// Input feed: the wire format between the host loader and the engine is what OpenCV hands the reference's datasets --
// uint8 HWC BGR crop windows -- not fp32 tensors: 4x fewer bytes over PCIe, and the conversions the reference does per
// sample on DataLoader worker CPUs run here in one launch per batch, on the copy stream, overlapped with the step:
//   paired flip / 90-degree rotation   dataops/augmentations.py:790-830 (flip = np.flip(axis 1); rotate90 = optional
//                                      np.flip(axis 0) then np.rot90(k = 1)), parameters from get_params (:457-511)
//   np2tensor                          dataops/common.py:470-499: x * data_range / 255 (in double, then float),
//                                      HWC -> CHW, BGR(A) -> RGB(A), optional norm() = clamp((x - 0.5) * 2, -1, 1)
// flags[n]: bit 0 flip, bit 1 rot, bit 2 vflip (only meaningful with rot, like the reference).
#include "common.h"

namespace {

__global__ void feed_u8_to_tensor_kernel(const uint8_t *src, int N, int H, int W, int C, const int32_t *flags, float *dst, int Ho,
                                         int Wo, int bgr2rgb, double data_range, int normalize) {
    const int64_t total = (int64_t)N * C * Ho * Wo;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(e % Wo);
        int64_t q = e / Wo;
        const int i = (int)(q % Ho);
        q /= Ho;
        const int c = (int)(q % C);
        const int n = (int)(q / C);
        const int f = flags ? flags[n] : 0;
        int y = i, x = j;
        if (f & 2) {                 // np.rot90(m, 1)[i][j] = m[j][Wm - 1 - i], m = (vflip ? flipud : id)(flipped crop), shape H x W
            y = j;
            x = W - 1 - i;
            if (f & 4) y = H - 1 - y;
        }
        if (f & 1) x = W - 1 - x;
        int cs = c;
        if (bgr2rgb && (C == 3 || C == 4) && c < 3) cs = 2 - c;
        const uint8_t v = src[(((size_t)n * H + y) * W + x) * C + cs];
        float r = (float)((double)v * data_range / 255.0);
        if (normalize) r = fminf(fmaxf((r - 0.5f) * 2.0f, -1.0f), 1.0f);
        dst[e] = r;
    }
}

}  // namespace

extern "C" int tnr_feed_u8_to_tensor(const uint8_t *src, int32_t N, int32_t H, int32_t W, int32_t C, const int32_t *flags,
                                     int32_t any_rot, float *dst, int32_t bgr2rgb, float data_range, int32_t normalize, void *stream) {
    TNR_REQUIRE(src && dst && N >= 1 && H >= 1 && W >= 1 && C >= 1 && C <= 4, "feed_u8_to_tensor: bad arguments");
    TNR_REQUIRE(!any_rot || H == W, "feed_u8_to_tensor: rotated samples need square windows (one output shape per batch)");
    const int64_t total = (int64_t)N * C * H * W;
    int64_t blocks = tnr_cdiv64(total, 256);
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(feed_u8_to_tensor_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, N, H, W, C, flags, dst,
                       H, W, bgr2rgb, (double)data_range, normalize);
    return tnr_check_launch("feed_u8_to_tensor");
}
