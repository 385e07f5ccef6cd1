// Tiled implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32,
// exact f32 == an fmaf chain), NHWC views, fused bias / activation / residual / mask epilogue.
//
// One workgroup = 256 threads = 4 waves (one per SIMD), two workgroups per CU (LDS <= 80 KiB).
//   M  = 128*MT output pixels (TH x TW spatial tile; TW in {32,16,8}; MT = 2, or 4 for 32-cout 3x3 layers);
//        wave w owns MT 32-pixel MFMA tiles
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
//   1x1                  no halo, one tap: the GEMM over tnr_im2col's patch matrix (small-spatial layers)
//   3x3 over <= 4 ch     the 9 taps gathered into K = 36 -> 48 by the stager (RGB image layers)
// This file: the one-launch-per-layer kernel (optionally split along K, with a reduce launch) and the host
// side of tnr_conv_forward.  The device body is conv_body.h; conv_chain.hip runs it for several layers per launch.
#include "conv_body.h"
#include "conv_x3w8.h"
#include <cstdlib>
#ifdef TNR_CONV_DL_EXPERIMENT     /* tools/build_variant.py dl -DTNR_CONV_DL_EXPERIMENT: the LDS-DMA staging experiments, conv_body_dl.h */
#include "conv_body_dl.h"
#endif

namespace {

template <int MODE, int TW, int NT, int MT, int BF>
__global__ void __launch_bounds__(256, 2) conv_tile_kernel(const ConvK a) {
    TNR_STAMP_CALL(0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int bid = blockIdx.x;
    const int split = bid % a.ksplit;   // innermost: the splits of a tile run side by side and share its input in L2
    bid /= a.ksplit;
    const int cb = bid % a.ncb;
    bid /= a.ncb;
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    bid /= a.tiles_y;
    int par = 0;
    if (MODE == TNR_DGRAD_4x4_S2) {
        par = bid & 3;
        bid >>= 2;
    }
    ConvK b = a;
    b.y = a.y + (size_t)split * a.split_stride;
    conv_tile_body<MODE, TW, NT, MT, false, BF>(b, cb, tx, ty, bid, par, smem, -1, NoWait(), a.ksplit, split);
}

// Second launch of a split-K convolution: y = act(sum_s ws[s] + bias) * alpha, splits summed in index order.
struct SplitRedK {
    const float *ws; size_t split_stride; int ksplit;
    float *y; int y_ct, y_co, Cout, CoutP;
    long long pixels;
    const float *bias; int act; float slope, alpha;
};
__global__ void __launch_bounds__(256) conv_splitk_reduce_kernel(const SplitRedK a) {
    const int c4n = a.CoutP >> 2;
    const long long total = a.pixels * c4n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long p = i / c4n;
        const int co = (int)(i - p * c4n) * 4;
        if (co >= a.Cout) continue;
        const float *src = a.ws + (size_t)p * a.CoutP + co;
        f32x4 v = *reinterpret_cast<const f32x4 *>(src);
        for (int s = 1; s < a.ksplit; ++s) v += *reinterpret_cast<const f32x4 *>(src + (size_t)s * a.split_stride);
        float *yp = a.y + (size_t)p * a.y_ct + a.y_co + co;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (co + k < a.Cout) {
                const float b = a.bias != nullptr ? a.bias[co + k] : 0.f;
                yp[k] = tnr_act(v[k] + b, a.act, a.slope) * a.alpha;
            }
        }
    }
}

template <int MODE, int TW, int NT, int MT, int BF>
int launch_conv_t(const ConvK &k, int tiles, hipStream_t s) {
    constexpr int TH = 128 * MT / TW;
    constexpr int KH = (MODE == TNR_CONV_4x4_S2) ? 2 : 3;
    constexpr int NTAPS = (MODE == TNR_CONV_4x4_S2 || MODE == TNR_DGRAD_4x4_S2) ? 4 : 9;
    constexpr int ROWA = (BF == 2 && TNR_X3_REFILL != 0) ? TNR_X3_ROW : TNR_PST;      // (conv_body.h: input rows of the split-operand form)
    constexpr int ROWW = (BF == 2 && TNR_X3_REFILL != 0 && TNR_X3_WSPLIT != 0 && NTAPS == 4) ? TNR_X3_ROW : TNR_PST;       // (conv_body.h: X3W)
    constexpr size_t lds_main = (size_t)((TH + KH - 1) * (TW + KH - 1) * ROWA + NTAPS * NT * 32 * ROWW) * sizeof(float);
    constexpr size_t lds_epi = (size_t)4 * MT * 32 * NT * 32 * sizeof(float);   // output transpose tiles of the 4 waves
#ifdef TNR_DEBUG_LDS_PAD   /* experiment knob: force one workgroup per CU */
    constexpr size_t lds = (lds_main > lds_epi ? lds_main : lds_epi) + TNR_DEBUG_LDS_PAD;
#else
    constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
    static_assert(lds <= 80 * 1024, "conv tile exceeds the 2-workgroups-per-CU LDS budget");
#endif
    static bool attr_done = false;
    auto fn = conv_tile_kernel<MODE, TW, NT, MT, BF>;
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

template <int MODE, int TW, int NT, int MT = 2>
int launch_conv(const ConvK &k, int tiles, hipStream_t s) {
    if (k.bf == 2) return launch_conv_t<MODE, TW, NT, MT, 2>(k, tiles, s);
    return k.bf ? launch_conv_t<MODE, TW, NT, MT, 1>(k, tiles, s) : launch_conv_t<MODE, TW, NT, MT, 0>(k, tiles, s);
}

template <int MODE>
int dispatch_conv(const ConvK &k, int tw, int nt, int tiles, hipStream_t s) {
    if (tw == 32) return nt == 2 ? launch_conv<MODE, 32, 2>(k, tiles, s) : launch_conv<MODE, 32, 1>(k, tiles, s);
    if (tw == 16) return nt == 2 ? launch_conv<MODE, 16, 2>(k, tiles, s) : launch_conv<MODE, 16, 1>(k, tiles, s);
    return nt == 2 ? launch_conv<MODE, 8, 2>(k, tiles, s) : launch_conv<MODE, 8, 1>(k, tiles, s);
}

// Tile width of a launch (MT = 2: 256-pixel tiles of 8 x 32, 16 x 16 or 32 x 8): the shape that covers the tile space with the
// fewest padded pixels, the widest on ties (power-of-two sizes keep 8 x 32; a 66 x 66 reflection-padded grid takes 16 x 16:
// 25 instead of 27 tiles per image).
int conv_pick_tw(int sw, int sh) {
    int best = 0;
    int64_t best_area = 0;
    for (int tw = 32; tw >= 8; tw >>= 1) {
        if (tw > 8 && sw < tw) continue;                     // (narrow images: as before)
        const int th = 256 / tw;
        const int64_t area = (int64_t)tnr_cdiv(sw, tw) * tw * tnr_cdiv(sh, th) * th;
        if (best == 0 || area < best_area) { best = tw; best_area = area; }
    }
    return best;
}

// Split-K factor of a launch.  Only plain epilogues (bias / activation / alpha) can be deferred to the reduce
// launch, and only launches that leave most of the 512 workgroup slots empty while looping over >= 32 input
// chunks are worth a second launch: the 512-channel discriminator layers at 16x16 and below.
int conv_ksplit(const tnr_conv_desc *d, int64_t tiles) {
    if (d->r1.ptr || d->r2.ptr || d->m.ptr || d->noise_pos || (d->Cout % 4) != 0 || d->mode == TNR_DGRAD_4x4_S2) return 1;
    const int nchunks = (d->KinP / TNR_CK) * (d->mode == TNR_CONV_4x4_S2 ? 4 : 1);
    if (tiles >= 192 || nchunks < 32) return 1;
    int want = (int)((512 + tiles - 1) / tiles);
    const int max_split = nchunks / 8;          // at least 8 chunks per split
    if (want > max_split) want = max_split;
    if (want < 2) return 1;
    const int per = (nchunks + want - 1) / want;
    return (nchunks + per - 1) / per;           // every split gets a non-empty chunk range
}

}  // namespace

int tnr_launch_conv3x3_d4(const tnr_conv_desc *d, void *stream);      // conv_sweep.hip (1: not for that kernel)
int tnr_launch_conv3x3_wino(const tnr_conv_desc *d, void *stream);    // conv_wino.hip (1: not for that kernel)
int tnr_launch_conv_s2_d4(const tnr_conv_desc *d, void *stream);      // conv_sweep.hip: the four-tap forms (1: not for that kernel)

extern "C" int tnr_conv_forward(const tnr_conv_desc *d, void *stream) {
    TNR_REQUIRE(d != nullptr && d->x.ptr && d->y.ptr && d->wp, "conv: null pointer");
    TNR_REQUIRE(d->mode >= TNR_CONV_3x3 && d->mode <= TNR_CONV_7x7_C4, "conv: bad mode %d", d->mode);
    if (d->mode == TNR_CONV_3x3_C4)
        TNR_REQUIRE(d->x.ctot == 4 && d->x.coff == 0 && d->Cin == 4 && d->KinP == 48, "conv3x3_c4: needs an NHWC4 input view and a C4 packing");
    if (d->mode == TNR_CONV_7x7_C4)
        TNR_REQUIRE(d->x.ctot == 4 && d->x.coff == 0 && d->Cin == 4 && d->KinP == 208 && d->wq == nullptr && d->ws == nullptr && d->shuffle == 0,
                    "conv7x7_c4: needs an NHWC4 input view and a 7x7 C4 packing (no weight stream, no split-K)");
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
    TNR_REQUIRE(d->Cout % 4 == 0 || (d->r1.ptr == nullptr && d->r2.ptr == nullptr && d->m.ptr == nullptr && d->noise_pos == 0),
                "conv: residual / mask / noise epilogues need Cout %% 4 == 0");
    TNR_REQUIRE(d->noise_pos >= 0 && d->noise_pos <= 2, "conv: bad noise_pos %d", d->noise_pos);
    ConvK k;
    k.x = d->x.ptr; k.x_ct = d->x.ctot; k.x_co = d->x.coff;
    k.N = d->N; k.H = d->H; k.W = d->W; k.Cin = d->Cin;
    k.wp = d->wp; k.KinP = d->KinP; k.KoutP = d->KoutP;
    k.y = d->y.ptr; k.y_ct = d->y.ctot; k.y_co = d->y.coff; k.Ho = d->Ho; k.Wo = d->Wo; k.Cout = d->Cout;
    k.bias = d->bias; k.act = d->act; k.slope = d->slope; k.alpha = d->alpha;
    k.r1 = d->r1.ptr; k.r1_ct = d->r1.ctot; k.r1_co = d->r1.coff; k.r1_ch = d->r1_ch; k.beta1 = d->beta1;
    k.r2 = d->r2.ptr; k.r2_ct = d->r2.ctot; k.r2_co = d->r2.coff; k.alpha2 = d->alpha2;
    k.m = d->m.ptr; k.m_ct = d->m.ctot; k.m_co = d->m.coff; k.m_lo = d->m_lo; k.m_hi = d->m_hi; k.m_slope = d->m_slope;
    k.noise_pos = d->noise_pos; k.noise_sigma = d->noise_sigma; k.noise_k0 = d->noise_key0; k.noise_k1 = d->noise_key1; k.noise_pix0 = d->noise_pix0;

    int sh, sw;  // tile space
    switch (d->mode) {
        case TNR_CONV_3x3:
        case TNR_CONV_1x1:
        case TNR_CONV_3x3_C4:
        case TNR_CONV_7x7_C4:
            TNR_REQUIRE(d->Ho == d->H && d->Wo == d->W, "conv3x3 / conv1x1 / conv7x7_c4: output must match input size");
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
    const int tw = conv_pick_tw(sw, sh);
    const int nt = d->Cout > 32 ? 2 : 1;
    // 32-cout 3x3 layers (the dense-block convs and their gradients): 4 M-tiles per wave (16x32 pixel
    // tile) so the weight slab and the fixed per-workgroup costs are amortised like in the 64-cout kernel
    const bool big_m = (d->mode == TNR_CONV_3x3) && nt == 1 && tw == 32 && sh >= 16;
    const int th = (big_m ? 512 : 256) / tw;
    k.tiles_x = tnr_cdiv(sw, tw);
    k.tiles_y = tnr_cdiv(sh, th);
    k.ncb = tnr_cdiv(d->Cout, nt * 32);
    int64_t tiles = (int64_t)k.tiles_x * k.tiles_y * k.ncb * d->N * (d->mode == TNR_DGRAD_4x4_S2 ? 4 : 1);
    TNR_REQUIRE(tiles > 0 && tiles < (1LL << 31), "conv: grid too large");
    hipStream_t s = (hipStream_t)stream;
    // split-K for launches that cannot fill the chip (see tnr_conv_workspace_bytes)
    const int ksplit = conv_ksplit(d, tiles);
    TNR_REQUIRE(d->mma >= TNR_MMA_F32 && d->mma <= TNR_MMA_BF16X3, "conv: bad mma %d", d->mma);
    k.bf = d->mma;
    k.reflect = d->pad_mode == 1;
    TNR_REQUIRE(d->pad_mode == 0 || (d->pad_mode == 1 && d->mode == TNR_CONV_3x3 && d->H >= 2 && d->W >= 2) ||
                (d->pad_mode == 1 && d->mode == TNR_CONV_7x7_C4 && d->H >= 4 && d->W >= 4), "conv: pad_mode 1 (reflection) is for TNR_CONV_3x3 and TNR_CONV_7x7_C4");
    k.ksplit = 1;
    k.split_stride = 0;
    SplitRedK red;
    if (ksplit > 1 && d->ws != nullptr) {
        const int64_t plane = (int64_t)d->N * d->Ho * d->Wo * k.KoutP;
        TNR_REQUIRE(plane * ksplit * (int64_t)sizeof(float) <= d->ws_bytes, "conv: split-K workspace too small");
        red.ws = d->ws; red.split_stride = (size_t)plane; red.ksplit = ksplit;
        red.y = k.y; red.y_ct = k.y_ct; red.y_co = k.y_co; red.Cout = k.Cout; red.CoutP = k.KoutP;
        red.pixels = (long long)d->N * d->Ho * d->Wo;
        red.bias = k.bias; red.act = k.act; red.slope = k.slope; red.alpha = k.alpha;
        k.y = d->ws; k.y_ct = k.KoutP; k.y_co = 0; k.Cout = k.KoutP;   // raw partial sums, all padded channels
        k.bias = nullptr; k.act = TNR_ACT_NONE; k.alpha = 1.f;
        k.ksplit = ksplit; k.split_stride = (size_t)plane;
        tiles *= ksplit;
    }
    int rc;
    if (d->wq != nullptr && d->wq_form == 1) {     // transform-domain weight stream: the Winograd F(2x2, 3x3) kernel (conv_wino.hip) or nothing
        TNR_REQUIRE(k.ksplit == 1, "conv: a Winograd weight stream cannot be combined with a split-K workspace");
        rc = tnr_launch_conv3x3_wino(d, (void *)s);
        TNR_REQUIRE(rc != 1, "conv: wq_form = 1 but the launch does not qualify for the Winograd kernel (tnr_conv_wino_bytes() == 0)");
        return rc;
    }
    if (d->wq != nullptr && k.ksplit == 1) {       // pre-split weight stream: the direct four-wave kernel (conv_sweep.hip), when the launch qualifies
        static const bool d4 = [] { const char *e = std::getenv("TNR_X3_D4"); return e == nullptr || e[0] != '0'; }();
        if (d4 || d->shuffle != 0) {
            rc = (d->mode == TNR_CONV_4x4_S2 || d->mode == TNR_DGRAD_4x4_S2) ? tnr_launch_conv_s2_d4(d, (void *)s) : tnr_launch_conv3x3_d4(d, (void *)s);
            if (rc <= 0) return rc;
        }
    }
    TNR_REQUIRE(d->shuffle == 0, "conv: a pixel-shuffle store (shuffle = %d) exists in the weight-stream kernel only (tnr_conv_wq_bytes() == 0 for this launch)", d->shuffle);
    {   // TNR_MMA_BF16X3, 64-cout 3x3 layers: the 8-wave kernel with both operands pre-split in LDS (conv_x3w8.h; TNR_X3_W8=0: off)
        static const bool w8 = [] { const char *e = std::getenv("TNR_X3_W8"); return e == nullptr || e[0] != '0'; }();
        if (w8 && d->mode == TNR_CONV_3x3 && nt == 2 && tw == 32 && sh >= 16 && conv3x3_x3w8_ok(k)) return launch_conv3x3_x3w8(k, s);
    }
#ifdef TNR_CONV_DL_EXPERIMENT
    const char *dl_env = std::getenv("TNR_CONV_DL");     // (experiment switch, read per call so that one process can compare both paths)
    const int dl = dl_env != nullptr ? dl_env[0] - '0' : 0;      // 1: 8-wave / 16-channel form, 2: 4-wave / 8-channel form
    if (dl == 1 && d->mode == TNR_CONV_3x3 && nt == 2 && tw == 32 && sh >= 16 && conv3x3_dl_ok(k)) return launch_conv3x3_dl<0>(k, s);
    if (dl == 4 && d->mode == TNR_CONV_3x3 && nt == 2 && tw == 32 && sh >= 16 && conv3x3_dl_ok(k)) return launch_conv3x3_dl<2>(k, s);     // 4: form 1 with two loader waves
    if (dl == 5 && d->mode == TNR_CONV_3x3 && nt == 2 && tw == 32 && sh >= 16 && conv3x3_dl_ok(k)) return launch_conv3x3_dl<4>(k, s);     // 5: four loader waves
    if (dl == 2 && d->mode == TNR_CONV_3x3 && tw == 32 && conv3x3_dl_ok(k)) {
        if (big_m) return launch_conv3x3_dk8<1, 4>(k, tiles, s);
        if (nt == 2) return launch_conv3x3_dk8<2, 2>(k, tiles, s);
    }
#endif
    if (big_m) {
        rc = launch_conv<TNR_CONV_3x3, 32, 1, 4>(k, (int)tiles, s);
    } else {
        switch (d->mode) {
            case TNR_CONV_3x3: rc = dispatch_conv<TNR_CONV_3x3>(k, tw, nt, (int)tiles, s); break;
            case TNR_CONV_3x3_UP2: rc = dispatch_conv<TNR_CONV_3x3_UP2>(k, tw, nt, (int)tiles, s); break;
            case TNR_CONV_4x4_S2: rc = dispatch_conv<TNR_CONV_4x4_S2>(k, tw, nt, (int)tiles, s); break;
            case TNR_CONV_1x1: rc = dispatch_conv<TNR_CONV_1x1>(k, tw, nt, (int)tiles, s); break;
            case TNR_CONV_3x3_C4: rc = dispatch_conv<TNR_CONV_3x3_C4>(k, tw, nt, (int)tiles, s); break;
            case TNR_CONV_7x7_C4: rc = dispatch_conv<TNR_CONV_7x7_C4>(k, tw, nt, (int)tiles, s); break;
            default: rc = dispatch_conv<TNR_DGRAD_4x4_S2>(k, tw, nt, (int)tiles, s); break;
        }
    }
    if (rc != TNR_OK || k.ksplit == 1) return rc;
    const long long work = red.pixels * (red.CoutP >> 2);
    long long blocks = (work + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, red);
    return tnr_check_launch("conv_splitk_reduce");
}

extern "C" int64_t tnr_conv_workspace_bytes(const tnr_conv_desc *d) {
    if (d == nullptr) return 0;
    int sh = d->Ho, sw = d->Wo;
    if (d->mode == TNR_DGRAD_4x4_S2) { sh = d->H; sw = d->W; }
    const int tw = conv_pick_tw(sw, sh);
    const int nt = d->Cout > 32 ? 2 : 1;
    const bool big_m = (d->mode == TNR_CONV_3x3) && nt == 1 && tw == 32 && sh >= 16;
    const int th = (big_m ? 512 : 256) / tw;
    const int64_t tiles = (int64_t)tnr_cdiv(sw, tw) * tnr_cdiv(sh, th) * tnr_cdiv(d->Cout, nt * 32) * d->N *
                          (d->mode == TNR_DGRAD_4x4_S2 ? 4 : 1);
    const int ks = conv_ksplit(d, tiles);
    return ks > 1 ? (int64_t)ks * d->N * d->Ho * d->Wo * tnr_round_up(d->Cout, 32) * (int64_t)sizeof(float) : 0;
}
