/*
 * trainner_hip.h -- C ABI of libtrainner_hip.so: the MI355X (gfx950) kernels behind the
 * ESRGAN/RRDBNet SR training step of victorca25/traiNNer.
 *
 * The reference has NO native boundary (it is 100 % Python on stock ATen kernels), so this ABI
 * is the boundary a maintainer would bind with ctypes.  Every entry point
 * names the reference call site whose arithmetic it replaces; paths are relative to
 * /root/reference/codes.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.  All pointers are DEVICE pointers
 *     owned by the caller; nothing is allocated inside (workspaces are passed in explicitly).
 *   - every call is asynchronous on the hipStream_t passed as `stream` (void*; 0 = null stream).
 *   - return value: 0 = ok, negative = TNR_E*; tnr_last_error() gives a message.
 *   - activations are fp32 NHWC "views": element (n,y,x,c) of a view lives at
 *         ptr[ ((n*H + y)*W + x) * ctot + coff + c ]
 *     so a dense-block buffer (ctot = 192) is addressed in place -- torch.cat
 *     (RRDBNet_arch.py:152-160) never materialises.  ctot and coff are multiples of 4.
 *   - weights cross the boundary in PyTorch's OIHW layout (state_dict compatible) and are
 *     re-laid out on device by tnr_pack_weights into the K-chunked [tap][kout][kin] form.
 */
#ifndef TRAINNER_HIP_H
#define TRAINNER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TNR_OK 0
#define TNR_EINVAL (-1)   /* bad argument / unsupported shape */
#define TNR_ELAUNCH (-2)  /* HIP launch / runtime error */

enum { TNR_ACT_NONE = 0, TNR_ACT_LRELU = 1, TNR_ACT_RELU = 2 };

/* conv geometries on the path */
enum {
    TNR_CONV_3x3 = 0,      /* k3 s1 p1: every G / VGG conv, D odd layers (block.py:214-256)              */
    TNR_CONV_3x3_UP2 = 1,  /* nearest x2 (block.py:326-371,390-404) folded into the k3 s1 p1 gather       */
    TNR_CONV_4x4_S2 = 2,   /* k4 s2 p1: Discriminator_VGG even layers (discriminators.py:24-34)           */
    TNR_DGRAD_4x4_S2 = 3,  /* data-gradient of k4 s2 p1 (aten convolution_backward), parity-decomposed    */
    TNR_CONV_1x1 = 4,      /* k1 s1: the GEMM behind tnr_im2col for the small-spatial layers of the       */
                           /* discriminator tail (discriminators.py:24-36 at 16x16 and below)             */
    TNR_CONV_3x3_C4 = 5,   /* k3 s1 p1 over a <= 4-channel NHWC4 image (VGG conv1_1, D conv0, and the     */
                           /* data-gradient of G's last conv): the 9 taps are folded into K = 36 -> 48    */
                           /* inside the stager instead of padding 3 channels to 16 per tap               */
    TNR_CONV_7x7_C4 = 6    /* k7 s1 p3 over a <= 4-channel NHWC4 image (ResnetGenerator's first layer,    */
                           /* ResNet_arch.py:52-55, and the data-gradient of its last, :86-88): the 49    */
                           /* taps folded into K = 196 -> 208; pad_mode 1 = ReflectionPad2d(3)            */
};

/* weight packings produced by tnr_pack_weights */
enum {
    TNR_PACK_FWD = 0,       /* [kh*kw][KoutP][KinP]            <- W[co][ci][ky][kx]                         */
    TNR_PACK_DGRAD_3x3 = 1, /* [9][KoutP=ci][KinP=co]          <- W[co][ci][2-ky][2-kx]                     */
    TNR_PACK_FWD_S2D = 2,   /* [4][KoutP][4*KinP] (space-to-depth view of k4 s2)                           */
    TNR_PACK_DGRAD_S2 = 3,  /* [4 parities][4][KoutP=ci][KinP=co]                                          */
    TNR_PACK_COL_FWD = 4,   /* [1][KoutP][KinP = kh*kw*Cin]: column (ky*kw+kx)*Cin+ci of tnr_im2col          */
    TNR_PACK_COL_DGRAD3 = 5,/* [1][KoutP=ci][KinP = k*k*Cout]: column t*Cout+co <- W[co][ci][k-1-ty][k-1-tx]  */
                            /* (any square k: the stride-1 data-gradient as im2col(g, pad k-1-p) x this)   */
    TNR_PACK_C4_FWD = 6,    /* [1][KoutP][48]: column 4*t + ci (Cin <= 4)        for TNR_CONV_3x3_C4             */
    TNR_PACK_C4_DGRAD3 = 7  /* [1][KoutP=ci][48]: column 4*t + co (Cout <= 4), taps flipped                     */
                            /* (both also with kh = kw = 7: [..][208] for TNR_CONV_7x7_C4)                     */
};

#define TNR_MMA_F32 0
#define TNR_MMA_BF16 1
#define TNR_MMA_BF16X3 2

typedef struct tnr_view {
    float *ptr;
    int32_t ctot; /* channels per pixel of the underlying buffer */
    int32_t coff; /* first channel of this view                  */
} tnr_view;

/* One tiled implicit-GEMM convolution (forward, or data-gradient with dgrad-packed weights).
 * Replaces nn.Conv2d.forward + bias + (Leaky)ReLU(inplace) of block.conv_block (block.py:214-256),
 * the `x5*0.2 + x` / `out*0.2 + x` residuals (RRDBNet_arch.py:96,163), ShortcutBlock's add
 * (block.py:184-195), and in the backward direction aten::convolution_backward(input) with the
 * following leaky_relu_backward fused as a mask.
 *   v = act(acc + bias[c]);  v = v*alpha + (c < r1_ch ? beta1*r1[c] : 0);  if r2: v = v*alpha2 + r2[c];
 *   if m and m_lo <= c < m_hi:  v *= (m[c] > 0 ? 1 : m_slope);   y[c] = v                              */
typedef struct tnr_conv_desc {
    tnr_view x;            /* input activation (or incoming gradient for dgrad)               */
    int32_t N, H, W;       /* batch and SOURCE spatial size of x                               */
    int32_t Cin;           /* valid reduction channels in x                                    */
    const float *wp;       /* packed weights (tnr_pack_weights)                                */
    int32_t KinP, KoutP;   /* padded dims of the packing                                       */
    tnr_view y;            /* output                                                           */
    int32_t Ho, Wo, Cout;  /* output spatial size and channels stored                          */
    int32_t mode;          /* TNR_CONV_* */
    const float *bias;     /* [Cout] or NULL                                                   */
    int32_t act;           /* TNR_ACT_* applied to acc+bias                                    */
    float slope;
    float alpha;
    tnr_view r1; int32_t r1_ch; float beta1;   /* r1.ptr NULL = none */
    tnr_view r2; float alpha2;                 /* r2.ptr NULL = none */
    tnr_view m;  int32_t m_lo, m_hi; float m_slope; /* m.ptr NULL = none */
    /* optional split-K workspace (tnr_conv_workspace_bytes()).  A launch with too few output tiles to
     * fill the chip and a long reduction (the 512-channel layers at 16x16 and below) is split along K
     * into partial sums in ws, reduced in a fixed order by a second launch that applies bias / act / alpha.
     * NULL: never split.                                                                               */
    float *ws; int64_t ws_bytes;
    /* matrix-core operand precision: TNR_MMA_F32 = v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate: the default, bit
     * parity class of the CPU reference); TNR_MMA_BF16 = activations and weights rounded to bf16 (round-to-nearest-even) as
     * they enter the matrix core, v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 epilogue and storage -- the engine's
     * `use_amp: true` policy (base_model.py:736-744 autocasts the convolutions to half precision);
     * TNR_MMA_BF16X3 = fp32 arithmetic on the bf16 matrix core: every operand is split EXACTLY into three bf16 values
     * (hi + mid + lo = x), the six largest of the nine partial products (each exact) are accumulated in fp32 -- the dropped
     * ones are below 2^-23 of the product by construction (2^-28 on average: less than fp32 rounding) -- at 6/16 of the fp32
     * matrix-core cycles.  Accepted by
     * tnr_conv_forward, tnr_conv_chain and tnr_wgrad / tnr_wgrad_group.  Finite inputs only: an infinite operand splits into
     * inf + NaN (the fp32 matrix core would deliver inf or NaN there as well); operands below 2^-110 lose their low split.  */
    int32_t mma;
    /* border handling of the TNR_CONV_3x3 mode: 0 = zero padding (nn.Conv2d padding 1), 1 = nn.ReflectionPad2d(1) in front of
     * an unpadded convolution (the ResnetGenerator's residual blocks, ResNet_arch.py:118-146): the stager reads row -1 as row 1
     * and row H as row H - 2, so the reflection-padded tensor is never materialised.                                      */
    int32_t pad_mode;
    /* ESRGAN+ GaussianNoise (block.py:587-600; ResidualDenseBlock_5C.forward RRDBNet_arch.py:160-163: `noise(x5*0.2 + x)`, on by
     * default, defaults.py:59): the epilogue value is multiplied by m = 1 + noise_sigma * n, n ~ N(0, 1), one draw per output element
     * -- `x + n * (sigma * x)` with the gradient through both terms, i.e. d/dx = m.  noise_pos: 0 none; 1 after the r1 step (forward:
     * v = (act(acc + bias) * alpha + beta1 r1) * m, then the r2 step: the RRDB residual sees the noised tensor); 2 after the r2 step
     * (backward: the gradient written by this launch enters a noised tensor).  n is never stored: it is a counter-based function of
     * (noise_key0, noise_key1, (noise_pix0 + output pixel) * Cout / 4 + channel / 4) -- see tnr_gauss_mult, which evaluates the same
     * function on a plain tensor -- so the backward pass regenerates the forward pass's draw bit for bit from the same key.
     * noise_pix0: first pixel of this rank's shard in the global batch (data-parallel ranks draw what one process would draw on the
     * concatenated batch).  Cout % 4 == 0; never combined with split-K; tnr_conv_sweep accepts it on the last stage only.          */
    float noise_sigma; int32_t noise_pos; uint32_t noise_key0, noise_key1, noise_pix0;
    /* optional: the layer's weights as a PRE-SPLIT stream (tnr_conv_wq_pack from the packed fp32 weights `wp`; valid until those
     * change) for TNR_MMA_BF16X3 / TNR_CONV_3x3 launches with Cout % 64 == 0 and Cin % 16 == 0: the kernel then takes its weight
     * fragments straight from L2 into registers (three bf16 planes in MFMA operand order, one coalesced 1 KB read per plane) instead
     * of splitting and staging the slab through LDS per workgroup and chunk.  NULL: the launch stages `wp` itself.  Same results
     * bit for bit either way.                                                                                                    */
    const void *wq; int64_t wq_bytes;
    /* which stream `wq` is: 0 = the tap-major one above (tnr_conv_wq_pack), 1 = the Winograd F(2x2, 3x3) transform-domain stream
     * (tnr_conv_wino_pack): the launch then runs as 16 transform-domain GEMMs per 2 x 2 output patch -- 2.25 x fewer matrix-core
     * instructions for the same convolution; the transforms cost a few extra fp32 roundings per element (error vs fp64 held to 3 x
     * the fp32 matrix core's in the tests, against 1.5 x for the direct forms) and the results are NOT bit-identical to them.      */
    int32_t wq_form;
    /* nn.PixelShuffle(2) behind the convolution (block.pixelshuffle_block, block.py:374-387: conv nf -> 4 nf, shuffle, act) folded into
     * the STORE: 0 = off; 2 = y is the SHUFFLED tensor [N, 2 Ho, 2 Wo, Cout / 4] and conv output channel 4 c + 2 dy + dx of pixel (oy, ox)
     * is written to channel c of pixel (2 oy + dy, 2 ox + dx) -- the [N, Ho, Wo, Cout] intermediate and the depth-to-space pass never
     * exist.  The epilogue (bias, act, alpha) is applied as usual (an elementwise activation commutes with the shuffle); residuals, mask
     * and noise are not supported.  Needs the pre-split weight stream (wq, wq_form = 0, built with this field set: the stream's cout
     * order follows the sub-pixel) and Cout % 256 == 0; anything else is refused -- the caller then runs tnr_depth_to_space itself.  */
    int32_t shuffle;
} tnr_conv_desc;

/* Weight-gradient of one convolution: dW[co][ci][ky][kx] = beta*dW + alpha * sum_pixels g * x
 * (aten::convolution_backward(weight, bias)), pixel-split into deterministic partial slabs in `ws`
 * and reduced in a fixed order.  The launch covers input channels [cin_begin, cin_begin+Cin) of a
 * layer whose weight tensor has cin_total input channels.                                        */
typedef struct tnr_wgrad_desc {
    tnr_view x; int32_t N, H, W; int32_t Cin;      /* layer input view (already offset to cin_begin) */
    tnr_view g; int32_t Ho, Wo; int32_t Cout;      /* gradient w.r.t. the conv output (pre-activation) */
    int32_t mode;                                  /* TNR_CONV_3x3 | _3x3_UP2 | _4x4_S2               */
    float *dw; int32_t cin_total, cin_begin;       /* OIHW gradient tensor                            */
    float *db;                                     /* [Cout] or NULL (only with cin_begin == 0)        */
    float alpha, beta;
    float *ws; int64_t ws_bytes;                   /* >= tnr_wgrad_workspace_bytes()                  */
    int32_t mma;                                   /* TNR_MMA_F32 | TNR_MMA_BF16 (see tnr_conv_desc)   */
    int32_t pad_mode;                              /* TNR_CONV_3x3: 0 zero padding, 1 reflection (tnr_conv_desc) */
    /* Two layers in one job ("cout pair"; cout_split = 0: off).  When the gradients of two layers that read the SAME input channels
     * stand side by side in one buffer -- a dense block's [g4 | g3] and [g2 | g1] (RRDBNet_arch.py:150-163) -- `g` may cover both
     * (Cout = 64): output channels [0, cout_split) then belong to the layer of dw / cin_total / db, channels [cout_split, Cout) to a
     * second layer dw2 / cin_total2 / db2 (same cin_begin, alpha, beta).  The pair runs in the 64-cout workgroup tile class: one
     * read of the input tile feeds both layers.  cout_split is a multiple of 32.                                               */
    float *dw2; int32_t cout_split, cin_total2;
    float *db2;
} tnr_wgrad_desc;

typedef struct tnr_pack_item {
    const float *w;  /* OIHW */
    float *wp;
    int32_t Cout, Cin, kh, kw;
    int32_t kind;    /* TNR_PACK_* */
    int32_t KoutP, KinP;
    int64_t n_out;   /* number of packed floats */
} tnr_pack_item;

/* "Gradient dense block": the data-gradient of a ResidualDenseBlock_5C (RRDBNet_arch.py:150-163) run as
 * the mirror image of its forward.  With the gradient buffer laid out [g5(nf) | g4 | g3 | g2 | g1]
 * (g_k = gradient w.r.t. conv_k's pre-activation output), the gradient w.r.t. the block's t-th feature
 * group (t = 0..4 -> x4, x3, x2, x1, x) is ONE convolution over the first nf + t*gc channels of that
 * buffer, with this packing of the transposed / flipped slices of W5..W(5-t) (conv5's slice pre-scaled by
 * scale5 = 0.2 * s).  No read-modify-write accumulation, K grows like in the forward pass.             */
typedef struct tnr_dense_pack_item {
    const float *w[5]; /* conv1..conv5 weights, OIHW */
    float *wp;
    int32_t nf, gc;
    int32_t t;         /* 0..4 */
    int32_t KoutP, KinP;
    float scale5;
    int64_t n_out;
} tnr_dense_pack_item;

const char *tnr_last_error(void);
/* ABI version = TNR_ABI_VERSION of the header the library was built from.  It moves whenever a descriptor struct changes size or
 * layout: a caller built against another header would pass a shorter struct and the library would read fields from adjacent memory.
 * Consumers compare it with THEIR header's TNR_ABI_VERSION before the first call (trainner_amd/hip.py does).
 *   1: rounds 1-4.   2: tnr_wgrad_desc gained dw2 / cout_split / cin_total2 / db2, TNR_WGRAD_GROUP_MAX 8 -> 12 (round 5; the number
 *      itself was bumped in round 6).   3: tnr_conv_desc gained wq_form and shuffle (round 6).                                      */
#define TNR_ABI_VERSION 3
int tnr_version(void);

/* --- convolution family ---------------------------------------------------------------------- */
int tnr_pack_dims(int32_t Cout, int32_t Cin, int32_t kh, int32_t kw, int32_t kind,
                  int32_t *KoutP, int32_t *KinP, int64_t *n_out);
/* items: DEVICE array of n descriptors; max_out = max n_out over the items */
int tnr_pack_weights(const tnr_pack_item *items_dev, int32_t n, int64_t max_out, void *stream);
int tnr_pack_dense_dims(int32_t nf, int32_t gc, int32_t t, int32_t *KoutP, int32_t *KinP, int64_t *n_out);
int tnr_pack_dense_dgrad(const tnr_dense_pack_item *items_dev, int32_t n, int64_t max_out, void *stream);
int tnr_conv_forward(const tnr_conv_desc *d, void *stream);
/* dst[p][c] = (src ? src[p][c] : 1) * (1 + sigma * n(key0, key1, (pix0 + p) * C / 4 + c / 4)) for `pixels` pixels of C channels: the
 * multiplier field of tnr_conv_desc.noise_* on a plain NHWC tensor (dst may alias src).  Used where the gradient of a noised tensor
 * is needed next to the plain one (the RRDB skip, RRDBNet_arch.py:96), and by tests to read the engine's draw.  C % 4 == 0.          */
int tnr_gauss_mult(tnr_view dst, tnr_view src, int64_t pixels, int32_t C, float sigma, uint32_t key0, uint32_t key1, uint32_t pix0,
                   void *stream);
int64_t tnr_conv_workspace_bytes(const tnr_conv_desc *d);   /* 0 when the launch would not be split */
/* size of / build the pre-split weight stream of a launch (tnr_conv_desc.wq); 0: the launch cannot use one */
int64_t tnr_conv_wq_bytes(const tnr_conv_desc *d);
int tnr_conv_wq_pack(const tnr_conv_desc *d, void *image, int64_t image_bytes, void *stream);
/* the same for the Winograd F(2x2, 3x3) form (tnr_conv_desc.wq with wq_form = 1): U = G g G^T of every (cout, cin) filter, computed in
 * fp64 from the packed fp32 weights, rounded once to fp32 and split into three bf16 planes in MFMA operand order.  64-cout blocks,
 * Cin % 16 == 0, TNR_MMA_BF16X3, stride 1, zero or reflection padding (nn.Conv2d k3 s1 p1, block.py:214-256; ResNet_arch.py:118-146). */
int64_t tnr_conv_wino_bytes(const tnr_conv_desc *d);
int tnr_conv_wino_pack(const tnr_conv_desc *d, void *image, int64_t image_bytes, void *stream);
/* out[(n*Ho + oy)*Wo + ox][(ky*kw + kx)*C + c] = x[n][oy*stride - pad + ky][ox*stride - pad + kx][c] (0 outside):
 * the patch matrix of a k x k convolution as an NHWC "image" of Ho*Wo*N pixels with kh*kw*C channels, for
 * TNR_CONV_1x1 with TNR_PACK_COL_* weights.  C % 4 == 0.                                                */
int tnr_im2col(tnr_view x, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, int32_t pad,
               int32_t Ho, int32_t Wo, float *out, void *stream);
/* n <= TNR_CHAIN_MAX dependent 3x3 convolutions over one pixel grid in ONE launch: the five convolutions of a
 * ResidualDenseBlock_5C (RRDBNet_arch.py:150-163) or of its gradient mirror.  Stage i may read what stages
 * < i of the same call wrote: fresh_from[i] is the first input channel of stage i that stage i-1 produced
 * (-1: none; channels below it must predate the call or come from stages <= i-2).  A workgroup keeps its
 * 16x32-pixel tile through all stages and waits, at that channel, for the 8 neighbouring tiles (3x3 halo)
 * through per-tile progress counters in ws (tnr_conv_chain_workspace_bytes(), zeroed once; the last word
 * becomes nonzero if a wait ever gave up).  epoch must grow by 1 per call on the same ws.  Every stage:
 * mode TNR_CONV_3x3, same N/H/W, Cout % 32 == 0, KoutP == Cout; epilogues as in tnr_conv_forward.  Outputs of
 * different stages must not overlap, and no stage may write channels another stage of the call reads
 * before its fresh_from point.  Results are bit-identical to n tnr_conv_forward calls.                    */
#define TNR_CHAIN_MAX 6
int64_t tnr_conv_chain_workspace_bytes(const tnr_conv_desc *stage0);
int tnr_conv_chain(const tnr_conv_desc *stages, const int32_t *fresh_from, int32_t n, uint32_t *ws, int64_t ws_bytes,
                   uint32_t epoch, void *stream);
/* The same five convolutions of a residual dense block (RRDBNet_arch.py:150-163: stage k reads channels [0, nf + 32 k) of ONE buffer
 * and, except the last, writes the next 32 channels of it; 32, 32, 32, 32, 64 output channels) -- or of its gradient mirror -- as a
 * SWEEP: TNR_MMA_BF16X3 only.  One workgroup keeps the accumulators of all 192 output channels of its 8 x 32 pixel tile in
 * registers and reads every input channel chunk once per phase for all the stages that consume it, with the weights streamed
 * pre-split (three bf16 planes) from `image`, which tnr_conv_sweep_pack builds from the stages' packed fp32 weights (d->wp) and
 * which stays valid until those change.  tnr_conv_sweep_image_bytes returns 0 when the stages are not such a block (or an
 * image's tiles exceed the co-resident workgroups): use tnr_conv_chain then.  ws / epoch as for tnr_conv_chain (same buffer,
 * same counter; the buffer also holds the launch's tile dispenser, which every launch leaves at zero: one launch at a time per ws,
 * i.e. one ws per stream).  Results are bit-identical to tnr_conv_chain / five tnr_conv_forward calls in TNR_MMA_BF16X3.    */
int64_t tnr_conv_sweep_image_bytes(const tnr_conv_desc *stages, int32_t n);
int tnr_conv_sweep_pack(const tnr_conv_desc *stages, int32_t n, void *image, int64_t image_bytes, void *stream);
/* the images of MANY blocks in one launch: tnr_conv_sweep_pack_item validates a block like tnr_conv_sweep_pack and fills one HOST entry
 * (opaque but for `units`); the caller keeps the entries of all its blocks in a DEVICE array and rebuilds every image with one
 * tnr_conv_sweep_pack_batch(items_dev, n, max over the entries' units) whenever the packed weights changed (once per optimiser step:
 * 2 launches -- forward and gradient-mirror blocks -- instead of 2 x 69 for RRDBNet-23).  An entry stays valid while the stages'
 * packed-weight pointers and the image pointer do.                                                                                  */
typedef struct tnr_sweep_pack_item { int32_t units; int32_t reserved; uint64_t opaque[15]; } tnr_sweep_pack_item;
int tnr_conv_sweep_pack_item(const tnr_conv_desc *stages, int32_t n, void *image, int64_t image_bytes, tnr_sweep_pack_item *item);
int tnr_conv_sweep_pack_batch(const tnr_sweep_pack_item *items_dev, int32_t n, int32_t max_units, void *stream);
int tnr_conv_sweep(const tnr_conv_desc *stages, int32_t n, const void *image, uint32_t *ws, int64_t ws_bytes, uint32_t epoch,
                   void *stream);
/* 3x3 s1 p1 convolution with Cout <= 4 on the vector ALUs (one thread = one pixel x 4 outputs): G's last conv
 * (RRDBNet_arch.py:44) and the data-gradients that end in the RGB image (VGG features[0], D's first conv).
 * y[c] = (sum + bias[c]) * alpha for c < Cout.  Weights: tnr_conv_thin_pack (dgrad = 0: forward of a layer with
 * Cout <= 4; dgrad = 1: data-gradient of a layer with Cin <= 4, then Cin/Cout of tnr_conv_thin are the layer's
 * Cout/Cin) into tnr_conv_thin_pack_floats(reduction channels) floats.                                     */
int64_t tnr_conv_thin_pack_floats(int32_t reduce_channels);
int tnr_conv_thin_pack(const float *w_oihw, float *wp, int32_t Cout, int32_t Cin, int32_t dgrad, void *stream);
int tnr_conv_thin(tnr_view x, int32_t N, int32_t H, int32_t W, int32_t Cin, const float *wp, tnr_view y, int32_t Cout,
                  const float *bias, float alpha, void *stream);
/* The same for the 7x7 image-side layers of ResnetGenerator (ResNet_arch.py:52-55, :86-88), one launch for the 49 taps:
 *   y[n, oy, ox, c] = (sum_{t, ci} w[t][ci][c] * x[n, oy + ty - pad, ox + tx - pad, ci] + bias[c]) * alpha   (c < Cout <= 4)
 * over an Ho x Wo output grid; reflect = 1: ReflectionPad2d(pad) borders (pad <= 3, Ho x Wo = H x W), reflect = 0: zeros outside
 * the H x W input -- with pad = 6 and Ho x Wo = (H + 6) x (W + 6) that is the data-gradient of a 7x7 layer with respect to its
 * reflection-padded input (fold it back with tnr_unpad2d).  Weights: tnr_conv_thin7_pack (dgrad as in tnr_conv_thin_pack) into
 * tnr_conv_thin7_pack_floats(reduction channels) floats.  Cout <= 3 and Cin <= 64 run with one lane per INPUT channel (weights in
 * registers, coalesced input rows, per-pixel sums through LDS); a 4-channel output or more input channels with one lane per pixel.  */
int64_t tnr_conv_thin7_pack_floats(int32_t reduce_channels);
int tnr_conv_thin7_pack(const float *w_oihw, float *wp, int32_t Cout, int32_t Cin, int32_t dgrad, void *stream);
int tnr_conv_thin7(tnr_view x, int32_t N, int32_t H, int32_t W, int32_t Cin, const float *wp, tnr_view y, int32_t Ho, int32_t Wo,
                   int32_t Cout, int32_t pad, int32_t reflect, const float *bias, float alpha, void *stream);
int64_t tnr_wgrad_workspace_bytes(const tnr_wgrad_desc *d);
/* Weight (+ bias) gradient of a 3x3 s1 p1 layer with <= 3 channels on one side, on the vector ALUs:
 * flip = 0: a Csmall -> Cbig layer (big = gradient of its output, small = its NHWC4 input image),
 *           dw = [Cbig][Csmall][3][3], db = [Cbig];
 * flip = 1: a Cbig -> Csmall layer (big = its input, small = NHWC4 gradient of its output),
 *           dw = [Csmall][Cbig][3][3], db = [Csmall].
 * dw = beta*dw + alpha*sum (db likewise, may be NULL).  Cbig in {16, 32, 64}.  Deterministic.               */
int64_t tnr_wgrad_thin_workspace_bytes(int32_t N, int32_t H, int32_t Cbig);
int tnr_wgrad_thin(tnr_view big, tnr_view small, int32_t N, int32_t H, int32_t W, int32_t Cbig, int32_t Csmall, int32_t flip,
                   float *dw, float *db, float alpha, float beta, float *ws, int64_t ws_bytes, void *stream);
/* The same for a 7x7 layer, one launch for the 49 taps (+ one reduce launch):
 *   acc[cb][cs][ty][tx] = sum over the big grid q of big(q)[cb] * small[q + (ty, tx) + off][cs]      (small: zeros outside Hs x Ws)
 * big grid = (H + 2 rpad) x (W + 2 rpad), big(q) = the H x W buffer at the ReflectionPad2d(rpad) image of q (rpad = 0: the buffer).
 * flip = 0: dw[cb][cs][t] = acc[..][t]       -- an image -> Cbig layer: big = gradient of its output, small = its reflection-padded
 *           NHWC4 input image ((H + 6) x (W + 6)), off = 0; db[cb] = sum of big (may be NULL);
 * flip = 1: dw[cs][cb][t] = acc[..][48 - t]  -- a Cbig -> image layer: big = its input read through ReflectionPad2d(3) (rpad = 3),
 *           small = the NHWC4 gradient of its output (H x W), off = -6; db must be NULL (sum the 4-channel gradient elsewhere).
 * dw = beta*dw + alpha*acc.  Cbig in {16, 32, 64}, Csmall <= 3.  Deterministic.                               */
int64_t tnr_wgrad_thin7_workspace_bytes(int32_t N, int32_t Hbig_grid, int32_t Cbig);
int tnr_wgrad_thin7(tnr_view big, int32_t N, int32_t H, int32_t W, int32_t rpad, tnr_view small, int32_t Hs, int32_t Ws, int32_t off,
                    int32_t Cbig, int32_t Csmall, int32_t flip, float *dw, float *db, float alpha, float beta, float *ws, int64_t ws_bytes,
                    void *stream);
int tnr_conv_wgrad(const tnr_wgrad_desc *d, void *stream);
/* n <= TNR_WGRAD_GROUP_MAX layers in ONE launch (+ one reduce launch).  The layers must share mode and
 * N/H/W/Ho/Wo and fall into the same workgroup tile class (same Cout <= 32 | > 32 and the same number of
 * 32-channel input blocks per workgroup: e.g. the 64-input pieces of a dense block's conv1/conv3/conv4);
 * each needs its own workspace of tnr_wgrad_workspace_bytes().  Results equal n single launches up to
 * the summation order of the split-K partials (still run-to-run deterministic).                     */
#define TNR_WGRAD_GROUP_MAX 12
int tnr_conv_wgrad_group(const tnr_wgrad_desc *descs, int32_t n, void *stream);

/* --- image-to-image family (Pix2Pix / CycleGAN: ResnetGenerator ResNet_arch.py:11-90, NLayerDiscriminator
 * discriminators.py:472-579).  Generic convolution on the vector ALUs for the layers outside the matrix-core geometries
 * (7x7 reflection-padded first / last convolutions, the PatchGAN's 4x4 stride-1 layers): any square kernel k, stride,
 * zero (reflect = 0) or nn.ReflectionPad2d (reflect = 1, stride 1) padding; OIHW weights as stored in the state_dict.
 *   fwd:   y = act(conv(x, w) + bias);  dgrad: gx = conv_backward_data(g, w) (reflection folded back);
 *   wgrad: dw = beta dw + alpha sum g (x) x, db likewise (db may be NULL); two-stage fp64 reduction in a fixed order,
 *          ws: tnr_gconv_wgrad_workspace_bytes(Cout, Cin, k).
 * tnr_pad2d / tnr_unpad2d: materialise a zero (mode 0) or reflection (mode 1) border of `pad` pixels / crop it (mode 0) or
 * apply the adjoint of the reflection padding (mode 1) -- the residual blocks' reflection-padded 3x3 convolutions run on
 * the MFMA 3x3 kernel over the padded tensor.  tnr_tanh_*: the generator's output activation.  tnr_gan_loss: GANLoss
 * against a constant label (modules/loss.py:61-137): type 0 vanilla (BCE with logits), 1 lsgan (MSE), mean reduction,
 * 2 = mean(pred) (the D_real / D_fake log entries, losses.py:519-520); out[0] = loss, grad = d loss / d pred (may be NULL). */
int tnr_gconv_fwd(tnr_view x, int32_t N, int32_t H, int32_t W, int32_t Cin, const float *w, const float *bias, tnr_view y, int32_t Ho,
                  int32_t Wo, int32_t Cout, int32_t k, int32_t stride, int32_t pad, int32_t reflect, int32_t act, float slope, void *stream);
int tnr_gconv_dgrad(tnr_view g, int32_t N, int32_t H, int32_t W, int32_t Cin, const float *w, tnr_view gx, int32_t Ho, int32_t Wo,
                    int32_t Cout, int32_t k, int32_t stride, int32_t pad, int32_t reflect, void *stream);
int64_t tnr_gconv_wgrad_workspace_bytes(int32_t Cout, int32_t Cin, int32_t k);
int tnr_gconv_wgrad(tnr_view x, int32_t N, int32_t H, int32_t W, int32_t Cin, tnr_view g, int32_t Ho, int32_t Wo, int32_t Cout, int32_t k,
                    int32_t stride, int32_t pad, int32_t reflect, float *dw, float *db, float alpha, float beta, double *ws,
                    int64_t ws_bytes, void *stream);
/* db = beta db + alpha sum_pixels g[p][c] (bias gradient of a layer whose weight gradient runs on another kernel);
 * ws: >= 512 * C doubles (two-stage column sum in a fixed order; >= 64 * C selects the slow generic path).          */
int tnr_bias_grad(tnr_view g, int64_t pixels, int32_t C, float *db, float alpha, float beta, double *ws, int64_t ws_bytes, void *stream);
int tnr_pad2d(tnr_view x, tnr_view y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t pad, int32_t mode, void *stream);
int tnr_unpad2d(tnr_view xp, tnr_view y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t pad, int32_t mode, void *stream);
/* dst[n,y,x,:] (+)= src[n, y+oy, x+ox, :] inside the source, 0 outside (zero-embedding at an offset / offset crop; acc != 0
 * adds to dst): kernel geometries without their own tile run as shifted 3x3 windows of the 3x3 kernels -- the PatchGAN's 4x4
 * stride-1 layers (discriminators.py:543-560: weight gradient = four windows) and the ResnetGenerator's 7x7 reflection-padded
 * first / last convolutions (ResNet_arch.py:52-55,86-88: nine 3x3 blocks of taps).                                      */
int tnr_window2d(tnr_view src, int32_t Hs, int32_t Ws, tnr_view dst, int32_t N, int32_t Hd, int32_t Wd, int32_t C, int32_t oy,
                 int32_t ox, int32_t acc, void *stream);
int tnr_tanh_fwd(const float *x, float *y, int64_t n, void *stream);
int tnr_tanh_bwd(const float *g, const float *y, float *gx, int64_t n, void *stream);
int tnr_gan_loss(const float *pred, int64_t n, int32_t type, float target, float *out, float *grad, void *stream);
/* --- data-parallel collectives (RCCL over xGMI; replaces nn.DataParallel's reduce_add of replica gradients,
 * networks.py:252-255).  One communicator per process.  Rank 0 fills a 128-byte id with tnr_dp_unique_id and shares it
 * out of band; every rank calls tnr_dp_init(id, rank, world, &comm).  Per optimiser step: tnr_dp_allreduce_bucket(comm,
 * bucket, count, average = 1, side_stream) for every gradient bucket as soon as its last gradient kernel is enqueued
 * (in place; ncclAvg = mean over ranks), tnr_dp_broadcast to align replicas at start, tnr_dp_finalize at exit.
 * librccl is dlopen()ed on first use.  Calls on one communicator must be issued in the same order on every rank.     */
int tnr_dp_unique_id(void *id128);
int tnr_dp_init(const void *id128, int32_t rank, int32_t world, void **comm);
int tnr_dp_allreduce_bucket(void *comm, float *buf, int64_t count, int32_t average, void *stream);
int tnr_dp_broadcast(void *comm, float *buf, int64_t count, int32_t root, void *stream);
/* the communicator's own rank count (ncclCommCount): what bench.py reports as config.world_size_observed */
int tnr_dp_comm_count(void *comm, int32_t *nranks);
int tnr_dp_finalize(void *comm);
/* Real-ESRGAN style degradations on fp32 NCHW images in [0, 1] (the LR synthesis the reference runs per sample with
 * OpenCV on DataLoader workers: dataops/augmentations.py:1666-1801, options/presets/resrgan_*.yaml).
 * tnr_filter2d: cv2.filter2D semantics (correlation, centred anchor, BORDER_REFLECT_101) with one 21 x 21 kernel slot per
 *   image (smaller kernels centred, zeros around) -- iso / aniso Gaussian and sinc blurs of 7..21 taps.
 * tnr_resize: cv2.resize semantics for mode 0 INTER_AREA, 1 INTER_LINEAR, 2 INTER_CUBIC on NC planes of H x W -> Ho x Wo.
 * tnr_noise_gaussian: x += sigma255[n][c] / 255 * N(0, 1) (grey[n]: one draw per pixel for all channels, sigma255[n][0]).
 * tnr_noise_poisson: x += scale[n] * (Poisson(x * vals[n]) / vals[n] - x), the difference reduced to luma when grey[n].
 *   Both in place, counter-based RNG (Philox4x32-10 keyed by seed and sample): reproducible for a given seed.
 * tnr_jpeg_sim: in place JPEG round trip of RGB images at quality[n] (JFIF YCbCr, 4:2:0, 8x8 DCT, Annex-K tables with
 *   libjpeg's quality scaling, triangle chroma up-sampling); ws: tnr_jpeg_workspace_bytes(N, H, W).            */
int tnr_filter2d(const float *src, float *dst, const float *kernels21, int32_t N, int32_t C, int32_t H, int32_t W, void *stream);
int tnr_resize(const float *src, float *dst, int32_t NC, int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t mode, void *stream);
int tnr_noise_gaussian(float *img, int32_t N, int32_t C, int32_t H, int32_t W, const float *sigma255, const int32_t *grey,
                       uint64_t seed, int32_t clip, void *stream);
int tnr_noise_poisson(float *img, int32_t N, int32_t C, int32_t H, int32_t W, const float *vals, const float *scale,
                      const int32_t *grey, uint64_t seed, int32_t clip, void *stream);
int64_t tnr_jpeg_workspace_bytes(int32_t N, int32_t H, int32_t W);
int tnr_jpeg_sim(float *img, int32_t N, int32_t H, int32_t W, const int32_t *quality, float *ws, int64_t ws_bytes, void *stream);
/* Input feed (wire format: uint8 HWC BGR crop windows, as OpenCV hands the reference's datasets): one launch per batch
 * does the paired flip / rot90 of dataops/augmentations.py:790-830 and np2tensor of dataops/common.py:470-499
 * (x * data_range / 255, HWC -> CHW, BGR(A) -> RGB(A), optional norm()).  src [N,H,W,C] uint8 (device), dst
 * [N,C,H,W] float; flags[n] (device int32, may be NULL): bit 0 flip (np.flip axis 1), bit 1 rot (np.rot90 k = 1),
 * bit 2 vflip (np.flip axis 0 before the rotation; only with rot).  any_rot != 0 requires H == W.            */
int tnr_feed_u8_to_tensor(const uint8_t *src, int32_t N, int32_t H, int32_t W, int32_t C, const int32_t *flags, int32_t any_rot,
                          float *dst, int32_t bgr2rgb, float data_range, int32_t normalize, void *stream);
/* Validation metrics on the device (the reference does these on the host: tensor2np dataops/common.py:502-566,
 * calculate_psnr / calculate_ssim utils/metrics.py:110-126,180-223, driven by train.py:335-372).
 * tnr_tensor2np_u8: NCHW float -> NHWC uint8 = round_half_even(clip(255 * x, 0, 255)), x -> (x + 1) / 2 first if
 *   denormalize, channels RGB(A) -> BGR(A) if rgb2bgr (every image of the batch; the reference keeps image 0).
 * tnr_psnr_ssim_u8: per image n, out[4n .. 4n+3] = {sum of squared differences, element count, sum of the SSIM map,
 *   SSIM map element count} over the images cropped by `crop` border pixels (SSIM: 11x11 Gaussian window sigma 1.5,
 *   'valid' region, fp64; count 0 if the cropped image is smaller than the window or want_ssim == 0).
 *   PSNR = 20 log10(255 / sqrt(sum / count)); SSIM = sum / count.  ws: tnr_metrics_workspace_bytes(N).           */
int tnr_tensor2np_u8(const float *src, int32_t N, int32_t C, int32_t H, int32_t W, uint8_t *dst, int32_t rgb2bgr,
                     int32_t denormalize, void *stream);
int64_t tnr_metrics_workspace_bytes(int32_t N);
int tnr_psnr_ssim_u8(const uint8_t *a, const uint8_t *b, int32_t N, int32_t H, int32_t W, int32_t C, int32_t crop,
                     int32_t want_ssim, double *out, double *ws, int64_t ws_bytes, void *stream);
/* Bilinear x2 up-sampling, align_corners = False, and its adjoint -- F.interpolate(x, scale_factor=2, mode='bilinear',
 * align_corners=False) between the decoder convolutions of UNetDiscriminator (discriminators.py:745-769).
 * fwd: x [N,H,W,C] -> y [N,2H,2W,C].  bwd: gy [N,2H,2W,C] -> gx (plain, may be null) and / or
 * gz = gx * LeakyReLU'(mask) (may be null; mask is the activation whose sign gates: > 0 -> 1, else mslope).   */
int tnr_bilinear2x_fwd(tnr_view x, tnr_view y, int32_t N, int32_t H, int32_t W, int32_t C, void *stream);
int tnr_bilinear2x_bwd(tnr_view gy, tnr_view gx, tnr_view gz, tnr_view mask, float mslope, int32_t N, int32_t H, int32_t W,
                       int32_t C, void *stream);
/* dst = a + b over `pixels` x C (the U-Net skip connections x4 + x2, x5 + x1, x6 + x0: discriminators.py:755,762,769);
 * dst = src * LeakyReLU'(y) out of place (src stays intact for the skip path of the backward pass).          */
int tnr_add2(tnr_view dst, tnr_view a, tnr_view b, int64_t pixels, int32_t C, void *stream);
int tnr_mask_copy(tnr_view dst, tnr_view src, tnr_view y, int64_t pixels, int32_t C, float mslope, void *stream);
/* --- layout / resampling (block.py:326-371 Upsample, :374-387,434-460 PixelShuffle; nn.MaxPool2d) */
int tnr_nchw_to_nhwc(const float *src, int32_t N, int32_t C, int32_t H, int32_t W, tnr_view dst,
                     int32_t Cpad, const float *scale, const float *shift, void *stream);
int tnr_nhwc_to_nchw(tnr_view src, int32_t N, int32_t C, int32_t H, int32_t W, float *dst,
                     const float *scale, int32_t accumulate, void *stream);
int tnr_upsample2x_bwd(tnr_view gup, tnr_view gx, int32_t N, int32_t H, int32_t W, int32_t C,
                       tnr_view mask, float mslope, void *stream);
int tnr_depth_to_space(tnr_view x, tnr_view y, int32_t N, int32_t H, int32_t W, int32_t Cout, void *stream);
int tnr_space_to_depth_bwd(tnr_view gy, tnr_view gx, int32_t N, int32_t H, int32_t W, int32_t Cout,
                           tnr_view mask, float mslope, void *stream);
int tnr_maxpool2_fwd(tnr_view x, tnr_view y, int32_t N, int32_t H, int32_t W, int32_t C, void *stream);
int tnr_maxpool2_bwd(tnr_view gy, tnr_view x, tnr_view gx, int32_t N, int32_t H, int32_t W, int32_t C,
                     void *stream);
int tnr_axpby(tnr_view dst, tnr_view src, int64_t pixels, int32_t C, float a, float b, void *stream);
int tnr_mask_mul(tnr_view g, tnr_view y, int64_t pixels, int32_t C, float mslope, void *stream);
int tnr_fill(float *p, int64_t n, float v, void *stream);

/* --- BatchNorm2d training mode (nn.BatchNorm2d via block.norm, block.py:112-134) + LeakyReLU ---- */
int64_t tnr_bn_workspace_bytes(int32_t C);
int tnr_bn_train_fwd(tnr_view z, tnr_view y, int64_t pixels, int32_t C, const float *gamma,
                     const float *beta, float *running_mean, float *running_var, int64_t *num_batches,
                     float momentum, float eps, float *save_mean, float *save_invstd, int32_t act,
                     float slope, void *ws, void *stream);
/* tnr_bn_train_fwd that also records, in stat64[2 C] (fp64: batch mean, unbiased variance; may be NULL), the exact values of its
 * running-statistics update, and tnr_bn_replay_running, which applies that update (and the batch counter increment) once more:
 * the side effects of a second training-mode forward over the same batch with the same parameters.  The SR step shows the
 * discriminator the real and the fake batch twice (generator stage sr_model.py:170-177 -> losses.py:398-403; discriminator stage
 * :190-193 -> losses.py:471-478) between two discriminator updates: the engine keeps the first pass's activations and replays
 * only these side effects instead of recomputing an identical forward (models/sr_model.py, HipNet.memoize).               */
int tnr_bn_train_fwd_stats(tnr_view z, tnr_view y, int64_t pixels, int32_t C, const float *gamma, const float *beta,
                           float *running_mean, float *running_var, int64_t *num_batches, float momentum, float eps,
                           float *save_mean, float *save_invstd, double *stat64, int32_t act, float slope, void *ws, void *stream);
int tnr_bn_replay_running(float *running_mean, float *running_var, int64_t *num_batches, const double *stat64, int32_t C,
                          float momentum, void *stream);
int tnr_bn_train_bwd(tnr_view gy, tnr_view y, tnr_view z, tnr_view gz, int64_t pixels, int32_t C,
                     const float *gamma, const float *save_mean, const float *save_invstd, float mslope,
                     float *dgamma, float *dbeta, float acc_beta, void *ws, void *stream);
/* The same with the LeakyReLU / ReLU mask RECOMPUTED from z instead of read from y: y > 0 <=> ((z - mean) invstd) gamma + beta > 0 is
 * the forward's own expression (tnr_bn_train_fwd; the library is built with -ffp-contract=off), so the mask -- and every result -- is
 * bit-for-bit that of tnr_bn_train_bwd; one of the two passes' three reads per element disappears.  gamma / beta must be the
 * values the forward ran with (true between a forward and its backward).                                                       */
int tnr_bn_train_bwd_z(tnr_view gy, tnr_view z, tnr_view gz, int64_t pixels, int32_t C, const float *gamma, const float *beta,
                       const float *save_mean, const float *save_invstd, float mslope, float *dgamma, float *dbeta,
                       float acc_beta, void *ws, void *stream);
/* InstanceNorm2d without affine parameters or running statistics (the ResnetGenerator's norm layer, ResNet_arch.py:40-50 ->
 * nn.InstanceNorm2d defaults) over a whole batch: the BatchNorm kernels with one statistics group per image.
 * y = act((z - mean[n,c]) * invstd[n,c]); save_mean / save_invstd: [N * C].  bwd: gz from gy (masked by act'(y), slope mslope:
 * 0 ReLU, 1 none).  ws: tnr_instnorm_workspace_bytes(N, C).                                                            */
int64_t tnr_instnorm_workspace_bytes(int32_t N, int32_t C);
int tnr_instnorm_fwd(tnr_view z, tnr_view y, int32_t N, int64_t pixels_per_image, int32_t C, float eps, float *save_mean,
                     float *save_invstd, int32_t act, float slope, void *ws, void *stream);
int tnr_instnorm_bwd(tnr_view gy, tnr_view y, tnr_view z, tnr_view gz, int32_t N, int64_t pixels_per_image, int32_t C,
                     const float *save_mean, const float *save_invstd, float mslope, void *ws, void *stream);

/* --- classifier (nn.Linear, discriminators.py:40-45) ------------------------------------------- */
int tnr_linear_fwd(const float *x, const float *w, const float *b, float *y, int32_t N, int32_t In,
                   int32_t Out, int32_t act, float slope, void *stream);
int tnr_linear_bwd(const float *x, const float *w, const float *gy, const float *yact, float mslope,
                   float *gx, float *dw, float *db, int32_t N, int32_t In, int32_t Out, float acc_beta,
                   float *gpre_ws, void *stream);

/* --- losses (nn.L1Loss losses.py:37-39; GANLoss loss.py:61-137; Adversarial losses.py:428-433,503-512) */
int64_t tnr_reduce_workspace_bytes(void);
int tnr_l1_mean_fwd(const float *a, const float *b, int64_t n, float scale, float *loss, void *ws,
                    void *stream);
int tnr_l1_mean_bwd(const float *a, const float *b, int64_t n, float scale, const float *gscale,
                    float *ga, int32_t accumulate, void *stream);
/* relativistic BCE, three phases so that the two 2-scalar sums can be all-reduced between them
 * (SURVEY.md 8(e)); sums = 8 floats on device.  stage 0 = generator, 1 = discriminator.
 * ws: tnr_reduce_workspace_bytes() of scratch for the two-stage (fixed-order) reductions over per-pixel logit maps (n >= 4096:
 * UNetDiscriminator); NULL or smaller n: one block reduces everything.                                                      */
int tnr_ragan_phase_a(const float *pf, const float *pr, int32_t n, float *sums, void *ws, void *stream);
int tnr_ragan_phase_b(const float *pf, const float *pr, int32_t n, int32_t stage, float *sums, void *ws, void *stream);
int tnr_ragan_phase_c(const float *pf, const float *pr, int32_t n, int32_t stage, float weight,
                      const float *sums, float *loss_out, float *gf, float *gr, void *stream);
int tnr_scale_by(float *dst, const float *src, int64_t n, const float *gscale, void *stream);

/* --- optimiser (torch.optim.Adam optimizers.py:130-132; clip_grad_norm_ base_model.py:911-922) -- */
int tnr_sumsq(const float *g, int64_t n, double *out, void *ws, void *stream);
int tnr_clip_by_norm(float *g, int64_t n, const double *sumsq, float max_norm, void *stream);
int tnr_adam_step(float *p, const float *g, float *m, float *v, int64_t n, float step_size, float b1,
                  float b2, float bc2_sqrt, float eps, float weight_decay, void *stream);
/* The same update behind a guard: if *fault (a device word, see tnr_set_fault_word) is nonzero when the launch runs, p / m / v are
 * left untouched -- a step whose activations came out of a timed-out tile hand-off is never applied. */
int tnr_adam_step_guarded(float *p, const float *g, float *m, float *v, int64_t n, float step_size, float b1,
                          float b2, float bc2_sqrt, float eps, float weight_decay, const uint32_t *fault, void *stream);
/* Register ONE caller-owned, zero-initialised device word per process (one process drives one GPU) as the engine's fault latch:
 * the one-launch dense-block kernels (tnr_conv_chain, tnr_conv_sweep) set it to 1 when a bounded wait for a neighbouring tile gives
 * up (instead of the last word of their own workspace, which they use while no latch is registered).  Sticky: nothing in the
 * library ever clears it.  NULL unregisters. */
int tnr_set_fault_word(uint32_t *dev_word);

#ifdef __cplusplus
}
#endif
#endif
