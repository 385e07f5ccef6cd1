// 64-cout 3x3 convolutions (and data-gradients) of TNR_MMA_BF16X3 in the Winograd F(2x2, 3x3) form: 16 transform-domain products per
// 2 x 2 output patch and (cin, cout) pair instead of 36 -- 2.25 x fewer matrix-core instructions for the SAME convolution
// (nn.Conv2d k3 s1 p1, block.py:214-256; what cuDNN selects for these layers under `cudnn.benchmark = True`, codes/train.py:482).
//
//     Y = A^T [ (G g G^T) . (B^T d B) ] A        d: 4 x 4 input patch, g: 3 x 3 filter, Y: 2 x 2 outputs, "." element-wise, summed over cin
//     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]     G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]     A^T = [1 1 1 0; 0 1 -1 -1]
//
// Arithmetic: U = G g G^T is computed in fp64 from the packed fp32 weights, rounded once to fp32 and split exactly into three bf16
// planes by the pack kernel (tnr_conv_wino_pack: once per optimiser step); V = B^T d B is formed in fp32 by the stager (+- only) and
// split exactly into three bf16 planes; the 16 products per transform position xi = (i, j) are the six kept partial products of
// the split arithmetic on v_mfma_f32_32x32x16_bf16, fp32 accumulate -- 16 independent GEMMs [tiles x cin] x [cin x cout]; the inverse
// transform and the usual epilogue (bias / act / residuals / mask / noise: conv_epilogue.h) run on the accumulators.  Error against
// fp64: the transforms add a few fp32 roundings per element on top of the direct form's (tests/test_gpu_kernels.py holds it to <= 3 x
// the fp32 matrix-core path's error + 2e-7 of the scale; the direct forms are held to 1.5 x).  NOT bit-identical to the direct kernels.
//
// Geometry.  Workgroup tile = 16 x 16 output pixels (8 x 8 patches = 64 GEMM rows per xi = two M-tiles) x 64 couts (two N-tiles) x 16 xi
// = 64 accumulator tiles of 32 x 32: 8 waves x 8 tiles (128 registers each), two waves per SIMD, ONE workgroup per CU.
//   wave w: xi-half h = w >> 2, transform row i = 2 h + (w & 1) (all four columns j, BOTH M-tiles), N-tile nt = (w >> 1) & 1.
// The two halves work in ANTI-PHASE, one workgroup barrier per phase:
//     phase 2 c     : half 0 TRANSFORMS chunk c (16 channels): raw fp32 patch in LDS -> its 8 x 64 rows of V (three bf16 planes, the
//                     96-byte row layout of conv_body.h) -- vector ALU + LDS only;   half 1 MULTIPLIES chunk c - 1 (48 MFMAs per wave)
//     phase 2 c + 1 : half 1 transforms chunk c, half 0 multiplies chunk c
// so on every SIMD one wave feeds the matrix core while its partner runs the transform on the vector ALU (the pipes are separate),
// and a V half is single-buffered: the waves that write it are the waves that read it, one phase later.  The weights arrive as a
// pre-split stream in MFMA operand order straight from L2 into registers: the four units of a multiply phase (one per column j, each
// feeding 12 MFMAs) are requested at the top of the transform phase in front of it; the raw input patch of chunk c + 2 (18 x 18 pixels
// x 16 channels, zero / reflected borders resolved by the loader) is requested by both halves at the top of their multiply phase of
// chunk c and written to the free raw buffer at the top of their next transform phase.
// LDS: 2 x 48 KB (V halves) + 2 x 30.4 KB (raw patches, 96-byte pixel stride: conflict-free 16-byte reads at a 2-pixel lane stride).
// Output: a wave holds one transform row of every (patch, cout) of its N-tile; the inverse transform's column step is local, the row step
// is one exchange through LDS among the four waves of an N-tile, after which every wave finishes one output row of one M-tile.
#include <stddef.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "conv_body.h"
#include "conv_epilogue.h"

namespace {

constexpr int WN_T = 16;                            // output pixels per tile side
constexpr int WN_PT = WN_T / 2;                     // patches per tile side
constexpr int WN_NP = WN_PT * WN_PT;                // 64 patches = GEMM rows per transform position
constexpr int WN_RAW_W = WN_T + 2;                  // 18
constexpr int WN_RAW_PIX = WN_RAW_W * WN_RAW_W;     // 324
constexpr int WN_RAW_STRIDE = 24;                   // floats per raw pixel (16 used): rows 2 pixels apart start 192 B apart -> four patches of a
                                                    // ds_read_b128 lane group hit four disjoint 64-byte bank windows
constexpr int WN_RAW_FLOATS = WN_RAW_PIX * WN_RAW_STRIDE;
constexpr int WN_ROW = TNR_X3_ROW;                  // floats per V row: three planes of 16 bf16
constexpr int WN_VH_ROWS = 8 * WN_NP;               // rows of a V half: 8 transform positions x 64 patches
constexpr int WN_VH_FLOATS = WN_VH_ROWS * WN_ROW;
constexpr size_t WN_LDS_BYTES = (size_t)(2 * WN_VH_FLOATS + 2 * WN_RAW_FLOATS) * sizeof(float);      // 160 512 B
constexpr int WN_UNIT_FLOATS = 32 * WN_ROW;         // one transform position x one N-tile x 16 channels: 3 KB (three 1 KB planes in lane order)
constexpr int WN_STAGE_ITEMS = WN_RAW_PIX * 4;      // float4 items of a raw patch: 1296
constexpr int WN_STAGE_IT = (WN_STAGE_ITEMS / 2 + 255) / 256;      // per thread of a half: 3
static_assert(WN_LDS_BYTES <= 160 * 1024, "one Winograd workgroup per CU");
static_assert(8 * 4 * 16 * 64 * sizeof(float) <= WN_LDS_BYTES, "the inverse transform's exchange fits the LDS");
static_assert(WN_STAGE_IT * 256 * 2 >= WN_STAGE_ITEMS, "staging plan covers the patch");

#ifndef WN_PRIO
#define WN_PRIO 3           /* priority of a wave while it multiplies (its SIMD partner transforms on the vector ALU meanwhile) */
#endif
#ifdef WN_TIMELINE          /* probe build: cycles waves 0 (half 0) and 4 (half 1) of every workgroup spend per part of a tile */
__device__ unsigned long long wn_tl[2][16];
#define WN_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define WN_ADD(i, d) do { if (lane == 0 && (wave & 3) == 0) tl[i] += (d); } while (0)
#else
#define WN_T(v) do { } while (0)
#define WN_ADD(i, d) do { } while (0)
#endif

struct WinoK {
    ConvK a;
    const float *wq;                   // [cb][nt][i][chunk][j] units
    int wq_bytes;
    int nck;                           // input chunks (Cin / 16)
    int tiles_x, tiles_y, ncb, tiles;
};

struct WinoPackK {
    const float *wp;                   // packed fp32 weights [tap][KoutP][KinP]
    int KinP, KoutP, nck, ncb, units;
    float *out;
};

// unit u = (((cb * 2 + nt) * 4 + i) * nck + ck) * 4 + j;  thread = (unit, cout row r, channel octet oh)
__global__ void __launch_bounds__(256) wino_pack_kernel(const WinoPackK a) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int unit = g >> 6, r = (g >> 1) & 31, oh = g & 1;
    if (unit >= a.units) return;
    const int j = unit & 3;
    int rest = unit >> 2;
    const int ck = rest % a.nck;
    rest /= a.nck;
    const int i = rest & 3, nt = (rest >> 2) & 1, cb = rest >> 3;
    const int co = cb * 64 + nt * 32 + r;
    // G: row 0 = (1, 0, 0), row 1 = (.5, .5, .5), row 2 = (.5, -.5, .5), row 3 = (0, 0, 1)
    const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    f32x4 q[2];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = 16 * ck + 8 * oh + e;
        double u = 0.0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                u += G[i][ky] * G[j][kx] * (double)a.wp[((size_t)(ky * 3 + kx) * a.KoutP + co) * a.KinP + ch];
        q[e >> 2][e & 3] = (float)u;
    }
    tnr_bf16x8 pl[3];
    tnr_split_bf16x3(q[0], q[1], pl);
    float *dst = a.out + (size_t)unit * WN_UNIT_FLOATS + (oh * 32 + r) * 4;
#pragma unroll
    for (int k = 0; k < 3; ++k) *reinterpret_cast<tnr_bf16x8 *>(dst + k * 256) = pl[k];
}

template <int B, class F, int... I>
__device__ __forceinline__ void wn_static_for_seq(F &f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, B + I>{}), ...);
}
template <int B, int E, class F>
__device__ __forceinline__ void wn_static_for(F &&f) {
    wn_static_for_seq<B>(f, std::make_integer_sequence<int, (E > B ? E - B : 0)>{});
}

__global__ void __launch_bounds__(512, 2) conv3x3_wino_kernel(const WinoK c) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_v = smem;                                   // [2 halves][512 rows][24]
    float *s_raw = smem + 2 * WN_VH_FLOATS;              // [2 buffers][324 pixels][24]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, half = lane >> 5;
    // wave w: half h = w >> 2 (waves w and w + 4 share a SIMD: opposite halves), transform row i = 2 h + (w & 1), N-tile nt = (w >> 1) & 1
    const int wil = wave & 1, wn = (wave >> 1) & 1, wh = wave >> 2;
    const int wi = 2 * wh + wil;
    const int tt = tid & 255;                            // thread of its half
    const ConvK &a = c.a;
    const __amdgpu_buffer_rsrc_t x_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)((unsigned)a.N * a.H * a.W * a.x_ct * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(c.wq), 0, c.wq_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rs =
        __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((unsigned)a.N * a.Ho * a.Wo * a.y_ct * 4u), 0x00020000);
    float *s_vh = s_v + wh * WN_VH_FLOATS;               // this half's V rows

    // ---- transform plan of this thread: patch tp = tt / 4 (py, px), channel quad tq = tt % 4
    const int tp = tt >> 2, tq = tt & 3;
    const int tpy = tp >> 3, tpx = tp & 7;
    // raw rows this half needs: h = 0 -> patch rows 0, 1, 2 (V rows i = 0, 1); h = 1 -> rows 1, 2, 3 (i = 2, 3)
    const int raw_off = ((2 * tpy + wh) * WN_RAW_W + 2 * tpx) * WN_RAW_STRIDE + 4 * tq;
    // V rows written: xl * 64 + tp (xl = 4 (i & 1) + j), 8 bytes of each plane at channel quad tq; fragment rows read: xl * 64 + 32 m + li.
    // The slot swizzle bit (row >> 3) & 1 does not depend on xl or m (multiples of 16 rows): one base address each, the rest are immediates
    static_assert(TNR_X3_SWZ < 4, "the swizzle bit must not depend on the transform position or the M-tile");
    const int v_dst0 = tp * WN_ROW + 4 * ((tq >> 1) ^ ((tp >> TNR_X3_SWZ) & 1)) + 2 * (tq & 1);
    const int a_src0 = (wil * 4 * WN_NP + li) * WN_ROW + 4 * (half ^ ((li >> TNR_X3_SWZ) & 1));
    constexpr int XL_STRIDE = WN_NP * WN_ROW;            // floats between the rows of consecutive transform positions

#ifdef WN_TIMELINE
    unsigned long long tl[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tl_begin = __builtin_amdgcn_s_memtime();
#endif
    for (int tile = blockIdx.x; tile < c.tiles; tile += gridDim.x) {
        // (cb innermost: the ncb workgroups of a pixel tile run side by side and share its input in L2)
        const int cb = tile % c.ncb;
        int rest = tile / c.ncb;
        const int tx = rest % c.tiles_x;
        rest /= c.tiles_x;
        const int ty = rest % c.tiles_y, n = rest / c.tiles_y;
        const int ty0 = ty * WN_T, tx0 = tx * WN_T;
        // ---- staging plan of the raw patch: this half's items i = wh * 648 + tt + 256 it -> pixel i / 4, channel quad i % 4
        int in_off[WN_STAGE_IT], r_dst[WN_STAGE_IT];
#pragma unroll
        for (int it = 0; it < WN_STAGE_IT; ++it) {
            const int il = tt + it * 256;
            const int i = wh * (WN_STAGE_ITEMS / 2) + il, pix = i >> 2, q = i & 3;
            const int pr = pix / WN_RAW_W, pc = pix - pr * WN_RAW_W;
            int Y = ty0 + pr - 1, X = tx0 + pc - 1;
            if (a.reflect) {         // nn.ReflectionPad2d(1) in front of the layer: row / column -1 is 1, H / W is H - 2 / W - 2
                Y = Y == -1 ? 1 : (Y == a.H ? a.H - 2 : Y);
                X = X == -1 ? 1 : (X == a.W ? a.W - 2 : X);
            }
            const bool mine = il < WN_STAGE_ITEMS / 2;
            const bool in = mine & (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
            in_off[it] = in ? (((n * a.H + Y) * a.W + X) * a.x_ct + a.x_co + q * 4) : -1;
            r_dst[it] = mine ? pix * WN_RAW_STRIDE + 4 * q : -1;
        }
        f32x4 rin[WN_STAGE_IT];
        auto raw_load = [&](int ch, bool valid) __attribute__((always_inline)) {        // on every path (past-the-end addresses read zeros)
#pragma unroll
            for (int it = 0; it < WN_STAGE_IT; ++it) {
                const unsigned bo = (valid && in_off[it] >= 0) ? (unsigned)(in_off[it] + ch) * 4u : 0xfffffff0u;
                rin[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (int)bo, 0, 0));
            }
        };
        auto raw_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int it = 0; it < WN_STAGE_IT; ++it)
                if (r_dst[it] >= 0) *reinterpret_cast<f32x4 *>(s_raw + buf * WN_RAW_FLOATS + r_dst[it]) = rin[it];
        };

        f32x16 acc[4][2];                // [j][m]
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][m][r] = 0.f;

        // weight fragments of one multiply phase: the four units (j = 0 .. 3) of this wave's (nt, i) at one chunk -- units 0 and 1 requested at
        // the end of the transform phase in front of it, units 2 and 3 at the top of the multiply phase itself (24 MFMAs = 768 cycles ahead of
        // their first use)
        // (a unit feeds 12 MFMAs: both M-tiles; per CU and chunk 96 KB of weights
        // through the vector memory path, what the direct kernel moves -- the first version gave a wave one M-tile and four units more
        // and was bound by that path: 128 B / cycle / CU needed at the matrix core's rate)
        tnr_bf16x8 fbr[4][3];
        const int bq0 = (((cb * 2 + wn) * 4 + wi) * c.nck) * 4;       // first unit of this wave's part of the stream
        auto b_fetch = [&](int ck, auto j0c) __attribute__((always_inline)) {      // units j0, j0 + 1 of chunk ck
            constexpr int J0 = decltype(j0c)::value;
            const int u0 = bq0 + ck * 4;
#pragma unroll
            for (int j = J0; j < J0 + 2; ++j)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp)
                    fbr[j][sp] = __builtin_bit_cast(tnr_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rs, lane * 16, ((u0 + j) * 3 + sp) * 1024, 0));
        };

        // ---- transform of the chunk in raw buffer `rb` into this half's V rows.  Row step first (per patch column q: the two transform
        // rows of this half from the three patch rows it needs), then the column step per transform row: 8 + 6 values live, not 12 + 8.
        auto transform_h = [&](int rb, auto hc) __attribute__((always_inline)) {
            constexpr int H = decltype(hc)::value;
            const float *src = s_raw + rb * WN_RAW_FLOATS + raw_off;
            f32x4 ea[4], eb[4];
            f32x4 r[4][3];               // all twelve reads in flight at once: one LDS round trip per transform, not four
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int p = 0; p < 3; ++p) r[q][p] = *reinterpret_cast<const f32x4 *>(src + (p * WN_RAW_W + q) * WN_RAW_STRIDE);
            __builtin_amdgcn_sched_barrier(0);
#ifdef WN_TIMELINE
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            { WN_T(ts); WN_ADD(9, ts); }
#endif
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // half 0 (patch rows 0, 1, 2): i = 0: d0 - d2, i = 1: d1 + d2;   half 1 (patch rows 1, 2, 3): i = 2: d2 - d1, i = 3: d1 - d3
                ea[q] = H == 0 ? r[q][0] - r[q][2] : r[q][1] - r[q][0];
                eb[q] = H == 0 ? r[q][1] + r[q][2] : r[q][0] - r[q][2];
            }
            auto colstep = [&](const f32x4 (&e)[4], int j) __attribute__((always_inline)) {
                return j == 0 ? e[0] - e[2] : (j == 1 ? e[1] + e[2] : (j == 2 ? e[2] - e[1] : e[1] - e[3]));
            };
            auto store3 = [&](int xl, const tnr_bf16x4 &h, const tnr_bf16x4 &m, const tnr_bf16x4 &l) __attribute__((always_inline)) {
                float *d = s_vh + v_dst0 + xl * XL_STRIDE;
                *reinterpret_cast<tnr_f32x2 *>(d) = __builtin_bit_cast(tnr_f32x2, h);
                *reinterpret_cast<tnr_f32x2 *>(d + 8) = __builtin_bit_cast(tnr_f32x2, m);
                *reinterpret_cast<tnr_f32x2 *>(d + 16) = __builtin_bit_cast(tnr_f32x2, l);
            };
            // two transform positions at a time, level by level: the split is a chain of seven dependent steps per value, and one position
            // (four channels) alone leaves the vector ALU waiting on itself
            auto emit = [&](const f32x4 (&e)[4], int xl0) __attribute__((always_inline)) {
#ifdef WN_EMIT1
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    tnr_f32x2 pc[3];
                    tnr_split4_bf16x3_pk(colstep(e, j), pc);
                    float *d = s_vh + v_dst0 + (xl0 + j) * XL_STRIDE;
                    *reinterpret_cast<tnr_f32x2 *>(d) = pc[0];
                    *reinterpret_cast<tnr_f32x2 *>(d + 8) = pc[1];
                    *reinterpret_cast<tnr_f32x2 *>(d + 16) = pc[2];
                }
#elif defined(WN_EMIT4)     /* (probe: all four positions of a transform row level by level) */
                f32x4 v[4], r[4], q[4];
                tnr_bf16x4 h[4], m[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = colstep(e, j);
#pragma unroll
                for (int j = 0; j < 4; ++j) tnr_pk_level(v[j], h[j], r[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) tnr_pk_level(r[j], m[j], q[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const tnr_f32x2 a0 = {q[j][0], q[j][1]}, a1 = {q[j][2], q[j][3]};
                    l[j] = __builtin_bit_cast(tnr_bf16x4, tnr_f32x2{__builtin_bit_cast(float, __builtin_convertvector(a0, tnr_bf16x2)),
                                                                   __builtin_bit_cast(float, __builtin_convertvector(a1, tnr_bf16x2))});
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) store3(xl0 + j, h[j], m[j], l[j]);
#else
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const f32x4 va = colstep(e, j), vb = colstep(e, j + 1);
                    tnr_bf16x4 ha, hb, ma, mb;
                    f32x4 ra, rb, qa, qb;
                    tnr_pk_level(va, ha, ra);
                    tnr_pk_level(vb, hb, rb);
                    tnr_pk_level(ra, ma, qa);
                    tnr_pk_level(rb, mb, qb);
                    const tnr_f32x2 a0 = {qa[0], qa[1]}, a1 = {qa[2], qa[3]}, b0 = {qb[0], qb[1]}, b1 = {qb[2], qb[3]};
                    const tnr_bf16x4 la = __builtin_bit_cast(tnr_bf16x4, tnr_f32x2{__builtin_bit_cast(float, __builtin_convertvector(a0, tnr_bf16x2)),
                                                                                 __builtin_bit_cast(float, __builtin_convertvector(a1, tnr_bf16x2))});
                    const tnr_bf16x4 lb = __builtin_bit_cast(tnr_bf16x4, tnr_f32x2{__builtin_bit_cast(float, __builtin_convertvector(b0, tnr_bf16x2)),
                                                                                 __builtin_bit_cast(float, __builtin_convertvector(b1, tnr_bf16x2))});
                    store3(xl0 + j, ha, ma, la);
                    store3(xl0 + j + 1, hb, mb, lb);
                }
#endif
            };
#ifdef WN_TIMELINE
            __builtin_amdgcn_sched_barrier(0);
            { WN_T(ts); WN_ADD(10, ts); }
#endif
            emit(ea, 0);
#ifdef WN_TIMELINE
            __builtin_amdgcn_sched_barrier(0);
            { WN_T(ts); WN_ADD(11, ts); }
#endif
            emit(eb, 4);
#ifdef WN_TIMELINE
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            { WN_T(ts); WN_ADD(12, ts); }
#endif
        };
        auto transform = [&](int rb) __attribute__((always_inline)) {      // (wave-uniform branch: each half runs its own straight-line code)
#ifdef WN_TPRIO
            __builtin_amdgcn_s_setprio(WN_TPRIO);      // (probe: the transforming wave ahead of its multiplying partner)
#endif
            if (wh == 0) transform_h(rb, std::integral_constant<int, 0>{});
            else transform_h(rb, std::integral_constant<int, 1>{});
#ifdef WN_TPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        };

        // ---- multiply: this wave's transform row (4 positions j) x both M-tiles x 16 channels: 8 x 6 MFMAs, fragments of the next
        // (j, m) read from LDS in front of the MFMAs of the current one
        auto multiply = [&]() __attribute__((always_inline)) {
            constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};      // the six kept partial products, smallest first
            tnr_bf16x8 fa[2][3];
#ifndef WN_NOPRIO
            __builtin_amdgcn_s_setprio(WN_PRIO);
#endif
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) fa[0][sp] = *reinterpret_cast<const tnr_bf16x8 *>(s_vh + a_src0 + 8 * sp);
            wn_static_for<0, 8>([&](auto uc) __attribute__((always_inline)) {
                constexpr int u = decltype(uc)::value, j = u >> 1, m = u & 1;
#ifndef WN_X_NOFA           /* (ablation builds: timing only, results invalid) */
                if constexpr (u + 1 < 8) {
                    constexpr int j1 = (u + 1) >> 1, m1 = (u + 1) & 1;
#pragma unroll
                    for (int sp = 0; sp < 3; ++sp)
                        fa[(u + 1) & 1][sp] = *reinterpret_cast<const tnr_bf16x8 *>(s_vh + a_src0 + j1 * XL_STRIDE + m1 * 32 * WN_ROW + 8 * sp);
                }
#else
                if constexpr (u == 0) {
#pragma unroll
                    for (int sp = 0; sp < 3; ++sp) fa[1][sp] = fa[0][sp];
                }
#endif
                __builtin_amdgcn_sched_barrier(0);       // (the next fragments are requested before this unit's MFMAs, and not earlier)
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    acc[j][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u & 1][TA[p]], fbr[j][TB[p]], acc[j][m], 0, 0, 0);
#ifdef WN_MFMA_NOPS
#pragma unroll
                    for (int k = 0; k < WN_MFMA_NOPS; ++k) asm volatile("s_nop 7");      // (probe: leave the issue port to the partner wave)
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            });
#ifndef WN_NOPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        };

        // ---- prologue: raw patches of chunks 0 and 1 (each half stages its share).  (Requesting the NEXT tile's two patches in front of the
        // inverse transform and storing them behind its exchange -- no global-memory latency between tiles -- was 2-7 % SLOWER on every
        // shape: 24 more live registers through the epilogue and one more barrier; profiles/r10d_wino_tile_pipelining_ab.txt)
        __syncthreads();                 // the previous tile's exchange / fragments are consumed
        raw_load(0, true);
        raw_store(0);
        raw_load(16, c.nck > 1);
        if (c.nck > 1) raw_store(1);
        __syncthreads();
        // ---- phases 0 .. 2 nck: half (phase & 1) transforms chunk phase / 2, the other half multiplies the chunk it transformed a phase
        // ago.  Written as ONE straight-line loop body per wave -- transform, barrier, multiply, barrier -- that half 1 enters one
        // barrier late and half 0 leaves one barrier early: with the roles under a run-time branch inside the loop the accumulators
        // were copied in front of every unit's MFMAs (the join of a path that updates them with one that does not) and spilled.
        // The raw patch of chunk ck + 2 is requested at the top of the multiply phase of chunk ck and written to LDS at the top of the
        // transform phase of chunk ck + 1 (a whole phase later; the buffer -- chunk ck's -- was last read two phases ago).
        WN_T(t_loop0);
        if (wh == 1) __syncthreads();
#pragma unroll 1
        for (int ck = 0; ck < c.nck; ++ck) {
            WN_T(t0);
#ifndef WN_X_NORAW
            if (ck >= 1 && ck + 1 < c.nck) raw_store((ck + 1) & 1);      // (at the END of the phase instead: 8 % slower, profiles/r09l_wino_raw_store_late.txt)
            // (the in-kernel timeline shows ~1.6 k cycles at this point, profiles/r10j_wino_timeline_transform_top.txt; requesting the patch two
            //  phases ahead instead of one -- at this place, held through the transform -- made the kernel 3-6 % SLOWER, profiles/r10k: the
            //  cycles are not load latency)
#endif
#if !defined(WN_X_NOB) && defined(WN_B_EARLY)
            b_fetch(ck, std::integral_constant<int, 0>{});
#elif defined(WN_X_NOB)
            if (ck == 0) { b_fetch(0, std::integral_constant<int, 0>{}); b_fetch(0, std::integral_constant<int, 2>{}); }
#endif
            { WN_T(ts); WN_ADD(8, ts); WN_ADD(13, t0); }
#ifndef WN_X_NOT
            transform(ck & 1);
#endif
#if !defined(WN_X_NOB) && !defined(WN_B_EARLY)
            // units 0 and 1 go out at the END of the transform (units 2 and 3 at the top of the multiply phase): requested at its top they
            // cost the transform 24 registers it schedules better without -- x 1.03-1.07 on every shape (profiles/r09o_wino_blate_ab.txt), and the
            // multiply phase, which has slack next to the transform, opens on the L2 latency instead
            b_fetch(ck, std::integral_constant<int, 0>{});
#endif
            WN_T(t1);
            __syncthreads();
            WN_T(t2);
#ifndef WN_X_NOB
            b_fetch(ck, std::integral_constant<int, 2>{});
#endif
#ifndef WN_X_NORAW
            raw_load(16 * (ck + 2), ck + 2 < c.nck);
#endif
#ifndef WN_X_NOM
            multiply();
#endif
            WN_T(t3);
            __syncthreads();
            WN_T(t4);
            WN_ADD(0, t1 - t0); WN_ADD(1, t2 - t1); WN_ADD(2, t3 - t2); WN_ADD(3, t4 - t3); WN_ADD(7, 1ull);
        }
        if (wh == 0) __syncthreads();
        WN_T(t_loop1);
        WN_ADD(4, t_loop1 - t_loop0);

#ifdef WN_X_NOEPI
        if (a.alpha == 1.2345e30f)
#endif
        {
        // ---- inverse transform.  Column step (local): R[i][l] = sum_j M[i][j] A^T[l][j] for this wave's row i, both M-tiles;
        // row step across the four waves of an N-tile through LDS: Y[0][l] = (R0 + R1) + R2, Y[1][l] = (R1 - R2) - R3.
        // Slot of wave w = [w][l][m][r][lane] (16 KB; the whole LDS is free behind the loop's last barrier).
        {
            float *xch = smem + wave * (4 * 16 * 64);
#pragma unroll
            for (int l = 0; l < 2; ++l)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const f32x16 rr = l == 0 ? (acc[0][m] + acc[1][m]) + acc[2][m] : (acc[1][m] - acc[2][m]) - acc[3][m];
#pragma unroll
                    for (int r = 0; r < 16; ++r) xch[((l * 2 + m) * 16 + r) * 64 + lane] = rr[r];
                }
        }
        __syncthreads();
        // this wave finishes output row k = w & 1 of the patches of M-tile m = h, couts of N-tile nt: rows i' = k, k + 1, k + 2 of the four
        // waves (h', il') of its N-tile, i' = 2 h' + il' -> wave index (i' >> 1) * 4 + nt * 2 + (i' & 1)
        f32x16 keep[2][1];
        {
            const int k = wil, m = wh;
            auto slot = [&](int ip, int l) __attribute__((always_inline)) {
                return smem + (((ip >> 1) * 4 + wn * 2 + (ip & 1)) * 4 + l * 2 + m) * (16 * 64) + lane;
            };
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                const float *p0 = slot(k, l), *p1 = slot(k + 1, l), *p2 = slot(k + 2, l);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float x0 = p0[r * 64], x1 = p1[r * 64], x2 = p2[r * 64];
                    keep[l][0][r] = k == 0 ? (x0 + x1) + x2 : (x0 - x1) - x2;
                }
            }
        }
        // ---- epilogue: 32 patches x 32 couts, output row k of every patch, columns l = 0, 1
        conv_epilogue_dpp_map<TNR_CONV_3x3, WN_T, 1, 2, false, 1>(a, keep, cb * 2 + wn, n, ty0, tx0, 0, li, half, y_rs,
                                                                  [&](int mi, int row, int &rr, int &cc) __attribute__((always_inline)) {
                                                                      const int p = 32 * wh + row;
                                                                      rr = 2 * (p >> 3) + wil;
                                                                      cc = 2 * (p & 7) + mi;
                                                                  });
        }
    }
#ifdef WN_TIMELINE
    if (lane == 0 && (wave & 3) == 0) {
        tl[5] = __builtin_amdgcn_s_memtime() - tl_begin;
        for (int i = 0; i < 16; ++i) atomicAdd(&wn_tl[wh][i], tl[i]);
    }
#endif
}

// =====================================================================================================================================
// The sixteen-wave form (TNR_WINO_WAVES=16; probe).  Same tile, stream, LDS image and arithmetic as conv3x3_wino_kernel -- another division
// of the work: FOUR waves per SIMD (128 registers), two of each half, so that a transforming wave's dependent vector-ALU chains and LDS
// round trips are covered by a second transforming wave and a multiplying wave's fragment waits by a second multiplying wave.
//   wave w: half h = (w >> 2) & 1 (waves w, w + 4, w + 8, w + 12 share a SIMD: two of each half); role index r = (w & 3) | ((w >> 3) << 2):
//   transform row i = 2 h + (r & 1), N-tile nt = (r >> 1) & 1, column pair jp = r >> 2 (columns j = 2 jp, 2 jp + 1), BOTH M-tiles:
//   4 accumulator tiles (64 registers), a weight unit feeds 12 MFMAs as before.
//   Transform: the 512 threads of a half take (patch, channel quad, row selector): ONE transform row per thread (4 positions).
//   Inverse transform: column pairs are joined through LDS first (jp = 1 -> jp = 0), then the rows as in the eight-wave form; every wave
//   finishes ONE (output row k, output column l, M-tile) of its N-tile.
constexpr int W16_STAGE_IT = (WN_STAGE_ITEMS / 2 + 511) / 512;      // raw-patch items per thread of a half: 2

__global__ void __launch_bounds__(1024, 4) conv3x3_wino16_kernel(const WinoK c) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_v = smem;
    float *s_raw = smem + 2 * WN_VH_FLOATS;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, half = lane >> 5;
    const int wh = (wave >> 2) & 1, role = (wave & 3) | ((wave >> 3) << 2);
    const int wil = role & 1, wn = (role >> 1) & 1, wjp = role >> 2;
    const int wi = 2 * wh + wil;
    const int tt = role * 64 + lane;                     // thread of its half (0 .. 511)
    const ConvK &a = c.a;
    const __amdgpu_buffer_rsrc_t x_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)((unsigned)a.N * a.H * a.W * a.x_ct * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(c.wq), 0, c.wq_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rs =
        __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((unsigned)a.N * a.Ho * a.Wo * a.y_ct * 4u), 0x00020000);
    float *s_vh = s_v + wh * WN_VH_FLOATS;

    // transform plan: patch tp, channel quad tq, row selector rs (wave-uniform): this thread builds V row i_t = 2 h + rs of its patch
    const int tp = (tt >> 2) & 63, tq = tt & 3, rs = tt >> 8;
    const int tpy = tp >> 3, tpx = tp & 7;
    // the two patch rows the transform row needs and the sign of the second: i = 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
    const int it_ = 2 * wh + rs;
    const int pa = it_ == 0 ? 0 : (it_ == 2 ? 2 : 1), pb = it_ == 0 ? 2 : (it_ == 1 ? 2 : (it_ == 2 ? 1 : 3));
    const bool plus = it_ == 1;
    const int raw_a = ((2 * tpy + pa) * WN_RAW_W + 2 * tpx) * WN_RAW_STRIDE + 4 * tq;
    const int raw_b = ((2 * tpy + pb) * WN_RAW_W + 2 * tpx) * WN_RAW_STRIDE + 4 * tq;
    static_assert(TNR_X3_SWZ < 4, "the swizzle bit must not depend on the transform position or the M-tile");
    const int v_dst0 = (rs * 4 * WN_NP + tp) * WN_ROW + 4 * ((tq >> 1) ^ ((tp >> TNR_X3_SWZ) & 1)) + 2 * (tq & 1);
    const int a_src0 = ((wil * 4 + 2 * wjp) * WN_NP + li) * WN_ROW + 4 * (half ^ ((li >> TNR_X3_SWZ) & 1));
    constexpr int XL_STRIDE = WN_NP * WN_ROW;

    for (int tile = blockIdx.x; tile < c.tiles; tile += gridDim.x) {
        const int cb = tile % c.ncb;
        int rest = tile / c.ncb;
        const int tx = rest % c.tiles_x;
        rest /= c.tiles_x;
        const int ty = rest % c.tiles_y, n = rest / c.tiles_y;
        const int ty0 = ty * WN_T, tx0 = tx * WN_T;
        int in_off[W16_STAGE_IT], r_dst[W16_STAGE_IT];
#pragma unroll
        for (int it = 0; it < W16_STAGE_IT; ++it) {
            const int il = tt + it * 512;
            const int i = wh * (WN_STAGE_ITEMS / 2) + il, pix = i >> 2, q = i & 3;
            const int pr = pix / WN_RAW_W, pc = pix - pr * WN_RAW_W;
            int Y = ty0 + pr - 1, X = tx0 + pc - 1;
            if (a.reflect) {
                Y = Y == -1 ? 1 : (Y == a.H ? a.H - 2 : Y);
                X = X == -1 ? 1 : (X == a.W ? a.W - 2 : X);
            }
            const bool mine = il < WN_STAGE_ITEMS / 2;
            const bool in = mine & (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
            in_off[it] = in ? (((n * a.H + Y) * a.W + X) * a.x_ct + a.x_co + q * 4) : -1;
            r_dst[it] = mine ? pix * WN_RAW_STRIDE + 4 * q : -1;
        }
        f32x4 rin[W16_STAGE_IT];
        auto raw_load = [&](int ch, bool valid) __attribute__((always_inline)) {
#pragma unroll
            for (int it = 0; it < W16_STAGE_IT; ++it) {
                const unsigned bo = (valid && in_off[it] >= 0) ? (unsigned)(in_off[it] + ch) * 4u : 0xfffffff0u;
                rin[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (int)bo, 0, 0));
            }
        };
        auto raw_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int it = 0; it < W16_STAGE_IT; ++it)
                if (r_dst[it] >= 0) *reinterpret_cast<f32x4 *>(s_raw + buf * WN_RAW_FLOATS + r_dst[it]) = rin[it];
        };

        f32x16 acc[2][2];                // [jj][m]
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[jj][m][r] = 0.f;

        tnr_bf16x8 fbr[2][3];
        const int bq0 = (((cb * 2 + wn) * 4 + wi) * c.nck) * 4 + 2 * wjp;       // this wave's two units of a chunk: j = 2 jp, 2 jp + 1
        auto b_fetch = [&](int ck) __attribute__((always_inline)) {
            const int u0 = bq0 + ck * 4;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp)
                    fbr[jj][sp] = __builtin_bit_cast(tnr_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rs, lane * 16, ((u0 + jj) * 3 + sp) * 1024, 0));
        };
        auto transform = [&](int rb) __attribute__((always_inline)) {
            const float *sa_ = s_raw + rb * WN_RAW_FLOATS + raw_a, *sb_ = s_raw + rb * WN_RAW_FLOATS + raw_b;
            f32x4 e[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 da = *reinterpret_cast<const f32x4 *>(sa_ + q * WN_RAW_STRIDE), db = *reinterpret_cast<const f32x4 *>(sb_ + q * WN_RAW_STRIDE);
                e[q] = plus ? da + db : da - db;
            }
            auto store3 = [&](int j, const tnr_bf16x4 &h_, const tnr_bf16x4 &m_, const tnr_bf16x4 &l_) __attribute__((always_inline)) {
                float *d = s_vh + v_dst0 + j * XL_STRIDE;
                *reinterpret_cast<tnr_f32x2 *>(d) = __builtin_bit_cast(tnr_f32x2, h_);
                *reinterpret_cast<tnr_f32x2 *>(d + 8) = __builtin_bit_cast(tnr_f32x2, m_);
                *reinterpret_cast<tnr_f32x2 *>(d + 16) = __builtin_bit_cast(tnr_f32x2, l_);
            };
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 v = j == 0 ? e[0] - e[2] : (j == 1 ? e[1] + e[2] : (j == 2 ? e[2] - e[1] : e[1] - e[3]));
                tnr_bf16x4 hh, mm;
                f32x4 r1, r2;
                tnr_pk_level(v, hh, r1);
                tnr_pk_level(r1, mm, r2);
                const tnr_f32x2 a0 = {r2[0], r2[1]}, a1 = {r2[2], r2[3]};
                const tnr_bf16x4 ll = __builtin_bit_cast(tnr_bf16x4, tnr_f32x2{__builtin_bit_cast(float, __builtin_convertvector(a0, tnr_bf16x2)),
                                                                             __builtin_bit_cast(float, __builtin_convertvector(a1, tnr_bf16x2))});
                store3(j, hh, mm, ll);
            }
        };
        auto multiply = [&]() __attribute__((always_inline)) {
            constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
            tnr_bf16x8 fa[2][3];
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) fa[0][sp] = *reinterpret_cast<const tnr_bf16x8 *>(s_vh + a_src0 + 8 * sp);
            wn_static_for<0, 4>([&](auto uc) __attribute__((always_inline)) {
                constexpr int u = decltype(uc)::value, jj = u >> 1, m = u & 1;
                if constexpr (u + 1 < 4) {
                    constexpr int j1 = (u + 1) >> 1, m1 = (u + 1) & 1;
#pragma unroll
                    for (int sp = 0; sp < 3; ++sp)
                        fa[(u + 1) & 1][sp] = *reinterpret_cast<const tnr_bf16x8 *>(s_vh + a_src0 + j1 * XL_STRIDE + m1 * 32 * WN_ROW + 8 * sp);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 6; ++p)
                    acc[jj][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u & 1][TA[p]], fbr[jj][TB[p]], acc[jj][m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        };

        __syncthreads();
        raw_load(0, true);
        raw_store(0);
        raw_load(16, c.nck > 1);
        if (c.nck > 1) raw_store(1);
        __syncthreads();
        if (wh == 1) __syncthreads();
#pragma unroll 1
        for (int ck = 0; ck < c.nck; ++ck) {
            if (ck >= 1 && ck + 1 < c.nck) raw_store((ck + 1) & 1);
            transform(ck & 1);
            b_fetch(ck);
            __syncthreads();
            raw_load(16 * (ck + 2), ck + 2 < c.nck);
            multiply();
            __syncthreads();
        }
        if (wh == 0) __syncthreads();

        // ---- inverse transform.  Column step over this wave's two columns: l = 0: (M0 + M1) + M2, l = 1: (M1 - M2) - M3
        //   jp = 0 holds M0, M1: P[0] = M0 + M1, P[1] = M1;   jp = 1 holds M2, M3: P[0] = M2, P[1] = -M2 - M3   ->   R[l] = P0[l] + P1[l]
        // (the same order of additions as the eight-wave form: (M0 + M1) + M2 and (M1 - M2) - M3 = M1 + (-M2 - M3) differs in the last bit
        //  -- the probe's results are compared with a tolerance, not bit for bit)
        f32x16 pr_[2][2];               // [l][m]
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (wjp == 0) {
                pr_[0][m] = acc[0][m] + acc[1][m];
                pr_[1][m] = acc[1][m];
            } else {
                pr_[0][m] = acc[0][m];
                pr_[1][m] = -acc[0][m] - acc[1][m];
            }
        }
        // step A: jp = 1 waves hand their partials to their jp = 0 partner (same h, il, nt: wave ^ 8); slot = [role & 3][h][l][m][r][lane]
        float *xa = smem + (((wave & 3) | (wh << 2)) * 4) * (16 * 64);
        if (wjp == 1) {
#pragma unroll
            for (int l = 0; l < 2; ++l)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) xa[((l * 2 + m) * 16 + r) * 64 + lane] = pr_[l][m][r];
        }
        __syncthreads();
        if (wjp == 0) {
#pragma unroll
            for (int l = 0; l < 2; ++l)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) pr_[l][m][r] = l == 0 ? pr_[l][m][r] + xa[((l * 2 + m) * 16 + r) * 64 + lane]
                                                                       : pr_[l][m][r] + xa[((l * 2 + m) * 16 + r) * 64 + lane];
        }
        __syncthreads();
        // step B: the jp = 0 waves now hold R[i][l][m] of their (i, nt): written to slot [i][nt] = the eight-wave form's layout
        float *xb = smem + ((wh * 4 + wn * 2 + wil) * 4) * (16 * 64);
        if (wjp == 0) {
#pragma unroll
            for (int l = 0; l < 2; ++l)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) xb[((l * 2 + m) * 16 + r) * 64 + lane] = pr_[l][m][r];
        }
        __syncthreads();
        // every wave finishes ONE (output row k, column l, M-tile m) of its N-tile: k = il, m = h, l = jp
        f32x16 keep[1][1];
        {
            const int k = wil, m = wh, l = wjp;
            auto slot = [&](int ip) __attribute__((always_inline)) {
                return smem + (((ip >> 1) * 4 + wn * 2 + (ip & 1)) * 4 + l * 2 + m) * (16 * 64) + lane;
            };
            const float *p0 = slot(k), *p1 = slot(k + 1), *p2 = slot(k + 2);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x0 = p0[r * 64], x1 = p1[r * 64], x2 = p2[r * 64];
                keep[0][0][r] = k == 0 ? (x0 + x1) + x2 : (x0 - x1) - x2;
            }
        }
        conv_epilogue_dpp_map<TNR_CONV_3x3, WN_T, 1, 1, false, 1>(a, keep, cb * 2 + wn, n, ty0, tx0, 0, li, half, y_rs,
                                                                  [&](int, int row, int &rr, int &cc) __attribute__((always_inline)) {
                                                                      const int p = 32 * wh + row;
                                                                      rr = 2 * (p >> 3) + wil;
                                                                      cc = 2 * (p & 7) + wjp;
                                                                  });
    }
}

bool wino_ok(const tnr_conv_desc *d) {
    return d->mode == TNR_CONV_3x3 && d->mma == TNR_MMA_BF16X3 && (d->pad_mode == 0 || (d->pad_mode == 1 && d->H >= 2 && d->W >= 2)) &&
           (d->Cout % 64) == 0 && d->KoutP == d->Cout && d->Cin == d->KinP && (d->Cin % 16) == 0 && d->Cin >= 32 && d->Ho == d->H &&
           d->Wo == d->W && d->W >= 8 && d->H >= 8 && (int64_t)d->N * d->H * d->W * d->x.ctot < (1LL << 30) &&
           (int64_t)d->N * d->H * d->W * d->y.ctot < (1LL << 30) &&
           (int64_t)(d->Cout / 64) * (d->Cin / 16) * 32 * WN_UNIT_FLOATS * (int64_t)sizeof(float) < (1LL << 31);
}

}  // namespace

#ifdef WN_TIMELINE
extern "C" int tnr_debug_wino_timeline(unsigned long long *out16, int reset) {
    if (out16 != nullptr && hipMemcpyFromSymbol(out16, HIP_SYMBOL(wn_tl), sizeof(wn_tl)) != hipSuccess) return TNR_ELAUNCH;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(wn_tl), z, sizeof(z)) != hipSuccess) return TNR_ELAUNCH;
    }
    return TNR_OK;
}
#endif

extern "C" int64_t tnr_conv_wino_bytes(const tnr_conv_desc *d) {
    if (d == nullptr || !wino_ok(d)) return 0;
    return (int64_t)(d->Cout / 64) * (d->Cin / 16) * 32 * WN_UNIT_FLOATS * (int64_t)sizeof(float);
}

extern "C" int tnr_conv_wino_pack(const tnr_conv_desc *d, void *image, int64_t image_bytes, void *stream) {
    TNR_REQUIRE(d != nullptr && image != nullptr && d->wp != nullptr && wino_ok(d), "conv_wino_pack: the launch cannot run in the Winograd form");
    WinoPackK a;
    a.wp = d->wp; a.KinP = d->KinP; a.KoutP = d->KoutP; a.nck = d->Cin / 16; a.ncb = d->Cout / 64;
    a.units = a.ncb * a.nck * 32;
    TNR_REQUIRE((int64_t)a.units * WN_UNIT_FLOATS * (int64_t)sizeof(float) <= image_bytes, "conv_wino_pack: image buffer too small");
    a.out = static_cast<float *>(image);
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)tnr_cdiv(a.units * 64, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return tnr_check_launch("conv_wino_pack");
}

// called by tnr_conv_forward (conv_tile.hip) for a launch whose weight stream is the transform-domain one (wq_form = 1); 1: not for this kernel
int tnr_launch_conv3x3_wino(const tnr_conv_desc *d, void *stream) {
    if (!wino_ok(d) || d->wq == nullptr || d->wq_bytes < tnr_conv_wino_bytes(d) || d->noise_pos < 0 || d->noise_pos > 2) return 1;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WN_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_wino16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WN_LDS_BYTES) != hipSuccess ||
            cus < 1) {
            cus = 0;
            tnr_set_error("conv3x3_wino: cannot set up the kernel");
            return TNR_ELAUNCH;
        }
    }
    WinoK c;
    ConvK &k = c.a;
    k.x = d->x.ptr; k.x_ct = d->x.ctot; k.x_co = d->x.coff;
    k.N = d->N; k.H = d->H; k.W = d->W; k.Cin = d->Cin;
    k.wp = d->wp; k.KinP = d->KinP; k.KoutP = d->KoutP;
    k.y = d->y.ptr; k.y_ct = d->y.ctot; k.y_co = d->y.coff; k.Ho = d->Ho; k.Wo = d->Wo; k.Cout = d->Cout;
    k.bias = d->bias; k.act = d->act; k.slope = d->slope; k.alpha = d->alpha;
    k.r1 = d->r1.ptr; k.r1_ct = d->r1.ctot; k.r1_co = d->r1.coff; k.r1_ch = d->r1_ch; k.beta1 = d->beta1;
    k.r2 = d->r2.ptr; k.r2_ct = d->r2.ctot; k.r2_co = d->r2.coff; k.alpha2 = d->alpha2;
    k.m = d->m.ptr; k.m_ct = d->m.ctot; k.m_co = d->m.coff; k.m_lo = d->m_lo; k.m_hi = d->m_hi; k.m_slope = d->m_slope;
    k.noise_pos = d->noise_pos; k.noise_sigma = d->noise_sigma; k.noise_k0 = d->noise_key0; k.noise_k1 = d->noise_key1; k.noise_pix0 = d->noise_pix0;
    k.th_space = d->Ho; k.tw_space = d->Wo;
    k.ksplit = 1; k.split_stride = 0; k.bf = d->mma; k.reflect = d->pad_mode == 1;
    c.wq = static_cast<const float *>(d->wq);
    c.wq_bytes = (int)tnr_conv_wino_bytes(d);
    c.nck = d->Cin / 16;
    c.tiles_x = tnr_cdiv(d->Wo, WN_T);
    c.tiles_y = tnr_cdiv(d->Ho, WN_T);
    c.ncb = d->Cout / 64;
    const int64_t tiles = (int64_t)c.tiles_x * c.tiles_y * c.ncb * d->N;
    if (tiles >= (1LL << 31)) return 1;
    c.tiles = (int)tiles;
    k.tiles_x = c.tiles_x; k.tiles_y = c.tiles_y; k.ncb = c.ncb;
    const int grid = c.tiles < cus ? c.tiles : cus;
    static const bool w16 = [] { const char *e = getenv("TNR_WINO_WAVES"); return e != nullptr && atoi(e) == 16; }();
    if (w16) {
        hipLaunchKernelGGL(conv3x3_wino16_kernel, dim3((unsigned)grid), dim3(1024), WN_LDS_BYTES, (hipStream_t)stream, c);
        return tnr_check_launch("conv3x3_wino16");
    }
    hipLaunchKernelGGL(conv3x3_wino_kernel, dim3((unsigned)grid), dim3(512), WN_LDS_BYTES, (hipStream_t)stream, c);
    return tnr_check_launch("conv3x3_wino");
}
