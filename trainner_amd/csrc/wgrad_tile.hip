// Weight gradient as a pixel-reduction GEMM on the fp32 matrix cores:
//     dW[tap][co][ci] = sum_pixels g[p][co] * x[p + tap][ci]
// MFMA 32x32x2: A[i=co][k=pixel parity], B[k][j=ci]; both fragments are conflict-free ds_read_b32
// from NHWC LDS tiles (lanes = consecutive channels).  A workgroup (4 waves, 2 per CU) owns a
// (32*A_T couts) x (32*B_T cins) x all-taps block and walks a contiguous range of TH x 16 pixel
// tiles (its "split"), keeping every accumulator in registers; the A_T*B_T*taps 32x32 tiles are
// dealt round-robin to the 4 waves.  Each split writes one partial slab
//     ws[split][tap][KoutP][KinVP]
// and wgrad_reduce sums the slabs in a fixed order (deterministic split-K) into the OIHW gradient.
// The 4x4 s2 convolution is handled in its space-to-depth form (2x2 taps over 4*Cin virtual
// channels), exactly like conv_tile.hip.
#include <type_traits>
#include <utility>
#include "common.h"

namespace {

// A launch carries up to TNR_WGRAD_GROUP_MAX layers that share the pixel geometry and the workgroup tile
// class ("group"): blockIdx.y enumerates (layer, cin block, cout block), so the chip-filling split count
// -- and with it the partial-slab round trip and the per-launch prologue -- is paid once per group
// instead of once per layer.
struct WgJob {
    const float *x; const float *g; float *ws; float *dbp;   // dbp: partial bias sums [KoutP][split] or null
    int x_ct, x_co, Cin, g_ct, g_co, Cout;
    int KoutP, KinVP;             // slab dims
    int cinp32;                   // per-parity padded channel count (S2D); == KinVP otherwise
    int ncib, job_begin;          // blockIdx.y range of this layer: [job_begin, job_begin + ncib * ncob)
};
struct WgK {
    int N, H, W, Ho, Wo;
    int tiles_x, tiles_y, tiles_total, tiles_per_split, nsplits, njobs;
    int bf;                       // operands rounded to bf16 (tnr_wgrad_desc.mma); one setting per launch
    int reflect;                  // TNR_CONV_3x3: x is read with ReflectionPad2d(1) borders (tnr_wgrad_desc.pad_mode); per launch
    WgJob job[TNR_WGRAD_GROUP_MAX];
};

// Two occupancy regimes (chosen by the accumulator count J = ceil(A_T*B_T*taps/4) per wave):
//   J < 9  : <= 200 VGPRs, TWO workgroups per CU; staging loads are issued in batches of 8 float4 per
//            thread and the other workgroup's MFMA phase hides their latency;
//   J >= 9 : 144 accumulator VGPRs, ONE workgroup per CU with the whole 512-entry register file: the
//            global loads of tile i+1 are all issued before the MFMA phase of tile i and written to LDS
//            after it (issue-early / write-late), and the k-loop is unrolled 4 k-steps deep.
//
// Pixel split inside the workgroup (KS): with AB = A_T*B_T < 4 channel blocks the 9*AB tiles do not divide by the
// 4 waves (18 tiles -> 5 slots per wave, 10 % of the issued MFMAs were padding).  For the 3x3 modes the waves are
// therefore grouped: WPG = AB waves per group, KS = 4 / AB groups; each group owns ALL tiles (one (aa, bb) block
// per wave, its 9 taps: J = 9, no padding) over its own 1/KS of the tile's pixel rows; after the last tile the
// groups' accumulators are added in a fixed order through LDS, so the workgroup still writes ONE partial slab.
template <int A_T, int B_T, int NTAPS_>
struct WgCfg {
    static constexpr int AB = A_T * B_T;
    static constexpr int KS = (NTAPS_ == 9 && (AB == 1 || AB == 2)) ? 4 / AB : 1;
    static constexpr int WPG = 4 / KS;                       // waves per pixel group
    static constexpr int J = (AB * NTAPS_ + WPG - 1) / WPG;
    static constexpr bool PIPE = true;
    static constexpr int WAVES_PER_SIMD = (J >= 9) ? 1 : 2;
};

typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x4 __attribute__((ext_vector_type(4)));
typedef short wg_s16x4 __attribute__((ext_vector_type(4)));
typedef float wg_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 wg_bf16x2 __attribute__((ext_vector_type(2)));
// one level of the operand split for four channels with ONE v_cvt_pk_bf16_f32 per channel pair: the packed pair-of-pairs and the residual
// x - float(pk) (conv_body.h's tnr_pk_level; __builtin_convertvector over four channels compiles to six conversions per level)
__device__ __forceinline__ void wg_pk_level(const f32x4 x, wg_bf16x4 &pk, f32x4 &res) {
    const wg_f32x2 a = {x[0], x[1]}, b = {x[2], x[3]};
    const unsigned p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(a, wg_bf16x2));
    const unsigned p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(b, wg_bf16x2));
    pk = __builtin_bit_cast(wg_bf16x4, wg_f32x2{__builtin_bit_cast(float, p0), __builtin_bit_cast(float, p1)});
    const f32x4 back = {__builtin_bit_cast(float, p0 << 16), __builtin_bit_cast(float, p0 & 0xffff0000u),
                        __builtin_bit_cast(float, p1 << 16), __builtin_bit_cast(float, p1 & 0xffff0000u)};
    res = x - back;
}
#ifndef WG_PK_SPLIT
#define WG_PK_SPLIT 1       /* 0: the split through __builtin_convertvector (30 instead of 22 vector instructions per item; bit-identical) */
#endif
typedef __attribute__((address_space(3))) wg_s16x4 wg_lds_s16x4;

// ---- TNR_MMA_BF16X3: the LDS image is PRE-SPLIT -- three bf16 planes (hi, mid, lo) written once by the stager -- and pixel-major,
// which is what this GEMM reduces over: an MFMA operand fragment needs, per lane (= channel), 8 consecutive PIXELS.  gfx950's
// transposing LDS read delivers exactly that: ds_read_b64_tr_b16 lets the 16 lanes of a group fetch 4 pixels x 16 channels (each lane 4
// consecutive channels of one pixel: 8 contiguous bytes) and hands lane i channel i of the 4 pixels (tools/probes/tr_read.hip,
// profiles/r03a_tr_read.txt) -- 6 reads per fragment (2 x 4 pixels x 3 planes) and NO vector arithmetic, against 8 scalar reads +
// ~56 VALU operations per fragment when the split happens at the read (the weight gradient was VALU-bound there: 36 % MFMA-busy).
// Layout of a tile with C channels (a multiple of 32): 768-byte blocks of 4 pixels x 32 channels x 3 planes,
//     byte address (pixel p, channel c, plane s) = ((p >> 2) * (C / 32) + (c >> 5)) * 768 + s * 256 + (p & 3) * 64 + (c & 31) * 2
// so the 4 pixels a half-wave reads are always the four 64-byte quarters of all 64 banks (any starting pixel: the tap offset shifts
// the start), plane and second-read offsets are immediates, and a fragment costs one address computation.
__device__ __forceinline__ int wg_x3_off(int p, int c, int C) {          // byte offset of (pixel p, channel c), plane 0
    return ((p >> 2) * (C >> 5) + (c >> 5)) * 768 + (p & 3) * 64 + (c & 31) * 2;
}

// compile-time loop: f(integral_constant<int, B>) ... f(integral_constant<int, E - 1>) (register arrays need constant indices)
template <int B, class F, int... I>
__device__ __forceinline__ void wg_static_for_seq(F &f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, B + I>{}), ...);          // one flat fold: nothing for the inliner to give up on
}
template <int B, int E, class F>
__device__ __forceinline__ void wg_static_for(F &&f) {
    wg_static_for_seq<B>(f, std::make_integer_sequence<int, (E > B ? E - B : 0)>{});
}

// BF: operands rounded to bf16 in front of the matrix core (tnr_wgrad_desc.mma = TNR_MMA_BF16).  The reduction index of
// this GEMM is the pixel: a 16-pixel tile row is exactly the k = 16 of one v_mfma_f32_32x32x16_bf16 (lane-half h supplies
// pixels h, 2 + h, .., 14 + h: the same 8 values it feeds to 8 fp32 k-steps), so a row costs J MFMAs instead of 8 J.
// WPS: workgroups per CU (= waves per SIMD).  The accumulator count picks it (WgCfg) except for the TNR_MMA_BF16X3 half-height classes
// (plan_wgrad): two workgroups of 4-row tiles, so that one workgroup's refill (global loads, the operand split, LDS writes: ~1/3 of a
// tile's time with one workgroup per CU, during which the matrix core idles) runs under the other's MFMA phase.
// DB (TNR_MMA_BF16X3, plain 3x3 only): ONE workgroup per CU with TWO LDS tile sets -- the pipelined form described at wg_db_phase below.
template <int MODE, int A_T, int B_T, int THG, int BF, int WPS, bool DB = false>
__global__ void __launch_bounds__(256, WPS)
wgrad_tile_kernel(const WgK ga) {
    constexpr bool S2D = (MODE == TNR_CONV_4x4_S2);
    constexpr bool UP = (MODE == TNR_CONV_3x3_UP2);
    constexpr int TWG = 16, PX = THG * TWG;
    constexpr int KH = S2D ? 2 : 3;
    constexpr int NTAPS = KH * KH;
    constexpr int HT = THG + KH - 1, WT = TWG + KH - 1;
    constexpr int COB = 32 * A_T, CIB = 32 * B_T;
    constexpr int AB = A_T * B_T;
    constexpr int KS = WgCfg<A_T, B_T, NTAPS>::KS, WPG = WgCfg<A_T, B_T, NTAPS>::WPG;
    constexpr int T = AB * NTAPS, J = WgCfg<A_T, B_T, NTAPS>::J;
    constexpr int ROWS = THG / KS;                           // pixel rows of the tile one wave group reduces over
    static_assert(THG % KS == 0, "tile rows must split evenly over the pixel groups");
#ifdef TNR_WG_X3_PIPE2
    constexpr bool PIPE = WgCfg<A_T, B_T, NTAPS>::PIPE;
#else
    // (TNR_MMA_BF16X3 with two workgroups per CU: the staging registers of an in-flight next tile do not fit next to 144 accumulators
    //  in 256 registers -- the tile is loaded at its start instead, under the OTHER workgroup's MFMA phase)
    constexpr bool PIPE = WgCfg<A_T, B_T, NTAPS>::PIPE && !(BF == 2 && WPS == 2 && WgCfg<A_T, B_T, NTAPS>::J >= 9);
#endif

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_g = smem;             // PX * COB
    float *s_x = smem + PX * COB;  // HT*WT * CIB

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> SGPR
    const int lane = tid & 63, li = lane & 31, half = lane >> 5;
    const int split = blockIdx.x;
    const int pg = wave / WPG, wq = wave - pg * WPG;            // pixel group, position inside the group (SGPRs)
    int ji = 0;
#pragma unroll
    for (int q = 1; q < TNR_WGRAD_GROUP_MAX; ++q)
        if (q < ga.njobs && (int)blockIdx.y >= ga.job[q].job_begin) ji = q;
    // flatten the per-launch + per-layer fields into the names the body uses (all wave-uniform scalars)
    struct {
        const float *x; int x_ct, x_co; int N, H, W, Cin;
        const float *g; int g_ct, g_co; int Ho, Wo, Cout;
        float *ws; int KoutP, KinVP, cinp32; float *dbp;
        int tiles_x, tiles_y, tiles_total, tiles_per_split, nsplits, reflect;
    } a;
    a.x = ga.job[ji].x; a.x_ct = ga.job[ji].x_ct; a.x_co = ga.job[ji].x_co;
    a.N = ga.N; a.H = ga.H; a.W = ga.W; a.Cin = ga.job[ji].Cin;
    a.g = ga.job[ji].g; a.g_ct = ga.job[ji].g_ct; a.g_co = ga.job[ji].g_co;
    a.Ho = ga.Ho; a.Wo = ga.Wo; a.Cout = ga.job[ji].Cout;
    a.ws = ga.job[ji].ws; a.KoutP = ga.job[ji].KoutP; a.KinVP = ga.job[ji].KinVP; a.cinp32 = ga.job[ji].cinp32;
    a.dbp = ga.job[ji].dbp;
    a.tiles_x = ga.tiles_x; a.tiles_y = ga.tiles_y; a.tiles_total = ga.tiles_total;
    a.tiles_per_split = ga.tiles_per_split; a.nsplits = ga.nsplits; a.reflect = ga.reflect;
    const int local = (int)blockIdx.y - ga.job[ji].job_begin;
    const int ncib = ga.job[ji].ncib;
    const int cob = local / ncib, cib = local - cob * ncib;

    // tile list of this wave (wave-uniform scalars)
    // t = wave + 4 j  ->  (tap, aa, bb) = (t / AB, (t % AB) / B_T, (t % AB) % B_T).  With A_T == 2 the
    // block count AB is 2 or 4, so t % AB == wave % AB: every tile of a wave uses the SAME cout half
    // aa_w, and the wave reads a single A fragment per k-step.
    static_assert(A_T == 1 || (WPG % AB) == 0, "A_T == 2 needs AB to divide the waves of a group");
    const int aa_w = (A_T == 1) ? 0 : (wq % AB) / B_T;
    int t_ok[J], t_tap[J], t_boff[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int t = wq + WPG * j;
        t_ok[j] = t < T;
        const int tt = t_ok[j] ? t : 0;
        const int tap = tt / AB, ab = tt - tap * AB;
        const int aa = ab / B_T, bb = ab - aa * B_T;
        const int ty = tap / KH, tx = tap - ty * KH;
        t_tap[j] = tap * 1024 + aa * 32 + bb;  // packed for the store phase
        t_boff[j] = (ty * WT + tx) * CIB + bb * 32;  // slots beyond T alias tile 0: computed, never stored
    }

    f32x16 acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float bsum = 0.f;  // per-lane partial bias gradient: sum over pixels of g[p][aa_w*32 + li] (this lane's parity)
    const bool want_bias = (a.dbp != nullptr) && (cib == 0);

    const int t_begin = split * a.tiles_per_split;
    int t_end = t_begin + a.tiles_per_split;
    if (t_end > a.tiles_total) t_end = a.tiles_total;

    // ---- staging: g tile [PX][COB] then x halo tile [HT*WT][CIB], one float4 per item
    constexpr int G_ITEMS = PX * (COB / 4), G_IT = (G_ITEMS + 255) / 256;
    constexpr int X_ITEMS = HT * WT * (CIB / 4), X_IT = (X_ITEMS + 255) / 256;
    constexpr int N_IT = G_IT + X_IT;
    constexpr int BATCH = (PIPE || N_IT <= 12) ? N_IT : 8;
    constexpr int NBATCH = (N_IT + BATCH - 1) / BATCH;
    f32x4 rr[BATCH];
    // staging item k (a compile-time index at every call site) of batch `batch` of the pixel tile at (n, ty0, tx0) -> rr[k]
    struct TileAt { int n, ty0, tx0; };
    auto tile_at = [&](int tile) __attribute__((always_inline)) {
        int q = tile;
        const int tx = q % a.tiles_x;
        q /= a.tiles_x;
        const int ty = q % a.tiles_y;
        return TileAt{q / a.tiles_y, ty * THG, tx * TWG};
    };
    auto load_item = [&](const TileAt &ta, int batch, auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        const int n = ta.n, ty0 = ta.ty0, tx0 = ta.tx0;
        {
            const int it = batch * BATCH + k;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (it < G_IT) {
                const int i = tid + it * 256;
                const int p = i / (COB / 4), c4 = i - p * (COB / 4);
                const int r = p / TWG, c = p - r * TWG;
                const int oy = ty0 + r, ox = tx0 + c;
                const int co = cob * COB + c4 * 4;
                if (i < G_ITEMS && oy < a.Ho && ox < a.Wo && co < a.Cout) {
                    const int off = ((n * a.Ho + oy) * a.Wo + ox) * a.g_ct + a.g_co + co;
                    v = *reinterpret_cast<const f32x4 *>(a.g + off);
                }
            } else if (it < N_IT) {
                const int i = tid + (it - G_IT) * 256;
                const int pix = i / (CIB / 4), c4 = i - pix * (CIB / 4);
                const int hr = pix / WT, hc = pix - hr * WT;
                const int vb = cib * B_T + (c4 >> 3);  // global 32-wide virtual block index
                int Y, X, c;
                bool ok;
                if (S2D) {
                    const int nb32 = a.cinp32 >> 5;
                    const int pp = vb / nb32;
                    c = (vb - pp * nb32) * 32 + (c4 & 7) * 4;
                    Y = 2 * (ty0 + hr) - 1 + (pp >> 1);
                    X = 2 * (tx0 + hc) - 1 + (pp & 1);
                    ok = (pp < 4) & (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
                } else if (UP) {
                    c = vb * 32 + (c4 & 7) * 4;
                    Y = ty0 + hr - 1;
                    X = tx0 + hc - 1;
                    ok = (Y >= 0) & (Y < 2 * a.H) & (X >= 0) & (X < 2 * a.W);
                    Y >>= 1;
                    X >>= 1;
                } else {
                    c = vb * 32 + (c4 & 7) * 4;
                    Y = ty0 + hr - 1;
                    X = tx0 + hc - 1;
                    if (a.reflect) {
                        Y = Y < 0 ? -Y : (Y >= a.H ? 2 * a.H - 2 - Y : Y);
                        X = X < 0 ? -X : (X >= a.W ? 2 * a.W - 2 - X : X);
                    }
                    ok = (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
                }
                if (i < X_ITEMS && ok && c < a.Cin) {
                    const int off = ((n * a.H + Y) * a.W + X) * a.x_ct + a.x_co + c;
                    v = *reinterpret_cast<const f32x4 *>(a.x + off);
                }
            }
            rr[k] = v;
        }
    };
    auto load_batch = [&](int tile, int batch) __attribute__((always_inline)) {
        const TileAt ta = tile_at(tile);
        wg_static_for<0, BATCH>([&](auto kc) __attribute__((always_inline)) { load_item(ta, batch, kc); });
    };
    // PIPE: the loads of tile i+1 fly during the MFMA phase of tile i, issued as one burst in front of it.  Handing them out over
    // the first MFMA groups instead (what pays in conv_body.h: -3 % per dense-block chain) was measured here and LOSES: grouped
    // 32-cout jobs 104 -> 134 us, 192 -> 64: 491 -> 518 us, weight gradients 99 -> 116 ms per step (the staging index arithmetic
    // moves into the k-loop; the 256-register classes spill).  Kept behind -DTNR_WG_SPREAD_ALL for the record.
#if defined(TNR_WG_SPREAD_ALL) && !defined(TNR_NO_LOAD_SPREAD)
    constexpr bool SPREAD = PIPE;
#else
    constexpr bool SPREAD = false;
#endif
    constexpr int SPREAD_N = 3, IPG = (BATCH + SPREAD_N - 1) / SPREAD_N;
    auto load_group = [&](int tile, auto gc) __attribute__((always_inline)) {          // group g of SPREAD_N: items [g IPG, (g + 1) IPG) of batch 0
        constexpr int g = decltype(gc)::value;
        constexpr int k0 = g * IPG < BATCH ? g * IPG : BATCH, k1 = (g + 1) * IPG < BATCH ? (g + 1) * IPG : BATCH;
        const TileAt ta = tile_at(tile);
        wg_static_for<k0, k1>([&](auto kc) __attribute__((always_inline)) { load_item(ta, 0, kc); });
    };
    auto split4 = [&](const f32x4 v, wg_f32x2 (&out)[3]) {
        wg_bf16x4 h, m, l;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const __bf16 hh = (__bf16)v[i];
            const float r1 = v[i] - (float)hh;
            const __bf16 mm = (__bf16)r1;
            h[i] = hh;
            m[i] = mm;
            l[i] = (__bf16)(r1 - (float)mm);
        }
        out[0] = __builtin_bit_cast(wg_f32x2, h);
        out[1] = __builtin_bit_cast(wg_f32x2, m);
        out[2] = __builtin_bit_cast(wg_f32x2, l);
    };
    char *const s_gb = reinterpret_cast<char *>(smem);                          // (BF == 2) byte views of the two pre-split tiles
    char *const s_xb = s_gb + (size_t)PX * COB * 6;
    auto store_batch = [&](int batch) {
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int it = batch * BATCH + k;
            if constexpr (BF == 2) {
                // item i = (pixel, channel quad): three 8-byte pieces, one per plane (wg_x3_off)
                wg_f32x2 pc[3];
                if (it < G_IT) {
                    const int i = tid + it * 256;
                    if (i < G_ITEMS) {
                        split4(rr[k], pc);
                        char *d = s_gb + wg_x3_off(i / (COB / 4), (i % (COB / 4)) * 4, COB);
#pragma unroll
                        for (int sp = 0; sp < 3; ++sp) *reinterpret_cast<wg_f32x2 *>(d + 256 * sp) = pc[sp];
                    }
                } else if (it < N_IT) {
                    const int i = tid + (it - G_IT) * 256;
                    if (i < X_ITEMS) {
                        split4(rr[k], pc);
                        char *d = s_xb + wg_x3_off(i / (CIB / 4), (i % (CIB / 4)) * 4, CIB);
#pragma unroll
                        for (int sp = 0; sp < 3; ++sp) *reinterpret_cast<wg_f32x2 *>(d + 256 * sp) = pc[sp];
                    }
                }
                continue;
            }
            if (it < G_IT) {
                const int i = tid + it * 256;
                if (i < G_ITEMS) *reinterpret_cast<f32x4 *>(s_g + i * 4) = rr[k];
            } else if (it < N_IT) {
                const int i = tid + (it - G_IT) * 256;
                if (i < X_ITEMS) *reinterpret_cast<f32x4 *>(s_x + i * 4) = rr[k];
            }
        }
    };

    if constexpr (DB) {
        // ============================ the pipelined form (TNR_MMA_BF16X3, zero-padded 3x3) ============================
        // What the two-workgroup form still loses (profiles/r03o): a workgroup's refill -- global loads, the operand split, LDS
        // stores, two barriers -- is hidden only by the OTHER workgroup's MFMAs, which the matrix core does not interleave with its
        // own (tools/probes/mfma_chain.hip), and each fragment's six transposing reads sit in a block in front of a pair of tiles.
        // Here: one wave per SIMD and, as in conv_sweep4_kernel, everything that is not an MFMA is handed out BEHIND the MFMAs, a
        // few instructions each (the wave issues them while the MFMA it just issued executes):
        //   * LDS holds two tile sets; while the MFMAs of pixel tile i read set i & 1, the float4 items of tile i + 1 are loaded
        //     (one item behind every second MFMA of the first rows), split and stored into the other set (five steps per item in
        //     the second half of the phase) -- one barrier per tile;
        //   * the x halo tile is stored with rows of 20 pixels (5 blocks of 4): a tile row is a CONSTANT address step, so every
        //     fragment read is base register + immediate and the phase is straight-line code with no address arithmetic;
        //   * the reads of the next pair of tiles' fragments (12) and of the next row's gradient fragment (6) are spread over
        //     the current pair's MFMAs.
        static_assert(BF == 2 && MODE == TNR_CONV_3x3 && WPS == 1, "pipelined form: bf16x3, plain 3x3, one workgroup per CU");
        constexpr int WTP = 20;                                       // padded halo row (pixels) in LDS
        constexpr int G_BYTES = PX * COB * 6, X_BYTES = HT * WTP * CIB * 6, TILES_BYTES = G_BYTES + X_BYTES;
        constexpr int SET_BYTES = TILES_BYTES + 3072;               // + a dump area for the staging slots beyond the last item (no lane masks on the stores)
        constexpr int XBLK = (CIB / 32) * 768, GBLK = (COB / 32) * 768;   // bytes per block of 4 pixels
        constexpr int NP = (J + 1) / 2, MPR = 6 * J, NM = ROWS * MPR;     // pairs per row, MFMAs per row / per tile
        constexpr int G0 = NM - 5 * N_IT - 6;                             // first MFMA that carries an item step
        static_assert(G0 >= 2 * N_IT, "the phase is too short for the staging plan");
        char *const lds = reinterpret_cast<char *>(smem);
        const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.g), 0,
                                                                              (int)((unsigned)a.N * a.Ho * a.Wo * a.g_ct * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0,
                                                                              (int)((unsigned)a.N * a.H * a.W * a.x_ct * 4u), 0x00020000);
        // ---- staging plan, once per kernel: item k of this thread -> element offset from the tile origin (or invalid), its position
        // in the tile (border test) and its byte address in a tile set
        int it_src[N_IT], it_yx[N_IT], it_dst[N_IT];
#pragma unroll
        for (int k = 0; k < N_IT; ++k) {
            if (k < G_IT) {
                const int i = tid + k * 256;
                const int pp = i / (COB / 4), c4 = i - pp * (COB / 4);
                const int r = pp / TWG, cc = pp - r * TWG;
                const int co = cob * COB + c4 * 4;
                const bool v = i < G_ITEMS && co < a.Cout;
                it_src[k] = v ? (r * a.Wo + cc) * a.g_ct + a.g_co + co : -1;
                it_yx[k] = r | (cc << 8);
                it_dst[k] = i < G_ITEMS ? wg_x3_off(pp, c4 * 4, COB) : TILES_BYTES + tid * 8;
            } else {
                const int i = tid + (k - G_IT) * 256;
                const int pix = i / (CIB / 4), c4 = i - pix * (CIB / 4);
                const int hr = pix / WT, hc = pix - hr * WT;
                const int ch = (cib * B_T + (c4 >> 3)) * 32 + (c4 & 7) * 4;
                const bool v = i < X_ITEMS && ch < a.Cin;
                it_src[k] = v ? ((hr - 1) * a.W + (hc - 1)) * a.x_ct + a.x_co + ch : (1 << 30);
                it_yx[k] = hr | (hc << 8);
                it_dst[k] = i < X_ITEMS ? G_BYTES + wg_x3_off(hr * WTP + hc, c4 * 4, CIB) : TILES_BYTES + tid * 8;
            }
        }
        int ld_ty0 = 0, ld_tx0 = 0, ld_gbase = 0, ld_xbase = 0;      // the tile being loaded (wave-uniform)
        bool ld_valid = true;       // false behind the last tile: the loads and stores still run (straight-line code: a branch around them makes
                                    // the compiler wait for ALL pending loads in front of every load), on zeros
        // (the tiles of a split are consecutive: the coordinates advance by counting -- tile_at's two divisions are ~70 scalar instructions
        //  at the top of every tile, where nothing hides them)
        int l_n = 0, l_ty = 0, l_tx = 0;
        auto load_setup_first = [&](int tile) __attribute__((always_inline)) {
            const TileAt ta = tile_at(tile);
            l_n = ta.n; l_ty = ta.ty0 / THG; l_tx = ta.tx0 / TWG;
        };
        auto load_setup_next = [&]() __attribute__((always_inline)) {
            if (++l_tx == a.tiles_x) {
                l_tx = 0;
                if (++l_ty == a.tiles_y) {
                    l_ty = 0;
                    ++l_n;
                }
            }
        };
        auto load_setup = [&]() __attribute__((always_inline)) {
            ld_ty0 = l_ty * THG; ld_tx0 = l_tx * TWG;
            ld_gbase = ((l_n * a.Ho + ld_ty0) * a.Wo + ld_tx0) * a.g_ct;
            ld_xbase = ((l_n * a.H + ld_ty0) * a.W + ld_tx0) * a.x_ct;
        };
        auto item_load = [&](auto kc) __attribute__((always_inline)) {       // (an offset past the end reads zeros: borders and padding)
            constexpr int k = decltype(kc)::value;
            const int y = it_yx[k] & 255, xx = it_yx[k] >> 8;
            if constexpr (k < G_IT) {
                const bool ok = ld_valid & (it_src[k] >= 0) & (ld_ty0 + y < a.Ho) & (ld_tx0 + xx < a.Wo);
                const unsigned bo = ok ? (unsigned)(ld_gbase + it_src[k]) * 4u : 0xfffffff0u;
                rr[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rs, (int)bo, 0, 0));
            } else {
                const int Y = ld_ty0 + y - 1, X = ld_tx0 + xx - 1;
                const bool ok = ld_valid & (it_src[k] != (1 << 30)) & ((unsigned)Y < (unsigned)a.H) & ((unsigned)X < (unsigned)a.W);
                const unsigned bo = ok ? (unsigned)(ld_xbase + it_src[k]) * 4u : 0xfffffff0u;
                rr[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (int)bo, 0, 0));
            }
        };
        // item k in five steps, each over the item's four channels at once: 0 hi plane + first residual, 1 mid plane + second
        // residual, 2 lo plane + store hi, 3 store mid, 4 store lo (split4's arithmetic)
        wg_bf16x4 ih, im, il;
        f32x4 ir;
        auto item_step = [&](auto kc, auto sc, char *set) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value, st = decltype(sc)::value;
            if constexpr (st == 0) {
#if WG_PK_SPLIT
                wg_pk_level(rr[k], ih, ir);
#else
                ih = __builtin_convertvector(rr[k], wg_bf16x4);
                ir = rr[k] - __builtin_convertvector(ih, f32x4);
#endif
            } else if constexpr (st == 1) {
#if WG_PK_SPLIT
                const f32x4 r1 = ir;
                wg_pk_level(r1, im, ir);
#else
                im = __builtin_convertvector(ir, wg_bf16x4);
                ir = ir - __builtin_convertvector(im, f32x4);
#endif
            } else {
#if WG_PK_SPLIT
                if constexpr (st == 2) {
                    const wg_f32x2 a = {ir[0], ir[1]}, b = {ir[2], ir[3]};
                    il = __builtin_bit_cast(wg_bf16x4, wg_f32x2{__builtin_bit_cast(float, __builtin_convertvector(a, wg_bf16x2)),
                                                                __builtin_bit_cast(float, __builtin_convertvector(b, wg_bf16x2))});
                }
#else
                if constexpr (st == 2) il = __builtin_convertvector(ir, wg_bf16x4);
#endif
                *reinterpret_cast<wg_f32x2 *>(set + it_dst[k] + 256 * (st - 2)) = __builtin_bit_cast(wg_f32x2, st == 2 ? ih : (st == 3 ? im : il));
            }
        };
        // ---- fragment addresses, once per kernel: lane roles of the transposing read as in the two-workgroup form
        const int cg = (lane >> 4) & 1, m4 = (lane >> 2) & 3, q4 = lane & 3;
        const int lch = 16 * cg + 4 * q4;
        const int ga_off = ((pg * ROWS * 4 + 2 * half) * (COB / 32) + aa_w) * 768 + m4 * 64 + lch * 2;      // row r: + r * 4 * GBLK
        int xb_off[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int tap = t_tap[j] >> 10, bb = t_tap[j] & 31;
            const int ty = tap / KH, tx = tap - ty * KH;
            const int q0 = tx + 8 * half + m4;
            xb_off[j] = G_BYTES + ((pg * ROWS + ty) * (WTP / 4) + (q0 >> 2)) * XBLK + (q0 & 3) * 64 + bb * 768 + lch * 2;   // row r: + r * 5 * XBLK
        }
        constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
        // one transposing read = 4 pixels of a plane: half `hh` (0: pixels 0 .. 3 of the lane's 8, 1: pixels 4 .. 7) of plane sp
        auto rd = [&](const char *base, int imm) __attribute__((always_inline)) {
            return __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_s16x4 *)(base + imm));
        };
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        // raw halves.  B fragments are CACHED per halo row: tap (ty, tx) of tile row r reads halo row r + ty, which tap (ty - 1, tx) of row
        // r + 1 reads again -- a tile row brings in ONE new halo row (3 fragments, one per tx) instead of 9 fragments (-45 % transposing
        // reads at 4 rows, the kernel's first limiter: profiles/r03y_wgrad_ablation.txt).  [halo row mod 3][tx][plane][half]
        static_assert(J == 9 && WPG == AB, "every wave owns the 9 taps of one (cout half, cin block) pair: tile j = tap j");
        wg_s16x4 ra[2][3][2], rbx[3][3][3][2];
        auto pack = [&](const wg_s16x4 lo4, const wg_s16x4 hi4) __attribute__((always_inline)) {
            const s16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            return __builtin_bit_cast(wg_bf16x8, v);
        };

        if (t_begin < t_end) {
            // ---- prologue: tile t_begin into set 0
            load_setup_first(t_begin);
            load_setup();
            wg_static_for<0, N_IT>([&](auto kc) __attribute__((always_inline)) { item_load(kc); });
            wg_static_for<0, N_IT>([&](auto kc) __attribute__((always_inline)) {
                wg_static_for<0, 5>([&](auto sc) __attribute__((always_inline)) { item_step(kc, sc, lds); });
            });
            __syncthreads();
        }
        for (int tile = t_begin; tile < t_end; ++tile) {
            const int cur = (tile - t_begin) & 1;
            const char *set = lds + cur * SET_BYTES;
            char *oset = lds + (cur ^ 1) * SET_BYTES;
            ld_valid = tile + 1 < t_end;
            if (ld_valid) load_setup_next();
            load_setup();
            const char *gab = set + ga_off;
            const char *xbb[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) xbb[j] = set + xb_off[j];
            // read id -> one transposing read.  A fragment of row r: ids 0 .. 5 = (plane, half); B fragment of tile j, row r likewise
            auto read_a = [&](auto rc, auto idc) __attribute__((always_inline)) {
                constexpr int r = decltype(rc)::value, id = decltype(idc)::value, sp = id >> 1, hh = id & 1;
                ra[r & 1][sp][hh] = rd(gab, r * 4 * GBLK + 256 * sp + hh * GBLK);
            };
            auto read_x = [&](auto xrc, auto txc, auto idc) __attribute__((always_inline)) {        // halo row xr (relative to the pixel group), tap column tx
                constexpr int xr = decltype(xrc)::value, tx = decltype(txc)::value, id = decltype(idc)::value, sp = id >> 1, hh = id & 1;
                rbx[xr % 3][tx][sp][hh] = rd(xbb[tx], xr * 5 * XBLK + 256 * sp + hh * XBLK);
            };
            // (the fragments of row 0, pair 0 and its gradient fragment were read behind the last MFMAs of the previous tile; see below)
            if (tile == t_begin) {
                wg_static_for<0, 6>([&](auto idc) __attribute__((always_inline)) { read_a(std::integral_constant<int, 0>{}, idc); });
                wg_static_for<0, 6>([&](auto idc) __attribute__((always_inline)) { read_x(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, idc); });
                wg_static_for<0, 6>([&](auto idc) __attribute__((always_inline)) { read_x(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, idc); });
            }
            // the same 18 reads for the NEXT tile, out of the other set (complete behind this tile's barrier)
            const char *gab_n = oset + ga_off;
            const char *xbb_n[2] = {oset + xb_off[0], oset + xb_off[1]};
            auto read_next = [&](auto idc) __attribute__((always_inline)) {
                constexpr int id = decltype(idc)::value;
                if constexpr (id < 6) {
                    constexpr int sp = id >> 1, hh = id & 1;
                    ra[0][sp][hh] = rd(gab_n, 256 * sp + hh * GBLK);
                } else {
                    constexpr int tx = (id - 6) / 6, k = (id - 6) % 6, sp = k >> 1, hh = k & 1;
                    rbx[0][tx][sp][hh] = rd(xbb_n[tx], 256 * sp + hh * XBLK);
                }
            };
            __builtin_amdgcn_sched_barrier(0);
            wg_static_for<0, ROWS>([&](auto rc) __attribute__((always_inline)) {
                constexpr int r = decltype(rc)::value;
                wg_static_for<0, NP>([&](auto jpc) __attribute__((always_inline)) {
                    constexpr int jp = decltype(jpc)::value;
                    constexpr int j0 = 2 * jp, nt = (j0 + 1 < J) ? 2 : 1, nm = 6 * nt;        // tiles / MFMAs of this pair
                    constexpr int gbase = r * MPR + jp * 12;
                    // reads spread over this pair's MFMAs.  Row 0 fills the cache pair by pair (the fragments of pair jp + 1, front-loaded: the
                    // last read is issued >= 4 MFMAs before its pair starts; the gradient fragment of row 1 behind pair 3); a row r >= 1
                    // brings in halo row r + 2 and the gradient fragment of row r + 1, one read per MFMA of its pairs 0 and 1 (first
                    // needed by pair 3)
                    constexpr int nj0 = 2 * (jp + 1), nnt = (nj0 + 1 < J) ? 2 : 1;
                    constexpr int NRD = (r == 0 && jp + 1 < NP) ? (6 * nnt + ((jp + 2 == NP && ROWS > 1) ? 6 : 0)) : 0;
                    constexpr int RDEN = nm > 5 ? nm - 4 : 2, RPH = (NRD + RDEN - 1) / RDEN;   // reads per MFMA (row 0)
                    // operands of this pair
                    wg_bf16x8 fa[3], fbv[2][3];
#pragma unroll
                    for (int sp = 0; sp < 3; ++sp) {
                        fa[sp] = pack(ra[r & 1][sp][0], ra[r & 1][sp][1]);
#pragma unroll
                        for (int q = 0; q < nt; ++q) fbv[q][sp] = pack(rbx[(r + (j0 + q) / 3) % 3][(j0 + q) % 3][sp][0], rbx[(r + (j0 + q) / 3) % 3][(j0 + q) % 3][sp][1]);
                    }
                    // The tile's barrier stands in front of the LAST pair (6 MFMAs; its operands are in registers, every store into the other
                    // set was issued at least 7 MFMAs ago, and nothing reads this set any more): the first fragments of the next tile
                    // are read behind those 6 MFMAs instead of in front of an idle matrix core at the top of the next tile.
                    static_assert(((ROWS - 1) & 1) == 1, "the next tile's gradient fragment goes into register set 0");
                    if constexpr (r + 1 == ROWS && jp + 1 == NP) __syncthreads();
                    if constexpr (jp == 0) {
                        if (want_bias) {           // hi + mid + lo reconstructs the fp32 value exactly
#pragma unroll
                            for (int kk = 0; kk < 8; ++kk) bsum += ((float)fa[0][kk] + (float)fa[1][kk]) + (float)fa[2][kk];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    wg_static_for<0, nm>([&](auto ic) __attribute__((always_inline)) {
                        constexpr int i = decltype(ic)::value, pq = nt == 2 ? i / 2 : i, q = nt == 2 ? i % 2 : 0;
                        constexpr int g = gbase + i;
                        acc[j0 + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[TA[pq]], fbv[q][TB[pq]], acc[j0 + q], 0, 0, 0);
                        if constexpr (r + 1 == ROWS && jp + 1 == NP) {
                            wg_static_for<0, 3>([&](auto kc) __attribute__((always_inline)) { read_next(std::integral_constant<int, 3 * i + decltype(kc)::value>{}); });
                        }
                        if constexpr (r == 0) {
                            wg_static_for<0, RPH>([&](auto kc) __attribute__((always_inline)) {
                                constexpr int id = i * RPH + decltype(kc)::value;
                                if constexpr (id < NRD) {
                                    if constexpr (id < 6 * nnt) {
                                        constexpr int jn = nj0 + id / 6;
                                        read_x(std::integral_constant<int, jn / 3>{}, std::integral_constant<int, jn % 3>{}, std::integral_constant<int, id % 6>{});
                                    } else {
                                        read_a(std::integral_constant<int, 1>{}, std::integral_constant<int, id - 6 * nnt>{});
                                    }
                                }
                            });
                        } else {
                            constexpr int m = jp * 12 + i;             // MFMA index inside the row
                            if constexpr (m < 18) read_x(std::integral_constant<int, r + 2>{}, std::integral_constant<int, (m < 18 ? m / 6 : 0)>{}, std::integral_constant<int, m % 6>{});
                            else if constexpr (m < 24 && r + 1 < ROWS) read_a(std::integral_constant<int, (r + 1 < ROWS ? r + 1 : r)>{}, std::integral_constant<int, (m >= 18 && m < 24 ? m - 18 : 0)>{});
                        }
                        // the next pixel tile: loads behind every second MFMA from the start, item steps in the tail of the phase
#ifndef WG_ABL_NOITEMS       /* (ablation builds: timing only, results invalid) */
                        if constexpr ((g & 1) == 0 && g / 2 < N_IT) {
                            item_load(std::integral_constant<int, (g / 2 < N_IT ? g / 2 : 0)>{});
                        }
                        if constexpr (g >= G0 && g < G0 + 5 * N_IT) {
                            item_step(std::integral_constant<int, (g >= G0 && g < G0 + 5 * N_IT ? (g - G0) / 5 : 0)>{},
                                      std::integral_constant<int, (g >= G0 ? (g - G0) % 5 : 0)>{}, oset);
                        }
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
            });
        }
        __syncthreads();              // (the pixel-group exchange below reuses the tile sets)
    } else {
    if (PIPE && t_begin < t_end) load_batch(t_begin, 0);
        for (int tile = t_begin; tile < t_end; ++tile) {
            if (PIPE) {
                __syncthreads();  // previous tile's fragments are consumed
                store_batch(0);
                __syncthreads();
                if (!SPREAD && tile + 1 < t_end) load_batch(tile + 1, 0);  // in flight during the MFMA phase below
            } else {
                load_batch(tile, 0);
                __syncthreads();
                store_batch(0);
    #pragma unroll
                for (int bt = 1; bt < NBATCH; ++bt) {
                    load_batch(tile, bt);
                    store_batch(bt);
                }
                __syncthreads();
            }
            // ---- K loop: two pixels per MFMA, software-pipelined one k-step deep: the fragments of step
            // k+1 (one A value, J B values per lane) are read into the second register set BEFORE the J MFMAs
            // of step k are issued, so the LDS latency hides under >= J*64 matrix-core cycles even with a
            // single wave per SIMD.  A tile row (16 pixels = 8 k-steps) is fully unrolled: every LDS address
            // is row base + compile-time offset, including the first step of the next row.
            {
                // integer float-offsets into smem[] (keeps the accesses provably LDS: ds_read with immediates)
                int go = half * COB + li + aa_w * 32 + pg * ROWS * TWG * COB;
                int xo[J];
    #pragma unroll
                for (int j = 0; j < J; ++j) xo[j] = PX * COB + half * CIB + li + t_boff[j] + pg * ROWS * WT * CIB;
                if constexpr (BF == 2) {
                    // TNR_MMA_BF16X3: both tiles are pre-split in LDS (wg_x3_off); a fragment = 6 transposing reads (2 x 4 pixels x 3
                    // planes), six MFMAs per output tile and tile row.  The J tiles of a wave are taken two at a time (two independent
                    // accumulator chains); the fragments of the next pair are read while the current pair's 12 MFMAs run.
                    constexpr int NP = (J + 1) / 2;
                    constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
                    // lane roles of the transposing read: 16-lane group -> (pixel half h = lane >> 5, channel half cg); inside the group
                    // lane 4 m + q supplies pixel m, channels 4 q .. 4 q + 3
                    const int cg = (lane >> 4) & 1, m4 = (lane >> 2) & 3, q4 = lane & 3;
                    const int lch = 16 * cg + 4 * q4;                                    // channel inside the 32-channel block
                    auto frag = [&](const char *base, wg_bf16x8 (&out)[3], int rd_stride) {
    #pragma unroll
                        for (int sp = 0; sp < 3; ++sp) {
                            const wg_s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_s16x4 *)(base + 256 * sp));
                            const wg_s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_s16x4 *)(base + 256 * sp + rd_stride));
                            typedef short s16x8 __attribute__((ext_vector_type(8)));
                            const s16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                            out[sp] = __builtin_bit_cast(wg_bf16x8, v);
                        }
                    };
                    // A (gradient tile, 16 pixels per row: the pixel phase is the lane's m): constant per-lane offset + a row stride
                    const char *ga_ptr = s_gb + ((pg * ROWS * 4 + 2 * half) * (COB / 32) + aa_w) * 768 + m4 * 64 + lch * 2;
                    constexpr int GA_ROW = 4 * (COB / 32) * 768, GA_RD = (COB / 32) * 768, XB_RD = (CIB / 32) * 768;
                    // B (input halo tile): pixel = (row + ty) * WT + tx + 8 half + m (+ 4 for the second read)
                    // wave-uniform part (SGPRs) + one per-lane term: 2 address registers instead of 2 J
                    int xpix[J], xblk[J];
                    const int lpix = 8 * half + m4, lblk = lch * 2;
    #pragma unroll
                    for (int j = 0; j < J; ++j) {
                        const int tap = t_tap[j] >> 10, bb = t_tap[j] & 31;
                        const int ty = tap / KH, tx = tap - ty * KH;
                        xpix[j] = __builtin_amdgcn_readfirstlane((pg * ROWS + ty) * WT + tx);
                        xblk[j] = __builtin_amdgcn_readfirstlane(bb * 768);
                    }
                    auto xaddr = [&](int j) {
                        const int P = xpix[j] + lpix;
                        return s_xb + (P >> 2) * ((CIB / 32) * 768) + (P & 3) * 64 + xblk[j] + lblk;
                    };
                    // (two workgroups per CU: the other workgroup's wave covers the LDS latency, one fragment set is enough -- 24 registers)
                    constexpr int NSET = WPS == 1 ? 2 : 1;
                    wg_bf16x8 ca[3], cb[NSET][2][3];
                    auto read_pair = [&](int jp, int set) {
    #pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int j = 2 * jp + q < J ? 2 * jp + q : J - 1;
                            frag(xaddr(j), cb[set][q], XB_RD);
                        }
                    };
    #pragma unroll 1
                    for (int r = 0; r < ROWS; ++r) {
                        frag(ga_ptr, ca, GA_RD);
                        if (NSET == 2) read_pair(0, 0);
                        if (want_bias) {           // hi + mid + lo reconstructs the fp32 value exactly
    #pragma unroll
                            for (int kk = 0; kk < 8; ++kk) bsum += ((float)ca[0][kk] + (float)ca[1][kk]) + (float)ca[2][kk];
                        }
    #pragma unroll
                        for (int jp = 0; jp < NP; ++jp) {
                            if (NSET == 1) read_pair(jp, 0);
                            else if (jp + 1 < NP) read_pair(jp + 1, (jp + 1) & 1);
                            __builtin_amdgcn_sched_barrier(0);
                            const int cs = NSET == 1 ? 0 : (jp & 1);
    #pragma unroll
                            for (int p = 0; p < 6; ++p) {
                                acc[2 * jp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[TA[p]], cb[cs][0][TB[p]], acc[2 * jp], 0, 0, 0);
                                if (2 * jp + 1 < J)
                                    acc[2 * jp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[TA[p]], cb[cs][1][TB[p]], acc[2 * jp + 1], 0, 0, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        ga_ptr += GA_ROW;
    #pragma unroll
                        for (int j = 0; j < J; ++j) xpix[j] += WT;
                    }
                } else if constexpr (BF) {
                    float ra[8], rb[8][J];
                    wg_bf16x8 ca, cb[J];
                    auto read_row = [&]() {
    #pragma unroll
                        for (int kk = 0; kk < 8; ++kk) {
                            ra[kk] = smem[go + 2 * kk * COB];
    #pragma unroll
                            for (int j = 0; j < J; ++j) rb[kk][j] = smem[xo[j] + 2 * kk * CIB];
                        }
                    };
                    auto pack_row = [&]() {
    #pragma unroll
                        for (int kk = 0; kk < 8; ++kk) {
                            bsum += ra[kk];
                            ca[kk] = (__bf16)ra[kk];
    #pragma unroll
                            for (int j = 0; j < J; ++j) cb[j][kk] = (__bf16)rb[kk][j];
                        }
                    };
                    read_row();
                    pack_row();
    #pragma unroll 1
                    for (int r = 0; r < ROWS; ++r) {
                        go += TWG * COB;
    #pragma unroll
                        for (int j = 0; j < J; ++j) xo[j] += WT * CIB;
                        if (r + 1 < ROWS) read_row();
                        if constexpr (SPREAD) {
                            if (tile + 1 < t_end) {
                                if constexpr (ROWS >= SPREAD_N) {
                                    if (r == 0) load_group(tile + 1, std::integral_constant<int, 0>{});
                                    if (r == 1) load_group(tile + 1, std::integral_constant<int, 1>{});
                                    if (r == 2) load_group(tile + 1, std::integral_constant<int, 2>{});
                                } else if (r == 0) {
                                    load_batch(tile + 1, 0);
                                }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                        for (int j = 0; j < J; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca, cb[j], acc[j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (r + 1 < ROWS) pack_row();
                    }
                } else {
                float fa[2], fb[2][J];
                fa[0] = smem[go];
    #pragma unroll
                for (int j = 0; j < J; ++j) fb[0][j] = smem[xo[j]];
    #pragma unroll 1
                for (int r = 0; r < ROWS; ++r) {
    #pragma unroll
                    for (int k = 0; k < TWG / 2; ++k) {
                        const int cur = k & 1, nxt = cur ^ 1;
                        // step k+1 of this row, or step 0 of the next row (row THG is valid LDS: never consumed)
                        const int goff = (k + 1 < TWG / 2) ? 2 * (k + 1) * COB : TWG * COB;
                        const int xoff = (k + 1 < TWG / 2) ? 2 * (k + 1) * CIB : WT * CIB;
                        fa[nxt] = smem[go + goff];
    #pragma unroll
                        for (int j = 0; j < J; ++j) fb[nxt][j] = smem[xo[j] + xoff];
                        if constexpr (SPREAD) {
                            if (r == 0 && tile + 1 < t_end) {
                                if (k == 0) load_group(tile + 1, std::integral_constant<int, 0>{});
                                if (k == 1) load_group(tile + 1, std::integral_constant<int, 1>{});
                                if (k == 2) load_group(tile + 1, std::integral_constant<int, 2>{});
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        bsum += fa[cur];
    #pragma unroll
                        for (int j = 0; j < J; ++j)
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur], fb[cur][j], acc[j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    go += TWG * COB;
    #pragma unroll
                    for (int j = 0; j < J; ++j) xo[j] += WT * CIB;
                }
                }
            }
        }
    
    }

    // ---- pixel groups: group p = 1 .. KS-1 hands its accumulators (and bias sums) to group 0 through LDS, in order
    float btot = bsum + __shfl_xor(bsum, 32);
    if constexpr (KS > 1) {
        constexpr int ACC_FLOATS = WPG * J * 16 * 64;
        // (ACC_FLOATS + WPG * 64 floats: launch_wgrad_t sizes the dynamic LDS for the larger of the tile image and this exchange)
#pragma unroll 1
        for (int p = 1; p < KS; ++p) {
            __syncthreads();                       // the tiles (p == 1) / the previous round's values are consumed
            if (pg == p) {
#pragma unroll
                for (int j = 0; j < J; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) smem[((wq * J + j) * 16 + r) * 64 + lane] = acc[j][r];
                smem[ACC_FLOATS + wq * 64 + lane] = btot;
            }
            __syncthreads();
            if (pg == 0) {
#pragma unroll
                for (int j = 0; j < J; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] += smem[((wq * J + j) * 16 + r) * 64 + lane];
                btot += smem[ACC_FLOATS + wq * 64 + lane];
            }
        }
        if (pg != 0) return;                       // (no barrier follows)
    }

    // ---- write the partial slab [tap][co][blk][split][32]: the split axis is contiguous (128-B granules), so the
    // reducer streams each (tap, co, blk) row; one base pointer per tile, a constant stride per output row
    {
        const int nblk = a.KinVP >> 5;
        const int nsl = a.nsplits, sl = split;
        const size_t row_stride = (size_t)nblk * nsl * 32;             // floats between consecutive co
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int tap = t_tap[j] >> 10, aa = (t_tap[j] >> 5) & 31, bb = t_tap[j] & 31;
            const int blk = cib * B_T + bb;                // 32-wide virtual input-channel block
            const int co0 = cob * COB + aa * 32;           // KoutP % 32 == 0: a 32-cout block is valid as a whole
            if (!t_ok[j] || blk >= nblk || co0 >= a.KoutP) continue;
            float *p = a.ws + (((size_t)tap * a.KoutP + co0 + 4 * half) * nblk + blk) * nsl * 32 + (size_t)sl * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) p[(size_t)((r & 3) + 8 * (r >> 2)) * row_stride] = acc[j][r];
        }
    }
    // every wave accumulated the sums of its cout half; the first wave of each half (bb == 0) publishes
    if (want_bias && wq < AB && (wq % B_T) == 0) {
        const int co = cob * COB + aa_w * 32 + li;
        if (half == 0 && co < a.KoutP) a.dbp[(size_t)co * a.nsplits + split] = btot;   // [co][split]
    }
}

struct RedJob {
    const float *ws; const float *dbp; float *dw; float *db;
    float *dw2; float *db2;       // cout pair (tnr_wgrad_desc.cout_split): output channels >= split belong to a second layer
    int split, cin_total2;        // split = Cout when the job is a single layer
    int KoutP, KinVP, cinp32, Cout, Cin, cin_total, cin_begin;
    int blk_begin, nrows;         // blockIdx.x range [blk_begin, blk_begin + nrows + bias blocks)
    float alpha, beta;
};
struct RedK {
    int splits, ntaps, kh, kw, s2d, njobs;
    int rpb;                      // slab rows per block: 1 (8 split lanes per row, many splits) or 8 (one thread per row element, <= 32 splits)
    RedJob job[TNR_WGRAD_GROUP_MAX];
};

// One block per slab row (tap, co, 32-channel block): 256 threads = 32 channels x 8 split lanes.  The
// row's partials [split][32] are contiguous, so the 8 lanes stream 1 KiB per step; lane sums are combined
// in a fixed order through LDS (deterministic).
// Grouped launches have few splits per job (one chip-filling wave of workgroups over all their jobs: 14 for an RRDB's 18 jobs): there a
// block takes EIGHT rows, one thread per (row, channel) walking all splits with eight independent partial sums in a fixed order -- no
// LDS, no barrier, 1/8 of the blocks (rpb = 8; a block per row spent its time on the prologue and the barrier: 32 us per launch).
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const RedK ga) {
    __shared__ float sh[8][33];
    int ji = 0;
#pragma unroll
    for (int q = 1; q < TNR_WGRAD_GROUP_MAX; ++q)
        if (q < ga.njobs && (int)blockIdx.x >= ga.job[q].blk_begin) ji = q;
    const RedJob &a = ga.job[ji];
    const int bx = (int)blockIdx.x - a.blk_begin;
    const int splits = ga.splits;
    const int nblk = a.KinVP >> 5;
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int nrows = a.nrows;
    const int nrowblk = (nrows + ga.rpb - 1) / ga.rpb;
    if (bx >= nrowblk) {
        // bias blocks: 32 output channels each, the 8 split lanes stream dbp[co][split]
        const int c = (bx - nrowblk) * 32 + el;
        float part = 0.f;
        if (c < a.Cout)
            for (int q = sl; q < splits; q += 8) part += a.dbp[(size_t)c * splits + q];
        sh[sl][el] = part;
        __syncthreads();
        if (sl == 0 && c < a.Cout) {
            float sum = 0.f;
#pragma unroll
            for (int l = 0; l < 8; ++l) sum += sh[l][el];
            float *dbc = c < a.split ? a.db + c : a.db2 + (c - a.split);
            const float prev = (a.beta != 0.f) ? a.beta * *dbc : 0.f;
            *dbc = prev + a.alpha * sum;
        }
        return;
    }
    const int row = ga.rpb == 8 ? bx * 8 + sl : bx;     // ((tap * KoutP) + co) * nblk + blk
    if (row >= nrows) return;                           // (rpb = 8: the last block's spare rows; no barrier on that path)
    const int blk = row % nblk;
    const int co = (row / nblk) % a.KoutP;
    const int tap = row / (nblk * a.KoutP);
    const float *src = a.ws + (size_t)row * splits * 32 + el;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    float sum_all = 0.f;
    if (ga.rpb == 8) {
        float q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int s8 = 0;
        for (; s8 + 8 <= splits; s8 += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) q[k] += src[(size_t)(s8 + k) * 32];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (s8 + k < splits) q[k] += src[(size_t)(s8 + k) * 32];
        sum_all = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
    }
    int s = ga.rpb == 8 ? splits : sl;
    for (; s + 24 < splits; s += 32) {                  // four independent loads in flight per lane
        p0 += src[(size_t)s * 32];
        p1 += src[(size_t)(s + 8) * 32];
        p2 += src[(size_t)(s + 16) * 32];
        p3 += src[(size_t)(s + 24) * 32];
    }
    for (; s < splits; s += 8) p0 += src[(size_t)s * 32];
    if (ga.rpb != 8) {
        sh[sl][el] = (p0 + p1) + (p2 + p3);
        __syncthreads();
    }
    if (sl == 0 || ga.rpb == 8) {
        float sum = sum_all;
        if (ga.rpb != 8) {
#pragma unroll
            for (int l = 0; l < 8; ++l) sum += sh[l][el];
        }
        const int civ = blk * 32 + el;
        int ci, ky, kx;
        if (ga.s2d) {
            const int pp = civ / a.cinp32;
            ci = civ - pp * a.cinp32;
            ky = 2 * (tap >> 1) + (pp >> 1);
            kx = 2 * (tap & 1) + (pp & 1);
        } else {
            ci = civ;
            ky = tap / ga.kw;
            kx = tap - ky * ga.kw;
        }
        if (co < a.Cout && ci < a.Cin) {
            const bool second = co >= a.split;
            const size_t o = (((size_t)(second ? co - a.split : co) * (second ? a.cin_total2 : a.cin_total) + a.cin_begin + ci) * ga.kh + ky) * ga.kw + kx;
            float *dst = (second ? a.dw2 : a.dw) + o;
            const float prev = (a.beta != 0.f) ? a.beta * *dst : 0.f;
            *dst = prev + a.alpha * sum;
        }
    }
}

struct WgPlan {
    int a_t, b_t, thg, ks, wps, db;
    int ncib, ncob;
    int KoutP, KinVP, cinp32;
    int tiles_x, tiles_y, tiles_total, splits, tiles_per_split;
    int ntaps, resident;
    int64_t ws_floats, db_floats;
};

// Tile class and slab geometry of one layer; the split count is chosen for `group_jobs` (cin block, cout
// block) pairs sharing the launch (0: this layer alone).
int plan_wgrad(const tnr_wgrad_desc *d, WgPlan &p, int group_jobs) {
    const bool s2d = d->mode == TNR_CONV_4x4_S2;
    p.a_t = d->Cout > 32 ? 2 : 1;
    const int vch = s2d ? 4 * tnr_round_up(d->Cin, 32) : tnr_round_up(d->Cin, 32);
    const int vblocks = vch / 32;
    if (p.a_t == 2) {
        p.b_t = vblocks >= 2 ? 2 : 1;
    } else {
        p.b_t = vblocks >= 4 ? 4 : vblocks;
        if (vblocks == 5) p.b_t = 3;  // caller normally splits 160 = 96 + 64 itself
    }
    // LDS budget: two workgroups per CU (<= 80 KiB each) except the 1-workgroup regime (J >= 9, <= 160 KiB)
    p.thg = (p.a_t == 2 || p.b_t <= 2 || (p.b_t == 4 && d->mode != TNR_CONV_4x4_S2)) ? 8 : 4;
    const int ab = p.a_t * p.b_t;
    p.ks = (!s2d && (ab == 1 || ab == 2)) ? 4 / ab : 1;      // WgCfg::KS
    if (p.ks > 1) p.thg = 16;                                // each pixel group keeps >= 4 rows (J = 9: one workgroup per CU)
    // the pre-split LDS image of TNR_MMA_BF16X3 takes 6 bytes per element: the 32 x 64 class halves its tile to stay inside 160 KB
    if (p.ks > 1 && d->mma == TNR_MMA_BF16X3 && p.a_t == 1 && p.b_t == 2) p.thg = 8;
    const int wpg0 = 4 / p.ks;
    p.wps = ((p.a_t * p.b_t * (s2d ? 4 : 9) + wpg0 - 1) / wpg0 >= 9) ? 1 : 2;      // WgCfg::WAVES_PER_SIMD
    // TNR_MMA_BF16X3, 3x3 classes: half-height tiles, two workgroups per CU (see wgrad_tile_kernel; TNR_WG_X3_OCC=1 keeps one)
    static const int x3_occ = [] { const char *e = getenv("TNR_WG_X3_OCC"); return e ? atoi(e) : 3; }();
    p.db = 0;
    if (d->mma == TNR_MMA_BF16X3 && s2d && x3_occ >= 2 && p.a_t == 2 && p.b_t == 2) p.thg = 4;       // (dispatch_wgrad: two workgroups per CU)
    if (d->mma == TNR_MMA_BF16X3 && !s2d && x3_occ >= 2 && p.wps == 1) {
        if (p.b_t == 4) p.b_t = 2;               // 32 x 128 jobs become two 32 x 64 jobs (the 128-wide halo tile alone is 95 KB)
        const int ab2 = p.a_t * p.b_t;
        p.ks = (ab2 == 1 || ab2 == 2) ? 4 / ab2 : 1;
        p.thg = ab2 == 1 ? 8 : 4;                // <= 66 KB of pre-split LDS image per tile set, >= 2 tile rows per pixel group
        p.wps = 2;
        // TNR_WG_X3_OCC=3 (default): the pipelined one-workgroup form with two tile sets (wgrad_tile_kernel<.., DB>), zero-padded 3x3 only
        if (x3_occ == 3 && d->mode == TNR_CONV_3x3 && d->pad_mode == 0 && p.b_t <= 2 &&
            (int64_t)d->N * d->H * d->W * d->x.ctot < (1LL << 30) && (int64_t)d->N * d->Ho * d->Wo * d->g.ctot < (1LL << 30)) {
            p.wps = 1;
            p.db = 1;
        }
    }
    p.cinp32 = tnr_round_up(d->Cin, 32);
    p.KinVP = vch;
    p.KoutP = tnr_round_up(d->Cout, 32);
    p.ncib = tnr_cdiv(vblocks, p.b_t);
    p.ncob = tnr_cdiv(p.KoutP, 32 * p.a_t);
    p.ntaps = s2d ? 4 : 9;
    p.tiles_x = tnr_cdiv(d->Wo, 16);
    p.tiles_y = tnr_cdiv(d->Ho, p.thg);
    p.tiles_total = p.tiles_x * p.tiles_y * d->N;
    p.resident = 256 * p.wps;                    // workgroups that fit the chip at once in this regime
    const int jobs = group_jobs > 0 ? group_jobs : p.ncib * p.ncob;
    int want = p.resident / jobs;                // one full wave of workgroups, never a straggler
    if (want < 1) want = 1;
    int max_splits = tnr_cdiv(p.tiles_total, 4); // never fewer than 4 tiles of work per split
    if (max_splits < 1) max_splits = 1;
    p.splits = want < max_splits ? want : max_splits;
    if (p.splits < 1) p.splits = 1;
    p.tiles_per_split = tnr_cdiv(p.tiles_total, p.splits);
    p.splits = tnr_cdiv(p.tiles_total, p.tiles_per_split);
    p.ws_floats = (int64_t)p.splits * p.ntaps * p.KoutP * p.KinVP;
    p.db_floats = (int64_t)p.splits * p.KoutP;
    return 0;
}

template <int MODE, int A_T, int B_T, int THG, int BF, int WPS = WgCfg<A_T, B_T, (MODE == TNR_CONV_4x4_S2 ? 4 : 9)>::WAVES_PER_SIMD, bool DB = false>
int launch_wgrad_t(const WgK &k, int jobs, hipStream_t s) {
    constexpr int KH = (MODE == TNR_CONV_4x4_S2) ? 2 : 3;
    // + one halo row: the k-loop's last prefetch reads one row past the x tile (never consumed)
    // (TNR_MMA_BF16X3: both tiles pre-split into three bf16 planes, 6 bytes per element, pixels in blocks of 4: wg_x3_off)
    constexpr size_t lds_f32 = (size_t)(THG * 16 * 32 * A_T + (THG + KH) * (16 + KH - 1) * 32 * B_T) * sizeof(float);
    constexpr size_t lds_x3 = (size_t)(THG * 16 * 32 * A_T + (((THG + KH - 1) * (16 + KH - 1) + 3) / 4) * 4 * 32 * B_T) * 6;
    using Cfg = WgCfg<A_T, B_T, (MODE == TNR_CONV_4x4_S2 ? 4 : 9)>;
    constexpr size_t lds_red = Cfg::KS > 1 ? (size_t)(Cfg::WPG * Cfg::J * 16 * 64 + Cfg::WPG * 64) * sizeof(float) : 0;   // pixel-group exchange
    constexpr size_t lds_db = 2 * ((size_t)(THG * 16 * 32 * A_T + (THG + KH - 1) * 20 * 32 * B_T) * 6 + 3072);      // two tile sets (halo rows of 20 pixels) + their dump areas
    constexpr size_t lds_tile = DB ? lds_db : (BF == 2 ? lds_x3 : lds_f32);
    constexpr size_t lds = lds_tile > lds_red ? lds_tile : lds_red;
    constexpr bool one_wg = WPS == 1;
    if constexpr (BF == 2 && lds > 160 * 1024) {       // (a tile class plan_wgrad never picks in this mode)
        tnr_set_error("wgrad_tile: tile class %d x %d x %d rows does not fit the LDS in TNR_MMA_BF16X3", A_T, B_T, THG);
        return TNR_EINVAL;
    } else {
    // (TNR_MMA_BF16X3 classes inherited from the fp32 plan may exceed 80 KB: the LDS then limits them to one workgroup per CU)
    static_assert(lds <= ((one_wg || BF == 2) ? 160 : 80) * 1024, "wgrad tile exceeds the LDS budget of its occupancy regime");
    static bool attr_done = false;
    auto fn = wgrad_tile_kernel<MODE, A_T, B_T, THG, BF, WPS, DB>;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
            tnr_set_error("wgrad_tile: cannot raise dynamic LDS to %zu bytes", lds);
            return TNR_ELAUNCH;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(fn, dim3(k.nsplits, jobs, 1), dim3(256), lds, s, k);
    return tnr_check_launch("wgrad_tile");
    }
}

template <int MODE, int A_T, int B_T, int THG>
int launch_wgrad(const WgK &k, int jobs, hipStream_t s) {
    if (k.bf == 2) return launch_wgrad_t<MODE, A_T, B_T, THG, 2>(k, jobs, s);
    return k.bf ? launch_wgrad_t<MODE, A_T, B_T, THG, 1>(k, jobs, s) : launch_wgrad_t<MODE, A_T, B_T, THG, 0>(k, jobs, s);
}

template <int MODE>
int dispatch_wgrad(const WgK &k, const WgPlan &p, int jobs, hipStream_t s) {
    constexpr int THS = (MODE == TNR_CONV_4x4_S2) ? 8 : 16;   // tile rows of the pixel-split classes (plan_wgrad: ks > 1)
    if constexpr (MODE == TNR_CONV_3x3) {
        if (k.bf == 2 && p.db) {                              // TNR_MMA_BF16X3: the pipelined form (plan_wgrad)
            if (p.a_t == 2) return p.b_t == 2 ? launch_wgrad_t<MODE, 2, 2, 4, 2, 1, true>(k, jobs, s) : launch_wgrad_t<MODE, 2, 1, 4, 2, 1, true>(k, jobs, s);
            return p.b_t == 2 ? launch_wgrad_t<MODE, 1, 2, 4, 2, 1, true>(k, jobs, s) : launch_wgrad_t<MODE, 1, 1, 8, 2, 1, true>(k, jobs, s);
        }
    }
    if constexpr (MODE != TNR_CONV_4x4_S2) {
        if (k.bf == 2 && p.wps == 2 && p.b_t <= 2) {          // TNR_MMA_BF16X3: half-height tiles, two workgroups per CU (plan_wgrad)
            if (p.a_t == 2) return p.b_t == 2 ? launch_wgrad_t<MODE, 2, 2, 4, 2, 2>(k, jobs, s) : launch_wgrad_t<MODE, 2, 1, 4, 2, 2>(k, jobs, s);
            return p.b_t == 2 ? launch_wgrad_t<MODE, 1, 2, 4, 2, 2>(k, jobs, s) : launch_wgrad_t<MODE, 1, 1, 8, 2, 2>(k, jobs, s);
        }
    }
    if (p.a_t == 2) {
        if constexpr (MODE == TNR_CONV_4x4_S2) {
            // TNR_MMA_BF16X3: the 8-row tile's pre-split image is 109 KB (one workgroup per CU, nothing hides the refill); 4 rows: 58 KB, two
            if (p.b_t == 2 && k.bf == 2 && p.thg == 4) return launch_wgrad_t<MODE, 2, 2, 4, 2, 2>(k, jobs, s);
        }
        if (p.b_t == 2) return launch_wgrad<MODE, 2, 2, 8>(k, jobs, s);
        return launch_wgrad<MODE, 2, 1, THS>(k, jobs, s);
    }
    switch (p.b_t) {
        case 1: return launch_wgrad<MODE, 1, 1, THS>(k, jobs, s);
        case 2:
            if constexpr (MODE != TNR_CONV_4x4_S2) {
                if (p.thg == 8) return launch_wgrad_t<MODE, 1, 2, 8, 2>(k, jobs, s);          // (bf16x3 only: plan_wgrad)
            }
            return launch_wgrad<MODE, 1, 2, THS>(k, jobs, s);
        case 3: return launch_wgrad<MODE, 1, 3, 4>(k, jobs, s);
        default:
            if constexpr (MODE != TNR_CONV_4x4_S2) {
                if (p.thg == 8) return launch_wgrad<MODE, 1, 4, 8>(k, jobs, s);
            }
            return launch_wgrad<MODE, 1, 4, 4>(k, jobs, s);
    }
}

int check_wgrad_desc(const tnr_wgrad_desc *d) {
    TNR_REQUIRE(d->x.ptr && d->g.ptr && d->dw && d->ws, "wgrad: null pointer");
    TNR_REQUIRE(d->mode == TNR_CONV_3x3 || d->mode == TNR_CONV_3x3_UP2 || d->mode == TNR_CONV_4x4_S2,
                "wgrad: bad mode %d", d->mode);
    TNR_REQUIRE((d->x.ctot % 4) == 0 && (d->x.coff % 4) == 0 && (d->g.ctot % 4) == 0 && (d->g.coff % 4) == 0,
                "wgrad: views must be 4-channel aligned");
    TNR_REQUIRE(d->db == nullptr || d->cin_begin == 0, "wgrad: bias gradient only with cin_begin == 0");
    if (d->cout_split != 0) {
        TNR_REQUIRE(d->cout_split > 0 && d->cout_split < d->Cout && d->cout_split % 32 == 0 && d->dw2 != nullptr && d->cin_total2 >= d->cin_begin + d->Cin,
                    "wgrad: bad cout pair (split %d of %d output channels)", d->cout_split, d->Cout);
        TNR_REQUIRE((d->db == nullptr) == (d->db2 == nullptr), "wgrad: a cout pair takes both bias gradients or none");
    }
    if (d->mode == TNR_CONV_3x3) TNR_REQUIRE(d->Ho == d->H && d->Wo == d->W, "wgrad3x3: size mismatch");
    if (d->mode == TNR_CONV_3x3_UP2) TNR_REQUIRE(d->Ho == 2 * d->H && d->Wo == 2 * d->W, "wgrad3x3_up2: size mismatch");
    if (d->mode == TNR_CONV_4x4_S2) TNR_REQUIRE(2 * d->Ho == d->H && 2 * d->Wo == d->W, "wgrad4x4s2: size mismatch");
    TNR_REQUIRE((int64_t)d->N * d->H * d->W * d->x.ctot < (1LL << 31) && (int64_t)d->N * d->Ho * d->Wo * d->g.ctot < (1LL << 31),
                "wgrad: buffers above 2^31 elements need 64-bit offsets");
    return TNR_OK;
}

}  // namespace

extern "C" int64_t tnr_wgrad_workspace_bytes(const tnr_wgrad_desc *d) {
    if (d == nullptr) return 0;
    WgPlan p;
    plan_wgrad(d, p, 0);   // a layer alone uses the most splits: this bound also covers any group
    return (p.ws_floats + p.db_floats) * (int64_t)sizeof(float);
}

extern "C" int tnr_conv_wgrad_group(const tnr_wgrad_desc *descs, int32_t n, void *stream) {
    TNR_REQUIRE(descs != nullptr && n >= 1 && n <= TNR_WGRAD_GROUP_MAX, "wgrad_group: 1..%d layers per group", TNR_WGRAD_GROUP_MAX);
    WgPlan plans[TNR_WGRAD_GROUP_MAX];
    int jobs = 0;
    for (int i = 0; i < n; ++i) {
        const int rc = check_wgrad_desc(&descs[i]);
        if (rc != TNR_OK) return rc;
        plan_wgrad(&descs[i], plans[i], 0);
        jobs += plans[i].ncib * plans[i].ncob;
        const tnr_wgrad_desc &d0 = descs[0], &di = descs[i];
        TNR_REQUIRE(di.mode == d0.mode && di.N == d0.N && di.H == d0.H && di.W == d0.W && di.Ho == d0.Ho && di.Wo == d0.Wo,
                    "wgrad_group: layer %d does not share the pixel geometry of layer 0", i);
        TNR_REQUIRE(plans[i].a_t == plans[0].a_t && plans[i].b_t == plans[0].b_t && plans[i].thg == plans[0].thg && plans[i].ks == plans[0].ks && plans[i].db == plans[0].db,
                    "wgrad_group: layer %d (%d->%d channels) is not in the tile class of layer 0 (%d->%d)", i, di.Cin,
                    di.Cout, d0.Cin, d0.Cout);
    }
    TNR_REQUIRE(jobs <= 65535, "wgrad_group: too many channel blocks");
    WgK k;
    RedK r;
    static const int red_rpb = [] { const char *e = getenv("TNR_WGRAD_REDUCE_RPB"); return e ? atoi(e) : 8; }();      // (A/B switch: 1 = a block per row always)
    WgPlan pg = plans[0];
    if (n > 1) plan_wgrad(&descs[0], pg, jobs);
    const int rpb = (red_rpb == 8 && pg.splits <= 32) ? 8 : 1;
    int job_begin = 0, blk_begin = 0;
    for (int i = 0; i < n; ++i) {
        const tnr_wgrad_desc *d = &descs[i];
        WgPlan &p = plans[i];
        if (n > 1) plan_wgrad(d, p, jobs);
        TNR_REQUIRE(p.splits == plans[0].splits && p.tiles_per_split == plans[0].tiles_per_split, "wgrad_group: split mismatch");
        TNR_REQUIRE((p.ws_floats + p.db_floats) * (int64_t)sizeof(float) <= d->ws_bytes,
                    "wgrad: workspace too small (%lld < %lld)", (long long)d->ws_bytes,
                    (long long)((p.ws_floats + p.db_floats) * sizeof(float)));
        for (int q = 0; q < i; ++q) TNR_REQUIRE(descs[q].ws != d->ws, "wgrad_group: layers %d and %d share a workspace", q, i);
        WgJob &j = k.job[i];
        j.x = d->x.ptr; j.x_ct = d->x.ctot; j.x_co = d->x.coff; j.Cin = d->Cin;
        j.g = d->g.ptr; j.g_ct = d->g.ctot; j.g_co = d->g.coff; j.Cout = d->Cout;
        j.ws = d->ws; j.dbp = d->db ? d->ws + p.ws_floats : nullptr;
        j.KoutP = p.KoutP; j.KinVP = p.KinVP; j.cinp32 = p.cinp32;
        j.ncib = p.ncib; j.job_begin = job_begin;
        job_begin += p.ncib * p.ncob;
        RedJob &q = r.job[i];
        q.ws = d->ws; q.dbp = j.dbp; q.dw = d->dw; q.db = d->db;
        q.dw2 = d->dw2; q.db2 = d->db2; q.split = d->cout_split > 0 ? d->cout_split : d->Cout; q.cin_total2 = d->cin_total2;
        q.KoutP = p.KoutP; q.KinVP = p.KinVP; q.cinp32 = p.cinp32; q.Cout = d->Cout; q.Cin = d->Cin;
        q.cin_total = d->cin_total; q.cin_begin = d->cin_begin; q.alpha = d->alpha; q.beta = d->beta;
        q.blk_begin = blk_begin;
        q.nrows = p.ntaps * p.KoutP * (p.KinVP / 32);
        blk_begin += tnr_cdiv(q.nrows, rpb) + (d->db ? tnr_cdiv(d->Cout, 32) : 0);
    }
    for (int i = n; i < TNR_WGRAD_GROUP_MAX; ++i) { k.job[i] = k.job[0]; r.job[i] = r.job[0]; }
    const tnr_wgrad_desc &d0 = descs[0];
    const WgPlan &p0 = plans[0];
    k.N = d0.N; k.H = d0.H; k.W = d0.W; k.Ho = d0.Ho; k.Wo = d0.Wo;
    k.tiles_x = p0.tiles_x; k.tiles_y = p0.tiles_y; k.tiles_total = p0.tiles_total;
    k.tiles_per_split = p0.tiles_per_split; k.nsplits = p0.splits; k.njobs = n;
    TNR_REQUIRE(d0.mma >= TNR_MMA_F32 && d0.mma <= TNR_MMA_BF16X3, "wgrad: bad mma %d", d0.mma);
    k.bf = d0.mma;
    k.reflect = d0.pad_mode == 1;
    TNR_REQUIRE(d0.pad_mode == 0 || (d0.pad_mode == 1 && d0.mode == TNR_CONV_3x3 && d0.H >= 2 && d0.W >= 2), "wgrad: pad_mode 1 (reflection) is for TNR_CONV_3x3");
    for (int i = 1; i < n; ++i) TNR_REQUIRE(descs[i].pad_mode == d0.pad_mode, "wgrad_group: layer %d: one border mode per launch", i);
    for (int i = 1; i < n; ++i) TNR_REQUIRE(descs[i].mma == d0.mma, "wgrad_group: layer %d: one matrix-core precision per launch", i);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    switch (d0.mode) {
        case TNR_CONV_3x3: rc = dispatch_wgrad<TNR_CONV_3x3>(k, p0, jobs, s); break;
        case TNR_CONV_3x3_UP2: rc = dispatch_wgrad<TNR_CONV_3x3_UP2>(k, p0, jobs, s); break;
        default: rc = dispatch_wgrad<TNR_CONV_4x4_S2>(k, p0, jobs, s); break;
    }
    if (rc != TNR_OK) return rc;
    r.splits = p0.splits; r.ntaps = p0.ntaps; r.s2d = d0.mode == TNR_CONV_4x4_S2;
    r.kh = r.s2d ? 4 : 3; r.kw = r.kh; r.njobs = n; r.rpb = rpb;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blk_begin), dim3(256), 0, s, r);
    return tnr_check_launch("wgrad_reduce");
}

extern "C" int tnr_conv_wgrad(const tnr_wgrad_desc *d, void *stream) {
    TNR_REQUIRE(d != nullptr, "wgrad: null descriptor");
    return tnr_conv_wgrad_group(d, 1, stream);
}
