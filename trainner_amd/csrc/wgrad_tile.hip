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
#include "common.h"

namespace {

struct WgK {
    const float *x; int x_ct, x_co; int N, H, W, Cin;
    const float *g; int g_ct, g_co; int Ho, Wo, Cout;
    float *ws; int KoutP, KinVP;  // slab dims
    int cinp32;                   // per-parity padded channel count (S2D); == KinVP otherwise
    float *dbp;                   // partial bias sums [split][KoutP] or null
    int tiles_x, tiles_y, tiles_total, tiles_per_split, nsplits;
};

// Two occupancy regimes (chosen by the accumulator count J = ceil(A_T*B_T*taps/4) per wave):
//   J < 9  : <= 200 VGPRs, TWO workgroups per CU; staging loads are issued in batches of 8 float4 per
//            thread and the other workgroup's MFMA phase hides their latency;
//   J >= 9 : 144 accumulator VGPRs, ONE workgroup per CU with the whole 512-entry register file: the
//            global loads of tile i+1 are all issued before the MFMA phase of tile i and written to LDS
//            after it (issue-early / write-late), and the k-loop is unrolled 4 k-steps deep.
template <int A_T, int B_T, int NTAPS_>
struct WgCfg {
    static constexpr int J = (A_T * B_T * NTAPS_ + 3) / 4;
    static constexpr bool PIPE = J >= 9;
    static constexpr int WAVES_PER_SIMD = PIPE ? 1 : 2;
};

template <int MODE, int A_T, int B_T, int THG>
__global__ void __launch_bounds__(256, (WgCfg<A_T, B_T, (MODE == TNR_CONV_4x4_S2 ? 4 : 9)>::WAVES_PER_SIMD))
wgrad_tile_kernel(const WgK a) {
    constexpr bool S2D = (MODE == TNR_CONV_4x4_S2);
    constexpr bool UP = (MODE == TNR_CONV_3x3_UP2);
    constexpr int TWG = 16, PX = THG * TWG;
    constexpr int KH = S2D ? 2 : 3;
    constexpr int NTAPS = KH * KH;
    constexpr int HT = THG + KH - 1, WT = TWG + KH - 1;
    constexpr int COB = 32 * A_T, CIB = 32 * B_T;
    constexpr int AB = A_T * B_T;
    constexpr int T = AB * NTAPS, J = (T + 3) / 4;
    constexpr bool PIPE = WgCfg<A_T, B_T, NTAPS>::PIPE;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_g = smem;             // PX * COB
    float *s_x = smem + PX * COB;  // HT*WT * CIB

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> SGPR
    const int lane = tid & 63, li = lane & 31, half = lane >> 5;
    const int split = blockIdx.x, cib = blockIdx.y, cob = blockIdx.z;

    // tile list of this wave (wave-uniform scalars)
    int t_ok[J], t_tap[J], t_aa[J], t_boff[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int t = wave + 4 * j;
        t_ok[j] = t < T;
        const int tt = t_ok[j] ? t : 0;
        const int tap = tt / AB, ab = tt - tap * AB;
        const int aa = ab / B_T, bb = ab - aa * B_T;
        const int ty = tap / KH, tx = tap - ty * KH;
        t_tap[j] = tap * 1024 + aa * 32 + bb;  // packed for the store phase
        t_aa[j] = aa;
        t_boff[j] = (ty * WT + tx) * CIB + bb * 32;  // slots beyond T alias tile 0: computed, never stored
    }

    f32x16 acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float bsum[A_T];   // per-lane partial bias gradient: sum over pixels of g[p][aa*32 + li] (both pixel parities)
#pragma unroll
    for (int aa = 0; aa < A_T; ++aa) bsum[aa] = 0.f;
    const bool want_bias = (a.dbp != nullptr) && (cib == 0);

    const int t_begin = split * a.tiles_per_split;
    int t_end = t_begin + a.tiles_per_split;
    if (t_end > a.tiles_total) t_end = a.tiles_total;

    // ---- staging: g tile [PX][COB] then x halo tile [HT*WT][CIB], one float4 per item
    constexpr int G_ITEMS = PX * (COB / 4), G_IT = (G_ITEMS + 255) / 256;
    constexpr int X_ITEMS = HT * WT * (CIB / 4), X_IT = (X_ITEMS + 255) / 256;
    constexpr int N_IT = G_IT + X_IT;
    constexpr int BATCH = PIPE ? N_IT : 8;
    constexpr int NBATCH = (N_IT + BATCH - 1) / BATCH;
    f32x4 rr[BATCH];
    auto load_batch = [&](int tile, int batch) {
        int q = tile;
        const int tx = q % a.tiles_x;
        q /= a.tiles_x;
        const int ty = q % a.tiles_y;
        const int n = q / a.tiles_y;
        const int ty0 = ty * THG, tx0 = tx * TWG;
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
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
    auto store_batch = [&](int batch) {
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int it = batch * BATCH + k;
            if (it < G_IT) {
                const int i = tid + it * 256;
                if (i < G_ITEMS) *reinterpret_cast<f32x4 *>(s_g + i * 4) = rr[k];
            } else if (it < N_IT) {
                const int i = tid + (it - G_IT) * 256;
                if (i < X_ITEMS) *reinterpret_cast<f32x4 *>(s_x + i * 4) = rr[k];
            }
        }
    };

    if (PIPE && t_begin < t_end) load_batch(t_begin, 0);
    for (int tile = t_begin; tile < t_end; ++tile) {
        if (PIPE) {
            __syncthreads();  // previous tile's fragments are consumed
            store_batch(0);
            __syncthreads();
            if (tile + 1 < t_end) load_batch(tile + 1, 0);  // in flight during the MFMA phase below
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
        // ---- K loop: two pixels per MFMA, one tile row (16 pixels = 8 k-steps) per outer iteration.
        // Inside a row every LDS address is row base + compile-time offset, so the unrolled body is
        // ds_read (immediate offsets) + MFMA only; the scheduler hoists the reads of later k-steps above
        // the MFMAs of earlier ones.
#pragma unroll 1
        for (int r = 0; r < THG; ++r) {
            const float *gr = s_g + (r * TWG + half) * COB + li;
            const float *xr = s_x + (r * WT + half) * CIB + li;
            const float *xj[J];
#pragma unroll
            for (int j = 0; j < J; ++j) xj[j] = xr + t_boff[j];
            constexpr int KU = PIPE ? 4 : 8;   // k-steps unrolled together
#pragma unroll 1
            for (int k0 = 0; k0 < TWG / 2; k0 += KU) {
#pragma unroll
                for (int kk = 0; kk < KU; ++kk) {
                    const int k = k0 + kk;
                    float av[A_T];
#pragma unroll
                    for (int aa = 0; aa < A_T; ++aa) {
                        av[aa] = gr[2 * k * COB + aa * 32];
                        bsum[aa] += av[aa];
                    }
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        float aval = av[0];
                        if (A_T > 1) aval = t_aa[j] ? av[A_T - 1] : av[0];
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aval, xj[j][2 * k * CIB], acc[j], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- write the partial slab
#pragma unroll
    for (int j = 0; j < J; ++j) {
        if (!t_ok[j]) continue;
        const int tap = t_tap[j] >> 10, aa = (t_tap[j] >> 5) & 31, bb = t_tap[j] & 31;
        const int blk = cib * B_T + bb;                // 32-wide virtual input-channel block
        const int nblk = a.KinVP >> 5;
        if (blk >= nblk) continue;
        // slab layout [tap][co][blk][split][32]: the split axis is contiguous (128-B granules), so the
        // reducer streams each (tap, co, blk) row; a wave store still writes 128 B per half-wave
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int co = cob * COB + aa * 32 + i;
            if (co < a.KoutP)
                a.ws[((((size_t)tap * a.KoutP + co) * nblk + blk) * a.nsplits + split) * 32 + li] = acc[j][r];
        }
    }
    if (want_bias && wave == 0) {   // every wave accumulated the same sums; wave 0 publishes them
#pragma unroll
        for (int aa = 0; aa < A_T; ++aa) {
            const float tot = bsum[aa] + __shfl_xor(bsum[aa], 32);
            const int co = cob * COB + aa * 32 + li;
            if (half == 0 && co < a.KoutP) a.dbp[(size_t)co * a.nsplits + split] = tot;   // [co][split]
        }
    }
}

struct RedK {
    const float *ws; const float *dbp;
    int splits, ntaps, KoutP, KinVP, cinp32;
    float *dw; float *db;
    int Cout, Cin, cin_total, cin_begin, kh, kw, s2d;
    float alpha, beta;
};

// One block per slab row (tap, co, 32-channel block): 256 threads = 32 channels x 8 split lanes.  The
// row's partials [split][32] are contiguous, so the 8 lanes stream 1 KiB per step; lane sums are combined
// in a fixed order through LDS (deterministic).
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const RedK a) {
    __shared__ float sh[8][33];
    const int nblk = a.KinVP >> 5;
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int nrows = a.ntaps * a.KoutP * nblk;
    if ((int)blockIdx.x >= nrows) {
        // bias blocks: 32 output channels each, the 8 split lanes stream dbp[co][split]
        const int c = ((int)blockIdx.x - nrows) * 32 + el;
        float part = 0.f;
        if (c < a.Cout)
            for (int q = sl; q < a.splits; q += 8) part += a.dbp[(size_t)c * a.splits + q];
        sh[sl][el] = part;
        __syncthreads();
        if (sl == 0 && c < a.Cout) {
            float sum = 0.f;
#pragma unroll
            for (int l = 0; l < 8; ++l) sum += sh[l][el];
            const float prev = (a.beta != 0.f) ? a.beta * a.db[c] : 0.f;
            a.db[c] = prev + a.alpha * sum;
        }
        return;
    }
    const int row = blockIdx.x;                         // ((tap * KoutP) + co) * nblk + blk
    const int blk = row % nblk;
    const int co = (row / nblk) % a.KoutP;
    const int tap = row / (nblk * a.KoutP);
    const float *src = a.ws + (size_t)row * a.splits * 32 + el;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    int s = sl;
    for (; s + 24 < a.splits; s += 32) {                // four independent loads in flight per lane
        p0 += src[(size_t)s * 32];
        p1 += src[(size_t)(s + 8) * 32];
        p2 += src[(size_t)(s + 16) * 32];
        p3 += src[(size_t)(s + 24) * 32];
    }
    for (; s < a.splits; s += 8) p0 += src[(size_t)s * 32];
    sh[sl][el] = (p0 + p1) + (p2 + p3);
    __syncthreads();
    if (sl == 0) {
        float sum = 0.f;
#pragma unroll
        for (int l = 0; l < 8; ++l) sum += sh[l][el];
        const int civ = blk * 32 + el;
        int ci, ky, kx;
        if (a.s2d) {
            const int pp = civ / a.cinp32;
            ci = civ - pp * a.cinp32;
            ky = 2 * (tap >> 1) + (pp >> 1);
            kx = 2 * (tap & 1) + (pp & 1);
        } else {
            ci = civ;
            ky = tap / a.kw;
            kx = tap - ky * a.kw;
        }
        if (co < a.Cout && ci < a.Cin) {
            const size_t o = (((size_t)co * a.cin_total + a.cin_begin + ci) * a.kh + ky) * a.kw + kx;
            const float prev = (a.beta != 0.f) ? a.beta * a.dw[o] : 0.f;
            a.dw[o] = prev + a.alpha * sum;
        }
    }
}

struct WgPlan {
    int a_t, b_t, thg;
    int ncib, ncob;
    int KoutP, KinVP, cinp32;
    int tiles_x, tiles_y, tiles_total, splits, tiles_per_split;
    int ntaps;
    int64_t ws_floats, db_floats;
};

int plan_wgrad(const tnr_wgrad_desc *d, WgPlan &p) {
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
    p.cinp32 = tnr_round_up(d->Cin, 32);
    p.KinVP = vch;
    p.KoutP = tnr_round_up(d->Cout, 32);
    p.ncib = tnr_cdiv(vblocks, p.b_t);
    p.ncob = tnr_cdiv(p.KoutP, 32 * p.a_t);
    p.ntaps = s2d ? 4 : 9;
    p.tiles_x = tnr_cdiv(d->Wo, 16);
    p.tiles_y = tnr_cdiv(d->Ho, p.thg);
    p.tiles_total = p.tiles_x * p.tiles_y * d->N;
    // enough workgroups for ~2 per CU, but never fewer than 4 tiles of work per split
    const int J = (p.a_t * p.b_t * p.ntaps + 3) / 4;
    const int resident = (J >= 9) ? 256 : 512;   // workgroups that fit the chip at once in this regime
    int want = resident / (p.ncib * p.ncob);     // one full wave of workgroups, never a straggler
    if (want < 1) want = 1;
    int max_splits = tnr_cdiv(p.tiles_total, 4);
    if (max_splits < 1) max_splits = 1;
    p.splits = want < max_splits ? want : max_splits;
    if (p.splits < 1) p.splits = 1;
    p.tiles_per_split = tnr_cdiv(p.tiles_total, p.splits);
    p.splits = tnr_cdiv(p.tiles_total, p.tiles_per_split);
    p.ws_floats = (int64_t)p.splits * p.ntaps * p.KoutP * p.KinVP;
    p.db_floats = (int64_t)p.splits * p.KoutP;
    return 0;
}

template <int MODE, int A_T, int B_T, int THG>
int launch_wgrad(const WgK &k, const WgPlan &p, hipStream_t s) {
    constexpr int KH = (MODE == TNR_CONV_4x4_S2) ? 2 : 3;
    constexpr size_t lds = (size_t)(THG * 16 * 32 * A_T + (THG + KH - 1) * (16 + KH - 1) * 32 * B_T) * sizeof(float);
    constexpr bool pipe = WgCfg<A_T, B_T, (MODE == TNR_CONV_4x4_S2 ? 4 : 9)>::PIPE;
    static_assert(lds <= (pipe ? 160 : 80) * 1024, "wgrad tile exceeds the LDS budget of its occupancy regime");
    static bool attr_done = false;
    auto fn = wgrad_tile_kernel<MODE, A_T, B_T, THG>;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
            tnr_set_error("wgrad_tile: cannot raise dynamic LDS to %zu bytes", lds);
            return TNR_ELAUNCH;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(fn, dim3(p.splits, p.ncib, p.ncob), dim3(256), lds, s, k);
    return tnr_check_launch("wgrad_tile");
}

template <int MODE>
int dispatch_wgrad(const WgK &k, const WgPlan &p, hipStream_t s) {
    if (p.a_t == 2) {
        if (p.b_t == 2) return launch_wgrad<MODE, 2, 2, 8>(k, p, s);
        return launch_wgrad<MODE, 2, 1, 8>(k, p, s);
    }
    switch (p.b_t) {
        case 1: return launch_wgrad<MODE, 1, 1, 8>(k, p, s);
        case 2: return launch_wgrad<MODE, 1, 2, 8>(k, p, s);
        case 3: return launch_wgrad<MODE, 1, 3, 4>(k, p, s);
        default:
            if constexpr (MODE != TNR_CONV_4x4_S2) {
                if (p.thg == 8) return launch_wgrad<MODE, 1, 4, 8>(k, p, s);
            }
            return launch_wgrad<MODE, 1, 4, 4>(k, p, s);
    }
}

}  // namespace

extern "C" int64_t tnr_wgrad_workspace_bytes(const tnr_wgrad_desc *d) {
    if (d == nullptr) return 0;
    WgPlan p;
    plan_wgrad(d, p);
    return (p.ws_floats + p.db_floats) * (int64_t)sizeof(float);
}

extern "C" int tnr_conv_wgrad(const tnr_wgrad_desc *d, void *stream) {
    TNR_REQUIRE(d != nullptr && d->x.ptr && d->g.ptr && d->dw && d->ws, "wgrad: null pointer");
    TNR_REQUIRE(d->mode == TNR_CONV_3x3 || d->mode == TNR_CONV_3x3_UP2 || d->mode == TNR_CONV_4x4_S2,
                "wgrad: bad mode %d", d->mode);
    TNR_REQUIRE((d->x.ctot % 4) == 0 && (d->x.coff % 4) == 0 && (d->g.ctot % 4) == 0 && (d->g.coff % 4) == 0,
                "wgrad: views must be 4-channel aligned");
    TNR_REQUIRE(d->db == nullptr || d->cin_begin == 0, "wgrad: bias gradient only with cin_begin == 0");
    if (d->mode == TNR_CONV_3x3) TNR_REQUIRE(d->Ho == d->H && d->Wo == d->W, "wgrad3x3: size mismatch");
    if (d->mode == TNR_CONV_3x3_UP2) TNR_REQUIRE(d->Ho == 2 * d->H && d->Wo == 2 * d->W, "wgrad3x3_up2: size mismatch");
    if (d->mode == TNR_CONV_4x4_S2) TNR_REQUIRE(2 * d->Ho == d->H && 2 * d->Wo == d->W, "wgrad4x4s2: size mismatch");
    TNR_REQUIRE((int64_t)d->N * d->H * d->W * d->x.ctot < (1LL << 31) && (int64_t)d->N * d->Ho * d->Wo * d->g.ctot < (1LL << 31),
                "wgrad: buffers above 2^31 elements need 64-bit offsets");
    WgPlan p;
    plan_wgrad(d, p);
    TNR_REQUIRE((p.ws_floats + p.db_floats) * (int64_t)sizeof(float) <= d->ws_bytes, "wgrad: workspace too small (%lld < %lld)",
                (long long)d->ws_bytes, (long long)((p.ws_floats + p.db_floats) * sizeof(float)));
    WgK k;
    k.x = d->x.ptr; k.x_ct = d->x.ctot; k.x_co = d->x.coff; k.N = d->N; k.H = d->H; k.W = d->W; k.Cin = d->Cin;
    k.g = d->g.ptr; k.g_ct = d->g.ctot; k.g_co = d->g.coff; k.Ho = d->Ho; k.Wo = d->Wo; k.Cout = d->Cout;
    k.ws = d->ws; k.KoutP = p.KoutP; k.KinVP = p.KinVP; k.cinp32 = p.cinp32;
    k.dbp = d->db ? d->ws + p.ws_floats : nullptr;
    k.tiles_x = p.tiles_x; k.tiles_y = p.tiles_y; k.tiles_total = p.tiles_total; k.tiles_per_split = p.tiles_per_split;
    k.nsplits = p.splits;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    switch (d->mode) {
        case TNR_CONV_3x3: rc = dispatch_wgrad<TNR_CONV_3x3>(k, p, s); break;
        case TNR_CONV_3x3_UP2: rc = dispatch_wgrad<TNR_CONV_3x3_UP2>(k, p, s); break;
        default: rc = dispatch_wgrad<TNR_CONV_4x4_S2>(k, p, s); break;
    }
    if (rc != TNR_OK) return rc;
    RedK r;
    r.ws = d->ws; r.dbp = k.dbp; r.splits = p.splits; r.ntaps = p.ntaps; r.KoutP = p.KoutP; r.KinVP = p.KinVP;
    r.cinp32 = p.cinp32; r.dw = d->dw; r.db = d->db; r.Cout = d->Cout; r.Cin = d->Cin; r.cin_total = d->cin_total;
    r.cin_begin = d->cin_begin; r.s2d = d->mode == TNR_CONV_4x4_S2;
    r.kh = r.s2d ? 4 : 3; r.kw = r.kh; r.alpha = d->alpha; r.beta = d->beta;
    const int64_t rows = (int64_t)p.ntaps * p.KoutP * (p.KinVP / 32);
    const int bias_blocks = d->db ? tnr_cdiv(d->Cout, 32) : 0;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(rows + bias_blocks)), dim3(256), 0, s, r);
    return tnr_check_launch("wgrad_reduce");
}
