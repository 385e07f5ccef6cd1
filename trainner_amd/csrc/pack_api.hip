// Error plumbing + weight re-layout (OIHW state_dict tensors -> K-chunked [tap][kout][kin] slabs).
#include <stdarg.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void tnr_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int tnr_check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        tnr_set_error("%s: %s", what, hipGetErrorString(e));
        return TNR_ELAUNCH;
    }
    return TNR_OK;
}

extern "C" const char *tnr_last_error(void) { return g_err; }

static unsigned *g_fault_word = nullptr;
unsigned *tnr_fault_word_or(unsigned *fallback) { return g_fault_word ? g_fault_word : fallback; }
extern "C" int tnr_set_fault_word(uint32_t *dev_word) {
    g_fault_word = dev_word;
    return TNR_OK;
}
extern "C" int tnr_version(void) { return TNR_ABI_VERSION; }

extern "C" int tnr_pack_dims(int32_t Cout, int32_t Cin, int32_t kh, int32_t kw, int32_t kind, int32_t *KoutP,
                             int32_t *KinP, int64_t *n_out) {
    int ko, ki;
    int64_t n;
    switch (kind) {
        case TNR_PACK_FWD:
            ko = tnr_round_up(Cout, 32); ki = tnr_round_up(Cin, TNR_CK);
            n = (int64_t)kh * kw * ko * ki;
            break;
        case TNR_PACK_DGRAD_3x3:
            TNR_REQUIRE(kh == 3 && kw == 3, "pack dgrad3x3: kernel must be 3x3");
            ko = tnr_round_up(Cin, 32); ki = tnr_round_up(Cout, TNR_CK);
            n = (int64_t)9 * ko * ki;
            break;
        case TNR_PACK_FWD_S2D:
            TNR_REQUIRE(kh == 4 && kw == 4, "pack s2d: kernel must be 4x4");
            ko = tnr_round_up(Cout, 32); ki = tnr_round_up(Cin, TNR_CK);
            n = (int64_t)4 * ko * 4 * ki;
            break;
        case TNR_PACK_DGRAD_S2:
            TNR_REQUIRE(kh == 4 && kw == 4, "pack dgrad s2: kernel must be 4x4");
            ko = tnr_round_up(Cin, 32); ki = tnr_round_up(Cout, TNR_CK);
            n = (int64_t)16 * ko * ki;
            break;
        case TNR_PACK_COL_FWD:
            ko = tnr_round_up(Cout, 32); ki = tnr_round_up(kh * kw * Cin, TNR_CK);
            n = (int64_t)ko * ki;
            break;
        case TNR_PACK_COL_DGRAD3:
            TNR_REQUIRE(kh == kw && kh >= 1, "pack col dgrad: kernel must be square");
            ko = tnr_round_up(Cin, 32); ki = tnr_round_up(kh * kw * Cout, TNR_CK);
            n = (int64_t)ko * ki;
            break;
        case TNR_PACK_C4_FWD:
            TNR_REQUIRE(kh == kw && (kh == 3 || kh == 7) && Cin <= 4, "pack c4: 3x3 or 7x7 kernel with Cin <= 4");
            ko = tnr_round_up(Cout, 32); ki = tnr_round_up(4 * kh * kw, TNR_CK);      // 48 | 208
            n = (int64_t)ko * ki;
            break;
        case TNR_PACK_C4_DGRAD3:
            TNR_REQUIRE(kh == kw && (kh == 3 || kh == 7) && Cout <= 4, "pack c4 dgrad: 3x3 or 7x7 kernel with Cout <= 4");
            ko = tnr_round_up(Cin, 32); ki = tnr_round_up(4 * kh * kw, TNR_CK);
            n = (int64_t)ko * ki;
            break;
        default:
            tnr_set_error("pack: bad kind %d", kind);
            return TNR_EINVAL;
    }
    if (KoutP) *KoutP = ko;
    if (KinP) *KinP = ki;
    if (n_out) *n_out = n;
    return TNR_OK;
}

namespace {
__global__ void pack_weights_kernel(const tnr_pack_item *items) {
    const tnr_pack_item it = items[blockIdx.y];
    const int Cout = it.Cout, Cin = it.Cin, kh = it.kh, kw = it.kw, ko = it.KoutP, ki = it.KinP;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < it.n_out; e += (int64_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (it.kind == TNR_PACK_FWD) {
            const int ci = (int)(e % ki);
            int64_t q = e / ki;
            const int co = (int)(q % ko);
            const int t = (int)(q / ko);
            if (co < Cout && ci < Cin) v = it.w[((size_t)co * Cin + ci) * kh * kw + t];
        } else if (it.kind == TNR_PACK_DGRAD_3x3) {
            const int co = (int)(e % ki);  // reduction channel of the dgrad conv = forward cout
            int64_t q = e / ki;
            const int ci = (int)(q % ko);  // produced channel = forward cin
            const int t = (int)(q / ko);
            const int ky = 2 - t / 3, kx = 2 - t % 3;
            if (co < Cout && ci < Cin) v = it.w[(((size_t)co * Cin + ci) * 3 + ky) * 3 + kx];
        } else if (it.kind == TNR_PACK_FWD_S2D) {
            const int vch = (int)(e % (4 * ki));
            int64_t q = e / (4 * ki);
            const int co = (int)(q % ko);
            const int t = (int)(q / ko);
            const int pp = vch / ki, ci = vch - pp * ki;
            const int ky = 2 * (t >> 1) + (pp >> 1), kx = 2 * (t & 1) + (pp & 1);
            if (co < Cout && ci < Cin) v = it.w[(((size_t)co * Cin + ci) * 4 + ky) * 4 + kx];
        } else if (it.kind == TNR_PACK_COL_FWD) {      // [co][t*Cin + ci]
            const int v_ = (int)(e % ki);
            const int co = (int)(e / ki);
            const int t = v_ / Cin, ci = v_ - t * Cin;
            if (co < Cout && t < kh * kw) v = it.w[((size_t)co * Cin + ci) * kh * kw + t];
        } else if (it.kind == TNR_PACK_COL_DGRAD3) {   // [ci][t*Cout + co], taps flipped
            const int v_ = (int)(e % ki);
            const int ci = (int)(e / ki);
            const int t = v_ / Cout, co = v_ - t * Cout;
            if (ci < Cin && t < kh * kw) v = it.w[(((size_t)co * Cin + ci) * kh + (kh - 1 - t / kw)) * kw + (kw - 1 - t % kw)];
        } else if (it.kind == TNR_PACK_C4_FWD) {       // [co][4*t + ci]
            const int v_ = (int)(e % ki);
            const int co = (int)(e / ki);
            const int t = v_ >> 2, ci = v_ & 3;
            if (co < Cout && t < kh * kw && ci < Cin) v = it.w[((size_t)co * Cin + ci) * kh * kw + t];
        } else if (it.kind == TNR_PACK_C4_DGRAD3) {    // [ci][4*t + co], taps flipped
            const int v_ = (int)(e % ki);
            const int ci = (int)(e / ki);
            const int t = v_ >> 2, co = v_ & 3;
            if (ci < Cin && t < kh * kw && co < Cout) v = it.w[(((size_t)co * Cin + ci) * kh + (kh - 1 - t / kw)) * kw + (kw - 1 - t % kw)];
        } else {  // TNR_PACK_DGRAD_S2: [par][t][ci][co]
            const int co = (int)(e % ki);
            int64_t q = e / ki;
            const int ci = (int)(q % ko);
            q /= ko;
            const int t = (int)(q & 3);
            const int par = (int)(q >> 2);
            const int py = par >> 1, px = par & 1;
            const int ky = ((py + 1) & 1) + 2 * (t >> 1), kx = ((px + 1) & 1) + 2 * (t & 1);
            if (co < Cout && ci < Cin) v = it.w[(((size_t)co * Cin + ci) * 4 + ky) * 4 + kx];
        }
        it.wp[e] = v;
    }
}
}  // namespace

namespace {
__global__ void pack_dense_kernel(const tnr_dense_pack_item *items) {
    const tnr_dense_pack_item it = items[blockIdx.y];
    const int nf = it.nf, gc = it.gc, t = it.t, ko = it.KoutP, ki = it.KinP;
    const int ntarget = (t == 4) ? nf : gc;
    const int tlo = (t == 4) ? 0 : nf + (3 - t) * gc;   // first forward-input channel of the target group
    const int K = nf + t * gc;                          // valid reduction channels
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < it.n_out; e += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(e % ki);
        int64_t q = e / ki;
        const int cl = (int)(q % ko);
        const int tap = (int)(q / ko);
        float v = 0.f;
        if (cl < ntarget && r < K) {
            const int ky = 2 - tap / 3, kx = 2 - tap % 3;
            const int c = tlo + cl;                      // channel of the forward conv's input
            if (r < nf) {                                // conv5 (nf outputs, nf + 4 gc inputs)
                const int cin = nf + 4 * gc;
                v = it.scale5 * it.w[4][(((size_t)r * cin + c) * 3 + ky) * 3 + kx];
            } else {
                const int m = (r - nf) / gc;             // 0 -> conv4, 1 -> conv3, ...
                const int ro = (r - nf) - m * gc;
                const int k = 3 - m;                     // index into w[] (conv_{k+1})
                const int cin = nf + k * gc;
                v = it.w[k][(((size_t)ro * cin + c) * 3 + ky) * 3 + kx];
            }
        }
        it.wp[e] = v;
    }
}
}  // namespace

extern "C" int tnr_pack_dense_dims(int32_t nf, int32_t gc, int32_t t, int32_t *KoutP, int32_t *KinP, int64_t *n_out) {
    TNR_REQUIRE(nf > 0 && gc > 0 && t >= 0 && t <= 4 && (nf % 4) == 0 && (gc % 4) == 0, "pack_dense: bad arguments");
    const int ko = tnr_round_up(t == 4 ? nf : gc, 32), ki = tnr_round_up(nf + t * gc, TNR_CK);
    if (KoutP) *KoutP = ko;
    if (KinP) *KinP = ki;
    if (n_out) *n_out = (int64_t)9 * ko * ki;
    return TNR_OK;
}

extern "C" int tnr_pack_dense_dgrad(const tnr_dense_pack_item *items_dev, int32_t n, int64_t max_out, void *stream) {
    TNR_REQUIRE(items_dev != nullptr && n > 0 && max_out > 0, "pack_dense: bad arguments");
    int64_t bx = tnr_cdiv64(max_out, 256 * 4);
    if (bx > 1024) bx = 1024;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(pack_dense_kernel, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, items_dev);
    return tnr_check_launch("pack_dense_dgrad");
}

extern "C" int tnr_pack_weights(const tnr_pack_item *items_dev, int32_t n, int64_t max_out, void *stream) {
    TNR_REQUIRE(items_dev != nullptr && n > 0 && max_out > 0, "pack: bad arguments");
    int64_t bx = tnr_cdiv64(max_out, 256 * 4);
    if (bx > 1024) bx = 1024;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, items_dev);
    return tnr_check_launch("pack_weights");
}
