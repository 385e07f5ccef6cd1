// Probe: per-workgroup timeline of conv_tile launches (s_memtime stamps: entry, first chunk in LDS, main loop
// done, epilogue issued).  Shows how much of a launch is prologue / epilogue / ragged tail.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DTNR_TIMELINE tools/probes/conv_timeline.hip -o tools/probes/conv_timeline
#include "../../trainner_amd/csrc/pack_api.hip"
#include "../../trainner_amd/csrc/conv_tile.hip"
#include "../../trainner_amd/csrc/conv_chain.hip"
#include <algorithm>
#include <cstdlib>
#include <vector>

static void run(int Cin, int Cout, int N, int H, int W) {
    const size_t px = (size_t)N * H * W;
    float *x, *y, *wp, *bias;
    const int KinP = (Cin + 15) / 16 * 16, KoutP = (Cout + 31) / 32 * 32;
    hipMalloc(&x, px * 192 * 4);
    hipMalloc(&y, px * KoutP * 4);
    hipMalloc(&wp, (size_t)9 * KoutP * KinP * 4);
    hipMalloc(&bias, KoutP * 4);
    hipMemset(x, 0, px * 192 * 4);
    hipMemset(wp, 0, (size_t)9 * KoutP * KinP * 4);
    hipMemset(bias, 0, KoutP * 4);
    tnr_conv_desc d = {};
    d.x.ptr = x; d.x.ctot = 192; d.x.coff = 0; d.N = N; d.H = H; d.W = W; d.Cin = Cin;
    d.wp = wp; d.KinP = KinP; d.KoutP = KoutP;
    d.y.ptr = y; d.y.ctot = KoutP; d.y.coff = 0; d.Ho = H; d.Wo = W; d.Cout = Cout;
    d.mode = TNR_CONV_3x3; d.bias = bias; d.act = 1; d.slope = 0.2f; d.alpha = 1.f;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) tnr_conv_forward(&d, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    if (tnr_conv_forward(&d, nullptr) != 0) { printf("error: %s\n", tnr_last_error()); return; }
    hipEventRecord(e1, nullptr);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(8 * 8 * 8192);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(tnr_timeline), h.size() * 8);
    const int tiles = (int)(px / (Cout > 32 ? 256 : 512));
    // per-XCD clocks may differ in offset: report per-workgroup deltas and, per XCD (b % 8), the span
    std::vector<double> pro, mainl, epi, tot, p_issue, e_bar1, e_tr, e_bar2, e_store;
    for (int b = 0; b < tiles && b < 8192; ++b) {
        const unsigned long long *t = &h[(size_t)b * 64];
        pro.push_back((double)(t[1] - t[0])); mainl.push_back((double)(t[2] - t[1])); epi.push_back((double)(t[3] - t[2]));
        tot.push_back((double)(t[3] - t[0]));
        p_issue.push_back((double)(t[4] - t[0])); e_bar1.push_back((double)(t[5] - t[2])); e_tr.push_back((double)(t[6] - t[5]));
        e_bar2.push_back((double)(t[7] - t[6])); e_store.push_back((double)(t[3] - t[7]));
    }
    auto stat = [](std::vector<double> v, const char *n) {
        std::sort(v.begin(), v.end());
        double s = 0; for (double q : v) s += q;
        printf("   %-9s mean %9.0f  min %9.0f  p50 %9.0f  max %9.0f ticks\n", n, s / v.size(), v.front(), v[v.size() / 2], v.back());
    };
    printf("conv %d->%d  N=%d %dx%d  tiles=%d  event time %.1f us\n", Cin, Cout, N, H, W, tiles, ms * 1e3);
    stat(pro, "prologue"); stat(p_issue, " p.issue"); stat(mainl, "main"); stat(epi, "epilogue");
    stat(e_bar1, " e.bar1"); stat(e_tr, " e.transp"); stat(e_bar2, " e.bar2"); stat(e_store, " e.store"); stat(tot, "total");
    hipFree(x); hipFree(y); hipFree(wp); hipFree(bias);
}

// dense-block chain: per-stage-pass breakdown (6 passes: conv1..4 and the two cout blocks of conv5)
static void run_chain(int N, int H, int W) {
    const size_t px = (size_t)N * H * W;
    const int nf = 64, gc = 32;
    float *buf, *out, *bias;
    hipMalloc(&buf, px * 192 * 4); hipMalloc(&out, px * 64 * 4); hipMalloc(&bias, 64 * 4);
    hipMemset(buf, 0, px * 192 * 4); hipMemset(bias, 0, 64 * 4);
    tnr_conv_desc d[5] = {};
    int fresh[5];
    float *wps[5];
    for (int k = 0; k < 5; ++k) {
        const int cin = nf + gc * k, cout = k < 4 ? gc : nf;
        hipMalloc(&wps[k], (size_t)9 * cout * cin * 4);
        hipMemset(wps[k], 0, (size_t)9 * cout * cin * 4);
        d[k].x.ptr = buf; d[k].x.ctot = 192; d[k].x.coff = 0; d[k].N = N; d[k].H = H; d[k].W = W; d[k].Cin = cin;
        d[k].wp = wps[k]; d[k].KinP = cin; d[k].KoutP = cout;
        d[k].y.ptr = k < 4 ? buf : out; d[k].y.ctot = k < 4 ? 192 : 64; d[k].y.coff = k < 4 ? cin : 0;
        d[k].Ho = H; d[k].Wo = W; d[k].Cout = cout; d[k].mode = TNR_CONV_3x3; d[k].bias = bias; d[k].act = k < 4 ? 1 : 0;
        d[k].slope = 0.2f; d[k].alpha = 1.f;
        d[k].mma = getenv("TNR_PROBE_MMA") ? atoi(getenv("TNR_PROBE_MMA")) : 0;     // 0 fp32 matrix core, 1 bf16 operands, 2 bf16x3
        fresh[k] = k ? cin - gc : -1;
    }
    const int64_t wsb = tnr_conv_chain_workspace_bytes(&d[0]);
    uint32_t *ws;
    hipMalloc(&ws, wsb); hipMemset(ws, 0, wsb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned epoch = 0;
    for (int i = 0; i < 3; ++i) tnr_conv_chain(d, fresh, 5, ws, wsb, ++epoch, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    if (tnr_conv_chain(d, fresh, 5, ws, wsb, ++epoch, nullptr) != 0) { printf("error: %s\n", tnr_last_error()); return; }
    hipEventRecord(e1, nullptr);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(8 * 8 * 8192);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(tnr_timeline), h.size() * 8);
    std::vector<unsigned long long> ph(8 * 8 * 8192);
    hipMemcpyFromSymbol(ph.data(), HIP_SYMBOL(tnr_phase), ph.size() * 8);
    const int tiles = (int)(px / 512);
    {
        const char *pn[7] = {"barriers A + B", "load wait (vmcnt)", "LDS refill", "drain (store acks)", "neighbour wait", "next-chunk load issue", "MFMA phase"};
        double tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int b = 0; b < tiles; ++b)
            for (int cc = 0; cc < 6; ++cc)
                for (int i = 0; i < 8; ++i) tot[i] += (double)ph[((size_t)b * 8 + cc) * 8 + i];
        printf("chain chunk-loop phases, wave 0, mean cycles per chunk over %.0f chunks/workgroup (4 launches accumulated):\n", tot[7] / tiles);
        for (int i = 0; i < 7; ++i) printf("   %-24s %9.0f\n", pn[i], tot[i] / tot[7]);
    }
    printf("chain 5 stages N=%d %dx%d tiles=%d event time %.1f us (ticks/us from workgroup 0: %.0f)\n", N, H, W, tiles, ms * 1e3,
           (double)(h[5 * 8 + 3] - h[0]) / (ms * 1e3));
    const char *names[6] = {"conv1 64->32", "conv2 96->32", "conv3 128->32", "conv4 160->32", "conv5 cb0", "conv5 cb1"};
    for (int c = 0; c < 6; ++c) {
        double pro = 0, mainl = 0, epi = 0, gap = 0;
        for (int b = 0; b < tiles; ++b) {
            const unsigned long long *t = &h[((size_t)b * 8 + c) * 8];
            pro += (double)(t[1] - t[0]); mainl += (double)(t[2] - t[1]); epi += (double)(t[3] - t[2]);
            if (c > 0) gap += (double)(t[0] - h[((size_t)b * 8 + c - 1) * 8 + 3]);
        }
        printf("   %-14s entry->first chunk in LDS %7.0f   main loop %8.0f   epilogue %7.0f   gap before %6.0f ticks (mean over workgroups)\n",
               names[c], pro / tiles, mainl / tiles, epi / tiles, gap / tiles);
    }
}

int main() {
    run_chain(16, 128, 128);
    if (getenv("TNR_PROBE_MMA")) return 0;
    run(160, 32, 16, 128, 128);
    run(64, 32, 16, 128, 128);
    run(192, 64, 16, 128, 128);
    return 0;
}
