// Probe: v_mfma_f32_32x32x16_bf16 issue rate against the number of INDEPENDENT accumulator chains per wave and the waves per SIMD.
// Each wave runs 4096 MFMAs round-robin over NCH accumulators (chain length 4096 / NCH each); reports cycles per MFMA per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_chain.hip -o tools/probes/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NCH>
__global__ void __launch_bounds__(512) k(float *out, unsigned long long *cyc, int iters) {
    f32x16 acc[NCH];
    for (int c = 0; c < NCH; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NCH>
void run(int waves_per_simd) {
    float *out; unsigned long long *cyc;
    const int threads = 256 * waves_per_simd, blocks = 256;
    (void)hipMalloc(&out, blocks * threads * 4); (void)hipMalloc(&cyc, blocks * 8);
    const int iters = 4096 / (8 * NCH);
    hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, nullptr);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters * 16);
    (void)hipEventRecord(e1, nullptr);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = 20.0 * blocks * (threads / 64) * (double)(iters * 16 * 8 * NCH) * 32768.0;
    const double tf = flop / (ms * 1e-3) / 1e12;
    hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    unsigned long long h[256];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += (double)h[i];
    m /= 256;
    const double per_wave = m / (iters * 8 * NCH);
    printf("chains %d, waves/SIMD %d: %.1f cycles per MFMA per wave -> %.1f cycles per MFMA per SIMD; wall clock: %.0f TFLOP/s (bf16 dense)\n", NCH,
           waves_per_simd, per_wave, per_wave / waves_per_simd, tf);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    for (int w = 1; w <= 2; ++w) { run<1>(w); run<2>(w); run<3>(w); run<4>(w); run<6>(w); }
    return 0;
}
