// Probe: what a workgroup barrier per "slot" of MFMAs costs with one and with two waves per SIMD (the conv_sweep.hip question).
// A slot = 108 MFMAs per SIMD (9 units x 6 x 2): one wave does all 108 (six accumulators x 2 M-tiles), or two waves 54 each.
// Variants: barrier only; barrier + an LDS fragment read the first MFMA depends on; + fragment reads (ds_read_b128) in front of every unit.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_slots.hip -o tools/probes/mfma_slots
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// WPS waves per SIMD; READS: 0 none, 1 one dependent fragment read per slot, 2 three b128 reads per 6 MFMAs (a unit's B planes)
template <int WPS, int READS, bool BAR>
__global__ void __launch_bounds__(256 * WPS, 1) k(float *out, int slots) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    constexpr int NACC = 12 / WPS;
    f32x16 acc[NACC];
    for (int c = 0; c < NACC; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)(i & 7);
    bf16x8 a, b[3];
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(float)(threadIdx.x + i);
    for (int p = 0; p < 3; ++p) b[p] = a;
    __syncthreads();
    const float *src = lds + (threadIdx.x & 63) * 24;
    for (int s = 0; s < slots; ++s) {
        if (BAR) __syncthreads();
        if (READS == 1) b[0] = *reinterpret_cast<const bf16x8 *>(src + (s & 7) * 1536);
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            if (READS == 2) {
#pragma unroll
                for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8 *>(src + ((u + s) & 7) * 1536 + 8 * p);
            }
#pragma unroll
            for (int c = 0; c < NACC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[c % 3], acc[c], 0, 0, 0);
        }
    }
    float sum = 0.f;
    for (int c = 0; c < NACC; ++c) sum += acc[c][0] + acc[c][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int WPS, int READS, bool BAR>
void run() {
    float *out;
    const int threads = 256 * WPS, blocks = 256, slots = 2000;
    (void)hipMalloc(&out, blocks * threads * 4);
    hipLaunchKernelGGL((k<WPS, READS, BAR>), dim3(blocks), dim3(threads), 0, 0, out, slots);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, nullptr);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<WPS, READS, BAR>), dim3(blocks), dim3(threads), 0, 0, out, slots);
    (void)hipEventRecord(e1, nullptr);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = 5.0 * blocks * 4 * (double)slots * 108 * 32768.0;
    printf("waves/SIMD %d, reads %d, barrier %d: %.0f TFLOP/s (bf16 dense)\n", WPS, READS, (int)BAR, flop / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
}
int main() {
    run<1, 0, false>(); run<2, 0, false>(); run<1, 0, true>(); run<2, 0, true>();
    run<1, 1, true>(); run<2, 1, true>(); run<1, 2, true>(); run<2, 2, true>(); run<1, 2, false>(); run<2, 2, false>();
    return 0;
}
