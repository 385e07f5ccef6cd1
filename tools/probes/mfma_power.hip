// Probe: what rate of bf16 MFMAs does the part SUSTAIN (about a second each) by instruction shape and by operand content?  Four waves per CU x
// 256 CUs, independent accumulator chains, operands in registers only (no LDS, no memory): v_mfma_f32_32x32x16_bf16 and v_mfma_f32_16x16x32_bf16,
// on (a) the same small-integer operands in every instruction, (b) four rotating operand pairs of random bf16 values.  The split arithmetic
// (TNR_MMA_BF16X3) feeds the matrix core white noise in its mid / lo planes: case (b) is its power regime.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_power.hip -o tools/probes/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ unsigned hash32(unsigned h) { h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16; return h; }

template <int SHAPE, int RANDOM>      // SHAPE 0: 32x32x16 (4 chains of 16 accumulators), 1: 16x16x32 (8 chains of 4)
__global__ void __launch_bounds__(256) k(float *out, int iters) {
    bf16x8 a[4], b[4];
    for (int s = 0; s < 4; ++s)
        for (int i = 0; i < 8; ++i) {
            if (RANDOM) {
                const unsigned h = hash32((threadIdx.x * 64u + blockIdx.x * 16384u + s * 16u + i) * 2654435761u);
                a[s][i] = (__bf16)(((int)(h & 0xFFFF) - 32768) * (1.0f / 32768.0f));
                b[s][i] = (__bf16)(((int)(h >> 16) - 32768) * (1.0f / 32768.0f));
            } else {
                a[s][i] = (__bf16)(float)(threadIdx.x & 7);
                b[s][i] = (__bf16)(float)((threadIdx.x * 3) & 7);
            }
        }
    float s_out = 0.f;
    if (SHAPE == 0) {
        f32x16 acc[4];
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(rep + c) & 3], b[(rep * 3 + c) & 3], acc[c], 0, 0, 0);
        }
        for (int c = 0; c < 4; ++c) s_out += acc[c][0] + acc[c][15];
    } else {
        f32x4 acc[8];
        for (int c = 0; c < 8; ++c)
            for (int r = 0; r < 4; ++r) acc[c][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(rep + c) & 3], b[(rep * 3 + c) & 3], acc[c], 0, 0, 0);
        }
        for (int c = 0; c < 8; ++c) s_out += acc[c][0] + acc[c][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s_out;
}

template <int SHAPE, int RANDOM>
void run(const char *name) {
    float *out;
    const int threads = 256, blocks = 256;
    (void)hipMalloc(&out, blocks * threads * 4);
    const int per_iter = SHAPE == 0 ? 32 : 64;                // MFMAs per wave and loop iteration
    const double flop_per = SHAPE == 0 ? 32768.0 : 16384.0;   // 2 x M x N x K
    const int iters = 200000 / per_iter;                      // ~ 200 k MFMAs per wave and launch
    hipLaunchKernelGGL((k<SHAPE, RANDOM>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 300;
    (void)hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<SHAPE, RANDOM>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1, nullptr);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)reps * blocks * 4 * (double)iters * per_iter * flop_per;
    printf("%-52s %7.0f ms  %7.0f TFLOP/s bf16 dense = %5.3f of 2516.6 (fp32-equivalent / 6: %5.1f)\n", name, ms, flop / (ms * 1e-3) / 1e12,
           flop / (ms * 1e-3) / 1e12 / 2516.6, flop / (ms * 1e-3) / 1e12 / 6.0);
    fflush(stdout);
    (void)hipFree(out);
}
int main() {
    run<0, 0>("32x32x16, constant small-integer operands");
    run<0, 1>("32x32x16, four rotating pairs of random operands");
    run<1, 0>("16x16x32, constant small-integer operands");
    run<1, 1>("16x16x32, four rotating pairs of random operands");
    run<0, 1>("32x32x16, random operands (again)");
    return 0;
}
