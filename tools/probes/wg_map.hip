// Probe: which CU / XCD does workgroup b of a 2-per-CU (LDS-limited) launch land on?
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/wg_map.hip -o gpurun_out/wg_map ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256, 2) probe(unsigned *out, int spin) {
    extern __shared__ float smem[];
    unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID, offset 0, size 32
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
    for (int i = 0; i < spin; ++i) { smem[threadIdx.x] = acc; __syncthreads(); acc += smem[(threadIdx.x + 1) & 255]; }
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = hw;
        out[blockIdx.x * 4 + 1] = xcc;
        out[blockIdx.x * 4 + 2] = (unsigned)(t0 & 0xffffffffu);
        out[blockIdx.x * 4 + 3] = (unsigned)acc;
    }
}
int main() {
    const int nb = 1024;
    unsigned *d;
    hipMalloc(&d, nb * 16);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 72 * 1024, 0, d, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 4);
    hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
    for (int b = 0; b < nb; ++b) {
        unsigned hw = h[b * 4], xcc = h[b * 4 + 1] & 15;
        printf("%d xcc=%u se=%u sh=%u cu=%u simd=%u wave=%u t=%u\n", b, xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15,
               (hw >> 4) & 3, hw & 15, h[b * 4 + 2]);
    }
    return 0;
}
