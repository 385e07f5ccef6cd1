// Probe: occupy ONE compute unit (a workgroup that owns most of its LDS and spins for `ms` milliseconds) so that a kernel which wants all
// 256 CUs resident finds 255.  Built as a shared library; tools/probes/sweep_hog.py launches it on a side stream next to tnr_conv_sweep.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/cu_hog.hip -o tools/probes/libcu_hog.so
#include <hip/hip_runtime.h>
__global__ void __launch_bounds__(256) hog_kernel(unsigned long long cycles, int *out) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = 1.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) out[0] = (int)lds[1];
}
extern "C" int cu_hog(void *stream, double ms, int *out, int blocks) {
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        done = true;
    }
    hipLaunchKernelGGL(hog_kernel, dim3(blocks), dim3(256), 100 * 1024, (hipStream_t)stream, (unsigned long long)(ms * 2.0e6), out);   // s_memtime counts shader-clock cycles (~2 GHz)
    return (int)hipGetLastError();
}
