// Probe: cost and correctness of tile-level producer/consumer hand-off between co-resident workgroups
// (agent-scope release -> flag -> acquire), the building block a persistent dense-block kernel would need.
// 512 workgroups (2 per CU), each owns a 64 KiB tile; per iteration it waits for its 8 ring neighbours'
// flags, reads 8 KiB from each neighbour tile (checks the values), then writes its own tile and publishes.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/flag_sync.hip -o tools/probes/flag_sync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int TILE_F = 16384;   // floats per tile (64 KiB)
constexpr int NB = 8;

__device__ __forceinline__ int nbr(int t, int k, int ntiles) {   // symmetric neighbour set: +-1, +-37, +-74, +-111
    const int d = (k >> 1) == 0 ? 1 : 37 * (k >> 1);
    return (t + ((k & 1) ? ntiles - d : d)) % ntiles;
}
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_sc(const float *p) {
    f4v v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 load_sc1(const float *p) {      // agent scope only
    f4v v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store_sc1(float *p, float4 q) {
    f4v v = {q.x, q.y, q.z, q.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_sc(float *p, float4 q) {
    f4v v = {q.x, q.y, q.z, q.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}

template <int MODE>   // 0: fences + flags, 1: flags only (no fences), 2: no sync (floor), 3: flags + sc0 sc1 loads/stores
__global__ void __launch_bounds__(256, 2) ring(float *buf0, float *buf1, unsigned *flags, unsigned *errs, int iters, int ntiles) {
    extern __shared__ float smem[];
    const int t = blockIdx.x, tid = threadIdx.x;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        float *src = (it & 1) ? buf1 : buf0;
        float *dst = (it & 1) ? buf0 : buf1;
        if (MODE != 2 && it > 0) {   // modes: 0 fences, 1 none, 3 sc0 sc1, 4 sc1, 5 wbl2
            if (tid < NB) {
                // spread neighbours over the grid: other CUs and other XCDs
                const int nb = nbr(t, tid, ntiles);
                int spins = 0;
                while (__hip_atomic_load(&flags[nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1 << 22)) { atomicAdd(&errs[1], 1u); break; }
                }
            }
            if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __syncthreads();
        }
        // read 8 KiB of each neighbour's tile from the previous iteration
        float acc = 0.f;
        if (it > 0) {
            for (int k = 0; k < NB; ++k) {
                const int nb = nbr(t, k, ntiles);
                const float *ap = src + (size_t)nb * TILE_F + ((tid + it * 7 + k * 13) & 511) * 4 * 8;
                const float4 v = (MODE == 3) ? load_sc(ap) : (MODE == 4 || MODE == 5) ? load_sc1(ap) : *reinterpret_cast<const float4 *>(ap);
                if (v.x != (float)it || v.w != (float)it) ++bad;
                acc += v.x;
            }
        }
        smem[tid] = acc;
        // write the own tile: 16 float4 per thread
        for (int k = 0; k < 16; ++k) {
            const float f = (float)(it + 1);
            float *sp = dst + (size_t)t * TILE_F + (k * 256 + tid) * 4;
            if (MODE == 3) store_sc(sp, make_float4(f, f, f, f));
            else if (MODE == 4) store_sc1(sp, make_float4(f, f, f, f));
            else *reinterpret_cast<float4 *>(sp) = make_float4(f, f, f, f);
        }
        if (MODE != 2) {
            if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (MODE == 3 || MODE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (MODE == 5) asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");   // plain stores + L2 write-back
            __syncthreads();
            if (tid == 0) __hip_atomic_store(&flags[t], (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (bad) atomicAdd(&errs[0], bad);
}

template <int MODE>
static void run(const char *name, int iters) {
    const int ntiles = 512;
    float *b0, *b1;
    unsigned *flags, *errs;
    hipMalloc(&b0, (size_t)ntiles * TILE_F * 4);
    hipMalloc(&b1, (size_t)ntiles * TILE_F * 4);
    hipMalloc(&flags, ntiles * 4);
    hipMalloc(&errs, 8);
    hipMemset(flags, 0, ntiles * 4);
    hipMemset(errs, 0, 8);
    hipMemset(b0, 0, (size_t)ntiles * TILE_F * 4);
    hipFuncSetAttribute((const void *)ring<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL(ring<MODE>, dim3(ntiles), dim3(256), 72 * 1024, 0, b0, b1, flags, errs, iters, ntiles);
    hipEventRecord(e1, nullptr);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned h[2];
    hipMemcpy(h, errs, 8, hipMemcpyDeviceToHost);
    printf("%-28s %d iters: %8.2f us/iter   stale reads %u   spin timeouts %u\n", name, iters, ms * 1e3 / iters, h[0], h[1]);
    hipFree(b0); hipFree(b1); hipFree(flags); hipFree(errs);
}

int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<2>("no sync (floor)", 200);
        run<1>("flags, no fences", 200);
        run<0>("flags + release/acquire", 200);
        run<3>("flags + sc0 sc1 ld/st", 200);
        run<4>("flags + sc1 ld/st (agent)", 200);
        run<5>("plain st + wbl2 sc1, sc1 ld", 200);
    }
    return 0;
}
