// Probe: semantics of ds_read_b64_tr_b16 (gfx950 LDS transpose read).  LDS holds u16 values equal to their own index; lane l reads
// 8 bytes at byte address base(l); the probe prints, per lane, the four 16-bit values it received, for two address patterns.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/tr_read.hip -o /tmp/tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(unsigned short *out, int pattern) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // pattern 0: lane l -> element offset 4 l (contiguous 8-byte pieces)
    // pattern 1: lane l -> row (l & 15) of a [16][64] u16 image, 4-element column group (l >> 4): offset (l & 15) * 64 + (l >> 4) * 4
    int off = pattern == 0 ? 4 * l : (l & 15) * 64 + (l >> 4) * 4;
    unsigned addr = (unsigned)(size_t)(lds) + 2u * off;   // LDS byte address (low 32 bits of the generic pointer are the LDS offset)
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[4 * l + 0] = (unsigned short)(r.x & 0xffff);
    out[4 * l + 1] = (unsigned short)(r.x >> 16);
    out[4 * l + 2] = (unsigned short)(r.y & 0xffff);
    out[4 * l + 3] = (unsigned short)(r.y >> 16);
}

int main() {
    unsigned short *d;
    hipMalloc(&d, 256 * 2);
    for (int p = 0; p < 2; ++p) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, p);
        std::vector<unsigned short> h(256);
        hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        printf("pattern %d (lane: the four u16 it received = LDS element indices)\n", p);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], (l & 3) == 3 ? "\n" : "");
    }
    return 0;
}
