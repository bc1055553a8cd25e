// Does a wave's issue port block when it has several LDS-DMA requests outstanding?  One wave (or all eight of a workgroup) issues a burst of 8
// global_load_lds_dwordx4 pieces (1 KiB each, L2-resident source) back to back with an s_memtime stamp after every one, then waits for all; the
// deltas say what each request cost THE ISSUING WAVE.  Same burst with one cheap VALU instruction between the requests, and with plain
// global_load_dwordx4 into VGPRs for comparison.  Round 4 (profiles/r04_gemm_fr.md section 3: the path's rate does not depend on the ring depth).
// hipcc --offload-arch=gfx950 -O3 -w tools/probe/lds_dma_burst.hip -o tools/probe/lds_dma_burst && ./tools/probe/lds_dma_burst
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))
#define STAMP(x) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x) :: "memory")

template <int MODE>   // 0 LDS-DMA, 1 VGPR loads
__global__ __launch_bounds__(512) void burst_kernel(const unsigned char* src, int64_t ld, unsigned long long* out, int rounds, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned char* ring = lds + w * 8 * 1024;
    const unsigned char* base = src + ((int64_t)blockIdx.x * 64 + w * 8) * 8 * ld + (lane >> 3) * ld + (lane & 7) * 16;
    unsigned long long t[10], acc[10] = {};
    uint4 v[8]; uint4 x = make_uint4(0, 0, 0, 0);
    for (int r = 0; r < rounds; r++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        STAMP(t[0]);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const unsigned char* g = base + (int64_t)i * 8 * ld + (r & 3) * 128;
            if (MODE == 0) __builtin_amdgcn_global_load_lds(GPTR(g), LPTR(ring + i * 1024), 16, 0, 0);
            else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[i]) : "v"(g) : "memory");
            STAMP(t[i + 1]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(t[9]);
        if (MODE == 1) for (int i = 0; i < 8; i++) { x.x ^= v[i].x; x.y ^= v[i].w; }
        if (r >= rounds / 2) for (int i = 0; i < 9; i++) acc[i] += t[i + 1] - t[i];
    }
    if ((x.x ^ x.y) == 0x1234567u) sink[0] = 1;
    if (lane == 0 && blockIdx.x == 0) for (int i = 0; i < 9; i++) out[w * 9 + i] = acc[i] / (rounds - rounds / 2);
}

template <int MODE>
static void run(const char* what, const unsigned char* src, unsigned long long* d, unsigned* sink, int waves, int wgs) {
    hipFuncSetAttribute((const void*)burst_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    burst_kernel<MODE><<<wgs, waves * 64, 64 * 1024>>>(src, 1536, d, 400, sink);
    hipDeviceSynchronize();
    unsigned long long h[72]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-40s %d wave(s)/workgroup, %3d workgroups | wave 0: ticks per request", what, waves, wgs);
    for (int i = 0; i < 8; i++) printf(" %4llu", h[i]);
    printf(" | drain %5llu", h[8]);
    if (waves > 1) { printf(" || wave %d:", waves - 1); for (int i = 0; i < 8; i++) printf(" %4llu", h[(waves - 1) * 9 + i]); printf(" | %5llu", h[(waves - 1) * 9 + 8]); }
    printf("\n");
}

int main() {
    unsigned char* src; hipMalloc(&src, (size_t)256 * 64 * 8 * 1536 + (1 << 20)); hipMemset(src, 1, (size_t)256 * 64 * 8 * 1536 + (1 << 20));
    unsigned long long* d; hipMalloc(&d, 72 * 8); unsigned* sink; hipMalloc(&sink, 64);
    printf("s_memtime ticks (100 MHz-class constant-rate counter x ~24: compare columns, not absolute)\n");
    run<0>("LDS-DMA burst of 8", src, d, sink, 1, 1);
    run<0>("LDS-DMA burst of 8", src, d, sink, 8, 1);
    run<0>("LDS-DMA burst of 8", src, d, sink, 8, 256);
    run<1>("global_load_dwordx4 -> VGPR burst of 8", src, d, sink, 1, 1);
    run<1>("global_load_dwordx4 -> VGPR burst of 8", src, d, sink, 8, 1);
    run<1>("global_load_dwordx4 -> VGPR burst of 8", src, d, sink, 8, 256);
    return 0;
}
