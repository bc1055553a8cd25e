// What does s_memtime count on MI355X?  256 workgroups (one per CU, ONE round) each run until their own s_memtime has advanced by X ticks --
// spinning on s_sleep, or issuing back-to-back MFMAs on every SIMD (power-capped clock) -- and the host times the launch with HIP events:
// ticks / wall = the counter's frequency in each regime.  Also: MFMAs retired per tick under load.
// hipcc --offload-arch=gfx950 -O3 tools/probe/memtime_rate.hip -o tools/probe/memtime_rate && ./tools/probe/memtime_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(256) void spin_kernel(unsigned long long* out, unsigned long long ticks, int heavy) {
    bf16x8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = (short)(0x3f80 + threadIdx.x + e); b[e] = (short)(0x3f00 + e); }
    f32x16 c[4];
    for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) c[i][e] = (float)(i + e);
    unsigned long long t0, t1, r0, r1, n = 0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    do {
        if (heavy) {
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int i = 0; i < 4; i++) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[i], 0, 0, 0);
            n += 32;
        } else {
            __builtin_amdgcn_s_sleep(32);
        }
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    } while (t1 - t0 < ticks);
    float s = 0; for (int i = 0; i < 4; i++) s += c[i][0];
    if (s == 1.2345f) out[7] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = n; }
}

int main() {
    unsigned long long* d; hipMalloc(&d, 64); unsigned long long h[3];
    const unsigned long long X = 20000000ull;
    for (int heavy = 0; heavy < 2; heavy++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        spin_kernel<<<256, 256>>>(d, X / 10, heavy); hipDeviceSynchronize();
        hipEventRecord(e0); spin_kernel<<<256, 256>>>(d, X, heavy); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("%s: %llu s_memtime ticks / %llu s_memrealtime ticks in %.3f ms wall -> s_memtime %.1f MHz, s_memrealtime %.1f MHz", heavy ? "MFMAs on every SIMD" : "s_sleep spin", h[0], h[1], ms, h[0] / ms / 1e3, h[1] / ms / 1e3);
        if (heavy) printf("; %.2f ticks per MFMA = %.1f ns (32 shader cycles -> %.2f GHz shader clock)", (double)h[0] / h[2], ms * 1e6 / h[2], 32.0 / (ms * 1e6 / h[2]));
        printf("\n");
    }
    return 0;
}
