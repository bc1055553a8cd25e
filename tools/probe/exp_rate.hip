#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = 0.001f * (threadIdx.x + i);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) a[i] = __builtin_amdgcn_exp2f(a[i]);
            else if (MODE == 1) a[i] = a[i] + 1.5f;
            else if (MODE == 2) a[i] = __builtin_amdgcn_rcpf(a[i]);
            else if (MODE == 3) { a[i] = __builtin_amdgcn_exp2f(a[i]); a[i] = a[i] * 0.5f; }       // exp + 1 plain
            else if (MODE == 4) { a[i] = __builtin_amdgcn_exp2f(a[i]); a[i] = a[i] * 0.5f; a[i] = a[i] + 0.25f; a[i] = a[i] * 1.01f; }   // exp + 3 plain
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 64);
    const int iters = 1000;
    const char* names[] = {"v_exp_f32", "v_add_f32", "v_rcp_f32", "exp + 1 mul", "exp + 3 plain"};
    for (int waves = 1; waves <= 3; waves++)       // waves per SIMD: block of 256 threads = 1 wave per SIMD; launch `waves` blocks per CU
        for (int m = 0; m < 5; m++) {
            unsigned long long h = 0;
            for (int rep = 0; rep < 2; rep++) {
                if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256 * waves), dim3(256), 0, 0, out, cyc, iters);
                if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256 * waves), dim3(256), 0, 0, out, cyc, iters);
                if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256 * waves), dim3(256), 0, 0, out, cyc, iters);
                if (m == 3) hipLaunchKernelGGL(k<3>, dim3(256 * waves), dim3(256), 0, 0, out, cyc, iters);
                if (m == 4) hipLaunchKernelGGL(k<4>, dim3(256 * waves), dim3(256), 0, 0, out, cyc, iters);
                hipDeviceSynchronize();
            }
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            int per = (m <= 2) ? 8 : (m == 3 ? 16 : 32);
            printf("%d wave(s)/SIMD  %-14s: %.2f s_memtime ticks per instruction per wave (%d instr per iteration)\n", waves, names[m], (double)h / iters / per, per);
        }
    return 0;
}
