// Probe of ds_read_b64_tr_b16 on gfx950: LDS holds bf16 "values" = their own element index; every lane supplies
// the address  base + ((lane&15)>>2)*row_stride + (lane&3)*8 + (lane>>4)*group_stride  (hypothesis: natural row-major
// coverage of a [4][16] block per 16-lane group); prints what each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(unsigned short* out, int row_stride_b, int group_stride_b) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    unsigned addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)lds) + ((lane & 15) >> 2) * row_stride_b + (lane & 3) * 8 + (lane >> 4) * group_stride_b;
    typedef __attribute__((ext_vector_type(2))) unsigned u2;
    u2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[lane * 4 + 0] = v[0] & 0xffff; out[lane * 4 + 1] = v[0] >> 16; out[lane * 4 + 2] = v[1] & 0xffff; out[lane * 4 + 3] = v[1] >> 16;
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    int cfgs[2][2] = {{32, 128}, {512, 32}};
    for (int c = 0; c < 2; c++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, cfgs[c][0], cfgs[c][1]);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("row_stride %d B, group_stride %d B (element indices; row r col c of group g = g*gs/2 + r*rs/2 + c)\n", cfgs[c][0], cfgs[c][1]);
        for (int l = 0; l < 64; l++) { printf("lane %2d: %5d %5d %5d %5d%s", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (l % 4 == 3) ? "\n" : " | "); }
    }
    return 0;
}
