// Epilogue store path of one CU: 8 waves each issue 16 x global_store_dwordx4 (1 KiB per instruction) of a 256 x 256 bf16 tile into a row-major
// matrix with `ld` columns, (a) as the GEMM epilogue does it -- an instruction covers 32 rows x 32 contiguous bytes -- (b) 16 rows x 64 bytes,
// (c) 8 rows x 128 bytes (whole cache lines).  256 workgroups (one per CU), `iters` tiles each, marching down the matrix; wall time by HIP events.
// hipcc --offload-arch=gfx950 -O3 tools/probe/store_pattern.hip -o tools/probe/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(unsigned short* out, int64_t ld, int iters, int col_tiles) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int grp = w >> 2, wc = w & 3;                    // wave: rows grp*128 .. +128, columns wc*64 .. +64 (as gemm_pp2)
    uint4 v = make_uint4(lane, w, blockIdx.x, 7);
    for (int it = 0; it < iters; it++) {
        const int tile = blockIdx.x + it * gridDim.x;
        const int64_t tm = tile / col_tiles, tn = tile % col_tiles;
        unsigned short* base = out + (tm * 256 + grp * 128) * ld + tn * 256 + wc * 64;
#pragma unroll
        for (int s = 0; s < 16; s++) {                     // 16 instructions cover the wave's 128 rows x 64 columns (128 B per row)
            int row, col;
            if (MODE == 0) {        // 32 rows x 32 B: instruction s = (row tile s>>2, 16-column chunk s&3); lane -> row lane&31, 16 B at (lane>>5)*8
                row = (s >> 2) * 32 + (lane & 31); col = (s & 3) * 16 + (lane >> 5) * 8;
            } else if (MODE == 1) { // 16 rows x 64 B
                row = (s >> 1) * 16 + (lane >> 2); col = (s & 1) * 32 + (lane & 3) * 8;
            } else if (MODE == 2) { // 8 rows x 128 B: whole lines, 8 consecutive lanes per row
                row = s * 8 + (lane >> 3); col = (lane & 7) * 8;
            } else if (MODE == 3) { // whole lines as two in-register butterfly stages leave them (lanes r, r+8, r+16, r+24 and their +32 partners
                                    // transposed against the four chunk registers): a row's eight pieces sit in lanes r0 + 8a + 16b + 32hi
                const int r0 = lane & 7, a = (lane >> 3) & 1, b = (lane >> 4) & 1, hi = lane >> 5;
                row = (s >> 2) * 32 + r0 + 8 * (s & 1) + 16 * ((s >> 1) & 1); col = (2 * (a + 2 * b) + hi) * 8;
            } else if (MODE == 5) { // 32 rows x 32 B, the two pieces of a row in CONSECUTIVE lanes
                row = (s >> 2) * 32 + (lane >> 1); col = (s & 3) * 16 + (lane & 1) * 8;
            } else if (MODE == 6) { // 16 rows x 64 B in consecutive lanes, the rows of an instruction 8 apart (any 16 rows do)
                row = (s >> 2) * 32 + ((lane >> 2) & 7) + 8 * (s & 3) - 8 * (s & 3) + 16 * (lane >> 5) + ((s & 1) ? 8 : 0) - ((s & 1) ? 8 : 0); row = (s >> 2) * 32 + (s & 1) * 16 + (lane >> 2); col = ((s >> 1) & 1) * 32 + (lane & 3) * 8 ; row = (s >> 2) * 32 + ((lane >> 2) * 2 + (s & 1)) % 32 ;
            } else {                // MODE 4: half lines as ONE v_permlane16_swap stage leaves them: a row's four pieces in lanes r, r+16, r+32, r+48
                const int b = (lane >> 4) & 1, hi = lane >> 5;
                row = (s >> 2) * 32 + (lane & 15) + 16 * (s & 1); col = (4 * ((s >> 1) & 1) + 2 * b + hi) * 8;
            }
            *(uint4*)(base + (int64_t)row * ld + col) = v;
        }
        v.x += 1;
    }
}

int main() {
    const int64_t M = 73984, N = 2304;
    unsigned short* d; hipMalloc(&d, (M + 256) * N * 2);
    const int col_tiles = N / 256, tiles = (M / 256) * col_tiles, iters = tiles / 256;
    for (int grid : {256, 64, 16})
        for (int mode = 0; mode < 7; mode++) {
            const int iters = tiles / 256 * (256 / grid) / (256 / grid);      // same tiles per workgroup
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&]() {
                if (mode == 0) store_kernel<0><<<grid, 512>>>(d, N, iters, col_tiles);
                else if (mode == 1) store_kernel<1><<<grid, 512>>>(d, N, iters, col_tiles);
                else if (mode == 2) store_kernel<2><<<grid, 512>>>(d, N, iters, col_tiles);
                else if (mode == 3) store_kernel<3><<<grid, 512>>>(d, N, iters, col_tiles);
                else if (mode == 4) store_kernel<4><<<grid, 512>>>(d, N, iters, col_tiles);
                else if (mode == 5) store_kernel<5><<<grid, 512>>>(d, N, iters, col_tiles);
                else store_kernel<6><<<grid, 512>>>(d, N, iters, col_tiles);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); for (int i = 0; i < 10; i++) launch(); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            const double bytes = (double)iters * grid * 256 * 256 * 2;
            printf("%d workgroups, %s: %.1f us per launch (%d tiles per workgroup, %.0f MB) = %.2f TB/s = %.2f us per 128-KiB tile per CU\n",
                   grid, mode == 0 ? "32 rows x 32 B per instruction (the GEMM epilogue)" : mode == 1 ? "16 rows x 64 B (4 consecutive lanes)" : mode == 2 ? "8 rows x 128 B (whole lines, 8 consecutive lanes)"
                   : mode == 3 ? "8 rows x 128 B, butterfly lane order" : mode == 4 ? "16 rows x 64 B, permlane16 lane order"
                   : mode == 5 ? "32 rows x 32 B, consecutive lane pairs" : "16 rows x 64 B (4 consecutive lanes), every other row",
                   ms * 1e3, iters, bytes / 1e6, bytes / ms / 1e9, ms * 1e3 / iters);
        }
    return 0;
}
