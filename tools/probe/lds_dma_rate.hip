// What does the CU's HBM/L2 -> LDS path (global_load_lds_dwordx4, "LDS-DMA") sustain on MI355X, with nothing else running?
// Round 4: the free-running GEMM (csrc/gemm_fr.hip) needs 1.5x the LDS-DMA bytes per FLOP of the shipped 256 x 256 tile and runs 1.3-1.45x as long;
// without its requests it would be 21 % FASTER than the shipped kernel.  Both move ~9 TB/s through this path inside the GEMM.  Is that the path's rate?
//
// Each wave streams 1-KiB pieces (64 lanes x 16 B) from a window of `window` bytes per workgroup into its own LDS ring, `depth` pieces in flight
// (counted vmcnt), no barrier, no MFMA.  Pieces are either 8 rows x 128 B (whole cache lines: the 256 x 256 x 64 tile's image) or 16 rows x 64 B
// (half lines: the BK = 32 image), rows `ld` bytes apart as in a [M, K] operand.  Also the same traffic as plain global_load_dwordx4 into VGPRs.
// hipcc --offload-arch=gfx950 -O3 tools/probe/lds_dma_rate.hip -o tools/probe/lds_dma_rate && ./tools/probe/lds_dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int DEPTH> __device__ __forceinline__ void wait_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory"); }

// MODE 0: LDS-DMA, 8 rows x 128 B per piece; 1: LDS-DMA, 16 rows x 64 B per piece; 2: global_load_dwordx4 to VGPRs, 8 rows x 128 B
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// MFMAS: 32x32x16 bf16 MFMAs each wave issues behind every request (the GEMMs run 16 per 6 requests = 2.7 at BK = 32, 32 per 8 = 4 at the 256 x 256 tile)
template <int MODE, int DEPTH, int MFMAS = 0>
__global__ __launch_bounds__(256) void stream_kernel(const unsigned char* src, int64_t ld, int rows_per_wg, int k_bytes, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned char* ring = lds + w * (DEPTH + 1) * 1024;
    // the workgroup's window: rows [blockIdx.x * rows_per_wg, +rows_per_wg) x k_bytes; wave w walks pieces w, w + 4, ...
    const unsigned char* base = src + (int64_t)blockIdx.x * rows_per_wg * ld;
    const int rpp = MODE == 1 ? 16 : 8, bpr = MODE == 1 ? 64 : 128;            // rows per piece, bytes per row
    const int lane_row = MODE == 1 ? (lane >> 2) : (lane >> 3), lane_col = MODE == 1 ? (lane & 3) * 16 : (lane & 7) * 16;
    const int pieces_k = k_bytes / bpr, pieces_m = rows_per_wg / rpp;
    // wave w walks row groups w, w + 4, ... and, inside a row group, the K pieces in order: pointer increments only (no division in the loop --
    // the first version of this probe divided per piece and measured its own scalar division: 6.9 GB/s per wave whatever the depth)
    uint4 acc = make_uint4(0, 0, 0, 0);
    bf16x8 fa, fb; f32x16 c[4];
    for (int e = 0; e < 8; e++) { fa[e] = (short)(0x3f80 + lane + e); fb[e] = (short)(0x3f00 + e); }
    for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) c[i][e] = (float)(i + e);
    int slot = 0;
    for (int it = 0; it < iters; it++) {
        for (int pm = w; pm < pieces_m; pm += 4) {
            const unsigned char* g = base + (int64_t)(pm * rpp + lane_row) * ld + lane_col;
            if (MODE == 2) {                      // four loads in flight per wave, then consumed (pieces_k % 4 == 0)
                for (int pk = 0; pk < pieces_k; pk += 4) {
                    uint4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = *(const uint4*)(g + (pk + u) * bpr);
#pragma unroll
                    for (int u = 0; u < 4; u++) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
                }
            } else {
                for (int pk = 0; pk < pieces_k; pk++) {
                    __builtin_amdgcn_global_load_lds(GPTR(g + pk * bpr), LPTR(ring + slot * 1024), 16, 0, 0);
                    slot = slot == DEPTH ? 0 : slot + 1;
#pragma unroll
                    for (int m = 0; m < MFMAS; m++) c[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c[m & 3], 0, 0, 0);
                    wait_n<DEPTH>();
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
    if (MODE != 2 && ring[lane] == 0x7f && lane == 99) sink[1] = 1;
    if (MFMAS && c[0][0] + c[1][1] + c[2][2] + c[3][3] == 1.2345f) sink[2] = 1;
}

template <int MODE, int DEPTH, int MFMAS = 0>
static void run(const char* what, const unsigned char* src, int64_t ld, int wgs, int rows_per_wg, int k_bytes, unsigned* sink) {
    const int iters = 200;
    const size_t lds = 4 * (DEPTH + 1) * 1024;
    hipFuncSetAttribute((const void*)stream_kernel<MODE, DEPTH, MFMAS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    stream_kernel<MODE, DEPTH, MFMAS><<<wgs, 256, lds>>>(src, ld, rows_per_wg, k_bytes, 20, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    stream_kernel<MODE, DEPTH, MFMAS><<<wgs, 256, lds>>>(src, ld, rows_per_wg, k_bytes, iters, sink);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)wgs * rows_per_wg * k_bytes * iters;
    printf("  %-44s depth %2d, %4d workgroups (%d per CU), window %4d KiB/WG: %7.2f TB/s = %6.1f GB/s per CU", what, DEPTH, wgs, wgs / 256,
           rows_per_wg * k_bytes / 1024, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
    if (MFMAS) printf("   with %d MFMAs per request: %.0f TFLOP/s beside it", MFMAS, bytes / 1024 * MFMAS * 32768.0 / ms / 1e9);
    printf("\n");
}

int main() {
    const int64_t ld = 1536;                       // a [M, 768] bf16 operand
    const int64_t rows = 1 << 17;                  // 192 MiB: beyond every cache when walked once; windows below keep what they touch in L2
    unsigned char* src; hipMalloc(&src, rows * ld); hipMemset(src, 1, rows * ld);
    unsigned* sink; hipMalloc(&sink, 64);
    printf("L2-resident windows (each workgroup re-reads its own 48-KiB window: 32 rows x 1536 B; 256 workgroups x 48 KiB = 12 MiB over 8 L2s):\n");
    run<0, 4>("LDS-DMA, 8 rows x 128 B per piece", src, ld, 256, 32, 1536, sink);
    run<0, 8>("LDS-DMA, 8 rows x 128 B per piece", src, ld, 256, 32, 1536, sink);
    run<0, 16>("LDS-DMA, 8 rows x 128 B per piece", src, ld, 256, 32, 1536, sink);
    run<0, 8>("LDS-DMA, 8 rows x 128 B per piece", src, ld, 512, 32, 1536, sink);
    run<0, 16>("LDS-DMA, 8 rows x 128 B per piece", src, ld, 512, 32, 1536, sink);
    run<0, 8>("LDS-DMA, 8 rows x 128 B per piece", src, ld, 1024, 32, 1536, sink);
    run<1, 8>("LDS-DMA, 16 rows x 64 B per piece", src, ld, 256, 32, 1536, sink);
    run<1, 16>("LDS-DMA, 16 rows x 64 B per piece", src, ld, 512, 32, 1536, sink);
    run<1, 8>("LDS-DMA, 16 rows x 64 B per piece", src, ld, 1024, 32, 1536, sink);
    run<2, 8>("global_load_dwordx4 -> VGPR, 8 rows x 128 B", src, ld, 256, 32, 1536, sink);
    run<2, 8>("global_load_dwordx4 -> VGPR, 8 rows x 128 B", src, ld, 1024, 32, 1536, sink);
    printf("The same streams with MFMAs issued by the same waves behind every request (8 waves per CU):\n");
    run<0, 8, 2>("LDS-DMA, 8 rows x 128 B per piece", src, ld, 512, 32, 1536, sink);
    run<0, 8, 4>("LDS-DMA, 8 rows x 128 B per piece", src, ld, 512, 32, 1536, sink);
    run<0, 8, 8>("LDS-DMA, 8 rows x 128 B per piece", src, ld, 512, 32, 1536, sink);
    run<1, 8, 3>("LDS-DMA, 16 rows x 64 B per piece", src, ld, 512, 32, 1536, sink);
    run<0, 8, 4>("LDS-DMA, 8 rows x 128 B, Infinity-Cache-resident", src, ld, 512, 128, 1536, sink);
    printf("Infinity-Cache-resident windows (each workgroup walks 256 rows x 1536 B = 384 KiB; 256 workgroups: 96 MiB):\n");
    run<0, 8>("LDS-DMA, 8 rows x 128 B per piece", src, ld, 256, 256, 1536, sink);
    run<0, 16>("LDS-DMA, 8 rows x 128 B per piece", src, ld, 512, 128, 1536, sink);
    run<1, 16>("LDS-DMA, 16 rows x 64 B per piece", src, ld, 512, 128, 1536, sink);
    run<2, 8>("global_load_dwordx4 -> VGPR, 8 rows x 128 B", src, ld, 1024, 64, 1536, sink);
    return 0;
}
