// Shared device/host helpers for libowlhip (gfx950 / CDNA4 only -- no CUDA, no dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

// The shared library is linked with -fvisibility=hidden: the dynamic symbol table holds the C ABI of include/owl_hip.h (+ owl_hip_tuning.h in a tuning
// build) and nothing else -- no kernel stubs, no C++ helpers (tests/test_abi.py).
#define OWL_API extern "C" __attribute__((visibility("default")))

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;     // MFMA A/B fragment (8 bf16, 4 VGPR)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;    // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) unsigned short us4;
typedef __attribute__((ext_vector_type(8))) unsigned short us8;

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32 (RNE)
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((unsigned)u) << 16); }
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));   // ONE v_cvt_pk_bf16_f32
}

// Host side: run a statement once per (call site, device) -- the hipFuncSetAttribute calls that raise a kernel's dynamic-LDS limit.  The limit is a
// per-device property of the function: a process that drives several GPUs (not this repo's one-process-per-GPU launchers, but a caller of the C ABI
// may) has to raise it on each.  The device's bit is set AFTER the statement, so a second thread never launches ahead of the attribute.
#define OWL_ONCE_PER_DEVICE(mask, ...)                                                                   \
    do {                                                                                                 \
        int owl_dev_ = 0;                                                                                \
        (void)hipGetDevice(&owl_dev_);                                                                   \
        const unsigned long long owl_bit_ = 1ull << (owl_dev_ & 63);                                     \
        if (!(__atomic_load_n(&(mask), __ATOMIC_ACQUIRE) & owl_bit_)) {                                  \
            __VA_ARGS__;                                                                                 \
            __atomic_fetch_or(&(mask), owl_bit_, __ATOMIC_RELEASE);                                      \
        }                                                                                                \
    } while (0)

// Streaming (non-temporal) accesses for data that is dead once read, or not read again soon once written: such rows then do not age out what
// the neighbouring GEMMs re-read through the L2 / Infinity Cache (profiles/r02_encoder_streams.md: the add + LayerNorm kernel alone was worth
// 1.3 % of the train step and 2.5 % of every GEMM launch).
typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned nt_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ld_stream_f4(const float* p) {
    const nt_f32x4 t = __builtin_nontemporal_load((const nt_f32x4*)p);
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ uint2 ld_stream_u2(const void* p) {
    const nt_u32x2 t = __builtin_nontemporal_load((const nt_u32x2*)p);
    return make_uint2(t.x, t.y);
}
__device__ __forceinline__ void st_stream_f4(float* p, const float4& v) {
    const nt_f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, (nt_f32x4*)p);
}

// d/du [u Phi(u)] = Phi(u) + u phi(u) for the BACKWARD of the erf-GELU (box head): Phi through the Abramowitz-Stegun 7.1.26 rational form of erf
// (|error| <= 1.5e-7: an f32 ulp of Phi), which shares its exp(-u^2/2) with phi -- one v_exp, one v_rcp and a dozen FMAs instead of libm's
// branchy erff plus an expf (the kernels that use it are bound by exactly these instructions).  Gradient-only: the forward GELU keeps erff.
__device__ __forceinline__ float dgelu_erf_f(float u) {
    const float x = fabsf(u) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    const float e = __builtin_amdgcn_exp2f(u * u * -0.72134752044448170f);          // exp(-u^2 / 2)
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float erf_abs = fmaf(-poly, e, 1.0f);                                     // erf(|u| / sqrt 2)
    return fmaf(0.5f, copysignf(erf_abs, u), 0.5f) + u * 0.39894228040143268f * e;
}

// LDS reads the compiler must not "protect": after a global_load_lds the compiler makes every later C++ LDS read
// wait for vmcnt(0) -- and on gfx950 vmcnt also counts STORES, so one bias read between two epilogue stores turns
// the whole store tail into store -> ack -> store -> ack (measured: ~9 us of a 34 us K=768 tile).  These helpers
// issue the ds_read from inline asm (invisible to that logic) and wait for their own lgkmcnt only.  The caller is
// responsible for the data really being in LDS (its own counted vmcnt + barrier).
__device__ __forceinline__ f32x4 lds_read_f4(const void* p) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(uintptr_t)LPTR(p)) : "memory");
    return v;
}
__device__ __forceinline__ float lds_read_f1(const void* p) {
    float v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(uintptr_t)LPTR(p)) : "memory");
    return v;
}

// ---- LDS hardware transpose (ds_read_b64_tr_b16) of a ROW-MAJOR [64 token][64 feature] bf16 tile with 128-byte rows -----------------
// 16-byte-chunk swizzle: the transpose-reads of one half-wave touch rows r0..r0+3 x 64 B, rows r0 and r0+2 on the same half of the
// bank row, so bit 2 of the chunk index must differ between them; the plain ds_read_b128 fragment reads (lane = row) stay conflict-free
// under the same permutation, so ONE image serves both access kinds.
__device__ __forceinline__ int swz_vrow(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
// Per-lane byte offset (inside the tile) of the transpose-read that yields, for feature block `fb` (32 features) and half h2 of the
// lane's 8 tokens, feature fb*32 + (lane&31) of tokens tok16*16 + (lane>>5)*8 + h2*4 + 0..3 -- WITHOUT the tok16*2048 term, which is an
// immediate for the caller (the swizzle does not depend on it).  Inside its 16-lane group (g = lane>>4) lane j supplies token row
// (g>>1)*8 + h2*4 + (j>>2) and feature quad (j&3) of the group's 16 features.
__device__ __forceinline__ unsigned tr_lane_off(int lane, int fb, int h2) {
    const int j = lane & 15, g = lane >> 4;
    const int row = (g >> 1) * 8 + h2 * 4 + (j >> 2);
    const int chunk = fb * 4 + (g & 1) * 2 + ((j & 3) >> 1);
    return (unsigned)(row * 128 + ((chunk ^ swz_vrow(row)) << 4) + (j & 1) * 8);
}
// the 8-element MFMA fragment = two transpose-reads (absolute 32-bit LDS addresses)
__device__ __forceinline__ bf16x8 lds_tr8(unsigned addr_lo, unsigned addr_hi) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(uintptr_t)addr_lo);
    const s16x4_t hh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(uintptr_t)addr_hi);
    return __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Two 32x32 MFMA accumulators (rows = 2 x 32 features d, column = the lane's token) -> 4 x 16 bytes of the token's output row.
// A lane holds 4 consecutive d per register quad and its partner lane^32 the adjacent 4; one v_permlane32_swap per packed word
// pairs them into 8 consecutive d per lane: st[d][pr] belongs at feature offset d*32 + 16*pr + 8*(lane>>5).  Half the store
// instructions of the natural 8-byte layout (the store tail of an attention workgroup is issue-bound).  Call with all lanes active.
__device__ __forceinline__ void pack_token_rows(const f32x16 (&acc)[2], float scale, uint4 (&st)[2][2]) {
#pragma unroll
    for (int d = 0; d < 2; d++)
#pragma unroll
        for (int pr = 0; pr < 2; pr++) {
            unsigned x[2], y[2];                    // x = even quad (2*pr), y = odd quad (2*pr + 1)
            x[0] = pack_bf2(acc[d][(2 * pr) * 4 + 0] * scale, acc[d][(2 * pr) * 4 + 1] * scale);
            x[1] = pack_bf2(acc[d][(2 * pr) * 4 + 2] * scale, acc[d][(2 * pr) * 4 + 3] * scale);
            y[0] = pack_bf2(acc[d][(2 * pr + 1) * 4 + 0] * scale, acc[d][(2 * pr + 1) * 4 + 1] * scale);
            y[1] = pack_bf2(acc[d][(2 * pr + 1) * 4 + 2] * scale, acc[d][(2 * pr + 1) * 4 + 3] * scale);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                auto r = __builtin_amdgcn_permlane32_swap(x[k], y[k], false, false);   // lanes 32-63 of x <-> lanes 0-31 of y
                x[k] = r[0]; y[k] = r[1];
            }
            st[d][pr] = make_uint4(x[0], x[1], y[0], y[1]);
        }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- error plumbing (thread-local last error) ----------------------------
void owl_set_error(const char* fmt, ...);
// out[i] (+)= sum_s slabs[s*stride + i]  (gemm.hip; shared with backward.hip)
int owl_slab_reduce_impl(hipStream_t s, const float* slabs, float* out, int64_t n, int64_t slab_stride, int nsplit, int accumulate);
#define OWL_CHECK_ARG(cond, ...)                  \
    do {                                          \
        if (!(cond)) {                            \
            owl_set_error(__VA_ARGS__);           \
            return -1;                            \
        }                                         \
    } while (0)
#define OWL_LAUNCH_CHECK()                                                        \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            owl_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return -2;                                                            \
        }                                                                         \
    } while (0)
