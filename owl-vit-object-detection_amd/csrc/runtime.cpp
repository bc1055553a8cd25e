// Thread-local error string + version for libowlhip's C ABI (include/owl_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/owl_hip.h"
#define OWL_API extern "C" __attribute__((visibility("default")))   // (as in common.h: the library is linked with -fvisibility=hidden)

static thread_local char g_err[512] = "";

void owl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

OWL_API const char* owl_last_error(void) { return g_err; }
OWL_API int owl_abi_version(void) { return OWL_ABI_VERSION; }

// ---------------------------------------------------------------------------------------------------------------------
// Host side of the device input pipeline (ref src/dataset.py:69-71 -> HF OwlViTImageProcessor
// -> PIL Image.resize(BICUBIC)).  Pillow's Resample.c computes the separable filter taps in f64 on the host and converts
// them to 22-bit fixed point; the device kernels (preprocess.hip) consume exactly these tables, so the resize is
// bit-identical to Pillow's.  Built with -ffp-contract=off: the f64 expressions must round like Pillow's.
// ---------------------------------------------------------------------------------------------------------------------
#include <math.h>
#include <stdint.h>

static inline double pil_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// bounds[2*out] = {first tap, tap count}; kk[out*ksize] fixed-point weights (zero padded).  HOST pointers.
OWL_API int owl_bicubic_coeffs(int64_t in_size, int64_t out_size, int* bounds, int* kk, int64_t kk_capacity, int* ksize_out) {
    if (in_size <= 0 || out_size <= 0 || !bounds || !kk || !ksize_out) {
        owl_set_error("owl_bicubic_coeffs: bad arguments (in=%lld out=%lld)", (long long)in_size, (long long)out_size);
        return -1;
    }
    const int PRECISION_BITS = 32 - 8 - 2;
    double scale, filterscale;
    filterscale = scale = (double)in_size / (double)out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    if ((int64_t)ksize * out_size > kk_capacity) {
        owl_set_error("owl_bicubic_coeffs: kk capacity %lld < %lld", (long long)kk_capacity, (long long)ksize * out_size);
        return -1;
    }
    *ksize_out = ksize;
    const double ss = 1.0 / filterscale;
    double w[4096];
    if (ksize > 4096) { owl_set_error("owl_bicubic_coeffs: ksize %d too large", ksize); return -1; }
    for (int64_t xx = 0; xx < out_size; xx++) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = (int)in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; x++) {
            w[x] = pil_bicubic((x + xmin - center + 0.5) * ss);
            ww += w[x];
        }
        int* k = kk + xx * ksize;
        for (int x = 0; x < xmax; x++) {
            double v = w[x];
            if (ww != 0.0) v /= ww;
            k[x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
        }
        for (int x = xmax; x < ksize; x++) k[x] = 0;
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// The path's ONE collective: SUM of the flat f32 gradient bucket over the data-parallel ranks, in place, on the caller's
// stream, through the caller's RCCL communicator.  For hosts WITHOUT PyTorch: the Python host of this repo issues the same collective through
// torch.distributed (backend "nccl" = RCCL; ddp.py) because torch owns its communicator and does not hand it out.  librccl is bound lazily
// (dlopen at the first call), so libowlhip.so has no load-time dependency on it: single-GPU users never touch it.
// ---------------------------------------------------------------------------------------------------------------------
#include <dlfcn.h>
#include <stddef.h>

typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, void*);    // ncclAllReduce(send, recv, count, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t)
typedef const char* (*nccl_errstr_fn)(int);

OWL_API int owl_allreduce_sum_f32(void* stream, void* rccl_comm, float* buf, int64_t n) {
    if (!rccl_comm || !buf || n <= 0) {
        owl_set_error("owl_allreduce_sum_f32: null communicator / buffer or n = %lld", (long long)n);
        return -1;
    }
    static nccl_allreduce_fn allreduce = nullptr;          // (resolved once; a benign race resolves the same symbol twice)
    static nccl_errstr_fn errstr = nullptr;
    if (!allreduce) {
        // the RCCL instance the process ALREADY has (the one the caller built `rccl_comm` with: e.g. PyTorch's bundled copy) before loading another one
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { owl_set_error("owl_allreduce_sum_f32: cannot load librccl.so (%s)", dlerror()); return -2; }
        allreduce = (nccl_allreduce_fn)dlsym(h, "ncclAllReduce");
        errstr = (nccl_errstr_fn)dlsym(h, "ncclGetErrorString");
        if (!allreduce) { owl_set_error("owl_allreduce_sum_f32: librccl.so has no ncclAllReduce"); return -2; }
    }
    const int rc = allreduce(buf, buf, (size_t)n, /* ncclFloat32 */ 7, /* ncclSum */ 0, rccl_comm, stream);
    if (rc != 0) {
        owl_set_error("owl_allreduce_sum_f32: ncclAllReduce failed: %s", errstr ? errstr(rc) : "unknown RCCL error");
        return -3;
    }
    return 0;
}
