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
