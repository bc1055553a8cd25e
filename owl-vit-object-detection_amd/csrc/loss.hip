// Hungarian-matched push-pull loss on device (no host sync anywhere on the path).
// Reference: src/matcher.py:85-159 (cost matrix + scipy linear_sum_assignment + target scatter),
// src/losses.py:42-69 (L1 / GIoU on matched pairs), src/losses.py:100-106 (sequential IoU > 0.85
// label spreading), src/losses.py:16-40 (focal-modulated weighted BCE on |cos sim|, split into
// positive / background rows).  Batched semantics: every term is the MEAN over images of the
// reference's batch-1 value.
//
// This file is compiled with -ffp-contract=off: IoU thresholding (> 0.85) and the assignment solve
// must see the same IEEE results as the reference's unfused f32 / f64 arithmetic.
#include "common.h"
#include <math.h>

// ---------------------------------------------------------------------------------------------------
// 1. matching cost, transposed:  costT[b][j][p] = |box_p - tgt_j|_1 - softmax(sims_p)[label_j] - GIoU
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float giou_pair(const float4 a, const float4 b, float* iou_out) {
    const float area_a = (a.z - a.x) * (a.w - a.y), area_b = (b.z - b.x) * (b.w - b.y);
    const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f), h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
    const float inter = w * h;
    const float uni = area_a + area_b - inter;
    const float iou = inter / uni;
    const float cw = fmaxf(fmaxf(a.z, b.z) - fminf(a.x, b.x), 0.f), ch = fmaxf(fmaxf(a.w, b.w) - fminf(a.y, b.y), 0.f);
    const float area_c = cw * ch;
    if (iou_out) *iou_out = iou;
    return iou - (area_c - uni) / area_c;
}

__global__ __launch_bounds__(256) void match_cost_kernel(const float* __restrict__ sims, const float* __restrict__ boxes,
                                                         const int64_t* __restrict__ labels, const float* __restrict__ tgt,
                                                         const int* __restrict__ counts, float* costT, int P, int C, int Nmax, float w_class, float w_bbox, float w_giou) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int n = counts[b];
    const float* s = sims + ((int64_t)b * P + p) * C;
    float mx = s[0];
    for (int c = 1; c < C; c++) mx = fmaxf(mx, s[c]);
    float den = 0.f;
    for (int c = 0; c < C; c++) den += expf(s[c] - mx);
    const float4 bx = *(const float4*)(boxes + ((int64_t)b * P + p) * 4);
    for (int j = 0; j < n; j++) {
        // a label outside [0, C) must not index sims (the reference raises IndexError at src/matcher.py:118); it gets a zero class
        // term here and turns loss_ce into NaN in class_loss_kernel, so the failure is loud without a host sync
        const int64_t lab = labels[(int64_t)b * Nmax + j];
        const float prob = (lab >= 0 && lab < C) ? expf(s[lab] - mx) / den : 0.f;
        const float4 t = *(const float4*)(tgt + ((int64_t)b * Nmax + j) * 4);
        const float l1 = fabsf(bx.x - t.x) + fabsf(bx.y - t.y) + fabsf(bx.z - t.z) + fabsf(bx.w - t.w);
        const float g = giou_pair(bx, t, nullptr);
        costT[((int64_t)b * Nmax + j) * P + p] = (w_bbox * l1 + w_class * (-prob)) + w_giou * (-g);
    }
}

// ---------------------------------------------------------------------------------------------------
// 2. rectangular LSAP, one workgroup per image.  Rows = targets (n <= P), columns = predictions:
//    the transposed problem scipy solves for a tall cost matrix.  Shortest augmenting paths with duals
//    in f64 (Crouse 2016, as scipy `_lsap`), the column scan parallel over the workgroup; the arg-min
//    reproduces scipy's sequential tie rule exactly (first minimum, unless a later equal minimum is an
//    unassigned column -- then the last such one).
// ---------------------------------------------------------------------------------------------------
struct Cand { double val; int first; int ulast; };

__device__ __forceinline__ Cand cand_merge(const Cand a, const Cand b) {
    if (a.val < b.val) return a;
    if (b.val < a.val) return b;
    Cand r; r.val = a.val; r.first = min(a.first, b.first); r.ulast = max(a.ulast, b.ulast);
    return r;
}

__global__ __launch_bounds__(512) void hungarian_kernel(const float* __restrict__ costT, const int64_t* __restrict__ labels,
                                                        const int* __restrict__ counts, int64_t* pred_idx, int64_t* tgt_idx,
                                                        int64_t* target_classes, int P, int Nmax, int bg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
    const int n = counts[b];
    double* v = (double*)sm;                    // [P]
    double* spc = v + P;                        // [P]
    double* u = spc + P;                        // [Nmax]
    int* path = (int*)(u + Nmax);               // [P]
    int* row4col = path + P;                    // [P]
    int* remaining = row4col + P;               // [P]
    int* col4row = remaining + P;               // [Nmax]
    unsigned char* SC = (unsigned char*)(col4row + Nmax);   // [P]
    unsigned char* SR = SC + P;                 // [Nmax]
    __shared__ Cand red[8];
    __shared__ int s_i, s_sink, s_nrem;
    __shared__ double s_min;

    int64_t* tc = target_classes + (int64_t)b * P;
    for (int j = tid; j < P; j += NT) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; tc[j] = bg; }
    for (int i = tid; i < Nmax; i += NT) { u[i] = 0.0; col4row[i] = -1; }
    __syncthreads();
    const float* cb = costT + (int64_t)b * Nmax * P;

    for (int cur = 0; cur < n; cur++) {
        for (int j = tid; j < P; j += NT) { remaining[j] = P - j - 1; spc[j] = INFINITY; SC[j] = 0; }
        for (int i = tid; i < Nmax; i += NT) SR[i] = 0;
        if (tid == 0) { s_i = cur; s_sink = -1; s_nrem = P; s_min = 0.0; }
        __syncthreads();
        while (true) {
            const int i = s_i, nrem = s_nrem;
            const double min_val = s_min, ui = u[i];
            const float* crow = cb + (int64_t)i * P;
            Cand c; c.val = INFINITY; c.first = 0x7fffffff; c.ulast = -1;
            // a thread's columns of this scan, eight at a time with every cost-row element requested before the first is used (round 6: the plain loop paid
            // one L2 round trip per column and thread, five in a row at P = 2304 -- most of a step's time); same columns in the same order
            constexpr int SK = 8;
            for (int base = 0; base < nrem; base += SK * NT) {
                int jj[SK]; float cc[SK];
#pragma unroll
                for (int k = 0; k < SK; k++) { const int it = base + tid + k * NT; jj[k] = remaining[it < nrem ? it : 0]; }
#pragma unroll
                for (int k = 0; k < SK; k++) cc[k] = crow[jj[k]];
#pragma unroll
                for (int k = 0; k < SK; k++) {
                    const int it = base + tid + k * NT;
                    if (it < nrem) {
                        const int j = jj[k];
                        const double r = min_val + (double)cc[k] - ui - v[j];
                        double sj = spc[j];
                        if (r < sj) { path[j] = i; spc[j] = r; sj = r; }
                        Cand m; m.val = sj; m.first = it; m.ulast = (row4col[j] == -1) ? it : -1;
                        c = cand_merge(c, m);
                    }
                }
            }
            // wave reduce, then across the 8 waves
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                Cand oth;
                oth.val = __shfl_xor(c.val, o, 64); oth.first = __shfl_xor(c.first, o, 64); oth.ulast = __shfl_xor(c.ulast, o, 64);
                c = cand_merge(c, oth);
            }
            if ((tid & 63) == 0) red[tid >> 6] = c;
            __syncthreads();
            if (tid == 0) {
                Cand r = red[0];
                for (int k = 1; k < (NT >> 6); k++) r = cand_merge(r, red[k]);
                const int index = (r.ulast >= 0) ? r.ulast : r.first;
                SR[i] = 1;
                s_min = r.val;
                const int j = remaining[index];
                if (row4col[j] == -1) s_sink = j; else s_i = row4col[j];
                SC[j] = 1;
                remaining[index] = remaining[nrem - 1];
                s_nrem = nrem - 1;
            }
            __syncthreads();
            if (s_sink >= 0) break;
        }
        // dual update
        const double min_val = s_min;
        if (tid == 0) u[cur] += min_val;
        for (int i = tid; i < n; i += NT)
            if (SR[i] && i != cur) u[i] += min_val - spc[col4row[i]];
        for (int j = tid; j < P; j += NT)
            if (SC[j]) v[j] -= min_val - spc[j];
        __syncthreads();
        if (tid == 0) {   // augment
            int j = s_sink;
            while (true) {
                const int i = path[j];
                row4col[j] = i;
                const int t = col4row[i]; col4row[i] = j; j = t;
                if (i == cur) break;
            }
        }
        __syncthreads();
    }
    // emit pairs ordered by prediction index (as scipy returns for the tall problem) + scatter labels
    for (int i = tid; i < n; i += NT) {
        const int c = col4row[i];
        int rank = 0;
        for (int k = 0; k < n; k++) rank += (col4row[k] < c);
        pred_idx[(int64_t)b * Nmax + rank] = c;
        tgt_idx[(int64_t)b * Nmax + rank] = i;
        tc[c] = labels[(int64_t)b * Nmax + i];
    }
    for (int i = n + tid; i < Nmax; i += NT) { pred_idx[(int64_t)b * Nmax + i] = 0; tgt_idx[(int64_t)b * Nmax + i] = 0; }   // padding is defined
}

// ---------------------------------------------------------------------------------------------------
// 3. label spreading (src/losses.py:100-106): sequential over p, in place, transitive.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void spread_kernel(const float* __restrict__ boxes, int64_t* target_classes, int P, int bg, float thr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    float4* bx = (float4*)sm;          // [P]
    int* tc = (int*)(bx + P);          // [P]
    // The reference's loop visits p = 0..P-1 in order and acts only on rows that are (by then) non-background.  Reading
    // tc[p] one row at a time costs an LDS round trip per row (2304 x ~100 cycles = the whole kernel); instead a bit mask
    // of the non-background rows is kept in LDS and the NEXT such row >= p is found by one wave (64 mask words per probe).
    // Rows relabelled behind the cursor are, as in the reference, not revisited; rows relabelled ahead of it are.
    __shared__ unsigned posmask[256];  // P <= 8192
    __shared__ int next_s;
    const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
    const int nwords = (P + 31) >> 5;
    int64_t* tcg = target_classes + (int64_t)b * P;
    for (int i = tid; i < nwords; i += NT) posmask[i] = 0u;
    __syncthreads();
    for (int r = tid; r < P; r += NT) {
        bx[r] = *(const float4*)(boxes + ((int64_t)b * P + r) * 4);
        const int t = (int)tcg[r];
        tc[r] = t;
        if (t != bg) atomicOr(&posmask[r >> 5], 1u << (r & 31));
    }
    __syncthreads();
    int p = 0;
    while (true) {
        if (tid < 64) {                // wave 0: first set bit at position >= p
            int found = -1;
            for (int w0 = p >> 5; w0 < nwords && found < 0; w0 += 64) {
                const int wi = w0 + tid;
                unsigned m = wi < nwords ? posmask[wi] : 0u;
                if (wi == (p >> 5)) m &= ~0u << (p & 31);
                const unsigned long long any = __ballot(m != 0u);
                if (any) {
                    const int l = __ffsll((long long)any) - 1;
                    const unsigned ml = (unsigned)__shfl((int)m, l, 64);
                    found = (w0 + l) * 32 + (__ffs((int)ml) - 1);
                }
            }
            if (tid == 0) next_s = found;
        }
        __syncthreads();
        const int cur = next_s;
        if (cur < 0) break;            // uniform
        const int lab = tc[cur];
        const float4 a = bx[cur];
        const float area_a = (a.z - a.x) * (a.w - a.y);
        __syncthreads();               // everybody has read next_s / tc[cur] before anyone rewrites them
        for (int r = tid; r < P; r += NT) {
            const float4 q = bx[r];
            const float area_q = (q.z - q.x) * (q.w - q.y);
            const float w = fmaxf(fminf(a.z, q.z) - fmaxf(a.x, q.x), 0.f), h = fmaxf(fminf(a.w, q.w) - fmaxf(a.y, q.y), 0.f);
            const float inter = w * h;
            const float uni = area_a + area_q - inter;
            if (inter / uni > thr) { tc[r] = lab; atomicOr(&posmask[r >> 5], 1u << (r & 31)); }
        }
        __syncthreads();
        p = cur + 1;
    }
    for (int r = tid; r < P; r += NT) tcg[r] = tc[r];
}

// ---------------------------------------------------------------------------------------------------
// 4. class loss (src/losses.py:16-40) + gradient wrt sims; one workgroup per image.
//    per_image[b] = {loss_ce, loss_bg, -, -};  dsims holds d(loss_kind_b)/d sims (kind = row's kind).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* red, int tid, int nt) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int k = 0; k < (nt >> 6); k++) s += red[k];
    return s;
}

__global__ __launch_bounds__(1024) void class_loss_kernel(const float* __restrict__ sims, const int64_t* __restrict__ target_classes,
                                                          const float* __restrict__ scales, float* per_image, float* dsims,
                                                          int P, int C, int bg) {
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
    const int64_t* tc = target_classes + (int64_t)b * P;
    float npos = 0.f;
    for (int p = tid; p < P; p += NT) npos += (tc[p] != bg) ? 1.f : 0.f;
    npos = block_sum(npos, red, tid, NT);
    const float nbg = (float)P - npos;
    float lpos = 0.f, lbg = 0.f;
    for (int p = tid; p < P; p += NT) {
        const int lab = (int)tc[p];
        const bool pos = lab != bg;
        const float inv_rows = pos ? 1.f / npos : 1.f / nbg;
        // (a matched / spread label outside [0, C): the reference's one_hot raises, src/losses.py:29 -> poison loss_ce instead of a silent all-zero target)
        float row = (pos && (lab < 0 || lab >= C)) ? __builtin_nanf("") : 0.f;
        for (int c = 0; c < C; c++) {
            const float s = sims[((int64_t)b * P + p) * C + c];
            const float a = fabsf(s);
            const float y = (pos && c == lab) ? 1.f : 0.f;
            const float wgt = scales ? scales[c] : 1.f;
            // torch BCELoss: -w [ y log a + (1-y) log(1-a) ], logs clamped at -100
            const float la = fmaxf(logf(a), -100.f), l1a = fmaxf(log1pf(-a), -100.f);
            const float l = -wgt * (y * la + (1.f - y) * l1a);
            const float em = expf(-l);
            const float om = 1.f - em;
            row += om * om * l;
            if (dsims) {
                const float dF = 2.f * om * em * l + om * om;                        // d[(1-e^-l)^2 l]/dl
                const float dl = wgt * (a - y) / fmaxf((1.f - a) * a, 1e-12f);         // torch BCE backward
                const float sg = (s > 0.f) ? 1.f : ((s < 0.f) ? -1.f : 0.f);
                dsims[((int64_t)b * P + p) * C + c] = dF * dl * sg * inv_rows;
            }
        }
        if (pos) lpos += row; else lbg += row;
    }
    lpos = block_sum(lpos, red, tid, NT);
    lbg = block_sum(lbg, red, tid, NT);
    if (tid == 0) { per_image[b * 4 + 0] = lpos / npos; per_image[b * 4 + 1] = lbg / nbg; }
}

// ---------------------------------------------------------------------------------------------------
// 5. box losses on matched pairs (src/losses.py:42-69) + gradients wrt predicted boxes.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wmax(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }   // d max(a,b)/da
__device__ __forceinline__ float wmin(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }   // d min(a,b)/da

__global__ __launch_bounds__(256) void box_loss_kernel(const float* __restrict__ boxes, const float* __restrict__ tgt,
                                                       const int64_t* __restrict__ pred_idx, const int64_t* __restrict__ tgt_idx,
                                                       const int* __restrict__ counts, float* per_image, float* dl1, float* dgiou,
                                                       int P, int Nmax) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
    const int n = counts[b];
    if (dl1) for (int i = tid; i < P * 4; i += NT) { dl1[(int64_t)b * P * 4 + i] = 0.f; dgiou[(int64_t)b * P * 4 + i] = 0.f; }
    __syncthreads();
    float s_l1 = 0.f, s_g = 0.f;
    const float invn = 1.f / (float)n;
    for (int k = tid; k < n; k += NT) {
        const int64_t pi = pred_idx[(int64_t)b * Nmax + k], ti = tgt_idx[(int64_t)b * Nmax + k];
        const float4 a = *(const float4*)(boxes + ((int64_t)b * P + pi) * 4);
        const float4 t = *(const float4*)(tgt + ((int64_t)b * Nmax + ti) * 4);
        s_l1 += fabsf(a.x - t.x) + fabsf(a.y - t.y) + fabsf(a.z - t.z) + fabsf(a.w - t.w);
        const float area_a = (a.z - a.x) * (a.w - a.y), area_b = (t.z - t.x) * (t.w - t.y);
        const float ltx = fmaxf(a.x, t.x), lty = fmaxf(a.y, t.y), rbx = fminf(a.z, t.z), rby = fminf(a.w, t.w);
        const float w = fmaxf(rbx - ltx, 0.f), h = fmaxf(rby - lty, 0.f);
        const float inter = w * h, uni = area_a + area_b - inter, iou = inter / uni;
        const float cx0 = fminf(a.x, t.x), cy0 = fminf(a.y, t.y), cx1 = fmaxf(a.z, t.z), cy1 = fmaxf(a.w, t.w);
        const float cw = fmaxf(cx1 - cx0, 0.f), ch = fmaxf(cy1 - cy0, 0.f), area_c = cw * ch;
        const float giou = iou - (area_c - uni) / area_c;
        s_g += 1.f - giou;
        if (dl1) {
            auto sgn = [](float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); };
            float* o1 = dl1 + ((int64_t)b * P + pi) * 4;
            o1[0] = sgn(a.x - t.x) * invn; o1[1] = sgn(a.y - t.y) * invn; o1[2] = sgn(a.z - t.z) * invn; o1[3] = sgn(a.w - t.w) * invn;
            // forward-mode partials wrt (a.x, a.y, a.z, a.w)
            const float dw_on = (rbx - ltx >= 0.f) ? 1.f : 0.f, dh_on = (rby - lty >= 0.f) ? 1.f : 0.f;
            const float dcw_on = (cx1 - cx0 >= 0.f) ? 1.f : 0.f, dch_on = (cy1 - cy0 >= 0.f) ? 1.f : 0.f;
            float d_inter[4], d_area_a[4], d_area_c[4];
            d_inter[0] = dw_on * (-wmax(a.x, t.x)) * h;   d_inter[2] = dw_on * (wmin(a.z, t.z)) * h;
            d_inter[1] = w * dh_on * (-wmax(a.y, t.y));   d_inter[3] = w * dh_on * (wmin(a.w, t.w));
            d_area_a[0] = -(a.w - a.y); d_area_a[2] = (a.w - a.y); d_area_a[1] = -(a.z - a.x); d_area_a[3] = (a.z - a.x);
            d_area_c[0] = dcw_on * (-wmin(a.x, t.x)) * ch; d_area_c[2] = dcw_on * (wmax(a.z, t.z)) * ch;
            d_area_c[1] = cw * dch_on * (-wmin(a.y, t.y)); d_area_c[3] = cw * dch_on * (wmax(a.w, t.w));
            float* o2 = dgiou + ((int64_t)b * P + pi) * 4;
            for (int e = 0; e < 4; e++) {
                const float d_uni = d_area_a[e] - d_inter[e];
                const float d_iou = (d_inter[e] * uni - inter * d_uni) / (uni * uni);
                const float d_ratio = (d_uni * area_c - uni * d_area_c[e]) / (area_c * area_c);   // d(union/area_c)
                o2[e] = -(d_iou + d_ratio) * invn;
            }
        }
    }
    s_l1 = wave_sum(s_l1); s_g = wave_sum(s_g);
    if ((tid & 63) == 0) { red[tid >> 6] = s_l1; }
    __syncthreads();
    float t1 = 0.f;
    for (int k = 0; k < (NT >> 6); k++) t1 += red[k];
    __syncthreads();
    if ((tid & 63) == 0) { red[tid >> 6] = s_g; }
    __syncthreads();
    float t2 = 0.f;
    for (int k = 0; k < (NT >> 6); k++) t2 += red[k];
    if (tid == 0) { per_image[b * 4 + 2] = t1 * invn; per_image[b * 4 + 3] = t2 * invn; }
}

// 6. mean over images (fixed order -> deterministic)
__global__ void loss_reduce_kernel(const float* __restrict__ per_image, float* losses, int B) {
    const int k = threadIdx.x;
    if (k >= 4) return;
    float s = 0.f;
    for (int b = 0; b < B; b++) s += per_image[b * 4 + k];
    losses[k] = s / (float)B;
}

// 7. backward combine: d_sims = g[kind(row)] * dsims / B ; d_boxes = (g[2]*dl1 + g[3]*dgiou) / B
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ g, const int64_t* __restrict__ target_classes,
                                                       const float* __restrict__ dsims, const float* __restrict__ dl1,
                                                       const float* __restrict__ dgiou, float* out_sims, float* out_boxes,
                                                       int64_t rows, int C, int bg, float invB) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float gk = ((target_classes[r] != bg) ? g[0] : g[1]) * invB;
    for (int c = 0; c < C; c++) out_sims[r * C + c] = gk * dsims[r * C + c];
    const float g2 = g[2] * invB, g3 = g[3] * invB;
    for (int e = 0; e < 4; e++) out_boxes[r * 4 + e] = g2 * dl1[r * 4 + e] + g3 * dgiou[r * 4 + e];
}

// 8. pairwise IoU / union / GIoU (free functions box_iou / generalized_box_iou, ref src/matcher.py:8-44)
__global__ __launch_bounds__(256) void box_pairwise_kernel(const float* __restrict__ b1, const float* __restrict__ b2, float* out, int64_t N, int64_t M) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N * M) return;
    const int64_t r = i / M, c = i - r * M;
    const float4 a = *(const float4*)(b1 + r * 4), b = *(const float4*)(b2 + c * 4);
    const float area_a = (a.z - a.x) * (a.w - a.y), area_b = (b.z - b.x) * (b.w - b.y);
    const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f), h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
    const float inter = w * h, uni = area_a + area_b - inter;
    float iou;
    const float g = giou_pair(a, b, &iou);
    out[i] = iou; out[N * M + i] = uni; out[2 * N * M + i] = g;
}

OWL_API int owl_box_pairwise(void* stream, const float* boxes1, const float* boxes2, float* out3, int64_t N, int64_t M) {
    OWL_CHECK_ARG(boxes1 && boxes2 && out3, "owl_box_pairwise: null pointer");
    if (N * M == 0) return 0;
    hipLaunchKernelGGL(box_pairwise_kernel, dim3((unsigned)((N * M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boxes1, boxes2, out3, N, M);
    OWL_LAUNCH_CHECK();
    return 0;
}

// 9. ragged -> padded targets: one launch for the whole batch (the DETR-style per-image lists of ref main.py:77-79 /
//    src/matcher.py:94-104 concatenated by the caller; offsets[B+1] are the image boundaries)
__global__ __launch_bounds__(256) void pack_targets_kernel(const int64_t* __restrict__ labels_cat, const float* __restrict__ boxes_cat,
                                                           const int* __restrict__ offsets, int64_t* labels, float* boxes, int* counts, int Nmax) {
    const int b = blockIdx.x;
    const int o = offsets[b], n = offsets[b + 1] - o;
    if (threadIdx.x == 0) counts[b] = n;
    for (int j = threadIdx.x; j < Nmax; j += blockDim.x) {
        const bool live = j < n;
        labels[(int64_t)b * Nmax + j] = live ? labels_cat[o + j] : 0;
        *(float4*)(boxes + ((int64_t)b * Nmax + j) * 4) = live ? *(const float4*)(boxes_cat + (int64_t)(o + j) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

OWL_API int owl_pack_targets(void* stream, const int64_t* labels_cat, const float* boxes_cat, const int* offsets, int64_t* labels,
                                float* boxes, int* counts, int64_t B, int64_t Nmax) {
    OWL_CHECK_ARG(labels_cat && boxes_cat && offsets && labels && boxes && counts, "owl_pack_targets: null pointer");
    OWL_CHECK_ARG(B >= 1 && Nmax >= 1, "owl_pack_targets: need B >= 1 and Nmax >= 1");
    hipLaunchKernelGGL(pack_targets_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, labels_cat, boxes_cat, offsets, labels,
                       boxes, counts, (int)Nmax);
    OWL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------
OWL_API int owl_match_cost(void* stream, const float* sims, const float* boxes, const int64_t* labels, const float* tgt_boxes,
                              const int* counts, float* costT, int64_t B, int64_t P, int64_t C, int64_t Nmax, float w_class, float w_bbox, float w_giou) {
    OWL_CHECK_ARG(sims && boxes && labels && tgt_boxes && counts && costT, "owl_match_cost: null pointer");
    hipLaunchKernelGGL(match_cost_kernel, dim3((unsigned)((P + 255) / 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       sims, boxes, labels, tgt_boxes, counts, costT, (int)P, (int)C, (int)Nmax, w_class, w_bbox, w_giou);
    OWL_LAUNCH_CHECK();
    return 0;
}

OWL_API int owl_hungarian(void* stream, const float* costT, const int64_t* labels, const int* counts, int64_t* pred_idx,
                             int64_t* tgt_idx, int64_t* target_classes, int64_t B, int64_t P, int64_t Nmax, int64_t bg) {
    OWL_CHECK_ARG(costT && labels && counts && pred_idx && tgt_idx && target_classes, "owl_hungarian: null pointer");
    OWL_CHECK_ARG(Nmax >= 1 && Nmax <= P, "owl_hungarian: need 1 <= Nmax <= P (more targets than predictions is unsupported)");
    size_t sh = (size_t)P * (8 + 8 + 4 + 4 + 4 + 1) + (size_t)Nmax * (8 + 4 + 1) + 64;
    sh = (sh + 15) / 16 * 16;
    OWL_CHECK_ARG(sh <= 160 * 1024, "owl_hungarian: P=%lld too large for the LDS-resident solver", (long long)P);
    static unsigned long long attr_done = 0;
    OWL_ONCE_PER_DEVICE(attr_done, (void)hipFuncSetAttribute((const void*)hungarian_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    hipLaunchKernelGGL(hungarian_kernel, dim3((unsigned)B), dim3(512), sh, (hipStream_t)stream, costT, labels, counts, pred_idx,
                       tgt_idx, target_classes, (int)P, (int)Nmax, (int)bg);
    OWL_LAUNCH_CHECK();
    return 0;
}

OWL_API int owl_spread_labels(void* stream, const float* boxes, int64_t* target_classes, int64_t B, int64_t P, int64_t bg, float thr) {
    OWL_CHECK_ARG(boxes && target_classes, "owl_spread_labels: null pointer");
    const size_t sh = (size_t)P * 20;
    OWL_CHECK_ARG(sh <= 156 * 1024 && P <= 8192, "owl_spread_labels: P too large");
    static unsigned long long attr_done = 0;
    OWL_ONCE_PER_DEVICE(attr_done, (void)hipFuncSetAttribute((const void*)spread_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    hipLaunchKernelGGL(spread_kernel, dim3((unsigned)B), dim3(1024), sh, (hipStream_t)stream, boxes, target_classes, (int)P, (int)bg, thr);
    OWL_LAUNCH_CHECK();
    return 0;
}

OWL_API int owl_push_pull_loss(void* stream, const float* sims, const float* boxes, const int64_t* target_classes,
                                  const float* scales, const float* tgt_boxes, const int64_t* pred_idx, const int64_t* tgt_idx,
                                  const int* counts, float* per_image, float* losses, float* dsims, float* dl1, float* dgiou,
                                  int64_t B, int64_t P, int64_t C, int64_t Nmax, int64_t bg) {
    OWL_CHECK_ARG(sims && boxes && target_classes && tgt_boxes && pred_idx && tgt_idx && counts && per_image && losses, "owl_push_pull_loss: null pointer");
    OWL_CHECK_ARG((dsims == nullptr) == (dl1 == nullptr) && (dl1 == nullptr) == (dgiou == nullptr), "owl_push_pull_loss: gradient buffers are all-or-none");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(class_loss_kernel, dim3((unsigned)B), dim3(1024), 0, s, sims, target_classes, scales, per_image, dsims, (int)P, (int)C, (int)bg);
    OWL_LAUNCH_CHECK();
    hipLaunchKernelGGL(box_loss_kernel, dim3((unsigned)B), dim3(256), 0, s, boxes, tgt_boxes, pred_idx, tgt_idx, counts, per_image, dl1, dgiou, (int)P, (int)Nmax);
    OWL_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(64), 0, s, per_image, losses, (int)B);
    OWL_LAUNCH_CHECK();
    return 0;
}

OWL_API int owl_push_pull_loss_bwd(void* stream, const float* g4, const int64_t* target_classes, const float* dsims, const float* dl1,
                                      const float* dgiou, float* out_sims, float* out_boxes, int64_t B, int64_t P, int64_t C, int64_t bg) {
    OWL_CHECK_ARG(g4 && target_classes && dsims && dl1 && dgiou && out_sims && out_boxes, "owl_push_pull_loss_bwd: null pointer");
    const int64_t rows = B * P;
    hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g4, target_classes,
                       dsims, dl1, dgiou, out_sims, out_boxes, rows, (int)C, (int)bg, 1.0f / (float)B);
    OWL_LAUNCH_CHECK();
    return 0;
}
