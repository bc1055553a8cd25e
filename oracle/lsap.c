/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into or called from the product path.
 *
 * CPU restatement of the rectangular linear-sum-assignment solve the reference performs at
 * reference src/matcher.py:134-137 (`scipy.optimize.linear_sum_assignment(c[i])`).
 *
 * The arithmetic lives in a third-party dependency that is NOT under /root/reference:
 *   scipy (unpinned in reference requirements.txt:2; 1.15.3 in this image), C++ `_lsap`
 *   (scipy/optimize/rectangular_lsap/rectangular_lsap.cpp).
 * Published algorithm restated here: D. F. Crouse, "On implementing 2D rectangular assignment
 * algorithms", IEEE Trans. Aerospace and Electronic Systems 52(4), 2016 -- shortest augmenting
 * path with dual variables (u, v), float64, tall inputs solved on the transpose, rows returned
 * ascending.  Tie handling (prefer an unassigned column among equal minima; candidate list filled
 * in reverse and compacted by swap-with-last) follows the scipy implementation so indices, not
 * just costs, agree on tie-heavy inputs.
 *
 * Pinned by tests/test_oracle_lsap.py against scipy itself (random, tie-heavy integer, tall, wide,
 * degenerate shapes) and against tests/golden/lsap_cases.npz.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int64_t augmenting_path(int64_t nc, const double *cost, double *u, double *v, int64_t *path,
                               int64_t *row4col, double *spc, int64_t i, uint8_t *SR, uint8_t *SC,
                               int64_t *remaining, double *p_min)
{
    double min_val = 0.0;
    int64_t num_remaining = nc;
    for (int64_t it = 0; it < nc; it++) remaining[it] = nc - it - 1;
    /* SR is sized by the caller to nr; cleared there */
    memset(SC, 0, (size_t)nc);
    for (int64_t j = 0; j < nc; j++) spc[j] = INFINITY;

    int64_t sink = -1;
    while (sink == -1) {
        int64_t index = -1;
        double lowest = INFINITY;
        SR[i] = 1;
        for (int64_t it = 0; it < num_remaining; it++) {
            int64_t j = remaining[it];
            double r = min_val + cost[i * nc + j] - u[i] - v[j];
            if (r < spc[j]) {
                path[j] = i;
                spc[j] = r;
            }
            if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) {
                lowest = spc[j];
                index = it;
            }
        }
        min_val = lowest;
        if (min_val == INFINITY) return -1; /* infeasible */
        int64_t j = remaining[index];
        if (row4col[j] == -1)
            sink = j;
        else
            i = row4col[j];
        SC[j] = 1;
        remaining[index] = remaining[--num_remaining];
    }
    *p_min = min_val;
    return sink;
}

/* cost: nr x nc row-major float64.  Writes min(nr, nc) pairs to (a, b).  Returns 0, or -1 if
 * infeasible, -2 on invalid entries (NaN / -inf). */
int oracle_lsap(int64_t nr, int64_t nc, const double *cost_in, int64_t *a, int64_t *b)
{
    if (nr == 0 || nc == 0) return 0;
    int transpose = nc < nr;
    double *cost = (double *)malloc(sizeof(double) * (size_t)(nr * nc));
    if (transpose) {
        for (int64_t i = 0; i < nr; i++)
            for (int64_t j = 0; j < nc; j++) cost[j * nr + i] = cost_in[i * nc + j];
        int64_t t = nr; nr = nc; nc = t;
    } else {
        memcpy(cost, cost_in, sizeof(double) * (size_t)(nr * nc));
    }
    for (int64_t k = 0; k < nr * nc; k++)
        if (cost[k] != cost[k] || cost[k] == -INFINITY) { free(cost); return -2; }

    double *u = (double *)calloc((size_t)nr, sizeof(double));
    double *v = (double *)calloc((size_t)nc, sizeof(double));
    double *spc = (double *)malloc(sizeof(double) * (size_t)nc);
    int64_t *path = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
    int64_t *col4row = (int64_t *)malloc(sizeof(int64_t) * (size_t)nr);
    int64_t *row4col = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
    int64_t *remaining = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
    uint8_t *SR = (uint8_t *)malloc((size_t)nr);
    uint8_t *SC = (uint8_t *)malloc((size_t)nc);
    for (int64_t k = 0; k < nc; k++) { path[k] = -1; row4col[k] = -1; }
    for (int64_t k = 0; k < nr; k++) col4row[k] = -1;

    int rc = 0;
    for (int64_t cur = 0; cur < nr; cur++) {
        double min_val;
        memset(SR, 0, (size_t)nr);
        int64_t sink = augmenting_path(nc, cost, u, v, path, row4col, spc, cur, SR, SC, remaining, &min_val);
        if (sink < 0) { rc = -1; break; }
        u[cur] += min_val;
        for (int64_t i = 0; i < nr; i++)
            if (SR[i] && i != cur) u[i] += min_val - spc[col4row[i]];
        for (int64_t j = 0; j < nc; j++)
            if (SC[j]) v[j] -= min_val - spc[j];
        int64_t j = sink;
        for (;;) {
            int64_t i = path[j];
            row4col[j] = i;
            int64_t t = col4row[i]; col4row[i] = j; j = t;
            if (i == cur) break;
        }
    }
    if (rc == 0) {
        if (transpose) {
            /* rows of the original problem are the columns here: emit ascending in them */
            int64_t k = 0;
            for (int64_t j = 0; j < nc; j++)
                if (row4col[j] != -1) { a[k] = j; b[k] = row4col[j]; k++; }
        } else {
            for (int64_t i = 0; i < nr; i++) { a[i] = i; b[i] = col4row[i]; }
        }
    }
    free(cost); free(u); free(v); free(spc); free(path); free(col4row); free(row4col); free(remaining); free(SR); free(SC);
    return rc;
}
