/* oracle/held_karp.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Exact TSP optimum by the Held-Karp dynamic programme, for instances small enough (n <= 21) that the table fits in
 * memory.  It is the "ref_cost" of SURVEY.md 8(d) config 1: the reference computes its gap against Concorde's optimum
 * stored in the instance files (scripts/test.py:62,104, gnngls/__init__.py:55-60); those files are git-LFS stubs and
 * Concorde is not installed, so for synthetic TSP<=20 instances the optimum comes from this DP instead.
 *
 *   C[S][j] = cost of the cheapest path that starts at node 0, visits exactly the nodes of S (a subset of 1..n-1) and
 *             ends at j in S;   C[{j}][j] = D[0][j];   C[S][j] = min_{k in S\{j}} C[S\{j}][k] + D[k][j]
 *   optimum = min_j C[all][j] + D[j][0]
 *
 * The optimal tour is rebuilt from the table and its cost re-summed left to right like gnngls.tour_cost
 * (gnngls/__init__.py:17-21), so the value is directly comparable with the costs the search reports. */
#include <stdint.h>
#include <stdlib.h>
#include <float.h>

double held_karp(const double *D, int n, int32_t *tour_out /* n+1 entries, may be NULL */) {
    if (n < 2 || n > 21) return -1.0;
    if (n == 2) {
        if (tour_out) { tour_out[0] = 0; tour_out[1] = 1; tour_out[2] = 0; }
        return D[1] + D[n];
    }
    const int m = n - 1;                       /* nodes 1..n-1 are bits 0..m-1 */
    const size_t full = (size_t)1 << m;
    double *C = (double *)malloc(full * m * sizeof(double));
    if (!C) return -2.0;
    for (size_t S = 1; S < full; ++S) {
        for (int j = 0; j < m; ++j) {
            if (!(S >> j & 1)) continue;
            const size_t prev = S & ~((size_t)1 << j);
            double best;
            if (prev == 0) {
                best = D[j + 1];               /* D[0][j+1] */
            } else {
                best = DBL_MAX;
                for (int k = 0; k < m; ++k) {
                    if (!(prev >> k & 1)) continue;
                    const double c = C[prev * m + k] + D[(size_t)(k + 1) * n + (j + 1)];
                    if (c < best) best = c;
                }
            }
            C[S * m + j] = best;
        }
    }
    /* rebuild the tour backwards from the best closing node */
    int32_t order[32];
    size_t S = full - 1;
    int last = -1;
    double best = DBL_MAX;
    for (int j = 0; j < m; ++j) {
        const double c = C[S * m + j] + D[(size_t)(j + 1) * n];
        if (c < best) { best = c; last = j; }
    }
    for (int pos = m - 1; pos >= 0; --pos) {
        order[pos] = last + 1;
        const size_t prev = S & ~((size_t)1 << last);
        if (prev == 0) break;
        int arg = -1;
        double bb = DBL_MAX;
        for (int k = 0; k < m; ++k) {
            if (!(prev >> k & 1)) continue;
            const double c = C[prev * m + k] + D[(size_t)(k + 1) * n + (last + 1)];
            if (c < bb) { bb = c; arg = k; }
        }
        S = prev;
        last = arg;
    }
    free(C);
    double cost = 0.0;                         /* gnngls.tour_cost: c = 0; c += w(e) left to right */
    int cur = 0;
    for (int pos = 0; pos < m; ++pos) { cost += D[(size_t)cur * n + order[pos]]; cur = order[pos]; }
    cost += D[(size_t)cur * n];
    if (tour_out) {
        tour_out[0] = 0;
        for (int pos = 0; pos < m; ++pos) tour_out[pos + 1] = order[pos];
        tour_out[n] = 0;
    }
    return cost;
}
