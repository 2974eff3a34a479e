/* oracle/one_tree.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Held-Karp lower bound of the symmetric TSP optimum by subgradient ascent over 1-trees (Held & Karp 1970/71): the other
 * side of the optimality-gap bracket bench.py reports for n > 20.  The reference divides by Concorde's optimum stored in
 * its instance files (scripts/test.py:62,104, gnngls/__init__.py:55-60); those files are git-LFS stubs and Concorde is
 * not installed, so for synthetic instances above the reach of the exact DP (oracle/held_karp.c, n <= 21)
 *
 *      lower bound  <=  optimum  <=  best-known tour length
 *
 * brackets the true gap: gap vs best-known <= true gap <= gap vs this bound.
 *
 * For node potentials pi, every tour T has c_pi(T) = c(T) + 2 sum(pi) with c_pi[i][j] = c[i][j] + pi[i] + pi[j], and every
 * tour is a 1-tree (a spanning tree of nodes 1..n-1 plus two edges at node 0), so
 *      w(pi) = min over 1-trees of c_pi  -  2 sum(pi)   <=   optimum          for ANY pi.
 * The ascent moves pi along the subgradient (degree - 2) with Polyak steps towards an upper bound (the search result);
 * whatever the step rule does, the value returned is max_k w(pi_k), each a valid bound (up to rounding of ~n sums:
 * callers compare with a 1e-9 relative slack). */
#include <float.h>
#include <stdlib.h>

/* minimum 1-tree for costs c + pi_i + pi_j: Prim on nodes 1..n-1 (O(n^2)), then the two cheapest edges at node 0.
 * Returns its modified cost, fills deg[0..n-1]. */
static double min_one_tree(const double *c, const double *pi, int n, int *deg, double *key, int *parent, char *in) {
    double total = 0.0;
    for (int i = 0; i < n; ++i) { deg[i] = 0; in[i] = 0; key[i] = DBL_MAX; parent[i] = -1; }
    key[1] = 0.0;
    for (int it = 1; it < n; ++it) {
        int u = -1;
        for (int v = 1; v < n; ++v) if (!in[v] && (u < 0 || key[v] < key[u])) u = v;
        in[u] = 1;
        if (parent[u] >= 0) { total += key[u]; deg[u]++; deg[parent[u]]++; }
        const double *cu = c + (size_t)u * n;
        for (int v = 1; v < n; ++v) {
            if (in[v]) continue;
            const double w = cu[v] + pi[u] + pi[v];
            if (w < key[v]) { key[v] = w; parent[v] = u; }
        }
    }
    double m1 = DBL_MAX, m2 = DBL_MAX; int a1 = -1, a2 = -1;
    for (int v = 1; v < n; ++v) {
        const double w = c[v] + pi[0] + pi[v];
        if (w < m1) { m2 = m1; a2 = a1; m1 = w; a1 = v; } else if (w < m2) { m2 = w; a2 = v; }
    }
    total += m1 + m2; deg[0] = 2; deg[a1]++; deg[a2]++;
    return total;
}

/* c [n,n] symmetric; ub = length of any tour (steers the step size only); max_iters subgradient steps.
 * Returns the best bound found (a tour's own length if the ascent lands on one: then it is the optimum). */
double one_tree_lower_bound(const double *c, int n, double ub, int max_iters) {
    if (n < 3) return n == 2 ? c[1] + c[n] : 0.0;
    double *pi = (double *)calloc((size_t)n, sizeof(double)), *best_pi = (double *)calloc((size_t)n, sizeof(double));
    double *key = (double *)malloc((size_t)n * sizeof(double));
    int *deg = (int *)malloc((size_t)n * sizeof(int)), *parent = (int *)malloc((size_t)n * sizeof(int));
    char *in = (char *)malloc((size_t)n);
    double best = -DBL_MAX, lambda = 2.0;
    int stall = 0;
    const int period = n < 50 ? 25 : n / 2;
    for (int it = 0; it < max_iters && lambda > 1e-5; ++it) {
        double sum_pi = 0.0;
        for (int i = 0; i < n; ++i) sum_pi += pi[i];
        const double w = min_one_tree(c, pi, n, deg, key, parent, in) - 2.0 * sum_pi;
        if (w > best) { best = w; stall = 0; for (int i = 0; i < n; ++i) best_pi[i] = pi[i]; } else stall++;
        long norm = 0;
        for (int i = 0; i < n; ++i) norm += (long)(deg[i] - 2) * (deg[i] - 2);
        if (norm == 0) break;                                  /* the 1-tree is a tour: w is the optimum */
        if (stall >= period) { lambda *= 0.5; stall = 0; for (int i = 0; i < n; ++i) pi[i] = best_pi[i]; continue; }
        const double gap = ub > w ? ub - w : 1e-3 * (ub > 0 ? ub : 1.0);
        const double step = lambda * gap / (double)norm;
        for (int i = 0; i < n; ++i) pi[i] += step * (double)(deg[i] - 2);
    }
    free(pi); free(best_pi); free(key); free(deg); free(parent); free(in);
    return best;
}
