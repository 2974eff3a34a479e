/*
 * oracle/gls_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement (plain C, fp64 + int) of the reference's guided-local-search
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link
 * or call this file; the product path (gnngls_amd/) never does.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit in
 * tests/test_oracle_golden.py against golden vectors produced by importing the
 * reference's own Python (oracle/gen_golden.py -> tests/golden/(star).npz).
 *
 * Each function cites the reference lines (/root/reference/...) it restates.
 * Compile WITHOUT fp contraction (-ffp-contract=off) so `D + k*P` keeps its two
 * roundings (gnngls/algorithms.py:164).
 *
 * Conventions: n = number of TSP nodes; a tour is int32[n+1] with the depot 0 at both
 * ends (algorithms.py:10,17); D is a row-major n*n fp64 matrix indexed by node label
 * (nx.attr_matrix, algorithms.py:140); valid move indices are 1..n-1
 * (operators.py:36,59,112,133).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define DIDX(D, n, a, b) ((D)[(size_t)(a) * (size_t)(n) + (size_t)(b)])

/* np.isclose(0, delta) with numpy defaults rtol=1e-5, atol=1e-8 (operators.py:42,65,118,139):
 * |0 - delta| <= atol + rtol*|delta|, evaluated literally in fp64. */
static int is_close_to_zero(double delta) {
    double ad = fabs(delta);
    double rhs = 1e-8 + 1e-5 * ad; /* contraction is off: one mul rounding, one add rounding */
    return ad <= rhs;
}

/* gnngls/operators.py:14-29 */
double gls_oracle_two_opt_cost(const int32_t *tour, const double *D, int n, int i, int j) {
    if (i == j) return 0.0;
    if (j < i) { int t = i; i = j; j = t; }
    int a = tour[i], b = tour[i - 1], c = tour[j], d = tour[j - 1];
    double delta = DIDX(D, n, a, c) + DIDX(D, n, b, d);
    delta = delta - DIDX(D, n, a, b);
    delta = delta - DIDX(D, n, c, d);
    return delta;
}

/* gnngls/operators.py:6-11 : tour[:i] + tour[j-1:i-1:-1] + tour[j:]  (reverse positions i..j-1) */
void gls_oracle_two_opt(int32_t *tour, int n, int i, int j) {
    (void)n;
    if (i == j) return;
    if (j < i) { int t = i; i = j; j = t; }
    int lo = i, hi = j - 1;
    while (lo < hi) { int32_t t = tour[lo]; tour[lo] = tour[hi]; tour[hi] = t; lo++; hi--; }
}

/* gnngls/operators.py:83-103 */
double gls_oracle_relocate_cost(const int32_t *tour, const double *D, int n, int i, int j) {
    if (i == j) return 0.0;
    int a = tour[i - 1], b = tour[i], c = tour[i + 1];
    int d, e;
    if (i < j) { d = tour[j]; e = tour[j + 1]; }
    else       { d = tour[j - 1]; e = tour[j]; }
    double delta = -DIDX(D, n, a, b);
    delta = delta - DIDX(D, n, b, c);
    delta = delta + DIDX(D, n, a, c);
    delta = delta - DIDX(D, n, d, e);
    delta = delta + DIDX(D, n, d, b);
    delta = delta + DIDX(D, n, b, e);
    return delta;
}

/* gnngls/operators.py:76-80 : pop(i) then insert(j, node) */
void gls_oracle_relocate(int32_t *tour, int n, int i, int j) {
    (void)n;
    if (i == j) return;
    int32_t node = tour[i];
    if (i < j) { for (int p = i; p < j; ++p) tour[p] = tour[p + 1]; }
    else       { for (int p = i; p > j; --p) tour[p] = tour[p - 1]; }
    tour[j] = node;
}

/* gnngls/operators.py:32-50.  Returns 1 and (*bi,*bj,*bdelta) if a move was found, else 0. */
int gls_oracle_two_opt_a2a(const int32_t *tour, const double *D, int n, int first_improvement,
                           double *bdelta, int *bi, int *bj) {
    double best = 0.0; int found = 0;
    for (int i = 1; i <= n - 1; ++i) {
        for (int j = i + 1; j <= n - 1; ++j) {       /* itertools.combinations(range(1,n),2) */
            if (abs(i - j) < 2) continue;
            double delta = gls_oracle_two_opt_cost(tour, D, n, i, j);
            if (delta < best && !is_close_to_zero(delta)) {
                best = delta; *bi = i; *bj = j; found = 1;
                if (first_improvement) goto done;
            }
        }
    }
done:
    *bdelta = found ? best : 0.0;
    return found;
}

/* gnngls/operators.py:53-73 */
int gls_oracle_two_opt_o2a(const int32_t *tour, const double *D, int n, int i, int first_improvement,
                           double *bdelta, int *bj) {
    double best = 0.0; int found = 0;
    for (int j = 1; j <= n - 1; ++j) {
        if (abs(i - j) < 2) continue;
        double delta = gls_oracle_two_opt_cost(tour, D, n, i, j);
        if (delta < best && !is_close_to_zero(delta)) {
            best = delta; *bj = j; found = 1;
            if (first_improvement) break;
        }
    }
    *bdelta = found ? best : 0.0;
    return found;
}

/* gnngls/operators.py:129-147 */
int gls_oracle_relocate_a2a(const int32_t *tour, const double *D, int n, int first_improvement,
                            double *bdelta, int *bi, int *bj) {
    double best = 0.0; int found = 0;
    for (int i = 1; i <= n - 1; ++i) {
        for (int j = 1; j <= n - 1; ++j) {           /* itertools.permutations(range(1,n),2) */
            if (i == j) continue;
            if (i - j == 1) continue;
            double delta = gls_oracle_relocate_cost(tour, D, n, i, j);
            if (delta < best && !is_close_to_zero(delta)) {
                best = delta; *bi = i; *bj = j; found = 1;
                if (first_improvement) goto done;
            }
        }
    }
done:
    *bdelta = found ? best : 0.0;
    return found;
}

/* gnngls/operators.py:106-126 */
int gls_oracle_relocate_o2a(const int32_t *tour, const double *D, int n, int i, int first_improvement,
                            double *bdelta, int *bj) {
    double best = 0.0; int found = 0;
    for (int j = 1; j <= n - 1; ++j) {
        if (i == j) continue;
        double delta = gls_oracle_relocate_cost(tour, D, n, i, j);
        if (delta < best && !is_close_to_zero(delta)) {
            best = delta; *bj = j; found = 1;
            if (first_improvement) break;
        }
    }
    *bdelta = found ? best : 0.0;
    return found;
}

/* gnngls/__init__.py:17-21 : c = 0; c += w(e) left to right */
double gls_oracle_tour_cost(const int32_t *tour, const double *D, int n) {
    double c = 0.0;
    for (int p = 0; p < n; ++p) c += DIDX(D, n, tour[p], tour[p + 1]);
    return c;
}

/* gnngls/algorithms.py:9-18 : greedy, ties -> lowest node id (first minimum of an ascending scan) */
void gls_oracle_nearest_neighbor(const double *W, int n, int depot, int32_t *tour) {
    uint8_t *visited = (uint8_t *)calloc((size_t)n, 1);
    tour[0] = depot; visited[depot] = 1;
    for (int len = 1; len < n; ++len) {
        int i = tour[len - 1], bj = -1; double bw = 0.0;
        for (int j = 0; j < n; ++j) {
            if (j == i || visited[j]) continue;
            double w = DIDX(W, n, i, j);
            if (bj < 0 || w < bw) { bj = j; bw = w; }
        }
        tour[len] = bj; visited[bj] = 1;
    }
    tour[n] = depot;
    free(visited);
}

typedef struct {
    double *cost; int cap; int len;   /* cost after every accepted move, in order */
} trace_t;

static void trace_push(trace_t *t, double c) {
    if (t && t->cost && t->len < t->cap) t->cost[t->len] = c;
    if (t) t->len++;
}

/* gnngls/algorithms.py:111-132.  tour is updated in place; returns the number of accepted moves. */
static int local_search_impl(int32_t *tour, double *cost, const double *D, int n, int first_improvement,
                             trace_t *tr, int64_t *evals) {
    int moves = 0, improved = 1;
    while (improved) {
        improved = 0;
        for (int op = 0; op < 2; ++op) {
            double delta; int bi = 0, bj = 0, found;
            if (op == 0) {
                found = gls_oracle_two_opt_a2a(tour, D, n, first_improvement, &delta, &bi, &bj);
                if (evals) *evals += (int64_t)(n - 2) * (n - 3) / 2;
            } else {
                found = gls_oracle_relocate_a2a(tour, D, n, first_improvement, &delta, &bi, &bj);
                if (evals) *evals += (int64_t)(n - 2) * (n - 2);
            }
            if (found && delta < 0) {
                improved = 1;
                *cost += delta;
                if (op == 0) gls_oracle_two_opt(tour, n, bi, bj); else gls_oracle_relocate(tour, n, bi, bj);
                trace_push(tr, *cost);
                moves++;
            }
        }
    }
    return moves;
}

int gls_oracle_local_search(int32_t *tour, double *cost, const double *D, int n, int first_improvement,
                            double *trace_cost, int trace_cap, int *trace_len) {
    trace_t tr = { trace_cost, trace_cap, 0 };
    int m = local_search_impl(tour, cost, D, n, first_improvement, &tr, NULL);
    if (trace_len) *trace_len = tr.len;
    return m;
}

static double now_s(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/*
 * gnngls/algorithms.py:135-195.
 *   D        n*n fp64 weights (edge_weight, :140)
 *   guides   n_guides matrices n*n fp64 (G.edges[e][guide], :155), cycled per outer iteration (:147)
 *   tour     in: init_tour, out: best_tour
 *   max_outer_iters >= 0 : run exactly that many outer iterations (deterministic / parity mode)
 *   max_outer_iters <  0 : run until time_limit_s of wall clock has elapsed (reference mode, :146)
 *   penalty_out (optional) n*n int32 final penalties
 *   imp_cost/imp_iter/imp_len (optional): the returned best and the number of completed outer iterations at every
 *            point where `best_cost` is (re)assigned (:143 -> iteration 0, :190-191 -> iter_i + 1), then one
 *            terminal entry (best_cost, outer iterations completed); *imp_len counts all of them
 * Returns best_cost.
 */
/* Diagnostics for kernel design (scripts/perturbation_stats.py): per penalty step, which of the four one-to-all scans
 * (2 endpoint + operator) accepted the step's FIRST move (index 4: none), and how many moves each scan accepted. */
static int64_t g_first_move_hist[5], g_scan_moves[4], g_steps;
void gls_oracle_perturbation_stats(int64_t *first_move_hist5, int64_t *scan_moves4, int64_t *steps, int reset) {
    for (int q = 0; q < 5; ++q) { if (first_move_hist5) first_move_hist5[q] = g_first_move_hist[q]; if (reset) g_first_move_hist[q] = 0; }
    for (int q = 0; q < 4; ++q) { if (scan_moves4) scan_moves4[q] = g_scan_moves[q]; if (reset) g_scan_moves[q] = 0; }
    if (steps) *steps = g_steps;
    if (reset) g_steps = 0;
}

static double gls_impl(const double *D, const double *guides, int n_guides, int n,
                                      int32_t *tour, double init_cost,
                                      int perturbation_moves, int first_improvement,
                                      int64_t max_outer_iters, double time_limit_s,
                                      double *trace_cost, int trace_cap, int *trace_len,
                                      int32_t *penalty_out, int64_t *outer_iters_out,
                                      int64_t *evals_out,
                                      double *imp_cost, int64_t *imp_iter, int imp_cap, int *imp_len, double *imp_time) {
    double t0 = now_s();
    size_t nn = (size_t)n * (size_t)n;
    double k = 0.1 * init_cost / (double)n;                      /* :137 (left to right) */
    double *pen = (double *)calloc(nn, sizeof(double));          /* :138 */
    double *Dg = (double *)malloc(nn * sizeof(double));
    int32_t *cur = (int32_t *)malloc((size_t)(n + 1) * sizeof(int32_t));
    trace_t tr = { trace_cost, trace_cap, 0 };
    int64_t evals = 0;

    memcpy(cur, tour, (size_t)(n + 1) * sizeof(int32_t));
    for (size_t q = 0; q < nn; ++q) Dg[q] = D[q] + k * 0.0;
    double cur_cost = init_cost;
    local_search_impl(cur, &cur_cost, D, n, first_improvement, &tr, &evals);   /* :142 */
    double best_cost = cur_cost;                                  /* :143 */
    memcpy(tour, cur, (size_t)(n + 1) * sizeof(int32_t));
    int n_imp = 0;
    if (imp_cost && n_imp < imp_cap) { imp_cost[n_imp] = best_cost; if (imp_iter) imp_iter[n_imp] = 0; if (imp_time) imp_time[n_imp] = now_s() - t0; }
    n_imp++;

    int64_t iter_i = 0;
    for (;;) {
        if (max_outer_iters >= 0) { if (iter_i >= max_outer_iters) break; }
        else if (!(now_s() - t0 < time_limit_s)) break;           /* :146 */
        const double *guide = guides + (size_t)(iter_i % n_guides) * nn;   /* :147 */

        int moves = 0;
        while (moves < perturbation_moves) {                      /* :151 */
            double max_util = 0.0; int me = -1;                   /* :153-159 */
            for (int p = 0; p < n; ++p) {
                int u = cur[p], v = cur[p + 1];
                double util = DIDX(guide, n, u, v) / (1.0 + DIDX(pen, n, u, v));
                if (util > max_util || me < 0) { max_util = util; me = p; }
            }
            int eu = cur[me], ev = cur[me + 1];
            DIDX(pen, n, eu, ev) += 1.0;                          /* :161 */
            DIDX(pen, n, ev, eu) = DIDX(pen, n, eu, ev);
            {   /* :163-164 `edge_weight + k * edge_penalties`, two roundings.  Only the two mirrored
                 * entries of the penalised edge change, every other entry keeps D + k*pen from before
                 * (initially D + k*0 == D bit-for-bit), so the rebuilt matrix is updated incrementally. */
                double kp = k * DIDX(pen, n, eu, ev);
                DIDX(Dg, n, eu, ev) = DIDX(D, n, eu, ev) + kp;
                DIDX(Dg, n, ev, eu) = DIDX(D, n, ev, eu) + kp;
            }
            int ends[2] = { eu, ev };
            int first_move = 4;
            for (int s = 0; s < 2; ++s) {                         /* :167 */
                int node = ends[s];
                if (node == 0) continue;                          /* :168 */
                int i = 0;
                while (cur[i] != node) ++i;                       /* :169 cur_tour.index(n) */
                for (int op = 0; op < 2; ++op) {                  /* :171 */
                    double delta; int bj = 0, found;
                    if (op == 0) { found = gls_oracle_two_opt_o2a(cur, Dg, n, i, first_improvement, &delta, &bj); evals += n - 3; }
                    else         { found = gls_oracle_relocate_o2a(cur, Dg, n, i, first_improvement, &delta, &bj); evals += n - 2; }
                    if (found && delta < 0) {                     /* :175 */
                        if (op == 0) gls_oracle_two_opt(cur, n, i, bj); else gls_oracle_relocate(cur, n, i, bj);
                        cur_cost = gls_oracle_tour_cost(cur, D, n);   /* :176 real weights */
                        trace_push(&tr, cur_cost);
                        moves += 1;                               /* :185 */
                        g_scan_moves[2 * s + op] += 1;
                        if (first_move == 4) first_move = 2 * s + op;
                    }
                }
            }
            g_first_move_hist[first_move] += 1; g_steps += 1;
        }
        local_search_impl(cur, &cur_cost, D, n, first_improvement, &tr, &evals);   /* :188 */
        if (cur_cost < best_cost) {                               /* :190-191 */
            best_cost = cur_cost;
            memcpy(tour, cur, (size_t)(n + 1) * sizeof(int32_t));
            if (imp_cost && n_imp < imp_cap) {
                imp_cost[n_imp] = best_cost; if (imp_iter) imp_iter[n_imp] = iter_i + 1; if (imp_time) imp_time[n_imp] = now_s() - t0;
            }
            n_imp++;
        }
        iter_i++;
    }
    if (penalty_out) for (size_t q = 0; q < nn; ++q) penalty_out[q] = (int32_t)pen[q];
    if (trace_len) *trace_len = tr.len;
    if (outer_iters_out) *outer_iters_out = iter_i;
    if (evals_out) *evals_out = evals;
    if (imp_cost && imp_cap > 0) {                                /* terminal entry: returned best, completed iterations */
        int q = n_imp < imp_cap ? n_imp : imp_cap - 1;
        imp_cost[q] = best_cost; if (imp_iter) imp_iter[q] = iter_i; if (imp_time) imp_time[q] = now_s() - t0;
    }
    n_imp++;
    if (imp_len) *imp_len = n_imp;
    free(pen); free(Dg); free(cur);
    return best_cost;
}

double gls_oracle_guided_local_search(const double *D, const double *guides, int n_guides, int n,
                                      int32_t *tour, double init_cost,
                                      int perturbation_moves, int first_improvement,
                                      int64_t max_outer_iters, double time_limit_s,
                                      double *trace_cost, int trace_cap, int *trace_len,
                                      int32_t *penalty_out, int64_t *outer_iters_out,
                                      int64_t *evals_out,
                                      double *imp_cost, int64_t *imp_iter, int imp_cap, int *imp_len) {
    return gls_impl(D, guides, n_guides, n, tour, init_cost, perturbation_moves, first_improvement, max_outer_iters,
                    time_limit_s, trace_cost, trace_cap, trace_len, penalty_out, outer_iters_out, evals_out,
                    imp_cost, imp_iter, imp_cap, imp_len, NULL);
}

/* The same search with the wall-clock time (seconds since the call started) of every improvement-trace entry in imp_time
 * [imp_cap] -- bench.py's cpu_baseline leg reports the CPU's gap-versus-budget curve from it (test.py:97-117). */
double gls_oracle_guided_local_search_timed(const double *D, const double *guides, int n_guides, int n,
                                            int32_t *tour, double init_cost,
                                            int perturbation_moves, int first_improvement,
                                            int64_t max_outer_iters, double time_limit_s,
                                            double *trace_cost, int trace_cap, int *trace_len,
                                            int32_t *penalty_out, int64_t *outer_iters_out,
                                            int64_t *evals_out,
                                            double *imp_cost, int64_t *imp_iter, int imp_cap, int *imp_len, double *imp_time) {
    return gls_impl(D, guides, n_guides, n, tour, init_cost, perturbation_moves, first_improvement, max_outer_iters,
                    time_limit_s, trace_cost, trace_cap, trace_len, penalty_out, outer_iters_out, evals_out,
                    imp_cost, imp_iter, imp_cap, imp_len, imp_time);
}

/* Full delta tables (parity with the K3u unit kernels): out[i*(n+1)+j] for 0<i,j<n, NaN elsewhere. */
void gls_oracle_two_opt_delta_all(const int32_t *tour, const double *D, int n, double *out) {
    for (int i = 0; i <= n; ++i) for (int j = 0; j <= n; ++j) {
        double v = NAN;
        if (i >= 1 && i <= n - 1 && j >= 1 && j <= n - 1) v = gls_oracle_two_opt_cost(tour, D, n, i, j);
        out[(size_t)i * (size_t)(n + 1) + (size_t)j] = v;
    }
}
void gls_oracle_relocate_delta_all(const int32_t *tour, const double *D, int n, double *out) {
    for (int i = 0; i <= n; ++i) for (int j = 0; j <= n; ++j) {
        double v = NAN;
        if (i >= 1 && i <= n - 1 && j >= 1 && j <= n - 1) v = gls_oracle_relocate_cost(tour, D, n, i, j);
        out[(size_t)i * (size_t)(n + 1) + (size_t)j] = v;
    }
}
