/* oracle/bnb_tsp.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Exact optimum of a symmetric TSP instance by branch and bound on the Held-Karp 1-tree bound (Held & Karp 1971;
 * branching and edge elimination after Volgenant & Jonker 1982) -- the denominator the reference's gap uses
 * (scripts/test.py:62,104 divides by the Concorde optimum stored in its instance files, gnngls/__init__.py:55-60; those
 * files are git-LFS stubs and Concorde is not installed).  The checker side of bench.py's gap report only: it certifies,
 * for a sample of the seeded TSP100 instances, that no tour shorter than the best-known one exists (or finds it).
 *
 * Subproblem = sets of required / forbidden edges.  Bound: max over a short subgradient ascent of
 *      w(pi) = min 1-tree over (c + pi_i + pi_j) that contains the required and avoids the forbidden edges  -  2 sum(pi),
 * a lower bound of every tour of the subproblem for ANY pi.  With the minimum 1-tree T at the best pi, an edge e outside T
 * costs at least  c_pi(e) - (heaviest edge of the cycle it closes in T)  more: edges whose marginal bound reaches the
 * incumbent are forbidden in the subtree (most edges, at the root already).  Branch on a node of degree > 2 in T with free
 * tree edges e1, e2: {forbid e1}, {require e1, forbid e2}, {require e1, require e2} (a node with two required edges has all
 * its other edges forbidden).  A subproblem is closed when its bound reaches incumbent * (1 - 1e-12) or T is a tour.
 *
 * The incumbent starts at the caller's tour length (ub): "proven" = the search finished within the node limit, the result
 * is then min(ub, best tour found) = the optimum (costs are compared with a 1e-12 relative slack: fp64 sums of ~n terms).
 */
#include <float.h>
#include <stdlib.h>
#include <string.h>

#define FREE 0
#define REQ 1
#define FORB 2

typedef struct {
    int n;
    const double *c;
    unsigned char *st;                  /* [n,n] symmetric edge states */
    double ub;                          /* incumbent (length of the best tour known) */
    int *best_tour; int have_tour;      /* tour found by the search itself (when shorter than the caller's ub) */
    long nodes, max_nodes;
    int aborted;
    /* undo log of state changes */
    int *log_i, *log_j; unsigned char *log_old; long log_len, log_cap;
    /* scratch */
    double *key; int *parent, *deg; char *in; double *pmax; int *order;
} Ctx;

static void set_state(Ctx *x, int i, int j, unsigned char s) {
    const int n = x->n;
    if (x->log_len == x->log_cap) {
        x->log_cap *= 2;
        x->log_i = (int *)realloc(x->log_i, (size_t)x->log_cap * sizeof(int));
        x->log_j = (int *)realloc(x->log_j, (size_t)x->log_cap * sizeof(int));
        x->log_old = (unsigned char *)realloc(x->log_old, (size_t)x->log_cap);
    }
    x->log_i[x->log_len] = i; x->log_j[x->log_len] = j; x->log_old[x->log_len] = x->st[(size_t)i * n + j]; x->log_len++;
    x->st[(size_t)i * n + j] = s; x->st[(size_t)j * n + i] = s;
}
static void rollback(Ctx *x, long mark) {
    const int n = x->n;
    while (x->log_len > mark) {
        x->log_len--;
        const int i = x->log_i[x->log_len], j = x->log_j[x->log_len];
        x->st[(size_t)i * n + j] = x->log_old[x->log_len]; x->st[(size_t)j * n + i] = x->log_old[x->log_len];
    }
}

/* degree rules: a node with two required edges loses its other edges; a node with fewer than two usable edges, or three
 * required ones, closes the subproblem.  Returns 0 if infeasible. */
static int propagate(Ctx *x) {
    const int n = x->n;
    for (int again = 1; again;) {
        again = 0;
        for (int i = 0; i < n; ++i) {
            int req = 0, usable = 0;
            const unsigned char *row = x->st + (size_t)i * n;
            for (int j = 0; j < n; ++j) if (j != i) { if (row[j] == REQ) req++; if (row[j] != FORB) usable++; }
            if (req > 2 || usable < 2) return 0;
            if (req == 2 && usable > 2) {
                for (int j = 0; j < n; ++j) if (j != i && row[j] == FREE) set_state(x, i, j, FORB);
                again = 1;
            } else if (usable == 2 && req < 2) {
                for (int j = 0; j < n; ++j) if (j != i && row[j] == FREE) set_state(x, i, j, REQ);
                again = 1;
            }
        }
    }
    /* required edges must not close a cycle shorter than n */
    int *comp = x->order;
    for (int i = 0; i < n; ++i) comp[i] = i;
    int nreq = 0;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j)
            if (x->st[(size_t)i * n + j] == REQ) {
                int a = i, b = j;
                while (comp[a] != a) a = comp[a];
                while (comp[b] != b) b = comp[b];
                nreq++;
                if (a == b) { if (nreq < n) return 0; } else comp[a] = b;
            }
    return 1;
}

#define BIGM 1.0e6
/* minimum constrained 1-tree for c + pi_i + pi_j: Prim on nodes 1..n-1, then the two best edges at node 0.  Required edges
 * are shifted by -BIGM (taken first), forbidden ones skipped.  Returns the TRUE modified cost (no shift) or DBL_MAX if no
 * such 1-tree exists; fills deg[] and parent[] (parent[0] unused; a0 / b0 = the two neighbours of node 0). */
static double one_tree(Ctx *x, const double *pi, int *a0, int *b0) {
    const int n = x->n;
    const double *c = x->c;
    double *key = x->key; int *parent = x->parent, *deg = x->deg; char *in = x->in;
    for (int i = 0; i < n; ++i) { deg[i] = 0; in[i] = 0; key[i] = DBL_MAX; parent[i] = -1; }
    key[1] = -DBL_MAX;
    double total = 0.0;
    for (int it = 1; it < n; ++it) {
        int u = -1;
        for (int v = 1; v < n; ++v) if (!in[v] && (u < 0 || key[v] < key[u])) u = v;
        if (key[u] == DBL_MAX) return DBL_MAX;                 /* disconnected by the forbidden edges */
        in[u] = 1;
        if (parent[u] >= 0) { total += c[(size_t)u * n + parent[u]] + pi[u] + pi[parent[u]]; deg[u]++; deg[parent[u]]++; }
        const double *cu = c + (size_t)u * n;
        const unsigned char *su = x->st + (size_t)u * n;
        for (int v = 1; v < n; ++v) {
            if (in[v] || su[v] == FORB) continue;
            const double w = cu[v] + pi[u] + pi[v] - (su[v] == REQ ? BIGM : 0.0);
            if (w < key[v]) { key[v] = w; parent[v] = u; }
        }
    }
    double m1 = DBL_MAX, m2 = DBL_MAX; int a1 = -1, a2 = -1;
    for (int v = 1; v < n; ++v) {
        if (x->st[v] == FORB) continue;
        const double w = c[v] + pi[0] + pi[v] - (x->st[v] == REQ ? BIGM : 0.0);
        if (w < m1) { m2 = m1; a2 = a1; m1 = w; a1 = v; } else if (w < m2) { m2 = w; a2 = v; }
    }
    if (a2 < 0) return DBL_MAX;
    total += c[a1] + pi[0] + pi[a1] + c[a2] + pi[0] + pi[a2];
    deg[0] = 2; deg[a1]++; deg[a2]++;
    *a0 = a1; *b0 = a2;
    /* every required edge must be in the 1-tree (else the requirements are not satisfiable by any 1-tree) */
    for (int i = 1; i < n; ++i)
        for (int j = i + 1; j < n; ++j)
            if (x->st[(size_t)i * n + j] == REQ && parent[i] != j && parent[j] != i) return DBL_MAX;
    for (int v = 1; v < n; ++v) if (x->st[v] == REQ && v != a1 && v != a2) return DBL_MAX;
    return total;
}

static void solve(Ctx *x, const double *pi_in, int depth) {
    const int n = x->n;
    if (x->aborted) return;
    if (++x->nodes > x->max_nodes) { x->aborted = 1; return; }
    const long mark = x->log_len;
    if (!propagate(x)) { rollback(x, mark); return; }
    double *pi = (double *)malloc((size_t)n * sizeof(double)), *best_pi = (double *)malloc((size_t)n * sizeof(double));
    memcpy(pi, pi_in, (size_t)n * sizeof(double)); memcpy(best_pi, pi_in, (size_t)n * sizeof(double));
    const double slack = 1e-12 * (x->ub > 1.0 ? x->ub : 1.0);
    double best = -DBL_MAX, lambda = depth == 0 ? 2.0 : 0.6;
    const int iters = depth == 0 ? 4000 : 30 + n / 2;
    const int period = depth == 0 ? n / 2 : 8;
    int stall = 0, a0 = -1, b0 = -1, closed = 0;
    for (int it = 0; it < iters && lambda > 1e-6; ++it) {
        double sum_pi = 0.0;
        for (int i = 0; i < n; ++i) sum_pi += pi[i];
        const double t = one_tree(x, pi, &a0, &b0);
        if (t == DBL_MAX) { closed = 1; break; }
        const double w = t - 2.0 * sum_pi;
        if (w > best) { best = w; stall = 0; memcpy(best_pi, pi, (size_t)n * sizeof(double)); } else stall++;
        if (best >= x->ub - slack) { closed = 1; break; }
        long norm = 0;
        for (int i = 0; i < n; ++i) norm += (long)(x->deg[i] - 2) * (x->deg[i] - 2);
        if (norm == 0) {                                       /* the 1-tree is a tour: the optimum of this subproblem */
            double len = 0.0;
            int *tour = x->order, k = 0;
            /* adjacency of the tour from parent[] and node 0's two edges (pmax doubles as scratch: 2 ints per node) */
            int *adj = (int *)x->pmax;
            for (int v = 0; v < n; ++v) { adj[2 * v] = -1; adj[2 * v + 1] = -1; }
#define ADD_ADJ(p, q) do { if (adj[2 * (p)] < 0) adj[2 * (p)] = (q); else adj[2 * (p) + 1] = (q); } while (0)
            for (int v = 1; v < n; ++v) if (x->parent[v] >= 0) { ADD_ADJ(v, x->parent[v]); ADD_ADJ(x->parent[v], v); }
            ADD_ADJ(0, a0); ADD_ADJ(a0, 0); ADD_ADJ(0, b0); ADD_ADJ(b0, 0);
#undef ADD_ADJ
            int prev = -1, cur = 0;
            while (k < n) {
                tour[k++] = cur;
                const int next = adj[2 * cur] != prev ? adj[2 * cur] : adj[2 * cur + 1];
                prev = cur; cur = next;
                if (cur <= 0) break;
            }
            if (k == n && cur == 0) {
                for (int q = 0; q < n; ++q) len += x->c[(size_t)tour[q] * n + tour[(q + 1) % n]];
                if (len < x->ub - slack) { x->ub = len; memcpy(x->best_tour, tour, (size_t)n * sizeof(int)); x->have_tour = 1; }
                closed = 1;
            }
            break;
        }
        if (stall >= period) { lambda *= 0.5; stall = 0; memcpy(pi, best_pi, (size_t)n * sizeof(double)); continue; }
        const double gap = x->ub - w > 1e-9 ? x->ub - w : 1e-9;
        const double step = lambda * gap / (double)norm;
        for (int i = 0; i < n; ++i) pi[i] += step * (double)(x->deg[i] - 2);
    }
    if (!closed && best >= x->ub - slack) closed = 1;
    if (!closed) {
        /* the minimum 1-tree at the best multipliers: edge elimination, then the branching node */
        double sum_pi = 0.0;
        for (int i = 0; i < n; ++i) sum_pi += best_pi[i];
        const double t = one_tree(x, best_pi, &a0, &b0);
        if (t == DBL_MAX) closed = 1;
        else {
            const double w = t - 2.0 * sum_pi;
            const double room = x->ub - slack - w;             /* an edge whose marginal cost reaches this cannot be in a better tour */
            /* pmax[u][v] = heaviest modified edge on the tree path u .. v (nodes 1..n-1): one traversal per source */
            int *par = x->parent;
            double *pm = x->pmax;
            int *stack = x->order;
            /* children lists on the fly: O(n^2) per source is fine at n ~ 100 */
            for (int u = 1; u < n; ++u) {
                for (int v = 1; v < n; ++v) pm[(size_t)u * n + v] = -1.0;
                pm[(size_t)u * n + u] = -DBL_MAX;
                int sp = 0; stack[sp++] = u;
                while (sp) {
                    const int a = stack[--sp];
                    for (int b = 1; b < n; ++b) {
                        if (pm[(size_t)u * n + b] != -1.0 || !(par[b] == a || par[a] == b)) continue;
                        const double e = x->c[(size_t)a * n + b] + best_pi[a] + best_pi[b];
                        const double m = pm[(size_t)u * n + a];
                        pm[(size_t)u * n + b] = e > m ? e : m;
                        stack[sp++] = b;
                    }
                }
            }
            const double e0b = x->c[b0] + best_pi[0] + best_pi[b0], e0a = x->c[a0] + best_pi[0] + best_pi[a0];
            const double second0 = e0a > e0b ? e0a : e0b;
            for (int u = 0; u < n && !x->aborted; ++u)
                for (int v = u + 1; v < n; ++v) {
                    if (x->st[(size_t)u * n + v] != FREE) continue;
                    double marg;
                    if (u == 0) { if (v == a0 || v == b0) continue; marg = x->c[v] + best_pi[0] + best_pi[v] - second0; }
                    else { if (par[u] == v || par[v] == u) continue; marg = x->c[(size_t)u * n + v] + best_pi[u] + best_pi[v] - pm[(size_t)u * n + v]; }
                    if (marg >= room) set_state(x, u, v, FORB);
                }
            /* branching node: degree > 2 in the 1-tree, as few free tree edges as possible (>= 1) */
            int bv = -1, bfree = 1 << 30, f1 = -1, f2 = -1;
            for (int v = 0; v < n; ++v) {
                if (x->deg[v] <= 2) continue;
                int cnt = 0, e1 = -1, e2 = -1;
                for (int u = 0; u < n; ++u) {
                    if (u == v) continue;
                    const int in_tree = v == 0 ? (u == a0 || u == b0) : u == 0 ? (v == a0 || v == b0) : (par[u] == v || par[v] == u);
                    if (in_tree && x->st[(size_t)v * n + u] == FREE) { cnt++; if (e1 < 0) e1 = u; else if (e2 < 0) e2 = u; }
                }
                if (cnt >= 1 && cnt < bfree) { bfree = cnt; bv = v; f1 = e1; f2 = e2; }
            }
            if (bv < 0) closed = 1;                            /* (cannot happen for a non-tour 1-tree with free edges; be safe) */
            else {
                int req = 0;
                for (int u = 0; u < n; ++u) if (u != bv && x->st[(size_t)bv * n + u] == REQ) req++;
                const long m2 = x->log_len;
                /* child 1: both edges required (only if the node can still take two) */
                if (f2 >= 0 && req == 0) { set_state(x, bv, f1, REQ); set_state(x, bv, f2, REQ); solve(x, best_pi, depth + 1); rollback(x, m2); }
                /* child 2: e1 required, e2 forbidden */
                if (f2 >= 0) { set_state(x, bv, f1, REQ); set_state(x, bv, f2, FORB); solve(x, best_pi, depth + 1); rollback(x, m2); }
                else { set_state(x, bv, f1, REQ); solve(x, best_pi, depth + 1); rollback(x, m2); }
                /* child 3: e1 forbidden */
                set_state(x, bv, f1, FORB); solve(x, best_pi, depth + 1); rollback(x, m2);
            }
        }
    }
    free(pi); free(best_pi);
    rollback(x, mark);
}

/* c [n,n] symmetric, ub = length of a known tour.  Returns the optimum if *proven = 1 (search completed within max_nodes),
 * else the best upper bound it has (<= ub).  *nodes = subproblems visited; tour_out (n ints or NULL) receives a tour of that
 * length only if the search found one shorter than ub (*found = 1). */
double bnb_tsp(const double *c, int n, double ub, long max_nodes, int *proven, long *nodes, int *tour_out, int *found) {
    Ctx x;
    memset(&x, 0, sizeof(x));
    x.n = n; x.c = c; x.ub = ub; x.max_nodes = max_nodes;
    x.st = (unsigned char *)calloc((size_t)n * n, 1);
    x.best_tour = (int *)malloc((size_t)n * sizeof(int));
    x.log_cap = 1 << 16;
    x.log_i = (int *)malloc((size_t)x.log_cap * sizeof(int)); x.log_j = (int *)malloc((size_t)x.log_cap * sizeof(int));
    x.log_old = (unsigned char *)malloc((size_t)x.log_cap);
    x.key = (double *)malloc((size_t)n * sizeof(double)); x.parent = (int *)malloc((size_t)n * sizeof(int));
    x.deg = (int *)malloc((size_t)n * sizeof(int)); x.in = (char *)malloc((size_t)n);
    x.pmax = (double *)malloc((size_t)n * n * sizeof(double)); x.order = (int *)malloc((size_t)(n + 1) * sizeof(int));
    double *pi0 = (double *)calloc((size_t)n, sizeof(double));
    if (n >= 3) solve(&x, pi0, 0);
    if (proven) *proven = !x.aborted;
    if (nodes) *nodes = x.nodes;
    if (found) *found = x.have_tour;
    if (tour_out && x.have_tour) memcpy(tour_out, x.best_tour, (size_t)n * sizeof(int));
    const double r = x.ub;
    free(pi0); free(x.st); free(x.best_tour); free(x.log_i); free(x.log_j); free(x.log_old);
    free(x.key); free(x.parent); free(x.deg); free(x.in); free(x.pmax); free(x.order);
    return r;
}
