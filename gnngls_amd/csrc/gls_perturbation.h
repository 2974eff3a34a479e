// gls_perturbation.h -- part of gls_kernels.hip (one translation unit; included inside namespace gnngls, in this order:
// gls_common.h, gls_descent_scans.h, gls_perturbation.h).  One-to-all scans on the guided matrix, move application, and the three forms of the perturbation phase (algorithms.py:150-185): scan-by-scan pieces, team form, edge form.
#pragma once

// o2a scans with an arbitrary distance functor (guided matrix in the perturbation phase).
template <class F, bool FI, class TT>
__device__ __forceinline__ void scan_two_opt_o2a(const TT *t, const F &f, int n, int i,
                                                 int tid, int nthr, double &bd, int &bk) {
    for (int j = 1 + tid; j <= n - 1; j += nthr) {
        int dj = i - j; if (dj < 0) dj = -dj;
        if (dj < 2) continue;                                // operators.py:61-62
        consider<FI>(two_opt_cost(t, f, i, j), j, bd, bk);
    }
}
template <class F, bool FI, class TT>
__device__ __forceinline__ void scan_relocate_o2a(const TT *t, const F &f, int n, int i,
                                                  int tid, int nthr, double &bd, int &bk) {
    for (int j = 1 + tid; j <= n - 1; j += nthr) {
        if (j == i) continue;                                // operators.py:114-115
        consider<FI>(relocate_cost(t, f, i, j), j, bd, bk);
    }
}

// Guided one-to-all scans of the perturbation phase (algorithms.py:171-174 on edge_weight + k*penalties).
// Same arithmetic as two_opt_cost / relocate_cost with GuidedDist, but all penalty and distance loads of an
// evaluation are issued back to back BEFORE any arithmetic: written through the generic functor the compiler
// (scheduling for minimum register pressure) emits load -> wait -> use six times in a row, and each wait is a
// full L1/L2 (penalties) or LDS (distances) round trip on the serial chain of the search.
// j0 / jstep: the default walks all j = 1 .. n-1 in passes of 64 lanes; the team form of the perturbation phase
// (team_perturbation below) gives every wavefront ONE pass (j0 = 1 + 64 pass, jstep >= n).
// KNOWN (team form): the counter at packed index qk is pk whatever the load returns -- the edge a penalty step has just
// incremented, whose store by another wavefront may or may not have landed yet.
template <class S, bool FI, class TT, bool KNOWN = false>
__device__ __forceinline__ void scan_two_opt_o2a_guided(const S &s, double k, const TT *t, int n, int i,
                                                        int lane, double &bd, int &bk, int j0 = 1, int jstep = kWave,
                                                        int qk = -1, int pk = 0) {
    for (int j = j0 + lane; j <= n - 1; j += jstep) {
        int dj = i - j; if (dj < 0) dj = -dj;
        if (dj < 2) continue;                                // operators.py:61-62
        const int ii = i < j ? i : j, jj = i < j ? j : i;    // operators.py:17-18
        const int a = t[ii], b = t[ii - 1], c = t[jj], d = t[jj - 1];
        const int q0 = s.idx(a, c), q1 = s.idx(b, d), q2 = s.idx(a, b), q3 = s.idx(c, d);
        int p0 = s.pen_at(q0), p1 = s.pen_at(q1), p2 = s.pen_at(q2), p3 = s.pen_at(q3);
        const double d0 = s.dist_at(q0), d1 = s.dist_at(q1), d2 = s.dist_at(q2), d3 = s.dist_at(q3);
        if constexpr (KNOWN) { p0 = q0 == qk ? pk : p0; p1 = q1 == qk ? pk : p1; p2 = q2 == qk ? pk : p2; p3 = q3 == qk ? pk : p3; }
        const double g0 = d0 + k * (double)p0, g1 = d1 + k * (double)p1;   // [exact] product rounded, then sum
        const double g2 = d2 + k * (double)p2, g3 = d3 + k * (double)p3;
        double delta = g0 + g1;                              // operators.py:25-28, left to right
        delta = delta - g2;
        delta = delta - g3;
        consider<FI>(delta, j, bd, bk);
    }
}

template <class S, bool FI, class TT, bool KNOWN = false>
__device__ __forceinline__ void scan_relocate_o2a_guided(const S &s, double k, const TT *t, int n, int i,
                                                         int lane, double &bd, int &bk, int j0 = 1, int jstep = kWave,
                                                         int qk = -1, int pk = 0) {
    const int a = t[i - 1], b = t[i], c = t[i + 1];
    const int qab = s.idx(a, b), qbc = s.idx(b, c), qac = s.idx(a, c);
    int pab = s.pen_at(qab), pbc = s.pen_at(qbc), pac = s.pen_at(qac);
    if constexpr (KNOWN) { pab = qab == qk ? pk : pab; pbc = qbc == qk ? pk : pbc; pac = qac == qk ? pk : pac; }
    const double gab = s.dist_at(qab) + k * (double)pab;
    const double gbc = s.dist_at(qbc) + k * (double)pbc;
    const double gac = s.dist_at(qac) + k * (double)pac;
    double base = -gab;                                      // operators.py:97-99, left to right
    base = base - gbc;
    base = base + gac;
    for (int j = j0 + lane; j <= n - 1; j += jstep) {
        if (j == i) continue;                                // operators.py:114-115
        int d, e;
        if (i < j) { d = t[j]; e = t[j + 1]; } else { d = t[j - 1]; e = t[j]; }
        const int q0 = s.idx(d, e), q1 = s.idx(d, b), q2 = s.idx(b, e);
        int p0 = s.pen_at(q0), p1 = s.pen_at(q1), p2 = s.pen_at(q2);
        const double d0 = s.dist_at(q0), d1 = s.dist_at(q1), d2 = s.dist_at(q2);
        if constexpr (KNOWN) { p0 = q0 == qk ? pk : p0; p1 = q1 == qk ? pk : p1; p2 = q2 == qk ? pk : p2; }
        const double g0 = d0 + k * (double)p0, g1 = d1 + k * (double)p1, g2 = d2 + k * (double)p2;
        double delta = base - g0;                            // operators.py:100-102
        delta = delta + g1;
        delta = delta + g2;
        consider<FI>(delta, j, bd, bk);
    }
}

// new tour + edge arrays in one pass; caller synchronises afterwards.
// pos (descent with node-indexed relocate lanes): node -> position table, kept current here.
// ppos / ctl / Lmax (descent with pruned scans): the same table in the tour's element type and the upper bound of Ef[].
template <class S, class TT>
__device__ __forceinline__ void apply_move(const S &s, const TT *told, TT *tnew, double *Ef, double *Eb,
                                           int n, int op, int i, int j, int tid, int nthr, bool want_edges,
                                           uint8_t *pos = nullptr, TT *ppos = nullptr, Ctl *ctl = nullptr, double Lmax = 0.0) {
    for (int p = tid; p <= n; p += nthr) {
        int np = told[move_src(op, p, i, j)];
        tnew[p] = (TT)np;
        if (pos && p < n) pos[np] = (uint8_t)p;
        if (ppos && p < n) ppos[np] = (TT)p;
        if (want_edges && p >= 1) {
            int nq = told[move_src(op, p - 1, i, j)];
            const double e = s.dist(nq, np);
            Ef[p] = e;
            if (!S::kSymmetric) Eb[p] = s.dist(np, nq);
            if (ppos && e > Lmax) lmax_raise(ctl, e);
        }
    }
}

template <class S, class TT>
__device__ __forceinline__ void build_edges(const S &s, const TT *t, double *Ef, double *Eb, int n,
                                            int tid, int nthr) {
    for (int p = 1 + tid; p <= n; p += nthr) {
        int u = t[p - 1], v = t[p];
        Ef[p] = s.dist(u, v);
        if (!S::kSymmetric) Eb[p] = s.dist(v, u);
    }
}

// [exact] tour_cost (__init__.py:17-21): c = 0; c += w left to right.  Ef must be current.
__device__ __forceinline__ double tour_cost_from_edges(const double *Ef, int n) {
    double c = 0.0;
#pragma unroll 8
    for (int p = 1; p <= n; ++p) c += Ef[p];      // the adds stay in order; the LDS reads of 8 steps overlap
    return c;
}

// One pass (j = j0 + lane) of the guided one-to-all scans for the row-major penalty matrix of TriDGlobalPF: same operands,
// same arithmetic as scan_*_o2a_guided; a counter is read at [wave-uniform node][the lane's node] wherever the pair has a
// wave-uniform node.  (qk1, qk2) are the two matrix cells of the edge this penalty step incremented, pk its new count.
// eu, ev: the nodes of that edge.  Which counters can BE that edge is mostly decided on the scalar unit (GLS_TEAM_NODE_SUBST):
// a pair with a wave-uniform node matches iff its other node is the edge's other endpoint -- one vector compare instead of
// two per pair, none for the pairs of two uniform nodes; only the lane's own tour edge keeps the two-cell test.
template <class S, bool FI, class TT>
__device__ __forceinline__ void scan_two_opt_o2a_guided_rm(const S &s, double k, const TT *t, int n, int i, int j,
                                                           int qk1, int qk2, int pk, int eu, int ev, double &bd, int &bk) {
    int dj = i - j; if (dj < 0) dj = -dj;
    if (j > n - 1 || dj < 2) return;                         // operators.py:61-62
    const bool lt = i < j;
    // {a,c} = {t[i],t[j]} and {b,d} = {t[i-1],t[j-1]} whichever of i, j is smaller; the removed edges are the scan's own
    // (t[i-1],t[i]) and the lane's (t[j-1],t[j]), only their order in the sum depends on i < j (operators.py:17-18,25-28)
    const int ti = t[i], tim = t[i - 1], tj = t[j], tjm = t[j - 1];
    const int r0 = ti * n + tj;                              // row of t[i]
    const int r1 = tim * n + tjm;                            // row of t[i-1]
    const int ru = ti * n + tim, rl = tj * n + tjm;          // the two tour edges (ru wave-uniform)
    int p0 = s.cell(r0), p1 = s.cell(r1), pu = s.cell(ru), pl = s.cell(rl);
    const double d0 = s.dist(ti, tj), d1 = s.dist(tim, tjm), du = s.dist(ti, tim), dl = s.dist(tj, tjm);
    // all counter loads in flight before the first one is consumed: left alone, the compiler sinks each load to its
    // substitution below and waits for it there -- four (relocate: three) memory round trips in a row per unit
    asm volatile("" : "+v"(p0), "+v"(p1), "+v"(pu), "+v"(pl));
#if GLS_TEAM_NODE_SUBST
    {
        const int ti_u = __builtin_amdgcn_readfirstlane(ti), tim_u = __builtin_amdgcn_readfirstlane(tim);
        const int oth_i = ti_u == eu ? ev : (ti_u == ev ? eu : -1);          // the node that makes {t[i], x} the incremented edge
        const int oth_im = tim_u == eu ? ev : (tim_u == ev ? eu : -1);
        p0 = tj == oth_i ? pk : p0;                          // {t[i], t[j]}
        p1 = tjm == oth_im ? pk : p1;                        // {t[i-1], t[j-1]}
        pu = tim_u == oth_i ? pk : pu;                       // {t[i], t[i-1]}: both uniform
        pl = (rl == qk1 || rl == qk2) ? pk : pl;             // the lane's tour edge
        (void)r0; (void)r1; (void)ru;
    }
#else
    (void)eu; (void)ev;
    p0 = (r0 == qk1 || r0 == qk2) ? pk : p0; p1 = (r1 == qk1 || r1 == qk2) ? pk : p1;
    pu = (ru == qk1 || ru == qk2) ? pk : pu; pl = (rl == qk1 || rl == qk2) ? pk : pl;
#endif
    const double g0 = d0 + k * (double)p0, g1 = d1 + k * (double)p1;   // [exact] product rounded, then sum
    const double gu = du + k * (double)pu, gl = dl + k * (double)pl;
    double delta = g0 + g1;                                  // operators.py:25-28, left to right
    delta = delta - (lt ? gu : gl);                          // - G[a,b]
    delta = delta - (lt ? gl : gu);                          // - G[c,d]
    consider<FI>(delta, j, bd, bk);
}

template <class S, bool FI, class TT>
__device__ __forceinline__ void scan_relocate_o2a_guided_rm(const S &s, double k, const TT *t, int n, int i, int j,
                                                            int qk1, int qk2, int pk, int eu, int ev, double &bd, int &bk) {
    const int a = t[i - 1], b = t[i], c = t[i + 1];
    const int rab = b * n + a, rbc = b * n + c, rac = a * n + c;
    int pab = s.cell(rab), pbc = s.cell(rbc), pac = s.cell(rac);
    const bool live = j <= n - 1 && j != i;                  // operators.py:114-115
    const int jc = live ? j : (i == 1 ? 2 : 1);
    int d, e;
    if (i < jc) { d = t[jc]; e = t[jc + 1]; } else { d = t[jc - 1]; e = t[jc]; }
    const int r0 = d * n + e, r1 = b * n + d, r2 = b * n + e;        // {d,e}: the lane's tour edge; {d,b}, {b,e}: row of b
    int p0 = s.cell(r0), p1 = s.cell(r1), p2 = s.cell(r2);
    const double dab = s.dist(a, b), dbc = s.dist(b, c), dac = s.dist(a, c);
    const double d0 = s.dist(d, e), d1 = s.dist(d, b), d2 = s.dist(b, e);
    asm volatile("" : "+v"(pab), "+v"(pbc), "+v"(pac), "+v"(p0), "+v"(p1), "+v"(p2));      // see scan_two_opt_o2a_guided_rm
#if GLS_TEAM_NODE_SUBST
    {
        const int a_u = __builtin_amdgcn_readfirstlane(a), b_u = __builtin_amdgcn_readfirstlane(b), c_u = __builtin_amdgcn_readfirstlane(c);
        const int oth_b = b_u == eu ? ev : (b_u == ev ? eu : -1);
        const int oth_a = a_u == eu ? ev : (a_u == ev ? eu : -1);
        pab = a_u == oth_b ? pk : pab; pbc = c_u == oth_b ? pk : pbc; pac = c_u == oth_a ? pk : pac;      // uniform pairs
        p0 = (r0 == qk1 || r0 == qk2) ? pk : p0;             // the lane's tour edge {d, e}
        p1 = d == oth_b ? pk : p1;                           // {d, b}
        p2 = e == oth_b ? pk : p2;                           // {b, e}
        (void)rab; (void)rbc; (void)rac; (void)r1; (void)r2;
    }
#else
    (void)eu; (void)ev;
    pab = (rab == qk1 || rab == qk2) ? pk : pab; pbc = (rbc == qk1 || rbc == qk2) ? pk : pbc; pac = (rac == qk1 || rac == qk2) ? pk : pac;
    p0 = (r0 == qk1 || r0 == qk2) ? pk : p0; p1 = (r1 == qk1 || r1 == qk2) ? pk : p1; p2 = (r2 == qk1 || r2 == qk2) ? pk : p2;
#endif
    const double gab = dab + k * (double)pab, gbc = dbc + k * (double)pbc, gac = dac + k * (double)pac;
    double base = -gab;                                      // operators.py:97-99, left to right
    base = base - gbc;
    base = base + gac;
    const double g0 = d0 + k * (double)p0, g1 = d1 + k * (double)p1, g2 = d2 + k * (double)p2;
    double delta = base - g0;                                // operators.py:100-102
    delta = delta + g1;
    delta = delta + g2;
    if (live) consider<FI>(delta, j, bd, bk);
}

// ---------------------------------------------------------------------------------------------
// Team form of the perturbation phase (algorithms.py:150-185) for workgroups that own their CU
// ---------------------------------------------------------------------------------------------
// The default form runs the phase on wavefront 0 while the other wavefronts of the workgroup park on a barrier: right
// when the CU is shared by four workgroups (TSP100 x 1024: their descents fill the SIMDs), wasteful when the workgroup
// has the CU to itself -- TSP200 (159 KB distance triangle, ONE 16-wave workgroup per CU) spends three quarters of an
// outer iteration in this phase with 15 of 16 wavefronts idle.  Here every wavefront takes part:
//   * a penalty step's four one-to-all scans (two endpoints x {two_opt_o2a, relocate_o2a}, algorithms.py:167-174) times
//     their P = ceil((n-1)/64) passes of 64 lanes are 4 P independent UNITS, evaluated concurrently on the current tour
//     (unit u -> wavefront u mod nwaves); each writes its (delta, j) candidate to an LDS slot; after ONE barrier every
//     thread reads the slots in the reference's order (endpoint, operator, pass) and finds the first scan with an improving
//     move.  Scans behind it were speculative: the move changes the tour, so they are evaluated again in the next round
//     (most scans find no move, so a step usually takes one or two rounds instead of 4 P sequential passes);
//   * the index of an endpoint is taken once per endpoint on the tour at that moment and reused by relocate_o2a after
//     two_opt_o2a changed the tour (algorithms.py:169-174) exactly as in the serial form: i of endpoint 0 is the arg-max
//     position, i of endpoint 1 is looked up at the start of every round in which endpoint 1 has not started yet;
//   * the utilities of the tour edges (algorithms.py:153-159) are cached by position: wavefront q holds positions
//     64 q .. 64 q + 63, the partial arg-max of each goes through LDS (first maximum wins: slots are combined in
//     position order with a strict >), the lane that caches the winning edge stores its incremented counter.
// Same arithmetic, same candidates, same order of consumption as the serial form: all results stay bit-exact.
struct TeamCtl {
    double arg_u[4]; int arg_p[4];       // partial arg-max of the utilities, per block of 64 tour positions
    int arg_c[4];                        // penalty counter of that block's arg-max edge
    double res_d[2][16]; int res_k[2][16];   // candidate of unit (scan, pass): scan = 2 endpoint + operator; two sets, by round parity
    int stop; int pad[3];
};

// Workgroup barrier that orders LDS accesses only: global loads in flight (the asynchronous reloads of the cached
// utilities after a move: guide matrix entries from L2 / HBM) keep flying, where __syncthreads() would wait for them.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <class S, bool FI, class TT, class TRC>
__device__ __forceinline__ void team_perturbation(const S &s, const double k, TT *&t, TT *&t2, double *Ef, double *Eb,
                                                  const int n, TeamCtl *tc, const double *guide, const GlsArgs &A,
                                                  const long long t_start, const bool eager_cost, double &cur_cost,
                                                  TRC &tr, long long &evals, int &status, Stamps &st) {
    static_assert(sizeof(typename S::pen_t) == 4, "team form: 32-bit penalty counters only");
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nthr >> 6;
    const int P = (n - 2 + kWave) / kWave;                   // passes of 64 lanes over j = 1 .. n-1
    const int units = 4 * P;                                 // n <= 255: at most 16
    const int NQ = (n + kWave - 1) / kWave;                  // blocks of 64 tour positions 0 .. n-1
    double gq = 0.0; int pq = 0;                             // utility numerator and penalty of tour edge (p, p+1), p = 64 wave + lane
    const int myp = wave * kWave + lane;
    // (qk, qk2, pk): the counter this step incremented (packed index / both matrix cells, new count) -- substituted for
    // whatever the load returns, as in the scans: another wavefront's store of it may still be in flight
    auto reload = [&](int qk, int qk2, int pk) {             // asynchronous: consumed by the next arg-max
        if (wave < NQ && myp < n) {
            const int u = t[myp], v = t[myp + 1];
            gq = guide[(size_t)u * n + v];
            const int q = PenRowMajor<S>::value ? u * n + v : s.idx(u, v);
            pq = s.pen(u, v);
            if (q == qk || (PenRowMajor<S>::value && q == qk2)) pq = pk;
        }
    };
    reload(-1, -1, 0);
    if (tid == 0) tc->stop = 0;
    int moves = 0;
    long long steps = 0;
    bool any_moved = false;
    while (moves < A.perturbation_moves) {
        // ---- arg-max utility over the tour edges, first maximum wins (algorithms.py:153-159) ----
        if (wave < NQ) {
            double bu = 0.0; int bp = kNoKey;
            if (myp < n) { bu = gq / (1.0 + (double)pq); bp = myp; }
            wave_argmax_first(bu, bp);
            if (lane == 0) { tc->arg_u[wave] = bu; tc->arg_p[wave] = bp; }
            if (myp == bp) tc->arg_c[wave] = pq;
        }
        if (tid == 0 && (steps & 63) == 63) {
            const long long el = wall_clock64() - t_start;
            if (el > (long long)(A.watchdog_s * 1e8)) tc->stop = 1;
        }
        // the one full fence of a step: the counter stored in the previous step (global memory for the compact store) is
        // complete before any wavefront's scans of this step load it; the utilities' reloads were consumed above anyway
        __syncthreads();
        if (tc->stop) { status = GNNGLS_STATUS_WATCHDOG_DEV; break; }
        double bu = tc->arg_u[0]; int bp = tc->arg_p[0];
        for (int q = 1; q < NQ; ++q) {
            const double u = tc->arg_u[q];
            if (u > bu) { bu = u; bp = tc->arg_p[q]; }
        }
        STAMP_END(0);
        const int eu = t[bp], ev = t[bp + 1];
        // algorithms.py:161.  The store is not waited for: every wavefront knows the new count (old + 1, published with the
        // arg-max) and its scans of this step substitute it for whatever a load of that counter returns
        const int p_inc = tc->arg_c[bp >> 6] + 1;
        const int q_inc = PenRowMajor<S>::value ? eu * n + ev : s.idx(eu, ev), q_inc2 = ev * n + eu;   // (row-major: both cells)
        if (myp == bp && wave < NQ) { (void)s.pen_set(eu, ev, pq); pq += 1; }
        bool moved_this_step = false;
        int s_begin = 0;                                     // first scan (2 endpoint + operator) not consumed yet
        int i1 = bp + 1;                                     // index of endpoint 1 (algorithms.py:169), see header
        for (int round = 0;; ++round) {
            const int s_end = (s_begin / GLS_TEAM_SCANS + 1) * GLS_TEAM_SCANS;      // scans evaluated in this round: [s_begin, s_end)
            double *res_d = tc->res_d[round & 1]; int *res_k = tc->res_k[round & 1];
            if (moved_this_step && s_begin <= 2) {           // endpoint 1 not started: cur_tour.index(ev) on the current tour
                for (int p0 = 0; p0 <= n; p0 += kWave) {
                    const int p = p0 + lane;
                    const unsigned long long m = __ballot(p <= n && t[p] == ev);
                    if (m) { i1 = p0 + __ffsll((long long)m) - 1; break; }
                }
            }
#ifdef GLS_STAMPS
            const long long tu0 = clock64();
#endif
            for (int unit = wave; unit < units; unit += nwaves) {
                const int sc = unit / P, pass = unit - sc * P;
                const int node = sc >= 2 ? ev : eu;
                double bd = 0.0; int bk = kNoKey;
                if (sc >= s_begin && sc < s_end && node != 0) {      // algorithms.py:168
                    const int i = sc >= 2 ? i1 : bp;
                    if constexpr (PenRowMajor<S>::value) {
                        const int j = 1 + pass * kWave + lane;
                        if ((sc & 1) == 0) scan_two_opt_o2a_guided_rm<S, FI>(s, k, t, n, i, j, q_inc, q_inc2, p_inc, eu, ev, bd, bk);
                        else               scan_relocate_o2a_guided_rm<S, FI>(s, k, t, n, i, j, q_inc, q_inc2, p_inc, eu, ev, bd, bk);
                    } else {
                        if ((sc & 1) == 0) scan_two_opt_o2a_guided<S, FI, TT, true>(s, k, t, n, i, lane, bd, bk, 1 + pass * kWave, n, q_inc, p_inc);
                        else               scan_relocate_o2a_guided<S, FI, TT, true>(s, k, t, n, i, lane, bd, bk, 1 + pass * kWave, n, q_inc, p_inc);
                    }
                    if (__ballot(bk != kNoKey)) wave_reduce_best<FI>(bd, bk);
                }
                if (lane == 0) { res_d[unit] = bd; res_k[unit] = bk; }
            }
            STAMP_END(1);
#ifdef GLS_STAMPS
            st.acc[12] += clock64() - tu0;       // this wavefront's unit(s) of the round
            st.acc[13] += 1;
#endif
            lds_barrier();
            // consume in the reference's order: endpoint, operator; inside a scan the passes ascend in j.  Lane u of every
            // wavefront reads slot u (one LDS round trip for all of them), the first scan with a candidate is one ballot
            // away and its <= 4 passes are compared through v_readlane
            int found = -1, fk = kNoKey;
            {
                const int ku = lane < units ? res_k[lane] : kNoKey;
                const double du = lane < units ? res_d[lane] : 0.0;
                const unsigned long long m = __ballot(ku != kNoKey);     // skipped units carry kNoKey
                const int last = m ? (__ffsll((long long)m) - 1) / P : s_end - 1;   // last scan consumed in this round
                if (tid == 0)
                    for (int sc = s_begin; sc <= last; ++sc)
                        if ((sc >= 2 ? ev : eu) != 0) evals += (sc & 1) == 0 ? (n - 3) : (n - 2);
                if (m) {
                    found = last;
                    double bd = 0.0;
                    const long long dbits = __double_as_longlong(du);
                    for (int pass = 0; pass < P; ++pass) {
                        const int li = found * P + pass;
                        if (!((m >> li) & 1ull)) continue;
                        const int ok = __builtin_amdgcn_readlane(ku, li);
                        const int lo = __builtin_amdgcn_readlane((int)dbits, li), hi = __builtin_amdgcn_readlane((int)(dbits >> 32), li);
                        const double od = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
                        if (fk == kNoKey || better<FI>(od, ok, bd, fk)) { bd = od; fk = ok; }
                    }
                }
            }
            STAMP_END(2);
            if (found < 0) {
                if (s_end >= 4) break;
                s_begin = s_end;                             // nothing in this group of scans: on to the next one
                continue;
            }
            apply_move(s, t, t2, Ef, Eb, n, found & 1, found >= 2 ? i1 : bp, fk, tid, nthr, eager_cost);   // algorithms.py:175-177
            { TT *x = t; t = t2; t2 = x; }
            lds_barrier();
            any_moved = true; moved_this_step = true;
            // the cached utilities are only read by the next arg-max: reloaded after EVERY move, so that the guide-matrix
            // loads (HBM / MALL latency) of all but a step's last move fly under its remaining rounds.  The positions belong to
            // wavefronts 0 .. NQ-1, whose units (scan 0: two_opt_o2a of the first endpoint) only run in a step's first round:
            // their vector-memory queue (loads return in order) is idle until the next step
            reload(q_inc, q_inc2, p_inc);
            moves += 1;                                      // algorithms.py:185
            if (eager_cost) {
                cur_cost = tour_cost_from_edges(Ef, n);      // algorithms.py:176 (every thread: the value stays uniform)
                if (tid == 0) tr.push(cur_cost);
            } else if (tid == 0) {
                tr.len++;                                    // move counted, cost deferred
            }
            STAMP_END(3);
            s_begin = found + 1;
            if (s_begin >= 4) break;
        }
        steps++;
        STAMP_COUNT(6);
    }
    if (any_moved && !eager_cost) {
        build_edges(s, t, Ef, Eb, n, tid, nthr);
        __syncthreads();
        cur_cost = tour_cost_from_edges(Ef, n);
    }
    STAMP_END(4);
}

// ---------------------------------------------------------------------------------------------
// Edge form of the serial perturbation phase (algorithms.py:150-185), symmetric stores, best improvement
// ---------------------------------------------------------------------------------------------
// A penalty step is a dependent chain on ONE wavefront: arg-max, counter store, then up to four guided one-to-all scans
// (two endpoints x {two_opt_o2a, relocate_o2a}, algorithms.py:167-174) with a move after any of them.  What that chain
// costs was measured instruction by instruction (scripts/isa_probe/latency_probe.hip, profiles/r05_isa/): one wavefront
// issues ANY instruction -- fp64 or integer, vector or scalar -- every ~4.3 cycles at best, an exec-masked `if` (v_cmp,
// s_and_saveexec, s_cbranch_execz, s_or) costs ~45 cycles even when nothing is skipped, a scalar branch 15-35, a
// v_cndmask on VCC 8-17 (on another scalar pair: 4.3), a DPP reduction step 12, an fp64 division 68, an LDS round trip
// ~60 and a counter load from L2 270 and more.  The scan-by-scan form above pays, per 64-lane pass, a tour read, three or
// four packed indices with their loads and four exec-masked branches: ~650 cycles, eight passes per step at n = 100.  Here
//   * lane l keeps, per slot q, tour EDGE p = l + 64 q in registers: its nodes (u, v) = (t[p], t[p+1]) with their packed
//     row offsets, its counter, distance and guide value (TourEdges) -- the arg-max's operands and the tour-edge terms of
//     every scan; the tour itself is only WRITTEN to LDS (for the descent, and as the source of the next move);
//   * a scan is enumerated by the edge k = p the lane owns (two_opt_o2a: j = k + 1; relocate_o2a: target edge k, i.e.
//     j = k for i < j and j = k + 1 for i > j, operators.py:91-96), so its per-lane terms are G[a, v], G[b, u] (2-opt) and
//     G[a, u], G[a, v] (relocate) with a = t[i], b = t[i-1] wave-uniform: TWO guided values per lane and slot instead of
//     three or four, all slots of the scan in flight at once; every other term (the lane's own edge, the scan's own edges)
//     comes from the registers (v_readlane for the uniform ones); G[x,y] = D[x,y] + k P[x,y], product rounded first [exact];
//   * conditions live in scalar register pairs (v_cmp ... e64 / lane masks computed once per phase), selects take them
//     from there, and a scan none of whose lanes has a negative delta -- most scans -- costs one compare per slot and one
//     scalar branch: np.isclose, the strict-< bookkeeping and the wave reduction only run behind that test;
//   * the move is applied from LDS to registers: new (u, v) = told[src(p)], told[src(p + 1)], tnew[p] = u is a store nobody
//     waits for.
// Same operands, same operand order, same keys, same order of consumption as the scan-by-scan form: bit-exact.  Measured and
// dropped: evaluating several pending scans of a step at once, speculatively, in the reference's order of consumption (blocks
// of two to four scans chosen by running acceptance estimates): bit-exact and 0.5-3 % slower than one scan at a time
// (profiles/r05_experiments/).
#ifndef GLS_EDGE_PREFETCH
#define GLS_EDGE_PREFETCH 0           // edge form: issue the NEXT scan's loads under the current scan's wait, speculatively.  Measured
                                      // (profiles/r05_experiments/ab_edge_unrolled_and_prefetch.log, bit-exact): -1 % at TSP100 x 1024,
                                      // -8 % at TSP20 / TSP50 against the same unrolled step without it -- off
#endif
#ifndef GLS_EDGE_PERTURB
#define GLS_EDGE_PERTURB 1           // 0: the scan-by-scan serial form everywhere (A/B builds)
#endif

template <int GP>
struct TourEdges {
    int u[GP], v[GP];                // nodes of tour edge p = lane + 64 q (lanes with p >= n hold edge 0: valid nodes, results masked)
    int ur[GP], vr[GP];              // 4 u (u - 1) / 2, 4 v (v - 1) / 2: byte offset of the node's row in a packed int32 triangle
    int pq[GP];                      // penalty counter of the edge
    double gq[GP];                   // utility numerator G.edges[e][guide] (algorithms.py:155)
    double de[GP];                   // D[u, v]
};

// value of a per-edge register at tour edge p (wave-uniform p; slot p >> 6, lane p & 63: v_readlane takes the lane modulo 64)
template <int GP>
__device__ __forceinline__ int edge_bcast(const int (&x)[GP], int p) {
    int r[GP];
#pragma unroll
    for (int q = 0; q < GP; ++q) r[q] = __builtin_amdgcn_readlane(x[q], p);      // all slots, then scalar selects: no branch
    int v = r[0];
#pragma unroll
    for (int q = 1; q < GP; ++q) v = p >= q * kWave ? r[q] : v;
    return v;
}
template <int GP>
__device__ __forceinline__ double edge_bcast_f64(const double (&x)[GP], int p) {
    int lo[GP], hi[GP];
#pragma unroll
    for (int q = 0; q < GP; ++q) { const long long b = __double_as_longlong(x[q]); lo[q] = (int)b; hi[q] = (int)(b >> 32); }
    const int l = edge_bcast<GP>(lo, p), h = edge_bcast<GP>(hi, p);
    return __longlong_as_double(((long long)h << 32) | (unsigned)l);
}

// byte offset of the pair {x, y} in a packed int32 triangle, x wave-uniform (xr = 4 x(x-1)/2 on the scalar unit), y per lane
// with its row offset yr: 4 vector instructions; the packed fp64 triangle is at twice that offset
__device__ __forceinline__ int pair_offset(int x, int xr, int y, int yr) {
    const lanemask_t gt = __builtin_amdgcn_ballot_w64(y > x);
    return sel_b32(gt, yr + 4 * x, xr + 4 * y);
}
// the same for two wave-uniform nodes: scalar max / min, no branch
__device__ __forceinline__ int uniform_pair_offset(int x, int y) {
    const int hi = x > y ? x : y, lo = x > y ? y : x;
    return 2 * hi * (hi - 1) + 4 * lo;
}
// counter and distance of a node pair, in flight
struct PairLoad { int p; double d; };
template <class S>
__device__ __forceinline__ PairLoad pair_issue(const S &s, int off4) {
    PairLoad g;
    g.p = s.pen_at_byte(off4);
    g.d = s.dist_at_byte(2 * off4);
    return g;
}
__device__ __forceinline__ double guided(double k, const PairLoad &g) { return g.d + k * (double)g.p; }   // [exact] algorithms.py:164
__device__ __forceinline__ void pin(PairLoad &g) { asm volatile("" : "+v"(g.p), "+v"(g.d)); }

// (counter, distance, guide) of every tour edge, and the row offsets of its nodes, from E.u / E.v
template <class S, int GP>
__device__ __forceinline__ void edges_fetch(const S &s, TourEdges<GP> &E, const double *guide, int n) {
    // counters and distances first, the guide values (the farthest memory: 80 KB per TSP100 instance, read once per edge) last: loads
    // return in order, and the scan that follows a move needs the former, only the next arg-max the latter (round 6: +2.5 % outer
    // iterations at TSP100 x 1024, +1 % at TSP200 x 256, profiles/r06_experiments/ab_duo4.log, variants _duo0 / _gl0)
#pragma unroll
    for (int q = 0; q < GP; ++q) {
        const int u = E.u[q], v = E.v[q];
        E.ur[q] = 2 * __mul24(u, u - 1); E.vr[q] = 2 * __mul24(v, v - 1);
        const int off = sel_b32(__builtin_amdgcn_ballot_w64(v > u), E.vr[q] + 4 * u, E.ur[q] + 4 * v);
        E.pq[q] = s.pen_at_byte(off);
        E.de[q] = s.dist_at_byte(2 * off);
    }
#pragma unroll
    for (int q = 0; q < GP; ++q) E.gq[q] = guide[(unsigned)(E.u[q] * n + E.v[q])];
}
template <class S, int GP, class TT>
__device__ __forceinline__ void edges_load(const S &s, TourEdges<GP> &E, const TT *t, const double *guide, int n, int lane) {
#pragma unroll
    for (int q = 0; q < GP; ++q) {
        const int p = lane + q * kWave, pc = p < n ? p : 0;
        E.u[q] = t[pc]; E.v[q] = t[pc + 1];
    }
    edges_fetch(s, E, guide, n);
}
// the move (op, i, j) applied: new edges straight from the old tour (operators.py:6-11, 76-80), new tour written behind.
// (lanes past the tour rewrite position 0 with the depot; position n always holds the depot in both tour arrays)
// Both moves are "positions lo .. hi take the node of position sg p + add, one special position takes a given one":
//   two_opt (i < j, operators.py:6-11):  lo = i, hi = j-1: i + j - 1 - p
//   relocate i < j (operators.py:76-80): lo = i, hi = j:   p + 1, position j takes i;   i > j: lo = j, hi = i: p - 1, position j takes i
// -- the parameters on the scalar unit once per move, the per-position part without a branch (move_src: three per call)
struct MoveMap { int lo, span, sg, add, sp, sps; };
__device__ __forceinline__ MoveMap move_map(int op, int i, int j) {
    MoveMap m;
    if (op == 0) {
        const int a = i < j ? i : j, b = i < j ? j : i;
        m.lo = a; m.span = b - 1 - a; m.sg = -1; m.add = a + b - 1; m.sp = -1; m.sps = 0;
    } else if (i < j) {
        m.lo = i; m.span = j - i; m.sg = 1; m.add = 1; m.sp = j; m.sps = i;
    } else {
        m.lo = j; m.span = i - j; m.sg = 1; m.add = -1; m.sp = j; m.sps = i;
    }
    return m;
}
__device__ __forceinline__ int move_src_flat(const MoveMap &m, int p) {
    const lanemask_t in = __builtin_amdgcn_ballot_w64((unsigned)(p - m.lo) <= (unsigned)m.span);
    const lanemask_t sp = __builtin_amdgcn_ballot_w64(p == m.sp);
    return sel_b32(sp, m.sps, sel_b32(in, m.sg * p + m.add, p));
}
template <class S, int GP, class TT>
__device__ __forceinline__ void edges_move(const S &s, TourEdges<GP> &E, const TT *told, TT *tnew, const double *guide,
                                           int n, int op, int i, int j, int lane) {
    const MoveMap mm = move_map(op, i, j);
#pragma unroll
    for (int q = 0; q < GP; ++q) {
        const int p = lane + q * kWave, pc = p < n ? p : 0;
        const int u = told[move_src_flat(mm, pc)];
        E.u[q] = u;
        tnew[pc] = (TT)u;
    }
    // v = the node of the NEXT position = u of the next lane (DPP wave_shl:1; lane 63 takes lane 0 of the next slot; behind the
    // tour every lane holds the depot, which is also what closes the tour): no second source position, no second LDS read
#pragma unroll
    for (int q = 0; q < GP; ++q) {
        int v = __builtin_amdgcn_update_dpp(0, E.u[q], 0x130, 0xf, 0xf, false);
        if (q + 1 < GP) v = sel_b32(1ull << 63, __builtin_amdgcn_readlane(E.u[q + 1 < GP ? q + 1 : q], 0), v);
        E.v[q] = v;
    }
    edges_fetch(s, E, guide, n);
}

// key image of a utility for "first maximum wins": larger value -> smaller unsigned 64-bit key (-0.0 and +0.0 share one)
__device__ __forceinline__ void argmax_key(double v, unsigned &hi, unsigned &lo) {
    v = v + 0.0;                                             // -0.0 -> +0.0, every other value unchanged
    const long long b = __double_as_longlong(v);
    const int h = (int)(b >> 32), t = ~(h >> 31);            // t = all ones for v >= 0
    hi = (unsigned)h ^ ((unsigned)t >> 1);
    lo = (unsigned)b ^ (unsigned)t;
}
// position of the first maximum of the tour edges' utilities (algorithms.py:153-159): util[q] = the lane's utility of edge
// lane + 64 q, nm0[q] = lanes whose edge exists.  Straight-line: per-slot selects on scalar masks, one DPP reduction over the
// high words of the key image, the low words and positions only on a tie.  The maximum itself is not needed.
template <int GP>
__device__ __forceinline__ int argmax_first_pos(const double (&util)[GP], const lanemask_t (&nm0)[GP], int lane) {
    double bu = util[0];
    int pos = lane;
#pragma unroll
    for (int q = 1; q < GP; ++q) {                           // (a lane's edges exist in ascending slots: strict > keeps the first)
        const lanemask_t up = nm0[q] & __builtin_amdgcn_ballot_w64(util[q] > bu);
        bu = sel_f64(up, util[q], bu);
        pos = sel_b32(up, lane + q * kWave, pos);
    }
    unsigned hi, lo;
    argmax_key(bu, hi, lo);
    hi = (unsigned)sel_b32(nm0[0], (int)hi, -1);             // (n < 64: lanes without an edge lose)
    const unsigned mhi = wave_umin(hi);
    const lanemask_t tie = __builtin_amdgcn_ballot_w64(hi == mhi);
    if ((tie & (tie - 1)) == 0ull)                           // one lane holds the smallest high word: the usual case
        return __builtin_amdgcn_readlane(pos, __ffsll((long long)tie) - 1);
    ISA_MARK("rare_argmax_tie");
    const unsigned mlo = wave_umin(hi == mhi ? lo : 0xffffffffu);
    const int r = (int)wave_umin((hi == mhi && lo == mlo) ? (unsigned)pos : 0x7fffffffu);
    ISA_MARK("argmax_reduce");
    return r;
}

// key of the lexicographic minimum of (delta, key) over the lanes with a candidate (key != kNoKey; their delta < 0): the wave
// reduction of wave_min_value_key with a select-free order-preserving image (negative deltas only: the image of a negative
// double is its complemented bit pattern) -- only the key is needed, the move's delta is not used by the perturbation phase
__device__ __forceinline__ int wave_argmin_key(double d, int k) {
    const long long b = __double_as_longlong(d);
    const lanemask_t cand = __builtin_amdgcn_ballot_w64(k != kNoKey);
    const unsigned hi = (unsigned)sel_b32(cand, ~(int)(b >> 32), -1), lo = ~(unsigned)b;
    const unsigned mhi = wave_umin(hi);
    const lanemask_t tie = cand & __builtin_amdgcn_ballot_w64(hi == mhi);
    if ((tie & (tie - 1)) == 0ull) return __builtin_amdgcn_readlane(k, __ffsll((long long)tie) - 1);
    const unsigned mlo = wave_umin((unsigned)sel_b32(tie, (int)lo, -1));
    const lanemask_t tie2 = tie & __builtin_amdgcn_ballot_w64(lo == mlo);
    return (int)wave_umin((unsigned)sel_b32(tie2, k, 0x7fffffff));
}

// One guided one-to-all scan at tour index i on the tour held by E: RELOC = false two_opt_o2a, true relocate_o2a -- in two
// halves, so that the loads of the NEXT scan of a step can be issued under the wait of the current one (scan_issue: the
// wave-uniform nodes by v_readlane, packed offsets, counter / distance loads; scan_finish: everything else).
template <bool RELOC, int GP>
struct ScanLoads { PairLoad x0[GP], x1[GP], xac; };

template <bool RELOC, class S, int GP>
__device__ __forceinline__ void scan_issue(const S &s, const TourEdges<GP> &E, const int i, ScanLoads<RELOC, GP> &L) {
    ISA_MARK("scan_issue");
    const int na = edge_bcast<GP>(E.u, i), nb = edge_bcast<GP>(E.u, i - 1);       // a = t[i], b = t[i-1]
    const int nar = 2 * na * (na - 1), nbr = 2 * nb * (nb - 1);
    if (RELOC) {                                             // G[t[i-1], t[i+1]]: a wave-uniform pair
        const int nc = edge_bcast<GP>(E.v, i);
        L.xac = pair_issue(s, uniform_pair_offset(nb, nc));
    }
#pragma unroll
    for (int q = 0; q < GP; ++q) {
        L.x0[q] = pair_issue(s, pair_offset(na, nar, E.v[q], E.vr[q]));                         // G[t[i], t[k+1]]
        L.x1[q] = RELOC ? pair_issue(s, pair_offset(na, nar, E.u[q], E.ur[q]))                  // G[t[i], t[k]]
                        : pair_issue(s, pair_offset(nb, nbr, E.u[q], E.ur[q]));                 // G[t[i-1], t[k]]
    }
}

// ok[q] = the lanes of slot q with a valid move, delta[q] their deltas; returns the lanes (any slot) with a negative one.
template <bool RELOC, int GP>
__device__ __forceinline__ lanemask_t scan_finish(const double k, const TourEdges<GP> &E, const int lane, const int i,
                                                  const lanemask_t (&nm)[GP], ScanLoads<RELOC, GP> &L, double (&delta)[GP], lanemask_t (&ok)[GP]) {
    ISA_MARK("scan_under_latency");
    // under the loads' latency: everything that does not need them -- the guided lengths of the tour edges (registers), the
    // scan's own edges (v_readlane), the subtrahends and validity masks of the lanes
    double ge[GP];                                           // guided length of the lane's own edges
#pragma unroll
    for (int q = 0; q < GP; ++q) ge[q] = E.de[q] + k * (double)E.pq[q];
    const double gab = edge_bcast_f64<GP>(ge, i - 1);        // G[t[i-1], t[i]]
    double gbc = 0.0;
    if (RELOC) gbc = edge_bcast_f64<GP>(ge, i);              // G[t[i], t[i+1]]
    double s1[GP], s2[GP];                                   // two_opt_o2a: first and second subtrahend of the lane
#pragma unroll
    for (int q = 0; q < GP; ++q) {
        const int kk = lane + q * kWave;
        if (!RELOC) {
            // j = k + 1 > i: - G[a,b] of the scan's edge, then - G[c,d] of the lane's; j < i (operators.py:17-18 swap): the other way
            const lanemask_t lt = __builtin_amdgcn_ballot_w64(kk >= i);
            s1[q] = sel_f64(lt, gab, ge[q]); s2[q] = sel_f64(lt, ge[q], gab);
            // j = 1 .. n-1, |i - j| >= 2 (operators.py:59-62): k <= n-2, k not in {i-2, i-1, i}
            ok[q] = nm[q] & __builtin_amdgcn_ballot_w64((unsigned)(kk - i + 2) > 2u);
        } else {
            // j != i (operators.py:114-115): k <= n-1, k not in {i-1, i}
            ok[q] = nm[q] & __builtin_amdgcn_ballot_w64((unsigned)(kk - i + 1) > 1u);
        }
    }
    // every load of the scan was issued before the first is consumed (cf. scan_two_opt_o2a_guided_rm)
    ISA_MARK("scan_wait_and_arith");
#pragma unroll
    for (int q = 0; q < GP; ++q) { pin(L.x0[q]); pin(L.x1[q]); asm volatile("" : "+v"(ge[q])); if (!RELOC) asm volatile("" : "+v"(s1[q]), "+v"(s2[q])); }
    double base = 0.0;
    if (RELOC) {
        pin(L.xac);
        base = -gab;                                         // operators.py:97-99, left to right
        base = base - gbc;
        base = base + guided(k, L.xac);
    }
    lanemask_t neg = 0ull;
#pragma unroll
    for (int q = 0; q < GP; ++q) {
        const double gav = guided(k, L.x0[q]), gxu = guided(k, L.x1[q]);
        double d;
        if (!RELOC) {
            // two_opt_o2a (operators.py:53-73), j = k + 1: c = t[j] = v, d = t[j-1] = u: ((G[a,c] + G[b,d]) - G[a,b]) - G[c,d]
            d = gav + gxu;
            d = d - s1[q];
            d = d - s2[q];
        } else {
            // relocate_o2a (operators.py:106-126) by target edge k = (d, e) = (u, v): ((base - G[d,e]) + G[d,b]) + G[b,e], b = t[i];
            // j = k for i < j, j = k + 1 for i > j (operators.py:91-96); j != i: k not in {i-1, i}
            d = base - ge[q];
            d = d + gxu;
            d = d + gav;
        }
        delta[q] = d;
        neg |= ok[q] & __builtin_amdgcn_ballot_w64(d < 0.0);
    }
    return neg;
}

template <class S, int GP, bool TR, bool CNT, class TT, class TRC>
__device__ __forceinline__ void serial_perturbation_edges(const S &s, const double k, TT *&t, TT *&t2, double *Ef, double *Eb,
                                                          const int n, const double *guide, const GlsArgs &A,
                                                          const long long t_start, double &cur_cost, TRC &tr,
                                                          long long &evals, int &status, long long &steps_total, Stamps &st) {
    static_assert(S::kSymmetric && sizeof(typename S::pen_t) == 4, "edge form: symmetric stores with 32-bit counters");
    constexpr bool eager_cost = TR;
    const int lane = threadIdx.x & (kWave - 1);
    lanemask_t nm0[GP], nm2[GP];                             // lanes whose edge k = lane + 64 q is <= n-1 / <= n-2
#pragma unroll
    for (int q = 0; q < GP; ++q) {
        nm0[q] = __builtin_amdgcn_ballot_w64(lane + q * kWave < n);
        nm2[q] = __builtin_amdgcn_ballot_w64(lane + q * kWave < n - 1);
    }
    if (lane == 0) { t2[0] = (TT)t[0]; t2[n] = (TT)t[n]; }   // positions 0 and n hold the depot in both tour arrays
    TourEdges<GP> E;
    edges_load(s, E, t, guide, n, lane);
    bool any_moved = false;
    int moves = 0, scans_to = 0, scans_re = 0;               // one-to-all scans executed (evaluation count, booked at the end)
    long long steps = 0;
    int max_moves = A.perturbation_moves;
    asm volatile("" : "+s"(max_moves));                      // a register, not a kernel-argument load (and its wait) per step
    while (moves < max_moves) {
        // ---- arg-max utility over the tour edges, first maximum wins (algorithms.py:153-159); only its position is used ----
        STAMP_END(12);      // (diagnostic builds: end of the previous step / loop latch)
        ISA_MARK("argmax_divisions");
        double util[GP];
#pragma unroll
        for (int q = 0; q < GP; ++q) util[q] = E.gq[q] / (1.0 + (double)E.pq[q]);     // lanes past the tour hold edge 0: masked
#ifdef GLS_STAMPS
#pragma unroll
        for (int q = 0; q < GP; ++q) asm volatile("" : "+v"(util[q]));
        STAMP_END(13);      // (the divisions, behind the loads of the last move)
#endif
        ISA_MARK("argmax_reduce");
        const int bp = argmax_first_pos<GP>(util, nm0, lane);
        STAMP_END(0);
        ISA_MARK("penalise");
        const int eu = edge_bcast<GP>(E.u, bp), ev = edge_bcast<GP>(E.v, bp);
        // algorithms.py:161: the lane that holds edge bp stores count + 1 itself (no load -> add -> store round trip)
        {
            int cnt = 0;
            lanemask_t own = 0ull;
#pragma unroll
            for (int q = 0; q < GP; ++q) {
                const lanemask_t o = __builtin_amdgcn_ballot_w64(lane + q * kWave == bp);
                E.pq[q] += sel_b32(o, 1, 0);
                cnt = sel_b32(o, E.pq[q], cnt);
                own |= o;
            }
            s.pen_store_byte_if(own, uniform_pair_offset(eu, ev), cnt);
        }
        // The four scans of the step (scan = 2 endpoint + operator; algorithms.py:167-171), unrolled: each with its operator as a
        // compile-time constant (against one loop body that selects the operator at run time: +1 % at TSP100 x 1024, +6 .. 9 % at
        // TSP20 / TSP50 / TSP150).  GLS_EDGE_PREFETCH (off, see there): while a scan waits for its counters the loads of the scan
        // BEHIND it are issued on the current tour and dropped if the scan moves; the result never depends on them.
        constexpr bool kPrefetch = GLS_EDGE_PREFETCH && GP <= 2;
        const int sc_lo = eu == 0 ? 2 : 0, sc_hi = ev == 0 ? 2 : 4;
        bool moved_this_step = false;
        ScanLoads<false, GP> LT;
        ScanLoads<true, GP> LR;
        bool ahead = false;                                  // the loads of the scan about to run are already in flight
        // index of endpoint 1: bp + 1 (algorithms.py:169: the edge was read at positions bp, bp + 1); cur_tour.index(ev), searched
        // only after a move
        auto index_of_ev = [&]() {
            int i1 = bp + 1;
            if (moved_this_step) {
#pragma unroll
                for (int q = GP - 1; q >= 0; --q) {
                    const lanemask_t m = nm0[q] & __builtin_amdgcn_ballot_w64(E.u[q] == ev);
                    if (m) i1 = q * kWave + __ffsll((long long)m) - 1;
                }
            }
            return i1;
        };
        int i = bp;
        auto one_scan = [&](auto reloc_tag, const int sc) {
            constexpr bool RELOC = decltype(reloc_tag)::value;
            if (sc < sc_lo || sc >= sc_hi) return;
            ISA_MARK("scan_loop_head");
            if (sc == 2) i = index_of_ev();
            auto &L = [&]() -> auto & { if constexpr (RELOC) return LR; else return LT; }();
            if (!ahead) scan_issue<RELOC>(s, E, i, L);
            ahead = false;
            if constexpr (kPrefetch) {
                if (sc + 1 < sc_hi) {                        // the next scan: the other operator; endpoint 1 after scan 1
                    const int i_next = sc == 1 ? index_of_ev() : i;
                    if constexpr (RELOC) scan_issue<false>(s, E, i_next, LT); else scan_issue<true>(s, E, i_next, LR);
                    ahead = true;
                }
            }
            double delta[GP];
            lanemask_t ok[GP];
            const lanemask_t neg = RELOC ? scan_finish<RELOC>(k, E, lane, i, nm0, L, delta, ok) : scan_finish<RELOC>(k, E, lane, i, nm2, L, delta, ok);
            ISA_MARK("accept_fast");
            if (RELOC) scans_re += 1; else scans_to += 1;
            STAMP_END(1);
            if (neg == 0ull) return;                         // no negative delta: no candidate (most scans)
            ISA_MARK("accept_slow");
            // np.isclose evaluated literally; the keys of a lane ascend with its slots, so a strict < keeps the lane's first
            // minimum (operators.py:65,118)
            double bd = 0.0; int bk = kNoKey;
#pragma unroll
            for (int q = 0; q < GP; ++q) {
                const double d = delta[q];
                const int kk = lane + q * kWave;
                const int key = RELOC ? (kk >= i + 1 ? kk : kk + 1) : kk + 1;
                const lanemask_t take = ok[q] & __builtin_amdgcn_ballot_w64(d < bd) & ~__builtin_amdgcn_ballot_w64(close_to_zero(d));
                bd = sel_f64(take, d, bd);
                bk = sel_b32(take, key, bk);
            }
            if (__builtin_amdgcn_ballot_w64(bk != kNoKey) == 0ull) { STAMP_END(2); return; }
            bk = wave_argmin_key(bd, bk);
            STAMP_END(2);
            ISA_MARK("move");
            edges_move(s, E, t, t2, guide, n, RELOC ? 1 : 0, i, bk, lane);     // algorithms.py:175-177
            { TT *x = t; t = t2; t2 = x; }
            any_moved = true; moved_this_step = true;
            ahead = false;                                   // what was issued ahead was on the old tour
            moves += 1;                                      // algorithms.py:185
            if (eager_cost) {
#pragma unroll
                for (int q = 0; q < GP; ++q) if (lane + q * kWave < n) Ef[lane + q * kWave + 1] = E.de[q];
                wave_sync();
                cur_cost = tour_cost_from_edges(Ef, n);      // algorithms.py:176
                if (lane == 0) tr.push(cur_cost);
            }
            STAMP_END(3);
            ISA_MARK("scan_loop_tail");
        };
        one_scan(std::false_type{}, 0);
        one_scan(std::true_type{}, 1);
        one_scan(std::false_type{}, 2);
        one_scan(std::true_type{}, 3);
        ISA_MARK("step_tail");
        steps++;
        STAMP_COUNT(6);
        if ((steps & 63) == 0) {
            ISA_MARK("rare_watchdog");
            const long long el = wall_clock64() - t_start;
            if (el > (long long)(A.watchdog_s * 1e8)) { status = GNNGLS_STATUS_WATCHDOG_DEV; break; }
        }
        ISA_MARK("step_tail");
    }
    ISA_MARK("phase_end");
    if (lane == 0) {
        evals += (long long)scans_to * (n - 3) + (long long)scans_re * (n - 2);
        if (!eager_cost) tr.len += moves;                    // moves counted, costs deferred
    }
    if constexpr (CNT) steps_total += steps;                 // (measurement builds: penalty steps of the run, GlsArgs::evals_exec)
    if (any_moved && !eager_cost) {
        wave_sync();
        build_edges(s, t, Ef, Eb, n, lane, kWave);
        wave_sync();
        cur_cost = tour_cost_from_edges(Ef, n);
    }
}

