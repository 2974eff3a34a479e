// gls_descent_scans.h -- part of gls_kernels.hip (one translation unit; included inside namespace gnngls, in this order:
// gls_common.h, gls_descent_scans.h, gls_perturbation.h).  All-to-all scans of the descent (operators.py:32-50,129-147): generic, row-on-the-lane, lean, half-wave and pruned forms, the neighbour lists and the symmetry check.
#pragma once

// ---------------------------------------------------------------------------------------------
// a2a scans on the plain matrix.  Ef[p] = dist(t[p-1], t[p]), Eb[p] = dist(t[p], t[p-1]), p=1..n.
// ---------------------------------------------------------------------------------------------
// Work items of a wavefront = (row i, pass of 64 lanes over j = 1 + 64*pass + lane).  U independent items are
// evaluated per step, loads first, so their LDS latencies overlap (the descent is latency-bound: one or two
// wavefronts per SIMD, a chain of dependent ds_reads per evaluation).
template <class S, bool FI, class TT, int U>
__device__ __forceinline__ void scan_two_opt_a2a(const S &s, const TT *t, const double *Eb, int n,
                                                 int wave, int nwaves, int lane, double &bd, int &bk) {
    // itertools.combinations(range(1,n),2), |i-j| >= 2  (operators.py:36-39): rows i = 1..n-3, j = i+2..n-1
    if constexpr (U == 1) {      // register-starved variants: plain row loop
        for (int i = 1 + wave; i <= n - 3; i += nwaves) {
            const int a = t[i], b = t[i - 1];
            const double eab = Eb[i];                            // D[a,b]
            for (int j = i + 2 + lane; j <= n - 1; j += kWave) {
                const int c = t[j], d = t[j - 1];
                double delta = s.dist(a, c) + s.dist(b, d);
                delta = delta - eab;
                delta = delta - Eb[j];                           // D[c,d]
                consider<FI>(delta, make_key(i, j), bd, bk);
            }
        }
        return;
    }
    const int P = (n - 1 + kWave - 1) / kWave;
    const int rows = n - 3;
    const int my_rows = rows > wave ? (rows - wave + nwaves - 1) / nwaves : 0;
    int r = 0, pass = 0;
    for (int q0 = 0; q0 < my_rows * P; q0 += U) {
        int ii[U], jj[U]; bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = r < my_rows;
            const int i = 1 + wave + (live ? r : 0) * nwaves;
            const int j = 1 + pass * kWave + lane;
            ok[u] = live && j >= i + 2 && j <= n - 1;
            ii[u] = i; jj[u] = ok[u] ? j : i + 2;
            if (++pass == P) { pass = 0; ++r; }
        }
        int a[U], b[U], c[U], d[U]; double eab[U], ecd[U], x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = t[ii[u]]; b[u] = t[ii[u] - 1]; c[u] = t[jj[u]]; d[u] = t[jj[u] - 1];
            eab[u] = Eb[ii[u]]; ecd[u] = Eb[jj[u]];                        // D[a,b], D[c,d]
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { x[u] = s.dist(a[u], c[u]); y[u] = s.dist(b[u], d[u]); }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double delta = x[u] + y[u];
            delta = delta - eab[u];
            delta = delta - ecd[u];
            if (ok[u]) consider<FI>(delta, make_key(ii[u], jj[u]), bd, bk);
        }
    }
}

template <class S, bool FI, class TT, int U>
__device__ __forceinline__ void scan_relocate_a2a(const S &s, const TT *t, const double *Ef, int n,
                                                  int wave, int nwaves, int lane, double &bd, int &bk) {
    // itertools.permutations(range(1,n),2), skip i-j == 1  (operators.py:133-136): rows i = 1..n-1
    if constexpr (U == 1) {
        for (int i = 1 + wave; i <= n - 1; i += nwaves) {
            const int a = t[i - 1], b = t[i], c = t[i + 1];
            double base = -Ef[i];                                // -D[a,b]
            base = base - Ef[i + 1];                             // -D[b,c]
            base = base + s.dist(a, c);                          // +D[a,c]
            for (int j = 1 + lane; j <= n - 1; j += kWave) {
                if (j == i || j == i - 1) continue;
                int d, e; double de;
                if (i < j) { d = t[j]; e = t[j + 1]; de = Ef[j + 1]; }
                else       { d = t[j - 1]; e = t[j]; de = Ef[j]; }
                double delta = base - de;                        // -D[d,e]
                delta = delta + s.dist(d, b);
                delta = delta + s.dist(b, e);
                consider<FI>(delta, make_key(i, j), bd, bk);
            }
        }
        return;
    }
    const int P = (n - 1 + kWave - 1) / kWave;
    const int rows = n - 1;
    const int my_rows = rows > wave ? (rows - wave + nwaves - 1) / nwaves : 0;
    int r = 0, pass = 0;
    for (int q0 = 0; q0 < my_rows * P; q0 += U) {
        int ii[U], jj[U]; bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = r < my_rows;
            const int i = 1 + wave + (live ? r : 0) * nwaves;
            const int j = 1 + pass * kWave + lane;
            ok[u] = live && j <= n - 1 && j != i && j != i - 1;
            ii[u] = i; jj[u] = ok[u] ? j : (i == n - 1 ? 1 : n - 1);      // any in-range j != i
            if (++pass == P) { pass = 0; ++r; }
        }
        int b[U], d[U], e[U]; double base[U], de[U], x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = ii[u], j = jj[u];
            const int a = t[i - 1], c = t[i + 1];
            b[u] = t[i];
            double bs = -Ef[i];                              // -D[a,b]
            bs = bs - Ef[i + 1];                             // -D[b,c]
            base[u] = bs + s.dist(a, c);                     // +D[a,c]
            if (i < j) { d[u] = t[j]; e[u] = t[j + 1]; de[u] = Ef[j + 1]; }
            else       { d[u] = t[j - 1]; e[u] = t[j]; de[u] = Ef[j]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { x[u] = s.dist(d[u], b[u]); y[u] = s.dist(b[u], e[u]); }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double delta = base[u] - de[u];                  // -D[d,e]
            delta = delta + x[u];
            delta = delta + y[u];
            if (ok[u]) consider<FI>(delta, make_key(ii[u], jj[u]), bd, bk);
        }
    }
}

// a2a scans, "row on the lane" mapping: lane l of row-wave rw owns tour row i = 1 + 64*rw + l for the whole scan (its
// row constants stay in registers), and the wavefront walks j uniformly (j-groups of waves take j strided).  The
// tour bytes and edge lengths of position j are wave-uniform (one broadcast LDS read, moved to SGPRs), so an evaluation is
// ONE dependent LDS round trip (the two random distance reads) instead of tour bytes -> index -> distances, with
// no per-row prologue.  Same deltas, same keys, same arg-min as the row-per-wavefront scans above.
template <class S, bool FI, class TT>
__device__ __forceinline__ void scan_two_opt_a2a_rowlane(const S &s, const TT *t, const double *Eb, int n,
                                                         int wave, int nwaves, int lane, double &bd, int &bk) {
    const int RW = (n - 1 + kWave - 1) / kWave;              // row blocks of 64 rows
    const int JG = nwaves / RW > 0 ? nwaves / RW : 1;        // wave groups sharing a row block, striding over j
    const int RWE = nwaves / JG;
    const int rw = __builtin_amdgcn_readfirstlane(wave % RWE), jg = __builtin_amdgcn_readfirstlane(wave / RWE);
    if (jg >= JG) return;
    for (int rb = rw; rb < RW; rb += RWE) {
        const int i = 1 + rb * kWave + lane;
        const bool row_ok = i <= n - 3;                      // combinations(range(1,n),2), |i-j| >= 2 (operators.py:36-39)
        const int ic = row_ok ? i : 1;
        const int a = t[ic], b = t[ic - 1];
        const int a2 = (a * (a - 1)) >> 1, b2 = (b * (b - 1)) >> 1;     // triangular row offsets, once per scan
        const double eab = Eb[ic];                           // D[a,b]
        const int jlo = 3 + rb * kWave;                      // smallest j any lane of this block can use (i+2, i >= 1+64rb)
        for (int j = jlo + jg; j <= n - 1; j += JG) {
            const int c = __builtin_amdgcn_readfirstlane((int)t[j]), d = __builtin_amdgcn_readfirstlane((int)t[j - 1]);
            const int c2 = (c * (c - 1)) >> 1, d2 = (d * (d - 1)) >> 1;   // wave-uniform: scalar ALU
            const double ecd = Eb[j];                        // D[c,d] (uniform address: broadcast)
            if (row_ok && j >= i + 2) {
                double delta = s.dist_at(s.idx2(a, a2, c, c2)) + s.dist_at(s.idx2(b, b2, d, d2));
                delta = delta - eab;
                delta = delta - ecd;
                consider<FI>(delta, make_key(i, j), bd, bk);
            }
        }
    }
}

template <class S, bool FI, class TT>
__device__ __forceinline__ void scan_relocate_a2a_rowlane(const S &s, const TT *t, const double *Ef, int n,
                                                          int wave, int nwaves, int lane, double &bd, int &bk) {
    const int RW = (n - 1 + kWave - 1) / kWave;
    const int JG = nwaves / RW > 0 ? nwaves / RW : 1;
    const int RWE = nwaves / JG;
    const int rw = __builtin_amdgcn_readfirstlane(wave % RWE), jg = __builtin_amdgcn_readfirstlane(wave / RWE);
    if (jg >= JG) return;
    for (int rb = rw; rb < RW; rb += RWE) {
        const int i = 1 + rb * kWave + lane;
        const bool row_ok = i <= n - 1;                      // permutations(range(1,n),2), skip i-j == 1 (operators.py:133-136)
        const int ic = row_ok ? i : 1;
        const int a = t[ic - 1], b = t[ic], cc = t[ic + 1];
        double base = -Ef[ic];                               // -D[a,b]
        base = base - Ef[ic + 1];                            // -D[b,c]
        base = base + s.dist(a, cc);                         // +D[a,c]
        const int b2 = (b * (b - 1)) >> 1;
        for (int j = 1 + jg; j <= n - 1; j += JG) {
            const int tjm = __builtin_amdgcn_readfirstlane((int)t[j - 1]);
            const int tj = __builtin_amdgcn_readfirstlane((int)t[j]);
            const int tjp = __builtin_amdgcn_readfirstlane((int)t[j + 1]);
            const int tjm2 = (tjm * (tjm - 1)) >> 1, tj2 = (tj * (tj - 1)) >> 1, tjp2 = (tjp * (tjp - 1)) >> 1;
            const double ej = Ef[j], ejp = Ef[j + 1];
            if (row_ok && j != i && j != i - 1) {
                const int d = i < j ? tj : tjm, e = i < j ? tjp : tj;
                const int d2 = i < j ? tj2 : tjm2, e2 = i < j ? tjp2 : tj2;
                double delta = base - (i < j ? ejp : ej);    // -D[d,e]
                delta = delta + s.dist_at(s.idx2(b, b2, d, d2));   // +D[d,b] (symmetric stores) / D[b,d] index order n/a here
                delta = delta + s.dist_at(s.idx2(b, b2, e, e2));   // +D[b,e]
                consider<FI>(delta, make_key(i, j), bd, bk);
            }
        }
    }
}

// Workgroup arg-min of (delta, key) for the best-improvement descent, by LDS atomics instead of DPP chains: the lanes
// that may hold the minimum issue ONE ds_min_u64 on the order-preserving image of their delta; after a barrier the lanes
// that hold the minimum issue one ds_min_u32 on their key.  Same result as the lexicographic min of block_reduce_best (value,
// then key); of its three dependent 6-step DPP reductions per wavefront, the per-wave exchange and the compare chain over the
// waves' results (~100 dependent instructions, ~2,600 cycles per scan at four waves per SIMD: profiles/r02_stamps_per_wave.log)
// one DPP reduction, two LDS atomics and one extra barrier remain.  Measured (same box, outer iterations per instance):
// TSP100 x 1024 +1.4 .. +1.9 %, TSP50 x 128 +4.8 %, TSP200 x 256 +1.6 %, LDS-penalty store x 512 +5.3 % (profiles/r02_ab_lds_atomic_argmin.log).  Three slots rotate: slot `phase` is in use, the next one is reset by
// thread 0 before the first barrier (its last readers passed the previous reduction's barrier long ago).
__device__ __forceinline__ void block_reduce_lds_init(Ctl *ctl, int tid) {
    if (tid < 3) {
        reinterpret_cast<unsigned long long *>(&ctl->red_d[0][0])[tid] = ~0ull;
        reinterpret_cast<unsigned *>(&ctl->red_k[0][0])[tid] = 0x7fffffffu;
    }
}
__device__ __forceinline__ void block_reduce_best_lds(Ctl *ctl, int &phase, int tid, double &d, int &k) {
    typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;
    typedef __attribute__((address_space(3))) unsigned lds_u32_t;
    lds_u64_t *av = (lds_u64_t *)reinterpret_cast<unsigned long long *>(&ctl->red_d[0][0]);
    lds_u32_t *ak = (lds_u32_t *)reinterpret_cast<unsigned *>(&ctl->red_k[0][0]);
    ISA_MARK("argmin_lds_begin");
    const int sl = phase, nx = phase == 2 ? 0 : phase + 1;
    const bool cand = k != kNoKey;
    const unsigned long long sk = sortable(d);
    // 64 lanes on one address serialise in the LDS atomic unit (with every candidate lane going, a noise guide lost 3 %):
    // one DPP min over the high words first, and only the lanes that share the wavefront's smallest high word go
    const unsigned hi = cand ? (unsigned)(sk >> 32) : 0xffffffffu;
    const unsigned mhi = wave_umin(hi);
    if (cand && hi == mhi) __hip_atomic_fetch_min(&av[sl], sk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (tid == 0) { av[nx] = ~0ull; ak[nx] = 0x7fffffffu; }
    __syncthreads();
    phase = nx;
    const unsigned long long m = av[sl];
    if (m == ~0ull) { k = kNoKey; return; }                  // no candidate in the workgroup (uniform)
    if (cand && sk == m) __hip_atomic_fetch_min(&ak[sl], (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    k = (int)ak[sl];
    d = unsortable(m);
    ISA_MARK("argmin_lds_end");
}

// ---- lean "row on the lane" scans (best improvement, symmetric stores, n <= 255) -------------------------------------
// The descent is bound by vector-instruction issue (VALU ~76 % busy at full residency, LDS ~30 %), not by LDS
// bandwidth: what counts is the number of VALU instructions per evaluation.  These versions keep the row-on-the-lane
// mapping (lane l owns tour row i = 1 + 64 rb + l; the wavefront walks the other index uniformly) and strip the inner
// loop to the arithmetic of the reference plus one add/add/max address:
//   * the tour node of a step is wave-uniform: every lane keeps positions l, l + 64, ... of the tour in registers
//     (loaded once per scan) and the step broadcasts it with v_readlane into an SGPR, where its triangular row
//     address is scalar arithmetic; the tour-edge length of the step is one LDS read at a wave-uniform address
//     (a broadcast read: the LDS pipe has the headroom, the vector ALU does not);
//   * the packed-triangle address of a pair needs no compare/select (tri_addr_max below): 3 VALU instead of 5;
//   * relocate is enumerated by target EDGE k = (t[k], t[k+1]) instead of by j: for i < j the reference inserts
//     between t[j], t[j+1] (k = j), for i > j between t[j-1], t[j] (k = j - 1) (operators.py:91-96), so
//     delta(i, k) = ((base_i - Ef[k+1]) + D[t[k], b]) + D[b, t[k+1]] has ONE form, no per-lane selects, and
//     D[b, t[k+1]] of step k is D[t[k+1], b] of step k + 1 (symmetric store: same bits): one random LDS read per step;
//   * within a lane the keys (i, j) ascend, so "first minimum wins" is a plain strict `delta < best` (no key compare);
//     np.isclose and the excluded positions of a row (|i-j| < 2 in 2-opt, three targets in relocate) are only
//     evaluated for a candidate that already beats the lane's best; rows past the end carry delta = +inf and never do.
// Same deltas (same operands in the same order, read from the same addresses), same keys, same arg-min as the scans above.
// Measured (same box, outer iterations per instance in 2 s, TSP100 x 1024 noise / weight guide, TSP200 x 256, TSP50 x 128;
// profiles/r02_ab_lean_scan_v2.log): select-free address 7.15k -> 7.46k / 11.96k -> 12.34k / 5.46k -> 5.61k / 9.96k -> 10.24k;
// + edge lengths from LDS instead of two v_readlane 8.06k / 12.91k / 6.07k / 10.43k; + late validity 8.20k / 13.06k / 6.12k / 10.42k.
// Round 5: the deltas of a group against the lane's best, all at once.  `delta < bd` can only hold for a step of the group if the
// group's minimum is below bd (bd only falls inside a group; v_min_f64 returns the other operand for a NaN, and a NaN delta never
// beats anything): a filter in front of the per-step tests, which run unchanged behind it -- one exec-masked region (v_cmp,
// s_and_saveexec, s_cbranch_execz, s_or: four instructions and ~45 cycles of the wavefront, entered or not) per group instead of
// one per step, for U - 1 v_min_f64.  Same-box A/B (profiles/r05_experiments/ab_descent_*.log, outer iterations per 2 s, TSP100 x
// 1024, model guide): 22.40k -> 23.13k; with six steps per group in the relocate scan (GLS_LEAN_UNROLL_RELOCATE) 23.48k, TSP50 x 128
// 27.80k -> 28.04k (six steps in the 2-opt scan too: 27.47k; eight: 26.2k).  Per build: MF / UR template arguments of the scans.
#ifndef GLS_LEAN_MINFILTER
#define GLS_LEAN_MINFILTER 1
#endif
__device__ __forceinline__ double min_f64_raw(double a, double b) {      // (fmin puts a canonicalising v_max_f64 x, x in front of operands it cannot see through)
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int LO, int HI, int U>
__device__ __forceinline__ double min_tree(const double (&dl)[U]) {
    if constexpr (HI - LO == 1) return dl[LO];
    else return min_f64_raw(min_tree<LO, (LO + HI) / 2>(dl), min_tree<(LO + HI) / 2, HI>(dl));
}
template <bool MF, int U>
__device__ __forceinline__ bool group_may_improve(const double (&dl)[U], double bd) {
    if constexpr (!MF || !GLS_LEAN_MINFILTER || U == 1) return true;
    else return min_tree<0, U>(dl) < bd;
}
template <int SL>
struct LaneTour {      // positions lane, lane + 64, ... (SL slots) of the tour in registers
    int t[SL];
};
template <int SL, class TT>
__device__ __forceinline__ LaneTour<SL> load_lane_tour(const TT *t, int n, int lane) {
    LaneTour<SL> L;
    // (four-slot builds: the clamped positions are loop-invariant, and as such the compiler kept them alive across the whole kernel -- in
    // scratch, reloaded with a wait in front of every one of these reads; recomputed from an opaque copy of the lane they cost two
    // vector instructions each)
    if constexpr (SL >= 4) asm volatile("" : "+v"(lane));
#pragma unroll
    for (int q = 0; q < SL; ++q) {
        const int p = lane + q * kWave <= n ? lane + q * kWave : n;
        L.t[q] = t[p];
    }
    return L;
}
__device__ __forceinline__ int bcast_int(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }
// LDS byte addresses as integers: the lean scans fold the base and the element size into the per-lane / per-step terms
typedef __attribute__((address_space(3))) const double lds_cf64_t;
__device__ __forceinline__ int lds_byte_addr(const double *p) { return (int)(size_t)(lds_cf64_t *)p; }
__device__ __forceinline__ double lds_read_f64(int byte_addr) { return *(lds_cf64_t *)(size_t)(unsigned)byte_addr; }
// Packed-triangle address of the pair {x, y} without compare/select: with rx = base + 8 x(x-1)/2 and x8 = 8 x,
//   max(rx + y8, ry + x8) is the address of D[max(x,y), min(x,y)] whenever x != y and x + y >= 3
// (for x > y: (rx + y8) - (ry + x8) = 4 (x - y)(x + y - 3)).  The pairs {0,1} and {0,2} are the exceptions: callers keep
// node 0 out of it (a lane whose own node is 0 passes rx = kNoRow so the other candidate always wins; a step whose
// uniform node is 0 takes the exact index).
constexpr int kNoRow = -(1 << 30);
// keeps a per-lane address term as ONE register the optimiser cannot look through: left alone it re-associates
// (base + 8 r) + 8 y back into ((r + y) << 3) + base, one more vector instruction per address in the inner loops
__device__ __forceinline__ int opaque_vgpr(int v) { asm volatile("" : "+v"(v)); return v; }
// placed between an outer and an inner condition: keeps the inner one (and its operands) out of the common path --
// without it the two side-effect-free tests are merged and both are evaluated for every candidate
__device__ __forceinline__ void rare_path() { asm volatile(""); }
__device__ __forceinline__ int tri_addr_max(int rx, int x8, int ry, int y8) {
    const int p = rx + y8, q = ry + x8;
    return p > q ? p : q;
}
// position p (wave-uniform) -> tour node, from the lane-resident copies (v_readlane takes the lane index modulo 64)
template <int SL>
__device__ __forceinline__ int lane_tour_node(const LaneTour<SL> &L, int p) {
    int r = bcast_int(L.t[0], p);
#pragma unroll
    for (int q = 1; q < SL; ++q) if (p >= q * kWave) r = bcast_int(L.t[q], p);
    return r;
}

// split [lo, hi) over `parts` consecutive chunks; chunk `c` -> [*a, *b)
__device__ __forceinline__ void chunk_range(int lo, int hi, int parts, int c, int &a, int &b) {
    const int len = hi - lo, per = (len + parts - 1) / parts;
    a = lo + c * per; b = a + per;
    if (a > hi) a = hi;
    if (b > hi) b = hi;
}

// `nwaves` wavefronts over R row blocks in proportion to the blocks' work len[0..R): wave -> (rb, part, parts).
// Every block gets at least one wavefront (callers guarantee nwaves >= R); all values are wave-uniform.
__device__ __forceinline__ void assign_waves(const int *len, int R, int nwaves, int wave, int &rb, int &part, int &parts) {
    int total = 0;
    for (int r = 0; r < R; ++r) total += len[r];
    int cnt[4], given = 0;
    for (int r = 0; r < R; ++r) { cnt[r] = 1 + (int)((long)(nwaves - R) * len[r] / (total > 0 ? total : 1)); given += cnt[r]; }
    for (int r = 0; given < nwaves; r = (r + 1) % R) { cnt[r] += 1; given += 1; }      // leftovers: heaviest (first) blocks first
    int first = 0;
    rb = R - 1; part = 0; parts = cnt[R - 1];
    for (int r = 0; r < R; ++r) {
        if (wave < first + cnt[r]) { rb = r; part = wave - first; parts = cnt[r]; break; }
        first += cnt[r];
    }
}

// ---- quiet rows of the relocate descent scan (best improvement, symmetric stores, 80 <= n <= 127 with neighbour lists on) ------
// relocate_a2a (operators.py:129-147) returns the lexicographic minimum of (delta, i, j) over the QUALIFYING moves (delta < 0 and
// not np.isclose(0, delta): delta < -1.00001e-8).  A candidate is (node b with its tour neighbours a, c; target tour edge {d, e}):
//     delta = (((((-D[a,b]) - D[b,c]) + D[a,c]) - D[d,e]) + D[d,b]) + D[b,e]
// As a real number it depends on the UNORDERED neighbour pair and the UNORDERED target edge only; the five roundings move it by
// < 2.2e-15 max|D|, so two evaluations of the same candidate in different operand orders (a reversed segment) differ by
// < 4.4e-9 for max|D| <= 1e6 (prune_ok, neighbor_lists_kernel).  Hence: a row b ALL of whose candidates -- every tour edge but
// its own two, the one the reference skips for orientation (i - j == 1) included -- evaluated >= kQuietThr = -2.5e-9 has no
// qualifying move (>= -6.9e-9 in any order) until either its neighbours change or an edge enters the tour that was not there
// when the row was evaluated.  One flag per node says "this row may hold a qualifying move":
//   * the first relocate scan of a descent is the full lean scan; it flags every row with a delta < kQuietThr (LDS bit mask);
//   * every accepted move of the descent appends its 2 or 3 NEW tour edges to a pending list and flags their endpoints (thread 0);
//   * the following relocate scans first REFRESH: node b = 1 + nwaves l + w belongs to lane l of wavefront w for the whole descent;
//     the lane recomputes the row constants of its node on the current tour and evaluates the pending edges for it (one candidate
//     per row and edge, 5 n evaluations at most, not n^2); endpoints of pending edges are flagged (their neighbours changed) --
//     then evaluate the flagged rows only, in full, with the reference's operand order and keys -- the arg-min over a superset
//     of the qualifying moves is the reference's arg-min -- and drop the flags of rows found quiet.  Flags are wave-private.
// With the bench's guide ~10 of 99 rows are flagged from the second relocate scan of a descent on (simulated on the oracle's
// trajectory: identical moves in 300 outer iterations x 3 guides, profiles/r06_experiments/quiet_rows_simulation.txt).
#ifndef GLS_QUIET_ROWS
#define GLS_QUIET_ROWS 1
#endif
constexpr double kQuietThr = -2.5e-9;
#ifndef GLS_QUIET_PEND_CAP
#define GLS_QUIET_PEND_CAP 5            // (test builds set 2: the overflow path -- back to a full scan -- then runs in every descent)
#endif
constexpr int kQuietPendCap = GLS_QUIET_PEND_CAP;      // new tour edges between two relocate scans: <= 3 (relocate move) + 2 (2-opt move)
// LDS of the scheme, in exchange slots of Ctl the best-improvement descent does not use (its LDS-atomic arg-min takes bytes 0 .. 23 of
// red_d and the first three ints of red_k; lmax_slot is red_d[1][7]):
//   words (red_k[0][4] .. red_k[1][3], four 64-bit words): bit q of word w <-> node 1 + 64 w + q: rows to flag at the next refresh
//          (the full scan's result, the endpoints of new edges)
//   pend  (red_d[0][4] .. red_d[1][5], five records): the new tour edges since the last relocate scan, ready for the refresh (the
//          index arithmetic is done ONCE per move, not by every wavefront per refresh): row address terms of x and y for
//          tri_addr_max, the LDS address of D[x,y], 8x | 8y << 16
//   count (red_k[0][3])
// The pruned relocate scan's long-edge list (n >= 128) moved to the remaining slots (16 bytes of red_k, its count to red_d[0][3]).
struct QuietPend { int xr, yr, laddr, x8y8; };
struct QuietLds { unsigned long long *words; QuietPend *pend; int *count; };
static_assert(kQuietPendCap * sizeof(QuietPend) <= 10 * sizeof(double), "red_d[0][4] .. red_d[1][5]");
__device__ __forceinline__ QuietLds quiet_lds(Ctl *ctl) {
    return QuietLds{reinterpret_cast<unsigned long long *>(&ctl->red_k[0][4]), reinterpret_cast<QuietPend *>(&ctl->red_d[0][4]), &ctl->red_k[0][3]};
}
__device__ __forceinline__ uint8_t *long_edge_list(Ctl *ctl) { return reinterpret_cast<uint8_t *>(&ctl->red_k[1][4]); }      // 16 positions (n <= 255)
__device__ __forceinline__ int *long_edge_count(Ctl *ctl) { return reinterpret_cast<int *>(&ctl->red_d[0][3]); }
// start of a descent (thread 0): nothing flagged, nothing pending; every record holds addresses inside the distance triangle
template <int NW>
__device__ __forceinline__ void quiet_reset(const QuietLds &q, int dbase) {
#pragma unroll
    for (int w = 0; w < NW; ++w) q.words[w] = 0ull;
    *q.count = 0;
    for (int e = 0; e < kQuietPendCap; ++e) q.pend[e] = QuietPend{dbase, dbase, dbase, 0};
}
__device__ __forceinline__ void quiet_set(unsigned long long *qmask, int node) {
    typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;
    __hip_atomic_fetch_or((lds_u64_t *)(qmask + ((node - 1) >> 6)), 1ull << ((node - 1) & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the new tour edges of an accepted move, from the OLD tour, beside apply_move -- by lanes 0 .. 2 of the workgroup's LAST wavefront,
// which has no tour position to move for n <= 64 (nwaves - 1) (one thread doing all of it was ~1,000 cycles in front of the barrier
// every other wavefront waits at):
//   2-opt (i < j): adds (t[i-1],t[j-1]), (t[i],t[j])                       (operators.py:6-11)
//   relocate:      adds (a,c), (d,b), (b,e)                                 (operators.py:76-80, 91-96)
// Their endpoints are the nodes whose neighbours change: flagged through `words` (LDS atomics: several lanes may hit one word).
template <class TT>
__device__ __forceinline__ void quiet_note_move(const QuietLds &q, int dbase, const TT *told, int op, int i, int j, int lane) {
    int *count = q.count;
    const int c0 = *count;                                   // uniform
    const int npairs = op == 0 ? 2 : 3;
    if (lane < npairs && c0 + lane < kQuietPendCap) {
        int x, y;
        if (op == 0) {
            const int lo = i < j ? i : j, hi = i < j ? j : i;
            x = told[lo - 1 + lane]; y = told[hi - 1 + lane];                      // lane 0: (t[lo-1], t[hi-1]); lane 1: (t[lo], t[hi])
        } else {
            const int jd = i < j ? j : j - 1;                                      // (d, e) = (t[jd], t[jd+1])
            // lane 0: (a, c) = (t[i-1], t[i+1]); lane 1: (d, b) = (t[jd], t[i]); lane 2: (b, e) = (t[i], t[jd+1])
            x = told[lane == 0 ? i - 1 : lane == 1 ? jd : i];
            y = told[lane == 0 ? i + 1 : lane == 1 ? i : jd + 1];
        }
        const int hi = x > y ? x : y, lo = x > y ? y : x;
        // (x or y may be the depot: kNoRow makes the lane's own row term win in tri_addr_max -- D[b,0] is the first entry of row b)
        q.pend[c0 + lane] = QuietPend{x == 0 ? kNoRow : dbase + 4 * x * (x - 1), y == 0 ? kNoRow : dbase + 4 * y * (y - 1),
                                       dbase + 4 * hi * (hi - 1) + 8 * lo, (8 * x) | ((8 * y) << 16)};
        if (x >= 1) quiet_set(q.words, x);
        if (y >= 1) quiet_set(q.words, y);
    }
    if (lane == 0) *count = c0 + npairs;                     // (> kQuietPendCap cannot happen between two relocate scans; the refresh checks anyway)
}

// Per-lane state of the scheme across the scans of a descent: the flag of the lane's node (node 1 + nwaves l + w, fixed)
struct QuietLane {
    bool act;            // the row of this lane's node may hold a qualifying move
};

// Refresh + reduced relocate scan of one wavefront (see above).  Returns false if the pending list overflowed: the caller runs
// the full scan instead (cannot happen with one 2-opt and one relocate move between two relocate scans).
// At four wavefronts per SIMD every instruction of a wavefront costs ~6-15 cycles whatever it is (profiles/r06_experiments/
// quiet_rows.md: the first version of this function was 717 instructions for 2.4 rows and took 4.5k cycles -- 60 % of the full
// lean scan it replaces), so both parts are written for instruction count: pending pairs and the row terms of the uniform node on
// the scalar unit, packed-triangle addresses in the select-free max form (tri_addr_max), the endpoints of the pending edges as a
// scalar bit mask, the row constants of a flagged row by v_readlane from the lane that owns its node.
template <bool CNT, int NP, class S, class TT>
__device__ __forceinline__ bool scan_relocate_a2a_quiet(const S &s, const TT *t, const TT *pos, const double *Ef, int n,
                                                        const QuietLds &q, QuietLane &me,
                                                        int wave, int nwaves, int lane, double &bd, int &bk, int &xe) {
    // NP = passes of 64 target edges per row = 64-bit flag words: 2 (n <= 127) or 4 (n <= 255)
    const int dbase = lds_byte_addr(s.d);
    ISA_MARK("quiet_refresh_begin");
    // level 0 reads: the pending records and their count, the rows to flag, the lane's node position, the lane's target edges
    const int npend = __builtin_amdgcn_readfirstlane(*q.count);
    const int wsh = nwaves == 2 ? 1 : nwaves == 4 ? 2 : nwaves == 8 ? 3 : 4;      // log2(nwaves): 2 .. 16 wavefronts
    const int b = 1 + (lane << wsh) + wave;                  // lane l <-> node b = 1 + nwaves l + wave
    const bool mine = b <= n - 1;
    const int bc = mine ? b : 1;
    const int i = pos[bc];
    const unsigned long long flagword = q.words[(bc - 1) >> 6];
    int dnode[NP], enode[NP];
    double len[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int k = lane + p * kWave;
        const bool ok = k <= n - 1;
        const int kc = ok ? k : 0;
        dnode[p] = t[kc]; enode[p] = t[kc + 1];
        len[p] = ok ? Ef[kc + 1] : -__builtin_inf();         // dead lanes: delta = +inf
    }
    if (npend > kQuietPendCap) return false;                 // uniform
    // level 1: the row's neighbours and edge lengths; the pending edges' lengths and the row's distances to their endpoints
    const int a = t[i - 1], c = t[i + 1];
    const double eab = Ef[i], ebc = Ef[i + 1];
    const int bXl = opaque_vgpr(dbase + 4 * bc * (bc - 1)), b8l = opaque_vgpr(8 * bc);      // row address terms of the lane's node (b >= 1)
    // pending edges in two batches (3 + 2: the registers of five candidates in flight at once spill): per record one broadcast
    // 16-byte read, then D[x,y], D[x,b], D[b,y] (records past npend are stale or quiet_reset's: valid addresses, evaluated and
    // masked out -- no branches; the lanes of an edge's endpoints read garbage: they are flagged through `words`)
    unsigned long long hitm0 = 0ull;                         // scalar
    double base = 0.0;
    auto batch = [&](auto first, auto count_, bool with_base) {
        constexpr int E0 = decltype(first)::value, EN = decltype(count_)::value;
        double lxy[EN], dxb[EN], dby[EN];
#pragma unroll
        for (int e = 0; e < EN; ++e) {
            const int4 rec = *reinterpret_cast<const int4 *>(&q.pend[E0 + e]);       // uniform address
            const int x8 = rec.w & 0xffff, y8 = (int)((unsigned)rec.w >> 16);
            lxy[e] = lds_read_f64(rec.z);
            dxb[e] = lds_read_f64(tri_addr_max(bXl, b8l, rec.x, x8));
            dby[e] = lds_read_f64(tri_addr_max(bXl, b8l, rec.y, y8));
        }
        if (with_base) {                                     // level 2 of the row constants, under the first batch's reads
            const double dac = s.dist(a, c);
            base = -eab;                                     // -D[a,b]          (operators.py:97-99, left to right)
            base = base - ebc;                               // -D[b,c]
            base = base + dac;                               // +D[a,c]
        }
#pragma unroll
        for (int e = 0; e < EN; ++e) {
            double d = base - lxy[e];
            d = d + dxb[e];
            d = d + dby[e];
            const unsigned long long m = __ballot(d < kQuietThr);
            hitm0 |= E0 + e < npend ? m : 0ull;
        }
    };
    constexpr int kFirstBatch = kQuietPendCap < 3 ? kQuietPendCap : 3;
    batch(std::integral_constant<int, 0>{}, std::integral_constant<int, kFirstBatch>{}, true);
    if constexpr (kQuietPendCap > 3) batch(std::integral_constant<int, 3>{}, std::integral_constant<int, kQuietPendCap - kFirstBatch>{}, false);
    const bool hit = ((hitm0 >> lane) & 1ull) != 0ull || ((flagword >> ((bc - 1) & 63)) & 1ull) != 0ull;
    if constexpr (CNT) xe += npend * __popcll(__ballot(mine));
    me.act = me.act || (mine && hit);
    ISA_MARK("quiet_refresh_end");

    // ---- reduced scan: lane l holds target edges k = l + 64 p -- (d, e) = (t[k], t[k+1]) with their packed row addresses and
    // D[d,e] in registers --; a row is NP passes of two random LDS reads, three fp64 adds, two compares ----
    unsigned long long rows = __ballot(me.act);
    if (!rows) return true;
    int dx[NP], d8[NP], ex[NP], e8[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int d = dnode[p], e = enode[p];
        // tri_addr_max: the depot (position 0 / n only) passes kNoRow so that the row's own address term wins
        dx[p] = opaque_vgpr(d == 0 ? kNoRow : dbase + 4 * d * (d - 1)); d8[p] = opaque_vgpr(8 * d);
        ex[p] = opaque_vgpr(e == 0 ? kNoRow : dbase + 4 * e * (e - 1)); e8[p] = opaque_vgpr(8 * e);
    }
    const long long basebits = __double_as_longlong(base);
    const int klane1 = lane + 1;                             // k - i + 1 = klane1 - i for the first pass, + 64 p for pass p
    unsigned long long quiet = 0ull;                         // rows found quiet
    ISA_MARK("quiet_row_loop");
    while (rows) {
        const int l = __ffsll((long long)rows) - 1;
        rows &= rows - 1;
        // the row's constants from the lane that owns its node
        const int ri = __builtin_amdgcn_readlane(i, l);
        const int bX = __builtin_amdgcn_readlane(bXl, l), b8 = __builtin_amdgcn_readlane(b8l, l);
        const double rbase = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(basebits >> 32), l) << 32) |
                                                  (unsigned)__builtin_amdgcn_readlane((int)basebits, l));
        double vd[NP], ve[NP], delta[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {                       // (dead lanes read a valid address, their delta is +inf)
            vd[p] = lds_read_f64(tri_addr_max(dx[p], d8[p], bX, b8));       // D[d,b]  (garbage on the lanes k in {i-1, i}: masked below)
            ve[p] = lds_read_f64(tri_addr_max(ex[p], e8[p], bX, b8));       // D[b,e]
        }
        const unsigned u0 = (unsigned)(klane1 - ri);         // quiet test: every target edge but the row's own two, (unsigned)(k - i + 1) < 2
        unsigned long long hitm = 0ull;
        double dmin = __builtin_inf();
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            double d = rbase - len[p];                       // -D[d,e]          (operators.py:100-102, left to right)
            d = d + vd[p];                                   // +D[d,b]
            d = d + ve[p];                                   // +D[b,e]
            delta[p] = d;
            hitm |= __ballot(d < kQuietThr) & ~__ballot(u0 + (unsigned)(p * kWave) < 2u);
            dmin = p == 0 ? d : min_f64_raw(dmin, d);
        }
        if constexpr (CNT) xe += n;
        if (dmin <= bd) {
            rare_path();
            ISA_MARK("quiet_row_rare");
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const double d = delta[p];
                const int kk = lane + p * kWave;
                // valid targets: k not in {i-2, i-1, i} (operators.py:133-136: i - j == 1 <=> k = i - 2)
                if (d <= bd && (unsigned)(kk - ri + 2) > 2u && !close_to_zero(d)) {
                    const int key = make_key(ri, kk < ri ? kk + 1 : kk);
                    if (d < bd || key < bk) { bd = d; bk = key; }
                }
            }
        }
        ISA_MARK("quiet_row_tail");
        if (!hitm) quiet |= 1ull << l;
    }
    ISA_MARK("quiet_scan_end");
    if ((quiet >> lane) & 1ull) me.act = false;
    return true;
}

// pos != nullptr: lane l of row block rb owns NODE b = 1 + 64 rb + l (wherever it sits in the tour: i = pos[b]) instead of
// tour POSITION 1 + 64 rb + l.  The random read of a step is D[b, e] with e wave-uniform: with consecutive node ids on the
// lanes the half of the lanes with b < e reads 64 consecutive doubles of row e and the other half a fixed quadratic
// pattern (b(b-1)/2 + e), instead of 64 arbitrary rows / columns: simulated 2.5 instead of 4.9 bank passes per
// ds_read_b64 at n = 100.  Keys (i, j) and deltas are the same set; within a lane they still ascend with k.
// MF: the group filter above; UR: steps per group -- both chosen by the caller from the register budget of the build (the filter costs
// 16-28 B of scratch on the 64- / 80-VGPR builds, six steps per group 8-20 B on the single-slot 128-VGPR ones: those keep round 4's code)
template <int SL, bool MF, int UR, class S, class TT, bool QT = false>
__device__ __forceinline__ void scan_relocate_a2a_lean(const S &s, const TT *t, const double *Ef, int n,
                                                       int wave, int nwaves, int lane, double &bd, int &bk,
                                                       const uint8_t *pos = nullptr, unsigned long long *qmask = nullptr) {
    const int RW = (n - 1 + kWave - 1) / kWave;              // row blocks of 64 rows
    const int per_rb = nwaves / RW;                          // waves sharing a row block, each a contiguous k range
    const int rb = __builtin_amdgcn_readfirstlane(wave / (per_rb > 0 ? per_rb : 1));
    const int part = __builtin_amdgcn_readfirstlane(wave - rb * per_rb);
    if (per_rb == 0 || rb >= RW) return;                     // callers guarantee nwaves >= RW; surplus waves idle
    ISA_MARK("reloc_lean_begin");
    const LaneTour<SL> L = load_lane_tour<SL>(t, n, lane);
    const int own = 1 + rb * kWave + lane;                   // the lane's row: a position, or a node id (pos != nullptr)
    const bool row_ok = own <= n - 1;                        // permutations(range(1,n),2), skip i-j == 1 (operators.py:133-136)
    const int i = pos ? (int)pos[row_ok ? own : 1] : own;
    const int ic = row_ok ? i : 1;
    const int a = t[ic - 1], b = pos ? (row_ok ? own : (int)t[1]) : (int)t[ic], cc = t[ic + 1];
    double base = -Ef[ic];                                   // -D[a,b]
    base = base - Ef[ic + 1];                                // -D[b,c]
    base = base + s.dist(a, cc);                             // +D[a,c]
    if (!row_ok) base = __builtin_inf();                     // delta = +inf: never below the best (bd <= 0)
    const int b2 = (b * (b - 1)) >> 1;
    const int dbase = lds_byte_addr(s.d);
    const int bx = opaque_vgpr(dbase + 8 * b2), b8 = opaque_vgpr(8 * b);      // b = t[i] >= 1
    int k0, k1;
    chunk_range(0, n, per_rb, part, k0, k1);                 // target edges k = 0 .. n-1
    if (k0 >= k1) return;
    const int d0 = lane_tour_node(L, k0);
    double vd = s.dist_at(s.idx2(b, b2, d0, (d0 * (d0 - 1)) >> 1));       // D[t[k0], b]   (garbage, unused, where t[k0] == b)
    double rowmin = __builtin_inf();                         // QT: minimum of every delta this lane forms (quiet rows, see scan_relocate_a2a_quiet)
    // U steps at a time: all wave-uniform operands, addresses and the U random distance reads are issued before the first
    // dependent add, so the LDS round trips of a group overlap (a step alone is a ~280-cycle dependent chain)
    auto group = [&](int k, int te, auto ucount, auto fast_addr) {        // te: the register slot holding positions k+1 .. k+U
        constexpr int U = decltype(ucount)::value;
        constexpr bool FA = decltype(fast_addr)::value;      // every t[k+1] of the group is a node >= 1
        if constexpr (U > 1) ISA_MARK("reloc_lean_group"); else ISA_MARK("reloc_lean_single");
        double ve[U], de[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = bcast_int(te, k + u + 1);          // v_readlane uses the lane index modulo 64
            const int e2 = (e * (e - 1)) >> 1;               // wave-uniform: scalar ALU
            de[u] = Ef[k + u + 1];                           // D[t[k], t[k+1]]: wave-uniform address, one broadcast LDS read
            if constexpr (FA) ve[u] = lds_read_f64(tri_addr_max(bx, b8, dbase + 4 * e * (e - 1), 8 * e));   // 8 e(e-1)/2, no shift pair
            else ve[u] = s.dist_at(s.idx2(b, b2, e, e2));    // D[b, t[k+1]]
        }
        double dl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double delta = base - de[u];                     // -D[d,e]          (operators.py:100-102, left to right)
            delta = delta + vd;                              // +D[d,b]
            dl[u] = delta + ve[u];                           // +D[b,e]
            vd = ve[u];
        }
        if constexpr (QT) {
            const double gm = min_tree<0, U>(dl);            // (the group filter's minimum: one more v_min_f64 per group keeps the row's)
            rowmin = min_f64_raw(rowmin, gm);
            if (MF && GLS_LEAN_MINFILTER && U > 1 && !(gm < bd)) return;
        } else {
            if (!group_may_improve<MF>(dl, bd)) return;      // one exec-masked region per group instead of one per step
        }
        ISA_MARK("reloc_lean_candidates");
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = k + u;
            const double delta = dl[u];
            if (delta < bd) {
                rare_path();
                // valid targets: k <= i-3 (j = k+1 < i-1) or k >= i+1 (j = k); k in {i-2, i-1, i} <=> (unsigned)(k - i + 2) <= 2
                if ((unsigned)(kk - i + 2) > 2u && !close_to_zero(delta)) { bd = delta; bk = make_key(i, kk < i ? kk + 1 : kk); }
            }
        }
    };
    using UN = std::integral_constant<int, UR>;
    using U1 = std::integral_constant<int, 1>;
    using FAST = std::integral_constant<bool, true>;
    using EXACT = std::integral_constant<bool, false>;
    const int k1f = k1 == n ? n - 1 : k1;                    // step k = n-1 meets t[n] = node 0: exact index, after the loops
#pragma unroll
    for (int q = 0; q < SL; ++q) {                           // slot q holds positions 64q .. 64q+63, i.e. k + 1 of k in [64q-1, 64q+62]
        const int lo = q * kWave - 1, hi = q * kWave + kWave - 1;
        int k = k0 > lo ? k0 : lo;
        const int ke = k1f < hi ? k1f : hi;
        for (; k + UR <= ke; k += UR) group(k, L.t[q], UN{}, FAST{});
        for (; k < ke; ++k) group(k, L.t[q], U1{}, FAST{});
    }
    if (k1f != k1) {                                         // keys ascend with k within a lane: the last step stays last
#pragma unroll
        for (int q = 0; q < SL; ++q)                         // the slot that holds position n
            if (n / kWave == q) group(n - 1, L.t[q], U1{}, EXACT{});
    }
    ISA_MARK("reloc_lean_end");
    if constexpr (QT) {
        // rows some delta of which is below the quiet threshold stay active (the two garbage steps k in {i-1, i} included: a spurious
        // bit costs one exact re-evaluation of the row by the next reduced scan, never a lost move)
        if (row_ok && rowmin < kQuietThr) quiet_set(qmask, b);
    }
}

template <int SL, bool MF, class S, class TT>
__device__ __forceinline__ void scan_two_opt_a2a_lean(const S &s, const TT *t, const double *Eb, int n,
                                                      int wave, int nwaves, int lane, double &bd, int &bk) {
    // combinations(range(1,n),2), |i-j| >= 2 (operators.py:36-39): rows i = 1..n-3, j = i+2..n-1.  Row block rb can use
    // j >= 3 + 64 rb: the first block has the most work, so the waves are shared out in proportion to the j ranges.
    const int RW = (n - 3 + kWave - 1) / kWave;              // row blocks with at least one valid row
    if (RW <= 0) return;
    int len[4] = {0, 0, 0, 0};
    for (int r = 0; r < RW && r < 4; ++r) len[r] = n - 3 - r * kWave;        // number of j values block r walks
    int rb, part, parts;
    assign_waves(len, RW, nwaves, wave, rb, part, parts);
    rb = __builtin_amdgcn_readfirstlane(rb); part = __builtin_amdgcn_readfirstlane(part); parts = __builtin_amdgcn_readfirstlane(parts);
    const LaneTour<SL> L = load_lane_tour<SL>(t, n, lane);
    const int i = 1 + rb * kWave + lane;
    const bool row_ok = i <= n - 3;
    const int ic = row_ok ? i : 1;
    const int a = t[ic], b = t[ic - 1];
    const int a2 = (a * (a - 1)) >> 1, b2 = (b * (b - 1)) >> 1;
    const double eab = row_ok ? Eb[ic] : -__builtin_inf();   // D[a,b]; rows past the end: delta = +inf, never below the best
    const int dbase = lds_byte_addr(s.d);
    const int ax = opaque_vgpr(dbase + 8 * a2), a8 = opaque_vgpr(8 * a);      // a = t[i] >= 1; c = t[j] >= 1 and d = t[j-1] >= 1 (j >= 3)
    const int bx = opaque_vgpr(b == 0 ? kNoRow : dbase + 8 * b2), b8 = opaque_vgpr(8 * b);   // b = t[i-1] is node 0 on row 1: D[0,d] sits in row d, column 0
    int j0, j1;
    chunk_range(3 + rb * kWave, n, parts, part, j0, j1);     // j = j0 .. j1-1
    if (j0 >= j1) return;
    int d = lane_tour_node(L, j0 - 1);
    int d2 = dbase + 4 * d * (d - 1);
    auto group = [&](int j, int tj, auto ucount) {           // tj: the register slot holding positions j .. j+U-1
        constexpr int U = decltype(ucount)::value;
        double vac[U], vbd[U], ecd[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = bcast_int(tj, j + u);
            ecd[u] = Eb[j + u];                              // D[c,d]: wave-uniform address, one broadcast LDS read
            const int c2 = dbase + 4 * c * (c - 1);          // wave-uniform row address 8 c(c-1)/2: scalar ALU
            vac[u] = lds_read_f64(tri_addr_max(ax, a8, c2, 8 * c));      // D[a,c]
            vbd[u] = lds_read_f64(tri_addr_max(bx, b8, d2, 8 * d));      // D[b,d]
            d = c; d2 = c2;
        }
        double dl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double delta = vac[u] + vbd[u];                  // operators.py:25-28, left to right
            delta = delta - eab;
            dl[u] = delta - ecd[u];
        }
        if (!group_may_improve<MF>(dl, bd)) return;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double delta = dl[u];
            if (delta < bd) {                                // j < i + 2 holds the mirrored move's delta: rarely below the best either
                rare_path();
                if (j + u >= i + 2 && !close_to_zero(delta)) { bd = delta; bk = make_key(i, j + u); }
            }
        }
    };
    using UN = std::integral_constant<int, GLS_LEAN_UNROLL>;
    using U1 = std::integral_constant<int, 1>;
#pragma unroll
    for (int q = 0; q < SL; ++q) {                           // slot q holds positions j in [64q, 64q+63]
        int j = j0 > q * kWave ? j0 : q * kWave;
        const int je = j1 < (q + 1) * kWave ? j1 : (q + 1) * kWave;
        for (; j + GLS_LEAN_UNROLL <= je; j += GLS_LEAN_UNROLL) group(j, L.t[q], UN{});
        for (; j < je; ++j) group(j, L.t[q], U1{});
    }
}

// ---- lean scans on the two 32-lane halves of ONE wavefront (single-wavefront workgroups, n <= 33: TSP20) ------------------
// With at most 32 rows the upper half of the wavefront idles in the scans above.  Here lane (h, l) = (lane >> 5, lane & 31)
// owns row 1 + l in both halves and half h walks half of the other index (relocate: target edges k, 2-opt: j), so a scan
// takes half the steps.  The node of a step is no longer wave-uniform: two v_readlane and a select per step, and its row
// address is vector arithmetic.  Same deltas (same operands, same order), same keys; within a lane the keys still ascend.
#ifndef GLS_HALF_SCANS
#define GLS_HALF_SCANS 1
#endif
#ifndef GLS_HALF_UNROLL
#define GLS_HALF_UNROLL 3             // steps per group of the half-wave scans (4: 2 % faster at TSP20, 8 B of scratch)
#endif
constexpr int kHalfScanMinNodes = 8, kHalfScanMaxNodes = 33;

template <int HUN, class S, class TT>
__device__ __forceinline__ void scan_relocate_a2a_lean_half(const S &s, const TT *t, const double *Ef, int n, int lane,
                                                            double &bd, int &bk) {
    const bool hi = lane >= 32;
    const int tl = t[lane <= n ? lane : n];                  // tour position `lane` (n <= 33: one register per lane)
    const int K0 = (n + 1) >> 1;                             // half 0: k = 0 .. K0-1; half 1: k = K0 .. n-1
    const int kofs = hi ? K0 : 0;
    const int i = 1 + (lane & 31);
    const bool row_ok = i <= n - 1;                          // permutations(range(1,n),2), skip i-j == 1 (operators.py:133-136)
    const int ic = row_ok ? i : 1;
    const int a = t[ic - 1], b = t[ic], cc = t[ic + 1];
    double base = -Ef[ic];                                   // -D[a,b]
    base = base - Ef[ic + 1];                                // -D[b,c]
    base = base + s.dist(a, cc);                             // +D[a,c]
    if (!row_ok) base = __builtin_inf();                     // delta = +inf: never below the best (bd <= 0)
    const int b2 = (b * (b - 1)) >> 1;
    const int dbase = lds_byte_addr(s.d);
    const int bx = opaque_vgpr(dbase + 8 * b2), b8 = opaque_vgpr(8 * b);      // b = t[i] >= 1
    const int d0a = bcast_int(tl, 0), d0b = bcast_int(tl, K0);
    const int d0 = hi ? d0b : d0a;                           // t[k] of the half's first step
    double vd = s.dist_at(s.idx2(b, b2, d0, (d0 * (d0 - 1)) >> 1));           // D[t[k0], b]   (garbage, unused, where t[k0] == b)
    const int efb = opaque_vgpr(lds_byte_addr(Ef) + 8 * kofs);                // address of Ef[k] of the half's first step
    const int live1 = n - K0;                                // steps of half 1 (K0, or K0 - 1 for odd n)
    auto group = [&](int ss, auto ucount, auto safe_addr) {
        constexpr int U = decltype(ucount)::value;
        constexpr bool SAFE = decltype(safe_addr)::value;    // a step of the group may meet node 0 (k = n-1: t[n]) or be idle in half 1
        double ve[U], de[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e0 = bcast_int(tl, ss + u + 1), e1 = bcast_int(tl, ss + u + 1 + K0);
            const int e = hi ? e1 : e0;                      // t[k+1]
            de[u] = lds_read_f64(efb + 8 * (ss + u + 1));    // D[t[k], t[k+1]]
            const int fast = tri_addr_max(bx, b8, dbase + 4 * e * (e - 1), 8 * e);
            ve[u] = lds_read_f64(SAFE ? (e == 0 ? bx : fast) : fast);         // D[b, t[k+1]]; D[b, 0] is the first entry of row b
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = ss + u + kofs;
            double delta = base - de[u];                     // -D[d,e]          (operators.py:100-102, left to right)
            delta = delta + vd;                              // +D[d,b]
            delta = delta + ve[u];                           // +D[b,e]
            vd = ve[u];
            if (SAFE && hi && ss + u >= live1) delta = __builtin_inf();
            if (delta < bd) {
                rare_path();
                // valid targets: k <= i-3 (j = k+1 < i-1) or k >= i+1 (j = k); k in {i-2, i-1, i} <=> (unsigned)(k - i + 2) <= 2
                if ((unsigned)(kk - i + 2) > 2u && !close_to_zero(delta)) { bd = delta; bk = make_key(i, kk < i ? kk + 1 : kk); }
            }
        }
    };
    using UN = std::integral_constant<int, HUN>;
    using U1 = std::integral_constant<int, 1>;
    using FAST = std::integral_constant<bool, false>;
    using SAFE = std::integral_constant<bool, true>;
    int ss = 0;
    for (; ss + HUN <= K0 - 2; ss += HUN) group(ss, UN{}, FAST{});
    for (; ss < K0 - 2; ++ss) group(ss, U1{}, FAST{});
    for (; ss < K0; ++ss) group(ss, U1{}, SAFE{});           // the last two steps: node 0 closes the tour; odd n: half 1 is one short
}

template <int HUN, class S, class TT>
__device__ __forceinline__ void scan_two_opt_a2a_lean_half(const S &s, const TT *t, const double *Eb, int n, int lane,
                                                           double &bd, int &bk) {
    // combinations(range(1,n),2), |i-j| >= 2 (operators.py:36-39): rows i = 1..n-3, j = 3..n-1 (j >= i + 2 checked late)
    const bool hi = lane >= 32;
    const int tl = t[lane <= n ? lane : n];
    const int cnt = n - 3, J0 = (cnt + 1) >> 1;              // half 0: j = 3 .. 2+J0; half 1: j = 3+J0 .. n-1
    const int jofs = hi ? J0 : 0;
    const int i = 1 + (lane & 31);
    const bool row_ok = i <= n - 3;
    const int ic = row_ok ? i : 1;
    const int a = t[ic], b = t[ic - 1];
    const int a2 = (a * (a - 1)) >> 1, b2 = (b * (b - 1)) >> 1;
    const double eab = row_ok ? Eb[ic] : -__builtin_inf();   // D[a,b]; rows past the end: delta = +inf, never below the best
    const int dbase = lds_byte_addr(s.d);
    const int ax = opaque_vgpr(dbase + 8 * a2), a8 = opaque_vgpr(8 * a);      // a = t[i] >= 1; c = t[j] >= 1 and d = t[j-1] >= 1 (j >= 3)
    const int bx = opaque_vgpr(b == 0 ? kNoRow : dbase + 8 * b2), b8 = opaque_vgpr(8 * b);   // b = t[i-1] is node 0 on row 1
    const int da = bcast_int(tl, 2), db = bcast_int(tl, 2 + J0);
    int d = hi ? db : da;                                    // t[j-1] of the half's first step
    int d2 = dbase + 4 * d * (d - 1);
    const int ebb = opaque_vgpr(lds_byte_addr(Eb) + 8 * jofs);
    const int live1 = cnt - J0;                              // steps of half 1 (J0, or J0 - 1 for odd n - 3)
    auto group = [&](int ss, auto ucount, auto tail) {
        constexpr int U = decltype(ucount)::value;
        constexpr bool TAIL = decltype(tail)::value;
        double vac[U], vbd[U], ecd[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c0 = bcast_int(tl, 3 + ss + u), c1 = bcast_int(tl, 3 + ss + u + J0);
            const int c = hi ? c1 : c0;                      // t[j]
            ecd[u] = lds_read_f64(ebb + 8 * (3 + ss + u));   // D[c,d]
            const int c2 = dbase + 4 * c * (c - 1);
            vac[u] = lds_read_f64(tri_addr_max(ax, a8, c2, 8 * c));      // D[a,c]
            vbd[u] = lds_read_f64(tri_addr_max(bx, b8, d2, 8 * d));      // D[b,d]
            d = c; d2 = c2;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = 3 + ss + u + jofs;
            double delta = vac[u] + vbd[u];                  // operators.py:25-28, left to right
            delta = delta - eab;
            delta = delta - ecd[u];
            if (TAIL && hi && ss + u >= live1) delta = __builtin_inf();
            if (delta < bd) {
                rare_path();
                if (j >= i + 2 && !close_to_zero(delta)) { bd = delta; bk = make_key(i, j); }
            }
        }
    };
    using UN = std::integral_constant<int, HUN>;
    using U1 = std::integral_constant<int, 1>;
    using BODY = std::integral_constant<bool, false>;
    using TAIL = std::integral_constant<bool, true>;
    int ss = 0;
    for (; ss + HUN <= J0 - 1; ss += HUN) group(ss, UN{}, BODY{});
    for (; ss < J0 - 1; ++ss) group(ss, U1{}, BODY{});
    for (; ss < J0; ++ss) group(ss, U1{}, TAIL{});           // odd n - 3: half 1 is one step short
}

// ---- pruned a2a scans (best improvement, symmetric stores; 2-opt from n = 80, relocate from n = 128) -------------------
// The descent needs, per scan, the lexicographic minimum of (delta, i, j) over the moves that qualify (delta < 0 and not
// np.isclose(0, delta), operators.py:42).  Any SUPERSET of the qualifying moves gives the same minimum, and most of the
// O(n^2) moves of a tour that is a few moves away from a local optimum cannot qualify:
//   2-opt   delta = ((D[a,c] + D[b,d]) - D[a,b]) - D[c,d] < 0 needs D[a,c] < D[a,b] or D[b,d] < D[c,d] (exactly so in real
//           arithmetic; the three roundings move delta by < 1.4e-15 max|D| and a qualifying delta is below -1e-8, so with
//           max|D| <= 1e6 -- checked once per instance by neighbor_lists_kernel, else the full scans run -- no qualifying
//           move is lost): only pairs (x, y) with y closer to x than one of x's two tour neighbours;
//   relocate delta = ((base_i - D[d,e]) + D[d,b]) + D[b,e] with T = fl(D[d,e] - base_i): if 2 D[d,b] >= T and 2 D[b,e] >= T
//           then fl(-T + D[d,b]) >= -T/2 and the last sum is >= 0 (rounding is monotone, T/2 is exact): a qualifying
//           move has an endpoint y of its target edge with 2 D[b,y] < T, for ANY symmetric matrix (no triangle inequality).
// Every node has a list of its kNL = 32 nearest nodes, ascending (ids, one byte each, built once per instance by
// neighbor_lists_kernel into global memory: 6.4 KB per TSP200 instance, L1-resident; the distances come from the LDS
// triangle).  Eight lanes share a tour row (position p, node x = t[p]) and take two list entries each per level of 16,
// eight rows per wavefront.  Entries that
// pass the row's threshold look their node's position up and the few surviving candidates (~1000 / ~250 of 19,503 /
// 39,204 moves per scan at n = 200) are evaluated with the reference's operand order -- same operands, same bits as the
// full scans -- all at once under the exec mask.  A row whose 16th entry still passes takes entries 17..32 in a second
// step; a row whose 32nd entry passes is evaluated in full by its wavefront (rare).  Tour-edge lengths are bounded by
// Lmax (an upper bound of Ef[], kept by local_search_dev).
constexpr int kNL = 32;
// The rows of the pruned scans belong to NODES (lane group g of pass `it` owns node 1 + (it * nthr + tid) / 8 for the whole
// kernel), so a lane's four list entries per row -- entries m, m + 8 (level 0) and 16 + m, 24 + m (level 1) of its node's
// list -- are four bytes of ONE register per pass, loaded once at kernel start: no memory access for the ids in a scan,
// and the first reads of a row (its position, the distances to its list entries and their positions) are independent.
constexpr int kNlPasses = 4;             // 8 (n - 1) tasks over >= 256 (n <= 127) / >= 512 (n <= 255) threads
struct NlWords {
    unsigned w0, w1, w2, w3;             // (plain members: an indexed array ends up in scratch memory)
    __device__ __forceinline__ unsigned of(int it) const {           // `it` is wave-uniform
        return it == 0 ? w0 : it == 1 ? w1 : it == 2 ? w2 : w3;
    }
};
constexpr int kPruneMinNodes = 80;       // 2-opt scan pruned from here up (same-box A/B at n = 66 .. 127), relocate from n = 128

template <bool CNT, class S, class TT>
__device__ __forceinline__ void scan_two_opt_a2a_pruned(const S &s, const TT *t, const TT *pos, const double *Ef,
                                                        const NlWords &nlw, int n,
                                                        int tid, int nthr, int lane, double &bd, int &bk, int &xe,
                                                        long long *dbg = nullptr) {
    // CNT (the counting instantiations, GlsArgs::evals_exec): xe (wave-uniform, scalar registers) += delta evaluations this
    // wavefront executes -- candidates under the exec mask, whole rows on overflow.  Measured cost of counting: 1.8-3.2 % of the
    // outer iterations (profiles/r04_ab_exec_counter.log), hence instantiations of their own that only bench.py's counting
    // pass launches
    const PlainDist<S> f{s};
    const int tasks = 8 * (n - 1);                           // 8 lanes per tour row, two list entries per lane and level
    const int rowbit = (lane & 56) + 7;                      // lane that holds entries 15 / 31 of this lane's row
    int it = 0;
    ISA_MARK("twoopt_pruned_begin");
    for (int task0 = 0; task0 < tasks; task0 += nthr, ++it) {      // wave-uniform trip count; a wavefront's tasks are whole rows
        ISA_MARK("twoopt_pruned_pass");
        const int task = task0 + tid;
        const bool live = task < tasks;
#if GLS_SKIP_DEAD_PASS
        // a wavefront none of whose lanes has a row in this pass (n = 100: 792 tasks on 256 threads -- the fourth pass only
        // has rows for wavefront 0) leaves the scan here: later passes have none for it either
        if (__builtin_amdgcn_readfirstlane(task0 + (tid & ~(kWave - 1))) >= tasks) break;
#endif
        // the row of NODE x (every node but the depot has one), wherever it sits in the tour: p = pos[x]
        const int x = 1 + (live ? task >> 3 : 0), m = task & 7;
        const unsigned ids4 = nlw.of(it);
        const int p = pos[x];
        const int xm = t[p - 1], xp = t[p + 1];
        const double ep = Ef[p], es = Ef[p + 1];             // D[x, t[p-1]], D[x, t[p+1]]
        const double thr = ep > es ? ep : es;
        bool more = live;                                    // the row may hold candidates among its next 16 list entries
#pragma unroll 1
        for (int lvl = 0; lvl < 2; ++lvl) {
            ISA_MARK("twoopt_pruned_level");
            const int y0 = (ids4 >> (16 * lvl)) & 0xff, y1 = (ids4 >> (16 * lvl + 8)) & 0xff;     // entries m and m + 8 of this level
            const double d0 = s.dist(x, y0), d1 = s.dist(x, y1);
            const int q0 = pos[y0], q1 = pos[y1];
            bool last = false;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const double d = u ? d1 : d0;
                const int q = u ? q1 : q0;
                const bool act = more && d < thr;
                // y = t[q]: (a, c) = (x, y) of the move (i, j) = (p, q) if q >= p + 2; (d, b) = (x, y) of (q + 1, p + 1) if q <= p - 2
                const bool ca = act && d < ep && q >= p + 2;
                const bool cb = act && d < es && q <= p - 2 && p <= n - 2;
                if constexpr (CNT) xe += __popcll(__ballot(ca || cb));
                if (ca || cb) {
                    ISA_MARK("twoopt_pruned_candidate");
                    const int o2 = t[ca ? q - 1 : q + 1];
                    const double e2 = Ef[ca ? q : q + 1];
                    const double dpair = s.dist(ca ? xm : xp, o2);       // D[b,d] (A) / D[a,c] (B)
                    double delta = d + dpair;                            // operators.py:25-28 (the sum of two terms commutes)
                    delta = delta - (ca ? ep : e2);                      // - D[a,b]
                    delta = delta - (ca ? e2 : es);                      // - D[c,d]
                    consider<false>(delta, ca ? make_key(p, q) : make_key(q + 1, p + 1), bd, bk);
                }
                if (u == 1) last = act;
            }
            const unsigned long long need = __ballot(last && m == 7);    // rows whose last entry of this level still passes
            more = (need >> rowbit) & 1ull;
            if (!need) break;
        }
        // all 32 entries below the row's threshold: every move that has x as `a` (row p) or as `d` (column p + 1)
        ISA_MARK("twoopt_pruned_overflow");
        unsigned long long om = __ballot(more && m == 7);
#ifdef GLS_STAMPS
        if (dbg && lane == 0) { dbg[0] += __popcll(om); dbg[2] += 1; }
#endif
        while (om) {
            const int src = __ffsll((long long)om) - 1;
            om &= om - 1;
            const int pr = __builtin_amdgcn_readlane(p, src);
            if constexpr (CNT) xe += (n - 2 - pr > 0 ? n - 2 - pr : 0) + (pr + 1 <= n - 1 ? pr - 1 : 0);
            for (int j = pr + 2 + lane; j <= n - 1; j += kWave) consider<false>(two_opt_cost(t, f, pr, j), make_key(pr, j), bd, bk);
            if (pr + 1 <= n - 1)
                for (int i = 1 + lane; i <= pr - 1; i += kWave) consider<false>(two_opt_cost(t, f, i, pr + 1), make_key(i, pr + 1), bd, bk);
        }
    }
}

// QT (quiet rows, see scan_relocate_a2a_quiet): also flag, in qwords, every row with an evaluated candidate below the quiet threshold.
// A candidate this scan prunes away has delta >= 0 (the argument above), so a row it does not flag is quiet; the candidate the
// reference skips for orientation (k = p - 2) is evaluated for the flag, never considered as a move.
template <bool CNT, bool QT, class S, class TT>
__device__ __forceinline__ void scan_relocate_a2a_pruned(const S &s, const TT *t, const TT *pos, const double *Ef,
                                                         const NlWords &nlw, int n, double Lcap,
                                                         const uint8_t *longk, int nlong,
                                                         int tid, int nthr, int lane, double &bd, int &bk, int &xe,
                                                         long long *dbg = nullptr, unsigned long long *qwords = nullptr) {
    const PlainDist<S> f{s};
    const int tasks = 8 * (n - 1);
    const int rowbit = (lane & 56) + 7;
    int it = 0;
    for (int task0 = 0; task0 < tasks; task0 += nthr, ++it) {
        const int task = task0 + tid;
        const bool live = task < tasks;
#if GLS_SKIP_DEAD_PASS
        if (__builtin_amdgcn_readfirstlane(task0 + (tid & ~(kWave - 1))) >= tasks) break;      // see scan_two_opt_a2a_pruned
#endif
        const int b = 1 + (live ? task >> 3 : 0), m = task & 7;      // the row of NODE b, at tour position p
        const unsigned ids4 = nlw.of(it);
        const int p = pos[b];
        double base = -Ef[p];                                    // -D[a,b]          (operators.py:97-99, left to right)
        base = base - Ef[p + 1];                                 // -D[b,c]
        base = base + s.dist(t[p - 1], t[p + 1]);                // +D[a,c]
        // Target edges no longer than Lcap: fl(D[d,e] - base) <= Tmax, so an endpoint of a qualifying move's target edge has
        // 2 D[b,y] < Tmax and sits in the list prefix walked below.  The (few) longer tour edges are in longk[] and every
        // row evaluates them directly: a single long edge left by the perturbation phase would otherwise push every row's
        // threshold beyond its list (12.9 of 16 rows per wavefront overflowed with the tour's maximum edge as the bound).
        const double Tmax = Lcap - base;
        bool qhit = false;                                       // QT: an evaluated candidate of this lane's row is below the quiet threshold
        // targets evaluated: k not in {p-1, p} with QT (the row's own two edges), k not in {p-2, p-1, p} else
        auto target_ok = [&](int k) { return QT ? (unsigned)(k - p + 1) > 1u : (unsigned)(k - p + 2) > 2u; };
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool lk = live && m + 8 * u < nlong;
            const int k = lk ? (int)longk[m + 8 * u] : p;        // (k = p is never a valid target)
            const bool ev = target_ok(k);
            if constexpr (CNT) xe += __popcll(__ballot(ev));
            if (ev) {
                double delta = base - Ef[k + 1];                 // operators.py:100-102, left to right
                delta = delta + s.dist(t[k], b);                 // +D[d,b]
                delta = delta + s.dist(b, t[k + 1]);             // +D[b,e]
                if (QT) qhit = qhit || delta < kQuietThr;
                if (!QT || k != p - 2) consider<false>(delta, make_key(p, k < p ? k + 1 : k), bd, bk);
            }
        }
        bool more = live;
#pragma unroll 1
        for (int lvl = 0; lvl < 2; ++lvl) {
            const int y0 = (ids4 >> (16 * lvl)) & 0xff, y1 = (ids4 >> (16 * lvl + 8)) & 0xff;
            const double d0 = s.dist(b, y0), d1 = s.dist(b, y1);
            const int q0 = pos[y0], q1 = pos[y1];
            // y = t[q] is d of target edge k1 = q and e of target edge k2 = q - 1 (the depot closes the tour: e = t[n])
            const int k20 = y0 == 0 ? n - 1 : q0 - 1, k21 = y1 == 0 ? n - 1 : q1 - 1;
            const double e10 = Ef[q0 + 1], e20 = Ef[k20 + 1], e11 = Ef[q1 + 1], e21 = Ef[k21 + 1];     // D[d,e]
            bool last = false;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const double d = u ? d1 : d0, e1 = u ? e11 : e10, e2 = u ? e21 : e20;
                const int k1 = u ? q1 : q0, k2 = u ? k21 : k20;
                const double two_d = d + d;
                const bool act = more && two_d < Tmax;
                // valid targets of row p: k not in {p-2, p-1, p} (operators.py:133-136: i - j == 1 <=> k = p - 2)
                const bool c1 = act && target_ok(k1) && two_d < e1 - base;
                const bool c2 = act && target_ok(k2) && two_d < e2 - base;
                if constexpr (CNT) xe += __popcll(__ballot(c1)) + __popcll(__ballot(c2));
                if (c1) {
                    double delta = base - e1;                    // operators.py:100-102, left to right
                    delta = delta + d;                           // +D[d,b]
                    delta = delta + s.dist(b, t[k1 + 1]);        // +D[b,e]
                    if (QT) qhit = qhit || delta < kQuietThr;
                    if (!QT || k1 != p - 2) consider<false>(delta, make_key(p, k1 < p ? k1 + 1 : k1), bd, bk);
                }
                if (c2) {
                    double delta = base - e2;
                    delta = delta + s.dist(t[k2], b);            // +D[d,b]
                    delta = delta + d;                           // +D[b,e]
                    if (QT) qhit = qhit || delta < kQuietThr;
                    if (!QT || k2 != p - 2) consider<false>(delta, make_key(p, k2 < p ? k2 + 1 : k2), bd, bk);
                }
                if (u == 1) last = act;
            }
            const unsigned long long need = __ballot(last && m == 7);
            more = (need >> rowbit) & 1ull;
            if (!need) break;
        }
        unsigned long long om = __ballot(more && m == 7);
#ifdef GLS_STAMPS
        if (dbg && lane == 0) { dbg[1] += __popcll(om); dbg[3] += 1; }
#endif
        while (om) {                                             // whole row i = pr (operators.py:133-136)
            const int src = __ffsll((long long)om) - 1;
            om &= om - 1;
            const int pr = __builtin_amdgcn_readlane(p, src);
            if constexpr (CNT) xe += n - 2 - (pr > 1 ? 1 : 0);
            bool rhit = false;
            for (int j = 1 + lane; j <= n - 1; j += kWave) {
                if (j == pr || (!QT && pr - j == 1)) continue;
                const double delta = relocate_cost(t, f, pr, j);
                if (QT) rhit = rhit || delta < kQuietThr;
                if (!QT || pr - j != 1) consider<false>(delta, make_key(pr, j), bd, bk);
            }
            if (QT && rhit) quiet_set(qwords, (int)t[pr]);       // (pr >= 1: never the depot)
        }
        if (QT && qhit && live) quiet_set(qwords, b);
    }
}

// wave-wide maximum of doubles (order-preserving integer image, two 32-bit DPP reductions); result uniform
__device__ __forceinline__ unsigned long long wave_max_sortable(unsigned long long sk) {
    const unsigned hi = ~(unsigned)(sk >> 32), lo = ~(unsigned)sk;
    const unsigned mhi = wave_umin(hi);
    const unsigned mlo = wave_umin(hi == mhi ? lo : 0xffffffffu);
    return ~(((unsigned long long)mhi << 32) | mlo);
}
// upper bound of the tour-edge lengths Ef[1..n] for the pruned relocate scan, as an LDS slot in the order-preserving
// image: aliases an exchange slot of block_reduce_best that the best-improvement descent never touches
__device__ __forceinline__ unsigned long long *lmax_slot(Ctl *ctl) { return reinterpret_cast<unsigned long long *>(&ctl->red_d[1][7]); }
__device__ __forceinline__ void lmax_raise(Ctl *ctl, double v) {
    typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;
    __hip_atomic_fetch_max((lds_u64_t *)lmax_slot(ctl), sortable(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Nearest-neighbour lists of the pruned scans: one workgroup per instance, thread x selects the kNL nearest nodes of x by
// (D[x,y], y) ascending.  prune_ok[b] = every entry finite and |D| <= 1e6 (see above).
__global__ void neighbor_lists_kernel(const double *D, int n, uint8_t *nl_id, int32_t *prune_ok) {
    const int b = blockIdx.x;
    const double *Dg = D + (size_t)b * n * n;
    int bad = 0;
    for (int x = threadIdx.x; x < n; x += blockDim.x) {
        const double *row = Dg + (size_t)x * n;
        double last_d = -__builtin_inf(); int last_y = -1;
        for (int m = 0; m < kNL; ++m) {
            double best = __builtin_inf(); int by = -1;
            for (int y = 0; y < n; ++y) {
                if (y == x) continue;
                // the element of the LOWER triangle, D[max, min] -- the one the search kernel keeps in LDS and compares the
                // lists against: the order of a list is then exact for that image even if D is asymmetric by a few ulps
                const double v = y < x ? row[y] : Dg[(size_t)y * n + x];
                if (m == 0 && !(fabs(v) <= 1e6)) bad = 1;        // also catches NaN / inf
                if ((v > last_d || (v == last_d && y > last_y)) && (by < 0 || v < best)) { best = v; by = y; }
            }
            nl_id[((size_t)b * n + x) * kNL + m] = (uint8_t)(by < 0 ? (x == 0 ? 1 : 0) : by);      // (never x itself)
            last_d = best; last_y = by;
        }
    }
    bad = __syncthreads_or(bad);
    if (threadIdx.x == 0) prune_ok[b] = !bad;
}

// asym[b] = 1 iff D_b is not bitwise symmetric (compared as bit patterns: NaNs and signed zeros included) -- one pass over
// the matrices before a search on a symmetric store (1024 x TSP100: 82 MB, ~20 us)
__global__ void symmetry_kernel(const double *D, int n, int32_t *asym) {
    const int b = blockIdx.x;
    const unsigned long long *Dg = reinterpret_cast<const unsigned long long *>(D) + (size_t)b * n * n;
    int bad = 0;
    for (int q = threadIdx.x; q < n * n; q += blockDim.x) {
        const int a = q / n, c = q - a * n;
        if (a < c && Dg[q] != Dg[(size_t)c * n + a]) bad = 1;
    }
    bad = __syncthreads_or(bad);
    if (threadIdx.x == 0) asym[b] = bad;
}

