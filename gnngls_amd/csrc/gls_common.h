// gls_common.h -- part of gls_kernels.hip (one translation unit; included inside namespace gnngls, in this order:
// gls_common.h, gls_descent_scans.h, gls_perturbation.h).  Build switches, cycle stamps, storage policies (where the distance / penalty triangles live), move evaluation in the reference's operand order, selection keys, DPP wave reductions, the workgroup arg-min.
#pragma once


#ifndef GLS_PERTURB_PRIO
#define GLS_PERTURB_PRIO 3           // s_setprio of the wavefront that carries the perturbation phase
#endif
#ifndef GLS_LCAP_FACTOR
#define GLS_LCAP_FACTOR 3.0          // pruned relocate scan: tour edges longer than this many mean edge lengths are listed per scan
#endif
#ifndef GLS_TEAM_SCANS
#define GLS_TEAM_SCANS 4             // team form: one-to-all scans evaluated per round (4 = both endpoints, 2 = one endpoint, 1)
#endif
#ifndef GLS_NODE_LANES
#define GLS_NODE_LANES 0             // relocate descent scan: lanes own tour positions (0) or node ids (1: 21 % fewer LDS
                                     // bank-conflict cycles, 1 % FEWER iterations -- profiles/r03_experiments/README.md)
#endif
#ifndef GLS_WPS2
#define GLS_WPS2 1                   // 256-VGPR build of the one-slot kernel for batches of <= 2 single-wavefront workgroups per SIMD (TSP20 x 1000:
                                     // groups of 4 steps in the half-wave scans without scratch, +3 %; profiles/r04_experiments)
#endif
#ifndef GLS_PRUNE_MAX_WPS
#define GLS_PRUNE_MAX_WPS 6          // register budgets (waves per SIMD) whose instantiations carry the pruned descent scans: not the
                                     // 64-VGPR builds (batches of small instances: scratch 148 -> 100 B, +0.8 %; profiles/r04_experiments)
#endif
#ifndef GLS_TEAM_NODE_SUBST
#define GLS_TEAM_NODE_SUBST 1        // team form: known-count substitution decided by node compares (uniform part on the scalar unit)
#endif
#ifndef GLS_PEN_BUFFER
#define GLS_PEN_BUFFER 1             // compact store: penalty counters through a raw buffer descriptor (32-bit offsets)
#endif
#ifndef GLS_SKIP_DEAD_PASS
#define GLS_SKIP_DEAD_PASS 1         // pruned descent scans: a wavefront without rows in a pass skips it
#endif
#ifndef GLS_LEAN_UNROLL
#define GLS_LEAN_UNROLL 4            // evaluations per group in the lean descent scans (loads of a group issued up front)
#endif
#ifndef GLS_LEAN_UNROLL_RELOCATE
#define GLS_LEAN_UNROLL_RELOCATE 6   // the relocate scan: with one exec-masked test per group (group_may_improve) six steps per group win
#endif
constexpr int kWave = 64;
constexpr int kNoKey = INT_MAX;
constexpr int kGuidePassesMax = 4;   // register-cached guide values cover n <= 256

// Diagnostic build only (-DGLS_STAMPS): per-phase shader-cycle totals of the search kernel, written to
// a side buffer that nothing else reads.  The shipped library is built without it.
struct Stamps {
#ifdef GLS_STAMPS
    long long acc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // 20..23: quiet rows (reduced scans, their cycles, update cycles, active rows)
    long long t0 = 0;
    __device__ __forceinline__ void begin() { t0 = clock64(); }
    __device__ __forceinline__ void end(int i) { const long long n = clock64(); acc[i] += n - t0; t0 = n; }
    __device__ __forceinline__ void count(int i) { acc[i] += 1; }
#else
    __device__ __forceinline__ void begin() {}
    __device__ __forceinline__ void end(int) {}
    __device__ __forceinline__ void count(int) {}
#endif
};
#define STAMP_DECL Stamps st
#define STAMP_BEGIN() st.begin()
#define STAMP_END(i) st.end(i)
#define STAMP_COUNT(i) st.count(i)

// -DGLS_ISA_MARKS (scripts/isa_critical_path.py): comment lines in the disassembly that delimit the regions of a penalty step
#ifdef GLS_ISA_MARKS
#define ISA_MARK(name) asm volatile("; GLSMARK " name)
#else
#define ISA_MARK(name) do {} while (0)
#endif

__device__ __forceinline__ int make_key(int i, int j) { return (i << 16) | j; }

// [exact] np.isclose(0, delta): |delta| <= atol + rtol*|delta| with rtol=1e-5, atol=1e-8.
__device__ __forceinline__ bool close_to_zero(double delta) {
    double ad = fabs(delta);
    double r = 1e-5 * ad;      // one rounding
    double rhs = 1e-8 + r;     // second rounding (contraction is off)
    return ad <= rhs;
}

// ---------------------------------------------------------------------------------------------
// Storage policies
// ---------------------------------------------------------------------------------------------
// Packed lower triangle without diagonal (symmetric D only).  The diagonal is never read by a
// valid move evaluation for n >= 3 (all four/six endpoints are distinct nodes).
// PT = penalty element type in LDS: int32_t, or uint16_t (half the footprint -> one more resident
// workgroup per CU at n=100); a 16-bit counter that would pass 65535 aborts the instance with
// GNNGLS_STATUS_PENALTY_OVERFLOW_DEV and the host reruns it with 32-bit counters.
template <class PT>
struct TriStore {
    const double *d;   // LDS
    PT *p;             // LDS
    using pen_t = PT;
    using tour_t = int32_t;
    static constexpr bool kSymmetric = true;
    static constexpr bool kPenInLds = true;
    static constexpr int kWavesPerSimd = 6;      // 3 workgroups of 8 waves per CU
    static constexpr int kScanUnroll = 1;        // 80-VGPR budget: no room for batched evaluations
    __device__ __forceinline__ static int idx(int a, int b) {
        int hi = a > b ? a : b, lo = a > b ? b : a;
        return (__mul24(hi, hi - 1) >> 1) + lo;              // (nodes < 2^23: the full-rate 24-bit multiply)
    }
    __device__ __forceinline__ double dist(int a, int b) const { return d[idx(a, b)]; }
    __device__ __forceinline__ int pen(int a, int b) const { return (int)p[idx(a, b)]; }
    // index with the triangular row offsets precomputed: a2 = a(a-1)/2 (per lane), c2 = c(c-1)/2 (wave-uniform, SALU)
    __device__ __forceinline__ static int idx2(int a, int a2, int c, int c2) { return a > c ? a2 + c : c2 + a; }
    __device__ __forceinline__ double dist_at(int q) const { return d[q]; }
    __device__ __forceinline__ int pen_at(int q) const { return (int)p[q]; }
    // byte-offset forms (block form of the serial perturbation phase; 32-bit counters): off4 = 4 x packed index
    __device__ __forceinline__ double dist_at_byte(int off8) const { return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(d) + off8); }
    __device__ __forceinline__ int pen_at_byte(int off4) const { return (int)*reinterpret_cast<const PT *>(reinterpret_cast<const char *>(p) + off4); }
    __device__ __forceinline__ void pen_store_byte_if(unsigned long long lanes, int off4, int v) const {
        if ((lanes >> (threadIdx.x & 63)) & 1ull) *reinterpret_cast<PT *>(reinterpret_cast<char *>(p) + off4) = (PT)v;
    }
    int limit;         // largest representable count (65535 for 16-bit counters; lowered only by the test hook)
    // the caller already holds the current count (register-cached): store old + 1 without reading the counter back
    __device__ __forceinline__ bool pen_set(int a, int b, int old_count) const {     // true = counter overflow
        if (sizeof(PT) == 2 && old_count >= limit) return true;
        p[idx(a, b)] = (PT)(old_count + 1);
        return false;
    }
    __device__ __forceinline__ bool pen_inc(int a, int b) const {     // true = counter overflow
        const int q = idx(a, b);
        const PT v = p[q];
        if (sizeof(PT) == 2 && (int)v >= limit) return true;
        p[q] = (PT)(v + 1);
        return false;
    }
};

// Compact store: only the fp64 distance triangle is LDS-resident (n=100: 39.6 KB, with byte-sized tour
// arrays exactly 40 KiB per workgroup -> FOUR resident workgroups per CU); the penalty triangle lives in
// global memory as int32 (19.8 KB per TSP100 instance, L1/L2-resident).  It is written only by wavefront 0 of
// the owning workgroup and read only by that wavefront, with plain loads/stores: in-order within the wave
// through the CU's write-through L1, so no atomics or cache maintenance are needed.  (uint16 counters were
// 4 % faster but overflow within a 10 s run when an uninformative guide concentrates the penalties on few
// edges -- 800k penalty steps per instance -- and an overflow costs a whole rerun.)
struct TriDGlobalP {
    const double *d;   // LDS
    int32_t *p;        // global, packed triangle
#if GLS_PEN_BUFFER
    // the same triangle as a raw buffer: loads and stores take a 32-bit byte offset (buffer_load_dword ... offen) instead of
    // a 64-bit per-lane address -- no sign extension and 64-bit add per scattered counter load of the guided scans
    __amdgpu_buffer_rsrc_t prs;
#endif
    using pen_t = int32_t;
    using tour_t = uint8_t;                       // n <= 255
    static constexpr bool kSymmetric = true;
    static constexpr bool kPenInLds = false;
    static constexpr int kWavesPerSimd = 4;      // default register budget (128 VGPRs); the launcher also builds an 8-wave variant
    static constexpr int kScanUnroll = 1;        // measured: 2-deep batching costs more in spills than it hides (8.6k vs 10.0k)
    __device__ __forceinline__ static int idx(int a, int b) {
        int hi = a > b ? a : b, lo = a > b ? b : a;
        return (__mul24(hi, hi - 1) >> 1) + lo;              // (n <= 255: the full-rate 24-bit multiply)
    }
    __device__ __forceinline__ double dist(int a, int b) const { return d[idx(a, b)]; }
    // index with the triangular row offsets precomputed: a2 = a(a-1)/2 (per lane), c2 = c(c-1)/2 (wave-uniform, SALU)
    __device__ __forceinline__ static int idx2(int a, int a2, int c, int c2) { return a > c ? a2 + c : c2 + a; }
    __device__ __forceinline__ double dist_at(int q) const { return d[q]; }
#if GLS_PEN_BUFFER
    __device__ __forceinline__ void bind(int ntri) { prs = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, ntri * 4, 0x00020000); }
    __device__ __forceinline__ int pen_at(int q) const { return __builtin_amdgcn_raw_buffer_load_b32(prs, q << 2, 0, 0); }
    __device__ __forceinline__ void pen_store(int q, int v) const { __builtin_amdgcn_raw_buffer_store_b32(v, prs, q << 2, 0, 0); }
#else
    __device__ __forceinline__ void bind(int) {}
    __device__ __forceinline__ int pen_at(int q) const { return (int)p[q]; }
    __device__ __forceinline__ void pen_store(int q, int v) const { p[q] = v; }
#endif
    __device__ __forceinline__ int pen(int a, int b) const { return pen_at(idx(a, b)); }
    // byte-offset forms (block form of the serial perturbation phase): off4 = 4 x packed index
    __device__ __forceinline__ double dist_at_byte(int off8) const { return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(d) + off8); }
#if GLS_PEN_BUFFER
    __device__ __forceinline__ int pen_at_byte(int off4) const { return __builtin_amdgcn_raw_buffer_load_b32(prs, off4, 0, 0); }
    // the lanes of `lanes` store, the others aim past the end of the buffer: the range check drops their store -- no branch
    __device__ __forceinline__ void pen_store_byte_if(unsigned long long lanes, int off4, int v) const {
        int o;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(o) : "v"(0x7ffffffc), "v"(off4), "s"(lanes));
        __builtin_amdgcn_raw_buffer_store_b32(v, prs, o, 0, 0);
    }
#else
    __device__ __forceinline__ int pen_at_byte(int off4) const { return *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(p) + off4); }
    __device__ __forceinline__ void pen_store_byte_if(unsigned long long lanes, int off4, int v) const {
        if ((lanes >> (threadIdx.x & 63)) & 1ull) *reinterpret_cast<int32_t *>(reinterpret_cast<char *>(p) + off4) = v;
    }
#endif
    __device__ __forceinline__ bool pen_inc(int a, int b) const {
        const int q = idx(a, b);
        pen_store(q, pen_at(q) + 1);
        return false;
    }
    __device__ __forceinline__ bool pen_set(int a, int b, int old_count) const {   // no read-back: store only
        pen_store(idx(a, b), old_count + 1);
        return false;
    }
};

// The compact store as the TEAM form of the perturbation phase uses it (one workgroup per CU, all wavefronts scan at
// once): the counters are a full symmetric n x n matrix in global memory, not a packed triangle.  A scan reads
// P[u, t[j]] with u wave-uniform, so the 64 lanes of a load fall into the 4n bytes of row u -- a handful of cache lines
// instead of 64: sixteen wavefronts issuing scattered triangle loads at once were bound by the address path of the CU
// (~3.5k cycles per round waiting for the slowest wavefront, profiles/r03_team_*.log).  Both orientations of a pair are
// stored (two stores per penalty step, by the one lane that owns the edge).
struct TriDGlobalPF : TriDGlobalP {
    int n;
    // the packed-triangle accessors of the base (its buffer descriptor is not bound here) must not be reached through this store
    int pen_at(int) const = delete;
    void pen_store(int, int) const = delete;
    int pen_at_byte(int) const = delete;
    void pen_store_byte_if(unsigned long long, int, int) const = delete;
    __device__ __forceinline__ int pen(int a, int b) const { return p[a * n + b]; }
    // matrix cell r = row * n + column >= 0: base (scalar registers) + an unsigned 32-bit byte offset -- the load takes the
    // offset register as it is; an int index costs a sign extension and a 64-bit add per load
    __device__ __forceinline__ int cell(int r) const {
        return *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(p) + ((unsigned)r << 2));
    }
    __device__ __forceinline__ bool pen_inc(int a, int b) const { p[a * n + b] += 1; p[b * n + a] += 1; return false; }
    __device__ __forceinline__ bool pen_set(int a, int b, int old_count) const {
        p[a * n + b] = old_count + 1; p[b * n + a] = old_count + 1;
        return false;
    }
};
template <class S> struct PenRowMajor { static constexpr bool value = false; };
template <> struct PenRowMajor<TriDGlobalPF> { static constexpr bool value = true; };

// Full row-major matrices in global memory (any n, asymmetric D allowed: index order follows the
// reference exactly).  Used when the triangles do not fit in LDS and by the unit kernels.
struct GlobalStore {
    const double *d;
    int32_t *p;
    int n;
    static constexpr bool kSymmetric = false;
    __device__ __forceinline__ int idx(int a, int b) const { return a * n + b; }
    __device__ __forceinline__ int idx2(int a, int, int c, int) const { return a * n + c; }
    __device__ __forceinline__ double dist(int a, int b) const { return d[(size_t)a * n + b]; }
    __device__ __forceinline__ int pen(int a, int b) const { return p[(size_t)a * n + b]; }
    __device__ __forceinline__ double dist_at(int q) const { return d[q]; }
    __device__ __forceinline__ int pen_at(int q) const { return p[q]; }
    using pen_t = int32_t;
    using tour_t = int32_t;
    static constexpr bool kPenInLds = false;
    static constexpr int kWavesPerSimd = 4;
    static constexpr int kScanUnroll = 2;
    __device__ __forceinline__ bool pen_inc(int a, int b) const {
        p[(size_t)a * n + b] += 1;
        p[(size_t)b * n + a] += 1;
        return false;
    }
    __device__ __forceinline__ bool pen_set(int a, int b, int old_count) const {
        p[(size_t)a * n + b] = old_count + 1;
        p[(size_t)b * n + a] = old_count + 1;
        return false;
    }
};

template <class S>
struct PlainDist {
    const S &s;
    __device__ __forceinline__ double operator()(int a, int b) const { return s.dist(a, b); }
};

// ---------------------------------------------------------------------------------------------
// Move evaluation, reference operand order   [exact]
// ---------------------------------------------------------------------------------------------
template <class TT, class F>
__device__ __forceinline__ double two_opt_cost(const TT *t, const F &f, int i, int j) {
    if (i == j) return 0.0;
    if (j < i) { int x = i; i = j; j = x; }
    int a = t[i], b = t[i - 1], c = t[j], d = t[j - 1];
    double delta = f(a, c) + f(b, d);      // operators.py:25-28, left to right
    delta = delta - f(a, b);
    delta = delta - f(c, d);
    return delta;
}

template <class TT, class F>
__device__ __forceinline__ double relocate_cost(const TT *t, const F &f, int i, int j) {
    if (i == j) return 0.0;
    int a = t[i - 1], b = t[i], c = t[i + 1];
    int d, e;
    if (i < j) { d = t[j]; e = t[j + 1]; } else { d = t[j - 1]; e = t[j]; }
    double delta = -f(a, b);               // operators.py:97-102, left to right
    delta = delta - f(b, c);
    delta = delta + f(a, c);
    delta = delta - f(d, e);
    delta = delta + f(d, b);
    delta = delta + f(b, e);
    return delta;
}

// tour after a move, as a function of the old tour (operators.py:6-11, 76-80)
__device__ __forceinline__ int two_opt_src(int p, int i, int j) {   // requires i < j
    return (p >= i && p < j) ? (i + j - 1 - p) : p;
}
__device__ __forceinline__ int relocate_src(int p, int i, int j) {
    if (i < j) {
        if (p < i || p > j) return p;
        return p < j ? p + 1 : i;
    }
    if (p < j || p > i) return p;
    return p == j ? i : p - 1;
}
__device__ __forceinline__ int move_src(int op, int p, int i, int j) {
    if (op == 0) {
        int lo = i < j ? i : j, hi = i < j ? j : i;
        return two_opt_src(p, lo, hi);
    }
    return relocate_src(p, i, j);
}

// ---------------------------------------------------------------------------------------------
// Selection:  candidate (delta, key); "no candidate" = (0.0, kNoKey)
//   best improvement : lexicographic min of (delta, key)       == sequential strict-< scan
//   first improvement: min key among qualifying candidates     == first hit in enumeration order
// ---------------------------------------------------------------------------------------------
template <bool FI>
__device__ __forceinline__ bool better(double d1, int k1, double d2, int k2) {
    if (FI) return k1 < k2;
    return d1 < d2 || (d1 == d2 && k1 < k2);
}

template <bool FI>
__device__ __forceinline__ void consider(double delta, int key, double &bd, int &bk) {
    if (delta < 0.0 && better<FI>(delta, key, bd, bk) && !close_to_zero(delta)) { bd = delta; bk = key; }
}

typedef unsigned long long lanemask_t;
// lane-wise m ? a : b with the condition in a scalar register pair (v_cndmask_b32 e64: 4.3 cycles; on VCC the same select
// measures 8-17)
__device__ __forceinline__ int sel_b32(lanemask_t m, int a, int b) {
    int r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
__device__ __forceinline__ double sel_f64(lanemask_t m, double a, double b) {
    const long long ab = __double_as_longlong(a), bb = __double_as_longlong(b);
    const int lo = sel_b32(m, (int)ab, (int)bb), hi = sel_b32(m, (int)(ab >> 32), (int)(bb >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// ---- wavefront reductions on DPP (no LDS crossbar round trips) ---------------------------------
// Inclusive min-scan inside each row of 16 lanes (row_shr 1,2,4,8), then row_bcast15 / row_bcast31
// carry the row results upwards; lane 63 ends up with the minimum of all 64 lanes (the gfx9
// wave64 reduction sequence), read back with v_readlane into an SGPR.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_umin_step(unsigned x) {
    unsigned y = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)x, CTRL, ROW_MASK, 0xf, false);
    return y < x ? y : x;
}
__device__ __forceinline__ unsigned wave_umin(unsigned x) {
    x = dpp_umin_step<0x111, 0xf>(x);   // row_shr:1
    x = dpp_umin_step<0x112, 0xf>(x);   // row_shr:2
    x = dpp_umin_step<0x114, 0xf>(x);   // row_shr:4
    x = dpp_umin_step<0x118, 0xf>(x);   // row_shr:8
    x = dpp_umin_step<0x142, 0xa>(x);   // row_bcast:15 -> rows 1,3
    x = dpp_umin_step<0x143, 0xc>(x);   // row_bcast:31 -> rows 2,3
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}

// order-preserving map fp64 -> uint64 (a < b  <=>  key(a) < key(b), for non-NaN values)
__device__ __forceinline__ unsigned long long sortable(double v) {
    v = (v == 0.0) ? 0.0 : v;            // -0.0 and +0.0 compare equal: give them one key
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double unsortable(unsigned long long k) {
    unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

// lexicographic wave-wide min of (value, key): three 32-bit DPP reductions; result uniform in all lanes
__device__ __forceinline__ void wave_min_value_key(double &d, int &k) {
    const unsigned long long sk = sortable(d);
    const unsigned hi = (unsigned)(sk >> 32), lo = (unsigned)sk;
    const unsigned mhi = wave_umin(hi);
    // the high words of two candidates' deltas almost never tie: one lane left -> its low word and key by v_readlane,
    // the other two reductions (2 x 6 dependent DPP steps) only run on a tie
    const unsigned long long tie = __ballot(hi == mhi);
    if ((tie & (tie - 1)) == 0ull) {
        const int src = __ffsll((long long)tie) - 1;
        const unsigned mlo = (unsigned)__builtin_amdgcn_readlane((int)lo, src);
        d = unsortable(((unsigned long long)mhi << 32) | mlo);
        k = __builtin_amdgcn_readlane(k, src);
        return;
    }
    const unsigned mlo = wave_umin(hi == mhi ? lo : 0xffffffffu);
    const unsigned mk = wave_umin((hi == mhi && lo == mlo) ? (unsigned)k : 0x7fffffffu);
    d = unsortable(((unsigned long long)mhi << 32) | mlo);
    k = (int)mk;
}

template <bool FI>
__device__ __forceinline__ void wave_reduce_best(double &d, int &k) {
    if (FI) {
        // first improvement: smallest key among the candidates; its delta is fetched from the owning lane
        const unsigned mk = wave_umin((unsigned)k);
        const unsigned long long own = __ballot((unsigned)k == mk);
        const int src = __ffsll((long long)own) - 1;
        const long long bits = __double_as_longlong(d);
        const int lo = __builtin_amdgcn_readlane((int)bits, src);
        const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), src);
        d = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
        k = (int)mk;
    } else {
        wave_min_value_key(d, k);
    }
}

// arg-max with "first maximum wins": min over (-value order, position)
__device__ __forceinline__ void wave_argmax_first(double &v, int &pos) {
    // lanes without a candidate carry pos == kNoKey and must lose: give them the largest key
    unsigned long long sk = ~sortable(v);
    if (pos == kNoKey) sk = ~0ull;
    const unsigned hi = (unsigned)(sk >> 32), lo = (unsigned)sk;
    const unsigned mhi = wave_umin(hi);
    const unsigned long long tie = __ballot(hi == mhi);      // as in wave_min_value_key: usually one lane is left
    if ((tie & (tie - 1)) == 0ull) {
        const int src = __ffsll((long long)tie) - 1;
        const unsigned mlo = (unsigned)__builtin_amdgcn_readlane((int)lo, src);
        v = unsortable(~(((unsigned long long)mhi << 32) | mlo));
        pos = __builtin_amdgcn_readlane(pos, src);
        return;
    }
    const unsigned mlo = wave_umin(hi == mhi ? lo : 0xffffffffu);
    const unsigned mp = wave_umin((hi == mhi && lo == mlo) ? (unsigned)pos : 0x7fffffffu);
    v = unsortable(~(((unsigned long long)mhi << 32) | mlo));
    pos = (int)mp;
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// LDS control block shared by the workgroup.  red_d / red_k are the exchange slots of block_reduce_best; the
// best-improvement descent reuses the first 24 + 12 bytes as the three rotating (value, key) slots of block_reduce_best_lds.
struct Ctl {
    double red_d[2][8];
    int red_k[2][8];
    double cost;
    int flag;
    int pad;
};

template <bool FI>
__device__ __forceinline__ void block_reduce_best(Ctl *ctl, int &phase, int wave, int nwaves, int lane,
                                                  double &d, int &k) {
    wave_reduce_best<FI>(d, k);
    if (nwaves == 1) return;
    if (nwaves > 8) {        // 9..16 wavefronts: the upper ones hand their result to wavefront w - 8 first (8 exchange slots)
        if (wave >= 8 && lane == 0) { ctl->red_d[phase][wave - 8] = d; ctl->red_k[phase][wave - 8] = k; }
        __syncthreads();
        if (wave + 8 < nwaves) {
            const double od = ctl->red_d[phase][wave]; const int ok = ctl->red_k[phase][wave];
            if (better<FI>(od, ok, d, k)) { d = od; k = ok; }
        }
        phase ^= 1;
        nwaves = 8;
    }
    if (lane == 0 && wave < 8) { ctl->red_d[phase][wave] = d; ctl->red_k[phase][wave] = k; }
    __syncthreads();
    d = ctl->red_d[phase][0]; k = ctl->red_k[phase][0];
    for (int w = 1; w < nwaves; ++w) {
        double od = ctl->red_d[phase][w]; int ok = ctl->red_k[phase][w];
        if (better<FI>(od, ok, d, k)) { d = od; k = ok; }
    }
    phase ^= 1;
}

