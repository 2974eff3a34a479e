// gls_kernels.hip -- guided local search on MI355X (gfx950), hand-written HIP.
//
// Replaces (reference file:line, /root/reference/gnngls/...):
//   operators.py:6-147     two_opt / relocate move evaluation, a2a / o2a scans, move application
//   algorithms.py:111-132  local_search
//   algorithms.py:135-195  guided_local_search
//   algorithms.py:9-18     nearest_neighbor            __init__.py:17-21  tour_cost
//
// Design (MI355X-first):
//   * one persistent workgroup per TSP instance; the instance never leaves the CU: the fp64
//     distance matrix lives in LDS as a packed lower triangle (n=100: 39.6 KB), the tour is
//     ping-ponged between two LDS arrays, and there is no host round trip per move.  Where the
//     penalty counters live is a storage policy (TriStore: LDS triangle, 2-3 workgroups per CU;
//     TriDGlobalP: global memory behind the L1, exactly 40 KiB of LDS -> 4 workgroups per CU =
//     1024 resident TSP100 instances; GlobalStore: everything in HBM/L2 for any n);
//   * an a2a scan gives one tour row (fixed i) to a wavefront and the j axis to its 64 lanes, so
//     t[i], t[i-1] and the row constants are wave-uniform, tour reads are conflict-free and each
//     evaluation costs two random LDS reads (the other terms are the per-position edge lengths
//     Ef[], rebuilt in O(n) after every move);
//     -- the generic form; the best-improvement descent (the reference's default) runs on specialised scans with the
//     same deltas and keys: lean scans (a tour row per lane, the other index wave-uniform), on both 32-lane halves of the
//     wavefront for single-wavefront workgroups (n <= 33), and from n = 80 (2-opt) / n = 128 (relocate) pruned scans that
//     only evaluate the moves that can qualify (32-nearest-neighbour lists, rows owned by nodes: exact, see there);
//   * best-improvement selection is an arg-min on the key (delta, i, j): identical to the
//     reference's sequential strict-< scan (first minimum in enumeration order wins); inside a
//     wavefront it is three 32-bit DPP min-reductions, no LDS-crossbar shuffles;
//   * the perturbation phase (O(n) work per step, long serial chain) runs in wavefront 0 only, so
//     it needs no workgroup barriers; the other wavefronts park on one barrier; utilities of the
//     tour edges are cached in registers, guided evaluations issue all their loads up front
//     (16-wave workgroups that own their CU -- TSP200 -- run it on ALL wavefronts instead: the four
//     one-to-all scans of a penalty step at once, consumed in the reference's order: team_perturbation);
//   * all floating point is fp64 with contraction OFF: the guided matrix `D + k*P`
//     (algorithms.py:164) must round twice, np.isclose (operators.py:42) is evaluated literally.
//
// Bit-exactness notes are marked [exact].
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>

#include <atomic>
#include <type_traits>

#include "gls_kernels.h"

#pragma clang fp contract(off)

namespace gnngls {

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
constexpr int kWave = 64;
constexpr int kNoKey = INT_MAX;
constexpr int kGuidePassesMax = 4;   // register-cached guide values cover n <= 256

// Diagnostic build only (-DGLS_STAMPS): per-phase shader-cycle totals of the search kernel, written to
// a side buffer that nothing else reads.  The shipped library is built without it.
struct Stamps {
#ifdef GLS_STAMPS
    long long acc[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
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
template <int SL>
struct LaneTour {      // positions lane, lane + 64, ... (SL slots) of the tour in registers
    int t[SL];
};
template <int SL, class TT>
__device__ __forceinline__ LaneTour<SL> load_lane_tour(const TT *t, int n, int lane) {
    LaneTour<SL> L;
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

// pos != nullptr: lane l of row block rb owns NODE b = 1 + 64 rb + l (wherever it sits in the tour: i = pos[b]) instead of
// tour POSITION 1 + 64 rb + l.  The random read of a step is D[b, e] with e wave-uniform: with consecutive node ids on the
// lanes the half of the lanes with b < e reads 64 consecutive doubles of row e and the other half a fixed quadratic
// pattern (b(b-1)/2 + e), instead of 64 arbitrary rows / columns: simulated 2.5 instead of 4.9 bank passes per
// ds_read_b64 at n = 100.  Keys (i, j) and deltas are the same set; within a lane they still ascend with k.
template <int SL, class S, class TT>
__device__ __forceinline__ void scan_relocate_a2a_lean(const S &s, const TT *t, const double *Ef, int n,
                                                       int wave, int nwaves, int lane, double &bd, int &bk,
                                                       const uint8_t *pos = nullptr) {
    const int RW = (n - 1 + kWave - 1) / kWave;              // row blocks of 64 rows
    const int per_rb = nwaves / RW;                          // waves sharing a row block, each a contiguous k range
    const int rb = __builtin_amdgcn_readfirstlane(wave / (per_rb > 0 ? per_rb : 1));
    const int part = __builtin_amdgcn_readfirstlane(wave - rb * per_rb);
    if (per_rb == 0 || rb >= RW) return;                     // callers guarantee nwaves >= RW; surplus waves idle
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
    // U steps at a time: all wave-uniform operands, addresses and the U random distance reads are issued before the first
    // dependent add, so the LDS round trips of a group overlap (a step alone is a ~280-cycle dependent chain)
    auto group = [&](int k, int te, auto ucount, auto fast_addr) {        // te: the register slot holding positions k+1 .. k+U
        constexpr int U = decltype(ucount)::value;
        constexpr bool FA = decltype(fast_addr)::value;      // every t[k+1] of the group is a node >= 1
        double ve[U], de[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = bcast_int(te, k + u + 1);          // v_readlane uses the lane index modulo 64
            const int e2 = (e * (e - 1)) >> 1;               // wave-uniform: scalar ALU
            de[u] = Ef[k + u + 1];                           // D[t[k], t[k+1]]: wave-uniform address, one broadcast LDS read
            if constexpr (FA) ve[u] = lds_read_f64(tri_addr_max(bx, b8, dbase + 4 * e * (e - 1), 8 * e));   // 8 e(e-1)/2, no shift pair
            else ve[u] = s.dist_at(s.idx2(b, b2, e, e2));    // D[b, t[k+1]]
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = k + u;
            double delta = base - de[u];                     // -D[d,e]          (operators.py:100-102, left to right)
            delta = delta + vd;                              // +D[d,b]
            delta = delta + ve[u];                           // +D[b,e]
            vd = ve[u];
            if (delta < bd) {
                rare_path();
                // valid targets: k <= i-3 (j = k+1 < i-1) or k >= i+1 (j = k); k in {i-2, i-1, i} <=> (unsigned)(k - i + 2) <= 2
                if ((unsigned)(kk - i + 2) > 2u && !close_to_zero(delta)) { bd = delta; bk = make_key(i, kk < i ? kk + 1 : kk); }
            }
        }
    };
    using UN = std::integral_constant<int, GLS_LEAN_UNROLL>;
    using U1 = std::integral_constant<int, 1>;
    using FAST = std::integral_constant<bool, true>;
    using EXACT = std::integral_constant<bool, false>;
    const int k1f = k1 == n ? n - 1 : k1;                    // step k = n-1 meets t[n] = node 0: exact index, after the loops
#pragma unroll
    for (int q = 0; q < SL; ++q) {                           // slot q holds positions 64q .. 64q+63, i.e. k + 1 of k in [64q-1, 64q+62]
        const int lo = q * kWave - 1, hi = q * kWave + kWave - 1;
        int k = k0 > lo ? k0 : lo;
        const int ke = k1f < hi ? k1f : hi;
        for (; k + GLS_LEAN_UNROLL <= ke; k += GLS_LEAN_UNROLL) group(k, L.t[q], UN{}, FAST{});
        for (; k < ke; ++k) group(k, L.t[q], U1{}, FAST{});
    }
    if (k1f != k1) {                                         // keys ascend with k within a lane: the last step stays last
#pragma unroll
        for (int q = 0; q < SL; ++q)                         // the slot that holds position n
            if (n / kWave == q) group(n - 1, L.t[q], U1{}, EXACT{});
    }
}

template <int SL, class S, class TT>
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
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double delta = vac[u] + vbd[u];                  // operators.py:25-28, left to right
            delta = delta - eab;
            delta = delta - ecd[u];
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
    for (int task0 = 0; task0 < tasks; task0 += nthr, ++it) {      // wave-uniform trip count; a wavefront's tasks are whole rows
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

template <bool CNT, class S, class TT>
__device__ __forceinline__ void scan_relocate_a2a_pruned(const S &s, const TT *t, const TT *pos, const double *Ef,
                                                         const NlWords &nlw, int n, double Lcap,
                                                         const int *longk, int nlong,
                                                         int tid, int nthr, int lane, double &bd, int &bk, int &xe,
                                                         long long *dbg = nullptr) {
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
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool lk = live && m + 8 * u < nlong;
            const int k = lk ? longk[m + 8 * u] : p;             // (k = p is never a valid target)
            const bool ev = (unsigned)(k - p + 2) > 2u;
            if constexpr (CNT) xe += __popcll(__ballot(ev));
            if (ev) {
                double delta = base - Ef[k + 1];                 // operators.py:100-102, left to right
                delta = delta + s.dist(t[k], b);                 // +D[d,b]
                delta = delta + s.dist(b, t[k + 1]);             // +D[b,e]
                consider<false>(delta, make_key(p, k < p ? k + 1 : k), bd, bk);
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
                const bool c1 = act && (unsigned)(k1 - p + 2) > 2u && two_d < e1 - base;
                const bool c2 = act && (unsigned)(k2 - p + 2) > 2u && two_d < e2 - base;
                if constexpr (CNT) xe += __popcll(__ballot(c1)) + __popcll(__ballot(c2));
                if (c1) {
                    double delta = base - e1;                    // operators.py:100-102, left to right
                    delta = delta + d;                           // +D[d,b]
                    delta = delta + s.dist(b, t[k1 + 1]);        // +D[b,e]
                    consider<false>(delta, make_key(p, k1 < p ? k1 + 1 : k1), bd, bk);
                }
                if (c2) {
                    double delta = base - e2;
                    delta = delta + s.dist(t[k2], b);            // +D[d,b]
                    delta = delta + d;                           // +D[b,e]
                    consider<false>(delta, make_key(p, k2 < p ? k2 + 1 : k2), bd, bk);
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
            for (int j = 1 + lane; j <= n - 1; j += kWave) {
                if (j == pr || pr - j == 1) continue;
                consider<false>(relocate_cost(t, f, pr, j), make_key(pr, j), bd, bk);
            }
        }
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
// The persistent GLS kernel
// ---------------------------------------------------------------------------------------------
// Search trace (algorithms.py:127-130,180-183).  TR=false (trace_cap == 0, the throughput path) keeps only the
// accepted-move counter: no pointers, no clock, and the per-move tour_cost is deferred (see header).
template <bool TR>
struct Trace {
    double *cost; float *time; int cap; int len; long long t0;
    __device__ __forceinline__ void push(double c) {
        if (len < cap) {
            cost[len] = c;
            if (time) time[len] = (float)((double)(wall_clock64() - t0) * 1e-8);
        }
        len++;
    }
};
template <>
struct Trace<false> {
    int len;
    __device__ __forceinline__ void push(double) { len++; }
};

// neighbour lists of the pruned descent scans for this instance (on = false: the full scans run); ppos = node -> position
struct PruneCtx {
    NlWords nlw; bool on;
};

template <class S, bool FI, int GP, bool CNT, int WPS, class TT, class TRC>
__device__ __forceinline__ void local_search_dev(const S &s, TT *&t, TT *&t2, double *Ef, double *Eb, int n,
                                 Ctl *ctl, int &phase, double &cur_cost, TRC &tr, long long &evals, long long &xe, Stamps &st,
                                 TT *ppos, const PruneCtx &pc) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & (kWave - 1), wave = tid >> 6, nwaves = nthr >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // pruned descent scans (given neighbour lists): the 2-opt scan from n = 80 up, the relocate scan from n = 128 up (the
    // 4-slot instantiations) -- where each was measured faster (profiles/r03_experiments/README.md)
    constexpr bool kCanPrune = !FI && S::kSymmetric && WPS <= GLS_PRUNE_MAX_WPS;
    // steps per group of the half-wave scans: 4 on the 256-VGPR build (no scratch there), else GLS_HALF_UNROLL
    constexpr int kHalfUnroll = WPS <= 2 ? 4 : GLS_HALF_UNROLL;
    constexpr bool kPruneRelocate = kCanPrune && GP == 4;
    const bool prune = kCanPrune && pc.on;
    double Lmax = 0.0;
    if (prune) {
        if (tid == 0) *lmax_slot(ctl) = 0ull;                        // below the image of every double
        __syncthreads();
    }
    // node -> position table of the node-indexed relocate scan: n <= 104 bytes in the exchange slots of Ctl that the
    // best-improvement descent never touches (its LDS-atomic arg-min uses the first 24 bytes of red_d and 12 of red_k;
    // the compact store's 40 KiB at n = 100 have no other byte to spare)
    constexpr int kPosBytes = (int)sizeof(ctl->red_d) - 3 * (int)sizeof(double);
    uint8_t *pos = (!FI && S::kSymmetric && GLS_NODE_LANES && n <= kPosBytes && n <= GP * kWave - 1 &&
                    nwaves >= (n - 1 + kWave - 1) / kWave)
                       ? reinterpret_cast<uint8_t *>(&ctl->red_d[0][3]) : nullptr;
    build_edges(s, t, Ef, Eb, n, tid, nthr);
    if (pos) for (int p = tid; p < n; p += nthr) pos[t[p]] = (uint8_t)p;
    if constexpr (kCanPrune) {
        if (prune) {         // node -> position (the depot keeps position 0) and the maximum tour-edge length
            unsigned long long mine = 0ull;
            for (int p = tid; p <= n; p += nthr) {
                if (p < n) ppos[t[p]] = (TT)p;
                if (p >= 1) { const unsigned long long sk = sortable(s.dist(t[p - 1], t[p])); mine = sk > mine ? sk : mine; }
            }
            mine = wave_max_sortable(mine);
            if (lane == 0 && mine != 0ull) {
                typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;
                __hip_atomic_fetch_max((lds_u64_t *)lmax_slot(ctl), mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    if (prune) Lmax = unsortable(*lmax_slot(ctl));
    // pruned relocate scan: tour edges longer than Lcap (three mean edge lengths of the tour the descent starts from) are
    // listed per scan (at most kLongCap, else that scan runs unpruned) in exchange slots the descent does not use
    constexpr int kLongCap = 16;
    const double Lcap = GLS_LCAP_FACTOR * cur_cost / (double)n;
    int *longk = reinterpret_cast<int *>(&ctl->red_d[0][3]);     // 16 ints: bytes 24 .. 87 of red_d
    int *nlong_slot = &ctl->red_k[1][0];
    (void)Lmax;
    bool improved = true;
    while (improved) {                                               // algorithms.py:116
        improved = false;
#pragma unroll 1
        for (int op = 0; op < 2; ++op) {                             // algorithms.py:119
            double bd = 0.0; int bk = kNoKey;
            bool lean = false, pruned_scan = false;
            int xs = 0;          // evaluations this wavefront executes in a pruned scan (scalar; booked once per scan below)
            if constexpr (kCanPrune) {
                int nlong = 0;
                if (kPruneRelocate && prune && op == 1) {
                    if (tid == 0) *nlong_slot = 0;
                    __syncthreads();
                    for (int q = 1 + tid; q <= n; q += nthr)
                        if (Ef[q] > Lcap) {
                            typedef __attribute__((address_space(3))) int lds_i32_t;
                            const int slot = __hip_atomic_fetch_add((lds_i32_t *)nlong_slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (slot < kLongCap) longk[slot] = q - 1;            // target edge k = (t[k], t[k+1]), Ef[k+1] its length
                        }
                    __syncthreads();
                    nlong = *nlong_slot;
                }
#ifdef GLS_STAMPS
                if (prune && op == 1 && nlong > kLongCap) st.acc[15] += 1;      // relocate scans that fell back to the full scan
#endif
                if (prune && (op == 0 || (kPruneRelocate && nlong <= kLongCap))) {
#ifdef GLS_STAMPS
                    long long *dbg = &st.acc[16];            // overflow rows (2-opt, relocate), wave-passes (2-opt, relocate)
#else
                    long long *dbg = nullptr;
#endif
                    if (op == 0) scan_two_opt_a2a_pruned<CNT, S, TT>(s, t, ppos, Ef, pc.nlw, n, tid, nthr, lane, bd, bk, xs, dbg);
                    else if constexpr (kPruneRelocate)
                        scan_relocate_a2a_pruned<CNT, S, TT>(s, t, ppos, Ef, pc.nlw, n, Lcap, longk, nlong, tid, nthr, lane, bd, bk, xs, dbg);
                    lean = true; pruned_scan = true;
                }
            }
            // measured (outer iterations per instance): TSP50 7.2k -> 8.2k, TSP100 9.9k -> 10.4k, TSP200 3.8k -> 3.6k;
            // software-pipelining the uniform operands one step ahead costs registers: 9.6k at TSP100
            if constexpr (!FI && S::kSymmetric) {
                // positions 0..n fit 2 (n <= 127) or 4 (n <= 255) register slots per lane; every block of 64 rows needs
                // a wavefront of its own
                // (GP = register slots of the perturbation phase = the same 2 / 4, chosen by the launcher from n)
                if (GLS_HALF_SCANS && GP == 1 && !lean && nwaves == 1 && n >= kHalfScanMinNodes && n <= kHalfScanMaxNodes) {
                    if (op == 0) scan_two_opt_a2a_lean_half<kHalfUnroll, S, TT>(s, t, Eb, n, lane, bd, bk);
                    else         scan_relocate_a2a_lean_half<kHalfUnroll, S, TT>(s, t, Ef, n, lane, bd, bk);
                    lean = true;
                }
                if (!lean && nwaves >= (n - 1 + kWave - 1) / kWave && n <= GP * kWave - 1) {
                    if (op == 0) scan_two_opt_a2a_lean<GP, S, TT>(s, t, Eb, n, wave, nwaves, lane, bd, bk);
                    else         scan_relocate_a2a_lean<GP, S, TT>(s, t, Ef, n, wave, nwaves, lane, bd, bk, pos);
                    lean = true;
                }
            }
            if (lean) {
            } else if (FI && S::kScanUnroll == 1 && S::kSymmetric && n - 1 <= 2 * kWave) {
                if (op == 0) scan_two_opt_a2a_rowlane<S, FI, TT>(s, t, Eb, n, wave, nwaves, lane, bd, bk);
                else         scan_relocate_a2a_rowlane<S, FI, TT>(s, t, Ef, n, wave, nwaves, lane, bd, bk);
            } else {
                if (op == 0) scan_two_opt_a2a<S, FI, TT, S::kScanUnroll>(s, t, Eb, n, wave, nwaves, lane, bd, bk);
                else         scan_relocate_a2a<S, FI, TT, S::kScanUnroll>(s, t, Ef, n, wave, nwaves, lane, bd, bk);
            }
#ifdef GLS_STAMPS
            if (!kPruneRelocate && op == 1) st.acc[15] += clock64() - st.t0;       // relocate's share of the scan cycles (GP = 2 builds)
#endif
            STAMP_END(8);    // a2a scan (this wave's share)
            if (!FI && nwaves > 1) block_reduce_best_lds(ctl, phase, tid, bd, bk);
            else block_reduce_best<FI>(ctl, phase, wave, nwaves, lane, bd, bk);
            STAMP_END(9);    // wave + workgroup arg-min (includes waiting for the slowest wave)
            STAMP_COUNT(11);
            if (tid == 0) evals += (op == 0) ? (long long)(n - 2) * (n - 3) / 2 : (long long)(n - 2) * (n - 2);
            // executed count = evals + sum over the wavefronts of xe: a pruned scan booked the candidates it evaluated, and
            // wavefront 0 takes the scan's reference count back (uniform branch: xe stays in scalar registers)
            if (CNT && pruned_scan)
                xe += xs - (wave_u == 0 ? ((op == 0) ? (n - 2) * (n - 3) / 2 : (n - 2) * (n - 2)) : 0);
            if (bk != kNoKey) {                                      // delta < 0 (algorithms.py:122)
                improved = true;
                cur_cost += bd;                                      // algorithms.py:124
                apply_move(s, t, t2, Ef, Eb, n, op, bk >> 16, bk & 0xffff, tid, nthr, true, pos, prune ? ppos : nullptr, ctl, Lmax);
                TT *x = t; t = t2; t2 = x;
                if (tid == 0) tr.push(cur_cost);
                __syncthreads();
                if (prune) Lmax = unsortable(*lmax_slot(ctl));
                STAMP_END(10);   // move application + barrier
            }
        }
    }
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
// -DGLS_ISA_MARKS (scripts/isa_critical_path.py): comment lines in the disassembly that delimit the regions of a penalty step
#ifdef GLS_ISA_MARKS
#define ISA_MARK(name) asm volatile("; GLSMARK " name)
#else
#define ISA_MARK(name) do {} while (0)
#endif
#ifndef GLS_EDGE_PERTURB
#define GLS_EDGE_PERTURB 1           // 0: the scan-by-scan serial form everywhere (A/B builds)
#endif

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
#pragma unroll
    for (int q = 0; q < GP; ++q) {
        const int u = E.u[q], v = E.v[q];
        E.gq[q] = guide[(unsigned)(u * n + v)];
        E.ur[q] = 2 * __mul24(u, u - 1); E.vr[q] = 2 * __mul24(v, v - 1);
        const int off = sel_b32(__builtin_amdgcn_ballot_w64(v > u), E.vr[q] + 4 * u, E.ur[q] + 4 * v);
        E.pq[q] = s.pen_at_byte(off);
        E.de[q] = s.dist_at_byte(2 * off);
    }
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
        const int u = told[move_src_flat(mm, pc)], v = told[move_src_flat(mm, pc + 1)];
        E.u[q] = u; E.v[q] = v;
        tnew[pc] = (TT)u;
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

// One guided one-to-all scan at tour index i on the tour held by E: RELOC = false two_opt_o2a, true relocate_o2a.
// ok[q] = the lanes of slot q with a valid move, delta[q] their deltas; returns the lanes (any slot) with a negative one.
template <bool RELOC, class S, int GP>
__device__ __forceinline__ lanemask_t eval_scan(const S &s, const double k, const TourEdges<GP> &E, const int lane, const int i,
                                                const lanemask_t (&nm)[GP], double (&delta)[GP], lanemask_t (&ok)[GP]) {
    ISA_MARK("scan_issue");
    const int na = edge_bcast<GP>(E.u, i), nb = edge_bcast<GP>(E.u, i - 1);       // a = t[i], b = t[i-1]
    const int nar = 2 * na * (na - 1), nbr = 2 * nb * (nb - 1);
    PairLoad x0[GP], x1[GP], xac;
    if (RELOC) {                                             // G[t[i-1], t[i+1]]: a wave-uniform pair
        const int nc = edge_bcast<GP>(E.v, i);
        xac = pair_issue(s, uniform_pair_offset(nb, nc));
    }
#pragma unroll
    for (int q = 0; q < GP; ++q) {
        x0[q] = pair_issue(s, pair_offset(na, nar, E.v[q], E.vr[q]));                           // G[t[i], t[k+1]]
        x1[q] = RELOC ? pair_issue(s, pair_offset(na, nar, E.u[q], E.ur[q]))                    // G[t[i], t[k]]
                      : pair_issue(s, pair_offset(nb, nbr, E.u[q], E.ur[q]));                   // G[t[i-1], t[k]]
    }
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
    for (int q = 0; q < GP; ++q) { pin(x0[q]); pin(x1[q]); asm volatile("" : "+v"(ge[q])); if (!RELOC) asm volatile("" : "+v"(s1[q]), "+v"(s2[q])); }
    double base = 0.0;
    if (RELOC) {
        pin(xac);
        base = -gab;                                         // operators.py:97-99, left to right
        base = base - gbc;
        base = base + guided(k, xac);
    }
    lanemask_t neg = 0ull;
#pragma unroll
    for (int q = 0; q < GP; ++q) {
        const double gav = guided(k, x0[q]), gxu = guided(k, x1[q]);
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
        int i = bp;                                          // algorithms.py:169: the edge was read at positions bp, bp + 1
        bool moved_this_step = false;
#pragma unroll 1
        for (int sc = eu == 0 ? 2 : 0; sc < (ev == 0 ? 2 : 4); ++sc) {      // scan = 2 endpoint + operator; algorithms.py:167-171
            ISA_MARK("scan_loop_head");
            if (sc == 2) {                                   // endpoint 1: cur_tour.index(ev), searched only after a move
                i = bp + 1;
                if (moved_this_step) {
#pragma unroll
                    for (int q = GP - 1; q >= 0; --q) {
                        const lanemask_t m = nm0[q] & __builtin_amdgcn_ballot_w64(E.u[q] == ev);
                        if (m) i = q * kWave + __ffsll((long long)m) - 1;
                    }
                }
            }
            double delta[GP];
            lanemask_t ok[GP];
            const bool reloc = (sc & 1) != 0;
            const lanemask_t neg = reloc ? eval_scan<true, S, GP>(s, k, E, lane, i, nm0, delta, ok)
                                         : eval_scan<false, S, GP>(s, k, E, lane, i, nm2, delta, ok);
            ISA_MARK("accept_fast");
            if (reloc) scans_re += 1; else scans_to += 1;
            STAMP_END(1);
            if (neg == 0ull) continue;                       // no negative delta: no candidate (most scans)
            ISA_MARK("accept_slow");
            // np.isclose evaluated literally; the keys of a lane ascend with its slots, so a strict < keeps the lane's first
            // minimum (operators.py:65,118)
            double bd = 0.0; int bk = kNoKey;
#pragma unroll
            for (int q = 0; q < GP; ++q) {
                const double d = delta[q];
                const int kk = lane + q * kWave;
                const int key = reloc ? (kk >= i + 1 ? kk : kk + 1) : kk + 1;
                const lanemask_t take = ok[q] & __builtin_amdgcn_ballot_w64(d < bd) & ~__builtin_amdgcn_ballot_w64(close_to_zero(d));
                bd = sel_f64(take, d, bd);
                bk = sel_b32(take, key, bk);
            }
            if (__builtin_amdgcn_ballot_w64(bk != kNoKey) == 0ull) { STAMP_END(2); continue; }
            wave_reduce_best<false>(bd, bk);
            bk = __builtin_amdgcn_readfirstlane(bk);
            STAMP_END(2);
            ISA_MARK("move");
            edges_move(s, E, t, t2, guide, n, sc & 1, i, bk, lane);          // algorithms.py:175-177
            { TT *x = t; t = t2; t2 = x; }
            any_moved = true; moved_this_step = true;
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
        }
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

// launch bounds: 8-wave workgroups, 6 waves per SIMD for the LDS-resident variants (3 workgroups per
// CU at n=100 need <= 80 VGPRs), 4 for the global-memory fallback.
// WPS = resident wavefronts per SIMD the kernel is compiled for = its register budget (512 / WPS VGPRs).  The compact store
// exists twice: WPS 4 (128 VGPRs, no spills: the instantiation a full TSP100 device load runs on, four 4-wave workgroups
// per CU) and WPS 8 (64 VGPRs, ~100 B of scratch) for batches of small instances that need more than 16 waves per CU.
// TEAM: the perturbation phase runs on all wavefronts (team_perturbation above) -- for workgroups that own their CU.
// CNT: also count the delta evaluations the pruned descent scans execute (GlsArgs::evals_exec; measurement builds, see there).
template <class S, bool FI, int GP, bool TR, int WPS, bool TEAM = false, bool CNT = false>
__global__ __launch_bounds__(WPS <= 2 ? 256 : WPS <= 4 ? 1024 : 512, WPS) void gls_kernel(GlsArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x;
    const int n = A.n;
    const int tid = threadIdx.x, nthr = blockDim.x;
    // the edge form of the serial perturbation phase: best improvement, 32-bit counters, the 128-VGPR (and wider) builds
    constexpr bool kEdgeForm = GLS_EDGE_PERTURB && S::kSymmetric && !FI && !TEAM && sizeof(typename S::pen_t) == 4 && WPS <= 4;
    // wave-uniform by construction; as a SCALAR it keeps `if (wave == 0)` -- the serial perturbation phase -- a scalar branch and
    // what the phase modifies (tour pointers, cost, counters) out of the exec-masked phis of a divergent one.  Measured
    // (profiles/r05_experiments/): +1 % for the team form, needed by the edge form, -4 % for the scan-by-scan form (whose code
    // was tuned around the vector form of `wave`), which therefore keeps it
    const int lane = tid & (kWave - 1);
    const int wave = (TEAM || kEdgeForm) ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
    const size_t nn = (size_t)n * n;
    const double *Dg = A.D + (size_t)b * nn;

    if constexpr (S::kSymmetric) {
        // the symmetric stores keep D[max, min] only: an instance whose matrix is not bitwise symmetric (symmetry_kernel) is NOT
        // searched -- it would be a different search than the reference's (operators.py reads D[a,b] as indexed) -- but handed
        // back untouched with GNNGLS_STATUS_ASYMMETRIC; the caller reruns it on the global-memory store (gnngls_amd.ops does)
        if (A.asym && A.asym[b]) {
            for (int p = tid; p <= n; p += nthr) A.best_tour[(size_t)b * (n + 1) + p] = A.init_tour[(size_t)b * (n + 1) + p];
            if (tid == 0) {
                A.best_cost[b] = A.init_cost[b];
                if (A.outer_iters) A.outer_iters[b] = 0;
                if (A.trace_len) A.trace_len[b] = 0;
                if (A.evals) A.evals[b] = 0;
                if (A.imp_len) A.imp_len[b] = 0;
                if (A.status) A.status[b] = GNNGLS_STATUS_ASYMMETRIC_DEV;
            }
            if (A.penalty_out) for (size_t q = tid; q < nn; q += nthr) A.penalty_out[(size_t)b * nn + q] = 0;
            return;
        }
    }
    // ---- LDS carve (all offsets multiples of 16) ----
    size_t off = 0;
    Ctl *ctl = reinterpret_cast<Ctl *>(smem + off);            off += (sizeof(Ctl) + 15) & ~size_t(15);
    TeamCtl *tc = nullptr;
    if constexpr (TEAM) { tc = reinterpret_cast<TeamCtl *>(smem + off); off += (sizeof(TeamCtl) + 15) & ~size_t(15); }
    double *Ef = reinterpret_cast<double *>(smem + off);       off += ((size_t)(n + 2) * 8 + 15) & ~size_t(15);
    double *Eb = Ef;
    if (!S::kSymmetric) { Eb = reinterpret_cast<double *>(smem + off); off += ((size_t)(n + 2) * 8 + 15) & ~size_t(15); }
    using TT = typename S::tour_t;
    TT *t = reinterpret_cast<TT *>(smem + off);                off += ((size_t)(n + 1) * sizeof(TT) + 15) & ~size_t(15);
    TT *t2 = reinterpret_cast<TT *>(smem + off);               off += ((size_t)(n + 1) * sizeof(TT) + 15) & ~size_t(15);
    // the best tour lives in the output array (written whenever the best improves: a few dozen times per run); its former
    // LDS slot holds the node -> position table of the pruned descent scans, so those cost no LDS (the compact store's
    // 40 KiB at n = 100 have no spare byte)
    int32_t *bt = A.best_tour + (size_t)b * (n + 1);
    TT *ppos = reinterpret_cast<TT *>(smem + off);             off += ((size_t)(n + 1) * sizeof(TT) + 15) & ~size_t(15);

    S s;
    if constexpr (S::kSymmetric) {
        const int ntri = n * (n - 1) / 2;
        double *dtri = reinterpret_cast<double *>(smem + off);   off += ((size_t)ntri * 8 + 15) & ~size_t(15);
        // row a of the lower triangle is contiguous in both the source row and the packed image
        for (int a = 1 + wave; a < n; a += (nthr >> 6)) {
            const double *src = Dg + (size_t)a * n;
            double *dst = dtri + ((a * (a - 1)) >> 1);
            for (int c = lane; c < a; c += kWave) dst[c] = src[c];
        }
        s.d = dtri;
        if constexpr (S::kPenInLds) {
            using PT = typename S::pen_t;
            PT *ptri = reinterpret_cast<PT *>(smem + off);
            for (int q = tid; q < ntri; q += nthr) ptri[q] = (PT)0;          // algorithms.py:138
            s.p = ptri; s.limit = A.pen16_limit;
        } else if constexpr (PenRowMajor<S>::value) {
            s.p = A.pen_ws + (size_t)b * nn; s.n = n;                         // full matrix, zeroed by the host
        } else {
            s.p = A.pen_ws + (size_t)b * ntri;                                // zeroed by the host
            s.bind(ntri);
        }
    } else {
        s.d = Dg; s.p = A.pen_ws + (size_t)b * nn; s.n = n;                   // workspace zeroed by the host
    }
    for (int p = tid; p <= n; p += nthr) { int v = A.init_tour[(size_t)b * (n + 1) + p]; t[p] = (TT)v; }
    __syncthreads();

    const long long t_start = wall_clock64();
    Trace<TR> tr;
    tr.len = 0;
    if constexpr (TR) {
        tr.cap = A.trace_cap; tr.t0 = t_start;
        tr.cost = A.trace_cost + (size_t)b * A.trace_cap;
        tr.time = A.trace_time ? A.trace_time + (size_t)b * A.trace_cap : nullptr;
    }
    constexpr bool eager_cost = TR;

    const double init_cost = A.init_cost[b];
    const double k = 0.1 * init_cost / (double)n;                             // algorithms.py:137
    double cur_cost = init_cost;
    long long evals = 0;
    int phase = 0;
    int status = 0;
    STAMP_DECL;

    // improvement trace: (cost, device time, completed outer iterations) whenever the returned best improves
    // (algorithms.py:143,190-191) -- a few hundred entries per run however long it is.  The counter lives in the
    // output array itself (thread 0 is its only reader and writer): no register, no LDS.
    auto push_improvement = [&](double c, long long it) {
        if (!A.imp_len) return;
        const int l = A.imp_len[b];
        if (l < A.imp_cap) {
            const size_t q = (size_t)b * A.imp_cap + l;
            if (A.imp_cost) A.imp_cost[q] = c;
            if (A.imp_time) A.imp_time[q] = (float)((double)(wall_clock64() - t_start) * 1e-8);
            if (A.imp_iter) A.imp_iter[q] = it;
        }
        A.imp_len[b] = l + 1;
    };
    if (tid == 0 && A.imp_len) A.imp_len[b] = 0;

    if (!FI && nthr > kWave) { block_reduce_lds_init(ctl, tid); __syncthreads(); }
    STAMP_BEGIN();
    PruneCtx pc{{0u, 0u, 0u, 0u}, false};
    if constexpr (!FI && S::kSymmetric && WPS <= GLS_PRUNE_MAX_WPS) {
        // (a workgroup too small to hold its rows' list words in kNlPasses registers per lane runs the full scans)
        if (A.nl_id && A.prune_ok[b] && 8 * (n - 1) <= kNlPasses * nthr) {
            const uint8_t *nl = A.nl_id + (size_t)b * n * kNL;
            auto word = [&](int it) -> unsigned {
                const int task = it * nthr + tid;
                if (task >= 8 * (n - 1)) return 0u;
                const uint8_t *r = nl + (size_t)(1 + (task >> 3)) * kNL + (task & 7);
                return (unsigned)r[0] | ((unsigned)r[8] << 8) | ((unsigned)r[16] << 16) | ((unsigned)r[24] << 24);
            };
            static_assert(kNlPasses == 4, "NlWords has four members");
            pc.nlw.w0 = word(0); pc.nlw.w1 = word(1); pc.nlw.w2 = word(2); pc.nlw.w3 = word(3);
            pc.on = true;
        }
    }
    long long xe = 0;        // executed minus reference-equivalent evaluations of this wavefront's pruned scans (CNT builds)
    local_search_dev<S, FI, GP, CNT, WPS>(s, t, t2, Ef, Eb, n, ctl, phase, cur_cost, tr, evals, xe, st, ppos, pc);   // algorithms.py:142
    double best_cost = cur_cost;                                              // algorithms.py:143
    if (tid == 0) push_improvement(best_cost, 0);
    for (int p = tid; p <= n; p += nthr) bt[p] = (int32_t)t[p];
    __syncthreads();

    long long iter_i = 0;
    long long pert_cycles = 0, pert_steps = 0;      // counting instantiations: shader cycles and penalty steps of the serial perturbation phase
    const long long c_start = CNT ? clock64() : 0;
    for (;;) {
        // ---- loop condition (algorithms.py:146) ----
        if (tid == 0) {
            long long el = wall_clock64() - t_start;
            int go;
            if (A.max_outer_iters >= 0) go = iter_i < A.max_outer_iters;
            else go = el < (long long)(A.time_limit_s * 1e8);
            if (el > (long long)(A.watchdog_s * 1e8)) { go = 0; status = GNNGLS_STATUS_WATCHDOG_DEV; }
            if (status != 0) go = 0;
            ctl->flag = go;
        }
        __syncthreads();
        if (!ctl->flag) break;
        const double *guide = A.guides + ((size_t)(iter_i % A.n_guides) * A.B + b) * nn;   // algorithms.py:147

        // ---- perturbation (algorithms.py:150-185): all wavefronts (TEAM) or wavefront 0 only ----
        if constexpr (TEAM) {
            STAMP_BEGIN();
            team_perturbation<S, FI>(s, k, t, t2, Ef, Eb, n, tc, guide, A, t_start, eager_cost, cur_cost, tr, evals, status, st);
        } else {
        if (wave == 0) {
            STAMP_BEGIN();
            // the serial chain of this instance competes for issue slots with the (latency-tolerant) descent
            // waves of the other resident workgroups on the same SIMD: give it priority while it runs
            __builtin_amdgcn_s_setprio(GLS_PERTURB_PRIO);
            if constexpr (kEdgeForm) {
                long long pc0 = 0;
                if constexpr (CNT) pc0 = clock64();
                serial_perturbation_edges<S, GP, TR, CNT>(s, k, t, t2, Ef, Eb, n, guide, A, t_start, cur_cost, tr, evals, status, pert_steps, st);
                if constexpr (CNT) pert_cycles += clock64() - pc0;
            } else {
            bool any_moved = false;
            int moves = 0;
            long long steps = 0;
            // utility numerators of the current tour edges, G.edges[e][guide] (algorithms.py:155), cached in
            // registers: lane p holds positions p, p+64, ... ; reloaded (asynchronously -- the values are only
            // consumed by the next arg-max) whenever the tour changes.
            double gq[GP];
            int pq[GP];                  // penalties of the same tour edges
            const bool greg = n <= GP * kWave;
            auto reload_guides = [&]() {
                if (!greg) return;
#pragma unroll
                for (int q = 0; q < GP; ++q) {
                    const int p = lane + q * kWave;
                    if (p < n) { const int u = t[p], v = t[p + 1]; gq[q] = guide[(size_t)u * n + v]; pq[q] = s.pen(u, v); }
                }
            };
            reload_guides();
            while (moves < A.perturbation_moves) {
                // arg-max utility over tour edges, first maximum wins (algorithms.py:153-159)
                double bu = 0.0; int bp = kNoKey;
                if (greg) {
#pragma unroll
                    for (int q = 0; q < GP; ++q) {
                        const int p = lane + q * kWave;
                        if (p < n) {
                            double util = gq[q] / (1.0 + (double)pq[q]);
                            if (bp == kNoKey || util > bu) { bu = util; bp = p; }
                        }
                    }
                } else {
                    for (int p = lane; p < n; p += kWave) {
                        int u = t[p], v = t[p + 1];
                        double util = guide[(size_t)u * n + v] / (1.0 + (double)s.pen(u, v));
                        if (bp == kNoKey || util > bu) { bu = util; bp = p; }
                    }
                }
                wave_argmax_first(bu, bp);
                STAMP_END(0);   // utility arg-max
                const int eu = t[bp], ev = t[bp + 1];
                bool ovf = false;
                if (greg) {
                    // algorithms.py:161.  The lane that caches tour edge bp holds its current count: it stores count + 1
                    // itself, so the serial chain has no load -> add -> store round trip through L1/L2
#pragma unroll
                    for (int q = 0; q < GP; ++q)
                        if (bp == lane + q * kWave) { ovf = s.pen_set(eu, ev, pq[q]); pq[q] += 1; }
                    ovf = __ballot(ovf) != 0ull;
                } else {
                    if (lane == 0) ovf = s.pen_inc(eu, ev);
                    ovf = __builtin_amdgcn_readfirstlane((int)ovf) != 0;
                }
                if (ovf) { status = GNNGLS_STATUS_PENALTY_OVERFLOW_DEV; break; }
                wave_sync();
                bool moved_this_step = false;
                for (int side = 0; side < 2; ++side) {                         // algorithms.py:167
                    const int node = side == 0 ? eu : ev;
                    if (node == 0) continue;                                   // algorithms.py:168
                    int i = bp + side;                                         // algorithms.py:169 cur_tour.index(n): the edge was read
                    if (moved_this_step) {                                     // at positions bp, bp+1; search only after a move
                        for (int p0 = 0; p0 <= n; p0 += kWave) {
                            int p = p0 + lane;
                            unsigned long long m = __ballot(p <= n && t[p] == node);
                            if (m) { i = p0 + __ffsll((long long)m) - 1; break; }
                        }
                    }
#pragma unroll 1
                    for (int op = 0; op < 2; ++op) {                           // algorithms.py:171
                        double bd = 0.0; int bk = kNoKey;
                        if (op == 0) scan_two_opt_o2a_guided<S, FI>(s, k, t, n, i, lane, bd, bk);
                        else         scan_relocate_o2a_guided<S, FI>(s, k, t, n, i, lane, bd, bk);
                        // most one-to-all scans find no improving move: one ballot decides whether the
                        // three-stage arg-min is needed at all
                        STAMP_END(1);   // penalty update + position search + o2a scan
                        if (__ballot(bk != kNoKey)) wave_reduce_best<FI>(bd, bk);
                        STAMP_END(2);   // reduction
                        if (lane == 0) evals += (op == 0) ? (n - 3) : (n - 2);
                        if (bk != kNoKey) {                                    // algorithms.py:175
                            apply_move(s, t, t2, Ef, Eb, n, op, i, bk, lane, kWave, eager_cost);
                            TT *x = t; t = t2; t2 = x;
                            wave_sync();
                            any_moved = true;
                            moved_this_step = true;
                            moves += 1;                                        // algorithms.py:185
                            reload_guides();
                            if (eager_cost) {
                                cur_cost = tour_cost_from_edges(Ef, n);        // algorithms.py:176
                                if (lane == 0) tr.push(cur_cost);
                            } else if (lane == 0) {
                                tr.len++;                                      // move counted, cost deferred
                            }
                            STAMP_END(3);   // apply move + reload
                        }
                    }
                }
                steps++;
                STAMP_COUNT(6);
                if ((steps & 63) == 0) {
                    long long el = wall_clock64() - t_start;
                    if (el > (long long)(A.watchdog_s * 1e8)) { status = GNNGLS_STATUS_WATCHDOG_DEV; break; }
                }
            }
            if (any_moved && !eager_cost) {
                build_edges(s, t, Ef, Eb, n, lane, kWave);
                wave_sync();
                cur_cost = tour_cost_from_edges(Ef, n);
            }
            }
            // tour buffers may have been swapped an odd number of times: publish which one is current
            if (lane == 0) { ctl->cost = cur_cost; ctl->pad = (int)((unsigned char *)t - smem); }
            __builtin_amdgcn_s_setprio(0);
            STAMP_END(4);       // phase tail
        }
        __syncthreads();
        {
            TT *cur = reinterpret_cast<TT *>(smem + ctl->pad);
            if (cur != t) { TT *x = t; t = t2; t2 = x; }
            cur_cost = ctl->cost;
        }
        }

        // ---- optimisation (algorithms.py:188) ----
        STAMP_BEGIN();
        local_search_dev<S, FI, GP, CNT, WPS>(s, t, t2, Ef, Eb, n, ctl, phase, cur_cost, tr, evals, xe, st, ppos, pc);
        STAMP_END(5);           // descent
        if (cur_cost < best_cost) {                                            // algorithms.py:190-191
            best_cost = cur_cost;
            for (int p = tid; p <= n; p += nthr) bt[p] = (int32_t)t[p];
            if (tid == 0) push_improvement(best_cost, iter_i + 1);
        }
        iter_i++;
        __syncthreads();
    }

    // ---- outputs ----
    if (tid == 0 && A.imp_len) {
        // terminal entry (returned best, end of the search, completed iterations): always the last one, always counted
        // (imp_len = improvements + 1 also when imp_cap == 0, as the header says) and, given a buffer, always stored --
        // if the improvements overflowed the buffer it takes the last slot
        const int l = A.imp_len[b];
        if (A.imp_cap > 0) {
            const size_t q = (size_t)b * A.imp_cap + (l < A.imp_cap ? l : A.imp_cap - 1);
            if (A.imp_cost) A.imp_cost[q] = best_cost;
            if (A.imp_time) A.imp_time[q] = (float)((double)(wall_clock64() - t_start) * 1e-8);
            if (A.imp_iter) A.imp_iter[q] = iter_i;
        }
        A.imp_len[b] = l + 1;
    }
    if constexpr (CNT) {
        // executed evaluations = evals (thread 0: everything the reference evaluates) + every wavefront's xe (candidates of
        // the pruned scans minus those scans' reference counts).  Zeroed by the host.
        if (A.evals_exec && lane == 0)
            atomicAdd(reinterpret_cast<unsigned long long *>(A.evals_exec + b), (unsigned long long)(xe + (tid == 0 ? evals : 0ll)));
        // ... and the cycle budget of the run for bench.py's critical-path figure: [B .. 5B) of the same buffer = shader cycles
        // of the kernel, shader cycles and penalty steps of the serial perturbation phase (edge form; else 0), 100 MHz ticks
        if (A.evals_exec && tid == 0) {
            A.evals_exec[(size_t)A.B + b] = clock64() - c_start;
            A.evals_exec[(size_t)2 * A.B + b] = pert_cycles;
            A.evals_exec[(size_t)3 * A.B + b] = pert_steps;
            A.evals_exec[(size_t)4 * A.B + b] = wall_clock64() - t_start;
        }
    } else if (A.evals_exec && tid == 0) {
        // the host only hands this instantiation the buffer when no scan of the run is pruned: executed = reference count
        A.evals_exec[b] = evals;
    }
    if (tid == 0) {
        A.best_cost[b] = best_cost;
        if (A.outer_iters) A.outer_iters[b] = iter_i;
        if (A.trace_len) A.trace_len[b] = tr.len;
        if (A.evals) A.evals[b] = evals;
        if (A.status) A.status[b] = status;
#ifdef GLS_STAMPS
        if (A.stamps) { long long *o = A.stamps + (size_t)b * 16; for (int q = 0; q < 12; ++q) if (q != 7) o[q] = st.acc[q]; }
#endif
    }
#ifdef GLS_STAMPS
    if (A.stamps && lane == 0 && wave < 16) {
        // further regions of the stamp buffer, [B,16] each, one slot per wavefront: unit cycles of the team rounds, then the
        // descent's scan cycles and its arg-min + wait cycles
        long long *o2 = A.stamps + (size_t)A.B * 16 + (size_t)b * 16;
        if (TEAM) o2[wave] = st.acc[12];
        else if (wave == 0) { o2[0] = st.acc[16]; o2[1] = st.acc[17]; o2[2] = st.acc[18]; o2[3] = st.acc[19]; o2[4] = st.acc[15];      // pruned-scan counters
                              o2[5] = st.acc[12]; o2[6] = st.acc[13]; o2[7] = st.acc[14]; }   // edge form: latch, divisions
        o2[(size_t)A.B * 16 + wave] = st.acc[8];
        o2[(size_t)A.B * 32 + wave] = st.acc[9];
    }
    // per-wavefront view of the descent (waves 1..3; wave 0 is slots 8 / 9) and where the hardware placed each wave:
    // slot 12 = wave 1's arg-min + wait, slots 13..15 = scan cycles of waves 1..3, slot 7 = (SIMD id + 1) << 8 wave
    if (A.stamps && lane == 0) {
        long long *o = A.stamps + (size_t)b * 16;
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        if (wave < 8) atomicAdd((unsigned long long *)&o[7], (unsigned long long)(((hw >> 4) & 3) + 1) << (8 * wave));
        if (wave == 1) o[12] = st.acc[9];
        if (wave >= 1 && wave <= 3) o[12 + wave] = st.acc[8];
    }
#endif
    if (A.penalty_out) {
        int32_t *po = A.penalty_out + (size_t)b * nn;
        for (size_t q = tid; q < nn; q += nthr) {
            int a = (int)(q / n), c = (int)(q % n);
            po[q] = (a == c) ? 0 : s.pen(a, c);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Unit / operator kernels (global-memory store: exact reference index order, asymmetric D allowed)
// ---------------------------------------------------------------------------------------------
__global__ void delta_all_kernel(const int32_t *tour, const double *D, int n, int op, double *out) {
    const int b = blockIdx.x;
    const int32_t *t = tour + (size_t)b * (n + 1);
    GlobalStore s{D + (size_t)b * n * n, nullptr, n};
    PlainDist<GlobalStore> f{s};
    double *o = out + (size_t)b * (n + 1) * (n + 1);
    const int total = (n + 1) * (n + 1);
    for (int q = threadIdx.x; q < total; q += blockDim.x) {
        int i = q / (n + 1), j = q % (n + 1);
        double v = __builtin_nan("");
        if (i >= 1 && i <= n - 1 && j >= 1 && j <= n - 1)
            v = op == 0 ? two_opt_cost(t, f, i, j) : relocate_cost(t, f, i, j);
        o[q] = v;
    }
}

template <bool FI>
__global__ void best_move_kernel(const int32_t *tour, const double *D, int n, int op, const int32_t *pos_i,
                                 double *delta_out, int32_t *move_out, int32_t *new_tour) {
    __shared__ Ctl ctl;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & (kWave - 1), wave = tid >> 6, nwaves = nthr >> 6;
    const int32_t *t = tour + (size_t)b * (n + 1);
    GlobalStore s{D + (size_t)b * n * n, nullptr, n};
    PlainDist<GlobalStore> f{s};
    double bd = 0.0; int bk = kNoKey;
    int i0 = pos_i ? pos_i[b] : 0;
    if (pos_i) {
        if (op == 0) scan_two_opt_o2a<PlainDist<GlobalStore>, FI>(t, f, n, i0, tid, nthr, bd, bk);
        else         scan_relocate_o2a<PlainDist<GlobalStore>, FI>(t, f, n, i0, tid, nthr, bd, bk);
    } else if (op == 0) {
        for (int i = 1 + wave; i <= n - 3; i += nwaves)
            for (int j = i + 2 + lane; j <= n - 1; j += kWave)
                consider<FI>(two_opt_cost(t, f, i, j), make_key(i, j), bd, bk);
    } else {
        for (int i = 1 + wave; i <= n - 1; i += nwaves)
            for (int j = 1 + lane; j <= n - 1; j += kWave) {
                if (j == i || i - j == 1) continue;
                consider<FI>(relocate_cost(t, f, i, j), make_key(i, j), bd, bk);
            }
    }
    int phase = 0;
    block_reduce_best<FI>(&ctl, phase, wave, nwaves, lane, bd, bk);
    int mi = 0, mj = 0;
    if (bk != kNoKey) { if (pos_i) { mi = i0; mj = bk; } else { mi = bk >> 16; mj = bk & 0xffff; } }
    if (tid == 0) {
        delta_out[b] = (bk != kNoKey) ? bd : 0.0;
        move_out[2 * b] = mi; move_out[2 * b + 1] = mj;
    }
    if (new_tour) {
        int32_t *o = new_tour + (size_t)b * (n + 1);
        for (int p = tid; p <= n; p += nthr) o[p] = (bk != kNoKey) ? t[move_src(op, p, mi, mj)] : t[p];
    }
}

__global__ void tour_cost_kernel(const int32_t *tour, const double *D, int B, int n, double *out) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int32_t *t = tour + (size_t)b * (n + 1);
    const double *d = D + (size_t)b * n * n;
    double c = 0.0;
    for (int p = 0; p < n; ++p) c += d[(size_t)t[p] * n + t[p + 1]];
    out[b] = c;
}

// nearest_neighbor (algorithms.py:9-18): one wavefront per instance; ties -> lowest node id.
__global__ void nearest_neighbor_kernel(const double *W, int n, int depot, int32_t *tour_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint8_t *visited = smem;
    const int b = blockIdx.x, lane = threadIdx.x;
    const double *w = W + (size_t)b * n * n;
    int32_t *t = tour_out + (size_t)b * (n + 1);
    for (int j = lane; j < n; j += kWave) visited[j] = (j == depot);
    wave_sync();
    int cur = depot;
    if (lane == 0) { t[0] = depot; t[n] = depot; }
    for (int len = 1; len < n; ++len) {
        double bw = 0.0; int bj = kNoKey;
        for (int j = lane; j < n; j += kWave) {
            if (visited[j]) continue;
            double x = w[(size_t)cur * n + j];
            if (bj == kNoKey || x < bw) { bw = x; bj = j; }
        }
        { double nb = -bw; wave_argmax_first(nb, bj); }   // min weight, ties -> lowest node id
        if (lane == 0) { t[len] = bj; visited[bj] = 1; }
        wave_sync();
        cur = bj;
    }
}

// ---------------------------------------------------------------------------------------------
// Host-side launchers
// ---------------------------------------------------------------------------------------------
size_t gls_lds_bytes(int n, int store, int penalty_bits, bool team) {
    auto r16 = [](size_t x) { return (x + 15) & ~size_t(15); };
    const size_t tour_elem = store == GLS_STORE_COMPACT ? 1 : 4;
    size_t off = r16(sizeof(Ctl)) + r16((size_t)(n + 2) * 8) + 3 * r16((size_t)(n + 1) * tour_elem);
    if (team) off += r16(sizeof(TeamCtl));
    if (store == GLS_STORE_GLOBAL) off += r16((size_t)(n + 2) * 8);
    size_t ntri = (size_t)n * (n - 1) / 2;
    if (store != GLS_STORE_GLOBAL) off += r16(ntri * 8);
    if (store == GLS_STORE_TRI) off += r16(ntri * (size_t)(penalty_bits / 8));
    return off;
}

static std::atomic<int> g_threads_override{0};      // experiments only (gnngls_debug_set_gls_threads)
void gls_set_block_threads_override(int threads) { g_threads_override.store(threads, std::memory_order_relaxed); }

int gls_block_threads(int n, int store, int penalty_bits, bool half_scans) {
    const int forced = g_threads_override.load(std::memory_order_relaxed);
    if (forced > 0) return forced;
    if (n <= 24) return 64;
    // n <= 33 on the stores that have the half-wave descent scans: ONE wavefront using both its 32-lane halves beats two
    // wavefronts sharing the lean scans (outer iterations in 2 s, x 1000, noise guide: n = 26 24.2k -> 25.7k, n = 30 22.6k ->
    // 24.1k, n = 33 21.1k -> 22.4k)
    // (half_scans = false: the caller knows the launch ends up on an instantiation without them -- first improvement, or the
    // 64- / 80-VGPR builds -- where one wavefront would run the two-wavefront scans alone)
    if (GLS_HALF_SCANS && half_scans && n <= kHalfScanMaxNodes && penalty_bits == 32 && (store == GLS_STORE_COMPACT || store == GLS_STORE_TRI))
        return 64;
    if (n <= 48) return 128;
    if (n <= 80) return 256;
    // compact store with the lean descent scans (n <= 127), four workgroups per CU, measured at TSP100 x 1024 (outer
    // iterations per instance in 2 s, weight / noise guide): 8 waves at 64 VGPRs (108 B of scratch) 11.6k / 7.0k;
    // 4 waves at 128 VGPRs (no scratch) 12.3k / 7.5k  <- used.  (One workgroup alone on a CU prefers 8 waves, 16.1k vs
    // 14.8k, but such small batches run on the LDS-penalty store anyway.)
    if (store == GLS_STORE_COMPACT && n <= 2 * kWave - 1) return 256;
    // compact store with ONE workgroup per CU (distance triangle > 80 KB, n >= 144; TSP200): the descent is latency-bound
    // at two waves per SIMD -- 16 waves share the scans (measured TSP200 x 256, noise guide, iterations in 2 s: 8 waves
    // 4.7k, 16 waves see profiles/)
    if (store == GLS_STORE_COMPACT && 2 * gls_lds_bytes(n, GLS_STORE_COMPACT, 32) > 160 * 1024) return 1024;
    return 512;
}

int gls_waves_per_simd(int store, int n, int batch, int num_cus, int threads, size_t lds) {
    if (store == GLS_STORE_GLOBAL) return GlobalStore::kWavesPerSimd;
    if (store == GLS_STORE_TRI) {
        // LDS-penalty store: the 128-VGPR build (no scratch) while it keeps the batch resident, else the 80-VGPR one
        const int waves = threads / kWave;
        const int by_lds = (int)((160 * 1024) / lds);
        const int per_cu4 = by_lds < 16 / waves ? by_lds : 16 / waves;
        // ... and only where the smaller register budget actually buys residency (n = 150: one workgroup per CU by LDS
        // either way -- the 80-VGPR build spills for nothing)
        const int per_cu6 = by_lds < 24 / waves ? by_lds : 24 / waves;
        if (batch > 0 && per_cu4 >= 1 && ((long)per_cu4 * num_cus >= batch || per_cu6 <= per_cu4)) return 4;
        return TriStore<int32_t>::kWavesPerSimd;
    }
    // compact store: the 128-VGPR build unless the batch only fits with 8 waves per SIMD
    const int waves = threads / kWave;
    const int by_lds = (int)((160 * 1024) / lds);
    const int per_cu4 = by_lds < 16 / waves ? by_lds : 16 / waves;
    (void)n;
    return (batch > 0 && (long)per_cu4 * num_cus < batch && 32 / waves > per_cu4 && by_lds > per_cu4) ? 8 : 4;
}

// resource query (gls_kernel_resources): with the slot set, the launch chain stops at the selected instantiation and hands back
// its function instead of launching it
static thread_local const void **t_query_fn = nullptr;

template <class S, bool FI, int GP, bool TR, int WPS, bool TEAM, bool CNT = false>
static hipError_t launch_gls_k(const GlsArgs &A, size_t lds, int threads, hipStream_t stream) {
    auto kern = gls_kernel<S, FI, GP, TR, WPS, TEAM, CNT>;
    if (t_query_fn) { *t_query_fn = reinterpret_cast<const void *>(kern); return hipSuccess; }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3(A.B), dim3(threads), lds, stream, A);
    return hipGetLastError();
}

template <class S, bool FI, int GP, int WPS, bool TEAM>
static hipError_t launch_gls_g(const GlsArgs &A, size_t lds, int threads, hipStream_t stream) {
    // counting instantiations (executed evaluations of the pruned scans): compact store, best improvement, 128-VGPR build,
    // no per-move trace -- gls_count_supported() says the same to the host
    if constexpr (!FI && WPS == 4 && GP >= 2 && std::is_base_of<TriDGlobalP, S>::value) {
        if (A.evals_exec && A.nl_id && !(A.trace_cap > 0 && A.trace_cost)) return launch_gls_k<S, FI, GP, false, WPS, TEAM, true>(A, lds, threads, stream);
    }
    // (a run that prunes AND asks for the executed-evaluation count must have landed on a counting instantiation above: the
    // others would report executed == reference evaluations, a silently wrong ratio -- gls_count_supported() is the host's copy
    // of the rule)
    if (A.evals_exec && A.nl_id) return hipErrorInvalidValue;
    // trace_cap == 0 (no trace buffer): the trace-free instantiation (fewer live registers in the serial phase)
    if (A.trace_cap > 0 && A.trace_cost) return launch_gls_k<S, FI, GP, true, WPS, TEAM>(A, lds, threads, stream);
    return launch_gls_k<S, FI, GP, false, WPS, TEAM>(A, lds, threads, stream);
}

template <class S, bool FI, int WPS, bool TEAM>
static hipError_t launch_gls_t(const GlsArgs &A, size_t lds, int threads, hipStream_t stream) {
    // small instances on single-wavefront workgroups (TSP20): the one-slot instantiation, whose descent scans use both
    // 32-lane halves of the wavefront (scan_*_a2a_lean_half) -- an instantiation of its own so that the register
    // allocation of the others (the TSP100 headline runs on GP = 2) does not see that code
    if constexpr (!FI && !TEAM && WPS == 4 && S::kSymmetric && sizeof(typename S::pen_t) == 4) {
        if (GLS_HALF_SCANS && A.n >= kHalfScanMinNodes && A.n <= kHalfScanMaxNodes && threads == kWave)
            return launch_gls_g<S, FI, 1, WPS, TEAM>(A, lds, threads, stream);
        // the edge form of the serial perturbation phase evaluates every register slot of a lane: n <= 63 (tour positions
        // 0 .. n in one slot) runs on the one-slot instantiation whatever the workgroup shape (TSP50)
        if (GLS_EDGE_PERTURB && A.n <= kWave - 1) return launch_gls_g<S, FI, 1, WPS, TEAM>(A, lds, threads, stream);
    }
    // register-cached guide/penalty values of the tour edges: 2 passes of 64 lanes cover positions 0..n for n <= 127
    if (A.n + 1 <= 2 * kWave) return launch_gls_g<S, FI, 2, WPS, TEAM>(A, lds, threads, stream);
    return launch_gls_g<S, FI, kGuidePassesMax, WPS, TEAM>(A, lds, threads, stream);
}

template <class S, int WPS, bool TEAM = false>
static hipError_t launch_gls_f(const GlsArgs &A, size_t lds, int threads, bool first_improvement, hipStream_t stream) {
    return first_improvement ? launch_gls_t<S, true, WPS, TEAM>(A, lds, threads, stream)
                             : launch_gls_t<S, false, WPS, TEAM>(A, lds, threads, stream);
}

// 256-VGPR instantiation (two waves per SIMD) of the one-slot kernel: single-wavefront workgroups, n = 8 .. 33, best improvement
bool gls_wps2_supported(int store, int penalty_bits, int n, int threads, bool first_improvement) {
    return GLS_WPS2 && GLS_HALF_SCANS && !first_improvement && threads == kWave && n >= kHalfScanMinNodes && n <= kHalfScanMaxNodes &&
           penalty_bits == 32 && (store == GLS_STORE_COMPACT || store == GLS_STORE_TRI);
}

// the serial perturbation phase runs in its edge form (serial_perturbation_edges) on the symmetric stores with 32-bit counters,
// best improvement, the 128-VGPR and wider builds -- the host's copy of kEdgeForm in gls_kernel
bool gls_edge_form(int store, int penalty_bits, int wps, bool team, bool first_improvement) {
    return GLS_EDGE_PERTURB && store != GLS_STORE_GLOBAL && !(store == GLS_STORE_TRI && penalty_bits == 16) && !first_improvement && !team && wps <= 4;
}

bool gls_team_supported(int store, int penalty_bits, int wps, int n, int threads) {
    // the team form exists for the 128-VGPR builds of the two symmetric stores with 32-bit counters, n <= 255; it caches the
    // utilities of the tour edges by position on wavefronts 0 .. ceil(n / 64) - 1, so the workgroup needs that many
    return (store == GLS_STORE_COMPACT || (store == GLS_STORE_TRI && penalty_bits == 32)) && wps == 4 && n >= 4 && n <= 255 &&
           threads / kWave >= (n + kWave - 1) / kWave;
}

hipError_t launch_gls(const GlsArgs &A, int store, int penalty_bits, int threads, int wps, bool team, bool first_improvement,
                      hipStream_t stream) {
    if (team && !gls_team_supported(store, penalty_bits, wps, A.n, threads)) return hipErrorInvalidValue;
    size_t lds = gls_lds_bytes(A.n, store, penalty_bits, team);
#ifdef GLS_DEV_ONLY_HEADLINE
    // development builds (ISA inspection, fast compiles): only the instantiation the TSP100 x 1024 headline runs on
    (void)store; (void)penalty_bits; (void)wps; (void)first_improvement;
    return launch_gls_k<TriDGlobalP, false, GLS_DEV_ONLY_HEADLINE + 0 == 4 ? 4 : 2, false, 4, false>(A, lds, threads, stream);   // (-DGLS_DEV_ONLY_HEADLINE=4: TSP200's)
#else
#if GLS_WPS2
    if (wps == 2) {          // single-wavefront workgroups on the 256-VGPR build (gls_wps2_supported)
        if (!gls_wps2_supported(store, penalty_bits, A.n, threads, first_improvement) || team) return hipErrorInvalidValue;
        if (store == GLS_STORE_COMPACT) return launch_gls_g<TriDGlobalP, false, 1, 2, false>(A, lds, threads, stream);
        return launch_gls_g<TriStore<int32_t>, false, 1, 2, false>(A, lds, threads, stream);
    }
#endif
    if (store == GLS_STORE_COMPACT) {
        if (team) return launch_gls_f<TriDGlobalPF, 4, true>(A, lds, threads, first_improvement, stream);
        return wps == 8 ? launch_gls_f<TriDGlobalP, 8>(A, lds, threads, first_improvement, stream)
                        : launch_gls_f<TriDGlobalP, 4>(A, lds, threads, first_improvement, stream);
    }
    if (store == GLS_STORE_TRI && penalty_bits == 16)
        return launch_gls_f<TriStore<uint16_t>, TriStore<uint16_t>::kWavesPerSimd>(A, lds, threads, first_improvement, stream);
    if (store == GLS_STORE_TRI) {
        if (team) return launch_gls_f<TriStore<int32_t>, 4, true>(A, lds, threads, first_improvement, stream);
        return wps == 4 ? launch_gls_f<TriStore<int32_t>, 4>(A, lds, threads, first_improvement, stream)
                        : launch_gls_f<TriStore<int32_t>, TriStore<int32_t>::kWavesPerSimd>(A, lds, threads, first_improvement, stream);
    }
    return launch_gls_f<GlobalStore, GlobalStore::kWavesPerSimd>(A, lds, threads, first_improvement, stream);
#endif
}

// registers and scratch of the instantiation launch_gls would run for these arguments (hipFuncGetAttributes: needs the device)
hipError_t gls_kernel_resources(const GlsArgs &A, int store, int penalty_bits, int threads, int wps, bool team, bool first_improvement,
                                int *vgprs, int *scratch_bytes) {
    const void *fn = nullptr;
    t_query_fn = &fn;
    const hipError_t e = launch_gls(A, store, penalty_bits, threads, wps, team, first_improvement, nullptr);
    t_query_fn = nullptr;
    if (e != hipSuccess) return e;
    if (!fn) return hipErrorInvalidValue;
    hipFuncAttributes at;
    const hipError_t e2 = hipFuncGetAttributes(&at, fn);
    if (e2 != hipSuccess) return e2;
    if (vgprs) *vgprs = at.numRegs;
    if (scratch_bytes) *scratch_bytes = (int)at.localSizeBytes;
    return hipSuccess;
}

// executed-evaluation counting (measurement hook) exists where gls_count_supported says; elsewhere a run that prunes cannot
// report it
bool gls_count_supported(int store, int wps, int n, bool first_improvement, bool trace) {
    (void)n;
    return store == GLS_STORE_COMPACT && wps == 4 && !first_improvement && !trace;
}

// pruned descent scans exist in the 4-slot instantiations of the symmetric stores (n >= 128), best improvement only; the
// lists must be full (n - 1 >= 32)
bool gls_prune_supported(int store, int n, bool first_improvement, int wps) {
    return store != GLS_STORE_GLOBAL && !first_improvement && n >= kPruneMinNodes && n <= 255 && wps <= GLS_PRUNE_MAX_WPS;
}

hipError_t launch_symmetry_check(const double *D, int B, int n, int32_t *asym, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(symmetry_kernel, dim3(B), dim3(256), 0, stream, D, n, asym);
    return hipGetLastError();
}

hipError_t launch_neighbor_lists(const double *D, int B, int n, uint8_t *nl_id, int32_t *prune_ok, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(neighbor_lists_kernel, dim3(B), dim3(256), 0, stream, D, n, nl_id, prune_ok);
    return hipGetLastError();
}

hipError_t launch_delta_all(const int32_t *tour, const double *D, int B, int n, int op, double *out, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(delta_all_kernel, dim3(B), dim3(256), 0, stream, tour, D, n, op, out);
    return hipGetLastError();
}

hipError_t launch_best_move(const int32_t *tour, const double *D, int B, int n, int op, const int32_t *pos_i,
                            bool first_improvement, double *delta_out, int32_t *move_out, int32_t *new_tour,
                            hipStream_t stream) {
    int threads = pos_i ? 64 : gls_block_threads(n, GLS_STORE_GLOBAL);
    (void)hipGetLastError();
    if (first_improvement)
        hipLaunchKernelGGL(best_move_kernel<true>, dim3(B), dim3(threads), 0, stream, tour, D, n, op, pos_i, delta_out, move_out, new_tour);
    else
        hipLaunchKernelGGL(best_move_kernel<false>, dim3(B), dim3(threads), 0, stream, tour, D, n, op, pos_i, delta_out, move_out, new_tour);
    return hipGetLastError();
}

hipError_t launch_tour_cost(const int32_t *tour, const double *D, int B, int n, double *out, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(tour_cost_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, tour, D, B, n, out);
    return hipGetLastError();
}

hipError_t launch_nearest_neighbor(const double *W, int B, int n, int depot, int32_t *tour_out, hipStream_t stream) {
    size_t lds = ((size_t)n + 15) & ~size_t(15);
    (void)hipGetLastError();
    hipLaunchKernelGGL(nearest_neighbor_kernel, dim3(B), dim3(64), lds, stream, W, n, depot, tour_out);
    return hipGetLastError();
}

}  // namespace gnngls
