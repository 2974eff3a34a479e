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
//
// One translation unit in four files (round 5; the code generation of the ~70 kernel instantiations depends on all of it, so it
// stays one unit): gls_common.h (switches, storage policies, move evaluation, selection, reductions), gls_descent_scans.h (the
// all-to-all scans of the descent), gls_perturbation.h (one-to-all scans and the three forms of the perturbation phase), and this
// file (local_search, the persistent kernel, unit kernels, launch policy).
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>

#include <atomic>
#include <type_traits>

#include "gls_kernels.h"

#pragma clang fp contract(off)

namespace gnngls {

#include "gls_common.h"
#include "gls_descent_scans.h"
#include "gls_perturbation.h"

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

// cycle account of the descent (counting instantiations only; bench.py's critical_path record): shader cycles of wavefront 0 in the
// scans of each kind, in the workgroup arg-min (incl. waiting for the slowest wavefront) and in move application + barrier.  Thread
// 0 adds them straight to the instance's records in global memory (returnless atomics: no accumulators live across the kernel --
// nine 64-bit counters in registers made the counting build 18 % slower than the product it is meant to describe).
//   rec[q * B]: 0 descent cycles, 1/2 count and cycles of two_opt_a2a, 3/4 of relocate_a2a in full, 5/6 of relocate_a2a over the
//   flagged rows, 7 arg-min, 8 move application, 9 accepted moves   (include/gnngls_hip.h, records 5 .. 14)
struct DescentCycles {
    long long *rec; size_t B;
    __device__ __forceinline__ void add(int q, long long v) const {
        __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(rec + (size_t)q * B), (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

template <class S, bool FI, int GP, bool CNT, int WPS, class TT, class TRC>
__device__ __forceinline__ void local_search_dev(const S &s, TT *&t, TT *&t2, double *Ef, double *Eb, int n,
                                 Ctl *ctl, int &phase, double &cur_cost, TRC &tr, long long &evals, long long &xe, Stamps &st,
                                 TT *ppos, const PruneCtx &pc, const DescentCycles &dc) {
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
    if constexpr (GLS_QUIET_ROWS && kCanPrune && (GP == 2 || (GP == 4 && S::kPenInLds)) && WPS <= 4) {
        if (prune && tid == 0) quiet_reset<GP>(quiet_lds(ctl), lds_byte_addr(s.d));
    }
    __syncthreads();
    if (prune) Lmax = unsortable(*lmax_slot(ctl));
    // pruned relocate scan: tour edges longer than Lcap (three mean edge lengths of the tour the descent starts from) are
    // listed per scan (at most kLongCap, else that scan runs unpruned) in exchange slots the descent does not use
    constexpr int kLongCap = 16;
    const double Lcap = GLS_LCAP_FACTOR * cur_cost / (double)n;
    uint8_t *longk = long_edge_list(ctl);                        // 16 positions (n <= 255), see QuietLds for the layout of the slots
    int *nlong_slot = long_edge_count(ctl);
    (void)Lmax;
    // quiet rows of the relocate scan (gls_descent_scans.h): the two-slot builds with neighbour lists on, i.e. 80 <= n <= 127 on the
    // lean relocate scan; the first relocate scan of a descent is the full one and sets the bits
    // four-slot builds: the LDS-penalty store only (n = 128 .. 163 at one instance per CU, 8 wavefronts: +6 .. +13 %); in the compact
    // store's four-slot instantiation -- TSP200 on 16 wavefronts, where the scheme does not pay -- the code's mere presence cost 2.8 %
    // (scratch 88 -> 104 B), so it is compiled out there
    constexpr bool kQuietRows = GLS_QUIET_ROWS && kCanPrune && (GP == 2 || (GP == 4 && S::kPenInLds)) && WPS <= 4;
    // (with 16 wavefronts per instance -- the compact store from n = 144 up: TSP200 -- every wavefront pays the refresh for 13 nodes and the
    // pruned relocate scan it replaces is only 1.55 passes: measured -1 .. -2 % at n = 170 .. 200, +6 % (model guide) / +12 % (noise) at
    // n = 140 .. 160 on 8 wavefronts, profiles/r06_experiments/ab_quiet_rows_gp4*.log)
    const bool quiet = kQuietRows && prune && nwaves >= 2 && (nwaves & (nwaves - 1)) == 0 && n <= GP * kWave - 1 && n - 1 <= nwaves * kWave &&
                       (GP == 2 || nwaves <= 8);
    const QuietLds ql = quiet_lds(ctl);
    QuietLane qme{false};
    bool have_bits = false;
    bool improved = true;
    while (improved) {                                               // algorithms.py:116
        improved = false;
#pragma unroll 1
        for (int op = 0; op < 2; ++op) {                             // algorithms.py:119
            double bd = 0.0; int bk = kNoKey;
            bool lean = false, pruned_scan = false, quiet_scan = false;
            long long cyc0 = 0;
            if constexpr (CNT) cyc0 = clock64();
            int xs = 0;          // evaluations this wavefront executes in a pruned scan (scalar; booked once per scan below)
            if constexpr (kCanPrune) {
                int nlong = 0;
                if (kPruneRelocate && prune && op == 1 && !(kQuietRows && quiet && have_bits)) {      // (a flagged-rows scan needs no long-edge list)
                    if (tid == 0) *nlong_slot = 0;
                    __syncthreads();
                    for (int q = 1 + tid; q <= n; q += nthr)
                        if (Ef[q] > Lcap) {
                            typedef __attribute__((address_space(3))) int lds_i32_t;
                            const int slot = __hip_atomic_fetch_add((lds_i32_t *)nlong_slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (slot < kLongCap) longk[slot] = (uint8_t)(q - 1); // target edge k = (t[k], t[k+1]), Ef[k+1] its length
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
                    else if constexpr (kPruneRelocate) {
                        if (kQuietRows && quiet && have_bits) {
                            // (n >= 128: the flagged-rows scan below takes this relocate scan; the list walk only runs for the first of a descent)
                        } else if (kQuietRows && quiet) {
                            scan_relocate_a2a_pruned<CNT, kQuietRows, S, TT>(s, t, ppos, Ef, pc.nlw, n, Lcap, longk, nlong, tid, nthr, lane, bd, bk, xs, dbg, ql.words);
                            have_bits = true; lean = true; pruned_scan = true;
                        } else {
                            scan_relocate_a2a_pruned<CNT, false, S, TT>(s, t, ppos, Ef, pc.nlw, n, Lcap, longk, nlong, tid, nthr, lane, bd, bk, xs, dbg);
                            lean = true; pruned_scan = true;
                        }
                    }
                    if (op == 0) { lean = true; pruned_scan = true; }
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
                if constexpr (kQuietRows) {
                    if (!lean && op == 1 && quiet && have_bits) {
#ifdef GLS_STAMPS
                        const long long q0 = clock64();
#endif
                        const bool ok = scan_relocate_a2a_quiet<CNT, GP, S, TT>(s, t, ppos, Ef, n, ql, qme, wave_u, nwaves, lane, bd, bk, xs);
                        if (ok) {
#ifdef GLS_STAMPS
                            st.acc[20] += 1; st.acc[21] += clock64() - q0; st.acc[23] += __popcll(__ballot(qme.act));
#endif
                            lean = true; pruned_scan = true; quiet_scan = true;
                        } else {                                     // pending list overflow (cannot happen): start over with a full scan
                            have_bits = false; qme.act = false;
                            __syncthreads();
                            if (tid == 0) quiet_reset<GP>(ql, lds_byte_addr(s.d));
                            __syncthreads();
                        }
                    }
                }
                if (!lean && nwaves >= (n - 1 + kWave - 1) / kWave && n <= GP * kWave - 1) {
                    // group filter on the 128-VGPR and wider builds, six steps per relocate group where two register slots leave room
                    constexpr bool kMF = WPS <= 4;
                    constexpr int kUR = (WPS <= 4 && GP == 2) ? GLS_LEAN_UNROLL_RELOCATE : GLS_LEAN_UNROLL;
                    if (op == 0) scan_two_opt_a2a_lean<GP, kMF, S, TT>(s, t, Eb, n, wave, nwaves, lane, bd, bk);
                    else if (kQuietRows && quiet) {
                        scan_relocate_a2a_lean<GP, kMF, kUR, S, TT, kQuietRows>(s, t, Ef, n, wave, nwaves, lane, bd, bk, pos, ql.words);
                        have_bits = true;
                    }
                    else         scan_relocate_a2a_lean<GP, kMF, kUR, S, TT>(s, t, Ef, n, wave, nwaves, lane, bd, bk, pos);
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
            long long cyc1 = 0;
            if constexpr (CNT) {
                cyc1 = clock64();
                const int kind = op == 0 ? 0 : quiet_scan ? 2 : 1;
                if (tid == 0 && dc.rec) { dc.add(1 + 2 * kind, 1); dc.add(2 + 2 * kind, cyc1 - cyc0); }
            }
            if (!FI && nwaves > 1) block_reduce_best_lds(ctl, phase, tid, bd, bk);
            else block_reduce_best<FI>(ctl, phase, wave, nwaves, lane, bd, bk);
            if constexpr (CNT) { cyc0 = clock64(); if (tid == 0 && dc.rec) dc.add(7, cyc0 - cyc1); }
            if constexpr (kQuietRows) {
                // every wavefront has consumed the pending records and the rows to flag (two barriers ago at least): the wavefront that
                // notes the moves (the last one) clears them, in program order before it notes this scan's own move
                if (quiet_scan && wave_u == nwaves - 1 && lane == 0) {
#pragma unroll
                    for (int w = 0; w < GP; ++w) ql.words[w] = 0ull;
                    *ql.count = 0;
                }
            }
            STAMP_END(9);    // wave + workgroup arg-min (includes waiting for the slowest wave)
            STAMP_COUNT(11);
            if (tid == 0) evals += (op == 0) ? (long long)(n - 2) * (n - 3) / 2 : (long long)(n - 2) * (n - 2);
            // executed count = evals + sum over the wavefronts of xe: a pruned scan booked the candidates it evaluated, and
            // wavefront 0 takes the scan's reference count back (uniform branch: xe stays in scalar registers)
            if (CNT && pruned_scan)
                xe += xs - (wave_u == 0 ? ((op == 0) ? (n - 2) * (n - 3) / 2 : (n - 2) * (n - 2)) : 0);
            if (bk != kNoKey) {                                      // delta < 0 (algorithms.py:122)
                ISA_MARK("descent_apply_begin");
                improved = true;
                cur_cost += bd;                                      // algorithms.py:124
                apply_move(s, t, t2, Ef, Eb, n, op, bk >> 16, bk & 0xffff, tid, nthr, true, pos, prune ? ppos : nullptr, ctl, Lmax);
                if constexpr (kQuietRows) {
                    // the move's new tour edges go to the pending list of the next relocate scan's refresh (from the OLD tour)
                    if (quiet && have_bits && wave_u == nwaves - 1) quiet_note_move(ql, lds_byte_addr(s.d), t, op, bk >> 16, bk & 0xffff, lane);
                }
                TT *x = t; t = t2; t2 = x;
                if (tid == 0) tr.push(cur_cost);
                __syncthreads();
                if (prune) Lmax = unsortable(*lmax_slot(ctl));
                ISA_MARK("descent_apply_end");
                if constexpr (CNT) { if (tid == 0 && dc.rec) { dc.add(8, clock64() - cyc0); dc.add(9, 1); } }
                STAMP_END(10);   // move application + barrier
            }
        }
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
    // (the start descent of algorithms.py:142 is not in the account: rec = nullptr)
    const DescentCycles dc_first{nullptr, 0}, dc{(CNT && A.evals_exec) ? A.evals_exec + (size_t)5 * A.B + b : nullptr, (size_t)A.B};
    local_search_dev<S, FI, GP, CNT, WPS>(s, t, t2, Ef, Eb, n, ctl, phase, cur_cost, tr, evals, xe, st, ppos, pc, dc_first);   // algorithms.py:142
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
        long long dcy0 = 0;
        if constexpr (CNT) dcy0 = clock64();
        local_search_dev<S, FI, GP, CNT, WPS>(s, t, t2, Ef, Eb, n, ctl, phase, cur_cost, tr, evals, xe, st, ppos, pc, dc);
        if constexpr (CNT) { if (tid == 0 && dc.rec) dc.add(0, clock64() - dcy0); }
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
            // (records 5 .. 14, the cycle account of the descent, were added to in place: DescentCycles)
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
                              o2[5] = st.acc[12]; o2[6] = st.acc[13]; o2[7] = st.acc[14];     // edge form: latch, divisions
                              o2[8] = st.acc[20]; o2[9] = st.acc[21]; o2[10] = st.acc[22]; o2[11] = st.acc[23]; }   // quiet rows
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
