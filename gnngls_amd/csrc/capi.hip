// capi.hip -- extern "C" entry points of libgnngls_hip.so (declared in include/gnngls_hip.h).
// Argument checking, error strings and launch glue only; kernels live in gls_kernels.hip and
// model_kernels.hip.  No CPU fallback: every entry point either enqueues HIP work or fails.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/gnngls_hip.h"
#include "gls_kernels.h"
#include "model_kernels.h"
#include "train_kernels.h"

namespace {
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what) {
    return fail(GNNGLS_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

// ---- optional per-kernel-class timing with HIP events on the caller's stream -------------------
// (measurement hook for bench.py: process-global; the span log is guarded by a mutex so that concurrent callers on
// different host threads / streams may log safely; off unless gnngls_profile_enable(1) was called)
struct ProfSpan { int kind; hipEvent_t a, b; };
std::atomic<bool> g_prof_on{false};
std::mutex g_prof_mutex;
std::vector<ProfSpan> g_spans;

struct ProfScope {
    int kind; hipStream_t st; hipEvent_t a = nullptr;
    ProfScope(int k, hipStream_t s) : kind(k), st(s) {
        if (g_prof_on && hipEventCreate(&a) == hipSuccess) (void)hipEventRecord(a, st); else a = nullptr;
    }
    ~ProfScope() {
        if (!a) return;
        hipEvent_t b;
        if (hipEventCreate(&b) == hipSuccess) {
            (void)hipEventRecord(b, st);
            std::lock_guard<std::mutex> lock(g_prof_mutex);
            g_spans.push_back({kind, a, b});
        }
        else (void)hipEventDestroy(a);
    }
};

int g_pen16_limit = 65535;
std::atomic<int> g_prune_mode{-1};     // -1 / 1 = pruned descent scans where they exist, 0 = full scans (experiments / tests)
std::atomic<int> g_team_mode{-1};      // -1 = policy (gls_config), 0 = never, 1 = wherever the team form exists (experiments / tests)
long long *g_stamp_buffer = nullptr;
std::atomic<long long *> g_exec_evals{nullptr};   // measurement hook (gnngls_profile_set_executed_evals)

constexpr size_t kLdsPerCU = 160 * 1024;
constexpr int kMaxWavesPerCU = 32;

int num_cus() {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        return prop.multiProcessorCount;
    (void)hipGetLastError();
    return 256;   // MI355X
}

// Storage configuration of the persistent search kernel for instances of n nodes: the one with the
// most resident workgroups per CU wins; ties go to the faster store (LDS penalties, 32-bit first).
struct GlsConfig { int store; int penalty_bits; int threads; size_t lds; int per_cu; int wps; bool team = false; bool prune = false; };

GlsConfig gls_config_store(int n, int requested_bits, int batch, bool first_improvement);

// + the form of the perturbation phase: on all wavefronts of the workgroup (team) when every workgroup of the batch gets a
// CU of its own (B <= number of CUs) and is a 16-wave workgroup (TSP200 x 256: one per CU by LDS anyway) -- else on
// wavefront 0
GlsConfig gls_config(int n, int requested_bits, int batch = 0, bool first_improvement = false) {
    GlsConfig c = gls_config_store(n, requested_bits, batch, first_improvement);
    const int mode = g_team_mode.load(std::memory_order_relaxed);
    if (mode != 0 && gnngls::gls_team_supported(c.store, c.penalty_bits, c.wps, n, c.threads)) {
        const size_t lds = gnngls::gls_lds_bytes(n, c.store, c.penalty_bits, true);
        // ... and only for the 16-wave workgroups that own a CU by their LDS footprint (distance triangle > 80 KB: n >= 144,
        // TSP200).  Measured (outer iterations in 2 s, team vs serial): TSP200 x 256 13.0k vs 10.9k (weight guide), 10.5k vs
        // 10.3k (regret_pred of the synthetic model), 6.7k vs 6.8k (noise); but TSP100 x 256 on 8-wave workgroups 20.3k vs
        // 24.2k (model guide) and TSP50 x 128 (one pass per scan) 18.3k vs 21.2k: a round's barriers, slot exchange and
        // the re-evaluation after a move cost more than the few passes they save (profiles/r03_experiments/README.md)
        // Round 5: the edge form of the serial phase (best improvement; gls_kernels.hip) beats the team form there too -- TSP200 x 256,
        // outer iterations in 2 s, serial edge form vs team: 18.4k vs 15.1k (model guide), 18.8k vs 17.2k (weight) -- so the
        // policy keeps the team form for first-improvement runs only (which have no edge form): profiles/r05_experiments/
        const bool pays = first_improvement && batch > 0 && batch <= num_cus() && c.store == gnngls::GLS_STORE_COMPACT && c.threads == 1024;
        if (lds <= kLdsPerCU && (mode == 1 || pays)) { c.team = true; c.lds = lds; }
    }
    return c;
}

// + the pruned descent scans (nearest-neighbour lists: 2-opt scan from n = 80, relocate scan from n = 128; the position table
// sits in the LDS slot the best tour used to have, so the footprint does not change)
GlsConfig gls_config_run(int n, int requested_bits, int batch, bool first_improvement) {
    GlsConfig c = gls_config(n, requested_bits, batch, first_improvement);
    c.prune = g_prune_mode.load(std::memory_order_relaxed) != 0 && gnngls::gls_prune_supported(c.store, n, first_improvement, c.wps);
    return c;
}

GlsConfig gls_config_store(int n, int requested_bits, int batch, bool first_improvement) {
    GlsConfig pick{gnngls::GLS_STORE_GLOBAL, 32, gnngls::gls_block_threads(n, gnngls::GLS_STORE_GLOBAL),
                   gnngls::gls_lds_bytes(n, gnngls::GLS_STORE_GLOBAL, 32), 0, 4};
    bool have = false, done = false;
    const int cus = num_cus();
    // candidates are visited fastest store first (LDS penalties 32-bit, LDS penalties 16-bit, compact); the first one
    // that keeps the whole batch resident wins, otherwise the one with the most resident workgroups per CU
    auto by_lds_of = [](size_t lds) { return (int)(kLdsPerCU / lds); };
    auto consider = [&](int store, int bits) {
        if (done) return;
        size_t lds = gnngls::gls_lds_bytes(n, store, bits);
        if (lds > kLdsPerCU) return;
        // single-wavefront workgroups for n <= 33 only where the half-wave descent scans exist: best improvement, 128-VGPR build
        int threads = gnngls::gls_block_threads(n, store, bits, !first_improvement);
        // compact store, batch larger than the 128-VGPR build keeps resident at the default workgroup size: halve the
        // workgroup (down to the wavefronts the lean scans need, one per block of 64 rows) before falling back to the
        // 64-VGPR build -- TSP50 x 2048 on 2-wave workgroups at 128 VGPRs: 6.2k outer iterations per second vs 5.6k on
        // 4-wave workgroups at 64 VGPRs (profiles/r02_ab_small_n_threads.log)
        if (store == gnngls::GLS_STORE_COMPACT && batch > 0) {
            const int by_lds = (int)(kLdsPerCU / lds);
            const int min_threads = 64 * ((n - 1 + 63) / 64);
            while (threads > min_threads && threads > 64) {
                const int per4 = by_lds < 16 / (threads / 64) ? by_lds : 16 / (threads / 64);
                if ((long)per4 * cus >= batch || per4 == by_lds) break;
                threads /= 2;
            }
        }
        // wave slots per CU at the register budget of the kernel instantiation: LDS-penalty stores 80 VGPRs -> 6 waves
        // per SIMD (24 per CU); compact store 128 VGPRs -> 4 per SIMD, or its 64-VGPR build -> 8 per SIMD when only that
        // keeps the batch resident (and, without a batch size, for the capacity query)
        int wps = gnngls::gls_waves_per_simd(store, n, batch, cus, threads, lds);
        if (store == gnngls::GLS_STORE_COMPACT && batch <= 0) wps = 8;
        if (store == gnngls::GLS_STORE_TRI && bits == 16) wps = 6;            // the uint16 variant only exists as the 80-VGPR build
        // (n = 25..33 beyond the 128-VGPR residency end up on the 64-VGPR build, which has no half-wave scans, still as ONE
        // wavefront per instance: measured TSP30 x 8192, 1 s -- 64 threads keep all 8192 resident, 6.2k iterations each;
        // 128 threads halve the residency: 8.8k iterations each in two rounds of 1 s, half the aggregate rate
        // (profiles/r04_ab_threads_tsp30x8192.log))
        // batches of at most two single-wavefront workgroups per SIMD: the 256-VGPR build of the one-slot kernel
        // (not while the test hook forces the team form, which only exists on the 128-VGPR build)
        if (wps == 4 && batch > 0 && g_team_mode.load(std::memory_order_relaxed) != 1 &&
            gnngls::gls_wps2_supported(store, bits, n, threads, first_improvement)) {
            const int per2 = by_lds_of(lds) < 8 ? by_lds_of(lds) : 8;
            if ((long)per2 * cus >= batch) wps = 2;
        }
        const int by_waves = (wps * 4) / (threads / 64);
        int per_cu = (int)(kLdsPerCU / lds);
        if (per_cu > by_waves) per_cu = by_waves;
        if (!have || per_cu > pick.per_cu) { pick = GlsConfig{store, bits, threads, lds, per_cu, wps}; have = true; }
        if (batch > 0 && (long)per_cu * cus >= batch) { pick = GlsConfig{store, bits, threads, lds, per_cu, wps}; done = true; }
    };
    if (requested_bits == -2) {               // forced compact store (falls through to the global store if it cannot fit)
        if (n <= 255) consider(gnngls::GLS_STORE_COMPACT, 32);
        return pick;
    }
    if (requested_bits < 0) return pick;      // forced global-memory store (exact index order, asymmetric D allowed)
    if (requested_bits == 0 || requested_bits == 32) consider(gnngls::GLS_STORE_TRI, 32);
    // 16-bit LDS counters only on request: they overflow within a 10 s run when an uninformative guide
    // concentrates the penalties on a few edges, and an overflow costs a whole rerun of that instance
    if (requested_bits == 16) consider(gnngls::GLS_STORE_TRI, 16);
    if (requested_bits == 0 && n <= 255) consider(gnngls::GLS_STORE_COMPACT, 32);
    return pick;
}
}  // namespace

extern "C" {

int gnngls_abi_version(void) { return 4; }
const char *gnngls_last_error(void) { return g_err; }

int gnngls_gls_resident_capacity(int n) {
    if (n < 3) return 0;
    GlsConfig c = gls_config(n, 0);
    if (c.store == gnngls::GLS_STORE_GLOBAL) return 0;
    return c.per_cu * num_cus();
}

int gnngls_gls_describe_config(int n, int B, int penalty_bits, int *store, int *threads, int *lds_bytes, int *per_cu) {
    if (n < 3 || B < 0 || (penalty_bits != 0 && penalty_bits != 16 && penalty_bits != 32 && penalty_bits != -1 && penalty_bits != -2))
        return fail(GNNGLS_ERR_ARG, "gls_describe_config: bad argument");
    const GlsConfig c = gls_config_run(n, penalty_bits, B, false);     // as a best-improvement run would be launched
    if (c.lds > kLdsPerCU)
        return fail(GNNGLS_ERR_UNSUPPORTED, "gls_describe_config: n=%d needs %zu B of LDS for tours and edge lengths (> 160 KiB)", n, c.lds);
    if (store) *store = c.store * 100 + (c.store == gnngls::GLS_STORE_TRI ? c.penalty_bits : 0);
    if (threads) *threads = c.threads;
    if (lds_bytes) *lds_bytes = (int)c.lds;
    if (per_cu) *per_cu = c.store == gnngls::GLS_STORE_GLOBAL ? 0 : c.per_cu;
    return GNNGLS_OK;
}

int gnngls_gls_describe_run(int n, int B, int penalty_bits, int first_improvement, int *store, int *threads, int *lds_bytes,
                            int *per_cu, int *waves_per_simd, int *team, int *edge_form) {
    if (n < 3 || B < 0 || (penalty_bits != 0 && penalty_bits != 16 && penalty_bits != 32 && penalty_bits != -1 && penalty_bits != -2))
        return fail(GNNGLS_ERR_ARG, "gls_describe_run: bad argument");
    const GlsConfig c = gls_config_run(n, penalty_bits, B, first_improvement != 0);      // exactly what gnngls_gls_run launches
    if (c.lds > kLdsPerCU)
        return fail(GNNGLS_ERR_UNSUPPORTED, "gls_describe_run: n=%d needs %zu B of LDS for tours and edge lengths (> 160 KiB)", n, c.lds);
    if (store) *store = c.store * 100 + (c.store == gnngls::GLS_STORE_TRI ? c.penalty_bits : 0);
    if (threads) *threads = c.threads;
    if (lds_bytes) *lds_bytes = (int)c.lds;
    if (per_cu) *per_cu = c.store == gnngls::GLS_STORE_GLOBAL ? 0 : c.per_cu;
    if (waves_per_simd) *waves_per_simd = c.wps;
    if (team) *team = c.team ? 1 : 0;
    if (edge_form) *edge_form = gnngls::gls_edge_form(c.store, c.penalty_bits, c.wps, c.team, first_improvement != 0) ? 1 : 0;
    return GNNGLS_OK;
}

int gnngls_gls_kernel_resources(int n, int B, int penalty_bits, int first_improvement, int trace, int *vgprs, int *scratch_bytes) {
    if (n < 3 || B < 0 || (penalty_bits != 0 && penalty_bits != 16 && penalty_bits != 32 && penalty_bits != -1 && penalty_bits != -2))
        return fail(GNNGLS_ERR_ARG, "gls_kernel_resources: bad argument");
    const GlsConfig c = gls_config_run(n, penalty_bits, B, first_improvement != 0);
    gnngls::GlsArgs A;
    memset(&A, 0, sizeof(A));
    A.n = n; A.B = B;
    static double dummy;
    if (trace) { A.trace_cap = 1; A.trace_cost = &dummy; }                     // (selects the tracing instantiation; never dereferenced)
    if (c.prune) { A.nl_id = reinterpret_cast<const uint8_t *>(&dummy); }     // (likewise)
    const hipError_t e = gnngls::gls_kernel_resources(A, c.store, c.penalty_bits, c.threads, c.wps, c.team, first_improvement != 0, vgprs, scratch_bytes);
    return e == hipSuccess ? GNNGLS_OK : hip_fail(e, "gls_kernel_resources");
}

int gnngls_gls_waves_per_simd(int n, int B, int penalty_bits) {
    if (n < 3 || B < 0 || (penalty_bits != 0 && penalty_bits != 16 && penalty_bits != 32 && penalty_bits != -1 && penalty_bits != -2))
        return 0;
    return gls_config(n, penalty_bits, B).wps;
}

int gnngls_gls_uses_team(int n, int B, int penalty_bits) {
    if (n < 3 || B < 0 || (penalty_bits != 0 && penalty_bits != 16 && penalty_bits != 32 && penalty_bits != -1 && penalty_bits != -2))
        return 0;
    return gls_config(n, penalty_bits, B).team ? 1 : 0;
}

int gnngls_two_opt_delta_all(const int32_t *tour, const double *D, int B, int n, double *out, void *stream) {
    if (B == 0) return GNNGLS_OK;   // empty batch: nothing to enqueue (data pointers may be NULL)
    if (!tour || !D || !out || B < 0 || n < 3) return fail(GNNGLS_ERR_ARG, "two_opt_delta_all: bad argument");
    hipError_t e = gnngls::launch_delta_all(tour, D, B, n, 0, out, (hipStream_t)stream);
    return e == hipSuccess ? GNNGLS_OK : hip_fail(e, "two_opt_delta_all");
}

int gnngls_relocate_delta_all(const int32_t *tour, const double *D, int B, int n, double *out, void *stream) {
    if (B == 0) return GNNGLS_OK;   // empty batch: nothing to enqueue (data pointers may be NULL)
    if (!tour || !D || !out || B < 0 || n < 3) return fail(GNNGLS_ERR_ARG, "relocate_delta_all: bad argument");
    hipError_t e = gnngls::launch_delta_all(tour, D, B, n, 1, out, (hipStream_t)stream);
    return e == hipSuccess ? GNNGLS_OK : hip_fail(e, "relocate_delta_all");
}

int gnngls_best_move(const int32_t *tour, const double *D, int B, int n, int op, const int32_t *pos_i,
                     int first_improvement, double *delta_out, int32_t *move_out, int32_t *new_tour, void *stream) {
    if (B == 0) return GNNGLS_OK;   // empty batch: nothing to enqueue (data pointers may be NULL)
    if (!tour || !D || !delta_out || !move_out || B < 0 || n < 3 || (op != 0 && op != 1) || n > 65535)
        return fail(GNNGLS_ERR_ARG, "best_move: bad argument");
    hipError_t e = gnngls::launch_best_move(tour, D, B, n, op, pos_i, first_improvement != 0, delta_out, move_out,
                                            new_tour, (hipStream_t)stream);
    return e == hipSuccess ? GNNGLS_OK : hip_fail(e, "best_move");
}

int gnngls_tour_cost(const int32_t *tour, const double *D, int B, int n, double *cost_out, void *stream) {
    if (B == 0) return GNNGLS_OK;   // empty batch: nothing to enqueue (data pointers may be NULL)
    if (!tour || !D || !cost_out || B < 0 || n < 1) return fail(GNNGLS_ERR_ARG, "tour_cost: bad argument");
    ProfScope ps(GNNGLS_PROF_TOUR_COST, (hipStream_t)stream);
    hipError_t e = gnngls::launch_tour_cost(tour, D, B, n, cost_out, (hipStream_t)stream);
    return e == hipSuccess ? GNNGLS_OK : hip_fail(e, "tour_cost");
}

int gnngls_nearest_neighbor(const double *W, int B, int n, int depot, int32_t *tour_out, void *stream) {
    if (B == 0) return GNNGLS_OK;   // empty batch: nothing to enqueue (data pointers may be NULL)
    if (!W || !tour_out || B < 0 || n < 1 || depot < 0 || depot >= n)
        return fail(GNNGLS_ERR_ARG, "nearest_neighbor: bad argument");
    ProfScope ps(GNNGLS_PROF_NEAREST_NEIGHBOR, (hipStream_t)stream);
    hipError_t e = gnngls::launch_nearest_neighbor(W, B, n, depot, tour_out, (hipStream_t)stream);
    return e == hipSuccess ? GNNGLS_OK : hip_fail(e, "nearest_neighbor");
}

int gnngls_gls_run(const double *D, const double *guides, int n_guides, int B, int n,
                   const int32_t *init_tour, const double *init_cost,
                   int perturbation_moves, int first_improvement, int penalty_bits,
                   int64_t max_outer_iters, double time_limit_s, double watchdog_s,
                   int32_t *best_tour, double *best_cost, int64_t *outer_iters,
                   double *trace_cost, float *trace_time, int trace_cap, int32_t *trace_len,
                   int32_t *penalty_out, int64_t *evals_out, int32_t *status,
                   double *imp_cost, float *imp_time, int64_t *imp_iter, int imp_cap, int32_t *imp_len, void *stream) {
    if (B == 0) return GNNGLS_OK;   // empty batch: nothing to enqueue (data pointers may be NULL)
    if (!D || !init_tour || !init_cost || !best_tour || !best_cost || B < 0 || n < 3 || n > 65535 || trace_cap < 0)
        return fail(GNNGLS_ERR_ARG, "gls_run: bad argument");
    if (imp_cap < 0 || ((imp_cost || imp_time || imp_iter) && !imp_len))
        return fail(GNNGLS_ERR_ARG, "gls_run: improvement trace needs imp_len and imp_cap >= 0");
    if (max_outer_iters != 0 && (!guides || n_guides < 1))
        return fail(GNNGLS_ERR_ARG, "gls_run: guides required when outer iterations are requested");
    if (!(watchdog_s > 0.0)) return fail(GNNGLS_ERR_ARG, "gls_run: watchdog_s must be > 0");
    if (penalty_bits != 0 && penalty_bits != 16 && penalty_bits != 32 && penalty_bits != -1 && penalty_bits != -2)
        return fail(GNNGLS_ERR_ARG, "gls_run: penalty_bits must be 0 (auto), 16, 32, -1 (global-memory store) or -2 (compact store)");
    hipStream_t st = (hipStream_t)stream;
    gnngls::GlsArgs A;
    memset(&A, 0, sizeof(A));
    A.D = D; A.guides = guides ? guides : D; A.n_guides = n_guides > 0 ? n_guides : 1; A.B = B; A.n = n;
    A.init_tour = init_tour; A.init_cost = init_cost;
    A.perturbation_moves = perturbation_moves;
    A.max_outer_iters = max_outer_iters; A.time_limit_s = time_limit_s; A.watchdog_s = watchdog_s;
    A.best_tour = best_tour; A.best_cost = best_cost; A.outer_iters = (long long *)outer_iters;
    A.trace_cost = trace_cost; A.trace_time = trace_time; A.trace_cap = trace_cost ? trace_cap : 0;
    A.pen16_limit = g_pen16_limit;
    A.imp_cost = imp_cost; A.imp_time = imp_time; A.imp_iter = (long long *)imp_iter; A.imp_cap = imp_cap; A.imp_len = imp_len;
    A.stamps = g_stamp_buffer;
    A.evals_exec = g_exec_evals.load(std::memory_order_relaxed);
    A.trace_len = trace_len; A.penalty_out = penalty_out; A.evals = (long long *)evals_out; A.status = status;
    const GlsConfig cfg = gls_config_run(n, penalty_bits, B, first_improvement != 0);
    if (cfg.lds > kLdsPerCU)     // even the global-memory store keeps tours and per-position edge lengths in LDS
        return fail(GNNGLS_ERR_UNSUPPORTED, "gls_run: n=%d needs %zu B of LDS for tours and edge lengths (> 160 KiB)", n, cfg.lds);
    int32_t *ws = nullptr;
    if (cfg.store != gnngls::GLS_STORE_TRI) {
        // penalties in global memory (zeroed workspace): full matrices for the global store, packed
        // triangles for the compact store
        // global store: int32 [n,n] per instance; compact store: int32 packed triangle
        // (the team form of the compact store keeps a full symmetric matrix too: row-contiguous reads, see TriDGlobalPF)
        size_t per = (cfg.store == gnngls::GLS_STORE_GLOBAL || cfg.team) ? (size_t)n * n : (size_t)n * (n - 1) / 2;
        size_t bytes = (size_t)B * per * sizeof(int32_t);
        hipError_t e = hipMallocAsync((void **)&ws, bytes, st);
        if (e != hipSuccess) return hip_fail(e, "gls_run: workspace alloc");
        e = hipMemsetAsync(ws, 0, bytes, st);
        if (e != hipSuccess) return hip_fail(e, "gls_run: workspace memset");
        A.pen_ws = ws;
    }
    hipError_t e;
    if (A.evals_exec) {          // the wavefronts of an instance add their counts atomically
        // a run that prunes on an instantiation without the counting code cannot report the executed evaluations: -1
        const bool unknown = cfg.prune && !gnngls::gls_count_supported(cfg.store, cfg.wps, n, first_improvement != 0, A.trace_cap > 0);
        e = hipMemsetAsync(A.evals_exec, 0, (size_t)GNNGLS_EXEC_RECORDS * B * sizeof(long long), st);       // counts + the cycle records
        if (e == hipSuccess && unknown) e = hipMemsetAsync(A.evals_exec, 0xff, (size_t)B * sizeof(long long), st);
        if (e != hipSuccess) { if (ws) (void)hipFreeAsync(ws, st); return hip_fail(e, "gls_run: executed-evaluations buffer"); }
        if (unknown) A.evals_exec = nullptr;
    }
    int32_t *asym = nullptr;     // symmetric stores keep D[max, min] only: instances with an asymmetric matrix are flagged, not searched
    if (cfg.store != gnngls::GLS_STORE_GLOBAL) {
        e = hipMallocAsync((void **)&asym, (size_t)B * sizeof(int32_t), st);
        if (e == hipSuccess) e = gnngls::launch_symmetry_check(D, B, n, asym, st);
        if (e != hipSuccess) { if (asym) (void)hipFreeAsync(asym, st); if (ws) (void)hipFreeAsync(ws, st); return hip_fail(e, "gls_run: symmetry check"); }
        A.asym = asym;
    }
    void *nl = nullptr;          // nearest-neighbour lists of the pruned descent scans
    if (cfg.prune) {
        const size_t entries = (size_t)B * n * gnngls::kNeighborListLen;
        e = hipMallocAsync(&nl, (size_t)B * sizeof(int32_t) + entries, st);
        if (e != hipSuccess) { if (asym) (void)hipFreeAsync(asym, st); if (ws) (void)hipFreeAsync(ws, st); return hip_fail(e, "gls_run: neighbour-list alloc"); }
        int32_t *ok = (int32_t *)nl;
        uint8_t *nl_id = (uint8_t *)(ok + B);
        e = gnngls::launch_neighbor_lists(D, B, n, nl_id, ok, st);
        if (e != hipSuccess) { (void)hipFreeAsync(nl, st); if (asym) (void)hipFreeAsync(asym, st); if (ws) (void)hipFreeAsync(ws, st); return hip_fail(e, "gls_run: neighbour lists"); }
        A.nl_id = nl_id; A.prune_ok = ok;
    }
    {
        ProfScope ps(GNNGLS_PROF_GLS, st);
        e = gnngls::launch_gls(A, cfg.store, cfg.penalty_bits, cfg.threads, cfg.wps, cfg.team, first_improvement != 0, st);
    }
    if (nl) (void)hipFreeAsync(nl, st);
    if (asym) (void)hipFreeAsync(asym, st);
    if (ws) (void)hipFreeAsync(ws, st);
    return e == hipSuccess ? GNNGLS_OK : hip_fail(e, "gls_run");
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// GNN forward
// ---------------------------------------------------------------------------------------------
namespace {
constexpr long kLayerFloats = 128L * 128 + 128 + 128 + 128 + 128 + 512L * 128 + 512 + 128L * 512 + 128 + 128 + 128;
constexpr long kBytesPerNode = (128 + 128 + 256 + 32 + 128) * 4L;   // h, ft, part, part_ms, h2 (ping-pong)
}  // namespace

extern "C" {

int64_t gnngls_model_packed_floats(int in_dim, int n_layers) {
    if (in_dim < 0 || n_layers < 0) return 0;
    return 128L * in_dim + 128 + (long)n_layers * kLayerFloats + 128 + 4;
}

int64_t gnngls_regret_forward_workspace_bytes(int B, int n) {
    if (B < 1 || n < 2 || n > 65535) return 0;          // 65535 nodes x 2^31 instances still fits an int64
    long N = (long)n * (n - 1) / 2;
    return (int64_t)B * N * kBytesPerNode + 256;
}

int64_t gnngls_regret_prepared_bytes(int n_layers) {
    if (n_layers < 0) return 0;
    // per layer the feed-forward weights in bf16 pieces; behind them A = We^T Wfc^T and b' = Wfc be of the fused embed + first fc
    return (int64_t)n_layers * (int64_t)gnngls::ffn_packed_bytes() + (int64_t)gnngls::embed_fc_bytes() + 256;
}

int gnngls_regret_prepare(const float *weights, int in_dim, int n_layers, void *prepared, int64_t prepared_bytes, void *stream) {
    if (!weights || !prepared || in_dim < 1 || n_layers < 0) return fail(GNNGLS_ERR_ARG, "regret_prepare: bad argument");
    if (prepared_bytes < gnngls_regret_prepared_bytes(n_layers))
        return fail(GNNGLS_ERR_ARG, "regret_prepare: buffer too small (%lld B, need %lld B)", (long long)prepared_bytes,
                    (long long)gnngls_regret_prepared_bytes(n_layers));
    unsigned char *base = (unsigned char *)(((uintptr_t)prepared + 255) & ~(uintptr_t)255);
    const float *layers = weights + 128L * in_dim + 128;
    for (int l = 0; l < n_layers; ++l) {
        const float *w = layers + (long)l * kLayerFloats;
        const float *w1 = w + 128L * 128 + 4 * 128, *w2 = w1 + 512L * 128 + 512;
        const float *fc_next = l + 1 < n_layers ? layers + (long)(l + 1) * kLayerFloats : nullptr;
        hipError_t e = gnngls::launch_ffn_pack(w1, w2, fc_next, base + (size_t)l * gnngls::ffn_packed_bytes(), (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(e, "regret_prepare");
    }
    if (n_layers > 0 && in_dim <= gnngls::embed_fc_max_in_dim()) {
        hipError_t e = gnngls::launch_embed_fc_prepare(weights, weights + 128L * in_dim, layers, layers + 128L * 128, layers + 128L * 128 + 128,
                                                       in_dim, base + (size_t)n_layers * gnngls::ffn_packed_bytes(), (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(e, "regret_prepare");
    }
    return GNNGLS_OK;
}

int gnngls_regret_forward_prepared(const float *feat, const float *weights, const void *prepared, int64_t prepared_bytes,
                                   int B, int n, int in_dim, int n_layers,
                                   float *y_out, void *workspace, int64_t workspace_bytes, void *stream) {
    if (B == 0) return GNNGLS_OK;   // empty batch: nothing to enqueue (data pointers may be NULL)
    if (!feat || !weights || !y_out || !workspace || B < 0 || n < 3 || in_dim < 1 || n_layers < 0)
        return fail(GNNGLS_ERR_ARG, "regret_forward: bad argument");
    if ((in_dim * 128) % 4 != 0) return fail(GNNGLS_ERR_UNSUPPORTED, "regret_forward: in_dim*128 must be a multiple of 4");
    if (gnngls::gat_rows_lds_bytes(n) > kLdsPerCU)
        return fail(GNNGLS_ERR_UNSUPPORTED, "regret_forward: n=%d needs %zu B of LDS per row tile (> 160 KiB)", n,
                    gnngls::gat_rows_lds_bytes(n));
    // prepared == NULL keeps the feed-forward block on the fp32 matrix pipe (as GNNGLS_FFN_FP32=1 does: A/B runs)
    static const bool ffn_fp32 = getenv("GNNGLS_FFN_FP32") && atoi(getenv("GNNGLS_FFN_FP32")) != 0;
    if (prepared && prepared_bytes < gnngls_regret_prepared_bytes(n_layers))
        return fail(GNNGLS_ERR_ARG, "regret_forward: prepared image too small (%lld B, need %lld B)", (long long)prepared_bytes,
                    (long long)gnngls_regret_prepared_bytes(n_layers));
    const unsigned char *prep = (prepared && !ffn_fp32) ? (const unsigned char *)(((uintptr_t)prepared + 255) & ~(uintptr_t)255) : nullptr;
    const long N = (long)n * (n - 1) / 2;
    uintptr_t base = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
    int64_t avail = workspace_bytes - (int64_t)(base - (uintptr_t)workspace);
    long Bc = avail / (N * kBytesPerNode);
    if (Bc < 1) return fail(GNNGLS_ERR_ARG, "regret_forward: workspace too small (%lld B, need >= %lld B)",
                            (long long)workspace_bytes, (long long)gnngls_regret_forward_workspace_bytes(1, n));
    if (Bc > B) Bc = B;
    hipStream_t st = (hipStream_t)stream;
    const long Mc = Bc * N;
    float *h = (float *)base;
    float *ft = h + Mc * 128;
    float *part = ft + Mc * 128;
    float *part_ms = part + 2 * Mc * 128;
    float *h2 = part_ms + 2 * Mc * 16;
    const float *emb_w = weights, *emb_b = weights + 128L * in_dim;
    const float *layers = emb_b + 128;
    const float *dec_w = layers + (long)n_layers * kLayerFloats, *dec_b = dec_w + 128;
    hipError_t e = hipSuccess;
#define GNNGLS_TRY(x) do { e = (x); if (e != hipSuccess) return hip_fail(e, #x); } while (0)
    for (long b0 = 0; b0 < B; b0 += Bc) {
        const int bc = (int)((B - b0) < Bc ? (B - b0) : Bc);
        const long M = (long)bc * N;
        // models.py:66; with a prepared image also ft = fc(h) of layer 0 (models.py:23), both straight from the input features
        const bool fused_fc0 = prep && n_layers > 0 && in_dim <= gnngls::embed_fc_max_in_dim();
        static const bool no_rank1 = getenv("GNNGLS_GAT_RANK1") && atoi(getenv("GNNGLS_GAT_RANK1")) == 0;      // (A/B runs)
        const bool rank1_gat0 = fused_fc0 && in_dim == 1 && n <= 255 && !no_rank1;
        // ... and then the first feed-forward launch forms its input from the one feature and the compact partials (LR0): no embedding
        // pass at all, neither h_0 nor ft_0 nor 128-wide partials of the first layer in memory (GNNGLS_GAT_RANK1=2: keep them, A/B runs)
        static const bool keep_h0 = getenv("GNNGLS_GAT_RANK1") && atoi(getenv("GNNGLS_GAT_RANK1")) == 2;
        const bool lr0 = rank1_gat0 && !keep_h0;
        const float *img0 = (const float *)(prep ? prep + (size_t)n_layers * gnngls::ffn_packed_bytes() : nullptr);
        if (!lr0) { ProfScope ps(GNNGLS_PROF_EMBED, st);
          // (one input feature: the first GATConv runs in its rank-1 form below and no ft is written)
          if (fused_fc0) GNNGLS_TRY(gnngls::launch_embed_fc(feat + b0 * N * in_dim, emb_w, emb_b, prep + (size_t)n_layers * gnngls::ffn_packed_bytes(), h,
                                                            rank1_gat0 ? nullptr : ft, M, in_dim, st));
          else GNNGLS_TRY(gnngls::launch_embed(feat + b0 * N * in_dim, emb_w, emb_b, h, M, in_dim, st)); }
        for (int l = 0; l < n_layers; ++l) {                                                          // models.py:67-68
            const float *w = layers + (long)l * kLayerFloats;
            const float *fc_w = w, *attn_l = fc_w + 128L * 128, *attn_r = attn_l + 128;
            const float *bn1_s = attn_r + 128, *bn1_b = bn1_s + 128;
            const float *w1 = bn1_b + 128, *b1 = w1 + 512L * 128, *w2 = b1 + 512, *b2 = w2 + 128L * 512;
            const float *bn2_s = b2 + 128, *bn2_b = bn2_s + 128;
            // ft = fc(h), models.py:23: a launch of its own for the first layer (and for every layer on the fp32 path); on the bf16x3 path
            // the feed-forward launch of layer l - 1 has already written it (fc folded into that kernel's tail)
            if ((l == 0 && !fused_fc0) || !prep) {
              ProfScope ps(GNNGLS_PROF_GEMM_FC, st);
              GNNGLS_TRY(gnngls::launch_gemm(gnngls::GEMM_EPI_STORE, h, fc_w, ft, M, 128, 128, nullptr, nullptr, nullptr, nullptr, st)); }
            if (l == 0 && rank1_gat0) {
              ProfScope ps(GNNGLS_PROF_GAT_ROWS_RANK1, st);
              GNNGLS_TRY(gnngls::launch_gat_rows_rank1(feat + b0 * N, prep + (size_t)n_layers * gnngls::ffn_packed_bytes(), bc, n, part, part_ms, st, lr0));
            } else {
              ProfScope ps(GNNGLS_PROF_GAT_ROWS, st);
              GNNGLS_TRY(gnngls::launch_gat_rows(ft, attn_l, attn_r, bc, n, part, part_ms, st)); }
            // gat_combine + FFN1 + FFN2 (+ the next layer's fc) in one launch; the hidden layer and x = BN1(h + GAT) never touch HBM
            { ProfScope ps(GNNGLS_PROF_FFN_FUSED, st);
              // (the last layer's launch also applies the decision layer, models.py:69: its output is never stored)
              const bool last = prep && l + 1 == n_layers;
              const bool lr = l == 0 && lr0;
              GNNGLS_TRY(gnngls::launch_ffn_fused(part, part_ms, lr ? feat + b0 * N : h, bn1_s, bn1_b, w1, b1, w2, b2, bn2_s, bn2_b, h2, M,
                                                  prep ? prep + (size_t)l * gnngls::ffn_packed_bytes() : nullptr, prep && l + 1 < n_layers, ft, st,
                                                  last ? dec_w : nullptr, last ? dec_b : nullptr, last ? y_out + b0 * N : nullptr,
                                                  lr ? img0 : nullptr, lr ? emb_w : nullptr, lr ? emb_b : nullptr)); }
            { float *x = h; h = h2; h2 = x; }
        }
        if (!(prep && n_layers > 0)) {
          ProfScope ps(GNNGLS_PROF_DECISION, st);
          GNNGLS_TRY(gnngls::launch_decision(h, dec_w, dec_b, y_out + b0 * N, M, st)); }               // models.py:69
    }
#undef GNNGLS_TRY
    return GNNGLS_OK;
}

// The one-call form: splits the weights into stream-ordered scratch of its own on every call (gnngls_regret_prepare +
// gnngls_regret_forward_prepared keep the image across calls: 2 launches per layer and the allocation saved per forward).
int gnngls_regret_forward(const float *feat, const float *weights, int B, int n, int in_dim, int n_layers,
                          float *y_out, void *workspace, int64_t workspace_bytes, void *stream) {
    if (B == 0) return GNNGLS_OK;
    if (!feat || !weights || !y_out || !workspace || B < 0 || n < 3 || in_dim < 1 || n_layers < 0)
        return fail(GNNGLS_ERR_ARG, "regret_forward: bad argument");
    hipStream_t st = (hipStream_t)stream;
    struct StreamScratch {
        void *p = nullptr; hipStream_t st;
        ~StreamScratch() { if (p) (void)hipFreeAsync(p, st); }
    } ffn_ws;
    ffn_ws.st = st;
    static const bool ffn_fp32 = getenv("GNNGLS_FFN_FP32") && atoi(getenv("GNNGLS_FFN_FP32")) != 0;
    const int64_t pb = gnngls_regret_prepared_bytes(n_layers);
    if (n_layers > 0 && !ffn_fp32) {
        hipError_t e = hipMallocAsync(&ffn_ws.p, (size_t)pb, st);
        if (e != hipSuccess) { ffn_ws.p = nullptr; return hip_fail(e, "regret_forward: scratch alloc"); }
        const int rc = gnngls_regret_prepare(weights, in_dim, n_layers, ffn_ws.p, pb, stream);
        if (rc != GNNGLS_OK) return rc;
    }
    return gnngls_regret_forward_prepared(feat, weights, ffn_ws.p, pb, B, n, in_dim, n_layers, y_out, workspace, workspace_bytes, stream);
}

int gnngls_pack_features(const double *D, int B, int n, double scale, double min_, float *feat, void *stream) {
    if (B == 0) return GNNGLS_OK;   // empty batch: nothing to enqueue (data pointers may be NULL)
    if (!D || !feat || B < 0 || n < 2) return fail(GNNGLS_ERR_ARG, "pack_features: bad argument");
    ProfScope ps(GNNGLS_PROF_PACK, (hipStream_t)stream);
    hipError_t e = gnngls::launch_pack_features(D, B, n, scale, min_, feat, (hipStream_t)stream);
    return e == hipSuccess ? GNNGLS_OK : hip_fail(e, "pack_features");
}

int gnngls_unpack_regret(const float *y, int B, int n, double scale, double min_, double *out, void *stream) {
    if (B == 0) return GNNGLS_OK;   // empty batch: nothing to enqueue (data pointers may be NULL)
    if (!y || !out || B < 0 || n < 2) return fail(GNNGLS_ERR_ARG, "unpack_regret: bad argument");
    ProfScope ps(GNNGLS_PROF_UNPACK, (hipStream_t)stream);
    hipError_t e = gnngls::launch_unpack_regret(y, B, n, scale, min_, out, (hipStream_t)stream);
    return e == hipSuccess ? GNNGLS_OK : hip_fail(e, "unpack_regret");
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// GNN training step (forward with batch-statistics BatchNorm + backward)
// ---------------------------------------------------------------------------------------------
namespace {

// Device workspace of one training step for M = B*N rows and L layers.  The first group is written by the forward and
// read by the backward (it must survive between the two calls); the second group is scratch.
struct TrainWs {
    float *H;        // [(L+1)][M][128]  layer inputs (H[0] = embedding) and the final hidden state
    float *FT;       // [L][M][128]      fc(h)
    float *G;        // [L][M][128]      GATConv output
    float *H1;       // [L][M][128]      h + GATConv(h)
    float *H3;       // [L][M][128]      x + MLP(x), x = BN1(h1)
    float *HID;      // [L][M][512]      ReLU(W1 x + b1); overwritten by its gradient in the backward
    float *ATT;      // [L][M][16]       softmax statistics (row max, 1/Z) per head
    float *BN;       // [L][8][128]      mean1 invstd1 scale1 shift1 mean2 invstd2 scale2 shift2
    float *PART;     // [2][M][128]      attention partials (forward) / P partials (backward)
    float *PMS;      // [2][M][16]
    float *DA, *DB;  // [M][128]         gradient ping-pong
    float *X2;       // [M][128]         BN1 output recomputed in the backward
    float *DFT;      // [M][128]
    float *DLR;      // [2][M][8]        d el, d er
    float *WT;       // [2][512*128]     W2^T, W1^T of the layer being differentiated
    float *COEF;     // [3][128] + ones[128] + zeros[512]   BatchNorm backward coefficients, constants
    double *CSP;     // column-sum partials
    float *TNP;      // weight-gradient partial tiles
    size_t bytes;
};

TrainWs train_layout(uintptr_t base, long M, int L) {
    TrainWs w;
    uintptr_t p = (base + 255) & ~(uintptr_t)255;
    auto take = [&](size_t bytes) { uintptr_t q = p; p = (p + bytes + 255) & ~(uintptr_t)255; return q; };
    const size_t row = (size_t)M * 128 * sizeof(float);
    w.H = (float *)take(row * (L + 1));
    w.FT = (float *)take(row * L);
    w.G = (float *)take(row * L);
    w.H1 = (float *)take(row * L);
    w.H3 = (float *)take(row * L);
    w.HID = (float *)take(4 * row * L);
    w.ATT = (float *)take((size_t)M * 16 * sizeof(float) * L);
    w.BN = (float *)take((size_t)L * 8 * 128 * sizeof(float));
    w.PART = (float *)take(2 * row);
    w.PMS = (float *)take((size_t)2 * M * 16 * sizeof(float));
    w.DA = (float *)take(row);
    w.DB = (float *)take(row);
    w.X2 = (float *)take(row);
    w.DFT = (float *)take(row);
    w.DLR = (float *)take((size_t)2 * M * 8 * sizeof(float));
    w.WT = (float *)take((size_t)2 * 512 * 128 * sizeof(float));
    w.COEF = (float *)take((4 * 128 + 512) * sizeof(float));
    w.CSP = (double *)take((size_t)gnngls::kColsumMaxBlocks * 2 * 512 * sizeof(double));
    w.TNP = (float *)take((size_t)gnngls::gemm_tn_chunks(M) * (128 * 512 + 512) * sizeof(float));
    w.bytes = (size_t)(p - base);
    return w;
}

struct LayerParams {
    const float *fc_w, *attn_l, *attn_r, *bn1_g, *bn1_b, *w1, *b1, *w2, *b2, *bn2_g, *bn2_b;
};

template <typename T>
void layer_pointers(T *w, T *&fc_w, T *&attn_l, T *&attn_r, T *&bn1_g, T *&bn1_b, T *&w1, T *&b1, T *&w2, T *&b2, T *&bn2_g,
                    T *&bn2_b) {
    fc_w = w; attn_l = fc_w + 128L * 128; attn_r = attn_l + 128; bn1_g = attn_r + 128; bn1_b = bn1_g + 128;
    w1 = bn1_b + 128; b1 = w1 + 512L * 128; w2 = b1 + 512; b2 = w2 + 128L * 512; bn2_g = b2 + 128; bn2_b = bn2_g + 128;
}

int train_check(const char *what, const void *feat, const void *params, const void *io, const void *workspace, int B, int n,
                int in_dim, int n_layers, int64_t workspace_bytes) {
    if (!feat || !params || !io || !workspace || B < 1 || n < 3 || in_dim < 1 || n_layers < 0)
        return fail(GNNGLS_ERR_ARG, "%s: bad argument", what);
    if (n > gnngls::gat_bwd_max_nodes())
        return fail(GNNGLS_ERR_UNSUPPORTED, "%s: n=%d exceeds the attention-backward tile limit (n <= %d)", what, n,
                    gnngls::gat_bwd_max_nodes());
    if (gnngls::gat_rows_lds_bytes(n) > kLdsPerCU || gnngls::gat_bwd_lds_bytes(n) > kLdsPerCU)
        return fail(GNNGLS_ERR_UNSUPPORTED, "%s: n=%d needs %zu B of LDS per row tile (> 160 KiB)", what, n,
                    gnngls::gat_bwd_lds_bytes(n));
    const int64_t need = gnngls_regret_train_workspace_bytes(B, n, n_layers);
    if (workspace_bytes < need)
        return fail(GNNGLS_ERR_ARG, "%s: workspace too small (%lld B, need %lld B)", what, (long long)workspace_bytes,
                    (long long)need);
    return GNNGLS_OK;
}

}  // namespace

extern "C" {

int64_t gnngls_regret_train_workspace_bytes(int B, int n, int n_layers) {
    if (B < 1 || n < 2 || n > 65535 || n_layers < 0 || n_layers > 4096) return 0;
    const long M = (long)B * ((long)n * (n - 1) / 2);
    return (int64_t)train_layout(0, M, n_layers).bytes + 256;
}

#define GNNGLS_TRY(x) do { e = (x); if (e != hipSuccess) return hip_fail(e, #x); } while (0)

int gnngls_regret_train_forward(const float *feat, const float *params, int B, int n, int in_dim, int n_layers, float bn_eps,
                                float *y_out, float *bn_batch_stats, void *workspace, int64_t workspace_bytes, void *stream) {
    int rc = train_check("regret_train_forward", feat, params, y_out, workspace, B, n, in_dim, n_layers, workspace_bytes);
    if (rc != GNNGLS_OK) return rc;
    if (!bn_batch_stats && n_layers > 0) return fail(GNNGLS_ERR_ARG, "regret_train_forward: bn_batch_stats is NULL");
    hipStream_t st = (hipStream_t)stream;
    const long N = (long)n * (n - 1) / 2, M = (long)B * N;
    const TrainWs w = train_layout((uintptr_t)workspace, M, n_layers);
    const float *emb_w = params, *emb_b = params + 128L * in_dim, *layers = emb_b + 128;
    const float *dec_w = layers + (long)n_layers * kLayerFloats, *dec_b = dec_w + 128;
    float *ones = w.COEF + 3 * 128, *zeros = w.COEF + 4 * 128;
    hipError_t e = hipSuccess;
    const float one = 1.f;
    uint32_t one_bits;
    memcpy(&one_bits, &one, 4);
    GNNGLS_TRY(hipMemsetD32Async((hipDeviceptr_t)ones, (int)one_bits, 128, st));
    GNNGLS_TRY(hipMemsetAsync(zeros, 0, 512 * sizeof(float), st));
    const size_t row = (size_t)M * 128;
    { ProfScope ps(GNNGLS_PROF_EMBED, st);
      GNNGLS_TRY(gnngls::launch_embed(feat, emb_w, emb_b, w.H, M, in_dim, st)); }                       // models.py:66
    for (int l = 0; l < n_layers; ++l) {                                                                // models.py:67-68
        const float *fc_w, *attn_l, *attn_r, *bn1_g, *bn1_b, *w1, *b1, *w2, *b2, *bn2_g, *bn2_b;
        layer_pointers(layers + (long)l * kLayerFloats, fc_w, attn_l, attn_r, bn1_g, bn1_b, w1, b1, w2, b2, bn2_g, bn2_b);
        const float *h = w.H + row * l;
        float *ft = w.FT + row * l, *g = w.G + row * l, *h1 = w.H1 + row * l, *h3 = w.H3 + row * l;
        float *att = w.ATT + (size_t)M * 16 * l, *bn = w.BN + (size_t)l * 8 * 128;
        float *stats = bn_batch_stats + (size_t)l * 4 * 128;
        int nb = 0;
        { ProfScope ps(GNNGLS_PROF_GEMM_FC, st);
          GNNGLS_TRY(gnngls::launch_gemm(gnngls::GEMM_EPI_STORE, h, fc_w, ft, M, 128, 128, nullptr, nullptr, nullptr, nullptr, st)); }
        { ProfScope ps(GNNGLS_PROF_GAT_ROWS, st);
          GNNGLS_TRY(gnngls::launch_gat_rows(ft, attn_l, attn_r, B, n, w.PART, w.PMS, st)); }
        { ProfScope ps(GNNGLS_PROF_TRAIN_ELEMENTWISE, st);
          GNNGLS_TRY(gnngls::launch_gat_combine_train(w.PART, w.PMS, h, M, g, h1, att, st)); }          // models.py:12-15
        { ProfScope ps(GNNGLS_PROF_TRAIN_COLSUM, st);                                                   // models.py:27 (train mode)
          GNNGLS_TRY(gnngls::launch_colsum(gnngls::CS_SUM_SQ, h1, nullptr, nullptr, M, 128, 0, w.CSP, &nb, st));
          GNNGLS_TRY(gnngls::launch_bn_stats_finalize(w.CSP, nb, M, bn1_g, bn1_b, bn_eps, bn + 2 * 128, bn + 3 * 128, bn,
                                                      bn + 128, stats, stats + 128, st)); }
        { ProfScope ps(GNNGLS_PROF_FFN_FUSED, st);                                                      // models.py:28-33
          GNNGLS_TRY(gnngls::launch_ffn_fused_train(h1, bn + 2 * 128, bn + 3 * 128, w1, b1, w2, b2, ones, zeros, h3,
                                                    w.HID + 4 * row * l, M, st)); }
        { ProfScope ps(GNNGLS_PROF_TRAIN_COLSUM, st);                                                   // models.py:35 (train mode)
          GNNGLS_TRY(gnngls::launch_colsum(gnngls::CS_SUM_SQ, h3, nullptr, nullptr, M, 128, 0, w.CSP, &nb, st));
          GNNGLS_TRY(gnngls::launch_bn_stats_finalize(w.CSP, nb, M, bn2_g, bn2_b, bn_eps, bn + 6 * 128, bn + 7 * 128,
                                                      bn + 4 * 128, bn + 5 * 128, stats + 256, stats + 384, st)); }
        { ProfScope ps(GNNGLS_PROF_TRAIN_ELEMENTWISE, st);
          GNNGLS_TRY(gnngls::launch_affine_cols(h3, bn + 6 * 128, bn + 7 * 128, w.H + row * (l + 1), M, st)); }
    }
    { ProfScope ps(GNNGLS_PROF_DECISION, st);
      GNNGLS_TRY(gnngls::launch_decision(w.H + row * n_layers, dec_w, dec_b, y_out, M, st)); }          // models.py:69
    return GNNGLS_OK;
}

int gnngls_regret_train_backward(const float *feat, const float *params, const float *dy, int B, int n, int in_dim,
                                 int n_layers, float *grads, void *workspace, int64_t workspace_bytes, void *stream) {
    int rc = train_check("regret_train_backward", feat, params, grads, workspace, B, n, in_dim, n_layers, workspace_bytes);
    if (rc != GNNGLS_OK) return rc;
    if (!dy) return fail(GNNGLS_ERR_ARG, "regret_train_backward: dy is NULL");
    hipStream_t st = (hipStream_t)stream;
    const long N = (long)n * (n - 1) / 2, M = (long)B * N;
    const TrainWs w = train_layout((uintptr_t)workspace, M, n_layers);
    const float *layers = params + 128L * in_dim + 128;
    const float *dec_w = layers + (long)n_layers * kLayerFloats;
    float *g_emb_w = grads, *g_emb_b = grads + 128L * in_dim, *g_layers = g_emb_b + 128;
    float *g_dec_w = g_layers + (long)n_layers * kLayerFloats, *g_dec_b = g_dec_w + 128;
    const size_t row = (size_t)M * 128;
    hipError_t e = hipSuccess;
    int nb = 0;
    GNNGLS_TRY(hipMemsetAsync(g_dec_b + 1, 0, 3 * sizeof(float), st));                                   // pad
    // decision layer (models.py:69): y = h.w + b
    { ProfScope ps(GNNGLS_PROF_TRAIN_COLSUM, st);
      GNNGLS_TRY(gnngls::launch_colsum(gnngls::CS_ROWSCALE, w.H + row * n_layers, dy, nullptr, M, 128, 1, w.CSP, &nb, st));
      GNNGLS_TRY(gnngls::launch_colsum_store(w.CSP, nb, 128, 1, g_dec_w, nullptr, st));
      GNNGLS_TRY(gnngls::launch_sum_vector(dy, M, g_dec_b, st)); }
    { ProfScope ps(GNNGLS_PROF_TRAIN_ELEMENTWISE, st);
      GNNGLS_TRY(gnngls::launch_outer_rows(dy, dec_w, w.DA, M, st)); }
    for (int l = n_layers - 1; l >= 0; --l) {
        const float *fc_w, *attn_l, *attn_r, *bn1_g, *bn1_b, *w1, *b1, *w2, *b2, *bn2_g, *bn2_b;
        layer_pointers(layers + (long)l * kLayerFloats, fc_w, attn_l, attn_r, bn1_g, bn1_b, w1, b1, w2, b2, bn2_g, bn2_b);
        float *d_fc_w, *d_attn_l, *d_attn_r, *d_bn1_g, *d_bn1_b, *d_w1, *d_b1, *d_w2, *d_b2, *d_bn2_g, *d_bn2_b;
        layer_pointers(g_layers + (long)l * kLayerFloats, d_fc_w, d_attn_l, d_attn_r, d_bn1_g, d_bn1_b, d_w1, d_b1, d_w2,
                       d_b2, d_bn2_g, d_bn2_b);
        const float *h = w.H + row * l, *ft = w.FT + row * l, *g = w.G + row * l, *h1 = w.H1 + row * l, *h3 = w.H3 + row * l;
        const float *att = w.ATT + (size_t)M * 16 * l, *bn = w.BN + (size_t)l * 8 * 128;
        // BatchNorm 2 backward (models.py:35): DA = d(layer output) -> DB = d h3
        { ProfScope ps(GNNGLS_PROF_TRAIN_COLSUM, st);
          GNNGLS_TRY(gnngls::launch_colsum(gnngls::CS_SUM_PROD, w.DA, h3, nullptr, M, 128, 0, w.CSP, &nb, st));
          GNNGLS_TRY(gnngls::launch_bn_bwd_finalize(w.CSP, nb, M, bn2_g, bn + 4 * 128, bn + 5 * 128, d_bn2_g, d_bn2_b, w.COEF, st)); }
        { ProfScope ps(GNNGLS_PROF_TRAIN_ELEMENTWISE, st);
          // d h3 = BatchNorm-2 backward of DA, and x = BN1(h1) recomputed, in one elementwise pass
          GNNGLS_TRY(gnngls::launch_bn_bwd_apply_and_affine(w.DA, h3, bn + 4 * 128, w.COEF, w.DB, h1, bn + 2 * 128, bn + 3 * 128,
                                                            w.X2, M, st));
          GNNGLS_TRY(gnngls::launch_transpose_pair(w2, w1, w.WT, w.WT + 512 * 128, st)); }              // W2^T, W1^T
        // feed-forward block backward (models.py:28-33): h3 = x + W2 relu(W1 x + b1) + b2
        float *hid = w.HID + 4 * row * l;                    // saved ReLU(W1 x + b1)
        { ProfScope ps(GNNGLS_PROF_TRAIN_GEMM_TN, st);
          GNNGLS_TRY(gnngls::launch_gemm_tn(w.DB, hid, M, 128, 512, w.TNP, d_w2, d_b2, st)); }   // d W2 and d b2 = colsum(d h3)
        { ProfScope ps(GNNGLS_PROF_TRAIN_GEMM_BWD, st);      // d x = ((d h3 * W2) . [relu > 0]) * W1 + d h3; hid <- d pre
          GNNGLS_TRY(gnngls::launch_ffn_fused_bwd(w.DB, w.WT, w.WT + 512 * 128, w.COEF + 3 * 128, w.COEF + 4 * 128, w.DA, hid, M, st)); }
        { ProfScope ps(GNNGLS_PROF_TRAIN_GEMM_TN, st);
          GNNGLS_TRY(gnngls::launch_gemm_tn(hid, w.X2, M, 512, 128, w.TNP, d_w1, d_b1, st)); }   // d W1 and d b1 = colsum(d pre)
        // BatchNorm 1 backward (models.py:27): DA = d x -> DB = d h1 (= d h through the skip, = d GATConv output)
        { ProfScope ps(GNNGLS_PROF_TRAIN_COLSUM, st);
          GNNGLS_TRY(gnngls::launch_colsum(gnngls::CS_SUM_PROD, w.DA, h1, nullptr, M, 128, 0, w.CSP, &nb, st));
          GNNGLS_TRY(gnngls::launch_bn_bwd_finalize(w.CSP, nb, M, bn1_g, bn, bn + 128, d_bn1_g, d_bn1_b, w.COEF, st)); }
        { ProfScope ps(GNNGLS_PROF_TRAIN_ELEMENTWISE, st);
          GNNGLS_TRY(gnngls::launch_bn_bwd_apply(w.DA, h1, bn, w.COEF, w.DB, M, st)); }
        // GATConv backward (models.py:23)
        { ProfScope ps(GNNGLS_PROF_TRAIN_GAT_BWD, st);
          GNNGLS_TRY(gnngls::launch_gat_bwd_rows(ft, w.DB, g, att, attn_l, attn_r, B, n, w.PART, w.PMS, st)); }
        { ProfScope ps(GNNGLS_PROF_TRAIN_ELEMENTWISE, st);
          GNNGLS_TRY(gnngls::launch_gat_bwd_combine(w.PART, w.PMS, attn_l, attn_r, M, w.DFT, w.DLR, w.DLR + (size_t)M * 8, st)); }
        { ProfScope ps(GNNGLS_PROF_TRAIN_COLSUM, st);
          GNNGLS_TRY(gnngls::launch_colsum(gnngls::CS_HEADSCALE, ft, w.DLR, w.DLR + (size_t)M * 8, M, 128, 0, w.CSP, &nb, st));
          GNNGLS_TRY(gnngls::launch_colsum_store(w.CSP, nb, 128, 1, d_attn_l, d_attn_r, st)); }
        { ProfScope ps(GNNGLS_PROF_TRAIN_GEMM_TN, st);
          GNNGLS_TRY(gnngls::launch_gemm_tn(w.DFT, h, M, 128, 128, w.TNP, d_fc_w, nullptr, st)); }
        { ProfScope ps(GNNGLS_PROF_TRAIN_GEMM_BWD, st);      // d h = d ft * Wfc + d h1 (skip, models.py:15)
          GNNGLS_TRY(gnngls::launch_gemm_wkn(gnngls::GEMM_EPI_ADD, w.DFT, fc_w, w.DA, M, 128, 128, w.DB, st)); }
    }
    // embedding (models.py:66): h0 = x We^T + be
    { ProfScope ps(GNNGLS_PROF_TRAIN_COLSUM, st);
      for (int d = 0; d < in_dim; ++d) {
          GNNGLS_TRY(gnngls::launch_colsum(gnngls::CS_ROWSCALE, w.DA, feat + d, nullptr, M, 128, in_dim, w.CSP, &nb, st));
          GNNGLS_TRY(gnngls::launch_colsum_store(w.CSP, nb, 128, in_dim, g_emb_w + d, d == 0 ? g_emb_b : nullptr, st));
      } }
    return GNNGLS_OK;
}

#undef GNNGLS_TRY

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// per-kernel-class timing
// ---------------------------------------------------------------------------------------------
extern "C" {

int gnngls_debug_set_penalty16_limit(int limit) {
    if (limit < 1 || limit > 65535) return fail(GNNGLS_ERR_ARG, "penalty16 limit must be in 1..65535");
    g_pen16_limit = limit;
    return GNNGLS_OK;
}

int gnngls_profile_set_executed_evals(int64_t *device_buffer) {
    g_exec_evals.store((long long *)device_buffer, std::memory_order_relaxed);
    return GNNGLS_OK;
}

int gnngls_debug_set_gls_prune(int mode) {
    if (mode < -1 || mode > 1) return fail(GNNGLS_ERR_ARG, "gls prune mode must be -1 (default: on), 0 (full scans) or 1 (on)");
    g_prune_mode.store(mode, std::memory_order_relaxed);
    return GNNGLS_OK;
}

int gnngls_debug_set_gls_team(int mode) {
    if (mode < -1 || mode > 1) return fail(GNNGLS_ERR_ARG, "gls team mode must be -1 (policy), 0 (never) or 1 (wherever it exists)");
    g_team_mode.store(mode, std::memory_order_relaxed);
    return GNNGLS_OK;
}

int gnngls_debug_set_gls_threads(int threads) {
    // 1024 is not accepted: only the 128-VGPR instantiations are compiled for 16-wave workgroups (the default policy
    // picks them itself where a workgroup owns a CU); the 64- and 80-VGPR builds are bounded at 512 threads
    if (threads != 0 && threads != 64 && threads != 128 && threads != 256 && threads != 512)
        return fail(GNNGLS_ERR_ARG, "gls threads override must be 0 (default policy), 64, 128, 256 or 512");
    gnngls::gls_set_block_threads_override(threads);
    return GNNGLS_OK;
}

int gnngls_debug_set_stamp_buffer(void *device_buffer) {
    g_stamp_buffer = (long long *)device_buffer;
    return GNNGLS_OK;
}

int gnngls_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    for (auto &sp : g_spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    g_spans.clear();
    g_prof_on = on != 0;
    return GNNGLS_OK;
}

int gnngls_profile_collect(double *ms_by_kind, int64_t *launches_by_kind) {
    if (!ms_by_kind || !launches_by_kind) return fail(GNNGLS_ERR_ARG, "profile_collect: bad argument");
    for (int k = 0; k < GNNGLS_PROF_KINDS; ++k) { ms_by_kind[k] = 0.0; launches_by_kind[k] = 0; }
    std::vector<ProfSpan> spans;
    { std::lock_guard<std::mutex> lock(g_prof_mutex); spans.swap(g_spans); }
    for (auto &sp : spans) {
        hipError_t e = hipEventSynchronize(sp.b);
        if (e != hipSuccess) return hip_fail(e, "profile_collect");
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, sp.a, sp.b);
        if (e != hipSuccess) return hip_fail(e, "profile_collect");
        ms_by_kind[sp.kind] += ms; launches_by_kind[sp.kind] += 1;
        (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b);
    }
    return GNNGLS_OK;
}

}  // extern "C"
