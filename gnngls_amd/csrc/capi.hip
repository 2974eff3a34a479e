// capi.hip -- extern "C" entry points of libgnngls_hip.so (declared in include/gnngls_hip.h).
// Argument checking, error strings and launch glue only; kernels live in gls_kernels.hip and
// model_kernels.hip.  No CPU fallback: every entry point either enqueues HIP work or fails.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/gnngls_hip.h"
#include "gls_kernels.h"

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

bool tri_fits(int n) { return gnngls::gls_lds_bytes(n, true) <= kLdsPerCU; }
}  // namespace

extern "C" {

int gnngls_abi_version(void) { return 1; }
const char *gnngls_last_error(void) { return g_err; }

int gnngls_gls_resident_capacity(int n) {
    if (n < 3 || !tri_fits(n)) return 0;
    size_t lds = gnngls::gls_lds_bytes(n, true);
    int by_lds = (int)(kLdsPerCU / lds);
    int by_waves = kMaxWavesPerCU / (gnngls::gls_block_threads(n) / 64);
    int per_cu = by_lds < by_waves ? by_lds : by_waves;
    return per_cu * num_cus();
}

int gnngls_two_opt_delta_all(const int32_t *tour, const double *D, int B, int n, double *out, void *stream) {
    if (!tour || !D || !out || B < 0 || n < 3) return fail(GNNGLS_ERR_ARG, "two_opt_delta_all: bad argument");
    if (B == 0) return GNNGLS_OK;
    hipError_t e = gnngls::launch_delta_all(tour, D, B, n, 0, out, (hipStream_t)stream);
    return e == hipSuccess ? GNNGLS_OK : hip_fail(e, "two_opt_delta_all");
}

int gnngls_relocate_delta_all(const int32_t *tour, const double *D, int B, int n, double *out, void *stream) {
    if (!tour || !D || !out || B < 0 || n < 3) return fail(GNNGLS_ERR_ARG, "relocate_delta_all: bad argument");
    if (B == 0) return GNNGLS_OK;
    hipError_t e = gnngls::launch_delta_all(tour, D, B, n, 1, out, (hipStream_t)stream);
    return e == hipSuccess ? GNNGLS_OK : hip_fail(e, "relocate_delta_all");
}

int gnngls_best_move(const int32_t *tour, const double *D, int B, int n, int op, const int32_t *pos_i,
                     int first_improvement, double *delta_out, int32_t *move_out, int32_t *new_tour, void *stream) {
    if (!tour || !D || !delta_out || !move_out || B < 0 || n < 3 || (op != 0 && op != 1) || n > 65535)
        return fail(GNNGLS_ERR_ARG, "best_move: bad argument");
    if (B == 0) return GNNGLS_OK;
    hipError_t e = gnngls::launch_best_move(tour, D, B, n, op, pos_i, first_improvement != 0, delta_out, move_out,
                                            new_tour, (hipStream_t)stream);
    return e == hipSuccess ? GNNGLS_OK : hip_fail(e, "best_move");
}

int gnngls_tour_cost(const int32_t *tour, const double *D, int B, int n, double *cost_out, void *stream) {
    if (!tour || !D || !cost_out || B < 0 || n < 1) return fail(GNNGLS_ERR_ARG, "tour_cost: bad argument");
    if (B == 0) return GNNGLS_OK;
    hipError_t e = gnngls::launch_tour_cost(tour, D, B, n, cost_out, (hipStream_t)stream);
    return e == hipSuccess ? GNNGLS_OK : hip_fail(e, "tour_cost");
}

int gnngls_nearest_neighbor(const double *W, int B, int n, int depot, int32_t *tour_out, void *stream) {
    if (!W || !tour_out || B < 0 || n < 1 || depot < 0 || depot >= n)
        return fail(GNNGLS_ERR_ARG, "nearest_neighbor: bad argument");
    if (B == 0) return GNNGLS_OK;
    hipError_t e = gnngls::launch_nearest_neighbor(W, B, n, depot, tour_out, (hipStream_t)stream);
    return e == hipSuccess ? GNNGLS_OK : hip_fail(e, "nearest_neighbor");
}

int gnngls_gls_run(const double *D, const double *guides, int n_guides, int B, int n,
                   const int32_t *init_tour, const double *init_cost,
                   int perturbation_moves, int first_improvement,
                   int64_t max_outer_iters, double time_limit_s, double watchdog_s,
                   int32_t *best_tour, double *best_cost, int64_t *outer_iters,
                   double *trace_cost, float *trace_time, int trace_cap, int32_t *trace_len,
                   int32_t *penalty_out, int64_t *evals_out, int32_t *status, void *stream) {
    if (!D || !init_tour || !init_cost || !best_tour || !best_cost || B < 0 || n < 3 || n > 65535 || trace_cap < 0)
        return fail(GNNGLS_ERR_ARG, "gls_run: bad argument");
    if (max_outer_iters != 0 && (!guides || n_guides < 1))
        return fail(GNNGLS_ERR_ARG, "gls_run: guides required when outer iterations are requested");
    if (!(watchdog_s > 0.0)) return fail(GNNGLS_ERR_ARG, "gls_run: watchdog_s must be > 0");
    if (B == 0) return GNNGLS_OK;
    hipStream_t st = (hipStream_t)stream;
    gnngls::GlsArgs A;
    memset(&A, 0, sizeof(A));
    A.D = D; A.guides = guides ? guides : D; A.n_guides = n_guides > 0 ? n_guides : 1; A.B = B; A.n = n;
    A.init_tour = init_tour; A.init_cost = init_cost;
    A.perturbation_moves = perturbation_moves;
    A.max_outer_iters = max_outer_iters; A.time_limit_s = time_limit_s; A.watchdog_s = watchdog_s;
    A.best_tour = best_tour; A.best_cost = best_cost; A.outer_iters = (long long *)outer_iters;
    A.trace_cost = trace_cost; A.trace_time = trace_time; A.trace_cap = trace_cost ? trace_cap : 0;
    A.trace_len = trace_len; A.penalty_out = penalty_out; A.evals = (long long *)evals_out; A.status = status;
    bool tri = tri_fits(n);
    int32_t *ws = nullptr;
    if (!tri) {
        // global-memory fallback for instances whose triangles exceed LDS: penalties in a zeroed workspace
        size_t bytes = (size_t)B * n * n * sizeof(int32_t);
        hipError_t e = hipMallocAsync((void **)&ws, bytes, st);
        if (e != hipSuccess) return hip_fail(e, "gls_run: workspace alloc");
        e = hipMemsetAsync(ws, 0, bytes, st);
        if (e != hipSuccess) return hip_fail(e, "gls_run: workspace memset");
        A.pen_ws = ws;
    }
    hipError_t e = gnngls::launch_gls(A, tri, first_improvement != 0, st);
    if (ws) (void)hipFreeAsync(ws, st);
    return e == hipSuccess ? GNNGLS_OK : hip_fail(e, "gls_run");
}

}  // extern "C"
