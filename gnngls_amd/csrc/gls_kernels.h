// gls_kernels.h -- internal interface between the GLS kernels (gls_kernels.hip) and the C ABI (capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GNNGLS_STATUS_WATCHDOG_DEV 1

namespace gnngls {

struct GlsArgs {
    const double *D;           // [B,n,n]
    const double *guides;      // [G,B,n,n]
    int n_guides, B, n;
    const int32_t *init_tour;  // [B,n+1]
    const double *init_cost;   // [B]
    int perturbation_moves;
    long long max_outer_iters;
    double time_limit_s, watchdog_s;
    int32_t *best_tour;
    double *best_cost;
    long long *outer_iters;
    double *trace_cost;
    float *trace_time;
    int trace_cap;
    int32_t *trace_len;
    int32_t *penalty_out;
    long long *evals;
    int32_t *status;
    int32_t *pen_ws;           // global-store mode only: [B,n,n] int32, zeroed by the host
};

size_t gls_lds_bytes(int n, bool tri);
int gls_block_threads(int n);
hipError_t launch_gls(const GlsArgs &A, bool tri, bool first_improvement, hipStream_t stream);
hipError_t launch_delta_all(const int32_t *tour, const double *D, int B, int n, int op, double *out, hipStream_t stream);
hipError_t launch_best_move(const int32_t *tour, const double *D, int B, int n, int op, const int32_t *pos_i,
                            bool first_improvement, double *delta_out, int32_t *move_out, int32_t *new_tour,
                            hipStream_t stream);
hipError_t launch_tour_cost(const int32_t *tour, const double *D, int B, int n, double *out, hipStream_t stream);
hipError_t launch_nearest_neighbor(const double *W, int B, int n, int depot, int32_t *tour_out, hipStream_t stream);

}  // namespace gnngls
