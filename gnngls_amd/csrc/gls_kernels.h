// gls_kernels.h -- internal interface between the GLS kernels (gls_kernels.hip) and the C ABI (capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GNNGLS_STATUS_WATCHDOG_DEV 1
#define GNNGLS_STATUS_PENALTY_OVERFLOW_DEV 2
#define GNNGLS_STATUS_ASYMMETRIC_DEV 3

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
    long long *evals_exec;     // [B] or NULL: evaluations actually executed (the counting instantiations book the pruned scans'
                               // candidates; the others copy `evals`), zeroed by the host
    int32_t *status;
    int32_t *pen_ws;           // global store: [B,n,n] int32; compact store: [B,n(n-1)/2] int32; zeroed by the host
    int pen16_limit;           // 65535 (see gnngls_debug_set_penalty16_limit)
    long long *stamps;         // diagnostic builds (-DGLS_STAMPS) only: [B,16] cycle totals, else unused
    // improvement trace: one entry whenever the returned best improves (algorithms.py:143,190-191)
    double *imp_cost;          // [B,imp_cap] or NULL
    float *imp_time;           // [B,imp_cap] seconds since workgroup start, or NULL
    long long *imp_iter;       // [B,imp_cap] completed outer iterations at that moment, or NULL
    int imp_cap;
    int32_t *imp_len;          // [B] number of improvements (may exceed imp_cap); written by the kernel
    // pruned descent scans (best improvement, symmetric stores, n >= 80): 32 nearest neighbours per node, see
    // neighbor_lists_kernel; NULL = full scans
    const uint8_t *nl_id;      // [B,n,32] node ids, nearest first
    const int32_t *prune_ok;   // [B] 1 = the instance's matrix is within the magnitude bound of the pruning argument
    const int32_t *asym;       // [B] 1 = the instance's matrix is not bitwise symmetric (symmetric stores: not searched, status 3); or NULL
};

enum { GLS_STORE_GLOBAL = 0, GLS_STORE_TRI = 1, GLS_STORE_COMPACT = 2 };
size_t gls_lds_bytes(int n, int store, int penalty_bits, bool team = false);
int gls_block_threads(int n, int store, int penalty_bits = 32, bool half_scans = true);
void gls_set_block_threads_override(int threads);   // 0 = default policy (experiments only)
// resident wavefronts per SIMD (= register budget) of the kernel instantiation for this configuration: 4 or 8 for the
// compact store, fixed for the others
int gls_waves_per_simd(int store, int n, int batch, int num_cus, int threads, size_t lds);
// team: perturbation phase on all wavefronts of the workgroup (for workgroups that own their CU); only where
// gls_team_supported() says so
bool gls_team_supported(int store, int penalty_bits, int wps, int n, int threads);
bool gls_edge_form(int store, int penalty_bits, int wps, bool team, bool first_improvement);
bool gls_wps2_supported(int store, int penalty_bits, int n, int threads, bool first_improvement);
hipError_t launch_gls(const GlsArgs &A, int store, int penalty_bits, int threads, int wps, bool team, bool first_improvement,
                      hipStream_t stream);
hipError_t gls_kernel_resources(const GlsArgs &A, int store, int penalty_bits, int threads, int wps, bool team, bool first_improvement,
                                int *vgprs, int *scratch_bytes);
constexpr int kNeighborListLen = 32;
bool gls_prune_supported(int store, int n, bool first_improvement, int wps);
bool gls_count_supported(int store, int wps, int n, bool first_improvement, bool trace);
hipError_t launch_symmetry_check(const double *D, int B, int n, int32_t *asym, hipStream_t stream);
hipError_t launch_neighbor_lists(const double *D, int B, int n, uint8_t *nl_id, int32_t *prune_ok, hipStream_t stream);
hipError_t launch_delta_all(const int32_t *tour, const double *D, int B, int n, int op, double *out, hipStream_t stream);
hipError_t launch_best_move(const int32_t *tour, const double *D, int B, int n, int op, const int32_t *pos_i,
                            bool first_improvement, double *delta_out, int32_t *move_out, int32_t *new_tour,
                            hipStream_t stream);
hipError_t launch_tour_cost(const int32_t *tour, const double *D, int B, int n, double *out, hipStream_t stream);
hipError_t launch_nearest_neighbor(const double *W, int B, int n, int depot, int32_t *tour_out, hipStream_t stream);

}  // namespace gnngls
