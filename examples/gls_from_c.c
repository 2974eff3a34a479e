/* examples/gls_from_c.c -- the C ABI of include/gnngls_hip.h used from plain C (HIP runtime only: no Python, no torch).
 *
 *   gcc -O2 examples/gls_from_c.c -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Lgnngls_amd -lgnngls_hip \
 *       -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/gnngls_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/gls_from_c
 *   /tmp/gls_from_c [n] [instances] [outer_iterations]
 *
 * Random Euclidean instances -> nearest-neighbour tours (algorithms.py:9-18) -> tour_cost (__init__.py:17-21) ->
 * guided_local_search with the classical `weight` guide (algorithms.py:135-195) for a fixed number of outer iterations.
 * Prints the mean initial and final tour lengths. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "gnngls_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_GLS(x) do { int rc_ = (x); if (rc_ != GNNGLS_OK) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, gnngls_last_error()); return 3; } } while (0)

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 50, B = argc > 2 ? atoi(argv[2]) : 64;
    const long K = argc > 3 ? atol(argv[3]) : 200;
    const size_t nn = (size_t)n * n;
    double *D = (double *)malloc(B * nn * sizeof(double)), *pos = (double *)malloc((size_t)n * 2 * sizeof(double));
    srand(7);
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < 2 * n; ++i) pos[i] = rand() / (double)RAND_MAX;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                D[b * nn + (size_t)i * n + j] = i == j ? 0.0 : hypot(pos[2 * i] - pos[2 * j], pos[2 * i + 1] - pos[2 * j + 1]);
    }
    double *dD, *dinit_cost, *dbest_cost;
    int32_t *dinit, *dbest, *dstatus;
    int64_t *diters;
    CHECK_HIP(hipMalloc((void **)&dD, B * nn * sizeof(double)));
    CHECK_HIP(hipMalloc((void **)&dinit, (size_t)B * (n + 1) * sizeof(int32_t)));
    CHECK_HIP(hipMalloc((void **)&dbest, (size_t)B * (n + 1) * sizeof(int32_t)));
    CHECK_HIP(hipMalloc((void **)&dinit_cost, B * sizeof(double)));
    CHECK_HIP(hipMalloc((void **)&dbest_cost, B * sizeof(double)));
    CHECK_HIP(hipMalloc((void **)&diters, B * sizeof(int64_t)));
    CHECK_HIP(hipMalloc((void **)&dstatus, B * sizeof(int32_t)));
    CHECK_HIP(hipMemcpy(dD, D, B * nn * sizeof(double), hipMemcpyHostToDevice));

    CHECK_GLS(gnngls_nearest_neighbor(dD, B, n, 0, dinit, NULL));
    CHECK_GLS(gnngls_tour_cost(dinit, dD, B, n, dinit_cost, NULL));
    CHECK_GLS(gnngls_gls_run(dD, dD /* one guide: the weights */, 1, B, n, dinit, dinit_cost, 20, 0, 0, (int64_t)K, 0.0, 60.0,
                             dbest, dbest_cost, diters, NULL, NULL, 0, NULL, NULL, NULL, dstatus,
                             NULL, NULL, NULL, 0, NULL /* no improvement trace */, NULL));
    CHECK_HIP(hipDeviceSynchronize());

    double *c0 = (double *)malloc(B * sizeof(double)), *c1 = (double *)malloc(B * sizeof(double));
    int32_t *status = (int32_t *)malloc(B * sizeof(int32_t));
    CHECK_HIP(hipMemcpy(c0, dinit_cost, B * sizeof(double), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(c1, dbest_cost, B * sizeof(double), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(status, dstatus, B * sizeof(int32_t), hipMemcpyDeviceToHost));
    double m0 = 0, m1 = 0;
    int bad = 0;
    for (int b = 0; b < B; ++b) { m0 += c0[b]; m1 += c1[b]; bad += status[b] != GNNGLS_STATUS_OK || !(c1[b] <= c0[b]); }
    printf("abi %d: %d TSP%d instances, %ld outer iterations: mean tour length %.6f -> %.6f, %d anomalies\n",
           gnngls_abi_version(), B, n, K, m0 / B, m1 / B, bad);
    return bad ? 1 : 0;
}
