// tests/native/capi_fuzz.cpp -- TEST INFRASTRUCTURE (CPU, no GPU needed).
// Drives the host side of the C ABI (gnngls_amd/csrc/capi.hip compiled host-only with -fsanitize=address,undefined; the
// kernel launchers come from the regular libgnngls_hip.so and are never reached by a rejected call; a call whose shape
// passes validation is only made with all pointers set when no GPU is visible -- it then fails cleanly inside HIP --
// and keeps one required pointer NULL when there is one) with hostile
// arguments: negative / huge B and n, NULL pointers, undersized workspaces, bad enums.  Every call must return an error
// code with a message (or GNNGLS_OK for an empty batch) -- never crash, never read a data pointer on the host (all
// data pointers handed in are 1-byte heap blocks, so AddressSanitizer traps any host-side dereference).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/gnngls_hip.h"

extern "C" int hipGetDeviceCount(int *count);     // libamdhip64 (hipError_t is an int-sized enum; 0 = hipSuccess)

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static unsigned long long rnd() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return rng_state;
}
static int pick(const int *v, int k) { return v[rnd() % k]; }

static int calls = 0, rejected = 0, hip_errors = 0, ok = 0;
static void tally(int rc, const char *what) {
    ++calls;
    if (rc == GNNGLS_OK) { ++ok; return; }
    const char *msg = gnngls_last_error();
    if (!msg || !msg[0]) { fprintf(stderr, "%s: error code %d without a message\n", what, rc); exit(2); }
    if (rc == GNNGLS_ERR_ARG || rc == GNNGLS_ERR_UNSUPPORTED) ++rejected;
    else if (rc == GNNGLS_ERR_HIP) ++hip_errors;
    else { fprintf(stderr, "%s: unknown return code %d\n", what, rc); exit(2); }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    static const int Bs[] = {-2147483647 - 1, -5, -1, 0, 0, 1, 3};
    static const int ns[] = {-2147483647 - 1, -1, 0, 1, 2, 3, 5, 146, 255, 256, 65535, 65536, 1 << 30, 2147483647};
    static const int bits[] = {-3, -2, -1, 0, 8, 16, 32, 64};
    static const int small[] = {-7, -1, 0, 1, 2, 8, 1 << 20};
    static const int64_t wsb[] = {-1, 0, 1, 255, 256, 4096, 1ll << 40};
    void *blk = malloc(1);                      // stands in for a device pointer: never to be touched on the host
    int device_count = 0;
    const int has_device = hipGetDeviceCount(&device_count) == 0 && device_count > 0;
    for (int it = 0; it < iters; ++it) {
        void *p[12];
        for (int k = 0; k < 12; ++k) p[k] = (rnd() & 3) ? blk : NULL;
        const int B = pick(Bs, 7), n = pick(ns, 14);
        // a call that passes validation with B > 0 would enqueue real work on fake pointers: only allowed without a GPU
        // (then the HIP calls fail cleanly); with a GPU present keep at least one argument invalid
        const bool valid_shape = B > 0 && n >= 3 && n <= 65535;
        if (valid_shape && has_device) p[rnd() % 2] = NULL;          // D / tour is always among the first two pointers
        switch (rnd() % 12) {
        case 0: tally(gnngls_two_opt_delta_all((const int32_t *)p[0], (const double *)p[1], B, n, (double *)p[2], NULL), "two_opt_delta_all"); break;
        case 1: tally(gnngls_relocate_delta_all((const int32_t *)p[0], (const double *)p[1], B, n, (double *)p[2], NULL), "relocate_delta_all"); break;
        case 2: tally(gnngls_best_move((const int32_t *)p[0], (const double *)p[1], B, n, pick(small, 7), (const int32_t *)p[2], (int)(rnd() & 1),
                                       (double *)p[3], (int32_t *)p[4], (int32_t *)p[5], NULL), "best_move"); break;
        case 3: tally(gnngls_tour_cost((const int32_t *)p[0], (const double *)p[1], B, n, (double *)p[2], NULL), "tour_cost"); break;
        case 4: tally(gnngls_nearest_neighbor((const double *)p[0], B, n, pick(small, 7), (int32_t *)p[1], NULL), "nearest_neighbor"); break;
        case 5: {
            const double wd = (rnd() & 1) ? 1.0 : ((rnd() & 1) ? 0.0 : -1.0);
            tally(gnngls_gls_run((const double *)p[0], (const double *)p[2], pick(small, 7), B, n, (const int32_t *)p[1], (const double *)p[3],
                                 pick(small, 7), (int)(rnd() & 1), pick(bits, 8), (int64_t)pick(small, 7), 0.01, wd,
                                 (int32_t *)p[4], (double *)p[5], (int64_t *)p[6], (double *)p[7], (float *)p[8], pick(small, 7),
                                 (int32_t *)p[9], NULL, NULL, (int32_t *)p[10],
                                 (double *)p[11], NULL, NULL, pick(small, 7), (rnd() & 1) ? (int32_t *)blk : NULL, NULL), "gls_run");
            break;
        }
        case 6: tally(gnngls_regret_forward((const float *)p[0], (const float *)p[1], B, n, pick(small, 7), pick(small, 7), (float *)p[2], p[3],
                                            wsb[rnd() % 7], NULL), "regret_forward"); break;
        case 7: tally(gnngls_regret_train_forward((const float *)p[0], (const float *)p[1], B, n, pick(small, 7), pick(small, 7), 1e-5f,
                                                  (float *)p[2], (float *)p[3], p[4], wsb[rnd() % 7], NULL), "regret_train_forward"); break;
        case 8: tally(gnngls_regret_train_backward((const float *)p[0], (const float *)p[1], (const float *)p[2], B, n, pick(small, 7),
                                                   pick(small, 7), (float *)p[3], p[4], wsb[rnd() % 7], NULL), "regret_train_backward"); break;
        case 9: tally(gnngls_pack_features((const double *)p[0], B, n, 1.0, 0.0, (float *)p[1], NULL), "pack_features"); break;
        case 10: tally(gnngls_unpack_regret((const float *)p[0], B, n, 1.0, 0.0, (double *)p[1], NULL), "unpack_regret"); break;
        default: {
            int store = -1, threads = -1, lds = -1, per_cu = -1;
            const int rc = gnngls_gls_describe_config(n, B, pick(bits, 8), &store, &threads, &lds, &per_cu);
            tally(rc, "gls_describe_config");
            if (rc == GNNGLS_OK && (threads < 64 || threads > 1024 || lds < 0 || lds > 160 * 1024 || per_cu < 0)) {
                fprintf(stderr, "describe_config(n=%d,B=%d): threads %d lds %d per_cu %d\n", n, B, threads, lds, per_cu);
                return 2;
            }
            (void)gnngls_gls_resident_capacity(n);
            (void)gnngls_model_packed_floats(pick(small, 7), pick(small, 7));
            (void)gnngls_regret_forward_workspace_bytes(B, n);
            (void)gnngls_regret_train_workspace_bytes(B, n, pick(small, 7));
        } }
    }
    tally(gnngls_debug_set_penalty16_limit(0), "penalty16_limit");
    tally(gnngls_debug_set_penalty16_limit(70000), "penalty16_limit");
    tally(gnngls_profile_collect(NULL, NULL), "profile_collect");
    free(blk);
    printf("capi_fuzz: %d calls, %d rejected, %d hip errors, %d ok (empty batches / queries)\n", calls, rejected, hip_errors, ok);
    return rejected > calls / 2 ? 0 : 3;
}
