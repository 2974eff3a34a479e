/*
 * include/gnngls_hip.h -- C ABI of libgnngls_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the hot path of proroklab/gnngls (edge-regret GNN forward +
 * guided local search).  The reference has no FFI layer of its own: its boundary is the Python
 * call surface.  Each entry point below cites the reference function it
 * replaces (file:line into the reference tree); the gnngls_amd Python package binds them with ctypes and
 * re-exposes the reference's Python signatures (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch CUDA tensor .data_ptr()) unless the
 *     name ends in _host; buffers are caller-owned, contiguous, row-major; nothing is retained;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue work;
 *   - B = number of independent TSP instances in the batch, n = nodes per instance;
 *     a tour is int32[n+1] with the depot (node 0) at both ends (algorithms.py:10,17);
 *     D is an n*n fp64 matrix indexed by node label (nx.attr_matrix, algorithms.py:140);
 *     N = n(n-1)/2 line-graph nodes (= TSP edges i<j in itertools.combinations order);
 *   - return value: 0 on success, negative on error; gnngls_last_error() gives the message.
 *     There is NO CPU fallback anywhere in this library.
 */
#ifndef GNNGLS_HIP_H
#define GNNGLS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNNGLS_OK 0
#define GNNGLS_ERR_ARG (-1)
#define GNNGLS_ERR_HIP (-2)
#define GNNGLS_ERR_UNSUPPORTED (-3)

/* operator codes for gnngls_best_move */
#define GNNGLS_OP_TWO_OPT 0
#define GNNGLS_OP_RELOCATE 1

/* per-instance status words written by gnngls_gls_run */
#define GNNGLS_STATUS_OK 0
#define GNNGLS_STATUS_WATCHDOG 1      /* watchdog fired (search aborted, best-so-far returned) */
#define GNNGLS_STATUS_PENALTY_OVERFLOW 2   /* a 16-bit LDS penalty counter would pass 65535: rerun with penalty_bits=32 */
#define GNNGLS_STATUS_ASYMMETRIC 3    /* the instance's matrix is not bitwise symmetric and the run was on a store that keeps one
                                         triangle (every store but the global-memory one): NOT searched -- best = the start tour,
                                         0 iterations -- because the reference reads D[a,b] as indexed (operators.py:25-28,97-102)
                                         and the search would differ; rerun the instance with penalty_bits = -1 */

int gnngls_abi_version(void);
const char *gnngls_last_error(void);

/* Number of instances of size n that can be co-resident on the device for gnngls_gls_run
 * (one workgroup per instance, LDS-resident distance/penalty triangles); 0 if n needs the
 * global-memory fallback.  Host-side query, no device work. */
int gnngls_gls_resident_capacity(int n);

/* Host-side query: the storage configuration gnngls_gls_run would use for B instances of n nodes.
 *   *store     0 = global-memory store, 116 / 132 = distance + 16/32-bit penalty triangles in LDS,
 *              200 = compact store (distance triangle in LDS, penalties in global memory)
 *   *threads   workgroup size, *lds_bytes dynamic LDS per workgroup, *per_cu resident workgroups per CU (0: n/a) */
int gnngls_gls_describe_config(int n, int B, int penalty_bits, int *store, int *threads, int *lds_bytes, int *per_cu);

/* The same query for exactly the launch gnngls_gls_run(…, first_improvement, penalty_bits, …) makes (ABI v3).  The three
 * older queries -- gnngls_gls_describe_config, gnngls_gls_waves_per_simd, gnngls_gls_uses_team -- answer for a
 * best-improvement run; a first-improvement run can use another workgroup size (n = 25..33), register budget and form of
 * the perturbation phase.  *waves_per_simd, *team as documented there; *edge_form = 1 if the serial perturbation phase
 * (algorithms.py:150-185) runs in its edge form (tour edges in registers; symmetric stores, 32-bit counters, best
 * improvement), 0 for the scan-by-scan form.  Any output pointer may be NULL. */
int gnngls_gls_describe_run(int n, int B, int penalty_bits, int first_improvement, int *store, int *threads, int *lds_bytes,
                            int *per_cu, int *waves_per_simd, int *team, int *edge_form);

/* Vector registers per lane and scratch bytes per lane of the kernel instantiation that launch would run (trace != 0: with a
 * per-move trace buffer) -- hipFuncGetAttributes on the selected function, so it needs the device.  Every BASELINE.json shape
 * runs on an instantiation without scratch; tests/test_perf_floor_gpu.py asserts it so that a code-generation regression in
 * this ~70-instantiation kernel shows in the driver's GPU tests. */
int gnngls_gls_kernel_resources(int n, int B, int penalty_bits, int first_improvement, int trace, int *vgprs, int *scratch_bytes);

/* ---- K3u: move-evaluation tables (parity/unit kernels) ---------------------------------------
 * out[b][i][j] (shape [B, n+1, n+1]) = two_opt_cost(tour_b, D_b, i, j)   operators.py:14-29
 *                                    = relocate_cost(tour_b, D_b, i, j)  operators.py:83-103
 * for 1 <= i,j <= n-1; NaN elsewhere.  D may be asymmetric (index order follows the reference). */
int gnngls_two_opt_delta_all(const int32_t *tour, const double *D, int B, int n, double *out, void *stream);
int gnngls_relocate_delta_all(const int32_t *tour, const double *D, int B, int n, double *out, void *stream);

/* ---- one scan of a move operator --------------------------------------------------------------
 * op = GNNGLS_OP_TWO_OPT : two_opt_a2a (operators.py:32-50) / two_opt_o2a (operators.py:53-73)
 * op = GNNGLS_OP_RELOCATE: relocate_a2a (operators.py:129-147) / relocate_o2a (operators.py:106-126)
 * pos_i == NULL -> all-to-all scan; else one-to-all from position pos_i[b] (must be in 1..n-1,
 * the reference asserts this at operators.py:54,107; violating it returns GNNGLS_ERR_ARG only if
 * checked on the host side by the caller -- the kernel clamps nothing).
 * Acceptance rule operators.py:42,65,118,139: delta < best and not np.isclose(0, delta).
 * Outputs: delta_out[b] (0.0 if no move), move_out[b] = {i, j} ({0,0} if none),
 *          new_tour[b] = tour after the move (copy of the input if none). */
int gnngls_best_move(const int32_t *tour, const double *D, int B, int n, int op, const int32_t *pos_i,
                     int first_improvement, double *delta_out, int32_t *move_out, int32_t *new_tour,
                     void *stream);

/* ---- tour_cost (gnngls/__init__.py:17-21): sequential left-to-right fp64 sum ------------------ */
int gnngls_tour_cost(const int32_t *tour, const double *D, int B, int n, double *cost_out, void *stream);

/* ---- nearest_neighbor (algorithms.py:9-18) on a dense weight matrix W[B,n,n]; ties -> lowest id */
int gnngls_nearest_neighbor(const double *W, int B, int n, int depot, int32_t *tour_out, void *stream);

/* ---- K3: guided_local_search (algorithms.py:135-195), local_search (algorithms.py:111-132) -----
 * One persistent workgroup per instance.
 *   D            [B,n,n] fp64, symmetric (it comes from nx.attr_matrix of an undirected graph) unless
 *                penalty_bits == -1
 *   guides       [n_guides,B,n,n] fp64 utility numerators G.edges[e][guide] (algorithms.py:155),
 *                cycled per outer iteration (algorithms.py:147); may be NULL iff max_outer_iters == 0
 *   init_tour    [B,n+1], init_cost [B]
 *   max_outer_iters >= 0 : run exactly this many outer iterations (deterministic / parity mode);
 *                          0 = local_search only (algorithms.py:142)
 *   max_outer_iters <  0 : run until time_limit_s seconds of device wall clock have elapsed since
 *                          the workgroup started (reference mode, `while time.time() < t_lim`)
 *   penalty_bits width of the LDS-resident penalty counters (algorithms.py:138,161): 32, 16, or
 *                0 = auto: the fastest store that keeps all B instances resident (32-bit LDS counters, then the
 *                compact store with 32-bit counters in global memory), else the one with the largest
 *                residency.  16 = uint16 LDS counters (n=100: 3 workgroups per CU): a counter about to
 *                overflow stops that instance with GNNGLS_STATUS_PENALTY_OVERFLOW; the caller reruns it with 32.
 *                -1 = force the global-memory store: matrices stay in HBM/L2 and every evaluation keeps the
 *                reference's exact index order, so D may be ASYMMETRIC (the LDS stores keep one triangle).
 *                -2 = force the compact store (distance triangle in LDS, 32-bit counters in global memory; n <= 255).
 *   watchdog_s   hard abort (status GNNGLS_STATUS_WATCHDOG) if a workgroup runs longer than this
 *   outputs      best_tour [B,n+1], best_cost [B], outer_iters [B] (int64),
 *                trace_cost [B,trace_cap] cost after every accepted move (algorithms.py:127-130,
 *                180-183), trace_time [B,trace_cap] seconds since workgroup start (optional),
 *                trace_len [B] = number of accepted moves (may exceed trace_cap),
 *                penalty_out [B,n,n] int32 final penalties (optional),
 *                evals_out [B] int64 number of delta evaluations (optional), status [B] (optional)
 *   improvement trace (optional, any of the arrays may be NULL; imp_len == NULL disables it): one entry whenever the
 *                returned best improves -- after the initial descent (algorithms.py:143) and after an outer
 *                iteration whose descent ends below the best so far (algorithms.py:190-191):
 *                imp_cost [B,imp_cap] the new best cost, imp_time [B,imp_cap] seconds since workgroup start,
 *                imp_iter [B,imp_cap] int64 number of completed outer iterations (0 = initial descent),
 *                followed by ONE terminal entry (best_cost[b], time at the end of the search, outer_iters[b]);
 *                imp_len [B] = improvements + 1 (may exceed imp_cap: the terminal entry then takes the last slot, so
 *                with imp_cap >= 1 entry min(imp_len, imp_cap) - 1 is always the terminal one).
 *                A 10 s TSP100 search accepts ~2e6 moves but improves its best a few dozen times: this is the
 *                search-progress record (test.py:97-117, `best_cost` = cummin) that stays bounded at any run length;
 *                the per-move trace is exact while trace_len[b] <= trace_cap and TRUNCATED beyond (callers must check).
 * If trace_cap == 0 the per-move tour_cost recomputation (algorithms.py:176) is deferred to the
 * end of each perturbation phase -- the values that drive decisions are unchanged. */
int gnngls_gls_run(const double *D, const double *guides, int n_guides, int B, int n,
                   const int32_t *init_tour, const double *init_cost,
                   int perturbation_moves, int first_improvement, int penalty_bits,
                   int64_t max_outer_iters, double time_limit_s, double watchdog_s,
                   int32_t *best_tour, double *best_cost, int64_t *outer_iters,
                   double *trace_cost, float *trace_time, int trace_cap, int32_t *trace_len,
                   int32_t *penalty_out, int64_t *evals_out, int32_t *status,
                   double *imp_cost, float *imp_time, int64_t *imp_iter, int imp_cap, int32_t *imp_len, void *stream);

/* ---- K1/K2: edge-regret GNN forward ------------------------------------------------------------
 * EdgePropertyPredictionModel.forward (models.py:44-70) on the line graph of K_n (datasets.py:56-60),
 * specialised to the reference architecture: embed_dim 128, 8 heads x 16 (GATConv(128,16,8),
 * models.py:23), MLP hidden 512 (models.py:60), out_dim 1, eval-mode BatchNorm.  Layer count is a
 * run-time argument (the reference builds n_heads layers, models.py:59-61).
 *
 * Packed fp32 weight image (all blocks 16-byte aligned), in this order:
 *   embed_w[128*in_dim] embed_b[128]
 *   per layer: fc_w[128*128] attn_l[128] attn_r[128] bn1_scale[128] bn1_shift[128]
 *              w1[512*128] b1[512] w2[128*512] b2[128] bn2_scale[128] bn2_shift[128]
 *   dec_w[128] dec_b[1] pad[3]
 * where bn_scale = gamma / sqrt(running_var + eps), bn_shift = beta - running_mean * bn_scale
 * (eval-mode BatchNorm1d, models.py:28,35).  gnngls_amd.models packs a reference state_dict.
 *
 *   feat [B,N,in_dim] fp32 scaled edge features in line-graph node order (node = rank of (i<j) in
 *        itertools.combinations order); y_out [B,N] fp32.
 *   workspace: device scratch of at least gnngls_regret_forward_workspace_bytes(1, n) bytes; if it
 *        is smaller than the full-batch size the batch is processed in instance chunks. */
#define GNNGLS_MODEL_EMBED_DIM 128
#define GNNGLS_MODEL_HEADS 8
#define GNNGLS_MODEL_HIDDEN 512
int64_t gnngls_model_packed_floats(int in_dim, int n_layers);
int64_t gnngls_regret_forward_workspace_bytes(int B, int n);
int gnngls_regret_forward(const float *feat, const float *weights, int B, int n, int in_dim, int n_layers,
                          float *y_out, void *workspace, int64_t workspace_bytes, void *stream);

/* ABI v4 -- the same forward with the weight-only work hoisted out of it (test.py:43-54 loads a checkpoint ONCE and then
 * calls the model per instance, test.py:72-77): the feed-forward block (models.py:26-36) and the folded GATConv fc (models.py:23)
 * run on the bf16 matrix pipe from three bf16 pieces per fp32 weight in MFMA fragment order; that image depends on the weights
 * only.  gnngls_regret_prepare writes it for all layers into `prepared` (caller-owned device memory of
 * gnngls_regret_prepared_bytes(n_layers) bytes, valid until the weights change); gnngls_regret_forward_prepared is
 * gnngls_regret_forward reading it (prepared == NULL: the feed-forward block stays on the fp32 matrix pipe).
 * The image also holds A = We^T Wfc0^T [in_dim,128] and b' = Wfc0 be: the embedding (models.py:57,66) and the first layer's fc
 * (models.py:23) are both linear in the in_dim input features, so with an image (in_dim <= 32) ft = x A + b' is written by the
 * embedding pass itself instead of a [B N,128] x [128,128] product.
 * gnngls_regret_forward itself prepares into stream-ordered scratch on every call (2 launches per layer more). */
int64_t gnngls_regret_prepared_bytes(int n_layers);
int gnngls_regret_prepare(const float *weights, int in_dim, int n_layers, void *prepared, int64_t prepared_bytes, void *stream);
int gnngls_regret_forward_prepared(const float *feat, const float *weights, const void *prepared, int64_t prepared_bytes,
                                   int B, int n, int in_dim, int n_layers,
                                   float *y_out, void *workspace, int64_t workspace_bytes, void *stream);

/* ---- N4: one training step of the same model (scripts/train.py:20-32) -------------------------------------------
 * model.train(); y_pred = model(batch, x); loss = criterion(y_pred, y); loss.backward() for a dgl.batch of B line graphs
 * of K_n (train.py:118-121): BatchNorm1d uses the statistics of all B*N rows (models.py:27,35), GATConv is
 * differentiated through its edge softmax (models.py:23).  The criterion itself (MSELoss / BCEWithLogitsLoss,
 * train.py:106-112) is elementwise on [B*N] values and stays with the caller: the forward returns y_pred, the backward
 * takes dy = d loss / d y_pred.
 *
 * `params` is the RAW parameter image: the layout of the packed inference image above with the BatchNorm slots holding
 * gamma / beta instead of the folded scale / shift (gnngls_model_packed_floats floats); `grads` has the same layout.
 *   forward : feat [B,N,in_dim] -> y_out [B,N]; bn_batch_stats [n_layers][2 (BN1, BN2)][2 (mean, unbiased var)][128]
 *             is what the caller folds into running_mean / running_var (momentum update, torch semantics).
 *   backward: must follow a forward on the SAME workspace (activations are kept there); writes every gradient.
 * n <= 257 (register-resident source tiles of the attention backward: instantiations for n <= 145, <= 209, <= 257).
 * A GATConv bias (DGL >= 0.7 checkpoints) is not part of the image: in training mode BatchNorm-1's batch mean absorbs
 * it (prediction unchanged, gradient exactly zero); the caller adds it to the BatchNorm-1 batch mean it folds into
 * running_mean (gnngls_amd/models.py:_TrainStep does). */
int64_t gnngls_regret_train_workspace_bytes(int B, int n, int n_layers);
int gnngls_regret_train_forward(const float *feat, const float *params, int B, int n, int in_dim, int n_layers, float bn_eps,
                                float *y_out, float *bn_batch_stats, void *workspace, int64_t workspace_bytes, void *stream);
int gnngls_regret_train_backward(const float *feat, const float *params, const float *dy, int B, int n, int in_dim,
                                 int n_layers, float *grads, void *workspace, int64_t workspace_bytes, void *stream);

/* get_scaled_features (datasets.py:73-95) for features=[weight] (datasets.py:14-20):
 * feat[b, rank(i<j)] = MinMaxScaler.transform(float32(D[b,i,j])) with sklearn's fp32 arithmetic
 * (x*scale_ rounded to fp32, + min_ rounded to fp32). */
int gnngls_pack_features(const double *D, int B, int n, double scale, double min_, float *feat, void *stream);

/* test.py:79-83: regret_pred = max(MinMaxScaler.inverse_transform(y_pred), 0) as a symmetric fp64
 * [B,n,n] matrix (zero diagonal) -- the 'regret_pred' guide of guided_local_search. */
int gnngls_unpack_regret(const float *y, int B, int n, double scale, double min_, double *out, void *stream);

/* Test hook: lowers the value at which a 16-bit penalty counter reports overflow (default 65535) so
 * the overflow -> rerun-with-32-bit path can be exercised in seconds.  Never called by the product. */
int gnngls_debug_set_penalty16_limit(int limit);

/* Experiment hook: forces the workgroup size of gnngls_gls_run (64, 128, 256 or 512 threads; 0 = the default policy by
 * instance size).  Used by scripts/probe_gls.py to measure the policy; never called by the product. */
int gnngls_debug_set_gls_threads(int threads);

/* Register budget of the kernel instantiation gnngls_gls_run would launch for (n, B, penalty_bits), as resident wavefronts
 * per SIMD: 4 = the 128-VGPR builds (no scratch), 2 = the 256-VGPR build of the single-wavefront kernel (n = 8 .. 33, at
 * most 2048 instances: TSP20 x 1000; no scratch) -- every BASELINE.json shape runs on these two --, 6 / 8 = the 80- / 64-VGPR
 * builds (56-148 B of scratch per lane) that only batches of small instances beyond 16 workgroups per CU select
 * (e.g. TSP20 x 5000).  0 = bad argument. */
int gnngls_gls_waves_per_simd(int n, int B, int penalty_bits);

/* 1 if gnngls_gls_run would run the perturbation phase of this (n, B, penalty_bits) on ALL wavefronts of the workgroup
 * (the "team" form: the four one-to-all scans of a penalty step, algorithms.py:167-174, evaluated concurrently and consumed
 * in the reference's order), 0 if on wavefront 0 only.  Since round 5 the policy picks it for FIRST-IMPROVEMENT runs only
 * (16-wave workgroups that own a CU by their LDS footprint -- compact store, n >= 144: TSP200 -- when B <= number of CUs):
 * for best-improvement runs the edge form of the serial phase is faster there too, so this query (a best-improvement
 * answer) returns 0 unless the test hook below forces the form; gnngls_gls_describe_run reports a first-improvement launch.
 * Same results either way (bit-exact); this only reports the policy. */
int gnngls_gls_uses_team(int n, int B, int penalty_bits);

/* Experiment / test hook: -1 = the policy above (default), 0 = never use the team form, 1 = use it wherever it exists
 * (symmetric stores with 32-bit counters, n <= 255) whatever the batch size.  Never called by the product. */
int gnngls_debug_set_gls_team(int mode);

/* Experiment / test hook: 0 = the descent (algorithms.py:111-132) evaluates every move of an all-to-all scan; -1 / 1
 * (default) = the pruned scans where they exist (best improvement, symmetric LDS stores, max |D| <= 1e6; the 2-opt scan from
 * n = 80, the relocate scan from n = 128): per
 * node a list of its 32 nearest nodes, only moves that can qualify are evaluated -- a superset of the qualifying moves,
 * hence the same arg-min, bit-exact.  Never called by the product. */
int gnngls_debug_set_gls_prune(int mode);

/* Diagnostic hook: device buffer of 16 int64 per instance that a library built with -DGLS_STAMPS fills with
 * per-phase shader-cycle totals of gnngls_gls_run (scripts/probe_gls_stamps.py).  Ignored by normal builds. */
int gnngls_debug_set_stamp_buffer(void *device_buffer);

/* ---- measurement hooks (bench.py): per-kernel-class device time via HIP events recorded on the
 * caller's stream around every launch made while profiling is enabled.  gnngls_profile_collect
 * synchronises the recorded events, sums milliseconds and launch counts per class (arrays of
 * GNNGLS_PROF_KINDS entries) and clears the log. */
enum {
    GNNGLS_PROF_PACK = 0, GNNGLS_PROF_EMBED, GNNGLS_PROF_GEMM_FC, GNNGLS_PROF_GAT_ROWS, GNNGLS_PROF_GAT_ROWS_RANK1 /* (the slot of the retired gat_combine) */,
    GNNGLS_PROF_GEMM_FFN1, GNNGLS_PROF_GEMM_FFN2, GNNGLS_PROF_DECISION, GNNGLS_PROF_UNPACK,
    GNNGLS_PROF_NEAREST_NEIGHBOR, GNNGLS_PROF_TOUR_COST, GNNGLS_PROF_GLS, GNNGLS_PROF_FFN_FUSED,
    GNNGLS_PROF_TRAIN_COLSUM, GNNGLS_PROF_TRAIN_ELEMENTWISE, GNNGLS_PROF_TRAIN_GEMM_BWD, GNNGLS_PROF_TRAIN_GEMM_TN,
    GNNGLS_PROF_TRAIN_GAT_BWD, GNNGLS_PROF_KINDS
};
int gnngls_profile_enable(int on);
int gnngls_profile_collect(double *ms_by_kind, int64_t *launches_by_kind);

/* Measurement hook (bench.py): while a device buffer is set (NULL = off, the default), every gnngls_gls_run launch fills its
 * first B int64 entries with the number of delta evaluations the kernel actually EXECUTED per instance.
 * evals_out of gnngls_gls_run counts what the reference evaluates (operators.py:36-39,133-136: every move of every scan);
 * the pruned descent scans (n >= 80) evaluate only the moves that can qualify, the full scans and the one-to-all scans of
 * the perturbation phase every move once -- so executed <= evals_out, equal where nothing is pruned.  Counting costs the
 * pruned scans 2-3 % (profiles/r04_ab_exec_counter.log), so it lives in kernel instantiations of their own that ONLY a
 * launch with this hook set selects: compact store (penalty_bits 0 / -2 where that store is picked), best improvement, no
 * per-move trace.  A pruning run on any other configuration reports -1 per instance.
 * ABI v3: the buffer holds 5 x B int64 -- [0, B) the counts above, then four records of the same launch for bench.py's
 * critical-path figure (written by the counting instantiations, else 0): [B, 2B) shader cycles of the instance's workgroup,
 * [2B, 3B) shader cycles of its serial perturbation phase (algorithms.py:150-185, wavefront 0), [3B, 4B) penalty steps of
 * that phase (edge form), [4B, 5B) 100 MHz device-clock ticks of the workgroup.
 * ABI v4: GNNGLS_EXEC_RECORDS = 16 records of B int64 -- after the five above the cycle account of the descent of the outer
 * iterations (algorithms.py:188 -> 111-132; wavefront 0 of the workgroup): [5B, 6B) shader cycles in local_search, then count and
 * cycles of the two_opt_a2a scans [6B, 8B), of the relocate_a2a scans evaluated in full [8B, 10B) and over the flagged rows only
 * [10B, 12B) (operators.py:32-50,129-147), [12B, 13B) cycles of the workgroup arg-min incl. waiting for the slowest wavefront,
 * [13B, 14B) of move application + barrier (algorithms.py:122-126), [14B, 15B) accepted moves, [15B, 16B) reserved.
 * The buffer must hold GNNGLS_EXEC_RECORDS x the largest B launched while it is set; process-wide, not per stream. */
#define GNNGLS_EXEC_RECORDS 16
int gnngls_profile_set_executed_evals(int64_t *device_buffer);

#ifdef __cplusplus
}
#endif
#endif /* GNNGLS_HIP_H */
