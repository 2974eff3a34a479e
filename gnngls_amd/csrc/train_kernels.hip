// train_kernels.hip -- one optimisation step of the edge-regret GNN (forward in training mode + backward) on MI355X
// (gfx950), hand-written HIP, fp32 with fp64 column reductions.
//
// Replaces (reference file:line, /root/reference/...):
//   scripts/train.py:20-32   model.train(); y_pred = model(batch, x); loss.backward()   (the autograd graph of
//                            gnngls/models.py:5-70 over a dgl.batch of line graphs, train.py:118-121)
//   gnngls/models.py:27,35   nn.BatchNorm1d in training mode: batch statistics over ALL nodes of the batched graph
//   gnngls/models.py:23      dgl.nn.GATConv backward (edge softmax, u_mul_e sum) -- third party, DGL 0.6.1
//
// The forward reuses the inference kernels (model_kernels.hip: gemm_f32_kernel, gat_rows_kernel, ffn_fused_kernel) and
// adds what training needs: the merged attention output and its softmax statistics are kept, BatchNorm uses batch
// statistics (two-stage fp64 column sums, deterministic).  The backward is new:
//   gat_bwd_rows_kernel : one workgroup per (instance, TSP row u), the mirror image of gat_rows_kernel.  ft and dOut of
//                         the row's n-1 line-graph nodes are staged in LDS once; per (source tile, head) the kernel forms
//                         T = dOut_h * ft_h^T on the MFMA, turns it into the softmax-backward term in registers
//                         (a_ij recomputed from the saved (max, 1/Z)), and feeds the attention weights straight back
//                         into a second MFMA chain a^T * dOut (the accumulator layout of the first product IS the
//                         A-operand layout of the second, the same trick ffn_fused_kernel uses).
//   gemm_tn_kernel      : weight gradients dW = X^T * Y (reduction over the B*N rows split across workgroups, fp32 MFMA
//                         partial tiles, fp64 fixed-order final sum)
//   colsum_kernel       : every per-column reduction of the step (BatchNorm statistics, BatchNorm backward sums, bias
//                         gradients, attention-vector gradients) in one templated two-stage kernel
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "train_kernels.h"

namespace gnngls {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int kD = 128, kH = 8, kF = 16;
constexpr float kSlope = 0.2f;

__device__ __forceinline__ int tri_index(int i, int j, int n) {   // i < j, rank in itertools.combinations order
    return i * n - ((i * (i + 1)) >> 1) + (j - i - 1);
}

int grid_cap(long want, int cap) {
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    return (int)want;
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// Column reductions: partial[block][k][c] = sum over the block's rows of f_k(m, c), k = 0,1, accumulated in fp64.
//   CS_SUM_SQ    f0 = X          f1 = X*X                        (BatchNorm batch statistics)
//   CS_SUM_PROD  f0 = X          f1 = X*Y[m,c]                   (BatchNorm backward: X = dy, Y = x; bias gradients)
//   CS_ROWSCALE  f0 = X*Y[m*ys]  f1 = X                          (decision / embed weight gradients, Y a per-row scalar)
//   CS_HEADSCALE f0 = X*Y[m,c/16] f1 = X*Y2[m,c/16]              (attn_l / attn_r gradients: X = ft, Y = d el, Y2 = d er)
// C = 128 or 512 columns; 256 threads = (C/4 column groups) x (1024/C row lanes).
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ X, const float *__restrict__ Y,
                                                     const float *__restrict__ Y2, long M, int C, int ystride,
                                                     double *__restrict__ partial) {
    __shared__ double red[2 * 1024];                 // [k][row lane][C]  (row lanes * C = 1024)
    const int tid = threadIdx.x;
    const int cg = C >> 2, rl = 256 / cg;
    const int c = (tid % cg) * 4, r = tid / cg;
    double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
    for (long m = (long)blockIdx.x * rl + r; m < M; m += (long)gridDim.x * rl) {
        const f32x4 x = *reinterpret_cast<const f32x4 *>(X + m * C + c);
        if (MODE == CS_SUM_SQ) {
#pragma unroll
            for (int v = 0; v < 4; ++v) { a0[v] += (double)x[v]; a1[v] += (double)x[v] * (double)x[v]; }
        } else if (MODE == CS_SUM_PROD) {
            f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f};
            if (Y) y = *reinterpret_cast<const f32x4 *>(Y + m * C + c);
#pragma unroll
            for (int v = 0; v < 4; ++v) { a0[v] += (double)x[v]; a1[v] += (double)x[v] * (double)y[v]; }
        } else if (MODE == CS_ROWSCALE) {
            const double s = (double)Y[m * ystride];
#pragma unroll
            for (int v = 0; v < 4; ++v) { a0[v] += (double)x[v] * s; a1[v] += (double)x[v]; }
        } else {
            const double s = (double)Y[m * kH + (c >> 4)], s2 = (double)Y2[m * kH + (c >> 4)];
#pragma unroll
            for (int v = 0; v < 4; ++v) { a0[v] += (double)x[v] * s; a1[v] += (double)x[v] * s2; }
        }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) { red[r * C + c + v] = a0[v]; red[1024 + r * C + c + v] = a1[v]; }
    __syncthreads();
    for (int q = tid; q < 2 * C; q += 256) {
        const int k = q / C, cc = q % C;
        double s = 0.0;
        for (int rr = 0; rr < rl; ++rr) s += red[k * 1024 + rr * C + cc];
        partial[((long)blockIdx.x * 2 + k) * C + cc] = s;
    }
}

// Final stage: 1024 threads, thread (c, part) sums every (1024/C)-th partial of column c (independent loads in flight),
// the parts meet in LDS; returns the two column sums to the threads with part == 0 (tid < C).
__device__ __forceinline__ void sum_partials(const double *__restrict__ partial, int nblocks, int C, double *red,
                                             double &s0, double &s1) {
    const int tid = threadIdx.x, c = tid % C, part = tid / C, P = 1024 / C;
    double a0 = 0.0, a1 = 0.0;
#pragma unroll 8
    for (int b = part; b < nblocks; b += P) {
        a0 += partial[((long)b * 2) * C + c];
        a1 += partial[((long)b * 2 + 1) * C + c];
    }
    red[tid] = a0; red[1024 + tid] = a1;
    __syncthreads();
    s0 = 0.0; s1 = 0.0;
    if (part == 0)
        for (int q = 0; q < P; ++q) { s0 += red[q * C + c]; s1 += red[1024 + q * C + c]; }
}

// out0[c] = sum f0 (if out0), out1[c] = sum f1 (if out1), rounded to fp32 -- plain gradient vectors
// (out0 with element stride `ostride`: a column of the [128, in_dim] embedding weight)
__global__ __launch_bounds__(1024) void colsum_store_kernel(const double *partial, int nblocks, int C, int ostride, float *out0,
                                                            float *out1) {
    __shared__ double red[2048];
    double s0, s1;
    sum_partials(partial, nblocks, C, red, s0, s1);
    const int c = threadIdx.x;
    if (c >= C) return;
    if (out0) out0[(long)c * ostride] = (float)s0;
    if (out1) out1[c] = (float)s1;
}

// out[0] = sum_m v[m] in fp64 (decision bias gradient, out_dim = 1); one workgroup
__global__ __launch_bounds__(1024) void sum_vector_kernel(const float *__restrict__ v, long M, float *out) {
    __shared__ double red[1024];
    double s = 0.0;
    for (long m = threadIdx.x; m < M; m += 1024) s += (double)v[m];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)red[0];
}

// BatchNorm1d training forward (models.py:27,35; torch: biased variance for the normalisation, unbiased for the running
// estimate): scale = gamma * invstd, shift = beta - mean * scale; saves mean / invstd for the backward and hands the
// batch mean / unbiased variance to the caller (running-statistics update)
__global__ __launch_bounds__(1024) void bn_stats_finalize_kernel(const double *partial, int nblocks, long M, const float *gamma,
                                                                 const float *beta, float eps, float *scale, float *shift,
                                                                 float *mean_out, float *invstd_out, float *batch_mean,
                                                                 float *batch_var_unbiased) {
    __shared__ double red[2048];
    double s0, s1;
    sum_partials(partial, nblocks, kD, red, s0, s1);
    const int c = threadIdx.x;
    if (c >= kD) return;
    const double mean = s0 / (double)M;
    double var = s1 / (double)M - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const double sc = (double)gamma[c] * invstd;
    scale[c] = (float)sc;
    shift[c] = (float)((double)beta[c] - mean * sc);
    mean_out[c] = (float)mean;
    invstd_out[c] = (float)invstd;
    batch_mean[c] = (float)mean;
    batch_var_unbiased[c] = (float)(M > 1 ? var * (double)M / (double)(M - 1) : var);
}

// BatchNorm backward: dgamma = invstd * (sum dy*x - mean * sum dy), dbeta = sum dy,
//   dx = A*dy + Bc*(x - mean) + Cc   with A = gamma*invstd, Bc = -A*invstd*dgamma/M, Cc = -A*dbeta/M
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const double *partial, int nblocks, long M, const float *gamma,
                                                               const float *mean, const float *invstd, float *dgamma,
                                                               float *dbeta, float *coef /*[3][128]*/) {
    __shared__ double red[2048];
    double sdy, sdyx;
    sum_partials(partial, nblocks, kD, red, sdy, sdyx);
    const int c = threadIdx.x;
    if (c >= kD) return;
    const double is = (double)invstd[c], mu = (double)mean[c];
    const double dg = is * (sdyx - mu * sdy);
    dgamma[c] = (float)dg;
    dbeta[c] = (float)sdy;
    const double A = (double)gamma[c] * is;
    coef[c] = (float)A;
    coef[kD + c] = (float)(-A * is * dg / (double)M);
    coef[2 * kD + c] = (float)(-A * sdy / (double)M);
}

__global__ void bn_bwd_apply_kernel(const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ mean,
                                    const float *__restrict__ coef, float *__restrict__ dx, long M) {
    const long total = M * (kD / 4);
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const int c = (int)(q & 31) * 4;
        const f32x4 d = *reinterpret_cast<const f32x4 *>(dy + q * 4);
        const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + q * 4);
        f32x4 o;
#pragma unroll
        for (int v = 0; v < 4; ++v)
            o[v] = coef[c + v] * d[v] + coef[kD + c + v] * (xv[v] - mean[c + v]) + coef[2 * kD + c + v];
        *reinterpret_cast<f32x4 *>(dx + q * 4) = o;
    }
}

// The two elementwise passes that open a layer's backward, one launch: d h3 = BatchNorm-2 backward of dy (as
// bn_bwd_apply_kernel) and x = BatchNorm-1 applied to h1 (as affine_cols_kernel), both [M,128]
__global__ void bn_bwd_apply_and_affine_kernel(const float *__restrict__ dy, const float *__restrict__ h3,
                                               const float *__restrict__ mean, const float *__restrict__ coef,
                                               float *__restrict__ dx, const float *__restrict__ h1,
                                               const float *__restrict__ scale, const float *__restrict__ shift,
                                               float *__restrict__ xout, long M) {
    const long total = M * (kD / 4);
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const int c = (int)(q & 31) * 4;
        const f32x4 d = *reinterpret_cast<const f32x4 *>(dy + q * 4);
        const f32x4 xv = *reinterpret_cast<const f32x4 *>(h3 + q * 4);
        const f32x4 hv = *reinterpret_cast<const f32x4 *>(h1 + q * 4);
        f32x4 o, x2;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            o[v] = coef[c + v] * d[v] + coef[kD + c + v] * (xv[v] - mean[c + v]) + coef[2 * kD + c + v];
            x2[v] = hv[v] * scale[c + v] + shift[c + v];
        }
        *reinterpret_cast<f32x4 *>(dx + q * 4) = o;
        *reinterpret_cast<f32x4 *>(xout + q * 4) = x2;
    }
}

// out = x * scale[c] + shift[c]   (BatchNorm apply with precomputed affine)
__global__ void affine_cols_kernel(const float *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift,
                                   float *__restrict__ out, long M) {
    const long total = M * (kD / 4);
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const int c = (int)(q & 31) * 4;
        const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + q * 4);
        f32x4 o;
#pragma unroll
        for (int v = 0; v < 4; ++v) o[v] = xv[v] * scale[c + v] + shift[c + v];
        *reinterpret_cast<f32x4 *>(out + q * 4) = o;
    }
}

// decision layer backward wrt its input (models.py:69, out_dim = 1): dh[m,c] = dy[m] * w[c]
__global__ void outer_rows_kernel(const float *__restrict__ dy, const float *__restrict__ w, float *__restrict__ dh, long M) {
    const long total = M * (kD / 4);
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const long m = q >> 5;
        const int c = (int)(q & 31) * 4;
        const float s = dy[m];
        f32x4 o;
#pragma unroll
        for (int v = 0; v < 4; ++v) o[v] = s * w[c + v];
        *reinterpret_cast<f32x4 *>(dh + q * 4) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// Training-mode merge of the two attention partials written by gat_rows_kernel (the inference path fuses this into
// ffn_fused_kernel): g = GATConv(h) (kept for the backward), h1 = h + g (models.py:15), att[m] = (row max, 1/Z) per head.
// ---------------------------------------------------------------------------------------------
__global__ void gat_combine_train_kernel(const float *__restrict__ part, const float *__restrict__ part_ms,
                                         const float *__restrict__ h, long M, float *__restrict__ g,
                                         float *__restrict__ h1, float *__restrict__ att) {
    const long total = M * (kD / 4);
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const long m = q >> 5;
        const int c4 = (int)(q & 31), hd = c4 >> 2;
        const float *ms0 = part_ms + m * (2 * kH), *ms1 = part_ms + (M + m) * (2 * kH);
        const float m0 = ms0[hd], s0 = ms0[kH + hd], m1 = ms1[hd], s1 = ms1[kH + hd];
        const float mx = m0 > m1 ? m0 : m1;
        const float a0 = __expf(m0 - mx), a1 = __expf(m1 - mx);
        const float inv = 1.f / (s0 * a0 + s1 * a1);
        const f32x4 p0 = *reinterpret_cast<const f32x4 *>(part + q * 4);
        const f32x4 p1 = *reinterpret_cast<const f32x4 *>(part + (M * kD) + q * 4);
        const f32x4 hv = *reinterpret_cast<const f32x4 *>(h + q * 4);
        f32x4 gv, o;
#pragma unroll
        for (int v = 0; v < 4; ++v) { gv[v] = (p0[v] * a0 + p1[v] * a1) * inv; o[v] = hv[v] + gv[v]; }
        *reinterpret_cast<f32x4 *>(g + q * 4) = gv;
        *reinterpret_cast<f32x4 *>(h1 + q * 4) = o;
        if ((c4 & 3) == 0) { att[m * (2 * kH) + hd] = mx; att[m * (2 * kH) + kH + hd] = inv; }
    }
}

// ---------------------------------------------------------------------------------------------
// GATConv backward over the line graph of K_n, one workgroup per (instance b, TSP row u).
// The row's line-graph nodes {u,k} (slot s = k < u ? k : k-1, ns = n-1 slots) are at once the destinations i and the
// sources j of all arcs that share endpoint u.  Given dOut (gradient of the GATConv output), per head h:
//     a_ij  = exp(LeakyReLU(el_j + er_i) - max_i) / Z_i                     (max_i, 1/Z_i saved by the forward)
//     t_ij  = <dOut_i, ft_j>_h ,  c_i = <dOut_i, out_i>_h = sum_j a_ij t_ij
//     ds_ij = a_ij (t_ij - c_i) * LeakyReLU'(el_j + er_i)                   (softmax + LeakyReLU backward)
//     P_j   = sum_i a_ij dOut_i      del_j = sum_i ds_ij      der_i = sum_j ds_ij      (i != j)
// written to side (u < k ? 0 : 1) of the partial buffers; the other endpoint's row supplies the second half.
// gat_bwd_combine_kernel then forms  dft = P + del*attn_l + der*attn_r  (el = <ft,attn_l>, er = <ft,attn_r>).
// ---------------------------------------------------------------------------------------------
constexpr int kGatBwdHeads = 2;                    // heads per workgroup (one wave each): the LDS tile of a workgroup covers
                                                   // 16*kGatBwdHeads columns of ft / dOut, so several workgroups share a CU
                                                   // and the staging of one overlaps the MFMA phase of the others
constexpr int kGatBwdThreads = 64 * kGatBwdHeads;
constexpr int kGatBwdMaxTiles = 16;                // 16-node source tiles per row the largest instantiation holds: n - 1 <= 256
constexpr int LDG = 16 * kGatBwdHeads + 4;         // LDS row stride (floats): a ds_read_b128 of 16 consecutive rows at one
                                                   // column offset touches 16 disjoint groups of 4 banks; 4 rows 4 apart
                                                   // (MFMA B fragment) land on disjoint 16-bank groups  (36, 68, 132)

__device__ __forceinline__ float row16_sum(float v) {   // inclusive prefix over the 16-lane DPP row; lane 15 holds the total
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));   // row_shr:1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true));   // row_shr:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true));   // row_shr:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, true));   // row_shr:8
    return v;
}

// (Measured and rejected: the factorised exponentials of gat_rows_kernel here, with its wave-uniform direct-path branch in
// the unrolled tile loop: 3.3 -> 4.0 ms per step; taking the source tiles two at a time with both T chains issued ahead of the arithmetic, and
// skipping the self-loop mask on interior tiles -- 3.4 -> 3.6 ms per step at TSP100 x 32, more registers, no gain.)
// Workgroup = (instance b, TSP row u, head group); wave w owns head h = group * kGatBwdHeads + w.  Outer loop:
// destination tiles dt (dOut fragments, softmax statistics and the B fragments of the second product are fetched once
// per tile); inner loop: source tiles st, whose P / del accumulators stay in registers for the whole kernel
// (kGatBwdMaxTiles x 5 VGPRs), while der of the current destination tile accumulates across st and is reduced over the
// 16 source lanes once per tile.
// MAXT = source tiles the instantiation keeps accumulators for (5 VGPRs each): 9 (n <= 145, the reference's training sizes
// up to TSP100), 13 (n <= 209: TSP200), 16 (n <= 257); the launcher picks the smallest that covers n.
template <int MAXT>
__global__ __launch_bounds__(kGatBwdThreads) void gat_bwd_rows_kernel(const float *__restrict__ ft, const float *__restrict__ dout,
                                                                      const float *__restrict__ gout, const float *__restrict__ att,
                                                                      const float *__restrict__ attn_l, const float *__restrict__ attn_r,
                                                                      int n, float *__restrict__ P, float *__restrict__ dlr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int HG = kGatBwdHeads, CW = 16 * HG;                // heads / columns per workgroup
    constexpr int groups = kH / HG;
    const int N = n * (n - 1) / 2;
    const int ns = n - 1, nt = (ns + 15) >> 4;
    const int grp = blockIdx.x % groups, bu = blockIdx.x / groups;
    const int b = bu / n, u = bu % n;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int h = grp * HG + w, c0 = grp * CW;
    f32x4 *stS = reinterpret_cast<f32x4 *>(smem);                 // [ns][HG] (er, -max*log2e, 1/Z, c)   (16-byte rows first)
    float *ftS = reinterpret_cast<float *>(stS + (size_t)ns * HG);  // [ns][LDG]  columns c0 .. c0+CW of ft
    float *dgS = ftS + (size_t)ns * LDG;                          // [ns][LDG]
    float *elS = dgS + (size_t)ns * LDG;                          // [ns][HG]
    int *nodeS = reinterpret_cast<int *>(elS + (size_t)ns * HG);  // [ns] line-graph node of slot s
    const float kLog2e = 1.4426950408889634f;
    const size_t Mtot = (size_t)gridDim.x / (n * groups) * N;     // B*N rows per side

    const size_t base = (size_t)b * N;
    for (int s = tid; s < ns; s += kGatBwdThreads) {
        const int k = s < u ? s : s + 1;
        nodeS[s] = k < u ? tri_index(k, u, n) : tri_index(u, k, n);
    }
    __syncthreads();
    // (the global loads of a batch are all issued before the first LDS store: one round trip per batch, not per row)
    constexpr int SB = 4;
    for (int q0 = tid; q0 < ns * (CW / 4); q0 += kGatBwdThreads * SB) {
        f32x4 vf[SB], vd[SB];
#pragma unroll
        for (int u2 = 0; u2 < SB; ++u2) {
            const int q = q0 + u2 * kGatBwdThreads;
            if (q < ns * (CW / 4)) {
                const size_t row = (base + nodeS[q / (CW / 4)]) * kD + c0 + (q % (CW / 4)) * 4;
                vf[u2] = *reinterpret_cast<const f32x4 *>(ft + row);
                vd[u2] = *reinterpret_cast<const f32x4 *>(dout + row);
            }
        }
#pragma unroll
        for (int u2 = 0; u2 < SB; ++u2) {
            const int q = q0 + u2 * kGatBwdThreads;
            if (q < ns * (CW / 4)) {
                const int s = q / (CW / 4), c = (q % (CW / 4)) * 4;
                *reinterpret_cast<f32x4 *>(ftS + (size_t)s * LDG + c) = vf[u2];
                *reinterpret_cast<f32x4 *>(dgS + (size_t)s * LDG + c) = vd[u2];
            }
        }
    }
    __syncthreads();
    {   // el, er and c = <dOut, out> per (slot, head); a thread's head is fixed (the stride is a multiple of HG)
        const int hl = tid % HG, hh = grp * HG + hl;
        f32x4 al[4], ar[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            al[v] = *reinterpret_cast<const f32x4 *>(attn_l + hh * kF + 4 * v);
            ar[v] = *reinterpret_cast<const f32x4 *>(attn_r + hh * kF + 4 * v);
        }
        for (int q = tid; q < ns * HG; q += kGatBwdThreads) {
            const int s = q / HG;
            const float *f = ftS + (size_t)s * LDG + hl * kF;
            const float *d = dgS + (size_t)s * LDG + hl * kF;
            const float *gop = gout + (base + nodeS[s]) * kD + hh * kF;
            const float *a = att + (base + nodeS[s]) * (2 * kH);
            f32x4 go4[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) go4[v] = *reinterpret_cast<const f32x4 *>(gop + 4 * v);
            const float a_max = a[hh], a_zinv = a[kH + hh];
            float l = 0.f, r = 0.f, c = 0.f;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const f32x4 fv = *reinterpret_cast<const f32x4 *>(f + 4 * v);
                const f32x4 dv = *reinterpret_cast<const f32x4 *>(d + 4 * v);
#pragma unroll
                for (int u2 = 0; u2 < 4; ++u2) {
                    l = fmaf(fv[u2], al[v][u2], l);
                    r = fmaf(fv[u2], ar[v][u2], r);
                    c = fmaf(dv[u2], go4[v][u2], c);
                }
            }
            elS[q] = l;
            stS[q] = f32x4{r, -a_max * kLog2e, a_zinv, c};
        }
    }
    __syncthreads();

    const int jl = lane & 15, q4 = lane >> 4;
    // output row of slot s with the side folded in: slot s is node {u,k}, k = s < u ? s : s+1; side 0 iff u < k iff u <= s
    auto out_row = [&](int s) -> size_t { return ((u <= s) ? 0 : Mtot) + base + nodeS[s]; };
    f32x4 accP[MAXT];
    float del[MAXT];
#pragma unroll
    for (int st = 0; st < MAXT; ++st) { accP[st] = f32x4{0.f, 0.f, 0.f, 0.f}; del[st] = 0.f; }

    for (int dt = 0; dt < nt; ++dt) {
        const int ia = dt * 16 + jl, iac = ia < ns ? ia : ns - 1;
        // A operand of T = dOut * ft^T: lane (jl, q4) supplies dOut[i = 16dt + jl][16h + 4*q4 + ks] (one ds_read_b128)
        const f32x4 adg = *reinterpret_cast<const f32x4 *>(dgS + (size_t)iac * LDG + w * kF + 4 * q4);
        f32x4 sv[4];
        float bdg[4], der[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = dt * 16 + 4 * q4 + r, ic = i < ns ? i : ns - 1;
            sv[r] = stS[ic * HG + w];
            bdg[r] = dgS[(size_t)ic * LDG + w * kF + jl];
            der[r] = 0.f;
        }
#pragma unroll
        for (int st = 0; st < MAXT; ++st) {
            if (st < nt) {
                const int j = st * 16 + jl, jc = j < ns ? j : ns - 1;
                const float el_j = elS[jc * HG + w];
                // B operand: lane (jl, q4) supplies ft[j][16h + 4*q4 + ks]
                const f32x4 bft = *reinterpret_cast<const f32x4 *>(ftS + (size_t)jc * LDG + w * kF + 4 * q4);
                f32x4 T = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) T = __builtin_amdgcn_mfma_f32_16x16x4f32(adg[ks], bft[ks], T, 0, 0, 0);
                // T[r] = t_ij for destination i = 16*dt + 4*q4 + r, source j = 16*st + jl
                float av[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = dt * 16 + 4 * q4 + r;
                    const float x = el_j + sv[r][0];
                    const float lx = fmaxf(x, kSlope * x);
                    float a = __builtin_amdgcn_exp2f(fmaf(lx, kLog2e, sv[r][1])) * sv[r][2];
                    const bool live = (i < ns) && (j < ns) && (i != j);
                    a = live ? a : 0.f;
                    const float ds = a * (T[r] - sv[r][3]) * (x > 0.f ? 1.f : kSlope);
                    av[r] = a;
                    del[st] += ds;
                    der[r] += ds;
                }
                // P[j][:] += sum_i a_ij dOut_i[:]: the T accumulator layout (row = 4*q4 + r, col = jl) is the A-operand
                // layout of step r (row = jl, k = q4) of a^T * dOut
#pragma unroll
                for (int r = 0; r < 4; ++r) accP[st] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bdg[r], accP[st], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float rs = row16_sum(der[r]);
            const int i = dt * 16 + 4 * q4 + r;
            if (jl == 15 && i < ns) dlr[out_row(i) * (2 * kH) + kH + h] = rs;
        }
    }
#pragma unroll
    for (int st = 0; st < MAXT; ++st) {
        if (st < nt) {
            float d = del[st];
            d += __shfl_xor(d, 16, 64);
            d += __shfl_xor(d, 32, 64);
            const int j = st * 16 + jl;
            if (q4 == 0 && j < ns) dlr[out_row(j) * (2 * kH) + h] = d;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jr = st * 16 + 4 * q4 + r;
                if (jr < ns) P[out_row(jr) * kD + h * kF + jl] = accP[st][r];
            }
        }
    }
}

// dft = P0 + P1 + (del0 + del1) * attn_l + (der0 + der1) * attn_r ; dl / dr = the summed d el / d er (for the attn
// gradients, colsum CS_HEADSCALE)
__global__ void gat_bwd_combine_kernel(const float *__restrict__ P, const float *__restrict__ dlr,
                                       const float *__restrict__ attn_l, const float *__restrict__ attn_r, long M,
                                       float *__restrict__ dft, float *__restrict__ dl, float *__restrict__ dr) {
    const long total = M * (kD / 4);
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const long m = q >> 5;
        const int c4 = (int)(q & 31), hd = c4 >> 2, c = c4 * 4;
        const float *d0 = dlr + m * (2 * kH), *d1 = dlr + (M + m) * (2 * kH);
        const float l = d0[hd] + d1[hd], r = d0[kH + hd] + d1[kH + hd];
        const f32x4 p0 = *reinterpret_cast<const f32x4 *>(P + q * 4);
        const f32x4 p1 = *reinterpret_cast<const f32x4 *>(P + M * kD + q * 4);
        f32x4 o;
#pragma unroll
        for (int v = 0; v < 4; ++v) o[v] = (p0[v] + p1[v]) + l * attn_l[c + v] + r * attn_r[c + v];
        *reinterpret_cast<f32x4 *>(dft + q * 4) = o;
        if ((c4 & 3) == 0) { dl[m * kH + hd] = l; dr[m * kH + hd] = r; }
    }
}

// ---------------------------------------------------------------------------------------------
// Weight gradients: C[N1,N2] = X[M,N1]^T * Y[M,N2]; workgroup (chunk, tile1, tile2) reduces its slice of the M rows
// into a 128x128 fp32 MFMA tile (partial[chunk][N1][N2]); gemm_tn_reduce_kernel sums the chunks in fp64, fixed order.
// Both operands are read row-major: the k-major LDS image of gemm_f32_kernel is a straight copy here.
// ---------------------------------------------------------------------------------------------
constexpr int TBK = 32, TLD = 129;

// xsum (optional): the column sums of X over the chunk's rows, [chunk][N1] -- the bias gradient that goes with the weight
// gradient (X = d output of the Linear), taken from the LDS image of X the tile loop has staged anyway.
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float *__restrict__ X, const float *__restrict__ Y, long M, int N1,
                                                      int N2, long rows_per_chunk, float *__restrict__ partial,
                                                      float *__restrict__ xsum) {
    __shared__ float Xs[TBK * TLD];
    __shared__ float Ys[TBK * TLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c1 = blockIdx.y * 128, c2 = blockIdx.z * 128;
    const long m_lo = (long)blockIdx.x * rows_per_chunk;
    long m_hi = m_lo + rows_per_chunk;
    if (m_hi > M) m_hi = M;
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][bb][r] = 0.f;
    const int tk = tid >> 5, tc = (tid & 31) * 4;
    f32x4 rx[4], ry[4];
    auto gload = [&](long m0) {
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            const long m = m0 + 8 * uu + tk;
            const bool ok = m < m_hi;
            rx[uu] = ok ? *reinterpret_cast<const f32x4 *>(X + m * N1 + c1 + tc) : f32x4{0.f, 0.f, 0.f, 0.f};
            ry[uu] = ok ? *reinterpret_cast<const f32x4 *>(Y + m * N2 + c2 + tc) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int uu = 0; uu < 4; ++uu)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                Xs[(8 * uu + tk) * TLD + tc + c] = rx[uu][c];
                Ys[(8 * uu + tk) * TLD + tc + c] = ry[uu][c];
            }
    };
    const bool want_xsum = xsum != nullptr && blockIdx.z == 0;
    float xs = 0.f;
    if (m_lo < m_hi) gload(m_lo);
    for (long m0 = m_lo; m0 < m_hi; m0 += TBK) {
        __syncthreads();
        lstore();
        __syncthreads();
        if (m0 + TBK < m_hi) gload(m0 + TBK);
        if (want_xsum) {                                   // thread -> column tid & 127, k half tid >> 7 (rows past m_hi are zero)
            const float *xc = Xs + (tid >> 7) * 16 * TLD + (tid & 127);
#pragma unroll
            for (int k = 0; k < 16; ++k) xs += xc[k * TLD];
        }
        const int lr = lane & 31, lk = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < TBK; kk += 2) {
            const float *xp = Xs + (kk + lk) * TLD, *yp = Ys + (kk + lk) * TLD;
            const float a0 = xp[wr + lr], a1 = xp[wr + 32 + lr];
            const float b0 = yp[wc + lr], b1 = yp[wc + 32 + lr];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    if (want_xsum) {
        __syncthreads();
        if (tid >= 128) Xs[tid & 127] = xs;
        __syncthreads();
        if (tid < 128) xsum[(size_t)blockIdx.x * N1 + c1 + tid] = xs + Xs[tid];
    }
    float *out = partial + (size_t)blockIdx.x * N1 * N2;
    const int lc = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = c1 + wr + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int col = c2 + wc + tb * 32 + lc;
                out[(size_t)row * N2 + col] = acc[ta][tb][r];
            }
}

// 256 threads = 64 outputs x 4 chunk groups (4x the loads in flight of a one-thread-per-output sum); the groups meet in LDS
// in fixed order.  Two segments in one launch: the E weight-gradient entries, then (optional) the E2 column sums of X.
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const float *__restrict__ partial, int chunks, long E,
                                                             float *__restrict__ out, const float *__restrict__ partial2,
                                                             long E2, float *__restrict__ out2) {
    __shared__ double red[256];
    const int tid = threadIdx.x, el = tid & 63, part = tid >> 6;
    long e = (long)blockIdx.x * 64 + el;
    const float *src = partial;
    float *dst = out;
    long width = E;
    if (e >= E) { e -= (E + 63) / 64 * 64; src = partial2; dst = out2; width = E2; }      // second segment starts on a block boundary
    double s = 0.0;
    if (e >= 0 && e < width) {
#pragma unroll 8
        for (int k = part; k < chunks; k += 4) s += (double)src[(size_t)k * width + e];
    }
    red[tid] = s;
    __syncthreads();
    if (part == 0 && e >= 0 && e < width) dst[e] = (float)(((red[el] + red[64 + el]) + red[128 + el]) + red[192 + el]);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
int colsum_blocks(long M, int C) {
    const int rl = 1024 / C;
    return grid_cap((M + (long)rl * 16 - 1) / ((long)rl * 16), kColsumMaxBlocks);
}

hipError_t launch_colsum(int mode, const float *X, const float *Y, const float *Y2, long M, int C, int ystride,
                         double *partial, int *nblocks, hipStream_t st) {
    const int nb = colsum_blocks(M, C);
    *nblocks = nb;
    (void)hipGetLastError();
    switch (mode) {
    case CS_SUM_SQ: hipLaunchKernelGGL(colsum_kernel<CS_SUM_SQ>, dim3(nb), dim3(256), 0, st, X, Y, Y2, M, C, ystride, partial); break;
    case CS_SUM_PROD: hipLaunchKernelGGL(colsum_kernel<CS_SUM_PROD>, dim3(nb), dim3(256), 0, st, X, Y, Y2, M, C, ystride, partial); break;
    case CS_ROWSCALE: hipLaunchKernelGGL(colsum_kernel<CS_ROWSCALE>, dim3(nb), dim3(256), 0, st, X, Y, Y2, M, C, ystride, partial); break;
    default: hipLaunchKernelGGL(colsum_kernel<CS_HEADSCALE>, dim3(nb), dim3(256), 0, st, X, Y, Y2, M, C, ystride, partial); break;
    }
    return hipGetLastError();
}

hipError_t launch_colsum_store(const double *partial, int nblocks, int C, int ostride, float *out0, float *out1,
                               hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(colsum_store_kernel, dim3(1), dim3(1024), 0, st, partial, nblocks, C, ostride, out0, out1);
    return hipGetLastError();
}

hipError_t launch_sum_vector(const float *v, long M, float *out, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(sum_vector_kernel, dim3(1), dim3(1024), 0, st, v, M, out);
    return hipGetLastError();
}

hipError_t launch_bn_stats_finalize(const double *partial, int nblocks, long M, const float *gamma, const float *beta,
                                    float eps, float *scale, float *shift, float *mean, float *invstd, float *batch_mean,
                                    float *batch_var, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(1), dim3(1024), 0, st, partial, nblocks, M, gamma, beta, eps, scale, shift,
                       mean, invstd, batch_mean, batch_var);
    return hipGetLastError();
}

hipError_t launch_bn_bwd_finalize(const double *partial, int nblocks, long M, const float *gamma, const float *mean,
                                  const float *invstd, float *dgamma, float *dbeta, float *coef, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(1024), 0, st, partial, nblocks, M, gamma, mean, invstd, dgamma,
                       dbeta, coef);
    return hipGetLastError();
}

static int ew_grid(long M) { return grid_cap((M * 32 + 255) / 256, 256 * 16); }

hipError_t launch_bn_bwd_apply(const float *dy, const float *x, const float *mean, const float *coef, float *dx, long M,
                               hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(M)), dim3(256), 0, st, dy, x, mean, coef, dx, M);
    return hipGetLastError();
}

hipError_t launch_bn_bwd_apply_and_affine(const float *dy, const float *h3, const float *mean, const float *coef, float *dx,
                                          const float *h1, const float *scale, const float *shift, float *xout, long M,
                                          hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(bn_bwd_apply_and_affine_kernel, dim3(ew_grid(M)), dim3(256), 0, st, dy, h3, mean, coef, dx, h1, scale,
                       shift, xout, M);
    return hipGetLastError();
}

hipError_t launch_affine_cols(const float *x, const float *scale, const float *shift, float *out, long M, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(affine_cols_kernel, dim3(ew_grid(M)), dim3(256), 0, st, x, scale, shift, out, M);
    return hipGetLastError();
}

hipError_t launch_outer_rows(const float *dy, const float *w, float *dh, long M, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(outer_rows_kernel, dim3(ew_grid(M)), dim3(256), 0, st, dy, w, dh, M);
    return hipGetLastError();
}

hipError_t launch_gat_combine_train(const float *part, const float *part_ms, const float *h, long M, float *g, float *h1,
                                    float *att, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(gat_combine_train_kernel, dim3(ew_grid(M)), dim3(256), 0, st, part, part_ms, h, M, g, h1, att);
    return hipGetLastError();
}

int gat_bwd_max_nodes() { return 16 * kGatBwdMaxTiles + 1; }   // register-resident accumulators: one per 16-node source tile

size_t gat_bwd_lds_bytes(int n) {
    const size_t ns = (size_t)n - 1, nt = (ns + 15) / 16, nsp = nt * 16;
    (void)nt; (void)nsp;
    return 2 * ns * LDG * 4 + ns * kGatBwdHeads * 4 + ns * kGatBwdHeads * 16 + ns * 4 + 16;
}

hipError_t launch_gat_bwd_rows(const float *ft, const float *dout, const float *gout, const float *att, const float *attn_l,
                               const float *attn_r, int B, int n, float *P, float *dlr, hipStream_t st) {
    const size_t lds = gat_bwd_lds_bytes(n);
    const int nt = (n - 1 + 15) / 16;
    auto kern = nt <= 9 ? gat_bwd_rows_kernel<9> : nt <= 13 ? gat_bwd_rows_kernel<13> : gat_bwd_rows_kernel<16>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3((unsigned)(B * n * (kH / kGatBwdHeads))), dim3(kGatBwdThreads), lds, st, ft, dout, gout, att, attn_l,
                       attn_r, n, P, dlr);
    return hipGetLastError();
}

hipError_t launch_gat_bwd_combine(const float *P, const float *dlr, const float *attn_l, const float *attn_r, long M,
                                  float *dft, float *dl, float *dr, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(gat_bwd_combine_kernel, dim3(ew_grid(M)), dim3(256), 0, st, P, dlr, attn_l, attn_r, M, dft, dl, dr);
    return hipGetLastError();
}

int gemm_tn_chunks(long M) {           // >= 128 rows (4 k-tiles) per chunk, at most kGemmTnMaxChunks chunks
    long c = (M + 127) / 128;
    return grid_cap(c, kGemmTnMaxChunks);
}

hipError_t launch_gemm_tn(const float *X, const float *Y, long M, int N1, int N2, float *partial, float *out,
                          float *xsum_out, hipStream_t st) {
    const int chunks = gemm_tn_chunks(M);
    long rows = (M + chunks - 1) / chunks;
    rows = (rows + TBK - 1) / TBK * TBK;
    const long E = (long)N1 * N2;
    float *xpart = xsum_out ? partial + (size_t)chunks * E : nullptr;      // [chunks][N1] behind the tile partials
    (void)hipGetLastError();
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(chunks, N1 / 128, N2 / 128), dim3(256), 0, st, X, Y, M, N1, N2, rows, partial, xpart);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const unsigned blocks = (unsigned)((E + 63) / 64 + (xsum_out ? (N1 + 63) / 64 : 0));
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(blocks), dim3(256), 0, st, partial, chunks, E, out, xpart, (long)N1, xsum_out);
    return hipGetLastError();
}

}  // namespace gnngls
