// model_kernels.hip -- edge-regret GNN forward on MI355X (gfx950), hand-written HIP, fp32 results (the feed-forward block of the
// inference forward computes them on the bf16 matrix pipe from three bf16 pieces per fp32 operand: ffn_fused_bf16x3_kernel).
//
// Replaces (reference file:line, /root/reference/gnngls/...):
//   models.py:44-70   EdgePropertyPredictionModel.forward
//   models.py:18-41   AttentionLayer  (h + GATConv ; BN ; h + MLP ; BN)
//   models.py:23      dgl.nn.GATConv(128, 16, 8) on the line graph of K_n   (third party, DGL 0.6.1)
//   datasets.py:73-95 get_scaled_features (MinMax transform, line-graph node order)
//   test.py:79-83     inverse_transform + clamp at 0 -> 'regret_pred' edge attribute
//
// MI355X-first formulation: the line graph of K_n needs no graph structure.  GNN node
// d = TSP edge {i,j}; its in-neighbours are {i,k} and {k,j}, k not in {i,j}.  Activations are
// stored packed, h[B, N=n(n-1)/2, 128], node id = rank of (i<j) in itertools.combinations order.
//   K2  gemm_f32_kernel   : node-wise linears on the f32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32
//                           fma chain), 128x128x32 tiles, fused bias/ReLU/skip/BatchNorm epilogues
//   K1  gat_rows_kernel   : one workgroup owns TSP row i of one instance: ft[{i,*}] is staged once in
//                           LDS (n=100: 50.7 KB), every destination {i,j} streams over that tile;
//                           el/er are recomputed from the LDS tile (no [N,8] tensors in HBM); the
//                           softmax shift is the exact row maximum (top-2 trick excludes k=j)
//       (the log-sum-exp merge of the two row partials + skip + BN1 is fused into ffn_fused_kernel)
//   K2b ffn_fused_bf16x3_kernel (inference) / ffn_fused_kernel (training, fp32 pipe): merge + BN1 + Linear/ReLU/Linear + skip + BN2
//                           in one launch, the hidden layer never leaves the registers; the inference form also carries the next layer's fc
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>

#include "model_kernels.h"

namespace gnngls {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// packed fp32 (two lanes of work per instruction); the compiler scalarises <2 x float> arithmetic next to scalar selects
#ifndef GAT_PACKED
#define GAT_PACKED 0            // 1: v_pk_mul_f32 / v_pk_add_f32 on head pairs (round 2: +2 % with the weight arithmetic of that round; round 5, with the maximum form: 5.07 -> 4.95 ms WITHOUT packing)
#endif
#if GAT_PACKED
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
#else
// (A/B: two single instructions per pair -- MI355X_MICROARCH.md prices a packed f32 instruction beside MFMAs above two plain ones)
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
    float d0, d1;
    asm("v_mul_f32 %0, %1, %2" : "=v"(d0) : "v"(a[0]), "v"(b[0]));
    asm("v_mul_f32 %0, %1, %2" : "=v"(d1) : "v"(a[1]), "v"(b[1]));
    return f32x2{d0, d1};
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    float d0, d1;
    asm("v_add_f32 %0, %1, %2" : "=v"(d0) : "v"(a[0]), "v"(b[0]));
    asm("v_add_f32 %0, %1, %2" : "=v"(d1) : "v"(a[1]), "v"(b[1]));
    return f32x2{d0, d1};
}
#endif
// one v_max_f32: fmaxf() on values that come out of inline asm gets a canonicalising v_max_f32 x, x, x per operand first
// (three instructions per maximum)
__device__ __forceinline__ float max_f32(float a, float b) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }

constexpr int kD = 128;        // embed_dim
constexpr int kEmbedFcMaxIn = 32;     // input features the fused embed + first fc is prepared for (embed_fc_kernel)
constexpr int kH = 8;          // heads
constexpr int kF = 16;         // head dim
constexpr float kSlope = 0.2f; // GATConv negative_slope default

// ---------------------------------------------------------------------------------------------
// pack / unpack
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int pair_index(int i, int j, int n) {   // i < j
    return i * n - ((i * (i + 1)) >> 1) + (j - i - 1);
}

// datasets.py:84-89: features = float32(weight); sklearn MinMaxScaler.transform on a float32 array:
// X *= scale_ (fp64 math, rounded to fp32), X += min_ (fp64 math, rounded to fp32).
__global__ void pack_features_kernel(const double *D, int B, int n, double scale, double minv, float *feat) {
    const int N = n * (n - 1) / 2;
    const long total = (long)B * N;
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        int b = (int)(q / N), e = (int)(q % N);
        // invert the pair index: row i such that start(i) <= e < start(i+1)
        int i = 0, rem = e;
        while (rem >= n - 1 - i) { rem -= n - 1 - i; ++i; }
        int j = i + 1 + rem;
        float w = (float)D[((size_t)b * n + i) * n + j];
        float x = (float)((double)w * scale);
        x = (float)((double)x + minv);
        feat[q] = x;
    }
}

// test.py:79-83: inverse_transform ((y - min_) / scale_ in fp64, rounded to fp32 twice), .item(),
// np.maximum(., 0) -> float64 edge attribute.  Written as a symmetric [B,n,n] fp64 guide matrix.
__global__ void unpack_regret_kernel(const float *y, int B, int n, double scale, double minv, double *out) {
    const long total = (long)B * n * n;
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        int b = (int)(q / ((long)n * n));
        int r = (int)(q % ((long)n * n));
        int i = r / n, j = r % n;
        double v = 0.0;
        if (i != j) {
            int lo = i < j ? i : j, hi = i < j ? j : i;
            float p = y[(size_t)b * (n * (n - 1) / 2) + pair_index(lo, hi, n)];
            p = (float)((double)p - minv);
            p = (float)((double)p / scale);
            v = (double)p;
            if (!(v > 0.0)) v = 0.0;      // np.maximum(x, 0)
        }
        out[q] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// embed (models.py:57,66): h[m,c] = sum_d x[m,d] W[c,d] + b[c]
// ---------------------------------------------------------------------------------------------
__global__ void embed_kernel(const float *x, const float *W, const float *bias, float *h, long M, int in_dim) {
    const long total = M * (kD / 4);
    const long stride = (long)gridDim.x * blockDim.x;          // a multiple of 32: the 4 columns of a thread never change
    const long q0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const int c = (int)(q0 % (kD / 4)) * 4;
    if (in_dim == 1) {                                          // the reference's feature set (datasets.py:14-20)
        const f32x4 w = f32x4{W[c], W[c + 1], W[c + 2], W[c + 3]};
        const f32x4 bb = *reinterpret_cast<const f32x4 *>(bias + c);
        for (long q = q0; q < total; q += stride) {
            const float xv = x[q / (kD / 4)];
            f32x4 acc;
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = fmaf(xv, w[u], 0.f) + bb[u];
            *reinterpret_cast<f32x4 *>(h + q * 4) = acc;
        }
        return;
    }
    for (long q = q0; q < total; q += stride) {
        const long m = q / (kD / 4);
        f32x4 acc;
        for (int u = 0; u < 4; ++u) {
            float a = 0.f;
            for (int d = 0; d < in_dim; ++d) a = fmaf(x[m * in_dim + d], W[(c + u) * in_dim + d], a);
            acc[u] = a + bias[c + u];
        }
        *reinterpret_cast<f32x4 *>(h + m * kD + c) = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// embed + the first layer's fc in one pass (models.py:66 then models.py:23 of layer 0): both are linear maps of the in_dim input
// features, so ft = fc(embed(x)) = x A + b' with A = We^T Wfc^T [in_dim, 128], b' = Wfc be -- a rank-in_dim update per node instead
// of a [M,128] x [128,128] GEMM behind a 2.6 GB round trip of h through HBM (1,024 x TSP100: embed 0.52 + gemm 2.18 ms -> one
// kernel bound by its two stores).  A and b' are made once per weight image (embed_fc_prepare_kernel, fp64 sums rounded once:
// closer to the exact product than the reference's two fp32 stages); h keeps embed_kernel's arithmetic.
// ---------------------------------------------------------------------------------------------
__global__ void embed_fc_prepare_kernel(const float *__restrict__ We, const float *__restrict__ be, const float *__restrict__ Wfc,
                                        const float *__restrict__ attn_l, const float *__restrict__ attn_r,
                                        int in_dim, float *__restrict__ A, float *__restrict__ bp) {
    const int c = threadIdx.x;                                  // 128 threads: output column c of the fc
    for (int d = 0; d < in_dim; ++d) {
        double a = 0.0;
        for (int k = 0; k < kD; ++k) a += (double)Wfc[c * kD + k] * (double)We[k * in_dim + d];
        A[d * kD + c] = (float)a;
    }
    double b = 0.0;
    for (int k = 0; k < kD; ++k) b += (double)Wfc[c * kD + k] * (double)be[k];
    bp[c] = (float)b;
    // in_dim == 1: the first layer's attention logits are affine in the node's one feature, el = x al[h] + bl[h], er likewise
    // (GATConv: (ft * attn).sum(-1) with ft = x A + b'): al | bl | ar | br, 8 heads each, behind b' (gat_rows_rank1_kernel)
    if (in_dim == 1 && c < 4 * kH) {
        const int what = c / kH, h = c % kH;
        const float *att = (what < 2) ? attn_l : attn_r;
        double v = 0.0;
        for (int j = 0; j < kF; ++j) {
            const int col = h * kF + j;
            double t = 0.0;
            for (int k = 0; k < kD; ++k) t += (double)Wfc[col * kD + k] * ((what & 1) ? (double)be[k] : (double)We[k]);
            v += t * (double)att[col];
        }
        bp[kD + c] = (float)v;
    }
}

__global__ void embed_fc_kernel(const float *__restrict__ x, const float *__restrict__ W, const float *__restrict__ bias,
                                const float *__restrict__ A, const float *__restrict__ bp, float *__restrict__ h,
                                float *__restrict__ ft, long M, int in_dim) {
    const long total = M * (kD / 4);
    const long stride = (long)gridDim.x * blockDim.x;          // a multiple of 32: the 4 columns of a thread never change
    const long q0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const int c = (int)(q0 % (kD / 4)) * 4;
    const f32x4 bb = *reinterpret_cast<const f32x4 *>(bias + c), b2 = *reinterpret_cast<const f32x4 *>(bp + c);
    if (in_dim == 1) {                                          // the reference's feature set (datasets.py:14-20)
        const f32x4 w = f32x4{W[c], W[c + 1], W[c + 2], W[c + 3]};
        const f32x4 a = *reinterpret_cast<const f32x4 *>(A + c);
        for (long q = q0; q < total; q += stride) {
            const float xv = x[q / (kD / 4)];
            f32x4 hv, fv;
#pragma unroll
            for (int u = 0; u < 4; ++u) { hv[u] = fmaf(xv, w[u], 0.f) + bb[u]; fv[u] = fmaf(xv, a[u], b2[u]); }
            *reinterpret_cast<f32x4 *>(h + q * 4) = hv;
            if (ft) *reinterpret_cast<f32x4 *>(ft + q * 4) = fv;      // (nullptr: the first GATConv runs in its rank-1 form and needs no ft)
        }
        return;
    }
    for (long q = q0; q < total; q += stride) {
        const long m = q / (kD / 4);
        f32x4 hv, fv = b2;
        for (int u = 0; u < 4; ++u) {
            float s = 0.f;
            for (int d = 0; d < in_dim; ++d) s = fmaf(x[m * in_dim + d], W[(c + u) * in_dim + d], s);
            hv[u] = s + bias[c + u];
        }
        for (int d = 0; d < in_dim; ++d) {
            const float xv = x[m * in_dim + d];
            const f32x4 a = *reinterpret_cast<const f32x4 *>(A + d * kD + c);
#pragma unroll
            for (int u = 0; u < 4; ++u) fv[u] = fmaf(xv, a[u], fv[u]);
        }
        *reinterpret_cast<f32x4 *>(h + m * kD + c) = hv;
        *reinterpret_cast<f32x4 *>(ft + m * kD + c) = fv;
    }
}

// ---------------------------------------------------------------------------------------------
// K2: C[M,N] = A[M,K] * W[N,K]^T (+ epilogue), fp32 MFMA 32x32x2.
//   block 256 threads = 4 waves, tile 128x128, wave tile 64x64 (2x2 MFMA tiles), BK = 32
//   LDS images are k-major ([k][row], row stride 129 floats): fragment reads are conflict-free.
// ---------------------------------------------------------------------------------------------
// training-only epilogues (train_kernels.hip): EPI_MASK  C = acc * (skip[row,col] > 0)   (ReLU backward; C may alias skip)
//                                               EPI_ADD   C = acc + skip[row,col]         (skip-connection backward)
// WKN: the second operand is given as W[K,N] row-major (C = A * W) instead of W[N,K] (C = A * W^T) -- the data-gradient
// GEMMs of the backward pass read the forward weights in place, no transposed copies.
enum { EPI_STORE = 0, EPI_BIAS_RELU = 1, EPI_BIAS_SKIP_BN = 2, EPI_MASK = 3, EPI_ADD = 4 };

constexpr int BM = 128, BN = 128, BK = 32, LDT = BM + 1;

template <int EPI, bool WKN = false>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float *__restrict__ A, const float *__restrict__ W,
                                                       float *C, long M, int N, int K,
                                                       const float *__restrict__ bias,
                                                       const float *skip,
                                                       const float *__restrict__ bn_scale,
                                                       const float *__restrict__ bn_shift) {
    __shared__ float As[BK * LDT];
    __shared__ float Ws[BK * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row0 = (long)blockIdx.x * BM;
    const int col0 = blockIdx.y * BN;
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // staging map: thread -> (row = tid/2, 16 consecutive k at (tid&1)*16)
    const int srow = tid >> 1, sk = (tid & 1) * 16;
    const long arow = row0 + srow;
    const bool arow_ok = arow < M;
    const float *aptr = A + (arow_ok ? arow : 0) * (long)K + sk;
    const float *wptr = W + (long)(col0 + srow) * K + sk;
    // W[K,N] staging map: slot u of the thread covers k = (u*256 + tid) / 32, 4 consecutive columns at ((u*256 + tid) % 32) * 4
    const int tk = tid >> 5, tc = (tid & 31) * 4;

    f32x4 ra[4], rw[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ra[u] = arow_ok ? *reinterpret_cast<const f32x4 *>(aptr + k0 + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
            rw[u] = WKN ? *reinterpret_cast<const f32x4 *>(W + (long)(k0 + 8 * u + tk) * N + col0 + tc)
                        : *reinterpret_cast<const f32x4 *>(wptr + k0 + 4 * u);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                As[(sk + 4 * u + c) * LDT + srow] = ra[u][c];
                if (WKN) Ws[(8 * u + tk) * LDT + tc + c] = rw[u][c];
                else Ws[(sk + 4 * u + c) * LDT + srow] = rw[u][c];
            }
    };

    gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();          // previous tile fully consumed
        lstore();
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);   // prefetch next tile into registers under the MFMAs
        const int lr = lane & 31, lk = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float *ap = As + (kk + lk) * LDT;
            const float *wp = Ws + (kk + lk) * LDT;
            float a0 = ap[wr + lr], a1 = ap[wr + 32 + lr];
            float b0 = wp[wc + lr], b1 = wp[wc + 32 + lr];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int lc = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const int col = col0 + wc + tb * 32 + lc;
        float bv = 0.f, sc = 1.f, sh = 0.f;
        if (EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_SKIP_BN) bv = bias[col];
        if (EPI == EPI_BIAS_SKIP_BN) { sc = bn_scale[col]; sh = bn_shift[col]; }
#pragma unroll
        for (int ta = 0; ta < 2; ++ta) {
            // (C may alias `skip` for EPI_MASK, so neither is __restrict__: fetch the 16 auxiliary values of the tile
            // before its first store, otherwise every load waits for the previous store)
            float auxv[16];
            if (EPI == EPI_MASK || EPI == EPI_ADD) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long row = row0 + wr + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    auxv[r] = row < M ? skip[row * (long)N + col] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = row0 + wr + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row >= M) continue;
                float v = acc[ta][tb][r];
                if (EPI == EPI_BIAS_RELU) { v = v + bv; v = v > 0.f ? v : 0.f; }
                if (EPI == EPI_BIAS_SKIP_BN) {
                    v = v + bv;                               // Linear2 output (models.py:31)
                    v = skip[row * (long)N + col] + v;        // x + y        (models.py:15)
                    v = v * sc + sh;                          // BatchNorm1d eval (models.py:35)
                }
                if (EPI == EPI_MASK) v = auxv[r] > 0.f ? v : 0.f;
                if (EPI == EPI_ADD) v = v + auxv[r];
                C[row * (long)N + col] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K1: GAT attention + aggregation over the line graph of K_n, one workgroup per (instance, row i).
//   sources of row i: nodes {i,k}, k != i  (slot k' = k < i ? k : k-1, n-1 slots)
//   for every destination {i,j} (same slots) and head h:
//       score_k = LeakyReLU(el[{i,k},h] + er[{i,j},h]),  k not in {i,j}
//       m = max_k score_k ; s = sum_k exp(score_k - m) ; P[:] = sum_k exp(score_k - m) * ft[{i,k},h,:]
//   written to side (i<j ? 0 : 1) of the partial buffers; the other endpoint's row supplies the
//   second half of the 2(n-2) in-neighbours.
// ---------------------------------------------------------------------------------------------
// HS = heads per workgroup.  HS = 8: one workgroup stages the whole ft rows of TSP row i (n = 100: 57 KB of LDS, two
// workgroups per CU).  HS = 4: the heads are split over two workgroups (columns [64 hg, 64 hg + 64) of the same rows):
// the same arithmetic per head; used for n > 140 where the unsplit tile (n = 200: 115 KB) leaves one workgroup per CU
// and nothing overlaps its prologue (staging, logits, top-2, factor tables: ~40 % of a workgroup's cycles).
// LDS row stride of the staged ft tile (floats) = 16 HS + 16 = 16 (mod 64): the 4 source rows an MFMA B-fragment read
// touches land on 4 disjoint groups of 16 banks.
template <int HS>
__global__ __launch_bounds__(512) void gat_rows_kernel(const float *__restrict__ ft, const float *__restrict__ attn_l,
                                                       const float *__restrict__ attn_r, int n,
                                                       float *__restrict__ part, float *__restrict__ part_ms) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int LDF = HS * kF + 16;
    constexpr int HG = kH / HS;                                  // workgroups per (instance, row)
    const int N = n * (n - 1) / 2;
    const int ns = n - 1;
    const int hb = (blockIdx.x % HG) * HS;                       // first head of this workgroup
    const int b = blockIdx.x / (n * HG), i = (blockIdx.x / HG) % n;
#ifndef GAT_DBG
#define GAT_DBG 0                     // bit 0: no stores of the partials (time attribution builds)
#endif
#ifndef GAT_UNIFORM_WAVE
#define GAT_UNIFORM_WAVE 1
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    // the wavefront index as a SCALAR: the unit loop, the four source runs of a unit and their addresses' uniform parts then run on the
    // scalar ALU with s_cbranch_scc loop ends instead of exec-masked loop control
    const int wave = GAT_UNIFORM_WAVE ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;      // 4..8 waves, chosen by the launcher to balance the units
    float *ftS = reinterpret_cast<float *>(smem);            // [ns][LDF]
    float *elS = ftS + (size_t)ns * LDF;                     // [ns][HS]
    float *erS = elS + (size_t)ns * HS;                      // [ns][HS]
    float *eaS = erS + (size_t)ns * HS;                      // [ns][HS] exp(el - max1)            } factorised softmax weights,
    float *ebS = eaS + (size_t)ns * HS;                      // [ns][HS] exp(0.2 (el - max1))      } see the aggregation loop
    float *top = ebS + (size_t)ns * HS;                      // [HS][4]: max1, max2, argmax1 (int bits), direct-path flag
    int *nodeS = reinterpret_cast<int *>(top + HS * 4);      // [ns] global node id of slot

    const float *ftb = ft + (size_t)b * N * kD;
    for (int s = tid; s < ns; s += nthreads) {
        int k = s < i ? s : s + 1;
        nodeS[s] = k < i ? pair_index(k, i, n) : pair_index(i, k, n);
    }
    __syncthreads();
    // stage the workgroup's head columns of the ft rows: 4 HS x 16 B per node, coalesced; the loads of a batch are all
    // issued before the first LDS store (a thread moves ~7 float4 at n = 100: one global round trip instead of seven)
    constexpr int SB = 8;
    constexpr int V4 = HS * kF / 4;                              // float4 per staged row
    for (int q0 = tid; q0 < ns * V4; q0 += nthreads * SB) {
        f32x4 v[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int q = q0 + u * nthreads;
            if (q < ns * V4) v[u] = *reinterpret_cast<const f32x4 *>(ftb + (size_t)nodeS[q / V4] * kD + hb * kF + (q % V4) * 4);
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int q = q0 + u * nthreads;
            if (q < ns * V4) *reinterpret_cast<f32x4 *>(ftS + (size_t)(q / V4) * LDF + (q % V4) * 4) = v[u];
        }
    }
    __syncthreads();
    {   // el / er (GATConv: (feat * attn).sum(-1)); the head of a thread is fixed (the stride is a multiple of HS): its
        // attention vectors live in registers, the ft slice comes in as four ds_read_b128
        const int h = tid % HS;
        f32x4 al[4], ar[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            al[v] = *reinterpret_cast<const f32x4 *>(attn_l + (hb + h) * kF + 4 * v);
            ar[v] = *reinterpret_cast<const f32x4 *>(attn_r + (hb + h) * kF + 4 * v);
        }
        for (int q = tid; q < ns * HS; q += nthreads) {
            const float *f = ftS + (size_t)(q / HS) * LDF + h * kF;
            float l = 0.f, r = 0.f;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const f32x4 fv = *reinterpret_cast<const f32x4 *>(f + 4 * v);
#pragma unroll
                for (int u = 0; u < 4; ++u) { l = fmaf(fv[u], al[v][u], l); r = fmaf(fv[u], ar[v][u], r); }
            }
            elS[q] = l; erS[q] = r;
        }
    }
    __syncthreads();
    if (tid < 32 * HS) {   // top-2 of el per head over the row's sources: 32 lanes per head, merge (max1, arg1, max2) by shuffles
        const int h = tid >> 5, l32 = tid & 31;
        float m1 = -INFINITY, m2 = -INFINITY; int a1 = -1;
        for (int s = l32; s < ns; s += 32) {
            float v = elS[s * HS + h];
            if (v > m1) { m2 = m1; m1 = v; a1 = s; } else if (v > m2) { m2 = v; }
        }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) {
            float om1 = __shfl_xor(m1, o, 32), om2 = __shfl_xor(m2, o, 32);
            int oa1 = __shfl_xor(a1, o, 32);
            // merge two (max1, arg1, max2) triples; on equal maxima keep the lower source index
            bool take = om1 > m1 || (om1 == m1 && oa1 >= 0 && (a1 < 0 || oa1 < a1));
            float lo1 = take ? m1 : om1;                 // the loser's best is a candidate for second place
            float hi2 = take ? om2 : m2;
            if (take) { m1 = om1; a1 = oa1; }
            m2 = lo1 > hi2 ? lo1 : hi2;
        }
        // the factorisation below scales by exp(max1 - max2) for the destination that is itself the arg-max source:
        // keep the direct evaluation when that could overflow
        if (l32 == 0) { top[h * 4 + 0] = m1; top[h * 4 + 1] = m2; top[h * 4 + 2] = __int_as_float(a1);
                        top[h * 4 + 3] = (m1 - m2 > 60.f) ? 1.f : 0.f; }
    }
    __syncthreads();
    for (int q = tid; q < ns * HS; q += nthreads) {
        const float d = (elS[q] - top[(q % HS) * 4]) * 1.4426950408889634f;
        eaS[q] = __builtin_amdgcn_exp2f(d);
        ebS[q] = __builtin_amdgcn_exp2f(kSlope * d);
    }
    __syncthreads();

    // ---- weighted aggregation on v_mfma_f32_16x16x4_f32 -------------------------------------------
    // one unit = (tile of 16 destinations, pair of heads); D[16 dest x 16 feat] += A[16 dest x 4 src] * B[4 src x 16 feat]
    //   A[i = lane&15][k = lane>>4] = exp(LeakyReLU(el[src] + er[dest]) - m[dest])   (computed in the lane)
    //   B[k = lane>>4][j = lane&15] = ft[src][head*16 + j]                            (LDS tile)
    const float kLog2e = 1.4426950408889634f;
    const int n_dt = (ns + 15) >> 4;
    const int jl = lane & 15, kq = lane >> 4;
    float *pb = part + (size_t)b * N * kD;
    float *mb = part_ms + (size_t)b * N * (2 * kH);
    const size_t side_stride = (size_t)gridDim.x / (n * HG) * N;     // B*N nodes per side
    constexpr int HU = 4;                                      // heads per unit (independent MFMA chains)
    for (int unit = wave; unit < n_dt * (HS / HU); unit += nwaves) {
        const int dt = unit / (HS / HU), h0 = (unit % (HS / HU)) * HU;      // h0: head index inside the workgroup's HS heads
        const int js = dt * 16 + jl;
        const int jsc = js < ns ? js : ns - 1;
        float er[HU], mm[HU], nm[HU], ws[HU], cpos[HU], cneg[HU], ner[HU];
        f32x4 acc[HU];
        bool direct = false;
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            const int h = h0 + u;
            er[u] = erS[jsc * HS + h];
            float m = ((__float_as_int(top[h * 4 + 2]) == js) ? top[h * 4 + 1] : top[h * 4 + 0]) + er[u];
            mm[u] = m > 0.f ? m : kSlope * m;      // LeakyReLU is monotone: max score = score of max el
            nm[u] = -mm[u] * kLog2e;
            ws[u] = 0.f;
            acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            // exp(LeakyReLU(el + er) - mm) = exp(el - M) * exp(er + M - mm)              if el + er > 0
            //                              = exp(0.2 (el - M)) * exp(0.2 (er + M) - mm)  otherwise      (M = max1 of the head)
            // the source factors are in eaS / ebS, the destination factors are computed here: no v_exp_f32 (quarter rate)
            // in the inner loop
            const float t = er[u] + top[h * 4 + 0];
            cpos[u] = __builtin_amdgcn_exp2f(fminf(t - mm[u], 80.f) * kLog2e);
            cneg[u] = __builtin_amdgcn_exp2f(fminf(kSlope * t - mm[u], 80.f) * kLog2e);
            ner[u] = -er[u];
            direct = direct || top[h * 4 + 3] != 0.f;          // wave-uniform
        }
        // two source groups (2 x 4 sources) per iteration: 8 independent weights + 8 MFMAs in flight.  The loop is bound by
        // the vector ALU beside the matrix pipe (one weight feeds only 16 features), so the weight arithmetic is packed:
        // both products and the denominator sums run two heads per instruction (v_pk_mul_f32 / v_pk_add_f32), and the
        // self-loop / padding mask is only applied in the iterations that can meet either (the destination tile's own
        // sources and the last, partial group) -- same values, same order of the sums.
        if (!direct) {
            static_assert(HU == 4, "two packed head pairs per unit");
            f32x2 cpos2[2] = {f32x2{cpos[0], cpos[1]}, f32x2{cpos[2], cpos[3]}};
            f32x2 cneg2[2] = {f32x2{cneg[0], cneg[1]}, f32x2{cneg[2], cneg[3]}};
            f32x2 ws2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
            auto body = [&](int s0, auto masked) {
                constexpr bool MASK = decltype(masked)::value;
                int sidx[2]; bool live[2];
                f32x4 ea[2], eb[2];
                float bv[2][HU];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int s = s0 + 4 * g + kq;
                    sidx[g] = MASK ? (s < ns ? s : ns - 1) : s;
                    live[g] = (s < ns) && (s != js);              // no self loop, no padding
                    ea[g] = *reinterpret_cast<const f32x4 *>(eaS + sidx[g] * HS + h0);
                    eb[g] = *reinterpret_cast<const f32x4 *>(ebS + sidx[g] * HS + h0);
#pragma unroll
                    for (int u = 0; u < HU; ++u) bv[g][u] = ftS[(size_t)sidx[g] * LDF + (h0 + u) * kF + jl];
                }
#pragma unroll
                for (int g = 0; g < 2; ++g) {
#pragma unroll
                    for (int hp = 0; hp < 2; ++hp) {
                        const f32x2 ea2 = hp == 0 ? f32x2{ea[g][0], ea[g][1]} : f32x2{ea[g][2], ea[g][3]};
                        const f32x2 eb2 = hp == 0 ? f32x2{eb[g][0], eb[g][1]} : f32x2{eb[g][2], eb[g][3]};
                        const f32x2 wp = pk_mul(ea2, cpos2[hp]), wn = pk_mul(eb2, cneg2[hp]);
                        f32x2 w;
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            // exp is monotone and LeakyReLU(x) = max(x, 0.2 x): exp(LeakyReLU(x) - m) = max(exp(x - m),
                            // exp(0.2 x - m)) -- one v_max_f32 instead of a compare against the logits and a select, and the
                            // el table is not read in this loop at all (where the two products are within rounding of each
                            // other, x ~ 0, either is the weight to ~1 ulp)
                            float x = max_f32(wp[c], wn[c]);
                            if (MASK) x = live[g] ? x : 0.f;
                            w[c] = x;
                        }
                        ws2[hp] = pk_add(ws2[hp], w);
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            acc[2 * hp + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c], bv[g][2 * hp + c], acc[2 * hp + c], 0, 0, 0);
                    }
                }
            };
            // sources in ascending order, as four runs (one loop per body: a per-iteration choice between the two bodies
            // costs a copy of the 16 accumulator registers every iteration): below the tile's own destinations, the two
            // groups that contain them, above them, and the last partial group
            const int t0 = dt * 16;
            int s0 = 0;
            for (; s0 < t0; s0 += 8) body(s0, std::false_type{});                      // t0 <= ns - 1: full groups
            for (; s0 < t0 + 16 && s0 < ns; s0 += 8) body(s0, std::true_type{});
            for (; s0 + 8 <= ns; s0 += 8) body(s0, std::false_type{});
            for (; s0 < ns; s0 += 8) body(s0, std::true_type{});
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) { ws[2 * hp] = ws2[hp][0]; ws[2 * hp + 1] = ws2[hp][1]; }
        } else {
            for (int s0 = 0; s0 < ns; s0 += 8) {
                int sidx[2]; bool live[2];
                f32x4 e[2];
                float bv[2][HU];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int s = s0 + 4 * g + kq;
                    sidx[g] = s < ns ? s : ns - 1;
                    live[g] = (s < ns) && (s != js);
                    e[g] = *reinterpret_cast<const f32x4 *>(elS + sidx[g] * HS + h0);
#pragma unroll
                    for (int u = 0; u < HU; ++u) bv[g][u] = ftS[(size_t)sidx[g] * LDF + (h0 + u) * kF + jl];
                }
#pragma unroll
                for (int g = 0; g < 2; ++g) {
#pragma unroll
                    for (int u = 0; u < HU; ++u) {
                        float x = e[g][u] + er[u];
                        x = fmaxf(x, kSlope * x);                                  // LeakyReLU(x) = max(x, 0.2x)
                        float w = __builtin_amdgcn_exp2f(fmaf(x, kLog2e, nm[u]));
                        w = live[g] ? w : 0.f;
                        ws[u] += w;
                        acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, bv[g][u], acc[u], 0, 0, 0);
                    }
                }
            }
        }
        // softmax denominators: sum the 4 source lanes (lane>>4) of every destination
#pragma unroll
        for (int u = 0; u < HU; ++u) { ws[u] += __shfl_xor(ws[u], 16, 64); ws[u] += __shfl_xor(ws[u], 32, 64); }
        // C/D layout 16x16: col = lane&15 (feature), row = (lane>>4)*4 + reg (destination in the tile)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int jd = dt * 16 + kq * 4 + r;
            if (jd < ns) {
                const int j = jd < i ? jd : jd + 1;
                const size_t node = (size_t)nodeS[jd];
                float *po = pb + (i < j ? 0 : side_stride * kD) + node * kD;
#pragma unroll
                for (int u = 0; u < HU; ++u) if (!(GAT_DBG & 1) || acc[u][r] == 12345.678f) po[(hb + h0 + u) * kF + jl] = acc[u][r];
            }
        }
        if (kq == 0 && js < ns) {
            const int j = js < i ? js : js + 1;
            float *mo = mb + (i < j ? 0 : side_stride * (2 * kH)) + (size_t)nodeS[js] * (2 * kH);
#pragma unroll
            for (int u = 0; u < HU; ++u) { mo[hb + h0 + u] = mm[u]; mo[kH + hb + h0 + u] = ws[u]; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K1': the FIRST GATConv when the model has ONE input feature (the reference's default feature set, datasets.py:14-20).  Embedding and
// fc are linear, so ft[s] = x_s A + b' (embed_fc_kernel), the logits are affine in x_s (el = x al[h] + bl[h], er likewise) and the
// aggregation sum_s w[d,s,h] ft[s, 16 h + j] = (sum_s w x_s) A[16 h + j] + (sum_s w) b'[16 h + j]: TWO weighted sums per
// (destination, head) instead of sixteen -- the weights, shifts and exclusions are gat_rows_kernel's, term for term, but nothing is
// left for the matrix pipe to do and no ft tile is staged.  One workgroup per (instance, TSP row i); a wavefront owns 64 destinations
// x 4 heads, lane = destination, and walks the n - 1 sources with uniform LDS reads.  Writes gat_rows_kernel's partials.
// ---------------------------------------------------------------------------------------------
template <bool COMPACT>
__global__ __launch_bounds__(512) void gat_rows_rank1_kernel(const float *__restrict__ x, const float *__restrict__ img, int n,
                                                             float *__restrict__ part, float *__restrict__ part_ms) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int N = n * (n - 1) / 2, ns = n - 1;
    const int b = blockIdx.x / n, i = blockIdx.x % n;
    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *eaS = reinterpret_cast<float *>(smem);            // [ns][8] exp(el - max1)
    float *ebS = eaS + (size_t)ns * kH;                      // [ns][8] exp(0.2 (el - max1))
    float *xS = ebS + (size_t)ns * kH;                       // [ns] the sources' feature
    int *nodeS = reinterpret_cast<int *>(xS + ns);           // [ns] global node id of slot
    float *top = reinterpret_cast<float *>(nodeS + ns);      // [8][4]: max1, max2, argmax1 (int bits), direct-path flag
    const float *A = img, *bp = img + kD, *al = bp + kD, *bl = al + kH, *ar = bl + kH, *br = ar + kH;
    const float kLog2e = 1.4426950408889634f;
    for (int s = tid; s < ns; s += nthreads) {
        const int k = s < i ? s : s + 1;
        const int node = k < i ? pair_index(k, i, n) : pair_index(i, k, n);
        nodeS[s] = node;
        xS[s] = x[(size_t)b * N + node];
    }
    __syncthreads();
    if (tid < 32 * kH) {   // top-2 of el per head over the row's sources, as in gat_rows_kernel
        const int h = tid >> 5, l32 = tid & 31;
        const float a = al[h], c = bl[h];
        float m1 = -INFINITY, m2 = -INFINITY; int a1 = -1;
        for (int s = l32; s < ns; s += 32) {
            const float v = fmaf(xS[s], a, c);
            if (v > m1) { m2 = m1; m1 = v; a1 = s; } else if (v > m2) { m2 = v; }
        }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) {
            const float om1 = __shfl_xor(m1, o, 32), om2 = __shfl_xor(m2, o, 32);
            const int oa1 = __shfl_xor(a1, o, 32);
            const bool take = om1 > m1 || (om1 == m1 && oa1 >= 0 && (a1 < 0 || oa1 < a1));
            const float lo1 = take ? m1 : om1;
            const float hi2 = take ? om2 : m2;
            if (take) { m1 = om1; a1 = oa1; }
            m2 = lo1 > hi2 ? lo1 : hi2;
        }
        if (l32 == 0) { top[h * 4 + 0] = m1; top[h * 4 + 1] = m2; top[h * 4 + 2] = __int_as_float(a1);
                        top[h * 4 + 3] = (m1 - m2 > 60.f) ? 1.f : 0.f; }
    }
    __syncthreads();
    for (int q = tid; q < ns * kH; q += nthreads) {
        const int h = q % kH;
        const float d = (fmaf(xS[q / kH], al[h], bl[h]) - top[h * 4]) * kLog2e;
        eaS[q] = __builtin_amdgcn_exp2f(d);
        ebS[q] = __builtin_amdgcn_exp2f(kSlope * d);
    }
    __syncthreads();

    constexpr int HU = 4;
    if ((wave >> 1) * 64 >= ns) return;                      // (wavefronts that only helped with the tables: no barrier follows)
    const int h0 = (wave & 1) * HU, js = (wave >> 1) * 64 + lane;     // this lane's destination slot, this wavefront's heads
    const bool dlive = js < ns;
    const int jsc = dlive ? js : ns - 1;
    const float xd = xS[jsc];
    float er[HU], mm[HU], nm[HU], cpos[HU], cneg[HU], ws[HU], wx[HU];
    bool direct = false;
#pragma unroll
    for (int u = 0; u < HU; ++u) {
        const int h = h0 + u;
        er[u] = fmaf(xd, ar[h], br[h]);
        const float m = ((__float_as_int(top[h * 4 + 2]) == js) ? top[h * 4 + 1] : top[h * 4 + 0]) + er[u];
        mm[u] = m > 0.f ? m : kSlope * m;
        nm[u] = -mm[u] * kLog2e;
        const float t = er[u] + top[h * 4 + 0];
        cpos[u] = __builtin_amdgcn_exp2f(fminf(t - mm[u], 80.f) * kLog2e);
        cneg[u] = __builtin_amdgcn_exp2f(fminf(kSlope * t - mm[u], 80.f) * kLog2e);
        ws[u] = 0.f; wx[u] = 0.f;
        direct = direct || top[h * 4 + 3] != 0.f;            // wave-uniform
    }
    if (!direct) {
        auto run = [&](int s_lo, int s_hi, auto masked) {
            constexpr bool MASK = decltype(masked)::value;
            for (int s = s_lo; s < s_hi; ++s) {              // (uniform addresses: one LDS cycle per read)
                const float xs = xS[s];
                const f32x4 ea = *reinterpret_cast<const f32x4 *>(eaS + s * kH + h0);
                const f32x4 eb = *reinterpret_cast<const f32x4 *>(ebS + s * kH + h0);
#pragma unroll
                for (int u = 0; u < HU; ++u) {
                    float w = max_f32(ea[u] * cpos[u], eb[u] * cneg[u]);
                    if (MASK) w = (s == js) ? 0.f : w;           // no self loop
                    ws[u] += w;
                    wx[u] = fmaf(w, xs, wx[u]);
                }
            }
        };
        // only the sources in this wavefront's own destination range can be a lane's own node
        const int d_lo = (wave >> 1) * 64, d_hi = d_lo + 64 < ns ? d_lo + 64 : ns;
        run(0, d_lo, std::false_type{});
        run(d_lo, d_hi, std::true_type{});
        run(d_hi, ns, std::false_type{});
    } else {
        for (int s = 0; s < ns; ++s) {
            const float xs = xS[s];
#pragma unroll
            for (int u = 0; u < HU; ++u) {
                float v = fmaf(xs, al[h0 + u], bl[h0 + u]) + er[u];
                v = fmaxf(v, kSlope * v);                    // LeakyReLU(x) = max(x, 0.2x)
                float w = __builtin_amdgcn_exp2f(fmaf(v, kLog2e, nm[u]));
                w = (s == js) ? 0.f : w;
                ws[u] += w;
                wx[u] = fmaf(w, xs, wx[u]);
            }
        }
    }
    if (dlive) {
        const int j = js < i ? js : js + 1;
        const size_t node = (size_t)nodeS[js];
        const size_t side_stride = (size_t)gridDim.x / n * N;        // B*N nodes per side
        if constexpr (COMPACT) {
            // (shift, sum of weights, sum of weights x feature) per head: all the first layer's feed-forward launch needs (LR0)
            float *qo = part + ((size_t)b * N + (i < j ? 0 : side_stride) + node) * (3 * kH);
#pragma unroll
            for (int u = 0; u < HU; ++u) { qo[h0 + u] = mm[u]; qo[kH + h0 + u] = ws[u]; qo[2 * kH + h0 + u] = wx[u]; }
            return;
        }
        float *po = part + ((size_t)b * N + (i < j ? 0 : side_stride) + node) * kD;
        float *mo = part_ms + ((size_t)b * N + (i < j ? 0 : side_stride) + node) * (2 * kH);
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            const int h = h0 + u;
#pragma unroll
            for (int v4 = 0; v4 < 4; ++v4) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(A + h * kF + 4 * v4), c = *reinterpret_cast<const f32x4 *>(bp + h * kF + 4 * v4);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaf(wx[u], a[e], ws[u] * c[e]);
                *reinterpret_cast<f32x4 *>(po + h * kF + 4 * v4) = o;
            }
            mo[h] = mm[u]; mo[kH + h] = ws[u];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused feed-forward block (models.py:26-36,40), one launch per layer:
//     x  = BN1(h + GATConv(h))          <- log-sum-exp merge of the two attention partials, fused into the
//                                          staging of the x tile (never written to HBM)
//     y  = BN2(x + W2 * ReLU(W1 * x + b1) + b2)
// One workgroup = 64 rows, 4 waves, wave w owns rows [16w, 16w+16); two workgroups per CU (74 KB of LDS each)
// so the memory phases of one overlap the MFMA phases of the other.  The 512-wide hidden layer is processed
// in 4 chunks of 128 and never leaves the registers: GEMM1 is computed TRANSPOSED on v_mfma_f32_16x16x4_f32,
//     Hc^T[hid x rows] = W1c[hid x K] * x^T[K x rows]      (A = W1 tile from LDS, B = x tile from LDS)
// so that in the accumulator the data row sits on the lane (col = lane&15) and the hidden index in
// (lane>>4, register): hid = 16*ht + 4*(lane>>4) + reg.  Those registers are then fed directly as the B operand of
//     Y^T[out x rows] += W2c[out x hid] * Hc^T[hid x rows] (A = W2 tile from LDS, B = accumulator registers)
// (the MFMA's k sub-index IS lane>>4, so register `reg` of tile ht pairs with W2 column 16*ht + 4*(lane>>4) + reg).
// The k order inside GEMM1 is permuted the same way (lane group q walks k = 16*blk + 4*q + s, s = 0..3) so that ONE
// ds_read_b128 feeds four MFMA steps for A and for B; every product a[k]*b[k] is still formed exactly once.
// Weight tiles (128 x 32 fp32 = 16 KB) stream L2 -> registers -> LDS through a 2-deep ring, one barrier per
// tile, 64 MFMAs (2048 cycles) per wave between barriers.
// ---------------------------------------------------------------------------------------------
constexpr int FT_M = 64;             // rows per workgroup
constexpr int LDX = 132;             // x tile row stride (floats): 16-B aligned rows; a ds_read_b128 of 16 consecutive
                                     // rows at one k offset touches 16 disjoint groups of 4 banks (132 = 4 mod 64)
constexpr int LDW = 36;              // weight tile row stride (36*r mod 64 hits every multiple of 4 once per 16 rows)

template <int SUB>
__device__ __forceinline__ void ffn_gemm1_stage(f32x4 (&accH)[8], const float *Xs, const float *Wt, int wrow, int lr, int lq) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const f32x4 b = *reinterpret_cast<const f32x4 *>(Xs + (wrow + lr) * LDX + SUB * 32 + 16 * blk + 4 * lq);
        f32x4 a[8];
#pragma unroll
        for (int ht = 0; ht < 8; ++ht) a[ht] = *reinterpret_cast<const f32x4 *>(Wt + (ht * 16 + lr) * LDW + 16 * blk + 4 * lq);
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int ht = 0; ht < 8; ++ht)
                accH[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ht][st], b[st], accH[ht], 0, 0, 0);
    }
}

template <int SUB>
__device__ __forceinline__ void ffn_gemm2_stage(f32x4 (&accY)[8], const f32x4 (&accH)[8], const float *Wt, int lr, int lq) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {            // the two 16-wide hidden tiles covered by this 32-wide W2 tile
        f32x4 a[8];
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) a[ot] = *reinterpret_cast<const f32x4 *>(Wt + (ot * 16 + lr) * LDW + 16 * hh + 4 * lq);
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const float b = accH[2 * SUB + hh][st];
#pragma unroll
            for (int ot = 0; ot < 8; ++ot)
                accY[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ot][st], b, accY[ot], 0, 0, 0);
        }
    }
}

// (Measured and rejected, r01i: a mapping where a wave owns 32 rows x one half of the hidden tiles, so each weight fragment
// read from LDS feeds two row tiles -- 41 % fewer ds_read_b128, partial outputs exchanged once in the epilogue: 196 VGPRs,
// 11.5 ms vs 11.1 ms per 1024 TSP100 instances.  LDS fragment traffic is not what holds this kernel at 75 % of the MFMA peak.)
// Modes.  FFN_INFER: as above.  FFN_TRAIN_FWD: the x tile is BN1 applied to an already merged h1 = h + GATConv(h) passed
// in `hin` (part / part_ms unused; the batch statistics of h1 must be known before BN1 can be applied) and the hidden
// activations ReLU(W1 x + b1) are also written to `hid_out` [M,512] for the backward.  FFN_BWD: the SAME two chained
// GEMMs compute the data gradient of the block, d x = ((d h3 * W2) . [hidden > 0]) * W1 + d h3, when called with
// hin = d h3, W1 := W2^T [512,128], W2 := W1^T [128,512], zero biases and identity affines: the "activation" becomes the
// ReLU mask read from `hid_in` (the saved activations) and the masked hidden gradient is written to `hid_out` (it may
// alias hid_in) for the weight-gradient GEMMs.
enum { FFN_INFER = 0, FFN_TRAIN_FWD = 1, FFN_BWD = 2 };

template <int MODE>
__global__ __launch_bounds__(256, 2) void ffn_fused_kernel(const float *__restrict__ part, const float *__restrict__ part_ms,
                                                           const float *__restrict__ hin,
                                                           const float *__restrict__ bn1_s, const float *__restrict__ bn1_b,
                                                           const float *__restrict__ W1, const float *__restrict__ b1,
                                                           const float *__restrict__ W2, const float *__restrict__ b2,
                                                           const float *__restrict__ bn2_s, const float *__restrict__ bn2_b,
                                                           float *__restrict__ hout, long M, const float *hid_in,
                                                           float *hid_out) {
    constexpr bool PRE = MODE != FFN_INFER;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *Xs = reinterpret_cast<float *>(smem_raw);        // [64][LDX]
    float *Wb0 = Xs + FT_M * LDX;                           // [128][LDW]
    float *Wb1 = Wb0 + 128 * LDW;
    float *vecs = Wb1 + 128 * LDW;                          // b1[512] b2[128] bn2_s[128] bn2_b[128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int wrow = wave * 16;
    const long row0 = (long)blockIdx.x * FT_M;

    for (int q = tid; q < 512; q += 256) vecs[q] = b1[q];
    if (tid < 128) { vecs[512 + tid] = b2[tid]; vecs[640 + tid] = bn2_s[tid]; vecs[768 + tid] = bn2_b[tid]; }

    // ---- stage the x tile: x = BN1(h + merge(partials))  (gat_combine fused; models.py:15,24,28) ----
    // 8 float4 slots per thread, loaded in batches of 4 (all loads of a batch in flight together)
    constexpr int PB = 4;
    for (int it0 = 0; it0 < (FT_M * 32) / 256; it0 += PB) {
        f32x4 p0[PB], p1[PB], hv[PB];
        float m0[PB], s0[PB], m1[PB], s1[PB];
        bool live[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int idx = (it0 + u) * 256 + tid;
            const int row = idx >> 5, c = (idx & 31) * 4, hd = c >> 4;
            const long m = row0 + row;
            live[u] = m < M;
            const long mc = live[u] ? m : 0;
            if (!PRE) {
                const float *ms0 = part_ms + mc * (2 * kH), *ms1 = part_ms + (M + mc) * (2 * kH);
                m0[u] = ms0[hd]; s0[u] = ms0[kH + hd]; m1[u] = ms1[hd]; s1[u] = ms1[kH + hd];
                p0[u] = *reinterpret_cast<const f32x4 *>(part + mc * kD + c);
                p1[u] = *reinterpret_cast<const f32x4 *>(part + (M + mc) * kD + c);
            }
            hv[u] = *reinterpret_cast<const f32x4 *>(hin + mc * kD + c);
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int idx = (it0 + u) * 256 + tid;
            const int row = idx >> 5, c = (idx & 31) * 4;
            float a0 = 0.f, a1 = 0.f, inv = 0.f;
            if (!PRE) {
                const float mx = m0[u] > m1[u] ? m0[u] : m1[u];
                a0 = __expf(m0[u] - mx); a1 = __expf(m1[u] - mx);
                inv = 1.f / (s0[u] * a0 + s1[u] * a1);
            }
            f32x4 o4;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float g = PRE ? 0.f : (p0[u][v] * a0 + p1[u][v] * a1) * inv;
                const float o = (PRE ? hv[u][v] : hv[u][v] + g) * bn1_s[c + v] + bn1_b[c + v];
                o4[v] = live[u] ? o : 0.f;
            }
            *reinterpret_cast<f32x4 *>(Xs + row * LDX + c) = o4;
        }
    }

    // ---- weight tile stream: tile t in [0,32): chunk c = t>>3, phase (t>>2)&1 (0: W1, 1: W2), sub = t&3 ----
    const int srow = tid >> 1, sk = (tid & 1) * 16;
    f32x4 rw[4];
    auto gload = [&](int t) {
        const int c = t >> 3, ph = (t >> 2) & 1, sub = t & 3;
        const float *src = ph == 0 ? W1 + (long)(c * 128 + srow) * 128 + sub * 32 + sk       // W1[hid][k]
                                   : W2 + (long)srow * 512 + c * 128 + sub * 32 + sk;        // W2[out][hid]
#pragma unroll
        for (int u = 0; u < 4; ++u) rw[u] = *reinterpret_cast<const f32x4 *>(src + 4 * u);
    };
    auto lstore = [&](float *Wt) {
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4 *>(Wt + srow * LDW + sk + 4 * u) = rw[u];
    };
    gload(0);
    lstore(Wb0);
    __syncthreads();

    f32x4 accY[8], accH[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) accY[a] = f32x4{0.f, 0.f, 0.f, 0.f};

#define FFN_STAGE(S, BODY)                                                                  \
    {                                                                                       \
        const int t = c * 8 + (S);                                                          \
        if (t + 1 < 32) gload(t + 1);                                                       \
        const float *Wt = ((S) & 1) ? Wb1 : Wb0;                                            \
        BODY;                                                                               \
        if (t + 1 < 32) lstore(((S) & 1) ? Wb0 : Wb1);                                      \
        __syncthreads();                                                                    \
    }

#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        // hidden pre-activations start at the bias (C-in of the MFMA chain), models.py:30
#pragma unroll
        for (int ht = 0; ht < 8; ++ht) accH[ht] = *reinterpret_cast<const f32x4 *>(vecs + c * 128 + ht * 16 + 4 * lq);
        FFN_STAGE(0, (ffn_gemm1_stage<0>(accH, Xs, Wt, wrow, lr, lq)))
        FFN_STAGE(1, (ffn_gemm1_stage<1>(accH, Xs, Wt, wrow, lr, lq)))
        FFN_STAGE(2, (ffn_gemm1_stage<2>(accH, Xs, Wt, wrow, lr, lq)))
        FFN_STAGE(3, (ffn_gemm1_stage<3>(accH, Xs, Wt, wrow, lr, lq)))
        if (MODE == FFN_BWD) {
            // ReLU backward: keep d hidden where the saved activation is positive (8 float4 loads in flight per lane)
            const long hrow = row0 + wrow + lr;
            const bool ok = hrow < M;
            f32x4 msk[8];
#pragma unroll
            for (int ht = 0; ht < 8; ++ht)
                msk[ht] = ok ? *reinterpret_cast<const f32x4 *>(hid_in + hrow * 512 + c * 128 + ht * 16 + 4 * lq)
                             : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ht = 0; ht < 8; ++ht)
#pragma unroll
                for (int r = 0; r < 4; ++r) accH[ht][r] = msk[ht][r] > 0.f ? accH[ht][r] : 0.f;
        } else {
#pragma unroll
            for (int ht = 0; ht < 8; ++ht)
#pragma unroll
                for (int r = 0; r < 4; ++r) accH[ht][r] = accH[ht][r] > 0.f ? accH[ht][r] : 0.f;   // ReLU, models.py:31
        }
        if (MODE != FFN_INFER) {
            // lane (lr, lq) holds hidden units 16*ht + 4*lq .. +3 of data row lr: one 16-byte store each, the four lq lanes
            // of a row complete a 64-byte segment
            const long hrow = row0 + wrow + lr;
            if (hrow < M) {
#pragma unroll
                for (int ht = 0; ht < 8; ++ht)
                    *reinterpret_cast<f32x4 *>(hid_out + hrow * 512 + c * 128 + ht * 16 + 4 * lq) = accH[ht];
            }
        }
        FFN_STAGE(4, (ffn_gemm2_stage<0>(accY, accH, Wt, lr, lq)))
        FFN_STAGE(5, (ffn_gemm2_stage<1>(accY, accH, Wt, lr, lq)))
        FFN_STAGE(6, (ffn_gemm2_stage<2>(accY, accH, Wt, lr, lq)))
        FFN_STAGE(7, (ffn_gemm2_stage<3>(accY, accH, Wt, lr, lq)))
    }
#undef FFN_STAGE

    // ---- epilogue: y = BN2(x + (acc + b2)); transposed accumulator -> x tile in LDS -> coalesced rows ----
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) {
        const int out = ot * 16 + 4 * lq;
        float *xp = Xs + (wrow + lr) * LDX + out;
        f32x4 x = *reinterpret_cast<const f32x4 *>(xp);
        const f32x4 bb = *reinterpret_cast<const f32x4 *>(vecs + 512 + out);
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(vecs + 640 + out);
        const f32x4 sh = *reinterpret_cast<const f32x4 *>(vecs + 768 + out);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = accY[ot][r] + bb[r];            // Linear2 output (models.py:32)
            v = x[r] + v;                             // x + y            (models.py:15)
            x[r] = v * sc[r] + sh[r];                 // BatchNorm1d eval (models.py:35)
        }
        *reinterpret_cast<f32x4 *>(xp) = x;
    }
    __syncthreads();
    for (int it = 0; it < (FT_M * 32) / 256; ++it) {
        const int idx = it * 256 + tid;
        const int row = idx >> 5, c = (idx & 31) * 4;
        const long m = row0 + row;
        if (m < M) *reinterpret_cast<f32x4 *>(hout + m * kD + c) = *reinterpret_cast<const f32x4 *>(Xs + row * LDX + c);
    }
}

// ---------------------------------------------------------------------------------------------
// The same block on the bf16 matrix pipe with fp32 accuracy ("bf16x3 operands, six products").  CDNA4 runs
// v_mfma_f32_16x16x32_bf16 at SIXTEEN times the fp32 MFMA rate (2.5 PFLOP/s against 157 TFLOP/s, and no xf32 form): an fp32
// value is the sum of three bf16 pieces, x = x0 + x1 + x2 (8 + 8 + 8 significand bits: x0 = bf16(x), x1 = bf16(x - x0),
// x2 = bf16(x - x0 - x1); bf16 has the fp32 exponent, so nothing over- or underflows), every piece product is exact in the
// fp32 accumulator, and a b = a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0 + O(2^-23 |a b|): six MFMAs of K = 32 (96 cycles of
// the matrix pipe) do the work of eight fp32 MFMAs of K = 4 (256 cycles) with the rounding error of fp32 arithmetic -- measured
// against the fp64 oracle the block's error is that of the fp32 kernel (tests/test_model_gpu.py, unchanged 1e-5 bar; the numpy
// model of both paths: 4.5e-7 against 4.8e-7 of max|y|; three products, i.e. two pieces, would be 6.5e-6).  Inference only; the
// training kernels above stay on the fp32 pipe.
// The weights are split and laid out in fragment order by ffn_pack_bf16x3_kernel, once per weight image (gnngls_regret_prepare; 786 KB per layer):
//     W1p[c][kb][ht][p][lane][8]   = piece p of W1[128 c + 16 ht + (lane & 15)][32 kb + 16 (e >> 2) + 4 (lane >> 4) + (e & 3)]
//     W2p[c][j][ot][p][lane][8]    = piece p of W2[16 ot + (lane & 15)][128 c + 32 j + 16 (e >> 2) + 4 (lane >> 4) + (e & 3)]
// so that a stage's 24 KB are a flat copy into LDS and a wavefront's A fragment is one conflict-free ds_read_b128.  The k order of
// W2p is the accumulator layout of GEMM1 (lane (row, q) holds hidden units 16 ht + 4 q + r of its row): as in the fp32 kernel
// the hidden layer goes from the accumulators of GEMM1 through ReLU and the split straight into the B operand of GEMM2.
// One workgroup = 128 rows, 8 wavefronts of 16 rows (two per SIMD), one workgroup per CU (142 KB of LDS: the x tile in fp32 for the
// skip connection, split on the fly per k block; a ring of three 24 KB weight stages filled by LDS-DMA).  With the next layer's fc
// weights given, four more stages chain ft = fc(h_out) (models.py:23 of layer l + 1) from the output's accumulator layout in the same
// way and write it next to h_out: gemm_fc runs once per forward.  Measured at 1024 x TSP100 (profiles/r05_experiments/README.md):
// 8.2 ms per launch including the fc against 11.2 + 1.8 ms of the fp32 kernels, 180 fp32-equivalent TFLOP/s; matrix pipe 50 % busy.
// ---------------------------------------------------------------------------------------------
#ifndef FFN_DBG
#define FFN_DBG 0                    // bits: 1 no main loop, 2 no HBM traffic in the prologue / epilogue, 4 no weight streaming, 8 no MFMAs, 16 no fragment reads, 32 no operand split (time attribution builds)
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#ifndef FB_M_ROWS
#define FB_M_ROWS 128                // rows per workgroup: 128 (8 wavefronts, one workgroup per CU) or 64 (4 wavefronts, two per CU: the ring takes the x tile's place; A/B builds)
#endif
constexpr int FB_M = FB_M_ROWS;                     // rows per workgroup
constexpr int FB_T = FB_M * 4, FB_W = FB_M / 16;      // threads, wavefronts (16 rows each)
constexpr bool kFbAlias = FB_M < 128;                // the weight ring in the place of the x tile (staging area of the prologue only)
constexpr int FB_STAGE = 8 * 3 * 64 * 16;          // bytes of one weight stage: 8 tiles x 3 pieces x 64 lanes x 16 B
constexpr size_t kFfnPackedBytes = (size_t)(2 * 16 + 4) * FB_STAGE;  // W1p + W2p + the next layer's fc: 884,736 B

__device__ __forceinline__ void split_bf16x3(const float (&x)[8], bf16x8 &p0, bf16x8 &p1, bf16x8 &p2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h0 = (__bf16)x[e];
        const float r1 = x[e] - (float)h0;               // exact
        const __bf16 h1 = (__bf16)r1;
        const float r2 = r1 - (float)h1;                 // exact
        p0[e] = h0; p1[e] = h1; p2[e] = (__bf16)r2;
    }
}
// acc += a b with a = a0 + a1 + a2, b = b0 + b1 + b2 (the three products of order 2^-24 and below are dropped); small terms first
__device__ __forceinline__ f32x4 mfma_bf16x3(const bf16x8 &a0, const bf16x8 &a1, const bf16x8 &a2, const bf16x8 &b0,
                                             const bf16x8 &b1, const bf16x8 &b2, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc, 0, 0, 0);
    return acc;
}

// one thread per (matrix, chunk, stage, tile, lane): the three pieces of its eight weights
__global__ void ffn_pack_bf16x3_kernel(const float *__restrict__ W1, const float *__restrict__ W2, unsigned char *__restrict__ packed) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;             // 2 x 4 x 4 x 8 x 64 = 16384 threads
    const int lane = g & 63, tile = (g >> 6) & 7, sub = (g >> 9) & 3, c = (g >> 11) & 3, second = g >> 13;
    const int lr = lane & 15, lq = lane >> 4;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
        v[e] = second ? W2[(long)(tile * 16 + lr) * 512 + c * 128 + 32 * sub + 16 * (e >> 2) + 4 * lq + (e & 3)]
                      : W1[(long)(c * 128 + tile * 16 + lr) * 128 + 32 * sub + 16 * (e >> 2) + 4 * lq + (e & 3)];
    bf16x8 p0, p1, p2;
    split_bf16x3(v, p0, p1, p2);
    unsigned char *dst = packed + (size_t)second * 16 * FB_STAGE + (size_t)(c * 4 + sub) * FB_STAGE + (size_t)(tile * 3) * 1024 + lane * 16;
    *reinterpret_cast<bf16x8 *>(dst) = p0;
    *reinterpret_cast<bf16x8 *>(dst + 1024) = p1;
    *reinterpret_cast<bf16x8 *>(dst + 2048) = p2;
}

// the next layer's fc weights [128 f][128 k] in the W2p form: Wfp[j][tile][p][lane][8] = piece p of Wfc[16 tile + (lane & 15)][32 j + 16 (e >> 2) + 4 (lane >> 4) + (e & 3)]
__global__ void ffn_pack_fc_bf16x3_kernel(const float *__restrict__ Wfc, unsigned char *__restrict__ packed_fc) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;             // 4 x 8 x 64 = 2048 threads
    const int lane = g & 63, tile = (g >> 6) & 7, j = g >> 9;
    const int lr = lane & 15, lq = lane >> 4;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = Wfc[(long)(tile * 16 + lr) * 128 + 32 * j + 16 * (e >> 2) + 4 * lq + (e & 3)];
    bf16x8 p0, p1, p2;
    split_bf16x3(v, p0, p1, p2);
    unsigned char *dst = packed_fc + (size_t)j * FB_STAGE + (size_t)(tile * 3) * 1024 + lane * 16;
    *reinterpret_cast<bf16x8 *>(dst) = p0;
    *reinterpret_cast<bf16x8 *>(dst + 1024) = p1;
    *reinterpret_cast<bf16x8 *>(dst + 2048) = p2;
}

// LR0 (the first layer of a model with ONE input feature, behind gat_rows_rank1_kernel<true>): `part` holds that kernel's compact
// partials [2][M][24] = (shift, sum of weights, sum of weights x feature) per head, `hin` the [M] input features; the layer's input
// h = x We + be (models.py:66, embed_kernel's arithmetic) and its GATConv output rho A + b' (rho = the softmax-weighted mean of the
// neighbours' features, per head) are formed here from 13 scalars per row instead of 1.7 KB -- neither h nor ft nor the 128-wide
// partials of the first layer exist in memory.
template <bool LR0>
__global__ __launch_bounds__(FB_T, kFbAlias ? 2 : 1) void ffn_fused_bf16x3_kernel(const float *__restrict__ part, const float *__restrict__ part_ms,
                                                                  const float *__restrict__ hin,
                                                                  const float *__restrict__ bn1_s, const float *__restrict__ bn1_b,
                                                                  const unsigned char *__restrict__ packed, const float *__restrict__ b1,
                                                                  const float *__restrict__ b2,
                                                                  const float *__restrict__ bn2_s, const float *__restrict__ bn2_b,
                                                                  float *__restrict__ hout, long M,
                                                                  const unsigned char *__restrict__ packed_fc, float *__restrict__ ft_out,
                                                                  const float *__restrict__ dec_w, const float *__restrict__ dec_b,
                                                                  float *__restrict__ y_out,
                                                                  const float *__restrict__ lr_img, const float *__restrict__ emb_w,
                                                                  const float *__restrict__ emb_b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *Xs = reinterpret_cast<float *>(smem_raw);                               // [128][LDX] fp32
    unsigned char *Wb0 = smem_raw + (kFbAlias ? 0 : (size_t)FB_M * LDX * sizeof(float));      // ring of three weight stages
    float *vecs = reinterpret_cast<float *>(Wb0 + 3 * FB_STAGE);                   // b1[512] b2[128] bn2_s[128] bn2_b[128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int wrow = wave * 16;
    const long row0 = (long)blockIdx.x * FB_M;

    for (int q = tid; q < 512; q += FB_T) vecs[q] = b1[q];
    if (tid < 128) { vecs[512 + tid] = b2[tid]; vecs[640 + tid] = bn2_s[tid]; vecs[768 + tid] = bn2_b[tid]; }

    // ---- stage the x tile: x = BN1(h + merge(partials))  (gat_combine fused; models.py:15,24,28), as in ffn_fused_kernel ----
    constexpr int PB = 4;
    if constexpr (!LR0) {
    for (int it0 = 0; it0 < (FB_M * 32) / FB_T; it0 += PB) {
        f32x4 p0[PB], p1[PB], hv[PB];
        float m0[PB], s0[PB], m1[PB], s1[PB];
        bool live[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int idx = (it0 + u) * FB_T + tid;
            const int row = idx >> 5, c = (idx & 31) * 4, hd = c >> 4;
            const long m = row0 + row;
            live[u] = m < M;
            const long mc = (FFN_DBG & 2) ? 0 : (live[u] ? m : 0);
            const float *ms0 = part_ms + mc * (2 * kH), *ms1 = part_ms + (M + mc) * (2 * kH);
            m0[u] = ms0[hd]; s0[u] = ms0[kH + hd]; m1[u] = ms1[hd]; s1[u] = ms1[kH + hd];
            p0[u] = *reinterpret_cast<const f32x4 *>(part + mc * kD + c);
            p1[u] = *reinterpret_cast<const f32x4 *>(part + (M + mc) * kD + c);
            hv[u] = *reinterpret_cast<const f32x4 *>(hin + mc * kD + c);
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int idx = (it0 + u) * FB_T + tid;
            const int row = idx >> 5, c = (idx & 31) * 4;
            const float mx = m0[u] > m1[u] ? m0[u] : m1[u];
            const float a0 = __expf(m0[u] - mx), a1 = __expf(m1[u] - mx);
            const float inv = 1.f / (s0[u] * a0 + s1[u] * a1);
            f32x4 o4;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float g = (p0[u][v] * a0 + p1[u][v] * a1) * inv;
                const float o = (hv[u][v] + g) * bn1_s[c + v] + bn1_b[c + v];
                o4[v] = live[u] ? o : 0.f;
            }
            *reinterpret_cast<f32x4 *>(Xs + row * LDX + c) = o4;
        }
    }
    } else {
        const int c = (tid & 31) * 4, hd = c >> 4;           // (the columns of a thread are the same in every slot)
        const f32x4 we = f32x4{emb_w[c], emb_w[c + 1], emb_w[c + 2], emb_w[c + 3]}, be4 = *reinterpret_cast<const f32x4 *>(emb_b + c);
        const f32x4 a4 = *reinterpret_cast<const f32x4 *>(lr_img + c), bp4 = *reinterpret_cast<const f32x4 *>(lr_img + kD + c);
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(bn1_s + c), sh = *reinterpret_cast<const f32x4 *>(bn1_b + c);
        for (int it0 = 0; it0 < (FB_M * 32) / FB_T; it0 += PB) {
            float xv[PB], m0[PB], s0[PB], w0[PB], m1[PB], s1[PB], w1[PB];
            bool live[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int row = ((it0 + u) * FB_T + tid) >> 5;
                const long m = row0 + row;
                live[u] = m < M;
                const long mc = (FFN_DBG & 2) ? 0 : (live[u] ? m : 0);
                const float *q0 = part + mc * (3 * kH), *q1 = part + (M + mc) * (3 * kH);
                xv[u] = hin[mc];
                m0[u] = q0[hd]; s0[u] = q0[kH + hd]; w0[u] = q0[2 * kH + hd];
                m1[u] = q1[hd]; s1[u] = q1[kH + hd]; w1[u] = q1[2 * kH + hd];
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int row = ((it0 + u) * FB_T + tid) >> 5;
                const float mx = m0[u] > m1[u] ? m0[u] : m1[u];
                const float a0 = __expf(m0[u] - mx), a1 = __expf(m1[u] - mx);
                const float inv = 1.f / (s0[u] * a0 + s1[u] * a1);
                const float rho = (w0[u] * a0 + w1[u] * a1) * inv;
                f32x4 o4;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float hv = fmaf(xv[u], we[v], 0.f) + be4[v];            // models.py:66, as embed_kernel
                    const float g = fmaf(rho, a4[v], bp4[v]);                      // GATConv of layer 0: rho A + b'
                    const float o = (hv + g) * sc[v] + sh[v];
                    o4[v] = live[u] ? o : 0.f;
                }
                *reinterpret_cast<f32x4 *>(Xs + row * LDX + c) = o4;
            }
        }
    }

    // ---- weight stage stream: stage t in [0,32): chunk c = t>>3, phase (t>>2)&1 (0: W1p, 1: W2p), sub = t&3; a flat 24 KB copy.
    // Ring of THREE stage buffers: stage t computes from buffer t % 3 while the copy of stage t + 2 is in flight (issued at the start of
    // stage t, waited for at its end), one barrier per stage.  What a wavefront needs to START stage t + 1 -- the pieces
    // of its B operand and the A fragments of the first tile -- is prepared before that barrier (buffer (t + 1) % 3 has been
    // complete since the barrier before), so the matrix pipe does not idle through an LDS round trip and a split after every
    // barrier (measured: 2.5k cycles per stage for 1.5k cycles of MFMAs without this).
    const int nst = packed_fc ? 36 : 32;                 // stages 32 .. 35: the NEXT layer's fc (GATConv's linear map, models.py:23) on this layer's output
    // LDS-DMA (global_load_lds_dwordx4: every wavefront instruction moves 1 KB, lane l's 16 bytes to LDS base + 16 l): no staging
    // registers, no ds_write, the copy is tracked by vmcnt alone
    auto dma_stage = [&](int t, unsigned char *Wt) {
        const int c = t >> 3, ph = (t >> 2) & 1, sub = t & 3;
        const unsigned char *src = t < 32 ? packed + (size_t)ph * 16 * FB_STAGE + (size_t)(c * 4 + sub) * FB_STAGE
                                          : packed_fc + (size_t)(t - 32) * FB_STAGE;
#pragma unroll
        for (int u = 0; u < 24 / FB_W; ++u) {
            const int piece = (u * FB_W + wave) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + piece + lane * 16),
                                             (__attribute__((address_space(3))) void *)(Wt + piece), 16, 0, 0);
        }
    };
    unsigned char *const Wb = Wb0;                       // ring base: buffers at Wb + {0, 1, 2} * FB_STAGE
    if constexpr (!kFbAlias) {
        dma_stage(0, Wb);
        dma_stage(1, Wb + FB_STAGE);
        __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): the copies have landed
    }
    __syncthreads();

    f32x4 accY[8], accH[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) accY[a] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 q0, q1, q2;                                   // B pieces of the current stage
    bf16x8 fa0, fa1, fa2;                                // A fragments of the current stage's first tile
    // The wavefront's x rows in the accumulator layout, in registers for the whole block: lane (row, q) holds columns 16 ot + 4 q + r.
    // W1p's k order is that layout (as W2p's is for the hidden layer), so xa is the B operand of GEMM1 AND the skip connection of the
    // epilogue; the x tile in LDS is only the staging area of the prologue's coalesced loads.
    f32x4 xa[8];
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) xa[ot] = *reinterpret_cast<const f32x4 *>(Xs + (wrow + lr) * LDX + ot * 16 + 4 * lq);
    if constexpr (kFbAlias) {                            // the x tile has been read: the ring may take its place
        __syncthreads();
        dma_stage(0, Wb);
        dma_stage(1, Wb + FB_STAGE);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
    }
    auto x_values = [&](int kb, float (&v)[8]) {         // GEMM1, k block kb: columns 32 kb + 16 (e >> 2) + 4 q + (e & 3)
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = xa[2 * kb][e]; v[4 + e] = xa[2 * kb + 1][e]; }
    };
    bf16x8 fb0, fb1, fb2;                                // ... and of its second tile (fragments are read two tiles ahead: one tile's
                                                         // six MFMAs, 96 cycles, do not cover an LDS round trip with eight wavefronts reading)
    auto first_frags = [&](const unsigned char *Wt) {
        const unsigned char *f = Wt + lane * 16;
        fa0 = *reinterpret_cast<const bf16x8 *>(f);
        fa1 = *reinterpret_cast<const bf16x8 *>(f + 1024);
        fa2 = *reinterpret_cast<const bf16x8 *>(f + 2048);
        fb0 = *reinterpret_cast<const bf16x8 *>(f + 3072);
        fb1 = *reinterpret_cast<const bf16x8 *>(f + 4096);
        fb2 = *reinterpret_cast<const bf16x8 *>(f + 5120);
    };
    // One stage's MFMAs with the issue order fixed by hand (nothing crosses a sched_barrier): the fragments two tiles ahead -- tiles 2 .. 7 of
    // this stage, then tiles 0 and 1 of the NEXT stage's buffer (complete since the last barrier), so that the wave reaches the barrier with
    // the next stage's first operands in registers --, the six MFMAs of tile tl, and a slice of the next stage's split in the issue slots the
    // MFMAs leave free (an MFMA of 16 cycles holds the issue port for 8)
    auto run_tiles = [&](f32x4 (&acc)[8], const unsigned char *Wt, const unsigned char *Wnext, auto &&slice) {
        bf16x8 a0 = fa0, a1 = fa1, a2 = fa2, b0 = fb0, b1 = fb1, b2 = fb2;
#pragma unroll
        for (int tl = 0; tl < 8; ++tl) {
            const unsigned char *f = (tl < 6 ? Wt + (size_t)((tl + 2) * 3) * 1024 : Wnext + (size_t)((tl - 6) * 3) * 1024) + lane * 16;
            bf16x8 m0 = a0, m1 = a1, m2 = a2;
            if (!(FFN_DBG & 16)) {
                m0 = *reinterpret_cast<const bf16x8 *>(f);
                m1 = *reinterpret_cast<const bf16x8 *>(f + 1024);
                m2 = *reinterpret_cast<const bf16x8 *>(f + 2048);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(FFN_DBG & 8)) acc[tl] = mfma_bf16x3(a0, a1, a2, q0, q1, q2, acc[tl]);
            if (tl >= 3 && tl < 7 && !(FFN_DBG & 32)) slice(tl - 3);
            __builtin_amdgcn_sched_barrier(0);
            a0 = b0; a1 = b1; a2 = b2;
            b0 = m0; b1 = m1; b2 = m2;
        }
        fa0 = a0; fa1 = a1; fa2 = a2; fb0 = b0; fb1 = b1; fb2 = b2;
    };
    // elements 2 k, 2 k + 1 of a B operand: the three bf16 pieces of v[.]
    auto split_pair = [&](const float (&v)[8], int k, bf16x8 &n0, bf16x8 &n1, bf16x8 &n2) {
#pragma unroll
        for (int e = 2 * k; e < 2 * k + 2; ++e) {
            const __bf16 h0 = (__bf16)v[e];
            const float r1 = v[e] - (float)h0;
            const __bf16 h1 = (__bf16)r1;
            const float r2 = r1 - (float)h1;
            n0[e] = h0; n1[e] = h1; n2[e] = (__bf16)r2;
        }
    };
    {
        float v[8];
        x_values(0, v);
        split_bf16x3(v, q0, q1, q2);
    }
    first_frags(Wb);
    int rb = 0;                                          // ring slot of the current stage (wave-uniform)

#pragma unroll 1
    for (int c = 0; c < ((FFN_DBG & 1) ? 0 : 4); ++c) {
        // hidden pre-activations start at the bias (C-in of the MFMA chain), models.py:30
#pragma unroll
        for (int ht = 0; ht < 8; ++ht) accH[ht] = *reinterpret_cast<const f32x4 *>(vecs + c * 128 + ht * 16 + 4 * lq);
#pragma unroll
        for (int sub = 0; sub < 8; ++sub) {
            const int t = c * 8 + sub;
            const int rb1 = rb == 2 ? 0 : rb + 1, rb2 = rb1 == 2 ? 0 : rb1 + 1;
            if (t + 2 < nst && !(FFN_DBG & 4)) dma_stage(t + 2, Wb + rb2 * FB_STAGE);      // (buffer rb2 was last read in stage t - 1)
            __builtin_amdgcn_sched_barrier(0);           // the copy is issued at the start of the stage
            const unsigned char *Wt = Wb + rb * FB_STAGE;
            // the B operand of the NEXT stage, prepared in four slices under this stage's MFMAs: x values of the next k block (sub 0..2, and
            // 7: the next chunk's first), or ReLU(hidden) of block j = sub - 3, whose two tiles are complete once this stage's tiles 0 and 1
            // are (sub 3) or have been since the last GEMM1 stage (sub 4..6), models.py:31
            float v[8];
            if (sub < 3) x_values(sub + 1, v);
            else if (sub == 7) x_values(0, v);
            bf16x8 n0 = q0, n1 = q1, n2 = q2;            // (attribution builds without the split keep the last operand)
            auto slice = [&](int k) {                    // elements 2 k, 2 k + 1
                if (sub >= 3 && sub < 7) {
                    const int j = sub - 3, hh = k >> 1, e0 = 2 * (k & 1);
                    const float u0 = accH[2 * j + hh][e0], u1 = accH[2 * j + hh][e0 + 1];
                    v[2 * k] = u0 > 0.f ? u0 : 0.f; v[2 * k + 1] = u1 > 0.f ? u1 : 0.f;
                }
                split_pair(v, k, n0, n1, n2);
            };
            if (sub < 4) run_tiles(accH, Wt, Wb + rb1 * FB_STAGE, slice);
            else run_tiles(accY, Wt, Wb + rb1 * FB_STAGE, slice);
            q0 = n0; q1 = n1; q2 = n2;
            __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): this wavefront's share of stage t + 2 has landed
            __syncthreads();
            rb = rb1;
        }
    }

    // ---- epilogue: y = BN2(x + (acc + b2)), stored from the accumulator layout: a lane's four columns of a 16-column tile are one
    // 16-byte store, the eight tiles of a row complete its 512 bytes in L2 ----
    const long mrow = row0 + wrow + lr;
    const bool rlive = mrow < M && (!(FFN_DBG & 2) || mrow == 0);
    float ydot = 0.f;
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) {
        const int out = ot * 16 + 4 * lq;
        f32x4 x = xa[ot];
        const f32x4 bb = *reinterpret_cast<const f32x4 *>(vecs + 512 + out);
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(vecs + 640 + out);
        const f32x4 sh = *reinterpret_cast<const f32x4 *>(vecs + 768 + out);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = accY[ot][r] + bb[r];            // Linear2 output (models.py:32)
            v = x[r] + v;                             // x + y            (models.py:15)
            x[r] = v * sc[r] + sh[r];                 // BatchNorm1d eval (models.py:35)
        }
        // the LAST layer's output goes straight into the decision layer (models.py:63,69: y = h . w + b): no store of h
        if (dec_w) {
            const f32x4 dw = *reinterpret_cast<const f32x4 *>(dec_w + out);
#pragma unroll
            for (int r = 0; r < 4; ++r) ydot = fmaf(x[r], dw[r], ydot);
        } else if (rlive) *reinterpret_cast<f32x4 *>(hout + mrow * kD + out) = x;
        accH[ot] = x;                                 // (the layer's output in the accumulator layout: the B operand of the fc stages)
    }
    if (dec_w) {                                      // the row's four column quarters sit on lanes lr, lr + 16, lr + 32, lr + 48
        ydot += __shfl_xor(ydot, 16);
        ydot += __shfl_xor(ydot, 32);
        if (rlive && lq == 0) y_out[mrow] = ydot + dec_b[0];
    }
    if (packed_fc) {
        // ---- the next layer's ft = fc(h) chained from the registers exactly as GEMM2 is chained from GEMM1: lane (row, q) holds outputs
        // 16 ot + 4 q + r of its row, which is the k order the packed fc weights are laid out in (ffn_pack_fc_bf16x3_kernel) ----
#pragma unroll
        for (int a = 0; a < 8; ++a) accY[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = accH[0][e]; v[4 + e] = accH[1][e]; }
            split_bf16x3(v, q0, q1, q2);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = 32 + j;
            const int rb1 = rb == 2 ? 0 : rb + 1, rb2 = rb1 == 2 ? 0 : rb1 + 1;
            if (t + 2 < nst && !(FFN_DBG & 4)) dma_stage(t + 2, Wb + rb2 * FB_STAGE);
            __builtin_amdgcn_sched_barrier(0);
            float v[8];
            bf16x8 n0 = q0, n1 = q1, n2 = q2;
            auto slice = [&](int k) {
                if (j < 3) {
                    const int hh = k >> 1, e0 = 2 * (k & 1);
                    v[2 * k] = accH[2 * (j + 1) + hh][e0]; v[2 * k + 1] = accH[2 * (j + 1) + hh][e0 + 1];
                    split_pair(v, k, n0, n1, n2);
                }
            };
            run_tiles(accY, Wb + rb * FB_STAGE, Wb + rb1 * FB_STAGE, slice);
            q0 = n0; q1 = n1; q2 = n2;
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
            rb = rb1;
        }
        if (rlive) {
#pragma unroll
            for (int ot = 0; ot < 8; ++ot) *reinterpret_cast<f32x4 *>(ft_out + mrow * kD + ot * 16 + 4 * lq) = accY[ot];
        }
    }
}

// decision layer (models.py:63,69) for out_dim = 1: y[m] = h[m,:] . w + b ; 32 lanes per row
__global__ void decision_kernel(const float *__restrict__ h, const float *__restrict__ w, const float *__restrict__ bias,
                                float *__restrict__ y, long M) {
    const int sub = threadIdx.x & 31;
    const long rows_per_block = blockDim.x / 32;
    for (long m = blockIdx.x * rows_per_block + threadIdx.x / 32; m < M; m += (long)gridDim.x * rows_per_block) {
        f32x4 x = *reinterpret_cast<const f32x4 *>(h + m * kD + sub * 4);
        f32x4 ww = *reinterpret_cast<const f32x4 *>(w + sub * 4);
        float a = x[0] * ww[0];
        a = fmaf(x[1], ww[1], a); a = fmaf(x[2], ww[2], a); a = fmaf(x[3], ww[3], a);
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) a += __shfl_xor(a, o, 32);
        if (sub == 0) y[m] = a + bias[0];
    }
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
static int grid_for(long total, int block, int cap = 256 * 16) {
    long g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

static size_t gat_rows_lds_bytes_hs(int n, int hs) {
    size_t ns = (size_t)n - 1;
    return ns * (size_t)(hs * kF + 16) * 4 + 4 * ns * hs * 4 + (size_t)hs * 4 * 4 + ns * 4 + 16;
}
// heads per workgroup: all 8 while at least two such workgroups fit a CU, else the head-split form (4).  Measured per
// launch (profiles/r02_ab_gat_heads.log): TSP200 x 256 11.5 -> 8.5 ms with the split (one -> two workgroups per CU);
// TSP100 x 1024 5.33 -> 5.69 ms, TSP50 x 2048 1.73 -> 1.82 ms (two+ workgroups already hide the prologue; the split
// only adds workgroup starts), so it is used where the unsplit tile leaves a CU with a single workgroup.
static int gat_rows_heads(int n) { return gat_rows_lds_bytes_hs(n, kH) * 2 <= (size_t)160 * 1024 ? kH : 4; }
size_t gat_rows_lds_bytes(int n) { return gat_rows_lds_bytes_hs(n, gat_rows_heads(n)); }

hipError_t launch_pack_features(const double *D, int B, int n, double scale, double minv, float *feat, hipStream_t st) {
    long total = (long)B * (n * (n - 1) / 2);
    (void)hipGetLastError();
    hipLaunchKernelGGL(pack_features_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, D, B, n, scale, minv, feat);
    return hipGetLastError();
}

hipError_t launch_unpack_regret(const float *y, int B, int n, double scale, double minv, double *out, hipStream_t st) {
    long total = (long)B * n * n;
    (void)hipGetLastError();
    hipLaunchKernelGGL(unpack_regret_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, y, B, n, scale, minv, out);
    return hipGetLastError();
}

size_t embed_fc_bytes() { return ((size_t)(kEmbedFcMaxIn + 1) * kD + 4 * kH) * sizeof(float); }
int embed_fc_max_in_dim() { return kEmbedFcMaxIn; }
// A [in_dim,128], b' [128] (and for in_dim == 1 the logit coefficients al | bl | ar | br [8] each) at `image` (embed_fc_bytes() of
// device memory), from the embedding and the first layer's fc and attention weights
hipError_t launch_embed_fc_prepare(const float *We, const float *be, const float *Wfc, const float *attn_l, const float *attn_r,
                                   int in_dim, void *image, hipStream_t st) {
    float *A = (float *)image;
    (void)hipGetLastError();
    hipLaunchKernelGGL(embed_fc_prepare_kernel, dim3(1), dim3(kD), 0, st, We, be, Wfc, attn_l, attn_r, in_dim, A, A + (size_t)in_dim * kD);
    return hipGetLastError();
}
// ft == nullptr: only h is written (in_dim == 1 with the rank-1 first GATConv)
hipError_t launch_embed_fc(const float *x, const float *W, const float *b, const void *image, float *h, float *ft, long M, int in_dim,
                           hipStream_t st) {
    const float *A = (const float *)image;
    (void)hipGetLastError();
    hipLaunchKernelGGL(embed_fc_kernel, dim3(grid_for(M * 32, 256)), dim3(256), 0, st, x, W, b, A, A + (size_t)in_dim * kD, h, ft, M, in_dim);
    return hipGetLastError();
}
// the first GATConv from the one input feature (in_dim == 1): same partials as launch_gat_rows on ft = x A + b'
hipError_t launch_gat_rows_rank1(const float *x, const void *image, int B, int n, float *part, float *part_ms, hipStream_t st, bool compact) {
    const float *A = (const float *)image;
    const int units = ((n - 1 + 63) / 64) * 2;                 // (64 destinations, 4 heads) per wavefront
    const size_t lds = (size_t)(n - 1) * (2 + 2 * kH) * sizeof(float) + kH * 4 * sizeof(float) + 16;
    (void)hipGetLastError();
    if (compact) hipLaunchKernelGGL(gat_rows_rank1_kernel<true>, dim3((unsigned)(B * n)), dim3(64 * (units < 4 ? 4 : units)), lds, st, x, A, n, part, part_ms);
    else hipLaunchKernelGGL(gat_rows_rank1_kernel<false>, dim3((unsigned)(B * n)), dim3(64 * (units < 4 ? 4 : units)), lds, st, x, A, n, part, part_ms);
    return hipGetLastError();
}

hipError_t launch_embed(const float *x, const float *W, const float *b, float *h, long M, int in_dim, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(embed_kernel, dim3(grid_for(M * 32, 256)), dim3(256), 0, st, x, W, b, h, M, in_dim);
    return hipGetLastError();
}

hipError_t launch_gemm_wkn(int epi, const float *A, const float *W, float *C, long M, int N, int K, const float *aux,
                           hipStream_t st) {
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)(N / BN));
    (void)hipGetLastError();
    if (epi == EPI_MASK) {
        hipLaunchKernelGGL((gemm_f32_kernel<EPI_MASK, true>), grid, dim3(256), 0, st, A, W, C, M, N, K, nullptr, aux, nullptr, nullptr);
    } else if (epi == EPI_ADD) {
        hipLaunchKernelGGL((gemm_f32_kernel<EPI_ADD, true>), grid, dim3(256), 0, st, A, W, C, M, N, K, nullptr, aux, nullptr, nullptr);
    } else {
        hipLaunchKernelGGL((gemm_f32_kernel<EPI_STORE, true>), grid, dim3(256), 0, st, A, W, C, M, N, K, nullptr, aux, nullptr, nullptr);
    }
    return hipGetLastError();
}

hipError_t launch_gemm(int epi, const float *A, const float *W, float *C, long M, int N, int K, const float *bias,
                       const float *skip, const float *bn_scale, const float *bn_shift, hipStream_t st) {
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)(N / BN));
    (void)hipGetLastError();
    if (epi == EPI_STORE) {
        hipLaunchKernelGGL(gemm_f32_kernel<EPI_STORE>, grid, dim3(256), 0, st, A, W, C, M, N, K, bias, skip, bn_scale, bn_shift);
    } else if (epi == EPI_BIAS_RELU) {
        hipLaunchKernelGGL(gemm_f32_kernel<EPI_BIAS_RELU>, grid, dim3(256), 0, st, A, W, C, M, N, K, bias, skip, bn_scale, bn_shift);
    } else {
        hipLaunchKernelGGL(gemm_f32_kernel<EPI_BIAS_SKIP_BN>, grid, dim3(256), 0, st, A, W, C, M, N, K, bias, skip, bn_scale, bn_shift);
    }
    return hipGetLastError();
}

template <int HS>
static hipError_t launch_gat_rows_hs(const float *ft, const float *attn_l, const float *attn_r, int B, int n, float *part,
                                     float *part_ms, hipStream_t st) {
    const size_t lds = gat_rows_lds_bytes_hs(n, HS);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gat_rows_kernel<HS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    (void)hipGetLastError();
    // units = (16-destination tiles) x (head quads of the workgroup), spread over 4..8 waves: the wave count with the
    // fewest idle wave slots; on ties the one that brings the CU closest to 16 resident waves at the LDS-limited
    // workgroup count
    const int units = ((n - 1 + 15) / 16) * (HS / 4);
    const int wgs_per_cu = (int)((size_t)160 * 1024 / lds) > 0 ? (int)((size_t)160 * 1024 / lds) : 1;
    const int want = 16 / wgs_per_cu > 0 ? 16 / wgs_per_cu : 1;
    int waves = 4;
    for (int w = 5; w <= 8; ++w) {
        const int idle_w = (units + w - 1) / w * w - units, idle_b = (units + waves - 1) / waves * waves - units;
        const int dw = w > want ? w - want : want - w, db = waves > want ? waves - want : want - waves;
        if (idle_w < idle_b || (idle_w == idle_b && dw < db)) waves = w;
    }
    hipLaunchKernelGGL(gat_rows_kernel<HS>, dim3((unsigned)(B * n * (kH / HS))), dim3(64 * waves), lds, st, ft, attn_l, attn_r,
                       n, part, part_ms);
    return hipGetLastError();
}

hipError_t launch_gat_rows(const float *ft, const float *attn_l, const float *attn_r, int B, int n, float *part,
                           float *part_ms, hipStream_t st) {
    static const char *force = getenv("GNNGLS_GAT_HEADS");      // experiments: 8 or 4
    const int hs = force ? atoi(force) : gat_rows_heads(n);
    return hs == 4 ? launch_gat_rows_hs<4>(ft, attn_l, attn_r, B, n, part, part_ms, st)
                   : launch_gat_rows_hs<kH>(ft, attn_l, attn_r, B, n, part, part_ms, st);
}

template <int MODE>
static hipError_t launch_ffn_mode(const float *part, const float *part_ms, const float *hin, const float *bn1_s,
                                  const float *bn1_b, const float *W1, const float *b1, const float *W2, const float *b2,
                                  const float *bn2_s, const float *bn2_b, float *hout, long M, const float *hid_in,
                                  float *hid_out, hipStream_t st) {
    const size_t lds = (size_t)(FT_M * LDX + 2 * 128 * LDW + 896) * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ffn_fused_kernel<MODE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    (void)hipGetLastError();
    hipLaunchKernelGGL(ffn_fused_kernel<MODE>, dim3((unsigned)((M + FT_M - 1) / FT_M)), dim3(256), lds, st, part, part_ms, hin,
                       bn1_s, bn1_b, W1, b1, W2, b2, bn2_s, bn2_b, hout, M, hid_in, hid_out);
    return hipGetLastError();
}

size_t ffn_packed_bytes() { return kFfnPackedBytes; }

// One layer's feed-forward weights (and, if given, the NEXT layer's fc) split into bf16 pieces in fragment order: ffn_packed_bytes()
// of device memory.  The image depends on the weights only -- gnngls_regret_prepare builds it once per weight image for all layers.
hipError_t launch_ffn_pack(const float *W1, const float *W2, const float *fc_next, void *packed, hipStream_t st) {
    unsigned char *pk = (unsigned char *)packed, *pk_fc = pk + (size_t)2 * 16 * FB_STAGE;
    (void)hipGetLastError();
    hipLaunchKernelGGL(ffn_pack_bf16x3_kernel, dim3(64), dim3(256), 0, st, W1, W2, pk);
    if (fc_next) hipLaunchKernelGGL(ffn_pack_fc_bf16x3_kernel, dim3(8), dim3(256), 0, st, fc_next, pk_fc);
    return hipGetLastError();
}

// packed != nullptr (a layer image made by launch_ffn_pack): the bf16x3 kernel; else the fp32 kernel.  With the bf16x3 kernel and
// has_fc_next the NEXT layer's ft = fc_next(hout) [M,128] is written to ft_out by the same launch (models.py:23 of layer l + 1)
hipError_t launch_ffn_fused(const float *part, const float *part_ms, const float *hin, const float *bn1_s,
                            const float *bn1_b, const float *W1, const float *b1, const float *W2, const float *b2,
                            const float *bn2_s, const float *bn2_b, float *hout, long M, const void *packed, bool has_fc_next,
                            float *ft_out, hipStream_t st, const float *dec_w, const float *dec_b, float *y_out,
                            const float *lr_img, const float *emb_w, const float *emb_b) {
    if (packed) {
        const size_t xt = (size_t)FB_M * LDX * sizeof(float), ring = (size_t)3 * FB_STAGE;
        const size_t lds = (kFbAlias ? (xt > ring ? xt : ring) : xt + ring) + 896 * sizeof(float);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ffn_fused_bf16x3_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(ffn_fused_bf16x3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        (void)hipGetLastError();
        const unsigned char *pk = (const unsigned char *)packed, *pk_fc = pk + (size_t)2 * 16 * FB_STAGE;
        if (lr_img)        // (`part` = the compact partials of gat_rows_rank1_kernel<true>, `hin` = the [M] input features)
            hipLaunchKernelGGL(ffn_fused_bf16x3_kernel<true>, dim3((unsigned)((M + FB_M - 1) / FB_M)), dim3(FB_T), lds, st, part, part_ms, hin,
                               bn1_s, bn1_b, pk, b1, b2, bn2_s, bn2_b, hout, M, has_fc_next ? pk_fc : nullptr, ft_out,
                               dec_w, dec_b, y_out, lr_img, emb_w, emb_b);
        else
            hipLaunchKernelGGL(ffn_fused_bf16x3_kernel<false>, dim3((unsigned)((M + FB_M - 1) / FB_M)), dim3(FB_T), lds, st, part, part_ms, hin,
                               bn1_s, bn1_b, pk, b1, b2, bn2_s, bn2_b, hout, M, has_fc_next ? pk_fc : nullptr, ft_out,
                               dec_w, dec_b, y_out, nullptr, nullptr, nullptr);
        return hipGetLastError();
    }
    return launch_ffn_mode<FFN_INFER>(part, part_ms, hin, bn1_s, bn1_b, W1, b1, W2, b2, bn2_s, bn2_b, hout, M, nullptr,
                                      nullptr, st);
}

// training forward: h3 = x + W2*ReLU(W1*x + b1) + b2 with x = h1*bn1_s + bn1_b, hidden activations kept in `hidden`
// [M,512]; `bn2_s`/`bn2_b` are passed as ones/zeros by the caller (BatchNorm 2 needs the batch statistics of h3 first)
hipError_t launch_ffn_fused_train(const float *h1, const float *bn1_s, const float *bn1_b, const float *W1, const float *b1,
                                  const float *W2, const float *b2, const float *ones, const float *zeros, float *h3,
                                  float *hidden, long M, hipStream_t st) {
    return launch_ffn_mode<FFN_TRAIN_FWD>(nullptr, nullptr, h1, bn1_s, bn1_b, W1, b1, W2, b2, ones, zeros, h3, M, nullptr,
                                          hidden, st);
}

// data gradient of the block: dx = ((dh3 * W2) . [hidden > 0]) * W1 + dh3; `hidden` [M,512] holds the saved activations
// on entry and the masked hidden gradient on exit; W2T [512,128] / W1T [128,512] are the transposed weights;
// zeros must hold 512 floats, ones 128
hipError_t launch_ffn_fused_bwd(const float *dh3, const float *W2T, const float *W1T, const float *ones, const float *zeros,
                                float *dx, float *hidden, long M, hipStream_t st) {
    return launch_ffn_mode<FFN_BWD>(nullptr, nullptr, dh3, ones, zeros, W2T, zeros, W1T, zeros, ones, zeros, dx, M, hidden,
                                    hidden, st);
}

// The two weight transposes the FFN backward needs, one launch: z = 0: dst0[512,128] = W2[128,512]^T, z = 1:
// dst1[128,512] = W1[512,128]^T (64K elements each; both are 16 x 4 grids of 32 x 32 tiles)
__global__ void transpose_pair_kernel(const float *__restrict__ W2, const float *__restrict__ W1, float *__restrict__ W2T,
                                      float *__restrict__ W1T) {
    __shared__ float tile[32][33];
    const bool second = blockIdx.z != 0;
    const float *src = second ? W1 : W2;
    float *dst = second ? W1T : W2T;
    const int R = second ? 512 : 128, C = second ? 128 : 512;
    const int tiles_c = C / 32, tile_id = blockIdx.x;                // 64 tiles per matrix
    const int c0 = (tile_id % tiles_c) * 32, r0 = (tile_id / tiles_c) * 32;
    for (int i = threadIdx.y; i < 32; i += 8) tile[i][threadIdx.x] = src[(long)(r0 + i) * C + c0 + threadIdx.x];
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) dst[(long)(c0 + i) * R + r0 + threadIdx.x] = tile[threadIdx.x][i];
}

hipError_t launch_transpose_pair(const float *W2, const float *W1, float *W2T, float *W1T, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(transpose_pair_kernel, dim3(64, 1, 2), dim3(32, 8), 0, st, W2, W1, W2T, W1T);
    return hipGetLastError();
}

hipError_t launch_decision(const float *h, const float *w, const float *b, float *y, long M, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(decision_kernel, dim3(grid_for(M, 8)), dim3(256), 0, st, h, w, b, y, M);
    return hipGetLastError();
}

}  // namespace gnngls
