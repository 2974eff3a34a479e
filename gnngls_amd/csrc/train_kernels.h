// train_kernels.h -- internal interface between the training-step kernels (train_kernels.hip) and capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gnngls {

enum { CS_SUM_SQ = 0, CS_SUM_PROD = 1, CS_ROWSCALE = 2, CS_HEADSCALE = 3 };
constexpr int kColsumMaxBlocks = 512;     // partial buffer: kColsumMaxBlocks * 2 * 512 doubles
constexpr int kGemmTnMaxChunks = 256;      // partial buffer: kGemmTnMaxChunks * 128 * 512 floats

int colsum_blocks(long M, int C);
hipError_t launch_colsum(int mode, const float *X, const float *Y, const float *Y2, long M, int C, int ystride,
                         double *partial, int *nblocks, hipStream_t st);
hipError_t launch_colsum_store(const double *partial, int nblocks, int C, int ostride, float *out0, float *out1,
                               hipStream_t st);
hipError_t launch_sum_vector(const float *v, long M, float *out, hipStream_t st);
hipError_t launch_bn_stats_finalize(const double *partial, int nblocks, long M, const float *gamma, const float *beta,
                                    float eps, float *scale, float *shift, float *mean, float *invstd, float *batch_mean,
                                    float *batch_var, hipStream_t st);
hipError_t launch_bn_bwd_finalize(const double *partial, int nblocks, long M, const float *gamma, const float *mean,
                                  const float *invstd, float *dgamma, float *dbeta, float *coef, hipStream_t st);
hipError_t launch_bn_bwd_apply(const float *dy, const float *x, const float *mean, const float *coef, float *dx, long M,
                               hipStream_t st);
hipError_t launch_bn_bwd_apply_and_affine(const float *dy, const float *h3, const float *mean, const float *coef, float *dx,
                                          const float *h1, const float *scale, const float *shift, float *xout, long M,
                                          hipStream_t st);
hipError_t launch_affine_cols(const float *x, const float *scale, const float *shift, float *out, long M, hipStream_t st);
hipError_t launch_outer_rows(const float *dy, const float *w, float *dh, long M, hipStream_t st);
hipError_t launch_gat_combine_train(const float *part, const float *part_ms, const float *h, long M, float *g, float *h1,
                                    float *att, hipStream_t st);
size_t gat_bwd_lds_bytes(int n);
int gat_bwd_max_nodes();
hipError_t launch_gat_bwd_rows(const float *ft, const float *dout, const float *gout, const float *att, const float *attn_l,
                               const float *attn_r, int B, int n, float *P, float *dlr, hipStream_t st);
hipError_t launch_gat_bwd_combine(const float *P, const float *dlr, const float *attn_l, const float *attn_r, long M,
                                  float *dft, float *dl, float *dr, hipStream_t st);
int gemm_tn_chunks(long M);
// out[N1,N2] = X[M,N1]^T * Y[M,N2] and, if xsum_out, xsum_out[N1] = column sums of X;
// `partial` holds gemm_tn_chunks(M) * (N1 * N2 + N1) floats
hipError_t launch_gemm_tn(const float *X, const float *Y, long M, int N1, int N2, float *partial, float *out,
                          float *xsum_out, hipStream_t st);

}  // namespace gnngls
