// model_kernels.h -- internal interface between the GNN forward kernels (model_kernels.hip) and capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gnngls {

enum { GEMM_EPI_STORE = 0, GEMM_EPI_BIAS_RELU = 1, GEMM_EPI_BIAS_SKIP_BN = 2, GEMM_EPI_MASK = 3, GEMM_EPI_ADD = 4 };

size_t gat_rows_lds_bytes(int n);
hipError_t launch_pack_features(const double *D, int B, int n, double scale, double minv, float *feat, hipStream_t st);
hipError_t launch_unpack_regret(const float *y, int B, int n, double scale, double minv, double *out, hipStream_t st);
hipError_t launch_embed(const float *x, const float *W, const float *b, float *h, long M, int in_dim, hipStream_t st);
// embed + the first layer's fc as one rank-in_dim pass (in_dim <= embed_fc_max_in_dim()); `image`: embed_fc_bytes() of device memory
size_t embed_fc_bytes();
int embed_fc_max_in_dim();
hipError_t launch_embed_fc_prepare(const float *We, const float *be, const float *Wfc, const float *attn_l, const float *attn_r,
                                   int in_dim, void *image, hipStream_t st);
// in_dim == 1: the first GATConv from the one input feature and the same image (partials as launch_gat_rows on ft = x A + b')
// compact: [2][B N][24] = (shift, sum of weights, sum of weights x feature) per head into `part` (for launch_ffn_fused's lr_img form)
hipError_t launch_gat_rows_rank1(const float *x, const void *image, int B, int n, float *part, float *part_ms, hipStream_t st,
                                 bool compact = false);
hipError_t launch_embed_fc(const float *x, const float *W, const float *b, const void *image, float *h, float *ft, long M, int in_dim,
                           hipStream_t st);
hipError_t launch_gemm(int epi, const float *A, const float *W, float *C, long M, int N, int K, const float *bias,
                       const float *skip, const float *bn_scale, const float *bn_shift, hipStream_t st);
// C[M,N] = A[M,K] * W[K,N] (W row-major [K,N]); epi STORE / MASK (C = acc * (aux > 0)) / ADD (C = acc + aux)
hipError_t launch_gemm_wkn(int epi, const float *A, const float *W, float *C, long M, int N, int K, const float *aux,
                           hipStream_t st);
hipError_t launch_ffn_fused_train(const float *h1, const float *bn1_s, const float *bn1_b, const float *W1, const float *b1,
                                  const float *W2, const float *b2, const float *ones, const float *zeros, float *h3,
                                  float *hidden, long M, hipStream_t st);
hipError_t launch_ffn_fused_bwd(const float *dh3, const float *W2T, const float *W1T, const float *ones, const float *zeros,
                                float *dx, float *hidden, long M, hipStream_t st);
hipError_t launch_transpose_pair(const float *W2, const float *W1, float *W2T, float *W1T, hipStream_t st);
hipError_t launch_gat_rows(const float *ft, const float *attn_l, const float *attn_r, int B, int n, float *part,
                           float *part_ms, hipStream_t st);
hipError_t launch_ffn_fused(const float *part, const float *part_ms, const float *hin, const float *bn1_s,
                            const float *bn1_b, const float *W1, const float *b1, const float *W2, const float *b2,
                            const float *bn2_s, const float *bn2_b, float *hout, long M, const void *packed, bool has_fc_next,
                            float *ft_out, hipStream_t st,
                            // (bf16x3 form only) the decision layer folded into the epilogue: y_out[m] = hout[m,:] . dec_w + dec_b[0], hout not stored
                            const float *dec_w = nullptr, const float *dec_b = nullptr, float *y_out = nullptr,
                            // (bf16x3 form, first layer of a one-feature model) lr_img = the embed-fc image: `part` then holds
                            // launch_gat_rows_rank1's compact partials and `hin` the [M] input features; emb_w / emb_b = the embedding
                            const float *lr_img = nullptr, const float *emb_w = nullptr, const float *emb_b = nullptr);
size_t ffn_packed_bytes();      // bytes of one layer's image for the bf16x3 form of launch_ffn_fused (split weights in fragment order)
hipError_t launch_ffn_pack(const float *W1, const float *W2, const float *fc_next, void *packed, hipStream_t st);
hipError_t launch_decision(const float *h, const float *w, const float *b, float *y, long M, hipStream_t st);

}  // namespace gnngls
