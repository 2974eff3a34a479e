// Micro-probe for the dependent-chain table of gls_kernel's perturbation step (profiles/r05_isa/): cycles per instruction of the
// instruction kinds on that chain, measured with s_memtime on one wavefront, (a) alone on its SIMD and (b) with three more
// wavefronts per SIMD running the same loop (the residency of the TSP100 x 1024 headline: four workgroups of four per CU).
//   hipcc --offload-arch=gfx950 -O2 scripts/isa_probe/latency_probe.hip -o /tmp/latency_probe && /tmp/latency_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define REP 256
#define STR2(x) #x
#define STR(x) STR2(x)

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

enum {
    K_EMPTY, K_VADD_DEP, K_VADD_IND, K_VMUL24_DEP, K_VMULLO_DEP, K_DADD_DEP, K_DADD_IND, K_DMUL_DEP, K_DFMA_DEP, K_CVT_DEP, K_RCP_DEP,
    K_DIV_DEP, K_DIV2_IND, K_LDS_U8_CHASE, K_LDS_B64_CHASE, K_BUF_L1_CHASE, K_BUF_L2_CHASE, K_BUF_MALL_CHASE, K_READLANE_RT, K_DPP_MIN,
    K_EXEC_BRANCH, K_SALU_DEP, K_SMUL_DEP, K_CNDMASK_IND, K_CMP_SAVEEXEC, K_SBRANCH_TAKEN, K_CNDMASK_SGPR_IND, K_CNDMASK_DEP, K_VCMP_SGPR_IND, K_READLANE_IND, K_READFIRST_IND, K_SCSELECT_DEP,
    K_SBRANCH_NOT_TAKEN, K_WAITCNT_IDLE, K_SNOP, K_VMAX_DEP, K_LSHLADD_DEP, K_MAD24_DEP, K_VMOV_SGPR_IND, K_DSREAD_IND, K_BUFLOAD_IND, K_BFI_IND, K_COUNT
};
static const char *kNames[K_COUNT] = {
    "empty loop (overhead, per iteration)", "v_add_u32 dependent", "v_add_u32 independent (issue)", "v_mul_i32_i24 dependent",
    "v_mul_lo_u32 dependent", "v_add_f64 dependent", "v_add_f64 independent (issue)", "v_mul_f64 dependent", "v_fma_f64 dependent",
    "v_cvt_f64_i32 + v_cvt_i32_f64 dependent pair", "v_rcp_f64 dependent", "fp64 division (full sequence) dependent",
    "two independent fp64 divisions (per pair)", "ds_read_u8 pointer chase (address <- data)", "ds_read_b64 pointer chase",
    "buffer_load_dword chase, 2 KB (L1 hit)", "buffer_load_dword chase, 2 MB (L2 hit)", "buffer_load_dword chase, 96 MB (MALL / HBM)",
    "v_readlane -> s -> v_mov round trip", "v_min_u32_dpp step (s_nop 1 + dpp)", "v_cmp + s_and_saveexec + s_cbranch_execz + s_or exec",
    "s_add_i32 dependent", "s_mul_i32 dependent", "v_cndmask_b32 independent (issue)", "v_cmp_f64 + s_and + v_cndmask (select chain)",
    "s_cmp + taken s_cbranch_scc", "v_cndmask_b32 e64 (sgpr pair) independent", "v_cndmask_b32 dependent", "v_cmp_lt_i32 e64 -> sgpr independent",
    "v_readlane_b32 independent", "v_readfirstlane_b32 independent", "s_cmp + s_cselect dependent pair", "s_cmp + NOT taken s_cbranch_scc", "s_waitcnt (nothing outstanding)",
    "s_nop 0", "v_max_i32 dependent", "v_lshl_add_u32 dependent", "v_mad_u32_u24 dependent", "v_mov_b32 v, s independent", "ds_read_b64 independent (8 per wait)", "buffer_load_dword L1 independent (8 per wait)", "v_bfi_b32 independent"};

__global__ __launch_bounds__(256, 4) void probe(int kind, int *gbuf, unsigned gmask, unsigned long long *out, int *sink) {
    __shared__ unsigned char lds8[4096];
    __shared__ double lds64[512];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += blockDim.x) lds8[i] = (unsigned char)((i * 37 + 11) & 255);
    for (int i = tid; i < 512; i += blockDim.x) lds64[i] = __longlong_as_double((long long)(((i * 53 + 7) & 511) * 8));
    __syncthreads();
    int v = lane + 1, w = lane * 3 + 1;
    double d = 1.0 + lane * 1e-3, e = 1.0000001;
    int s = 3;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)gbuf, 0, (int)((gmask + 1u) * 4u), 0x00020000);
    const unsigned long long t0 = now();
#define LOOP(body)                      \
    _Pragma("unroll 1") for (int it = 0; it < REP / 16; ++it) { \
        body body body body body body body body body body body body body body body body }
    switch (kind) {
        case K_EMPTY: LOOP(asm volatile("" ::: "memory");) break;
        case K_VADD_DEP: LOOP(asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(w));) break;
        case K_VADD_IND: { int a0 = v, a1 = v, a2 = v, a3 = v;
            LOOP(asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w));)
            v = a0 + a1 + a2 + a3; } break;
        case K_VMUL24_DEP: LOOP(asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(v) : "v"(w));) break;
        case K_VMULLO_DEP: LOOP(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v) : "v"(w));) break;
        case K_DADD_DEP: LOOP(asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(e));) break;
        case K_DADD_IND: { double a0 = d, a1 = d, a2 = d, a3 = d;
            LOOP(asm volatile("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(e));)
            d = a0 + a1 + a2 + a3; } break;
        case K_DMUL_DEP: LOOP(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"(e));) break;
        case K_DFMA_DEP: LOOP(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d) : "v"(e));) break;
        case K_CVT_DEP: LOOP(asm volatile("v_cvt_f64_i32 %0, %1\n\tv_cvt_i32_f64 %1, %0" : "+v"(d), "+v"(v));) break;
        case K_RCP_DEP: LOOP(asm volatile("v_rcp_f64 %0, %0" : "+v"(d));) break;
        case K_DIV_DEP: LOOP(d = e / d; asm volatile("" : "+v"(d));) break;
        case K_DIV2_IND: { double a0 = d, a1 = d + 0.5;
            LOOP(a0 = e / a0; a1 = e / a1; asm volatile("" : "+v"(a0), "+v"(a1));)
            d = a0 + a1; } break;
        case K_LDS_U8_CHASE: LOOP(v = lds8[(v * 16 + lane) & 4095]; asm volatile("" : "+v"(v));) break;
        case K_LDS_B64_CHASE: { int a = lane * 8;
            LOOP({ const double x = *(const double *)((const char *)lds64 + (a & 4088)); a = (int)__double_as_longlong(x); asm volatile("" : "+v"(a)); })
            v = a; } break;
        case K_BUF_L1_CHASE: case K_BUF_L2_CHASE: case K_BUF_MALL_CHASE: {
            int a = (lane * 64) & gmask;
            LOOP({ a = __builtin_amdgcn_raw_buffer_load_b32(rs, (a & gmask) << 2, 0, 0); asm volatile("" : "+v"(a)); })
            v = a; } break;
        case K_READLANE_RT: LOOP(asm volatile("v_readlane_b32 %1, %0, 5\n\ts_nop 0\n\tv_mov_b32 %0, %1" : "+v"(v), "+s"(s));) break;
        case K_DPP_MIN: LOOP(asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v));) break;
        case K_EXEC_BRANCH: LOOP({ if (v > -5) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(w)); } asm volatile("" : "+v"(v)); }) break;
        case K_SALU_DEP: LOOP(asm volatile("s_add_i32 %0, %0, 7" : "+s"(s) : : "scc");) break;
        case K_SMUL_DEP: LOOP(asm volatile("s_mul_i32 %0, %0, 7" : "+s"(s));) break;
        case K_CNDMASK_IND: { int a0 = v, a1 = v, a2 = v, a3 = v;
            LOOP(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w) : "vcc");)
            v = a0 + a1 + a2 + a3; } break;
        case K_CMP_SAVEEXEC: LOOP({ const bool c = d < e; d = c ? d + e : d; asm volatile("" : "+v"(d)); }) break;
        case K_SBRANCH_TAKEN: LOOP(asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_add_i32 %0, %0, 1\n1:" : "+s"(s) : : "scc");) break;
        case K_CNDMASK_SGPR_IND: { int a0 = v, a1 = v, a2 = v, a3 = v; unsigned long long m = 0x5555555555555555ull;
            LOOP(asm volatile("v_cndmask_b32 %0, %0, %4, %5\n\tv_cndmask_b32 %1, %1, %4, %5\n\tv_cndmask_b32 %2, %2, %4, %5\n\tv_cndmask_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w), "s"(m));)
            v = a0 + a1 + a2 + a3; } break;
        case K_CNDMASK_DEP: LOOP(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v) : "v"(w) : "vcc");) break;
        case K_VCMP_SGPR_IND: { unsigned long long m0, m1, m2, m3;
            LOOP(asm volatile("v_cmp_lt_i32 %0, %4, %5\n\tv_cmp_lt_i32 %1, %4, %5\n\tv_cmp_lt_i32 %2, %4, %5\n\tv_cmp_lt_i32 %3, %4, %5" : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3) : "v"(v), "v"(w));)
            s += (int)(m0 + m1 + m2 + m3); } break;
        case K_READLANE_IND: { int s0, s1, s2, s3;
            LOOP(asm volatile("v_readlane_b32 %0, %4, 1\n\tv_readlane_b32 %1, %4, 2\n\tv_readlane_b32 %2, %4, 3\n\tv_readlane_b32 %3, %4, 4" : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(v));)
            s += s0 + s1 + s2 + s3; } break;
        case K_READFIRST_IND: { int s0, s1, s2, s3;
            LOOP(asm volatile("v_readfirstlane_b32 %0, %4\n\tv_readfirstlane_b32 %1, %4\n\tv_readfirstlane_b32 %2, %4\n\tv_readfirstlane_b32 %3, %4" : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(v));)
            s += s0 + s1 + s2 + s3; } break;
        case K_SCSELECT_DEP: LOOP(asm volatile("s_cmp_gt_i32 %0, 100\n\ts_cselect_b32 %0, %0, 7" : "+s"(s) : : "scc");) break;
        case K_SBRANCH_NOT_TAKEN: LOOP(asm volatile("s_cmp_eq_u32 %0, 0x7654321\n\ts_cbranch_scc1 1f\n\ts_add_i32 %0, %0, 1\n1:" : "+s"(s) : : "scc");) break;
        case K_WAITCNT_IDLE: LOOP(asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");) break;
        case K_SNOP: LOOP(asm volatile("s_nop 0");) break;
        case K_VMAX_DEP: LOOP(asm volatile("v_max_i32 %0, %0, %1" : "+v"(v) : "v"(w));) break;
        case K_LSHLADD_DEP: LOOP(asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(v) : "v"(w));) break;
        case K_MAD24_DEP: LOOP(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(v) : "v"(w));) break;
        case K_VMOV_SGPR_IND: { int a0, a1, a2, a3;
            LOOP(asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %4\n\tv_mov_b32 %3, %4" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "s"(s));)
            v += a0 + a1 + a2 + a3; } break;
        case K_DSREAD_IND: { const int a = (lane * 8) & 4088; double x0, x1, x2, x3;
            LOOP(asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:16\n\tds_read_b64 %3, %4 offset:24\n\ts_waitcnt lgkmcnt(0)" : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(a) : "memory");)
            d += x0 + x1 + x2 + x3; } break;
        case K_BUFLOAD_IND: { const int a = (lane * 4) & 2044; int x0, x1, x2, x3;
            LOOP(asm volatile("buffer_load_dword %0, %4, %5, 0 offen\n\tbuffer_load_dword %1, %4, %5, 0 offen offset:4\n\tbuffer_load_dword %2, %4, %5, 0 offen offset:8\n\tbuffer_load_dword %3, %4, %5, 0 offen offset:12\n\ts_waitcnt vmcnt(0)" : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(a), "s"(rs) : "memory");)
            v += x0 + x1 + x2 + x3; } break;
        case K_BFI_IND: { int a0 = v, a1 = v, a2 = v, a3 = v;
            LOOP(asm volatile("v_bfi_b32 %0, %4, %0, %5\n\tv_bfi_b32 %1, %4, %1, %5\n\tv_bfi_b32 %2, %4, %2, %5\n\tv_bfi_b32 %3, %4, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w), "v"(lane));)
            v = a0 + a1 + a2 + a3; } break;
    }
    const unsigned long long t1 = now();
    if (tid == 0 && blockIdx.x == 0) out[kind] = t1 - t0;
    if (v == 0x7fffffff || d == 123.456 || s == 0x7ffffff1) sink[0] = v + (int)d + s;     // keep the chains alive
}

int main() {
    const size_t big = 24u << 20;      // ints: 96 MB
    int *gbuf, *sink; unsigned long long *out;
    hipMalloc(&gbuf, big * 4); hipMalloc(&sink, 64); hipMalloc(&out, K_COUNT * 8);
    int *h = (int *)malloc(big * 4);
    unsigned x = 12345;
    for (size_t i = 0; i < big; ++i) { x = x * 1664525u + 1013904223u; h[i] = (int)((x >> 4) & 0x7fffffff); }
    hipMemcpy(gbuf, h, big * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        // mode 0: ONE workgroup of 256 threads (one wavefront per SIMD of one CU); mode 1: 1024 workgroups (four per CU, 4 waves per SIMD)
        const int grid = mode == 0 ? 1 : 1024;
        fflush(stdout);
        printf("=== %s\n", mode == 0 ? "one wavefront per SIMD (one workgroup on the device)" : "four wavefronts per SIMD, every CU busy with the same loop (1024 workgroups)");
        unsigned long long base = 0;
        for (int k = 0; k < K_COUNT; ++k) {
            const unsigned gmask = k == K_BUF_L1_CHASE ? 511u : k == K_BUF_L2_CHASE ? (512u * 1024u - 1u) : (16u * 1024u * 1024u - 1u);
            unsigned long long best = ~0ull, r;
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, k, gbuf, gmask, out, sink);
                hipDeviceSynchronize();
                hipMemcpy(&r, out + k, 8, hipMemcpyDeviceToHost);
                if (r < best) best = r;
            }
            if (k == K_EMPTY) base = best;
            const int per = (k == K_VADD_IND || k == K_DADD_IND || k == K_CNDMASK_IND || k == K_CNDMASK_SGPR_IND || k == K_VCMP_SGPR_IND || k == K_READLANE_IND || k == K_READFIRST_IND || k == K_VMOV_SGPR_IND || k == K_DSREAD_IND || k == K_BUFLOAD_IND || k == K_BFI_IND) ? 4 : 1;
            printf("%-58s %8.2f cycles per %s\n", kNames[k], (double)(best - (k == K_EMPTY ? 0 : base)) / REP / per + 0.0, per == 4 ? "instruction (4 independent streams)" : "step");
            fflush(stdout);
        }
    }
    return 0;
}
