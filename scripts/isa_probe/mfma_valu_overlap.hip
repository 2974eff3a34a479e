// Do vector-ALU instructions overlap with MFMAs on a SIMD?  (what bounds gat_rows and the feed-forward block)
// One wavefront per SIMD (256 threads per workgroup, one workgroup per CU), REP iterations of a loop whose body is
//   (a) M MFMAs on independent accumulators, (b) V independent v_fma_f32, (c) both interleaved (V / M vector instructions behind each
//   MFMA), for the fp32 MFMA (v_mfma_f32_16x16x4_f32: gat_rows) and the bf16 MFMA (v_mfma_f32_16x16x32_bf16: the feed-forward block).
// Prints shader cycles per iteration (s_memtime around the loop, wavefront 0 of workgroup 0; every CU runs the same loop).
//   hipcc --offload-arch=gfx950 -O2 scripts/isa_probe/mfma_valu_overlap.hip -o /tmp/mfma_valu_overlap && /tmp/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

constexpr int REP = 2048, M = 8;

// MODE bit 0: MFMAs, bit 1: vector instructions; BF16: which MFMA; VPM: vector instructions per MFMA
template <int MODE, bool BF16, int VPM>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long *out, float *sink, float seed) {
    f32x4 acc[M];
    float v[VPM > 0 ? VPM : 1];
    const float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f;
    bf16x8 pa, pb;
#pragma unroll
    for (int e = 0; e < 8; ++e) { pa[e] = (__bf16)(a + e); pb[e] = (__bf16)(b + e); }
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m] = f32x4{a, b, a, b};
#pragma unroll
    for (int k = 0; k < VPM; ++k) v[k] = a + k;
    const unsigned long long t0 = now();
#pragma unroll 1
    for (int it = 0; it < REP; ++it) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            if (MODE & 1) {
                if (BF16) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, pb, acc[m], 0, 0, 0);
                else acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
            }
            if (MODE & 2) {
#pragma unroll
                for (int k = 0; k < VPM; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = now();
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < M; ++m) s += acc[m][0] + acc[m][3];
#pragma unroll
    for (int k = 0; k < VPM; ++k) s += v[k];
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE, bool BF16, int VPM>
static double run(unsigned long long *out, float *sink) {
    hipLaunchKernelGGL((probe<MODE, BF16, VPM>), dim3(256), dim3(256), 0, 0, out, sink, 1.25f);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((probe<MODE, BF16, VPM>), dim3(256), dim3(256), 0, 0, out, sink, 1.25f);
    hipDeviceSynchronize();
    unsigned long long c = 0;
    hipMemcpy(&c, out, sizeof c, hipMemcpyDeviceToHost);
    return (double)c / REP;
}

template <bool BF16, int VPM>
static void line(const char *name, unsigned long long *out, float *sink) {
    const double m = run<1, BF16, VPM>(out, sink), v = run<2, BF16, VPM>(out, sink), both = run<3, BF16, VPM>(out, sink);
    printf("%-28s %d MFMAs alone %7.1f   %3d v_fma_f32 alone %7.1f   interleaved %7.1f   (sum %7.1f, max %7.1f)\n", name, M, m, M * VPM, v,
           both, m + v, m > v ? m : v);
}

int main() {
    unsigned long long *out;
    float *sink;
    hipMalloc(&out, 64);
    hipMalloc(&sink, 64);
    printf("shader cycles per loop iteration, one wavefront per SIMD, every CU busy\n");
    line<false, 2>("v_mfma_f32_16x16x4_f32, 2/MFMA", out, sink);
    line<false, 4>("v_mfma_f32_16x16x4_f32, 4/MFMA", out, sink);
    line<false, 6>("v_mfma_f32_16x16x4_f32, 6/MFMA", out, sink);
    line<false, 8>("v_mfma_f32_16x16x4_f32, 8/MFMA", out, sink);
    line<true, 1>("v_mfma_f32_16x16x32_bf16, 1/MFMA", out, sink);
    line<true, 2>("v_mfma_f32_16x16x32_bf16, 2/MFMA", out, sink);
    line<true, 3>("v_mfma_f32_16x16x32_bf16, 3/MFMA", out, sink);
    line<true, 4>("v_mfma_f32_16x16x32_bf16, 4/MFMA", out, sink);
    return 0;
}
