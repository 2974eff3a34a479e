// Where do the wavefronts of co-resident workgroups sit?  1024 workgroups of 256 threads with 40 KB of LDS each (the headline shape of
// gls_kernel: four workgroups per CU, four wavefronts each); every wavefront records HW_ID (SIMD, CU, SE), XCC_ID and the LDS base.
//   hipcc --offload-arch=gfx950 -O2 scripts/isa_probe/placement_probe.hip -o /tmp/placement_probe && /tmp/placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned *out, int spin) {
    extern __shared__ char smem[];
    unsigned hw, xcc, lds;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(lds));
    smem[threadIdx.x] = (char)spin;
    long long t0 = clock64();
    while (clock64() - t0 < spin) { }                      // keep the workgroup resident so that the CU fills up
    if ((threadIdx.x & 63) == 0) {
        unsigned *o = out + 4 * (blockIdx.x * 4 + (threadIdx.x >> 6));
        o[0] = hw; o[1] = xcc; o[2] = lds; o[3] = smem[threadIdx.x];
    }
}
int main() {
    const int B = 1024;
    unsigned *d; hipMalloc(&d, B * 4 * 4 * sizeof(unsigned));
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
    probe<<<B, 256, 40960>>>(d, 2000000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(B * 16);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // per (xcc, se, cu): which SIMD holds wavefront w of each resident workgroup
    std::map<unsigned, std::vector<int>> cu_blocks;
    int simd_of_wave[4][4] = {};
    int distinct_simds = 0;
    for (int b = 0; b < B; ++b) {
        unsigned seen = 0;
        for (int w = 0; w < 4; ++w) {
            const unsigned hw = h[(b * 4 + w) * 4];
            const int simd = (hw >> 4) & 3;
            simd_of_wave[w][simd]++;
            seen |= 1u << simd;
        }
        distinct_simds += __builtin_popcount(seen) == 4;
        const unsigned hw0 = h[b * 16], xcc = h[b * 16 + 1] & 0xf;
        const unsigned key = (xcc << 16) | (((hw0 >> 13) & 7) << 8) | ((hw0 >> 8) & 0xf);
        cu_blocks[key].push_back(b);
    }
    printf("workgroups whose four wavefronts sit on four different SIMDs: %d of %d\n", distinct_simds, B);
    for (int w = 0; w < 4; ++w)
        printf("wavefront %d of a workgroup: SIMD 0/1/2/3 = %d %d %d %d\n", w, simd_of_wave[w][0], simd_of_wave[w][1], simd_of_wave[w][2], simd_of_wave[w][3]);
    printf("CUs used: %zu\n", cu_blocks.size());
    int shown = 0, collide = 0, cus4 = 0;
    for (auto &kv : cu_blocks) {
        if (kv.second.size() == 4) {
            ++cus4;
            unsigned seen = 0;
            for (int b : kv.second) seen |= 1u << ((h[b * 16] >> 4) & 3);
            collide += __builtin_popcount(seen) < 4;
        }
        if (shown < 6) {
            printf("xcc %u se %u cu %u:", kv.first >> 16, (kv.first >> 8) & 0xff, kv.first & 0xff);
            for (int b : kv.second) {
                printf("  wg %d lds_base %u simd", b, h[b * 16 + 2] & 0xff);
                for (int w = 0; w < 4; ++w) printf(" %u", (h[(b * 4 + w) * 4] >> 4) & 3);
            }
            printf("\n");
            ++shown;
        }
    }
    printf("CUs with four workgroups: %d; of those with two or more wavefront-0s on one SIMD: %d\n", cus4, collide);
    return 0;
}
