// Tiny standalone probe: semantics of the DPP wave shifts on gfx950 (which lane feeds which).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int *o) {
    int x = threadIdx.x;
    int y = __builtin_amdgcn_update_dpp(-1, x, 0x138, 0xf, 0xf, false);   // wave_shr:1
    int z = __builtin_amdgcn_update_dpp(-1, x, 0x130, 0xf, 0xf, false);   // wave_shl:1
    o[threadIdx.x] = y;
    o[64 + threadIdx.x] = z;
}
int main() {
    int *d, h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("wave_shr:1 lane0..3=%d %d %d %d lane31,32,63=%d %d %d\n", h[0], h[1], h[2], h[3], h[31], h[32], h[63]);
    printf("wave_shl:1 lane0..3=%d %d %d %d lane31,32,62,63=%d %d %d %d\n", h[64], h[65], h[66], h[67], h[95], h[96], h[126], h[127]);
    return 0;
}
