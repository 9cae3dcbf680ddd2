// Which SIMD does wave i of a 512-thread workgroup land on?  (gfx950; HW_REG_HW_ID bits [5:4] = SIMD_ID, [3:0] = WAVE_ID)
// hipcc --offload-arch=gfx950 -O3 tools/micro/wave_simd.hip -o variants/wave_simd && variants/wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* out) {
    extern __shared__ float lds[];
    unsigned id = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
    lds[threadIdx.x] = id;
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 8 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    k<<<256, 512, 150 * 1024>>>(d);
    unsigned h[256 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 0;
    for (int b = 0; b < 256; ++b) {
        bool good = true;
        for (int w = 0; w < 4; ++w) good &= ((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + w + 4] >> 4) & 3);
        ok += good;
        if (b < 6) { printf("wg %d: simd of waves 0..7:", b); for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3); printf("  cu %u\n", (h[b*8] >> 8) & 15); }
    }
    printf("%d of 256 workgroups have wave w and wave w+4 on the same SIMD\n", ok);
    return 0;
}
