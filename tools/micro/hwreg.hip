// hwreg.hip - what HW_REG_HW_ID / HW_REG_LDS_ALLOC / XCC_ID hold for the workgroups of a 1024-block launch with 40 KB of LDS each (4 per
// CU): which linear block ids share a CU, and whether the LDS allocation base tells co-resident workgroups apart.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
    __shared__ float lds[10240];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float v = lds[(threadIdx.x * 7) & 255];
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;   // keep the block resident long enough for the whole grid to be placed
    if (v == 12345.f) out[0] = 1;
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        unsigned* o = out + ((size_t)blockIdx.x * 4 + w) * 4;
        o[0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_ID, 32 bits
        o[1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 6);    // LDS_ALLOC
        o[2] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);   // XCC_ID
        o[3] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 5);    // GPR_ALLOC
    }
}
int main() {
    const int nb = 1024;
    unsigned* d;
    hipMalloc(&d, nb * 16 * 4);
    hipMemset(d, 0, nb * 16 * 4);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 16);
    hipMemcpy(h.data(), d, nb * 16 * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < nb; ++b) {
        const unsigned hw = h[b * 16], lds = h[b * 16 + 1], xcc = h[b * 16 + 2] & 0xf;
        const unsigned key = (xcc << 16) | (hw & 0xff00) | ((hw >> 12) & 0xf) << 4;   // xcc, cu/sh/se bits
        cu[(xcc << 20) | ((hw >> 8) & 0xfff)].push_back(b);
        if (b < 24 || (b & 255) == 0)
            printf("block %4d: waves' HW_ID %08x %08x %08x %08x  LDS_ALLOC %08x %08x  XCC %x GPR_ALLOC %08x\n", b, hw, h[b * 16 + 4], h[b * 16 + 8],
                   h[b * 16 + 12], lds, h[b * 16 + 5], xcc, h[b * 16 + 3]);
    }
    printf("%zu distinct (xcc, cu/sh/se) keys\n", cu.size());
    int n = 0;
    for (auto& kv : cu) {
        if (n++ >= 12) break;
        printf("key %06x:", kv.first);
        for (int b : kv.second) printf(" %d(lds %x, wave0 slot %x simd %x)", b, h[b * 16 + 1] & 0xfff, h[b * 16] & 0xf, (h[b * 16] >> 4) & 3);
        printf("\n");
    }
    return 0;
}
