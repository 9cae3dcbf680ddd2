// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 streams shaped like the step kernel's products (gfx950).
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o gpurun_out/mfma_rate && gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int NACC, int MODE>
__global__ __launch_bounds__(512) void rate_kernel(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32 * 1024; i += 256) lds[i] = 0.001f * (i & 63);
    __syncthreads();
    f32x4 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a4[2][8];   // operands, double-buffered: the reads of group it+1 are requested before the products of group it
#pragma unroll
    for (int t = 0; t < 8; ++t) a4[0][t] = a4[1][t] = *(const f32x4*)&lds[(lane & 15) * 140 + 4 * (lane >> 4) + 16 * t * 140 % 8192];
    f32x4 b = f32x4{1.f + lane, 2.f, 3.f, 4.f};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (threadIdx.x >= 256) {   // second wave on every SIMD (512-thread launches only): VALU work only, same duration
        float w[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
        if (MODE == 20) {   // helper role: per group read 8 tiles from LDS, ~64 VALU instructions, write 8 tiles back
            float* h = lds + 16384 + ((threadIdx.x - 256) * 4) % 8192;
            for (int it = 0; it < iters; ++it) {
                f32x4 tl[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) tl[t] = *(const f32x4*)(h + 1024 * (t & 3));
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tl[t][r] = fmaf(fmaxf(tl[t][r], 0.f), 1.0001f, w[r]);
#pragma unroll
                for (int t = 0; t < 8; ++t) *(f32x4*)(h + 1024 * (t & 3) + 4096) = tl[t];
            }
            out[blockIdx.x * 512 + threadIdx.x] = w[0];
            return;
        }
        for (int it = 0; it < iters * (MODE == 0 ? 6 : 1); ++it) {
#pragma unroll
            for (int k = 0; k < 32; ++k) w[k & 7] = fmaf(w[k & 7], 1.0001f, 0.5f);
        }
        out[blockIdx.x * 512 + threadIdx.x] = w[0] + w[1] + w[2] + w[3] + w[4] + w[5] + w[6] + w[7];
        return;
    }
    float vf[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    auto body = [&](auto par, int it) {
        constexpr int cur = decltype(par)::value, nx = cur ^ 1;
        if (MODE == 1 || MODE == 3) {  // 8 x ds_read_b128 per 32 MFMAs, like the forward product
#pragma unroll
            for (int t = 0; t < 8; ++t) a4[nx][t] = *(const f32x4*)&lds[((lane & 15) * 140 + 4 * (lane >> 4) + 16 * ((it + t) & 31) + 2240 * t) & 8191];
        }
        if (MODE == 2) {  // 32 x ds_read_b32 per 32 MFMAs (column reads), like the backward product
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) a4[nx][t][r] = lds[((lane & 15) + 16 * t + (4 * (lane >> 4) + r + 16 * (it & 3)) * 140) & 16383];
        }
        if (MODE == 4) {  // the same bytes with 8 x ds_read_b128 (permuted unit order)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 q = *(const f32x4*)&lds[(4 * (lane & 15) + 64 * h + (4 * (lane >> 4) + r + 16 * (it & 3)) * 140) & 16383];
#pragma unroll
                    for (int t = 0; t < 4; ++t) a4[nx][4 * h + t][r] = q[t];
                }
        }
        if (MODE >= 10 && MODE < 20) {   // K = MODE - 10 independent v_fma_f32 behind every MFMA (no LDS): how many VALU fillers fit a gap?
            constexpr int K = MODE - 10;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    acc[t] = MFMA16(a4[cur][t][r], b[r], acc[t]);
#pragma unroll
                    for (int k = 0; k < K; ++k) vf[k] = fmaf(vf[k], 1.0001f, b[(k + r) & 3]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            return;
        }
        if (MODE >= 30 && MODE < 40) {   // the SAME 2 v_fma per MFMA as mode 12, but issued in bursts: after every B-th MFMA a block of 2B v_fma
            constexpr int B = MODE == 30 ? 1 : (MODE == 31 ? 4 : (MODE == 32 ? 8 : (MODE == 33 ? 16 : 32)));
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    acc[t] = MFMA16(a4[cur][t][r], b[r], acc[t]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (((8 * r + t) % B) == B - 1) {
#pragma unroll
                        for (int k = 0; k < 2 * B; ++k) vf[k & 7] = fmaf(vf[k & 7], 1.0001f, b[(k + r) & 3]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            return;
        }
        if (MODE >= 40 && MODE < 50) {   // 1 v_fma per MFMA (the step kernel's ratio), spread (40) or in bursts of 8 (41) / 32 (42); + one b128 read per 4 MFMAs
            constexpr int B = MODE == 40 ? 1 : (MODE == 41 ? 8 : 32);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    acc[t] = MFMA16(a4[cur][t][r], b[r], acc[t]);
                    const int q = 8 * r + t;
                    const float* base = &lds[((lane & 15) * 140 + 4 * (lane >> 4) + 16 * (it & 31)) & 8191];
                    if ((q & 3) == 3) a4[nx][q >> 2] = *(const f32x4*)(base + 2240 * (q >> 2));
                    __builtin_amdgcn_sched_barrier(0);
                    if ((q % B) == B - 1) {
#pragma unroll
                        for (int k = 0; k < B; ++k) vf[k & 7] = fmaf(vf[k & 7], 1.0001f, b[(k + r) & 3]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            return;
        }
        if (MODE != 0) __builtin_amdgcn_sched_barrier(0x676);
        if (MODE == 5 || MODE == 6 || MODE == 7 || MODE == 20) {   // the same reads, one after every 4th (5: b128) / 2nd (6: b64) / every (7: b32) MFMA, pinned
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    acc[t] = MFMA16(a4[cur][t][r], b[r], acc[t]);
                    const int q = 8 * r + t;   // 0..31
                    const float* base = &lds[((lane & 15) * 140 + 4 * (lane >> 4) + 16 * (it & 31)) & 8191];
                    if ((MODE == 5 || MODE == 20) && (q & 3) == 3) a4[nx][q >> 2] = *(const f32x4*)(base + 2240 * (q >> 2));
                    if (MODE == 20 && (q & 3) == 1) *(f32x4*)&lds[(24576 + lane * 4 + 256 * (q >> 2)) & 32767] = acc[(t + 4) & 7];   // hand over a tile whose last product issued 4 MFMAs ago
                    if (MODE == 6 && (q & 1) == 1) {
                        const float2 v = *(const float2*)(base + 2240 * (q >> 2) + 2 * ((q >> 1) & 1));
                        a4[nx][q >> 2][2 * ((q >> 1) & 1)] = v.x;
                        a4[nx][q >> 2][2 * ((q >> 1) & 1) + 1] = v.y;
                    }
                    if (MODE == 7) a4[nx][q >> 2][q & 3] = base[2240 * (q >> 2) + (q & 3)];
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = MFMA16(a4[cur][t & 7][r], b[r], acc[t]);
            __builtin_amdgcn_sched_barrier(0x7F6);
        }
        }
        if (MODE == 3) {  // VALU consumer of a finished tile, like the relu of the next z0 tile
#pragma unroll
            for (int r = 0; r < 4; ++r) b[r] = fmaxf(acc[0][r] * 1e-30f, 1.f);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int it = 0; it < iters; it += 2) {
        body(std::integral_constant<int, 0>{}, it);
        body(std::integral_constant<int, 1>{}, it + 1);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s + b[0] + vf[0] + vf[1] + vf[2] + vf[3] + vf[4] + vf[5] + vf[6] + vf[7];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int MODE>
static void run(const char* name, int blocks, int threads) {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 1024 * 512 * 4);
    hipMalloc(&cyc, 1024 * 8);
    const int iters = 2000;
    hipFuncSetAttribute((const void*)rate_kernel<NACC, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<NACC, MODE><<<blocks, threads, 140 * 1024>>>(out, cyc, 10);
    hipEventRecord(e0);
    rate_kernel<NACC, MODE><<<blocks, threads, 140 * 1024>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 4 * NACC;
    printf("%-44s blocks %4d waves/blk %d: %6.2f memtime-ticks/MFMA, %6.2f ns/MFMA  (%.3f GHz if 32 cyc)\n", name, blocks, threads / 64,
           h[0] / n, ms * 1e6 / n, 32.0 / (ms * 1e6 / n));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int blocks : {1}) {
        run<8, 0>("8 acc, no fillers", blocks, 256);
        run<18, 0>("18 acc, no fillers", blocks, 256);
        run<8, 1>("8 acc + 8 ds_read_b128 per 32", blocks, 256);
        run<8, 2>("8 acc + 32 ds_read_b32 per 32", blocks, 256);
        run<8, 3>("8 acc + b128 + VALU consumer", blocks, 256);
        run<8, 4>("8 acc + 8 ds_read_b128 (column data) per 32", blocks, 256);
        run<8, 5>("8 acc + 8 b128, one per 4 MFMAs", blocks, 256);
        run<8, 6>("8 acc + 16 b64, one per 2 MFMAs", blocks, 256);
        run<8, 7>("8 acc + 32 b32, one per MFMA", blocks, 256);
        run<8, 0>("8 acc, no fillers + a VALU-only wave per SIMD", blocks, 512);
        run<8, 5>("8 acc + 8 b128 spread + a VALU-only wave", blocks, 512);
        run<8, 20>("MFMA wave: spread b128 reads + 8 tile writes; helper wave: 8 reads, 64 VALU, 8 writes", blocks, 512);
        run<8, 12>("8 acc + 2 v_fma per MFMA", blocks, 256);
        run<8, 14>("8 acc + 4 v_fma per MFMA", blocks, 256);
        run<8, 16>("8 acc + 6 v_fma per MFMA", blocks, 256);
        run<8, 18>("8 acc + 8 v_fma per MFMA", blocks, 256);
        run<8, 30>("2 v_fma per MFMA, after every MFMA", blocks, 256);
        run<8, 31>("2 v_fma per MFMA, 8 after every 4th MFMA", blocks, 256);
        run<8, 32>("2 v_fma per MFMA, 16 after every 8th MFMA", blocks, 256);
        run<8, 33>("2 v_fma per MFMA, 32 after every 16th MFMA", blocks, 256);
        run<8, 34>("2 v_fma per MFMA, 64 after every 32nd MFMA", blocks, 256);
        run<8, 40>("1 v_fma per MFMA spread + b128 per 4", blocks, 256);
        run<8, 41>("1 v_fma per MFMA, 8 after every 8th + b128 per 4", blocks, 256);
        run<8, 42>("1 v_fma per MFMA, 32 after every 32nd + b128 per 4", blocks, 256);
        run<8, 1>("8 acc + 8 ds_read_b128 burst, 1 wave", blocks, 64);
        run<8, 5>("8 acc + 8 b128 spread, 1 wave", blocks, 64);
        run<8, 0>("8 acc, no fillers, 1 wave", blocks, 64);
    }
    return 0;
}
