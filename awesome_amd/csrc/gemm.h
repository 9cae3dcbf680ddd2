// gemm.h - hand-written fp32 MFMA GEMM for the layer-by-layer path (wide.h) and the star prior's minibatch products (star.h).
//
//     C [M x N] = op(A) . op(B)        op(A) = A [M][K] or A^T with A stored [K][M];   op(B) = B^T with B stored [N][K], or B [K][N]
//
// row-major operands with arbitrary row strides, any M, N, K (edges are guarded, never padded in memory), optional split of the
// contraction over blockIdx.z (every z writes its own [M x N] partial: the weight gradients contract over the 65 536 points of an
// image and are added up in chunk order by a second kernel - no atomics, reproducible), and an epilogue functor applied to every
// output element in registers (bias + skip + relu of a hidden layer; the relu / periodic-activation mask of the backward pass), so
// the activations make one trip to HBM per layer and direction instead of three.
//
// Shape of the kernel (gfx950): 256 threads = 4 waves, a 128 x 128 output tile per workgroup, 64 x 64 per wave = 4 x 4 accumulator
// tiles of v_mfma_f32_16x16x4_f32 (64 accumulator registers), K in steps of 16 through LDS, global loads of step s + 1 in flight
// while step s multiplies (register-staged double buffer: two 20 KB LDS buffers, one barrier per step).
//
// How the operands reach the matrix pipe with ONE ds_read_b128 per operand tile and 16 k (8 LDS reads per 64 MFMAs):
//   * an operand that is contiguous along k (A [M][K], B [N][K]) sits in LDS as [row][16 k] (row stride 20 floats: 16-byte aligned,
//     conflict-free); lane (g, l15) reads row 16 i + l15, k = 4 g .. 4 g + 3.  MFMA number kk of the step takes element kk of that
//     vector from every lane group, i.e. it contracts k in {kk, 4 + kk, 8 + kk, 12 + kk} - any partition of the 16 k will do as long
//     as both operands use the same one;
//   * an operand that is contiguous along its free index (A^T stored [K][M], B [K][N]) sits in LDS as [k][128 (+4)]; lane (g, l15)
//     reads row k = 4 g + kk, columns 4 l15 .. 4 l15 + 3 - element i of the vector is its value for accumulator tile i, i.e. tile i
//     holds the rows / columns 4 q + i instead of 16 i + q.  The permutation is undone where the tile is stored (for the columns it
//     even helps: a lane's four tiles are four consecutive floats of C, one 16-byte store).
// The reference arithmetic these products serve: awesome/model/convex_net.py:205-214 (z_{k+1} = relu(W_k z_k + b_k + S_k x)) and its
// backward pass; star.ipynb cell 2 / 3 for the star prior.
#pragma once
#include "icnn_step.h"

namespace {

constexpr int GM_BM = 128, GM_BN = 128, GM_BK = 16;
constexpr int GM_LDK = 20;            // [row][k] layout: floats per row
constexpr int GM_LDF = GM_BM + 4;     // [k][free] layout: floats per k-row
constexpr int GM_STAGE = GM_BM * GM_LDK > GM_BK * GM_LDF ? GM_BM * GM_LDK : GM_BK * GM_LDF;   // floats per operand and buffer

enum { GEMM_EPI_STORE = 0, GEMM_EPI_HIDDEN = 1, GEMM_EPI_MASK = 2 };

struct GemmArgs {
    const float* A;        // TA ? [K][lda] : [M][lda]
    const float* B;        // TB ? [N][ldb] : [K][ldb]      (TB: C = A . B^T)
    float* C;              // [M][ldc] (+ z * c_split_stride)
    int M, N, K, lda, ldb, ldc;
    int k_per_split;       // multiple of GM_BK; gridDim.z = ceil(K / k_per_split)
    long long c_split_stride;
    // epilogue
    int epi;
    const float* bias;     // HIDDEN: b [N]
    const float* skip;     // HIDDEN: S [N][C]
    const float* ext;      // HIDDEN: x_c of row m = ext[m * ext_ld + 1 + c] (the ext columns (1, x) of the previous layer's activations)
    int ext_ld, C_in;
    const float* mask;     // MASK: multiply by [mask[m * mask_ld + n] > 0] (mask_act == relu) or by the activation's derivative at it
    int mask_ld, mask_act;
    float omega;
    int vecA, vecB;        // widest aligned load of the operand: 4 floats (base and row stride multiples of 16 bytes), 2, or 1
};

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float As[2][GM_STAGE];
    __shared__ __attribute__((aligned(16))) float Bs[2][GM_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GM_BM, n0 = blockIdx.x * GM_BN;   // n fastest: the column tiles of one row tile run together and share its A rows in L2
    const int k_lo = blockIdx.z * a.k_per_split;
    const int k_hi = min(a.K, k_lo + a.k_per_split);
    const int steps = (k_hi - k_lo + GM_BK - 1) / GM_BK;

    // ---- global -> registers: every thread owns 8 consecutive floats of each operand tile -----------------------------------------
    // operand contiguous along k: tile [128 rows][16 k], thread -> row tid >> 1, k offset (tid & 1) * 8
    // operand contiguous along its free index: tile [16 k][128], thread -> k row tid >> 4, offset (tid & 15) * 8
    auto load8 = [&](const float* base, int ld, bool kcontig, int row0, int rows, int kbase, int vec, f32x4 (&v)[2]) {
        int r, c;          // r: index along the slow (row) dimension of the STORED matrix, c: along its contiguous dimension
        bool rok;
        int cmax;
        if (kcontig) {
            r = row0 + (tid >> 1);
            c = kbase + (tid & 1) * 8;
            rok = r < rows;
            cmax = k_hi;
        } else {
            r = kbase + (tid >> 4);
            c = row0 + (tid & 15) * 8;
            rok = r < k_hi;
            cmax = rows;
        }
        const float* p = base + (size_t)r * ld + c;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int cc = c + 4 * h;
            if (rok && vec == 4 && cc + 3 < cmax) {
                v[h] = *(const f32x4*)(p + 4 * h);
            } else if (rok && vec == 2 && cc + 3 < cmax) {   // rows on 8-byte boundaries (an even row length, e.g. 350)
                const f32x2 lo = *(const f32x2*)(p + 4 * h), hi = *(const f32x2*)(p + 4 * h + 2);
                v[h] = f32x4{lo[0], lo[1], hi[0], hi[1]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[h][e] = (rok && cc + e < cmax) ? p[4 * h + e] : 0.f;
            }
        }
    };
    auto store8 = [&](float* stage, bool kcontig, const f32x4 (&v)[2]) {
        float* d = kcontig ? stage + (tid >> 1) * GM_LDK + (tid & 1) * 8 : stage + (tid >> 4) * GM_LDF + (tid & 15) * 8;
        *(f32x4*)d = v[0];
        *(f32x4*)(d + 4) = v[1];
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 ra[2], rb[2];
    if (steps > 0) {
        load8(a.A, a.lda, !TA, m0, a.M, k_lo, a.vecA, ra);
        load8(a.B, a.ldb, TB, n0, a.N, k_lo, a.vecB, rb);
        store8(As[0], !TA, ra);
        store8(Bs[0], TB, rb);
    }
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        const int cur = s & 1;
        if (s + 1 < steps) {   // next step's tiles: in flight while this step multiplies
            load8(a.A, a.lda, !TA, m0, a.M, k_lo + (s + 1) * GM_BK, a.vecA, ra);
            load8(a.B, a.ldb, TB, n0, a.N, k_lo + (s + 1) * GM_BK, a.vecB, rb);
        }
        const float* sa = As[cur];
        const float* sb = Bs[cur];
        if (!TA && TB) {          // both contiguous along k: one read per tile, then 4 x 16 MFMAs
            f32x4 av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = *(const f32x4*)(sa + (wm * 64 + 16 * i + l15) * GM_LDK + 4 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = *(const f32x4*)(sb + (wn * 64 + 16 * j + l15) * GM_LDK + 4 * g);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = MFMA16(av[i][kk], bv[j][kk], acc[i][j]);
        } else {
            f32x4 av[4], bv[4];   // k-contiguous operand: [tile]; free-contiguous operand: read per kk
            if (!TA) {
#pragma unroll
                for (int i = 0; i < 4; ++i) av[i] = *(const f32x4*)(sa + (wm * 64 + 16 * i + l15) * GM_LDK + 4 * g);
            }
            if (TB) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[j] = *(const f32x4*)(sb + (wn * 64 + 16 * j + l15) * GM_LDK + 4 * g);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                f32x4 af, bf;
                if (TA) af = *(const f32x4*)(sa + (4 * g + kk) * GM_LDF + wm * 64 + 4 * l15);
                if (!TB) bf = *(const f32x4*)(sb + (4 * g + kk) * GM_LDF + wn * 64 + 4 * l15);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = MFMA16(TA ? af[i] : av[i][kk], TB ? bv[j][kk] : bf[j], acc[i][j]);
            }
        }
        if (s + 1 < steps) {
            store8(As[cur ^ 1], !TA, ra);
            store8(Bs[cur ^ 1], TB, rb);
        }
        __syncthreads();
    }

    // ---- epilogue: acc[i][j][r] = C[m][n] with m = m0 + 64 wm + mrow(i, 4 g + r), n = n0 + 64 wn + ncol(j, l15) ---------------------
    float* __restrict__ Cz = a.C + (size_t)blockIdx.z * a.c_split_stride;
    auto ncol = [&](int j) { return n0 + wn * 64 + (TB ? 16 * j + l15 : 4 * l15 + j); };
    // HIDDEN: the per-column constants (bias, skip weights) of this lane's four columns once, the point's coordinates once per row
    float bn[4] = {0.f, 0.f, 0.f, 0.f}, sn[4][3] = {};
    if (a.epi == GEMM_EPI_HIDDEN) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = ncol(j);
            if (n < a.N) {
                bn[j] = a.bias[n];
                for (int c = 0; c < a.C_in; ++c) sn[j][c] = a.skip[n * a.C_in + c];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = 4 * g + r;
            const int m = m0 + wm * 64 + (TA ? 4 * q + i : 16 * i + q);
            if (m >= a.M) continue;
            float xm[3] = {0.f, 0.f, 0.f};
            if (a.epi == GEMM_EPI_HIDDEN)
                for (int c = 0; c < a.C_in; ++c) xm[c] = a.ext[(size_t)m * a.ext_ld + 1 + c];
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = ncol(j);
                float v = acc[i][j][r];
                if (a.epi == GEMM_EPI_HIDDEN) {
                    v += bn[j];
#pragma unroll
                    for (int c = 0; c < 3; ++c) v = fmaf(sn[j][c], xm[c], v);   // (unused channels: 0 * 0)
                    v = fmaxf(v, 0.f);
                } else if (a.epi == GEMM_EPI_MASK && n < a.N) {
                    const float z = a.mask[(size_t)m * a.mask_ld + n];
                    if (a.mask_act == INR_ACT_COS) v *= -hw_sin(z);
                    else if (a.mask_act == INR_ACT_SIN) v *= a.omega * hw_cos(a.omega * z);
                    else v = z > 0.f ? v : 0.f;
                }
                o[j] = v;
            }
            if (!TB) {   // a lane's four column tiles are four consecutive columns 4 l15 + j: one 16-byte store where it may
                const int n = ncol(0);
                float* dst = Cz + (size_t)m * a.ldc + n;
                if (n + 3 < a.N && ((a.ldc & 3) == 0) && ((((size_t)Cz) & 15) == 0)) {
                    *(f32x4*)dst = o;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (n + j < a.N) dst[j] = o[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = ncol(j);
                    if (n < a.N) Cz[(size_t)m * a.ldc + n] = o[j];
                }
            }
        }
    }
}

inline int gemm_vec_width(const float* p, int ld) {
    if ((((size_t)p) & 15) == 0 && (ld & 3) == 0) return 4;
    if ((((size_t)p) & 7) == 0 && (ld & 1) == 0) return 2;
    return 1;
}

// row-major C[M x N] = op(A) op(B); `splits` > 1 cuts K into that many z-slices of k_per_split (multiple of 16) writing
// C + z * c_split_stride.  Returns INR_OK / INR_ELAUNCH.
inline int gemm_launch(hipStream_t s, bool tA, bool tB, GemmArgs g) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return INR_EINVAL;
    if (g.k_per_split <= 0) g.k_per_split = (g.K + GM_BK - 1) / GM_BK * GM_BK;
    const int splits = (g.K + g.k_per_split - 1) / g.k_per_split;
    g.vecA = gemm_vec_width(g.A, g.lda);
    g.vecB = gemm_vec_width(g.B, g.ldb);
    const dim3 grid((g.N + GM_BN - 1) / GM_BN, (g.M + GM_BM - 1) / GM_BM, splits);
    if (!tA && tB) hipLaunchKernelGGL((gemm_kernel<false, true>), grid, dim3(256), 0, s, g);
    else if (!tA && !tB) hipLaunchKernelGGL((gemm_kernel<false, false>), grid, dim3(256), 0, s, g);
    else if (tA && !tB) hipLaunchKernelGGL((gemm_kernel<true, false>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_kernel<true, true>), grid, dim3(256), 0, s, g);
    return hipGetLastError() == hipSuccess ? INR_OK : INR_ELAUNCH;
}

// the plain product (no epilogue, no split): what wide.h / star.h used to hand to a BLAS library.
// tA: A is stored [K][lda] (C = A^T ...);  tB: B is stored [N][ldb] (C = ... B^T)
inline int gemm_rm(hipStream_t s, bool tA, bool tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                   int ldc) {
    GemmArgs g{};
    g.A = A; g.B = B; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.epi = GEMM_EPI_STORE;
    return gemm_launch(s, tA, tB, g);
}

// C [M x N] = sum over the z-slices of part [splits][M x N], slices added in order (the fixed-order second half of a split-K product)
__global__ __launch_bounds__(256) void gemm_splitk_sum_kernel(const float* __restrict__ part, int splits, long long mn, float* __restrict__ C) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= mn) return;
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += part[(size_t)z * mn + e];
    C[e] = v;
}

constexpr int GEMM_SPLITK_MAX = 16;
// k_per_split of a contraction of length K cut into at most GEMM_SPLITK_MAX slices of at least 128
inline int gemm_splitk_len(int K) {
    int len = (K + GEMM_SPLITK_MAX - 1) / GEMM_SPLITK_MAX;
    len = (len + GM_BK - 1) / GM_BK * GM_BK;
    return len < 128 ? 128 : len;
}
// the plain product with the contraction cut into slices (a long K with a small M x N: a weight gradient over a minibatch):
// `part` holds GEMM_SPLITK_MAX x M x N floats; C is dense [M][N]
inline int gemm_rm_splitk(hipStream_t s, bool tA, bool tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                          float* part) {
    GemmArgs g{};
    g.A = A; g.B = B; g.C = part;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = N;
    g.k_per_split = gemm_splitk_len(K);
    g.c_split_stride = (long long)M * N;
    g.epi = GEMM_EPI_STORE;
    const int splits = (K + g.k_per_split - 1) / g.k_per_split;
    int rc = gemm_launch(s, tA, tB, g);
    if (rc) return rc;
    const long long mn = (long long)M * N;
    hipLaunchKernelGGL(gemm_splitk_sum_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, s, (const float*)part, splits, mn, C);
    return hipGetLastError() == hipSuccess ? INR_OK : INR_ELAUNCH;
}

}  // namespace
