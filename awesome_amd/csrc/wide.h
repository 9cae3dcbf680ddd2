// wide.h - the layer-by-layer path for ICNN shapes the fused step kernels cannot hold (n_hidden > 130, or more than two hidden layers).
//
// The fused kernels (icnn_step.h, icnn_step2.h) keep the whole weight image of the network in LDS and every activation in registers;
// that is what makes them fast and what limits them to n_hidden <= 130 (77 KB per hidden layer for h = 130; 160 KB of LDS per CU) and
// L <= 2.  A `n_hidden: 256` or the 3 x 350 relu net of notebooks/imageRepresentationTest.ipynb cell 5 needs another shape of program:
// activations of a whole image in HBM (N x h floats per layer: 92 MB at 256x256 x 350 - nothing on a 288 GB part), one launch per
// layer and direction, and the h x h contractions as PLAIN GEMMs.  Plain library GEMMs are what rocBLAS is for (its fp32 kernels run on
// the same v_mfma_f32 instructions); it is opened with dlopen at the first wide call, so the fused path never depends on it, and
// atomics are switched off for the handle (the point-contraction of the weight gradient has K = n_points: no split-K atomics, results
// reproducible).  Everything that is not a plain GEMM is a hand-written kernel below: layer 0 with its activation (the encode stage),
// bias + skip + relu epilogues, output layer + sigmoid + data term + dL/dlogit, relu / activation masks of the backward pass,
// fixed-order reductions; the optimizer step is icnn_update_kernel itself on a one-"slab" view of the gradient vector.
//
// Same arithmetic as the reference (awesome/model/convex_net.py:205-214): z0 = act0(W_in x + b_in); z_{k+1} = relu(W_k z_k + b_k + S_k x);
// y = w_o . z_L + b_o + s_o . x.  Layout: activations row-major [N][h] (a point's units contiguous), parameters in the flat order of
// include/inrfit.h (torch's row-major [out][in] Linear weights), so every GEMM reads the parameters where they lie and every weight
// gradient is written by the GEMM straight into the flat gradient vector.
#pragma once
#include <dlfcn.h>

#include "icnn_step.h"

namespace {

// ---- rocBLAS through dlopen (column-major library; row-major C = op(A) op(B) is the column-major product of the swapped operands) ------
enum { RB_OP_N = 111, RB_OP_T = 112 };   // rocblas_operation_none / _transpose
struct WideBlas {
    void* lib = nullptr;
    void* handle = nullptr;
    int (*create)(void**) = nullptr;
    int (*set_stream)(void*, hipStream_t) = nullptr;
    int (*set_atomics)(void*, int) = nullptr;
    int (*sgemm)(void*, int, int, int, int, int, const float*, const float*, int, const float*, int, const float*, float*, int) = nullptr;
    int (*sgemv)(void*, int, int, int, const float*, const float*, int, const float*, int, const float*, float*, int) = nullptr;
    int (*sgemm_sb)(void*, int, int, int, int, int, const float*, const float*, int, long long, const float*, int, long long, const float*,
                    float*, int, long long, int) = nullptr;
    bool ok = false;
};

inline WideBlas& wide_blas() {
    static WideBlas b = [] {
        WideBlas w;
        for (const char* name : {"librocblas.so", "librocblas.so.5", "/opt/rocm/lib/librocblas.so"}) {
            w.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (w.lib) break;
        }
        if (!w.lib) return w;
        w.create = (int (*)(void**))dlsym(w.lib, "rocblas_create_handle");
        w.set_stream = (int (*)(void*, hipStream_t))dlsym(w.lib, "rocblas_set_stream");
        w.set_atomics = (int (*)(void*, int))dlsym(w.lib, "rocblas_set_atomics_mode");
        w.sgemm = (decltype(w.sgemm))dlsym(w.lib, "rocblas_sgemm");
        w.sgemv = (decltype(w.sgemv))dlsym(w.lib, "rocblas_sgemv");
        w.sgemm_sb = (decltype(w.sgemm_sb))dlsym(w.lib, "rocblas_sgemm_strided_batched");
        if (!w.create || !w.set_stream || !w.set_atomics || !w.sgemm || !w.sgemv || !w.sgemm_sb) return w;
        if (w.create(&w.handle) != 0) return w;
        if (w.set_atomics(w.handle, 0 /* rocblas_atomics_not_allowed */) != 0) return w;
        w.ok = true;
        return w;
    }();
    return b;
}

// row-major C[M x N] = alpha op(A) op(B) + beta C; lda / ldb / ldc = row lengths of the stored matrices
inline int gemm_rm(hipStream_t s, bool tA, bool tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                   int ldc, float beta = 0.f) {
    WideBlas& b = wide_blas();
    const float alpha = 1.f;
    if (b.set_stream(b.handle, s) != 0) return INR_ELAUNCH;
    return b.sgemm(b.handle, tB ? RB_OP_T : RB_OP_N, tA ? RB_OP_T : RB_OP_N, N, M, K, &alpha, B, ldb, A, lda, &beta, C, ldc) == 0
               ? INR_OK : INR_ELAUNCH;
}
// y[rows] = A x for a row-major A[rows x cols] with row stride lda >= cols
inline int gemv_rm(hipStream_t s, int rows, int cols, const float* A, int lda, const float* x, float* y) {
    WideBlas& b = wide_blas();
    const float alpha = 1.f, beta = 0.f;
    if (b.set_stream(b.handle, s) != 0) return INR_ELAUNCH;
    // row-major A [rows x lda] = column-major [lda x rows]: A x is the TRANSPOSED column-major product
    return b.sgemv(b.handle, RB_OP_T, cols, rows, &alpha, A, lda, x, 1, &beta, y, 1) == 0 ? INR_OK : INR_ELAUNCH;
}

// Contractions over the POINTS (weight gradients: out[a x b] = A^T B with A [N x a], B [N x b], N = 65 536 and a, b a few hundred):
// one GEMM would keep a handful of workgroups busy for milliseconds (no split-K: atomics are off, the sums must have a fixed order).
// The points are cut into chunks of WIDE_CHUNK rows, every chunk is one member of a strided-batched GEMM writing its own partial
// [a x b] tile, and wide_reduce_kernel adds the partials in chunk order and scatters them to the flat gradient vector.
constexpr int WIDE_CHUNK = 1024;
inline int splitk_tn(hipStream_t s, long long N, int a, int b, const float* A, int lda, const float* B, int ldb, float* part) {
    WideBlas& w = wide_blas();
    const float alpha = 1.f, beta = 0.f;
    if (w.set_stream(w.handle, s) != 0) return INR_ELAUNCH;
    const int full = (int)(N / WIDE_CHUNK), rem = (int)(N - (long long)full * WIDE_CHUNK);
    // row-major C[a x b] = A^T B  ==  column-major C'[b x a] = B' A'^T with B' = B viewed column-major [ldb x K], A' [lda x K]
    if (full > 0 && w.sgemm_sb(w.handle, RB_OP_N, RB_OP_T, b, a, WIDE_CHUNK, &alpha, B, ldb, (long long)WIDE_CHUNK * ldb, A, lda,
                               (long long)WIDE_CHUNK * lda, &beta, part, b, (long long)a * b, full) != 0) return INR_ELAUNCH;
    if (rem > 0 && w.sgemm(w.handle, RB_OP_N, RB_OP_T, b, a, rem, &alpha, B + (size_t)full * WIDE_CHUNK * ldb, ldb,
                           A + (size_t)full * WIDE_CHUNK * lda, lda, &beta, part + (size_t)full * a * b, b) != 0) return INR_ELAUNCH;
    return INR_OK;
}
inline int splitk_parts(long long N) { return (int)((N + WIDE_CHUNK - 1) / WIDE_CHUNK); }

// ---- flat parameter offsets for any (h, C, L) ---------------------------------------------------------------------------------------------
struct WideMap {
    int h, C, L, P;
    __host__ __device__ int p_win() const { return 0; }
    __host__ __device__ int p_bin() const { return h * C; }
    __host__ __device__ int p_w(int k) const { return h * C + h + k * (h * h + h + h * C); }
    __host__ __device__ int p_b(int k) const { return p_w(k) + h * h; }
    __host__ __device__ int p_s(int k) const { return p_b(k) + h; }
    __host__ __device__ int p_wo() const { return p_w(L); }
    __host__ __device__ int p_bo() const { return p_wo() + h; }
    __host__ __device__ int p_so() const { return p_bo() + 1; }
};
inline WideMap make_wide_map(int h, int C, int L) {
    WideMap m{h, C, L, 0};
    m.P = m.p_so() + C;
    return m;
}

constexpr int WIDE_MAX_HIDDEN = 1024, WIDE_MAX_LAYERS = 8;

// ---- kernels ---------------------------------------------------------------------------------------------------------------------------
// Activations are stored [N][hs] with hs = h + 1 + C: the h units of the layer, then the "ext" inputs (1, x_0 .. x_{C-1}) - as in the
// fused kernels, bias and skip weights ride in the contractions: dW_ext [h x hs] = dz^T Z_ext holds dW, db and dS of a layer at once.

// z0[p][j] = act0(W_in[j] . x_p + b_in[j]) for j < h, the ext columns for j >= h; pre0 (optional, [N][h]) keeps the pre-activation for
// the periodic activations' derivative.  Reads the coordinates from the grid descriptor.
template <int C>
__global__ __launch_bounds__(256) void wide_layer0_kernel(InrGridDesc gd, int img, const float* __restrict__ win, const float* __restrict__ bin,
                                                          long long N, int h, int act0, float omega, float* __restrict__ z0,
                                                          float* __restrict__ pre0) {
    const int hs = h + 1 + C;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= N * hs) return;
    const long long p = e / hs;
    const int j = (int)(e - p * hs);
    float x[C];
    if (gd.mode == INR_GRID_SEPARABLE) {
        const int row = (int)(p / gd.width);
        x[0] = gd.xs[p - (long long)row * gd.width];
        x[1] = gd.ys[row];
        if (C > 2) x[C - 1] = gd.ts ? gd.ts[img] : 0.f;
    } else {
        const float* cp = gd.coords + (size_t)img * gd.coords_image_stride;
#pragma unroll
        for (int c = 0; c < C; ++c) x[c] = cp[(size_t)c * N + p];
    }
    if (j >= h) {
        float v = 1.f;
#pragma unroll
        for (int c = 0; c < C; ++c) v = (j == h + 1 + c) ? x[c] : v;
        z0[e] = v;
        return;
    }
    float v = bin[j];
#pragma unroll
    for (int c = 0; c < C; ++c) v = fmaf(win[j * C + c], x[c], v);
    if (pre0) pre0[p * h + j] = v;
    z0[e] = act0 == INR_ACT_COS ? hw_cos(v) : (act0 == INR_ACT_SIN ? hw_sin(omega * v) : fmaxf(v, 0.f));
}

// in place: z[p][j] = relu(z[p][j] + b[j] + S[j] . x_p) for j < h (z holds W z_prev from the GEMM); the ext columns copied from zprev
template <int C>
__global__ __launch_bounds__(256) void wide_hidden_epilogue_kernel(float* __restrict__ z, const float* __restrict__ zprev,
                                                                   const float* __restrict__ b, const float* __restrict__ S, long long N,
                                                                   int h) {
    const int hs = h + 1 + C;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= N * hs) return;
    const long long p = e / hs;
    const int j = (int)(e - p * hs);
    if (j >= h) {
        z[e] = zprev[e];
        return;
    }
    float v = z[e] + b[j];
#pragma unroll
    for (int c = 0; c < C; ++c) v = fmaf(S[j * C + c], zprev[p * hs + h + 1 + c], v);
    z[e] = fmaxf(v, 0.f);
}

// y[p] (in: w_o . z_L from the GEMV) += b_o + s_o . x_p; logits out; TRAIN: data term -> dy[p], per-block partials of the loss
struct WideOutArgs {
    float* y;               // [N] in/out
    const float* zl;        // [N][hs] (its ext columns carry x)
    const float* sc;        // b_o, s_o[C]  (flat parameters at p_bo)
    const float* target;    // [N] or null
    const float* coef;      // c_fg, c_bg of this image
    float* logits;          // [N] or null
    float* dy;              // [N] (train)
    float* part;            // [blocks] loss partials
    long long N;
    int h, C, loss_kind, train;
};
__global__ __launch_bounds__(256) void wide_out_kernel(const WideOutArgs a) {
    __shared__ float sm[4];
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool valid = p < a.N;
    float l = 0.f;
    if (valid) {
        const int hs = a.h + 1 + a.C;
        float y = a.y[p] + a.sc[0];
        for (int c = 0; c < a.C; ++c) y = fmaf(a.sc[1 + c], a.zl[p * hs + a.h + 1 + c], y);
        if (a.logits) a.logits[p] = y;
        if (a.train) {
            const float tg = a.target[p];
            float dy;
            if (a.loss_kind == INR_LOSS_EXTERNAL) {
                dy = tg;
            } else {
                const float pr = 1.f / (1.f + expf(-y));
                const float cw = tg < 0.5f ? a.coef[0] : a.coef[1];
                if (a.loss_kind == INR_LOSS_SE) {
                    const float d = tg - pr;
                    l = d * d * cw;
                    dy = 2.f * (pr - tg) * pr * (1.f - pr) * cw;
                } else {
                    const float lp = bce_log(pr), lq = bce_log(1.f - pr);   // clamped at -100, NaN kept (torch.nn.BCELoss)
                    l = -(tg * lp + (1.f - tg) * lq) * cw;
                    const float pq = pr * (1.f - pr);
                    dy = (pr - tg) / fmaxf(pq, 1e-12f) * pq * cw;
                }
            }
            a.dy[p] = dy;
        }
    }
    if (!a.train) return;
    const float v = sum_over_groups(sum_over_points(l));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) a.part[blockIdx.x] = ((sm[0] + sm[1]) + sm[2]) + sm[3];
}

// one block: fixed-order total of the loss partials -> grads[P]
__global__ __launch_bounds__(256) void wide_loss_finish_kernel(const float* __restrict__ part, int blocks, float* __restrict__ grads, int P) {
    __shared__ float sm[4];
    float v = 0.f;
    for (int b = threadIdx.x; b < blocks; b += 256) v += part[b];
    v = sum_over_groups(sum_over_points(v));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) grads[P] = ((sm[0] + sm[1]) + sm[2]) + sm[3];
}

// dz[p][j] = dy[p] w_o[j] [z_L[p][j] > 0]      (dz: [N][h], z_L: [N][hs])
__global__ __launch_bounds__(256) void wide_dz_last_kernel(const float* __restrict__ dy, const float* __restrict__ wo,
                                                           const float* __restrict__ zl, long long N, int h, int hs, float* __restrict__ dz) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= N * h) return;
    const long long p = e / h;
    const int j = (int)(e - p * h);
    dz[e] = zl[p * hs + j] > 0.f ? dy[p] * wo[j] : 0.f;
}

// in place: dz[p][j] *= act'(.) of the layer that produced z: relu mask [z > 0], or the periodic activations' derivative from pre0
__global__ __launch_bounds__(256) void wide_mask_kernel(float* __restrict__ dz, const float* __restrict__ z, const float* __restrict__ pre0,
                                                        long long N, int h, int hs, int act, float omega) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= N * h) return;
    const long long p = e / h;
    const int j = (int)(e - p * h);
    float m;
    if (act == INR_ACT_COS) m = -hw_sin(pre0[e]);
    else if (act == INR_ACT_SIN) m = omega * hw_cos(omega * pre0[e]);
    else m = z[p * hs + j] > 0.f ? 1.f : 0.f;
    dz[e] *= m;
}

// sum of the split-K partials [parts][a][b] in chunk order, scattered into the flat gradient vector:
//   mode 0 (hidden layer k): row i = unit, column j < h -> W_k[i][j]; j == h -> b_k[i]; j > h -> S_k[i][j - h - 1]
//   mode 1 (layer 0, B = the ext columns):           column 0 -> b_in[i]; j >= 1 -> W_in[i][j - 1]
//   mode 2 (output layer, a = 1, b = hs):            column j < h -> w_o[j]; j == h -> b_o; j > h -> s_o[j - h - 1]
__global__ __launch_bounds__(256) void wide_reduce_kernel(const float* __restrict__ part, int parts, int a, int b, int mode, WideMap m, int k,
                                                          float* __restrict__ grads) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= a * b) return;
    const int i = e / b, j = e - i * b;
    float v = 0.f;
    for (int q = 0; q < parts; ++q) v += part[(size_t)q * a * b + e];
    int dst;
    if (mode == 0) dst = j < m.h ? m.p_w(k) + i * m.h + j : (j == m.h ? m.p_b(k) + i : m.p_s(k) + i * m.C + (j - m.h - 1));
    else if (mode == 1) dst = j == 0 ? m.p_bin() + i : m.p_win() + i * m.C + (j - 1);
    else dst = j < m.h ? m.p_wo() + j : (j == m.h ? m.p_bo() : m.p_so() + (j - m.h - 1));
    grads[dst] = v;
}

// ---- workspace --------------------------------------------------------------------------------------------------------------------------
struct WideWs {
    float *z[WIDE_MAX_LAYERS + 1], *pre0, *dza, *dzb, *y, *dy, *part, *lossp, *grads, *coef;
    int blocks, hs;
    long long bytes;
};

inline long long wide_align(long long b) { return (b + 255) / 256 * 256; }

inline WideWs carve_wide(const WideMap& m, long long N, bool need_pre0, void* base) {
    WideWs w{};
    char* b = (char*)base;
    long long off = 0;
    auto take = [&](long long bytes) { float* p = (float*)(b + off); off += wide_align(bytes); return p; };
    w.blocks = (int)((N + 255) / 256);
    w.hs = m.h + 1 + m.C;
    for (int k = 0; k <= m.L; ++k) w.z[k] = take(N * w.hs * 4);
    w.pre0 = need_pre0 ? take(N * m.h * 4) : nullptr;
    w.dza = take(N * m.h * 4);
    w.dzb = take(N * m.h * 4);
    w.y = take(N * 4);
    w.dy = take(N * 4);
    w.part = take((long long)splitk_parts(N) * m.h * w.hs * 4);
    w.lossp = take((long long)w.blocks * 4);
    w.grads = take(((long long)m.P + 1 + 31) / 32 * 32 * 4);
    w.coef = nullptr;
    w.bytes = off;
    return w;
}

inline long long wide_total_bytes(const WideMap& m, long long N, bool pre0, int n_images) {   // + the per-image loss coefficients
    return carve_wide(m, N, pre0, nullptr).bytes + wide_align((long long)n_images * 2 * 4);
}

inline bool wide_shape_ok(const InrModelDesc* md) {
    return md && md->kind == INR_MODEL_ICNN && md->n_hidden >= 1 && md->n_hidden <= WIDE_MAX_HIDDEN && md->n_layers >= 1 &&
           md->n_layers <= WIDE_MAX_LAYERS && (md->in_features == 2 || md->in_features == 3);
}

#define WIDE_EW(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, s

// forward of ONE image; with `train`: also dy and the loss (w.grads[P])
inline int wide_forward(const WideMap& m, const WideWs& w, const InrModelDesc* md, const float* params, const InrGridDesc* grid, int img,
                        const float* target, int loss_kind, bool train, float* logits, hipStream_t s) {
    const long long N = grid->n_points;
    const int h = m.h, C = m.C, hs = w.hs;
    if (C == 2) hipLaunchKernelGGL(wide_layer0_kernel<2>, WIDE_EW(N * hs), *grid, img, params + m.p_win(), params + m.p_bin(), N, h, md->act0, md->act_omega, w.z[0], w.pre0);
    else hipLaunchKernelGGL(wide_layer0_kernel<3>, WIDE_EW(N * hs), *grid, img, params + m.p_win(), params + m.p_bin(), N, h, md->act0, md->act_omega, w.z[0], w.pre0);
    for (int k = 0; k < m.L; ++k) {
        // z_{k+1}pre [N x h] = z_k [N x h] . W_k^T   (W_k stored [h_out][h_in]; rows of the z buffers are hs long)
        int rc = gemm_rm(s, false, true, (int)N, h, h, w.z[k], hs, params + m.p_w(k), h, w.z[k + 1], hs);
        if (rc) return rc;
        if (C == 2) hipLaunchKernelGGL(wide_hidden_epilogue_kernel<2>, WIDE_EW(N * hs), w.z[k + 1], w.z[k], params + m.p_b(k), params + m.p_s(k), N, h);
        else hipLaunchKernelGGL(wide_hidden_epilogue_kernel<3>, WIDE_EW(N * hs), w.z[k + 1], w.z[k], params + m.p_b(k), params + m.p_s(k), N, h);
    }
    int rc = gemv_rm(s, (int)N, h, w.z[m.L], hs, params + m.p_wo(), w.y);   // y = Z_L w_o
    if (rc) return rc;
    WideOutArgs a{};
    a.y = w.y; a.zl = w.z[m.L]; a.sc = params + m.p_bo(); a.target = target; a.coef = w.coef; a.logits = logits; a.dy = w.dy; a.part = w.lossp;
    a.N = N; a.h = h; a.C = C; a.loss_kind = loss_kind; a.train = train ? 1 : 0;
    hipLaunchKernelGGL(wide_out_kernel, dim3(w.blocks), dim3(256), 0, s, a);
    if (train) hipLaunchKernelGGL(wide_loss_finish_kernel, dim3(1), dim3(256), 0, s, w.lossp, w.blocks, w.grads, m.P);
    return hipGetLastError() == hipSuccess ? INR_OK : INR_ELAUNCH;
}

// backward of ONE image from w.dy: every parameter gradient into w.grads (flat order)
inline int wide_backward(const WideMap& m, const WideWs& w, const InrModelDesc* md, const float* params, long long N, hipStream_t s) {
    const int h = m.h, C = m.C, hs = w.hs, parts = splitk_parts(N);
    float* g = w.grads;
    int rc;
    // output layer: (dw_o | db_o | ds_o) [1 x hs] = dy^T Z_L,ext
    if ((rc = splitk_tn(s, N, 1, hs, w.dy, 1, w.z[m.L], hs, w.part))) return rc;
    hipLaunchKernelGGL(wide_reduce_kernel, WIDE_EW(hs), w.part, parts, 1, hs, 2, m, 0, g);
    hipLaunchKernelGGL(wide_dz_last_kernel, WIDE_EW(N * h), w.dy, params + m.p_wo(), w.z[m.L], N, h, hs, w.dza);
    float *dz = w.dza, *dzn = w.dzb;
    for (int k = m.L - 1; k >= 0; --k) {
        if ((rc = splitk_tn(s, N, h, hs, dz, h, w.z[k], hs, w.part))) return rc;                             // (dW_k | db_k | dS_k) = dz^T Z_k,ext
        hipLaunchKernelGGL(wide_reduce_kernel, WIDE_EW(h * hs), w.part, parts, h, hs, 0, m, k, g);
        if ((rc = gemm_rm(s, false, false, (int)N, h, h, dz, h, params + m.p_w(k), h, dzn, h))) return rc;   // dz_k = dz W_k
        hipLaunchKernelGGL(wide_mask_kernel, WIDE_EW(N * h), dzn, w.z[k], w.pre0, N, h, hs, k == 0 ? md->act0 : INR_ACT_RELU, md->act_omega);
        float* t = dz; dz = dzn; dzn = t;
    }
    if ((rc = splitk_tn(s, N, h, 1 + C, dz, h, w.z[0] + h, hs, w.part))) return rc;                          // (db_in | dW_in) = dz0^T (1, X)
    hipLaunchKernelGGL(wide_reduce_kernel, WIDE_EW(h * (1 + C)), w.part, parts, h, 1 + C, 1, m, 0, g);
    return hipGetLastError() == hipSuccess ? INR_OK : INR_ELAUNCH;
}

}  // namespace
