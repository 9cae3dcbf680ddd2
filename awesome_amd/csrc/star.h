// star.h - the star-shape prior of the teaser as a device-resident fit (notebooks/icml_teaser_code/star_shaped/
// star.ipynb cells 2-3).
//
//     x' = x + offset,  r = |x'|,  x^ = x' / (0.01 + r)
//     x_old = relu(W0 x^ + b0)                               [h]      a function of the direction alone
//     r_aug = relu(W1 x_old + b1 + W1_r r + b1_r)            [h]
//     out   = r (W2 x_old + b2 + W2_r r_aug + b2_r) - 1                W2_r >= 0, projected after every optimizer step
//     loss  = mean((sigmoid(out) - label)^2) over a minibatch of pixels; Adam over every parameter, `offset` from a given epoch on
//
// The read-out takes two layers at once (x_old AND r_aug) and the logit is scaled per point by r: neither is in the layer chain of the
// fused ICNN step kernels.  The notebook trains on 1000-pixel minibatches (500 background + 500 foreground pixels drawn per epoch),
// i.e. 1000 x h activations per layer - a launch-bound problem, not an MFMA-bound one.  It therefore runs layer by layer like wide.h:
// activations of the minibatch in HBM ([batch][h] row-major), the three h x h contractions as the fp32-MFMA GEMMs of csrc/gemm.h (the minibatch contraction in slices added in order), and
// everything else hand-written below: the polar split with layer 0, the epilogues, read-out + sigmoid + MSE + dL/dout, the relu masks
// of the backward pass, fixed-order column and point reductions, the gradient of the centre, torch's single-tensor Adam with the
// projection.  Twelve launches per optimizer step, none of them synchronises with the host; the minibatch indices of every epoch are an
// input ([steps][batch], drawn on the host side of the boundary like the notebook's torch.randperm).
//
// Parameters: ONE flat vector in the order of the notebook class's named_parameters():
//     offset[2] | W0.weight[h][2] | W0.bias[h] | W1.weight[h][h] | W1.bias[h] | W2.weight[h] | W2.bias | W1_r.weight[h] | W1_r.bias[h]
//     | W2_r.weight[h] | W2_r.bias                                                     P = h^2 + 8 h + 4
#pragma once
#include "wide.h"

namespace {

struct StarMap {
    int h, P;
    __host__ __device__ int p_off() const { return 0; }
    __host__ __device__ int p_w0() const { return 2; }
    __host__ __device__ int p_b0() const { return 2 + 2 * h; }
    __host__ __device__ int p_w1() const { return 2 + 3 * h; }
    __host__ __device__ int p_b1() const { return p_w1() + h * h; }
    __host__ __device__ int p_w2() const { return p_b1() + h; }
    __host__ __device__ int p_b2() const { return p_w2() + h; }
    __host__ __device__ int p_w1r() const { return p_b2() + 1; }
    __host__ __device__ int p_b1r() const { return p_w1r() + h; }
    __host__ __device__ int p_w2r() const { return p_b1r() + h; }
    __host__ __device__ int p_b2r() const { return p_w2r() + h; }
};
inline StarMap make_star_map(int h) {
    StarMap m{h, 0};
    m.P = m.p_b2r() + 1;
    return m;
}

constexpr int STAR_FEAT = 8;        // per point: x^0, x^1, r, x'0, x'1, 1 / (0.01 + r), -, -
constexpr int STAR_COLQ = 7;        // column sums per hidden unit: db1(=db1_r), dW1_r, dW2, dW2_r, db0, dW0[.][0], dW0[.][1]
constexpr int STAR_CHUNKS = 16;     // point chunks of the column sums (added in chunk order by the update kernel)

__device__ __forceinline__ float wave_sum(float v) {   // fixed butterfly order: reproducible
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// pixel number of minibatch entry p, clamped into the image (an index outside it must not become a fault)
__device__ __forceinline__ long long star_pixel(const int32_t* __restrict__ idx, const long long p, const long long n_pixels) {
    if (idx == nullptr) return p;
    const long long v = idx[p];
    return v < 0 ? 0 : (v >= n_pixels ? n_pixels - 1 : v);
}

// polar split + layer 0: thread = (point, unit).  idx: this epoch's minibatch (pixel numbers into coords / labels), or null = all points.
__global__ __launch_bounds__(256) void star_l0_kernel(const float* __restrict__ prm, const StarMap m, const float* __restrict__ coords,
                                                      const int32_t* __restrict__ idx, const long long n_pixels, const long long N,
                                                      float* __restrict__ feat, float* __restrict__ A0) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= N * m.h) return;
    const long long p = e / m.h;
    const int j = (int)(e - p * m.h);
    const long long pix = star_pixel(idx, p, n_pixels);
    const float x0 = coords[2 * pix] + prm[0], x1 = coords[2 * pix + 1] + prm[1];
    const float r = __fsqrt_rn(__fadd_rn(__fmul_rn(x0, x0), __fmul_rn(x1, x1)));
    const float den = __fadd_rn(0.01f, r);
    const float u0 = __fdiv_rn(x0, den), u1 = __fdiv_rn(x1, den);
    if (j == 0) {
        float* f = feat + p * STAR_FEAT;
        f[0] = u0; f[1] = u1; f[2] = r; f[3] = x0; f[4] = x1; f[5] = __fdiv_rn(1.f, den);
    }
    const float pre = fmaf(prm[m.p_w0() + 2 * j + 1], u1, fmaf(prm[m.p_w0() + 2 * j], u0, prm[m.p_b0() + j]));
    A0[e] = fmaxf(pre, 0.f);
}

// r_aug = relu((P1 + b1) + (W1_r r + b1_r)), in place over the GEMM's output
__global__ __launch_bounds__(256) void star_l1_kernel(const float* __restrict__ prm, const StarMap m, const float* __restrict__ feat,
                                                      const long long N, float* __restrict__ A1) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= N * m.h) return;
    const long long p = e / m.h;
    const int j = (int)(e - p * m.h);
    const float a = __fadd_rn(A1[e], prm[m.p_b1() + j]);
    const float b = fmaf(prm[m.p_w1r() + j], feat[p * STAR_FEAT + 2], prm[m.p_b1r() + j]);
    A1[e] = fmaxf(__fadd_rn(a, b), 0.f);
}

// read-out, sigmoid, squared error and dL/dout: one wave per point, four points per block.
// labels == null: forward only (logits out).  partial: [blocks][2] = (sum of squared errors, sum of ds) of the block's points.
__global__ __launch_bounds__(256) void star_out_kernel(const float* __restrict__ prm, const StarMap m, const float* __restrict__ feat,
                                                       const float* __restrict__ A0, const float* __restrict__ A1,
                                                       const float* __restrict__ labels, const int32_t* __restrict__ idx,
                                                       const long long n_pixels, const long long N, float* __restrict__ logits, float* __restrict__ ds, float* __restrict__ drd,
                                                       float* __restrict__ partial) {
    __shared__ float sh[4][2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long p = (long long)blockIdx.x * 4 + wv;
    float se = 0.f, sds = 0.f;
    if (p < N) {
        float sa = 0.f, sr = 0.f;
        for (int j = lane; j < m.h; j += 64) {
            sa = fmaf(prm[m.p_w2() + j], A0[p * m.h + j], sa);
            sr = fmaf(prm[m.p_w2r() + j], A1[p * m.h + j], sr);
        }
        sa = wave_sum(sa);
        sr = wave_sum(sr);
        const float r = feat[p * STAR_FEAT + 2];
        const float s = __fadd_rn(__fadd_rn(sa, prm[m.p_b2()]), __fadd_rn(sr, prm[m.p_b2r()]));
        const float out = __fsub_rn(__fmul_rn(r, s), 1.f);
        if (lane == 0) {
            if (logits) logits[p] = out;
            if (labels) {
                const float pr = 1.f / (1.f + expf(-out));
                const float er = pr - labels[star_pixel(idx, p, n_pixels)];
                const float dout = 2.f * er / (float)N * pr * (1.f - pr);
                ds[p] = dout * r;
                drd[p] = dout * s;
                se = er * er;
                sds = dout * r;
            }
        }
    }
    if (partial == nullptr) return;
    if (lane == 0) { sh[wv][0] = se; sh[wv][1] = sds; }
    __syncthreads();
    if (threadIdx.x < 2) partial[2 * blockIdx.x + threadIdx.x] = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
}

// D1 = ds W2_r (.) [r_aug > 0]
__global__ __launch_bounds__(256) void star_bwd1_kernel(const float* __restrict__ prm, const StarMap m, const float* __restrict__ ds,
                                                        const float* __restrict__ A1, const long long N, float* __restrict__ D1) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= N * m.h) return;
    const long long p = e / m.h;
    const int j = (int)(e - p * m.h);
    D1[e] = A1[e] > 0.f ? ds[p] * prm[m.p_w2r() + j] : 0.f;
}

// D0 = (D1 W1 + ds W2) (.) [x_old > 0], in place over the GEMM's output
__global__ __launch_bounds__(256) void star_bwd0_kernel(const float* __restrict__ prm, const StarMap m, const float* __restrict__ ds,
                                                        const float* __restrict__ A0, const long long N, float* __restrict__ D0) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= N * m.h) return;
    const long long p = e / m.h;
    const int j = (int)(e - p * m.h);
    D0[e] = A0[e] > 0.f ? fmaf(ds[p], prm[m.p_w2() + j], D0[e]) : 0.f;
}

// Column sums over the points of one chunk: grid (ceil(h / 32), STAR_CHUNKS), thread = (unit jj of 32, point group pg of 8).
// colp[chunk][q][h]; fixed order: a thread walks its points in order, the 8 groups are added in order.
__global__ __launch_bounds__(256) void star_col_kernel(const StarMap m, const float* __restrict__ feat, const float* __restrict__ ds,
                                                       const float* __restrict__ A0, const float* __restrict__ A1,
                                                       const float* __restrict__ D0, const float* __restrict__ D1, const long long N,
                                                       float* __restrict__ colp) {
    __shared__ float sh[8][STAR_COLQ][32];
    const int jj = threadIdx.x & 31, pg = threadIdx.x >> 5, j = blockIdx.x * 32 + jj, c = blockIdx.y;
    const long long cl = (N + STAR_CHUNKS - 1) / STAR_CHUNKS, p0 = c * cl, p1 = p0 + cl < N ? p0 + cl : N;
    float a[STAR_COLQ];
#pragma unroll
    for (int q = 0; q < STAR_COLQ; ++q) a[q] = 0.f;
    if (j < m.h) {
        for (long long p = p0 + pg; p < p1; p += 8) {
            const float d1 = D1[p * m.h + j], d0 = D0[p * m.h + j], s = ds[p];
            const float* f = feat + p * STAR_FEAT;
            a[0] += d1;
            a[1] = fmaf(d1, f[2], a[1]);
            a[2] = fmaf(s, A0[p * m.h + j], a[2]);
            a[3] = fmaf(s, A1[p * m.h + j], a[3]);
            a[4] += d0;
            a[5] = fmaf(d0, f[0], a[5]);
            a[6] = fmaf(d0, f[1], a[6]);
        }
    }
#pragma unroll
    for (int q = 0; q < STAR_COLQ; ++q) sh[pg][q][jj] = a[q];
    __syncthreads();
    for (int t = threadIdx.x; t < STAR_COLQ * 32; t += 256) {
        const int q = t >> 5, k = t & 31;
        if (blockIdx.x * 32 + k >= m.h) continue;
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) v += sh[g][q][k];
        colp[((size_t)c * STAR_COLQ + q) * m.h + blockIdx.x * 32 + k] = v;
    }
}

// Gradient of the centre: per point dx^ = D0 W0, dr = dout s + D1 . W1_r, chained through x^ = x' / (0.01 + r), r = |x'|
// (a pixel exactly at -offset has r = 0 and a NaN gradient, in torch as here).  One wave per point; offp[blocks][2].
__global__ __launch_bounds__(256) void star_point_kernel(const float* __restrict__ prm, const StarMap m, const float* __restrict__ feat,
                                                         const float* __restrict__ drd, const float* __restrict__ D0,
                                                         const float* __restrict__ D1, const long long N, float* __restrict__ offp) {
    __shared__ float sh[4][2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long p = (long long)blockIdx.x * 4 + wv;
    float g0 = 0.f, g1 = 0.f;
    if (p < N) {
        float du0 = 0.f, du1 = 0.f, drh = 0.f;
        for (int j = lane; j < m.h; j += 64) {
            const float d0 = D0[p * m.h + j];
            du0 = fmaf(d0, prm[m.p_w0() + 2 * j], du0);
            du1 = fmaf(d0, prm[m.p_w0() + 2 * j + 1], du1);
            drh = fmaf(D1[p * m.h + j], prm[m.p_w1r() + j], drh);
        }
        du0 = wave_sum(du0);
        du1 = wave_sum(du1);
        drh = wave_sum(drh);
        const float* f = feat + p * STAR_FEAT;
        const float r = f[2], x0 = f[3], x1 = f[4], q = f[5];
        const float dden = -(du0 * x0 + du1 * x1) * q * q;     // d / d(0.01 + r) of x' / (0.01 + r)
        const float dr = drd[p] + drh + dden;
        g0 = du0 * q + dr * x0 / r;                            // r = sqrt(x'0^2 + x'1^2): dr / dx' = x' / r
        g1 = du1 * q + dr * x1 / r;
    }
    if (lane == 0) { sh[wv][0] = g0; sh[wv][1] = g1; }
    __syncthreads();
    if (threadIdx.x < 2) offp[2 * blockIdx.x + threadIdx.x] = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
}

// One block: the per-block partials of the read-out (loss, sum ds) and of the centre's gradient, each in a fixed order.
// scal[0] = loss, scal[1] = db2 (= db2_r), scal[2..3] = doffset
__global__ __launch_bounds__(256) void star_scalar_kernel(const float* __restrict__ partial, const float* __restrict__ offp, const int nb,
                                                          const long long N, float* __restrict__ scal, float* __restrict__ loss_hist,
                                                          const int hist_idx) {
    __shared__ float sm[4];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < nb; b += 256) {
        v[0] += partial[2 * b];
        v[1] += partial[2 * b + 1];
        v[2] += offp[2 * b];
        v[3] += offp[2 * b + 1];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float t = block_sum256(v[k], sm);
        if (threadIdx.x == 0) {
            const float o = k == 0 ? t / (float)N : t;
            scal[k] = o;
            if (k == 0 && loss_hist) loss_hist[hist_idx] = o;
        }
    }
}

struct StarUpdArgs {
    float* prm;          // [P] in place
    float* opt;          // [2 P] exp_avg | exp_avg_sq
    const float* gW1;    // [h][h] from the GEMM
    const float* colp;   // [STAR_CHUNKS][STAR_COLQ][h]
    const float* scal;   // loss, db2, doffset[2]
    float* grads_out;    // [P] or null: the assembled gradient (mode 1: no optimizer step)
    StarMap m;
    float lr, beta1, beta2, eps, one_minus_b1, one_minus_b2;
    double bc1, bc1_off;             // 1 - beta1^t for the network / for the centre (its own step count)
    float bc2_sqrt, bc2_sqrt_off;
    int offset_on;                   // the centre takes this step
    int mode;                        // 0 = Adam + projection, 1 = gradients only
};

// torch.optim.Adam, single-tensor arithmetic in torch's operation order (as icnn_update_kernel), then W2_r.weight <- relu(W2_r.weight)
__global__ __launch_bounds__(256) void star_update_kernel(const StarUpdArgs u) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const StarMap& m = u.m;
    if (i >= m.P) return;
    const int h = m.h;
    float g;
    int colq = -1, j = 0;
    if (i < m.p_w0()) g = u.scal[2 + i];
    else if (i < m.p_b0()) { j = (i - m.p_w0()) >> 1; colq = 5 + ((i - m.p_w0()) & 1); }
    else if (i < m.p_w1()) { j = i - m.p_b0(); colq = 4; }
    else if (i < m.p_b1()) g = u.gW1[i - m.p_w1()];
    else if (i < m.p_w2()) { j = i - m.p_b1(); colq = 0; }
    else if (i < m.p_b2()) { j = i - m.p_w2(); colq = 2; }
    else if (i < m.p_w1r()) g = u.scal[1];
    else if (i < m.p_b1r()) { j = i - m.p_w1r(); colq = 1; }
    else if (i < m.p_w2r()) { j = i - m.p_b1r(); colq = 0; }
    else if (i < m.p_b2r()) { j = i - m.p_w2r(); colq = 3; }
    else g = u.scal[1];
    if (colq >= 0) {
        g = 0.f;
#pragma unroll
        for (int c = 0; c < STAR_CHUNKS; ++c) g += u.colp[((size_t)c * STAR_COLQ + colq) * h + j];
    }
    if (u.grads_out) u.grads_out[i] = g;
    if (u.mode != 0) return;
    const bool is_off = i < m.p_w0();
    if (is_off && !u.offset_on) return;     // requires_grad = False: torch's Adam skips a parameter without a gradient
    float p = u.prm[i], mm = u.opt[i], v = u.opt[m.P + i];
    mm = __fadd_rn(mm, __fmul_rn(u.one_minus_b1, __fsub_rn(g, mm)));
    v = __fadd_rn(__fmul_rn(v, u.beta2), __fmul_rn(__fmul_rn(u.one_minus_b2, g), g));
    const float step_size = (float)((double)u.lr / (is_off ? u.bc1_off : u.bc1));
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), is_off ? u.bc2_sqrt_off : u.bc2_sqrt), u.eps);
    p = __fadd_rn(p, __fdiv_rn(__fmul_rn(-step_size, mm), denom));
    if (i >= m.p_w2r() && i < m.p_b2r()) p = fmaxf(p, 0.f);
    u.prm[i] = p;
    u.opt[i] = mm;
    u.opt[m.P + i] = v;
}

// ---- workspace ---------------------------------------------------------------------------------------------------------------------------
struct StarWs {
    float *feat, *A0, *A1, *D1, *D0, *ds, *drd, *partial, *offp, *colp, *gW1, *gpart, *scal;
    int nb4;           // blocks of the one-wave-per-point kernels
    long long bytes;
};
inline StarWs carve_star(int h, long long N, void* base) {
    StarWs w{};
    char* b = (char*)base;
    long long off = 0;
    auto take = [&](long long bytes) {
        float* p = (float*)(b + off);
        off += (bytes + 255) / 256 * 256;
        return p;
    };
    w.nb4 = (int)((N + 3) / 4);
    w.feat = take(N * STAR_FEAT * 4);
    w.A0 = take(N * h * 4);
    w.A1 = take(N * h * 4);
    w.D1 = take(N * h * 4);
    w.D0 = take(N * h * 4);
    w.ds = take(N * 4);
    w.drd = take(N * 4);
    w.partial = take((long long)w.nb4 * 2 * 4);
    w.offp = take((long long)w.nb4 * 2 * 4);
    w.colp = take((long long)STAR_CHUNKS * STAR_COLQ * h * 4);
    w.gW1 = take((long long)h * h * 4);
    w.gpart = take((long long)GEMM_SPLITK_MAX * h * h * 4);   // slices of the minibatch contraction (gemm_rm_splitk)
    w.scal = take(256);
    w.bytes = off;
    return w;
}

constexpr int STAR_MAX_HIDDEN = 1024;
constexpr long long STAR_MAX_BATCH = 1 << 16;   // the weight-gradient GEMM contracts over the minibatch in one call

inline dim3 star_elem_grid(long long N, int h) { return dim3((unsigned)((N * h + 255) / 256)); }

// forward of N points (idx: minibatch or null); labels != null also leaves dL/dout behind for the backward pass
inline int star_forward_pass(const StarMap& m, const StarWs& w, const float* prm, const float* coords, const int32_t* idx,
                             long long n_pixels, long long N, const float* labels, float* logits, hipStream_t s) {
    const int h = m.h;
    hipLaunchKernelGGL(star_l0_kernel, star_elem_grid(N, h), dim3(256), 0, s, prm, m, coords, idx, n_pixels, N, w.feat, w.A0);
    int rc = gemm_rm(s, false, true, (int)N, h, h, w.A0, h, prm + m.p_w1(), h, w.A1, h);      // P1 = x_old W1^T
    if (rc) return rc;
    hipLaunchKernelGGL(star_l1_kernel, star_elem_grid(N, h), dim3(256), 0, s, prm, m, (const float*)w.feat, N, w.A1);
    hipLaunchKernelGGL(star_out_kernel, dim3(w.nb4), dim3(256), 0, s, prm, m, (const float*)w.feat, (const float*)w.A0, (const float*)w.A1,
                       labels, idx, n_pixels, N, logits, w.ds, w.drd, labels ? w.partial : nullptr);
    return INR_OK;
}

// backward of the pass above: every gradient into gW1 / colp / scal (assembled by star_update_kernel)
inline int star_backward_pass(const StarMap& m, const StarWs& w, const float* prm, long long N, float* loss_hist, int hist_idx,
                              hipStream_t s) {
    const int h = m.h;
    hipLaunchKernelGGL(star_bwd1_kernel, star_elem_grid(N, h), dim3(256), 0, s, prm, m, (const float*)w.ds, (const float*)w.A1, N, w.D1);
    int rc = gemm_rm(s, false, false, (int)N, h, h, w.D1, h, prm + m.p_w1(), h, w.D0, h);     // D1 W1
    if (rc) return rc;
    hipLaunchKernelGGL(star_bwd0_kernel, star_elem_grid(N, h), dim3(256), 0, s, prm, m, (const float*)w.ds, (const float*)w.A0, N, w.D0);
    if ((rc = gemm_rm_splitk(s, true, false, h, h, (int)N, w.D1, h, w.A0, h, w.gW1, w.gpart))) return rc;   // dW1 = D1^T x_old, slices added in order
    hipLaunchKernelGGL(star_col_kernel, dim3((h + 31) / 32, STAR_CHUNKS), dim3(256), 0, s, m, (const float*)w.feat, (const float*)w.ds,
                       (const float*)w.A0, (const float*)w.A1, (const float*)w.D0, (const float*)w.D1, N, w.colp);
    hipLaunchKernelGGL(star_point_kernel, dim3(w.nb4), dim3(256), 0, s, prm, m, (const float*)w.feat, (const float*)w.drd,
                       (const float*)w.D0, (const float*)w.D1, N, w.offp);
    hipLaunchKernelGGL(star_scalar_kernel, dim3(1), dim3(256), 0, s, (const float*)w.partial, (const float*)w.offp, w.nb4, N, w.scal,
                       loss_hist, hist_idx);
    return INR_OK;
}

}  // namespace
