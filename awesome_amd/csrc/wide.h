// wide.h - the layer-by-layer path for ICNN shapes the fused step kernels cannot hold (n_hidden > 130, or more than two hidden layers).
//
// The fused kernels (icnn_step.h, icnn_step2.h) keep the whole weight image of the network in LDS and every activation in registers;
// that is what makes them fast and what limits them to n_hidden <= 130 (77 KB per hidden layer for h = 130; 160 KB of LDS per CU) and
// L <= 2.  A `n_hidden: 256` or the 3 x 350 relu net of notebooks/imageRepresentationTest.ipynb cell 5 needs another shape of program:
// activations of a whole image in HBM (N x h floats per layer: 92 MB at 256x256 x 350 - nothing on a 288 GB part), one launch per
// layer and direction, and the h x h contractions as tiled fp32-MFMA GEMMs - csrc/gemm.h, hand-written for gfx950 (round 4; rounds
// 2-3 called rocBLAS here).  What used to be separate element-wise passes over the activations rides in the GEMMs' epilogues (bias +
// skip + relu of a hidden layer; the relu / periodic-activation mask of the backward pass) or in the one pass over Z_L that computes
// the output layer, the data term, dL/dlogit, dZ_L and the output layer's weight gradients (wide_out_kernel); the gradients of what the
// ext inputs (1, x) multiply - (db_k | dS_k), (db_in | dW_in) = dz^T (1, X) - are summed by the kernel that writes dz (wide_out_kernel
// for the last layer, the backward GEMM's epilogue for the others).  The contractions over the
// points (weight gradients, K = n_points) are split over blockIdx.z into chunks of WIDE_CHUNK points whose partial products
// wide_reduce_kernel adds in chunk order: no atomics, reproducible.  The optimizer step is icnn_update_kernel itself on a one-"slab"
// view of the gradient vector.
//
// Same arithmetic as the reference (awesome/model/convex_net.py:205-214): z0 = act0(W_in x + b_in); z_{k+1} = relu(W_k z_k + b_k + S_k x);
// y = w_o . z_L + b_o + s_o . x.  Layout: activations row-major [N][hs] (a point's h units, then the "ext" inputs (1, x), then zeros up
// to a multiple of 4 floats so that rows start on 16-byte boundaries), parameters in the flat order of include/inrfit.h (torch's
// row-major [out][in] Linear weights); the hidden layers' weight matrices are copied once per step into [L][h][hp] with zero padding
// (0.25 - 1.5 MB, by the last blocks of the layer-0 launch) so that every GEMM operand loads 16 bytes per lane.
#pragma once
#include "gemm.h"

namespace {

constexpr int WIDE_CHUNK = 512;      // points per split of the point-contractions (GEMM partials): at least; see wide_chunk
#ifndef INR_WIDE_OUT_CHUNK
#define INR_WIDE_OUT_CHUNK 128
#endif
constexpr int WIDE_OUT_CHUNK = INR_WIDE_OUT_CHUNK;  // points per block of wide_out_kernel / wide_extgrad_kernel (their partials)
// a weight gradient's contraction over the points: 512 points per slice up to 3 x 3 tiles of 128, 1024 from 4 x 4 on (the launch keeps >= 1024
// workgroups at 65 536 points; half the partial-product bytes to write and to add up: h = 512 L = 2 1 879 -> 1 849 us; at 3 x 3 tiles the 576
// workgroups of 1024-point slices were 1 % slower)
inline int wide_chunk(int h) {
    const int t = (h + GM_BM - 1) / GM_BM;
    return t >= 4 ? 2 * WIDE_CHUNK : WIDE_CHUNK;
}
inline int splitk_parts(long long N, int h) { return (int)((N + wide_chunk(h) - 1) / wide_chunk(h)); }
inline int out_parts(long long N) { return (int)((N + WIDE_OUT_CHUNK - 1) / WIDE_OUT_CHUNK); }

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
constexpr int WIDE_MAX_HIDDEN_PAD = (WIDE_MAX_HIDDEN + 1 + 3 + 3) / 4 * 4;   // longest activation row (h + 1 + C, padded to 4 floats)

// ---- kernels ---------------------------------------------------------------------------------------------------------------------------
// Activations are stored [N][hs] with hs = h + 1 + C: the h units of the layer, then the "ext" inputs (1, x_0 .. x_{C-1}) - as in the
// fused kernels, bias and skip weights ride in the contractions: dW_ext [h x hs] = dz^T Z_ext holds dW, db and dS of a layer at once.

// z0[p][j] = act0(W_in[j] . x_p + b_in[j]) for j < h, the ext columns (1, x) for h <= j < h + 1 + C, zeros in the padding; pre0
// (optional, [N][hp]) keeps the pre-activation for the periodic activations' derivative.  Reads the coordinates from the grid descriptor.
constexpr int WIDE_L0_POINTS = 64, WIDE_L0_REP = 4;
template <int C>
__global__ __launch_bounds__(256) void wide_layer0_kernel(InrGridDesc gd, int img, const float* __restrict__ win, const float* __restrict__ bin,
                                                          long long N, int h, int hs, int hp, int act0, float omega, float* __restrict__ z0,
                                                          float* __restrict__ pre0, int pt_blocks, const float* __restrict__ params, WideMap pk,
                                                          float* __restrict__ wp) {
    // a thread writes four consecutive columns (hs is a multiple of 4: rows are cut into hs / 4 quads) of WIDE_L0_REP points'
    // rows, WIDE_L0_POINTS apart: its layer-0 weights are read once, the points' coordinates are all requested before the first
    // row is computed (32-bit index arithmetic inside the block)
    const int nq = hs >> 2;
    if ((int)blockIdx.x >= pt_blocks) {   // the launch's last blocks: hidden-layer weights -> [L][hq][hq], hq = h rounded up to 16, zero padded:
                                          // whatever a k-step reads past h of the other operand meets a zero here (gemm.h, GemmArgs::buf)
        const int hq = (pk.h + 15) / 16 * 16, per = hq * hq, e = (((int)blockIdx.x - pt_blocks) * gridDim.y + blockIdx.y) * 256 + threadIdx.x;
        if (e < pk.L * per) {
            const int k = e / per, r = e - k * per, i = r / hq, j = r - i * hq;
            wp[e] = i < pk.h && j < pk.h ? params[pk.p_w(k) + i * pk.h + j] : 0.f;
        }
        return;
    }
    const int idx = blockIdx.y * 256 + threadIdx.x;
    if (idx >= WIDE_L0_POINTS * nq) return;
    const int pl = idx / nq;
    const int j0 = 4 * (idx - pl * nq);
    const long long pb = (long long)blockIdx.x * (WIDE_L0_POINTS * WIDE_L0_REP) + pl;
    float x[WIDE_L0_REP][C];
#pragma unroll
    for (int r = 0; r < WIDE_L0_REP; ++r) {
        const long long pv = pb + r * WIDE_L0_POINTS, p = pv < N ? pv : N - 1;
        if (gd.mode == INR_GRID_SEPARABLE) {
            const unsigned pu = (unsigned)p, row = pu / (unsigned)gd.width;   // (an image of this path has < 2^32 points: 1 KB of activations each)
            x[r][0] = gd.xs[pu - row * (unsigned)gd.width];
            x[r][1] = gd.ys[row];
            if (C > 2) x[r][C - 1] = gd.ts ? gd.ts[img] : 0.f;
        } else {
            const float* cp = gd.coords + (size_t)img * gd.coords_image_stride;
#pragma unroll
            for (int c = 0; c < C; ++c) x[r][c] = cp[(size_t)c * N + p];
        }
    }
    float bq[4], wq[4][C];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = j0 + q < h ? j0 + q : h - 1;
        bq[q] = bin[j];
#pragma unroll
        for (int c = 0; c < C; ++c) wq[q][c] = win[j * C + c];
    }
#pragma unroll
    for (int r = 0; r < WIDE_L0_REP; ++r) {
        const long long p = pb + r * WIDE_L0_POINTS;
        if (p >= N) break;
        f32x4 out;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + q;
            float v;
            if (j >= h) {
                v = j == h ? 1.f : 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) v = (j == h + 1 + c) ? x[r][c] : v;
            } else {
                v = bq[q];
#pragma unroll
                for (int c = 0; c < C; ++c) v = fmaf(wq[q][c], x[r][c], v);
                if (pre0) pre0[p * hp + j] = v;
                v = act0 == INR_ACT_COS ? hw_cos(v) : (act0 == INR_ACT_SIN ? hw_sin(omega * v) : fmaxf(v, 0.f));
            }
            out[q] = v;
        }
        *(f32x4*)(z0 + (size_t)p * hs + j0) = out;
    }
}

// ONE pass over the last layer's activations Z_L [N][hs] (16 lanes per point, 16 points per iteration, WIDE_OUT_CHUNK points per block):
//   y = w_o . z_L + b_o + s_o . x  -> logits;   TRAIN: sigmoid, data term -> dy, the loss partial of the block;
//   dZ_L[p][j] = dy w_o[j] [z_L[p][j] > 0]  (the row is still in the cache);
//   and the block's share of the output layer's gradients (dw_o | db_o | ds_o)[j] = sum_p dy[p] Z_L,ext[p][j], summed over the block's
//   points in a fixed order -> part[block][hs_valid] (wide_reduce_kernel adds the blocks in order).
struct WideOutArgs {
    const float* zl;        // [N][hs]
    const float* wo;        // [h]
    const float* sc;        // b_o, s_o[C]  (flat parameters at p_bo)
    const float* target;    // [N] or null
    const float* coef;      // c_fg, c_bg of this image
    float* logits;          // [N] or null
    float* dz;              // [N][hp] (train)
    float* part;            // [blocks][hsv + 1] output-layer gradient partials, then the block's loss partial (train)
    float* part_ext;        // [blocks][h][1 + C] partials of (db | dS) of the last hidden layer = dZ_L^T (1, X) (train, EXT instantiations)
    long long N;
    int h, C, hs, hp, hsv, loss_kind, train;
};
constexpr int WIDE_OUT_MAXQ = (WIDE_MAX_HIDDEN_PAD + 63) / 64;   // f32x4 per lane and row
template <int WIDE_OUT_NQ, bool EXT>   // 64-column slices a row may have (register budget of the instantiation); EXT: also dZ_L^T (1, X)
__global__ __launch_bounds__(256) void wide_out_kernel(const WideOutArgs a) {
    __shared__ float sm[4];
    __shared__ float colw[4][5][64];        // per wave: the column sums of one 64-column slice (output layer; (1, x) sums)
    const int tid = threadIdx.x, l15 = tid & 15, rg = tid >> 4;      // 16 lanes per point, 16 row groups
    const long long p0 = (long long)blockIdx.x * WIDE_OUT_CHUNK;
    const int nq = (a.hs + 63) / 64;        // 64-column slices of a row (16 lanes x 4 floats)
    f32x4 gacc[WIDE_OUT_NQ];
    f32x4 eacc[EXT ? WIDE_OUT_NQ : 1][4];   // [slice][column of the lane's four] = sum_p dZ_L[p][j] (1, x_p)
#pragma unroll
    for (int q = 0; q < WIDE_OUT_NQ; ++q) gacc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < (EXT ? WIDE_OUT_NQ : 1); ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) eacc[q][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    float lsum = 0.f;
    // the lane's columns: 4 l15 .. + 3 of every 64-column slice; their output weights once (0 past h), the rows of the NEXT 16 points
    // requested (unguarded, from clamped addresses) before this iteration's arithmetic
    f32x4 wq[WIDE_OUT_NQ];
    bool cq[WIDE_OUT_NQ];
#pragma unroll
    for (int q = 0; q < WIDE_OUT_NQ; ++q) {
        const int j = 64 * q + 4 * l15;
        cq[q] = q < nq && j < a.hs;
#pragma unroll
        for (int e = 0; e < 4; ++e) wq[q][e] = j + e < a.h ? a.wo[j + e] : 0.f;
    }
    auto load_row = [&](int it, f32x4 (&z)[WIDE_OUT_NQ]) {
        const long long p = p0 + it * 16 + rg;
        const float* zr = a.zl + (size_t)(p < a.N ? p : 0) * a.hs;
#pragma unroll
        for (int q = 0; q < WIDE_OUT_NQ; ++q) z[q] = *(const f32x4*)(zr + (cq[q] ? 64 * q + 4 * l15 : 0));
    };
    f32x4 zn[WIDE_OUT_NQ];
    load_row(0, zn);
    for (int it = 0; it < WIDE_OUT_CHUNK / 16; ++it) {
        const long long p = p0 + it * 16 + rg;
        const bool valid = p < a.N;
        const float* zr = a.zl + (size_t)(valid ? p : 0) * a.hs;
        f32x4 zq[WIDE_OUT_NQ];
        float ypart = 0.f;
#pragma unroll
        for (int q = 0; q < WIDE_OUT_NQ; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                zq[q][e] = cq[q] ? zn[q][e] : 0.f;
                ypart = fmaf(wq[q][e], zq[q][e], ypart);
            }
        }
        if (it + 1 < WIDE_OUT_CHUNK / 16) load_row(it + 1, zn);
        ypart = sum_over_points(ypart);     // over the 16 lanes of the point (one DPP row)
        float y = ypart + a.sc[0];
        f32x4 xe = f32x4{1.f, 0.f, 0.f, 0.f};   // the point's ext inputs (1, x)
        for (int c = 0; c < a.C; ++c) {
            xe[1 + c] = zr[a.h + 1 + c];
            y = fmaf(a.sc[1 + c], xe[1 + c], y);
        }
        if (a.logits && valid && l15 == 0) a.logits[p] = y;
        if (!a.train) continue;
        float l = 0.f, dy = 0.f;
        if (valid) {
            const float tg = a.target[p];
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
        }
        if (l15 == 0) lsum += l;
        float* dr = a.dz + (size_t)(valid ? p : 0) * a.hp;
#pragma unroll
        for (int q = 0; q < WIDE_OUT_NQ; ++q) {
            const int j = 64 * q + 4 * l15;
            if (q < nq && j < a.hs) {
#pragma unroll
                for (int e = 0; e < 4; ++e) gacc[q][e] = fmaf(dy, zq[q][e], gacc[q][e]);   // (dy = 0 for invalid points)
                if (valid && j < a.hp) {
                    f32x4 d;
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] = zq[q][e] > 0.f ? dy * wq[q][e] : 0.f;   // (w_o = 0 past h)
                    *(f32x4*)(dr + j) = d;   // hp is a multiple of 4; the padding columns get zeros
                    if (EXT) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) eacc[q][e] += d[e] * xe;
                    }
                }
            }
        }
    }
    if (!a.train) return;
    {   // loss partial of the block: the 16 row groups in order
        const float v = sum_over_groups(sum_over_points(lsum));
        if ((tid & 63) == 0) sm[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) a.part[(size_t)blockIdx.x * (a.hsv + 1) + a.hsv] = ((sm[0] + sm[1]) + sm[2]) + sm[3];   // one more column of the partials
    }
    // column sums over the block's 16 row groups: the four groups of a wave by lane exchange (every lane group then holds the wave's
    // totals; group g keeps column 4 l15 + g), the four waves through LDS, added in order
    const int wv = tid >> 6, gl = (tid >> 4) & 3;
    for (int q = 0; q < nq; ++q) {
        f32x4 gs = f32x4{0.f, 0.f, 0.f, 0.f}, es[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) es[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int qq = 0; qq < WIDE_OUT_NQ; ++qq)
            if (qq == q) {
                gs = gacc[qq];
                if (EXT) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) es[e] = eacc[EXT ? qq : 0][e];
                }
            }
        float mine[5] = {0.f, 0.f, 0.f, 0.f, 0.f};   // column 4 l15 + gl: the output-layer sum, then the (1, x) sums
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = sum_over_groups(gs[e]);
            mine[0] = gl == e ? t : mine[0];
            if (EXT) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float u = sum_over_groups(es[e][c]);
                    mine[1 + c] = gl == e ? u : mine[1 + c];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < (EXT ? 5 : 1); ++c) colw[wv][c][4 * l15 + gl] = mine[c];
        __syncthreads();
        if (tid < 64) {
            const int j = 64 * q + tid;
            float t[5];
#pragma unroll
            for (int c = 0; c < (EXT ? 5 : 1); ++c) t[c] = ((colw[0][c][tid] + colw[1][c][tid]) + colw[2][c][tid]) + colw[3][c][tid];
            if (j < a.hsv) a.part[(size_t)blockIdx.x * (a.hsv + 1) + j] = t[0];
            if (EXT && j < a.h)
                for (int c = 0; c <= a.C; ++c) a.part_ext[((size_t)blockIdx.x * a.h + j) * (1 + a.C) + c] = t[1 + c];
        }
    }
}

// dL/dcoords (an ICNN behind a learned deformation: convex_diffeomorphism_net.py:170-177): every pre-activation gradient dz [N][hp] the
// backward pass produces contributes dz . Wx with Wx [h][C] = the weights the coordinates enter that layer with (S_k of a hidden layer,
// W_in of layer 0); the output layer contributes s_o dL/dlogit (`init`: dx is started from it).  One pass over dz per layer, 16 lanes per
// point; dx is planar [C][N] like the fused kernels' (icnn_step.h, DX).  Only the autograd bridge asks for it.
__global__ __launch_bounds__(256) void wide_dx_kernel(const float* __restrict__ dz, int hp, const float* __restrict__ wx, int h, int C, long long N,
                                                      const float* __restrict__ dlogits, const float* __restrict__ so, float* __restrict__ dx,
                                                      int init) {
    const int l15 = threadIdx.x & 15;
    const long long p = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool valid = p < N;
    const float* row = dz + (size_t)(valid ? p : 0) * hp;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int j = 4 * l15; j < h; j += 64) {
        const f32x4 d = *(const f32x4*)(row + j);      // hp is a multiple of 4; the padding columns hold zeros
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (j + e < h)
                for (int c = 0; c < C; ++c) acc[c] = fmaf(d[e], wx[(j + e) * C + c], acc[c]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] = sum_over_points(acc[c]);
    if (valid && l15 < C) {
        const float a = l15 == 0 ? acc[0] : (l15 == 1 ? acc[1] : acc[2]);
        float* o = dx + (size_t)l15 * N + p;
        *o = (init ? so[l15] * dlogits[p] : *o) + a;
    }
}

// gradients w.r.t. everything the "ext" inputs (1, x) multiply - (db_k | dS_k) of a hidden layer, (db_in | dW_in) of layer 0:
//   part[block][i][c] = sum over the block's WIDE_OUT_CHUNK points of dz[p][i] ext_c[p],  ext = (1, x_0 ..)
// A thread owns four consecutive units (one 16-byte load per point), 256 / (hp / 4) points are in flight per block and four loads per
// thread; the row groups are added in order through LDS.
__global__ __launch_bounds__(256) void wide_extgrad_kernel(const float* __restrict__ dz, int hp, const float* __restrict__ ext, int hs, long long N,
                                                           int h, int C, float* __restrict__ part) {
    __shared__ f32x4 red[256][4];
    const int nqc = hp / 4;                         // threads per point row
    const int RL = 256 / nqc > 0 ? 256 / nqc : 1;   // rows in flight (hp <= 1024: nqc <= 256)
    const int tid = threadIdx.x, q = tid % nqc, rl = tid / nqc;
    const long long p0 = (long long)blockIdx.x * WIDE_OUT_CHUNK;
    const long long p1 = p0 + WIDE_OUT_CHUNK < N ? p0 + WIDE_OUT_CHUNK : N;
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (rl < RL) {
#pragma unroll 4
        for (long long p = p0 + rl; p < p1; p += RL) {
            const f32x4 d = *(const f32x4*)(dz + (size_t)p * hp + 4 * q);
            const float* e = ext + (size_t)p * hs;
            acc[0] += d;
            for (int c = 0; c < C; ++c) acc[1 + c] += d * e[1 + c];
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) red[tid][c] = acc[c];
    __syncthreads();
    if (tid < nqc) {
        for (int c = 0; c <= C; ++c) {
            f32x4 t = red[tid][c];
            for (int r = 1; r < RL; ++r) t += red[r * nqc + tid][c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * tid + e;
                if (i < h) part[((size_t)blockIdx.x * h + i) * (1 + C) + c] = t[e];
            }
        }
    }
}

// sum of the split-K partials [parts][a][b] in chunk order, scattered into the flat gradient vector:
//   mode 0 (hidden layer k): row i = unit, column j < h -> W_k[i][j]; j == h -> b_k[i]; j > h -> S_k[i][j - h - 1]
//   mode 1 (layer 0, B = the ext columns):           column 0 -> b_in[i]; j >= 1 -> W_in[i][j - 1]
//   mode 3 (hidden layer k, the ext columns only):   column 0 -> b_k[i];  j >= 1 -> S_k[i][j - 1]
//   mode 2 (output layer, a = 1, b = hsv + 1):       column j < h -> w_o[j]; j == h -> b_o; h < j <= h + C -> s_o[j - h - 1]; the last -> the loss
// (one launch takes up to two such reductions - e.g. a layer's weight-gradient partials and the (db | dS) partials its backward GEMM
// left: the first j0.nblocks blocks serve the first)
struct WideRedJob {
    const float* part;
    int parts, a, b, mode, k, nblocks;
};
inline WideRedJob wide_red_job(const float* part, int parts, int a, int b, int mode, int k) {
    return WideRedJob{part, parts, a, b, mode, k, (a * b + 63) / 64};
}
__global__ __launch_bounds__(1024) void wide_reduce_kernel(const WideRedJob j0, const WideRedJob j1, WideMap m, float* __restrict__ grads) {
    // 64 consecutive elements per block, the partials in 16 contiguous ranges (one per wave) summed in order, ranges added in order
    __shared__ float sm[16][64];
    const bool first = (int)blockIdx.x < j0.nblocks;
    const WideRedJob& jb = first ? j0 : j1;
    const float* __restrict__ part = jb.part;
    const int parts = jb.parts, a = jb.a, b = jb.b, mode = jb.mode, k = jb.k;
    const int el = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const int e = ((int)blockIdx.x - (first ? 0 : j0.nblocks)) * 64 + el;
    const int per = (parts + 15) / 16, q0 = pg * per, q1 = q0 + per < parts ? q0 + per : parts;
    float v = 0.f;
    if (e < a * b) {
#pragma unroll 8
        for (int q = q0; q < q1; ++q) v += part[(size_t)q * a * b + e];
    }
    sm[pg][el] = v;
    __syncthreads();
    if (pg != 0 || e >= a * b) return;
    v = sm[0][el];
#pragma unroll
    for (int r = 1; r < 16; ++r) v += sm[r][el];
    const int i = e / b, j = e - i * b;
    int dst;
    if (mode == 0) dst = j < m.h ? m.p_w(k) + i * m.h + j : (j == m.h ? m.p_b(k) + i : m.p_s(k) + i * m.C + (j - m.h - 1));
    else if (mode == 3) dst = j == 0 ? m.p_b(k) + i : m.p_s(k) + i * m.C + (j - 1);
    else if (mode == 1) dst = j == 0 ? m.p_bin() + i : m.p_win() + i * m.C + (j - 1);
    else dst = j < m.h ? m.p_wo() + j : (j == m.h ? m.p_bo() : (j <= m.h + m.C ? m.p_so() + (j - m.h - 1) : m.P));   // (last column: the loss)
    grads[dst] = v;
}
inline void wide_reduce(hipStream_t s, const WideMap& m, float* grads, WideRedJob j0, WideRedJob j1 = WideRedJob{nullptr, 0, 0, 0, 0, 0, 0}) {
    hipLaunchKernelGGL(wide_reduce_kernel, dim3((unsigned)(j0.nblocks + j1.nblocks)), dim3(1024), 0, s, j0, j1, m, grads);
}

// ---- workspace --------------------------------------------------------------------------------------------------------------------------
struct WideWs {
    float *z[WIDE_MAX_LAYERS + 1], *pre0, *dza, *dzb, *part, *part2, *grads, *coef, *wp;
    int blocks;            // blocks of wide_out_kernel / wide_l0grad_kernel (WIDE_OUT_CHUNK points each)
    int hs, hp, hsv;       // row length of the activations (multiple of 4), of the dz / pre0 buffers (multiple of 4), h + 1 + C
    long long bytes;
};

inline long long wide_align(long long b) { return (b + 255) / 256 * 256; }

inline WideWs carve_wide(const WideMap& m, long long N, bool need_pre0, void* base) {
    WideWs w{};
    char* b = (char*)base;
    long long off = 0;
    auto take = [&](long long bytes) { float* p = (float*)(b + off); off += wide_align(bytes); return p; };
    w.blocks = out_parts(N);
    w.hsv = m.h + 1 + m.C;
    w.hs = (w.hsv + 3) / 4 * 4;
    w.hp = (m.h + 3) / 4 * 4;
    for (int k = 0; k <= m.L; ++k) w.z[k] = take(N * w.hs * 4);
    w.pre0 = need_pre0 ? take(N * w.hp * 4) : nullptr;
    w.dza = take(N * w.hp * 4);
    w.dzb = take(N * w.hp * 4);
    const long long part_gemm = (long long)splitk_parts(N, m.h) * m.h * m.h, part_out = (long long)w.blocks * (w.hsv + 1 > m.h * (1 + m.C) ? w.hsv + 1 : m.h * (1 + m.C));
    w.part = take((part_gemm > part_out ? part_gemm : part_out) * 4);
    {   // (db | dS) partials: one [h][1 + C] block per wide_out_kernel block / per 128-row tile of the backward GEMM
        const long long tiles = (N + GM_BM - 1) / GM_BM;
        w.part2 = take((w.blocks > tiles ? w.blocks : tiles) * m.h * (1 + m.C) * 4);
    }
    w.grads = take(((long long)m.P + 1 + 31) / 32 * 32 * 4);
    {
        const long long hq = (m.h + 15) / 16 * 16;
        w.wp = take((long long)m.L * hq * hq * 4);
    }
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

// rows short enough for the wide_out_kernel instantiations that also sum dZ_L^T (1, X) (16 more accumulators per 64-column slice)
inline bool wide_out_has_ext(int hs) { return hs <= 9 * 64; }

// forward of ONE image; with `train`: also dZ_L (w.dza), the output layer's gradients and the loss (w.grads)
inline int wide_forward(const WideMap& m, const WideWs& w, const InrModelDesc* md, const float* params, const InrGridDesc* grid, int img,
                        const float* target, int loss_kind, bool train, float* logits, hipStream_t s) {
    const long long N = grid->n_points;
    const int h = m.h, C = m.C, hs = w.hs;
    // layer 0 (+ in the launch's last blocks: the padded copy of the hidden layers' weights)
    const int pt_blocks = (int)((N + WIDE_L0_POINTS * WIDE_L0_REP - 1) / (WIDE_L0_POINTS * WIDE_L0_REP));
    const unsigned l0y = (unsigned)((WIDE_L0_POINTS * (hs / 4) + 255) / 256);
    const int hq = (h + 15) / 16 * 16;   // row / column count of a packed weight matrix
    const int pk_blocks = (int)(((long long)m.L * hq * hq + 256ll * l0y - 1) / (256ll * l0y));
    const dim3 l0grid((unsigned)(pt_blocks + pk_blocks), l0y);
    if (C == 2) hipLaunchKernelGGL(wide_layer0_kernel<2>, l0grid, dim3(256), 0, s, *grid, img, params + m.p_win(), params + m.p_bin(), N, h, hs, w.hp, md->act0, md->act_omega, w.z[0], w.pre0, pt_blocks, params, m, w.wp);
    else hipLaunchKernelGGL(wide_layer0_kernel<3>, l0grid, dim3(256), 0, s, *grid, img, params + m.p_win(), params + m.p_bin(), N, h, hs, w.hp, md->act0, md->act_omega, w.z[0], w.pre0, pt_blocks, params, m, w.wp);
    for (int k = 0; k < m.L; ++k) {
        // z_{k+1} [N x h] = relu(z_k [N x h] . W_k^T + b_k + S_k x)   (W_k stored [h_out][h_in]; bias, skip and relu in the GEMM's epilogue)
        GemmArgs g{};
        g.A = w.z[k]; g.lda = hs; g.B = w.wp + (size_t)k * hq * hq; g.ldb = hq; g.C = w.z[k + 1]; g.ldc = hs;
        g.M = (int)N; g.N = h; g.K = h; g.padA = g.padB = 1; g.buf = 1;
        g.epi = GEMM_EPI_HIDDEN; g.bias = params + m.p_b(k); g.skip = params + m.p_s(k); g.ext = w.z[k] + h; g.ext_ld = hs; g.C_in = C; g.ext_copy = hs - h;
        int rc = gemm_launch(s, false, true, g);
        if (rc) return rc;
    }
    WideOutArgs a{};
    a.zl = w.z[m.L]; a.wo = params + m.p_wo(); a.sc = params + m.p_bo(); a.target = target; a.coef = w.coef; a.logits = logits;
    a.dz = w.dza; a.part = w.part;
    a.N = N; a.h = h; a.C = C; a.hs = hs; a.hp = w.hp; a.hsv = w.hsv; a.loss_kind = loss_kind; a.train = train ? 1 : 0;
    a.part_ext = w.part2;
    if (hs <= 5 * 64) hipLaunchKernelGGL((wide_out_kernel<5, true>), dim3(w.blocks), dim3(256), 0, s, a);
    else if (hs <= 6 * 64) hipLaunchKernelGGL((wide_out_kernel<6, true>), dim3(w.blocks), dim3(256), 0, s, a);   // (h = 350: no AGPR spills)
    else if (hs <= 7 * 64) hipLaunchKernelGGL((wide_out_kernel<7, true>), dim3(w.blocks), dim3(256), 0, s, a);
    else if (hs <= 9 * 64) hipLaunchKernelGGL((wide_out_kernel<9, true>), dim3(w.blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((wide_out_kernel<WIDE_OUT_MAXQ, false>), dim3(w.blocks), dim3(256), 0, s, a);
    if (train) {
        // (dw_o | db_o | ds_o | loss), and (db | dS) of the last hidden layer = dZ_L^T (1, X) where the same pass summed it
        const WideRedJob jo = wide_red_job(w.part, w.blocks, 1, w.hsv + 1, 2, 0);
        if (wide_out_has_ext(hs)) wide_reduce(s, m, w.grads, jo, wide_red_job(w.part2, w.blocks, h, 1 + C, 3, m.L - 1));
        else wide_reduce(s, m, w.grads, jo);
    }
    return hipGetLastError() == hipSuccess ? INR_OK : INR_ELAUNCH;
}

// backward of ONE image from dZ_L (w.dza, written by wide_forward): every remaining parameter gradient into w.grads (flat order)
inline int wide_backward(const WideMap& m, const WideWs& w, const InrModelDesc* md, const float* params, long long N, hipStream_t s,
                         const float* dlogits = nullptr, float* dcoords = nullptr) {   // dcoords [C][N] (with dlogits = dL/dlogits): also dL/dcoords
    const int h = m.h, C = m.C, hs = w.hs, hp = w.hp, parts = splitk_parts(N, h);
    float* gr = w.grads;
    int rc;
    float *dz = w.dza, *dzn = w.dzb;
    const int tiles = (int)((N + GM_BM - 1) / GM_BM), hq = (h + 15) / 16 * 16;
    const dim3 dxgrid((unsigned)((N + 15) / 16));
    if (dcoords)   // s_o dL/dlogit + dZ_L . S_{L-1}
        hipLaunchKernelGGL(wide_dx_kernel, dxgrid, dim3(256), 0, s, dz, hp, params + m.p_s(m.L - 1), h, C, N, dlogits, params + m.p_so(), dcoords, 1);
    for (int k = m.L - 1; k >= 0; --k) {
        {   // dW_k [h x h] = dz^T Z_k: the contraction over the points, split into chunks of WIDE_CHUNK
            GemmArgs g{};
            g.A = dz; g.lda = hp; g.B = w.z[k]; g.ldb = hs; g.C = w.part; g.ldc = h;
            g.M = h; g.N = h; g.K = (int)N; g.k_per_split = wide_chunk(h); g.c_split_stride = (long long)h * h; g.padA = g.padB = 1; g.buf = 1;
            if ((rc = gemm_launch(s, true, false, g))) return rc;   // (its partials are added up with the (1, x) sums below: one launch)
            // (db_k | dS_k) = dz^T (1, X): summed by the kernel that wrote dz (wide_out_kernel for the last layer, the backward GEMM's
            // epilogue below for the others); only rows too long for wide_out_kernel's accumulators take a pass of their own
            if (k == m.L - 1 && !wide_out_has_ext(hs)) {
                hipLaunchKernelGGL(wide_extgrad_kernel, dim3(w.blocks), dim3(256), 0, s, dz, hp, w.z[k] + h, hs, N, h, C, w.part2);
                wide_reduce(s, m, gr, wide_red_job(w.part2, w.blocks, h, 1 + C, 3, k));
            }
        }
        {   // dz_k = (dz W_k) (.) act'(layer k)     (the mask in the GEMM's epilogue, and dz_k^T (1, X) per 128-row tile)
            GemmArgs g{};
            g.A = dz; g.lda = hp; g.B = w.wp + (size_t)k * hq * hq; g.ldb = hq; g.C = dzn; g.ldc = hp;   // (the weights wide_forward packed)
            g.M = (int)N; g.N = h; g.K = h; g.padA = g.padB = 1; g.buf = 1; g.c_zero_to = hp;
            g.epi = GEMM_EPI_MASK;
            const int act = k == 0 ? md->act0 : INR_ACT_RELU;
            g.mask_act = act; g.omega = md->act_omega;
            if (act == INR_ACT_RELU) { g.mask = w.z[k]; g.mask_ld = hs; }
            else { g.mask = w.pre0; g.mask_ld = hp; }
            g.extsum = w.part2; g.ext = w.z[k] + h; g.ext_ld = hs; g.C_in = C;
            if ((rc = gemm_launch(s, false, false, g))) return rc;
            // what the ext inputs multiply in the layer below: (db_{k-1} | dS_{k-1}), or (db_in | dW_in) of layer 0
            wide_reduce(s, m, gr, wide_red_job(w.part, parts, h, h, 0, k), wide_red_job(w.part2, tiles, h, 1 + C, k > 0 ? 3 : 1, k > 0 ? k - 1 : 0));            if (dcoords)   // ... and what the coordinates multiply there: S_{k-1}, or W_in
                hipLaunchKernelGGL(wide_dx_kernel, dxgrid, dim3(256), 0, s, dzn, hp, params + (k > 0 ? m.p_s(k - 1) : m.p_win()), h, C, N, dlogits,
                                   params + m.p_so(), dcoords, 0);
        }
        float* t = dz; dz = dzn; dzn = t;
    }
    return hipGetLastError() == hipSuccess ? INR_OK : INR_ELAUNCH;
}

}  // namespace
