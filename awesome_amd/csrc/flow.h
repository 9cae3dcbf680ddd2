// flow.h - HIP kernels for the weight-normalised coupling flow in front of the ICNN (the path-connected prior
// ICNN(flow(Ax + b)), jp-schneider/awesome awesome/model/convex_diffeomorphism_net.py:173-178).
//
// Reference arithmetic (awesome/model/diffeomorphism_net.py:169-192, 208-232, 286-300; real_nvp/resnet_1d.py:39-63):
//   x <- A x + b                                                  (nn.Linear(2,2))
//   for i in 0..K-1:  u = (i even ? x1 : x2)
//       s = scale_i * NB_s,i(u) ;  t = NB_t,i(u)                   NB(u) = tanh(w2 . leaky_relu(w1 u + b1) + b2)
//       (i even ? x2 : x1) <- exp(s) * (i even ? x2 : x1) + t      w = g v / ||v||_F  (weight_norm, scalar g)
//   scale_i = (g v/|v|) * weight + bias                            (WNScale: weight-normed 1x1 linear of a scalar)
//
// This stage is elementwise per point with two tiny 1 -> W -> 1 MLPs per coupling: VALU work, no matrix shape.
//   flow_fwd_kernel         lane = point; weights are wave-uniform (scalar loads)            -> deformed coords
//   flow_bwd_points_kernel  lane = point; recomputes the forward, walks the couplings backwards, emits per point and
//                           coupling (u, dL/dpre_s, dL/dpre_t) and the per-point-scalar parameter gradients
//   flow_bwd_units_kernel   lane = hidden unit; streams those per-point scalars (scalar loads) and accumulates
//                           dw1, db1, dw2 of its unit in registers - no cross-lane reduction over points
//   flow_update_kernel      per coupling net: fixed-order slab reduction, weight-norm chain rule, Adam (weight decay on
//                           weight_g only, awesome/util/torch.py:19-35), and the new effective weights for the next step
#pragma once
#include "icnn_step.h"

namespace {

constexpr float LEAKY_SLOPE = 0.01f;  // F.leaky_relu default

struct FlowMap {   // layouts derived from (W, K)
    int W, K;
    int nb_stride;    // flat params per coupling net: v1[W] g1 b1[W] v2[W] g2 b2
    int p_nb, p_scale, FP;
    int e_nb_stride;  // effective weights per coupling net: [W][4] = (w1, b1, w2, w1*w2) per unit, then b2 (padded to 4)
    int e_nb, e_scale, FE;
};

__host__ __device__ inline FlowMap make_flow_map(int W, int K) {
    FlowMap m;
    m.W = W;
    m.K = K;
    m.nb_stride = 3 * W + 3;
    m.p_nb = 6;
    m.p_scale = m.p_nb + 2 * K * m.nb_stride;
    m.FP = m.p_scale + 4 * K;
    m.e_nb_stride = 4 * W + 4;
    m.e_nb = 8;
    m.e_scale = m.e_nb + 2 * K * m.e_nb_stride;
    m.FE = (m.e_scale + K + 3) / 4 * 4;
    return m;
}

__device__ __forceinline__ void load_coords2(const InrGridDesc& gd, int img, long long N, int pc, float (&x)[2]) {
    if (gd.mode == INR_GRID_SEPARABLE) {
        const int row = pc / gd.width;
        x[0] = gd.xs[pc - row * gd.width];
        x[1] = gd.ys[row];
    } else {
        const float* cp = gd.coords + (size_t)img * gd.coords_image_stride;
        x[0] = cp[pc];
        x[1] = cp[(size_t)N + pc];
    }
}

// The effective weights of one image (<= 25 KB) are copied into LDS once per block; inside the unit loops every lane reads
// the same 16-byte record (w1, b1, w2, w1*w2) of unit j: one broadcast ds_read_b128 per unit, pipelined by unrolling.
__device__ __forceinline__ void flow_weights_to_lds(const float* __restrict__ src, float* dst, int n_floats) {
    const f32x4* __restrict__ s4 = (const f32x4*)src;
    for (int i = threadIdx.x; i < n_floats / 4; i += blockDim.x) ((f32x4*)dst)[i] = s4[i];
    __syncthreads();
}

// tanh(w2 . leaky_relu(w1 u + b1) + b2) of one coupling net
__device__ __forceinline__ float nb_forward(const float* e, int W, float u) {
    float acc0 = e[4 * W], acc1 = 0.f;
    int j = 0;
    for (; j + 8 <= W; j += 8) {
        f32x4 q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = *(const f32x4*)(e + 4 * (j + k));
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const float p0 = fmaf(q[k][0], u, q[k][1]), p1 = fmaf(q[k + 1][0], u, q[k + 1][1]);
            acc0 = fmaf(q[k][2], fmaxf(p0, LEAKY_SLOPE * p0), acc0);
            acc1 = fmaf(q[k + 1][2], fmaxf(p1, LEAKY_SLOPE * p1), acc1);
        }
    }
    for (; j < W; ++j) {
        const f32x4 q = *(const f32x4*)(e + 4 * j);
        const float pre = fmaf(q[0], u, q[1]);
        acc0 = fmaf(q[2], fmaxf(pre, LEAKY_SLOPE * pre), acc0);
    }
    return tanhf(acc0 + acc1);
}

struct FlowFwdArgs {
    const float* FE;   // [n_images][FE] effective weights
    float* xd;         // [n_images][2][N]
    InrGridDesc grid;
    long long N;
    FlowMap m;
};

__global__ __launch_bounds__(256) void flow_fwd_kernel(const FlowFwdArgs a) {
    const int img = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int N = (int)a.N;
    const int pc = p < N ? p : N - 1;
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    flow_weights_to_lds(a.FE + (size_t)img * a.m.FE, fsm, a.m.FE);
    const float* e = fsm;
    float xin[2];
    load_coords2(a.grid, img, a.N, pc, xin);
    float x1 = fmaf(e[0], xin[0], fmaf(e[1], xin[1], e[4]));
    float x2 = fmaf(e[2], xin[0], fmaf(e[3], xin[1], e[5]));
    for (int i = 0; i < a.m.K; ++i) {
        const float u = (i & 1) ? x2 : x1;
        const float* es = e + a.m.e_nb + (2 * i) * a.m.e_nb_stride;
        const float s = nb_forward(es, a.m.W, u);
        const float t = nb_forward(es + a.m.e_nb_stride, a.m.W, u);
        const float ex = expf(e[a.m.e_scale + i] * s);
        if (i & 1) x1 = fmaf(ex, x1, t);
        else x2 = fmaf(ex, x2, t);
    }
    if (p < N) {
        a.xd[((size_t)img * 2) * N + p] = x1;
        a.xd[((size_t)img * 2 + 1) * N + p] = x2;
    }
}

// ---- backward, lane = point ----------------------------------------------------------------------------------------------
struct FlowBwdArgs {
    const float* FE;    // [n_images][FE]
    const float* dxd;   // [n_images][2][N] gradient w.r.t. the deformed coordinates
    float* ps;          // [n_images][K][3][N]: u, dL/dpre_s, dL/dpre_t per point and coupling
    float* slab1;       // [n_images][blocks][S1]: per-block partial sums of the per-point-scalar gradients
    InrGridDesc grid;
    long long N;
    FlowMap m;
    int S1;             // K (dscale) + 2K (db2 s,t) + 6 (dA, db)
};

template <int K>
__global__ __launch_bounds__(256) void flow_bwd_points_kernel(const FlowBwdArgs a) {
    const int img = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int N = (int)a.N, W = a.m.W;
    const bool valid = p < N;
    const int pc = valid ? p : N - 1;
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    flow_weights_to_lds(a.FE + (size_t)img * a.m.FE, fsm, a.m.FE);
    const float* e = fsm;
    float xin[2];
    load_coords2(a.grid, img, a.N, pc, xin);
    // forward, keeping the state in front of every coupling and the net outputs
    float x1s[K], x2s[K], sv[K], tv[K], ev[K];
    float x1 = fmaf(e[0], xin[0], fmaf(e[1], xin[1], e[4]));
    float x2 = fmaf(e[2], xin[0], fmaf(e[3], xin[1], e[5]));
#pragma unroll
    for (int i = 0; i < K; ++i) {
        x1s[i] = x1;
        x2s[i] = x2;
        const float u = (i & 1) ? x2 : x1;
        const float* es = e + a.m.e_nb + (2 * i) * a.m.e_nb_stride;
        sv[i] = nb_forward(es, W, u);
        tv[i] = nb_forward(es + a.m.e_nb_stride, W, u);
        ev[i] = expf(e[a.m.e_scale + i] * sv[i]);
        if (i & 1) x1 = fmaf(ev[i], x1, tv[i]);
        else x2 = fmaf(ev[i], x2, tv[i]);
    }
    float d1 = valid ? a.dxd[((size_t)img * 2) * N + pc] : 0.f;
    float d2 = valid ? a.dxd[((size_t)img * 2 + 1) * N + pc] : 0.f;
    float acc[3 * K + 6];  // dscale[K] | db2_s[K] | db2_t[K] | dA[4] | db[2]
#pragma unroll
    for (int k = 0; k < 3 * K + 6; ++k) acc[k] = 0.f;
#pragma unroll
    for (int ii = 0; ii < K; ++ii) {
        const int i = K - 1 - ii;
        const bool odd = i & 1;
        const float u = odd ? x2s[i] : x1s[i];
        const float tpre = odd ? x1s[i] : x2s[i];
        const float dpost = odd ? d1 : d2;
        const float de = dpost * tpre;             // d/d exp(s)
        const float sc = e[a.m.e_scale + i];
        const float dse = de * ev[i];              // d/d (scale * s_raw)
        acc[i] += dse * sv[i];                     // d/d scale_i
        const float gqs = dse * sc * (1.f - sv[i] * sv[i]);   // d/d pre-tanh of the s net
        const float gqt = dpost * (1.f - tv[i] * tv[i]);      // d/d pre-tanh of the t net
        acc[K + i] += gqs;
        acc[2 * K + i] += gqt;
        // du = sum_j gq * w2_j * leaky'(pre_j) * w1_j  over both nets
        const float* es = e + a.m.e_nb + (2 * i) * a.m.e_nb_stride;
        const float* et = es + a.m.e_nb_stride;
        float dus = 0.f, dut = 0.f;
#pragma unroll 4
        for (int j = 0; j < W; ++j) {
            const f32x4 qs = *(const f32x4*)(es + 4 * j), qt = *(const f32x4*)(et + 4 * j);
            const float ps_ = fmaf(qs[0], u, qs[1]);
            const float pt_ = fmaf(qt[0], u, qt[1]);
            dus = fmaf(qs[3], ps_ > 0.f ? 1.f : LEAKY_SLOPE, dus);   // qs[3] = w1 * w2
            dut = fmaf(qt[3], pt_ > 0.f ? 1.f : LEAKY_SLOPE, dut);
        }
        const float du = gqs * dus + gqt * dut;
        if (valid) {
            float* pp = a.ps + (((size_t)img * K + i) * 3) * N + p;
            pp[0] = u;
            pp[(size_t)N] = gqs;
            pp[2 * (size_t)N] = gqt;
        }
        const float dtpre = dpost * ev[i];
        if (odd) {
            d1 = dtpre;
            d2 += du;
        } else {
            d2 = dtpre;
            d1 += du;
        }
    }
    // nn.Linear(2,2): y_r = sum_c A[r][c] x_c + b_r
    acc[3 * K + 0] = d1 * xin[0];
    acc[3 * K + 1] = d1 * xin[1];
    acc[3 * K + 2] = d2 * xin[0];
    acc[3 * K + 3] = d2 * xin[1];
    acc[3 * K + 4] = d1;
    acc[3 * K + 5] = d2;
    // block reduction (fixed order): wave sums by DPP/permlane, then the 4 waves through LDS
    __shared__ float red[4][3 * K + 6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3 * K + 6; ++k) {
        const float v = sum_over_groups(sum_over_points(acc[k]));
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3 * K + 6) {
        const int k = threadIdx.x;
        a.slab1[((size_t)img * gridDim.x + blockIdx.x) * a.S1 + k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    }
}

// ---- backward, lane = hidden unit -----------------------------------------------------------------------------------------
struct FlowUnitsArgs {
    const float* FE;
    const float* ps;     // [n_images][K][3][N]
    float* slab2;        // [n_images][chunks][K*2][3][Wp]  (dw1, db1, dw2 by unit)
    long long N;
    FlowMap m;
    int chunks, Wp;      // Wp = unit blocks * 64
};

__global__ __launch_bounds__(256) void flow_bwd_units_kernel(const FlowUnitsArgs a) {
    // grid: x = chunk, y = (coupling*2 + net) * unit_blocks + ub, z = image; wave w of the block takes a quarter of the chunk
    const int UB = a.Wp / 64;
    const int img = blockIdx.z, chunk = blockIdx.x;
    const int nb = blockIdx.y / UB, ub = blockIdx.y - nb * UB;
    const int i = nb >> 1, net = nb & 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = (int)a.N, W = a.m.W;
    const int unit = ub * 64 + lane;
    const bool on = unit < W;
    const float* __restrict__ e = a.FE + (size_t)img * a.m.FE + a.m.e_nb + nb * a.m.e_nb_stride;
    const float w1 = on ? e[4 * unit] : 0.f, b1 = on ? e[4 * unit + 1] : 0.f, w2 = on ? e[4 * unit + 2] : 0.f;
    const int per_chunk = (N + a.chunks - 1) / a.chunks;
    const int per_wave = (per_chunk + 3) / 4;
    const int p0 = chunk * per_chunk + wave * per_wave;
    int p1 = p0 + per_wave;
    const int cend = (chunk + 1) * per_chunk;
    if (p1 > cend) p1 = cend;
    if (p1 > N) p1 = N;
    const float* __restrict__ pu = a.ps + (((size_t)img * a.m.K + i) * 3) * N;
    const float* __restrict__ pg = pu + (size_t)(1 + net) * N;
    float aw1 = 0.f, ab1 = 0.f, aw2 = 0.f;
    auto one = [&](float u, float gq) {
        const float pre = fmaf(w1, u, b1);
        const float h = fmaxf(pre, LEAKY_SLOPE * pre);
        aw2 = fmaf(gq, h, aw2);
        const float dh = gq * w2 * (pre > 0.f ? 1.f : LEAKY_SLOPE);
        aw1 = fmaf(dh, u, aw1);
        ab1 += dh;
    };
    // 64 points per trip: one coalesced vector load per array (the next trip's loads are already in flight), then every
    // point's (u, gq) is broadcast to the wave with v_readlane - no memory access inside the 64-point body
    float un = 0.f, gn = 0.f;
    if (p0 + lane < p1) {
        un = pu[p0 + lane];
        gn = pg[p0 + lane];
    }
    for (int p = p0; p < p1; p += 64) {
        const float uc = un, gc = gn;   // gq = 0 for the lanes past p1: those points contribute nothing
        un = 0.f;
        gn = 0.f;
        if (p + 64 + lane < p1) {
            un = pu[p + 64 + lane];
            gn = pg[p + 64 + lane];
        }
#pragma unroll
        for (int k = 0; k < 64; ++k)
            one(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, uc), k)),
                __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gc), k)));
    }
    __shared__ float red[4][3][64];
    red[wave][0][lane] = aw1;
    red[wave][1][lane] = ab1;
    red[wave][2][lane] = aw2;
    __syncthreads();
    if (threadIdx.x < 192) {
        const int q = threadIdx.x >> 6, l = threadIdx.x & 63;
        const float v = ((red[0][q][l] + red[1][q][l]) + red[2][q][l]) + red[3][q][l];
        a.slab2[((((size_t)img * a.chunks + chunk) * (a.m.K * 2) + nb) * 3 + q) * a.Wp + ub * 64 + l] = v;
    }
}

// ---- reduction + weight-norm chain rule + Adam + new effective weights --------------------------------------------------
struct FlowUpdArgs {
    float* FP;            // [n_images][FP] flow parameters (in/out)
    float* FE;            // [n_images][FE] effective weights (out)
    float* opt;           // [n_images][2*FP] exp_avg | exp_avg_sq (mode 0)
    float* grads_out;     // [n_images][FP]               (mode 1)
    const float* slab1;   // [n_images][blocks1][S1]
    const float* slab2;   // [n_images][chunks][K*2][3][Wp]
    const float* lr_hdr;  // ICNN opt-state header of image 0 stride...: lr of this step is hdr[t & 1]; null -> opt.lr
    long long hdr_stride;
    InrOptDesc opt_desc;
    FlowMap m;
    int blocks1, S1, chunks, Wp;
    int t;
    double bc1;
    float bc2_sqrt, one_minus_b1, one_minus_b2, wd_g;
    int mode;             // 0 = Adam step + effective weights, 1 = gradients only, 2 = effective weights only (prep)
};

__device__ __forceinline__ float block_sum256(float v, float* sm) {  // fixed-order sum over a 256-thread block
    v = sum_over_groups(sum_over_points(v));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((sm[0] + sm[1]) + sm[2]) + sm[3];
}

__device__ __forceinline__ float adam_apply(const FlowUpdArgs& u, float p, float g, float lr, float wd, float* m_, float* v_) {
    if (wd != 0.f) g = __fadd_rn(g, __fmul_rn(wd, p));
    float m = *m_, v = *v_;
    m = __fadd_rn(m, __fmul_rn(u.one_minus_b1, __fsub_rn(g, m)));
    v = __fadd_rn(__fmul_rn(v, u.opt_desc.beta2), __fmul_rn(__fmul_rn(u.one_minus_b2, g), g));
    const float step_size = (float)((double)lr / u.bc1);
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), u.bc2_sqrt), u.opt_desc.eps);
    *m_ = m;
    *v_ = v;
    return __fadd_rn(p, __fdiv_rn(__fmul_rn(-step_size, m), denom));
}

// grid: x = K*2 coupling nets + 1 (block K*2: scales + linear), y = image; 256 threads
__global__ __launch_bounds__(256) void flow_update_kernel(const FlowUpdArgs u) {
    __shared__ float sm[4];
    const int img = blockIdx.y, nb = blockIdx.x, tid = threadIdx.x;
    const FlowMap& m = u.m;
    const int W = m.W, K = m.K;
    float* __restrict__ fp = u.FP + (size_t)img * m.FP;
    float* __restrict__ fe = u.FE + (size_t)img * m.FE;
    float* __restrict__ om = u.opt ? u.opt + (size_t)img * 2 * m.FP : nullptr;
    float* __restrict__ ov = om ? om + m.FP : nullptr;
    float* __restrict__ go = u.grads_out ? u.grads_out + (size_t)img * m.FP : nullptr;
    const float lr = u.lr_hdr ? u.lr_hdr[(size_t)img * u.hdr_stride + (u.t & 1)] : u.opt_desc.lr;
    if (nb < 2 * K) {
        const int pb = m.p_nb + nb * m.nb_stride;    // v1[W] g1 b1[W] v2[W] g2 b2
        const int eb = m.e_nb + nb * m.e_nb_stride;  // w1[W] b1[W] w2[W] b2
        const int i = nb >> 1, net = nb & 1;
        const bool on = tid < W;
        float v1 = on ? fp[pb + tid] : 0.f, b1 = on ? fp[pb + W + 1 + tid] : 0.f, v2 = on ? fp[pb + 2 * W + 1 + tid] : 0.f;
        float g1 = fp[pb + W], g2 = fp[pb + 3 * W + 1], b2 = fp[pb + 3 * W + 2];
        if (u.mode != 2) {
            // effective-weight gradients: fixed-order sums over the chunks / blocks
            float dw1 = 0.f, db1 = 0.f, dw2 = 0.f;
            if (on) {
                const float* s2 = u.slab2 + (((size_t)img * u.chunks * (K * 2) + nb) * 3) * u.Wp + tid;
                for (int c = 0; c < u.chunks; ++c) {
                    const float* q = s2 + (size_t)c * (K * 2) * 3 * u.Wp;
                    dw1 += q[0];
                    db1 += q[u.Wp];
                    dw2 += q[2 * u.Wp];
                }
            }
            float db2 = 0.f;
            {
                float part = 0.f;
                for (int b = tid; b < u.blocks1; b += 256) part += u.slab1[((size_t)img * u.blocks1 + b) * u.S1 + (1 + net) * K + i];
                db2 = block_sum256(part, sm);
            }
            // weight norm (dim=None): w = g v / n  =>  dg = <dw, v> / n ;  dv = g/n (dw - v <dw, v> / n^2)
            const float n1 = sqrtf(block_sum256(v1 * v1, sm)), n2 = sqrtf(block_sum256(v2 * v2, sm));
            const float dot1 = block_sum256(dw1 * v1, sm), dot2 = block_sum256(dw2 * v2, sm);
            const float dg1 = dot1 / n1, dg2 = dot2 / n2;
            const float dv1 = g1 / n1 * (dw1 - v1 * dot1 / (n1 * n1)), dv2 = g2 / n2 * (dw2 - v2 * dot2 / (n2 * n2));
            if (u.mode == 1) {
                if (on) {
                    go[pb + tid] = dv1;
                    go[pb + W + 1 + tid] = db1;
                    go[pb + 2 * W + 1 + tid] = dv2;
                }
                if (tid == 0) {
                    go[pb + W] = dg1;
                    go[pb + 3 * W + 1] = dg2;
                    go[pb + 3 * W + 2] = db2;
                }
                return;
            }
            if (on) {
                v1 = adam_apply(u, v1, dv1, lr, 0.f, &om[pb + tid], &ov[pb + tid]);
                b1 = adam_apply(u, b1, db1, lr, 0.f, &om[pb + W + 1 + tid], &ov[pb + W + 1 + tid]);
                v2 = adam_apply(u, v2, dv2, lr, 0.f, &om[pb + 2 * W + 1 + tid], &ov[pb + 2 * W + 1 + tid]);
                fp[pb + tid] = v1;
                fp[pb + W + 1 + tid] = b1;
                fp[pb + 2 * W + 1 + tid] = v2;
            }
            // scalars: every thread computes the same values (no divergence in the block sums below); thread 0 stores
            {
                float m1 = om[pb + W], q1 = ov[pb + W], m2 = om[pb + 3 * W + 1], q2 = ov[pb + 3 * W + 1];
                float m3 = om[pb + 3 * W + 2], q3 = ov[pb + 3 * W + 2];
                g1 = adam_apply(u, g1, dg1, lr, u.wd_g, &m1, &q1);
                g2 = adam_apply(u, g2, dg2, lr, u.wd_g, &m2, &q2);
                b2 = adam_apply(u, b2, db2, lr, 0.f, &m3, &q3);
                __syncthreads();  // everyone has read the old state
                if (tid == 0) {
                    om[pb + W] = m1; ov[pb + W] = q1; om[pb + 3 * W + 1] = m2; ov[pb + 3 * W + 1] = q2;
                    om[pb + 3 * W + 2] = m3; ov[pb + 3 * W + 2] = q3;
                    fp[pb + W] = g1; fp[pb + 3 * W + 1] = g2; fp[pb + 3 * W + 2] = b2;
                }
            }
        }
        // effective weights for the next forward
        const float n1 = sqrtf(block_sum256(v1 * v1, sm)), n2 = sqrtf(block_sum256(v2 * v2, sm));
        if (on) {
            const float w1e = v1 * (g1 / n1), w2e = v2 * (g2 / n2);
            *(f32x4*)(fe + eb + 4 * tid) = f32x4{w1e, b1, w2e, w1e * w2e};
        }
        if (tid == 0) fe[eb + 4 * W] = b2;
        return;
    }
    // last block: WNScale parameters of every coupling + the 2x2 linear: first reduce their per-block partial sums
    __shared__ float tot[64];
    if (u.mode != 2) {
        for (int k = 0; k < 3 * K + 6; ++k) {
            if (k >= K && k < 3 * K) continue;   // db2 sums belong to the coupling-net blocks
            float part = 0.f;
            for (int b = tid; b < u.blocks1; b += 256) part += u.slab1[((size_t)img * u.blocks1 + b) * u.S1 + k];
            const float t = block_sum256(part, sm);
            if (tid == 0) tot[k] = t;
        }
        __syncthreads();
    }
    if (tid < K) {
        const int i = tid, pb = m.p_scale + 4 * i;  // weight, sc_bias, sc_g, sc_v
        float w = fp[pb], sb = fp[pb + 1], sg = fp[pb + 2], sv = fp[pb + 3];
        if (u.mode != 2) {
            const float dsc = tot[i];
            const float sgn = sv / fabsf(sv);        // weight_norm(dim=0) of a 1x1 weight: v / |v|
            const float dw = dsc * sg * sgn, dsb = dsc, dsg = dsc * w * sgn, dsv = 0.f;
            if (u.mode == 1) {
                go[pb] = dw; go[pb + 1] = dsb; go[pb + 2] = dsg; go[pb + 3] = dsv;
            } else {
                w = adam_apply(u, w, dw, lr, 0.f, &om[pb], &ov[pb]);
                sb = adam_apply(u, sb, dsb, lr, 0.f, &om[pb + 1], &ov[pb + 1]);
                sg = adam_apply(u, sg, dsg, lr, u.wd_g, &om[pb + 2], &ov[pb + 2]);
                sv = adam_apply(u, sv, dsv, lr, 0.f, &om[pb + 3], &ov[pb + 3]);
                fp[pb] = w; fp[pb + 1] = sb; fp[pb + 2] = sg; fp[pb + 3] = sv;
            }
        }
        if (u.mode != 1) fe[m.e_scale + i] = sg * (sv / fabsf(sv)) * w + sb;
    } else if (tid >= 64 && tid < 70) {
        const int k = tid - 64;  // A[0][0], A[0][1], A[1][0], A[1][1], b[0], b[1]
        float p = fp[k];
        if (u.mode != 2) {
            const float d = tot[3 * K + k];
            if (u.mode == 1) go[k] = d;
            else {
                p = adam_apply(u, p, d, lr, 0.f, &om[k], &ov[k]);
                fp[k] = p;
            }
        }
        if (u.mode != 1) fe[k] = p;
    }
}

}  // namespace
