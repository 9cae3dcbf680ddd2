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
#include <stdlib.h>

namespace {

constexpr float LEAKY_SLOPE = 0.01f;  // F.leaky_relu default (NormalBlock); SimpleBackbone uses relu = slope 0 (FlowMap::slope)

struct FlowMap {   // layouts derived from (W, K)
    int W, K;
    int nb_stride;    // flat params per coupling net: v1[W] g1 b1[W] v2[W] g2 b2
    int p_nb, p_scale, FP;
    int e_cp_stride;  // effective weights per COUPLING (see "relu form" below): [W][8] records
                      // (w1s, w1t, b1s, b1t, w2's, w2't, w1s*w2's, w1t*w2't) per unit, then (b2's, b2't, a's, a't):
                      // the s and t nets share their input, so they are evaluated as one packed pair
    int e_nb, e_scale, FE;
    float slope;      // negative slope of the hidden activation: 0.01 (NormalBlock, diffeomorphism_net.py:169-192) or 0 (SimpleBackbone, :83-104)
};

__host__ __device__ inline FlowMap make_flow_map(int W, int K, float slope = LEAKY_SLOPE) {
    FlowMap m;
    m.W = W;
    m.K = K;
    m.slope = slope;
    m.nb_stride = 3 * W + 3;
    m.p_nb = 6;
    m.p_scale = m.p_nb + 2 * K * m.nb_stride;
    m.FP = m.p_scale + 4 * K;
    m.e_cp_stride = 8 * W + 4;
    m.e_nb = 8;
    m.e_scale = m.e_nb + K * m.e_cp_stride;
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
// The copy is LDS-DMA (global_load_lds_dwordx4: a wave moves 1 KB per instruction straight into LDS, no registers, every piece in
// flight at once); the register-staged loop it replaces waited for each of its ~7 loads per thread before issuing the next one.
__device__ __forceinline__ void flow_weights_to_lds(const float* __restrict__ src, float* dst, int n_floats) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int pieces = n_floats / 256;
    for (int pc = wave; pc < pieces; pc += nw)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pc * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(dst + pc * 256), 16, 0, 0);
    const f32x4* __restrict__ s4 = (const f32x4*)src;
    for (int i = pieces * 64 + threadIdx.x; i < n_floats / 4; i += blockDim.x) ((f32x4*)dst)[i] = s4[i];
    __syncthreads();   // (its fence waits for vmcnt(0): the DMA pieces have landed)
}


// "relu form" of a coupling net.  leaky_relu(p) = slope p + (1 - slope) relu(p), and the slope part is affine in u:
//   w2 . leaky_relu(w1 u + b1) + b2  =  (b2 + slope <w2, b1>)  +  slope <w2, w1> u  +  sum_j w2'_j relu(w1_j u + b1_j)
// with w2' = (1 - slope) w2.  The update kernel keeps b2' = b2 + slope <w2, b1>, a' = slope <w2, w1> and w2' in the
// effective weights, so a unit costs fma + max + fma here, and its derivative is a' + sum_j (w1_j w2'_j) step(pre_j).
// step(p) = (p > 0) is one multiply by 2^126 with the [0, 1] output clamp (exact for every normal p).
// The library is built with -ffp-contract=off (arithmetic = what the source says, awesome_amd/build.py): the fused multiply-adds of
// the unit loops are written out.  v_pk_fma_f32.
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }

// relu through the CLAMP modifier.  The records hold the first layer scaled by REC_DOWN = 2^-32 and w2' scaled by REC_UP = 2^32 (exact:
// powers of two), so one `v_pk_fma_f32 ... clamp` gives relu(pre) 2^-32 for every pre <= 2^32 (clamp = [0, 1]; coordinates and weights
// are O(1)), and the second fma multiplies the 2^32 back in: two instructions per unit instead of fma + 2 x v_max + fma (there is no
// packed fp32 max), the same bits.  src1 broadcasts its low half to both lanes (op_sel_hi).
constexpr float REC_DOWN = 0x1p-32f, REC_UP = 0x1p32f;
__device__ __forceinline__ f32x2 pk_fma_clamp_bcast(f32x2 a, float u, f32x2 c) {   // clamp01(a * (u, u) + c)
    f32x2 r;
    const f32x2 uu = f32x2{u, u};
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(uu), "v"(c));
    return r;
}

__device__ __forceinline__ f32x2 step01(f32x2 pre) {
    f32x2 r;
    const f32x2 big = f32x2{0x1p126f, 0x1p126f};
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(pre), "v"(big));
    return r;
}

// ---- tanh / exp of the coupling outputs ---------------------------------------------------------------------------------------
// The coupling needs 2 tanh + 1 exp per point, flow and output channel.  libm's tanhf / expf are ~35 / ~12 VALU instructions with
// data-dependent branches (cheap when a whole wave takes the same one, e.g. all arguments small; both sides when it does not); the
// forms below are branch-free at ~19 / 5 and keep libm's accuracy class - RELATIVE error, so tiny outputs of the zero-initialised nets
// stay exact to rounding.  They did not change the kernels' time (profiles/NOTES.md); what they buy is a run time that does not depend
// on the data and error bounds that are asserted on the device:
//   exp:  e^x = 2^hi (1 + lo ln 2) with x log2(e) = hi + lo split exactly by one fma (v_exp_f32: 1 ulp), |x| <~ 80;
//         measured max rel. error 1.27e-7 on [-10, 10] (tests/test_gpu_rnvp.py::test_fast_tanh_exp_error_bounds asserts 2.5e-7)
//   tanh: |x| < 0.625: odd minimax polynomial x + x^3 P(x^2) (the Cephes tanhf coefficients); else 1 - 2 / (e^{2|x|} + 1) with the exp
//         above and v_rcp_f32 (1 ulp); measured max rel. error 1.61e-7 (asserted: 5e-7; libm: 1.2e-7).
// A first attempt in round 1 (tanh x = 1 - 2/(1 + e^2x) for ALL x, plain v_exp_f32 of x log2e) had ABSOLUTE error 2e-7, i.e. large
// relative error near 0 where the couplings start: "3x noisier" gradients.  The relative-error forms do not have that problem
// (test_accuracy_against_float64 holds with the same x4 bar as libm).
__device__ __forceinline__ float fast_exp(float x) {
    constexpr float L2E = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-8f, LN2 = 0.693147182464599609375f;
    const float hi = x * L2E;
    const float lo = fmaf(x, L2E_LO, fmaf(x, L2E, -hi));   // x log2(e) - hi
    const float e = __builtin_amdgcn_exp2f(hi);
    return fmaf(e * LN2, lo, e);
}
__device__ __forceinline__ float fast_tanh(float x) {
    const float ax = fabsf(x);
    const float z = x * x;
    float p = fmaf(-5.70498872745e-3f, z, 2.06390887954e-2f);
    p = fmaf(p, z, -5.37397155531e-2f);
    p = fmaf(p, z, 1.33314422036e-1f);
    p = fmaf(p, z, -3.33332819422e-1f);
    const float small = fmaf(p * z, x, x);
    const float e = fast_exp(fminf(2.f * ax, 40.f));        // tanh saturates to 1 long before; keeps 2^hi finite
    const float big = fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
    return ax < 0.625f ? small : copysignf(big, x);
}

// ---- where the effective weights come from, and what bounds the point kernels -----------------------------------------------------------
// The weights are WAVE-UNIFORM (every lane evaluates the same nets on its own point) and are read from an LDS copy of the image with
// broadcast ds_read_b128, 2 per hidden unit.  Round 3 took round 2's "VALU-issue bound at 0.6 busy" apart with four experiments
// (rocprofv3 kernel stats of tools/kbench_{pcn,cdn}.py in profiles/r03_*; the table is in profiles/NOTES.md):
//   * mask-specialised bodies (no scratch, no select chains) and branch-free tanh / exp (~19 / 5 instructions; libm's are longer but
//     skip whole branches when a wave agrees): the forward kernels did not move (55.7 vs 53.2 us at configs[3], 18.2 vs 18.3 at
//     256x256) - and by SQ_INSTS_VALU the instruction count did not fall either (270 vs 246 per wave and flow);
//   * the scalar data path (image read from HBM through a constant-address-space pointer -> s_load_dwordx4/x8 into SGPRs, RecK below;
//     no LDS traffic at all): 2-3x SLOWER (coupling-flow forward 64 vs 21.5 us): the scalar cache cannot feed the loop;
//   * Q = 2 / 4 points per lane (a record read serves Q points; 64-thread blocks so the fewer waves still spread over all CUs):
//     1.1-1.6x SLOWER, with and without the pipelining below: not LDS return bandwidth either;
//   * software-pipelined unit loops (the next 4 units' records requested before the current 4 are evaluated; hipcc's own order was
//     "reads -> wait -> arithmetic -> next reads"): RealNVP backward at 256x256 38.1 -> 28.9 us, coupling-flow backward 27.6 -> 25.4,
//     forward kernels +-1 us, configs[3] (4 waves per SIMD) unchanged.  KEPT.
// What is left is the size of the problem: one 256x256 image is 1024 waves for 1024 SIMDs - ONE wave per SIMD, 66 clocks per hidden
// unit for 16 clocks of VALU issue - and a fit step is a chain of such launches.
typedef const float __attribute__((address_space(4))) kfloat;
typedef const f32x4 __attribute__((address_space(4))) kf32x4;
struct RecK {   // image in global memory, scalar loads
    static constexpr bool SCALAR = true;
    const kfloat* p;
    __device__ __forceinline__ f32x4 v4(int off) const { return *(const kf32x4*)(p + off); }
    __device__ __forceinline__ float f(int off) const { return p[off]; }
    __device__ __forceinline__ RecK at(int off) const { return RecK{p + off}; }
};
struct RecL {   // image in LDS
    static constexpr bool SCALAR = false;
    const float* p;
    __device__ __forceinline__ f32x4 v4(int off) const { return *(const f32x4*)(p + off); }
    __device__ __forceinline__ float f(int off) const { return p[off]; }
    __device__ __forceinline__ RecL at(int off) const { return RecL{p + off}; }
};
__device__ __forceinline__ RecK rec_global(const float* image) { return RecK{(const kfloat*)image}; }

// Launch shape of the point kernels: U lanes per point, Q points per lane, 256 threads.  A launch with at least two waves per SIMD
// runs at (1, 1): more points per lane were measured slower at every size (profiles/NOTES.md).  A small launch (one image of 256x256:
// ONE wave per SIMD at (1, 1)) cuts the unit loop over U lanes - forward U = 4 with every record read serving Q = 2 points, backward
// U = 2 (it carries 7 saved values per coupling and point: fewer registers, fewer partial-sum blocks).
// Measured (us, 256x256, K = 6, W = 130; tools/experiments/expq.sh; (U, Q); with the clamp-form unit evaluation):
//   forward            (1,1) 19.9 | (2,1) 15.6 | (4,1) 16.7 | (2,2) 13.9 | (4,2) 13.0 | (4,4) 14.2      [before the clamp form: 20.3 .. 15.6]
//   backward, points   (1,1) 22.3 | (2,1) 19.6 | (4,1) 21.7 | (4,2) 19.4
//   Q alone (one lane per point, 256 / Q threads), forward, before the clamp form: Q = 2 30.6, Q = 4 46.3
struct FlowShape { int Q, U, threads, blocks; };
inline FlowShape flow_launch_shape(long long n_points, int n_images, bool backward) {
    FlowShape s;
    const char* fe = getenv("INR_FLOW_SHAPE");   // measurement / test switch: 10 U + Q for both kernels (read per call)
    const int force = fe ? atoi(fe) : 0;
    const long long total = n_points * n_images;
    const bool small = total <= 98304, large = total >= 196608;
    // small: lanes per point (above).  large (batches of images: several waves per SIMD anyway): two points per lane, every record
    // read serves both - 16 images of 256x256 per launch: (1,1) 2692 | (1,2) 2547 | (2,1) 2700 | (4,2) 2628 us per optimizer step.
    s.U = force ? force / 10 : (small ? (backward ? 2 : 4) : 1);
    s.Q = force ? force % 10 : (small ? (backward ? 1 : 2) : (large ? 2 : 1));
    const bool fwd_only = (s.U == 2 && s.Q == 2) || (s.U == 4 && s.Q == 4);   // forward instantiations without a backward twin
    if (backward && fwd_only) s.Q = 1;
    if (!((s.U == 1 && (s.Q == 1 || s.Q == 2)) || (s.U == 4 && s.Q == 2) || (s.U == 4 && s.Q == 1) || (s.U == 2 && s.Q == 1) ||
          (!backward && fwd_only)))
        s.U = s.Q = 1;
    s.threads = 256;
    s.blocks = (int)((n_points * s.U + 256 * s.Q - 1) / (256 * s.Q));
    return s;
}

inline FlowShape plain_launch_shape(long long n_points) { return FlowShape{1, 1, 256, (int)((n_points + 255) / 256)}; }

// sum over the U adjacent lanes that share a point (U = 2, 4: inside a DPP quad); every lane ends with the same bits
template <int U>
__device__ __forceinline__ float lanes_sum(float v) {
    if constexpr (U >= 2) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm(1,0,3,2)
    if constexpr (U >= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm(2,3,0,1)
    if constexpr (U >= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror: the other quad
    return v;
}

// U LANES PER POINT (small launches).  One 256x256 image is one point per lane of the chip: one wave per SIMD, and nothing covers the
// latency of a wave's dependent chain through the W = 130 units of a coupling (flow.h header: 66 clocks per unit for 16 of issue; more
// points per lane made it worse).  The sum over the units is a reduction, so it can be cut the other way: the U lanes of a point take
// the units s, s + U, s + 2U, ... (their records are U consecutive 32-byte records: one conflict-free ds_read_b128 per wave), add their
// partial (acc, d) pairs through DPP, and evaluate the rest of the coupling redundantly.  The chain per wave is ~U times shorter and U
// waves per SIMD cover each other.  The order of the sum over the units changes (per-lane partials, then lanes): results differ from
// U = 1 by rounding; which U a launch uses depends only on its size, so a given problem is reproducible.
template <bool DU, int U, int Q, class Rec>
__device__ __forceinline__ void nb_pair_forward_split(const Rec e, int W, const int s, const float (&u)[Q], f32x2 (&st)[Q], f32x2 (&dpre_du)[Q]) {
    const f32x4 tail = e.v4(8 * W);
    f32x2 acc[Q], d[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = d[q] = f32x2{0.f, 0.f};
    auto unit = [&](const f32x4& lo, const f32x4& hi) {   // lo = (w1s, w1t, b1s, b1t), hi = (w2's, w2't, w1s w2's, w1t w2't)
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const f32x2 h = pk_fma_clamp_bcast(f32x2{lo[0], lo[1]}, u[q], f32x2{lo[2], lo[3]});   // relu(pre) 2^-32
            acc[q] = pk_fma(f32x2{hi[0], hi[1]}, h, acc[q]);
            if (DU) d[q] = pk_fma(f32x2{hi[2], hi[3]}, step01(h), d[q]);
        }
    };
    const int steps = W / U;
    int t = 0;
    for (; t + 4 <= steps; t += 4) {
        f32x4 r[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r[2 * k] = e.v4(8 * ((t + k) * U + s));
            r[2 * k + 1] = e.v4(8 * ((t + k) * U + s) + 4);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) unit(r[2 * k], r[2 * k + 1]);
    }
    for (; t < steps; ++t) unit(e.v4(8 * (t * U + s)), e.v4(8 * (t * U + s) + 4));
    if (steps * U + s < W) unit(e.v4(8 * (steps * U + s)), e.v4(8 * (steps * U + s) + 4));   // the W mod U last units
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        f32x2 a2 = f32x2{lanes_sum<U>(acc[q][0]), lanes_sum<U>(acc[q][1])};
        a2 += pk_fma(f32x2{tail[2], tail[3]}, splat2(u[q]), f32x2{tail[0], tail[1]});
        st[q] = f32x2{fast_tanh(a2[0]), fast_tanh(a2[1])};
        if (DU) dpre_du[q] = f32x2{lanes_sum<U>(d[q][0]), lanes_sum<U>(d[q][1])} + f32x2{tail[2], tail[3]};
    }
}

// (NB_s(u), NB_t(u)) of one coupling for both nets at once on packed f32 pairs (v_pk_fma_f32: half the VALU
// instructions of two scalar evaluations); DU adds the derivatives d(pre-tanh)/du: every unit adds (w1 w2') step(pre)
template <bool DU, int Q, class Rec>
__device__ __forceinline__ void nb_pair_forward(const Rec e, int W, const float (&u)[Q], f32x2 (&st)[Q], f32x2 (&dpre_du)[Q]) {
    const f32x4 tail = e.v4(8 * W);
    f32x2 acc[Q], d[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        acc[q] = pk_fma(f32x2{tail[2], tail[3]}, splat2(u[q]), f32x2{tail[0], tail[1]});
        d[q] = f32x2{tail[2], tail[3]};
    }
    auto unit = [&](const f32x4& lo, const f32x4& hi) {   // lo = (w1s, w1t, b1s, b1t), hi = (w2's, w2't, w1s w2's, w1t w2't)
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            f32x2 h;   // relu(pre) 2^-32 (pk_fma_clamp_bcast)
            if constexpr (Rec::SCALAR) {   // (experiment path: one SGPR pair per instruction)
                const f32x2 pre = f32x2{lo[0], lo[1]} * splat2(u[q]) + f32x2{lo[2], lo[3]};
                h = f32x2{fminf(fmaxf(pre[0], 0.f), 1.f), fminf(fmaxf(pre[1], 0.f), 1.f)};
            } else {
                h = pk_fma_clamp_bcast(f32x2{lo[0], lo[1]}, u[q], f32x2{lo[2], lo[3]});
            }
            acc[q] = pk_fma(f32x2{hi[0], hi[1]}, h, acc[q]);
            if (DU) d[q] = pk_fma(f32x2{hi[2], hi[3]}, step01(h), d[q]);
        }
    };
    // Software-pipelined over batches of 4 units (two register sets, ping-pong): the records of the NEXT batch are requested before the
    // current batch is evaluated.  hipcc's own order for the plain loop was "6 reads -> wait -> 16 VALU -> the next 6 reads": with one
    // wave per SIMD (a 256x256 image = 1024 waves) nothing covers the LDS latency of every batch - 66 clocks per unit for 16 clocks of
    // VALU issue (round 3, flow.h header).  Reads past the last unit stay inside the LDS allocation (the launch adds slack) and are unused.
    auto load4 = [&](f32x4 (&r)[8], int j0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = e.v4(8 * j0 + 4 * k);
    };
    auto eval4 = [&](const f32x4 (&r)[8]) {
        unit(r[0], r[1]);
        unit(r[2], r[3]);
        unit(r[4], r[5]);
        unit(r[6], r[7]);
    };
    int j = 0;
    if constexpr (DU) {   // backward: pipelined (27.6 -> 25.4 us); the forward keeps hipcc's own order (21.5 vs 22.7 us pipelined)
        if (W >= 8) {
            f32x4 ra[8], rb[8];
            load4(ra, 0);
            for (; j + 8 <= W; j += 8) {
                load4(rb, j + 4);
                __builtin_amdgcn_sched_barrier(0);   // keep the requests ahead of the arithmetic
                eval4(ra);
                load4(ra, j + 8);
                __builtin_amdgcn_sched_barrier(0);
                eval4(rb);
            }
        }
    } else {
        for (; j + 4 <= W; j += 4) {
            f32x4 r[8];
            load4(r, j);
            eval4(r);
        }
    }
    for (; j < W; ++j) unit(e.v4(8 * j), e.v4(8 * j + 4));
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        st[q] = f32x2{fast_tanh(acc[q][0]), fast_tanh(acc[q][1])};
        dpre_du[q] = d[q];
    }
}

struct FlowFwdArgs {
    const float* FE;   // [n_images][FE] effective weights
    float* xd;         // [n_images][2][N]
    InrGridDesc grid;
    long long N;
    FlowMap m;
};

// grid: x = blocks of Q * blockDim.x / U points, y = image.  U lanes per point (adjacent lanes), Q points per lane: lane t of the block
// owns points base + q * (blockDim.x / U) + t / U and the units (t mod U), (t mod U) + U, ... of every coupling (nb_pair_forward_split).
template <int Q, int U>
__global__ __launch_bounds__(256) void flow_fwd_kernel(const FlowFwdArgs a) {
    const int img = blockIdx.y;
    const int N = (int)a.N, PB = blockDim.x / U;
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    flow_weights_to_lds(a.FE + (size_t)img * a.m.FE, fsm, a.m.FE);
    const RecL e{fsm};
    const int sl = threadIdx.x & (U - 1);
    int p[Q];
    float x1[Q], x2[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        p[q] = (blockIdx.x * Q + q) * PB + threadIdx.x / U;
        float xin[2];
        load_coords2(a.grid, img, a.N, p[q] < N ? p[q] : N - 1, xin);
        x1[q] = fmaf(e.f(0), xin[0], fmaf(e.f(1), xin[1], e.f(4)));
        x2[q] = fmaf(e.f(2), xin[0], fmaf(e.f(3), xin[1], e.f(5)));
    }
    for (int i = 0; i < a.m.K; ++i) {
        float u[Q];
        f32x2 st[Q], dd[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) u[q] = (i & 1) ? x2[q] : x1[q];
        if constexpr (U == 1) nb_pair_forward<false, Q>(e.at(a.m.e_nb + i * a.m.e_cp_stride), a.m.W, u, st, dd);
        else nb_pair_forward_split<false, U, Q>(e.at(a.m.e_nb + i * a.m.e_cp_stride), a.m.W, sl, u, st, dd);
        const float sc = e.f(a.m.e_scale + i);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const float ex = fast_exp(sc * st[q][0]);
            if (i & 1) x1[q] = fmaf(ex, x1[q], st[q][1]);
            else x2[q] = fmaf(ex, x2[q], st[q][1]);
        }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q)
        if (p[q] < N && sl == 0) {
            a.xd[((size_t)img * 2) * N + p[q]] = x1[q];
            a.xd[((size_t)img * 2 + 1) * N + p[q]] = x2[q];
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

template <int K, int Q, int U>
__global__ __launch_bounds__(256) void flow_bwd_points_kernel(const FlowBwdArgs a) {
    const int img = blockIdx.y;
    const int N = (int)a.N, W = a.m.W, BS = blockDim.x, PB = BS / U;
    const int sl = threadIdx.x & (U - 1);   // U lanes per point (flow_fwd_kernel): lane 0 of a point owns its outputs and sums
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    flow_weights_to_lds(a.FE + (size_t)img * a.m.FE, fsm, a.m.FE);
    const RecL e{fsm};
    int p[Q];
    bool valid[Q];
    float xin[Q][2];
    // forward, keeping the state in front of every coupling, the net outputs and their input derivatives
    float x1s[K][Q], x2s[K][Q], sv[K][Q], tv[K][Q], ev[K][Q], dsu[K][Q], dtu[K][Q];
    float x1[Q], x2[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        p[q] = (blockIdx.x * Q + q) * PB + threadIdx.x / U;
        valid[q] = p[q] < N && sl == 0;
        load_coords2(a.grid, img, a.N, p[q] < N ? p[q] : N - 1, xin[q]);
        x1[q] = fmaf(e.f(0), xin[q][0], fmaf(e.f(1), xin[q][1], e.f(4)));
        x2[q] = fmaf(e.f(2), xin[q][0], fmaf(e.f(3), xin[q][1], e.f(5)));
    }
#pragma unroll
    for (int i = 0; i < K; ++i) {
        float u[Q];
        f32x2 st[Q], dd[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            x1s[i][q] = x1[q];
            x2s[i][q] = x2[q];
            u[q] = (i & 1) ? x2[q] : x1[q];
        }
        if constexpr (U == 1) nb_pair_forward<true, Q>(e.at(a.m.e_nb + i * a.m.e_cp_stride), W, u, st, dd);
        else nb_pair_forward_split<true, U, Q>(e.at(a.m.e_nb + i * a.m.e_cp_stride), W, sl, u, st, dd);
        const float sc = e.f(a.m.e_scale + i);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            sv[i][q] = st[q][0];
            tv[i][q] = st[q][1];
            dsu[i][q] = dd[q][0];
            dtu[i][q] = dd[q][1];
            ev[i][q] = fast_exp(sc * sv[i][q]);
            if (i & 1) x1[q] = fmaf(ev[i][q], x1[q], tv[i][q]);
            else x2[q] = fmaf(ev[i][q], x2[q], tv[i][q]);
        }
    }
    float acc[3 * K + 6];  // dscale[K] | db2_s[K] | db2_t[K] | dA[4] | db[2]
#pragma unroll
    for (int k = 0; k < 3 * K + 6; ++k) acc[k] = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        float d1 = valid[q] ? a.dxd[((size_t)img * 2) * N + p[q]] : 0.f;
        float d2 = valid[q] ? a.dxd[((size_t)img * 2 + 1) * N + p[q]] : 0.f;
#pragma unroll
        for (int ii = 0; ii < K; ++ii) {
            const int i = K - 1 - ii;
            const bool odd = i & 1;
            const float u = odd ? x2s[i][q] : x1s[i][q];
            const float tpre = odd ? x1s[i][q] : x2s[i][q];
            const float dpost = odd ? d1 : d2;
            const float de = dpost * tpre;                // d/d exp(s)
            const float sc = e.f(a.m.e_scale + i);
            const float dse = de * ev[i][q];              // d/d (scale * s_raw)
            acc[i] += dse * sv[i][q];                     // d/d scale_i
            const float gqs = dse * sc * (1.f - sv[i][q] * sv[i][q]);   // d/d pre-tanh of the s net
            const float gqt = dpost * (1.f - tv[i][q] * tv[i][q]);      // d/d pre-tanh of the t net
            acc[K + i] += gqs;
            acc[2 * K + i] += gqt;
            const float du = gqs * dsu[i][q] + gqt * dtu[i][q];   // through both nets' inputs
            if (valid[q]) {
                float* pp = a.ps + (((size_t)img * K + i) * 3) * N + p[q];
                pp[0] = u;
                pp[(size_t)N] = gqs;
                pp[2 * (size_t)N] = gqt;
            }
            const float dtpre = dpost * ev[i][q];
            if (odd) {
                d1 = dtpre;
                d2 += du;
            } else {
                d2 = dtpre;
                d1 += du;
            }
        }
        // nn.Linear(2,2): y_r = sum_c A[r][c] x_c + b_r
        acc[3 * K + 0] += d1 * xin[q][0];
        acc[3 * K + 1] += d1 * xin[q][1];
        acc[3 * K + 2] += d2 * xin[q][0];
        acc[3 * K + 3] += d2 * xin[q][1];
        acc[3 * K + 4] += d1;
        acc[3 * K + 5] += d2;
    }
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
        float v = red[0][k];
        for (int w2 = 1; w2 < (BS >> 6); ++w2) v += red[w2][k];   // fixed order
        a.slab1[((size_t)img * gridDim.x + blockIdx.x) * a.S1 + k] = v;
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

// UPL units per lane: lane l owns units l, l + 64, ...; the REM units past 64 UPL (W = 130: 2) are evaluated the other way round -
// lane = point, the unit's weights wave-uniform, one wave sum at the end - instead of costing every lane a third, mostly empty, unit
// slot (18 -> 14 VALU instructions per pair of points).
template <int UPL, int REM = 0>
__global__ __launch_bounds__(256) void flow_bwd_units_kernel(const FlowUnitsArgs a) {
    // grid: x = chunk, y = coupling*2 + net, z = image; wave w of the block takes a quarter of the chunk.
    // In the relu form (above) the three effective-weight gradients of unit j need only two sums over the points,
    //   A0_j = sum_p gq_p step(pre_jp),  A1_j = sum_p gq_p u_p step(pre_jp)      (and the unit-independent G0, G1 = sum gq, gq u):
    //   db1_j = w2_j (slope G0 + (1-slope) A0_j),  dw1_j = w2_j (slope G1 + (1-slope) A1_j),
    //   dw2_j = w1_j (slope G1 + (1-slope) A1_j) + b1_j (slope G0 + (1-slope) A0_j)
    // i.e. per unit and PAIR of points one packed fma (pre), one packed step and two packed fmas.
    const int img = blockIdx.z, chunk = blockIdx.x, nb = blockIdx.y;
    const int i = nb >> 1, net = nb & 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = (int)a.N, W = a.m.W;
    const float* __restrict__ e = a.FE + (size_t)img * a.m.FE + a.m.e_nb + i * a.m.e_cp_stride + net;
    float w1[UPL], b1[UPL], w2p[UPL];
#pragma unroll
    for (int r = 0; r < UPL; ++r) {
        const int unit = r * 64 + lane;
        const bool on = unit < W;
        w1[r] = on ? e[8 * unit] * REC_UP : 0.f;          // the records are scaled (pk_fma_clamp_bcast): undo, exact
        b1[r] = on ? e[8 * unit + 2] * REC_UP : 0.f;
        w2p[r] = on ? e[8 * unit + 4] * REC_DOWN : 0.f;
    }
    // step(pre) in ONE instruction: the first layer scaled by 2^100, the fma clamped to [0, 1] (1 for every pre >= 2^-100, 0 for pre <= 0)
    f32x2 w1s[UPL], b1s[UPL];
#pragma unroll
    for (int r = 0; r < UPL; ++r) {
        w1s[r] = splat2(w1[r] * 0x1p100f);
        b1s[r] = splat2(b1[r] * 0x1p100f);
    }
    const int per_chunk = (N + a.chunks - 1) / a.chunks;
    const int per_wave = (per_chunk + 3) / 4;
    const int p0 = chunk * per_chunk + wave * per_wave;
    int p1 = p0 + per_wave;
    const int cend = (chunk + 1) * per_chunk;
    if (p1 > cend) p1 = cend;
    if (p1 > N) p1 = N;
    const float* __restrict__ pu = a.ps + (((size_t)img * a.m.K + i) * 3) * N;
    const float* __restrict__ pg = pu + (size_t)(1 + net) * N;
    f32x2 A0[UPL], A1[UPL];
#pragma unroll
    for (int r = 0; r < UPL; ++r) A0[r] = A1[r] = f32x2{0.f, 0.f};
    constexpr int REMA = REM > 0 ? REM : 1;
    float w1m[REMA], b1m[REMA], w2pm[REMA], a0m[REMA], a1m[REMA];   // leftover units (wave-uniform weights, per-lane partial sums)
#pragma unroll
    for (int m = 0; m < REMA; ++m) {
        const int unit = UPL * 64 + m;
        const bool on = REM > 0 && unit < W;
        w1m[m] = on ? e[8 * unit] * REC_UP : 0.f;
        b1m[m] = on ? e[8 * unit + 2] * REC_UP : 0.f;
        w2pm[m] = on ? e[8 * unit + 4] * REC_DOWN : 0.f;
        a0m[m] = a1m[m] = 0.f;
    }
    float g0 = 0.f, g1 = 0.f;
    // 64 points per trip: one coalesced vector load per array (the next trip's loads are already in flight), then every
    // point's (u, gq, gq u) is broadcast to the wave with v_readlane - no memory access inside the 64-point body
    float un = 0.f, gn = 0.f;
    if (p0 + lane < p1) {
        un = pu[p0 + lane];
        gn = pg[p0 + lane];
    }
    auto bc = [](float v, int k) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k)); };
    for (int p = p0; p < p1; p += 64) {
        const float uc = un, gc = gn, guc = gn * un;   // gq = 0 for the lanes past p1: those points contribute nothing
        g0 += gc;
        g1 += guc;
        if constexpr (REM > 0) {
#pragma unroll
            for (int m = 0; m < REM; ++m) {
                const float st = __builtin_amdgcn_fmed3f(fmaf(w1m[m], uc, b1m[m]) * 0x1p126f, 0.f, 1.f);   // step(pre): as step01
                a0m[m] = fmaf(gc, st, a0m[m]);
                a1m[m] = fmaf(guc, st, a1m[m]);
            }
        }
        un = 0.f;
        gn = 0.f;
        if (p + 64 + lane < p1) {
            un = pu[p + 64 + lane];
            gn = pg[p + 64 + lane];
        }
#pragma unroll
        for (int k = 0; k < 64; k += 2) {
            const f32x2 u2 = f32x2{bc(uc, k), bc(uc, k + 1)}, gq2 = f32x2{bc(gc, k), bc(gc, k + 1)};
            const f32x2 gu2 = f32x2{bc(guc, k), bc(guc, k + 1)};
#pragma unroll
            for (int r = 0; r < UPL; ++r) {
                f32x2 st;
                asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(st) : "v"(w1s[r]), "s"(u2), "v"(b1s[r]));
                A0[r] = pk_fma(gq2, st, A0[r]);
                A1[r] = pk_fma(gu2, st, A1[r]);
            }
        }
    }
    const float G0 = sum_over_groups(sum_over_points(g0)), G1 = sum_over_groups(sum_over_points(g1));   // in every lane
    constexpr int ROWS = UPL + (REM > 0 ? 1 : 0);
    __shared__ float red[4][ROWS][3][64];
    const float SL = a.m.slope / (1.f - a.m.slope);
#pragma unroll
    for (int r = 0; r < UPL; ++r) {
        const float a0 = fmaf(SL, G0, A0[r][0] + A0[r][1]), a1 = fmaf(SL, G1, A1[r][0] + A1[r][1]);   // / (1 - slope)
        red[wave][r][0][lane] = w2p[r] * a1;                                      // dw1
        red[wave][r][1][lane] = w2p[r] * a0;                                      // db1
        red[wave][r][2][lane] = (1.f - a.m.slope) * fmaf(w1[r], a1, b1[r] * a0);   // dw2
    }
    if constexpr (REM > 0) {
        float o1 = 0.f, o0 = 0.f, o2 = 0.f;   // lane m < REM ends up with leftover unit m
#pragma unroll
        for (int m = 0; m < REM; ++m) {
            const float a0 = fmaf(SL, G0, sum_over_groups(sum_over_points(a0m[m])));
            const float a1 = fmaf(SL, G1, sum_over_groups(sum_over_points(a1m[m])));
            if (lane == m) {
                o1 = w2pm[m] * a1;
                o0 = w2pm[m] * a0;
                o2 = (1.f - a.m.slope) * fmaf(w1m[m], a1, b1m[m] * a0);
            }
        }
        red[wave][UPL][0][lane] = o1;
        red[wave][UPL][1][lane] = o0;
        red[wave][UPL][2][lane] = o2;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ROWS * 192; t += 256) {
        const int r = t / 192, q = (t - r * 192) >> 6, l = t & 63;
        if (r * 64 + l >= a.Wp) continue;
        const float v = ((red[0][r][q][l] + red[1][r][q][l]) + red[2][r][q][l]) + red[3][r][q][l];
        a.slab2[((((size_t)img * a.chunks + chunk) * (a.m.K * 2) + nb) * 3 + q) * a.Wp + r * 64 + l] = v;
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
    const int32_t* status;   // per image (mode 0): frozen by a non-finite loss -> no step, like the ICNN and RealNVP updates
    const float* gscale;     // [n_images] factor on every reduced gradient (the joint step's detached clip factor), or null
    // set when the ICNN update of the same optimizer step runs in the SAME launch (cdn_update_kernel): the loss column of its slabs
    const float* loss_slabs;   // slab entry "loss" of image 0, workgroup 0 (stride loss_PS per workgroup, loss_wgs * loss_PS per image)
    int loss_wgs;
    long long loss_PS;
};

// fixed-order sum over the first 256 threads of a block (tid = linear thread index; those 256 threads are waves 0-3)
__device__ __forceinline__ float block_sum256(float v, float* sm, const int tid) {
    v = sum_over_groups(sum_over_points(v));
    __syncthreads();
    if ((tid & 63) == 0) sm[tid >> 6] = v;
    __syncthreads();
    return ((sm[0] + sm[1]) + sm[2]) + sm[3];
}
__device__ __forceinline__ float block_sum256(float v, float* sm) { return block_sum256(v, sm, threadIdx.x); }

// The loss of one image = the sum of the loss column over its slabs, in the order icnn_update_kernel uses: 16 groups (group g takes the
// workgroups g, g + 16, ...), then the groups in order.  One definition for every kernel that needs the "non-finite loss" decision.
constexpr int LOSS_GROUPS = 16;
__device__ __forceinline__ float loss_column_group_sum(const float* __restrict__ sl, const int wgs, const size_t PS, const int grp) {
    float lp = 0.f;
    int w = grp;
    for (; w + 15 * LOSS_GROUPS < wgs; w += 16 * LOSS_GROUPS) {   // 256 slabs: one trip, 16 loads in flight
        float q[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) q[k] = sl[(size_t)(w + k * LOSS_GROUPS) * PS];
#pragma unroll
        for (int k = 0; k < 16; ++k) lp += q[k];
    }
    for (; w < wgs; w += LOSS_GROUPS) lp += sl[(size_t)w * PS];
    return lp;
}

// "this image takes no optimizer step" exactly as the ICNN update of the same launch decides it: frozen by an earlier step (the flag
// the PREVIOUS launch wrote, hdr[6 + (t & 1)]) or a non-finite loss now.  Called by the first 256 threads of a block.
__device__ __forceinline__ bool frozen_in_launch(const float* __restrict__ loss_slabs, const int wgs, const long long PS,
                                                 const float* __restrict__ hdr0, const long long hdr_stride, const int t, const int img,
                                                 const int tid) {
    __shared__ float redl[LOSS_GROUPS];
    if (tid < LOSS_GROUPS) redl[tid] = loss_column_group_sum(loss_slabs + (size_t)img * wgs * PS, wgs, (size_t)PS, tid);
    __syncthreads();
    float loss_now = 0.f;
#pragma unroll
    for (int k = 0; k < LOSS_GROUPS; ++k) loss_now += redl[k];
    const bool bad_before = hdr0 != nullptr && hdr0[(size_t)img * hdr_stride + 6 + (t & 1)] != 0.f;
    return bad_before || !isfinite(loss_now);
}

__device__ __forceinline__ float adam_apply(const FlowUpdArgs& u, float gmul, float p, float g, float lr, float wd, float* m_, float* v_) {
    g = g * gmul;   // the joint step's detached clip factor (x 1.0 is exact: the plain fits are bit-identical)
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

// block nb of image img: coupling net nb (nb < 2K) or the scales + linear (nb == 2K); the first 256 threads of the block
// (tid = linear thread index).  BATCH = chunk partials requested before the first add (64 needs 192 registers).
template <int BATCH>
__device__ __forceinline__ void flow_update_body(const FlowUpdArgs& u, const int nb, const int img, const int tid) {
    __shared__ float sm[4];
    const float gmul = u.gscale != nullptr ? u.gscale[img] : 1.f;
    const FlowMap& m = u.m;
    // an image the ICNN update of this step has frozen (non-finite loss) takes no optimizer step: its effective weights are rebuilt
    // from the unchanged parameters.  ONE source of truth with the ICNN update: the "frozen" flag it has written for step t
    // into the header's double buffer (hdr[6 + ((t + 1) & 1)], one launch earlier on this stream; `status` may be NULL) - or, when
    // that update runs in THIS launch (cdn_update_kernel), the same decision from the same numbers (frozen_in_launch).
    const bool frozen = !isfinite(gmul) ||    // the joint step's composite loss was not finite (joint_step_finish_kernel)
                        ((u.mode == 0 && u.loss_slabs != nullptr)
                             ? frozen_in_launch(u.loss_slabs, u.loss_wgs, u.loss_PS, u.lr_hdr, u.hdr_stride, u.t, img, tid)
                             : ((u.lr_hdr != nullptr && u.lr_hdr[(size_t)img * u.hdr_stride + 6 + ((u.t + 1) & 1)] != 0.f) ||
                                (u.status != nullptr && u.status[img] != INR_STATUS_OK)));
    const int mode = (u.mode == 0 && frozen) ? 2 : u.mode;
    const int W = m.W, K = m.K;
    float* __restrict__ fp = u.FP + (size_t)img * m.FP;
    float* __restrict__ fe = u.FE + (size_t)img * m.FE;
    float* __restrict__ om = u.opt ? u.opt + (size_t)img * 2 * m.FP : nullptr;
    float* __restrict__ ov = om ? om + m.FP : nullptr;
    float* __restrict__ go = u.grads_out ? u.grads_out + (size_t)img * m.FP : nullptr;
    const float lr = u.lr_hdr ? u.lr_hdr[(size_t)img * u.hdr_stride + (u.t & 1)] : u.opt_desc.lr;
    if (nb < 2 * K) {
        const int pb = m.p_nb + nb * m.nb_stride;    // v1[W] g1 b1[W] v2[W] g2 b2
        const int eb = m.e_nb + (nb >> 1) * m.e_cp_stride + (nb & 1);  // this net's slots in the coupling's records
        const int i = nb >> 1, net = nb & 1;
        const bool on = tid < W;
        float v1 = on ? fp[pb + tid] : 0.f, b1 = on ? fp[pb + W + 1 + tid] : 0.f, v2 = on ? fp[pb + 2 * W + 1 + tid] : 0.f;
        float g1 = fp[pb + W], g2 = fp[pb + 3 * W + 1], b2 = fp[pb + 3 * W + 2];
        if (mode != 2) {
            // effective-weight gradients: fixed-order sums over the chunks / blocks
            float dw1 = 0.f, db1 = 0.f, dw2 = 0.f;
            if (on) {
                const float* s2 = u.slab2 + (((size_t)img * u.chunks * (K * 2) + nb) * 3) * u.Wp + tid;
                const size_t cs = (size_t)(K * 2) * 3 * u.Wp;
                if (u.chunks == 64) {   // the usual count: all partials requested before the first add (a latency chain on 2K + 1 blocks)
#pragma unroll
                    for (int c0 = 0; c0 < 64; c0 += BATCH) {
                        float q0[BATCH], q1[BATCH], q2[BATCH];
#pragma unroll
                        for (int c = 0; c < BATCH; ++c) {
                            const float* q = s2 + (c0 + c) * cs;
                            q0[c] = q[0];
                            q1[c] = q[u.Wp];
                            q2[c] = q[2 * u.Wp];
                        }
#pragma unroll
                        for (int c = 0; c < BATCH; ++c) {
                            dw1 += q0[c];
                            db1 += q1[c];
                            dw2 += q2[c];
                        }
                    }
                } else {
#pragma unroll 16
                    for (int c = 0; c < u.chunks; ++c) {
                        const float* q = s2 + c * cs;
                        dw1 += q[0];
                        db1 += q[u.Wp];
                        dw2 += q[2 * u.Wp];
                    }
                }
            }
            float db2 = 0.f;
            {
                float part = 0.f;
                for (int b = tid; b < u.blocks1; b += 256) part += u.slab1[((size_t)img * u.blocks1 + b) * u.S1 + (1 + net) * K + i];
                db2 = block_sum256(part, sm, tid);
            }
            // weight norm (dim=None): w = g v / n  =>  dg = <dw, v> / n ;  dv = g/n (dw - v <dw, v> / n^2)
            const float n1 = sqrtf(block_sum256(v1 * v1, sm, tid)), n2 = sqrtf(block_sum256(v2 * v2, sm, tid));
            const float dot1 = block_sum256(dw1 * v1, sm, tid), dot2 = block_sum256(dw2 * v2, sm, tid);
            const float dg1 = dot1 / n1, dg2 = dot2 / n2;
            const float dv1 = g1 / n1 * (dw1 - v1 * dot1 / (n1 * n1)), dv2 = g2 / n2 * (dw2 - v2 * dot2 / (n2 * n2));
            if (mode == 1) {
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
                v1 = adam_apply(u, gmul, v1, dv1, lr, 0.f, &om[pb + tid], &ov[pb + tid]);
                b1 = adam_apply(u, gmul, b1, db1, lr, 0.f, &om[pb + W + 1 + tid], &ov[pb + W + 1 + tid]);
                v2 = adam_apply(u, gmul, v2, dv2, lr, 0.f, &om[pb + 2 * W + 1 + tid], &ov[pb + 2 * W + 1 + tid]);
                fp[pb + tid] = v1;
                fp[pb + W + 1 + tid] = b1;
                fp[pb + 2 * W + 1 + tid] = v2;
            }
            // scalars: every thread computes the same values (no divergence in the block sums below); thread 0 stores
            {
                float m1 = om[pb + W], q1 = ov[pb + W], m2 = om[pb + 3 * W + 1], q2 = ov[pb + 3 * W + 1];
                float m3 = om[pb + 3 * W + 2], q3 = ov[pb + 3 * W + 2];
                g1 = adam_apply(u, gmul, g1, dg1, lr, u.wd_g, &m1, &q1);
                g2 = adam_apply(u, gmul, g2, dg2, lr, u.wd_g, &m2, &q2);
                b2 = adam_apply(u, gmul, b2, db2, lr, 0.f, &m3, &q3);
                __syncthreads();  // everyone has read the old state
                if (tid == 0) {
                    om[pb + W] = m1; ov[pb + W] = q1; om[pb + 3 * W + 1] = m2; ov[pb + 3 * W + 1] = q2;
                    om[pb + 3 * W + 2] = m3; ov[pb + 3 * W + 2] = q3;
                    fp[pb + W] = g1; fp[pb + 3 * W + 1] = g2; fp[pb + 3 * W + 2] = b2;
                }
            }
        }
        // effective weights for the next forward
        const float n1 = sqrtf(block_sum256(v1 * v1, sm, tid)), n2 = sqrtf(block_sum256(v2 * v2, sm, tid));
        const float w1e = v1 * (g1 / n1), w2e = v2 * (g2 / n2);   // 0 for the threads past W
        const float sa = block_sum256(w2e * w1e, sm, tid), sb = block_sum256(w2e * b1, sm, tid);
        if (on) {
            const float w2p = (1.f - m.slope) * w2e;
            fe[eb + 8 * tid] = w1e * REC_DOWN;       // the first layer scaled down, w2' scaled up (pk_fma_clamp_bcast): exact
            fe[eb + 8 * tid + 2] = b1 * REC_DOWN;
            fe[eb + 8 * tid + 4] = w2p * REC_UP;
            fe[eb + 8 * tid + 6] = w1e * w2p;
        }
        if (tid == 0) {
            fe[eb + 8 * W] = fmaf(m.slope, sb, b2);
            fe[eb + 8 * W + 2] = m.slope * sa;
        }
        return;
    }
    // last block: WNScale parameters of every coupling + the 2x2 linear: first reduce their per-block partial sums
    __shared__ float tot[64];
    if (mode != 2) {
        for (int k = 0; k < 3 * K + 6; ++k) {
            if (k >= K && k < 3 * K) continue;   // db2 sums belong to the coupling-net blocks
            float part = 0.f;
            for (int b = tid; b < u.blocks1; b += 256) part += u.slab1[((size_t)img * u.blocks1 + b) * u.S1 + k];
            const float t = block_sum256(part, sm, tid);
            if (tid == 0) tot[k] = t;
        }
        __syncthreads();
    }
    if (tid < K) {
        const int i = tid, pb = m.p_scale + 4 * i;  // weight, sc_bias, sc_g, sc_v
        float w = fp[pb], sb = fp[pb + 1], sg = fp[pb + 2], sv = fp[pb + 3];
        if (mode != 2) {
            const float dsc = tot[i];
            const float sgn = sv / fabsf(sv);        // weight_norm(dim=0) of a 1x1 weight: v / |v|
            const float dw = dsc * sg * sgn, dsb = dsc, dsg = dsc * w * sgn, dsv = 0.f;
            if (mode == 1) {
                go[pb] = dw; go[pb + 1] = dsb; go[pb + 2] = dsg; go[pb + 3] = dsv;
            } else {
                w = adam_apply(u, gmul, w, dw, lr, 0.f, &om[pb], &ov[pb]);
                sb = adam_apply(u, gmul, sb, dsb, lr, 0.f, &om[pb + 1], &ov[pb + 1]);
                sg = adam_apply(u, gmul, sg, dsg, lr, u.wd_g, &om[pb + 2], &ov[pb + 2]);
                sv = adam_apply(u, gmul, sv, dsv, lr, 0.f, &om[pb + 3], &ov[pb + 3]);
                fp[pb] = w; fp[pb + 1] = sb; fp[pb + 2] = sg; fp[pb + 3] = sv;
            }
        }
        if (mode != 1) fe[m.e_scale + i] = sg * (sv / fabsf(sv)) * w + sb;
    } else if (tid >= 64 && tid < 70) {
        const int k = tid - 64;  // A[0][0], A[0][1], A[1][0], A[1][1], b[0], b[1]
        float p = fp[k];
        if (mode != 2) {
            const float d = tot[3 * K + k];
            if (mode == 1) go[k] = d;
            else {
                p = adam_apply(u, gmul, p, d, lr, 0.f, &om[k], &ov[k]);
                fp[k] = p;
            }
        }
        if (mode != 1) fe[k] = p;
    }
}

// grid: x = K*2 coupling nets + 1 (block K*2: scales + linear), y = image; 256 threads
__global__ __launch_bounds__(256) void flow_update_kernel(const FlowUpdArgs u) { flow_update_body<64>(u, blockIdx.x, blockIdx.y, threadIdx.x); }

}  // namespace
