// rnvp.h - HIP kernels for the RealNVP deformation of PathConnectedNet (jp-schneider/awesome awesome/model/path_connected_net.py:
// 79-85; flow built by awesome/model/net_factory.py:70-114,124-175 from the third-party package normflows==1.7.3).
//
// The arithmetic of the flow lives in normflows, which is NOT part of the reference checkout: what follows restates its
// published definitions (PARITY UNPINNED), anchored on the reference's call sites:
//   v   = a (.) x + b                                       1x1 depthwise conv "linear"      path_connected_net.py:65-77
//   z   = (v - min)/(max - min) * (new_max - new_min) + new_min      MinMax.transform       transforms/min_max.py:8-19,53-55
//   for f in 0..F-1:                                        init_realnvp                     net_factory.py:70-114
//       zm = b_f (.) z                                      nf.flows.MaskedAffineFlow(b, t, s)
//       s  = out(W2s relu(W1s zm + b1s) + b2s) ;  t likewise   nf.nets.MLP([C, hid, C], init_zeros, output_fn='tanh')
//       z  = zm + (1 - b_f) (.) (z (.) exp(s) + t)
//       z  = z (.) exp(as_f) + at_f                         nf.flows.ActNorm(C) (data-dependent init at first forward)
//   xd  = (z - new_min)/(new_max - new_min) * (max - min) + min       MinMax.inverse_transform   norm_net.py:17-27
// Masks b_f count in binary over the channels (net_factory.py:86-99): every flow has NIN = |b| inputs and NOUT = C - NIN
// outputs; C = 2 -> (1,1), C = 3 -> (1,2) or (2,1).  Only those rows/columns of the MLPs ever receive a gradient.
//
// Elementwise per point with two tiny C -> hid -> C MLPs per flow: VALU work (hid = 32: 1.5k MAC per point at C = 2,
// 4 % of the ICNN behind it).  The s and t nets share their input and are evaluated as packed f32 pairs.
//   rnvp_fwd_kernel<C>         lane = point; optionally keeps the state in front of every flow (zs) for the backward
//   rnvp_bwd_points_kernel<C>  lane = point; walks the flows backwards from zs: per point and flow the MLP inputs and the
//                              gradients at the MLP outputs (ps), block sums of the per-point-scalar gradients (b2, ActNorm, a, b)
//   rnvp_bwd_units_kernel      lane = hidden unit; streams ps and accumulates the moment sums that give dW1, db1, dW2
//   rnvp_update_kernel         fixed-order slab sums + Adam/Adamax (weight decay on every flow parameter,
//                              path_connected_net.py:924-929)
//   rnvp_actnorm_init_kernel   ActNorm's data-dependent initialisation, flow after flow
#pragma once
#include "flow.h"

#ifndef RNVP_FWD_CLAMP
#define RNVP_FWD_CLAMP 1   // the forward kernels use the clamped asm form too (0: fma + 2 x v_max; see rnvp_nets)
#endif

namespace {

constexpr int RNVP_MAX_FLOWS = 32;
constexpr int RNVP_REC = 8;     // LDS floats per hidden unit: (W1s,W1t)[NIN] (b1s,b1t) (W2s,W2t)[NOUT], 2(C+1) <= 8
constexpr int RNVP_TAIL = 16;   // per flow: (b2s,b2t)[2] | exp(as)[3] | at[3] | pad (records stay 32-byte aligned)
constexpr int RNVP_HDR = 16;    // a[3] | b[3] | pad

struct RnvpMap {
    int C, HID, F;
    int net;    // floats per MLP: W1 [HID][C] | b1 [HID] | W2 [C][HID] | b2 [C]
    int pf;     // floats per flow: s-net | t-net | as [C] | at [C]
    int RP;     // lin.weight [C] | lin.bias [C] | flows
    int fl;     // LDS floats per flow
    int LDSF;   // LDS floats of the whole image
    int HIDp;   // hidden units rounded up to 64 (row stride of the unit-gradient slabs)
    int A;      // per-point arrays per flow in ps: do_s [NOUT] | do_t [NOUT]  (<= 2(C - 1))
    int out_fn; // 0 = none, 1 = tanh
    float out_scale;
    float vmin[3], vmax[3], nmin, nmax;
    unsigned masks[RNVP_MAX_FLOWS];
};

struct FlowIdx {   // wave-uniform; scalars + selects only (an indexed member array would live in scratch)
    int nin, nout;
    int in0, in1, out0, out1;
    __host__ __device__ int in(int q) const { return q == 0 ? in0 : in1; }
    __host__ __device__ int out(int q) const { return q == 0 ? out0 : out1; }
};

template <int C>
__host__ __device__ inline FlowIdx flow_idx(unsigned mask) {
    const bool b0 = mask & 1u, b1 = (mask >> 1) & 1u, b2 = C > 2 && ((mask >> 2) & 1u);
    FlowIdx x;
    x.nin = (int)b0 + (int)b1 + (int)b2;
    x.nout = C - x.nin;
    x.in0 = b0 ? 0 : (b1 ? 1 : 2);
    x.in1 = (b0 && b1) ? 1 : 2;
    x.out0 = !b0 ? 0 : (!b1 ? 1 : 2);
    x.out1 = (!b0 && !b1) ? 1 : 2;
    return x;
}

// (fast_tanh / fast_exp: flow.h)
// coupling log-scale: tanh-bounded -> fast_exp; unbounded (output_fn none) -> libm
template <bool BOUNDED>
__device__ __forceinline__ float coupling_exp(float s) {
    if constexpr (BOUNDED) return fast_exp(s);
    else return expf(s);
}

// Channel roles of a flow as COMPILE-TIME constants of its mask (bit c set = channel c passes through and feeds the MLPs).  The
// point kernels dispatch once per flow on the (wave-uniform) mask value - 2 masks at C = 2, 6 at C = 3 - into code in which the
// state z[C], the inputs and the outputs are plain registers: no select chains, no run-time indexed per-lane arrays (which hipcc
// had placed in scratch memory: private_segment 28 bytes, scratch_store_dwordx3 / SGPR-indexed scratch_load in round 2's C = 3
// kernels; now private_segment_fixed_size == 0, checked by tests/test_abi.py on the code object).
template <int C, unsigned MASK>
struct MaskT {
    static constexpr bool b0 = (MASK & 1u) != 0, b1 = ((MASK >> 1) & 1u) != 0, b2 = C > 2 && ((MASK >> 2) & 1u) != 0;
    static constexpr int NIN = (int)b0 + (int)b1 + (int)b2, NOUT = C - NIN;
    static_assert(NIN >= 1 && NOUT >= 1, "a flow needs at least one input and one output channel");
    static constexpr int in0 = b0 ? 0 : (b1 ? 1 : 2), in1 = (b0 && b1) ? 1 : 2;
    static constexpr int out0 = !b0 ? 0 : (!b1 ? 1 : 2), out1 = (!b0 && !b1) ? 1 : 2;
    static constexpr int in(int q) { return q == 0 ? in0 : in1; }
    static constexpr int out(int q) { return q == 0 ? out0 : out1; }
};

template <unsigned V> using MaskC = std::integral_constant<unsigned, V>;
template <int C, class Fn>
__device__ __forceinline__ void with_mask(unsigned mask, Fn&& fn) {   // mask is wave-uniform (a kernel argument indexed by the flow)
    if constexpr (C == 2) {
        if (mask == 1u) fn(MaskC<1>{});
        else fn(MaskC<2>{});
    } else {
        switch (mask) {
            case 1u: fn(MaskC<1>{}); break;
            case 2u: fn(MaskC<2>{}); break;
            case 3u: fn(MaskC<3>{}); break;
            case 4u: fn(MaskC<4>{}); break;
            case 5u: fn(MaskC<5>{}); break;
            default: fn(MaskC<6>{}); break;
        }
    }
}

__device__ __forceinline__ float minmax_fwd(float v, float lo, float hi, float nlo, float nhi) {   // transforms/min_max.py:8-19
    return (v - lo) / (hi - lo) * (nhi - nlo) + nlo;
}

// flat parameters of one image -> the image the kernels read from LDS (records by flow; only the active rows/columns of
// every flow's MLPs): header, then one block of `fl` floats per flow
template <int C>
__device__ __forceinline__ void rnvp_header_image(const float* __restrict__ rp, float* dst, int tid, int nt) {
    for (int i = tid; i < RNVP_HDR; i += nt) dst[i] = i < 3 ? (i < C ? rp[i] : 0.f) : (i < 6 ? (i - 3 < C ? rp[C + i - 3] : 0.f) : 0.f);
}

template <int C>
__device__ __forceinline__ void rnvp_flow_image(const float* __restrict__ rp, float* dst, const RnvpMap& m, int f, int tid, int nt,
                                                const float s1 = REC_DOWN, const float s2 = REC_UP) {
    const FlowIdx x = flow_idx<C>(m.masks[f]);
    const float* __restrict__ pf = rp + 2 * C + (size_t)f * m.pf;
    for (int i = tid; i < m.HID * RNVP_REC; i += nt) {
        const int j = i >> 3, slot = i & 7, q = slot >> 1;
        const float* __restrict__ pn = pf + (slot & 1) * m.net;
        float v = 0.f;   // first layer x 2^-32, second layer x 2^32: relu through the clamp modifier (flow.h, pk_fma_clamp_bcast); exact
        if (q < x.nin) v = pn[j * C + x.in(q)] * s1;
        else if (q == x.nin) v = pn[m.HID * C + j] * s1;
        else if (q < x.nin + 1 + x.nout) v = pn[m.HID * C + m.HID + x.out(q - x.nin - 1) * m.HID + j] * s2;
        dst[i] = v;
    }
    for (int i = tid; i < RNVP_TAIL; i += nt) {
        float v = 0.f;
        if (i < 4) {
            const int k = i >> 1;
            if (k < x.nout) v = pf[(i & 1) * m.net + 2 * m.HID * C + m.HID + x.out(k)];
        } else if (i < 7) {
            if (i - 4 < C) v = expf(pf[2 * m.net + i - 4]);
        } else if (i < 10) {
            if (i - 7 < C) v = pf[2 * m.net + C + i - 7];
        }
        dst[m.HID * RNVP_REC + i] = v;
    }
}

template <int C>
__device__ __forceinline__ void rnvp_header_image(const float* __restrict__ rp, float* dst) { rnvp_header_image<C>(rp, dst, threadIdx.x, blockDim.x); }
template <int C>
__device__ __forceinline__ void rnvp_flow_image(const float* __restrict__ rp, float* dst, const RnvpMap& m, int f) {
    rnvp_flow_image<C>(rp, dst, m, f, threadIdx.x, blockDim.x);
}

// header + flows [f0, f1) into LDS
template <int C>
__device__ __forceinline__ void rnvp_params_to_lds(const float* __restrict__ rp, float* lds, const RnvpMap& m, int f0, int f1) {
    rnvp_header_image<C>(rp, lds);
    for (int f = f0; f < f1; ++f) rnvp_flow_image<C>(rp, lds + RNVP_HDR + (f - f0) * m.fl, m, f);
    __syncthreads();
}

// the whole image into HBM once per optimizer step (grid: x = flow, y = image): the point kernels then start with a straight
// float4 copy instead of a thousand blocks each gathering and transposing the parameters
struct RnvpPackArgs {
    const float* RP;
    float* RE;   // [n_images][LDSF]
    RnvpMap m;
    int unit_linear;   // 1: a = 1, b = 0 (flow_net alone, learn_flow_identity)
};

template <int C>
__global__ __launch_bounds__(256) void rnvp_pack_kernel(const RnvpPackArgs a) {
    const int img = blockIdx.y, f = blockIdx.x;
    const float* __restrict__ rp = a.RP + (size_t)img * a.m.RP;
    float* dst = a.RE + (size_t)img * a.m.LDSF;
    if (f == 0) {
        rnvp_header_image<C>(rp, dst);
        if (a.unit_linear && threadIdx.x < 6) dst[threadIdx.x] = threadIdx.x < 3 ? (threadIdx.x < C ? 1.f : 0.f) : 0.f;   // same threads wrote it
    }
    rnvp_flow_image<C>(rp, dst + RNVP_HDR + f * a.m.fl, a.m, f);
}

// ---- where the flow records come from ----------------------------------------------------------------------------------------------
// The records (weights of the two MLPs of a flow) are WAVE-UNIFORM: every lane of a point kernel needs the same 8 floats per hidden
// unit.  Read from LDS that is one ds_read_b128 pair per unit and wave, and the LDS return path (128 B per clock and CU) delivers
// 64 lanes x 16 B = 8 clocks per read whatever the address pattern: 64 reads per flow x 8 clocks x 16 waves per CU put the LDS pipe at
// 100 % - rounds 1-2 called these kernels "VALU-issue bound" at 0.62 VALU busy; they were LDS-return bound (the arithmetic gives
// 61 us for configs[3]'s forward, measured 53).  Uniform data belongs on the SCALAR data path: the packed image in HBM is read through
// a constant-address-space pointer with uniform addresses, which hipcc selects as s_load_dwordx4/x8 into SGPRs (scalar cache, no LDS,
// no VGPRs); VALU instructions take one SGPR pair as an operand directly.  (One SGPR pair per instruction: pre = w1 z + b1 is issued
// as v_pk_mul + v_pk_add instead of v_mov x2 + v_pk_fma.)  The one-flow init / unit-gradient kernels keep their small LDS image.
// (RecK / RecL: flow.h)

// pre-activation outputs o[q][k] = (o_s, o_t) of the two MLPs of one flow for the NOUT active output channels of Q points
// per lane (a record read from LDS serves all Q); DU: also J[q][k][m] = d o[k] / d zin[m] (packed for both nets).
// U > 1: the U adjacent lanes of a point share the unit loop (flow.h, nb_pair_forward_split): lane sl takes the units sl, sl + U, ...,
// the partial (o, J) sums are added over the lanes with DPP, every lane ends with the same values.
template <int NIN, int NOUT, bool DU, int Q, class Rec, int U = 1>
__device__ __forceinline__ void rnvp_nets(const Rec rec, int HID, const float (&zin)[Q][NIN], f32x2 (&o)[Q][NOUT],
                                          f32x2 (&J)[Q][NOUT][NIN], const int sl = 0) {
    const f32x4 tl = rec.v4(HID * RNVP_REC);
    if constexpr (U > 1) {
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int k = 0; k < NOUT; ++k) {
                o[q][k] = f32x2{0.f, 0.f};
#pragma unroll
                for (int mm = 0; mm < NIN; ++mm) J[q][k][mm] = f32x2{0.f, 0.f};
            }
        auto unit = [&](const f32x4& r0, const f32x4& r1) {
            const float v[8] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                f32x2 pre = f32x2{v[2 * NIN], v[2 * NIN + 1]};
#pragma unroll
                for (int mm = 0; mm + 1 < NIN; ++mm) pre = pk_fma(f32x2{v[2 * mm], v[2 * mm + 1]}, splat2(zin[q][mm]), pre);
                f32x2 h;   // relu(pre) 2^-32
                if constexpr (DU || RNVP_FWD_CLAMP) {
                    h = pk_fma_clamp_bcast(f32x2{v[2 * (NIN - 1)], v[2 * (NIN - 1) + 1]}, zin[q][NIN - 1], pre);
                } else {
                    pre = pk_fma(f32x2{v[2 * (NIN - 1)], v[2 * (NIN - 1) + 1]}, splat2(zin[q][NIN - 1]), pre);
                    h = f32x2{fmaxf(pre[0], 0.f), fmaxf(pre[1], 0.f)};
                }
                if (DU) {
                    const f32x2 st = step01(h);
                    f32x2 t[NIN];
#pragma unroll
                    for (int mm = 0; mm < NIN; ++mm) t[mm] = st * f32x2{v[2 * mm], v[2 * mm + 1]};
#pragma unroll
                    for (int k = 0; k < NOUT; ++k) {
                        const f32x2 w2 = f32x2{v[2 * (NIN + 1 + k)], v[2 * (NIN + 1 + k) + 1]};
                        o[q][k] = pk_fma(w2, h, o[q][k]);
#pragma unroll
                        for (int mm = 0; mm < NIN; ++mm) J[q][k][mm] = pk_fma(w2, t[mm], J[q][k][mm]);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < NOUT; ++k) o[q][k] = pk_fma(f32x2{v[2 * (NIN + 1 + k)], v[2 * (NIN + 1 + k) + 1]}, h, o[q][k]);
                }
            }
        };
        const int steps = HID / U;
        int t = 0;
        for (; t + 4 <= steps; t += 4) {   // batches of 4 units written out: hipcc does not unroll a loop around an asm statement
            f32x4 r[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r[2 * k] = rec.v4(RNVP_REC * ((t + k) * U + sl));
                r[2 * k + 1] = rec.v4(RNVP_REC * ((t + k) * U + sl) + 4);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) unit(r[2 * k], r[2 * k + 1]);
        }
        for (; t < steps; ++t) unit(rec.v4(RNVP_REC * (t * U + sl)), rec.v4(RNVP_REC * (t * U + sl) + 4));
        if (steps * U + sl < HID) unit(rec.v4(RNVP_REC * (steps * U + sl)), rec.v4(RNVP_REC * (steps * U + sl) + 4));
#pragma unroll
        for (int q = 0; q < Q; ++q) {
#pragma unroll
            for (int k = 0; k < NOUT; ++k) {
                o[q][k] = f32x2{lanes_sum<U>(o[q][k][0]), lanes_sum<U>(o[q][k][1])};
                if (DU) {
#pragma unroll
                    for (int mm = 0; mm < NIN; ++mm) J[q][k][mm] = f32x2{lanes_sum<U>(J[q][k][mm][0]), lanes_sum<U>(J[q][k][mm][1])};
                }
            }
            o[q][0] += f32x2{tl[0], tl[1]};
            if (NOUT > 1) o[q][NOUT - 1] += f32x2{tl[2], tl[3]};
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        o[q][0] = f32x2{tl[0], tl[1]};
        if (NOUT > 1) o[q][NOUT - 1] = f32x2{tl[2], tl[3]};
#pragma unroll
        for (int k = 0; k < NOUT; ++k)
#pragma unroll
            for (int mm = 0; mm < NIN; ++mm) J[q][k][mm] = f32x2{0.f, 0.f};
    }
    auto unit = [&](const f32x4& r0, const f32x4& r1) {
        const float v[8] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            f32x2 h;   // relu(pre) 2^-32: the last fma of the pre-activation carries the clamp (flow.h, pk_fma_clamp_bcast)
            if constexpr (Rec::SCALAR) {   // (experiment path) one SGPR pair per instruction: mul, (fma,) add
                f32x2 pre = f32x2{v[0], v[1]} * splat2(zin[q][0]);
#pragma unroll
                for (int mm = 1; mm < NIN; ++mm) pre = pk_fma(f32x2{v[2 * mm], v[2 * mm + 1]}, splat2(zin[q][mm]), pre);
                pre = pre + f32x2{v[2 * NIN], v[2 * NIN + 1]};
                h = f32x2{fminf(fmaxf(pre[0], 0.f), 1.f), fminf(fmaxf(pre[1], 0.f), 1.f)};
            } else {
                f32x2 pre = f32x2{v[2 * NIN], v[2 * NIN + 1]};
#pragma unroll
                for (int mm = 0; mm + 1 < NIN; ++mm) pre = pk_fma(f32x2{v[2 * mm], v[2 * mm + 1]}, splat2(zin[q][mm]), pre);
                if constexpr (DU || RNVP_FWD_CLAMP) {
                    h = pk_fma_clamp_bcast(f32x2{v[2 * (NIN - 1)], v[2 * (NIN - 1) + 1]}, zin[q][NIN - 1], pre);
                } else {
                    // (RNVP_FWD_CLAMP = 0) fma + 2 x v_max on the scaled records.  The clamped asm form was first measured SLOWER in the
                    // forward kernels (16.95 vs 14.7 us at 256x256, 55.8 vs 53.9 at configs[3]): hipcc does not unroll a loop around an asm
                    // statement, so every unit paid its own LDS round trip; with the batches of 4 written out it is faster (13.8 / 52.0 us).
                    pre = pk_fma(f32x2{v[2 * (NIN - 1)], v[2 * (NIN - 1) + 1]}, splat2(zin[q][NIN - 1]), pre);
                    h = f32x2{fmaxf(pre[0], 0.f), fmaxf(pre[1], 0.f)};
                }
            }
            if (DU) {
                const f32x2 st = step01(h);
                f32x2 t[NIN];
#pragma unroll
                for (int mm = 0; mm < NIN; ++mm) t[mm] = st * f32x2{v[2 * mm], v[2 * mm + 1]};
#pragma unroll
                for (int k = 0; k < NOUT; ++k) {
                    const f32x2 w2 = f32x2{v[2 * (NIN + 1 + k)], v[2 * (NIN + 1 + k) + 1]};
                    o[q][k] = pk_fma(w2, h, o[q][k]);
#pragma unroll
                    for (int mm = 0; mm < NIN; ++mm) J[q][k][mm] = pk_fma(w2, t[mm], J[q][k][mm]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < NOUT; ++k) o[q][k] = pk_fma(f32x2{v[2 * (NIN + 1 + k)], v[2 * (NIN + 1 + k) + 1]}, h, o[q][k]);
            }
        }
    };
    // software-pipelined over batches of 4 units, two register sets (see flow.h nb_pair_forward: the next batch's records are
    // requested before the current batch is evaluated; reads past the last unit stay inside the allocation and are unused)
    auto load4 = [&](f32x4 (&r)[8], int j0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = rec.v4(RNVP_REC * j0 + 4 * k);
    };
    auto eval4 = [&](const f32x4 (&r)[8]) {
        unit(r[0], r[1]);
        unit(r[2], r[3]);
        unit(r[4], r[5]);
        unit(r[6], r[7]);
    };
    int j = 0;
    if constexpr (DU && NIN + NOUT == 2) {   // the backward's longer arithmetic covers the next batch's LDS latency: pipelined (38.1 -> 28.9 us
                                             // at 256x256, C = 2).  Not at C = 3: configs[3] runs 4 waves per SIMD and the second register set costs one
                                             // of them (144 registers; 107.5 vs 106.8 us)
        if (HID >= 8) {
            f32x4 ra[8], rb[8];
            load4(ra, 0);
            for (; j + 8 <= HID; j += 8) {
                load4(rb, j + 4);
                __builtin_amdgcn_sched_barrier(0);
                eval4(ra);
                load4(ra, j + 8);
                __builtin_amdgcn_sched_barrier(0);
                eval4(rb);
            }
        }
    } else if constexpr (DU || RNVP_FWD_CLAMP) {   // batches of 4 written out: hipcc does not unroll a loop around an asm statement
                                                 // (step01 / the clamped fma); one register set (the C = 3 kernels run 4 waves per SIMD)
        for (; j + 4 <= HID; j += 4) {
            f32x4 r[8];
            load4(r, j);
            eval4(r);
        }
    } else {   // hipcc's own order (it also narrows the record reads to what is used): the forward kernels are faster with it
#pragma unroll 4
        for (; j < HID; ++j) unit(rec.v4(RNVP_REC * j), rec.v4(RNVP_REC * j + 4));
    }
    for (; j < HID; ++j) unit(rec.v4(RNVP_REC * j), rec.v4(RNVP_REC * j + 4));
}

// one flow on z (MaskedAffineFlow, then ActNorm unless !ACTNORM), channel roles fixed at compile time
template <int C, unsigned MASK, bool ACTNORM, bool TANH, int Q, class Rec, int U = 1>
__device__ __forceinline__ void rnvp_flow_forward_m(const Rec rec, const RnvpMap& m, float (&z)[Q][C], const int sl = 0) {
    using M = MaskT<C, MASK>;
    constexpr int NIN = M::NIN, NOUT = M::NOUT;
    float zin[Q][NIN];
    f32x2 o[Q][NOUT], J[Q][NOUT][NIN];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        zin[q][0] = z[q][M::in0];
        if constexpr (NIN > 1) zin[q][NIN - 1] = z[q][M::in1];
    }
    rnvp_nets<NIN, NOUT, false, Q, Rec, U>(rec, m.HID, zin, o, J, sl);
    const Rec tl = rec.at(m.HID * RNVP_REC);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
#pragma unroll
        for (int k = 0; k < NOUT; ++k) {
            constexpr int c0 = M::out0, c1 = M::out1;
            float s = o[q][k][0], t = o[q][k][1];
            if constexpr (TANH) {
                s = fast_tanh(s) * m.out_scale;
                t = fast_tanh(t) * m.out_scale;
            }
            const float es = coupling_exp<TANH>(s);
            if (k == 0) z[q][c0] = fmaf(z[q][c0], es, t);
            else z[q][c1 < C ? c1 : 0] = fmaf(z[q][c1 < C ? c1 : 0], es, t);
        }
        if (ACTNORM) {
#pragma unroll
            for (int c = 0; c < C; ++c) z[q][c] = fmaf(z[q][c], tl.f(4 + c), tl.f(7 + c));
        }
    }
}

// run-time mask (wave-uniform) and output function -> the specialised body
template <int C, bool ACTNORM, int Q, class Rec, int U = 1>
__device__ __forceinline__ void rnvp_flow_forward(const Rec rec, const RnvpMap& m, unsigned mask, float (&z)[Q][C], const int sl = 0) {
    with_mask<C>(mask, [&](auto mk) {
        constexpr unsigned MASK = decltype(mk)::value;
        if (m.out_fn) rnvp_flow_forward_m<C, MASK, ACTNORM, true, Q, Rec, U>(rec, m, z, sl);
        else rnvp_flow_forward_m<C, MASK, ACTNORM, false, Q, Rec, U>(rec, m, z, sl);
    });
}

template <int C>
__device__ __forceinline__ void load_coords(const InrGridDesc& gd, int img, long long N, int pc, float (&x)[C]) {
    if (gd.mode == INR_GRID_SEPARABLE) {
        const int row = pc / gd.width;
        x[0] = gd.xs[pc - row * gd.width];
        x[1] = gd.ys[row];
        if (C > 2) x[C - 1] = gd.ts[img];
    } else {
        const float* cp = gd.coords + (size_t)img * gd.coords_image_stride;
#pragma unroll
        for (int c = 0; c < C; ++c) x[c] = cp[(size_t)c * N + pc];
    }
}

struct RnvpFwdArgs {
    const float* RE;   // [n_images][LDSF] packed image (rnvp_pack_kernel)
    float* xd;         // [n_images][C][N] deformed coordinates
    float* zs;         // [n_images][F][C][N] state in front of every flow, or null
    InrGridDesc grid;
    long long N;
    RnvpMap m;
};

// grid: x = blocks of 256 Q points (lane t of the block owns points base + q 256 + t), y = image.  Q = 2 when there are
// enough points to keep every SIMD busy with half the waves: the broadcast record reads are the bottleneck of the unit loops.
template <int C, int Q, int U = 1>
__global__ __launch_bounds__(256) void rnvp_fwd_kernel(const RnvpFwdArgs a) {
    const int img = blockIdx.y;
    const int N = (int)a.N;
    extern __shared__ __attribute__((aligned(16))) float rsm[];
    flow_weights_to_lds(a.RE + (size_t)img * a.m.LDSF, rsm, a.m.LDSF);
    const RecL img_rec{rsm};   // (flow.h: why LDS, and U lanes per point for small launches)
    const int BS = blockDim.x / U;
    const int sl = threadIdx.x & (U - 1);
    const bool own = sl == 0;   // lane 0 of a point stores its values
    int p[Q];
    float z[Q][C];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        p[q] = (blockIdx.x * Q + q) * BS + threadIdx.x / U;
        float x[C];
        load_coords<C>(a.grid, img, a.N, p[q] < N ? p[q] : N - 1, x);
#pragma unroll
        for (int c = 0; c < C; ++c) z[q][c] = minmax_fwd(fmaf(img_rec.f(c), x[c], img_rec.f(3 + c)), a.m.vmin[c], a.m.vmax[c], a.m.nmin, a.m.nmax);
    }
    for (int f = 0; f < a.m.F; ++f) {
        if (a.zs != nullptr) {
#pragma unroll
            for (int q = 0; q < Q; ++q)
                if (p[q] < N && own) {
#pragma unroll
                    for (int c = 0; c < C; ++c) a.zs[(((size_t)img * a.m.F + f) * C + c) * N + p[q]] = z[q][c];
                }
        }
        rnvp_flow_forward<C, true, Q, RecL, U>(img_rec.at(RNVP_HDR + f * a.m.fl), a.m, a.m.masks[f], z, sl);
    }
#pragma unroll
    for (int q = 0; q < Q; ++q)
        if (p[q] < N && own) {
#pragma unroll
            for (int c = 0; c < C; ++c)
                a.xd[((size_t)img * C + c) * N + p[q]] = minmax_fwd(z[q][c], a.m.nmin, a.m.nmax, a.m.vmin[c], a.m.vmax[c]);
        }
}

// ---- inverse: PathConnectedNet.inverse (path_connected_net.py:107-122): NormNet.inverse (MinMax, flows reversed, MinMax^-1),
// then the inverse of the 1x1 linear.  MaskedAffineFlow.inverse: z = zm + (1 - b)(z - t(zm)) exp(-s(zm)); ActNorm.inverse:
// z = (z - t) exp(-s) -----------------------------------------------------------------------------------------------------
struct RnvpInvArgs {
    const float* RE;    // packed image
    const float* in;    // [n_images][C][N] deformed coordinates
    float* out;         // [n_images][C][N]
    long long N;
    long long in_image_stride;   // 0 = one input shared by all images
    RnvpMap m;
};

template <int C>
__global__ __launch_bounds__(256) void rnvp_inverse_kernel(const RnvpInvArgs a) {
    const int img = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int N = (int)a.N;
    extern __shared__ __attribute__((aligned(16))) float rsm[];
    flow_weights_to_lds(a.RE + (size_t)img * a.m.LDSF, rsm, a.m.LDSF);
    const RecL img_rec{rsm};
    const int pc = p < N ? p : N - 1;
    float z[1][C];
#pragma unroll
    for (int c = 0; c < C; ++c)
        z[0][c] = minmax_fwd(a.in[(size_t)img * a.in_image_stride + (size_t)c * N + pc], a.m.vmin[c], a.m.vmax[c], a.m.nmin, a.m.nmax);
    for (int f = a.m.F - 1; f >= 0; --f) {
        const RecL rec = img_rec.at(RNVP_HDR + f * a.m.fl);
        const RecL tl = rec.at(a.m.HID * RNVP_REC);
#pragma unroll
        for (int c = 0; c < C; ++c) z[0][c] = (z[0][c] - tl.f(7 + c)) / tl.f(4 + c);   // ActNorm^-1
        with_mask<C>(a.m.masks[f], [&](auto mk) {
            using M = MaskT<C, decltype(mk)::value>;
            constexpr int NIN = M::NIN, NOUT = M::NOUT;
            float zin[1][NIN];
            zin[0][0] = z[0][M::in0];
            if constexpr (NIN > 1) zin[0][NIN - 1] = z[0][M::in1];
            f32x2 o[1][NOUT], J[1][NOUT][NIN];
            rnvp_nets<NIN, NOUT, false, 1>(rec, a.m.HID, zin, o, J);   // the masked channels are unchanged by the coupling
#pragma unroll
            for (int k = 0; k < NOUT; ++k) {
                float s = o[0][k][0], t = o[0][k][1];
                if (a.m.out_fn) {   // (libm here: the inverse is not on the fit path)
                    s = tanhf(s) * a.m.out_scale;
                    t = tanhf(t) * a.m.out_scale;
                }
                const float e = expf(-s);
                constexpr int c0 = M::out0, c1 = M::out1 < C ? M::out1 : 0;
                if (k == 0) z[0][c0] = (z[0][c0] - t) * e;
                else z[0][c1] = (z[0][c1] - t) * e;
            }
        });
    }
    if (p < N) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float v = minmax_fwd(z[0][c], a.m.nmin, a.m.nmax, a.m.vmin[c], a.m.vmax[c]);
            a.out[((size_t)img * C + c) * N + p] = (1.f / img_rec.f(c)) * (v - img_rec.f(3 + c));   // inverse_1b1_linear (:87-104)
        }
    }
}

// ---- backward, lane = point ----------------------------------------------------------------------------------------------
struct RnvpBwdArgs {
    const float* RE;
    const float* dxd;   // [n_images][C][N]
    const float* zs;    // [n_images][F][C][N]
    float* ps;          // [n_images][F][A][N]: gradients at the MLP outputs (do_s | do_t) per point and flow
    float* slab1;       // [n_images][blocks][S1]; S1 = F*4C (b2s[C] b2t[C] as[C] at[C] per flow) + 2C (a, b)
    InrGridDesc grid;
    long long N;
    RnvpMap m;
    int S1;
};

// One flow of the backward walk for Q points per lane, channel roles fixed at compile time.  In: g = d loss / d (state behind this
// flow's ActNorm); out: g = d loss / d (state in front of the flow), per-lane partial sums of the per-point-scalar gradients in
// acc (db2s [C] | db2t [C] | das [C] | dat [C]), and per point the gradients at the MLP outputs in ps (do_s [NOUT] | do_t [NOUT]).
template <int C, unsigned MASK, bool TANH, int Q, int U = 1>
__device__ __forceinline__ void rnvp_flow_backward_m(const RnvpBwdArgs& a, const RecL rec, const RecL tl, int img, int f, int N,
                                                     const int (&p)[Q], const int (&pc)[Q], const bool (&valid)[Q], float (&g)[Q][C],
                                                     float (&acc)[4 * C], const float (&z)[Q][C], const int sl = 0) {
    using M = MaskT<C, MASK>;
    constexpr int NIN = M::NIN, NOUT = M::NOUT;
    const int F = a.m.F;
    float zin[Q][NIN];   // z = the state in front of this flow (zs), loaded by the caller one flow ahead
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        zin[q][0] = z[q][M::in0];
        if constexpr (NIN > 1) zin[q][NIN - 1] = z[q][M::in1 < C ? M::in1 : 0];
    }
    f32x2 o[Q][NOUT], J[Q][NOUT][NIN];
    rnvp_nets<NIN, NOUT, true, Q, RecL, U>(rec, a.m.HID, zin, o, J, sl);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        // post-coupling state and the outputs of the nets
        float zc[C], es[NOUT], dfs[NOUT], dft[NOUT];
#pragma unroll
        for (int c = 0; c < C; ++c) zc[c] = z[q][c];
#pragma unroll
        for (int k = 0; k < NOUT; ++k) {
            float s = o[q][k][0], t = o[q][k][1];
            dfs[k] = dft[k] = 1.f;
            if constexpr (TANH) {
                const float ths = fast_tanh(s), tht = fast_tanh(t);
                s = ths * a.m.out_scale;
                t = tht * a.m.out_scale;
                dfs[k] = fmaf(-ths, ths, 1.f) * a.m.out_scale;
                dft[k] = fmaf(-tht, tht, 1.f) * a.m.out_scale;
            }
            es[k] = coupling_exp<TANH>(s);
            zc[M::out(k)] = fmaf(z[q][M::out(k)], es[k], t);
        }
        // ActNorm: y = zc * exp(as) + at
        float gz[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float ea = tl.f(4 + c);
            acc[2 * C + c] += g[q][c] * zc[c] * ea;
            acc[3 * C + c] += g[q][c];
            gz[c] = g[q][c] * ea;
        }
        // coupling
        float dos[NOUT], dot[NOUT], gin[NIN];
#pragma unroll
        for (int mm = 0; mm < NIN; ++mm) gin[mm] = 0.f;
#pragma unroll
        for (int k = 0; k < NOUT; ++k) {
            const float gk = gz[M::out(k)], zk = z[q][M::out(k)];
            dos[k] = gk * zk * es[k] * dfs[k];
            dot[k] = gk * dft[k];
            gz[M::out(k)] = gk * es[k];
            acc[M::out(k)] += dos[k];
            acc[C + M::out(k)] += dot[k];
#pragma unroll
            for (int mm = 0; mm < NIN; ++mm) gin[mm] = fmaf(dos[k], J[q][k][mm][0], fmaf(dot[k], J[q][k][mm][1], gin[mm]));
        }
#pragma unroll
        for (int mm = 0; mm < NIN; ++mm) gz[M::in(mm)] += gin[mm];
        if (valid[q]) {   // (the MLP inputs are not stored again: the unit kernel reads them from zs)
            float* pp = a.ps + (((size_t)img * F + f) * a.m.A) * N + p[q];
#pragma unroll
            for (int k = 0; k < NOUT; ++k) {
                pp[(size_t)k * N] = dos[k];
                pp[(size_t)(NOUT + k) * N] = dot[k];
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) g[q][c] = gz[c];
    }
}

template <int C, int Q, int U = 1>
__global__ __launch_bounds__(256) void rnvp_bwd_points_kernel(const RnvpBwdArgs a) {
    const int img = blockIdx.y;
    const int N = (int)a.N, F = a.m.F;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    extern __shared__ __attribute__((aligned(16))) float rsm[];
    float* red = rsm + a.m.LDSF;   // [waves][S1]
    flow_weights_to_lds(a.RE + (size_t)img * a.m.LDSF, rsm, a.m.LDSF);
    const RecL img_rec{rsm};
    const int BS = blockDim.x, PB = BS / U;
    const int sl = threadIdx.x & (U - 1);   // U lanes per point (rnvp_fwd_kernel): lane 0 of a point owns its outputs and sums;
    int p[Q], pc[Q];                        // the others start from a zero gradient, so everything they add is zero
    bool valid[Q];
    // gradient at the flow output: through MinMax.inverse_transform
    float g[Q][C];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        p[q] = (blockIdx.x * Q + q) * PB + threadIdx.x / U;
        valid[q] = p[q] < N && sl == 0;
        pc[q] = p[q] < N ? p[q] : N - 1;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float d = valid[q] ? a.dxd[((size_t)img * C + c) * N + pc[q]] : 0.f;
            g[q][c] = d * (a.m.vmax[c] - a.m.vmin[c]) / (a.m.nmax - a.m.nmin);
        }
    }
    // The state in front of flow f (saved by the forward kernel) is requested one flow AHEAD: loaded where it is used, every flow of the
    // walk began with a trip to memory that nothing covered at one wave per SIMD (12 flows: ~8 of the kernel's 29 us at 256x256).
    auto load_state = [&](int f, float (&z)[Q][C]) {
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int c = 0; c < C; ++c) z[q][c] = a.zs[(((size_t)img * F + f) * C + c) * N + pc[q]];
    };
    float znext[Q][C];
    load_state(F - 1, znext);
    for (int f = F - 1; f >= 0; --f) {
        const RecL rec = img_rec.at(RNVP_HDR + f * a.m.fl);
        const RecL tl = rec.at(a.m.HID * RNVP_REC);
        float acc[4 * C];   // db2s[C] | db2t[C] | das[C] | dat[C]
#pragma unroll
        for (int k = 0; k < 4 * C; ++k) acc[k] = 0.f;
        float zcur[Q][C];
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int c = 0; c < C; ++c) zcur[q][c] = znext[q][c];
        if (f > 0) load_state(f - 1, znext);
        with_mask<C>(a.m.masks[f], [&](auto mk) {
            if (a.m.out_fn) rnvp_flow_backward_m<C, decltype(mk)::value, true, Q, U>(a, rec, tl, img, f, N, p, pc, valid, g, acc, zcur, sl);
            else rnvp_flow_backward_m<C, decltype(mk)::value, false, Q, U>(a, rec, tl, img, f, N, p, pc, valid, g, acc, zcur, sl);
        });
#pragma unroll
        for (int k = 0; k < 4 * C; ++k) {
            const float v = sum_over_groups(sum_over_points(acc[k]));
            if (lane == 0) red[wave * a.S1 + f * 4 * C + k] = v;
        }
    }
    // MinMax.transform and the 1x1 "linear": z0 = ((a x + b) - min)/(max - min) * (nmax - nmin) + nmin
    {
        float da[C], db[C];
#pragma unroll
        for (int c = 0; c < C; ++c) da[c] = db[c] = 0.f;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            float x[C];
            load_coords<C>(a.grid, img, a.N, pc[q], x);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float gv = g[q][c] * (a.m.nmax - a.m.nmin) / (a.m.vmax[c] - a.m.vmin[c]);
                da[c] += gv * x[c];
                db[c] += gv;
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float sa = sum_over_groups(sum_over_points(da[c])), sb = sum_over_groups(sum_over_points(db[c]));
            if (lane == 0) {
                red[wave * a.S1 + F * 4 * C + c] = sa;
                red[wave * a.S1 + F * 4 * C + C + c] = sb;
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < a.S1; k += BS) {
        float v = red[k];
        for (int w2 = 1; w2 < (BS >> 6); ++w2) v += red[w2 * a.S1 + k];   // fixed order
        a.slab1[((size_t)img * gridDim.x + blockIdx.x) * a.S1 + k] = v;
    }
}

// ---- backward, lane = hidden unit -----------------------------------------------------------------------------------------
struct RnvpUnitsArgs {
    const float* RP;
    const float* zs;     // [n_images][F][C][N] state in front of every flow (the MLP inputs are its masked channels)
    const float* ps;     // [n_images][F][A][N] gradients at the MLP outputs: do_s [NOUT] | do_t [NOUT]
    float* slab2;        // [n_images][chunks][F*2][2C+1][64]  rows: dW1[:, c] (C) | db1 | dW2[c, :] (C)
    long long N;
    RnvpMap m;
    int chunks;
};

// relu nets: with st = step(pre_j), S0[k] = sum_p do_k st and S1[k][m] = sum_p do_k zin_m st give
//   db1_j = sum_k W2[k][j] S0[k];  dW1[j][m] = sum_k W2[k][j] S1[k][m];  dW2[k][j] = sum_m W1[j][m] S1[k][m] + b1_j S0[k]
// Lane = point (4 points per lane and trip), the moment sums of a unit live in registers as packed (s-net, t-net) pairs:
// per unit and point one packed fma (pre), one packed step and NOUT (1 + NIN) packed fmas for BOTH nets, no broadcasts.
// The 4 waves of a block split the hidden units (UPW each) and all walk the block's points; one cross-lane reduction per
// accumulator at the very end.
template <int C, int NIN, int NOUT, int UPW>
__device__ __forceinline__ void rnvp_units_body(const RnvpUnitsArgs& a, const FlowIdx& x, float* lds) {
    constexpr int PPL = 4;
    const int UB = (a.m.HID + 4 * UPW - 1) / (4 * UPW);   // unit blocks of 4 UPW units
    const int img = blockIdx.z, chunk = blockIdx.x, f = blockIdx.y / UB, ub = blockIdx.y - f * UB;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = (int)a.N, HID = a.m.HID;
    const float* __restrict__ rp = a.RP + (size_t)img * a.m.RP;
    // this kernel's own image of flow f: the first layer scaled by 2^100, so that the LAST fma of a unit's pre-activation, clamped to
    // [0, 1], IS step(pre) (1 for every pre >= 2^-100, 0 for pre <= 0): one instruction less per unit and point
    rnvp_header_image<C>(rp, lds);
    rnvp_flow_image<C>(rp, lds + RNVP_HDR, a.m, f, threadIdx.x, blockDim.x, 0x1p100f, 1.f);
    __syncthreads();
    const float* rec = lds + RNVP_HDR;
    const int per_chunk = (N + a.chunks - 1) / a.chunks;
    const int p0 = chunk * per_chunk;
    int p1 = p0 + per_chunk;
    if (p1 > N) p1 = N;
    const float* __restrict__ base = a.ps + (((size_t)img * a.m.F + f) * a.m.A) * N;
    const float* __restrict__ zbase = a.zs + (((size_t)img * a.m.F + f) * C) * N;
    f32x2 S0[UPW][NOUT], S1[UPW][NOUT][NIN];
#pragma unroll
    for (int u = 0; u < UPW; ++u)
#pragma unroll
        for (int k = 0; k < NOUT; ++k) {
            S0[u][k] = f32x2{0.f, 0.f};
#pragma unroll
            for (int mm = 0; mm < NIN; ++mm) S1[u][k][mm] = f32x2{0.f, 0.f};
        }
    // The (zin, do) values of the NEXT trip are requested before the current trip's units are evaluated: a trip (256 points x UPW units)
    // is ~700 clocks of arithmetic, and loaded where they are used its four vector loads per array were an uncovered trip to memory each
    // time (2.2 waves per SIMD resident).
    float zin_n[PPL][NIN], ds_n[PPL][NOUT], dt_n[PPL][NOUT];
    auto load_trip = [&](int p) {
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            const int idx = p + q * 64 + lane;
            const bool in = idx < p1;   // do = 0 past the end: those points contribute nothing
#pragma unroll
            for (int mm = 0; mm < NIN; ++mm) zin_n[q][mm] = in ? zbase[(size_t)x.in(mm) * N + idx] : 0.f;
#pragma unroll
            for (int k = 0; k < NOUT; ++k) {
                ds_n[q][k] = in ? base[(size_t)k * N + idx] : 0.f;
                dt_n[q][k] = in ? base[(size_t)(NOUT + k) * N + idx] : 0.f;
            }
        }
    };
    load_trip(p0);
    for (int p = p0; p < p1; p += 64 * PPL) {
        float zin[PPL][NIN];
        f32x2 dd[PPL][NOUT], dz[PPL][NOUT][NIN];
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
#pragma unroll
            for (int mm = 0; mm < NIN; ++mm) zin[q][mm] = zin_n[q][mm];
#pragma unroll
            for (int k = 0; k < NOUT; ++k) {
                dd[q][k] = f32x2{ds_n[q][k], dt_n[q][k]};
#pragma unroll
                for (int mm = 0; mm < NIN; ++mm) dz[q][k][mm] = dd[q][k] * f32x2{zin[q][mm], zin[q][mm]};
            }
        }
        if (p + 64 * PPL < p1) load_trip(p + 64 * PPL);
#pragma unroll
        for (int u = 0; u < UPW; ++u) {
            const int j = (ub * 4 + wave) * UPW + u;
            if (j < HID) {   // wave-uniform
                const f32x4 r0 = *(const f32x4*)(rec + RNVP_REC * j), r1 = *(const f32x4*)(rec + RNVP_REC * j + 4);
                const float v[8] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
#pragma unroll
                for (int q = 0; q < PPL; ++q) {
                    f32x2 pre = f32x2{v[2 * NIN], v[2 * NIN + 1]};
#pragma unroll
                    for (int mm = 0; mm + 1 < NIN; ++mm) pre = pk_fma(f32x2{v[2 * mm], v[2 * mm + 1]}, splat2(zin[q][mm]), pre);
                    const f32x2 st = pk_fma_clamp_bcast(f32x2{v[2 * (NIN - 1)], v[2 * (NIN - 1) + 1]}, zin[q][NIN - 1], pre);
#pragma unroll
                    for (int k = 0; k < NOUT; ++k) {
                        S0[u][k] = pk_fma(dd[q][k], st, S0[u][k]);
#pragma unroll
                        for (int mm = 0; mm < NIN; ++mm) S1[u][k][mm] = pk_fma(dz[q][k][mm], st, S1[u][k][mm]);
                    }
                }
            }
        }
    }
    // wave totals (in every lane), then lane l < 2 UPW keeps the sums of unit l >> 1, net l & 1
    float s0[NOUT], s1[NOUT][NIN];
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
        s0[k] = 0.f;
#pragma unroll
        for (int mm = 0; mm < NIN; ++mm) s1[k][mm] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < UPW; ++u)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const bool me = lane == 2 * u + n;
#pragma unroll
            for (int k = 0; k < NOUT; ++k) {
                const float t0 = sum_over_groups(sum_over_points(S0[u][k][n]));
                s0[k] = me ? t0 : s0[k];
#pragma unroll
                for (int mm = 0; mm < NIN; ++mm) {
                    const float t1 = sum_over_groups(sum_over_points(S1[u][k][mm][n]));
                    s1[k][mm] = me ? t1 : s1[k][mm];
                }
            }
        }
    const int j = (ub * 4 + wave) * UPW + (lane >> 1), net = lane & 1;
    if (lane < 2 * UPW && j < HID) {
        const float* __restrict__ pn = rp + 2 * C + (size_t)f * a.m.pf + net * a.m.net;
        float w1[NIN], w2[NOUT];
        const float b1 = pn[HID * C + j];
#pragma unroll
        for (int mm = 0; mm < NIN; ++mm) w1[mm] = pn[j * C + x.in(mm)];
#pragma unroll
        for (int k = 0; k < NOUT; ++k) w2[k] = pn[HID * C + HID + x.out(k) * HID + j];
        float rows[2 * C + 1];   // dW1[:, c] (C) | db1 | dW2[c, :] (C); rows of inactive channels stay 0
#pragma unroll
        for (int r = 0; r < 2 * C + 1; ++r) rows[r] = 0.f;
        float db1 = 0.f;
#pragma unroll
        for (int k = 0; k < NOUT; ++k) {
            db1 = fmaf(w2[k], s0[k], db1);
            float dw2 = b1 * s0[k];
#pragma unroll
            for (int mm = 0; mm < NIN; ++mm) dw2 = fmaf(w1[mm], s1[k][mm], dw2);
#pragma unroll
            for (int c = 0; c < C; ++c)
                if (c == x.out(k)) rows[C + 1 + c] = dw2;
        }
#pragma unroll
        for (int mm = 0; mm < NIN; ++mm) {
            float dw1 = 0.f;
#pragma unroll
            for (int k = 0; k < NOUT; ++k) dw1 = fmaf(w2[k], s1[k][mm], dw1);
#pragma unroll
            for (int c = 0; c < C; ++c)
                if (c == x.in(mm)) rows[c] = dw1;
        }
        rows[C] = db1;
        float* __restrict__ dst = a.slab2 + ((((size_t)img * a.chunks + chunk) * (a.m.F * 2) + f * 2 + net) * (2 * C + 1)) * a.m.HIDp + j;
#pragma unroll
        for (int r = 0; r < 2 * C + 1; ++r) dst[r * a.m.HIDp] = rows[r];
    }
}

template <int C, int UPW>
__global__ __launch_bounds__(256) void rnvp_bwd_units_kernel(const RnvpUnitsArgs a) {
    // grid: x = chunk of points, y = flow * unit blocks + unit block, z = image; wave w of unit block ub owns hidden units
    // [(4 ub + w) UPW, (4 ub + w + 1) UPW)
    extern __shared__ __attribute__((aligned(16))) float rsm[];
    const int UBk = (a.m.HID + 4 * UPW - 1) / (4 * UPW);
    const FlowIdx x = flow_idx<C>(a.m.masks[blockIdx.y / UBk]);
    if (C == 2) rnvp_units_body<C, 1, 1, UPW>(a, x, rsm);
    else if (x.nin == 1) rnvp_units_body<C, 1, C == 2 ? 1 : 2, UPW>(a, x, rsm);
    else rnvp_units_body<C, C == 2 ? 1 : 2, 1, UPW>(a, x, rsm);
}

// ---- learn_flow_identity: SE('mean')(flow_net(x), x)  (path_connected_net.py:155-250) -------------------------------------
struct RnvpIdArgs {
    const float* xd;   // [n_images][C][N] flow output
    float* dxd;        // [n_images][C][N] d loss / d xd = 2 (xd - x) / (C N)
    float* lossp;      // [n_images][blocks] partial sums of (xd - x)^2
    InrGridDesc grid;
    long long N;
};

template <int C>
__global__ __launch_bounds__(256) void rnvp_identity_loss_kernel(const RnvpIdArgs a) {
    __shared__ float sm[4];
    const int img = blockIdx.y, N = (int)a.N;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool valid = p < N;
    float x[C];
    load_coords<C>(a.grid, img, a.N, valid ? p : N - 1, x);
    const float sc = 2.f / ((float)C * (float)N);
    float part = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const size_t o = ((size_t)img * C + c) * N + (valid ? p : 0);
        const float d = valid ? a.xd[o] - x[c] : 0.f;
        if (valid) a.dxd[o] = sc * d;
        part = fmaf(d, d, part);
    }
    const float t = block_sum256(part, sm);
    if (threadIdx.x == 0) a.lossp[(size_t)img * gridDim.x + blockIdx.x] = t;
}

// ---- reduction + optimizer --------------------------------------------------------------------------------------------------
struct RnvpUpdArgs {
    float* RP;            // [n_images][RP] (in/out)
    float* opt;           // [n_images][2*RP] exp_avg | exp_avg_sq or exp_inf (mode 0)
    float* grads_out;     // [n_images][RP] (mode 1)
    const float* slab1;
    const float* slab2;
    const float* lr_hdr;  // ICNN opt-state header: lr of this step is hdr[t & 1]; null -> opt_desc.lr
    long long hdr_stride;
    const int32_t* status;   // per image: a non-finite loss freezes the image (like the ICNN update)
    InrOptDesc opt_desc;
    RnvpMap m;
    int blocks1, S1, chunks;
    int t;
    double bc1;
    float bc2_sqrt, one_minus_b1, one_minus_b2;
    float wd_flow;        // weight decay of the flow_net group (path_connected_net.py:925); the linear has none
    int mode;             // 0 = optimizer step, 1 = gradients only
    // learn_flow_identity: the 1x1 linear is not part of the model; the loss comes as per-block partial sums
    int skip_linear;
    const float* lossp;
    int lossp_blocks;
    float loss_scale;
    float* loss_hist;     // [n_images][hist_stride] or null
    int hist_idx, hist_stride;
    float* RE;            // [n_images][LDSF] packed image to refresh after the step (mode 0), or null
    int unit_linear;      // header of RE with a = 1, b = 0 (learn_flow_identity)
    const float* gscale;  // [n_images] factor on every reduced gradient (the joint step's detached clip factor), or null
    // set when the ICNN update of the same optimizer step runs in the SAME launch (pcn_update_kernel): the loss column of its slabs
    const float* loss_slabs;   // slab entry "loss" of image 0, workgroup 0 (stride loss_PS per workgroup, loss_wgs * loss_PS per image)
    int loss_wgs;
    long long loss_PS;
};

__device__ __forceinline__ float opt_apply(const RnvpUpdArgs& u, float gmul, float p, float g, float lr, float wd, float* m_, float* v_) {
    g = g * gmul;   // the joint step's detached clip factor (x 1.0 is exact: the plain fits are bit-identical)
    if (wd != 0.f) g = __fadd_rn(g, __fmul_rn(wd, p));
    float m = *m_, v = *v_;
    m = __fadd_rn(m, __fmul_rn(u.one_minus_b1, __fsub_rn(g, m)));
    const float clr = (float)((double)lr / u.bc1);
    if (u.opt_desc.kind == INR_OPT_ADAM) {
        v = __fadd_rn(__fmul_rn(v, u.opt_desc.beta2), __fmul_rn(__fmul_rn(u.one_minus_b2, g), g));
        const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), u.bc2_sqrt), u.opt_desc.eps);
        p = __fadd_rn(p, __fdiv_rn(__fmul_rn(-clr, m), denom));
    } else {
        v = fmaxf(__fmul_rn(v, u.opt_desc.beta2), __fadd_rn(fabsf(g), u.opt_desc.eps));
        p = __fadd_rn(p, __fdiv_rn(__fmul_rn(-clr, m), v));
    }
    *m_ = m;
    *v_ = v;
    return p;
}

// block f of image img: flow f (f < F) or the linear (f == F); the first 256 threads of the block (tid = linear thread index)
template <int C>
__device__ __forceinline__ void rnvp_update_body(const RnvpUpdArgs& u, const int f, const int img, const int tid) {
    __shared__ float sm[4];
    __shared__ float tot[4 * 3 + 2 * 3];
    const float gmul = u.gscale != nullptr ? u.gscale[img] : 1.f;
    const RnvpMap& m = u.m;
    const int F = m.F, HID = m.HID;
    // one source of truth with the ICNN update: the flag it has written for step t (hdr[6 + ((t + 1) & 1)]; `status` may be NULL) -
    // or, when that update runs in THIS launch (pcn_update_kernel), the same decision from the same numbers (frozen_in_launch)
    const bool frozen = !isfinite(gmul) ||    // the joint step's composite loss was not finite (joint_step_finish_kernel)
                        (u.loss_slabs != nullptr
                             ? frozen_in_launch(u.loss_slabs, u.loss_wgs, u.loss_PS, u.lr_hdr, u.hdr_stride, u.t, img, tid)
                             : ((u.lr_hdr != nullptr && u.lr_hdr[(size_t)img * u.hdr_stride + 6 + ((u.t + 1) & 1)] != 0.f) ||
                                (u.status != nullptr && u.status[img] != INR_STATUS_OK)));
    float* __restrict__ rp = u.RP + (size_t)img * m.RP;
    float* __restrict__ om = u.opt ? u.opt + (size_t)img * 2 * m.RP : nullptr;
    float* __restrict__ ov = om ? om + m.RP : nullptr;
    float* __restrict__ go = u.grads_out ? u.grads_out + (size_t)img * m.RP : nullptr;
    const float lr = u.lr_hdr ? u.lr_hdr[(size_t)img * u.hdr_stride + (u.t & 1)] : u.opt_desc.lr;
    // per-point-scalar gradients of this block: fixed-order sums over the point blocks
    const int k0 = f < F ? f * 4 * C : F * 4 * C, nk = f < F ? 4 * C : 2 * C;
    {   // all nk sums at once: every thread walks its share of the point blocks for all scalars (their nk slots are adjacent
        // in a slab row), then one wave reduction per scalar and a single pass through LDS - instead of nk block reductions
        __shared__ float wsum[4][4 * 3];
        float part[4 * C];
#pragma unroll
        for (int k = 0; k < 4 * C; ++k) part[k] = 0.f;
        int b = tid;
        for (; b + 768 < u.blocks1; b += 1024) {   // four rows in flight (262 144 points = 1024 point blocks = one trip); same order of adds
            float v[4][4 * C];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* __restrict__ row = u.slab1 + ((size_t)img * u.blocks1 + b + 256 * r) * u.S1 + k0;
#pragma unroll
                for (int k = 0; k < 4 * C; ++k) v[r][k] = k < nk ? row[k] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 4 * C; ++k)
                    if (k < nk) part[k] += v[r][k];
        }
        for (; b < u.blocks1; b += 256) {
            const float* __restrict__ row = u.slab1 + ((size_t)img * u.blocks1 + b) * u.S1 + k0;
#pragma unroll
            for (int k = 0; k < 4 * C; ++k)
                if (k < nk) part[k] += row[k];
        }
#pragma unroll
        for (int k = 0; k < 4 * C; ++k) {
            const float v = sum_over_groups(sum_over_points(part[k]));
            if ((tid & 63) == 0) wsum[tid >> 6][k] = v;
        }
        __syncthreads();
        if (tid < nk) tot[tid] = ((wsum[0][tid] + wsum[1][tid]) + wsum[2][tid]) + wsum[3][tid];
    }
    __syncthreads();
    if (f == F && u.lossp != nullptr) {
        float part = 0.f;
        for (int b = tid; b < u.lossp_blocks; b += 256) part += u.lossp[(size_t)img * u.lossp_blocks + b];
        const float t = block_sum256(part, sm, tid);
        if (tid == 0 && u.loss_hist) u.loss_hist[(size_t)img * u.hist_stride + u.hist_idx] = t * u.loss_scale;
    }
    if (f == F && u.skip_linear) {
        if (u.mode == 0 && u.RE != nullptr) {
            float* re = u.RE + (size_t)img * m.LDSF;
            rnvp_header_image<C>(rp, re, tid, 256);
            if (u.unit_linear && tid < 6) re[tid] = tid < 3 ? (tid < C ? 1.f : 0.f) : 0.f;
        }
        return;
    }
    const int base = f < F ? 2 * C + f * m.pf : 0;
    const int count = f < F ? m.pf : 2 * C;
    // where parameter i's gradient comes from: the sum of its unit-slab partials over the chunks (s2), or a point-scalar total (g)
    const size_t cs = (size_t)(F * 2) * (2 * C + 1) * m.HIDp;
    auto source = [&](const int i, const float*& s2, float& g) {
        s2 = nullptr;
        g = 0.f;
        if (f == F) {
            g = tot[i];   // a[C] | b[C]
        } else if (i >= 2 * m.net) {
            g = tot[2 * C + (i - 2 * m.net)];   // as[C] | at[C]
        } else {
            const int net = i >= m.net ? 1 : 0, r = i - net * m.net;
            int row, j;
            if (r < HID * C) { j = r / C; row = r - j * C; }                          // W1[j][c]
            else if (r < HID * C + HID) { j = r - HID * C; row = C; }                 // b1[j]
            else if (r < 2 * HID * C + HID) { const int q = r - HID * C - HID; row = C + 1 + q / HID; j = q - (q / HID) * HID; }   // W2[c][j]
            else { row = -1; j = r - (2 * HID * C + HID); }                           // b2[c]
            if (row < 0) g = tot[net * C + j];
            else s2 = u.slab2 + ((((size_t)img * u.chunks) * (F * 2) + f * 2 + net) * (2 * C + 1) + row) * m.HIDp + j;
        }
    };
    auto apply = [&](const int i, const float g) {
        if (u.mode == 1) {
            go[base + i] = g;
        } else if (!frozen && isfinite(g)) {
            rp[base + i] = opt_apply(u, gmul, rp[base + i], g, lr, f < F ? u.wd_flow : 0.f, &om[base + i], &ov[base + i]);
        }
    };
    // This kernel is a latency chain on F + 1 blocks: a thread's (up to) two parameters are fetched together, and with the usual 64
    // chunks every partial is requested before the first add (`#pragma unroll 64` on the runtime bound falls back to a scalar loop).
    for (int i0 = tid; i0 < count; i0 += 512) {
        const int i1 = i0 + 256;
        const bool has1 = i1 < count;
        const float *sa, *sb = nullptr;
        float ga, gb = 0.f;
        source(i0, sa, ga);
        if (has1) source(i1, sb, gb);
        if (u.chunks == 64) {
            float qa[64], qb[64];
            if (sa != nullptr) {
#pragma unroll
                for (int c = 0; c < 64; ++c) qa[c] = sa[c * cs];
            }
            if (sb != nullptr) {
#pragma unroll
                for (int c = 0; c < 64; ++c) qb[c] = sb[c * cs];
            }
            if (sa != nullptr) {
#pragma unroll
                for (int c = 0; c < 64; ++c) ga += qa[c];
            }
            if (sb != nullptr) {
#pragma unroll
                for (int c = 0; c < 64; ++c) gb += qb[c];
            }
        } else {
            if (sa != nullptr) {
#pragma unroll 8
                for (int c = 0; c < u.chunks; ++c) ga += sa[c * cs];
            }
            if (sb != nullptr) {
#pragma unroll 8
                for (int c = 0; c < u.chunks; ++c) gb += sb[c * cs];
            }
        }
        apply(i0, ga);
        if (has1) apply(i1, gb);
    }
    if (u.mode == 0 && u.RE != nullptr) {
        // the next forward's packed image, straight from the parameters this block has just written (one launch less per step)
        __syncthreads();
        float* re = u.RE + (size_t)img * m.LDSF;
        if (f < F) {
            rnvp_flow_image<C>(rp, re + RNVP_HDR + f * m.fl, m, f, tid, 256);
        } else {
            rnvp_header_image<C>(rp, re, tid, 256);
            if (u.unit_linear && tid < 6) re[tid] = tid < 3 ? (tid < C ? 1.f : 0.f) : 0.f;
        }
    }
}

// grid: x = F flows + 1 (the linear), y = image; 256 threads
template <int C>
__global__ __launch_bounds__(256) void rnvp_update_kernel(const RnvpUpdArgs u) {
    rnvp_update_body<C>(u, blockIdx.x, blockIdx.y, threadIdx.x);
}

// ---- ActNorm data-dependent initialisation (nf.flows.ActNorm: first forward sets s = -log(std + 1e-6), t = -mean exp(s),
// statistics over the batch = all points of the image, unbiased std as torch.std) ---------------------------------------
struct RnvpInitArgs {
    float* RP;
    float* z;     // [n_images][C][N] scratch
    InrGridDesc grid;
    long long N;
    RnvpMap m;
};

__device__ __forceinline__ double block_sum_d(double v, double* sm) {   // 1024 threads, fixed order
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sm[w];
    return t;
}

template <int C>
__global__ __launch_bounds__(1024) void rnvp_actnorm_init_kernel(const RnvpInitArgs a) {
    const int img = blockIdx.x, N = (int)a.N;
    extern __shared__ __attribute__((aligned(16))) float rsm[];   // header + ONE flow
    __shared__ double smd[16];
    __shared__ float stat[2 * 3];
    float* __restrict__ rp = a.RP + (size_t)img * a.m.RP;
    float* __restrict__ zb = a.z + (size_t)img * C * N;
    rnvp_params_to_lds<C>(rp, rsm, a.m, 0, 0);
    for (int p = threadIdx.x; p < N; p += blockDim.x) {
        float x[C];
        load_coords<C>(a.grid, img, a.N, p, x);
#pragma unroll
        for (int c = 0; c < C; ++c)
            zb[(size_t)c * N + p] = minmax_fwd(fmaf(rsm[c], x[c], rsm[3 + c]), a.m.vmin[c], a.m.vmax[c], a.m.nmin, a.m.nmax);
    }
    for (int f = 0; f < a.m.F; ++f) {
        __syncthreads();
        rnvp_params_to_lds<C>(rp, rsm, a.m, f, f + 1);
        const unsigned mask = a.m.masks[f];
        double s[C];
#pragma unroll
        for (int c = 0; c < C; ++c) s[c] = 0.0;
        for (int p = threadIdx.x; p < N; p += blockDim.x) {
            float z[1][C];
#pragma unroll
            for (int c = 0; c < C; ++c) z[0][c] = zb[(size_t)c * N + p];
            rnvp_flow_forward<C, false, 1>(RecL{rsm + RNVP_HDR}, a.m, mask, z);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                zb[(size_t)c * N + p] = z[0][c];
                s[c] += (double)z[0][c];
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double t = block_sum_d(s[c], smd);
            if (threadIdx.x == 0) stat[c] = (float)(t / (double)N);
        }
        __syncthreads();
        double q[C];
#pragma unroll
        for (int c = 0; c < C; ++c) q[c] = 0.0;
        for (int p = threadIdx.x; p < N; p += blockDim.x) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const double d = (double)zb[(size_t)c * N + p] - (double)stat[c];
                q[c] += d * d;
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double t = block_sum_d(q[c], smd);
            if (threadIdx.x == 0) {
                const float sd = (float)sqrt(t / (double)(N > 1 ? N - 1 : 1));
                const float as = -logf(sd + 1e-6f);
                const float at = -stat[c] * expf(as);
                rp[2 * C + (size_t)f * a.m.pf + 2 * a.m.net + c] = as;
                rp[2 * C + (size_t)f * a.m.pf + 2 * a.m.net + C + c] = at;
                stat[3 + c] = as;
            }
        }
        __syncthreads();
        for (int p = threadIdx.x; p < N; p += blockDim.x) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float as = stat[3 + c];
                zb[(size_t)c * N + p] = fmaf(zb[(size_t)c * N + p], expf(as), -stat[c] * expf(as));
            }
        }
    }
}

// ---- the same initialisation spread over the chip (large grids: one block walking 262 144 points 3 x 18 times takes 13 ms) ------
// Per flow two launches over all points: (A) finish the previous flow - its (s, t) from the statistics every block re-derives from
// the per-block partial sums, in fixed order - apply them, run this flow's coupling, write per-block sums of the result;
// (B) per-block sums of squared deviations from the mean.  A last (A) launch with f = F only finishes flow F - 1.
// part: [n_images][2][nb][C] doubles (sums | squared deviations), one slot per block, no atomics: reproducible.
struct RnvpInitParArgs {
    RnvpInitArgs b;
    double* part;
    int nb, f;
};

__device__ __forceinline__ double block_sum_d256(double v, double* sm) {   // 256 threads, fixed order
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((sm[0] + sm[1]) + sm[2]) + sm[3];
}

template <int C>
__device__ __forceinline__ void rnvp_init_stats(const RnvpInitParArgs& a, int img, double (&mean)[C], double (&ssq)[C]) {
    const double* ps = a.part + ((size_t)img * 2 + 0) * a.nb * C;
    const double* pq = a.part + ((size_t)img * 2 + 1) * a.nb * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        double t = 0.0, q = 0.0;
        for (int k = 0; k < a.nb; ++k) {
            t += ps[(size_t)k * C + c];
            q += pq[(size_t)k * C + c];
        }
        mean[c] = t / (double)a.b.N;
        ssq[c] = q;
    }
}

template <int C>
__global__ __launch_bounds__(256) void rnvp_init_couple_kernel(const RnvpInitParArgs a) {
    const int img = blockIdx.y, N = (int)a.b.N, f = a.f;
    extern __shared__ __attribute__((aligned(16))) float rsm[];   // header + ONE flow
    __shared__ double smd[4];
    float* __restrict__ rp = a.b.RP + (size_t)img * a.b.m.RP;
    float* __restrict__ zb = a.b.z + (size_t)img * C * N;
    float sc[C], sh[C];   // ActNorm of the previous flow: z <- z * sc + sh
#pragma unroll
    for (int c = 0; c < C; ++c) {
        sc[c] = 1.f;
        sh[c] = 0.f;
    }
    if (f > 0) {
        double mean[C], ssq[C];
        rnvp_init_stats<C>(a, img, mean, ssq);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float mu = (float)mean[c];
            const float sd = (float)sqrt(ssq[c] / (double)(N > 1 ? N - 1 : 1));
            const float as = -logf(sd + 1e-6f);
            sc[c] = expf(as);
            sh[c] = -mu * expf(as);
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                rp[2 * C + (size_t)(f - 1) * a.b.m.pf + 2 * a.b.m.net + c] = as;
                rp[2 * C + (size_t)(f - 1) * a.b.m.pf + 2 * a.b.m.net + C + c] = sh[c];
            }
        }
    }
    if (f >= a.b.m.F) return;   // finishing launch
    rnvp_params_to_lds<C>(rp, rsm, a.b.m, f, f + 1);
    const unsigned mask = a.b.m.masks[f];
    double s[C];
#pragma unroll
    for (int c = 0; c < C; ++c) s[c] = 0.0;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < N; p += a.nb * 256) {
        float z[1][C];
        if (f == 0) {
            float xin[C];
            load_coords<C>(a.b.grid, img, a.b.N, p, xin);
#pragma unroll
            for (int c = 0; c < C; ++c)
                z[0][c] = minmax_fwd(fmaf(rsm[c], xin[c], rsm[3 + c]), a.b.m.vmin[c], a.b.m.vmax[c], a.b.m.nmin, a.b.m.nmax);
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) z[0][c] = fmaf(zb[(size_t)c * N + p], sc[c], sh[c]);
        }
        rnvp_flow_forward<C, false, 1>(RecL{rsm + RNVP_HDR}, a.b.m, mask, z);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            zb[(size_t)c * N + p] = z[0][c];
            s[c] += (double)z[0][c];
        }
    }
    double* ps = a.part + (((size_t)img * 2 + 0) * a.nb + blockIdx.x) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const double t = block_sum_d256(s[c], smd);
        if (threadIdx.x == 0) ps[c] = t;
    }
}

template <int C>
__global__ __launch_bounds__(256) void rnvp_init_var_kernel(const RnvpInitParArgs a) {
    const int img = blockIdx.y, N = (int)a.b.N;
    __shared__ double smd[4];
    const float* __restrict__ zb = a.b.z + (size_t)img * C * N;
    double mean[C], dummy[C];
    rnvp_init_stats<C>(a, img, mean, dummy);
    float mu[C];
#pragma unroll
    for (int c = 0; c < C; ++c) mu[c] = (float)mean[c];   // the statistic is kept in fp32, like the one-block kernel does
    double q[C];
#pragma unroll
    for (int c = 0; c < C; ++c) q[c] = 0.0;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < N; p += a.nb * 256) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double d = (double)zb[(size_t)c * N + p] - (double)mu[c];
            q[c] += d * d;
        }
    }
    double* pq = a.part + (((size_t)img * 2 + 1) * a.nb + blockIdx.x) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const double t = block_sum_d256(q[c], smd);
        if (threadIdx.x == 0) pq[c] = t;
    }
}

}  // namespace
