// joint_loss.h - the composite training losses of the joint segmentation + prior step, fused on the device (SURVEY.md §8(f).1,
// row a12).  One set of three kernels serves the three reference classes (InrJointLossDesc.form):
//
//   INR_JOINT_FBMS           awesome/measures/fbms_joint_loss.py:35-59        output (B, 2, H, W) = [seg, prior], target (B, 1, H, W)
//       seg_loss = alpha * mean(w (.) crit(seg, target));  penalty = beta * mean((prior - seg)^2)   (BOTH arguments carry gradient)
//       if clip_penalty and penalty > seg_loss: penalty *= (seg_loss / penalty).detach();   loss = seg_loss + penalty
//   INR_JOINT_AWESOME_IMAGE  awesome/measures/awesome_image_loss.py:34-53     same layout
//       loss = mean(w crit(seg, t)) + alpha * mean(w' pcrit(prior, t));
//       extra_penalty:  loss = gamma * loss + beta * mean((prior - (seg > 0.5))^2)                   (the indicator has no gradient)
//   INR_JOINT_AWESOME_PIXEL  awesome/measures/awesome_loss.py:45-65           output (B, n, 2): (seg, prior) per pixel, interleaved;
//       the first n_scribble pixels carry targets (B, n_scribble, 1), the rest are random pixels for the align term
//       loss = mean(w crit(seg_s, t)) + alpha * mean(w crit(prior_s, t));
//       extra_penalty and n > n_scribble:  loss = gamma * loss + beta * mean((prior_r - (seg_r > 0.5))^2) over pixels [n - n_scribble, n)
//       (the reference slices [random:] with random = n - n_scribble, awesome_loss.py:58-59; gamma = 0.1, beta = 100 are its constants)
//   crit / pcrit = BCELoss | SE, w = UnariesWeightedLoss._compute_weight (awesome/measures/unaries_weighted_loss.py:35-69; fg/bg counts
//   over the whole batch of targets) or WeightedLoss._compute_weight on class labels (weighted_loss.py:38-62, target_rule 1), after
//   the `noneclass` pixels have left the data terms (weighted_loss.py:71-74).
//
// The reference decides FBMS's clip on the host (one device -> host sync per training step) and runs ~15 elementwise torch kernels
// for value + autograd.  Here: three launches, no sync - partial sums per block, one block that combines them in fixed order and
// turns them into the loss and the gradient coefficients, one pass that writes d loss / d output.  HBM-bound: reads 3 floats and
// writes 2 per pixel (12 + 8 B), 1.3 MB at 256x256.
#pragma once
#include "icnn_step.h"

namespace {

constexpr int JL_MAX_BLOCKS = 512;
constexpr int JL_PART = 8;    // partial sums per block
constexpr int JL_RES = 16;    // result / coefficient slots

struct JointLossArgs {
    const float* output;   // image forms: [B][2][n]; pixel form: [B][n][2]
    const float* target;   // [B][n_data]
    float* doutput;        // like output, or null
    float* part;           // [blocks][JL_PART]: seg loss over fg, over the rest, fg count, penalty, prior loss over fg, over the rest,
                           //                    valid pixels (not noneclass), bg count
    float* res;            // [JL_RES], see joint_loss_finish_kernel
    long long n;           // pixels per batch item
    long long n_data;      // leading pixels of a batch item that carry the data terms (n for the image forms)
    long long pen_lo;      // first pixel of a batch item inside the penalty / align term
    long long total;       // B * n
    long long es, cs, bs;  // strides of output: element, channel, batch item
    int blocks, batch;
    int pen_on;            // the penalty / align term is part of the loss
    InrJointLossDesc d;
};

__device__ __forceinline__ float jl_crit(int kind, float x, float t) {
    if (kind == INR_LOSS_SE) {
        const float d = t - x;
        return d * d;
    }
    return -(t * bce_log(x) + (1.f - t) * bce_log(1.f - x));   // torch.nn.BCELoss (log clamped at -100, a NaN stays a NaN)
}
__device__ __forceinline__ float jl_dcrit(int kind, float x, float t) {
    if (kind == INR_LOSS_SE) return 2.f * (x - t);
    return (x - t) / fmaxf((1.f - x) * x, 1e-12f);                                       // binary_cross_entropy_backward
}
__device__ __forceinline__ float jl_class_weight(int mode, float ratio, float nfg, float nbg) {
    if (mode == INR_WEIGHT_NONE || !(nfg > 0.f)) return 1.f;
    const float cc = nbg / nfg;
    if (mode == INR_WEIGHT_EQUAL) return cc;
    if (mode == INR_WEIGHT_RATIO) return (cc - 1.f) * ratio + 1.f;
    return rintf(cc / 10.f) + 1.f;   // sssdms (torch.round: half to even)
}

// which pixels the class weight applies to / which count as background for the ratio (InrJointLossDesc.target_rule)
__device__ __forceinline__ bool jl_is_fg(int rule, float t) { return rule ? t == 0.f : t < 0.5f; }
__device__ __forceinline__ bool jl_is_bg(int rule, float t) { return rule ? t == 1.f : t >= 0.5f; }

__device__ __forceinline__ float jl_block_sum(float v, float* sm) {   // 256 threads, fixed order
    v = sum_over_groups(sum_over_points(v));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((sm[0] + sm[1]) + sm[2]) + sm[3];
}

// PRIOR = false: the segmentation-side sums only (the fused joint step: the prior's share of the loss comes out of the ICNN step
// kernel's own loss column, joint_step_finish_kernel)
template <bool PRIOR>
__global__ __launch_bounds__(256) void joint_loss_partial_kernel(const JointLossArgs a) {
    __shared__ float sm[4];
    float lfg = 0.f, lbg = 0.f, nfg = 0.f, pen = 0.f, pfg = 0.f, pbg = 0.f, nval = 0.f, nbg = 0.f;
    const bool fbms = a.d.form == INR_JOINT_FBMS;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < a.total; e += (long long)a.blocks * 256) {
        const long long b = e / a.n, i = e - b * a.n;
        const float s = a.output[b * a.bs + i * a.es];
        const float p = PRIOR ? a.output[b * a.bs + i * a.es + a.cs] : 0.f;
        if (i < a.n_data) {
            const float t = a.target[b * a.n_data + i];
            if (!(a.d.use_noneclass && t == a.d.noneclass)) {
                const float l = jl_crit(a.d.kind, s, t);
                const float lp = (PRIOR && !fbms) ? jl_crit(a.d.prior_kind, p, t) : 0.f;
                nval += 1.f;
                if (jl_is_fg(a.d.target_rule, t)) {
                    lfg += l;
                    pfg += lp;
                    nfg += 1.f;
                } else {
                    lbg += l;
                    pbg += lp;
                    if (jl_is_bg(a.d.target_rule, t)) nbg += 1.f;
                }
            }
        }
        if (PRIOR && a.pen_on && i >= a.pen_lo) {
            const float d = fbms ? s - p : p - (s > 0.5f ? 1.f : 0.f);
            pen = fmaf(d, d, pen);
        }
    }
    const float r0 = jl_block_sum(lfg, sm), r1 = jl_block_sum(lbg, sm), r2 = jl_block_sum(nfg, sm), r3 = jl_block_sum(pen, sm);
    const float r4 = jl_block_sum(pfg, sm), r5 = jl_block_sum(pbg, sm), r6 = jl_block_sum(nval, sm), r7 = jl_block_sum(nbg, sm);
    if (threadIdx.x == 0) {
        float* o = a.part + JL_PART * blockIdx.x;
        o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3; o[4] = r4; o[5] = r5; o[6] = r6; o[7] = r7;
    }
}

// res: [0] loss, [1] mean weighted crit(seg) (before alpha / gamma), [2] mean penalty (before beta), [3] FBMS clip factor,
//      [4] [5] d loss / d crit(seg_i) for a foreground / background pixel, [6] coefficient of the penalty gradient,
//      [7] fg count, [8] [9] d loss / d pcrit(prior_i) fg / bg, [10] mean weighted pcrit(prior) (before alpha)
//      tot: [0..5] as `part`, [6] valid pixels, [7] bg count
__device__ __forceinline__ void jl_finish(const JointLossArgs& a, const float (&tot)[8], float pen_sum) {
    const float nd = tot[6], npen = (float)((long long)a.batch * (a.n - a.pen_lo));
    const float nfg = tot[2], nbg = tot[7];
    const float w = jl_class_weight(a.d.weight_mode, a.d.ratio, nfg, nbg);
    const float seg_raw = (w * tot[0] + tot[1]) / nd;
    const float pen_raw = a.pen_on ? pen_sum / npen : 0.f;
    float loss, scale = 1.f, cfg, cbg, cpen, pfg = 0.f, pbg = 0.f, pri_raw = 0.f;
    if (a.d.form == INR_JOINT_FBMS) {
        const float seg_loss = a.d.alpha * seg_raw;
        float pen = a.d.beta * pen_raw;
        if (a.d.clip_penalty && pen > seg_loss) {
            scale = seg_loss / pen;
            pen = pen * scale;
        }
        loss = seg_loss + pen;
        cfg = a.d.alpha * w / nd;
        cbg = a.d.alpha / nd;
        cpen = scale * a.d.beta * 2.f / npen;      // d penalty / d (seg_i - prior_i) = cpen (seg_i - prior_i)
    } else {
        const float wp = jl_class_weight(a.d.prior_weight_mode, a.d.prior_ratio, nfg, nbg);
        pri_raw = (wp * tot[4] + tot[5]) / nd;
        const float g = a.pen_on ? a.d.gamma : 1.f;
        loss = g * (seg_raw + a.d.alpha * pri_raw) + (a.pen_on ? a.d.beta * pen_raw : 0.f);
        cfg = g * w / nd;
        cbg = g / nd;
        pfg = g * a.d.alpha * wp / nd;
        pbg = g * a.d.alpha / nd;
        cpen = a.pen_on ? a.d.beta * 2.f / npen : 0.f;   // d align / d prior_i = cpen (prior_i - [seg_i > 0.5])
    }
    a.res[0] = loss;
    a.res[1] = seg_raw;
    a.res[2] = pen_raw;
    a.res[3] = scale;
    a.res[4] = cfg;
    a.res[5] = cbg;
    a.res[6] = cpen;
    a.res[7] = nfg;
    a.res[8] = pfg;
    a.res[9] = pbg;
    a.res[10] = pri_raw;
}

__global__ __launch_bounds__(256) void joint_loss_finish_kernel(const JointLossArgs a) {
    __shared__ float sm[4];
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < a.blocks; b += 256)
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += a.part[JL_PART * b + k];
    float tot[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) tot[k] = jl_block_sum(v[k], sm);
    if (threadIdx.x != 0) return;
    jl_finish(a, tot, tot[3]);
}

// SEG_ONLY: d loss / d seg into a compact [B][n] array (the fused joint step: the prior's gradient never leaves the prior kernels);
// `prior` then comes from `logits` (the prior's pre-sigmoid output written by the step kernel)
template <bool SEG_ONLY>
__global__ __launch_bounds__(256) void joint_loss_grad_kernel(const JointLossArgs a, const float* __restrict__ logits, float* __restrict__ dseg) {
    const float cfg = a.res[4], cbg = a.res[5], cpen = a.res[6], pfg = a.res[8], pbg = a.res[9];
    const bool fbms = a.d.form == INR_JOINT_FBMS;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < a.total; e += (long long)gridDim.x * 256) {
        const long long b = e / a.n, i = e - b * a.n;
        const float s = a.output[b * a.bs + i * a.es];
        const float p = SEG_ONLY ? 1.f / (1.f + expf(-logits[e])) : a.output[b * a.bs + i * a.es + a.cs];
        float gs = 0.f, gp = 0.f;
        if (i < a.n_data) {
            const float t = a.target[b * a.n_data + i];
            if (!(a.d.use_noneclass && t == a.d.noneclass)) {
                const bool fg = jl_is_fg(a.d.target_rule, t);
                gs = (fg ? cfg : cbg) * jl_dcrit(a.d.kind, s, t);
                if (!fbms && !SEG_ONLY) gp = (fg ? pfg : pbg) * jl_dcrit(a.d.prior_kind, p, t);
            }
        }
        if (a.pen_on && i >= a.pen_lo) {
            if (fbms) {
                const float g = cpen * (s - p);
                gs += g;
                gp -= g;
            } else {
                gp = fmaf(cpen, p - (s > 0.5f ? 1.f : 0.f), gp);
            }
        }
        if (SEG_ONLY) {
            dseg[e] = gs;
        } else {
            a.doutput[b * a.bs + i * a.es] = gs;
            a.doutput[b * a.bs + i * a.es + a.cs] = gp;
        }
    }
}

}  // namespace
