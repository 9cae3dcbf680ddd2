// joint_loss.h - FBMSJointLoss fused on the device (awesome/measures/fbms_joint_loss.py:35-59).
//
//   output (B, 2, H, W) = [seg, prior] (both after their sigmoids, awesome/model/wrapper_module.py:230-273), target (B, 1, H, W)
//   seg_loss = alpha * mean(w (.) crit(seg, target))        crit = BCELoss | SE, w = UnariesWeightedLoss._compute_weight
//                                                           (awesome/measures/unaries_weighted_loss.py:35-69; counts over the batch)
//   penalty  = beta * mean((seg - prior)^2)                 SE('mean')(output_convx, output_seg): BOTH arguments carry gradient
//   if clip_penalty and penalty > seg_loss: penalty *= (seg_loss / penalty).detach()
//   loss = seg_loss + penalty
//
// The reference decides the clip on the host (one device -> host sync per training step) and runs ~15 elementwise torch kernels
// for value + autograd.  Here: three launches, no sync - partial sums per block, one block that combines them in fixed order and
// turns them into the loss and the per-class coefficients, one pass that writes d loss / d output.  HBM-bound: reads 3 floats and
// writes 2 per pixel (12 + 8 B), 1.3 MB at 256x256.
#pragma once
#include "icnn_step.h"

namespace {

constexpr int JL_MAX_BLOCKS = 512;

struct JointLossArgs {
    const float* output;   // [B][2][HW]
    const float* target;   // [B][HW]
    float* doutput;        // [B][2][HW] or null
    float* part;           // [blocks][4] partial sums: loss over fg pixels, over bg pixels, fg count, penalty
    float* res;            // [8]: loss, seg_loss_raw, penalty_raw, clip scale | c_fg, c_bg, c_pen, (unused)
    long long hw, n;       // n = B * HW
    int blocks;
    InrJointLossDesc d;
};

__device__ __forceinline__ float jl_crit(int kind, float x, float t) {
    if (kind == INR_LOSS_SE) {
        const float d = t - x;
        return d * d;
    }
    return -(t * fmaxf(logf(x), -100.f) + (1.f - t) * fmaxf(logf(1.f - x), -100.f));   // torch.nn.BCELoss (log clamped at -100)
}
__device__ __forceinline__ float jl_dcrit(int kind, float x, float t) {
    if (kind == INR_LOSS_SE) return 2.f * (x - t);
    return (x - t) / fmaxf((1.f - x) * x, 1e-12f);                                       // binary_cross_entropy_backward
}

__device__ __forceinline__ float jl_block_sum(float v, float* sm) {   // 256 threads, fixed order
    v = sum_over_groups(sum_over_points(v));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((sm[0] + sm[1]) + sm[2]) + sm[3];
}

__global__ __launch_bounds__(256) void joint_loss_partial_kernel(const JointLossArgs a) {
    __shared__ float sm[4];
    float lfg = 0.f, lbg = 0.f, nfg = 0.f, pen = 0.f;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < a.n; e += (long long)a.blocks * 256) {
        const long long b = e / a.hw, i = e - b * a.hw;
        const float s = a.output[(2 * b) * a.hw + i], p = a.output[(2 * b + 1) * a.hw + i], t = a.target[e];
        const float l = jl_crit(a.d.kind, s, t);
        if (t < 0.5f) {
            lfg += l;
            nfg += 1.f;
        } else {
            lbg += l;
        }
        const float d = s - p;
        pen = fmaf(d, d, pen);
    }
    const float r0 = jl_block_sum(lfg, sm), r1 = jl_block_sum(lbg, sm), r2 = jl_block_sum(nfg, sm), r3 = jl_block_sum(pen, sm);
    if (threadIdx.x == 0) {
        float* o = a.part + 4 * blockIdx.x;
        o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
    }
}

__global__ __launch_bounds__(256) void joint_loss_finish_kernel(const JointLossArgs a) {
    __shared__ float sm[4];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < a.blocks; b += 256)
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += a.part[4 * b + k];
    float tot[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) tot[k] = jl_block_sum(v[k], sm);
    if (threadIdx.x != 0) return;
    const float n = (float)a.n, nfg = tot[2], nbg = n - nfg;
    float w = 1.f;
    if (a.d.weight_mode != INR_WEIGHT_NONE && nfg > 0.f) {
        const float cc = nbg / nfg;
        if (a.d.weight_mode == INR_WEIGHT_EQUAL) w = cc;
        else if (a.d.weight_mode == INR_WEIGHT_RATIO) w = (cc - 1.f) * a.d.ratio + 1.f;
        else w = rintf(cc / 10.f) + 1.f;   // sssdms (torch.round: half to even)
    }
    const float seg_raw = (w * tot[0] + tot[1]) / n, pen_raw = tot[3] / n;
    const float seg_loss = a.d.alpha * seg_raw;
    float pen = a.d.beta * pen_raw, scale = 1.f;
    if (a.d.clip_penalty && pen > seg_loss) {
        scale = seg_loss / pen;
        pen = pen * scale;
    }
    a.res[0] = seg_loss + pen;
    a.res[1] = seg_raw;
    a.res[2] = pen_raw;
    a.res[3] = scale;
    a.res[4] = a.d.alpha * w / n;              // d seg_loss / d l_i for a foreground pixel
    a.res[5] = a.d.alpha / n;                  // ... background pixel
    a.res[6] = scale * a.d.beta * 2.f / n;     // d penalty / d (seg_i - prior_i) = c_pen * (seg_i - prior_i)
    a.res[7] = nfg;
}

__global__ __launch_bounds__(256) void joint_loss_grad_kernel(const JointLossArgs a) {
    const float cfg = a.res[4], cbg = a.res[5], cpen = a.res[6];
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < a.n; e += (long long)gridDim.x * 256) {
        const long long b = e / a.hw, i = e - b * a.hw;
        const float s = a.output[(2 * b) * a.hw + i], p = a.output[(2 * b + 1) * a.hw + i], t = a.target[e];
        const float g = cpen * (s - p);
        a.doutput[(2 * b) * a.hw + i] = fmaf(t < 0.5f ? cfg : cbg, jl_dcrit(a.d.kind, s, t), g);
        a.doutput[(2 * b + 1) * a.hw + i] = -g;
    }
}

}  // namespace
