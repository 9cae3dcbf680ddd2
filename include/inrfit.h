/*
 * inrfit.h - C ABI of libinrfit.so: the MI355X (gfx950) implementation of the per-image INR fit hot path of
 * jp-schneider/awesome (dense-grid input-convex coordinate MLP: forward, loss, backward, Adam/Adamax, convexity clamp).
 *
 * The reference has no FFI: its hot path is Python/PyTorch.  Each entry point below states the reference
 * code it replaces (paths relative to the reference checkout).  The binding a maintainer would add on the reference side
 * is a ctypes stub; this repo's own host side is awesome_amd/_lib.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch tensors' data_ptr()), unless marked "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous w.r.t. the host;
 *   - return value: 0 on success, negative INR_E* for argument/launch errors (synchronous);
 *     per-image numerical failures (NaN/Inf loss) are reported in the `status` device array of inrfit_fit;
 *   - no global mutable state: thread-safe for distinct streams; one process per GPU;
 *   - all arithmetic is fp32 (the reference's AwesomeConfig.dtype default, awesome/run/awesome_config.py:193).
 *
 * Flat parameter vector of the ICNN (ConvexNet == ConvexNextNet(L=1), awesome/model/convex_net.py:10-40,177-220),
 * h = n_hidden, C = in_features, L = n_layers; row-major like the torch state_dict tensors:
 *     input.weight [h][C] | input.bias [h] |
 *     for k in 0..L-1: skip.k.ln.weight [h][h] | skip.k.ln.bias [h] | skip.k.skp.weight [h][C] |
 *     out.ln.weight [h] | out.ln.bias [1] | out.skp.weight [C]
 *   P = h*C + h + L*(h*h + h + h*C) + h + 1 + C            (17813 for h=130, C=2, L=1)
 */
#ifndef INRFIT_H
#define INRFIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: exactly the declarations between this push and the pop at the end of the file are
 * exported (nm -D shows the inrfit_* entry points and nothing else of the library's own). */
#pragma GCC visibility push(default)

#define INRFIT_ABI_VERSION 7

enum {
    INR_OK = 0,
    INR_EINVAL = -1,      /* bad argument (null pointer, non-positive size, unknown enum) */
    INR_EUNSUPPORTED = -2,/* model shape not built into this library (see inrfit_query / inrfit_supported) */
    INR_EWORKSPACE = -3,  /* workspace too small */
    INR_ELAUNCH = -4,     /* HIP launch failure (hipGetLastError) */
    INR_ENODEVICE = -5    /* no gfx950 device / wrong architecture */
};

enum { INR_MODEL_ICNN = 1 };
enum { INR_GRID_SEPARABLE = 0, INR_GRID_EXPLICIT = 1 };
enum { INR_LOSS_SE = 0, INR_LOSS_BCE = 1, INR_LOSS_EXTERNAL = 2 /* inrfit_backward only: dL/dlogit supplied */ };
enum { INR_WEIGHT_NONE = 0, INR_WEIGHT_EQUAL = 1, INR_WEIGHT_RATIO = 2, INR_WEIGHT_SSSDMS = 3, INR_WEIGHT_EXPLICIT = 4 };
enum { INR_OPT_ADAM = 0, INR_OPT_ADAMAX = 1 };
enum { INR_STATUS_OK = 0, INR_STATUS_NONFINITE = 1 };

/* Activation of layer 0 = the ENCODE stage of the coordinate network (the hidden skip layers are always relu):
 *   INR_ACT_RELU  z0 = relu(W_in x + b_in)                  the packaged models (convex_net.py:205-214; FCNet)
 *   INR_ACT_COS   z0 = cos(W_in x + b_in)                   random Fourier features cos(x @ A + b) with W_in = A^T, b_in = b
 *                                                           frozen (InrOptDesc.freeze_input) - notebooks/imageRepresentationTest.ipynb
 *                                                           cell 5, `ourSimpleNetwork` (features -> relu layers -> sigmoid)
 *   INR_ACT_SIN   z0 = sin(act_omega (W_in x + b_in))       learnable sine layer, act_omega = 10 pi in
 *                                                           notebooks/icml_teaser_code/repeating/repeating.ipynb cell 3 (`myNet`)
 * Evaluated in registers where relu is (the D tile of the layer-0 MFMA); sin / cos are v_sin_f32 / v_cos_f32 after a v_fract_f32
 * range reduction (|argument| <~ 1e3: absolute error <~ 1e-5).  The coordinate-gradient (DX) kernels exist for relu only. */
enum { INR_ACT_RELU = 0, INR_ACT_COS = 1, INR_ACT_SIN = 2 };

/* awesome/model/convex_net.py:177-203 constructor arguments (n_hidden, in_features, n_hidden_layers). */
typedef struct InrModelDesc {
    int32_t kind;        /* INR_MODEL_ICNN */
    int32_t n_hidden;    /* h */
    int32_t in_features; /* C: 2 (x,y) or 3 (x,y,t) */
    int32_t n_layers;    /* L hidden skip layers (ConvexNet: 1) */
    int32_t act0;        /* INR_ACT_* (0 = relu: a zero-initialised tail keeps old callers' meaning) */
    float act_omega;     /* INR_ACT_SIN only */
} InrModelDesc;

/* The dense coordinate grid every image is evaluated on.
 * SEPARABLE: point p = row*width + col has coords (xs[col], ys[row][, ts[image]]) - the layout produced by
 *            Transformator.get_positional_matrices (awesome/dataset/transformator.py:25-61) and the how-to grid
 *            (notebooks/how_to/convexity.ipynb cell 7); xs/ys are tiny 1-D arrays so their values are exactly the
 *            caller's (torch.linspace / arange/n), nothing is re-derived in the kernel.
 * EXPLICIT:  coords is channel-planar [C][n_points] per image (the (B,C,H,W) tensor the reference feeds through
 *            @pixelize, awesome/util/pixelize.py:31-33); image i starts at coords + i*coords_image_stride
 *            (stride 0 = one grid shared by all images). */
typedef struct InrGridDesc {
    int32_t mode;
    int32_t width, height;
    int64_t n_points; /* width*height for SEPARABLE */
    const float* xs;  /* [width]  */
    const float* ys;  /* [height] */
    const float* ts;  /* [n_images] or NULL (C == 2) */
    const float* coords;
    int64_t coords_image_stride; /* in floats */
} InrGridDesc;

/* Data term on sigmoid(logit) vs. unaries: SE (awesome/measures/se.py:21-23) or BCE (torch.nn.BCELoss), 'mean'
 * reduction, optionally re-weighted per class as UnariesWeightedLoss._compute_weight does
 * (awesome/measures/unaries_weighted_loss.py:35-69; weight applies to target < 0.5).
 * EXPLICIT: per-element coefficient is c_fg (target < 0.5) or c_bg, normalisation included - expresses the how-to
 * loop's (1-w)*mean_bg + w*mean_fg (notebooks/how_to/convexity.ipynb cell 9). */
typedef struct InrLossDesc {
    int32_t kind;
    int32_t weight_mode;
    float ratio;
    float c_fg, c_bg;
} InrLossDesc;

/* torch.optim.Adam / Adamax defaults (awesome/run/awesome_config.py:34-41; path_connected_net.py:924-929),
 * enforce_convexity after every step (convex_net.py:216-220; hook awesome/run/awesome_runner.py:294-297),
 * ReduceLROnPlateau(mode='min', rel threshold) stepped with the loss every step (path_connected_net.py:932-933,951). */
typedef struct InrOptDesc {
    int32_t kind;
    float lr, beta1, beta2, eps, weight_decay;
    int32_t clamp;
    int32_t plateau;
    int32_t plateau_patience;
    float plateau_factor, plateau_threshold, plateau_min_lr, plateau_eps;
    int32_t freeze_skips; /* 1: no skp.weight (incl. out.skp) is ever updated.  With zero skips and clamp = 0 the ICNN is the plain
                             relu MLP Linear(C,h) [Linear(h,h)]xL Linear(h,1) = FCNet(in_type='xy') (awesome/model/fc_net.py:10-59),
                             the "no prior" coordinate network of configs[0]. */
    int32_t freeze_input; /* 1: input.weight / input.bias are never updated: fixed random Fourier features (buffers A, b of
                             `ourSimpleNetwork`, imageRepresentationTest.ipynb cell 5). */
    int32_t logits_at_last_forward; /* 1: `final_logits` of the fit calls = the output of the LAST training forward, i.e. at the
                             parameters before the last optimizer step - the tensor the reference's IoU gate reads
                             (device_prior_output, path_connected_net.py:939-972; convex_diffeomorphism_net.py:405-438) - written
                             by that step's launch (no extra forward launch).  0: logits at the final parameters. */
} InrOptDesc;

/* Per-image optimizer state, `opt_state` = n_images * inrfit_opt_state_floats(model) floats:
 *   exp_avg [P] | exp_avg_sq or exp_inf [P] | header [INR_OPT_HEADER_FLOATS]
 * header: [0],[1] lr of odd/even steps (double buffer, internal), [2] current lr, [3] plateau best,
 *         [4] plateau num_bad (as float), [5] last loss, [6],[7] "frozen by a non-finite loss" flag of odd/even steps
 *         (double buffer, internal; cleared at the start of every fit call, like `status`).
 * Zero-initialise for a cold fit (step0 == 0 takes lr from InrOptDesc and resets the plateau state;
 * step0 > 0 continues from header[2..4]). */
#define INR_OPT_HEADER_FLOATS 8

/* Capabilities. max_hidden: largest n_hidden any built kernel supports; lds_bytes: LDS used by the h=130 kernel. */
int inrfit_query(int* abi_version, int* max_hidden, int* lds_bytes);
/* Static, host string: ABI version, target, the slab base and the compiler + code-generation flags this binary was built
 * with (awesome_amd/build.py passes them in).  The fit is a chaotic iteration: two builds whose VALU arithmetic rounds
 * differently (e.g. another fused-multiply-add contraction) end a 2000-step fit a few mask pixels apart, so bench.py prints
 * this string next to the parameter checksum.  The library is built with -ffp-contract=off: every fma is an explicit fmaf. */
const char* inrfit_build_info(void);
/* Gradient slabs (= workgroups) one image gets in a launch of n_images images over n_points points: min(256 / n_images, chunks),
 * at least 1.  A constant of the library (NOT the device's CU count), because it fixes the summation order of the gradients. */
int inrfit_slabs_per_image(int64_t n_points, int n_images);
/* Test / measurement hook: replace the 256 above (0 = default).  Process-global; not for production use - it changes the
 * rounding of every gradient sum and therefore the trajectory of a fit (tests/test_gpu_determinism.py bounds by how much). */
int inrfit_debug_set_slab_base(int slab_base);
/* 1 if (h, C, L) has a compiled kernel, else 0. */
int inrfit_supported(const InrModelDesc* model);
int64_t inrfit_param_count(const InrModelDesc* model);
int64_t inrfit_opt_state_floats(const InrModelDesc* model);
/* Scratch every compute call needs (parameter images in LDS layout, per-workgroup gradient slabs, loss coefficients). */
int64_t inrfit_workspace_bytes(const InrModelDesc* model, const InrGridDesc* grid, int n_images);

/* logits[n_images][n_points] = f_theta(grid).  Replaces ConvexNet/ConvexNextNet.forward
 * (awesome/model/convex_net.py:26-35, 205-214) incl. the @pixelize reshapes, for n_images parameter sets at once
 * (the PriorCache axis, awesome/util/prior_cache.py:49-59). */
int inrfit_forward(const InrModelDesc* model, const float* params, const InrGridDesc* grid, int n_images,
                   float* logits, void* workspace, int64_t workspace_bytes, void* stream);

/* loss_out[n_images], grads[n_images][P] (same flat order as params) of the data term at `params`.
 * Replaces one forward + criterion + loss.backward() of the hot loop (awesome/model/path_connected_net.py:941-948;
 * awesome/agent/torch_agent.py:470-491) - used by the autograd.Function bridge so the stock training loop still works. */
int inrfit_loss_grad(const InrModelDesc* model, const float* params, const InrGridDesc* grid, const float* targets,
                     const InrLossDesc* loss, int n_images, float* loss_out, float* grads, void* workspace,
                     int64_t workspace_bytes, void* stream);

/* grads[n_images][P] = sum_p dlogits[image][p] * d logit_p / d params: the vector-Jacobian product of the forward
 * (forward is recomputed, nothing is saved).  Replaces autograd's backward through ConvexNextNet.forward for an
 * arbitrary downstream criterion (AwesomeImageLoss, FBMSJointLoss, ... - awesome/agent/torch_agent.py:478-491).
 * dcoords (optional, may be NULL) [n_images][C][n_points]: dlogits[p] * d logit_p / d coords_p - the gradient that flows
 * on into a learned deformation of the grid (ConvexDiffeomorphismNet.forward, awesome/model/convex_diffeomorphism_net.py:
 * 173-178: ICNN(flow(Ax+b))).  Shapes with a fused kernel: relu layer 0 only; the layer-by-layer shapes (n_hidden > 130 or more than
 * two hidden layers): any layer-0 activation. */
int inrfit_backward(const InrModelDesc* model, const float* params, const InrGridDesc* grid, const float* dlogits,
                    int n_images, float* grads, float* dcoords, void* workspace, int64_t workspace_bytes, void* stream);

/* `steps` full-batch optimisation steps of n_images independent fits, entirely on device:
 *   E x { forward, loss, backward, Adam/Adamax step, clamp, plateau.step(loss) }
 * Replaces the inner loops of _prior_based_pretrain (awesome/model/path_connected_net.py:937-962), the how-to loop
 * (notebooks/how_to/convexity.ipynb cell 9) and learn_convex_net (path_connected_net.py:364-379).
 * params/opt_state are updated in place; step0 = number of optimizer steps already taken (bias correction);
 * loss_hist (optional) [n_images][steps]: loss at the parameters BEFORE each step, as the reference logs it;
 * final_logits (optional) [n_images][n_points]: logits at the final parameters; status [n_images] int32. */
int inrfit_fit(const InrModelDesc* model, float* params, float* opt_state, const InrGridDesc* grid, const float* targets,
               const InrLossDesc* loss, const InrOptDesc* opt, int n_images, int steps, int step0, float* loss_hist,
               float* final_logits, int32_t* status, void* workspace, int64_t workspace_bytes, void* stream);

/* iou[n_images]: binary Jaccard of the class selected by `invert` between (out > thr_out) and (tgt > thr_tgt);
 * 0 if the target has no member of that class.  Replaces MIOU(invert=True, average='binary')
 * (awesome/measures/miou.py:29-48) and the IoU gate of the fit loop (path_connected_net.py:964-972). */
int inrfit_miou(const float* out, const float* tgt, int n_images, int64_t n_points, float thr_out, float thr_tgt, int invert,
                float* iou, void* stream);

/* bits[n_images][ceil(n_points/64)]: (values > threshold) - or its complement with `invert` - one bit per point, LSB = lowest
 * index.  The on-device form of the mask export after evaluation (awesome/run/functions.py:2315-2361 save_result_mask writes
 * `prior > 0.5` as an image): 8 KB instead of 256 KB leave the device per 256x256 mask. */
int inrfit_pack_masks(const float* values, int n_images, int64_t n_points, float threshold, int invert, uint64_t* bits,
                      void* stream);

/* ---- path-connected prior: ICNN(flow(Ax + b)), ConvexDiffeomorphismNet (awesome/model/convex_diffeomorphism_net.py:130-188)
 * with the weight-normalised coupling flow NormalizingFlow1D(backbone='normal_block') of awesome/model/diffeomorphism_net.py:
 * 169-302 (2-D grids only, like the reference).  Flat flow parameter vector (W = width, K = num_coupling):
 *     linear.weight [2][2] | linear.bias [2] |
 *     for i in 0..K-1, for net in (s, t):  in_linear weight_v [W] | weight_g | bias [W] | out_linear weight_v [W] | weight_g | bias
 *     for i in 0..K-1 (WNScale):  weight | scale.bias | scale.weight_g | scale.weight_v
 *   FP = 6 + 2K(3W + 3) + 4K.   flow_opt_state = n_images * 2 * FP floats (exp_avg | exp_avg_sq), zero for a cold fit. */
enum { INR_FLOW_NORMAL_BLOCK = 0, /* NormalBlock: tanh(WN2 leaky_relu(WN1 u)), keys in_linear / out_linear (diffeomorphism_net.py:169-192;
                                     backbone 'normal_block' / 'residual_block' - every reference config) */
       INR_FLOW_SIMPLE = 1        /* SimpleBackbone: tanh(WN2 relu(WN1 u)), keys linear1 / linear2 (:83-104; NormalizingFlow1D's
                                     'default' backbone, i.e. what ConvexDiffeomorphismNet() builds without diffeo_args) */ };
typedef struct InrFlowDesc {
    int32_t width;        /* W <= 256 */
    int32_t num_coupling; /* K in {2, 4, 6, 8} */
    int32_t backbone;     /* INR_FLOW_NORMAL_BLOCK | INR_FLOW_SIMPLE: same flat parameter layout, the hidden activation differs */
} InrFlowDesc;

int64_t inrfit_flow_param_count(const InrFlowDesc* flow);
int64_t inrfit_cdn_workspace_bytes(const InrModelDesc* model, const InrFlowDesc* flow, const InrGridDesc* grid, int n_images);
/* out_coords[n_images][2][n_points] = flow(A x + b): ConvexDiffeomorphismNet.get_deformation (:179-184). */
int inrfit_flow_forward(const InrFlowDesc* flow, const float* flow_params, const InrGridDesc* grid, int n_images,
                        float* out_coords, void* workspace, int64_t workspace_bytes, void* stream);
/* flow_grads[n_images][FP] = sum_p dout_coords[image][c][p] * d out_coords / d flow_params: the vector-Jacobian product of
 * inrfit_flow_forward (autograd's backward through NormalizingFlow1D.forward o Linear, diffeomorphism_net.py:286-300; the forward
 * is recomputed from the grid, nothing is saved).  dout_coords [n_images][2][n_points]. */
int inrfit_flow_backward(const InrFlowDesc* flow, const float* flow_params, const InrGridDesc* grid, const float* dout_coords,
                         int n_images, float* flow_grads, void* workspace, int64_t workspace_bytes, void* stream);
/* logits[n_images][n_points] = ICNN(flow(A x + b)): ConvexDiffeomorphismNet.forward (:173-178). */
int inrfit_cdn_forward(const InrModelDesc* model, const InrFlowDesc* flow, const float* icnn_params, const float* flow_params,
                       const InrGridDesc* grid, int n_images, float* logits, void* workspace, int64_t workspace_bytes,
                       void* stream);
/* loss and the gradient w.r.t. every parameter (ICNN flat order, flow flat order) of the data term. */
int inrfit_cdn_loss_grad(const InrModelDesc* model, const InrFlowDesc* flow, const float* icnn_params,
                         const float* flow_params, const InrGridDesc* grid, const float* targets, const InrLossDesc* loss,
                         int n_images, float* loss_out, float* icnn_grads, float* flow_grads, void* workspace,
                         int64_t workspace_bytes, void* stream);
/* `steps` full-batch steps of ConvexDiffeomorphismNet.pretrain's inner loop (:405-430): forward, criterion, backward,
 * Adam over the parameter groups of get_weight_normalized_param_groups (awesome/util/torch.py:19-35: weight decay
 * `wd_on_weight_g` on every *weight_g, none elsewhere; opt->kind must be INR_OPT_ADAM), ReduceLROnPlateau, enforce_convexity. */
int inrfit_cdn_fit(const InrModelDesc* model, const InrFlowDesc* flow, float* icnn_params, float* flow_params,
                   float* icnn_opt_state, float* flow_opt_state, const InrGridDesc* grid, const float* targets,
                   const InrLossDesc* loss, const InrOptDesc* opt, float wd_on_weight_g, int n_images, int steps, int step0,
                   float* loss_hist, float* final_logits, int32_t* status, void* workspace, int64_t workspace_bytes,
                   void* stream);

/* ---- path-connected prior with the RealNVP deformation: PathConnectedNet.forward (awesome/model/path_connected_net.py:79-85)
 * as built by real_nvp_path_connected_net (awesome/model/net_factory.py:124-175):
 *     ICNN( MinMax^-1( RealNVP( MinMax( a (.) x + b ) ) ) ),   C = 2 (x, y) or 3 (x, y, t)
 * RealNVP = n_flows x [ MaskedAffineFlow(mask_f, t = MLP[C, hid, C], s = MLP[C, hid, C]), ActNorm(C) ] from the third-party
 * package normflows==1.7.3 (net_factory.py:70-114), which is NOT in the reference checkout: the kernels restate its published
 * definitions (awesome_amd/csrc/rnvp.h) - parity for this variant is UNPINNED.
 * masks[f]: bit c set = channel c passes through flow f unchanged and feeds its MLPs (net_factory.py:86-99 counts 1 .. 2^C-2
 * in binary, LSB = channel 0).  vmin/vmax/new_min/new_max: the fitted MinMax buffers (awesome/transforms/min_max.py:22-58).
 * Flat flow parameter vector (C, hid = hidden_units, F = n_flows):
 *     linear.weight [C] | linear.bias [C] |
 *     for f in 0..F-1:  for net in (s, t):  net.0.weight [hid][C] | net.0.bias [hid] | net.2.weight [C][hid] | net.2.bias [C]
 *                       then ActNorm  s [C] | t [C]
 *   RP = 2C + F (2 (2 hid C + hid + C) + 2C).   flow_opt_state = n_images * 2 * RP floats, zero for a cold fit. */
#define INR_RNVP_MAX_FLOWS 32
typedef struct InrRnvpDesc {
    int32_t channels;      /* C in {2, 3}; equals the ICNN's in_features */
    int32_t hidden_units;  /* hid <= 256 (and F * (8 hid + 12) floats must fit the 160 KB of LDS) */
    int32_t n_flows;       /* F <= 32 */
    int32_t output_fn;     /* 0 = none, 1 = tanh (flow_output_fn) */
    float output_scale;    /* flow_output_scale; 0 or 1 = none */
    float vmin[3], vmax[3];
    float new_min, new_max;
    uint32_t masks[INR_RNVP_MAX_FLOWS];
} InrRnvpDesc;

int64_t inrfit_rnvp_param_count(const InrRnvpDesc* rnvp);
int64_t inrfit_pcn_workspace_bytes(const InrModelDesc* model, const InrRnvpDesc* rnvp, const InrGridDesc* grid, int n_images);
/* ActNorm's data-dependent initialisation (first forward of nf.flows.ActNorm): flow after flow, s = -log(std + 1e-6),
 * t = -mean * exp(s) over all points of the image; writes the ActNorm entries of flow_params in place. */
int inrfit_rnvp_actnorm_init(const InrRnvpDesc* rnvp, float* flow_params, const InrGridDesc* grid, int n_images,
                             void* workspace, int64_t workspace_bytes, void* stream);
/* out_coords[n_images][C][n_points]: PathConnectedNet.get_deformation (path_connected_net.py:124-128). */
int inrfit_rnvp_forward(const InrRnvpDesc* rnvp, const float* flow_params, const InrGridDesc* grid, int n_images,
                        float* out_coords, void* workspace, int64_t workspace_bytes, void* stream);
/* out_coords[n_images][C][n_points] = linear^-1(flow_net^-1(in_coords)): PathConnectedNet.inverse (path_connected_net.py:87-122).
 * in_coords is channel-planar [C][n_points] per image; in_image_stride (floats) = 0 shares one input among all images. */
int inrfit_rnvp_inverse(const InrRnvpDesc* rnvp, const float* flow_params, const float* in_coords, int64_t in_image_stride,
                        int64_t n_points, int n_images, float* out_coords, void* workspace, int64_t workspace_bytes, void* stream);
/* `steps` steps of PathConnectedNet.learn_flow_identity (path_connected_net.py:155-250): Adamax/Adam on the flow_net parameters
 * only (weight decay opt->weight_decay, constant lr; the 1x1 linear is not part of this model) for the loss
 * SE('mean')(flow_net(x), x) on the grid x.  loss_hist (optional) [n_images][steps]. */
int inrfit_rnvp_fit_identity(const InrRnvpDesc* rnvp, float* flow_params, float* flow_opt_state, const InrGridDesc* grid,
                             const InrOptDesc* opt, int n_images, int steps, int step0, float* loss_hist, void* workspace,
                             int64_t workspace_bytes, void* stream);
/* logits[n_images][n_points]: PathConnectedNet.forward (:79-85). */
int inrfit_pcn_forward(const InrModelDesc* model, const InrRnvpDesc* rnvp, const float* icnn_params, const float* flow_params,
                       const InrGridDesc* grid, int n_images, float* logits, void* workspace, int64_t workspace_bytes,
                       void* stream);
int inrfit_pcn_loss_grad(const InrModelDesc* model, const InrRnvpDesc* rnvp, const float* icnn_params,
                         const float* flow_params, const InrGridDesc* grid, const float* targets, const InrLossDesc* loss,
                         int n_images, float* loss_out, float* icnn_grads, float* flow_grads, void* workspace,
                         int64_t workspace_bytes, void* stream);
/* `steps` full-batch steps of _prior_based_pretrain's inner loop (path_connected_net.py:937-962) for PathConnectedNet:
 * Adamax (or Adam) over the groups of :922-929 - `flow_weight_decay` on every flow_net parameter, none on the ICNN and the
 * 1x1 linear - ReduceLROnPlateau on the loss, enforce_convexity. */
int inrfit_pcn_fit(const InrModelDesc* model, const InrRnvpDesc* rnvp, float* icnn_params, float* flow_params,
                   float* icnn_opt_state, float* flow_opt_state, const InrGridDesc* grid, const float* targets,
                   const InrLossDesc* loss, const InrOptDesc* opt, float flow_weight_decay, int n_images, int steps, int step0,
                   float* loss_hist, float* final_logits, int32_t* status, void* workspace, int64_t workspace_bytes,
                   void* stream);

/* ---- joint segmentation + prior training: the composite losses on the device (row a12).
 * INR_JOINT_FBMS - FBMSJointLoss (awesome/measures/fbms_joint_loss.py:35-59).  output [batch][2][hw] = [seg, prior] probabilities (the
 * WrapperModule's output, image mode), target [batch][hw]:
 *     loss = alpha * mean(w (.) crit(seg, target)) + clip(beta * mean((seg - prior)^2))
 * crit = kind (INR_LOSS_SE | INR_LOSS_BCE = torch.nn.BCELoss), w = UnariesWeightedLoss weights by weight_mode / ratio with the
 * fg/bg counts taken over the whole batch (unaries_weighted_loss.py:35-69), clip: the penalty is rescaled to the segmentation
 * loss when it exceeds it (factor detached).  The reference takes that decision on the host - one sync per training step; here
 * nothing leaves the device.
 * INR_JOINT_AWESOME_IMAGE - AwesomeImageLoss (awesome/measures/awesome_image_loss.py:34-53), same layout:
 *     loss = mean(w crit(seg, t)) + alpha * mean(w' pcrit(prior, t));   extra_penalty (the runner's hook, awesome_runner.py:351-371):
 *     loss = gamma * loss + beta * mean((prior - (seg > 0.5))^2)
 * pcrit / w' = prior_kind / prior_weight_mode / prior_ratio.
 * INR_JOINT_AWESOME_PIXEL - AwesomeLoss (awesome/measures/awesome_loss.py:45-65), pixel mode: output [batch][hw][2] = (seg, prior) per
 * pixel, the first n_scribble pixels carry targets [batch][n_scribble], the others are random pixels of the align term:
 *     loss = mean(w crit(seg_s, t)) + alpha * mean(w crit(prior_s, t));   extra_penalty and n_scribble < hw:
 *     loss = gamma * loss + beta * mean((prior_r - (seg_r > 0.5))^2) over pixels [hw - n_scribble, hw)   (the reference's slice, :58-59;
 *     its constants are gamma = 0.1, beta = 100; prior_kind / prior_weight_mode are taken equal to kind / weight_mode).
 * loss_out [4] (device): loss, mean weighted crit(seg) (before alpha / gamma), mean penalty (before beta), FBMS's clip factor.
 * doutput (optional), laid out like output: d loss / d output, BOTH channels. */
enum { INR_JOINT_FBMS = 0, INR_JOINT_AWESOME_IMAGE = 1, INR_JOINT_AWESOME_PIXEL = 2 };
typedef struct InrJointLossDesc {
    int32_t kind;
    int32_t weight_mode;
    float ratio;
    float alpha, beta;
    int32_t clip_penalty;     /* INR_JOINT_FBMS */
    int32_t form;             /* INR_JOINT_* (0 = FBMS: a zero-initialised tail keeps ABI v4 callers' meaning) */
    int32_t prior_kind;       /* INR_JOINT_AWESOME_IMAGE: criterion / weights of the prior term */
    int32_t prior_weight_mode;
    float prior_ratio;
    float gamma;              /* AWESOME_*: factor on the data terms once extra_penalty is on */
    int32_t extra_penalty;
    int64_t n_scribble;       /* INR_JOINT_AWESOME_PIXEL: leading pixels with targets (0 = all) */
    /* ABI v7 - the targets' meaning for the data terms (a zero-initialised tail keeps the v6 meaning):
     * target_rule 0: unaries, UnariesWeightedLoss (awesome/measures/unaries_weighted_loss.py:35-69): fg = target < 0.5, bg = the rest;
     * target_rule 1: class labels, WeightedLoss (awesome/measures/weighted_loss.py:38-62): fg = target == 0, bg = target == 1 (the
     *                counts behind the class weight), any other value takes weight 1;
     * use_noneclass: pixels whose target equals `noneclass` leave the data terms altogether - sums, counts and the mean's
     *                denominator (weighted_loss.py:71-74; 2 in the reference's FBMS configs).  The penalty / align term has no
     *                targets and keeps every pixel. */
    int32_t target_rule;
    int32_t use_noneclass;
    float noneclass;
} InrJointLossDesc;
int64_t inrfit_joint_loss_workspace_bytes(int64_t n_elems);
int inrfit_joint_loss(const float* output, const float* target, int batch, int64_t hw, const InrJointLossDesc* desc,
                      float* loss_out, float* doutput, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- ONE fused training step of the joint segmentation + prior optimisation for one image:
 * TorchAgent._perform_step (awesome/agent/torch_agent.py:428-551) with the PriorManager swap (awesome/dataset/prior_dataset.py:96-110)
 * reduced to "which parameter row".  The segmentation network stays a torch module; everything behind its output runs here:
 *     prior forward (this image's parameters) -> sigmoid -> composite loss (value + d loss / d seg) -> prior backward FROM THE
 *     ACTIVATIONS OF THAT SAME PASS (the fused step kernel; no second forward) -> Adam / Adamax on the row in place -> enforce_convexity
 * seg [n_points]: the segmentation module's probabilities of this image (output channel 0 of the WrapperModule, after its sigmoid /
 * inversion, wrapper_module.py:230-273); target [n_points].  dseg [n_points] (out) = d loss / d seg for the torch backbone's own
 * backward.  prior_logits [n_points] (out, optional): the prior's pre-sigmoid output of this step's forward.  loss_out [4] as in
 * inrfit_joint_loss.  params / opt_state: ONE row (inrfit_param_count / inrfit_opt_state_floats floats); the reference keeps ONE
 * optimizer state for the single prior model whose VALUES are swapped per image (torch_agent.py:812-839), so a caller reproduces it
 * by passing the same opt_state with every row and `step` = the global step count; a state per row with its own count is the
 * per-image variant.  `step` >= 1 is torch's state['step'] after this step (bias corrections); the learning rate is opt->lr (host
 * schedulers stay on the host), opt->plateau is ignored.  A non-finite loss leaves the row untouched and sets *status (optional).
 * Forms: INR_JOINT_FBMS (any target rule / noneclass: the prior's term has no targets), and INR_JOINT_AWESOME_IMAGE while
 * extra_penalty is off (with it on the prior has two data terms: use inrfit_joint_loss + inrfit_backward) on unaries without a
 * noneclass (its prior term is evaluated by the step kernel, which reads unaries); otherwise INR_EUNSUPPORTED.
 * A non-finite COMPOSITE loss (a NaN in `seg` as much as in the prior) freezes the row.
 * How the clip stays on the device with a single prior pass: the prior's gradient is linear in the penalty's coefficient, so the step
 * kernel runs with the unclipped coefficient, its own loss column gives the penalty, and the update kernel multiplies the reduced
 * gradient by the (detached) clip factor. */
int64_t inrfit_joint_step_workspace_bytes(const InrModelDesc* model, const InrGridDesc* grid);
int inrfit_joint_step(const InrModelDesc* model, float* params, float* opt_state, const InrGridDesc* grid, const float* seg,
                      const float* target, const InrJointLossDesc* desc, const InrOptDesc* opt, int step, float* loss_out,
                      float* dseg, float* prior_logits, int32_t* status, void* workspace, int64_t workspace_bytes, void* stream);
/* The same step with the path-connected priors (ICNN behind a learned deformation of the grid): PathConnectedNet with the RealNVP
 * flow (optimizer over ICNN + flow_net + 1x1 linear, `flow_weight_decay` on the flow_net parameters) and ConvexDiffeomorphismNet
 * (weight decay `wd_on_weight_g` on every *weight_g; opt->kind must be INR_OPT_ADAM).  Workspace: inrfit_pcn_workspace_bytes /
 * inrfit_cdn_workspace_bytes (n_images = 1) + inrfit_joint_loss_workspace_bytes(n_points). */
int inrfit_pcn_joint_step(const InrModelDesc* model, const InrRnvpDesc* rnvp, float* icnn_params, float* flow_params,
                          float* icnn_opt_state, float* flow_opt_state, const InrGridDesc* grid, const float* seg,
                          const float* target, const InrJointLossDesc* desc, const InrOptDesc* opt, float flow_weight_decay,
                          int step, float* loss_out, float* dseg, float* prior_logits, int32_t* status, void* workspace,
                          int64_t workspace_bytes, void* stream);
int inrfit_cdn_joint_step(const InrModelDesc* model, const InrFlowDesc* flow, float* icnn_params, float* flow_params,
                          float* icnn_opt_state, float* flow_opt_state, const InrGridDesc* grid, const float* seg,
                          const float* target, const InrJointLossDesc* desc, const InrOptDesc* opt, float wd_on_weight_g,
                          int step, float* loss_out, float* dseg, float* prior_logits, int32_t* status, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* Measurement hook (bench.py, rocprof): launch ONLY the fused forward+loss+backward step kernel `iters` times
 * back-to-back on `stream` (no optimizer step), so its average duration can be bracketed with events. */
int inrfit_step_only(const InrModelDesc* model, const float* params, const InrGridDesc* grid, const float* targets,
                     const InrLossDesc* loss, int n_images, int iters, void* workspace, int64_t workspace_bytes,
                     void* stream);

/* Measurement hook (bench.py): one launch of `workgroups` x 4 waves that issue nothing but independent fp32 16x16x4 MFMAs
 * (32 * iters per wave); *flop = the products' flops.  Timed by the caller, it gives the matrix-pipe rate the chip sustains at
 * the clock it holds under full MFMA load - the practical ceiling next to the nominal peak.  `scratch`: any device buffer of
 * >= workgroups * 1 KiB (never written in practice). */
int inrfit_mfma_stream(int workgroups, int iters, double* flop, void* scratch, void* stream);

/* Measurement hook (bench.py): between begin and end, inrfit_fit and inrfit_step_only calls of this thread bracket each of
 * their (first max_samples) step-kernel launches - inrfit_fit also its update-kernel launches - with a pair of HIP events on
 * the launch stream; end waits for them and returns the average elapsed time of a bracket in microseconds (0 if none).
 * A bracket = the kernel + what the two event packets add to the sequence; bench.py removes that excess by comparing the sum
 * of the two brackets with the un-instrumented time per optimizer step. */
int inrfit_timing_begin(int max_samples);
int inrfit_timing_end(float* avg_step_bracket_us, float* avg_update_bracket_us, int* n_samples);

/* Test hook (tests/test_gpu_rnvp.py): the kernels' own tanh / exp of the coupling outputs (awesome_amd/csrc/rnvp.h fast_tanh /
 * fast_exp) evaluated on x [n] -> tanh_out [n], exp_out [n], so their error bounds are asserted against float64 on the device
 * that runs them. */
int inrfit_debug_tanh_exp(const float* x, int64_t n, float* tanh_out, float* exp_out, void* stream);

/* ---- star-shape prior of the teaser ----------------------------------------------------------------------
 * Replaces: `myNet` of notebooks/icml_teaser_code/star_shaped/star.ipynb cell 2 (forward) and the training loop of cell 3 (minibatch of
 * pixels -> sigmoid -> nn.MSELoss -> torch.optim.Adam(net.parameters(), lr) -> W2_r.weight <- relu(W2_r.weight); `offset` joins the
 * optimizer at a given epoch).  out = r (W2 x_old + W2_r relu(W1 x_old + W1_r r)) - 1, x_old = relu(W0 x / (0.01 + r)), r = |x + offset|.
 * params: ONE flat vector in the order of the class's named_parameters():
 *   offset[2] | W0.weight[h][2] | W0.bias[h] | W1.weight[h][h] | W1.bias[h] | W2.weight[h] | W2.bias | W1_r.weight[h] | W1_r.bias[h] |
 *   W2_r.weight[h] | W2_r.bias                                                      (inrfit_star_param_count = h^2 + 8 h + 4)
 * coords: [n_pixels][2] row-major (the notebook's pixel_info), labels [n_pixels]; 1 <= n_hidden <= 1024.  Layer-by-layer kernels with
 * the hand-written fp32-MFMA GEMMs of csrc/gemm.h for the h x h contractions (csrc/star.h); results reproducible run to run. */
typedef struct InrStarDesc {
    int32_t n_hidden;
} InrStarDesc;

int64_t inrfit_star_param_count(const InrStarDesc* star);
/* workspace for n_points points evaluated at once (the minibatch size for _fit / _loss_grad, all pixels for _forward) */
int64_t inrfit_star_workspace_bytes(const InrStarDesc* star, int64_t n_points);
/* logits[n_points] = net(coords) (star.ipynb cell 4 / 6: inference on every pixel) */
int inrfit_star_forward(const InrStarDesc* star, const float* params, const float* coords, int64_t n_points, float* logits,
                        void* workspace, int64_t workspace_bytes, void* stream);
/* One minibatch (index[batch] pixel numbers, or NULL = the first `batch` pixels): loss[1] = MSE(sigmoid(net), labels) and its gradient
 * with respect to every parameter, the centre included, in the parameter layout (grads[P]); batch <= 65536. */
int inrfit_star_loss_grad(const InrStarDesc* star, const float* params, const float* coords, const float* labels, int64_t n_pixels,
                          const int32_t* index, int64_t batch, float* loss, float* grads, void* workspace, int64_t workspace_bytes,
                          void* stream);
/* The training loop of cell 3 on the device: for epoch = step0 .. step0 + steps - 1: minibatch batch_index[epoch - step0][batch]
 * (values clamped into [0, n_pixels)), forward, MSE, backward, Adam (opt->lr, beta1, beta2, eps; kind must be INR_OPT_ADAM) on params
 * IN PLACE with opt_state [2 P] (exp_avg | exp_avg_sq; zero it before epoch 0), then W2_r.weight <- max(W2_r.weight, 0).  `offset`
 * takes its first step at epoch `offset_first_step` with its own step count, as torch's Adam treats a parameter that starts to
 * receive gradients (the notebook sets requires_grad after the forward pass of epoch 1000: offset_first_step = 1001; < 0: never).
 * loss_hist [steps] or NULL.  Nothing synchronises with the host. */
int inrfit_star_fit(const InrStarDesc* star, float* params, float* opt_state, const float* coords, const float* labels,
                    int64_t n_pixels, const int32_t* batch_index, int64_t batch, const InrOptDesc* opt, int32_t steps, int32_t step0,
                    int32_t offset_first_step, float* loss_hist, void* workspace, int64_t workspace_bytes, void* stream);

const char* inrfit_strerror(int code);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* INRFIT_H */
