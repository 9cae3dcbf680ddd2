"""CPU oracle for the per-image INR fit hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain CPU PyTorch (fp32), the algorithm of the reference
hot path (jp-schneider/awesome @ 2024_08_07) so that the HIP path can be checked
against it.  It is pinned against golden vectors produced by the real reference
classes (tests/golden/*.npz, generator: tools/gen_golden.py) by
tests/test_oracle_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module - and only as the checker / the timed CPU baseline.  The product
(awesome_amd/) never imports it and has no CPU fallback.

Every function cites the reference lines it follows (paths relative to the
reference checkout).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# --------------------------------------------------------------------------------------
# a1  coordinate grids
# --------------------------------------------------------------------------------------


def positional_grid(w: int, h: int, t: Optional[float] = None, t_max: Optional[float] = None) -> Tensor:
    """awesome/dataset/transformator.py:25-61 - (2|3, h, w) grid, channels (x, y[, t/t_max]), linspace(0,1)."""
    y = torch.linspace(0, 1, h)
    x = torch.linspace(0, 1, w)
    yy, xx = torch.meshgrid(y, x, indexing="ij")
    if t is None:
        return torch.stack((xx, yy), dim=0).float()
    if t_max is None:
        raise ValueError("t_max must be set if t is set")
    return torch.stack((xx, yy, torch.ones_like(xx) * t / t_max), dim=0).float()


def howto_grid(h: int, w: int) -> Tensor:
    """notebooks/how_to/convexity.ipynb cell 7 `create_grid` - (1,2,h,w), x = i/w, y = j/h."""
    x = torch.arange(0, w)
    y = torch.arange(0, h)
    xx, yy = torch.meshgrid(x, y, indexing="xy")
    grid = torch.stack((xx, yy), dim=0)
    return grid.unsqueeze(0).float() / torch.tensor([w, h]).float().unsqueeze(-1).unsqueeze(-1)


def pixelize(x: Tensor) -> Tensor:
    """awesome/util/pixelize.py:31-33 - (B,C,H,W) -> (B*H*W, C)."""
    v = x.permute(0, 2, 3, 1)
    return v.reshape(-1, v.shape[-1])


def unpixelize(x: Tensor, b: int, h: int, w: int) -> Tensor:
    """awesome/util/pixelize.py:35-37."""
    return x.reshape(b, h, w, -1).permute(0, 3, 1, 2)


# --------------------------------------------------------------------------------------
# a3/a4/a5  ICNN family.  Parameters are kept in a dict with ConvexNextNet key names
# (awesome/model/convex_net.py:177-220): input.{weight,bias}, skip.k.ln.{weight,bias},
# skip.k.skp.weight, out.ln.{weight,bias}, out.skp.weight.
# --------------------------------------------------------------------------------------

CONVEXNET_KEYMAP = {  # ConvexNet (convex_net.py:10-40) == ConvexNextNet(L=1) under this renaming
    "W0y.weight": "input.weight", "W0y.bias": "input.bias",
    "W1z.weight": "skip.0.ln.weight", "W1z.bias": "skip.0.ln.bias", "W1y.weight": "skip.0.skp.weight",
    "W2z.weight": "out.ln.weight", "W2z.bias": "out.ln.bias", "W2y.weight": "out.skp.weight",
}


def icnn_keys(n_hidden_layers: int) -> List[str]:
    keys = ["input.weight", "input.bias"]
    for k in range(n_hidden_layers):
        keys += [f"skip.{k}.ln.weight", f"skip.{k}.ln.bias", f"skip.{k}.skp.weight"]
    keys += ["out.ln.weight", "out.ln.bias", "out.skp.weight"]
    return keys


def icnn_num_layers(p: Dict[str, Tensor]) -> int:
    return sum(1 for k in p if k.startswith("skip.") and k.endswith(".ln.weight"))


def encode_layer(pre: Tensor, act0: str = "relu", omega: float = 1.0) -> Tensor:
    """Layer 0's activation = the encode stage: relu (the packaged models), cos (random Fourier features cos(x @ A + b),
    notebooks/imageRepresentationTest.ipynb cell 5) or sin(omega .) (the sine layer sin(10 pi W1(x + offset)),
    notebooks/icml_teaser_code/repeating/repeating.ipynb cell 3).  Pinned by tests/golden/encode_notebooks.npz (the two cells'
    own classes executed in the build container)."""
    if act0 == "relu":
        return F.relu(pre)
    if act0 == "cos":
        return torch.cos(pre)
    if act0 == "sin":
        return torch.sin(omega * pre)
    raise ValueError(act0)


def fourier_mlp_forward(sd: Dict[str, Tensor], x: Tensor) -> Tensor:
    """`ourSimpleNetwork.forward` (imageRepresentationTest.ipynb cell 5) restated: cos(x @ A + b) -> relu(fc1) -> relu(fc2) ->
    relu(fc3) -> sigmoid(fc4), any number of fcK layers (the last one is the output layer)."""
    z = torch.cos(x @ sd["A"] + sd["b"])
    n = sum(1 for k in sd if k.startswith("fc") and k.endswith(".weight"))
    for k in range(1, n):
        z = F.relu(F.linear(z, sd[f"fc{k}.weight"], sd[f"fc{k}.bias"]))
    return torch.sigmoid(F.linear(z, sd[f"fc{n}.weight"], sd[f"fc{n}.bias"]))


def sine_net_forward(sd: Dict[str, Tensor], x: Tensor) -> Tensor:
    """`myNet.forward` (repeating.ipynb cell 3) restated: W2(sin(10 * 3.141592 * W1(x + offset)))."""
    z = 10 * 3.141592 * F.linear(x + sd["offset"], sd["W1.weight"], sd["W1.bias"])
    return F.linear(torch.sin(z), sd["W2.weight"], sd["W2.bias"])


def rotation_symmetric_features(x: Tensor, offset: Tensor, orientation: Tensor, symmetry_prior: bool) -> Tensor:
    """First half of `myNet.forward` (icml_teaser_code/rotation_symmetric/rotation_symmetric.ipynb cell 2): centre, polar split,
    rotate by `orientation`, mirror the second direction component (the symmetry prior) -> (N, 3) = [direction, radius]."""
    x = x + offset
    r = (x * x).sum(1, keepdim=True).sqrt()
    x = x / (0.001 + r)
    c, s = torch.cos(orientation), torch.sin(orientation)
    u, v = x[:, 0] * c - x[:, 1] * s, x[:, 0] * s + x[:, 1] * c
    if symmetry_prior:
        v = v.abs()
    return torch.stack((u, v, r[:, 0]), 1)


def rotation_symmetric_forward(sd: Dict[str, Tensor], x: Tensor, symmetry_prior: bool) -> Tensor:
    """`myNet.forward` of the same cell: W2(relu(W1(relu(W0([direction, radius])))))."""
    z = rotation_symmetric_features(x, sd["offset"], sd["orientation"], symmetry_prior)
    z = F.relu(F.linear(z, sd["W0.weight"], sd["W0.bias"]))
    z = F.relu(F.linear(z, sd["W1.weight"], sd["W1.bias"]))
    return F.linear(z, sd["W2.weight"], sd["W2.bias"])


def star_shaped_forward(sd: Dict[str, Tensor], x: Tensor) -> Tensor:
    """`myNet.forward` (icml_teaser_code/star_shaped/star.ipynb cell 2): r * (W2 x_old + W2_r relu(W1 x_old + W1_r r)) - 1 with
    x_old = relu(W0 (x / (0.01 + r))), r = |x + offset|."""
    x = x + sd["offset"]
    r = (x * x).sum(1, keepdim=True).sqrt()
    x_old = F.relu(F.linear(x / (0.01 + r), sd["W0.weight"], sd["W0.bias"]))
    r_aug = F.relu(F.linear(x_old, sd["W1.weight"], sd["W1.bias"]) + F.linear(r, sd["W1_r.weight"], sd["W1_r.bias"]))
    return r * (F.linear(x_old, sd["W2.weight"], sd["W2.bias"]) + F.linear(r_aug, sd["W2_r.weight"], sd["W2_r.bias"])) - 1


def star_shaped_fit(sd: Dict[str, Tensor], pixel_info: Tensor, labels: Tensor, batch_index: Tensor, lr: float = 1e-2,
                    offset_free_epoch: Optional[int] = 1000, offset_trainable_from_start: bool = False):
    """The training loop of star.ipynb cell 3 on given minibatches (batch_index [epochs, batch] replaces its two torch.randperm draws):
    outputs = sigmoid(net(pixels)); loss = nn.MSELoss(); `if epoch == 1000: net.offset.requires_grad = True` AFTER the forward pass
    (so the centre's first Adam step is the next epoch's); optimizer.step(); W2_r.weight <- relu(W2_r.weight).
    Returns (state dict after the last epoch, losses)."""
    p = {k: v.detach().clone().requires_grad_(k != "offset" or offset_trainable_from_start) for k, v in sd.items()}
    opt = torch.optim.Adam(list(p.values()), lr=lr)
    losses = []
    for epoch in range(batch_index.shape[0]):
        idx = batch_index[epoch].long()
        outputs = torch.sigmoid(star_shaped_forward(p, pixel_info[idx])).squeeze()
        loss = F.mse_loss(outputs, labels[idx])
        if offset_free_epoch is not None and epoch == offset_free_epoch:
            p["offset"].requires_grad = True
        opt.zero_grad()
        loss.backward()
        opt.step()
        with torch.no_grad():
            p["W2_r.weight"].data = F.relu(p["W2_r.weight"].data)
        losses.append(float(loss.detach()))
    return {k: v.detach().clone() for k, v in p.items()}, losses


def icnn_forward(p: Dict[str, Tensor], x: Tensor, act0: str = "relu", omega: float = 1.0) -> Tensor:
    """ConvexNextNet.forward (convex_net.py:205-214) on (N,C) rows -> (N,1) logits; `act0`: layer 0's activation (encode_layer)."""
    x_in = x
    z = encode_layer(F.linear(x_in, p["input.weight"], p["input.bias"]), act0, omega)
    for k in range(icnn_num_layers(p)):
        z = F.relu(F.linear(z, p[f"skip.{k}.ln.weight"], p[f"skip.{k}.ln.bias"]) + F.linear(x_in, p[f"skip.{k}.skp.weight"]))
    return F.linear(z, p["out.ln.weight"], p["out.ln.bias"]) + F.linear(x_in, p["out.skp.weight"])


def icnn_forward_image(p: Dict[str, Tensor], grid: Tensor, act0: str = "relu", omega: float = 1.0) -> Tensor:
    """@pixelize wrapper: (B,C,H,W) -> (B,1,H,W)."""
    b, c, h, w = grid.shape
    return unpixelize(icnn_forward(p, pixelize(grid), act0, omega), b, h, w)


def icnn_clamp_keys(p: Dict[str, Tensor]) -> List[str]:
    """Keys projected onto >= 0 by enforce_convexity (convex_net.py:151-154, 216-220; ConvexNet :37-40):
    hidden->hidden and hidden->out `ln.weight`; the skip weights stay free."""
    return [f"skip.{k}.ln.weight" for k in range(icnn_num_layers(p))] + ["out.ln.weight"]


def icnn_enforce_convexity(p: Dict[str, Tensor]) -> None:
    with torch.no_grad():
        for k in icnn_clamp_keys(p):
            p[k].copy_(F.relu(p[k]))


# --------------------------------------------------------------------------------------
# a11  data terms
# --------------------------------------------------------------------------------------


def unaries_weight(target: Tensor, mode: str, ratio: float = 1.0) -> Tensor:
    """UnariesWeightedLoss._compute_weight (awesome/measures/unaries_weighted_loss.py:35-69).
    Weight applied to elements with target < 0.5 (foreground); others get 1."""
    if mode == "none":
        return torch.ones_like(target)
    fg = (target < 0.5).sum()
    bg = (target >= 0.5).sum()
    cc = bg.float() / fg.float()
    if mode == "ratio":
        wv = (cc - 1) * ratio + 1
    elif mode == "sssdms":
        wv = torch.round(cc / 10) + 1
    elif mode == "equal":
        wv = cc
    else:
        raise ValueError(f"Mode {mode} is not supported")
    w = torch.ones_like(target)
    w[target < 0.5] = wv
    return w


def class_weight(target: Tensor, mode: str) -> Tensor:
    """WeightedLoss._compute_weight on CLASS labels (awesome/measures/weighted_loss.py:38-62): fg = target == 0, bg = target == 1,
    the fg pixels weighted by bg / fg ('equal') or round((bg / fg) / 10) + 1 ('sssdms'); everything else weight 1."""
    if mode == "none":
        return torch.ones_like(target)
    cc = (target == 1).sum().float() / (target == 0).sum().float()
    if mode == "sssdms":
        wv = torch.round(cc / 10) + 1
    elif mode == "equal":
        wv = cc
    else:
        raise ValueError(f"Mode {mode} is not supported")
    w = torch.ones_like(target)
    w[target == 0] = wv
    return w


def weighted_loss(output: Tensor, target: Tensor, kind: str = "se", mode: str = "none", ratio: float = 1.0,
                  noneclass: Optional[float] = None, class_targets: bool = False) -> Tensor:
    """WeightedLoss.__call__ (awesome/measures/weighted_loss.py:67-92) with criterion SE
    (awesome/measures/se.py:21-23) or nn.BCELoss, reduction 'mean' over all elements.  `noneclass`: the pixels whose target equals
    it are removed first (:71-74) - from the criterion, the class counts and the mean.  `class_targets`: the base class's weights on
    labels {0, 1} (class_weight) instead of UnariesWeightedLoss's on unaries (unaries_weight).
    Pinned by tests/golden/weighted_loss_noneclass.npz (the reference class, three modes x two criteria, with and without noneclass)."""
    if noneclass is not None:
        keep = target != noneclass
        output, target = output[keep], target[keep]
    if kind == "se":
        l = (target - output) ** 2
    elif kind == "bce":
        l = F.binary_cross_entropy(output, target, reduction="none")
    else:
        raise ValueError(kind)
    if mode != "none":
        l = l * (class_weight(target, mode) if class_targets else unaries_weight(target, mode, ratio))
    return l.mean()


def awesome_image_loss(output: Tensor, target: Tensor, alpha=1.0, beta=100.0, gamma=0.1, extra_penalty=False) -> Tensor:
    """AwesomeImageLoss.__call__ with default BCE criteria (awesome/measures/awesome_image_loss.py:34-53)."""
    c = output.shape[1] // 2
    seg, prior = output[:, :c], output[:, c:]
    loss = F.binary_cross_entropy(seg, target) + alpha * F.binary_cross_entropy(prior, target)
    if extra_penalty:
        loss = gamma * loss + beta * torch.mean((prior - (seg > 0.5).float()) ** 2)
    return loss


def fbms_joint_loss(output: Tensor, target: Tensor, alpha: float = 1.0, beta: float = 1.0, clip_penalty: bool = True,
                    kind: str = "bce", mode: str = "sssdms", ratio: float = 1.0, noneclass: Optional[float] = None,
                    class_targets: bool = False) -> Tensor:
    """FBMSJointLoss.__call__ (awesome/measures/fbms_joint_loss.py:35-59): output (B, 2, H, W) = [seg, prior];
    alpha * crit(seg, target) + beta * SE_mean(prior, seg), the penalty rescaled (detached factor) to the segmentation loss
    when it exceeds it.  Pinned by tests/golden/fbms_joint_loss.npz (both branches)."""
    c = output.shape[1] // 2
    seg, prior = output[:, :c], output[:, c:]
    seg_loss = alpha * weighted_loss(seg, target, kind=kind, mode=mode, ratio=ratio, noneclass=noneclass, class_targets=class_targets)
    pen = beta * torch.mean((seg - prior) ** 2)      # the penalty has no targets: every pixel, noneclass or not
    if clip_penalty and bool(pen > seg_loss):     # the reference branches on the host exactly like this
        pen = pen * (seg_loss / pen).detach()
    return seg_loss + pen


def wrapper_forward(seg_logits: Tensor, prior_logits: Tensor, invert_seg: bool = False) -> Tensor:
    """WrapperModule.forward in image mode (awesome/model/wrapper_module.py:157-260): per batch element
    cat([sigmoid(seg) or 1 - sigmoid(seg), sigmoid(prior)], channel).  seg_logits, prior_logits: (B, 1, H, W).
    Pinned by tests/golden/wrapper_module.npz."""
    seg = torch.sigmoid(seg_logits)
    if invert_seg:
        seg = 1 - seg
    return torch.cat([seg, torch.sigmoid(prior_logits)], dim=1)


# --------------------------------------------------------------------------------------
# a17  metric
# --------------------------------------------------------------------------------------


def miou_binary(output: Tensor, target: Tensor, invert: bool = True) -> float:
    """MIOU.__call__ (awesome/measures/miou.py:29-48), average='binary': Jaccard of the positive class
    after the optional 1-x inversion; 0 when the (inverted) target has no positives."""
    o = output.detach().reshape(-1).float()
    t = target.detach().reshape(-1).float()
    if invert:
        o, t = 1.0 - o, 1.0 - t
    if bool(torch.all(t == 0.0)):
        return 0.0
    ob, tb = o == 1.0, t == 1.0
    inter = int((ob & tb).sum())
    union = int((ob | tb).sum())
    return float(inter) / float(union) if union > 0 else 0.0


# --------------------------------------------------------------------------------------
# a14  optimizers (single-tensor restatement of torch.optim.Adam / Adamax defaults, as used at
# awesome/run/awesome_config.py:34-41, path_connected_net.py:924-933, convexity how-to cell 7)
# --------------------------------------------------------------------------------------


class AdamState:
    def __init__(self, params: Dict[str, Tensor]):
        self.step = 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}


def adam_step(p: Dict[str, Tensor], g: Dict[str, Tensor], st: AdamState, lr: float, betas=(0.9, 0.999), eps=1e-8,
              weight_decay: float = 0.0) -> None:
    b1, b2 = betas
    st.step += 1
    bc1 = 1 - b1 ** st.step
    bc2 = 1 - b2 ** st.step
    step_size = lr / bc1
    bc2_sqrt = math.sqrt(bc2)
    with torch.no_grad():
        for k in p:
            grad = g[k]
            if weight_decay != 0:
                grad = grad + weight_decay * p[k]
            st.m[k].lerp_(grad, 1 - b1)
            st.v[k].mul_(b2).addcmul_(grad, grad, value=1 - b2)
            denom = (st.v[k].sqrt() / bc2_sqrt).add_(eps)
            p[k].addcdiv_(st.m[k], denom, value=-step_size)


def adamax_step(p: Dict[str, Tensor], g: Dict[str, Tensor], st: AdamState, lr: float, betas=(0.9, 0.999), eps=1e-8,
                weight_decay: float = 0.0) -> None:
    """torch.optim.Adamax: exp_inf = max(b2*exp_inf, |g|+eps); p -= lr/(1-b1^t) * m/exp_inf (st.v holds exp_inf)."""
    b1, b2 = betas
    st.step += 1
    clr = lr / (1 - b1 ** st.step)
    with torch.no_grad():
        for k in p:
            grad = g[k]
            if weight_decay != 0:
                grad = grad + weight_decay * p[k]
            st.m[k].lerp_(grad, 1 - b1)
            norm_buf = torch.cat([st.v[k].mul_(b2).unsqueeze(0), grad.abs().add_(eps).unsqueeze(0)], 0)
            st.v[k].copy_(torch.amax(norm_buf, 0, keepdim=False))
            p[k].addcdiv_(st.m[k], st.v[k], value=-clr)


class PlateauState:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', threshold=1e-4 rel, cooldown=0, min_lr=0, eps=1e-8)
    as used at path_connected_net.py:932-933, 951."""

    def __init__(self, lr: float, patience: int = 200, factor: float = 0.5, threshold: float = 1e-4, min_lr: float = 0.0,
                 eps: float = 1e-8):
        self.lr, self.patience, self.factor, self.threshold, self.min_lr, self.eps = lr, patience, factor, threshold, min_lr, eps
        self.best = math.inf
        self.num_bad = 0

    def step(self, metric: float) -> float:
        if metric < self.best * (1.0 - self.threshold):
            self.best = metric
            self.num_bad = 0
        else:
            self.num_bad += 1
        if self.num_bad > self.patience:
            new_lr = max(self.lr * self.factor, self.min_lr)
            if self.lr - new_lr > self.eps:
                self.lr = new_lr
            self.num_bad = 0
        return self.lr


# --------------------------------------------------------------------------------------
# a13  the hot loop
# --------------------------------------------------------------------------------------


def fit_icnn(p0: Dict[str, Tensor], grid: Tensor, unaries: Tensor, steps: int, lr: float = 2e-3, loss_kind: str = "se",
             weight_mode: str = "none", ratio: float = 1.0, optimizer: str = "adam", betas=(0.9, 0.999), eps: float = 1e-8,
             weight_decay: float = 0.0, plateau: Optional[dict] = None, record_every: int = 0
             ) -> Tuple[Dict[str, Tensor], List[float], Tensor]:
    """E x {zero_grad, forward, loss, backward, Adam/Adamax step, clamp[, plateau step]} on one image.
    Follows _prior_based_pretrain's inner loop (awesome/model/path_connected_net.py:937-962) and the how-to loop
    (notebooks/how_to/convexity.ipynb cell 9, with plain mean instead of fg/bg re-weighting).
    grid: (1,C,H,W); unaries: (1,1,H,W).  Returns (params, per-step losses, final logits (1,1,H,W))."""
    p = {k: v.detach().clone().requires_grad_(True) for k, v in p0.items()}
    st = AdamState(p)
    sched = PlateauState(lr, **plateau) if plateau is not None else None
    losses: List[float] = []
    cur_lr = lr
    for _ in range(steps):
        for v in p.values():
            v.grad = None
        out = torch.sigmoid(icnn_forward_image(p, grid))
        loss = weighted_loss(out, unaries, loss_kind, weight_mode, ratio)
        loss.backward()
        g = {k: v.grad for k, v in p.items()}
        if optimizer == "adam":
            adam_step(p, g, st, cur_lr, betas, eps, weight_decay)
        elif optimizer == "adamax":
            adamax_step(p, g, st, cur_lr, betas, eps, weight_decay)
        else:
            raise ValueError(optimizer)
        icnn_enforce_convexity(p)
        lv = float(loss.item())
        losses.append(lv)
        if sched is not None:
            cur_lr = sched.step(lv)
    with torch.no_grad():
        logits = icnn_forward_image(p, grid)
    return {k: v.detach() for k, v in p.items()}, losses, logits


def loss_and_grads(p0: Dict[str, Tensor], grid: Tensor, unaries: Tensor, loss_kind="se", weight_mode="none", ratio=1.0):
    p = {k: v.detach().clone().requires_grad_(True) for k, v in p0.items()}
    out = torch.sigmoid(icnn_forward_image(p, grid))
    loss = weighted_loss(out, unaries, loss_kind, weight_mode, ratio)
    loss.backward()
    return float(loss.item()), {k: v.grad.detach() for k, v in p.items()}


# --------------------------------------------------------------------------------------
# a6-a9  weight-normalised coupling flow (path-connected prior, in-tree variant)
# --------------------------------------------------------------------------------------


def wn_weight(v: Tensor, g: Tensor) -> Tensor:
    """nn.utils.weight_norm(dim=None): w = g * v / ||v||_F with scalar g (real_nvp/resnet_1d.py:48)."""
    return v * (g / torch.linalg.norm(v))


def wn_linear(sd: Dict[str, Tensor], prefix: str, x: Tensor) -> Tensor:
    """WNLinear.forward (awesome/model/real_nvp/resnet_1d.py:39-63)."""
    w = wn_weight(sd[prefix + "linear.weight_v"], sd[prefix + "linear.weight_g"])
    return F.linear(x, w, sd.get(prefix + "linear.bias"))


def normal_block(sd: Dict[str, Tensor], prefix: str, x: Tensor) -> Tensor:
    """The s / t net of a coupling, by the keys the state_dict holds:
    NormalBlock.forward (awesome/model/diffeomorphism_net.py:169-192): tanh(WN2(leaky_relu(WN1 x))), keys in_linear / out_linear;
    SimpleBackbone.forward (:83-104, NormalizingFlow1D's 'default' backbone): tanh(WN2(relu(WN1 x))), keys linear1 / linear2."""
    if prefix + "linear1.linear.weight_v" in sd:
        return torch.tanh(wn_linear(sd, prefix + "linear2.", F.relu(wn_linear(sd, prefix + "linear1.", x))))
    h = F.leaky_relu(wn_linear(sd, prefix + "in_linear.", x))
    return torch.tanh(wn_linear(sd, prefix + "out_linear.", h))


def wn_scale(sd: Dict[str, Tensor], prefix: str) -> Tensor:
    """WNScale.forward (diffeomorphism_net.py:208-232): weight-normed 1x1 linear (default dim=0) of a learnable scalar."""
    v, g = sd[prefix + "scale.weight_v"], sd[prefix + "scale.weight_g"]
    w = v * (g / torch.linalg.norm(v, dim=1, keepdim=True))
    return F.linear(sd[prefix + "weight"], w, sd[prefix + "scale.bias"])


def flow1d_forward(sd: Dict[str, Tensor], x: Tensor, num_coupling: int, prefix: str = "") -> Tensor:
    """NormalizingFlow1D.forward (diffeomorphism_net.py:286-300), NormalBlock or SimpleBackbone nets."""
    x1, x2 = x[:, :1], x[:, 1:]
    for i in range(num_coupling):
        if i % 2 == 0:
            s = wn_scale(sd, f"{prefix}scale.{i}.") * normal_block(sd, f"{prefix}s.{i}.", x1)
            x2 = torch.exp(s) * x2 + normal_block(sd, f"{prefix}t.{i}.", x1)
        else:
            s = wn_scale(sd, f"{prefix}scale.{i}.") * normal_block(sd, f"{prefix}s.{i}.", x2)
            x1 = torch.exp(s) * x1 + normal_block(sd, f"{prefix}t.{i}.", x2)
    return torch.cat([x1, x2], 1)


def convex_diffeo_forward(sd: Dict[str, Tensor], x: Tensor, num_coupling: int) -> Tensor:
    """ConvexDiffeomorphismNet.forward (awesome/model/convex_diffeomorphism_net.py:173-178): ICNN(flow(Ax+b))."""
    x = F.linear(x, sd["linear.weight"], sd["linear.bias"])
    xd = flow1d_forward(sd, x, num_coupling, prefix="diffeo_net.")
    p = {k[len("convex_net."):]: v for k, v in sd.items() if k.startswith("convex_net.")}
    return icnn_forward(p, xd)


# --------------------------------------------------------------------------------------
# helpers for tests
# --------------------------------------------------------------------------------------


def load_npz_state(npz, prefix: str) -> Dict[str, Tensor]:
    return {k[len(prefix):]: torch.from_numpy(np.asarray(npz[k])).clone() for k in npz.files if k.startswith(prefix)}


def to_convexnext_keys(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    if "W0y.weight" in sd:
        return {CONVEXNET_KEYMAP[k]: v for k, v in sd.items()}
    return dict(sd)


# --------------------------------------------------------------------------------------
# a13 (path-connected prior)  the inner loop of ConvexDiffeomorphismNet.pretrain
# --------------------------------------------------------------------------------------


def fit_convex_diffeo(sd0: Dict[str, Tensor], grid: Tensor, unaries: Tensor, steps: int, num_coupling: int, lr: float = 3e-3,
                      loss_kind: str = "bce", weight_decay_on_weight_g: float = 5e-5, plateau: Optional[dict] = None
                      ) -> Tuple[Dict[str, Tensor], List[float], Tensor]:
    """awesome/model/convex_diffeomorphism_net.py:405-430: Adam over two parameter groups (every `*weight_g` with weight
    decay, the rest without - awesome/util/torch.py:19-35), criterion on sigmoid(model(grid)), ReduceLROnPlateau stepped
    with the loss, enforce_convexity (ICNN part only) after every step.  grid (1,2,H,W), unaries (1,1,H,W)."""
    p = {k: v.detach().clone().requires_grad_(True) for k, v in sd0.items()}
    st = AdamState(p)
    sched = PlateauState(lr, **plateau) if plateau is not None else None
    rows = pixelize(grid)
    _, _, h, w = grid.shape
    losses: List[float] = []
    cur_lr = lr
    norm_keys = [k for k in p if k.endswith("weight_g")]
    other_keys = [k for k in p if not k.endswith("weight_g")]
    for _ in range(steps):
        for v in p.values():
            v.grad = None
        out = torch.sigmoid(unpixelize(convex_diffeo_forward(p, rows, num_coupling), 1, h, w))
        loss = weighted_loss(out, unaries, loss_kind)
        loss.backward()
        st.step += 1
        for keys, wd in ((norm_keys, weight_decay_on_weight_g), (other_keys, 0.0)):
            sub, g = {k: p[k] for k in keys}, {k: p[k].grad for k in keys}
            sst = AdamState.__new__(AdamState)
            sst.step, sst.m, sst.v = st.step - 1, st.m, st.v
            adam_step(sub, g, sst, cur_lr, weight_decay=wd)
        with torch.no_grad():
            for k in p:
                if k.startswith("convex_net.") and (k.endswith("ln.weight")) and not k.startswith("convex_net.input"):
                    p[k].copy_(F.relu(p[k]))
        lv = float(loss.item())
        losses.append(lv)
        if sched is not None:
            cur_lr = sched.step(lv)
    with torch.no_grad():
        logits = unpixelize(convex_diffeo_forward(p, rows, num_coupling), 1, h, w)
    return {k: v.detach() for k, v in p.items()}, losses, logits


# --------------------------------------------------------------------------------------
# a10  PathConnectedNet with the normflows RealNVP deformation.  PARITY UNPINNED:
# the flow's arithmetic is normflows==1.7.3 (poetry.lock pin), which is not part of the
# reference checkout and not installed here.  The functions below restate its published
# definitions (nf.nets.MLP, nf.flows.MaskedAffineFlow, nf.flows.ActNorm / AffineConstFlow,
# nf.NormalizingFlow.forward) and are anchored on the reference's call sites
# (awesome/model/net_factory.py:70-114,124-175; model/norm_net.py:17-27;
# transforms/min_max.py:8-58; model/path_connected_net.py:65-85,922-962).
# --------------------------------------------------------------------------------------


def rnvp_masks(channels: int, n_flows: int) -> Tensor:
    """init_realnvp's coupling masks (net_factory.py:82-99): 1 .. 2^C-2 in binary (LSB = channel 0), repeated / cropped."""
    vals = torch.arange(1, 2 ** channels - 1)
    bits = 2 ** torch.arange(channels)
    base = (vals.unsqueeze(-1).bitwise_and(bits) != 0).to(torch.uint8)
    rep, crop = divmod(n_flows, base.shape[0])
    masks = torch.zeros((n_flows, channels), dtype=torch.uint8)
    if rep > 0:
        masks[:rep * base.shape[0]] = base.repeat((rep, 1))
    masks[rep * base.shape[0]:] = base[:crop]
    return masks


def minmax(v: Tensor, v_min, v_max, new_min, new_max) -> Tensor:
    """awesome/transforms/min_max.py:8-19."""
    return (v - v_min) / (v_max - v_min) * (new_max - new_min) + new_min


def rnvp_mlp(sd: Dict[str, Tensor], prefix: str, x: Tensor, output_fn: Optional[str], output_scale: Optional[float]) -> Tensor:
    """nf.nets.MLP([C, hid, C], leaky=0.0, init_zeros=True, output_fn, output_scale): Linear, LeakyReLU(0.0), Linear[, Tanh[, x scale]]."""
    h = F.leaky_relu(F.linear(x, sd[prefix + "net.0.weight"], sd[prefix + "net.0.bias"]), 0.0)
    o = F.linear(h, sd[prefix + "net.2.weight"], sd[prefix + "net.2.bias"])
    if output_fn == "tanh":
        o = torch.tanh(o)
        if output_scale is not None:
            o = o * output_scale
    return o


def rnvp_flow_forward(sd: Dict[str, Tensor], z: Tensor, masks: Tensor, output_fn: Optional[str] = "tanh",
                      output_scale: Optional[float] = None, prefix: str = "flow_net.net.network.",
                      actnorm_init: bool = False) -> Tensor:
    """nf.NormalizingFlow.forward over [MaskedAffineFlow(b, t, s), ActNorm(C)] x F on rows z (N, C).
    MaskedAffineFlow.forward: z_masked = b z; z' = z_masked + (1 - b)(z exp(s(z_masked)) + t(z_masked)).
    ActNorm (AffineConstFlow): z'' = z' exp(s) + t; its first forward sets s = -log(std(z', 0) + 1e-6), t = -mean exp(s)
    (actnorm_init=True writes them into sd, like the data-dependent init does)."""
    for f in range(masks.shape[0]):
        b = masks[f].to(z.dtype).view(1, -1)
        zm = b * z
        s = rnvp_mlp(sd, f"{prefix}flows.{2 * f}.s.", zm, output_fn, output_scale)
        t = rnvp_mlp(sd, f"{prefix}flows.{2 * f}.t.", zm, output_fn, output_scale)
        z = zm + (1 - b) * (z * torch.exp(s) + t)
        ks, kt = f"{prefix}flows.{2 * f + 1}.s", f"{prefix}flows.{2 * f + 1}.t"
        if actnorm_init:
            with torch.no_grad():
                s_init = -torch.log(z.std(dim=0, keepdim=True) + 1e-6)
                sd[ks] = s_init.detach().clone()
                sd[kt] = (-z.mean(dim=0, keepdim=True) * torch.exp(s_init)).detach().clone()
        z = z * torch.exp(sd[ks]) + sd[kt]
    return z


def pcn_deformation(sd: Dict[str, Tensor], x: Tensor, masks: Tensor, vmin: Tensor, vmax: Tensor, new_min: float = -1.0,
                    new_max: float = 1.0, output_fn: Optional[str] = "tanh", output_scale: Optional[float] = None,
                    actnorm_init: bool = False) -> Tensor:
    """PathConnectedNet.get_deformation (path_connected_net.py:124-128) on rows x (N, C): the 1x1 depthwise conv
    (:65, per-channel a x + b), then NormNet.forward (norm_net.py:17-27): MinMax.transform, flow, MinMax.inverse_transform."""
    v = x * sd["linear.weight"].view(1, -1) + sd["linear.bias"].view(1, -1)
    z = minmax(v, vmin.view(1, -1), vmax.view(1, -1), new_min, new_max)
    z = rnvp_flow_forward(sd, z, masks, output_fn, output_scale, actnorm_init=actnorm_init)
    return minmax(z, new_min, new_max, vmin.view(1, -1), vmax.view(1, -1))


def pcn_forward(sd: Dict[str, Tensor], x: Tensor, masks: Tensor, vmin: Tensor, vmax: Tensor, **kw) -> Tensor:
    """PathConnectedNet.forward (path_connected_net.py:79-85) on rows: convex_net(flow_net(linear(x)))."""
    xd = pcn_deformation(sd, x, masks, vmin, vmax, **kw)
    return icnn_forward({k[len("convex_net."):]: v for k, v in sd.items() if k.startswith("convex_net.")}, xd)


def fit_pcn(sd0: Dict[str, Tensor], rows: Tensor, unaries_rows: Tensor, steps: int, masks: Tensor, vmin: Tensor, vmax: Tensor,
            lr: float = 1e-3, optimizer: str = "adamax", flow_weight_decay: float = 1e-5, loss_kind: str = "se",
            weight_mode: str = "none", plateau: Optional[dict] = None, **kw) -> Tuple[Dict[str, Tensor], List[float], Tensor]:
    """_prior_based_pretrain's inner loop for PathConnectedNet (path_connected_net.py:922-962): Adamax over the groups
    {flow_net: weight_decay=flow_weight_decay, convex_net, linear}, ReduceLROnPlateau(patience 200, factor .5) if `plateau`
    is given, enforce_convexity after every step.  rows (N, C), unaries_rows (N, 1)."""
    p = {k: v.detach().clone().requires_grad_(True) for k, v in sd0.items()}
    st = AdamState(p)
    sched = PlateauState(lr, **plateau) if plateau is not None else None
    cur_lr = lr
    losses: List[float] = []
    flow_keys = [k for k in p if k.startswith("flow_net.")]
    other_keys = [k for k in p if not k.startswith("flow_net.")]
    step_fn = adamax_step if optimizer == "adamax" else adam_step
    for _ in range(steps):
        for v in p.values():
            v.grad = None
        out = torch.sigmoid(pcn_forward(p, rows, masks, vmin, vmax, **kw))
        loss = weighted_loss(out.reshape(1, 1, -1, 1), unaries_rows.reshape(1, 1, -1, 1), loss_kind, weight_mode)
        loss.backward()
        st.step += 1
        for keys, wd in ((flow_keys, flow_weight_decay), (other_keys, 0.0)):
            sub, g = {k: p[k] for k in keys}, {k: p[k].grad for k in keys}
            sst = AdamState.__new__(AdamState)
            sst.step, sst.m, sst.v = st.step - 1, st.m, st.v
            step_fn(sub, g, sst, cur_lr, weight_decay=wd)
        with torch.no_grad():
            for k in p:
                if k.startswith("convex_net.") and k.endswith("ln.weight") and not k.startswith("convex_net.input"):
                    p[k].copy_(F.relu(p[k]))
        lv = float(loss.item())
        losses.append(lv)
        if sched is not None:
            cur_lr = sched.step(lv)
    with torch.no_grad():
        logits = pcn_forward(p, rows, masks, vmin, vmax, **kw)
    return {k: v.detach() for k, v in p.items()}, losses, logits


def fit_flow_identity(sd0: Dict[str, Tensor], rows: Tensor, steps: int, masks: Tensor, vmin: Tensor, vmax: Tensor, lr: float = 1e-2,
                      weight_decay: float = 1e-5, optimizer: str = "adamax", new_min: float = -1.0, new_max: float = 1.0, **kw
                      ) -> Tuple[Dict[str, Tensor], List[float]]:
    """PathConnectedNet.learn_flow_identity (path_connected_net.py:155-250): Adamax(flow_net.parameters(), lr, weight_decay) on
    SE('mean')(flow_net(x), x); flow_net = NormNet (MinMax, flow, MinMax^-1) WITHOUT the 1x1 linear.  rows (N, C)."""
    p = {k: v.detach().clone().requires_grad_(k.startswith("flow_net.")) for k, v in sd0.items()}
    keys = [k for k in p if k.startswith("flow_net.")]
    sub = {k: p[k] for k in keys}
    st = AdamState(sub)
    step_fn = adamax_step if optimizer == "adamax" else adam_step
    losses: List[float] = []
    for _ in range(steps):
        for k in keys:
            p[k].grad = None
        z = minmax(rows, vmin.view(1, -1), vmax.view(1, -1), new_min, new_max)
        z = rnvp_flow_forward(p, z, masks, **kw)
        y = minmax(z, new_min, new_max, vmin.view(1, -1), vmax.view(1, -1))
        loss = ((rows - y) ** 2).mean()
        loss.backward()
        step_fn(sub, {k: p[k].grad for k in keys}, st, lr, weight_decay=weight_decay)
        losses.append(float(loss.item()))
    return {k: v.detach() for k, v in p.items()}, losses


def pcn_inverse(sd: Dict[str, Tensor], xd: Tensor, masks: Tensor, vmin: Tensor, vmax: Tensor, new_min: float = -1.0,
                new_max: float = 1.0, output_fn: Optional[str] = "tanh", output_scale: Optional[float] = None,
                prefix: str = "flow_net.net.network.") -> Tensor:
    """PathConnectedNet.inverse (path_connected_net.py:87-122) on rows (N, C): NormNet.inverse (MinMax.transform, the flows'
    inverses in reverse order - ActNorm: (z - t) exp(-s); MaskedAffineFlow: zm + (1 - b)(z - t(zm)) exp(-s(zm)) - then
    MinMax.inverse_transform), then inverse_1b1_linear: (x - bias) / weight."""
    z = minmax(xd, vmin.view(1, -1), vmax.view(1, -1), new_min, new_max)
    for f in reversed(range(masks.shape[0])):
        z = (z - sd[f"{prefix}flows.{2 * f + 1}.t"]) * torch.exp(-sd[f"{prefix}flows.{2 * f + 1}.s"])
        b = masks[f].to(z.dtype).view(1, -1)
        zm = b * z
        s = rnvp_mlp(sd, f"{prefix}flows.{2 * f}.s.", zm, output_fn, output_scale)
        t = rnvp_mlp(sd, f"{prefix}flows.{2 * f}.t.", zm, output_fn, output_scale)
        z = zm + (1 - b) * (z - t) * torch.exp(-s)
    v = minmax(z, new_min, new_max, vmin.view(1, -1), vmax.view(1, -1))
    return (1.0 / sd["linear.weight"].view(1, -1)) * (v - sd["linear.bias"].view(1, -1))


def awesome_loss(output: Tensor, target: Tensor, alpha: float = 1.0, scribble_percentage: float = 1.0, extra_penalty: bool = False
                 ) -> Tensor:
    """AwesomeLoss.__call__ with the default BCE criterion (awesome/measures/awesome_loss.py:39-65): output (..., n, 2) =
    (seg, prior) per pixel; the first floor(n * scribble_percentage) pixels carry targets; with extra_penalty the loss becomes
    0.1 * loss + 100 * mean((prior - (seg > .5))^2) over output[..., random:, :] (the reference uses its COUNT of random
    pixels as the start index, :58-59)."""
    total = output.shape[-2]
    n_scr = int(math.floor(total * scribble_percentage))
    rnd = total - n_scr
    seg, prior = output[..., :n_scr, 0][..., None], output[..., :n_scr, 1][..., None]
    loss = F.binary_cross_entropy(seg, target) + alpha * F.binary_cross_entropy(prior, target)
    if extra_penalty and rnd > 0:
        seg_r, prior_r = output[..., rnd:, 0][..., None], output[..., rnd:, 1][..., None]
        loss = 0.1 * loss + 100 * torch.mean((prior_r - (seg_r > 0.5).float()) ** 2)
    return loss


def fit_pcn_minibatch(sd0: Dict[str, Tensor], frame_rows: Sequence[Tensor], frame_unaries: Sequence[Tensor], num_epochs: int,
                      batch_size: int, masks: Tensor, vmin: Tensor, vmax: Tensor, lr: float = 1e-3, flow_weight_decay: float = 1e-5,
                      plateau: Optional[dict] = None, **kw) -> Tuple[Dict[str, Tensor], List[float]]:
    """_non_prior_based_pretrain (path_connected_net.py:631-713): one network over all frames, one Adamax step per mini-batch of
    frames (DataLoader without shuffle), enforce_convexity after every step, ReduceLROnPlateau stepped per epoch with the mean
    batch loss.  frame_rows[t] (HW, C), frame_unaries[t] (HW, 1).  Returns (params, epoch losses)."""
    p = {k: v.detach().clone().requires_grad_(True) for k, v in sd0.items()}
    st = AdamState(p)
    sched = PlateauState(lr, **plateau) if plateau is not None else None
    cur_lr = lr
    flow_keys = [k for k in p if k.startswith("flow_net.")]
    other_keys = [k for k in p if not k.startswith("flow_net.")]
    T = len(frame_rows)
    epoch_losses: List[float] = []
    for _ in range(num_epochs):
        batches = [list(range(i, min(i + batch_size, T))) for i in range(0, T, batch_size)]
        el = 0.0
        for b in batches:
            rows = torch.cat([frame_rows[t] for t in b], 0)
            un = torch.cat([frame_unaries[t] for t in b], 0)
            for v in p.values():
                v.grad = None
            out = torch.sigmoid(pcn_forward(p, rows, masks, vmin, vmax, **kw))
            loss = weighted_loss(out.reshape(1, 1, -1, 1), un.reshape(1, 1, -1, 1), "se", "none")
            loss.backward()
            st.step += 1
            for keys, wd in ((flow_keys, flow_weight_decay), (other_keys, 0.0)):
                sub, g = {k: p[k] for k in keys}, {k: p[k].grad for k in keys}
                sst = AdamState.__new__(AdamState)
                sst.step, sst.m, sst.v = st.step - 1, st.m, st.v
                adamax_step(sub, g, sst, cur_lr, weight_decay=wd)
            with torch.no_grad():
                for k in p:
                    if k.startswith("convex_net.") and k.endswith("ln.weight") and not k.startswith("convex_net.input"):
                        p[k].copy_(F.relu(p[k]))
            el += float(loss.item()) / len(batches)
        epoch_losses.append(el)
        if sched is not None:
            cur_lr = sched.step(el)
    return {k: v.detach() for k, v in p.items()}, epoch_losses


def fcnet_forward(sd: Dict[str, Tensor], rows: Tensor) -> Tensor:
    """FCNet(in_type='xy').forward (awesome/model/fc_net.py:44-59): Linear, ReLU, depth x [Linear, ReLU], Linear on rows (N, C).
    Keys: model.0.*, model.{2+k}.0.*, model.{2+depth}.*."""
    depth = sum(1 for k in sd if k.endswith(".0.weight") and k.count(".") == 3)
    h = F.relu(F.linear(rows, sd["model.0.weight"], sd["model.0.bias"]))
    for k in range(depth):
        h = F.relu(F.linear(h, sd[f"model.{2 + k}.0.weight"], sd[f"model.{2 + k}.0.bias"]))
    return F.linear(h, sd[f"model.{2 + depth}.weight"], sd[f"model.{2 + depth}.bias"])


def fcnet_as_icnn(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """The same network in ConvexNextNet keys with zero skip weights (what the HIP kernels evaluate)."""
    depth = sum(1 for k in sd if k.endswith(".0.weight") and k.count(".") == 3)
    h, c = sd["model.0.weight"].shape
    out = {"input.weight": sd["model.0.weight"], "input.bias": sd["model.0.bias"]}
    for k in range(depth):
        out[f"skip.{k}.ln.weight"], out[f"skip.{k}.ln.bias"] = sd[f"model.{2 + k}.0.weight"], sd[f"model.{2 + k}.0.bias"]
        out[f"skip.{k}.skp.weight"] = torch.zeros(h, c)
    out["out.ln.weight"], out["out.ln.bias"] = sd[f"model.{2 + depth}.weight"], sd[f"model.{2 + depth}.bias"]
    out["out.skp.weight"] = torch.zeros(1, c)
    return out
