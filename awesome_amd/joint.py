"""Host-side driver of the fused joint-training step (include/inrfit.h: inrfit_joint_step / _pcn_joint_step / _cdn_joint_step).

One call = everything of TorchAgent._perform_step (awesome/agent/torch_agent.py:428-551) that lies behind the segmentation
network's output, for ONE image: prior forward on this image's parameter row, sigmoid, the composite loss (FBMSJointLoss, or
AwesomeImageLoss before its extra penalty), d loss / d seg for the backbone, the prior's backward from the activations of that same
pass, Adam / Adamax + enforce_convexity on the row in place.  Nothing syncs with the host."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib as L
from . import icnn as K

Tensor = torch.Tensor


@dataclass
class JointStepResult:
    loss: Tensor                 # [4] device: loss, mean weighted crit(seg), mean penalty, clip factor
    dseg: Tensor                 # [N] d loss / d seg
    prior_logits: Tensor         # [N] the prior's pre-sigmoid output of this step's forward
    status: Tensor               # [1] int32: 1 = non-finite loss, nothing was updated


def joint_desc(kind: str = "bce", weight_mode: str = "sssdms", ratio: float = 1.0, alpha: float = 1.0, beta: float = 1.0,
               clip_penalty: bool = True, form: int = L.JOINT_FBMS, prior_kind: str = "bce", prior_weight_mode: str = "none",
               prior_ratio: float = 1.0, gamma: float = 1.0, extra_penalty: bool = False, n_scribble: int = 0,
               class_targets: bool = False, noneclass=None) -> L.InrJointLossDesc:
    return L.InrJointLossDesc(L.LOSS_KINDS[kind], L.WEIGHT_MODES[weight_mode], float(ratio), float(alpha), float(beta),
                              int(bool(clip_penalty)), int(form), L.LOSS_KINDS[prior_kind], L.WEIGHT_MODES[prior_weight_mode],
                              float(prior_ratio), float(gamma), int(bool(extra_penalty)), int(n_scribble),
                              int(bool(class_targets)), int(noneclass is not None), float(noneclass if noneclass is not None else 0.0))


def _opt_desc(optimizer: str, lr: float, betas, eps: float, weight_decay: float, clamp: bool) -> L.InrOptDesc:
    return L.InrOptDesc(L.OPT_KINDS[optimizer], float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
                        int(bool(clamp)), 0, 0, 1.0, 0.0, 0.0, 0.0, 0, 0, 0)


def _outputs(n: int, dev) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    return (L.scratch(4, dtype=torch.float32, device=dev), L.scratch(n, dtype=torch.float32, device=dev),
            L.scratch(n, dtype=torch.float32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))


_ws_cache = {}


def _workspace(key, nbytes_fn, dev) -> Tensor:
    """One workspace per (model shape, grid size, device): a joint epoch calls the step thousands of times."""
    k = (key, str(dev))
    ws = _ws_cache.get(k)
    if ws is None or L.POISON:
        nbytes = int(nbytes_fn())
        if nbytes < 0:
            L.check(nbytes, "joint step workspace")
        ws = L.scratch(nbytes // 4 + 64, dtype=torch.float32, device=dev)
        _ws_cache[k] = ws
    return ws


def joint_step(spec: K.IcnnSpec, params: Tensor, opt_state: Tensor, grid: K.Grid, seg: Tensor, target: Tensor,
               desc: L.InrJointLossDesc, step: int, lr: float, optimizer: str = "adam", betas=(0.9, 0.999), eps: float = 1e-8,
               weight_decay: float = 0.0, clamp: bool = True) -> JointStepResult:
    """ICNN prior (ConvexNet / ConvexNextNet).  `params` [P] and `opt_state` [2P + 8] are updated IN PLACE."""
    params, seg, target = K._check_dev(params, "params"), K._check_dev(seg, "seg"), K._check_dev(target, "target")
    dev, n = params.device, grid.n_points
    assert params.numel() == spec.n_params and seg.numel() == n and target.numel() == n
    assert opt_state.numel() == 2 * spec.n_params + L.INR_OPT_HEADER_FLOATS and opt_state.is_contiguous()
    md, gd, od = spec.desc(), grid.desc(), _opt_desc(optimizer, lr, betas, eps, weight_decay, clamp)
    lib = L.load()
    ws = _workspace(("icnn", spec, n), lambda: lib.inrfit_joint_step_workspace_bytes(C.byref(md), C.byref(gd)), dev)
    loss, dseg, logits, status = _outputs(n, dev)
    rc = lib.inrfit_joint_step(C.byref(md), params.data_ptr(), opt_state.data_ptr(), C.byref(gd), seg.data_ptr(), target.data_ptr(),
                               C.byref(desc), C.byref(od), int(step), loss.data_ptr(), dseg.data_ptr(), logits.data_ptr(),
                               status.data_ptr(), ws.data_ptr(), ws.numel() * 4, K._stream_ptr(dev))
    L.check(rc, "inrfit_joint_step")
    return JointStepResult(loss, dseg, logits, status)


def pcn_joint_step(ispec: K.IcnnSpec, rspec, icnn_params: Tensor, flow_params: Tensor, icnn_opt_state: Tensor, flow_opt_state: Tensor,
                   grid: K.Grid, seg: Tensor, target: Tensor, desc: L.InrJointLossDesc, step: int, lr: float,
                   optimizer: str = "adam", betas=(0.9, 0.999), eps: float = 1e-8, flow_weight_decay: float = 0.0) -> JointStepResult:
    """PathConnectedNet prior (ICNN behind the RealNVP deformation); every tensor is one row, updated in place."""
    dev, n = icnn_params.device, grid.n_points
    md, rd, gd, od = ispec.desc(), rspec.desc(), grid.desc(), _opt_desc(optimizer, lr, betas, eps, 0.0, True)
    lib = L.load()
    ws = _workspace(("pcn", ispec, rspec, n),
                    lambda: lib.inrfit_pcn_workspace_bytes(C.byref(md), C.byref(rd), C.byref(gd), 1) + lib.inrfit_joint_loss_workspace_bytes(n)
                    + 4 * n + 1024, dev)
    loss, dseg, logits, status = _outputs(n, dev)
    rc = lib.inrfit_pcn_joint_step(C.byref(md), C.byref(rd), icnn_params.data_ptr(), flow_params.data_ptr(), icnn_opt_state.data_ptr(),
                                   flow_opt_state.data_ptr(), C.byref(gd), seg.data_ptr(), target.data_ptr(), C.byref(desc),
                                   C.byref(od), float(flow_weight_decay), int(step), loss.data_ptr(), dseg.data_ptr(),
                                   logits.data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel() * 4, K._stream_ptr(dev))
    L.check(rc, "inrfit_pcn_joint_step")
    return JointStepResult(loss, dseg, logits, status)


def cdn_joint_step(ispec: K.IcnnSpec, fspec, icnn_params: Tensor, flow_params: Tensor, icnn_opt_state: Tensor, flow_opt_state: Tensor,
                   grid: K.Grid, seg: Tensor, target: Tensor, desc: L.InrJointLossDesc, step: int, lr: float,
                   betas=(0.9, 0.999), eps: float = 1e-8, weight_decay_on_weight_g: float = 0.0) -> JointStepResult:
    """ConvexDiffeomorphismNet prior (ICNN behind the weight-normed coupling flow); Adam only, like its pretrain loop."""
    dev, n = icnn_params.device, grid.n_points
    md, fd, gd, od = ispec.desc(), fspec.desc(), grid.desc(), _opt_desc("adam", lr, betas, eps, 0.0, True)
    lib = L.load()
    ws = _workspace(("cdn", ispec, fspec, n),
                    lambda: lib.inrfit_cdn_workspace_bytes(C.byref(md), C.byref(fd), C.byref(gd), 1) + lib.inrfit_joint_loss_workspace_bytes(n)
                    + 4 * n + 1024, dev)
    loss, dseg, logits, status = _outputs(n, dev)
    rc = lib.inrfit_cdn_joint_step(C.byref(md), C.byref(fd), icnn_params.data_ptr(), flow_params.data_ptr(), icnn_opt_state.data_ptr(),
                                   flow_opt_state.data_ptr(), C.byref(gd), seg.data_ptr(), target.data_ptr(), C.byref(desc),
                                   C.byref(od), float(weight_decay_on_weight_g), int(step), loss.data_ptr(), dseg.data_ptr(),
                                   logits.data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel() * 4, K._stream_ptr(dev))
    L.check(rc, "inrfit_cdn_joint_step")
    return JointStepResult(loss, dseg, logits, status)
