"""Loss / metric objects with the reference's call contract `(output, target, **kwargs)`, usable as
`loss_type:` in a config.  Called on tensors they run as torch ops on whatever device the tensors live on; handed to
the fused fitter (awesome_amd.fitter) they are translated into the kernel's InrLossDesc by `criterion_to_desc`, so the
E-step loop never leaves the device.  MIOU on GPU tensors is the HIP counting kernel."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from .. import icnn as K


_REDUCTIONS = {"sum": torch.sum, "mean": torch.mean, "max": torch.max, "min": torch.min}


def _reduce(loss: torch.Tensor, reduction: str, reduction_dim=None) -> torch.Tensor:
    """awesome/measures/torch_reducable_metric.py:36-55 compute_return_value."""
    if reduction == "none":
        return loss
    return _REDUCTIONS[reduction](loss, **({} if reduction_dim is None else {"dim": reduction_dim}))


class SE:
    """awesome/measures/se.py:21-23 - squared error with sum/mean/none reduction."""

    def __init__(self, reduction: str = "mean", name: Optional[str] = None, reduction_dim=None, **kwargs):
        if reduction not in ("sum", "mean", "none", "max", "min"):
            raise ValueError(f"Value {reduction} for reduction is invalid.")
        self.reduction, self.name, self.reduction_dim = reduction, name, reduction_dim

    def __call__(self, output: torch.Tensor, target: torch.Tensor, **kwargs) -> torch.Tensor:
        return _reduce((target - output) ** 2, self.reduction, self.reduction_dim)

    def get_name(self) -> str:
        return self.name or (self.reduction[0].upper() + "SE")


class WeightedLoss:
    """awesome/measures/weighted_loss.py:11-92: criterion (reduction forced to none) on CLASS targets {0 = fg, 1 = bg}, pixels whose
    target equals `noneclass` dropped first (:71-74; the weak labels of the FBMS configs mark unlabeled pixels with 2), per-class
    weight on the fg pixels from the bg / fg count ratio (modes equal / sssdms, :38-62), then the reduction.  The criterion of 153
    of the reference's FBMSJointLoss configs is WeightedLoss(BCELoss, mode sssdms, noneclass 2)."""

    MODES = ("none", "sssdms", "equal")

    def __init__(self, criterion=None, noneclass: Optional[float] = None, name: Optional[str] = None,
                 forward_kwargs_criterion: bool = False, reduction: str = "mean", reduction_dim=None, mode: str = "none", **kwargs):
        if reduction not in ("sum", "mean", "none", "max", "min"):
            raise ValueError(f"Value {reduction} for reduction is invalid.")
        if criterion is None:
            raise ValueError("criterion must be specified")
        if mode not in self.MODES:
            raise ValueError(f"Mode {mode} is not supported")
        self.name, self.reduction, self.reduction_dim = name, reduction, reduction_dim
        self.criterion, self.forward_kwargs_criterion, self.noneclass, self.mode = criterion, forward_kwargs_criterion, noneclass, mode
        if hasattr(criterion, "reduction"):
            criterion.reduction = "none"

    def _classes(self, target: torch.Tensor):
        """(fg pixels, bg pixels) as masks: class targets 0 / 1 (weighted_loss.py:42-43)."""
        return target == 0, target == 1

    def _weight(self, target: torch.Tensor) -> torch.Tensor:
        fg, bg = self._classes(target)
        cc = bg.sum().float() / fg.sum().float()
        if self.mode == "sssdms":
            wv = torch.round(cc / 10) + 1
        elif self.mode == "ratio":
            wv = (cc - 1) * self.ratio + 1
        else:
            wv = cc
        return torch.where(fg, wv, torch.ones_like(target))

    def __call__(self, output: torch.Tensor, target: torch.Tensor, **kwargs) -> torch.Tensor:
        o, t = output, target
        if self.noneclass is not None:
            keep = target != self.noneclass
            o, t = output[keep], target[keep]
        if t.dim() >= 2:
            if t.dim() == 3:
                raise ValueError("Expected 4D target, got 3D target")
            shape = (t.shape[0] * t.shape[2] * t.shape[3], t.shape[1])
            o, t = o.permute(0, 2, 3, 1).reshape(shape), t.permute(0, 2, 3, 1).reshape(shape)
        loss = self.criterion(o, t, **(kwargs if self.forward_kwargs_criterion else {}))
        if self.mode != "none":
            loss = loss * self._weight(t)
        return _reduce(loss, self.reduction, self.reduction_dim)

    def get_name(self) -> str:
        return self.name or type(self).__name__


class UnariesWeightedLoss(WeightedLoss):
    """awesome/measures/unaries_weighted_loss.py:9-69: WeightedLoss on UNARIES - fg is `target < 0.5`, the counts are those of
    `target >= 0.5` - plus the `ratio` mode."""

    MODES = ("none", "equal", "ratio", "sssdms")

    def __init__(self, criterion=None, noneclass=None, name=None, forward_kwargs_criterion: bool = False, reduction: str = "mean",
                 reduction_dim=None, mode: str = "none", ratio: float = 1.0, **kwargs):
        super().__init__(criterion=criterion, noneclass=noneclass, name=name, forward_kwargs_criterion=forward_kwargs_criterion,
                         reduction=reduction, reduction_dim=reduction_dim, mode=mode)
        self.ratio = ratio

    def _classes(self, target: torch.Tensor):
        return target < 0.5, target >= 0.5      # unaries_weighted_loss.py:38-39


class UnariesConversionLoss:
    """awesome/measures/unaries_conversion_loss.py:8-31: the criterion on BINARISED unaries, `target = (target >= 0.5).float()`
    (the pretrain criterion of every ConvexDiffeomorphismNet config: UnariesConversionLoss(SE('mean')))."""

    def __init__(self, criterion=None, name: Optional[str] = None, **kwargs):
        self.criterion, self.name = criterion, name

    def __call__(self, output: torch.Tensor, target: torch.Tensor, **kwargs) -> torch.Tensor:
        return self.criterion(output, (target >= 0.5).float(), **kwargs)

    def get_name(self) -> str:
        return self.name or ("UC" + self.criterion.get_name())


def criterion_targets(criterion, unaries: torch.Tensor) -> torch.Tensor:
    """The targets the fused fit kernels must see for `criterion`: binarised under UnariesConversionLoss, unchanged otherwise."""
    return (unaries >= 0.5).to(unaries.dtype) if isinstance(criterion, UnariesConversionLoss) else unaries


class MIOU:
    """awesome/measures/miou.py:29-48 with average='binary'."""

    def __init__(self, invert: bool = False, average: str = "binary", name: Optional[str] = None, **kwargs):
        if average != "binary":
            raise ValueError("only average='binary' (the mode the reference runner uses) is implemented")
        self.invert, self.name = invert, name

    def __call__(self, output: torch.Tensor, target: torch.Tensor, **kwargs) -> torch.Tensor:
        if output.is_cuda:
            return K.miou(output.reshape(1, -1).float(), target.reshape(1, -1).float(), 0.5, 0.5, self.invert)[0]
        o, t = output.reshape(-1).float(), target.reshape(-1).float()
        if self.invert:
            o, t = 1.0 - o, 1.0 - t
        if bool(torch.all(t == 0)):
            return torch.tensor(0.0)
        ob, tb = o == 1.0, t == 1.0
        return (ob & tb).sum().float() / (ob | tb).sum().float()

    def get_name(self) -> str:
        return self.name or "MIOU"


def joint_criterion_form(criterion):
    """How the composite losses' kernels (csrc/joint_loss.h) evaluate `criterion`:
    (loss kind, weight mode, ratio, class_targets, noneclass) - or TypeError if it has no kernel form.  Unlike the per-image fits
    (`criterion_to_desc`) these kernels know WeightedLoss on class labels (fg = target == 0, bg = target == 1) and the `noneclass`
    mask (weighted_loss.py:38-74); a UnariesConversionLoss would need its targets binarised first and is refused."""
    mode, ratio, inner, class_targets, noneclass = "none", 1.0, criterion, False, None
    if isinstance(criterion, UnariesConversionLoss):
        raise TypeError("UnariesConversionLoss binarises its targets; the composite losses' kernels read them as they are")
    if isinstance(criterion, WeightedLoss):
        if criterion.reduction != "mean" or criterion.reduction_dim is not None or criterion.forward_kwargs_criterion:
            raise TypeError("only reduction='mean' over all pixels has a kernel form")
        mode, ratio, inner = criterion.mode, getattr(criterion, "ratio", 1.0), criterion.criterion
        class_targets = not isinstance(criterion, UnariesWeightedLoss)
        noneclass = None if criterion.noneclass is None else float(criterion.noneclass)
    elif getattr(inner, "reduction", "mean") != "mean":
        raise TypeError("only reduction='mean' has a kernel form")
    if isinstance(inner, SE) and inner.reduction_dim is None:
        return "se", mode, ratio, class_targets, noneclass
    if isinstance(inner, torch.nn.BCELoss) and inner.weight is None:
        return "bce", mode, ratio, class_targets, noneclass
    raise TypeError(f"{type(criterion).__name__} has no kernel form (supported: SE, BCELoss, WeightedLoss / UnariesWeightedLoss of those)")


def _target_fields(class_targets: bool, noneclass):
    return int(bool(class_targets)), int(noneclass is not None), float(noneclass if noneclass is not None else 0.0)


class AwesomeImageLoss:
    """awesome/measures/awesome_image_loss.py:34-53: crit(seg,t) + alpha*crit(prior,t) [+ penalty].

    On CUDA tensors (B, 2, H, W) with criteria the kernels know (BCELoss / SE, optionally inside UnariesWeightedLoss) it runs as the
    fused HIP loss `inrfit_joint_loss` (form INR_JOINT_AWESOME_IMAGE: value and both gradient channels in three launches); any other
    criterion is composed from torch ops."""

    def __init__(self, criterion=None, prior_criterion=None, alpha=1.0, beta=100.0, gamma=0.1, name=None, **kwargs):
        self.criterion = criterion or torch.nn.BCELoss()
        self.prior_criterion = prior_criterion or torch.nn.BCELoss()
        self.alpha, self.beta, self.gamma, self.name = alpha, beta, gamma, name
        self.extra_penalty = False  # toggled by the runner (awesome/run/awesome_runner.py:351-371)

    def joint_desc(self):
        """InrJointLossDesc of this loss, or None if a criterion has no kernel form."""
        from .. import _lib as L
        try:
            kind, mode, ratio, ct, nc = joint_criterion_form(self.criterion)
            pkind, pmode, pratio, pct, pnc = joint_criterion_form(self.prior_criterion)
        except TypeError:
            return None
        if (ct, nc) != (pct, pnc):      # one reading of the targets per kernel pass
            return None
        return L.InrJointLossDesc(L.LOSS_KINDS[kind], L.WEIGHT_MODES[mode], float(ratio), float(self.alpha), float(self.beta), 0,
                                  L.JOINT_AWESOME_IMAGE, L.LOSS_KINDS[pkind], L.WEIGHT_MODES[pmode], float(pratio),
                                  float(self.gamma), int(bool(self.extra_penalty)), 0, *_target_fields(ct, nc))

    def __call__(self, output: torch.Tensor, target: torch.Tensor, **kwargs) -> torch.Tensor:
        if output.is_cuda and output.dim() == 4 and output.shape[1] == 2:
            desc = self.joint_desc()
            if desc is not None:
                return _FusedJointLoss.apply(output, target, desc)
        c = output.shape[1] // 2
        seg, prior = output[:, :c], output[:, c:]
        loss = self.criterion(seg, target) + self.alpha * self.prior_criterion(prior, target)
        if self.extra_penalty:
            loss = self.gamma * loss + self.beta * torch.mean((prior - (seg > 0.5).float()) ** 2)
        return loss

    def get_name(self) -> str:
        return self.name or type(self).__name__


class AwesomeLoss:
    """awesome/measures/awesome_loss.py:45-65 (pixel mode): output (..., n_pixels, 2) = (segmentation, prior) per pixel, the
    first floor(n * scribble_percentage) pixels are scribbles with targets, the rest random pixels for the align term:
    crit(seg, t) + alpha*crit(prior, t)  [ -> 0.1*loss + 100*mean((prior_rand - (seg_rand > .5))^2) with extra_penalty ].
    On CUDA tensors with a criterion the kernels know: the fused HIP loss (form INR_JOINT_AWESOME_PIXEL)."""

    def __init__(self, criterion=None, alpha: float = 1.0, name=None, scribble_percentage: float = 1.0, **kwargs):
        self.criterion = criterion or torch.nn.BCELoss()
        self.alpha, self.name, self.scribble_percentage = alpha, name, scribble_percentage
        self.extra_penalty = False

    def joint_desc(self, total: int = 0):
        from .. import _lib as L
        try:
            kind, mode, ratio, ct, nc = joint_criterion_form(self.criterion)
        except TypeError:
            return None
        n_scr = int(total * self.scribble_percentage // 1)
        return L.InrJointLossDesc(L.LOSS_KINDS[kind], L.WEIGHT_MODES[mode], float(ratio), float(self.alpha), 100.0, 0,
                                  L.JOINT_AWESOME_PIXEL, L.LOSS_KINDS[kind], L.WEIGHT_MODES[mode], float(ratio), 0.1,
                                  int(bool(self.extra_penalty)), n_scr, *_target_fields(ct, nc))

    def __call__(self, output: torch.Tensor, target: torch.Tensor, **kwargs) -> torch.Tensor:
        total = output.shape[-2]
        n_scr = int(total * self.scribble_percentage // 1)
        n_rand = total - n_scr
        if output.is_cuda and output.shape[-1] == 2 and n_scr > 0:
            desc = self.joint_desc(total)
            if desc is not None:
                return _FusedJointLoss.apply(output, target, desc)
        seg, prior = output[..., :n_scr, 0:1], output[..., :n_scr, 1:2]
        loss = self.criterion(seg, target) + self.alpha * self.criterion(prior, target)
        if self.extra_penalty and n_rand > 0:
            # the reference slices [random:] (its count of random pixels used as a start index, awesome_loss.py:58-59)
            seg_r, prior_r = output[..., n_rand:, 0:1], output[..., n_rand:, 1:2]
            loss = 0.1 * loss + 100.0 * torch.mean((prior_r - (seg_r > 0.5).float()) ** 2)
        return loss

    def get_name(self) -> str:
        return self.name or type(self).__name__


class _FusedJointLoss(torch.autograd.Function):
    """A composite joint loss (InrJointLossDesc.form) value + gradient in three HIP launches (inrfit_joint_loss), no host sync.
    Image forms: output (B, 2, H, W); pixel form: output (..., n, 2)."""

    @staticmethod
    def forward(ctx, output: torch.Tensor, target: torch.Tensor, desc):
        import ctypes as C
        from .. import _lib as L
        out = output.detach().contiguous().to(torch.float32)
        if desc.form == L.JOINT_AWESOME_PIXEL:
            hw = out.shape[-2]
            b = out.numel() // (2 * hw)
        else:
            b, c2 = out.shape[0], out.shape[1]
            hw = out.numel() // (b * c2)
        tgt = target.detach().contiguous().to(torch.float32)
        lib = L.load()
        nbytes = int(lib.inrfit_joint_loss_workspace_bytes(b * hw))
        ws = L.scratch(nbytes // 4 + 1, dtype=torch.float32, device=out.device)
        res = L.scratch(4, dtype=torch.float32, device=out.device)
        dout = L.scratch_like(out)
        rc = lib.inrfit_joint_loss(out.data_ptr(), tgt.data_ptr(), b, hw, C.byref(desc), res.data_ptr(), dout.data_ptr(), ws.data_ptr(),
                                   ws.numel() * 4, K._stream_ptr(out.device))
        L.check(rc, "inrfit_joint_loss")
        ctx.save_for_backward(dout)
        ctx.terms = res
        return res[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        (dout,) = ctx.saved_tensors
        return dout * grad_out, None, None


class FBMSJointLoss:
    """awesome/measures/fbms_joint_loss.py:35-59: alpha*crit(seg,t) + clip(beta*SE(prior, seg)).

    On CUDA tensors with a criterion the kernels know (BCELoss / SE, optionally inside UnariesWeightedLoss) and the default SE
    'mean' penalty it runs as the fused HIP loss `inrfit_joint_loss` (value and gradient, the clip decided on the device);
    any other criterion is composed from torch ops on whatever device the tensors live on."""

    def __init__(self, criterion=None, penalty_criterion=None, alpha=1.0, beta=1.0, clip_penalty=True, name=None, **kwargs):
        self.criterion = criterion or UnariesWeightedLoss(torch.nn.BCELoss(), mode="sssdms")
        self.penalty_criterion = penalty_criterion or SE("mean")
        self.alpha, self.beta, self.clip_penalty, self.name = alpha, beta, clip_penalty, name

    def _fused_desc(self, output: torch.Tensor):
        from .. import _lib as L
        if not output.is_cuda or output.dim() != 4 or output.shape[1] != 2:
            return None
        return self.joint_desc()

    def joint_desc(self):
        """InrJointLossDesc of this loss for the fused joint step (awesome_amd.agent.JointTrainer), or None."""
        from .. import _lib as L
        if not (isinstance(self.penalty_criterion, SE) and self.penalty_criterion.reduction == "mean"
                and self.penalty_criterion.reduction_dim is None):
            return None
        try:
            kind, mode, ratio, ct, nc = joint_criterion_form(self.criterion)
        except TypeError:
            return None
        return L.InrJointLossDesc(L.LOSS_KINDS[kind], L.WEIGHT_MODES[mode], float(ratio), float(self.alpha), float(self.beta),
                                  int(bool(self.clip_penalty)), L.JOINT_FBMS, 0, 0, 1.0, 1.0, 0, 0, *_target_fields(ct, nc))

    def __call__(self, output: torch.Tensor, target: torch.Tensor, **kwargs) -> torch.Tensor:
        desc = self._fused_desc(output)
        if desc is not None:
            return _FusedJointLoss.apply(output, target, desc)
        c = output.shape[1] // 2
        seg, prior = output[:, :c], output[:, c:]
        seg_loss = self.alpha * self.criterion(seg, target)
        pen = self.beta * self.penalty_criterion(prior, seg)
        if self.clip_penalty:
            # the reference branches on the host (`if penalty > seg_loss`, one device->host sync per step); same values, no sync
            scale = torch.where(pen > seg_loss, seg_loss / pen, torch.ones_like(pen)).detach()
            pen = pen * scale
        return seg_loss + pen

    def get_name(self) -> str:
        return self.name or type(self).__name__


class TV(torch.nn.Module):
    """awesome/measures/tv.py:6-55: mean squared forward differences of a (B, C, H, W) tensor along both image axes, optionally
    weighted by exp(-5 * the same statistic of the clean image) (the `_input[-1]["clean_image"]` convention of the joint losses)."""

    def forward(self, x: torch.Tensor, _input=None, **kwargs) -> torch.Tensor:
        b, _, h, w = x.shape
        count_h = x[:, :, 1:, :][0].numel()
        count_w = x[:, :, :, 1:][0].numel()
        h_tv = torch.pow(x[:, :, 1:, :] - x[:, :, :h - 1, :], 2).sum()
        w_tv = torch.pow(x[:, :, :, 1:] - x[:, :, :, :w - 1], 2).sum()
        weight = 1
        image = None
        if _input is not None and len(_input) > 0 and isinstance(_input[-1], dict):
            image = _input[-1].get("clean_image", None)
        if image is not None:
            g = torch.mean(image, dim=1)
            gh = torch.pow(g[:, 1:, :] - g[:, :-1, :], 2).sum()
            gw = torch.pow(g[:, :, 1:] - g[:, :, :-1], 2).sum()
            weight = torch.exp(-5 * (torch.abs(gh / count_h) + torch.abs(gw / count_w)) / b) / 2
        return weight * 2 * (h_tv / count_h + w_tv / count_w) / b


class RegularizerLoss:
    """awesome/measures/regularizer_loss.py:9-43: criterion(output, target) + tau * regularizer(output, **kwargs)."""

    def __init__(self, criterion=None, tau: float = 0.0, regularizer=None, name: Optional[str] = None, **kwargs):
        if criterion is None:
            raise ValueError("criterion must not be None")
        if regularizer is None and tau > 0.0:
            raise ValueError("regularizer must not be None if tau is larger zero!")
        self.name, self.criterion, self.tau, self.regularizer = name, criterion, tau, regularizer

    def __call__(self, output: torch.Tensor, target: torch.Tensor, **kwargs) -> torch.Tensor:
        loss = self.criterion(output, target)
        if self.tau > 0.0:
            loss = loss + self.tau * self.regularizer(output, **kwargs)
        return loss

    def get_name(self) -> str:
        return self.name or type(self).__name__


class GradientPenaltyLoss:
    """awesome/measures/gradient_penalty_loss.py:10-118 (the criterion of the CNNNet convexity configs): the inner criterion on the pixels
    whose target is not `noneclass`, plus - while `apply_gradient_penalty` is on and the step's `_input` = (image, xy / features, ...)
    is handed in - penalties on the mean absolute gradient of sum(output) w.r.t. the coordinates / semantic features (`xygrad`,
    `featgrad`, split by `xytype`) and w.r.t. the image (`rgbgrad`), differentiated through (create_graph).  The penalties are
    second-order autograd through the SEGMENTATION network: torch ops on whatever device the tensors live on, no kernel form (the
    inputs must require grad, `dataset_args.model_input_requires_grad` in the reference's configs)."""

    def __init__(self, criterion=None, apply_gradient_penalty: bool = False, xygrad: float = 0.0, rgbgrad: float = 0.0,
                 featgrad: float = 0.0, xytype: str = "xy", noneclass: Optional[float] = None, name: Optional[str] = None, **kwargs):
        if xytype not in ("xy", "feat", "featxy", "edge"):
            raise ValueError(f"xytype must be one of [xy, feat, featxy, edge] but is {xytype}")
        if criterion is None:
            raise ValueError("criterion must not be None")
        self.name, self.criterion = name, criterion
        self.xygrad, self.rgbgrad, self.featgrad, self.xytype = xygrad, rgbgrad, featgrad, xytype
        self.apply_gradient_penalty, self.noneclass = apply_gradient_penalty, noneclass

    def __call__(self, output: torch.Tensor, target: torch.Tensor, _input=None, **kwargs) -> torch.Tensor:
        original_output = output
        if self.noneclass is not None:
            keep = target != self.noneclass
            output, target = output[keep], target[keep]
        takes_kwargs = not isinstance(self.criterion, torch.nn.modules.loss._Loss)
        loss = self.criterion(output, target, **(kwargs if takes_kwargs else {}))
        if not self.apply_gradient_penalty:
            return loss
        if _input is None:
            raise ValueError("GradientPenaltyLoss needs _input to apply gradient penalty")
        img, raw_xy = _input[0], _input[1]
        output_sum = torch.sum(original_output)
        if self.xygrad > 0.0 or self.featgrad > 0.0:
            grad_raw = torch.autograd.grad(output_sum, raw_xy, retain_graph=True, create_graph=True)[0]
            mean_xy = mean_feat = None
            if self.xytype == "feat":
                mean_feat = torch.mean(torch.abs(grad_raw))
            elif self.xytype == "xy":
                mean_xy = torch.mean(torch.abs(grad_raw))
            elif self.xytype == "featxy":
                mean_xy = torch.mean(torch.abs(grad_raw[:, :2, ...]))
                mean_feat = torch.mean(torch.abs(grad_raw[:, 2:, ...]))   # semantic features are the last channels
            if self.xygrad > 0.0 and mean_xy is not None:
                loss = loss + self.xygrad * mean_xy
            if self.featgrad > 0.0 and mean_feat is not None:
                loss = loss + self.featgrad * mean_feat
        if self.rgbgrad > 0.0:
            grad_rgb = torch.autograd.grad(output_sum, img, retain_graph=True, create_graph=True)[0]
            loss = loss + self.rgbgrad * torch.mean(torch.abs(grad_rgb))
        return loss

    def get_name(self) -> str:
        return self.name or type(self).__name__


class AwesomeImageLossJoint:
    """awesome/measures/awesome_image_loss_joint.py:11-70 (the `segmentation_training_mode: multi` convexity configs): the same data
    terms as AwesomeImageLoss with ONE criterion for both channels; the alignment term is `mean((prior - seg)^2)` - on the soft
    segmentation output, where AwesomeImageLoss thresholds it - once `extra_penalty` is on, or `mean((prior - (seg > .5))^2)` from the
    start with `map_initially_on_segmentation`.  The criterion is called with the step's kwargs and told through its
    `apply_gradient_penalty` attribute not to add its penalty on the prior channel (:41-43)."""

    def __init__(self, criterion=None, alpha: float = 1.0, beta: float = 1.0, gamma: float = 1, name: Optional[str] = None,
                 map_initially_on_segmentation: bool = False, **kwargs):
        self.name = name
        self.criterion = criterion or torch.nn.BCELoss()
        self.alpha, self.beta, self.gamma = alpha, beta, gamma
        self.extra_penalty = False
        self.map_initially_on_segmentation = map_initially_on_segmentation

    def __call__(self, output: torch.Tensor, target: torch.Tensor, **kwargs) -> torch.Tensor:
        c = output.shape[1] // 2
        seg, prior = output[:, :c], output[:, c:]
        takes_kwargs = not isinstance(self.criterion, torch.nn.modules.loss._Loss)
        kw = kwargs if takes_kwargs else {}
        seg_loss = self.criterion(seg, target, **kw)
        self.criterion.apply_gradient_penalty = False
        prior_loss = self.criterion(prior, target, **kw)
        self.criterion.apply_gradient_penalty = True
        loss = seg_loss + self.alpha * prior_loss
        if self.extra_penalty:
            loss = self.gamma * loss + self.beta * torch.mean((prior - seg) ** 2)
        elif self.map_initially_on_segmentation:
            loss = self.gamma * loss + self.beta * torch.mean((prior - (seg > 0.5).float()) ** 2)
        return loss

    def get_name(self) -> str:
        return self.name or type(self).__name__


class AwesomeLossJoint:
    """awesome/measures/awesome_loss_joint.py:10-89 (pixel mode of the joint convexity configs): data terms on the first
    floor(n * scribble_percentage) pixels for both channels, and with `extra_penalty` `gamma * loss + beta * mean((prior - seg)^2)` on
    the pixels `[n - n_scribble, n)` - the reference's own slice, on the SOFT segmentation output."""

    def __init__(self, criterion=None, alpha: float = 1.0, beta: float = 1.0, gamma: float = 1, name: Optional[str] = None,
                 scribble_percentage: float = 1.0, **kwargs):
        self.name = name
        self.criterion = criterion or torch.nn.BCELoss()
        self.alpha, self.beta, self.gamma = alpha, beta, gamma
        self.extra_penalty = False
        self.scribble_percentage = scribble_percentage

    def __call__(self, output: torch.Tensor, target: torch.Tensor, **kwargs) -> torch.Tensor:
        total = output.shape[-2]
        n_scr = int(total * self.scribble_percentage // 1)
        n_rand = total - n_scr
        seg, prior = output[..., :n_scr, 0:1], output[..., :n_scr, 1:2]
        seg_loss = self.criterion(seg, target)
        self.criterion.apply_gradient_penalty = False
        prior_loss = self.criterion(prior, target)
        self.criterion.apply_gradient_penalty = True
        loss = seg_loss + self.alpha * prior_loss
        if self.extra_penalty and n_rand > 0:
            seg_r, prior_r = output[..., n_rand:, 0:1], output[..., n_rand:, 1:2]
            loss = self.gamma * loss + self.beta * torch.mean((prior_r - seg_r) ** 2)
        return loss

    def get_name(self) -> str:
        return self.name or type(self).__name__


def criterion_to_desc(criterion, conversion: str = "reject") -> Tuple[str, str, float]:
    """(loss kind, weight mode, ratio) for the fused kernels, or raise TypeError if the criterion has no fused form.

    A UnariesConversionLoss changes the TARGETS, not the criterion: a caller that binarises its targets with `criterion_targets`
    (the per-image fits) passes conversion="targets" and gets the inner criterion's form; everyone else (the composite joint losses,
    whose kernels read the targets as they are) keeps the default and gets the TypeError -> the torch composition."""
    mode, ratio, inner = "none", 1.0, criterion
    if isinstance(criterion, UnariesConversionLoss):
        if conversion != "targets":
            raise TypeError("UnariesConversionLoss binarises its targets; this caller does not (no fused form)")
        criterion = inner = criterion.criterion
    if isinstance(criterion, WeightedLoss):
        if criterion.noneclass is not None or type(criterion) is WeightedLoss:
            # class targets {0, 1, noneclass}: a form of the composite losses' kernels (joint_weighting), not of the per-image fits
            raise TypeError("WeightedLoss on class targets / with a noneclass has no fused per-image form")
        if criterion.reduction != "mean" or criterion.reduction_dim is not None:
            raise TypeError("only reduction='mean' over all pixels has a fused form")
        mode, ratio, inner = criterion.mode, criterion.ratio, criterion.criterion
    if isinstance(inner, SE) and inner.reduction_dim is None:
        return "se", mode, ratio
    if isinstance(inner, torch.nn.BCELoss) and inner.weight is None:
        return "bce", mode, ratio
    raise TypeError(f"{type(criterion).__name__} has no fused kernel form (supported: SE, BCELoss, UnariesWeightedLoss of those)")
