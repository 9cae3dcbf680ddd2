"""Weight-normalised coupling flow + the path-connected prior ICNN(flow(Ax + b)) with the reference's module surface.

Reference: WNLinear (awesome/model/real_nvp/resnet_1d.py:39-63), NormalBlock / WNScale / NormalizingFlow1D
(awesome/model/diffeomorphism_net.py:169-302), ConvexDiffeomorphismNet (awesome/model/convex_diffeomorphism_net.py:41-188).
Same constructor kwargs and state_dict keys (torch's legacy weight_norm parametrisation: `weight_g` / `weight_v`), same
creation order of the leaves (seeded construction = reference init).

`ConvexDiffeomorphismNet.forward` and its autograd backward run entirely on the HIP path (flow kernels + ICNN kernels,
`inrfit_cdn_forward` / `inrfit_cdn_loss_grad` with an external dL/dlogits); `fit_images` is the fused device-resident
form of `ConvexDiffeomorphismNet.pretrain`'s inner loop (`inrfit_cdn_fit`).  `NormalizingFlow1D.forward` (with autograd)
and `ConvexDiffeomorphismNet.get_deformation` run on `inrfit_flow_forward` / `inrfit_flow_backward`.  The leaf modules
(`WNLinear`, `NormalBlock`, `SimpleBackbone`, `WNScale`) are the containers of the parameters; called on their own they are
plain torch expressions of their one-line definitions (no kernel exists for a single 1 -> W -> 1 net: the unit of work on
the device is the whole flow)."""
from __future__ import annotations

import math
from typing import Any, Dict, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import flow as FL
from .. import icnn as K
from .convex_net import ConvexNextNet, _kaiming_uniform_reset
from .pretrainable_module import PriorFitMixin, center_of_mass


class _CdnFunction(torch.autograd.Function):
    """logits = ICNN(flow(A x + b)) on the HIP path; backward = gradients w.r.t. every parameter for a given dL/dlogits."""

    @staticmethod
    def forward(ctx, coords: torch.Tensor, ispec, fspec, n_icnn: int, *params: torch.Tensor):
        flat = lambda ps: torch.cat([p.reshape(-1) for p in ps]).to(torch.float32)[None].contiguous()  # noqa: E731
        ip, fp = flat(params[:n_icnn]), flat(params[n_icnn:])
        grid = K.Grid.explicit(coords)
        ctx.ispec, ctx.fspec, ctx.grid, ctx.n_icnn = ispec, fspec, grid, n_icnn
        ctx.shapes = [p.shape for p in params]
        ctx.save_for_backward(ip, fp)
        return FL.cdn_forward(ispec, fspec, ip, fp, grid)[0]

    @staticmethod
    def backward(ctx, dlogits: torch.Tensor):
        ip, fp = ctx.saved_tensors
        _, gi, gf = FL.cdn_loss_grad(ctx.ispec, ctx.fspec, ip, fp, ctx.grid, dlogits.contiguous()[None], loss="external")
        g = torch.cat([gi[0], gf[0]])
        outs, off = [], 0
        for shp in ctx.shapes:
            n = math.prod(shp) if len(shp) else 1
            outs.append(g[off:off + n].reshape(shp))
            off += n
        return (None, None, None, None, *outs)


class _FlowFunction(torch.autograd.Function):
    """deformed coords (2, N) = flow(A x + b) on the HIP path; backward = inrfit_flow_backward (parameter gradients only - the
    grid is an input, never a learned quantity)."""

    @staticmethod
    def forward(ctx, coords: torch.Tensor, fspec, *params: torch.Tensor):
        fp = torch.cat([p.reshape(-1) for p in params]).to(torch.float32)[None].contiguous()
        grid = K.Grid.explicit(coords)
        ctx.fspec, ctx.grid, ctx.shapes = fspec, grid, [p.shape for p in params]
        ctx.save_for_backward(fp)
        return FL.flow_forward(fspec, fp, grid)[0]

    @staticmethod
    def backward(ctx, dout: torch.Tensor):
        (fp,) = ctx.saved_tensors
        g = FL.flow_backward(ctx.fspec, fp, ctx.grid, dout.contiguous()[None])[0]
        outs, off = [], 0
        for shp in ctx.shapes:
            n = math.prod(shp) if len(shp) else 1
            outs.append(g[off:off + n].reshape(shp))
            off += n
        return (None, None, *outs)


class WNLinear(nn.Module):
    """Linear layer with w = g * v / ||v||_F, scalar g (weight_norm(dim=None))."""

    def __init__(self, in_channels: int = 1, out_channels: int = 1, bias: bool = True, **kwargs):
        super().__init__()
        self.linear = nn.utils.weight_norm(nn.Linear(in_channels, out_channels, bias=bias), dim=None)

    def reset_parameters(self, activation: str = "relu") -> None:
        with torch.no_grad():
            self.linear.weight_g.fill_(1)
            std = nn.init.calculate_gain(activation, 0) / math.sqrt(self.linear.weight_v.shape[1])
            nn.init.kaiming_uniform_(self.linear.weight_v, mode="fan_in", nonlinearity=activation)
            if self.linear.bias is not None:
                self.linear.bias.uniform_(-std, std)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.linear(x)


def _apply_uniform(module: nn.Module, activation: str) -> None:
    # weights_init_uniform only touches nn.Linear instances (resnet_1d.py:24-37): under .apply() that is the inner
    # weight-normed nn.Linear, whose `.weight` is the derived tensor - so, as in the reference, only the bias moves.
    for m in module.modules():
        if isinstance(m, nn.Linear):
            with torch.no_grad():
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity=activation)
                if m.bias is not None:
                    std = nn.init.calculate_gain(activation, 0) / math.sqrt(m.weight.shape[1])
                    m.bias.uniform_(-std, std)


class NormalBlock(nn.Module):
    """tanh(WN2(leaky_relu(WN1 x)))  (diffeomorphism_net.py:169-192)."""

    def __init__(self, in_channels: int = 1, mid_channels: int = 128, out_channels: int = 1, **kwargs):
        super().__init__()
        self.in_linear = WNLinear(in_channels, mid_channels, bias=True)
        self.out_linear = WNLinear(mid_channels, out_channels, bias=True)

    def reset_parameters(self) -> None:
        _apply_uniform(self.in_linear, "leaky_relu")
        _apply_uniform(self.out_linear, "tanh")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.tanh(self.out_linear(F.leaky_relu(self.in_linear(x))))


class SimpleBackbone(nn.Module):
    """tanh(WN2(relu(WN1 x)))  (diffeomorphism_net.py:83-104) - NormalizingFlow1D's 'default' backbone."""

    def __init__(self, in_channels: int = 2, network_width: int = 10, **kwargs):
        super().__init__()
        self.linear1 = WNLinear(in_channels, network_width)
        self.linear2 = WNLinear(network_width, in_channels)

    def reset_parameters(self) -> None:
        _apply_uniform(self.linear1, "relu")
        _apply_uniform(self.linear2, "tanh")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.tanh(self.linear2(F.relu(self.linear1(x))))


def _apply_normal(module: nn.Module, activation: str) -> None:
    # weights_init_normal under .apply() (resnet_1d.py:9-22): as with _apply_uniform only the inner nn.Linear is touched; its `.weight`
    # is the tensor weight_norm derives from (g, v) - the normal draws are consumed (seeded construction stays in step with the
    # reference) and overwritten by the next forward; the bias is what moves.
    for m in module.modules():
        if isinstance(m, nn.Linear):
            with torch.no_grad():
                nn.init.kaiming_normal_(m.weight, mode="fan_in", nonlinearity=activation)
                if m.bias is not None:
                    std = nn.init.calculate_gain(activation, 0) / math.sqrt(m.weight.shape[1])
                    m.bias.uniform_(-std, std)


class ResidualBlock1D(nn.Module):
    """x + WN2(relu(BN2(WN1(relu(BN1 x)))))  (real_nvp/resnet_1d.py:66-95); the batch norms use the statistics of the rows they are
    given (track_running_stats=False): the block couples all points of an image."""

    def __init__(self, in_channels: int = 1, out_channels: int = 1, **kwargs):
        super().__init__()
        self.in_norm = nn.BatchNorm1d(in_channels, track_running_stats=False)
        self.in_linear = WNLinear(in_channels, out_channels, bias=False)
        self.out_norm = nn.BatchNorm1d(out_channels, track_running_stats=False)
        self.out_linear = WNLinear(out_channels, out_channels, bias=True)
        _apply_normal(self.in_linear, "relu")
        _apply_normal(self.out_linear, "relu")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = self.in_linear(F.relu(self.in_norm(x)))
        return self.out_linear(F.relu(self.out_norm(y))) + x


class SimpleResnet(nn.Module):
    """NormalizingFlow1D's 'resnet' backbone (diffeomorphism_net.py:107-166): batch-normed input, (x, -x) -> relu -> WN linear, a sum of
    weight-normed skips over `num_blocks` residual blocks, batch norm, relu, WN linear, tanh.  Batch statistics over the points of an
    image make it a whole-image function: it has no per-point kernel and runs as torch operations on the device (forward and
    autograd); no reference config selects it."""

    def __init__(self, in_channels: int = 2, mid_channels: int = 128, out_channels: int = 2, num_blocks: int = 1,
                 double_after_norm: bool = False):
        super().__init__()
        self.in_norm = nn.BatchNorm1d(in_channels, track_running_stats=False)
        self.double_after_norm = double_after_norm
        self.in_linear = WNLinear(2 * in_channels, mid_channels, bias=True)
        self.in_skip = WNLinear(mid_channels, mid_channels, bias=True)
        self.blocks = nn.ModuleList([ResidualBlock1D(mid_channels, mid_channels) for _ in range(num_blocks)])
        self.skips = nn.ModuleList([WNLinear(mid_channels, mid_channels, bias=True) for _ in range(num_blocks)])
        self.out_norm = nn.BatchNorm1d(mid_channels, track_running_stats=False)
        self.out_linear = WNLinear(mid_channels, out_channels, bias=True)
        self._init()

    def _init(self) -> None:
        _apply_normal(self.in_linear, "relu")
        _apply_normal(self.in_skip, "relu")
        _apply_normal(self.skips, "relu")
        _apply_normal(self.out_linear, "tanh")

    def reset_parameters(self) -> None:
        """(The reference class has no reset_parameters - NormalizingFlow1D.reset_parameters raises on this backbone there; here the
        constructor's initialisation is applied again.)"""
        for blk in self.blocks:
            _apply_normal(blk.in_linear, "relu")
            _apply_normal(blk.out_linear, "relu")
        self._init()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.in_norm(x)
        if self.double_after_norm:
            x = x * 2.0
        x = self.in_linear(F.relu(torch.cat((x, -x), dim=1)))
        acc = self.in_skip(x)
        for blk, skip in zip(self.blocks, self.skips):
            x = blk(x)
            acc = acc + skip(x)
        return torch.tanh(self.out_linear(F.relu(self.out_norm(acc))))


class WNScale(nn.Module):
    """Learnable scalar through a weight-normed 1x1 linear (diffeomorphism_net.py:208-232)."""

    def __init__(self, dim: int = 1, **kwargs):
        super().__init__()
        self.scale = nn.utils.weight_norm(nn.Linear(dim, dim))
        self._init_scale()
        self.weight = nn.Parameter(torch.tensor([1.0 + 0.01 * torch.randn((1,))]))

    def _init_scale(self) -> None:
        self.scale.weight.data.normal_(0.0, 1 / np.sqrt(self.scale.in_features))
        self.scale.bias.data.fill_(0)

    def reset_parameters(self) -> None:
        self._init_scale()
        with torch.no_grad():
            self.weight.data = torch.tensor([1.0 + 0.01 * torch.randn((1,))], dtype=self.weight.dtype, device=self.weight.device)

    def forward(self, *args, **kwargs) -> torch.Tensor:
        return self.scale(self.weight)


class NormalizingFlow1D(nn.Module):
    """Alternating affine couplings on the two coordinates (diffeomorphism_net.py:235-302).  Backbones as in the reference:
    'default' = SimpleBackbone (relu; what a bare ConvexDiffeomorphismNet() builds), 'normal_block' / 'residual_block' =
    NormalBlock (leaky_relu; every reference config) - both on the HIP flow kernels.  'resnet' = SimpleResnet (no config uses it): batch
    norms over the points, so the couplings are evaluated as torch operations on the device (with autograd); `_spec` (the fused flow
    kernels) does not exist for it and ConvexDiffeomorphismNet.pretrain runs the reference's loop as device-side autograd."""

    def __init__(self, num_coupling: int = 4, width: int = 130, num_blocks: int = 1, in_features: int = 2,
                 backbone: str = "default", **kwargs):
        super().__init__()
        if num_coupling % in_features != 0:
            raise ValueError(f"Number of coupling layers should be divisible by in_features ({in_features})")
        if backbone not in ("default", "normal_block", "residual_block", "resnet"):
            raise ValueError(f"Unknown backbone: {backbone}")
        self.num_coupling, self.in_features = num_coupling, in_features
        self.backbone = backbone if backbone in ("default", "resnet") else "normal_block"
        if backbone == "default":
            mk = lambda: SimpleBackbone(in_channels=1, network_width=width)  # noqa: E731
        elif backbone == "resnet":
            mk = lambda: SimpleResnet(in_channels=1, mid_channels=width, out_channels=1, num_blocks=num_blocks)  # noqa: E731
        else:
            mk = lambda: NormalBlock(in_channels=1, mid_channels=width, out_channels=1)  # noqa: E731
        self.s = nn.ModuleList([mk() for _ in range(num_coupling)])
        self.t = nn.ModuleList([mk() for _ in range(num_coupling)])
        self.scale = nn.ModuleList([WNScale(dim=1) for _ in range(num_coupling)])

    def reset_parameters(self) -> bool:
        for s, t, sc in zip(self.s, self.t, self.scale):
            s.reset_parameters()
            t.reset_parameters()
            sc.reset_parameters()
        return True

    def _spec(self) -> "FL.FlowSpec":
        if self.backbone == "resnet":
            raise NotImplementedError("the 'resnet' backbone (SimpleResnet: batch norms over the points) has no fused flow kernel")
        first = self.s[0].linear1 if self.backbone == "default" else self.s[0].in_linear
        return FL.FlowSpec(first.linear.weight_v.shape[0], self.num_coupling, self.backbone)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(N, 2) -> (N, 2) on the HIP flow kernels (forward and autograd backward); the kernels' leading 2x2 linear is the
        identity here."""
        if not x.is_cuda:
            raise RuntimeError("awesome_amd modules run on the MI355X only (no CPU fallback); move module and input to cuda")
        if self.in_features != 2 or x.dim() != 2 or x.shape[1] != 2:
            raise ValueError("the coupling flow maps (N, 2) rows (2-D only, like the reference, diffeomorphism_net.py:288)")
        if self.backbone == "resnet":   # whole-image backbones: the couplings as device-side torch operations (:283-300)
            x1, x2 = x[:, :1], x[:, 1:]
            for i in range(self.num_coupling):
                if i % 2 == 0:
                    x2 = torch.exp(self.scale[i]() * self.s[i](x1)) * x2 + self.t[i](x1)
                else:
                    x1 = torch.exp(self.scale[i]() * self.s[i](x2)) * x1 + self.t[i](x2)
            return torch.cat([x1, x2], 1)
        fspec = self._spec()
        sd = dict(self.named_parameters())
        own = [sd[k] for k, _ in fspec.keys_shapes(prefix="") if not k.startswith("linear.")]
        eye = torch.eye(2, device=x.device, dtype=torch.float32)
        out = _FlowFunction.apply(x.t().contiguous(), fspec, eye, torch.zeros(2, device=x.device), *own)
        return out.t()


def translate_linear(weight: torch.Tensor, bias: torch.Tensor, from_points: torch.Tensor, to_points: torch.Tensor):
    """ConvexDiffeomorphismNet.translate (convex_diffeomorphism_net.py:79-128): new (weight, bias) of the input linear layer such
    that `to_points` are mapped where `from_points` were - least squares on the augmented points, same operations."""
    if from_points.shape != to_points.shape:
        raise ValueError("From and to points must have the same shape.")
    if from_points.shape[0] < weight.shape[0]:
        raise ValueError("Not enough points to sample from. Need at least {} points, got {}.".format(weight.shape[0], from_points.shape[0]))
    to_points = to_points.to(dtype=weight.dtype, device=weight.device)
    from_points = from_points.to(dtype=weight.dtype, device=weight.device)
    from_transf = F.linear(from_points, weight, bias)
    X = torch.cat((to_points, torch.ones((to_points.shape[0], 1), device=to_points.device, dtype=to_points.dtype)), dim=1)
    theta = torch.linalg.inv(X.T @ X) @ (X.T @ from_transf)
    return theta[:-1, :].T.contiguous(), theta[-1, :].contiguous()


def translate_only_point_args(from_point: torch.Tensor, to_point: torch.Tensor, grid: torch.Tensor, in_features: int = 2):
    """translate_only_point (:43-77): the pixel (x, y) `from_point` / `to_point` plus two helper points 3 pixels along each axis,
    looked up in the (C, H, W) coordinate grid -> (from_points, to_points) [C + 1, C]."""
    res_from = torch.zeros((in_features + 1, in_features), device=from_point.device, dtype=from_point.dtype)
    res_to = torch.zeros((in_features + 1, in_features), device=to_point.device, dtype=to_point.dtype)
    res_from[0, :], res_to[0, :] = from_point, to_point
    for i in range(in_features):
        v = torch.zeros(in_features, device=from_point.device, dtype=from_point.dtype)
        v[i] += 3
        res_from[i + 1, :] = res_from[0, :] + v
        res_to[i + 1, :] = res_to[0, :] + v
    grid = grid.squeeze()
    pick = lambda pts: grid[..., pts[:, 1].to(dtype=torch.long), pts[:, 0].to(dtype=torch.long)].T  # noqa: E731
    return pick(res_from), pick(res_to)


class ConvexDiffeomorphismNet(nn.Module, PriorFitMixin):
    """ICNN(flow(Ax + b))  (convex_diffeomorphism_net.py:130-188).  `pretrain` / `pretrain_load_state`: PriorFitMixin (the
    reference's :190-490 on `inrfit_cdn_fit`, incl. the centre-of-mass translate of the warm start, :337-348)."""

    # -- the reference's translate helpers (:43-128) -----------------------------------------------------------------------------
    def translate(self, from_points: torch.Tensor, to_points: torch.Tensor) -> None:
        w, b = translate_linear(self.linear.weight.data, self.linear.bias.data, from_points, to_points)
        self.linear.weight.data, self.linear.bias.data = w, b

    def translate_only_point(self, from_point: torch.Tensor, to_point: torch.Tensor, grid: torch.Tensor) -> None:
        self.translate(*translate_only_point_args(from_point, to_point, grid, self.in_features))

    # -- PriorFitMixin engine ------------------------------------------------------------------------------------------------
    def _pretrain_defaults(self):
        return dict(num_epochs=2000, lr=0.003, reuse_state=True, reuse_state_epochs=200, proper_prior_fit_threshold=0.5,
                    proper_prior_fit_retrys=1, weight_decay_on_weight_g=5e-5, weight_decay_on_convex_weight=False)

    # -- the 'resnet' flow backbone: no fused fit; the reference's loop (:376-421) as device-side autograd ---------------------------
    def _generic(self) -> bool:
        """No fused composite for this module: a whole-image flow backbone, or an ICNN shape without a fused kernel (n_hidden > 130 or more
        than two hidden layers: the layer-by-layer path) - forward / backward compose the flow and ICNN kernels through autograd (the
        ICNN's dL/dcoords feeds the flow's backward), `pretrain` runs the reference's loop on that."""
        return self.diffeo_net.backbone == "resnet" or not self.convex_net.spec.fused()

    def _generic_fit(self, grid, unaries, flat, epochs, opts):
        """Adam (weight decay on the weight_g group only) + ReduceLROnPlateau(patience 200, factor 0.5) + enforce_convexity per step,
        one image after the other, each from its row of `flat` (named_parameters order); returns what the fused engines return: the
        fitted rows, the logits of the LAST forward (the IoU gate looks at those, :424) and a status per image."""
        names = [k for k, _ in self.named_parameters()]
        prm = dict(self.named_parameters())
        keep = [p.detach().clone() for p in prm.values()]
        crit = opts.get("criterion")
        lr, wd = float(opts.get("lr", 0.003)), float(opts.get("weight_decay_on_weight_g", 5e-5))
        out_flat, out_logits, status = [], [], []
        for i in range(flat.shape[0]):
            with torch.no_grad():
                off = 0
                for k in names:
                    n = prm[k].numel()
                    prm[k].copy_(flat[i, off:off + n].reshape(prm[k].shape))
                    off += n
            rows = (grid.coords if grid.coords.dim() == 2 else grid.coords[i]).t().contiguous()
            groups = [dict(params=[prm[k] for k in names if k.endswith("weight_g")], weight_decay=wd),
                      dict(params=[prm[k] for k in names if not k.endswith("weight_g")], weight_decay=0.0)]
            opt = torch.optim.Adam(groups, lr=lr)
            sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=200, factor=0.5) if opts.get("use_plateau", True) else None
            tgt, logits, bad = unaries[i].reshape(1, 1, -1), None, 0
            with torch.enable_grad():
                for _ in range(int(epochs)):
                    opt.zero_grad()
                    logits = self(rows)[:, 0]
                    prob = torch.sigmoid(logits).reshape(1, 1, -1)
                    loss = crit(prob, tgt) if crit is not None else F.binary_cross_entropy(prob, tgt)
                    if not bool(torch.isfinite(loss)):
                        bad = 1
                        break
                    loss.backward()
                    opt.step()
                    if sched is not None:
                        sched.step(loss.detach())
                    self.enforce_convexity()
            if logits is None:
                with torch.no_grad():
                    logits = self(rows)[:, 0]
            out_flat.append(torch.cat([prm[k].detach().reshape(-1) for k in names]))
            out_logits.append(logits.detach())
            status.append(bad)
        with torch.no_grad():
            for p_, k_ in zip(prm.values(), keep):
                p_.copy_(k_)
        return torch.stack(out_flat), torch.stack(out_logits), torch.tensor(status, dtype=torch.int32, device=flat.device)

    def _engine_pack(self, sd):
        if self._generic():
            return torch.cat([sd[k].detach().reshape(-1).to(torch.float32).cpu() for k, _ in self.named_parameters()])
        ispec, fspec = self._specs()
        i, f = FL.split_cdn_state_dict(ispec, fspec, sd)
        return torch.cat([i.cpu(), f.cpu()])

    def _engine_unpack(self, flat):
        if self._generic():
            out, off = {}, 0
            for k, p_ in self.named_parameters():
                out[k] = flat[off:off + p_.numel()].reshape(p_.shape).clone()
                off += p_.numel()
            return out
        ispec, fspec = self._specs()
        return FL.merge_cdn_state_dict(ispec, fspec, flat[:ispec.n_params], flat[ispec.n_params:])

    def _engine_fit(self, grid, unaries, flat, epochs, cold, opts, states=None):
        from ..measures import criterion_to_desc
        if opts.get("weight_decay_on_convex_weight", False):
            raise NotImplementedError("weight_decay_on_convex_weight=True (no reference config sets it) has no fused form")
        if self._generic():
            return self._generic_fit(grid, unaries, flat, epochs, opts)
        ispec, fspec = self._specs()
        P = ispec.n_params
        crit = opts.get("criterion")
        kind, wmode, ratio = criterion_to_desc(crit, "targets") if crit is not None else ("bce", "none", 1.0)   # UnariesWeightedLoss(BCELoss, 'none')
        res = FL.cdn_fit(ispec, fspec, flat[:, :P].contiguous(), flat[:, P:].contiguous(), grid, unaries, epochs,
                         lr=float(opts.get("lr", 0.003)), loss=kind, weight_mode=wmode, ratio=ratio,
                         weight_decay_on_weight_g=float(opts.get("weight_decay_on_weight_g", 5e-5)),
                         plateau=dict(patience=200, factor=0.5) if opts.get("use_plateau", True) else None, record_loss=False,
                    want_logits=True, gate_logits=True)
        return torch.cat([res.icnn_params, res.flow_params], 1), res.logits, res.status

    def _engine_warm_start(self, flat, ctx, image, opts):
        """:337-348: shift the previous frame's prior to this frame's centre of mass before the short refit."""
        com = center_of_mass(image.unaries)
        if ctx is not None and self._generic():   # the same translate on this layout's `linear.weight` / `linear.bias` slices
            off, where = 0, {}
            for k, p_ in self.named_parameters():
                where[k] = (off, p_.numel())
                off += p_.numel()
            (ow, nw), (ob, nb) = where["linear.weight"], where["linear.bias"]
            pts = translate_only_point_args(ctx.flip(dims=(-1,)), com.flip(dims=(-1,)), image.grid.squeeze(), self.in_features)
            w, b = translate_linear(flat[ow:ow + nw].reshape(2, 2).clone(), flat[ob:ob + nb].clone(), *pts)
            flat[ow:ow + nw], flat[ob:ob + nb] = w.reshape(-1), b
            return flat, com
        if ctx is not None:
            P = self._specs()[0].n_params
            w, b = flat[P:P + 4].reshape(2, 2).clone(), flat[P + 4:P + 6].clone()
            pts = translate_only_point_args(ctx.flip(dims=(-1,)), com.flip(dims=(-1,)), image.grid.squeeze(), self.in_features)
            w, b = translate_linear(w, b, *pts)
            flat[P:P + 4], flat[P + 4:P + 6] = w.reshape(-1), b
        return flat, com

    def _engine_chain_context(self, ctx, image):
        return center_of_mass(image.unaries) if ctx is None else ctx

    def __init__(self, n_hidden: int = 130, n_hidden_layers: int = 1, nf_layers: int = 4, nf_hidden: int = 70,
                 in_features: int = 2, diffeo_args: Optional[Dict[str, Any]] = None, **kwargs):
        super().__init__()
        self.convex_net = ConvexNextNet(n_hidden=n_hidden, in_features=in_features, n_hidden_layers=n_hidden_layers)
        diffeo_args = dict(diffeo_args or {})
        diffeo_args.setdefault("num_coupling", nf_layers)
        diffeo_args.setdefault("width", nf_hidden)
        diffeo_args.setdefault("in_features", in_features)
        self.in_features = in_features
        self.diffeo_net = NormalizingFlow1D(**diffeo_args)
        self.linear = nn.Linear(in_features, in_features)
        self._init_linear()

    def _init_linear(self) -> None:
        self.linear.weight.data.normal_(0.0, 1 / np.sqrt(self.linear.in_features))
        self.linear.bias.data.fill_(0)

    def reset_parameters(self) -> bool:
        self.convex_net.reset_parameters()
        self.diffeo_net.reset_parameters()
        self._init_linear()
        return True

    def get_deformation(self, x: torch.Tensor) -> torch.Tensor:
        """(B,C,H,W) or (N,C) -> deformed coordinates flow(Ax + b) in the same layout (convex_diffeomorphism_net.py:179-184), on
        `inrfit_flow_forward`; no autograd (it is an inspection output)."""
        if not x.is_cuda:
            raise RuntimeError("awesome_amd modules run on the MI355X only (no CPU fallback); move module and input to cuda")
        if self.diffeo_net.backbone == "resnet":
            with torch.no_grad():
                if x.dim() == 4:
                    b, c, h, w = x.shape
                    rows = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
                    return torch.stack([self.diffeo_net(self.linear(rows[i])) for i in range(b)], 0).reshape(b, h, w, c).permute(0, 3, 1, 2)
                return self.diffeo_net(self.linear(x))
        _, fspec, _, flow = self._ordered_params()
        fp = torch.cat([p.detach().reshape(-1) for p in flow]).to(torch.float32)[None].contiguous()
        if x.dim() == 4:
            b, c, h, w = x.shape
            outs = [FL.flow_forward(fspec, fp, K.Grid.explicit(x[i].reshape(c, h * w)))[0].reshape(c, h, w) for i in range(b)]
            return torch.stack(outs, 0)
        return FL.flow_forward(fspec, fp, K.Grid.explicit(x.t().contiguous()))[0].t()

    def _specs(self):
        nf = self.diffeo_net
        return self.convex_net.spec, self.diffeo_net._spec()

    def _ordered_params(self):
        ispec, fspec = self._specs()
        sd = dict(self.named_parameters())
        icnn = [sd["convex_net." + k] for k, _ in ispec.keys_shapes()]
        flow = [sd[k] for k, _ in fspec.keys_shapes()]
        return ispec, fspec, icnn, flow

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B,2,H,W) -> (B,1,H,W) or (N,2) -> (N,1), forward and backward on the HIP path."""
        if not x.is_cuda:
            raise RuntimeError("awesome_amd modules run on the MI355X only (no CPU fallback); move module and input to cuda")
        if self.in_features != 2:
            raise ValueError("the coupling flow is 2-D only (like the reference, diffeomorphism_net.py:288)")
        if self.diffeo_net.backbone == "resnet":
            # no fused composite for the whole-image backbone: deformation in torch on the device, the ICNN on its kernels (forward,
            # parameter gradients and dL/dcoords for the flow's backward: convex_net._IcnnFunction)
            if x.dim() == 4:
                b, c, h, w = x.shape
                rows = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
                return torch.stack([self.convex_net(self.diffeo_net(self.linear(rows[i]))).reshape(1, h, w) for i in range(b)], 0)
            return self.convex_net(self.diffeo_net(self.linear(x)))
        ispec, fspec, icnn, flow = self._ordered_params()
        if not ispec.fused():   # flow kernels, then the layer-by-layer ICNN: two autograd bridges instead of the fused composite
            from .convex_net import _IcnnFunction
            run = lambda coords: _IcnnFunction.apply(_FlowFunction.apply(coords, fspec, *flow), ispec, *icnn)  # noqa: E731
        else:
            run = lambda coords: _CdnFunction.apply(coords, ispec, fspec, len(icnn), *icnn, *flow)  # noqa: E731
        if x.dim() == 4:
            b, c, h, w = x.shape
            return torch.stack([run(x[i].reshape(c, h * w)).reshape(1, h, w) for i in range(b)], 0)
        return run(x.t().contiguous())[:, None]

    def enforce_convexity(self) -> None:
        self.convex_net.enforce_convexity()

    def fit_images(self, grid: "K.Grid", unaries: torch.Tensor, num_epochs: int = 2000, lr: float = 3e-3, loss: str = "bce",
                   weight_decay_on_weight_g: float = 5e-5, plateau=None, init_from_self: bool = True):
        """Fused device-resident form of `pretrain`'s inner loop (convex_diffeomorphism_net.py:405-430) for a batch of
        images: every image starts from this module's current parameters; returns the CdnFitResult (flat parameters per
        image; `FL.merge_cdn_state_dict` turns a row back into a state_dict for the PriorCache)."""
        ispec, fspec, icnn, flow = self._ordered_params()
        n, dev = unaries.shape[0], unaries.device
        ip = torch.cat([p.detach().reshape(-1) for p in icnn]).to(torch.float32)[None].repeat(n, 1).contiguous().to(dev)
        fp = torch.cat([p.detach().reshape(-1) for p in flow]).to(torch.float32)[None].repeat(n, 1).contiguous().to(dev)
        return FL.cdn_fit(ispec, fspec, ip, fp, grid, unaries, num_epochs, lr=lr, loss=loss,
                          weight_decay_on_weight_g=weight_decay_on_weight_g,
                          plateau=dict(patience=200, factor=0.5) if plateau is None else (plateau or None))
