"""Weight-normalised coupling flow + the path-connected prior ICNN(flow(Ax + b)) with the reference's module surface.

Reference: WNLinear (awesome/model/real_nvp/resnet_1d.py:39-63), NormalBlock / WNScale / NormalizingFlow1D
(awesome/model/diffeomorphism_net.py:169-302), ConvexDiffeomorphismNet (awesome/model/convex_diffeomorphism_net.py:41-188).
Same constructor kwargs and state_dict keys (torch's legacy weight_norm parametrisation: `weight_g` / `weight_v`), same
creation order of the leaves (seeded construction = reference init).

`ConvexDiffeomorphismNet.forward` and its autograd backward run entirely on the HIP path (flow kernels + ICNN kernels,
`inrfit_cdn_forward` / `inrfit_cdn_loss_grad` with an external dL/dlogits); `fit_images` is the fused device-resident
form of `ConvexDiffeomorphismNet.pretrain`'s inner loop (`inrfit_cdn_fit`).  The stand-alone flow modules
(`NormalizingFlow1D`, `NormalBlock`, ...) are torch-op modules (they are the containers of the parameters)."""
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
    """Alternating affine couplings on the two coordinates (diffeomorphism_net.py:235-302), `normal_block` backbone."""

    def __init__(self, num_coupling: int = 4, width: int = 130, num_blocks: int = 1, in_features: int = 2,
                 backbone: str = "normal_block", **kwargs):
        super().__init__()
        if num_coupling % in_features != 0:
            raise ValueError(f"Number of coupling layers should be divisible by in_features ({in_features})")
        if backbone not in ("normal_block", "residual_block"):
            raise ValueError("only the normal_block backbone (all path-connectedness configs) is implemented")
        self.num_coupling, self.in_features = num_coupling, in_features
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

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x1, x2 = x[:, :1], x[:, 1:]
        for i in range(self.num_coupling):
            if i % 2 == 0:
                x2 = torch.exp(self.scale[i]() * self.s[i](x1)) * x2 + self.t[i](x1)
            else:
                x1 = torch.exp(self.scale[i]() * self.s[i](x2)) * x1 + self.t[i](x2)
        return torch.cat([x1, x2], 1)


class ConvexDiffeomorphismNet(nn.Module):
    """ICNN(flow(Ax + b))  (convex_diffeomorphism_net.py:130-188)."""

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
        """(B,C,H,W) or (N,C) -> deformed coordinates in the same layout."""
        if x.dim() == 4:
            b, c, h, w = x.shape
            rows = x.permute(0, 2, 3, 1).reshape(-1, c)
            return self.diffeo_net(self.linear(rows)).reshape(b, h, w, c).permute(0, 3, 1, 2)
        return self.diffeo_net(self.linear(x))

    def _specs(self):
        nf = self.diffeo_net
        return self.convex_net.spec, FL.FlowSpec(nf.s[0].in_linear.linear.weight_v.shape[0], nf.num_coupling)

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
        ispec, fspec, icnn, flow = self._ordered_params()
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
