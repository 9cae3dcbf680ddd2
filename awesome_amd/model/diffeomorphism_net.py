"""Weight-normalised coupling flow + the path-connected prior ICNN(flow(Ax + b)) with the reference's module surface.

Reference: WNLinear (awesome/model/real_nvp/resnet_1d.py:39-63), NormalBlock / WNScale / NormalizingFlow1D
(awesome/model/diffeomorphism_net.py:169-302), ConvexDiffeomorphismNet (awesome/model/convex_diffeomorphism_net.py:41-188).
Same constructor kwargs and state_dict keys (torch's legacy weight_norm parametrisation: `weight_g` / `weight_v`), same
creation order of the leaves (seeded construction = reference init).

Round-1 status: the ICNN stage runs on the HIP path (forward, parameter gradients and the coordinate gradient that flows
back into the deformation); the coupling layers themselves are still plain torch ops on the GPU - a fused HIP flow stage
is the next row of SURVEY.md §8 (a6-a8)."""
from __future__ import annotations

import math
from typing import Any, Dict, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .convex_net import ConvexNextNet, _kaiming_uniform_reset


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

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.convex_net(self.get_deformation(x))

    def enforce_convexity(self) -> None:
        self.convex_net.enforce_convexity()
