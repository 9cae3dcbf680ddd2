"""FCNet as a coordinate network: the "no prior" model of BASELINE configs[0].

Reference: FCNet (awesome/model/fc_net.py:10-59) = Linear(in_chn, width), ReLU, depth x [Linear(width, width), ReLU], Linear(width,
out_chn), fed by concat_input(in_type, image, grid) (awesome/model/cnn_net.py).  With in_type='xy' its input is the coordinate
grid alone and it is the ICNN of awesome_amd/csrc without skip connections and without the convexity clamp: it runs on the same
HIP kernels with the skip weights held at zero (`InrOptDesc.freeze_skips`) and `clamp = 0`.  Same `model.*` state_dict keys and
layer creation order as the reference (a seeded construction yields the reference's initial weights)."""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn as nn

from .. import icnn as K
from .convex_net import _IcnnFunction, _IcnnModule
from .pretrainable_module import PriorFitMixin


def linear_relu(width: int) -> nn.Sequential:
    return nn.Sequential(nn.Linear(width, width), nn.ReLU())


class FCNet(nn.Module, PriorFitMixin):
    #: options the fitters pass to `awesome_amd.fit` for this model (BatchedPriorFitter reads them)
    fit_options = dict(clamp=False, freeze_skips=True)

    # -- PriorFitMixin engine (the same per-image loop as the ICNN priors, on this module's own key layout) -------------------
    _pretrain_defaults = _IcnnModule._pretrain_defaults
    _engine_fit = _IcnnModule._engine_fit

    def _engine_pack(self, sd):
        keep = {k: v.detach().clone() for k, v in self.state_dict().items()}
        self.load_state_dict({k: v.to(self.model[0].weight.device) for k, v in sd.items()})
        flat = self.flat_parameters().cpu()
        self.load_state_dict(keep)
        return flat

    def _engine_unpack(self, flat):
        return self.unpack_flat(flat)

    def reset_parameters(self) -> None:
        for lin in self._linears():
            lin.reset_parameters()

    def enforce_convexity(self) -> None:   # nothing is constrained in the "no prior" network
        return None

    def __init__(self, in_chn: int = 2, out_chn: int = 1, width: int = 130, depth: int = 1, in_type: str = "xy", **kwargs):
        super().__init__()
        if in_type != "xy":
            raise ValueError("the HIP path implements the coordinate network (in_type='xy'); image inputs belong to the "
                             "segmentation backbones, which are out of scope (SURVEY.md §8)")
        if out_chn != 1:
            raise ValueError("scalar output only (out_chn=1)")
        self.in_chn, self.out_chn, self.in_type, self.depth = in_chn, out_chn, in_type, depth
        self.spec = K.IcnnSpec(width, in_chn, depth)
        self.model = nn.Sequential(nn.Linear(in_chn, width), nn.ReLU(), *[linear_relu(width) for _ in range(depth)],
                                   nn.Linear(width, out_chn))

    # ---- flat views: the ICNN layout with zero skip weights ---------------------------------------------------------------
    def _linears(self) -> List[nn.Linear]:
        return [self.model[0]] + [self.model[2 + k][0] for k in range(self.depth)] + [self.model[2 + self.depth]]

    def flat_parameters(self) -> torch.Tensor:
        lin = self._linears()
        h, c = self.spec.n_hidden, self.spec.in_features
        dev = lin[0].weight.device
        parts = [lin[0].weight, lin[0].bias]
        for k in range(self.depth):
            parts += [lin[1 + k].weight, lin[1 + k].bias, torch.zeros(h, c, device=dev)]
        parts += [lin[-1].weight, lin[-1].bias, torch.zeros(1, c, device=dev)]
        return torch.cat([p.detach().reshape(-1) for p in parts]).to(torch.float32)

    def unpack_flat(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        """One row of a fit result -> this module's state_dict keys (the PriorCache entry)."""
        sd = K.unpack_params(self.spec, flat)
        out = {"model.0.weight": sd["input.weight"], "model.0.bias": sd["input.bias"]}
        for k in range(self.depth):
            out[f"model.{2 + k}.0.weight"], out[f"model.{2 + k}.0.bias"] = sd[f"skip.{k}.ln.weight"], sd[f"skip.{k}.ln.bias"]
        out[f"model.{2 + self.depth}.weight"], out[f"model.{2 + self.depth}.bias"] = sd["out.ln.weight"], sd["out.ln.bias"]
        return out

    def load_flat_parameters(self, flat: torch.Tensor) -> None:
        self.load_state_dict({k: v.to(self.model[0].weight.device) for k, v in self.unpack_flat(flat).items()})

    def forward(self, image: torch.Tensor, grid: torch.Tensor = None, *args, **kwargs) -> torch.Tensor:
        """(image, grid) like the reference; only the grid is used (in_type='xy'): (B,C,H,W) -> (B,1,H,W) or (N,C) -> (N,1)."""
        x = grid if grid is not None else image
        if not x.is_cuda:
            raise RuntimeError("awesome_amd modules run on the MI355X only (no CPU fallback); move module and input to cuda")
        lin = self._linears()
        h, c = self.spec.n_hidden, self.spec.in_features
        z = lambda *shape: torch.zeros(*shape, device=x.device)  # noqa: E731  (skip weights: constants, no gradient)
        params = [lin[0].weight, lin[0].bias]
        for k in range(self.depth):
            params += [lin[1 + k].weight, lin[1 + k].bias, z(h, c)]
        params += [lin[-1].weight, lin[-1].bias, z(1, c)]
        if x.dim() == 4:
            b, cc, hh, ww = x.shape
            return torch.stack([_IcnnFunction.apply(x[i].reshape(cc, hh * ww), self.spec, *params).reshape(1, hh, ww)
                                for i in range(b)], 0)
        return _IcnnFunction.apply(x.t().contiguous(), self.spec, *params)[:, None]
