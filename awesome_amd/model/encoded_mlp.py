"""Coordinate MLPs behind a periodic ENCODE stage (SURVEY.md §0 "Fourier/SIREN positional encoding -> small MLP").

The `awesome/` package has no such model; its two instances live in notebooks, and these classes follow them:

  FourierFeatureNet   `ourSimpleNetwork` (notebooks/imageRepresentationTest.ipynb cell 5):
                          x -> cos(x @ A + b)   with fixed buffers A = factor * randn(d_in, d_features), b = randn(d_features)
                            -> relu(fc1) -> relu(fc2) [-> relu(fc3)] -> fc_out     (the notebook ends in a sigmoid; here, as for
                          every model of this package, the sigmoid belongs to the wrapper / the data term)
  SineLayerNet        `myNet` (notebooks/icml_teaser_code/repeating/repeating.ipynb cell 3):
                          x -> sin(10 pi * W1(x + offset))  with a learnable W1 and a fixed offset  -> ... -> W_out

On the HIP path the encode IS layer 0 of the fused step kernels with another activation (include/inrfit.h INR_ACT_COS /
INR_ACT_SIN): the features are computed in registers from the coordinates, exactly where relu(W_in x + b_in) is otherwise, so the
network is  encode (n_hidden features) -> L relu layers of n_hidden units -> scalar output  with L = n_hidden_layers in {1, 2}
(the kernels' shapes: the notebook's 3 x 350 relu layers on 20 features become 1-2 x 130 on 130 features; its sine net's direct
read-out gets one relu layer in between).  No skip connections, no convexity clamp (`fit_options`); the Fourier features stay
fixed (`freeze_input`).  `pretrain`, `fit` and autograd work as for every other prior module (PriorFitMixin, _IcnnFunction)."""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn as nn

from .. import icnn as K
from .convex_net import _IcnnFunction, _IcnnModule


class _EncodedMLP(_IcnnModule):
    """Shared plumbing: the flat ICNN parameter vector with zero skip weights; subclasses provide layer 0."""

    def _layer0(self):
        raise NotImplementedError

    def _hidden(self) -> List[nn.Linear]:
        return [getattr(self, f"fc{k + 1}") for k in range(self.spec.n_layers)]

    def _param_list(self, detach: bool):
        w0, b0 = self._layer0()
        h, c = self.spec.n_hidden, self.spec.in_features
        dev = self.out.weight.device
        z = lambda *shape: torch.zeros(*shape, device=dev)  # noqa: E731  (skip weights: constants of value 0)
        ps = [w0, b0]
        for lin in self._hidden():
            ps += [lin.weight, lin.bias, z(h, c)]
        ps += [self.out.weight, self.out.bias, z(1, c)]
        return [p.detach() for p in ps] if detach else ps

    def flat_parameters(self) -> torch.Tensor:
        return torch.cat([p.reshape(-1) for p in self._param_list(True)]).to(torch.float32)

    def forward(self, x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        """(B,C,H,W) -> (B,1,H,W) or (N,C) -> (N,1) logits, forward and autograd backward on the HIP path."""
        if not x.is_cuda:
            raise RuntimeError("awesome_amd modules run on the MI355X only (no CPU fallback); move module and input to cuda")
        params = self._param_list(False)
        if x.dim() == 4:
            b, c, h, w = x.shape
            return torch.stack([_IcnnFunction.apply(x[i].reshape(c, h * w), self.spec, *params).reshape(1, h, w) for i in range(b)], 0)
        return _IcnnFunction.apply(x.t().contiguous(), self.spec, *params)[:, None]

    def enforce_convexity(self) -> None:   # the prior-module contract; nothing is constrained here
        return None

    def reset_parameters(self) -> None:
        """New draw of every linear layer (the retry of a failed fit, PriorFitMixin._engine_fresh_state); fixed features and poses stay."""
        for m in self.children():
            if isinstance(m, nn.Linear):
                m.reset_parameters()

    # -- PriorFitMixin engine: state_dict <-> flat -----------------------------------------------------------------------------
    def _engine_pack(self, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
        keep = {k: v.detach().clone() for k, v in self.state_dict().items()}
        self.load_state_dict({k: v.to(self.out.weight.device) for k, v in sd.items()})
        flat = self.flat_parameters().cpu()
        self.load_state_dict(keep)
        return flat

    def _engine_unpack(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        sd = K.unpack_params(self.spec, flat)
        out = self._unpack_layer0(sd)
        for k in range(self.spec.n_layers):
            out[f"fc{k + 1}.weight"], out[f"fc{k + 1}.bias"] = sd[f"skip.{k}.ln.weight"], sd[f"skip.{k}.ln.bias"]
        out["out.weight"], out["out.bias"] = sd["out.ln.weight"], sd["out.ln.bias"]
        return out

    unpack_flat = _engine_unpack   # the name the fitters use for modules with their own key layout (like FCNet)


class FourierFeatureNet(_EncodedMLP):
    #: `awesome_amd.fit` options for this model: unconstrained MLP, fixed features
    fit_options = dict(clamp=False, freeze_skips=True, freeze_input=True)

    def __init__(self, d_in: int = 2, n_hidden: int = 130, n_hidden_layers: int = 1, factor: float = 30.0, **kwargs):
        super().__init__()
        self.spec = K.IcnnSpec(n_hidden, d_in, n_hidden_layers, act0="cos")
        # same creation order and expressions as the notebook's constructor (buffers first, then the linear layers)
        self.register_buffer("A", factor * torch.randn(d_in, n_hidden))
        self.register_buffer("b", torch.randn(n_hidden))
        for k in range(n_hidden_layers):
            setattr(self, f"fc{k + 1}", nn.Linear(n_hidden, n_hidden))
        self.out = nn.Linear(n_hidden, 1)

    def _layer0(self):
        return self.A.t().contiguous(), self.b          # cos(x @ A + b) = cos(W_in x + b_in) with W_in = A^T

    def _unpack_layer0(self, sd):
        return {"A": sd["input.weight"].t().contiguous(), "b": sd["input.bias"]}


class SineLayerNet(_EncodedMLP):
    fit_options = dict(clamp=False, freeze_skips=True)

    def __init__(self, in_features: int = 2, n_hidden: int = 130, n_hidden_layers: int = 1, omega: float = 10 * 3.141592, **kwargs):
        super().__init__()
        self.spec = K.IcnnSpec(n_hidden, in_features, n_hidden_layers, act0="sin", omega=float(omega))   # 10*3.141592 as in the cell
        self.offset = nn.Parameter(torch.zeros(1, in_features), requires_grad=False)
        self.W1 = nn.Linear(in_features, n_hidden)
        for k in range(n_hidden_layers):
            setattr(self, f"fc{k + 1}", nn.Linear(n_hidden, n_hidden))
        self.out = nn.Linear(n_hidden, 1)

    def _layer0(self):
        # W1(x + offset) = W1 x + (W1 offset + b1): the fixed offset folds into the bias the kernel sees
        return self.W1.weight, self.W1.bias + (self.offset @ self.W1.weight.t()).reshape(-1)

    def _unpack_layer0(self, sd):
        w = sd["input.weight"]
        off = self.offset.detach().cpu()
        return {"offset": off.clone(), "W1.weight": w, "W1.bias": sd["input.bias"] - (off @ w.t()).reshape(-1)}
