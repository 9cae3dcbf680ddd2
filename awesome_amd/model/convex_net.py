"""Drop-in prior modules for `prior_model_type: awesome_amd.model.ConvexNet | ConvexNextNet`.

Same constructor kwargs, state_dict keys, `enforce_convexity()` / `reset_parameters()` contract as the reference
classes (awesome/model/convex_net.py:10-40, 134-220) so PriorCache round-trips and YAML configs keep working
(SURVEY.md §8b); the arithmetic is the HIP C-ABI library - forward is `inrfit_forward`, backward is `inrfit_backward`.
Parameters are ordinary nn.Linear leaves created in the reference's order, so a seeded construction yields the same
initial weights as the reference module."""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn as nn

from .. import icnn as K
from .pretrainable_module import PriorFitMixin


class _IcnnFunction(torch.autograd.Function):
    """logits = f_theta(coords); backward = vector-Jacobian product w.r.t. theta on the device."""

    @staticmethod
    def forward(ctx, coords: torch.Tensor, spec: K.IcnnSpec, *params: torch.Tensor):
        flat = torch.cat([p.reshape(-1) for p in params]).to(torch.float32)[None].contiguous()
        grid = K.Grid.explicit(coords)
        ctx.spec, ctx.grid, ctx.shapes = spec, grid, [p.shape for p in params]
        ctx.save_for_backward(flat)
        return K.forward(spec, flat, grid)[0]

    @staticmethod
    def backward(ctx, dlogits: torch.Tensor):
        (flat,) = ctx.saved_tensors
        dco = None
        if ctx.needs_input_grad[0]:   # the ICNN sits behind a learned deformation: also return dL/dcoords (C, N)
            g, dco = K.backward(ctx.spec, flat, ctx.grid, dlogits.contiguous()[None], want_dcoords=True)
            g, dco = g[0], dco[0]
        else:
            g = K.backward(ctx.spec, flat, ctx.grid, dlogits.contiguous()[None])[0]
        outs, off = [], 0
        for shp in ctx.shapes:
            n = math.prod(shp)
            outs.append(g[off:off + n].reshape(shp))
            off += n
        return (dco, None, *outs)


def _kaiming_uniform_reset(linear: nn.Linear, activation: str) -> None:
    """weights_init_uniform (awesome/model/real_nvp/resnet_1d.py:24-37)."""
    with torch.no_grad():
        nn.init.kaiming_uniform_(linear.weight, mode="fan_in", nonlinearity=activation)
        if linear.bias is not None:
            std = nn.init.calculate_gain(activation, 0) / math.sqrt(linear.weight.shape[1])
            linear.bias.uniform_(-std, std)


class _Block(nn.Module):
    """ln: hidden -> out (clamped >= 0), skp: input -> out (free).  Mirrors SkipBlock/OutBlock (convex_net.py:134-175)."""

    def __init__(self, in_features: int, out_features: int, in_skip_features: int, activation: str):
        super().__init__()
        self.ln = nn.Linear(in_features, out_features)
        self.skp = nn.Linear(in_skip_features, out_features, bias=False)
        self._activation = activation

    def reset_parameters(self) -> None:
        _kaiming_uniform_reset(self.ln, self._activation)
        _kaiming_uniform_reset(self.skp, self._activation)

    def enforce_convexity(self) -> None:
        with torch.no_grad():
            self.ln.weight.clamp_(min=0.0)


class _IcnnModule(nn.Module, PriorFitMixin):
    """Common part of the ICNN drop-ins.  `pretrain(...)` (PriorFitMixin) is an extension: the reference's ConvexNet /
    ConvexNextNet are not PretrainableModules (their configs train through TorchAgent._perform_step); here the same
    `_prior_based_pretrain` loop semantics (path_connected_net.py:730-1007: Adamax lr 1e-3, UnariesWeightedLoss(SE),
    ReduceLROnPlateau(200, 0.5), IoU gate + retry, reuse_state) run on `inrfit_fit`, so one entry point serves every prior."""
    spec: K.IcnnSpec

    # -- PriorFitMixin engine -----------------------------------------------------------------------------------------------
    def _pretrain_defaults(self):
        return dict(num_epochs=2000, lr=1e-3, optimizer="adamax", weight_decay=0.0, reuse_state=True, reuse_state_epochs=200,
                    proper_prior_fit_threshold=0.5, proper_prior_fit_retrys=1)

    def _engine_pack(self, sd):
        return K.pack_state_dict(self.spec, sd)

    def _engine_unpack(self, flat):
        return K.unpack_params(self.spec, flat, convexnet_keys=hasattr(self, "W0y"))

    def _engine_fit(self, grid, unaries, flat, epochs, cold, opts, states=None):
        from ..measures import criterion_to_desc
        crit = opts.get("criterion")
        kind, wmode, ratio = criterion_to_desc(crit, "targets") if crit is not None else ("se", "none", 1.0)
        res = K.fit(self.spec, flat, grid, unaries, epochs, lr=float(opts.get("lr", 1e-3)), loss=kind, weight_mode=wmode, ratio=ratio,
                    optimizer=opts.get("optimizer", "adamax"), weight_decay=float(opts.get("weight_decay", 0.0)),
                    plateau=dict(patience=200, factor=0.5) if opts.get("use_plateau", True) else None, record_loss=False,
                    want_logits=True, gate_logits=True,
                    **dict(getattr(self, "fit_options", None) or dict(clamp=True)))
        return res.params, res.logits, res.status

    def _ordered_params(self) -> List[torch.Tensor]:
        sd = dict(self.named_parameters())
        names = [k for k, _ in self.spec.keys_shapes()]
        if "W0y.weight" in sd:
            names = [K.CONVEXNET_KEYMAP_INV[k] for k in names]
        return [sd[k] for k in names]

    def flat_parameters(self) -> torch.Tensor:
        """Current parameters as the C ABI's flat vector [P] (detached copy)."""
        return torch.cat([p.detach().reshape(-1) for p in self._ordered_params()]).to(torch.float32)

    def load_flat_parameters(self, flat: torch.Tensor) -> None:
        off = 0
        with torch.no_grad():
            for p in self._ordered_params():
                n = p.numel()
                p.copy_(flat[off:off + n].reshape(p.shape))
                off += n

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B,C,H,W) -> (B,1,H,W)  or  (N,C) -> (N,1); the @pixelize contract of awesome/util/pixelize.py:7-53."""
        if not x.is_cuda:
            raise RuntimeError("awesome_amd modules run on the MI355X only (no CPU fallback); move module and input to cuda")
        params = self._ordered_params()
        if x.dim() == 4:
            b, c, h, w = x.shape
            outs = [_IcnnFunction.apply(x[i].reshape(c, h * w), self.spec, *params).reshape(1, h, w) for i in range(b)]
            return torch.stack(outs, 0)
        if x.dim() == 2:
            return _IcnnFunction.apply(x.t().contiguous(), self.spec, *params)[:, None]
        raise ValueError(f"expected (B,C,H,W) or (N,C), got {tuple(x.shape)}")


class ConvexNextNet(_IcnnModule):
    """awesome/model/convex_net.py:177-220."""

    def __init__(self, n_hidden: int = 130, in_features: int = 2, out_features: int = 1, n_hidden_layers: int = 1, **kwargs):
        super().__init__()
        if out_features != 1:
            raise ValueError("the HIP path implements the scalar-output ICNN (out_features=1)")
        self.spec = K.IcnnSpec(n_hidden, in_features, n_hidden_layers)
        self.input = nn.Linear(in_features, n_hidden)
        self.skip = nn.ModuleList([_Block(n_hidden, n_hidden, in_features, "relu") for _ in range(n_hidden_layers)])
        self.out = _Block(n_hidden, out_features, in_features, "linear")

    def reset_parameters(self) -> bool:
        _kaiming_uniform_reset(self.input, "linear")
        for blk in self.skip:
            blk.reset_parameters()
        self.out.reset_parameters()
        return True  # "children handled" (awesome/util/torch.py:160-194)

    def enforce_convexity(self) -> None:
        for blk in self.skip:
            blk.enforce_convexity()
        self.out.enforce_convexity()


class ConvexNet(_IcnnModule):
    """awesome/model/convex_net.py:10-40 (same network as ConvexNextNet(L=1), different key names)."""

    def __init__(self, n_hidden: int = 130, in_channels: int = 2, **kwargs):
        super().__init__()
        self.spec = K.IcnnSpec(n_hidden, in_channels, 1)
        self.W0y = nn.Linear(in_channels, n_hidden)
        self.W1z = nn.Linear(n_hidden, n_hidden)
        self.W2z = nn.Linear(n_hidden, 1)
        self.W1y = nn.Linear(in_channels, n_hidden, bias=False)
        self.W2y = nn.Linear(in_channels, 1, bias=False)

    def enforce_convexity(self) -> None:
        with torch.no_grad():
            self.W1z.weight.clamp_(min=0.0)
            self.W2z.weight.clamp_(min=0.0)
