"""Rotational / mirror-symmetry prior of the teaser (notebooks/icml_teaser_code/rotation_symmetric).

`RotationSymmetricNet` follows `myNet` of rotation_symmetric.ipynb cell 2:

    x  <- x + offset                          learnable centre, (1, 2)
    r  <- |x|,  x <- x / (0.001 + r)          polar split: direction and radius
    x  <- R(orientation) x                    learnable axis
    x  <- (x_0, |x_1|)                        the prior: mirror symmetry about the axis (when `symmetry_prior`)
    logits <- W2 relu(W1 relu(W0 [x, r]))

The pose (`offset`, `orientation`: three scalars) is a handful of elementwise device operations on the (N, 2) coordinates and stays
in torch so that autograd differentiates it; the network behind it - 3 -> h -> h -> 1, all of the arithmetic - is the fused HIP step
kernel with C = 3 input features, no skips, no clamp (the FCNet shape).  Its input gradient (`inrfit_backward`, dcoords) is what
reaches the pose.  Two ways to train, both on the MI355X path:

  * `forward` + autograd + any torch optimizer over `parameters()` (the notebook's loop, pose included);
  * `fit(coords, targets, steps)`: the fused `inrfit_fit` of the network under the current, fixed pose - and
    `fit_alternating`, which interleaves it with pose steps through autograd.

Only the kernels' widths are available (n_hidden in {32, 64, 130}; the notebook used 150).  The star-shape prior of the same folder
is awesome_amd/model/star_net.py."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import icnn as K
from .convex_net import _IcnnFunction
from .encoded_mlp import _EncodedMLP


def polar_symmetry_features(x: torch.Tensor, offset: torch.Tensor, orientation: torch.Tensor, symmetry_prior: bool = True) -> torch.Tensor:
    """(N, 2) coordinates -> (N, 3) features [direction (rotated, mirrored), radius]; the first half of cell 2's `forward`."""
    x = x + offset
    r = torch.sqrt(torch.sum(x ** 2, dim=1, keepdim=True))
    x = x / (0.001 + r)
    c, s = torch.cos(orientation), torch.sin(orientation)
    x = torch.cat(((x[:, 0] * c - x[:, 1] * s)[:, None], (x[:, 0] * s + x[:, 1] * c)[:, None]), dim=1)
    if symmetry_prior:
        x = torch.cat((x[:, 0][:, None], torch.abs(x[:, 1])[:, None]), dim=1)
    return torch.cat((x, r), dim=1)


class RotationSymmetricNet(_EncodedMLP):
    #: `awesome_amd.fit` options of the network behind the pose: unconstrained MLP
    fit_options = dict(clamp=False, freeze_skips=True)

    def __init__(self, n_hidden: int = 130, symmetry_prior: bool = True, **kwargs):
        super().__init__()
        self.spec = K.IcnnSpec(n_hidden, 3, 1)
        self.symmetry_prior = bool(symmetry_prior)
        # the notebook's names and creation order (a seeded construction gives its initial weights)
        self.offset = nn.Parameter(torch.zeros(1, 2))
        self.orientation = nn.Parameter(-0.05 * torch.ones(1))
        self.W0 = nn.Linear(3, n_hidden)
        self.W1 = nn.Linear(n_hidden, n_hidden)
        self.W2 = nn.Linear(n_hidden, 1)

    # -- _EncodedMLP plumbing ------------------------------------------------------------------------------------------------
    @property
    def out(self) -> nn.Linear:
        return self.W2

    def _hidden(self) -> List[nn.Linear]:
        return [self.W1]

    def _layer0(self):
        return self.W0.weight, self.W0.bias

    def _unpack_layer0(self, sd):
        return {"offset": self.offset.detach().cpu().clone(), "orientation": self.orientation.detach().cpu().clone(),
                "W0.weight": sd["input.weight"], "W0.bias": sd["input.bias"]}

    def _engine_unpack(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        sd = K.unpack_params(self.spec, flat)
        out = self._unpack_layer0(sd)
        out["W1.weight"], out["W1.bias"] = sd["skip.0.ln.weight"], sd["skip.0.ln.bias"]
        out["W2.weight"], out["W2.bias"] = sd["out.ln.weight"], sd["out.ln.bias"]
        return out

    unpack_flat = _engine_unpack

    def _engine_fit(self, grid, unaries, flat, epochs, cold, opts, states=None):
        """`pretrain(...)` (PriorFitMixin): the per-image fits run on the features of this module's current pose."""
        co = grid.coords
        off, ori = self.offset.detach().to(co.device), self.orientation.detach().to(co.device)
        f = lambda c: polar_symmetry_features(c.t(), off, ori, self.symmetry_prior).t()  # noqa: E731
        feats = f(co) if co.dim() == 2 else torch.stack([f(c) for c in co])
        return super()._engine_fit(K.Grid.explicit(feats.contiguous()), unaries, flat, epochs, cold, opts, states)

    # -- forward ---------------------------------------------------------------------------------------------------------------
    def features(self, x: torch.Tensor, symmetry_prior: Optional[bool] = None) -> torch.Tensor:
        sp = self.symmetry_prior if symmetry_prior is None else bool(symmetry_prior)
        return polar_symmetry_features(x, self.offset, self.orientation, sp)

    def forward(self, x: torch.Tensor, symmetry_prior: Optional[bool] = None, *args, **kwargs) -> torch.Tensor:
        """(N, 2) -> (N, 1) logits, or (B, 2, H, W) -> (B, 1, H, W); `symmetry_prior` as in the notebook's forward (default: the
        constructor's)."""
        if not x.is_cuda:
            raise RuntimeError("awesome_amd modules run on the MI355X only (no CPU fallback); move module and input to cuda")
        if x.dim() == 4:
            b, c, h, w = x.shape
            flat = x.permute(0, 2, 3, 1).reshape(b * h * w, c)
            return self.forward(flat, symmetry_prior).reshape(b, h, w, 1).permute(0, 3, 1, 2)
        if x.dim() != 2 or x.shape[1] != 2:
            raise ValueError(f"expected (N, 2) or (B, 2, H, W) coordinates, got {tuple(x.shape)}")
        feats = self.features(x, symmetry_prior)
        return _IcnnFunction.apply(feats.t().contiguous(), self.spec, *self._param_list(False))[:, None]

    # -- fused fits ------------------------------------------------------------------------------------------------------------
    def fit(self, coords: torch.Tensor, targets: torch.Tensor, steps: int, lr: float = 1e-3, symmetry_prior: Optional[bool] = None,
            optimizer: str = "adam", loss: str = "se", **fit_kwargs) -> K.FitResult:
        """`steps` full-batch optimizer steps of W0, W1, W2 on `inrfit_fit` under the current pose (coords (N, 2), targets (N,) in
        [0, 1]: the value sigmoid(logits) is fitted to).  Updates this module's weights; returns the FitResult (logits, loss_hist)."""
        with torch.no_grad():
            feats = self.features(coords, symmetry_prior).t().contiguous()
        flat = self.flat_parameters()[None].to(coords.device).contiguous()
        res = K.fit(self.spec, flat, K.Grid.explicit(feats), targets.reshape(1, -1).to(torch.float32).contiguous(), int(steps), lr=lr,
                    loss=loss, optimizer=optimizer, **{**self.fit_options, **fit_kwargs})
        if int(res.status[0]) != 0:
            raise ValueError("Loss is nan or inf!")
        self.load_state_dict({k: v.to(coords.device) for k, v in self.unpack_flat(res.params[0].cpu()).items()})
        return res

    def fit_alternating(self, coords: torch.Tensor, targets: torch.Tensor, rounds: int, net_steps: int, pose_steps: int,
                        lr: float = 1e-3, pose_lr: float = 1e-3, symmetry_prior: Optional[bool] = None) -> List[float]:
        """Block-coordinate version of the notebook's loop: per round `net_steps` fused steps of the network under the fixed pose,
        then `pose_steps` Adam steps of (offset, orientation) through the HIP forward/backward.  Returns the loss after each round."""
        pose_opt = torch.optim.Adam([self.offset, self.orientation], lr=pose_lr)
        tgt = targets.reshape(-1, 1).to(torch.float32)
        hist = []
        for _ in range(int(rounds)):
            self.fit(coords, targets, net_steps, lr=lr, symmetry_prior=symmetry_prior, record_loss=False, want_logits=False)
            for _ in range(int(pose_steps)):
                pose_opt.zero_grad(set_to_none=True)
                for p in (self.W0, self.W1, self.W2):
                    p.zero_grad(set_to_none=True)
                loss = ((torch.sigmoid(self(coords, symmetry_prior)) - tgt) ** 2).mean()
                loss.backward()
                pose_opt.step()
            with torch.no_grad():
                hist.append(float(((torch.sigmoid(self(coords, symmetry_prior)) - tgt) ** 2).mean()))
        return hist
