"""Star-shape prior of the teaser (notebooks/icml_teaser_code/star_shaped/star.ipynb cell 2).

    x <- x + offset,  r <- |x|,  x^ <- x / (0.01 + r)
    x_old <- relu(W0 x^)                                  a function of the direction alone
    r_aug <- relu(W1 x_old + W1_r r)
    out   <- r * (W2 x_old + W2_r r_aug) - 1              W2_r >= 0 (projected after every optimizer step)

Along every ray from the centre the bracket is convex and non-decreasing in r where W1_r >= 0, and out(0) = -1: the sub-level set
{out < 0} is star-shaped about -offset.

The read-out takes two layers at once, which the fused step kernels' layer chain (input -> z0 -> z1 -> w_o z1 + s_o x) does not have.
Both terms are nevertheless instances of that chain on the features [x^, r] (C = 3), so the module evaluates them as TWO calls of the
same HIP network sharing W0:

    W2_r r_aug + b   = chain(W_in = [W0 | 0], W_1 = W1, S_1 = [0 | W1_r], w_o = W2_r, s_o = 0)
    W2 x_old         = chain(W_in = [W0 | 0], W_1 = I,  S_1 = 0,          w_o = W2,   s_o = 0)     relu(I z0) = z0 exactly (z0 >= 0)

with the polar split, the product with r and the shared-parameter bookkeeping left to torch autograd on the device (a few
elementwise operations per point).  Forward and backward of everything with a matrix in it run through `inrfit_forward` /
`inrfit_backward`; `offset` receives its gradient through the kernels' coordinate gradient.  That is the module / autograd surface:
the notebook's loop (random 1000-pixel minibatches, a torch optimizer) runs unchanged on this class.

The loop itself also exists on the device (`fit` -> `inrfit_star_fit`, csrc/star.h: the network evaluated layer by layer on the
minibatch with its own read-out, MSE on the sigmoid, every gradient incl. the centre's, torch's Adam and the projection - no autograd,
no torch optimizer, no host synchronisation), and `forward_fused` is the inference of cells 4 / 6 (`inrfit_star_forward`).  Any width
up to 1024 (the notebook's 150 included); the autograd surface takes what `inrfit_supported` takes."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import icnn as K
from .. import star as S
from .convex_net import _IcnnFunction


class StarShapedNet(nn.Module):
    def __init__(self, n_hidden: int = 130, **kwargs):
        super().__init__()
        self.spec = K.IcnnSpec(n_hidden, 3, 1)
        # the notebook's names and creation order (a seeded construction gives its initial weights)
        self.offset = nn.Parameter(torch.zeros(1, 2), requires_grad=False)   # the notebook frees it after 1000 epochs
        self.W0 = nn.Linear(2, n_hidden)
        self.W1 = nn.Linear(n_hidden, n_hidden)
        self.W2 = nn.Linear(n_hidden, 1)
        self.W1_r = nn.Linear(1, n_hidden)
        self.W2_r = nn.Linear(n_hidden, 1)

    def enforce_star_shape(self) -> None:
        """The projection the notebook applies after every optimizer step: W2_r.weight <- relu(W2_r.weight)."""
        with torch.no_grad():
            self.W2_r.weight.clamp_(min=0.0)

    enforce_convexity = enforce_star_shape   # the prior-module contract name (WrapperModule calls it after each step)

    # ---- the device-resident surface (awesome_amd/star.py) -----------------------------------------------------------------------
    @property
    def star_spec(self) -> "S.StarSpec":
        return S.StarSpec(self.spec.n_hidden)

    def flat_parameters(self) -> torch.Tensor:
        """[P] float32 on the module's device, in the order of named_parameters() (= include/inrfit.h's layout)."""
        return S.flatten_state_dict(self.star_spec, {k: v for k, v in self.named_parameters()}, self.W0.weight.device)

    def load_flat(self, flat: torch.Tensor) -> None:
        with torch.no_grad():
            named = dict(self.named_parameters())
            for k, v in S.unflatten(self.star_spec, flat).items():
                named[k].copy_(v)

    @torch.no_grad()
    def forward_fused(self, x: torch.Tensor) -> torch.Tensor:
        """Inference without autograd: (N, 2) -> (N, 1) through inrfit_star_forward (star.ipynb cells 4 / 6)."""
        if not x.is_cuda:
            raise RuntimeError("awesome_amd modules run on the MI355X only (no CPU fallback); move module and input to cuda")
        return S.star_forward(self.star_spec, self.flat_parameters(), x.to(torch.float32).contiguous())[:, None]

    def fit(self, pixel_info: torch.Tensor, labels: torch.Tensor, num_epochs: int = 10000, number: int = 500, lr: float = 1e-2,
            offset_free_epoch: int = 1000, batch_index: torch.Tensor = None, generator: torch.Generator = None,
            state: dict = None) -> "S.StarFitResult":
        """star.ipynb cell 3 as one device-resident call: per epoch `number` random background + `number` random foreground pixels
        (labels = 1 - likelihood, so background has label 1), sigmoid -> MSELoss -> Adam(lr) over every parameter -> W2_r.weight <-
        relu(W2_r.weight); `offset` is freed after the forward pass of epoch `offset_free_epoch` (None: never), i.e. its first step is
        the next epoch's.  `batch_index` ([epochs, batch] pixel numbers) replaces the random draw.  `state` (a dict this call fills:
        opt_state, epoch) continues an earlier call.  The fitted parameters are loaded back into the module."""
        dev = self.W0.weight.device
        pixel_info = pixel_info.to(device=dev, dtype=torch.float32).contiguous()
        labels = labels.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        if batch_index is None:
            batch_index = S.notebook_minibatches(labels, num_epochs, number, generator)
        state = {} if state is None else state
        flat = self.flat_parameters()
        res = S.star_fit(self.star_spec, flat, pixel_info, labels, batch_index, lr=lr, step0=int(state.get("epoch", 0)),
                         offset_first_step=-1 if offset_free_epoch is None else int(offset_free_epoch) + 1,
                         opt_state=state.get("opt_state"))
        state["opt_state"], state["epoch"] = res.opt_state, int(state.get("epoch", 0)) + int(batch_index.shape[0])
        self.load_flat(res.params)
        return res

    def forward(self, x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        """(N, 2) -> (N, 1), or (B, 2, H, W) -> (B, 1, H, W)."""
        if not x.is_cuda:
            raise RuntimeError("awesome_amd modules run on the MI355X only (no CPU fallback); move module and input to cuda")
        if x.dim() == 4:
            b, c, h, w = x.shape
            return self.forward(x.permute(0, 2, 3, 1).reshape(b * h * w, c)).reshape(b, h, w, 1).permute(0, 3, 1, 2)
        if x.dim() != 2 or x.shape[1] != 2:
            raise ValueError(f"expected (N, 2) or (B, 2, H, W) coordinates, got {tuple(x.shape)}")
        hdn, dev = self.spec.n_hidden, x.device
        z = lambda *shape: torch.zeros(*shape, device=dev)  # noqa: E731
        x = x + self.offset
        r = torch.sqrt(torch.sum(x ** 2, dim=1, keepdim=True))
        feats = torch.cat((x / (0.01 + r), r), dim=1).t().contiguous()                      # (3, N): [direction, radius]
        w_in = torch.cat((self.W0.weight, z(hdn, 1)), dim=1)                               # the radius does not enter layer 0
        radial = _IcnnFunction.apply(feats, self.spec, w_in, self.W0.bias,
                                     self.W1.weight, self.W1.bias + self.W1_r.bias, torch.cat((z(hdn, 2), self.W1_r.weight), dim=1),
                                     self.W2_r.weight, self.W2_r.bias + self.W2.bias, z(1, 3))
        angular = _IcnnFunction.apply(feats, self.spec, w_in, self.W0.bias,
                                      torch.eye(hdn, device=dev), z(hdn), z(hdn, 3),
                                      self.W2.weight, z(1), z(1, 3))
        return r * (angular + radial)[:, None] - 1
