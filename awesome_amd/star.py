"""Host side of the star-shape prior's device-resident entry points (include/inrfit.h: inrfit_star_forward / _loss_grad / _fit;
kernels in csrc/star.h).  Mirrors notebooks/icml_teaser_code/star_shaped/star.ipynb: `myNet` (cell 2) and its training loop (cell 3).

Parameters travel as ONE flat vector in the order of the notebook class's named_parameters() (PARAM_ORDER)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import _lib as L
from . import icnn as K

Tensor = torch.Tensor

PARAM_ORDER = ("offset", "W0.weight", "W0.bias", "W1.weight", "W1.bias", "W2.weight", "W2.bias", "W1_r.weight", "W1_r.bias",
               "W2_r.weight", "W2_r.bias")
MAX_BATCH = 1 << 16


@dataclass(frozen=True)
class StarSpec:
    n_hidden: int

    def desc(self) -> L.InrStarDesc:
        return L.InrStarDesc(int(self.n_hidden))

    @property
    def shapes(self) -> Dict[str, Tuple[int, ...]]:
        h = self.n_hidden
        return {"offset": (1, 2), "W0.weight": (h, 2), "W0.bias": (h,), "W1.weight": (h, h), "W1.bias": (h,), "W2.weight": (1, h),
                "W2.bias": (1,), "W1_r.weight": (h, 1), "W1_r.bias": (h,), "W2_r.weight": (1, h), "W2_r.bias": (1,)}

    @property
    def n_params(self) -> int:
        h = self.n_hidden
        return h * h + 8 * h + 4


def flatten_state_dict(spec: StarSpec, sd: Dict[str, Tensor], device) -> Tensor:
    """state_dict of the notebook class (or of StarShapedNet) -> [P] float32 on `device`."""
    parts = []
    for k, shape in spec.shapes.items():
        v = sd[k].detach().to(torch.float32)
        if tuple(v.shape) != shape:
            raise ValueError(f"{k}: expected {shape}, got {tuple(v.shape)}")
        parts.append(v.reshape(-1))
    return torch.cat(parts).to(device).contiguous()


def unflatten(spec: StarSpec, flat: Tensor) -> Dict[str, Tensor]:
    out, o = {}, 0
    for k, shape in spec.shapes.items():
        n = 1
        for s in shape:
            n *= s
        out[k] = flat[o:o + n].reshape(shape).clone()
        o += n
    return out


def _workspace(spec: StarSpec, n_points: int, dev) -> Tensor:
    lib = L.load()
    d = spec.desc()
    nbytes = int(lib.inrfit_star_workspace_bytes(C.byref(d), int(n_points)))
    if nbytes < 0:
        L.check(nbytes, "inrfit_star_workspace_bytes")
    return L.scratch(nbytes // 4 + 64, dtype=torch.float32, device=dev)


def _check(params: Tensor, spec: StarSpec, coords: Tensor):
    if not params.is_contiguous():
        raise ValueError("params must be one contiguous flat vector (it is updated in place)")
    params = K._check_dev(params, "params")
    coords = K._check_dev(coords, "coords")
    if params.numel() != spec.n_params:
        raise ValueError(f"params: expected {spec.n_params} floats, got {params.numel()}")
    if coords.dim() != 2 or coords.shape[1] != 2:
        raise ValueError(f"coords: expected (n_pixels, 2), got {tuple(coords.shape)}")
    return params, coords


def star_forward(spec: StarSpec, params: Tensor, coords: Tensor) -> Tensor:
    """logits [n] of every row of coords [n, 2] (cell 4 / 6 of the notebook: inference on all pixels)."""
    params, coords = _check(params, spec, coords)
    n, dev = coords.shape[0], params.device
    out = L.scratch(n, dtype=torch.float32, device=dev)
    ws = _workspace(spec, n, dev)
    d = spec.desc()
    rc = L.load().inrfit_star_forward(C.byref(d), params.data_ptr(), coords.data_ptr(), n, out.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                      K._stream_ptr(dev))
    L.check(rc, "inrfit_star_forward")
    return out


def star_loss_grad(spec: StarSpec, params: Tensor, coords: Tensor, labels: Tensor, index: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """(loss [1], grads [P]) of MSE(sigmoid(net(coords[index])), labels[index]); index None = every pixel."""
    params, coords = _check(params, spec, coords)
    labels = K._check_dev(labels, "labels")
    dev, npx = params.device, coords.shape[0]
    if labels.numel() != npx:
        raise ValueError("labels: one per pixel")
    if index is not None:
        index = index.to(device=dev, dtype=torch.int32).contiguous()
    batch = int(index.numel()) if index is not None else npx
    loss = L.scratch(1, dtype=torch.float32, device=dev)
    grads = L.scratch(spec.n_params, dtype=torch.float32, device=dev)
    ws = _workspace(spec, batch, dev)
    d = spec.desc()
    rc = L.load().inrfit_star_loss_grad(C.byref(d), params.data_ptr(), coords.data_ptr(), labels.data_ptr(), npx,
                                        index.data_ptr() if index is not None else None, batch, loss.data_ptr(), grads.data_ptr(),
                                        ws.data_ptr(), ws.numel() * 4, K._stream_ptr(dev))
    L.check(rc, "inrfit_star_loss_grad")
    return loss, grads


@dataclass
class StarFitResult:
    params: Tensor              # [P], same storage as the input
    opt_state: Tensor           # [2P] exp_avg | exp_avg_sq
    loss_hist: Optional[Tensor]  # [steps]


def star_fit(spec: StarSpec, params: Tensor, coords: Tensor, labels: Tensor, batch_index: Tensor, lr: float = 1e-2,
             betas=(0.9, 0.999), eps: float = 1e-8, step0: int = 0, offset_first_step: int = 1001,
             opt_state: Optional[Tensor] = None, record_loss: bool = True) -> StarFitResult:
    """The loop of star.ipynb cell 3 on the device: epoch e = step0 + k takes the minibatch batch_index[k] ([steps, batch] pixel numbers),
    Adam(lr) on every parameter IN PLACE, then W2_r.weight <- relu(W2_r.weight).  `offset` takes its first step at epoch
    offset_first_step (the notebook: requires_grad after the forward of epoch 1000 -> 1001; < 0: never)."""
    params, coords = _check(params, spec, coords)
    labels = K._check_dev(labels, "labels")
    dev, npx = params.device, coords.shape[0]
    if labels.numel() != npx:
        raise ValueError("labels: one per pixel")
    if batch_index.dim() != 2:
        raise ValueError("batch_index: expected (steps, batch)")
    batch_index = batch_index.to(device=dev, dtype=torch.int32).contiguous()
    steps, batch = int(batch_index.shape[0]), int(batch_index.shape[1])
    if batch < 1 or batch > MAX_BATCH:
        raise ValueError(f"batch must be in [1, {MAX_BATCH}]")
    if opt_state is None:
        opt_state = torch.zeros(2 * spec.n_params, dtype=torch.float32, device=dev)
    assert opt_state.numel() == 2 * spec.n_params and opt_state.is_contiguous() and opt_state.device == dev
    hist = L.scratch(max(steps, 1), dtype=torch.float32, device=dev) if record_loss else None
    ws = _workspace(spec, batch, dev)
    d = spec.desc()
    od = L.InrOptDesc(L.OPT_KINDS["adam"], float(lr), float(betas[0]), float(betas[1]), float(eps), 0.0, 0, 0, 0, 1.0, 0.0, 0.0, 0.0, 0, 0, 0)
    rc = L.load().inrfit_star_fit(C.byref(d), params.data_ptr(), opt_state.data_ptr(), coords.data_ptr(), labels.data_ptr(), npx,
                                  batch_index.data_ptr(), batch, C.byref(od), steps, int(step0), int(offset_first_step),
                                  hist.data_ptr() if hist is not None else None, ws.data_ptr(), ws.numel() * 4, K._stream_ptr(dev))
    L.check(rc, "inrfit_star_fit")
    return StarFitResult(params, opt_state, hist[:steps] if hist is not None else None)


def notebook_minibatches(labels: Tensor, steps: int, number: int = 500, generator: Optional[torch.Generator] = None,
                         chunk: int = 64) -> Tensor:
    """[steps, 2 * number] pixel numbers: per epoch `number` random background pixels (label > 0.5: cell 3's labels = 1 - likelihood)
    followed by `number` random foreground pixels, each a uniformly random subset without repetition - what
    `torch.randperm(n)[:number]` of cell 3 draws.  Random keys + top-k on the device, `chunk` epochs at a time."""
    dev = labels.device
    back = torch.nonzero(labels.reshape(-1) > 0.5).reshape(-1)
    fore = torch.nonzero(labels.reshape(-1) <= 0.5).reshape(-1)
    if back.numel() < number or fore.numel() < number:
        raise ValueError(f"need at least {number} pixels of each class (background {back.numel()}, foreground {fore.numel()})")
    out = torch.empty(steps, 2 * number, dtype=torch.int32, device=dev)
    for s0 in range(0, steps, chunk):
        n = min(chunk, steps - s0)
        for col, pool in ((0, back), (number, fore)):
            keys = torch.rand(n, pool.numel(), device=dev, generator=generator)
            out[s0:s0 + n, col:col + number] = pool[torch.topk(keys, number, dim=1).indices].to(torch.int32)
    return out
