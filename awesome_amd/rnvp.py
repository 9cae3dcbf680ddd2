"""Host-side driver of the HIP RealNVP kernels and of the fused PathConnectedNet fit (include/inrfit.h, InrRnvpDesc part).

  rnvp_forward  <- PathConnectedNet.get_deformation                 (awesome/model/path_connected_net.py:124-128)
  pcn_forward   <- PathConnectedNet.forward                         (:79-85)
  pcn_loss_grad <- criterion(sigmoid(model(grid)), unaries).backward()  w.r.t. every parameter
  pcn_fit       <- the inner loop of _prior_based_pretrain          (:937-962, Adamax + param groups :922-929)
  actnorm_init  <- the data-dependent first forward of nf.flows.ActNorm

The flow itself (MaskedAffineFlow / ActNorm / MLP) is normflows==1.7.3 code that is not part of the reference checkout:
parity for this variant is UNPINNED; the flow is built as awesome/model/net_factory.py:70-114 builds it.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from . import icnn as K

Tensor = torch.Tensor


def rnvp_masks(channels: int, n_flows: int) -> List[int]:
    """Coupling masks of init_realnvp (awesome/model/net_factory.py:86-99): 1 .. 2^C - 2 counted in binary (bit c = channel c),
    repeated to n_flows entries."""
    base = list(range(1, 2 ** channels - 1))
    return [base[i % len(base)] for i in range(n_flows)]


@dataclass(frozen=True)
class RnvpSpec:
    channels: int = 2
    hidden_units: int = 32
    n_flows: int = 12
    output_fn: Optional[str] = "tanh"
    output_scale: Optional[float] = None
    vmin: Tuple[float, ...] = (0.0, 0.0)
    vmax: Tuple[float, ...] = (1.0, 1.0)
    new_min: float = -1.0
    new_max: float = 1.0
    masks: Tuple[int, ...] = field(default=())

    def __post_init__(self):
        if self.output_fn not in (None, "tanh"):
            raise ValueError("output_fn must be None or 'tanh' (the only ones the reference configs use)")
        if not self.masks:
            object.__setattr__(self, "masks", tuple(rnvp_masks(self.channels, self.n_flows)))
        if len(self.vmin) != self.channels or len(self.vmax) != self.channels:
            object.__setattr__(self, "vmin", tuple([0.0] * self.channels) if len(self.vmin) != self.channels else self.vmin)
            object.__setattr__(self, "vmax", tuple([1.0] * self.channels) if len(self.vmax) != self.channels else self.vmax)

    def desc(self) -> L.InrRnvpDesc:
        d = L.InrRnvpDesc()
        d.channels, d.hidden_units, d.n_flows = self.channels, self.hidden_units, self.n_flows
        d.output_fn = 1 if self.output_fn == "tanh" else 0
        d.output_scale = float(self.output_scale) if self.output_scale is not None else 1.0
        for c in range(self.channels):
            d.vmin[c], d.vmax[c] = float(self.vmin[c]), float(self.vmax[c])
        d.new_min, d.new_max = float(self.new_min), float(self.new_max)
        for f, m in enumerate(self.masks):
            d.masks[f] = int(m)
        return d

    @property
    def net_params(self) -> int:
        return 2 * self.hidden_units * self.channels + self.hidden_units + self.channels

    @property
    def n_params(self) -> int:
        return 2 * self.channels + self.n_flows * (2 * self.net_params + 2 * self.channels)

    def keys_shapes(self, prefix: str = "flow_net.net.network.", linear_prefix: str = "linear.") -> List[Tuple[str, Tuple[int, ...]]]:
        """state_dict keys of PathConnectedNet's flow part (NormNet(PixelizeNet(nf.NormalizingFlow))) in flat-vector order."""
        c, h = self.channels, self.hidden_units
        out = [(linear_prefix + "weight", (c, 1, 1, 1)), (linear_prefix + "bias", (c,))]
        for f in range(self.n_flows):
            for net in ("s", "t"):
                b = f"{prefix}flows.{2 * f}.{net}.net."
                out += [(b + "0.weight", (h, c)), (b + "0.bias", (h,)), (b + "2.weight", (c, h)), (b + "2.bias", (c,))]
            b = f"{prefix}flows.{2 * f + 1}."
            out += [(b + "s", (1, c)), (b + "t", (1, c))]
        return out

    def actnorm_slices(self) -> List[Tuple[int, int]]:
        """(start, stop) of every flow's ActNorm s|t block in the flat vector."""
        c, pf = self.channels, 2 * self.net_params + 2 * self.channels
        return [(2 * c + f * pf + 2 * self.net_params, 2 * c + (f + 1) * pf) for f in range(self.n_flows)]


def pack_rnvp_state_dict(spec: RnvpSpec, sd: Dict[str, Tensor], device=None) -> Tensor:
    parts = []
    for k, shp in spec.keys_shapes():
        t = sd[k]
        if tuple(t.shape) != shp:
            raise ValueError(f"{k}: expected shape {shp}, got {tuple(t.shape)}")
        parts.append(t.detach().reshape(-1).to(torch.float32))
    flat = torch.cat(parts)
    return flat.to(device) if device is not None else flat


def unpack_rnvp_params(spec: RnvpSpec, flat: Tensor) -> Dict[str, Tensor]:
    flat = flat.detach().reshape(-1).clone()
    out, off = {}, 0
    for k, shp in spec.keys_shapes():
        n = 1
        for s in shp:
            n *= s
        out[k] = flat[off:off + n].reshape(shp)
        off += n
    return out


def init_rnvp_params(spec: RnvpSpec) -> Tensor:
    """Fresh flow parameters as the reference factory creates them (global torch RNG, like the reference): nn.Linear default
    init for the hidden layers, zeros for the MLPs' last layers (init_zeros=True, net_factory.py:104-105), ActNorm s = t = 0
    (set by actnorm_init on first use), linear weight 1 / bias 0 (path_connected_net.py:72-77)."""
    sd = {}
    for k, shp in spec.keys_shapes():
        if k.endswith("net.0.weight"):
            lin = torch.nn.Linear(shp[1], shp[0])
            sd[k] = lin.weight.data
            sd[k[:-len("weight")] + "bias"] = lin.bias.data
        elif k.endswith("net.0.bias"):
            continue
        elif k == "linear.weight":
            sd[k] = torch.ones(shp)
        else:
            sd[k] = torch.zeros(shp)
    return pack_rnvp_state_dict(spec, sd)


def _ws(ispec: Optional[K.IcnnSpec], rspec: RnvpSpec, grid: K.Grid, n_images: int) -> Tensor:
    md = ispec.desc() if ispec is not None else None
    rd, gd = rspec.desc(), grid.desc()
    nbytes = L.load().inrfit_pcn_workspace_bytes(C.byref(md) if md is not None else None, C.byref(rd), C.byref(gd), n_images)
    if nbytes < 0:
        L.check(int(nbytes), "inrfit_pcn_workspace_bytes")
    return L.scratch(int(nbytes) // 4 + 1, dtype=torch.float32, device=grid.device)


def actnorm_init(rspec: RnvpSpec, flow_params: Tensor, grid: K.Grid) -> Tensor:
    """Data-dependent ActNorm initialisation, in place on flow_params [n_images, RP]."""
    fp = K._check_dev(flow_params, "flow_params")
    n = fp.shape[0]
    ws = _ws(None, rspec, grid, n)
    rd, gd = rspec.desc(), grid.desc()
    rc = L.load().inrfit_rnvp_actnorm_init(C.byref(rd), fp.data_ptr(), C.byref(gd), n, ws.data_ptr(), ws.numel() * 4,
                                           K._stream_ptr(fp.device))
    L.check(rc, "inrfit_rnvp_actnorm_init")
    return fp


def rnvp_forward(rspec: RnvpSpec, flow_params: Tensor, grid: K.Grid) -> Tensor:
    """flow_params [n_images, RP] -> deformed coordinates [n_images, C, N]."""
    fp = K._check_dev(flow_params, "flow_params")
    n = fp.shape[0]
    out = L.scratch(n, rspec.channels, grid.n_points, dtype=torch.float32, device=fp.device)
    ws = _ws(None, rspec, grid, n)
    rd, gd = rspec.desc(), grid.desc()
    rc = L.load().inrfit_rnvp_forward(C.byref(rd), fp.data_ptr(), C.byref(gd), n, out.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                      K._stream_ptr(fp.device))
    L.check(rc, "inrfit_rnvp_forward")
    return out


def rnvp_inverse(rspec: RnvpSpec, flow_params: Tensor, coords: Tensor) -> Tensor:
    """PathConnectedNet.inverse (path_connected_net.py:107-122): coords (C, N) shared or (n_images, C, N) -> [n_images, C, N]."""
    fp = K._check_dev(flow_params, "flow_params")
    coords = K._check_dev(coords, "coords")
    n = fp.shape[0]
    stride = 0 if coords.dim() == 2 else coords.shape[1] * coords.shape[2]
    N = coords.shape[-1]
    out = L.scratch(n, rspec.channels, N, dtype=torch.float32, device=fp.device)
    ws = _ws(None, rspec, K.Grid.explicit(coords), n)
    rd = rspec.desc()
    rc = L.load().inrfit_rnvp_inverse(C.byref(rd), fp.data_ptr(), coords.data_ptr(), stride, N, n, out.data_ptr(), ws.data_ptr(),
                                      ws.numel() * 4, K._stream_ptr(fp.device))
    L.check(rc, "inrfit_rnvp_inverse")
    return out


def fit_identity(rspec: RnvpSpec, flow_params: Tensor, grid: K.Grid, steps: int = 100, lr: float = 1e-2, weight_decay: float = 1e-5,
                 optimizer: str = "adamax", betas=(0.9, 0.999), eps: float = 1e-8, flow_opt_state: Optional[Tensor] = None,
                 step0: int = 0) -> Tuple[Tensor, Tensor]:
    """PathConnectedNet.learn_flow_identity (path_connected_net.py:155-250; defaults of the prefit kwargs :771-774): the flow_net
    alone is trained towards the identity on the grid, in place on flow_params [n_images, RP].  Returns (loss_hist, opt_state)."""
    fp = K._check_dev(flow_params, "flow_params")
    n, dev = fp.shape[0], fp.device
    if flow_opt_state is None:
        flow_opt_state = torch.zeros(n, 2 * rspec.n_params, dtype=torch.float32, device=dev)
    hist = L.scratch(n, max(steps, 1), dtype=torch.float32, device=dev)
    od = L.InrOptDesc(L.OPT_KINDS[optimizer], float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), 0, 0,
                      0, 0.0, 0.0, 0.0, 0.0)
    ws = _ws(None, rspec, grid, n)
    rd, gd = rspec.desc(), grid.desc()
    rc = L.load().inrfit_rnvp_fit_identity(C.byref(rd), fp.data_ptr(), flow_opt_state.data_ptr(), C.byref(gd), C.byref(od), n,
                                           int(steps), int(step0), hist.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                           K._stream_ptr(dev))
    L.check(rc, "inrfit_rnvp_fit_identity")
    return hist[:, :steps], flow_opt_state


def pcn_forward(ispec: K.IcnnSpec, rspec: RnvpSpec, icnn_params: Tensor, flow_params: Tensor, grid: K.Grid) -> Tensor:
    ip, fp = K._check_dev(icnn_params, "icnn_params"), K._check_dev(flow_params, "flow_params")
    n = ip.shape[0]
    logits = L.scratch(n, grid.n_points, dtype=torch.float32, device=ip.device)
    ws = _ws(ispec, rspec, grid, n)
    md, rd, gd = ispec.desc(), rspec.desc(), grid.desc()
    rc = L.load().inrfit_pcn_forward(C.byref(md), C.byref(rd), ip.data_ptr(), fp.data_ptr(), C.byref(gd), n, logits.data_ptr(),
                                     ws.data_ptr(), ws.numel() * 4, K._stream_ptr(ip.device))
    L.check(rc, "inrfit_pcn_forward")
    return logits


def pcn_loss_grad(ispec: K.IcnnSpec, rspec: RnvpSpec, icnn_params: Tensor, flow_params: Tensor, grid: K.Grid, targets: Tensor,
                  loss: str = "se", weight_mode: str = "none", ratio: float = 1.0) -> Tuple[Tensor, Tensor, Tensor]:
    ip, fp = K._check_dev(icnn_params, "icnn_params"), K._check_dev(flow_params, "flow_params")
    n = ip.shape[0]
    targets = K._check_dev(targets, "targets").reshape(n, -1)
    lo = L.scratch(n, dtype=torch.float32, device=ip.device)
    gi, gf = L.scratch_like(ip), L.scratch_like(fp)
    ws = _ws(ispec, rspec, grid, n)
    md, rd, gd, ld = ispec.desc(), rspec.desc(), grid.desc(), K._loss_desc(loss, weight_mode, ratio, 0.0, 0.0)
    rc = L.load().inrfit_pcn_loss_grad(C.byref(md), C.byref(rd), ip.data_ptr(), fp.data_ptr(), C.byref(gd), targets.data_ptr(),
                                       C.byref(ld), n, lo.data_ptr(), gi.data_ptr(), gf.data_ptr(), ws.data_ptr(),
                                       ws.numel() * 4, K._stream_ptr(ip.device))
    L.check(rc, "inrfit_pcn_loss_grad")
    return lo, gi, gf


@dataclass
class PcnFitResult:
    icnn_params: Tensor
    flow_params: Tensor
    icnn_opt_state: Tensor
    flow_opt_state: Tensor
    loss_hist: Optional[Tensor]
    logits: Optional[Tensor]
    status: Tensor


def pcn_fit(ispec: K.IcnnSpec, rspec: RnvpSpec, icnn_params: Tensor, flow_params: Tensor, grid: K.Grid, targets: Tensor,
            steps: int, lr: float = 1e-3, optimizer: str = "adamax", loss: str = "se", weight_mode: str = "none",
            ratio: float = 1.0, flow_weight_decay: float = 1e-5, betas=(0.9, 0.999), eps: float = 1e-8,
            plateau: Optional[dict] = None, icnn_opt_state: Optional[Tensor] = None, flow_opt_state: Optional[Tensor] = None,
            step0: int = 0, record_loss: bool = True, want_logits: bool = True, gate_logits: bool = False) -> PcnFitResult:
    """_prior_based_pretrain's inner loop for PathConnectedNet on the device (defaults of path_connected_net.py:756-760,
    922-933: Adamax lr 1e-3, flow weight decay 1e-5, UnariesWeightedLoss(SE); pass plateau={} for ReduceLROnPlateau(200, 0.5))."""
    ip, fp = K._check_dev(icnn_params, "icnn_params"), K._check_dev(flow_params, "flow_params")
    n, dev = ip.shape[0], ip.device
    targets = K._check_dev(targets, "targets").reshape(n, -1)
    if icnn_opt_state is None:
        icnn_opt_state = K.new_opt_state(ispec, n, dev)
    if flow_opt_state is None:
        flow_opt_state = torch.zeros(n, 2 * rspec.n_params, dtype=torch.float32, device=dev)
    hist = L.scratch(n, max(steps, 1), dtype=torch.float32, device=dev) if record_loss else None
    logits = L.scratch(n, grid.n_points, dtype=torch.float32, device=dev) if want_logits else None
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    pl = plateau or {}
    od = L.InrOptDesc(L.OPT_KINDS[optimizer], float(lr), float(betas[0]), float(betas[1]), float(eps), 0.0, 1,
                      int(plateau is not None), int(pl.get("patience", 200)), float(pl.get("factor", 0.5)),
                      float(pl.get("threshold", 1e-4)), float(pl.get("min_lr", 0.0)), float(pl.get("eps", 1e-8)), 0, 0, int(bool(gate_logits)))
    ws = _ws(ispec, rspec, grid, n)
    md, rd, gd, ld = ispec.desc(), rspec.desc(), grid.desc(), K._loss_desc(loss, weight_mode, ratio, 0.0, 0.0)
    rc = L.load().inrfit_pcn_fit(C.byref(md), C.byref(rd), ip.data_ptr(), fp.data_ptr(), icnn_opt_state.data_ptr(),
                                 flow_opt_state.data_ptr(), C.byref(gd), targets.data_ptr(), C.byref(ld), C.byref(od),
                                 float(flow_weight_decay), n, int(steps), int(step0),
                                 hist.data_ptr() if hist is not None else None,
                                 logits.data_ptr() if logits is not None else None, status.data_ptr(), ws.data_ptr(),
                                 ws.numel() * 4, K._stream_ptr(dev))
    L.check(rc, "inrfit_pcn_fit")
    return PcnFitResult(ip, fp, icnn_opt_state, flow_opt_state, hist[:, :steps] if hist is not None else None, logits, status)
