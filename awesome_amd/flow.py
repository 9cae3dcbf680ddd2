"""Host-side driver of the HIP coupling-flow kernels and of the fused ICNN(flow(Ax+b)) fit (include/inrfit.h, flow part).

  flow_forward  <- ConvexDiffeomorphismNet.get_deformation            (awesome/model/convex_diffeomorphism_net.py:179-184)
  cdn_forward   <- ConvexDiffeomorphismNet.forward                    (:173-178)
  cdn_loss_grad <- criterion(sigmoid(model(grid)), unaries).backward()  w.r.t. every parameter
  cdn_fit       <- the inner loop of ConvexDiffeomorphismNet.pretrain (:405-430)
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L
from . import icnn as K

Tensor = torch.Tensor


@dataclass(frozen=True)
class FlowSpec:
    width: int = 130
    num_coupling: int = 6
    backbone: str = "normal_block"   # 'normal_block' (= 'residual_block') | 'default' (SimpleBackbone), diffeomorphism_net.py:252-268

    def desc(self) -> L.InrFlowDesc:
        return L.InrFlowDesc(self.width, self.num_coupling, L.INR_FLOW_SIMPLE if self.backbone == "default" else L.INR_FLOW_NORMAL_BLOCK)

    @property
    def n_params(self) -> int:
        return 6 + 2 * self.num_coupling * (3 * self.width + 3) + 4 * self.num_coupling

    def keys_shapes(self, prefix: str = "diffeo_net.", linear_prefix: str = "linear.") -> List[Tuple[str, Tuple[int, ...]]]:
        """state_dict keys of ConvexDiffeomorphismNet's flow part in flat-vector order."""
        W, out = self.width, [(linear_prefix + "weight", (2, 2)), (linear_prefix + "bias", (2,))]
        l1, l2 = ("linear1", "linear2") if self.backbone == "default" else ("in_linear", "out_linear")
        for i in range(self.num_coupling):
            for net in ("s", "t"):
                b = f"{prefix}{net}.{i}."
                out += [(b + l1 + ".linear.weight_v", (W, 1)), (b + l1 + ".linear.weight_g", ()),
                        (b + l1 + ".linear.bias", (W,)), (b + l2 + ".linear.weight_v", (1, W)),
                        (b + l2 + ".linear.weight_g", ()), (b + l2 + ".linear.bias", (1,))]
        for i in range(self.num_coupling):
            b = f"{prefix}scale.{i}."
            out += [(b + "weight", (1,)), (b + "scale.bias", (1,)), (b + "scale.weight_g", (1, 1)), (b + "scale.weight_v", (1, 1))]
        return out

    def weight_g_mask(self) -> Tensor:
        """1 where the flat flow parameter is a *weight_g (the group that gets weight decay, awesome/util/torch.py:19-35)."""
        parts = []
        for k, shp in self.keys_shapes():
            n = 1
            for s in shp:
                n *= s
            parts.append(torch.full((n,), 1.0 if k.endswith("weight_g") else 0.0))
        return torch.cat(parts)


def pack_flow_state_dict(spec: FlowSpec, sd: Dict[str, Tensor], device=None) -> Tensor:
    parts = []
    for k, shp in spec.keys_shapes():
        t = sd[k]
        if tuple(t.shape) != shp:
            raise ValueError(f"{k}: expected shape {shp}, got {tuple(t.shape)}")
        parts.append(t.detach().reshape(-1).to(torch.float32))
    flat = torch.cat(parts)
    return flat.to(device) if device is not None else flat


def unpack_flow_params(spec: FlowSpec, flat: Tensor) -> Dict[str, Tensor]:
    flat = flat.detach().reshape(-1).clone()
    out, off = {}, 0
    for k, shp in spec.keys_shapes():
        n = 1
        for s in shp:
            n *= s
        out[k] = flat[off:off + n].reshape(shp)
        off += n
    return out


def split_cdn_state_dict(ispec: K.IcnnSpec, fspec: FlowSpec, sd: Dict[str, Tensor], device=None) -> Tuple[Tensor, Tensor]:
    """ConvexDiffeomorphismNet.state_dict() -> (flat ICNN params, flat flow params)."""
    icnn_sd = {k[len("convex_net."):]: v for k, v in sd.items() if k.startswith("convex_net.")}
    return K.pack_state_dict(ispec, icnn_sd, device), pack_flow_state_dict(fspec, sd, device)


def merge_cdn_state_dict(ispec: K.IcnnSpec, fspec: FlowSpec, icnn_flat: Tensor, flow_flat: Tensor) -> Dict[str, Tensor]:
    sd = {"convex_net." + k: v for k, v in K.unpack_params(ispec, icnn_flat).items()}
    sd.update(unpack_flow_params(fspec, flow_flat))
    return sd


def _ws(ispec: Optional[K.IcnnSpec], fspec: FlowSpec, grid: K.Grid, n_images: int) -> Tensor:
    md = ispec.desc() if ispec is not None else None
    fd, gd = fspec.desc(), grid.desc()
    nbytes = L.load().inrfit_cdn_workspace_bytes(C.byref(md) if md is not None else None, C.byref(fd), C.byref(gd), n_images)
    if nbytes < 0:
        L.check(int(nbytes), "inrfit_cdn_workspace_bytes")
    return L.scratch(int(nbytes) // 4 + 1, dtype=torch.float32, device=grid.device)


def flow_forward(fspec: FlowSpec, flow_params: Tensor, grid: K.Grid) -> Tensor:
    """flow_params [n_images, FP] -> deformed coordinates [n_images, 2, N]."""
    fp = K._check_dev(flow_params, "flow_params")
    n = fp.shape[0]
    out = L.scratch(n, 2, grid.n_points, dtype=torch.float32, device=fp.device)
    ws = _ws(None, fspec, grid, n)
    fd, gd = fspec.desc(), grid.desc()
    rc = L.load().inrfit_flow_forward(C.byref(fd), fp.data_ptr(), C.byref(gd), n, out.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                      K._stream_ptr(fp.device))
    L.check(rc, "inrfit_flow_forward")
    return out


def flow_backward(fspec: FlowSpec, flow_params: Tensor, grid: K.Grid, dout_coords: Tensor) -> Tensor:
    """Vector-Jacobian product of flow_forward: dout_coords [n_images, 2, N] -> gradients [n_images, FP] (forward recomputed)."""
    fp = K._check_dev(flow_params, "flow_params")
    d = K._check_dev(dout_coords, "dout_coords")
    n = fp.shape[0]
    assert d.shape == (n, 2, grid.n_points), (d.shape, n, grid.n_points)
    grads = L.scratch_like(fp)
    ws = _ws(None, fspec, grid, n)
    fd, gd = fspec.desc(), grid.desc()
    rc = L.load().inrfit_flow_backward(C.byref(fd), fp.data_ptr(), C.byref(gd), d.data_ptr(), n, grads.data_ptr(), ws.data_ptr(),
                                       ws.numel() * 4, K._stream_ptr(fp.device))
    L.check(rc, "inrfit_flow_backward")
    return grads


def cdn_forward(ispec: K.IcnnSpec, fspec: FlowSpec, icnn_params: Tensor, flow_params: Tensor, grid: K.Grid) -> Tensor:
    ip, fp = K._check_dev(icnn_params, "icnn_params"), K._check_dev(flow_params, "flow_params")
    n = ip.shape[0]
    logits = L.scratch(n, grid.n_points, dtype=torch.float32, device=ip.device)
    ws = _ws(ispec, fspec, grid, n)
    md, fd, gd = ispec.desc(), fspec.desc(), grid.desc()
    rc = L.load().inrfit_cdn_forward(C.byref(md), C.byref(fd), ip.data_ptr(), fp.data_ptr(), C.byref(gd), n, logits.data_ptr(),
                                     ws.data_ptr(), ws.numel() * 4, K._stream_ptr(ip.device))
    L.check(rc, "inrfit_cdn_forward")
    return logits


def cdn_loss_grad(ispec: K.IcnnSpec, fspec: FlowSpec, icnn_params: Tensor, flow_params: Tensor, grid: K.Grid, targets: Tensor,
                  loss: str = "bce", weight_mode: str = "none", ratio: float = 1.0) -> Tuple[Tensor, Tensor, Tensor]:
    ip, fp = K._check_dev(icnn_params, "icnn_params"), K._check_dev(flow_params, "flow_params")
    n = ip.shape[0]
    targets = K._check_dev(targets, "targets").reshape(n, -1)
    lo = L.scratch(n, dtype=torch.float32, device=ip.device)
    gi, gf = L.scratch_like(ip), L.scratch_like(fp)
    ws = _ws(ispec, fspec, grid, n)
    md, fd, gd, ld = ispec.desc(), fspec.desc(), grid.desc(), K._loss_desc(loss, weight_mode, ratio, 0.0, 0.0)
    rc = L.load().inrfit_cdn_loss_grad(C.byref(md), C.byref(fd), ip.data_ptr(), fp.data_ptr(), C.byref(gd), targets.data_ptr(),
                                       C.byref(ld), n, lo.data_ptr(), gi.data_ptr(), gf.data_ptr(), ws.data_ptr(),
                                       ws.numel() * 4, K._stream_ptr(ip.device))
    L.check(rc, "inrfit_cdn_loss_grad")
    return lo, gi, gf


@dataclass
class CdnFitResult:
    icnn_params: Tensor
    flow_params: Tensor
    icnn_opt_state: Tensor
    flow_opt_state: Tensor
    loss_hist: Optional[Tensor]
    logits: Optional[Tensor]
    status: Tensor


def cdn_fit(ispec: K.IcnnSpec, fspec: FlowSpec, icnn_params: Tensor, flow_params: Tensor, grid: K.Grid, targets: Tensor,
            steps: int, lr: float = 3e-3, loss: str = "bce", weight_mode: str = "none", ratio: float = 1.0,
            weight_decay_on_weight_g: float = 5e-5, betas=(0.9, 0.999), eps: float = 1e-8, plateau: Optional[dict] = None,
            icnn_opt_state: Optional[Tensor] = None, flow_opt_state: Optional[Tensor] = None, step0: int = 0,
            record_loss: bool = True, want_logits: bool = True, gate_logits: bool = False) -> CdnFitResult:
    """ConvexDiffeomorphismNet.pretrain's inner loop on the device (defaults: Adam lr 3e-3, BCE, wd 5e-5 on weight_g)."""
    ip, fp = K._check_dev(icnn_params, "icnn_params"), K._check_dev(flow_params, "flow_params")
    n, dev = ip.shape[0], ip.device
    targets = K._check_dev(targets, "targets").reshape(n, -1)
    if icnn_opt_state is None:
        icnn_opt_state = K.new_opt_state(ispec, n, dev)
    if flow_opt_state is None:
        flow_opt_state = torch.zeros(n, 2 * fspec.n_params, dtype=torch.float32, device=dev)
    hist = L.scratch(n, max(steps, 1), dtype=torch.float32, device=dev) if record_loss else None
    logits = L.scratch(n, grid.n_points, dtype=torch.float32, device=dev) if want_logits else None
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    pl = plateau or {}
    od = L.InrOptDesc(L.INR_OPT_ADAM, float(lr), float(betas[0]), float(betas[1]), float(eps), 0.0, 1, int(plateau is not None),
                      int(pl.get("patience", 200)), float(pl.get("factor", 0.5)), float(pl.get("threshold", 1e-4)),
                      float(pl.get("min_lr", 0.0)), float(pl.get("eps", 1e-8)), 0, 0, int(bool(gate_logits)))
    ws = _ws(ispec, fspec, grid, n)
    md, fd, gd, ld = ispec.desc(), fspec.desc(), grid.desc(), K._loss_desc(loss, weight_mode, ratio, 0.0, 0.0)
    rc = L.load().inrfit_cdn_fit(C.byref(md), C.byref(fd), ip.data_ptr(), fp.data_ptr(), icnn_opt_state.data_ptr(),
                                 flow_opt_state.data_ptr(), C.byref(gd), targets.data_ptr(), C.byref(ld), C.byref(od),
                                 float(weight_decay_on_weight_g), n, int(steps), int(step0),
                                 hist.data_ptr() if hist is not None else None,
                                 logits.data_ptr() if logits is not None else None, status.data_ptr(), ws.data_ptr(),
                                 ws.numel() * 4, K._stream_ptr(dev))
    L.check(rc, "inrfit_cdn_fit")
    return CdnFitResult(ip, fp, icnn_opt_state, flow_opt_state, hist[:, :steps] if hist is not None else None, logits, status)
