"""Host-side driver of the HIP ICNN kernels (torch tensors in, torch tensors out; the compute is libinrfit.so).

The functions here are the batched, device-resident form of the reference hot path:
  forward   <- ConvexNet/ConvexNextNet.forward                      (awesome/model/convex_net.py:26-35, 205-214)
  loss_grad <- criterion(sigmoid(model(grid)), unaries).backward()  (awesome/model/path_connected_net.py:941-948)
  fit       <- the E-step inner loop incl. Adam/Adamax, clamp, ReduceLROnPlateau (path_connected_net.py:937-962)
  miou      <- MIOU(invert=True, average='binary')                  (awesome/measures/miou.py:29-48)
One leading "image" axis = independent parameter sets (the PriorCache axis, awesome/util/prior_cache.py:49-59).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------------------------------
# model description + flat parameter layout (include/inrfit.h)
# ----------------------------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class IcnnSpec:
    n_hidden: int = 130
    in_features: int = 2
    n_layers: int = 1
    act0: str = "relu"     # layer-0 activation = the encode stage: 'relu' | 'cos' (Fourier features) | 'sin' (sine layer)
    omega: float = 1.0     # 'sin' only: z0 = sin(omega (W_in x + b_in))

    def desc(self) -> L.InrModelDesc:
        return L.InrModelDesc(L.INR_MODEL_ICNN, self.n_hidden, self.in_features, self.n_layers, L.ACT_KINDS[self.act0],
                              float(self.omega))

    def keys_shapes(self) -> List[Tuple[str, Tuple[int, ...]]]:
        """state_dict keys of ConvexNextNet in flat-vector order (awesome/model/convex_net.py:188-203)."""
        h, c = self.n_hidden, self.in_features
        out = [("input.weight", (h, c)), ("input.bias", (h,))]
        for k in range(self.n_layers):
            out += [(f"skip.{k}.ln.weight", (h, h)), (f"skip.{k}.ln.bias", (h,)), (f"skip.{k}.skp.weight", (h, c))]
        out += [("out.ln.weight", (1, h)), ("out.ln.bias", (1,)), ("out.skp.weight", (1, c))]
        return out

    @property
    def n_params(self) -> int:
        n = 0
        for _, shp in self.keys_shapes():
            m = 1
            for s in shp:
                m *= s
            n += m
        return n

    def clamp_keys(self) -> List[str]:
        """Weights projected onto >= 0 by enforce_convexity (convex_net.py:151-154, 216-220)."""
        return [f"skip.{k}.ln.weight" for k in range(self.n_layers)] + ["out.ln.weight"]

    def fused(self) -> bool:
        """A fused MFMA step kernel serves this shape (n_hidden <= 130, L <= 2; other widths zero-padded) - what the composite
        priors and the fused joint step need; wider / deeper nets run layer by layer (csrc/wide.h) through inrfit_fit etc. only."""
        return self.supported() and self.n_hidden <= 130 and self.n_layers <= 2

    def supported(self) -> bool:
        d = self.desc()
        return bool(L.load().inrfit_supported(C.byref(d)))


# ConvexNet (convex_net.py:10-40) uses different key names for the same L=1 network
CONVEXNET_KEYMAP = {
    "W0y.weight": "input.weight", "W0y.bias": "input.bias",
    "W1z.weight": "skip.0.ln.weight", "W1z.bias": "skip.0.ln.bias", "W1y.weight": "skip.0.skp.weight",
    "W2z.weight": "out.ln.weight", "W2z.bias": "out.ln.bias", "W2y.weight": "out.skp.weight",
}
CONVEXNET_KEYMAP_INV = {v: k for k, v in CONVEXNET_KEYMAP.items()}


def pack_state_dict(spec: IcnnSpec, sd: Dict[str, Tensor], device=None) -> Tensor:
    """state_dict (ConvexNextNet or ConvexNet key names) -> flat fp32 vector [P]."""
    if "W0y.weight" in sd:
        sd = {CONVEXNET_KEYMAP[k]: v for k, v in sd.items()}
    parts = []
    for k, shp in spec.keys_shapes():
        t = sd[k]
        if tuple(t.shape) != shp:
            raise ValueError(f"{k}: expected shape {shp}, got {tuple(t.shape)}")
        parts.append(t.detach().reshape(-1).to(dtype=torch.float32))
    flat = torch.cat(parts)
    return flat.to(device) if device is not None else flat


def unpack_params(spec: IcnnSpec, flat: Tensor, convexnet_keys: bool = False) -> Dict[str, Tensor]:
    """flat [P] -> state_dict-shaped tensors (views into a clone)."""
    flat = flat.detach().reshape(-1).clone()
    out, off = {}, 0
    for k, shp in spec.keys_shapes():
        n = 1
        for s in shp:
            n *= s
        out[CONVEXNET_KEYMAP_INV[k] if convexnet_keys else k] = flat[off:off + n].reshape(shp)
        off += n
    return out


# ----------------------------------------------------------------------------------------------------------------------
# grid description
# ----------------------------------------------------------------------------------------------------------------------
class Grid:
    """Dense coordinate grid in HBM.  `separable`: xs[W], ys[H] (+ ts[n_images]); `explicit`: coords [n|1][C][N]."""

    def __init__(self, mode: int, n_points: int, width: int = 0, height: int = 0, xs: Optional[Tensor] = None,
                 ys: Optional[Tensor] = None, ts: Optional[Tensor] = None, coords: Optional[Tensor] = None,
                 image_stride: int = 0):
        self.mode, self.n_points, self.width, self.height = mode, n_points, width, height
        self.xs, self.ys, self.ts, self.coords, self.image_stride = xs, ys, ts, coords, image_stride

    @staticmethod
    def separable(xs: Tensor, ys: Tensor, ts: Optional[Tensor] = None) -> "Grid":
        xs = xs.detach().to(dtype=torch.float32).contiguous()
        ys = ys.detach().to(dtype=torch.float32).contiguous()
        if ts is not None:
            ts = ts.detach().to(dtype=torch.float32).contiguous()
        return Grid(L.INR_GRID_SEPARABLE, xs.numel() * ys.numel(), xs.numel(), ys.numel(), xs, ys, ts)

    @staticmethod
    def linspace(width: int, height: int, device, t_over_tmax: Optional[Tensor] = None) -> "Grid":
        """Transformator.get_positional_matrices (awesome/dataset/transformator.py:25-61): linspace(0,1,w) x linspace(0,1,h)."""
        xs = torch.linspace(0, 1, width).to(device)   # computed on CPU exactly like the reference, then moved
        ys = torch.linspace(0, 1, height).to(device)
        return Grid.separable(xs, ys, t_over_tmax)

    @staticmethod
    def howto(width: int, height: int, device) -> "Grid":
        """notebooks/how_to/convexity.ipynb cell 7 create_grid: x = i/w, y = j/h."""
        xs = (torch.arange(0, width).float() / torch.tensor(float(width))).to(device)
        ys = (torch.arange(0, height).float() / torch.tensor(float(height))).to(device)
        return Grid.separable(xs, ys)

    @staticmethod
    def explicit(coords: Tensor) -> "Grid":
        """coords: (C, N) shared by all images, or (n_images, C, N)."""
        coords = coords.detach().to(dtype=torch.float32).contiguous()
        if coords.dim() == 2:
            return Grid(L.INR_GRID_EXPLICIT, coords.shape[1], coords=coords, image_stride=0)
        if coords.dim() == 3:
            return Grid(L.INR_GRID_EXPLICIT, coords.shape[2], coords=coords, image_stride=coords.shape[1] * coords.shape[2])
        raise ValueError("coords must be (C,N) or (n_images,C,N)")

    @staticmethod
    def from_image_grid(grid: Tensor) -> "Grid":
        """(B,C,H,W) or (C,H,W) tensor as the reference passes it through @pixelize (awesome/util/pixelize.py:31-33)."""
        if grid.dim() == 3:
            c, h, w = grid.shape
            return Grid.explicit(grid.reshape(c, h * w))
        b, c, h, w = grid.shape
        return Grid.explicit(grid.reshape(b, c, h * w))

    @property
    def device(self):
        return (self.xs if self.xs is not None else self.coords).device

    def desc(self) -> L.InrGridDesc:
        p = lambda t: (t.data_ptr() if t is not None else None)
        return L.InrGridDesc(self.mode, self.width, self.height, self.n_points, p(self.xs), p(self.ys), p(self.ts),
                             p(self.coords), self.image_stride)


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _check_dev(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise L.InrfitError(f"{name} must live on the GPU (awesome_amd has no CPU path)")
    if t.dtype != torch.float32:
        raise L.InrfitError(f"{name} must be float32")
    return t.contiguous()


def _check_spec(spec: IcnnSpec) -> None:
    if not spec.supported():
        raise L.InrfitError(f"no kernel path for {spec}: n_hidden <= 1024, in_features in {{2, 3}}, 1..8 hidden layers (fused MFMA kernels "
                            f"for n_hidden <= 130 and L <= 2, zero-padded to the next compiled width; the layer-by-layer path beyond)")


def _workspace(spec: IcnnSpec, grid: Grid, n_images: int) -> Tensor:
    md, gd = spec.desc(), grid.desc()
    nbytes = L.load().inrfit_workspace_bytes(C.byref(md), C.byref(gd), n_images)
    if nbytes < 0:
        L.check(int(nbytes), "inrfit_workspace_bytes")
    return L.scratch(int(nbytes) // 4 + 1, dtype=torch.float32, device=grid.device)


# ----------------------------------------------------------------------------------------------------------------------
# entry points
# ----------------------------------------------------------------------------------------------------------------------
def forward(spec: IcnnSpec, params: Tensor, grid: Grid) -> Tensor:
    """params [n_images, P] -> logits [n_images, N]."""
    _check_spec(spec)
    params = _check_dev(params, "params")
    if params.dim() == 1:
        params = params[None]
    n_images = params.shape[0]
    assert params.shape[1] == spec.n_params, (params.shape, spec.n_params)
    logits = L.scratch(n_images, grid.n_points, dtype=torch.float32, device=params.device)
    md, gd = spec.desc(), grid.desc()
    ws = _workspace(spec, grid, n_images)
    rc = L.load().inrfit_forward(C.byref(md), params.data_ptr(), C.byref(gd), n_images, logits.data_ptr(), ws.data_ptr(),
                                 ws.numel() * 4, _stream_ptr(params.device))
    L.check(rc, "inrfit_forward")
    return logits


def _loss_desc(kind: str, weight_mode: str, ratio: float, c_fg: float, c_bg: float) -> L.InrLossDesc:
    return L.InrLossDesc(L.LOSS_KINDS[kind], L.WEIGHT_MODES[weight_mode], float(ratio), float(c_fg), float(c_bg))


def loss_grad(spec: IcnnSpec, params: Tensor, grid: Grid, targets: Tensor, loss: str = "se", weight_mode: str = "none",
              ratio: float = 1.0, c_fg: float = 0.0, c_bg: float = 0.0) -> Tuple[Tensor, Tensor]:
    """-> (loss [n_images], grads [n_images, P]) of the data term at `params`."""
    _check_spec(spec)
    params = _check_dev(params, "params")
    targets = _check_dev(targets, "targets")
    if params.dim() == 1:
        params = params[None]
    n_images = params.shape[0]
    targets = targets.reshape(n_images, -1)
    assert targets.shape[1] == grid.n_points
    ws = _workspace(spec, grid, n_images)
    loss_out = L.scratch(n_images, dtype=torch.float32, device=params.device)
    grads = L.scratch_like(params)
    md, gd, ld = spec.desc(), grid.desc(), _loss_desc(loss, weight_mode, ratio, c_fg, c_bg)
    rc = L.load().inrfit_loss_grad(C.byref(md), params.data_ptr(), C.byref(gd), targets.data_ptr(), C.byref(ld), n_images,
                                   loss_out.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                   _stream_ptr(params.device))
    L.check(rc, "inrfit_loss_grad")
    return loss_out, grads


def backward(spec: IcnnSpec, params: Tensor, grid: Grid, dlogits: Tensor, want_dcoords: bool = False):
    """Vector-Jacobian product: grads [n_images, P] = dlogits [n_images, N] . d logits / d params (forward recomputed).
    want_dcoords: also return dL/dcoords [n_images, C, N]."""
    _check_spec(spec)
    params = _check_dev(params, "params")
    dlogits = _check_dev(dlogits, "dlogits")
    if params.dim() == 1:
        params = params[None]
    n_images = params.shape[0]
    dlogits = dlogits.reshape(n_images, -1)
    assert dlogits.shape[1] == grid.n_points
    ws = _workspace(spec, grid, n_images)
    grads = L.scratch_like(params)
    dco = L.scratch(n_images, spec.in_features, grid.n_points, dtype=torch.float32, device=params.device) if want_dcoords else None
    md, gd = spec.desc(), grid.desc()
    rc = L.load().inrfit_backward(C.byref(md), params.data_ptr(), C.byref(gd), dlogits.data_ptr(), n_images,
                                  grads.data_ptr(), dco.data_ptr() if dco is not None else None, ws.data_ptr(),
                                  ws.numel() * 4, _stream_ptr(params.device))
    L.check(rc, "inrfit_backward")
    return (grads, dco) if want_dcoords else grads


def step_only(spec: IcnnSpec, params: Tensor, grid: Grid, targets: Tensor, iters: int, loss: str = "se",
              weight_mode: str = "none", workspace: Optional[Tensor] = None) -> Tensor:
    """Measurement hook: launch only the fused step kernel `iters` times on the current stream."""
    _check_spec(spec)
    params = _check_dev(params, "params")
    targets = _check_dev(targets, "targets")
    n_images = params.shape[0]
    ws = workspace if workspace is not None else _workspace(spec, grid, n_images)
    md, gd, ld = spec.desc(), grid.desc(), _loss_desc(loss, weight_mode, 1.0, 0.0, 0.0)
    rc = L.load().inrfit_step_only(C.byref(md), params.data_ptr(), C.byref(gd), targets.data_ptr(), C.byref(ld), n_images,
                                   int(iters), ws.data_ptr(), ws.numel() * 4, _stream_ptr(params.device))
    L.check(rc, "inrfit_step_only")
    return ws


def mfma_stream_tflops(device, workgroups: int = 256, iters: int = 12000) -> float:
    """Measurement hook: TFLOP/s of a launch that issues nothing but independent fp32 16x16x4 MFMAs on `workgroups` x 4 waves -
    what the matrix pipes sustain at the clock the chip holds under that load (the practical ceiling under the nominal peak)."""
    scratch = L.scratch(workgroups * 256, dtype=torch.float32, device=device)
    flop = C.c_double(0.0)
    lib = L.load()
    sp = _stream_ptr(torch.device(device))
    L.check(lib.inrfit_mfma_stream(workgroups, 2000, C.byref(flop), scratch.data_ptr(), sp), "inrfit_mfma_stream")   # warm (clocks up)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.inrfit_mfma_stream(workgroups, int(iters), C.byref(flop), scratch.data_ptr(), sp), "inrfit_mfma_stream")
    e1.record()
    torch.cuda.synchronize(device)
    return flop.value / (e0.elapsed_time(e1) * 1e-3) / 1e12


@dataclass
class FitResult:
    params: Tensor            # [n_images, P] final parameters (same storage as the input)
    opt_state: Tensor         # [n_images, 2P + 8]
    loss_hist: Optional[Tensor]   # [n_images, steps]
    logits: Optional[Tensor]      # [n_images, N]
    status: Tensor            # [n_images] int32, 0 = ok, 1 = non-finite loss seen


def new_opt_state(spec: IcnnSpec, n_images: int, device) -> Tensor:
    return torch.zeros(n_images, 2 * spec.n_params + L.INR_OPT_HEADER_FLOATS, dtype=torch.float32, device=device)


def fit(spec: IcnnSpec, params: Tensor, grid: Grid, targets: Tensor, steps: int, lr: float = 2e-3, loss: str = "se",
        weight_mode: str = "none", ratio: float = 1.0, c_fg: float = 0.0, c_bg: float = 0.0, optimizer: str = "adam",
        betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, clamp: bool = True,
        plateau: Optional[dict] = None, opt_state: Optional[Tensor] = None, step0: int = 0, record_loss: bool = True,
        want_logits: bool = True, freeze_skips: bool = False, freeze_input: bool = False, gate_logits: bool = False) -> FitResult:
    """`steps` optimisation steps of n_images independent fits on the device (params updated IN PLACE).
    gate_logits: `logits` = the output of the last training forward (what the reference's IoU gate reads) instead of the logits at
    the final parameters."""
    _check_spec(spec)
    params = _check_dev(params, "params")
    targets = _check_dev(targets, "targets")
    if params.dim() != 2:
        raise ValueError("params must be [n_images, P]")
    n_images = params.shape[0]
    targets = targets.reshape(n_images, -1)
    assert targets.shape[1] == grid.n_points
    dev = params.device
    if opt_state is None:
        opt_state = new_opt_state(spec, n_images, dev)
    ws = _workspace(spec, grid, n_images)
    hist = L.scratch(n_images, max(steps, 1), dtype=torch.float32, device=dev) if record_loss else None
    logits = L.scratch(n_images, grid.n_points, dtype=torch.float32, device=dev) if want_logits else None
    status = torch.zeros(n_images, dtype=torch.int32, device=dev)
    pl = plateau or {}
    od = L.InrOptDesc(L.OPT_KINDS[optimizer], float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
                      int(bool(clamp)), int(plateau is not None), int(pl.get("patience", 200)), float(pl.get("factor", 0.5)),
                      float(pl.get("threshold", 1e-4)), float(pl.get("min_lr", 0.0)), float(pl.get("eps", 1e-8)),
                      int(bool(freeze_skips)), int(bool(freeze_input)), int(bool(gate_logits)))
    md, gd, ld = spec.desc(), grid.desc(), _loss_desc(loss, weight_mode, ratio, c_fg, c_bg)
    rc = L.load().inrfit_fit(C.byref(md), params.data_ptr(), opt_state.data_ptr(), C.byref(gd), targets.data_ptr(),
                             C.byref(ld), C.byref(od), n_images, int(steps), int(step0),
                             hist.data_ptr() if hist is not None else None,
                             logits.data_ptr() if logits is not None else None, status.data_ptr(), ws.data_ptr(),
                             ws.numel() * 4, _stream_ptr(dev))
    L.check(rc, "inrfit_fit")
    return FitResult(params, opt_state, hist[:, :steps] if hist is not None else None, logits, status)


def miou(out: Tensor, tgt: Tensor, thr_out: float = 0.5, thr_tgt: float = 0.5, invert: bool = True) -> Tensor:
    """[n_images, N] x2 -> [n_images] binary IoU of the (inverted) class, 0 when the target has none of it."""
    out = _check_dev(out, "out")
    tgt = _check_dev(tgt, "tgt")
    n_images = out.shape[0] if out.dim() > 1 else 1
    out = out.reshape(n_images, -1)
    tgt = tgt.reshape(n_images, -1)
    res = L.scratch(n_images, dtype=torch.float32, device=out.device)
    rc = L.load().inrfit_miou(out.data_ptr(), tgt.data_ptr(), n_images, out.shape[1], float(thr_out), float(thr_tgt),
                              int(bool(invert)), res.data_ptr(), _stream_ptr(out.device))
    L.check(rc, "inrfit_miou")
    return res


class step_kernel_brackets:
    """Measurement hook (context manager): while active, `fit` and `step_only` bracket every step-kernel launch with HIP
    events (`fit` also every update-kernel launch); `.avg_us` / `.update_avg_us` / `.samples` afterwards = average elapsed
    time of a bracket (kernel + what the two event packets add)."""

    def __init__(self, max_samples: int):
        self.max_samples, self.avg_us, self.update_avg_us, self.samples = int(max_samples), 0.0, 0.0, 0

    def __enter__(self):
        L.check(L.load().inrfit_timing_begin(self.max_samples), "inrfit_timing_begin")
        return self

    def __exit__(self, *exc):
        us, uu, n = C.c_float(0.0), C.c_float(0.0), C.c_int(0)
        L.check(L.load().inrfit_timing_end(C.byref(us), C.byref(uu), C.byref(n)), "inrfit_timing_end")
        self.avg_us, self.update_avg_us, self.samples = float(us.value), float(uu.value), int(n.value)
        return False


def pack_masks(values: Tensor, threshold: float = 0.5, invert: bool = False) -> Tensor:
    """[n_images, N] floats -> [n_images, ceil(N/64)] int64 words, bit i of word w = (values[w*64+i] > threshold) (complemented
    with `invert`): the bit-packed form of the mask the reference writes after evaluation (awesome/run/functions.py:2315-2361)."""
    values = _check_dev(values, "values")
    n = values.shape[0] if values.dim() > 1 else 1
    values = values.reshape(n, -1)
    words = (values.shape[1] + 63) // 64
    bits = torch.zeros(n, words, dtype=torch.int64, device=values.device)
    rc = L.load().inrfit_pack_masks(values.data_ptr(), n, values.shape[1], float(threshold), int(bool(invert)), bits.data_ptr(),
                                    _stream_ptr(values.device))
    L.check(rc, "inrfit_pack_masks")
    return bits
