"""PathConnectedNet with the RealNVP deformation, with the reference's module surface and state_dict keys.

Reference: PathConnectedNet (awesome/model/path_connected_net.py:53-133), built by real_nvp_path_connected_net
(awesome/model/net_factory.py:124-175) as
    convex_net = ConvexNextNet(...),  flow_net = NormNet(net=PixelizeNet(network=nf.NormalizingFlow(...)), norm=MinMax),
    linear = Conv2d(C, C, 1, groups=C)
with nf = normflows==1.7.3 (not part of the reference checkout; its MaskedAffineFlow / ActNorm / MLP definitions are
restated by the HIP kernels, awesome_amd/csrc/rnvp.h - parity for this variant is UNPINNED).

The sub-modules below are the containers of the parameters (same names, shapes and nesting as the reference's
state_dict: `flow_net.net.network.flows.{2f}.{s,t}.net.{0,2}.{weight,bias}`, `flow_net.net.network.flows.{2f+1}.{s,t}`,
`flow_net.norm.{min,max,new_min,new_max}`, `linear.{weight,bias}`, `convex_net.*`).  Evaluation - forward, autograd
backward, get_deformation, the ActNorm data-dependent init and the fused fit - runs on the HIP path only.
"""
from __future__ import annotations

import math
import os
import re
import copy
from typing import Any, Dict, Optional, Tuple

import torch
import torch.nn as nn

from .. import icnn as K
from .. import rnvp as R
from .convex_net import ConvexNextNet
from .pretrainable_module import PriorFitMixin


def _hip_only(name: str):
    raise RuntimeError(f"{name} is a parameter container: it is evaluated through PathConnectedNet on the HIP path "
                       "(awesome_amd has no torch/CPU fallback)")


class MLP(nn.Module):
    """nf.nets.MLP([C, hid, C], init_zeros=True, output_fn=...): `net` = Linear, LeakyReLU(0.0), Linear[, Tanh]."""

    def __init__(self, layers, init_zeros: bool = True, output_fn: Optional[str] = None):
        super().__init__()
        mods = [nn.Linear(layers[0], layers[1]), nn.LeakyReLU(0.0), nn.Linear(layers[1], layers[2])]
        if init_zeros:
            nn.init.zeros_(mods[-1].weight)
            nn.init.zeros_(mods[-1].bias)
        if output_fn == "tanh":
            mods.append(nn.Tanh())
        self.net = nn.Sequential(*mods)

    def forward(self, x):
        _hip_only("MLP")


class MaskedAffineFlow(nn.Module):
    def __init__(self, b: torch.Tensor, t: nn.Module, s: nn.Module):
        super().__init__()
        self.register_buffer("b", b.view(1, *b.size()))
        self.add_module("s", s)
        self.add_module("t", t)

    def forward(self, z):
        _hip_only("MaskedAffineFlow")


class ActNorm(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.s = nn.Parameter(torch.zeros(channels)[None])
        self.t = nn.Parameter(torch.zeros(channels)[None])
        self.register_buffer("data_dep_init_done", torch.tensor(0.0))

    def forward(self, z):
        _hip_only("ActNorm")


class NormalizingFlow(nn.Module):
    def __init__(self, flows):
        super().__init__()
        self.flows = nn.ModuleList(flows)

    def forward(self, z):
        _hip_only("NormalizingFlow")


class PixelizeNet(nn.Module):
    def __init__(self, network: nn.Module):
        super().__init__()
        self.network = network

    def reset_parameters(self) -> None:
        # TensorUtil.reset_parameters (awesome/util/torch.py:160-194): every leaf with reset_parameters() is re-drawn - the
        # nn.Linear layers (incl. the zero-initialised last ones); ActNorm has none and keeps its state
        for m in self.network.modules():
            if isinstance(m, nn.Linear):
                m.reset_parameters()

    def forward(self, x):
        _hip_only("PixelizeNet")


class MinMax(nn.Module):
    """awesome/transforms/min_max.py:22-61 (buffers only; the transform itself is inside the HIP kernels)."""

    def __init__(self, new_min: float = -1.0, new_max: float = 1.0, dim=None):
        super().__init__()
        self.register_buffer("min", torch.zeros(1))
        self.register_buffer("max", torch.ones(1))
        self.register_buffer("new_min", torch.tensor(float(new_min)))
        self.register_buffer("new_max", torch.tensor(float(new_max)))
        self.dim = dim
        self.fitted = False

    def fit(self, x: torch.Tensor) -> None:
        dims = tuple(self.dim) if self.dim is not None and not isinstance(self.dim, int) else self.dim
        self.min = torch.amin(x, dim=dims, keepdim=True) if dims is not None else x.min()
        self.max = torch.amax(x, dim=dims, keepdim=True) if dims is not None else x.max()
        self.fitted = True


class NormNet(nn.Module):
    def __init__(self, net: nn.Module, norm: nn.Module):
        super().__init__()
        self.net = net
        self.norm = norm

    def reset_parameters(self) -> None:
        self.net.reset_parameters()

    def forward(self, x):
        _hip_only("NormNet")


def init_realnvp(channels: int, hidden_units: int = 8, n_flows: int = 6, output_fn: Optional[str] = None,
                 output_scale: Optional[float] = None, **kwargs) -> NormalizingFlow:
    """awesome/model/net_factory.py:70-114 (same creation order of the layers, so a seeded build = the reference's init)."""
    if output_scale is not None and output_fn is None:
        raise ValueError("output_scale needs an output_fn")
    masks = R.rnvp_masks(channels, n_flows)
    flows = []
    for i in range(n_flows):
        s = MLP([channels, hidden_units, channels], init_zeros=True, output_fn=output_fn)
        t = MLP([channels, hidden_units, channels], init_zeros=True, output_fn=output_fn)
        b = torch.tensor([(masks[i] >> c) & 1 for c in range(channels)], dtype=torch.uint8)
        flows += [MaskedAffineFlow(b, t, s), ActNorm(channels)]
    nf = NormalizingFlow(flows)
    nf.output_fn, nf.output_scale = output_fn, output_scale
    return nf


class _PcnFunction(torch.autograd.Function):
    """logits = ICNN(flow_net(linear(x))) on the HIP path; backward = gradients w.r.t. every parameter for a given dL/dlogits."""

    @staticmethod
    def forward(ctx, coords: torch.Tensor, ispec, rspec, n_icnn: int, *params: torch.Tensor):
        flat = lambda ps: torch.cat([p.reshape(-1) for p in ps]).to(torch.float32)[None].contiguous()  # noqa: E731
        ip, fp = flat(params[:n_icnn]), flat(params[n_icnn:])
        grid = K.Grid.explicit(coords)
        ctx.ispec, ctx.rspec, ctx.grid = ispec, rspec, grid
        ctx.shapes = [p.shape for p in params]
        ctx.save_for_backward(ip, fp)
        return R.pcn_forward(ispec, rspec, ip, fp, grid)[0]

    @staticmethod
    def backward(ctx, dlogits: torch.Tensor):
        ip, fp = ctx.saved_tensors
        _, gi, gf = R.pcn_loss_grad(ctx.ispec, ctx.rspec, ip, fp, ctx.grid, dlogits.contiguous()[None], loss="external")
        g = torch.cat([gi[0], gf[0]])
        outs, off = [], 0
        for shp in ctx.shapes:
            n = math.prod(shp) if len(shp) else 1
            outs.append(g[off:off + n].reshape(shp))
            off += n
        return (None, None, None, None, *outs)


class PathConnectedNet(nn.Module, PriorFitMixin):
    """convex_net(flow_net(linear(x)))  (path_connected_net.py:53-85).  `pretrain` / `pretrain_load_state`: PriorFitMixin
    (the reference's :472-509, 730-1019 on `inrfit_pcn_fit`)."""

    # -- PriorFitMixin engine (per-image fits of _prior_based_pretrain, :871-962) ---------------------------------------------
    def _pretrain_defaults(self):
        return dict(num_epochs=2000, lr=1e-3, flow_weight_decay=1e-5, optimizer="adamax", reuse_state=True, reuse_state_epochs=200,
                    prefit_flow_net_identity=False, prefit_flow_net_identity_lr=1e-2, prefit_flow_net_identity_weight_decay=1e-5,
                    prefit_flow_net_identity_num_epochs=100, prefit_convex_net=False, prefit_convex_net_lr=1e-3,
                    prefit_convex_net_weight_decay=0.0, prefit_convex_net_num_epochs=200, proper_prior_fit_threshold=0.5,
                    proper_prior_fit_retrys=1, zoo=None)

    def _engine_pack(self, sd):
        ispec, rspec = self._specs()
        icnn = K.pack_state_dict(ispec, {k[len("convex_net."):]: v for k, v in sd.items() if k.startswith("convex_net.")})
        return torch.cat([icnn.cpu(), R.pack_rnvp_state_dict(rspec, sd).cpu()])

    def _engine_unpack(self, flat):
        ispec, rspec = self._specs()
        P = ispec.n_params
        out = {"convex_net." + k: v for k, v in K.unpack_params(ispec, flat[:P]).items()}
        out.update(R.unpack_rnvp_params(rspec, flat[P:]))
        return out

    def _engine_after_fit(self, sd):
        for k in sd:
            if k.endswith("data_dep_init_done"):
                sd[k] = torch.ones_like(sd[k])      # ActNorm's data-dependent init has happened on this image

    def _engine_retry_state(self, fresh, failed):
        # TensorUtil.reset_parameters (awesome/util/torch.py:160-194) re-draws modules that have reset_parameters(): the Linear
        # layers and the ICNN.  nf.flows.ActNorm has none: its s, t and data_dep_init_done survive the reset (:975-978).
        out = dict(fresh)
        for k, v in failed.items():
            if k.endswith("data_dep_init_done") or (k.startswith("flow_net.") and re.search(r"flows\.\d+\.(s|t)$", k)):
                out[k] = v.detach().clone().to(fresh[k].device if k in fresh else v.device)
        return out

    def _engine_fit(self, grid, unaries, flat, epochs, cold, opts, states=None):
        from ..measures import criterion_to_desc
        ispec, rspec = self._specs()
        P = ispec.n_params
        ip, fp = flat[:, :P].contiguous(), flat[:, P:].contiguous()
        n = ip.shape[0]
        # ActNorm initialises itself on the first batch it sees (nf.flows.ActNorm): per image, unless its state says it has.
        # A warm start (`states is None`: the previous frame's fitted state, :867-870 load_state_dict incl. data_dep_init_done = 1)
        # never re-initialises: the 200-epoch refit continues from the previous deformation.
        need = []
        if states is not None:
            need = [j for j in range(n) if not all(float(v) > 0 for k, v in states[j].items() if k.endswith("data_dep_init_done"))]
        if need:
            sub = fp[need].contiguous()
            g = grid if grid.coords is None or grid.coords.dim() == 2 else K.Grid.explicit(grid.coords[need].contiguous())
            R.actnorm_init(rspec, sub, g)
            fp[need] = sub
        prefit = cold and opts.get("_prefit", True)   # once per cold image, not again on a retry (:871-894 sit before the retry loop)
        if prefit and opts.get("prefit_flow_net_identity", False):
            zoo = opts.get("zoo")
            kw = dict(lr=float(opts.get("prefit_flow_net_identity_lr", 1e-2)),
                      weight_decay=float(opts.get("prefit_flow_net_identity_weight_decay", 1e-5)))
            steps = int(opts.get("prefit_flow_net_identity_num_epochs", 100))
            if zoo is None:
                R.fit_identity(rspec, fp, grid, steps=steps, **kw)
            else:
                # the reference stores the first identity fit of an (architecture, grid, hyper-parameter) combination in the zoo and
                # LOADS it for every later image (:177-194, 246-248): all images then share that flow
                keep = copy.deepcopy(self.state_dict())
                self.load_flat(ip[0], fp[0])
                for m in self.flow_net.net.network.flows:
                    if hasattr(m, "data_dep_init_done"):
                        m.data_dep_init_done.fill_(1.0)
                g0 = grid if grid.coords is None or grid.coords.dim() == 2 else K.Grid.explicit(grid.coords[0].contiguous())
                self.learn_flow_identity(g0, max_iter=steps, zoo=zoo, **kw)
                _, _, _, flow = self._ordered_params()
                shared = self._flat(flow).to(fp.device)[0]
                lin = 2 * rspec.channels
                fp[:, lin:] = shared[lin:]
                self.load_state_dict(keep)
        if prefit and opts.get("prefit_convex_net", False):
            xd = R.rnvp_forward(rspec, fp, grid)
            K.fit(ispec, ip, K.Grid.explicit(xd), unaries, int(opts.get("prefit_convex_net_num_epochs", 200)),
                  lr=float(opts.get("prefit_convex_net_lr", 1e-3)), loss="se", optimizer="adam",
                  weight_decay=float(opts.get("prefit_convex_net_weight_decay", 0.0)), plateau=None, record_loss=False,
                  want_logits=False)
        crit = opts.get("criterion")
        kind, wmode, ratio = criterion_to_desc(crit, "targets") if crit is not None else ("se", "none", 1.0)
        res = R.pcn_fit(ispec, rspec, ip, fp, grid, unaries, epochs, lr=float(opts.get("lr", 1e-3)),
                        optimizer=opts.get("optimizer", "adamax"), loss=kind, weight_mode=wmode, ratio=ratio,
                        flow_weight_decay=float(opts.get("flow_weight_decay", 1e-5)),
                        plateau=dict(patience=200, factor=0.5) if opts.get("use_plateau", True) else None,
                        record_loss=False, want_logits=True, gate_logits=True)
        return torch.cat([res.icnn_params, res.flow_params], 1), res.logits, res.status

    def _non_prior_based_pretrain(self, train_set, test_set, device, agent, use_progress_bar: bool = True,
                                  wrapper_module=None, **kwargs):
        """Spatio-temporal pretraining (path_connected_net.py:511-728): the data set has no per-image priors - ONE network
        is fitted to all frames, whose clean grids carry t / t_max as a third channel.  Unaries of every frame from the
        segmentation module (prior off), optional pre-fits (flow towards the identity on the frames' grid, convex net on the
        deformed grid), then `num_epochs` epochs of mini-batches of `batch_size` frames (fit_sequence).  Returns
        self.state_dict() like the reference (:721)."""
        if wrapper_module is None:
            raise ValueError("Wrapper model must be provided for pretraining.")
        from .pretrainable_module import decompose_training_item
        opts = dict(self._pretrain_defaults(), batch_size=1, dataloader_shuffle=False)
        opts.update(kwargs)
        ds = agent.training_dataset
        device = torch.device(device)
        was, training_state = getattr(wrapper_module, "evaluate_prior", True), wrapper_module.training
        coords, unaries = [], []
        try:
            wrapper_module.eval()
            wrapper_module.evaluate_prior = False
            for pos in range(len(train_set)):
                inputs, _, _, _ = decompose_training_item(train_set[pos], ds)
                dev_in = [t.to(device)[None] if isinstance(t, torch.Tensor) else t for t in inputs]
                with torch.no_grad():
                    un = wrapper_module(*dev_in)
                    pargs, _ = wrapper_module.get_prior_args(dev_in[0], *dev_in[1:], segm=un[0, ...])
                g = pargs[0] if pargs[0].dim() == 4 else pargs[0][None]
                coords.append(g[0].reshape(g.shape[1], -1).to(torch.float32))
                unaries.append(un.reshape(-1).to(torch.float32))
        finally:
            wrapper_module.evaluate_prior = was
            wrapper_module.train(training_state)
        frame_coords, frame_unaries = torch.stack(coords), torch.stack(unaries)          # (T, C, HW), (T, HW)
        T = frame_coords.shape[0]
        whole = K.Grid.explicit(frame_coords.permute(1, 0, 2).reshape(frame_coords.shape[1], -1).contiguous())
        if opts.get("prefit_flow_net_identity", False):
            self.learn_flow_identity(whole, lr=float(opts["prefit_flow_net_identity_lr"]),
                                     weight_decay=float(opts["prefit_flow_net_identity_weight_decay"]),
                                     max_iter=int(opts["prefit_flow_net_identity_num_epochs"]), zoo=opts.get("zoo"))
        if opts.get("prefit_convex_net", False):
            self.learn_convex_net(whole, frame_unaries.reshape(1, -1), lr=float(opts["prefit_convex_net_lr"]),
                                  weight_decay=float(opts["prefit_convex_net_weight_decay"]),
                                  max_iter=int(opts["prefit_convex_net_num_epochs"]))
        from ..measures import criterion_targets, criterion_to_desc
        crit = opts.get("criterion")
        kind, wmode, _ = criterion_to_desc(crit, "targets") if crit is not None else ("se", "none", 1.0)
        frame_unaries = self._sequence_unaries(frame_unaries, opts, agent)
        # criterion_to_desc unwraps UnariesConversionLoss; its effect - binarised targets - is applied here
        frame_unaries = criterion_targets(crit, frame_unaries).contiguous()
        self.pretrain_epoch_losses = self.fit_sequence(frame_coords, frame_unaries, num_epochs=int(opts["num_epochs"]),
                                                       lr=float(opts["lr"]), flow_weight_decay=float(opts["flow_weight_decay"]),
                                                       batch_size=min(int(opts["batch_size"]), T),
                                                       dataloader_shuffle=bool(opts["dataloader_shuffle"]), loss=kind,
                                                       weight_mode=wmode, optimizer=opts.get("optimizer", "adamax"))
        for m in self.flow_net.net.network.flows:
            if hasattr(m, "data_dep_init_done"):
                m.data_dep_init_done.fill_(1.0)
        return self.state_dict()

    def _sequence_unaries(self, frame_unaries: torch.Tensor, opts: Dict[str, Any], agent=None) -> torch.Tensor:
        """The unaries (T, HW) the mini-batch epochs train on; hook of NoisyPathConnectedNet."""
        return frame_unaries

    def __init__(self, convex_net: ConvexNextNet, flow_net: NormNet, in_channels: int = 2, **kwargs):
        super().__init__()
        self.convex_net = convex_net
        self.flow_net = flow_net
        self.in_channels = in_channels
        self.linear = nn.Conv2d(in_channels, in_channels, 1, groups=in_channels)   # per-channel scale + translation
        self._init_linear()

    def _init_linear(self) -> None:
        self.linear.weight.data.fill_(1)
        self.linear.bias.data.fill_(0)

    def reset_parameters(self) -> None:
        self.convex_net.reset_parameters()
        self.flow_net.reset_parameters()
        self._init_linear()

    def enforce_convexity(self) -> None:
        self.convex_net.enforce_convexity()

    # ---- flat views ---------------------------------------------------------------------------------------------------
    def _specs(self) -> Tuple[K.IcnnSpec, R.RnvpSpec]:
        nf = self.flow_net.net.network
        norm = self.flow_net.norm
        C = self.in_channels
        lin0 = nf.flows[0].s.net[0]
        vmin = tuple(float(v) for v in norm.min.reshape(-1).expand(C).tolist()) if norm.min.numel() in (1, C) else None
        vmax = tuple(float(v) for v in norm.max.reshape(-1).expand(C).tolist()) if norm.max.numel() in (1, C) else None
        if vmin is None or vmax is None:
            raise ValueError("MinMax must be fitted per channel (dim=(0, 2, 3)) or globally")
        masks = tuple(int(sum(int(nf.flows[2 * f].b.reshape(-1)[c]) << c for c in range(C))) for f in range(len(nf.flows) // 2))
        rspec = R.RnvpSpec(C, lin0.out_features, len(nf.flows) // 2, getattr(nf, "output_fn", None),
                           getattr(nf, "output_scale", None), vmin, vmax, float(norm.new_min), float(norm.new_max), masks)
        return self.convex_net.spec, rspec

    def _ordered_params(self):
        ispec, rspec = self._specs()
        sd = dict(self.named_parameters())
        icnn = [sd["convex_net." + k] for k, _ in ispec.keys_shapes()]
        flow = [sd[k] for k, _ in rspec.keys_shapes()]
        return ispec, rspec, icnn, flow

    def _flat(self, ps) -> torch.Tensor:
        return torch.cat([p.detach().reshape(-1) for p in ps]).to(torch.float32)[None].contiguous()

    def _actnorm_init_if_needed(self, coords: torch.Tensor) -> None:
        """nf.flows.ActNorm's data-dependent init happens inside its first forward; here: one HIP pass over the grid."""
        acts = [m for m in self.flow_net.net.network.flows if isinstance(m, ActNorm)]
        if all(float(a.data_dep_init_done) > 0 for a in acts):
            return
        ispec, rspec, _, flow = self._ordered_params()
        fp = self._flat(flow)
        R.actnorm_init(rspec, fp, K.Grid.explicit(coords))
        new = R.unpack_rnvp_params(rspec, fp[0])
        sd = dict(self.named_parameters())
        with torch.no_grad():
            for f, a in enumerate(acts):
                if float(a.data_dep_init_done) > 0:
                    continue
                for nm in ("s", "t"):
                    k = f"flow_net.net.network.flows.{2 * f + 1}.{nm}"
                    sd[k].copy_(new[k].to(sd[k].device))
                a.data_dep_init_done.fill_(1.0)

    @staticmethod
    def _planar(x: torch.Tensor):
        """(B,C,H,W) -> list of (C, H*W); (N,C) -> [(C, N)]."""
        if x.dim() == 4:
            b, c, h, w = x.shape
            return [x[i].reshape(c, h * w) for i in range(b)], (b, h, w)
        return [x.t().contiguous()], None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B,C,H,W) -> (B,1,H,W) or (N,C) -> (N,1), forward and backward on the HIP path."""
        if not x.is_cuda:
            raise RuntimeError("awesome_amd modules run on the MI355X only (no CPU fallback); move module and input to cuda")
        planes, bhw = self._planar(x)
        # the reference's ActNorm sees the whole pixelized batch at its first forward (pixelize_net.py:15-19)
        self._actnorm_init_if_needed(torch.cat(planes, dim=1))
        ispec, rspec, icnn, flow = self._ordered_params()
        run = lambda coords: _PcnFunction.apply(coords, ispec, rspec, len(icnn), *icnn, *flow)  # noqa: E731
        if bhw is not None:
            b, h, w = bhw
            return torch.stack([run(p).reshape(1, h, w) for p in planes], 0)
        return run(planes[0])[:, None]

    def get_deformation(self, x: torch.Tensor) -> torch.Tensor:
        """path_connected_net.py:124-128: flow_net(linear(x)) in the layout of x (no autograd)."""
        planes, bhw = self._planar(x)
        self._actnorm_init_if_needed(torch.cat(planes, dim=1))
        _, rspec, _, flow = self._ordered_params()
        fp = self._flat(flow)
        outs = [R.rnvp_forward(rspec, fp, K.Grid.explicit(p))[0] for p in planes]
        if bhw is not None:
            b, h, w = bhw
            return torch.stack([o.reshape(-1, h, w) for o in outs], 0)
        return outs[0].t()

    def inverse(self, x: torch.Tensor) -> torch.Tensor:
        """path_connected_net.py:107-122: the inverse of get_deformation, in the layout of x (no autograd)."""
        planes, bhw = self._planar(x)
        _, rspec, _, flow = self._ordered_params()
        fp = self._flat(flow)
        outs = [R.rnvp_inverse(rspec, fp, p.contiguous())[0] for p in planes]
        if bhw is not None:
            b, h, w = bhw
            return torch.stack([o.reshape(-1, h, w) for o in outs], 0)
        return outs[0].t()

    @staticmethod
    def _first_image_coords(grid: "K.Grid") -> torch.Tensor:
        if grid.coords is not None:
            return grid.coords if grid.coords.dim() == 2 else grid.coords[0]
        xs, ys = grid.xs, grid.ys
        chans = [xs[None, :].expand(ys.numel(), -1).reshape(-1), ys[:, None].expand(-1, xs.numel()).reshape(-1)]
        if grid.ts is not None:
            chans.append(grid.ts[0].expand(chans[0].numel()))
        return torch.stack(chans, 0)

    def learn_flow_identity(self, grid: "K.Grid", lr: float = 1e-2, weight_decay: float = 1e-5, max_iter: int = 1000,
                            zoo=None) -> torch.Tensor:
        """path_connected_net.py:155-250 on the HIP path: Adamax on the flow_net alone towards flow_net(x) = x; updates this
        module's flow parameters in place and returns the loss history.  `zoo` (awesome_amd.model.zoo.Zoo): like the reference
        (:177-194, :246-248) a stored flow for this architecture, grid and hyper-parameters is loaded instead of fitted."""
        zoo_config = None
        if zoo is not None:
            from .zoo import tensor_hash
            zoo_config = dict(lr=lr, weight_decay=weight_decay, max_iter=max_iter, criterion="SE(mean)",
                              x_data=tensor_hash(self._first_image_coords(grid)))
            loaded, context = zoo.load_model_state("flow_identity", self.flow_net, config=zoo_config)
            if loaded:
                return context.get("loss_hist", None)
        self._actnorm_init_if_needed(self._first_image_coords(grid))
        _, rspec, _, flow = self._ordered_params()
        fp = self._flat(flow)
        hist, _ = R.fit_identity(rspec, fp, grid, steps=max_iter, lr=lr, weight_decay=weight_decay)
        new = R.unpack_rnvp_params(rspec, fp[0])
        own = dict(self.named_parameters())
        with torch.no_grad():
            for k, v in new.items():
                if not k.startswith("linear."):
                    own[k].copy_(v.to(own[k].device))
        if zoo is not None:
            zoo.save_model_state("flow_identity", self.flow_net, config=zoo_config, context=dict(loss_hist=hist[0].cpu()))
        return hist[0]

    def learn_convex_net(self, grid: "K.Grid", unaries: torch.Tensor, lr: float = 1e-3, weight_decay: float = 0.0,
                         max_iter: int = 1000, icnn_params: Optional[torch.Tensor] = None):
        """path_connected_net.py:307-390 (mode 'unaries', use_deformed_grid): Adam on the convex_net alone on the grid deformed by
        the current flow, SE('mean') on sigmoid(logits) vs. unaries [n_images, N], clamp after every step.  Returns the FitResult
        (flat ICNN parameters per image; with one image they are also written back into the module)."""
        self._actnorm_init_if_needed(self._first_image_coords(grid))
        ispec, rspec, icnn, flow = self._ordered_params()
        n, dev = unaries.shape[0], unaries.device
        xd = R.rnvp_forward(rspec, self._flat(flow).to(dev), grid)[0]
        ip = icnn_params if icnn_params is not None else self._flat(icnn).repeat(n, 1).contiguous().to(dev)
        res = K.fit(ispec, ip, K.Grid.explicit(xd), unaries, max_iter, lr=lr, loss="se", optimizer="adam", weight_decay=weight_decay,
                    plateau=None, want_logits=False)
        if n == 1 and icnn_params is None:
            own = dict(self.named_parameters())
            with torch.no_grad():
                for k, v in K.unpack_params(ispec, res.params[0]).items():
                    own["convex_net." + k].copy_(v.to(own["convex_net." + k].device))
        return res

    def fit_images(self, grid: "K.Grid", unaries: torch.Tensor, num_epochs: int = 2000, lr: float = 1e-3,
                   flow_weight_decay: float = 1e-5, loss: str = "se", weight_mode: str = "none", optimizer: str = "adamax",
                   plateau=None, prefit_flow_net_identity: bool = False, prefit_flow_net_identity_lr: float = 1e-2,
                   prefit_flow_net_identity_weight_decay: float = 1e-5, prefit_flow_net_identity_num_epochs: int = 100,
                   prefit_convex_net: bool = False, prefit_convex_net_lr: float = 1e-3, prefit_convex_net_weight_decay: float = 0.0,
                   prefit_convex_net_num_epochs: int = 200, zoo=None):
        """Fused device-resident form of _prior_based_pretrain (path_connected_net.py:871-962) for a batch of images: the
        optional pre-fit stages (flow towards the identity - it only depends on the grid, so it runs once on this module -
        then the convex net of every image on the deformed grid), then the joint inner loop.  Every image starts from this
        module's current parameters (ActNorm initialised on the first image's grid if it is not yet); returns the PcnFitResult
        (flat parameters per image)."""
        n, dev = unaries.shape[0], unaries.device
        self._actnorm_init_if_needed(self._first_image_coords(grid))
        if prefit_flow_net_identity:
            self.learn_flow_identity(grid, lr=prefit_flow_net_identity_lr, weight_decay=prefit_flow_net_identity_weight_decay,
                                     max_iter=prefit_flow_net_identity_num_epochs, zoo=zoo)   # `zoo`: kwargs.get("zoo"), :560
        ispec, rspec, icnn, flow = self._ordered_params()
        ip = self._flat(icnn).repeat(n, 1).contiguous().to(dev)
        fp = self._flat(flow).repeat(n, 1).contiguous().to(dev)
        if prefit_convex_net:
            ip = self.learn_convex_net(grid, unaries, lr=prefit_convex_net_lr, weight_decay=prefit_convex_net_weight_decay,
                                       max_iter=prefit_convex_net_num_epochs, icnn_params=ip).params
        return R.pcn_fit(ispec, rspec, ip, fp, grid, unaries, num_epochs, lr=lr, optimizer=optimizer, loss=loss,
                         weight_mode=weight_mode, flow_weight_decay=flow_weight_decay,
                         plateau=dict(patience=200, factor=0.5) if plateau is None else (plateau or None))

    def fit_frames(self, grid: "K.Grid", frame_unaries: torch.Tensor, num_epochs: int = 2000, reuse_state: bool = True,
                   reuse_state_epochs: int = 200, proper_prior_fit_threshold: float = 0.5, proper_prior_fit_retrys: int = 1,
                   model_factory=None, **fit_kwargs):
        """_prior_based_pretrain over the frames of ONE sequence with one prior per frame (path_connected_net.py:803-1001): frame
        0 is fitted from this module's state for `num_epochs` (incl. the pre-fit stages if requested in fit_kwargs); with
        `reuse_state` every later frame starts from the previous properly fitted frame's parameters and trains
        `reuse_state_epochs` (:867-870, 899-908); a frame whose fg-IoU against its unaries stays below the threshold is
        re-fitted from a fresh model for `num_epochs`, up to `proper_prior_fit_retrys` times (:964-985; the fresh model comes
        from `model_factory`, default: a new real_nvp_path_connected_net with this module's sizes).  frame_unaries [T, N] on the
        device.  Returns (icnn_params [T, P], flow_params [T, RP], iou [T], retries)."""
        from .. import icnn as KK
        T, dev = frame_unaries.shape[0], frame_unaries.device
        ispec, rspec = self._specs()
        if model_factory is None:
            nf = self.flow_net.net.network
            model_factory = lambda: real_nvp_path_connected_net(  # noqa: E731
                channels=self.in_channels, hidden_units=rspec.hidden_units, flow_n_flows=rspec.n_flows,
                flow_output_fn=getattr(nf, "output_fn", None), flow_output_scale=getattr(nf, "output_scale", None),
                convex_net_hidden_units=ispec.n_hidden, convex_net_hidden_layers=ispec.n_layers).to(dev)
        prefit = {k: v for k, v in fit_kwargs.items() if k.startswith("prefit_")}
        plain = {k: v for k, v in fit_kwargs.items() if not k.startswith("prefit_") and k != "zoo"}   # pcn_fit's own kwargs only
        out_i, out_f, ious, retries = [], [], [], []
        prev = None
        for t in range(T):
            un = frame_unaries[t:t + 1].contiguous()
            if reuse_state and prev is not None:
                res = R.pcn_fit(ispec, rspec, prev[0].clone(), prev[1].clone(), grid, un, reuse_state_epochs,
                                plateau=dict(patience=200, factor=0.5), **{k: v for k, v in plain.items() if k != "plateau"})
            else:
                res = self.fit_images(grid, un, num_epochs=num_epochs, **fit_kwargs)
            iou = float(KK.miou(torch.sigmoid(res.logits), un)[0])
            n_retry = 0
            while iou < proper_prior_fit_threshold and n_retry < proper_prior_fit_retrys:
                res = model_factory().fit_images(grid, un, num_epochs=num_epochs, **plain, **prefit, zoo=fit_kwargs.get("zoo"))
                iou = float(KK.miou(torch.sigmoid(res.logits), un)[0])
                n_retry += 1
            if iou >= proper_prior_fit_threshold:
                prev = (res.icnn_params, res.flow_params)   # only a proper fit is handed on (:987-994)
            out_i.append(res.icnn_params[0])
            out_f.append(res.flow_params[0])
            ious.append(iou)
            retries.append(n_retry)
        return torch.stack(out_i), torch.stack(out_f), torch.tensor(ious), retries

    def fit_sequence(self, frame_coords: torch.Tensor, frame_unaries: torch.Tensor, num_epochs: int = 2000, lr: float = 1e-3,
                     flow_weight_decay: float = 1e-5, batch_size: int = 1, dataloader_shuffle: bool = False, loss: str = "se",
                     weight_mode: str = "none", optimizer: str = "adamax", plateau=None, generator=None):
        """_non_prior_based_pretrain (path_connected_net.py:511-728): ONE network over all frames of a sequence, trained in
        mini-batches of `batch_size` frames - one optimizer step per batch on the mean loss of the batch's pixels, Adamax over the
        same parameter groups, enforce_convexity after every step, ReduceLROnPlateau stepped once per EPOCH with the mean of the
        batch losses (:700-713).  frame_coords (T, C, H*W), frame_unaries (T, H*W).  Every step is one `inrfit_pcn_fit(steps=1)`
        call that continues the optimizer state; the epoch loss and the learning rate live on the host.  With
        batch_size == T this is `fit_images` on the whole (x, y, t) grid (full batch), which is the fast form.
        Updates this module's parameters in place; returns the list of epoch losses."""
        T, dev = frame_coords.shape[0], frame_coords.device
        self._actnorm_init_if_needed(frame_coords[:batch_size].permute(1, 0, 2).reshape(frame_coords.shape[1], -1))
        ispec, rspec, icnn, flow = self._ordered_params()
        ip, fp = self._flat(icnn).to(dev), self._flat(flow).to(dev)
        iopt = K.new_opt_state(ispec, 1, dev)
        fopt = torch.zeros(1, 2 * rspec.n_params, dtype=torch.float32, device=dev)
        pl = dict(patience=200, factor=0.5) if plateau is None else (plateau or None)
        cur_lr, best, num_bad = float(lr), float("inf"), 0
        epoch_losses, k = [], 0
        for _ in range(num_epochs):
            order = torch.randperm(T, generator=generator).tolist() if dataloader_shuffle else list(range(T))
            batches = [order[i:i + batch_size] for i in range(0, T, batch_size)]
            acc = torch.zeros((), device=dev)
            for b in batches:
                coords = frame_coords[b].permute(1, 0, 2).reshape(frame_coords.shape[1], -1).contiguous()
                un = frame_unaries[b].reshape(1, -1).contiguous()
                if k > 0:
                    iopt[:, 2 * ispec.n_params + 2] = cur_lr   # header[2]: the learning rate a continued fit starts from
                res = R.pcn_fit(ispec, rspec, ip, fp, K.Grid.explicit(coords), un, 1, lr=cur_lr, optimizer=optimizer, loss=loss,
                                weight_mode=weight_mode, flow_weight_decay=flow_weight_decay, plateau=None, icnn_opt_state=iopt,
                                flow_opt_state=fopt, step0=k, want_logits=False)
                acc = acc + res.loss_hist[0, 0] / len(batches)
                k += 1
            el = float(acc)
            epoch_losses.append(el)
            if pl is not None:   # torch ReduceLROnPlateau(mode='min', threshold=1e-4 rel), stepped with the epoch loss
                if el < best * (1.0 - pl.get("threshold", 1e-4)):
                    best, num_bad = el, 0
                else:
                    num_bad += 1
                if num_bad > pl.get("patience", 200):
                    new_lr = max(cur_lr * pl.get("factor", 0.5), pl.get("min_lr", 0.0))
                    if cur_lr - new_lr > pl.get("eps", 1e-8):
                        cur_lr = new_lr
                    num_bad = 0
        self.load_flat(ip[0], fp[0])
        return epoch_losses

    def load_flat(self, icnn_flat: torch.Tensor, flow_flat: torch.Tensor) -> None:
        """Write one row of a PcnFitResult back into the module (the PriorCache round trip)."""
        ispec, rspec = self._specs()
        sd = {"convex_net." + k: v for k, v in K.unpack_params(ispec, icnn_flat).items()}
        sd.update(R.unpack_rnvp_params(rspec, flow_flat))
        own = dict(self.named_parameters())
        with torch.no_grad():
            for k, v in sd.items():
                own[k].copy_(v.to(own[k].device))


class NoisyPathConnectedNet(PathConnectedNet):
    """awesome/model/noisy_path_connected_net.py:35-280 - the noisy spatio-temporal demonstration (`network_type` of the 21
    `config/path-connectedness/noisy-spatio-temporal` YAMLs): the spatio-temporal pretrain with the unaries of
    `round(T * noisy_percentage)` frames (never the first or the last one, :141-143) replaced by `clamp(randn + 0.5, 0, 1)`, the same
    noise in every epoch (:178-191); the replaced frames and their noise are saved next to the agent's outputs (:236-237)."""

    def _sequence_unaries(self, frame_unaries: torch.Tensor, opts: Dict[str, Any], agent=None) -> torch.Tensor:
        import numpy as np
        T = frame_unaries.shape[0]
        pct = float(opts.get("noisy_percentage", 0.333))
        candidates = np.arange(1, T - 1)
        n = int(round(T * pct))
        if n > len(candidates):
            raise ValueError(f"noisy_percentage {pct} asks for {n} noisy frames of {T} (the first and the last are never replaced)")
        idx = sorted(np.random.choice(candidates, size=n, replace=False).tolist())
        out = frame_unaries.clone()
        noisy = {}
        for k in idx:
            noisy[k] = torch.clamp(torch.randn_like(out[k]) + 0.5, 0.0, 1.0)
            out[k] = noisy[k]
        self.noisy_unaries_dict = noisy
        folder = getattr(agent, "agent_folder", None)
        if folder:
            os.makedirs(folder, exist_ok=True)
            torch.save({k: v.cpu() for k, v in noisy.items()}, os.path.join(folder, "noisy_unaries_dict.pth"))
        return out


def real_nvp_path_connected_net(channels: int = 2, hidden_units: int = 130, flow_n_flows: int = 6,
                                flow_output_fn: Optional[str] = None, flow_output_scale: Optional[float] = None,
                                norm: str = "minmax", spatial_shape: tuple = (1000, 1000), convex_net_hidden_units: int = 130,
                                convex_net_hidden_layers: int = 2, dtype: torch.dtype = torch.float32,
                                network_type: Optional[type] = None, network_args: Optional[Dict[str, Any]] = None,
                                **kwargs) -> PathConnectedNet:
    """awesome/model/net_factory.py:124-175.  The MinMax is fitted on create_normalized_grid (values in [0, 1] per channel,
    path_connected_net.py:273-296), i.e. min = 0, max = 1 per channel whatever spatial_shape is."""
    if norm != "minmax":
        raise ValueError("only norm='minmax' is built (every reference config uses it)")
    flow_net = init_realnvp(channels=channels, hidden_units=hidden_units, output_fn=flow_output_fn,
                            output_scale=flow_output_scale, n_flows=flow_n_flows)
    mm = MinMax(dim=(0, 2, 3))
    mm.min = torch.zeros(1, channels, 1, 1)
    mm.max = torch.ones(1, channels, 1, 1)
    mm.fitted = True
    norm_flow = NormNet(net=PixelizeNet(flow_net), norm=mm)
    if network_type is None:
        network_type = PathConnectedNet
    if not (isinstance(network_type, type) and issubclass(network_type, PathConnectedNet)):
        raise TypeError(f"network_type must be a PathConnectedNet class of this build, got {network_type!r}")
    if dtype != torch.float32:
        raise ValueError("the kernels compute in float32 (every reference config: dtype torch.float32)")
    return network_type(convex_net=ConvexNextNet(n_hidden=convex_net_hidden_units, n_hidden_layers=convex_net_hidden_layers,
                                                 in_features=channels),
                        flow_net=norm_flow, in_channels=channels, **(network_args or {}))
