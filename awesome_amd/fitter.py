"""BatchedPriorFitter - the device-resident form of the reference's per-image prior fit.

Reference: PathConnectedNet._prior_based_pretrain (awesome/model/path_connected_net.py:730-1007): for every image,
E full-batch steps of {forward, UnariesWeightedLoss(SE), backward, Adamax step, clamp, ReduceLROnPlateau}, an IoU gate
with parameter reset + retry (:964-985), an optional warm start from the previous frame's fitted state (`reuse_state`,
:867-870) and the result stored per image in the PriorCache (:1001).  Here all images of a batch are fitted at once
on the device (one `inrfit_fit` call); the host only handles the gate/retry decisions and the cache bookkeeping.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import torch

from . import icnn as K
from .measures import criterion_targets, criterion_to_desc


class NonFiniteLossError(ValueError):
    """A fit hit a NaN/Inf loss.  The reference raises ValueError("Loss is nan or inf!") out of the per-image loop
    (awesome/model/path_connected_net.py:232,374); `.images` lists the positions in the batch, `.report` is the FitReport with
    the other images' results (the bad images keep the parameters they had before the bad step)."""

    def __init__(self, images, report):
        super().__init__(f"Loss is nan or inf! (images {list(images)} of the batch)")
        self.images, self.report = list(images), report


@dataclass
class FitReport:
    params: torch.Tensor              # [n_images, P] fitted flat parameters (device)
    iou: torch.Tensor                 # [n_images] fg-IoU of (prior > .5) vs (unaries > .5) - the reference's gate metric
    final_loss: torch.Tensor          # [n_images]
    retries: List[int]                # retries used per image
    skipped: List[bool]               # unaries all-fg or all-bg -> not fitted (path_connected_net.py:848-855)
    status: torch.Tensor              # [n_images] int32 (1 = non-finite loss seen)
    logits: Optional[torch.Tensor] = None


class BatchedPriorFitter:
    def __init__(self, model_factory: Callable[[], torch.nn.Module], num_epochs: int = 2000, lr: float = 1e-3,
                 optimizer: str = "adamax", weight_decay: float = 0.0, criterion=None, plateau: Optional[dict] = None,
                 proper_prior_fit_threshold: float = 0.5, proper_prior_fit_retrys: int = 1, reuse_state: bool = False,
                 reuse_state_epochs: int = 200, betas=(0.9, 0.999), eps: float = 1e-8, on_nonfinite: str = "raise"):
        """Defaults follow _prior_based_pretrain's kwargs (path_connected_net.py:756-790): Adamax lr 1e-3,
        ReduceLROnPlateau(patience=200, factor=0.5), UnariesWeightedLoss(SE('mean')), threshold 0.5, 1 retry."""
        self.model_factory = model_factory
        self.num_epochs, self.lr, self.optimizer, self.weight_decay = num_epochs, lr, optimizer, weight_decay
        self.plateau = dict(patience=200, factor=0.5) if plateau is None else (plateau or None)
        self.threshold, self.retrys = proper_prior_fit_threshold, proper_prior_fit_retrys
        self.reuse_state, self.reuse_state_epochs = reuse_state, reuse_state_epochs
        self.betas, self.eps = betas, eps
        if on_nonfinite not in ("raise", "report"):
            raise ValueError("on_nonfinite must be 'raise' (the reference's behaviour) or 'report' (status only)")
        self.on_nonfinite = on_nonfinite
        if criterion is None:
            self.loss_kind, self.weight_mode, self.ratio = "se", "none", 1.0
        else:
            self.loss_kind, self.weight_mode, self.ratio = criterion_to_desc(criterion, "targets")   # _run converts the targets
        self.criterion = criterion
        probe = model_factory()
        self.spec: K.IcnnSpec = probe.spec
        self._convexnet_keys = hasattr(probe, "W0y")
        self._fit_options = dict(getattr(probe, "fit_options", None) or dict(clamp=True))   # FCNet: no clamp, skips frozen at 0

    # -- helpers --------------------------------------------------------------------------------------------------
    def fresh_params(self, n: int, device) -> torch.Tensor:
        """n independently initialised parameter sets (what PriorCache.generate_prior does, prior_cache.py:29-32)."""
        return torch.stack([self.model_factory().flat_parameters() for _ in range(n)]).to(device)

    def _run(self, params, grid, unaries, epochs):
        unaries = criterion_targets(self.criterion, unaries).contiguous()      # UnariesConversionLoss: binarised targets
        return K.fit(self.spec, params, grid, unaries, epochs, lr=self.lr, loss=self.loss_kind, weight_mode=self.weight_mode,
                     ratio=self.ratio, optimizer=self.optimizer, betas=self.betas, eps=self.eps,
                     weight_decay=self.weight_decay, plateau=self.plateau, record_loss=True, want_logits=True, **self._fit_options)

    # -- independent images (no warm-start chain): one batched device fit + batched retries ---------------------------
    def fit_batch(self, grid: K.Grid, unaries: torch.Tensor, init_params: Optional[torch.Tensor] = None,
                  epochs: Optional[int] = None) -> FitReport:
        """unaries [n_images, N] on the device (fg < 0.5).  Returns the fitted parameters and the gate metric."""
        n = unaries.shape[0]
        dev = unaries.device
        params = (init_params.clone() if init_params is not None else self.fresh_params(n, dev)).contiguous()
        has_fg = ((unaries < 0.5).any(dim=1) & (unaries >= 0.5).any(dim=1)).cpu().tolist()
        skipped = [not h for h in has_fg]
        active = [i for i in range(n) if has_fg[i]]
        iou = torch.zeros(n, device=dev)
        final_loss = torch.full((n,), float("nan"), device=dev)
        status = torch.zeros(n, dtype=torch.int32, device=dev)
        logits = torch.zeros(n, grid.n_points, device=dev)
        retries = [0] * n
        todo, ep = active, (epochs or self.num_epochs)
        for attempt in range(self.retrys + 1):
            if not todo:
                break
            idx = torch.tensor(todo, device=dev)
            sub = params[idx].contiguous()
            res = self._run(sub, grid, unaries[idx].contiguous(), ep)
            gate = K.miou((torch.sigmoid(res.logits) > 0.5).float(), (unaries[idx] > 0.5).float(), 0.5, 0.5, invert=True)
            params[idx] = res.params
            iou[idx], final_loss[idx], status[idx], logits[idx] = gate, res.loss_hist[:, -1], res.status, res.logits
            failed = [todo[k] for k, ok in enumerate((gate >= self.threshold).cpu().tolist()) if not ok]
            if attempt < self.retrys and failed:
                # reset parameters and retry with the full number of epochs (path_connected_net.py:975-985)
                params[torch.tensor(failed, device=dev)] = self.fresh_params(len(failed), dev)
                for i in failed:
                    retries[i] += 1
            todo, ep = (failed if attempt < self.retrys else []), self.num_epochs
        report = FitReport(params, iou, final_loss, retries, skipped, status, logits)
        bad = torch.nonzero(status != 0).reshape(-1).cpu().tolist()
        if bad and self.on_nonfinite == "raise":
            raise NonFiniteLossError(bad, report)
        return report

    # -- sequences with warm start: frames in order inside a sequence, sequences batched ----------------------------
    def fit_sequences(self, grid: K.Grid, unaries: torch.Tensor, seq_ids: Sequence[int]) -> FitReport:
        """`reuse_state` semantics (path_connected_net.py:867-870, 899-908, 987-994): frame k of a sequence starts from the
        fitted state of frame k-1 and trains `reuse_state_epochs`; the first frame (and every retry) trains `num_epochs`.
        Frames at the same position of different sequences are independent and are fitted in one batch."""
        n, dev = unaries.shape[0], unaries.device
        seqs: Dict[int, List[int]] = {}
        for i, sid in enumerate(seq_ids):
            seqs.setdefault(int(sid), []).append(i)
        out = FitReport(torch.zeros(n, self.spec.n_params, device=dev), torch.zeros(n, device=dev),
                        torch.zeros(n, device=dev), [0] * n, [False] * n, torch.zeros(n, dtype=torch.int32, device=dev),
                        torch.zeros(n, grid.n_points, device=dev))
        prev: Dict[int, torch.Tensor] = {}
        depth = max(len(v) for v in seqs.values())
        for k in range(depth):
            frames = [(sid, v[k]) for sid, v in seqs.items() if k < len(v)]
            warm = [(sid, i) for sid, i in frames if self.reuse_state and sid in prev]
            cold = [(sid, i) for sid, i in frames if not (self.reuse_state and sid in prev)]
            for group, init, ep in ((warm, True, self.reuse_state_epochs), (cold, False, self.num_epochs)):
                if not group:
                    continue
                ids = [i for _, i in group]
                ip = torch.stack([prev[sid] for sid, _ in group]) if init else None
                rep = self.fit_batch(grid, unaries[ids].contiguous(), ip, epochs=ep)
                for j, (sid, i) in enumerate(group):
                    out.params[i], out.iou[i], out.final_loss[i] = rep.params[j], rep.iou[j], rep.final_loss[j]
                    out.status[i], out.logits[i] = rep.status[j], rep.logits[j]
                    out.retries[i], out.skipped[i] = rep.retries[j], rep.skipped[j]
                    if not rep.skipped[j] and float(rep.iou[j]) >= self.threshold:
                        prev[sid] = rep.params[j].clone()   # only a proper fit is handed on (:987-994)
        return out

    # -- PriorCache-compatible export ---------------------------------------------------------------------------------
    def prior_cache_state(self, report: FitReport, indices: Optional[Sequence[int]] = None, model_type: Optional[str] = None,
                          model_args: Optional[dict] = None) -> dict:
        """The dict PriorCache.get_state() returns / `prior_cache_epoch_N.pth` holds (awesome/util/prior_cache.py:61-71):
        {model_type, model_args (json), store_device, cache {str(idx): state_dict}} with the reference's key names."""
        probe = self.model_factory()
        cache = {}
        for k in range(report.params.shape[0]):
            if report.skipped[k]:
                continue
            if hasattr(probe, "unpack_flat"):
                sd = probe.unpack_flat(report.params[k].cpu())
            else:
                sd = K.unpack_params(self.spec, report.params[k].cpu(), convexnet_keys=self._convexnet_keys)
            cache[str(indices[k] if indices is not None else k)] = sd
        return {"model_type": model_type or f"{type(probe).__module__}.{type(probe).__name__}",
                "model_args": json.dumps(model_args or {}), "store_device": "cpu", "cache": cache}
