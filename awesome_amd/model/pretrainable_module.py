"""The reference's pretrain entry point on the fused HIP fits.

Reference interface (same names, arguments and error behaviour):
    PretrainableModule.pretrain / pretrain_load_state              awesome/model/pretrainable_module.py:15-83
    PathConnectedNet.pretrain -> _prior_based_pretrain             awesome/model/path_connected_net.py:472-509, 730-1007
    ConvexDiffeomorphismNet.pretrain                               awesome/model/convex_diffeomorphism_net.py:190-475
    caller: TorchAgent._pretrain                                   awesome/agent/torch_agent.py:553-627
    via WrapperModule.pretrain(..., wrapper_module=self)           awesome/model/wrapper_module.py:325-340

What the reference does per image of the training set (batch size 1, dataset order): load the image's prior state
(PriorManager), read the unaries from the segmentation module (prior evaluation off), take the clean grid from
`get_prior_args`, skip images without foreground or background, optionally load a per-image checkpoint or the previous image's
fitted state (`reuse_state`), run E full-batch optimizer steps, gate the result by fg-IoU with reset + retry, store the state
in the PriorCache (+ optional checkpoint).  The returned state is `PriorCache.get_state()`.

Here the E-step loops run on the device for MANY images at once (`inrfit_fit` / `inrfit_pcn_fit` / `inrfit_cdn_fit`): the
images of the set are independent unless `reuse_state` chains them, so the driver below first walks the data set (states are
generated in dataset order, i.e. the global RNG is consumed like the reference's DataLoader pass does), then fits all cold
images of equal grid shape in batched device calls, gates them, re-fits the failures from fresh parameters, and writes every
state back under its key.  With `reuse_state=True` the chain is inherently sequential: frame k starts from frame k-1's fit and
trains `reuse_state_epochs`, exactly as in the reference.

The IoU gate reads what the reference's reads: the output of the LAST training forward (`InrOptDesc.logits_at_last_forward`; the
stored state is the one after the last optimizer step, as in the reference).  One deviation (documented, not hidden): a retry
draws its fresh parameters after all first attempts of the batch, so the global RNG stream differs from the reference's
image-by-image order from the first retry on."""
from __future__ import annotations

import copy
import logging
import os
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch

from .. import icnn as K


class PretrainableModule:
    """Marks a module that can be pre-trained (awesome/model/pretrainable_module.py:12-83)."""

    def pretrain(self, train_set, test_set, device: torch.device, agent, use_progress_bar: bool = True, **kwargs) -> Any:
        raise NotImplementedError

    def pretrain_load_state(self, train_set, test_set, device: torch.device, agent, state: Any, use_progress_bar: bool = True,
                            do_pretrain_checkpoints: bool = False, pretrain_checkpoint_dir: Optional[str] = None, **kwargs) -> None:
        raise NotImplementedError


def decompose_training_item(item: Any, training_dataset) -> Tuple[Any, Any, Any, Optional[Tuple[int, Any]]]:
    """TorchAgent.decompose_training_item (awesome/agent/torch_agent.py:380-426) for an UN-collated dataset item:
    (inputs, labels, indices, prior_state)."""
    prior_state = None
    if getattr(training_dataset, "has_prior", False) and getattr(training_dataset, "return_prior", True):
        (key, state), item = item
        prior_state = (int(key), state)
    inputs, labels = item[0], item[1]
    indices = item[2] if getattr(training_dataset, "returns_index", False) and len(item) > 2 else None
    return inputs, labels, indices, prior_state


def load_pretrain_checkpoint(model: torch.nn.Module, path: str, device=None) -> bool:
    """path_connected_net.py:36-43."""
    try:
        model.load_state_dict(torch.load(path, map_location=device, weights_only=True))
        return True
    except Exception as e:   # noqa: BLE001 (the reference logs and carries on)
        logging.error(f"Could not load pretrain checkpoint from {path}. Error: {e}")
        return False


def save_pretrain_checkpoint(state_dict: Dict[str, torch.Tensor], path: str) -> bool:
    """path_connected_net.py:45-51 (atomic: a crash never leaves half a checkpoint to be resumed from)."""
    try:
        tmp = path + ".tmp"
        torch.save(state_dict, tmp)
        os.replace(tmp, path)
        return True
    except Exception as e:   # noqa: BLE001
        logging.error(f"Could not save pretrain checkpoint to {path}. Error: {e}")
        return False


def center_of_mass(unaries: torch.Tensor) -> torch.Tensor:
    """convex_diffeomorphism_net.py:262-266: (row, col) of the foreground (unaries <= 0.5), truncated to long."""
    fg = (1 - (unaries > 0.5).float()).bool().squeeze()
    com = torch.sum(torch.argwhere(fg), dim=0) / torch.sum(fg).to(dtype=torch.long)
    return com.to(dtype=torch.long)


class _Image:
    """One item of the training set after the data pass."""
    __slots__ = ("pos", "key", "state", "unaries", "grid", "skip", "fitted", "from_checkpoint", "iou", "retries", "status")

    def __init__(self, pos, key, state, unaries, grid, skip):
        self.pos, self.key, self.state, self.unaries, self.grid, self.skip = pos, key, state, unaries, grid, skip
        self.fitted: Optional[Dict[str, torch.Tensor]] = None
        self.from_checkpoint, self.iou, self.retries, self.status = False, float("nan"), 0, 0


class PriorFitMixin(PretrainableModule):
    """`pretrain` for a prior module whose per-image fit has a fused device form.  The module provides:

        _pretrain_defaults() -> dict                     default pretrain kwargs of the reference's method
        _engine_pack(state_dict) -> flat [Ptot]           parameters of a state_dict in the C ABI's flat order
        _engine_unpack(flat) -> {key: tensor}             ... and back (parameters only, reference key names)
        _engine_fit(grid, unaries [n, N], flat [n, Ptot], epochs, cold, opts, states) -> (flat, logits [n, N], status [n])
                                                          (`states`: the images' starting state_dicts on a cold start - buffers)
        _engine_after_fit(state_dict)                     buffers a fit changes (e.g. ActNorm's data_dep_init_done)
        _engine_warm_start(flat, ctx, image, opts) -> (flat, ctx)     adjust the previous frame's fit before the short refit
        _engine_retry_state(fresh, failed) -> state_dict               what reset_parameters leaves untouched, for a retry

    `states is None` in _engine_fit = a warm start (the previous frame's fitted state, loaded with load_state_dict in the
    reference: every buffer, e.g. ActNorm's data_dep_init_done = 1, comes along); `opts["_prefit"]`: run the pre-fit stages.
    """

    # -- hooks with defaults ----------------------------------------------------------------------------------------------
    def _pretrain_defaults(self) -> Dict[str, Any]:
        return {}

    def _engine_after_fit(self, sd: Dict[str, torch.Tensor]) -> None:
        return None

    def _engine_warm_start(self, flat: torch.Tensor, ctx: Any, image: "_Image", opts: Dict[str, Any]):
        return flat, ctx

    def _engine_retry_state(self, fresh: Dict[str, torch.Tensor], failed: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """The state a retry starts from: `fresh` (reset_parameters) with whatever reset_parameters does NOT touch in the reference
        taken over from the failed fit (modules without a reset_parameters method keep their fitted values and buffers)."""
        return fresh

    def _engine_fresh_state(self, key: int = 0, attempt: int = 0) -> Dict[str, torch.Tensor]:
        """reset_parameters() (the retry of path_connected_net.py:975-985) as a new state_dict; this module's own parameters are
        restored afterwards.  With a per-key seeded cache (PriorCache.key_seed, an extension for sharded runs) the redraw of key
        k, attempt a comes from its own generator too, so a retry does not depend on which other images share the rank."""
        keep = copy.deepcopy(self.state_dict())
        seed = getattr(self, "_fresh_seed", None)
        if seed is None:
            self.reset_parameters()
        else:
            with torch.random.fork_rng(devices=[]):
                torch.manual_seed(int(seed) + int(key) + 1000003 * (int(attempt) + 1))
                self.reset_parameters()
        fresh = copy.deepcopy(self.state_dict())
        self.load_state_dict(keep)
        return fresh

    # -- the entry point -----------------------------------------------------------------------------------------------------
    def pretrain(self, train_set, test_set, device: torch.device, agent, use_progress_bar: bool = True,
                 do_pretrain_checkpoints: bool = False, use_pretrain_checkpoints: bool = False,
                 pretrain_checkpoint_dir: Optional[str] = None, wrapper_module: Optional[torch.nn.Module] = None, **kwargs) -> Any:
        if wrapper_module is None:
            raise ValueError("Wrapper model must be provided for pretraining.")
        ds = getattr(agent, "training_dataset", None)
        # path_connected_net.py:485-509: `isinstance(ds, PriorDataset) and ds.has_prior` -> per-image fits, anything else -> the
        # spatio-temporal fit (a module without one raises the reference's "Agent must be trained on a prior dataset.")
        if ds is None or not hasattr(ds, "__prior_cache__") or not getattr(ds, "has_prior", False):
            return self._non_prior_based_pretrain(train_set=train_set, test_set=test_set, device=device, agent=agent,
                                                  use_progress_bar=use_progress_bar, wrapper_module=wrapper_module, **kwargs)
        if not kwargs.get("use_prior_sigmoid", True):
            # the fused fits evaluate criterion(sigmoid(logits), unaries) (wrapper_module.py:265-273 with use_sigmoid=True, every
            # reference config); raw-logit criteria have no device form - refuse instead of silently applying the sigmoid
            raise NotImplementedError("use_prior_sigmoid=False has no fused form (no reference config sets it)")
        if (do_pretrain_checkpoints or use_pretrain_checkpoints) and pretrain_checkpoint_dir is None:
            raise ValueError("Pretrain checkpoint dir must be provided.")
        if do_pretrain_checkpoints:
            os.makedirs(pretrain_checkpoint_dir, exist_ok=True)
        opts = dict(self._pretrain_defaults())
        opts.update(kwargs)
        cache = ds.__prior_cache__
        self._fresh_seed = getattr(cache, "key_seed", None)
        device = torch.device(device)
        training_state = wrapper_module.training
        try:
            wrapper_module.eval()
            images = self._collect(train_set, ds, wrapper_module, device)
            ckpt = (lambda im: os.path.join(pretrain_checkpoint_dir, f"pretrain_checkpoint_{im.pos}.pth"))
            if use_pretrain_checkpoints:
                for im in images:
                    if not im.skip and os.path.exists(ckpt(im)):
                        try:
                            sd = torch.load(ckpt(im), map_location="cpu", weights_only=True)
                            self._engine_pack(sd)   # validates keys and shapes like load_state_dict would
                            im.fitted, im.from_checkpoint = sd, True
                            logging.info(f"Loaded pretrain checkpoint from {ckpt(im)}. Continuing with next image.")
                        except Exception as e:   # noqa: BLE001
                            logging.error(f"Could not load pretrain checkpoint from {ckpt(im)}. Error: {e}")
            if opts.get("reuse_state", True):
                self._fit_chain(images, device, opts)
            else:
                self._fit_independent(images, device, opts)
            bad = [im.pos for im in images if im.status != 0]
            for im in images:
                if im.fitted is not None:
                    cache[im.key] = im.fitted
                    if do_pretrain_checkpoints and not im.from_checkpoint and im.status == 0:
                        save_pretrain_checkpoint(im.fitted, ckpt(im))
            self.pretrain_report = [dict(index=im.pos, key=im.key, skipped=im.skip, iou=im.iou, retries=im.retries,
                                         from_checkpoint=im.from_checkpoint, status=im.status) for im in images]
            if bad:
                raise ValueError(f"Loss is nan or inf! (images {bad})")
            return cache.get_state()
        finally:
            wrapper_module.train(training_state)

    def pretrain_load_state(self, train_set, test_set, device: torch.device, agent, state: Any, use_progress_bar: bool = True,
                            wrapper_module: Optional[torch.nn.Module] = None, **kwargs) -> None:
        """path_connected_net.py:1010-1019: the pretrain state IS the prior cache."""
        agent.training_dataset.__prior_cache__.set_state(state)

    def _non_prior_based_pretrain(self, **kwargs) -> Any:
        raise ValueError("Agent must be trained on a prior dataset.")

    # -- data pass ---------------------------------------------------------------------------------------------------------
    def _collect(self, train_set, ds, wrapper_module, device) -> List["_Image"]:
        images: List[_Image] = []
        was = getattr(wrapper_module, "evaluate_prior", True)
        wrapper_module.evaluate_prior = False   # TemporaryProperty(wrapper_module, evaluate_prior=False), :832-836
        try:
            for pos in range(len(train_set)):
                inputs, _, _, prior_state = decompose_training_item(train_set[pos], ds)
                if prior_state is None:
                    raise ValueError("the training set returned no prior state (has_prior / return_prior must be set)")
                key, state = prior_state
                dev_in = [t.to(device)[None] if isinstance(t, torch.Tensor) else t for t in (inputs if isinstance(inputs, (list, tuple)) else [inputs])]
                with torch.no_grad():
                    unaries = wrapper_module(*dev_in)                               # (1, 1, H, W)
                    pargs, _ = wrapper_module.get_prior_args(dev_in[0], *dev_in[1:], segm=unaries[0, ...])
                grid = pargs[0].detach()
                if grid.dim() == 3:
                    grid = grid[None]
                skip = len(torch.unique(unaries >= 0.5)) == 1                        # :848-855
                if skip:
                    logging.warning(f"Unaries of segmentation model contain no foreground. Skipping image. {pos}")
                images.append(_Image(pos, key, state, unaries.detach().to(torch.float32), grid.to(torch.float32), skip))
        finally:
            wrapper_module.evaluate_prior = was
        return images

    # -- fitting -----------------------------------------------------------------------------------------------------------
    def _merge(self, base_state: Dict[str, torch.Tensor], flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        sd = {k: v.detach().clone().cpu() for k, v in base_state.items()}
        for k, v in self._engine_unpack(flat.detach().cpu()).items():
            sd[k] = v.reshape(sd[k].shape).to(sd[k].dtype).clone()
        self._engine_after_fit(sd)
        return sd

    def _gate(self, logits: torch.Tensor, unaries: torch.Tensor) -> torch.Tensor:
        """proper_prior_fit_metric = MIOU(average='binary', invert=True) on (prior > 0.5) vs (unaries > 0.5) (:964-972)."""
        return K.miou((torch.sigmoid(logits) > 0.5).float(), (unaries > 0.5).float(), 0.5, 0.5, invert=True)

    def _fit_group(self, group: List["_Image"], flats: torch.Tensor, epochs: int, cold: bool, device, opts, states=None,
                   prefit: bool = False):
        """One device call for images that share a grid shape.  Returns (flat [n, Ptot], iou [n], status [n]).  `prefit`: the
        pre-fit stages (learn_flow_identity / learn_convex_net) run - in the reference once per cold image BEFORE the retry loop
        (path_connected_net.py:871-894), never again after reset_parameters()."""
        opts = dict(opts, _prefit=bool(prefit))
        g0 = group[0].grid
        same = all(im.grid.shape == g0.shape and (im.grid is g0 or torch.equal(im.grid, g0)) for im in group[1:])
        c, h, w = g0.shape[1], g0.shape[2], g0.shape[3]
        if same:
            grid = K.Grid.explicit(g0[0].reshape(c, h * w))
        else:
            grid = K.Grid.explicit(torch.stack([im.grid[0].reshape(c, h * w) for im in group]))
        un = torch.stack([im.unaries.reshape(-1) for im in group]).contiguous()
        from ..measures import criterion_targets   # UnariesConversionLoss (every ConvexDiffeomorphismNet config) binarises the targets
        un_fit = criterion_targets(opts.get("criterion"), un).contiguous()
        flat, logits, status = self._engine_fit(grid, un_fit, flats.to(device).contiguous(), int(epochs), cold, opts, states)
        return flat, self._gate(logits, un), status

    def _fit_independent(self, images: List["_Image"], device, opts) -> None:
        todo = [im for im in images if not im.skip and im.fitted is None]
        thr, retrys = float(opts.get("proper_prior_fit_threshold", 0.5)), int(opts.get("proper_prior_fit_retrys", 1))
        chunk = int(opts.get("device_batch_size", 64))
        by_shape: Dict[Tuple[int, ...], List[_Image]] = {}
        for im in todo:
            by_shape.setdefault(tuple(im.grid.shape), []).append(im)
        for shape_group in by_shape.values():
            for off in range(0, len(shape_group), chunk):
                group = shape_group[off:off + chunk]
                pending, states = group, [im.state for im in group]
                for attempt in range(retrys + 1):
                    flats = torch.stack([self._engine_pack(sd) for sd in states])
                    flat, iou, status = self._fit_group(pending, flats, int(opts.get("num_epochs", 2000)), True, device, opts, states,
                                                        prefit=attempt == 0)
                    ok = (iou >= thr).cpu().tolist()
                    failed, failed_states = [], []
                    for j, im in enumerate(pending):
                        im.fitted = self._merge(states[j], flat[j])
                        im.iou, im.status = float(iou[j]), int(status[j])
                        if ok[j]:
                            logging.info(f"Proper prior fit on image index: {im.pos} got metric: {im.iou}.")
                        elif attempt < retrys and im.status == 0:
                            logging.info(f"Prior fit not proper on image index: {im.pos}. Retrying. Metric: {im.iou} Threshold: {thr}")
                            im.retries += 1
                            failed.append(im)
                            failed_states.append(self._engine_retry_state(self._engine_fresh_state(im.key, attempt), im.fitted))
                        else:
                            logging.info(f"Prior fit not proper on image index: {im.pos}. Retries exceeded. Metric: {im.iou} Threshold: {thr}")
                    pending, states = failed, failed_states
                    if not pending:
                        break

    def _fit_chain(self, images: List["_Image"], device, opts) -> None:
        """reuse_state (:867-870, 899-908, 987-994): frame k starts from the previous PROPER fit and trains reuse_state_epochs; the
        first frame and every retry train num_epochs from their own / fresh parameters."""
        thr, retrys = float(opts.get("proper_prior_fit_threshold", 0.5)), int(opts.get("proper_prior_fit_retrys", 1))
        prev_flat, ctx = None, None
        for im in images:
            if im.skip:
                continue
            if im.fitted is not None:   # from a checkpoint: counts as a proper fit and is handed on
                prev_flat = self._engine_pack(im.fitted).to(device)
                ctx = self._engine_chain_context(ctx, im)
                continue
            if prev_flat is not None:
                start, ctx = self._engine_warm_start(prev_flat.clone(), ctx, im, opts)
                base_state = im.state
                epochs, cold = int(opts.get("reuse_state_epochs", 200)), False
            else:
                start, base_state = self._engine_pack(im.state), im.state
                epochs, cold = int(opts.get("num_epochs", 2000)), True
            proper = False
            for attempt in range(retrys + 1):
                flat, iou, status = self._fit_group([im], start[None], epochs, cold, device, opts, [base_state] if cold else None,
                                                    prefit=cold and attempt == 0)
                im.fitted, im.iou, im.status = self._merge(base_state, flat[0]), float(iou[0]), int(status[0])
                if im.iou >= thr:
                    proper = True
                    break
                if attempt < retrys and im.status == 0:
                    im.retries += 1
                    base_state = self._engine_retry_state(self._engine_fresh_state(im.key, attempt), im.fitted)
                    start, epochs, cold = self._engine_pack(base_state), int(opts.get("num_epochs", 2000)), True
                else:
                    break
            if proper:
                prev_flat = flat[0].detach().clone()
                ctx = self._engine_chain_context(ctx, im)

    def _engine_chain_context(self, ctx: Any, image: "_Image") -> Any:
        return ctx
