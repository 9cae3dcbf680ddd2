"""The slice of the reference's TorchAgent that the hot path touches (awesome/agent/torch_agent.py).

    PretrainAgent._pretrain          <- TorchAgent._pretrain (:553-627): load `pretrain_state.pth` if present, else call
                                        model.pretrain(train_set, test_set, device, agent, use_progress_bar, **pretrain_args) and
                                        save the returned state
    PretrainAgent._decompose_training_item  <- :380-426
    JointTrainer.perform_step        <- TorchAgent._perform_step (:428-551): per-image prior swap, forward through the
                                        WrapperModule, criterion(output, labels), backward, optimizer.step, enforce_convexity
                                        (the batch_processed hook of awesome/run/awesome_runner.py:294-297)

The reference keeps every image's prior state in a CPU PriorCache and swaps it in and out with load_state_dict / deepcopy
around every step (dataset/prior_dataset.py:96-110), then pulls loss and outputs back to the host (:501-504).  Here the priors
live in one device-resident table (awesome_amd.PriorBank), the swap re-points parameter views, and nothing syncs with the host
unless the caller reads a value.  Everything else the reference agent does (tracker, tensorboard, events, checkpoint objects) is
control plane and out of scope."""
from __future__ import annotations

import inspect
import logging
import os
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import torch

from .model.pretrainable_module import PretrainableModule, decompose_training_item
from .prior_bank import PriorBank


def _cpu_state(state: Any) -> Any:
    if isinstance(state, torch.Tensor):
        return state.detach().cpu()
    if isinstance(state, dict):
        return {k: _cpu_state(v) for k, v in state.items()}
    if isinstance(state, (list, tuple)):
        return type(state)(_cpu_state(v) for v in state)
    return state


class PretrainAgent:
    def __init__(self, training_dataset, device="cuda:0", agent_folder: str = "runs/agent", pretrain_args: Optional[Dict[str, Any]] = None,
                 do_pretraining: Optional[bool] = True, force_pretrain: Optional[bool] = False,
                 pretrain_state_path: Optional[str] = None, logger=None):
        self.training_dataset = training_dataset
        self.device = torch.device(device)
        self.agent_folder = agent_folder
        self.pretrain_args = dict(pretrain_args or {})
        self.do_pretraining, self.force_pretrain = do_pretraining, force_pretrain
        self.pretrain_state_path = pretrain_state_path
        self.logger = logger

    def _decompose_training_item(self, item: Any):
        return decompose_training_item(item, self.training_dataset)

    def _pretrain(self, model: torch.nn.Module, train_set, test_set=None, use_progress_bar: bool = True, **kwargs) -> Optional[Any]:
        """TorchAgent._pretrain (:553-627).  Returns the pretrain state (the PriorCache state for prior datasets)."""
        if not (self.do_pretraining is True or (self.do_pretraining is None and kwargs.get("do_pretraining", False))):
            return None
        if not isinstance(model, PretrainableModule):
            raise ValueError("Model is not pretrainable!")
        pre = dict(self.pretrain_args)
        pre.update(kwargs.get("pretrain_args", {}))
        state, loaded = None, False
        if self.pretrain_state_path is not None:
            if os.path.exists(self.pretrain_state_path):
                try:
                    state = torch.load(self.pretrain_state_path, map_location="cpu", weights_only=False)
                    model.pretrain_load_state(train_set=train_set, test_set=test_set, device=self.device, agent=self,
                                              use_progress_bar=use_progress_bar, state=state, **pre)
                    loaded = True
                    logging.info(f"Pretrain state loaded from {self.pretrain_state_path}")
                except Exception as err:   # noqa: BLE001 (the reference logs and pretrains instead)
                    logging.error(f"Error loading pretrain state: {err}")
        else:
            self.pretrain_state_path = os.path.join(self.agent_folder, "pretrain_state.pth")
        force = self.force_pretrain is True or (self.force_pretrain is None and kwargs.get("force_pretrain", False))
        if not loaded or force:
            logging.info("Starting pretraining...")
            state = model.pretrain(train_set=train_set, test_set=test_set, device=self.device, agent=self,
                                   use_progress_bar=use_progress_bar, **pre)
            if state is not None:
                os.makedirs(os.path.dirname(self.pretrain_state_path) or ".", exist_ok=True)
                tmp = self.pretrain_state_path + ".tmp"
                torch.save(_cpu_state(state), tmp)
                os.replace(tmp, self.pretrain_state_path)
                logging.info(f"Pretrain state saved to {self.pretrain_state_path}")
            else:
                logging.info("No pretrain state returned, not saving...")
        logging.info("Pretraining done!")
        return state


def _loss_takes_input(loss: Callable) -> bool:
    """TorchAgent.forward_additional_loss_args (torch_agent.py:150-164): a loss whose `__call__` (for an nn.Module: whose `forward`)
    names `_input` or `kwargs` also receives `_input=device_inputs`."""
    try:
        spec = inspect.signature(loss.__call__).parameters
    except (TypeError, ValueError):
        return False
    if "_input" in spec or "kwargs" in spec:
        if isinstance(loss, torch.nn.Module):
            spec = inspect.signature(loss.forward).parameters
            return "_input" in spec or "kwargs" in spec
        return True
    return False


class JointTrainer:
    """One optimisation step of the joint segmentation + prior training (TorchAgent._perform_step, :428-551) on device-resident
    priors.  `wrapper` is a WrapperModule(segmentation_module, prior_module); `bank` holds one parameter row per image of the
    prior module; `criterion(output, labels)` e.g. FBMSJointLoss.  `optimizer`: ONE torch optimizer over the segmentation module's
    parameters and the prior module's Parameter objects, as the reference builds it (torch_agent.py:812-839).

    Two implementations of the same step:

    * **fused** (default whenever it applies): the segmentation module runs in torch; everything behind its output - prior forward on
      the image's bank row, sigmoid, the composite loss, d loss / d seg, the prior's backward from the activations of that same pass,
      Adam / Adamax and enforce_convexity on the row in place - is ONE C-ABI call (`inrfit_joint_step` and its path-connected
      variants, awesome_amd.joint).  The torch optimizer then steps the segmentation parameters only (the prior's Parameters never
      receive a .grad, which torch.optim skips); the prior's hyper-parameters are read from their param group every step, so host
      LR schedulers keep working.  Applies to: batch size 1 (what a PriorManager swap implies), FBMSJointLoss / AwesomeImageLoss
      (before its extra penalty) with fusable criteria, Adam or Adamax without amsgrad, a prior with a device form (ICNN,
      PathConnectedNet, ConvexDiffeomorphismNet; these two only without weight decay, which torch would also apply to their 1x1 /
      linear layer).
    * **autograd**: WrapperModule forward -> criterion -> loss.backward() through the HIP autograd bridges -> optimizer.step() ->
      enforce_convexity(); any criterion, any optimizer.

    The reference shares ONE optimizer state over all images' priors (it has one prior model whose VALUES are swapped): that is
    `shared_prior_moments=True`, the default; False keeps Adam moments and step counts per image (fused path only)."""

    def __init__(self, wrapper: torch.nn.Module, bank: PriorBank, criterion: Callable, optimizer: torch.optim.Optimizer,
                 fused: Optional[bool] = None, shared_prior_moments: bool = True, check_finite: str = "epoch"):
        if check_finite not in ("step", "epoch"):
            raise ValueError("check_finite: 'step' (raise before backward like the reference, one host sync per step) or 'epoch' "
                             "(device flag, the caller polls raise_if_failed())")
        self.check_finite = check_finite
        self.wrapper, self.bank, self.criterion, self.optimizer = wrapper, bank, criterion, optimizer
        self.forward_additional_loss_args = _loss_takes_input(criterion)
        self.shared_prior_moments = bool(shared_prior_moments)
        self._fused_plan = self._plan_fused()
        if fused and self._fused_plan is None:
            raise ValueError("this combination of prior module / criterion / optimizer has no fused joint step")
        self.fused = (self._fused_plan is not None) if fused is None else bool(fused)
        self._t: Dict[Any, int] = {}
        self._opt_state: Dict[Any, Tuple[torch.Tensor, Optional[torch.Tensor]]] = {}
        self.last_status: Optional[torch.Tensor] = None
        self.failed = torch.zeros((), dtype=torch.bool, device=bank.device)   # any step so far had a non-finite loss (device flag)
        self._path: Optional[str] = None       # which implementation took the previous step ("fused" / "autograd")

    # -- ONE optimizer state for the prior, whichever implementation steps it -----------------------------------------------
    # The reference has one torch optimizer for the whole run (torch_agent.py:812-839).  The fused step keeps the prior's moments
    # in its own flat buffers; when a step has to take the other implementation (AwesomeImageLoss once the runner's hook has set
    # `extra_penalty`, a batch of more than one image, the step that initialises ActNorm) the moments and the step count move
    # with it, so the trajectory is the single-optimizer one.  Per-image moments (shared_prior_moments=False) exist
    # in the fused buffers only and are not carried over.
    def _prior_params(self):
        from .prior_bank import _ordered_parameters
        return list(_ordered_parameters(self.wrapper.prior_module))

    def _moment_keys(self) -> Tuple[str, str]:
        return ("exp_avg", "exp_inf") if self._fused_plan["kind"] == "adamax" else ("exp_avg", "exp_avg_sq")

    def _flat_moment_views(self, plan):
        """[(parameter, first-moment view, second-moment view)] into the fused buffers, in bank-row order."""
        iopt, fopt = self._state_for(None, plan)
        P = plan["ispec"].n_params
        Pd = 0 if plan["dspec"] is None else plan["dspec"].n_params
        out, off = [], 0
        for p in self._prior_params():
            n = p.numel()
            if off < P:
                m, v = iopt[off:off + n], iopt[P + off:P + off + n]
            else:
                o = off - P
                m, v = fopt[o:o + n], fopt[Pd + o:Pd + o + n]
            out.append((p, m, v))
            off += n
        assert off == P + Pd, (off, P, Pd)
        return out

    def _hand_over(self, to: str) -> None:
        plan = self._fused_plan
        if plan is None or not self.shared_prior_moments or self._path in (None, to):
            self._path = to
            return
        k1, k2 = self._moment_keys()
        views = self._flat_moment_views(plan)
        if to == "autograd":      # fused buffers -> torch.optim state
            t = self._t.get(None, 0)
            if t > 0:
                for p, m, v in views:
                    st = self.optimizer.state[p]
                    st["step"] = torch.tensor(float(t), dtype=torch.float32)
                    st[k1], st[k2] = m.detach().clone().view_as(p), v.detach().clone().view_as(p)
        else:                     # torch.optim state -> fused buffers
            steps = [float(self.optimizer.state[p]["step"]) for p, _, _ in views if "step" in self.optimizer.state.get(p, {})]
            if steps:
                for p, m, v in views:
                    st = self.optimizer.state.get(p, {})
                    if k1 in st:
                        m.copy_(st[k1].reshape(-1))
                        v.copy_(st[k2].reshape(-1))
                self._t[None] = int(max(steps))
        self._path = to

    # -- fused path ----------------------------------------------------------------------------------------------------------
    def _plan_fused(self) -> Optional[Dict[str, Any]]:
        prior = getattr(self.wrapper, "prior_module", None)
        if prior is None or self.bank.device.type != "cuda":
            return None
        if not getattr(self.wrapper, "use_prior_sigmoid", True) or not getattr(self.wrapper, "evaluate_prior", True):
            return None
        opt = self.optimizer
        if isinstance(opt, torch.optim.Adam) and not isinstance(opt, torch.optim.AdamW):
            kind = "adam"
        elif isinstance(opt, torch.optim.Adamax):
            kind = "adamax"
        else:
            return None
        from .prior_bank import _ordered_parameters
        pids = {id(p) for p in _ordered_parameters(prior)}
        groups = [g for g in opt.param_groups if any(id(p) in pids for p in g["params"])]
        if len(groups) != 1 or groups[0].get("amsgrad", False) or groups[0].get("maximize", False):
            return None
        if not pids <= {id(p) for p in groups[0]["params"]}:
            return None
        if hasattr(prior, "_specs"):
            try:
                ispec, dspec = prior._specs()
            except NotImplementedError:   # no fused form of this composite at all ('resnet' flow backbone): autograd step
                return None
            if not ispec.fused():         # its ICNN runs layer by layer: the composite's autograd bridges, no fused joint step
                return None
            family = "pcn" if type(dspec).__name__ == "RnvpSpec" else "cdn"
            if kind != "adam" and family == "cdn":
                return None
        elif hasattr(prior, "spec"):
            ispec, dspec, family = prior.spec, None, "icnn"
            if not ispec.fused():   # n_hidden > 130 or L > 2: inrfit_joint_step has no kernel (INR_EUNSUPPORTED); autograd step
                return None
            if getattr(prior, "fit_options", None):   # FCNet / encode nets: frozen ranges, no clamp - not part of the joint configs
                return None
        else:
            return None
        if not hasattr(self.criterion, "joint_desc") or self.criterion.joint_desc() is None:
            return None
        return dict(kind=kind, group=groups[0], family=family, ispec=ispec, dspec=dspec)

    def _state_for(self, key: Any, plan) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        k = None if self.shared_prior_moments else key
        st = self._opt_state.get(k)
        if st is None:
            from . import _lib as L
            P = plan["ispec"].n_params
            dev = self.bank.device
            iopt = torch.zeros(2 * P + L.INR_OPT_HEADER_FLOATS, dtype=torch.float32, device=dev)
            fopt = None if plan["dspec"] is None else torch.zeros(2 * plan["dspec"].n_params, dtype=torch.float32, device=dev)
            st = self._opt_state[k] = (iopt, fopt)
        return st

    def _perform_step_fused(self, key: Any, inputs: Sequence[torch.Tensor], labels: torch.Tensor):
        from . import icnn as K
        from . import joint as J
        plan, w = self._fused_plan, self.wrapper
        desc = self.criterion.joint_desc()
        g = plan["group"]
        if desc is None or (plan["family"] != "icnn" and float(g.get("weight_decay", 0.0)) != 0.0):
            return None
        if desc.form != 0 and (desc.form != 1 or desc.extra_penalty or desc.target_rule or desc.use_noneclass):
            return None   # AwesomeImageLoss with its extra penalty on: two data terms on the prior (inrfit_joint_step: INR_EUNSUPPORTED)
        xi = inputs[0] if inputs[0].dim() == 4 else inputs[0][None]
        if xi.shape[0] != 1:
            return None
        if plan["family"] == "pcn" and not all(float(b) > 0 for n, b in w.prior_module.named_buffers() if n.endswith("data_dep_init_done")):
            return None   # ActNorm's data-dependent initialisation happens in the module's first forward: take the autograd step once
        ai = tuple(a if not isinstance(a, torch.Tensor) or a.dim() == 4 else a[None] for a in inputs[1:])
        self._hand_over("fused")
        self.optimizer.zero_grad()
        seg = w.segmentation_output(xi, ai)                                   # (1, H, W), autograd attached
        pa, _ = w.get_prior_args(xi, *ai, segm=seg)
        coords = pa[0]
        coords = coords[0] if coords.dim() == 4 else coords
        grid = K.Grid.explicit(coords.reshape(coords.shape[0], -1).to(torch.float32).contiguous())
        row = self.bank.row(key)
        P = plan["ispec"].n_params
        kc = None if self.shared_prior_moments else key
        t = self._t[kc] = self._t.get(kc, 0) + 1
        iopt, fopt = self._state_for(key, plan)
        segd = seg.detach().reshape(-1).to(torch.float32).contiguous()
        tgt = labels.detach().reshape(-1).to(torch.float32).contiguous()
        hp = dict(step=t, lr=float(g["lr"]), betas=tuple(g.get("betas", (0.9, 0.999))), eps=float(g.get("eps", 1e-8)))
        if plan["family"] == "icnn":
            res = J.joint_step(plan["ispec"], row, iopt, grid, segd, tgt, desc, optimizer=plan["kind"],
                               weight_decay=float(g.get("weight_decay", 0.0)), clamp=True, **hp)
        elif plan["family"] == "pcn":
            res = J.pcn_joint_step(plan["ispec"], plan["dspec"], row[:P], row[P:], iopt, fopt, grid, segd, tgt, desc,
                                   optimizer=plan["kind"], flow_weight_decay=0.0, **hp)
        else:
            res = J.cdn_joint_step(plan["ispec"], plan["dspec"], row[:P], row[P:], iopt, fopt, grid, segd, tgt, desc,
                                   weight_decay_on_weight_g=0.0, **hp)
        self.last_status = res.status
        # A non-finite loss froze the prior's row on the device.  The reference raises ValueError('Loss is nan or inf!') before
        # backward (torch_agent.py:484-487), which needs the loss on the host: check_finite="step" does exactly that (one 4-byte
        # read per step).  The default keeps the step free of host syncs: the failure is latched in `failed`, the caller polls it
        # (scripts/run.py: once per epoch) and raises then - the run is over either way, and a backbone that has produced a NaN
        # cannot be stepped meaningfully (its own backward multiplies any gradient by NaN).
        self.failed |= res.status.reshape(-1)[0] != 0
        if self.check_finite == "step":
            self.raise_if_failed()
        if seg.requires_grad:
            seg.backward(res.dseg.view_as(seg))
        self.optimizer.step()                                                  # segmentation parameters only
        out = torch.cat([seg.detach(), torch.sigmoid(res.prior_logits).view_as(seg)], dim=0)[None]
        return res.loss[0], out

    # -- the step ------------------------------------------------------------------------------------------------------------
    def perform_step(self, key: Any, inputs: Sequence[torch.Tensor], labels: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (loss, output) as device tensors (no host sync)."""
        if self.fused and self._fused_plan is not None:
            done = self._perform_step_fused(key, inputs, labels)
            if done is not None:
                return done
        self._hand_over("autograd")
        self.optimizer.zero_grad()
        with self.bank.manager(self.wrapper.prior_module, key):
            out = self.wrapper(*inputs)
            loss = self.criterion(out, labels, _input=list(inputs)) if self.forward_additional_loss_args else self.criterion(out, labels)
            self.failed |= ~torch.isfinite(loss.detach())
            if self.check_finite == "step":
                self.raise_if_failed()
            loss.backward()
            self.optimizer.step()
            self.wrapper.enforce_convexity()   # batch_processed hook (awesome_runner.py:294-297)
        return loss.detach(), out.detach()

    def raise_if_failed(self) -> None:
        """One host sync: the reference's `ValueError("Loss is nan or inf!")` (torch_agent.py:484-487), checked when the caller asks."""
        if bool(self.failed):
            raise ValueError("Loss is nan or inf!")
