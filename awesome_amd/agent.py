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
    prior module; `criterion(output, labels)` e.g. FBMSJointLoss.  One optimizer covers the segmentation module's parameters
    and the prior module's Parameter objects (whose storage the bank re-points per image); per-image Adam moments of the prior
    live in `prior_opt_state[key]`, swapped with the rows - the reference shares ONE optimizer state over all images' priors
    (torch_agent.py:812-839 builds it once over model.parameters()), which is what `shared_prior_moments=True` reproduces."""

    def __init__(self, wrapper: torch.nn.Module, bank: PriorBank, criterion: Callable, optimizer: torch.optim.Optimizer):
        self.wrapper, self.bank, self.criterion, self.optimizer = wrapper, bank, criterion, optimizer
        self.forward_additional_loss_args = _loss_takes_input(criterion)

    def perform_step(self, key: Any, inputs: Sequence[torch.Tensor], labels: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (loss, output) as device tensors (no host sync)."""
        self.optimizer.zero_grad()
        with self.bank.manager(self.wrapper.prior_module, key):
            out = self.wrapper(*inputs)
            loss = self.criterion(out, labels, _input=list(inputs)) if self.forward_additional_loss_args else self.criterion(out, labels)
            loss.backward()
            self.optimizer.step()
            self.wrapper.enforce_convexity()   # batch_processed hook (awesome_runner.py:294-297)
        return loss.detach(), out.detach()
