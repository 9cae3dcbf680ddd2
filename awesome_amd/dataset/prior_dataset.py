"""Per-image prior states attached to a dataset - the reference's `@prior()` / PriorManager / PriorDataset trio
(awesome/dataset/prior_dataset.py:13-158), same names and behaviour:

  * `PriorDataset` (mixin): owns a PriorCache built from (prior_model_type, prior_model_args); `has_prior`, `return_prior`,
    `prior_save` / `prior_load`.
  * `@prior()` on `__getitem__`: item -> `((index, state), item)` while the dataset has a prior and `return_prior` is set; a new
    index gets a freshly generated state on the way.
  * `PriorManager(model, prior_state=(key, state), prior_cache=...)`: on enter the state is applied to the model
    (`apply_prior` override or load_state_dict), on exit the model's current state is stored under the key.

The joint-training path does not need the copies this implies: `awesome_amd.PriorBank.manager` is the zero-copy equivalent on
a device-resident table.  This module exists so that code written against the reference's interface (the pretrain entry points,
the agent's step loop) runs unchanged."""
from __future__ import annotations

from functools import wraps
from typing import Any, Callable, Dict, Optional, Tuple, Type, Union

import torch

from ..util.prior_cache import PriorCache, _to_device


def prior():
    def decorator(function: Callable[..., Any]) -> Callable[..., Any]:
        @wraps(function)
        def wrapper(*args, **kwargs):
            self, item = args[0], args[1]
            out = function(*args, **kwargs)
            if self.has_prior and self.return_prior:
                return (item, self.__prior_cache__[item]), out
            return out
        return wrapper
    return decorator


class PriorDataset:
    def __init__(self, prior_model_type: Optional[Type[torch.nn.Module]] = None, prior_model_args: Optional[Dict[str, Any]] = None,
                 **kwargs) -> None:
        super().__init__(**kwargs)
        self.__has_prior__ = prior_model_type is not None
        self.return_prior = True
        self.__prior_cache__ = PriorCache(prior_model_type, prior_model_args or {}) if prior_model_type is not None else None

    @property
    def has_prior(self) -> bool:
        return self.__has_prior__

    def prior_save(self, f) -> None:
        if self.has_prior:
            self.__prior_cache__.save(f)

    def prior_load(self, f) -> None:
        self.__prior_cache__ = PriorCache.load(f)
        self.__has_prior__ = True


class PriorManager:
    def __init__(self, model: torch.nn.Module, prior_state: Optional[Tuple[int, Any]] = None,
                 prior_cache: Union[PriorCache, PriorDataset, None] = None, model_device: Optional[torch.device] = None,
                 store_device: Optional[torch.device] = None, training: bool = False) -> None:
        self.model, self.state = model, prior_state
        if isinstance(prior_cache, PriorDataset):
            prior_cache = prior_cache.__prior_cache__
        self.prior_cache = prior_cache if isinstance(prior_cache, PriorCache) else None
        if store_device is None and self.prior_cache is not None:
            # as in the reference (:84-85): a manager built without store_device clears the cache's one, so what the swap stores
            # stays on the model's device from then on (get_state() then says store_device 'None'; set_state maps it to cpu)
            self.prior_cache.store_device = store_device
        if model_device is None:
            model_device = next(model.parameters()).device
        self.model_device, self.training = model_device, training

    def __enter__(self) -> None:
        if self.state is None or self.prior_cache is None:
            return
        _, state = self.state
        if self.model_device != self.prior_cache.store_device:
            state = _to_device(state, self.model_device)
        PriorCache.apply_prior(self.model, state)

    def __exit__(self, exc_type, exc_value, traceback) -> bool:
        if self.state is None or self.prior_cache is None:
            return False
        key, _ = self.state
        self.prior_cache[key] = PriorCache.extract_prior(self.model)
        return False
