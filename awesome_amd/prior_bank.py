"""Device-resident prior cache for the joint-training path.

The reference keeps one `state_dict` per image in `PriorCache` (awesome/util/prior_cache.py:9-59) and swaps it into the single
prior model around every training step with `PriorManager` (awesome/dataset/prior_dataset.py:70-110): `load_state_dict` on
enter, `deepcopy(state_dict())` on exit, plus host<->device copies when `store_device` differs - per image, per step.

Here all parameter sets of a prior model live in ONE `(n_images, P)` tensor in HBM (the flat layout of include/inrfit.h) and the
swap is an index: `with bank.manager(model, key):` re-points the model's parameters at row `key` (views into the bank; nothing is
copied, what the step writes is already "stored").  The same tensor is what `awesome_amd.fit` / `BatchedPriorFitter` update in
place, so per-image pre-fits and joint steps share one resident copy.  `get_state()` writes the dict `PriorCache.get_state()`
holds, so the reference's notebooks can read the result.
"""
from __future__ import annotations

import contextlib
import json
from typing import Any, Callable, Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import torch

from . import icnn as K


def _ordered_parameters(model: torch.nn.Module) -> List[torch.nn.Parameter]:
    """The model's parameters in the order of its flat vector."""
    if hasattr(model, "_ordered_params"):
        try:
            op = model._ordered_params()
        except NotImplementedError:   # a composite without a fused form (ConvexDiffeomorphismNet with the 'resnet' flow backbone)
            return list(model.parameters())
        if isinstance(op, tuple) and len(op) == 4:   # PathConnectedNet / ConvexDiffeomorphismNet: (ispec, fspec, icnn, flow)
            return list(op[2]) + list(op[3])
        return list(op)
    return list(model.parameters())


class PriorBank:
    def __init__(self, model_factory: Callable[[], torch.nn.Module], n_images: int, device="cuda:0",
                 keys: Optional[Sequence[Any]] = None):
        """`model_factory()` builds one prior model (its fresh initialisation is the prior of a key, like
        PriorCache.generate_prior, prior_cache.py:29-32); rows are generated lazily, at the first access of a key."""
        self.model_factory = model_factory
        probe = model_factory()
        self._shapes: List[Tuple[int, ...]] = [tuple(p.shape) for p in _ordered_parameters(probe)]
        self._numels = [int(torch.Size(s).numel()) for s in self._shapes]
        self.P = int(sum(self._numels))
        self.device = torch.device(device)
        self.params = torch.zeros(int(n_images), self.P, dtype=torch.float32, device=self.device)
        self._keys: Dict[Any, int] = {}
        self._generated = [False] * int(n_images)
        if keys is not None:
            for k in keys:
                self.index_of(k)
        self._probe_type = f"{type(probe).__module__}.{type(probe).__name__}"

    # -- key <-> row ------------------------------------------------------------------------------------------------------
    def __len__(self) -> int:
        return self.params.shape[0]

    def __contains__(self, key: Any) -> bool:
        return key in self._keys and self._generated[self._keys[key]]

    def index_of(self, key: Any) -> int:
        if key not in self._keys:
            if len(self._keys) >= len(self):
                raise KeyError(f"prior bank is full ({len(self)} rows); key {key!r} has no row")
            self._keys[key] = len(self._keys)
        return self._keys[key]

    def row(self, key: Any) -> torch.Tensor:
        """The flat parameter vector of `key` (a view: writing it updates the bank), generated on first access."""
        i = self.index_of(key)
        if not self._generated[i]:
            fresh = self.model_factory()
            flat = torch.cat([p.detach().reshape(-1).to(torch.float32) for p in _ordered_parameters(fresh)])
            self.params[i].copy_(flat.to(self.device))
            self._generated[i] = True
        return self.params[i]

    def rows(self, keys: Iterable[Any]) -> torch.Tensor:
        """(len(keys), P) gather of the rows (a copy; use `scatter` to write a fitted batch back)."""
        return torch.stack([self.row(k) for k in keys])

    def scatter(self, keys: Iterable[Any], params: torch.Tensor) -> None:
        for j, k in enumerate(keys):
            self.row(k).copy_(params[j])

    # -- the swap ---------------------------------------------------------------------------------------------------------
    def bind(self, model: torch.nn.Module, key: Any) -> None:
        """Re-point the model's parameters at row `key`: views, no copy.  Gradients accumulate in the parameters as usual; an
        optimizer built over `model.parameters()` keeps working (it holds the Parameter objects, not their storage)."""
        row = self.row(key)
        params = _ordered_parameters(model)
        if [tuple(p.shape) for p in params] != self._shapes:
            raise ValueError("model does not have the parameter shapes this bank was built for")
        off = 0
        for p, n, shp in zip(params, self._numels, self._shapes):
            p.data = row[off:off + n].view(shp)
            off += n

    @contextlib.contextmanager
    def manager(self, model: torch.nn.Module, key: Any) -> Iterator[torch.nn.Module]:
        """Drop-in for `with PriorManager(model, prior_state=(key, state), prior_cache=cache):` (prior_dataset.py:96-110).
        Enter: the model computes with the prior of `key`.  Exit: nothing to do - the step already wrote into the bank."""
        self.bind(model, key)
        yield model

    # -- export in the reference's format -----------------------------------------------------------------------------------
    def state_dict(self, key: Any) -> Dict[str, torch.Tensor]:
        """Detached copy of the prior of `key` under the model's own parameter names (what PriorCache.extract_prior stores)."""
        probe = self.model_factory()
        names = [n for n, _ in probe.named_parameters()]
        order = {id(p): n for n, p in probe.named_parameters()}
        out: Dict[str, torch.Tensor] = {}
        row = self.row(key).detach().cpu()
        off = 0
        for p, n, shp in zip(_ordered_parameters(probe), self._numels, self._shapes):
            out[order[id(p)]] = row[off:off + n].view(shp).clone()
            off += n
        return {n: out[n] for n in names}   # the model's own key order

    def get_state(self, model_type: Optional[str] = None, model_args: Optional[dict] = None) -> Dict[str, Any]:
        """The dict PriorCache.get_state() returns / `prior_cache_epoch_N.pth` holds (prior_cache.py:61-71)."""
        cache = {str(k): self.state_dict(k) for k, i in self._keys.items() if self._generated[i]}
        return {"model_type": model_type or self._probe_type, "model_args": json.dumps(model_args or {}),
                "store_device": "cpu", "cache": cache}

    # -- per-image pre-fit on the resident rows (the "sequential" path feeding the joint one) --------------------------------
    def fit(self, keys: Sequence[Any], grid: "K.Grid", unaries: torch.Tensor, steps: int, **fit_kwargs) -> "K.FitResult":
        """Fit the priors of `keys` to `unaries` (len(keys), N) with the fused HIP loop; the bank rows are updated in place."""
        probe = self.model_factory()
        spec = getattr(probe, "spec", None)
        if spec is None or not hasattr(probe, "_ordered_params"):
            raise TypeError("PriorBank.fit needs an ICNN prior model (ConvexNet / ConvexNextNet): rows = the C ABI's flat vector")
        params = self.rows(keys).contiguous()
        res = K.fit(spec, params, grid, unaries, steps, **dict(getattr(probe, "fit_options", {}), **fit_kwargs))
        self.scatter(keys, res.params)
        return res
