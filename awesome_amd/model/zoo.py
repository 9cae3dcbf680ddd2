"""Cache of pre-trained model states, keyed by what was trained and how - the role of awesome/model/zoo.py:188-318 on this path:
`PathConnectedNet.learn_flow_identity(..., zoo=zoo)` (path_connected_net.py:155-250) asks the zoo for a flow that already maps
the grid to itself before it spends `max_iter` optimizer steps on it, and stores the result afterwards, so the identity pre-fit
is paid once per (architecture, grid, hyper-parameters) instead of once per image / retry.

Same interface (`load_model_state` / `save_model_state` with a name, the model and a config dict; the key is a hash of the
name, `repr(model)` and the config - NOT of the weights, exactly like the reference: a hit overrides whatever initialisation
the model carries) with a plain on-disk format (one `torch.save` file per entry); optional in-memory only (`zoo_folder=None`)."""
from __future__ import annotations

import hashlib
import json
import os
from typing import Any, Dict, Optional, Tuple

import torch


def tensor_hash(t: torch.Tensor) -> str:
    """Content hash of a tensor (the reference's TensorUtil.to_hash in the zoo config of learn_flow_identity)."""
    b = t.detach().to("cpu", torch.float32).contiguous().numpy().tobytes()
    return hashlib.sha256(b).hexdigest()


def _canonical(o: Any) -> Any:
    if isinstance(o, torch.Tensor):
        return {"__tensor__": tensor_hash(o)}
    if isinstance(o, dict):
        return {str(k): _canonical(v) for k, v in sorted(o.items(), key=lambda kv: str(kv[0]))}
    if isinstance(o, (list, tuple)):
        return [_canonical(v) for v in o]
    if isinstance(o, (int, float, str, bool)) or o is None:
        return o
    return repr(o)   # criterion objects and the like: their printed form


class Zoo:
    def __init__(self, zoo_folder: Optional[str] = "./data/zoo", **kwargs):
        self.zoo_folder = zoo_folder
        self._mem: Dict[str, Dict[str, Any]] = {}      # the folder is created by the first save, not by decoding a config

    @staticmethod
    def compute_query_hash(name: str, str_repr: str, config: Optional[Dict[str, Any]]) -> str:
        s = name + "\n" + str_repr + "\n" + json.dumps(_canonical(config or {}), sort_keys=True)
        return hashlib.sha256(s.encode("utf-8")).hexdigest()

    def _path(self, query_hash: str) -> Optional[str]:
        return None if self.zoo_folder is None else os.path.join(self.zoo_folder, query_hash + ".pth")

    def query_state(self, name: str, model: torch.nn.Module, config: Optional[Dict[str, Any]] = None):
        h = self.compute_query_hash(name, repr(model), config)
        if h in self._mem:
            e = self._mem[h]
            return e["state_dict"], e["context"]
        p = self._path(h)
        if p is not None and os.path.isfile(p):
            e = torch.load(p, map_location="cpu", weights_only=True)   # tensors, strings, numbers only: nothing is unpickled
            self._mem[h] = e
            return e["state_dict"], e["context"]
        return None, None

    def load_model_state(self, name: str, model: torch.nn.Module, config: Optional[Dict[str, Any]] = None) -> Tuple[bool, Optional[Dict[str, Any]]]:
        """True + the stored context if an entry exists (its state is loaded into `model`), else (False, None)."""
        state, context = self.query_state(name, model, config)
        if state is None:
            return False, None
        model.load_state_dict(state)
        return True, context

    def save_model_state(self, name: str, model: torch.nn.Module, config: Optional[Dict[str, Any]] = None,
                         context: Optional[Dict[str, Any]] = None) -> None:
        h = self.compute_query_hash(name, repr(model), config)
        e = {"name": name, "str_repr": repr(model), "config": _canonical(config or {}), "query_hash": h,
             "state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, "context": dict(context or {})}
        self._mem[h] = e
        p = self._path(h)
        if p is not None:
            # ranks of one job can reach the same entry at once: write to a private temp file, then rename (atomic on POSIX)
            tmp = f"{p}.{os.getpid()}.tmp"
            os.makedirs(self.zoo_folder, exist_ok=True)
            torch.save(e, tmp)
            os.replace(tmp, p)
