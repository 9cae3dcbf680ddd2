"""The config object of the entrypoint: the fields of the reference's `AwesomeConfig` (awesome/run/awesome_config.py:67-245, base
`Config` awesome/run/config.py:11-36) that the hot path reads, with the reference's defaults, loaded from the reference's own
file format (`__class__`-tagged YAML / JSON, `awesome_amd.serialization`).

    cfg = AwesomeConfig.load_from_file("config/...yaml")       # json_convertible.py:632-662
    loss = cfg.build_loss()                                    # awesome_runner.py:256-264: loss_type(**loss_args)
    prior = cfg.build_prior_model()                            # awesome_runner.py:218-236

Fields this build has no consumer for (plot / CRF / tensorboard switches, the semantic-soft-segmentation paths) are kept as plain
attributes so that a loaded file round-trips; nothing is dropped."""
from __future__ import annotations

import copy
import inspect
from typing import Any, Dict, Optional

from .. import serialization as S


def _default_loss_args() -> Dict[str, Any]:
    import torch
    return {"criterion": torch.nn.BCELoss(), "alpha": 1.0}          # awesome_config.py:58-62


_DEFAULTS = {
    # Config (run/config.py:15-36)
    "name_experiment": "Test", "runs_path": "./runs/", "output_folder": None, "diff_config": None, "use_progress_bar": True,
    "run_script_path": None, "used_runner_type": None, "seed": 42,
    # AwesomeConfig (run/awesome_config.py:72-245)
    "agent_args": dict, "dataset_type": "awesome.dataset.sisbosi_dataset.SISBOSIDataset", "dataset_args": dict,
    "combined_segmentation_module_type": "awesome.model.wrapper_module.WrapperModule", "combined_segmentation_module_args": dict,
    "segmentation_model_type": "awesome.model.cnn_net.CNNNet", "segmentation_model_args": dict,
    "segmentation_training_mode": "multi", "segmentation_model_gets_targets": False, "segmentation_model_state_dict_path": None,
    "use_segmentation_output_inversion": False,
    "prior_model_type": "awesome.model.convex_net.ConvexNet", "prior_model_args": dict, "use_prior_model": True,
    "loss_type": "awesome.measures.awesome_loss.AwesomeLoss", "loss_args": _default_loss_args,
    "use_extra_penalty_hook": False, "extra_penalty_after_n_epochs": 200, "use_reduce_lr_in_extra_penalty_hook": False,
    "reduce_lr_in_extra_penalty_hook_factor": 0.05,
    "optimizer_type": "torch.optim.adam.Adam",
    "optimizer_args": lambda: {"lr": 0.02, "betas": (0.9, 0.999), "eps": 1e-08, "weight_decay": 0, "amsgrad": False},
    "weight_decay_on_weight_norm_modules": 5e-5, "split_params_in_param_groups": False,
    "device": "cuda", "dtype": "torch.float32",
    "use_lr_on_plateau_scheduler": False, "lr_on_plateau_scheduler_args": dict, "use_step_lr_scheduler": False,
    "step_lr_scheduler_args": lambda: {"gamma": 0.1, "step_size": 100, "verbose": True},
    "use_lr_stop_training_watchdog": True,
    "lr_stop_training_watchdog_args": lambda: {"learning_rate": 1e-7, "mode": "lte", "verbose": True},
    "num_epochs": 800, "scribble_percentage": 0.8, "use_binary_classification": True, "validation_each_nth_epoch": 100,
}


class AwesomeConfig:
    def __init__(self, **fields: Any):
        for k, d in _DEFAULTS.items():
            setattr(self, k, d() if callable(d) else copy.deepcopy(d))
        self.explicit = set(fields)          # what the file / caller actually set (defaults of the reference fill the rest)
        for k, v in fields.items():
            setattr(self, k, v)

    # ---- dict-style access for the entrypoint ----
    def get(self, key: str, default: Any = None) -> Any:
        return getattr(self, key, default)

    def __getitem__(self, key: str) -> Any:
        return getattr(self, key)

    def __setitem__(self, key: str, value: Any) -> None:
        self.explicit.add(key)
        setattr(self, key, value)

    def __contains__(self, key: str) -> bool:
        return hasattr(self, key)

    def to_dict(self) -> Dict[str, Any]:
        return {k: v for k, v in vars(self).items() if k != "explicit"}

    # ---- file format ----
    @classmethod
    def load_from_file(cls, path: str) -> "AwesomeConfig":
        return cls(**S.load_config_file(path))

    def to_tagged_dict(self) -> Dict[str, Any]:
        """`{AwesomeConfig: {__class__: ..., **fields}}` - what the reference's `save_to_file` writes (json_convertible.py:614-629)."""
        body = {S.CLASS_KEY: "awesome.run.awesome_config.AwesomeConfig"}
        body.update({k: S.encode(v) for k, v in sorted(self.to_dict().items())})
        return {"AwesomeConfig": body}

    def save_to_file(self, path: str) -> str:
        import yaml
        with open(path, "w") as f:
            yaml.safe_dump(self.to_tagged_dict(), f)
        return path

    # ---- what the runner builds from the fields (awesome/run/awesome_runner.py) ----
    def torch_dtype(self):
        """awesome_runner.py:243-253 `_get_dtype`."""
        import torch
        d = self.dtype
        if isinstance(d, str):
            if not d.startswith("torch."):
                raise ValueError(f"Invalid dtype: {d}")
            d = getattr(torch, d[6:], None)
        return d if isinstance(d, torch.dtype) else torch.float32

    def loss_class(self):
        return self.loss_type if not isinstance(self.loss_type, str) else S.resolve_type(self.loss_type, "loss_type")

    def build_loss(self):
        """awesome_runner.py:256-264: `loss_type(**loss_args)`, with `scribble_percentage` handed to a loss that takes it."""
        import torch
        if self.loss_type is None:
            return None
        loss_type = self.loss_class()
        args = dict(self.loss_args or {})
        crit = args.get("criterion")
        if isinstance(crit, str):          # alias of this repo's first YAMLs: a bare dotted type name
            args["criterion"] = S.resolve_type(crit, "loss_args.criterion")()
        if "scribble_percentage" in inspect.signature(loss_type.__init__).parameters:
            args["scribble_percentage"] = self.scribble_percentage
        if isinstance(loss_type, type) and issubclass(loss_type, torch.nn.modules.loss._Loss):
            args = {k: v for k, v in args.items() if k in inspect.signature(loss_type.__init__).parameters}
        return loss_type(**args)

    def prior_model_factory(self):
        """awesome_runner.py:218-224: a class or a factory function returning an nn.Module."""
        t = self.prior_model_type
        return t if not isinstance(t, str) else S.resolve_type(t, "prior_model_type")

    def build_prior_model(self):
        import torch
        model = self.prior_model_factory()(**dict(self.prior_model_args or {}))
        if not isinstance(model, torch.nn.Module):
            raise ValueError(f"prior_model_type must give a torch module, got {type(model).__name__}")
        return model

    def pretrain_args(self) -> Dict[str, Any]:
        return dict((self.agent_args or {}).get("pretrain_args") or {})

    def optimizer_name(self) -> str:
        t = self.optimizer_type
        return (t if isinstance(t, str) else t.__name__).rsplit(".", 1)[-1].lower()
