"""Decoder (and encoder) of the reference's `__class__`-tagged config objects, for the types on the hot path.

The reference writes every config as the JSON/YAML image of its `AwesomeConfig` object: each nested object is a mapping
`{__class__: dotted.Type, **fields}` (awesome/serialization/json_convertible.py:632-727 `load_from_file` -> `ObjectDecoder` ->
`configurable_object_hook`, awesome/serialization/object_hook.py:18-98).  All 207 YAMLs under the reference's `config/` nest their
criteria that way, e.g. `agent_args.pretrain_args.criterion: {__class__: ...UnariesConversionLoss, criterion: {__class__: ...SE}}`.

`decode` follows the hook's rules for the classes this build mirrors:

* inner objects first, then outwards (object_decoder.py: "applies the hook to inner objects first");
* an `Enum` subclass is rebuilt from its `value` (object_hook.py:74-75);
* any other class: construct, then `setattr` the remaining serialised fields, then `after_decoding()` if the class has one
  (object_hook.py:77-91).  The reference passes only the constructor's REQUIRED arguments and sets the rest afterwards; the mirrors
  here take their fields as keyword arguments with the same names, so every serialised field the constructor knows is passed to it
  (its validation runs) and only the others are set as attributes;
* value wrappers turn into python values through their `to_python` (rules/json_tuple_serialization_rule.py:9-21 tuple,
  rules/json_set_serialization_rule.py:8-22 set, rules/json_type_serialization_rule.py:9-21 type,
  rules/torch/json_torch_dtype_serialization_rule.py:12-24 torch.dtype);
* a serialised `torch.nn.Module` (BCELoss, TV, ...) carries its private dictionaries (`_backward_hooks`, `_buffers`, `_modules`, ...):
  they are dropped, the module is rebuilt from its public fields; a non-empty buffer / parameter / sub-module would be state this
  decoder cannot restore and raises.

What it does NOT do is guess: a tag that has no mirror raises `UnmappedClassError` when it sits under an in-scope key
(`prior_model_args`, `loss_args`, `agent_args`, `optimizer_args`, ...) and is kept as an `OpaqueObject` (type name + decoded fields,
nothing deleted) under a key whose consumer is out of this build's scope (`dataset_args.dataset`: the FBMS / SISBOSI loaders read
files that are not in the image; `segmentation_model_args`).  The `{type: ..., args: {...}}` nesting of this repo's own early YAMLs is
accepted as an alias of `{__class__: type, **args}`.
"""
from __future__ import annotations

import enum
import importlib
import inspect
from typing import Any, Dict, Iterable, Optional

CLASS_KEY = "__class__"

# reference type -> the mirror in this build (classes and factory functions alike)
ALIASES: Dict[str, str] = {
    "awesome.model.convex_net.ConvexNextNet": "awesome_amd.model.ConvexNextNet",
    "awesome.model.convex_net.ConvexNet": "awesome_amd.model.ConvexNet",
    "awesome.model.convex_diffeomorphism_net.ConvexDiffeomorphismNet": "awesome_amd.model.ConvexDiffeomorphismNet",
    "awesome.model.net_factory.real_nvp_path_connected_net": "awesome_amd.model.real_nvp_path_connected_net",
    "awesome.model.path_connected_net.PathConnectedNet": "awesome_amd.model.PathConnectedNet",
    "awesome.model.noisy_path_connected_net.NoisyPathConnectedNet": "awesome_amd.model.NoisyPathConnectedNet",
    "awesome.model.fc_net.FCNet": "awesome_amd.model.FCNet",
    "awesome.model.forward_module.ForwardModule": "awesome_amd.model.ForwardModule",
    "awesome.model.wrapper_module.WrapperModule": "awesome_amd.model.WrapperModule",
    "awesome.model.zoo.Zoo": "awesome_amd.model.Zoo",
    "awesome.measures.se.SE": "awesome_amd.measures.SE",
    "awesome.measures.weighted_loss.WeightedLoss": "awesome_amd.measures.WeightedLoss",
    "awesome.measures.unaries_weighted_loss.UnariesWeightedLoss": "awesome_amd.measures.UnariesWeightedLoss",
    "awesome.measures.unaries_conversion_loss.UnariesConversionLoss": "awesome_amd.measures.UnariesConversionLoss",
    "awesome.measures.fbms_joint_loss.FBMSJointLoss": "awesome_amd.measures.FBMSJointLoss",
    "awesome.measures.awesome_loss.AwesomeLoss": "awesome_amd.measures.AwesomeLoss",
    "awesome.measures.awesome_image_loss.AwesomeImageLoss": "awesome_amd.measures.AwesomeImageLoss",
    "awesome.measures.awesome_loss_joint.AwesomeLossJoint": "awesome_amd.measures.AwesomeLossJoint",
    "awesome.measures.awesome_image_loss_joint.AwesomeImageLossJoint": "awesome_amd.measures.AwesomeImageLossJoint",
    "awesome.measures.regularizer_loss.RegularizerLoss": "awesome_amd.measures.RegularizerLoss",
    "awesome.measures.gradient_penalty_loss.GradientPenaltyLoss": "awesome_amd.measures.GradientPenaltyLoss",
    "awesome.measures.tv.TV": "awesome_amd.measures.TV",
    "awesome.measures.miou.MIOU": "awesome_amd.measures.MIOU",
    "awesome.run.awesome_config.AwesomeConfig": "awesome_amd.run.config.AwesomeConfig",
}

# value wrappers of the reference's serialisation rules: (field holding the payload, conversion)
_WRAPPERS = {
    "awesome.serialization.rules.json_tuple_serialization_rule.TupleValueWrapper": ("value", "tuple"),
    "awesome.serialization.rules.json_set_serialization_rule.SetValueWrapper": ("values", "set"),
    "awesome.serialization.rules.json_type_serialization_rule.TypeValueWrapper": ("value", "type"),
    "awesome.serialization.rules.torch.json_torch_dtype_serialization_rule.TorchDtypeValueWrapper": ("value", "dtype"),
}

# modules a config may name directly (they are not the reference's own code)
_FOREIGN_PREFIXES = ("torch.", "awesome_amd.")

# top-level AwesomeConfig keys whose consumers are built here: an unmapped tag below them is an error, not an opaque record
IN_SCOPE_KEYS = ("prior_model_args", "loss_args", "agent_args", "optimizer_args", "combined_segmentation_module_args",
                 "lr_on_plateau_scheduler_args", "step_lr_scheduler_args", "lr_stop_training_watchdog_args", "dtype")


class UnmappedClassError(ValueError):
    """An in-scope `__class__` tag without a mirror in this build (the reference would have instantiated it)."""

    def __init__(self, class_name: str, path: str):
        super().__init__(f"config object at '{path}' has __class__ {class_name}, which this build does not implement; "
                         f"refusing to drop it (the reference instantiates it: awesome/serialization/object_hook.py:60-98)")
        self.class_name, self.path = class_name, path


class OpaqueObject:
    """A tagged object outside this build's scope (dataset loaders, ...): its type name and decoded fields, kept, not instantiated."""

    def __init__(self, class_name: str, fields: Dict[str, Any]):
        self.class_name, self.fields = class_name, fields

    def __repr__(self) -> str:
        return f"OpaqueObject({self.class_name}, {sorted(self.fields)})"

    def __eq__(self, other) -> bool:
        return isinstance(other, OpaqueObject) and (self.class_name, self.fields) == (other.class_name, other.fields)


def dynamic_import(path: str):
    """awesome/util/reflection.py dynamic_import with the reference's own type names mapped to the mirrors."""
    path = ALIASES.get(path, path)
    mod, _, name = path.rpartition(".")
    if not mod:
        raise ImportError(f"'{path}' is not a dotted type name")
    return getattr(importlib.import_module(mod), name)


def resolve_type(class_name: str, path: str = ""):
    """The python type a tag names: a mirror (ALIASES), a torch / awesome_amd type as it is; anything else is unmapped."""
    if class_name in ALIASES or class_name.startswith(_FOREIGN_PREFIXES):
        try:
            return dynamic_import(class_name)
        except (ImportError, AttributeError) as err:
            raise UnmappedClassError(class_name, path) from err
    raise UnmappedClassError(class_name, path)


def _unwrap(kind: str, payload: Any, path: str):
    if kind == "tuple":
        return tuple(payload)
    if kind == "set":
        return set(payload)
    if kind == "type":
        return resolve_type(str(payload), path)
    if kind == "dtype":
        import torch
        name = str(payload)
        if not name.startswith("torch.") or not isinstance(getattr(torch, name[6:], None), torch.dtype):
            raise ValueError(f"'{path}': {name!r} is not a torch dtype")
        return getattr(torch, name[6:])
    raise AssertionError(kind)


def _build_module(cls, fields: Dict[str, Any], path: str):
    """A serialised torch.nn.Module: public fields only; state the decoder cannot restore raises."""
    for k in ("_parameters", "_modules"):
        if fields.get(k):
            raise ValueError(f"'{path}': serialised {cls.__name__} carries {k} = {sorted(fields[k])}; only stateless criteria can be decoded")
    if any(v is not None for v in (fields.get("_buffers") or {}).values()):
        raise ValueError(f"'{path}': serialised {cls.__name__} carries non-empty buffers (e.g. a class weight), which are not restored")
    public = {k: v for k, v in fields.items() if not k.startswith("_")}
    training = public.pop("training", True)
    params = inspect.signature(cls.__init__).parameters
    init = {k: public.pop(k) for k in list(public) if k in params}
    obj = cls(**init)
    for k, v in public.items():
        setattr(obj, k, v)
    obj.train(bool(training))
    return obj


def _build_object(cls, fields: Dict[str, Any], path: str):
    import torch
    if isinstance(cls, type) and issubclass(cls, enum.Enum):
        return cls(fields.get("value"))
    if isinstance(cls, type) and issubclass(cls, torch.nn.Module) and "_modules" in fields:
        return _build_module(cls, fields, path)
    sig = inspect.signature(cls)
    names = set(sig.parameters)
    init = {k: v for k, v in fields.items() if k in names}
    rest = {k: v for k, v in fields.items() if k not in names}
    obj = cls(**init)
    for k, v in rest.items():            # object_hook.py:83-84
        setattr(obj, k, v)
    fn = getattr(obj, "after_decoding", None)
    if callable(fn):
        fn()
    return obj


def _is_type_args_alias(obj: Dict[str, Any]) -> bool:
    return "type" in obj and isinstance(obj["type"], str) and set(obj) <= {"type", "args"} and "." in obj["type"]


def decode(obj: Any, in_scope: bool = True, path: str = "") -> Any:
    """Recursive decode of a parsed YAML / JSON tree (inner objects first)."""
    if isinstance(obj, list):
        return [decode(v, in_scope, f"{path}[{i}]") for i, v in enumerate(obj)]
    if not isinstance(obj, dict):
        return obj
    if CLASS_KEY not in obj and _is_type_args_alias(obj):
        obj = dict(obj.get("args") or {}, **{CLASS_KEY: obj["type"]})
    fields = {k: decode(v, in_scope, f"{path}.{k}" if path else str(k)) for k, v in obj.items() if k != CLASS_KEY}
    if CLASS_KEY not in obj:
        return fields
    class_name = str(obj[CLASS_KEY])
    if class_name in _WRAPPERS:
        key, kind = _WRAPPERS[class_name]
        return _unwrap(kind, fields.get(key), path)
    try:
        cls = resolve_type(class_name, path)
    except UnmappedClassError:
        if in_scope:
            raise
        return OpaqueObject(class_name, fields)
    return _build_object(cls, fields, path)


def decode_config(tree: Dict[str, Any]) -> Dict[str, Any]:
    """The root mapping of a config file -> {field: decoded value}.  Accepts the reference's `{AwesomeConfig: {...}}` wrapping
    (json_convertible.py:614-629 writes `{type name: dict}`, :694-700 unwraps it) and a flat mapping."""
    if isinstance(tree, dict) and len(tree) == 1:
        (name, inner), = tree.items()
        if isinstance(inner, dict) and str(inner.get(CLASS_KEY, name)).rsplit(".", 1)[-1] == name:
            tree = inner
    if not isinstance(tree, dict):
        raise ValueError("a config file must hold a mapping")
    out = {}
    for k, v in tree.items():
        if k == CLASS_KEY:
            continue
        out[k] = decode(v, in_scope=k in IN_SCOPE_KEYS, path=str(k))
    return out


def load_config_file(path: str) -> Dict[str, Any]:
    """YAML or JSON by extension (json_convertible.py:652-662)."""
    import json
    import os
    ext = os.path.basename(path).rsplit(os.path.extsep, 1)[-1].lower()
    with open(path) as f:
        if ext == "json":
            tree = json.load(f)
        elif ext in ("yaml", "yml"):
            import yaml
            tree = yaml.safe_load(f)
        else:
            raise ValueError(f"Unsupported file extension: {ext} Only json and yaml / yml are supported!")
    return decode_config(tree)


# ------------------------------------------------------------------------------------------------------------------------------
# the other direction: objects -> tagged trees in the reference's naming (what `AwesomeConfig.save_to_file` writes), so that a
# config assembled in python can be handed to scripts/run.py - or to the reference - as a file

_REVERSE = {v: k for k, v in ALIASES.items()}


def reference_class_name(cls) -> str:
    full = f"{cls.__module__}.{cls.__qualname__}"
    for mirror, ref in _REVERSE.items():
        mod, _, name = mirror.rpartition(".")
        if name == cls.__qualname__ and (cls.__module__ == mod or cls.__module__.startswith(mod + ".")):
            return ref
    return full


def encode(obj: Any) -> Any:
    """Tagged tree of `obj` (public attributes; tuples / sets / dtypes / types through the reference's wrappers)."""
    import torch
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    if isinstance(obj, dict):
        return {str(k): encode(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [encode(v) for v in obj]
    inv = {v[1]: (k, v[0]) for k, v in _WRAPPERS.items()}
    if isinstance(obj, tuple):
        name, key = inv["tuple"]
        return {CLASS_KEY: name, key: [encode(v) for v in obj]}
    if isinstance(obj, (set, frozenset)):
        name, key = inv["set"]
        return {CLASS_KEY: name, key: [encode(v) for v in sorted(obj, key=repr)]}
    if isinstance(obj, torch.dtype):
        name, key = inv["dtype"]
        return {CLASS_KEY: name, key: str(obj)}
    if isinstance(obj, type) or inspect.isfunction(obj):
        name, key = inv["type"]
        return {CLASS_KEY: name, key: reference_class_name(obj)}
    if isinstance(obj, enum.Enum):
        return {CLASS_KEY: reference_class_name(type(obj)), "value": obj.value}
    if isinstance(obj, OpaqueObject):
        return dict({CLASS_KEY: obj.class_name}, **encode(obj.fields))
    if isinstance(obj, torch.nn.Module):
        fields = {k: v for k, v in vars(obj).items() if not k.startswith("_")}
        if any(True for _ in obj.parameters()) or any(True for _ in obj.buffers()):
            raise ValueError(f"{type(obj).__name__} holds parameters / buffers; only stateless modules are config values")
    else:
        fields = {k: v for k, v in vars(obj).items() if not k.startswith("_")}
    ignore: Iterable[str] = getattr(obj, "__ignore_on_iter__", lambda: ())()
    return dict({CLASS_KEY: reference_class_name(type(obj))}, **{k: encode(v) for k, v in fields.items() if k not in set(ignore)})


def describe(obj: Any) -> Any:
    """Type names + public values of a decoded tree, as plain JSON (what tests/golden/config_trees.json holds)."""
    import torch
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    if isinstance(obj, dict):
        return {str(k): describe(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return {"<tuple>": [describe(v) for v in obj]} if isinstance(obj, tuple) else [describe(v) for v in obj]
    if isinstance(obj, (set, frozenset)):
        return {"<set>": sorted(describe(v) for v in obj)}
    if isinstance(obj, torch.dtype):
        return {"<dtype>": str(obj)}
    if isinstance(obj, type) or inspect.isfunction(obj):
        return {"<type>": obj.__qualname__}
    if isinstance(obj, enum.Enum):
        return {"<enum>": type(obj).__qualname__, "value": obj.value}
    if isinstance(obj, OpaqueObject):
        return {"<opaque>": obj.class_name.rsplit(".", 1)[-1]}
    fields = {k: v for k, v in vars(obj).items() if not k.startswith("_") and k not in ("training", "logger", "tracker")}
    return dict({"<class>": type(obj).__qualname__}, **{k: describe(v) for k, v in sorted(fields.items())})
