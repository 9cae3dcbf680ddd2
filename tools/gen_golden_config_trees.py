#!/usr/bin/env python3
"""tests/golden/config_trees.json: what the REFERENCE's own loader makes of every YAML under its config/ folder, for the keys on the
hot path.  Build container only.

For each file: `JsonConvertible.load_from_file(path, on_error='ignore')` (awesome/serialization/json_convertible.py:632-701 ->
ObjectDecoder -> configurable_object_hook, awesome/serialization/object_hook.py:60-98) - the reference's decoder, executed, not
restated - then a description of the decoded values of

    prior_model_type, prior_model_args, loss_type, loss_args, agent_args.pretrain_args, optimizer_type, optimizer_args,
    and the scalar switches the entrypoint reads (num_epochs, seed, scribble_percentage, the extra-penalty hook fields, ...)

as type names + public attribute values (no YAML text, no source).  `on_error='ignore'` leaves an object whose module cannot be
imported here (cv2 / torchvision / normflows behind the dataset loaders and the net factory) as its raw mapping: those are recorded
as {"<undecoded>": type name}.  The root `AwesomeConfig` itself is such a case (awesome.run.runner imports the dataset package), so
the root stays a mapping of decoded values - which is all the test compares.

The three packages that are not installed (`toml`, `jsonpickle`, `simple_parsing`) are replaced by the inert modules of
tools/gen_golden_boundary.py (they read a package name, pickle unknown objects, extract docstrings - none touches these values).

Identical descriptions are stored once: {"trees": {sha1: description}, "files": {relative path: sha1}}.
Usage:  python tools/gen_golden_config_trees.py [--out tests/golden/config_trees.json]
"""
import argparse
import enum
import glob
import hashlib
import inspect
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

KEYS = ("prior_model_type", "prior_model_args", "loss_type", "loss_args", "optimizer_type", "optimizer_args", "num_epochs", "seed",
        "scribble_percentage", "use_extra_penalty_hook", "extra_penalty_after_n_epochs", "use_reduce_lr_in_extra_penalty_hook",
        "reduce_lr_in_extra_penalty_hook_factor", "segmentation_training_mode", "use_segmentation_output_inversion",
        "weight_decay_on_weight_norm_modules", "dtype", "device", "use_prior_model")


def describe(obj):
    """The format of awesome_amd.serialization.describe, written independently of it (objects here are the reference's)."""
    import torch
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    if isinstance(obj, dict):
        if "__class__" in obj:      # on_error='ignore' left it undecoded
            return {"<undecoded>": str(obj["__class__"]).rsplit(".", 1)[-1]}
        return {str(k): describe(v) for k, v in obj.items()}
    if isinstance(obj, tuple):
        return {"<tuple>": [describe(v) for v in obj]}
    if isinstance(obj, list):
        return [describe(v) for v in obj]
    if isinstance(obj, (set, frozenset)):
        return {"<set>": sorted(describe(v) for v in obj)}
    if isinstance(obj, torch.dtype):
        return {"<dtype>": str(obj)}
    if isinstance(obj, type) or inspect.isfunction(obj):
        return {"<type>": obj.__qualname__}
    if isinstance(obj, enum.Enum):
        return {"<enum>": type(obj).__qualname__, "value": obj.value}
    fields = {k: v for k, v in vars(obj).items() if not k.startswith("_") and k not in ("training", "logger", "tracker")}
    return dict({"<class>": type(obj).__qualname__}, **{k: describe(v) for k, v in sorted(fields.items())})


def main(out):
    import gen_golden_boundary as GB
    from gen_golden import REF
    GB._install_inert_modules()
    sys.path.insert(0, REF)
    for sub in ("model", "dataset"):      # the eager package __init__ files pull cv2 / torchvision
        pkg = types.ModuleType(f"awesome.{sub}")
        pkg.__path__ = [os.path.join(REF, "awesome", sub)]
        sys.modules[f"awesome.{sub}"] = pkg
    import logging
    logging.disable(logging.CRITICAL)     # on_error='ignore' logs a traceback per object it leaves alone
    from awesome.serialization.json_convertible import JsonConvertible
    files = sorted(glob.glob(os.path.join(REF, "config", "**", "*.yaml"), recursive=True))
    trees, index = {}, {}
    for path in files:
        cfg = JsonConvertible.load_from_file(path, on_error="ignore")
        if isinstance(cfg, dict) and list(cfg) == ["AwesomeConfig"]:
            cfg = cfg["AwesomeConfig"]
        get = (lambda k: cfg.get(k)) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k, None))
        d = {k: describe(get(k)) for k in KEYS}
        d["pretrain_args"] = describe((get("agent_args") or {}).get("pretrain_args"))
        d["agent_switches"] = {k: v for k, v in describe(get("agent_args") or {}).items() if k != "pretrain_args"}
        blob = json.dumps(d, sort_keys=True)
        h = hashlib.sha1(blob.encode()).hexdigest()[:12]
        trees[h] = d
        index[os.path.relpath(path, os.path.join(REF, "config"))] = h
    with open(out, "w") as f:
        json.dump({"generator": "tools/gen_golden_config_trees.py", "n_files": len(index), "trees": trees, "files": index}, f,
                  indent=0, sort_keys=True)
    print(f"{len(index)} configs, {len(trees)} distinct trees -> {out} ({os.path.getsize(out) // 1024} KB)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden", "config_trees.json"))
    main(ap.parse_args().out)
