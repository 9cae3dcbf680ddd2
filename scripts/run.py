#!/usr/bin/env python3
"""scripts/run.py - config entrypoint for the MI355X hot path, with the reference's surface:

    python scripts/run.py --config-path config/c2_blob256.yaml [--num-epochs N --seed S --device cuda:0 ...]

Mirrors jp-schneider/awesome scripts/run.py:29-79 (argparse + YAML -> config -> runner.build() -> runner.train()).
The YAML uses the reference's schema (`AwesomeConfig:` root or flat; `__class__` keys ignored) for the fields on the
hot path: prior_model_type / prior_model_args (dotted type strings, resolved like awesome/util/reflection.py's
dynamic_import; `awesome.model.convex_net.*` names map to the drop-in modules), dataset_type / dataset_args,
loss_type / loss_args, agent_args.pretrain_args (num_epochs, lr, reuse_state...), seed, device, output_folder.
Everything else the reference runner does (tensorboard, plots, UNet joint training) is out of scope.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import yaml  # noqa: E402

ALIASES = {
    "awesome.model.convex_net.ConvexNextNet": "awesome_amd.model.ConvexNextNet",
    "awesome.model.convex_net.ConvexNet": "awesome_amd.model.ConvexNet",
    "awesome.model.convex_diffeomorphism_net.ConvexDiffeomorphismNet": "awesome_amd.model.ConvexDiffeomorphismNet",
    "awesome.model.net_factory.real_nvp_path_connected_net": "awesome_amd.model.real_nvp_path_connected_net",
    "awesome.model.path_connected_net.PathConnectedNet": "awesome_amd.model.PathConnectedNet",
    "awesome.model.fc_net.FCNet": "awesome_amd.model.FCNet",
    "awesome.measures.se.SE": "awesome_amd.measures.SE",
    "awesome.measures.unaries_weighted_loss.UnariesWeightedLoss": "awesome_amd.measures.UnariesWeightedLoss",
}


def dynamic_import(path: str):
    path = ALIASES.get(path, path)
    mod, _, name = path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def strip_class_tags(obj):
    if isinstance(obj, dict):
        return {k: strip_class_tags(v) for k, v in obj.items() if k != "__class__"}
    if isinstance(obj, list):
        return [strip_class_tags(v) for v in obj]
    return obj


def build_criterion(loss_type, loss_args):
    import torch
    if loss_type is None:
        return None
    args = dict(loss_args or {})
    crit = args.get("criterion")
    if isinstance(crit, dict):  # nested {"type": ..., "args": {...}}
        args["criterion"] = build_criterion(crit.get("type"), crit.get("args"))
    elif isinstance(crit, str):
        args["criterion"] = torch.nn.BCELoss() if crit.endswith("BCELoss") else dynamic_import(crit)()
    if loss_type.endswith("BCELoss"):
        return torch.nn.BCELoss()
    return dynamic_import(loss_type)(**args)


def get_config():
    ap = argparse.ArgumentParser(description="MI355X INR prior fit (awesome-compatible config entrypoint)")
    ap.add_argument("--config-path", type=str, required=True)
    ap.add_argument("--num-epochs", type=int, default=None)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--device", type=str, default=None)
    ap.add_argument("--output-folder", type=str, default=None)
    ap.add_argument("--name-experiment", type=str, default=None)
    args = ap.parse_args()
    with open(args.config_path) as f:
        cfg = yaml.safe_load(f)
    cfg = strip_class_tags(cfg.get("AwesomeConfig", cfg))
    for k in ("num_epochs", "seed", "device", "output_folder", "name_experiment"):
        v = getattr(args, k)
        if v is not None:
            cfg[k] = v
    return cfg


def main(cfg):
    import torch
    import awesome_amd as A
    from awesome_amd.fitter import BatchedPriorFitter
    from awesome_amd import parallel

    rank, world, local = parallel.init()
    device = torch.device(cfg.get("device", "cuda"))
    if device.type != "cuda" or not torch.cuda.is_available():
        raise SystemExit("this entrypoint drives the MI355X path; no CPU fallback exists (use the reference for CPU runs)")
    if device.index is None:
        device = torch.device("cuda", local % torch.cuda.device_count())   # (% only matters for a gloo rehearsal on one GPU)
    torch.cuda.set_device(device)
    seed = int(cfg.get("seed", 42))
    torch.manual_seed(seed)

    model_type = dynamic_import(cfg.get("prior_model_type", "awesome_amd.model.ConvexNextNet"))
    model_args = dict(cfg.get("prior_model_args") or {})
    dataset = dynamic_import(cfg.get("dataset_type", "awesome_amd.dataset.SyntheticUnariesDataset"))(**(cfg.get("dataset_args") or {}))
    pre = dict((cfg.get("agent_args") or {}).get("pretrain_args") or {})
    num_epochs = int(cfg.get("num_epochs", pre.get("num_epochs", 2000)))
    criterion = build_criterion(cfg.get("loss_type"), cfg.get("loss_args"))
    opt_type = cfg.get("optimizer_type", "torch.optim.Adamax").rsplit(".", 1)[-1].lower()
    opt_args = dict(cfg.get("optimizer_args") or {})
    mine = list(parallel.shard_range(len(dataset), rank, world))
    size = dataset.size
    unaries = dataset.batch(mine).to(device)
    probe = model_type(**model_args)
    flow_prior = hasattr(probe, "fit_images")   # ConvexDiffeomorphismNet / PathConnectedNet: ICNN behind a learned deformation
    if hasattr(dataset, "coords"):              # (x, y, t) sequence: one network over all frames
        grid = A.Grid.explicit(dataset.coords().to(device))
    else:
        grid = A.Grid.linspace(size, size, device)
    lr = float(opt_args.get("lr", pre.get("lr", 1e-3)))
    thr, retrys = float(pre.get("proper_prior_fit_threshold", 0.5)), int(pre.get("proper_prior_fit_retrys", 1))
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if not flow_prior:
        fitter = BatchedPriorFitter(lambda: model_type(**model_args), num_epochs=num_epochs, lr=lr,
                                    optimizer=opt_type, weight_decay=float(opt_args.get("weight_decay", 0.0)), criterion=criterion,
                                    plateau=pre.get("plateau", None if pre.get("use_plateau", True) else False),
                                    proper_prior_fit_threshold=thr, proper_prior_fit_retrys=retrys,
                                    reuse_state=bool(pre.get("reuse_state", False)),
                                    reuse_state_epochs=int(pre.get("reuse_state_epochs", 200)))
        rep = fitter.fit_batch(grid, unaries)
        iou, retries = rep.iou, rep.retries
        cache_state = lambda: fitter.prior_cache_state(rep, indices=mine, model_args=model_args)  # noqa: E731
    else:
        # _prior_based_pretrain / ConvexDiffeomorphismNet.pretrain semantics (path_connected_net.py:897-985): fit, IoU gate,
        # reset_parameters + full refit of the images that failed it
        from awesome_amd.measures import criterion_to_desc
        kind, wmode, _ = criterion_to_desc(criterion) if criterion is not None else ("se", "none", 1.0)
        kw = dict(num_epochs=num_epochs, lr=lr, loss=kind)
        if hasattr(probe, "flow_net"):
            kw.update(weight_mode=wmode, flow_weight_decay=float(pre.get("flow_weight_decay", 1e-5)), optimizer=opt_type)
            kw.update({k: v for k, v in pre.items() if k.startswith("prefit_")})   # the reference's pre-fit stage kwargs
            if pre.get("zoo"):   # pretrain_args.zoo: a folder (or "memory"): reuse the flow-identity pre-fit (path_connected_net.py:560)
                from awesome_amd.model import Zoo
                kw["zoo"] = Zoo(None if pre["zoo"] == "memory" else str(pre["zoo"]))
        model = probe.to(device)
        res = model.fit_images(grid, unaries, **kw)
        iou = A.miou(torch.sigmoid(res.logits), unaries)
        retries = [0] * len(mine)
        for attempt in range(retrys):
            bad = [i for i in range(len(mine)) if float(iou[i]) < thr]
            if not bad:
                break
            # the reference calls self.reset_parameters() here (path_connected_net.py:978), which re-draws the zero-initialised
            # output layers of a RealNVP flow and keeps its ActNorm statistics; a freshly built model is the same retry with a
            # well-defined start (identity deformation)
            model = model_type(**model_args).to(device)
            sub = model.fit_images(grid, unaries[bad].contiguous(), **kw)
            sub_iou = A.miou(torch.sigmoid(sub.logits), unaries[bad].contiguous())
            for n, i in enumerate(bad):
                res.icnn_params[i], res.flow_params[i], res.logits[i] = sub.icnn_params[n], sub.flow_params[n], sub.logits[n]
                iou[i] = sub_iou[n]
                retries[i] += 1

        def cache_state():
            from awesome_amd import flow as FL, rnvp as R
            cache = {}
            for n, i in enumerate(mine):
                ispec, fspec = model._specs()
                sd = {"convex_net." + k: v for k, v in A.unpack_params(ispec, res.icnn_params[n].cpu()).items()}
                un = R.unpack_rnvp_params if hasattr(model, "flow_net") else FL.unpack_flow_params
                sd.update(un(fspec, res.flow_params[n].cpu()))
                cache[str(i)] = sd
            return {"model_type": cfg.get("prior_model_type"), "model_args": json.dumps(model_args), "store_device": "cpu",
                    "cache": cache}
    torch.cuda.synchronize()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, device)
    iou_all = parallel.gather_per_image(iou, len(dataset), rank, world)
    gt_all = None
    if hasattr(dataset, "ground_truth_batch"):   # refinement configs: score the fitted prior against the clean mask too
        logits = rep.logits if not flow_prior else res.logits
        if logits is not None:
            gt = dataset.ground_truth_batch(mine).to(device)
            gt_all = parallel.gather_per_image(A.miou(torch.sigmoid(logits), gt), len(dataset), rank, world)
            noisy_all = parallel.gather_per_image(A.miou(unaries, gt), len(dataset), rank, world)
    if rank == 0:
        out_dir = os.path.join(cfg.get("output_folder", "runs"), cfg.get("name_experiment", "inr_fit"))
        os.makedirs(out_dir, exist_ok=True)
        torch.save(cache_state(), os.path.join(out_dir, "prior_cache_epoch_0.pth"))
        summary = {"images": len(dataset), "ranks": world, "epochs": num_epochs, "seconds": round(dt, 4),
                   "fits_per_s": round(len(dataset) / dt, 4), "ForegroundBinaryMIOU_vs_unaries": round(float(iou_all.mean()), 5),
                   "retries": retries, "output": out_dir}
        if gt_all is not None:
            summary["ForegroundBinaryMIOU_vs_ground_truth"] = round(float(gt_all.mean()), 5)
            summary["input_labels_MIOU_vs_ground_truth"] = round(float(noisy_all.mean()), 5)
        with open(os.path.join(out_dir, "summary.json"), "w") as f:
            json.dump(summary, f, indent=1)
        print(json.dumps(summary))
    parallel.barrier()


if __name__ == "__main__":
    main(get_config())
