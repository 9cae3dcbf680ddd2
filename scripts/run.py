#!/usr/bin/env python3
"""scripts/run.py - config entrypoint for the MI355X hot path, with the reference's surface:

    python scripts/run.py --config-path config/c2_blob256.yaml [--num-epochs N --seed S --device cuda:0 ...]

Mirrors jp-schneider/awesome scripts/run.py:29-79 (argparse + YAML -> config -> runner.build() -> runner.train()).
The config file is the reference's own format: the YAML / JSON image of an `AwesomeConfig`, every nested object a mapping
`{__class__: dotted.Type, **fields}` (awesome/serialization/json_convertible.py:632-727).  `awesome_amd.serialization` decodes the
tags into this build's mirrors (criteria, value wrappers, Zoo ...) and RAISES on an in-scope tag it has no mirror for; objects under
out-of-scope keys (`dataset_args.dataset`: loaders of files that are not in the image) are kept opaque.  Fields on the hot path:
prior_model_type / prior_model_args (dotted type strings, resolved like awesome/util/reflection.py's dynamic_import;
`awesome.model.*` names map to the drop-in modules), agent_args.pretrain_args (criterion, num_epochs, lr, reuse_state, zoo ...),
loss_type / loss_args, optimizer_type / optimizer_args, the extra-penalty hook fields, seed, device, output_folder; a reference
YAML runs with `--dataset-type / --dataset-args / --segmentation-model-type` replacing what is not in the image.
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

from awesome_amd import serialization as S  # noqa: E402
from awesome_amd.run.config import AwesomeConfig  # noqa: E402

ALIASES = S.ALIASES                 # reference type names -> the mirrors of this build
dynamic_import = S.dynamic_import


def get_config(argv=None) -> AwesomeConfig:
    ap = argparse.ArgumentParser(description="MI355X INR prior fit (awesome-compatible config entrypoint)")
    ap.add_argument("--config-path", type=str, required=True)
    ap.add_argument("--num-epochs", type=int, default=None)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--device", type=str, default=None)
    ap.add_argument("--output-folder", type=str, default=None)
    ap.add_argument("--name-experiment", type=str, default=None)
    ap.add_argument("--dataset-type", type=str, default=None, help="replaces the config's dataset_type (a reference YAML names a "
                    "loader of files that are not in this image; e.g. awesome_amd.dataset.SyntheticUnariesDataset)")
    ap.add_argument("--dataset-args", type=str, default=None, help="JSON dict; merged into the config's dataset_args, or replacing "
                    "them when --dataset-type is given")
    ap.add_argument("--segmentation-model-type", type=str, default=None)
    ap.add_argument("--prior-model-args", type=str, default=None, help="JSON dict merged into the config's prior_model_args")
    ap.add_argument("--override", type=str, default=None, help="JSON dict deep-merged into the whole config (e.g. agent_args); "
                    "may hold __class__-tagged objects")
    ap.add_argument("--save-masks", action="store_true", help="export every fitted prior's mask as <output>/masks/<index>.png "
                    "(1 bit per pixel; thresholded and bit-packed on the device, awesome_amd.run.evaluate_dataset)")
    args = ap.parse_args(argv)
    # the reference's file format: {AwesomeConfig: {__class__: ..., field: value | {__class__: dotted.Type, **fields}}}
    # (awesome/serialization/json_convertible.py:632-727); nested objects are decoded into the mirrors, nothing is dropped
    cfg = AwesomeConfig.load_from_file(args.config_path)
    for k in ("num_epochs", "seed", "device", "output_folder", "name_experiment", "dataset_type", "segmentation_model_type"):
        v = getattr(args, k)
        if v is not None:
            cfg[k] = v
    if args.num_epochs is not None:   # this entrypoint's shorthand: the flag also sets the per-image fit's epochs
        cfg.agent_args = dict(cfg.agent_args or {})
        cfg.agent_args["pretrain_args"] = dict(cfg.agent_args.get("pretrain_args") or {}, num_epochs=args.num_epochs)
    if args.dataset_args is not None:
        extra = S.decode(json.loads(args.dataset_args), in_scope=False, path="dataset_args")
        cfg["dataset_args"] = extra if args.dataset_type else dict(cfg.get("dataset_args") or {}, **extra)
    if args.prior_model_args is not None:
        cfg["prior_model_args"] = dict(cfg.get("prior_model_args") or {}, **S.decode(json.loads(args.prior_model_args), path="prior_model_args"))

    def deep_merge(dst, src):
        for k, v in src.items():
            if isinstance(v, dict) and isinstance(dst.get(k), dict):
                deep_merge(dst[k], v)
            else:
                dst[k] = v
    if args.override:
        over = {k: S.decode(v, in_scope=k in S.IN_SCOPE_KEYS, path=k) for k, v in json.loads(args.override).items()}
        merged = cfg.to_dict()
        deep_merge(merged, over)
        for k in over:
            cfg[k] = merged[k]
    if args.save_masks:
        cfg["save_masks"] = True
    return cfg


def _fusable(criterion):
    """The pretrain loops take SE / BCELoss / UnariesWeightedLoss of those as their `criterion` kwarg (fused in the kernels);
    composite training losses (FBMSJointLoss, AwesomeImageLoss) belong to the joint step, not to the per-image fit."""
    from awesome_amd.measures import criterion_to_desc
    try:
        criterion_to_desc(criterion, "targets")
        return True
    except TypeError:
        return False


def main(cfg):
    import torch
    import awesome_amd as A
    from awesome_amd import parallel
    from awesome_amd.agent import PretrainAgent
    from awesome_amd.dataset import SyntheticPriorDataset
    from awesome_amd.model import ForwardModule, WrapperModule

    rank, world, local = parallel.init()
    device = torch.device(cfg.get("device", "cuda"))
    if device.type != "cuda" or not torch.cuda.is_available():
        raise SystemExit("this entrypoint drives the MI355X path; no CPU fallback exists (use the reference for CPU runs)")
    if device.index is None:
        device = torch.device("cuda", local % torch.cuda.device_count())   # (% only matters for a gloo rehearsal on one GPU)
    torch.cuda.set_device(device)
    seed = int(cfg.get("seed", 42))
    torch.manual_seed(seed)

    model_type = cfg.prior_model_factory()
    model_args = dict(cfg.prior_model_args or {})
    ds_name = cfg.dataset_type if "dataset_type" in cfg.explicit else "awesome_amd.dataset.SyntheticUnariesDataset"
    if isinstance(ds_name, str) and ds_name.startswith("awesome.dataset."):
        raise SystemExit(f"dataset_type {ds_name} reads files that are not part of this build (SURVEY.md section 8: data loaders are "
                         "out of scope); pass --dataset-type awesome_amd.dataset.SyntheticUnariesDataset --dataset-args '{...}'")
    dataset_type = dynamic_import(ds_name) if isinstance(ds_name, str) else ds_name
    dataset_args = dict(cfg.dataset_args or {})
    # the per-image fit takes ITS arguments from agent_args.pretrain_args (criterion, lr, num_epochs, reuse_state, zoo ... -
    # torch_agent.py:561-627 hands them to PretrainableModule.pretrain); optimizer_type / optimizer_args / loss_type / num_epochs of
    # the config belong to the joint training that follows (awesome_runner.py:246-283)
    pre = cfg.pretrain_args()
    num_epochs = int(pre.get("num_epochs", 2000))
    if "num_epochs" in cfg.explicit and "num_epochs" not in pre:   # this repo's own YAMLs / --num-epochs without a pretrain block
        num_epochs = int(cfg.num_epochs)
    criterion = cfg.build_loss() if "loss_type" in cfg.explicit else None
    opt_type = cfg.optimizer_name()
    opt_args = dict(cfg.optimizer_args or {})
    out_dir = os.path.join(cfg.get("output_folder") or "runs", cfg.get("name_experiment") or "inr_fit")
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    parallel.barrier()

    if hasattr(dataset_type, "coords"):
        return _run_sequence(cfg, A, parallel, rank, world, device, model_type, model_args, dataset_type(**dataset_args), pre,
                             num_epochs, criterion, opt_type, opt_args, out_dir)

    # ---- per-image priors through the reference's interface: dataset with a PriorCache -> WrapperModule -> agent._pretrain ----
    ds = SyntheticPriorDataset(prior_model_type=model_type, prior_model_args=model_args, **dataset_args)
    ds.__prior_cache__.key_seed = seed          # image k starts from the same parameters on 1 rank and on 8
    mine = list(parallel.shard_range(len(ds), rank, world))
    wrapper = WrapperModule(ForwardModule(), model_type(**model_args), use_segmentation_output_inversion=True).to(device)
    kw = dict(pre)
    kw["num_epochs"] = num_epochs
    if getattr(ds, "independent_images", True):
        kw.setdefault("reuse_state", False)     # the synthetic images are unrelated fits (FBMS sequence configs chain the frames)
    if criterion is not None and _fusable(criterion):
        kw.setdefault("criterion", criterion)   # extension: a fusable loss_type doubles as the fit's criterion when pretrain_args has none
    if kw.get("zoo") and not hasattr(kw["zoo"], "load_model_state"):
        from awesome_amd.model import Zoo
        kw["zoo"] = Zoo(None if kw["zoo"] == "memory" else str(kw["zoo"]))
    aa = dict(cfg.agent_args or {})          # TorchAgent's pretraining switches (torch_agent.py:553-627)
    state_path = aa.get("pretrain_state_path")
    if state_path is not None and world > 1:
        state_path = f"{state_path}.rank{rank}"     # every rank fits (and caches) its own shard
    agent = PretrainAgent(ds, device=device, agent_folder=os.path.join(out_dir, f"rank{rank}"), pretrain_args=kw,
                          do_pretraining=aa.get("do_pretraining", True), force_pretrain=aa.get("force_pretrain", False),
                          pretrain_state_path=state_path)
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # joint training after the fits: `agent_args.joint_epochs` (this entrypoint's switch), or - a reference YAML - `num_epochs` epochs
    # unless `agent_args.pretrain_only` is set (awesome_runner.py:318-340 trains `num_epochs` after the agent's pretraining)
    agent_args = dict(cfg.agent_args or {})
    joint_epochs = int(agent_args.get("joint_epochs", 0 if agent_args.get("pretrain_only", True) else cfg.num_epochs))
    report, joint_losses, error = [], [], None
    try:                                        # rank-local work: an exception here must not strand the other ranks (below)
        report, joint_losses = _fit_shard(cfg, ds, mine, agent, wrapper, criterion, model_type, model_args, opt_args, device, joint_epochs)
    except Exception as err:   # noqa: BLE001 - reported, agreed on by all ranks, and turned into a non-zero exit
        import traceback
        traceback.print_exc()
        error = err
    # One flag all-reduce BEFORE the data collectives: if any rank failed (e.g. ValueError("Loss is nan or inf!") from pretrain),
    # every rank reports and exits non-zero together instead of hanging in max_over_ranks / gather_per_image / barrier.
    if parallel.any_rank_failed(error is not None, device):
        print(f"[run.py] rank {rank}: " + (f"failed: {error!r}" if error is not None else "another rank failed; exiting"),
              file=sys.stderr, flush=True)
        parallel.shutdown()
        raise SystemExit(1)
    torch.cuda.synchronize()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, device)

    cache = ds.__prior_cache__
    iou = retries = gt_iou = noisy_iou = None
    masks_saved, error = 0, None
    try:                                        # second rank-local phase (scores, mask export): same agreement as above
        f32 = lambda xs: torch.tensor(xs, dtype=torch.float32, device=device).reshape(len(mine))  # noqa: E731
        iou = f32([0.0 if r["skipped"] else r["iou"] for r in report])
        retries = f32([r["retries"] for r in report])
        gt_iou = noisy_iou = None
        if hasattr(ds, "ground_truth_batch") and dataset_args.get("kind") == "noisy_blob":
            # refinement configs: score the fitted prior (and the input labels) against the clean mask
            prior, size = wrapper.prior_module, ds.size
            vals, nvals = [], []
            for k in mine:
                prior.load_state_dict({n: v.to(device) for n, v in cache[k].items()})
                (_, _), ((_, _, xy), _) = ds[k]
                with torch.no_grad():
                    p = torch.sigmoid(prior(xy[None].to(device))).reshape(1, -1)
                gt = ds.ground_truth(k).reshape(1, -1).to(device)
                vals.append(float(A.miou(p, gt)[0]))
                nvals.append(float(A.miou(ds.unaries(k).reshape(1, -1).to(device), gt)[0]))
            gt_iou, noisy_iou = f32(vals), f32(nvals)
        if cfg.get("save_masks") and mine:
            # evaluation + export on the device (reference: get_result / split_model_result / save_result_mask per image on the host,
            # run/functions.py:2111-2151, 2315-2361, 2432-2487): every rank writes the masks of its own images into the shared folder
            from awesome_amd.run import evaluate_dataset
            masks_saved = len(evaluate_dataset(wrapper, ds, indices=mine, out_dir=os.path.join(out_dir, "masks"))["indices"])
    except Exception as err:   # noqa: BLE001
        import traceback
        traceback.print_exc()
        error = err
    if parallel.any_rank_failed(error is not None, device):
        print(f"[run.py] rank {rank}: " + (f"failed: {error!r}" if error is not None else "another rank failed; exiting"),
              file=sys.stderr, flush=True)
        parallel.shutdown()
        raise SystemExit(1)
    iou_all = parallel.gather_per_image(iou, len(ds), rank, world)
    retries_all = parallel.gather_per_image(retries, len(ds), rank, world)
    if gt_iou is not None:
        gt_all = parallel.gather_per_image(gt_iou, len(ds), rank, world)
        noisy_all = parallel.gather_per_image(noisy_iou, len(ds), rank, world)

    # every rank's priors reach the saved cache: one shard file per rank, merged by rank 0 (single node, shared folder)
    shard = os.path.join(out_dir, f"prior_cache_rank{rank}.pth")
    cache.__cache__ = {k: v for k, v in cache.__cache__.items() if k in set(mine)}   # only what this rank fitted
    cache.save(shard + ".tmp")
    os.replace(shard + ".tmp", shard)
    parallel.barrier()
    if rank == 0:
        merged = None
        for r in range(world):
            path = os.path.join(out_dir, f"prior_cache_rank{r}.pth")
            st = torch.load(path, map_location="cpu", weights_only=False)
            if merged is None:
                merged = st
            else:
                merged["cache"].update(st["cache"])
            os.remove(path)
        merged["cache"] = dict(sorted(merged["cache"].items(), key=lambda kv: int(kv[0])))
        torch.save(merged, os.path.join(out_dir, "prior_cache_epoch_0.pth"))
        torch.save(merged, os.path.join(out_dir, "pretrain_state.pth"))
        summary = {"images": len(ds), "ranks": world, "epochs": num_epochs, "seconds": round(dt, 4),
                   "fits_per_s": round(len(ds) / dt, 4), "ForegroundBinaryMIOU_vs_unaries": round(float(iou_all.mean()), 5),
                   "retries": [int(v) for v in retries_all.tolist()], "priors_saved": len(merged["cache"]), "output": out_dir}
        if cfg.get("save_masks"):
            summary["masks_saved"] = len([f for f in os.listdir(os.path.join(out_dir, "masks")) if f.endswith(".png")])
        if gt_iou is not None:
            summary["ForegroundBinaryMIOU_vs_ground_truth"] = round(float(gt_all.mean()), 5)
            summary["input_labels_MIOU_vs_ground_truth"] = round(float(noisy_all.mean()), 5)
        if joint_losses:
            summary["joint_epochs"] = joint_epochs
            summary["joint_loss_first_last"] = [round(joint_losses[0], 6), round(joint_losses[-1], 6)]
            if cfg.get("use_extra_penalty_hook"):
                summary["extra_penalty"] = bool(getattr(criterion, "extra_penalty", False))
        with open(os.path.join(out_dir, "summary.json"), "w") as f:
            json.dump(summary, f, indent=1)
        print(json.dumps(summary))
    parallel.barrier()


def _fit_shard(cfg, ds, mine, agent, wrapper, criterion, model_type, model_args, opt_args, device, joint_epochs):
    """This rank's share of the work: the per-image pretrain fits, then the joint-training epochs.  -> (report, joint_losses)"""
    import torch
    from awesome_amd.model import WrapperModule
    report = []
    if mine:                                    # a rank without images still joins every collective below
        agent._pretrain(wrapper, torch.utils.data.Subset(ds, mine), None, use_progress_bar=False)
        report = wrapper.prior_module.pretrain_report
    # ---- joint training epochs (TorchAgent._perform_step, torch_agent.py:428-551): segmentation module + per-image priors +
    # the composite loss (FBMSJointLoss / AwesomeImageLoss), the priors resident on the device in a PriorBank.  Each rank trains
    # its own copy of the segmentation stand-in on its shard (sharing a backbone over ranks is ordinary DDP, out of scope §8e).
    joint_losses = []
    if joint_epochs > 0 and mine and criterion is not None and not _fusable(criterion):
        from awesome_amd.agent import JointTrainer
        from awesome_amd.prior_bank import PriorBank
        seg_name = cfg.segmentation_model_type if "segmentation_model_type" in cfg.explicit else "awesome_amd.model.ConvSegStandIn"
        if isinstance(seg_name, str) and seg_name.startswith("awesome.model.") and seg_name not in ALIASES:
            raise SystemExit(f"segmentation_model_type {seg_name} is outside this build (SURVEY.md section 8: backbones run on torch as "
                             "they are); pass --segmentation-model-type with an importable torch module type")
        seg_type = dynamic_import(seg_name) if isinstance(seg_name, str) else seg_name
        seg = seg_type(**dict(cfg.get("segmentation_model_args") or {})).to(device)
        prior = wrapper.prior_module
        jw = WrapperModule(seg, prior, use_segmentation_output_inversion=True).to(device)
        bank = PriorBank(lambda: model_type(**model_args).to(device), n_images=len(mine), device=device, keys=mine)
        cache0 = ds.__prior_cache__
        for k in mine:
            prior.load_state_dict({n: v.to(device) for n, v in cache0[k].items()})
            bank.row(k).copy_(torch.cat([p.detach().reshape(-1) for p in bank_params(prior)]))
        params = [p for p in seg.parameters()] + bank_params(prior)
        opt_cls = cfg.optimizer_type if not isinstance(cfg.optimizer_type, str) else dynamic_import(cfg.optimizer_type)
        opt = opt_cls(params, **opt_args)       # awesome_runner.py:246-252: optimizer_type(**optimizer_args)
        trainer = JointTrainer(jw, bank, criterion, opt)
        for epoch in range(joint_epochs):
            # the runner's extra-penalty hook (awesome/run/awesome_runner.py:351-371; config fields awesome_config.py:164-173): from
            # epoch N on the loss adds its penalty term, optionally with the learning rate scaled once
            if (cfg.get("use_extra_penalty_hook") and epoch >= int(cfg.get("extra_penalty_after_n_epochs", 200))
                    and hasattr(criterion, "extra_penalty") and not criterion.extra_penalty):
                criterion.extra_penalty = True
                if cfg.get("use_reduce_lr_in_extra_penalty_hook"):
                    for group in opt.param_groups:
                        group["lr"] = group["lr"] * float(cfg.get("reduce_lr_in_extra_penalty_hook_factor", 0.05))
            acc = torch.zeros((), device=device)
            for k in mine:
                (_, _), ((image, feat, xy), target) = ds[k]
                loss, _ = trainer.perform_step(k, (image[None].to(device), feat[None].to(device), xy[None].to(device)),
                                               target[None].to(device))
                acc = acc + loss
            trainer.raise_if_failed()          # ValueError("Loss is nan or inf!") like the reference, one sync per epoch
            joint_losses.append(float(acc) / len(mine))
        for k in mine:   # the jointly trained priors replace the pretrained ones in the cache
            with bank.manager(prior, k):
                cache0[k] = {n: v.detach().cpu().clone() for n, v in prior.state_dict().items()}
    return report, joint_losses


def bank_params(prior):
    """The prior module's Parameter objects in the order of its flat vector (what PriorBank rows hold)."""
    op = prior._ordered_params()
    return (list(op[2]) + list(op[3])) if isinstance(op, tuple) and len(op) == 4 else list(op)


def _run_sequence(cfg, A, parallel, rank, world, device, model_type, model_args, dataset, pre, num_epochs, criterion, opt_type,
                  opt_args, out_dir):
    """(x, y, t) sequences: ONE network over all frames of a sequence (the spatio-temporal mode, path_connected_net.py:511-728);
    whole sequences are the unit that is sharded over the ranks."""
    import torch
    from awesome_amd.measures import criterion_targets, criterion_to_desc
    mine = list(parallel.shard_range(len(dataset), rank, world))
    grid = A.Grid.explicit(dataset.coords().to(device))
    lr = float(pre.get("lr", 1e-3))
    opt_type = str(pre.get("optimizer", "adamax")).lower()
    kind, wmode, _ = criterion_to_desc(criterion, "targets") if criterion is not None else ("se", "none", 1.0)
    kw = dict(num_epochs=num_epochs, lr=lr, loss=kind, weight_mode=wmode, flow_weight_decay=float(pre.get("flow_weight_decay", 1e-5)),
              optimizer=opt_type)
    kw.update({k: v for k, v in pre.items() if k.startswith("prefit_")})
    if pre.get("zoo"):
        from awesome_amd.model import Zoo
        kw["zoo"] = Zoo(None if pre["zoo"] == "memory" else str(pre["zoo"]))
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cache, ious, error = {}, [], None
    try:
        for i in mine:
            torch.manual_seed(int(cfg.get("seed", 42)) + i)
            model = model_type(**model_args).to(device)
            un = criterion_targets(criterion, dataset.batch([i]).to(device))
            res = model.fit_images(grid, un, **kw)
            if int(res.status[0]) != 0:
                raise ValueError(f"Loss is nan or inf! (sequence {i})")
            ious.append(float(A.miou(torch.sigmoid(res.logits), un)[0]))
            model.load_flat(res.icnn_params[0], res.flow_params[0])
            cache[str(i)] = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    except Exception as err:   # noqa: BLE001 - agreed on by all ranks before the collectives (see main)
        import traceback
        traceback.print_exc()
        error = err
    if parallel.any_rank_failed(error is not None, device):
        print(f"[run.py] rank {rank}: " + (f"failed: {error!r}" if error is not None else "another rank failed; exiting"),
              file=sys.stderr, flush=True)
        parallel.shutdown()
        raise SystemExit(1)
    torch.cuda.synchronize()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, device)
    iou_all = parallel.gather_per_image(torch.tensor(ious, dtype=torch.float32, device=device).reshape(len(mine)), len(dataset), rank, world)
    shard = os.path.join(out_dir, f"prior_cache_rank{rank}.pth")
    torch.save({"model_type": cfg.get("prior_model_type"), "model_args": json.dumps(model_args, indent=4), "store_device": "cpu",
                "cache": cache}, shard)
    parallel.barrier()
    if rank == 0:
        merged = None
        for r in range(world):
            path = os.path.join(out_dir, f"prior_cache_rank{r}.pth")
            st = torch.load(path, map_location="cpu", weights_only=False)
            if merged is None:
                merged = st
            else:
                merged["cache"].update(st["cache"])
            os.remove(path)
        torch.save(merged, os.path.join(out_dir, "prior_cache_epoch_0.pth"))
        summary = {"images": len(dataset), "ranks": world, "epochs": num_epochs, "seconds": round(dt, 4),
                   "fits_per_s": round(len(dataset) / dt, 4), "ForegroundBinaryMIOU_vs_unaries": round(float(iou_all.mean()), 5),
                   "retries": [0] * len(dataset), "priors_saved": len(merged["cache"]), "output": out_dir}
        with open(os.path.join(out_dir, "summary.json"), "w") as f:
            json.dump(summary, f, indent=1)
        print(json.dumps(summary))
    parallel.barrier()


if __name__ == "__main__":
    main(get_config())
