"""scripts/run.py reads the reference's YAML as written.

`tests/golden/config_trees.json` holds what the REFERENCE's own loader (JsonConvertible.load_from_file -> ObjectDecoder ->
object_hook, executed in the build container by tools/gen_golden_config_trees.py) makes of every YAML under its config/ folder for
the keys on the hot path: type names + public attribute values, no YAML text.  Here the same files go through this build's decoder
(awesome_amd.serialization / awesome_amd.run.config.AwesomeConfig) and the instantiated trees must describe themselves identically.
The reference checkout exists in the build container only: without it the comparison is skipped (the decoder's own unit tests
below always run)."""
import json
import os

import pytest
import torch

from awesome_amd import serialization as S
from awesome_amd.run.config import AwesomeConfig

REF_CONFIG = "/root/reference/config"
KEYS = ("prior_model_type", "prior_model_args", "loss_type", "loss_args", "optimizer_type", "optimizer_args", "num_epochs", "seed",
        "scribble_percentage", "use_extra_penalty_hook", "extra_penalty_after_n_epochs", "use_reduce_lr_in_extra_penalty_hook",
        "reduce_lr_in_extra_penalty_hook_factor", "segmentation_training_mode", "use_segmentation_output_inversion",
        "weight_decay_on_weight_norm_modules", "dtype", "device", "use_prior_model")      # = tools/gen_golden_config_trees.py KEYS
# in-scope classes this build does not implement: the decoder must REFUSE the file, never drop the object (round 4: none is left -
# GradientPenaltyLoss of the CNNNet convexity configs is mirrored as a torch composition)
UNBUILT = set()


def _classes(tree, acc=None):
    acc = set() if acc is None else acc
    if isinstance(tree, dict):
        for k in ("<class>", "<type>"):
            if k in tree:
                acc.add(tree[k])
        for v in tree.values():
            _classes(v, acc)
    elif isinstance(tree, list):
        for v in tree:
            _classes(v, acc)
    return acc


def _without_disk_state(tree):
    """The reference's Zoo lists the files of its folder in `files` when constructed: state of the disk, not of the config."""
    if isinstance(tree, dict):
        return {k: _without_disk_state(v) for k, v in tree.items() if not (tree.get("<class>") == "Zoo" and k == "files")}
    return [_without_disk_state(v) for v in tree] if isinstance(tree, list) else tree


def _describe_config(cfg: AwesomeConfig):
    """Same keys as the generator.  A key the file does not set is None in the fixture (the reference's root object could not be
    instantiated in the build container, so its defaults were not filled in) and the reference's default here."""
    d = {k: (S.describe(cfg.get(k)) if k in cfg.explicit else None) for k in KEYS}
    agent_args = cfg.agent_args or {}
    d["pretrain_args"] = S.describe(agent_args.get("pretrain_args"))
    d["agent_switches"] = {k: v for k, v in S.describe(agent_args).items() if k != "pretrain_args"}
    return d


@pytest.fixture(scope="module")
def trees(golden_dir):
    with open(os.path.join(golden_dir, "config_trees.json")) as f:
        return json.load(f)


@pytest.mark.skipif(not os.path.isdir(REF_CONFIG), reason="reference checkout (build container only)")
def test_every_reference_yaml_decodes_to_the_reference_loaders_trees(trees):
    assert trees["n_files"] == len(trees["files"]) >= 200
    refused, compared = [], 0
    for rel, h in sorted(trees["files"].items()):
        want = _without_disk_state(trees["trees"][h])
        path = os.path.join(REF_CONFIG, rel)
        unbuilt = _classes({k: want[k] for k in ("loss_args", "prior_model_args", "pretrain_args")}) & UNBUILT
        if unbuilt:
            with pytest.raises(S.UnmappedClassError) as err:
                AwesomeConfig.load_from_file(path)
            assert err.value.class_name.rsplit(".", 1)[-1] in unbuilt, rel
            refused.append(rel)
            continue
        cfg = AwesomeConfig.load_from_file(path)
        got = _describe_config(cfg)
        assert got == want, (rel, {k: (got[k], want[k]) for k in want if got.get(k) != want[k]})
        # ... and the objects the runner builds from them exist (awesome_runner.py:218-236, 256-264)
        loss = cfg.build_loss()
        assert type(loss).__name__ == want["loss_type"].rsplit(".", 1)[-1], rel
        assert type(loss.criterion).__name__ == want["loss_args"]["criterion"]["<class>"], rel   # (a WeightedLoss sets its criterion's reduction)
        assert callable(cfg.prior_model_factory()), rel
        compared += 1
    assert compared + len(refused) == trees["n_files"]
    assert len(refused) == 0 and compared == 207, (compared, len(refused))


def test_fixture_names_only_types_the_decoder_maps_or_refuses(trees):
    """Runs everywhere: every class / type name the reference's loader produced for an in-scope key is either in the mirror table
    or in the (stated) unbuilt set."""
    mirrors = {v.rsplit(".", 1)[-1] for v in S.ALIASES.values()} | {"BCELoss"}
    seen = set()
    for t in trees["trees"].values():
        seen |= _classes({k: t[k] for k in ("loss_args", "prior_model_args", "pretrain_args")})
    assert seen - mirrors == UNBUILT, seen - mirrors


def _tagged_cdn_config():
    """The in-scope part of a reference ConvexDiffeomorphismNet config, assembled from objects and ENCODED by this build (the tags are
    the reference's type names)."""
    from awesome_amd import measures as M
    cfg = AwesomeConfig(
        name_experiment="cdn", seed=42, num_epochs=200,
        prior_model_type="awesome.model.convex_diffeomorphism_net.ConvexDiffeomorphismNet",
        prior_model_args=dict(n_hidden=130, n_hidden_layers=2, diffeo_args=dict(backbone="normal_block", num_coupling=6, width=130)),
        loss_type="awesome.measures.fbms_joint_loss.FBMSJointLoss",
        loss_args=dict(alpha=1, beta=1, criterion=M.WeightedLoss(torch.nn.BCELoss(), mode="sssdms", noneclass=2)),
        optimizer_type="torch.optim.adam.Adam", optimizer_args=dict(lr=0.01, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False),
        agent_args=dict(pretrain_only=True, pretrain_args=dict(criterion=M.UnariesConversionLoss(M.SE("mean")), lr=0.001, num_epochs=2000,
                                                               proper_prior_fit_retrys=1)),
        dtype=torch.float32)
    return cfg


def test_encode_writes_the_reference_tags_and_decode_restores_the_objects(tmp_path):
    cfg = _tagged_cdn_config()
    tree = cfg.to_tagged_dict()["AwesomeConfig"]
    assert tree["__class__"] == "awesome.run.awesome_config.AwesomeConfig"
    crit = tree["agent_args"]["pretrain_args"]["criterion"]
    assert crit["__class__"] == "awesome.measures.unaries_conversion_loss.UnariesConversionLoss"
    assert crit["criterion"] == {"__class__": "awesome.measures.se.SE", "name": None, "reduction": "mean", "reduction_dim": None}
    assert tree["loss_args"]["criterion"]["criterion"]["__class__"] == "torch.nn.modules.loss.BCELoss"
    assert tree["optimizer_args"]["betas"] == {"__class__": "awesome.serialization.rules.json_tuple_serialization_rule.TupleValueWrapper",
                                               "value": [0.9, 0.999]}
    assert tree["dtype"]["value"] == "torch.float32"
    path = cfg.save_to_file(str(tmp_path / "cdn.yaml"))
    back = AwesomeConfig.load_from_file(path)
    assert S.describe(back.to_dict()) == S.describe(cfg.to_dict())
    loss = back.build_loss()
    assert type(loss.criterion).__name__ == "WeightedLoss" and loss.criterion.noneclass == 2 and loss.criterion.mode == "sssdms"
    assert isinstance(loss.criterion.criterion, torch.nn.BCELoss) and loss.criterion.criterion.reduction == "none"
    assert back.optimizer_args["betas"] == (0.9, 0.999) and back.torch_dtype() is torch.float32


def test_in_scope_tag_without_a_mirror_raises_and_out_of_scope_is_kept():
    bad = {"AwesomeConfig": {"loss_args": {"criterion": {"__class__": "awesome.measures.some_future_loss.SomeFutureLoss",
                                                         "criterion": {"__class__": "torch.nn.modules.loss.BCELoss", "reduction": "mean",
                                                                       "_modules": {}, "_buffers": {"weight": None}}}}}}
    with pytest.raises(S.UnmappedClassError, match="SomeFutureLoss"):
        S.decode_config(bad)
    ok = S.decode_config({"AwesomeConfig": {"dataset_args": {"dataset": {"__class__": "awesome.dataset.fbms_sequence_dataset.FBMSSequenceDataset",
                                                                         "dataset_path": "x", "dtype": {"__class__": "awesome.serialization.rules.torch."
                                                                                                        "json_torch_dtype_serialization_rule.TorchDtypeValueWrapper",
                                                                                                        "value": "torch.float32"}}}}})
    ds = ok["dataset_args"]["dataset"]
    assert isinstance(ds, S.OpaqueObject) and ds.fields["dataset_path"] == "x" and ds.fields["dtype"] is torch.float32
    # a serialised module with state the decoder cannot restore is refused, too
    with pytest.raises(ValueError, match="buffers"):
        S.decode({"__class__": "torch.nn.modules.loss.BCELoss", "_modules": {}, "_buffers": {"weight": [1.0]}, "reduction": "mean"})
    # the {type, args} nesting of this repo's first YAMLs stays an alias of the tagged form
    se = S.decode({"type": "awesome.measures.se.SE", "args": {"reduction": "sum"}})
    assert type(se).__name__ == "SE" and se.reduction == "sum"


def test_weighted_loss_with_noneclass_matches_its_definition():
    """weighted_loss.py:67-92: noneclass pixels leave before the criterion; sssdms weight round((bg / fg) / 10) + 1 on the fg pixels."""
    from awesome_amd.measures import WeightedLoss
    g = torch.Generator().manual_seed(0)
    out = torch.rand(2, 1, 8, 9, generator=g).clamp(0.05, 0.95)
    tgt = torch.ones(2, 1, 8, 9)
    tgt[:, :, :1, :2] = 0.0
    tgt[:, :, 4:, 5:] = 2.0
    loss = WeightedLoss(torch.nn.BCELoss(), mode="sssdms", noneclass=2)(out, tgt)
    keep = tgt != 2
    o, t = out[keep], tgt[keep]
    n_fg, n_bg = float((t == 0).sum()), float((t == 1).sum())
    w = torch.where(t == 0, torch.tensor(round((n_bg / n_fg) / 10) + 1.0), torch.tensor(1.0))
    want = (torch.nn.functional.binary_cross_entropy(o, t, reduction="none") * w).mean()
    torch.testing.assert_close(loss, want)


def test_gradient_penalty_loss_mirror_matches_its_definition():
    """gradient_penalty_loss.py:44-112 on a small differentiable 'segmentation network': criterion on the labeled pixels + xygrad *
    mean|d sum(out) / d xy| + featgrad * mean|d sum(out) / d feat| + rgbgrad * mean|d sum(out) / d image|, differentiable through."""
    from awesome_amd.measures import GradientPenaltyLoss
    torch.manual_seed(0)
    img = torch.rand(1, 3, 6, 7, requires_grad=True)
    xyf = torch.rand(1, 4, 6, 7, requires_grad=True)          # (x, y, 2 semantic features)
    conv = torch.nn.Conv2d(7, 1, 3, padding=1)
    out = torch.sigmoid(conv(torch.cat([img, xyf], 1)))
    tgt = (torch.rand(1, 1, 6, 7) > 0.5).float()
    tgt[0, 0, :2] = 2.0
    crit = GradientPenaltyLoss(torch.nn.BCELoss(), apply_gradient_penalty=True, xygrad=0.01, rgbgrad=0.02, featgrad=0.03, xytype="featxy", noneclass=2.0)
    loss = crit(out, tgt, _input=(img, xyf))
    g_xy = torch.autograd.grad(out.sum(), xyf, retain_graph=True, create_graph=True)[0]
    g_im = torch.autograd.grad(out.sum(), img, retain_graph=True, create_graph=True)[0]
    keep = tgt != 2.0
    want = (torch.nn.functional.binary_cross_entropy(out[keep], tgt[keep]) + 0.01 * g_xy[:, :2].abs().mean() + 0.03 * g_xy[:, 2:].abs().mean()
            + 0.02 * g_im.abs().mean())
    torch.testing.assert_close(loss, want)
    loss.backward()                                            # second order through the network
    assert conv.weight.grad is not None and torch.isfinite(conv.weight.grad).all()
    crit.apply_gradient_penalty = False                        # what the joint losses do around the prior channel
    torch.testing.assert_close(crit(out.detach(), tgt), torch.nn.functional.binary_cross_entropy(out.detach()[keep], tgt[keep]))
