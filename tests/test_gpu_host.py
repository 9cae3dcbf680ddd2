"""GPU tests of the host layer: drop-in modules (autograd bridge), BatchedPriorFitter (gate/retry/warm start), run.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import inr_oracle as O  # noqa: E402  (checker only)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_module_forward_backward_like_reference_loop(dev, golden_dir):
    """The stock loop (zero_grad, forward, criterion, backward, Adam.step, enforce_convexity) driven through the drop-in
    module reproduces the reference trajectory (golden adam10 weights)."""
    from awesome_amd.model import ConvexNextNet
    from awesome_amd.measures import SE, UnariesWeightedLoss
    z = np.load(os.path.join(golden_dir, "icnn_convexnext_h130_c2_l1.npz"))
    model = ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=1)
    model.load_state_dict(O.load_npz_state(z, "sd0."))
    model.to(dev)
    grid = torch.from_numpy(z["grid"]).to(dev)
    un = torch.from_numpy(z["unaries"]).to(dev)
    crit = UnariesWeightedLoss(SE("mean"))
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    losses = []
    for _ in range(10):
        opt.zero_grad()
        out = torch.sigmoid(model(grid))
        assert out.shape == un.shape
        loss = crit(out, un)
        loss.backward()
        opt.step()
        model.enforce_convexity()
        losses.append(float(loss.detach()))
    np.testing.assert_allclose(np.asarray(losses, np.float32), z["adam.losses"], rtol=2e-5)
    sd = model.state_dict()
    for k in sd:
        np.testing.assert_allclose(sd[k].cpu().numpy(), z["adam10." + k], rtol=2e-4, atol=5e-6, err_msg=k)
    # (N, C) rows in, (N, 1) out
    rows = grid[0].reshape(2, -1).t().contiguous()
    assert model(rows).shape == (rows.shape[0], 1)


def test_fitter_gate_retry_and_cache_layout(dev):
    import awesome_amd as A
    from awesome_amd.dataset import SyntheticUnariesDataset
    from awesome_amd.fitter import BatchedPriorFitter
    from awesome_amd.model import ConvexNextNet
    torch.manual_seed(0)
    ds = SyntheticUnariesDataset(n_images=3, size=64, kind="blob")
    un = ds.batch([0, 1, 2]).to(dev)
    un[2] = 1.0  # an image without foreground is skipped like the reference does
    f = BatchedPriorFitter(lambda: ConvexNextNet(n_hidden=130), num_epochs=300, lr=2e-3, optimizer="adam", plateau=False)
    rep = f.fit_batch(A.Grid.linspace(64, 64, dev), un)
    assert rep.skipped == [False, False, True]
    assert float(rep.iou[0]) > 0.9 and float(rep.iou[1]) > 0.9
    state = f.prior_cache_state(rep, indices=[10, 11, 12], model_args={"n_hidden": 130})
    assert set(state) == {"model_type", "model_args", "store_device", "cache"} and set(state["cache"]) == {"10", "11"}
    assert list(state["cache"]["10"].keys()) == list(ConvexNextNet().state_dict().keys())
    # an impossible threshold forces the retry path: parameters are re-initialised and refitted once
    f2 = BatchedPriorFitter(lambda: ConvexNextNet(n_hidden=130), num_epochs=20, lr=2e-3, optimizer="adam", plateau=False,
                            proper_prior_fit_threshold=1.1, proper_prior_fit_retrys=1)
    rep2 = f2.fit_batch(A.Grid.linspace(64, 64, dev), un[:2].contiguous())
    assert rep2.retries == [1, 1]


def test_fitter_warm_start_chain(dev):
    import awesome_amd as A
    from awesome_amd.dataset import convex_blob_unaries
    from awesome_amd.fitter import BatchedPriorFitter
    from awesome_amd.model import ConvexNextNet
    torch.manual_seed(0)
    base = convex_blob_unaries(64, 1)
    frames = torch.stack([torch.roll(base, shifts=2 * k, dims=1).reshape(-1) for k in range(3)] * 2).to(dev)  # 2 sequences
    f = BatchedPriorFitter(lambda: ConvexNextNet(n_hidden=130), num_epochs=300, lr=2e-3, optimizer="adam", plateau=False,
                           reuse_state=True, reuse_state_epochs=60)
    rep = f.fit_sequences(A.Grid.linspace(64, 64, dev), frames, seq_ids=[0, 0, 0, 1, 1, 1])
    assert float(rep.iou.min()) > 0.85   # 60 warm epochs are enough when starting from the previous frame


def test_run_py_config_entrypoint(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run.py"), "--config-path",
                          os.path.join(ROOT, "config", "c1_disc64.yaml"), "--output-folder", str(tmp_path), "--save-masks"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    summary = json.loads(out.stdout.strip().splitlines()[-1])
    assert summary["images"] == 1 and summary["ForegroundBinaryMIOU_vs_unaries"] > 0.98
    cache = torch.load(os.path.join(summary["output"], "prior_cache_epoch_0.pth"))
    assert "0" in cache["cache"] and "skip.0.ln.weight" in cache["cache"]["0"]
    # --save-masks: the fitted prior's mask, thresholded and bit-packed on the device, as a 1-bit PNG (white = object)
    assert summary["masks_saved"] == 1
    PIL = pytest.importorskip("PIL.Image")
    im = np.asarray(PIL.open(os.path.join(summary["output"], "masks", "0.png"))).astype(bool)
    from awesome_amd.dataset import disc_unaries
    disc = disc_unaries(64, 64, 32, 32, 15.0).numpy() < 0.5
    assert im.shape == (64, 64) and (im & disc).sum() / (im | disc).sum() > 0.97


def test_run_py_two_ranks_save_every_prior(tmp_path):
    """scripts/run.py on 2 ranks (gloo for the few bytes of collectives, both ranks on this one GPU): every image's prior - not just
    rank 0's shard - must be in prior_cache_epoch_0.pth, with the same parameters as the 1-rank run (per-key seeding)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cfg = os.path.join(ROOT, "config", "c3_blobs512.yaml")
    common = ["--config-path", cfg, "--num-epochs", "40", "--dataset-args", json.dumps({"n_images": 5, "size": 64})]
    env = dict(os.environ, INRFIT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "scripts", "run.py"), *common,
                          "--output-folder", str(tmp_path / "two")], env=env, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    s2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    one = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run.py"), *common, "--output-folder", str(tmp_path / "one")],
                         env=env, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    s1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert s2["images"] == s1["images"] == 5 and s2["ranks"] == 2 and s2["priors_saved"] == s1["priors_saved"] == 5
    c2 = torch.load(os.path.join(s2["output"], "prior_cache_epoch_0.pth"), weights_only=False)
    c1 = torch.load(os.path.join(s1["output"], "prior_cache_epoch_0.pth"), weights_only=False)
    assert sorted(c2["cache"]) == sorted(c1["cache"]) == [str(i) for i in range(5)]
    assert not any(f.startswith("prior_cache_rank") for f in os.listdir(s2["output"]))     # the shards were merged away
    # same initial parameters per image whatever the sharding; the fits differ only by the slab count of the batch they ran in
    for i in range(5):
        a, b = c1["cache"][str(i)]["out.ln.weight"], c2["cache"][str(i)]["out.ln.weight"]
        assert torch.allclose(a, b, rtol=5e-3, atol=5e-4), i


def test_run_py_reads_a_reference_schema_yaml(tmp_path):
    """a config in the REFERENCE's file format - every nested object a `{__class__: dotted.Type, **fields}`
    mapping with the reference's own type names, the shape of config/path-connectedness/refit-unet-prior-only/*diffeo*.yaml:
    ConvexDiffeomorphismNet (6 couplings x 130, ICNN 130 x 2), pretrain criterion UnariesConversionLoss(SE), FBMSJointLoss(WeightedLoss(
    BCELoss, sssdms, noneclass 2)), Adam with a TupleValueWrapper for betas, an FBMSSequenceDataset under dataset_args, a UNet backbone -
    runs through scripts/run.py with ONLY what is not in the image replaced on the command line (dataset, backbone, paths, epochs)."""
    import yaml
    from awesome_amd import serialization as S
    from tests.test_config_trees import _tagged_cdn_config
    cfg = _tagged_cdn_config()
    cfg["dataset_type"] = "awesome.dataset.awesome_dataset.AwesomeDataset"
    cfg["dataset_args"] = dict(batch_size=1, dimension="3d", xytype="edge",
                               dataset=S.OpaqueObject("awesome.dataset.fbms_sequence_dataset.FBMSSequenceDataset",
                                                      dict(dataset_path="data/local_datasets/FBMS-59/train/bear01", all_frames=True, dtype=torch.float32)))
    cfg["segmentation_model_type"] = "awesome.model.unet.UNet"
    cfg["segmentation_model_args"] = dict(in_chn=4)
    cfg["use_segmentation_output_inversion"] = True
    cfg.agent_args["pretrain_args"].update(do_pretrain_checkpoints=True, use_pretrain_checkpoints=True, use_logger=True, use_step_logger=False,
                                           pretrain_checkpoint_dir="./data/checkpoints/pretrain_states/x")
    cfg.agent_args.update(do_pretraining=True, force_pretrain=True, pretrain_state_path="./data/checkpoints/pretrain_states/x.pth")
    path = cfg.save_to_file(str(tmp_path / "UNET+bear01+edge+diffeo+only_prior+REFIT.yaml"))
    tree = yaml.safe_load(open(path))["AwesomeConfig"]
    assert tree["agent_args"]["pretrain_args"]["criterion"]["__class__"] == "awesome.measures.unaries_conversion_loss.UnariesConversionLoss"
    assert tree["dataset_args"]["dataset"]["__class__"] == "awesome.dataset.fbms_sequence_dataset.FBMSSequenceDataset"
    run = [sys.executable, os.path.join(ROOT, "scripts", "run.py"), "--config-path", path, "--output-folder", str(tmp_path / "out")]
    # as written: the loaders of the reference are not part of this build - a clear refusal, not a silent default
    refused = subprocess.run(run, capture_output=True, text=True, timeout=600)
    assert refused.returncode != 0 and "dataset_type" in refused.stderr and "out of scope" in refused.stderr
    override = {"agent_args": {"pretrain_state_path": str(tmp_path / "state.pth"),
                               "pretrain_args": {"num_epochs": 2000, "pretrain_checkpoint_dir": str(tmp_path / "ckpt")}}}
    out = subprocess.run(run + ["--dataset-type", "awesome_amd.dataset.SyntheticUnariesDataset", "--dataset-args",
                                json.dumps(dict(n_images=1, size=64, kind="blob")), "--segmentation-model-type",
                                "awesome_amd.model.ConvSegStandIn", "--override", json.dumps(override)],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    summary = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert summary["images"] == 1 and summary["priors_saved"] == 1 and summary["epochs"] == 2000
    assert summary["ForegroundBinaryMIOU_vs_unaries"] > 0.9, summary
    assert "joint_epochs" not in summary                                   # agent_args.pretrain_only
    cache = torch.load(os.path.join(summary["output"], "prior_cache_epoch_0.pth"), weights_only=False)
    assert any(k.startswith("diffeo_net.") for k in cache["cache"]["0"]) and "convex_net.skip.1.ln.weight" in cache["cache"]["0"]
    assert os.path.exists(tmp_path / "state.pth")                          # TorchAgent's pretrain_state_path
    assert sorted(os.listdir(tmp_path / "ckpt")) == ["pretrain_checkpoint_0.pth"]


@pytest.mark.parametrize("config,override,checks", [
    ("c5_refine_noisy256.yaml", {"dataset_args": {"n_images": 2, "size": 64}, "agent_args": {"joint_epochs": 3, "pretrain_args": {"num_epochs": 80}}},
     dict(images=2, joint_epochs=3)),
    # the runner's extra-penalty hook in the joint epochs (awesome_runner.py:351-371): AwesomeImageLoss switches its penalty on at epoch 1
    ("c5_refine_noisy256.yaml", {"dataset_args": {"n_images": 2, "size": 64}, "agent_args": {"joint_epochs": 3, "pretrain_args": {"num_epochs": 80}},
                                 "loss_type": "awesome_amd.measures.AwesomeImageLoss", "loss_args": {"alpha": 1.0},
                                 "use_extra_penalty_hook": True, "extra_penalty_after_n_epochs": 1,
                                 "use_reduce_lr_in_extra_penalty_hook": True},
     dict(images=2, joint_epochs=3, extra_penalty=True)),
    ("c2_blob256_path_connected.yaml", {"dataset_args": {"size": 64}, "agent_args": {"pretrain_args": {"num_epochs": 80}}}, dict(images=1)),
    ("c4_sequence128x16.yaml", {"dataset_args": {"size": 32, "frames": 4}, "agent_args": {"pretrain_args": {"num_epochs": 60}}}, dict(images=1)),
    ("c1_disc64_siren.yaml", {}, dict(images=1)),
    ("c1_disc64_no_prior.yaml", {}, dict(images=1)),
    ("c2_blob256_wide256.yaml", {"dataset_args": {"size": 64}, "agent_args": {"pretrain_args": {"num_epochs": 300}}}, dict(images=1)),   # layer-by-layer path
])
def test_run_py_configs_of_the_flow_priors(tmp_path, config, override, checks):
    """The configs of BASELINE configs[4] (noisy pseudo-labels: per-image pre-fit, then joint epochs with FBMSJointLoss through
    WrapperModule + PriorBank), of the path-connected prior on one image and of the (x, y, t) sequence, at reduced size."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run.py"), "--config-path", os.path.join(ROOT, "config", config),
                          "--output-folder", str(tmp_path), "--override", json.dumps(override)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    summary = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    for k, v in checks.items():
        assert summary[k] == v, (k, summary)
    assert summary["priors_saved"] == summary["images"]
    cache = torch.load(os.path.join(summary["output"], "prior_cache_epoch_0.pth"), weights_only=False)
    assert all(torch.isfinite(v).all() for sd in cache["cache"].values() for v in sd.values() if v.is_floating_point())
    if "joint_epochs" in checks:
        first, last = summary["joint_loss_first_last"]
        assert np.isfinite(first) and np.isfinite(last)
        if not checks.get("extra_penalty"):      # (the hook changes the loss's definition between the first and the last epoch)
            assert last <= first * 1.05
        assert "ForegroundBinaryMIOU_vs_ground_truth" in summary


def test_wrapper_module_joint_step(dev):
    """WrapperModule(ForwardModule, ConvexNextNet) + AwesomeImageLoss: one joint training step like TorchAgent._perform_step
    (forward -> (B,2,H,W), criterion, backward, step, enforce_convexity) against the oracle."""
    from awesome_amd.measures import AwesomeImageLoss
    from awesome_amd.model import ConvexNextNet, ForwardModule, WrapperModule
    torch.manual_seed(4)
    prior = ConvexNextNet(n_hidden=130)
    sd = {k: v.clone() for k, v in prior.state_dict().items()}
    w = WrapperModule(ForwardModule(), prior).to(dev)
    H, W = 12, 9
    grid = O.positional_grid(W, H)[None]
    seg_logits = torch.randn(1, 1, H, W)
    tgt = (torch.rand(1, 1, H, W) > 0.5).float()
    out = w(seg_logits.to(dev), torch.zeros(1, 1, H, W, device=dev), grid.to(dev))
    assert out.shape == (1, 2, H, W)
    crit = AwesomeImageLoss(alpha=0.7)
    crit.extra_penalty = True
    loss = crit(out, tgt.to(dev))
    loss.backward()
    # oracle
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_out = torch.cat([torch.sigmoid(seg_logits), torch.sigmoid(O.icnn_forward_image(p, grid))], dim=1)
    ref_loss = O.awesome_image_loss(ref_out, tgt, alpha=0.7, extra_penalty=True)
    ref_loss.backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref_out.detach().numpy(), atol=2e-6)
    assert float(loss.detach()) == pytest.approx(float(ref_loss.detach()), rel=1e-5)
    for k, prm in prior.named_parameters():
        r = p[k].grad.numpy()
        np.testing.assert_allclose(prm.grad.cpu().numpy(), r, rtol=5e-4, atol=2e-6 * float(np.abs(r).max()) + 1e-9, err_msg=k)
    seg, pr = w.split_model_output(out)[0]
    assert seg.shape == (1, H, W) and pr.shape == (1, H, W)
    w.evaluate_prior = False
    assert w(seg_logits.to(dev), torch.zeros(1, 1, H, W, device=dev), grid.to(dev)).shape == (1, 1, H, W)


@pytest.mark.parametrize("name,width,depth", [("fcnet_w130_d1", 130, 1), ("fcnet_w64_d2", 64, 2)])
def test_fcnet_no_prior_network_on_hip(golden_dir, name, width, depth):
    """FCNet(in_type='xy') on the ICNN kernels (zero skips, frozen; no clamp): forward, autograd gradients and a 10-step Adam
    trajectory against the reference class' golden vectors."""
    import awesome_amd as A
    from awesome_amd.model import FCNet
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    m = FCNet(in_chn=2, out_chn=1, width=width, depth=depth, in_type="xy")
    m.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")})
    m = m.to(dev)
    rows, un = torch.from_numpy(z["rows"]).to(dev), torch.from_numpy(z["unaries"]).to(dev)
    y = m(None, rows)
    np.testing.assert_allclose(y.detach().cpu().numpy(), z["logits"], rtol=1e-4, atol=5e-6)
    loss = ((un - torch.sigmoid(y)) ** 2).mean()
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(z["loss"]), rel=1e-5)
    for k, p in m.named_parameters():
        ref = z["grad." + k]
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=2e-4, atol=2e-6 * float(np.abs(ref).max()) + 1e-9, err_msg=k)
    # the fused fit with the skips frozen at zero and no clamp = the reference's Adam loop
    res = A.fit(m.spec, m.flat_parameters()[None].to(dev), A.Grid.explicit(rows.t().contiguous()), un.reshape(1, -1), 10, lr=2e-3,
                loss="se", optimizer="adam", **FCNet.fit_options)
    np.testing.assert_allclose(res.loss_hist[0].cpu().numpy(), z["adam10.losses"], rtol=2e-4)
    got = m.unpack_flat(res.params[0].cpu())
    for k, v in got.items():
        np.testing.assert_allclose(v.numpy(), z["adam10." + k], rtol=2e-4, atol=2e-6, err_msg=k)
    flat = res.params[0].cpu()
    sd_i = A.unpack_params(m.spec, flat)
    assert all(float(sd_i[k].abs().max()) == 0.0 for k in sd_i if k.endswith("skp.weight"))   # never moved


def test_mfma_stream_hook():
    """inrfit_mfma_stream (bench.py's `mfma_only_stream_tflops`): a launch of nothing but fp32 MFMAs must land between a third of
    and just above the nominal 157.3 TFLOP/s - it is the yardstick next to roofline.peak, so a broken count would mislead."""
    import awesome_amd as A
    t = A.icnn.mfma_stream_tflops("cuda:0", workgroups=256, iters=3000)
    assert 50.0 < t < 165.0, t


def test_prior_bank_joint_step_and_prefit(golden_dir):
    """Joint-training step on a bank-bound prior model (forward/backward on the HIP path, torch optimizer on the views) equals the
    same step on a stand-alone copy bit for bit, and PriorBank.fit updates the resident rows exactly like awesome_amd.fit."""
    import copy
    import awesome_amd as A
    from awesome_amd.model import ConvexNextNet
    from awesome_amd.prior_bank import PriorBank
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    factory = lambda: ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=1).to(dev)
    bank = PriorBank(factory, n_images=4, device=dev)
    model = factory()
    x = torch.rand(1, 2, 24, 24, device=dev)
    target = (torch.rand(1, 1, 24, 24, device=dev) > 0.5).float()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    for key in (3, 1, 3):
        with bank.manager(model, key):
            twin = copy.deepcopy(model)                       # same values, own storage
            topt = torch.optim.Adam(twin.parameters(), lr=1e-2)
            topt.load_state_dict(copy.deepcopy(opt.state_dict()))
            for m, o in ((model, opt), (twin, topt)):
                o.zero_grad()
                loss = ((torch.sigmoid(m(x)) - target) ** 2).mean()
                loss.backward()
                o.step()
                m.enforce_convexity()
            assert torch.equal(bank.row(key), twin.flat_parameters())
    # per-image pre-fit straight on the rows
    S = 32
    grid = A.Grid.linspace(S, S, dev)
    un = torch.stack([(torch.rand(S * S, device=dev) > 0.5).float() for _ in range(2)])
    expect = A.fit(model.spec, bank.rows([1, 2]).contiguous(), grid, un, 50, lr=2e-3).params
    bank.fit([1, 2], grid, un, 50, lr=2e-3)
    assert torch.equal(bank.rows([1, 2]), expect)
