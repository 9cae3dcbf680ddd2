"""GPU tests of the reference's boundary on the HIP path: WrapperModule.forward and ConvexDiffeomorphismNet against fixtures
dumped from the real classes (tools/gen_golden_boundary.py), and the pretrain entry point
`wrapper.pretrain(train_set, test_set, device, agent, ...)` (awesome/model/pretrainable_module.py:15-83,
path_connected_net.py:472-509, 730-1019, convex_diffeomorphism_net.py:190-490, wrapper_module.py:325-340) on the fused fits."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import inr_oracle as O  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _z(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_wrapper_module_forward_matches_reference_class(dev, golden_dir):
    from awesome_amd.model import ConvexNextNet, ForwardModule, WrapperModule
    z = _z(golden_dir, "wrapper_module.npz")
    prior = ConvexNextNet(n_hidden=32, in_features=2, n_hidden_layers=1)
    prior.load_state_dict(O.load_npz_state(z, "prior."))
    img, xy = torch.from_numpy(z["img"]).to(dev), torch.from_numpy(z["xy"]).to(dev)
    feat = torch.zeros_like(img)
    for inv in (0, 1):
        wm = WrapperModule(ForwardModule(), prior, prior_arg_mode="param_clean_grid", input_mode="image",
                           use_segmentation_sigmoid=True, use_prior_sigmoid=True, use_segmentation_output_inversion=bool(inv)).to(dev)
        with torch.no_grad():
            out = wm(img, feat, xy)
            np.testing.assert_allclose(out.cpu().numpy(), z[f"out_inv{inv}"], atol=2e-6, rtol=1e-5)
            wm.evaluate_prior = False
            np.testing.assert_allclose(wm(img, feat, xy).cpu().numpy(), z[f"seg_only_inv{inv}"], atol=1e-6)
            wm.evaluate_prior = True
            seg, pr = wm.split_model_output(out)[0]
            np.testing.assert_allclose(seg.cpu().numpy(), z[f"split0_seg_inv{inv}"], atol=2e-6)
            np.testing.assert_allclose(pr.cpu().numpy(), z[f"split0_prior_inv{inv}"], atol=2e-6)
            pa, _ = wm.get_prior_args(img[0], feat[0], xy[0])
            np.testing.assert_array_equal(pa[0].cpu().numpy(), z[f"prior_arg_inv{inv}"])
        # the oracle's restatement of the composition
        ref = O.wrapper_forward(img.cpu(), O.icnn_forward_image(O.load_npz_state(z, "prior."), xy.cpu()), invert_seg=bool(inv))
        np.testing.assert_allclose(ref.numpy(), z[f"out_inv{inv}"], atol=2e-6, rtol=1e-5)


def test_wrapper_module_pixel_mode_matches_reference_class(dev, golden_dir):
    """input_mode='pixel' (the scribble-trained convexity configs: segmentation_training_mode 'single', AwesomeLoss): forward on
    (img, n_pixels, 5) pixels with the coordinates pre-attached, split, the prior's arguments, and AwesomeLoss (HIP loss kernel, form
    INR_JOINT_AWESOME_PIXEL) with its gradients w.r.t. the prior (HIP backward bridge) and the segmentation module (torch), against
    the reference WrapperModule / AwesomeLoss (fixture wrapper_module_pixel.npz; the fixture's header notes the reference's own
    enum-vs-string slip in combine_outputs and the documented layout used here)."""
    from awesome_amd.measures import AwesomeLoss
    from awesome_amd.model import ConvexNextNet, WrapperModule
    z = _z(golden_dir, "wrapper_module_pixel.npz")
    prior = ConvexNextNet(n_hidden=32, in_features=2, n_hidden_layers=1)
    prior.load_state_dict(O.load_npz_state(z, "prior."))
    seg = torch.nn.Sequential(torch.nn.Linear(5, 8), torch.nn.ReLU(), torch.nn.Linear(8, 1))
    seg.load_state_dict(O.load_npz_state(z, "seg."))
    wm = WrapperModule(seg, prior, prior_arg_mode="xy_c_preattached", input_mode="pixel").to(dev)
    x, tgt = torch.from_numpy(z["x"]).to(dev), torch.from_numpy(z["target"]).to(dev)
    with torch.no_grad():
        out = wm(x)
    assert out.shape == (2, 40, 2)
    np.testing.assert_allclose(out.cpu().numpy(), z["out"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(out.cpu().numpy()[..., 0], z["out_raw"][:, :40, 0], atol=2e-6, rtol=1e-5)   # the class's raw rows: seg ...
    np.testing.assert_allclose(out.cpu().numpy()[..., 1], z["out_raw"][:, 40:, 0], atol=2e-6, rtol=1e-5)   # ... then prior
    s1, p1 = wm.split_model_output(out)[1]
    np.testing.assert_allclose(s1.cpu().numpy(), z["out"][1, :, :1], atol=2e-6)
    np.testing.assert_allclose(p1.cpu().numpy(), z["out"][1, :, 1:], atol=2e-6)
    pa, _ = wm.get_prior_args(x[0])
    np.testing.assert_array_equal(pa[0].cpu().numpy(), z["prior_arg"])
    for penalty, tag in ((False, "plain"), (True, "pen")):
        crit = AwesomeLoss(alpha=0.6, scribble_percentage=0.75)
        crit.extra_penalty = penalty
        assert crit.joint_desc(40) is not None            # the HIP loss kernel is what runs on device tensors
        wm.zero_grad()
        loss = crit(wm(x), tgt)
        loss.backward()
        assert float(loss.detach()) == pytest.approx(float(z[f"loss_{tag}"]), rel=5e-6)
        for k, p_ in prior.named_parameters():
            ref = z[f"grad_{tag}.prior.{k}"]
            np.testing.assert_allclose(p_.grad.cpu().numpy(), ref, rtol=3e-4, atol=3e-6 * float(np.abs(ref).max()) + 1e-9, err_msg=f"{tag} prior {k}")
        for k, p_ in seg.named_parameters():
            ref = z[f"grad_{tag}.seg.{k}"]
            np.testing.assert_allclose(p_.grad.cpu().numpy(), ref, rtol=3e-4, atol=3e-6 * float(np.abs(ref).max()) + 1e-9, err_msg=f"{tag} seg {k}")


@pytest.mark.parametrize("tag", ["l2_w130_k6", "l1_w24_k4"])
def test_convex_diffeomorphism_net_class_fixture(dev, golden_dir, tag):
    """The class itself (both backbones: the configs' normal_block and the constructor's default SimpleBackbone): forward,
    deformation, BCE gradients of every parameter, and the centre-of-mass translate - HIP vs the reference class."""
    from awesome_amd.model import ConvexDiffeomorphismNet
    z = _z(golden_dir, f"cdn_class_{tag}.npz")
    m = ConvexDiffeomorphismNet(**json.loads(str(z["kwargs"])))
    sd = O.load_npz_state(z, "sd.")
    assert list(m.state_dict().keys()) == list(sd.keys())       # same names in the same order as the class
    m.load_state_dict(sd)
    m.to(dev)
    grid, un = torch.from_numpy(z["grid"]).to(dev), torch.from_numpy(z["unaries"]).to(dev)
    logits = m(grid)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), z["logits"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(m.get_deformation(grid).cpu().numpy(), z["deformation"], atol=2e-6, rtol=1e-5)
    loss = torch.nn.BCELoss()(torch.sigmoid(logits), un)
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(z["loss"]), rel=1e-5)
    for k, p in m.named_parameters():
        ref = z["grad." + k]
        np.testing.assert_allclose(p.grad.cpu().numpy().reshape(ref.shape), ref, rtol=1e-3, atol=1e-5 * float(np.abs(ref).max()) + 1e-7,
                                   err_msg=k)
    m.translate_only_point(torch.from_numpy(z["tr_src"]).to(dev), torch.from_numpy(z["tr_dst"]).to(dev), grid=grid.squeeze())
    with torch.no_grad():
        np.testing.assert_allclose(m(grid).cpu().numpy(), z["tr_logits_after"], atol=5e-5, rtol=1e-4)


def test_convex_diffeomorphism_net_with_the_resnet_backbone(dev, golden_dir):
    """The 'resnet' flow backbone (SimpleResnet: batch norms over the points - torch operations on the device) in front of the ICNN
    kernels: forward, deformation and the BCE gradients of every parameter (the flow's through the ICNN kernel's dL/dcoords) vs the
    reference class."""
    from awesome_amd.model import ConvexDiffeomorphismNet
    z = _z(golden_dir, "cdn_class_resnet.npz")
    m = ConvexDiffeomorphismNet(**json.loads(str(z["kwargs"])))
    m.load_state_dict(O.load_npz_state(z, "sd."))
    m.to(dev)
    grid, un = torch.from_numpy(z["grid"]).to(dev), torch.from_numpy(z["unaries"]).to(dev)
    logits = m(grid)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), z["logits"], atol=3e-5, rtol=1e-4)
    # (four couplings x two nets x five batch norms over 120 points: the device's and the CPU's fp32 reductions differ by ~1e-4)
    np.testing.assert_allclose(m.get_deformation(grid).cpu().numpy(), z["deformation"], atol=3e-4, rtol=1e-3)
    loss = torch.nn.BCELoss()(torch.sigmoid(logits), un)
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(z["loss"]), rel=1e-5)
    # gradients against the SCALE of the whole gradient (max |g| over all parameters): biases in front of a batch norm have an
    # analytically zero gradient (1e-8 of rounding on both sides), and the norm chain amplifies fp32 rounding to ~2e-3 of that scale
    gmax = max(float(np.abs(z["grad." + k]).max()) for k, _ in m.named_parameters())
    for k, p in m.named_parameters():
        ref = z["grad." + k]
        np.testing.assert_allclose(p.grad.cpu().numpy().reshape(ref.shape), ref, rtol=1e-2, atol=5e-3 * gmax, err_msg=k)


def test_normalizing_flow_module_on_hip(dev, golden_dir):
    """NormalizingFlow1D.forward + autograd on inrfit_flow_forward / inrfit_flow_backward vs the reference module's fixture."""
    from awesome_amd.model import NormalizingFlow1D
    z = _z(golden_dir, "flow.npz")
    x = torch.from_numpy(z["x"]).to(dev)
    for tag, kw in [("nf6_w130", dict(num_coupling=6, width=130)), ("nf4_w16", dict(num_coupling=4, width=16))]:
        nf = NormalizingFlow1D(in_features=2, backbone="normal_block", **kw)
        nf.load_state_dict(O.load_npz_state(z, f"{tag}.sd."))
        nf.to(dev)
        y = nf(x)
        np.testing.assert_allclose(y.detach().cpu().numpy(), z[f"{tag}.y"], rtol=1e-5, atol=2e-6)
        (y ** 2).mean().backward()
        for k, prm in nf.named_parameters():
            ref = z[f"{tag}.grad.{k}"]
            np.testing.assert_allclose(prm.grad.cpu().numpy().reshape(ref.shape), ref, rtol=1e-3,
                                       atol=1e-5 * float(np.abs(ref).max()) + 1e-7, err_msg=k)


# ---- the pretrain entry point ----------------------------------------------------------------------------------------------
class _Agent:
    """What a prior module's pretrain touches of TorchAgent: training_dataset (+ device, logger)."""

    def __init__(self, ds, dev):
        self.training_dataset, self.device, self.logger = ds, dev, None


def _setup(dev, prior_type, prior_args, n=3, size=48, kind="blob"):
    from awesome_amd.dataset import SyntheticPriorDataset
    from awesome_amd.model import ForwardModule, WrapperModule
    ds = SyntheticPriorDataset(n_images=n, size=size, kind=kind, prior_model_type=prior_type, prior_model_args=prior_args)
    wrapper = WrapperModule(ForwardModule(), prior_type(**prior_args), use_segmentation_output_inversion=True).to(dev)
    return ds, wrapper, _Agent(ds, dev)


def _item(ds, i, dev):
    """(generated prior state, unaries as the wrapper derives them: 1 - sigmoid(segmentation logits)) of item i."""
    (_, state), ((image, _, _), _) = ds[i]
    return state, (1 - torch.sigmoid(image.to(dev))).reshape(-1)


def test_pretrain_icnn_batched_equals_direct_fit_and_fills_the_cache(dev, tmp_path):
    import awesome_amd as A
    from awesome_amd.measures import SE, UnariesWeightedLoss
    from awesome_amd.model import ConvexNextNet
    args = dict(n_hidden=130, in_features=2, n_hidden_layers=1)
    torch.manual_seed(7)
    ds, wrapper, agent = _setup(dev, ConvexNextNet, args, n=3)
    state = wrapper.pretrain(train_set=ds, test_set=None, device=dev, agent=agent, use_progress_bar=False, num_epochs=300, lr=2e-3,
                             optimizer="adam", reuse_state=False, criterion=UnariesWeightedLoss(SE("mean")), do_pretrain_checkpoints=True,
                             pretrain_checkpoint_dir=str(tmp_path / "ckpt"))
    # the reference's state layout (PriorCache.get_state)
    assert set(state) == {"model_type", "model_args", "store_device", "cache"} and sorted(state["cache"]) == ["0", "1", "2"]
    assert json.loads(state["model_args"]) == args and state["model_type"].endswith("ConvexNextNet")
    assert list(state["cache"]["0"].keys()) == list(ConvexNextNet(**args).state_dict().keys())
    rep = wrapper.prior_module.pretrain_report
    assert all(r["iou"] > 0.8 and r["retries"] == 0 and not r["skipped"] for r in rep), rep
    # same result as the direct device fit of the same batch from the same generated priors (plateau(200, .5) as in the reference)
    torch.manual_seed(7)
    ds2, _, _ = _setup(dev, ConvexNextNet, args, n=3)
    spec = A.IcnnSpec(130, 2, 1)
    items = [_item(ds2, i, dev) for i in range(3)]
    init = torch.stack([A.pack_state_dict(spec, st) for st, _ in items]).to(dev)
    un = torch.stack([u for _, u in items])
    direct = A.fit(spec, init, A.Grid.linspace(48, 48, dev), un, 300, lr=2e-3, optimizer="adam", plateau=dict(patience=200, factor=0.5))
    for i in range(3):
        got = A.pack_state_dict(spec, state["cache"][str(i)])
        np.testing.assert_allclose(got.numpy(), direct.params[i].cpu().numpy(), rtol=2e-5, atol=1e-7)
    # per-image checkpoints were written and resume the run without fitting (path_connected_net.py:857-864, 996-998)
    assert sorted(os.listdir(tmp_path / "ckpt")) == [f"pretrain_checkpoint_{i}.pth" for i in range(3)]
    torch.manual_seed(99)
    ds3, wrapper3, agent3 = _setup(dev, ConvexNextNet, args, n=3)
    state3 = wrapper3.pretrain(train_set=ds3, test_set=None, device=dev, agent=agent3, use_progress_bar=False, num_epochs=150,
                               reuse_state=False, use_pretrain_checkpoints=True, pretrain_checkpoint_dir=str(tmp_path / "ckpt"))
    assert all(r["from_checkpoint"] for r in wrapper3.prior_module.pretrain_report)
    for i in range(3):
        for k, v in state["cache"][str(i)].items():
            assert torch.equal(v, state3["cache"][str(i)][k]), (i, k)
    # pretrain_load_state puts a saved state back into the data set's cache (path_connected_net.py:1010-1019)
    torch.manual_seed(5)
    ds4, wrapper4, agent4 = _setup(dev, ConvexNextNet, args, n=3)
    wrapper4.pretrain_load_state(train_set=ds4, test_set=None, device=dev, agent=agent4, state=state, use_progress_bar=False)
    assert torch.equal(ds4.__prior_cache__[1]["out.ln.weight"], state["cache"]["1"]["out.ln.weight"])


def test_pretrain_argument_errors_like_the_reference(dev):
    from awesome_amd.model import ConvexNextNet, ForwardModule, WrapperModule
    ds, wrapper, agent = _setup(dev, ConvexNextNet, dict(n_hidden=32), n=1, size=16)
    with pytest.raises(ValueError, match="Wrapper model must be provided"):
        wrapper.prior_module.pretrain(train_set=ds, test_set=None, device=dev, agent=agent)
    with pytest.raises(ValueError, match="Agent must be trained on a prior dataset"):
        wrapper.pretrain(train_set=ds, test_set=None, device=dev, agent=_Agent(object(), dev))
    with pytest.raises(ValueError, match="Pretrain checkpoint dir must be provided"):
        wrapper.pretrain(train_set=ds, test_set=None, device=dev, agent=agent, do_pretrain_checkpoints=True)
    w2 = WrapperModule(ForwardModule(), torch.nn.Linear(2, 1))
    with pytest.raises(ValueError, match="Prior module must be a PretrainableModule"):
        w2.pretrain(train_set=ds, test_set=None, device=dev, agent=agent)


def test_pretrain_gate_retry_and_skip(dev):
    """An impossible threshold forces reset + refit (retries counted, :964-985); an image without foreground is skipped and keeps
    its generated prior (:848-855)."""
    from awesome_amd.model import ConvexNextNet
    torch.manual_seed(3)
    ds, wrapper, agent = _setup(dev, ConvexNextNet, dict(n_hidden=32), n=2, size=32)
    full_bg = torch.full((32, 32), 1.0)
    ds._inner.unaries = lambda i, orig=ds._inner.unaries: full_bg if i == 1 else orig(i)
    init1 = {k: v.clone() for k, v in ds[1][0][1].items()}
    state = wrapper.pretrain(train_set=ds, test_set=None, device=dev, agent=agent, use_progress_bar=False, num_epochs=20,
                             reuse_state=False, proper_prior_fit_threshold=1.1, proper_prior_fit_retrys=2)
    rep = wrapper.prior_module.pretrain_report
    assert rep[0]["retries"] == 2 and rep[1]["skipped"]
    for k, v in init1.items():
        assert torch.equal(state["cache"]["1"][k], v)


def test_pretrain_path_connected_net_with_prefits(dev):
    """real_nvp_path_connected_net through the same entry point: ActNorm init, flow-identity and convex pre-fits, the joint
    Adamax loop, the cache in the reference's key layout; against the same stages called by hand on the HIP API."""
    import awesome_amd as A
    from awesome_amd import rnvp as R
    from awesome_amd.model import real_nvp_path_connected_net
    args = dict(channels=2, hidden_units=16, flow_n_flows=4, flow_output_fn="tanh", convex_net_hidden_units=64, convex_net_hidden_layers=2)
    pre = dict(num_epochs=60, lr=1e-3, reuse_state=False, proper_prior_fit_threshold=0.0, prefit_flow_net_identity=True, prefit_flow_net_identity_num_epochs=20,
               prefit_convex_net=True, prefit_convex_net_num_epochs=30)
    torch.manual_seed(11)
    ds, wrapper, agent = _setup(dev, real_nvp_path_connected_net, args, n=2, size=32)
    state = wrapper.pretrain(train_set=ds, test_set=None, device=dev, agent=agent, use_progress_bar=False, **pre)
    sd0 = state["cache"]["0"]
    assert list(sd0.keys()) == list(real_nvp_path_connected_net(**args).state_dict().keys())
    assert all(float(v) == 1.0 for k, v in sd0.items() if k.endswith("data_dep_init_done"))
    # by hand
    torch.manual_seed(11)
    ds2, wrapper2, _ = _setup(dev, real_nvp_path_connected_net, args, n=2, size=32)
    m = wrapper2.prior_module
    ispec, rspec = m._specs()
    items = [_item(ds2, i, dev) for i in range(2)]
    flats = torch.stack([m._engine_pack(st) for st, _ in items]).to(dev)
    ip, fp = flats[:, :ispec.n_params].contiguous(), flats[:, ispec.n_params:].contiguous()
    grid = A.Grid.explicit(ds2._xy.reshape(2, -1).to(dev))
    un = torch.stack([u for _, u in items])
    R.actnorm_init(rspec, fp, grid)
    R.fit_identity(rspec, fp, grid, steps=20, lr=1e-2, weight_decay=1e-5)
    A.fit(ispec, ip, A.Grid.explicit(R.rnvp_forward(rspec, fp, grid)), un, 30, lr=1e-3, loss="se", optimizer="adam", plateau=None,
          want_logits=False)
    res = R.pcn_fit(ispec, rspec, ip, fp, grid, un, 60, lr=1e-3, optimizer="adamax", flow_weight_decay=1e-5,
                    plateau=dict(patience=200, factor=0.5))
    for i in range(2):
        got = m._engine_pack(state["cache"][str(i)])
        ref = torch.cat([res.icnn_params[i], res.flow_params[i]]).cpu()
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-5, atol=1e-7)


def test_pretrain_convex_diffeomorphism_net_warm_start_chain(dev):
    """ConvexDiffeomorphismNet.pretrain with reuse_state: frame 0 trains num_epochs, the later frames start from the previous
    proper fit, are shifted to the new centre of mass (translate_only_point, :337-348) and train reuse_state_epochs."""
    import awesome_amd as A
    from awesome_amd import flow as FL
    from awesome_amd.dataset import convex_blob_unaries
    from awesome_amd.model import ConvexDiffeomorphismNet
    from awesome_amd.model.diffeomorphism_net import translate_linear, translate_only_point_args
    from awesome_amd.model.pretrainable_module import center_of_mass
    args = dict(n_hidden=64, n_hidden_layers=1, nf_layers=4, nf_hidden=24, diffeo_args=dict(backbone="normal_block"))
    torch.manual_seed(13)
    ds, wrapper, agent = _setup(dev, ConvexDiffeomorphismNet, args, n=3, size=32)
    base = (convex_blob_unaries(256, 2).reshape(256, 256)[::8, ::8] > 0.5).float()
    frames = [torch.roll(base, shifts=(0, 2 * k), dims=(0, 1)) for k in range(3)]      # the blob moves 2 px per frame
    ds._inner.unaries = lambda i: frames[i]
    state = wrapper.pretrain(train_set=ds, test_set=None, device=dev, agent=agent, use_progress_bar=False, num_epochs=250, lr=3e-3,
                             reuse_state=True, reuse_state_epochs=40)
    rep = wrapper.prior_module.pretrain_report
    assert [r["retries"] for r in rep] == [0, 0, 0] and min(r["iou"] for r in rep) > 0.6, rep
    # by hand: the same chain on the HIP API
    torch.manual_seed(13)
    ds2, wrapper2, _ = _setup(dev, ConvexDiffeomorphismNet, args, n=3, size=32)
    ds2._inner.unaries = lambda i: frames[i]
    items = [_item(ds2, i, dev) for i in range(3)]
    m = wrapper2.prior_module
    ispec, fspec = m._specs()
    P = ispec.n_params
    grid_img = ds2._xy.to(dev)
    grid = A.Grid.explicit(grid_img.reshape(2, -1))
    un = [u[None] for _, u in items]
    flat = m._engine_pack(items[0][0]).to(dev)[None]
    r0 = FL.cdn_fit(ispec, fspec, flat[:, :P].contiguous(), flat[:, P:].contiguous(), grid, un[0], 250, lr=3e-3, loss="bce",
                    weight_decay_on_weight_g=5e-5, plateau=dict(patience=200, factor=0.5))
    prev = torch.cat([r0.icnn_params[0], r0.flow_params[0]])
    com_prev = center_of_mass(un[0].reshape(1, 1, 32, 32))
    for i in (1, 2):
        com = center_of_mass(un[i].reshape(1, 1, 32, 32))
        w, b = prev[P:P + 4].reshape(2, 2).clone(), prev[P + 4:P + 6].clone()
        w, b = translate_linear(w, b, *translate_only_point_args(com_prev.flip(dims=(-1,)), com.flip(dims=(-1,)), grid_img, 2))
        start = prev.clone()
        start[P:P + 4], start[P + 4:P + 6] = w.reshape(-1), b
        com_prev = com
        r = FL.cdn_fit(ispec, fspec, start[None, :P].contiguous(), start[None, P:].contiguous(), grid, un[i], 40, lr=3e-3, loss="bce",
                       weight_decay_on_weight_g=5e-5, plateau=dict(patience=200, factor=0.5))
        prev = torch.cat([r.icnn_params[0], r.flow_params[0]])
        got = m._engine_pack(state["cache"][str(i)])
        np.testing.assert_allclose(got.numpy(), prev.cpu().numpy(), rtol=2e-5, atol=1e-7, err_msg=f"frame {i}")


def test_pretrain_convex_diffeomorphism_net_with_the_resnet_backbone(dev):
    """`pretrain` with the 'resnet' flow backbone: no fused fit exists (batch norms over the points), the reference's loop runs as
    device-side autograd (Adam with weight decay on the weight_g group, ReduceLROnPlateau, enforce_convexity per step) behind the same
    entry point - cold fit of frame 0, centre-of-mass warm starts for the later frames, states into the PriorCache."""
    from awesome_amd.dataset import convex_blob_unaries
    from awesome_amd.model import ConvexDiffeomorphismNet
    args = dict(n_hidden=32, n_hidden_layers=1, nf_layers=2, nf_hidden=8, diffeo_args=dict(backbone="resnet", num_blocks=1))
    torch.manual_seed(5)
    ds, wrapper, agent = _setup(dev, ConvexDiffeomorphismNet, args, n=2, size=32)
    base = (convex_blob_unaries(256, 2).reshape(256, 256)[::8, ::8] > 0.5).float()
    frames = [torch.roll(base, shifts=(0, 2 * k), dims=(0, 1)) for k in range(2)]
    ds._inner.unaries = lambda i: frames[i]
    before = {k: v.detach().clone() for k, v in wrapper.prior_module.state_dict().items()}
    state = wrapper.pretrain(train_set=ds, test_set=None, device=dev, agent=agent, use_progress_bar=False, num_epochs=150, lr=3e-3,
                             reuse_state=True, reuse_state_epochs=30)
    rep = wrapper.prior_module.pretrain_report
    assert len(rep) == 2 and all(np.isfinite(r["iou"]) for r in rep) and rep[0]["iou"] > 0.5, rep
    m = wrapper.prior_module
    for i in range(2):
        sd = state["cache"][str(i)]
        assert list(sd.keys()) == list(m.state_dict().keys())
        assert all(bool(torch.isfinite(v).all()) for v in sd.values())
        assert float((sd["convex_net.linear_out.weight"] if "convex_net.linear_out.weight" in sd else next(iter(sd.values()))).abs().sum()) > 0
    moved = sum(float((state["cache"]["0"][k].cpu() - before[k].cpu()).abs().max()) > 0 for k in before)
    assert moved > len(before) // 2                      # the fit trained the flow and the ICNN
    for k, v in m.state_dict().items():                   # ... and left the module's own parameters where they were
        assert torch.equal(v.cpu(), before[k].cpu()), k


def test_convex_diffeomorphism_net_with_an_icnn_shape_without_a_fused_kernel(dev):
    """ConvexDiffeomorphismNet(n_hidden=160, n_hidden_layers=3): no fused composite exists for that ICNN - forward and backward
    compose the flow kernels and the layer-by-layer ICNN through two autograd bridges (the ICNN's dL/dcoords feeds the flow's backward).
    Forward and the BCE gradients of every parameter vs autograd through the oracle; `pretrain` runs the reference's loop on it."""
    from awesome_amd.dataset import convex_blob_unaries
    from awesome_amd.model import ConvexDiffeomorphismNet
    args = dict(n_hidden=160, n_hidden_layers=3, nf_layers=4, nf_hidden=24, diffeo_args=dict(backbone="normal_block"))
    torch.manual_seed(17)
    m = ConvexDiffeomorphismNet(**args)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    m.to(dev)
    H = W = 24
    xs = torch.linspace(0, 1, W)
    grid = torch.stack([xs[None, :].expand(H, W), xs[:, None].expand(H, W)], 0)[None].contiguous()
    un = (convex_blob_unaries(256, 1).reshape(256, 256)[::11, ::11][:H, :W] > 0.5).float()[None, None]
    logits = m(grid.to(dev))
    ref = O.convex_diffeo_forward(sd, grid[0].reshape(2, -1).t(), 4)
    np.testing.assert_allclose(logits.detach().cpu().numpy().reshape(-1), ref.detach().numpy().reshape(-1), atol=3e-5, rtol=1e-4)
    torch.nn.BCELoss()(torch.sigmoid(logits), un.to(dev)).backward()
    torch.nn.BCELoss()(torch.sigmoid(ref).reshape(un.shape), un).backward()
    gmax = max(float(v.grad.abs().max()) for v in sd.values())
    for k, p in m.named_parameters():
        r = sd[k].grad.numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy().reshape(r.shape), r, rtol=2e-3, atol=2e-5 * gmax, err_msg=k)
    # the pretrain entry point on the same kind of module (generic engine: the reference's loop as device-side autograd) against the
    # oracle's restatement of that loop from the same state (a 3 x 160 ICNN leaves the reference's lr = 3e-3 after a few steps - on
    # the CPU oracle as on the device, to the same logits - hence the smaller rate; the gate is not what this checks)
    torch.manual_seed(17)
    ds, wrapper, agent = _setup(dev, ConvexDiffeomorphismNet, args, n=1, size=32)
    base = (convex_blob_unaries(256, 2).reshape(256, 256)[::8, ::8] > 0.5).float()
    ds._inner.unaries = lambda i: base
    state0, un0 = _item(ds, 0, dev)
    state = wrapper.pretrain(train_set=ds, test_set=None, device=dev, agent=agent, use_progress_bar=False, num_epochs=25, lr=5e-4,
                             proper_prior_fit_threshold=0.0)
    rep = wrapper.prior_module.pretrain_report
    assert len(rep) == 1 and rep[0]["retries"] == 0, rep
    got = state["cache"]["0"]
    sd0 = {k: v.detach().cpu().clone() for k, v in state0.items()}
    pf, losses, _ = O.fit_convex_diffeo(sd0, ds._xy[None].cpu(), un0.reshape(1, 1, 32, 32).cpu(), 25, 4, lr=5e-4, loss_kind="bce",
                                        weight_decay_on_weight_g=5e-5, plateau=dict(patience=200, factor=0.5))
    assert losses[-1] < losses[0]
    for k in pf:
        # (weight_v of a weight-normed 1x1 layer has an analytically zero gradient: Adam turns its rounding noise into +- lr steps, a
        # walk of at most 25 lr on either side)
        tol = dict(rtol=0, atol=2 * 25 * 5e-4) if k.endswith("scale.weight_v") else dict(rtol=2e-3, atol=2e-5)
        np.testing.assert_allclose(got[k].cpu().numpy(), pf[k].numpy(), err_msg=k, **tol)


def test_pretrain_path_connected_net_warm_start_chain_keeps_actnorm(dev):
    """with reuse_state (the default) frame k starts from frame k-1's fitted state, loaded in the reference with
    load_state_dict (path_connected_net.py:867-870) - data_dep_init_done = 1 included - so ActNorm is NOT re-initialised from the
    data and the short refit continues from the previous deformation.  Against the same chain on the HIP API by hand: frame 0 cold
    (ActNorm init + num_epochs), frames 1.. = previous fit -> reuse_state_epochs, no init; and against the wrong chain (init before
    every frame), which must differ."""
    import awesome_amd as A
    from awesome_amd import rnvp as R
    from awesome_amd.dataset import convex_blob_unaries
    from awesome_amd.model import real_nvp_path_connected_net
    args = dict(channels=2, hidden_units=16, flow_n_flows=4, flow_output_fn="tanh", convex_net_hidden_units=64, convex_net_hidden_layers=1)
    pre = dict(num_epochs=80, lr=2e-3, reuse_state=True, reuse_state_epochs=25, proper_prior_fit_threshold=0.0)
    base = (convex_blob_unaries(256, 3).reshape(256, 256)[::8, ::8] > 0.5).float()
    frames = [torch.roll(base, shifts=(0, 2 * k), dims=(0, 1)) for k in range(3)]
    torch.manual_seed(17)
    ds, wrapper, agent = _setup(dev, real_nvp_path_connected_net, args, n=3, size=32)
    ds._inner.unaries = lambda i: frames[i]
    state = wrapper.pretrain(train_set=ds, test_set=None, device=dev, agent=agent, use_progress_bar=False, **pre)
    torch.manual_seed(17)
    ds2, wrapper2, _ = _setup(dev, real_nvp_path_connected_net, args, n=3, size=32)
    ds2._inner.unaries = lambda i: frames[i]
    items = [_item(ds2, i, dev) for i in range(3)]
    m = wrapper2.prior_module
    ispec, rspec = m._specs()
    P = ispec.n_params
    grid = A.Grid.explicit(ds2._xy.reshape(2, -1).to(dev))

    def chain(reinit: bool):
        flat = m._engine_pack(items[0][0]).to(dev)[None]
        ip, fp = flat[:, :P].contiguous(), flat[:, P:].contiguous()
        out = []
        for k in range(3):
            if k == 0 or reinit:
                R.actnorm_init(rspec, fp, grid)
            res = R.pcn_fit(ispec, rspec, ip, fp, grid, items[k][1][None], 80 if k == 0 else 25, lr=2e-3, optimizer="adamax",
                            flow_weight_decay=1e-5, plateau=dict(patience=200, factor=0.5))
            ip, fp = res.icnn_params.clone(), res.flow_params.clone()
            out.append(torch.cat([ip[0], fp[0]]).cpu())
        return out

    good, wrong = chain(False), chain(True)
    for k in range(3):
        got = m._engine_pack(state["cache"][str(k)])
        np.testing.assert_allclose(got.numpy(), good[k].numpy(), rtol=2e-5, atol=1e-7, err_msg=f"frame {k}")
    assert not np.allclose(good[2].numpy(), wrong[2].numpy(), rtol=1e-3, atol=1e-5)
    # the ActNorm parameters ENTERING frame 1 are frame 0's fitted ones: after 0 refit epochs they are unchanged
    torch.manual_seed(17)
    ds3, wrapper3, agent3 = _setup(dev, real_nvp_path_connected_net, args, n=2, size=32)
    ds3._inner.unaries = lambda i: frames[i]
    st3 = wrapper3.pretrain(train_set=ds3, test_set=None, device=dev, agent=agent3, use_progress_bar=False, **dict(pre, reuse_state_epochs=0))
    for key in st3["cache"]["0"]:
        if key.endswith(".s") or key.endswith(".t"):
            assert torch.equal(st3["cache"]["1"][key], st3["cache"]["0"][key]), key


def test_flow_priors_freeze_without_a_status_array(dev):
    """inrfit_cdn_fit / inrfit_pcn_fit called with status == NULL (a direct C-ABI caller) on an image whose loss
    is non-finite: the ICNN AND its deformation stay at their parameters - the flow updates read the frozen flag the ICNN update
    wrote into the optimizer header, not `status`."""
    import ctypes as C
    from awesome_amd import _lib as L, flow as FL, icnn as K, rnvp as R
    from awesome_amd.model import ConvexDiffeomorphismNet, real_nvp_path_connected_net
    import awesome_amd as A
    lib = L.load()
    g = A.Grid.linspace(32, 32, dev)
    un = (torch.rand(1, 32 * 32, device=dev) > 0.5).float()
    un[0, 5] = float("nan")
    od = L.InrOptDesc(L.OPT_KINDS["adam"], 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 0, 200, 0.5, 1e-4, 0.0, 1e-8, 0, 0, 0)
    ld = K._loss_desc("bce", "none", 1.0, 0.0, 0.0)
    torch.manual_seed(4)
    cdn = ConvexDiffeomorphismNet(n_hidden=64, n_hidden_layers=1, nf_layers=4, nf_hidden=24, diffeo_args=dict(backbone="normal_block"))
    ispec, fspec = cdn._specs()
    flat = cdn._engine_pack(cdn.state_dict()).to(dev)[None]
    ip, fp = flat[:, :ispec.n_params].contiguous(), flat[:, ispec.n_params:].contiguous()
    ip0, fp0 = ip.clone(), fp.clone()
    iopt, fopt = K.new_opt_state(ispec, 1, dev), torch.zeros(1, 2 * fspec.n_params, device=dev)
    ws = FL._ws(ispec, fspec, g, 1)
    md, fd, gd = ispec.desc(), fspec.desc(), g.desc()
    rc = lib.inrfit_cdn_fit(C.byref(md), C.byref(fd), ip.data_ptr(), fp.data_ptr(), iopt.data_ptr(), fopt.data_ptr(), C.byref(gd),
                            un.data_ptr(), C.byref(ld), C.byref(od), 5e-5, 1, 4, 0, None, None, None, ws.data_ptr(), ws.numel() * 4,
                            K._stream_ptr(dev))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(ip, ip0) and torch.equal(fp, fp0)
    pcn = real_nvp_path_connected_net(channels=2, hidden_units=16, flow_n_flows=4, flow_output_fn="tanh", convex_net_hidden_units=64)
    ispec, rspec = pcn._specs()
    flat = pcn._engine_pack(pcn.state_dict()).to(dev)[None]
    ip, fp = flat[:, :ispec.n_params].contiguous(), flat[:, ispec.n_params:].contiguous()
    R.actnorm_init(rspec, fp, g)
    ip0, fp0 = ip.clone(), fp.clone()
    iopt, fopt = K.new_opt_state(ispec, 1, dev), torch.zeros(1, 2 * rspec.n_params, device=dev)
    ws = R._ws(ispec, rspec, g, 1)
    md, rd = ispec.desc(), rspec.desc()
    ld = K._loss_desc("se", "none", 1.0, 0.0, 0.0)
    rc = lib.inrfit_pcn_fit(C.byref(md), C.byref(rd), ip.data_ptr(), fp.data_ptr(), iopt.data_ptr(), fopt.data_ptr(), C.byref(gd),
                            un.data_ptr(), C.byref(ld), C.byref(od), 1e-5, 1, 4, 0, None, None, None, ws.data_ptr(), ws.numel() * 4,
                            K._stream_ptr(dev))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(ip, ip0) and torch.equal(fp, fp0)


def test_pretrain_agent_saves_and_reloads_pretrain_state(dev, tmp_path):
    """PretrainAgent._pretrain = TorchAgent._pretrain (:553-627): runs the pretraining once, writes pretrain_state.pth, and a
    second agent pointed at the file loads it instead of fitting."""
    from awesome_amd.agent import PretrainAgent
    from awesome_amd.model import ConvexNextNet
    args = dict(n_hidden=32)
    torch.manual_seed(1)
    ds, wrapper, _ = _setup(dev, ConvexNextNet, args, n=2, size=24)
    agent = PretrainAgent(ds, device=dev, agent_folder=str(tmp_path), pretrain_args=dict(num_epochs=30, reuse_state=False))
    state = agent._pretrain(wrapper, ds, None, use_progress_bar=False)
    path = os.path.join(str(tmp_path), "pretrain_state.pth")
    assert os.path.exists(path) and agent.pretrain_state_path == path
    torch.manual_seed(2)
    ds2, wrapper2, _ = _setup(dev, ConvexNextNet, args, n=2, size=24)
    agent2 = PretrainAgent(ds2, device=dev, agent_folder=str(tmp_path / "other"), pretrain_state_path=path,
                           pretrain_args=dict(num_epochs=30, reuse_state=False))
    agent2._pretrain(wrapper2, ds2, None, use_progress_bar=False)
    assert not hasattr(wrapper2.prior_module, "pretrain_report")          # nothing was fitted
    for k, v in state["cache"]["1"].items():
        assert torch.equal(ds2.__prior_cache__[1][k].cpu(), v.cpu())


def test_pretrain_spatio_temporal_dispatch(dev):
    """A prior data set WITHOUT per-image priors sends PathConnectedNet.pretrain down the spatio-temporal branch
    (path_connected_net.py:472-509 -> :511-728): one (x, y, t) network over all frames, mini-batches of frames, the module's own
    state_dict as the returned state - against the same stages called by hand."""
    from awesome_amd.dataset import SyntheticPriorDataset
    from awesome_amd.model import ForwardModule, WrapperModule, real_nvp_path_connected_net
    args = dict(channels=3, hidden_units=16, flow_n_flows=6, flow_output_fn="tanh", convex_net_hidden_units=64, convex_net_hidden_layers=2)
    pre = dict(num_epochs=6, lr=1e-3, batch_size=2, prefit_flow_net_identity=True, prefit_flow_net_identity_num_epochs=10,
               prefit_convex_net=True, prefit_convex_net_num_epochs=15)
    ds = SyntheticPriorDataset(n_images=4, size=24, kind="sequence")           # 4 frames, no prior_model_type
    assert not ds.has_prior and ds[2][0][2].shape == (3, 24, 24)
    torch.manual_seed(17)
    model = real_nvp_path_connected_net(**args).to(dev)
    start = {k: v.detach().clone() for k, v in model.state_dict().items()}
    wrapper = WrapperModule(ForwardModule(), model, use_segmentation_output_inversion=True).to(dev)
    state = wrapper.pretrain(train_set=ds, test_set=None, device=dev, agent=_Agent(ds, dev), use_progress_bar=False, **pre)
    assert list(state.keys()) == list(start.keys()) and len(model.pretrain_epoch_losses) == 6
    assert model.pretrain_epoch_losses[-1] < model.pretrain_epoch_losses[0]
    # by hand
    import awesome_amd as A
    twin = real_nvp_path_connected_net(**args).to(dev)
    twin.load_state_dict(start)
    frames = [(1 - torch.sigmoid(ds[i][0][0].to(dev))).reshape(-1) for i in range(4)]
    coords = torch.stack([ds[i][0][2].reshape(3, -1) for i in range(4)]).to(dev)
    whole = A.Grid.explicit(coords.permute(1, 0, 2).reshape(3, -1).contiguous())
    twin.learn_flow_identity(whole, lr=1e-2, weight_decay=1e-5, max_iter=10)
    twin.learn_convex_net(whole, torch.stack(frames).reshape(1, -1), lr=1e-3, weight_decay=0.0, max_iter=15)
    losses = twin.fit_sequence(coords, torch.stack(frames), num_epochs=6, lr=1e-3, flow_weight_decay=1e-5, batch_size=2)
    np.testing.assert_allclose(model.pretrain_epoch_losses, losses, rtol=1e-6)
    for k, v in twin.state_dict().items():
        if v.is_floating_point() and not k.endswith("data_dep_init_done"):
            assert torch.equal(v, state[k]), k


def test_evaluation_and_export_after_pretrain(dev, tmp_path):
    """SURVEY §8 f2 on the HIP path: after `wrapper.pretrain(...)` the reference-style `get_result` / `split_model_result` /
    `save_result_mask` chain per image, and `evaluate_dataset` for the whole set at once on the device (threshold + integer-count
    IoU + bit-packed masks, one kernel each): same IoU as the per-image chain with the oracle's IoU, PNGs that decode to the
    thresholded prior masks."""
    PIL = pytest.importorskip("PIL.Image")
    from awesome_amd.model import ConvexNextNet
    from awesome_amd.run import evaluate_dataset, get_result, save_result_mask, split_model_result
    args = dict(n_hidden=130, in_features=2, n_hidden_layers=1)
    torch.manual_seed(11)
    ds, wrapper, agent = _setup(dev, ConvexNextNet, args, n=3, size=40)
    wrapper.pretrain(train_set=ds, test_set=None, device=dev, agent=agent, use_progress_bar=False, num_epochs=300, lr=2e-3,
                     optimizer="adam", reuse_state=False)
    ev = evaluate_dataset(wrapper, ds, out_dir=str(tmp_path / "masks"))
    assert ev["shape"] == (40, 40) and ev["bits"].shape == (3, 25) and ev["indices"] == [0, 1, 2]
    for k in range(3):
        res, gt, image, _, _ = get_result(wrapper, ds, k, model_gets_targets=False)
        assert res.device.type == "cpu" and res.shape == (1, 2, 40, 40)
        ret = split_model_result(res, wrapper, ds, image)
        prior = ret["prior"]                                     # (1, H, W), after the wrapper's sigmoid (use_prior_sigmoid)
        assert prior.shape == (1, 40, 40) and ret["segmentation"].shape == (1, 40, 40)
        p = prior if getattr(wrapper, "use_prior_sigmoid", False) else torch.sigmoid(prior)
        obj = (p[0] <= 0.5)                                      # foreground -> 0 convention
        gt_obj = gt[0] <= 0.5                                    # the data set's masks use the same convention
        inter, union = float((obj & gt_obj).sum()), float((obj | gt_obj).sum())
        assert float(ev["iou"][k]) == pytest.approx(inter / union, abs=1e-6)
        assert float(ev["iou"][k]) == pytest.approx(O.miou_binary((p > 0.5).float(), gt, invert=True), abs=1e-6)   # MIOU (awesome/measures/miou.py:29-48)
        im = np.asarray(PIL.open(str(tmp_path / "masks" / f"{k}.png"))).astype(bool)
        np.testing.assert_array_equal(im, obj.numpy())
        # the reference's own export: binary channel mask (object = 0) -> colour-index PNG
        save_result_mask((p > 0.5).float(), str(tmp_path / f"ref_{k}.png"))
        np.testing.assert_array_equal(np.asarray(PIL.open(str(tmp_path / f"ref_{k}.png"))), obj.numpy().astype(np.uint8))
    assert ev["miou"] == pytest.approx(float(ev["iou"].mean())) and ev["miou"] > 0.8
