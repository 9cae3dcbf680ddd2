"""GPU parity of the joint-training step path (BASELINE configs[4]) and of configs[3] at full size.

configs[4]: UNet-logit refinement - noisy 256x256 pseudo-labels, a segmentation module, a per-image shape prior and
FBMSJointLoss, stepped like TorchAgent._perform_step (awesome/agent/torch_agent.py:428-551) with the per-image prior swap of
PriorManager (awesome/dataset/prior_dataset.py:96-110).  HIP path: WrapperModule -> prior forward/backward kernels, the fused
FBMSJointLoss (`inrfit_joint_loss`), PriorBank's zero-copy swap.  Checker: the CPU oracle (pure torch restatement pinned on
fixtures of the real WrapperModule / FBMSJointLoss / ConvexNextNet classes), never a copy of the HIP path itself."""
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


def test_fused_fbms_joint_loss_matches_reference_class(dev, golden_dir):
    """inrfit_joint_loss (value, both gradient channels, both clip branches) against the reference class's fixture, and the
    class-weighted default criterion (UnariesWeightedLoss(BCELoss, 'sssdms'), counts over the batch) against the oracle."""
    from awesome_amd.measures import FBMSJointLoss, SE, UnariesWeightedLoss
    z = np.load(os.path.join(golden_dir, "fbms_joint_loss.npz"))
    for case in range(3):
        out = torch.from_numpy(z[f"c{case}.output"]).to(dev).requires_grad_(True)
        tgt = torch.from_numpy(z[f"c{case}.target"]).to(dev)
        crit = FBMSJointLoss(criterion=torch.nn.BCELoss(), alpha=1.0, beta=float(z[f"c{case}.beta"]))
        assert crit._fused_desc(out) is not None                      # the HIP loss is what runs
        loss = crit(out, tgt)
        loss.backward()
        assert float(loss.detach()) == pytest.approx(float(z[f"c{case}.loss"]), rel=2e-6)
        np.testing.assert_allclose(out.grad.cpu().numpy(), z[f"c{case}.grad"], rtol=2e-5, atol=1e-9)
    torch.manual_seed(3)
    for kind, mode, alpha, beta in (("bce", "sssdms", 0.7, 3.0), ("se", "equal", 1.0, 0.2), ("se", "ratio", 1.3, 40.0), ("bce", "none", 1.0, 1.0)):
        out_c = torch.rand(3, 2, 17, 13) * 0.96 + 0.02
        tgt_c = (torch.rand(3, 1, 17, 13) > 0.8).float()              # few foreground pixels: a non-trivial class weight
        inner = torch.nn.BCELoss() if kind == "bce" else SE("mean")
        crit = FBMSJointLoss(criterion=UnariesWeightedLoss(inner, mode=mode, ratio=0.5), alpha=alpha, beta=beta)
        o_h = out_c.to(dev).requires_grad_(True)
        l_h = crit(o_h, tgt_c.to(dev))
        l_h.backward()
        o_r = out_c.clone().requires_grad_(True)
        l_r = O.fbms_joint_loss(o_r, tgt_c, alpha=alpha, beta=beta, kind=kind, mode=mode, ratio=0.5)
        l_r.backward()
        assert float(l_h.detach()) == pytest.approx(float(l_r.detach()), rel=5e-6), (kind, mode)
        np.testing.assert_allclose(o_h.grad.cpu().numpy(), o_r.grad.numpy(), rtol=5e-5, atol=1e-9, err_msg=f"{kind} {mode}")


def _oracle_joint_epochs(conv_w, conv_b, prior_states, images, grid, targets, order, lr, alpha, beta):
    """TorchAgent._perform_step restated on the CPU: ONE torch Adam over (segmentation conv, the single prior model's
    parameters); the prior's parameter VALUES are swapped per image (PriorManager), its Adam moments are shared - exactly what
    the reference's single optimizer over model.parameters() does."""
    conv = torch.nn.Conv2d(1, 1, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(conv_w)
        conv.bias.copy_(conv_b)
    keys = list(prior_states[0].keys())
    prior = {k: torch.nn.Parameter(prior_states[0][k].clone()) for k in keys}
    opt = torch.optim.Adam(list(conv.parameters()) + [prior[k] for k in keys], lr=lr)
    states = [{k: v.clone() for k, v in st.items()} for st in prior_states]
    losses = []
    for i in order:
        with torch.no_grad():
            for k in keys:
                prior[k].copy_(states[i][k])                                  # PriorManager.__enter__
        opt.zero_grad()
        seg_logits = conv(images[i][None])
        out = O.wrapper_forward(seg_logits, O.icnn_forward_image(prior, grid), invert_seg=True)
        loss = O.fbms_joint_loss(out, targets[i][None], alpha=alpha, beta=beta, kind="bce", mode="sssdms")
        loss.backward()
        opt.step()
        O.icnn_enforce_convexity(prior)                                       # batch_processed hook
        losses.append(float(loss.detach()))
        states[i] = {k: prior[k].detach().clone() for k in keys}              # PriorManager.__exit__
    return conv, states, losses


class _SegStandIn(torch.nn.Module):
    """Stand-in for the UNet (out of scope: the backbone runs on torch / MIOpen as it is): one 3x3 convolution over the noisy
    logit image; like the reference's backbones it is called with the image plus the extra inputs (features ...)."""

    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(1, 1, 3, padding=1)

    def forward(self, image, *args, **kwargs):
        return self.conv(image)


@pytest.mark.parametrize("fused", [True, False])
def test_configs4_joint_steps_on_noisy_pseudo_labels(dev, fused):
    """(fused = True: everything behind the segmentation output is ONE C-ABI call per step, `inrfit_joint_step`; False: the
    autograd bridges + torch.optim.  Both against the same oracle.)
    BASELINE configs[4] at its full grid size (256x256 noisy pseudo-labels): two epochs over three images, every step = swap
    the image's prior in, WrapperModule forward, FBMSJointLoss, backward, Adam, enforce_convexity.  HIP vs the oracle (six CPU
    steps at 65 536 points): losses, the shared segmentation weights and every image's prior parameters."""
    from awesome_amd.agent import JointTrainer
    from awesome_amd.dataset import SyntheticPriorDataset
    from awesome_amd.measures import FBMSJointLoss
    from awesome_amd.model import ConvexNextNet, WrapperModule
    from awesome_amd.prior_bank import PriorBank
    S, n, lr, alpha, beta = 256, 3, 2e-3, 1.0, 2.0
    torch.manual_seed(21)
    ds = SyntheticPriorDataset(n_images=n, size=S, kind="noisy_blob")
    items = [ds[i] for i in range(n)]                       # no prior attached: ((image, feat, xy), target)
    images = [it[0][0] for it in items]
    targets = [it[1] for it in items]
    xy = items[0][0][2]
    seg = _SegStandIn()
    conv_w, conv_b = seg.conv.weight.detach().clone(), seg.conv.bias.detach().clone()
    factory_states = []

    def factory():
        m = ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=1)
        factory_states.append({k: v.detach().clone() for k, v in m.state_dict().items()})
        return m

    wrapper = WrapperModule(seg, ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=1), use_segmentation_output_inversion=True).to(dev)
    bank = PriorBank(lambda: factory().to(dev), n_images=n, device=dev)
    factory_states.clear()                                    # (the bank's probe model)
    for k in range(n):
        bank.row(k)                                           # generate the three priors in key order
    init_states = [dict(s) for s in factory_states[:n]]
    crit = FBMSJointLoss(alpha=alpha, beta=beta)              # default criterion: UnariesWeightedLoss(BCELoss, 'sssdms')
    opt = torch.optim.Adam(list(seg.parameters()) + list(wrapper.prior_module._ordered_params()), lr=lr)
    trainer = JointTrainer(wrapper, bank, crit, opt, fused=fused)
    assert trainer.fused == fused
    order = [0, 1, 2, 0, 1, 2]
    feat = torch.zeros(1, 1, 1, 1, device=dev)
    losses = []
    for i in order:
        loss, out = trainer.perform_step(i, (images[i][None].to(dev), feat, xy[None].to(dev)), targets[i][None].to(dev))
        assert out.shape == (1, 2, S, S)
        losses.append(float(loss))
    conv_ref, states_ref, losses_ref = _oracle_joint_epochs(conv_w, conv_b, init_states, images, xy[None], targets, order, lr, alpha, beta)
    np.testing.assert_allclose(losses, losses_ref, rtol=2e-5)
    np.testing.assert_allclose(seg.conv.weight.detach().cpu().numpy(), conv_ref.weight.detach().numpy(), rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(seg.conv.bias.detach().cpu().numpy(), conv_ref.bias.detach().numpy(), rtol=2e-4, atol=2e-6)
    for k in range(n):
        got = bank.state_dict(k)
        for name, ref in states_ref[k].items():
            np.testing.assert_allclose(got[name].numpy(), ref.numpy(), rtol=5e-4, atol=5e-6, err_msg=f"image {k} {name}")
    # the bank exports the reference's cache layout
    st = bank.get_state(model_args=dict(n_hidden=130))
    assert sorted(st["cache"]) == ["0", "1", "2"] and list(st["cache"]["0"]) == list(init_states[0])


def test_prior_bank_with_path_connected_prior_joint_step(dev):
    """The configs[4] prior proper (convexity + path-connectedness: PathConnectedNet) in the bank: one joint step through
    WrapperModule + FBMSJointLoss vs the oracle's RealNVP restatement with autograd (loss and every prior gradient)."""
    from awesome_amd.dataset import SyntheticPriorDataset
    from awesome_amd.measures import FBMSJointLoss
    from awesome_amd.model import ForwardModule, WrapperModule, real_nvp_path_connected_net
    from awesome_amd.prior_bank import PriorBank
    S = 32
    torch.manual_seed(8)
    args = dict(channels=2, hidden_units=16, flow_n_flows=4, flow_output_fn="tanh", convex_net_hidden_units=64, convex_net_hidden_layers=2)
    ds = SyntheticPriorDataset(n_images=2, size=S, kind="noisy_blob")
    (image, _, xy), target = ds[1]
    bank = PriorBank(lambda: real_nvp_path_connected_net(**args).to(dev), n_images=2, device=dev)
    model = real_nvp_path_connected_net(**args).to(dev)
    wrapper = WrapperModule(ForwardModule(), model, use_segmentation_output_inversion=True).to(dev)
    with torch.no_grad():                                     # non-trivial last layers (they are zero-initialised)
        for name, p in model.named_parameters():
            if ".net.2." in name:
                p.add_(0.05 * torch.randn_like(p))
    with bank.manager(model, 1):
        with torch.no_grad():
            for name, p in model.named_parameters():          # the bank row now holds the perturbed values too
                if ".net.2." in name:
                    p.add_(0.05 * torch.randn_like(p))
        out = wrapper(image[None].to(dev), torch.zeros(1, 1, 1, 1, device=dev), xy[None].to(dev))
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}     # after ActNorm's data-dependent init
        loss = FBMSJointLoss(alpha=1.0, beta=2.0)(out, target[None].to(dev))
        loss.backward()
        grads = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}
    _, rspec = model._specs()
    sdo = {k: v.clone().requires_grad_(v.is_floating_point() and k in grads) for k, v in sd.items()}
    rows = O.pixelize(xy[None])
    masks = O.rnvp_masks(2, args["flow_n_flows"])
    y = O.pcn_forward(sdo, rows, masks, torch.tensor(rspec.vmin), torch.tensor(rspec.vmax), output_fn="tanh", output_scale=None)
    out_ref = O.wrapper_forward(image[None], O.unpixelize(y, 1, S, S), invert_seg=True)
    loss_ref = O.fbms_joint_loss(out_ref, target[None], alpha=1.0, beta=2.0, kind="bce", mode="sssdms")
    loss_ref.backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), out_ref.detach().numpy(), atol=5e-6, rtol=1e-4)
    assert float(loss.detach()) == pytest.approx(float(loss_ref.detach()), rel=2e-5)
    for k, g in grads.items():
        ref = sdo[k].grad
        if ref is None:
            continue
        ref = ref.numpy()
        np.testing.assert_allclose(g.numpy().reshape(ref.shape), ref, rtol=2e-3, atol=2e-5 * float(np.abs(ref).max()) + 1e-8, err_msg=k)


def test_configs3_full_size_xyt_loss_and_gradients(dev):
    """BASELINE configs[3] at FULL size: PathConnectedNet (C = 3, 18 flows x 32 hidden units, tanh outputs, ICNN 130 x 2) on the
    128 x 128 x 16 (x, y, t) grid = 262 144 points in one launch sequence - loss and the gradient of every parameter against
    the oracle's autograd (parity for this variant is UNPINNED: the oracle restates normflows' published definitions)."""
    import awesome_amd as A
    from awesome_amd import rnvp as R
    from awesome_amd.dataset import SyntheticSequenceDataset
    from tests.test_gpu_rnvp import _case, _merge, _split
    ispec, rspec, sd = _case(3, 32, 18, 2, seed=33)
    rspec = R.RnvpSpec(3, 32, 18, "tanh", None, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))   # the factory's MinMax: [0, 1] per channel
    ds = SyntheticSequenceDataset(1, 128, 16)
    coords, un = ds.coords(), ds.batch([0])
    assert coords.shape == (3, 262144)
    masks = O.rnvp_masks(3, 18)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    torch.set_num_threads(8)
    y = O.pcn_forward(sdo, coords.t(), masks, torch.tensor(rspec.vmin), torch.tensor(rspec.vmax), output_fn="tanh", output_scale=None)
    lo = O.weighted_loss(torch.sigmoid(y).reshape(1, 1, -1, 1), un.reshape(1, 1, -1, 1), "se", "none")
    lo.backward()
    ip, fp = _split(ispec, rspec, sd, dev)
    grid = A.Grid.explicit(coords.to(dev))
    logits = R.pcn_forward(ispec, rspec, ip, fp, grid)
    np.testing.assert_allclose(logits[0].cpu().numpy(), y.detach().reshape(-1).numpy(), atol=2e-4, rtol=1e-3)
    loss, gi, gf = R.pcn_loss_grad(ispec, rspec, ip, fp, grid, un.to(dev), loss="se")
    assert float(loss[0]) == pytest.approx(float(lo.detach()), rel=2e-5)
    got = _merge(ispec, rspec, gi[0].cpu(), gf[0].cpu())
    for k, v in got.items():
        ref = sdo[k].grad.numpy()
        np.testing.assert_allclose(v.numpy().reshape(ref.shape), ref, rtol=5e-3, atol=5e-5 * float(np.abs(ref).max()) + 1e-9, err_msg=k)


def _joint_setup(dev, prior_factory, S=48, n=2, seed=5):
    from awesome_amd.dataset import SyntheticPriorDataset
    from awesome_amd.model import WrapperModule
    from awesome_amd.prior_bank import PriorBank
    torch.manual_seed(seed)
    ds = SyntheticPriorDataset(n_images=n, size=S, kind="noisy_blob")
    items = [ds[i] for i in range(n)]
    seg = _SegStandIn()
    wrapper = WrapperModule(seg, prior_factory(), use_segmentation_output_inversion=True).to(dev)
    bank = PriorBank(lambda: prior_factory().to(dev), n_images=n, device=dev)
    for k in range(n):
        bank.row(k)
    return items, seg, wrapper, bank


def _run_joint(dev, prior_factory, crit_factory, fused, steps, lr=2e-3, opt_type=torch.optim.Adam, perturb=None, relabel=None):
    """`steps` joint steps from identical starting points; returns (losses, seg weights, bank rows)."""
    from awesome_amd.agent import JointTrainer
    items, seg, wrapper, bank = _joint_setup(dev, prior_factory)
    if perturb is not None:
        perturb(wrapper.prior_module, bank)
    crit = crit_factory()
    from awesome_amd.prior_bank import _ordered_parameters
    opt = opt_type(list(seg.parameters()) + list(_ordered_parameters(wrapper.prior_module)), lr=lr)
    trainer = JointTrainer(wrapper, bank, crit, opt, fused=fused)
    assert trainer.fused == fused
    feat = torch.zeros(1, 1, 1, 1, device=dev)
    losses, outs = [], []
    for s in range(steps):
        i = s % len(items)
        (image, _, xy), target = items[i]
        if relabel is not None:
            target = relabel(target, 100 + i)
        loss, out = trainer.perform_step(i, (image[None].to(dev), feat, xy[None].to(dev)), target[None].to(dev))
        losses.append(float(loss))
        outs.append(out.cpu())
    return losses, seg.conv.weight.detach().cpu().clone(), bank.params.detach().cpu().clone(), outs, trainer


def _nonzero_last_layers(model, bank):
    with torch.no_grad():
        g = torch.Generator().manual_seed(3)
        for i in range(len(bank)):
            with bank.manager(model, i):
                for name, p in model.named_parameters():
                    if ".net.2." in name or "out_linear" in name or "linear2" in name:
                        p.add_((0.05 * torch.randn(p.shape, generator=g)).to(p.device))
        for b_name, b in model.named_buffers():
            if b_name.endswith("data_dep_init_done"):
                b.fill_(1.0)


@pytest.mark.parametrize("family", ["pcn", "cdn"])
def test_fused_joint_step_with_path_connected_priors(dev, family):
    """inrfit_pcn_joint_step / inrfit_cdn_joint_step (one C-ABI call per step: deformation forward, ICNN step kernel with the
    coordinate gradient, the composite loss with the clip on the device, both update kernels with the detached clip factor) against
    the autograd step (WrapperModule forward -> fused FBMSJointLoss -> HIP backward bridges -> torch.optim.Adam -> enforce_convexity)
    from the same start: losses, outputs, the segmentation weights and every prior row.  (The autograd step itself is pinned on the
    oracle in test_prior_bank_with_path_connected_prior_joint_step / test_configs4...)"""
    from awesome_amd.measures import FBMSJointLoss
    from awesome_amd.model import ConvexDiffeomorphismNet, real_nvp_path_connected_net
    if family == "pcn":
        factory = lambda: real_nvp_path_connected_net(channels=2, hidden_units=16, flow_n_flows=4, flow_output_fn="tanh",   # noqa: E731
                                                      convex_net_hidden_units=64, convex_net_hidden_layers=2)
    else:
        factory = lambda: ConvexDiffeomorphismNet(n_hidden=64, n_hidden_layers=2, nf_layers=4, nf_hidden=24,                # noqa: E731
                                                  diffeo_args=dict(backbone="normal_block"))
    for beta in (2.0, 400.0):   # without and with the penalty clip
        crit = lambda: FBMSJointLoss(alpha=1.0, beta=beta)   # noqa: E731
        lf, wf, rf, of, tf = _run_joint(dev, factory, crit, True, 4, perturb=_nonzero_last_layers)
        la, wa, ra, oa, _ = _run_joint(dev, factory, crit, False, 4, perturb=_nonzero_last_layers)
        assert int(tf.last_status[0]) == 0
        np.testing.assert_allclose(lf, la, rtol=2e-5)
        for a, b in zip(of, oa):
            np.testing.assert_allclose(a.numpy(), b.numpy(), atol=2e-5)
        np.testing.assert_allclose(wf.numpy(), wa.numpy(), rtol=2e-4, atol=2e-6)
        # Adam's first steps move every parameter by ~lr whatever the gradient's size: compare on that scale
        np.testing.assert_allclose(rf.numpy(), ra.numpy(), rtol=1e-3, atol=2e-5)
        assert not np.allclose(rf.numpy(), _joint_setup(dev, factory)[3].params.cpu().numpy())   # ... and they did move


def test_fused_joint_step_awesome_image_loss_and_adamax(dev):
    """AwesomeImageLoss (before its extra penalty) in the fused joint step, Adam and Adamax, against the autograd step; with the
    extra penalty on the trainer takes the autograd step by itself (two data terms on the prior have no fused form)."""
    from awesome_amd.measures import AwesomeImageLoss, SE, UnariesWeightedLoss
    from awesome_amd.model import ConvexNextNet
    factory = lambda: ConvexNextNet(n_hidden=64, in_features=2, n_hidden_layers=2)   # noqa: E731
    for opt_type, crit in ((torch.optim.Adam, lambda: AwesomeImageLoss(alpha=0.7)),
                           (torch.optim.Adamax, lambda: AwesomeImageLoss(criterion=UnariesWeightedLoss(SE("mean"), mode="sssdms"),
                                                                         prior_criterion=UnariesWeightedLoss(torch.nn.BCELoss(), mode="equal"),
                                                                         alpha=1.3))):
        lf, wf, rf, of, _ = _run_joint(dev, factory, crit, True, 4, opt_type=opt_type)
        la, wa, ra, oa, _ = _run_joint(dev, factory, crit, False, 4, opt_type=opt_type)
        np.testing.assert_allclose(lf, la, rtol=2e-5)
        np.testing.assert_allclose(wf.numpy(), wa.numpy(), rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(rf.numpy(), ra.numpy(), rtol=1e-3, atol=2e-5)

    def with_penalty():
        c = AwesomeImageLoss(alpha=0.7)
        c.extra_penalty = True
        return c
    from awesome_amd.agent import JointTrainer
    items, seg, wrapper, bank = _joint_setup(dev, factory)
    c = with_penalty()
    tr = JointTrainer(wrapper, bank, c, torch.optim.Adam(list(seg.parameters()) + list(wrapper.prior_module._ordered_params()), lr=1e-3))
    (image, _, xy), target = items[0]
    loss, out = tr.perform_step(0, (image[None].to(dev), torch.zeros(1, 1, 1, 1, device=dev), xy[None].to(dev)), target[None].to(dev))
    assert torch.isfinite(loss) and tr.last_status is None        # the fused call refused (INR_EUNSUPPORTED is never reached: planned out)


def test_fused_joint_step_nonfinite_loss_freezes_the_row(dev):
    """A NaN in the segmentation output: status = 1, the prior row and its optimizer state stay untouched."""
    import awesome_amd as A
    from awesome_amd import joint as J
    from awesome_amd.model import ConvexNextNet
    torch.manual_seed(0)
    m = ConvexNextNet(n_hidden=64, in_features=2, n_hidden_layers=1)
    row = m.flat_parameters().to(dev)
    keep = row.clone()
    S = 32
    grid = A.Grid.linspace(S, S, dev)
    seg = torch.rand(S * S, device=dev) * 0.9 + 0.05
    tgt = (torch.rand(S * S, device=dev) > 0.5).float()
    opt = torch.zeros(2 * m.spec.n_params + 8, device=dev)
    good = J.joint_step(m.spec, row, opt, grid, seg, tgt, J.joint_desc(), step=1, lr=1e-3)
    assert int(good.status[0]) == 0 and not torch.equal(row, keep) and torch.isfinite(good.loss).all()
    row2, opt2 = keep.clone(), torch.zeros_like(opt)
    seg_bad = seg.clone()
    seg_bad[7] = float("nan")
    bad = J.joint_step(m.spec, row2, opt2, grid, seg_bad, tgt, J.joint_desc(), step=1, lr=1e-3)
    assert int(bad.status[0]) == 1 and torch.equal(row2, keep) and float(opt2[: 2 * m.spec.n_params].abs().sum()) == 0.0


def test_awesome_image_and_pixel_losses_on_device(dev, golden_dir):
    """Row a12: AwesomeImageLoss (with and without extra_penalty) and AwesomeLoss (pixel mode) as forms of inrfit_joint_loss: values
    against the reference classes' fixtures (losses.npz, pixel_losses.npz), gradients against autograd through the same classes'
    torch composition on the CPU (which the CPU suite pins on the same fixtures)."""
    from awesome_amd.measures import AwesomeImageLoss, AwesomeLoss, SE, UnariesWeightedLoss
    z = np.load(os.path.join(golden_dir, "losses.npz"))
    out2, tb = torch.from_numpy(z["output2"]), torch.from_numpy(z["target_bin"])
    for penalty, key in ((False, "ail.plain"), (True, "ail.penalty")):
        crit = AwesomeImageLoss(alpha=0.7)
        crit.extra_penalty = penalty
        assert crit.joint_desc() is not None
        o_h = out2.to(dev).requires_grad_(True)
        l_h = crit(o_h, tb.to(dev))
        l_h.backward()
        assert float(l_h.detach()) == pytest.approx(float(z[key]), rel=3e-6)
        o_c = out2.clone().requires_grad_(True)
        crit(o_c, tb).backward()                                           # CPU tensors: the torch composition
        np.testing.assert_allclose(o_h.grad.cpu().numpy(), o_c.grad.numpy(), rtol=3e-5, atol=1e-9)
    torch.manual_seed(11)
    out_c = torch.rand(3, 2, 17, 13) * 0.96 + 0.02
    tgt_c = (torch.rand(3, 1, 17, 13) > 0.8).float()
    for penalty in (False, True):
        crit = AwesomeImageLoss(criterion=UnariesWeightedLoss(SE("mean"), mode="sssdms"),
                                prior_criterion=UnariesWeightedLoss(torch.nn.BCELoss(), mode="ratio", ratio=0.5), alpha=1.4, beta=30.0, gamma=0.2)
        crit.extra_penalty = penalty
        o_h, o_c = out_c.to(dev).requires_grad_(True), out_c.clone().requires_grad_(True)
        l_h, l_c = crit(o_h, tgt_c.to(dev)), crit(o_c, tgt_c)
        l_h.backward()
        l_c.backward()
        assert float(l_h.detach()) == pytest.approx(float(l_c.detach()), rel=5e-6)
        np.testing.assert_allclose(o_h.grad.cpu().numpy(), o_c.grad.numpy(), rtol=5e-5, atol=1e-9)
    zp = np.load(os.path.join(golden_dir, "pixel_losses.npz"))
    out, tgt = torch.from_numpy(zp["al.output"]), torch.from_numpy(zp["al.target"])
    for penalty, key in ((False, "al.plain"), (True, "al.penalty")):
        crit = AwesomeLoss(alpha=0.6, scribble_percentage=0.75)
        crit.extra_penalty = penalty
        assert crit.joint_desc(out.shape[-2]) is not None
        o_h, o_c = out.to(dev).requires_grad_(True), out.clone().requires_grad_(True)
        l_h = crit(o_h, tgt.to(dev))
        l_h.backward()
        crit(o_c, tgt).backward()
        assert float(l_h.detach()) == pytest.approx(float(zp[key]), rel=3e-6)
        np.testing.assert_allclose(o_h.grad.cpu().numpy(), o_c.grad.numpy(), rtol=3e-5, atol=1e-9)


# ----------------------------------------------------------------------------------------------------------------------------
# round 4: the reference's own criterion (WeightedLoss on class labels with a noneclass), the path-connected joint steps at
# configs[4] size, one optimizer state across both step implementations, and the trainer's handling of shapes / failures
# ----------------------------------------------------------------------------------------------------------------------------


def test_weighted_loss_noneclass_in_the_device_losses(dev, golden_dir):
    """inrfit_joint_loss with InrJointLossDesc.target_rule = 1 / use_noneclass (ABI v7): FBMSJointLoss(WeightedLoss(BCELoss, sssdms,
    noneclass 2)) - the criterion of 153 reference configs - against the REFERENCE CLASS's fixture (value + both gradient channels,
    both clip branches); AwesomeImageLoss and the other modes against the host mirror's torch composition on the CPU (itself
    pinned on the same fixture, tests/test_boundary_golden.py)."""
    from awesome_amd.measures import AwesomeImageLoss, FBMSJointLoss, SE, WeightedLoss
    z = np.load(os.path.join(golden_dir, "weighted_loss_noneclass.npz"))
    out, tgt = torch.from_numpy(z["output"]), torch.from_numpy(z["target"])
    for case in range(2):
        crit = FBMSJointLoss(criterion=WeightedLoss(torch.nn.BCELoss(), mode="sssdms", noneclass=2), alpha=1.0, beta=float(z[f"fbms{case}.beta"]))
        d = crit.joint_desc()
        assert d is not None and d.target_rule == 1 and d.use_noneclass == 1 and d.noneclass == 2.0
        o = out.to(dev).requires_grad_(True)
        loss = crit(o, tgt.to(dev))
        loss.backward()
        assert float(loss.detach()) == pytest.approx(float(z[f"fbms{case}.loss"]), rel=3e-6)
        np.testing.assert_allclose(o.grad.cpu().numpy(), z[f"fbms{case}.grad"], rtol=3e-5, atol=1e-9)
    for kind, mode, nc in (("se", "equal", 2.0), ("bce", "none", 2.0), ("se", "sssdms", None)):
        inner = (lambda: torch.nn.BCELoss()) if kind == "bce" else (lambda: SE("mean"))
        t = tgt if nc is not None else torch.where(tgt == 2.0, torch.ones_like(tgt), tgt)
        for make in (lambda: FBMSJointLoss(criterion=WeightedLoss(inner(), mode=mode, noneclass=nc), alpha=0.8, beta=3.0),
                     lambda: AwesomeImageLoss(criterion=WeightedLoss(inner(), mode=mode, noneclass=nc),
                                              prior_criterion=WeightedLoss(torch.nn.BCELoss(), mode="sssdms", noneclass=nc), alpha=0.6)):
            crit = make()
            assert crit.joint_desc() is not None
            o_h, o_c = out.to(dev).requires_grad_(True), out.clone().requires_grad_(True)
            l_h, l_c = crit(o_h, t.to(dev)), make()(o_c, t)
            l_h.backward()
            l_c.backward()
            assert float(l_h.detach()) == pytest.approx(float(l_c.detach()), rel=5e-6), (kind, mode, nc)
            np.testing.assert_allclose(o_h.grad.cpu().numpy(), o_c.grad.numpy(), rtol=5e-5, atol=1e-9, err_msg=f"{kind} {mode} {nc}")


def _labels_with_noneclass(target: torch.Tensor, seed: int) -> torch.Tensor:
    """Class labels {0, 1} with ~30 % of the pixels marked unlabeled (2), like the FBMS weak labels."""
    g = torch.Generator().manual_seed(seed)
    lab = (target >= 0.5).float()
    return torch.where(torch.rand(lab.shape, generator=g) < 0.3, torch.full_like(lab, 2.0), lab)


def test_fused_joint_step_with_the_reference_configs_criterion(dev):
    """The fused step (ICNN and both path-connected priors) under FBMSJointLoss(WeightedLoss(BCELoss, sssdms, noneclass 2)) on labels
    with unlabeled pixels, against the autograd step with the same (fixture-pinned) loss: losses, backbone weights, prior rows."""
    from awesome_amd.measures import FBMSJointLoss, WeightedLoss
    from awesome_amd.model import ConvexNextNet, real_nvp_path_connected_net
    crit = lambda: FBMSJointLoss(criterion=WeightedLoss(torch.nn.BCELoss(), mode="sssdms", noneclass=2), alpha=1.0, beta=2.0)   # noqa: E731
    for factory, perturb in ((lambda: ConvexNextNet(n_hidden=64, in_features=2, n_hidden_layers=2), None),
                             (lambda: real_nvp_path_connected_net(channels=2, hidden_units=16, flow_n_flows=4, flow_output_fn="tanh",
                                                                  convex_net_hidden_units=64, convex_net_hidden_layers=2), _nonzero_last_layers)):
        runs = []
        for fused in (True, False):
            runs.append(_run_joint(dev, factory, crit, fused, 4, perturb=perturb, relabel=_labels_with_noneclass))
        (lf, wf, rf, of, tf), (la, wa, ra, oa, _) = runs
        assert int(tf.last_status[0]) == 0
        np.testing.assert_allclose(lf, la, rtol=2e-5)
        np.testing.assert_allclose(wf.numpy(), wa.numpy(), rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(rf.numpy(), ra.numpy(), rtol=1e-3, atol=2e-5)


def test_one_optimizer_state_across_fused_and_autograd_steps(dev):
    """AwesomeImageLoss with the runner's extra-penalty hook switches the trainer from the fused to the autograd step in
    the middle of a run (and `use_reduce_lr_in_extra_penalty_hook` rescales the learning rate at that moment); the reference has ONE
    optimizer throughout (torch_agent.py:812-839).  The prior's moments and step count move with the step: the mixed run must follow
    the pure autograd run - and must NOT look like a run whose prior moments restart at the switch."""
    from awesome_amd.agent import JointTrainer
    from awesome_amd.measures import AwesomeImageLoss
    from awesome_amd.model import ConvexNextNet
    from awesome_amd.prior_bank import _ordered_parameters
    factory = lambda: ConvexNextNet(n_hidden=64, in_features=2, n_hidden_layers=2)   # noqa: E731
    for opt_type in (torch.optim.Adam, torch.optim.Adamax):
        rows, paths = {}, {}
        for mode in ("mixed", "autograd", "restart"):
            items, seg, wrapper, bank = _joint_setup(dev, factory)
            crit = AwesomeImageLoss(alpha=0.7, beta=5.0, gamma=0.5)
            opt = opt_type(list(seg.parameters()) + list(_ordered_parameters(wrapper.prior_module)), lr=2e-3)
            tr = JointTrainer(wrapper, bank, crit, opt, fused=(mode != "autograd"))
            feat, taken = torch.zeros(1, 1, 1, 1, device=dev), []
            for s in range(8):
                if s == 5:
                    crit.extra_penalty = True               # awesome_runner.py:351-371
                    for g in opt.param_groups:
                        g["lr"] = g["lr"] * 0.5
                    if mode == "restart":                    # what round 3 did: the autograd step starts from empty moments
                        tr._path = None
                i = s % len(items)
                (image, _, xy), target = items[i]
                tr.perform_step(i, (image[None].to(dev), feat, xy[None].to(dev)), target[None].to(dev))
                taken.append(tr._path)
            rows[mode], paths[mode] = bank.params.detach().cpu().clone(), taken
        assert paths["mixed"] == ["fused"] * 5 + ["autograd"] * 3 and paths["autograd"] == ["autograd"] * 8
        np.testing.assert_allclose(rows["mixed"].numpy(), rows["autograd"].numpy(), rtol=1e-3, atol=3e-5)
        # the restarted run is measurably somewhere else (Adam's bias correction makes the first step after a restart ~lr * sign(g))
        assert np.abs(rows["restart"].numpy() - rows["autograd"].numpy()).max() > 20 * np.abs(rows["mixed"].numpy() - rows["autograd"].numpy()).max()
        # ... and back: the fused step continues from the autograd step's state
        items, seg, wrapper, bank = _joint_setup(dev, factory)
        crit = AwesomeImageLoss(alpha=0.7)
        crit.extra_penalty = True
        opt = opt_type(list(seg.parameters()) + list(_ordered_parameters(wrapper.prior_module)), lr=2e-3)
        tr = JointTrainer(wrapper, bank, crit, opt)
        feat = torch.zeros(1, 1, 1, 1, device=dev)
        for s in range(6):
            if s == 3:
                crit.extra_penalty = False
            (image, _, xy), target = items[s % 2]
            tr.perform_step(s % 2, (image[None].to(dev), feat, xy[None].to(dev)), target[None].to(dev))
        assert tr._path == "fused" and tr._t[None] == 6


def test_joint_trainer_takes_the_autograd_step_for_shapes_without_a_fused_kernel(dev):
    """ConvexNextNet(n_hidden=256) is a supported shape (layer-by-layer path) but has no fused joint step; under the
    default `fused=None` the trainer must plan the autograd step instead of failing in the C-ABI call."""
    from awesome_amd.agent import JointTrainer
    from awesome_amd.measures import FBMSJointLoss
    from awesome_amd.model import ConvexNextNet
    from awesome_amd.prior_bank import _ordered_parameters
    factory = lambda: ConvexNextNet(n_hidden=256, in_features=2, n_hidden_layers=1)   # noqa: E731
    items, seg, wrapper, bank = _joint_setup(dev, factory, S=32)
    opt = torch.optim.Adam(list(seg.parameters()) + list(_ordered_parameters(wrapper.prior_module)), lr=1e-3)
    tr = JointTrainer(wrapper, bank, FBMSJointLoss(alpha=1.0, beta=2.0), opt)
    assert tr.fused is False and not wrapper.prior_module.spec.fused() and wrapper.prior_module.spec.supported()
    with pytest.raises(ValueError, match="no fused joint step"):
        JointTrainer(wrapper, bank, FBMSJointLoss(), opt, fused=True)
    before = bank.params.detach().clone()
    (image, _, xy), target = items[0]
    l0, _ = tr.perform_step(0, (image[None].to(dev), torch.zeros(1, 1, 1, 1, device=dev), xy[None].to(dev)), target[None].to(dev))
    l1, _ = tr.perform_step(0, (image[None].to(dev), torch.zeros(1, 1, 1, 1, device=dev), xy[None].to(dev)), target[None].to(dev))
    assert torch.isfinite(l0) and torch.isfinite(l1) and float(l1) != float(l0) and not torch.equal(bank.params[0], before[0])


@pytest.mark.parametrize("kind", ["resnet_flow", "wide_icnn"])
def test_joint_trainer_with_composites_that_have_no_fused_form(dev, kind):
    """ConvexDiffeomorphismNet with the 'resnet' flow backbone, or with an ICNN of the layer-by-layer shapes: PriorBank rows, the
    planned autograd step (the module's composed bridges) and the optimizer all work; nothing reaches a fused entry point."""
    from awesome_amd.agent import JointTrainer
    from awesome_amd.measures import FBMSJointLoss
    from awesome_amd.model import ConvexDiffeomorphismNet
    from awesome_amd.prior_bank import _ordered_parameters
    if kind == "resnet_flow":
        factory = lambda: ConvexDiffeomorphismNet(n_hidden=32, n_hidden_layers=1, nf_layers=2, nf_hidden=8,   # noqa: E731
                                                  diffeo_args=dict(backbone="resnet", num_blocks=1))
    else:
        factory = lambda: ConvexDiffeomorphismNet(n_hidden=136, n_hidden_layers=3, nf_layers=2, nf_hidden=16,   # noqa: E731
                                                  diffeo_args=dict(backbone="normal_block"))
    items, seg, wrapper, bank = _joint_setup(dev, factory, S=32)
    opt = torch.optim.Adam(list(seg.parameters()) + list(_ordered_parameters(wrapper.prior_module)), lr=2e-4)
    tr = JointTrainer(wrapper, bank, FBMSJointLoss(alpha=1.0, beta=2.0), opt)
    assert tr.fused is False
    before = bank.params.detach().clone()
    (image, _, xy), target = items[0]
    args = (image[None].to(dev), torch.zeros(1, 1, 1, 1, device=dev), xy[None].to(dev))
    l0, _ = tr.perform_step(0, args, target[None].to(dev))
    l1, _ = tr.perform_step(0, args, target[None].to(dev))
    assert torch.isfinite(l0) and torch.isfinite(l1) and float(l1) != float(l0) and not torch.equal(bank.params[0], before[0])


def test_nonfinite_segmentation_output_freezes_row_and_backbone(dev):
    """a NaN in the segmentation output.  AWESOME_IMAGE form: the prior's own loss column does not contain `seg`, the
    COMPOSITE loss does - the row freezes (status 1) all the same; the trainer hands the backbone a zero gradient instead of NaN and
    raises the reference's ValueError when asked."""
    import awesome_amd as A
    from awesome_amd import joint as J
    from awesome_amd import _lib as L
    from awesome_amd.agent import JointTrainer
    from awesome_amd.measures import AwesomeImageLoss
    from awesome_amd.model import ConvexNextNet
    from awesome_amd.prior_bank import _ordered_parameters
    torch.manual_seed(0)
    m = ConvexNextNet(n_hidden=64, in_features=2, n_hidden_layers=1)
    row = m.flat_parameters().to(dev)
    keep, S = row.clone(), 32
    grid = A.Grid.linspace(S, S, dev)
    seg = torch.rand(S * S, device=dev) * 0.9 + 0.05
    seg[11] = float("nan")
    tgt = (torch.rand(S * S, device=dev) > 0.5).float()
    opt = torch.zeros(2 * m.spec.n_params + 8, device=dev)
    bad = J.joint_step(m.spec, row, opt, grid, seg, tgt, J.joint_desc(form=L.JOINT_AWESOME_IMAGE, weight_mode="none", alpha=0.7), step=1, lr=1e-3)
    assert int(bad.status[0]) == 1 and torch.equal(row, keep) and float(opt[: 2 * m.spec.n_params].abs().sum()) == 0.0

    class NanSeg(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = torch.nn.Conv2d(1, 1, 3, padding=1)

        def forward(self, image, *a, **k):
            out = self.conv(image)
            return out + torch.where(torch.arange(out.numel(), device=out.device).view_as(out) == 5, float("nan"), 0.0)

    factory = lambda: ConvexNextNet(n_hidden=64, in_features=2, n_hidden_layers=1)   # noqa: E731
    items, _, wrapper, bank = _joint_setup(dev, factory, S=32)
    wrapper.segmentation_module = NanSeg().to(dev)
    seg_mod = wrapper.segmentation_module
    w0 = seg_mod.conv.weight.detach().clone()
    opt = torch.optim.Adam(list(seg_mod.parameters()) + list(_ordered_parameters(wrapper.prior_module)), lr=1e-2)
    tr = JointTrainer(wrapper, bank, AwesomeImageLoss(alpha=0.7), opt, fused=True)        # check_finite="epoch": no sync per step
    before = bank.params.detach().clone()
    (image, _, xy), target = items[0]
    args = (0, (image[None].to(dev), torch.zeros(1, 1, 1, 1, device=dev), xy[None].to(dev)), target[None].to(dev))
    tr.perform_step(*args)
    assert int(tr.last_status[0]) == 1 and torch.equal(bank.params, before)
    with pytest.raises(ValueError, match="Loss is nan or inf!"):
        tr.raise_if_failed()
    # check_finite="step": the reference's behaviour to the letter - the error BEFORE backward (torch_agent.py:484-487), the backbone
    # untouched (one 4-byte device -> host read per step, which the reference pays anyway for loss.item())
    seg_mod.load_state_dict({"conv.weight": w0, "conv.bias": seg_mod.conv.bias.detach().clone().nan_to_num(0.0)})
    tr2 = JointTrainer(wrapper, bank, AwesomeImageLoss(alpha=0.7), opt, fused=True, check_finite="step")
    with pytest.raises(ValueError, match="Loss is nan or inf!"):
        tr2.perform_step(*args)
    assert torch.equal(seg_mod.conv.weight.detach(), w0) and torch.equal(bank.params, before)


@pytest.mark.parametrize("family", ["pcn", "cdn"])
def test_configs4_path_connected_joint_step_at_full_size(dev, family):
    """VERDICT r03 item 2 - BASELINE configs[4]'s named prior at ITS size: one `inrfit_pcn_joint_step` / `inrfit_cdn_joint_step` on a
    256x256 noisy pseudo-label image with the CONFIG networks (PathConnectedNet: C = 2, 12 flows x 32 hidden units, tanh outputs, ICNN
    130 x 2, `config/c5_refine_noisy256.yaml` = the reference's realnvp YAMLs; ConvexDiffeomorphismNet: 6 couplings x 130,
    normal_block, ICNN 130 x 2, `config/path-connectedness/joint/*diffeo*.yaml`) against the oracle's restatement of
    TorchAgent._perform_step (awesome/agent/torch_agent.py:428-551) with FBMSJointLoss (awesome/measures/fbms_joint_loss.py:35-59):
    forward output, loss (rel 2e-5), d loss / d seg, and EVERY prior parameter after the Adam step + enforce_convexity.
    The cdn oracle is PINNED (the reference's ConvexDiffeomorphismNet / FBMSJointLoss classes' fixtures); the pcn oracle restates
    normflows 1.7.3's published definitions: parity UNPINNED for that variant."""
    import awesome_amd as A
    from awesome_amd import flow as FL
    from awesome_amd import joint as J
    from awesome_amd import rnvp as R
    from awesome_amd.dataset import SyntheticPriorDataset
    S, lr, alpha, beta = 256, 1e-3, 1.0, 2.0
    torch.manual_seed(4)
    ds = SyntheticPriorDataset(n_images=1, size=S, kind="noisy_blob")
    (image, _, xy), target = ds[0]
    seg = (1 - torch.sigmoid(image)).reshape(-1)                                  # WrapperModule: inverted sigmoid of the backbone's logits
    rows = O.pixelize(xy[None])
    if family == "pcn":
        from tests.test_gpu_rnvp import _case, _merge, _split
        ispec, dspec, sd = _case(2, 32, 12, 2, seed=12)
        dspec = R.RnvpSpec(2, 32, 12, "tanh", None, (0.0, 0.0), (1.0, 1.0))
        masks = O.rnvp_masks(2, 12)
        fwd = lambda p: O.pcn_forward(p, rows, masks, torch.tensor(dspec.vmin), torch.tensor(dspec.vmax), output_fn="tanh", output_scale=None)  # noqa: E731
        ip, fp = _split(ispec, dspec, sd, dev)
        merge = lambda i, f: _merge(ispec, dspec, i, f)   # noqa: E731
    else:
        from awesome_amd.model import ConvexDiffeomorphismNet
        torch.manual_seed(42)
        net = ConvexDiffeomorphismNet(n_hidden=130, n_hidden_layers=2, nf_layers=6, nf_hidden=130, diffeo_args=dict(backbone="normal_block"))
        sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
        ispec, dspec = A.IcnnSpec(130, 2, 2), FL.FlowSpec(130, 6)
        fwd = lambda p: O.convex_diffeo_forward(p, rows, 6)   # noqa: E731
        i0, f0 = FL.split_cdn_state_dict(ispec, dspec, sd, dev)
        ip, fp = i0[None].contiguous(), f0[None].contiguous()
        merge = lambda i, f: FL.merge_cdn_state_dict(ispec, dspec, i, f)   # noqa: E731
    clamp_keys = [k for k in sd if k.endswith("ln.weight")]      # enforce_convexity: hidden->hidden and hidden->out weights (convex_net.py:151-154, 216-220)
    assert len(clamp_keys) == 3
    # ---- the oracle's step: one Adam over the prior's parameters (the backbone's share is checked through d loss / d seg)
    params = {k: torch.nn.Parameter(v.clone()) for k, v in sd.items() if v.is_floating_point() and not k.endswith("data_dep_init_done")}
    full = dict(sd)
    full.update(params)
    seg_ref = seg.clone().requires_grad_(True)
    opt = torch.optim.Adam(list(params.values()), lr=lr)
    y = fwd(full)
    out_ref = torch.cat([seg_ref.reshape(1, 1, S, S), torch.sigmoid(O.unpixelize(y, 1, S, S))], dim=1)
    loss_ref = O.fbms_joint_loss(out_ref, target[None], alpha=alpha, beta=beta, kind="bce", mode="sssdms")
    loss_ref.backward()
    opt.step()
    with torch.no_grad():
        for k in clamp_keys:
            params[k].clamp_(min=0)
    # ---- the device's step
    grid = A.Grid.explicit(xy.reshape(2, -1).to(dev).contiguous())
    iopt = torch.zeros(2 * ispec.n_params + 8, device=dev)
    fopt = torch.zeros(2 * dspec.n_params, device=dev)
    desc = J.joint_desc(kind="bce", weight_mode="sssdms", alpha=alpha, beta=beta)
    step = J.pcn_joint_step if family == "pcn" else J.cdn_joint_step
    res = step(ispec, dspec, ip[0], fp[0], iopt, fopt, grid, seg.to(dev).contiguous(), target.reshape(-1).to(dev).contiguous(), desc, step=1, lr=lr)
    assert int(res.status[0]) == 0
    np.testing.assert_allclose(res.prior_logits.cpu().numpy(), y.detach().reshape(-1).numpy(), atol=3e-4, rtol=1e-3)
    assert float(res.loss[0]) == pytest.approx(float(loss_ref.detach()), rel=2e-5)
    g_ref = seg_ref.grad.numpy()
    np.testing.assert_allclose(res.dseg.cpu().numpy(), g_ref, rtol=2e-4, atol=2e-6 * float(np.abs(g_ref).max()))
    got = merge(ip[0].cpu(), fp[0].cpu())
    worst, n_moved, n_loose = 0.0, 0, 0
    gmax = max(float(p.grad.abs().max()) for p in params.values())
    for k, p in params.items():
        ref, new, g = p.detach().numpy(), got[k].numpy().reshape(p.shape), p.grad.numpy()
        err = np.abs(new - ref)
        worst = max(worst, float(err.max()))
        n_moved += int((np.abs(ref - sd[k].numpy()) > 0).sum())
        # Adam's FIRST step moves every parameter by lr * g / (|g| + eps): by lr in the gradient's direction whatever the gradient's
        # size.  Where the gradient is above rounding level the two implementations agree to 2 % of that step; an entry whose
        # gradient is at rounding level (|g| < 1e-4 of the model's largest; e.g. the 1x1 weight_v of a WNScale, whose gradient is
        # exactly zero in exact arithmetic) may come out anywhere inside +-lr on either side
        assert err.max() <= 2.0 * lr + 1e-7, k
        loose = err > 0.02 * lr
        n_loose += int(loose.sum())
        assert np.all(np.abs(g[loose]) <= 1e-4 * gmax), (k, np.abs(g[loose]).max() / gmax, int(loose.sum()))
    n_total = sum(p.numel() for p in params.values())
    # (masked input columns of the RealNVP MLPs have an exactly zero gradient: Adam leaves them where they are, here and there)
    assert n_moved > 0.5 * n_total and n_loose < 0.002 * n_total, (n_moved, n_loose, n_total)
    print(f"\nconfigs[4] {family} joint step at 256x256: loss {float(res.loss[0]):.6f} vs {float(loss_ref.detach()):.6f}; max |d param| {worst:.2e} (lr {lr}), "
          f"{n_loose} of {n_total} entries with rounding-level gradients outside 2 % of the step")
