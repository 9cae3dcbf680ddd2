"""Host-side logic that needs no GPU: parameter packing, drop-in module surface, loss objects, synthetic data, sharding."""
import os

import numpy as np
import pytest
import torch

import awesome_amd as A
from awesome_amd import parallel
from awesome_amd.dataset import SyntheticUnariesDataset, convex_blob_mask
from awesome_amd.measures import MIOU, SE, AwesomeImageLoss, UnariesWeightedLoss, criterion_to_desc
from awesome_amd.model import ConvexNet, ConvexNextNet
from oracle import inr_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_pack_unpack_roundtrip_and_layout():
    spec = A.IcnnSpec(130, 2, 1)
    assert spec.n_params == 17813
    torch.manual_seed(0)
    sd = {k: torch.randn(shp) for k, shp in spec.keys_shapes()}
    flat = A.pack_state_dict(spec, sd)
    assert flat.shape == (17813,)
    back = A.unpack_params(spec, flat)
    assert all(torch.equal(back[k], sd[k]) for k in sd)
    assert float(flat[0]) == float(sd["input.weight"][0, 0]) and float(flat[-1]) == float(sd["out.skp.weight"][0, -1])
    # ConvexNet key names map onto the same vector
    sd_cn = {A.icnn.CONVEXNET_KEYMAP_INV[k]: v for k, v in sd.items()}
    assert torch.equal(A.pack_state_dict(spec, sd_cn), flat)


def test_dropin_modules_match_reference_init_and_keys(golden_dir):
    """Seeded construction reproduces the reference module's initial weights bit for bit (fixture: fit_disc64 sd0)."""
    z = _load(golden_dir, "fit_disc64.npz")
    torch.manual_seed(0)
    m = ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=1)
    sd = m.state_dict()
    ref = O.load_npz_state(z, "sd0.")
    assert list(sd.keys()) == list(ref.keys())
    for k in ref:
        assert torch.equal(sd[k], ref[k]), k
    z = _load(golden_dir, "icnn_convexnet_h130_c2.npz")
    assert list(ConvexNet().state_dict().keys()) == list(O.load_npz_state(z, "sd0.").keys())
    # enforce_convexity clamps exactly the reference's weights
    with torch.no_grad():
        for p in m.parameters():
            p.fill_(-1.0)
    m.enforce_convexity()
    sd = m.state_dict()
    assert float(sd["skip.0.ln.weight"].min()) == 0.0 and float(sd["out.ln.weight"].min()) == 0.0
    assert float(sd["skip.0.skp.weight"].max()) == -1.0 and float(sd["input.weight"].max()) == -1.0
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 2, 4, 4))  # CPU tensors are refused: there is no CPU fallback


def test_loss_objects_match_reference(golden_dir):
    z = _load(golden_dir, "losses.npz")
    out, tgt = torch.from_numpy(z["output"]), torch.from_numpy(z["target"])
    for mode in ["none", "equal", "ratio", "sssdms"]:
        for kind, crit in [("se", SE("mean")), ("bce", torch.nn.BCELoss())]:
            l = UnariesWeightedLoss(crit, mode=mode, ratio=0.35)
            assert float(l(out, tgt)) == pytest.approx(float(z[f"uwl.{kind}.{mode}"]), rel=1e-6)
            assert criterion_to_desc(l) == (kind, mode, 0.35)
    out2, tb = torch.from_numpy(z["output2"]), torch.from_numpy(z["target_bin"])
    ail = AwesomeImageLoss(alpha=0.7, beta=100.0, gamma=0.1)
    assert float(ail(out2, tb)) == pytest.approx(float(z["ail.plain"]), rel=1e-6)
    ail.extra_penalty = True
    assert float(ail(out2, tb)) == pytest.approx(float(z["ail.penalty"]), rel=1e-6)
    with pytest.raises(TypeError):
        criterion_to_desc(ail)


def test_miou_object_cpu_matches_reference(golden_dir):
    z = _load(golden_dir, "miou.npz")
    m = MIOU(invert=True)
    for i in range(int(z["n"])):
        assert float(m(torch.from_numpy(z[f"o{i}"]), torch.from_numpy(z[f"t{i}"]))) == pytest.approx(float(z[f"iou{i}"]), abs=1e-7)


def test_synthetic_blob_is_convex_and_deterministic():
    a, b = convex_blob_mask(256, 3), convex_blob_mask(256, 3)
    assert np.array_equal(a, b) and 2000 < a.sum() < 30000
    ys, xs = np.nonzero(a)   # convex: every row/column run of the mask is contiguous
    for r in np.unique(ys):
        cols = xs[ys == r]
        assert cols.max() - cols.min() + 1 == len(cols)
    ds = SyntheticUnariesDataset(n_images=3, size=64, kind="blob")
    assert ds.batch([0, 1, 2]).shape == (3, 64 * 64)


def test_grid_descriptors_match_reference_grid(golden_dir):
    z = _load(golden_dir, "grid.npz")
    g = A.Grid.linspace(7, 5, "cpu")
    assert np.array_equal(g.xs.numpy(), z["g_7x5"][0, 0]) and np.array_equal(g.ys.numpy(), z["g_7x5"][1, :, 0])
    g = A.Grid.howto(8, 4, "cpu")
    ref = O.howto_grid(4, 8)
    assert torch.equal(g.xs, ref[0, 0, 0]) and torch.equal(g.ys, ref[0, 1, :, 0])
    e = A.Grid.from_image_grid(torch.zeros(2, 3, 4, 5))
    assert e.n_points == 20 and e.image_stride == 60


def test_sharding_rules():
    assert [list(parallel.shard_range(10, r, 4)) for r in range(4)] == [[0, 1, 2], [3, 4, 5], [6, 7], [8, 9]]
    assert sum(len(parallel.shard_range(512, r, 8)) for r in range(8)) == 512
    owners = [parallel.shard_sequences([30, 5, 12, 12, 7], r, 2) for r in range(2)]
    assert sorted(owners[0] + owners[1]) == [0, 1, 2, 3, 4] and not set(owners[0]) & set(owners[1])


# ---- RealNVP / PathConnectedNet host logic (row a10; parity unpinned - see tests/test_gpu_rnvp.py) ---------------------
def test_rnvp_spec_layout_and_masks():
    import ctypes
    from awesome_amd import _lib as L
    from awesome_amd import rnvp as R
    for C, hid, F in ((2, 32, 12), (3, 32, 18), (3, 20, 5), (2, 64, 3), (2, 130, 6), (3, 130, 18)):
        spec = R.RnvpSpec(C, hid, F)
        assert spec.n_params == sum(int(np.prod(s)) for _, s in spec.keys_shapes())
        d = spec.desc()
        assert L.load().inrfit_rnvp_param_count(ctypes.byref(d)) == spec.n_params   # host-only entry point
        ref = O.rnvp_masks(C, F)   # net_factory.py:86-99
        assert [int(sum(int(ref[f, c]) << c for c in range(C))) for f in range(F)] == list(spec.masks)
        flat = torch.arange(spec.n_params, dtype=torch.float32)
        sd = R.unpack_rnvp_params(spec, flat)
        assert torch.equal(R.pack_rnvp_state_dict(spec, sd), flat)
        for (a, b) in spec.actnorm_slices():
            assert b - a == 2 * C
    bad = R.RnvpSpec(2, 300, 12).desc()   # hidden_units > 256: not built
    assert L.load().inrfit_rnvp_param_count(ctypes.byref(bad)) < 0


def test_path_connected_net_module_surface():
    from awesome_amd.model import PathConnectedNet, real_nvp_path_connected_net
    torch.manual_seed(0)
    m = real_nvp_path_connected_net(channels=3, hidden_units=32, flow_n_flows=18, flow_output_fn="tanh")
    assert isinstance(m, PathConnectedNet)
    sd = m.state_dict()
    ispec, rspec = m._specs()
    assert (rspec.channels, rspec.hidden_units, rspec.n_flows, rspec.output_fn) == (3, 32, 18, "tanh")
    assert rspec.vmin == (0.0, 0.0, 0.0) and rspec.vmax == (1.0, 1.0, 1.0) and (rspec.new_min, rspec.new_max) == (-1.0, 1.0)
    for k, shp in rspec.keys_shapes():
        assert tuple(sd[k].shape) == shp, k
    # init_zeros=True (net_factory.py:104-105): the last layer of every MLP starts at zero, ActNorm at s = t = 0
    for f in range(18):
        for net in "st":
            assert float(sd[f"flow_net.net.network.flows.{2 * f}.{net}.net.2.weight"].abs().max()) == 0.0
        assert float(sd[f"flow_net.net.network.flows.{2 * f + 1}.s"].abs().max()) == 0.0
    assert torch.equal(sd["linear.weight"], torch.ones(3, 1, 1, 1)) and torch.equal(sd["linear.bias"], torch.zeros(3))
    assert sd["flow_net.net.network.flows.4.b"].tolist() == [[1, 1, 0]]   # mask value 3
    with pytest.raises(RuntimeError):   # no CPU path
        m(torch.zeros(1, 3, 4, 4))
    m.reset_parameters()                # TensorUtil.reset_parameters: every nn.Linear is re-drawn, ActNorm untouched
    assert float(m.state_dict()["flow_net.net.network.flows.0.s.net.2.weight"].abs().max()) > 0.0


def test_oracle_rnvp_restatement_properties():
    """The RealNVP restatement has no golden vectors (normflows is absent); these are the properties its definition implies."""
    torch.manual_seed(0)
    C, F, hid = 3, 6, 8
    masks = O.rnvp_masks(C, F)
    assert masks.tolist() == [[1, 0, 0], [0, 1, 0], [1, 1, 0], [0, 0, 1], [1, 0, 1], [0, 1, 1]]
    sd = {"linear.weight": torch.ones(C, 1, 1, 1), "linear.bias": torch.zeros(C)}
    for f in range(F):
        for n in "st":
            b = f"flow_net.net.network.flows.{2 * f}.{n}.net."
            sd[b + "0.weight"], sd[b + "0.bias"] = torch.randn(hid, C), torch.randn(hid)
            sd[b + "2.weight"], sd[b + "2.bias"] = torch.zeros(C, hid), torch.zeros(C)
        sd[f"flow_net.net.network.flows.{2 * f + 1}.s"] = torch.zeros(1, C)
        sd[f"flow_net.net.network.flows.{2 * f + 1}.t"] = torch.zeros(1, C)
    x = torch.rand(500, C)
    vmin, vmax = torch.zeros(C), torch.ones(C)
    # init_zeros: every coupling is the identity, ActNorm s = t = 0: the deformation is the identity
    np.testing.assert_allclose(O.pcn_deformation(sd, x, masks, vmin, vmax).numpy(), x.numpy(), atol=1e-6)
    # data-dependent init: the first ActNorm whitens its input, the following ones see whitened data
    O.pcn_deformation(sd, x, masks, vmin, vmax, actnorm_init=True)
    z0 = O.minmax(x, vmin, vmax, -1.0, 1.0)
    np.testing.assert_allclose(sd["flow_net.net.network.flows.1.s"].numpy(), -torch.log(z0.std(0, keepdim=True) + 1e-6).numpy(), rtol=1e-5)
    assert float(sd["flow_net.net.network.flows.3.s"].abs().max()) < 1e-4
    # a masked channel passes through its coupling unchanged
    for n in "st":
        sd[f"flow_net.net.network.flows.0.{n}.net.2.weight"] = torch.randn(C, hid) * 0.3
    z = torch.randn(50, C)
    z1 = O.rnvp_flow_forward(sd, z, masks[:1], actnorm_init=False)
    ea, at = torch.exp(sd["flow_net.net.network.flows.1.s"]), sd["flow_net.net.network.flows.1.t"]
    np.testing.assert_allclose(z1[:, 0].numpy(), (z[:, 0] * ea[0, 0] + at[0, 0]).numpy(), rtol=1e-6, atol=1e-6)
    assert float((z1[:, 1] - (z[:, 1] * ea[0, 1] + at[0, 1])).abs().max()) > 1e-3


def test_prior_bank_swap_is_an_index():
    """PriorBank: the PriorManager / PriorCache pair of the joint-training path with all priors in one tensor.
    Host logic only (CPU tensors): binding re-points the parameters at a row, a step inside the manager is stored without a
    copy, keys get fresh initialisations lazily, and the export has PriorCache.get_state()'s layout."""
    from awesome_amd.model import ConvexNextNet
    from awesome_amd.prior_bank import PriorBank
    torch.manual_seed(0)
    factory = lambda: ConvexNextNet(n_hidden=32, in_features=2, n_hidden_layers=1)
    bank = PriorBank(factory, n_images=3, device="cpu")
    model = factory()
    opt = torch.optim.SGD(model.parameters(), lr=0.5)
    ref_keys = list(model.state_dict().keys())
    assert 7 not in bank and len(bank) == 3
    with bank.manager(model, 7):
        assert 7 in bank
        before = bank.row(7).clone()
        for p in model.parameters():
            assert p.data_ptr() >= bank.params.data_ptr() and p.data_ptr() < bank.params.data_ptr() + bank.params.numel() * 4
        loss = sum((p ** 2).sum() for p in model.parameters())     # any differentiable function of the parameters
        opt.zero_grad(); loss.backward(); opt.step()
    torch.testing.assert_close(bank.row(7), before * (1 - 0.5 * 2))      # p <- p - lr * 2p, written straight into the bank
    with bank.manager(model, "b"):                                        # another key: a fresh prior, row 7 untouched
        assert not torch.equal(bank.row("b"), bank.row(7))
        model.enforce_convexity()
        assert float(model.state_dict()["skip.0.ln.weight"].min()) >= 0.0
    torch.testing.assert_close(bank.row(7), before * 0.0)
    with bank.manager(model, 7):                                          # back to key 7: the model sees the stored step
        flat = torch.cat([p.detach().reshape(-1) for p in model._ordered_params()])
        assert torch.equal(flat, bank.row(7))
    st = bank.get_state(model_args={"n_hidden": 32})
    assert set(st) == {"model_type", "model_args", "store_device", "cache"} and set(st["cache"]) == {"7", "b"}
    assert list(st["cache"]["b"].keys()) == ref_keys
    for k, v in st["cache"]["b"].items():
        assert v.shape == model.state_dict()[k].shape
    bank.index_of("c")
    with pytest.raises(KeyError):
        bank.index_of("d")


def test_fbms_joint_loss_clip_without_host_sync():
    """FBMSJointLoss (awesome/measures/fbms_joint_loss.py:35-59): alpha crit(seg, t) + clipped beta SE(prior, seg).  The clip is
    a device-side select here (the reference branches on the host); values and gradients must be those of the reference formula
    in both regimes."""
    from awesome_amd.measures.losses import FBMSJointLoss, SE
    torch.manual_seed(1)
    for beta in (0.05, 50.0):                                   # penalty below / above the segmentation loss
        out = torch.rand(2, 2, 8, 8, requires_grad=True)
        tgt = (torch.rand(2, 1, 8, 8) > 0.5).float()
        crit = FBMSJointLoss(criterion=SE("mean"), penalty_criterion=SE("mean"), alpha=1.0, beta=beta)
        loss = crit(out, tgt)
        g, = torch.autograd.grad(loss, out)
        o2 = out.detach().clone().requires_grad_(True)
        seg, pri = o2[:, :1], o2[:, 1:]
        sl = ((tgt - seg) ** 2).mean()
        pl = beta * ((seg - pri) ** 2).mean()
        clipped = bool(pl > sl)
        if clipped:
            pl = pl * (sl / pl).detach()
        ref = sl + pl
        g2, = torch.autograd.grad(ref, o2)
        assert clipped == (beta > 1.0)
        torch.testing.assert_close(loss, ref)
        torch.testing.assert_close(g, g2)


def test_zoo_cache_round_trip(tmp_path):
    """awesome_amd.model.Zoo (the role of awesome/model/zoo.py): keyed by name + repr(model) + config, not by the weights; a hit
    loads the stored state and returns the context; entries survive on disk."""
    from awesome_amd.model.zoo import Zoo, tensor_hash
    torch.manual_seed(0)
    make = lambda: torch.nn.Sequential(torch.nn.Linear(3, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))
    x = torch.rand(7, 3)
    cfg = dict(lr=1e-2, max_iter=100, x_data=tensor_hash(x), criterion=torch.nn.MSELoss())
    zoo = Zoo(str(tmp_path))
    a = make()
    assert zoo.load_model_state("flow_identity", a, config=cfg) == (False, None)
    zoo.save_model_state("flow_identity", a, config=cfg, context=dict(loss_hist=torch.arange(3.0)))
    b = make()                                                   # other weights, same architecture: a hit overrides them
    assert not torch.equal(b[0].weight, a[0].weight)
    ok, ctx = Zoo(str(tmp_path)).load_model_state("flow_identity", b, config=cfg)      # a new instance reads the file
    assert ok and torch.equal(ctx["loss_hist"], torch.arange(3.0))
    for k, v in a.state_dict().items():
        assert torch.equal(b.state_dict()[k], v)
    assert zoo.load_model_state("flow_identity", make(), config=dict(cfg, lr=2e-2))[0] is False        # other hyper-parameters
    assert zoo.load_model_state("flow_identity", make(), config=dict(cfg, x_data=tensor_hash(x + 1)))[0] is False   # other grid
    assert zoo.load_model_state("other_name", make(), config=cfg)[0] is False
    wide = torch.nn.Sequential(torch.nn.Linear(3, 6), torch.nn.Tanh(), torch.nn.Linear(6, 2))
    assert zoo.load_model_state("flow_identity", wide, config=cfg)[0] is False                        # other architecture
    mem = Zoo(None)                                                                                   # memory only
    mem.save_model_state("n", a, config=cfg)
    assert mem.load_model_state("n", make(), config=cfg)[0] is True


def test_loss_contract_additional_arguments_rule():
    """TorchAgent.forward_additional_loss_args (awesome/agent/torch_agent.py:150-164): `_input=device_inputs` goes to losses whose
    call signature names `_input` or `kwargs` - for an nn.Module it is the signature of `forward` that counts."""
    import torch
    from awesome_amd.agent import _loss_takes_input
    from awesome_amd.measures import SE, AwesomeImageLoss, FBMSJointLoss

    class WithInput(torch.nn.Module):
        def forward(self, output, target, _input=None):
            return output.sum()

    class Plain:
        def __call__(self, output, target):
            return output.sum()

    assert _loss_takes_input(AwesomeImageLoss()) and _loss_takes_input(FBMSJointLoss()) and _loss_takes_input(SE())
    assert _loss_takes_input(WithInput())
    assert not _loss_takes_input(torch.nn.BCELoss()) and not _loss_takes_input(Plain())


def test_synthetic_prior_dataset_follows_the_dataset_contract():
    """SURVEY §8(b) dataset contract: items `((key, state), ((image, features, xy), target))` under `@prior()`, `split_indices`,
    `get_config`, `training_batch_size`, `returns_index`, `decode_encoding` (>= 0.5, awesome_dataset.py:393-412), `__prior_cache__`."""
    import numpy as np
    import torch
    from awesome_amd.dataset import SyntheticPriorDataset

    class Tiny(torch.nn.Module):
        def __init__(self, n=3):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(n))

    ds = SyntheticPriorDataset(n_images=3, size=16, kind="blob", prior_model_type=Tiny, prior_model_args=dict(n=3))
    (key, state), ((image, feat, xy), target) = ds[1]
    assert key == 1 and set(state) == {"w"} and image.shape == (1, 16, 16) and xy.shape == (2, 16, 16) and target.shape == (1, 16, 16)
    train, val = ds.split_indices()
    assert list(train) == [0, 1, 2] and len(val) == 0
    cfg = ds.get_config()
    assert cfg["n_images"] == 3 and cfg["size"] == 16 and cfg["kind"] == "blob"
    assert ds.training_batch_size == 1 and ds.returns_index is False and ds.has_prior and 1 in ds.__prior_cache__.__cache__
    out = torch.tensor([[0.49999, 0.5], [0.7, 0.1]])
    np.testing.assert_array_equal(ds.decode_encoding(out).numpy(), np.array([[0.0, 1.0], [1.0, 0.0]], np.float32))
    ds.return_prior = False
    (image2, _, _), _ = ds[1]
    assert torch.equal(image2, image)


def test_unaries_conversion_loss_and_targets():
    """awesome/measures/unaries_conversion_loss.py:25-27: the criterion on (target >= 0.5); the fused fits get the binarised targets."""
    from awesome_amd.measures import SE, UnariesConversionLoss, UnariesWeightedLoss, criterion_targets, criterion_to_desc
    out, tgt = torch.rand(2, 1, 5, 7), torch.rand(2, 1, 5, 7)
    tgt[0, 0, 0, 0] = 0.5
    crit = UnariesConversionLoss(SE("mean"))
    assert float(crit(out, tgt)) == pytest.approx(float(((out - (tgt >= 0.5).float()) ** 2).mean()))
    assert crit.get_name() == "UCMSE"
    # a caller that binarises its targets (the per-image fits) asks for the inner criterion's form; any other caller is refused
    # (the composite losses' kernels read the targets as they are, so they take the torch composition instead)
    assert criterion_to_desc(crit, "targets") == ("se", "none", 1.0)
    assert criterion_to_desc(UnariesConversionLoss(UnariesWeightedLoss(torch.nn.BCELoss(), mode="sssdms")), "targets") == ("bce", "sssdms", 1.0)
    with pytest.raises(TypeError):
        criterion_to_desc(crit)
    from awesome_amd.measures import AwesomeImageLoss, AwesomeLoss, FBMSJointLoss
    assert FBMSJointLoss(criterion=crit).joint_desc() is None and AwesomeImageLoss(criterion=crit).joint_desc() is None
    assert AwesomeLoss(criterion=crit).joint_desc(10) is None and FBMSJointLoss().joint_desc() is not None
    soft_out = torch.rand(1, 2, 5, 7)
    want = SE("mean")(soft_out[:, :1], (tgt[:1] >= 0.5).float()) + SE("mean")(soft_out[:, 1:], soft_out[:, :1])
    assert float(FBMSJointLoss(criterion=crit, clip_penalty=False)(soft_out, tgt[:1])) == pytest.approx(float(want), rel=1e-6)
    assert torch.equal(criterion_targets(crit, tgt), (tgt >= 0.5).float())
    assert criterion_targets(SE("mean"), tgt) is tgt
