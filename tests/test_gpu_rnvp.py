"""GPU parity of the RealNVP / PathConnectedNet path (awesome_amd/csrc/rnvp.h through the C ABI) against the oracle's
restatement of normflows' MaskedAffineFlow / ActNorm / MLP (oracle/inr_oracle.py, section a10).

PARITY UNPINNED for this variant: normflows==1.7.3 is not part of the reference checkout, so the oracle here is pinned
only on the reference's call sites (net_factory.py:70-114), not on outputs of the real package."""
import numpy as np
import pytest
import torch

from oracle import inr_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import awesome_amd._lib as L
    L.load()
    return torch.device("cuda:0")


def _case(C, hid, F, layers, seed=0, output_fn="tanh", output_scale=None, h=130):
    """Random PathConnectedNet state (every parameter non-trivial, incl. the zero-initialised last layers)."""
    import awesome_amd as A
    from awesome_amd import rnvp as R
    torch.manual_seed(seed)
    vmin = tuple(float(v) for v in (-0.1, 0.0, 0.05)[:C])
    vmax = tuple(float(v) for v in (1.2, 1.0, 0.9)[:C])
    rspec = R.RnvpSpec(C, hid, F, output_fn, output_scale, vmin, vmax)
    ispec = A.IcnnSpec(h, C, layers)
    sd = {}
    for k, shp in rspec.keys_shapes():
        if k == "linear.weight":
            sd[k] = 1.0 + 0.2 * torch.randn(shp)
        elif k == "linear.bias":
            sd[k] = 0.1 * torch.randn(shp)
        elif k.endswith("net.0.weight"):
            sd[k] = torch.randn(shp) * 0.8
        elif k.endswith("net.0.bias"):
            sd[k] = torch.randn(shp) * 0.5
        elif k.endswith("net.2.weight"):
            # same output scale for every width: 18 flows x 130 units with O(1) couplings are chaotic (the fp32 torch oracle
            # itself then loses 2 digits against fp64, and single relus switch between implementations)
            sd[k] = torch.randn(shp) * 0.15 * (32.0 / hid) ** 0.5
        elif k.endswith("net.2.bias"):
            sd[k] = torch.randn(shp) * 0.1
        else:   # ActNorm s, t
            sd[k] = torch.randn(shp) * 0.1
    from awesome_amd.model import ConvexNextNet
    m = ConvexNextNet(n_hidden=h, n_hidden_layers=layers, in_features=C)
    for k, v in m.state_dict().items():
        sd["convex_net." + k] = v.detach().clone()
    return ispec, rspec, sd


def _split(ispec, rspec, sd, dev):
    import awesome_amd as A
    from awesome_amd import rnvp as R
    ip = A.pack_state_dict(ispec, {k[len("convex_net."):]: v for k, v in sd.items() if k.startswith("convex_net.")}, dev)
    fp = R.pack_rnvp_state_dict(rspec, sd, dev)
    return ip[None].contiguous(), fp[None].contiguous()


def _merge(ispec, rspec, gi, gf):
    import awesome_amd as A
    from awesome_amd import rnvp as R
    out = {"convex_net." + k: v for k, v in A.unpack_params(ispec, gi).items()}
    out.update(R.unpack_rnvp_params(rspec, gf))
    return out


def _rows(C, H, W, t=0.37):
    g = O.positional_grid(W, H) if C == 2 else O.positional_grid(W, H, t, 1.0)
    return g, O.pixelize(g[None])


def test_masks_match_reference_factory():
    from awesome_amd import rnvp as R
    for C, F in ((2, 12), (3, 18), (3, 7), (2, 5)):
        ref = O.rnvp_masks(C, F)
        got = R.rnvp_masks(C, F)
        assert [int(sum(int(ref[f, c]) << c for c in range(C))) for f in range(F)] == got


@pytest.fixture
def rnvp_shape(request):
    """Lanes per point of the RealNVP point kernels (csrc/inrfit.hip carve_pcn; C = 2 only): None = what the library picks for the
    launch's size (small launches: forward 2, backward 1), "1" / "2" / "4" force both kernels."""
    import os
    old = os.environ.pop("INR_RNVP_SHAPE", None)
    if request.param is not None:
        os.environ["INR_RNVP_SHAPE"] = request.param
    yield request.param
    os.environ.pop("INR_RNVP_SHAPE", None)
    if old is not None:
        os.environ["INR_RNVP_SHAPE"] = old


@pytest.mark.parametrize("rnvp_shape", [None, "1", "2", "4"], indirect=True)
@pytest.mark.parametrize("C,hid,F,layers,fn,scale", [(2, 32, 12, 2, "tanh", None), (3, 32, 18, 2, "tanh", None),
                                                      (3, 20, 5, 1, None, None), (2, 64, 3, 1, "tanh", 0.5),
                                                      (2, 130, 6, 2, "tanh", None),     # the factory defaults for the sizes
                                                      (3, 130, 18, 1, "tanh", None),    # 76 KB of flow records in LDS
                                                      (3, 70, 4, 1, "tanh", None)])
def test_forward_and_gradients(dev, C, hid, F, layers, fn, scale, rnvp_shape):
    from awesome_amd import rnvp as R
    import awesome_amd as A
    ispec, rspec, sd = _case(C, hid, F, layers, seed=C * 7 + F, output_fn=fn, output_scale=scale)
    assert rspec.n_params == sum(int(np.prod(s)) for _, s in rspec.keys_shapes())
    H, W = 11, 19   # N = 209: ragged
    grid_t, rows = _rows(C, H, W)
    un = torch.rand(H * W, 1)
    masks = O.rnvp_masks(C, F)
    vmin, vmax = torch.tensor(rspec.vmin), torch.tensor(rspec.vmax)
    kw = dict(output_fn=fn, output_scale=scale)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xd_ref = O.pcn_deformation(sdo, rows, masks, vmin, vmax, **kw)
    yo = O.pcn_forward(sdo, rows, masks, vmin, vmax, **kw)
    lo = O.weighted_loss(torch.sigmoid(yo).reshape(1, 1, -1, 1), un.reshape(1, 1, -1, 1), "se", "sssdms")
    lo.backward()
    ip, fp = _split(ispec, rspec, sd, dev)
    grid = A.Grid.from_image_grid(grid_t.to(dev))
    xd = R.rnvp_forward(rspec, fp, grid)
    np.testing.assert_allclose(xd[0].cpu().numpy(), xd_ref.detach().t().numpy(), rtol=3e-5, atol=5e-6)
    y = R.pcn_forward(ispec, rspec, ip, fp, grid)
    np.testing.assert_allclose(y[0].cpu().numpy(), yo.detach().reshape(-1).numpy(), rtol=1e-4, atol=5e-5)
    loss, gi, gf = R.pcn_loss_grad(ispec, rspec, ip, fp, grid, un.reshape(1, -1).to(dev), loss="se", weight_mode="sssdms")
    assert float(loss[0]) == pytest.approx(float(lo.detach()), rel=3e-5)
    got = _merge(ispec, rspec, gi[0].cpu(), gf[0].cpu())
    for k in sdo:
        ref = sdo[k].grad.numpy()
        scale_ = float(np.abs(ref).max())
        np.testing.assert_allclose(got[k].numpy(), ref, rtol=1e-3, atol=3e-5 * scale_ + 1e-7, err_msg=k)


def test_separable_grid_with_time_channel(dev):
    """C = 3 through the separable grid (xs, ys, ts per image), two images with different t and parameters."""
    from awesome_amd import rnvp as R
    import awesome_amd as A
    C, hid, F = 3, 32, 6
    H, W = 8, 16
    cases = [_case(C, hid, F, 1, seed=s) for s in (1, 2)]
    ispec, rspec = cases[0][0], cases[0][1]
    ts = torch.tensor([0.25, 0.8])
    ips, fps = zip(*[_split(ispec, rspec, c[2], dev) for c in cases])
    ip, fp = torch.cat(ips), torch.cat(fps)
    grid = A.Grid.linspace(W, H, dev, ts.to(dev))
    y = R.pcn_forward(ispec, rspec, ip, fp, grid)
    masks = O.rnvp_masks(C, F)
    for i, c in enumerate(cases):
        _, rows = _rows(C, H, W, float(ts[i]))
        yo = O.pcn_forward(c[2], rows, masks, torch.tensor(rspec.vmin), torch.tensor(rspec.vmax))
        np.testing.assert_allclose(y[i].cpu().numpy(), yo.reshape(-1).numpy(), rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("C,F,H,W", [(2, 12, 24, 20), (3, 6, 24, 20), (2, 12, 136, 128), (3, 18, 136, 128)])
def test_actnorm_data_dependent_init(dev, C, F, H, W):
    """One block per image below 16 384 points, 2 F + 1 launches over all points above (same statistics, other summation order)."""
    from awesome_amd import rnvp as R
    import awesome_amd as A
    ispec, rspec, sd = _case(C, 32, F, 1, seed=5)
    grid_t, rows = _rows(C, H, W)
    masks = O.rnvp_masks(C, F)
    sdo = {k: v.clone() for k, v in sd.items()}
    O.pcn_deformation(sdo, rows, masks, torch.tensor(rspec.vmin), torch.tensor(rspec.vmax), actnorm_init=True)
    _, fp = _split(ispec, rspec, sd, dev)
    R.actnorm_init(rspec, fp, A.Grid.from_image_grid(grid_t.to(dev)))
    got = R.unpack_rnvp_params(rspec, fp[0].cpu())
    for f in range(F):
        for nm in ("s", "t"):
            k = f"flow_net.net.network.flows.{2 * f + 1}.{nm}"
            np.testing.assert_allclose(got[k].numpy(), sdo[k].numpy(), rtol=2e-4, atol=2e-5, err_msg=k)
    # everything else untouched
    for k in got:
        if ".flows." in k and k.rsplit(".", 1)[1] in ("s", "t") and int(k.split(".flows.")[1].split(".")[0]) % 2 == 1:
            continue
        np.testing.assert_array_equal(got[k].numpy(), sd[k].numpy(), err_msg=k)


@pytest.mark.parametrize("C,optimizer", [(2, "adamax"), (3, "adamax"), (2, "adam")])
def test_fit_trajectory(dev, C, optimizer):
    """15 steps of the fused PathConnectedNet fit (param groups with flow weight decay, plateau, clamp) vs the oracle loop."""
    from awesome_amd import rnvp as R
    import awesome_amd as A
    F, hid, layers = (12, 32, 2) if C == 2 else (6, 32, 1)
    ispec, rspec, sd = _case(C, hid, F, layers, seed=11 + C)
    H, W = 16, 16
    grid_t, rows = _rows(C, H, W, 0.5)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    un = (((yy - 8) ** 2 + (xx - 7) ** 2) > 20).float().reshape(-1, 1)
    steps = 15
    masks = O.rnvp_masks(C, F)
    pf, losses, logits = O.fit_pcn(sd, rows, un, steps, masks, torch.tensor(rspec.vmin), torch.tensor(rspec.vmax), lr=2e-3,
                                   optimizer=optimizer, flow_weight_decay=1e-2, plateau=dict(patience=3, factor=0.5))
    ip, fp = _split(ispec, rspec, sd, dev)
    res = R.pcn_fit(ispec, rspec, ip, fp, A.Grid.from_image_grid(grid_t.to(dev)), un.reshape(1, -1).to(dev), steps, lr=2e-3,
                    optimizer=optimizer, flow_weight_decay=1e-2, plateau=dict(patience=3, factor=0.5))
    assert int(res.status[0]) == 0
    np.testing.assert_allclose(res.loss_hist[0].cpu().numpy(), np.asarray(losses, np.float32), rtol=5e-4)
    got = _merge(ispec, rspec, res.icnn_params[0].cpu(), res.flow_params[0].cpu())
    for k in pf:
        np.testing.assert_allclose(got[k].numpy(), pf[k].numpy(), rtol=5e-3, atol=3e-4, err_msg=k)
    np.testing.assert_allclose(res.logits[0].cpu().numpy(), logits.reshape(-1).numpy(), rtol=5e-3, atol=2e-3)


def test_fresh_flow_is_identity_after_actnorm_init(dev):
    """init_zeros=True: every coupling starts as the identity, so after ActNorm's init the deformation is a per-channel
    affine map of the grid (net_factory.py:104-105)."""
    from awesome_amd import rnvp as R
    import awesome_amd as A
    torch.manual_seed(0)
    rspec = R.RnvpSpec(2, 32, 12)
    fp = R.init_rnvp_params(rspec)[None].to(dev)
    grid = A.Grid.linspace(32, 32, dev)
    R.actnorm_init(rspec, fp, grid)
    xd = R.rnvp_forward(rspec, fp, grid)[0].cpu()
    g = O.positional_grid(32, 32).reshape(2, -1)
    for c in range(2):
        A_ = torch.stack([g[c], torch.ones_like(g[c])], 1)
        sol = torch.linalg.lstsq(A_, xd[c][:, None]).solution
        assert float((A_ @ sol - xd[c][:, None]).abs().max()) < 1e-4


def test_module_surface_and_autograd(dev):
    """real_nvp_path_connected_net(...) (net_factory.py:124-175): state_dict keys, ActNorm init on first forward, forward
    and autograd backward through the HIP path vs the oracle on the module's own state_dict."""
    from awesome_amd.model import real_nvp_path_connected_net
    torch.manual_seed(3)
    m = real_nvp_path_connected_net(channels=2, hidden_units=32, flow_n_flows=12, flow_output_fn="tanh",
                                    convex_net_hidden_units=130, convex_net_hidden_layers=2).to(dev)
    keys = set(m.state_dict().keys())
    for k in ("linear.weight", "linear.bias", "convex_net.input.weight", "flow_net.norm.min", "flow_net.norm.new_max",
              "flow_net.net.network.flows.0.b", "flow_net.net.network.flows.0.s.net.0.weight",
              "flow_net.net.network.flows.0.t.net.2.bias", "flow_net.net.network.flows.1.s", "flow_net.net.network.flows.1.t",
              "flow_net.net.network.flows.1.data_dep_init_done", "flow_net.net.network.flows.23.t"):
        assert k in keys, k
    with torch.no_grad():   # move off the identity so that every gradient is non-trivial
        for k, p in m.named_parameters():
            if k.endswith("net.2.weight") or k.endswith("net.2.bias"):
                p.copy_(0.1 * torch.randn_like(p))
    H, W = 12, 20
    grid_t, rows = _rows(2, H, W)
    un = torch.rand(1, 1, H, W)
    out = torch.sigmoid(m(grid_t[None].to(dev)))
    assert out.shape == (1, 1, H, W)
    assert float(m.flow_net.net.network.flows[1].data_dep_init_done) == 1.0
    loss = ((out - un.to(dev)) ** 2).mean()
    loss.backward()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype == torch.float32 and not k.startswith("flow_net.norm")
           and not k.endswith("data_dep_init_done")}
    masks = O.rnvp_masks(2, 12)
    yo = O.pcn_forward(sdo, rows, masks, torch.zeros(2), torch.ones(2))
    lo = ((torch.sigmoid(yo).reshape(1, 1, H, W) - un) ** 2).mean()
    lo.backward()
    assert float(loss.detach()) == pytest.approx(float(lo.detach()), rel=3e-5)
    for k, p in m.named_parameters():
        ref = sdo[k].grad.numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=1e-3, atol=3e-5 * float(np.abs(ref).max()) + 1e-7, err_msg=k)
    xd = m.get_deformation(grid_t[None].to(dev))
    xr = O.pcn_deformation(sd, rows, masks, torch.zeros(2), torch.ones(2))
    np.testing.assert_allclose(xd[0].reshape(2, -1).cpu().numpy(), xr.t().numpy(), rtol=3e-5, atol=5e-6)


def test_fit_images_path_connected_shape(dev):
    """A non-convex but path-connected target (two discs joined by a bar): the convex prior alone cannot represent it, the
    path-connected prior fits it (the point of PathConnectedNet)."""
    import awesome_amd as A
    from awesome_amd.model import real_nvp_path_connected_net, ConvexNextNet
    torch.manual_seed(0)
    S = 64
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    mask = (((yy - 20) ** 2 + (xx - 18) ** 2) < 100) | (((yy - 44) ** 2 + (xx - 46) ** 2) < 100) | \
           (((yy - 20).abs() < 4) & (xx >= 18) & (xx <= 46)) | (((xx - 46).abs() < 4) & (yy >= 20) & (yy <= 44))
    un = (1.0 - mask.float()).reshape(1, -1).to(dev)
    grid = A.Grid.linspace(S, S, dev)
    m = real_nvp_path_connected_net(channels=2, hidden_units=32, flow_n_flows=12, flow_output_fn="tanh",
                                    convex_net_hidden_layers=2).to(dev)
    res = m.fit_images(grid, un, num_epochs=1500, lr=2e-3)
    iou_pc = float(A.miou(torch.sigmoid(res.logits), un)[0])
    cn = ConvexNextNet(n_hidden=130, n_hidden_layers=2, in_features=2)
    rc = A.fit(cn.spec, cn.flat_parameters()[None].to(dev), grid, un, 1500, lr=2e-3, optimizer="adamax")
    iou_cvx = float(A.miou(torch.sigmoid(rc.logits), un)[0])
    assert int(res.status[0]) == 0
    assert iou_pc > 0.9, (iou_pc, iou_cvx)
    assert iou_pc > iou_cvx + 0.1, (iou_pc, iou_cvx)


@pytest.mark.parametrize("q", [None, "1", "2", "4"])
@pytest.mark.parametrize("C", [2, 3])
def test_large_grid_two_points_per_lane(dev, C, q):
    """Launches of >= 196608 points run the RealNVP point kernels with Q points per lane (default: 2;
    INR_RNVP_QF / INR_RNVP_QB force 1 / 2 / 4 - the backward has 1 and 2): same results against the oracle (ragged tail included)."""
    import os
    from awesome_amd import rnvp as R
    import awesome_amd as A
    for k in ("INR_RNVP_QF", "INR_RNVP_QB"):
        os.environ.pop(k, None)
    if q is not None:
        os.environ["INR_RNVP_QF"] = q
        os.environ["INR_RNVP_QB"] = "2" if q == "4" else q
    try:
        _large_grid_case(dev, C)
    finally:
        for k in ("INR_RNVP_QF", "INR_RNVP_QB"):
            os.environ.pop(k, None)


def _large_grid_case(dev, C):
    from awesome_amd import rnvp as R
    import awesome_amd as A
    F = 4 if C == 2 else 6
    ispec, rspec, sd = _case(C, 32, F, 1, seed=21 + C, h=64)
    H, W = 517, 509
    grid_t, rows = _rows(C, H, W)
    torch.manual_seed(1)
    un = (torch.rand(H * W, 1) > 0.5).float()
    masks = O.rnvp_masks(C, F)
    vmin, vmax = torch.tensor(rspec.vmin), torch.tensor(rspec.vmax)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yo = O.pcn_forward(sdo, rows, masks, vmin, vmax)
    lo = O.weighted_loss(torch.sigmoid(yo).reshape(1, 1, -1, 1), un.reshape(1, 1, -1, 1), "se")
    lo.backward()
    ip, fp = _split(ispec, rspec, sd, dev)
    grid = A.Grid.from_image_grid(grid_t.to(dev))
    y = R.pcn_forward(ispec, rspec, ip, fp, grid)
    np.testing.assert_allclose(y[0].cpu().numpy(), yo.detach().reshape(-1).numpy(), rtol=1e-4, atol=5e-5)
    loss, gi, gf = R.pcn_loss_grad(ispec, rspec, ip, fp, grid, un.reshape(1, -1).to(dev), loss="se")
    assert float(loss[0]) == pytest.approx(float(lo.detach()), rel=3e-5)
    got = _merge(ispec, rspec, gi[0].cpu(), gf[0].cpu())
    for k in sdo:
        ref = sdo[k].grad.numpy()
        np.testing.assert_allclose(got[k].numpy(), ref, rtol=2e-3, atol=1e-4 * float(np.abs(ref).max()) + 1e-8, err_msg=k)


@pytest.mark.parametrize("C", [2, 3])
def test_learn_flow_identity(dev, C):
    """inrfit_rnvp_fit_identity vs the oracle's learn_flow_identity loop (Adamax on the flow_net alone, weight decay; the
    trained 1x1 linear must be ignored)."""
    from awesome_amd import rnvp as R
    import awesome_amd as A
    F = 12 if C == 2 else 6
    ispec, rspec, sd = _case(C, 32, F, 1, seed=31 + C)
    H, W = 16, 24
    grid_t, rows = _rows(C, H, W, 0.6)
    masks = O.rnvp_masks(C, F)
    steps = 12
    pf, losses = O.fit_flow_identity(sd, rows, steps, masks, torch.tensor(rspec.vmin), torch.tensor(rspec.vmax), lr=1e-2, weight_decay=1e-3)
    _, fp = _split(ispec, rspec, sd, dev)
    hist, _ = R.fit_identity(rspec, fp, A.Grid.from_image_grid(grid_t.to(dev)), steps=steps, lr=1e-2, weight_decay=1e-3)
    np.testing.assert_allclose(hist[0].cpu().numpy(), np.asarray(losses, np.float32), rtol=5e-4)
    got = R.unpack_rnvp_params(rspec, fp[0].cpu())
    for k in got:
        np.testing.assert_allclose(got[k].numpy(), pf[k].numpy(), rtol=5e-3, atol=3e-4, err_msg=k)
    assert torch.equal(got["linear.weight"], sd["linear.weight"]) and torch.equal(got["linear.bias"], sd["linear.bias"])


def test_prefit_stages_through_the_module(dev):
    """fit_images with the reference's prefit kwargs (prefit_flow_net_identity, prefit_convex_net): after the identity stage the
    deformation is close to the identity, and the whole pipeline fits the two-disc shape."""
    import awesome_amd as A
    from awesome_amd.model import real_nvp_path_connected_net
    torch.manual_seed(0)
    S = 64
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    mask = (((yy - 20) ** 2 + (xx - 18) ** 2) < 100) | (((yy - 44) ** 2 + (xx - 46) ** 2) < 100) | \
           (((yy - 20).abs() < 4) & (xx >= 18) & (xx <= 46)) | (((xx - 46).abs() < 4) & (yy >= 20) & (yy <= 44))
    un = (1.0 - mask.float()).reshape(1, -1).to(dev)
    grid = A.Grid.linspace(S, S, dev)
    m = real_nvp_path_connected_net(channels=2, hidden_units=32, flow_n_flows=12, flow_output_fn="tanh").to(dev)
    g = O.positional_grid(S, S)[None].to(dev)
    hist = m.learn_flow_identity(grid, lr=1e-2, weight_decay=1e-5, max_iter=100)
    assert float(hist[-1]) < 0.05 * float(hist[0])
    assert float((m.get_deformation(g) - g).abs().max()) < 0.1
    res = m.fit_images(grid, un, num_epochs=1200, lr=2e-3, prefit_flow_net_identity=True, prefit_convex_net=True)
    assert float(A.miou(torch.sigmoid(res.logits), un)[0]) > 0.9


@pytest.mark.parametrize("C", [2, 3])
def test_inverse(dev, C):
    """inrfit_rnvp_inverse vs the oracle's inverse, and the round trip inverse(get_deformation(x)) = x."""
    from awesome_amd import rnvp as R
    import awesome_amd as A
    F = 12 if C == 2 else 6
    ispec, rspec, sd = _case(C, 32, F, 1, seed=41 + C)
    H, W = 13, 17
    grid_t, rows = _rows(C, H, W, 0.4)
    masks = O.rnvp_masks(C, F)
    vmin, vmax = torch.tensor(rspec.vmin), torch.tensor(rspec.vmax)
    _, fp = _split(ispec, rspec, sd, dev)
    xd = R.rnvp_forward(rspec, fp, A.Grid.from_image_grid(grid_t.to(dev)))
    back = R.rnvp_inverse(rspec, fp, xd[0].contiguous())
    np.testing.assert_allclose(back[0].cpu().numpy(), rows.t().numpy(), rtol=2e-4, atol=2e-5)
    pts = torch.rand(200, C) * 0.8 + 0.1
    ref = O.pcn_inverse(sd, pts, masks, vmin, vmax)
    got = R.rnvp_inverse(rspec, fp, pts.t().contiguous().to(dev))
    np.testing.assert_allclose(got[0].cpu().numpy(), ref.t().numpy(), rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("C,hid,F", [(3, 130, 18), (2, 32, 12)])
def test_accuracy_against_float64(dev, C, hid, F):
    """How far is the HIP path from the exact result, compared with how far the fp32 torch restatement is?  Both are measured
    against the same restatement evaluated in float64: the HIP flow gradients must be as accurate as fp32 torch (x4 + floor)."""
    from awesome_amd import rnvp as R
    import awesome_amd as A
    ispec, rspec, sd = _case(C, hid, F, 1, seed=C * 7 + F)
    H, W = 11, 19
    grid_t, rows = _rows(C, H, W)
    torch.manual_seed(1)
    un = torch.rand(H * W, 1)
    masks = O.rnvp_masks(C, F)

    def run(dtype):
        sdo = {k: v.clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
        yo = O.pcn_forward(sdo, rows.to(dtype), masks, torch.tensor(rspec.vmin, dtype=dtype), torch.tensor(rspec.vmax, dtype=dtype))
        lo = O.weighted_loss(torch.sigmoid(yo).reshape(1, 1, -1, 1), un.to(dtype).reshape(1, 1, -1, 1), "se", "none")
        lo.backward()
        return {k: v.grad.double().numpy() for k, v in sdo.items()}

    g32, g64 = run(torch.float32), run(torch.float64)
    ip, fp = _split(ispec, rspec, sd, dev)
    _, gi, gf = R.pcn_loss_grad(ispec, rspec, ip, fp, A.Grid.from_image_grid(grid_t.to(dev)), un.reshape(1, -1).to(dev), loss="se")
    got = _merge(ispec, rspec, gi[0].cpu(), gf[0].cpu())
    e_hip, e_ref = [], []
    for k in g64:
        sc = np.abs(g64[k]).max() + 1e-30
        e_hip.append(np.abs(got[k].double().numpy() - g64[k]).max() / sc)
        e_ref.append(np.abs(g32[k] - g64[k]).max() / sc)
    assert max(e_hip) <= 4.0 * max(e_ref) + 2e-6, (max(e_hip), max(e_ref))
    assert np.median(e_hip) <= 4.0 * np.median(e_ref) + 5e-7, (np.median(e_hip), np.median(e_ref))


def test_fit_sequence_minibatches(dev):
    """_non_prior_based_pretrain semantics: one (x, y, t) network, one optimizer step per mini-batch of frames, plateau per
    epoch - the host loop over `inrfit_pcn_fit(steps=1)` against the oracle's loop."""
    from awesome_amd.model import real_nvp_path_connected_net
    torch.manual_seed(2)
    C, F, T, H, W = 3, 6, 4, 8, 10
    m = real_nvp_path_connected_net(channels=3, hidden_units=32, flow_n_flows=F, flow_output_fn="tanh",
                                    convex_net_hidden_units=130, convex_net_hidden_layers=1).to(dev)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith("net.2.weight") or k.endswith("net.2.bias"):
                p.copy_(0.1 * torch.randn_like(p))
        for a in m.flow_net.net.network.flows:
            if hasattr(a, "data_dep_init_done"):
                a.data_dep_init_done.fill_(1.0)   # keep s = t = 0: the ActNorm init is tested elsewhere
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()
           if v.dtype == torch.float32 and not k.startswith("flow_net.norm") and not k.endswith("data_dep_init_done")}
    frames = [O.positional_grid(W, H, float(t), float(T - 1)) for t in range(T)]
    rows = [O.pixelize(g[None]) for g in frames]
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    uns = [(((yy - 4) ** 2 + (xx - 3 - t) ** 2) > 6).float().reshape(-1, 1) for t in range(T)]
    masks = O.rnvp_masks(C, F)
    pf, ref_losses = O.fit_pcn_minibatch(sd0, rows, uns, 3, 2, masks, torch.zeros(3), torch.ones(3), lr=2e-3, flow_weight_decay=1e-2,
                                        plateau=dict(patience=0, factor=0.5))
    coords = torch.stack([g.reshape(3, -1) for g in frames]).to(dev)
    un_t = torch.stack([u.reshape(-1) for u in uns]).to(dev)
    losses = m.fit_sequence(coords, un_t, num_epochs=3, lr=2e-3, flow_weight_decay=1e-2, batch_size=2,
                            plateau=dict(patience=0, factor=0.5))
    np.testing.assert_allclose(np.asarray(losses), np.asarray(ref_losses), rtol=5e-4)
    got = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    for k in pf:
        np.testing.assert_allclose(got[k].numpy(), pf[k].numpy(), rtol=5e-3, atol=3e-4, err_msg=k)


def test_batch_of_images_equals_single_fits(dev):
    """Independent images in one call: every image's result equals its own single-image fit (separable grid, C = 2) and
    an (x, y, t) batch with per-image explicit grids works the same way."""
    from awesome_amd import rnvp as R
    import awesome_amd as A
    ispec, rspec, sd = _case(2, 32, 12, 2, seed=51)
    H, W = 16, 24
    ip, fp = _split(ispec, rspec, sd, dev)
    torch.manual_seed(3)
    un = (torch.rand(3, H * W) > 0.5).float().to(dev)
    grid = A.Grid.linspace(W, H, dev)
    kw = dict(lr=2e-3, flow_weight_decay=1e-3, plateau=dict(patience=2, factor=0.5))
    res = R.pcn_fit(ispec, rspec, ip.repeat(3, 1).contiguous(), fp.repeat(3, 1).contiguous(), grid, un, 8, **kw)
    for i in range(3):
        one = R.pcn_fit(ispec, rspec, ip.clone(), fp.clone(), grid, un[i:i + 1].contiguous(), 8, **kw)
        np.testing.assert_allclose(res.loss_hist[i].cpu().numpy(), one.loss_hist[0].cpu().numpy(), rtol=1e-5)
        np.testing.assert_allclose(res.flow_params[i].cpu().numpy(), one.flow_params[0].cpu().numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(res.icnn_params[i].cpu().numpy(), one.icnn_params[0].cpu().numpy(), rtol=1e-4, atol=1e-6)
    # (x, y, t): per-image explicit grids
    ispec3, rspec3, sd3 = _case(3, 32, 6, 1, seed=52)
    ip3, fp3 = _split(ispec3, rspec3, sd3, dev)
    coords = torch.stack([_rows(3, H, W, t)[0].reshape(3, -1) for t in (0.2, 0.9)]).to(dev)
    un3 = (torch.rand(2, H * W) > 0.5).float().to(dev)
    both = R.pcn_fit(ispec3, rspec3, ip3.repeat(2, 1).contiguous(), fp3.repeat(2, 1).contiguous(), A.Grid.explicit(coords), un3, 5, **kw)
    for i in range(2):
        one = R.pcn_fit(ispec3, rspec3, ip3.clone(), fp3.clone(), A.Grid.explicit(coords[i].contiguous()), un3[i:i + 1].contiguous(), 5, **kw)
        np.testing.assert_allclose(both.loss_hist[i].cpu().numpy(), one.loss_hist[0].cpu().numpy(), rtol=1e-5)
        np.testing.assert_allclose(both.flow_params[i].cpu().numpy(), one.flow_params[0].cpu().numpy(), rtol=1e-4, atol=1e-6)


def test_fit_frames_warm_start(dev):
    """reuse_state chain of _prior_based_pretrain: frame 1 starts from frame 0's fit and needs only a few epochs."""
    import awesome_amd as A
    from awesome_amd.model import real_nvp_path_connected_net
    torch.manual_seed(0)
    S = 48
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")

    def frame(dx):
        m = (((yy - 16) ** 2 + (xx - 14 - dx) ** 2) < 60) | (((yy - 32) ** 2 + (xx - 32 - dx) ** 2) < 60) | \
            (((yy - 16).abs() < 3) & (xx >= 14 + dx) & (xx <= 32 + dx)) | (((xx - 32 - dx).abs() < 3) & (yy >= 16) & (yy <= 32))
        return 1.0 - m.float().reshape(-1)

    un = torch.stack([frame(0), frame(2), frame(4)]).to(dev)
    grid = A.Grid.linspace(S, S, dev)
    m = real_nvp_path_connected_net(channels=2, hidden_units=32, flow_n_flows=12, flow_output_fn="tanh").to(dev)
    ip, fp, iou, retries = m.fit_frames(grid, un, num_epochs=1200, reuse_state_epochs=150, lr=2e-3,
                                        prefit_flow_net_identity=True, prefit_convex_net=True)
    assert ip.shape[0] == 3 and fp.shape[0] == 3
    assert float(iou.min()) > 0.85, iou
    assert retries == [0, 0, 0]


@pytest.mark.parametrize("C,H,W", [(2, 1, 5), (3, 3, 1)])
def test_tiny_grids(dev, C, H, W):
    """Fewer points than one wave: forward, gradients and a few fit steps still match the oracle."""
    from awesome_amd import rnvp as R
    import awesome_amd as A
    F = 4
    ispec, rspec, sd = _case(C, 32, F, 1, seed=61 + C)
    grid_t, rows = _rows(C, H, W, 0.3)
    un = torch.rand(H * W, 1)
    masks = O.rnvp_masks(C, F)
    vmin, vmax = torch.tensor(rspec.vmin), torch.tensor(rspec.vmax)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lo = O.weighted_loss(torch.sigmoid(O.pcn_forward(sdo, rows, masks, vmin, vmax)).reshape(1, 1, -1, 1), un.reshape(1, 1, -1, 1), "se")
    lo.backward()
    ip, fp = _split(ispec, rspec, sd, dev)
    grid = A.Grid.from_image_grid(grid_t.to(dev))
    loss, gi, gf = R.pcn_loss_grad(ispec, rspec, ip, fp, grid, un.reshape(1, -1).to(dev), loss="se")
    assert float(loss[0]) == pytest.approx(float(lo.detach()), rel=3e-5)
    got = _merge(ispec, rspec, gi[0].cpu(), gf[0].cpu())
    for k in sdo:
        ref = sdo[k].grad.numpy()
        np.testing.assert_allclose(got[k].numpy(), ref, rtol=1e-3, atol=3e-5 * float(np.abs(ref).max()) + 1e-7, err_msg=k)
    pf, losses, _ = O.fit_pcn(sd, rows, un, 4, masks, vmin, vmax, lr=2e-3, flow_weight_decay=1e-3)
    res = R.pcn_fit(ispec, rspec, ip, fp, grid, un.reshape(1, -1).to(dev), 4, lr=2e-3, flow_weight_decay=1e-3)
    np.testing.assert_allclose(res.loss_hist[0].cpu().numpy(), np.asarray(losses, np.float32), rtol=5e-4)


def test_learn_flow_identity_with_zoo(dev, tmp_path):
    """learn_flow_identity(zoo=...) (path_connected_net.py:177-194, 246-248): the first model fits and stores, a second, freshly
    initialised model of the same architecture on the same grid loads that state instead of fitting."""
    import awesome_amd as A
    from awesome_amd.model import Zoo, real_nvp_path_connected_net
    zoo = Zoo(str(tmp_path))
    grid = A.Grid.linspace(32, 32, dev)
    torch.manual_seed(11)
    m1 = real_nvp_path_connected_net(channels=2, flow_n_flows=4, hidden_units=16).to(dev)
    h1 = m1.learn_flow_identity(grid, lr=1e-2, max_iter=30, zoo=zoo)
    assert h1.shape[0] == 30 and float(h1[-1]) < float(h1[0])
    torch.manual_seed(12)
    m2 = real_nvp_path_connected_net(channels=2, flow_n_flows=4, hidden_units=16).to(dev)
    before = {k: v.clone() for k, v in m2.flow_net.state_dict().items()}
    h2 = m2.learn_flow_identity(grid, lr=1e-2, max_iter=30, zoo=zoo)
    assert torch.equal(h2.cpu(), h1.cpu())
    changed = 0
    for k, v in m1.flow_net.state_dict().items():
        assert torch.equal(m2.flow_net.state_dict()[k].cpu(), v.cpu()), k
        changed += int(not torch.equal(before[k].cpu(), v.cpu()))
    assert changed > 0
    h3 = m2.learn_flow_identity(grid, lr=5e-3, max_iter=30, zoo=zoo)      # other hyper-parameters: a miss, fits again
    assert not torch.equal(h3.cpu(), h1.cpu())


def test_fast_tanh_exp_error_bounds(dev):
    """The kernels' own tanh / exp of the coupling outputs (rnvp.h fast_tanh / fast_exp: v_exp_f32 / v_rcp_f32 forms with a
    compensated argument, an odd polynomial near 0) against float64 on the device that runs them, with EXPLICIT bounds on the
    RELATIVE error (libm's float32 functions: ~1e-7): tanh <= 5e-7 everywhere incl. tiny arguments (the couplings start at exactly
    0, so absolute-error forms are not good enough), exp <= 2.5e-7 on [-10, 10] (the log-scales are tanh-bounded)."""
    import ctypes as C
    from awesome_amd import _lib as L
    from awesome_amd import icnn as K
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.linspace(-10, 10, 200001), (torch.rand(100000, generator=g) - 0.5) * 1.3, torch.logspace(-30, 0, 4000),
                   -torch.logspace(-30, 0, 4000), torch.tensor([0.0, 0.625, -0.625, 0.6249999, 20.0, -20.0, 50.0])]).float()
    xd = x.to(dev)
    th, ex = torch.empty_like(xd), torch.empty_like(xd)
    L.check(L.load().inrfit_debug_tanh_exp(xd.data_ptr(), xd.numel(), th.data_ptr(), ex.data_ptr(), K._stream_ptr(dev)), "debug_tanh_exp")
    x64 = x.double()
    t64, e64 = torch.tanh(x64), torch.exp(x64)
    rel_t = ((th.cpu().double() - t64).abs() / t64.abs().clamp_min(1e-300))[x != 0]
    assert float(th.cpu()[x == 0].abs().max()) == 0.0
    assert float(rel_t.max()) <= 5e-7, float(rel_t.max())
    sel = x.abs() <= 10
    rel_e = ((ex.cpu().double() - e64).abs() / e64)[sel]
    assert float(rel_e.max()) <= 2.5e-7, float(rel_e.max())
    print(f"fast_tanh max rel err {float(rel_t.max()):.3e}, fast_exp max rel err {float(rel_e.max()):.3e}")
