"""GPU parity of the teaser priors: rotational / mirror symmetry (awesome_amd/model/symmetric_net.py: pose in torch
on the device, the 3 -> h -> h -> 1 network on the fused HIP kernels) and star shape (awesome_amd/model/star_net.py: two chains of
the same kernels sharing W0) against the notebooks' own classes (tests/golden/teaser_rotation_symmetric.npz, teaser_star_shaped.npz)
and against the oracle's loop.  fp32: logits 1e-5 absolute, gradients 2e-4 relative
(of the tensor maximum), 8-step Adam trajectory 1e-4."""
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


@pytest.fixture(scope="module")
def z(golden_dir):
    return np.load(os.path.join(golden_dir, "teaser_rotation_symmetric.npz"))


def _module(z, dev, prefix="sd."):
    from awesome_amd.model import RotationSymmetricNet
    m = RotationSymmetricNet(130)
    m.load_state_dict({k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)})
    return m.to(dev)


@pytest.mark.parametrize("tag,sp", [("free", False), ("sym", True)])
def test_forward_and_all_gradients_match_the_notebook_class(dev, z, tag, sp):
    m = _module(z, dev)
    x = torch.from_numpy(z["x"]).to(dev).requires_grad_(True)
    y = m(x, sp)
    assert y.shape == (x.shape[0], 1)
    np.testing.assert_allclose(y.detach().cpu().numpy(), z[tag + ".y"], rtol=1e-5, atol=1e-5)
    (torch.sigmoid(y) ** 2).mean().backward()
    chk = lambda got, ref, name: np.testing.assert_allclose(  # noqa: E731
        got.cpu().numpy(), ref, rtol=2e-4, atol=2e-4 * float(np.abs(ref).max()) + 1e-9, err_msg=name)
    chk(x.grad, z[tag + ".dx"], "dx")
    for k, p in m.named_parameters():
        chk(p.grad, z[f"{tag}.grad.{k}"], k)     # offset and orientation get theirs through inrfit_backward's dcoords


def test_notebook_training_loop_through_autograd(dev, z):
    """8 full-batch Adam steps of the notebook's loss (2 MSE(background) + MSE(foreground)) over ALL parameters, pose included,
    with torch.optim.Adam driving the HIP forward/backward: the recorded trajectory of the notebook's class."""
    m = _module(z, dev)
    x, labels = torch.from_numpy(z["x"]).to(dev), torch.from_numpy(z["labels"]).to(dev)
    back, fore = labels < 0.5, labels > 0.5
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    losses = []
    for _ in range(8):
        ob, of = torch.sigmoid(m(x[back], True)).squeeze(), torch.sigmoid(m(x[fore], True)).squeeze()
        loss = 2 * ((ob - labels[back]) ** 2).mean() + ((of - labels[fore]) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    np.testing.assert_allclose(losses, z["adam8.loss"], rtol=1e-4)
    for k, v in m.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), z["adam8.sd." + k], rtol=1e-3, atol=2e-5, err_msg=k)


def test_fused_fit_under_a_fixed_pose_matches_the_oracle_loop(dev, z):
    """`fit`: inrfit_fit on the features of the current pose - 30 Adam steps against the oracle's loop over W0, W1, W2 (pose
    constant), then the pose stays untouched and the loss went down."""
    m = _module(z, dev)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x, labels = torch.from_numpy(z["x"]), torch.from_numpy(z["labels"])
    res = m.fit(x.to(dev), labels.to(dev), 30, lr=1e-3)
    p = {k: v.clone().requires_grad_(k not in ("offset", "orientation")) for k, v in sd.items()}
    opt = torch.optim.Adam([v for v in p.values() if v.requires_grad], lr=1e-3)
    losses = []
    for _ in range(30):
        opt.zero_grad()
        loss = ((torch.sigmoid(O.rotation_symmetric_forward(p, x, True))[:, 0] - labels) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    np.testing.assert_allclose(res.loss_hist[0].cpu().numpy(), np.asarray(losses, np.float32), rtol=1e-3)
    now = m.state_dict()
    assert torch.equal(now["offset"].cpu(), sd["offset"]) and torch.equal(now["orientation"].cpu(), sd["orientation"])
    for k in ("W0.weight", "W1.weight", "W2.weight", "W2.bias"):
        np.testing.assert_allclose(now[k].cpu().numpy(), p[k].detach().numpy(), rtol=2e-3, atol=2e-5, err_msg=k)


def test_alternating_fit_finds_the_mirror_axis(dev):
    """The teaser's use: a mirror-symmetric shape about an unknown, off-centre axis with one half of it corrupted.  Network steps
    on the fused fit, pose steps through autograd: the loss falls, the fitted mask is mirror-symmetric about the learned axis by
    construction and covers the clean shape better than the corrupted labels do."""
    from awesome_amd.model import RotationSymmetricNet
    torch.manual_seed(5)
    S = 96
    ii, jj = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    # pixel centres: no coordinate coincides with -offset (the radius sqrt has no gradient at 0, in the notebook as here)
    x = torch.stack([(ii.reshape(-1) + 0.5) / S - 0.5, (jj.reshape(-1) + 0.5) / S - 0.5], 1).float()
    ang, cx, cy = 0.5, 0.04, -0.03
    u = (x[:, 0] - cx) * np.cos(ang) + (x[:, 1] - cy) * np.sin(ang)
    v = -(x[:, 0] - cx) * np.sin(ang) + (x[:, 1] - cy) * np.cos(ang)
    clean = (((u / 0.32) ** 2 + ((v.abs() - 0.07) / 0.15) ** 2) < 1.0).float()
    noisy = clean.clone()
    noisy[(v > 0.05) & (u.abs() < 0.08)] = 0.0                 # a bite out of one half only
    m = RotationSymmetricNet(130).to(dev)
    with torch.no_grad():
        m.orientation.fill_(-ang + 0.15)                        # (u, v) above = R(-ang)(x - centre): start near, not at, the axis
    xd = x.to(dev)
    hist = m.fit_alternating(xd, noisy.to(dev), rounds=6, net_steps=150, pose_steps=10, lr=2e-3, pose_lr=2e-3)
    assert hist[-1] < 0.5 * hist[0] or hist[-1] < 0.03, hist
    with torch.no_grad():
        logits = m(xd)
        pred = (torch.sigmoid(logits)[:, 0] > 0.5).float().cpu()
        # mirror-symmetric about the learned axis by construction: reflect every point in the module's own frame
        # (centre -offset, axis angle `orientation`) and evaluate again
        c, s = torch.cos(m.orientation), torch.sin(m.orientation)
        xm = xd + m.offset
        a, b = xm[:, 0] * c - xm[:, 1] * s, xm[:, 0] * s + xm[:, 1] * c          # R(orientation) xm
        x_mirror = torch.stack((a * c - b * s, -a * s - b * c), 1) - m.offset      # R^T (a, -b) - offset
        np.testing.assert_allclose(m(x_mirror).cpu().numpy(), logits.cpu().numpy(), atol=2e-3)
    iou = lambda a, b: float(((a > 0.5) & (b > 0.5)).sum()) / float(((a > 0.5) | (b > 0.5)).sum())  # noqa: E731
    assert iou(pred, clean) > iou(noisy, clean) - 0.02 and iou(pred, clean) > 0.85, (iou(pred, clean), iou(noisy, clean))


# ---- star-shape prior (awesome_amd/model/star_net.py) ------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def zs(golden_dir):
    return np.load(os.path.join(golden_dir, "teaser_star_shaped.npz"))


def _star(zs, dev):
    from awesome_amd.model import StarShapedNet
    m = StarShapedNet(130)
    m.load_state_dict({k[3:]: torch.from_numpy(zs[k]) for k in zs.files if k.startswith("sd.")})
    m.offset.requires_grad = True
    return m.to(dev)


def test_star_forward_and_all_gradients_match_the_notebook_class(dev, zs):
    m = _star(zs, dev)
    x = torch.from_numpy(zs["x"]).to(dev).requires_grad_(True)
    y = m(x)
    assert y.shape == (x.shape[0], 1)
    np.testing.assert_allclose(y.detach().cpu().numpy(), zs["y"], rtol=1e-5, atol=1e-5)
    (torch.sigmoid(y) ** 2).mean().backward()
    chk = lambda got, ref, name: np.testing.assert_allclose(  # noqa: E731
        got.cpu().numpy(), ref, rtol=2e-4, atol=2e-4 * float(np.abs(ref).max()) + 1e-9, err_msg=name)
    chk(x.grad, zs["dx"], "dx")
    for k, p in m.named_parameters():
        chk(p.grad, zs["grad." + k], k)     # W0 is shared by the two kernel chains: its gradient is their sum


def test_star_notebook_training_loop_through_autograd(dev, zs):
    """8 full-batch Adam steps (lr 1e-2) of the notebook's loop - MSE on sigmoid outputs, then W2_r.weight <- relu(W2_r.weight) -
    over all parameters, the centre included, against the recorded trajectory of the notebook's class."""
    m = _star(zs, dev)
    x, labels = torch.from_numpy(zs["x"]).to(dev), torch.from_numpy(zs["labels"]).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    losses = []
    for _ in range(8):
        loss = ((torch.sigmoid(m(x)).squeeze() - labels) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        m.enforce_star_shape()
        losses.append(float(loss.detach()))
    np.testing.assert_allclose(losses, zs["adam8.loss"], rtol=2e-4)
    assert float(m.W2_r.weight.detach().min()) >= 0.0
    for k, v in m.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), zs["adam8.sd." + k], rtol=2e-3, atol=1e-4, err_msg=k)


def test_star_net_fits_a_non_convex_star_and_stays_star_shaped(dev):
    """The teaser's use: a five-armed star (not convex).  Minibatch Adam as in the notebook; the fitted region {out < 0} covers the
    star and is star-shaped about the centre: along (nearly) every ray from -offset, once outside always outside."""
    from awesome_amd.model import StarShapedNet
    torch.manual_seed(9)
    S = 64
    ii, jj = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    x = torch.stack([(ii.reshape(-1) + 0.5) / S - 0.5, (jj.reshape(-1) + 0.5) / S - 0.5], 1).float().to(dev)
    rad, phi = (x ** 2).sum(1).sqrt(), torch.atan2(x[:, 1], x[:, 0])
    inside = (rad < 0.24 + 0.11 * torch.cos(5 * phi)).float()
    labels = 1 - inside
    m = StarShapedNet(130).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    for _ in range(400):
        idx = torch.randperm(x.shape[0], device=dev)[:1000]
        loss = ((torch.sigmoid(m(x[idx])).squeeze() - labels[idx]) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        m.enforce_star_shape()
    with torch.no_grad():
        pred_in = (m(x)[:, 0] < 0).float()
        iou = float((pred_in * inside).sum() / ((pred_in + inside) > 0).float().sum())
        assert iou > 0.8, iou
        # rays from the centre (offset is frozen at 0 here): out along a ray changes sign at most once, from inside to outside
        t = torch.linspace(0.01, 0.7, 200, device=dev)
        ang = torch.linspace(0, 2 * np.pi, 91, device=dev)[:-1]
        pts = torch.stack((t[None, :] * torch.cos(ang)[:, None], t[None, :] * torch.sin(ang)[:, None]), -1).reshape(-1, 2)
        out = m(pts)[:, 0].reshape(90, 200)
        outside = (out >= 0).int()
        reenter = ((outside[:, 1:] - outside[:, :-1]).min(1).values < 0).float().mean()
        # (the architecture makes the bracket convex in r, W2_r >= 0; monotone only where W1_r >= 0, which the notebook does not
        # project - so this is a property of the trained network, checked with a small allowance)
        assert float(reenter) <= 0.05, f"{float(reenter):.3f} of the rays re-enter the region: not star-shaped"


# ---- the star prior's device-resident entry points (awesome_amd/star.py, csrc/star.h) --------------------------------------------------
def test_star_fused_forward_and_loss_grad_match_the_notebook_class(dev, zs):
    """inrfit_star_forward / inrfit_star_loss_grad against the fixture of the notebook's own class: logits, and - through the chain rule
    of the fixture's scalar sum(sigmoid(y)^2)/n, which is MSE against zero labels - every parameter gradient, the centre's included."""
    from awesome_amd import star as S
    m = _star(zs, dev)
    spec = m.star_spec
    assert spec.n_params == sum(p.numel() for p in m.parameters())
    flat = m.flat_parameters()
    x = torch.from_numpy(zs["x"]).to(dev)
    y = S.star_forward(spec, flat, x)
    np.testing.assert_allclose(y.cpu().numpy(), zs["y"][:, 0], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(m.forward_fused(x).cpu().numpy(), zs["y"], rtol=1e-5, atol=1e-5)
    loss, grads = S.star_loss_grad(spec, flat, x, torch.zeros(x.shape[0], device=dev))
    ref_loss = float((1.0 / (1.0 + np.exp(-zs["y"].astype(np.float64))) ** 2).mean())
    assert abs(float(loss) - ref_loss) <= 1e-5 * ref_loss
    g = S.unflatten(spec, grads)
    for k in S.PARAM_ORDER:
        ref = zs["grad." + k]
        np.testing.assert_allclose(g[k].cpu().numpy(), ref, rtol=2e-4, atol=2e-4 * float(np.abs(ref).max()) + 1e-9, err_msg=k)
    # a minibatch through the index path = the same pixels gathered by hand
    idx = torch.tensor([5, 17, 300, 44, 44, 359, 0, 128], dtype=torch.int32, device=dev)
    lab = torch.from_numpy(zs["labels"]).to(dev)
    l1, g1 = S.star_loss_grad(spec, flat, x, lab, idx)
    l2, g2 = S.star_loss_grad(spec, flat, x[idx.long()].contiguous(), lab[idx.long()].contiguous())
    assert torch.equal(l1, l2) and torch.equal(g1, g2)


def test_star_fused_fit_reproduces_the_notebook_trajectory(dev, zs):
    """inrfit_star_fit on the fixture's problem: 8 full-batch epochs, the centre trainable from the first one - losses and every
    parameter against the trajectory the notebook's class recorded (the same bars as the autograd loop above), run twice: same bits."""
    from awesome_amd import star as S
    m = _star(zs, dev)
    spec = m.star_spec
    x, labels = torch.from_numpy(zs["x"]).to(dev), torch.from_numpy(zs["labels"]).to(dev)
    full = torch.arange(x.shape[0], dtype=torch.int32, device=dev)[None].repeat(8, 1)
    runs = []
    for _ in range(2):
        res = S.star_fit(spec, m.flat_parameters(), x, labels, full, lr=1e-2, offset_first_step=0)
        runs.append(res)
    assert torch.equal(runs[0].params, runs[1].params) and torch.equal(runs[0].loss_hist, runs[1].loss_hist)
    np.testing.assert_allclose(runs[0].loss_hist.cpu().numpy(), zs["adam8.loss"], rtol=2e-4)
    out = S.unflatten(spec, runs[0].params)
    assert float(out["W2_r.weight"].min()) >= 0.0
    for k in S.PARAM_ORDER:
        np.testing.assert_allclose(out[k].cpu().numpy(), zs["adam8.sd." + k], rtol=2e-3, atol=1e-4, err_msg=k)


@pytest.mark.parametrize("h", [150, 37])
def test_star_fused_fit_against_the_oracle_loop_on_minibatches(dev, h):
    """The notebook's width (150) and an odd one: 12 epochs of 200-pixel minibatches, the centre freed after the forward pass of epoch 4
    (first step at epoch 5, its own Adam step count), continued in a second call - against oracle.star_shaped_fit on the CPU."""
    from awesome_amd import star as S
    from awesome_amd.model import StarShapedNet
    from oracle import inr_oracle as O
    torch.manual_seed(21 + h)
    m = StarShapedNet(h)
    with torch.no_grad():
        m.offset.copy_(torch.tensor([[0.02, -0.04]]))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    n = 48
    ii, jj = torch.meshgrid(torch.arange(n), torch.arange(n), indexing="ij")
    x = torch.stack([ii.reshape(-1) / (n - 1) - 0.5, jj.reshape(-1) / (n - 1) - 0.5], 1).float()
    rad, phi = ((x[:, 0] - 0.03) ** 2 + (x[:, 1] + 0.05) ** 2).sqrt(), torch.atan2(x[:, 1] + 0.05, x[:, 0] - 0.03)
    labels = 1 - (rad < 0.22 + 0.12 * torch.cos(5 * phi)).float()
    g = torch.Generator().manual_seed(1)
    idx = torch.stack([torch.randperm(n * n, generator=g)[:200] for _ in range(12)]).to(torch.int32)
    ref, ref_losses = O.star_shaped_fit(sd, x, labels, idx, lr=1e-2, offset_free_epoch=4)
    m = m.to(dev)
    state = {}
    r1 = m.fit(x.to(dev), labels.to(dev), lr=1e-2, offset_free_epoch=4, batch_index=idx[:7].to(dev), state=state)
    r2 = m.fit(x.to(dev), labels.to(dev), lr=1e-2, offset_free_epoch=4, batch_index=idx[7:].to(dev), state=state)
    assert state["epoch"] == 12
    losses = torch.cat([r1.loss_hist, r2.loss_hist]).cpu().numpy()
    np.testing.assert_allclose(losses, ref_losses, rtol=5e-4)
    assert not torch.equal(m.offset.detach().cpu(), sd["offset"])
    for k, v in m.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), ref[k].numpy(), rtol=5e-3, atol=2e-4, err_msg=k)
    # inference of the fitted module: the fused forward and the autograd surface agree
    xs = x[::7].to(dev).contiguous()
    np.testing.assert_allclose(m.forward_fused(xs).cpu().numpy(), m(xs).detach().cpu().numpy(), rtol=1e-4, atol=1e-4)


def test_star_fused_fit_of_the_notebook_loop_is_star_shaped(dev):
    """star.ipynb cell 3 end to end on the device (random minibatches drawn like its randperm pairs, 600 epochs, the centre freed at
    epoch 300): the five-armed star is covered, the region stays star-shaped along rays from the learned centre, and the call is
    faster than the same epochs through autograd + torch.optim by a wide margin (printed)."""
    import time
    from awesome_amd.model import StarShapedNet
    torch.manual_seed(9)
    S_ = 96
    ii, jj = torch.meshgrid(torch.arange(S_), torch.arange(S_), indexing="ij")
    x = torch.stack([ii.reshape(-1) / (S_ - 1) - 0.5, jj.reshape(-1) / (S_ - 1) - 0.5], 1).float().to(dev)
    rad, phi = ((x[:, 0] - 0.04) ** 2 + (x[:, 1] + 0.03) ** 2).sqrt(), torch.atan2(x[:, 1] + 0.03, x[:, 0] - 0.04)
    inside = (rad < 0.24 + 0.11 * torch.cos(5 * phi)).float()
    labels = 1 - inside
    m = StarShapedNet(150).to(dev)
    gen = torch.Generator(device=dev).manual_seed(4)
    m.fit(x, labels, num_epochs=10, generator=gen)                      # warm-up (rocBLAS handle, workspaces)
    m = StarShapedNet(150).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = m.fit(x, labels, num_epochs=600, number=500, lr=1e-2, offset_free_epoch=300, generator=gen)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    h = res.loss_hist.cpu().numpy()
    assert np.isfinite(h).all() and h[-50:].mean() < 0.35 * h[:5].mean(), (h[:5], h[-5:])
    print(f"\nstar fused fit: 600 epochs x 1000 pixels, h = 150: {dt * 1e3:.0f} ms ({dt / 600 * 1e6:.0f} us per epoch incl. the minibatch draw)")
    with torch.no_grad():
        pred_in = (m.forward_fused(x)[:, 0] < 0).float()
        iou = float((pred_in * inside).sum() / ((pred_in + inside) > 0).float().sum())
        assert iou > 0.8, iou
        c = -m.offset.detach()[0]
        t = torch.linspace(0.01, 0.7, 200, device=dev)
        ang = torch.linspace(0, 2 * np.pi, 91, device=dev)[:-1]
        pts = torch.stack((c[0] + t[None, :] * torch.cos(ang)[:, None], c[1] + t[None, :] * torch.sin(ang)[:, None]), -1).reshape(-1, 2)
        out = m.forward_fused(pts.contiguous())[:, 0].reshape(90, 200)
        outside = (out >= 0).int()
        reenter = ((outside[:, 1:] - outside[:, :-1]).min(1).values < 0).float().mean()
        assert float(reenter) <= 0.05, f"{float(reenter):.3f} of the rays re-enter the region: not star-shaped"


@pytest.mark.parametrize("h,n,batch", [(1, 1, 1), (6, 9, 5), (7, 5, 3), (64, 19, 19), (257, 9, 4), (350, 40, 16)])
def test_star_fused_edge_sizes(dev, h, n, batch):
    """Tiny and odd sizes (one hidden unit, one pixel, batches that fill no block, a width past 256; rows on 4- / 8- / 16-byte
    boundaries: the three load widths of gemm.h's general kernel): forward, loss and gradients
    against the oracle's restatement of the notebook class, and a 3-epoch fit that stays finite."""
    from awesome_amd import star as S
    from awesome_amd.model import StarShapedNet
    from oracle import inr_oracle as O
    torch.manual_seed(100 + h)
    m = StarShapedNet(h)
    with torch.no_grad():
        m.offset.copy_(torch.tensor([[0.03, -0.02]]))
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    x = (torch.rand(n, 2) - 0.5)
    labels = (torch.rand(n) > 0.5).float()
    idx = torch.randint(0, n, (batch,), dtype=torch.int32)
    spec = S.StarSpec(h)
    flat = S.flatten_state_dict(spec, {k: v.detach() for k, v in sd.items()}, dev)
    y = S.star_forward(spec, flat, x.to(dev))
    yo = O.star_shaped_forward(sd, x)
    np.testing.assert_allclose(y.cpu().numpy(), yo.detach().numpy()[:, 0], rtol=2e-5, atol=2e-5)
    lo = ((torch.sigmoid(O.star_shaped_forward(sd, x[idx.long()])).reshape(-1) - labels[idx.long()]) ** 2).mean()
    lo.backward()
    loss, grads = S.star_loss_grad(spec, flat, x.to(dev), labels.to(dev), idx.to(dev))
    assert float(loss) == pytest.approx(float(lo.detach()), rel=2e-5, abs=1e-7)
    g = S.unflatten(spec, grads)
    for k in S.PARAM_ORDER:
        ref = sd[k].grad.numpy()
        np.testing.assert_allclose(g[k].cpu().numpy(), ref, rtol=5e-4, atol=5e-4 * float(np.abs(ref).max()) + 1e-8, err_msg=k)
    res = S.star_fit(spec, flat, x.to(dev), labels.to(dev), idx[None].repeat(3, 1).to(dev), offset_first_step=1)
    assert bool(torch.isfinite(res.params).all()) and bool(torch.isfinite(res.loss_hist).all())
