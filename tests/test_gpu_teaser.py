"""GPU parity of the rotational-symmetry teaser prior (awesome_amd/model/symmetric_net.py): the module - pose in
torch on the device, the 3 -> h -> h -> 1 network on the fused HIP kernels - against the notebook's own class
(tests/golden/teaser_rotation_symmetric.npz) and against the oracle's loop.  fp32: logits 1e-5 absolute, gradients 2e-4 relative
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
