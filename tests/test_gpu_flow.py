"""GPU parity: coordinate gradients of the HIP ICNN and the path-connected prior ICNN(flow(Ax+b)) vs the CPU oracle."""
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


@pytest.mark.parametrize("h,c,l", [(130, 2, 1), (130, 3, 1), (130, 2, 2), (64, 3, 2), (32, 2, 1)])
def test_coordinate_gradient(dev, h, c, l):
    import awesome_amd as A
    torch.manual_seed(3)
    spec = A.IcnnSpec(h, c, l)
    p = {k: (torch.rand(shp) - 0.4) * 0.4 for k, shp in spec.keys_shapes()}
    N = 150  # ragged: not a multiple of 64
    coords = (torch.rand(c, N) * 2 - 0.5)
    dlog = torch.randn(N)
    # oracle: autograd through the restated forward
    xr = coords.t().clone().requires_grad_(True)
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    y = O.icnn_forward(pr, xr)[:, 0]
    (y * dlog).sum().backward()
    flat = A.pack_state_dict(spec, p, dev)[None]
    grads, dco = A.icnn.backward(spec, flat, A.Grid.explicit(coords.to(dev)), dlog[None].to(dev), want_dcoords=True)
    ref = xr.grad.t().numpy()
    np.testing.assert_allclose(dco[0].cpu().numpy(), ref, rtol=2e-4, atol=2e-6 * float(np.abs(ref).max()))
    g = A.unpack_params(spec, grads[0].cpu())
    for k in pr:
        r = pr[k].grad.numpy()
        np.testing.assert_allclose(g[k].numpy(), r, rtol=2e-4, atol=2e-6 * float(np.abs(r).max()) + 1e-9, err_msg=k)


def test_flow_modules_match_reference_fixture(dev, golden_dir):
    from awesome_amd.model import NormalizingFlow1D
    z = np.load(os.path.join(golden_dir, "flow.npz"))
    x = torch.from_numpy(z["x"]).to(dev)
    for tag, kw in [("nf6_w130", dict(num_coupling=6, width=130)), ("nf4_w16", dict(num_coupling=4, width=16))]:
        nf = NormalizingFlow1D(in_features=2, backbone="normal_block", **kw)
        nf.load_state_dict(O.load_npz_state(z, f"{tag}.sd."))
        nf.to(dev)
        y = nf(x)
        np.testing.assert_allclose(y.detach().cpu().numpy(), z[f"{tag}.y"], rtol=1e-5, atol=1e-6)
        (y ** 2).mean().backward()
        for k, prm in nf.named_parameters():
            np.testing.assert_allclose(prm.grad.cpu().numpy(), z[f"{tag}.grad.{k}"], rtol=2e-4, atol=1e-7, err_msg=k)


@pytest.mark.parametrize("layers", [1, 2])
def test_convex_diffeomorphism_net(dev, layers):
    """ICNN(flow(Ax+b)): forward and every parameter gradient (ICNN on the HIP path incl. the coordinate gradient that
    drives the flow) against the oracle's restatement with autograd."""
    from awesome_amd.model import ConvexDiffeomorphismNet
    torch.manual_seed(11)
    m = ConvexDiffeomorphismNet(n_hidden=130, n_hidden_layers=layers, nf_layers=6, nf_hidden=24, in_features=2)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    H, W = 12, 10
    grid = O.positional_grid(W, H)[None]
    un = (torch.rand(1, 1, H, W) > 0.5).float()
    # oracle
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    rows = O.pixelize(grid)
    yo = O.convex_diffeo_forward(sdo, rows, 6)
    lo = O.weighted_loss(torch.sigmoid(O.unpixelize(yo, 1, H, W)), un, "se")
    lo.backward()
    # HIP path
    m.to(dev)
    y = m(grid.to(dev))
    assert y.shape == (1, 1, H, W)
    np.testing.assert_allclose(y.detach().cpu().numpy().reshape(-1), yo.detach().numpy().reshape(-1), atol=2e-5, rtol=1e-4)
    loss = ((torch.sigmoid(y) - un.to(dev)) ** 2).mean()
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(lo.detach()), rel=1e-5)
    for k, prm in m.named_parameters():
        ref = sdo[k].grad.numpy()
        scale = float(np.abs(ref).max())
        np.testing.assert_allclose(prm.grad.cpu().numpy(), ref, rtol=1e-3, atol=1e-5 * scale + 1e-7, err_msg=k)  # floor: d/dv of g*v/|v| is exactly 0 for a scalar v
