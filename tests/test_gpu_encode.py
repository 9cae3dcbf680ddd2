"""GPU parity of the ENCODE stage (layer-0 activations INR_ACT_COS / INR_ACT_SIN of the fused step kernels):
random Fourier features and the sine layer of the reference's notebooks, HIP vs the oracle (whose encode arithmetic is pinned on
the notebooks' own classes, tests/golden/encode_notebooks.npz).  fp32 tolerance: the kernels use v_sin_f32 / v_cos_f32 after a
v_fract_f32 range reduction, absolute error <~ 1e-5 at |argument| ~ 100; logits 1e-4, gradients 3e-3 of the tensor maximum."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import inr_oracle as O  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _oracle_params(m):
    """The module's parameters as the oracle's ICNN dict (zero skip weights)."""
    import awesome_amd as A
    return A.unpack_params(m.spec, m.flat_parameters().cpu())


@pytest.mark.parametrize("kind,layers,h", [("fourier", 1, 130), ("fourier", 2, 130), ("sine", 1, 130), ("sine", 2, 64), ("fourier", 1, 32)])
def test_encode_forward_loss_and_gradients(dev, kind, layers, h):
    import awesome_amd as A
    from awesome_amd.model import FourierFeatureNet, SineLayerNet
    torch.manual_seed(61 + layers)
    m = FourierFeatureNet(d_in=2, n_hidden=h, n_hidden_layers=layers, factor=30.0) if kind == "fourier" else \
        SineLayerNet(in_features=2, n_hidden=h, n_hidden_layers=layers)
    if kind == "sine":
        with torch.no_grad():
            m.offset.copy_(torch.tensor([[0.03, -0.07]]))
    act0, omega = m.spec.act0, m.spec.omega
    H, W = 13, 11   # N = 143: ragged
    grid = O.positional_grid(W, H)[None] - (0.5 if kind == "sine" else 0.0)      # the sine notebook centres its coordinates
    un = (torch.rand(1, 1, H, W) > 0.5).float()
    p = {k: v.clone().requires_grad_(True) for k, v in _oracle_params(m).items()}
    yo = O.icnn_forward_image(p, grid, act0, omega)
    lo = O.weighted_loss(torch.sigmoid(yo), un, "se")
    lo.backward()
    flat = m.flat_parameters()[None].to(dev)
    g = A.Grid.from_image_grid(grid.to(dev))
    y = A.forward(m.spec, flat, g)
    np.testing.assert_allclose(y[0].cpu().numpy(), yo.detach().reshape(-1).numpy(), atol=1e-4, rtol=1e-4)
    loss, grads = A.loss_grad(m.spec, flat, g, un.reshape(1, -1).to(dev))
    assert float(loss[0]) == pytest.approx(float(lo.detach()), rel=1e-4)
    got = A.unpack_params(m.spec, grads[0].cpu())
    for k, v in got.items():
        if k.endswith("skp.weight"):
            continue                                   # the skip weights are constants of this model family
        ref = p[k].grad.numpy()
        np.testing.assert_allclose(v.numpy(), ref, rtol=3e-3, atol=3e-3 * float(np.abs(ref).max()) + 1e-9, err_msg=k)
    # the module surface: forward + autograd through the drop-in class
    m.to(dev)
    ym = m(grid.to(dev))
    assert ym.shape == (1, 1, H, W)
    np.testing.assert_allclose(ym.detach().cpu().numpy().reshape(-1), yo.detach().reshape(-1).numpy(), atol=1e-4, rtol=1e-4)
    ((torch.sigmoid(ym) - un.to(dev)) ** 2).mean().backward()
    np.testing.assert_allclose(m.out.weight.grad.cpu().numpy(), p["out.ln.weight"].grad.numpy(), rtol=3e-3,
                               atol=3e-3 * float(p["out.ln.weight"].grad.abs().max()))
    if kind == "fourier":
        assert m.A.grad is None                          # buffers: the features are fixed


def test_configs0_disc64_fit_with_encode(dev):
    """BASELINE configs[0] with the encode stage: the 64x64 disc, a Fourier-feature MLP and a sine-layer MLP, no prior - the
    fused fit against the oracle's loop on the same arithmetic (first steps), frozen features, and a proper final mask."""
    import awesome_amd as A
    from awesome_amd.dataset import disc_unaries
    from awesome_amd.model import FourierFeatureNet, SineLayerNet
    S = 64
    un = disc_unaries(S, S, 32, 32, 15.0).reshape(1, -1).to(dev)
    grid = A.Grid.linspace(S, S, dev)
    for make, lr in ((lambda: FourierFeatureNet(n_hidden=130, n_hidden_layers=1, factor=10.0), 2e-3),
                     (lambda: SineLayerNet(n_hidden=130, n_hidden_layers=1, omega=10.0), 2e-3)):
        torch.manual_seed(71)
        m = make()
        p0 = _oracle_params(m)
        flat = m.flat_parameters()[None].to(dev)
        res = A.fit(m.spec, flat.clone(), grid, un, 600, lr=lr, loss="se", optimizer="adam", **type(m).fit_options)
        iou = float(A.miou((torch.sigmoid(res.logits) > 0.5).float(), (un > 0.5).float())[0])
        assert iou > 0.97, (type(m).__name__, iou)
        got = A.unpack_params(m.spec, res.params[0].cpu())
        assert all(float(v.abs().max()) == 0.0 for k, v in got.items() if k.endswith("skp.weight"))
        if isinstance(m, FourierFeatureNet):
            assert torch.equal(got["input.weight"], p0["input.weight"]) and torch.equal(got["input.bias"], p0["input.bias"])
        # the oracle's loop on the same arithmetic: 30 Adam steps (frozen tensors masked out of the update like the kernel does)
        p = {k: v.clone().requires_grad_(True) for k, v in p0.items()}
        frozen = {k for k in p if k.endswith("skp.weight")} | ({"input.weight", "input.bias"} if isinstance(m, FourierFeatureNet) else set())
        opt = torch.optim.Adam([v for k, v in p.items() if k not in frozen], lr=lr)
        losses = []
        for _ in range(30):
            opt.zero_grad()
            out = torch.sigmoid(O.icnn_forward_image(p, O.positional_grid(S, S)[None], m.spec.act0, m.spec.omega))
            loss = O.weighted_loss(out, un.cpu().reshape(1, 1, S, S), "se")
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        np.testing.assert_allclose(res.loss_hist[0, :30].cpu().numpy(), np.asarray(losses, np.float32), rtol=2e-3)
