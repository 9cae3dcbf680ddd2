"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import inr_oracle as O  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def amd():
    import awesome_amd
    awesome_amd._lib.load()
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return awesome_amd


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _spec_from_sd(amd, p):
    h, c = p["input.weight"].shape
    return amd.IcnnSpec(n_hidden=h, in_features=c, n_layers=O.icnn_num_layers(p))


L1_CASES = ["convexnet_h130_c2", "convexnext_h130_c2_l1", "convexnext_h130_c3_l1", "convexnext_h32_c2_l1",
            "convexnext_h130_c2_l2", "convexnext_h64_c3_l2"]   # all golden model shapes, one and two hidden skip layers


@pytest.mark.parametrize("name", L1_CASES)
def test_forward_matches_reference(amd, golden_dir, name):
    z = _load(golden_dir, f"icnn_{name}.npz")
    p = O.to_convexnext_keys(O.load_npz_state(z, "sd0."))
    spec = _spec_from_sd(amd, p)
    dev = torch.device("cuda:0")
    grid_t = torch.from_numpy(z["grid"])  # (1,C,H,W)
    flat = amd.pack_state_dict(spec, p, dev)[None]
    logits = amd.forward(spec, flat, amd.Grid.from_image_grid(grid_t.to(dev)))
    ref = z["logits"].reshape(1, -1)
    # fp32 tolerance: |logit| ~ 1, K = 130 dot products in a different summation order
    np.testing.assert_allclose(logits.cpu().numpy(), ref, rtol=0, atol=5e-6)
    # separable grid description gives the same result as the explicit one
    _, C, H, W = grid_t.shape
    ts = None
    if C == 3:
        ts = grid_t[0, 2, 0, :1].to(dev)
    g2 = amd.Grid.separable(grid_t[0, 0, 0, :].to(dev), grid_t[0, 1, :, 0].to(dev), ts)
    logits2 = amd.forward(spec, flat, g2)
    assert torch.equal(logits, logits2)


@pytest.mark.parametrize("name", L1_CASES)
@pytest.mark.parametrize("tag,kind,mode", [("se", "se", "none"), ("bce", "bce", "none"), ("sssdms", "se", "sssdms")])
def test_loss_and_grads_match_reference(amd, golden_dir, name, tag, kind, mode):
    z = _load(golden_dir, f"icnn_{name}.npz")
    sd_raw = O.load_npz_state(z, "sd0.")
    p = O.to_convexnext_keys(sd_raw)
    spec = _spec_from_sd(amd, p)
    dev = torch.device("cuda:0")
    grid = amd.Grid.from_image_grid(torch.from_numpy(z["grid"]).to(dev))
    un = torch.from_numpy(z["unaries"])
    if mode == "sssdms":
        un = (un >= 0.5).float()
    flat = amd.pack_state_dict(spec, p, dev)[None]
    loss, grads = amd.loss_grad(spec, flat, grid, un.reshape(1, -1).to(dev), loss=kind, weight_mode=mode)
    assert float(loss[0]) == pytest.approx(float(z[f"{tag}.loss"]), rel=1e-5)
    g = amd.unpack_params(spec, grads[0].cpu())
    is_cn = "W0y.weight" in sd_raw
    for k_raw in sd_raw:
        k = O.CONVEXNET_KEYMAP[k_raw] if is_cn else k_raw
        ref = z[f"{tag}.grad.{k_raw}"]
        scale = max(1e-8, float(np.abs(ref).max()))
        np.testing.assert_allclose(g[k].numpy(), ref, rtol=2e-4, atol=2e-6 * scale + 1e-9, err_msg=f"{tag} {k_raw}")


@pytest.mark.parametrize("name", L1_CASES)
def test_adam_clamp_trajectory(amd, golden_dir, name):
    z = _load(golden_dir, f"icnn_{name}.npz")
    sd_raw = O.load_npz_state(z, "sd0.")
    is_cn = "W0y.weight" in sd_raw
    p = O.to_convexnext_keys(sd_raw)
    spec = _spec_from_sd(amd, p)
    dev = torch.device("cuda:0")
    grid = amd.Grid.from_image_grid(torch.from_numpy(z["grid"]).to(dev))
    un = torch.from_numpy(z["unaries"]).reshape(1, -1).to(dev)
    for steps, tag in [(1, "adam1."), (10, "adam10.")]:
        flat = amd.pack_state_dict(spec, p, dev)[None].clone()
        res = amd.fit(spec, flat, grid, un, steps, lr=2e-3)
        assert int(res.status[0]) == 0
        np.testing.assert_allclose(res.loss_hist[0].cpu().numpy(), z["adam.losses"][:steps], rtol=2e-5)
        got = amd.unpack_params(spec, res.params[0].cpu())
        for k_raw in sd_raw:
            k = O.CONVEXNET_KEYMAP[k_raw] if is_cn else k_raw
            np.testing.assert_allclose(got[k].numpy(), z[tag + k_raw], rtol=2e-4, atol=5e-6, err_msg=f"{tag}{k_raw}")
        for k in spec.clamp_keys():
            assert float(got[k].min()) >= 0.0


def test_adamax_plateau_weight_decay(amd, golden_dir):
    z = _load(golden_dir, "adamax_plateau_h32.npz")
    p = O.load_npz_state(z, "sd0.")
    spec = _spec_from_sd(amd, p)
    dev = torch.device("cuda:0")
    grid = amd.Grid.from_image_grid(torch.from_numpy(z["grid"]).to(dev))
    un = torch.from_numpy(z["unaries"]).reshape(1, -1).to(dev)
    n = len(z["losses"])
    flat = amd.pack_state_dict(spec, p, dev)[None].clone()
    res = amd.fit(spec, flat, grid, un, n, lr=1e-2, optimizer="adamax", weight_decay=1e-5,
                  plateau=dict(patience=5, factor=0.5))
    np.testing.assert_allclose(res.loss_hist[0].cpu().numpy(), z["losses"], rtol=5e-4)
    hdr = res.opt_state[0, 2 * spec.n_params:].cpu().numpy()
    assert hdr[2] == pytest.approx(float(z["lrs"][-1]), rel=1e-6)  # same LR schedule decisions as torch
    got = amd.unpack_params(spec, res.params[0].cpu())
    for k in p:
        np.testing.assert_allclose(got[k].numpy(), z["final." + k], rtol=5e-3, atol=5e-5, err_msg=k)


def test_fit_disc64_end_to_end(amd, golden_dir):
    """600 steps on the 64x64 disc: loss curve, mask and mIoU of the reference run."""
    z = _load(golden_dir, "fit_disc64.npz")
    p = O.load_npz_state(z, "sd0.")
    spec = _spec_from_sd(amd, p)
    dev = torch.device("cuda:0")
    grid = amd.Grid.linspace(64, 64, dev)
    un = torch.from_numpy(z["unaries"]).reshape(1, -1).to(dev)
    flat = amd.pack_state_dict(spec, p, dev)[None].clone()
    res = amd.fit(spec, flat, grid, un, 600, lr=2e-3)
    hist = res.loss_hist[0].cpu().numpy()
    np.testing.assert_allclose(hist[:50], z["losses"][:50], rtol=2e-4)
    assert hist[-1] == pytest.approx(float(z["losses"][-1]), rel=5e-2)
    prob = torch.sigmoid(res.logits)
    iou = amd.miou((prob > 0.5).float(), (un > 0.5).float(), invert=True)
    assert abs(float(iou[0]) - float(z["final_miou"])) <= 1e-3  # north-star tolerance on mIoU
    mask = (prob > 0.5).cpu().numpy().reshape(64, 64)
    assert (mask != z["final_mask"].reshape(64, 64)).mean() < 2e-3


def test_batched_ragged_and_explicit(amd):
    """Several images at once, N not a multiple of the 64-point chunk, per-image explicit grids; vs the oracle."""
    torch.manual_seed(5)
    dev = torch.device("cuda:0")
    spec = amd.IcnnSpec(n_hidden=64, in_features=3, n_layers=1)
    n_img, H, W = 3, 7, 11  # N = 77
    ps, flats, grids, uns = [], [], [], []
    for i in range(n_img):
        p = {k: (torch.rand(shp) - 0.4) * 0.5 for k, shp in spec.keys_shapes()}
        ps.append(p)
        flats.append(amd.pack_state_dict(spec, p))
        grids.append(torch.rand(1, 3, H, W) * 2 - 0.5)  # negative coordinates too
        uns.append(torch.rand(1, 1, H, W))
    flat = torch.stack(flats).to(dev)
    grid = amd.Grid.explicit(torch.cat(grids).reshape(n_img, 3, H * W).to(dev))
    un = torch.cat(uns).reshape(n_img, -1).to(dev)
    logits = amd.forward(spec, flat, grid).cpu()
    loss, grads = amd.loss_grad(spec, flat, grid, un, loss="bce", weight_mode="equal")
    for i in range(n_img):
        ref = O.icnn_forward_image(ps[i], grids[i]).reshape(-1)
        np.testing.assert_allclose(logits[i].numpy(), ref.numpy(), atol=5e-6, rtol=1e-5)
        l_ref, g_ref = O.loss_and_grads(ps[i], grids[i], uns[i], "bce", "equal")
        assert float(loss[i]) == pytest.approx(l_ref, rel=2e-5)
        g = amd.unpack_params(spec, grads[i].cpu())
        for k in g_ref:
            scale = float(g_ref[k].abs().max())
            np.testing.assert_allclose(g[k].numpy(), g_ref[k].numpy(), rtol=2e-4, atol=2e-6 * scale + 1e-9, err_msg=k)


def test_miou_kernel(amd, golden_dir):
    z = _load(golden_dir, "miou.npz")
    dev = torch.device("cuda:0")
    for i in range(int(z["n"])):
        o, t = torch.from_numpy(z[f"o{i}"]).reshape(1, -1).to(dev), torch.from_numpy(z[f"t{i}"]).reshape(1, -1).to(dev)
        got = float(amd.miou(o, t, invert=True)[0])
        assert got == pytest.approx(float(z[f"iou{i}"]), abs=1e-7), i


def test_large_grid_properties(amd):
    """256x256 (BASELINE size): size-independent properties - determinism, batch independence, linearity of the
    gradient in the loss coefficient, clamp invariant."""
    dev = torch.device("cuda:0")
    spec = amd.IcnnSpec(130, 2, 1)
    torch.manual_seed(1)
    p = {k: (torch.rand(shp) - 0.45) * 0.3 for k, shp in spec.keys_shapes()}
    flat1 = amd.pack_state_dict(spec, p, dev)[None]
    grid = amd.Grid.linspace(256, 256, dev)
    yy, xx = torch.meshgrid(torch.arange(256), torch.arange(256), indexing="ij")
    un = (((yy - 120) ** 2 + (xx - 140) ** 2) > 60 ** 2).float().reshape(1, -1).to(dev)
    l1, g1 = amd.loss_grad(spec, flat1, grid, un)
    l2, g2 = amd.loss_grad(spec, flat1, grid, un)
    assert torch.equal(l1, l2) and torch.equal(g1, g2)  # bitwise reproducible (no atomics)
    # the same image twice in a batch behaves like two independent single fits
    lb, gb = amd.loss_grad(spec, flat1.repeat(2, 1), grid, un.repeat(2, 1))
    np.testing.assert_allclose(gb[0].cpu().numpy(), gb[1].cpu().numpy(), rtol=0, atol=0)
    np.testing.assert_allclose(gb[0].cpu().numpy(), g1[0].cpu().numpy(), rtol=1e-4, atol=1e-7)
    # gradient is linear in the per-class coefficients
    la, ga = amd.loss_grad(spec, flat1, grid, un, weight_mode="explicit", c_fg=1e-5, c_bg=2e-5)
    lc, gc = amd.loss_grad(spec, flat1, grid, un, weight_mode="explicit", c_fg=3e-5, c_bg=6e-5)
    np.testing.assert_allclose(gc.cpu().numpy(), 3 * ga.cpu().numpy(), rtol=1e-4, atol=1e-9)
    # oracle agreement at full size for loss + gradient
    l_ref, g_ref = O.loss_and_grads(p, O.positional_grid(256, 256)[None], un.cpu().reshape(1, 1, 256, 256))
    assert float(l1[0]) == pytest.approx(l_ref, rel=2e-5)
    g = amd.unpack_params(spec, g1[0].cpu())
    for k in g_ref:
        scale = float(g_ref[k].abs().max())
        np.testing.assert_allclose(g[k].numpy(), g_ref[k].numpy(), rtol=5e-4, atol=5e-6 * scale + 1e-10, err_msg=k)
    # 20 steps: clamp invariant + loss decreases
    params = flat1.clone()
    res = amd.fit(spec, params, grid, un, 20, lr=2e-3)
    got = amd.unpack_params(spec, res.params[0].cpu())
    for k in spec.clamp_keys():
        assert float(got[k].min()) >= 0.0
    h = res.loss_hist[0].cpu().numpy()
    assert np.isfinite(h).all() and h[-1] < h[0]


def test_edge_sizes_tiny_and_many_images(amd):
    """N < one chunk (also N = 1), and more images than compute units (one workgroup per image)."""
    dev = torch.device("cuda:0")
    spec = amd.IcnnSpec(32, 2, 1)
    torch.manual_seed(2)
    for n_img, N in [(1, 1), (2, 5), (300, 37)]:
        ps = [{k: (torch.rand(shp) - 0.4) * 0.6 for k, shp in spec.keys_shapes()} for _ in range(min(n_img, 3))]
        flat = torch.stack([amd.pack_state_dict(spec, ps[i % len(ps)]) for i in range(n_img)]).to(dev)
        coords = torch.rand(n_img, 2, N)
        un = torch.rand(n_img, N)
        grid = amd.Grid.explicit(coords.to(dev))
        logits = amd.forward(spec, flat, grid).cpu()
        loss, grads = amd.loss_grad(spec, flat, grid, un.to(dev))
        for i in ([0, n_img - 1] if n_img > 1 else [0]):
            p = ps[i % len(ps)]
            g4 = coords[i].reshape(1, 2, 1, N)
            ref = O.icnn_forward_image(p, g4).reshape(-1)
            np.testing.assert_allclose(logits[i].numpy(), ref.numpy(), atol=5e-6, rtol=1e-5)
            l_ref, g_ref = O.loss_and_grads(p, g4, un[i].reshape(1, 1, 1, N))
            assert float(loss[i]) == pytest.approx(l_ref, rel=2e-5)
            g = amd.unpack_params(spec, grads[i].cpu())
            for k in g_ref:
                sc = float(g_ref[k].abs().max())
                np.testing.assert_allclose(g[k].numpy(), g_ref[k].numpy(), rtol=3e-4, atol=3e-6 * sc + 1e-9, err_msg=k)


def test_spatio_temporal_grid_c4(amd):
    """BASELINE configs[3] shape: ONE network over a 128x128x16 (x, y, t) volume (path_connected_net.py:511-728 fits one
    prior for all frames): N = 262144 points, 3 coordinate channels, explicit planar grid.  Size-independent properties
    plus oracle agreement of loss and gradient."""
    dev = torch.device("cuda:0")
    spec = amd.IcnnSpec(130, 3, 1)
    torch.manual_seed(8)
    p = {k: (torch.rand(shp) - 0.45) * 0.3 for k, shp in spec.keys_shapes()}
    T, H, W = 16, 128, 128
    frames = [O.positional_grid(W, H, t=float(t), t_max=float(T - 1)) for t in range(T)]     # (3,H,W) each
    coords = torch.stack(frames, 1).reshape(3, T * H * W)                                        # planar [C][N]
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    un = torch.stack([(((yy - 60) ** 2 + (xx - 40 - 2 * t) ** 2) > 25 ** 2).float() for t in range(T)]).reshape(1, -1)
    flat = amd.pack_state_dict(spec, p, dev)[None]
    grid = amd.Grid.explicit(coords.to(dev))
    l1, g1 = amd.loss_grad(spec, flat, grid, un.to(dev), loss="bce", weight_mode="sssdms")
    l2, g2 = amd.loss_grad(spec, flat, grid, un.to(dev), loss="bce", weight_mode="sssdms")
    assert torch.equal(l1, l2) and torch.equal(g1, g2)                      # reproducible at 4096 chunks
    g4 = coords.reshape(1, 3, 1, -1)
    l_ref, g_ref = O.loss_and_grads(p, g4, un.reshape(1, 1, 1, -1), "bce", "sssdms")
    assert float(l1[0]) == pytest.approx(l_ref, rel=3e-5)
    g = amd.unpack_params(spec, g1[0].cpu())
    for k in g_ref:
        sc = float(g_ref[k].abs().max())
        np.testing.assert_allclose(g[k].numpy(), g_ref[k].numpy(), rtol=1e-3, atol=1e-5 * sc + 1e-10, err_msg=k)
    res = amd.fit(spec, flat.clone(), grid, un.to(dev), 30, lr=1e-3, optimizer="adamax", loss="bce", weight_mode="sssdms",
                  plateau=dict(patience=200, factor=0.5))
    h = res.loss_hist[0].cpu().numpy()
    assert np.isfinite(h).all() and h[-1] < h[0] and int(res.status[0]) == 0


def test_pack_masks(amd):
    """Bit-packed masks: bit i of word w = values[w*64+i] > thr (ragged tail zero-filled), both polarities."""
    A = amd
    torch.manual_seed(0)
    v = torch.rand(3, 1000).to("cuda:0")
    for invert in (False, True):
        bits = A.pack_masks(v, 0.5, invert=invert).cpu().numpy().view(np.uint64)
        ref = (v.cpu().numpy() > 0.5) ^ invert
        pad = np.zeros((3, 1024), dtype=bool)
        pad[:, :1000] = ref
        want = (pad.reshape(3, 16, 64).astype(np.uint64) << np.arange(64, dtype=np.uint64)).sum(-1, dtype=np.uint64)
        np.testing.assert_array_equal(bits, want)


@pytest.mark.parametrize("layers,C", [(1, 2), (2, 2), (2, 3)])
def test_gradient_accuracy_against_float64(amd, layers, C):
    """How exact is the MFMA path?  The same loss/gradient in float64 is the yardstick; the fp32 torch restatement is the
    competitor.  v_mfma_f32_16x16x4_f32 is an exact-fp32 FMA chain, so the HIP gradients must be as close to the float64 result
    as fp32 torch is (x3 + floor) - at the BASELINE grid size of 256x256 for the headline model."""
    A = amd
    from awesome_amd.model import ConvexNextNet
    S = 256 if (layers, C) == (1, 2) else 96
    torch.manual_seed(5)
    m = ConvexNextNet(n_hidden=130, n_hidden_layers=layers, in_features=C)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    grid = O.positional_grid(S, S) if C == 2 else O.positional_grid(S, S, 0.3, 1.0)
    un = torch.from_numpy(np.random.RandomState(0).rand(1, 1, S, S).astype(np.float32))
    _, g32 = O.loss_and_grads(sd, grid[None], un, "bce")
    _, g64 = O.loss_and_grads({k: v.double() for k, v in sd.items()}, grid[None].double(), un.double(), "bce")
    spec = A.IcnnSpec(130, C, layers)
    params = A.pack_state_dict(spec, sd, "cuda:0")[None].contiguous()
    _, g = A.loss_grad(spec, params, A.Grid.from_image_grid(grid.to("cuda:0")), un.reshape(1, -1).to("cuda:0"), loss="bce")
    got = A.unpack_params(spec, g[0].cpu())
    e_hip, e_ref = [], []
    for k in g64:
        sc = float(g64[k].abs().max()) + 1e-30
        e_hip.append(float((got[k].double() - g64[k]).abs().max()) / sc)
        e_ref.append(float((g32[k].double() - g64[k]).abs().max()) / sc)
    assert max(e_hip) <= 3.0 * max(e_ref) + 1e-6, (max(e_hip), max(e_ref))


@pytest.mark.parametrize("h,C,layers", [(130, 2, 1), (130, 3, 1), (64, 2, 1), (64, 3, 1), (32, 2, 1), (32, 3, 1),
                                         (130, 2, 2), (130, 3, 2), (64, 2, 2), (64, 3, 2)])
def test_every_compiled_shape_every_parameter_gradient(amd, h, C, layers):
    """All ten kernel entries of the library on a ragged grid: the loss and the gradient of EVERY parameter against the oracle - in
    particular every gradient-slab column reaches its parameter (the slabs are stored in accumulator-tile order and mapped back by
    the update kernel, icnn_step.h `slab_param_of_col`), for widths with and without leftover units - and one clamped Adam step."""
    A = amd
    from awesome_amd.model import ConvexNextNet
    torch.manual_seed(17 + h + C + layers)
    m = ConvexNextNet(n_hidden=h, n_hidden_layers=layers, in_features=C)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    H_, W_ = 19, 23     # N = 437: 6 full chunks + a ragged one
    grid = O.positional_grid(W_, H_) if C == 2 else O.positional_grid(W_, H_, 0.4, 1.0)
    un = torch.from_numpy(np.random.RandomState(h + C).rand(1, 1, H_, W_).astype(np.float32))
    lo, go = O.loss_and_grads(sd, grid[None], un, "se")
    spec = A.IcnnSpec(h, C, layers)
    params = A.pack_state_dict(spec, sd, "cuda:0")[None].contiguous()
    g_ = A.Grid.from_image_grid(grid.to("cuda:0"))
    loss, g = A.loss_grad(spec, params, g_, un.reshape(1, -1).to("cuda:0"), loss="se")
    assert float(loss[0]) == pytest.approx(float(lo), rel=2e-5)
    got = A.unpack_params(spec, g[0].cpu())
    assert set(got) == set(go)
    for k, ref in go.items():
        ref = ref.numpy()
        np.testing.assert_allclose(got[k].numpy(), ref, rtol=2e-4, atol=2e-4 * float(np.abs(ref).max()) + 1e-10, err_msg=k)
    # one Adam step with the convexity clamp through the fused fit: every parameter moves like the oracle's
    res = A.fit(spec, params.clone(), g_, un.reshape(1, -1).to("cuda:0"), 1, lr=1e-2, loss="se", optimizer="adam")
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(p.values()), lr=1e-2)
    O.weighted_loss(torch.sigmoid(O.icnn_forward_image(p, grid[None])), un, "se").backward()
    opt.step()
    O.icnn_enforce_convexity(p)
    new = A.unpack_params(spec, res.params[0].cpu())
    for k, v in p.items():
        np.testing.assert_allclose(new[k].numpy(), v.detach().numpy(), rtol=1e-4, atol=2e-5, err_msg=k)


def test_baseline_config_full_fit_matches_reference(amd, golden_dir):
    """BASELINE configs[1] end to end at full size: 2000 full-batch Adam steps of ConvexNextNet(h=130, L=1) on the 256x256 blob,
    HIP vs the REAL reference classes' fit of the same seeded problem (tools/gen_golden.py gen_fit_blob256: final mask, loss
    curve, fg-mIoU).  north_star's tolerance: mIoU within +-1e-3."""
    A = amd
    from awesome_amd.dataset import convex_blob_unaries
    from awesome_amd.model import ConvexNextNet
    z = np.load(os.path.join(golden_dir, "fit_blob256_reference.npz"))
    S, E = 256, 2000
    torch.manual_seed(0)
    m = ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=1)
    un = convex_blob_unaries(S, 0).reshape(1, -1).to("cuda:0")
    res = A.fit(m.spec, m.flat_parameters()[None].to("cuda:0"), A.Grid.linspace(S, S, "cuda:0"), un, E, lr=2e-3)
    mask_h = (torch.sigmoid(res.logits[0]) > 0.5)
    mask_r = torch.from_numpy(z["final_mask"]).to("cuda:0")
    iou_h = float(A.miou(mask_h[None].float(), (un > 0.5).float())[0])
    assert abs(iou_h - float(z["final_miou"])) <= 1e-3, (iou_h, float(z["final_miou"]))
    assert int((mask_h != mask_r).sum()) <= 0.002 * S * S
    h = res.loss_hist[0].cpu().numpy()
    np.testing.assert_allclose(h[:100], z["losses"][:100], rtol=5e-4)          # same trajectory while rounding has not piled up
    assert abs(h[-1] - z["losses"][-1]) <= 0.1 * z["losses"][-1]               # and the same end point


def test_full_fit_two_hidden_layers_matches_reference(amd, golden_dir):
    """The same full-size fit with the two-hidden-layer net every flow prior evaluates (icnn2_step_kernel): 2000 Adam steps of
    ConvexNextNet(h=130, L=2) on the 256x256 blob vs the reference classes' fit (golden fit_blob256_l2_reference.npz)."""
    A = amd
    from awesome_amd.dataset import convex_blob_unaries
    from awesome_amd.model import ConvexNextNet
    z = np.load(os.path.join(golden_dir, "fit_blob256_l2_reference.npz"))
    S, E = 256, 2000
    torch.manual_seed(0)
    m = ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=2)
    un = convex_blob_unaries(S, 0).reshape(1, -1).to("cuda:0")
    res = A.fit(m.spec, m.flat_parameters()[None].to("cuda:0"), A.Grid.linspace(S, S, "cuda:0"), un, E, lr=2e-3)
    mask_h = (torch.sigmoid(res.logits[0]) > 0.5)
    mask_r = torch.from_numpy(z["final_mask"]).to("cuda:0")
    iou_h = float(A.miou(mask_h[None].float(), (un > 0.5).float())[0])
    assert abs(iou_h - float(z["final_miou"])) <= 1e-3, (iou_h, float(z["final_miou"]))
    assert int((mask_h != mask_r).sum()) <= 0.002 * S * S
    h = res.loss_hist[0].cpu().numpy()
    np.testing.assert_allclose(h[:100], z["losses"][:100], rtol=5e-4)          # same trajectory while rounding has not piled up
    assert abs(h[-1] - z["losses"][-1]) <= 0.1 * z["losses"][-1]               # and the same end point


def test_gate_logits_are_the_last_training_forward(amd):
    """InrOptDesc.logits_at_last_forward: the fit returns the output of its last TRAINING forward (parameters before the last
    optimizer step) - the tensor the reference's IoU gate reads (path_connected_net.py:939-972) - instead of the logits at the
    final parameters; the parameters themselves are the same either way."""
    dev = torch.device("cuda:0")
    for layers in (1, 2):
        spec = amd.IcnnSpec(130, 2, layers)
        torch.manual_seed(9)
        p = {k: (torch.rand(shp) - 0.45) * 0.3 for k, shp in spec.keys_shapes()}
        flat = amd.pack_state_dict(spec, p, dev)[None]
        grid = amd.Grid.linspace(40, 33, dev)
        un = (torch.rand(1, 40 * 33, device=dev) > 0.5).float()
        four = amd.fit(spec, flat.clone(), grid, un, 4, lr=2e-3)
        before_last = amd.forward(spec, four.params, grid)
        five_gate = amd.fit(spec, flat.clone(), grid, un, 5, lr=2e-3, gate_logits=True)
        five = amd.fit(spec, flat.clone(), grid, un, 5, lr=2e-3)
        assert torch.equal(five_gate.params, five.params)
        np.testing.assert_allclose(five_gate.logits.cpu().numpy(), before_last.cpu().numpy(), rtol=0, atol=1e-6)
        np.testing.assert_allclose(five.logits.cpu().numpy(), amd.forward(spec, five.params, grid).cpu().numpy(), rtol=0, atol=1e-6)
        assert float((five.logits - five_gate.logits).abs().max()) > 1e-5      # one optimizer step apart


@pytest.mark.parametrize("h,C,L", [(100, 2, 1), (77, 3, 1), (48, 2, 2), (100, 2, 2), (20, 2, 1)])
def test_any_hidden_width_up_to_130_runs_zero_padded(amd, h, C, L):
    """VERDICT r02 item 7 (the part the LDS-resident design allows): an n_hidden without a kernel of its own runs zero-padded on the
    next compiled width; parameters, gradients and optimizer state keep the caller's layout.  Forward, loss, every gradient and a
    25-step Adam + clamp trajectory against the oracle at the model's OWN shape."""
    A, dev = amd, torch.device("cuda:0")
    torch.manual_seed(h + C + L)
    spec = A.IcnnSpec(n_hidden=h, in_features=C, n_layers=L)
    assert spec.supported()
    p = {k: (torch.rand(shp) - 0.45) * 0.3 for k, shp in spec.keys_shapes()}
    H, W = 24, 20
    grid_t = O.positional_grid(W, H) if C == 2 else O.positional_grid(W, H, 2.0, 5.0)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    un = (((yy - 11) ** 2 + (xx - 9) ** 2) > 36).float()
    flat = A.pack_state_dict(spec, p, dev)[None].contiguous()
    grid = A.Grid.from_image_grid(grid_t[None].to(dev)) if C == 3 else A.Grid.linspace(W, H, dev)
    logits = A.forward(spec, flat, grid)
    np.testing.assert_allclose(logits[0].cpu().numpy(), O.icnn_forward_image(p, grid_t[None]).reshape(-1).numpy(), atol=5e-6, rtol=1e-5)
    loss, grads = A.loss_grad(spec, flat, grid, un.reshape(1, -1).to(dev), loss="bce")
    lo, go = O.loss_and_grads(p, grid_t[None], un[None, None], "bce")
    assert float(loss[0]) == pytest.approx(lo, rel=2e-5)
    got = A.unpack_params(spec, grads[0].cpu())
    for k in go:
        np.testing.assert_allclose(got[k].numpy(), go[k].numpy(), rtol=2e-4, atol=2e-6 * float(go[k].abs().max()) + 1e-9, err_msg=k)
    pf, losses, _ = O.fit_icnn(p, grid_t[None], un[None, None], 25, lr=2e-3)
    res = A.fit(spec, flat.clone(), grid, un.reshape(1, -1).to(dev), 25, lr=2e-3)
    np.testing.assert_allclose(res.loss_hist[0].cpu().numpy(), np.asarray(losses, np.float32), rtol=2e-4)
    gotp = A.unpack_params(spec, res.params[0].cpu())
    for k in pf:
        np.testing.assert_allclose(gotp[k].numpy(), pf[k].numpy(), rtol=5e-4, atol=2e-6, err_msg=k)


@pytest.mark.parametrize("h,C,L,act0,hw", [(256, 2, 1, "relu", (20, 24)), (350, 2, 3, "relu", (20, 24)), (160, 3, 2, "relu", (20, 24)),
                                           (64, 2, 4, "relu", (20, 24)), (200, 2, 2, "cos", (20, 24)), (131, 2, 3, "relu", (40, 50)),
                                           (256, 3, 2, "relu", (33, 47)), (600, 2, 1, "relu", (20, 24)), (72, 2, 3, "sin", (20, 24))])
def test_wide_and_deep_shapes_on_the_layer_by_layer_path(amd, h, C, L, act0, hw):
    """n_hidden > 130 (and more than two hidden layers) have no fused kernel - the weight image of such a layer does
    not fit the LDS - and run layer by layer (awesome_amd/csrc/wide.h: activations in HBM, plain GEMMs, the same update kernel).  Forward,
    loss, every gradient and a 12-step Adam + clamp trajectory against the oracle; 350 x 3 is the relu stack of
    notebooks/imageRepresentationTest.ipynb cell 5.  The two larger grids (2000 / 1551 points) have several chunks of the weight
    gradients' split contraction, enough tiles for the XCD-aware tile order, a ragged last row tile and an odd width (131); width 600
    has rows too long for wide_out_kernel's (1, x) accumulators: the last layer's (db | dS) take wide_extgrad_kernel's pass."""
    A, dev = amd, torch.device("cuda:0")
    torch.manual_seed(h + L)
    omega = 3.0 if act0 == "sin" else 1.0
    spec = A.IcnnSpec(n_hidden=h, in_features=C, n_layers=L, act0=act0, omega=omega) if act0 != "relu" else A.IcnnSpec(n_hidden=h, in_features=C, n_layers=L)
    assert spec.supported()
    p = {k: (torch.rand(shp) - 0.45) * (0.6 / np.sqrt(h)) for k, shp in spec.keys_shapes()}
    H, W = hw
    grid_t = O.positional_grid(W, H) if C == 2 else O.positional_grid(W, H, 2.0, 5.0)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    un = (((yy - 0.45 * H) ** 2 + (xx - 0.46 * W) ** 2) > 0.0625 * H * W).float()
    flat = A.pack_state_dict(spec, p, dev)[None].contiguous()
    grid = A.Grid.from_image_grid(grid_t[None].to(dev)) if C == 3 else A.Grid.linspace(W, H, dev)
    logits = A.forward(spec, flat, grid)
    ref = O.icnn_forward_image(p, grid_t[None], act0=act0, omega=omega)
    np.testing.assert_allclose(logits[0].cpu().numpy(), ref.reshape(-1).numpy(), atol=2e-5, rtol=2e-5)
    if act0 != "relu":
        # a periodic layer 0: the backward GEMM's mask epilogue multiplies by the activation's derivative at the kept pre-activation
        dl = torch.randn(1, H * W)
        gb = A.icnn.backward(spec, flat, grid, dl.to(dev))
        pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        (O.icnn_forward_image(pr, grid_t[None], act0=act0, omega=omega).reshape(-1) * dl[0]).sum().backward()
        gotb = A.unpack_params(spec, gb[0].cpu())
        for k in pr:
            np.testing.assert_allclose(gotb[k].numpy(), pr[k].grad.numpy(), rtol=1e-3, atol=2e-5 * float(pr[k].grad.abs().max()) + 1e-10, err_msg=k)
        return
    loss, grads = A.loss_grad(spec, flat, grid, un.reshape(1, -1).to(dev), loss="se")
    lo, go = O.loss_and_grads(p, grid_t[None], un[None, None], "se")
    assert float(loss[0]) == pytest.approx(lo, rel=2e-5)
    got = A.unpack_params(spec, grads[0].cpu())
    for k in go:
        np.testing.assert_allclose(got[k].numpy(), go[k].numpy(), rtol=5e-4, atol=5e-6 * float(go[k].abs().max()) + 1e-10, err_msg=k)
    dl = torch.randn(1, H * W)
    gb = A.icnn.backward(spec, flat, grid, dl.to(dev))
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    (O.icnn_forward_image(pr, grid_t[None]).reshape(-1) * dl[0]).sum().backward()
    gotb = A.unpack_params(spec, gb[0].cpu())
    for k in pr:
        np.testing.assert_allclose(gotb[k].numpy(), pr[k].grad.numpy(), rtol=5e-4, atol=5e-6 * float(pr[k].grad.abs().max()) + 1e-10, err_msg=k)
    pf, losses, _ = O.fit_icnn(p, grid_t[None], un[None, None], 12, lr=2e-3)
    res = A.fit(spec, flat.clone(), grid, un.reshape(1, -1).to(dev), 12, lr=2e-3)
    np.testing.assert_allclose(res.loss_hist[0].cpu().numpy(), np.asarray(losses, np.float32), rtol=3e-4)
    gotp = A.unpack_params(spec, res.params[0].cpu())
    for k in pf:
        np.testing.assert_allclose(gotp[k].numpy(), pf[k].numpy(), rtol=1e-3, atol=5e-6, err_msg=k)
    np.testing.assert_allclose(res.logits[0].cpu().numpy(), O.icnn_forward_image(pf, grid_t[None]).reshape(-1).numpy(), atol=1e-4, rtol=1e-4)


def test_wide_path_fits_several_images_per_call(amd):
    """The layer-by-layer path with n_images = 3 in one C-ABI call (own parameters, targets and optimizer state per image, one shared
    workspace): every image's loss curve and final parameters equal its own single-image oracle fit."""
    A, dev = amd, torch.device("cuda:0")
    h, L, H, W, n = 160, 2, 20, 24, 3
    spec = A.IcnnSpec(n_hidden=h, in_features=2, n_layers=L)
    grid_t = O.positional_grid(W, H)
    grid = A.Grid.linspace(W, H, dev)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    ps, uns = [], []
    for i in range(n):
        torch.manual_seed(40 + i)
        ps.append({k: (torch.rand(shp) - 0.45) * (0.6 / np.sqrt(h)) for k, shp in spec.keys_shapes()})
        uns.append((((yy - 7 - 2 * i) ** 2 + (xx - 9 - 3 * i) ** 2) > 25 + 5 * i).float())
    flat = torch.stack([A.pack_state_dict(spec, p, dev) for p in ps]).contiguous()
    un = torch.stack([u.reshape(-1) for u in uns]).to(dev)
    res = A.fit(spec, flat.clone(), grid, un, 10, lr=2e-3)
    for i in range(n):
        pf, losses, _ = O.fit_icnn(ps[i], grid_t[None], uns[i][None, None], 10, lr=2e-3)
        np.testing.assert_allclose(res.loss_hist[i].cpu().numpy(), np.asarray(losses, np.float32), rtol=3e-4, err_msg=f"image {i}")
        got = A.unpack_params(spec, res.params[i].cpu())
        for k in pf:
            np.testing.assert_allclose(got[k].numpy(), pf[k].numpy(), rtol=1e-3, atol=5e-6, err_msg=f"image {i} {k}")
