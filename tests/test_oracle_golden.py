"""Pin the CPU oracle against golden vectors captured from the real reference classes
(tools/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import inr_oracle as O

ICNN_CASES = ["convexnet_h130_c2", "convexnext_h130_c2_l1", "convexnext_h130_c2_l2", "convexnext_h130_c3_l1",
              "convexnext_h32_c2_l1", "convexnext_h64_c3_l2"]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_grid(golden_dir):
    z = _load(golden_dir, "grid.npz")
    assert np.array_equal(O.positional_grid(7, 5).numpy(), z["g_7x5"])
    assert np.array_equal(O.positional_grid(64, 64).numpy(), z["g_64x64"])
    assert np.array_equal(O.positional_grid(256, 256).numpy()[0, 0], z["g_256x256_row0"])
    assert np.array_equal(O.positional_grid(6, 4, t=3.0, t_max=15.0).numpy(), z["g_6x4_t"])


def test_miou(golden_dir):
    z = _load(golden_dir, "miou.npz")
    for i in range(int(z["n"])):
        got = O.miou_binary(torch.from_numpy(z[f"o{i}"]), torch.from_numpy(z[f"t{i}"]), invert=True)
        assert abs(got - float(z[f"iou{i}"])) < 1e-7, i


def test_weighted_losses(golden_dir):
    z = _load(golden_dir, "losses.npz")
    out, tgt = torch.from_numpy(z["output"]), torch.from_numpy(z["target"])
    for mode in ["none", "equal", "ratio", "sssdms"]:
        for kind in ["se", "bce"]:
            got = float(O.weighted_loss(out, tgt, kind, mode, ratio=0.35))
            assert got == pytest.approx(float(z[f"uwl.{kind}.{mode}"]), rel=1e-6), (mode, kind)
    out2, tb = torch.from_numpy(z["output2"]), torch.from_numpy(z["target_bin"])
    assert float(O.awesome_image_loss(out2, tb, alpha=0.7)) == pytest.approx(float(z["ail.plain"]), rel=1e-6)
    assert float(O.awesome_image_loss(out2, tb, alpha=0.7, extra_penalty=True)) == pytest.approx(float(z["ail.penalty"]), rel=1e-6)


@pytest.mark.parametrize("name", ICNN_CASES)
def test_icnn_forward_loss_grads(golden_dir, name):
    z = _load(golden_dir, f"icnn_{name}.npz")
    sd_raw = O.load_npz_state(z, "sd0.")
    p = O.to_convexnext_keys(sd_raw)
    grid, un = torch.from_numpy(z["grid"]), torch.from_numpy(z["unaries"])
    logits = O.icnn_forward_image(p, grid)
    np.testing.assert_allclose(logits.numpy(), z["logits"], rtol=0, atol=1e-6)
    for tag, kind, mode, target in [("se", "se", "none", un), ("bce", "bce", "none", un),
                                    ("sssdms", "se", "sssdms", (un >= 0.5).float())]:
        loss, grads = O.loss_and_grads(p, grid, target, kind, mode)
        assert loss == pytest.approx(float(z[f"{tag}.loss"]), rel=2e-6)
        for k_raw in sd_raw:
            k = O.CONVEXNET_KEYMAP.get(k_raw, k_raw) if "W0y.weight" in sd_raw else k_raw
            ref = z[f"{tag}.grad.{k_raw}"]
            np.testing.assert_allclose(grads[k].numpy(), ref, rtol=1e-5, atol=1e-8, err_msg=f"{tag} {k_raw}")


@pytest.mark.parametrize("name", ICNN_CASES)
def test_icnn_adam_clamp_trajectory(golden_dir, name):
    z = _load(golden_dir, f"icnn_{name}.npz")
    sd_raw = O.load_npz_state(z, "sd0.")
    is_cn = "W0y.weight" in sd_raw
    p = O.to_convexnext_keys(sd_raw)
    grid, un = torch.from_numpy(z["grid"]), torch.from_numpy(z["unaries"])
    for steps, tag in [(1, "adam1."), (10, "adam10.")]:
        pf, losses, _ = O.fit_icnn(p, grid, un, steps, lr=2e-3)
        np.testing.assert_allclose(np.asarray(losses, np.float32), z["adam.losses"][:steps], rtol=1e-5)
        for k_raw in sd_raw:
            k = O.CONVEXNET_KEYMAP[k_raw] if is_cn else k_raw
            np.testing.assert_allclose(pf[k].numpy(), z[tag + k_raw], rtol=1e-4, atol=2e-6, err_msg=f"{tag}{k_raw}")
    for k in O.icnn_clamp_keys(pf):
        assert float(pf[k].min()) >= 0.0


def test_adamax_plateau(golden_dir):
    z = _load(golden_dir, "adamax_plateau_h32.npz")
    p = O.load_npz_state(z, "sd0.")
    grid, un = torch.from_numpy(z["grid"]), torch.from_numpy(z["unaries"])
    n = len(z["losses"])
    pf, losses, _ = O.fit_icnn(p, grid, un, n, lr=1e-2, optimizer="adamax", weight_decay=1e-5,
                               plateau=dict(patience=5, factor=0.5))
    np.testing.assert_allclose(np.asarray(losses, np.float32), z["losses"], rtol=2e-4)
    for k in p:
        np.testing.assert_allclose(pf[k].numpy(), z["final." + k], rtol=2e-3, atol=2e-5, err_msg=k)


def test_plateau_lr_schedule(golden_dir):
    z = _load(golden_dir, "adamax_plateau_h32.npz")
    s = O.PlateauState(1e-2, patience=5, factor=0.5)
    lrs = [s.step(float(l)) for l in z["losses"]]
    np.testing.assert_allclose(np.asarray(lrs), z["lrs"], rtol=1e-12)


def test_flow(golden_dir):
    z = _load(golden_dir, "flow.npz")
    sd = O.load_npz_state(z, "wn.sd.")
    np.testing.assert_allclose(O.wn_linear(sd, "", torch.from_numpy(z["wn.x"])).numpy(), z["wn.y"], rtol=1e-6, atol=1e-7)
    sd = O.load_npz_state(z, "nb.sd.")
    np.testing.assert_allclose(O.normal_block(sd, "", torch.from_numpy(z["nb.x"])).numpy(), z["nb.y"], rtol=1e-6, atol=1e-7)
    sd = O.load_npz_state(z, "sc.sd.")
    np.testing.assert_allclose(O.wn_scale(sd, "").numpy(), z["sc.y"], rtol=1e-6)
    x = torch.from_numpy(z["x"])
    for tag, nc in [("nf6_w130", 6), ("nf4_w16", 4)]:
        sd = {k: v.requires_grad_(True) for k, v in O.load_npz_state(z, f"{tag}.sd.").items()}
        y = O.flow1d_forward(sd, x, nc)
        np.testing.assert_allclose(y.detach().numpy(), z[f"{tag}.y"], rtol=1e-5, atol=1e-6)
        (y ** 2).mean().backward()
        for k, v in sd.items():
            np.testing.assert_allclose(v.grad.numpy(), z[f"{tag}.grad.{k}"], rtol=1e-4, atol=1e-7, err_msg=k)


def test_fit_disc64_end_to_end(golden_dir):
    """600-step fit of the 64x64 disc reproduces the reference loss curve, mask and mIoU."""
    z = _load(golden_dir, "fit_disc64.npz")
    p = O.load_npz_state(z, "sd0.")
    grid = O.positional_grid(64, 64)[None]
    un = torch.from_numpy(z["unaries"])
    pf, losses, logits = O.fit_icnn(p, grid, un, 600, lr=2e-3)
    np.testing.assert_allclose(np.asarray(losses[:50], np.float32), z["losses"][:50], rtol=1e-4)
    assert losses[-1] == pytest.approx(float(z["losses"][-1]), rel=5e-2)
    mask = (torch.sigmoid(logits) > 0.5)
    miou = O.miou_binary(mask.float(), (un > 0.5).float(), invert=True)
    assert abs(miou - float(z["final_miou"])) <= 1e-3
    assert (mask.numpy() != z["final_mask"]).mean() < 2e-3


def test_minmax_transform(golden_dir):
    """MinMax around the RealNVP flow (awesome/transforms/min_max.py:8-58) - the one piece of the a10 path that is the
    reference's own code besides the ICNN."""
    z = np.load(os.path.join(golden_dir, "minmax.npz"))
    for C in (2, 3):
        vmin, vmax = torch.from_numpy(z[f"c{C}.min"]), torch.from_numpy(z[f"c{C}.max"])
        fit_x = torch.from_numpy(z[f"c{C}.fit_x"])
        np.testing.assert_array_equal(vmin.numpy(), fit_x.amin(dim=(0, 2, 3), keepdim=True).numpy())
        np.testing.assert_array_equal(vmax.numpy(), fit_x.amax(dim=(0, 2, 3), keepdim=True).numpy())
        x = torch.from_numpy(z[f"c{C}.x"])
        np.testing.assert_array_equal(O.minmax(x, vmin, vmax, -1.0, 1.0).numpy(), z[f"c{C}.transform"])
        np.testing.assert_array_equal(O.minmax(x, -1.0, 1.0, vmin, vmax).numpy(), z[f"c{C}.inverse"])


def test_awesome_loss_pixel_mode(golden_dir):
    z = np.load(os.path.join(golden_dir, "pixel_losses.npz"))
    out, tgt = torch.from_numpy(z["al.output"]), torch.from_numpy(z["al.target"])
    assert float(O.awesome_loss(out, tgt, 0.6, 0.75)) == pytest.approx(float(z["al.plain"]), rel=1e-6)
    assert float(O.awesome_loss(out, tgt, 0.6, 0.75, extra_penalty=True)) == pytest.approx(float(z["al.penalty"]), rel=1e-6)
    from awesome_amd.measures import AwesomeLoss
    crit = AwesomeLoss(alpha=0.6, scribble_percentage=0.75)
    assert float(crit(out, tgt)) == pytest.approx(float(z["al.plain"]), rel=1e-6)
    crit.extra_penalty = True
    assert float(crit(out, tgt)) == pytest.approx(float(z["al.penalty"]), rel=1e-6)


@pytest.mark.parametrize("name", ["fcnet_w130_d1", "fcnet_w64_d2"])
def test_fcnet_no_prior_network(golden_dir, name):
    """FCNet(in_type='xy'): the oracle's forward, its ICNN-with-zero-skips form, gradients and a 10-step Adam trajectory
    (no clamp) against the reference class."""
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    sd = O.load_npz_state(z, "sd.")
    rows, un = torch.from_numpy(z["rows"]), torch.from_numpy(z["unaries"])
    np.testing.assert_allclose(O.fcnet_forward(sd, rows).numpy(), z["logits"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(O.icnn_forward(O.fcnet_as_icnn(sd), rows).numpy(), z["logits"], rtol=1e-5, atol=1e-6)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss = ((un - torch.sigmoid(O.fcnet_forward(p, rows))) ** 2).mean()
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(z["loss"]), rel=1e-6)
    for k in p:
        np.testing.assert_allclose(p[k].grad.numpy(), z["grad." + k], rtol=1e-4, atol=1e-7, err_msg=k)
    st = O.AdamState(p)
    losses = []
    for _ in range(10):
        for v in p.values():
            v.grad = None
        l = ((un - torch.sigmoid(O.fcnet_forward(p, rows))) ** 2).mean()
        l.backward()
        O.adam_step(p, {k: v.grad for k, v in p.items()}, st, 2e-3)
        losses.append(float(l.detach()))
    np.testing.assert_allclose(np.asarray(losses, np.float32), z["adam10.losses"], rtol=1e-5)
    for k in p:
        np.testing.assert_allclose(p[k].detach().numpy(), z["adam10." + k], rtol=1e-4, atol=1e-6, err_msg=k)


def test_full_size_fit_oracle_vs_reference(golden_dir):
    """BASELINE configs[1], 2000 steps at 256x256: the oracle's fit (cached by tests/manual_parity_c2.py, ~2 min of CPU) against
    the real reference classes' fit of the same seeded problem (tools/gen_golden.py gen_fit_blob256)."""
    o = np.load(os.path.join(golden_dir, "c2_oracle_fit2000.npz"))
    r = np.load(os.path.join(golden_dir, "fit_blob256_reference.npz"))
    np.testing.assert_allclose(o["losses"][:200], r["losses"][:200], rtol=1e-4)
    assert abs(o["losses"][-1] - r["losses"][-1]) <= 0.02 * r["losses"][-1]
    assert int((o["mask"].astype(bool) != r["final_mask"].astype(bool)).sum()) <= 30   # of 65 536 pixels


def test_cdn_fit_end_to_end(golden_dir):
    """ConvexNextNet(L=2) o NormalizingFlow1D(6 x 130) o Linear(2,2), 300 Adam steps with the weight_g parameter group
    (golden cdn_fit48.npz: the reference's own modules composed like ConvexDiffeomorphismNet.forward): the oracle's fused loop."""
    z = np.load(os.path.join(golden_dir, "cdn_fit48.npz"))
    sd0 = O.load_npz_state(z, "sd0.")
    un = torch.from_numpy(z["unaries"])
    S = un.shape[-1]
    pf, losses, logits = O.fit_convex_diffeo(sd0, O.positional_grid(S, S)[None], un, 300, 6, lr=3e-3, loss_kind="bce",
                                             weight_decay_on_weight_g=5e-5)
    # identical to 1e-7 for 20 steps and 1e-5 for 60; afterwards the (still unconverged, lr 3e-3) trajectories drift apart the
    # way any two fp32 summation orders do - the end point is compared loosely
    np.testing.assert_allclose(np.asarray(losses[:60], np.float32), z["losses"][:60], rtol=1e-4)
    assert abs(losses[-1] - float(z["losses"][-1])) <= 0.05 * float(z["losses"][-1])
    agree = ((logits.reshape(-1) > 0) == (torch.from_numpy(z["final_logits"]).reshape(-1) > 0)).float().mean()
    assert float(agree) > 0.95
