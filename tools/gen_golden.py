#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference classes (CPU, this container only).

This script imports the hot-path classes of jp-schneider/awesome from
/root/reference (read-only) and records inputs + outputs as small .npz fixtures
under tests/golden/.  It never runs on the GPU box (the reference does not
travel); only the fixtures do.  Nothing in the product imports this file.

Import recipe: `awesome/model/__init__.py` eagerly pulls in
modules that need `toml`, so an empty `awesome.model` package object is
registered first; the classes used here then import with no stubs.

Usage:  python tools/gen_golden.py [--out tests/golden]
"""
import argparse
import os
import random
import sys
import types

import numpy as np
import torch

REF = "/root/reference"


def _import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference checkout not present; fixtures can only be generated in the build container")
    sys.path.insert(0, REF)
    for sub in ("model", "dataset"):  # skip the eager package __init__ files (they need toml/cv2)
        pkg = types.ModuleType(f"awesome.{sub}")
        pkg.__path__ = [os.path.join(REF, "awesome", sub)]
        sys.modules[f"awesome.{sub}"] = pkg
    import awesome.model.convex_net as convex_net
    import awesome.model.diffeomorphism_net as diffeo
    import awesome.model.fc_net as fc_net
    import awesome.model.real_nvp.resnet_1d as resnet_1d
    import awesome.measures.se as se
    import awesome.measures.unaries_weighted_loss as uwl
    import awesome.measures.miou as miou
    import awesome.measures.awesome_image_loss as ail
    import awesome.measures.awesome_loss as al
    import awesome.dataset.transformator as transformator
    import awesome.transforms.min_max as min_max
    import awesome.util.torch as autil
    return types.SimpleNamespace(convex_net=convex_net, diffeo=diffeo, fc_net=fc_net, resnet_1d=resnet_1d, se=se, uwl=uwl,
                                 miou=miou, ail=ail, al=al, autil=autil, transformator=transformator, min_max=min_max)


def seed_all(seed: int) -> None:
    # same calls as awesome/run/runner.py:19-25 (CPU part)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def sd_np(module, prefix="sd."):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def grads_np(module, prefix="grad."):
    return {prefix + k: p.grad.detach().cpu().numpy().copy() for k, p in module.named_parameters()}


def disc_unaries(h, w, cy, cx, r):
    yy, xx = np.mgrid[0:h, 0:w]
    disc = ((yy - cy) ** 2 + (xx - cx) ** 2) <= r * r
    return (1.0 - disc.astype(np.float32))  # fg = 0, bg = 1


def blob_unaries(h, w, seed):
    """Smooth random blob with soft (non-binary) unaries in (0,1)."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    cy, cx = rng.uniform(0.35, 0.65, 2) * np.array([h, w])
    ry, rx = rng.uniform(0.15, 0.3, 2) * np.array([h, w])
    d = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2
    u = 1.0 / (1.0 + np.exp(-(d - 1.0) * 4.0))
    return u.astype(np.float32)


def gen_icnn(ref, out):
    """a3/a4/a5/a11/a14: forward, loss, grads, Adam(+clamp) for the ICNN family."""
    T = ref.transformator.Transformator
    cases = [
        ("convexnet_h130_c2", lambda: ref.convex_net.ConvexNet(n_hidden=130, in_channels=2), 2, 16, 16),
        ("convexnext_h130_c2_l1", lambda: ref.convex_net.ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=1), 2, 16, 24),
        ("convexnext_h130_c2_l2", lambda: ref.convex_net.ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=2), 2, 16, 16),
        ("convexnext_h130_c3_l1", lambda: ref.convex_net.ConvexNextNet(n_hidden=130, in_features=3, n_hidden_layers=1), 3, 16, 16),
        ("convexnext_h32_c2_l1", lambda: ref.convex_net.ConvexNextNet(n_hidden=32, in_features=2, n_hidden_layers=1), 2, 20, 12),
        ("convexnext_h64_c3_l2", lambda: ref.convex_net.ConvexNextNet(n_hidden=64, in_features=3, n_hidden_layers=2), 3, 8, 8),
    ]
    for ci, (name, ctor, C, H, W) in enumerate(cases):
        seed_all(100 + ci)
        model = ctor()
        if C == 2:
            grid = T.get_positional_matrices(W, H)
        else:
            grid = T.get_positional_matrices(W, H, t=3.0, t_max=7.0)
        grid = grid[None]  # (1,C,H,W)
        unaries = torch.from_numpy(blob_unaries(H, W, 7 + ci))[None, None]
        rec = {"grid": grid.numpy(), "unaries": unaries.numpy()}
        rec.update(sd_np(model, "sd0."))
        # forward
        with torch.no_grad():
            rec["logits"] = model(grid).numpy()
        # SE(sigmoid) mean loss + grads (the loss of _prior_based_pretrain, path_connected_net.py:943-951)
        crit = ref.uwl.UnariesWeightedLoss(ref.se.SE("mean"))
        model.zero_grad()
        loss = crit(torch.sigmoid(model(grid)), unaries)
        loss.backward()
        rec["se.loss"] = np.float32(loss.item())
        rec.update(grads_np(model, "se.grad."))
        # BCE(sigmoid) mean loss + grads
        model.zero_grad()
        loss = torch.nn.BCELoss()(torch.sigmoid(model(grid)), unaries)
        loss.backward()
        rec["bce.loss"] = np.float32(loss.item())
        rec.update(grads_np(model, "bce.grad."))
        # SE + sssdms re-weighting on binary unaries
        ub = (unaries >= 0.5).float()
        crit = ref.uwl.UnariesWeightedLoss(ref.se.SE("mean"), mode="sssdms")
        model.zero_grad()
        loss = crit(torch.sigmoid(model(grid)), ub)
        loss.backward()
        rec["sssdms.loss"] = np.float32(loss.item())
        rec.update(grads_np(model, "sssdms.grad."))
        # Adam + clamp trajectory (SE mean), snapshots after 1 and 10 steps
        opt = torch.optim.Adam(model.parameters(), lr=2e-3)
        crit = ref.uwl.UnariesWeightedLoss(ref.se.SE("mean"))
        losses = []
        for step in range(10):
            opt.zero_grad()
            loss = crit(torch.sigmoid(model(grid)), unaries)
            loss.backward()
            opt.step()
            model.enforce_convexity()
            losses.append(loss.item())
            if step == 0:
                rec.update(sd_np(model, "adam1."))
        rec.update(sd_np(model, "adam10."))
        rec["adam.losses"] = np.asarray(losses, dtype=np.float32)
        np.savez_compressed(os.path.join(out, f"icnn_{name}.npz"), **rec)
        print("wrote", name, "loss", losses[0], "->", losses[-1])


def gen_adamax(ref, out):
    """a13/a14: Adamax + ReduceLROnPlateau + clamp as in _prior_based_pretrain (path_connected_net.py:924-951)."""
    T = ref.transformator.Transformator
    seed_all(7)
    model = ref.convex_net.ConvexNextNet(n_hidden=32, in_features=2, n_hidden_layers=1)
    grid = T.get_positional_matrices(16, 16)[None]
    unaries = torch.from_numpy(disc_unaries(16, 16, 8, 7, 4.5))[None, None]
    rec = {"grid": grid.numpy(), "unaries": unaries.numpy()}
    rec.update(sd_np(model, "sd0."))
    opt = torch.optim.Adamax(model.parameters(), lr=1e-2, weight_decay=1e-5)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=5, factor=0.5)
    crit = ref.uwl.UnariesWeightedLoss(ref.se.SE("mean"))
    losses, lrs = [], []
    for step in range(60):
        opt.zero_grad()
        loss = crit(torch.sigmoid(model(grid)), unaries)
        loss.backward()
        opt.step()
        model.enforce_convexity()
        sched.step(loss)
        losses.append(loss.item())
        lrs.append(opt.param_groups[0]["lr"])
    rec.update(sd_np(model, "final."))
    rec["losses"] = np.asarray(losses, dtype=np.float32)
    rec["lrs"] = np.asarray(lrs, dtype=np.float64)
    np.savez_compressed(os.path.join(out, "adamax_plateau_h32.npz"), **rec)
    print("wrote adamax", losses[0], losses[-1], lrs[-1])


def gen_losses(ref, out):
    """a11/a12: UnariesWeightedLoss modes, AwesomeImageLoss."""
    seed_all(11)
    outp = torch.rand(2, 1, 12, 10)
    tgt = (torch.rand(2, 1, 12, 10) > 0.8).float() * 0.9 + 0.05  # soft unaries, fg (<0.5) majority
    rec = {"output": outp.numpy(), "target": tgt.numpy()}
    for mode in ["none", "equal", "ratio", "sssdms"]:
        for cname, crit in [("se", ref.se.SE("mean")), ("bce", torch.nn.BCELoss())]:
            kw = dict(mode=mode)
            if mode == "ratio":
                kw["ratio"] = 0.35
            l = ref.uwl.UnariesWeightedLoss(crit, **kw)
            rec[f"uwl.{cname}.{mode}"] = np.float32(l(outp, tgt).item())
    # AwesomeImageLoss (awesome_image_loss.py:34-53)
    out2 = torch.rand(2, 2, 12, 10)
    rec["output2"] = out2.numpy()
    tb = (tgt >= 0.5).float()
    rec["target_bin"] = tb.numpy()
    l = ref.ail.AwesomeImageLoss(alpha=0.7, beta=100., gamma=0.1, forward_kwargs_criterion=False)
    rec["ail.plain"] = np.float32(l(out2, tb).item())
    l.extra_penalty = True
    rec["ail.penalty"] = np.float32(l(out2, tb).item())
    np.savez_compressed(os.path.join(out, "losses.npz"), **rec)
    print("wrote losses")


def gen_miou(ref, out):
    """a17: MIOU(invert=True, average='binary') (miou.py:29-48)."""
    seed_all(13)
    m = ref.miou.MIOU(average="binary", invert=True)
    rec = {}
    cases = []
    for i in range(6):
        o = (torch.rand(9, 11) > 0.5).float()
        t = (torch.rand(9, 11) > (0.3 + 0.1 * i)).float()
        cases.append((o, t))
    cases.append((torch.zeros(5, 5), torch.ones(5, 5)))   # target has no fg after inversion -> 0
    cases.append((torch.ones(5, 5), torch.zeros(5, 5)))   # pred no fg, target all fg
    cases.append((torch.zeros(5, 5), torch.zeros(5, 5)))  # perfect, all fg
    for i, (o, t) in enumerate(cases):
        rec[f"o{i}"] = o.numpy()
        rec[f"t{i}"] = t.numpy()
        rec[f"iou{i}"] = np.float32(m(o, t).item())
    rec["n"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(out, "miou.npz"), **rec)
    print("wrote miou")


def gen_grid(ref, out):
    """a1: Transformator.get_positional_matrices (transformator.py:25-61)."""
    T = ref.transformator.Transformator
    rec = {}
    rec["g_7x5"] = T.get_positional_matrices(7, 5).numpy()
    rec["g_64x64"] = T.get_positional_matrices(64, 64).numpy()
    rec["g_256x256_row0"] = T.get_positional_matrices(256, 256).numpy()[0, 0]
    rec["g_6x4_t"] = T.get_positional_matrices(6, 4, t=3.0, t_max=15.0).numpy()
    np.savez_compressed(os.path.join(out, "grid.npz"), **rec)
    print("wrote grid")


def gen_fit_disc(ref, out):
    """End-to-end pin: 64x64 disc, ConvexNextNet(L=1), Adam lr 2e-3, 600 steps."""
    T = ref.transformator.Transformator
    seed_all(0)
    model = ref.convex_net.ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=1)
    grid = T.get_positional_matrices(64, 64)[None]
    unaries = torch.from_numpy(disc_unaries(64, 64, 32, 32, 15))[None, None]
    rec = {"unaries": unaries.numpy()}
    rec.update(sd_np(model, "sd0."))
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    crit = ref.uwl.UnariesWeightedLoss(ref.se.SE("mean"))
    losses = []
    for step in range(600):
        opt.zero_grad()
        outp = torch.sigmoid(model(grid))
        loss = crit(outp, unaries)
        loss.backward()
        opt.step()
        model.enforce_convexity()
        losses.append(loss.item())
    with torch.no_grad():
        logits = model(grid)
        outp = torch.sigmoid(logits)
    miou = ref.miou.MIOU(average="binary", invert=True)
    rec["losses"] = np.asarray(losses, dtype=np.float32)
    rec["final_logits"] = logits.numpy().astype(np.float32)
    rec["final_mask"] = (outp > 0.5).numpy()
    rec["final_miou"] = np.float32(miou((outp > 0.5).float(), (unaries > 0.5).float()).item())
    rec.update(sd_np(model, "final."))
    np.savez_compressed(os.path.join(out, "fit_disc64.npz"), **rec)
    print("wrote fit_disc64: loss", losses[0], "->", losses[-1], "miou", rec["final_miou"])


def gen_fit_blob256(ref, out, layers=1, name="fit_blob256_reference"):
    """BASELINE configs[1] end to end with the REAL reference classes: 256x256 convex blob (awesome_amd.dataset.convex_blob_unaries,
    seed 0 - pure numpy, no reference code), ConvexNextNet(h=130, L=1) seeded like bench.py rank 0, UnariesWeightedLoss(SE('mean')),
    torch.optim.Adam(lr 2e-3), enforce_convexity, 2000 full-batch steps (~3 min on 8 threads).  Only the final mask, the loss
    curve and the fg-mIoU are kept (8 KB).  `layers=2` writes the same fit for the two-hidden-layer net every flow prior uses
    (fit_blob256_l2_reference.npz, ~5 min)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from awesome_amd.dataset.synthetic import convex_blob_unaries   # numpy-only synthetic input (not product compute)
    T = ref.transformator.Transformator
    torch.manual_seed(0)
    model = ref.convex_net.ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=layers)
    grid = T.get_positional_matrices(256, 256)[None]
    unaries = convex_blob_unaries(256, 0)[None, None]
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    crit = ref.uwl.UnariesWeightedLoss(ref.se.SE("mean"))
    losses = []
    for step in range(2000):
        opt.zero_grad()
        loss = crit(torch.sigmoid(model(grid)), unaries)
        loss.backward()
        opt.step()
        model.enforce_convexity()
        losses.append(loss.item())
    with torch.no_grad():
        outp = torch.sigmoid(model(grid))
    miou = ref.miou.MIOU(average="binary", invert=True)
    rec = {"losses": np.asarray(losses, dtype=np.float32), "final_mask": (outp > 0.5).numpy().reshape(-1),
           "final_miou": np.float32(miou((outp > 0.5).float(), (unaries > 0.5).float()).item())}
    np.savez_compressed(os.path.join(out, name + ".npz"), **rec)
    print("wrote", name, ": loss", losses[0], "->", losses[-1], "miou", rec["final_miou"])


def gen_cdn_fit(ref, out):
    """The path-connected prior end to end with the reference's own modules: ConvexNextNet(130, L=2) behind
    NormalizingFlow1D(6 couplings, width 130, normal_block) behind nn.Linear(2, 2) - composed exactly as
    ConvexDiffeomorphismNet.forward does (convex_diffeomorphism_net.py:173-178; the class itself needs `toml` to import) - and
    trained like its pretrain loop (:405-430): Adam over get_weight_normalized_param_groups(5e-5), lr 3e-3, BCE on the sigmoid,
    enforce_convexity; 300 steps on a 48x48 two-disc shape."""
    T = ref.transformator.Transformator
    seed_all(4)
    convex_net = ref.convex_net.ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=2)
    diffeo_net = ref.diffeo.NormalizingFlow1D(num_coupling=6, width=130, in_features=2, backbone="normal_block")
    linear = torch.nn.Linear(2, 2)
    linear.weight.data.normal_(0.0, 1 / np.sqrt(2))
    linear.bias.data.fill_(0)
    net = torch.nn.ModuleDict(dict(convex_net=convex_net, diffeo_net=diffeo_net, linear=linear))
    S = 48
    yy, xx = np.mgrid[0:S, 0:S]
    mask = (((yy - 16) ** 2 + (xx - 14) ** 2) < 60) | (((yy - 32) ** 2 + (xx - 32) ** 2) < 60) | \
           ((np.abs(yy - 16) < 3) & (xx >= 14) & (xx <= 32)) | ((np.abs(xx - 32) < 3) & (yy >= 16) & (yy <= 32))
    unaries = torch.from_numpy(1.0 - mask.astype(np.float32))[None, None]
    grid = T.get_positional_matrices(S, S)[None]
    rows = grid.permute(0, 2, 3, 1).reshape(-1, 2)

    def forward():
        return convex_net(diffeo_net(linear(rows))).reshape(1, S, S, 1).permute(0, 3, 1, 2)

    rec = {"unaries": unaries.numpy()}
    rec.update(sd_np(net, "sd0."))
    opt = torch.optim.Adam(ref.autil.get_weight_normalized_param_groups(net, 5e-5, norm_suffix="weight_g"), lr=3e-3)
    crit = torch.nn.BCELoss()
    losses = []
    for step in range(300):
        opt.zero_grad()
        loss = crit(torch.sigmoid(forward()), unaries)
        loss.backward()
        opt.step()
        convex_net.enforce_convexity()
        losses.append(loss.item())
    with torch.no_grad():
        logits = forward()
    rec["losses"] = np.asarray(losses, dtype=np.float32)
    rec["final_logits"] = logits.numpy().astype(np.float32)
    rec.update(sd_np(net, "final."))
    np.savez_compressed(os.path.join(out, "cdn_fit48.npz"), **rec)
    print("wrote cdn_fit48: loss", losses[0], "->", losses[-1])


def gen_flow(ref, out):
    """a6-a9: WNLinear, NormalBlock, WNScale, NormalizingFlow1D forward/grad (diffeomorphism_net.py)."""
    seed_all(21)
    x = torch.rand(40, 2) * 2 - 0.5
    rec = {"x": x.numpy()}
    wn = ref.resnet_1d.WNLinear(3, 5)
    wn.reset_parameters("relu")
    x3 = torch.rand(17, 3)
    rec["wn.x"] = x3.numpy()
    rec.update(sd_np(wn, "wn.sd."))
    rec["wn.y"] = wn(x3).detach().numpy()
    nb = ref.diffeo.NormalBlock(in_channels=1, mid_channels=24, out_channels=1)
    nb.reset_parameters()
    x1 = torch.rand(17, 1) * 2 - 1
    rec["nb.x"] = x1.numpy()
    rec.update(sd_np(nb, "nb.sd."))
    rec["nb.y"] = nb(x1).detach().numpy()
    sc = ref.diffeo.WNScale(dim=1)
    rec.update(sd_np(sc, "sc.sd."))
    rec["sc.y"] = sc().detach().numpy()
    for tag, kw in [("nf6_w130", dict(num_coupling=6, width=130, backbone="normal_block")),
                    ("nf4_w16", dict(num_coupling=4, width=16, backbone="normal_block"))]:
        nf = ref.diffeo.NormalizingFlow1D(in_features=2, **kw)
        nf.reset_parameters()
        rec.update(sd_np(nf, f"{tag}.sd."))
        y = nf(x)
        rec[f"{tag}.y"] = y.detach().numpy()
        (y ** 2).mean().backward()
        rec.update(grads_np(nf, f"{tag}.grad."))
    np.savez_compressed(os.path.join(out, "flow.npz"), **rec)
    print("wrote flow")


def gen_pixel_losses(ref, out):
    """AwesomeLoss (pixel mode, awesome/measures/awesome_loss.py:45-65)."""
    rec = {}
    seed_all(11)
    out_px = torch.rand(3, 40, 2)
    tgt_px = (torch.rand(3, 30, 1) > 0.5).float()
    crit = ref.al.AwesomeLoss(alpha=0.6, scribble_percentage=0.75)
    rec["al.output"], rec["al.target"] = out_px.numpy(), tgt_px.numpy()
    rec["al.plain"] = crit(out_px, tgt_px).numpy()
    crit.extra_penalty = True
    rec["al.penalty"] = crit(out_px, tgt_px).numpy()
    # (FBMSJointLoss is not importable here: fbms_joint_loss.py -> tracker_loss -> ... -> package_tools needs `toml`)
    np.savez_compressed(os.path.join(out, "pixel_losses.npz"), **rec)
    print("wrote pixel_losses", sorted(rec))


def gen_fcnet(ref, out):
    """FCNet(in_type='xy') (awesome/model/fc_net.py:10-59), the "no prior" coordinate network: logits on a 16x16 grid, loss +
    gradients of SE(sigmoid(f), u), parameters after 10 Adam steps (lr 2e-3, no clamp)."""
    for width, depth in ((130, 1), (64, 2)):
        seed_all(21 + depth)
        net = ref.fc_net.FCNet(in_chn=2, out_chn=1, width=width, depth=depth, in_type="xy")
        rec = sd_np(net)
        ys, xs = torch.meshgrid(torch.linspace(0, 1, 16), torch.linspace(0, 1, 16), indexing="ij")
        rows = torch.stack((xs, ys), -1).reshape(-1, 2).float()
        un = torch.from_numpy(blob_unaries(16, 16, 3)).reshape(-1, 1)
        rec["rows"], rec["unaries"] = rows.numpy(), un.numpy()
        y = net(None, rows)
        rec["logits"] = y.detach().numpy()
        loss = ((un - torch.sigmoid(y)) ** 2).mean()
        loss.backward()
        rec["loss"] = loss.detach().numpy()
        rec.update(grads_np(net))
        opt = torch.optim.Adam(net.parameters(), lr=2e-3)
        losses = []
        for _ in range(10):
            opt.zero_grad()
            l = ((un - torch.sigmoid(net(None, rows))) ** 2).mean()
            l.backward()
            opt.step()
            losses.append(float(l.item()))
        rec["adam10.losses"] = np.asarray(losses, np.float32)
        rec.update(sd_np(net, prefix="adam10."))
        np.savez_compressed(os.path.join(out, f"fcnet_w{width}_d{depth}.npz"), **rec)
    print("wrote fcnet")


def gen_minmax(ref, out):
    """MinMax as NormNet uses it around the RealNVP flow of PathConnectedNet (awesome/transforms/min_max.py:22-61;
    net_factory.py:160-162: MinMax(dim=(0, 2, 3)) fitted on the normalized grid, new range [-1, 1])."""
    rec = {}
    for C, (H, W) in ((2, (9, 13)), (3, (7, 11))):
        seed_all(5 + C)
        fit_x = torch.rand(2, C, H, W) * torch.tensor([1.3, 0.7, 2.0][:C]).view(1, C, 1, 1) + torch.tensor([-0.2, 0.1, 0.5][:C]).view(1, C, 1, 1)
        mm = ref.min_max.MinMax(new_min=-1.0, new_max=1.0, dim=(0, 2, 3))
        mm.fit(fit_x)
        x = torch.randn(1, C, H, W)
        y = mm.transform(x)
        rec[f"c{C}.fit_x"], rec[f"c{C}.x"] = fit_x.numpy(), x.numpy()
        rec[f"c{C}.min"], rec[f"c{C}.max"] = mm.min.numpy(), mm.max.numpy()
        rec[f"c{C}.transform"], rec[f"c{C}.inverse"] = y.numpy(), mm.inverse_transform(x).numpy()
    np.savez_compressed(os.path.join(out, "minmax.npz"), **rec)
    print("wrote minmax")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
    ap.add_argument("--only", default=None, help="run a single generator, e.g. fit_blob256_l2")
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    torch.set_num_threads(4)
    ref = _import_reference()
    if args.only == "fit_blob256_l2":
        return gen_fit_blob256(ref, out, layers=2, name="fit_blob256_l2_reference")
    gen_grid(ref, out)
    gen_miou(ref, out)
    gen_losses(ref, out)
    gen_icnn(ref, out)
    gen_adamax(ref, out)
    gen_flow(ref, out)
    gen_fit_disc(ref, out)
    gen_fit_blob256(ref, out)
    gen_fit_blob256(ref, out, layers=2, name="fit_blob256_l2_reference")
    gen_cdn_fit(ref, out)
    gen_minmax(ref, out)
    gen_pixel_losses(ref, out)
    gen_fcnet(ref, out)
    with open(os.path.join(out, "PROVENANCE.txt"), "w") as f:
        f.write("Generated by tools/gen_golden.py from jp-schneider/awesome @ 2024_08_07 (reference classes imported on CPU),\n")
        f.write(f"torch {torch.__version__}, numpy {np.__version__}.\n")
        f.write("All files regenerate bit for bit except the long fp32 trajectories fit_blob256*_reference.npz and cdn_fit48.npz:\n"
                "the reference's multi-threaded CPU fit differs between runs from the 1-ulp level on (two runs of the 2000-step\n"
                "fit: 1 mask pixel, 1.7e-4 mIoU, 0.7 % of the final loss apart); the committed files are one such run.\n")


if __name__ == "__main__":
    main()
