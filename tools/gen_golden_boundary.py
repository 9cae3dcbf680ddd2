#!/usr/bin/env python3
"""Golden vectors for the BOUNDARY classes of the hot path, generated from the REAL reference classes (CPU, build container
only; VERDICT r01 item 3, SURVEY.md §8c fixtures 3, 4, 8):

    ConvexDiffeomorphismNet (class forward / gradients / translate)   awesome/model/convex_diffeomorphism_net.py:41-188
    FBMSJointLoss (both clip branches)                                 awesome/measures/fbms_joint_loss.py:35-59
    WrapperModule.forward / split_model_output                         awesome/model/wrapper_module.py:157-319
    PriorCache.get_state() / PriorManager swap                         awesome/util/prior_cache.py:10-90, dataset/prior_dataset.py:70-110

These modules import, directly or through awesome/serialization, three third-party packages that are not installed here
(`toml`, `jsonpickle`, `simple_parsing`).  None of them takes part in the arithmetic being recorded: `toml` reads the package
name out of pyproject.toml, `jsonpickle` is a serialisation rule for unknown objects, `simple_parsing` extracts attribute
docstrings for argparse help texts.  They are replaced IN THIS PROCESS by the inert module objects below (SURVEY.md §8c's
recipe); every number written to the fixtures comes out of the reference's own code.  PROVENANCE.txt lists these fixtures
separately.  Nothing in the product imports this file; the reference never travels.

Usage:  python tools/gen_golden_boundary.py [--out tests/golden]
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import REF, seed_all, sd_np, grads_np, blob_unaries  # noqa: E402


def _install_inert_modules():
    toml = types.ModuleType("toml")
    toml.load = lambda *a, **k: {"tool": {"poetry": {"name": "awesome", "version": "0.0.0"}}}
    sys.modules["toml"] = toml
    jp = types.ModuleType("jsonpickle")
    jp.encode = lambda obj, *a, **k: json.dumps(str(obj))
    jp.decode = lambda s, *a, **k: s
    sys.modules["jsonpickle"] = jp
    sp = types.ModuleType("simple_parsing")
    spd = types.ModuleType("simple_parsing.docstring")
    spd.get_attribute_docstring = lambda *a, **k: types.SimpleNamespace(docstring_below="", comment_above="", comment_inline="")
    sp.docstring = spd
    sys.modules["simple_parsing"] = sp
    sys.modules["simple_parsing.docstring"] = spd


def _import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference checkout not present; fixtures can only be generated in the build container")
    _install_inert_modules()
    sys.path.insert(0, REF)
    for sub in ("model", "dataset"):  # skip the eager package __init__ files (they pull cv2 / torchvision)
        pkg = types.ModuleType(f"awesome.{sub}")
        pkg.__path__ = [os.path.join(REF, "awesome", sub)]
        sys.modules[f"awesome.{sub}"] = pkg
    import awesome.model.convex_net as convex_net
    import awesome.model.convex_diffeomorphism_net as cdn
    import awesome.model.wrapper_module as wrapper_module
    import awesome.model.forward_module as forward_module
    import awesome.measures.fbms_joint_loss as fbms
    import awesome.measures.se as se
    import awesome.measures.unaries_weighted_loss as uwl
    import awesome.util.prior_cache as prior_cache
    import awesome.dataset.prior_dataset as prior_dataset
    return types.SimpleNamespace(convex_net=convex_net, cdn=cdn, wrapper_module=wrapper_module, forward_module=forward_module,
                                 fbms=fbms, se=se, uwl=uwl, prior_cache=prior_cache, prior_dataset=prior_dataset)


def linspace_grid(h, w):
    xs, ys = torch.linspace(0, 1, w), torch.linspace(0, 1, h)
    return torch.stack([xs[None, :].expand(h, w), ys[:, None].expand(h, w)], 0)[None].contiguous()   # (1, 2, H, W)


def gen_cdn_class(ref, out):
    """The class itself (fixture 3): forward on a (1,2,H,W) grid, BCE(sigmoid) loss gradients w.r.t. every parameter, and
    translate / translate_only_point (the centre-of-mass warm start, :43-128)."""
    # the configs' form (diffeo_args: backbone normal_block, 6 couplings, width 130: config/path-connectedness/refit-unet-prior-
    # only/*.yaml:144-148) and the class's own default (NormalizingFlow1D's 'default' backbone = SimpleBackbone)
    for tag, kw, hw in (("l2_w130_k6", dict(n_hidden=130, n_hidden_layers=2, nf_layers=6, nf_hidden=130,
                                            diffeo_args=dict(backbone="normal_block")), (24, 20)),
                        ("l1_w24_k4", dict(n_hidden=64, n_hidden_layers=1, nf_layers=4, nf_hidden=24), (16, 16))):
        seed_all(21)
        import copy
        m = ref.cdn.ConvexDiffeomorphismNet(**copy.deepcopy(kw))
        grid = linspace_grid(*hw)
        un = torch.from_numpy(blob_unaries(hw[0], hw[1], 3))[None, None]
        logits = m(grid)
        loss = torch.nn.BCELoss()(torch.sigmoid(logits), un)
        loss.backward()
        rec = dict(grid=grid.numpy(), unaries=un.numpy(), logits=logits.detach().numpy(), loss=np.float32(loss.item()),
                   deformation=m.get_deformation(grid).detach().numpy(), kwargs=json.dumps(kw))
        rec.update(sd_np(m))
        rec.update(grads_np(m))
        # the centre-of-mass warm start between frames (:337-348): translate_only_point(prev_com (x, y), com (x, y), grid)
        src = torch.tensor([7, 9], dtype=torch.long)     # pixel (x, y) of the previous frame's centre of mass
        dst = torch.tensor([10, 6], dtype=torch.long)    # ... of the current frame
        m.zero_grad()
        with torch.no_grad():
            before = m(grid).numpy()
        m.translate_only_point(src, dst, grid=grid.squeeze())
        with torch.no_grad():
            after = m(grid).numpy()
        rec.update(tr_src=src.numpy(), tr_dst=dst.numpy(), tr_logits_before=before, tr_logits_after=after)
        rec.update(sd_np(m, "sd_tr."))
        np.savez_compressed(os.path.join(out, f"cdn_class_{tag}.npz"), **rec)
        print("cdn_class", tag, float(loss))


def gen_cdn_class_resnet(ref, out):
    """ConvexDiffeomorphismNet with NormalizingFlow1D's 'resnet' backbone (SimpleResnet: diffeomorphism_net.py:107-166, 256-260; no
    config selects it): forward on a (1,2,H,W) grid, the deformation, BCE(sigmoid) loss gradients w.r.t. every parameter."""
    import copy
    kw = dict(n_hidden=32, n_hidden_layers=1, nf_layers=4, nf_hidden=16, diffeo_args=dict(backbone="resnet", num_blocks=2))
    seed_all(23)
    m = ref.cdn.ConvexDiffeomorphismNet(**copy.deepcopy(kw))
    hw = (12, 10)
    grid = linspace_grid(*hw)
    un = torch.from_numpy(blob_unaries(hw[0], hw[1], 3))[None, None]
    logits = m(grid)
    loss = torch.nn.BCELoss()(torch.sigmoid(logits), un)
    loss.backward()
    rec = dict(grid=grid.numpy(), unaries=un.numpy(), logits=logits.detach().numpy(), loss=np.float32(loss.item()),
               deformation=m.get_deformation(grid).detach().numpy(), kwargs=json.dumps(kw))
    rec.update(sd_np(m))
    rec.update(grads_np(m))
    np.savez_compressed(os.path.join(out, "cdn_class_resnet.npz"), **rec)
    print("cdn_class_resnet", float(loss), len([k for k in rec if k.startswith("sd.")]))


def gen_fbms_joint_loss(ref, out):
    """FBMSJointLoss (fixture 4): output (B, 2, H, W) = [seg, prior]; both clip branches, loss value and gradient w.r.t. the
    output."""
    rec = {}
    rng = np.random.RandomState(5)
    for case, (beta, scale) in enumerate(((0.5, 1.0), (50.0, 1.0), (2.0, 0.2))):
        out_t = torch.from_numpy(rng.uniform(0.02, 0.98, (2, 2, 12, 10)).astype(np.float32)).requires_grad_(True)
        tgt = torch.from_numpy((rng.uniform(size=(2, 1, 12, 10)) > 0.5).astype(np.float32))
        if scale != 1.0:   # make prior ~ seg so that the penalty is tiny
            with torch.no_grad():
                out_t[:, 1] = out_t[:, 0] + (out_t[:, 1] - out_t[:, 0]) * scale
        crit = ref.fbms.FBMSJointLoss(criterion=torch.nn.BCELoss(), alpha=1.0, beta=beta)
        loss = crit(out_t, tgt)
        loss.backward()
        rec[f"c{case}.output"] = out_t.detach().numpy()
        rec[f"c{case}.target"] = tgt.numpy()
        rec[f"c{case}.beta"] = np.float32(beta)
        rec[f"c{case}.loss"] = np.float32(loss.item())
        rec[f"c{case}.grad"] = out_t.grad.numpy().copy()
        print("fbms case", case, float(loss))
    import inspect
    rec["signature"] = str(inspect.signature(ref.fbms.FBMSJointLoss.__init__))
    np.savez_compressed(os.path.join(out, "fbms_joint_loss.npz"), **rec)


def gen_weighted_loss_noneclass(ref, out):
    """WeightedLoss on CLASS labels with a `noneclass` (awesome/measures/weighted_loss.py:11-92) - the criterion of 153 of the
    reference's FBMSJointLoss configs: WeightedLoss(BCELoss, mode sssdms, noneclass 2) - alone (three modes x two criteria) and inside
    FBMSJointLoss (value + gradient w.r.t. both output channels, both clip branches)."""
    import awesome.measures.weighted_loss as wl
    rng = np.random.RandomState(17)
    B, H, W = 2, 12, 15
    out_t = torch.from_numpy(rng.uniform(0.03, 0.97, size=(B, 2, H, W)).astype(np.float32))
    lab = rng.uniform(size=(B, 1, H, W))
    tgt = torch.from_numpy(np.where(lab < 0.07, 0.0, np.where(lab < 0.55, 1.0, 2.0)).astype(np.float32))   # few fg, ~45 % unlabeled
    rec = dict(output=out_t.numpy(), target=tgt.numpy())
    for kind in ("bce", "se"):
        for mode in ("none", "sssdms", "equal"):
            inner = torch.nn.BCELoss() if kind == "bce" else ref.se.SE("mean")
            for nc in (None, 2.0):
                crit = wl.WeightedLoss(inner, mode=mode, noneclass=nc)
                o = out_t[:, :1].clone().requires_grad_(True)
                t = tgt if nc is not None else torch.where(tgt == 2.0, torch.ones_like(tgt), tgt)
                loss = crit(o, t)
                loss.backward()
                tag = f"{kind}.{mode}.{'nc2' if nc is not None else 'all'}"
                rec[tag + ".loss"] = np.float32(loss.item())
                rec[tag + ".grad"] = o.grad.numpy().copy()
    for case, beta in enumerate((1.0, 300.0)):
        crit = ref.fbms.FBMSJointLoss(criterion=wl.WeightedLoss(torch.nn.BCELoss(), mode="sssdms", noneclass=2), alpha=1.0, beta=beta)
        o = out_t.clone().requires_grad_(True)
        loss = crit(o, tgt)
        loss.backward()
        rec[f"fbms{case}.beta"] = np.float32(beta)
        rec[f"fbms{case}.loss"] = np.float32(loss.item())
        rec[f"fbms{case}.grad"] = o.grad.numpy().copy()
        print("fbms + WeightedLoss(noneclass 2) case", case, float(loss))
    np.savez_compressed(os.path.join(out, "weighted_loss_noneclass.npz"), **rec)


def gen_wrapper(ref, out):
    """WrapperModule(ForwardModule, ConvexNextNet) forward (fixture 4b): image mode, param_clean_grid; with and without the
    segmentation inversion; evaluate_prior off (what the pretrain loop reads the unaries from)."""
    seed_all(31)
    prior = ref.convex_net.ConvexNextNet(n_hidden=32, in_features=2, n_hidden_layers=1)
    seg = ref.forward_module.ForwardModule()
    H, W = 10, 12
    grid = linspace_grid(H, W)[0]
    rng = np.random.RandomState(7)
    img = torch.from_numpy(rng.normal(size=(2, 1, H, W)).astype(np.float32))     # "image" = the logits the ForwardModule hands on
    feat = torch.zeros(2, 1, H, W)
    xy = grid[None].repeat(2, 1, 1, 1)
    rec = dict(img=img.numpy(), xy=xy.numpy())
    rec.update(sd_np(prior, "prior."))
    for inv in (False, True):
        wm = ref.wrapper_module.WrapperModule(segmentation_module=seg, prior_module=prior, prior_arg_mode="param_clean_grid",
                                              input_mode="image", use_segmentation_sigmoid=True, use_prior_sigmoid=True,
                                              use_segmentation_output_inversion=inv)
        with torch.no_grad():
            o = wm(img, feat, xy)
            rec[f"out_inv{int(inv)}"] = o.numpy()
            wm.evaluate_prior = False
            rec[f"seg_only_inv{int(inv)}"] = wm(img, feat, xy).numpy()
            wm.evaluate_prior = True
            parts = wm.split_model_output(o)
            rec[f"split0_seg_inv{int(inv)}"] = parts[0][0].numpy()
            rec[f"split0_prior_inv{int(inv)}"] = parts[0][1].numpy()
            pa, pk = wm.get_prior_args(img[0], feat[0], xy[0])
            rec[f"prior_arg_inv{int(inv)}"] = pa[0].numpy()
    np.savez_compressed(os.path.join(out, "wrapper_module.npz"), **rec)
    print("wrapper", rec["out_inv0"].shape)


def gen_wrapper_pixel(ref, out):
    """WrapperModule in PIXEL mode (the scribble-trained convexity configs, segmentation_training_mode 'single'): input (img, n_pixels,
    5) = (x, y, r, g, b) per pixel, prior_arg_mode 'xy_c_preattached', outputs concatenated per pixel; AwesomeLoss on it with 75 %
    scribble pixels, with and without the extra penalty: output, split, loss values and the gradient of every parameter."""
    import awesome.measures.awesome_loss as al
    seed_all(33)
    prior = ref.convex_net.ConvexNextNet(n_hidden=32, in_features=2, n_hidden_layers=1)
    seg = torch.nn.Sequential(torch.nn.Linear(5, 8), torch.nn.ReLU(), torch.nn.Linear(8, 1))   # stand-in per-pixel classifier
    rng = np.random.RandomState(9)
    x = torch.from_numpy(rng.uniform(0, 1, size=(2, 40, 5)).astype(np.float32))
    tgt = torch.from_numpy((rng.uniform(size=(2, 30, 1)) > 0.5).astype(np.float32))
    wm = ref.wrapper_module.WrapperModule(segmentation_module=seg, prior_module=prior, prior_arg_mode="xy_c_preattached",
                                          input_mode="pixel", use_segmentation_sigmoid=True, use_prior_sigmoid=True)
    rec = dict(x=x.numpy(), target=tgt.numpy())
    rec.update(sd_np(prior, "prior."))
    rec.update(sd_np(seg, "seg."))
    # At this commit `combine_outputs` compares its InputMode ENUM with the string 'pixel' (wrapper_module.py:236), so even in pixel mode
    # the two outputs are concatenated along dim 0: the class returns (img, 2 n_pixels, 1) = [seg pixels; prior pixels], not the
    # (img, n_pixels, 2) its docstring and AwesomeLoss (awesome_loss.py:49-50) expect.  The fixture keeps the class's raw output AND the
    # documented layout built from it (the same numbers, re-arranged); the losses are the reference's AwesomeLoss on the latter.
    n = x.shape[1]

    def documented(o):
        return o if o.shape[-1] == 2 else torch.cat([o[:, :n], o[:, n:]], dim=-1)

    o = wm(x)
    rec["out_raw"] = o.detach().numpy()
    rec["out"] = documented(o).detach().numpy()
    pa, _ = wm.get_prior_args(x[0])
    rec["prior_arg"] = pa[0].numpy()
    for penalty in (False, True):
        crit = al.AwesomeLoss(alpha=0.6, scribble_percentage=0.75)
        crit.extra_penalty = penalty
        wm.zero_grad()
        loss = crit(documented(wm(x)), tgt)
        loss.backward()
        tag = "pen" if penalty else "plain"
        rec[f"loss_{tag}"] = np.float32(loss.item())
        rec.update({f"grad_{tag}.prior.{k}": p.grad.detach().numpy().copy() for k, p in prior.named_parameters()})
        rec.update({f"grad_{tag}.seg.{k}": p.grad.detach().numpy().copy() for k, p in seg.named_parameters()})
    np.savez_compressed(os.path.join(out, "wrapper_module_pixel.npz"), **rec)
    print("wrapper pixel", rec["out"].shape, float(rec["loss_plain"]), float(rec["loss_pen"]))


def gen_prior_cache(ref, out):
    """PriorCache.get_state() (fixture 8): the layout a fitted cache has on disk - key names, dtypes, the JSON of model_args -
    and the PriorManager swap (enter loads the state of a key, exit stores the model's state back)."""
    PriorCache, PriorManager = ref.prior_cache.PriorCache, ref.prior_dataset.PriorManager
    model_args = dict(n_hidden=16, in_features=2, n_hidden_layers=1)
    seed_all(41)
    cache = PriorCache(ref.convex_net.ConvexNextNet, model_args)
    model = ref.convex_net.ConvexNextNet(**model_args)
    s3 = cache[3]            # generated on first access
    s7 = cache[7]
    with PriorManager(model, prior_state=(3, s3), prior_cache=cache):
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.25)
    state = cache.get_state()
    rec = {"model_type": state["model_type"], "model_args": state["model_args"], "store_device": state["store_device"],
           "cache_keys": json.dumps(sorted(state["cache"].keys())),
           "state_keys": json.dumps(list(state["cache"]["3"].keys()))}
    for k, sd in state["cache"].items():
        for n, v in sd.items():
            rec[f"cache.{k}.{n}"] = v.numpy()
    for n, v in s7.items():
        rec[f"generated7.{n}"] = v.numpy()
    # round trip: set_state on a fresh cache gives the same entries
    c2 = PriorCache(None, None)
    c2.set_state(state)
    assert sorted(c2.get_state()["cache"].keys()) == sorted(state["cache"].keys())
    np.savez_compressed(os.path.join(out, "prior_cache_state.npz"), **rec)
    print("prior_cache", rec["model_type"], rec["model_args"], rec["cache_keys"])


def _notebook_class(path, cell, name):
    """Execute ONE code cell of a reference notebook in an empty namespace and return the class it defines.  The cell text is read
    from the read-only reference checkout at generation time; it is not copied into this repository."""
    with open(os.path.join(REF, path)) as f:
        src = "".join(json.load(f)["cells"][cell]["source"])
    import torch.nn as nn
    import torch.nn.functional as F
    ns = {"np": np, "torch": torch, "nn": nn, "F": F}   # the names the notebook's import cell provides
    exec(compile(src, f"{path}#cell{cell}", "exec"), ns)   # noqa: S102 - the reference's own class definition
    return ns[name]


def gen_encode_notebooks(out):
    """The two encode networks of the notebooks, run as written: forward outputs and input/parameter gradients
    on a small grid of coordinates."""
    import warnings
    rec = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Net = _notebook_class("notebooks/imageRepresentationTest.ipynb", 5, "ourSimpleNetwork")
        seed_all(51)
        net = Net(2, 20, 24, 1, 30)       # d_in, d_features, d_hidden, d_out, factor (the notebook: 2, 20, 350, 3, 30)
    ys, xs = torch.meshgrid(torch.arange(0, 12), torch.arange(0, 10), indexing="ij")
    x = (torch.stack([ys.reshape(-1), xs.reshape(-1)], 1).float() / 10.0).requires_grad_(True)   # cell 7's x_test convention
    y = net(x)
    (y ** 2).mean().backward()
    rec["fourier.x"], rec["fourier.y"], rec["fourier.dx"] = x.detach().numpy(), y.detach().numpy(), x.grad.numpy().copy()
    rec.update(sd_np(net, "fourier.sd."))
    rec.update(grads_np(net, "fourier.grad."))
    Sine = _notebook_class("notebooks/icml_teaser_code/repeating/repeating.ipynb", 3, "myNet")
    seed_all(52)
    sn = Sine(16)
    with torch.no_grad():
        sn.offset.copy_(torch.tensor([[0.05, -0.1]]))
    x2 = (torch.rand(90, 2) - 0.5).requires_grad_(True)                                         # cell 4: coordinates in [-0.5, 0.5)
    y2 = sn(x2)
    (torch.sigmoid(y2) ** 2).mean().backward()
    rec["sine.x"], rec["sine.y"], rec["sine.dx"] = x2.detach().numpy(), y2.detach().numpy(), x2.grad.numpy().copy()
    rec.update(sd_np(sn, "sine.sd."))
    rec.update({"sine.grad." + k: p.grad.detach().numpy().copy() for k, p in sn.named_parameters() if p.grad is not None})
    np.savez_compressed(os.path.join(out, "encode_notebooks.npz"), **rec)
    print("encode notebooks", rec["fourier.y"].shape, rec["sine.y"].shape)


def gen_teaser_rotation_symmetric(out):
    """The rotational-symmetry teaser network, run as written in its notebook: outputs with and without the symmetry
    prior, gradients of every parameter (pose included) and of the input, and a short full-batch Adam trajectory of the notebook's
    loss (2 MSE(background) + MSE(foreground) on sigmoid outputs) on a mirror-symmetric shape."""
    Net = _notebook_class("notebooks/icml_teaser_code/rotation_symmetric/rotation_symmetric.ipynb", 2, "myNet")
    seed_all(53)
    net = Net(130)
    with torch.no_grad():
        net.offset.copy_(torch.tensor([[0.04, -0.06]]))
        net.orientation.fill_(0.3)
    rec = {}
    rec.update(sd_np(net, "sd."))
    ii, jj = torch.meshgrid(torch.arange(0, 20), torch.arange(0, 18), indexing="ij")
    x = torch.stack([ii.reshape(-1) / 20 - 0.5, jj.reshape(-1) / 18 - 0.5], 1).float()            # cell 3: indices / n - 0.5
    rec["x"] = x.numpy()
    for sp in (False, True):
        xr = x.clone().requires_grad_(True)
        net.zero_grad()
        y = net(xr, sp)
        (torch.sigmoid(y) ** 2).mean().backward()
        tag = "sym" if sp else "free"
        rec[f"{tag}.y"], rec[f"{tag}.dx"] = y.detach().numpy(), xr.grad.numpy().copy()
        rec.update({f"{tag}.grad.{k}": p.grad.detach().numpy().copy() for k, p in net.named_parameters()})
    # a heart-like shape, mirror-symmetric about an axis that is neither of the image axes and off-centre
    ang, cx, cy = 0.6, 0.05, -0.03
    u = (x[:, 0] - cx) * np.cos(ang) + (x[:, 1] - cy) * np.sin(ang)
    v = -(x[:, 0] - cx) * np.sin(ang) + (x[:, 1] - cy) * np.cos(ang)
    labels = (((u / 0.3) ** 2 + ((v.abs() - 0.08) / 0.16) ** 2) < 1.0).float()
    rec["labels"] = labels.numpy()
    back, fore = labels < 0.5, labels > 0.5
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    crit = torch.nn.MSELoss()
    losses = []
    for _ in range(8):
        ob, of = torch.sigmoid(net(x[back], True)).squeeze(), torch.sigmoid(net(x[fore], True)).squeeze()
        loss = 2 * crit(ob, labels[back]) + 1 * crit(of, labels[fore])
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    rec["adam8.loss"] = np.asarray(losses, np.float32)
    rec.update(sd_np(net, "adam8.sd."))
    np.savez_compressed(os.path.join(out, "teaser_rotation_symmetric.npz"), **rec)
    print("teaser rotation symmetric", rec["sym.y"].shape, losses[0], losses[-1], net.orientation.item(), net.offset.tolist())


def gen_teaser_star_shaped(out):
    """The star-shape teaser network, run as written in its notebook: outputs, gradients of every parameter (the
    centre included) and of the input, and a short full-batch Adam trajectory of the notebook's loop (MSE on sigmoid outputs, then the
    projection W2_r.weight <- relu(W2_r.weight)) on a star-shaped, non-convex target."""
    Net = _notebook_class("notebooks/icml_teaser_code/star_shaped/star.ipynb", 2, "myNet")
    seed_all(54)
    net = Net(130)
    with torch.no_grad():
        net.offset.copy_(torch.tensor([[-0.03, 0.05]]))
    net.offset.requires_grad = True                                                               # cell 3 frees it at epoch 1000
    rec = {}
    rec.update(sd_np(net, "sd."))
    ii, jj = torch.meshgrid(torch.arange(0, 20), torch.arange(0, 18), indexing="ij")
    x = torch.stack([ii.reshape(-1) / 19 - 0.5, jj.reshape(-1) / 17 - 0.5], 1).float()            # cell 3: indices / (n - 1) - 0.5
    rec["x"] = x.numpy()
    xr = x.clone().requires_grad_(True)
    y = net(xr)
    (torch.sigmoid(y) ** 2).mean().backward()
    rec["y"], rec["dx"] = y.detach().numpy(), xr.grad.numpy().copy()
    rec.update({f"grad.{k}": p.grad.detach().numpy().copy() for k, p in net.named_parameters()})
    # a five-armed star about (0.03, -0.05): inside <=> radius below a bound that depends on the direction
    dx, dy = x[:, 0] - 0.03, x[:, 1] + 0.05
    rad, phi = (dx ** 2 + dy ** 2).sqrt(), torch.atan2(dy, dx)
    inside = (rad < 0.22 + 0.12 * torch.cos(5 * phi)).float()
    labels = 1 - inside                                                                            # cell 3: labels = 1 - likelihood
    rec["labels"] = labels.numpy()
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    crit = torch.nn.MSELoss()
    losses = []
    for _ in range(8):
        loss = crit(torch.sigmoid(net(x)).squeeze(), labels)
        opt.zero_grad()
        loss.backward()
        opt.step()
        with torch.no_grad():
            net.W2_r.weight.data = torch.nn.functional.relu(net.W2_r.weight.data)
        losses.append(float(loss.detach()))
    rec["adam8.loss"] = np.asarray(losses, np.float32)
    rec.update(sd_np(net, "adam8.sd."))
    np.savez_compressed(os.path.join(out, "teaser_star_shaped.npz"), **rec)
    print("teaser star shaped", rec["y"].shape, losses[0], losses[-1], net.offset.tolist(), float((net.W2_r.weight == 0).float().mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    ap.add_argument("--only", default=None, help="run a single generator, e.g. wrapper_pixel")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.set_num_threads(4)
    ref = _import_reference()
    if args.only == "wrapper_pixel":
        return gen_wrapper_pixel(ref, args.out)
    if args.only == "weighted_loss_noneclass":
        return gen_weighted_loss_noneclass(ref, args.out)
    if args.only == "cdn_resnet":
        return gen_cdn_class_resnet(ref, args.out)
    gen_weighted_loss_noneclass(ref, args.out)
    gen_wrapper_pixel(ref, args.out)
    gen_encode_notebooks(args.out)
    gen_teaser_rotation_symmetric(args.out)
    gen_teaser_star_shaped(args.out)
    gen_fbms_joint_loss(ref, args.out)
    gen_wrapper(ref, args.out)
    gen_prior_cache(ref, args.out)
    gen_cdn_class(ref, args.out)
    gen_cdn_class_resnet(ref, args.out)


if __name__ == "__main__":
    main()
