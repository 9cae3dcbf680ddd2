"""Parity at BASELINE scale against the REFERENCE CLASSES' own fits, not against the HIP path itself.

Fixtures (tools/gen_golden_scale.py, build container, the reference's classes imported from /root/reference):
  * fits_blob256_multi_{a,b}.npz - configs[2]'s first 16 images (blob seed s, ConvexNextNet seeded with torch.manual_seed(s)), the
    2000-step fit of each with the reference classes, TWICE (3 and 2 OpenMP threads).  The reference's CPU fit is not reproducible
    across summation orders: the two runs disagree per image by up to several 1e-3 of fg-mIoU - that disagreement is the reference's
    own run-to-run floor, and the per-image bar of the HIP fit is that floor times a stated factor.
  * cdn_fit256_reference[_b].npz - one full-size fit of the path-connected prior class itself (ConvexDiffeomorphismNet K6 w130 L2,
    256x256, the hyper-parameters of config/path-connectedness/refit-unet-prior-only/*.yaml), again twice.
What north_star bounds by +-1e-3 is the DATASET mIoU: asserted against the reference's mean."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

S, E = 256, 2000
FLOOR_FACTOR = 2.0     # per-image bar = FLOOR_FACTOR x the reference's own largest run-to-run difference of the same statistic


@pytest.fixture(scope="module")
def amd():
    import awesome_amd
    awesome_amd._lib.load()
    assert torch.cuda.is_available()
    return awesome_amd


def _multi(golden_dir):
    za = np.load(os.path.join(golden_dir, "fits_blob256_multi_a.npz"))
    zb = np.load(os.path.join(golden_dir, "fits_blob256_multi_b.npz"))
    seeds = sorted(int(k[1:].split(".")[0]) for k in za.files if k.endswith(".final_miou") and k in zb.files)
    return za, zb, seeds


def _hist(d, edges=(0.0, 1e-4, 3e-4, 1e-3, 3e-3, 1e-2, 1.0)):
    return np.histogram(np.asarray(d), edges)[0].tolist()


def _tail_fit(amd, spec, init, grid, un, tail):
    """The 2000-step fit with the outputs of its last `tail` TRAINING forwards (the tensors the reference's loop holds at steps
    E - tail .. E - 1): one call for E - tail steps, then `tail` one-step calls that continue it (same optimizer state, step count) and
    return the logits of that step's forward.  -> (final result, per-step fg-mIoU [tail, n], mean probability [n, N], losses [n, E])."""
    res = amd.fit(spec, init.clone(), grid, un, E - tail, lr=2e-3, record_loss=True, want_logits=False)
    losses, psum, per_step = [res.loss_hist], torch.zeros_like(un), []
    gt = (un > 0.5).float()
    for k in range(tail):
        res = amd.fit(spec, res.params, grid, un, 1, lr=2e-3, opt_state=res.opt_state, step0=E - tail + k, record_loss=True,
                      want_logits=True, gate_logits=True)
        prob = torch.sigmoid(res.logits)
        psum += prob
        per_step.append(amd.miou((prob > 0.5).float(), gt, invert=True))
        losses.append(res.loss_hist)
    assert int(res.status.sum()) == 0
    return res, torch.stack(per_step), psum / tail, torch.cat(losses, 1)


def _final_iou(amd, spec, res, grid, un):
    """fg-mIoU of the mask at the FINAL parameters (the reference's `final_miou`: a forward after the last optimizer step)."""
    prob = torch.sigmoid(amd.forward(spec, res.params, grid))
    return amd.miou((prob > 0.5).float(), (un > 0.5).float(), invert=True)


def test_configs2_batched_fit_against_the_reference_classes(amd, golden_dir):
    """configs[2] against the REFERENCE CLASSES' own fits, per image, with a statistic that can fail.

    The end-of-fit snapshot cannot discriminate: full-batch Adam at lr 2e-3 spikes, and two runs of the reference itself (3 and 2 OpenMP
    threads) end up to 6.5e-3 of fg-mIoU apart on the same image.  What both reference runs DO agree on, to 2.9e-4 over the 16 images,
    are spike-robust statistics of the fit's last TAIL = 50 training forwards, recorded in the fixtures by tools/gen_golden_scale.py:
      * tailmean: fg-mIoU of the mask of the MEAN probability over the tail,
      * tailbest: the best per-step fg-mIoU of the tail (what an IoU gate polling the last steps would accept).
    Asserted, for the batched fit (16 reference-fitted images + 48 more = configs[2]'s per-GPU share of 64, 4 gradient slabs per image)
    AND for every image fitted alone (256 slabs): per image both statistics within FLOOR_FACTOR x the reference's own run-to-run
    difference of that statistic (~6e-4: below north_star's 1e-3), the tail-mean mask within a handful of pixels, the dataset means
    within 1e-3 (also for the end-of-fit snapshot, the number the reference reports), and the head of every loss curve.  Printed: the
    step-wise divergence of the device fit from reference run a next to the reference's own a-vs-b divergence (fixture keys `ab.*`)."""
    from tests.test_gpu_determinism import _problem
    za, zb, seeds = _multi(golden_dir)
    assert len(seeds) >= 16 and seeds[:16] == list(range(16)) and int(za["tail"]) == 50
    n_ref, TAIL = len(seeds), int(za["tail"])
    ref = {st: (np.array([float(za[f"s{s}.{st}"]) for s in seeds]), np.array([float(zb[f"s{s}.{st}"]) for s in seeds]))
           for st in ("final_miou", "tailmean_miou", "tailbest_miou")}
    floor = {st: float(np.abs(a - b).max()) for st, (a, b) in ref.items()}
    assert floor["tailmean_miou"] < 1e-3 and floor["tailbest_miou"] < 1e-3 < floor["final_miou"]     # why the snapshot is not the bar
    spec, init, grid, un = _problem(amd, seeds=tuple(range(64)))
    div_steps = [int(t) for t in za["ab.div_steps"]]

    def check(tag, per_step, pmean, final_iou, losses, un_k, idx):
        """idx: positions (in `seeds`) of the images in this launch"""
        gt = (un_k > 0.5).float()
        tm = amd.miou((pmean > 0.5).float(), gt, invert=True).cpu().numpy().astype(np.float64)
        tb = per_step.max(0).values.cpu().numpy().astype(np.float64)
        got = {"tailmean_miou": tm, "tailbest_miou": tb, "final_miou": final_iou.cpu().numpy().astype(np.float64)}
        out = {}
        for st in ("tailmean_miou", "tailbest_miou"):
            a, b = ref[st][0][idx], ref[st][1][idx]
            d = np.minimum(np.abs(got[st] - a), np.abs(got[st] - b))
            out[st] = d
            assert d.max() <= FLOOR_FACTOR * floor[st] + 1e-6, (tag, st, d.max(), floor[st], got[st], a, b)
        for k, i in enumerate(idx):   # the tail-mean mask against the nearer reference run's
            m = (pmean[k] > 0.5).cpu().numpy().reshape(-1)
            ma = np.unpackbits(za[f"s{seeds[i]}.tailmean_mask_bits"])[: S * S].astype(bool)
            mb = np.unpackbits(zb[f"s{seeds[i]}.tailmean_mask_bits"])[: S * S].astype(bool)
            assert min(int((m != ma).sum()), int((m != mb).sum())) <= max(2 * int((ma != mb).sum()), 12), (tag, seeds[i])
            np.testing.assert_allclose(losses[k, :100].cpu().numpy(), za[f"s{seeds[i]}.losses"][:100], rtol=1e-3, err_msg=f"{tag} seed {seeds[i]}")
        return got, out

    # ---- one launch of 64 images
    res, per_step, pmean, losses = _tail_fit(amd, spec, init, grid, un, TAIL)
    final_iou = _final_iou(amd, spec, res, grid, un)
    idx = np.arange(n_ref)
    got, dist = check("batched", per_step[:, :n_ref], pmean[:n_ref], final_iou[:n_ref], losses[:n_ref], un[:n_ref], idx)
    div = np.array([[abs(float(losses[k, t]) - float(za[f"s{seeds[k]}.losses"][t])) / float(za[f"s{seeds[k]}.losses"][t]) for t in div_steps]
                    for k in range(n_ref)])
    print(f"\nconfigs[2], {n_ref} images of one 64-image launch vs the reference classes' two runs (floors: "
          + ", ".join(f"{k} {v:.1e}" for k, v in floor.items()) + ")")
    for st in ("tailmean_miou", "tailbest_miou"):
        print(f"  {st:14s}: max |d| to the nearer reference run {dist[st].max():.2e}  hist {_hist(dist[st])}  mean hip {got[st].mean():.5f} "
              f"ref a {ref[st][0].mean():.5f} b {ref[st][1].mean():.5f}")
    print(f"  final snapshot: mean hip {got['final_miou'].mean():.5f} ref a {ref['final_miou'][0].mean():.5f} b {ref['final_miou'][1].mean():.5f}; "
          f"max |d| {np.minimum(np.abs(got['final_miou'] - ref['final_miou'][0]), np.abs(got['final_miou'] - ref['final_miou'][1])).max():.2e} "
          f"(reference a vs b: {floor['final_miou']:.2e})")
    print("  step-wise rel. loss divergence at steps", div_steps)
    print("    device vs reference a, median", np.median(div, 0).round(6).tolist(), "max", div.max(0).round(5).tolist())
    print("    reference a vs b,      median", np.median(za["ab.div_rel_loss"], 0).round(6).tolist(), "max", za["ab.div_rel_loss"].max(0).round(5).tolist())
    for st in ("final_miou", "tailmean_miou", "tailbest_miou"):
        for r in ref[st]:
            assert abs(got[st].mean() - r.mean()) <= 1e-3, (st, got[st].mean(), r.mean())
    # ---- every image alone (its own slab count): the same bars
    tm1, tb1, fin1 = [], [], []
    for k in range(n_ref):
        r1, ps1, pm1, l1 = _tail_fit(amd, spec, init[k:k + 1], grid, un[k:k + 1], TAIL)
        g1, _ = check(f"single {k}", ps1, pm1, _final_iou(amd, spec, r1, grid, un[k:k + 1]), l1, un[k:k + 1], np.array([k]))
        tm1.append(g1["tailmean_miou"][0]); tb1.append(g1["tailbest_miou"][0]); fin1.append(g1["final_miou"][0])
    print(f"  single-image fits: tailmean mean {np.mean(tm1):.5f} tailbest mean {np.mean(tb1):.5f} final mean {np.mean(fin1):.5f}")
    for r in ref["final_miou"]:
        assert abs(np.mean(fin1) - r.mean()) <= 1e-3


def test_convex_diffeomorphism_net_full_size_fit_against_the_reference_class(amd, golden_dir):
    """The pinned path-connected variant end to end at full size: the reference CLASS's own 2000-step fit (Adam over the weight-norm
    param groups, ReduceLROnPlateau, UnariesConversionLoss(SE), enforce_convexity; twice: 3 and 4 OpenMP threads) vs the fused
    inrfit_cdn_fit from the same state_dict.  Asserted: the loss curve's head, the final loss, and - per VERDICT r03 item 3 - the
    spike-robust statistics of the last 50 training forwards (mean-probability mask, best step) within 2 x the reference's own
    run-to-run difference of the same statistic (floored at 5e-4); the end-of-fit gate snapshot, which the reference's two runs
    reproduce to 5.6e-4 but two summation orders of the device fit only to ~3e-3 (0.96811 with round 3's first flow kernels, 0.96545 with
    its last ones), is printed and held to the 5e-3 it can support."""
    from awesome_amd import flow as FL
    from awesome_amd.dataset import convex_blob_unaries
    z = np.load(os.path.join(golden_dir, "cdn_fit256_reference.npz"))
    zb = np.load(os.path.join(golden_dir, "cdn_fit256_reference_b.npz"))
    TAIL = int(z["tail"])
    dev = torch.device("cuda:0")
    ispec, fspec = amd.IcnnSpec(130, 2, 2), FL.FlowSpec(130, 6)
    sd0 = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd0.")}
    ip, fp = FL.split_cdn_state_dict(ispec, fspec, sd0, dev)
    un = convex_blob_unaries(S, 0).reshape(1, -1).to(dev)
    un = (un >= 0.5).float()                                  # UnariesConversionLoss
    gt = (un > 0.5).float()
    grid = amd.Grid.linspace(S, S, dev)
    kw = dict(lr=1e-3, loss="se", weight_decay_on_weight_g=5e-5, plateau=dict(patience=200, factor=0.5))
    res = FL.cdn_fit(ispec, fspec, ip[None].contiguous(), fp[None].contiguous(), grid, un, E - TAIL, want_logits=False, **kw)
    hist, psum, per_step = [res.loss_hist], torch.zeros_like(un), []
    for k in range(TAIL):   # the last TAIL steps one by one, continuing the fit, with the output of each step's training forward
        res = FL.cdn_fit(ispec, fspec, res.icnn_params, res.flow_params, grid, un, 1, icnn_opt_state=res.icnn_opt_state,
                         flow_opt_state=res.flow_opt_state, step0=E - TAIL + k, gate_logits=True, **kw)
        prob = torch.sigmoid(res.logits)
        psum += prob
        per_step.append(float(amd.miou((prob > 0.5).float(), gt, invert=True)[0]))
        hist.append(res.loss_hist)
    assert int(res.status[0]) == 0
    h = torch.cat(hist, 1)[0].cpu().numpy()
    np.testing.assert_allclose(h[:60], z["losses"][:60], rtol=2e-3)
    assert h[-1] <= 1.5 * float(z["losses"][-1]) and float(z["losses"][-1]) <= 1.5 * h[-1]
    got = {"tailmean_miou": float(amd.miou(((psum / TAIL) > 0.5).float(), gt, invert=True)[0]), "tailbest_miou": max(per_step),
           "gate_miou": per_step[-1]}
    print(f"\nCDN 256x256 K6 w130 L2 (final loss hip {h[-1]:.3e} reference {float(z['losses'][-1]):.3e} / {float(zb['losses'][-1]):.3e}):")
    for st in ("tailmean_miou", "tailbest_miou", "gate_miou"):
        ra, rb = float(z[st]), float(zb[st])
        floor = abs(ra - rb)
        d = min(abs(got[st] - ra), abs(got[st] - rb))
        print(f"  {st:14s}: hip {got[st]:.5f} reference a {ra:.5f} b {rb:.5f} (floor {floor:.2e}) -> |d| {d:.2e}")
        bar = 5e-3 if st == "gate_miou" else FLOOR_FACTOR * max(floor, 5e-4)
        assert d <= bar, (st, got[st], ra, rb, bar)
    m_hip = ((psum / TAIL)[0] > 0.5).cpu().numpy()
    m_a = np.unpackbits(z["tailmean_mask_bits"])[: S * S].astype(bool)
    m_b = np.unpackbits(zb["tailmean_mask_bits"])[: S * S].astype(bool)
    assert min(int((m_hip != m_a).sum()), int((m_hip != m_b).sum())) <= max(3 * int((m_a != m_b).sum()), int(0.003 * S * S))
