#!/usr/bin/env python3
"""Golden fixtures at BASELINE scale, generated from the REAL reference classes (CPU, build container only).

    multi   BASELINE configs[2], the first `--seeds` images: for every seed s the reference's ConvexNextNet(h=130, L=1) seeded with
            torch.manual_seed(s), the 256x256 convex blob of seed s, UnariesWeightedLoss(SE('mean')) on the sigmoid, Adam(lr 2e-3),
            enforce_convexity, 2000 full-batch steps - the loop of tools/gen_golden.py gen_fit_blob256, which is image 0 of this set.
            Kept per seed: the final mask (bit-packed, 8 KB), its fg-mIoU against the unaries and the loss curve; and, because
            an end-of-fit snapshot of a spiking loss cannot discriminate, three SPIKE-ROBUST statistics of the
            last TAIL = 50 training forwards (outputs at the parameters in front of steps 1951..2000, the tensors the loop already
            computes): `tail_miou` (fg-mIoU of each), `tailmean_miou` / `tailmean_mask_bits` (mask of the MEAN probability over the
            tail) and `tailbest_miou` (the best of the tail - what an IoU gate polling the last K steps would accept).
            Run TWICE with different OpenMP thread counts (`--tag a --threads 3`, `--tag b --threads 2`): the reference's CPU fit is
            not reproducible across summation orders (tests/golden/PROVENANCE.txt), and the per-image |dIoU| between the two runs
            is the reference's own run-to-run floor that the GPU parity test's per-image bar is set against.
    cdn     one full-size fit of the path-connected prior class itself: ConvexDiffeomorphismNet(n_hidden=130, n_hidden_layers=2,
            nf_layers=6, nf_hidden=130, diffeo_args.backbone=normal_block) on the 256x256 blob of seed 0 with the hyper-parameters
            of config/path-connectedness/refit-unet-prior-only/*.yaml (pretrain_args: lr 1e-3, num_epochs 2000, criterion
            UnariesConversionLoss(SE('mean'))) through the inner loop of ConvexDiffeomorphismNet.pretrain
            (awesome/model/convex_diffeomorphism_net.py:377-430: Adam over get_weight_normalized_param_groups(5e-5),
            ReduceLROnPlateau(patience=200, factor=0.5) stepped on the loss, sigmoid, enforce_convexity).

Nothing in the product imports this file; the reference never travels - only the .npz files do.
Usage:  python tools/gen_golden_scale.py multi --tag a --threads 3 --seeds 16
        python tools/gen_golden_scale.py multi --tag b --threads 2 --seeds 16
        python tools/gen_golden_scale.py finalize
        python tools/gen_golden_scale.py cdn --threads 3
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))


TAIL = 50


def _blob(seed):
    from awesome_amd.dataset.synthetic import convex_blob_unaries   # numpy-only synthetic input (not product compute)
    return convex_blob_unaries(256, seed)[None, None]


def gen_multi(out, tag, threads, n_seeds, first=0):
    import gen_golden as G
    ref = G._import_reference()
    torch.set_num_threads(threads)
    T = ref.transformator.Transformator
    grid = T.get_positional_matrices(256, 256)[None]
    crit = ref.uwl.UnariesWeightedLoss(ref.se.SE("mean"))
    metric = ref.miou.MIOU(average="binary", invert=True)
    path = os.path.join(out, f"fits_blob256_multi_{tag}.npz")
    rec = dict(np.load(path)) if os.path.exists(path) else {}
    for s in range(first, first + n_seeds):
        if f"s{s}.final_miou" in rec:
            continue
        t0 = time.time()
        torch.manual_seed(s)
        model = ref.convex_net.ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=1)
        unaries = _blob(s)
        opt = torch.optim.Adam(model.parameters(), lr=2e-3)
        losses, tail, psum = [], [], torch.zeros(1, 1, 256, 256)
        gt = (unaries > 0.5).float()
        for step in range(2000):
            opt.zero_grad()
            prob = torch.sigmoid(model(grid))
            loss = crit(prob, unaries)
            if step >= 2000 - TAIL:
                with torch.no_grad():
                    psum += prob.detach()
                    tail.append(metric((prob.detach() > 0.5).float(), gt).item())
            loss.backward()
            opt.step()
            model.enforce_convexity()
            losses.append(loss.item())
        with torch.no_grad():
            outp = torch.sigmoid(model(grid))
        mask = (outp > 0.5).numpy().reshape(-1)
        rec[f"s{s}.final_mask_bits"] = np.packbits(mask)
        rec[f"s{s}.final_miou"] = np.float32(metric((outp > 0.5).float(), (unaries > 0.5).float()).item())
        rec[f"s{s}.losses"] = np.asarray(losses, dtype=np.float32)
        pm = psum / TAIL
        rec[f"s{s}.tail_miou"] = np.asarray(tail, dtype=np.float32)
        rec[f"s{s}.tailbest_miou"] = np.float32(max(tail))
        rec[f"s{s}.tailmean_miou"] = np.float32(metric((pm > 0.5).float(), gt).item())
        rec[f"s{s}.tailmean_mask_bits"] = np.packbits((pm > 0.5).numpy().reshape(-1))
        rec["tail"] = np.int32(TAIL)
        rec["threads"] = np.int32(threads)
        rec["torch_version"] = np.array(torch.__version__)
        np.savez_compressed(path + ".tmp.npz", **rec)
        os.replace(path + ".tmp.npz", path)
        print(f"[multi {tag}] seed {s}: miou {float(rec[f's{s}.final_miou']):.5f} tailmean {float(rec[f's{s}.tailmean_miou']):.5f} "
              f"tailbest {max(tail):.5f} loss {losses[-1]:.3e} ({time.time() - t0:.0f} s)", flush=True)


DIV_STEPS = (1, 3, 10, 30, 100, 300, 1000, 1999)


def finalize_multi(out):
    """After both runs: the step-wise divergence of the reference FROM ITSELF (hard part 1) - relative loss
    difference |loss_a - loss_b| / loss_a at DIV_STEPS, per image - and the run-to-run floors of every end-of-fit statistic, written
    into run a's file (`ab.*` keys).  This is what a per-image bar of the device fit can be set against."""
    pa, pb = (os.path.join(out, f"fits_blob256_multi_{t}.npz") for t in "ab")
    za, zb = dict(np.load(pa)), np.load(pb)
    seeds = sorted(int(k[1:].split(".")[0]) for k in za if k.endswith(".final_miou") and k in zb.files)
    div = np.array([[abs(float(za[f"s{s}.losses"][t]) - float(zb[f"s{s}.losses"][t])) / float(za[f"s{s}.losses"][t]) for t in DIV_STEPS]
                    for s in seeds], dtype=np.float32)
    za["ab.seeds"] = np.asarray(seeds, np.int32)
    za["ab.div_steps"] = np.asarray(DIV_STEPS, np.int32)
    za["ab.div_rel_loss"] = div                                          # [n_seeds, len(DIV_STEPS)]
    for stat in ("final_miou", "tailmean_miou", "tailbest_miou"):
        d = np.array([abs(float(za[f"s{s}.{stat}"]) - float(zb[f"s{s}.{stat}"])) for s in seeds], dtype=np.float32)
        za[f"ab.absdiff.{stat}"] = d
        print(f"[finalize] reference run a vs run b, {stat:14s}: max {d.max():.2e} median {np.median(d):.2e}  "
              f"mean a {np.mean([float(za[f's{s}.{stat}']) for s in seeds]):.5f} b {np.mean([float(zb[f's{s}.{stat}']) for s in seeds]):.5f}")
    print("[finalize] rel. loss divergence a vs b at steps", DIV_STEPS, ": median", np.median(div, 0).round(6).tolist(), "max", div.max(0).round(6).tolist())
    np.savez_compressed(pa + ".tmp.npz", **za)
    os.replace(pa + ".tmp.npz", pa)


def gen_cdn(out, threads, tag=""):
    import gen_golden as G
    import gen_golden_boundary as GB
    ref = GB._import_reference()
    import awesome.measures.unaries_conversion_loss as ucl
    import awesome.measures.miou as miou
    import awesome.util.torch as autil
    torch.set_num_threads(threads)
    G.seed_all(42)   # the configs' seed
    model = ref.cdn.ConvexDiffeomorphismNet(n_hidden=130, n_hidden_layers=2, nf_layers=6, nf_hidden=130,
                                            diffeo_args=dict(backbone="normal_block"))
    grid = GB.linspace_grid(256, 256)
    unaries = _blob(0)
    crit = ucl.UnariesConversionLoss(ref.se.SE("mean"))
    rec = {"sd0." + k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    groups = autil.get_weight_normalized_param_groups(model, 5e-5, norm_suffix="weight_g")
    opt = torch.optim.Adam(groups, lr=1e-3)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=200, factor=0.5)
    losses, lrs, tail, psum = [], [], [], torch.zeros(1, 1, 256, 256)
    metric = miou.MIOU(average="binary", invert=True)
    gt = (unaries > 0.5).float()
    model.train()
    t0 = time.time()
    out_t = None
    for step in range(2000):
        opt.zero_grad()
        out_t = torch.sigmoid(model(grid))           # WrapperModule.process_prior_output(use_sigmoid=True), wrapper_module.py:265-273
        if step >= 2000 - TAIL:                      # spike-robust statistics of the last TAIL training forwards (see `multi`)
            with torch.no_grad():
                psum += out_t.detach()
                tail.append(metric((out_t.detach() > 0.5).float(), gt).item())
        loss = crit(out_t, unaries)
        loss.backward()
        opt.step()
        sched.step(loss)
        model.enforce_convexity()
        losses.append(loss.item())
        lrs.append(opt.param_groups[0]["lr"])
        if step % 100 == 0:
            print(f"[cdn] step {step} loss {losses[-1]:.4e} lr {lrs[-1]:.2e} ({time.time() - t0:.0f} s)", flush=True)
    pm = psum / TAIL
    rec["tail"] = np.int32(TAIL)
    rec["tail_miou"] = np.asarray(tail, dtype=np.float32)
    rec["tailbest_miou"] = np.float32(max(tail))
    rec["tailmean_miou"] = np.float32(metric((pm > 0.5).float(), gt).item())
    rec["tailmean_mask_bits"] = np.packbits((pm > 0.5).numpy().reshape(-1))
    # the gate's metric is taken on the output of the LAST training forward (convex_diffeomorphism_net.py:432-434) ...
    rec["gate_miou"] = np.float32(metric((out_t.detach() > 0.5).float(), (unaries > 0.5).float()).item())
    rec["gate_mask_bits"] = np.packbits((out_t.detach() > 0.5).numpy().reshape(-1))
    with torch.no_grad():   # ... the stored state is the one after the last optimizer step
        fin = torch.sigmoid(model(grid))
    rec["final_miou"] = np.float32(metric((fin > 0.5).float(), (unaries > 0.5).float()).item())
    rec["final_mask_bits"] = np.packbits((fin > 0.5).numpy().reshape(-1))
    rec["losses"] = np.asarray(losses, dtype=np.float32)
    rec["lrs"] = np.asarray(lrs, dtype=np.float64)
    rec["threads"] = np.int32(threads)
    np.savez_compressed(os.path.join(out, f"cdn_fit256_reference{tag}.npz"), **rec)
    print(f"[cdn] wrote: loss {losses[0]:.4e} -> {losses[-1]:.4e}, gate miou {float(rec['gate_miou']):.5f}, final {float(rec['final_miou']):.5f}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["multi", "cdn", "finalize"])
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    ap.add_argument("--tag", default="a")
    ap.add_argument("--threads", type=int, default=3)
    ap.add_argument("--seeds", type=int, default=16)
    ap.add_argument("--first", type=int, default=0)
    a = ap.parse_args()
    if a.what == "finalize":
        finalize_multi(a.out)
    elif a.what == "multi":
        gen_multi(a.out, a.tag, a.threads, a.seeds, a.first)
    else:
        gen_cdn(a.out, a.threads, "" if a.tag == "a" else "_" + a.tag)
