#!/usr/bin/env python3
"""Micro-bench of the fused PathConnectedNet fit (RealNVP deformation + ICNN L=2): the spatial config (C=2, 12 flows,
256x256) and the (x,y,t) config of BASELINE configs[3] (C=3, 18 flows, 128x128x16)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import awesome_amd as A
from awesome_amd.dataset import convex_blob_unaries, SyntheticSequenceDataset
from awesome_amd.model import real_nvp_path_connected_net

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--case", type=str, default="both")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)


def run(name, m, grid, un, steps):
    m.fit_images(grid, un, num_epochs=10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = m.fit_images(grid, un, num_epochs=steps, lr=1e-3)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    h = res.loss_hist[0].cpu()
    iou = float(A.miou(torch.sigmoid(res.logits), un)[0])
    import hashlib
    ck = hashlib.sha1(b"".join(t.detach().cpu().numpy().tobytes() for t in (res.icnn_params, res.flow_params, res.flow_opt_state, res.loss_hist))).hexdigest()[:16]
    print(f"checksum {ck}")
    print(f"{name}: {dt / steps * 1e6:.1f} us per optimizer step; loss {float(h[0]):.4f} -> {float(h[-1]):.4f}; fg-IoU {iou:.4f}", flush=True)


if a.case in ("both", "xy"):
    m = real_nvp_path_connected_net(channels=2, hidden_units=32, flow_n_flows=12, flow_output_fn="tanh").to(dev)
    un = convex_blob_unaries(256, 0).reshape(1, -1).to(dev)
    run("PCN fit 256x256 C=2 F=12 hid=32 ICNN L=2", m, A.Grid.linspace(256, 256, dev), un, a.steps)
if a.case in ("both", "xyt"):
    ds = SyntheticSequenceDataset(1, 128, 16)
    m = real_nvp_path_connected_net(channels=3, hidden_units=32, flow_n_flows=18, flow_output_fn="tanh").to(dev)
    run("PCN fit 128x128x16 C=3 F=18 hid=32 ICNN L=2", m, A.Grid.explicit(ds.coords().to(dev)), ds.batch([0]).to(dev), a.steps)
