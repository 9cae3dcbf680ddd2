#!/usr/bin/env python3
"""Micro-bench of the fused ICNN(flow(Ax+b)) fit (ConvexDiffeomorphismNet: 6 couplings, width 130, ICNN L layers)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import awesome_amd as A
from awesome_amd import flow as FL
from awesome_amd.dataset import convex_blob_unaries
from awesome_amd.model import ConvexDiffeomorphismNet

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--layers", type=int, default=2)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--images", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = ConvexDiffeomorphismNet(n_hidden=130, n_hidden_layers=a.layers, nf_layers=6, nf_hidden=130, diffeo_args=dict(backbone="normal_block"))
ispec, fspec = A.IcnnSpec(130, 2, a.layers), FL.FlowSpec(130, 6)
ip, fp = FL.split_cdn_state_dict(ispec, fspec, m.state_dict(), dev)
ip, fp = ip[None].repeat(a.images, 1).contiguous(), fp[None].repeat(a.images, 1).contiguous()
S = a.size
un = torch.stack([convex_blob_unaries(S, s).reshape(-1) for s in range(a.images)]).to(dev)
grid = A.Grid.linspace(S, S, dev)
FL.cdn_fit(ispec, fspec, ip.clone(), fp.clone(), grid, un, 10, record_loss=False, want_logits=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
res = FL.cdn_fit(ispec, fspec, ip.clone(), fp.clone(), grid, un, a.steps, lr=3e-3, loss="bce", plateau=dict(patience=200, factor=0.5))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
h = res.loss_hist[0].cpu()
import hashlib
ck = hashlib.sha1(b"".join(t.detach().cpu().numpy().tobytes() for t in (res.icnn_params, res.flow_params, res.flow_opt_state, res.loss_hist))).hexdigest()[:16]
print(f"checksum {ck}")
print(f"CDN fit {S}x{S} x{a.images} L={a.layers} K=6 W=130: {dt / a.steps * 1e6:.1f} us per optimizer step "
      f"({dt / a.steps / a.images * 1e6:.1f} per image); loss {float(h[0]):.4f} -> {float(h[-1]):.4f}")
