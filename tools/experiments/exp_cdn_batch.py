#!/usr/bin/env python3
"""ConvexDiffeomorphismNet, a batch of images per launch: us per optimizer step per launch shape of the point kernels (INR_FLOW_SHAPE)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import awesome_amd as A
from awesome_amd import flow as FL
from awesome_amd.dataset import convex_blob_unaries
from awesome_amd.model import ConvexDiffeomorphismNet
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = ConvexDiffeomorphismNet(n_hidden=130, n_hidden_layers=2, nf_layers=6, nf_hidden=130, diffeo_args=dict(backbone="normal_block"))
ispec, fspec = A.IcnnSpec(130, 2, 2), FL.FlowSpec(130, 6)
ip, fp = FL.split_cdn_state_dict(ispec, fspec, m.state_dict(), dev)
ip, fp = ip[None].repeat(n, 1).contiguous(), fp[None].repeat(n, 1).contiguous()
un = torch.stack([convex_blob_unaries(256, s).reshape(-1) for s in range(n)]).to(dev)
grid = A.Grid.linspace(256, 256, dev)
FL.cdn_fit(ispec, fspec, ip.clone(), fp.clone(), grid, un, 5, record_loss=False, want_logits=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
res = FL.cdn_fit(ispec, fspec, ip.clone(), fp.clone(), grid, un, 100, lr=3e-3, loss="bce")
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"INR_FLOW_SHAPE={os.environ.get('INR_FLOW_SHAPE')}: {n} images: {dt / 100 * 1e6:.1f} us per step ({dt / 100 / n * 1e6:.1f} per image); loss {float(res.loss_hist[0, -1]):.5f}")
