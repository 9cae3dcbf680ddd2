#!/usr/bin/env python3
"""PathConnectedNet C = 2, a batch of images per launch: us per optimizer step per RealNVP launch shape (INR_RNVP_QF / _QB set by the caller)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import awesome_amd as A
from awesome_amd.dataset import convex_blob_unaries
from awesome_amd.model import real_nvp_path_connected_net
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = real_nvp_path_connected_net(channels=2, hidden_units=32, flow_n_flows=12, flow_output_fn="tanh").to(dev)
un = torch.stack([convex_blob_unaries(256, s).reshape(-1) for s in range(n)]).to(dev)
grid = A.Grid.linspace(256, 256, dev)
m.fit_images(grid, un, num_epochs=5)
torch.cuda.synchronize()
t0 = time.perf_counter()
res = m.fit_images(grid, un, num_epochs=100, lr=1e-3)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"QF={os.environ.get('INR_RNVP_QF')} QB={os.environ.get('INR_RNVP_QB')}: {n} images: {dt / 100 * 1e6:.1f} us per step ({dt / 100 / n * 1e6:.1f} per image); loss {float(res.loss_hist[0, -1]):.5f}")
