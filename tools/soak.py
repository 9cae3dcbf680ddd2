#!/usr/bin/env python3
"""Soak: repeat complete fits of every prior and check that each repeat reproduces the first bit for bit (no atomics, fixed-order
reductions) and never reports a non-finite status."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import awesome_amd as A
from awesome_amd.dataset import convex_blob_unaries
from awesome_amd.model import ConvexNextNet, ConvexDiffeomorphismNet, real_nvp_path_connected_net

dev = torch.device("cuda:0")
S, reps, steps = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 10, 300
un = convex_blob_unaries(S, 0).reshape(1, -1).to(dev)
grid = A.Grid.linspace(S, S, dev)


def check(name, run, n_steps=None):
    ref, t0 = None, time.time()
    for r in range(reps):
        out = run()
        torch.cuda.synchronize()
        if ref is None:
            ref = [o.clone() for o in out]
        else:
            assert all(torch.equal(a, b) for a, b in zip(ref, out)), f"{name}: repeat {r} differs"
    print(f"{name}: {reps} x {n_steps or steps} steps bitwise identical ({time.time() - t0:.1f} s)", flush=True)


for L in (1, 2):
    torch.manual_seed(0)
    m = ConvexNextNet(n_hidden=130, n_hidden_layers=L)
    p0 = m.flat_parameters()[None].to(dev)

    def run(p0=p0, spec=m.spec):
        res = A.fit(spec, p0.clone(), grid, un, steps, lr=2e-3)
        assert int(res.status.sum()) == 0
        return [res.params, res.loss_hist, res.logits]
    check(f"ConvexNextNet L={L}", run)

for h, L in ((256, 1), (350, 3), (131, 3)):   # the layer-by-layer path (csrc/wide.h + gemm.h: split contractions added in a fixed order)
    torch.manual_seed(0)
    m = ConvexNextNet(n_hidden=h, n_hidden_layers=L)
    p0 = m.flat_parameters()[None].to(dev)

    def run(p0=p0, spec=m.spec):
        res = A.fit(spec, p0.clone(), grid, un, 100, lr=2e-3)
        assert int(res.status.sum()) == 0
        return [res.params, res.loss_hist, res.logits]
    check(f"ConvexNextNet h={h} L={L} (layer by layer)", run, 100)

torch.manual_seed(0)
cdn = ConvexDiffeomorphismNet(n_hidden=130, n_hidden_layers=2, nf_layers=6, nf_hidden=130, diffeo_args=dict(backbone="normal_block")).to(dev)


def run_cdn():
    res = cdn.fit_images(grid, un, num_epochs=steps)
    assert int(res.status.sum()) == 0
    return [res.icnn_params, res.flow_params, res.loss_hist]


check("ConvexDiffeomorphismNet", run_cdn)
torch.manual_seed(0)
pcn = real_nvp_path_connected_net(channels=2, hidden_units=32, flow_n_flows=12, flow_output_fn="tanh").to(dev)
pcn.fit_images(grid, un, num_epochs=1)   # ActNorm init once


def run_pcn():
    res = pcn.fit_images(grid, un, num_epochs=steps)
    assert int(res.status.sum()) == 0
    return [res.icnn_params, res.flow_params, res.loss_hist]


check("PathConnectedNet (RealNVP)", run_pcn)
