#!/usr/bin/env python3
"""Kernel micro-bench: average duration of the fused step kernel (and a whole optimizer step) for several grid sizes."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import awesome_amd as A
from awesome_amd.dataset import convex_blob_unaries
from awesome_amd.model import ConvexNextNet

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="+", default=[128, 256, 512])
    ap.add_argument("--images", type=int, nargs="+", default=[1])
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--fit-steps", type=int, default=0)
    ap.add_argument("--layers", type=int, default=1)
    ap.add_argument("--features", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    spec = A.IcnnSpec(130, a.features, a.layers)
    torch.manual_seed(0)
    p0 = ConvexNextNet(in_features=a.features, n_hidden_layers=a.layers).flat_parameters().to(dev)
    for S in a.sizes:
        for B in a.images:
            un = torch.stack([convex_blob_unaries(S, s).reshape(-1) for s in range(B)]).to(dev)
            params = p0[None].repeat(B, 1).contiguous()
            grid = A.Grid.linspace(S, S, dev, torch.full((B,), 0.3, device=dev) if a.features == 3 else None)
            ws = A.icnn.step_only(spec, params, grid, un, 10)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(a.rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); A.icnn.step_only(spec, params, grid, un, a.iters, workspace=ws); e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / a.iters * 1e3)
            h, c, L = 130, a.features, a.layers
            flop = 3 * (2 * h * c + 2 * h + L * (2 * h * h + 2 * h * c + 2 * h) + 2 * h + 2 * c + 1) * S * S * B   # ~3 x forward flops
            line = f"size {S:4d} images {B:3d}: step kernel {best:9.2f} us  {flop / best / 1e6:7.2f} TFLOP/s ({flop / best / 1e6 / 157.3:.3f} of peak)"
            if a.fit_steps:
                pr = params.clone()
                A.fit(spec, pr, grid, un, 20, record_loss=False, want_logits=False); torch.cuda.synchronize()
                t0 = time.perf_counter()
                A.fit(spec, pr, grid, un, a.fit_steps, record_loss=False, want_logits=False); torch.cuda.synchronize()
                line += f" | optimizer step {(time.perf_counter() - t0) / a.fit_steps * 1e6:8.2f} us"
            print(line, flush=True)

if __name__ == "__main__":
    main()
