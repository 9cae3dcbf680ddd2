#!/usr/bin/env python3
"""Micro-bench of the layer-by-layer path (awesome_amd/csrc/wide.h): optimizer-step time of ICNN shapes without a fused kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import awesome_amd as A
from awesome_amd.dataset import convex_blob_unaries

dev = torch.device("cuda:0")
SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(256, 1), (350, 3), (130, 3), (512, 2)]
for h, L in SHAPES:
    spec = A.IcnnSpec(h, 2, L)
    torch.manual_seed(0)
    p = {k: (torch.rand(s) - 0.45) * (0.6 / h ** 0.5) for k, s in spec.keys_shapes()}
    flat = A.pack_state_dict(spec, p, dev)[None].contiguous()
    un = convex_blob_unaries(256, 0).reshape(1, -1).to(dev)
    g = A.Grid.linspace(256, 256, dev)
    A.fit(spec, flat.clone(), g, un, 5, lr=2e-3, record_loss=False, want_logits=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = A.fit(spec, flat.clone(), g, un, 100, lr=2e-3, record_loss=True, want_logits=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    flop = 3 * (2 * h * 2 + L * (2 * h * h + 2 * h * 2) + 2 * h + 4) * 65536
    print(f"wide h={h} L={L} 256x256: {dt * 1e6:.0f} us per optimizer step, {flop / dt / 1e12:.1f} TFLOP/s ({flop / dt / 1e12 / 157.3:.3f} of the fp32 "
          f"MFMA peak), loss {float(r.loss_hist[0, 0]):.4f} -> {float(r.loss_hist[0, -1]):.4f}", flush=True)
