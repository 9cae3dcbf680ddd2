#!/usr/bin/env python3
"""Premise check for pipelining the layer-by-layer path over two HIP streams: two independent fits of the same shape, back to back on one
stream vs concurrently on two (each with its own workspace): how much of a step's memory-bound kernels and GEMM epilogues hides behind
the other stream's GEMMs?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import awesome_amd as A
from awesome_amd.dataset import convex_blob_unaries
dev = torch.device("cuda:0")
S = int(os.environ.get("SIZE", 256))
un = convex_blob_unaries(256, 0).reshape(256, 256)[:S, :S].reshape(1, -1).contiguous().to(dev)
g = A.Grid.linspace(S, S, dev)
for h, L in [(256, 1), (350, 3)]:
    spec = A.IcnnSpec(h, 2, L)
    torch.manual_seed(0)
    p = {k: (torch.rand(s) - 0.45) * (0.6 / h ** 0.5) for k, s in spec.keys_shapes()}
    flat = A.pack_state_dict(spec, p, dev)[None].contiguous()
    steps = 60
    A.fit(spec, flat.clone(), g, un, 5, lr=2e-3, record_loss=False, want_logits=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        A.fit(spec, flat.clone(), g, un, steps, lr=2e-3, record_loss=False, want_logits=False)
    torch.cuda.synchronize()
    seq = time.perf_counter() - t0
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for st in (s1, s2):
        with torch.cuda.stream(st):
            A.fit(spec, flat.clone(), g, un, 2, lr=2e-3, record_loss=False, want_logits=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for st in (s1, s2):
        with torch.cuda.stream(st):
            A.fit(spec, flat.clone(), g, un, steps, lr=2e-3, record_loss=False, want_logits=False)
    torch.cuda.synchronize()
    par = time.perf_counter() - t0
    print(f"h={h} L={L} {S}x{S}: two fits of {steps} steps back to back {seq * 1e3:.1f} ms, on two streams {par * 1e3:.1f} ms  (ratio {par / seq:.3f})", flush=True)
