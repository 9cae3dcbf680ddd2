#!/usr/bin/env python3
"""Diagnostic (GPU box): run the bench workload's fit repeatedly in one process and print, per fit, a 64-bit checksum of the
final parameters, the fg-IoU and the allocator addresses involved - to tell run-to-run nondeterminism from state leaking
between fits."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import awesome_amd as A
from awesome_amd.dataset import convex_blob_unaries
from awesome_amd.model import ConvexNextNet


def csum(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]


def main():
    n_fits = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    dev = torch.device("cuda", 0)
    props = torch.cuda.get_device_properties(0)
    print("device", props.name, "CUs", props.multi_processor_count, flush=True)
    S = 256
    spec = A.IcnnSpec(130, 2, 1)
    un = convex_blob_unaries(S, 0).reshape(1, -1).to(dev)
    torch.manual_seed(0)
    init = ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=1).flat_parameters()[None].to(dev)
    grid = A.Grid.linspace(S, S, dev)
    keep = None
    for k in range(n_fits):
        res = A.fit(spec, init.clone(), grid, un, E, lr=2e-3, record_loss=True, want_logits=True)
        iou = A.miou((torch.sigmoid(res.logits) > 0.5).float(), (un > 0.5).float(), invert=True)
        hist = res.loss_hist[0].cpu()
        if keep is None:
            keep = hist
        diff = (hist != keep).nonzero()
        first = int(diff[0]) if diff.numel() else -1
        print(f"fit {k:2d} params {csum(res.params)} iou {float(iou[0]):.5f} loss_end {float(hist[-1]):.8e} "
              f"first_step_differing_from_fit0 {first} params_ptr {res.params.data_ptr():#x}", flush=True)
    ws = A.icnn.step_only(spec, init, grid, un, 5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    A.icnn.step_only(spec, init, grid, un, 200, workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    print(f"step kernel back to back: {e0.elapsed_time(e1) / 200 * 1e3:.2f} us", flush=True)
    # the queued (no sync between fits) form the bench uses
    outs = [A.fit(spec, init.clone(), grid, un, E, lr=2e-3, record_loss=False, want_logits=True) for _ in range(6)]
    torch.cuda.synchronize()
    print("queued:", [csum(o.params) for o in outs])


if __name__ == "__main__":
    main()
