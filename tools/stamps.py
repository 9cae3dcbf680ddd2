#!/usr/bin/env python3
"""Read the per-phase cycle sums of a -DINR_STAMPS=1 build (workgroup 0 / wave 0 of the last launch)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["INRFIT_LIB"] = os.path.join(ROOT, "variants", "libinrfit_stamps.so")
import torch
import awesome_amd as A
from awesome_amd.dataset import convex_blob_unaries
from awesome_amd.model import ConvexNextNet
dev = torch.device("cuda:0")
spec = A.IcnnSpec(130, 2, 1)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
params = ConvexNextNet().flat_parameters().to(dev)[None].contiguous()
un = convex_blob_unaries(S, 0).reshape(1, -1).to(dev)
grid = A.Grid.linspace(S, S, dev)
A.icnn.step_only(spec, params, grid, un, 20)
torch.cuda.synchronize()
lib = A._lib.load()
buf = (C.c_ulonglong * 16)()
lib.inrfit_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
assert lib.inrfit_debug_stamps(buf) == 0
names = ["loop top -> (prefetch, xe)", "z0[TM] tables", "fwd GEMM", "epilogue (relu,y,dy)", "bwd GEMM (+dz1 tiles, staging)",
         "mask + layer-0 grads", "barrier 1", "dW phase", "barrier 2"]
tot = buf[10]
for k, n in enumerate(names):
    print(f"{n:36s} {buf[k]:9d} cycles  {100.0 * buf[k] / tot:5.1f}%")
print(f"{'loop total':36s} {tot:9d} cycles ; epilogue after loop {buf[11]} cycles")
if buf[12]:
    print(f"forward k-groups 2..6: {buf[12]} cycles = {buf[12] / (4 * 5 * 33):.2f} cycles per MFMA (4 chunks x 5 k-groups x 33)")

# ---- per-workgroup wall clock of ONE launch (s_memrealtime, 100 MHz) against the HIP-event duration of the same launch ----
import numpy as np
wg = (C.c_ulonglong * 4096)()
lib.inrfit_debug_wgtimes.argtypes = [C.POINTER(C.c_ulonglong)]
lib.inrfit_debug_wgtimes(wg)                      # clear
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(); A.icnn.step_only(spec, params, grid, un, 1); e1.record()
    torch.cuda.synchronize()
    assert lib.inrfit_debug_wgtimes(wg) == 0
    t = np.frombuffer(wg, dtype=np.uint64).reshape(1024, 4).astype(np.int64)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    us = lambda x: x / 100.0
    print(f"launch {rep}: {len(t)} workgroups, HIP events {e0.elapsed_time(e1) * 1e3:7.2f} us | first entry -> last store done {us(t[:, 3].max() - t0):6.2f} us")
    print(f"   entry skew (last - first workgroup start)   {us(t[:, 0].max() - t0):6.2f} us")
    print(f"   prologue (entry -> loop)        median {us(np.median(t[:, 1] - t[:, 0])):6.2f}  max {us((t[:, 1] - t[:, 0]).max()):6.2f} us")
    print(f"   chunk loop                      median {us(np.median(t[:, 2] - t[:, 1])):6.2f}  max {us((t[:, 2] - t[:, 1]).max()):6.2f} us")
    print(f"   epilogue (loop end -> stores)   median {us(np.median(t[:, 3] - t[:, 2])):6.2f}  max {us((t[:, 3] - t[:, 2]).max()):6.2f} us")
    print(f"   end skew (last - first workgroup end)       {us(t[:, 3].max() - t[:, 3].min()):6.2f} us")

# ---- the same stamps for the LAST step launch of an optimisation sequence (step, update, step, update, ...) ----
def report(tag):
    assert lib.inrfit_debug_wgtimes(wg) == 0
    t = np.frombuffer(wg, dtype=np.uint64).reshape(1024, 4).astype(np.int64)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    us = lambda x: x / 100.0
    print(f"{tag}: first entry -> last store done {us(t[:, 3].max() - t0):6.2f} us | entry skew {us(t[:, 0].max() - t0):5.2f} | prologue med {us(np.median(t[:, 1] - t[:, 0])):5.2f} max {us((t[:, 1] - t[:, 0]).max()):5.2f}"
          f" | loop med {us(np.median(t[:, 2] - t[:, 1])):6.2f} max {us((t[:, 2] - t[:, 1]).max()):6.2f} | epilogue med {us(np.median(t[:, 3] - t[:, 2])):5.2f} max {us((t[:, 3] - t[:, 2]).max()):5.2f}"
          f" | end skew {us(t[:, 3].max() - t[:, 3].min()):5.2f}")
for rep in range(3):
    A.icnn.step_only(spec, params, grid, un, 20); torch.cuda.synchronize(); report("back-to-back (20th launch)")
    pr = params.clone()
    A.fit(spec, pr, grid, un, 20, record_loss=False, want_logits=False); torch.cuda.synchronize(); report("in sequence   (20th step)  ")
