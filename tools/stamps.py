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
