#!/usr/bin/env python3
"""Per-phase cycle sums of the L = 2 step kernel in a -DINR_STAMPS=1 build (workgroup 0 / wave 0 of the last launch):
    bash tools/build_variant.sh stamps -DINR_STAMPS=1 && python tools/stamps2.py [size] [dx]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["INRFIT_LIB"] = os.path.join(ROOT, "variants", "libinrfit_stamps.so")
import torch
import awesome_amd as A
from awesome_amd.dataset import convex_blob_unaries
from awesome_amd.model import ConvexNextNet
dev = torch.device("cuda:0")
spec = A.IcnnSpec(130, 2, 2)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
params = ConvexNextNet(n_hidden_layers=2).flat_parameters().to(dev)[None].contiguous()
un = convex_blob_unaries(S, 0).reshape(1, -1).to(dev)
grid = A.Grid.linspace(S, S, dev)
A.icnn.step_only(spec, params, grid, un, 20)
torch.cuda.synchronize()
lib = A._lib.load()
buf = (C.c_ulonglong * 16)()
lib.inrfit_debug_stamps2.argtypes = [C.POINTER(C.c_ulonglong)]
assert lib.inrfit_debug_stamps2(buf) == 0
names = ["loop top (point loads, prefetch)", "z0 + layer-1 product + relu", "barrier A (W2 refetch landed)", "layer-2 product + output layer",
         "loss, dz2, backward through layer 2", "barrier B", "stage (dz2 | z1)", "backward through layer 1 + layer-0 grads (+ DX)",
         "barrier C", "dW2 phase", "barrier D", "z0 rebuild + stage (dz1 | z0)", "barrier E + dW1 phase", "barrier F"]
tot = buf[14]
mf = [0, 8 + 264, 0, 264, 264, 0, 0, 264 + 8 + 32, 0, 288, 0, 8, 288, 0]
for k, n in enumerate(names):
    extra = f"  ({mf[k]} MFMAs x 4 chunks = {mf[k] * 4 * 32} pipe cycles)" if mf[k] else ""
    print(f"{n:48s} {buf[k]:9d} cycles  {100.0 * buf[k] / tot:5.1f}%{extra}")
print(f"{'loop total':48s} {tot:9d} cycles")
