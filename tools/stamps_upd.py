#!/usr/bin/env python3
"""Where the update kernel's time goes (-DINR_STAMPS=1 build, variants/libinrfit_stamps.so): s_memrealtime stamps (100 MHz) of every
block of the LAST update launch of an optimisation sequence against the workgroup stamps of the step kernel launch in front of it."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["INRFIT_LIB"] = os.path.join(ROOT, "variants", "libinrfit_stamps.so")
import numpy as np
import torch
import awesome_amd as A
from awesome_amd.dataset import convex_blob_unaries
from awesome_amd.model import ConvexNextNet
dev = torch.device("cuda:0")
spec = A.IcnnSpec(130, 2, 1)
S = 256
torch.manual_seed(0)
params = ConvexNextNet().flat_parameters().to(dev)[None].contiguous()
un = convex_blob_unaries(S, 0).reshape(1, -1).to(dev)
grid = A.Grid.linspace(S, S, dev)
lib = A._lib.load()
wg = (C.c_ulonglong * 4096)()
up = (C.c_ulonglong * 2048)()
lib.inrfit_debug_wgtimes.argtypes = [C.POINTER(C.c_ulonglong)]
lib.inrfit_debug_updtimes.argtypes = [C.POINTER(C.c_ulonglong)]
us = lambda x: x / 100.0
for rep in range(3):
    A.fit(spec, params.clone(), grid, un, 40, record_loss=False, want_logits=False)
    torch.cuda.synchronize()
    lib.inrfit_debug_wgtimes(wg)      # (also clears)
    A.fit(spec, params.clone(), grid, un, 40, record_loss=False, want_logits=False)
    torch.cuda.synchronize()
    assert lib.inrfit_debug_wgtimes(wg) == 0 and lib.inrfit_debug_updtimes(up) == 0
    t = np.frombuffer(wg, dtype=np.uint64).reshape(1024, 4).astype(np.int64)
    t = t[t[:, 3] > 0]
    u = np.frombuffer(up, dtype=np.uint64).reshape(512, 4).astype(np.int64)
    u = u[u[:, 0] > 0]
    step_end = t[:, 3].max()            # (atomicMax over the launches: the last step launch's stores done)
    step_first_end = t[:, 3].min()
    print(f"rep {rep}: {len(t)} step workgroups, {len(u)} update blocks")
    print(f"   step kernel: workgroups' stores done over {us(step_end - step_first_end):5.2f} us")
    print(f"   update entry after the last step store : first {us(u[:, 0].min() - step_end):5.2f}  median {us(np.median(u[:, 0]) - step_end):5.2f}  last {us(u[:, 0].max() - step_end):5.2f} us")
    for k, name in ((1, "slab loads back   "), (2, "reduced (LDS, sum)"), (3, "stores done       ")):
        d = u[:, k] - u[:, 0]
        d = d[u[:, k] > 0]
        print(f"   entry -> {name}: median {us(np.median(d)):5.2f}  max {us(d.max()):5.2f} us   ({len(d)} blocks)")
    print(f"   last step store -> last update store: {us(u[:, 3].max() - step_end):5.2f} us ; first update entry -> last update store: {us(u[:, 3].max() - u[:, 0].min()):5.2f} us")
