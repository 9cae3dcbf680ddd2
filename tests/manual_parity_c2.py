"""Manual (slow, not collected by pytest): BASELINE configs[1] end to end on the CPU oracle - 2000 full-batch Adam steps of
ConvexNextNet(h=130, L=1) on the 256x256 blob, seeded exactly like bench.py rank 0 - for the mIoU-vs-reference figure in DESIGN.md.

    python tests/manual_parity_c2.py --save tests/golden/c2_oracle_fit2000.npz   # ~2-4 min on 8 threads: oracle fit, mask + losses
    python tests/manual_parity_c2.py --load tests/golden/c2_oracle_fit2000.npz --hip   # on a GPU box: HIP fit vs the saved oracle result
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import inr_oracle as O
from awesome_amd.dataset import convex_blob_unaries
from awesome_amd.model import ConvexNextNet

S, E = 256, 2000
torch.manual_seed(0)
m = ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=1)
sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
un = convex_blob_unaries(S, 0)
grid = O.positional_grid(S, S)[None]
t0 = time.time()
if "--load" in sys.argv:
    z = np.load(sys.argv[sys.argv.index("--load") + 1])
    mask_o, losses = torch.from_numpy(z["mask"]), z["losses"].tolist()
else:
    p, losses, logits = O.fit_icnn(sd, grid, un[None, None], E, lr=2e-3, loss_kind="se", optimizer="adam")
    mask_o = (torch.sigmoid(logits) > 0.5).float().reshape(-1)
    if "--save" in sys.argv:
        np.savez_compressed(sys.argv[sys.argv.index("--save") + 1], mask=mask_o.numpy(), losses=np.asarray(losses, np.float32))
iou_o = O.miou_binary(mask_o.reshape(1, -1), (un.reshape(1, -1) > 0.5).float())
print(f"oracle: {time.time() - t0:.0f} s, final loss {losses[-1]:.6f}, fg-mIoU vs unaries {iou_o:.5f}", flush=True)
if "--hip" in sys.argv:
    import awesome_amd as A
    dev = torch.device("cuda:0")
    res = A.fit(m.spec, m.flat_parameters()[None].to(dev), A.Grid.linspace(S, S, dev), un.reshape(1, -1).to(dev), E, lr=2e-3)
    mask_h = (torch.sigmoid(res.logits[0]).cpu() > 0.5).float()
    iou_h = float(A.miou(mask_h[None].to(dev), (un.reshape(1, -1) > 0.5).float().to(dev))[0])
    print(f"hip: final loss {float(res.loss_hist[0, -1]):.6f}, fg-mIoU vs unaries {iou_h:.5f}, |d mIoU| {abs(iou_h - iou_o):.2e}, "
          f"pixels that differ {int((mask_h != mask_o).sum())} of {S * S}")
