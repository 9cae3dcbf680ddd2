"""Determinism of the HIP fit and its failure path (item 8).

What round 1 got wrong: the same source built with two sets of code-generation switches rounded some VALU expressions
differently (clang's default -ffp-contract=fast decides per build which a*b+c it fuses), and 2000 chaotic optimizer steps turned
that last-bit difference into masks 4 pixels apart (fg-IoU 0.99866 vs 0.99799, both within the +-1e-3 bar).  The library is now
built with -ffp-contract=off (arithmetic = what the source says) and reports its flags; these tests pin the rest: no kernel
reads a byte nobody wrote, repeated fits are bit-identical, and the one remaining source of rounding differences - how many
gradient slabs an image is split into - moves the result by a bounded amount."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

S, E = 256, 2000


@pytest.fixture(scope="module")
def amd():
    import awesome_amd
    awesome_amd._lib.load()
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return awesome_amd


@pytest.fixture()
def poison(amd):
    amd._lib.POISON = True
    yield
    amd._lib.POISON = False


def _problem(amd, layers=1, seeds=(0,)):
    from awesome_amd.dataset import convex_blob_unaries
    from awesome_amd.model import ConvexNextNet
    dev = torch.device("cuda:0")
    un = torch.stack([convex_blob_unaries(S, s).reshape(-1) for s in seeds]).to(dev)
    init = []
    for s in seeds:
        torch.manual_seed(s)
        m = ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=layers)
        init.append(m.flat_parameters())
    return m.spec, torch.stack(init).to(dev), amd.Grid.linspace(S, S, dev), un


def _mask_iou(amd, res, un):
    mask = torch.sigmoid(res.logits) > 0.5
    return mask, amd.miou(mask.float(), (un > 0.5).float(), invert=True)


def test_build_pins_the_arithmetic(amd):
    info = amd._lib.build_info()
    assert "-ffp-contract=off" in info and "gfx950" in info, info
    assert amd._lib.load().inrfit_slabs_per_image(S * S, 1) == 256
    assert amd._lib.load().inrfit_slabs_per_image(S * S, 64) == 4
    assert amd._lib.load().inrfit_slabs_per_image(100, 1) == 2       # never more slabs than 64-point chunks


@pytest.mark.parametrize("layers", [1, 2])
def test_full_fit_bitwise_with_poisoned_workspaces(amd, poison, layers):
    """BASELINE configs[1] (and its L = 2 variant): the 2000-step fit twice with every uninitialised host-side allocation
    (workspace = parameter images + gradient slabs + loss coefficients, loss history, logits) pre-filled with NaN, and once more
    without: identical bits, all finite."""
    spec, init, grid, un = _problem(amd, layers)
    runs = []
    for k in range(3):
        amd._lib.POISON = k < 2
        res = amd.fit(spec, init.clone(), grid, un, E, lr=2e-3, record_loss=True, want_logits=True)
        assert int(res.status[0]) == 0
        assert torch.isfinite(res.params).all() and torch.isfinite(res.loss_hist).all() and torch.isfinite(res.logits).all()
        runs.append(res)
    for r in runs[1:]:
        assert torch.equal(r.params, runs[0].params)
        assert torch.equal(r.loss_hist, runs[0].loss_hist)
        assert torch.equal(r.logits, runs[0].logits)
        assert torch.equal(r.opt_state, runs[0].opt_state)


def test_flow_priors_bitwise_with_poisoned_workspaces(amd, poison):
    """The two path-connected priors (coupling flow, RealNVP) through their fused fits, 64x64, 60 steps, poisoned scratch."""
    from awesome_amd.dataset import convex_blob_unaries
    from awesome_amd.model import ConvexDiffeomorphismNet, real_nvp_path_connected_net
    dev = torch.device("cuda:0")
    un = (convex_blob_unaries(256, 3).reshape(256, 256)[::4, ::4] > 0.5).float().reshape(1, -1).to(dev)
    grid = amd.Grid.linspace(64, 64, dev)
    for make in (lambda: ConvexDiffeomorphismNet(n_hidden=130, n_hidden_layers=2, nf_layers=6, nf_hidden=130,
                                                 diffeo_args=dict(backbone="normal_block")),
                 lambda: real_nvp_path_connected_net(channels=2, hidden_units=32, flow_n_flows=12, flow_output_fn="tanh")):
        outs = []
        for k in range(2):
            torch.manual_seed(5)
            m = make().to(dev)
            rep = m.fit_images(grid, un, num_epochs=60)
            sd = {k_: v.detach().clone() for k_, v in m.state_dict().items()}
            assert all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())
            outs.append(sd)
        for k_ in outs[0]:
            assert torch.equal(outs[0][k_], outs[1][k_]), k_


_UNION_SNIPPET = r"""
import hashlib, sys, torch
sys.path.insert(0, %r)
import awesome_amd as amd
from awesome_amd.dataset import convex_blob_unaries
from awesome_amd.model import ConvexDiffeomorphismNet, real_nvp_path_connected_net
dev = torch.device("cuda:0")
un = (convex_blob_unaries(256, 3).reshape(256, 256)[::4, ::4] > 0.5).float().reshape(1, -1).to(dev)
bad = un.clone(); bad[0, 7] = float("nan")
grid = amd.Grid.linspace(64, 64, dev)
for make in (lambda: ConvexDiffeomorphismNet(n_hidden=130, n_hidden_layers=2, nf_layers=6, nf_hidden=130, diffeo_args=dict(backbone="normal_block")),
             lambda: real_nvp_path_connected_net(channels=2, hidden_units=32, flow_n_flows=12, flow_output_fn="tanh"),
             lambda: real_nvp_path_connected_net(channels=2, hidden_units=32, flow_n_flows=12, flow_output_fn="tanh", convex_net_hidden_layers=1)):
    for targets in (un, bad):
        torch.manual_seed(5)
        m = make().to(dev)
        res = m.fit_images(grid, targets, num_epochs=40)
        h = hashlib.sha1()
        for t in (res.icnn_params, res.flow_params, res.icnn_opt_state, res.flow_opt_state, res.loss_hist, res.status):
            h.update(t.detach().cpu().numpy().tobytes())
        print("CK", h.hexdigest(), int(res.status[0]))
"""


def test_one_launch_for_both_updates_equals_the_two_launches(amd):
    """cdn_update_kernel / pcn_update_kernel (ICNN update + the deformation's update in one launch, the deformation's blocks deriving the
    non-finite-loss decision from the slabs themselves) against the two separate launches (INRFIT_SPLIT_UPDATES=1): parameters, both
    optimizer states, loss curves and status bit for bit - on a healthy image and on one whose targets hold a NaN (frozen at step 1)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for split in (False, True):
        env = dict(os.environ)
        env.pop("INRFIT_SPLIT_UPDATES", None)
        if split:
            env["INRFIT_SPLIT_UPDATES"] = "1"
        r = subprocess.run([sys.executable, "-c", _UNION_SNIPPET % root], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("CK")])
    assert len(outs[0]) == 6 and outs[0] == outs[1], outs
    assert [l.split()[2] for l in outs[0]] == ["0", "1"] * 3     # the NaN image is reported and frozen on both paths


def test_slab_count_moves_the_result_by_rounding_only(amd):
    """The gradient of one image is summed over 256 / n_images workgroup slabs in fixed order; another slab count is another
    (equally valid) summation order.  One step: gradients agree to fp32 rounding.  Whole fit: the masks stay within the
    north-star tolerance of each other (fg-IoU +-1e-3) and of the reference's fit."""
    spec, init, grid, un = _problem(amd)
    lib = amd._lib.load()
    out = {}
    try:
        for base in (256, 128, 32):
            assert lib.inrfit_debug_set_slab_base(base) == 0
            assert lib.inrfit_slabs_per_image(S * S, 1) == base
            loss, grad = amd.loss_grad(spec, init, grid, un)
            res = amd.fit(spec, init.clone(), grid, un, E, lr=2e-3, record_loss=False, want_logits=True)
            mask, iou = _mask_iou(amd, res, un)
            out[base] = (float(loss[0]), grad[0].cpu().numpy(), mask[0].cpu().numpy(), float(iou[0]))
    finally:
        lib.inrfit_debug_set_slab_base(0)
    l0, g0, m0, i0 = out[256]
    for base in (128, 32):
        l, g, m, i = out[base]
        assert l == pytest.approx(l0, rel=2e-6)
        np.testing.assert_allclose(g, g0, rtol=2e-4, atol=2e-6 * float(np.abs(g0).max()))
        assert abs(i - i0) <= 1e-3, (base, i, i0)
        assert int((m != m0).sum()) <= 0.002 * S * S, (base, int((m != m0).sum()))


def test_nonfinite_loss_freezes_the_image_and_is_surfaced(amd):
    """Failure path (reference: ValueError("Loss is nan or inf!"), path_connected_net.py:232,374; torch_agent.py:484-487).
    A learning rate of 3e38 overflows the first Adam step (lr / (1 - beta1) = inf), so the second forward is non-finite:
    status = NONFINITE, the parameters stay exactly what the one good step left, later steps change nothing, the other image
    of a batch is unaffected, and the fitter raises."""
    from awesome_amd.fitter import BatchedPriorFitter, NonFiniteLossError
    from awesome_amd.model import ConvexNextNet
    spec, init, grid, un = _problem(amd, seeds=(0, 1))
    one = amd.fit(spec, init[:1].clone(), grid, un[:1], 1, lr=3e38, record_loss=True, want_logits=False)
    assert int(one.status[0]) == 0 and np.isfinite(float(one.loss_hist[0, 0]))
    res = amd.fit(spec, init[:1].clone(), grid, un[:1], 6, lr=3e38, record_loss=True, want_logits=False)
    assert int(res.status[0]) == 1
    h = res.loss_hist[0].cpu().numpy()
    assert np.isfinite(h[0]) and not np.isfinite(h[1:]).any()
    bits = lambda t: t.contiguous().view(torch.int32)               # the overflowed parameters hold inf / NaN: compare bit patterns
    assert not torch.isfinite(one.params).all()
    assert torch.equal(bits(res.params), bits(one.params))          # frozen at the parameters before the bad step
    assert torch.equal(bits(res.opt_state[:, :2 * spec.n_params]), bits(one.opt_state[:, :2 * spec.n_params]))
    # NaN in one image's unaries: that image never moves, its batch neighbour fits as if alone
    un_bad = un.clone()
    un_bad[0, 1234] = float("nan")
    both = amd.fit(spec, init.clone(), grid, un_bad, 50, lr=2e-3, record_loss=False, want_logits=False)
    assert both.status.cpu().tolist() == [1, 0]
    assert torch.equal(both.params[0], init[0])
    amd._lib.load().inrfit_debug_set_slab_base(128)                 # image 1 alone with the slab count it has in a batch of 2
    try:
        alone = amd.fit(spec, init[1:].clone(), grid, un[1:], 50, lr=2e-3, record_loss=False, want_logits=False)
    finally:
        amd._lib.load().inrfit_debug_set_slab_base(0)
    assert torch.equal(both.params[1], alone.params[0])
    # the flow priors freeze BOTH their parameter sets (ICNN and deformation) the same way
    from awesome_amd import flow as FL, rnvp as R
    from awesome_amd.model import ConvexDiffeomorphismNet, real_nvp_path_connected_net
    g64 = amd.Grid.linspace(64, 64, torch.device("cuda:0"))
    u64 = un[:1, ::16].contiguous().clone()
    u64[0, 7] = float("nan")
    torch.manual_seed(4)
    cdn = ConvexDiffeomorphismNet(n_hidden=64, n_hidden_layers=1, nf_layers=4, nf_hidden=24, diffeo_args=dict(backbone="normal_block"))
    ispec, fspec = cdn._specs()
    flat = cdn._engine_pack(cdn.state_dict()).to("cuda:0")[None]
    ip, fp = flat[:, :ispec.n_params].contiguous(), flat[:, ispec.n_params:].contiguous()
    r = FL.cdn_fit(ispec, fspec, ip.clone(), fp.clone(), g64, u64, 5)
    assert int(r.status[0]) == 1 and torch.equal(r.icnn_params, ip) and torch.equal(r.flow_params, fp)
    pcn = real_nvp_path_connected_net(channels=2, hidden_units=16, flow_n_flows=4, flow_output_fn="tanh", convex_net_hidden_units=64)
    ispec, rspec = pcn._specs()
    flat = pcn._engine_pack(pcn.state_dict()).to("cuda:0")[None]
    ip, fp = flat[:, :ispec.n_params].contiguous(), flat[:, ispec.n_params:].contiguous()
    R.actnorm_init(rspec, fp, g64)
    r = R.pcn_fit(ispec, rspec, ip.clone(), fp.clone(), g64, u64, 5)
    assert int(r.status[0]) == 1 and torch.equal(r.icnn_params, ip) and torch.equal(r.flow_params, fp)
    # host reaction
    torch.manual_seed(0)
    fitter = BatchedPriorFitter(lambda: ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=1), num_epochs=20,
                                proper_prior_fit_retrys=0)
    with pytest.raises(NonFiniteLossError, match="Loss is nan or inf") as ei:
        fitter.fit_batch(grid, un_bad)
    assert ei.value.images == [0] and int(ei.value.report.status[1]) == 0
    fitter.on_nonfinite = "report"
    rep = fitter.fit_batch(grid, un_bad)
    assert rep.status.cpu().tolist() == [1, 0]


def test_configs2_batch_of_64_matches_single_fits_and_reference(amd, golden_dir):
    """BASELINE configs[2], one GPU's share: 64 independent 256x256 blobs fitted in ONE call (4 gradient slabs per image).

    * Batch independence is exact: image k of the batch is bit-identical to the single-image fit run with the same slab count,
      and does not depend on which images are its neighbours.
    * Against the single-image fit at its own slab count (256) only the summation order of the gradients differs.  The fit is
      a chaotic iteration (Adam, relu kinks): the relative loss difference grows from 1 ulp at step 10 to ~1e-6 at step 100 and
      ~1e-2 at step 2000, the same growth the reference's own CPU fit shows against itself (tests/golden/PROVENANCE.txt).  On
      top of that full-batch Adam at lr 2e-3 produces loss spikes (within the last 200 steps the loss moves by a factor 1.14 in
      the median image, up to 2.5), and a fit that happens to end inside a spike has a visibly worse mask: measured over these 64
      images between two summation orders, |dIoU| has median 1e-4 and a tail up to 1.8e-2 (image 43), whichever order is
      taken as the base.  So the bars HERE (HIP against HIP, two summation orders) are: identical early trajectory; per-image
      fg-IoU within 1e-3 in the median and 2.5e-2 at worst; and the DATASET mIoU - the number the reference reports and north_star
      bounds by +-1e-3 - within 1e-3 (measured: 0.99849 vs 0.99816).  The bars against the REFERENCE (its own 16 fits of these images,
      run twice; per-image bar = its own run-to-run floor x 2) are in tests/test_gpu_scale_parity.py.
    * Image 0 against the reference classes' own fit of the same problem (golden fit_blob256_reference.npz)."""
    import os
    seeds = tuple(range(64))
    spec, init, grid, un = _problem(amd, seeds=seeds)
    res = amd.fit(spec, init.clone(), grid, un, E, lr=2e-3, record_loss=True, want_logits=True)
    assert int(res.status.sum()) == 0
    mask, iou = _mask_iou(amd, res, un)
    assert float(iou.min()) > 0.98                                    # every blob is fitted
    z = np.load(os.path.join(golden_dir, "fit_blob256_reference.npz"))
    np.testing.assert_allclose(res.loss_hist[0, :100].cpu().numpy(), z["losses"][:100], rtol=5e-4)
    assert abs(float(res.loss_hist[0, -1]) - float(z["losses"][-1])) <= 0.1 * float(z["losses"][-1])
    # (per-image parity with the reference is asserted on spike-robust statistics of the last 50 steps, which the reference's own runs
    #  agree on to 3e-4 - tests/test_gpu_scale_parity.py; its end-of-fit snapshots differ by up to 6.5e-3 between two of its own runs)
    iou_single = []
    for k in range(64):
        single = amd.fit(spec, init[k:k + 1].clone(), grid, un[k:k + 1], E, lr=2e-3, record_loss=True, want_logits=True)
        m1, i1 = _mask_iou(amd, single, un[k:k + 1])
        iou_single.append(float(i1[0]))
        assert abs(float(i1[0]) - float(iou[k])) <= 2.5e-2, (k, float(i1[0]), float(iou[k]))
        np.testing.assert_allclose(single.loss_hist[0, :50].cpu().numpy(), res.loss_hist[k, :50].cpu().numpy(), rtol=2e-4)
    assert abs(float(np.mean(iou_single)) - float(iou.mean())) <= 1e-3, (float(np.mean(iou_single)), float(iou.mean()))
    assert float(np.median(np.abs(np.asarray(iou_single) - iou.cpu().numpy()))) <= 1e-3
    # exact batch independence: the same slab count alone, and other neighbours
    lib = amd._lib.load()
    lib.inrfit_debug_set_slab_base(4)
    try:
        for k in (0, 63):
            alone = amd.fit(spec, init[k:k + 1].clone(), grid, un[k:k + 1], E, lr=2e-3, record_loss=False, want_logits=False)
            assert torch.equal(alone.params[0], res.params[k]), k
    finally:
        lib.inrfit_debug_set_slab_base(0)
    again = amd.fit(spec, init.clone(), grid, un, 200, lr=2e-3, record_loss=False, want_logits=False)
    un2, init2 = un.clone(), init.clone()
    un2[1:], init2[1:] = un[1:].flip(0), init[1:].flip(0)
    shuffled = amd.fit(spec, init2, grid, un2, 200, lr=2e-3, record_loss=False, want_logits=False)
    assert torch.equal(again.params[0], shuffled.params[0])
    assert torch.equal(again.params[1:], shuffled.params[1:].flip(0))
