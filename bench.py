#!/usr/bin/env python3
"""bench.py - INR-fits/sec on synthetic 256x256 grids with the convexity prior (BASELINE.json metric).

One "step" = one complete cold fit of the workload on every GPU:
    ConvexNextNet(n_hidden=130, n_hidden_layers=1, in_features=2), 256x256 linspace grid, binary unaries of a seeded
    convex blob (fg = 0), loss mean((sigmoid(f) - u)^2), Adam(lr=2e-3), clamp after every step, E = 2000 full-batch
    steps, fp32  (notebooks/how_to/convexity.ipynb cells 7-9; path_connected_net.py:756 num_epochs).
Workload per GPU: BASELINE configs[1] - a single image (`--images-per-gpu 1`, the default).  Every rank fits its own
images (no data-path collective; "weak" scaling), RCCL is used for the barrier / max-time reduction only.

    python bench.py [--gpus N --steps K --warmup W]
N > 1: one process per GPU.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment: how the driver starts it) this
process is one rank; started plainly with --gpus N > 1 it launches `python -m torch.distributed.run --nproc-per-node N bench.py ...`
as a child BEFORE touching the GPU, relays rank 0's JSON line and exits with the child's code.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (icnn_step_kernel): algorithmic FLOPs per launch
(105,312 flop/point/step x 65,536 points x images) / its average launch duration measured live with events on the launch
stream.  `cpu_baseline` times the CPU oracle (torch, all host threads) on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FWD_FLOP_PER_POINT = 35104           # 2h^2 + 10h + 4, h = 130, C = 2, L = 1
STEP_FLOP_PER_POINT = 3 * FWD_FLOP_PER_POINT   # forward + backward(dX) + backward(dW)  
PEAK_FP32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32, dense


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed fits per GPU")
    ap.add_argument("--warmup", type=int, default=1, help="untimed fits per GPU")
    ap.add_argument("--epochs", type=int, default=2000, help="optimizer steps per fit (E)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--images-per-gpu", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra 'variants' timings (other priors of the same path)")
    ap.add_argument("--cpu-sample-steps", type=int, default=60)
    ap.add_argument("--kernel-iters", type=int, default=200, help="step-kernel launches for the roofline timing")
    ap.add_argument("--throughput-images", type=int, default=64,
                    help="also time ONE full fit of this many images per GPU (BASELINE configs[2] = 512/8) and report it "
                         "under 'throughput_mode' (0 = skip); never part of `value`")
    return ap.parse_args()


def launch_ranks(args):
    """`--gpus N` (N > 1) without a torch.distributed environment: start the N ranks as a child process group.  This parent never
    imports torch or touches the GPU (a process that has initialised the GPU must not exec or fork workers), it only relays the
    child's output and exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout.splitlines():
        if line.startswith("{"):
            print(line, flush=True)
        elif line.strip():
            print(line, file=sys.stderr, flush=True)
    return proc.returncode


@contextlib.contextmanager
def _stdout_to_stderr():
    """File-descriptor level (C libraries included): everything written to stdout inside the block goes to stderr."""
    import ctypes
    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_BACKEND=gloo: rehearsal of the N > 1 code path on a box with fewer GPUs than ranks (ranks share devices and the
    # collectives run on CPU tensors); the real multi-GPU run uses "nccl" (= RCCL over xGMI on ROCm), one rank per GPU.
    # BENCH_REHEARSAL=1 (needs BENCH_BACKEND=gloo; tests/test_parallel_gloo.py): no GPU at all - the fit is replaced by a labelled
    # stand-in so that the launcher, the rendezvous and every collective of this file run on a CPU-only box.  Its line says
    # "rehearsal": true and carries no value: it is a plumbing check, never a measurement.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    rehearsal = os.environ.get("BENCH_REHEARSAL", "0") == "1"
    if rehearsal and backend != "gloo":
        raise SystemExit("[bench] BENCH_REHEARSAL=1 needs BENCH_BACKEND=gloo")
    if args.gpus != world:
        raise SystemExit(f"[bench] --gpus {args.gpus} but WORLD_SIZE {world}: start it as "
                         f"`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...` "
                         f"(or plainly, `python bench.py --gpus {args.gpus}`, which does that itself)")
    dist = None
    if rehearsal:
        dev = cdev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (the product has no CPU path)")
        if backend != "nccl":
            local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)           # before the process group: RCCL binds the communicator to this device
        dev = torch.device("cuda", local_rank)
        cdev = dev if backend == "nccl" else torch.device("cpu")   # where the (tiny) collective payloads live
    # BENCH_FORCE_DIST=1: create the process group and run every collective even at WORLD_SIZE 1 - the RCCL code path of this file
    # (device-bound init, barrier, all_reduce / all_gather on device tensors) on a one-GPU box.
    if world > 1 or os.environ.get("BENCH_FORCE_DIST", "0") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        kw = dict(device_id=dev) if (backend == "nccl" and not rehearsal) else {}
        with _stdout_to_stderr():   # RCCL prints a version banner on stdout when the communicator is created; stdout carries the JSON line only
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
            (dist.barrier(device_ids=[dev.index]) if kw else dist.barrier())

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    def barrier():
        sync()
        if dist is not None:
            dist.barrier(device_ids=[dev.index]) if (backend == "nccl" and not rehearsal) else dist.barrier()
        sync()

    def max_over_ranks(x: float) -> float:
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(n: int) -> int:
        if dist is None:
            return n
        t = torch.tensor([n], dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item())

    def gather_cat(v):
        """the only data collective: a few bytes of per-image metrics"""
        if dist is None:
            return v
        vc = v.to(cdev)
        parts = [torch.zeros_like(vc) for _ in range(world)]
        dist.all_gather(parts, vc)
        return torch.cat(parts)

    S, E, B = args.size, args.epochs, args.images_per_gpu
    N = S * S
    # image i of rank r is blob seed r*B + i; initial weights: the reference's seeded default init
    seeds = [rank * B + i for i in range(B)]

    if rehearsal:
        import types

        def one_fit():
            time.sleep(0.01)
            return types.SimpleNamespace(status=torch.zeros(B, dtype=torch.int32), iou=torch.tensor([float(sd) for sd in seeds]))

        for _ in range(args.warmup):
            one_fit()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = one_fit()
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        iou_all = gather_cat(res.iou)
        bad = sum_over_ranks(int((res.status != 0).sum()))
        if rank == 0:
            print(json.dumps({"metric": "REHEARSAL of the rank launcher and collectives (no fit ran)", "value": None, "unit": "fits/s",
                              "rehearsal": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(elapsed / args.steps * 1e3, 3), "gathered_image_seeds": iou_all.tolist(),
                              "nonfinite_fits": bad, "backend": backend}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    import awesome_amd as A
    from awesome_amd.dataset import convex_blob_unaries
    from awesome_amd.model import ConvexNextNet

    spec = A.IcnnSpec(130, 2, 1)
    unaries = torch.stack([convex_blob_unaries(S, s).reshape(-1) for s in seeds]).to(dev)
    init = []
    for s in seeds:
        torch.manual_seed(s)
        init.append(ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=1).flat_parameters())
    init = torch.stack(init).to(dev)
    grid = A.Grid.linspace(S, S, dev)

    def one_fit():
        params = init.clone()
        return A.fit(spec, params, grid, unaries, E, lr=2e-3, loss="se", optimizer="adam", clamp=True, record_loss=False,
                     want_logits=True)

    res = None
    for _ in range(args.warmup):
        res = one_fit()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = one_fit()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    # quality of the last fit: fg-mIoU of the thresholded prior vs the unaries (the reference's IoU gate metric)
    prob = torch.sigmoid(res.logits)
    iou = A.miou((prob > 0.5).float(), (unaries > 0.5).float(), invert=True)
    status_bad = sum_over_ranks(int((res.status != 0).sum().item()))
    iou_all = gather_cat(iou)
    miou = float(iou_all.mean().item())

    # ---- roofline of the dominant kernel: average launch duration by HIP events on the launch stream ------------------------
    # In the timed region every optimizer step is two back-to-back launches (step kernel, update kernel; rocprofv3's kernel
    # trace of this command shows no gap between them).  One more fit of the same workload brackets every launch of both
    # kernels with its own pair of events (inrfit_timing_*); the event packets stretch the sequence a little, and that excess -
    # (step bracket + update bracket) minus the un-instrumented time per optimizer step measured above - is taken off the two
    # brackets in equal parts.  -> `kernel_us`, the step kernel's duration IN the optimisation sequence (parameter image just
    # rewritten by the update), which `frac` is quoted on.  A plain train of identical step-kernel launches between one pair of
    # events is ~3 us faster per launch (`kernel_us_back_to_back`).
    params = res.params
    ws = A.icnn.step_only(spec, params, grid, unaries, 5)           # warm
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    A.icnn.step_only(spec, params, grid, unaries, args.kernel_iters, workspace=ws)
    e1.record()
    torch.cuda.synchronize(dev)
    kernel_b2b_ms = e0.elapsed_time(e1) / args.kernel_iters
    n_seq = min(E, 500)
    with A.icnn.step_kernel_brackets(n_seq) as seq:
        A.fit(spec, init.clone(), grid, unaries, n_seq, lr=2e-3, loss="se", optimizer="adam", clamp=True, record_loss=False,
              want_logits=False)
    us_per_step = elapsed / args.steps / E * 1e6
    excess = max(seq.avg_us + seq.update_avg_us - us_per_step, 0.0)
    kernel_ms = (seq.avg_us - 0.5 * excess) * 1e-3
    update_us = seq.update_avg_us - 0.5 * excess
    flop_per_launch = STEP_FLOP_PER_POINT * N * B
    achieved = flop_per_launch / (kernel_ms * 1e-3) / 1e12
    # HBM bytes per launch of this kernel from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes of
    # this same command; profiles/r02_e_pmc_step_kernel.json) - only quoted for the configuration it was measured on
    traffic = None
    tf = os.path.join(ROOT, "profiles", "r02_e_pmc_step_kernel.json")
    if B == 1 and S == 256 and os.path.exists(tf):
        with open(tf) as f:
            traffic = json.load(f).get("traffic_bytes_per_launch")
    # what a launch of nothing but independent fp32 MFMAs on all 1024 SIMDs reaches on THIS box (~2.1 GHz under that load instead
    # of the 2.4 GHz the nominal peak assumes): the practical ceiling, reported next to `peak`, never instead of it
    mfma_stream = A.icnn.mfma_stream_tflops(dev)
    roofline = {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP32_MATRIX_TFLOPS, 4), "traffic": traffic,
                "kernel": "icnn_step_kernel<130,2,train>", "kernel_us": round(kernel_ms * 1e3, 2),
                "kernel_samples": seq.samples, "kernel_us_back_to_back": round(kernel_b2b_ms * 1e3, 2),
                "update_kernel_us": round(update_us, 2), "event_bracket_excess_us": round(excess, 2),
                "flop_per_launch": flop_per_launch, "mfma_only_stream_tflops": round(mfma_stream, 1),
                "frac_of_mfma_only_stream": round(achieved / mfma_stream, 4)}

    # ---- throughput mode (extra, not `value`): configs[2]'s per-GPU share, one complete E-step fit of a batch ---------
    thr = None
    if args.throughput_images > 0:
        TB = args.throughput_images
        tseeds = [1000 + rank * TB + i for i in range(TB)]
        tun = torch.stack([convex_blob_unaries(S, s).reshape(-1) for s in tseeds]).to(dev)
        tinit = init[:1].repeat(TB, 1).contiguous()   # same seeded init for every image (weights are per image on device)
        A.fit(spec, tinit.clone(), grid, tun, 20, lr=2e-3, record_loss=False, want_logits=False)  # warm
        barrier()
        t1 = time.perf_counter()
        tres = A.fit(spec, tinit.clone(), grid, tun, E, lr=2e-3, loss="se", optimizer="adam", clamp=True, record_loss=False,
                     want_logits=True)
        barrier()
        tdt = max_over_ranks(time.perf_counter() - t1)
        tiou = A.miou((torch.sigmoid(tres.logits) > 0.5).float(), (tun > 0.5).float(), invert=True)
        thr = {"images_per_gpu": TB, "fits_per_s": round(TB * world / tdt, 4), "seconds": round(tdt, 3),
               "roofline_frac_wall_clock": round(STEP_FLOP_PER_POINT * N * TB * E / tdt / 1e12 / PEAK_FP32_MATRIX_TFLOPS, 4),
               "nonfinite_fits": int((tres.status != 0).sum().item()), "min_iou_vs_unaries": round(float(tiou.min()), 5),
               "us_per_optimizer_step_per_image": round(tdt / E / TB * 1e6, 2), "miou_vs_unaries": round(float(tiou.mean()), 5),
               "note": "one complete E-step fit of the batch; all images step together (BASELINE configs[2] per-GPU share)"}

    # parity of the timed workload with the real reference classes: rank 0's image 0 is exactly the problem of the golden
    # fixture tests/golden/fit_blob256_reference.npz (tools/gen_golden.py gen_fit_blob256: same seeds, E = 2000, 256x256)
    reference_parity = {}
    fx = os.path.join(ROOT, "tests", "golden", "fit_blob256_reference.npz")
    if rank == 0 and S == 256 and E == 2000 and os.path.exists(fx):
        import numpy as np
        z = np.load(fx)
        mask0 = (prob[0] > 0.5).cpu().numpy().reshape(-1)
        reference_parity = {"reference_miou": round(float(z["final_miou"]), 5),
                            "miou_abs_diff_vs_reference": round(abs(float(iou[0]) - float(z["final_miou"])), 6),
                            "mask_pixels_differing_from_reference": int((mask0 != z["final_mask"].astype(bool)).sum())}
    # What the result depends on besides the inputs (the driver's fit differed from every builder-side one): the
    # build (compiler, code-generation flags; -ffp-contract=off pins the arithmetic to the source) and the number of gradient
    # slabs per image (a constant of the library, not the box's CU count).  `fit_checksum`: first 64 bits of the SHA-256 of rank
    # 0's final parameters of the LAST timed fit - every fit of a run, and every run of one build, prints the same value.
    import hashlib
    lib = A._lib.load()
    determinism = {"fit_checksum": hashlib.sha256(res.params.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16],
                   "slabs_per_image": int(lib.inrfit_slabs_per_image(N, B)),
                   "cu_count": int(torch.cuda.get_device_properties(dev).multi_processor_count),
                   "build": A._lib.build_info()}
    out = None
    if rank == 0:
        fits = args.steps * B * world
        value = fits / elapsed
        out = {
            "metric": "INR-fits/sec (256x256 grid, convexity prior)", "value": round(value, 4), "unit": "fits/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{B} x {S}x{S} synthetic convex blob per GPU, ConvexNextNet(h=130,L=1), "
                                   f"SE(sigmoid) mean, Adam lr 2e-3, clamp, E={E} full-batch steps (BASELINE configs[1])",
                       "images_per_gpu": B, "grid": f"{S}x{S}", "epochs_per_fit": E, "parallelism": f"dp{world} (independent fits)"},
            "miou_vs_unaries": round(miou, 5), "nonfinite_fits": status_bad, **reference_parity, **determinism,
            "us_per_optimizer_step": round(elapsed / args.steps / E * 1e6, 2),
            "roofline": roofline,
        }
        if thr is not None:
            out["throughput_mode"] = thr
        if world == 1 and not args.no_variants:
            out["variants"] = path_variants(dev, S)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, init[0].cpu(), unaries[0].cpu(), spec)
            out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        barrier()
        dist.destroy_process_group()


def path_variants(dev, S, steps=300):
    """Extra, never part of `value`: the other priors on the same path (one image / sequence on this GPU): ICNN L=2,
    ConvexDiffeomorphismNet, PathConnectedNet (RealNVP) on (x, y) and on (x, y, t).  Per variant: microseconds per optimizer
    step of the fused fit (wall clock over `steps` steps), a `roofline` object on the ALGORITHMIC flops of one step (3 x the
    forward flops of the ICNN and of the deformation, SURVEY.md 8d's convention; bound = the fp32 matrix/vector peak, which are
    the same number on this chip) and a `check`: the loss of the timed fit is finite and falls."""
    import torch
    import awesome_amd as A
    from awesome_amd.dataset import SyntheticSequenceDataset, convex_blob_unaries
    from awesome_amd.model import ConvexDiffeomorphismNet, ConvexNextNet, real_nvp_path_connected_net

    def icnn_fwd_flop(h, c, layers):          # 4h + L(2h^2 + 4h) + 2h + 4 for c = 2 (2hc + L(2h^2 + 2hc) + 2h + 2c in general)
        return 2 * h * c + layers * (2 * h * h + 2 * h * c) + 2 * h + 2 * c

    def entry(us, n_points, fwd_flop_icnn, fwd_flop_flow, hist):
        flop = 3.0 * (fwd_flop_icnn + fwd_flop_flow) * n_points
        tf = flop / (us * 1e-6) / 1e12
        h = hist.float().cpu()
        return {"us_per_step": us,
                "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(tf / PEAK_FP32_MATRIX_TFLOPS, 4), "flop_per_step": int(flop),
                             "flop_per_point_icnn_fwd": fwd_flop_icnn, "flop_per_point_flow_fwd": fwd_flop_flow,
                             "basis": "wall clock per optimizer step (all kernels of the step), not one kernel"},
                "check": {"loss_first": round(float(h[0]), 6), "loss_last": round(float(h[-1]), 6),
                          "ok": bool(torch.isfinite(h).all() and h[-1] < h[0])}}

    def timed(fn):
        fn(10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = fn(steps)
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / steps * 1e6, 1), res.loss_hist[0]

    torch.manual_seed(0)
    N = S * S
    un = convex_blob_unaries(S, 0).reshape(1, -1).to(dev)
    grid = A.Grid.linspace(S, S, dev)
    out = {"unit": "us per optimizer step", "steps": steps}
    m2 = ConvexNextNet(n_hidden=130, n_hidden_layers=2, in_features=2)
    p2 = m2.flat_parameters()[None].to(dev)
    us, h = timed(lambda n: A.fit(m2.spec, p2.clone(), grid, un, n, lr=2e-3, record_loss=True, want_logits=False))
    out[f"ConvexNextNet_L2_{S}x{S}"] = entry(us, N, icnn_fwd_flop(130, 2, 2), 0, h)
    cdn = ConvexDiffeomorphismNet(n_hidden=130, n_hidden_layers=2, nf_layers=6, nf_hidden=130,
                                  diffeo_args=dict(backbone="normal_block")).to(dev)   # the reference configs' form
    us, h = timed(lambda n: cdn.fit_images(grid, un, num_epochs=n))
    out[f"ConvexDiffeomorphismNet_K6_w130_L2_{S}x{S}"] = entry(us, N, icnn_fwd_flop(130, 2, 2), 6 * 8 * 130, h)   # 8 w flop per coupling
    pc2 = real_nvp_path_connected_net(channels=2, hidden_units=32, flow_n_flows=12, flow_output_fn="tanh").to(dev)
    us, h = timed(lambda n: pc2.fit_images(grid, un, num_epochs=n))
    out[f"PathConnectedNet_RealNVP_C2_F12_L2_{S}x{S}"] = entry(us, N, icnn_fwd_flop(130, 2, 2), 12 * 32 * 2 * 2 * 2, h)   # F hid 2 nets C 2
    ds = SyntheticSequenceDataset(1, 128, 16)
    g3, u3 = A.Grid.explicit(ds.coords().to(dev)), ds.batch([0]).to(dev)
    pc3 = real_nvp_path_connected_net(channels=3, hidden_units=32, flow_n_flows=18, flow_output_fn="tanh").to(dev)
    us, h = timed(lambda n: pc3.fit_images(g3, u3, num_epochs=n))
    out["PathConnectedNet_RealNVP_C3_F18_L2_128x128x16"] = entry(us, 128 * 128 * 16, icnn_fwd_flop(130, 3, 2), 18 * 32 * 2 * 3 * 2, h)
    return out


def cpu_baseline(args, flat0, unaries0, spec):
    """The CPU oracle (pure torch restatement of the reference loop, parity-pinned to the reference's golden vectors)
    timed on this host: `cpu_sample_steps` optimizer steps of the same 256x256 fit, extrapolated to E steps."""
    import torch
    import awesome_amd as A
    from oracle import inr_oracle as O   # the thing being timed as the CPU baseline (kind = "port")

    S, E = args.size, args.epochs
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    threads = max(1, min(cores, 16))   # the GPU box's CPU share for one GPU; more threads oversubscribe and run slower
    torch.set_num_threads(threads)
    p = A.unpack_params(spec, flat0)
    grid = O.positional_grid(S, S)[None]
    un = unaries0.reshape(1, 1, S, S)
    O.fit_icnn(p, grid, un, 3, lr=2e-3)   # warm-up
    n = args.cpu_sample_steps
    t0 = time.perf_counter()
    O.fit_icnn(p, grid, un, n, lr=2e-3)
    dt = time.perf_counter() - t0
    s_per_step = dt / n
    return {"value": round(1.0 / (s_per_step * E), 6), "unit": "fits/s", "cores": threads, "kind": "port",
            "sample": f"{n} of {E} optimizer steps of the same {S}x{S} fit ({dt:.1f} s, {s_per_step * 1e3:.1f} ms/step), "
                      f"torch {torch.__version__} CPU, extrapolated linearly",
            "ms_per_optimizer_step": round(s_per_step * 1e3, 2)}


if __name__ == "__main__":
    main()
