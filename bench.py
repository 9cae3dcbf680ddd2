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
    ap.add_argument("--cpu-sample-steps", type=int, default=120, help="timed optimizer steps of the CPU baseline at its best thread setting")
    ap.add_argument("--cpu-sample-steps-8", type=int, default=60, help="... of its 8-thread setting (SURVEY 6's yardstick); 0 = skip")
    ap.add_argument("--cpu-sweep-steps", type=int, default=25, help="timed optimizer steps per thread setting of the sweep (16, 32, 64, "
                    "128, all host threads) that picks the CPU baseline's setting; 0 = the all-thread setting without a sweep")
    ap.add_argument("--strong-images", type=int, default=512,
                    help="STRONG-scaling entry: this many images in total (BASELINE configs[2] = 512), image i on rank i // ceil(total / N), "
                         "fitted in device batches of --throughput-images; reported under 'strong_scaling_configs2' (0 = skip); never `value`")
    ap.add_argument("--no-variant-cpu", action="store_true", help="skip the CPU leg + parity field of every variant")
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
    # HBM bytes per launch of this kernel from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes of
    # this same command: tools/profile_round.sh).  A --pmc pass cannot run inside this process, so the figure is read from the
    # committed summary of the newest round that has one and only quoted for the configuration it was measured on;
    # `traffic_source` names the file.  `algorithmic_bytes`: SURVEY 8(d)'s 4-12 B per point (targets [+ grid + logits]) PLUS what
    # this design adds on purpose - one gradient slab per workgroup (wgs x PS x 4 B written here, read by the update kernel) and
    # the parameter image every workgroup copies into LDS - so `traffic_over_algorithmic` separates "wasted re-reads" (none: ~1.0)
    # from the slab round trip the design pays for (`traffic_over_survey_bytes`).
    traffic, traffic_source = None, None
    for tag in ("r04_b", "r03_f", "r03_e", "r02_e"):
        tf = os.path.join(ROOT, "profiles", f"{tag}_pmc_step_kernel.json")
        if B == 1 and S == 256 and os.path.exists(tf):
            with open(tf) as f:
                traffic = json.load(f).get("traffic_bytes_per_launch")
            traffic_source = os.path.relpath(tf, ROOT)
            break
    lib0 = A._lib.load()
    wgs = int(lib0.inrfit_slabs_per_image(N, B))
    slab_cols = (19222 + 31) // 32 * 32 if spec.n_hidden == 130 and spec.in_features == 2 else None   # Cfg<130,2>::SL_COLS (icnn_step.h), rows on 128-byte lines
    survey_bytes = 12 * N * B                                   # <= 8 B grid + 4 B target per point
    design_bytes = None
    if slab_cols is not None:
        design_bytes = 4 * N * B + B * wgs * slab_cols * 4 + 8 * 77 * 1024   # targets + slabs + the parameter image once per XCD
    # what a launch of nothing but independent fp32 MFMAs on all 1024 SIMDs reaches on THIS box (~2.1 GHz under that load instead
    # of the 2.4 GHz the nominal peak assumes): the practical ceiling, reported next to `peak`, never instead of it
    mfma_stream = A.icnn.mfma_stream_tflops(dev)
    roofline = {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP32_MATRIX_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes": design_bytes, "survey_bytes_per_launch": survey_bytes,
                "traffic_over_algorithmic": round(traffic / design_bytes, 3) if traffic and design_bytes else None,
                "traffic_over_survey_bytes": round(traffic / survey_bytes, 1) if traffic else None,
                "kernel": "icnn_step_kernel<130,2,train>", "kernel_us": round(kernel_ms * 1e3, 2),
                "kernel_samples": seq.samples, "kernel_us_back_to_back": round(kernel_b2b_ms * 1e3, 2),
                "update_kernel_us": round(update_us, 2), "event_bracket_excess_us": round(excess, 2),
                "flop_per_launch": flop_per_launch, "mfma_only_stream_tflops": round(mfma_stream, 1),
                "frac_of_mfma_only_stream": round(achieved / mfma_stream, 4)}

    # ---- which devices the ranks sat on: (rank, local rank, PCI bus id / uuid, the backend's world size), gathered so that the
    # driver's record can show that the N-rank line came from N distinct GPUs over RCCL -------------------------------------
    props = torch.cuda.get_device_properties(dev)
    dev_id = str(getattr(props, "uuid", "")) or ""
    pci = f"{getattr(props, 'pci_domain_id', 0):04x}:{getattr(props, 'pci_bus_id', 0):02x}:{getattr(props, 'pci_device_id', 0):02x}"
    me = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(), "pci_bus_id": pci, "uuid": dev_id,
          "name": props.name, "backend": (dist.get_backend() if dist is not None else None),
          "world_size": (dist.get_world_size() if dist is not None else 1)}
    if dist is not None:
        placements = [None] * world
        dist.all_gather_object(placements, me)
    else:
        placements = [me]

    # ---- throughput mode (extra, not `value`): configs[2]'s per-GPU share, one complete E-step fit of a batch ---------
    # image i of rank r = configs[2]'s image r*TB + i: blob seed s, initial weights torch.manual_seed(s) + the reference's default
    # init - so that the first images coincide with the reference classes' own fits in tests/golden/fits_blob256_multi_{a,b}.npz
    thr = None
    if args.throughput_images > 0:
        TB = args.throughput_images
        tseeds = [rank * TB + i for i in range(TB)]
        tun = torch.stack([convex_blob_unaries(S, s).reshape(-1) for s in tseeds]).to(dev)
        tinit = []
        for s in tseeds:
            torch.manual_seed(s)
            tinit.append(ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=1).flat_parameters())
        tinit = torch.stack(tinit).to(dev)
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
        if rank == 0 and S == 256 and E == 2000:
            thr.update(miou_delta_vs_reference(tseeds, tiou.cpu(), (torch.sigmoid(tres.logits) > 0.5).cpu()))

    # ---- strong scaling (extra, not `value`): BASELINE configs[2] as ONE job - 512 images in total, a contiguous block of
    # ceil(512 / N) per rank (awesome_amd.parallel.shard_range: the split scripts/run.py uses), each rank in device batches of
    # `--throughput-images`.  Total work is fixed, so fits/s over N = 1, 2, 4, 8 is a curve that is NOT linear by construction:
    # the last rank's remainder batch, the per-batch launch shapes (fewer images per launch = more gradient slabs per image) and
    # the barrier show up in it.
    strong = None
    if args.strong_images > 0:
        from awesome_amd import parallel as PAR
        total, DB = args.strong_images, max(1, args.throughput_images or 64)
        mine = list(PAR.shard_range(total, rank, world))
        t_prep = time.perf_counter()
        sun = torch.stack([convex_blob_unaries(S, s).reshape(-1) for s in mine]).to(dev) if mine else torch.zeros(0, N, device=dev)
        sinit = []
        for s in mine:
            torch.manual_seed(s)
            sinit.append(ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=1).flat_parameters())
        sinit = torch.stack(sinit).to(dev) if mine else torch.zeros(0, spec.n_params, device=dev)
        prep_s = time.perf_counter() - t_prep
        barrier()
        t2 = time.perf_counter()
        s_iou, s_bad = [], 0
        for off in range(0, len(mine), DB):
            r_ = A.fit(spec, sinit[off:off + DB].clone(), grid, sun[off:off + DB].contiguous(), E, lr=2e-3, loss="se", optimizer="adam",
                       clamp=True, record_loss=False, want_logits=True)
            s_iou.append(A.miou((torch.sigmoid(r_.logits) > 0.5).float(), (sun[off:off + DB] > 0.5).float(), invert=True))
            s_bad += int((r_.status != 0).sum().item())
        sync()
        my_s = time.perf_counter() - t2
        barrier()
        sdt = max_over_ranks(time.perf_counter() - t2)
        rank_s = gather_cat(torch.tensor([my_s], dtype=torch.float32, device=dev)).tolist()
        n_all = gather_cat(torch.tensor([float(len(mine))], dtype=torch.float32, device=dev)).tolist()
        iou_sum = sum(float(v.sum()) for v in s_iou)
        tot_iou = gather_cat(torch.tensor([iou_sum], dtype=torch.float32, device=dev))
        strong = {"scaling": "strong", "images_total": total, "images_per_rank": [int(v) for v in n_all], "device_batch": DB,
                  "fits_per_s": round(total / sdt, 4), "seconds": round(sdt, 3), "per_rank_seconds": [round(v, 3) for v in rank_s],
                  "miou_vs_unaries": round(float(tot_iou.sum()) / total, 5), "nonfinite_fits": sum_over_ranks(s_bad),
                  "input_generation_seconds_rank0": round(prep_s, 2),
                  "note": "BASELINE configs[2] (512 independent 256x256 fits) as one job sharded over the ranks; time = barrier to "
                          "barrier, max over ranks; inputs resident on the device before the first barrier"}
        del sun, sinit

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
            "placements": placements,
        }
        if thr is not None:
            out["throughput_mode"] = thr
        if strong is not None:
            out["strong_scaling_configs2"] = strong
        if world == 1 and not args.no_variants:
            out["variants"] = path_variants(dev, S, cpu_legs=not args.no_variant_cpu)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, init[0].cpu(), unaries[0].cpu(), spec)
            out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        barrier()
        dist.destroy_process_group()


def miou_delta_vs_reference(seeds, iou, masks):
    """Per-image parity of the batched fit with the REFERENCE CLASSES' own 2000-step fits of the same problems
    (tests/golden/fits_blob256_multi_{a,b}.npz: tools/gen_golden_scale.py, two runs of the reference with different OpenMP
    thread counts - their disagreement is the reference's own run-to-run floor): the dataset-mean difference north_star bounds
    by 1e-3, and the histogram of per-image |dIoU| next to the reference-vs-reference one."""
    import numpy as np
    fa, fb = (os.path.join(ROOT, "tests", "golden", f"fits_blob256_multi_{t}.npz") for t in "ab")
    if not (os.path.exists(fa) and os.path.exists(fb)):
        return {}
    za, zb = np.load(fa), np.load(fb)
    common = [k for k, s in enumerate(seeds) if f"s{s}.final_miou" in za.files and f"s{s}.final_miou" in zb.files]
    if not common:
        return {}
    ra = np.array([float(za[f"s{seeds[k]}.final_miou"]) for k in common])
    rb = np.array([float(zb[f"s{seeds[k]}.final_miou"]) for k in common])
    hip = np.array([float(iou[k]) for k in common])
    edges = [0.0, 1e-4, 3e-4, 1e-3, 3e-3, 1e-2, 1.0]
    d_hip = np.minimum(np.abs(hip - ra), np.abs(hip - rb))        # distance to the nearer of the two reference runs
    d_ref = np.abs(ra - rb)
    px = []
    for k in common:
        m = np.unpackbits(za[f"s{seeds[k]}.final_mask_bits"])[: masks.shape[1]].astype(bool)
        px.append(int((masks[k].numpy() != m).sum()))
    return {"miou_delta_hist": {"images": len(common), "bin_edges": edges,
                                "hip_vs_reference": np.histogram(d_hip, edges)[0].tolist(),
                                "reference_vs_reference": np.histogram(d_ref, edges)[0].tolist(),
                                "max_hip_vs_reference": round(float(d_hip.max()), 6), "max_reference_vs_reference": round(float(d_ref.max()), 6),
                                "mean_miou_hip": round(float(hip.mean()), 6),
                                "mean_miou_reference": [round(float(ra.mean()), 6), round(float(rb.mean()), 6)],
                                "mean_abs_diff_of_means": round(float(abs(hip.mean() - 0.5 * (ra.mean() + rb.mean()))), 6),
                                "mask_pixels_differing_from_reference_run_a": px,
                                "source": "tests/golden/fits_blob256_multi_{a,b}.npz (reference classes, tools/gen_golden_scale.py)"}}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def _host_threads():
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    return max(1, cores)


def path_variants(dev, S, steps=300, cpu_legs=True):
    """Extra, never part of `value`: the other priors on the same path (one image / sequence on this GPU): ICNN L=2,
    ConvexDiffeomorphismNet, PathConnectedNet (RealNVP) on (x, y) and on (x, y, t), and the fused joint-training step of
    configs[4].  Per variant: microseconds per optimizer step of the fused fit (wall clock over `steps` steps), a `roofline` object
    on the ALGORITHMIC flops of one step (3 x the forward flops of the ICNN and of the deformation, SURVEY.md 8d's convention;
    bound = the fp32 matrix/vector peak, which are the same number on this chip), a `check` (the loss of the timed fit is finite
    and falls), and - from ONE short run of the CPU oracle's loop on the same problem from the same parameters - `cpu` (its time per
    step on this host) and `parity` (the head of the HIP loss curve against the oracle's, relative)."""
    import torch
    import awesome_amd as A
    from awesome_amd.dataset import SyntheticSequenceDataset, convex_blob_unaries
    from awesome_amd.model import ConvexDiffeomorphismNet, ConvexNextNet, real_nvp_path_connected_net

    def icnn_fwd_flop(h, c, layers):          # 4h + L(2h^2 + 4h) + 2h + 4 for c = 2 (2hc + L(2h^2 + 2hc) + 2h + 2c in general)
        return 2 * h * c + layers * (2 * h * h + 2 * h * c) + 2 * h + 2 * c

    threads = min(_host_threads(), 16)

    def entry(us, n_points, fwd_flop_icnn, fwd_flop_flow, hist, oracle=None, oracle_steps=0):
        flop = 3.0 * (fwd_flop_icnn + fwd_flop_flow) * n_points
        tf = flop / (us * 1e-6) / 1e12
        h = hist.float().cpu()
        e = {"us_per_step": us,
             "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                          "frac": round(tf / PEAK_FP32_MATRIX_TFLOPS, 4), "flop_per_step": int(flop),
                          "flop_per_point_icnn_fwd": fwd_flop_icnn, "flop_per_point_flow_fwd": fwd_flop_flow,
                          "basis": "wall clock per optimizer step (all kernels of the step), not one kernel"},
             "check": {"loss_first": round(float(h[0]), 6), "loss_last": round(float(h[-1]), 6),
                       "ok": bool(torch.isfinite(h).all() and h[-1] < h[0])}}
        if oracle is not None and cpu_legs:
            from oracle import inr_oracle as O   # noqa: F401  (checker + CPU baseline leg only)
            torch.set_num_threads(threads)
            oracle(1)                                               # warm-up (allocator, thread pool)
            t0 = time.perf_counter()
            losses = oracle(oracle_steps)
            dt = time.perf_counter() - t0
            ref = torch.tensor(losses, dtype=torch.float32)
            rel = ((h[:oracle_steps] - ref).abs() / ref.abs().clamp_min(1e-12))
            e["cpu"] = {"ms_per_step": round(dt / oracle_steps * 1e3, 2), "threads": threads, "kind": "port",
                        "sample": f"{oracle_steps} optimizer steps of the oracle's loop on the same problem ({dt:.1f} s)",
                        "speedup": round(dt / oracle_steps * 1e6 / us, 1)}
            e["parity"] = {"loss_head_steps": oracle_steps, "loss_head_max_rel_diff_vs_oracle": float(f"{float(rel.max()):.3e}"),
                           "first_loss_hip": float(h[0]), "first_loss_oracle": float(ref[0])}
        return e

    def timed(fn):
        fn(10)
        best, res = None, None
        for _ in range(2):   # best of two runs of `steps` steps (one run right after the 64-image fit was once seen at half speed)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = fn(steps)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return round(best / steps * 1e6, 1), res.loss_hist[0]

    from oracle import inr_oracle as O   # the checker / CPU leg (never on the product path)
    torch.manual_seed(0)
    N = S * S
    un = convex_blob_unaries(S, 0).reshape(1, -1).to(dev)
    un_img = un.reshape(1, 1, S, S).cpu()
    grid = A.Grid.linspace(S, S, dev)
    grid_t = O.positional_grid(S, S)[None]
    out = {"unit": "us per optimizer step", "steps": steps}
    m2 = ConvexNextNet(n_hidden=130, n_hidden_layers=2, in_features=2)
    p2 = m2.flat_parameters()[None].to(dev)
    sd2 = {k: v.detach().clone() for k, v in m2.state_dict().items()}
    us, h = timed(lambda n: A.fit(m2.spec, p2.clone(), grid, un, n, lr=2e-3, record_loss=True, want_logits=False))
    out[f"ConvexNextNet_L2_{S}x{S}"] = entry(us, N, icnn_fwd_flop(130, 2, 2), 0, h,
                                             lambda n: O.fit_icnn(sd2, grid_t, un_img, n, lr=2e-3)[1], 20)
    # shapes without a fused kernel: the layer-by-layer path (csrc/wide.h + csrc/gemm.h; parity: tests/test_gpu_icnn.py)
    for hw_, lw_ in ((256, 1), (350, 3)):
        mw = ConvexNextNet(n_hidden=hw_, n_hidden_layers=lw_, in_features=2)
        pw = mw.flat_parameters()[None].to(dev)
        sdw = {k: v.detach().clone() for k, v in mw.state_dict().items()}
        us, h = timed(lambda n: A.fit(mw.spec, pw.clone(), grid, un, n, lr=2e-3, record_loss=True, want_logits=False))
        out[f"ConvexNextNet_h{hw_}_L{lw_}_layer_by_layer_{S}x{S}"] = entry(us, N, icnn_fwd_flop(hw_, 2, lw_), 0, h,
                                                                         lambda n, sdw=sdw: O.fit_icnn(sdw, grid_t, un_img, n, lr=2e-3)[1], 6 if lw_ == 1 else 3)
    cdn = ConvexDiffeomorphismNet(n_hidden=130, n_hidden_layers=2, nf_layers=6, nf_hidden=130,
                                  diffeo_args=dict(backbone="normal_block")).to(dev)   # the reference configs' form
    sdc = {k: v.detach().cpu().clone() for k, v in cdn.state_dict().items()}
    us, h = timed(lambda n: cdn.fit_images(grid, un, num_epochs=n))
    out[f"ConvexDiffeomorphismNet_K6_w130_L2_{S}x{S}"] = entry(
        us, N, icnn_fwd_flop(130, 2, 2), 6 * 8 * 130, h,   # 8 w flop per coupling
        lambda n: O.fit_convex_diffeo(sdc, grid_t, un_img, n, 6, lr=3e-3, loss_kind="bce", weight_decay_on_weight_g=5e-5,
                                      plateau=dict(patience=200, factor=0.5))[1], 8)

    def pcn_oracle(model, rows, un_rows, C, F):
        _, rspec = model._specs()
        sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()
               if v.dtype == torch.float32 and not k.startswith("flow_net.norm") and not k.endswith("data_dep_init_done")}
        masks = O.rnvp_masks(C, F)
        return lambda n: O.fit_pcn(sd0, rows, un_rows, n, masks, torch.tensor(rspec.vmin), torch.tensor(rspec.vmax), lr=1e-3,
                                   optimizer="adamax", flow_weight_decay=1e-5, plateau=dict(patience=200, factor=0.5))[1]

    pc2 = real_nvp_path_connected_net(channels=2, hidden_units=32, flow_n_flows=12, flow_output_fn="tanh").to(dev)
    pc2._actnorm_init_if_needed(pc2._first_image_coords(grid))
    rows2 = O.pixelize(grid_t)
    us, h = timed(lambda n: pc2.fit_images(grid, un, num_epochs=n))
    out[f"PathConnectedNet_RealNVP_C2_F12_L2_{S}x{S}"] = entry(us, N, icnn_fwd_flop(130, 2, 2), 12 * 32 * 2 * 2 * 2, h,   # F hid 2 nets C 2
                                                               pcn_oracle(pc2, rows2, un.cpu().reshape(-1, 1), 2, 12), 6)
    ds = SyntheticSequenceDataset(1, 128, 16)
    g3, u3 = A.Grid.explicit(ds.coords().to(dev)), ds.batch([0]).to(dev)
    pc3 = real_nvp_path_connected_net(channels=3, hidden_units=32, flow_n_flows=18, flow_output_fn="tanh").to(dev)
    pc3._actnorm_init_if_needed(pc3._first_image_coords(g3))
    us, h = timed(lambda n: pc3.fit_images(g3, u3, num_epochs=n))
    out["PathConnectedNet_RealNVP_C3_F18_L2_128x128x16"] = entry(us, 128 * 128 * 16, icnn_fwd_flop(130, 3, 2), 18 * 32 * 2 * 3 * 2, h,
                                                                 pcn_oracle(pc3, ds.coords().t().contiguous(), u3.cpu().reshape(-1, 1), 3, 18), 3)
    # The same two 256x256 priors with a BATCH of images per launch (configs[2]'s shape of work for the path-connected priors: all
    # images step together; the launches then hold several waves per SIMD and the per-launch costs are shared).
    nb = 16
    unb = torch.stack([convex_blob_unaries(S, s).reshape(-1) for s in range(nb)]).to(dev)

    def batched(fn, fwd_flow):
        fn(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = fn(100)
        torch.cuda.synchronize()
        us_img = (time.perf_counter() - t0) / 100 / nb * 1e6
        flop = 3.0 * (icnn_fwd_flop(130, 2, 2) + fwd_flow) * N
        hh = res.loss_hist.float().cpu()
        return {"images_per_launch": nb, "us_per_step_per_image": round(us_img, 1),
                "roofline_frac_wall_clock": round(flop / (us_img * 1e-6) / 1e12 / PEAK_FP32_MATRIX_TFLOPS, 4),
                "check": {"ok": bool(torch.isfinite(hh).all() and (hh[:, -1] < hh[:, 0]).all())}}
    out["batched_16_images"] = {
        f"ConvexDiffeomorphismNet_K6_w130_L2_{S}x{S}": batched(lambda n: cdn.fit_images(grid, unb, num_epochs=n), 6 * 8 * 130),
        f"PathConnectedNet_RealNVP_C2_F12_L2_{S}x{S}": batched(lambda n: pc2.fit_images(grid, unb, num_epochs=n), 12 * 32 * 2 * 2 * 2),
    }
    out["joint_step_configs4"] = joint_step_variant(dev, S, icnn_fwd_flop)
    return out


def joint_step_variant(dev, S, icnn_fwd_flop, steps=300):
    """BASELINE configs[4]'s training step (TorchAgent._perform_step with FBMSJointLoss on noisy 256x256 pseudo-labels), fused: per
    step the torch backbone (a 3x3 convolution stand-in for the UNet, which is out of scope) + ONE C-ABI call for everything behind
    its output.  Wall clock per step over a round-robin of 8 images whose priors sit in a PriorBank, for the convexity prior and for
    the path-connected one (RealNVP), fused and through the autograd bridges (round 2's path); `c_abi_call_us` = the device time of
    the fused call alone (HIP events around 200 calls)."""
    import torch
    import awesome_amd as A
    from awesome_amd import joint as J
    from awesome_amd.agent import JointTrainer
    from awesome_amd.dataset import SyntheticPriorDataset
    from awesome_amd.measures import FBMSJointLoss
    from awesome_amd.model import ConvSegStandIn, ConvexNextNet, WrapperModule, real_nvp_path_connected_net
    from awesome_amd.prior_bank import PriorBank, _ordered_parameters
    n_img, N = 8, S * S
    torch.manual_seed(3)
    ds = SyntheticPriorDataset(n_images=n_img, size=S, kind="noisy_blob")
    items = [ds[i] for i in range(n_img)]
    feat = torch.zeros(1, 1, 1, 1, device=dev)
    batch = [((it[0][0][None].to(dev), feat, it[0][2][None].to(dev)), it[1][None].to(dev)) for it in items]
    res = {"unit": "us per joint training step", "images_round_robin": n_img, "steps": steps,
           "note": "segmentation backbone = 3x3 conv stand-in in torch (UNet out of scope); loss FBMSJointLoss(sssdms BCE, clip), Adam"}

    def run(factory, fused, launches, fwd_flop_icnn, fwd_flop_flow):
        seg = ConvSegStandIn().to(dev)
        wrapper = WrapperModule(seg, factory().to(dev), use_segmentation_output_inversion=True).to(dev)
        bank = PriorBank(lambda: factory().to(dev), n_images=n_img, device=dev)
        for k in range(n_img):
            bank.row(k)
        for b in wrapper.prior_module.buffers():
            pass
        for name, b in wrapper.prior_module.named_buffers():
            if name.endswith("data_dep_init_done"):
                b.fill_(1.0)
        opt = torch.optim.Adam(list(seg.parameters()) + list(_ordered_parameters(wrapper.prior_module)), lr=1e-3)
        tr = JointTrainer(wrapper, bank, FBMSJointLoss(alpha=1.0, beta=1.0), opt, fused=fused)
        first = None
        for k in range(2 * n_img):   # warm
            loss, _ = tr.perform_step(k % n_img, *batch[k % n_img])
            first = first if first is not None else float(loss)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            loss, _ = tr.perform_step(k % n_img, *batch[k % n_img])
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / steps * 1e6
        flop = 3.0 * (fwd_flop_icnn + fwd_flop_flow) * N
        e = {"us_per_step": round(us, 1), "kernel_launches_per_step_behind_the_backbone": launches,
             "loss_first": round(first, 6), "loss_last": round(float(loss), 6), "ok": bool(torch.isfinite(loss))}
        if fused:
            tf = flop / (us * 1e-6) / 1e12
            e["roofline"] = {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(tf / PEAK_FP32_MATRIX_TFLOPS, 4), "flop_per_step": int(flop),
                             "basis": "wall clock per joint step incl. the torch backbone and Python, on 3 x the prior's forward flops"}
        return e

    # what the torch side of a step costs by itself (backbone forward + backward + its Adam step, same Python loop, no prior)
    seg = ConvSegStandIn().to(dev)
    opt = torch.optim.Adam(seg.parameters(), lr=1e-3)
    dummy = torch.zeros(1, S, S, device=dev)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps if rep else 16):
            opt.zero_grad()
            out = 1 - torch.sigmoid(seg(batch[k % n_img][0][0])[0])
            out.backward(dummy)
            opt.step()
        torch.cuda.synchronize()
    res["torch_backbone_only_us"] = round((time.perf_counter() - t0) / steps * 1e6, 1)
    icnn = lambda: ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=1)   # noqa: E731
    pcn = lambda: real_nvp_path_connected_net(channels=2, hidden_units=32, flow_n_flows=12, flow_output_fn="tanh")   # noqa: E731
    res["convex_prior_fused"] = run(icnn, True, 8, icnn_fwd_flop(130, 2, 1), 0)
    res["convex_prior_autograd_bridges"] = run(icnn, False, None, icnn_fwd_flop(130, 2, 1), 0)
    res["path_connected_prior_fused"] = run(pcn, True, 13, icnn_fwd_flop(130, 2, 2), 12 * 32 * 2 * 2 * 2)
    res["path_connected_prior_autograd_bridges"] = run(pcn, False, None, icnn_fwd_flop(130, 2, 2), 12 * 32 * 2 * 2 * 2)
    # the C-ABI call alone
    m = ConvexNextNet(n_hidden=130, in_features=2, n_hidden_layers=1)
    row, opt_state = m.flat_parameters().to(dev), torch.zeros(2 * m.spec.n_params + 8, device=dev)
    grid = A.Grid.linspace(S, S, dev)
    seg = torch.sigmoid(items[0][0][0].reshape(-1).to(dev))
    tgt = items[0][1].reshape(-1).to(dev)
    desc = J.joint_desc()
    for t in range(1, 11):
        J.joint_step(m.spec, row, opt_state, grid, seg, tgt, desc, step=t, lr=1e-3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(11, 211):
        J.joint_step(m.spec, row, opt_state, grid, seg, tgt, desc, step=t, lr=1e-3)
    e1.record()
    torch.cuda.synchronize()
    res["convex_prior_fused"]["c_abi_call_us"] = round(e0.elapsed_time(e1) / 200 * 1e3, 1)
    return res


def cpu_baseline(args, flat0, unaries0, spec):
    """The CPU oracle (pure torch restatement of the reference loop, parity-pinned to the reference's golden vectors) timed on
    this host on a bounded sample of the SAME 256x256 fit, extrapolated linearly to E steps: all host threads (`value`) and the
    8-thread setting SURVEY 6 measured the reference classes at (77 ms per step in the build container)."""
    import torch
    import awesome_amd as A
    from oracle import inr_oracle as O   # the thing being timed as the CPU baseline (kind = "port")

    S, E = args.size, args.epochs
    cores = _host_threads()
    p = A.unpack_params(spec, flat0)
    grid = O.positional_grid(S, S)[None]
    un = unaries0.reshape(1, 1, S, S)

    def leg(threads, n):
        torch.set_num_threads(threads)
        O.fit_icnn(p, grid, un, 3, lr=2e-3)   # warm-up
        t0 = time.perf_counter()
        O.fit_icnn(p, grid, un, n, lr=2e-3)
        dt = time.perf_counter() - t0
        return dt, dt / n

    # Which thread setting is the CPU path's best on this host is MEASURED, not assumed (the 16-thread cap was an
    # assertion): a short sweep over 16, 32, 64, 128 and ALL host threads; `value` is the best setting's rate on a longer sample,
    # the all-thread leg ("all physical cores") is always reported beside it.
    sweep = {}
    if args.cpu_sweep_steps > 0:
        for t in sorted({t for t in (16, 32, 64, 128, cores) if t <= cores}):
            probe = leg(t, 3)[1]                    # 3 steps first: an oversubscribed setting (seconds per step) is not run longer
            if sweep and probe * 1e3 > 2.0 * min(sweep.values()):
                sweep[t] = round(probe * 1e3, 2)
            else:
                sweep[t] = round(leg(t, args.cpu_sweep_steps)[1] * 1e3, 2)
        threads = min(sweep, key=sweep.get)
    else:
        threads = cores
    n = args.cpu_sample_steps
    dt, s_per_step = leg(threads, n)
    out = {"value": round(1.0 / (s_per_step * E), 6), "unit": "fits/s", "cores": threads, "kind": "port",
           "cpu_model": _cpu_model(), "host_threads_available": cores,
           "sample": f"{n} of {E} optimizer steps of the same {S}x{S} fit ({dt:.1f} s, {s_per_step * 1e3:.1f} ms/step) at the best of the "
                     f"swept thread settings, torch {torch.__version__} CPU, extrapolated linearly",
           "ms_per_optimizer_step": round(s_per_step * 1e3, 2),
           "thread_sweep_ms_per_step": {str(k): v for k, v in sweep.items()},
           "all_threads": ({"cores": cores, "ms_per_optimizer_step": sweep[cores], "value": round(1.0 / (sweep[cores] * 1e-3 * E), 6),
                            "sample": "3 to %d optimizer steps (3 when more than 2x slower than the best setting)" % args.cpu_sweep_steps}
                           if cores in sweep else None)}
    if args.cpu_sample_steps_8 > 0 and cores >= 8:
        dt8, sp8 = leg(8, args.cpu_sample_steps_8)
        out["threads_8"] = {"value": round(1.0 / (sp8 * E), 6), "unit": "fits/s", "cores": 8, "ms_per_optimizer_step": round(sp8 * 1e3, 2),
                            "sample": f"{args.cpu_sample_steps_8} optimizer steps ({dt8:.1f} s)",
                            "survey_reference_classes_8_threads_ms_per_step": 77.0}
    return out


if __name__ == "__main__":
    main()
