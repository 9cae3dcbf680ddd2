"""N > 1 path on CPU: world_size 2, gloo - the sharding / gather / max-time logic bench.py and scripts/run.py use."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_images, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from awesome_amd import parallel
    r, w, _ = parallel.init(backend="gloo")
    mine = list(parallel.shard_range(n_images, r, w))
    # stand-in for the per-image fit result: metric = f(global image index), computed only for this rank's shard
    local = torch.tensor([[float(i), float(i) ** 2] for i in mine], dtype=torch.float32).reshape(len(mine), 2)
    parallel.barrier()
    full = parallel.gather_per_image(local, n_images, r, w)
    t = parallel.max_over_ranks(1.0 + r)
    if r == 0:
        q.put((full.tolist(), t))
    parallel.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [7, 8])
def test_shard_and_gather_world2(n_images):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_images, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, t = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert full == [[float(i), float(i) ** 2] for i in range(n_images)]   # every image exactly once, in global order
    assert t == 2.0                                                      # max over ranks


def _failing_worker(rank, world, port, q):
    """The pattern of scripts/run.py: rank-local work in a try block, ONE flag all-reduce before the data collectives, and every rank
    leaves non-zero when any rank failed (a rank that raised used to strand the others in a collective)."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from awesome_amd import parallel
    r, w, _ = parallel.init(backend="gloo")
    error = None
    try:
        if r == 1:
            raise ValueError("Loss is nan or inf! (images [3])")
    except Exception as err:   # noqa: BLE001
        error = err
    failed = parallel.any_rank_failed(error is not None)
    q.put((r, failed))
    parallel.shutdown()
    raise SystemExit(1 if failed else 0)


def test_a_failing_rank_is_agreed_on_before_the_collectives():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 1          # both ranks exit non-zero, neither hangs
    assert got == [(0, True), (1, True)]


def test_bench_launcher_starts_the_ranks_and_relays_one_line():
    """`python bench.py --gpus 2` with no torch.distributed environment must start 2 ranks itself (torch.distributed.run as a child,
    before any GPU call), run every collective of the bench (barrier, MAX of the time, all_gather of the per-image metric, SUM of
    the failure count) and relay rank 0's JSON line with the child's exit code.  On this CPU-only box the fit is replaced by
    bench.py's labelled rehearsal stand-in (BENCH_REHEARSAL=1 + gloo): the line carries no value."""
    import json
    import subprocess
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_REHEARSAL="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--images-per-gpu", "3"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["rehearsal"] is True and out["value"] is None and out["n_gpus"] == 2 and out["steps"] == 2
    assert out["gathered_image_seeds"] == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]      # rank r owns seeds r*B .. r*B+B-1, gathered in order
    assert out["nonfinite_fits"] == 0


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", BENCH_BACKEND="gloo", BENCH_REHEARSAL="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE 2" in (r.stderr + r.stdout)
