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
