"""Multi-GPU layer: independent fits are sharded over ranks, nothing is exchanged while fitting.

The reference has no distributed code at all; the axis that parallelises is the PriorCache axis - one
image <-> one private parameter set (awesome/util/prior_cache.py:49-59).  One process per GPU (`torch.distributed`,
backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests); the only collectives are a barrier, a MAX of the
elapsed time and an all_gather of per-image metrics (a few bytes per image)."""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def dist_env() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1-process defaults)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    rank, world, local = dist_env()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # INRFIT_DIST_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share devices, the
        # few bytes of collectives travel over CPU tensors); the real run is "nccl" (= RCCL over xGMI), one rank per GPU
        backend = backend or os.environ.get("INRFIT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _coll_device(device):
    """Where collective payloads must live: the caller's device under nccl, the CPU under gloo."""
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo":
        return torch.device("cpu")
    return device


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous block partition (first n % world ranks get one extra).  Contiguous, not round-robin, so that frames of a
    sequence that warm-start from each other (`reuse_state`, path_connected_net.py:867-870) stay on one rank."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def shard_sequences(seq_lengths: Sequence[int], rank: int, world: int) -> List[int]:
    """Sequence ids for this rank; whole sequences only (warm-start chains never cross ranks), greedy by length."""
    order = sorted(range(len(seq_lengths)), key=lambda i: -seq_lengths[i])
    load = [0] * world
    owner = {}
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += seq_lengths[i]
    return sorted(i for i, r in owner.items() if r == rank)


def barrier() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def any_rank_failed(failed: bool, device=None) -> bool:
    """True on EVERY rank if any rank reports a failure (one MAX all-reduce).  Called by a rank BEFORE it joins the data
    collectives, so that a rank whose work raised does not leave the others waiting in max_over_ranks / gather_per_image / barrier
    until the launcher times out: everybody learns of the failure and exits non-zero together."""
    if not (dist.is_available() and dist.is_initialized()):
        return bool(failed)
    t = torch.tensor([1 if failed else 0], dtype=torch.int32, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


def shutdown() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_per_image(local: torch.Tensor, n_total: int, rank: int, world: int) -> torch.Tensor:
    """all_gather of a per-image metric vector whose rows follow shard_range(); returns the global vector [n_total, ...]."""
    if not (dist.is_available() and dist.is_initialized()) or world == 1:
        return local
    sizes = [len(shard_range(n_total, r, world)) for r in range(world)]
    pad = max(sizes)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=_coll_device(local.device))
    buf[: local.shape[0]] = local
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], 0).to(local.device)
