"""Batch-level sharding across the GPUs of one node: one process per GPU, torch.distributed over
RCCL/xGMI (backend "nccl" on ROCm).  The reference has no distributed code at all;
this is new functionality named by BASELINE.json's north_star:

  * images are independent units -> rank r owns global image indices [r*b, (r+1)*b); per-image seeds
    are `seed + global_image_index`, mirroring the reference's metadata convention
    (modules/sd/image_generator.py:1135);
  * ONE collective at load time: the parameters are broadcast from rank 0 as a single flat buffer per
    dtype (1.72 GB bf16 UNet + 0.2 GB fp32 VAE) - few, large messages, which is what xGMI's
    point-to-point links want;
  * ONE collective per batch: all-gather of the decoded images (each rank decodes its own latents;
    decode is ~7 % of the FLOPs and parallelises perfectly);
  * NO per-step communication and no tensor-parallel split of the UNet.

Everything here also runs on the `gloo` backend with CPU tensors (tests/test_dist_cpu.py).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None, force: bool = False) -> tuple:
    """(rank, world_size, local_rank) from the torchrun environment; initialises the default group when
    WORLD_SIZE > 1 (rendezvous on MASTER_ADDR/MASTER_PORT).  `force`: initialise the group at world size 1 as well (a one-rank RCCL
    communicator: tests/test_rccl_gpu.py proves the communicator, the dmabuf-IPC environment and the bucket code on a one-GPU box)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


_DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2, torch.float64: 3, torch.int64: 4, torch.int32: 5, torch.uint8: 6,
               torch.bool: 7}


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced shard of `n_items` independent units (first `n_items % world` ranks get one more)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def image_seed(seed: int, global_index: int) -> int:
    return seed + global_index


@torch.no_grad()
def broadcast_parameters_(tensors: Iterable[torch.Tensor], src: int = 0, group=None, force: bool = False) -> int:
    """Broadcast all tensors from `src` as one flat buffer per (dtype, device); returns bytes sent.  `force`: run the collectives on a
    one-rank group too (otherwise a no-op there)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return 0
    buckets = {}
    for t in tensors:
        buckets.setdefault((t.dtype, t.device), []).append(t)
    # header first: every rank must be about to receive exactly what `src` sends (same buckets, in the same order, same sizes) - a
    # model built differently on one rank otherwise makes the flat broadcast hang or scatter bytes into the wrong parameters
    dev = next(iter(buckets))[1] if buckets else torch.device("cpu")
    mine = [len(buckets)] + [v for (dt, _d), ts in buckets.items() for v in (_DTYPE_CODE.get(dt, -1), len(ts), sum(t.numel() for t in ts))]
    n = torch.tensor([len(mine)], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src, group=group)
    head = torch.tensor(mine if len(mine) == int(n.item()) else [0] * int(n.item()), dtype=torch.int64, device=dev)
    theirs = head.clone()
    dist.broadcast(theirs, src=src, group=group)
    ok = torch.tensor([int(len(mine) == int(n.item()) and bool((theirs == head).all().item()))], dtype=torch.int64, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)  # every rank learns the verdict, so every rank raises (nobody waits alone)
    if int(ok.item()) != 1:
        raise RuntimeError(f"broadcast_parameters_: rank {dist.get_rank(group)} holds {mine} (bucket count, then dtype code / tensors / "
                           f"elements per bucket) which does not match rank {src} on every rank: the models were built differently")
    total = 0
    for (_dt, _dev), ts in buckets.items():
        flat = torch.cat([t.detach().reshape(-1) for t in ts])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in ts:
            n = t.numel()
            t.detach().copy_(flat[off:off + n].view_as(t))
            off += n
        total += flat.numel() * flat.element_size()
    return total


@torch.no_grad()
def broadcast_module_(module: torch.nn.Module, src: int = 0, group=None, force: bool = False) -> int:
    return broadcast_parameters_(list(module.parameters()) + list(module.buffers()), src=src, group=group, force=force)


@torch.no_grad()
def all_gather_batch(x: torch.Tensor, group=None, force: bool = False) -> torch.Tensor:
    """[b, ...] per rank -> [world*b, ...] on every rank, rank-major (equal b on all ranks).  `force`: as in broadcast_parameters_."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return x
    world = dist.get_world_size(group)
    x = x.contiguous()
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x, group=group)
    return out


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
