"""The N > 1 path on CPU: world_size-2 `gloo` processes exercise cremage_amd.dist (sharding, the flat
parameter broadcast, the all-gather of results) and the per-rank seeding convention."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from tests.conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from cremage_amd import dist as D
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)  # different initial weights on every rank
    m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.LayerNorm(16), torch.nn.Linear(16, 4).to(torch.bfloat16))
    m.register_buffer("acp", torch.arange(5, dtype=torch.float32) * (rank + 1))
    sent = D.broadcast_module_(m, src=0)
    flat = torch.cat([p.detach().float().reshape(-1) for p in m.parameters()] + [m.acp])
    # shard 5 images over 2 ranks, produce a per-image tensor from the per-image seed, gather
    items = list(D.shard_range(5, rank, world))
    imgs = torch.stack([torch.full((3, 2, 2), float(D.image_seed(42, i))) for i in items]) if rank == 0 else \
        torch.stack([torch.full((3, 2, 2), float(D.image_seed(42, i))) for i in items] + [torch.zeros(3, 2, 2)])  # pad to equal b
    out = D.all_gather_batch(imgs)
    mx = D.max_over_ranks(float(rank + 1), "cpu")
    D.barrier()
    q.put((rank, sent, flat, items, out, mx))
    torch.distributed.destroy_process_group()


def test_gloo_world2_broadcast_shard_gather():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=120)
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] > 0                    # both ranks moved the same number of bytes
    assert torch.equal(res[0][2], res[1][2])             # parameters + buffers identical after the broadcast
    assert res[0][3] == [0, 1, 2] and res[1][3] == [3, 4]  # contiguous balanced shards
    g = res[0][4]
    assert g.shape == (6, 3, 2, 2) and torch.equal(res[0][4], res[1][4])
    assert [float(g[i, 0, 0, 0]) for i in range(5)] == [42.0, 43.0, 44.0, 45.0, 46.0]  # seed + global image index
    assert res[0][5] == res[1][5] == 2.0


def _worker_mismatch(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from cremage_amd import dist as D
    D.init_from_env(backend="gloo")
    m = torch.nn.Linear(8, 16 if rank == 0 else 12)  # rank 1 built a different model
    try:
        D.broadcast_module_(m, src=0)
        q.put((rank, "no error"))
    except RuntimeError as e:
        q.put((rank, str(e)))
    torch.distributed.destroy_process_group()


def test_gloo_world2_broadcast_of_mismatched_models_fails_on_every_rank():
    """the header check of broadcast_parameters_: a rank whose buckets differ from the source's makes EVERY rank raise (no hang, no bytes
    scattered into the wrong parameters)"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_mismatch, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all("built differently" in res[r] for r in range(world)), res


def _worker_bench_model(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from cremage_amd import dist as D
    from cremage_amd import pipeline as P
    D.init_from_env(backend="gloo")
    torch.manual_seed(1000 + rank)  # the uninitialised tree of rank != 0 starts from different values
    unet_cfg = dict(P.SD15_UNET, model_channels=32, num_heads=4, context_dim=64)
    vae_dd = dict(P.SD15_VAE_DD, ch=32, ch_mult=[1, 2], num_res_blocks=1, resolution=32)
    ldm, sent = P.build_ldm_sharded(rank, "cpu", unet_cfg=unet_cfg, vae_dd=vae_dd)
    names = [n for n, _ in ldm.named_parameters()] + [n for n, _ in ldm.named_buffers()]
    dtypes = sorted({str(p.dtype) for p in ldm.parameters()})
    flat = torch.cat([t.detach().float().reshape(-1) for t in list(ldm.parameters()) + list(ldm.buffers())])
    q.put((rank, sent, names, dtypes, flat.numpy()))  # by value: a shared-memory tensor would need this process to outlive the read
    torch.distributed.destroy_process_group()


def test_gloo_world2_bench_model_construction_branches():
    """bench.py's own model construction (pipeline.build_ldm_sharded) on both of its branches: rank 0 fills the synthetic weights,
    rank 1 builds the uninitialised tree (`.to(bf16)` on the UNet only), and ONE broadcast_module_ leaves rank 1 with rank 0's
    parameters and buffers - i.e. the bucket order the header check guards is the order the real code produces."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bench_model, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=300)
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] > 0
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3] == ["torch.bfloat16", "torch.float32"]
    import numpy as np
    assert np.isfinite(res[0][4]).all() and np.abs(res[0][4]).sum() > 0
    assert np.array_equal(res[0][4], res[1][4])


def test_shard_range_partitions_everything():
    from cremage_amd import dist as D
    for n in [0, 1, 7, 16, 33]:
        for w in [1, 2, 3, 8]:
            got = [i for r in range(w) for i in D.shard_range(n, r, w)]
            assert got == list(range(n))
            sizes = [len(D.shard_range(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_collectives_are_identity():
    from cremage_amd import dist as D
    x = torch.randn(3, 4)
    assert D.all_gather_batch(x) is x
    assert D.broadcast_parameters_([x]) == 0
    assert D.max_over_ranks(3.5, "cpu") == 3.5


def test_bench_rank0_only_steps_do_not_enter_collectives():
    """bench.py runs extra, un-timed steps on rank 0 alone (per-kernel HIP events for the roofline object) after the timed region.  A
    step() that all-gathers its images there would wait for ranks that have already left: every `step(...)` call inside an
    `if rank == 0 ...` block must pass gather=False (it hung the multi-GPU launch otherwise - found by inspection in round 3, since no
    multi-GPU node had run it yet)."""
    import ast
    import os
    from tests.conftest import REPO
    tree = ast.parse(open(os.path.join(REPO, "bench.py")).read())

    def mentions_rank0(test):
        return any(isinstance(n, ast.Compare) and isinstance(n.left, ast.Name) and n.left.id == "rank" and
                   any(isinstance(c, ast.Constant) and c.value == 0 for c in n.comparators) for n in ast.walk(test))

    seen = 0
    for node in ast.walk(tree):
        if isinstance(node, ast.If) and mentions_rank0(node.test):
            for sub in node.body:
                for call in ast.walk(sub):
                    if isinstance(call, ast.Call) and isinstance(call.func, ast.Name) and call.func.id == "step":
                        seen += 1
                        kw = {k.arg: k.value for k in call.keywords}
                        assert "gather" in kw and isinstance(kw["gather"], ast.Constant) and kw["gather"].value is False, \
                            f"bench.py:{call.lineno}: rank-0-only step() must pass gather=False"
    assert seen >= 4  # sd15 (2), extra workloads (1), sdxl / c5 (1)
