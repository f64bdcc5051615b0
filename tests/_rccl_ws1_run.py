"""Child process of tests/test_rccl_gpu.py: a ONE-rank RCCL communicator on the GPU box (no 8-GPU node is available to this build).

Order matters: `init_process_group("nccl", device_id=...)` is the FIRST GPU call of the process, exactly as in `bench.py --gpus N`
(dist.init_from_env), with HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment (the pool's driver only supports dmabuf IPC).  Then the
two collectives of the data path run on device tensors: the flat-bucket parameter broadcast (bf16 + fp32 buckets, header check) and
the all-gather of a batch, plus the barrier / max-over-ranks of bench.py's timing protocol.  Any failure exits non-zero; no retry."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29731"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from cremage_amd import dist as D  # noqa: E402


def main():
    rank, world, local = D.init_from_env(backend="nccl", force=True)
    assert (rank, world, local) == (0, 1, 0) and torch.distributed.is_initialized()
    assert torch.distributed.get_backend() == "nccl"
    dev = torch.device("cuda", 0)
    torch.manual_seed(7)
    m = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.LayerNorm(256), torch.nn.Linear(256, 32).to(torch.bfloat16)).to(dev)
    m.register_buffer("acp", torch.arange(1000, dtype=torch.float32, device=dev))
    before = [p.detach().clone() for p in list(m.parameters()) + list(m.buffers())]
    sent = D.broadcast_module_(m, src=0, force=True)
    expect = sum(t.numel() * t.element_size() for t in before)
    assert sent == expect, (sent, expect)
    for a, b in zip(before, list(m.parameters()) + list(m.buffers())):
        assert torch.equal(a, b)  # rank 0 is the source: the round trip through the flat bucket must return the same bytes
    x = torch.randn((4, 3, 64, 64), device=dev)
    out = D.all_gather_batch(x, force=True)
    assert out.shape == x.shape and torch.equal(out, x) and out.data_ptr() != x.data_ptr()
    D.barrier()
    assert D.max_over_ranks(3.25, dev) == 3.25
    torch.cuda.synchronize()
    torch.distributed.destroy_process_group()
    print(f"RCCL_WS1_OK bytes_broadcast={sent}")


if __name__ == "__main__":
    main()
