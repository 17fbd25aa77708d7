"""world_size-2 gloo tests of the multi-GPU plumbing (runs on CPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from texir_code_amd import dist_util
    ids = torch.arange(0, 3 * n, 3, dtype=torch.int32)            # a compacted "valid texel" list
    mine = dist_util.shard_block_cyclic(ids, rank, world, block=64)
    # each rank "computes" its texels into a zero texture; SUM-all_reduce must reproduce the single-rank result
    full = torch.zeros(3 * n, 3)
    full[mine.long()] = mine.float().unsqueeze(-1) * torch.tensor([1.0, 2.0, 3.0])
    dist_util.assemble_sum(full)
    ref = torch.zeros(3 * n, 3)
    ref[ids.long()] = ids.float().unsqueeze(-1) * torch.tensor([1.0, 2.0, 3.0])
    assert torch.equal(full, ref)
    # shares are disjoint and complete
    cnt = torch.zeros(3 * n)
    cnt[mine.long()] = 1
    dist.all_reduce(cnt)
    assert torch.equal(cnt[ids.long()], torch.ones(n)) and cnt.sum() == n
    # load balance within one block
    sizes = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([float(mine.numel())]))
    assert max(s.item() for s in sizes) - min(s.item() for s in sizes) <= 64
    dist.destroy_process_group()


def test_block_cyclic_shard_and_assemble_world2():
    mp.spawn(_worker, args=(2, _free_port(), 1000), nprocs=2, join=True)


def test_shard_world1_identity():
    from texir_code_amd import dist_util
    ids = torch.arange(10)
    assert dist_util.shard_block_cyclic(ids, 0, 1) is ids
