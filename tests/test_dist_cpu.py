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
    # the compacted form the product uses: every rank sends only its own texels' values (all_gather), ragged last block included
    full2 = torch.zeros(3 * n, 3)
    full2[mine.long()] = mine.float().unsqueeze(-1) * torch.tensor([1.0, 2.0, 3.0])
    full2[1] = 7.0                                                  # a seam texel nobody lists stays untouched
    dist_util.assemble_shards(full2, ids, block=64)
    ref2 = ref.clone()
    ref2[1] = 7.0
    assert torch.equal(full2, ref2)
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


def test_block_cyclic_shard_and_assemble_world8_ragged():
    """8 ranks (the node the path is sharded for): a list whose last block is ragged (8 x 3 full rounds of 64-texel blocks + 37 entries) ..."""
    mp.spawn(_worker, args=(8, _free_port(), 64 * 8 * 3 + 37), nprocs=8, join=True)


def test_block_cyclic_shard_and_assemble_world8_fewer_blocks_than_ranks():
    """... and a list shorter than one round (4 blocks, the last one ragged): ranks 4..7 own nothing and must still take part in the collective"""
    mp.spawn(_worker, args=(8, _free_port(), 64 * 3 + 5), nprocs=8, join=True)


def _plan_worker(rank, world, port):
    """shard_plan + assemble_shards with a plan, on a Morton-ordered list (what the model and the bench use), incl. an all-empty list"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from texir_code_amd import dist_util
    W = 96
    g = torch.Generator().manual_seed(5)
    valid = torch.rand(W * W, generator=g) > 0.35
    ids = dist_util.morton_order(torch.nonzero(valid)[:, 0].to(torch.int32), W)
    plan = dist_util.shard_plan(ids, world, block=256)
    assert sum(int(p.numel()) for p in plan) == ids.numel() and torch.equal(torch.sort(torch.cat(plan))[0], torch.sort(ids.long())[0])
    mine = dist_util.shard_block_cyclic(ids, rank, world, block=256)
    assert torch.equal(mine.long(), plan[rank])
    val = lambda i: torch.stack([i.float(), i.float() * 0.5, -i.float()], -1)
    tex = torch.zeros(W * W, 3)
    tex[mine.long()] = val(mine)
    dist_util.comm_reset()
    dist_util.assemble_shards(tex, ids, block=256, plan=plan)
    ref = torch.zeros(W * W, 3)
    ref[ids.long()] = val(ids)
    assert torch.equal(tex, ref)
    mx = max(int(p.numel()) for p in plan)
    assert dist_util.COMM["bytes"] == world * mx * 12 and dist_util.COMM["calls"] == 1
    empty = torch.zeros(0, dtype=torch.int32)
    t0 = torch.full((16, 3), 2.0)
    assert torch.equal(dist_util.assemble_shards(t0.clone(), empty, block=256), t0)              # nothing listed anywhere: no collective, texture untouched
    dist.destroy_process_group()


def test_shard_plan_and_assemble_morton_world8():
    mp.spawn(_plan_worker, args=(8, _free_port()), nprocs=8, join=True)


def test_shard_world1_identity():
    from texir_code_amd import dist_util
    ids = torch.arange(10)
    assert dist_util.shard_block_cyclic(ids, 0, 1) is ids


def _grad_worker(rank, world, port):
    """reduce_texture_grads when the ranks hold DIFFERENT parts of the gradient: rank 0 parked level-1 and level-2 stacks and
    sampled no level-0 texel, rank 1's pixel shard was empty (nothing at all); a second parameter has a dense gradient on rank 1 only"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from texir_code_amd import dist_util
    H = W = 8
    p, q = torch.nn.Parameter(torch.zeros(H, W, 3)), torch.nn.Parameter(torch.zeros(H, W, 1))
    for t in (p, q):
        t._texir_grad_l1 = t._texir_grad_l2 = None
        t._texir_l0_touched = False
        t._texir_arena = None
    n1, n2 = (H // 2) * (W // 2) * 3, (H // 4) * (W // 4) * 3
    if rank == 0:
        p._texir_grad_l1, p._texir_grad_l2 = torch.full((n1,), 2.0), torch.full((n2,), 3.0)
    else:
        q.grad = torch.full((H, W, 1), 5.0)
        q._texir_l0_touched = True
        # rank 1's slot of a gradient arena: the zeros it contributes for p come from there
        buf = torch.full((n1 + n2 + 7,), 9.0)
        p._texir_arena, p._texir_arena_span = {"buf": buf, "params": [p], "clean": {id(p)}}, (0, n1 + n2 + 7)
    dist_util.reduce_texture_grads([p, q])
    assert p.grad is None                                                  # no rank sampled level 0 of p: the 4 * H * W * C bytes are never reduced
    assert torch.equal(p._texir_grad_l1, torch.full((n1,), 2.0)) and torch.equal(p._texir_grad_l2, torch.full((n2,), 3.0))
    assert torch.equal(q.grad, torch.full((H, W, 1), 5.0)) and q._texir_grad_l1 is None
    if rank == 1:
        assert id(p) not in p._texir_arena["clean"]                        # (the slot holds the reduced stacks now)
    dist.destroy_process_group()


def test_reduce_texture_grads_with_uneven_parts_world2():
    mp.spawn(_grad_worker, args=(2, _free_port()), nprocs=2, join=True)
