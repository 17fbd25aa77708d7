"""Multi-GPU sharding of the hot path: one process per GPU, texels / pixels partitioned across ranks,
BVH + radiance texture replicated; IrT: one RCCL all_gather of the ranks' compacted texel values assembles the texture; Mat: all_reduce(SUM) of
the texture gradients."""
import torch


# payload accounting of the data-path collectives this module (and sharded_step) issues: bytes of the tensors handed to RCCL, per call site.
# bench.py resets it around a timed loop and reports bytes per step; nothing on the device path reads it.
COMM = {"bytes": 0, "calls": 0}


def account(t, times=1):
    COMM["bytes"] += int(t.numel()) * t.element_size() * times
    COMM["calls"] += 1


def comm_reset():
    COMM["bytes"], COMM["calls"] = 0, 0


def shard_block_cyclic(ids, rank, world, block=4096):
    """rank's share of the compacted valid-texel list: blocks of `block` consecutive entries, block b -> rank b % world.
    Interleaving balances load (occupancy and ray cost vary over the atlas).  The shares are disjoint and cover ids."""
    if world == 1:
        return ids
    n = ids.numel()
    b = torch.arange(n, device=ids.device) // block
    return ids[(b % world) == rank]


def world_info():
    import os
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def assemble_sum(t):
    """all_reduce(SUM) across ranks when a process group is up (disjoint supports => a gather); no-op otherwise."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t)
    return t


def shard_plan(ids_all, world, block=4096, device=None):
    """the per-rank row lists of assemble_shards, computed once (a bench / trainer that assembles every step keeps it)"""
    if device is not None:
        ids_all = ids_all.to(device)
    return [shard_block_cyclic(ids_all, r, world, block).long() for r in range(world)]


def _flat_all_gather_ok(t):
    """all_gather_into_tensor exists for the nccl (= RCCL) backend on device tensors and for gloo from torch 2.x on; anything else takes the list form"""
    import torch.distributed as dist
    if not hasattr(dist, "all_gather_into_tensor"):
        return False
    be = dist.get_backend()
    if be == "nccl":
        return t.is_cuda
    if be == "gloo":
        return tuple(int(x) for x in torch.__version__.split("+")[0].split(".")[:2]) >= (2, 1)
    return False


def assemble_shards(tex, ids_all, block=4096, plan=None):
    """Assemble the irradiance texture `tex` [Nt, C] whose rows ids_all[shard r] were computed by rank r (shard_block_cyclic(ids_all, r, world, block)):
    every rank contributes only ITS texels' values, compacted ([n_r, C]: 12 bytes per valid texel instead of a dense all_reduce over the whole
    texture, seams and other ranks' zeros included), one all_gather moves them, and each rank scatters the others' rows into its copy.  Supports
    are disjoint, so this is bit-identical to the SUM all-reduce it replaces.  `ids_all` must be the same list on every rank (`plan` =
    shard_plan(ids_all, world, block, tex.device) skips recomputing the row lists).  No-op for one rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return tex
    rank, world = dist.get_rank(), dist.get_world_size()
    shards = plan if plan is not None else shard_plan(ids_all, world, block, tex.device)
    mx = max(int(s_.numel()) for s_ in shards)
    if mx == 0:
        return tex
    C = tex.shape[1]
    mine = torch.zeros((mx, C), device=tex.device, dtype=tex.dtype)
    mine[: shards[rank].numel()] = tex[shards[rank]]
    out = torch.empty((world * mx, C), device=tex.device, dtype=tex.dtype)
    # which form of the collective to issue is decided from the backend BEFORE anything is issued (and so identically on every rank): a fall-back
    # taken after a failed attempt would leave the ranks issuing different collectives -- real collective errors propagate
    account(mine, world)
    if _flat_all_gather_ok(tex):
        dist.all_gather_into_tensor(out, mine)
    else:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        out = torch.cat(parts, 0)
    for r in range(world):
        if r != rank and shards[r].numel():
            tex[shards[r]] = out[r * mx: r * mx + shards[r].numel()]
    return tex


def reduce_texture_grads(params):
    """sum the texture gradients over the ranks before the (replicated) optimiser step.  `params` must be the same list on every rank.
    With FusedAdam(fuse_mip_fold=True) a parameter's gradient is the triple (p.grad = level-0 scatter, p._texir_grad_l1 = level-1 stack,
    p._texir_grad_l2 = level-2 stack when the last two folds are left to the step): the coarse stacks (1/4 + 1/16 of the texture) are
    reduced whenever some rank parked them, the full-resolution part only if some rank's pixels sampled level 0 at all -- at 4k textures
    seen through 128^2 cube faces nothing does, and the reduction shrinks from 335 MB to 84 MB.

    What a rank holds depends on ITS pixels (an empty pixel shard parks nothing, a view beyond the tap-list budget parks no level-2
    stack), so the ranks first agree -- one small MAX all-reduce -- on which of the three parts exist anywhere; a rank that lacks a part
    another rank has contributes zeros (taken from its slot of the optimiser's gradient arena).  Every rank therefore issues the same
    collectives with the same sizes and ends up with the same parts attached."""
    import torch.distributed as dist
    params = list(params)
    if not params or not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    dev = params[0].device

    def local(p):
        g1, g2 = getattr(p, "_texir_grad_l1", None), getattr(p, "_texir_grad_l2", None)
        has_l0 = p.grad is not None and (g1 is None or bool(getattr(p, "_texir_l0_touched", True)))
        return [1.0 if has_l0 else 0.0, 0.0 if g1 is None else 1.0, 0.0 if g2 is None else 1.0]

    have = torch.tensor([local(p) for p in params], device=dev)
    dist.all_reduce(have, op=dist.ReduceOp.MAX)
    for p, (n0, n1, n2) in zip(params, have.tolist()):
        if n1 > 0 or n2 > 0:
            H, W, C = p.shape
            e1, e2 = (H // 2) * (W // 2) * C, (H // 4) * (W // 4) * C
            span = getattr(p, "_texir_arena_span", None) if getattr(p, "_texir_arena", None) is not None else None

            def zeros(offset, n):
                if span is not None and span[1] - span[0] >= offset + n and getattr(p, "_texir_grad_l1", None) is None:
                    z = p._texir_arena["buf"][span[0] + offset:span[0] + offset + n]       # (this rank's backward left its slot unused)
                    p._texir_arena["clean"].discard(id(p))
                    return z.zero_()
                return torch.zeros(n, device=p.device, dtype=torch.float32)

            if n2 > 0 and getattr(p, "_texir_grad_l2", None) is None:
                p._texir_grad_l2 = zeros(e1, e2)
            if getattr(p, "_texir_grad_l1", None) is None:
                p._texir_grad_l1 = zeros(0, e1)
            account(p._texir_grad_l1)
            dist.all_reduce(p._texir_grad_l1)
            p._texir_l1_zero = False            # (another rank's view may have written level 1: the summed stack must be read)
            if n2 > 0:                         # (the fold level 2 -> 1 is left to the optimiser step as well: both parts are linear in the ranks' sums)
                account(p._texir_grad_l2)
                dist.all_reduce(p._texir_grad_l2)
        if n0 > 0:
            if p.grad is None:                 # this rank's pixels touched no level-0 texel (the tensor was never made), another rank's did
                p.grad = torch.zeros_like(p)
            account(p.grad)
            dist.all_reduce(p.grad)


def morton_order(ids, width):
    """reorder a texel-id list (row-major ids of a [H,width] texture) along the Z-order curve, so that texels processed
    concurrently by neighbouring wavefronts are neighbours on the surface (their rays then share BVH leaves and texture
    lines in L2).  The estimator is per-texel, so the order is free."""
    ids64 = ids.to(torch.int64)
    r, c = ids64 // width, ids64 % width
    code = torch.zeros_like(ids64)
    for b in range(16):
        code |= ((c >> b) & 1) << (2 * b)
        code |= ((r >> b) & 1) << (2 * b + 1)
    return ids[torch.argsort(code)]


def pixel_range(n_pixels, rank, world):
    """contiguous slice [p0, p1) of a view's flattened pixel list owned by `rank` (sizes differ by at most one)"""
    base, rem = divmod(n_pixels, world)
    p0 = rank * base + min(rank, rem)
    return p0, p0 + base + (1 if rank < rem else 0)
