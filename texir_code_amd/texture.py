"""nvdiffrast `dr.texture` replacement (models/mat_nvdiffrast.py:131-139): bilinear ('linear') and trilinear
('linear-mipmap-linear') fetch with wrap boundary, differentiable wrt the texture.  Semantics restated from
nvdiffrast's public documentation (parity unpinned: nvdiffrast is not installable here).

Level 0 of the mip stack is the texture tensor itself; levels 1.. are built into a side buffer cached on the texture tensor
(keyed by its version): a frozen texture (the irradiance texture) builds its stack once, and several fetches of the same
parameter inside one step share one build."""
import torch

from . import _lib

FILTER = {"linear": 0, "linear-mipmap-linear": 1}

# TEXIR_TEX_BATCH=0: every texture of a step gets its own launches (mip build, fetch, gather, fold; optim.FusedAdam reads the same switch for its step) instead
# of one launch per kind over all of them (texir_*_batch, include/texir_hip.h).  TEXIR_GRAD_MASK=0: the gradient stacks of the deferring parameters are
# cleared by a fill every step and read densely, instead of being left as they are and read through the view's tap mask.  Both default to on; the parity tests
# run every combination and demand identical bits.
_BATCH = __import__("os").environ.get("TEXIR_TEX_BATCH", "1") != "0"
_GRAD_MASK = __import__("os").environ.get("TEXIR_GRAD_MASK", "1") != "0"


def _mask_enabled():
    # (the per-level reference folds, TEXIR_MIP_PER_LEVEL=1, read a cleared stack: no masks there)
    return _BATCH and _GRAD_MASK and not _per_level()


def _per_level():
    """TEXIR_MIP_PER_LEVEL as the LIBRARY reads it (its snapshot and its parsing rule, csrc/env.cpp): the two sides must never disagree on who clears
    the gradient stack"""
    return _lib.env_switch("TEXIR_MIP_PER_LEVEL") != 0


def _multi_rank(owner):
    """several ranks whose gradient parts are summed as dense tensors (dist_util.reduce_texture_grads) -- unless the texture side of the step is REPLICATED
    on every rank (sharded_step.ShardedMatStep): then nothing is reduced and the single-process forms apply"""
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and not getattr(owner, "_texir_replicated_grads", False)


def _mask_capable(owner, taps, H, W, levels, mode):
    """will the backward of this fetch read and write the parameter's gradient stack through the view's tap mask (no cleared stack needed)?"""
    return bool(_mask_enabled() and mode == 1 and taps is not None and taps[8] is not None and levels >= 4 and H % 4 == 0 and W % 4 == 0
                and owner is not None and getattr(owner, "_texir_defer_fold", False) and getattr(owner, "_texir_defer_levels", 1) >= 2
                and getattr(owner, "_texir_grad_l1", None) is None and not _multi_rank(owner))


def _mips_for(owner, t0, levels, launch=True):
    """levels 1.. of the contiguous [H,W,C] float32 tensor `t0`.  launch=False: returns (stack, build_from) and leaves the build to the caller's batched launch
    (build_from = -1: the stack is valid; 0 / 1: texir_mip_build's from_level).  The side buffer is cached ON the owning tensor object (a
    Parameter lives as long as its model), keyed by the tensor's version: no global state, nothing outlives its owner, and a
    texture updated by the optimiser rebuilds its stack into the SAME buffer (static address => safe under hipGraph replay)."""
    H, W, C = t0.shape
    L = _lib.lib()
    n = int(L.texir_mip_elems(H, W, C, levels))
    key = (t0.data_ptr(), t0._version, H, W, C, levels)
    hit = getattr(owner, "_texir_mips", None)
    # a frozen texture keeps its stack; a trainable one is rebuilt on every use (it changes every optimiser step, and under
    # hipGraph capture the build kernels must be part of the captured sequence)
    if hit is not None and hit[0] == key and not owner.requires_grad:
        return hit[1] if launch else (hit[1], -1)
    # one-shot promise of the caller (sharded_step.ShardedMatStep: the forward of a step fetches in one recorded phase, its backward re-fetches under
    # autograd in another): the stack built earlier in THIS step is still valid, do not build it again
    if getattr(owner, "_texir_reuse_mips", False) and hit is not None and hit[1].numel() == n and hit[1].device == t0.device:
        return hit[1] if launch else (hit[1], -1)
    if hit is not None and hit[1].numel() == n and hit[1].device == t0.device:
        rest = hit[1]
    else:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.TexirError("mip stack must be allocated before hipGraph capture (run one eager step first)")
        rest = torch.empty(n, device=t0.device, dtype=torch.float32)
    # FusedAdam(fuse_mip_fold=True) writes level 1 (the 2x2 average of the texels it has just updated) into this very buffer and raises a
    # ONE-SHOT flag on the parameter: the next build then makes levels 2.. only, and the pass over the whole level-0 texture disappears.
    # The flag is consumed here and never set by a plain forward build, so an edit between two forwards always gets the full build.  What
    # cannot be seen is a write through `.data` (it bumps no version counter) BETWEEN the optimiser step and the next forward -- the
    # reference's own clamp idiom (train_material.py:458): give that clamp to FusedAdam.set_clamp, or call refresh_mips(param) after it.
    # Under hipGraph capture the build is recorded once and replayed after every optimiser step: graph_step.GraphedMatStep marks its
    # parameters (_texir_mip1_graph) and promises a valid level 1 before every replay (it rebuilds the stack eagerly when the flag is stale).
    capturing = torch.cuda.is_current_stream_capturing()
    flagged = getattr(owner, "_texir_mip1_fresh", None) == (t0.data_ptr(), t0._version)
    fresh1 = hit is not None and hit[1] is rest and levels > 2 and (flagged or (capturing and getattr(owner, "_texir_mip1_graph", False)))
    if launch:
        _lib.check(L.texir_mip_build(_lib.ptr(t0), _lib.ptr(rest), H, W, C, levels, 1 if fresh1 else 0, _lib.stream_ptr()))
    try:
        owner._texir_mips = (key, rest)
        if not capturing:
            owner._texir_mip1_fresh = None          # consumed (a recorded build consumes nothing: it runs at replay time)
    except AttributeError:
        pass
    return rest if launch else (rest, 1 if fresh1 else 0)


def refresh_mips(param):
    """full rebuild of a trainable texture's cached mip stack (hipGraph replays call this when the texture was changed behind the
    optimiser's back, see graph_step.GraphedMatStep.step)"""
    hit = getattr(param, "_texir_mips", None)
    if hit is None:
        return
    H, W, C = param.shape
    levels = hit[0][5]
    t0 = param.detach()
    _lib.check(_lib.lib().texir_mip_build(_lib.ptr(t0), _lib.ptr(hit[1]), H, W, C, levels, 0, _lib.stream_ptr()))
    param._texir_mips = ((t0.data_ptr(), t0._version, H, W, C, levels), hit[1])       # the stack now describes THIS version of the texture
    param._texir_mip1_fresh = None


# Tap lists cost ~28 bytes per tap (8 taps per pixel and texture configuration); beyond this many bytes in total, further views fall
# back to the float-atomic scatter (hundreds of 1.5 M-pixel views would otherwise pin hundreds of GB).  TEXIR_TAP_CACHE_GB overrides.
_TAP_BUDGET = int(float(__import__("os").environ.get("TEXIR_TAP_CACHE_GB", "16")) * (1 << 30))
_tap_bytes = 0


def _tap_release(nbytes):
    global _tap_bytes
    _tap_bytes -= nbytes


def _pack_bits(idx, n):
    """int32 words, bit (i & 31) of word (i >> 5) set for every i in idx (n bits, padded to whole words)"""
    touched = torch.zeros(((n + 31) // 32) * 32, device=idx.device, dtype=torch.bool)
    touched[idx] = True
    w = (touched.view(-1, 32).to(torch.int64) << torch.arange(32, device=idx.device, dtype=torch.int64)).sum(1)
    w = w & 0xFFFFFFFF
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).contiguous()


def _tap_lists(cache, H, W, C, levels, mode, uv, uv_da):
    """sorted tap lists of a fixed set of fetch coordinates (one view): built once, kept in the caller's per-view cache dict"""
    global _tap_bytes
    key = ("_taps", H, W, levels, mode, uv.data_ptr(), uv.shape[0])
    hit = cache.get(key)
    if hit is not None:
        return hit
    if key in cache or _tap_bytes + 28 * 8 * uv.shape[0] > _TAP_BUDGET:
        cache[key] = None                  # over budget: this view keeps using the atomic scatter
        return None
    if torch.cuda.is_current_stream_capturing():
        raise _lib.TexirError("tap lists must be built before hipGraph capture (run one eager step first)")
    L = _lib.lib()
    P = uv.shape[0]
    keys = torch.empty(P * 8, device=uv.device, dtype=torch.int64)
    wts = torch.empty(P * 8, device=uv.device, dtype=torch.float32)
    _lib.check(L.texir_tex_taps(H, W, C, levels, _lib.ptr(uv), _lib.ptr(uv_da), mode, P, _lib.ptr(keys), _lib.ptr(wts), _lib.stream_ptr()))
    if mode == 0:
        keys.view(P, 8)[:, 4:] = -1
    ks, order = torch.sort(keys, stable=True)
    first = int((ks < 0).sum().item())
    ks, order = ks[first:], order[first:]
    seg_key, counts = torch.unique_consecutive(ks, return_counts=True)
    starts = (torch.cumsum(counts, 0) - counts).to(torch.int32)
    has_l0 = bool(seg_key.numel() > 0 and int(seg_key[0].item()) < H * W)        # keys are sorted: level-0 texels come first
    l0_mask = None
    if has_l0:
        # one bit per level-0 texel these lists write (bit t & 31 of word t >> 5): lets the fused optimiser read the level-0 gradient only
        # there, so that the (never cleared) gradient buffer needs neither a fill nor a full read per step
        l0_mask = _pack_bits(seg_key[seg_key < H * W], H * W)
    # does any tap of these lists write mip level 1 (keys [H*W, H*W + (H/2)(W/2)): level 1 leads the `rest` stack)?  When none does, the level-1 gradient
    # consists of what the fold from level 2 brings and the fused optimiser need not read the level-1 stack at all
    n0, n1 = H * W, (H // 2) * (W // 2)
    has_l1 = bool(((seg_key >= n0) & (seg_key < n0 + n1)).any().item()) if levels > 1 else False
    # one bit per texel of the mip levels 1.. (`rest` order) these lists write: with it the gradient stack is never cleared -- the folds and the fused
    # optimiser read it through this mask (texir_tex_gather_backward_batch, texir_adam_step_tex_dev_batch)
    rest_mask = None
    if mode == 1 and levels > 1:
        n_rest = sum((H >> l) * (W >> l) for l in range(1, levels))
        rest_mask = _pack_bits(seg_key[seg_key >= n0] - n0, n_rest)
    hit = (seg_key.contiguous(), starts.contiguous(), counts.to(torch.int32).contiguous(), (order // 8).to(torch.int32).contiguous(),
           wts[order].contiguous(), has_l0, l0_mask, has_l1, rest_mask)
    cache[key] = hit
    nbytes = sum(t.numel() * t.element_size() for t in hit[:5]) + (0 if l0_mask is None else l0_mask.numel() * 4) + (0 if rest_mask is None else rest_mask.numel() * 4)
    _tap_bytes += nbytes
    __import__("weakref").finalize(hit[0], _tap_release, nbytes)        # the budget counts LIVE lists: a dropped view cache gives its share back
    return hit


_NO_GRAD = (None,) * 8


def _bwd_prepare(meta, owner, taps, d_out, empty):
    """buffers and decisions of one fetch's backward, before any launch.  Returns a state dict, or {"early": value} when there is nothing to launch
    (value = what autograd gets for the texture)."""
    H, W, C, levels, mode = meta
    if empty:
        # no fetch coordinates (an empty pixel shard): a deferring parameter gets no gradient part at all from this pass
        # (dist_util.reduce_texture_grads supplies zeros where other ranks hold parts), any other texture a dense zero
        if owner is not None and getattr(owner, "_texir_defer_fold", False):
            return {"early": None}
        return {"early": torch.zeros((H, W, C), device=d_out.device, dtype=torch.float32)}
    L = _lib.lib()
    # FusedAdam(fuse_mip_fold=True) asks for the last fold (level 1 -> level 0, a read-modify-write of the whole texture) to be
    # left to its own read of the gradient: the level-1 gradient is parked on the parameter.  Only the first trilinear fetch of a
    # parameter per backward pass defers; a further one folds completely and autograd adds its d_tex as usual.
    defer = (mode == 1 and levels > 1 and owner is not None and getattr(owner, "_texir_defer_fold", False)
             and getattr(owner, "_texir_grad_l1", None) is None)
    # FusedAdam can take the last TWO folds over (level 2 -> 1 -> 0): the folds then stop at level 2 and the read-modify-write of the
    # level-1 stack disappears.  Gather path only (cached tap lists); needs a level above 2 and H, W divisible by 4.
    defer_levels = 0
    if defer:
        defer_levels = 2 if (taps is not None and levels >= 4 and H % 4 == 0 and W % 4 == 0 and getattr(owner, "_texir_defer_levels", 1) >= 2) else 1
    multi = _multi_rank(owner) if owner is not None else False
    # the gradient stack read and written through the view's tap mask: never cleared (same floats as a cleared stack)
    use_mask = bool(defer and defer_levels == 2 and _mask_enabled() and taps[8] is not None and not multi)
    n_rest = int(L.texir_mip_elems(H, W, C, levels))
    g_rest = None
    if levels > 1:
        if defer:
            # the gradient stack of a deferring parameter lives in a persistent buffer -- its slot of the optimiser's arena, or, without
            # one, a buffer owned by the parameter -- reused by every step and shared by all captured hipGraphs (they run one after the
            # other on one stream): nothing is allocated per step or per graph
            arena = getattr(owner, "_texir_arena", None)
            if arena is not None and arena["buf"].device == d_out.device and owner._texir_arena_span[1] - owner._texir_arena_span[0] >= n_rest:
                # FusedAdam's arena: the stacks of all its texture parameters in one buffer, cleared by ONE fill at the step's first
                # fetch (texture() below).  A backward pass whose forward did not clear it clears its own span here.
                lo, hi = owner._texir_arena_span
                g_rest = arena["buf"][lo:lo + n_rest]
                if id(owner) not in arena["clean"] and not use_mask:
                    arena["buf"][lo:hi].zero_()
                arena["clean"].discard(id(owner))          # (this backward writes into it)
            else:
                g_rest = getattr(owner, "_texir_grest", None)
                if g_rest is None or g_rest.numel() != n_rest or g_rest.device != d_out.device:
                    if torch.cuda.is_current_stream_capturing():
                        raise _lib.TexirError("gradient stack must be allocated before hipGraph capture (run one eager step first)")
                    g_rest = torch.empty(n_rest, device=d_out.device, dtype=torch.float32)
                    owner._texir_grest = g_rest
                if not use_mask:
                    g_rest.zero_()
        else:
            g_rest = torch.zeros(n_rest, device=d_out.device, dtype=torch.float32)
    # A deferred fetch over cached tap lists (single process): the level-0 gradient is SPARSE -- only the texels the lists name get a
    # value.  It goes to a buffer owned by the
    # parameter that is never cleared; the view's bit mask (one bit per texel) tells the fused optimiser where to read it.  No
    # 4*H*W*C-byte fill and no dense read per step, nothing allocated per step or per captured graph; autograd gets None.
    sparse_l0 = bool(defer and taps is not None and owner.grad is None and not multi)
    if sparse_l0:
        d_tex = None
        if taps[5]:
            d_tex = getattr(owner, "_texir_g0", None)
            if d_tex is None or d_tex.shape != (H, W, C) or d_tex.device != d_out.device:
                if torch.cuda.is_current_stream_capturing():
                    raise _lib.TexirError("gradient buffer must be allocated before hipGraph capture (run one eager step first)")
                d_tex = torch.zeros((H, W, C), device=d_out.device, dtype=torch.float32)
                owner._texir_g0 = d_tex
    else:
        d_tex = torch.zeros((H, W, C), device=d_out.device, dtype=torch.float32)
    return {"d_tex": d_tex, "g_rest": g_rest, "defer": defer, "defer_levels": defer_levels, "sparse_l0": sparse_l0, "use_mask": use_mask, "d_out": d_out}


def _gather_job(meta, taps, st):
    H, W, C, levels, mode = meta
    seg_key, starts, counts, pix, wts = taps[:5]
    A = _lib.addr
    return _lib.TexGatherJob(A(st["d_tex"]), A(st["g_rest"]), H, W, C, levels, A(seg_key), A(starts), A(counts), seg_key.numel(), A(pix), A(wts), A(st["d_out"]), mode,
                             st["defer_levels"], A(taps[8]) if st["use_mask"] else None, A(st.get("d_out2")))


def _bwd_launch(meta, taps, uv, uv_da, st):
    """the launches of ONE fetch's backward (the batched backward launches its gather jobs together instead)"""
    H, W, C, levels, mode = meta
    L = _lib.lib()
    d_tex, g_rest, d_out = st["d_tex"], st["g_rest"], st["d_out"]
    if taps is not None:
        # fixed fetch coordinates (a cached view): deterministic gather over the pre-sorted tap lists instead of float atomics
        if _BATCH:
            _lib.batch_call("texir_tex_gather_backward_batch", [_gather_job(meta, taps, st)])
        else:
            seg_key, starts, counts, pix, wts = taps[:5]
            _lib.check(L.texir_tex_gather_backward(_lib.ptr(d_tex), _lib.ptr(g_rest), H, W, C, levels, _lib.ptr(seg_key), _lib.ptr(starts),
                                                   _lib.ptr(counts), seg_key.numel(), _lib.ptr(pix), _lib.ptr(wts), _lib.ptr(d_out), mode,
                                                   st["defer_levels"], _lib.stream_ptr()))
    elif st["defer"]:
        _lib.check(L.texir_tex_fetch_backward_deferred(_lib.ptr(d_tex), _lib.ptr(g_rest), H, W, C, levels, _lib.ptr(uv), _lib.ptr(uv_da),
                                                       uv.shape[0], _lib.ptr(d_out), _lib.stream_ptr()))
    else:
        _lib.check(L.texir_tex_fetch_backward(_lib.ptr(d_tex), _lib.ptr(g_rest), H, W, C, levels, _lib.ptr(uv), _lib.ptr(uv_da), mode, uv.shape[0],
                                              _lib.ptr(d_out), _lib.stream_ptr()))


def _bwd_finish(meta, owner, taps, st):
    """bookkeeping on the parameter after the launches; returns what autograd gets for the texture"""
    H, W, C, levels, mode = meta
    defer, defer_levels, g_rest = st["defer"], st["defer_levels"], st["g_rest"]
    if defer:
        n1 = (H // 2) * (W // 2) * C
        g1 = g_rest[:n1]
        # (the mask travels WITH the parked stack: whoever holds this tensor -- the optimiser, a captured graph's record of it -- knows how to read it)
        g1._texir_mask = taps[8] if st["use_mask"] else None
        owner._texir_grad_l1 = g1
        owner._texir_grad_l2 = g_rest[n1:n1 + (H // 4) * (W // 4) * C] if defer_levels == 2 else None
        # level 1 untouched by this view's taps and the fold 2 -> 1 left to the optimiser: the parked level-1 stack is all zeros (the arena fill) and
        # FusedAdam.step passes NULL for it (the stack stays parked as the marker of a deferred gradient)
        owner._texir_l1_zero = bool(defer_levels == 2 and taps is not None and not taps[7])
    if owner is not None:
        # does the level-0 gradient of this parameter hold anything at all after this backward pass?  A deferred fetch none of whose
        # pixels samples level 0 leaves d_tex all zero (the multi-GPU reduction can then skip it); any other fetch of the parameter
        # writes it.  Sticky over the fetches of one backward pass; FusedAdam.zero_grad resets it.
        wrote_l0 = (taps[5] if taps is not None else True) if defer else True
        owner._texir_l0_touched = bool(getattr(owner, "_texir_l0_touched", False)) or wrote_l0
    if st["sparse_l0"]:
        owner._texir_l0_mask = taps[6] if taps[5] else None      # (None + no .grad: level-0 gradient identically zero)
        owner._texir_l0_sparse = True
        return None
    return st["d_tex"]


class _TexFetch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tex, rest, uv, uv_da, mode, levels, owner=None, taps=None):
        L = _lib.lib()
        H, W, C = tex.shape
        P = uv.shape[0]
        t0 = tex.detach()
        out = torch.empty((P, C), device=tex.device, dtype=torch.float32)
        if P > 0:                                    # (zero fetch coordinates: an empty result, and a zero gradient in the backward)
            _lib.check(L.texir_tex_fetch_forward(_lib.ptr(t0), _lib.ptr(rest), H, W, C, levels, _lib.ptr(uv), _lib.ptr(uv_da), mode, P, _lib.ptr(out),
                                                 _lib.stream_ptr()))
        ctx.save_for_backward(uv, uv_da)
        ctx.meta = (H, W, C, levels, mode)
        ctx.owner = owner
        ctx.taps = taps
        return out

    @staticmethod
    def backward(ctx, d_out):
        uv, uv_da = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return _NO_GRAD
        st = _bwd_prepare(ctx.meta, ctx.owner, ctx.taps, d_out.contiguous(), uv.shape[0] == 0)
        if "early" in st:
            return (st["early"],) + _NO_GRAD[1:]
        _bwd_launch(ctx.meta, ctx.taps, uv, uv_da, st)
        return (_bwd_finish(ctx.meta, ctx.owner, ctx.taps, st),) + _NO_GRAD[1:]


class _TexFetchBatch(torch.autograd.Function):
    """the fetches of several textures at the same coordinates (the albedo and the roughness texture of a view, models/mat_nvdiffrast.py:131-139) as ONE
    autograd node: one launch per kind of kernel over all of them, forward (mip pyramids, their tails, fetch) and backward (gather, folds).  Every
    texture's values and gradients are those of its own _TexFetch, bit for bit.
    A texture may hand its fetched values out TWICE (fan = 2: two tensors over one buffer, for two consumers -- the roughness goes to the specular term
    and to the loss): the two gradients then arrive separately and the gather adds them while it reads them (texir_tex_gather_job.d_out2), instead of
    autograd adding them in a launch of its own; the sum is the same float."""

    @staticmethod
    def forward(ctx, uv, uv_da, info, *texs):
        # info: per texture (rest, build_from, mode, levels, owner, taps, fan)
        P = uv.shape[0]
        outs, jobs, metas = [], [], []
        A = _lib.addr
        for k, (tex, (rest, build_from, mode, levels, owner, taps, fan)) in enumerate(zip(texs, info)):
            H, W, C = tex.shape
            out = torch.empty((P, C), device=tex.device, dtype=torch.float32)
            mine = [out] + [out.detach() for _ in range(fan - 1)]          # (the further hand-outs alias the first one's memory)
            outs += mine
            if not ctx.needs_input_grad[3 + k]:
                ctx.mark_non_differentiable(*mine)       # (a frozen texture of the batch: its output is a constant, as its own node's would be)
            metas.append((H, W, C, levels, mode))
            if P > 0 or build_from >= 0:
                jobs.append(_lib.TexFetchJob(A(tex.detach()), A(rest), H, W, C, levels, build_from, mode, A(uv), A(uv_da), P, A(out)))
        if jobs:
            _lib.batch_call("texir_tex_fetch_forward_batch", jobs)
        ctx.save_for_backward(uv, uv_da)
        ctx.metas, ctx.info = metas, info
        ctx.set_materialize_grads(False)             # an output nobody differentiates (the detached albedo of stage 1) gets no backward work
        return tuple(outs)

    @staticmethod
    def backward(ctx, *d_outs):
        uv, uv_da = ctx.saved_tensors
        grads = [None] * len(ctx.info)
        todo = []
        pos = 0
        for i, inf in enumerate(ctx.info):
            fan = inf[6]
            ds = [d.contiguous() for d in d_outs[pos:pos + fan] if d is not None]
            pos += fan
            if not ds or not ctx.needs_input_grad[3 + i]:
                continue
            owner, taps = inf[4], inf[5]
            if len(ds) == 2 and (taps is None or not _BATCH or _per_level()):
                ds = [ds[0] + ds[1]]                # (no gather to add them in: the float-atomic scatter and the per-level reference folds take one gradient)
            st = _bwd_prepare(ctx.metas[i], owner, taps, ds[0], uv.shape[0] == 0)
            if "early" in st:
                grads[i] = st["early"]
            else:
                st["d_out2"] = ds[1] if len(ds) == 2 else None
                todo.append((i, st))
        batched = [(i, st) for i, st in todo if ctx.info[i][5] is not None]
        if batched:
            _lib.batch_call("texir_tex_gather_backward_batch", [_gather_job(ctx.metas[i], ctx.info[i][5], st) for i, st in batched])
        for i, st in todo:
            if ctx.info[i][5] is None:
                _bwd_launch(ctx.metas[i], None, uv, uv_da, st)              # (no cached tap lists: the float-atomic scatter, one texture at a time)
            grads[i] = _bwd_finish(ctx.metas[i], ctx.info[i][4], ctx.info[i][5], st)
        return (None, None, None) + tuple(grads)


def _prep_fetch(tex, uv, uv_da, filter_mode, max_mip_level, cache, launch_mips):
    """common front half of texture() / texture_batch(): argument normalisation, the mip stack (built here, or left to the batched launch), the arena
    fill, the view's tap lists"""
    if tex.dim() == 4:
        tex = tex[0]
    mode = FILTER[filter_mode]
    if mode == 1 and uv_da is None:
        raise ValueError("linear-mipmap-linear needs uv_da")
    owner = tex
    if tex.dtype != torch.float32 or not tex.is_contiguous():
        tex = tex.to(torch.float32).contiguous()
    H, W, C = tex.shape
    levels = int(_lib.lib().texir_mip_levels(H, W, int(max_mip_level))) if mode == 1 else 1
    rest, build_from = None, -1
    if levels > 1:
        if launch_mips:
            rest = _mips_for(owner, tex.detach(), levels)
        else:
            rest, build_from = _mips_for(owner, tex.detach(), levels, launch=False)
    taps = None
    if cache is not None and tex.requires_grad and torch.is_grad_enabled() and uv.shape[0] > 0:
        taps = _tap_lists(cache, H, W, C, levels, mode, uv, uv_da)
    arena = getattr(owner, "_texir_arena", None)
    if (arena is not None and mode == 1 and tex.requires_grad and torch.is_grad_enabled() and id(owner) not in arena["clean"]
            and not _mask_capable(owner, taps, H, W, levels, mode)):
        # first deferring fetch of a step: clear the gradient stacks of every trainable parameter of the arena with one fill (in the forward:
        # stream-ordered before every backward of the step, whichever parameter's comes first)
        # ... unless a gradient parked by an earlier backward pass is still waiting for its optimiser step (gradient accumulation: fwd, bwd, fwd,
        # bwd, step): the fill would wipe it.  The later backward passes then do not defer and fold into private stacks (_bwd_prepare).
        # ... and not at all for a fetch whose backward goes through the view's tap mask (_mask_capable): that stack is never cleared.
        ps = [q for q in arena["params"] if q.requires_grad]
        if not any(getattr(q, "_texir_grad_l1", None) is not None for q in arena["params"]):
            arena["buf"][min(q._texir_arena_span[0] for q in ps):max(q._texir_arena_span[1] for q in ps)].zero_()
            arena["clean"] = set(id(q) for q in ps)
    return tex, owner if isinstance(owner, torch.nn.Parameter) else None, rest, build_from, mode, levels, taps


def texture(tex, uv, uv_da=None, filter_mode="linear", max_mip_level=13, cache=None):
    """tex [H,W,C] (or [1,H,W,C]); uv [...,2]; uv_da [...,4] (du/dX,du/dY,dv/dX,dv/dY) -> [...,C].
    cache: a dict that lives as long as (uv, uv_da) stay the same tensors (the per-view G-buffer cache): the backward then runs as a
    deterministic gather over tap lists sorted once, instead of a float-atomic scatter."""
    lead = uv.shape[:-1]
    uvf = uv.reshape(-1, 2).to(torch.float32).contiguous()
    daf = None if uv_da is None else uv_da.reshape(-1, 4).to(torch.float32).contiguous()
    tex, owner, rest, _, mode, levels, taps = _prep_fetch(tex, uvf, daf, filter_mode, max_mip_level, cache, True)
    out = _TexFetch.apply(tex, rest, uvf, daf, mode, levels, owner, taps)
    return out.reshape(*lead, tex.shape[2])


def texture_batch(texs, uv, uv_da=None, filter_mode="linear", max_mip_level=13, cache=None, fanout=None):
    """[texture(t, uv, uv_da, ...) for t in texs] with one launch per kind of kernel over all textures (at most _lib.MAX_BATCH), forward and backward.
    TEXIR_TEX_BATCH=0 (or a single texture) runs the plain per-texture fetches.
    fanout (optional, one entry per texture: 1 or 2): a texture with 2 is returned as a PAIR of tensors holding the same values, one per consumer; their
    gradients are added inside the gather instead of by autograd (see _TexFetchBatch).  Without batching the pair is the same tensor twice."""
    texs = list(texs)
    fan = [1] * len(texs) if fanout is None else [int(f) for f in fanout]
    if any(f not in (1, 2) for f in fan) or len(fan) != len(texs):
        raise ValueError("texture_batch: fanout takes one entry (1 or 2) per texture")
    if not _BATCH or len(texs) < 2 or len(texs) > _lib.MAX_BATCH:
        single = [texture(t, uv, uv_da, filter_mode, max_mip_level, cache) for t in texs]
        return [o if f == 1 else (o, o) for o, f in zip(single, fan)]
    lead = uv.shape[:-1]
    uvf = uv.reshape(-1, 2).to(torch.float32).contiguous()
    daf = None if uv_da is None else uv_da.reshape(-1, 4).to(torch.float32).contiguous()
    prepped = [_prep_fetch(t, uvf, daf, filter_mode, max_mip_level, cache, False) for t in texs]
    info = tuple((rest, build_from, mode, levels, owner, taps, f) for (_, owner, rest, build_from, mode, levels, taps), f in zip(prepped, fan))
    outs = list(_TexFetchBatch.apply(uvf, daf, info, *[q[0] for q in prepped]))
    res = []
    for q, f in zip(prepped, fan):
        mine = [outs.pop(0).reshape(*lead, q[0].shape[2]) for _ in range(f)]
        res.append(mine[0] if f == 1 else tuple(mine))
    return res
