"""nvdiffrast `dr.texture` replacement (models/mat_nvdiffrast.py:131-139): bilinear ('linear') and trilinear
('linear-mipmap-linear') fetch with wrap boundary, differentiable wrt the texture.  Semantics restated from
nvdiffrast's public documentation (parity unpinned: nvdiffrast is not installable here).

Level 0 of the mip stack is the texture tensor itself; levels 1.. are built into a side buffer cached on the texture tensor
(keyed by its version): a frozen texture (the irradiance texture) builds its stack once, and several fetches of the same
parameter inside one step share one build."""
import torch

from . import _lib

FILTER = {"linear": 0, "linear-mipmap-linear": 1}


def _mips_for(owner, t0, levels):
    """levels 1.. of the contiguous [H,W,C] float32 tensor `t0`.  The side buffer is cached ON the owning tensor object (a
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
        return hit[1]
    # one-shot promise of the caller (sharded_step.ShardedMatStep: the forward of a step fetches in one recorded phase, its backward re-fetches under
    # autograd in another): the stack built earlier in THIS step is still valid, do not build it again
    if getattr(owner, "_texir_reuse_mips", False) and hit is not None and hit[1].numel() == n and hit[1].device == t0.device:
        return hit[1]
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
    _lib.check(L.texir_mip_build(_lib.ptr(t0), _lib.ptr(rest), H, W, C, levels, 1 if fresh1 else 0, _lib.stream_ptr()))
    try:
        owner._texir_mips = (key, rest)
        if not capturing:
            owner._texir_mip1_fresh = None          # consumed (a recorded build consumes nothing: it runs at replay time)
    except AttributeError:
        pass
    return rest


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
        touched = torch.zeros(((H * W + 31) // 32) * 32, device=uv.device, dtype=torch.bool)
        touched[seg_key[seg_key < H * W]] = True
        w = (touched.view(-1, 32).to(torch.int64) << torch.arange(32, device=uv.device, dtype=torch.int64)).sum(1)
        l0_mask = (w & 0xFFFFFFFF).to(torch.int64)
        l0_mask = torch.where(l0_mask >= 2 ** 31, l0_mask - 2 ** 32, l0_mask).to(torch.int32).contiguous()
    # does any tap of these lists write mip level 1 (keys [H*W, H*W + (H/2)(W/2)): level 1 leads the `rest` stack)?  When none does, the level-1 gradient
    # consists of what the fold from level 2 brings and the fused optimiser need not read the level-1 stack at all
    n0, n1 = H * W, (H // 2) * (W // 2)
    has_l1 = bool(((seg_key >= n0) & (seg_key < n0 + n1)).any().item()) if levels > 1 else False
    hit = (seg_key.contiguous(), starts.contiguous(), counts.to(torch.int32).contiguous(), (order // 8).to(torch.int32).contiguous(),
           wts[order].contiguous(), has_l0, l0_mask, has_l1)
    cache[key] = hit
    nbytes = sum(t.numel() * t.element_size() for t in hit[:5]) + (0 if l0_mask is None else l0_mask.numel() * 4)
    _tap_bytes += nbytes
    __import__("weakref").finalize(hit[0], _tap_release, nbytes)        # the budget counts LIVE lists: a dropped view cache gives its share back
    return hit


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
        H, W, C, levels, mode = ctx.meta
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None, None, None, None
        if uv.shape[0] == 0:
            # no fetch coordinates (an empty pixel shard): a deferring parameter gets no gradient part at all from this pass
            # (dist_util.reduce_texture_grads supplies zeros where other ranks hold parts), any other texture a dense zero
            if ctx.owner is not None and getattr(ctx.owner, "_texir_defer_fold", False):
                return None, None, None, None, None, None, None, None
            return torch.zeros((H, W, C), device=d_out.device, dtype=torch.float32), None, None, None, None, None, None, None
        L = _lib.lib()
        d_out = d_out.contiguous()
        owner = ctx.owner
        # FusedAdam(fuse_mip_fold=True) asks for the last fold (level 1 -> level 0, a read-modify-write of the whole texture) to be
        # left to its own read of the gradient: the level-1 gradient is parked on the parameter.  Only the first trilinear fetch of a
        # parameter per backward pass defers; a further one folds completely and autograd adds its d_tex as usual.
        defer = (mode == 1 and levels > 1 and owner is not None and getattr(owner, "_texir_defer_fold", False)
                 and getattr(owner, "_texir_grad_l1", None) is None)
        # FusedAdam can take the last TWO folds over (level 2 -> 1 -> 0): the folds then stop at level 2 and the read-modify-write of the
        # level-1 stack disappears.  Gather path only (cached tap lists); needs a level above 2 and H, W divisible by 4.
        defer_levels = 0
        if defer:
            defer_levels = 2 if (ctx.taps is not None and levels >= 4 and H % 4 == 0 and W % 4 == 0 and getattr(owner, "_texir_defer_levels", 1) >= 2) else 1
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
                    if id(owner) not in arena["clean"]:
                        arena["buf"][lo:hi].zero_()
                    arena["clean"].discard(id(owner))          # (this backward writes into it)
                else:
                    g_rest = getattr(owner, "_texir_grest", None)
                    if g_rest is None or g_rest.numel() != n_rest or g_rest.device != d_out.device:
                        if torch.cuda.is_current_stream_capturing():
                            raise _lib.TexirError("gradient stack must be allocated before hipGraph capture (run one eager step first)")
                        g_rest = torch.empty(n_rest, device=d_out.device, dtype=torch.float32)
                        owner._texir_grest = g_rest
                    g_rest.zero_()
            else:
                g_rest = torch.zeros(n_rest, device=d_out.device, dtype=torch.float32)
        # A deferred fetch over cached tap lists (single process): the level-0 gradient is SPARSE -- only the texels the lists name get a
        # value, usually none or a handful (4k textures through 128^2 cube faces sample levels >= 3).  It goes to a buffer owned by the
        # parameter that is never cleared; the view's bit mask (one bit per texel) tells the fused optimiser where to read it.  No
        # 4*H*W*C-byte fill and no dense read per step, nothing allocated per step or per captured graph; autograd gets None.
        import torch.distributed as dist
        # (several ranks: the gradient parts are summed across ranks as dense tensors, dist_util.reduce_texture_grads -- unless the texture side of the step
        # is REPLICATED on every rank, sharded_step.ShardedMatStep: then nothing is reduced and the single-process form applies)
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and not getattr(owner, "_texir_replicated_grads", False)
        sparse_l0 = bool(defer and ctx.taps is not None and owner.grad is None and not multi)
        if sparse_l0:
            d_tex = None
            if ctx.taps[5]:
                d_tex = getattr(owner, "_texir_g0", None)
                if d_tex is None or d_tex.shape != (H, W, C) or d_tex.device != d_out.device:
                    if torch.cuda.is_current_stream_capturing():
                        raise _lib.TexirError("gradient buffer must be allocated before hipGraph capture (run one eager step first)")
                    d_tex = torch.zeros((H, W, C), device=d_out.device, dtype=torch.float32)
                    owner._texir_g0 = d_tex
        else:
            d_tex = torch.zeros((H, W, C), device=d_out.device, dtype=torch.float32)
        if ctx.taps is not None:
            # fixed fetch coordinates (a cached view): deterministic gather over the pre-sorted tap lists instead of float atomics
            seg_key, starts, counts, pix, wts = ctx.taps[:5]
            _lib.check(L.texir_tex_gather_backward(_lib.ptr(d_tex), _lib.ptr(g_rest), H, W, C, levels, _lib.ptr(seg_key), _lib.ptr(starts),
                                                   _lib.ptr(counts), seg_key.numel(), _lib.ptr(pix), _lib.ptr(wts), _lib.ptr(d_out), mode,
                                                   defer_levels, _lib.stream_ptr()))
        elif defer:
            _lib.check(L.texir_tex_fetch_backward_deferred(_lib.ptr(d_tex), _lib.ptr(g_rest), H, W, C, levels, _lib.ptr(uv), _lib.ptr(uv_da),
                                                           uv.shape[0], _lib.ptr(d_out), _lib.stream_ptr()))
        else:
            _lib.check(L.texir_tex_fetch_backward(_lib.ptr(d_tex), _lib.ptr(g_rest), H, W, C, levels, _lib.ptr(uv), _lib.ptr(uv_da), mode, uv.shape[0],
                                                  _lib.ptr(d_out), _lib.stream_ptr()))
        if defer:
            n1 = (H // 2) * (W // 2) * C
            owner._texir_grad_l1 = g_rest[:n1]
            owner._texir_grad_l2 = g_rest[n1:n1 + (H // 4) * (W // 4) * C] if defer_levels == 2 else None
            # level 1 untouched by this view's taps and the fold 2 -> 1 left to the optimiser: the parked level-1 stack is all zeros (the arena fill) and
            # FusedAdam.step passes NULL for it (the stack stays parked as the marker of a deferred gradient)
            owner._texir_l1_zero = bool(defer_levels == 2 and ctx.taps is not None and not ctx.taps[7])
        if owner is not None:
            # does the level-0 gradient of this parameter hold anything at all after this backward pass?  A deferred fetch none of whose
            # pixels samples level 0 leaves d_tex all zero (the multi-GPU reduction can then skip it); any other fetch of the parameter
            # writes it.  Sticky over the fetches of one backward pass; FusedAdam.zero_grad resets it.
            wrote_l0 = (ctx.taps[5] if ctx.taps is not None else True) if defer else True
            owner._texir_l0_touched = bool(getattr(owner, "_texir_l0_touched", False)) or wrote_l0
        if sparse_l0:
            owner._texir_l0_mask = ctx.taps[6] if ctx.taps[5] else None      # (None + no .grad: level-0 gradient identically zero)
            owner._texir_l0_sparse = True
            return None, None, None, None, None, None, None, None
        return d_tex, None, None, None, None, None, None, None


def texture(tex, uv, uv_da=None, filter_mode="linear", max_mip_level=13, cache=None):
    """tex [H,W,C] (or [1,H,W,C]); uv [...,2]; uv_da [...,4] (du/dX,du/dY,dv/dX,dv/dY) -> [...,C].
    cache: a dict that lives as long as (uv, uv_da) stay the same tensors (the per-view G-buffer cache): the backward then runs as a
    deterministic gather over tap lists sorted once, instead of a float-atomic scatter."""
    if tex.dim() == 4:
        tex = tex[0]
    lead = uv.shape[:-1]
    uvf = uv.reshape(-1, 2).to(torch.float32).contiguous()
    daf = None if uv_da is None else uv_da.reshape(-1, 4).to(torch.float32).contiguous()
    mode = FILTER[filter_mode]
    if mode == 1 and daf is None:
        raise ValueError("linear-mipmap-linear needs uv_da")
    owner = tex
    if tex.dtype != torch.float32 or not tex.is_contiguous():
        tex = tex.to(torch.float32).contiguous()
    H, W, C = tex.shape
    levels = int(_lib.lib().texir_mip_levels(H, W, int(max_mip_level))) if mode == 1 else 1
    rest = _mips_for(owner, tex.detach(), levels) if levels > 1 else None
    taps = None
    arena = getattr(owner, "_texir_arena", None)
    if arena is not None and mode == 1 and tex.requires_grad and torch.is_grad_enabled() and id(owner) not in arena["clean"]:
        # first deferring fetch of a step: clear the gradient stacks of every trainable parameter of the arena with one fill (in the forward:
        # stream-ordered before every backward of the step, whichever parameter's comes first)
        # ... unless a gradient parked by an earlier backward pass is still waiting for its optimiser step (gradient accumulation: fwd, bwd, fwd,
        # bwd, step): the fill would wipe it.  The later backward passes then do not defer and fold into private stacks (texture.py backward).
        ps = [q for q in arena["params"] if q.requires_grad]
        if not any(getattr(q, "_texir_grad_l1", None) is not None for q in arena["params"]):
            arena["buf"][min(q._texir_arena_span[0] for q in ps):max(q._texir_arena_span[1] for q in ps)].zero_()
            arena["clean"] = set(id(q) for q in ps)
    if cache is not None and tex.requires_grad and torch.is_grad_enabled() and uvf.shape[0] > 0:
        taps = _tap_lists(cache, H, W, C, levels, mode, uvf, daf)
    out = _TexFetch.apply(tex, rest, uvf, daf, mode, levels, owner if isinstance(owner, torch.nn.Parameter) else None, taps)
    return out.reshape(*lead, C)
