"""Tensor-level wrappers of the C-ABI (include/branchgan.h) with autograd where the path differentiates.

Everything here launches hand-written gfx950 kernels through ctypes on torch's current HIP stream.
torch supplies device memory and autograd bookkeeping only; there is no CPU or eager fallback.
"""
import ctypes
import itertools
import math

import torch

from . import _capi as C

_ws_cache = {}
MAX_SLOTS = 64        # BG_MAX_SLOTS of include/branchgan.h: tree nodes one attention call can address

# Optional per-kernel timing with HIP events on the launch stream (bench.py turns it on): name -> [(start, end), ...]
TIMING = None


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if TIMING is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record()

    def __exit__(self, *exc):
        if TIMING is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            TIMING.setdefault(self.name, []).append((self.start, end))


def timing_summary():
    """name -> (launches, mean ms) from the recorded events (synchronises)."""
    torch.cuda.synchronize()
    return {k: (len(v), sum(s.elapsed_time(e) for s, e in v) / len(v)) for k, v in (TIMING or {}).items()}


def _workspace(nbytes, device):
    """Grow-only scratch per (device, stream) for kernels that need one (top-n partials)."""
    if nbytes == 0:
        return None
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def _rows_2d(t):
    """(…, V) -> 2-D view with a contiguous last dim and one row stride (copy only if it must)."""
    t2 = t.reshape(-1, t.shape[-1])
    if t2.stride(-1) != 1:
        t2 = t2.contiguous()
    return t2


# ---------------------------------------------------------------------------------------------------
# sampling
# ---------------------------------------------------------------------------------------------------
def topn_select(scores, n, noise=None, blocked=None, blocked_value=0.0, row_map=None, return_keys=False):
    """bg_topn_select.  scores (R_in, V) fp32/bf16; noise (R, V) fp32 or None; blocked (R, nb) int64 or None;
    row_map (R,) int64 or None (R = R_in without it).  Returns ids (R, n) int64 [, keys (R, n) fp32]."""
    C.require_cuda(scores, noise, blocked, row_map)
    scores = _rows_2d(scores)
    rows = scores.shape[0] if row_map is None else row_map.shape[0]
    V = scores.shape[1]
    if noise is not None:
        noise = _rows_2d(noise)
        if noise.dtype != torch.float32 or noise.shape != (rows, V):
            raise ValueError(f"noise must be float32 of shape ({rows}, {V}), got {noise.dtype} {tuple(noise.shape)}")
    nb = 0
    if blocked is not None:
        blocked = blocked.reshape(rows, -1).contiguous().to(torch.int64)
        nb = blocked.shape[1]
    if row_map is not None:
        row_map = row_map.contiguous().to(torch.int64)
    ids = torch.empty((rows, n), dtype=torch.int64, device=scores.device)
    keys = torch.empty((rows, n), dtype=torch.float32, device=scores.device) if return_keys else None
    lib = C.lib()
    ws_bytes = lib.bg_topn_workspace_bytes(rows, V, n)
    ws = _workspace(ws_bytes, scores.device)
    with _timed("topn_select"):
        C.check(lib.bg_topn_select(C.ptr(scores), C.dtype_code(scores), rows, V, scores.stride(0), C.ptr(row_map),
                                   C.ptr(noise), noise.stride(0) if noise is not None else 0,
                                   C.ptr(blocked), nb, float(blocked_value), n, C.ptr(ids), C.ptr(keys),
                                   C.ptr(ws), ws.numel() if ws is not None else 0, C.stream_ptr()), "bg_topn_select")
    return (ids, keys) if return_keys else ids


def topp_select(probs, n, noise, top_p, blocked=None, return_keys=False):
    """bg_topp_select.  probs, noise (R, V) fp32.  Returns ids (R, n) int64 [, keys]."""
    C.require_cuda(probs, noise, blocked)
    probs, noise = _rows_2d(probs), _rows_2d(noise)
    if probs.dtype != torch.float32 or noise.dtype != torch.float32 or probs.shape != noise.shape:
        raise ValueError("top-p takes float32 probs and noise of the same shape")
    rows, V = probs.shape
    nb = 0
    if blocked is not None:
        blocked = blocked.reshape(rows, -1).contiguous().to(torch.int64)
        nb = blocked.shape[1]
    ids = torch.empty((rows, n), dtype=torch.int64, device=probs.device)
    keys = torch.empty((rows, n), dtype=torch.float32, device=probs.device) if return_keys else None
    lib = C.lib()
    ws_bytes = lib.bg_topp_workspace_bytes(rows, V, n)
    ws = _workspace(ws_bytes, probs.device)
    C.check(lib.bg_topp_select(C.ptr(probs), rows, V, probs.stride(0), C.ptr(noise), noise.stride(0), C.ptr(blocked), nb,
                               float(top_p), n, C.ptr(ids), C.ptr(keys), C.ptr(ws), ws.numel() if ws is not None else 0,
                               C.stream_ptr()), "bg_topp_select")
    return (ids, keys) if return_keys else ids


# ---------------------------------------------------------------------------------------------------
# fused softmax statistics (lse, log-probs of chosen ids) with backward into the logits
# ---------------------------------------------------------------------------------------------------
class _SoftmaxStats(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, ids):
        C.require_cuda(logits, ids)
        x = _rows_2d(logits)
        rows, V = x.shape
        ids2 = ids.reshape(rows, -1).contiguous().to(torch.int64)
        n_ids = ids2.shape[1]
        lse = torch.empty(rows, dtype=torch.float32, device=x.device)
        logp = torch.empty((rows, n_ids), dtype=torch.float32, device=x.device)
        with _timed(f"softmax_stats_fwd[{rows}]"):
            C.check(C.lib().bg_softmax_stats_fwd(C.ptr(x), C.dtype_code(x), rows, V, x.stride(0), C.ptr(ids2), n_ids,
                                                 C.ptr(lse), C.ptr(logp), C.stream_ptr()), "bg_softmax_stats_fwd")
        ctx.save_for_backward(x, ids2, lse)
        ctx.in_shape = logits.shape
        ctx.mark_non_differentiable(ids)
        return lse.view(logits.shape[:-1]), logp.view(*logits.shape[:-1], n_ids)

    @staticmethod
    def backward(ctx, g_lse, g_logp):
        x, ids2, lse = ctx.saved_tensors
        rows, V = x.shape
        n_ids = ids2.shape[1]
        g_lse = g_lse.reshape(rows).contiguous().float() if g_lse is not None else None
        g_logp = g_logp.reshape(rows, n_ids).contiguous().float() if g_logp is not None else None
        dx = torch.empty_strided(x.shape, x.stride(), dtype=x.dtype, device=x.device)
        with _timed(f"softmax_stats_bwd[{rows}]"):
            C.check(C.lib().bg_softmax_stats_bwd(C.ptr(x), C.dtype_code(x), rows, V, x.stride(0), C.ptr(ids2), n_ids, C.ptr(lse),
                                                 C.ptr(g_lse), C.ptr(g_logp), C.ptr(dx), C.stream_ptr()), "bg_softmax_stats_bwd")
        return dx.view(ctx.in_shape), None


def rank_stats(x, labels=None, ranks=(), entropy_eps=1e-8):
    """bg_rank_stats.  x (..., V) fp32; labels (...) int64 or None; ranks: up to 8 ints >= 1.  -> dict of per-row tensors shaped
    like x without its last dim: 'label_value', 'placement' (int64; both only with labels), 'min', 'max', 'entropy', and
    'top_mass' (..., len(ranks)) = the sum of the r largest values of each row."""
    C.require_cuda(x, labels)
    if x.dtype != torch.float32:
        raise TypeError("rank_stats: fp32 probabilities only")
    lead = x.shape[:-1]
    x2 = _rows_2d(x)
    rows, V = x2.shape
    dev = x.device
    ranks = [int(r) for r in ranks]
    if len(ranks) > C.MAX_RANKS or any(r < 1 for r in ranks):
        raise ValueError(f"rank_stats: at most {C.MAX_RANKS} ranks, each >= 1")
    f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    out = {'min': f32(rows), 'max': f32(rows), 'entropy': f32(rows)}
    lab = None
    if labels is not None:
        lab = labels.reshape(rows).contiguous().to(torch.int64)
        out['label_value'], out['placement'] = f32(rows), torch.empty(rows, dtype=torch.int64, device=dev)
    if ranks:
        out['top_mass'] = f32(rows, len(ranks))
    rk = (ctypes.c_int32 * max(len(ranks), 1))(*ranks)
    with _timed("rank_stats"):
        C.check(C.lib().bg_rank_stats(C.ptr(x2), rows, V, x2.stride(0), C.ptr(lab), rk, len(ranks), float(entropy_eps),
                                      C.ptr(out.get('label_value')), C.ptr(out.get('placement')), C.ptr(out['min']), C.ptr(out['max']),
                                      C.ptr(out['entropy']), C.ptr(out.get('top_mass')), C.stream_ptr()), "bg_rank_stats")
    return {k: v.view(*lead, *v.shape[1:]) for k, v in out.items()}


def softmax_stats(logits, ids):
    """-> (lse (...,), logp (..., n_ids)): log-sum-exp over the vocabulary and log-softmax at `ids` (id < 0: 0)."""
    return _SoftmaxStats.apply(logits, ids)


# ---------------------------------------------------------------------------------------------------
# branch attention
# ---------------------------------------------------------------------------------------------------
_path_cache = {}


def path_tables(paths, device):
    """paths: list (one per query node) of slot-index lists -> cached device (path int32 [NQ][max_path], path_len)."""
    key = (tuple(tuple(p) for p in paths), str(device))
    hit = _path_cache.get(key)
    if hit is None:
        max_path = max((len(p) for p in paths), default=0)
        tab = torch.zeros((len(paths), max(max_path, 1)), dtype=torch.int32)
        for i, p in enumerate(paths):
            tab[i, :len(p)] = torch.tensor(p, dtype=torch.int32)
        lens = torch.tensor([len(p) for p in paths], dtype=torch.int32)
        hit = (tab.to(device), lens.to(device), max_path)
        _path_cache[key] = hit
    return hit


def _ptr_array(tensors):
    arr = (C.c_vp * max(len(tensors), 1))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


# Dropout on the attention probabilities (csrc/attn_dropout.h).  One (seed, step counter) pair per device, in device memory: the
# kernels read it when they run, so a captured step draws fresh masks on every replay once dropout_advance() is part of it.
# A `drop` descriptor is (p, call, row0): call tells the attention calls of one step apart and is the same in forward and
# backward; row0 is added to the call's query rows before they enter the mask function.
#
# Two ways of numbering calls.  Call ORDER (drop_descriptor(p)): the n-th call of a step, top bit set - whoever calls the attention
# functions directly.  SITES (drop_site): (pass, model, layer, kind) - the models' fused path.  A site does not depend on how the
# step is scheduled: a no-grad branch step over ONE tree node and the later differentiated pass over ALL nodes name the same site
# and line their rows up through row0 (tree rows start at TREE_ROW_BASE + uid * K), so they draw the SAME masks - which is what
# lets the sampling phase run without autograd under dropout (tree.multiStepGenForward).  The pass number moves on with every
# forward over a text (dropout_new_pass), so two micro-batches inside one step do not share masks either.
_drop_states = {}
_drop_calls = itertools.count(1)
_drop_pass = 0
TREE_ROW_BASE = 1 << 20
SITE_ATTN, SITE_RESID, SITE_MLP, SITE_EMBED = 0, 1, 2, 3
_model_salts = itertools.count(1)


def new_model_salt():
    """A small number that tells the models of a process apart in their dropout sites (generator vs discriminator)."""
    return next(_model_salts) & 0xFF


def dropout_new_pass(device=None):
    """A new forward pass over a text begins: its dropout sites (and those of the branch steps / tree passes hanging off it)
    get a fresh pass number.  Host-side: a captured graph keeps the numbers it was recorded with, the device counter moves.
    A caller that never advances the step counter itself (a plain loop over an LM) would see the 11-bit pass number wrap and
    the masks repeat: the counter is advanced for it when the numbers run out."""
    global _drop_pass
    if _drop_pass >= 0x7FE and device is not None and torch.device(device).type == "cuda":
        dropout_advance(device)
    _drop_pass += 1


def drop_site(salt, layer, kind):
    """32-bit site word (top bit clear): pass 11 bits | model 8 | layer 9 | kind 3."""
    return ((_drop_pass & 0x7FF) << 20) | ((salt & 0xFF) << 12) | ((layer & 0x1FF) << 3) | (kind & 7)


def dropout_state(device):
    """int64 (2,) on `device`: [seed, step counter]; created from torch's initial seed (seed_dropout() replaces it)."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = str(device)
    st = _drop_states.get(key)
    if st is None:
        st = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)
        _drop_states[key] = st
    return st


def seed_dropout(seed, device="cuda"):
    """Restart the attention-dropout stream of `device` at (seed, step 0), in place (captured graphs keep reading it)."""
    dropout_state(device).copy_(torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64))


def dropout_advance(device="cuda"):
    """step counter += 1 on the current stream (bg_dropout_advance): once per train step, inside the captured region.  The
    call numbers restart with it: the n-th attention call of a step has number n whether the step runs eagerly or is being
    captured (a captured step and an eager one on the same (seed, counter) then draw the same masks)."""
    global _drop_calls, _drop_pass
    C.check(C.lib().bg_dropout_advance(C.ptr(dropout_state(device)), C.stream_ptr()), "bg_dropout_advance")
    _drop_calls = itertools.count(1)
    _drop_pass = 0


def drop_descriptor(p, site=None, row0=0):
    """(p, call word, row0) for one attention call, or None when p is 0 / None.  site None: a fresh call number (call order)."""
    if not p:
        return None
    if not 0.0 < float(p) < 1.0:
        raise ValueError(f"attention dropout rate {p} is not in [0, 1)")
    call = (0x80000000 | (next(_drop_calls) & 0x7FFFFFFF)) if site is None else (int(site) & 0x7FFFFFFF)
    return (float(p), call, int(row0) & 0xFFFFFFFF)


class _DropoutRows(torch.autograd.Function):
    """bg_dropout_rows: x (B, T, C) -> x * keep / (1 - p); backward = the same call on the gradient."""

    @staticmethod
    def forward(ctx, x, p, site, row0, text_rows):
        ctx.meta = (p, site, row0, text_rows)
        return _dropout_rows_call(x, p, site, row0, text_rows)

    @staticmethod
    def backward(ctx, dy):
        p, site, row0, text_rows = ctx.meta
        return _dropout_rows_call(dy, p, site, row0, text_rows), None, None, None, None


def _dropout_rows_call(x, p, site, row0, text_rows):
    B, T, Cc = x.shape
    if x.stride(-1) != 1 or x.stride(0) != T * x.stride(1) or x.dtype not in (torch.float32, torch.bfloat16):
        x = x.contiguous() if x.dtype in (torch.float32, torch.bfloat16) else x.float().contiguous()
    y = torch.empty((B, T, Cc), dtype=x.dtype, device=x.device)
    C.check(C.lib().bg_dropout_rows(C.ptr(x), C.ptr(y), C.dtype_code(x), B, T, Cc, x.stride(1), Cc, int(text_rows), int(row0) & 0xFFFFFFFF,
                                    float(p), int(site) & 0x7FFFFFFF, C.ptr(dropout_state(x.device)), C.stream_ptr()), "bg_dropout_rows")
    return y


def row_dropout_struct(drop, device):
    """drop = (p, site, row0, text_rows, rows_per_batch) -> the C descriptor of the LayerNorm kernels' dropout variants."""
    d = C.RowDropout()
    d.p, d.site, d.row0, d.text_rows, d.rows_per_batch = float(drop[0]), int(drop[1]) & 0x7FFFFFFF, int(drop[2]) & 0xFFFFFFFF, int(drop[3]), int(drop[4])
    d.drop_state = dropout_state(device).data_ptr()
    return d


def dropout_rows(x, p, site, row0=0, text_rows=None):
    """Dropout over a (B, T, C) activation with the counter-based generator of the attention dropout (bg_dropout_rows): the
    first `text_rows` rows (default: all) are text positions, the rest are tree rows whose logical number starts at row0."""
    if not p:
        return x
    if not 0.0 < float(p) < 1.0:
        raise ValueError(f"dropout rate {p} is not in [0, 1)")
    return _DropoutRows.apply(x, float(p), int(site), int(row0), x.shape[1] if text_rows is None else int(text_rows))


def attn_dropout_mask(sh_like, device):
    """The keep mask of an attention call (bg_attn_dropout_mask; tests): uint8 (B, H, NQ*K, S_text + max_path)."""
    keep = torch.empty((sh_like.B, sh_like.H, sh_like.NQ * sh_like.K, sh_like.S_text + sh_like.max_path), dtype=torch.uint8, device=device)
    C.check(C.lib().bg_attn_dropout_mask(ctypes.byref(sh_like), C.ptr(keep), C.stream_ptr()), "bg_attn_dropout_mask")
    return keep


def _attn_shape(q, k_text, slots_k, scale, max_path, out, drop=None):
    B, H, NQ, K, Dh = q.shape
    sh = C.AttnShape()
    sh.dtype = C.dtype_code(q)
    sh.B, sh.H, sh.NQ, sh.K, sh.Dh = B, H, NQ, K, Dh
    sh.S_text = 0 if k_text is None else k_text.shape[2]
    sh.n_slots, sh.max_path, sh.scale = len(slots_k), max_path, float(scale)
    sh.q_sb, sh.q_sh, sh.q_sn, sh.q_sk = q.stride()[:4]
    sh.o_sb, sh.o_sh, sh.o_sn, sh.o_sk = out.stride()[:4]
    if k_text is not None:
        sh.t_sb, sh.t_sh, sh.t_ss = k_text.stride()[:3]
    if slots_k:
        sh.s_sb, sh.s_sh, sh.s_sk = slots_k[0].stride()[:3]
    if drop is not None:
        sh.p_drop, sh.drop_call = drop[0], drop[1]
        sh.drop_row0 = drop[2] if len(drop) > 2 else 0
        sh.drop_state = dropout_state(q.device).data_ptr()
    return sh


def _unit_last(t):
    return t if t.stride(-1) == 1 else t.contiguous()


# bench.py sets this to a list for ONE eager step: every attention launch of the step is appended as
# (kind, launch closure, shape struct, ctx_len, path_len, extra) with its tensors kept alive by the closure, so the step's
# real launch mix - same shapes, strides, data - can be replayed from a HIP graph and timed with events.
ATTN_RECORD = None


def attn_fwd_launch(sh, q, kt, vt, slots_k, slots_v, ctx_len, path, path_len, out, lse):
    """The one place bg_branch_attn_fwd is called from."""
    ka, va = _ptr_array(slots_k), _ptr_array(slots_v)
    # few query rows against a long key range (c4's branch steps): the library splits the keys of a query tile over several
    # workgroups when it is handed a workspace for their partial states
    ws_bytes = C.lib().bg_branch_attn_fwd_workspace_bytes(ctypes.byref(sh))
    ws = _workspace(ws_bytes, q.device)

    def launch():
        C.check(C.lib().bg_branch_attn_fwd_ws(ctypes.byref(sh), C.ptr(q), C.ptr(kt), C.ptr(vt), ka, va, C.ptr(ctx_len), C.ptr(path),
                                              C.ptr(path_len), C.ptr(out), C.ptr(lse), C.ptr(ws), ws.numel() if ws is not None else 0,
                                              C.stream_ptr()), "bg_branch_attn_fwd_ws")
    with _timed("branch_attn_fwd"):
        launch()
    if ATTN_RECORD is not None:
        ATTN_RECORD.append(("fwd", launch, sh, ctx_len, path_len, (slots_k, slots_v, False)))


def attn_bwd_launch(sh, q, kt, vt, slots_k, slots_v, ctx_len, path, path_len, out, lse, dout, dq, dkt, dvt, dk_in, dv_in, dsk, dsv, delta):
    """The one place bg_branch_attn_bwd is called from."""
    ka, va, dka, dva = _ptr_array(slots_k), _ptr_array(slots_v), _ptr_array(dsk), _ptr_array(dsv)

    def launch():
        C.check(C.lib().bg_branch_attn_bwd(ctypes.byref(sh), C.ptr(q), C.ptr(kt), C.ptr(vt), ka, va, C.ptr(ctx_len), C.ptr(path),
                                           C.ptr(path_len), C.ptr(out), C.ptr(lse), C.ptr(dout), C.ptr(dq), C.ptr(dkt), C.ptr(dvt),
                                           C.ptr(dk_in), C.ptr(dv_in), dka, dva, C.ptr(delta), C.stream_ptr()), "bg_branch_attn_bwd")
    with _timed("branch_attn_bwd"):
        launch()
    if ATTN_RECORD is not None:
        ATTN_RECORD.append(("bwd", launch, sh, ctx_len, path_len, (slots_k, slots_v, dk_in is not None, dsk, dsv)))


def attn_launch_work(sh, ctx_len, path_len, kind, has_text_in=False):
    """Algorithmic bytes and FLOPs of one attention launch:
    forward  bytes = visible K/V read once (the longest visible text prefix + every slot's K rows) + Q read + O write;
             flops = 4 * Dh * (visible query-key pairs);
    backward bytes = K, V read + dK, dV written (all S_text text rows and every slot) + Q, O, dO read + dQ written
                     (+ the incoming text gradients when they are added in the kernel);  flops = 2.5 x forward."""
    e = 4 if sh.dtype == C.BG_F32 else 2
    ctx = torch.clamp(ctx_len.to(torch.int64), max=sh.S_text)
    plen = path_len.to(torch.int64)[:sh.NQ] if sh.max_path else torch.zeros(sh.NQ, dtype=torch.int64)
    pairs = sh.NQ * int(ctx.sum()) + sh.K * int(plen.sum())
    q_bytes = sh.B * sh.H * sh.NQ * sh.K * sh.Dh * e
    flops = 4.0 * sh.Dh * sh.H * sh.B * pairs
    if kind == "fwd":
        kv_bytes = 2 * sh.B * sh.H * sh.Dh * (int(ctx.max()) + sh.n_slots * sh.K) * e
        return kv_bytes + 2 * q_bytes, flops
    kv_bytes = 2 * sh.B * sh.H * sh.Dh * (sh.S_text + sh.n_slots * sh.K) * e
    text_bytes = 2 * sh.B * sh.H * sh.Dh * sh.S_text * e
    return 2 * kv_bytes + 4 * q_bytes + (text_bytes if has_text_in else 0), 2.5 * flops


class _BranchAttention(torch.autograd.Function):
    """own_k / own_v (optional, (B,H,NQ,K,Dh)): the keys / values of the query nodes themselves, ONE tensor for all nodes.
    They become slots n_slots .. n_slots + NQ - 1 as views made in here, so autograd sees one input and backward returns one
    gradient tensor (slicing them outside costs a zero-fill, a copy and an add per node, per layer, for k and for v)."""

    @staticmethod
    def forward(ctx, q, k_text, v_text, ctx_len, path, path_len, max_path, scale, n_slots, own_k, own_v, drop, *slots):
        slots_k, slots_v = list(slots[:n_slots]), list(slots[n_slots:])
        C.require_cuda(q, k_text, v_text, ctx_len, *slots)
        q = _unit_last(q)
        k_text, v_text = _unit_last(k_text), _unit_last(v_text)
        if k_text.stride() != v_text.stride():
            k_text, v_text = k_text.contiguous(), v_text.contiguous()
        B, H, NQ, K, Dh = q.shape
        if own_k is not None:
            C.require_cuda(own_k, own_v)
            if own_k.stride() != own_v.stride() or own_k.stride(-1) != 1:
                own_k, own_v = own_k.contiguous(), own_v.contiguous()
            slots_k = slots_k + [own_k[:, :, n] for n in range(NQ)]
            slots_v = slots_v + [own_v[:, :, n] for n in range(NQ)]
        if slots_k:
            ref = slots_k[0].stride()
            if any(t.stride() != ref or t.stride(-1) != 1 for t in slots_k + slots_v):
                if own_k is not None:
                    raise NotImplementedError("ancestor slots must share the layout of the query nodes' own keys / values")
                slots_k = [t.contiguous() for t in slots_k]
                slots_v = [t.contiguous() for t in slots_v]
        # (B, NQ, K, H, Dh) in memory: the projection that follows reads it as (B, NQ*K, H*Dh) with no permute copy
        out = torch.empty((B, NQ, K, H, Dh), dtype=q.dtype, device=q.device).permute(0, 3, 1, 2, 4)
        lse = torch.empty((B, H, NQ, K), dtype=torch.float32, device=q.device)
        sh = _attn_shape(q, k_text, slots_k, scale, max_path, out, drop)
        attn_fwd_launch(sh, q, k_text, v_text, slots_k, slots_v, ctx_len, path, path_len, out, lse)
        has_own = own_k is not None
        ctx.save_for_backward(q, k_text, v_text, ctx_len, path, path_len, out, lse, *slots[:n_slots], *slots[n_slots:],
                              *((own_k, own_v) if has_own else ()))
        ctx.meta = (max_path, scale, n_slots, has_own, drop)
        ctx.mark_non_differentiable(lse)
        return out, lse

    @staticmethod
    def backward(ctx, dout, _dlse):
        max_path, scale, n_slots, has_own, drop = ctx.meta
        q, k_text, v_text, ctx_len, path, path_len, out, lse = ctx.saved_tensors[:8]
        anc_k = list(ctx.saved_tensors[8:8 + n_slots])
        anc_v = list(ctx.saved_tensors[8 + n_slots:8 + 2 * n_slots])
        B, H, NQ, K, Dh = q.shape
        like = lambda t: torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device)
        d_anc_k, d_anc_v = [like(t) for t in anc_k], [like(t) for t in anc_k]
        slots_k, slots_v, dks, dvs = anc_k, anc_v, list(d_anc_k), list(d_anc_v)
        d_own_k = d_own_v = None
        if has_own:
            own_k, own_v = ctx.saved_tensors[8 + 2 * n_slots:]
            d_own_k, d_own_v = like(own_k), like(own_k)
            slots_k = slots_k + [own_k[:, :, n] for n in range(NQ)]
            slots_v = slots_v + [own_v[:, :, n] for n in range(NQ)]
            dks += [d_own_k[:, :, n] for n in range(NQ)]
            dvs += [d_own_v[:, :, n] for n in range(NQ)]
        if slots_k and any(t.stride() != slots_k[0].stride() for t in slots_k + slots_v):      # forward made them contiguous
            slots_k, slots_v = [t.contiguous() for t in slots_k], [t.contiguous() for t in slots_v]
            dks, dvs = [torch.empty_like(t) for t in slots_k], [torch.empty_like(t) for t in slots_k]
            d_anc_k, d_anc_v = dks[:n_slots], dvs[:n_slots]
        if dout.stride() != out.stride():
            dout = torch.empty_strided(out.shape, out.stride(), dtype=out.dtype, device=out.device).copy_(dout)
        dq = torch.empty((B, H, NQ, K, Dh), dtype=q.dtype, device=q.device)
        dk_text, dv_text = like(k_text), like(k_text)
        delta = torch.empty_like(lse)
        sh = _attn_shape(q, k_text, slots_k, scale, max_path, out, drop)
        sh.dq_sb, sh.dq_sh, sh.dq_sn, sh.dq_sk = dq.stride()[:4]
        attn_bwd_launch(sh, q, k_text, v_text, slots_k, slots_v, ctx_len, path, path_len, out, lse, dout, dq, dk_text, dv_text, None, None,
                        dks, dvs, delta)
        return (dq, dk_text, dv_text, None, None, None, None, None, None, d_own_k, d_own_v, None, *d_anc_k, *d_anc_v)


def branch_attention(q, k_text, v_text, ctx_len, slots_k=(), slots_v=(), paths=None, scale=None, own_k=None, own_v=None, drop=None):
    """q (B,H,NQ,K,Dh) [or (B,H,K,Dh) = one node]; k_text/v_text (B,H,S,Dh); ctx_len int32 (K,) device;
    slots_k/v: lists of (B,H,K,Dh); own_k/own_v (optional, (B,H,NQ,K,Dh)): the query nodes' own keys / values, which become
    the slots after slots_k (slot len(slots_k) + n = node n); paths: per query node, the slot indices it may see (ancestors
    + itself); drop: None or drop_descriptor(p) - dropout on the attention probabilities.
    Returns (out like q [a (B,NQ,K,H,Dh)-contiguous view], lse fp32)."""
    squeeze = q.dim() == 4
    if squeeze:
        q = q.unsqueeze(2)
    NQ = q.shape[2]
    n_total = len(slots_k) + (NQ if own_k is not None else 0)
    if paths is None:
        paths = [list(range(n_total))] * NQ
    if len(paths) != NQ:
        raise ValueError(f"{len(paths)} paths for {NQ} query nodes")
    for p in paths:
        if any(s < 0 or s >= n_total for s in p):
            raise ValueError("path refers to a slot that was not passed")
    if len(slots_k) != len(slots_v):
        raise ValueError("slots_k and slots_v differ in length")
    if (own_k is None) != (own_v is None) or (own_k is not None and (own_k.shape != q.shape or own_v.shape != q.shape)):
        raise ValueError("own_k / own_v must both be given, shaped like q")
    if n_total > C.MAX_SLOTS:
        raise ValueError(f"{n_total} slots (max {C.MAX_SLOTS})")
    if ctx_len.dtype != torch.int32 or ctx_len.numel() != q.shape[3]:
        raise ValueError("ctx_len must be int32 with one entry per branch point")
    if k_text.shape[-1] != q.shape[-1]:
        raise ValueError("head dims differ")
    path, path_len, max_path = path_tables(paths, q.device)
    if scale is None:
        scale = 1.0 / math.sqrt(q.shape[-1])
    if own_k is not None and slots_k and NQ == 1 and any(t.stride() != own_k[:, :, 0].stride() for t in list(slots_k) + list(slots_v)):
        # reference-form call: the ancestors are slices of one concatenated context (getFullPast), laid out differently from the
        # step's own keys - one packed copy each (differentiable) gives every slot the same strides
        slots_k, slots_v = [t.contiguous() for t in slots_k], [t.contiguous() for t in slots_v]
        own_k, own_v = own_k.contiguous(), own_v.contiguous()
    out, lse = _BranchAttention.apply(q, k_text, v_text, ctx_len.contiguous(), path, path_len, max_path, float(scale),
                                      len(slots_k), own_k, own_v, drop, *slots_k, *slots_v)
    return (out.squeeze(2), lse.squeeze(2)) if squeeze else (out, lse)


def branch_attention_weights(q, k_text, ctx_len, slots_k, paths, scale, lse, own_k=None, drop=None):
    """The softmax weights of a branch_attention call, on request (bg_branch_attn_weights; no gradient).  Arguments as
    branch_attention; lse is the one it returned.  -> fp32 (B,H,NQ,K,W), W = S + (max_path - 1) * K + 1, the reference's
    column order [text | one K-column block per ancestor | self]."""
    squeeze = q.dim() == 4
    if squeeze:
        q, lse = q.unsqueeze(2), lse.unsqueeze(2)
        own_k = own_k.unsqueeze(2) if own_k is not None and own_k.dim() == 4 else own_k
    q, k_text = _unit_last(q.detach()), _unit_last(k_text.detach())
    B, H, NQ, K, Dh = q.shape
    slots = [t.detach() for t in slots_k]
    if own_k is not None:
        own_k = _unit_last(own_k.detach())
        slots = slots + [own_k[:, :, n] for n in range(NQ)]
    if slots and any(t.stride() != slots[0].stride() or t.stride(-1) != 1 for t in slots):
        slots = [t.contiguous() for t in slots]
    if paths is None:
        paths = [list(range(len(slots)))] * NQ
    path, path_len, max_path = path_tables(paths, q.device)
    W = k_text.shape[2] + (max(max_path - 1, 0) * K + 1 if max_path else 0)
    weights = torch.empty((B, H, NQ, K, max(W, 1)), dtype=torch.float32, device=q.device)
    sh = _attn_shape(q, k_text, slots, scale, max_path, q, drop)     # drop: the forward's descriptor -> the weights after dropout
    lse = lse.detach().contiguous().float()
    C.check(C.lib().bg_branch_attn_weights(ctypes.byref(sh), C.ptr(q), C.ptr(k_text), _ptr_array(slots), C.ptr(ctx_len.contiguous()),
                                           C.ptr(path), C.ptr(path_len), C.ptr(lse), C.ptr(weights), weights.shape[-1], C.stream_ptr()),
            "bg_branch_attn_weights")
    return weights.squeeze(2) if squeeze else weights


_arange_cache = {}


def causal_ctx_len(T_, S_, device):
    """ctx_len of the root causal pass (query i sees keys j <= i + S - T), cached per device."""
    key = (T_, S_, str(device))
    ctx = _arange_cache.get(key)
    if ctx is None:
        ctx = torch.arange(S_ - T_ + 1, S_ + 1, dtype=torch.int32, device=device)
        _arange_cache[key] = ctx
    return ctx


def causal_attention(q, k, v, scale=None, drop=None):
    """Root causal pass as the same kernel: query i sees keys j <= i + (S - T).  q (B,H,T,Dh), k/v (B,H,S,Dh)."""
    ctx = causal_ctx_len(q.shape[2], k.shape[2], q.device)
    out, lse = branch_attention(q, k, v, ctx, (), (), paths=[[]], scale=scale, drop=drop)
    return out, lse


# ---------------------------------------------------------------------------------------------------
# NeoX prologue: split fused QKV, heads forward, partial rotary
# ---------------------------------------------------------------------------------------------------
class _NeoxQkvRope(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, cos_tab, sin_tab, position_ids, H, Dh, rot, grad_buffer=None):
        ctx.grad_buffer = grad_buffer
        C.require_cuda(qkv, cos_tab, sin_tab, position_ids)
        qkv = qkv.contiguous()
        B, T = qkv.shape[:2]
        pos = position_ids.to(torch.int32).contiguous()
        q, k, v = [torch.empty((B, H, T, Dh), dtype=qkv.dtype, device=qkv.device) for _ in range(3)]
        C.check(C.lib().bg_neox_qkv_rope_fwd(C.ptr(qkv), C.dtype_code(qkv), B, T, H, Dh, rot, C.ptr(cos_tab), C.ptr(sin_tab),
                                             cos_tab.shape[0], C.ptr(pos), C.ptr(q), C.ptr(k), C.ptr(v), C.stream_ptr()),
                "bg_neox_qkv_rope_fwd")
        ctx.save_for_backward(cos_tab, sin_tab, pos)
        ctx.meta = (B, T, H, Dh, rot, qkv.dtype)
        return q, k, v

    @staticmethod
    def backward(ctx, dq, dk, dv):
        cos_tab, sin_tab, pos = ctx.saved_tensors
        B, T, H, Dh, rot, dtype = ctx.meta
        zeros = lambda: torch.zeros((B, H, T, Dh), dtype=dtype, device=pos.device)
        dq = dq.contiguous() if dq is not None else zeros()
        dk = dk.contiguous() if dk is not None else zeros()
        dv = dv.contiguous() if dv is not None else zeros()
        dqkv = ctx.grad_buffer(B * T, H * 3 * Dh, dtype) if ctx.grad_buffer is not None else None
        dqkv = dqkv.view(B, T, H * 3 * Dh) if dqkv is not None else torch.empty((B, T, H * 3 * Dh), dtype=dtype, device=pos.device)
        C.check(C.lib().bg_neox_qkv_rope_bwd(C.ptr(dq), C.ptr(dk), C.ptr(dv), C.dtype_code(dq), B, T, H, Dh, rot, C.ptr(cos_tab),
                                             C.ptr(sin_tab), cos_tab.shape[0], C.ptr(pos), C.ptr(dqkv), C.stream_ptr()),
                "bg_neox_qkv_rope_bwd")
        return dqkv, None, None, None, None, None, None, None


def neox_qkv_rope(qkv, cos_tab, sin_tab, position_ids, n_head, head_dim, rot_dims, grad_buffer=None):
    """qkv (B,T,3*hidden) in the NeoX per-head interleaved order -> q, k (rotated), v, each (B,H,T,Dh).
    grad_buffer(rows, features, dtype) -> a contiguous buffer for the gradient of qkv, or None (fused.py: the dy stack of the projection)."""
    return _NeoxQkvRope.apply(qkv, cos_tab, sin_tab, position_ids, n_head, head_dim, rot_dims, grad_buffer)


# ---------------------------------------------------------------------------------------------------
# fused losses
# ---------------------------------------------------------------------------------------------------
_links_cache = {}


def link_tables(root_links, alt_links, device):
    key = (tuple(root_links), tuple(tuple(l) for l in alt_links), str(device))
    hit = _links_cache.get(key)
    if hit is None:
        max_child = max(1, max((len(l) for l in alt_links), default=1))
        child = torch.full((len(alt_links), max_child), -1, dtype=torch.int32)
        for i, l in enumerate(alt_links):
            if l:
                child[i, :len(l)] = torch.tensor(l, dtype=torch.int32)
        n_child = torch.tensor([len(l) for l in alt_links], dtype=torch.int32)
        root = torch.tensor(list(root_links), dtype=torch.int32)
        hit = (root.to(device), child.to(device), n_child.to(device), max_child)
        _links_cache[key] = hit
    return hit


def _f32c(t):
    """fp32 and contiguous in ONE copy (or none): .contiguous().float() was two launches for a strided bf16 tensor."""
    if t.dtype == torch.float32:
        return t.contiguous()
    return t.to(torch.float32, memory_format=torch.contiguous_format)


class _DiscLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input_scores, alt_scores, txt_in, txt_alt, alt_alt, depth_mask, root, child, n_child, max_child, n_update):
        C.require_cuda(input_scores, alt_scores, txt_in, txt_alt, alt_alt, depth_mask)
        B, S = input_scores.shape
        _, N, K = alt_scores.shape
        sh = C.DiscLossShape(B, S, N, K, min(int(n_update), N), root.numel(), max_child)
        args = [_f32c(t) for t in (input_scores, alt_scores, depth_mask, txt_in, txt_alt, alt_alt)]
        out = torch.empty(5, dtype=torch.float32, device=input_scores.device)
        C.check(C.lib().bg_disc_loss_fwd(ctypes.byref(sh), *[C.ptr(t) for t in args], C.ptr(root), C.ptr(child), C.ptr(n_child),
                                         C.ptr(out), C.stream_ptr()), "bg_disc_loss_fwd")
        ctx.save_for_backward(*args, root, child, n_child)
        ctx.sh = sh
        return out

    @staticmethod
    def backward(ctx, g_out):
        *args, root, child, n_child = ctx.saved_tensors
        sh = ctx.sh
        # out = [input, alt, vh_in, vh_alt, vh] with vh = (vh_in + vh_alt) / 2: fold the parts' gradients into vh's
        g = torch.stack((g_out[0], g_out[1], g_out[4])).contiguous().float()
        d = [torch.empty_like(args[i]) for i in (0, 1, 3, 4, 5)]
        C.check(C.lib().bg_disc_loss_bwd(ctypes.byref(sh), *[C.ptr(t) for t in args], C.ptr(root), C.ptr(child), C.ptr(n_child),
                                         C.ptr(g), *[C.ptr(t) for t in d], C.stream_ptr()), "bg_disc_loss_bwd")
        d_in, d_alt, d_txt_in, d_txt_alt, d_alt_alt = d
        return d_in, d_alt, d_txt_in, d_txt_alt, d_alt_alt, None, None, None, None, None, None


def disc_losses(input_scores, alt_scores, txt_in, txt_alt, alt_alt, depth_mask, root_links, alt_links, n_update):
    """Fused DiscriminatorLoss.py:4-74.  input_scores (B,S); alt_scores (B,N,K); txt_in/txt_alt (B,S); alt_alt (N,B,K);
    depth_mask (N,K).  -> fp32 [inputLoss, alterationLoss, ValueHead-InputLoss, ValueHead-AltLoss, ValueHeadLoss]
    (differentiate through elements 0, 1 and 4; 2 and 3 are reported parts of 4)."""
    root, child, n_child, max_child = link_tables(root_links, alt_links, input_scores.device)
    out = _DiscLosses.apply(input_scores, alt_scores, txt_in, txt_alt, alt_alt, depth_mask, root, child, n_child, max_child, n_update)
    return out


class _RlLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, label_lp, alt_lp, input_scores, alt_scores, in_head, alt_head, depth_mask, wA, wB, wC, ignore_in, ignore_alt,
                probs_input, diagnostics):
        C.require_cuda(label_lp, alt_lp, input_scores, alt_scores, in_head, alt_head, depth_mask)
        B, S = input_scores.shape
        _, N, K = alt_scores.shape
        sh = C.RlLossShape(B, S, N, K, float(wA), float(wB), float(wC), int(ignore_in), int(ignore_alt),
                           int(depth_mask is not None), int(probs_input))
        args = [_f32c(t) for t in (input_scores, alt_scores, in_head, alt_head, label_lp, alt_lp)]
        mask = _f32c(depth_mask) if depth_mask is not None else None
        out = torch.empty(2, dtype=torch.float32, device=input_scores.device)
        diag = torch.empty(C.RL_DIAG_COUNT if diagnostics else 0, dtype=torch.float32, device=input_scores.device)
        C.check(C.lib().bg_rl_loss_fwd(ctypes.byref(sh), *[C.ptr(t) for t in args], C.ptr(mask), C.ptr(out),
                                       C.ptr(diag) if diagnostics else None, C.stream_ptr()), "bg_rl_loss_fwd")
        ctx.save_for_backward(*args, *([mask] if mask is not None else []))
        ctx.sh = sh
        ctx.shapes = (label_lp.shape, alt_lp.shape)
        ctx.mark_non_differentiable(diag)
        return out, diag

    @staticmethod
    def backward(ctx, g_out, _g_diag):
        saved = ctx.saved_tensors
        args, mask = list(saved[:6]), (saved[6] if len(saved) > 6 else None)
        d_label, d_alt = torch.empty_like(args[4]), torch.empty_like(args[5])
        g = g_out.contiguous().float()
        C.check(C.lib().bg_rl_loss_bwd(ctypes.byref(ctx.sh), *[C.ptr(t) for t in args], C.ptr(mask), C.ptr(g), C.ptr(d_label),
                                       C.ptr(d_alt), C.stream_ptr()), "bg_rl_loss_bwd")
        return (d_label.view(ctx.shapes[0]), d_alt.view(ctx.shapes[1])) + (None,) * 12


RL_DIAG_KEYS = ('Rl-Loss-A-Input', 'Rl-Loss-B-Input', 'RL-Loss-C-Input', 'Rl-Loss-A-Alt', 'Rl-Loss-B-Alt', 'RL-Loss-C-Alt',
                'Rl-Advantage-A-Input', 'Rl-Advantage-B-Input', 'RL-Advantage-C-Input', 'Rl-Advantage-A-Alt', 'Rl-Advantage-B-Alt',
                'RL-Advantage-C-Alt', 'RL-ZoneA-Input-Factor', 'RL-ZoneB-Input-Factor', 'RL-ZoneC-Input-Factor',
                'RL-ZoneA-Alt-Factor', 'RL-ZoneB-Alt-Factor', 'RL-ZoneC-Alt-Factor', 'RL-CorrectValueHeads-Factor',
                'RL-CorrectInput-Factor')   # the order of GeneratorLosses.py:66-81 = the order of bg_rl_loss_fwd's out_diag


def rl_losses(label_lp, alt_lp, input_scores, alt_scores, in_head, alt_head, depth_mask, wA=1.0, wB=1.0, wC=1.0,
              ignore_input=True, ignore_alt=True, probs_input=False, diagnostics=False):
    """Fused AdvantageCalculations.py:5-46 + ReinforceLoss.py:21-52 + GeneratorLosses.py:96-121.
    label_lp (B,S-1), alt_lp (B,N,K): log-probs (or probs with probs_input=True) of the true / sampled tokens; the
    discriminator tensors carry no gradient.  -> fp32 [loss-RL-input, loss-RL-alterations]; with diagnostics=True a pair
    (that, fp32 [20] = _calculateDebuggingStats of GeneratorLosses.py:34-81 in RL_DIAG_KEYS order, no gradient)."""
    out, diag = _RlLosses.apply(label_lp, alt_lp, input_scores.detach(), alt_scores.detach(), in_head.detach(), alt_head.detach(),
                                depth_mask, wA, wB, wC, ignore_input, ignore_alt, probs_input, diagnostics)
    return (out, diag) if diagnostics else out


# ---------------------------------------------------------------------------------------------------
# The element-wise pieces of the generator loss under the reference's own function boundaries
# ---------------------------------------------------------------------------------------------------
def dual_head_advantages(input_scores, alt_scores, in_head, alt_head, wA, wB, wC, ignore_input, ignore_alt, epsilon=1e-5,
                         min_denominator=1e-5):
    """bg_dual_head_advantages (AdvantageCalculations.py:5-46).  input_scores (B,S), alt_scores (B,N,S), heads (B,S), fp32.
    -> advantages (B,1+N,S), zoneA, zoneB, zoneC (B,1+N,S), heads_correct (B,S), input_correct (B,S); no gradient."""
    C.require_cuda(input_scores, alt_scores, in_head, alt_head)
    f = lambda t: t.detach().contiguous().float()
    input_scores, alt_scores, in_head, alt_head = f(input_scores), f(alt_scores), f(in_head), f(alt_head)
    B, N, S = alt_scores.shape
    if input_scores.shape != (B, S) or in_head.shape != (B, S) or alt_head.shape != (B, S):
        raise ValueError(f"dual_head_advantages: scores (B,S) / (B,N,S) and heads (B,S) expected, got {tuple(input_scores.shape)}, "
                         f"{tuple(alt_scores.shape)}, {tuple(in_head.shape)}, {tuple(alt_head.shape)}")
    dev = alt_scores.device
    big = [torch.empty((B, 1 + N, S), dtype=torch.float32, device=dev) for _ in range(4)]
    small = [torch.empty((B, S), dtype=torch.float32, device=dev) for _ in range(2)]
    C.check(C.lib().bg_dual_head_advantages(C.ptr(input_scores), C.ptr(alt_scores), C.ptr(in_head), C.ptr(alt_head), B, N, S,
                                            float(wA), float(wB), float(wC), int(bool(ignore_input)), int(bool(ignore_alt)),
                                            float(epsilon), float(min_denominator), *[C.ptr(t) for t in big + small], C.stream_ptr()),
            "bg_dual_head_advantages")
    return (*big, *small)


class _ReinforceTerms(torch.autograd.Function):
    @staticmethod
    def forward(ctx, probs, adv, mask_c, use_min_prob, neg_push, eps):
        C.require_cuda(probs, adv, mask_c)
        shape = torch.broadcast_shapes(probs.shape, adv.shape, *(() if mask_c is None else (mask_c.shape,)))
        f = lambda t: t.detach().expand(shape).contiguous().float()
        p, a = f(probs), f(adv)
        m = None if mask_c is None else f(mask_c)
        loss = torch.empty(shape, dtype=torch.float32, device=p.device)
        C.check(C.lib().bg_reinforce_terms_fwd(C.ptr(p), C.ptr(a), C.ptr(m), p.numel(), int(bool(use_min_prob)), float(neg_push), float(eps),
                                               C.ptr(loss), C.stream_ptr()), "bg_reinforce_terms_fwd")
        ctx.save_for_backward(p, a, *(() if m is None else (m,)))
        ctx.meta = (int(bool(use_min_prob)), float(neg_push), float(eps), probs.shape, m is not None)
        return loss

    @staticmethod
    def backward(ctx, g):
        use_min_prob, neg_push, eps, in_shape, has_mask = ctx.meta
        p, a = ctx.saved_tensors[:2]
        m = ctx.saved_tensors[2] if has_mask else None
        g = g.contiguous().float()
        d = torch.empty_like(p)
        C.check(C.lib().bg_reinforce_terms_bwd(C.ptr(p), C.ptr(a), C.ptr(m), p.numel(), use_min_prob, neg_push, eps, C.ptr(g), C.ptr(d),
                                               C.stream_ptr()), "bg_reinforce_terms_bwd")
        if tuple(d.shape) != tuple(in_shape):
            d = d.sum_to_size(in_shape)
        return d, None, None, None, None, None


def reinforce_terms(probs, advantages, mask_zone_c=None, use_min_prob_limit=False, neg_push_prob_limit=1e-8, eps=1e-6):
    """bg_reinforce_terms_* (ReinforceLoss.py:21-52): per-element REINFORCE loss, mirrored inside zone C when a mask is given;
    differentiable in probs."""
    return _ReinforceTerms.apply(probs, advantages, mask_zone_c, use_min_prob_limit, neg_push_prob_limit, eps)
