"""Fused transformer-block path: the glue between the GEMMs as hand-written kernels, gradients accumulated in place.

Same mathematics as the module-by-module path of models.py (NonResidualDiscriminatorBase.py:46-90 block forward,
NonResidualAttention.py:253-281, GPT2MLP; NeoXNonResidualBase.py:73-133), but built from four autograd Functions whose
forward/backward launch exactly the kernels the data flow needs:

  * `layer_norm` / `add_layer_norm`  - residual add + LayerNorm in one pass (bg_add_layernorm_fwd); backward gives the
    residual gradient in fp32 AND its low-precision copy for the preceding projection, and dgamma/dbeta (bg_layernorm_bwd);
  * `linear`  - GEMM through hipBLASLt in the compute dtype with the weight's low-precision copy cached per optimizer step;
    backward accumulates dW (fp32) and the bias gradient (bg_bias_act_bwd, with the GELU backward folded in) straight into
    `.grad` - no cast kernel, no AccumulateGrad add, no separate reduction;
  * `packed_attention`  - branch / causal attention reading q, k, v as strided slices of the packed QKV projection and
    writing dq, dk, dv straight into one packed dqkv buffer (bg_attn_shape.dq_*): no split / permute / cat copies.

The residual stream and all statistics stay fp32 (what bf16 autocast computes too); `compute_dtype()` picks bf16 under
autocast and fp32 otherwise, so the parity tests exercise the same code in fp32.
"""
import ctypes
import os

import torch
import torch.nn.functional as F

from . import _capi as C
from . import ops

# Parameter gradients are added into `.grad` inside backward (and None is returned to autograd).  Set False to get the
# gradients through autograd instead (torch.autograd.grad, hooks).
ACCUMULATE_IN_PLACE = True
# BRANCHGAN_FUSED=0 switches the models back to the module-by-module path (A/B runs, debugging)
ENABLED = os.environ.get("BRANCHGAN_FUSED", "1") != "0"

module_path_forwards = 0       # model forwards that did NOT take this path (models._fused_eligible)
_cast_cache = {}
_capture_epoch = 0
_step_epoch = 0


def _after_optimizer_step(*_):
    """Fused / foreach optimizers update parameters through list kernels that do not always bump Tensor._version, so every
    optimizer step also invalidates the low-precision weight copies."""
    global _step_epoch
    _step_epoch += 1


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook  # noqa: E402

_register_step_hook(_after_optimizer_step)


_pinned = {}      # id(parameter) -> (weakref to it, low-precision copy, weakref to the optimizer that keeps the copy current)


def pin_weight_copy(p, copy, owner):
    """branchgan.optim.ArenaAdam writes the low-precision copy of each parameter in the same pass that updates it; such a
    copy is always current, lives at a fixed address (safe to read from a captured graph) and is never re-cast here."""
    import weakref
    _pinned[id(p)] = (weakref.ref(p), copy, weakref.ref(owner))


def _pinned_copy(p, dtype):
    pin = _pinned.get(id(p))
    if pin is None:
        return None
    if pin[0]() is not p or pin[2]() is None:        # the parameter or its optimizer is gone
        del _pinned[id(p)]
        return None
    return pin[1] if pin[1].dtype == dtype else None


def invalidate_weights():
    """Call after changing parameters by hand (load_state_dict, .data assignments) while a step is in flight."""
    _after_optimizer_step()
    for owner in {id(o): o for o in (pin[2]() for pin in _pinned.values()) if o is not None}.values():
        owner.refresh_weight_copies()


def begin_capture():
    """graphs.py calls this right before a HIP-graph capture: weight copies made earlier must not be reused inside it."""
    global _capture_epoch
    _capture_epoch += 1


def clear_cast_cache():
    _cast_cache.clear()


def compute_dtype(ref):
    if torch.is_autocast_enabled("cuda"):
        return torch.get_autocast_dtype("cuda")
    return ref.dtype


def _cache_key(p):
    return (p._version, p.data_ptr(), _step_epoch, torch.cuda.is_current_stream_capturing(), _capture_epoch)


def lowp(p, dtype):
    """The parameter in the compute dtype; one cast per parameter version (optimizer step), shared by every tree node's
    forward and backward of that step.  `prepare_weights` refreshes a whole model's copies with one multi-tensor launch."""
    if p.dtype == dtype:
        return p.detach()
    pinned = _pinned_copy(p, dtype)
    if pinned is not None:
        return pinned
    hit = _cast_cache.get(id(p))
    if hit is not None and hit[0] is p and hit[1] == _cache_key(p) and hit[2].dtype == dtype:
        return hit[2]
    capturing = torch.cuda.is_current_stream_capturing()
    reuse = hit is not None and hit[0] is p and hit[2].dtype == dtype and hit[3] == (capturing, _capture_epoch if capturing else 0)
    t = hit[2] if reuse else torch.empty_like(p, dtype=dtype)      # same memory pool (eager / this capture): overwrite in place
    t.copy_(p.detach())
    _cast_cache[id(p)] = (p, _cache_key(p), t, (capturing, _capture_epoch if capturing else 0))
    return t


def prepare_weights(module, dtype):
    """Refreshes the low-precision copies of every stale >= 2-D parameter (and bias) of `module` in one multi-tensor copy
    instead of one cast kernel per parameter.  Called at the top of a fused model forward; cheap when nothing is stale."""
    if dtype == torch.float32:
        return
    capturing = torch.cuda.is_current_stream_capturing()
    pool = (capturing, _capture_epoch if capturing else 0)
    src, dst = [], []
    params = list(module.parameters())
    if any(_pinned_copy(p, dtype) is not None for p in params):
        return      # an ArenaAdam keeps this model's copies current; a parameter outside its arenas (an unused head) is cast by
                    # lowp() if it is ever used - not every step
    for p in params:
        if p.dtype != torch.float32:
            continue
        hit = _cast_cache.get(id(p))
        key = _cache_key(p)
        if hit is not None and hit[0] is p and hit[1] == key and hit[2].dtype == dtype:
            continue
        reuse = hit is not None and hit[0] is p and hit[2].dtype == dtype and hit[3] == pool
        t = hit[2] if reuse else torch.empty_like(p, dtype=dtype)
        _cast_cache[id(p)] = (p, key, t, pool)
        src.append(p.detach())
        dst.append(t)
    if src:
        torch._foreach_copy_(dst, src)


def _grad_buffer(p):
    """fp32 buffer the kernels accumulate into: p.grad itself when in-place accumulation is on."""
    if ACCUMULATE_IN_PLACE and p.requires_grad:
        if p.grad is None:
            p.grad = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
        if p.grad.dtype == torch.float32 and p.grad.is_contiguous():
            return p.grad, None
    buf = torch.zeros(p.shape, dtype=torch.float32, device=p.device)
    return buf, buf


_ws = {}
# Work captured into graphs that are REPLAYED side by side (graphs.py: the discriminator's backward beside the generator's) is
# captured on one stream: its scratch is told apart by the lane the capture declares.
_lane = 0


def set_lane(lane):
    global _lane
    previous, _lane = _lane, lane
    return previous


def _workspace(nbytes, device):
    """Grow-only scratch per (device, stream, lane): kernels of two streams may run at the same time."""
    key = (device, torch.cuda.current_stream(device).cuda_stream, _lane)
    ws = _ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 4 << 20), dtype=torch.uint8, device=device)
        _ws[key] = ws
    return ws


# Column reductions (LayerNorm dgamma / dbeta, bias gradients) end in a small second kernel that adds per-workgroup partials
# into the fp32 gradient.  Nothing inside a backward pass reads those gradients, so the reduction launches (4 per transformer
# block, ~5 us each, latency-bound) are collected and run as ONE batched launch per ~96 reductions when the backward pass
# ends (an autograd-engine callback: `loss.backward()` returns with complete gradients, nothing for callers to remember).
# BRANCHGAN_DEFER_REDUCE=0 reduces behind every producer again.
DEFER_REDUCTIONS = os.environ.get("BRANCHGAN_DEFER_REDUCE", "1") != "0"


class _DeferredReductions:
    """Per device: a grow-only list of chunks the producers' partials are bump-allocated from, and the reductions waiting."""
    CHUNK = 64 << 20

    def __init__(self, device):
        self.device = device
        self.chunks, self.cur, self.off = [], 0, 0
        self.pending, self.streams, self.armed, self.task = [], set(), False, -1

    def begin(self):
        """Called by a producer about to defer.  False: not inside a backward pass (no engine callback possible), reduce now.
        The callback is queued once per backward pass (autograd graph task).  Reductions still pending from ANOTHER pass - a
        reentrant backward inside this one, or a pass that raised before its callback ran - are reduced first: none is ever
        lost, and their workspace is free again."""
        task = torch._C._current_graph_task_id()
        if task < 0:
            return False
        if self.armed and self.task == task:
            return True
        if self.pending:
            self.flush()
        try:
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)
        except RuntimeError:
            self.armed = False
            return False
        self.armed, self.task = True, task
        return True

    def alloc(self, nbytes):
        nbytes = (int(nbytes) + 255) & ~255
        while True:
            if self.cur == len(self.chunks):
                self.chunks.append(torch.empty(max(self.CHUNK, nbytes), dtype=torch.uint8, device=self.device))
            chunk = self.chunks[self.cur]
            if self.off + nbytes <= chunk.numel():
                view = chunk[self.off:self.off + nbytes]
                self.off += nbytes
                return view
            self.cur, self.off = self.cur + 1, 0

    def add(self, partial_ptr, out, groups, stride, n):
        self.pending.append((partial_ptr, out, groups, stride, n))
        self.streams.add(torch.cuda.current_stream(self.device))

    def flush(self):
        self.armed = False
        pending, self.pending = self.pending, []
        streams, self.streams = self.streams, set()
        self.cur = self.off = 0
        if not pending:
            return
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            for st in streams:
                if st != cur:
                    cur.wait_stream(st)
            descs = (C.ReduceDesc * len(pending))()
            for d, (pp, out, groups, stride, n) in zip(descs, pending):
                d.partial, d.out, d.groups, d.stride, d.n = pp, out.data_ptr(), groups, stride, n
            with ops._timed("partial_reduce"):
                C.check(C.lib().bg_partial_reduce(descs, len(pending), C.stream_ptr()), "bg_partial_reduce")


_deferred = {}


def _deferred_for(device):
    d = _deferred.get((device, _lane))
    if d is None:
        d = _deferred[(device, _lane)] = _DeferredReductions(device)
    return d


def supported_width(C_):
    return C_ % 8 == 0 and C_ <= 4096


# ---------------------------------------------------------------------------------------------------------------------
# LayerNorm (+ residual add)
# ---------------------------------------------------------------------------------------------------------------------
def _ln_fwd(x2, delta2, gamma, beta, eps, out_dtype, second=None, y=None, pair=None, y_b=None, drop=None):
    """pair = (gamma_b, beta_b): a second LayerNorm of the same rows (-> y_b, returned last).
    drop = (p, site, row0, text_rows, rows_per_batch): delta enters through a dropout (bg_add_layernorm_dropout_fwd)."""
    rows, C_ = x2.shape
    if y is None:
        y = torch.empty((rows, C_), dtype=out_dtype, device=x2.device)
    if drop is not None:
        mean = torch.empty(rows, dtype=torch.float32, device=x2.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x2.device)
        x_out = torch.empty_like(x2)
        desc = ops.row_dropout_struct(drop, x2.device)
        with ops._timed("add_layernorm_fwd"):
            C.check(C.lib().bg_add_layernorm_dropout_fwd(C.ptr(x2), C.ptr(delta2), C.dtype_code(delta2), rows, C_, C.ptr(gamma), C.ptr(beta), float(eps),
                                                         C.ptr(x_out), C.ptr(y), C.dtype_code(y), C.ptr(mean), C.ptr(rstd), ctypes.byref(desc),
                                                         C.stream_ptr()), "bg_add_layernorm_dropout_fwd")
        return x_out, y, mean, rstd
    if pair is not None and y_b is None:
        y_b = torch.empty((rows, C_), dtype=out_dtype, device=x2.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x2.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x2.device)
    x_out = torch.empty_like(x2) if delta2 is not None else None
    with ops._timed("add_layernorm_fwd"):
        C.check(C.lib().bg_add_layernorm_pair_fwd(C.ptr(x2), C.ptr(delta2), C.ptr(second), C.dtype_code(delta2) if delta2 is not None else 0, rows, C_,
                                                  C.ptr(gamma), C.ptr(beta), C.ptr(pair[0]) if pair else None, C.ptr(pair[1]) if pair else None,
                                                  float(eps), C.ptr(x_out), C.ptr(y), C.ptr(y_b) if pair else None, C.dtype_code(y),
                                                  C.ptr(mean), C.ptr(rstd), C.stream_ptr()), "bg_add_layernorm_pair_fwd")
    if pair is not None:
        return x_out, y, mean, rstd, y_b
    return x_out, y, mean, rstd


def _ln_bwd(dy2, x2, mean, rstd, gamma_p, beta_p, dres2, lp_dtype, colsum_p=None, delta_from=None, drop=None):
    rows, C_ = x2.shape
    dx = torch.empty_like(x2)
    dx_lp = None
    if drop is not None or lp_dtype not in (None, torch.float32):
        if lp_dtype is None:
            lp_dtype = torch.float32        # through a dropout the delta's gradient is not the residual stream's: its own tensor
        if delta_from is not None:          # the gradient of the projection that produced delta: into its dy stack
            dx_lp = _deferred_dw_for(x2.device).dy_slab(delta_from[0], delta_from[1], rows, C_, lp_dtype)
        if dx_lp is None:
            dx_lp = torch.empty((rows, C_), dtype=lp_dtype, device=x2.device)
    dg, dg_ret = _grad_buffer(gamma_p)
    db, db_ret = _grad_buffer(beta_p)
    dc, dc_ret = _grad_buffer(colsum_p) if colsum_p is not None else (None, None)
    lib = C.lib()
    nbytes = lib.bg_layernorm_bwd_workspace_bytes(rows, C_)
    later = _deferred_for(x2.device) if DEFER_REDUCTIONS and dg_ret is None and db_ret is None and dc_ret is None else None
    if later is not None and not later.begin():
        later = None
    ws = later.alloc(nbytes) if later is not None else _workspace(nbytes, x2.device)
    with ops._timed("layernorm_bwd"):
        common = (C.ptr(dy2), C.dtype_code(dy2), C.ptr(x2), C.ptr(mean), C.ptr(rstd), C.ptr(gamma_p.detach()),
                  C.ptr(dres2), rows, C_, C.ptr(dx), C.ptr(dx_lp), C.dtype_code(dx_lp) if dx_lp is not None else 0,
                  *((None, None, None, int(dc is not None)) if later is not None else (C.ptr(dg), C.ptr(db), C.ptr(dc), 0)),
                  C.ptr(ws), ws.numel())
        if drop is not None:
            desc = ops.row_dropout_struct(drop, x2.device)
            C.check(lib.bg_layernorm_dropout_bwd(*common, ctypes.byref(desc), C.stream_ptr()), "bg_layernorm_dropout_bwd")
        else:
            C.check(lib.bg_layernorm_bwd(*common, C.stream_ptr()), "bg_layernorm_bwd")
    if later is not None:
        groups, stride, base = lib.bg_layernorm_bwd_groups(rows), (3 if dc is not None else 2) * C_, ws.data_ptr()
        for i, out in enumerate((dg, db, dc)):
            if out is not None:
                later.add(base + 4 * C_ * i, out, groups, stride, C_)
    return dx, dx_lp, dg_ret, db_ret, dc_ret


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, out_dtype, y_buf=None):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        _, y, mean, rstd = _ln_fwd(x2, None, gamma.detach(), beta.detach(), eps, out_dtype, y=y_buf[0] if y_buf else None)
        ctx.save_for_backward(x2, mean, rstd)
        ctx.params = (gamma, beta)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd = ctx.saved_tensors
        gamma, beta = ctx.params
        dx, _, dg, db, _ = _ln_bwd(dy.reshape(x2.shape).contiguous(), x2, mean, rstd, gamma, beta, None, None)
        return dx.view(dy.shape), dg, db, None, None, None


class _AddLayerNorm(torch.autograd.Function):
    """delta_b (optional): a second update added in the same pass (x + delta + delta_b: the attention and MLP outputs of a
    parallel-residual block); both get the same gradient.  delta_bias (optional): the bias parameter of the projection that produced `delta`.  Its gradient is the column sum of
    delta's gradient, which this backward has in registers anyway - the projection is then built with bias_grad=False."""

    @staticmethod
    def forward(ctx, x, delta, gamma, beta, eps, out_dtype, delta_bias, delta_b, y_buf=None, delta_from=None, drop=None):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        d2 = delta.reshape(x2.shape).contiguous()
        db2 = delta_b.reshape(x2.shape).to(delta.dtype).contiguous() if delta_b is not None else None
        ctx.two = delta_b is not None
        ctx.drop = drop
        if drop is not None:
            if delta_b is not None:
                raise NotImplementedError("dropout inside the LayerNorm kernel: one delta")
            if d2.dtype != out_dtype:
                d2 = d2.to(out_dtype)
        x_out, y, mean, rstd = _ln_fwd(x2, d2, gamma.detach(), beta.detach(), eps, out_dtype, db2, y=y_buf[0] if y_buf else None, drop=drop)
        ctx.save_for_backward(x_out, mean, rstd)
        ctx.params = (gamma, beta, delta_bias)
        ctx.delta_from = delta_from
        ctx.delta_dtype = delta.dtype
        ctx.set_materialize_grads(False)
        return x_out.view(x.shape), y.view(x.shape)

    @staticmethod
    def backward(ctx, dx_new, dy):
        x_out, mean, rstd = ctx.saved_tensors
        gamma, beta, delta_bias = ctx.params
        if delta_bias is not None and not delta_bias.requires_grad:
            delta_bias = None
        if dy is None and dx_new is None:
            return None, None, None, None, None, None, None, None, None, None, None
        if dy is None:      # only the residual output was used
            dres = dx_new.reshape(x_out.shape).contiguous().float()
            dlp = dres.to(ctx.delta_dtype)
            if ctx.drop is not None:
                p, site, row0, text_rows, per_batch = ctx.drop
                dlp = ops._dropout_rows_call(dlp.view(-1, per_batch, dlp.shape[-1]), p, site, row0, text_rows).view(dlp.shape)
            dc_ret = None
            if delta_bias is not None:
                buf, dc_ret = _grad_buffer(delta_bias)
                buf.add_(dlp.float().sum(0))
            return (dres.view(dx_new.shape), dlp.view(dx_new.shape), None, None, None, None, dc_ret,
                    (dlp.view(dx_new.shape) if ctx.two else None), None, None, None)
        dres2 = dx_new.reshape(x_out.shape).contiguous().float() if dx_new is not None else None
        dy2 = dy.reshape(x_out.shape).contiguous()
        if ctx.drop is not None and dy2.dtype != ctx.delta_dtype:
            dy2 = dy2.to(ctx.delta_dtype)
        dx, dx_lp, dg, db, dc = _ln_bwd(dy2, x_out, mean, rstd, gamma, beta, dres2, ctx.delta_dtype,
                                        delta_bias, ctx.delta_from, drop=ctx.drop)
        shape = dy.shape
        dd = (dx_lp if dx_lp is not None else dx).view(shape)
        return dx.view(shape), dd, dg, db, None, None, dc, (dd if ctx.two else None), None, None, None


class _AddLayerNormPair(torch.autograd.Function):
    """(x [+ delta [+ delta_b]]) normalised TWICE (two LayerNorms, one eps) in one pass: the parallel-residual GPT-NeoX block, whose
    attention and MLP branches each have a LayerNorm of the same residual stream.  -> (x + deltas [fp32], y_a, y_b).  Backward:
    the two LayerNorm backward kernels in a row, the second adding onto the first's result (and writing the low-precision copy
    the deltas' producers want)."""

    @staticmethod
    def forward(ctx, x, delta, delta_b, gamma_a, beta_a, gamma_b, beta_b, eps, out_dtype, bufs, delta_from=None):
        ctx.delta_from = delta_from
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        d2 = delta.reshape(x2.shape).contiguous() if delta is not None else None
        db2 = delta_b.reshape(x2.shape).to(delta.dtype).contiguous() if delta_b is not None else None
        ya, yb = bufs if bufs else (None, None)
        x_out, y, mean, rstd, y_b = _ln_fwd(x2, d2, gamma_a.detach(), beta_a.detach(), eps, out_dtype, db2, y=ya,
                                            pair=(gamma_b.detach(), beta_b.detach()), y_b=yb)
        x_sum = x_out if x_out is not None else x2
        ctx.save_for_backward(x_sum, mean, rstd)
        ctx.params = (gamma_a, beta_a, gamma_b, beta_b)
        ctx.delta_dtype = delta.dtype if delta is not None else None
        ctx.n_delta = (delta is not None) + (delta_b is not None)
        ctx.set_materialize_grads(False)
        if x_out is None:                     # no update: the caller keeps its x (an input is not handed back as an output)
            first = x2.new_empty(0)
            ctx.mark_non_differentiable(first)
        else:
            first = x_out.view(x.shape)
        return first, y.view(x.shape), y_b.view(x.shape)

    @staticmethod
    def backward(ctx, dx_new, dy_a, dy_b):
        x_sum, mean, rstd = ctx.saved_tensors
        gamma_a, beta_a, gamma_b, beta_b = ctx.params
        none = (None,) * 11
        if dx_new is None and dy_a is None and dy_b is None:
            return none
        shape = (dx_new if dx_new is not None else (dy_a if dy_a is not None else dy_b)).shape
        dres = dx_new.reshape(x_sum.shape).contiguous().float() if dx_new is not None else None
        lp = ctx.delta_dtype if ctx.n_delta else None
        dg_a = db_a = dg_b = db_b = None
        dx_lp = None
        todo = [(dy, g, b) for dy, g, b in ((dy_a, gamma_a, beta_a), (dy_b, gamma_b, beta_b)) if dy is not None]
        for i, (dy, g, b) in enumerate(todo):
            last = i == len(todo) - 1
            dres, dx_lp, dg, db, _ = _ln_bwd(dy.reshape(x_sum.shape).contiguous(), x_sum, mean, rstd, g, b, dres, lp if last else None,
                                             None, ctx.delta_from if last else None)
            if g is gamma_a:
                dg_a, db_a = dg, db
            else:
                dg_b, db_b = dg, db
        if not todo and lp not in (None, torch.float32):
            dx_lp = dres.to(lp)
        dd = (dx_lp if dx_lp is not None else dres).view(shape) if ctx.n_delta else None
        return (dres.view(shape), dd, dd if ctx.n_delta == 2 else None, dg_a, db_a, dg_b, db_b, None, None, None, None)


def dy_buffer_of(weight, device):
    """-> a callable (rows, features, dtype) -> buffer-or-None for the producer of the output gradient of `weight`'s latest
    projection (its slab of the dy stack), or None when there is nothing to aim at."""
    target = _pass_of(weight, device)
    if target is None:
        return None
    return lambda rows, n, dtype: _deferred_dw_for(device).dy_slab(target[0], target[1], rows, n, dtype)


def add_layer_norm_pair(x, delta, ln_a, ln_b, out_dtype, delta_b=None, out_for=(None, None), delta_from=None):
    """-> (x + delta + delta_b [fp32; x itself without deltas], ln_a of it, ln_b of it) in one kernel.  Falls back to two calls when
    the two LayerNorms differ in eps."""
    if ln_a.eps != ln_b.eps:
        if delta is None:
            return x, layer_norm(x, ln_a, out_dtype, out_for=out_for[0]), layer_norm(x, ln_b, out_dtype, out_for=out_for[1])
        x1, ya = add_layer_norm(x, delta, ln_a, out_dtype, delta_b=delta_b, out_for=out_for[0])
        return x1, ya, layer_norm(x1, ln_b, out_dtype, out_for=out_for[1])
    rows, C_ = x.numel() // x.shape[-1], x.shape[-1]
    bufs = (_x_slab_for(out_for[0], rows, C_, out_dtype, x.device), _x_slab_for(out_for[1], rows, C_, out_dtype, x.device))
    x1, ya, yb = _AddLayerNormPair.apply(x, delta, delta_b, ln_a.weight, ln_a.bias, ln_b.weight, ln_b.bias, ln_a.eps, out_dtype,
                                         bufs if (bufs[0] is not None or bufs[1] is not None) else None,
                                         _pass_of(delta_from, x.device) if delta is not None else None)
    return (x if delta is None else x1), ya, yb


def _x_slab_for(weight, rows, n, dtype, device):
    """Stacked-operand slab (see _DeferredWeightGrads) for the producer of the next input of `weight`'s projection, or None."""
    if weight is None or not (STACKED_ACTIVATIONS and BATCH_DW and torch.is_grad_enabled() and weight.requires_grad):
        return None
    return _deferred_dw_for(device).x_slab(weight, rows, n, dtype)


def _pass_of(weight, device):
    """(weight, index of its projection's latest forward pass) for the producer of that projection's output gradient, or None."""
    if weight is None or not (STACKED_ACTIVATIONS and BATCH_DW and torch.is_grad_enabled() and weight.requires_grad):
        return None
    k = _deferred_dw_for(device).last_pass(weight)
    return (weight, k) if k is not None else None


def layer_norm(x, ln, out_dtype, out_for=None):
    """out_for: the weight of the projection that reads the result (its stacked dW operand is written in place)."""
    y_buf = _x_slab_for(out_for, x.numel() // x.shape[-1], x.shape[-1], out_dtype, x.device)
    return _LayerNorm.apply(x, ln.weight, ln.bias, ln.eps, out_dtype, (y_buf,) if y_buf is not None else None)


def add_layer_norm(x, delta, ln, out_dtype, delta_bias=None, delta_b=None, out_for=None, delta_from=None, drop=None):
    """-> (x + delta [+ delta_b] [fp32], LayerNorm of it [out_dtype]).  delta_bias, delta_b: see _AddLayerNorm.  out_for: as in
    layer_norm; delta_from: the weight of the projection that produced delta (its output gradient is written into its stack).
    drop = (p, site, row0, text_rows): delta enters through a dropout, x + dropout(delta), inside the kernels (x is (B, T, C))."""
    y_buf = _x_slab_for(out_for, x.numel() // x.shape[-1], x.shape[-1], out_dtype, x.device)
    if drop is not None:
        drop = (*drop[:4], x.shape[-2])
    return _AddLayerNorm.apply(x, delta, ln.weight, ln.bias, ln.eps, out_dtype, delta_bias, delta_b,
                               (y_buf,) if y_buf is not None else None, _pass_of(delta_from, x.device), drop)


# ---------------------------------------------------------------------------------------------------------------------
# Token + position embedding
# ---------------------------------------------------------------------------------------------------------------------
class _Embedding(torch.autograd.Function):
    """wte[ids] (+ wpe[pos]) in one launch; backward adds into the tables' fp32 gradients deterministically."""

    @staticmethod
    def forward(ctx, ids, pos, wte, wpe):
        rows, C_ = ids.numel(), wte.shape[1]
        out = torch.empty(ids.shape + (C_,), dtype=torch.float32, device=wte.device)
        with ops._timed("embed_fwd"):
            C.check(C.lib().bg_embed_fwd(C.ptr(ids), rows, C.ptr(pos), pos.numel() if pos is not None else 0, C.ptr(wte.detach()),
                                         wte.shape[0], C.ptr(wpe.detach()) if wpe is not None else None,
                                         wpe.shape[0] if wpe is not None else 0, C_, C.ptr(out), C.stream_ptr()), "bg_embed_fwd")
        ctx.save_for_backward(ids, pos)
        ctx.params = (wte, wpe)
        return out

    @staticmethod
    def backward(ctx, dy):
        ids, pos = ctx.saved_tensors
        rows = ids.numel()
        dy2 = dy.reshape(rows, -1).contiguous().float()
        lib = C.lib()
        ws = _workspace(lib.bg_embed_bwd_workspace_bytes(rows), dy2.device)
        rets = []
        with ops._timed("embed_bwd"):
            for table, idx in zip(ctx.params, (ids, pos)):
                if table is None or not table.requires_grad:
                    rets.append(None)
                    continue
                buf, ret = _grad_buffer(table)
                C.check(lib.bg_embed_bwd(C.ptr(dy2), C.ptr(idx), rows, idx.numel(), table.shape[0], dy2.shape[1], C.ptr(buf),
                                         C.ptr(ws), ws.numel(), C.stream_ptr()), "bg_embed_bwd")
                rets.append(ret)
        return None, None, rets[0], rets[1]


def embedding(input_ids, pos, wte, wpe=None):
    """input_ids (B, T) integer; pos (B, T) or (1, T) integer, possibly an expanded view (None with wpe=None: the token table
    alone, GPT-NeoX) -> fp32 (B, T, C)."""
    ids = input_ids.contiguous().long()
    if wpe is None:
        return _Embedding.apply(ids, None, wte, None)
    if pos.dim() == 2 and pos.shape[0] > 1 and pos.stride(0) == 0:
        pos = pos[0]                                   # one row of positions shared by the batch: rows index it modulo T
    pos = pos.contiguous().long().view(-1)
    if ids.numel() % pos.numel() != 0:
        raise ValueError("embedding: positions do not tile the ids")
    return _Embedding.apply(ids, pos, wte, wpe)


# ---------------------------------------------------------------------------------------------------------------------
# Linear (+ GELU) with in-place gradient accumulation
# ---------------------------------------------------------------------------------------------------------------------
_ACT = {None: 0, "gelu_tanh": 1, "gelu": 2}
_mm_out_dtype = None       # does torch.addmm(..., out_dtype=float32, out=) accumulate a bf16 GEMM into fp32 here?  probed once


def _probe_mm_out_dtype(device):
    global _mm_out_dtype
    if _mm_out_dtype is None:
        try:
            a = torch.ones((16, 16), dtype=torch.bfloat16, device=device)
            g = torch.ones((16, 16), dtype=torch.float32, device=device)
            torch.addmm(g, a, a, out_dtype=torch.float32, out=g)
            _mm_out_dtype = bool(float(g[0, 0]) == 17.0)
        except Exception:
            _mm_out_dtype = False
    return _mm_out_dtype


def _accumulate_matmul(p, a, b):
    """p.grad += a @ b (a, b in the compute dtype; the gradient is fp32): ONE GEMM with beta = 1 whose fp32 accumulator is
    written out unrounded - no bf16 dW, no cast, no add."""
    buf, ret = _grad_buffer(p)
    if a.dtype == torch.float32:
        buf.addmm_(a, b)
    elif _mm_out_dtype or (_mm_out_dtype is None and not torch.cuda.is_current_stream_capturing() and _probe_mm_out_dtype(a.device)):
        torch.addmm(buf, a, b, out_dtype=torch.float32, out=buf)
    else:
        buf.add_(torch.mm(a, b))
    return ret


# Weight gradients, batched.  A transformer's backward pass issues one dW GEMM per weight and pass - at BASELINE configs[1]
# 768 x 768 .. 768 x 3072 outputs over 2048 rows: 36 - 144 tiles for 256 CUs, ~20 us each, 3.9 ms of a 19 ms step
# (profiles/r1_n_bench_c2_by_class.txt).  Nothing inside the pass reads a weight gradient, so the GEMMs wait: when the pass ends
# (the autograd-engine callback the column reductions use) the waiting products are grouped - same shapes, same layout, gradient
# buffers at one constant stride (the layers of one weight kind in the flat arena) - and each group runs as ONE strided-batched
# GEMM straight into the arena, the passes that share a weight (root and tree pass) concatenated along the reduction dimension.
# What cannot be grouped runs as before, just later.  BRANCHGAN_BATCH_DW=0: every dW GEMM where it arises.
BATCH_DW = os.environ.get("BRANCHGAN_BATCH_DW", "1") != "0"


# The operands of those batched GEMMs are [layers][rows of every pass][features] stacks.  The first pass over a model copies the
# layers' saved activations and output gradients into them (bg_stack_rows: 1.1 ms a step at configs[1], 3.6 GB moved); the
# stacks then stay (per group of weights, `_Stack`), and from the next step on the PRODUCERS write there in the first place:
# the LayerNorm / attention / GELU kernels get the (layer, pass) slab of the x stack as their output buffer (`out_for=` the
# weight of the consuming projection), the LayerNorm / attention / GELU backward kernels the slab of the dy stack (`delta_from=`,
# `qkv_from=`).  flush() copies only what is not at its place.  A slab is handed out again only after the backward pass that read
# it has flushed (or its autograd node is gone); a saved activation overwritten anyway (retain_graph + another forward) raises in
# backward.  BRANCHGAN_STACKED_ACTIVATIONS=0: transient stacks, everything copied, as in the first step.
STACKED_ACTIVATIONS = os.environ.get("BRANCHGAN_STACKED_ACTIVATIONS", "1") != "0"
GELU_EPILOGUE = os.environ.get("BRANCHGAN_GELU_EPILOGUE", "1") != "0"      # no-grad passes: bias + tanh-GELU in the GEMM's epilogue
STACK_BYTES_CAP = int(float(os.environ.get("BRANCHGAN_STACK_GIB", "48")) * (1 << 30))
stack_stats = {"copied": 0, "in_place": 0}


class _Stack:
    def __init__(self, weights, rows, n_in, n_out, dtype, device):
        import weakref
        self.weights = [weakref.ref(w) for w in weights]
        self.rows, self.dtype = tuple(rows), dtype
        self.off = [sum(rows[:i]) for i in range(len(rows))]
        L, R = len(weights), sum(rows)
        self.xa = torch.empty((L, R, n_in), dtype=dtype, device=device)
        self.da = torch.empty((L, R, n_out), dtype=dtype, device=device)
        self.pinned = torch.cuda.is_current_stream_capturing()      # a recorded graph holds its addresses: never dropped
        self.tick = 0

    def nbytes(self):
        return (self.xa.numel() + self.da.numel()) * self.xa.element_size()

    def slab(self, which, l, k, rows, n, dtype):
        buf = self.xa if which == "x" else self.da
        if k >= len(self.rows) or self.rows[k] != rows or buf.shape[2] != n or dtype != self.dtype:
            return None
        return buf[l, self.off[k]:self.off[k] + rows].data      # .data: a version counter of its own (the slabs of a stack are
                                                                # written at different times; autograd checks saved tensors by version)


class _SlabToken:
    """Lives in the autograd node of the projection that read a slab; when the node goes without a backward pass the slab is free
    again."""
    def __init__(self, owner, wid, k, gen):
        self.owner, self.wid, self.k, self.gen = owner, wid, k, gen

    def __del__(self):
        try:
            self.owner.release(self.wid, self.k, self.gen)
        except Exception:
            pass


class _DeferredWeightGrads:
    def __init__(self, device):
        self.device = device
        self.pending, self.armed, self.task = [], False, -1
        self.streams = set()
        self.stacks, self.where, self.tick = {}, {}, 0      # group key -> _Stack; id(weight) -> (_Stack, layer index)
        self.next_k, self.busy, self.gen = {}, {}, {}       # per weight: forward passes since its last flush; slabs being read

    def begin(self):
        task = torch._C._current_graph_task_id()
        if task < 0:
            return False
        if self.armed and self.task == task:
            return True
        if self.pending:
            self.flush()
        try:
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)
        except RuntimeError:
            self.armed = False
            return False
        self.armed, self.task = True, task
        return True

    # ---- slabs -----------------------------------------------------------------------------------------------------------
    def _slab(self, weight, which, k, rows, n, dtype):
        ent = self.where.get(id(weight))
        if ent is None or ent[0].weights[ent[1]]() is not weight:
            return None
        st, l = ent
        t = st.slab(which, l, k, rows, n, dtype)
        if t is not None:
            self.tick += 1
            st.tick = self.tick
            st.pinned = st.pinned or torch.cuda.is_current_stream_capturing()
        return t

    def x_slab(self, weight, rows, n, dtype):
        """Output buffer for the producer of the NEXT forward input of `weight`'s projection (None: allocate as usual)."""
        k = self.next_k.get(id(weight), 0)
        if k in self.busy.get(id(weight), ()):
            return None
        return self._slab(weight, "x", k, rows, n, dtype)

    def dy_slab(self, weight, k, rows, n, dtype):
        return self._slab(weight, "dy", k, rows, n, dtype) if k is not None else None

    def forward_pass(self, weight, x2):
        """The projection of `weight` runs forward on x2 (grad mode): -> (pass index, token or None)."""
        wid = id(weight)
        k = self.next_k.get(wid, 0)
        self.next_k[wid] = k + 1
        t = self._slab(weight, "x", k, x2.shape[0], x2.shape[1], x2.dtype) if k not in self.busy.get(wid, ()) else None
        if t is None or t.data_ptr() != x2.data_ptr() or x2.stride() != t.stride():
            return k, None
        self.busy.setdefault(wid, set()).add(k)
        g = self.gen[(wid, k)] = self.gen.get((wid, k), 0) + 1
        return k, _SlabToken(self, wid, k, g)

    def last_pass(self, weight):
        k = self.next_k.get(id(weight), 0) - 1
        return k if k >= 0 else None

    def current(self, token):
        return self.gen.get((token.wid, token.k)) == token.gen

    def release(self, wid, k, gen=None):
        if gen is not None and self.gen.get((wid, k)) != gen:
            return
        b = self.busy.get(wid)
        if b is not None and k in b:
            b.discard(k)
            if not b:
                self.next_k[wid] = 0

    def _stack_for(self, key, members, rows, n_in, n_out, dtype):
        weights = [m[3] for m in members]
        key = key + tuple(id(w) for w in weights)
        st = self.stacks.get(key)
        # slabs still being read by a forward pass that is not part of this flush: leave every kept stack alone
        held = any(self.busy.get(id(w), set()) - ks for w, ks in zip(weights, (m[4] for m in members)))
        if st is not None and all(r() is w for r, w in zip(st.weights, weights)) and not held:
            for l, w in enumerate(weights):
                self.where[id(w)] = (st, l)
            return st
        st = _Stack(weights, rows, n_in, n_out, dtype, self.device)
        if not STACKED_ACTIVATIONS or held:
            return st
        total = st.nbytes() + sum(o.nbytes() for o in self.stacks.values())
        for okey, o in sorted(self.stacks.items(), key=lambda kv: kv[1].tick):
            if total <= STACK_BYTES_CAP:
                break
            if not o.pinned:
                total -= o.nbytes()
                del self.stacks[okey]
                for wid in [wid for wid, ent in self.where.items() if ent[0] is o]:
                    del self.where[wid]
        if total > STACK_BYTES_CAP:
            return st                                   # transient: nobody learns where it is
        self.stacks[key] = st
        for l, w in enumerate(weights):
            self.where[id(w)] = (st, l)
        return st

    # ---- the batched GEMMs ---------------------------------------------------------------------------------------------
    def add(self, weight, buf, layout, x2, dy2, k=None):
        self.pending.append((weight, buf, layout, x2, dy2, k))
        self.streams.add(torch.cuda.current_stream(self.device))

    def flush(self):
        self.armed = False
        pending, self.pending = self.pending, []
        streams, self.streams = self.streams, set()
        if not pending:
            return
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            for st in streams:
                if st != cur:
                    cur.wait_stream(st)
            per_weight = {}
            for weight, buf, layout, x2, dy2, k in pending:
                passes = per_weight.setdefault(id(weight), (buf, layout, [], weight))[2]
                passes.append((len(passes) if k is None else k, x2, dy2))
            groups = {}
            for buf, layout, passes, weight in per_weight.values():
                passes.sort(key=lambda t: t[0])               # forward order: the order the stacks are laid out in
                xs, dys = [t[1] for t in passes], [t[2] for t in passes]
                key = (layout, xs[0].dtype, tuple(buf.shape), tuple(x.shape[0] for x in xs), xs[0].shape[1], dys[0].shape[1])
                groups.setdefault(key, []).append((buf, xs, dys, weight, {t[0] for t in passes}))
            for key, members in groups.items():
                layout, dtype, wshape, rows, n_in, n_out = key
                members.sort(key=lambda m: m[0].data_ptr())
                stride = None
                if len(members) > 1 and dtype != torch.float32 and _mm_out_dtype:
                    d = [(b[0].data_ptr() - members[0][0].data_ptr()) for b in members]
                    step = d[1]
                    if step % 4 == 0 and step // 4 >= members[0][0].numel() and all(x == i * step for i, x in enumerate(d)) \
                            and all(m[0].is_contiguous() for m in members) \
                            and all(m[0].untyped_storage().data_ptr() == members[0][0].untyped_storage().data_ptr() for m in members):
                        stride = step // 4
                if stride is None:
                    for buf, xs, dys, _, _ in members:
                        for x2, dy2 in zip(xs, dys):
                            _accumulate_into(buf, x2.t(), dy2) if layout == "io" else _accumulate_into(buf, dy2.t(), x2)
                    continue
                L, R = len(members), sum(rows)
                st = self._stack_for(key, members, rows, n_in, n_out, dtype)
                xa, da = st.xa, st.da
                _stack_rows([x for _, xs, _, _, _ in members for x in xs], xa.view(L * R, n_in),
                            [d_ for _, _, dys, _, _ in members for d_ in dys], da.view(L * R, n_out))
                out = members[0][0].as_strided((L,) + wshape, (stride, wshape[1], 1))
                if layout == "io":
                    torch.baddbmm(out, xa.transpose(1, 2), da, out_dtype=torch.float32, out=out)
                else:
                    torch.baddbmm(out, da.transpose(1, 2), xa, out_dtype=torch.float32, out=out)
            for weight, _, _, _, _, k in pending:
                if k is not None:
                    self.release(id(weight), k)
            for wid in {id(p_[0]) for p_ in pending}:
                if not self.busy.get(wid):
                    self.next_k[wid] = 0


def _stack_rows(xs, xa, dys, da):
    """xa = the xs one under the other, da = the dys: ONE bg_stack_rows launch (torch.cat when a source is not 16-byte regular)."""
    descs, ok = [], True
    for srcs, dst in ((xs, xa), (dys, da)):
        row_bytes, off = dst.shape[1] * dst.element_size(), 0
        for t in srcs:
            es = t.element_size()
            if t.stride(1) != 1 or row_bytes % 16 or (t.stride(0) * es) % 16 or t.data_ptr() % 16 or (dst.data_ptr() + off) % 16:
                ok = False
            if t.data_ptr() == dst.data_ptr() + off and t.stride(1) == 1 and t.stride(0) * es == row_bytes:
                stack_stats["in_place"] += 1            # its producer wrote it here (a slab of a _Stack)
            else:
                stack_stats["copied"] += 1
                descs.append((t.data_ptr(), dst.data_ptr() + off, t.shape[0], row_bytes, t.stride(0) * es))
            off += t.shape[0] * row_bytes
    if not ok:
        # (sources that are slabs of the destination are copied onto themselves row by row: cat handles the overlap-free case only)
        off = 0
        for srcs, dst in ((xs, xa), (dys, da)):
            off = 0
            for t in srcs:
                part = dst[off:off + t.shape[0]]
                if part.data_ptr() != t.data_ptr() or part.stride() != t.stride():
                    part.copy_(t)
                off += t.shape[0]
        return
    if not descs:
        return
    arr = (C.StackDesc * len(descs))()
    for d, (src, dst, rows, rb, st) in zip(arr, descs):
        d.src, d.dst, d.rows, d.row_bytes, d.src_row_stride = src, dst, rows, rb, st
    with ops._timed("stack_rows"):
        C.check(C.lib().bg_stack_rows(arr, len(descs), C.stream_ptr()), "bg_stack_rows")


_deferred_dw = {}


def _deferred_dw_for(device):
    d = _deferred_dw.get(device)
    if d is None:
        d = _deferred_dw[device] = _DeferredWeightGrads(device)
    return d


def _accumulate_into(buf, a, b):
    if a.dtype == torch.float32:
        buf.addmm_(a, b)
    elif _mm_out_dtype or (_mm_out_dtype is None and not torch.cuda.is_current_stream_capturing() and _probe_mm_out_dtype(a.device)):
        torch.addmm(buf, a, b, out_dtype=torch.float32, out=buf)
    else:
        buf.add_(torch.mm(a, b))


class _Linear(torch.autograd.Function):
    """y = act(x @ W + b).  layout 'io': W is (in, out) (GPT-2 Conv1D); 'oi': (out, in) (nn.Linear)."""

    @staticmethod
    def forward(ctx, x, weight, bias, layout, act, dtype, bias_grad, out_for=None):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.dtype != dtype:
            x2 = x2.to(dtype)
        w = lowp(weight, dtype)
        wm = w if layout == "io" else w.t()
        pre = torch.addmm(lowp(bias, dtype), x2, wm) if bias is not None else torch.mm(x2, wm)
        ctx.pass_k = ctx.slab = None
        if out_for is not None:             # grad mode (linear() checked): this pass's place in the stacked dW operands
            ctx.pass_k, ctx.slab = _deferred_dw_for(x2.device).forward_pass(weight, x2)
        if act is None:
            y = pre
        else:
            y = out_for[0] if out_for else None         # the activation is the next projection's input: into its x stack
            if y is not None:
                torch.ops.aten.gelu.out(pre, approximate="tanh" if act == "gelu_tanh" else "none", out=y)
            else:
                y = F.gelu(pre, approximate="tanh" if act == "gelu_tanh" else "none")
        ctx.save_for_backward(x2, pre if act is not None else None)
        ctx.meta = (weight, bias if bias_grad else None, layout, act, dtype, x.shape)
        return y.view(*x.shape[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        x2, pre = ctx.saved_tensors
        weight, bias, layout, act, dtype, in_shape = ctx.meta
        if ctx.slab is not None and not _deferred_dw_for(x2.device).current(ctx.slab):
            raise RuntimeError("branchgan.fused: a saved activation in the stacked dW operands was overwritten by a later forward "
                               "pass before this backward ran (retain_graph with another forward in between?); "
                               "set BRANCHGAN_STACKED_ACTIVATIONS=0")
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.dtype != dtype:
            dy2 = dy2.to(dtype)
        if dy2.stride(-1) != 1 or dy2.stride(0) % 4 != 0:
            dy2 = dy2.contiguous()
        rows, n_out = dy2.shape
        db_ret = None
        if bias is not None or act is not None:
            lib = C.lib()
            if act is not None:
                dy2 = dy2.contiguous()
                dpre = _deferred_dw_for(dy2.device).dy_slab(weight, ctx.pass_k, rows, n_out, pre.dtype) if BATCH_DW else None
                if dpre is None:
                    dpre = torch.empty_like(pre)
            else:
                dpre = None
            db = None                       # no bias (or not ours to differentiate): the partial sums are simply never reduced
            if bias is not None:
                db, db_ret = _grad_buffer(bias)
            nbytes = lib.bg_bias_act_bwd_workspace_bytes(rows, n_out)
            later = _deferred_for(dy2.device) if DEFER_REDUCTIONS and db is not None and db_ret is None else None
            if later is not None and not later.begin():
                later = None
            ws = later.alloc(nbytes) if later is not None else _workspace(nbytes, dy2.device)
            with ops._timed("bias_act_bwd"):
                C.check(lib.bg_bias_act_bwd(C.ptr(dy2), C.dtype_code(dy2), rows, n_out, dy2.stride(0), _ACT[act], C.ptr(pre),
                                            C.ptr(dpre), None if later is not None else C.ptr(db), C.ptr(ws), ws.numel(), C.stream_ptr()),
                        "bg_bias_act_bwd")
            if later is not None:
                later.add(ws.data_ptr(), db, lib.bg_bias_act_bwd_groups(rows), n_out, n_out)
            if dpre is not None:
                dy2 = dpre
        w = lowp(weight, dtype)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.mm(dy2, w.t() if layout == "io" else w).view(in_shape)
        dw_ret = None
        later = None
        if BATCH_DW and weight.requires_grad:
            buf, ret = _grad_buffer(weight)
            if ret is None:                              # the gradient lives in p.grad (the arena): the GEMM can wait
                later = _deferred_dw_for(dy2.device)
                if later.begin():
                    later.add(weight, buf, layout, x2, dy2, ctx.pass_k)
                else:
                    later = None
            if later is None:
                _accumulate_into(buf, x2.t(), dy2) if layout == "io" else _accumulate_into(buf, dy2.t(), x2)
                dw_ret = ret
        else:
            dw_ret = _accumulate_matmul(weight, x2.t(), dy2) if layout == "io" else _accumulate_matmul(weight, dy2.t(), x2)
        return dx, dw_ret, db_ret, None, None, None, None, None


def linear(x, weight, bias, layout, dtype, act=None, bias_grad=True, out_for=None):
    """bias_grad=False: the bias is added in forward but its gradient is somebody else's job (add_layer_norm(delta_bias=)).
    out_for (with act): the weight of the projection that reads the activation (see layer_norm)."""
    if GELU_EPILOGUE and act == "gelu_tanh" and bias is not None and not torch.is_grad_enabled():
        # Passes nobody differentiates (the sampling phase: root pass + branch steps; decoding): bias and tanh-GELU ride in the
        # GEMM's epilogue (ONE hipBLASLt launch, the activation applied to the fp32 accumulator) - the pre-activation a backward
        # pass would need is never written.  benchmarks/probes/addmm_gelu_probe.py: 11.1 -> 8.1 us at 256 rows, 25.1 -> 19.0 at 2048.
        x2 = x.reshape(-1, x.shape[-1])
        if x2.dtype != dtype:
            x2 = x2.to(dtype)
        w = lowp(weight, dtype)
        y = torch._addmm_activation(lowp(bias, dtype), x2, w if layout == "io" else w.t(), use_gelu=True)
        return y.view(*x.shape[:-1], y.shape[-1])
    track = None
    if STACKED_ACTIVATIONS and BATCH_DW and torch.is_grad_enabled() and weight.requires_grad:
        y_buf = None
        if act is not None and out_for is not None:
            n_out = weight.shape[1] if layout == "io" else weight.shape[0]
            y_buf = _x_slab_for(out_for, x.numel() // x.shape[-1], n_out, dtype, x.device)
        track = (y_buf,) if y_buf is not None else ()
    return _Linear.apply(x, weight, bias, layout, act, dtype, bias_grad, track)


# ---------------------------------------------------------------------------------------------------------------------
# attention on the packed QKV projection
# ---------------------------------------------------------------------------------------------------------------------
def _attn_fwd_call(q5, kt, vt, slots_k, slots_v, ctx_len, paths, scale, o5, lse, drop=None):
    path, path_len, max_path = ops.path_tables(paths, q5.device)
    sh = ops._attn_shape(q5, kt, slots_k, scale, max_path, o5, drop)
    ops.attn_fwd_launch(sh, q5, kt, vt, slots_k, slots_v, ctx_len, path, path_len, o5, lse)


def _attn_bwd_call(q5, kt, vt, slots_k, slots_v, ctx_len, paths, scale, o5, lse, do5, dq5, dkt, dvt, dsk, dsv, dkt_in=None, dvt_in=None,
                   drop=None):
    path, path_len, max_path = ops.path_tables(paths, q5.device)
    sh = ops._attn_shape(q5, kt, slots_k, scale, max_path, o5, drop)
    sh.dq_sb, sh.dq_sh, sh.dq_sn, sh.dq_sk = dq5.stride()[:4]
    delta = torch.empty_like(lse)
    ops.attn_bwd_launch(sh, q5, kt, vt, slots_k, slots_v, ctx_len, path, path_len, o5, lse, do5, dq5, dkt, dvt, dkt_in, dvt_in, dsk, dsv, delta)


def _heads_first(t, NQ, K):
    """(B, NQ*K, H, Dh) slice of a packed buffer -> (B, H, NQ, K, Dh) strided view."""
    return t.unflatten(1, (NQ, K)).permute(0, 3, 1, 2, 4)


class _MergedTreeAttention(torch.autograd.Function):
    """Root text and every tree node in ONE sequence: qkv (B, S + N*K, 3, H, Dh).  Rows [0, S) attend causally among
    themselves; rows of node n, branch point k attend to text rows j < ctx_len[k] of THE SAME qkv, their ancestors' column k
    and themselves.  Two kernel calls on slices of the packed buffer each way; everything else of the layer (GEMMs,
    LayerNorms) sees one batch of S + N*K rows.  Returns (att (B, T, E), k view, v view)."""

    @staticmethod
    def forward(ctx, qkv, spec):
        B, T, _, H, Dh = qkv.shape
        S, N = spec["n_text"], spec["n_query_nodes"]
        K = (T - S) // N
        att = spec["att_buf"].view(B, T, H, Dh) if spec.get("att_buf") is not None else \
            torch.empty((B, T, H, Dh), dtype=qkv.dtype, device=qkv.device)
        lse_r = torch.empty((B, H, 1, S), dtype=torch.float32, device=qkv.device)
        lse_t = torch.empty((B, H, N, K), dtype=torch.float32, device=qkv.device)
        kview, vview = qkv[:, :, 1].permute(0, 2, 1, 3), qkv[:, :, 2].permute(0, 2, 1, 3)
        kt, vt = kview[:, :, :S], vview[:, :, :S]
        drops = (ops.drop_descriptor(spec.get("p_drop"), spec.get("drop_site"), 0),                  # root rows, tree rows
                 ops.drop_descriptor(spec.get("p_drop"), spec.get("drop_site"), spec.get("drop_row0", 0)))
        _attn_fwd_call(_heads_first(qkv[:, :S, 0], 1, S), kt, vt, [], [], ops.causal_ctx_len(S, S, qkv.device), [[]], spec["scale"],
                       _heads_first(att[:, :S], 1, S), lse_r, drops[0])
        own = lambda t: [t[:, :, S + n * K:S + (n + 1) * K] for n in range(N)]
        _attn_fwd_call(_heads_first(qkv[:, S:, 0], N, K), kt, vt, own(kview), own(vview), spec["ctx_len"], spec["paths"], spec["scale"],
                       _heads_first(att[:, S:], N, K), lse_t, drops[1])
        ctx.save_for_backward(qkv, att, lse_r, lse_t, spec["ctx_len"])
        ctx.meta = (S, N, K, spec["scale"], spec["paths"], drops)
        ctx.qkv_from = spec.get("qkv_from")
        ctx.set_materialize_grads(False)
        return att.view(B, T, H * Dh), kview, vview

    @staticmethod
    def backward(ctx, datt, dk_out, dv_out):
        qkv, att, lse_r, lse_t, ctx_len = ctx.saved_tensors
        S, N, K, scale, paths, drops = ctx.meta
        B, T, _, H, Dh = qkv.shape
        dqkv = _dqkv_buffer(qkv, ctx.qkv_from)
        if datt is None:
            datt = torch.zeros((B, T, H, Dh), dtype=qkv.dtype, device=qkv.device)
        do = datt.reshape(B, T, H, Dh).to(qkv.dtype).contiguous()
        kview, vview = qkv[:, :, 1].permute(0, 2, 1, 3), qkv[:, :, 2].permute(0, 2, 1, 3)
        dkview, dvview = dqkv[:, :, 1].permute(0, 2, 1, 3), dqkv[:, :, 2].permute(0, 2, 1, 3)
        kt, vt = kview[:, :, :S], vview[:, :, :S]
        _attn_bwd_call(_heads_first(qkv[:, :S, 0], 1, S), kt, vt, [], [], ops.causal_ctx_len(S, S, qkv.device), [[]], scale,
                       _heads_first(att[:, :S], 1, S), lse_r, _heads_first(do[:, :S], 1, S), _heads_first(dqkv[:, :S, 0], 1, S),
                       dkview[:, :, :S], dvview[:, :, :S], [], [], drop=drops[0])
        own = lambda t: [t[:, :, S + n * K:S + (n + 1) * K] for n in range(N)]
        _attn_bwd_call(_heads_first(qkv[:, S:, 0], N, K), kt, vt, own(kview), own(vview), ctx_len, paths, scale,
                       _heads_first(att[:, S:], N, K), lse_t, _heads_first(do[:, S:], N, K), _heads_first(dqkv[:, S:, 0], N, K),
                       dkview[:, :, :S], dvview[:, :, :S], own(dkview), own(dvview),
                       dkt_in=dkview[:, :, :S], dvt_in=dvview[:, :, :S],     # the tree rows' share of the text dK / dV: added in the kernel
                       drop=drops[1])
        if dk_out is not None:
            dkview.add_(dk_out)
        if dv_out is not None:
            dvview.add_(dv_out)
        return dqkv, None


class _PackedAttention(torch.autograd.Function):
    """qkv (B, T, 3, H, Dh) contiguous, T = NQ * K.  lp = None: root causal pass.  Otherwise lp carries the text KV
    (k_text, v_text as inputs so that their gradients flow), ctx_len and paths; ancestors' (k, v) follow as inputs.
    Returns (att (B, T, H*Dh), k (B,H,T,Dh) view, v view)."""

    @staticmethod
    def forward(ctx, qkv, k_text, v_text, spec, *anc):
        B, T, _, H, Dh = qkv.shape
        E = H * Dh
        n_anc = len(anc) // 2
        anc_k, anc_v = list(anc[:n_anc]), list(anc[n_anc:])
        root = k_text is None
        NQ = 1 if root else spec["n_query_nodes"]
        K = T // NQ
        att = spec["att_buf"].view(B, T, H, Dh) if spec.get("att_buf") is not None else \
            torch.empty((B, T, H, Dh), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((B, H, NQ, K), dtype=torch.float32, device=qkv.device)
        q5 = qkv[:, :, 0].unflatten(1, (NQ, K)).permute(0, 3, 1, 2, 4)          # (B,H,NQ,K,Dh) strided view
        o5 = att.unflatten(1, (NQ, K)).permute(0, 3, 1, 2, 4)
        kview = qkv[:, :, 1].permute(0, 2, 1, 3)                                # (B,H,T,Dh)
        vview = qkv[:, :, 2].permute(0, 2, 1, 3)
        if root:
            kt, vt = kview, vview
            ctx_len = ops.causal_ctx_len(T, T, qkv.device)
            slots_k, slots_v, paths = [], [], [[]]
        else:
            kt, vt, ctx_len, paths = k_text, v_text, spec["ctx_len"], spec["paths"]
            if kt.stride() != vt.stride() or kt.stride(-1) != 1:
                kt, vt = kt.contiguous(), vt.contiguous()
            own_k = [kview[:, :, n * K:(n + 1) * K] for n in range(NQ)]
            own_v = [vview[:, :, n * K:(n + 1) * K] for n in range(NQ)]
            ref = own_k[0].stride()
            if any(t.stride() != ref for t in anc_k + anc_v):
                raise NotImplementedError("ancestor KV slots must share the packed-QKV layout of the new tokens")
            slots_k, slots_v = anc_k + own_k, anc_v + own_v
        path, path_len, max_path = ops.path_tables(paths, qkv.device)
        drop = ops.drop_descriptor(spec.get("p_drop"), spec.get("drop_site"), 0 if root else spec.get("drop_row0", 0))
        sh = ops._attn_shape(q5, kt, slots_k, spec["scale"], max_path, o5, drop)
        ops.attn_fwd_launch(sh, q5, kt, vt, slots_k, slots_v, ctx_len, path, path_len, o5, lse)
        ctx.save_for_backward(qkv, att, lse, ctx_len, path, path_len, *([] if root else [kt, vt]), *anc_k, *anc_v)
        ctx.meta = (root, NQ, K, n_anc, spec["scale"], max_path, drop)
        ctx.qkv_from = spec.get("qkv_from")
        ctx.set_materialize_grads(False)
        return att.view(B, T, E), kview, vview

    @staticmethod
    def backward(ctx, datt, dk_out, dv_out):
        root, NQ, K, n_anc, scale, max_path, drop = ctx.meta
        saved = ctx.saved_tensors
        qkv, att, lse, ctx_len, path, path_len = saved[:6]
        rest = list(saved[6:])
        B, T, _, H, Dh = qkv.shape
        dev, dt = qkv.device, qkv.dtype
        dqkv = _dqkv_buffer(qkv, ctx.qkv_from)
        q5 = qkv[:, :, 0].unflatten(1, (NQ, K)).permute(0, 3, 1, 2, 4)
        o5 = att.unflatten(1, (NQ, K)).permute(0, 3, 1, 2, 4)
        dq5 = dqkv[:, :, 0].unflatten(1, (NQ, K)).permute(0, 3, 1, 2, 4)
        kview, vview = qkv[:, :, 1].permute(0, 2, 1, 3), qkv[:, :, 2].permute(0, 2, 1, 3)
        dkview, dvview = dqkv[:, :, 1].permute(0, 2, 1, 3), dqkv[:, :, 2].permute(0, 2, 1, 3)
        if datt is None:
            datt = torch.zeros((B, T, H, Dh), dtype=dt, device=dev)
        do5 = datt.reshape(B, T, H, Dh)
        if do5.dtype != dt:
            do5 = do5.to(dt)
        if not do5.is_contiguous():
            do5 = do5.contiguous()
        do5 = do5.unflatten(1, (NQ, K)).permute(0, 3, 1, 2, 4)
        d_anc_k, d_anc_v = [], []
        dk_in = dv_in = None
        if root:
            kt, vt = kview, vview
            dkt, dvt = dkview, dvview
            slots_k, slots_v, dsk, dsv = [], [], [], []
            # what the tree pass sent back to the text keys / values: same layout -> added inside the dkv kernel
            if dk_out is not None and dv_out is not None and dk_out.dtype == dt and dv_out.dtype == dt \
                    and dk_out.stride() == dkt.stride() and dv_out.stride() == dvt.stride() \
                    and dk_out.data_ptr() % 16 == 0 and dv_out.data_ptr() % 16 == 0:
                dk_in, dv_in, dk_out, dv_out = dk_out, dv_out, None, None
        else:
            kt, vt = rest[0], rest[1]
            anc_k, anc_v = rest[2:2 + n_anc], rest[2 + n_anc:]
            S = kt.shape[2]
            if kt.stride() == (S * 3 * H * Dh, Dh, 3 * H * Dh, 1):          # slices of the root's packed QKV
                dtext = torch.empty((B, S, 3, H, Dh), dtype=dt, device=dev)
                dkt, dvt = dtext[:, :, 1].permute(0, 2, 1, 3), dtext[:, :, 2].permute(0, 2, 1, 3)
            else:
                dkt = torch.empty_strided(kt.shape, kt.stride(), dtype=dt, device=dev)
                dvt = torch.empty_strided(kt.shape, kt.stride(), dtype=dt, device=dev)
            for _ in range(n_anc):
                buf = torch.empty((B, K, 3, H, Dh), dtype=dt, device=dev)
                d_anc_k.append(buf[:, :, 1].permute(0, 2, 1, 3))
                d_anc_v.append(buf[:, :, 2].permute(0, 2, 1, 3))
            own = lambda t: [t[:, :, n * K:(n + 1) * K] for n in range(NQ)]
            slots_k, slots_v = list(anc_k) + own(kview), list(anc_v) + own(vview)
            dsk, dsv = d_anc_k + own(dkview), d_anc_v + own(dvview)
        delta = torch.empty_like(lse)
        sh = ops._attn_shape(q5, kt, slots_k, scale, max_path, o5, drop)
        sh.dq_sb, sh.dq_sh, sh.dq_sn, sh.dq_sk = dq5.stride()[:4]
        ops.attn_bwd_launch(sh, q5, kt, vt, slots_k, slots_v, ctx_len, path, path_len, o5, lse, do5, dq5, dkt, dvt, dk_in, dv_in, dsk, dsv, delta)
        if dk_out is not None:      # what the descendants' attention sent back to this node's keys / values
            dkview.add_(dk_out)
        if dv_out is not None:
            dvview.add_(dv_out)
        if root:
            return (dqkv, None, None, None)
        return (dqkv, dkt, dvt, None, *d_anc_k, *d_anc_v)


def _dqkv_buffer(qkv, qkv_from):
    """The packed dq / dk / dv buffer = the output gradient of the QKV projection: in that projection's dy stack when it has one."""
    t = None
    if qkv_from is not None and qkv.is_contiguous():
        B, T = qkv.shape[:2]
        t = _deferred_dw_for(qkv.device).dy_slab(qkv_from[0], qkv_from[1], B * T, qkv[0, 0].numel(), qkv.dtype)
    return t.view(qkv.shape) if t is not None else torch.empty_like(qkv)


def packed_attention(qkv, lp, scale, out_for=None, qkv_from=None, p_drop=0.0, drop_site=None, drop_row0=0):
    """qkv (B,T,3,H,Dh).  lp: None (root causal) or the layer view of a BranchPast.  -> (att (B,T,E), (k, v) views).
    out_for: the weight of the projection that reads att; qkv_from: the weight of the projection that produced qkv (see layer_norm /
    add_layer_norm).  drop_site / drop_row0: the dropout masks as functions of (site, logical row) - text rows 0.., tree rows
    from drop_row0 - instead of the call order (ops.drop_site)."""
    B, T, _, H, Dh = qkv.shape
    extra = {"att_buf": _x_slab_for(out_for, B * T, H * Dh, qkv.dtype, qkv.device), "qkv_from": _pass_of(qkv_from, qkv.device),
             "p_drop": p_drop, "drop_site": drop_site, "drop_row0": drop_row0}
    if lp is None:
        att, k, v = _PackedAttention.apply(qkv, None, None, {"scale": float(scale), **extra})
    elif getattr(lp, "merged_text", 0):
        spec = {"scale": float(scale), "ctx_len": lp.ctx_len, "paths": lp.paths, "n_query_nodes": lp.n_query_nodes,
                "n_text": lp.merged_text, **extra}
        att, k, v = _MergedTreeAttention.apply(qkv, spec)
    else:
        spec = {"scale": float(scale), "ctx_len": lp.ctx_len, "paths": lp.paths, "n_query_nodes": lp.n_query_nodes, **extra}
        anc_k = [a[0] for a in lp.ancestors]
        anc_v = [a[1] for a in lp.ancestors]
        att, k, v = _PackedAttention.apply(qkv, lp.text[0], lp.text[1], spec, *anc_k, *anc_v)
    return att, (k, v)
