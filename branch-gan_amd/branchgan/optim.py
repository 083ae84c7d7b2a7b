"""Adam over flat arenas - the optimizer half of the train step as one HBM-streaming launch per model.

The reference builds `torch.optim.Adam(params, lr=lr)` (Training/Optimizers.py:6-16) and per optimizer step runs
`optimizer.step(); optimizer.zero_grad()` (Optimizers.py:75-92); autocast then re-casts every weight to bf16 on its first use
in the next forward.  On MI355X that is three passes over the model state per step (a multi-tensor Adam, a multi-tensor cast,
a fill), ~3 ms of a 25 ms GPT-2-small step.  `ArenaAdam` keeps parameters, both moments and the bf16 weight copies in flat
arenas next to branchgan.training.GradArena's flat gradients and updates all of them with ONE call of `bg_adam_step`
(include/branchgan.h): read p, g, m, v; write p, m, v, bf16(p), g = 0.  Same mathematics and order of operations as
torch.optim.Adam (amsgrad / maximize / capturable are not offered); the 1/world of the data-parallel mean is folded in.

The bf16 copies are handed to branchgan.fused as "pinned" weight copies, so a fused forward (eager or captured in a HIP graph)
reads them directly and never casts.
"""
import torch

from . import _capi as C
from . import fused


class ArenaAdam(torch.optim.Optimizer):
    zeroes_grads = True          # step() leaves every gradient it consumed at zero: no zero_grad pass afterwards
    folds_grad_scale = True      # step() multiplies the gradient by `grad_scale`: the all-reduce may stay a plain sum

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, lowp_dtype=torch.bfloat16):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("ArenaAdam: hyper-parameters out of range")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("ArenaAdam: one parameter group (one flat arena) only")
        self.lowp_dtype = lowp_dtype
        self.grad_scale = 1.0
        self.steps = 0
        self.grad_arena = None      # branchgan.training.GradArena, set by bind()
        self.param_arena = self.exp_avg = self.exp_avg_sq = self.lowp_arena = None
        self._outside = []
        self._pending_state = None  # a state dict loaded before the arenas exist: copied in by _build
        self.timing = None          # a list: step() appends (start, end) HIP events recorded around its launch (bench.py)

    def bind(self, grad_arena):
        """OptimizerWrapper hands over the model's GradArena: this optimizer steps exactly the parameters whose gradients
        live in it, at the same offsets."""
        self.grad_arena = grad_arena

    # -----------------------------------------------------------------------------------------------------------------
    def _build(self):
        ga = self.grad_arena
        if ga is None:
            raise RuntimeError("ArenaAdam: bind() a GradArena first (OptimizerWrapper does)")
        if ga.arena is None:
            ga._build()
        own = {id(p) for p in self.param_groups[0]['params']}
        if any(id(p) not in own for p in ga.params):
            raise RuntimeError("ArenaAdam: the gradient arena holds parameters this optimizer was not given")
        for p in ga.params:
            if p.dtype != torch.float32 or not p.is_cuda:
                raise RuntimeError("ArenaAdam: fp32 parameters on the GPU only")
        n, dev = ga.arena.numel(), ga.arena.device
        self.param_arena = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        cur = torch.cuda.current_stream(dev)
        with torch.no_grad():
            for p, off in zip(ga.params, ga.offsets):
                view = self.param_arena[off:off + p.numel()].view(p.shape)
                view.copy_(p.detach())
                # the old storage goes back to its allocation stream's pool the moment p.data is re-pointed: tell the
                # allocator that THIS stream still reads it (a no-op when this is the stream it was allocated on)
                p.data.record_stream(cur)
                p.data = view
        inside = {id(p) for p in ga.params}
        self._outside = [p for p in self.param_groups[0]['params'] if id(p) not in inside]
        if self.lowp_dtype is not None:
            self.lowp_arena = self.param_arena.to(self.lowp_dtype)
            for p, off in zip(ga.params, ga.offsets):
                fused.pin_weight_copy(p, self.lowp_arena[off:off + p.numel()].view(p.shape), self)
        fused.invalidate_weights()
        if self._pending_state:
            self._copy_state_in()

    def ensure_built(self):
        """Lays the arenas out now, on the caller's stream (training.OptimizerWrapper calls this right after the first
        backward, before any work moves to a side stream)."""
        if self.param_arena is None and self.grad_arena is not None and self.grad_arena.arena is not None:
            self._build()

    def reset_state(self):
        """Back to a freshly built optimizer: zero moments, step count 0 (the arenas and the parameter views stay)."""
        self.steps = 0
        self._pending_state = None
        if self.exp_avg is not None:
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
        self.state.clear()
        self._moments_gathered = True     # zero everywhere is whole on every rank

    def refresh_weight_copies(self):
        """After parameters were changed by hand (load_state_dict, .data edits): re-round the low-precision copies."""
        if self.lowp_arena is not None:
            self.lowp_arena.copy_(self.param_arena)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self.param_arena is None:
            self._build()
        for p in self._outside:
            if p.grad is not None:
                raise RuntimeError("ArenaAdam: a parameter received its first gradient after the arenas were laid out "
                                   f"(shape {tuple(p.shape)}); every trained parameter needs a gradient in the first step")
        g = self.param_groups[0]
        self.steps += 1
        lp = self.lowp_arena
        ga = self.grad_arena
        lo, hi = ga.shard_range()
        sharded = ga.sharded_active()
        if sharded and ga.shard is None:
            raise RuntimeError("ArenaAdam: sharded step before the gradients were reduce-scattered (OptimizerWrapper.backward does it)")
        grads = ga.shard if sharded else ga.arena          # the reduce-scatter left this rank's summed slice in ga.shard
        if self.timing is not None:
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
        C.check(C.lib().bg_adam_step(C.ptr(self.param_arena[lo:hi]), C.ptr(grads), C.ptr(self.exp_avg[lo:hi]), C.ptr(self.exp_avg_sq[lo:hi]),
                                     C.ptr(lp) if lp is not None and not sharded else None, C.dtype_code(lp) if lp is not None else 0, hi - lo,
                                     float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']),
                                     float(g['weight_decay']), self.steps, float(self.grad_scale), 1, C.stream_ptr()), "bg_adam_step")
        if self.timing is not None:
            t1.record()
            self.timing.append((t0, t1))
        if sharded:
            # every rank gets every slice of the fp32 parameters; the low-precision copies are re-rounded here (6 B per parameter
            # of HBM traffic against 2 B per parameter over the links) and the local gradients, consumed by nobody, are cleared
            from .training import all_gather_slices
            if getattr(self, "_slice_out", None) is None or self._slice_out.numel() != hi - lo:
                self._slice_out = torch.empty(hi - lo, dtype=torch.float32, device=self.param_arena.device)
            self._slice_out.copy_(self.param_arena[lo:hi])
            all_gather_slices(self.param_arena, self._slice_out, ga.group)
            if lp is not None:
                lp.copy_(self.param_arena)
            ga.arena.zero_()
            self._moments_gathered = False
        return loss

    def gather_state(self):
        """Sharded data parallelism: a collective that gives every rank the whole Adam moments (before state_dict / a save).
        Nothing to do otherwise."""
        if self.param_arena is None or self.grad_arena is None or getattr(self, "_moments_gathered", True):
            return
        from .training import all_gather_slices
        lo, hi = self.grad_arena.shard_range()
        for t in (self.exp_avg, self.exp_avg_sq):
            all_gather_slices(t, t[lo:hi].clone(), self.grad_arena.group)
        self._moments_gathered = True

    def zero_grad(self, set_to_none=False):
        if self.grad_arena is not None and self.grad_arena.arena is not None:
            self.grad_arena.arena.zero_()
        else:
            super().zero_grad(set_to_none=set_to_none)

    # torch.optim.Adam's state layout, so checkpoints move between the two ------------------------------------------------
    def state_dict(self):
        if not getattr(self, "_moments_gathered", True):
            raise RuntimeError("ArenaAdam.state_dict under BRANCHGAN_DP_SHARDED=1: call gather_state() on every rank first "
                               "(each rank holds the moments of its own slice only)")
        if self.param_arena is not None:
            step = torch.tensor(float(self.steps))
            for p, off in zip(self.grad_arena.params, self.grad_arena.offsets):
                sl = slice(off, off + p.numel())
                self.state[p] = {'step': step.clone(), 'exp_avg': self.exp_avg[sl].view(p.shape), 'exp_avg_sq': self.exp_avg_sq[sl].view(p.shape)}
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """torch.optim.Adam's layout.  Before the first step (the resume flow of SpawnedExperiment.py:27-34: build the
        optimizer, load its start weights, then train) the arenas do not exist yet: the state is kept and copied in when
        the first step lays them out."""
        super().load_state_dict(state_dict)
        if not self.state:
            return
        if self.param_arena is None:
            self._pending_state = True
            return
        self._copy_state_in()

    def _copy_state_in(self):
        with torch.no_grad():
            for p, off in zip(self.grad_arena.params, self.grad_arena.offsets):
                st = self.state.get(p)
                if st:
                    sl = slice(off, off + p.numel())
                    self.exp_avg[sl].view(p.shape).copy_(st['exp_avg'])
                    self.exp_avg_sq[sl].view(p.shape).copy_(st['exp_avg_sq'])
                    self.steps = int(float(st['step']))
        self._pending_state = None
        self.state.clear()
        self._moments_gathered = True     # a loaded state dict holds every slice
