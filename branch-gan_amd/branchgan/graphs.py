"""HIP-graph capture of the launch-bound part of the train step.

One Branch-GAN step is thousands of small launches (a GPT-2-small step at depth 4 issues ~5k kernels for ~50 ms of GPU work),
so the host cannot keep the queue full.  The forward of the whole GAN step and both backward passes are captured ONCE into a
set of hipGraphs (torch.cuda.graph) and replayed per step; inputs (token ids, branch points) live in static device tensors that are
overwritten before each replay, gradients accumulate into the static flat arenas of branchgan.training.GradArena.
The gradient all-reduce (RCCL) and the optimizer steps stay outside the graph, so the data-parallel path is unchanged.
The reference has no counterpart (it is eager PyTorch); results are identical to the eager step by construction.
"""
import copy
import os

import torch

from . import ops
from . import fused


def _sync_after_replay():
    """Multi-rank runs wait on the host for a replayed graph before they issue the gradient all-reduce that depends on it.
    Measured on MI355X / ROCm 7.2 with 2 ranks (gloo, one GPU): a collective enqueued while the graph it waits for is still
    executing can deadlock (the collective's stream wait never fires; eager launches and finished graphs are fine).  The
    all-reduce + Adam of the generator still overlap the discriminator's backward graph, which is launched right after; the
    cost is one host round trip (~0.1 ms) per graph.  BRANCHGAN_GRAPH_SYNC=0/1 overrides."""
    env = os.environ.get("BRANCHGAN_GRAPH_SYNC")
    if env is not None:
        return env == "1"
    from .training import REHEARSE_RCCL
    return torch.distributed.is_available() and torch.distributed.is_initialized() and (torch.distributed.get_world_size() > 1 or REHEARSE_RCCL)


def _require_fused_path(model, example_ids):
    """Graph replay is supported for models on the fused block path only.  On the module-by-module path the bias gradients
    come from ATen's multi-block reductions, and those return NaNs in a few columns from the SECOND replay of a captured
    graph (measured: GPT-2 small and Pythia-70M at the BASELINE shapes, bf16 autocast, ROCm 7.2 / torch 2.10; the same
    step is fine eagerly, in fp32, and at the tiny test shapes where the reduction fits one workgroup).  The fused path
    computes those gradients in this library's kernels.  BRANCHGAN_GRAPH_UNFUSED=1 overrides (debugging)."""
    if os.environ.get("BRANCHGAN_GRAPH_UNFUSED") == "1":
        return
    from . import models
    for name, lm in (("generator", model.generator.generator), ("discriminator", model.discriminator.nonResidualModel)):
        if not models._fused_eligible(lm, example_ids, None, None, False):
            raise NotImplementedError(
                f"GraphedTrainStep: the {name} does not run on the fused block path (BRANCHGAN_FUSED=0, dropout, a hidden size "
                "that is not a multiple of 8 or above 4096, a head size outside {32, 64, 128, 256}, non-fp32 parameters): "
                "use the eager branchgan.training.train_step for it")


class GraphedTrainStep:
    def __init__(self, model, sampleConf, generatorOptim, discriminatorOptim, example_ids, numUpdateSamples=9999,
                 reinforcementLearningConfig=None, amp_dtype=None, warmup=3):
        self.model, self.g_opt, self.d_opt = model, generatorOptim, discriminatorOptim
        self.rl, self.n_update, self.amp_dtype = reinforcementLearningConfig, numUpdateSamples, amp_dtype
        self.src_conf = sampleConf
        dev = example_ids.device
        _require_fused_path(model, example_ids)
        self.static_ids = example_ids.clone()
        K = sampleConf.numSampleSequences or example_ids.shape[1]
        self.static_sparse = torch.zeros(K, dtype=torch.long, device=dev)
        self.conf = copy.copy(sampleConf)
        self.conf.sparseIndex = self.static_sparse
        self.graph = None
        self.losses = None
        self._sync = _sync_after_replay()
        self._defer_d = self._sync if os.environ.get("BRANCHGAN_DEFER_D") is None else os.environ["BRANCHGAN_DEFER_D"] == "1"
        self._pending_d = None
        self.nan_flag = torch.zeros((), dtype=torch.bool, device=dev)
        self._refresh_sparse(example_ids)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # eager warm-up: builds arenas, caches, workspaces
            for i in range(warmup):
                self._forward_backward()
                self._finish(i)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        timing, ops.TIMING = ops.TIMING, None              # events cannot be recorded inside a capture
        # three graphs: [generator forward], [discriminator forward + losses + generator backward], [discriminator backward];
        # see step() for what runs between them.
        self.graph_g = torch.cuda.CUDAGraph()
        self.graph = torch.cuda.CUDAGraph()
        self.graph_d = torch.cuda.CUDAGraph()
        fused.begin_capture()                              # weight copies cast before the capture must be re-cast inside it
        # ONE stream for the warm-up and all three captures: autograd runs every AccumulateGrad node on the stream it was created
        # on, and a node created while one graph is captured must not end up on another graph's stream
        with torch.cuda.graph(self.graph_g, stream=side, capture_error_mode="thread_local"):   # collective backends run helper threads
            with self._autocast():
                self.gen_state = self.model._generatorForward(self.static_ids, self.conf)
        with torch.cuda.graph(self.graph, pool=self.graph_g.pool(), stream=side, capture_error_mode="thread_local"):
            self.losses = self._forward(self.gen_state)
            self.losses[1][self.g_opt.lossKey].backward()
        with torch.cuda.graph(self.graph_d, pool=self.graph_g.pool(), stream=side, capture_error_mode="thread_local"):
            self.losses[0][self.d_opt.lossKey].backward()
        ops.TIMING = timing
        # the capture ran the backward once for real bookkeeping purposes only: clear what it accumulated
        self.g_opt.arena.zero()
        self.d_opt.arena.zero()

    def _refresh_sparse(self, ids):
        src = self.src_conf
        sp = src.sparseIndex if src.sparseIndex is not None else src.getSparseIndex(ids, ids.device)
        if sp is None:
            sp = torch.arange(ids.shape[1], device=ids.device)
        self.static_sparse.copy_(sp.to(self.static_sparse.device), non_blocking=True)

    def _autocast(self):
        return torch.autocast("cuda", dtype=self.amp_dtype or torch.bfloat16, enabled=self.amp_dtype is not None)

    def _forward(self, generatorState=None):
        with self._autocast():
            return self.model.trainForwardPassNoMetrics(self.static_ids, self.conf, self.n_update, collectTreeDepthData=False,
                                                        reinforcementLearningConfig=self.rl, checkNaN=False,
                                                        generatorState=generatorState)

    def _forward_backward(self):
        d, g = self._forward()
        g[self.g_opt.lossKey].backward()
        d[self.d_opt.lossKey].backward()
        return d, g

    def _finish(self, stepCounter):
        for opt in (self.g_opt, self.d_opt):
            if stepCounter % opt.gradAccum == 0:
                opt.reduce_async()
        for opt in (self.g_opt, self.d_opt):
            opt.finish(stepCounter % opt.gradAccum == 0)

    def _reduce(self, opt, stepCounter):
        stepped = stepCounter % opt.gradAccum == 0
        if stepped:
            opt.reduce_async()
        return stepped

    def _finish_pending(self):
        if self._pending_d is not None:
            self.d_opt.finish(self._pending_d)
            self._pending_d = None

    def step(self, inputIDs, stepCounter):
        """One train step: returns the (static) loss dicts; GanSetup.py:249-256's NaN guard is evaluated on the device and
        raised by check_nan() (one host sync when the caller chooses, not one per loss key per step).

        Schedule: only the gradient all-reduces overlap compute - the generator's runs behind the discriminator's backward
        graph, the discriminator's behind the NEXT step's generator-forward graph (which reads no discriminator weight), so
        with more than one rank the discriminator's optimizer step is deferred to the next call (or `join()`).  The Adam
        launches themselves stay on the main stream: measured on MI355X, an Adam pass (121k workgroups streaming 4 GB)
        running beside a replaying graph costs 0.8 ms per step more than it hides."""
        self.static_ids.copy_(inputIDs, non_blocking=True)
        self._refresh_sparse(inputIDs)
        self.graph_g.replay()                              # generator forward
        self._finish_pending()                             # D's all-reduce of the last step is done by now: its Adam step
        self.graph.replay()                                # discriminator forward + losses + generator backward
        if self._sync:
            torch.cuda.current_stream().synchronize()
        d, g = self.losses
        g_stepped = self._reduce(self.g_opt, stepCounter)
        self.graph_d.replay()                              # discriminator backward, G's all-reduce behind it
        if self._sync:
            torch.cuda.current_stream().synchronize()
        self.nan_flag |= torch.isnan(g['loss'].detach())
        self.g_opt.finish(g_stepped)
        d_stepped = self._reduce(self.d_opt, stepCounter)
        if self._defer_d:
            self._pending_d = d_stepped
        else:
            self.d_opt.finish(d_stepped)
        return d, g

    def join(self):
        """With more than one rank the discriminator's optimizer step of the last `step()` is still pending: call this
        before anything outside `step()` reads the discriminator's parameters."""
        self._finish_pending()

    def check_nan(self):
        self.join()
        if bool(self.nan_flag):
            raise Exception("Generator loss loss is NaN")
