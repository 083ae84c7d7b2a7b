"""HIP-graph capture of the launch-bound part of the train step.

One Branch-GAN step is thousands of small launches (a GPT-2-small step at depth 4 issues ~5k kernels for ~50 ms of GPU work),
so the host cannot keep the queue full.  The forward of the whole GAN step and both backward passes are captured ONCE into a
set of hipGraphs (torch.cuda.graph) and replayed per step; inputs (token ids, branch points) live in static device tensors that are
overwritten before each replay, gradients accumulate into the static flat arenas of branchgan.training.GradArena.
The gradient all-reduce (RCCL) and the optimizer steps stay outside the graph, so the data-parallel path is unchanged.
The reference has no counterpart (it is eager PyTorch); results are identical to the eager step by construction.
"""
import copy
import ctypes
import os

import torch

from . import _capi as C
from . import ops
from . import fused
from .sampling import SamplingTechniques


def new_graph():
    """A CUDAGraph whose captured hipGraph is kept until `finalize_graph` has cleaned it (instantiation is deferred)."""
    return torch.cuda.CUDAGraph(keep_graph=True)


def finalize_graph(graph):
    """Replaces every MEMSET node of a freshly captured graph by a fill-kernel node, then instantiates it.

    Cause (measured, benchmarks/probes/graph_anomalies.py, profiles/r2_a_graph_anomalies.txt): on ROCm 7.2.0 a memset node - what a
    captured hipMemsetAsync becomes - clears its destination on the FIRST replay and fills it with garbage from the second
    on.  ATen's multi-block reductions clear their arrival counters with such a call before every launch, so from the second
    replay no workgroup sees itself arrive last and the result is never written: the stale / NaN bias gradients round 1
    met on the module-by-module path (the fused path computes those sums in this library's kernels, which use no memset).
    Returns (nodes, kernel nodes, memset nodes replaced, memcpy nodes)."""
    raw = ctypes.c_void_p(graph.raw_cuda_graph())
    c = [ctypes.c_int() for _ in range(4)]
    C.check(C.lib().bg_graph_node_census(raw, *[ctypes.byref(x) for x in c]), "bg_graph_node_census")
    n = ctypes.c_int()
    C.check(C.lib().bg_graph_replace_memset_nodes(raw, ctypes.byref(n)), "bg_graph_replace_memset_nodes")
    if n.value != c[2].value:
        raise RuntimeError(f"finalize_graph: {c[2].value} memset nodes, {n.value} replaced")
    graph.instantiate()
    return c[0].value, c[1].value, n.value, c[3].value


def _sync_after_replay():
    """Whether step() waits on the host for a replayed graph before it issues the gradient all-reduce.

    History: round 1 saw two gloo ranks sharing one GPU stall in `arena.wait()` and blamed a collective enqueued behind a
    RUNNING graph.  Round 2 put native stacks on it (benchmarks/probes/hang_probe.sh, profiles/r2_a_graph_anomalies.txt): the same
    stall happens with NO graph at all (eager train_step, GPT-2-small arenas of ~1 GB in 64 MB buckets; tiny models pass) -
    one rank's gloo workers sit in hipStreamSynchronize (gloo stages every bucket through pinned host memory from its own
    worker threads), the other rank's wait in TCP recv for them.  It is gloo's host-staged CUDA path under two processes on
    one device, not graph replay, and a host sync after a replay neither causes nor cures it.  RCCL collectives are
    stream-ordered device work, so the production path (backend nccl) does not sync; gloo (host-driven anyway) keeps the
    sync so its worker threads never poll a stream a graph is still running on.  BRANCHGAN_GRAPH_SYNC=0/1 overrides."""
    env = os.environ.get("BRANCHGAN_GRAPH_SYNC")
    if env is not None:
        return env == "1"
    dist = torch.distributed
    return dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo" and dist.get_world_size() > 1


def _multi_rank():
    from .training import REHEARSE_RCCL
    dist = torch.distributed
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or REHEARSE_RCCL)


def _rule_strategies(rules):
    for r in rules:
        yield r.samplingStrategy
        if r.firstBeamSamplingStrategy is not None:
            yield r.firstBeamSamplingStrategy


def _check_rules(rules):
    """A captured step replays the same launches with the same arguments.  TOP_K is a pure function of the logits; the
    exponential-race techniques read their noise from a tensor, which step() refreshes in place before every replay
    (`_static_noise`).  RANDOM / SAMPLE / EPSILON_* draw from host-side or global generators inside the sampling call
    (SamplingStrategies.py:133-146,190-194,217-233): captured, they would replay one frozen draw - refused."""
    T = SamplingTechniques
    ok = {T.TOP_K, T.DEBUG_GRAB_FIRST_IDS, T.SAMPLE_NO_REPLACEMENT, T.INVERSE_TOP_K_SAMPLE_NO_REPLACEMENT, T.TOP_P_SAMPLE_NO_REPLACEMENT}
    for strat in _rule_strategies(rules):
        strat = strat if isinstance(strat, T) else T(int(getattr(strat, "value", strat)))
        if strat not in ok:
            raise NotImplementedError(f"GraphedTrainStep: sampling technique {strat.name} draws random numbers inside the sampling call; "
                                      "a HIP graph would replay one frozen draw.  Use the eager branchgan.training.train_step for it")
    for r in rules:
        if 'noise_level' in (r.samplingParams or {}) or 'noise_level' in (r.firstBeamSamplingParams or {}):
            raise NotImplementedError("GraphedTrainStep: 'noise_level' (Dirichlet noise from torch's global generator) cannot be captured")


def _fresh_blas_workspaces():
    """Drops torch's per-(handle, stream) rocBLAS / hipBLASLt workspaces: the next GEMM allocates a new one - inside a capture, from
    that capture's memory pool, where it stays for the graph's lifetime."""
    torch._C._cuda_clearCublasWorkspaces()


def _settle_collectives():
    """Before a capture: every collective this library enqueued must have completed (training.settle_collectives - the handles are
    kept for exactly this).  They all run on RCCL's own stream (async_op=True), whose end events a capture on another stream does
    not disturb, so ProcessGroupNCCL's watchdog - which aborts the process when an event query fails - can poll whatever it has
    not retired yet while the captures run (training._INFLIGHT has the story and the probe).  Round 2 held the watchdog off with a
    0.35 s sleep instead; BRANCHGAN_CAPTURE_SETTLE_MS brings a delay back for a runtime where the probe says otherwise (failure
    signature: "Process group watchdog thread terminated with exception: HIP error: operation not permitted on an event last
    recorded in a capturing stream", then abort)."""
    from . import training
    training.settle_collectives()
    extra = float(os.environ.get("BRANCHGAN_CAPTURE_SETTLE_MS", "0"))
    if extra > 0:
        import time
        time.sleep(extra / 1e3)


def _both_models_large(model, hidden=1024):
    """Generator and discriminator both at least `hidden` wide (see GraphedTrainStep.__init__: stream-K GEMMs side by side)."""
    widths = []
    for part in (model.generator, model.discriminator):
        cfg = next((m.config for m in part.modules() if hasattr(getattr(m, "config", None), "hidden_size")), None)
        widths.append(cfg.hidden_size if cfg is not None else 0)
    return min(widths) >= hidden


class GraphedTrainStep:
    """`warmup` eager steps lay out the arenas, caches and workspaces; they run on `example_ids` and, unless
    restore_after_warmup=False, parameters, optimizer state (moments, step counts) and LR schedulers are put back to what
    they were before, so building a graphed step does not train the model."""

    def __init__(self, model, sampleConf, generatorOptim, discriminatorOptim, example_ids, numUpdateSamples=9999,
                 reinforcementLearningConfig=None, amp_dtype=None, warmup=3, restore_after_warmup=True):
        self.model, self.g_opt, self.d_opt = model, generatorOptim, discriminatorOptim
        self.rl, self.n_update, self.amp_dtype = reinforcementLearningConfig, numUpdateSamples, amp_dtype
        self.src_conf = sampleConf
        dev = example_ids.device
        _check_rules(sampleConf.rules)
        self.depth = len(sampleConf.rules)
        self.static_ids = example_ids.clone()
        K = sampleConf.numSampleSequences or example_ids.shape[1]
        self.static_sparse = torch.zeros(K, dtype=torch.long, device=dev)
        self.conf = copy.copy(sampleConf)
        self.conf.sparseIndex = self.static_sparse
        self.graph = None
        self.losses = None
        self._sync = _sync_after_replay()
        self._defer_d = _multi_rank() if os.environ.get("BRANCHGAN_DEFER_D") is None else os.environ["BRANCHGAN_DEFER_D"] == "1"
        self._pending_d = None
        # one rank: the discriminator's backward graph is replayed on a second stream BESIDE the generator's (see _step)
        self._overlap = (not self._defer_d) if os.environ.get("BRANCHGAN_OVERLAP_BWD") is None else os.environ["BRANCHGAN_OVERLAP_BWD"] == "1"
        if not hasattr(torch._C, "_cuda_clearCublasWorkspaces"):      # no way to give each graph its own GEMM workspace: one after the other
            self._overlap = False
        if self._overlap and os.environ.get("BRANCHGAN_OVERLAP_BWD") is None and _both_models_large(model):
            # Nearly every hipBLASLt kernel is a stream-K kernel (SK3): workgroups that spin on partial tiles of their siblings, i.e.
            # that count on the whole grid being resident.  Two of them from two streams, each with one 256 x 256 workgroup per CU,
            # can hold half the chip each and wait for the other half for ever.  Seen at configs[3] only (generator h 2048 beside
            # discriminator h 1024: a replay that never ends, 2 runs of 7 with one kernel variant, every run with another, none of
            # ~25 with the shipped kernels; 0 of 8 in turn), never at configs[1] / [2] in thousands of steps (one of the two models'
            # GEMMs is small there).  Two large models replay their backward graphs in turn (configs[3]: 77.8 ms instead of 73.2);
            # BRANCHGAN_OVERLAP_BWD=1 asks for side by side regardless.
            self._overlap = False
        self.nan_flag = torch.zeros((), dtype=torch.bool, device=dev)
        # exponential noise of the race techniques: persistent tensors, one per sampling call of a step, refreshed before a replay
        self._noise, self._noise_cursor, self._user_noise = [], 0, model.noise_fn
        self._refresh_sparse(example_ids)
        # ONE side stream per model, shared by every graphed step built on it (the training loop builds one per tree depth):
        # autograd binds a parameter's AccumulateGrad node to the stream of its first use and keeps the node while any captured
        # graph of the model lives, so a second graphed step warmed up on ANOTHER stream would trip the guard below
        side = model.__dict__.get("_graph_stream")
        if side is None or side.device != dev:
            side = torch.cuda.Stream(device=dev)
            model.__dict__["_graph_stream"] = side
        self.stream = side
        aux = model.__dict__.get("_graph_aux_stream")
        if aux is None or aux.device != dev:
            aux = torch.cuda.Stream(device=dev)
            model.__dict__["_graph_aux_stream"] = aux
        self.aux = aux
        if os.environ.get("BRANCHGAN_DEBUG_STREAMS"):
            print(f"[streams] side {side.cuda_stream:#x}  g_opt.stream {getattr(getattr(generatorOptim, 'stream', None), 'cuda_stream', 0):#x}  "
                  f"d_opt.stream {getattr(getattr(discriminatorOptim, 'stream', None), 'cuda_stream', 0):#x}  current {torch.cuda.current_stream().cuda_stream:#x}", flush=True)
        self._own_stream = os.environ.get("BRANCHGAN_REPLAY_STREAM", "0") == "1"
        side.wait_stream(torch.cuda.current_stream())
        if warmup < 1:
            raise ValueError("GraphedTrainStep: at least one warm-up step (it lays out the arenas, workspaces and noise tensors)")
        saved = self._snapshot() if restore_after_warmup else None
        module_forwards = fused.module_path_forwards
        model.noise_fn = self._static_noise
        try:
            import warnings
            # the warning looked for below is a warn-ONCE in libtorch: without set_warn_always a second graphed step of the same
            # process would never see it
            warn_always = torch.is_warn_always_enabled()
            torch.set_warn_always(True)
            try:
                with torch.cuda.stream(side), warnings.catch_warnings(record=True) as caught:   # eager warm-up: builds arenas, caches, workspaces
                    warnings.simplefilter("always")
                    for i in range(warmup):
                        self._refresh_noise()
                        self._forward_backward()
                        self._finish(i)
                    if saved is not None:
                        self._restore(saved)
            finally:
                torch.set_warn_always(warn_always)
            for w in caught:
                # autograd keeps a parameter's AccumulateGrad node alive, bound to the stream it was created on, for as long as an
                # autograd graph refers to it.  A capture would then make the capture stream synchronise with that other stream;
                # when it is the default stream, ROCm 7.2's hipStreamEndCapture segfaults (graph_anomalies.py segv-nonleaf)
                if "AccumulateGrad node's stream does not match" in str(w.message):
                    raise RuntimeError("GraphedTrainStep: an autograd graph of this model built on another stream is still alive (losses or "
                                       "outputs of an earlier eager step): drop those references before building the graphed step")
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            _settle_collectives()
            timing, ops.TIMING = ops.TIMING, None              # events cannot be recorded inside a capture
            # three graphs: [generator forward], [discriminator forward + losses + generator backward], [discriminator backward];
            # see step() for what runs between them.
            self.graph_g, self.graph, self.graph_d = new_graph(), new_graph(), new_graph()
            self.graph_gb = new_graph() if self._overlap else None
            fused.begin_capture()                              # weight copies cast before the capture must be re-cast inside it
            # ONE stream for the warm-up and all three captures: autograd runs every AccumulateGrad node on the stream it was created
            # on, and a node created while one graph is captured must not end up on another graph's stream.  (A node that lives on the
            # DEFAULT stream is worse: autograd then pulls the null stream into the capture and ROCm 7.2's hipStreamEndCapture
            # segfaults walking the capture's streams - benchmarks/probes/graph_anomalies.py segv-nonleaf.)
            self._noise_cursor = 0
            if self._overlap:
                _fresh_blas_workspaces()
            with torch.cuda.graph(self.graph_g, stream=side, capture_error_mode="thread_local"):   # collective backends run helper threads
                with self._autocast():
                    self.gen_state = self.model._generatorForward(self.static_ids, self.conf)
            # the sampled branch tokens of every tree node, (B, K) each, valid after a replay of graph_g (the second graph clears the tree)
            self.sampled_ids = [n.ids for n in self.gen_state[0].getSubNodes()]
            if not self._overlap:
                with torch.cuda.graph(self.graph, pool=self.graph_g.pool(), stream=side, capture_error_mode="thread_local"):
                    self.losses = self._forward(self.gen_state)
                    self.losses[1][self.g_opt.lossKey].backward()
                with torch.cuda.graph(self.graph_d, pool=self.graph_g.pool(), stream=side, capture_error_mode="thread_local"):
                    self.losses[0][self.d_opt.lossKey].backward()
            else:
                # Four graphs; graph_d runs BESIDE graph_gb, so the two must share nothing they write:
                #  * memory pools: a pool hands blocks freed by one capture to the next (which is right when the graphs replay one
                #    after the other) - here every graph gets its own, so no graph's temporaries sit in blocks another one uses;
                #  * GEMM workspaces: torch keeps one per (handle, stream) and all captures run on `side`; nearly every hipBLASLt
                #    kernel here is a stream-K one that keeps partial tiles there - dropped before each capture, so that each graph
                #    allocates its own (from its own pool);
                #  * this library's scratch (column-reduction partials, workspaces): fused.set_lane.
                _fresh_blas_workspaces()
                with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
                    self.losses = self._forward(self.gen_state)
                _fresh_blas_workspaces()
                with torch.cuda.graph(self.graph_gb, stream=side, capture_error_mode="thread_local"):
                    self.losses[1][self.g_opt.lossKey].backward()
                _fresh_blas_workspaces()
                lane = fused.set_lane(1)
                try:
                    with torch.cuda.graph(self.graph_d, stream=side, capture_error_mode="thread_local"):
                        self.losses[0][self.d_opt.lossKey].backward()
                finally:
                    fused.set_lane(lane)
                _fresh_blas_workspaces()
            ops.TIMING = timing
        finally:
            model.noise_fn = self._user_noise
        graphs = [g for g in (self.graph_g, self.graph, self.graph_gb, self.graph_d) if g is not None]
        self.census = [finalize_graph(g) for g in graphs]     # memset nodes -> fill kernels
        # Side by side only what this library accounts for.  The module-by-module path reduces bias gradients with ATen's
        # multi-block reductions (they are what brings memset nodes into a capture); replayed beside each other, the two backward
        # graphs then disagree with the eager step in a few elements of those gradients (benchmarks/probes/overlap_mismatch.py:
        # only *.bias, both models), while the fused path is bit-identical to it.  Such a step replays its four graphs in turn.
        self._side_by_side = self._overlap and fused.module_path_forwards == module_forwards and \
            all(c[2] == 0 for c in self.census[2:])
        # the capture ran the backward once for real bookkeeping purposes only: clear what it accumulated
        self.g_opt.arena.zero()
        self.d_opt.arena.zero()

    # ---- noise of the exponential-race techniques -------------------------------------------------------------------------
    def _static_noise(self, shape, device):
        """model.noise_fn while this object builds: the i-th sampling call of a step always gets the i-th persistent tensor."""
        i, self._noise_cursor = self._noise_cursor, self._noise_cursor + 1
        if i == len(self._noise):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("GraphedTrainStep: a sampling call asked for noise that the warm-up steps never drew")
            self._noise.append(torch.empty(tuple(shape), dtype=torch.float32, device=device))
            self._fill_noise(self._noise[i])
        if tuple(self._noise[i].shape) != tuple(shape):
            raise RuntimeError(f"GraphedTrainStep: noise request {i} changed shape {tuple(self._noise[i].shape)} -> {tuple(shape)}")
        return self._noise[i]

    def _fill_noise(self, t):
        if self._user_noise is not None:                     # an injected noise source (tests): asked eagerly, copied in place
            t.copy_(self._user_noise(tuple(t.shape), t.device))
        else:                                                # e = -log(U): Exp(1), what SampleBuffer.py:28 draws (fresh, not rolled)
            t.exponential_()

    def _refresh_noise(self):
        self._noise_cursor = 0
        for t in self._noise:
            self._fill_noise(t)

    # ---- warm-up bookkeeping ------------------------------------------------------------------------------------------------
    def _snapshot(self):
        opts = (self.g_opt, self.d_opt)
        for o in opts:               # sharded data parallelism: each rank holds the moments of its own slice only; every rank
            if hasattr(o.optimizer, "gather_state"):       # builds the step at the same point, so the collective lines up
                o.optimizer.gather_state()
        return {"params": [p.detach().clone() for p in self.model.parameters()],
                "optim": [copy.deepcopy(o.optimizer.state_dict()) for o in opts],
                "sched": [copy.deepcopy(o.scheduler.state_dict()) if o.scheduler is not None else None for o in opts]}

    def _restore(self, saved):
        with torch.no_grad():
            for p, v in zip(self.model.parameters(), saved["params"]):
                p.copy_(v)
        for o, st, sc in zip((self.g_opt, self.d_opt), saved["optim"], saved["sched"]):
            opt = o.optimizer
            if hasattr(opt, "reset_state"):                  # optim.ArenaAdam: moments and step count live in its arenas
                opt.reset_state()
            opt.load_state_dict(st)
            if hasattr(opt, "refresh_weight_copies"):
                opt.refresh_weight_copies()
            if sc is not None:
                o.scheduler.load_state_dict(sc)
            o.arena.zero()
        fused.invalidate_weights()

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

    def step(self, inputIDs, stepCounter, updateGen=True, updateDisc=True, sampleConf=None):
        """updateGen / updateDisc: the update schedule of TrainingLoop.py:129-134.  A model that is not updated this step still runs
        (its backward is part of the captured graphs) and its gradients are discarded, which equals skipping its
        performUpdateStep when gradAccum == 1 (refused otherwise).  sampleConf, when given, must have the captured depth."""
        if sampleConf is not None and len(sampleConf.rules) != self.depth:
            raise ValueError(f"GraphedTrainStep was captured at depth {self.depth}; got a sampling config of depth {len(sampleConf.rules)} "
                             "(capture one GraphedTrainStep per depth: training.efficentTrainingLoop(useGraphs=True) does)")
        for flag, opt in ((updateGen, self.g_opt), (updateDisc, self.d_opt)):
            if not flag and opt.gradAccum != 1:
                raise NotImplementedError("GraphedTrainStep: skipping a model's update needs gradAccum == 1")
        if not self._own_stream:
            return self._step(inputIDs, stepCounter, updateGen, updateDisc)
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            out = self._step(inputIDs, stepCounter, updateGen, updateDisc)
        cur.wait_stream(self.stream)
        return out

    def _step(self, inputIDs, stepCounter, updateGen=True, updateDisc=True):
        """One train step: returns the (static) loss dicts; GanSetup.py:249-256's NaN guard is evaluated on the device and
        raised by check_nan() (one host sync when the caller chooses, not one per loss key per step).

        Schedule: only the gradient all-reduces overlap compute - the generator's runs behind the discriminator's backward
        graph, the discriminator's behind the NEXT step's generator-forward graph (which reads no discriminator weight), so
        with more than one rank the discriminator's optimizer step is deferred to the next call (or `join()`).  The Adam
        launches themselves stay on the main stream: measured on MI355X, an Adam pass (121k workgroups streaming 4 GB)
        running beside a replaying graph costs 0.8 ms per step more than it hides."""
        self.static_ids.copy_(inputIDs, non_blocking=True)
        self._refresh_sparse(inputIDs)
        self._refresh_noise()
        if self._overlap:
            return self._step_overlapped(stepCounter, updateGen, updateDisc)
        self.graph_g.replay()                              # generator forward
        self._finish_pending()                             # D's all-reduce of the last step is done by now: its Adam step
        self.graph.replay()                                # discriminator forward + losses + generator backward
        if self._sync:
            torch.cuda.current_stream().synchronize()
        d, g = self.losses
        if not updateGen:
            self.g_opt.arena.zero()
        g_stepped = updateGen and self._reduce(self.g_opt, stepCounter)
        self.graph_d.replay()                              # discriminator backward, G's all-reduce behind it
        if self._sync:
            torch.cuda.current_stream().synchronize()
        self.nan_flag |= torch.isnan(g['loss'].detach())
        self.g_opt.finish(g_stepped)
        if not updateDisc:
            self.d_opt.arena.zero()
        d_stepped = updateDisc and self._reduce(self.d_opt, stepCounter)
        if self._defer_d:
            self._pending_d = d_stepped
        else:
            self.d_opt.finish(d_stepped)
        return d, g

    def replay_graphs(self):
        """The captured graphs in the order (and on the streams) a step replays them, without reductions and optimizer steps."""
        self.graph_g.replay()
        self.graph.replay()
        if not self._side_by_side:
            if self.graph_gb is not None:
                self.graph_gb.replay()
            self.graph_d.replay()
            return
        main = torch.cuda.current_stream()
        self.aux.wait_stream(main)
        with torch.cuda.stream(self.aux):
            self.graph_d.replay()
        self.graph_gb.replay()
        main.wait_stream(self.aux)

    def _step_overlapped(self, stepCounter, updateGen, updateDisc):
        """One rank.  The generator's backward is ~450 launches on 256-row batches (one tree level at a time) that leave most of
        the chip idle; the discriminator's backward does not depend on it.  graph_d is replayed on a second stream beside
        graph_gb (measured at configs[1]: 3.15 ms + 4.4 ms one after the other, ~6 ms side by side); both optimizer steps follow
        on the main stream once the streams have joined, so nothing outlives the call."""
        main = torch.cuda.current_stream()
        self.graph_g.replay()                              # generator forward
        self.graph.replay()                                # discriminator forward + losses
        if self._side_by_side:
            self.aux.wait_stream(main)
            with torch.cuda.stream(self.aux):
                self.graph_d.replay()                      # discriminator backward
            self.graph_gb.replay()                         # generator backward
        else:
            self.graph_gb.replay()
            self.graph_d.replay()
        if self._sync:
            main.synchronize()
        d, g = self.losses
        self.nan_flag |= torch.isnan(g['loss'].detach())
        if not updateGen:
            self.g_opt.arena.zero()
        self.g_opt.finish(updateGen and self._reduce(self.g_opt, stepCounter))
        main.wait_stream(self.aux)
        if not updateDisc:
            self.d_opt.arena.zero()
        self.d_opt.finish(updateDisc and self._reduce(self.d_opt, stepCounter))      # (on the second stream instead: no faster)
        return d, g

    def join(self):
        """With more than one rank the discriminator's optimizer step of the last `step()` is still pending: call this
        before anything outside `step()` reads the discriminator's parameters."""
        self._finish_pending()

    def check_nan(self):
        self.join()
        if bool(self.nan_flag):
            raise Exception("Generator loss loss is NaN")
