"""Batch-sharded data-parallel training step — host side.

Mirrors Training/SpawnedExperiment.py:47-82,165-191 (one process per GPU, per-rank batch split, rank-0-only logging),
Training/Optimizers.py:50-92 (OptimizerWrapper.performUpdateStep) and the step body of Training/TrainingLoop.py:129-158.

MI355X design: the reference wraps G and D in DDP (find_unused_parameters=True) and all-reduces on EVERY micro-step and every
DDP.forward (one per tree node).  Here each model owns one flat fp32 gradient arena that its parameters' .grad tensors are
views of; after the backward of an optimizer step the arena is all-reduced in a few large RCCL buckets (xGMI links are
point-to-point: fewer, larger collectives) on a side stream, overlapping the generator's reduction with the discriminator's
backward.  Parameters that never receive a gradient (generator valueHead, discriminator LM head) are left out of the arena.
Mathematically identical to the reference's averaging (sum / world).
"""
import contextlib
import os

import torch
import torch.distributed as dist


TUNED_GEMMS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tuning", "tunableop_gfx950.csv")


def use_tuned_gemms(path=TUNED_GEMMS):
    """The dense GEMMs go through hipBLASLt; for the GPT-2-small shapes of BASELINE configs[1] its default heuristic is a few
    percent off the best kernel it has.  This loads the choices a PYTORCH_TUNABLEOP tuning run recorded on an MI355X
    (`branch-gan_amd/tuning/`, regenerate with PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 python bench.py) and
    turns tuning itself off, so nothing is timed at run time and unknown shapes keep the default.  A file recorded with another
    hipBLASLt version is ignored by TunableOp's validators.  Returns whether the table was loaded."""
    try:
        import torch.cuda.tunable as tunable
        if not (torch.cuda.is_available() and os.path.exists(path)):
            return False
        tunable.enable(True)
        record = os.environ.get("BRANCHGAN_TUNE_GEMMS")       # a path: time the shapes the table lacks and write table + additions there
        tunable.tuning_enable(bool(record))
        tunable.record_untuned_enable(False)
        keep = getattr(torch._C, "_cuda_tunableop_write_file_on_exit", None)
        if keep is not None:
            keep(bool(record))    # otherwise read-only use of the table
        ok = bool(tunable.read_file(path))
        tunable.set_filename(record or path, insert_device_ordinal=False)
        return ok
    except Exception:           # an optional speed-up: never a reason to fail
        return False


def worker_batch_size(batchSize, worldSize, rank):
    """SpawnedExperiment.py:63-68: floor(B / W), the remainder spread over the first ranks."""
    return batchSize // worldSize + (1 if rank < batchSize % worldSize else 0)


# BRANCHGAN_REHEARSE_RCCL=1: run the multi-rank code path (process group, bucketed all-reduce on the side stream, the graph
# schedule's host syncs and deferred discriminator step) with ONE rank - the only way to put RCCL itself under the graph replay
# schedule on a one-GPU box.
REHEARSE_RCCL = os.environ.get("BRANCHGAN_REHEARSE_RCCL") == "1"


def init_data_parallel(backend=None):
    """Reads the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), one process per GPU.
    backend 'nccl' is RCCL on ROCm; 'gloo' is used by the CPU tests.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or REHEARSE_RCCL) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "12345")
        # BRANCHGAN_DIST_BACKEND=gloo: rehearse the multi-rank path with several ranks on ONE GPU (RCCL refuses that)
        backend = backend or os.environ.get("BRANCHGAN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


# BRANCHGAN_DP_SHARDED=1: instead of all-reducing the gradient arena and stepping every parameter on every rank, reduce-scatter it
# (rank r gets the sum of slice r), let optim.ArenaAdam step slice r only - 1/world of the optimizer's 34 B per parameter - and
# all-gather the fp32 parameters.  Same bytes on the links (reduce-scatter + all-gather = one ring all-reduce), same result bit
# for bit (the kernel is element-wise), the update's HBM time divided by the world size.
DP_SHARDED = os.environ.get("BRANCHGAN_DP_SHARDED") == "1"


class _Done:
    def wait(self):
        pass

    def is_completed(self):
        return True


# Every device collective of this library is enqueued with async_op=True, i.e. on the process group's OWN stream, and its Work
# handle is kept here until `settle_collectives` has seen it complete.  Why it matters: since torch 2.7 a collective with
# async_op=False runs on the CALLER's current stream, so its end event lives on that stream; ProcessGroupNCCL's watchdog thread
# polls the end events of every collective it has not retired, and on ROCm 7.2 hipEventQuery FAILS (hipErrorCapturedEvent) for an
# event whose stream is capturing - the watchdog turns that into abort().  Round 2's aborted rehearsals (gpurun_out/rs_4.err)
# were the sharded form's all-gather, then synchronous, enqueued from the stream graphs.GraphedTrainStep captures on.  Events on
# RCCL's own stream are not affected by a capture elsewhere: benchmarks/probes/capture_collectives_probe.py polls all twelve
# warm-up handles through the four captures without one failed query (profiles/r3_b_capture_collectives.txt).
_INFLIGHT = []


def _track(work):
    if work is not None:
        if len(_INFLIGHT) >= 64:         # a long run between captures: forget what has completed
            _INFLIGHT[:] = [w for w in _INFLIGHT if not w.is_completed()]
        _INFLIGHT.append(work)
    return work or _Done()


def settle_collectives(timeout_s=60.0):
    """Returns once every collective this library enqueued has completed on the device (a condition, not a delay); called
    before a stream capture begins.  The handles are dropped afterwards."""
    import time
    deadline = time.monotonic() + timeout_s
    while _INFLIGHT:
        if _INFLIGHT[-1].is_completed():
            _INFLIGHT.pop()
            continue
        if time.monotonic() > deadline:
            raise RuntimeError(f"settle_collectives: {len(_INFLIGHT)} collective(s) still running after {timeout_s:.0f} s")
        time.sleep(0.0005)


def _through_host(t, group):
    """gloo with GPU tensors (several ranks rehearsing on ONE GPU: tests/test_gpu_dp.py) goes through host copies made HERE:
    gloo's own device path stalls for good on transfers of a few MB between two processes that share a device
    (benchmarks/probes/gloo_cuda_collectives.py: 12 all-gathers of 1 MB pass, the first of 4 MB never returns)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_sum(t, group=None, async_op=True):
    if _through_host(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return _Done()
    if not async_op and t.is_cuda:
        raise ValueError("all_reduce_sum: device collectives are enqueued with async_op=True only (see _INFLIGHT)")
    return _track(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op))


def reduce_scatter_sum(out, inp, group=None, async_op=True):
    if _through_host(inp, group):
        h, o = inp.cpu(), torch.empty(out.shape, dtype=out.dtype)
        dist.reduce_scatter_tensor(o, h, op=dist.ReduceOp.SUM, group=group)
        out.copy_(o)
        return _Done()
    if not async_op and inp.is_cuda:
        raise ValueError("reduce_scatter_sum: device collectives are enqueued with async_op=True only (see _INFLIGHT)")
    return _track(dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=group, async_op=async_op))


def all_gather_slices(full, mine, group=None):
    """full = the ranks' `mine` slices one after the other.  Enqueued like the gradient collectives (async_op + wait on the
    calling stream), `mine` a buffer that outlives the call."""
    if _through_host(full, group):
        o = torch.empty(full.shape, dtype=full.dtype)
        dist.all_gather_into_tensor(o, mine.cpu(), group=group)
        full.copy_(o)
        return
    _track(dist.all_gather_into_tensor(full, mine, group=group, async_op=True)).wait()


class GradArena:
    """Flat gradient storage of one model + bucketed all-reduce(sum)/world over the process group (or, sharded, one
    reduce-scatter)."""

    def __init__(self, module, bucket_bytes=64 << 20, group=None, sharded=None):
        self.module, self.bucket_bytes, self.group = module, bucket_bytes, group
        self.arena = None
        self.params, self.offsets = [], []
        self.stream = None
        self._work = []
        self.sharded = DP_SHARDED if sharded is None else sharded
        self.shard = None            # (sharded, world > 1) this rank's slice of the summed gradients, filled by reduce_async

    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def sharded_active(self):
        """Sharded AND more than one rank (or the one-rank RCCL rehearsal, which runs the same collectives with world 1)."""
        return self.sharded and dist.is_initialized() and (self.world() > 1 or REHEARSE_RCCL)

    def shard_range(self):
        """[lo, hi) of the arena this rank steps; the whole arena unless sharded over more than one rank."""
        n, w = self.arena.numel(), self.world()
        if not self.sharded_active():
            return 0, n
        r = dist.get_rank(self.group)
        return r * (n // w), (r + 1) * (n // w)

    def _build(self):
        self.params = [p for p in self.module.parameters() if p.requires_grad and p.grad is not None]
        self.offsets, n = [], 0
        for p in self.params:                 # every parameter starts on a 32-byte boundary (16-byte vectors in fp32 and bf16)
            self.offsets.append(n)
            n += (p.numel() + 7) & ~7
        if self.sharded:                      # equal slices that start on 32-byte boundaries
            step = 8 * self.world()
            n = (n + step - 1) // step * step
        dev = self.params[0].device
        self.arena = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, off in zip(self.params, self.offsets):
            view = self.arena[off:off + p.numel()].view_as(p)
            view.copy_(p.grad)
            p.grad = view
        if dev.type == "cuda":
            self.stream = torch.cuda.Stream(device=dev)

    def zero(self):
        if self.arena is None:
            for p in self.module.parameters():
                p.grad = None
        else:
            self.arena.zero_()

    def nbytes(self):
        return 0 if self.arena is None else self.arena.numel() * 4

    def reduce_async(self, average=True):
        """Call after backward.  Launches the bucketed all-reduce; `wait()` before the optimizer step.  average=False leaves
        the SUM over ranks in the arena (an optimizer that folds 1/world into its update, optim.ArenaAdam)."""
        if self.arena is None:
            self._build()
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1 and not (REHEARSE_RCCL and dist.is_initialized()):
            return
        if self.sharded_active():
            lo, hi = self.shard_range()
            if self.shard is None:
                self.shard = torch.zeros(hi - lo, dtype=torch.float32, device=self.arena.device)
            if self.stream is not None:
                self.stream.wait_stream(torch.cuda.current_stream())
            with (torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()):
                if average:
                    self.arena.div_(world)
                self._work.append(reduce_scatter_sum(self.shard, self.arena, self.group))
            return
        per = max(1, self.bucket_bytes // 4)
        buckets = [self.arena[i:i + per] for i in range(0, self.arena.numel(), per)]
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                for b in buckets:
                    if average:
                        b.div_(world)
                    self._work.append(all_reduce_sum(b, self.group))
        else:
            for b in buckets:
                if average:
                    b.div_(world)
                self._work.append(all_reduce_sum(b, self.group))

    def wait(self):
        for w in self._work:
            w.wait()
        self._work = []
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)


class OptimizerWrapper:
    """Training/Optimizers.py:50-92.  backward every micro-step; optimizer step when stepCounter % gradAccum == 0 (so the
    very first micro-step steps, quirk 11).  Under data parallelism the gradients are reduced once, right before that step."""

    def __init__(self, model, optimizer, gradAccum=1, lossKey='loss', scheduler=None, arena=None):
        self.model, self.optimizer, self.gradAccum, self.lossKey, self.scheduler = model, optimizer, gradAccum, lossKey, scheduler
        self.arena = arena if arena is not None else GradArena(model)
        if hasattr(optimizer, "bind"):             # optim.ArenaAdam lays its own arenas out next to the gradient arena
            optimizer.bind(self.arena)
        elif self.arena.sharded and dist.is_initialized() and dist.get_world_size() > 1:
            raise ValueError("BRANCHGAN_DP_SHARDED=1 needs the arena optimizer (make_optimizer(..., 'adam') on the GPU): "
                             f"{type(optimizer).__name__} steps whole parameters")

    def reduce_async(self):
        """Starts the gradient all-reduce of an optimizer step.  The mean over ranks is either taken here or, for an
        optimizer that says so, folded into its update."""
        if self.arena.arena is None:
            self.arena._build()
        if hasattr(self.optimizer, "ensure_built"):
            self.optimizer.ensure_built()          # arenas laid out on the caller's stream, before any side-stream work
        folds = getattr(self.optimizer, "folds_grad_scale", False)
        if folds:
            world = dist.get_world_size(self.arena.group) if dist.is_initialized() else 1
            self.optimizer.grad_scale = 1.0 / world
        self.arena.reduce_async(average=not folds)

    def _step_and_zero(self):
        self.optimizer.step()
        if not getattr(self.optimizer, "zeroes_grads", False):
            self.arena.zero()

    def backward(self, losses, stepCounter):
        losses[self.lossKey].backward()
        if stepCounter % self.gradAccum == 0:
            self.reduce_async()
            return True
        return False

    def finish(self, stepped):
        if stepped:
            self.arena.wait()
            self._step_and_zero()
            if self.scheduler is not None:
                self.scheduler.step()

    def finish_async(self, stepped):
        """`finish` on the arena's side stream: the all-reduce wait, the optimizer step and the zeroing of this model's
        gradients run behind whatever the caller's stream does next (the OTHER model's backward).  `join()` before anything
        on the caller's stream touches this model's parameters again.  Without a GPU it is plain `finish`."""
        side = self.arena.stream
        if side is None or not stepped:
            return self.finish(stepped)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for w in self.arena._work:
                w.wait()
            self.arena._work = []
            self._step_and_zero()
        if self.scheduler is not None:
            self.scheduler.step()

    def join(self):
        if self.arena.stream is not None:
            torch.cuda.current_stream().wait_stream(self.arena.stream)

    def performUpdateStep(self, losses, stepCounter):
        self.finish(self.backward(losses, stepCounter))


def make_optimizer(params, name='adam', lr=1e-4, **kw):
    """Optimizers.py:6-16 (NAME_TO_OPTIM)."""
    table = {'adam': torch.optim.Adam, 'adagrad': torch.optim.Adagrad, 'sgd': torch.optim.SGD, 'rmsprop': torch.optim.RMSprop}
    params = list(params)
    if name.lower() == 'adam' and params and params[0].is_cuda and all(isinstance(p, torch.Tensor) for p in params) \
            and set(kw) <= {'betas', 'eps', 'weight_decay'} and os.environ.get("BRANCHGAN_ARENA_ADAM", "1") != "0":
        from .optim import ArenaAdam
        return ArenaAdam(params, lr=lr, **kw)     # one streaming launch per step: update + bf16 weight copies + zeroed gradients
    if name.lower() == 'adam' and 'fused' not in kw and params and params[0].is_cuda:
        kw['fused'] = True        # one multi-tensor kernel per step instead of a chain of foreach launches
    return table[name.lower()](params, lr=lr, **kw)


def train_step(model, inputIDs, sampleConf, generatorOptim, discriminatorOptim, stepCounter, numUpdateSamples=9999,
               reinforcementLearningConfig=None, updateGen=True, updateDisc=True, **kwargs):
    """The body of efficentTrainingLoop (TrainingLoop.py:141-152): one forward of the whole GAN step, generator backward,
    discriminator backward, gradient reduction, both optimizer steps."""
    discLosses, genLosses = model.trainForwardPassNoMetrics(inputIDs, sampleConf, numUpdateSamples, collectTreeDepthData=False,
                                                            reinforcementLearningConfig=reinforcementLearningConfig, **kwargs)
    g_step = generatorOptim.backward(genLosses, stepCounter) if updateGen else False
    generatorOptim.finish_async(g_step)          # G's all-reduce + Adam run on a side stream behind D's backward
    d_step = discriminatorOptim.backward(discLosses, stepCounter) if updateDisc else False
    discriminatorOptim.finish(d_step)
    generatorOptim.join()
    return discLosses, genLosses


def _appendToCollectedData(collected, new, ignoreTrackingKeys=()):
    """TrainingLoop.py:9-22: lists are extended, ints and 0-d tensors appended as Python numbers, other tensors as numpy."""
    for k, v in new.items():
        if k in ignoreTrackingKeys:
            continue
        slot = collected.setdefault(k, [])
        if isinstance(v, list):
            slot.extend(v)
        elif isinstance(v, (int, float)):
            slot.append(v)
        elif v.dim() >= 1:
            slot.append(v.detach().float().cpu().numpy() if v.is_floating_point() else v.detach().cpu().numpy())
        else:
            slot.append(v.item())


def forwardPass(model, inData, sampleConf, numUpdateSamples, stepCounter=None, collectedGenData=None, collectedDiscData=None,
                updateGenerator=False, updateDiscriminator=False, generatorOptim=None, discriminatorOptim=None, ignoreTrackingKeys=()):
    """TrainingLoop.py:25-52: one forward pass with metrics; losses are merged into the metric dicts (metrics win on a key
    clash) and both are appended to the running collections."""
    discLosses, genLosses, discMetrics, genMetrics = model.trainForwardPass(inData, sampleConf, numUpdateSamples, collectData=True)
    if updateGenerator:
        generatorOptim.performUpdateStep(genLosses, stepCounter)
    if updateDiscriminator:
        discriminatorOptim.performUpdateStep(discLosses, stepCounter)
    for origin, target in ((discLosses, discMetrics), (genLosses, genMetrics)):
        for k, v in origin.items():
            target.setdefault(k, v)
    collectedDiscData = {} if collectedDiscData is None else collectedDiscData
    collectedGenData = {} if collectedGenData is None else collectedGenData
    _appendToCollectedData(collectedDiscData, discMetrics, ignoreTrackingKeys)
    _appendToCollectedData(collectedGenData, genMetrics, ignoreTrackingKeys)
    return collectedDiscData, collectedGenData


def validateModel(model, testData, samplingStrategies={}, displayProgress=True):
    """TrainingLoop.py:83-104: every named sampling strategy over the whole test set, without gradients.
    -> {name: (collectedDiscData, collectedGenData)}."""
    if testData is None:
        return {}
    results = {}
    with torch.no_grad():
        for name, conf in samplingStrategies.items():
            discData, genData = {}, {}
            for bIDs in testData:
                discData, genData = forwardPass(model, bIDs, conf.getNextSampleConf(), numUpdateSamples=1, collectedGenData=genData,
                                                collectedDiscData=discData)
                results[name] = (discData, genData)
    return results


def logResults(validationResult, genData, discData, trainDataset, epochCounter, batchCounter, generatorOptim, discriminatorOptim,
               trainingConf, sink=print):
    """TrainingLoop.py:55-79: means of everything collected since the last evaluation, handed to `sink` one dict at a time
    (the reference prints; a logger can be plugged in).  Returns the dicts."""
    import numpy as np
    out = []

    def emit(data, prefix):
        row = {"{}-{}".format(prefix, k): np.mean(v) for k, v in data.items()}
        out.append(row)
        sink(row)

    for name, (valDisc, valGen) in validationResult.items():
        emit(valDisc, 'Val-{}_Discriminator'.format(name))
        emit(valGen, 'Val-{}_Generator'.format(name))
    if getattr(discriminatorOptim, 'scheduler', None) is not None:
        discData['lr'] = discriminatorOptim.scheduler.get_last_lr()[0]
    if getattr(generatorOptim, 'scheduler', None) is not None:
        discData['genData'] = generatorOptim.scheduler.get_last_lr()[0]          # the reference files the generator's lr here
    if trainingConf is not None and getattr(trainingConf, 'depthSchedule', None) is not None:
        genData['generationDepth'] = trainingConf.depthSchedule.currentDepth
    emit({'Epoch': epochCounter + (batchCounter / len(trainDataset))}, '')
    emit(discData, 'Discriminator')
    emit(genData, 'Generator')
    return out


def efficentTrainingLoop(model, trainDataset, testData, numEpochs, generatorOptim, discriminatorOptim, trainingSamplingConf,
                         numUpdateSamples, updatesPerEvaluation, validationSamplingConf, doLogData=True, displayProgression=False,
                         saveModel=True, saveName='SavedModel', genDiscUpdateSchedule=None, datasetConf=None, trainingConf=None,
                         **kwargs):
    """TrainingLoop.py:107-192, same arguments: epochs over an iterable of (B, S) id batches; per micro-step the update
    schedule decides which model steps, the depth schedule how deep the tree is; every `updatesPerEvaluation` steps (first
    after step 1) the validation strategies run, results are logged and a 'Current' checkpoint is written; further
    checkpoints by `saveFreq` / `epochSaveFreq` (schedules.ModelSaveFrequencyTracker).  `mainSavePath` (extra keyword) roots
    the checkpoints.  Returns the log rows.

    Extra keywords: `useGraphs=True` runs every micro-step as graphs.GraphedTrainStep replays (one captured step per distinct
    sampling config / depth, built on first use without training the model) - the path bench.py measures; `ampDtype`
    (torch.bfloat16 or None) is the autocast dtype of those steps."""
    from . import checkpoint, schedules
    sched = genDiscUpdateSchedule if genDiscUpdateSchedule is not None else schedules.AlwaysUpdateSchedule()
    # the save points are worked out on EVERY rank (gather_state is a collective); only ranks with saveModel write
    tracker = schedules.ModelSaveFrequencyTracker(kwargs.pop('saveFreq', None), len(trainDataset), enabled=saveModel or DP_SHARDED,
                                                  epochSaveFrequency=kwargs.pop('epochSaveFreq', None))
    save_root, sink = kwargs.pop('mainSavePath', ''), kwargs.pop('logSink', print)
    use_graphs, amp_dtype, graphed = kwargs.pop('useGraphs', False), kwargs.pop('ampDtype', None), {}

    def conf_key(conf):
        return (conf.numSampleSequences, conf.includeAllMultiStepSamplesForDisc,
                tuple((r.numAlterations, getattr(r.samplingStrategy, "name", r.samplingStrategy), repr(sorted((r.samplingParams or {}).items())),
                       r.continueFromRandom) for r in conf.rules))

    def join_graphs():
        for gs in graphed.values():
            gs.join()
    depthSchedule = getattr(trainingConf, 'depthSchedule', None)
    rlConf = getattr(trainingConf, 'reinforcementLearningConfig', None)

    def gather_state():
        for o in (generatorOptim, discriminatorOptim):          # sharded data parallelism: every rank holds a slice of the moments
            if hasattr(o.optimizer, "gather_state"):
                o.optimizer.gather_state()

    def save(marker):
        checkpoint.saveGannSetup(model, saveName, marker, discriminatorOptim, generatorOptim, mainSavePath=save_root, datasetConf=datasetConf,
                                 trainingConf=trainingConf, saveGen=sched.saveGen, saveDisc=sched.saveDisc,
                                 saveOptimizers=getattr(trainingConf, 'saveOptimizers', True))

    stepCounter, epochCounter, log = 0, 0, []
    for _ in range(int(numEpochs)):
        discData, genData = {}, {}
        tracker.initEpoch()
        for i, bIDs in enumerate(trainDataset):
            updateDisc, updateGen = sched.getGenDiscUpdates(stepCounter, epochCounter, generatorOptim, discriminatorOptim)
            conf = trainingSamplingConf.getNextSampleConf()
            if depthSchedule is not None:
                depthSchedule.updateSchedule(epochCounter, i, stepCounter)
                conf = depthSchedule.setSampleConfDepth(conf)
            if use_graphs:
                key = conf_key(conf)
                if key not in graphed:
                    from .graphs import GraphedTrainStep
                    join_graphs()
                    graphed[key] = GraphedTrainStep(model, conf, generatorOptim, discriminatorOptim, bIDs, numUpdateSamples, rlConf, amp_dtype)
                else:
                    for k, gs in graphed.items():              # a deferred discriminator step of another captured config
                        if k != key:
                            gs.join()
                discLosses, genLosses = graphed[key].step(bIDs, stepCounter, updateGen, updateDisc, sampleConf=conf)
            else:
                discLosses, genLosses = train_step(model, bIDs, conf, generatorOptim, discriminatorOptim, stepCounter, numUpdateSamples, rlConf,
                                                   updateGen=updateGen, updateDisc=updateDisc, **kwargs)
            _appendToCollectedData(discData, discLosses)
            _appendToCollectedData(genData, genLosses)
            stepCounter += 1
            if stepCounter % updatesPerEvaluation == 1:             # == 1: an early first evaluation, as in the reference
                join_graphs()
                validationResult = validateModel(model, testData, samplingStrategies=validationSamplingConf, displayProgress=displayProgression)
                if doLogData:
                    log.extend(logResults(validationResult, genData, discData, trainDataset, epochCounter, i, generatorOptim, discriminatorOptim,
                                          trainingConf, sink=sink))
                discData, genData = {}, {}
                gather_state()
                if saveModel:
                    save('Current')
            due, name = tracker.shouldModelBeSaved(i, epochCounter)
            if due:
                join_graphs()
                gather_state()
                if saveModel:
                    save(name)
        epochCounter += 1
        join_graphs()
        if tracker.epochSaveFrequency and tracker.shouldModelBeSavedAtThisEpoch(epochCounter):
            gather_state()
            if saveModel:
                save("Epoch-{}".format(epochCounter))
    return log
