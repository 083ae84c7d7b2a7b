"""Host-side schedules of a training run: which of the two models steps at a given micro-step, the
learning-rate schedules the reference wires into its optimizer wrappers, and when checkpoints are due.

Reference: Training/GanUpdateSchedule.py:1-100 (update policies; `getGenDiscUpdates` returns `(updateDisc, updateGen)` in THAT
order, TrainingLoop.py:132), Training/Optimizers.py:19-27,124-141 (scheduler construction, FreezeLRSchedule),
Training/Utils.py:120-154 (ModelSaveFrequencyTracker).  Pure Python, no device work: each policy is one predicate over
(step, epoch) instead of a class body per policy; the reference's class names are kept as the public surface.
"""
import numpy as np
from torch.optim.lr_scheduler import LRScheduler


class GenDiscUpdateSchedule:
    """Base: `saveGen` / `saveDisc` tell the checkpoint writer which model this schedule ever changes."""
    saves = (True, True)          # (saveGen, saveDisc)
    label = ''

    def __init__(self, saveGen=None, saveDisc=None):
        self.saveGen = self.saves[0] if saveGen is None else saveGen
        self.saveDisc = self.saves[1] if saveDisc is None else saveDisc

    def decide(self, step, epoch):
        """-> (updateDisc, updateGen)"""
        raise NotImplementedError()

    def getGenDiscUpdates(self, stepCounter, epochCounter, genOptimWraper=None, discOptimWraper=None):
        return self.decide(stepCounter, epochCounter)

    def __repr__(self):
        return self.label


def _policy(name, label, saves, decide, init=None):
    body = {'label': label, 'saves': saves, 'decide': decide, '__doc__': f"GanUpdateSchedule.py `{name}`: {label}"}
    if init is not None:
        body['__init__'] = init
    return type(name, (GenDiscUpdateSchedule,), body)


AlwaysUpdateSchedule = _policy('AlwaysUpdateSchedule', 'Always Update Both', (True, True), lambda self, s, e: (True, True))
NeverUpdateSchedule = _policy('NeverUpdateSchedule', 'Never Update Any', (False, False), lambda self, s, e: (False, False))
OnlyUpdateDisc = _policy('OnlyUpdateDisc', 'Only Update Disc', (False, True), lambda self, s, e: (True, False))
OnlyUpdateGen = _policy('OnlyUpdateGen', 'Only Update Generator', (True, False), lambda self, s, e: (False, True))
AlternateOverEpochsUpdateSchedule = _policy('AlternateOverEpochsUpdateSchedule', 'Alternate between updates between epochs', (True, True),
                                            lambda self, s, e: (False, True) if e % 2 == 0 else (True, False))


def _alt_init(self, gen2diskUpdateMod=2):
    GenDiscUpdateSchedule.__init__(self)
    self.gen2diskUpdateMod = gen2diskUpdateMod


def _slow_init(self, genStartStep):
    GenDiscUpdateSchedule.__init__(self)
    self.genStartStep = genStartStep


AlternateSequence = _policy('AlternateSequence', '', (True, True),
                            lambda self, s, e: (False, True) if s % self.gen2diskUpdateMod == 0 else (True, False), _alt_init)
AlternateSequence.__repr__ = lambda self: 'Alternate Updates. GenUpdateMod: {}'.format(self.gen2diskUpdateMod)
GeneratorSlowStart = _policy('GeneratorSlowStart', '', (True, True),
                             lambda self, s, e: (False, True) if s < self.genStartStep else (True, True), _slow_init)
GeneratorSlowStart.__repr__ = lambda self: 'Generator Slow Start. GenStartStep: {}'.format(self.genStartStep)


class FreezeLRSchedule(LRScheduler):
    """Optimizers.py:124-141: lr = 0 for `zero_steps` scheduler steps, then a linear ramp lr_start -> lr_end over `warmup_steps`,
    then lr_end.  The optimizer's own base lr is ignored (as in the reference)."""

    def __init__(self, optimizer, zero_steps, warmup_steps, lr_start, lr_end, last_epoch=-1, **kwargs):
        self.zero_steps, self.warmup_steps, self.lr_start, self.lr_end = zero_steps, warmup_steps, lr_start, lr_end
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        t = self.last_epoch - self.zero_steps
        if t < 0:
            lr = 0
        elif t < self.warmup_steps:
            lr = self.lr_start + (t / self.warmup_steps) * (self.lr_end - self.lr_start)
        else:
            lr = self.lr_end
        return [lr] * len(self.base_lrs)


def create_scheduler(schedulerType, schedulerArgs, optimizer, gradAccum, stepsPerEpoch, epochs):
    """Optimizers.py:19-27: the scheduler class is called with the user's arguments plus `step_size_down` = the optimizer steps
    of the run that `step_size_up` leaves over (the torch.optim.lr_scheduler.CyclicLR convention; FreezeLRSchedule swallows
    it).  None when no scheduler is configured."""
    if schedulerType is None or schedulerArgs is None:
        return None
    total = int(stepsPerEpoch * epochs / gradAccum)
    up = schedulerArgs.get('step_size_up', total)
    return schedulerType(optimizer, **schedulerArgs, step_size_down=total - up)


class ModelSaveFrequencyTracker:
    """Utils.py:120-154: within an epoch a checkpoint is due every `frequency` x datasetSize batches (named
    'Epoch-<epoch + fraction>', fraction rounded to 4 decimals) while epoch < maxEpoch; at epoch ends every
    `epochSaveFrequency` epochs."""

    def __init__(self, frequency, datasetSize, enabled=True, maxEpoch=2, epochSaveFrequency=1):
        self.frequency, self.datasetSize, self.enabled = frequency, datasetSize, enabled
        self.maxEpoch, self.epochSaveFrequency = maxEpoch, epochSaveFrequency
        self.nextSaveThreshold = None

    def initEpoch(self):
        active = self.enabled and self.frequency is not None
        self.nextSaveThreshold = self.frequency * self.datasetSize if active else None

    def shouldModelBeSaved(self, batchIDx, epoch):
        due = self.enabled and self.nextSaveThreshold is not None and not (self.maxEpoch is not None and epoch >= self.maxEpoch) \
            and batchIDx >= self.nextSaveThreshold
        if not due:
            return False, None
        name = "Epoch-{}".format(epoch + np.around(self.nextSaveThreshold / self.datasetSize, 4))
        self.nextSaveThreshold += self.frequency * self.datasetSize
        return True, name

    def shouldModelBeSavedAtThisEpoch(self, finishedEpoch):
        return finishedEpoch % self.epochSaveFrequency == 0
