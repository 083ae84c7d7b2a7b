"""One Branch-GAN step — host side.  Mirrors Models/GanSetup.py (reference): GannSetup, trainForwardPassNoMetrics and the
loss-dict keys; the padding of branch tensors to S columns is never materialised (the kernels evaluate it)."""
import math

import torch

from . import losses as Losses
from . import models, ops
from . import tree as MultiStepPrediction
from .configs import LMConfig, ReinforcementLearningConfig
from .models import NonResidualDiscriminatorWithDualValueHeads, NonResidualGenerator, strip_reference_buffers


class GannSetup(torch.nn.Module):
    def __init__(self, generatorConfig, discriminatorConfig, mleLossWeight=1, rlLossWeight=1, discPredLossWeight=1,
                 discValueHeadLossWeight=0.25, device='cuda', generatorDevice=None, discriminatorDevice=None,
                 generatorTeacherForcingInput=True, samplingConf=None, **modelCreationConfigs):
        super().__init__()
        self.device = device
        self.generatorDevice = generatorDevice if generatorDevice is not None else device
        self.discriminatorDevice = discriminatorDevice if discriminatorDevice is not None else device
        if str(self.generatorDevice) != str(self.discriminatorDevice):
            raise NotImplementedError("generator/discriminator on two devices is out of scope (288 GB per MI355X; SURVEY §2 #13)")
        self.generator = NonResidualGenerator(generatorConfig)
        self.discriminator = NonResidualDiscriminatorWithDualValueHeads(discriminatorConfig)
        self.generatorConfig, self.discriminatorConfig = LMConfig.from_any(generatorConfig), LMConfig.from_any(discriminatorConfig)
        self.generatorMleLossWeight, self.generatorRlLossWeight = mleLossWeight, rlLossWeight
        self.discValueHeadLossWeight, self.discPredLossWeight = discValueHeadLossWeight, discPredLossWeight
        self.samplingConf = samplingConf
        self.generatorTeacherForcingInput = generatorTeacherForcingInput
        self.noise_fn = None          # injection point for deterministic exponential noise (tests)

    def load_state_dict(self, state_dict, strict=True, assign=False):
        """Loads a reference `GannSetup` / Generator.pt / Discriminator.pt state dict (same key names)."""
        return super().load_state_dict(strip_reference_buffers(state_dict), strict=strict, assign=assign)

    def createDiscAlterationLossWeights(self, inputIDs, alternationNodes, K):
        """GanSetup.py:75-92 on the K real columns: 1 where sparseIndex + depth <= S - 1."""
        sp = alternationNodes[0].sparseIndex
        if all(n.sparseIndex is sp for n in alternationNodes) and sp.device == inputIDs.device \
                and max(n.depth for n in alternationNodes) < MultiStepPrediction._POS_DEPTHS:
            depths = MultiStepPrediction._depths_tensor(tuple(n.depth for n in alternationNodes), sp.device)
            pos = MultiStepPrediction._position_table(sp, sp.device).index_select(0, depths)      # one gather for all nodes
        else:
            pos = torch.stack([n.sparseIndex + n.depth for n in alternationNodes]).to(inputIDs.device)
        return torch.less_equal(pos, inputIDs.shape[-1] - 1).float()

    def _generatorForward(self, inputIDs, samplingConf):
        """The generator's half of the forward pass (GanSetup.py:140-150 multiStepGenForward): root pass, branch sampling and
        the branch log-probabilities.  It touches no discriminator weight, so a caller may run it while the discriminator's
        optimizer step of the previous train step is still in flight (graphs.GraphedTrainStep does)."""
        if models.dropout_active(self.generator) or models.dropout_active(self.discriminator) or \
                any(r.samplingParams.get('dropoutRate') for r in samplingConf.rules):
            # The attention-dropout masks are a function of a device-side step counter: advanced by a launch of its own at the
            # head of the step (this method is the first thing every step runs, also when its parts are captured as separate
            # graphs), so that a captured step draws fresh masks on every replay and backward sees the forward's masks.
            ops.dropout_advance(self.generatorDevice)
        sparseIndex = samplingConf.getSparseIndex(inputIDs, self.generatorDevice)
        return MultiStepPrediction.multiStepGenForward(
            self.generator, inputIDs, samplingConf.rules, self.generatorDevice, sparseIndex=sparseIndex, clearTreePast=True,
            noise_fn=self.noise_fn)

    def _multiStepForwardPass(self, inputIDs, samplingConf, collectTreeDepthData=True, generatorState=None):
        root, firstPred, probs, firstMaskedProbs = generatorState if generatorState is not None else self._generatorForward(inputIDs, samplingConf)
        nodes = root.getSubNodes() if samplingConf.includeAllMultiStepSamplesForDisc else root.getLeaves()
        if samplingConf.discDepthLossSchedule is not None or samplingConf.discpDepthLossWeight is not None:
            raise NotImplementedError("depth loss schedules are undefined in the reference (quirk 14)")
        K = nodes[0].ids.shape[1]
        depthMask = self.createDiscAlterationLossWeights(inputIDs, nodes, K)                     # (N, K)
        forwardData = {'TreeNodeInfo': [(n.depth, n.isRandom) for n in nodes]} if collectTreeDepthData else {}
        forwardData['AlterationLogProbs'] = MultiStepPrediction.restack([n.logProbs for n in nodes], 1)   # (B, N, K), autograd-connected
        forwardData['LabelLogProbs'] = root.labelLogProbs
        forwardData['NumBranchPoints'] = K
        MultiStepPrediction.multiStepDiscForward(self.discriminator, root, self.discriminatorDevice, dualValueHeads=True,
                                                 clearTreePast=False)
        forwardData['DiscValueHeads'] = {
            'txtInputPreds': root.valueHeadData['InputValue'], 'txtAlterationsPreds': root.valueHeadData['AlterationsValue'],
            'altInputPreds': MultiStepPrediction.restack([n.valueHeadData['InputValue'] for n in nodes], 0),
            'altAlterationsPreds': MultiStepPrediction.restack([n.valueHeadData['AlterationsValue'] for n in nodes], 0)}
        forwardData['SubNodeStructure'] = MultiStepPrediction.getSubNodeListTreeStructure(root, nodes)
        discInputPred = root.discPreds
        discAlterationScores = MultiStepPrediction.restack([n.discPreds.prediction for n in nodes], 1)   # (B, N, K)
        MultiStepPrediction.clearTree(root, clearAll=True)
        return inputIDs, (firstPred.logits, probs, firstMaskedProbs), (discInputPred, discAlterationScores, depthMask, None), forwardData

    def forwardPass(self, inputIDs, samplingConf, **kwargs):
        samplingConf = self.samplingConf if samplingConf is None else samplingConf
        if kwargs.get('mixedPrecision', False):
            raise NotImplementedError("fp16 autocast + GradScaler is replaced by a bf16 model dtype (SURVEY §2 #15)")
        return self._multiStepForwardPass(inputIDs, samplingConf, generatorState=kwargs.get('generatorState'))

    def _internalTrainForwardPass(self, inputIDs, samplingConf, numNegativeUpdateSamples=1, **kwargs):
        samplingConf = self.samplingConf if samplingConf is None else samplingConf
        inputIDs, genData, discData, fpd = self.forwardPass(inputIDs, samplingConf, **kwargs)
        genLogits, genProbs, _ = genData
        discInputPreds, discAlterationScores, depthMask, discDepthWeights = discData
        K = fpd['NumBranchPoints']
        discInputLoss, discAlterationLoss, info = Losses.calcDiscriminatorLoss(
            self, discInputPreds.prediction, discAlterationScores, numNegativeUpdateSamples, depthMask, discDepthWeights,
            collectTreeDepthData=kwargs.get('collectTreeDepthData', False), forwardPassData=fpd, numBranchPoints=K)
        if samplingConf.separateInputAltLoss:
            discPredLoss = (discInputLoss * samplingConf.inputLossWeight + discAlterationLoss) / 2
        else:
            n = info['numAlterationSamples']
            discPredLoss = (discInputLoss * samplingConf.inputLossWeight + discAlterationLoss * n) / (n + 1)
        discLoss = discPredLoss * self.discPredLossWeight + info['ValueHeadLoss'] * self.discValueHeadLossWeight
        discLosses = {'loss': discLoss, 'predLoss': discPredLoss, 'loss-Input': discInputLoss, 'loss-Alterations': discAlterationLoss}
        genLosses = Losses.calcGeneratorLoss(self, inputIDs, genLogits, genProbs, discInputPreds.prediction.detach(),
                                             discAlterationScores.detach(), fpd,
                                             reinforcementLearningConfig=kwargs.get('reinforcementLearningConfig') or ReinforcementLearningConfig(),
                                             depthMask=depthMask)
        discLosses.update(info)
        if kwargs.get('checkNaN', True):                 # GanSetup.py:249-256 (one host sync per step instead of one per key)
            bad = torch.isnan(torch.stack([v.detach().float() for v in genLosses.values() if isinstance(v, torch.Tensor)]))
            if bool(bad.any()):
                keys = [k for k, v in genLosses.items() if isinstance(v, torch.Tensor)]
                raise Exception("Generator loss {} is NaN".format(keys[int(bad.float().argmax())]))
        return discLosses, genLosses, genData, discData, fpd

    def trainForwardPass(self, inputIDs, samplingConf, numNegativeUpdateSamples=1, **kwargs):
        """GanSetup.py:260-285: the forward pass plus the evaluation metrics of both models (branchgan.metrics; the (B, S, V)
        work is two kernels, no sort).  As in the reference the metrics compare text and alteration scores position by
        position, so the sampling config must branch at every position (numSampleSequences=None)."""
        from . import metrics
        discLosses, genLosses, genData, discData, fpd = self._internalTrainForwardPass(inputIDs, samplingConf, numNegativeUpdateSamples, **kwargs)
        _, genProbs, maskedProbs = genData
        discInputPreds, discAlterationScores, _, _ = discData
        discInputScores = discInputPreds.prediction
        if discAlterationScores.shape[-1] != discInputScores.shape[-1]:
            raise ValueError("trainForwardPass: the metrics need a branch at every text position (K == S): evaluate with "
                             "SamplingConfig(numSampleSequences=None); got K={} S={}".format(discAlterationScores.shape[-1], discInputScores.shape[-1]))
        ids = inputIDs.to(self.generatorDevice)
        with torch.no_grad():
            generatorMetrics = metrics.calculateGeneratorMetrics(ids, genProbs, maskedProbs, discInputScores.detach(), discAlterationScores.detach())
            discMetrics = metrics.calcDiscriminatorMetrics(discInputScores.detach(), discAlterationScores.detach(), fpd)
        return discLosses, genLosses, discMetrics, generatorMetrics

    def trainForwardPassNoMetrics(self, inputIDs, samplingConf, numNegativeUpdateSamples=1, **kwargs):
        discLosses, genLosses, *_ = self._internalTrainForwardPass(inputIDs, samplingConf, numNegativeUpdateSamples, **kwargs)
        return discLosses, genLosses
