"""Discriminator and generator losses — host side over the fused HIP loss kernels.

Mirrors Models/Losses/{DiscriminatorLoss,GeneratorLosses,ReinforceLoss,AdvantageCalculations}.py (reference): the same
function names, argument order and returned dict keys, computed by bg_disc_loss_* / bg_rl_loss_* (one launch each way)
and bg_softmax_stats_* (MLE) instead of ~150 elementwise ATen launches and a (B, S, V) softmax.
"""
import torch

from . import ops
from .configs import IncorrectValueHeadsPolicy, ReinforcementLabelMode


def _real_columns(data, fallback):
    """The reference pads branch tensors from K to S columns with constants; the kernels want the K real ones.
    `altInputPreds` (N,B,K,1) is the one tensor the reference never pads (GanSetup.py:94-101,202-207)."""
    t = data.get('altInputPreds')
    return t.shape[2] if t is not None else fallback


def calcDiscriminatorLoss(gannSetup, inputScores, alterationScores, numNegativeUpdateSamples, discDepthMask, discDepthWeights,
                          collectTreeDepthData=False, forwardPassData=None, numBranchPoints=None):
    """DiscriminatorLoss.py:37-114.  alterationScores may be (B,N,K) or the reference's zero-padded (B,N,S)."""
    if discDepthWeights is not None:
        raise NotImplementedError("discDepthWeights: the schedule classes do not exist in the reference (quirk 14)")
    data = forwardPassData['DiscValueHeads']
    rootLinks, alterationLinks = forwardPassData['SubNodeStructure']
    K = numBranchPoints if numBranchPoints is not None else _real_columns(data, alterationScores.shape[-1])
    alt = alterationScores[:, :, :K]
    altAlt = data['altAlterationsPreds'].squeeze(-1)[:, :, :K]
    mask = discDepthMask.to(inputScores.device)[:, :K]
    out = ops.disc_losses(inputScores, alt, data['txtInputPreds'].squeeze(-1), data['txtAlterationsPreds'].squeeze(-1), altAlt,
                          mask, rootLinks, alterationLinks, numNegativeUpdateSamples)
    out = out.unbind(0)       # (one unbind: its backward is one stack; out[i] five times was a zeros + copy + add per element)
    info = {'numAlterationSamples': min(numNegativeUpdateSamples, alt.shape[1]), 'ValueHeadLoss': out[4],
            'ValueHead-InputLoss': out[2].detach(), 'ValueHead-AltLoss': out[3].detach()}
    if collectTreeDepthData:
        with torch.no_grad():                                                   # diagnostics (:76-112), tiny tensors
            B = alt.shape[0]
            n = info['numAlterationSamples']
            per = -torch.clamp(torch.log(1 - alt), min=-100) * mask.unsqueeze(0)
            per = torch.nn.functional.pad(per, (0, inputScores.shape[1] - K))
            denom = mask.sum(-1) * B
            acc = ((alt[:, :n] < 0.5) * mask[:n]).sum((0, 2)).float() / denom[:n]
            pred = alt[:, :n].sum((0, 2)) / denom[:n]
            groups = {}
            for i, (depth, rand) in enumerate(forwardPassData['TreeNodeInfo']):
                groups.setdefault((depth, rand), []).append(i)
            for (depth, rand), idx in groups.items():
                info['Depth-{}-{}'.format(depth, 'RandLoss' if rand else 'Loss')] = torch.stack([per[:, i] for i in idx]).mean()
            for depth in sorted({d for d, _ in groups}):
                idx = [i for (d, _), ii in groups.items() if d == depth for i in ii]
                info['Depth-{}-Acc'.format(depth)] = torch.stack([acc[i] for i in idx]).mean()
                info['Depth-{}-Pred'.format(depth)] = torch.stack([pred[i] for i in idx]).mean()
    return out[0], out[1], info


def reinforceLosses(labelLogProbs, alterationLogProbs, discInputScores, discAlterationScores, txtInputPreds, txtAlterationsPreds,
                    reinforcementLearningConfig, depthMask=None, probs_input=False, debugStats=None):
    """_reinforceLoss (GeneratorLosses.py:84-131) -> dict input / alterations / combined / unweightedCombined.  debugStats: a
    dict that receives the twenty entries of _calculateDebuggingStats (:34-81), 0-dim views of one device tensor."""
    cfg = reinforcementLearningConfig
    if cfg.teacherForcing:
        discInputScores = torch.ones_like(discInputScores)
    w = cfg.advantageSectionWeights
    K = alterationLogProbs.shape[-1]
    out = ops.rl_losses(labelLogProbs, alterationLogProbs, discInputScores, discAlterationScores[:, :, :K],
                        txtInputPreds.squeeze(-1), txtAlterationsPreds.squeeze(-1),
                        None if depthMask is None else depthMask[:, :K], w.A, w.B, w.C,
                        cfg.inputIncorrectValueHeadsPolicy == IncorrectValueHeadsPolicy.IGNORE_SAMPLE,
                        cfg.alterationIncorrectValueHeadsPolicy == IncorrectValueHeadsPolicy.IGNORE_SAMPLE, probs_input,
                        diagnostics=debugStats is not None)
    if debugStats is not None:
        out, diag = out
        debugStats.update(zip(ops.RL_DIAG_KEYS, diag.unbind(0)))
    inputLoss, alterationLoss = out.unbind(0)
    N = discAlterationScores.shape[1]
    return {'input': inputLoss, 'alterations': alterationLoss, 'combined': (inputLoss + alterationLoss) / 2,
            'unweightedCombined': (inputLoss + alterationLoss) / (1 + N)}


# ---- the reference's own function boundaries (the train step runs the fused kernels above) --------------------------------
def dualHeadAdvantages(discInputScores, alterationScores, inputValueHeadPred, alterationValueHeadPred, advantageSectionWeights,
                       inputIncorrectValueHeadsPolicy, alterationIncorrectValueHeadsPolicy, minZoneCLoss=0.1, epsilon=0.00001,
                       min_denominator_value=1e-5):
    """AdvantageCalculations.py:5-46, same arguments and return structure:
    (inputAdvantages (B,1,S), alterationAdvantages (B,N,S), (zoneA, zoneB, zoneC (B,1+N,S), advantageCorrectMask (B,S),
    inputCorrectMask (B,S))), all float32, no gradient.  One launch of bg_dual_head_advantages."""
    w = advantageSectionWeights
    ignore = lambda policy: IncorrectValueHeadsPolicy(int(getattr(policy, "value", policy))) == IncorrectValueHeadsPolicy.IGNORE_SAMPLE
    adv, zA, zB, zC, ok, in_ok = ops.dual_head_advantages(discInputScores, alterationScores, inputValueHeadPred.squeeze(-1),
                                                          alterationValueHeadPred.squeeze(-1), w.A, w.B, w.C,
                                                          ignore(inputIncorrectValueHeadsPolicy), ignore(alterationIncorrectValueHeadsPolicy),
                                                          epsilon, min_denominator_value)
    return adv[:, :1], adv[:, 1:], (zA, zB, zC, ok, in_ok)


def calculateMinimumProbLimit(advantages, policyProbs, negPushProbLimit=0.1):
    """ReinforceLoss.py:11-18 (a mask; reinforceLoss applies it inside the kernel)."""
    rewardMask = torch.greater_equal(advantages, 0).float()
    inverted = 1 - rewardMask
    return rewardMask + torch.where(policyProbs * inverted >= negPushProbLimit, inverted, rewardMask)


def calculateAdvantageMask(advantages, policyProbs, powerWeight=0.05):
    """ReinforceLoss.py:4-8 (unused by the reference's loss; kept for the import surface)."""
    rewardMask = torch.greater_equal(advantages, 0).float()
    return (policyProbs * (1 - rewardMask) + rewardMask) ** powerWeight


def reinforceLoss(policyProbs, advantages, useNegativeEntropyMask, useMinProbLimit, negPushProbLimit=1e-8, eps=1e-6):
    """ReinforceLoss.py:45-52: -log(clamp(p)) * advantages [* calculateMinimumProbLimit]; useNegativeEntropyMask is unused there too."""
    return ops.reinforce_terms(policyProbs, advantages, None, useMinProbLimit, negPushProbLimit, eps)


def mirroredNegativeReinforceLoss(policyProbs, advantages, maskZoneC):
    """ReinforceLoss.py:28-42: the plain term outside zone C, -log(1 - p) * advantages inside."""
    return ops.reinforce_terms(policyProbs, advantages, maskZoneC)


def mleLossFromLabelLogProbs(labelLogProbs):
    """_mleLossFromLogits (GeneratorLosses.py:133-137): CE = -mean log p(label); labelLogProbs comes from bg_softmax_stats."""
    return -labelLogProbs.mean()


def calcGeneratorLoss(gannSetup, inputIDs, generatorLogits, generatorProbs, discInputScores, discAlterationScores, forwardPassData,
                      reinforcementLearningConfig, depthMask=None):
    """GeneratorLosses.py:146-180.  Uses forwardPassData['LabelLogProbs'] / ['AlterationLogProbs'] when the tree engine put
    them there; otherwise derives them from generatorLogits with one fused pass (generatorProbs is never read)."""
    labelLogProbs = forwardPassData.get('LabelLogProbs')
    if labelLogProbs is None:
        nxt = torch.cat((inputIDs[:, 1:], torch.full_like(inputIDs[:, :1], -1)), dim=1)
        labelLogProbs = ops.softmax_stats(generatorLogits, nxt.unsqueeze(-1))[1][:, :-1, 0]
    genLosses = {'loss-MLE': mleLossFromLabelLogProbs(labelLogProbs)}
    if gannSetup.generatorRlLossWeight == 0:
        genLosses['loss'] = genLosses['loss-MLE'] * gannSetup.generatorMleLossWeight
        return genLosses
    altLogProbs = forwardPassData.get('AlterationLogProbs')
    probs_input = False
    data = forwardPassData['DiscValueHeads']
    if altLogProbs is None:                      # reference-style dict: probabilities, 1e-5 padded to S columns
        K = _real_columns(data, forwardPassData['AlterationProbs'].shape[-1])
        altLogProbs, labelLogProbs, probs_input = forwardPassData['AlterationProbs'][:, :, :K], labelLogProbs.exp(), True
    rlLossInfo = {}
    rl = reinforceLosses(labelLogProbs, altLogProbs, discInputScores.detach(), discAlterationScores.detach(),
                         data['txtInputPreds'].detach(), data['txtAlterationsPreds'].detach(), reinforcementLearningConfig,
                         depthMask=depthMask, probs_input=probs_input, debugStats=rlLossInfo)
    genLosses.update(rlLossInfo)                 # the reference's key order: MLE, the debugging stats, loss-RL-*, loss-RL, loss
    for k, v in rl.items():
        genLosses['loss-RL-{}'.format(k)] = v
    mode = reinforcementLearningConfig.labelMode
    rlLoss = {ReinforcementLabelMode.ALTERATIONS_ONLY: rl['alterations'], ReinforcementLabelMode.INPUT_ONLY: rl['input'],
              ReinforcementLabelMode.ALTERATIONS_AND_INPUT: rl['combined'],
              ReinforcementLabelMode.ALTERATIONS_AND_INPUT_UNWEIGHTED: rl['unweightedCombined']}[ReinforcementLabelMode(int(mode))]
    genLosses['loss-RL'] = rlLoss
    genLosses['loss'] = genLosses['loss-MLE'] * gannSetup.generatorMleLossWeight + rlLoss * gannSetup.generatorRlLossWeight
    return genLosses
