"""No-GPU checks of the drop-in boundary added in round 2: the reference's import paths resolve to this build's modules with
the reference's function names and parameter lists, and the oracle restatements of the new pieces equal the vectors captured
from the reference (tests/golden/g10_boundary.npz, written by tests/golden/make_goldens.py g10)."""
import inspect

import numpy as np
import pytest
import torch

from oracle import attention as A, losses as L, sampling as S


def test_reference_import_paths_resolve():
    from BranchPredictions import MultiStepPrediction, SampleBuffer, SamplingStrategies
    from Configs import ReinforceLabelMode, TrainingConfig
    from Models import Discriminator, GanSetup, Generator
    from Models.GPT2BranchAttention import NonResidualAttention, NonResidualDiscriminatorBase
    from Models.Losses import AdvantageCalculations, DiscriminatorLoss, GeneratorLosses, ReinforceLoss
    from Models.NeoXBranchAttention import NeoXNonResidualAttention, NeoXNonResidualBase
    from Training import DepthSchedule, GanUpdateSchedule, Optimizers, TrainingLoop
    import branchgan.sampling
    assert SamplingStrategies is branchgan.sampling             # module globals (SAMPLE_BUFFER) are the real ones
    # (module, name, the reference's parameter names in order)
    surface = [
        (SamplingStrategies, "sampleFromGeneratorDistribution", ["probs", "numSamples", "inputIDs", "samplingTechnique", "blockSamplingOfLabels", "includeFinalTimeStep"]),
        (SamplingStrategies, "sampleWithSamplingRule", ["probs", "inputIDs", "samplingRule", "blockSamplingOfLabels", "includeFinalTimeStep"]),
        (SamplingStrategies, "apply_temperature", ["tensor", "temperature"]),
        (SamplingStrategies, "apply_noise", ["tensor", "noise_level"]),
        (SamplingStrategies, "getRand", ["size", "device"]),
        (MultiStepPrediction, "multiStepGenForward", ["generator", "inputIDs", "sampleRules", "device", "clearTreePast", "clearFullTree", "verbose", "sparseIndex"]),
        (MultiStepPrediction, "multiStepDiscForward", ["discriminator", "predictionTreeRoot", "device", "clearTreePast", "clearFullTree", "dualValueHeads"]),
        (MultiStepPrediction, "clearTree", ["root", "clearAll"]),
        (MultiStepPrediction, "getSubNodeListTreeStructure", ["root", "subNodes"]),
        (AdvantageCalculations, "dualHeadAdvantages", ["discInputScores", "alterationScores", "inputValueHeadPred", "alterationValueHeadPred",
                                                       "advantageSectionWeights", "inputIncorrectValueHeadsPolicy", "alterationIncorrectValueHeadsPolicy",
                                                       "minZoneCLoss", "epsilon", "min_denominator_value"]),
        (ReinforceLoss, "reinforceLoss", ["policyProbs", "advantages", "useNegativeEntropyMask", "useMinProbLimit", "negPushProbLimit", "eps"]),
        (ReinforceLoss, "mirroredNegativeReinforceLoss", ["policyProbs", "advantages", "maskZoneC"]),
        (ReinforceLoss, "calculateMinimumProbLimit", ["advantages", "policyProbs", "negPushProbLimit"]),
        (DiscriminatorLoss, "calcDiscriminatorLoss", ["gannSetup", "inputScores", "alterationScores", "numNegativeUpdateSamples", "discDepthMask",
                                                      "discDepthWeights", "collectTreeDepthData", "forwardPassData"]),
        (GeneratorLosses, "calcGeneratorLoss", ["gannSetup", "inputIDs", "generatorLogits", "generatorProbs", "discInputScores", "discAlterationScores",
                                                "forwardPassData", "reinforcementLearningConfig", "depthMask"]),
    ]
    for mod, name, params in surface:
        got = list(inspect.signature(getattr(mod, name)).parameters)
        assert got[:len(params)] == params, (name, got)
    for mod, names in ((SamplingStrategies, ["SamplingTechniques", "SamplingRule", "RANDOM_TECHNIQUES", "SAMPLE_BUFFER"]), (SampleBuffer, ["SamplingBuffer"]),
                       (MultiStepPrediction, ["PredictionTreeNode", "getRandomNodesFromRule"]), (GanSetup, ["GannSetup"]),
                       (Generator, ["NonResidualGenerator"]), (Discriminator, ["NonResidualDiscriminatorWithDualValueHeads"]),
                       (NonResidualAttention, ["GPT2NonResidualAttention"]), (NonResidualDiscriminatorBase, ["NonResidualGPT2LMHeadModel", "NonResidualGPT2Block"]),
                       (NeoXNonResidualAttention, ["NeoXNonResidualAttention"]), (NeoXNonResidualBase, ["GPTNeoXForCausalLM", "GPTNeoXLayer"]),
                       (TrainingConfig, ["SamplingConfig", "RoundRobinSamplingConfig"]), (ReinforceLabelMode, ["ReinforcementLabelMode", "IncorrectValueHeadsPolicy"]),
                       (Optimizers, ["OptimizerWrapper"]), (TrainingLoop, ["efficentTrainingLoop"]), (DepthSchedule, ["LinearUpdatesDepthSchedule"]),
                       (GanUpdateSchedule, ["AlwaysUpdateSchedule"])):
        for n in names:
            assert hasattr(mod, n), (mod.__name__, n)
    att = NonResidualAttention.GPT2NonResidualAttention
    assert list(inspect.signature(att.nonResidualTextAttention).parameters)[:9] == [
        "self", "query", "key", "value", "textualKeyPast", "textualValuePast", "head_mask", "customCasualMask", "dropoutRate"]
    neox = NeoXNonResidualAttention.NeoXNonResidualAttention
    assert list(inspect.signature(neox._nonResidualTextAttention).parameters)[:9] == [
        "self", "query", "key", "value", "textualKeyPast", "textualValuePast", "head_mask", "customCasualMask", "dropoutRate"]
    assert list(inspect.signature(neox.forward).parameters)[:11] == [
        "self", "hidden_states", "attention_mask", "position_ids", "head_mask", "layer_past", "use_cache", "output_attentions", "isNonResidual",
        "textualPast", "customCasualMask"]


def test_oracle_default_mask_attention_at_kernel_head_sizes(golden):
    g = golden("g10_boundary")
    for fam in ("neox", "gpt2"):
        for Dh, Q, Kl in ((32, 16, 40), (64, 5, 9), (32, 7, 7)):
            p = f"default_{fam}_{Dh}_{Q}_{Kl}_"
            t = {k: torch.from_numpy(g[p + k]) for k in ("q", "k", "v", "kctx", "vctx")}
            out, w = A.branch_attention(A.NEOX if fam == "neox" else A.GPT2, t["q"], t["k"], t["v"], t["kctx"], t["vctx"], mask=None)
            np.testing.assert_allclose(out.numpy(), g[p + "out"], atol=2e-6)
            np.testing.assert_allclose(w.numpy(), g[p + "weights"], atol=2e-6)


def test_oracle_advantages_and_reinforce_terms(golden):
    g = golden("g10_boundary")
    t = lambda k: torch.from_numpy(g[k])
    wA, wB, wC = g["adv_weights"].tolist()
    for tag, pi, pa in (("ii", 1, 1), ("ni", 0, 1), ("nn", 0, 0)):
        pol = lambda x: L.IGNORE_SAMPLE if x else 1 - L.IGNORE_SAMPLE
        ia, aa, extra = L.dual_head_advantages_full(t("adv_in_scores"), t("adv_alt_scores"), t("adv_in_head"), t("adv_alt_head"), wA, wB, wC,
                                                    pol(pi), pol(pa))
        assert np.array_equal(ia.numpy(), g[f"adv_{tag}_in_adv"]) and np.array_equal(aa.numpy(), g[f"adv_{tag}_alt_adv"])
        for nm, x in zip(["zoneA", "zoneB", "zoneC", "adv_correct", "input_correct"], extra):
            assert np.array_equal(x.numpy(), g[f"adv_{tag}_{nm}"]), (tag, nm)
    probs = t("rl_probs").requires_grad_(True)
    adv, mc, gl = t("rl_adv"), t("rl_maskC"), t("rl_gloss")
    for tag, fn in (("mirrored", lambda: L.mirrored_negative_reinforce(probs, adv, mc)), ("plain", lambda: L.reinforce_loss(probs, adv)),
                    ("minprob", lambda: L.reinforce_loss(probs, adv, True, 0.3))):
        loss = fn()
        np.testing.assert_array_equal(loss.detach().numpy(), g[f"rl_{tag}_loss"])
        np.testing.assert_array_equal(torch.autograd.grad(loss, probs, gl)[0].numpy(), g[f"rl_{tag}_dprobs"])


def test_oracle_noise_buffer_roll_and_host_buffer(golden):
    """The reference's cat of two slices per dim == one torch.roll by -split (what branchgan.sampling.SamplingBuffer does)."""
    import branchgan.sampling as SS
    g = golden("g10_boundary")
    noise, splits, rolled = torch.from_numpy(g["buf_noise"]), g["buf_splits"], g["buf_rolled"]
    for c in range(6):
        assert np.array_equal(S.rolled_noise(noise, splits, c % 4).numpy(), rolled[c])
    buf = SS.SamplingBuffer("cpu", numShuffleSplits=4)
    buf.shapeRandBuffers[tuple(noise.shape)] = [noise, [splits[0], splits[1]], 0]
    for c in range(6):
        assert np.array_equal(buf.rand(noise.shape).numpy(), rolled[c]), c
