"""graphs.GraphedTrainStep (three HIP graphs per step, all-reduces behind the next graph) against the eager
train step of training.train_step (TrainingLoop.py:141-152): same losses at every step and the same parameters at the end.

Both runs execute the same kernels on the same inputs (greedy TOP_K sampling is deterministic), so fp32 results agree to
rounding noise: rtol 1e-5 / atol 1e-6 on the losses, rtol 1e-4 / atol 1e-6 on the parameters after 7 optimizer steps."""
import pytest
import torch

from branchgan import configs, gan, sampling, training
from branchgan.graphs import GraphedTrainStep

pytestmark = pytest.mark.gpu


def _setup(seed=0):
    torch.manual_seed(seed)
    model = gan.GannSetup("gpt2-tiny", "gpt2-tiny", mleLossWeight=1, rlLossWeight=1, discValueHeadLossWeight=0.25, device="cuda").cuda()
    rules = [sampling.SamplingRule(1, sampling.SamplingTechniques.TOP_K, {}) for _ in range(3)]
    conf = configs.SamplingConfig(rules, includeAllMultiStepSamplesForDisc=True, numSampleSequences=8, seed=77)
    rl = configs.ReinforcementLearningConfig(labelMode=configs.ReinforcementLabelMode.ALTERATIONS_AND_INPUT_UNWEIGHTED,
                                             sectionWeightA=1, sectionWeightB=1, sectionWeightC=1.5, teacherForcing=True)
    g_opt = training.OptimizerWrapper(model.generator, training.make_optimizer(model.generator.parameters(), 'adam', 1e-4, betas=(0.5, 0.999)))
    d_opt = training.OptimizerWrapper(model.discriminator, training.make_optimizer(model.discriminator.parameters(), 'adam', 1e-3))
    return model, conf, rl, g_opt, d_opt


@pytest.mark.parametrize("amp,defer_d", [(None, "0"), (None, "1"), (torch.bfloat16, "0")])
def test_graphed_step_matches_eager(amp, defer_d, monkeypatch):
    monkeypatch.setenv("BRANCHGAN_DEFER_D", defer_d)     # "1": the multi-rank schedule (D's optimizer step deferred to the next call)
    V = configs.LMConfig.from_any("gpt2-tiny").vocab_size
    data = torch.randint(0, V, (5, 4, 64), generator=torch.Generator().manual_seed(5)).cuda()
    # the sparse branch points are drawn by the sampling config: pin them so both runs branch at the same columns
    sparse = torch.tensor([2, 9, 17, 25, 33, 41, 50, 60], device="cuda")

    model, conf, rl, g_opt, d_opt = _setup()
    conf.sparseIndex = sparse
    eager = []
    for i in range(3):                       # GraphedTrainStep's warm-up: three eager steps on its example batch
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp is not None):
            training.train_step(model, data[0], conf, g_opt, d_opt, i, 9999, rl)
    for i in range(4):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp is not None):
            d, g = training.train_step(model, data[1 + i], conf, g_opt, d_opt, 3 + i, 9999, rl)
        eager.append((float(d["loss"].detach()), float(g["loss"].detach()), float(g["loss-RL"].detach())))
    ref = {k: v.detach().clone() for k, v in model.state_dict().items()}

    model, conf, rl, g_opt, d_opt = _setup()
    conf.sparseIndex = sparse
    graphed = GraphedTrainStep(model, conf, g_opt, d_opt, data[0], 9999, rl, amp, warmup=3)
    got = []
    for i in range(4):
        d, g = graphed.step(data[1 + i], 3 + i)
        got.append((float(d["loss"].detach()), float(g["loss"].detach()), float(g["loss-RL"].detach())))
    assert (graphed._pending_d is not None) == (defer_d == "1")
    graphed.check_nan()                      # also finishes a deferred discriminator step
    assert graphed._pending_d is None
    tol = dict(rtol=1e-5, atol=1e-6) if amp is None else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(torch.tensor(got), torch.tensor(eager), **tol)
    ptol = dict(rtol=1e-4, atol=1e-6) if amp is None else dict(rtol=0, atol=2.5e-3)      # bf16: Adam steps of 1e-3 can flip sign
    for k, v in model.state_dict().items():
        torch.testing.assert_close(v, ref[k], msg=lambda m: f"{k}: {m}", **ptol)


def test_graph_capture_requires_the_fused_path(monkeypatch):
    """The module-by-module path is eager-only (graphs._require_fused_path explains why)."""
    from branchgan import fused
    model, conf, rl, g_opt, d_opt = _setup()
    ids = torch.randint(0, 100, (4, 64), device="cuda")
    monkeypatch.setattr(fused, "ENABLED", False)
    with pytest.raises(NotImplementedError, match="fused block path"):
        GraphedTrainStep(model, conf, g_opt, d_opt, ids, 9999, rl, None)
