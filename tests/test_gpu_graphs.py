"""graphs.GraphedTrainStep (three HIP graphs per step, all-reduces behind the next graph) against the eager
train step of training.train_step (TrainingLoop.py:141-152): same losses at every step and the same parameters at the end,
at the test shapes AND at BASELINE configs[1] shapes (GPT-2 small, batch 8, seq 256, 32 branch points, depth 4), on the fused
block path and on the module-by-module path, fp32 and bf16, over four replays.

Both runs execute the same kernels on the same inputs (greedy TOP_K sampling is deterministic), so fp32 results agree to
rounding noise: rtol 1e-5 / atol 1e-6 on the losses, rtol 1e-4 / atol 1e-6 on the parameters after 4 optimizer steps."""
import os

import numpy as np
import pytest
import torch

from branchgan import configs, fused, gan, sampling, training
from branchgan.graphs import GraphedTrainStep

pytestmark = pytest.mark.gpu
T = sampling.SamplingTechniques


def _setup(seed=0, models=("gpt2-tiny", "gpt2-tiny"), depth=3, K=8, strategies=None, lr=(1e-4, 1e-3)):
    torch.manual_seed(seed)
    model = gan.GannSetup(models[0], models[1], mleLossWeight=1, rlLossWeight=1, discValueHeadLossWeight=0.25, device="cuda").cuda()
    strategies = strategies or [T.TOP_K] * depth
    rules = [sampling.SamplingRule(1, st, {}) for st in strategies]
    conf = configs.SamplingConfig(rules, includeAllMultiStepSamplesForDisc=True, numSampleSequences=K, seed=77)
    rl = configs.ReinforcementLearningConfig(labelMode=configs.ReinforcementLabelMode.ALTERATIONS_AND_INPUT_UNWEIGHTED,
                                             sectionWeightA=1, sectionWeightB=1, sectionWeightC=1.5, teacherForcing=True)
    g_opt = training.OptimizerWrapper(model.generator, training.make_optimizer(model.generator.parameters(), 'adam', lr[0], betas=(0.5, 0.999)))
    d_opt = training.OptimizerWrapper(model.discriminator, training.make_optimizer(model.discriminator.parameters(), 'adam', lr[1]))
    return model, conf, rl, g_opt, d_opt


SPARSE = [2, 9, 17, 25, 33, 41, 50, 60]


@pytest.mark.parametrize("amp,defer_d", [(None, "0"), (None, "1"), (torch.bfloat16, "0")])
def test_graphed_step_matches_eager(amp, defer_d, monkeypatch):
    monkeypatch.setenv("BRANCHGAN_DEFER_D", defer_d)     # "1": the multi-rank schedule (D's optimizer step deferred to the next call)
    V = configs.LMConfig.from_any("gpt2-tiny").vocab_size
    data = torch.randint(0, V, (5, 4, 64), generator=torch.Generator().manual_seed(5)).cuda()
    # the sparse branch points are drawn by the sampling config: pin them so both runs branch at the same columns
    sparse = torch.tensor(SPARSE, device="cuda")

    model, conf, rl, g_opt, d_opt = _setup()
    conf.sparseIndex = sparse
    eager = []
    for i in range(4):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp is not None):
            d, g = training.train_step(model, data[1 + i], conf, g_opt, d_opt, i, 9999, rl)
        eager.append((float(d["loss"].detach()), float(g["loss"].detach()), float(g["loss-RL"].detach())))
    ref = {k: v.detach().clone() for k, v in model.state_dict().items()}

    model, conf, rl, g_opt, d_opt = _setup()
    conf.sparseIndex = sparse
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    graphed = GraphedTrainStep(model, conf, g_opt, d_opt, data[0], 9999, rl, amp, warmup=3)
    # the warm-up steps ran on data[0] and were undone: building a graphed step does not train the model
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k]), k
    assert g_opt.optimizer.steps == 0 and float(g_opt.optimizer.exp_avg.abs().max()) == 0.0
    got = []
    for i in range(4):
        d, g = graphed.step(data[1 + i], i)
        got.append((float(d["loss"].detach()), float(g["loss"].detach()), float(g["loss-RL"].detach())))
    assert (graphed._pending_d is not None) == (defer_d == "1")
    graphed.check_nan()                      # also finishes a deferred discriminator step
    assert graphed._pending_d is None
    tol = dict(rtol=1e-5, atol=1e-6) if amp is None else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(torch.tensor(got), torch.tensor(eager), **tol)
    ptol = dict(rtol=1e-4, atol=1e-6) if amp is None else dict(rtol=0, atol=2.5e-3)      # bf16: Adam steps of 1e-3 can flip sign
    for k, v in model.state_dict().items():
        torch.testing.assert_close(v, ref[k], msg=lambda m: f"{k}: {m}", **ptol)


@pytest.mark.parametrize("amp", [None, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("fused_on", [True, False], ids=["fused", "unfused"])
def test_graph_replays_match_eager_at_c2_shapes(amp, fused_on, monkeypatch):
    """BASELINE configs[1] shapes, four replays on four different batches with the weights held fixed (no optimizer step), each
    compared with an eager forward + both backwards on the same batch: every loss and EVERY element of both gradient arenas.
    Round 1 saw NaN bias gradients from the SECOND replay on the module-by-module path at exactly these shapes; the cause was
    a hipGraph memset node (graphs.finalize_graph replaces them), so that path is covered too."""
    monkeypatch.setattr(fused, "ENABLED", fused_on)
    V = configs.LMConfig.from_any("gpt2-small").vocab_size
    data = torch.randint(0, V, (5, 8, 256), generator=torch.Generator().manual_seed(6)).cuda()
    model, conf, rl, g_opt, d_opt = _setup(models=("gpt2-small", "gpt2-small"), depth=4, K=32)
    conf.sparseIndex = torch.tensor(np.sort(np.random.RandomState(3).choice(255, 32, replace=False)), device="cuda")
    gs = GraphedTrainStep(model, conf, g_opt, d_opt, data[0], 9999, rl, amp, warmup=1)
    nodes, kernels, memsets, _ = map(sum, zip(*gs.census))
    assert kernels > 500
    import os
    if os.environ.get("BRANCHGAN_OVERLAP_BWD") != "0":       # (an A/B run may switch the overlap off)
        assert gs._side_by_side == fused_on  # the two backward graphs beside each other: only on the path whose scratch this library owns
    if not fused_on:
        assert memsets > 0                   # ATen's multi-block reductions brought memset nodes; all were replaced
    worst = 0.0
    for r in range(4):
        ids = data[1 + r]
        gs.static_ids.copy_(ids)
        gs.replay_graphs()                   # the discriminator's backward graph beside the generator's, as a step replays them
        torch.cuda.synchronize()
        got = {k: float(v.detach()) for k, v in (("d", gs.losses[0]["loss"]), ("g", gs.losses[1]["loss"]), ("rl", gs.losses[1]["loss-RL"]))}
        g_arena, d_arena = g_opt.arena.arena.clone(), d_opt.arena.arena.clone()
        g_opt.arena.zero(); d_opt.arena.zero()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp is not None):
            d, g = model.trainForwardPassNoMetrics(ids, conf, 9999, collectTreeDepthData=False, reinforcementLearningConfig=rl)
            g["loss"].backward()
            d["loss"].backward()
        torch.cuda.synchronize()
        want = {"d": float(d["loss"].detach()), "g": float(g["loss"].detach()), "rl": float(g["loss-RL"].detach())}
        for k in want:
            np.testing.assert_allclose(got[k], want[k], rtol=1e-5 if amp is None else 1e-3, err_msg=f"replay {r} loss {k}")
        for name, a, b in (("G", g_arena, g_opt.arena.arena), ("D", d_arena, d_opt.arena.arena)):
            assert bool(torch.isfinite(a).all()), f"replay {r}: non-finite gradient in the {name} arena"
            scale = float(b.abs().max())
            err = float((a - b).abs().max()) / scale
            worst = max(worst, err)
            assert err <= (1e-5 if amp is None else 2e-3), f"replay {r} {name} arena: max |graph - eager| = {err:.3e} of max |grad|"
        g_opt.arena.zero(); d_opt.arena.zero()
    print(f"worst arena mismatch (relative to max |grad|): {worst:.3e}")


def test_exponential_race_noise_is_fresh_on_every_replay():
    """SAMPLE_NO_REPLACEMENT inside a captured step: the noise lives in persistent tensors that step() refills before each
    replay, so two replays on the SAME batch sample different branch tokens; with an injected noise source the graphed
    step samples exactly what the eager step samples."""
    V = configs.LMConfig.from_any("gpt2-tiny").vocab_size
    data = torch.randint(0, V, (2, 4, 64), generator=torch.Generator().manual_seed(5)).cuda()
    model, conf, rl, g_opt, d_opt = _setup(strategies=[T.SAMPLE_NO_REPLACEMENT, T.TOP_K, T.SAMPLE_NO_REPLACEMENT])
    conf.sparseIndex = torch.tensor(SPARSE, device="cuda")
    gs = GraphedTrainStep(model, conf, g_opt, d_opt, data[0], 9999, rl, None, warmup=1)
    assert len(gs._noise) == 2 and model.noise_fn is None
    seen = []
    for i in range(2):
        gs.step(data[1], i)
        torch.cuda.synchronize()
        seen.append(torch.stack([t.clone() for t in gs.sampled_ids]))
    gs.check_nan()
    assert (seen[0] != seen[1]).float().mean() > 0.3          # fresh noise: most sampled tokens differ

    # injected noise: a fixed sequence of noise tensors, consumed in call order by the eager step and by the graphed one
    def make_source():
        g = torch.Generator(device="cuda").manual_seed(99)
        return lambda shape, device: -torch.log(torch.rand(shape, device=device, generator=g) + 1e-14)

    model, conf, rl, g_opt, d_opt = _setup(strategies=[T.SAMPLE_NO_REPLACEMENT, T.TOP_K, T.SAMPLE_NO_REPLACEMENT])
    conf.sparseIndex = torch.tensor(SPARSE, device="cuda")
    model.noise_fn = make_source()
    d, g = model.trainForwardPassNoMetrics(data[1], conf, 9999, collectTreeDepthData=False, reinforcementLearningConfig=rl)
    want = (float(d["loss"].detach()), float(g["loss"].detach()))
    model.noise_fn = make_source()
    with pytest.raises(RuntimeError, match="another stream is still alive"):     # d, g hold an autograd graph built on the default stream
        GraphedTrainStep(model, conf, g_opt, d_opt, data[1], 9999, rl, None, warmup=1)
    del d, g
    model, conf, rl, g_opt, d_opt = _setup(strategies=[T.SAMPLE_NO_REPLACEMENT, T.TOP_K, T.SAMPLE_NO_REPLACEMENT])
    conf.sparseIndex = torch.tensor(SPARSE, device="cuda")
    model.noise_fn = make_source()
    gs = GraphedTrainStep(model, conf, g_opt, d_opt, data[1], 9999, rl, None, warmup=1)
    # the warm-up step consumed the source's first two tensors: rewind it so the first step sees the same noise as the eager run
    gs._user_noise = make_source()
    d, g = gs.step(data[1], 0)
    torch.testing.assert_close(torch.tensor([float(d["loss"].detach()), float(g["loss"].detach())]), torch.tensor(want), rtol=1e-5, atol=1e-6)


def test_random_techniques_are_refused():
    model, conf, rl, g_opt, d_opt = _setup(strategies=[T.TOP_K, T.EPSILON_GREEDY, T.TOP_K])
    ids = torch.randint(0, 100, (4, 64), device="cuda")
    with pytest.raises(NotImplementedError, match="EPSILON_GREEDY"):
        GraphedTrainStep(model, conf, g_opt, d_opt, ids, 9999, rl, None)


def test_update_flags_and_depth_guard():
    """updateGen=False: the generator's gradients are discarded and its weights stay (TrainingLoop.py:129-134's schedule)."""
    V = configs.LMConfig.from_any("gpt2-tiny").vocab_size
    data = torch.randint(0, V, (2, 4, 64), generator=torch.Generator().manual_seed(5)).cuda()
    model, conf, rl, g_opt, d_opt = _setup()
    conf.sparseIndex = torch.tensor(SPARSE, device="cuda")
    gs = GraphedTrainStep(model, conf, g_opt, d_opt, data[0], 9999, rl, None, warmup=1)
    g_before = [p.detach().clone() for p in model.generator.parameters()]
    d_before = [p.detach().clone() for p in model.discriminator.parameters()]
    gs.step(data[1], 0, updateGen=False, updateDisc=True)
    gs.join()
    assert all(torch.equal(a, b.detach()) for a, b in zip(g_before, model.generator.parameters()))
    assert any(not torch.equal(a, b.detach()) for a, b in zip(d_before, model.discriminator.parameters()))
    assert float(g_opt.arena.arena.abs().max()) == 0.0
    shallow = configs.SamplingConfig(conf.rules[:2], includeAllMultiStepSamplesForDisc=True, numSampleSequences=8)
    with pytest.raises(ValueError, match="depth"):
        gs.step(data[1], 1, sampleConf=shallow)


def test_training_with_backward_graphs_side_by_side_equals_one_after_the_other():
    """40 optimizer steps of the bench workload c1 (graphed, bf16, Adam) with the discriminator's backward graph replayed beside
    the generator's (the default on one rank) and with the graphs replayed in turn: the same final losses to the last bit
    (at configs[1]: 200 steps, profiles / DESIGN §5 item 12).  Anything the two graphs shared - pool memory, a GEMM workspace,
    scratch - would show here."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    losses = {}
    for overlap in ("1", "0"):
        env = dict(os.environ, BRANCHGAN_OVERLAP_BWD=overlap)
        run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "c1", "--steps", "40", "--warmup", "2",
                              "--no-cpu-baseline", "--no-fp32", "--no-rooflines"], env=env, capture_output=True, text=True, timeout=280)
        assert run.returncode == 0, run.stderr[-2000:]
        losses[overlap] = json.loads(run.stdout.strip().splitlines()[-1])["config"]["final_losses"]
    assert losses["1"] == losses["0"], losses
    assert all(abs(v) < 1e4 for v in losses["1"].values())
