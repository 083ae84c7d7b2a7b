"""The evaluation path and the full training loop end to end on a tiny model: `trainForwardPass` returns
the reference's metric keys (key sets taken from the reference's own output, tests/golden/g7_metrics.npz), `validateModel`
collects them per strategy, `efficentTrainingLoop` honours the update schedule, evaluates, logs and writes checkpoints that
load back.  The metric VALUES are pinned by tests/test_gpu_metrics.py; here a few are re-derived with plain torch sorts from
the same forward pass as a plumbing check."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _setup():
    from branchgan import configs, gan, sampling
    torch.manual_seed(0)
    model = gan.GannSetup("gpt2-tiny", "gpt2-tiny", device="cuda").cuda()
    rules = lambda: [sampling.SamplingRule(1, sampling.SamplingTechniques.TOP_K, {}) for _ in range(2)]
    train_conf = configs.SamplingConfig(rules(), includeAllMultiStepSamplesForDisc=True, numSampleSequences=8, seed=3)
    eval_conf = configs.SamplingConfig(rules(), includeAllMultiStepSamplesForDisc=True, numSampleSequences=None)
    return model, train_conf, eval_conf


def test_train_forward_pass_returns_reference_metric_keys():
    from branchgan import metrics
    g = np.load(os.path.join(GOLDEN, "g7_metrics.npz"))
    keys = lambda prefix: {k[len(f"a::{prefix}::"):] for k in g.files if k.startswith(f"a::{prefix}::")}
    model, train_conf, eval_conf = _setup()
    ids = torch.randint(0, 2048, (3, 24), device="cuda")
    with torch.no_grad():
        discLosses, genLosses, discMetrics, genMetrics = model.trainForwardPass(ids, eval_conf, 1, collectData=True)
        assert set(genMetrics) == keys("gen") and set(discMetrics) == keys("disc")
        assert genMetrics["Top-20"].shape == (3, 24, 20) and genMetrics["RL-MeanAdvantage"].shape == (3, 1, 23)
        for name, d in (("gen", genMetrics), ("disc", discMetrics)):
            for k, v in d.items():
                assert bool(torch.isfinite(v.float()).all()), (name, k)
        # plumbing: the same quantities by plain torch sorts on the generator's probabilities of this batch
        probs = torch.softmax(model.generator({'input_ids': ids, 'returnLmLogits': True}).prediction.logits.float(), -1)
        top8 = torch.sort(probs, -1, descending=True).values[..., :8].sum(-1).mean()
        torch.testing.assert_close(genMetrics["RankProbMass-8"], top8, rtol=1e-5, atol=1e-6)
        lab = torch.gather(probs[:, :-1], -1, ids[:, 1:].unsqueeze(-1))
        place = (probs[:, :-1] > lab).sum(-1).float()
        torch.testing.assert_close(genMetrics["Input-PredPlacement-Mean"], place.mean())
        torch.testing.assert_close(metrics.mleLossFromLogits(ids, torch.log(probs)), genLosses["loss-MLE"].float(), rtol=1e-4, atol=1e-4)
        with pytest.raises(ValueError, match="K == S"):
            model.trainForwardPass(ids, train_conf, 1)


def test_training_loop_with_schedule_evaluation_and_checkpoints(tmp_path):
    from branchgan import checkpoint, configs, gan, schedules, training
    model, train_conf, eval_conf = _setup()
    g = torch.Generator().manual_seed(1)
    train = [torch.randint(0, 2048, (4, 32), generator=g).cuda() for _ in range(4)]
    test = [torch.randint(0, 2048, (2, 16), generator=g).cuda() for _ in range(2)]
    g_opt = training.OptimizerWrapper(model.generator, training.make_optimizer(model.generator.parameters(), 'adam', 1e-4, betas=(0.5, 0.999)))
    d_opt = training.OptimizerWrapper(model.discriminator, training.make_optimizer(model.discriminator.parameters(), 'adam', 1e-3))
    d_opt.scheduler = schedules.FreezeLRSchedule(d_opt.optimizer, zero_steps=1, warmup_steps=2, lr_start=1e-4, lr_end=1e-3)
    conf = types.SimpleNamespace(depthSchedule=configs.LinearUpdatesDepthSchedule(1, 1, 2, 2), reinforcementLearningConfig=None, saveOptimizers=True)
    gen_before = {k: v.detach().clone() for k, v in model.generator.state_dict().items()}
    rows = []
    log = training.efficentTrainingLoop(model, train, test, 2, g_opt, d_opt, train_conf, 9999, 3, {"greedy": eval_conf}, saveName="run",
                                        genDiscUpdateSchedule=schedules.GeneratorSlowStart(3), trainingConf=conf, saveFreq=0.5, epochSaveFreq=1,
                                        mainSavePath=str(tmp_path), logSink=rows.append)
    assert log == rows and len(rows) > 0
    val = [r for r in rows if any(k.startswith("Val-greedy_Generator") for k in r)]
    assert len(val) == 3                                              # evaluations after steps 1, 4, 7
    assert "Val-greedy_Generator-RankProbMass-16" in val[0] and "Val-greedy_Generator-loss" in val[0]
    disc_rows = [r for r in rows if "Discriminator-loss" in r]
    assert all(np.isfinite(list(r.values())).all() for r in disc_rows) and "Discriminator-lr" in disc_rows[-1]
    assert [r["Generator-generationDepth"] for r in rows if "Generator-generationDepth" in r] == [1, 2, 2]
    # GeneratorSlowStart(3): the generator must not have moved after the first three micro-steps ... it has by the end
    assert any(not torch.equal(v, gen_before[k]) for k, v in model.generator.state_dict().items())
    root = tmp_path / "run"
    marks = checkpoint.getAvailablCheckpointsFromModelFolder(str(root))
    assert marks[:4] == ["Epoch-0.5", "Epoch-1", "Epoch-1.5", "Epoch-2"] and marks[-1] == "Current"
    fresh = gan.GannSetup("gpt2-tiny", "gpt2-tiny", device="cuda").cuda()
    checkpoint.loadGannSetup(fresh, str(root / "Epoch-2"))
    for (k, a), (_, b) in zip(model.state_dict().items(), fresh.state_dict().items()):
        assert torch.equal(a, b), k


def test_update_schedule_freezes_the_generator():
    from branchgan import schedules, training
    model, train_conf, _ = _setup()
    g_opt = training.OptimizerWrapper(model.generator, training.make_optimizer(model.generator.parameters(), 'adam', 1e-3))
    d_opt = training.OptimizerWrapper(model.discriminator, training.make_optimizer(model.discriminator.parameters(), 'adam', 1e-3))
    before_g = {k: v.detach().clone() for k, v in model.generator.state_dict().items()}
    before_d = {k: v.detach().clone() for k, v in model.discriminator.state_dict().items()}
    data = [torch.randint(0, 2048, (4, 32), device="cuda") for _ in range(2)]
    training.efficentTrainingLoop(model, data, None, 1, g_opt, d_opt, train_conf, 9999, 100, {}, doLogData=False, saveModel=False,
                                  genDiscUpdateSchedule=schedules.OnlyUpdateDisc())
    assert all(torch.equal(v, before_g[k]) for k, v in model.generator.state_dict().items())
    assert any(not torch.equal(v, before_d[k]) for k, v in model.discriminator.state_dict().items())


def test_training_loop_on_graph_replays_matches_the_eager_loop():
    """efficentTrainingLoop(useGraphs=True): every micro-step is a GraphedTrainStep replay (one captured step per depth, built on
    first use without training the model), under the same update and depth schedules as the eager loop -> same parameters."""
    from branchgan import configs, schedules, training

    def run(use_graphs):
        model, train_conf, _ = _setup()
        g = torch.Generator().manual_seed(1)
        train = [torch.randint(0, 2048, (4, 32), generator=g).cuda() for _ in range(6)]
        g_opt = training.OptimizerWrapper(model.generator, training.make_optimizer(model.generator.parameters(), 'adam', 1e-4, betas=(0.5, 0.999)))
        d_opt = training.OptimizerWrapper(model.discriminator, training.make_optimizer(model.discriminator.parameters(), 'adam', 1e-3))
        train_conf.sparseIndex = torch.tensor([1, 5, 9, 13, 17, 21, 25, 29], device="cuda")
        conf = types.SimpleNamespace(depthSchedule=configs.LinearUpdatesDepthSchedule(1, 1, 2, 3), reinforcementLearningConfig=None, saveOptimizers=False)
        rows = []
        training.efficentTrainingLoop(model, train, None, 1, g_opt, d_opt, train_conf, 9999, 100, {}, saveModel=False,
                                      genDiscUpdateSchedule=schedules.GeneratorSlowStart(2), trainingConf=conf, logSink=rows.append,
                                      useGraphs=use_graphs)
        return {k: v.detach().clone() for k, v in model.state_dict().items()}, rows

    eager, rows_e = run(False)
    graphed, rows_g = run(True)
    for k in eager:
        torch.testing.assert_close(graphed[k], eager[k], rtol=1e-4, atol=1e-6, msg=lambda m: f"{k}: {m}")
    le = [r["Generator-loss"] for r in rows_e if "Generator-loss" in r]
    lg = [r["Generator-loss"] for r in rows_g if "Generator-loss" in r]
    np.testing.assert_allclose(lg, le, rtol=1e-5)


def test_token_stream_prefetch_to_gpu(tmp_path):
    """The pinned-buffer / side-stream staging hands out the same batches as the CPU stream, in order, over three epochs."""
    import json
    from branchgan import data
    rng = np.random.RandomState(0)
    files = {}
    for i, n in enumerate((37, 5, 64)):
        rng.randint(0, 50257, (n, 128)).astype('<i4').tofile(tmp_path / f"part{i}.ids")
        files[f"part{i}.ids"] = n
    with open(tmp_path / "dataset.json", "w") as fp:
        json.dump({"format": data.FORMAT, "seq_len": 128, "files": files}, fp)
    for shuffle in (False, True):
        cpu = data.loadPreTokenizedDataloader(8, str(tmp_path), doShuffle=shuffle, seed=9, ddpWorldSize=2, ddpRank=1)
        gpu = data.loadPreTokenizedDataloader(8, str(tmp_path), doShuffle=shuffle, seed=9, ddpWorldSize=2, ddpRank=1, device="cuda")
        assert len(cpu) == len(gpu) == 6
        for _ in range(3):
            for a, b in zip(cpu, gpu):
                assert b.is_cuda and b.dtype == torch.int64 and b.shape == (8, 128)
                assert torch.equal(a, (b + 0).cpu())             # a kernel on the current stream reads the batch: ordered after the copy
