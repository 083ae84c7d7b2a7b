"""SURVEY.md §8(f) rows 1 and 4 - checkpoints in the reference's folder layout, update / learning-rate / save schedules -
on the CPU.  Expected values: `tests/golden/g6_schedules.json`, captured from the reference's own classes by
tests/golden/make_goldens.py (Training/GanUpdateSchedule.py, Training/Optimizers.py FreezeLRSchedule + _createScheduler,
Training/Utils.py ModelSaveFrequencyTracker); the state-dict goldens of g35_* for the checkpoint round trip."""
import json
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def g6():
    with open(os.path.join(GOLDEN, "g6_schedules.json")) as f:
        return json.load(f)


def test_update_schedules_match_reference(g6):
    from branchgan import schedules
    for case in g6["update_schedules"]:
        sched = getattr(schedules, case["cls"])(*case["args"])
        assert repr(sched) == case["repr"]
        assert [sched.saveGen, sched.saveDisc] == case["saves"]
        got = [list(sched.getGenDiscUpdates(s, e, None, None)) for s, e in case["points"]]
        assert got == case["decisions"], case["cls"]
    with pytest.raises(NotImplementedError):
        schedules.GenDiscUpdateSchedule(True, True).getGenDiscUpdates(0, 0, None, None)


def test_lr_schedules_match_reference(g6):
    from branchgan import schedules
    for case in g6["freeze_lr"]:
        p = torch.nn.Parameter(torch.zeros(3))
        opt = torch.optim.SGD([p], lr=case["base_lr"])
        sch = schedules.FreezeLRSchedule(opt, **case["kwargs"])
        lrs = [opt.param_groups[0]["lr"]]
        for _ in range(case["steps"]):
            opt.step()
            sch.step()
            lrs.append(opt.param_groups[0]["lr"])
        np.testing.assert_allclose(lrs, case["lrs"], rtol=1e-12, atol=0)
    for case in g6["create_scheduler"]:
        p = torch.nn.Parameter(torch.zeros(3))
        opt = torch.optim.SGD([p], lr=0.1, momentum=0.9)
        sch = schedules.create_scheduler(torch.optim.lr_scheduler.CyclicLR, dict(case["args"]), opt, case["gradAccum"], case["stepsPerEpoch"],
                                         case["epochs"])
        lrs = []
        for _ in range(case["steps"]):
            opt.step()
            sch.step()
            lrs.append(opt.param_groups[0]["lr"])
        np.testing.assert_allclose(lrs, case["lrs"], rtol=1e-12, atol=0)
    assert schedules.create_scheduler(None, {}, None, 1, 10, 1) is None
    assert schedules.create_scheduler(torch.optim.lr_scheduler.CyclicLR, None, None, 1, 10, 1) is None


def test_save_tracker_matches_reference(g6):
    from branchgan import schedules
    for case in g6["save_tracker"]:
        tr = schedules.ModelSaveFrequencyTracker(*case["args"], **case["kwargs"])
        got = []
        for epoch in range(case["epochs"]):
            tr.initEpoch()
            for b in range(case["batches"]):
                due, name = tr.shouldModelBeSaved(b, epoch)
                if due:
                    got.append([epoch, b, name])
            got.append([epoch, "end", bool(tr.shouldModelBeSavedAtThisEpoch(epoch + 1))])
        assert got == case["events"]


@pytest.mark.parametrize("tag,family", [("neox_chain4", "gpt_neox"), ("gpt2_chain3", "gpt2")])
def test_checkpoint_round_trip_in_reference_layout(tmp_path, tag, family):
    """save -> folder layout of Utils.py:51-94 -> load into a fresh setup: same parameters and optimizer state; with
    referenceBuffers=True the files hold exactly the key set (and mask-buffer values) of the reference's own state dict."""
    from branchgan import checkpoint, configs, gan, training
    g = np.load(os.path.join(GOLDEN, f"g35_{tag}.npz"))
    ref = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    mk = lambda hidden, heads: configs.LMConfig(family, 97, hidden, 2, heads, 64, intermediate_size=2 * hidden if family == "gpt_neox" else 0)
    torch.manual_seed(1)
    setup = gan.GannSetup(mk(64, 2), mk(32, 1), device="cpu")
    setup.load_state_dict(ref)
    g_opt = training.OptimizerWrapper(setup.generator, torch.optim.Adam(setup.generator.parameters(), lr=1e-3))
    d_opt = training.OptimizerWrapper(setup.discriminator, torch.optim.Adam(setup.discriminator.parameters(), lr=1e-3))
    for opt, mod in ((g_opt, setup.generator), (d_opt, setup.discriminator)):      # one real Adam step so there is state to save
        sum((p ** 2).sum() for p in mod.parameters()).backward()
        opt.optimizer.step()
    here = checkpoint.saveGannSetup(setup, "run/A", "Epoch-0.5", d_opt, g_opt, mainSavePath=str(tmp_path), referenceBuffers=True)
    checkpoint.saveGannSetup(setup, "run/A", "Current", d_opt, g_opt, mainSavePath=str(tmp_path), saveGen=False)
    checkpoint.saveGannSetup(setup, "run/A", "Epoch-2", mainSavePath=str(tmp_path))
    root = tmp_path / "run-A"
    assert sorted(os.listdir(root)) == ["Current", "Discriminator-Config.json", "Epoch-0.5", "Epoch-2", "Generator-Config.json"]
    assert sorted(os.listdir(root / "Epoch-0.5" / "Generator")) == ["Generator.pt", "Optimizer.pt"]
    assert os.listdir(root / "Current" / "Generator") == []
    assert checkpoint.getAvailablCheckpointsFromModelFolder(str(root)) == ["Epoch-0.5", "Epoch-2"]     # 'Current' lacks Generator.pt
    assert configs.LMConfig.from_json_file(str(root / "Generator-Config.json")) == setup.generatorConfig

    # the two weight files together hold the reference's key set, mask buffers included, value for value
    files = {}
    for part, prefix in (("Generator", "generator."), ("Discriminator", "discriminator.")):
        sd = checkpoint.loadStateDict(os.path.join(here, part, part + ".pt"))
        files.update({prefix + k: v for k, v in sd.items()})
    assert set(files) == set(ref)
    for k, v in ref.items():
        assert files[k].dtype == v.dtype and files[k].shape == v.shape, k
        if k.endswith((".attn.bias", ".self_bias", ".masked_bias", ".inv_freq")):        # buffers: value for value
            assert torch.equal(files[k], v), k

    fresh = gan.GannSetup(mk(64, 2), mk(32, 1), device="cpu")
    g2 = training.OptimizerWrapper(fresh.generator, torch.optim.Adam(fresh.generator.parameters(), lr=1e-3))
    d2 = training.OptimizerWrapper(fresh.discriminator, torch.optim.Adam(fresh.discriminator.parameters(), lr=1e-3))
    checkpoint.loadGannSetup(fresh, here, d2, g2)
    for (k, a), (_, b) in zip(setup.state_dict().items(), fresh.state_dict().items()):
        assert torch.equal(a, b), k
    for a, b in ((g_opt, g2), (d_opt, d2)):
        sa, sb = a.optimizer.state_dict()["state"], b.optimizer.state_dict()["state"]
        assert sa.keys() == sb.keys() and len(sa) > 0
        for i in sa:
            assert torch.equal(sa[i]["exp_avg"], sb[i]["exp_avg"]) and torch.equal(sa[i]["exp_avg_sq"], sb[i]["exp_avg_sq"])

    # DDP-prefixed start weights (FileManagement.py:5-7, 25-32)
    ddp = {"module." + k: v for k, v in setup.generator.state_dict().items()}
    torch.save(ddp, tmp_path / "ddp.pt")
    other = gan.GannSetup(mk(64, 2), mk(32, 1), device="cpu")
    checkpoint.initModelWeights(other, genStartWeights=str(tmp_path / "ddp.pt"))
    for (k, a), (_, b) in zip(setup.generator.state_dict().items(), other.generator.state_dict().items()):
        assert torch.equal(a, b), k
