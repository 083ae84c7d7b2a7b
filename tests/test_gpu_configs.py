"""Every BASELINE.json config as a train step on the HIP path, at the config's own model sizes, sequence length, branch points
and depth (SURVEY.md §8 top: C1 Pythia-70M G + D, S 128, K 16, D 2; C2 GPT-2 small G + D, S 256, K 32, D 4; C3 Pythia-410M G /
Pythia-14M D, S 128, K 32, D 8 - ConfigInstances/PileTraining.py:46-130, the last depth of Training/DepthSchedule.py:39-50; C4
Pythia-1B G / 410M D, S 512, K 64, D 8).

C1-C3: the eager fp32 step (fused block path, one-pass generator gradient, merged / tree discriminator pass, fused losses)
against oracle.gan.train_forward - the CPU restatement of Models/GanSetup.py:193-258 that walks the tree node by node like the
reference - on the SAME weights, token ids and branch points.  Batch 1-2 keeps the CPU side to seconds; everything else is
the config's.  Bar: sampled branch tokens identical; every loss within 5e-5; every parameter gradient within
rtol 2e-3 / atol 5e-5 of the oracle's (fp32 GEMM reduction order differs between hipBLASLt and ATen's CPU kernels), the same
tolerance tests/test_gpu_tree.py holds the tiny reference goldens to.
C4 (attention-bound stress, Dh = 256): too large for the CPU oracle in a test; the one-pass schedule is compared with the
node-by-node schedule (the reference's order of operations) on the GPU, plus finiteness and distinct-id properties."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

def _c4_shapes():
    """c4's SHAPES on the oracle's budget: Pythia-1B generator dims (h 2048, 8 heads of Dh 256, rotary 64) and Pythia-410M
    discriminator dims (h 1024, 16 heads of Dh 64) with TWO layers each, at c4's S 512, K 64, D 8 - every attention launch
    shape of c4 (root causal at S 512, branch steps with 1-6 ancestor slots, the 7-node tree pass; the Dh 256 forward and
    its backward) against the CPU restatement."""
    from branchgan.configs import LMConfig
    return dict(gen=LMConfig("gpt_neox", 50304, 2048, 2, 8, 2048), disc=LMConfig("gpt_neox", 50304, 1024, 2, 16, 2048),
                S=512, K=64, D=8, B=1)


CONFIGS = {
    "c1": dict(gen="pythia-70m", disc="pythia-70m", S=128, K=16, D=2, B=4),      # the config's own batch
    "c2": dict(gen="gpt2-small", disc="gpt2-small", S=256, K=32, D=4, B=1),
    "c3": dict(gen="pythia-410m", disc="pythia-14m", S=128, K=32, D=8, B=1),
    "c4-shapes": _c4_shapes,
}


def _oracle_cfg(name):
    from branchgan import configs
    from oracle import models as OM
    c = configs.LMConfig.from_any(name)
    return OM.LMConfig("neox" if c.model_type == "gpt_neox" else "gpt2", c.num_hidden_layers, c.num_attention_heads, c.hidden_size,
                       c.vocab_size, c.max_position_embeddings)


def _build(wl, seed=0):
    from branchgan import configs, gan, sampling
    torch.manual_seed(seed)
    model = gan.GannSetup(wl["gen"], wl["disc"], mleLossWeight=1, rlLossWeight=1, discValueHeadLossWeight=0.25, device="cuda")
    rules = [sampling.SamplingRule(1, sampling.SamplingTechniques.TOP_K, {}) for _ in range(wl["D"])]
    sparse = np.sort(np.random.RandomState(1234).choice(wl["S"] - 1, wl["K"], replace=False))
    conf = configs.SamplingConfig(rules, includeAllMultiStepSamplesForDisc=True, numSampleSequences=wl["K"], sparseIndex=torch.tensor(sparse))
    rl = configs.ReinforcementLearningConfig(labelMode=configs.ReinforcementLabelMode.ALTERATIONS_AND_INPUT_UNWEIGHTED,
                                             sectionWeightA=1, sectionWeightB=1, sectionWeightC=1.5, teacherForcing=True)
    V = configs.LMConfig.from_any(wl["gen"]).vocab_size
    ids = torch.randint(0, V, (wl["B"], wl["S"]), generator=torch.Generator().manual_seed(7))
    return model, conf, rl, ids, sparse


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_full_size_step_matches_the_oracle(name):
    from branchgan import tree
    from oracle import gan as OG
    wl = CONFIGS[name]() if callable(CONFIGS[name]) else CONFIGS[name]
    model, conf, rl, ids, sparse = _build(wl)
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    torch.set_num_threads(16)
    d_ref, g_ref, extra = OG.train_forward(sd, _oracle_cfg(wl["gen"]), _oracle_cfg(wl["disc"]), ids, [(1, 2, {})] * wl["D"], sparse,
                                           rl=dict(label_mode=3, wA=1, wB=1, wC=1.5, teacher_forcing=True))
    g_ref["loss"].backward()
    d_ref["loss"].backward()

    model = model.cuda()
    conf.sparseIndex = conf.sparseIndex.cuda()
    with torch.no_grad():                    # the sampled branch tokens, node by node (pre-order, like root.getSubNodes())
        root, *_ = tree.multiStepGenForward(model.generator, ids.cuda(), conf.rules, "cuda", sparseIndex=conf.sparseIndex, clearTreePast=True)
        got_ids = [n.ids.cpu() for n in root.getSubNodes()]
    want_ids = [n.ids for n in extra["nodes"]]
    assert len(got_ids) == len(want_ids) == wl["D"]
    for i, (a, b) in enumerate(zip(got_ids, want_ids)):
        assert torch.equal(a, b), f"{name}: sampled ids of node {i} differ in {int((a != b).sum())} of {a.numel()} places"

    d, g = model.trainForwardPassNoMetrics(ids.cuda(), conf, 9999, collectTreeDepthData=False, reinforcementLearningConfig=rl)
    for k in ("loss", "predLoss", "loss-Input", "loss-Alterations", "ValueHeadLoss"):
        np.testing.assert_allclose(float(d[k].detach()), float(d_ref[k].detach()), atol=5e-5, rtol=0, err_msg=f"{name} disc {k}")
    for k in ("loss", "loss-MLE", "loss-RL", "loss-RL-input", "loss-RL-alterations"):
        np.testing.assert_allclose(float(g[k].detach()), float(g_ref[k].detach()), atol=5e-5, rtol=0, err_msg=f"{name} gen {k}")
    g["loss"].backward()
    d["loss"].backward()
    torch.cuda.synchronize()
    checked, worst = 0, 0.0
    for k, p in model.named_parameters():
        want = sd[k].grad
        if want is None or not bool(want.any()):
            assert p.grad is None or not bool(p.grad.any()), k
            continue
        got = p.grad.cpu()
        err = (got - want).abs()
        bound = 5e-5 + 2e-3 * want.abs()
        worst = max(worst, float((err / bound).max()))
        assert bool((err <= bound).all()), f"{name} grad {k}: max |err| {float(err.max()):.3e} (max |grad| {float(want.abs().max()):.3e})"
        checked += 1
    assert checked > 40
    print(f"{name}: {checked} gradient tensors, worst |err| / (5e-5 + 2e-3 |grad|) = {worst:.3f}")


def test_c4_step_one_pass_equals_node_by_node(monkeypatch):
    """BASELINE configs[3]: Pythia-1B generator + Pythia-410M discriminator, seq 512, 64 branch points, depth 8 (Dh = 256)."""
    from branchgan import ops
    wl = dict(gen="pythia-1b", disc="pythia-410m", S=512, K=64, D=8, B=1)
    model, conf, rl, ids, sparse = _build(wl)
    model = model.cuda()
    conf.sparseIndex = conf.sparseIndex.cuda()
    ids = ids.cuda()
    results = []
    for schedule in ("one-pass", "node-by-node"):
        if schedule == "node-by-node":
            monkeypatch.setattr(ops, "MAX_SLOTS", 0)     # generator re-forwards node by node with autograd; discriminator: root pass + tree pass
        for p in model.parameters():
            p.grad = None
        d, g = model.trainForwardPassNoMetrics(ids, conf, 9999, collectTreeDepthData=False, reinforcementLearningConfig=rl)
        g["loss"].backward()
        d["loss"].backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        results.append((float(d["loss"].detach()), float(g["loss"].detach()), float(g["loss-RL"].detach()), grads))
    (d1, g1, r1, grads1), (d2, g2, r2, grads2) = results
    assert all(np.isfinite(x) for x in (d1, g1, r1))
    np.testing.assert_allclose((d1, g1, r1), (d2, g2, r2), rtol=1e-5, atol=1e-6)
    assert set(grads1) == set(grads2) and len(grads1) > 300
    for k in grads1:
        a, b = grads1[k], grads2[k]
        assert bool(torch.isfinite(a).all()), k
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 1e-4 * scale + 1e-7, f"{k}: {float((a - b).abs().max()):.3e} vs max |grad| {scale:.3e}"
    ga = torch.sqrt(sum((v.double() ** 2).sum() for v in grads1.values()))
    gb = torch.sqrt(sum((v.double() ** 2).sum() for v in grads2.values()))
    np.testing.assert_allclose(float(ga), float(gb), rtol=1e-5)
    # the sampled branch tokens of one row are distinct positions' tokens, all inside the vocabulary
    from branchgan import tree
    with torch.no_grad():
        root, *_ = tree.multiStepGenForward(model.generator, ids, conf.rules, "cuda", sparseIndex=conf.sparseIndex, clearTreePast=True)
    nodes = root.getSubNodes()
    assert len(nodes) == 8 and all(n.ids.shape == (1, 64) for n in nodes)
    assert all(int(n.ids.min()) >= 0 and int(n.ids.max()) < 50304 for n in nodes)
    first = nodes[0].ids[0].cpu()
    labels = ids[0, torch.as_tensor(sparse) + 1].cpu()
    assert bool((first != labels).all())                 # step 1 blocks the true next token (SamplingStrategies.py:258-263)
