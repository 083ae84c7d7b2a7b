"""GPU parity of the whole hot path against the tiny-model fixtures captured from the reference's GannSetup
(tests/golden/g35_*.npz): the reference's own state dict is loaded by key name, then per-node sampled ids (bit-exact),
probabilities, discriminator scores, value heads, both loss dicts and every parameter gradient are compared.
Tolerance: fp32 end to end, |err| <= 5e-5 on O(1) quantities (GEMM reduction order through hipBLASLt differs from ATen's CPU
kernels), gradients rtol 2e-3 / atol 5e-5."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# neox_rand: random children (EPSILON_GREEDY), firstBeamSamplingStrategy / Params, continueFromRandom=False
# (MultiStepPrediction.py:292,302-307); the random ids come from torch's global CPU generator, seeded like the capture
CASES = [("neox_chain4", "gpt_neox"), ("neox_tree211", "gpt_neox"), ("gpt2_chain3", "gpt2"), ("gpt2_tree21", "gpt2"), ("neox_rand", "gpt_neox")]


def _setup(g, family):
    from branchgan import configs, gan, sampling
    mk = lambda hidden, heads: configs.LMConfig(family, 97, hidden, 2, heads, 64, intermediate_size=2 * hidden if family == "gpt_neox" else 0)
    setup = gan.GannSetup(mk(64, 2), mk(32, 1))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    setup.load_state_dict(sd, strict=True)
    setup = setup.cuda().eval()
    rules = [sampling.SamplingRule(int(a), sampling.SamplingTechniques(int(t)), {}) for a, t in g["rules"].tolist()]
    if "rules_json" in g.files:
        import json
        rules = []
        for spec in json.loads(str(g["rules_json"])):
            o = spec[3] if len(spec) > 3 else {}
            fb = o.get("firstBeam")
            rules.append(sampling.SamplingRule(int(spec[0]), sampling.SamplingTechniques(int(spec[1])), dict(spec[2]),
                                               continueFromRandom=o.get("continueFromRandom", False),
                                               firstBeamSamplingStrategy=sampling.SamplingTechniques(int(fb[0])) if fb else None,
                                               firstBeamSamplingParams=dict(fb[1]) if fb else None))
    ids, sparse = torch.from_numpy(g["inputIDs"]).cuda(), torch.from_numpy(g["sparseIndex"])
    noises = [torch.from_numpy(g[f"noise{i}"]) for i in range(sum(k.startswith("noise") for k in g.files))]
    calls = {"i": 0}

    def noise_fn(shape, device):
        t = noises[calls["i"]]
        calls["i"] += 1
        if t.shape[1] != shape[1]:            # the reference samples all S rows at step 1; only the K branch rows are used
            t = t[:, sparse]
        assert tuple(t.shape) == tuple(shape)
        return t.to(device)

    setup.noise_fn = noise_fn
    conf = configs.SamplingConfig(rules, numSampleSequences=len(sparse), sparseIndex=sparse)
    return setup, conf, ids, sparse, calls


@pytest.mark.parametrize("tag,family", CASES)
def test_tree_nodes_match_reference(golden, tag, family):
    from branchgan import tree
    g = golden("g35_" + tag)
    setup, conf, ids, sparse, calls = _setup(g, family)
    if "torch_seed" in g.files:
        torch.manual_seed(int(g["torch_seed"]))
    with torch.no_grad():
        root, firstPred, _, _ = tree.multiStepGenForward(setup.generator, ids, conf.rules, "cuda", sparseIndex=sparse,
                                                         noise_fn=setup.noise_fn, clearTreePast=False)
        np.testing.assert_allclose(firstPred.logits.cpu().numpy(), g["gen_logits"], atol=5e-5)
        np.testing.assert_allclose(root.probs.cpu().numpy(), g["gen_root_probs"], atol=5e-6)
        nodes = root.getSubNodes()
        for i, n in enumerate(nodes):
            assert np.array_equal(n.ids.cpu().numpy(), g[f"gen_node{i}_ids"]), (tag, i)
            np.testing.assert_allclose(n.probs.cpu().numpy(), g[f"gen_node{i}_probs"], atol=5e-6)
            depth, beam, is_random, parent = g[f"gen_node{i}_meta"].tolist()
            assert (n.depth, n.beam, int(n.isRandom)) == (depth, beam, is_random)
            assert (-1 if n.parent is root else nodes.index(n.parent)) == parent
            key = f"gen_node{i}_attmask"
            if key in g.files:                 # the arithmetic descriptor expands to the reference's mask tensor
                assert np.array_equal(n.getAttentionMask(ids.shape[0], "cpu").numpy(), g[key])
        tree.multiStepDiscForward(setup.discriminator, root, "cuda", dualValueHeads=True, clearTreePast=False)
        np.testing.assert_allclose(root.discPreds.prediction.cpu().numpy(), g["disc_root_pred"], atol=5e-6)
        np.testing.assert_allclose(root.valueHeadData["InputValue"].cpu().numpy(), g["disc_root_inV"], atol=5e-6)
        for i, n in enumerate(nodes):          # ONE forward over all nodes == the reference's node-by-node walk
            np.testing.assert_allclose(n.discPreds.prediction.cpu().numpy(), g[f"disc_node{i}_pred"], atol=5e-6, err_msg=f"{tag} node {i}")
            np.testing.assert_allclose(n.valueHeadData["InputValue"].cpu().numpy(), g[f"disc_node{i}_inV"], atol=5e-6)
            np.testing.assert_allclose(n.valueHeadData["AlterationsValue"].cpu().numpy(), g[f"disc_node{i}_altV"], atol=5e-6)


@pytest.mark.parametrize("schedule", ["one-pass", "node-by-node"])
@pytest.mark.parametrize("tag,family", CASES)
def test_train_step_losses_and_grads_match_reference(golden, tag, family, schedule, monkeypatch):
    """schedule "node-by-node": trees wider than one attention call can address (ops.MAX_SLOTS) take the fallbacks - the
    generator re-forwards node by node with autograd, the discriminator runs root pass + tree pass separately."""
    from branchgan import configs, ops
    if schedule == "node-by-node":
        monkeypatch.setattr(ops, "MAX_SLOTS", 0)
    g = golden("g35_" + tag)
    setup, conf, ids, sparse, calls = _setup(g, family)
    rl = configs.ReinforcementLearningConfig(labelMode=configs.ReinforcementLabelMode.ALTERATIONS_AND_INPUT_UNWEIGHTED,
                                             sectionWeightA=1, sectionWeightB=1, sectionWeightC=1.5, teacherForcing=True)
    if "torch_seed" in g.files:
        torch.manual_seed(int(g["torch_seed"]))
    d, gl = setup.trainForwardPassNoMetrics(ids, conf, 9999, collectTreeDepthData=True, reinforcementLearningConfig=rl)
    for k in g.files:
        if k.startswith("dloss::"):
            key = k[7:]
            got = d[key]
            got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
            np.testing.assert_allclose(got, g[k], atol=5e-5, rtol=1e-4, err_msg=key)
        if k.startswith("gloss::"):            # every key, the 20 debugging stats of GeneratorLosses.py:34-81 included
            np.testing.assert_allclose(gl[k[7:]].detach().cpu().numpy(), g[k], atol=5e-5, rtol=1e-4, err_msg=k)
    assert set(gl) == {k[7:] for k in g.files if k.startswith("gloss::")} and len(gl) == 27
    gl["loss"].backward()
    d["loss"].backward()
    params = dict(setup.named_parameters())
    n_checked = 0
    for k in g.files:
        if not k.startswith("grad::"):
            continue
        want, got = g[k], params[k[6:]].grad
        if want.size == 0 or not np.any(want):
            assert got is None or not bool(torch.any(got)), k
            continue
        np.testing.assert_allclose(got.cpu().numpy(), want, atol=5e-5, rtol=2e-3, err_msg=k)
        n_checked += 1
    assert n_checked > 40
