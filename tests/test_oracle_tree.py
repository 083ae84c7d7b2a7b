"""Pins oracle/{models,tree,gan}.py against the tiny-model tree and end-to-end fixtures captured from the
reference's GannSetup (tests/golden/g35_*.npz): per-node ids/probs/masks, discriminator predictions, value
heads, both loss dicts and every parameter gradient."""
import numpy as np
import pytest
import torch

from oracle import gan as G
from oracle import models as M
from oracle import tree as T

# neox_rand: the tree options of MultiStepPrediction.py:292,302-307 - random children (EPSILON_GREEDY draws from torch's global
# generator, seeded like the capture), a special rule for beam 0 (firstBeamSamplingStrategy), random leaves that do not continue
CASES = [("neox_chain4", "neox"), ("neox_tree211", "neox"), ("gpt2_chain3", "gpt2"), ("gpt2_tree21", "gpt2"), ("neox_rand", "neox")]


class NoiseReplay:
    """Serves the noise tensors the reference consumed, in call order."""

    def __init__(self, g):
        self.items = [torch.from_numpy(g[f"noise{i}"]) for i in range(sum(k.startswith("noise") for k in g.files))]
        self.i = 0

    def __call__(self, shape):
        t = self.items[self.i]
        self.i += 1
        assert tuple(t.shape) == tuple(shape)
        return t


def load_case(g, family, requires_grad=False):
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    if requires_grad:
        sd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    cfg_g = M.config_from_state_dict(sd, "generator.generator.", family, n_head=2)
    cfg_d = M.config_from_state_dict(sd, "discriminator.nonResidualModel.", family, n_head=1)
    rules = [(int(a), int(t), {}) for a, t in g["rules"].tolist()]
    if "rules_json" in g.files:
        import json
        rules = []
        for spec in json.loads(str(g["rules_json"])):
            o = spec[3] if len(spec) > 3 else {}
            opts = {"continue_from_random": o.get("continueFromRandom", False)}
            if o.get("firstBeam"):
                opts["first_beam"] = (int(o["firstBeam"][0]), dict(o["firstBeam"][1]))
            rules.append((int(spec[0]), int(spec[1]), dict(spec[2]), opts))
        torch.manual_seed(int(g["torch_seed"]))
    return sd, cfg_g, cfg_d, rules, torch.from_numpy(g["inputIDs"]), torch.from_numpy(g["sparseIndex"])


@pytest.mark.parametrize("tag,family", CASES)
def test_tree_matches_reference(golden, tag, family):
    g = golden("g35_" + tag)
    sd, cfg_g, cfg_d, rules, ids, sparse = load_case(g, family)
    with torch.no_grad():
        root, logits, probs, _ = T.multi_step_gen_forward(sd, cfg_g, ids, rules, sparse, NoiseReplay(g))
        np.testing.assert_allclose(logits.numpy(), g["gen_logits"], atol=2e-6)
        np.testing.assert_allclose(root.probs.numpy(), g["gen_root_probs"], atol=1e-7)
        nodes = root.sub_nodes()
        for i, n in enumerate(nodes):
            assert np.array_equal(n.ids.numpy(), g[f"gen_node{i}_ids"]), i
            np.testing.assert_allclose(n.probs.numpy(), g[f"gen_node{i}_probs"], atol=1e-6)
            depth, beam, is_random, parent = g[f"gen_node{i}_meta"].tolist()
            assert (n.depth, n.beam, int(n.is_random)) == (depth, beam, is_random)
            assert (-1 if n.parent is root else nodes.index(n.parent)) == parent
            key = f"gen_node{i}_attmask"
            if key in g.files:
                assert np.array_equal(T.node_mask(n, ids.shape[0], sparse, ids.shape[1]).numpy(), g[key])
        T.multi_step_disc_forward(sd, cfg_d, root, sparse)
        np.testing.assert_allclose(root.disc_pred.numpy(), g["disc_root_pred"], atol=1e-6)
        np.testing.assert_allclose(root.in_value.numpy(), g["disc_root_inV"], atol=1e-6)
        for i, n in enumerate(nodes):
            np.testing.assert_allclose(n.disc_pred.numpy(), g[f"disc_node{i}_pred"], atol=1e-6)
            np.testing.assert_allclose(n.alt_value.numpy(), g[f"disc_node{i}_altV"], atol=1e-6)


@pytest.mark.parametrize("tag,family", CASES)
def test_train_step_losses_and_grads_match_reference(golden, tag, family):
    g = golden("g35_" + tag)
    sd, cfg_g, cfg_d, rules, ids, sparse = load_case(g, family, requires_grad=True)
    rl = dict(label_mode=3, wA=1, wB=1, wC=1.5, teacher_forcing=True)
    d, gl, _ = G.train_forward(sd, cfg_g, cfg_d, ids, rules, sparse, NoiseReplay(g), rl=rl)
    for k in ["loss", "predLoss", "loss-Input", "loss-Alterations", "ValueHeadLoss", "ValueHead-InputLoss", "ValueHead-AltLoss"]:
        np.testing.assert_allclose(d[k].detach().numpy(), g["dloss::" + k], atol=1e-6, err_msg=k)
    gen_keys = [k[7:] for k in g.files if k.startswith("gloss::")]
    assert len(gen_keys) == 27 and set(gen_keys) == set(gl)          # incl. the 20 debugging stats (GeneratorLosses.py:34-81)
    for k in gen_keys:
        np.testing.assert_allclose(gl[k].detach().numpy(), g["gloss::" + k], atol=2e-6, err_msg=k)
    gl["loss"].backward()
    d["loss"].backward()
    n_checked = 0
    for k in g.files:
        if not k.startswith("grad::"):
            continue
        want, got = g[k], sd[k[6:]].grad
        if want.size == 0 or not np.any(want):
            assert got is None or not torch.any(got), k       # unused heads: generator valueHead, disc embed_out/lm_head
            continue
        np.testing.assert_allclose(got.numpy(), want, atol=2e-5, rtol=1e-4, err_msg=k)
        n_checked += 1
    assert n_checked > 40
