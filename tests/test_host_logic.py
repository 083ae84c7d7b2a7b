"""CPU tests of the host-side mirror of the reference interface (no kernels are launched)."""
import numpy as np
import pytest
import torch

from oracle import attention as A
from oracle import tree as OT


def test_sampling_enum_and_rule_surface():
    from branchgan import sampling as SS
    assert [t.value for t in SS.SamplingTechniques] == list(range(10))
    assert set(SS.RANDOM_TECHNIQUES) == {"RANDOM", "EPSILON_GREEDY", "EPSILON_SAMPLE", "EPSILON_SAMPLE_NO_REPLACEMENT"}
    r = SS.SamplingRule(2, SS.SamplingTechniques.TOP_K, {"a": 1})
    assert r.to_dict()["Strategy"] == "TOP_K" and r.firstBeamSamplingParams == {}
    with pytest.raises(ValueError):
        SS.apply_temperature(torch.ones(2, 3), 0)
    with pytest.raises(ValueError):
        SS.apply_noise(torch.ones(2, 3), 1.5)
    p = torch.softmax(torch.randn(2, 3, 7), -1)
    assert SS.apply_temperature(p, 1) is p


def test_row_selection_matches_reference_blocking():
    from branchgan import sampling as SS
    ids = torch.arange(12).view(2, 6)
    for block in (True, False):
        for final in (True, False):
            T_out, row_map, blocked = SS._row_selection(2, 6, 50, ids, block, final, "cpu")
            assert T_out == (6 if final else 5)
            assert (row_map is None) == final
            if block:
                want = ids[:, 1:6]
                got = blocked.view(2, T_out)
                assert torch.equal(got[:, :5], want)
                if final:
                    assert (got[:, 5] == -1).all()
            else:
                assert blocked is None


def test_bool_mask_descriptor_roundtrip_and_rejection():
    from branchgan.models import descriptor_from_bool_mask
    sparse = torch.tensor([1, 4, 8])
    m = OT.depth1_mask(2, sparse, 10)
    for d in range(1, 4):
        C = 10 + (d - 1) * 3
        ctx, n_anc = descriptor_from_bool_mask(m, 3, C)
        assert ctx.tolist() == [2, 5, 9] and n_anc == d - 1
        m = OT.deeper_mask(m, 2, 3)
    bad = OT.depth1_mask(2, sparse, 10).clone()
    bad[0, 0, 1, 0] = False                         # hole in a text prefix
    with pytest.raises(NotImplementedError):
        descriptor_from_bool_mask(bad, 3, 10)
    uneven = OT.depth1_mask(2, sparse, 10).clone()
    uneven[1, 0, 0, 5] = True                       # differs across the batch
    with pytest.raises(NotImplementedError):
        descriptor_from_bool_mask(uneven, 3, 10)


def test_tree_node_mask_and_structure_helpers():
    from branchgan import tree
    ids = torch.zeros(2, 10, dtype=torch.long)
    sparse = torch.tensor([1, 4, 8])
    root = tree.PredictionTreeNode(ids, sparseIndex=sparse, probs=torch.zeros(2, 10))
    root.addGeneratorPredictionToNode(torch.zeros(2, 2, 3, dtype=torch.long), None, None, torch.zeros(2, 3, 2))
    a, b = root.children
    a.addGeneratorPredictionToNode(torch.zeros(2, 1, 3, dtype=torch.long), None, None, torch.zeros(2, 3, 1))
    nodes = root.getSubNodes()
    assert [n.depth for n in nodes] == [1, 2, 1] and [n.beam for n in nodes] == [0, 0, 1]
    assert tree.getSubNodeListTreeStructure(root, nodes) == ([0, 2], [[1], [], []])
    assert [n.depth for n in root.getLeaves()] == [2, 1]
    # the arithmetic mask equals the reference's tensor mask
    want = OT.deeper_mask(OT.depth1_mask(2, sparse, 10), 2, 3)
    assert torch.equal(nodes[1].getAttentionMask(2, "cpu"), want)
    assert nodes[1].getPaddedPosIDs().tolist() == [3, 6, 10] + [9999] * 7
    assert tree.getRandomNodesFromRule(tree.SamplingTechniques.TOP_K, {}, 2) is None
    assert tree.getRandomNodesFromRule(tree.SamplingTechniques.EPSILON_GREEDY, {"numRandomSamples": 1}, 3) == [False, False, True]


def test_configs_and_schedules():
    from branchgan import configs, sampling as SS
    rules = [SS.SamplingRule(1, SS.SamplingTechniques.TOP_K, {}) for _ in range(16)]
    conf = configs.SamplingConfig(rules, numSampleSequences=32, seed=7)
    ids = torch.zeros(8, 128, dtype=torch.long)
    sp = conf.getSparseIndex(ids, "cpu")
    assert sp.numel() == 32 and bool((sp[1:] > sp[:-1]).all()) and int(sp.max()) <= 126
    sched = configs.LinearUpdatesDepthSchedule(1, 1, 100, 8)
    sched.updateSchedule(0, 0, 250)
    assert sched.currentDepth == 3 and len(sched.setSampleConfDepth(conf).rules) == 3 and len(conf.rules) == 16
    sched.updateSchedule(0, 0, 5000)
    assert sched.currentDepth == 8
    rr = configs.RoundRobinSamplingConfig([conf, "x"])
    assert rr.getNextSampleConf() == "x" and rr.getNextSampleConf() is conf
    c = configs.LMConfig.from_any("pythia-410m")
    assert (c.head_dim, c.rotary_ndims, c.intermediate_size) == (64, 16, 4096)
    assert configs.LMConfig.from_any("gpt2-small").vocab_size == 50257


def test_state_dict_keys_match_reference(golden):
    """Reference checkpoints load by key name: compare against the keys captured from the reference."""
    from branchgan import configs, gan
    for tag, family in (("neox_chain4", "gpt_neox"), ("gpt2_chain3", "gpt2")):
        g = golden("g35_" + tag)
        ref_keys = {k[4:] for k in g.files if k.startswith("sd::")}
        mk = lambda hidden, heads: configs.LMConfig(family, 97, hidden, 2, heads, 64, intermediate_size=2 * hidden if family == "gpt_neox" else 0)
        setup = gan.GannSetup(mk(64, 2), mk(32, 1), device="cpu")
        mine = set(setup.state_dict().keys())
        mask_buffers = {k for k in ref_keys if k.endswith((".bias", ".self_bias", ".masked_bias")) and (".attention." in k or ".attn." in k)
                        and not k.endswith(("c_attn.bias", "c_proj.bias", "query_key_value.bias", "dense.bias"))}
        assert mine == ref_keys - mask_buffers
        setup.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}, strict=True)


def test_worker_batch_size_split():
    from branchgan.training import worker_batch_size
    assert [worker_batch_size(8, 3, r) for r in range(3)] == [3, 3, 2]
    assert [worker_batch_size(64, 8, r) for r in range(8)] == [8] * 8


def test_restack_of_all_slices_is_the_whole_and_anything_else_a_stack():
    """tree.part_of / tree.restack: the slices 0 .. N-1 of one tensor, in order, restack to that tensor (a
    view: same values, gradient handed over in one piece); a subset, another order or an untagged piece fall back to torch.stack."""
    from branchgan import tree as T
    whole = torch.randn(2, 4, 3, requires_grad=True)
    parts = [T.part_of(whole, 1, i) for i in range(4)]
    for dim in (0, 1, 2):
        got, ref = T.restack(parts, dim), torch.stack([whole[:, i] for i in range(4)], dim=dim)
        assert torch.equal(got, ref)
        assert got._base is whole or got is whole                       # no copy was made
    w = torch.randn(4, 2, 3)
    g_view, = torch.autograd.grad((T.restack(parts, 0) * w).sum(), whole)
    g_ref, = torch.autograd.grad((torch.stack([whole[:, i] for i in range(4)], dim=0) * w).sum(), whole)
    assert torch.equal(g_view, g_ref)
    for other in (parts[:3], parts[::-1], parts[:3] + [whole[:, 3]], [T.part_of(whole, 1, i) for i in (0, 1, 2, 2)]):
        got = T.restack(other, 1)
        assert got._base is None and torch.equal(got, torch.stack(list(other), dim=1))
    two = [T.part_of(whole, 1, 0), T.part_of(whole.detach().clone(), 1, 1)] + parts[2:]
    assert T.restack(two, 1)._base is None                              # slices of different tensors


def test_restack_joins_runs_of_nested_slices_with_one_cat():
    """The generator's log-probs: node 1 is a slice of the root pass's (B, K, n) tensor, nodes 2 .. N slices of the tree pass's
    (B, N, K, n) tensor picked twice (node, then child): restack takes the second run as one view and joins the runs by a cat."""
    from branchgan import tree as T
    a = torch.randn(2, 3, 1, requires_grad=True)                  # root level: (B, K, n_root)
    t4 = torch.randn(2, 4, 3, 2, requires_grad=True)              # tree pass: (B, N, K, n_max)
    parts = [T.part_of(a, 2, 0)] + [T.part_of(T.part_of(t4, 1, i), 2, 1) for i in range(4)]
    ref = torch.stack([a[:, :, 0]] + [t4[:, i][:, :, 1] for i in range(4)], dim=1)
    got = T.restack(parts, 1)
    assert torch.equal(got, ref)
    w = torch.randn_like(ref)
    assert all(torch.equal(x, y) for x, y in zip(torch.autograd.grad((got * w).sum(), (a, t4)), torch.autograd.grad((ref * w).sum(), (a, t4))))
    run = T.restack(parts[1:], 0)                                 # one run alone: a view of the tree pass's tensor
    assert run._base is t4 and torch.equal(run, torch.stack([t4[:, i][:, :, 1] for i in range(4)], dim=0))
    # children picked at different positions, or nodes out of order: plain stacks inside the cat, same values
    mixed = [T.part_of(T.part_of(t4, 1, i), 2, i % 2) for i in range(4)]
    assert torch.equal(T.restack(mixed, 1), torch.stack([t4[:, i][:, :, i % 2] for i in range(4)], dim=1))
    swapped = [parts[2], parts[1], parts[3], parts[4]]
    assert torch.equal(T.restack(swapped, 1), torch.stack([t4[:, i][:, :, 1] for i in (1, 0, 2, 3)], dim=1))


def test_two_large_models_do_not_replay_their_backward_graphs_side_by_side():
    """graphs._both_models_large: the guard against two stream-K GEMMs of two streams waiting for each other."""
    from types import SimpleNamespace
    from branchgan import graphs

    def part(h):
        m = torch.nn.Linear(2, 2)
        m.config = SimpleNamespace(hidden_size=h)
        return torch.nn.Sequential(m)

    pair = lambda g, d: SimpleNamespace(generator=part(g), discriminator=part(d))
    assert graphs._both_models_large(pair(2048, 1024))              # configs[3]
    assert not graphs._both_models_large(pair(768, 768))            # configs[1]
    assert not graphs._both_models_large(pair(1024, 128))           # configs[2]
    assert not graphs._both_models_large(SimpleNamespace(generator=torch.nn.Linear(2, 2), discriminator=part(4096)))
