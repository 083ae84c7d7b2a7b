#!/usr/bin/env python3
"""Capture golden input/output vectors from the real reference (FreddeFrallan/Branch-GAN).

Runs ONLY in the build container, where the reference is mounted read-only at /root/reference.
It imports the reference on CPU through the three harness-side shims recorded in SURVEY.md §8(c)
(nothing in the reference is modified, nothing is fetched, no pretrained name is resolved) and
writes small `.npz` fixtures next to this file.  The GPU box never sees the reference: tests read
only the fixtures.  A fixture is data (inputs + expected outputs); no reference source is stored.

Usage:  python tests/golden/make_goldens.py            (rewrites every fixture, deterministic seeds)

Groups:  G1 attention, G2 sampling, G3 tree, G4 losses, G5 end-to-end tiny step.
"""
import os
import sys

sys.dont_write_bytecode = True
REF = os.environ.get("BRANCHGAN_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np
import torch
import transformers
import transformers.modeling_utils as mu
import transformers.pytorch_utils as pu

# --- shims for transformers 5.x -------------------------------------
mu.Conv1D = pu.Conv1D
for _n in ("SequenceSummary", "find_pruneable_heads_and_indices", "prune_conv1d_layer"):
    if not hasattr(mu, _n):
        setattr(mu, _n, None)
mu.PreTrainedModel.get_head_mask = lambda self, hm, n, *a: [None] * n

from transformers import GPT2Config, GPTNeoXConfig  # noqa: E402

from BranchPredictions import MultiStepPrediction as MSP  # noqa: E402
from BranchPredictions import SamplingStrategies as SS  # noqa: E402
from Configs import ReinforceLabelMode as RLM  # noqa: E402
from Configs import TrainingConfig  # noqa: E402
from Models import GanSetup  # noqa: E402
from Models.GPT2BranchAttention.NonResidualAttention import GPT2NonResidualAttention  # noqa: E402
from Models.Losses import DiscriminatorLoss, GeneratorLosses  # noqa: E402
from Models.NeoXBranchAttention.NeoXNonResidualAttention import NeoXNonResidualAttention  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
T = SS.SamplingTechniques


def neox_cfg(**kw):
    c = GPTNeoXConfig(**kw)
    c.rotary_pct = 0.25
    c.rotary_emb_base = 10000
    c.rope_scaling = None
    return c


def tiny_cfg(family, vocab=97, hidden=32, layers=2, heads=2, npos=64):
    if family == "neox":
        return neox_cfg(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=layers,
                        num_attention_heads=heads, intermediate_size=2 * hidden,
                        max_position_embeddings=npos, hidden_dropout=0.0, attention_dropout=0.0)
    return GPT2Config(vocab_size=vocab, n_positions=npos, n_embd=hidden, n_layer=layers, n_head=heads,
                      attn_pdrop=0.0, resid_pdrop=0.0, embd_pdrop=0.0)


def npy(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrs.items()})
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


class StubBuffer:
    """Deterministic stand-in for SamplingStrategies.SAMPLE_BUFFER (the noise injection point)."""

    def __init__(self, seed):
        self.gen = torch.Generator().manual_seed(seed)
        self.served = []

    def rand(self, shape):
        noise = -torch.log(torch.rand(tuple(shape), generator=self.gen) + 1e-14)
        self.served.append(noise)
        return noise


# =================================================================================================
# G1 attention
# =================================================================================================
def build_masks(B, S, sparse, depth):
    """Masks exactly as PredictionTreeNode.getAttentionMask builds them, for depths 1..depth."""
    K = len(sparse)
    root = MSP.PredictionTreeNode(torch.zeros(B, S, dtype=torch.long), sparseIndex=torch.tensor(sparse))
    dummy_past = [(torch.zeros(B, 1, S, 2), torch.zeros(B, 1, S, 2))]
    root.addGeneratorPredictionToNode(torch.zeros(B, 1, K, dtype=torch.long), dummy_past, None,
                                      torch.zeros(B, K, 1))
    node, masks = root.children[0], []
    for d in range(1, depth + 1):
        m = node.getAttentionMask(B, "cpu")
        masks.append(m)
        step_past = [(torch.zeros(B, 1, K, 2), torch.zeros(B, 1, K, 2))]
        node.addGeneratorPredictionToNode(torch.zeros(B, 1, K, dtype=torch.long), step_past, m,
                                          torch.zeros(B, K, 1))
        node = node.children[0]
    return masks


def g1_attention():
    out = {}
    cases = [  # (family, H, Dh, K, S, sparse, depth, extra)
        ("neox", 2, 8, 3, 10, [1, 4, 8], 3, {}),
        ("neox", 4, 32, 16, 33, sorted(np.random.RandomState(1).choice(32, 16, replace=False).tolist()), 3, {}),
        ("neox", 2, 64, 16, 33, sorted(np.random.RandomState(2).choice(32, 16, replace=False).tolist()), 2, {}),
        ("neox", 1, 256, 3, 10, [0, 5, 8], 2, {}),
        ("gpt2", 2, 8, 3, 10, [1, 4, 8], 3, {}),
        ("gpt2", 4, 32, 16, 33, sorted(np.random.RandomState(3).choice(32, 16, replace=False).tolist()), 3, {}),
        ("gpt2", 2, 64, 16, 33, sorted(np.random.RandomState(4).choice(32, 16, replace=False).tolist()), 2,
         {"scale_attn_by_inverse_layer_idx": True, "layer_idx": 2}),
    ]
    B = 2
    meta = []
    for ci, (fam, H, Dh, K, S, sparse, depth, extra) in enumerate(cases):
        torch.manual_seed(100 + ci)
        if fam == "neox":
            mod = NeoXNonResidualAttention(tiny_cfg("neox", hidden=H * Dh, heads=H))
            fn = mod._nonResidualTextAttention
        else:
            cfg = tiny_cfg("gpt2", hidden=H * Dh, heads=H)
            cfg.scale_attn_by_inverse_layer_idx = extra.get("scale_attn_by_inverse_layer_idx", False)
            mod = GPT2NonResidualAttention(cfg, layer_idx=extra.get("layer_idx", 0))
            fn = mod.nonResidualTextAttention
        masks = build_masks(B, S, sparse, depth)
        for d in range(1, depth + 1):
            C = S + (d - 1) * K
            q, k, v = [torch.randn(B, H, K, Dh, requires_grad=True) for _ in range(3)]
            kc, vc = [torch.randn(B, H, C, Dh, requires_grad=True) for _ in range(2)]
            o, w = fn(q, k, v, kc, vc, customCasualMask=masks[d - 1])
            go = torch.randn_like(o)
            grads = torch.autograd.grad(o, [q, k, v, kc, vc], go)
            p = f"c{ci}_d{d}_"
            for nm, t in zip(["q", "k", "v", "kctx", "vctx", "mask", "out", "weights", "gout",
                              "dq", "dk", "dv", "dkctx", "dvctx"],
                             [q, k, v, kc, vc, masks[d - 1], o, w, go, *grads]):
                out[p + nm] = t
        meta.append([ci, 0 if fam == "neox" else 1, H, Dh, K, S, depth,
                     int(extra.get("scale_attn_by_inverse_layer_idx", False)), extra.get("layer_idx", 0)])
        out[f"c{ci}_sparse"] = np.asarray(sparse, dtype=np.int64)
    # default mask (customCasualMask=None): legacy single-step pattern A3
    torch.manual_seed(199)
    for fam in ("neox", "gpt2"):
        H, Dh, Q, Kl = 2, 8, 4, 9
        if fam == "neox":
            fn = NeoXNonResidualAttention(tiny_cfg("neox", hidden=H * Dh, heads=H))._nonResidualTextAttention
        else:
            fn = GPT2NonResidualAttention(tiny_cfg("gpt2", hidden=H * Dh, heads=H), layer_idx=0).nonResidualTextAttention
        q, k, v = [torch.randn(B, H, Q, Dh) for _ in range(3)]
        kc, vc = [torch.randn(B, H, Kl, Dh) for _ in range(2)]
        o, w = fn(q, k, v, kc, vc)
        for nm, t in zip(["q", "k", "v", "kctx", "vctx", "out", "weights"], [q, k, v, kc, vc, o, w]):
            out[f"default_{fam}_{nm}"] = t
    out["meta"] = np.asarray(meta, dtype=np.int64)
    save("g1_attention", **out)


# =================================================================================================
# G2 sampling
# =================================================================================================
def g2_sampling():
    out, index = {}, []
    gen = torch.Generator().manual_seed(2)
    shapes = [(2, 5, 97), (2, 5, 1031), (1, 2, 50304)]
    case = 0
    for si, (B, Tn, V) in enumerate(shapes):
        logits = 3.0 * torch.randn(B, Tn, V, generator=gen)
        probs = torch.softmax(logits, dim=-1)
        if si == 0:  # make one row's top-1 mass > 0.5 and > 0.9 (all-masked top-p row, quirk 7 / S4)
            logits[0, 1, 11] += 30.0
            probs = torch.softmax(logits, dim=-1)
        ids = torch.randint(0, V, (B, Tn), generator=gen)
        out[f"s{si}_probs"], out[f"s{si}_ids"] = probs, ids
        techs = [(T.TOP_K, {}), (T.SAMPLE_NO_REPLACEMENT, {}),
                 (T.TOP_P_SAMPLE_NO_REPLACEMENT, {"topP": 0.5}),
                 (T.TOP_P_SAMPLE_NO_REPLACEMENT, {"topP": 0.9}),
                 (T.INVERSE_TOP_K_SAMPLE_NO_REPLACEMENT, {"inverseTopK": 3})]
        flag_sets = [(True, False), (True, True), (False, False), (False, True)] if si < 2 else [(True, True), (False, True)]
        ns = [1, 2, 16] if si < 2 else [1, 4]
        for tech, params in techs:
            for (block, final) in flag_sets:
                for n in ns:
                    stub = StubBuffer(1000 + 2 * si + int(final))  # one noise tensor per (shape, final)
                    SS.SAMPLE_BUFFER = stub
                    alt, masked = SS.sampleFromGeneratorDistribution(probs, n, ids, tech, blockSamplingOfLabels=block,
                                                                     includeFinalTimeStep=final, **params)
                    out[f"case{case}_alt"] = alt
                    if stub.served:
                        assert len(stub.served) == 1
                        out[f"s{si}_f{int(final)}_noise"] = stub.served[0]
                    if si == 0 and n == 1:
                        out[f"case{case}_masked"] = masked
                    index.append([case, si, tech.value, int(block), int(final), n,
                                  params.get("topP", -1.0), params.get("inverseTopK", -1)])
                    case += 1
    out["index"] = np.asarray(index, dtype=np.float64)
    # apply_temperature
    p = torch.softmax(torch.randn(2, 3, 97, generator=gen), dim=-1)
    out["temp_in"] = p
    out["temp_out_0p7"] = SS.apply_temperature(p, 0.7)
    out["temp_out_1p5"] = SS.apply_temperature(p, 1.5)
    save("g2_sampling", **out)


# =================================================================================================
# G4 losses
# =================================================================================================
class FakeSetup:
    def __init__(self, mle=1.0, rl=1.0):
        self.discriminatorLoss = torch.nn.functional.binary_cross_entropy
        self.discValueHeadLoss = torch.nn.functional.binary_cross_entropy
        self.generatorMleLossWeight = mle
        self.generatorRlLossWeight = rl


def g4_losses():
    out, index = {}, []
    gen = torch.Generator().manual_seed(4)
    B, S, K, V = 3, 12, 5, 53
    # tree: chain of 3 (A=1,1,1) and a branching tree A=(2,1): nodes pre-order [a, a1, b, b1]
    trees = {
        "chain3": dict(N=3, depths=[1, 2, 3], root=[0], links=[[1], [2], []]),
        "tree2x1": dict(N=4, depths=[1, 2, 1, 2], root=[0, 2], links=[[1], [], [3], []]),
    }
    sparse = torch.tensor([1, 4, 6, 9, 10])
    case = 0
    for tname, tr in trees.items():
        N = tr["N"]
        r = lambda *s: torch.rand(*s, generator=gen)
        base = dict(
            inputScores=r(B, S), altScoresK=r(B, N, K), txtIn=r(B, S, 1), txtAlt=r(B, S, 1),
            altIn=r(N, B, K, 1), altAlt=r(N, B, K, 1), altProbsK=r(B, N, K) * 0.5 + 1e-4,
            logits=torch.randn(B, S, V, generator=gen), ids=torch.randint(0, V, (B, S), generator=gen))
        for k, v in base.items():
            out[f"{tname}_{k}"] = v
        depthMask = torch.zeros(N, S)
        for i, d in enumerate(tr["depths"]):
            depthMask[i, :K] = (sparse + d <= S - 1).float()
        out[f"{tname}_depthMask"] = depthMask
        out[f"{tname}_depths"] = np.asarray(tr["depths"])
        out[f"{tname}_rootLinks"] = np.asarray(tr["root"])
        out[f"{tname}_links"] = np.asarray([(l + [-1])[0] if len(l) <= 1 else 1 / 0 for l in tr["links"]])

        for mode in range(4):
            for policy in (0, 1):
                for tf in (True, False):
                    leaf = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in base.items()}
                    pad = lambda x, dim, val=0.0: torch.cat(
                        [x, torch.full(x.shape[:dim] + (S - K,) + x.shape[dim + 1:], val)], dim=dim)
                    altScores = pad(leaf["altScoresK"], 2)                      # GanSetup.py:151-167
                    altAlt = pad(leaf["altAlt"], 2)                             # GanSetup.py:202-207
                    altProbs = pad(leaf["altProbsK"], 2, 0.00001)               # MultiStepPrediction.py:59-65
                    fpd = {"DiscValueHeads": {"txtInputPreds": leaf["txtIn"], "altInputPreds": leaf["altIn"],
                                              "txtAlterationsPreds": leaf["txtAlt"], "altAlterationsPreds": altAlt},
                           "SubNodeStructure": (tr["root"], tr["links"]),
                           "TreeNodeInfo": [(d, False) for d in tr["depths"]],
                           "AlterationProbs": altProbs}
                    setup = FakeSetup()
                    inL, altL, info = DiscriminatorLoss.calcDiscriminatorLoss(
                        setup, leaf["inputScores"], altScores, 9999, depthMask, None,
                        collectTreeDepthData=True, forwardPassData=fpd)
                    rl = RLM.ReinforcementLearningConfig(
                        labelMode=RLM.ReinforcementLabelMode(mode),
                        inputIncorrectValueHeadsPolicy=RLM.IncorrectValueHeadsPolicy(policy),
                        alterationIncorrectValueHeadsPolicy=RLM.IncorrectValueHeadsPolicy(policy),
                        sectionWeightA=1, sectionWeightB=1, sectionWeightC=1.5, teacherForcing=tf)
                    probs = torch.softmax(leaf["logits"], dim=-1)
                    gl = GeneratorLosses.calcGeneratorLoss(setup, leaf["ids"], leaf["logits"], probs,
                                                           leaf["inputScores"].detach(), altScores.detach(), fpd,
                                                           rl, depthMask=depthMask)
                    p = f"case{case}_"
                    discTotal = (inL + altL) / 2 + info["ValueHeadLoss"] * 0.25
                    gd = torch.autograd.grad(discTotal, [leaf["inputScores"], leaf["altScoresK"], leaf["txtIn"],
                                                         leaf["txtAlt"], leaf["altAlt"]], allow_unused=True)
                    gg = torch.autograd.grad(gl["loss"], [leaf["logits"], leaf["altProbsK"]], allow_unused=True)
                    out[p + "disc_inputLoss"], out[p + "disc_altLoss"] = inL, altL
                    for k, v in info.items():
                        out[p + "disc_" + k] = v if isinstance(v, torch.Tensor) else np.asarray(v)
                    for k, v in gl.items():
                        out[p + "gen_" + k] = v if isinstance(v, torch.Tensor) else np.asarray(float(v))
                    for nm, g in zip(["dInputScores", "dAltScoresK", "dTxtIn", "dTxtAlt", "dAltAlt"], gd):
                        out[p + nm] = g if g is not None else np.zeros(0)
                    for nm, g in zip(["dLogits", "dAltProbsK"], gg):
                        out[p + nm] = g if g is not None else np.zeros(0)
                    index.append([case, list(trees).index(tname), mode, policy, int(tf)])
                    case += 1
    out["index"] = np.asarray(index, dtype=np.int64)
    out["sparse"] = sparse
    save("g4_losses", **out)


# =================================================================================================
# G3 tree + G5 end-to-end tiny step
# =================================================================================================
def dump_tree(root, prefix, out):
    nodes = root.getSubNodes()
    for i, n in enumerate(nodes):
        out[f"{prefix}node{i}_ids"] = n.ids
        out[f"{prefix}node{i}_probs"] = n.probs
        out[f"{prefix}node{i}_meta"] = np.asarray([n.depth, n.beam, int(n.isRandom),
                                                   -1 if n.parent is root else nodes.index(n.parent)])
        if n.attMask is not None:
            out[f"{prefix}node{i}_attmask"] = n.attMask
    out[f"{prefix}root_probs"] = root.probs
    return nodes


def g3_g5(family, tag, rules_spec, seed, sparse, B=2, S=12):
    out = {}
    torch.manual_seed(seed)
    cfg_g = tiny_cfg(family, hidden=64, heads=2)      # head dim 32: the smallest the HIP attention kernels take
    cfg_d = tiny_cfg(family, hidden=32, heads=1)
    setup = GanSetup.GannSetup(cfg_g, cfg_d)
    # post_init() leaves heads at N(0, .02): scale up a little so scores are not all ~0.5
    with torch.no_grad():
        for p in setup.parameters():
            p.mul_(3.0)
    setup.eval()
    for k, v in setup.state_dict().items():
        out["sd::" + k] = v
    V = cfg_g.vocab_size
    ids = torch.randint(0, V, (B, S))
    out["inputIDs"], out["sparseIndex"] = ids, np.asarray(sparse, dtype=np.int64)

    def mk_rules():
        """(numAlterations, technique, params[, options]); options: continueFromRandom, firstBeam = (technique, params)
        (SamplingStrategies.py:53-63; used by MultiStepPrediction.py:292,302-307)."""
        rules = []
        for spec in rules_spec:
            a, t, p = spec[:3]
            o = spec[3] if len(spec) > 3 else {}
            fb = o.get("firstBeam")
            rules.append(SS.SamplingRule(a, T(t), dict(p), continueFromRandom=o.get("continueFromRandom", False),
                                         firstBeamSamplingStrategy=T(fb[0]) if fb else None,
                                         firstBeamSamplingParams=dict(fb[1]) if fb else None))
        return rules

    out["rules"] = np.asarray([[spec[0], spec[1]] for spec in rules_spec], dtype=np.int64)
    if any(len(spec) > 3 or spec[2] for spec in rules_spec):
        import json
        out["rules_json"] = np.asarray(json.dumps([list(spec) for spec in rules_spec]))
        out["torch_seed"] = np.asarray(seed + 100)      # the random techniques draw from torch's global CPU generator
    stub = StubBuffer(seed + 7)
    SS.SAMPLE_BUFFER = stub
    sp = torch.tensor(sparse)
    # --- G3: tree pieces --------------------------------------------------------------------------
    torch.manual_seed(seed + 100)
    root, firstPred, firstProbs, maskedProbs = MSP.multiStepGenForward(setup.generator, ids, mk_rules(), "cpu",
                                                                       sparseIndex=sp)
    nodes = dump_tree(root, "gen_", out)
    out["gen_logits"], out["gen_firstProbs"] = firstPred.logits, firstProbs
    for i, nz in enumerate(stub.served):
        out[f"noise{i}"] = nz
    MSP.multiStepDiscForward(setup.discriminator, root, "cpu", dualValueHeads=True)
    out["disc_root_pred"] = root.discPreds.prediction
    out["disc_root_inV"], out["disc_root_altV"] = root.valueHeadData["InputValue"], root.valueHeadData["AlterationsValue"]
    for i, n in enumerate(nodes):
        out[f"disc_node{i}_pred"] = n.discPreds.prediction
        out[f"disc_node{i}_inV"], out[f"disc_node{i}_altV"] = n.valueHeadData["InputValue"], n.valueHeadData["AlterationsValue"]
    # --- G5: the whole step through GannSetup, fresh noise stub with the same seed ----------------
    SS.SAMPLE_BUFFER = StubBuffer(seed + 7)
    conf = TrainingConfig.SamplingConfig(mk_rules(), numSampleSequences=len(sparse), sparseIndex=sparse)
    rl = RLM.ReinforcementLearningConfig(labelMode=RLM.ReinforcementLabelMode.ALTERATIONS_AND_INPUT_UNWEIGHTED,
                                         sectionWeightA=1, sectionWeightB=1, sectionWeightC=1.5, teacherForcing=True)
    setup.zero_grad()
    torch.manual_seed(seed + 100)
    discLosses, genLosses = setup.trainForwardPassNoMetrics(ids, conf, 9999, collectTreeDepthData=True,
                                                            reinforcementLearningConfig=rl)
    genLosses["loss"].backward()
    discLosses["loss"].backward()
    for k, v in discLosses.items():
        out["dloss::" + k] = v if isinstance(v, torch.Tensor) else np.asarray(float(v))
    for k, v in genLosses.items():
        out["gloss::" + k] = v if isinstance(v, torch.Tensor) else np.asarray(float(v))
    for k, p in setup.named_parameters():
        out["grad::" + k] = p.grad if p.grad is not None else np.zeros(0, dtype=np.float32)
    save(f"g35_{tag}", **out)


def g6_schedules():
    """Update / learning-rate / checkpoint schedules: decisions of the reference's own classes on fixed
    (step, epoch) grids -> g6_schedules.json.  Training/Utils.py imports wandb (absent here): an empty stand-in module is
    registered for the import only - nothing of it is called."""
    import contextlib
    import io
    import json
    import types
    sys.modules.setdefault("wandb", types.ModuleType("wandb"))
    from Training import GanUpdateSchedule as GUS
    from Training import Optimizers as RO
    from Training import Utils as RU
    quiet = contextlib.redirect_stdout(io.StringIO())
    points = [[s, e] for e in range(3) for s in (0, 1, 2, 3, 4, 5, 6, 7, 10, 99, 100, 101, 1000)]
    upd = []
    for cls, args in (("AlwaysUpdateSchedule", []), ("NeverUpdateSchedule", []), ("OnlyUpdateDisc", []), ("OnlyUpdateGen", []),
                      ("AlternateSequence", []), ("AlternateSequence", [3]), ("AlternateOverEpochsUpdateSchedule", []),
                      ("GeneratorSlowStart", [100]), ("GeneratorSlowStart", [0])):
        sched = getattr(GUS, cls)(*args)
        upd.append(dict(cls=cls, args=args, repr=repr(sched), saves=[sched.saveGen, sched.saveDisc], points=points,
                        decisions=[list(sched.getGenDiscUpdates(s, e, None, None)) for s, e in points]))
    freeze = []
    for base_lr, kw, steps in ((0.5, dict(zero_steps=3, warmup_steps=5, lr_start=1e-5, lr_end=1e-3), 12),
                               (1e-4, dict(zero_steps=0, warmup_steps=4, lr_start=0.0, lr_end=2e-4, step_size_down=7), 8),
                               (1e-4, dict(zero_steps=2, warmup_steps=1, lr_start=5e-5, lr_end=1e-5), 6)):
        p = torch.nn.Parameter(torch.zeros(3))
        opt = torch.optim.SGD([p], lr=base_lr)
        sch = RO.FreezeLRSchedule(opt, **kw)
        lrs = [opt.param_groups[0]["lr"]]
        for _ in range(steps):
            opt.step()
            sch.step()
            lrs.append(opt.param_groups[0]["lr"])
        freeze.append(dict(base_lr=base_lr, kwargs=kw, steps=steps, lrs=lrs))
    created = []
    for args, accum, n_batches, epochs, steps in ((dict(base_lr=1e-5, max_lr=1e-3, step_size_up=6), 2, 20, 2, 24),
                                                  (dict(base_lr=1e-4, max_lr=5e-4), 1, 10, 1, 12)):
        p = torch.nn.Parameter(torch.zeros(3))
        opt = torch.optim.SGD([p], lr=0.1, momentum=0.9)
        with quiet:
            sch = RO._createScheduler(torch.optim.lr_scheduler.CyclicLR, dict(args), opt, accum, [0] * n_batches, epochs)
        lrs = []
        for _ in range(steps):
            opt.step()
            sch.step()
            lrs.append(opt.param_groups[0]["lr"])
        created.append(dict(args=args, gradAccum=accum, stepsPerEpoch=n_batches, epochs=epochs, steps=steps, lrs=lrs))
    trackers = []
    for args, kw, epochs, batches in (([0.25, 40], {}, 3, 40), ([0.3, 10], dict(maxEpoch=None, epochSaveFrequency=2), 4, 10),
                                      ([None, 10], {}, 2, 10), ([0.5, 8], dict(enabled=False), 2, 8)):
        tr = RU.ModelSaveFrequencyTracker(*args, **kw)
        events = []
        with quiet:
            for epoch in range(epochs):
                tr.initEpoch()
                for b in range(batches):
                    due, name = tr.shouldModelBeSaved(b, epoch)
                    if due:
                        events.append([epoch, b, name])
                events.append([epoch, "end", bool(tr.shouldModelBeSavedAtThisEpoch(epoch + 1))])
        trackers.append(dict(args=args, kwargs=kw, epochs=epochs, batches=batches, events=events))
    path = os.path.join(OUT, "g6_schedules.json")
    with open(path, "w") as f:
        json.dump(dict(update_schedules=upd, freeze_lr=freeze, create_scheduler=created, save_tracker=trackers), f, indent=1)
    print("wrote", path)


def g7_metrics():
    """Evaluation metrics: the reference's Training/Metrics functions on random inputs -> g7_metrics.npz.
    Inputs and every returned value are stored; V = 97 exercises rank thresholds beyond the vocabulary."""
    import types
    sys.modules.setdefault("wandb", types.ModuleType("wandb"))
    from Training import Utils as RU
    from Training.Metrics import DiscriminatorMetrics as DM
    from Training.Metrics import GeneratorMetrics as GM
    from Training.Metrics import MetricUtils as MU
    out = {}
    for tag, (B, S, V, N) in {"a": (2, 12, 1031, 3), "b": (3, 9, 97, 2), "c": (1, 6, 2500, 4)}.items():
        g = torch.Generator().manual_seed(700 + V)
        probs = torch.softmax(3 * torch.randn(B, S, V, generator=g), dim=-1)
        ids = torch.randint(0, V, (B, S), generator=g)
        head = torch.scatter(probs[:, :-1], dim=-1, index=ids[:, 1:].unsqueeze(-1), value=0)
        masked = torch.cat((head, probs[:, -1:]), dim=1)
        d_in = torch.rand(B, S, generator=g)
        d_alt = torch.rand(B, N, S, generator=g)
        heads = {'txtInputPreds': torch.rand(B, S, 1, generator=g), 'txtAlterationsPreds': torch.rand(B, S, 1, generator=g),
                 'altInputPreds': torch.rand(N, B, S, 1, generator=g), 'altAlterationsPreds': torch.rand(N, B, S, 1, generator=g)}
        adv = torch.randn(B, 1 + N, S - 1, generator=g)
        out.update({f"{tag}::in::probs": probs, f"{tag}::in::ids": ids, f"{tag}::in::masked": masked, f"{tag}::in::d_in": d_in,
                    f"{tag}::in::d_alt": d_alt, f"{tag}::in::adv": adv})
        out.update({f"{tag}::in::heads::{k}": v for k, v in heads.items()})
        ent = RU.entropy(probs)
        out[f"{tag}::entropy"] = ent
        for k, v in GM.calculateGeneratorMetrics(ids, probs, masked, d_in, d_alt).items():
            out[f"{tag}::gen::{k}"] = v
        for k, v in GM.calculateGeneratorMetricsNoDiscriminator(ids, probs, ent).items():
            out[f"{tag}::gen_nodisc::{k}"] = v
        for k, v in DM.calcDiscriminatorMetrics(d_in, d_alt, {'DiscValueHeads': heads}).items():
            out[f"{tag}::disc::{k}"] = v
        a, b = DM.calcAdvantageRatios(adv)
        out[f"{tag}::adv_ratio"] = torch.stack((a, b))
        out[f"{tag}::mle_probs"] = MU.mleLossFromProbs(ids, probs)
        logits = torch.log(probs)
        out[f"{tag}::mle_logits"] = torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, V), ids[:, 1:].reshape(-1))
        # placements where the label's probability is unique in its row (elsewhere an unstable argsort may put it anywhere in the tie)
        lab = torch.gather(probs[:, :-1], -1, ids[:, 1:].unsqueeze(-1))
        out[f"{tag}::label_unique"] = (probs[:, :-1] == lab).sum(-1) == 1
        out[f"{tag}::placements"] = GM._calculateInputIDsPlacementMetric(ids, probs)[0]
    save("g7_metrics", **out)


def g8_dataloader():
    """Dataset streaming: the batch sequence of the reference's PickleBasedDataloader (doShuffle=False) over
    a small synthetic dataset this script writes into a temporary directory (three files of 10 / 7 / 13 samples, ids that
    encode file and row), for several (batch size, world, rank, maxNumSamples) and two epochs -> g8_dataloader.json with the
    dataset itself, so the test can rebuild the files."""
    import contextlib
    import io
    import json
    import pickle
    import tempfile
    from DataManagers import DataUtils
    files = {"part_a.pkl": 10, "part_b.pkl": 7, "part_c.pkl": 13}
    data = {name: [[1000 * (fi + 1) + r, r, fi, 7, 8] for r in range(n)] for fi, (name, n) in enumerate(files.items())}
    cases = []
    with tempfile.TemporaryDirectory() as tmp:
        for name, rows in data.items():
            with open(os.path.join(tmp, name), "wb") as fp:
                pickle.dump(rows, fp)
        with open(os.path.join(tmp, "meta.json"), "w") as fp:
            json.dump(files, fp)
        order = [os.path.basename(f) for f in DataUtils.getAllMatchingDatasetFiles(tmp)[0]]
        for batch, world, rank, cap in ((4, None, None, None), (3, 2, 0, None), (3, 2, 1, None), (2, 3, 2, None), (5, None, None, 18),
                                        (4, 4, 1, None)):
            with contextlib.redirect_stdout(io.StringIO()):
                dl = DataUtils.loadPreTokenizedDataloader(batch, tmp, doShuffle=False, maxNumSamples=cap, ddpWorldSize=world, ddpRank=rank,
                                                          verbose=False)
                epochs = [[b.tolist() for b in dl] for _ in range(2)]
            cases.append(dict(batch=batch, world=world, rank=rank, maxNumSamples=cap, numBatches=len(dl), numSamples=dl.numSamples,
                              epochs=epochs))
    path = os.path.join(OUT, "g8_dataloader.json")
    with open(path, "w") as f:
        json.dump(dict(files=files, file_order=order, data=data, cases=cases), f)
    print("wrote", path)


def g9_text_metrics():
    """Generated-text metrics: Training/Metrics/GeneratedTextsMetric.py on random id sequences."""
    import json
    from Training.Metrics import GeneratedTextsMetric as GT
    rng = np.random.RandomState(9)
    pairs = [(rng.randint(0, 40, 24).tolist(), rng.randint(0, 60, 16).tolist()) for _ in range(7)]
    orig, gen = rng.rand(7).tolist(), (rng.rand(7) * 2).tolist()
    out = dict(pairs=pairs, ttr=[list(map(float, GT.TTR(c, g))) for c, g in pairs], mean_ttr=list(map(float, GT.calculateTTR(pairs))),
               original=orig, generated=gen, self_cce={k: float(v) for k, v in GT.selfCrossEntropy(orig, gen).items()})
    path = os.path.join(OUT, "g9_text_metrics.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path)


def g10_boundary():
    """Round 2: the rest of the drop-in boundary - default-mask branch attention at head sizes the HIP kernel takes (A3), the
    generator-loss pieces under the reference's own function names, the noise buffer's roll-by-split, and the techniques
    that draw from torch's global CPU generator (reproducible under a fixed seed)."""
    import types
    from BranchPredictions import SampleBuffer
    from Models.Losses import AdvantageCalculations as AC, ReinforceLoss as RL
    out = {}
    B = 2
    torch.manual_seed(1001)
    for fam in ("neox", "gpt2"):
        for Dh, H, Q, Kl in ((32, 2, 16, 40), (64, 1, 5, 9), (32, 1, 7, 7)):
            if fam == "neox":
                fn = NeoXNonResidualAttention(tiny_cfg("neox", hidden=H * Dh, heads=H))._nonResidualTextAttention
            else:
                fn = GPT2NonResidualAttention(tiny_cfg("gpt2", hidden=H * Dh, heads=H), layer_idx=0).nonResidualTextAttention
            q, k, v = [torch.randn(B, H, Q, Dh, requires_grad=True) for _ in range(3)]
            kc, vc = [torch.randn(B, H, Kl, Dh, requires_grad=True) for _ in range(2)]
            o, w = fn(q, k, v, kc, vc)
            go = torch.randn_like(o)
            grads = torch.autograd.grad(o, [q, k, v, kc, vc], go)
            for nm, t in zip(["q", "k", "v", "kctx", "vctx", "out", "weights", "gout", "dq", "dk", "dv", "dkctx", "dvctx"],
                             [q, k, v, kc, vc, o, w, go, *grads]):
                out[f"default_{fam}_{Dh}_{Q}_{Kl}_{nm}"] = t
    # --- AdvantageCalculations.dualHeadAdvantages / ReinforceLoss.* ---
    g = torch.Generator().manual_seed(1002)
    Bn, N, S = 3, 4, 11
    in_scores, alt_scores = torch.rand(Bn, S, generator=g), torch.rand(Bn, N, S, generator=g)
    in_head, alt_head = torch.rand(Bn, S, 1, generator=g), torch.rand(Bn, S, 1, generator=g)
    alt_head[0, 3] = in_head[0, 3]                       # a tie between the two heads (denominator clamp, >= comparisons)
    alt_scores[1, 2, 5] = in_head[1, 5, 0]               # a score exactly on a zone boundary
    w = types.SimpleNamespace(A=1.0, B=0.7, C=1.5)
    P = RLM.IncorrectValueHeadsPolicy
    for tag, pi, pa in (("ii", P.IGNORE_SAMPLE, P.IGNORE_SAMPLE), ("ni", P.DO_NOTHING, P.IGNORE_SAMPLE), ("nn", P.DO_NOTHING, P.DO_NOTHING)):
        ia, aa, (zA, zB, zC, corr, incorr) = AC.dualHeadAdvantages(in_scores, alt_scores, in_head, alt_head, w, pi, pa)
        for nm, t in zip(["in_adv", "alt_adv", "zoneA", "zoneB", "zoneC", "adv_correct", "input_correct"], [ia, aa, zA, zB, zC, corr, incorr]):
            out[f"adv_{tag}_{nm}"] = t
    for nm, t in zip(["in_scores", "alt_scores", "in_head", "alt_head"], [in_scores, alt_scores, in_head, alt_head]):
        out["adv_" + nm] = t
    out["adv_weights"] = np.asarray([w.A, w.B, w.C], dtype=np.float32)
    probs = torch.rand(Bn, N, S, generator=g)
    probs[0, 0, :4] = torch.tensor([0.0, 1.0, 1e-7, 1 - 1e-8])      # outside the clamp: zero gradient
    probs.requires_grad_(True)
    adv = torch.randn(Bn, N, S, generator=g)
    maskC = (torch.rand(Bn, N, S, generator=g) > 0.5).float()
    gl = torch.randn(Bn, N, S, generator=g)
    for tag, fn in (("mirrored", lambda: RL.mirroredNegativeReinforceLoss(probs, adv, maskC)),
                    ("plain", lambda: RL.reinforceLoss(probs, adv, False, False)),
                    ("minprob", lambda: RL.reinforceLoss(probs, adv, False, True, negPushProbLimit=0.3))):
        loss = fn()
        out[f"rl_{tag}_loss"] = loss
        out[f"rl_{tag}_dprobs"] = torch.autograd.grad(loss, probs, gl)[0]
    for nm, t in zip(["probs", "adv", "maskC", "gloss"], [probs, adv, maskC, gl]):
        out["rl_" + nm] = t
    # --- SampleBuffer.SamplingBuffer: one noise tensor per shape, rolled by pre-drawn split points ---
    np.random.seed(5)
    torch.manual_seed(5)
    buf = SampleBuffer.SamplingBuffer('cpu', numShuffleSplits=4)
    shape = (2, 5, 13)
    rolled = [buf.rand(shape).clone() for _ in range(6)]          # wraps around after 4 calls
    noise, splits, _ = buf.shapeRandBuffers[shape]
    out["buf_noise"], out["buf_splits"], out["buf_rolled"] = noise, np.asarray(splits, dtype=np.int64), torch.stack(rolled)
    # --- techniques on torch's global CPU generator ---
    probs2 = torch.softmax(3 * torch.randn(2, 6, 97, generator=g), -1)
    ids2 = torch.randint(0, 97, (2, 6), generator=g)
    out["rand_probs"], out["rand_ids"] = probs2, ids2
    torch.manual_seed(77)
    out["rand_RANDOM"] = SS.sampleFromGeneratorDistribution(probs2, 5, ids2, T.RANDOM)[0]
    torch.manual_seed(78)
    out["rand_EPSILON_GREEDY"] = SS.sampleFromGeneratorDistribution(probs2, 6, ids2, T.EPSILON_GREEDY, numRandomSamples=2)[0]
    torch.manual_seed(79)
    out["rand_EPSILON_GREEDY_final"] = SS.sampleFromGeneratorDistribution(probs2, 6, ids2, T.EPSILON_GREEDY, blockSamplingOfLabels=False,
                                                                          includeFinalTimeStep=True, numRandomSamples=4)[0]
    save("g10_boundary", **out)


def main():
    torch.set_num_threads(1)  # fixed reduction order for reproducible fixtures
    if sys.argv[1:] == ["g10"]:
        return g10_boundary()
    if sys.argv[1:] == ["g9"]:
        return g9_text_metrics()
    if sys.argv[1:] == ["g8"]:
        return g8_dataloader()
    if sys.argv[1:] == ["g6"]:
        return g6_schedules()
    if sys.argv[1:] == ["g7"]:
        return g7_metrics()
    TK, NR, EG = T.TOP_K.value, T.SAMPLE_NO_REPLACEMENT.value, T.EPSILON_GREEDY.value
    # round 3: the tree options - random children (EPSILON_GREEDY), a special rule for beam 0, random leaves that stop
    rand_spec = [(2, TK, {}),
                 (2, EG, {"numRandomSamples": 1}, {"firstBeam": (EG, {"numRandomSamples": 2})}),
                 (1, TK, {}, {"continueFromRandom": False})]
    if sys.argv[1:] == ["g35rand"]:
        return g3_g5("neox", "neox_rand", rand_spec, 15, [0, 3, 5, 8, 10])
    g6_schedules()
    g7_metrics()
    g8_dataloader()
    g9_text_metrics()
    g1_attention()
    g2_sampling()
    g4_losses()
    g10_boundary()
    g3_g5("neox", "neox_rand", rand_spec, 15, [0, 3, 5, 8, 10])
    g3_g5("neox", "neox_chain4", [(1, TK, {})] * 4, 11, [1, 3, 6, 8, 10])
    g3_g5("neox", "neox_tree211", [(2, TK, {}), (1, NR, {}), (1, TK, {})], 12, [0, 2, 5, 9])
    g3_g5("gpt2", "gpt2_chain3", [(1, TK, {})] * 3, 13, [1, 3, 6, 8, 10])
    g3_g5("gpt2", "gpt2_tree21", [(2, NR, {}), (1, TK, {})], 14, [2, 4, 7, 10])


if __name__ == "__main__":
    main()
