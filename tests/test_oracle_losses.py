"""Pins oracle/losses.py against tests/golden/g4_losses.npz (captured from the reference's
calcDiscriminatorLoss / calcGeneratorLoss for every label mode x value-head policy x teacher forcing)."""
import numpy as np
import torch

from oracle import gan as G
from oracle import losses as L

TREES = ["chain3", "tree2x1"]


def _run_case(g, row):
    case, ti, mode, policy, tf = row
    tn = TREES[ti]
    t = {k: torch.from_numpy(g[f"{tn}_{k}"]) for k in ["inputScores", "altScoresK", "txtIn", "txtAlt", "altIn", "altAlt",
                                                       "altProbsK", "logits", "ids", "depthMask"]}
    S = t["inputScores"].shape[1]
    leaf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and k != "depthMask" else v) for k, v in t.items()}
    alt_scores = G.pad_last(leaf["altScoresK"], S)
    heads = {"txtInputPreds": leaf["txtIn"], "txtAlterationsPreds": leaf["txtAlt"],
             "altAlterationsPreds": G.pad_last(leaf["altAlt"], S, dim=2)}
    links_arr = g[f"{tn}_links"].tolist()
    links = (g[f"{tn}_rootLinks"].tolist(), [[] if l < 0 else [l] for l in links_arr])
    d = L.discriminator_loss(leaf["inputScores"], alt_scores, 9999, t["depthMask"], heads, links)
    gl = L.generator_loss(leaf["ids"], leaf["logits"], torch.softmax(leaf["logits"], -1), leaf["inputScores"].detach(),
                          alt_scores.detach(), G.pad_last(leaf["altProbsK"], S, 1e-5), heads, t["depthMask"],
                          label_mode=mode, in_policy=policy, alt_policy=policy, wA=1, wB=1, wC=1.5, teacher_forcing=bool(tf))
    return leaf, d, gl


def test_losses_and_grads_match_reference(golden):
    g = golden("g4_losses")
    assert len(g["index"]) == 32
    for row in g["index"].tolist():
        case = row[0]
        p = f"case{case}_"
        leaf, d, gl = _run_case(g, row)
        assert np.array_equal(d["loss-Input"].detach().numpy(), g[p + "disc_inputLoss"])
        assert np.array_equal(d["loss-Alterations"].detach().numpy(), g[p + "disc_altLoss"])
        for k in ["ValueHeadLoss", "ValueHead-InputLoss", "ValueHead-AltLoss"]:
            assert np.array_equal(d[k].detach().numpy(), g[p + "disc_" + k]), (case, k)
        assert d["numAlterationSamples"] == int(g[p + "disc_numAlterationSamples"])
        gen_keys = [k[len(p) + 4:] for k in g.files if k.startswith(p + "gen_")]
        assert len(gen_keys) == 27 and set(gen_keys) == set(gl)      # every key of the reference's generator loss dict
        for k in gen_keys:
            if k.startswith("loss"):
                assert np.array_equal(gl[k].detach().numpy(), g[p + "gen_" + k]), (case, k)
            else:                                                     # _calculateDebuggingStats (GeneratorLosses.py:34-81)
                np.testing.assert_allclose(gl[k].numpy(), g[p + "gen_" + k], rtol=2e-6, atol=1e-7, err_msg=f"{case} {k}")
        total = (d["loss-Input"] + d["loss-Alterations"]) / 2 + d["ValueHeadLoss"] * 0.25
        gd = torch.autograd.grad(total, [leaf["inputScores"], leaf["altScoresK"], leaf["txtIn"], leaf["txtAlt"], leaf["altAlt"]],
                                 allow_unused=True)
        for nm, gr in zip(["dInputScores", "dAltScoresK", "dTxtIn", "dTxtAlt", "dAltAlt"], gd):
            want = g[p + nm]
            if want.size == 0:
                assert gr is None
            else:
                np.testing.assert_allclose(gr.numpy(), want, rtol=0, atol=1e-7, err_msg=nm)
        gg = torch.autograd.grad(gl["loss"], [leaf["logits"], leaf["altProbsK"]], allow_unused=True)
        for nm, gr in zip(["dLogits", "dAltProbsK"], gg):
            if g[p + nm].size == 0:
                assert gr is None
            else:
                np.testing.assert_allclose(gr.numpy(), g[p + nm], rtol=0, atol=1e-7, err_msg=nm)


def test_combine_disc_losses():
    d = {"loss-Input": torch.tensor(0.6), "loss-Alterations": torch.tensor(0.2), "ValueHeadLoss": torch.tensor(0.4),
         "numAlterationSamples": 3}
    a = L.combine_disc_losses(d, True, 1.0, 1.0, 0.25)
    assert abs(float(a["predLoss"]) - 0.4) < 1e-7 and abs(float(a["loss"]) - 0.5) < 1e-7
    b = L.combine_disc_losses(d, False, 1.0, 1.0, 0.25)
    assert abs(float(b["predLoss"]) - (0.6 + 0.6) / 4) < 1e-7
