"""GPU parity of the fused loss kernels (vs goldens captured from the reference's calcDiscriminatorLoss / calcGeneratorLoss)
and of the NeoX QKV-split + rotary kernel (vs the oracle).  fp32, |err| <= 2e-6 on losses, 1e-6 on gradients."""
import numpy as np
import pytest
import torch

from oracle import attention as A

pytestmark = pytest.mark.gpu
TREES = ["chain3", "tree2x1"]


def test_fused_losses_match_reference_goldens(golden):
    from branchgan import configs, losses
    g = golden("g4_losses")
    for case, ti, mode, policy, tf in g["index"].tolist():
        tn, p = TREES[ti], f"case{case}_"
        t = {k: torch.from_numpy(g[f"{tn}_{k}"]).cuda() for k in ["inputScores", "altScoresK", "txtIn", "txtAlt", "altIn", "altAlt",
                                                                  "altProbsK", "logits", "ids", "depthMask"]}
        leaf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and k != "depthMask" else v) for k, v in t.items()}
        K = leaf["altScoresK"].shape[-1]
        links = (g[f"{tn}_rootLinks"].tolist(), [[] if l < 0 else [l] for l in g[f"{tn}_links"].tolist()])
        fpd = {"DiscValueHeads": {"txtInputPreds": leaf["txtIn"], "txtAlterationsPreds": leaf["txtAlt"], "altInputPreds": leaf["altIn"],
                                  "altAlterationsPreds": leaf["altAlt"]},
               "SubNodeStructure": links, "TreeNodeInfo": [(int(d), False) for d in g[f"{tn}_depths"]],
               "AlterationProbs": leaf["altProbsK"]}
        inL, altL, info = losses.calcDiscriminatorLoss(None, leaf["inputScores"], leaf["altScoresK"], 9999, t["depthMask"], None,
                                                       collectTreeDepthData=True, forwardPassData=fpd)
        f = lambda x: x.detach().cpu().numpy()
        np.testing.assert_allclose(f(inL), g[p + "disc_inputLoss"], atol=2e-6)
        np.testing.assert_allclose(f(altL), g[p + "disc_altLoss"], atol=2e-6)
        for k in ["ValueHeadLoss", "ValueHead-InputLoss", "ValueHead-AltLoss", "Depth-1-Loss", "Depth-2-Loss", "Depth-1-Acc", "Depth-2-Pred"]:
            np.testing.assert_allclose(f(info[k]), g[p + "disc_" + k], atol=2e-6, err_msg=k)
        total = (inL + altL) / 2 + info["ValueHeadLoss"] * 0.25
        gd = torch.autograd.grad(total, [leaf["inputScores"], leaf["altScoresK"], leaf["txtIn"], leaf["txtAlt"], leaf["altAlt"]])
        for nm, gr in zip(["dInputScores", "dAltScoresK", "dTxtIn", "dTxtAlt", "dAltAlt"], gd):
            np.testing.assert_allclose(f(gr), g[p + nm], atol=1e-6, err_msg=nm)

        rl = configs.ReinforcementLearningConfig(labelMode=configs.ReinforcementLabelMode(mode),
                                                 inputIncorrectValueHeadsPolicy=configs.IncorrectValueHeadsPolicy(policy),
                                                 alterationIncorrectValueHeadsPolicy=configs.IncorrectValueHeadsPolicy(policy),
                                                 sectionWeightA=1, sectionWeightB=1, sectionWeightC=1.5, teacherForcing=bool(tf))
        setup = type("S", (), {"generatorMleLossWeight": 1.0, "generatorRlLossWeight": 1.0})()
        gl = losses.calcGeneratorLoss(setup, leaf["ids"], leaf["logits"], None, leaf["inputScores"].detach(), leaf["altScoresK"].detach(),
                                      fpd, rl, depthMask=t["depthMask"])
        gen_keys = [k[len(p) + 4:] for k in g.files if k.startswith(p + "gen_")]
        assert len(gen_keys) == 27 and list(gl) == gen_keys          # the reference's dict, key for key and in its order
        for k in gen_keys:                                            # incl. _calculateDebuggingStats (GeneratorLosses.py:34-81)
            np.testing.assert_allclose(f(gl[k]), g[p + "gen_" + k], atol=3e-6, rtol=2e-6, err_msg=f"{case} {k}")
        gg = torch.autograd.grad(gl["loss"], [leaf["logits"], leaf["altProbsK"]], allow_unused=True)
        np.testing.assert_allclose(f(gg[0]), g[p + "dLogits"], atol=1e-6, err_msg="dLogits")
        if g[p + "dAltProbsK"].size:
            np.testing.assert_allclose(f(gg[1]), g[p + "dAltProbsK"], atol=1e-5, rtol=1e-5, err_msg="dAltProbsK")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_neox_qkv_rope_forward_backward(dtype):
    from branchgan import ops
    from branchgan.models import rope_tables
    gen = torch.Generator().manual_seed(0)
    B, T, H, Dh, rot, P = 2, 9, 3, 32, 8, 64
    qkv = torch.randn(B, T, H * 3 * Dh, generator=gen).to(dtype)
    pos = torch.randint(0, P, (B, T), generator=gen)
    cos, sin = A.rotary_tables(rot, P)
    x = qkv.float().clone().requires_grad_(True)
    v4 = x.view(B, T, H, 3 * Dh)
    q_ref = A.apply_rotary(v4[..., :Dh].permute(0, 2, 1, 3), cos, sin, pos, rot)
    k_ref = A.apply_rotary(v4[..., Dh:2 * Dh].permute(0, 2, 1, 3), cos, sin, pos, rot)
    v_ref = v4[..., 2 * Dh:].permute(0, 2, 1, 3)
    gq, gk, gv = [torch.randn(B, H, T, Dh, generator=gen) for _ in range(3)]
    (q_ref * gq + k_ref * gk + v_ref * gv).sum().backward()
    xd = qkv.cuda().requires_grad_(True)
    c, s = rope_tables(rot, P, 10000.0, xd.device)
    q, k, v = ops.neox_qkv_rope(xd, c, s, pos.cuda().int(), H, Dh, rot)
    (q.float() * gq.cuda() + k.float() * gk.cuda() + v.float() * gv.cuda()).sum().backward()
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    f = lambda t: t.detach().float().cpu().numpy()
    np.testing.assert_allclose(f(q), q_ref.detach().numpy(), atol=tol)
    np.testing.assert_allclose(f(k), k_ref.detach().numpy(), atol=tol)
    np.testing.assert_allclose(f(v), v_ref.detach().numpy(), atol=tol)
    np.testing.assert_allclose(f(xd.grad), x.grad.numpy(), atol=tol * 4)
