"""The BENCHMARKED precision against the oracle: the bf16-autocast HIP train step (what bench.py times) vs the fp32 CPU
restatement of Models/GanSetup.py:193-258 (oracle.gan.train_forward) on the same weights, tokens and branch points, at the
BASELINE configs' own model sizes, S, K and D (c1, c2) and at c4's SHAPES (Pythia-1B / 410M dims, Dh 256 and 64, S 512, K 64,
D 8) with two layers per model.

bf16 rounds every GEMM / attention operand to 8 significant bits, so the two runs cannot sample identical branch tokens
everywhere: a TOP_K pick flips wherever the two best logits of a row are closer than the rounding noise.  The comparison is
therefore TEACHER-FORCED: the oracle continues its tree from the tokens the HIP step sampled (oracle.tree._substitute), which
makes every node, loss and gradient comparable, and every pick the fp32 model would have made differently is listed with the
logit margin the bf16 run had to overcome.  The stated tolerance:

  * sampled ids: a flip is legitimate only if the fp32 logit margin between the two candidates is <= FLIP_MARGIN (in units
    of the row's logit standard deviation); anything larger is a bug.  The flip rate is reported and bounded.
  * every loss of both dicts: |got - want| <= LOSS_RTOL * |want| + LOSS_ATOL.
  * every parameter gradient tensor: ||got - want||_F <= GRAD_FROB * ||want||_F  (+ an absolute floor for all-but-zero ones),
    and the two models' whole gradients (all tensors as one vector) within GRAD_FROB_TOTAL.
"""
import numpy as np
import pytest
import torch

from test_gpu_configs import _build, _oracle_cfg

pytestmark = pytest.mark.gpu

FLIP_MARGIN = 0.06       # x the row's logit std: the largest fp32 margin a bf16 top-1 flip may overcome
FLIP_RATE = 0.08         # of all sampled branch tokens
LOSS_RTOL, LOSS_ATOL = 1.5e-2, 2e-3
GRAD_FROB, GRAD_FROB_TOTAL = 6e-2, 3e-2


def _c4_shapes():
    from branchgan.configs import LMConfig
    return dict(gen=LMConfig("gpt_neox", 50304, 2048, 2, 8, 2048), disc=LMConfig("gpt_neox", 50304, 1024, 2, 16, 2048),
                S=512, K=64, D=8, B=1)


CASES = {
    "c1": dict(gen="pythia-70m", disc="pythia-70m", S=128, K=16, D=2, B=4),
    "c2": dict(gen="gpt2-small", disc="gpt2-small", S=256, K=32, D=4, B=2),
    "c4-shapes": _c4_shapes,
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_bf16_step_against_the_fp32_oracle(name):
    from oracle import gan as OG
    wl = CASES[name]() if callable(CASES[name]) else CASES[name]
    model, conf, rl, ids, sparse = _build(wl)
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    model = model.cuda()
    conf.sparseIndex = conf.sparseIndex.cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        state = model._generatorForward(ids.cuda(), conf)                    # root pass, branch steps, sampling
        got_ids = [n.ids.cpu() for n in state[0].getSubNodes()]
        d, g = model.trainForwardPassNoMetrics(ids.cuda(), conf, 9999, collectTreeDepthData=False, reinforcementLearningConfig=rl,
                                               generatorState=state)
        g["loss"].backward()
        d["loss"].backward()
    torch.cuda.synchronize()
    assert len(got_ids) == wl["D"]

    torch.set_num_threads(16)
    d_ref, g_ref, extra = OG.train_forward(sd, _oracle_cfg(wl["gen"]), _oracle_cfg(wl["disc"]), ids, [(1, 2, {})] * wl["D"], sparse,
                                           rl=dict(label_mode=3, wA=1, wB=1, wC=1.5, teacher_forcing=True),
                                           forced_ids=[t.unsqueeze(1) for t in got_ids])
    g_ref["loss"].backward()
    d_ref["loss"].backward()

    # 1. sampled ids: which picks would the fp32 model have made differently, and by what margin
    sigma = float(extra["logits"].detach().std(-1).mean())
    flips = total = 0
    worst = 0.0
    for step, t in enumerate(extra["sampling_trace"]):
        differ = t["own"] != t["forced"]
        flips += int(differ.sum())
        total += differ.numel()
        assert bool((t["gap"] >= -1e-6).all()), f"{name} step {step}: the fp32 pick is not the fp32 arg-max"
        if bool(differ.any()):
            worst = max(worst, float(t["gap"][differ].max()) / sigma)
    print(f"{name}: {flips} of {total} bf16 picks differ from the fp32 model's; largest fp32 margin overcome {worst:.4f} sigma "
          f"(sigma = {sigma:.4f} logit units)")
    assert worst <= FLIP_MARGIN, f"{name}: a pick differs although the fp32 logits are {worst:.3f} sigma apart"
    assert flips <= FLIP_RATE * total

    # 2. losses
    rel = {}
    for tag, got, want, keys in (("disc", d, d_ref, ("loss", "predLoss", "loss-Input", "loss-Alterations", "ValueHeadLoss")),
                                 ("gen", g, g_ref, ("loss", "loss-MLE", "loss-RL", "loss-RL-input", "loss-RL-alterations"))):
        for k in keys:
            a, b = float(got[k].detach()), float(want[k].detach())
            rel[f"{tag} {k}"] = abs(a - b) / (LOSS_RTOL * abs(b) + LOSS_ATOL)
    worst_loss = max(rel, key=rel.get)
    print(f"{name}: worst loss {worst_loss}: {rel[worst_loss]:.3f} of the tolerance")
    assert rel[worst_loss] <= 1.0, rel

    # 3. gradients: relative Frobenius error per tensor and per model
    per, tot = {}, {"generator": [0.0, 0.0], "discriminator": [0.0, 0.0]}
    for k, p in model.named_parameters():
        want = sd[k].grad
        if want is None or not bool(want.any()):
            assert p.grad is None or not bool(p.grad.any()), k
            continue
        got = p.grad.float().cpu()
        assert bool(torch.isfinite(got).all()), k
        err, ref = float((got - want).double().norm()), float(want.double().norm())
        per[k] = err / (ref + 1e-7)
        m = tot[k.split(".")[0]]
        m[0] += err ** 2
        m[1] += ref ** 2
    worst_grad = max(per, key=per.get)
    totals = {m: (e / r) ** 0.5 for m, (e, r) in tot.items()}
    print(f"{name}: {len(per)} gradient tensors, worst relative Frobenius error {per[worst_grad]:.4f} ({worst_grad}); "
          f"whole models {totals}")
    assert len(per) > 40
    assert per[worst_grad] <= GRAD_FROB, (worst_grad, per[worst_grad])
    assert max(totals.values()) <= GRAD_FROB_TOTAL, totals
