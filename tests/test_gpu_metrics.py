"""Evaluation metrics on the GPU: bg_rank_stats against plain torch sorts of the same rows, and
branchgan.metrics against the values the reference's own Training/Metrics functions returned for the same inputs
(tests/golden/g7_metrics.npz, written by tests/golden/make_goldens.py).

Integer outputs (placements, top-k ids) are exact - except among exactly tied probabilities, where the reference's unstable
argsort may return any order (the golden marks rows whose label probability is unique).  Sums of probabilities: fp32, a
different summation order than torch's, 2e-6 absolute on values <= 1; means / stds of those: 1e-5 relative."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _reference(x, labels, ranks, eps):
    xs, order = torch.sort(x.double(), dim=-1, descending=True, stable=True)
    out = {"min": x.min(-1).values, "max": x.max(-1).values, "entropy": -(x.double() * torch.log(x.double() + eps)).sum(-1)}
    V = x.shape[-1]
    out["top_mass"] = torch.stack([xs[..., :min(r, V)].sum(-1) for r in ranks], dim=-1)
    ok = (labels >= 0) & (labels < V)
    safe = labels.clamp(0, V - 1)
    out["placement"] = torch.where(ok, (order == safe.unsqueeze(-1)).int().argmax(-1), torch.full_like(labels, -1))
    out["label_value"] = torch.where(ok, torch.gather(x, -1, safe.unsqueeze(-1)).squeeze(-1), torch.full_like(x[..., 0], float("nan")))
    return out


@pytest.mark.parametrize("rows,V", [(7, 97), (5, 1031), (3, 50257), (4, 50304), (2, 4096)])
@pytest.mark.parametrize("kind", ["softmax", "ties", "flat", "signed"])
def test_rank_stats_matches_sort(rows, V, kind):
    from branchgan import ops
    g = torch.Generator().manual_seed(rows * 131 + V)
    if kind == "softmax":
        x = torch.softmax(3 * torch.randn(rows, V, generator=g), -1)
    elif kind == "ties":                                   # 16 distinct values: every threshold sits inside a run of equal values
        x = torch.randint(0, 16, (rows, V), generator=g).float() / 64
    elif kind == "flat":                                   # two values: at V = 50k far more candidates than LDS holds (general path)
        x = (1 + torch.randint(0, 2, (rows, V), generator=g).float()) / 64
    else:                                                  # negative values, zeros of both signs, an infinity
        x = torch.randn(rows, V, generator=g)
        x[:, ::7] = 0.0
        x[:, 3::11] = -0.0
        x[0, 5] = float("inf")
    x = x.cuda()
    labels = torch.randint(0, V, (rows,), generator=g).cuda()
    labels[-1] = -1
    ranks = (1, 8, 16, 64, 128, 256, 512, 1024)
    eps = 1e-8 if kind != "signed" else 10.0               # log of a negative number is not the kernel's business
    if kind == "signed":
        x = x.clamp(min=-5)
    got = ops.rank_stats(x, labels, ranks, entropy_eps=eps)
    want = _reference(x, labels, ranks, eps)
    assert torch.equal(got["placement"], want["placement"])
    assert torch.equal(got["min"], want["min"]) and torch.equal(got["max"], want["max"])
    assert torch.equal(got["label_value"][:-1], want["label_value"][:-1]) and bool(torch.isnan(got["label_value"][-1]))
    if kind == "signed":
        finite = torch.isfinite(want["top_mass"])
        assert torch.equal(torch.isfinite(got["top_mass"]), finite)
        torch.testing.assert_close(got["top_mass"][finite].double(), want["top_mass"][finite], rtol=1e-5, atol=1e-3)
    else:
        torch.testing.assert_close(got["top_mass"].double(), want["top_mass"], rtol=0, atol=2e-6 * (64 if kind in ("ties", "flat") else 1) * max(1, V // 4096))
        torch.testing.assert_close(got["entropy"].double(), want["entropy"], rtol=1e-5, atol=1e-5)


def test_rank_stats_strided_rows_and_argument_checks():
    from branchgan import _capi as C
    from branchgan import ops
    big = torch.softmax(torch.randn(6, 1000, generator=torch.Generator().manual_seed(1)), -1).cuda()
    view = big[:, 3:900]                                   # rows start 12 bytes off a 16-byte boundary, stride 1000
    got = ops.rank_stats(view, None, (5,))
    want = torch.sort(view.double(), -1, descending=True).values[:, :5].sum(-1)
    torch.testing.assert_close(got["top_mass"][:, 0].double(), want, rtol=0, atol=2e-6)
    assert "placement" not in got
    with pytest.raises(ValueError):
        ops.rank_stats(big, None, tuple(range(1, 10)))
    with pytest.raises(TypeError):
        ops.rank_stats(big.bfloat16(), None, (1,))
    assert C.lib().bg_rank_stats(C.ptr(big), 6, 1000, 999, None, None, 0, 1e-8, None, None, None, None, None, None, C.stream_ptr()) != 0
    assert b"stride" in C.lib().bg_last_error()


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_metrics_match_reference_goldens(tag):
    from branchgan import metrics, sampling
    g = np.load(os.path.join(GOLDEN, "g7_metrics.npz"))
    t = lambda k: torch.from_numpy(g[f"{tag}::{k}"]).cuda()
    probs, ids, masked, d_in, d_alt = t("in::probs"), t("in::ids"), t("in::masked"), t("in::d_in"), t("in::d_alt")
    heads = {k: t("in::heads::" + k) for k in ("txtInputPreds", "txtAlterationsPreds", "altInputPreds", "altAlterationsPreds")}

    def check(prefix, got):
        want_keys = {k[len(f"{tag}::{prefix}::"):] for k in g.files if k.startswith(f"{tag}::{prefix}::")}
        assert set(got) == want_keys
        for k in sorted(want_keys):
            w, v = t(f"{prefix}::{k}"), got[k]
            if k.startswith("Top-"):
                assert torch.equal(v, w), k                                    # no ties among the 20 largest of these rows
            elif k == "Input-PredPlacement-Median" or k.startswith("Input-PredPlacement-RankRate"):
                assert torch.equal(v.float(), w.float()), k                      # integer-valued statistics: exact
            else:
                torch.testing.assert_close(v.float(), w.float().view(v.shape), rtol=1e-5, atol=2e-6, msg=lambda m: f"{k}: {m}")

    assert bool(t("label_unique").all())                                       # so placements are comparable one for one
    st = metrics._row_stats(ids, probs)
    assert torch.equal(st["placement"][:, :-1].float(), t("placements"))
    check("gen", metrics.calculateGeneratorMetrics(ids, probs, masked, d_in, d_alt))
    lazy = sampling.MaskedProbs(probs, ids, True, True)                          # what the forward pass hands over: same ids
    assert torch.equal(metrics.calculateGeneratorMetrics(ids, probs, lazy, d_in, d_alt)["Top-20"], t("gen::Top-20"))
    check("gen_nodisc", metrics.calculateGeneratorMetricsNoDiscriminator(ids, probs))
    torch.testing.assert_close(metrics.entropy(probs), t("entropy"), rtol=1e-5, atol=1e-5)
    check("disc", metrics.calcDiscriminatorMetrics(d_in, d_alt, {"DiscValueHeads": heads}))
    a, b = metrics.calcAdvantageRatios(t("in::adv"))
    torch.testing.assert_close(torch.stack((a, b)), t("adv_ratio"))
    torch.testing.assert_close(metrics.mleLossFromProbs(ids, probs), t("mle_probs"), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(metrics.mleLossFromLogits(ids, torch.log(probs)), t("mle_logits"), rtol=1e-5, atol=1e-5)
