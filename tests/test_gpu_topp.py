"""GPU parity: bg_topp_select vs the reference's golden ids and vs the oracle's top_p_filter on random / tied rows.
Ids must be identical; the only non-bit-exact ingredient is torch.sum's fp32 summation order of the kept mass (documented in
DESIGN.md), which can move a key by 1 ulp — the tests count mismatching rows and require 0."""
import numpy as np
import pytest
import torch

from oracle import sampling as S

pytestmark = pytest.mark.gpu


def test_golden_top_p_ids(golden):
    from branchgan import sampling
    g = golden("g2_sampling")
    checked = n_golden = 0
    for row in g["index"]:
        case, si, tech, block, final, n = [int(x) for x in row[:6]]
        if tech != S.TOP_P_SAMPLE_NO_REPLACEMENT:
            continue
        probs, ids = torch.from_numpy(g[f"s{si}_probs"]), torch.from_numpy(g[f"s{si}_ids"])
        noise = torch.from_numpy(g[f"s{si}_f{final}_noise"])
        alt, _ = sampling.sampleFromGeneratorDistribution(probs.cuda(), n, ids.cuda(), sampling.SamplingTechniques(tech),
                                                          blockSamplingOfLabels=bool(block), includeFinalTimeStep=bool(final),
                                                          noise=noise.cuda(), topP=float(row[6]))
        got, want = alt.cpu().numpy(), g[f"case{case}_alt"]
        # Rows that keep fewer than n tokens (or nothing: NaN rows) are filled from tied keys, whose order in the reference is an
        # introsort artefact (oracle.sampling.top_n): those rows are compared with the specified tie order, all others with
        # the reference's golden ids.
        masked = S.block_labels(probs, ids, bool(block), bool(final))
        filt = S.top_p_filter(masked, float(row[6]))
        spec = S.sample_without_replacement(filt, n, noise, specified=True).numpy()           # (B, n, T')
        assert np.array_equal(got[:, :, 1:], spec), (case, n)
        enough = ((filt > 0).sum(-1) >= n).unsqueeze(1).expand(-1, n, -1).numpy()
        assert np.array_equal(got[:, :, 1:][enough], want[:, :, 1:][enough]), (case, n)
        n_golden += int(enough.sum())
        checked += 1
    assert checked >= 20 and n_golden > 50


# V <= 53245: the row-in-registers kernel; beyond: the streaming kernel (60013 is odd: misaligned rows)
@pytest.mark.parametrize("rows,V,n,p", [(16, 50304, 4, 0.5), (16, 50304, 4, 0.9), (64, 1031, 16, 0.9), (8, 97, 3, 0.3), (4, 50257, 1, 0.99),
                                        (6, 53245, 4, 0.7), (6, 60013, 4, 0.7), (3, 131072, 2, 0.9)])
def test_vs_oracle_random(rows, V, n, p):
    from branchgan import ops
    gen = torch.Generator().manual_seed(V + n)
    probs = torch.softmax(1.5 * torch.randn(rows, V, generator=gen), -1)
    noise = S.exponential_noise((rows, V), gen)
    filt = S.top_p_filter(probs.unsqueeze(0), p)
    want = S.sample_without_replacement(filt, n, noise.unsqueeze(0), specified=True)[0].T
    ids = ops.topp_select(probs.cuda(), n, noise.cuda(), p).cpu()
    bad = (ids != want).any(1).sum().item()
    assert bad == 0, f"{bad} of {rows} rows differ"


def test_ties_at_the_cut_and_degenerate_rows():
    from branchgan import ops
    V, n = 4096, 4
    probs = torch.zeros(5, V)
    probs[0, :] = 1.0 / V                                   # uniform: every value ties, the cut falls inside the tie group
    probs[1, 5] = 0.6; probs[1, 6:] = 0.4 / (V - 6)         # top-1 mass > p: keeps nothing -> NaN row -> ids 0..n-1
    probs[2, 100:110] = 0.05; probs[2, 110:] = 0.5 / (V - 110)   # 10-way tie right at the cut (p = 0.5 -> 0.5 fits exactly)
    probs[3, 7] = 0.3; probs[3, 9] = 0.3; probs[3, 11] = 0.3; probs[3, 13] = 0.1
    probs[4, :] = torch.softmax(torch.arange(V).float() * 1e-3, 0)
    gen = torch.Generator().manual_seed(1)
    noise = S.exponential_noise((5, V), gen)
    for p in (0.5, 0.9, 1.5):
        filt = S.top_p_filter(probs.unsqueeze(0), p, specified=True)      # ties at the cut: index order (the contract)
        want = S.sample_without_replacement(filt, n, noise.unsqueeze(0), specified=True)[0].T
        got = ops.topp_select(probs.cuda(), n, noise.cuda(), p).cpu()
        assert np.array_equal(got.numpy(), want.numpy()), p
