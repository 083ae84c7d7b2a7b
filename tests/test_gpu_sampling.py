"""GPU parity: bg_topn_select (through the C-ABI) vs the oracle and the reference's golden ids.  Bit-exact."""
import numpy as np
import pytest
import torch

from oracle import sampling as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from branchgan import ops as _ops
    return _ops


def _run_hip(ops, probs, n, ids, tech, block, final, noise, params):
    """The product's row-level path for one (B, T, V) probs tensor, mirroring SamplingStrategies semantics."""
    from branchgan import sampling
    alt, _ = sampling.sampleFromGeneratorDistribution(probs.cuda(), n, ids.cuda(), tech, blockSamplingOfLabels=block,
                                                      includeFinalTimeStep=final, noise=None if noise is None else noise.cuda(),
                                                      **params)
    return alt.cpu()


def test_golden_ids_all_techniques(golden, ops):
    g = golden("g2_sampling")
    checked = 0
    for row in g["index"]:
        case, si, tech, block, final, n = [int(x) for x in row[:6]]
        top_p, inv_k = float(row[6]), int(row[7])
        if tech == S.TOP_P_SAMPLE_NO_REPLACEMENT:
            continue  # covered by test_gpu_topp.py
        probs, ids = torch.from_numpy(g[f"s{si}_probs"]), torch.from_numpy(g[f"s{si}_ids"])
        params = {"inverseTopK": inv_k} if inv_k >= 0 else {}
        noise = torch.from_numpy(g[f"s{si}_f{final}_noise"]) if tech != S.TOP_K else None
        alt = _run_hip(ops, probs, n, ids, tech, bool(block), bool(final), noise, params)
        assert np.array_equal(alt.numpy(), g[f"case{case}_alt"]), (case, tech, block, final, n)
        checked += 1
    assert checked > 60


@pytest.mark.parametrize("rows,V,n", [(8, 50304, 1), (8, 50304, 16), (3, 50257, 4), (300, 1031, 64), (2, 97, 5), (4096, 4099, 2)])
@pytest.mark.parametrize("with_noise", [False, True])
def test_topn_vs_oracle_random(ops, rows, V, n, with_noise):
    gen = torch.Generator().manual_seed(rows * 7 + V + n)
    logits = 3.0 * torch.randn(rows, V, generator=gen)
    probs = torch.softmax(logits, -1)
    noise = S.exponential_noise((rows, V), gen) if with_noise else None
    blocked = torch.randint(0, V, (rows, 1), generator=gen)
    masked = probs.clone()
    masked.scatter_(1, blocked, 0.0)
    keys = masked / noise if with_noise else masked
    want = S.top_n(keys.unsqueeze(0), n, specified=True)[0].T          # (rows, n)
    ids, got_keys = ops.topn_select(probs.cuda(), n, noise=None if noise is None else noise.cuda(), blocked=blocked.cuda(),
                                    blocked_value=0.0, return_keys=True)
    assert np.array_equal(ids.cpu().numpy(), want.numpy())
    assert np.array_equal(got_keys.cpu().numpy(), torch.gather(keys, 1, want).numpy())   # IEEE division, same bits


def test_ties_nan_and_negative_zero(ops):
    V = 5000
    x = torch.zeros(4, V)
    x[0, [7, 3000, 4999]] = 1.0                      # ties: index order
    x[1, :] = float("nan")                           # all NaN: index order
    x[2, 10] = float("nan"); x[2, 11] = float("inf")  # NaN sorts above +inf
    x[3, 5] = -0.0; x[3, 6] = -1.0                   # -0.0 == +0.0
    ids = ops.topn_select(x.cuda(), 4).cpu()
    assert ids[0].tolist() == [7, 3000, 4999, 0]
    assert ids[1].tolist() == [0, 1, 2, 3]
    assert ids[2].tolist() == [10, 11, 0, 1]
    assert ids[3].tolist() == [0, 1, 2, 3]
    want = S.top_n(x.unsqueeze(0), 4, specified=True)[0].T
    assert np.array_equal(ids.numpy(), want.numpy())


def test_row_map_and_bf16_and_strided(ops):
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(64, 3001, generator=gen)
    rm = torch.tensor([5, 5, 63, 0, 17])
    ids = ops.topn_select(x.cuda(), 3, row_map=rm.cuda()).cpu()
    want = S.top_n(x[rm].unsqueeze(0), 3, specified=True)[0].T
    assert np.array_equal(ids.numpy(), want.numpy())
    xb = x.bfloat16()
    ids = ops.topn_select(xb.cuda(), 8).cpu()
    want = S.top_n(xb.float().unsqueeze(0), 8, specified=True)[0].T
    assert np.array_equal(ids.numpy(), want.numpy())
    wide = torch.randn(16, 4000, generator=gen)
    view = wide[:, 3:3 + 2999]                      # misaligned rows, row stride != V
    ids = ops.topn_select(view.cuda(), 5).cpu() if False else ops.topn_select(wide.cuda()[:, 3:3 + 2999], 5).cpu()
    want = S.top_n(view.unsqueeze(0), 5, specified=True)[0].T
    assert np.array_equal(ids.numpy(), want.numpy())


def test_argument_errors(ops):
    x = torch.randn(2, 100).cuda()
    with pytest.raises(RuntimeError):
        ops.topn_select(x, 65)
    with pytest.raises(RuntimeError):
        ops.topn_select(x[:, :10], 11)
    with pytest.raises(RuntimeError):
        ops.topn_select(torch.randn(2, 100), 1)      # CPU tensor: no fallback
