"""BASELINE configs[4] at its OWN size: all 8192 x 50304 rows of the branch-sampling sweep (the inputs bench.py --workload c5
times: probs = softmax(3 N(0,1)), noise = -log(U + 1e-14), one blocked label per row) through bg_topn_select /
bg_topp_select, against the oracle's restatement of the reference's argsort path (BranchPredictions/SamplingStrategies.py:
149-205) on the host, chunk by chunk.  Bit-exact: the number of rows whose ids differ must be 0.

At this size the kernels take paths the smaller parity cases do not: 8192 rows run topn_partial_kernel with ONE split per
row (no merge kernel), and top-p runs 8192 workgroups of the list-form kernel with its leftover list (rows the list form
hands to the round-1 form) behind them."""
import numpy as np
import pytest
import torch

from oracle import sampling as S

pytestmark = pytest.mark.gpu

ROWS, V, CHUNK = 8192, 50304, 512


@pytest.fixture(scope="module")
def sweep():
    import bench
    dev = torch.device("cuda", 0)
    _, probs, noise, blocked = bench.sampling_cases(ROWS, V, dev)
    return probs, noise, blocked


def _chunks(*tensors):
    for lo in range(0, ROWS, CHUNK):
        yield lo, [t[lo:lo + CHUNK].cpu() for t in tensors]


def test_top_k_and_no_replacement_every_row(sweep):
    from branchgan import ops
    probs, noise, blocked = sweep
    got = {("k", n): ops.topn_select(probs, n, blocked=blocked).cpu() for n in (1, 4, 16)}
    got.update({("e", n): ops.topn_select(probs, n, noise=noise, blocked=blocked).cpu() for n in (1, 4)})
    torch.set_num_threads(16)
    bad = {k: 0 for k in got}
    for lo, (p, e, b) in _chunks(probs, noise, blocked):
        masked = p.scatter(1, b, 0.0)                                   # SamplingStrategies.py:258-263
        order = {"k": torch.argsort(masked, dim=-1, descending=True, stable=True)[:, :16],
                 "e": torch.argsort(masked / e, dim=-1, descending=True, stable=True)[:, :4]}
        for (kind, n), ids in got.items():
            bad[(kind, n)] += int((ids[lo:lo + CHUNK] != order[kind][:, :n]).any(1).sum())
    assert all(v == 0 for v in bad.values()), f"rows that differ from the argsort path, of {ROWS}: {bad}"


@pytest.mark.parametrize("p", [0.5, 0.9])
def test_top_p_every_row(sweep, p):
    from branchgan import ops
    probs, noise, blocked = sweep
    got = ops.topp_select(probs, 4, noise, p, blocked=blocked).cpu()
    torch.set_num_threads(16)
    bad = 0
    for lo, (pr, e, b) in _chunks(probs, noise, blocked):
        masked = pr.scatter(1, b, 0.0)
        filt = S.top_p_filter(masked.unsqueeze(0), p, specified=True)
        want = S.sample_without_replacement(filt, 4, e.unsqueeze(0), specified=True)[0].T
        bad += int((got[lo:lo + CHUNK] != want).any(1).sum())
    assert bad == 0, f"{bad} of {ROWS} rows differ from the reference's sort / cumsum / argsort path at p = {p}"
