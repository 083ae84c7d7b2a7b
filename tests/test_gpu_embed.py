"""bg_embed_fwd / bg_embed_bwd against the ATen ops the reference's model runs (`wte(input_ids) + wpe(position_ids)`,
Models/GPT2BranchAttention/NonResidualDiscriminatorBase.py GPT2Model.forward) and their autograd.

Forward: bit-exact (one fp32 add per element, like the reference).  Backward: sums of fp32 rows in a different (fixed) order
than ATen's - compared with a float64 scatter-add at rtol 1e-5 / atol 1e-5 * max|dy|, and bit-identical between two runs."""
import pytest
import torch

from branchgan import _capi as C
from branchgan import fused

pytestmark = pytest.mark.gpu


def _zipf(n, vocab, gen):
    ranks = torch.arange(1, vocab + 1, dtype=torch.float64)
    return torch.multinomial(1.0 / ranks, n, replacement=True, generator=gen)


def _bwd(dy, idx, table_rows, accumulate_into=None):
    rows, C_ = dy.shape
    dW = accumulate_into if accumulate_into is not None else torch.zeros(table_rows, C_, device="cuda")
    nbytes = C.lib().bg_embed_bwd_workspace_bytes(rows)
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device="cuda")
    C.check(C.lib().bg_embed_bwd(C.ptr(dy), C.ptr(idx), rows, idx.numel(), table_rows, C_, C.ptr(dW), C.ptr(ws), ws.numel(),
                                 C.stream_ptr()), "bg_embed_bwd")
    return dW


@pytest.mark.parametrize("rows,vocab,C_", [(1, 5, 4), (7, 3, 8), (300, 50, 768), (3072, 50257, 768), (3072, 2, 1028),
                                           (8192 + 5000, 11, 256), (20000, 50257, 1600)])
def test_embed_bwd_matches_scatter_add(rows, vocab, C_):
    gen = torch.Generator().manual_seed(rows + C_)
    idx = _zipf(rows, vocab, gen).cuda()
    dy = torch.randn(rows, C_, generator=gen).cuda()
    base = torch.randn(vocab, C_, generator=gen).cuda()
    got = _bwd(dy, idx, vocab, accumulate_into=base.clone())
    ref = base.double().index_add_(0, idx, dy.double())
    torch.testing.assert_close(got.double(), ref, rtol=1e-5, atol=1e-5 * float(dy.abs().max()))
    again = _bwd(dy, idx, vocab, accumulate_into=base.clone())
    assert torch.equal(got, again)          # no atomics: the same bits every time
    # ATen's own backward for the same op, at its fp32 accuracy
    w = base.clone().requires_grad_(True)
    torch.nn.functional.embedding(idx, w).backward(dy)
    torch.testing.assert_close(got - base, w.grad, rtol=1e-4, atol=1e-4 * float(dy.abs().max()))


def test_embed_bwd_broadcast_positions_and_bad_indices():
    gen = torch.Generator().manual_seed(3)
    B, T, C_ = 6, 37, 64
    pos = torch.arange(5, 5 + T).cuda()
    dy = torch.randn(B * T, C_, generator=gen).cuda()
    got = _bwd(dy, pos, 64)
    ref = torch.zeros(64, C_, device="cuda", dtype=torch.float64).index_add_(0, pos.repeat(B), dy.double())
    torch.testing.assert_close(got.double(), ref, rtol=1e-5, atol=1e-5)
    # indices outside the table are skipped (and never alias a valid row, whatever their low 32 bits are)
    idx = torch.tensor([3, -1, 70, 3, (1 << 32) + 3, 2], device="cuda")
    dy = torch.randn(6, C_, generator=gen).cuda()
    got = _bwd(dy, idx, 64)
    ref = torch.zeros(64, C_, device="cuda")
    ref[3] = dy[0] + dy[3]
    ref[2] = dy[5]
    torch.testing.assert_close(got, ref, rtol=1e-6, atol=1e-6)
    assert C.lib().bg_embed_bwd(C.ptr(dy), C.ptr(idx), 6, 6, 64, 6, C.ptr(got), C.ptr(got), 64, None) != 0     # C % 4


@pytest.mark.parametrize("B,T,C_,shared_pos", [(1, 1, 4, True), (4, 33, 768, True), (3, 50, 1028, False), (8, 384, 768, True)])
def test_embedding_function_matches_modules(B, T, C_, shared_pos):
    torch.manual_seed(B * T)
    vocab, n_pos = 211, 512
    wte, wpe = torch.nn.Embedding(vocab, C_).cuda(), torch.nn.Embedding(n_pos, C_).cuda()
    ids = torch.randint(0, vocab, (B, T), device="cuda")
    if shared_pos:
        pos = torch.arange(3, 3 + T, dtype=torch.int32, device="cuda").unsqueeze(0).expand(B, T)
    else:
        pos = torch.randint(0, n_pos, (B, T), device="cuda", dtype=torch.int32)
    ref = wte(ids) + wpe(pos.long())
    got = fused.embedding(ids, pos, wte.weight, wpe.weight)
    assert torch.equal(got, ref)
    dy = torch.randn_like(ref)
    ref.backward(dy)
    ref_g = (wte.weight.grad.clone(), wpe.weight.grad.clone())
    wte.weight.grad, wpe.weight.grad = None, None
    got.backward(dy)
    for g, r in zip((wte.weight.grad, wpe.weight.grad), ref_g):
        torch.testing.assert_close(g, r, rtol=1e-4, atol=1e-5)
    got2 = fused.embedding(ids, pos, wte.weight, wpe.weight)       # a second backward ACCUMULATES into .grad
    got2.backward(dy)
    torch.testing.assert_close(wte.weight.grad, 2 * ref_g[0], rtol=1e-4, atol=2e-5)


def test_embed_fwd_poisons_rows_with_bad_indices():
    wte, wpe = torch.randn(10, 8, device="cuda"), torch.randn(4, 8, device="cuda")
    ids = torch.tensor([1, 10, 2, -1, 3], device="cuda")
    pos = torch.tensor([0, 1, 4, 2, 3], device="cuda")
    out = torch.empty(5, 8, device="cuda")
    C.check(C.lib().bg_embed_fwd(C.ptr(ids), 5, C.ptr(pos), 5, C.ptr(wte), 10, C.ptr(wpe), 4, 8, C.ptr(out), C.stream_ptr()), "fwd")
    assert torch.equal(out[0], wte[1] + wpe[0]) and torch.equal(out[4], wte[3] + wpe[3])
    assert out[1:4].isnan().all()
    C.check(C.lib().bg_embed_fwd(C.ptr(ids[:1]), 1, None, 0, C.ptr(wte), 10, None, 0, 8, C.ptr(out), C.stream_ptr()), "fwd")   # token table only
    assert torch.equal(out[0], wte[1])
