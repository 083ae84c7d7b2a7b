"""Pins oracle/attention.py and the mask builders against tests/golden/g1_attention.npz (captured from the
reference's _nonResidualTextAttention / nonResidualTextAttention with real PredictionTreeNode masks)."""
import numpy as np
import torch

from oracle import attention as A
from oracle import tree as T


def _cases(g):
    for ci, fam, H, Dh, K, S, depth, inv, layer in g["meta"].tolist():
        yield ci, (A.NEOX if fam == 0 else A.GPT2), H, Dh, K, S, depth, bool(inv), layer


def test_branch_attention_forward_and_grads_match_reference(golden):
    g = golden("g1_attention")
    for ci, fam, H, Dh, K, S, depth, inv, layer in _cases(g):
        for d in range(1, depth + 1):
            p = f"c{ci}_d{d}_"
            t = {k: torch.from_numpy(g[p + k]) for k in ["q", "k", "v", "kctx", "vctx", "mask", "gout"]}
            leaves = [t[k].clone().requires_grad_(True) for k in ["q", "k", "v", "kctx", "vctx"]]
            out, w = A.branch_attention(fam, *leaves, mask=t["mask"], layer_idx=layer, scale_by_inverse_layer_idx=inv)
            assert np.array_equal(out.detach().numpy(), g[p + "out"]), p
            assert np.array_equal(w.detach().numpy(), g[p + "weights"]), p
            grads = torch.autograd.grad(out, leaves, t["gout"])
            for nm, gr in zip(["dq", "dk", "dv", "dkctx", "dvctx"], grads):
                np.testing.assert_allclose(gr.numpy(), g[p + nm], rtol=0, atol=1e-6, err_msg=p + nm)


def test_masks_match_reference_tree_masks(golden):
    g = golden("g1_attention")
    for ci, fam, H, Dh, K, S, depth, inv, layer in _cases(g):
        sparse = torch.from_numpy(g[f"c{ci}_sparse"])
        m = T.depth1_mask(2, sparse, S)
        for d in range(1, depth + 1):
            want = g[f"c{ci}_d{d}_mask"]
            assert np.array_equal(m.numpy(), want)
            # arithmetic descriptor form == materialised mask (what the HIP kernel evaluates)
            desc = A.tree_mask_from_descriptor(sparse + 1, d - 1, S, K)
            assert np.array_equal(desc.expand(2, 1, K, -1).numpy(), want)
            # rows sum to 1, zero outside the mask
            w = g[f"c{ci}_d{d}_weights"]
            np.testing.assert_allclose(w.sum(-1), 1.0, atol=1e-5)
            assert (w[~np.broadcast_to(want, w.shape)] == 0).all()
            m = T.deeper_mask(m, 2, K)


def test_default_mask_path(golden):
    g = golden("g1_attention")
    for fam in (A.NEOX, A.GPT2):
        t = {k: torch.from_numpy(g[f"default_{fam}_{k}"]) for k in ["q", "k", "v", "kctx", "vctx"]}
        out, w = A.branch_attention(fam, t["q"], t["k"], t["v"], t["kctx"], t["vctx"], mask=None)
        assert np.array_equal(out.numpy(), g[f"default_{fam}_out"])
        assert np.array_equal(w.numpy(), g[f"default_{fam}_weights"])


def test_spliced_sequence_identity():
    """branch attention for (k, depth d) == causal attention over text[0..s_k] ++ ancestors ++ self."""
    torch.manual_seed(0)
    B, H, Dh, S, K, d = 1, 2, 16, 12, 3, 3
    sparse = torch.tensor([2, 5, 9])
    q, ks, vs = [torch.randn(B, H, K, Dh) for _ in range(3)]
    kt, vt = torch.randn(B, H, S, Dh), torch.randn(B, H, S, Dh)
    ka, va = torch.randn(B, H, (d - 1) * K, Dh), torch.randn(B, H, (d - 1) * K, Dh)
    mask = A.tree_mask_from_descriptor(sparse + 1, d - 1, S, K)
    out, _ = A.branch_attention(A.NEOX, q, ks, vs, torch.cat((kt, ka), 2), torch.cat((vt, va), 2), mask)
    for i, s in enumerate(sparse.tolist()):
        anc = [j * K + i for j in range(d - 1)]
        kk = torch.cat((kt[:, :, :s + 1], ka[:, :, anc], ks[:, :, i:i + 1]), 2)
        vv = torch.cat((vt[:, :, :s + 1], va[:, :, anc], vs[:, :, i:i + 1]), 2)
        o, _ = A.causal_attention(A.NEOX, q[:, :, i:i + 1], kk, vv)
        np.testing.assert_allclose(o[:, :, 0].numpy(), out[:, :, i].numpy(), atol=1e-5)
