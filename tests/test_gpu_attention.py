"""GPU parity: bg_branch_attn_fwd/bwd (through the C-ABI) vs goldens captured from the reference's
_nonResidualTextAttention / nonResidualTextAttention, vs the oracle at other shapes, and properties.

Tolerances (stated, per BASELINE north_star):  fp32 inputs -> |err| <= 2e-5 (fp32 MFMA = exact fma chain; the
differences are summation order and the hardware exp);  bf16 inputs -> |err| <= 3e-2 on O(1) values, checked
against the oracle evaluated on the SAME bf16-rounded inputs in fp32."""
import math

import numpy as np
import pytest
import torch

from oracle import attention as A

pytestmark = pytest.mark.gpu


def _split_ctx(kctx, S, K, d):
    text = kctx[:, :, :S]
    anc = [kctx[:, :, S + j * K: S + (j + 1) * K] for j in range(d - 1)]
    return text, anc


def _scale(fam, Dh, inv, layer):
    s = 1.0 / math.sqrt(Dh)
    return s / (layer + 1) if (fam == A.GPT2 and inv) else s


def _run(ops, q, k, v, kctx, vctx, sparse, S, K, d, scale, dtype=torch.float32):
    dev = "cuda"
    cast = lambda t: t.to(dtype).to(dev).requires_grad_(True)
    kt, ka = _split_ctx(kctx, S, K, d)
    vt, va = _split_ctx(vctx, S, K, d)
    leaves = dict(q=cast(q), kt=cast(kt.contiguous()), vt=cast(vt.contiguous()),
                  ks=[cast(t.contiguous()) for t in ka] + [cast(k)], vs=[cast(t.contiguous()) for t in va] + [cast(v)])
    ctx_len = (torch.as_tensor(sparse) + 1).to(torch.int32).to(dev)
    out, lse = ops.branch_attention(leaves["q"], leaves["kt"], leaves["vt"], ctx_len, leaves["ks"], leaves["vs"], scale=scale)
    return out, lse, leaves


def test_golden_forward_backward_fp32(golden):
    from branchgan import ops
    g = golden("g1_attention")
    for ci, fam, H, Dh, K, S, depth, inv, layer in g["meta"].tolist():
        fam = A.NEOX if fam == 0 else A.GPT2
        if Dh < 32:
            continue  # kernels take Dh in {32,64,128,256}; Dh=8 cases are covered by the oracle tests only
        sparse = g[f"c{ci}_sparse"]
        for d in range(1, depth + 1):
            p = f"c{ci}_d{d}_"
            t = {k: torch.from_numpy(g[p + k]) for k in ["q", "k", "v", "kctx", "vctx", "gout"]}
            out, lse, leaves = _run(ops, t["q"], t["k"], t["v"], t["kctx"], t["vctx"], sparse, S, K, d, _scale(fam, Dh, inv, layer))
            np.testing.assert_allclose(out.detach().cpu().numpy(), g[p + "out"], atol=2e-5, rtol=0, err_msg=p)
            out.backward(t["gout"].cuda())
            np.testing.assert_allclose(leaves["q"].grad.cpu().numpy(), g[p + "dq"], atol=2e-5, err_msg=p + "dq")
            np.testing.assert_allclose(leaves["ks"][-1].grad.cpu().numpy(), g[p + "dk"], atol=2e-5, err_msg=p + "dk")
            np.testing.assert_allclose(leaves["vs"][-1].grad.cpu().numpy(), g[p + "dv"], atol=2e-5, err_msg=p + "dv")
            dkctx = torch.cat([leaves["kt"].grad] + [x.grad for x in leaves["ks"][:-1]], dim=2).cpu().numpy()
            dvctx = torch.cat([leaves["vt"].grad] + [x.grad for x in leaves["vs"][:-1]], dim=2).cpu().numpy()
            np.testing.assert_allclose(dkctx, g[p + "dkctx"], atol=2e-5, err_msg=p + "dkctx")
            np.testing.assert_allclose(dvctx, g[p + "dvctx"], atol=2e-5, err_msg=p + "dvctx")
            # lse vs the golden weights: weight of the self column = exp(score_self - lse)
            w_self = g[p + "weights"][..., -1]
            sc = (t["q"] * t["k"]).sum(-1).numpy() * _scale(fam, Dh, inv, layer)
            np.testing.assert_allclose(np.exp(sc - lse.cpu().numpy()), w_self, atol=2e-5)


CASES = [  # B, H, Dh, S, K, depth
    (2, 3, 64, 256, 32, 4), (1, 2, 128, 130, 40, 2), (2, 2, 32, 70, 33, 3), (1, 1, 256, 96, 64, 2), (1, 12, 64, 512, 64, 8),
    (2, 2, 256, 300, 48, 3),        # Dh 256 over ten key tiles with a ragged last one: in bf16 the key-split forward (round 2)
    (1, 4, 256, 512, 64, 4),        # c4's branch step: 8 workgroups x 16 key tiles - the key range split over 4 workgroups (round 3)
    (1, 2, 64, 64, 8, 11),          # a path of 11 entries: three per wave - the two preloaded with the prologue and one behind them
]


@pytest.mark.parametrize("B,H,Dh,S,K,depth", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_vs_oracle(B, H, Dh, S, K, depth, dtype):
    from branchgan import ops
    gen = torch.Generator().manual_seed(B * 1000 + Dh + S)
    sparse = np.sort(np.random.RandomState(S).choice(S - 1, K, replace=False))
    r = lambda *s: torch.randn(*s, generator=gen).to(dtype).float()      # oracle sees the same (rounded) values
    q, k, v = r(B, H, K, Dh), r(B, H, K, Dh), r(B, H, K, Dh)
    C = S + (depth - 1) * K
    kctx, vctx, gout = r(B, H, C, Dh), r(B, H, C, Dh), r(B, H, K, Dh)
    mask = A.tree_mask_from_descriptor(torch.as_tensor(sparse) + 1, depth - 1, S, K)
    lv = [t.clone().requires_grad_(True) for t in (q, k, v, kctx, vctx)]
    o_ref, w_ref = A.branch_attention(A.NEOX, *lv, mask=mask)
    g_ref = torch.autograd.grad(o_ref, lv, gout)
    out, lse, leaves = _run(ops, q, k, v, kctx, vctx, sparse, S, K, depth, 1.0 / math.sqrt(Dh), dtype)
    out.backward(gout.to(dtype).cuda())
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    f = lambda t: t.detach().float().cpu().numpy()
    np.testing.assert_allclose(f(out), o_ref.detach().numpy(), atol=tol)
    np.testing.assert_allclose(f(leaves["q"].grad), g_ref[0].numpy(), atol=tol)
    np.testing.assert_allclose(f(leaves["ks"][-1].grad), g_ref[1].numpy(), atol=tol)
    np.testing.assert_allclose(f(leaves["vs"][-1].grad), g_ref[2].numpy(), atol=tol)
    dkctx = torch.cat([leaves["kt"].grad] + [x.grad for x in leaves["ks"][:-1]], dim=2)
    dvctx = torch.cat([leaves["vt"].grad] + [x.grad for x in leaves["vs"][:-1]], dim=2)
    np.testing.assert_allclose(f(dkctx), g_ref[3].numpy(), atol=tol * 4)   # text rows sum up to K query rows
    np.testing.assert_allclose(f(dvctx), g_ref[4].numpy(), atol=tol * 4)
    assert out.dtype == dtype and leaves["kt"].grad.dtype == dtype


@pytest.mark.parametrize("K", [7, 40])
def test_multi_node_tree_one_pass_equals_node_by_node(K):
    """One launch over all nodes of a branching tree (paths differ per node) == per-node launches; gradients of shared
    slots accumulate over the nodes that see them.  K = 7: 35 query rows per head (key-split forward kernel);
    K = 40: 200 rows (query-split kernel), while the per-node launches it is compared with take the other one."""
    from branchgan import ops
    gen = torch.Generator().manual_seed(3)
    B, H, Dh, S = 2, 2, 64, 50 if K == 7 else 130
    sparse = torch.tensor([0, 3, 10, 11, 30, 40, 48]) if K == 7 else torch.from_numpy(np.sort(np.random.RandomState(5).choice(S - 1, K, replace=False)))
    ctx_len = (sparse + 1).int().cuda()
    paths = [[0], [0, 1], [2], [2, 3], [2, 4]]           # nodes: a, a1, b, b1, b2
    N = len(paths)
    mk = lambda *s: torch.randn(*s, generator=gen).cuda().requires_grad_(True)
    q = mk(B, H, N, K, Dh)
    kt, vt = mk(B, H, S, Dh), mk(B, H, S, Dh)
    ks, vs = [mk(B, H, K, Dh) for _ in range(N)], [mk(B, H, K, Dh) for _ in range(N)]
    gout = torch.randn(B, H, N, K, Dh, generator=gen).cuda()
    out, _ = ops.branch_attention(q, kt, vt, ctx_len, ks, vs, paths=paths)
    all_leaves = [q, kt, vt] + ks + vs
    g_one = torch.autograd.grad(out, all_leaves, gout)
    outs = []
    for n, p in enumerate(paths):
        o, _ = ops.branch_attention(q[:, :, n], kt, vt, ctx_len, [ks[i] for i in p], [vs[i] for i in p])
        outs.append(o)
    out2 = torch.stack(outs, dim=2)
    g_two = torch.autograd.grad(out2, all_leaves, gout)
    np.testing.assert_allclose(out.detach().cpu().numpy(), out2.detach().cpu().numpy(), atol=1e-6)
    for a, b in zip(g_one, g_two):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=1e-5)
    # and against the oracle for node b2 (path [2, 4]): ancestors = slot 2, self = slot 4
    cpu = lambda t: t.detach().cpu()
    mask = A.tree_mask_from_descriptor(sparse + 1, 1, S, K)
    o_ref, _ = A.branch_attention(A.NEOX, cpu(q[:, :, 4]), cpu(ks[4]), cpu(vs[4]), torch.cat((cpu(kt), cpu(ks[2])), 2),
                                  torch.cat((cpu(vt), cpu(vs[2])), 2), mask=mask)
    np.testing.assert_allclose(cpu(out[:, :, 4]).numpy(), o_ref.numpy(), atol=2e-5)


@pytest.mark.parametrize("fam", [A.NEOX, A.GPT2])
def test_root_causal_attention_matches_oracle(fam):
    from branchgan import ops
    gen = torch.Generator().manual_seed(9)
    B, H, S, Dh = 2, 3, 200, 64
    q, k, v, go = [torch.randn(B, H, S, Dh, generator=gen) for _ in range(4)]
    lv = [t.clone().requires_grad_(True) for t in (q, k, v)]
    o_ref, _ = A.causal_attention(fam, *lv)
    g_ref = torch.autograd.grad(o_ref, lv, go)
    ld = [t.cuda().requires_grad_(True) for t in (q, k, v)]
    o, _ = ops.causal_attention(*ld)
    g = torch.autograd.grad(o, ld, go.cuda())
    np.testing.assert_allclose(o.detach().cpu().numpy(), o_ref.detach().numpy(), atol=2e-5)
    for a, b in zip(g, g_ref):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), atol=1e-4)


def test_strided_views_of_fused_qkv():
    """q/k/v passed as views of one (B, T, 3, H, Dh) projection output: no copies, same numbers."""
    from branchgan import ops
    gen = torch.Generator().manual_seed(11)
    B, H, K, Dh, S = 2, 4, 16, 64, 40
    qkv = torch.randn(B, K, 3, H, Dh, generator=gen).cuda()
    q, k, v = [qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3)]       # (B,H,K,Dh) views, last dim contiguous
    kt, vt = torch.randn(B, H, S, Dh, generator=gen).cuda(), torch.randn(B, H, S, Dh, generator=gen).cuda()
    ctx = torch.arange(3, 3 + 2 * K, 2).int().cuda()
    o1, _ = ops.branch_attention(q, kt, vt, ctx, [k], [v])
    o2, _ = ops.branch_attention(q.contiguous(), kt, vt, ctx, [k.contiguous()], [v.contiguous()])
    assert torch.equal(o1, o2)


def test_shape_errors():
    from branchgan import ops
    q = torch.randn(1, 1, 4, 48).cuda()                   # Dh = 48 unsupported
    kt = torch.randn(1, 1, 8, 48).cuda()
    with pytest.raises(RuntimeError):
        ops.branch_attention(q, kt, kt, torch.ones(4, dtype=torch.int32).cuda(), [q], [q])
    q = torch.randn(1, 1, 4, 64).cuda()
    with pytest.raises(ValueError):
        ops.branch_attention(q, torch.randn(1, 1, 8, 64).cuda(), torch.randn(1, 1, 8, 64).cuda(),
                             torch.ones(5, dtype=torch.int32).cuda(), [q], [q])


@pytest.mark.parametrize("dtype,Dh", [(torch.bfloat16, 64), (torch.float32, 64), (torch.bfloat16, 256), (torch.float32, 256)])
def test_c_abi_root_pass_with_null_path_tables(dtype, Dh):
    """include/branchgan.h lets a C-ABI caller pass NULL path / path_len when max_path == 0 (the root causal pass): every kernel
    family behind bg_branch_attn_fwd / _bwd (key-split, query-split, one-launch and split backward, the round-1 kernels at fp32
    Dh 256) must then leave the tables alone.  Same bits as the call with tables."""
    from branchgan import ops
    gen = torch.Generator().manual_seed(21)
    B, H, S = 1, 2, 96
    q, k, v, go = [torch.randn(B, H, S, Dh, generator=gen).to(dtype).cuda() for _ in range(4)]
    ld = [t.clone().requires_grad_(True) for t in (q, k, v)]
    o_ref, lse_ref = ops.causal_attention(*ld)
    g_ref = torch.autograd.grad(o_ref, ld, go)
    q5 = q.unsqueeze(2)
    ctx_len = ops.causal_ctx_len(S, S, q.device)
    out = torch.empty((B, 1, S, H, Dh), dtype=dtype, device="cuda").permute(0, 3, 1, 2, 4)
    lse = torch.empty((B, H, 1, S), dtype=torch.float32, device="cuda")
    sh = ops._attn_shape(q5, k, [], 1.0 / Dh ** 0.5, 0, out)
    ops.attn_fwd_launch(sh, q5, k, v, [], [], ctx_len, None, None, out, lse)
    assert torch.equal(out.squeeze(2), o_ref) and torch.equal(lse.squeeze(2), lse_ref)
    dout = torch.empty_strided(out.shape, out.stride(), dtype=dtype, device="cuda").copy_(go.unsqueeze(2))
    dq = torch.empty((B, H, 1, S, Dh), dtype=dtype, device="cuda")
    dk, dv, delta = torch.empty_like(k), torch.empty_like(k), torch.empty_like(lse)
    sh.dq_sb, sh.dq_sh, sh.dq_sn, sh.dq_sk = dq.stride()[:4]
    ops.attn_bwd_launch(sh, q5, k, v, [], [], ctx_len, None, None, out, lse, dout, dq, dk, dv, None, None, [], [], delta)
    torch.cuda.synchronize()
    for got, want in zip((dq.squeeze(2), dk, dv), g_ref):
        assert torch.equal(got, want)


@pytest.mark.parametrize("dtype,Dh,B,H", [(torch.bfloat16, 256, 4, 8), (torch.float32, 64, 1, 12), (torch.bfloat16, 64, 2, 3)])
def test_key_range_split_forward_equals_unsplit(dtype, Dh, B, H):
    """bg_branch_attn_fwd_ws with a workspace deals the key tiles of a query tile to several workgroups and merges their
    (max, sum, output) states in a second kernel; without one it is bg_branch_attn_fwd.  Same attention, regrouped sums:
    outputs and lse agree to summation order, at c4's branch-step shape (B 4, H 8, Dh 256, K 64, S 512, 3 ancestor slots)."""
    import ctypes
    import os
    from branchgan import _capi as C, ops
    if os.environ.get("BRANCHGAN_ATTN_FWD_KVSPLIT") == "0":
        pytest.skip("the key-range split is switched off (A/B run)")
    gen = torch.Generator().manual_seed(31)
    K, S, n_anc = 64, 512, 3
    mk = lambda *s: torch.randn(*s, generator=gen).to(dtype).cuda()
    q = mk(B, H, 1, K, Dh)
    kt, vt = mk(B, H, S, Dh), mk(B, H, S, Dh)
    ks, vs = [mk(B, H, K, Dh) for _ in range(n_anc + 1)], [mk(B, H, K, Dh) for _ in range(n_anc + 1)]
    sparse = np.sort(np.random.RandomState(7).choice(S - 1, K, replace=False))
    ctx_len = (torch.as_tensor(sparse) + 1).int().cuda()
    path, path_len, max_path = ops.path_tables([list(range(n_anc + 1))], q.device)
    results = []
    for use_ws in (False, True):
        out = torch.empty((B, 1, K, H, Dh), dtype=dtype, device="cuda").permute(0, 3, 1, 2, 4)
        lse = torch.empty((B, H, 1, K), dtype=torch.float32, device="cuda")
        sh = ops._attn_shape(q, kt, ks, 1.0 / math.sqrt(Dh), max_path, out)
        need = C.lib().bg_branch_attn_fwd_workspace_bytes(ctypes.byref(sh))
        assert need > 0, "this shape is meant to take the key-range split"
        ws = torch.empty(need, dtype=torch.uint8, device="cuda") if use_ws else None
        C.check(C.lib().bg_branch_attn_fwd_ws(ctypes.byref(sh), C.ptr(q), C.ptr(kt), C.ptr(vt), ops._ptr_array(ks), ops._ptr_array(vs),
                                              C.ptr(ctx_len), C.ptr(path), C.ptr(path_len), C.ptr(out), C.ptr(lse), C.ptr(ws), need if use_ws else 0,
                                              C.stream_ptr()), "bg_branch_attn_fwd_ws")
        torch.cuda.synchronize()
        results.append((out.float().cpu(), lse.cpu()))
    (o0, l0), (o1, l1) = results
    tol = 2e-6 if dtype == torch.float32 else 1.6e-2          # bf16: one rounding of the output, at most one ulp apart
    np.testing.assert_allclose(o1.numpy(), o0.numpy(), atol=tol)
    np.testing.assert_allclose(l1.numpy(), l0.numpy(), atol=2e-6, rtol=1e-6)


@pytest.mark.parametrize("S,Dh,B,H", [(256, 64, 8, 12), (200, 64, 1, 3), (512, 32, 1, 4), (96, 64, 8, 12), (200, 32, 8, 8)])
def test_bf16_root_causal_backward_head_resident(S, Dh, B, H):
    """bf16 root causal pass at short sequences against the oracle on the same bf16-rounded inputs.  With at least 64 heads the
    backward is the head-resident kernel (attn_bwd_head.hip: the whole (batch, head) in one CU's LDS; configs[1]'s 96 heads at
    S 256, 96 at S 96, 64 at S 200 / Dh 32 with ragged last tiles), below that the one-launch kernel; BRANCHGAN_ATTN_BWD_HEAD=2
    sends every case that fits (S 512 at Dh 32 fills the LDS) through the head-resident one."""
    from branchgan import ops
    gen = torch.Generator().manual_seed(S + Dh)
    q, k, v, go = [torch.randn(B, H, S, Dh, generator=gen).bfloat16().float() for _ in range(4)]
    lv = [t.clone().requires_grad_(True) for t in (q, k, v)]
    o_ref, _ = A.causal_attention(A.NEOX, *lv)
    g_ref = torch.autograd.grad(o_ref, lv, go)
    ld = [t.bfloat16().cuda().requires_grad_(True) for t in (q, k, v)]
    o, _ = ops.causal_attention(*ld)
    g = torch.autograd.grad(o, ld, go.bfloat16().cuda())
    np.testing.assert_allclose(o.detach().float().cpu().numpy(), o_ref.detach().numpy(), atol=3e-2)
    for a, b_, name in zip(g, g_ref, "qkv"):
        np.testing.assert_allclose(a.float().cpu().numpy(), b_.numpy(), atol=6e-2, err_msg="d" + name)   # up to S rows sum into a dK / dV row
        rel = float((a.float().cpu() - b_).norm() / b_.norm())
        assert rel < 1e-2, (name, rel)


@pytest.mark.parametrize("K", [7, 40])
def test_bf16_multi_node_tree_backward_matches_fp32_kernels(K):
    """The branching tree of test_multi_node_tree_one_pass_equals_node_by_node in bf16 (head-resident backward: dQ tiles with
    branch keys, text dK / dV, slot rows of shared ancestors) against the fp32 kernels on the same rounded inputs."""
    from branchgan import ops
    gen = torch.Generator().manual_seed(3)
    B, H, Dh, S = 2, 2, 64, 50 if K == 7 else 130
    sparse = torch.tensor([0, 3, 10, 11, 30, 40, 48]) if K == 7 else torch.from_numpy(np.sort(np.random.RandomState(5).choice(S - 1, K, replace=False)))
    ctx_len = (sparse + 1).int().cuda()
    paths = [[0], [0, 1], [2], [2, 3], [2, 4]]
    N = len(paths)
    mk = lambda *s: torch.randn(*s, generator=gen).bfloat16()
    q, kt, vt = mk(B, H, N, K, Dh), mk(B, H, S, Dh), mk(B, H, S, Dh)
    ks, vs = [mk(B, H, K, Dh) for _ in range(N)], [mk(B, H, K, Dh) for _ in range(N)]
    gout = mk(B, H, N, K, Dh)
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        leaves = [t.to(dt).cuda().requires_grad_(True) for t in [q, kt, vt] + ks + vs]
        out, _ = ops.branch_attention(leaves[0], leaves[1], leaves[2], ctx_len, leaves[3:3 + N], leaves[3 + N:], paths=paths)
        res[dt] = [out.detach().float().cpu()] + [g.float().cpu() for g in torch.autograd.grad(out, leaves, gout.to(dt).cuda())]
    for i, (a, b_) in enumerate(zip(res[torch.bfloat16], res[torch.float32])):
        np.testing.assert_allclose(a.numpy(), b_.numpy(), atol=4e-2 if i in (2, 3) else 3e-2, err_msg=f"tensor {i}")
        assert float((a - b_).norm() / b_.norm()) < 1e-2, i
