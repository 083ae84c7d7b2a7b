"""GPU parity of the fused block glue (bg_add_layernorm_fwd / bg_layernorm_bwd / bg_bias_act_bwd, packed attention with
dq strides) against plain PyTorch fp32 references of the same ops, and of the fused GPT-2 block path against the
module-by-module path of the same model (which tests/test_gpu_tree.py pins to the reference's goldens).
Tolerances: fp32 kernels 2e-5 abs on O(1) values (reduction order differs); bf16 outputs half a bf16 ulp (2^-8 relative)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,C", [(37, 64), (256, 768), (1024, 2048), (5, 4096)])
@pytest.mark.parametrize("lowp", [False, True])
def test_add_layernorm_fwd_bwd(rows, C, lowp):
    from branchgan import fused
    fused.ACCUMULATE_IN_PLACE = True
    torch.manual_seed(rows + C)
    dt = torch.bfloat16 if lowp else torch.float32
    x = torch.randn(rows, C, device="cuda", requires_grad=True)
    delta = torch.randn(rows, C, device="cuda").to(dt).requires_grad_(True)
    ln = torch.nn.LayerNorm(C).cuda()
    with torch.no_grad():
        ln.weight.normal_(1.0, 0.2)
        ln.bias.normal_(0.0, 0.2)
    ln.weight.grad = torch.full_like(ln.weight, 0.5)          # the kernel accumulates
    ln.bias.grad = torch.full_like(ln.bias, -0.25)
    x_new, y = fused.add_layer_norm(x, delta, ln, dt)
    gx, gy = torch.randn_like(x_new), torch.randn(rows, C, device="cuda").to(dt)
    torch.autograd.backward([x_new, y], [gx, gy])
    dgam, dbet = ln.weight.grad.clone(), ln.bias.grad.clone()

    xr, dr = x.detach().clone().requires_grad_(True), delta.detach().float().requires_grad_(True)
    lr = torch.nn.LayerNorm(C).cuda()
    lr.load_state_dict(ln.state_dict())
    xs = xr + dr
    yr = lr(xs)
    torch.autograd.backward([xs, yr], [gx, gy.float()])
    assert torch.allclose(x_new, xs, atol=1e-6)
    tol = dict(atol=2e-2, rtol=2 ** -7) if lowp else dict(atol=2e-5, rtol=1e-5)
    assert torch.allclose(y.float(), yr, **tol)
    assert torch.allclose(x.grad, xr.grad, atol=2e-4, rtol=1e-4)
    assert torch.allclose(delta.grad.float(), dr.grad, **(dict(atol=3e-2, rtol=2 ** -7) if lowp else dict(atol=2e-4, rtol=1e-4)))
    scale = rows ** 0.5
    assert torch.allclose(dgam - 0.5, lr.weight.grad, atol=2e-4 * scale, rtol=1e-4)
    assert torch.allclose(dbet + 0.25, lr.bias.grad, atol=2e-4 * scale, rtol=1e-4)

    # plain LayerNorm (no residual input) through the same kernels
    x2 = x.detach().clone().requires_grad_(True)
    y2 = fused.layer_norm(x2, ln, dt)
    y2.backward(gy)
    x3 = x.detach().clone().requires_grad_(True)
    lr(x3).backward(gy.float())
    assert torch.allclose(x2.grad, x3.grad, atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("rows,n_out", [(33, 64), (256, 3072), (2048, 2304)])
@pytest.mark.parametrize("act", [None, "gelu_tanh", "gelu"])
@pytest.mark.parametrize("lowp", [False, True])
def test_linear_bias_act(rows, n_out, act, lowp):
    from branchgan import fused
    fused.ACCUMULATE_IN_PLACE = True
    torch.manual_seed(rows + n_out)
    dt = torch.bfloat16 if lowp else torch.float32
    n_in = 96
    x = (0.5 * torch.randn(rows, n_in, device="cuda")).to(dt).requires_grad_(True)
    w = torch.nn.Parameter(0.1 * torch.randn(n_in, n_out, device="cuda"))
    b = torch.nn.Parameter(0.1 * torch.randn(n_out, device="cuda"))
    y = fused.linear(x, w, b, "io", dt, act=act)
    gy = torch.randn(rows, n_out, device="cuda").to(dt)
    y.backward(gy)

    xr = x.detach().float().requires_grad_(True)
    wr, br = w.detach().to(dt).float().requires_grad_(True), b.detach().to(dt).float().requires_grad_(True)
    pre = xr @ wr + br
    if lowp:
        pre = pre.to(dt).float()
    yr = pre if act is None else F.gelu(pre, approximate="tanh" if act == "gelu_tanh" else "none")
    yr.backward(gy.float())
    tol = dict(atol=3e-2, rtol=2 ** -6) if lowp else dict(atol=5e-5, rtol=1e-4)
    assert torch.allclose(y.float(), yr, **tol)
    assert torch.allclose(x.grad.float(), xr.grad, **tol)
    gt = dict(atol=(0.15 if lowp else 2e-4) * rows ** 0.5, rtol=2e-2 if lowp else 1e-4)
    assert torch.allclose(w.grad, wr.grad, **gt)
    assert torch.allclose(b.grad, br.grad, **gt)


def _gpt2(hidden=64, heads=2, layers=2):
    from branchgan import configs, models
    cfg = configs.LMConfig("gpt2", 97, hidden, layers, heads, 64)
    torch.manual_seed(0)
    return models.NonResidualGPT2LMHeadModel(cfg).cuda()


@pytest.mark.parametrize("lm_head", [True, False])
def test_fused_gpt2_matches_modules(lm_head):
    """Root pass + one chain branch step + a 3-node one-pass tree, loss over everything; logits and every gradient of the
    fused path against the module-by-module path (fp32)."""
    from branchgan import fused, models
    model = _gpt2()
    B, S, K = 3, 24, 5
    ids = torch.randint(0, 97, (B, S), device="cuda")
    sparse = torch.tensor([2, 7, 11, 12, 20], device="cuda")
    ctx_len = (sparse + 1).to(torch.int32)
    b_ids = torch.randint(0, 97, (B, K), device="cuda")
    t_ids = torch.randint(0, 97, (B, 3 * K), device="cuda")

    def run(enabled):
        fused.ENABLED = enabled
        model.zero_grad(set_to_none=True)
        L = len(model.transformer.h)
        out = model(input_ids=ids, use_cache=True, returnLmLogits=lm_head)
        text = [out.past_key_values[l] for l in range(L)]
        pos1 = (sparse + 1).to(torch.int32).unsqueeze(0).expand(B, -1)
        past1 = models.BranchPast(text=text, ancestors=[[] for _ in range(L)], ctx_len=ctx_len, paths=[[0]], n_query_nodes=1)
        o1 = model(input_ids=b_ids, position_ids=pos1, textualPast=past1, use_cache=True, returnLmLogits=lm_head)
        past2 = models.BranchPast(text=text, ancestors=[[o1.past_key_values[l]] for l in range(L)], ctx_len=ctx_len, paths=[[0, 1]],
                                  n_query_nodes=1)
        o2 = model(input_ids=b_ids.flip(1), position_ids=pos1 + 1, textualPast=past2, use_cache=True, returnLmLogits=lm_head)
        # one-pass tree: node 0 and node 1 at depth 1, node 2 a child of node 0
        depths = torch.tensor([1, 1, 2], device="cuda")
        pos3 = (sparse.unsqueeze(0) + depths.unsqueeze(1)).reshape(1, 3 * K).expand(B, -1).to(torch.int32)
        past3 = models.BranchPast(text=text, ancestors=[[] for _ in range(L)], ctx_len=ctx_len, paths=[[0], [1], [0, 2]], n_query_nodes=3)
        o3 = model(input_ids=t_ids, position_ids=pos3, textualPast=past3, use_cache=True, returnLmLogits=lm_head)
        outs = [out.logits, o1.logits, o2.logits, o3.logits]
        loss = sum((o.float() ** 2).mean() * (i + 1) for i, o in enumerate(outs)) + o3.last_hidden_state.float().sum() * 1e-3
        loss.backward()
        return [o.detach().clone() for o in outs], {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    try:
        ref_out, ref_grad = run(False)
        got_out, got_grad = run(True)
    finally:
        fused.ENABLED = True
    for a, b in zip(got_out, ref_out):
        assert torch.allclose(a, b, atol=5e-5, rtol=1e-4)
    assert set(got_grad) == set(ref_grad)
    for n in ref_grad:
        assert torch.allclose(got_grad[n], ref_grad[n], atol=5e-5, rtol=2e-3), n


def test_fused_gpt2_bf16_autocast_close_to_modules():
    from branchgan import fused
    model = _gpt2(hidden=128, heads=2)
    ids = torch.randint(0, 97, (4, 32), device="cuda")

    def run(enabled):
        fused.ENABLED = enabled
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(input_ids=ids, returnLmLogits=True)
        F.cross_entropy(out.logits.float().flatten(0, 1), ids.flatten()).backward()
        return out.logits.float().detach(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    try:
        ref_out, ref_grad = run(False)
        got_out, got_grad = run(True)
    finally:
        fused.ENABLED = True
    assert got_out.dtype == torch.float32 and torch.allclose(got_out, ref_out, atol=3e-2, rtol=3e-2)
    for n in ref_grad:      # two bf16 computations of the same gradient: compare by relative Frobenius error
        err = (got_grad[n] - ref_grad[n]).norm() / (ref_grad[n].norm() + 1e-12)
        assert err < 5e-2, (n, float(err))


@pytest.mark.parametrize("rows,C", [(5, 64), (300, 768), (3072, 768), (1024, 1600)])
@pytest.mark.parametrize("lowp", [False, True])
@pytest.mark.parametrize("use_y", [True, False])
def test_projection_bias_gradient_from_layernorm_backward(rows, C, lowp, use_y):
    """linear(bias_grad=False) -> add_layer_norm(delta_bias=bias): the bias gradient comes out of the LayerNorm backward kernel
    (column sums of the rounded delta gradient) and equals what the projection's own column-sum pass accumulates - the same
    fp32 sums of the same values in another fixed order: rtol 1e-5, atol 1e-5 * sqrt(rows)."""
    from branchgan import fused
    fused.ACCUMULATE_IN_PLACE = True
    torch.manual_seed(rows + C)
    dt = torch.bfloat16 if lowp else torch.float32
    n_in = 96
    x = torch.randn(rows, C, device="cuda")
    h = (0.5 * torch.randn(rows, n_in, device="cuda")).to(dt)
    ln = torch.nn.LayerNorm(C).cuda()
    gx, gy = torch.randn(rows, C, device="cuda"), torch.randn(rows, C, device="cuda").to(dt)
    grads = []
    for split in (False, True):
        w = torch.nn.Parameter(0.1 * torch.randn(n_in, C, device="cuda", generator=torch.Generator("cuda").manual_seed(1)))
        b = torch.nn.Parameter(0.1 * torch.randn(C, device="cuda", generator=torch.Generator("cuda").manual_seed(2)))
        b.grad = torch.full_like(b, 0.125)              # accumulates
        ln.weight.grad = ln.bias.grad = None
        delta = fused.linear(h, w, b, "io", dt, bias_grad=not split)
        x_new, y = fused.add_layer_norm(x, delta, ln, dt, b if split else None)
        if use_y:
            torch.autograd.backward([x_new, y], [gx, gy])
        else:
            x_new.backward(gx)                          # only the residual output used: the fallback branch
        grads.append((b.grad.clone(), w.grad.clone()))
    torch.testing.assert_close(grads[1][0], grads[0][0], rtol=1e-5, atol=1e-5 * rows ** 0.5)
    assert torch.equal(grads[1][1], grads[0][1])
    assert float((grads[0][0] - 0.125).abs().max()) > 0.1


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_attention_backward_adds_incoming_text_gradients(dt):
    """bg_branch_attn_bwd(dk_text_in, dv_text_in): the text dK / dV other consumers already produced are added inside the dkv
    kernel (fp32 add, one rounding), also when they live in the output buffer itself.  fp32: equal to the separate add to
    1e-6; bf16: within one bf16 rounding of it."""
    from branchgan import fused, ops
    torch.manual_seed(3)
    B, H, S, Dh, K, N = 2, 3, 40, 64, 5, 2
    scale = Dh ** -0.5
    root = torch.randn(B, S, 3, H, Dh, device="cuda").to(dt)
    tree = torch.randn(B, N * K, 3, H, Dh, device="cuda").to(dt)
    kt, vt = root[:, :, 1].permute(0, 2, 1, 3), root[:, :, 2].permute(0, 2, 1, 3)
    tk, tv = tree[:, :, 1].permute(0, 2, 1, 3), tree[:, :, 2].permute(0, 2, 1, 3)
    own = lambda t: [t[:, :, n * K:(n + 1) * K] for n in range(N)]
    ctx = torch.tensor([3, 9, 17, 30, 40], dtype=torch.int32, device="cuda")
    paths = [[0], [0, 1]]
    hf = fused._heads_first
    att = torch.empty(B, N * K, H, Dh, device="cuda", dtype=dt)
    lse = torch.empty(B, H, N, K, device="cuda")
    fused._attn_fwd_call(hf(tree[:, :, 0], N, K), kt, vt, own(tk), own(tv), ctx, paths, scale, hf(att, N, K), lse)
    do = torch.randn_like(att)

    def run(dk_in, dv_in, alias):
        dtree, dtext = torch.zeros_like(tree), torch.zeros_like(root)
        dk, dv = dtext[:, :, 1].permute(0, 2, 1, 3), dtext[:, :, 2].permute(0, 2, 1, 3)
        if alias:
            dk.copy_(dk_in)
            dv.copy_(dv_in)
            dk_in, dv_in = dk, dv
        dtk, dtv = dtree[:, :, 1].permute(0, 2, 1, 3), dtree[:, :, 2].permute(0, 2, 1, 3)
        fused._attn_bwd_call(hf(tree[:, :, 0], N, K), kt, vt, own(tk), own(tv), ctx, paths, scale, hf(att, N, K), lse, hf(do, N, K),
                             hf(dtree[:, :, 0], N, K), dk, dv, own(dtk), own(dtv), dkt_in=dk_in, dvt_in=dv_in)
        return dk.float().clone(), dv.float().clone(), dtree.float().clone()

    inc = torch.randn(B, S, 3, H, Dh, device="cuda").to(dt)
    ik, iv = inc[:, :, 1].permute(0, 2, 1, 3), inc[:, :, 2].permute(0, 2, 1, 3)
    dk0, dv0, dq0 = run(None, None, False)
    for alias in (False, True):
        dk1, dv1, dq1 = run(ik, iv, alias)
        assert torch.equal(dq1, dq0)
        tol = dict(rtol=1e-6, atol=1e-6) if dt == torch.float32 else dict(rtol=2 ** -7, atol=2 ** -7)
        torch.testing.assert_close(dk1, dk0 + ik.float(), **tol)
        torch.testing.assert_close(dv1, dv0 + iv.float(), **tol)


@pytest.mark.parametrize("lowp", [False, True])
def test_add_layernorm_with_two_updates(lowp):
    """add_layer_norm(x, a, ln, delta_b=m) = LayerNorm(x + a + m) of a parallel-residual block (NeoXNonResidualBase.py:108-133:
    `mlp + attn + hidden`), both updates receiving the same gradient.  fp32: 1e-5; bf16 updates: the sum is formed in fp32
    here and in bf16 by the reference, so within one bf16 rounding of the updates."""
    from branchgan import fused
    torch.manual_seed(5)
    rows, C = 300, 512
    dt = torch.bfloat16 if lowp else torch.float32
    x = torch.randn(rows, C, device="cuda", requires_grad=True)
    a = torch.randn(rows, C, device="cuda").to(dt).requires_grad_(True)
    m = torch.randn(rows, C, device="cuda").to(dt).requires_grad_(True)
    ln = torch.nn.LayerNorm(C).cuda()
    x_new, y = fused.add_layer_norm(x, a, ln, dt, delta_b=m)
    gx, gy = torch.randn_like(x_new), torch.randn(rows, C, device="cuda").to(dt)
    torch.autograd.backward([x_new, y], [gx, gy])
    xr, ar, mr = x.detach().clone().requires_grad_(True), a.detach().float().requires_grad_(True), m.detach().float().requires_grad_(True)
    lr = torch.nn.LayerNorm(C).cuda()
    xs = mr + ar + xr
    yr = lr(xs)
    torch.autograd.backward([xs, yr], [gx, gy.float()])
    torch.testing.assert_close(x_new, xs, rtol=1e-6, atol=1e-5)
    tol = dict(atol=3e-2, rtol=2 ** -7) if lowp else dict(atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(y.float(), yr, **tol)
    torch.testing.assert_close(x.grad, xr.grad, atol=2e-4, rtol=1e-4)
    assert torch.equal(a.grad, m.grad)
    torch.testing.assert_close(a.grad.float(), ar.grad, **(dict(atol=3e-2, rtol=2 ** -7) if lowp else dict(atol=2e-4, rtol=1e-4)))


def test_partial_reduce_batches_descriptors():
    """bg_partial_reduce through the C-ABI: 230 reductions of ragged shapes (three launches of <= 96 descriptors) against
    torch column sums; accumulates into `out`.  fp32, fixed summation order: 1e-5 relative to the column's absolute sum."""
    import ctypes
    from branchgan import _capi as C
    g = torch.Generator().manual_seed(3)
    shapes = [(int(torch.randint(1, 300, (1,), generator=g)), int(torch.randint(1, 900, (1,), generator=g))) for _ in range(230)]
    parts, outs, want = [], [], []
    descs = (C.ReduceDesc * len(shapes))()
    for d, (G, n) in zip(descs, shapes):
        stride = n + int(torch.randint(0, 9, (1,), generator=g))
        p = torch.randn(G, stride, generator=g).cuda()
        o = torch.randn(n, generator=g).cuda()
        want.append((o + p[:, :n].double().sum(0).float(), p[:, :n].abs().sum(0)))
        parts.append(p)
        outs.append(o)
        d.partial, d.out, d.groups, d.stride, d.n = p.data_ptr(), o.data_ptr(), G, stride, n
    C.check(C.lib().bg_partial_reduce(descs, len(shapes), C.stream_ptr()), "bg_partial_reduce")
    torch.cuda.synchronize()
    for o, (w, mag) in zip(outs, want):
        assert bool(((o - w).abs() <= 1e-5 * mag + 1e-6).all())
    # five reductions into ONE output (a LayerNorm shared by five passes) and one into an overlapping slice of it
    same = torch.zeros(700, device="cuda")
    ps = [torch.randn(40, 700, generator=g).cuda() for _ in range(6)]
    d6 = (C.ReduceDesc * 6)()
    for i, (d, p) in enumerate(zip(d6, ps)):
        d.partial, d.out, d.groups, d.stride, d.n = p.data_ptr(), same.data_ptr() + (400 if i == 5 else 0), 40, 700, 100 if i == 5 else 700
    C.check(C.lib().bg_partial_reduce(d6, 6, C.stream_ptr()), "bg_partial_reduce")
    want6 = sum(p.double().sum(0) for p in ps[:5])
    want6[100:200] += ps[5][:, :100].double().sum(0)
    assert torch.allclose(same.double(), want6, atol=1e-4)
    assert C.lib().bg_partial_reduce(descs, 0, C.stream_ptr()) == 0
    descs[5].n = descs[5].stride + 1
    assert C.lib().bg_partial_reduce(descs, len(shapes), C.stream_ptr()) != 0
    assert b"descriptor 5" in C.lib().bg_last_error()


def test_deferred_reductions_match_immediate_ones(monkeypatch):
    """The column reductions collected until the end of the backward pass (fused._DeferredReductions) give bit-identical
    gradients to reducing behind every producer, and `backward()` returns with them complete - also when a previous backward
    pass died half way and left reductions behind."""
    from branchgan import fused
    model = _gpt2(hidden=128, heads=4, layers=3)
    ids = torch.randint(0, 97, (4, 48), device="cuda")

    def run(defer, dtype):
        monkeypatch.setattr(fused, "DEFER_REDUCTIONS", defer)
        grads = []
        for _ in range(2):                       # the second pass accumulates on top of the first
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype is not None):
                out = model(input_ids=ids, use_cache=True, returnLmLogits=True)
                (out.logits.float() ** 2).mean().backward()
            grads.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
        model.zero_grad(set_to_none=True)
        return grads

    for dtype in (None, torch.bfloat16):
        now = run(False, dtype)
        d = fused._deferred_for(ids.device)
        # what a backward pass that raised before its callback ran leaves behind: a pending reduction, armed for a dead task.
        # The next pass must reduce it (exactly once) before it reuses the workspace, and then run normally.
        left_part, left_out = torch.ones(3, 5, device="cuda"), torch.zeros(5, device="cuda")
        d.pending.append((left_part.data_ptr(), left_out, 3, 5, 5))
        d.armed, d.task = True, -5
        later = run(True, dtype)
        assert not d.pending and not d.armed
        assert torch.equal(left_out, torch.full((5,), 3.0, device="cuda"))
        for a, b in zip(now, later):
            assert set(a) == set(b)
            for n in a:
                assert torch.equal(a[n], b[n]), n


def test_stack_rows_matches_cat():
    """bg_stack_rows (the operand stacking of the batched weight-gradient GEMMs) against torch.cat: contiguous sources, a source with
    padded rows, bf16 and fp32, more descriptors than one launch holds; a source that is not 16-byte regular takes the cat path."""
    from branchgan import fused
    g = torch.Generator(device="cuda").manual_seed(3)
    for dtype, width in ((torch.bfloat16, 768), (torch.float32, 100)):
        xs = [torch.randn(37 + i, width, device="cuda", generator=g).to(dtype) for i in range(30)]
        wide = torch.randn(50, width + 64, device="cuda", generator=g).to(dtype)
        xs.append(wide[:, :width])                                   # rows 64 elements apart from dense
        dys = [torch.randn(x.shape[0], 3 * width, device="cuda", generator=g).to(dtype) for x in xs]
        R = sum(x.shape[0] for x in xs)
        xa = torch.zeros(R, width, dtype=dtype, device="cuda")
        da = torch.zeros(R, 3 * width, dtype=dtype, device="cuda")
        fused._stack_rows(xs, xa, dys, da)
        assert torch.equal(xa, torch.cat(xs)) and torch.equal(da, torch.cat(dys))
    odd = [torch.randn(5, 6, device="cuda", generator=g)]             # 24-byte rows: not regular
    out = torch.zeros(5, 6, device="cuda")
    dout = torch.zeros(5, 6, device="cuda")
    fused._stack_rows(odd, out, odd, dout)
    assert torch.equal(out, odd[0]) and torch.equal(dout, odd[0])


def test_stacked_activations_written_in_place(monkeypatch):
    """Once a model's batched weight-gradient GEMMs have run, their stacked operands stay and the producers (LayerNorm, attention,
    GELU forward; LayerNorm, attention, GELU backward) write them in place: nothing is copied any more, and the gradients are
    bit-identical to the all-copied form.  Root pass + tree pass share the weights (two passes per stack)."""
    from branchgan import fused, models, training
    if not fused.BATCH_DW:
        pytest.skip("batched weight-gradient GEMMs are switched off (A/B run)")
    model = _gpt2(hidden=128, heads=4, layers=3)
    B, S, K = 2, 40, 4
    g = torch.Generator(device="cuda").manual_seed(5)
    ids = torch.randint(0, 97, (B, S), device="cuda", generator=g)
    t_ids = torch.randint(0, 97, (B, 3 * K), device="cuda", generator=g)
    sparse = torch.tensor([3, 9, 20, 31], device="cuda")
    ctx_len = (sparse + 1).to(torch.int32)
    depths = torch.tensor([1, 1, 2], device="cuda")
    pos3 = (sparse.unsqueeze(0) + depths.unsqueeze(1)).reshape(1, 3 * K).expand(B, -1).to(torch.int32)
    L = len(model.transformer.h)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(input_ids=ids, use_cache=True)
            text = [out.past_key_values[l] for l in range(L)]
            past = models.BranchPast(text=text, ancestors=[[] for _ in range(L)], ctx_len=ctx_len, paths=[[0], [1], [0, 2]], n_query_nodes=3)
            o3 = model(input_ids=t_ids, position_ids=pos3, textualPast=past, use_cache=True)
        ((out.logits.float() ** 2).mean() + (o3.logits.float() ** 2).mean() * 2).backward()

    step()                                           # gradients exist; the fp32-output GEMM form is probed
    arena = training.GradArena(model)
    arena._build()                                   # the layers' gradients at one constant stride: the GEMMs batch from here on

    def grads(stacked):
        monkeypatch.setattr(fused, "STACKED_ACTIVATIONS", stacked)
        arena.zero()
        fused.stack_stats.update(copied=0, in_place=0)
        step()
        torch.cuda.synchronize()
        return arena.arena.clone(), dict(fused.stack_stats)

    ref, s0 = grads(False)
    assert s0["in_place"] == 0 and s0["copied"] == 4 * 2 * 2 * L          # 4 weight kinds x (x, dy) x 2 passes x layers
    first, s1 = grads(True)                          # lays the stacks down (still copies)
    second, s2 = grads(True)
    third, s3 = grads(True)
    assert s1["copied"] == s0["copied"]
    assert s2 == {"copied": 0, "in_place": s0["copied"]} and s3 == s2, (s2, s3)
    assert float(ref.abs().sum()) > 0
    # the stacks hold the passes in forward order, the all-copied form in the order the backward pass reached them: the same
    # products summed in another order
    assert torch.allclose(first, ref, atol=1e-5, rtol=1e-4)
    ref = first
    for got in (second, third):
        assert torch.equal(got, ref)
    # a forward pass whose backward never runs does not keep the slabs: its autograd nodes go, the slabs are free again
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lost = model(input_ids=ids, use_cache=True).logits
    del lost
    again, s4 = grads(True)
    assert s4 == s2 and torch.equal(again, ref)
    # ... and while such a pass is alive the next one simply allocates as before (nothing is overwritten)
    def root_only():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return (model(input_ids=ids.flip(1), use_cache=True).logits.float() ** 2).mean()

    arena.zero()
    root_only().backward()
    alone = arena.arena.clone()
    grads(True)
    grads(True)                                      # two-pass layout again, slabs in use
    held = root_only()                               # sits in the slabs of pass 0 ...
    other, s5 = grads(True)                          # ... while a whole step runs: it must neither use nor overwrite them
    assert torch.equal(other, ref) and s5["copied"] > 0
    arena.zero()
    held.backward()
    assert torch.equal(arena.arena, alone)
    del held
    relaid, s6 = grads(True)                       # the late backward pass laid out stacks of its own: back to the two-pass layout
    last, s7 = grads(True)
    assert torch.equal(relaid, ref) and s7 == s2 and torch.equal(last, ref)


@pytest.mark.parametrize("lowp", [False, True])
@pytest.mark.parametrize("n_delta", [0, 1, 2])
def test_add_layernorm_pair_equals_two_layernorms(lowp, n_delta):
    """fused.add_layer_norm_pair (one kernel, two LayerNorms of x + deltas: the parallel-residual NeoX block) against
    add_layer_norm / layer_norm one after the other: outputs bit-identical, every gradient equal to summation order."""
    from branchgan import fused
    torch.manual_seed(4)
    rows, C_ = 300, 1024
    dt = torch.bfloat16 if lowp else torch.float32
    ln_a, ln_b = torch.nn.LayerNorm(C_).cuda(), torch.nn.LayerNorm(C_).cuda()
    for ln in (ln_a, ln_b):
        ln.weight.data.normal_(1.0, 0.2)
        ln.bias.data.normal_(0.0, 0.2)
    x0 = torch.randn(2, rows // 2, C_, device="cuda")
    d0 = [torch.randn(2, rows // 2, C_, device="cuda").to(dt) for _ in range(n_delta)]
    wa, wb, wx = (torch.randn(2, rows // 2, C_, device="cuda") for _ in range(3))

    def run(pair):
        for ln in (ln_a, ln_b):
            ln.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        ds = [d.clone().requires_grad_(True) for d in d0]
        if pair:
            x1, ya, yb = fused.add_layer_norm_pair(x, ds[0] if ds else None, ln_a, ln_b, dt, delta_b=ds[1] if n_delta == 2 else None)
        elif not ds:
            x1, ya, yb = x, fused.layer_norm(x, ln_a, dt), fused.layer_norm(x, ln_b, dt)
        else:
            x1, ya = fused.add_layer_norm(x, ds[0], ln_a, dt, delta_b=ds[1] if n_delta == 2 else None)
            yb = fused.layer_norm(x1, ln_b, dt)
        ((ya.float() * wa).sum() + (yb.float() * wb).sum() * 0.5 + (x1 * wx).sum() * 0.25).backward()
        grads = [x.grad] + [d.grad for d in ds] + [ln_a.weight.grad, ln_a.bias.grad, ln_b.weight.grad, ln_b.bias.grad]
        return (x1.detach(), ya.detach(), yb.detach()), [g.detach().clone().float() for g in grads]

    (x1_r, ya_r, yb_r), g_ref = run(False)
    (x1_p, ya_p, yb_p), g_pair = run(True)
    assert torch.equal(x1_p, x1_r) and torch.equal(ya_p, ya_r) and torch.equal(yb_p, yb_r)
    for a, b in zip(g_pair, g_ref):
        tol = 2e-2 if lowp else 1e-4
        assert float((a - b).abs().max()) <= tol * (1.0 + float(b.abs().max())), float((a - b).abs().max())
