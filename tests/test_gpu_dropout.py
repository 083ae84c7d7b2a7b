"""GPU parity: dropout on the attention probabilities inside the attention kernels (csrc/attn_dropout.h).

The reference draws its mask from torch's global generator inside F.dropout (GPT-2: NonResidualAttention.py:107-109,157-159,
always on; NeoX: NeoXNonResidualAttention.py:224,287), which no other implementation can reproduce bit for bit.  What CAN be
held to the oracle's bar: given the SAME draw, the same numbers.  bg_attn_dropout_mask writes out the keep decisions the
kernels regenerate; the oracle applies them the way F.dropout does (w * keep / (1 - p)); forward output, every gradient and
the weights handed out under output_attentions must then agree to the attention tests' tolerances (fp32 2e-5, bf16 3e-2).
The draw itself is held to statistical properties (keep rate, independence of rows / heads / calls / steps)."""
import math

import numpy as np
import pytest
import torch

from oracle import attention as A

pytestmark = pytest.mark.gpu


def _reference_keep(keep, S, K, n_anc):
    """Kernel layout (B,H,K,S + n_anc + 1) [text | path entry p] -> the reference's (B,H,K,S + n_anc*K + 1) column order
    [text | one K-column block per ancestor, query k uses column k | self]; positions no query can see are kept (masked anyway)."""
    B, H = keep.shape[:2]
    out = torch.ones((B, H, K, S + n_anc * K + 1), dtype=torch.bool)
    out[..., :S] = keep[..., :S].bool()
    for a in range(n_anc):
        out[:, :, torch.arange(K), S + a * K + torch.arange(K)] = keep[..., S + a].bool()
    out[..., -1] = keep[..., S + n_anc].bool()
    return out


def _mask_of(ops, q5, kt, slots, max_path, drop, scale):
    sh = ops._attn_shape(q5, kt, slots, scale, max_path, q5, drop)
    keep = ops.attn_dropout_mask(sh, q5.device)
    torch.cuda.synchronize()
    return keep.cpu()


@pytest.mark.parametrize("B,H,Dh,S,K,depth", [(2, 3, 64, 256, 32, 4), (1, 2, 128, 130, 40, 2), (2, 2, 32, 70, 33, 3), (1, 12, 64, 512, 64, 8)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("p", [0.1, 0.5])
def test_branch_attention_with_the_same_draw_as_the_oracle(B, H, Dh, S, K, depth, dtype, p):
    from branchgan import ops
    if p == 0.5 and (dtype == torch.bfloat16 or Dh != 64):
        pytest.skip("p = 0.5 on the fp32 Dh 64 cases only")
    gen = torch.Generator().manual_seed(B * 1000 + Dh + S)
    sparse = np.sort(np.random.RandomState(S).choice(S - 1, K, replace=False))
    r = lambda *s: torch.randn(*s, generator=gen).to(dtype).float()
    q, k, v = r(B, H, K, Dh), r(B, H, K, Dh), r(B, H, K, Dh)
    n_anc = depth - 1
    C = S + n_anc * K
    kctx, vctx, gout = r(B, H, C, Dh), r(B, H, C, Dh), r(B, H, K, Dh)
    scale = 1.0 / math.sqrt(Dh)
    cast = lambda t: t.to(dtype).cuda().requires_grad_(True)
    lq, lkt, lvt = cast(q), cast(kctx[:, :, :S].contiguous()), cast(vctx[:, :, :S].contiguous())
    lks = [cast(kctx[:, :, S + j * K:S + (j + 1) * K].contiguous()) for j in range(n_anc)] + [cast(k)]
    lvs = [cast(vctx[:, :, S + j * K:S + (j + 1) * K].contiguous()) for j in range(n_anc)] + [cast(v)]
    ctx_len = (torch.as_tensor(sparse) + 1).int().cuda()
    ops.seed_dropout(1234 + Dh)
    drop = ops.drop_descriptor(p)
    out, lse = ops.branch_attention(lq, lkt, lvt, ctx_len, lks, lvs, scale=scale, drop=drop)
    out.backward(gout.to(dtype).cuda())
    keep = _mask_of(ops, lq.detach().unsqueeze(2), lkt.detach(), [t.detach() for t in lks], depth, drop, scale)
    assert keep.shape == (B, H, K, S + depth)
    rate = keep.float().mean().item()
    assert abs(rate - (1 - p)) < 4 * math.sqrt(p * (1 - p) / keep.numel()) + 1e-3, rate

    mask = A.tree_mask_from_descriptor(torch.as_tensor(sparse) + 1, n_anc, S, K)
    lv = [t.clone().requires_grad_(True) for t in (q, k, v, kctx, vctx)]
    o_ref, w_ref = A.branch_attention(A.NEOX, *lv, mask=mask, keep=_reference_keep(keep, S, K, n_anc), p_drop=p)
    g_ref = torch.autograd.grad(o_ref, lv, gout)
    tol = (2e-5 if dtype == torch.float32 else 3e-2) / (1 - p)
    f = lambda t: t.detach().float().cpu().numpy()
    np.testing.assert_allclose(f(out), o_ref.detach().numpy(), atol=tol)
    np.testing.assert_allclose(f(lq.grad), g_ref[0].numpy(), atol=tol)
    np.testing.assert_allclose(f(lks[-1].grad), g_ref[1].numpy(), atol=tol)
    np.testing.assert_allclose(f(lvs[-1].grad), g_ref[2].numpy(), atol=tol)
    dkctx = torch.cat([lkt.grad] + [x.grad for x in lks[:-1]], dim=2)
    dvctx = torch.cat([lvt.grad] + [x.grad for x in lvs[:-1]], dim=2)
    np.testing.assert_allclose(f(dkctx), g_ref[3].numpy(), atol=tol * 4)
    np.testing.assert_allclose(f(dvctx), g_ref[4].numpy(), atol=tol * 4)
    # lse is that of the undropped softmax: the same call without dropout returns the same lse
    _, lse0 = ops.branch_attention(lq.detach(), lkt.detach(), lvt.detach(), ctx_len, [t.detach() for t in lks], [t.detach() for t in lvs], scale=scale)
    np.testing.assert_allclose(lse.cpu().numpy(), lse0.cpu().numpy(), atol=2e-6, rtol=1e-6)      # (to summation order: the key-range split)
    if dtype == torch.float32:
        # the weights the modules hand out under output_attentions are the ones after dropout, in the reference's column order
        w = ops.branch_attention_weights(lq.detach(), lkt.detach(), ctx_len, [t.detach() for t in lks[:-1]], [list(range(depth))], scale, lse,
                                         own_k=lks[-1].detach().unsqueeze(2), drop=drop)
        np.testing.assert_allclose(w.cpu().numpy(), w_ref.detach().numpy(), atol=2e-5 / (1 - p))


@pytest.mark.parametrize("fam", [A.NEOX, A.GPT2])
@pytest.mark.parametrize("S", [200, 512])
def test_root_causal_attention_with_the_same_draw(fam, S):
    from branchgan import ops
    gen = torch.Generator().manual_seed(9)
    B, H, Dh, p = 2, 3, 64, 0.1
    q, k, v, go = [torch.randn(B, H, S, Dh, generator=gen) for _ in range(4)]
    ld = [t.cuda().requires_grad_(True) for t in (q, k, v)]
    ops.seed_dropout(77)
    drop = ops.drop_descriptor(p)
    o, _ = ops.causal_attention(*ld, drop=drop)
    g = torch.autograd.grad(o, ld, go.cuda())
    keep = _mask_of(ops, ld[0].detach().unsqueeze(2), ld[1].detach(), [], 0, drop, 1.0 / math.sqrt(Dh))      # (B,H,S,S)
    lv = [t.clone().requires_grad_(True) for t in (q, k, v)]
    o_ref, _ = A.causal_attention(fam, *lv, keep=keep.bool(), p_drop=p)
    g_ref = torch.autograd.grad(o_ref, lv, go)
    np.testing.assert_allclose(o.detach().cpu().numpy(), o_ref.detach().numpy(), atol=3e-5)
    for a, b in zip(g, g_ref):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), atol=1.5e-4)


def test_multi_node_tree_pass_with_dropout_equals_the_oracle_per_node():
    """One launch over the nodes of a branching tree: every (node, branch point) row has its own draw; gradients of shared
    slots and of the text accumulate over nodes.  Each node is checked against the oracle with that node's rows of the mask."""
    from branchgan import ops
    gen = torch.Generator().manual_seed(3)
    B, H, Dh, S, K, p = 2, 2, 64, 130, 40, 0.2
    sparse = torch.from_numpy(np.sort(np.random.RandomState(5).choice(S - 1, K, replace=False)))
    ctx_len = (sparse + 1).int().cuda()
    paths = [[0], [0, 1], [2], [2, 3], [2, 4]]
    N = len(paths)
    mk = lambda *s: torch.randn(*s, generator=gen)
    q, kt, vt = mk(B, H, N, K, Dh), mk(B, H, S, Dh), mk(B, H, S, Dh)
    ks, vs = [mk(B, H, K, Dh) for _ in range(N)], [mk(B, H, K, Dh) for _ in range(N)]
    gout = mk(B, H, N, K, Dh)
    dev = [t.cuda().requires_grad_(True) for t in [q, kt, vt] + ks + vs]
    ops.seed_dropout(5)
    drop = ops.drop_descriptor(p)
    out, _ = ops.branch_attention(dev[0], dev[1], dev[2], ctx_len, dev[3:3 + N], dev[3 + N:], paths=paths, drop=drop)
    g_dev = torch.autograd.grad(out, dev, gout.cuda())
    keep = _mask_of(ops, dev[0].detach(), dev[1].detach(), [t.detach() for t in dev[3:3 + N]], 2, drop, 1.0 / math.sqrt(Dh))   # (B,H,N*K,S+2)
    keep = keep.view(B, H, N, K, S + 2)
    cpu = [t.clone().requires_grad_(True) for t in [q, kt, vt] + ks + vs]
    outs = []
    for n, path in enumerate(paths):
        n_anc = len(path) - 1
        kctx = torch.cat([cpu[1]] + [cpu[3 + s] for s in path[:-1]], dim=2)
        vctx = torch.cat([cpu[2]] + [cpu[3 + N + s] for s in path[:-1]], dim=2)
        mask = A.tree_mask_from_descriptor(sparse + 1, n_anc, S, K)
        kp = _reference_keep(keep[:, :, n, :, :S + n_anc + 1], S, K, n_anc)
        o, _ = A.branch_attention(A.NEOX, cpu[0][:, :, n], cpu[3 + path[-1]], cpu[3 + N + path[-1]], kctx, vctx, mask=mask, keep=kp, p_drop=p)
        outs.append(o)
    o_ref = torch.stack(outs, dim=2)
    g_ref = torch.autograd.grad(o_ref, cpu, gout)
    np.testing.assert_allclose(out.detach().cpu().numpy(), o_ref.detach().numpy(), atol=3e-5)
    for a, b in zip(g_dev, g_ref):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), atol=2e-4)


def test_draw_statistics_and_counter():
    """Keep rate per row and per column, independence between neighbouring rows, heads, calls and steps; the same
    (seed, step, call) gives the same draw; bg_dropout_advance changes it."""
    from branchgan import ops
    B, H, K, S, Dh, p = 2, 4, 256, 1024, 64, 0.1
    q5 = torch.empty(B, H, 1, K, Dh, device="cuda")
    kt = torch.empty(B, H, S, Dh, device="cuda")
    ops.seed_dropout(42)
    d1, d2 = ops.drop_descriptor(p), ops.drop_descriptor(p)
    m1 = _mask_of(ops, q5, kt, [], 0, d1, 1.0).float()
    assert torch.equal(m1, _mask_of(ops, q5, kt, [], 0, d1, 1.0).float())
    m2 = _mask_of(ops, q5, kt, [], 0, d2, 1.0).float()
    ops.dropout_advance("cuda")
    m3 = _mask_of(ops, q5, kt, [], 0, d1, 1.0).float()
    ops.seed_dropout(43)
    m4 = _mask_of(ops, q5, kt, [], 0, d1, 1.0).float()
    n = m1.numel()
    sd = math.sqrt(p * (1 - p))
    assert abs(m1.mean().item() - (1 - p)) < 4 * sd / math.sqrt(n)
    assert (m1.mean(dim=-1) - (1 - p)).abs().max().item() < 5.5 * sd / math.sqrt(S)          # every row
    assert (m1.mean(dim=(0, 1, 2)) - (1 - p)).abs().max().item() < 5.5 * sd / math.sqrt(B * H * K)   # every column
    corr = lambda a, b: (((a - a.mean()) * (b - b.mean())).mean() / (a.std() * b.std())).item()
    lim = 5 / math.sqrt(n / 2)
    assert abs(corr(m1[:, :, :-1], m1[:, :, 1:])) < lim          # neighbouring rows
    assert abs(corr(m1[..., :-1], m1[..., 1:])) < lim            # neighbouring columns
    assert abs(corr(m1[:, :-1], m1[:, 1:])) < lim                # heads
    assert abs(corr(m1[:1], m1[1:])) < lim * 1.5                 # batch entries
    for other in (m2, m3, m4):                                   # another call, another step, another seed
        assert abs(corr(m1, other)) < lim


@pytest.mark.parametrize("family", ["gpt2", "gpt_neox"])
def test_models_with_dropout_run_fused_and_module_paths(family):
    """A GPT-2 config with the stock rates (attn / resid / embd 0.1) and a NeoX config with attention / hidden dropout run on
    the fused path and module by module: finite losses and gradients, fresh draws step to step, and - quirk 1 - GPT-2 drops in
    eval mode as well while NeoX does not."""
    from branchgan import configs, fused, models
    torch.manual_seed(0)
    cfg = configs.LMConfig(family, 101, 64, 2, 2, 48, attn_pdrop=0.1, resid_pdrop=0.1, embd_pdrop=0.1 if family == "gpt2" else 0.0)
    lm = models.make_lm(cfg).cuda()
    ids = torch.randint(0, 101, (2, 24)).cuda()
    for use_fused in (True, False):
        fused.ENABLED = use_fused
        try:
            lm.train()
            before = fused.module_path_forwards
            a = lm(input_ids=ids, returnLmLogits=True).logits
            assert (fused.module_path_forwards == before) == use_fused
            b = lm(input_ids=ids, returnLmLogits=True).logits
            assert torch.isfinite(a).all() and not torch.equal(a, b)
            a.float().square().mean().backward()
            assert all(torch.isfinite(p.grad).all() for p in lm.parameters() if p.grad is not None)
            lm.zero_grad()
            lm.eval()
            c, d = lm(input_ids=ids, returnLmLogits=True).logits, lm(input_ids=ids, returnLmLogits=True).logits
            assert torch.equal(c, d) == (family == "gpt_neox")
        finally:
            fused.ENABLED = True


def test_gpt2_attention_dropout_only_matches_between_paths_in_expectation():
    """attn_pdrop alone (no other dropout): the mean of many fused-path forwards approaches the no-dropout forward."""
    from branchgan import configs, models
    torch.manual_seed(1)
    base = configs.LMConfig("gpt2", 101, 64, 1, 2, 48)
    drop = configs.LMConfig("gpt2", 101, 64, 1, 2, 48, attn_pdrop=0.2)
    lm0, lm1 = models.make_lm(base).cuda(), models.make_lm(drop).cuda()
    lm1.load_state_dict(lm0.state_dict())
    ids = torch.randint(0, 101, (2, 32)).cuda()
    with torch.no_grad():
        ref = lm0(input_ids=ids, returnLmLogits=True).logits
        acc = torch.zeros_like(ref)
        n = 400
        for _ in range(n):
            acc += lm1(input_ids=ids, returnLmLogits=True).logits
        one = lm1(input_ids=ids, returnLmLogits=True).logits
    err_mean = (acc / n - ref).abs().mean().item()
    err_one = (one - ref).abs().mean().item()
    assert err_one > 0 and err_mean < 0.25 * err_one, (err_mean, err_one)


def test_rejected_shapes():
    from branchgan import ops
    q = torch.randn(1, 1, 1, 8, 256, device="cuda", dtype=torch.bfloat16)
    kt = torch.randn(1, 1, 16, 256, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(Exception, match="Dh <= 128"):
        ops.branch_attention(q, kt, kt, torch.arange(1, 9, dtype=torch.int32, device="cuda"), drop=ops.drop_descriptor(0.1))
    with pytest.raises(ValueError):
        ops.drop_descriptor(1.0)


def test_captured_step_with_attention_dropout_draws_fresh_masks_and_matches_eager():
    """A GPT-2 Branch-GAN step with the stock dropout rates (0.1 everywhere) captured into HIP graphs: (i) on the same (seed, counter) a replay
    reproduces the eager step - every loss and every element of both gradient arenas, so backward regenerates the forward's
    masks inside the graphs too; (ii) the counter advances inside the captured region: two replays on the same batch differ."""
    import dataclasses
    from branchgan import configs, gan, ops, sampling, training
    from branchgan.graphs import GraphedTrainStep
    torch.manual_seed(0)
    # the stock GPT-2 rates: attention probabilities (attention kernels), attention output and MLP output (LayerNorm kernels),
    # embeddings (bg_dropout_rows)
    cfg = dataclasses.replace(configs.LMConfig.from_any("gpt2-tiny"), attn_pdrop=0.1, resid_pdrop=0.1, embd_pdrop=0.1)
    model = gan.GannSetup(cfg, cfg, mleLossWeight=1, rlLossWeight=1, discValueHeadLossWeight=0.25, device="cuda").cuda()
    rules = [sampling.SamplingRule(1, sampling.SamplingTechniques.TOP_K, {}) for _ in range(3)]
    conf = configs.SamplingConfig(rules, includeAllMultiStepSamplesForDisc=True, numSampleSequences=8, seed=77)
    conf.sparseIndex = torch.tensor([2, 9, 17, 25, 33, 41, 50, 60], device="cuda")
    rl = configs.ReinforcementLearningConfig(labelMode=configs.ReinforcementLabelMode.ALTERATIONS_AND_INPUT_UNWEIGHTED,
                                             sectionWeightA=1, sectionWeightB=1, sectionWeightC=1.5, teacherForcing=True)
    g_opt = training.OptimizerWrapper(model.generator, training.make_optimizer(model.generator.parameters(), 'adam', 1e-4, betas=(0.5, 0.999)))
    d_opt = training.OptimizerWrapper(model.discriminator, training.make_optimizer(model.discriminator.parameters(), 'adam', 1e-3))
    data = torch.randint(0, cfg.vocab_size, (3, 4, 64), generator=torch.Generator().manual_seed(5)).cuda()
    gs = GraphedTrainStep(model, conf, g_opt, d_opt, data[0], 9999, rl, None, warmup=1)

    def replay(ids):
        gs.static_ids.copy_(ids)
        gs.replay_graphs()
        torch.cuda.synchronize()
        out = (float(gs.losses[0]["loss"]), float(gs.losses[1]["loss"]), g_opt.arena.arena.clone(), d_opt.arena.arena.clone())
        g_opt.arena.zero(); d_opt.arena.zero()
        return out

    for r in range(2):
        ids = data[1 + r]
        ops.seed_dropout(100 + r)
        got = replay(ids)
        again = replay(ids)                                   # counter advanced by the replay itself: another draw
        assert got[0] != again[0] and got[1] != again[1]
        assert not torch.equal(got[2], again[2])
        ops.seed_dropout(100 + r)
        d, g = model.trainForwardPassNoMetrics(ids, conf, 9999, collectTreeDepthData=False, reinforcementLearningConfig=rl)
        g["loss"].backward()
        d["loss"].backward()
        torch.cuda.synchronize()
        np.testing.assert_allclose(got[0], float(d["loss"]), rtol=1e-5)
        np.testing.assert_allclose(got[1], float(g["loss"]), rtol=1e-5)
        for name, a, b in (("G", got[2], g_opt.arena.arena), ("D", got[3], d_opt.arena.arena)):
            err = float((a - b).abs().max()) / float(b.abs().max())
            assert err <= 1e-5, f"replay {r} {name} arena: {err:.3e}"
        g_opt.arena.zero(); d_opt.arena.zero()


@pytest.mark.parametrize("dtype,cols", [(torch.float32, 768), (torch.bfloat16, 768), (torch.bfloat16, 100), (torch.float32, 37)])
def test_dropout_rows_is_a_function_of_site_and_logical_row(dtype, cols):
    """bg_dropout_rows: keep rate and scaling; the decision of a token depends on (seed, step, site, batch, LOGICAL row, column)
    only - a call over one tree node's rows (row0 = base + i K) reproduces rows [i K, (i + 1) K) of the call over all nodes,
    text rows keep their own numbers in a merged call - and the backward pass applies the forward's mask to the gradient."""
    from branchgan import ops
    B, S, N, K, p = 3, 20, 4, 8, 0.25
    ops.seed_dropout(11)
    site = ops.drop_site(1, 5, ops.SITE_RESID)
    x = (torch.rand(B, S + N * K, cols, device="cuda") + 0.5).to(dtype)
    base = ops.TREE_ROW_BASE
    full = ops.dropout_rows(x, p, site, base, S)                           # merged: S text rows, then N nodes of K rows
    kept = full != 0
    assert abs(float(kept.float().mean()) - (1 - p)) < 0.02
    torch.testing.assert_close(full[kept].float(), (x[kept].float() / (1 - p)).to(dtype).float(), rtol=1e-2 if dtype == torch.bfloat16 else 1e-6, atol=0)
    text = ops.dropout_rows(x[:, :S].contiguous(), p, site, 0, S)          # the text alone
    assert torch.equal(text, full[:, :S])
    tree = ops.dropout_rows(x[:, S:].contiguous(), p, site, base, 0)       # the tree alone
    assert torch.equal(tree, full[:, S:])
    for i in range(N):                                                     # one node alone (a branch step)
        rows = slice(S + i * K, S + (i + 1) * K)
        one = ops.dropout_rows(x[:, rows].contiguous(), p, site, base + i * K, 0)
        assert torch.equal(one, full[:, rows]), i
    other_site = ops.dropout_rows(x, p, ops.drop_site(1, 6, ops.SITE_RESID), base, S)
    assert not torch.equal(other_site != 0, kept)
    per_batch = kept[:, :, : min(cols, 32)].flatten(1).float()
    assert not torch.equal(per_batch[0], per_batch[1])                     # batch rows draw their own
    ops.dropout_advance("cuda")
    assert not torch.equal(ops.dropout_rows(x, p, site, base, S) != 0, kept)     # next step: another draw
    ops.seed_dropout(11)
    xr = x.float().requires_grad_(True)
    y = ops.dropout_rows(xr, p, site, base, S)
    g = torch.randn_like(y)
    y.backward(g)
    torch.testing.assert_close(xr.grad, g * (y.detach() != 0).float() / (1 - p))


def test_attention_dropout_rows_line_up_between_branch_step_and_tree_pass():
    """The attention-probability mask of a call over ONE tree node with drop_row0 = base + i K equals rows [i K, (i + 1) K) of
    the mask of the call over all nodes (drop_row0 = base) at the same site."""
    from branchgan import ops
    B, H, N, K, S, Dh = 2, 3, 3, 16, 40, 32
    ops.seed_dropout(3)
    site = ops.drop_site(1, 2, ops.SITE_ATTN)
    q_all = torch.zeros(B, H, N, K, Dh, device="cuda")
    kt = torch.zeros(B, H, S, Dh, device="cuda")
    slots = [torch.zeros(B, H, K, Dh, device="cuda") for _ in range(N)]
    full = _mask_of(ops, q_all, kt, slots, 3, ops.drop_descriptor(0.3, site, ops.TREE_ROW_BASE), 1.0)        # (B,H,N*K,S+3)
    for i in range(N):
        one = _mask_of(ops, q_all[:, :, :1], kt, slots, 3, ops.drop_descriptor(0.3, site, ops.TREE_ROW_BASE + i * K), 1.0)
        assert torch.equal(one, full[:, :, i * K:(i + 1) * K]), i
    assert abs(float(full.float().mean()) - 0.7) < 0.02


def test_generator_forward_under_dropout_samples_without_autograd_and_matches_the_per_node_graph(monkeypatch):
    """GPT-2 generator with the stock dropout rates: the sampling phase runs without autograd and ONE tree pass rebuilds the
    nodes' logits (as without dropout), because the masks are functions of (step, site, logical row).  Against the per-node
    autograd graph (the reference's schedule) on the same (seed, step): the same sampled ids at every node, the same log
    probabilities and the same parameter gradients to fp32 kernel tolerance."""
    import dataclasses
    from branchgan import configs, models, ops, sampling, tree
    torch.manual_seed(0)
    cfg = dataclasses.replace(configs.LMConfig.from_any("gpt2-tiny"), attn_pdrop=0.1, resid_pdrop=0.1, embd_pdrop=0.1)
    gen = models.NonResidualGenerator(cfg).cuda()
    gen.train()
    assert models.dropout_active(gen) and models.dropout_by_site(gen)
    ids = torch.randint(0, cfg.vocab_size, (3, 48), generator=torch.Generator().manual_seed(2)).cuda()
    sparse = torch.tensor([1, 5, 9, 14, 20, 27, 33, 40], device="cuda")
    noise = lambda shape, device: torch.rand(shape, generator=torch.Generator().manual_seed(9)).to(device)

    tree_passes = []
    attach = tree._attach_branch_gradients
    monkeypatch.setattr(tree, "_attach_branch_gradients", lambda *a, **k: (tree_passes.append(1), attach(*a, **k))[1])

    def run(defer):
        monkeypatch.setattr(tree, "DEFER_UNDER_DROPOUT", defer)
        rules = [sampling.SamplingRule(2, sampling.SamplingTechniques.TOP_K, {}), sampling.SamplingRule(1, sampling.SamplingTechniques.TOP_K, {}),
                 sampling.SamplingRule(1, sampling.SamplingTechniques.SAMPLE_NO_REPLACEMENT, {})]
        ops.seed_dropout(21)
        ops.dropout_advance("cuda")
        gen.zero_grad()
        root, *_ = tree.multiStepGenForward(gen, ids, rules, "cuda", clearTreePast=False, sparseIndex=sparse, noise_fn=noise)
        nodes = [n for n in root.getSubNodes() if n.depth > 0]
        loss = sum(n.logProbs.sum() for n in nodes) + root.labelLogProbs.sum()
        loss.backward()
        grads = {k: p.grad.clone() for k, p in gen.named_parameters() if p.grad is not None}
        return [n.ids.clone() for n in nodes], [n.logProbs.detach().clone() for n in nodes], grads, float(loss)

    ids_d, lp_d, g_d, loss_d = run(True)
    assert len(tree_passes) == 1                      # sampled without autograd, differentiated in one tree pass
    ids_n, lp_n, g_n, loss_n = run(False)
    assert len(tree_passes) == 1                      # the per-node graph: no tree pass
    assert len(ids_d) == len(ids_n) == 2 + 2 + 2
    for a, b in zip(ids_d, ids_n):
        assert torch.equal(a, b)
    for a, b in zip(lp_d, lp_n):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-4)
    assert abs(loss_d - loss_n) <= 2e-4 * abs(loss_n)
    for k in g_n:
        err = float((g_d[k] - g_n[k]).abs().max()) / (float(g_n[k].abs().max()) + 1e-12)
        assert err < 2e-3, (k, err)
    # and the masks ARE in play: another step draws other masks, other numbers
    monkeypatch.setattr(tree, "DEFER_UNDER_DROPOUT", True)
    ops.seed_dropout(22)
    root, *_ = tree.multiStepGenForward(gen, ids, [sampling.SamplingRule(2, sampling.SamplingTechniques.TOP_K, {})], "cuda", sparseIndex=sparse)
    assert abs(float(root.labelLogProbs.sum()) - float(sum(0 for _ in ()))) > 0 and not torch.allclose(root.children[0].logProbs.detach(), lp_d[0])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm_kernels_with_dropout_equal_dropout_then_add_layernorm(dtype):
    """x + dropout(delta) inside the LayerNorm kernels (bg_add_layernorm_dropout_fwd / bg_layernorm_dropout_bwd) against
    bg_dropout_rows followed by the plain kernels: the same residual stream and LayerNorm output, and the same gradients for
    x, delta, gamma, beta - fp32 to the last bit, bf16 to its rounding (the fused form does not round the scaled delta)."""
    from branchgan import fused, ops
    B, S, N, K, Cc, p = 2, 24, 3, 8, 256, 0.2
    T = S + N * K
    ops.seed_dropout(5)
    site = ops.drop_site(2, 3, ops.SITE_MLP)
    g = torch.Generator(device="cuda").manual_seed(1)
    x0 = torch.randn(B, T, Cc, device="cuda", generator=g)
    d0 = torch.randn(B, T, Cc, device="cuda", generator=g).to(dtype)
    gy = torch.randn(B, T, Cc, device="cuda", generator=g).to(dtype)
    gx = torch.randn(B, T, Cc, device="cuda", generator=g)
    res = []
    for fused_drop in (True, False):
        ln = torch.nn.LayerNorm(Cc).cuda()
        with torch.no_grad():
            ln.weight.copy_(torch.linspace(0.5, 1.5, Cc)); ln.bias.copy_(torch.linspace(-0.2, 0.2, Cc))
        x, d = x0.clone().requires_grad_(True), d0.clone().requires_grad_(True)
        if fused_drop:
            x1, y = fused.add_layer_norm(x, d, ln, dtype, drop=(p, site, ops.TREE_ROW_BASE, S))
        else:
            x1, y = fused.add_layer_norm(x, ops.dropout_rows(d, p, site, ops.TREE_ROW_BASE, S), ln, dtype)
        torch.autograd.backward([x1, y], [gx, gy])
        torch.cuda.synchronize()
        res.append((x1.detach(), y.detach(), x.grad, d.grad, ln.weight.grad, ln.bias.grad))
    tol = dict(rtol=0, atol=0) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    for a, b in zip(*res):
        torch.testing.assert_close(a.float(), b.float(), **tol)
    kept = (res[0][3] != 0).float().mean().item()
    assert abs(kept - (1 - p)) < 0.03


def test_pass_numbers_do_not_wrap_into_repeated_masks():
    """A caller that never advances the step counter: after 2046 forwards over a text the pass numbers run out and the counter
    is advanced on its behalf - the masks of pass 1 do not come back."""
    from branchgan import ops
    ops.seed_dropout(1)
    ops.dropout_advance("cuda")
    x = torch.ones(1, 4, 64, device="cuda")
    ops.dropout_new_pass("cuda")
    first = ops.dropout_rows(x, 0.5, ops.drop_site(1, 0, ops.SITE_EMBED))
    step0 = int(ops.dropout_state("cuda")[1])
    for _ in range(0x7FF):
        ops.dropout_new_pass("cuda")
    assert int(ops.dropout_state("cuda")[1]) == step0 + 1 and ops._drop_pass == 2
    ops._drop_pass = 1
    again = ops.dropout_rows(x, 0.5, ops.drop_site(1, 0, ops.SITE_EMBED))
    assert not torch.equal(first, again)


def test_gpt2_lm_under_stock_dropout_matches_the_oracle_given_the_same_draws():
    """The GPT-2 LM's root pass with the stock rates on the fused path - embedding dropout (bg_dropout_rows), attention
    probabilities (attention kernel), attention output and MLP output (LayerNorm kernels) - against the CPU oracle
    (oracle.models.lm_forward, the reference's op order) that is handed the SAME Bernoulli draws, regenerated here from the
    sites the model uses: logits and every parameter gradient to fp32 tolerance.  The reference's own draw comes from torch's
    generator inside F.dropout and cannot be reproduced; what is held to the oracle is everything else."""
    import dataclasses
    from branchgan import configs, models, ops
    from oracle import models as OM
    torch.manual_seed(3)
    p = 0.1
    cfg = dataclasses.replace(configs.LMConfig.from_any("gpt2-tiny"), attn_pdrop=p, resid_pdrop=p, embd_pdrop=p)
    lm = models.make_lm(cfg).cuda()
    lm.train()
    lm._drop_salt = 1
    B, T = 2, 40
    ids = torch.randint(0, cfg.vocab_size, (B, T), generator=torch.Generator().manual_seed(4))
    ops.seed_dropout(9)
    ops.dropout_advance("cuda")
    logits = lm(input_ids=ids.cuda(), returnLmLogits=True).logits
    assert ops._drop_pass == 1                                 # the pass number the sites of this forward carry
    w = torch.randn(logits.shape, generator=torch.Generator().manual_seed(5))
    (logits.float() * w.cuda()).sum().backward()
    torch.cuda.synchronize()

    # the draws, regenerated from (seed, step, site): ones through the same functions
    H, Dh, C = cfg.num_attention_heads, cfg.head_dim, cfg.hidden_size
    ones = torch.ones(B, T, C, device="cuda")
    rows_keep = lambda site: (ops.dropout_rows(ones, p, site) != 0).cpu()
    drop = {"embed": (rows_keep(ops.drop_site(1, 0x1FF, ops.SITE_EMBED)), p)}
    q5 = torch.zeros(B, H, 1, T, Dh, device="cuda")
    kt = torch.zeros(B, H, T, Dh, device="cuda")
    for i in range(cfg.num_hidden_layers):
        drop[("resid", i)] = (rows_keep(ops.drop_site(1, i, ops.SITE_RESID)), p)
        drop[("mlp", i)] = (rows_keep(ops.drop_site(1, i, ops.SITE_MLP)), p)
        keep = _mask_of(ops, q5, kt, [], 0, ops.drop_descriptor(p, ops.drop_site(1, i, ops.SITE_ATTN), 0), 1.0)      # (B, H, T, T)
        drop[("attn", i)] = (keep.bool(), p)
    for k, (m, _) in drop.items():
        assert abs(float(m.float().mean()) - (1 - p)) < 0.03, k

    sd = {k: v.detach().cpu().float().requires_grad_(True) for k, v in lm.state_dict().items()}
    ocfg = OM.LMConfig("gpt2", cfg.num_hidden_layers, H, C, cfg.vocab_size, cfg.max_position_embeddings)
    ref, _, _ = OM.lm_forward(sd, "", ocfg, ids, drop=drop)
    torch.testing.assert_close(logits.float().cpu(), ref, rtol=2e-4, atol=2e-4)
    plain, _, _ = OM.lm_forward(sd, "", ocfg, ids)
    assert float((plain - ref).detach().abs().max()) > 0.05     # the masks matter
    (ref * w).sum().backward()
    checked = 0
    for name, prm in lm.named_parameters():
        if prm.grad is None or name == "valueHead.weight":
            continue
        g_ref = sd[name].grad
        err = float((prm.grad.cpu() - g_ref).abs().max()) / (float(g_ref.abs().max()) + 1e-12)
        assert err < 2e-3, (name, err)
        checked += 1
    assert checked >= 2 + 12 * cfg.num_hidden_layers
