"""GPU parity of the rest of the drop-in boundary (round 2), against vectors captured from the reference
(tests/golden/g10_boundary.npz, g1_attention.npz, g2_sampling.npz):

  A3  default-mask branch attention (customCasualMask=None) through the modules' reference-signature methods, at head sizes the
      kernel takes: output, the weights under output_attentions=True, and all five gradients (fp32, |err| <= 2e-5);
  A1/A2 weights under output_attentions=True for real tree masks (g1), and for the root causal pass (rows sum to 1, upper
      triangle zero, weights @ v == output);
  L3/L4 dualHeadAdvantages / reinforceLoss / mirroredNegativeReinforceLoss with the reference's signatures (masks exact,
      values <= 1e-6 relative, gradients <= 1e-5 relative);
  S5  SamplingBuffer.rand == the reference's cat-by-split on the same stored noise; RANDOM / EPSILON_GREEDY ids under a fixed
      torch seed (the reference draws them from torch's global CPU generator: reproducible); apply_temperature vs the
      reference's outputs; SAMPLE / EPSILON_SAMPLE / EPSILON_SAMPLE_NO_REPLACEMENT / fresh-noise SAMPLE_NO_REPLACEMENT at the
      distribution level, the way the reference's own sanityCheckSampleWithoutReplacement does (SamplingStrategies.py:297-327)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _attention_module(fam, H, Dh, layer_idx=0, inv=False):
    from branchgan import configs, models
    if fam == "neox":
        return models.NeoXNonResidualAttention(configs.LMConfig("gpt_neox", 97, H * Dh, 2, H, 64)).cuda()
    cfg = configs.LMConfig("gpt2", 97, H * Dh, 2, H, 64)
    cfg.scale_attn_by_inverse_layer_idx = bool(inv)
    return models.GPT2NonResidualAttention(cfg, layer_idx=layer_idx).cuda()


def _call(mod, fam, *args, **kw):
    fn = mod._nonResidualTextAttention if fam == "neox" else mod.nonResidualTextAttention
    return fn(*args, **kw)


@pytest.mark.parametrize("fam", ["neox", "gpt2"])
def test_default_mask_attention_matches_reference(golden, fam):
    g = golden("g10_boundary")
    for Dh, H, Q, Kl in ((32, 2, 16, 40), (64, 1, 5, 9), (32, 1, 7, 7)):
        p = f"default_{fam}_{Dh}_{Q}_{Kl}_"
        t = {k: torch.from_numpy(g[p + k]).cuda().requires_grad_(True) for k in ("q", "k", "v", "kctx", "vctx")}
        mod = _attention_module(fam, H, Dh)
        out, w = _call(mod, fam, t["q"], t["k"], t["v"], t["kctx"], t["vctx"], output_attentions=True)
        np.testing.assert_allclose(out.detach().cpu().numpy(), g[p + "out"], atol=2e-5, rtol=0, err_msg=p)
        assert w.shape == g[p + "weights"].shape
        np.testing.assert_allclose(w.cpu().numpy(), g[p + "weights"], atol=2e-6, rtol=0, err_msg=p + "weights")
        out.backward(torch.from_numpy(g[p + "gout"]).cuda())
        for name, key in (("q", "dq"), ("k", "dk"), ("v", "dv"), ("kctx", "dkctx"), ("vctx", "dvctx")):
            np.testing.assert_allclose(t[name].grad.cpu().numpy(), g[p + key], atol=2e-5, rtol=0, err_msg=p + key)
        out2, none = _call(mod, fam, t["q"], t["k"], t["v"], t["kctx"], t["vctx"])
        assert none is None and torch.equal(out2, out)


def test_tree_mask_weights_under_output_attentions(golden):
    g = golden("g1_attention")
    checked = 0
    for ci, fam, H, Dh, K, S, depth, inv, layer in g["meta"].tolist():
        if Dh < 32:
            continue
        fam = "neox" if fam == 0 else "gpt2"
        mod = _attention_module(fam, H, Dh, layer_idx=layer, inv=inv)
        for d in range(1, depth + 1):
            p = f"c{ci}_d{d}_"
            t = {k: torch.from_numpy(g[p + k]).cuda() for k in ("q", "k", "v", "kctx", "vctx", "mask")}
            out, w = _call(mod, fam, t["q"], t["k"], t["v"], t["kctx"], t["vctx"], customCasualMask=t["mask"], output_attentions=True)
            np.testing.assert_allclose(out.cpu().numpy(), g[p + "out"], atol=2e-5, rtol=0, err_msg=p)
            np.testing.assert_allclose(w.cpu().numpy(), g[p + "weights"], atol=2e-6, rtol=0, err_msg=p + "weights")
            checked += 1
    assert checked >= 10


@pytest.mark.parametrize("fam", ["neox", "gpt2"])
def test_module_forward_returns_weights_on_request(fam):
    """Root causal pass through the module's forward: (attn_output, present, attn_weights) like the reference."""
    torch.manual_seed(0)
    H, Dh, B, T = 2, 32, 2, 19
    mod = _attention_module(fam, H, Dh)
    with torch.no_grad():
        for prm in mod.parameters():         # a bare Conv1D is uninitialised memory until the LM's init runs
            prm.normal_(0.0, 0.2)
    x = torch.randn(B, T, H * Dh, device="cuda")
    kw = dict(position_ids=torch.arange(T, device="cuda").expand(B, T)) if fam == "neox" else {}
    out, present, w = mod(x, use_cache=True, output_attentions=True, **kw)
    assert w.shape == (B, H, T, T)
    np.testing.assert_allclose(w.sum(-1).cpu().numpy(), 1.0, atol=1e-5)
    assert float(torch.triu(w, diagonal=1).abs().max()) == 0.0
    k, v = present
    ctx = (w @ v.float()).permute(0, 2, 1, 3).reshape(B, T, H * Dh)
    proj = mod.dense(ctx) if fam == "neox" else mod.c_proj(ctx)
    torch.testing.assert_close(proj, out, rtol=1e-4, atol=1e-5)
    assert len(mod(x, **kw)) == 2


def test_advantage_and_reinforce_functions_match_reference(golden):
    from types import SimpleNamespace
    from Models.Losses.AdvantageCalculations import dualHeadAdvantages
    from Models.Losses.ReinforceLoss import mirroredNegativeReinforceLoss, reinforceLoss
    from branchgan.configs import IncorrectValueHeadsPolicy as P
    g = golden("g10_boundary")
    t = lambda k: torch.from_numpy(g[k]).cuda()
    wA, wB, wC = g["adv_weights"].tolist()
    w = SimpleNamespace(A=wA, B=wB, C=wC)
    for tag, pi, pa in (("ii", P.IGNORE_SAMPLE, P.IGNORE_SAMPLE), ("ni", P.DO_NOTHING, P.IGNORE_SAMPLE), ("nn", P.DO_NOTHING, P.DO_NOTHING)):
        ia, aa, masks = dualHeadAdvantages(t("adv_in_scores"), t("adv_alt_scores"), t("adv_in_head"), t("adv_alt_head"), w, pi, pa)
        np.testing.assert_allclose(ia.cpu().numpy(), g[f"adv_{tag}_in_adv"], rtol=1e-6, atol=1e-7, err_msg=tag)
        np.testing.assert_allclose(aa.cpu().numpy(), g[f"adv_{tag}_alt_adv"], rtol=1e-6, atol=1e-7, err_msg=tag)
        for nm, x in zip(["zoneA", "zoneB", "zoneC", "adv_correct", "input_correct"], masks):
            assert np.array_equal(x.cpu().numpy(), g[f"adv_{tag}_{nm}"]), (tag, nm)
    adv, mc, gl = t("rl_adv"), t("rl_maskC"), t("rl_gloss")
    for tag, fn in (("mirrored", lambda p: mirroredNegativeReinforceLoss(p, adv, mc)), ("plain", lambda p: reinforceLoss(p, adv, False, False)),
                    ("minprob", lambda p: reinforceLoss(p, adv, False, True, negPushProbLimit=0.3))):
        probs = t("rl_probs").requires_grad_(True)
        loss = fn(probs)
        np.testing.assert_allclose(loss.detach().cpu().numpy(), g[f"rl_{tag}_loss"], rtol=2e-6, atol=1e-7, err_msg=tag)
        loss.backward(gl)
        np.testing.assert_allclose(probs.grad.cpu().numpy(), g[f"rl_{tag}_dprobs"], rtol=1e-5, atol=1e-6, err_msg=tag)


def test_sampling_buffer_roll_on_the_device(golden):
    from branchgan import sampling
    g = golden("g10_boundary")
    noise, splits, rolled = torch.from_numpy(g["buf_noise"]), g["buf_splits"], g["buf_rolled"]
    buf = sampling.SamplingBuffer("cuda", numShuffleSplits=4)
    buf.shapeRandBuffers[tuple(noise.shape)] = [noise.cuda(), [splits[0], splits[1]], 0]
    for c in range(6):                                   # wraps after 4 calls: the noise is REUSED, not redrawn (quirk 8)
        r = buf.rand(noise.shape)
        assert r.is_cuda and np.array_equal(r.cpu().numpy(), rolled[c]), c
    fresh = sampling.SamplingBuffer("cuda")
    a = fresh.rand((2, 3, 50))
    assert a.is_cuda and a.shape == (2, 3, 50) and float(a.min()) >= 0.0 and tuple(a.shape) in fresh.shapeRandBuffers
    stored = fresh.shapeRandBuffers[(2, 3, 50)][0]
    assert torch.equal(torch.sort(a.flatten()).values, torch.sort(stored.flatten()).values)     # a permutation of the stored draw


def test_global_rng_techniques_reproduce_the_reference_under_a_seed(golden):
    from branchgan.sampling import SamplingTechniques as T, sampleFromGeneratorDistribution
    g = golden("g10_boundary")
    probs, ids = torch.from_numpy(g["rand_probs"]).cuda(), torch.from_numpy(g["rand_ids"]).cuda()
    torch.manual_seed(77)
    alt, _ = sampleFromGeneratorDistribution(probs, 5, ids, T.RANDOM)
    assert alt.is_cuda and alt.dtype == torch.int64 and np.array_equal(alt.cpu().numpy(), g["rand_RANDOM"])
    torch.manual_seed(78)
    alt, _ = sampleFromGeneratorDistribution(probs, 6, ids, T.EPSILON_GREEDY, numRandomSamples=2)
    assert np.array_equal(alt.cpu().numpy(), g["rand_EPSILON_GREEDY"])
    torch.manual_seed(79)
    alt, _ = sampleFromGeneratorDistribution(probs, 6, ids, T.EPSILON_GREEDY, blockSamplingOfLabels=False, includeFinalTimeStep=True, numRandomSamples=4)
    assert np.array_equal(alt.cpu().numpy(), g["rand_EPSILON_GREEDY_final"])


def test_apply_temperature_on_the_device(golden):
    from branchgan.sampling import apply_temperature
    g = golden("g2_sampling")
    p = torch.from_numpy(g["temp_in"]).cuda()
    np.testing.assert_allclose(apply_temperature(p, 0.7).cpu().numpy(), g["temp_out_0p7"], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(apply_temperature(p, 1.5).cpu().numpy(), g["temp_out_1p5"], rtol=2e-6, atol=1e-9)


def test_random_draw_techniques_follow_the_distribution():
    """The reference's own sanity check, as a test: draw many times from a 5-way distribution, compare the empirical
    frequencies - of the first pick with the probabilities themselves, of a 3-pick without replacement with
    numpy.random.choice(replace=False) (SamplingStrategies.py:297-327)."""
    from branchgan.sampling import SamplingTechniques as T, sampleFromGeneratorDistribution
    dist = torch.tensor([0.5, 0.25, 0.125, 0.1, 0.025])
    R, V = 20000, 5
    probs = dist.view(1, 1, V).expand(1, R, V).contiguous().cuda()
    ids = torch.zeros((1, R + 1), dtype=torch.long, device="cuda")
    torch.manual_seed(3)
    kw = dict(blockSamplingOfLabels=False, includeFinalTimeStep=True)
    # SAMPLE: Categorical draws (with replacement): every column follows `dist`
    alt, _ = sampleFromGeneratorDistribution(probs, 2, ids, T.SAMPLE, **kw)
    assert alt.shape == (1, 2, R + 1) and alt.is_cuda
    for j in range(2):
        freq = torch.bincount(alt[0, j, 1:].cpu(), minlength=V).float() / R
        np.testing.assert_allclose(freq.numpy(), dist.numpy(), atol=0.012)
    # SAMPLE_NO_REPLACEMENT with the default (buffered) noise: distinct ids per row; inclusion frequencies == numpy's
    alt, _ = sampleFromGeneratorDistribution(probs, 3, ids, T.SAMPLE_NO_REPLACEMENT, **kw)
    picks = alt[0, :, 1:].t().cpu()                                        # (R, 3)
    assert bool((picks[:, 0] != picks[:, 1]).all() and (picks[:, 1] != picks[:, 2]).all() and (picks[:, 0] != picks[:, 2]).all())
    rng = np.random.RandomState(0)
    ref = np.stack([rng.choice(V, 3, replace=False, p=dist.numpy()) for _ in range(R)])
    got_incl = torch.bincount(picks.flatten(), minlength=V).float().numpy() / R
    ref_incl = np.bincount(ref.flatten(), minlength=V) / R
    np.testing.assert_allclose(got_incl, ref_incl, atol=0.015)
    np.testing.assert_allclose(torch.bincount(picks[:, 0], minlength=V).float().numpy() / R, dist.numpy(), atol=0.012)
    # the epsilon mixes: the first numSamples - numRandomSamples columns come from the distribution, the rest are uniform ids
    for tech in (T.EPSILON_SAMPLE, T.EPSILON_SAMPLE_NO_REPLACEMENT):
        alt, _ = sampleFromGeneratorDistribution(probs, 3, ids, tech, numRandomSamples=1, **kw)
        assert alt.shape == (1, 3, R + 1) and int(alt.min()) >= 0 and int(alt.max()) < V
        np.testing.assert_allclose(torch.bincount(alt[0, 0, 1:].cpu(), minlength=V).float().numpy() / R, dist.numpy(), atol=0.012)
        np.testing.assert_allclose(torch.bincount(alt[0, 2, 1:].cpu(), minlength=V).float().numpy() / R, np.full(V, 0.2), atol=0.012)
