"""Autoregressive decoding and the generated-text metrics (inference half).

HF `generate` cannot run on the reference's model classes here, so the decode loop has no reference OUTPUT to pin against.  It
is anchored to the oracle instead: greedy decoding through the KV slots / folded text K/V must emit, token for token, the
arg-max of oracle.models.lm_forward (the CPU restatement of the reference's LM forward, itself pinned by the g35 goldens)
re-run over the growing sequence - which is what HF's greedy `generate` computes by definition - for both model families and
across a fold boundary (more than 32 new tokens); the self-cross-entropy of Inference/GenerateTexts.py:41-78 is compared
with the same oracle.  A second, oracle-free form of the property (re-forwards on the HIP model) stays beside it.  The host
metrics are compared with the reference's GeneratedTextsMetric functions (tests/golden/g9_text_metrics.json)."""
import json
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_text_metrics_match_reference():
    from branchgan import generate
    with open(os.path.join(GOLDEN, "g9_text_metrics.json")) as f:
        g = json.load(f)
    assert [list(generate.TTR(c, s)) for c, s in g["pairs"]] == g["ttr"]
    np.testing.assert_allclose(generate.calculateTTR(g["pairs"]), g["mean_ttr"], rtol=1e-12)
    got = generate.selfCrossEntropy(g["original"], g["generated"])
    assert set(got) == set(g["self_cce"])
    for k, v in g["self_cce"].items():
        np.testing.assert_allclose(got[k], v, rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["gpt2", "gpt_neox"])
def test_greedy_decode_equals_full_reforward(family):
    from branchgan import configs, generate, models
    torch.manual_seed(3)
    cfg = configs.LMConfig(family, 211, 128, 2, 4, 128)
    lm = models.make_lm(cfg).cuda()
    with torch.no_grad():
        for p in lm.parameters():                      # large random weights: decisive arg-maxes, no near ties
            p.mul_(8.0) if p.dim() > 1 else None
    prompt = torch.randint(0, 211, (3, 9), device="cuda")
    out = generate.generate(lm, prompt, max_new_tokens=40)            # 40 > FOLD: slots are folded into the text once
    assert out.shape == (3, 49) and torch.equal(out[:, :9], prompt)
    with torch.no_grad():
        for t in range(9, 49):
            logits = lm(input_ids=out[:, :t], returnLmLogits=True).logits[:, -1].float()
            top2 = logits.topk(2, dim=-1).values
            decisive = (top2[:, 0] - top2[:, 1]) > 1e-3                 # kernels of two batch shapes: skip genuine near ties
            assert torch.equal(out[decisive, t], logits.argmax(-1)[decisive]), t
    # sampling: reproducible with a generator, tokens in range, eos handling pads the tail
    g1 = generate.generate(lm, prompt, max_new_tokens=12, do_sample=True, temperature=0.7, generator=torch.Generator(device="cuda").manual_seed(5))
    g2 = generate.generate(lm, prompt, max_new_tokens=12, do_sample=True, temperature=0.7, generator=torch.Generator(device="cuda").manual_seed(5))
    assert torch.equal(g1, g2) and int(g1.min()) >= 0 and int(g1.max()) < 211
    eos = int(out[0, 12])
    padded = generate.generate(lm, prompt, max_new_tokens=40, eos_token_id=eos, pad_token_id=0)
    row = padded[0, 9:].tolist()
    first = row.index(eos)
    assert all(v == 0 for v in row[first + 1:]) and row[:first + 1] == out[0, 9:9 + first + 1].tolist()
    with pytest.raises(NotImplementedError):
        generate.generate(lm, prompt, max_new_tokens=4, num_beams=2)
    with pytest.raises(ValueError, match="positions"):
        generate.generate(lm, prompt, max_new_tokens=200)


def _oracle_lm(lm, cfg):
    from oracle import models as OM
    sd = {k: v.detach().cpu().float() for k, v in lm.state_dict().items()}
    ocfg = OM.LMConfig("neox" if cfg.model_type == "gpt_neox" else "gpt2", cfg.num_hidden_layers, cfg.num_attention_heads, cfg.hidden_size,
                       cfg.vocab_size, cfg.max_position_embeddings)
    return lambda ids: OM.lm_forward(sd, "", ocfg, ids.cpu())[0]


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["gpt2", "gpt_neox"])
def test_greedy_decode_equals_oracle_argmax(family):
    """GenerateTexts.py:18 (`generator.generate(...)`, greedy): token t is the arg-max of the LM's logits at position t - 1 of the
    sequence so far.  The oracle re-forwards the growing sequence on the CPU in fp32; the cached HIP decode must agree wherever
    the oracle's two best logits are more than 1e-3 apart (all but a handful of positions, counted)."""
    from branchgan import configs, generate, models
    torch.manual_seed(4)
    cfg = configs.LMConfig(family, 211, 128, 2, 4, 128)
    lm = models.make_lm(cfg).cuda()
    with torch.no_grad():
        for p in lm.parameters():
            p.mul_(8.0) if p.dim() > 1 else None
    forward = _oracle_lm(lm, cfg)
    prompt = torch.randint(0, 211, (3, 9), device="cuda")
    out = generate.generate(lm, prompt, max_new_tokens=40).cpu()
    compared = 0
    with torch.no_grad():
        for t in range(9, 49):
            logits = forward(out[:, :t])[:, -1]
            top2 = logits.topk(2, dim=-1).values
            decisive = (top2[:, 0] - top2[:, 1]) > 1e-3
            assert torch.equal(out[decisive, t], logits.argmax(-1)[decisive]), (family, t)
            compared += int(decisive.sum())
    assert compared >= 110                                               # of 120 decoded tokens
    # self-cross-entropy (GenerateTexts.py:41-78) of contexts + generations against the oracle's log-softmax
    pairs = [(row[:9].tolist(), row[9:].tolist()) for row in out]
    data = [out]                                                         # "original" continuation = the same tokens here
    original, generated = generate.calculateSelfCrossEntropy(lm, "cuda", data, pairs)
    with torch.no_grad():
        lp = torch.log_softmax(forward(out), -1)
        want = -torch.gather(lp[:, :-1], -1, out[:, 1:].unsqueeze(-1)).squeeze(-1)[:, 9:].mean(-1)
    np.testing.assert_allclose(generated, want.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(original, want.numpy(), rtol=2e-4, atol=2e-4)


@pytest.mark.gpu
def test_generate_texts_and_self_cross_entropy():
    from branchgan import gan, generate
    torch.manual_seed(0)
    model = gan.GannSetup("gpt2-tiny", "gpt2-tiny", device="cuda").cuda()
    data = [torch.randint(0, 2048, (4, 40)) for _ in range(3)]
    settings = {'temperature': 1.0, 'num_beams': 1, 'do_sample': False, 'num_return_sequences': 1}
    results, texts = generate.generateTexts(model.generator, data, settings, "cuda", generationLength=8, startIndex=16, maxNumSamples=8)
    assert texts is None and len(results) == 8 and all(len(c) == 24 and len(g) == 8 for c, g in results)
    # the contexts the generations were made from are columns [16, 40) of the data; score against data cut the same way
    cut = [b[:, 16:] for b in data]
    original, generated = generate.calculateSelfCrossEntropy(model.generator, "cuda", [torch.cat((b, b[:, :8]), dim=1) for b in cut], results)
    assert len(original) == len(generated) == 8 and np.isfinite(original).all() and np.isfinite(generated).all()
    # greedy generations are the model's own arg-max: it must find them likelier than random continuations
    assert np.mean(generated) < np.mean(original)
    lm = model.generator.generator
    with torch.no_grad():
        ids = torch.tensor([c + g for c, g in results[:4]], device="cuda")
        lp = torch.log_softmax(lm(input_ids=ids, returnLmLogits=True).logits.float(), -1)
        want = -torch.gather(lp[:, :-1], -1, ids[:, 1:].unsqueeze(-1)).squeeze(-1)[:, 24:].mean(-1)
    np.testing.assert_allclose(generated[:4], want.cpu().numpy(), rtol=1e-4, atol=1e-4)
    summary = generate.selfCrossEntropy(original, generated)
    ttr, ctx_ttr = generate.calculateTTR(results)
    assert 0 < ttr <= 1 and 0 <= ctx_ttr <= ttr and summary['SelfCCE-Difference'] < 0
