"""Text generation and its metrics (the inference half): autoregressive decoding with this build's own KV
handling, self-cross-entropy and type/token ratios.

Reference: Inference/GenerateTexts.py:5-78 (`generateTexts` calls HuggingFace `generate` on the LM, `calculateSelfCrossEntropy`
re-scores contexts and generations) and Training/Metrics/GeneratedTextsMetric.py (TTR, self-cross-entropy summary).  HF
`generate` does not run on the reference's model classes under the installed transformers, so the decoding itself has no
reference output to pin against (parity unpinned for it); it is pinned by a property instead: with greedy selection the
cached decode must emit exactly the arg-max tokens of full re-forwards of the growing sequence (tests/test_gpu_generate.py).
The host-side metrics are checked against the reference's functions (tests/golden/g9_text_metrics.json).

Decoding reuses the branch machinery: the prompt is the text (one root pass, its per-layer K/V kept), every new token is a
branch node at the single branch point `S - 1` whose ancestors are the tokens generated so far - the same `bg_branch_attn_fwd`
call as a training branch step with K = 1, no concatenation per token.  Every `FOLD` tokens the slots are folded into the text
K/V (one concatenation per layer per 32 tokens) so that sequences longer than BG_MAX_SLOTS decode too.  Tokens are chosen by
`bg_topn_select`: arg-max, or an exponential race `p / e` (`e ~ Exp(1)`), which draws from the categorical distribution p.
"""
import numpy as np
import torch

from . import ops
from .models import BranchPast

FOLD = 32


def _lm_of(model):
    """Accepts a GannSetup, its generator wrapper, or the language model itself."""
    while not (hasattr(model, "transformer") or hasattr(model, "gpt_neox")) and hasattr(model, "generator"):
        model = model.generator
    return model


@torch.no_grad()
def generate(model, input_ids, max_new_tokens=128, do_sample=False, temperature=1.0, pad_token_id=0, eos_token_id=None, num_beams=1,
             num_return_sequences=1, generator=None, **unused):
    """HF-`generate`-shaped: (B, S) prompt ids -> (B, S + max_new_tokens) ids, prompt included.  Greedy (`do_sample=False`) or
    ancestral sampling at `temperature`; after `eos_token_id` a sequence is filled with `pad_token_id`.  Beam search and
    several return sequences are not offered.  `generator`: a torch.Generator on the device for reproducible sampling."""
    if num_beams != 1 or num_return_sequences != 1:
        raise NotImplementedError("generate: num_beams = num_return_sequences = 1 only")
    if do_sample and temperature <= 0:
        raise ValueError("generate: temperature must be positive")
    lm = _lm_of(model)
    ids = input_ids.long()
    B, S = ids.shape
    dev = ids.device
    n_pos = lm.config.max_position_embeddings
    if S + max_new_tokens > n_pos:
        raise ValueError(f"generate: {S} prompt + {max_new_tokens} new tokens exceed the model's {n_pos} positions")

    def select(logits):                                             # (B, V) -> (B, 1)
        if not do_sample:
            return ops.topn_select(logits, 1, blocked_value=float("-inf"))
        probs = torch.softmax(logits.float() / temperature, dim=-1)
        noise = -torch.log(torch.rand(probs.shape, device=dev, generator=generator) + 1e-14)
        return ops.topn_select(probs, 1, noise=noise)

    out = lm(input_ids=ids, use_cache=True, returnLmLogits=True)
    L = len(out.past_key_values)
    text = [out.past_key_values[l] for l in range(L)]
    slots = [[] for _ in range(L)]
    s_text = S
    done = torch.zeros(B, 1, dtype=torch.bool, device=dev)
    tok = select(out.logits[:, -1])
    new = []
    for t in range(max_new_tokens):
        if eos_token_id is not None:
            tok = torch.where(done, torch.full_like(tok, pad_token_id), tok)
            done = done | (tok == eos_token_id)
        new.append(tok)
        if t == max_new_tokens - 1:
            break
        d = len(slots[0])
        if d == FOLD:                                               # fold the generated tokens' K/V into the text
            text = [(torch.cat([text[l][0]] + [s[0] for s in slots[l]], dim=2), torch.cat([text[l][1]] + [s[1] for s in slots[l]], dim=2))
                    for l in range(L)]
            s_text, slots, d = s_text + d, [[] for _ in range(L)], 0
        ctx_len = torch.full((1,), s_text, dtype=torch.int32, device=dev)
        pos = torch.full((B, 1), s_text + d, dtype=torch.int32, device=dev)
        past = BranchPast(text=text, ancestors=slots, ctx_len=ctx_len, paths=[list(range(d + 1))], n_query_nodes=1)
        step = lm(input_ids=tok, position_ids=pos, textualPast=past, use_cache=True, returnLmLogits=True)
        for l in range(L):
            slots[l] = slots[l] + [step.past_key_values[l]]
        tok = select(step.logits[:, -1])
    return torch.cat([ids] + new, dim=1)


def generateTexts(generator, data, settings, device, tokenizer=None, generationLength=128, startIndex=None, maxNumSamples=None,
                  verbose=False):
    """GenerateTexts.py:5-38: -> ([(context ids, generated ids)], [(context text, generated text)] or None)."""
    results, seen = [], 0
    for batch in data:
        bIDs = torch.as_tensor(batch).to(device)
        bIDs = bIDs if startIndex is None else bIDs[:, startIndex:]
        seqs = generate(generator, bIDs, max_new_tokens=generationLength, pad_token_id=0, **settings)
        for c, g in zip(bIDs.tolist(), seqs[:, bIDs.shape[1]:].tolist()):
            results.append((c, g))
        seen += bIDs.shape[0]
        if maxNumSamples is not None and seen >= maxNumSamples:
            break
    if tokenizer is None:
        return results, None
    return results, [(tokenizer.decode(c), tokenizer.decode(g)) for c, g in results]


@torch.no_grad()
def _generated_part_loss(lm, ids, contextLength):
    """Mean -log p(token) over the positions after the context (GenerateTexts.py:47-52), from the fused softmax statistics."""
    logits = lm(input_ids=ids, returnLmLogits=True).logits
    nxt = torch.cat((ids[:, 1:], torch.full_like(ids[:, :1], -1)), dim=1).unsqueeze(-1)
    _, logp = ops.softmax_stats(logits, nxt)
    return (-logp[:, :-1, 0][:, contextLength:]).mean(-1)


def calculateSelfCrossEntropy(generator, device, data, generatedSeqs, verbose=False):
    """GenerateTexts.py:41-78: per-sample cross-entropy of the model on the ORIGINAL continuation of each context and on its
    own generation.  -> (originalCrossEntropy, generatedCrossEntropy), lists of floats."""
    lm = _lm_of(generator)
    contextLength, genLength = len(generatedSeqs[0][0]), len(generatedSeqs[0][1])
    stacked = torch.tensor([c + g for c, g in generatedSeqs])
    n, batchSize = len(generatedSeqs), None
    original, generated = [], []
    for batch in data:
        bIDs = torch.as_tensor(batch)[:, :contextLength + genLength].to(device)
        batchSize = bIDs.shape[0] if batchSize is None else batchSize
        original.extend(_generated_part_loss(lm, bIDs, contextLength).tolist())
        if len(original) >= n:
            break
    original = original[:n]
    for i in range(0, n, batchSize):
        generated.extend(_generated_part_loss(lm, stacked[i:i + batchSize].to(device), contextLength).tolist())
    return original, generated


def TTR(A, B):
    """GeneratedTextsMetric.py:3-17: share of distinct tokens in B, and of distinct tokens of B that do not occur in A."""
    a, b = set(A), set(B)
    return len(b) / len(B), len(b - a) / len(B)


def calculateTTR(sequencePairs):
    """GeneratedTextsMetric.py:19-34: means of TTR(context, generation) over the pairs."""
    m = np.mean(np.array([TTR(c, g) for c, g in sequencePairs]), axis=0)
    return m[0], m[1]


def selfCrossEntropy(originalCrossEntropy, generatedCrossEntropy):
    """GeneratedTextsMetric.py:37-51."""
    o, g = np.mean(originalCrossEntropy), np.mean(generatedCrossEntropy)
    return {'SelfCCE-Original': o, 'SelfCCE-Generated': g, 'SelfCCE-Difference': g - o,
            'SelfCCE-Original-Std': np.std(originalCrossEntropy), 'SelfCCE-Generated-Std': np.std(generatedCrossEntropy)}
