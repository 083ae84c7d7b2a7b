"""Oracle: the two LM families with branch plumbing, as pure functions over a reference state dict.

Restates Models/NeoXBranchAttention/NeoXNonResidualBase.py, Models/GPT2BranchAttention/
NonResidualDiscriminatorBase.py, Models/Generator.py and Models/Discriminator.py (reference).
Test infrastructure only (see oracle/__init__.py).  Dropout is the identity (p = 0 in parity runs).
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from . import attention as A


@dataclass
class LMConfig:
    family: str            # "neox" | "gpt2"
    n_layer: int
    n_head: int
    hidden: int
    vocab: int
    max_pos: int
    rotary_pct: float = 0.25
    rotary_base: float = 10000
    ln_eps: float = 1e-5
    parallel_residual: bool = True
    scale_attn_weights: bool = True
    scale_by_inverse_layer_idx: bool = False

    @property
    def head_dim(self):
        return self.hidden // self.n_head

    @property
    def rot_dims(self):
        return int(self.head_dim * self.rotary_pct)


def config_from_state_dict(sd, prefix, family, n_head, **kw):
    """Infer dims from the reference's own key names."""
    if family == A.NEOX:
        emb = sd[prefix + "gpt_neox.embed_in.weight"]
        n_layer = 1 + max(int(k[len(prefix):].split(".")[2]) for k in sd if k.startswith(prefix + "gpt_neox.layers."))
        max_pos = sd[prefix + "gpt_neox.layers.0.attention.self_bias"].shape[-1]
    else:
        emb = sd[prefix + "transformer.wte.weight"]
        n_layer = 1 + max(int(k[len(prefix):].split(".")[2]) for k in sd if k.startswith(prefix + "transformer.h."))
        max_pos = sd[prefix + "transformer.wpe.weight"].shape[0]
    return LMConfig(family, n_layer, n_head, emb.shape[1], emb.shape[0], max_pos, **kw)


def gelu_new(x):
    """transformers NewGELUActivation (GPT2MLP's `gelu_new`)."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def _split_heads(x, n_head):
    b, t, h = x.shape
    return x.view(b, t, n_head, h // n_head).permute(0, 2, 1, 3)


def _merge_heads(x):
    b, h, t, d = x.shape
    return x.permute(0, 2, 1, 3).reshape(b, t, h * d)


def _dropped(x, entry):
    """F.dropout with the Bernoulli draw handed in: entry = (keep, p) -> x * keep / (1 - p); None -> x."""
    if entry is None:
        return x
    keep, p = entry
    return x * keep.to(x.dtype) * (1.0 / (1.0 - p))


def lm_forward(sd, prefix, cfg, input_ids, position_ids=None, textual_past=None, mask=None, lm_logits=True, drop=None):
    """One forward of the LM.  textual_past: per layer (k_ctx, v_ctx) => branch ("non-residual") mode
    (NeoXNonResidualBase.py:173-177, NonResidualDiscriminatorBase.py:181-185); None => root causal pass.
    Returns (logits, presents, last_hidden) with presents = per layer (k, v) of the NEW tokens only.
    drop (GPT-2, root pass): the reference's dropouts with their draws handed in - {"embed": (keep, p), ("attn", i): (keep, p)
    on the attention probabilities (NonResidualAttention.py:107-109), ("resid", i) on the attention block's output (:274-275),
    ("mlp", i) GPT2MLP's nn.Dropout}; torch's generator cannot be matched, so parity tests give both sides the same draw."""
    drop = drop or {}
    B, T = input_ids.shape
    branch = textual_past is not None
    if position_ids is None:
        position_ids = torch.arange(T).unsqueeze(0).expand(B, T)
    position_ids = position_ids.long()
    g = lambda name: sd[prefix + name]
    presents = []
    if cfg.family == A.NEOX:
        x = F.embedding(input_ids, g("gpt_neox.embed_in.weight"))
        cos, sin = A.rotary_tables(cfg.rot_dims, cfg.max_pos, cfg.rotary_base)
        for i in range(cfg.n_layer):
            L = f"gpt_neox.layers.{i}."
            h1 = F.layer_norm(x, (cfg.hidden,), g(L + "input_layernorm.weight"), g(L + "input_layernorm.bias"), cfg.ln_eps)
            qkv = F.linear(h1, g(L + "attention.query_key_value.weight"), g(L + "attention.query_key_value.bias"))
            qkv = qkv.view(B, T, cfg.n_head, 3 * cfg.head_dim)                       # NeoXNonResidualAttention.py:94-100
            q, k, v = [qkv[..., j * cfg.head_dim:(j + 1) * cfg.head_dim].permute(0, 2, 1, 3) for j in range(3)]
            q = A.apply_rotary(q, cos, sin, position_ids, cfg.rot_dims)
            k = A.apply_rotary(k, cos, sin, position_ids, cfg.rot_dims)
            presents.append((k, v))
            if branch:
                o, _ = A.branch_attention(A.NEOX, q, k, v, textual_past[i][0], textual_past[i][1], mask)
            else:
                o, _ = A.causal_attention(A.NEOX, q, k, v)
            attn = F.linear(_merge_heads(o), g(L + "attention.dense.weight"), g(L + "attention.dense.bias"))
            mlp_in = x if cfg.parallel_residual else attn + x
            h2 = F.layer_norm(mlp_in, (cfg.hidden,), g(L + "post_attention_layernorm.weight"),
                              g(L + "post_attention_layernorm.bias"), cfg.ln_eps)
            m = F.linear(F.gelu(F.linear(h2, g(L + "mlp.dense_h_to_4h.weight"), g(L + "mlp.dense_h_to_4h.bias"))),
                         g(L + "mlp.dense_4h_to_h.weight"), g(L + "mlp.dense_4h_to_h.bias"))
            x = m + attn + x if cfg.parallel_residual else m + mlp_in            # NeoXNonResidualBase.py:113-126
        x = F.layer_norm(x, (cfg.hidden,), g("gpt_neox.final_layer_norm.weight"), g("gpt_neox.final_layer_norm.bias"), cfg.ln_eps)
        head = g("embed_out.weight") if lm_logits else g("valueHead.weight")
    else:
        x = F.embedding(input_ids, g("transformer.wte.weight")) + F.embedding(position_ids, g("transformer.wpe.weight"))
        x = _dropped(x, drop.get("embed"))                                         # NonResidualDiscriminatorBase.py:231-232
        for i in range(cfg.n_layer):
            L = f"transformer.h.{i}."
            h1 = F.layer_norm(x, (cfg.hidden,), g(L + "ln_1.weight"), g(L + "ln_1.bias"), cfg.ln_eps)
            qkv = h1 @ g(L + "attn.c_attn.weight") + g(L + "attn.c_attn.bias")     # Conv1D: weight is (in, out)
            q, k, v = [_split_heads(t, cfg.n_head) for t in qkv.split(cfg.hidden, dim=2)]
            presents.append((k, v))
            kw = dict(layer_idx=i, scale_attn_weights=cfg.scale_attn_weights,
                      scale_by_inverse_layer_idx=cfg.scale_by_inverse_layer_idx)
            if branch:
                o, _ = A.branch_attention(A.GPT2, q, k, v, textual_past[i][0], textual_past[i][1], mask, **kw)
            else:
                ad = drop.get(("attn", i))
                o, _ = A.causal_attention(A.GPT2, q, k, v, keep=ad[0] if ad else None, p_drop=ad[1] if ad else 0.0, **kw)
            x = x + _dropped(_merge_heads(o) @ g(L + "attn.c_proj.weight") + g(L + "attn.c_proj.bias"), drop.get(("resid", i)))
            h2 = F.layer_norm(x, (cfg.hidden,), g(L + "ln_2.weight"), g(L + "ln_2.bias"), cfg.ln_eps)
            m = gelu_new(h2 @ g(L + "mlp.c_fc.weight") + g(L + "mlp.c_fc.bias")) @ g(L + "mlp.c_proj.weight") + g(L + "mlp.c_proj.bias")
            x = x + _dropped(m, drop.get(("mlp", i)))
        x = F.layer_norm(x, (cfg.hidden,), g("transformer.ln_f.weight"), g("transformer.ln_f.bias"), cfg.ln_eps)
        head = g("lm_head.weight") if lm_logits else g("valueHead.weight")
    return F.linear(x, head), presents, x


def generator_forward(sd, cfg, input_ids, **kw):
    """NonResidualGenerator.forward (Generator.py:43-49): LM logits + softmax over the vocabulary."""
    logits, presents, _ = lm_forward(sd, "generator.generator.", cfg, input_ids, lm_logits=True, **kw)
    return logits, torch.softmax(logits, dim=-1), presents


def discriminator_forward(sd, cfg, input_ids, **kw):
    """doTransformerPassWithPrediction + doFinalizeSinglePrediction (Discriminator.py:87-110):
    sigmoid of the 1-unit valueHead, and the two extra value heads on the last hidden state."""
    logits, presents, hidden = lm_forward(sd, "discriminator.nonResidualModel.", cfg, input_ids, lm_logits=False, **kw)
    pred = torch.sigmoid(logits.squeeze(-1))
    in_v = torch.sigmoid(F.linear(hidden, sd["discriminator.inputValueHead.weight"], sd["discriminator.inputValueHead.bias"]))
    alt_v = torch.sigmoid(F.linear(hidden, sd["discriminator.alterationsValueHead.weight"],
                                   sd["discriminator.alterationsValueHead.bias"]))
    return pred, in_v, alt_v, presents


def discriminator_forward_legacy(sd, cfg, text_ids, alteration_ids):
    """The single-step `discForward` of the dual-head discriminator (Discriminator.py:28-79,112-127): the text pass, then
    every alteration sequence (B, n, S), stacked to (B*n, S), against its sample's text keys / values repeated n times, under
    the default branch mask (customCasualMask None).  -> (text preds (B,S), alteration preds (B,n,S), dict of the four value
    head tensors)."""
    txt_pred, txt_in, txt_alt, presents = discriminator_forward(sd, cfg, text_ids)
    B, n, S = alteration_ids.shape
    past = [(torch.repeat_interleave(k, n, dim=0), torch.repeat_interleave(v, n, dim=0)) for k, v in presents]
    alt_pred, alt_in, alt_alt, _ = discriminator_forward(sd, cfg, alteration_ids.reshape(B * n, S), textual_past=past)
    return txt_pred, alt_pred.reshape(B, n, S), {"txtInputPreds": txt_in, "txtAlterationsPreds": txt_alt,
                                                 "altInputPreds": alt_in.reshape(B, n, S), "altAlterationsPreds": alt_alt.reshape(B, n, S)}
