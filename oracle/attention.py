"""Oracle: branch ("non-residual") attention and root causal attention (test infrastructure).

Restates Models/NeoXBranchAttention/NeoXNonResidualAttention.py and
Models/GPT2BranchAttention/NonResidualAttention.py (reference), plain fp32 torch ops on the CPU.
"""
import math

import torch

NEOX, GPT2 = "neox", "gpt2"


def default_branch_mask(q_len, k_len):
    """getCausualAttentionMask(..., nonResidual=True) (NeoX :175-181, GPT-2 :70-76): query row i sees
    text keys j <= i + (k_len - q_len) - 1, plus the self column.  -> bool (1, 1, q_len, k_len + 1)."""
    rows = torch.arange(k_len - q_len, k_len).unsqueeze(1)
    cols = torch.arange(1, k_len + 1).unsqueeze(0)
    text = cols <= rows
    return torch.cat((text, torch.ones(q_len, 1, dtype=torch.bool)), dim=1)[None, None]


def _dropout(w, keep, p_drop):
    """F.dropout(w, p) with the Bernoulli draw handed in: torch computes `w * keep / (1 - p)` (GPT-2 :107-109,157-159 always
    on; NeoX :224,287 nn.Dropout).  keep None: the identity (p = 0)."""
    if keep is None:
        return w
    return w * keep.to(w.dtype) * (1.0 / (1.0 - p_drop))


def branch_attention(family, q, k_self, v_self, k_ctx, v_ctx, mask=None, layer_idx=0,
                     scale_attn_weights=True, scale_by_inverse_layer_idx=False, keep=None, p_drop=0.0):
    """`_nonResidualTextAttention` (NeoX :183-232) / `nonResidualTextAttention` (GPT-2 :119-172).

    q, k_self, v_self (B,H,K,Dh); k_ctx, v_ctx (B,H,C,Dh); mask bool broadcastable to (B,1,K,C+1) whose
    last column is the self column.  Returns (out (B,H,K,Dh), weights (B,H,K,C+1)).  Dropout over the
    weights: the identity unless the draw is handed in (keep, bool like the weights, with its rate p_drop):
    torch's generator cannot be matched, so parity tests give both sides the same draw."""
    dh = q.shape[-1]
    scores = torch.cat((q @ k_ctx.transpose(-1, -2), (q * k_self).sum(-1, keepdim=True)), dim=-1)
    if family == NEOX:
        scores = scores * (1 / torch.sqrt(torch.tensor(dh, dtype=torch.float32)))
        fill = torch.finfo(scores.dtype).min
    else:
        if scale_attn_weights:
            scores = scores / (dh ** 0.5)
        if scale_by_inverse_layer_idx:
            scores = scores / float(layer_idx + 1)
        fill = -1e4
    if mask is None:
        mask = default_branch_mask(q.shape[-2], k_ctx.shape[-2])
    scores = torch.where(mask, scores, torch.tensor(fill, dtype=scores.dtype))
    w = _dropout(torch.softmax(scores, dim=-1), keep, p_drop)
    out = w[..., :-1] @ v_ctx + w[..., -1:] * v_self
    return out, w


def causal_attention(family, q, k, v, layer_idx=0, scale_attn_weights=True, scale_by_inverse_layer_idx=False, keep=None, p_drop=0.0):
    """Root pass: NeoX `_attn` :234-290 with `_maskFunc` :292-301 (scores*m + (-1e9)(1-m)),
    GPT-2 `_attn` :78-117 (where(mask, scores, -1e4)).  q (B,H,T,Dh), k/v (B,H,S,Dh), T <= S."""
    t, s, dh = q.shape[-2], k.shape[-2], q.shape[-1]
    causal = torch.tril(torch.ones(s, s, dtype=torch.bool))[s - t:s, :s]
    if family == NEOX:
        b, h = q.shape[:2]
        alpha = float(torch.tensor(1.0) / torch.sqrt(torch.tensor(dh, dtype=torch.float32)))
        scores = torch.baddbmm(torch.zeros(b * h, t, s), q.reshape(b * h, t, dh), k.reshape(b * h, s, dh).transpose(1, 2),
                               beta=1.0, alpha=alpha).view(b, h, t, s)
        m = causal.float()
        scores = scores * m + torch.tensor(-1e9) * (1 - m)
    else:
        scores = q @ k.transpose(-1, -2)
        if scale_attn_weights:
            scores = scores / (dh ** 0.5)
        if scale_by_inverse_layer_idx:
            scores = scores / float(layer_idx + 1)
        scores = torch.where(causal, scores, torch.tensor(-1e4))
    w = _dropout(torch.softmax(scores, dim=-1), keep, p_drop)
    return w @ v, w


def rotary_tables(rot_dims, max_pos, base=10000):
    """GPTNeoXRotaryEmbedding :343-370 -> cos, sin (max_pos, rot_dims)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, rot_dims, 2).float() / rot_dims))
    freqs = torch.einsum("i,j->ij", torch.arange(max_pos, dtype=inv_freq.dtype), inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def apply_rotary(x, cos, sin, position_ids, rot_dims):
    """apply_rotary_pos_emb :425-432 on the first rot_dims of x (B,H,T,Dh); position_ids (B,T)."""
    c = cos[position_ids.long()][:, None]  # (B,1,T,rot)
    s = sin[position_ids.long()][:, None]
    xr, xp = x[..., :rot_dims], x[..., rot_dims:]
    half = rot_dims // 2
    rotated = torch.cat((-xr[..., half:], xr[..., :half]), dim=-1)
    return torch.cat((xr * c + rotated * s, xp), dim=-1)


# ---- arithmetic form of the mask (what the HIP kernel evaluates) ---------------------------------
def tree_mask_from_descriptor(ctx_len, n_anc, S, K):
    """Mask (1,1,K,S + n_anc*K + 1) for a node with n_anc branch ancestors: row k sees text j < ctx_len[k],
    its own column k of every ancestor block, and the self column.  Equals the mask
    PredictionTreeNode.getAttentionMask builds (MultiStepPrediction.py:92-127) — checked in tests."""
    ctx_len = torch.as_tensor(ctx_len)
    text = torch.arange(S)[None, :] < ctx_len[:, None]
    anc = torch.eye(K, dtype=torch.bool).repeat(1, n_anc)
    return torch.cat((text, anc, torch.ones(K, 1, dtype=torch.bool)), dim=1)[None, None]


def visible_pairs(ctx_len, n_anc):
    """Number of (query, key) pairs one (b, h) evaluates for a node: Σ_k (ctx_len_k + n_anc + 1)."""
    return int(sum(int(c) + n_anc + 1 for c in ctx_len))


def attention_flops(B, H, Dh, ctx_len, n_anc):
    """forward FLOPs = 4·Dh·H·B·Σ_k (ctx_len_k + d)."""
    return 4 * Dh * H * B * visible_pairs(ctx_len, n_anc)


__all__ = [n for n in dir() if not n.startswith("_") and n not in ("math", "torch")]
