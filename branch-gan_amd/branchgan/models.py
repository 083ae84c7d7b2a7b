"""The two LM families with branch ("non-residual") attention, generator and discriminator wrappers.

Host-side mirror of Models/NeoXBranchAttention/*, Models/GPT2BranchAttention/*, Models/Generator.py and
Models/Discriminator.py (reference): same class / method names, argument meaning and state-dict key names
(SURVEY.md §8f-1), own implementation.  Dense GEMMs (QKV, projections, MLP, LM head) go through torch
(hipBLASLt); attention (root causal AND branch), the NeoX QKV-split + rotary prologue and everything over the
vocabulary run in the hand-written HIP kernels of libbranchgan_hip.so (branchgan.ops).  No CPU path.

Branch mode is selected, like the reference, by passing `textualPast`.  Two forms are accepted:
  * engine form: `textualPast` is a BranchPast (text KV + ancestor KV slots + arithmetic mask descriptor:
    ctx_len and per-node slot paths) — no mask tensor, no KV concatenation;
  * reference form: `textualPast` = per layer (k_ctx, v_ctx) concatenated as getFullPast builds them, plus
    `customCasualMask` = the bool mask of PredictionTreeNode.getAttentionMask (validated to be that pattern and
    converted to the descriptor; anything else raises NotImplementedError) or None (legacy default mask).
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F
from torch import nn

from . import fused, ops
from .configs import LMConfig

_IGNORED_BUFFERS = (".attention.bias", ".attention.self_bias", ".attention.masked_bias", ".attn.bias", ".attn.self_bias",
                    ".attn.masked_bias")


class BranchPast:
    """KV context + arithmetic mask of one branch forward over NQ query nodes x K branch points."""

    def __init__(self, text, ancestors, ctx_len, paths, n_query_nodes=1, merged_text=0, node_uid0=0):
        self.node_uid0 = node_uid0          # number of the first query node among the tree's forwarded nodes (dropout sites: a
        #                                     branch step over node i and the tree pass over all nodes draw the same masks for it)
        self.merged_text = merged_text      # > 0: the first merged_text input tokens ARE the text (one pass over text + tree)
        self.text = text                    # per layer (k_text, v_text)  (B,H,S,Dh); None in merged form
        self.ancestors = ancestors          # per layer list of (k, v) slots (B,H,K,Dh), possibly empty
        self.ctx_len = ctx_len              # int32 (K,) on the device: sparseIndex + 1
        self.paths = paths                  # per query node: slot indices; i < n_anc -> ancestors[i], else own node i - n_anc
        self.n_query_nodes = n_query_nodes

    def layer(self, i):
        return SimpleNamespace(text=self.text[i] if self.text is not None else None, ancestors=self.ancestors[i],
                               ctx_len=self.ctx_len, paths=self.paths, n_query_nodes=self.n_query_nodes, merged_text=self.merged_text)


def descriptor_from_bool_mask(mask, K, C):
    """Reference mask (B,1,K,C+1) -> (ctx_len (K,) int32, n_anc).  Accepts exactly the T1 pattern
    [text prefix | n_anc identity blocks | self column] (MultiStepPrediction.py:92-127), identical over the batch."""
    if mask.dtype != torch.bool or mask.dim() != 4 or mask.shape[-2] != K or mask.shape[-1] != C + 1:
        raise NotImplementedError("customCasualMask must be the bool (B,1,K,C+1) mask of PredictionTreeNode.getAttentionMask")
    m = mask[0, 0].cpu()
    if not bool((mask == mask[:1]).all()):
        raise NotImplementedError("customCasualMask differs across the batch: not a branch-tree mask")
    eye = torch.eye(K, dtype=torch.bool)
    for n_anc in range(C // K, -1, -1):
        S = C - n_anc * K
        if S < 0:
            continue
        text, rest = m[:, :S], m[:, S:C]
        ctx = text.sum(1)
        prefix = torch.arange(S).unsqueeze(0) < ctx.unsqueeze(1)
        if bool((text == prefix).all()) and bool((rest == eye.repeat(1, n_anc)).all()) and bool(m[:, -1].all()):
            return ctx.to(torch.int32), n_anc
    raise NotImplementedError("customCasualMask is not a branch-tree mask (text prefix | per-depth diagonals | self)")


def _legacy_layer_past(layer_past, mask, K, device):
    """(k_ctx, v_ctx) [+ bool mask] of the reference -> the layer view the attention modules consume."""
    k_ctx, v_ctx = layer_past
    C = k_ctx.shape[-2]
    if mask is None:        # legacy single-step default mask (A3): row i sees text j <= i + (C - K) - 1
        ctx, n_anc = torch.arange(C - K, C, dtype=torch.int32), 0
    else:
        ctx, n_anc = descriptor_from_bool_mask(mask, K, C)
    S = C - n_anc * K
    anc = [(k_ctx[:, :, S + j * K: S + (j + 1) * K], v_ctx[:, :, S + j * K: S + (j + 1) * K]) for j in range(n_anc)]
    return SimpleNamespace(text=(k_ctx[:, :, :S], v_ctx[:, :, :S]), ancestors=anc, ctx_len=ctx.to(device),
                           paths=[list(range(n_anc + 1))], n_query_nodes=1)


def branch_attention_core(q, k, v, lp, scale, return_weights=False, p_drop=0.0):
    """q, k, v (B,H,NQ*K,Dh) of the new tokens; lp = layer view of a BranchPast.  -> (B,H,NQ*K,Dh), and with return_weights
    the softmax weights the reference materialises ((B,H,K,C+1) for one query node, (B,H,NQ,K,W) for several), rebuilt by
    bg_branch_attn_weights from the forward's log-sum-exp.  p_drop: dropout on the attention probabilities, inside the
    kernel (the weights handed out are then the ones after dropout, as in the reference)."""
    NQ = lp.n_query_nodes
    K = q.shape[2] // NQ
    q5, k5, v5 = q.unflatten(2, (NQ, K)), k.unflatten(2, (NQ, K)), v.unflatten(2, (NQ, K))
    anc_k = [a[0] for a in lp.ancestors]
    drop = ops.drop_descriptor(p_drop)
    out, lse = ops.branch_attention(q5, lp.text[0], lp.text[1], lp.ctx_len, anc_k, [a[1] for a in lp.ancestors],
                                    paths=lp.paths, scale=scale, own_k=k5, own_v=v5, drop=drop)
    if not return_weights:
        return out.flatten(2, 3)
    w = ops.branch_attention_weights(q5, lp.text[0], lp.ctx_len, anc_k, lp.paths, scale, lse, own_k=k5, drop=drop).to(v.dtype)
    return out.flatten(2, 3), (w[:, :, 0] if NQ == 1 else w)


def causal_attention_core(q, k, v, scale, return_weights=False, p_drop=0.0):
    """Root causal pass (+ its (B,H,T,S) weights on request)."""
    drop = ops.drop_descriptor(p_drop)
    out, lse = ops.causal_attention(q, k, v, scale=scale, drop=drop)
    if not return_weights:
        return out, None
    ctx = ops.causal_ctx_len(q.shape[2], k.shape[2], q.device)
    return out, ops.branch_attention_weights(q, k, ctx, [], [[]], scale, lse, drop=drop).to(v.dtype)


def apply_head_mask(out, weights, head_mask, n_head):
    """`attn_weights = attn_weights * head_mask` (NeoXNonResidualAttention.py:221-222, NonResidualAttention.py:161-162) for the
    per-head masks HuggingFace's get_head_mask produces ((H,), (1, H, 1, 1): one factor per head): scaling a head's weights
    scales that head's output, so the fused kernel's output (B, H, T, Dh) - and the weights, when they were asked for - are
    multiplied instead.  Masks that vary inside a head would need the weights themselves: not supported."""
    if head_mask is None:
        return out, weights
    if head_mask.numel() not in (1, n_head):
        raise NotImplementedError("head_mask must hold one factor per head ((H,) or (1, H, 1, 1))")
    hm = head_mask.reshape(1, -1, 1, 1)
    return out * hm.to(out.dtype), (weights * hm.to(weights.dtype) if weights is not None else None)


def dropout_active(model, dropoutRate=None):
    """Does a forward of `model` (an LM of either family, or a wrapper holding one) draw dropout masks?  GPT-2: the embedding,
    attention-probability and attention-output dropouts are always on (F.dropout without a training flag, quirk 1) at
    dropoutRate or the config's rates; NeoX: nn.Dropout modules, i.e. in training mode only, dropoutRate ignored."""
    lm = next((m for m in model.modules() if isinstance(m, _LMBase) and hasattr(m, "_fused_family")), None)
    if lm is None:
        return False
    cfg = lm.config
    if lm._fused_family == "gpt2":
        return bool(dropoutRate or cfg.attn_pdrop or cfg.embd_pdrop or cfg.resid_pdrop)
    return bool(lm.training and (cfg.attn_pdrop or cfg.resid_pdrop))


def dropout_by_site(model):
    """Are the dropout masks of `model`'s forwards functions of (step, site, logical row) alone - the same whether a tree node
    is forwarded alone (a branch step) or with every other node (the tree pass)?  True for the GPT-2 family on the fused path
    (ops.drop_site / ops.dropout_rows); the module-by-module path and the NeoX blocks draw theirs in call order / from torch's
    generator."""
    lm = next((m for m in model.modules() if isinstance(m, _LMBase) and hasattr(m, "_fused_family")), None)
    if lm is None or lm._fused_family != "gpt2" or not fused.ENABLED:
        return False
    p0 = next(lm.parameters())
    return p0.is_cuda and p0.dtype == torch.float32 and fused.supported_width(lm.config.hidden_size) and lm.config.head_dim in (32, 64, 128)


_rope_cache = {}


def rope_tables(rot, max_pos, base, device):
    """fp32 cos/sin tables (max_pos, rot) of GPTNeoXRotaryEmbedding (:343-364); kept out of the module's buffers so a
    bf16 cast of the model never rounds them."""
    key = (rot, max_pos, float(base), str(device))
    hit = _rope_cache.get(key)
    if hit is None:
        inv_freq = 1.0 / (base ** (torch.arange(0, rot, 2).float() / rot))
        emb = torch.cat([torch.einsum("i,j->ij", torch.arange(max_pos, dtype=inv_freq.dtype), inv_freq)] * 2, dim=-1)
        hit = (emb.cos().contiguous().to(device), emb.sin().contiguous().to(device))
        _rope_cache[key] = hit
    return hit


class Conv1D(nn.Module):
    """GPT-2's Conv1D: y = x @ W + b with W stored (in, out) — the reference state-dict layout."""

    def __init__(self, nf, nx):
        super().__init__()
        self.nf = nf
        self.weight = nn.Parameter(torch.empty(nx, nf))
        self.bias = nn.Parameter(torch.zeros(nf))

    def forward(self, x):
        return torch.addmm(self.bias, x.reshape(-1, x.size(-1)), self.weight).view(*x.shape[:-1], self.nf)


# =================================================================================================
# GPT-NeoX family
# =================================================================================================
class NeoXNonResidualAttention(nn.Module):
    """Models/NeoXBranchAttention/NeoXNonResidualAttention.py:7-148."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.num_attention_heads = config.num_attention_heads
        self.hidden_size = config.hidden_size
        self.head_size = config.head_dim
        self.rotary_ndims = config.rotary_ndims
        self.attention_dropout = config.attn_pdrop      # nn.Dropout(config.attention_dropout) in the reference (:31): training mode only
        inv_freq = 1.0 / (config.rotary_emb_base ** (torch.arange(0, self.rotary_ndims, 2).float() / self.rotary_ndims))
        self.rotary_emb = nn.Module()
        self.rotary_emb.register_buffer("inv_freq", inv_freq)
        self.norm_factor = math.sqrt(self.head_size)
        self.query_key_value = nn.Linear(config.hidden_size, 3 * config.hidden_size)
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)

    def _nonResidualTextAttention(self, query, key, value, textualKeyPast, textualValuePast, head_mask=None,
                                  customCasualMask=None, dropoutRate=None, output_attentions=False):
        """Reference signature (:183-184); dropoutRate is accepted and ignored there too.  The reference always returns the
        (B,H,K,C+1) weights; here they are formed only under output_attentions=True, else None."""
        lp = _legacy_layer_past((textualKeyPast, textualValuePast), customCasualMask, query.shape[-2], query.device)
        p = self._p_drop()
        if output_attentions:
            out, w = branch_attention_core(query, key, value, lp, 1.0 / self.norm_factor, return_weights=True, p_drop=p)
        else:
            out, w = branch_attention_core(query, key, value, lp, 1.0 / self.norm_factor, p_drop=p), None
        return apply_head_mask(out, w, head_mask, self.num_attention_heads)

    def _p_drop(self):
        return self.attention_dropout if self.training else 0.0

    def forward(self, hidden_states, attention_mask=None, position_ids=None, head_mask=None, layer_past=None,
                use_cache=False, output_attentions=False, isNonResidual=False, textualPast=None, customCasualMask=None):
        if layer_past is not None or attention_mask is not None:
            raise NotImplementedError("layer_past / attention_mask are not on the branch hot path")
        B, T, _ = hidden_states.shape
        qkv = self.query_key_value(hidden_states)
        cos, sin = rope_tables(self.rotary_ndims, self.config.max_position_embeddings, self.config.rotary_emb_base, qkv.device)
        query, key, value = ops.neox_qkv_rope(qkv, cos, sin, position_ids, self.num_attention_heads, self.head_size,
                                              self.rotary_ndims)
        present = (key, value) if use_cache else None
        scale = 1.0 / self.norm_factor
        weights = None
        if isNonResidual:
            lp = textualPast if hasattr(textualPast, "ctx_len") else _legacy_layer_past(textualPast, customCasualMask, T, qkv.device)
            attn_output = branch_attention_core(query, key, value, lp, scale, return_weights=output_attentions, p_drop=self._p_drop())
            if output_attentions:
                attn_output, weights = attn_output
        else:
            attn_output, weights = causal_attention_core(query, key, value, scale, return_weights=output_attentions, p_drop=self._p_drop())
        attn_output, weights = apply_head_mask(attn_output, weights, head_mask, self.num_attention_heads)
        attn_output = attn_output.permute(0, 2, 1, 3).reshape(B, T, self.hidden_size)
        attn_output = self.dense(attn_output)
        outputs = (attn_output, present)
        if output_attentions:
            outputs += (weights,)       # rebuilt from the forward's log-sum-exp on request; the hot path never forms them
        return outputs


def _neox_fused_attention(at, h1, pos, dtype, lp, dense_bias_grad=True):
    """NeoXNonResidualAttention.forward (:90-148) on the fused path: projections through fused.linear (cached bf16 weights,
    in-place fp32 dW / bias gradients), split + rotary in one kernel, branch / causal attention kernel."""
    B, T, _ = h1.shape
    qkv = fused.linear(h1, at.query_key_value.weight, at.query_key_value.bias, "oi", dtype)
    cos, sin = rope_tables(at.rotary_ndims, at.config.max_position_embeddings, at.config.rotary_emb_base, qkv.device)
    # the gradient of qkv goes straight into the stacked operands of the batched weight-gradient GEMMs
    # (fused._DeferredWeightGrads), like the LayerNorm / GELU outputs of the block
    query, key, value = ops.neox_qkv_rope(qkv, cos, sin, pos, at.num_attention_heads, at.head_size, at.rotary_ndims,
                                          grad_buffer=fused.dy_buffer_of(at.query_key_value.weight, qkv.device))
    scale = 1.0 / at.norm_factor
    if lp is not None:
        out = branch_attention_core(query, key, value, lp, scale, p_drop=at._p_drop())
    else:
        out, _ = ops.causal_attention(query, key, value, scale=scale, drop=ops.drop_descriptor(at._p_drop()))
    out = out.permute(0, 2, 1, 3).reshape(B, T, at.hidden_size)        # (into the x stack by copy_: autograd's CopySlices costs more than the stack copy)
    # dense_bias_grad=False: the LayerNorm backward that follows the residual add produces the bias gradient (add_layer_norm(delta_bias=))
    return fused.linear(out, at.dense.weight, at.dense.bias, "oi", dtype, bias_grad=dense_bias_grad), (key, value)


class GPTNeoXMLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense_h_to_4h = nn.Linear(config.hidden_size, config.intermediate_size)
        self.dense_4h_to_h = nn.Linear(config.intermediate_size, config.hidden_size)

    def forward(self, x):
        return self.dense_4h_to_h(F.gelu(self.dense_h_to_4h(x)))


class GPTNeoXLayer(nn.Module):
    """NeoXNonResidualBase.py:73-133."""

    def __init__(self, config):
        super().__init__()
        self.use_parallel_residual = config.use_parallel_residual
        self.input_layernorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.post_attention_layernorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.attention = NeoXNonResidualAttention(config)
        self.mlp = GPTNeoXMLP(config)
        self.hidden_dropout = config.resid_pdrop

    def forward(self, hidden_states, position_ids=None, use_cache=False, textualPast=None, isNonResidual=False,
                customCasualMask=None):
        attn_output, present = self.attention(self.input_layernorm(hidden_states), position_ids=position_ids, use_cache=use_cache,
                                              textualPast=textualPast, isNonResidual=isNonResidual,
                                              customCasualMask=customCasualMask)[:2]
        drop = (lambda x: F.dropout(x, self.hidden_dropout, self.training)) if self.hidden_dropout else (lambda x: x)
        attn_output = drop(attn_output)
        if self.use_parallel_residual:
            hidden_states = drop(self.mlp(self.post_attention_layernorm(hidden_states))) + attn_output + hidden_states
        else:
            attn_output = attn_output + hidden_states
            hidden_states = drop(self.mlp(self.post_attention_layernorm(attn_output))) + attn_output
        return hidden_states, present

    def fused_forward(self, x, delta, dtype, pos, lp, delta_from=None):
        """Same block on the fused path (see NonResidualGPT2Block.fused_forward; delta_from: the weight of the projection whose
        output gradient this block's LayerNorm backward produces - the previous block's attention.dense).  `x` is the fp32 residual stream BEFORE the
        previous block's update `delta` (None for the first block; a pair (attention, MLP) after a parallel-residual block,
        added in the LayerNorm kernel); returns (x + delta, this block's update, present).
        Parallel residual: both LayerNorms read x + delta, the update is attention + MLP."""
        qkv_w, up_w, down_w = self.attention.query_key_value.weight, self.mlp.dense_h_to_4h.weight, self.mlp.dense_4h_to_h.weight
        if self.use_parallel_residual:
            # both LayerNorms read the same x + delta: one kernel, one set of row statistics
            pair = (delta if isinstance(delta, tuple) else (delta, None))
            x1, h1, h2 = fused.add_layer_norm_pair(x, pair[0], self.input_layernorm, self.post_attention_layernorm, dtype, delta_b=pair[1],
                                                   out_for=(qkv_w, up_w), delta_from=delta_from)
            a, present = _neox_fused_attention(self.attention, h1, pos, dtype, lp)
            g = fused.linear(h2, up_w, self.mlp.dense_h_to_4h.bias, "oi", dtype, act="gelu", out_for=down_w)
            m = fused.linear(g, down_w, self.mlp.dense_4h_to_h.bias, "oi", dtype)
            return x1, (self._drop(a), self._drop(m)), present
        if delta is None:
            x1, h1 = x, fused.layer_norm(x, self.input_layernorm, dtype, out_for=qkv_w)
        elif isinstance(delta, tuple):
            x1, h1 = fused.add_layer_norm(x, delta[0], self.input_layernorm, dtype, delta_b=delta[1], out_for=qkv_w)
        else:
            x1, h1 = fused.add_layer_norm(x, delta, self.input_layernorm, dtype, out_for=qkv_w)
        a, present = _neox_fused_attention(self.attention, h1, pos, dtype, lp, dense_bias_grad=self.drops())
        if self.drops():      # a dropout between the projection and the residual add: the LayerNorm backward's shortcuts do not apply
            x2, h2 = fused.add_layer_norm(x1, self._drop(a), self.post_attention_layernorm, dtype, out_for=up_w)
        else:
            x2, h2 = fused.add_layer_norm(x1, a, self.post_attention_layernorm, dtype, self.attention.dense.bias, out_for=up_w,
                                          delta_from=self.attention.dense.weight)
        g = fused.linear(h2, up_w, self.mlp.dense_h_to_4h.bias, "oi", dtype, act="gelu", out_for=down_w)
        return x2, self._drop(fused.linear(g, self.mlp.dense_4h_to_h.weight, self.mlp.dense_4h_to_h.bias, "oi", dtype)), present

    def drops(self):
        """post_attention_dropout / post_mlp_dropout (NeoXNonResidualBase.py:79-80,110-125) are active."""
        return bool(self.hidden_dropout and self.training)

    def _drop(self, t):
        return F.dropout(t, self.hidden_dropout, True) if self.drops() else t


def strip_reference_buffers(state_dict):
    """Reference checkpoints carry mask buffers (bias / self_bias / masked_bias) and, when saved from DDP,
    a `module.` prefix (Models/FileManagement.py:5-7): neither has a counterpart here."""
    out = {}
    for k, v in state_dict.items():
        if k.endswith(_IGNORED_BUFFERS):
            continue
        out[k.replace("module.", "")] = v
    return out


def reference_buffers(module):
    """The persistent mask buffers a reference checkpoint of `module` would carry next to its parameters:
    per GPT-2 attention `bias` (causal lower triangle), `self_bias` (all ones) - uint8 (1, 1, P, P) - and `masked_bias`
    (-1e4); per NeoX attention `self_bias`.  This build evaluates the masks arithmetically and never stores them; they are
    regenerated only to write a checkpoint a strict reference-side `load_state_dict` accepts."""
    out = {}
    sd = module.state_dict()
    for k, v in sd.items():
        if k.endswith(".attn.c_attn.weight"):
            prefix, P = k[:-len("c_attn.weight")], sd[next(n for n in sd if n.endswith("wpe.weight"))].shape[0]
            out[prefix + "bias"] = torch.tril(torch.ones((P, P), dtype=torch.uint8)).view(1, 1, P, P)
            out[prefix + "self_bias"] = torch.ones((1, 1, P, P), dtype=torch.uint8)
            out[prefix + "masked_bias"] = torch.tensor(-1e4)
        elif k.endswith(".attention.query_key_value.weight"):
            cfg = next(m.config for m in module.modules() if isinstance(getattr(m, "config", None), LMConfig))
            P = cfg.max_position_embeddings
            out[k[:-len("query_key_value.weight")] + "self_bias"] = torch.ones((1, 1, P, P), dtype=torch.uint8)
    return out


class _LMBase(nn.Module):
    def _init_weights(self, std):
        for m in self.modules():
            if isinstance(m, (nn.Linear, Conv1D)):
                nn.init.normal_(m.weight, mean=0.0, std=std)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, mean=0.0, std=std)
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def generate(self, input_ids, **kwargs):
        """HF-style `model.generate(ids, max_new_tokens=..., do_sample=..., temperature=..., pad_token_id=...)` (what
        Inference/GenerateTexts.py:19 calls): branchgan.generate.generate."""
        from . import generate as _generate
        return _generate.generate(self, input_ids, **kwargs)

    def supportsMergedTree(self, input_ids):
        """Can text and tree nodes go through ONE forward (BranchPast.merged_text)?  Only on the fused block path."""
        return getattr(self, "_fused_family", None) == "gpt2" and _fused_eligible(self, input_ids, None, None, False)

    def load_state_dict(self, state_dict, strict=True, assign=False):
        """Accepts reference checkpoints: drops their mask buffers (bias / self_bias / masked_bias)."""
        return super().load_state_dict(strip_reference_buffers(state_dict), strict=strict, assign=assign)


class GPTNeoXModel(_LMBase):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embed_in = nn.Embedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([GPTNeoXLayer(config) for _ in range(config.num_hidden_layers)])
        self.final_layer_norm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


def _positions(position_ids, B, T, device, offset=0):
    if position_ids is None:
        return torch.arange(offset, offset + T, dtype=torch.int32, device=device).unsqueeze(0).expand(B, T)
    return position_ids.view(-1, T).to(torch.int32)


def _layer_past(textualPast, i):
    return textualPast.layer(i) if isinstance(textualPast, BranchPast) else textualPast[i]


def _fused_eligible(model, input_ids, textualPast, customCasualMask, output_hidden_states, dropoutRate=None):
    """The fused block path (branchgan.fused) covers the training hot path: fp32 parameters on the GPU (bf16 compute comes
    from autocast), engine-form past; dropout on the attention probabilities runs inside the attention kernels; the GPT-2
    family's residual / MLP dropouts inside the LayerNorm kernels and its embedding dropout through bg_dropout_rows (one
    generator, masks by site), the NeoX blocks' nn.Dropouts as ATen ops between the fused pieces.  Everything else runs module
    by module."""
    ok = _fused_covers(model, input_ids, textualPast, customCasualMask, output_hidden_states, dropoutRate)
    if not ok:
        fused.module_path_forwards += 1         # graphs.py: a step with module-by-module forwards is not replayed side by side
    return ok


def _fused_covers(model, input_ids, textualPast, customCasualMask, output_hidden_states, dropoutRate):
    cfg = model.config
    if not fused.ENABLED or not input_ids.is_cuda or output_hidden_states or customCasualMask is not None:
        return False
    if textualPast is not None and not isinstance(textualPast, BranchPast):
        return False
    if not hasattr(model, "_fused_family"):
        return False
    if (dropoutRate or cfg.attn_pdrop) and cfg.head_dim > 128:      # dropout inside the attention kernels: Dh <= 128
        return False
    if not fused.supported_width(cfg.hidden_size) or cfg.head_dim not in (32, 64, 128, 256):
        return False
    return next(model.parameters()).dtype == torch.float32


class GPTNeoXForCausalLM(_LMBase):
    """NeoXNonResidualBase.py:303-400: LM head `embed_out` or 1-unit `valueHead`, chosen by returnLmLogits."""
    _fused_family = "neox"

    def __init__(self, config):
        super().__init__()
        config = LMConfig.from_any(config)
        self.config = config
        self.gpt_neox = GPTNeoXModel(config)
        self.embed_out = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.valueHead = nn.Linear(config.hidden_size, 1, bias=False)
        self._init_weights(config.initializer_range)

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, inputs_embeds=None, head_mask=None,
                past_key_values=None, labels=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                return_dict=None, textualPast=None, returnLmLogits=False, dropoutRate=None, customCasualMask=None):
        if any(x is not None for x in (attention_mask, inputs_embeds, head_mask, past_key_values, labels)):
            raise NotImplementedError("attention_mask / inputs_embeds / head_mask / past_key_values / labels are off the hot path")
        B, T = input_ids.shape
        pos = _positions(position_ids, B, T, input_ids.device)
        branch = textualPast is not None
        if _fused_eligible(self, input_ids, textualPast, customCasualMask, output_hidden_states, dropoutRate) and \
                not (branch and getattr(textualPast, "merged_text", 0)):
            dtype = fused.compute_dtype(self.gpt_neox.embed_in.weight)
            fused.prepare_weights(self, dtype)
            pos = pos.contiguous()          # once, not per layer (an expanded view otherwise)
            with torch.autocast("cuda", enabled=False):
                x, delta, presents = fused.embedding(input_ids, None, self.gpt_neox.embed_in.weight), None, []
                if self.config.resid_pdrop and self.training:       # emb_dropout (NeoXNonResidualBase.py:142,239)
                    x = F.dropout(x, self.config.resid_pdrop, True)
                delta_from = None
                for i, layer in enumerate(self.gpt_neox.layers):
                    x, delta, present = layer.fused_forward(x, delta, dtype, pos, _layer_past(textualPast, i) if branch else None,
                                                            delta_from)
                    delta_from = layer.attention.dense.weight if layer.use_parallel_residual and not layer.drops() else None
                    presents.append(present)
                if isinstance(delta, tuple):
                    _, y = fused.add_layer_norm(x, delta[0], self.gpt_neox.final_layer_norm, dtype, delta_b=delta[1], delta_from=delta_from)
                else:
                    _, y = fused.add_layer_norm(x, delta, self.gpt_neox.final_layer_norm, dtype)
                head = self.embed_out if returnLmLogits else self.valueHead
                logits = fused.linear(y, head.weight, None, "oi", dtype)
            return SimpleNamespace(loss=None, logits=logits, past_key_values=tuple(presents) if use_cache else None,
                                   hidden_states=None, attentions=None, last_hidden_state=y)
        x = self.gpt_neox.embed_in(input_ids)
        hidden_dropout = self.config.resid_pdrop
        if hidden_dropout:
            x = F.dropout(x, hidden_dropout, self.training)
        presents, all_hidden = [], []
        for i, layer in enumerate(self.gpt_neox.layers):
            if output_hidden_states:
                all_hidden.append(x)
            x, present = layer(x, position_ids=pos, use_cache=bool(use_cache), isNonResidual=branch,
                               textualPast=_layer_past(textualPast, i) if branch else None, customCasualMask=customCasualMask)
            presents.append(present)
        x = self.gpt_neox.final_layer_norm(x)
        if output_hidden_states:
            all_hidden.append(x)
        logits = self.embed_out(x) if returnLmLogits else self.valueHead(x)
        return SimpleNamespace(loss=None, logits=logits, past_key_values=tuple(presents) if use_cache else None,
                               hidden_states=tuple(all_hidden) if output_hidden_states else None, attentions=None,
                               last_hidden_state=x)


# =================================================================================================
# GPT-2 family
# =================================================================================================
def gelu_new(x):
    """GPT2MLP's `gelu_new` (tanh form), as ONE fused kernel each way instead of ~8 + ~20 elementwise launches."""
    return F.gelu(x, approximate="tanh")


class GPT2NonResidualAttention(nn.Module):
    """Models/GPT2BranchAttention/NonResidualAttention.py:22-281."""

    def __init__(self, config, is_cross_attention=False, layer_idx=None):
        super().__init__()
        if is_cross_attention:
            raise NotImplementedError("cross attention is not on the branch hot path")
        self.config = config
        self.embed_dim = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = config.head_dim
        self.split_size = self.embed_dim
        self.scale_attn_weights = config.scale_attn_weights
        self.scale_attn_by_inverse_layer_idx = config.scale_attn_by_inverse_layer_idx
        self.layer_idx = layer_idx
        self.c_attn = Conv1D(3 * self.embed_dim, self.embed_dim)
        self.c_proj = Conv1D(self.embed_dim, self.embed_dim)

    def _scale(self):
        s = 1.0 / math.sqrt(self.head_dim) if self.scale_attn_weights else 1.0
        return s / float(self.layer_idx + 1) if self.scale_attn_by_inverse_layer_idx else s

    def _p_attn(self, dropoutRate):
        """Rate of the always-on dropout over the attention probabilities (:107-109,157-159; quirk 1: no training flag)."""
        return dropoutRate if dropoutRate is not None else self.config.attn_pdrop

    def _p_resid(self, dropoutRate):
        """Rate of the always-on dropout over the attention block's output (:274-275)."""
        return dropoutRate if dropoutRate is not None else self.config.resid_pdrop

    def nonResidualTextAttention(self, query, key, value, textualKeyPast, textualValuePast, head_mask=None,
                                 customCasualMask=None, dropoutRate=None, output_attentions=False):
        """Reference signature (:119-120); the weights only under output_attentions=True (else None)."""
        lp = _legacy_layer_past((textualKeyPast, textualValuePast), customCasualMask, query.shape[-2], query.device)
        p = self._p_attn(dropoutRate)
        if output_attentions:
            out, w = branch_attention_core(query, key, value, lp, self._scale(), return_weights=True, p_drop=p)
        else:
            out, w = branch_attention_core(query, key, value, lp, self._scale(), p_drop=p), None
        return apply_head_mask(out, w, head_mask, self.num_heads)

    def forward(self, hidden_states, layer_past=None, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, use_cache=False, output_attentions=False, nonResidual=False,
                nonResidualTextOnly=False, nonResdiaulPastOnly=False, promptPast=None, textualPast=None, promptMask=None,
                customCasualMask=None, dropoutRate=None):
        if nonResidual or nonResdiaulPastOnly or promptPast is not None or encoder_hidden_states is not None or \
                layer_past is not None or attention_mask is not None:
            raise NotImplementedError()      # the reference raises for `nonResidual` too (:265-266); prompt modes are out of scope
        p_attn = self._p_attn(dropoutRate)
        B, T, _ = hidden_states.shape
        qkv = self.c_attn(hidden_states).view(B, T, 3, self.num_heads, self.head_dim)
        query, key, value = [qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3)]     # strided views, no copies
        present = (key, value) if use_cache else None
        weights = None
        if nonResidualTextOnly:
            lp = textualPast if hasattr(textualPast, "ctx_len") else _legacy_layer_past(textualPast, customCasualMask, T, qkv.device)
            attn_output = branch_attention_core(query, key, value, lp, self._scale(), return_weights=output_attentions, p_drop=p_attn)
            if output_attentions:
                attn_output, weights = attn_output
        else:
            attn_output, weights = causal_attention_core(query, key, value, self._scale(), return_weights=output_attentions, p_drop=p_attn)
        attn_output, weights = apply_head_mask(attn_output, weights, head_mask, self.num_heads)
        attn_output = self.c_proj(attn_output.permute(0, 2, 1, 3).reshape(B, T, self.embed_dim))
        p = self._p_resid(dropoutRate)
        if p:
            attn_output = F.dropout(attn_output, p)          # always-on like the reference (:274-275, quirk 1)
        outputs = (attn_output, present)
        if output_attentions:
            outputs += (weights,)
        return outputs


class GPT2MLP(nn.Module):
    def __init__(self, inner, config):
        super().__init__()
        self.c_fc = Conv1D(inner, config.hidden_size)
        self.c_proj = Conv1D(config.hidden_size, inner)
        self.resid_pdrop = config.resid_pdrop

    def forward(self, x):
        x = self.c_proj(gelu_new(self.c_fc(x)))
        return F.dropout(x, self.resid_pdrop, self.training) if self.resid_pdrop else x


class NonResidualGPT2Block(nn.Module):
    def __init__(self, config, layer_idx=None):
        super().__init__()
        self.ln_1 = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.attn = GPT2NonResidualAttention(config, layer_idx=layer_idx)
        self.ln_2 = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.mlp = GPT2MLP(config.intermediate_size, config)

    def forward(self, hidden_states, use_cache=False, nonResidualTextOnly=False, textualPast=None, customCasualMask=None,
                dropoutRate=None):
        a, present = self.attn(self.ln_1(hidden_states), use_cache=use_cache, nonResidualTextOnly=nonResidualTextOnly,
                               textualPast=textualPast, customCasualMask=customCasualMask, dropoutRate=dropoutRate)[:2]
        hidden_states = a + hidden_states
        return hidden_states + self.mlp(self.ln_2(hidden_states)), present

    def fused_forward(self, x, delta, delta_bias, dtype, lp, delta_from=None, dropoutRate=None, drop_ctx=None, delta_drop=None):
        """Same block on the fused kernels (branchgan.fused).  The residual stream enters as x + delta (delta = the previous
        block's MLP output - of the projection with weight delta_from -, None for the first block) and leaves as (x2, m): the add is done by the next LayerNorm kernel.
        -> (x2, m, bias of m's projection, present, weight of m's projection, dropout m enters the residual stream through or
        None).  The dropouts in front of the two residual adds happen INSIDE the LayerNorm kernels that do the adds
        (fused.add_layer_norm(drop=)): no launch of their own, and the projections' bias gradients still come out of the
        LayerNorm backward.
        Dropouts as in the reference: attention probabilities (inside the attention kernels) and attention output always on
        at dropoutRate or the config's rates, the MLP's nn.Dropout in training mode only.  Their masks are functions of (step,
        site, logical row): drop_ctx = (model salt, text rows of this pass, logical number of its first tree row), ops.drop_site."""
        at = self.attn
        B, T, _ = x.shape
        salt, text_rows, row0 = drop_ctx if drop_ctx is not None else (0, T, 0)
        site = lambda kind: ops.drop_site(salt, at.layer_idx or 0, kind)
        # out_for / delta_from / qkv_from name the projection on the other side of a tensor: producers then write the stacked
        # operands of the batched weight-gradient GEMMs in place (fused._DeferredWeightGrads)
        if delta is None:
            x1, h1 = x, fused.layer_norm(x, self.ln_1, dtype, out_for=at.c_attn.weight)
        else:
            x1, h1 = fused.add_layer_norm(x, delta, self.ln_1, dtype, delta_bias, out_for=at.c_attn.weight, delta_from=delta_from,
                                          drop=delta_drop)
        qkv = fused.linear(h1, at.c_attn.weight, at.c_attn.bias, "io", dtype).view(B, T, 3, at.num_heads, at.head_dim)
        a, present = fused.packed_attention(qkv, lp, at._scale(), out_for=at.c_proj.weight, qkv_from=at.c_attn.weight,
                                            p_drop=at._p_attn(dropoutRate), drop_site=site(ops.SITE_ATTN), drop_row0=row0)
        # the two projections that feed a residual add leave their bias gradient to the LayerNorm backward that follows
        p_res = at._p_resid(dropoutRate)
        a = fused.linear(a, at.c_proj.weight, at.c_proj.bias, "io", dtype, bias_grad=False)
        x2, h2 = fused.add_layer_norm(x1, a, self.ln_2, dtype, at.c_proj.bias, out_for=self.mlp.c_fc.weight, delta_from=at.c_proj.weight,
                                      drop=(p_res, site(ops.SITE_RESID), row0, text_rows) if p_res else None)
        g = fused.linear(h2, self.mlp.c_fc.weight, self.mlp.c_fc.bias, "io", dtype, act="gelu_tanh", out_for=self.mlp.c_proj.weight)
        delta = fused.linear(g, self.mlp.c_proj.weight, self.mlp.c_proj.bias, "io", dtype, bias_grad=False)
        p_mlp = self.mlp.resid_pdrop if self.training else 0.0
        return x2, delta, self.mlp.c_proj.bias, present, self.mlp.c_proj.weight, \
            ((p_mlp, site(ops.SITE_MLP), row0, text_rows) if p_mlp else None)


class NonResidualGPT2Model(_LMBase):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.wte = nn.Embedding(config.vocab_size, config.hidden_size)
        self.wpe = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.h = nn.ModuleList([NonResidualGPT2Block(config, layer_idx=i) for i in range(config.num_hidden_layers)])
        self.ln_f = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class NonResidualGPT2LMHeadModel(_LMBase):
    """NonResidualDiscriminatorBase.py:383-511."""
    _fused_family = "gpt2"

    def __init__(self, config):
        super().__init__()
        config = LMConfig.from_any(config)
        self.config = config
        self.transformer = NonResidualGPT2Model(config)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.valueHead = nn.Linear(config.hidden_size, 1, bias=False)
        self._drop_salt = 0                 # tells the models of a step apart in their dropout sites (the wrappers set 1 / 2)
        self._init_weights(config.initializer_range)

    def forward(self, input_ids=None, past_key_values=None, attention_mask=None, token_type_ids=None, position_ids=None,
                head_mask=None, inputs_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None,
                customCasualMask=None, suffixOverrideEmbeddings=None, insertBlocks=None, labels=None, use_cache=None,
                output_attentions=None, output_hidden_states=None, return_dict=None, promptPast=None, textualPast=None,
                promptMask=None, returnLmLogits=False, dropoutRate=None):
        if any(x is not None for x in (past_key_values, attention_mask, token_type_ids, head_mask, inputs_embeds, encoder_hidden_states,
                                       encoder_attention_mask, suffixOverrideEmbeddings, insertBlocks, labels, promptPast, promptMask)):
            raise NotImplementedError("prompt-past / cross-attention / adapter modes are off the branch hot path")
        B, T = input_ids.shape
        pos = _positions(position_ids, B, T, input_ids.device)
        branch = textualPast is not None
        tr = self.transformer
        merged_text = getattr(textualPast, "merged_text", 0) if branch else 0
        if not branch or merged_text:
            ops.dropout_new_pass(input_ids.device)      # a forward over a text: fresh dropout sites for it and the tree passes that hang off it
        if _fused_eligible(self, input_ids, textualPast, customCasualMask, output_hidden_states, dropoutRate):
            dtype = fused.compute_dtype(tr.wte.weight)
            fused.prepare_weights(self, dtype)
            # logical rows of this pass for the dropout masks: text positions as they are, tree rows from their node's number
            text_rows = merged_text if branch else T
            row0 = ops.TREE_ROW_BASE + getattr(textualPast, "node_uid0", 0) * int(textualPast.ctx_len.numel()) if branch else 0
            drop_ctx = (self._drop_salt, text_rows, row0)
            with torch.autocast("cuda", enabled=False):
                x, delta, delta_bias, presents = fused.embedding(input_ids, pos, tr.wte.weight, tr.wpe.weight), None, None, []
                p_emb = dropoutRate if dropoutRate is not None else self.config.embd_pdrop
                if p_emb:                                           # always on (NonResidualDiscriminatorBase.py:231-232)
                    x = ops.dropout_rows(x, p_emb, ops.drop_site(self._drop_salt, 0x1FF, ops.SITE_EMBED), row0, text_rows)
                delta_from = delta_drop = None
                for i, block in enumerate(tr.h):
                    x, delta, delta_bias, present, delta_from, delta_drop = block.fused_forward(
                        x, delta, delta_bias, dtype, _layer_past(textualPast, i) if branch else None, delta_from, dropoutRate, drop_ctx, delta_drop)
                    presents.append(present)
                _, y = fused.add_layer_norm(x, delta, tr.ln_f, dtype, delta_bias, delta_from=delta_from, drop=delta_drop)
                head = self.lm_head if returnLmLogits else self.valueHead
                logits = fused.linear(y, head.weight, None, "oi", dtype)
            return SimpleNamespace(loss=None, logits=logits, past_key_values=tuple(presents) if use_cache else None,
                                   hidden_states=None, attentions=None, cross_attentions=None, last_hidden_state=y)
        x = tr.wte(input_ids) + tr.wpe(pos.long())
        p = dropoutRate if dropoutRate is not None else self.config.embd_pdrop
        if p:
            x = F.dropout(x, p)
        presents, all_hidden = [], []
        for i, block in enumerate(tr.h):
            if output_hidden_states:
                all_hidden.append(x)
            x, present = block(x, use_cache=bool(use_cache), nonResidualTextOnly=branch,
                               textualPast=_layer_past(textualPast, i) if branch else None, customCasualMask=customCasualMask,
                               dropoutRate=dropoutRate)
            presents.append(present)
        x = tr.ln_f(x)
        if output_hidden_states:
            all_hidden.append(x)
        logits = self.lm_head(x) if returnLmLogits else self.valueHead(x)
        return SimpleNamespace(loss=None, logits=logits, past_key_values=tuple(presents) if use_cache else None,
                               hidden_states=tuple(all_hidden) if output_hidden_states else None, attentions=None,
                               cross_attentions=None, last_hidden_state=x)


def make_lm(config):
    config = LMConfig.from_any(config)
    return GPTNeoXForCausalLM(config) if config.model_type == "gpt_neox" else NonResidualGPT2LMHeadModel(config)


# =================================================================================================
# Generator / Discriminator wrappers
# =================================================================================================
class LazyProbs:
    """softmax(logits) that is only formed if somebody asks (`.materialize()`); the hot path never does."""

    def __init__(self, logits):
        self.logits = logits
        self.shape = logits.shape

    def materialize(self):
        return torch.softmax(self.logits, dim=-1)


class GeneratorOutput:
    def __init__(self, prediction=None, classProbs=None, discPreds=None, inputValuePreds=None, alterationsValuePreds=None):
        self.prediction, self.classProbs, self.discPreds = prediction, classProbs, discPreds
        self.inputValuePreds, self.alterationsValuePreds = inputValuePreds, alterationsValuePreds


class NonResidualGenerator(nn.Module):
    """Models/Generator.py:36-49; classProbs is lazy (the kernels work from the logits)."""

    def __init__(self, generatorConfig):
        super().__init__()
        self.config = LMConfig.from_any(generatorConfig)
        self.generator = make_lm(self.config)
        self.generator._drop_salt = 1

    def forward(self, args):
        args = dict(args)
        args['output_hidden_states'] = args.get('output_hidden_states', False)
        prediction = self.generator(**args)
        return GeneratorOutput(prediction, LazyProbs(prediction.logits))


class DiscriminatorOutput:
    def __init__(self, prediction=None, classProbs=None, transformerPred=None):
        self.transformerPred, self.prediction, self.classProbs = transformerPred, prediction, classProbs


class NonResidualDiscriminatorWithDualValueHeads(nn.Module):
    """Models/Discriminator.py:15-127 (the tree path: doTransformerPassWithPrediction / doFinalizeSinglePrediction)."""

    def __init__(self, discriminatorConfig):
        super().__init__()
        self.discriminatorConfig = LMConfig.from_any(discriminatorConfig)
        self.nonResidualModel = make_lm(self.discriminatorConfig)
        self.nonResidualModel._drop_salt = 2
        self.inputValueHead = nn.Linear(self.discriminatorConfig.hidden_size, 1)
        self.alterationsValueHead = nn.Linear(self.discriminatorConfig.hidden_size, 1)
        self.hasValueHeads = True

    def _finalizeSinglePrediction(self, pred):
        hidden = pred.hidden_states[-1] if pred.hidden_states is not None else pred.last_hidden_state
        alterationsValuePreds = torch.sigmoid(self.alterationsValueHead(hidden))
        inputValuePreds = torch.sigmoid(self.inputValueHead(hidden))
        return torch.sigmoid(pred.logits.squeeze(-1)), inputValuePreds, alterationsValuePreds

    # ---- the legacy single-step path (Discriminator.py:28-79,112-127): text pass, then every alteration sequence of a sample
    # against that sample's text keys / values under the default branch mask (row i sees text j < i and itself).  The train
    # step uses the tree path; this is kept for the reference's call surface.
    @staticmethod
    def copyTextualPastToFitAlterations(textualPast, numAlterationsPerSample):
        if numAlterationsPerSample <= 1:
            return textualPast
        return [(torch.repeat_interleave(k, numAlterationsPerSample, dim=0), torch.repeat_interleave(v, numAlterationsPerSample, dim=0))
                for k, v in textualPast]

    def generateTextualPrediction(self, textualIDs):
        textualPrediction = self.nonResidualModel(input_ids=textualIDs, use_cache=True)
        return textualPrediction, textualPrediction.past_key_values

    def generateAlterationPrediction(self, textualPast, alternationIDs):
        stacked = alternationIDs.reshape(alternationIDs.shape[0] * alternationIDs.shape[1], alternationIDs.shape[2])
        past = self.copyTextualPastToFitAlterations(textualPast, alternationIDs.shape[1])
        return self.nonResidualModel(input_ids=stacked, textualPast=[(k, v) for k, v in past])

    def finalizeDiscForward(self, textualPrediction, nonResPrediction, alternationIDs):
        txtDiscPreds, txtInputValuePreds, txtAlterationsValuePreds = self._finalizeSinglePrediction(textualPrediction)
        _, altInputValuePreds, altAlterationsValuePreds = self._finalizeSinglePrediction(nonResPrediction)
        shape = alternationIDs.shape
        altDiscPreds = torch.sigmoid(nonResPrediction.logits.reshape(shape))
        return txtDiscPreds, altDiscPreds, {'txtInputPreds': txtInputValuePreds, 'txtAlterationsPreds': txtAlterationsValuePreds,
                                            'altInputPreds': altInputValuePreds.reshape(shape),
                                            'altAlterationsPreds': altAlterationsValuePreds.reshape(shape)}

    def discForward(self, textualIDs, alternationIDs):
        """textualIDs (B, S); alternationIDs (B, n, S) -> (text predictions (B, S), alteration predictions (B, n, S), value heads)."""
        textualPrediction, textualPast = self.generateTextualPrediction(textualIDs)
        nonResPrediction = self.generateAlterationPrediction(textualPast, alternationIDs)
        return self.finalizeDiscForward(textualPrediction, nonResPrediction, alternationIDs)

    def forward(self, inputArgs, doDiscForward=True, doTransformerPass=False, doFinalizeSinglePrediction=False,
                doTransformerPassWithPrediction=False):
        if doDiscForward:
            return self.discForward(**inputArgs)
        if doTransformerPass:
            return self.nonResidualModel(**inputArgs)
        if doTransformerPassWithPrediction:
            transPred = self.nonResidualModel(**inputArgs)
            return DiscriminatorOutput(torch.sigmoid(transPred.logits.squeeze(-1)), transformerPred=transPred)
        if doFinalizeSinglePrediction:
            return self._finalizeSinglePrediction(**inputArgs)


NonResidualDiscriminator = NonResidualDiscriminatorWithDualValueHeads
