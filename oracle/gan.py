"""Oracle: one Branch-GAN step's forward (both loss dicts), restating Models/GanSetup.py:106-258.

Test infrastructure only (see oracle/__init__.py).
"""
import torch

from . import losses as L
from . import tree as T


def pad_last(x, S, value=0.0, dim=-1):
    """Pads dim up to S (alteration scores/heads with 0: GanSetup.py:151-167,202-207; probs with 1e-5:
    MultiStepPrediction.py:59-65)."""
    dim = dim % x.dim()
    if x.shape[dim] >= S:
        return x
    shape = list(x.shape)
    shape[dim] = S - x.shape[dim]
    return torch.cat((x, torch.full(shape, value, dtype=x.dtype)), dim=dim)


def depth_mask_for(nodes, sparse_index, S):
    """createDiscAlterationLossWeights (GanSetup.py:75-82) with getPaddedPosIDs (MultiStepPrediction.py:53-57):
    1 where sparseIndex + depth <= S-1 on the K real columns, 0 on the 9999-padded tail."""
    sparse_index = torch.as_tensor(sparse_index)
    rows = []
    for n in nodes:
        pos = (sparse_index + n.depth).float()
        if len(sparse_index) != S:
            pos = torch.cat((pos, torch.full((S - len(sparse_index),), 9999.0)))
        rows.append(pos)
    return torch.less_equal(torch.stack(rows), S - 1).float()


def train_forward(sd, cfg_g, cfg_d, input_ids, rules, sparse_index, noise_fn=None, n_update=9999,
                  rl=None, mle_weight=1.0, rl_weight=1.0, pred_weight=1.0, value_head_weight=0.25,
                  separate_input_alt=True, input_loss_weight=1.0, forced_ids=None):
    """_internalTrainForwardPass (GanSetup.py:193-258) -> (discLosses, genLosses, extras).
    forced_ids: teacher-forced branch tokens (oracle.tree._substitute); extras["sampling_trace"] then holds, per sampling
    call, the ids this fp32 model would have sampled and the logit margin to the forced ones."""
    rl = dict(rl or {})
    B, S = input_ids.shape
    trace = [] if forced_ids is not None else None
    root, logits, probs, _ = T.multi_step_gen_forward(sd, cfg_g, input_ids, rules, sparse_index, noise_fn, forced_ids=forced_ids, trace=trace)
    nodes = root.sub_nodes()
    depth_mask = depth_mask_for(nodes, sparse_index, S)
    alt_probs = torch.stack([pad_last(n.probs, S, 1e-5) for n in nodes], dim=1)
    T.multi_step_disc_forward(sd, cfg_d, root, sparse_index)
    alt_scores = torch.stack([n.disc_pred for n in nodes], dim=1)
    if S > alt_scores.shape[1]:                                   # quirk 9: the reference compares S with N
        alt_scores = pad_last(alt_scores, S)
    alt_alt = torch.stack([n.alt_value for n in nodes], dim=0)
    if S > alt_alt.shape[2]:
        alt_alt = pad_last(alt_alt, S, dim=2)
    heads = {"txtInputPreds": root.in_value, "txtAlterationsPreds": root.alt_value, "altAlterationsPreds": alt_alt}
    links = T.sub_node_links(root, nodes)
    d = L.discriminator_loss(root.disc_pred, alt_scores, n_update, depth_mask, heads, links)
    d = L.combine_disc_losses(d, separate_input_alt, input_loss_weight, pred_weight, value_head_weight)
    g = L.generator_loss(input_ids, logits, probs, root.disc_pred, alt_scores, alt_probs, heads, depth_mask,
                         mle_weight=mle_weight, rl_weight=rl_weight, **rl)
    return d, g, dict(root=root, nodes=nodes, depth_mask=depth_mask, alt_scores=alt_scores, alt_probs=alt_probs,
                      logits=logits, probs=probs, sampling_trace=trace)
