"""Oracle: discriminator and generator losses (test infrastructure, see oracle/__init__.py).

Restates Models/Losses/{DiscriminatorLoss,GeneratorLosses,ReinforceLoss,AdvantageCalculations}.py.
`links` = (root_links, alteration_links) as getSubNodeListTreeStructure returns them.
"""
import torch
import torch.nn.functional as F

ALTERATIONS_ONLY, INPUT_ONLY, ALTERATIONS_AND_INPUT, ALTERATIONS_AND_INPUT_UNWEIGHTED = 0, 1, 2, 3
DO_NOTHING, IGNORE_SAMPLE = 0, 1

bce = F.binary_cross_entropy  # GanSetup.py:43-44 (log clamped at -100 inside ATen)


def value_head_loss(input_scores_shifted, alt_scores, depth_mask, heads, links):
    """_calcDiscriminatorValueHeadLoss (DiscriminatorLoss.py:4-34).  heads: dict with txtInputPreds (B,S,1),
    txtAlterationsPreds (B,S,1), altAlterationsPreds (N,B,S,1)."""
    root_links, alt_links = links
    in_loss = bce(heads["txtInputPreds"][:, :-1].squeeze(-1), input_scores_shifted.detach())
    alt_scores = alt_scores.detach()
    child_mean = lambda ids: torch.stack([alt_scores[:, i] for i in ids], dim=0).mean(dim=0)
    root_target = child_mean(root_links).to(heads["txtAlterationsPreds"].dtype)
    # quirk 10: text column t is paired with branch column t+1
    node_losses = [bce(heads["altAlterationsPreds"][i].squeeze(-1), child_mean(ids).squeeze(-1),
                       weight=depth_mask[i].unsqueeze(0))
                   for i, ids in enumerate(alt_links) if len(ids) > 0]
    node_losses.append(bce(heads["txtAlterationsPreds"][:, :-1].squeeze(-1), root_target[:, 1:]))
    alt_loss = torch.stack(node_losses).mean()
    return (in_loss + alt_loss) / 2, in_loss, alt_loss


def discriminator_loss(input_scores, alt_scores, n_update, depth_mask, heads, links, depth_weights=None):
    """calcDiscriminatorLoss (DiscriminatorLoss.py:37-74).  input_scores (B,S); alt_scores (B,N,S)
    zero-padded beyond K; depth_mask (N,S).  Means run over every element, padding included (quirk 9)."""
    shifted = input_scores[:, 1:]
    in_loss = bce(shifted, torch.ones_like(shifted))
    labels = torch.zeros_like(alt_scores) if depth_weights is None else depth_weights.unsqueeze(0).expand_as(alt_scores)
    alt_loss = bce(alt_scores[:, :n_update], labels[:, :n_update], depth_mask[:n_update])
    vh, vh_in, vh_alt = value_head_loss(shifted, alt_scores, depth_mask, heads, links)
    return {"loss-Input": in_loss, "loss-Alterations": alt_loss, "ValueHeadLoss": vh,
            "ValueHead-InputLoss": vh_in, "ValueHead-AltLoss": vh_alt,
            "numAlterationSamples": alt_scores[:, :n_update].shape[1]}


def combine_disc_losses(d, separate_input_alt=True, input_loss_weight=1.0, pred_weight=1.0, value_head_weight=0.25):
    """GanSetup.py:214-221."""
    if separate_input_alt:
        pred = (d["loss-Input"] * input_loss_weight + d["loss-Alterations"]) / 2
    else:
        n = d["numAlterationSamples"]
        pred = (d["loss-Input"] * input_loss_weight + d["loss-Alterations"] * n) / (n + 1)
    out = dict(d)
    out["predLoss"] = pred
    out["loss"] = pred * pred_weight + d["ValueHeadLoss"] * value_head_weight
    return out


def dual_head_advantages(input_scores, alt_scores, in_head, alt_head, wA, wB, wC, in_policy, alt_policy,
                         epsilon=1e-5, min_den=1e-5):
    """dualHeadAdvantages (AdvantageCalculations.py:5-46); heads are the TEXT value heads (B,S,1)."""
    in_head, alt_head = in_head.detach().squeeze(-1), alt_head.detach().squeeze(-1)
    heads_ok = 1 - torch.greater_equal(alt_head, in_head).int()
    shifted = torch.cat((input_scores[:, 1:], torch.zeros(input_scores.shape[0], 1)), dim=1)
    scores = torch.cat((shifted.unsqueeze(1), alt_scores), dim=1)                       # (B, 1+N, S)
    zone_a = torch.greater_equal(scores, in_head.unsqueeze(1))
    above_alt = torch.greater_equal(scores, alt_head.unsqueeze(1))
    zone_c = 1 - above_alt.int()
    zone_b = torch.logical_xor(zone_a, above_alt)
    den = torch.clamp(torch.abs(in_head - alt_head) + epsilon, min=min_den)
    scaled_b = (scores - alt_head.unsqueeze(1)) / den.unsqueeze(1)
    adv = zone_a.float() * wA + scaled_b * zone_b.float() * wB + zone_c.float() * wC
    in_adv, alt_adv = adv[:, :1], adv[:, 1:]
    if in_policy == IGNORE_SAMPLE:
        in_adv = in_adv * heads_ok.unsqueeze(1)
    if alt_policy == IGNORE_SAMPLE:
        alt_adv = alt_adv * heads_ok.unsqueeze(1)
    return in_adv, alt_adv, zone_c.float()


def dual_head_advantages_full(input_scores, alt_scores, in_head, alt_head, wA, wB, wC, in_policy, alt_policy, epsilon=1e-5, min_den=1e-5):
    """dualHeadAdvantages with its whole return structure (AdvantageCalculations.py:44-46): the advantages plus
    (zoneA, zoneB, zoneC, advantageCorrectMask, inputCorrectMask)."""
    ih, ah = in_head.detach().squeeze(-1), alt_head.detach().squeeze(-1)
    in_adv, alt_adv, zone_c = dual_head_advantages(input_scores, alt_scores, in_head, alt_head, wA, wB, wC, in_policy, alt_policy,
                                                   epsilon, min_den)
    shifted = torch.cat((input_scores[:, 1:], torch.zeros(input_scores.shape[0], 1)), dim=1)
    scores = torch.cat((shifted.unsqueeze(1), alt_scores), dim=1)
    zone_a = torch.greater_equal(scores, ih.unsqueeze(1))
    zone_b = torch.logical_xor(zone_a, torch.greater_equal(scores, ah.unsqueeze(1)))
    heads_ok = 1 - torch.greater_equal(ah, ih).int()
    return in_adv, alt_adv, (zone_a.float(), zone_b.float(), zone_c, heads_ok.float(), torch.greater_equal(input_scores, ah).float())


def reinforce_loss(policy_probs, advantages, use_min_prob_limit=False, neg_push_prob_limit=1e-8, eps=1e-6):
    """reinforceLoss (ReinforceLoss.py:45-52) with calculateMinimumProbLimit (:11-18)."""
    p = torch.clamp(policy_probs, eps, 1 - eps)
    loss = -torch.log(p) * advantages
    if use_min_prob_limit:
        reward = torch.greater_equal(advantages, 0).float()
        inverted = 1 - reward
        loss = loss * (reward + torch.where(p * inverted >= neg_push_prob_limit, inverted, reward))
    return loss


def mirrored_negative_reinforce(policy_probs, advantages, zone_c, eps=1e-6):
    """mirroredNegativeReinforceLoss (ReinforceLoss.py:21-52): -log p·adv outside zone C, -log(1-p)·adv inside."""
    p = torch.clamp(policy_probs, eps, 1 - eps)
    return (-torch.log(p) * advantages) * (1 - zone_c) + (-torch.log(-p + 1) * advantages) * zone_c


def debugging_stats(rl_in, rl_alt, in_adv, alt_adv, zones, heads_ok, input_ok, depth_mask=None):
    """_calculateDebuggingStats (GeneratorLosses.py:34-81).  rl_in (B,1,S-1), rl_alt (B,N,S) BEFORE the depth mask; in_adv
    (B,1,S), alt_adv (B,N,S); zones = (A, B, C) float masks (B,1+N,S); heads_ok, input_ok (B,S).  An empty zone reports 0."""
    def zone_mean(values, mask):
        n = mask.sum()
        return (values * mask).sum() / n if n > 0 else torch.zeros(())

    def alt_factor(zone):                                          # zoneMaskCalculation (:49-63) on a (B,N,S) mask
        if depth_mask is None:
            return zone.mean()
        den = depth_mask.sum() * zone.shape[0]
        return (zone * depth_mask.unsqueeze(0)).sum() / den if den != 0 else torch.zeros(())

    out = {}
    names = (("Rl-", "A"), ("Rl-", "B"), ("RL-", "C"))             # the reference's own capitalisation (:67-72)
    for (pre, z), m in zip(names, zones):
        out[f"{pre}Loss-{z}-Input"] = zone_mean(rl_in, m[:, :1, :-1])
    for (pre, z), m in zip(names, zones):
        out[f"{pre}Loss-{z}-Alt"] = zone_mean(rl_alt, m[:, 1:])
    for (pre, z), m in zip(names, zones):
        out[f"{pre}Advantage-{z}-Input"] = zone_mean(in_adv, m[:, :1])
    for (pre, z), m in zip(names, zones):
        out[f"{pre}Advantage-{z}-Alt"] = zone_mean(alt_adv, m[:, 1:])
    for z, m in zip("ABC", zones):
        out[f"RL-Zone{z}-Input-Factor"] = m[:, :1].mean()
    for z, m in zip("ABC", zones):
        out[f"RL-Zone{z}-Alt-Factor"] = alt_factor(m[:, 1:])
    out["RL-CorrectValueHeads-Factor"] = heads_ok.mean()
    out["RL-CorrectInput-Factor"] = input_ok.mean()
    return out


def mle_loss(input_ids, logits):
    """_mleLossFromLogits (GeneratorLosses.py:133-137)."""
    return F.cross_entropy(logits[..., :-1, :].reshape(-1, logits.shape[-1]), input_ids[..., 1:].reshape(-1),
                           ignore_index=-100)


def generator_loss(input_ids, logits, probs, input_scores, alt_scores, alt_probs, heads, depth_mask,
                   label_mode=ALTERATIONS_AND_INPUT, in_policy=IGNORE_SAMPLE, alt_policy=IGNORE_SAMPLE,
                   wA=1.0, wB=1.0, wC=1.0, teacher_forcing=True, mle_weight=1.0, rl_weight=1.0):
    """calcGeneratorLoss + _reinforceLoss (GeneratorLosses.py:84-180).  alt_probs (B,N,S) padded with 1e-5."""
    out = {"loss-MLE": mle_loss(input_ids, logits)}
    if rl_weight == 0:
        out["loss"] = out["loss-MLE"] * mle_weight
        return out
    input_scores, alt_scores = input_scores.detach(), alt_scores.detach()
    if teacher_forcing:
        input_scores = torch.ones_like(input_scores)                                  # GeneratorLosses.py:14-16
    in_adv, alt_adv, (zone_a, zone_b, zone_c, heads_ok, input_ok) = dual_head_advantages_full(
        input_scores, alt_scores, heads["txtInputPreds"], heads["txtAlterationsPreds"], wA, wB, wC, in_policy, alt_policy)
    in_p = torch.gather(probs[:, :-1], -1, input_ids[:, 1:].unsqueeze(-1)).permute(0, 2, 1)
    rl_in = mirrored_negative_reinforce(in_p, in_adv[:, :, :-1], zone_c[:, :1][:, :, :-1])
    rl_alt = mirrored_negative_reinforce(alt_probs, alt_adv, zone_c[:, 1:])
    with torch.no_grad():                                            # GeneratorLosses.py:110-112,161-162: logged, not trained on
        out.update(debugging_stats(rl_in, rl_alt, in_adv, alt_adv, (zone_a, zone_b, zone_c), heads_ok, input_ok, depth_mask))
    if depth_mask is None:
        rl_alt = rl_alt[:, :, :-1]
        alt_loss = rl_alt.mean()
    else:
        rl_alt = rl_alt * depth_mask
        alt_loss = rl_alt.sum() / depth_mask.sum()
    in_loss = rl_in.mean()
    out["loss-RL-input"], out["loss-RL-alterations"] = in_loss, alt_loss
    out["loss-RL-combined"] = (in_loss + alt_loss) / 2
    out["loss-RL-unweightedCombined"] = (in_loss + alt_loss) / (1 + rl_alt.shape[1])
    out["loss-RL"] = {ALTERATIONS_ONLY: alt_loss, INPUT_ONLY: in_loss, ALTERATIONS_AND_INPUT: out["loss-RL-combined"],
                      ALTERATIONS_AND_INPUT_UNWEIGHTED: out["loss-RL-unweightedCombined"]}[int(label_mode)]
    out["loss"] = out["loss-MLE"] * mle_weight + out["loss-RL"] * rl_weight
    return out
