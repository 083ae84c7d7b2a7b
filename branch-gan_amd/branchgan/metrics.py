"""Evaluation metrics of a forward pass - same dictionary keys and values as the reference's
Training/Metrics/{GeneratorMetrics,DiscriminatorMetrics,MetricUtils}.py, without its sorts.

Everything the reference derives from `torch.sort` / `torch.argsort` over the (B, S, V) probability tensor (label placement,
top-r probability masses for r up to 1024, top-5/10/20 ids) comes from two kernels that read each row once:
`bg_rank_stats` (csrc/metrics.hip) and `bg_topn_select` (csrc/topn.hip).  What is left are means over (B, S)-sized tensors.

Tied probabilities: torch.argsort is unstable, the kernels order equal values by increasing token id; a placement / top-k id
can differ from the reference's only among exactly equal probabilities.
"""
import torch

from . import ops

RANK_MASSES = (1, 8, 16, 64, 128, 256, 512, 1024)                 # GeneratorMetrics.py:24
LOSS_RANK_RATES = (0, 8, 16, 64, 128, 256, 512, 1024)             # GeneratorMetrics.py:5
PLACEMENT_RANK_RATES = (0, 3, 8, 16, 32)                          # GeneratorMetrics.py:29


def entropy(probs, epsilon=1e-8):
    """Training/Utils.py:20-21 per row, fused (one read of probs)."""
    return ops.rank_stats(_dense(probs), entropy_eps=epsilon)['entropy']


def _dense(p):
    return p.materialize() if hasattr(p, "materialize") else p


def mleLossFromProbs(inputIDs, generatorProbs, aggrigate=True):
    """MetricUtils.py:4-9."""
    selected = torch.gather(_dense(generatorProbs)[:, :-1], dim=-1, index=inputIDs[:, 1:].unsqueeze(-1))
    logp = torch.log(selected)
    return -logp.mean() if aggrigate else -logp


def mleLossFromLogits(inputIDs, logits):
    """MetricUtils.py:12-16 through the fused softmax statistics (no (B, S, V) log-softmax)."""
    ids = torch.cat((inputIDs[:, 1:], torch.full_like(inputIDs[:, :1], -1)), dim=1).unsqueeze(-1)
    _, logp = ops.softmax_stats(logits, ids)
    return -logp[:, :-1, 0].mean()


def _row_stats(inputIDs, generatorProbs, eps=1e-8):
    """One pass over (B, S, V): label prob / placement for rows 0..S-2 (label = next token), min, max, entropy, top masses."""
    probs = _dense(generatorProbs)
    labels = torch.cat((inputIDs[:, 1:], torch.full_like(inputIDs[:, :1], -1)), dim=1)
    return ops.rank_stats(probs, labels, RANK_MASSES, entropy_eps=eps)


def _shared_generator_metrics(st):
    placements = st['placement'][:, :-1].float()                  # (B, S-1), GeneratorMetrics.py:29-50
    labelProbs = st['label_value'][:, :-1]
    losses = -torch.log(labelProbs)                               # :5-19 (the double division by the loss sum is the reference's)
    lossSum = losses.sum()
    meanLosses = losses / lossSum
    data = {'Input-Prob-Mean': labelProbs.mean(), 'Input-Prob-STD': labelProbs.std(),
            'Probs-Min-Mean': st['min'].mean(), 'Probs-Min-Min': st['min'].min(), 'Probs-Max-Mean': st['max'].mean(),
            'Input-PredPlacement-Mean': placements.mean(), 'Input-PredPlacement-Median': placements.median(),
            'Input-PredPlacement-STD': placements.std()}
    for r in PLACEMENT_RANK_RATES:
        data['Input-PredPlacement-RankRate-{}'.format(r)] = (placements <= r).float().mean()
    for i, r in enumerate(RANK_MASSES):
        data['RankProbMass-{}'.format(r)] = st['top_mass'][..., i].mean()
    for r in LOSS_RANK_RATES:
        data['RankLossRatio-{}'.format(r)] = (meanLosses * (placements <= r).float()).sum() / lossSum
    return data


def calculateGeneratorMetricsNoDiscriminator(inputIDs, generatorProbs, generatorEntropies=None):
    """GeneratorMetrics.py:58-86.  generatorEntropies=None: taken from the same pass over the probabilities."""
    st = _row_stats(inputIDs, generatorProbs)
    ent = st['entropy'] if generatorEntropies is None else generatorEntropies
    data = _shared_generator_metrics(st)
    data['Entropy-Mean'], data['Entropy-STD'] = ent.mean(), ent.std()
    return data


def _top_ids(maskedProbs, n):
    """ids of the n largest masked probabilities per row, (B, T', n)."""
    if hasattr(maskedProbs, "input_ids"):                         # sampling.MaskedProbs: block the label inside the kernel
        probs = _dense(maskedProbs.probs)
        B, T, V = probs.shape
        T_out = T if maskedProbs.include_final else T - 1
        rows = probs if maskedProbs.include_final else probs[:, :-1]
        blocked = None
        if maskedProbs.block:
            nxt = maskedProbs.input_ids[:, 1:T]
            if maskedProbs.include_final:
                nxt = torch.cat((nxt, torch.full_like(nxt[:, :1], -1)), dim=1)
            blocked = nxt.reshape(-1, 1)
        rows2 = rows.reshape(B * T_out, V) if rows.is_contiguous() else rows.contiguous().view(B * T_out, V)
        return ops.topn_select(rows2, n, blocked=blocked, blocked_value=0.0).view(B, T_out, n)
    B, T_out, V = maskedProbs.shape
    return ops.topn_select(maskedProbs.reshape(B * T_out, V), n).view(B, T_out, n)


def calculateGeneratorMetrics(inputIDs, generatorProbs, maskedProbs, discInputScores, discAlterationScores):
    """GeneratorMetrics.py:89-140.  discAlterationScores (B, N, S): as in the reference this needs a branch at every position."""
    altScores = discAlterationScores[:, :, 1:]
    advantageScores = torch.cat((discInputScores[:, 1:].unsqueeze(1), altScores), dim=1)
    meanReward = altScores.mean()
    top20 = _top_ids(maskedProbs, 20)
    data = _shared_generator_metrics(_row_stats(inputIDs, generatorProbs))
    data.update({'RL-MeanReward': meanReward, 'RL-MeanAdvantage': advantageScores.mean(dim=1, keepdim=True), 'RL-STDReward': altScores.std(),
                 'RL-FirstSample-MeanReward': altScores[:, 0].mean(), 'RL-FirstSample-STDReward': altScores[:, 0].std(),
                 'Advantage-Mean': (altScores - meanReward).mean(),
                 'Top-5': top20[:, :, :5], 'Top-10': top20[:, :, :10], 'Top-20': top20})
    return data


def calcDicValueHeadMetrics(forwardPassData):
    """DiscriminatorMetrics.py:4-13."""
    heads = forwardPassData.get('DiscValueHeads')
    if heads is None:
        return {}
    return {'Txt-InputValueHead-Mean': heads['txtInputPreds'].mean(), 'Alt-InputValueHead-Mean': heads['altInputPreds'].mean(),
            'Txt-AltValueHead-Mean': heads['txtAlterationsPreds'].mean(), 'Alt-AltValueHead-Mean': heads['altAlterationsPreds'].mean()}


def calcAdvantageRatios(advantages):
    """DiscriminatorMetrics.py:33-47: share of positive advantages for the input row and for the alteration rows."""
    return (advantages[:, :1] > 0).float().mean(), (advantages[:, 1:] > 0).float().mean()


def calcDiscriminatorMetrics(discInputPreds, discAlterationPreds, forwardPassData):
    """DiscriminatorMetrics.py:50-77.  The placement of the real text among the N + 1 scores of a position is a count of
    the alterations scored higher (the reference: an argsort over N + 1 and a CPU one-hot)."""
    inp, alt = discInputPreds[:, 1:], discAlterationPreds[:, :, 1:]
    altWins = alt >= 0.5
    placement = (alt > inp.unsqueeze(1)).sum(dim=1).float()
    results = {'Input-Acc': (inp >= 0.5).float().mean(), 'Alteration-Acc': 1 - altWins.float().mean(),
               'Alteration-FirstSample-Acc': 1 - altWins[:, 0].float().mean(),
               'Input-Mean': inp.mean(), 'Input-STD': inp.std(), 'Alteration-Mean': alt.mean(), 'Alteration-STD': alt.std(),
               'Input-PredPlacement-Mean': placement.mean(), 'Input-PredPlacement-STD': placement.std(),
               'Input-PredPlacement-FirstRate': (placement == 0).float().mean()}
    results.update(calcDicValueHeadMetrics(forwardPassData))
    return results
