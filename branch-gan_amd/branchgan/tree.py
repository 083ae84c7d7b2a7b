"""Prediction tree and the generator / discriminator tree forwards — host side.

Mirrors BranchPredictions/MultiStepPrediction.py (reference): PredictionTreeNode, multiStepGenForward,
multiStepDiscForward, clearTree, getSubNodeListTreeStructure, getRandomNodesFromRule keep their names, arguments and the
node fields callers read.  What changed underneath (all exactness-preserving):
  * no mask tensors: a node's mask is the descriptor (ctx_len = sparseIndex + 1, its ancestor path);
  * no KV concatenation: ancestors' (k, v) are passed to the attention kernel as slots;
  * nothing over the vocabulary is materialised: ids come from bg_topn_select on the logits of the K branch rows only
    (quirk 4), label / sampled-token log-probs from bg_softmax_stats (one read of the logits, own backward);
  * the discriminator scores EVERY branch node in ONE forward (multiStepDiscForward walks node by node in the
    reference, :357-362): query (node, k) sees text[0..s_k], its ancestors' column k and itself.
"""
import contextlib
import os
import weakref

import torch

from . import fused, ops
from . import sampling as SamplingStrategies
from . import models
from .models import BranchPast, DiscriminatorOutput
from .sampling import SamplingTechniques

_NOISE_TECHNIQUES = (SamplingTechniques.SAMPLE_NO_REPLACEMENT, SamplingTechniques.EPSILON_SAMPLE_NO_REPLACEMENT,
                     SamplingTechniques.INVERSE_TOP_K_SAMPLE_NO_REPLACEMENT, SamplingTechniques.TOP_P_SAMPLE_NO_REPLACEMENT)


class PredictionTreeNode:
    """MultiStepPrediction.py:6-200.  `past` = per layer (k, v) of THIS node's tokens; `logProbs` is new (log of `probs`)."""

    def __init__(self, ids, parent=None, depth=0, beam=0, probs=None, isRandom=False, sparseIndex=None, rootSeqSize=None,
                 discPreds=None, valueHeadData=None, logProbs=None):
        self.isRandom, self.parent, self.children = isRandom, parent, []
        self.depth, self._probs, self.logProbs, self.beam, self.ids = depth, probs, logProbs, beam, ids
        self.numPredictionIDs = ids.shape[1]
        self.sparseIndex = sparseIndex
        self.rootSeqSize = ids.shape[1] if depth == 0 else rootSeqSize
        self.valueHeadData, self.discPreds = valueHeadData, discPreds
        self.attMask = None
        self.past = None

    @property
    def probs(self):
        """Probabilities of this node's tokens (the reference stores them; here they are exp(logProbs), formed when somebody
        asks - the train step works from the log probabilities and never does)."""
        if callable(self._probs):
            self._probs = self._probs()
        elif self._probs is None and self.logProbs is not None:
            self._probs = self.logProbs.exp()
        return self._probs

    @probs.setter
    def probs(self, value):
        self._probs = value

    def addGeneratorPredictionToNode(self, ids, past, attMask, probs, isRandomSamples=None, logProbs=None):
        self.past, self.attMask = past, attMask
        isRandomSamples = [False] * ids.shape[1] if isRandomSamples is None else isRandomSamples
        for i in range(ids.shape[1]):
            beam = i if self.parent is None else self.beam
            self.children.append(PredictionTreeNode(ids[:, i], self, self.depth + 1, beam, None if probs is None else probs[:, :, i], isRandomSamples[i],
                                                    sparseIndex=self.sparseIndex, rootSeqSize=self.rootSeqSize,
                                                    logProbs=None if logProbs is None else part_of(logProbs, 2, i)))

    def addDiscPredictionToNode(self, discPreds, past, attMask, valueHeadData=None):
        self.past, self.attMask, self.discPreds, self.valueHeadData = past, attMask, discPreds, valueHeadData

    def getPaddedPosIDs(self, padValue=9999):
        if self.depth == 0 or self.ids.shape[1] == self.rootSeqSize:
            return self.sparseIndex + self.depth
        pad = torch.ones(self.rootSeqSize - self.ids.shape[1], device=self.sparseIndex.device) * padValue
        return torch.cat((self.sparseIndex + self.depth, pad), dim=0)

    def getPaddedAlterationProbs(self, padValue=0.00001):
        if self.depth == 0 or self.ids.shape[1] == self.rootSeqSize:
            return self.probs
        pad = torch.zeros((self.probs.shape[0], self.rootSeqSize - self.probs.shape[1]), device=self.probs.device) + padValue
        return torch.cat((self.probs, pad), dim=1)

    def getLeaves(self, includeRandomNodes=True):
        if not self.children:
            return self
        out = []
        for c in self.children:
            if includeRandomNodes or not c.isRandom:
                r = c.getLeaves(includeRandomNodes)
                out.extend(r if isinstance(r, list) else [r])
        return out

    def getSubNodes(self):
        out = []
        for n in self.children:
            out.append(n)
            out.extend(n.getSubNodes())
        return out

    def ancestors(self):
        """Branch ancestors in depth order (root excluded)."""
        chain, n = [], self.parent
        while n is not None and n.depth > 0:
            chain.append(n)
            n = n.parent
        return chain[::-1]

    def root(self):
        n = self
        while n.parent is not None:
            n = n.parent
        return n

    def getAttentionMask(self, batchSize, device):
        """The reference's bool mask (:92-127), built arithmetically on demand (debug / compat only)."""
        if self.depth == 0:
            return None
        S, K = self.rootSeqSize, self.numPredictionIDs
        sp = self.sparseIndex.to(device)
        ctx = torch.arange(1, K + 1, device=device) if S == K else sp + 1
        text = torch.arange(S, device=device).unsqueeze(0) < ctx.unsqueeze(1)
        anc = torch.eye(K, dtype=torch.bool, device=device).repeat(1, self.depth - 1)
        m = torch.cat((text, anc, torch.ones(K, 1, dtype=torch.bool, device=device)), dim=1)
        return m[None, None].expand(batchSize, 1, K, -1)

    def getFullPast(self):
        """Reference form (:140-152): per layer (k, v) concatenated root-first.  The engine never calls this."""
        if self.depth == 0:
            return None
        chain = [self.root()] + self.ancestors()
        L = len(chain[0].past)
        return [(torch.cat([c.past[l][0] for c in chain], dim=-2), torch.cat([c.past[l][1] for c in chain], dim=-2)) for l in range(L)]

    def branchPast(self, ctx_len):
        """Engine form of (getFullPast, getAttentionMask): text KV, ancestor slots, descriptor."""
        root, anc = self.root(), self.ancestors()
        L = len(root.past)
        return BranchPast(text=[root.past[l] for l in range(L)], ancestors=[[a.past[l] for a in anc] for l in range(L)],
                          ctx_len=ctx_len, paths=[list(range(len(anc) + 1))], n_query_nodes=1)

    def producePredictionStep(self, device):
        """(:154-166) ids, int32 position ids, past, mask — in the engine form (BranchPast carries past AND mask)."""
        B = self.ids.shape[0]
        if self.depth == 0:
            pos = torch.arange(self.ids.shape[-1], dtype=torch.int32, device=device)
            return self.ids.to(device), pos.unsqueeze(0).expand(B, -1), None, None
        pos = (_position_table(self.sparseIndex, device)[self.depth] if self.depth < _POS_DEPTHS
               else (self.sparseIndex.to(device) + self.depth).to(torch.int32)).unsqueeze(0).expand(B, -1)
        return self.ids.to(device), pos, self.branchPast(ctx_len_of(self.sparseIndex, device)), None

    def clearNode(self, clearAll=False, clearAttMask=False, clearSparseIndex=False):
        if clearAll:
            self.ids = None
            self.discPreds = self.valueHeadData = self.probs = self.logProbs = None
        if clearAttMask or clearAll:
            self.attMask = None
        self.past = None
        if clearSparseIndex:
            self.sparseIndex = None

    def calculateNumNodes(self):
        return 1 + sum(n.calculateNumNodes() for n in self.children)


_depth_cache = {}


def _depths_tensor(depths, device):
    """Per-node depths on the device, cached per tree shape (no host-to-device copy inside a captured step)."""
    key = (depths, str(device))
    if key not in _depth_cache:
        _depth_cache[key] = torch.tensor(depths, device=device)
    return _depth_cache[key]


_POS_DEPTHS = 64
_pos_cache = {}


def _position_table(sparseIndex, device):
    """int32 (64, K): row d = sparseIndex + d - the position ids of depth d and, in row 1, ctx_len.  Built once per sparse
    index (two launches) instead of an add + cast per tree node and forward; keyed by the tensor's version and by the capture
    it was made in (memory of a captured graph's pool is not handed to eager code)."""
    capturing = torch.cuda.is_current_stream_capturing() if sparseIndex.is_cuda else False
    key = (id(sparseIndex), sparseIndex._version, sparseIndex.data_ptr(), str(device), capturing, fused._capture_epoch if capturing else 0)
    hit = _pos_cache.get(id(sparseIndex))
    if hit is None or hit[0] != key or hit[1]() is not sparseIndex:
        table = (sparseIndex.to(device).unsqueeze(0) + _depths_tensor(tuple(range(_POS_DEPTHS)), device).unsqueeze(1)).to(torch.int32)
        if len(_pos_cache) > 64:
            _pos_cache.clear()
        hit = _pos_cache[id(sparseIndex)] = (key, weakref.ref(sparseIndex), table)
    return hit[2]


def ctx_len_of(sparseIndex, device):
    return _position_table(sparseIndex, device)[1]


def clearTree(root, clearAll=False, clearAttMask=False, clearSparseIndex=False):
    for n in root.children:
        clearTree(n, clearAll, clearAttMask, clearSparseIndex)
    root.clearNode(clearAll, clearAttMask, clearSparseIndex)


def getSubNodeListTreeStructure(root, subNodes):
    for i, n in enumerate(subNodes):
        n.listIndex = i
    alterationLinks = [[c.listIndex for c in n.children] for n in subNodes]
    rootLinks = [n.listIndex for n in root.children if getattr(n, 'listIndex', None) is not None]
    return rootLinks, alterationLinks


def getRandomNodesFromRule(strat, params, numAlterations):
    if strat.name not in SamplingStrategies.RANDOM_TECHNIQUES:
        return None
    if strat == SamplingTechniques.RANDOM:
        return [True] * numAlterations
    return [False] * (numAlterations - params['numRandomSamples']) + [True] * params['numRandomSamples']


def _sample_rows(logits2d, row_map, labels, numSamples, strat, params, B, T_out, noise_fn):
    """ids (B, n, T_out) for the selected rows of a (rows, V) logits tensor.  TOP_K works on the logits directly
    (softmax is monotone); the noise-consuming techniques need the reference's probs bits, so the probs of those few
    rows are formed (B*K rows, not B*S) and go through the probs entry points."""
    V = logits2d.shape[-1]
    blocked = labels.reshape(-1, 1) if labels is not None else None
    if strat == SamplingTechniques.TOP_K:
        ids = ops.topn_select(logits2d, numSamples, blocked=blocked, blocked_value=float("-inf"), row_map=row_map)
        return ids.view(B, T_out, numSamples).permute(0, 2, 1)
    rows = logits2d if row_map is None else logits2d[row_map]
    probs = torch.softmax(rows.float(), dim=-1).view(B, T_out, V)
    noise = None
    if strat in _NOISE_TECHNIQUES:
        noise = noise_fn((B, T_out, V), probs.device) if noise_fn is not None else None
    # express "block label of row t" through a synthetic inputIDs whose entry t+1 is that label
    if labels is not None:
        lab = labels.view(B, T_out)
        fake_ids = torch.cat((torch.zeros(B, 1, dtype=lab.dtype, device=lab.device), lab), dim=1)      # (B, T_out+1)
        probs_ext = torch.cat((probs, probs[:, -1:]), dim=1)                                             # dummy last row
        alt, _ = SamplingStrategies.sampleFromGeneratorDistribution(probs_ext.detach(), numSamples, fake_ids, strat,
                                                                    blockSamplingOfLabels=True, includeFinalTimeStep=False,
                                                                    noise=noise, **_clean(params))
    else:
        fake_ids = torch.zeros(B, T_out, dtype=torch.int64, device=probs.device)
        alt, _ = SamplingStrategies.sampleFromGeneratorDistribution(probs.detach(), numSamples, fake_ids, strat,
                                                                    blockSamplingOfLabels=False, includeFinalTimeStep=True,
                                                                    noise=noise, **_clean(params))
    return alt[:, :, 1:]


def _clean(params):
    return {k: v for k, v in params.items() if k not in ('includeFinalTimeStep', 'blockSamplingOfLabels', 'dropoutRate')}


def _branch_step(generator, node, device, dropoutRate, numAlterations, strat, params, noise_fn, ids=None, uid=0):
    """One generator forward over a branch node and the sampling of its children: -> (log probs (B,K,n), past, ids (B,n,K)).
    With `ids` given nothing is sampled: the log probs of those ids are returned (re-forward with autograd).
    uid: the node's number among the tree's forwarded nodes (its rows in the tree pass: dropout masks line up through it)."""
    predIDs, positionalIDs, past, _ = node.producePredictionStep(device)
    past.node_uid0 = uid
    output = generator.forward({'input_ids': predIDs, 'position_ids': positionalIDs, 'textualPast': past,
                                'returnLmLogits': True, 'use_cache': True, 'dropoutRate': dropoutRate})
    step_logits = output.prediction.logits                                                # (B, K, V)
    B, K, V = step_logits.shape
    if ids is None:
        ids = _sample_rows(step_logits.detach().reshape(B * K, V), None, None, numAlterations, strat, params, B, K, noise_fn)
    _, lp = ops.softmax_stats(step_logits, ids.permute(0, 2, 1).contiguous())            # (B, K, n)
    return lp, output.prediction.past_key_values, ids


def _set_child_log_probs(node, lp):
    for j, child in enumerate(node.children):
        child.logProbs = part_of(lp, 2, j)               # (tagged: GanSetup stacks the nodes' log-probs again, see restack)
        child.probs = None                                  # lazily exp(logProbs)


_row_map_cache = {}
# A/B: sample without autograd and differentiate in one tree pass also when the generator draws dropout masks (by site)
DEFER_UNDER_DROPOUT = os.environ.get("BRANCHGAN_DEFER_UNDER_DROPOUT", "1") != "0"


def _node_rows(B, N, K, i, device):
    """Rows of node i in the flattened (B * N * K, V) logits of a tree pass."""
    key = (B, N, K, i, str(device))
    if key not in _row_map_cache:
        _row_map_cache[key] = (torch.arange(B, device=device).unsqueeze(1) * (N * K) + i * K + torch.arange(K, device=device).unsqueeze(0)).reshape(-1)
    return _row_map_cache[key]


def _attach_branch_gradients(generator, root, forwarded, device, pending=(), noise_fn=None):
    """Gives the children of every forwarded branch node autograd-connected log probs: one tree pass over all of them
    (<= BG_MAX_SLOTS nodes), else node by node in depth order.  `pending`: (node, strategy, params, numAlterations) of the
    deepest nodes, which have not been forwarded at all yet: they ride in the same pass and their children are sampled from
    ITS logits - the same numbers a separate branch step over them would compute, one forward less per step."""
    forwarded = list(forwarded) + [p[0] for p in pending]
    n_done = len(forwarded) - len(pending)
    B, K = forwarded[0].ids.shape
    N = len(forwarded)
    if N > ops.MAX_SLOTS:
        assert not pending
        for i, node in enumerate(forwarded):
            ids = torch.stack([c.ids for c in node.children], dim=1)
            lp, past, _ = _branch_step(generator, node, device, None, ids.shape[1], None, None, None, ids=ids, uid=i)
            node.past = past
            _set_child_log_probs(node, lp)
        return
    index = {id(n): i for i, n in enumerate(forwarded)}
    paths = [[index[id(a)] for a in n.ancestors()] + [i] for i, n in enumerate(forwarded)]
    sparseIndex = root.sparseIndex.to(device)
    all_ids = torch.stack([n.ids.detach() for n in forwarded], dim=1).reshape(B, N * K).to(device)
    depths = _depths_tensor(tuple(n.depth for n in forwarded), device)
    pos = _position_table(sparseIndex, device).index_select(0, depths).reshape(1, N * K).expand(B, -1)
    L = len(root.past)
    past = BranchPast(text=[root.past[l] for l in range(L)], ancestors=[[] for _ in range(L)],
                      ctx_len=ctx_len_of(sparseIndex, device), paths=paths, n_query_nodes=N)
    out = generator.forward({'input_ids': all_ids, 'position_ids': pos, 'textualPast': past, 'returnLmLogits': True,
                             'use_cache': False})
    if pending:
        logits2d = out.prediction.logits.detach().reshape(B * N * K, -1)
        with torch.no_grad():
            for i, (node, strat, params, numAlterations) in enumerate(pending, start=n_done):
                ids = _sample_rows(logits2d, _node_rows(B, N, K, i, device), None, numAlterations, strat, params, B, K, noise_fn)
                node.addGeneratorPredictionToNode(ids, None, None, None, getRandomNodesFromRule(strat, params, numAlterations))
    n_max = max(len(n.children) for n in forwarded)
    gather = torch.full((B, N, K, n_max), -1, dtype=torch.int64, device=device)
    for i, n in enumerate(forwarded):
        for j, c in enumerate(n.children):
            gather[:, i, :, j] = c.ids
    _, lp = ops.softmax_stats(out.prediction.logits, gather.view(B, N * K, n_max))
    lp = lp.view(B, N, K, n_max)
    for i, n in enumerate(forwarded):
        _set_child_log_probs(n, part_of(lp, 1, i))


def multiStepGenForward(generator, inputIDs, sampleRules, device, clearTreePast=True, clearFullTree=False, verbose=False,
                        sparseIndex=None, noise_fn=None):
    """MultiStepPrediction.py:243-332.  Returns (root, firstPrediction, firstProbs (lazy), maskedProbs (lazy));
    additionally root.labelLogProbs (B, S-1) = log prob of every true next token (for MLE and REINFORCE)."""
    inputIDs = inputIDs.to(device)
    B, S = inputIDs.shape
    firstRule = sampleRules[0]
    firstOutput = generator.forward({'input_ids': inputIDs, 'use_cache': True, 'returnLmLogits': True,
                                     'dropoutRate': firstRule.samplingParams.get('dropoutRate', None)})
    logits = firstOutput.prediction.logits
    V = logits.shape[-1]
    firstRule.samplingParams['includeFinalTimeStep'] = True
    if sparseIndex is None:
        sparseIndex = torch.arange(S, device=device)
    sparseIndex = sparseIndex.to(device).long()
    K = sparseIndex.numel()
    ctx_len = ctx_len_of(sparseIndex, device)
    noise_fn = noise_fn or SamplingStrategies.getRand

    # rows to sample: (b, s_k); the label of row s is inputIDs[s + 1] (none for the last position)   [quirks 4, 5]
    logits2d = logits.reshape(B * S, V)
    row_map = (torch.arange(B, device=device).unsqueeze(1) * S + sparseIndex.unsqueeze(0)).reshape(-1)
    nxt = torch.cat((inputIDs[:, 1:], torch.full((B, 1), -1, dtype=inputIDs.dtype, device=device)), dim=1)
    labels_k = nxt[:, sparseIndex]
    n0 = firstRule.numAlterations
    sparseAltIDs = _sample_rows(logits2d.detach(), row_map, labels_k, n0, firstRule.samplingStrategy, firstRule.samplingParams,
                                B, K, noise_fn)                                          # (B, n0, K)

    # one pass over the logits: lse, log p(label) for every row, log p(sampled ids) on the branch rows   [quirk 6]
    gather = torch.full((B, S, 1 + n0), -1, dtype=torch.int64, device=device)
    gather[:, :, 0] = nxt
    gather[:, sparseIndex, 1:] = sparseAltIDs.permute(0, 2, 1)
    _, logp = ops.softmax_stats(logits, gather)
    labelLogProbs = logp[:, :-1, 0]                                                      # (B, S-1)
    altLogProbs = logp[:, sparseIndex, 1:]                                               # (B, K, n0)
    inputProbs = lambda: torch.cat((torch.zeros(B, 1, device=device), labelLogProbs.exp()), dim=-1)     # formed on request

    root = PredictionTreeNode(inputIDs, sparseIndex=sparseIndex, probs=inputProbs)
    root.labelLogProbs = labelLogProbs
    randomNodeInfo = getRandomNodesFromRule(firstRule.samplingStrategy, firstRule.samplingParams, n0)
    root.addGeneratorPredictionToNode(sparseAltIDs, firstOutput.prediction.past_key_values, None, None,
                                      isRandomSamples=randomNodeInfo, logProbs=altLogProbs)

    # Sampling is sequential by depth, differentiation is not: the branch forwards below run WITHOUT autograd (they only have
    # to produce ids), then ONE pass over every forwarded node (the same tree attention the discriminator uses) rebuilds their
    # logits with autograd, so the backward is one pass over N*K tokens instead of one per node.  Same mathematics as the
    # reference's per-node graph (MultiStepPrediction.py:296-327).  Under dropout the two passes must draw the SAME masks for a
    # node's rows: true where the masks are functions of (step, site, logical row) (models.dropout_by_site: the GPT-2 family on
    # the fused path, at the config's rates); per-rule rates, the module path and NeoX's nn.Dropout keep the per-node graph.
    track = torch.is_grad_enabled() and logits.requires_grad
    deferred = track and not any(r.samplingParams.get('dropoutRate') for r in sampleRules) and \
        (not models.dropout_active(generator) or (DEFER_UNDER_DROPOUT and models.dropout_by_site(generator)))
    forwarded, pending = [], []
    rules = sampleRules[1:]
    for r_i, rule in enumerate(rules):
        for node in root.getLeaves(rule.continueFromRandom):
            if node.beam == 0 and rule.firstBeamSamplingStrategy is not None:
                strat, params = rule.firstBeamSamplingStrategy, rule.firstBeamSamplingParams
                numAlterations = rule.numAlterations + params['numRandomSamples']
            else:
                strat, params, numAlterations = rule.samplingStrategy, rule.samplingParams, rule.numAlterations
            params['includeFinalTimeStep'] = True
            params['blockSamplingOfLabels'] = False
            if deferred and r_i == len(rules) - 1:
                pending.append((node, strat, params, numAlterations))       # the deepest nodes: sampled from the gradient pass itself
                continue
            with torch.no_grad() if deferred else contextlib.nullcontext():
                lp, past, altIDs = _branch_step(generator, node, device, rule.samplingParams.get('dropoutRate', None),
                                                numAlterations, strat, params, noise_fn, uid=len(forwarded))
            node.addGeneratorPredictionToNode(altIDs, past, None, None, getRandomNodesFromRule(strat, params, numAlterations),
                                              logProbs=lp)
            forwarded.append(node)
    if pending and len(forwarded) + len(pending) > ops.MAX_SLOTS:          # too many nodes for one pass: a branch step each
        for node, strat, params, numAlterations in pending:
            with torch.no_grad():
                lp, past, altIDs = _branch_step(generator, node, device, None, numAlterations, strat, params, noise_fn, uid=len(forwarded))
            node.addGeneratorPredictionToNode(altIDs, past, None, None, getRandomNodesFromRule(strat, params, numAlterations),
                                              logProbs=lp)
            forwarded.append(node)
        pending = []
    if deferred and (forwarded or pending):
        _attach_branch_gradients(generator, root, forwarded, device, pending, noise_fn)
    if clearTreePast or clearFullTree:
        clearTree(root, clearFullTree)
    return root, firstOutput.prediction, firstOutput.classProbs, \
        SamplingStrategies.MaskedProbs(firstOutput.classProbs, inputIDs, True, True)


def discRootForward(discriminator, ids, dualValueHeads=False):
    """The discriminator's pass over the real text (MultiStepPrediction.py:340-347)."""
    out = discriminator.forward({'input_ids': ids, 'use_cache': True, 'output_hidden_states': False},
                                doDiscForward=False, doTransformerPassWithPrediction=True)
    inV = altV = None
    if dualValueHeads:
        _, inV, altV = discriminator.forward({'pred': out.transformerPred}, doDiscForward=False, doFinalizeSinglePrediction=True)
    return out, inV, altV


def part_of(whole, dim, i):
    """whole.select(dim, i), tagged with where it came from (see restack); `whole` may itself be such a part."""
    part = whole.select(dim, i)
    part._bg_part_of = (whole, dim, i)
    return part


def _origin(t):
    """(root tensor, {root dimension: index} picked on the way down to `t`) of a chain of part_of selections, else None."""
    tag = getattr(t, '_bg_part_of', None)
    if tag is None:
        return None
    chain = []
    while tag is not None:
        t, dim, i = tag
        chain.append((dim, i))
        tag = getattr(t, '_bg_part_of', None)
    left, picked = list(range(t.dim())), {}
    for dim, i in reversed(chain):                       # from the root outwards: `dim` counts the dimensions still there
        picked[left.pop(dim)] = i
    return t, picked


def restack(parts, dim):
    """torch.stack(parts, dim) without copying what is already together.  Consecutive parts that are the slices 0 .. n-1 (in
    order) of one tensor along one dimension (part_of, possibly nested: the other picked indices equal) are taken as that tensor
    with the dimension moved to `dim`; the runs are joined by one cat.  One run: a view, no launch.  The values are torch.stack's;
    the backward hands each run's gradient over in one piece instead of a zeros + copy + add per slice."""
    origins = [_origin(p) for p in parts]
    if not parts or any(o is None for o in origins):
        return torch.stack(parts, dim=dim)
    runs = []                                            # [root, picked dimensions, [(part, picked)]]
    for p, (root, picked) in zip(parts, origins):
        if runs and runs[-1][0] is root and runs[-1][1] == sorted(picked):
            runs[-1][2].append((p, picked))
        else:
            runs.append([root, sorted(picked), [(p, picked)]])
    pieces = []
    for root, dims, members in runs:
        varying = [d for d in dims if len({m[1][d] for m in members}) > 1]
        if len(members) > 1 and len(varying) == 1 and [m[1][varying[0]] for m in members] == list(range(root.shape[varying[0]])):
            v, t = varying[0], root
            for d in sorted((d for d in dims if d != v), reverse=True):      # the constant picks, highest dimension first
                t = t.select(d, members[0][1][d])
            pieces.append(t.movedim(v - sum(1 for d in dims if d < v), dim))
        elif len(members) == 1:
            pieces.append(members[0][0].unsqueeze(dim))
        else:
            pieces.append(torch.stack([m[0] for m in members], dim=dim))
    return pieces[0] if len(pieces) == 1 else torch.cat(pieces, dim=dim)


def multiStepDiscForward(discriminator, predictionTreeRoot, device, clearTreePast=True, clearFullTree=False,
                         dualValueHeads=False, rootPass=None):
    """MultiStepPrediction.py:335-366.  On the fused block path ONE forward over text + every branch node; otherwise TWO: the
    root pass (optionally precomputed: `rootPass` = the result of discRootForward on the same ids) and one pass over the nodes."""
    root = predictionTreeRoot
    nodes = root.getSubNodes()
    ids = root.ids.detach().to(device)
    B, S = ids.shape

    def finish(node, pred, sl=None):
        trans = pred.transformerPred
        if dualValueHeads:
            _, inV, altV = discriminator.forward({'pred': trans}, doDiscForward=False, doFinalizeSinglePrediction=True)
        return (inV, altV) if dualValueHeads else (None, None)

    tree_args = None
    if nodes:
        sparseIndex = root.sparseIndex.to(device)
        K, N = sparseIndex.numel(), len(nodes)
        index = {id(n): i for i, n in enumerate(nodes)}
        paths = [[index[id(a)] for a in n.ancestors()] + [i] for i, n in enumerate(nodes)]
        all_ids = torch.stack([n.ids.detach() for n in nodes], dim=1).reshape(B, N * K).to(device)
        depths = _depths_tensor(tuple(n.depth for n in nodes), device)
        pos = _position_table(sparseIndex, device).index_select(0, depths).reshape(1, N * K).expand(B, -1)
        tree_args = (K, N, paths, all_ids, pos, ctx_len_of(sparseIndex, device))

    def node_outputs(pred, inV, altV, kv, K, N):
        inV4, altV4 = (inV.view(B, N, K, 1), altV.view(B, N, K, 1)) if dualValueHeads else (None, None)
        for i, n in enumerate(nodes):
            # every per-node piece remembers the whole it is a slice of (part_of): a caller that stacks the pieces of ALL nodes
            # again (GanSetup's loss inputs) takes the whole instead - no cat, and no zeros + copy + add per node in backward
            vh = {'InputValue': part_of(inV4, 1, i), 'AlterationsValue': part_of(altV4, 1, i)} if dualValueHeads else None
            node_kv = [(k.unflatten(2, (N, K))[:, :, i], v.unflatten(2, (N, K))[:, :, i]) for k, v in kv]
            n.addDiscPredictionToNode(DiscriminatorOutput(part_of(pred, 1, i)), node_kv, None, vh)

    lm = getattr(discriminator, 'nonResidualModel', None)
    if tree_args is not None and rootPass is None and lm is not None and lm.supportsMergedTree(ids) and \
            tree_args[1] <= ops.MAX_SLOTS:
        # ONE forward over [text | all tree nodes] (BranchPast.merged_text): every GEMM / LayerNorm of a layer sees one batch
        K, N, paths, all_ids, pos, ctx_len = tree_args
        L = lm.config.num_hidden_layers
        pos_all = torch.cat((torch.arange(S, dtype=torch.int32, device=device).unsqueeze(0).expand(B, -1), pos), dim=1)
        past = BranchPast(text=None, ancestors=[[] for _ in range(L)], ctx_len=ctx_len, paths=paths, n_query_nodes=N, merged_text=S)
        out = discriminator.forward({'input_ids': torch.cat((ids, all_ids), dim=1), 'position_ids': pos_all, 'textualPast': past,
                                     'use_cache': True}, doDiscForward=False, doTransformerPassWithPrediction=True)
        inV, altV = finish(None, out)
        kv = out.transformerPred.past_key_values
        # [text | tree] parts by ONE split per tensor: its backward is one cat, where two slices were a zeros + copy each and an add
        cut = lambda t: t.split([S, t.shape[1] - S], dim=1)
        pred_txt, pred_alt = cut(out.prediction)
        (inV_txt, inV_alt), (altV_txt, altV_alt) = (cut(inV), cut(altV)) if dualValueHeads else ((None, None), (None, None))
        root.addDiscPredictionToNode(DiscriminatorOutput(pred_txt), [(k[:, :, :S], v[:, :, :S]) for k, v in kv], None,
                                     {'InputValue': inV_txt, 'AlterationsValue': altV_txt} if dualValueHeads else None)
        node_outputs(pred_alt.reshape(B, N, K), inV_alt, altV_alt, [(k[:, :, S:], v[:, :, S:]) for k, v in kv], K, N)
    else:
        out, inV, altV = rootPass if rootPass is not None else discRootForward(discriminator, ids, dualValueHeads)
        root.addDiscPredictionToNode(DiscriminatorOutput(out.prediction), out.transformerPred.past_key_values, None,
                                     {'InputValue': inV, 'AlterationsValue': altV} if dualValueHeads else None)
        if tree_args is not None:
            K, N, paths, all_ids, pos, ctx_len = tree_args
            L = len(root.past)
            past = BranchPast(text=[root.past[l] for l in range(L)], ancestors=[[] for _ in range(L)],
                              ctx_len=ctx_len, paths=paths, n_query_nodes=N)
            out = discriminator.forward({'input_ids': all_ids, 'position_ids': pos, 'textualPast': past, 'use_cache': True},
                                        doDiscForward=False, doTransformerPassWithPrediction=True)
            inV, altV = finish(None, out)
            node_outputs(out.prediction.view(B, N, K), inV, altV, out.transformerPred.past_key_values, K, N)
    if clearTreePast or clearFullTree:
        clearTree(root, clearFullTree, clearAttMask=True, clearSparseIndex=True)
    return root
