"""Oracle: prediction tree, its attention masks, and the generator/discriminator tree forwards.

Restates BranchPredictions/MultiStepPrediction.py (reference).  Test infrastructure only.
A sampling rule here is a tuple (num_alterations, technique_value, params_dict).
"""
import torch

from . import models as M
from . import sampling as S


RANDOM_TECHNIQUES = (S.RANDOM, S.EPSILON_GREEDY, S.EPSILON_SAMPLE, S.EPSILON_SAMPLE_NO_REPLACEMENT)   # SamplingStrategies.py:45-50


def random_node_info(tech, params, n):
    """getRandomNodesFromRule (MultiStepPrediction.py:229-235): which of a sampling call's n children are random draws."""
    if tech not in RANDOM_TECHNIQUES:
        return [False] * n
    if tech == S.RANDOM:
        return [True] * n
    return [False] * (n - params["numRandomSamples"]) + [True] * params["numRandomSamples"]


class Node:
    def __init__(self, ids, parent=None, depth=0, beam=0, probs=None, is_random=False):
        self.ids, self.parent, self.depth, self.beam, self.probs, self.is_random = ids, parent, depth, beam, probs, is_random
        self.children, self.kv, self.mask = [], None, None
        self.disc_pred = self.in_value = self.alt_value = None

    def sub_nodes(self):
        """Pre-order DFS, root excluded (getSubNodes, MultiStepPrediction.py:85-90)."""
        out = []
        for c in self.children:
            out.append(c)
            out.extend(c.sub_nodes())
        return out

    def leaves(self, include_random=True):
        """getLeaves (MultiStepPrediction.py:67-83): random children (and everything below them) are skipped on request."""
        if not self.children:
            return [self]
        return [l for c in self.children if include_random or not c.is_random for l in c.leaves(include_random)]


def depth1_mask(batch, sparse_index, S):
    """MultiStepPrediction.py:100-111: (B,1,K,S+1); row i sees text [0..s_i] (or lower-triangular when
    K == S) and the self column."""
    K = len(sparse_index)
    m = torch.zeros(batch, K, S + 1)
    m[:, :, -1] = 1
    for i, s in enumerate(sparse_index.tolist()):
        m[:, i, :(i + 1 if K == S else s + 1)] = 1
    return m.unsqueeze(1).bool()


def deeper_mask(parent_mask, batch, K):
    """MultiStepPrediction.py:116-127: parent mask minus its self column ‖ (K, K+1) diag + self column."""
    new = torch.zeros(batch, K, K + 1)
    new[:, :, -1] = 1
    new[:, torch.arange(K), torch.arange(K)] = 1
    return torch.cat((parent_mask[..., :-1], new.unsqueeze(1).bool()), dim=-1)


def node_mask(node, batch, sparse_index, S):
    if node.depth == 0:
        return None
    if node.mask is None:
        if node.depth == 1:
            node.mask = depth1_mask(batch, sparse_index, S)
        else:
            node.mask = deeper_mask(node_mask(node.parent, batch, sparse_index, S), batch, len(sparse_index))
    return node.mask


def full_past(node):
    """getFullPast (MultiStepPrediction.py:140-152): per layer, root KV ‖ ancestors' KV in depth order."""
    chain, n = [], node.parent
    while n is not None:
        chain.append(n.kv)
        n = n.parent
    chain.reverse()
    return [(torch.cat([c[l][0] for c in chain], dim=-2), torch.cat([c[l][1] for c in chain], dim=-2))
            for l in range(len(chain[0]))]


def _add_children(node, alt_ids, alt_probs, is_random=None):
    for i in range(alt_ids.shape[1]):
        beam = i if node.parent is None else node.beam
        node.children.append(Node(alt_ids[:, i], node, node.depth + 1, beam, alt_probs[:, :, i], bool(is_random[i]) if is_random else False))


def _substitute(own, forced, row_logits, trace):
    """Teacher forcing for precision studies (no counterpart in the reference): the tree continues from `forced`
    (B, n, K) instead of the ids sampled here; `trace` receives both and, per entry, how far the forced token's logit lies
    below the sampled one's in THIS model's logits (B, K, V) - the margin a lower-precision run had to overcome to pick it."""
    forced = forced.to(own.dtype)
    assert forced.shape == own.shape, (tuple(forced.shape), tuple(own.shape))
    if trace is not None:
        lg = row_logits.detach()
        pick = lambda ids: torch.gather(lg, -1, ids.permute(0, 2, 1)).permute(0, 2, 1)
        trace.append(dict(own=own.clone(), forced=forced.clone(), gap=pick(own) - pick(forced)))
    return forced


def multi_step_gen_forward(sd, cfg, input_ids, rules, sparse_index, noise_fn=None, forced_ids=None, trace=None):
    """multiStepGenForward (MultiStepPrediction.py:243-332).  noise_fn(shape) -> exponential noise.
    forced_ids: optional list with one (B, n, K) id tensor per sampling call, in call order (see _substitute).
    A rule is (numAlterations, technique, params) or, with the options of SamplingRule (SamplingStrategies.py:53-63),
    (numAlterations, technique, params, {"continue_from_random": bool, "first_beam": (technique, params)})."""
    forced_ids = list(forced_ids) if forced_ids is not None else None
    B, S_len = input_ids.shape
    sparse_index = torch.as_tensor(sparse_index)
    logits, probs, past = M.generator_forward(sd, cfg, input_ids)
    label_p = torch.gather(probs[:, :-1], -1, input_ids[:, 1:].unsqueeze(-1)).squeeze(-1)
    label_p = torch.cat((torch.zeros(B, 1), label_p), dim=-1)

    def sample(p, n, ids, tech, block, params):
        masked_shape = (p.shape[0], p.shape[1], p.shape[2])  # includeFinalTimeStep=True on every tree call
        noise = noise_fn(masked_shape) if tech in (3, 6, 8, 9) else None
        return S.sample_from_generator_distribution(p, n, ids, tech, block_sampling_of_labels=block,
                                                    include_final_time_step=True, noise=noise, **params)

    n0, t0, p0 = rules[0][:3]
    alt, masked = sample(probs, n0, input_ids, t0, True, p0)                    # quirk 5: step 1 blocks the label
    alt_p = S.alt_ids_probs(probs, alt)                                         # quirk 6: unmasked probs
    root = Node(input_ids, probs=label_p)
    root.kv = past
    first = alt[:, :, 1:][:, :, sparse_index]
    first_p = alt_p[:, sparse_index]
    if forced_ids is not None:
        first = _substitute(first, forced_ids.pop(0), logits[:, sparse_index], trace)
        first_p = torch.gather(probs[:, sparse_index], -1, first.permute(0, 2, 1))
    _add_children(root, first, first_p, random_node_info(t0, p0, n0))
    for rule in rules[1:]:
        n, t, p = rule[:3]
        opts = rule[3] if len(rule) > 3 else {}
        rule_n, rule_t, rule_p = n, t, p
        for node in root.leaves(opts.get("continue_from_random", False)):      # SamplingRule's default: random leaves stop
            n, t, p = rule_n, rule_t, rule_p
            if node.beam == 0 and opts.get("first_beam") is not None:           # MultiStepPrediction.py:302-305
                t, p = opts["first_beam"]
                n = rule_n + p["numRandomSamples"]
            pos = (sparse_index + node.depth).unsqueeze(0).expand(B, -1)
            mask = node_mask(node, B, sparse_index, S_len)
            lg, pr, kv = M.generator_forward(sd, cfg, node.ids, position_ids=pos, textual_past=full_past(node), mask=mask)
            labels = torch.cat((node.ids[:, :1], node.ids[:, -1:]), dim=1)
            alt, _ = sample(pr, n, labels, t, False, p)
            node.kv = kv
            if forced_ids is not None:
                alt = torch.cat((alt[:, :, :1], _substitute(alt[:, :, 1:], forced_ids.pop(0), lg, trace)), dim=2)
            _add_children(node, alt[:, :, 1:], S.alt_ids_probs(pr, alt), random_node_info(t, p, n))
    return root, logits, probs, masked


def multi_step_disc_forward(sd, cfg, root, sparse_index):
    """multiStepDiscForward (MultiStepPrediction.py:335-366), inputs detached, DFS over every node."""
    sparse_index = torch.as_tensor(sparse_index)
    B, S_len = root.ids.shape

    def visit(node):
        if node.depth == 0:
            out = M.discriminator_forward(sd, cfg, node.ids.detach())
        else:
            pos = (sparse_index + node.depth).unsqueeze(0).expand(B, -1)
            out = M.discriminator_forward(sd, cfg, node.ids.detach(), position_ids=pos, textual_past=full_past(node),
                                          mask=node_mask(node, B, sparse_index, S_len))
        node.disc_pred, node.in_value, node.alt_value, node.kv = out
        for c in node.children:
            visit(c)

    for n in [root] + root.sub_nodes():
        n.kv = None
    visit(root)
    return root


def sub_node_links(root, nodes):
    """getSubNodeListTreeStructure (MultiStepPrediction.py:212-226)."""
    idx = {id(n): i for i, n in enumerate(nodes)}
    return [idx[id(c)] for c in root.children], [[idx[id(c)] for c in n.children] for n in nodes]
