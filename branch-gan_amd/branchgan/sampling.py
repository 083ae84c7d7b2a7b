"""Per-step branch sampling over the vocabulary — host side.

Mirrors the call surface of the reference's BranchPredictions/SamplingStrategies.py (same names, argument
meaning and errors) on top of the HIP kernels bg_topn_select / bg_topp_select: no argsort, no scatter/cat
copies of the (B, T, V) tensor, ids straight from one streaming pass per row.
Extra keyword `noise=` injects the exponential noise explicitly (the reference's injection point is the module
global SAMPLE_BUFFER, SamplingStrategies.py:5-13, which is kept as well).
"""
from enum import Enum

import numpy as np
import torch

from . import ops

SAMPLE_BUFFER = None


class SamplingBuffer:
    """BranchPredictions/SampleBuffer.py:12-53: exponential noise -log(U + 1e-14) drawn ONCE per shape on the CPU,
    kept on the device, and circularly rolled along every non-batch dim by one of 128 pre-drawn split points
    per call (the reference's cat(noise[split:], noise[:split]) == torch.roll(noise, -split))."""

    def __init__(self, device, numShuffleSplits=128):
        self.device = device
        self.numShuffleSplits = numShuffleSplits
        self.shapeRandBuffers = {}

    def createShapeBufferData(self, shape, epsilon=1e-14):
        splits = [np.random.randint(0, shape[i + 1] - 1, self.numShuffleSplits) for i in range(len(shape) - 1)]
        noise = -torch.log(torch.rand(shape) + epsilon).to(self.device)
        return [noise, splits, 0]

    def rand(self, shape):
        shape = tuple(shape)
        if shape not in self.shapeRandBuffers:
            self.shapeRandBuffers[shape] = self.createShapeBufferData(shape)
        noise, splits, counter = self.shapeRandBuffers[shape]
        dims = tuple(range(1, len(shape)))
        rolled = torch.roll(noise, shifts=tuple(-int(splits[i][counter]) for i in range(len(dims))), dims=dims)
        self.shapeRandBuffers[shape][-1] = (counter + 1) % self.numShuffleSplits
        return rolled


def getRand(size, device):
    global SAMPLE_BUFFER
    if SAMPLE_BUFFER is None:
        SAMPLE_BUFFER = SamplingBuffer(device)
    return SAMPLE_BUFFER.rand(size)


class SamplingTechniques(Enum):  # SamplingStrategies.py:29-42
    RANDOM = 0
    SAMPLE = 1
    TOP_K = 2
    SAMPLE_NO_REPLACEMENT = 3
    EPSILON_GREEDY = 4
    EPSILON_SAMPLE = 5
    EPSILON_SAMPLE_NO_REPLACEMENT = 6
    DEBUG_GRAB_FIRST_IDS = 7
    INVERSE_TOP_K_SAMPLE_NO_REPLACEMENT = 8
    TOP_P_SAMPLE_NO_REPLACEMENT = 9


RANDOM_TECHNIQUES = {
    SamplingTechniques.RANDOM.name: 1,
    SamplingTechniques.EPSILON_GREEDY.name: 1,
    SamplingTechniques.EPSILON_SAMPLE.name: 1,
    SamplingTechniques.EPSILON_SAMPLE_NO_REPLACEMENT.name: 1,
}


class SamplingRule:  # SamplingStrategies.py:53-74
    def __init__(self, numAlterations, samplingStrategy, samplingParams, continueFromRandom=False,
                 firstBeamSamplingStrategy=None, firstBeamSamplingParams=None):
        self.numAlterations = numAlterations
        self.samplingStrategy = samplingStrategy
        self.samplingParams = samplingParams
        self.continueFromRandom = continueFromRandom
        self.firstBeamSamplingStrategy = firstBeamSamplingStrategy
        self.firstBeamSamplingParams = {} if firstBeamSamplingParams is None else firstBeamSamplingParams

    def __str__(self):
        return str(self.to_dict())

    def to_dict(self):
        return {'NumAlterations': self.numAlterations, 'Strategy': self.samplingStrategy.name,
                'Params': self.samplingParams, 'ContinueFromRandom': self.continueFromRandom,
                'FirstBeam SamplingParams': self.firstBeamSamplingParams,
                'FirstBeam SamplingStrategy': 'None' if self.firstBeamSamplingStrategy is None
                else self.firstBeamSamplingStrategy.name}


def apply_temperature(tensor, temperature):
    """SamplingStrategies.py:77-99 (elementwise on the device through torch; not on the shipped config's path)."""
    if temperature == 0:
        raise ValueError("Temperature must be non-zero.")
    if temperature == 1:
        return tensor
    warped = torch.exp(torch.log(tensor) / temperature)
    return warped / torch.sum(warped, dim=-1, keepdim=True)


def apply_noise(tensor, noise_level):
    """SamplingStrategies.py:102-130."""
    if noise_level < 0 or noise_level > 1:
        raise ValueError("Noise level must be between 0 and 1.")
    if noise_level == 0:
        return tensor
    draw = torch.distributions.dirichlet.Dirichlet(torch.ones_like(tensor) * noise_level).sample()
    mixed = tensor * (1 - noise_level) + draw * noise_level
    return mixed / torch.sum(mixed, dim=-1, keepdim=True)


class MaskedProbs:
    """Lazy stand-in for the reference's `maskedProbs` return value (SamplingStrategies.py:258-265): nothing on the
    hot path reads it, so the scatter + cat copies of the (B, T, V) tensor only happen if `.materialize()` is called."""

    def __init__(self, probs, input_ids, block, include_final):
        self.probs, self.input_ids, self.block, self.include_final = probs, input_ids, block, include_final
        self.shape = (probs.shape[0], probs.shape[1] if include_final else probs.shape[1] - 1, probs.shape[2])

    def materialize(self):
        p, ids = self.probs, self.input_ids
        if hasattr(p, "materialize"):
            p = p.materialize()
        if self.block:
            head = torch.scatter(p[:, :-1], dim=-1, index=ids[:, 1:].unsqueeze(-1), value=0)
            return torch.cat((head, p[:, -1:]), dim=1) if self.include_final else head
        return p if self.include_final else p[:, :-1]


def _row_selection(B, T, V, input_ids, block, include_final, device):
    """Which rows of the flattened (B*T, V) tensor are sampled and which id each one blocks."""
    T_out = T if include_final else T - 1
    row_map = None
    if not include_final:
        row_map = (torch.arange(B, device=device).unsqueeze(1) * T + torch.arange(T_out, device=device).unsqueeze(0)).reshape(-1)
    blocked = None
    if block:
        nxt = input_ids[:, 1:T]                                           # label of row t is ids[t + 1]
        if include_final:
            nxt = torch.cat((nxt, torch.full((B, 1), -1, dtype=nxt.dtype, device=device)), dim=1)
        blocked = nxt.reshape(-1, 1)
    return T_out, row_map, blocked


def _random_ids(B, n, T_out, V, device):
    return torch.randint(0, V, (B, n, T_out)).to(device)                  # CPU global RNG like the reference (:135-136)


def sampleFromGeneratorDistribution(probs, numSamples, inputIDs, samplingTechnique, blockSamplingOfLabels=True,
                                    includeFinalTimeStep=False, noise=None, **samplingTechniqueParams):
    """SamplingStrategies.py:248-294.  probs (B, T, V) fp32 on the GPU; returns (alterationSequences int64
    (B, n, T'+1), MaskedProbs).  `noise` (B, T', V): exponential noise for the no-replacement techniques
    (default: getRand, the rolled per-shape buffer)."""
    tech = samplingTechnique if isinstance(samplingTechnique, SamplingTechniques) else SamplingTechniques(int(getattr(samplingTechnique, "value", samplingTechnique)))
    if 'temperature' in samplingTechniqueParams:
        probs = apply_temperature(probs, samplingTechniqueParams['temperature'])
    if 'noise_level' in samplingTechniqueParams:
        probs = apply_noise(probs, samplingTechniqueParams['noise_level'])
    B, T, V = probs.shape
    dev = probs.device
    T_out, row_map, blocked = _row_selection(B, T, V, inputIDs, blockSamplingOfLabels, includeFinalTimeStep, dev)
    flat = probs.reshape(B * T, V)
    n_rand = samplingTechniqueParams.get('numRandomSamples', 4)

    def get_noise():
        nz = noise if noise is not None else getRand((B, T_out, V), dev)
        return nz.reshape(B * T_out, V)

    def select(n, nz=None, extra_blocked=None):
        blk = blocked
        if extra_blocked is not None:
            blk = extra_blocked if blk is None else torch.cat((blk, extra_blocked), dim=1)
        ids = ops.topn_select(flat, n, noise=nz, blocked=blk, blocked_value=0.0, row_map=row_map)
        return ids.view(B, T_out, n).permute(0, 2, 1)                     # (B, n, T')

    T_ = SamplingTechniques
    if tech == T_.RANDOM:
        samples = _random_ids(B, numSamples, T_out, V, dev)
    elif tech == T_.SAMPLE:
        masked = MaskedProbs(probs, inputIDs, blockSamplingOfLabels, includeFinalTimeStep).materialize()
        samples = torch.distributions.Categorical(masked).sample((numSamples,)).permute(1, 0, 2)
    elif tech == T_.TOP_K:
        samples = select(numSamples)
    elif tech == T_.EPSILON_GREEDY:
        samples = torch.cat((select(numSamples - n_rand), _random_ids(B, n_rand, T_out, V, dev)), dim=1)
    elif tech == T_.EPSILON_SAMPLE:
        masked = MaskedProbs(probs, inputIDs, blockSamplingOfLabels, includeFinalTimeStep).materialize()
        drawn = torch.distributions.Categorical(masked).sample((numSamples - n_rand,)).permute(1, 0, 2)
        samples = torch.cat((drawn, _random_ids(B, n_rand, T_out, V, dev)), dim=1)
    elif tech == T_.SAMPLE_NO_REPLACEMENT:
        samples = select(numSamples, get_noise())
    elif tech == T_.EPSILON_SAMPLE_NO_REPLACEMENT:
        samples = torch.cat((select(numSamples - n_rand, get_noise()), _random_ids(B, n_rand, T_out, V, dev)), dim=1)
    elif tech == T_.INVERSE_TOP_K_SAMPLE_NO_REPLACEMENT:
        top = ops.topn_select(flat, samplingTechniqueParams.get('inverseTopK', 3), blocked=blocked, row_map=row_map)
        samples = select(numSamples, get_noise(), extra_blocked=top)      # :236-240: the top-k are zeroed, then the race
    elif tech == T_.TOP_P_SAMPLE_NO_REPLACEMENT:
        rows = flat if row_map is None else flat[row_map]
        ids = ops.topp_select(rows, numSamples, get_noise(), samplingTechniqueParams.get('topP', 0.5), blocked=blocked)
        samples = ids.view(B, T_out, numSamples).permute(0, 2, 1)
    else:
        raise NotImplementedError()

    first = inputIDs[:, :1].unsqueeze(1).expand(-1, numSamples, -1)
    alterationSequences = torch.cat((first, samples), dim=-1)
    return alterationSequences, MaskedProbs(probs, inputIDs, blockSamplingOfLabels, includeFinalTimeStep)


def sampleWithSamplingRule(probs, inputIDs, samplingRule, blockSamplingOfLabels=True, includeFinalTimeStep=False):
    return sampleFromGeneratorDistribution(probs, samplingRule.numAlterations, inputIDs, samplingRule.samplingStrategy,
                                           blockSamplingOfLabels, includeFinalTimeStep, **samplingRule.samplingParams)
