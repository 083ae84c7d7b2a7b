"""Oracle: per-step branch sampling over the vocabulary (test infrastructure, see oracle/__init__.py).

Restates BranchPredictions/SamplingStrategies.py (reference) with the noise passed explicitly instead
of through the module-global SAMPLE_BUFFER (SamplingStrategies.py:5-13 is only an injection point).
Technique numbers are the reference enum values (SamplingStrategies.py:29-42).
"""
import torch

RANDOM, SAMPLE, TOP_K, SAMPLE_NO_REPLACEMENT = 0, 1, 2, 3
EPSILON_GREEDY, EPSILON_SAMPLE, EPSILON_SAMPLE_NO_REPLACEMENT = 4, 5, 6
DEBUG_GRAB_FIRST_IDS, INVERSE_TOP_K_SAMPLE_NO_REPLACEMENT, TOP_P_SAMPLE_NO_REPLACEMENT = 7, 8, 9


def exponential_noise(shape, generator=None, epsilon=1e-14):
    """e = -log(U + 1e-14), U ~ torch.rand on the CPU (SampleBuffer.py:28)."""
    return -torch.log(torch.rand(tuple(shape), generator=generator) + epsilon)


def rolled_noise(noise, splits, counter):
    """SampleBuffer.SamplingBuffer.getShapeRandBuffers (SampleBuffer.py:31-43): the stored noise with every non-batch dim i + 1
    cut at splits[i][counter] and its two pieces swapped."""
    for i in range(noise.dim() - 1):
        cut = int(splits[i][counter])
        noise = torch.cat((noise.narrow(i + 1, cut, noise.shape[i + 1] - cut), noise.narrow(i + 1, 0, cut)), dim=i + 1)
    return noise


def apply_temperature(probs, temperature):
    """SamplingStrategies.py:77-99: exp(log p / T), renormalised over the vocabulary."""
    if temperature == 0:
        raise ValueError("Temperature must be non-zero.")
    if temperature == 1:
        return probs
    warped = torch.exp(torch.log(probs) / temperature)
    return warped / warped.sum(dim=-1, keepdim=True)


def apply_noise(probs, noise_level):
    """SamplingStrategies.py:102-130: blend with a Dirichlet(noise_level) draw (torch global RNG)."""
    if noise_level < 0 or noise_level > 1:
        raise ValueError("Noise level must be between 0 and 1.")
    if noise_level == 0:
        return probs
    draw = torch.distributions.dirichlet.Dirichlet(torch.ones_like(probs) * noise_level).sample()
    mixed = probs * (1 - noise_level) + draw * noise_level
    return mixed / mixed.sum(dim=-1, keepdim=True)


def block_labels(probs, input_ids, block, include_final):
    """SamplingStrategies.py:258-265: zero the true next token's probability, choose the row range."""
    if block:
        head = torch.scatter(probs[:, :-1], dim=-1, index=input_ids[:, 1:].unsqueeze(-1), value=0)
        return torch.cat((head, probs[:, -1:]), dim=1) if include_final else head
    return probs if include_final else probs[:, :-1]


def top_n(keys, n, specified=False):
    """argsort(descending)[..., :n] -> (B, n, T)  (SamplingStrategies.py:197-205,156-158).

    specified=False calls argsort exactly as the reference does (stable=False): the order of EQUAL keys
    (and of NaN keys, which compare equal to each other and above everything) is then whatever the ATen
    CPU introsort leaves — an artefact torch documents as unspecified (an all-NaN 97-wide row yields id 60).
    specified=True is the tie rule the build's contract fixes instead: equal keys in index order, NaN first.
    The two agree whenever the n+1 largest keys are distinct, which is what bit-exact parity is defined on.
    """
    order = torch.argsort(keys, dim=-1, descending=True, stable=bool(specified))
    return order[:, :, :n].permute(0, 2, 1)


def sample_without_replacement(masked, n, noise, specified=False):
    """Exponential-race / Gumbel-top-k: keys = p / e (SamplingStrategies.py:149-158)."""
    return top_n(masked / noise, n, specified)


def top_p_filter(probs, p, specified=False):
    """SamplingStrategies.py:161-182.  Keeps sorted position i iff cumsum[i+1] <= p (the reference
    shifts its mask left by one, so the last token that fits is dropped); renormalises.  A row whose
    top-1 mass exceeds p keeps nothing and becomes 0/0 = NaN.
    specified=True sorts stably (equal probabilities in index order): which of several EQUAL probabilities
    straddling the cut are kept is otherwise an artefact of ATen's unstable sort (see top_n)."""
    sorted_p, sorted_i = torch.sort(probs, dim=-1, descending=True, stable=bool(specified))
    keep = torch.cumsum(sorted_p, dim=-1) <= p
    keep = torch.cat((keep[..., 1:], torch.zeros_like(keep[..., :1])), dim=-1)
    keep = torch.gather(keep, dim=-1, index=sorted_i.argsort())
    kept = probs * keep.float()
    return kept / kept.sum(dim=-1, keepdim=True)


def inverse_top_k_filter(masked, top_k):
    """SamplingStrategies.py:236-239: zero the top_k largest entries."""
    order = torch.argsort(masked, dim=-1, descending=True)
    return torch.scatter(masked, dim=-1, index=order[:, :, :top_k], value=0)


def random_ids(masked, n):
    """SamplingStrategies.py:133-146 (CPU randint; label blocking is a TODO in the reference)."""
    return torch.randint(0, masked.shape[-1], (masked.shape[0], n, masked.shape[1]))


def categorical_ids(masked, n):
    """SamplingStrategies.py:190-194."""
    return torch.distributions.Categorical(masked).sample((n,)).permute(1, 0, 2)


def sample_from_generator_distribution(probs, num_samples, input_ids, technique, block_sampling_of_labels=True,
                                       include_final_time_step=False, noise=None, **params):
    """SamplingStrategies.py:248-294.  `noise` replaces getRand(maskedProbs.shape) for the techniques that
    consume it; returns (alterationSequences int64 (B, n, T'+1), maskedProbs (B, T', V))."""
    technique = int(getattr(technique, "value", technique))
    if "temperature" in params:
        probs = apply_temperature(probs, params["temperature"])
    if "noise_level" in params:
        probs = apply_noise(probs, params["noise_level"])
    masked = block_labels(probs, input_ids, block_sampling_of_labels, include_final_time_step)

    def need_noise():
        if noise is None:
            raise ValueError("this technique consumes exponential noise: pass noise=")
        return noise

    n_rand = params.get("numRandomSamples", 4)
    if technique == RANDOM:
        samples = random_ids(masked, num_samples)
    elif technique == SAMPLE:
        samples = categorical_ids(masked, num_samples)
    elif technique == TOP_K:
        samples = top_n(masked, num_samples)
    elif technique == EPSILON_GREEDY:
        samples = torch.cat((top_n(masked, num_samples - n_rand), random_ids(masked, n_rand)), dim=1)
    elif technique == EPSILON_SAMPLE:
        samples = torch.cat((categorical_ids(masked, num_samples - n_rand), random_ids(masked, n_rand)), dim=1)
    elif technique == SAMPLE_NO_REPLACEMENT:
        samples = sample_without_replacement(masked, num_samples, need_noise())
    elif technique == EPSILON_SAMPLE_NO_REPLACEMENT:
        samples = torch.cat((sample_without_replacement(masked, num_samples - n_rand, need_noise()),
                             random_ids(masked, n_rand)), dim=1)
    elif technique == INVERSE_TOP_K_SAMPLE_NO_REPLACEMENT:
        samples = sample_without_replacement(inverse_top_k_filter(masked, params.get("inverseTopK", 3)),
                                             num_samples, need_noise())
    elif technique == TOP_P_SAMPLE_NO_REPLACEMENT:
        samples = sample_without_replacement(top_p_filter(masked, params.get("topP", 0.5)), num_samples, need_noise())
    else:
        raise NotImplementedError()

    first = input_ids[:, :1].unsqueeze(1).expand(-1, num_samples, -1)
    return torch.cat((first, samples.to(input_ids.device)), dim=-1), masked


def alt_ids_probs(probs, alt_ids):
    """MultiStepPrediction.py:238-240: probs of the sampled ids, (B, T, n)."""
    return torch.gather(probs, dim=-1, index=alt_ids.permute(0, 2, 1)[:, 1:])
