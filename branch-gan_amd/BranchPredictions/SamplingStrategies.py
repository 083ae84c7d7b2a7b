"""Import-compatibility alias: `BranchPredictions/SamplingStrategies.py` of the reference (FreddeFrallan/Branch-GAN) resolves to branchgan.sampling.
The module object itself is branchgan.sampling (so module globals the reference's callers assign to, e.g. SAMPLE_BUFFER, land in
the real module).  Names: SamplingTechniques, SamplingRule, RANDOM_TECHNIQUES, sampleFromGeneratorDistribution, sampleWithSamplingRule, getRand, SAMPLE_BUFFER, apply_temperature, apply_noise."""
import sys

import branchgan.sampling as _real

sys.modules[__name__] = _real
