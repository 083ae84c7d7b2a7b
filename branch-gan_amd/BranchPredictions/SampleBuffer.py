"""Import-compatibility alias: `BranchPredictions/SampleBuffer.py` of the reference (FreddeFrallan/Branch-GAN) resolves to branchgan.sampling.
The module object itself is branchgan.sampling (so module globals the reference's callers assign to, e.g. SAMPLE_BUFFER, land in
the real module).  Names: SamplingBuffer."""
import sys

import branchgan.sampling as _real

sys.modules[__name__] = _real
