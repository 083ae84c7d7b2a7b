"""Import-compatibility alias: `Configs/TrainingConfig.py` of the reference (FreddeFrallan/Branch-GAN) resolves to branchgan.configs.
The module object itself is branchgan.configs (so module globals the reference's callers assign to, e.g. SAMPLE_BUFFER, land in
the real module).  Names: SamplingConfig, RoundRobinSamplingConfig, ReinforcementLearningConfig."""
import sys

import branchgan.configs as _real

sys.modules[__name__] = _real
