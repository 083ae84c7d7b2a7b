"""Import-compatibility alias: `Training/TrainingLoop.py` of the reference (FreddeFrallan/Branch-GAN) resolves to branchgan.training.
The module object itself is branchgan.training (so module globals the reference's callers assign to, e.g. SAMPLE_BUFFER, land in
the real module).  Names: efficentTrainingLoop, forwardPass, validateModel, logResults."""
import sys

import branchgan.training as _real

sys.modules[__name__] = _real
