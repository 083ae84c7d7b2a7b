"""Import-compatibility alias: `Models/Losses/AdvantageCalculations.py` of the reference (FreddeFrallan/Branch-GAN) resolves to branchgan.losses.
The module object itself is branchgan.losses (so module globals the reference's callers assign to, e.g. SAMPLE_BUFFER, land in
the real module).  Names: dualHeadAdvantages."""
import sys

import branchgan.losses as _real

sys.modules[__name__] = _real
