"""Import-compatibility alias: `Models/GanSetup.py` of the reference (FreddeFrallan/Branch-GAN) resolves to branchgan.gan.
The module object itself is branchgan.gan (so module globals the reference's callers assign to, e.g. SAMPLE_BUFFER, land in
the real module).  Names: GannSetup."""
import sys

import branchgan.gan as _real

sys.modules[__name__] = _real
