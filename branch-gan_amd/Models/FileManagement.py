"""Import-compatibility alias: `Models/FileManagement.py` of the reference (FreddeFrallan/Branch-GAN) resolves to branchgan.checkpoint.
The module object itself is branchgan.checkpoint (so module globals the reference's callers assign to, e.g. SAMPLE_BUFFER, land in
the real module).  Names: saveGannSetup, loadGannSetup, initModelWeights."""
import sys

import branchgan.checkpoint as _real

sys.modules[__name__] = _real
