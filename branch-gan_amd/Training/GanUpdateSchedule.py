"""Import-compatibility alias: `Training/GanUpdateSchedule.py` of the reference (FreddeFrallan/Branch-GAN) resolves to branchgan.schedules.
The module object itself is branchgan.schedules (so module globals the reference's callers assign to, e.g. SAMPLE_BUFFER, land in
the real module).  Names: the seven update schedules."""
import sys

import branchgan.schedules as _real

sys.modules[__name__] = _real
