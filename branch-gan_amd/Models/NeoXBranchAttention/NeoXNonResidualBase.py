"""Import-compatibility alias: `Models/NeoXBranchAttention/NeoXNonResidualBase.py` of the reference (FreddeFrallan/Branch-GAN) resolves to branchgan.models.
The module object itself is branchgan.models (so module globals the reference's callers assign to, e.g. SAMPLE_BUFFER, land in
the real module).  Names: GPTNeoXLayer, GPTNeoXModel, GPTNeoXForCausalLM."""
import sys

import branchgan.models as _real

sys.modules[__name__] = _real
