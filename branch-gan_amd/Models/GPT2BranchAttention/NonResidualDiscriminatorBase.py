"""Import-compatibility alias: `Models/GPT2BranchAttention/NonResidualDiscriminatorBase.py` of the reference (FreddeFrallan/Branch-GAN) resolves to branchgan.models.
The module object itself is branchgan.models (so module globals the reference's callers assign to, e.g. SAMPLE_BUFFER, land in
the real module).  Names: NonResidualGPT2Block, NonResidualGPT2Model, NonResidualGPT2LMHeadModel."""
import sys

import branchgan.models as _real

sys.modules[__name__] = _real
