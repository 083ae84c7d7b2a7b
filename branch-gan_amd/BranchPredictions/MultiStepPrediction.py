"""Import-compatibility alias: `BranchPredictions/MultiStepPrediction.py` of the reference (FreddeFrallan/Branch-GAN) resolves to branchgan.tree.
The module object itself is branchgan.tree (so module globals the reference's callers assign to, e.g. SAMPLE_BUFFER, land in
the real module).  Names: PredictionTreeNode, multiStepGenForward, multiStepDiscForward, clearTree, getSubNodeListTreeStructure, getRandomNodesFromRule."""
import sys

import branchgan.tree as _real

sys.modules[__name__] = _real
