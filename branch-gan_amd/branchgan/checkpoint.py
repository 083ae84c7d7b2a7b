"""Checkpoints in the reference's folder layout, so runs move between the two code bases:

    <mainSavePath>/<runName with '/' -> '-'>/
        Generator-Config.json  Discriminator-Config.json  [Dataset-Config.json]  [Tokenizer.json]     written once
        <timeMarker>/Generator/Generator.pt          state dict of `model.generator`, reference key names
        <timeMarker>/Generator/Optimizer.pt          torch.optim.Adam state-dict layout (ArenaAdam writes the same)
        <timeMarker>/Discriminator/{Discriminator.pt, Optimizer.pt}

Reference: Training/Utils.py:51-117 (`saveGannSetup`, checkpoint discovery), Models/FileManagement.py:5-36 (start weights,
'module.' prefixes of DistributedDataParallel stripped on load).  Files are read with `weights_only=True`: a checkpoint is
data, nothing in it is executed.  The reference's persistent mask buffers (`attn.bias`, `attn.self_bias`, `masked_bias`,
`rotary_emb.inv_freq`: position-squared uint8 tables this build evaluates arithmetically) are dropped on load and can be
regenerated on save (`referenceBuffers=True`) for a loader that insists on them.
"""
import os

import torch

from . import models

_WEIGHTS = {'Generator': 'Generator.pt', 'Discriminator': 'Discriminator.pt'}


def loadStateDict(path):
    """A state dict from disk on the CPU, 'module.' (DDP) prefixes removed (Utils.py:97-99, FileManagement.py:5-7)."""
    sd = torch.load(path, map_location='cpu', weights_only=True)
    return {k.replace('module.', ''): v for k, v in sd.items()}


def _write_once(path, config):
    if config is not None and not os.path.exists(path):
        config.to_json_file(path)


def saveGannSetup(model, runName, timeMarker, discOptimizer=None, genOptimizer=None, mainSavePath='', datasetConf=None,
                  trainingConf=None, tokenizer=None, saveDisc=True, saveGen=True, saveOptimizers=True, referenceBuffers=False):
    """-> the checkpoint folder.  `discOptimizer` / `genOptimizer` are OptimizerWrappers (or bare optimizers)."""
    root = os.path.join(mainSavePath, runName.replace('/', '-'))
    os.makedirs(root, exist_ok=True)
    _write_once(os.path.join(root, 'Discriminator-Config.json'), getattr(model, 'discriminatorConfig', None))
    _write_once(os.path.join(root, 'Generator-Config.json'), getattr(model, 'generatorConfig', None))
    _write_once(os.path.join(root, 'Dataset-Config.json'), datasetConf)
    if tokenizer is not None:
        tokenizer.save(os.path.join(root, 'Tokenizer.json'))
    here = os.path.join(root, timeMarker)
    for part, module, optim, wanted in (('Generator', model.generator, genOptimizer, saveGen),
                                        ('Discriminator', model.discriminator, discOptimizer, saveDisc)):
        folder = os.path.join(here, part)
        os.makedirs(folder, exist_ok=True)
        if not wanted:
            continue
        sd = {k: v.detach().cpu() for k, v in module.state_dict().items()}
        if referenceBuffers:
            sd.update(models.reference_buffers(module))
        torch.save(sd, os.path.join(folder, _WEIGHTS[part]))
        if saveOptimizers and optim is not None:
            inner = getattr(optim, 'optimizer', optim)
            torch.save(inner.state_dict(), os.path.join(folder, 'Optimizer.pt'))
    return here


def loadGannSetup(model, checkpointFolder, discOptimizer=None, genOptimizer=None, strict=True):
    """Loads what `saveGannSetup` (here or in the reference) wrote under <run>/<timeMarker>."""
    from . import fused
    for part, module, optim in (('Generator', model.generator, genOptimizer), ('Discriminator', model.discriminator, discOptimizer)):
        path = os.path.join(checkpointFolder, part, _WEIGHTS[part])
        if os.path.exists(path):
            module.load_state_dict(models.strip_reference_buffers(loadStateDict(path)), strict=strict)
        opath = os.path.join(checkpointFolder, part, 'Optimizer.pt')
        if optim is not None and os.path.exists(opath):
            getattr(optim, 'optimizer', optim).load_state_dict(torch.load(opath, map_location='cpu', weights_only=True))
    fused.invalidate_weights()              # low-precision weight copies follow the new parameters
    return model


def initModelWeights(model, genStartWeights=None, discStartWeights=None, setupStartWeights=None):
    """FileManagement.py:25-34: optional start weights for either model, or one state dict of the whole GannSetup."""
    from . import fused
    if genStartWeights is not None:
        model.generator.load_state_dict(models.strip_reference_buffers(loadStateDict(genStartWeights)))
    if discStartWeights is not None:
        model.discriminator.load_state_dict(models.strip_reference_buffers(loadStateDict(discStartWeights)))
    if setupStartWeights is not None:
        model.load_state_dict(torch.load(setupStartWeights, map_location='cpu', weights_only=True))
    fused.invalidate_weights()
    return model


def getAvailablCheckpointsFromModelFolder(modelFolder):
    """Utils.py:102-117: sub-folders holding BOTH weight files, 'Epoch-<x>' ones by x, every other name ('Current') last."""
    found = []
    for name in os.listdir(modelFolder):
        sub = os.path.join(modelFolder, name)
        if os.path.isdir(sub) and all(os.path.exists(os.path.join(sub, part, f)) for part, f in _WEIGHTS.items()):
            found.append(name)
    return sorted(found, key=lambda n: float(n.split('-')[-1]) if 'Epoch-' in n else 999999)
