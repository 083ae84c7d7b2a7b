"""Config objects of the hot path: model dims, sampling config, reinforcement-learning config, depth schedule.

Mirrors Configs/TrainingConfig.py:135-212 (SamplingConfig, RoundRobinSamplingConfig), Configs/ReinforceLabelMode.py
and Training/DepthSchedule.py of the reference (same names / fields / semantics).  Model dims are local
presets (the reference resolves them with AutoConfig.from_pretrained(<name>), a network fetch that is never run
here).
"""
import copy
import enum
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class LMConfig:
    model_type: str                 # "gpt_neox" | "gpt2"
    vocab_size: int
    hidden_size: int
    num_hidden_layers: int
    num_attention_heads: int
    max_position_embeddings: int
    intermediate_size: int = 0
    rotary_pct: float = 0.25
    rotary_emb_base: float = 10000.0
    layer_norm_eps: float = 1e-5
    use_parallel_residual: bool = True
    scale_attn_weights: bool = True
    scale_attn_by_inverse_layer_idx: bool = False
    initializer_range: float = 0.02
    attn_pdrop: float = 0.0          # dropout on the attention probabilities: inside the attention kernels (Dh <= 128)
    resid_pdrop: float = 0.0
    embd_pdrop: float = 0.0

    def __post_init__(self):
        if not self.intermediate_size:
            self.intermediate_size = 4 * self.hidden_size
        if self.hidden_size % self.num_attention_heads:
            raise ValueError("The hidden size is not divisble by the number of attention heads! Make sure to update them")

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads

    @property
    def rotary_ndims(self):
        return int(self.head_dim * self.rotary_pct)

    @staticmethod
    def from_any(cfg):
        """Accepts an LMConfig, a preset name, or a HuggingFace GPTNeoXConfig / GPT2Config-like object."""
        if isinstance(cfg, LMConfig):
            return cfg
        if isinstance(cfg, str):
            return copy.deepcopy(PRESETS[cfg])
        g = lambda *names, default=None: next((getattr(cfg, n) for n in names if getattr(cfg, n, None) is not None), default)
        mt = g("model_type")
        if mt == "gpt_neox":
            return LMConfig("gpt_neox", g("vocab_size"), g("hidden_size"), g("num_hidden_layers"), g("num_attention_heads"),
                            g("max_position_embeddings"), g("intermediate_size", default=0), g("rotary_pct", default=0.25),
                            g("rotary_emb_base", default=10000.0), g("layer_norm_eps", default=1e-5),
                            g("use_parallel_residual", default=True), initializer_range=g("initializer_range", default=0.02),
                            attn_pdrop=g("attention_dropout", default=0.0), resid_pdrop=g("hidden_dropout", default=0.0))
        return LMConfig("gpt2", g("vocab_size"), g("hidden_size", "n_embd"), g("num_hidden_layers", "n_layer"),
                        g("num_attention_heads", "n_head"), g("max_position_embeddings", "n_positions"),
                        g("n_inner", default=0) or 0, layer_norm_eps=g("layer_norm_epsilon", default=1e-5),
                        scale_attn_weights=g("scale_attn_weights", default=True),
                        scale_attn_by_inverse_layer_idx=g("scale_attn_by_inverse_layer_idx", default=False),
                        initializer_range=g("initializer_range", default=0.02), attn_pdrop=g("attn_pdrop", default=0.0),
                        resid_pdrop=g("resid_pdrop", default=0.0), embd_pdrop=g("embd_pdrop", default=0.0))


    def to_dict(self):
        """HuggingFace key names of the family (what `config.to_json_file` of the reference's GPTNeoXConfig / GPT2Config
        writes for these fields), so a `Generator-Config.json` written here reads back there and vice versa."""
        if self.model_type == "gpt_neox":
            return dict(model_type="gpt_neox", vocab_size=self.vocab_size, hidden_size=self.hidden_size, num_hidden_layers=self.num_hidden_layers,
                        num_attention_heads=self.num_attention_heads, max_position_embeddings=self.max_position_embeddings,
                        intermediate_size=self.intermediate_size, rotary_pct=self.rotary_pct, rotary_emb_base=self.rotary_emb_base,
                        layer_norm_eps=self.layer_norm_eps, use_parallel_residual=self.use_parallel_residual,
                        initializer_range=self.initializer_range, attention_dropout=self.attn_pdrop, hidden_dropout=self.resid_pdrop)
        return dict(model_type="gpt2", vocab_size=self.vocab_size, n_embd=self.hidden_size, n_layer=self.num_hidden_layers,
                    n_head=self.num_attention_heads, n_positions=self.max_position_embeddings, n_inner=self.intermediate_size,
                    layer_norm_epsilon=self.layer_norm_eps, scale_attn_weights=self.scale_attn_weights,
                    scale_attn_by_inverse_layer_idx=self.scale_attn_by_inverse_layer_idx, initializer_range=self.initializer_range,
                    attn_pdrop=self.attn_pdrop, resid_pdrop=self.resid_pdrop, embd_pdrop=self.embd_pdrop)

    def to_json_file(self, path):
        import json
        with open(path, "w") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True)

    @staticmethod
    def from_json_file(path):
        import json
        import types
        with open(path) as f:
            return LMConfig.from_any(types.SimpleNamespace(**json.load(f)))


# public model-card dims (SURVEY.md §8 header); dropout 0 as in every parity / benchmark run
PRESETS = {
    "gpt2-small": LMConfig("gpt2", 50257, 768, 12, 12, 1024),
    "gpt2-tiny": LMConfig("gpt2", 2048, 128, 2, 2, 256),          # not a published model: rehearsals of the multi-rank path
    "pythia-14m": LMConfig("gpt_neox", 50304, 128, 6, 4, 2048),
    "pythia-70m": LMConfig("gpt_neox", 50304, 512, 6, 8, 2048),
    "pythia-410m": LMConfig("gpt_neox", 50304, 1024, 24, 16, 2048),
    "pythia-1b": LMConfig("gpt_neox", 50304, 2048, 16, 8, 2048),
}


class ReinforcementLabelMode(enum.IntEnum):
    ALTERATIONS_ONLY = 0
    INPUT_ONLY = 1
    ALTERATIONS_AND_INPUT = 2
    ALTERATIONS_AND_INPUT_UNWEIGHTED = 3
    ALTERATIONS_AND_INPUT_TEACHER_FORCING = 4
    ALTERATIONS_AND_INPUT_UNWEIGHTED_TEACHER_FORCING = 5


class IncorrectValueHeadsPolicy(enum.IntEnum):
    DO_NOTHING = 0
    IGNORE_SAMPLE = 1
    INPUT_HEAD_ONLY_SECTION_A = 2
    INPUT_HEAD_ONLY_SECTION_C = 3
    INPUT_HEAD_ONLY_SECTION_A_AND_C = 4
    ALTERATION_HEAD_ONLY_SECTION_A = 5
    ALTERATION_HEAD_ONLY_SECTION_C = 6
    ALTERATION_HEAD_ONLY_SECTION_A_AND_C = 7


class ReinforcementLearningConfig:
    class AdvantageSectionWeights:
        def __init__(self, A=1, B=1, C=1):
            self.A, self.B, self.C = A, B, C

    def __init__(self, labelMode=ReinforcementLabelMode.ALTERATIONS_AND_INPUT,
                 inputIncorrectValueHeadsPolicy=IncorrectValueHeadsPolicy.IGNORE_SAMPLE,
                 alterationIncorrectValueHeadsPolicy=IncorrectValueHeadsPolicy.IGNORE_SAMPLE,
                 sectionWeightA=1, sectionWeightB=1, sectionWeightC=1, teacherForcing=True):
        self.labelMode = labelMode
        self.teacherForcing = teacherForcing
        self.inputIncorrectValueHeadsPolicy = inputIncorrectValueHeadsPolicy
        self.alterationIncorrectValueHeadsPolicy = alterationIncorrectValueHeadsPolicy
        self.advantageSectionWeights = self.AdvantageSectionWeights(sectionWeightA, sectionWeightB, sectionWeightC)

    def __str__(self):
        w = self.advantageSectionWeights
        return "A: {}, B: {}, C: {}".format(w.A, w.B, w.C)


class SamplingConfig:
    """Configs/TrainingConfig.py:135-191."""

    def __init__(self, samplingRules, discDepthLossSchedule=None, includeAllMultiStepSamplesForDisc=True,
                 separateInputAltLoss=True, inputLossWeight=1, discpDepthLossWeight=None, numSampleSequences=None,
                 sparseIndex=None, seed=None):
        self.rules = samplingRules if isinstance(samplingRules, list) else [samplingRules]
        self.numAlterations = int(np.prod([r.numAlterations for r in self.rules])) if len(self.rules) > 1 else self.rules[0].numAlterations
        self.multiStepSampling = True
        self.includeAllMultiStepSamplesForDisc = includeAllMultiStepSamplesForDisc
        self.discDepthLossSchedule = discDepthLossSchedule
        self.discpDepthLossWeight = discpDepthLossWeight
        self.separateInputAltLoss = separateInputAltLoss
        self.inputLossWeight = inputLossWeight
        self.numSampleSequences = numSampleSequences
        if sparseIndex is not None and not isinstance(sparseIndex, torch.Tensor):
            sparseIndex = torch.tensor(sparseIndex)
        self.sparseIndex = sparseIndex
        self._sampleIndex = None
        # the reference draws from the unseeded global np.random; a per-rank seed makes runs reproducible
        self._rng = np.random.RandomState(seed) if seed is not None else np.random

    def getNextSampleConf(self):
        return self

    def getSparseIndex(self, ids, device):
        if self.sparseIndex is not None:
            return self.sparseIndex
        if self.numSampleSequences is None or self.numSampleSequences == ids.shape[1]:
            return None
        idx = self._rng.choice(ids.shape[1] - 1, self.numSampleSequences, replace=False)
        self._sampleIndex = torch.tensor(np.sort(idx), device=device)
        return self._sampleIndex

    def to_dict(self):
        data = {'IncludeAllMultiStepSamplesForDisc': self.includeAllMultiStepSamplesForDisc,
                'DiscDepthLossSchedule': None, 'separateInputAltLoss': self.separateInputAltLoss,
                'inputLossWeight': self.inputLossWeight, 'numSampleSequences': self.numSampleSequences}
        for i, rule in enumerate(self.rules):
            data['Rule-{}'.format(i)] = rule.to_dict()
        return data

    def __repr__(self):
        return str(self.to_dict())


class RoundRobinSamplingConfig:
    """Configs/TrainingConfig.py:194-212."""

    def __init__(self, samplingConfs):
        self.samplingConfs = samplingConfs
        self.sampleCounter = 0

    def getNextSampleConf(self):
        self.sampleCounter = (self.sampleCounter + 1) % len(self.samplingConfs)
        return self.samplingConfs[self.sampleCounter]


class DepthSchedule:
    """Training/DepthSchedule.py:3-22 (the copy keeps the RNG object shared, unlike a blind deepcopy)."""

    def __init__(self, startDepth):
        self.currentDepth = startDepth

    def updateSchedule(self, epoch, batch, totalUpdates):
        raise NotImplementedError("DepthSchedule is an abstract class. Please use a concrete implementation.")

    def setSampleConfDepth(self, sampleConf):
        conf = copy.copy(sampleConf)
        conf.rules = list(sampleConf.rules[:self.currentDepth])
        if self.currentDepth > 1:
            conf.multiStepSampling = True
        return conf


class LinearEpochDepthSchedule(DepthSchedule):
    def __init__(self, startDepth, incrementSize, epochsPerIncrement, maxDepth=None):
        super().__init__(startDepth)
        self.startDepth, self.incrementSize, self.epochsPerIncrement, self.maxDepth = startDepth, incrementSize, epochsPerIncrement, maxDepth

    def updateSchedule(self, epoch, batch, totalUpdates):
        self.currentDepth = int(self.startDepth + (epoch // self.epochsPerIncrement) * self.incrementSize)
        if self.maxDepth is not None:
            self.currentDepth = min(self.currentDepth, self.maxDepth)


class LinearUpdatesDepthSchedule(DepthSchedule):
    def __init__(self, startDepth, incrementSize, updatesPerIncrement, maxDepth):
        super().__init__(startDepth)
        self.startDepth, self.incrementSize, self.updatesPerIncrement, self.maxDepth = startDepth, incrementSize, updatesPerIncrement, maxDepth

    def updateSchedule(self, epoch, batch, totalUpdates):
        self.currentDepth = int(self.startDepth + (totalUpdates // self.updatesPerIncrement) * self.incrementSize)
        if self.maxDepth is not None:
            self.currentDepth = min(self.currentDepth, self.maxDepth)
