from .configuration_flamingo import FlamingoConfig
from .flamingo_processor import FlamingoProcessor
from .gated_cross_attention import GatedCrossAttentionBlock, MaskedCrossAttention, ModifiedLMBlock
from .modeling_flamingo import FlamingoBaseModel, FlamingoGPT2, FlamingoModel, FlamingoOPT
from .graphs import GraphedTrainStep
from .optim import FusedAdamW
from .perceiver_resampler import PerceiverAttentionLayer, PerceiverResampler

__all__ = ["FlamingoConfig", "FlamingoModel", "FlamingoProcessor", "FlamingoBaseModel", "FlamingoGPT2", "FlamingoOPT",
           "PerceiverResampler", "PerceiverAttentionLayer", "GatedCrossAttentionBlock", "MaskedCrossAttention", "ModifiedLMBlock",
           "FusedAdamW", "GraphedTrainStep"]
