from .perception import PerceptionAgent, SincConv1d
from .cpea import CorrelationPhaseEstimationAgent
from .msa import MaskSynthesisAgent
from .memory import EpisodicMemory
from .maa import MetacognitiveArbitrationAgent
