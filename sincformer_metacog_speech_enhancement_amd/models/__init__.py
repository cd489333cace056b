from .vq import VectorQuantizer, VQMaskQuantizer
