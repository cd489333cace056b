"""MI355X-native (gfx950) hot path of sincformer-metacog-speech-enhancement.

Host side mirrors the reference's module interface (agents/, models/,
training/); arithmetic runs in hand-written HIP kernels behind the C ABI of
csrc/ (include/sincformer_hip.h).  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"
