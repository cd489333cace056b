"""Constants of the reference's config.py that the hot path reads
(config.py:17-21 audio framing, :93-98 Conformer, :105-107 agents, :25 channels).
Only values are mirrored; nothing else of the reference configuration exists here."""
import os as _os

# checkpoints (config.py:13 of the reference: <repo>/saved_models); override with SFM_MODEL_DIR
MODEL_DIR = _os.environ.get("SFM_MODEL_DIR") or _os.path.join(_os.getcwd(), "saved_models")

SAMPLE_RATE = 8000
FRAME_SIZE_MS = 20
FRAME_SIZE = int(SAMPLE_RATE * FRAME_SIZE_MS / 1000)   # 160
HOP_SIZE = FRAME_SIZE // 2                              # 80
FFT_SIZE = 256
NUM_CHANNELS = 64
CONFORMER_NUM_BLOCKS = 6
CONFORMER_D_MODEL = 256
CONFORMER_NUM_HEADS = 4
CONFORMER_FF_DIM = 1024
CONFORMER_KERNEL_SIZE = 31
CONFORMER_DROPOUT = 0.1
CPEA_HIDDEN_SIZE = 128
CPEA_NUM_LAYERS = 2
PA_ENCODER_CHANNELS = 256

# SURVEY 8f N4 (config.py:101-108 of the reference)
VQ_NUM_CENTROIDS = 3
VQ_COMMITMENT_WEIGHT = 0.25
MAA_THRESHOLD_INIT = 0.5
