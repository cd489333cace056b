"""Batched, on-device counterparts of the reference's evaluation/ssnr.py and the fallback of evaluation/stoi.py."""
from .ssnr import compute_ssnr, compute_ssnr_improvement  # noqa: F401
from .stoi import compute_stoi  # noqa: F401
