"""Host-side mirror of models/vq.py (VectorQuantizer :28-122, VQMaskQuantizer :125-162): nearest-centroid quantisation with
the straight-through estimator and the commitment + codebook loss, in routing.hip.  SURVEY 8f N4."""
import torch
from torch import nn

from .. import config
from .._hostmod import HipModule


class VectorQuantizer(HipModule):
    def __init__(self, num_centroids=None, commitment_weight=None):
        super().__init__()
        self.num_centroids = num_centroids or config.VQ_NUM_CENTROIDS
        self.beta = commitment_weight or config.VQ_COMMITMENT_WEIGHT
        if self.num_centroids > 16:
            raise NotImplementedError("VectorQuantizer (HIP build): at most 16 centroids")
        self.centroids = nn.Parameter(torch.linspace(0, 1, self.num_centroids))

    def forward(self, x):
        """x: any shape -> (quantized like x, indices int64 like x, scalar loss)"""
        from .. import train
        self._require_device(x)
        return train.VQFunction.apply(x, self.centroids, float(self.beta))

    def get_centroids(self):
        return torch.sort(self.centroids)[0]

    @torch.no_grad()
    def get_utilization(self, indices):
        counts = torch.bincount(indices.reshape(-1), minlength=self.num_centroids)[:self.num_centroids]
        return (counts.float() / indices.numel()).cpu()


class VQMaskQuantizer(nn.Module):
    """mask_estimator -> soft mask -> VectorQuantizer (models/vq.py:125-162)"""

    def __init__(self, mask_estimator, num_centroids=None):
        super().__init__()
        self.mask_estimator = mask_estimator
        self.vq = VectorQuantizer(num_centroids=num_centroids)

    def forward(self, x, return_soft=False):
        soft_mask = self.mask_estimator(x)
        quantized_mask, _indices, vq_loss = self.vq(soft_mask)
        if return_soft:
            return quantized_mask, soft_mask, vq_loss
        return quantized_mask, vq_loss
