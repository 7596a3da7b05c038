"""Lifted-structure retrieval loss on the ground <-> aerial correlation matrix.

Counterpart of the reference's loss/lift_loss.py:8-22.  Restates the published algorithm of
pytorch_metric_learning.losses.LiftedStructureLoss(neg_margin=1, pos_margin=0) (absent, un-pinned):
for every ordered positive pair (i, j): relu(logsumexp over the negatives of i and of j of
(neg_margin - d) + d_ij - pos_margin)^2 / 2, mean over pairs.  PARITY UNPINNED.
"""
import torch
import torch.nn as nn

from .contrastive_loss import _pair_dist


class LiftedStructureLoss(nn.Module):
    def __init__(self, neg_margin=1, pos_margin=0, **kwargs):
        super().__init__()
        self.neg_margin, self.pos_margin = float(neg_margin), float(pos_margin)

    def get_loss(self, cmr_embeddings, map_embeddings):
        B = cmr_embeddings.shape[0]
        d = _pair_dist(cmr_embeddings, map_embeddings)                       # (2B, 2B)
        lab = torch.arange(B, device=d.device).repeat(2)
        same = lab[:, None] == lab[None, :]
        e = torch.where(same, torch.full_like(d, float("-inf")), self.neg_margin - d)
        lse_row = torch.logsumexp(e, dim=1)                                   # negatives of each anchor
        partner = (torch.arange(2 * B, device=d.device) + B) % (2 * B)       # the one positive of i
        both = torch.logaddexp(lse_row, lse_row[partner])
        d_pos = d[torch.arange(2 * B, device=d.device), partner]
        return (torch.relu(both + d_pos - self.pos_margin) ** 2 / 2.0).mean()

    forward = get_loss
