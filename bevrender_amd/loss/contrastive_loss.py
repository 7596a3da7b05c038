"""Contrastive retrieval loss on the ground <-> aerial correlation matrix.

Counterpart of the reference's loss/contrastive_loss.py:6-19 (`ContrastiveLoss().get_loss(cmr, map)`).
The reference delegates to pytorch_metric_learning.losses.ContrastiveLoss (un-vendored, version
un-pinned, absent here): its published defaults are restated -- LpDistance(normalize_embeddings=True, p=2),
labels [0..B-1, 0..B-1], pos term relu(d - pos_margin), neg term relu(neg_margin - d), each averaged over
its non-zero entries and summed.  PARITY UNPINNED (no reference test or fixture exists for it).
The (2B x 2B) correlation comes from the HIP kernel (ops.pairwise_corr, csrc/corr.hip).
"""
import torch
import torch.nn as nn

from .. import ops


def _pair_dist(cmr, mp):
    emb = torch.cat((cmr.flatten(1), mp.flatten(1)), 0)
    d2 = ops.pairwise_corr(emb, emb, normalize=True).clamp_min(0)
    return torch.where(d2 > 0, torch.sqrt(d2.clamp_min(1e-16)), torch.zeros_like(d2))


def _avg_nonzero(t):
    """AvgNonZeroReducer: mean over the entries > 0, 0 if there is none -- as a masked sum, without the host
    synchronisation a data-dependent index would cost inside the training step."""
    nz = t > 0
    return (t * nz).sum() / nz.sum().clamp_min(1)


class ContrastiveLoss(nn.Module):
    def __init__(self, pos_margin=0.0, neg_margin=1.0):
        super().__init__()
        self.pos_margin, self.neg_margin = pos_margin, neg_margin

    def get_loss(self, cmr_embeddings, map_embeddings):
        B = cmr_embeddings.shape[0]
        d = _pair_dist(cmr_embeddings, map_embeddings)
        lab = torch.arange(B, device=d.device).repeat(2)
        same = lab[:, None] == lab[None, :]
        eye = torch.eye(2 * B, dtype=torch.bool, device=d.device)
        pos = torch.relu(d - self.pos_margin)[same & ~eye]
        neg = torch.relu(self.neg_margin - d)[~same]
        return _avg_nonzero(pos) + _avg_nonzero(neg)

    forward = get_loss
