"""Semi-hard triplet retrieval loss on the ground <-> aerial correlation matrix.

Counterpart of the reference's loss/triplet_loss_metric.py:8-28 (`TripletLossMetricLearning().get_loss(cmr, map)`).
The reference delegates to pytorch_metric_learning (un-vendored, version un-pinned, absent here); its published
algorithm is restated: TripletMarginMiner(margin=0.2, "semihard") on the L2 distance of normalised embeddings picks
the triplets with 0 < d_an - d_ap <= 0.2; TripletMarginLoss(distance=CosineSimilarity()) scores each with
relu(cos_an - cos_ap + 0.05); ThresholdReducer(high=0.3) averages the scores below 0.3; LpRegularizer() adds the
mean L2 norm of the raw embeddings.  PARITY UNPINNED (no reference test or fixture exists for it).
With labels [0..B-1, 0..B-1] every anchor has exactly one positive (its partner in the other modality), so the
triplets are the (anchor, negative) pairs of one dense (2B x 2B) mask: no index lists, no host synchronisation.
The correlation matrix comes from the HIP kernel (ops.pairwise_corr, csrc/corr.hip).
"""
import torch
import torch.nn as nn

from .. import ops


class TripletLossMetricLearning(nn.Module):
    def __init__(self, miner_margin=0.2, loss_margin=0.05, reducer_high=0.3, embedding_reg_weight=1.0):
        super().__init__()
        self.miner_margin, self.loss_margin = float(miner_margin), float(loss_margin)
        self.reducer_high, self.embedding_reg_weight = float(reducer_high), float(embedding_reg_weight)

    def get_loss(self, cmr_embeddings, map_embeddings):
        B = cmr_embeddings.shape[0]
        emb = torch.cat((cmr_embeddings.flatten(1), map_embeddings.flatten(1)), 0)
        d2 = ops.pairwise_corr(emb, emb, normalize=True)                 # 2 - 2 cos on L2-normalised rows
        cos = 1.0 - 0.5 * d2
        idx = torch.arange(2 * B, device=emb.device)
        partner = (idx + B) % (2 * B)
        neg = (idx[:, None] % B) != (idx[None, :] % B)                    # label[n] != label[a]
        with torch.no_grad():                                             # the miner is not differentiated
            d = d2.clamp_min(0).sqrt()
            m = d - d[idx, partner][:, None]                              # d_an - d_ap
            mined = neg & (m > 0) & (m <= self.miner_margin)
        loss = torch.relu(cos - cos[idx, partner][:, None] + self.loss_margin)
        keep = mined & (loss.detach() < self.reducer_high)
        cnt = keep.sum()
        trip = (loss * keep).sum() / cnt.clamp_min(1)                     # 0 when nothing passes (the sum is 0 too)
        reg = emb.norm(p=2, dim=1).mean() * self.embedding_reg_weight
        return trip + reg

    forward = get_loss
