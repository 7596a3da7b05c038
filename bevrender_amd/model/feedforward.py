"""Counterpart of the reference's model/feedforward.py:4-21 (constructed by EncoderLayer, never called)."""
import torch.nn as nn


class FeedForwardLayer(nn.Module):
    def __init__(self, in_dim=256, hidden_dim=256, dropout=0.0):
        super().__init__()
        self.dim, self.hidden_dim, self.dropout = in_dim, hidden_dim, dropout
        self.ffn = nn.Sequential(nn.Linear(in_dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, in_dim), nn.Dropout(dropout))

    def forward(self, x):
        return self.ffn(x)
