"""Temporal aggregators.  Only ``Identity`` is on the AVT hot path (reference models/temporal_aggregation.py:21-31,
selected by expts/01_ek100_avt.txt:12); the baselines (Mean, Transformer encoder, RULSTM) are out of scope."""
import torch.nn as nn


class Identity(nn.Identity):
    def __init__(self, in_features):
        super().__init__()
        self.in_features = in_features

    def forward(self, *args, **kwargs):
        return super().forward(*args, **kwargs), {}

    @property
    def output_dim(self):
        return self.in_features
