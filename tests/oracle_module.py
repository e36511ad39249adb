"""TEST INFRASTRUCTURE: an nn.Module with ScoreTranscription's constructor/forward signature backed by the CPU oracle, so that the
recipe plumbing (yaml -> Brain -> fit/evaluate -> metrics -> checkpoints) can be exercised without a GPU.  Never used by the product."""
import random

import torch
import torch.nn as nn

from oracle import model_ref
from piano_a2s_amd import spec


class OracleTranscription(nn.Module):
    def __init__(self, **kw):
        super().__init__()
        self.cfg = spec.default_cfg(**kw)
        st = spec.procedural_state(self.cfg, 11, eos_bias=3.0, lively=True)
        self.names = [k for k in st if not spec.is_buffer(k)]
        self.params = nn.ParameterList([nn.Parameter(st[k].clone()) for k in self.names])
        self.bufs = {k: v.clone() for k, v in st.items() if spec.is_buffer(k)}

    def forward(self, spectrogram, inference=True, ground_truth=None, teacher_forcing_ratio=0., device=None):
        P = dict(zip(self.names, self.params))
        return model_ref.forward(P, self.bufs, self.cfg, spectrogram, inference=inference, ground_truth=ground_truth,
                                 teacher_forcing_ratio=teacher_forcing_ratio, training=self.training, rng=random, dropout=True)
