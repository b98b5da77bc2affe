"""`basicsr.archs.raft_arch.RAFT_SR` (reference raft_arch.py:668-807) — SURVEY.md §8(f) row 1, 'next': optical flow
is an INPUT of the hot path (every BASELINE config supplies synthetic flows).  The class exists so that the shipped
YAML (`flownet_config.target`) instantiates; calling it raises."""
import torch.nn as nn


class RAFT_SR(nn.Module):
    def __init__(self, model="normal", load_path=None, **kw):
        super().__init__()
        self.model, self.load_path = model, load_path

    def forward(self, *a, **k):
        raise NotImplementedError("RAFT flow estimation is outside the MI355X hot path (SURVEY.md §8(f)); "
                                  "pass precomputed flows to sample()/sample_canvas()")
