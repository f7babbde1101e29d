"""Parameter shell of RNNLayer (onpolicy/algorithms/utils/rnn.py:7-22): `rnn.rnn.*` (GRU) and `rnn.norm.*`."""
import torch.nn as nn

from .flat_modules import ParamPair, _view_param


class _GRUParams(nn.Module):
    def __init__(self, flat, entries):
        super().__init__()
        for name in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
            setattr(self, name, _view_param(flat, *entries["rnn.rnn." + name]))


class RNNShell(nn.Module):
    def __init__(self, flat, entries):
        super().__init__()
        self.rnn = _GRUParams(flat, entries)
        self.norm = ParamPair(_view_param(flat, *entries["rnn.norm.weight"]), _view_param(flat, *entries["rnn.norm.bias"]))
