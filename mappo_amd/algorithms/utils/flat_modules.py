"""nn.Module shells whose Parameters are VIEWS into one flat HBM buffer (mappo_amd/flat.py).

They exist so that `state_dict()` / `load_state_dict()` / `parameters()` keep the reference's key names and
checkpoints stay interchangeable, while the HIP kernels read and update the flat buffer
directly.  No module here has a torch forward: the compute lives in libmappo_hip.so."""
import math

import torch
import torch.nn as nn


def _view_param(flat, off, shape):
    n = 1
    for s in shape:
        n *= s
    return nn.Parameter(flat[off:off + n].view(shape), requires_grad=False)


class ParamPair(nn.Module):
    """A module with .weight / .bias (Linear or LayerNorm stand-in)."""

    def __init__(self, weight, bias):
        super().__init__()
        self.weight, self.bias = weight, bias


class Placeholder(nn.Module):
    """Parameter-free slot (the activation at index 1 of the reference's nn.Sequential blocks)."""


def linear_ln_block(flat, entries, prefix):
    """nn.Sequential(Linear, act, LayerNorm) look-alike: children '0' and '2' carry the parameters."""
    blk = nn.Sequential()
    blk.add_module("0", ParamPair(_view_param(flat, *entries[prefix + ".0.weight"]), _view_param(flat, *entries[prefix + ".0.bias"])))
    blk.add_module("1", Placeholder())
    blk.add_module("2", ParamPair(_view_param(flat, *entries[prefix + ".2.weight"]), _view_param(flat, *entries[prefix + ".2.bias"])))
    return blk


def owned_linear_ln_block(in_dim, hidden, device):
    """The never-used `fc_h` template layer (mlp.py:20-22): real parameters outside the flat buffer, so that
    state_dict keys and parameter counts match the reference; it gets no gradient and no optimizer state."""
    blk = nn.Sequential()
    blk.add_module("0", ParamPair(nn.Parameter(torch.zeros(hidden, in_dim, device=device), requires_grad=False),
                                  nn.Parameter(torch.zeros(hidden, device=device), requires_grad=False)))
    blk.add_module("1", Placeholder())
    blk.add_module("2", ParamPair(nn.Parameter(torch.ones(hidden, device=device), requires_grad=False),
                                  nn.Parameter(torch.zeros(hidden, device=device), requires_grad=False)))
    return blk


@torch.no_grad()
def reference_init_state(args, in_dim, out_dim, head_prefix, head_gain, recurrent):
    """Initial weights with the reference's scheme AND its RNG consumption order, so that the same
    `torch.manual_seed` yields the same initial network as the reference (mlp.py:11-22, rnn.py:13-22,
    distributions.py:58-62, r_actor_critic.py:136-142): every nn.Linear / nn.GRU is constructed (default init
    draws from the generator) and then re-initialised orthogonal|xavier with gain relu/tanh (trunk),
    `head_gain` (head) or 1 (GRU); biases 0; LayerNorm (1, 0).  Runs on the CPU generator."""
    H, D = args.hidden_size, in_dim
    trunk_gain = nn.init.calculate_gain("relu" if args.use_ReLU else "tanh")
    init = nn.init.orthogonal_ if args.use_orthogonal else nn.init.xavier_uniform_
    sd = {}

    def linear(i, o, gain):
        lin = nn.Linear(i, o)
        init(lin.weight.data, gain=gain)
        lin.bias.data.zero_()
        return lin.weight.data.clone(), lin.bias.data.clone()

    def put_block(prefix, w, b):
        sd[prefix + ".0.weight"], sd[prefix + ".0.bias"] = w.clone(), b.clone()
        sd[prefix + ".2.weight"], sd[prefix + ".2.bias"] = torch.ones(H), torch.zeros(H)

    if args.use_feature_normalization:
        sd["base.feature_norm.weight"], sd["base.feature_norm.bias"] = torch.ones(D), torch.zeros(D)
    put_block("base.mlp.fc1", *linear(D, H, trunk_gain))
    wh, bh = linear(H, H, trunk_gain)
    put_block("base.mlp.fc_h", wh, bh)
    for l in range(args.layer_N):                      # get_clones: deep copies of fc_h, no new draws
        put_block(f"base.mlp.fc2.{l}", wh, bh)
    if recurrent:
        gru = nn.GRU(H, H, num_layers=args.recurrent_N)
        for name, param in gru.named_parameters():
            if "bias" in name:
                nn.init.constant_(param, 0)
            elif "weight" in name:
                (nn.init.orthogonal_ if args.use_orthogonal else nn.init.xavier_uniform_)(param)
            sd["rnn.rnn." + name] = param.data.clone()
        sd["rnn.norm.weight"], sd["rnn.norm.bias"] = torch.ones(H), torch.zeros(H)
    w, b = linear(H, out_dim, head_gain)
    sd[head_prefix + ".weight"], sd[head_prefix + ".bias"] = w, b
    return sd
