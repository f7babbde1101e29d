"""R_Actor / R_Critic — API of `onpolicy/algorithms/r_mappo/algorithm/r_actor_critic.py:11-165`.

Same constructor signatures, method names, argument meaning and return conventions; the forward passes
run as fused MFMA kernels (mappo_amd/csrc/mlp.hip) on the flat parameter buffer.  There is no autograd
graph: gradients come from mappo_mlp_backward inside R_MAPPO.ppo_update."""
import torch
import torch.nn as nn

from mappo_amd import flat as flat_layout
from mappo_amd import ops
from mappo_amd.utils.util import obs_dim_of, to_device_f32
from mappo_amd.algorithms.utils.flat_modules import (ParamPair, _view_param, linear_ln_block, owned_linear_ln_block,
                                   reference_init_state)


class _MLPShell(nn.Module):
    def __init__(self, flat, entries, desc, device):
        super().__init__()
        self.fc1 = linear_ln_block(flat, entries, "base.mlp.fc1")
        self.fc_h = owned_linear_ln_block(desc.hidden, desc.hidden, device)
        self.fc2 = nn.ModuleList([linear_ln_block(flat, entries, f"base.mlp.fc2.{l}") for l in range(desc.layer_N)])


class _BaseShell(nn.Module):
    def __init__(self, flat, entries, desc, device):
        super().__init__()
        if desc.use_feature_norm:
            self.feature_norm = ParamPair(_view_param(flat, *entries["base.feature_norm.weight"]),
                                          _view_param(flat, *entries["base.feature_norm.bias"]))
        self.mlp = _MLPShell(flat, entries, desc, device)


class _LinearHolder(nn.Module):
    def __init__(self, linear):
        super().__init__()
        self.linear = linear


class _ActShell(nn.Module):
    def __init__(self, flat, entries):
        super().__init__()
        self.action_out = _LinearHolder(ParamPair(_view_param(flat, *entries["act.action_out.linear.weight"]),
                                                  _view_param(flat, *entries["act.action_out.linear.bias"])))


def _check_supported(args):
    if args.use_popart:
        raise NotImplementedError("use_popart: PopArt.update raises in the reference itself (SURVEY.md §8c)")
    if args.use_recurrent_policy or args.use_naive_recurrent_policy:
        if args.recurrent_N != 1:
            raise NotImplementedError("recurrent_N != 1")
    if args.hidden_size != 64:
        raise NotImplementedError(f"hidden_size {args.hidden_size}: the MFMA kernels are tiled for 64 (config default)")


class _NetBase(nn.Module):
    """Shared plumbing: flat slice, descriptor, layout."""

    def _setup(self, args, in_dim, out_dim, head_prefix, flat, device):
        _check_supported(args)
        self.hidden_size = args.hidden_size
        self._recurrent = bool(args.use_naive_recurrent_policy or args.use_recurrent_policy)
        self._recurrent_N = args.recurrent_N
        self.desc = ops.net_desc(in_dim, out_dim, args.layer_N, args.use_ReLU, args.use_feature_normalization,
                                 self._recurrent, args.hidden_size)
        layout, P = flat_layout.net_layout(self.desc, head_prefix)
        if flat is None:
            flat = torch.zeros(flat_layout.padded(P), dtype=torch.float32, device=device)
        assert flat.numel() >= P and flat.is_contiguous()
        self.flat, self.n_params, self.layout = flat, P, layout
        self.device_ = torch.device(device)
        return {k: (off, shape) for k, off, shape in layout}

    def _in(self, x):
        return to_device_f32(x, self.device_)


class R_Actor(_NetBase):
    def __init__(self, args, obs_space, action_space, device=torch.device("cuda"), flat=None):
        super().__init__()
        if action_space.__class__.__name__ != "Discrete":
            raise NotImplementedError(f"{action_space.__class__.__name__} action space (BASELINE configs are Discrete)")
        self._gain = args.gain
        self._use_policy_active_masks = args.use_policy_active_masks
        self.n_actions = action_space.n
        entries = self._setup(args, obs_dim_of(obs_space), action_space.n, "act.action_out.linear", flat, device)
        self.base = _BaseShell(self.flat, entries, self.desc, device)
        if self._recurrent:
            from mappo_amd.algorithms.utils.rnn_shell import RNNShell
            self.rnn = RNNShell(self.flat, entries)
        self.act = _ActShell(self.flat, entries)
        self.load_state_dict(reference_init_state(args, self.desc.in_dim, action_space.n, "act.action_out.linear",
                                                  args.gain, self._recurrent))
        self._sample_counter = 0
        from mappo_amd.distributed import sampling_seed
        self._seed = sampling_seed(int(getattr(args, "seed", 1)))       # rank-keyed under data parallelism (parameters are not)
        # device word added to the sampling counter inside the kernel: a captured hipGraph bakes the host counter,
        # so the rollout graph bumps this word once per replay to keep drawing fresh random numbers
        self._counter_dev = torch.zeros(1, dtype=torch.int64, device=self.device_)

    # r_actor_critic.py:43-70
    @torch.no_grad()
    def forward(self, obs, rnn_states, masks, available_actions=None, deterministic=False, out=None, counter=None):
        """`out=(actions_f32[B], logp[B])` writes in place (fused rollout); `counter` fixes the host part of the
        sampling counter (the runner passes the step index so that eager and graph-replayed rollouts agree)."""
        obs = self._in(obs)
        B = obs.shape[0]
        avail = self._in(available_actions) if available_actions is not None else None
        if out is None:
            actions_f = torch.empty(B, dtype=torch.float32, device=self.device_)
            logp = torch.empty(B, dtype=torch.float32, device=self.device_)
        else:
            actions_f, logp = out
        if self._recurrent:
            from mappo_amd.recurrent import actor_step
            rnn_states = actor_step(self, obs, self._in(rnn_states), self._in(masks), avail, deterministic, actions_f, logp, counter)
        else:
            if counter is None:
                counter = self._sample_counter
                self._sample_counter += 1
            ops.actor_act(self.flat, self.desc, obs, avail, B, deterministic, self._seed, counter, actions_f, logp,
                          self._counter_dev)
            rnn_states = rnn_states if torch.is_tensor(rnn_states) else self._in(rnn_states)
        if out is not None:
            return actions_f, logp, rnn_states
        return actions_f.long().view(B, 1), logp.view(B, 1), rnn_states

    # r_actor_critic.py:72-107 — forward-only evaluation (the training path fuses this into R_MAPPO.ppo_update)
    @torch.no_grad()
    def evaluate_actions(self, obs, rnn_states, action, masks, available_actions=None, active_masks=None):
        obs = self._in(obs)
        B = obs.shape[0]
        if self._recurrent:
            from mappo_amd.recurrent import actor_sequence_logits
            logits = actor_sequence_logits(self, obs, self._in(rnn_states), self._in(masks))
        else:
            logits = torch.empty(B, self.n_actions, dtype=torch.float32, device=self.device_)
            ops.mlp_forward(self.flat, self.desc, obs, None, B, logits)
        if available_actions is not None:
            logits = logits.masked_fill(self._in(available_actions) == 0, -1e10)
        logp_all = logits - torch.logsumexp(logits, dim=-1, keepdim=True)
        p = logp_all.exp()
        ent = -(p * logp_all.clamp(min=torch.finfo(torch.float32).min)).sum(-1)
        logp = logp_all.gather(-1, self._in(action).long().view(B, 1))
        if active_masks is not None and self._use_policy_active_masks:
            am = self._in(active_masks)
            ent = (ent * am.squeeze(-1)).sum() / am.sum()
        else:
            ent = ent.mean()
        return logp, ent


class R_Critic(_NetBase):
    def __init__(self, args, cent_obs_space, device=torch.device("cuda"), flat=None):
        super().__init__()
        entries = self._setup(args, obs_dim_of(cent_obs_space), 1, "v_out", flat, device)
        self.base = _BaseShell(self.flat, entries, self.desc, device)
        if self._recurrent:
            from mappo_amd.algorithms.utils.rnn_shell import RNNShell
            self.rnn = RNNShell(self.flat, entries)
        self.v_out = ParamPair(_view_param(self.flat, *entries["v_out.weight"]), _view_param(self.flat, *entries["v_out.bias"]))
        self.load_state_dict(reference_init_state(args, self.desc.in_dim, 1, "v_out", 1.0, self._recurrent))

    # r_actor_critic.py:146-165
    @torch.no_grad()
    def forward(self, cent_obs, rnn_states, masks, out=None):
        cent_obs = self._in(cent_obs)
        B = cent_obs.shape[0]
        values = out if out is not None else torch.empty(B, 1, dtype=torch.float32, device=self.device_)
        if self._recurrent:
            from mappo_amd.recurrent import critic_forward
            rnn_states = critic_forward(self, cent_obs, self._in(rnn_states), self._in(masks), values)
        else:
            ops.mlp_forward(self.flat, self.desc, cent_obs, None, B, values)
            rnn_states = rnn_states if torch.is_tensor(rnn_states) else self._in(rnn_states)
        return values, rnn_states
