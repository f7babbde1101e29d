"""R_MAPPOPolicy — API of `onpolicy/algorithms/r_mappo/algorithm/rMAPPOPolicy.py:6-127`.

Owns ONE flat HBM buffer `[actor params | critic params]` (each padded to 256 floats) plus Adam moments,
step counters and hyper-parameters on the device; `actor` / `critic` are nn.Module shells viewing it (same
state_dict keys as the reference), `actor_optimizer` / `critic_optimizer` are facades over the fused
clip+Adam kernel (mappo_clip_adam) that keep `param_groups[..]['lr']`, `state_dict()` and `zero_grad()`."""
import os

import torch

from mappo_amd import flat as flat_layout
from mappo_amd import ops
from mappo_amd.utils.util import update_linear_schedule, to_device_f32
from .r_actor_critic import R_Actor, R_Critic

# opt_hyper row: lr, beta1, beta2, eps, weight_decay, max_grad_norm, use_clip, enabled
H_LR, H_B1, H_B2, H_EPS, H_WD, H_MAXNORM, H_CLIP, H_ENABLED = range(8)


class FlatAdam:
    """torch.optim.Adam look-alike for one segment (actor or critic) of the policy's flat buffer."""

    def __init__(self, policy, seg, lr, eps, weight_decay):
        self._policy, self._seg = policy, seg
        self.param_groups = [dict(lr=lr, betas=(0.9, 0.999), eps=eps, weight_decay=weight_decay)]
        self.sync_lr()

    def sync_lr(self):
        g = self.param_groups[0]
        row = torch.tensor([g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"]], dtype=torch.float32)
        self._policy.opt_hyper[self._seg, :5].copy_(row, non_blocking=True)

    def zero_grad(self, set_to_none=True):
        pass      # the fused backward overwrites every gradient slab; nothing accumulates between updates

    def step(self):
        raise RuntimeError("FlatAdam.step(): the Adam update is fused into R_MAPPO.ppo_update (mappo_clip_adam); "
                           "there is no separate optimizer step and no autograd .grad to consume")

    def _range(self):
        b = self._policy.seg_bounds
        return b[self._seg], b[self._seg + 1]

    def state_dict(self):
        lo, hi = self._range()
        return dict(step=int(self._policy.opt_step[self._seg].item()), exp_avg=self._policy.exp_avg[lo:hi].clone(),
                    exp_avg_sq=self._policy.exp_avg_sq[lo:hi].clone(), param_groups=[dict(g) for g in self.param_groups])

    def load_state_dict(self, sd):
        lo, hi = self._range()
        self._policy.exp_avg[lo:hi].copy_(sd["exp_avg"])
        self._policy.exp_avg_sq[lo:hi].copy_(sd["exp_avg_sq"])
        self._policy.opt_step[self._seg] = int(sd["step"])
        self.param_groups = [dict(g) for g in sd["param_groups"]]
        self.sync_lr()


class R_MAPPOPolicy:
    def __init__(self, args, obs_space, cent_obs_space, act_space, device=torch.device("cuda")):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            from mappo_amd._lib import MappoHipError
            raise MappoHipError("R_MAPPOPolicy needs a GPU device: the MAPPO hot path runs as HIP kernels on gfx950 and "
                                "has no CPU fallback")
        self.lr = args.lr
        self.critic_lr = args.critic_lr
        self.opti_eps = args.opti_eps
        self.weight_decay = args.weight_decay
        self.obs_space = obs_space
        self.share_obs_space = cent_obs_space
        self.act_space = act_space

        # sizes first (descriptors only), then one allocation for both networks
        from mappo_amd.utils.util import obs_dim_of
        rec = bool(args.use_recurrent_policy or args.use_naive_recurrent_policy)
        da = ops.net_desc(obs_dim_of(obs_space), act_space.n, args.layer_N, args.use_ReLU, args.use_feature_normalization,
                          rec, args.hidden_size)
        dc = ops.net_desc(obs_dim_of(cent_obs_space), 1, args.layer_N, args.use_ReLU, args.use_feature_normalization,
                          rec, args.hidden_size)
        _, pa = flat_layout.net_layout(da, "act.action_out.linear")
        _, pc = flat_layout.net_layout(dc, "v_out")
        pa_pad, pc_pad = flat_layout.padded(pa), flat_layout.padded(pc)
        self.seg_bounds = [0, pa_pad, pa_pad + pc_pad]
        self.n_flat = pa_pad + pc_pad
        f32 = dict(dtype=torch.float32, device=self.device)
        self.flat_params = torch.zeros(self.n_flat, **f32)
        self.flat_grad = torch.zeros(self.n_flat, **f32)
        self.exp_avg = torch.zeros(self.n_flat, **f32)
        self.exp_avg_sq = torch.zeros(self.n_flat, **f32)
        self.opt_step = torch.zeros(2, dtype=torch.int32, device=self.device)
        self.opt_hyper = torch.zeros(2, 8, **f32)
        self.opt_hyper[:, H_MAXNORM] = float(args.max_grad_norm)
        self.opt_hyper[:, H_CLIP] = 1.0 if args.use_max_grad_norm else 0.0
        self.opt_hyper[:, H_ENABLED] = 1.0
        self.grad_norms = torch.zeros(2, **f32)
        self.opt_workspace = ops.optim_workspace(self.n_flat, self.device)

        self.actor = R_Actor(args, self.obs_space, self.act_space, self.device, flat=self.flat_params[:pa_pad])
        self.critic = R_Critic(args, self.share_obs_space, self.device, flat=self.flat_params[pa_pad:])
        self._side_stream = None
        self.actor_optimizer = FlatAdam(self, 0, self.lr, self.opti_eps, self.weight_decay)
        self.critic_optimizer = FlatAdam(self, 1, self.critic_lr, self.opti_eps, self.weight_decay)

    # rMAPPOPolicy.py:39-46
    def lr_decay(self, episode, episodes):
        update_linear_schedule(self.actor_optimizer, episode, episodes, self.lr)
        update_linear_schedule(self.critic_optimizer, episode, episodes, self.critic_lr)

    # rMAPPOPolicy.py:48-74
    def get_actions(self, cent_obs, obs, rnn_states_actor, rnn_states_critic, masks, available_actions=None,
                    deterministic=False):
        actions, action_log_probs, rnn_states_actor = self.actor(obs, rnn_states_actor, masks, available_actions, deterministic)
        values, rnn_states_critic = self.critic(cent_obs, rnn_states_critic, masks)
        return values, actions, action_log_probs, rnn_states_actor, rnn_states_critic

    # rMAPPOPolicy.py:76-86
    def get_values(self, cent_obs, rnn_states_critic, masks):
        values, _ = self.critic(cent_obs, rnn_states_critic, masks)
        return values

    # rMAPPOPolicy.py:88-114
    def evaluate_actions(self, cent_obs, obs, rnn_states_actor, rnn_states_critic, action, masks, available_actions=None,
                         active_masks=None):
        action_log_probs, dist_entropy = self.actor.evaluate_actions(obs, rnn_states_actor, action, masks,
                                                                     available_actions, active_masks)
        values, _ = self.critic(cent_obs, rnn_states_critic, masks)
        return values, action_log_probs, dist_entropy

    # rMAPPOPolicy.py:116-127
    def act(self, obs, rnn_states_actor, masks, available_actions=None, deterministic=False):
        actions, _, rnn_states_actor = self.actor(obs, rnn_states_actor, masks, available_actions, deterministic)
        return actions, rnn_states_actor

    # ---- one-launch rollout step (mappo_rollout_step) ----------------------------------------------------------
    def can_dual_update(self):
        """Both networks' PPO update in one launch (mappo_actor_critic_update): narrow inputs only."""
        a, c = self.actor.desc, self.critic.desc
        return (not a.recurrent and not c.recurrent and a.in_dim <= 64 and c.in_dim <= 64 and a.layer_N == c.layer_N
                and a.use_relu == c.use_relu)

    def can_fuse_step(self):
        a, c = self.actor.desc, self.critic.desc
        same_class = (a.in_dim <= 64) == (c.in_dim <= 64) and max(a.in_dim, c.in_dim) <= 512       # both narrow or both wide
        return (not a.recurrent and not c.recurrent and same_class and a.layer_N == c.layer_N and a.use_relu == c.use_relu)

    @torch.no_grad()
    def collect_step_fused(self, buffer, step, pending=None, centralized=True, use_available_actions=False, deterministic=False,
                           values_only=None):
        """get_actions on the rows of step `step` AND (pending = (obs, rewards, dones) of the env step before it) the
        insert of that env output into slot `step` — one kernel.  With `pending` the networks read the rows straight
        from the env's output (strided views are fine) while other workgroups of the same launch copy them into
        obs[step] / share_obs[step] / rewards[step-1] / masks[step]; without it they read the buffer slot.
        Returns the fp32 actions view [N, M, 1], or None when `pending` does not have the expected device layout
        (the caller then falls back to insert + collect_into).  values_only = a [R] fp32 tensor: the bootstrap step
        (step == episode_length) — only the critic runs, its values go to that tensor (base_runner.py:110-118)."""
        N, M = buffer.n_rollout_threads, buffer.num_agents
        R = N * M
        D = self.actor.desc.in_dim
        avail = buffer.available_actions[step].view(R, -1) if (use_available_actions and values_only is None) else None
        insert = None
        if pending is None:
            obs_src = (buffer.obs[step], 0, 0)
            share_src = (buffer.share_obs[step], 0, 0)
            Mk = 0
        else:
            obs, rewards, dones = pending
            ok = (torch.is_tensor(obs) and torch.is_tensor(rewards) and torch.is_tensor(dones) and obs.device == self.device
                  and obs.dtype == torch.float32 and obs.dim() == 3 and obs.stride(2) == 1 and tuple(obs.shape) == (N, M, D)
                  and rewards.device == self.device and rewards.dtype == torch.float32 and dones.device == self.device
                  and dones.dtype == torch.bool and dones.dim() == 2)
            if centralized:      # share row of (n, m) = the thread's agents side by side: needs them contiguous in the source
                ok = ok and obs.stride(1) == D and self.critic.desc.in_dim == M * D
            else:
                ok = ok and self.critic.desc.in_dim == D
            if not ok:
                return None
            if rewards.dim() == 3:
                rewards = rewards[..., 0]
            obs_src = (obs, obs.stride(0), obs.stride(1))
            share_src = (obs, obs.stride(0), 0 if centralized else obs.stride(1))
            Mk = M
            insert = dict(obs_dst=buffer.obs[step], share_dst=buffer.share_obs[step], rewards=(rewards, rewards.stride(0), rewards.stride(1)),
                          dones=(dones, dones.stride(0), dones.stride(1)), rew_dst=buffer.rewards[step - 1], mask_dst=buffer.masks[step],
                          centralized=centralized)
        if values_only is not None:
            ops.rollout_step(self.actor.flat, self.actor.desc, self.critic.flat, self.critic.desc, obs_src, share_src, Mk, R, None,
                             deterministic, self.actor._seed, step, None, None, None, values_only, insert)
            return values_only
        ops.rollout_step(self.actor.flat, self.actor.desc, self.critic.flat, self.critic.desc, obs_src, share_src, Mk, R, avail,
                         deterministic, self.actor._seed, step, self.actor._counter_dev, buffer.actions[step].view(R),
                         buffer.action_log_probs[step].view(R), buffer.value_preds[step].view(R), insert)
        return buffer.actions[step]

    # ---- recurrent policies, SMAC-style envs: insert of the previous env output + this step's get_actions in one launch ----
    def can_fuse_recurrent_step(self, n_rows):
        from mappo_amd import recurrent
        if not (getattr(self.actor, "_recurrent", False) and getattr(self.critic, "_recurrent", False)):
            return False
        if not recurrent.can_step_dual(self.actor, self.critic):
            return False
        a, c = self.actor.desc, self.critic.desc
        if max(a.in_dim, c.in_dim) <= 64:
            return a.layer_N <= 1 and n_rows <= 1024
        # wide inputs: one 16-row tile per workgroup, every workgroup fetching its own copy of the weights — beyond ~1 tile per CU and network (measured crossover between 2 560 and 5 120 rows at the MMM2 shape)
        # the separate launches (8 tiles per workgroup share a weight stream) are faster
        return min(a.in_dim, c.in_dim) > 64 and n_rows <= int(os.environ.get("MAPPO_WIDE_FUSED_MAX_ROWS", 4096))

    @torch.no_grad()
    def collect_step_fused_recurrent(self, buffer, step, pending, deterministic=False):
        """mappo_recurrent_rollout_step: `pending` = (obs, share_obs, rewards, dones, bad_transition, available_actions, rnn_states,
        rnn_states_critic) — what the env returned for step - 1 and the states that step's get_actions returned — is inserted into
        buffer slot `step` (smac_runner.py:129-151) by some workgroups while the others run get_actions / get_values of `step` on
        it in place.  Returns (actions view [N, M, 1], next actor states, next critic states), or None when `pending` does not have
        the device layout the kernel reads (the caller then inserts and collects separately)."""
        obs, share_obs, rewards, dones, bad, avail, rnn_a, rnn_c = pending
        b = buffer
        dev, N, M = b.device, b.n_rollout_threads, b.num_agents
        R = N * M
        f32 = lambda t, d: (torch.is_tensor(t) and t.device == dev and t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 3
                            and tuple(t.shape) == (N, M, d))
        ok = (f32(obs, b.obs.shape[-1]) and f32(share_obs, b.share_obs.shape[-1]) and torch.is_tensor(rewards) and rewards.device == dev
              and rewards.dtype == torch.float32 and rewards.dim() in (2, 3) and torch.is_tensor(dones) and dones.device == dev
              and dones.dtype == torch.bool and dones.dim() == 2
              and (bad is None or (torch.is_tensor(bad) and bad.device == dev and bad.dtype == torch.bool and bad.is_contiguous()
                                   and bad.numel() == R))
              and (avail is None or (b.available_actions is not None and f32(avail, b.available_actions.shape[-1])))
              and torch.is_tensor(rnn_a) and torch.is_tensor(rnn_c) and rnn_a.is_contiguous() and rnn_c.is_contiguous()
              and rnn_a.numel() == R * 64 and rnn_c.numel() == R * 64 and b.recurrent_N == 1 and step == b.step + 1)
        if not ok:
            return None
        ha = torch.empty(R, 1, 64, dtype=torch.float32, device=dev)
        hc = torch.empty(R, 1, 64, dtype=torch.float32, device=dev)
        slot = dict(obs=b.obs[step], share_obs=b.share_obs[step], available_actions=b.available_actions[step] if avail is not None else None,
                    rewards=b.rewards[step - 1], masks=b.masks[step], bad_masks=b.bad_masks[step], active_masks=b.active_masks[step],
                    rnn_states=b.rnn_states[step], rnn_states_critic=b.rnn_states_critic[step])
        ops.recurrent_rollout_step(self.actor.flat, self.actor.desc, self.critic.flat, self.critic.desc, obs, share_obs, avail, rewards, dones,
                                   bad, rnn_a, rnn_c, ha, hc, deterministic, self.actor._seed, step, self.actor._counter_dev,
                                   b.actions[step].view(R), b.action_log_probs[step].view(R), b.value_preds[step].view(R), slot)
        b.step = step % b.episode_length
        return b.actions[step], ha, hc

    # ---- fused rollout step (K8 subsumes K1): outputs land in buffer slot `step` ------------------------------
    @torch.no_grad()
    def collect_into(self, buffer, step, use_available_actions=False, deterministic=False):
        """get_actions on buffer slot `step` (mpe_runner.py:95-109 / smac_runner.py:110-125) with actions,
        log-probs and values written straight into buffer.{actions, action_log_probs, value_preds}[step];
        next rnn states (recurrent policies) go to slot step+1.  Returns the fp32 actions view [N, M, 1]."""
        R = buffer.n_rollout_threads * buffer.num_agents
        avail = buffer.available_actions[step].view(R, -1) if use_available_actions else None
        out = (buffer.actions[step].view(R), buffer.action_log_probs[step].view(R))
        masks = buffer.masks[step].view(R, 1)
        if getattr(self.actor, "_recurrent", False) and getattr(self.critic, "_recurrent", False):
            from mappo_amd import recurrent
            if recurrent.can_step_dual(self.actor, self.critic):
                rnn_a, rnn_c = recurrent.step_dual(self.actor, self.critic, buffer.obs[step].view(R, -1), buffer.share_obs[step].view(R, -1),
                                                   buffer.rnn_states[step].view(R, -1), buffer.rnn_states_critic[step].view(R, -1), masks,
                                                   avail, deterministic, out[0], out[1], buffer.value_preds[step].view(R), step)
                return buffer.actions[step], rnn_a, rnn_c
        # the two networks are independent: the critic runs on a side stream next to the actor (also inside a
        # captured hipGraph, where the fork/join becomes two parallel branches)
        cur = torch.cuda.current_stream()
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        side = self._side_stream
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            _, rnn_c = self.critic(buffer.share_obs[step].view(R, -1),
                                   buffer.rnn_states_critic[step].view(R, buffer.recurrent_N, -1), masks,
                                   out=buffer.value_preds[step].view(R, 1))
        _, _, rnn_a = self.actor(buffer.obs[step].view(R, -1), buffer.rnn_states[step].view(R, buffer.recurrent_N, -1),
                                 masks, avail, deterministic, out=out, counter=step)
        cur.wait_stream(side)
        return buffer.actions[step], rnn_a, rnn_c
