"""SharedReplayBuffer resident in HBM — API of `onpolicy/utils/shared_buffer.py:14-494`.

Layout: every array is a contiguous float32 tensor `[T(+1), N, M, D]` on the GPU, i.e. the reference's
C-contiguous memory order.  `arr[t]` is therefore a zero-copy `[N, M, D]` slot and `arr[t].view(N*M, D)` the
`np.concatenate(arr[t])` the runners pass to the policy (mpe_runner.py:99-103) — (rollout_threads x agents)
contiguous, so one rollout step touches one contiguous HBM range per array and the flat minibatch row of
sample (t, n, m) is `(t*N + n)*M + m` exactly as in `feed_forward_generator` (shared_buffer.py:249-261).

`compute_returns` is the HIP GAE scan.  The three generators keep the reference's index arithmetic (row
indices are integers and bit-exact); the `*_rows` variants hand the int32 row indices to the fused kernels
without materialising the minibatch, the classic generator methods gather with torch for outside callers."""
import numpy as np
import torch

from .. import ops
from .util import obs_dim_of, get_shape_from_act_space, to_device_f32


class SharedReplayBuffer(object):
    def __init__(self, args, num_agents, obs_space, cent_obs_space, act_space, device=None):
        self.episode_length = args.episode_length
        self.n_rollout_threads = args.n_rollout_threads
        self.num_agents = num_agents
        self.hidden_size = args.hidden_size
        self.recurrent_N = args.recurrent_N
        self.gamma = args.gamma
        self.gae_lambda = args.gae_lambda
        self._use_gae = args.use_gae
        self._use_popart = args.use_popart
        self._use_valuenorm = args.use_valuenorm
        self._use_proper_time_limits = args.use_proper_time_limits
        self.device = torch.device(device if device is not None else "cuda")
        self.perm_device = getattr(args, "perm_device", "cuda")

        T, N, M, H = self.episode_length, self.n_rollout_threads, num_agents, self.hidden_size
        obs_dim, share_dim = obs_dim_of(obs_space), obs_dim_of(cent_obs_space)
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=self.device)
        o = lambda *s: torch.ones(s, dtype=torch.float32, device=self.device)
        self.share_obs = z(T + 1, N, M, share_dim)
        self.obs = z(T + 1, N, M, obs_dim)
        self.rnn_states = z(T + 1, N, M, self.recurrent_N, H)
        self.rnn_states_critic = z(T + 1, N, M, self.recurrent_N, H)
        self.value_preds = z(T + 1, N, M, 1)
        self.returns = z(T + 1, N, M, 1)
        if act_space.__class__.__name__ == "Discrete":
            self.available_actions = o(T + 1, N, M, act_space.n)
        else:
            self.available_actions = None
        act_shape = get_shape_from_act_space(act_space)
        self.actions = z(T, N, M, act_shape)
        self.action_log_probs = z(T, N, M, act_shape)
        self.rewards = z(T, N, M, 1)
        self.masks = o(T + 1, N, M, 1)
        self.bad_masks = o(T + 1, N, M, 1)
        self.active_masks = o(T + 1, N, M, 1)
        self.device = self.obs.device                      # with its index ("cuda" -> "cuda:0"): tensors are compared against it
        self.step = 0

    # ---- slot writes (shared_buffer.py:79-166) -----------------------------------------------------------
    def _put(self, dst, src):
        """One copy kernel per slot write: device tensors (any float/bool dtype, strided or broadcast views) are copied
        straight into the slot; host arrays are uploaded first."""
        if torch.is_tensor(src) and src.device == dst.device:
            if src.shape != dst.shape:
                src = src.reshape(dst.shape) if src.numel() == dst.numel() else src.expand(dst.shape)
            dst.copy_(src)                                   # converts dtype / gathers strides in the same kernel
        else:
            dst.copy_(to_device_f32(src, self.device).view(dst.shape), non_blocking=True)

    def insert(self, share_obs, obs, rnn_states_actor, rnn_states_critic, actions, action_log_probs, value_preds, rewards,
               masks, bad_masks=None, active_masks=None, available_actions=None):
        s = self.step
        self._put(self.share_obs[s + 1], share_obs)
        self._put(self.obs[s + 1], obs)
        self._put(self.rnn_states[s + 1], rnn_states_actor)
        self._put(self.rnn_states_critic[s + 1], rnn_states_critic)
        self._put(self.actions[s], actions)
        self._put(self.action_log_probs[s], action_log_probs)
        self._put(self.value_preds[s], value_preds)
        self._put(self.rewards[s], rewards)
        self._put(self.masks[s + 1], masks)
        if bad_masks is not None:
            self._put(self.bad_masks[s + 1], bad_masks)
        if active_masks is not None:
            self._put(self.active_masks[s + 1], active_masks)
        if available_actions is not None:
            self._put(self.available_actions[s + 1], available_actions)
        self.step = (s + 1) % self.episode_length

    def insert_env(self, share_obs, obs, rewards, masks, rnn_states_actor=None, rnn_states_critic=None, bad_masks=None,
                   active_masks=None, available_actions=None):
        """Fused-rollout variant: the policy kernels already wrote actions / log-probs / values into slot
        `step` (R_MAPPOPolicy.collect_into), so only the environment's outputs remain to be stored."""
        s = self.step
        self._put(self.share_obs[s + 1], share_obs)
        self._put(self.obs[s + 1], obs)
        self._put(self.rewards[s], rewards)
        self._put(self.masks[s + 1], masks)
        if rnn_states_actor is not None:
            self._put(self.rnn_states[s + 1], rnn_states_actor)
        if rnn_states_critic is not None:
            self._put(self.rnn_states_critic[s + 1], rnn_states_critic)
        if bad_masks is not None:
            self._put(self.bad_masks[s + 1], bad_masks)
        if active_masks is not None:
            self._put(self.active_masks[s + 1], active_masks)
        if available_actions is not None:
            self._put(self.available_actions[s + 1], available_actions)
        self.step = (s + 1) % self.episode_length

    def insert_smac_fused(self, share_obs, obs, rewards, dones, bad_transition, available_actions, rnn_states=None,
                          rnn_states_critic=None):
        """SMAC rollout insert as ONE kernel (mappo_insert_smac; smac_runner.py:129-151): the slot copies plus masks /
        active_masks / bad_masks / rnn-state resets derived from `dones` [N, M] and `bad_transition` [N, M] (bool, or None).
        Returns False (nothing written) when the inputs are not device tensors of the expected layout."""
        dev = self.device
        f32 = lambda t, d: (torch.is_tensor(t) and t.device == dev and t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 3
                            and t.shape[-1] == d)
        ok = (f32(obs, self.obs.shape[-1]) and f32(share_obs, self.share_obs.shape[-1]) and torch.is_tensor(rewards)
              and rewards.device == dev and rewards.dtype == torch.float32 and rewards.dim() in (2, 3)
              and torch.is_tensor(dones) and dones.device == dev and dones.dtype == torch.bool and dones.dim() == 2
              and (bad_transition is None or (torch.is_tensor(bad_transition) and bad_transition.device == dev
                                              and bad_transition.dtype == torch.bool and bad_transition.is_contiguous()
                                              and bad_transition.numel() == dones.numel()))
              and (available_actions is None or (self.available_actions is not None
                                                 and f32(available_actions, self.available_actions.shape[-1]))))
        if rnn_states is not None:
            state_ok = lambda h: (torch.is_tensor(h) and h.device == dev and h.dtype == torch.float32 and h.is_contiguous()
                                  and h.numel() == self.rnn_states[0].numel() and h.data_ptr() % 16 == 0)
            ok = ok and state_ok(rnn_states) and state_ok(rnn_states_critic) and (self.recurrent_N * self.rnn_states.shape[-1]) % 4 == 0
        if not ok:
            return False
        s = self.step
        ops.insert_smac(obs, share_obs, available_actions, rewards, dones, bad_transition, rnn_states, rnn_states_critic, self.obs[s + 1],
                        self.share_obs[s + 1], self.available_actions[s + 1] if available_actions is not None else None,
                        self.rewards[s], self.masks[s + 1], self.bad_masks[s + 1], self.active_masks[s + 1],
                        self.rnn_states[s + 1] if rnn_states is not None else None,
                        self.rnn_states_critic[s + 1] if rnn_states is not None else None)
        self.step = (s + 1) % self.episode_length
        return True

    def insert_mpe_fused(self, obs, rewards, dones, centralized, rnn_states=None, rnn_states_critic=None):
        """MPE rollout insert as ONE kernel (mappo_insert_mpe): obs / share_obs -> slot step+1, rewards -> slot step,
        masks = 1 - done -> slot step+1; with rnn_states / rnn_states_critic (recurrent policies) also their slot step+1 =
        states * (1 - done) (mappo_insert_mpe_rnn).  Returns False (nothing written) when the inputs are not device tensors
        of the expected dtypes, so the caller can fall back to the generic slot copies."""
        ok = (torch.is_tensor(obs) and torch.is_tensor(rewards) and torch.is_tensor(dones) and obs.device == self.device
              and obs.dtype == torch.float32 and obs.dim() == 3 and obs.stride(2) == 1 and rewards.device == self.device
              and rewards.dtype == torch.float32 and dones.device == self.device and dones.dtype == torch.bool and dones.dim() == 2)
        if rnn_states is not None:
            state_ok = lambda h: (torch.is_tensor(h) and h.device == self.device and h.dtype == torch.float32 and h.is_contiguous()
                                  and h.numel() == self.rnn_states[0].numel() and h.data_ptr() % 16 == 0)
            ok = ok and state_ok(rnn_states) and state_ok(rnn_states_critic) and (self.recurrent_N * self.rnn_states.shape[-1]) % 4 == 0
        if not ok:
            return False
        s = self.step
        if rnn_states is not None:
            ops.insert_mpe_rnn(obs, rewards, dones, self.obs[s + 1], self.share_obs[s + 1], self.rewards[s], self.masks[s + 1],
                               centralized, rnn_states, rnn_states_critic, self.rnn_states[s + 1], self.rnn_states_critic[s + 1])
        else:
            ops.insert_mpe(obs, rewards, dones, self.obs[s + 1], self.share_obs[s + 1], self.rewards[s], self.masks[s + 1], centralized)
        self.step = (s + 1) % self.episode_length
        return True

    def chooseinsert(self, share_obs, obs, rnn_states, rnn_states_critic, actions, action_log_probs, value_preds, rewards,
                     masks, bad_masks=None, active_masks=None, available_actions=None):
        """Turn-based (Hanabi) insert, shared_buffer.py:114-147: obs/share_obs/active/avail go to slot `step`."""
        s = self.step
        self._put(self.share_obs[s], share_obs)
        self._put(self.obs[s], obs)
        self._put(self.rnn_states[s + 1], rnn_states)
        self._put(self.rnn_states_critic[s + 1], rnn_states_critic)
        self._put(self.actions[s], actions)
        self._put(self.action_log_probs[s], action_log_probs)
        self._put(self.value_preds[s], value_preds)
        self._put(self.rewards[s], rewards)
        self._put(self.masks[s + 1], masks)
        if bad_masks is not None:
            self._put(self.bad_masks[s + 1], bad_masks)
        if active_masks is not None:
            self._put(self.active_masks[s], active_masks)
        if available_actions is not None:
            self._put(self.available_actions[s], available_actions)
        self.step = (s + 1) % self.episode_length

    def after_update(self):
        """shared_buffer.py:114-131: slot T of every carried array becomes slot 0 — one launch (mappo_copy_batch)."""
        arrs = [a for a in (self.share_obs, self.obs, self.rnn_states, self.rnn_states_critic, self.masks, self.bad_masks,
                            self.active_masks, self.available_actions) if a is not None]
        ops.copy_batch([(a[0], a[-1]) for a in arrs])

    def chooseafter_update(self):
        for arr in (self.rnn_states, self.rnn_states_critic, self.masks, self.bad_masks):
            arr[0].copy_(arr[-1])

    # ---- returns (shared_buffer.py:168-224) --------------------------------------------------------------
    def compute_returns(self, next_value, value_normalizer=None):
        T = self.episode_length
        R = self.n_rollout_threads * self.num_agents
        nv = to_device_f32(next_value, self.device).reshape(R)
        vn_state = None
        if self._use_popart:
            raise NotImplementedError("use_popart: PopArt.update raises in the reference itself (SURVEY.md §8c)")
        if self._use_valuenorm:
            if value_normalizer is None:
                raise ValueError("use_valuenorm needs the trainer's value_normalizer")
            vn_state = value_normalizer.state
        ops.gae_scan(self.rewards.view(T, R), self.value_preds.view(T + 1, R), nv, self.masks.view(T + 1, R),
                     self.bad_masks.view(T + 1, R), self.returns.view(T + 1, R), vn_state, self.gamma, self.gae_lambda,
                     self._use_gae, self._use_proper_time_limits)

    # ---- minibatch index arithmetic ----------------------------------------------------------------------
    def _randperm(self, n):
        if self.perm_device == "cpu":       # the reference's stream: torch.randperm on the CPU generator
            return torch.randperm(n).to(self.device, non_blocking=True)
        return torch.randperm(n, device=self.device)

    def _dims(self):
        T, N, M = self.episode_length, self.n_rollout_threads, self.num_agents
        return T, N * M

    def feed_forward_rows(self, num_mini_batch=None, mini_batch_size=None, rand=None):
        """Row indices of `feed_forward_generator` (shared_buffer.py:233-247): int32 tensors of flat rows t*R + r."""
        T, R = self._dims()
        batch_size = T * R
        if mini_batch_size is None:
            assert batch_size >= num_mini_batch, (
                f"PPO requires the number of processes ({self.n_rollout_threads}) * number of steps ({T}) * number of "
                f"agents ({self.num_agents}) = {batch_size} to be greater than or equal to the number of PPO mini "
                f"batches ({num_mini_batch}).")
            mini_batch_size = batch_size // num_mini_batch
        if rand is None:
            rand = self._randperm(batch_size)
        rand = torch.as_tensor(rand, device=self.device)
        return [rand[i * mini_batch_size:(i + 1) * mini_batch_size].to(torch.int32).contiguous()
                for i in range(num_mini_batch)]

    def recurrent_rows(self, num_mini_batch, data_chunk_length, rand=None):
        """Row indices of `recurrent_generator` (shared_buffer.py:385-494): the reference re-orders to
        q = (n*M+m)*T + t (`_cast`, :10-11), cuts chunks [iL, iL+L) of that flat order (chunks straddle two
        series when T % L != 0 — kept) and stacks them time-major (L, mbs).  Returns [(rows[L*mbs], h0_rows[mbs])]."""
        T, R = self._dims()
        L = data_chunk_length
        data_chunks = (T * R) // L
        mbs = data_chunks // num_mini_batch
        if rand is None:
            rand = self._randperm(data_chunks)
        rand = torch.as_tensor(rand, device=self.device).to(torch.int64)
        steps = torch.arange(L, device=self.device, dtype=torch.int64)
        out = []
        for k in range(num_mini_batch):
            c = rand[k * mbs:(k + 1) * mbs]
            q = (c[None, :] * L + steps[:, None]).reshape(-1)
            rows = (q % T) * R + q // T
            q0 = c * L
            h0 = (q0 % T) * R + q0 // T
            out.append((rows.to(torch.int32).contiguous(), h0.to(torch.int32).contiguous()))
        return out

    def recurrent_rows_epochs(self, n_epochs, num_mini_batch, data_chunk_length):
        """`recurrent_rows` for all ppo epochs of a train() call with two random-number launches and ONE index kernel
        (mappo_recurrent_rows) instead of ~25 small launches per epoch: the epochs' chunk permutations are the argsort of
        [n_epochs, data_chunks] device uniforms (device permutation stream only — `perm_device="cpu"` keeps the reference's
        per-epoch torch.randperm).  Returns [[(rows, h0_rows) per minibatch] per epoch]."""
        T, R = self._dims()
        data_chunks = (T * R) // data_chunk_length
        # one flat sort instead of a segmented one (~25 launches): epoch e's keys lie in [e, e + 1), so after sorting the
        # flat array positions [e*chunks, (e+1)*chunks) hold epoch e's chunk indices in random order (float64 keys: no ties)
        e_idx = torch.arange(n_epochs, device=self.device, dtype=torch.float64)[:, None]
        keys = torch.rand(n_epochs, data_chunks, device=self.device, dtype=torch.float64) + e_idx
        perm = keys.view(-1).argsort().view(n_epochs, data_chunks) - (e_idx * data_chunks).to(torch.int64)
        rows, h0 = ops.recurrent_rows(perm, data_chunk_length, T, R, num_mini_batch)
        return [[(rows[e, k], h0[e, k]) for k in range(num_mini_batch)] for e in range(n_epochs)]

    def naive_recurrent_rows(self, num_mini_batch, perm=None):
        """Row indices of `naive_recurrent_generator` (shared_buffer.py:288-383): whole episodes per (n, m)."""
        T, R = self._dims()
        assert R >= num_mini_batch, (
            f"PPO requires the number of processes ({self.n_rollout_threads})* number of agents ({self.num_agents}) "
            f"to be greater than or equal to the number of PPO mini batches ({num_mini_batch}).")
        n = R // num_mini_batch
        if perm is None:
            perm = self._randperm(R)
        perm = torch.as_tensor(perm, device=self.device).to(torch.int64)
        t = torch.arange(T, device=self.device, dtype=torch.int64)
        out = []
        for start in range(0, R, n):
            cols = perm[start:start + n]
            if cols.numel() < n:
                break
            rows = (t[:, None] * R + cols[None, :]).reshape(-1)
            out.append((rows.to(torch.int32).contiguous(), cols.to(torch.int32).contiguous()))
        return out

    # ---- the reference's generators (materialised 12-tuples) ---------------------------------------------
    def _flat(self, arr):
        T = self.episode_length
        return arr[:T].reshape(T * self.n_rollout_threads * self.num_agents, *arr.shape[3:])

    def _sample(self, rows, advantages, h0_rows=None):
        rows = rows.long()
        hr = rows if h0_rows is None else h0_rows.long()
        g = lambda a: self._flat(a)[rows]
        adv = torch.as_tensor(advantages, device=self.device, dtype=torch.float32).reshape(-1, 1)
        avail = g(self.available_actions) if self.available_actions is not None else None
        return (g(self.share_obs), g(self.obs), self._flat(self.rnn_states)[hr], self._flat(self.rnn_states_critic)[hr],
                g(self.actions), g(self.value_preds), g(self.returns), g(self.masks), g(self.active_masks),
                g(self.action_log_probs), adv[rows], avail)

    def feed_forward_generator(self, advantages, num_mini_batch=None, mini_batch_size=None):
        for rows in self.feed_forward_rows(num_mini_batch, mini_batch_size):
            yield self._sample(rows, advantages)

    def recurrent_generator(self, advantages, num_mini_batch, data_chunk_length):
        for rows, h0 in self.recurrent_rows(num_mini_batch, data_chunk_length):
            yield self._sample(rows, advantages, h0)

    def naive_recurrent_generator(self, advantages, num_mini_batch):
        for rows, cols in self.naive_recurrent_rows(num_mini_batch):
            yield self._sample(rows, advantages, cols)
