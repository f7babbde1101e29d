"""SeparatedReplayBuffer resident in HBM — API of `onpolicy/utils/separated_buffer.py:13-393` (share_policy=False: one buffer
per agent, arrays `[T(+1), N, D]` without the agent dimension).

It IS the shared buffer of one agent: same memory order (`[T+1][N][1][D]` == `[T+1][N][D]`), same kernels (GAE scan, row
generators, the fused update kernels stream it in place), with the agent dimension squeezed out of every public array so that
callers written against the reference's separated buffer index it as they always did."""
from .shared_buffer import SharedReplayBuffer

_ARRAYS = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "value_preds", "returns", "available_actions", "actions",
           "action_log_probs", "rewards", "masks", "bad_masks", "active_masks")


class SeparatedReplayBuffer(SharedReplayBuffer):
    def __init__(self, args, obs_space, share_obs_space, act_space, device=None):
        super().__init__(args, 1, obs_space, share_obs_space, act_space, device=device)
        self.rnn_hidden_size = self.hidden_size                    # the reference's attribute name (separated_buffer.py:17)
        for name in _ARRAYS:
            arr = getattr(self, name)
            if arr is not None:
                setattr(self, name, arr.squeeze(2))                 # a view: [T(+1), N, 1, D] -> [T(+1), N, D]

    def _flat(self, arr):
        T = self.episode_length
        return arr[:T].reshape(T * self.n_rollout_threads, *arr.shape[2:])
