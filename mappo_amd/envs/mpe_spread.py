"""GPU-vectorised MPE `simple_spread` behind the reference's vec-env contract:

    reset() -> obs [N, M, 18];  step(actions_env) -> (obs [N, M, 18], rewards [N, M, 1], dones [N, M] bool, infos)

(`onpolicy/envs/env_wrappers.py:262-272`, `onpolicy/envs/mpe/environment.py:117-148`).  N environments x M agents x L
landmarks are stepped by ONE kernel launch (csrc/mpe_env.hip: action -> force, pairwise soft-collision force, damping +
integration, shared reward, observations, time-limit done, reset-on-done) and everything stays in HBM, so the real MPE
configuration trains with no process boundary and the runner can capture a whole episode into a hipGraph (`graph_safe`).

`step` takes the reference's one-hot `actions_env [N, M, 5]` or — `accepts_index_actions` — the buffer's own action indices
`[N, M, 1]` (fp32), which saves the one-hot gather.  Physics run in float64 as in the reference's NumPy code; initial
states come from a counter-based Philox stream (seed, episode, env), not from NumPy's global generator."""
import torch

from .. import ops
from ..utils.util import Discrete


class SimpleSpreadVecEnv:
    graph_safe = True               # step() is one kernel launch on the current stream, no host-side data dependence
    accepts_device_actions = True
    accepts_index_actions = True    # step(actions [N, M, 1] fp32 indices) is accepted besides the one-hot actions_env
    consumes_actions = True

    def __init__(self, n_rollout_threads, num_agents=3, num_landmarks=3, episode_length=25, seed=1, device="cuda"):
        self.N, self.M, self.L, self.T = int(n_rollout_threads), int(num_agents), int(num_landmarks), int(episode_length)
        self.device = torch.device(device)
        self.seed = int(seed)
        self.obs_dim = 4 + 2 * self.L + 4 * (self.M - 1)             # 18 for 3 agents / 3 landmarks
        self.observation_space = [[self.obs_dim] for _ in range(self.M)]
        self.share_observation_space = [[self.obs_dim * self.M] for _ in range(self.M)]
        self.action_space = [Discrete(5) for _ in range(self.M)]      # world.dim_p * 2 + 1 (environment.py:63-64)
        f64 = dict(dtype=torch.float64, device=self.device)
        self.agent_pos = torch.zeros(self.N, self.M, 2, **f64)
        self.agent_vel = torch.zeros(self.N, self.M, 2, **f64)
        self.landmark_pos = torch.zeros(self.N, self.L, 2, **f64)
        self.tstep = torch.zeros(self.N, dtype=torch.int32, device=self.device)
        self.episode = torch.zeros(self.N, dtype=torch.int64, device=self.device)
        # two output sets: the runner's fused step reads the previous env output while this step writes the next one
        self._out = [(torch.empty(self.N, self.M, self.obs_dim, device=self.device), torch.empty(self.N, self.M, 1, device=self.device),
                      torch.empty(self.N, self.M, dtype=torch.bool, device=self.device)) for _ in range(2)]
        self._k = 0

    def set_state(self, agent_pos, agent_vel, landmark_pos, tstep=0):
        """Test hook: load explicit states (float64 arrays / tensors [N, M|L, 2])."""
        self.agent_pos.copy_(torch.as_tensor(agent_pos, dtype=torch.float64))
        self.agent_vel.copy_(torch.as_tensor(agent_vel, dtype=torch.float64))
        self.landmark_pos.copy_(torch.as_tensor(landmark_pos, dtype=torch.float64))
        self.tstep.fill_(int(tstep))

    def reset(self):
        obs = self._out[self._k][0]
        ops.mpe_spread_reset(self.agent_pos, self.agent_vel, self.landmark_pos, self.tstep, self.episode, obs, self.N, self.M, self.L,
                             self.seed)
        self._k ^= 1
        return obs

    def step(self, actions):
        obs, rew, dones = self._out[self._k]
        self._k ^= 1
        a = actions if torch.is_tensor(actions) else torch.as_tensor(actions)
        a = a.to(self.device, torch.float32)
        if a.dim() == 3 and a.shape[-1] == 5:
            mode = 0
        elif a.numel() == self.N * self.M:
            mode = 1
        else:
            raise ValueError(f"actions of shape {tuple(a.shape)}: expected one-hot [N, M, 5] or indices [N, M(, 1)]")
        ops.mpe_spread_step(self.agent_pos, self.agent_vel, self.landmark_pos, self.tstep, self.episode, a.contiguous(), mode, obs, rew,
                            dones.view(torch.uint8), self.N, self.M, self.L, self.T, self.seed)
        return obs, rew, dones, None

    def close(self):
        pass
