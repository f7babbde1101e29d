"""Synthetic vec-envs with the reference's I/O contract, generating MPE- / SMAC-shaped
data on the GPU so that the measured path has no process boundary.

MPE contract (env_wrappers.py:262-272, envs/mpe/environment.py:117-148):
    reset() -> obs [N, M, D];  step(actions_env [N, M, A] one-hot) -> (obs, rewards [N, M, 1], dones [N, M] bool, infos)
    attributes observation_space[i], share_observation_space[i], action_space[i].
Observations ~ N(0,1); rewards ~ N(0,1) shared by the agents of a thread (shared_reward, environment.py:141-143);
every env reports done on each `episode_length`-th step (MPE time limit, environment.py:179-185)."""
import torch

from ..utils.util import Discrete


class SyntheticMPEEnv:
    graph_safe = True      # step() is a fixed sequence of device ops: the runner may capture an episode into a hipGraph

    def __init__(self, n_rollout_threads, num_agents=3, obs_dim=18, n_actions=5, episode_length=25, seed=1, device="cuda"):
        self.N, self.M, self.D, self.A, self.T = n_rollout_threads, num_agents, obs_dim, n_actions, episode_length
        self.device = torch.device(device)
        self.seed = seed
        self.observation_space = [[obs_dim] for _ in range(num_agents)]
        self.share_observation_space = [[obs_dim * num_agents] for _ in range(num_agents)]
        self.action_space = [Discrete(n_actions) for _ in range(num_agents)]
        self.t = 0
        self._done_true = torch.ones(self.N, self.M, dtype=torch.bool, device=self.device)
        self._done_false = torch.zeros(self.N, self.M, dtype=torch.bool, device=self.device)
        with torch.cuda.device(self.device):
            torch.cuda.manual_seed(seed)       # the device's default generator: the one hipGraph capture can advance

    def reset(self):
        self.t = 0
        return torch.randn(self.N, self.M, self.D, device=self.device)

    def step(self, actions_env=None):
        self.t += 1
        blk = torch.randn(self.N, self.M * self.D + 1, device=self.device)      # one launch: obs + shared reward
        obs = blk[:, :self.M * self.D].view(self.N, self.M, self.D)
        rewards = blk[:, self.M * self.D:].view(self.N, 1, 1).expand(self.N, self.M, 1)
        dones = self._done_true if (self.t % self.T == 0) else self._done_false   # fixed per step index of an episode
        return obs, rewards, dones, None

    def close(self):
        pass
