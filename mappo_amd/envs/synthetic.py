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
    consumes_actions = False   # the synthetic dynamics ignore the actions: the runner need not build one-hot actions

    def __init__(self, n_rollout_threads, num_agents=3, obs_dim=18, n_actions=5, episode_length=25, seed=1, device="cuda"):
        self.N, self.M, self.D, self.A, self.T = n_rollout_threads, num_agents, obs_dim, n_actions, episode_length
        self.device = torch.device(device)
        self.seed = seed
        self.observation_space = [[obs_dim] for _ in range(num_agents)]
        self.share_observation_space = [[obs_dim * num_agents] for _ in range(num_agents)]
        self.action_space = [Discrete(n_actions) for _ in range(num_agents)]
        self.t = 0
        self._pool = None
        self._done_true = torch.ones(self.N, self.M, dtype=torch.bool, device=self.device)
        self._done_false = torch.zeros(self.N, self.M, dtype=torch.bool, device=self.device)
        with torch.cuda.device(self.device):
            torch.cuda.manual_seed(seed)       # the device's default generator: the one hipGraph capture can advance

    def reset(self):
        self.t = 0
        return torch.randn(self.N, self.M, self.D, device=self.device)

    def step(self, actions_env=None):
        i = self.t % self.T
        if i == 0:          # one generator launch per episode: T x (obs + shared reward); episode-aligned, so a captured
            self._pool = torch.randn(self.T, self.N, self.M * self.D + 1, device=self.device)   # episode graph redraws it
        self.t += 1
        blk = self._pool[i]
        obs = blk[:, :self.M * self.D].view(self.N, self.M, self.D)
        rewards = blk[:, self.M * self.D:].view(self.N, 1, 1).expand(self.N, self.M, 1)
        dones = self._done_true if (self.t % self.T == 0) else self._done_false   # fixed per step index of an episode
        return obs, rewards, dones, None

    def close(self):
        pass


class SyntheticSMACEnv:
    """SMAC-shaped synthetic vec-env: `reset() -> (obs, share_obs, avail)`,
    `step(actions [N, M, 1]) -> (obs, share_obs, rewards, dones [N, M], bad_transition [N, M] bool, avail)`.
    Agents die with probability `p_death` per step, an env terminates with probability `p_term` per step (all its
    agents report done; it restarts alive), action 0 is always available and the others with probability 0.7;
    `bad_transition` is always False.  Everything stays on the device."""
    graph_safe = True

    def __init__(self, n_rollout_threads, num_agents=3, obs_dim=30, share_dim=48, n_actions=9, p_death=0.01, p_term=1.0 / 60,
                 seed=1, device="cuda", pool_steps=0):
        """pool_steps = P > 0: every P-th step() draws the next P steps in ONE launch (mappo_synth_smac_pool) and the steps hand out
        views of that pool — like SyntheticMPEEnv's per-episode block (pass the episode length: a captured episode graph then
        contains the pool launch of its first step).  0: one launch per step."""
        self.pool_steps = int(pool_steps)
        self._t = 0
        self.N, self.M, self.D, self.S, self.A = n_rollout_threads, num_agents, obs_dim, share_dim, n_actions
        self.p_death, self.p_term = p_death, p_term
        self.device = torch.device(device)
        self.observation_space = [[obs_dim] for _ in range(num_agents)]
        self.share_observation_space = [[share_dim] for _ in range(num_agents)]
        self.action_space = [Discrete(n_actions) for _ in range(num_agents)]
        self.dead = torch.zeros(self.N, self.M, dtype=torch.bool, device=self.device)
        self.seed = int(seed)
        with torch.cuda.device(self.device):
            torch.cuda.manual_seed(seed)

    def _draw(self):
        N, M, dev = self.N, self.M, self.device
        obs = torch.randn(N, M, self.D, device=dev)
        share = torch.randn(N, M, self.S, device=dev)
        avail = (torch.rand(N, M, self.A, device=dev) < 0.7).to(torch.float32)
        avail[:, :, 0] = 1.0
        return obs, share, avail

    def reset(self):
        self.dead.zero_()
        return self._draw()

    def _fused_state(self):
        # buffers of the one-launch step (csrc/synth_env.hip); the runner copies what it is handed before the next step
        if not hasattr(self, "_f"):
            N, M, dev = self.N, self.M, self.device
            self._f = dict(obs=torch.empty(N, M, self.D, device=dev), share=torch.empty(N, M, self.S, device=dev),
                           avail=torch.empty(N, M, self.A, device=dev), rew=torch.empty(N, device=dev),
                           dones=torch.zeros(N, M, dtype=torch.bool, device=dev), bad=torch.zeros(N, M, dtype=torch.bool, device=dev),
                           ctr=torch.tensor([1] + [0] * 33, dtype=torch.int64, device=dev))       # {Philox counter, 33 tickets}: the kernel advances it
        return self._f

    def step(self, actions=None):
        N, M, dev = self.N, self.M, self.device
        if dev.type == "cuda" and self.pool_steps > 0:
            from mappo_amd import ops
            P = self.pool_steps
            if not hasattr(self, "_pool"):
                self._pool = dict(obs=torch.empty(P, N, M, self.D, device=dev), share=torch.empty(P, N, M, self.S, device=dev),
                                  avail=torch.empty(P, N, M, self.A, device=dev), rew=torch.empty(P, N, device=dev),
                                  dones=torch.zeros(P, N, M, dtype=torch.bool, device=dev), bad=torch.zeros(N, M, dtype=torch.bool, device=dev),
                                  ctr=torch.tensor([1] + [0] * 33, dtype=torch.int64, device=dev))
            q = self._pool
            i = self._t % P
            if i == 0:
                ops.synth_smac_pool(q["obs"], q["share"], q["avail"], q["rew"], self.dead, q["dones"], self.p_death, self.p_term,
                                    self.seed, q["ctr"])
            self._t += 1
            return q["obs"][i], q["share"][i], q["rew"][i].view(N, 1, 1).expand(N, M, 1), q["dones"][i], q["bad"], q["avail"][i]
        if dev.type == "cuda":
            from mappo_amd import ops
            f = self._fused_state()
            ops.synth_smac_step(f["obs"], f["share"], f["avail"], f["rew"], self.dead, f["dones"], self.p_death, self.p_term,
                                self.seed, f["ctr"])
            return f["obs"], f["share"], f["rew"].view(N, 1, 1).expand(N, M, 1), f["dones"], f["bad"], f["avail"]
        obs, share, avail = self._draw()
        rewards = torch.randn(N, 1, 1, device=dev).expand(N, M, 1)
        self.dead |= torch.rand(N, M, device=dev) < self.p_death
        term = torch.rand(N, device=dev) < self.p_term
        dones = self.dead | term.view(N, 1)
        self.dead &= ~term.view(N, 1)                       # a terminated env restarts with every agent alive
        bad = torch.zeros(N, M, dtype=torch.bool, device=dev)
        return obs, share, rewards, dones, bad, avail

    def close(self):
        pass
