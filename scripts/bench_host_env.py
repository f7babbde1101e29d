"""PCIe-inclusive rate: config-2 iteration with a HOST vec-env (NumPy observations / rewards / dones uploaded every
step, one-hot actions downloaded), i.e. the path a CPU environment takes; eager launches (a host env is not graph-safe)."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mappo_amd.config import get_config
from mappo_amd.runner.shared.mpe_runner import MPERunner


class HostMPEEnv:
    """NumPy twin of mappo_amd.envs.synthetic.SyntheticMPEEnv (same shapes / contract), outputs on the host."""
    consumes_actions = True          # (no graph_safe / accepts_device_actions: a plain CPU vec-env, NumPy in and out)

    def __init__(self, N, M, D, A, T, seed=1):
        self.N, self.M, self.D, self.T, self.t = N, M, D, T, 0
        self.rng = np.random.default_rng(seed)
        Disc = type("Discrete", (), {})
        sp = Disc(); sp.n = A
        self.action_space = [sp for _ in range(M)]
        self.observation_space = [[D] for _ in range(M)]
        self.share_observation_space = [[D * M] for _ in range(M)]

    def _draw(self):
        # the pool is drawn ONCE (NumPy's generator needs ~12 ms for an episode's 1.4 M normals — that would be the whole
        # measurement): the timed path is the staging of the env's arrays, not the env
        if not hasattr(self, "pool_obs"):
            self.pool_obs = self.rng.standard_normal((self.T + 1, self.N, self.M, self.D), dtype=np.float32)
            self.pool_rew = np.repeat(self.rng.standard_normal((self.T + 1, self.N, 1, 1), dtype=np.float32), self.M, axis=2)

    def reset(self):
        self.t = 0
        self._draw()
        return self.pool_obs[0]

    def step(self, actions):
        assert isinstance(actions, np.ndarray)
        self.t += 1
        k = self.t % (self.T + 1)
        dones = np.full((self.N, self.M), self.t % self.T == 0)
        return self.pool_obs[k], self.pool_rew[k], dones, None


def run(host_staging):
    a = get_config().parse_known_args([])[0]
    a.use_recurrent_policy = a.use_naive_recurrent_policy = False
    a.episode_length, a.n_rollout_threads, a.ppo_epoch, a.num_mini_batch = 25, 1024, 10, 1
    a.lr = a.critic_lr = 7e-4
    a.env_name = "MPE"
    a.host_staging = host_staging
    torch.manual_seed(1)
    env = HostMPEEnv(1024, 3, 18, 5, 25)
    r = MPERunner(dict(all_args=a, envs=env, eval_envs=None, num_agents=3, device=torch.device("cuda"), run_dir=None))
    r.warmup()
    for _ in range(3):
        r.run_episode()
    torch.cuda.synchronize()
    steps = 10
    t0 = time.perf_counter()
    for _ in range(steps):
        r.rollout()
    torch.cuda.synchronize()
    t_roll = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    for _ in range(steps):
        r.run_episode()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return dict(host_staging=host_staging, ms_per_iteration=1e3 * dt, ms_rollout=1e3 * t_roll, agent_steps_per_s=25 * 1024 * 3 / dt)


print(json.dumps(dict(workload="config 2, host NumPy vec-env (uploads + one-hot download every step, eager rollout)",
                      pinned_double_buffered=run(True), plain_torch_copies=run(False))))
