"""Informational throughput of the other BASELINE.json configs on one GPU (not the bench.py contract line).
usage: python scripts/bench_configs.py [c1 c3 c4 c5 ...]"""
import json, sys, time
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from mappo_amd.config import get_config
from mappo_amd.envs.synthetic import SyntheticMPEEnv, SyntheticSMACEnv
from mappo_amd.runner.shared.mpe_runner import MPERunner
from mappo_amd.runner.shared.smac_runner import SMACRunner

CONFIGS = {
    # name: (runner, T, N per GPU, M, D_o, D_s, A, recurrent, ppo_epoch, nmb, gain, steps)
    "c1": ("mpe", 25, 8, 3, 18, 54, 5, False, 10, 1, 0.01, 20),
    "c2": ("mpe", 25, 1024, 3, 18, 54, 5, False, 10, 1, 0.01, 10),
    "c2r": ("mpe", 25, 1024, 3, 18, 54, 5, True, 10, 1, 0.01, 5),       # train_mpe_spread.sh actually runs rmappo
    "c3": ("smac", 400, 256, 3, 30, 48, 9, True, 15, 1, 0.01, 2),
    "c4": ("smac", 400, 64, 10, 176, 322, 18, True, 5, 2, 1.0, 2),       # 512 threads / 8 GPUs
    "c5": ("mpe5", 400, 256, 64, 512, 512, 5, False, 5, 1, 0.01, 1),      # 2048 threads / 8 GPUs, D_s = 512
}

def make_runner(name):
    kind, T, N, M, D, S, A, rec, epochs, nmb, gain, steps = CONFIGS[name]
    a = get_config().parse_known_args([])[0]
    a.algorithm_name = "rmappo" if rec else "mappo"
    a.use_recurrent_policy, a.use_naive_recurrent_policy = rec, False
    a.episode_length, a.n_rollout_threads, a.ppo_epoch, a.num_mini_batch, a.gain = T, N, epochs, nmb, gain
    a.lr = a.critic_lr = 7e-4 if kind.startswith("mpe") else 5e-4
    a.env_name = "MPE" if kind.startswith("mpe") else "StarCraft2"
    a.use_centralized_V = kind != "mpe5"
    torch.manual_seed(1)
    dev = torch.device("cuda:0")
    if kind == "smac":
        env = SyntheticSMACEnv(N, M, D, S, A, seed=1, device=dev, pool_steps=T); R = SMACRunner
    else:
        env = SyntheticMPEEnv(N, M, D, A, T, seed=1, device=dev); R = MPERunner
        if kind == "mpe5":
            env.share_observation_space = [[D] for _ in range(M)]
    return R(dict(all_args=a, envs=env, eval_envs=None, num_agents=M, device=dev, run_dir=None))


def run(name):
    kind, T, N, M, D, S, A, rec, epochs, nmb, gain, steps = CONFIGS[name]
    r = make_runner(name)
    r.warmup()
    for _ in range(2):
        r.run_episode()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        info, _ = r.run_episode()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = dict(config=name, T=T, N=N, M=M, obs=D, share_obs=S, A=A, policy="GRU" if rec else "MLP", ppo_epoch=epochs, num_mini_batch=nmb,
               ms_per_iteration=1e3 * dt, agent_steps_per_s=T * N * M / dt, mem_GiB=torch.cuda.max_memory_allocated() / 2 ** 30,
               train_info={k: round(float(v), 6) for k, v in info.items()})
    print(json.dumps(out), flush=True)
    del r
    torch.cuda.empty_cache()

if __name__ == "__main__":
    for n in (sys.argv[1:] or ["c1", "c2", "c2r", "c3", "c4"]):
        run(n)
