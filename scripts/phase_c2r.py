"""Diagnostic: rollout / train split of the recurrent config-2 iteration (device events)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mappo_amd.config import get_config
from mappo_amd.envs.synthetic import SyntheticMPEEnv
from mappo_amd.runner.shared.mpe_runner import MPERunner
a = get_config().parse_known_args([])[0]
a.algorithm_name = "rmappo"; a.use_recurrent_policy, a.use_naive_recurrent_policy = True, False
a.episode_length, a.n_rollout_threads, a.ppo_epoch, a.num_mini_batch = 25, 1024, 10, 1
a.lr = a.critic_lr = 7e-4; a.env_name = "MPE"
torch.manual_seed(1)
dev = torch.device("cuda:0")
env = SyntheticMPEEnv(1024, 3, 18, 5, 25, seed=1, device=dev)
r = MPERunner(dict(all_args=a, envs=env, eval_envs=None, num_agents=3, device=dev, run_dir=None))
r.warmup()
for _ in range(3):
    r.run_episode()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
ph = [0.0, 0.0]
for i in range(5):
    ev[0].record(); r.rollout(); ev[1].record(); r.train(); ev[2].record()
    torch.cuda.synchronize()
    ph[0] += ev[0].elapsed_time(ev[1]) / 5; ph[1] += ev[1].elapsed_time(ev[2]) / 5
print(json.dumps(dict(rollout_ms=ph[0], train_ms=ph[1], rollout_graph=str(type(r._rollout_graph)))))
