"""PMC workload: R_MAPPO.train of a recurrent policy at BASELINE configs[2] size (T=400, N=256, 3 agents, obs 30 / state 48 / 9 actions,
chunks of 10: 30 720 sequences per network) with 2 ppo epochs, eager launches — the gru16_* kernels at the size bench_configs c3 runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scripts.bench_configs import make_runner

r = make_runner(os.environ.get("PMC_CFG", "c3"))
r.trainer.ppo_epoch = 2
r.trainer._use_graph = False
r.warmup()
r.rollout()
for _ in range(2):
    info = r.train()
torch.cuda.synchronize()
print(info)
