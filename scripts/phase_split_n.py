"""Diagnostic: rollout / train split of a BASELINE config with an overridden rollout-thread count.
usage: python scripts/phase_split_n.py c4 512"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as BC
name, n = sys.argv[1], int(sys.argv[2])
c = list(BC.CONFIGS[name]); c[2] = n; BC.CONFIGS[name] = tuple(c)
r = BC.make_runner(name)
r.warmup()
for _ in range(3):
    r.run_episode()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
ph = [0.0, 0.0]
for i in range(3):
    ev[0].record(); r.rollout(); ev[1].record(); r.train(); ev[2].record()
    torch.cuda.synchronize()
    ph[0] += ev[0].elapsed_time(ev[1]) / 3; ph[1] += ev[1].elapsed_time(ev[2]) / 3
print(json.dumps(dict(config=name, n_rollout_threads=n, rollout_ms=ph[0], train_ms=ph[1])), flush=True)
