"""Diagnostic: rollout / train split of one iteration of a BASELINE config (device events).
usage: python scripts/phase_split.py [c2r c3 ...]"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_configs import CONFIGS, make_runner

for name in (sys.argv[1:] or ["c2r"]):
    r = make_runner(name)
    r.warmup()
    for _ in range(3):
        r.run_episode()
    n = 3
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ph = [0.0, 0.0]
    for i in range(n):
        ev[0].record(); r.rollout(); ev[1].record(); r.train(); ev[2].record()
        torch.cuda.synchronize()
        ph[0] += ev[0].elapsed_time(ev[1]) / n; ph[1] += ev[1].elapsed_time(ev[2]) / n
    print(json.dumps(dict(config=name, rollout_ms=ph[0], train_ms=ph[1])), flush=True)
    del r
    torch.cuda.empty_cache()
