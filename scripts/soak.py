"""Soak run (GPU box): many iterations of every BASELINE config through the hipGraph path; every train_info value must stay finite and the
parameters must keep moving.  usage: python scripts/soak.py [c2:300 c2r:100 c3:15 c4:15 c5:4]"""
import json, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as BC
plan = [a.split(":") for a in (sys.argv[1:] or ["c1:300", "c2:300", "c2r:100", "c3:15", "c4:15", "c5:4"])]
for name, n in plan:
    r = BC.make_runner(name)
    r.warmup()
    p0 = r.trainer.policy.flat_params.clone()
    bad = 0
    for i in range(int(n)):
        info, _ = r.run_episode()
        if not all(math.isfinite(float(v)) for v in info.values()):
            bad += 1
    torch.cuda.synchronize()
    moved = float((r.trainer.policy.flat_params - p0).abs().max())
    print(json.dumps(dict(config=name, iterations=int(n), non_finite_iterations=bad, max_param_change=moved,
                          last={k: round(float(v), 6) for k, v in info.items()})), flush=True)
    assert bad == 0 and moved > 0 and math.isfinite(moved), (name, bad, moved)
    del r
    torch.cuda.empty_cache()
print("soak ok")
