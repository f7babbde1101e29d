"""Diagnostic (GPU box): the fused MLP rollout step (actor act + critic value [+ insert]) back to back in a hipGraph at several
batch sizes, with and without the insert role: where the ~13 us of a config-2 step go."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mappo_amd import ops
torch.manual_seed(0)
D, M, A = 18, 3, 5
da, dc = ops.net_desc(D, A), ops.net_desc(D * M, 1)
pa = torch.randn(ops.net_param_count(da), device="cuda") * 0.1
pc = torch.randn(ops.net_param_count(dc), device="cuda") * 0.1
for N in (8, 64, 1024, 4096):
    B = N * M
    obs = torch.randn(N, M, D, device="cuda"); share = torch.randn(N, M, D * M, device="cuda")
    act, lp, val = torch.empty(B, device="cuda"), torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
    obs_dst, share_dst = torch.empty(N, M, D, device="cuda"), torch.empty(N, M, D * M, device="cuda")
    rew, dones = torch.randn(N, M, 1, device="cuda"), torch.zeros(N, M, dtype=torch.bool, device="cuda")
    rew_dst, mask_dst = torch.empty(N, M, 1, device="cuda"), torch.empty(N, M, 1, device="cuda")
    ctr = torch.zeros(1, dtype=torch.int64, device="cuda")
    ins = dict(obs_dst=obs_dst, share_dst=share_dst, rewards=(rew, M, 1), dones=(dones, M, 1), rew_dst=rew_dst, mask_dst=mask_dst, centralized=True)
    for tag, insert in (("with insert", ins), ("no insert", None)):
        def run():
            ops.rollout_step(pa, da, pc, dc, (obs, M * D, D), (share, M * D * M, D * M), M, B, None, False, 1, 0, ctr, act, lp, val, insert)
        for _ in range(3): run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(50): run()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        print(f"N={N:5d} ({B} rows) {tag:12s}: {e0.elapsed_time(e1) * 20:.1f} us per step (back to back in a graph)")
