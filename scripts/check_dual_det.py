"""Diagnostic (GPU box): determinism of the dual actor+critic update launch with a row gather — repeated launches on the same
inputs must give bit-identical slabs / statistics, and agree with the two single-network launches."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mappo_amd import ops
class A_: pass
a = A_(); a.clip_param=0.2; a.entropy_coef=0.01; a.value_loss_coef=1.0; a.huber_delta=10.0; a.use_huber_loss=True; a.use_clipped_value_loss=True; a.use_policy_active_masks=True; a.use_value_active_masks=True; a.use_valuenorm=True
cfg = ops.ppo_cfg(a)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 76800
torch.manual_seed(0)
da, dc = ops.net_desc(18, 5), ops.net_desc(54, 1)
Pa, Pc = ops.net_param_count(da), ops.net_param_count(dc)
col_c = ((Pa + 255) // 256) * 256
P = col_c + ((Pc + 255) // 256) * 256
pa = torch.randn(Pa, device="cuda") * 0.1; pc = torch.randn(Pc, device="cuda") * 0.1
NR = B + 1000
obs = torch.randn(NR, 18, device="cuda"); sobs = torch.randn(NR, 54, device="cuda")
ret = torch.randn(NR, device="cuda"); active = (torch.rand(NR, device="cuda") > 0.1).float()
rows = torch.randperm(NR, device="cuda")[:B].to(torch.int32).contiguous()
mom = torch.zeros(4, dtype=torch.float64, device="cuda"); ops.minibatch_moments(ret, active, rows, B, mom)
av = (torch.rand(NR, 5, device="cuda") > 0.2).float(); av[:, 0] = 1
act = torch.zeros(NR, device="cuda"); olp = -torch.rand(NR, device="cuda") - 1
adv = torch.randn(NR, device="cuda"); vold = torch.randn(NR, device="cuda"); vn = torch.tensor([0., 1., 1.], device="cuda")
nd = ops.dual_update_slabs(da, dc, B)
ns = ops.mlp_backward_slabs(B)
def dual():
    slabs = torch.zeros(max(nd, ns), P, device="cuda"); pda, pdc = ops.update_partials("cuda"), ops.update_partials("cuda")
    ops.actor_critic_update(pa, da, obs, pc, dc, sobs, rows, B, av, act, olp, adv, active, vold, ret, vn, mom, cfg, slabs, P, 0, col_c, pda, pdc)
    stats = torch.zeros(6, dtype=torch.float64, device="cuda")
    ops.update_stats(pda, nd, pdc, nd, mom, cfg, stats)
    torch.cuda.synchronize()
    global last_pdc
    last_pdc = pdc.cpu().numpy().reshape(-1, 4).copy()
    return slabs.double().sum(0).cpu().numpy(), stats.cpu().numpy()
def single():
    slabs = torch.zeros(max(nd, ns), P, device="cuda"); pda, pdc = ops.update_partials("cuda"), ops.update_partials("cuda")
    ops.actor_update(pa, da, obs, rows, B, av, act, olp, adv, active, mom, cfg, slabs, P, 0, pda)
    ops.critic_update(pc, dc, sobs, rows, B, vold, ret, active, vn, mom, cfg, slabs, P, col_c, pdc)
    stats = torch.zeros(6, dtype=torch.float64, device="cuda")
    ops.update_stats(pda, ns, pdc, ns, mom, cfg, stats)
    torch.cuda.synchronize()
    return slabs.double().sum(0).cpu().numpy(), stats.cpu().numpy()
g0, s0 = dual()
p0 = last_pdc.copy()
gs, ss = single()
print("single vs dual: grad", np.abs(g0 - gs).max() / np.abs(gs).max(), "stats", np.abs(s0 - ss).max(), ss)
bad = 0
for i in range(6):
    g, s = dual()
    dg, ds = np.abs(g - g0).max() / np.abs(g0).max(), np.abs(s - s0).max()
    if dg > 0 or ds > 0:
        bad += 1
        d = np.abs(last_pdc - p0)
        rows_ = np.nonzero(d.max(1))[0]
        print("run", i, "differs: grad", dg, "stats", ds, "critic partial rows that differ:", rows_[:12], "n", len(rows_), "cols", np.nonzero(d.max(0))[0],
              "example", last_pdc[rows_[0]] if len(rows_) else None, p0[rows_[0]] if len(rows_) else None)
print("dual launches that differ from the first:", bad, "of 30")
for i in range(10):
    g, s = single()
    dg, ds = np.abs(g - gs).max() / np.abs(gs).max(), np.abs(s - ss).max()
    if dg > 0 or ds > 0: print("single run", i, "differs", dg, ds)
