"""Workload for rocprofv3 --pmc / --kernel-trace runs: a few dual actor+critic update launches (mappo_actor_critic_update) at
BASELINE config-2 buffer size, product library.  PMC_B overrides the sample count."""
import os, sys
import torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from mappo_amd import ops
class A_: pass
a = A_(); a.clip_param=0.2; a.entropy_coef=0.01; a.value_loss_coef=1.0; a.huber_delta=10.0; a.use_huber_loss=True; a.use_clipped_value_loss=True; a.use_policy_active_masks=True; a.use_value_active_masks=True; a.use_valuenorm=True
cfg = ops.ppo_cfg(a)
B = int(os.environ.get('PMC_B', 76800))
torch.manual_seed(0)
da, dc = ops.net_desc(18, 5), ops.net_desc(54, 1)
Pa, Pc = ops.net_param_count(da), ops.net_param_count(dc)
col_c = ((Pa + 255) // 256) * 256
P = col_c + ((Pc + 255) // 256) * 256
pa = torch.randn(Pa, device="cuda") * 0.1; pc = torch.randn(Pc, device="cuda") * 0.1
obs = torch.randn(B, 18, device="cuda"); sobs = torch.randn(B, 54, device="cuda")
ret = torch.randn(B, device="cuda"); active = (torch.rand(B, device="cuda") > 0.1).float()
mom = torch.zeros(4, dtype=torch.float64, device="cuda"); ops.minibatch_moments(ret, active, None, B, mom)
av = (torch.rand(B, 5, device="cuda") > 0.2).float(); av[:, 0] = 1
act = torch.zeros(B, device="cuda"); olp = -torch.rand(B, device="cuda") - 1
adv = torch.randn(B, device="cuda"); vold = torch.randn(B, device="cuda"); vn = torch.tensor([0., 1., 1.], device="cuda")
nd = ops.dual_update_slabs(da, dc, B)
slabs = torch.zeros(nd, P, device="cuda"); pda, pdc = ops.update_partials("cuda"), ops.update_partials("cuda")
for _ in range(int(os.environ.get('PMC_N', 6))):
    ops.actor_critic_update(pa, da, obs, pc, dc, sobs, None, B, av, act, olp, adv, active, vold, ret, vn, mom, cfg, slabs, P, 0, col_c, pda, pdc)
torch.cuda.synchronize()
# calibration for FETCH_SIZE / WRITE_SIZE (MI355X_MICROARCH.md, HBM section): a streaming copy of a known byte count, 16 B per lane
if os.environ.get('PMC_CALIB', '1') == '1':
    src = torch.randn(64 * 1024 * 1024 // 4, device="cuda"); dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)
    torch.cuda.synchronize()
print("done")
