"""Diagnostic (GPU box): the three launches of a wide-input critic update (wide_l1_fwd16 -> mlp_update16x -> wide_l1_bwd16) at
config-5 width, timed as a whole with events; run it under `rocprofv3 --kernel-trace --stats` (scripts/prof_any.sh) for the
per-kernel split.  usage: python scripts/time_wide.py [B] [D]      (MAPPO_HIP_LIB selects an experiment build)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mappo_amd import ops
class A_: pass
a = A_(); a.clip_param=0.2; a.entropy_coef=0.01; a.value_loss_coef=1.0; a.huber_delta=10.0; a.use_huber_loss=True; a.use_clipped_value_loss=True; a.use_policy_active_masks=True; a.use_value_active_masks=True; a.use_valuenorm=True
cfg = ops.ppo_cfg(a)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1638400
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
torch.manual_seed(0)
dc = ops.net_desc(D, 1)
Pc = ops.net_param_count(dc)
P = ((Pc + 255) // 256) * 256
pc = torch.randn(Pc, device="cuda") * 0.1
sobs = torch.randn(B, D, device="cuda")
ret = torch.randn(B, device="cuda"); active = (torch.rand(B, device="cuda") > 0.1).float()
mom = torch.zeros(4, dtype=torch.float64, device="cuda"); ops.minibatch_moments(ret, active, None, B, mom)
vold = torch.randn(B, device="cuda"); vn = torch.tensor([0., 1., 1.], device="cuda")
nd = max(ops.mlp_backward_slabs(B), ops.wide_l1_slabs(B))
slabs = torch.zeros(nd, P, device="cuda"); pdc = ops.update_partials("cuda")
def run():
    ops.critic_update(pc, dc, sobs, None, B, vold, ret, active, vn, mom, cfg, slabs, P, 0, pdc)
for _ in range(3): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
n = 10
e0.record()
for _ in range(n): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"{os.environ.get('MAPPO_HIP_LIB', 'product')}: B={B} D={D}  {ms:.3f} ms per update  ({B * D * 4 / ms / 1e6:.0f} GB/s of x per pass-equivalent, "
      f"{2 * 2 * B * D * 64 / ms / 1e9:.1f} TFLOP/s layer-1 fwd+wgrad)  |g|={slabs.sum(0).norm().item():.6f}")
