"""Diagnostic (GPU box): back-to-back timing of the actor / critic update kernels of one or more builds.
usage: python scripts/time_update.py tag=libpath [tag=libpath ...]   (each lib is timed in its own subprocess)"""
import os, subprocess, sys
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from mappo_amd import _lib
_lib.LIB_PATH = os.environ["TU_LIB"]
from mappo_amd import ops
class A_: pass
a = A_(); a.clip_param=0.2; a.entropy_coef=0.01; a.value_loss_coef=1.0; a.huber_delta=10.0; a.use_huber_loss=True; a.use_clipped_value_loss=True; a.use_policy_active_masks=True; a.use_value_active_masks=True; a.use_valuenorm=True
cfg = ops.ppo_cfg(a)
B = 76800
torch.manual_seed(0)
out = []
for name, D, A in (("actor", 18, 5), ("critic", 54, 1)):
    desc = ops.net_desc(D, A); P = ops.net_param_count(desc)
    params = torch.randn(P, device="cuda") * 0.1
    x = torch.randn(B, D, device="cuda"); ns = ops.mlp_backward_slabs(B)
    slabs = torch.zeros(ns, P, device="cuda"); part = ops.update_partials("cuda")
    ret = torch.randn(B, device="cuda"); active = torch.ones(B, device="cuda")
    mom = torch.zeros(4, dtype=torch.float64, device="cuda"); ops.minibatch_moments(ret, active, None, B, mom)
    av = torch.ones(B, A, device="cuda"); act = torch.randint(0, A, (B,), device="cuda").float(); olp = -torch.rand(B, device="cuda") - 1
    adv = torch.randn(B, device="cuda"); vold = torch.randn(B, device="cuda"); vn = torch.tensor([0., 1., 1.], device="cuda")
    def run():
        if name == "actor":
            ops.actor_update(params, desc, x, None, B, av, act, olp, adv, active, mom, cfg, slabs, P, 0, part)
        else:
            ops.critic_update(params, desc, x, None, B, vold, ret, active, vn, mom, cfg, slabs, P, 0, part)
    for _ in range(10): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): run()
    e1.record(); torch.cuda.synchronize()
    out.append(f"{name} {e0.elapsed_time(e1) * 10:.1f} us |g|={slabs.sum(0).norm().item():.5f}")
print(os.environ["TU_TAG"], " | ".join(out))
'''
for spec in sys.argv[1:]:
    tag, lib = spec.split("=", 1)
    env = dict(os.environ, TU_LIB=os.path.abspath(lib), TU_TAG=tag)
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
