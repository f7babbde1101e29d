"""Diagnostic (GPU box): timing of the dual actor+critic update launch at BASELINE config 2 size, new 16-sample-tile kernel
(default) against the pair kernel (MAPPO_UPD16=0), each in its own subprocess; also cross-checks the two gradients.
usage: python scripts/time_dual.py [B] [--exp]"""
import os, subprocess, sys
CHILD = r'''
import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from mappo_amd import ops
class A_: pass
a = A_(); a.clip_param=0.2; a.entropy_coef=0.01; a.value_loss_coef=1.0; a.huber_delta=10.0; a.use_huber_loss=True; a.use_clipped_value_loss=True; a.use_policy_active_masks=True; a.use_value_active_masks=True; a.use_valuenorm=True
cfg = ops.ppo_cfg(a)
B = int(os.environ.get("TD_B", "76800"))
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
def run():
    ops.actor_critic_update(pa, da, obs, pc, dc, sobs, None, B, av, act, olp, adv, active, vold, ret, vn, mom, cfg, slabs, P, 0, col_c, pda, pdc)
for _ in range(10): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): run()
e1.record(); torch.cuda.synchronize()
g = slabs.sum(0)
stats = torch.zeros(6, dtype=torch.float64, device="cuda")
ops.update_stats(pda, nd, pdc, nd, mom, cfg, stats)
np.save(os.environ["TD_OUT"], np.concatenate([g.cpu().numpy().astype(np.float64), stats.cpu().numpy()]))
print(os.environ["TD_TAG"], f"rows {nd}  {e0.elapsed_time(e1) * 5:.1f} us per launch  |g|={g.norm().item():.6f}  stats={stats.cpu().numpy()}")
'''
import glob
import numpy as np
B = next((a for a in sys.argv[1:] if a.isdigit()), "76800")
variants = [("upd16", {"MAPPO_UPD16": "1"}), ("pair", {"MAPPO_UPD16": "0"})]
if "--split" in sys.argv:                       # sweep the actor's share of the 256 workgroups
    for na in (112, 116, 120, 124, 128, 132):
        variants.append((f"nA={na}", {"MAPPO_UPD16": "1", "MAPPO_UPD16_NA": str(na)}))
if "--exp" in sys.argv:                         # every experiment build of scripts/exp_build.py
    for lib in sorted(glob.glob("build_diag/lib_*.so")):
        variants.append((os.path.basename(lib)[4:-3], {"MAPPO_UPD16": "1", "MAPPO_HIP_LIB": os.path.abspath(lib)}))
ref = None
for tag, extra in variants:
    out = f"/tmp/td_{tag}.npy"
    env = dict(os.environ, TD_TAG=tag, TD_OUT=out, TD_B=B, **extra)
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
    try:
        a = np.load(out)
        if ref is None:
            ref = a
        else:
            print(f"   {tag}: max |g - g_upd16| / max|g| = {np.abs(a[:-6] - ref[:-6]).max() / np.abs(ref[:-6]).max():.3e}   stats diff {np.abs(a[-6:] - ref[-6:]).max():.3e}")
    except Exception as e:
        print("compare failed:", e)
