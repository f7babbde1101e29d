"""Experiment (GPU box): the update kernel with 4 vs 8 waves per workgroup (1 vs 2 waves per SIMD) on a network small
enough for 8 waves' tiles to fit the LDS (layer_N = 0).  STAMP_FLAGS=-DEXP_WAVES8 selects the 8-wave build."""
import ctypes, os, sys
import torch
sys.path.insert(0, '.')
ROOT = os.getcwd()
tag = 'w8' if 'EXP_WAVES8' in os.environ.get('STAMP_FLAGS', '') else 'w4'
out = os.path.join(ROOT, 'gpurun_out', f'libmappo_hip_{tag}.so')
from mappo_amd import build as _build
objdir = os.path.join(ROOT, 'gpurun_out', f'obj_{tag}'); os.makedirs(objdir, exist_ok=True)
_build.build(force=True, verbose=False, extra_flags=['-w'] + os.environ.get('STAMP_FLAGS', '').split(), lib=out, objdir=objdir)
from mappo_amd import _lib
_lib.LIB_PATH = out
from mappo_amd import ops
class A_: pass
a = A_(); a.clip_param=0.2; a.entropy_coef=0.01; a.value_loss_coef=1.0; a.huber_delta=10.0; a.use_huber_loss=True; a.use_clipped_value_loss=True; a.use_policy_active_masks=True; a.use_value_active_masks=True; a.use_valuenorm=True
cfg = ops.ppo_cfg(a)
for B in (76800, 76800 * 4):
  for name, D, A, LN in (('actor', 18, 5, 0), ('critic', 54, 1, 0)):
    desc = ops.net_desc(D, A, layer_N=LN); P = ops.net_param_count(desc)
    params = torch.randn(P, device='cuda') * 0.1
    x = torch.randn(B, D, device='cuda'); ns = ops.mlp_backward_slabs(B)
    slabs = torch.zeros(ns, P, device='cuda'); part = ops.update_partials('cuda')
    ret = torch.randn(B, device='cuda'); active = torch.ones(B, device='cuda')
    mom = torch.zeros(4, dtype=torch.float64, device='cuda'); ops.minibatch_moments(ret, active, None, B, mom)
    av = torch.ones(B, A, device='cuda'); act = torch.randint(0, A, (B,), device='cuda').float(); olp = -torch.rand(B, device='cuda') - 1
    adv = torch.randn(B, device='cuda'); vold = torch.randn(B, device='cuda'); vn = torch.tensor([0., 1., 1.], device='cuda')
    def run():
        if name == 'actor':
            ops.actor_update(params, desc, x, None, B, av, act, olp, adv, active, mom, cfg, slabs, P, 0, part)
        else:
            ops.critic_update(params, desc, x, None, B, vold, ret, active, vn, mom, cfg, slabs, P, 0, part)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    print(f"{tag} {name} D={D} A={A} LN={LN} B={B}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us  |grad|={slabs.sum(0).norm().item():.6f}")
