"""Diagnostic (GPU box): per-phase cycle shares (wave 0 of every workgroup) of mlp_update16_kernel from a -DMLP_STAMPS build.
Builds gpurun_out/libmappo_hip_stamps.so, loads it INSTEAD of the product library, runs actor/critic updates."""
import ctypes, os, subprocess, sys
import torch
sys.path.insert(0, '.')
ROOT = os.getcwd()
out = os.path.join(ROOT, 'gpurun_out', 'libmappo_hip_stamps.so')
from mappo_amd import build as _build
objdir = os.path.join(ROOT, 'gpurun_out', 'stamps_obj'); os.makedirs(objdir, exist_ok=True)
_build.build(force=True, verbose=False, extra_flags=['-DMLP_STAMPS', '-w'] + os.environ.get('STAMP_FLAGS', '').split(), lib=out, objdir=objdir)
from mappo_amd import _lib
_lib.LIB_PATH = out
_lib.SIGNATURES['mappo_debug_set_stamps'] = (ctypes.c_int, [ctypes.c_void_p])
from mappo_amd import ops
lib = _lib.load()
NAMES = ['staging + fold', 'xhat0', 'L1 MFMA', 'prefetch + act/LN1', 'L2 MFMA', 'act/LN2', 'head + loss + head products', 'LN2 bwd + write + db2',
         'dW2', 'd xhat1', 'LN1 bwd + write + db1', 'dW1', 'loop exit', 'loss statistics', 'vector sums (reduction tail)', 'raw->grad transform', 'slab write', 'first barrier', 'accumulator chunks']
class A_: pass
a = A_(); a.clip_param=0.2; a.entropy_coef=0.01; a.value_loss_coef=1.0; a.huber_delta=10.0; a.use_huber_loss=True; a.use_clipped_value_loss=True; a.use_policy_active_masks=True; a.use_value_active_masks=True; a.use_valuenorm=True
cfg = ops.ppo_cfg(a)
B = 76800
for name, D, A in (('actor', 18, 5), ('critic', 54, 1)):
    desc = ops.net_desc(D, A); P = ops.net_param_count(desc)
    params = torch.randn(P, device='cuda') * 0.1
    x = torch.randn(B, D, device='cuda'); ns = ops.mlp_backward_slabs(B)
    slabs = torch.zeros(ns, P, device='cuda'); part = ops.update_partials('cuda')
    ret = torch.randn(B, device='cuda'); active = torch.ones(B, device='cuda')
    mom = torch.zeros(4, dtype=torch.float64, device='cuda'); ops.minibatch_moments(ret, active, None, B, mom)
    stamps = torch.zeros(ns * 24, dtype=torch.int64, device='cuda')
    assert lib.mappo_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
    def run():
        if name == 'actor':
            ops.actor_update(params, desc, x, None, B, torch.ones(B, A, device='cuda'), torch.randint(0, A, (B,), device='cuda').float(),
                             -torch.rand(B, device='cuda') - 1, torch.randn(B, device='cuda'), active, mom, cfg, slabs, P, 0, part)
        else:
            ops.critic_update(params, desc, x, None, B, torch.randn(B, device='cuda'), ret, active, torch.tensor([0., 1., 1.], device='cuda'),
                              mom, cfg, slabs, P, 0, part)
    for _ in range(3): run()
    torch.cuda.synchronize()
    st = stamps.view(ns, 24).double().cpu()
    mean = st.mean(0); tot = mean.sum()
    print(f"--- {name} D={D} A={A} B={B}: {ns} blocks, mean cycles per block (wave 0) = {tot:.0f}")
    for i, n in enumerate(NAMES):
        if mean[i] > 0: print(f"  {n:28s} {mean[i]:9.0f} cyc  {100*mean[i]/tot:5.1f} %")
