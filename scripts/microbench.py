"""Microbenchmarks of single kernels (run on the GPU box): time vs batch size -> fixed cost + per-tile cost."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from mappo_amd import ops, flat

def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n

def net(D, A):
    desc = ops.net_desc(D, A)
    P = ops.net_param_count(desc)
    params = (torch.randn(P, device='cuda') * 0.1)
    return desc, params, P

which = sys.argv[1] if len(sys.argv) > 1 else 'all'
for (D, A, name) in ((18, 5, 'actor'), (54, 1, 'critic')):
    desc, params, P = net(D, A)
    for B in (32 * 256, 32 * 1024, 2 * 32 * 1024, 76800, 3 * 32 * 1024, 6 * 32 * 1024):
        x = torch.randn(B, D, device='cuda')
        out = torch.zeros(B, A, device='cuda')
        dout = torch.randn(B, A, device='cuda') / B
        ns = ops.mlp_backward_slabs(B)
        slabs = torch.zeros(ns, P, device='cuda')
        grad = torch.zeros(P, device='cuda')
        tf = timeit(lambda: ops.mlp_forward(params, desc, x, None, B, out))
        tb = timeit(lambda: ops.mlp_backward(params, desc, x, None, B, dout, slabs, P, 0))
        tr = timeit(lambda: ops.slab_reduce(slabs, ns, P, P, grad))
        print(f"{name} D={D} A={A} B={B:7d} tiles={B//32:5d}  fwd {tf:7.1f} us  bwd {tb:7.1f} us  slab_reduce {tr:6.1f} us", flush=True)
# loss kernel
for B, A in ((76800, 5), (76800 * 8, 5), (6553600, 5)):
    logits = torch.randn(B, A, device='cuda'); values = torch.randn(B, device='cuda')
    avail = torch.ones(B, A, device='cuda'); actions = torch.randint(0, A, (B,), device='cuda').float()
    oldlp = -torch.rand(B, device='cuda') - 0.5; adv = torch.randn(B, device='cuda'); active = torch.ones(B, device='cuda')
    vold = torch.randn(B, device='cuda'); ret = torch.randn(B, device='cuda')
    vn = torch.tensor([0.0, 1.0, 1.0], device='cuda'); mom = torch.zeros(4, dtype=torch.float64, device='cuda')
    ops.minibatch_moments(ret, active, None, B, mom)
    dl = torch.zeros(B, A, device='cuda'); dv = torch.zeros(B, device='cuda'); st = torch.zeros(6, dtype=torch.float64, device='cuda')
    class Aa: pass
    a = Aa(); a.clip_param=0.2; a.entropy_coef=0.01; a.value_loss_coef=1.0; a.huber_delta=10.0; a.use_huber_loss=True; a.use_clipped_value_loss=True; a.use_policy_active_masks=True; a.use_value_active_masks=True; a.use_valuenorm=True
    cfg = ops.ppo_cfg(a)
    t = timeit(lambda: ops.ppo_loss_fwd_bwd(logits, values, None, avail, actions, oldlp, adv, active, vold, ret, vn, mom, dl, dv, st, cfg))
    print(f"ppo_loss B={B} A={A}: {t:.1f} us (2 launches)  -> {B*4*(3*A+8)/t/1e3:.0f} GB/s algorithmic", flush=True)
# gae
for T, R in ((25, 3072), (400, 16384)):
    rw = torch.randn(T, R, device='cuda'); vp = torch.randn(T + 1, R, device='cuda'); mk = torch.ones(T + 1, R, device='cuda')
    ret = torch.zeros(T + 1, R, device='cuda'); nv = torch.randn(R, device='cuda'); vn = torch.tensor([0.0, 1.0, 1.0], device='cuda')
    t = timeit(lambda: ops.gae_scan(rw, vp, nv, mk, None, ret, vn, 0.99, 0.95, True, False))
    print(f"gae T={T} R={R}: {t:.1f} us -> {T*R*16/t/1e3:.0f} GB/s algorithmic", flush=True)
