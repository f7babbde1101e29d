import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import mappo_oracle as O
from mappo_amd import ops, flat
import test_gpu_kernels as TK
def run(D, A, relu, LN, fn, B):
    torch.manual_seed(D * 100 + A + 1)
    a = O.default_args(use_ReLU=relu, layer_N=LN, use_feature_normalization=fn)
    net = O.ActorRef(a, D, A) if A > 1 else O.CriticRef(a, D)
    TK._randomize(net, 9)
    head = "act.action_out.linear" if A > 1 else "v_out"
    desc = ops.net_desc(D, A, LN, relu, fn)
    params, layout, P = TK._flat_from_module(ops, net, desc, head)
    rng = np.random.default_rng(B + 1)
    x = rng.standard_normal((B, D)).astype(np.float32) * 2
    dout = (rng.standard_normal((B, A)) / np.sqrt(B)).astype(np.float32)
    xt = torch.from_numpy(x)
    if A > 1:
        feats, _ = net.features(xt, None, None); out = net.act.action_out.linear(feats)
    else:
        out = net(xt, None, None)[0]
    (out * torch.from_numpy(dout)).sum().backward()
    ref = {k: p_.grad.numpy() for k, p_ in net.named_parameters() if p_.grad is not None}
    n_slabs = ops.mlp_backward_slabs(B)
    slabs = torch.zeros(n_slabs, P, device="cuda")
    ops.mlp_backward(params, desc, TK.dev(x), None, B, TK.dev(dout), slabs, P, 0)
    grad = torch.zeros(P, device="cuda")
    ops.slab_reduce(slabs, n_slabs, P, P, grad)
    g = grad.cpu().numpy()
    print(f"--- D={D} A={A} relu={relu} LN={LN} fn={fn} B={B}")
    for key, off, shape in layout:
        got = g[off: off + int(np.prod(shape))].reshape(shape)
        r = ref[key]
        err = np.abs(got - r).max() / max(np.abs(r).max(), 1e-12)
        ratio = (got.reshape(-1) @ r.reshape(-1)) / max((r.reshape(-1) @ r.reshape(-1)), 1e-30)
        print(f"{key:32s} err={err:.3e} proj={ratio:.4f}")
run(18, 5, True, 1, True, 32)
run(18, 5, True, 1, True, 3072)
run(54, 1, True, 1, True, 64)
