"""Diagnostic: wide-input fused update vs unfused vs float64 autograd, per-parameter error (scripts only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import mappo_oracle as O
from mappo_amd import ops
import test_gpu_kernels as TK

def run(D, S, A, B, verbose=False):
    relu = True
    torch.manual_seed(B + D)
    rng = np.random.default_rng(B + A)
    f = np.float32
    a = O.default_args(use_ReLU=relu)
    actor, critic = O.ActorRef(a, D, A), O.CriticRef(a, S)
    TK._randomize(actor, 3); TK._randomize(critic, 4)
    da, dc = ops.net_desc(D, A, 1, relu, True), ops.net_desc(S, 1, 1, relu, True)
    pa, la, Pa = TK._flat_from_module(ops, actor, da, "act.action_out.linear")
    pc, lc, Pc = TK._flat_from_module(ops, critic, dc, "v_out")
    n_rows = B
    obs = rng.standard_normal((n_rows, D)).astype(f)
    sobs = rng.standard_normal((n_rows, S)).astype(f)
    avail = (rng.random((n_rows, A)) > 0.3).astype(f)
    actions = rng.integers(0, A, n_rows).astype(f)
    avail[np.arange(n_rows), actions.astype(int)] = 1.0
    old_logp = (-np.abs(rng.standard_normal(n_rows)) * 0.3 - np.log(A)).astype(f)
    adv = rng.standard_normal(n_rows).astype(f)
    active = (rng.random(n_rows) > 0.25).astype(f)
    if os.environ.get("DBG_SAFE_ROWS", "0") == "1":
        unsafe = np.minimum(TK._relu_margin(actor, torch.from_numpy(obs)), TK._relu_margin(critic, torch.from_numpy(sobs))) < 1e-4
        active[unsafe] = 0.0
        print("unsafe rows:", unsafe.mean())
    ret = (rng.standard_normal(n_rows) * 3).astype(f)
    with torch.no_grad():
        v_now = critic(torch.from_numpy(sobs), None, None)[0].numpy().reshape(-1)
    v_old = (v_now + rng.standard_normal(n_rows) * 0.25).astype(f)
    vn = O.ValueNormRef(); vn.update(ret[:50].reshape(-1, 1)); vn.update(ret.reshape(-1, 1))
    dev = TK.dev
    g = dict(obs=dev(obs), sobs=dev(sobs), avail=dev(avail), actions=dev(actions), old=dev(old_logp), adv=dev(adv),
             active=dev(active), ret=dev(ret), vold=dev(v_old), vn=dev(vn.state()))
    mom = torch.zeros(4, dtype=torch.float64, device="cuda")
    ops.minibatch_moments(g["ret"], g["active"], None, B, mom)
    cfg = ops.ppo_cfg(a)
    ns = ops.mlp_backward_slabs(B)
    col_c = ((Pa + 255) // 256) * 256
    P = col_c + ((Pc + 255) // 256) * 256
    slabs = torch.zeros(ns, P, device="cuda")
    part_a, part_c = ops.update_partials("cuda"), ops.update_partials("cuda")
    ops.actor_update(pa, da, g["obs"], None, B, g["avail"], g["actions"], g["old"], g["adv"], g["active"], mom, cfg, slabs, P, 0, part_a)
    ops.critic_update(pc, dc, g["sobs"], None, B, g["vold"], g["ret"], g["active"], g["vn"], mom, cfg, slabs, P, col_c, part_c)
    grad_f = slabs.double().sum(0).cpu().numpy()
    logits, values = torch.zeros(B, A, device="cuda"), torch.zeros(B, device="cuda")
    ops.mlp_forward(pa, da, g["obs"], None, B, logits)
    ops.mlp_forward(pc, dc, g["sobs"], None, B, values)
    dl, dv = torch.zeros(B, A, device="cuda"), torch.zeros(B, device="cuda")
    stats_u = torch.zeros(6, dtype=torch.float64, device="cuda")
    ops.ppo_loss_fwd_bwd(logits, values, None, g["avail"], g["actions"], g["old"], g["adv"], g["active"], g["vold"], g["ret"], g["vn"], mom, dl, dv, stats_u, cfg)
    slabs_u = torch.zeros(ns, P, device="cuda")
    ops.mlp_backward(pa, da, g["obs"], None, B, dl, slabs_u, P, 0)
    ops.mlp_backward(pc, dc, g["sobs"], None, B, dv.view(B, 1), slabs_u, P, col_c)
    grad_u = slabs_u.double().sum(0).cpu().numpy()
    # float64 autograd
    actor.double(); critic.double()
    t = lambda x: torch.from_numpy(x).double()
    lp, ent, _ = actor.evaluate_actions(t(obs), None, t(actions).view(-1, 1), None, t(avail), t(active).view(-1, 1))
    vals = critic(t(sobs), None, None)[0]
    act_t, adv_t, old_t = t(active).view(-1, 1), t(adv).view(-1, 1), t(old_logp).view(-1, 1)
    imp = torch.exp(lp - old_t)
    surr = torch.min(imp * adv_t, torch.clamp(imp, 1 - a.clip_param, 1 + a.clip_param) * adv_t)
    pl = (-surr * act_t).sum() / act_t.sum()
    (pl - a.entropy_coef * ent).backward()
    m, sd = vn.state()[0], None
    vns = vn.state().astype(np.float64)
    mean = vns[0] / vns[2]; var = max(vns[1] / vns[2] - mean * mean, 1e-2)
    tgt = (t(ret).view(-1, 1) - mean) / np.sqrt(var)
    vo = t(v_old).view(-1, 1)
    vclip = vo + (vals - vo).clamp(-a.clip_param, a.clip_param)
    l = torch.max(O.huber_ref(tgt - vals, a.huber_delta), O.huber_ref(tgt - vclip, a.huber_delta))
    vl = (l * act_t).sum() / act_t.sum()
    (vl * a.value_loss_coef).backward()
    worst = dict(actor=(0, 0), critic=(0, 0))
    for name, net, lay, c0 in (("actor", actor, la, 0), ("critic", critic, lc, col_c)):
        for key, off, shape in lay:
            n = int(np.prod(shape))
            ref = dict(net.named_parameters())[key].grad.numpy().reshape(-1)
            sc = max(np.abs(ref).max(), 1e-30)
            ef = np.abs(grad_f[c0 + off:c0 + off + n] - ref).max() / sc
            eu = np.abs(grad_u[c0 + off:c0 + off + n] - ref).max() / sc
            worst[name] = (max(worst[name][0], ef), max(worst[name][1], eu))
            if verbose: print(f"{name:6s} {key:32s} max|ref|={sc:.3e}  fused err={ef:.2e}  unfused err={eu:.2e}")
    print(f"D={D} S={S} A={A} B={B}: actor fused {worst['actor'][0]:.1e} unfused {worst['actor'][1]:.1e} | critic fused {worst['critic'][0]:.1e} unfused {worst['critic'][1]:.1e}", flush=True)



if __name__ == "__main__":
    D, S, A = [int(v) for v in sys.argv[1:4]]
    for B in [int(v) for v in sys.argv[4:]]:
        run(D, S, A, B)
