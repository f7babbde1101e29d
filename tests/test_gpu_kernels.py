"""GPU parity tests: every HIP kernel, called through the C ABI (mappo_amd.ops -> libmappo_hip.so), against
the CPU oracle (oracle/mappo_oracle.py) and, where they exist, the committed golden vectors of the reference.

Tolerances: bit-exact for indices / actions / masks; 1e-5 relative (north_star) for fp32 returns, losses and
forward outputs, with an absolute floor of 1e-6 where values cross zero.  Gradients of a 76 800-term fp32 sum
are compared at 1e-4 relative to the tensor's max magnitude (re-association of the sum; the oracle itself
differs from the reference by 5e-7 there, see tests/test_oracle_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import golden, sub
from oracle import mappo_oracle as O

pytestmark = pytest.mark.gpu


def dev(x, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).cuda()


def close(a, b, rtol=1e-5, atol=1e-6, msg=""):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a.astype(np.float64), b.astype(np.float64), rtol=rtol, atol=atol, err_msg=msg)


def close_rel_max(a, b, tol=1e-4, msg=""):
    a = a.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-12)
    err = np.abs(a - b).max() / scale
    assert err <= tol, f"{msg}: max err / max|ref| = {err:.3e} > {tol}"


@pytest.fixture(scope="module")
def ops(gpu_device):
    from mappo_amd import ops as _ops
    return _ops


# ------------------------------------------------------------------------------------------------------
def test_mfma_lane_maps(ops):
    """v_mfma_f32_32x32x2_f32 operand / accumulator lane maps, asymmetric data (cdna guide §3)."""
    rng = np.random.default_rng(0)
    A = rng.integers(-8, 9, (32, 2)).astype(np.float32)
    B = rng.integers(-8, 9, (2, 32)).astype(np.float32)
    D = torch.zeros(32, 32, device="cuda")
    ops.selftest_mfma(dev(A), dev(B), D)
    np.testing.assert_array_equal(D.cpu().numpy(), A @ B)


# ------------------------------------------------------------------------------------------------------
def _run_gae(ops, rewards, value_preds, masks, bad_masks, next_value, vn_state, gamma, lam, use_gae, ptl):
    T = rewards.shape[0]
    R = rewards[0].size
    rw, vp, mk, bm = dev(rewards.reshape(T, R)), dev(value_preds.reshape(T + 1, R)), dev(masks.reshape(T + 1, R)), dev(bad_masks.reshape(T + 1, R))
    ret = torch.zeros(T + 1, R, device="cuda")
    nv = dev(next_value.reshape(R))
    vs = dev(vn_state) if vn_state is not None else None
    ops.gae_scan(rw, vp, nv, mk, bm, ret, vs, gamma, lam, use_gae, ptl)
    return ret.cpu().numpy(), vp.cpu().numpy()


def test_gae_golden_all_branches(ops):
    g = golden("gae")
    for c in range(int(g["n_cases"])):
        d = sub(g, f"c{c}")
        use_gae, ptl, use_vn = [bool(x) for x in d["flags"]]
        ret, vp = _run_gae(ops, d["rewards"], d["value_preds"], d["masks"], d["bad_masks"], d["next_value"],
                           d["vn_state"] if use_vn else None, float(d["hyper"][0]), float(d["hyper"][1]), use_gae, ptl)
        close(ret.reshape(d["returns"].shape), d["returns"], 1e-5, 2e-6, f"gae case {c}")
        if use_gae:
            np.testing.assert_array_equal(vp.reshape(d["value_preds_after"].shape), d["value_preds_after"])


@pytest.mark.parametrize("T,N,M", [(25, 1024, 3), (400, 64, 3), (1100, 5, 3), (1, 3, 2), (33, 7, 1)])
def test_gae_vs_oracle_sizes(ops, T, N, M):
    """BASELINE config-2 size, SMAC-length episodes, the multi-chunk path (T > 512), ragged R."""
    rng = np.random.default_rng(T * 1000 + N)
    f = np.float32
    rewards = rng.standard_normal((T, N, M, 1)).astype(f)
    vp = rng.standard_normal((T + 1, N, M, 1)).astype(f)
    masks = (rng.random((T + 1, N, M, 1)) > 0.1).astype(f)
    bad = (rng.random((T + 1, N, M, 1)) > 0.1).astype(f)
    nv = rng.standard_normal((N, M, 1)).astype(f)
    vn = O.ValueNormRef(); vn.update(rng.standard_normal((64, 1)).astype(f) * 2 + 1)
    for use_gae in (True, False):
        for ptl in (False, True):
            ref = O.compute_returns_ref(rewards, vp.copy(), masks, bad, nv, 0.99, 0.95, use_gae, ptl, vn.denormalize)
            ret, _ = _run_gae(ops, rewards, vp, masks, bad, nv, vn.state(), 0.99, 0.95, use_gae, ptl)
            close(ret.reshape(ref.shape), ref, 1e-5, 1e-5 if T > 500 else 3e-6, f"T={T} gae={use_gae} ptl={ptl}")


def test_gae_property_linearity(ops):
    """Size-independent property at full size: without value bootstrap, returns are linear in the rewards."""
    T, R = 400, 16384
    g = torch.Generator(device="cuda").manual_seed(3)
    r1 = torch.randn(T, R, device="cuda", generator=g)
    r2 = torch.randn(T, R, device="cuda", generator=g)
    masks = (torch.rand(T + 1, R, device="cuda", generator=g) > 0.05).float()
    zeros = torch.zeros(T + 1, R, device="cuda")
    nv = torch.zeros(R, device="cuda")
    outs = []
    for rw in (r1, r2, r1 + 2 * r2):
        ret = torch.zeros(T + 1, R, device="cuda")
        ops.gae_scan(rw.contiguous(), zeros.clone(), nv, masks, None, ret, None, 0.99, 0.95, True, False)
        outs.append(ret)
    close(outs[2], outs[0] + 2 * outs[1], 1e-4, 1e-4)


# ------------------------------------------------------------------------------------------------------
def test_advantage_golden_and_oracle(ops):
    g = golden("advnorm")
    for c in range(int(g["n_cases"])):
        d = sub(g, f"c{c}")
        use_vn = bool(d["use_vn"])
        n = d["adv"].size
        adv = torch.zeros(n, device="cuda")
        mom = torch.zeros(3, dtype=torch.float64, device="cuda")
        ops.adv_moments(dev(d["returns"][:-1].reshape(-1)), dev(d["value_preds"][:-1].reshape(-1)),
                        dev(d["active_masks"][:-1].reshape(-1)), dev(d["vn_state"]) if use_vn else None, adv, mom)
        ops.adv_normalize(adv, mom)
        close(adv.cpu().numpy().reshape(d["adv"].shape), d["adv"], 1e-5, 2e-6)
        m = mom.cpu().numpy()
        close(m[0] / m[2], d["mean"], 1e-5, 1e-6)
    # BASELINE config-2 size against the oracle
    rng = np.random.default_rng(5)
    T, R = 25, 3072
    ret = rng.standard_normal((T + 1, R, 1)).astype(np.float32) * 3
    vp = rng.standard_normal((T + 1, R, 1)).astype(np.float32)
    act = (rng.random((T + 1, R, 1)) > 0.2).astype(np.float32)
    vn = O.ValueNormRef(); vn.update(ret[:5].reshape(-1, 1))
    ref, mean, std = O.normalized_advantages_ref(ret, vp, act, vn.denormalize)
    adv = torch.zeros(T * R, device="cuda")
    mom = torch.zeros(3, dtype=torch.float64, device="cuda")
    ops.adv_moments(dev(ret[:-1].reshape(-1)), dev(vp[:-1].reshape(-1)), dev(act[:-1].reshape(-1)), dev(vn.state()), adv, mom)
    ops.adv_normalize(adv, mom)
    close(adv.cpu().numpy(), ref.reshape(-1), 1e-5, 3e-6)


def test_valuenorm_update(ops):
    g = golden("valuenorm")
    st = torch.zeros(3, device="cuda")
    for i in range(4):
        x = g[f"x{i}"].reshape(-1)
        mom = torch.zeros(4, dtype=torch.float64, device="cuda")
        ops.minibatch_moments(dev(x), torch.ones(x.size, device="cuda"), None, x.size, mom)
        ops.valuenorm_update(st, mom)
        close(st, g[f"state{i}"], 2e-6, 1e-9)
    # gathered rows
    rng = np.random.default_rng(1)
    ret = rng.standard_normal(5000).astype(np.float32)
    act = (rng.random(5000) > 0.3).astype(np.float32)
    rows = rng.permutation(5000)[:3000].astype(np.int32)
    mom = torch.zeros(4, dtype=torch.float64, device="cuda")
    ops.minibatch_moments(dev(ret), dev(act), dev(rows, torch.int32), 3000, mom)
    m = mom.cpu().numpy()
    close(m, [ret[rows].astype(np.float64).sum(), (ret[rows].astype(np.float64) ** 2).sum(), act[rows].sum(), 3000], 1e-12, 1e-9)


# ------------------------------------------------------------------------------------------------------
def _loss_case(ops, B, A, n_rows, seed, with_rows, with_avail, **flags):
    rng = np.random.default_rng(seed)
    f = np.float32
    a = O.default_args(**flags)
    logits = (rng.standard_normal((B, A)) * 2).astype(f)
    values = rng.standard_normal(B).astype(f)
    rows = rng.permutation(n_rows)[:B].astype(np.int32) if with_rows else np.arange(B, dtype=np.int32)
    avail = (rng.random((n_rows, A)) > 0.3).astype(f)
    actions = rng.integers(0, A, n_rows).astype(f)
    avail[np.arange(n_rows), actions.astype(int)] = 1.0
    old_logp = (-np.abs(rng.standard_normal(n_rows)) - 0.3).astype(f)
    adv = rng.standard_normal(n_rows).astype(f)
    active = (rng.random(n_rows) > 0.25).astype(f)
    v_old = (values.mean() + rng.standard_normal(n_rows) * 0.3).astype(f)
    v_old[rows] = values + rng.standard_normal(B).astype(f) * 0.25       # both sides of the clip range
    ret = (rng.standard_normal(n_rows) * 4).astype(f)
    ret[rng.random(n_rows) > 0.9] *= 20                                    # Huber's linear branch
    vn = O.ValueNormRef(); vn.update(ret[:50].reshape(-1, 1))
    vn.update(ret[rows].reshape(-1, 1))
    mean, var = vn.mean_var()
    ref = O.ppo_loss_fwd_bwd_ref(logits, avail[rows] if with_avail else None, actions[rows], old_logp[rows], adv[rows],
                                 active[rows], values, v_old[rows], ret[rows], float(mean), float(var), a.clip_param,
                                 a.entropy_coef, a.value_loss_coef, a.huber_delta, a.use_huber_loss,
                                 a.use_clipped_value_loss, a.use_policy_active_masks, a.use_value_active_masks,
                                 a.use_valuenorm)
    mom = torch.zeros(4, dtype=torch.float64, device="cuda")
    d_rows = dev(rows, torch.int32) if with_rows else None
    ops.minibatch_moments(dev(ret), dev(active), d_rows, B, mom)
    dl, dv = torch.zeros(B, A, device="cuda"), torch.zeros(B, device="cuda")
    stats = torch.zeros(6, dtype=torch.float64, device="cuda")
    ops.ppo_loss_fwd_bwd(dev(logits), dev(values), d_rows, dev(avail) if with_avail else None, dev(actions), dev(old_logp),
                         dev(adv), dev(active), dev(v_old), dev(ret), dev(vn.state()), mom, dl, dv, stats, ops.ppo_cfg(a))
    s = stats.cpu().numpy()
    close(s[0], ref["value_loss"], 1e-5, 1e-7, "value_loss")
    close(s[1], ref["policy_loss"], 1e-5, 1e-7, "policy_loss")
    close(s[2], ref["dist_entropy"], 1e-5, 1e-7, "entropy")
    close(s[3], ref["ratio_mean"], 1e-5, 1e-7, "ratio")
    assert s[4] == active[rows].sum() and s[5] == B
    close_rel_max(dl, ref["dlogits"], 2e-5, "dlogits")
    close_rel_max(dv, ref["dvalues"], 2e-5, "dvalues")


@pytest.mark.parametrize("B,A", [(600, 5), (76800, 5), (1000, 9), (777, 18), (256, 1), (300, 32)])
def test_ppo_loss_sizes(ops, B, A):
    _loss_case(ops, B, A, B + 123, B + A, with_rows=True, with_avail=True)
    _loss_case(ops, B, A, B, B + A + 1, with_rows=False, with_avail=False)


@pytest.mark.parametrize("flags", [dict(use_huber_loss=False), dict(use_clipped_value_loss=False),
                                   dict(use_policy_active_masks=False, use_value_active_masks=False),
                                   dict(use_valuenorm=False), dict(clip_param=0.05, entropy_coef=0.1, value_loss_coef=0.5,
                                                                   huber_delta=1.0)])
def test_ppo_loss_flag_variants(ops, flags):
    _loss_case(ops, 2000, 5, 2500, 11, with_rows=True, with_avail=True, **flags)


def test_ppo_loss_golden_sample(ops):
    """The reference's own ppo_update sample (golden c0): losses as returned by R_MAPPO.ppo_update."""
    g = golden("ppo_update")
    d = sub(g, "c0")
    T, N, M, D, S, A, H = [int(x) for x in d["dims"]]
    a = O.default_args(hidden_size=H, lr=7e-4, critic_lr=7e-4)
    pol = O.PolicyRef(a, D, S, A)
    pol.actor.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sub(g, "c0/actor0").items()})
    pol.critic.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sub(g, "c0/critic0").items()})
    sm = {k: d[f"sample/{k}"] for k in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds",
                                        "returns", "masks", "active_masks", "old_action_log_probs", "adv_targ", "available_actions")}
    t = torch.from_numpy
    with torch.no_grad():
        feats, _ = pol.actor.features(t(sm["obs"]), t(sm["rnn_states"]), t(sm["masks"]))
        logits = pol.actor.act.action_out.linear(feats).numpy()
        values = pol.critic(t(sm["share_obs"]), t(sm["rnn_states_critic"]), t(sm["masks"]))[0].numpy().reshape(-1)
    B = logits.shape[0]
    vn = torch.tensor(d["vn0"]).cuda()
    mom = torch.zeros(4, dtype=torch.float64, device="cuda")
    ret, act = dev(sm["returns"].reshape(-1)), dev(sm["active_masks"].reshape(-1))
    ops.minibatch_moments(ret, act, None, B, mom)
    ops.valuenorm_update(vn, mom)
    close(vn, d["r0/vn"], 2e-6, 1e-9)
    dl, dv = torch.zeros(B, A, device="cuda"), torch.zeros(B, device="cuda")
    stats = torch.zeros(6, dtype=torch.float64, device="cuda")
    ops.ppo_loss_fwd_bwd(dev(logits), dev(values), None, dev(sm["available_actions"]), dev(sm["actions"].reshape(-1)),
                         dev(sm["old_action_log_probs"].reshape(-1)), dev(sm["adv_targ"].reshape(-1)), act,
                         dev(sm["value_preds"].reshape(-1)), ret, vn, mom, dl, dv, stats, ops.ppo_cfg(a))
    s = stats.cpu().numpy()
    ref = d["r0/stats"]          # value_loss, critic_grad_norm, policy_loss, dist_entropy, actor_grad_norm, ratio
    close([s[0], s[1], s[2], s[3]], [ref[0], ref[2], ref[3], ref[5]], 1e-5, 1e-7)


# ------------------------------------------------------------------------------------------------------
def _flat_from_module(ops, module, desc, head_prefix):
    from mappo_amd import flat
    layout, P = flat.net_layout(desc, head_prefix)
    assert P == ops.net_param_count(desc)
    buf = torch.zeros(P)
    flat.pack_state_dict(buf, layout, {k: v.detach() for k, v in module.state_dict().items()})
    return buf.cuda(), layout, P


def _randomize(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n_, p_ in module.named_parameters():
            if "norm" in n_ or "bias" in n_ or ".2." in n_:
                p_.add_(0.2 * torch.randn(p_.shape, generator=g))
            if "action_out" in n_ and "weight" in n_:
                p_.mul_(40.0)


NET_CASES = [  # D, A, relu, layer_N, feature_norm, B
    (18, 5, True, 1, True, 3072), (54, 1, True, 1, True, 3072), (18, 5, False, 1, True, 700), (54, 1, False, 1, True, 100),
    (30, 9, True, 1, True, 333), (48, 1, True, 1, True, 64), (64, 18, True, 1, False, 257), (7, 3, True, 0, True, 90),
    (33, 32, False, 2, True, 130), (18, 5, True, 1, True, 1),
    # wide observations (K-chunked layer 1): SMAC MMM2 shapes (BASELINE configs[3]) and the 512-wide stress config
    (176, 18, True, 1, True, 300), (322, 1, True, 1, True, 257), (512, 5, False, 1, True, 100), (70, 3, True, 1, False, 64),
    (130, 1, True, 0, True, 33),
    # more than 256 tiles: the streamed wide forward (below that the split-K one-tile-per-workgroup kernel runs)
    (322, 1, True, 1, True, 4200), (512, 5, True, 1, True, 4133)]


@pytest.mark.parametrize("D,A,relu,LN,fn,B", NET_CASES)
def test_mlp_forward_vs_oracle(ops, D, A, relu, LN, fn, B):
    torch.manual_seed(D * 100 + A)
    a = O.default_args(use_ReLU=relu, layer_N=LN, use_feature_normalization=fn)
    net = O.ActorRef(a, D, A) if A > 1 else O.CriticRef(a, D)
    _randomize(net, 7)
    desc = ops.net_desc(D, A, LN, relu, fn)
    params, _, _ = _flat_from_module(ops, net, desc, "act.action_out.linear" if A > 1 else "v_out")
    rng = np.random.default_rng(B)
    n_rows = B + 50
    x = rng.standard_normal((n_rows, D)).astype(np.float32) * 2
    rows = rng.permutation(n_rows)[:B].astype(np.int32)
    with torch.no_grad():
        xt = torch.from_numpy(x[rows])
        if A > 1:
            feats, _ = net.features(xt, None, None)
            ref = net.act.action_out.linear(feats).numpy()
        else:
            ref = net(xt, None, None)[0].numpy()
    out = torch.zeros(B, A, device="cuda")
    ops.mlp_forward(params, desc, dev(x), dev(rows, torch.int32), B, out)
    close(out, ref, 1e-5, 2e-5 if A > 1 else 2e-6, f"forward D={D} A={A}")
    # identity rows
    out2 = torch.zeros(B, A, device="cuda")
    ops.mlp_forward(params, desc, dev(x[rows]), None, B, out2)
    np.testing.assert_array_equal(out.cpu().numpy(), out2.cpu().numpy())


@pytest.mark.parametrize("D,LN,relu,fn", [(65, 1, True, True), (128, 1, True, True), (130, 0, True, True), (200, 1, False, True), (256, 1, True, False),
                                          (300, 1, True, True), (322, 1, True, True), (400, 1, True, True), (448, 0, False, True), (511, 1, True, True),
                                          (512, 1, True, True)])
def test_wide_trunk_features_resident_kernel_every_chunk_count(ops, monkeypatch, D, LN, relu, fn):
    """Trunk features of a training-sized batch of a wide-input recurrent network (mappo_mlp_features_seq -> wide_features16_resident_kernel,
    W1' resident, the chunk count ceil(D / 64) = 2..8 a template parameter, row ends inside the last chunk at D % 64 != 0 and D % 4 != 0)
    vs the reference trunk (mlp.py:18-55) on a strided subset of rows, and vs the streamed kernel (MAPPO_WIDE_RESIDENT=0) on all rows."""
    L, Nc = 8, 8192 + 16                                   # 65 664 rows = 4 104 tiles (>= 4 096: the resident kernel's threshold)
    B = L * Nc
    torch.manual_seed(D)
    a = O.default_args(use_ReLU=relu, layer_N=LN, use_feature_normalization=fn)
    net = O.CriticRef(a, D)
    _randomize(net, D + 3)
    desc = ops.net_desc(D, 1, LN, relu, fn)
    params, _, _ = _flat_from_module(ops, net, desc, "v_out")
    g = torch.Generator(device="cuda").manual_seed(D + 1)
    n_rows = B + 64
    x = torch.randn(n_rows, D, device="cuda", generator=g) * 1.5 + 0.25
    rows = torch.randperm(n_rows, device="cuda", generator=g)[:B].to(torch.int32)
    outs = []
    for res in ("1", "0"):
        monkeypatch.setenv("MAPPO_WIDE_RESIDENT", res)
        feat = torch.full((ops.gru16_blocked_floats(L, Nc),), float("nan"), device="cuda")
        ops.mlp_features_seq(params, desc, x, rows, L, Nc, feat)
        torch.cuda.synchronize()
        outs.append(feat.view(L * (Nc // 16), 4, 4, 16, 4))          # [tile][b][q][n][i]: feature 16 b + 4 q + i of sample 16 tile + n
    f1 = outs[0].permute(0, 3, 1, 2, 4).reshape(B, 64)               # -> [row][feature]
    f0 = outs[1].permute(0, 3, 1, 2, 4).reshape(B, 64)
    np.testing.assert_allclose(f1.cpu().numpy(), f0.cpu().numpy(), rtol=1e-5, atol=1e-5)      # (different summation orders through two LayerNorms)
    sel = torch.arange(0, B, 97, device="cuda")
    with torch.no_grad():
        ref = net.base(x[rows.long()[sel]].cpu()).numpy() if hasattr(net, "base") else None
    assert ref is not None
    close(f1[sel], ref, 1e-5, 2e-5, f"resident features D={D}")


def test_forward_golden_reference_outputs(ops):
    """get_actions(deterministic) / get_values / evaluate_actions of the reference itself (golden forward c0, c1)."""
    g = golden("forward")
    for c in (0, 1):
        d = sub(g, f"c{c}")
        relu, rec, D, S, A, B, H = [int(x) for x in d["spec"]]
        from mappo_amd import flat
        da, dc = ops.net_desc(D, A, 1, relu, True), ops.net_desc(S, 1, 1, relu, True)
        la, Pa = flat.net_layout(da, "act.action_out.linear")
        lc, Pc = flat.net_layout(dc, "v_out")
        pa, pc = torch.zeros(Pa), torch.zeros(Pc)
        flat.pack_state_dict(pa, la, sub(g, f"c{c}/actor")); flat.pack_state_dict(pc, lc, sub(g, f"c{c}/critic"))
        pa, pc = pa.cuda(), pc.cuda()
        for tag in ("avail", "noavail"):
            av = dev(d["avail"]) if tag == "avail" else None
            actions, logp = torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda")
            ops.actor_act(pa, da, dev(d["obs"]), av, B, True, 1, 0, actions, logp)
            np.testing.assert_array_equal(actions.cpu().numpy().astype(np.int64), d[f"{tag}/actions"].reshape(-1))
            close(logp, d[f"{tag}/logp"].reshape(-1), 1e-5, 1e-6)
            vals = torch.zeros(B, 1, device="cuda")
            ops.mlp_forward(pc, dc, dev(d["share_obs"]), None, B, vals)
            close(vals, d[f"{tag}/values"], 1e-5, 1e-6)


def test_actor_act_sampling_distribution(ops):
    """Sampling cannot match torch's CPU multinomial stream; check the distribution instead."""
    torch.manual_seed(3)
    D, A, B = 18, 5, 64
    a = O.default_args()
    net = O.ActorRef(a, D, A); _randomize(net, 5)
    desc = ops.net_desc(D, A)
    params, _, _ = _flat_from_module(ops, net, desc, "act.action_out.linear")
    x = np.tile(np.random.default_rng(0).standard_normal((1, D)).astype(np.float32), (B, 1))
    avail = np.ones((B, A), np.float32); avail[:, 2] = 0
    with torch.no_grad():
        feats, _ = net.features(torch.from_numpy(x[:1]), None, None)
        z = net.act.logits(feats, torch.from_numpy(avail[:1]))
        p = torch.softmax(z, -1).numpy()[0]
    counts = np.zeros(A)
    actions, logp = torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda")
    n_rounds = 400
    for it in range(n_rounds):
        ops.actor_act(params, desc, dev(x), dev(avail), B, False, 1234, it, actions, logp)
        acts = actions.cpu().numpy().astype(int)
        counts += np.bincount(acts, minlength=A)
        lp = logp.cpu().numpy()
        close(lp, np.log(p[acts]), 1e-5, 1e-5)
    n = B * n_rounds
    assert counts[2] == 0
    mask = p > 0
    chi2 = (((counts - n * p) ** 2)[mask] / (n * p[mask])).sum()
    assert chi2 < 30.0, (chi2, counts / n, p)          # 3 dof, p ~ 1e-6
    # same (seed, counter) -> same draw; different counter -> different stream
    a1, a2 = torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda")
    ops.actor_act(params, desc, dev(x), dev(avail), B, False, 99, 5, a1, logp)
    ops.actor_act(params, desc, dev(x), dev(avail), B, False, 99, 5, a2, logp)
    np.testing.assert_array_equal(a1.cpu().numpy(), a2.cpu().numpy())


@pytest.mark.parametrize("D,A,relu,LN,fn,B", NET_CASES)
def test_mlp_backward_vs_autograd(ops, D, A, relu, LN, fn, B):
    torch.manual_seed(D * 100 + A + 1)
    a = O.default_args(use_ReLU=relu, layer_N=LN, use_feature_normalization=fn)
    net = O.ActorRef(a, D, A) if A > 1 else O.CriticRef(a, D)
    _randomize(net, 9)
    head = "act.action_out.linear" if A > 1 else "v_out"
    desc = ops.net_desc(D, A, LN, relu, fn)
    params, layout, P = _flat_from_module(ops, net, desc, head)
    rng = np.random.default_rng(B + 1)
    n_rows = B + 17
    x = rng.standard_normal((n_rows, D)).astype(np.float32) * 2
    rows = rng.permutation(n_rows)[:B].astype(np.int32)
    dout = (rng.standard_normal((B, A)) / np.sqrt(B)).astype(np.float32)
    xt = torch.from_numpy(x[rows])
    if A > 1:
        feats, _ = net.features(xt, None, None)
        out = net.act.action_out.linear(feats)
    else:
        out = net(xt, None, None)[0]
    (out * torch.from_numpy(dout)).sum().backward()
    ref = {k: p_.grad.numpy() for k, p_ in net.named_parameters() if p_.grad is not None}
    n_slabs = ops.mlp_backward_slabs(B)
    stride = ((P + 255) // 256) * 256 + 256
    slabs = torch.full((n_slabs, stride), float("nan"), device="cuda")
    slabs[:, 256:256 + P] = 0.0      # contract: the caller zero-initialises its column range once (wide inputs write fewer rows of W1)
    ops.mlp_backward(params, desc, dev(x), dev(rows, torch.int32), B, dev(dout), slabs, stride, 256)
    grad = torch.zeros(stride, device="cuda")
    ops.slab_reduce(slabs[:, 256:].contiguous(), n_slabs, stride - 256, P, grad)
    gflat = grad.cpu().numpy()
    assert np.isfinite(gflat[:P]).all()
    assert torch.isnan(slabs[:, :256]).all() and torch.isnan(slabs[:, 256 + P:]).all()    # wrote only its columns
    from mappo_amd import flat
    for key, off, shape in layout:
        got = gflat[off: off + int(np.prod(shape))].reshape(shape)
        close_rel_max(got, ref[key], 1e-4, f"grad {key} (D={D} A={A} B={B})")


def test_mlp_full_size_linearity(ops):
    """BASELINE config-2 size (76 800 rows): backward is linear in dout and matches a finite slab count."""
    D, A, B = 54, 1, 76800
    torch.manual_seed(0)
    a = O.default_args()
    net = O.CriticRef(a, D); _randomize(net, 2)
    desc = ops.net_desc(D, A)
    params, layout, P = _flat_from_module(ops, net, desc, "v_out")
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, D, device="cuda", generator=g)
    d1 = torch.randn(B, 1, device="cuda", generator=g) / B
    d2 = torch.randn(B, 1, device="cuda", generator=g) / B
    n_slabs = ops.mlp_backward_slabs(B)
    outs = []
    for dd in (d1, d2, (d1 - 3 * d2).contiguous()):
        slabs = torch.zeros(n_slabs, P, device="cuda")
        ops.mlp_backward(params, desc, x, None, B, dd, slabs, P, 0)
        grad = torch.zeros(P, device="cuda")
        ops.slab_reduce(slabs, n_slabs, P, P, grad)
        outs.append(grad)
    close_rel_max(outs[2], (outs[0] - 3 * outs[1]).cpu().numpy(), 1e-4, "linearity")
    # forward at full size vs the oracle on a strided subset
    out = torch.zeros(B, 1, device="cuda")
    ops.mlp_forward(params, desc, x, None, B, out)
    idx = torch.arange(0, B, 97)
    with torch.no_grad():
        ref = net(x.cpu()[idx], None, None)[0].numpy()
    close(out.cpu()[idx], ref, 1e-5, 2e-6)


def _relu_margin(net, x):
    """min |pre-activation| per row over the trunk's Linear layers (CPU oracle network).  ReLU'(z) jumps at z = 0: a row with a
    pre-activation inside the fp32 rounding band of a D-term dot product may take the other branch on the GPU than in the
    CPU reference, and ONE such element moves every gradient below it by one sample's worth (~ 1 / sqrt(B) of the entry, about
    1e-3 of the largest at B = 70 000) — measured: scripts/dbg_wide_parity.py, a handful of flips per launch at B >= 16 384 and
    D = 512, in either direction between any two implementations (fused, unfused, float64 autograd)."""
    margins = []
    hooks = [m.register_forward_hook(lambda mod, inp, out: margins.append(out.detach().abs().min(dim=1).values))
             for m in net.base.mlp.modules() if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        net.base(x)
    for h in hooks:
        h.remove()
    return torch.stack(margins).min(dim=0).values.numpy()


@pytest.mark.parametrize("D,S,A,relu,B,with_rows", [(18, 54, 5, True, 3072, True), (18, 54, 5, False, 500, False),
                                                    (30, 48, 9, True, 333, True), (64, 64, 18, True, 100, False),
                                                    (18, 54, 5, True, 76800, False), (176, 322, 18, True, 700, True),
                                                    (512, 512, 5, True, 260, False),
                                                    # wide inputs in their STEADY STATE: more 16-row tiles than the 2 048 waves of a
                                                    # launch, so wide_l1_fwd16_kernel refills its row registers in place, and
                                                    # mlp_update16x_kernel / wide_l1_bwd16_kernel walk several tiles per wave
                                                    # (configs[3] / configs[4] sizes; ragged last tile; gathered rows in the second)
                                                    (512, 512, 5, True, 70001, False), (176, 322, 18, True, 47019, True),
                                                    (130, 65, 3, False, 33333, False),
                                                    # the remaining exact chunk counts of wide_l1_fwd16_kernel (4, 5, 7 chunks; in_dim % 4 != 0)
                                                    (256, 300, 5, True, 33011, True), (448, 401, 4, True, 33017, False)])
def test_fused_update_kernels_vs_unfused_and_autograd(ops, D, S, A, relu, B, with_rows):
    """mappo_actor_update / mappo_critic_update (forward + in-kernel PPO loss + backward in one launch) against
    (a) the standalone sequence mlp_forward -> ppo_loss_fwd_bwd -> mlp_backward and (b) torch autograd through
    the oracle networks with the reference's loss expressions (mlp.py:18-55, r_mappo.py:91-164)."""
    torch.manual_seed(B + D)
    rng = np.random.default_rng(B + A)
    f = np.float32
    a = O.default_args(use_ReLU=relu)
    actor, critic = O.ActorRef(a, D, A), O.CriticRef(a, S)
    _randomize(actor, 3); _randomize(critic, 4)
    da, dc = ops.net_desc(D, A, 1, relu, True), ops.net_desc(S, 1, 1, relu, True)
    pa, la, Pa = _flat_from_module(ops, actor, da, "act.action_out.linear")
    pc, lc, Pc = _flat_from_module(ops, critic, dc, "v_out")
    n_rows = B + 64 if with_rows else B
    rows = rng.permutation(n_rows)[:B].astype(np.int32) if with_rows else np.arange(B, dtype=np.int32)
    obs = rng.standard_normal((n_rows, D)).astype(f)
    sobs = rng.standard_normal((n_rows, S)).astype(f)
    avail = (rng.random((n_rows, A)) > 0.3).astype(f)
    actions = rng.integers(0, A, n_rows).astype(f)
    avail[np.arange(n_rows), actions.astype(int)] = 1.0
    old_logp = (-np.abs(rng.standard_normal(n_rows)) * 0.3 - np.log(A)).astype(f)
    adv = rng.standard_normal(n_rows).astype(f)
    active = (rng.random(n_rows) > 0.25).astype(f)
    if relu and B > 4000 and D > 64:
        # large wide-input cases: rows whose ReLU pre-activations sit within 1e-4 of zero (about 1 % of them) are made inactive,
        # so that all three implementations differentiate the SAME function (see _relu_margin); with the active-mask means
        # (r_mappo.py:84,130-141) an inactive row contributes exactly nothing to any gradient
        unsafe = np.minimum(_relu_margin(actor, torch.from_numpy(obs)), _relu_margin(critic, torch.from_numpy(sobs))) < 1e-4
        assert 0 < unsafe.mean() < 0.05
        active[unsafe] = 0.0
    ret = (rng.standard_normal(n_rows) * 3).astype(f)
    ret[rng.random(n_rows) > 0.9] *= 20
    with torch.no_grad():
        v_now = critic(torch.from_numpy(sobs), None, None)[0].numpy().reshape(-1)
    v_old = (v_now + rng.standard_normal(n_rows) * 0.25).astype(f)
    vn = O.ValueNormRef(); vn.update(ret[:50].reshape(-1, 1)); vn.update(ret[rows].reshape(-1, 1))
    d_rows = dev(rows, torch.int32) if with_rows else None
    g = dict(obs=dev(obs), sobs=dev(sobs), avail=dev(avail), actions=dev(actions), old=dev(old_logp), adv=dev(adv),
             active=dev(active), ret=dev(ret), vold=dev(v_old), vn=dev(vn.state()))
    mom = torch.zeros(4, dtype=torch.float64, device="cuda")
    ops.minibatch_moments(g["ret"], g["active"], d_rows, B, mom)
    cfg = ops.ppo_cfg(a)
    ns = ops.mlp_backward_slabs(B)
    P = ((Pa + 255) // 256) * 256 + ((Pc + 255) // 256) * 256
    col_c = ((Pa + 255) // 256) * 256
    # (1) fused
    slabs = torch.zeros(ns, P, device="cuda")
    part_a, part_c = ops.update_partials("cuda"), ops.update_partials("cuda")
    ops.actor_update(pa, da, g["obs"], d_rows, B, g["avail"], g["actions"], g["old"], g["adv"], g["active"], mom, cfg, slabs, P, 0, part_a)
    ops.critic_update(pc, dc, g["sobs"], d_rows, B, g["vold"], g["ret"], g["active"], g["vn"], mom, cfg, slabs, P, col_c, part_c)
    stats_f = torch.zeros(6, dtype=torch.float64, device="cuda")
    ops.update_stats(part_a, ns, part_c, ns, mom, cfg, stats_f)
    grad_f = torch.zeros(P, device="cuda")
    ops.slab_reduce(slabs, ns, P, P, grad_f)
    # (1b) both networks in ONE launch (mappo_actor_critic_update), in_dim <= 64
    if D <= 64 and S <= 64:
        nd = ops.dual_update_slabs(da, dc, B)
        slabs_d = torch.zeros(nd, P, device="cuda")
        pda, pdc = ops.update_partials("cuda"), ops.update_partials("cuda")
        ops.actor_critic_update(pa, da, g["obs"], pc, dc, g["sobs"], d_rows, B, g["avail"], g["actions"], g["old"], g["adv"], g["active"],
                                g["vold"], g["ret"], g["vn"], mom, cfg, slabs_d, P, 0, col_c, pda, pdc)
        stats_d = torch.zeros(6, dtype=torch.float64, device="cuda")
        ops.update_stats(pda, nd, pdc, nd, mom, cfg, stats_d)
        grad_d = torch.zeros(P, device="cuda")
        ops.slab_reduce(slabs_d, nd, P, P, grad_d)
        close(stats_d, stats_f, 1e-6, 1e-9, "stats dual vs separate launches")    # lanes accumulate a handful of samples in fp32, the rest in double
        close_rel_max(grad_d, grad_f.cpu().numpy(), 2e-6, "grad dual vs separate launches")
    # (2) unfused
    logits, values = torch.zeros(B, A, device="cuda"), torch.zeros(B, device="cuda")
    ops.mlp_forward(pa, da, g["obs"], d_rows, B, logits)
    ops.mlp_forward(pc, dc, g["sobs"], d_rows, B, values)
    dl, dv = torch.zeros(B, A, device="cuda"), torch.zeros(B, device="cuda")
    stats_u = torch.zeros(6, dtype=torch.float64, device="cuda")
    ops.ppo_loss_fwd_bwd(logits, values, d_rows, g["avail"], g["actions"], g["old"], g["adv"], g["active"], g["vold"], g["ret"],
                         g["vn"], mom, dl, dv, stats_u, cfg)
    slabs_u = torch.zeros(ns, P, device="cuda")
    ops.mlp_backward(pa, da, g["obs"], d_rows, B, dl, slabs_u, P, 0)
    ops.mlp_backward(pc, dc, g["sobs"], d_rows, B, dv.view(B, 1), slabs_u, P, col_c)
    grad_u = torch.zeros(P, device="cuda")
    ops.slab_reduce(slabs_u, ns, P, P, grad_u)
    close(stats_f, stats_u, 1e-6, 1e-9, "stats fused vs unfused")
    close_rel_max(grad_f[:Pa], grad_u[:Pa].cpu().numpy(), 2e-5, "actor grad fused vs unfused")
    close_rel_max(grad_f[col_c:col_c + Pc], grad_u[col_c:col_c + Pc].cpu().numpy(), 2e-5, "critic grad fused vs unfused")
    # (3) autograd through the oracle (CPU torch; the narrow full-size case is covered by the e2e tests instead)
    if B <= 4000 or D > 64:
        if B > 4000:
            # tens of thousands of terms per gradient entry: the CPU reference itself runs in float64, so that the tolerance
            # below bounds the KERNELS' fp32 summation error and not the reference's
            import copy
            actor, critic = copy.deepcopy(actor).double(), copy.deepcopy(critic).double()
            t = lambda x: torch.from_numpy(x).double()
        else:
            t = lambda x: torch.from_numpy(x)
        r_ = rows.astype(np.int64)
        lp, ent, _ = actor.evaluate_actions(t(obs[r_]), None, t(actions[r_]).view(-1, 1), None, t(avail[r_]), t(active[r_]).view(-1, 1))
        vals = critic(t(sobs[r_]), None, None)[0]
        act_t, adv_t, old_t = t(active[r_]).view(-1, 1), t(adv[r_]).view(-1, 1), t(old_logp[r_]).view(-1, 1)
        imp = torch.exp(lp - old_t)
        surr = torch.min(imp * adv_t, torch.clamp(imp, 1 - a.clip_param, 1 + a.clip_param) * adv_t)
        pl = (-surr * act_t).sum() / act_t.sum()
        (pl - a.entropy_coef * ent).backward()
        tgt = vn.normalize(t(ret[r_]).view(-1, 1))
        vo = t(v_old[r_]).view(-1, 1)
        vclip = vo + (vals - vo).clamp(-a.clip_param, a.clip_param)
        l = torch.max(O.huber_ref(tgt - vals, a.huber_delta), O.huber_ref(tgt - vclip, a.huber_delta))
        vl = (l * act_t).sum() / act_t.sum()
        (vl * a.value_loss_coef).backward()
        s = stats_f.cpu().numpy()
        close([s[0], s[1], s[2], s[3]], [vl.item(), pl.item(), ent.item(), imp.mean().item()], 1e-5, 1e-7, "stats vs autograd")
        gf = grad_f.cpu().numpy()
        for key, off, shape in la:
            close_rel_max(gf[off: off + int(np.prod(shape))].reshape(shape), dict(actor.named_parameters())[key].grad.numpy(), 1e-4, f"actor {key}")
        for key, off, shape in lc:
            close_rel_max(gf[col_c + off: col_c + off + int(np.prod(shape))].reshape(shape), dict(critic.named_parameters())[key].grad.numpy(), 1e-4, f"critic {key}")


# ------------------------------------------------------------------------------------------------------
def test_clip_adam_vs_oracle_and_torch(ops):
    rng = np.random.default_rng(2)
    Pa, Pc = 768, 1280
    P = Pa + Pc
    p0 = rng.standard_normal(P).astype(np.float32)
    params = dev(p0)
    m, v = torch.zeros(P, device="cuda"), torch.zeros(P, device="cuda")
    hyper = torch.tensor([[7e-4, 0.9, 0.999, 1e-5, 0.0, 10.0, 1.0, 1.0], [5e-4, 0.9, 0.999, 1e-5, 0.0, 0.5, 1.0, 1.0]],
                         dtype=torch.float32).cuda()
    step = torch.zeros(2, dtype=torch.int32, device="cuda")
    norms = torch.zeros(2, device="cuda")
    ws = ops.optim_workspace(P, params.device)
    ref_p = [p0[:Pa].astype(np.float64), p0[Pa:].astype(np.float64)]
    ref_m = [np.zeros(Pa), np.zeros(Pc)]; ref_v = [np.zeros(Pa), np.zeros(Pc)]
    tp = [torch.nn.Parameter(torch.from_numpy(p0[:Pa].copy())), torch.nn.Parameter(torch.from_numpy(p0[Pa:].copy()))]
    topt = [torch.optim.Adam([tp[0]], lr=7e-4, eps=1e-5), torch.optim.Adam([tp[1]], lr=5e-4, eps=1e-5)]
    for it in range(4):
        gr = (rng.standard_normal(P) * (0.5 if it % 2 else 0.01)).astype(np.float32)
        ops.clip_adam(params, dev(gr), m, v, [0, Pa, P], hyper, step, norms, ws)
        for s, (lo, hi, lr, mx) in enumerate(((0, Pa, 7e-4, 10.0), (Pa, P, 5e-4, 0.5))):
            ref_p[s], ref_m[s], ref_v[s], n = O.clip_adam_ref(ref_p[s], gr[lo:hi], ref_m[s], ref_v[s], it, lr, mx)
            close(norms[s], n, 1e-6, 1e-9)
            close(params[lo:hi], ref_p[s], 1e-6, 1e-7, f"params seg {s} it {it}")
            tp[s].grad = torch.from_numpy(gr[lo:hi].copy())
            torch.nn.utils.clip_grad_norm_([tp[s]], mx)
            topt[s].step()
            close(params[lo:hi], tp[s].detach().numpy(), 1e-6, 1e-7, f"torch seg {s} it {it}")
    assert step.cpu().tolist() == [4, 4]
    # a disabled segment (update_actor=False path) is left untouched
    hyper[0, 7] = 0.0
    before = params[:Pa].clone()
    ops.clip_adam(params, dev(rng.standard_normal(P).astype(np.float32)), m, v, [0, Pa, P], hyper, step, norms, ws)
    np.testing.assert_array_equal(before.cpu().numpy(), params[:Pa].cpu().numpy())
    assert step.cpu().tolist() == [4, 5]


def test_ops_reject_cpu_tensors(ops):
    from mappo_amd._lib import MappoHipError
    with pytest.raises(MappoHipError):
        ops.adv_normalize(torch.zeros(8), torch.zeros(3, dtype=torch.float64))


@pytest.mark.parametrize("centralized", [True, False])
def test_insert_mpe_kernel(ops, centralized):
    """K1: one launch == the four slot writes of mpe_runner.insert / SharedReplayBuffer.insert, bit for bit, with strided
    (slice of a wider block) obs, broadcast rewards and bool dones."""
    N, M, D = 37, 3, 18
    g = torch.Generator(device="cuda").manual_seed(0)
    blk = torch.randn(N, M * D + 1, device="cuda", generator=g)
    obs = blk[:, :M * D].view(N, M, D)
    rew = blk[:, M * D:].view(N, 1, 1).expand(N, M, 1)
    dones = torch.rand(N, M, device="cuda", generator=g) > 0.5
    S = M * D if centralized else D
    od, sd, rd, md = (torch.full(s, float("nan"), device="cuda") for s in ((N, M, D), (N, M, S), (N, M, 1), (N, M, 1)))
    ops.insert_mpe(obs, rew, dones, od, sd, rd, md, centralized)
    share_ref = obs.reshape(N, 1, -1).expand(N, M, -1) if centralized else obs
    np.testing.assert_array_equal(od.cpu().numpy(), obs.cpu().numpy())
    np.testing.assert_array_equal(sd.cpu().numpy(), share_ref.cpu().numpy())
    np.testing.assert_array_equal(rd.cpu().numpy(), rew.cpu().numpy())
    np.testing.assert_array_equal(md.cpu().numpy(), (~dones).float().view(N, M, 1).cpu().numpy())


@pytest.mark.parametrize("centralized", [True, False])
def test_insert_mpe_rnn_kernel(ops, centralized):
    """K1, recurrent policies: mappo_insert_mpe_rnn == mappo_insert_mpe + rnn_states * (1 - done) written to the slot
    (mpe_runner.py:126-128, shared_buffer.py:96-97), bit for bit."""
    N, M, D, H = 37, 3, 18, 64
    g = torch.Generator(device="cuda").manual_seed(0)
    blk = torch.randn(N, M * D + 1, device="cuda", generator=g)
    obs = blk[:, :M * D].view(N, M, D)
    rew = blk[:, M * D:].view(N, 1, 1).expand(N, M, 1)
    dones = torch.rand(N, M, device="cuda", generator=g) > 0.5
    ha, hc = torch.randn(N * M, 1, H, device="cuda", generator=g), torch.randn(N * M, 1, H, device="cuda", generator=g)
    S = M * D if centralized else D
    nan = lambda *s: torch.full(s, float("nan"), device="cuda")
    od, sd, rd, md, da, dc = nan(N, M, D), nan(N, M, S), nan(N, M, 1), nan(N, M, 1), nan(N, M, 1, H), nan(N, M, 1, H)
    ops.insert_mpe_rnn(obs, rew, dones, od, sd, rd, md, centralized, ha, hc, da, dc)
    od2, sd2, rd2, md2 = nan(N, M, D), nan(N, M, S), nan(N, M, 1), nan(N, M, 1)
    ops.insert_mpe(obs, rew, dones, od2, sd2, rd2, md2, centralized)
    for a, b in ((od, od2), (sd, sd2), (rd, rd2), (md, md2)):
        np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
    keep = (~dones).float().view(N, M, 1, 1)
    np.testing.assert_array_equal(da.cpu().numpy(), (ha.view(N, M, 1, H) * keep).cpu().numpy())
    np.testing.assert_array_equal(dc.cpu().numpy(), (hc.view(N, M, 1, H) * keep).cpu().numpy())


@pytest.mark.parametrize("centralized", [True, False])
def test_rollout_step_fused_matches_separate_launches(ops, centralized):
    """mappo_rollout_step (one launch: insert of the env output + get_actions + get_values, rows read in place from a
    strided env block) == mappo_insert_mpe, then mappo_actor_act / mappo_mlp_forward on the buffer slots, bit for bit."""
    N, M, D, A = 70, 3, 18, 5
    R = N * M
    g = torch.Generator(device="cuda").manual_seed(1)
    blk = torch.randn(N, M * D + 1, device="cuda", generator=g)
    obs = blk[:, :M * D].view(N, M, D)
    rew = blk[:, M * D:].view(N, 1, 1).expand(N, M, 1)[..., 0]
    dones = torch.rand(N, M, device="cuda", generator=g) > 0.5
    S = M * D if centralized else D
    da, dc = ops.net_desc(D, A), ops.net_desc(S, 1)
    pa = torch.randn(ops.net_param_count(da), device="cuda", generator=g) * 0.2
    pc = torch.randn(ops.net_param_count(dc), device="cuda", generator=g) * 0.2
    # reference: separate launches
    od, sd, rd, md = (torch.empty(s, device="cuda") for s in ((N, M, D), (N, M, S), (N, M, 1), (N, M, 1)))
    ops.insert_mpe(obs, rew, dones, od, sd, rd, md, centralized)
    act0, lp0, v0 = torch.empty(R, device="cuda"), torch.empty(R, device="cuda"), torch.empty(R, 1, device="cuda")
    ops.actor_act(pa, da, od.view(R, D), None, R, False, 1234, 7, act0, lp0)
    ops.mlp_forward(pc, dc, sd.view(R, S), None, R, v0)
    # fused
    od1, sd1, rd1, md1 = (torch.full(s, float("nan"), device="cuda") for s in ((N, M, D), (N, M, S), (N, M, 1), (N, M, 1)))
    act1, lp1, v1 = torch.empty(R, device="cuda"), torch.empty(R, device="cuda"), torch.empty(R, device="cuda")
    ins = dict(obs_dst=od1, share_dst=sd1, rewards=(rew, rew.stride(0), rew.stride(1)), dones=(dones, dones.stride(0), dones.stride(1)),
               rew_dst=rd1, mask_dst=md1, centralized=centralized)
    ops.rollout_step(pa, da, pc, dc, (obs, obs.stride(0), obs.stride(1)), (obs, obs.stride(0), 0 if centralized else obs.stride(1)), M, R,
                     None, False, 1234, 7, None, act1, lp1, v1, ins)
    for a_, b_ in ((od, od1), (sd, sd1), (rd, rd1), (md, md1)):
        np.testing.assert_array_equal(a_.cpu().numpy(), b_.cpu().numpy())

    def same_policy_outputs(act, lp, v):
        # the step kernel runs the networks on 16x16x4 MFMA tiles (mlp_fwd16.h): same arithmetic, different order of the
        # fp32 sums, so values / log-probs agree to rounding and a sampled action may flip only at a CDF boundary
        same = (act == act0).float().mean().item()
        assert same >= 0.99, same
        keep = (act == act0).cpu().numpy()
        np.testing.assert_allclose(lp.cpu().numpy()[keep], lp0.cpu().numpy()[keep], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(v.cpu().numpy(), v0.view(R).cpu().numpy(), rtol=2e-5, atol=2e-6)

    same_policy_outputs(act1, lp1, v1)
    # without the insert, reading the (contiguous) slots
    act2, lp2, v2 = torch.empty(R, device="cuda"), torch.empty(R, device="cuda"), torch.empty(R, device="cuda")
    ops.rollout_step(pa, da, pc, dc, (od, 0, 0), (sd, 0, 0), 0, R, None, False, 1234, 7, None, act2, lp2, v2, None)
    same_policy_outputs(act2, lp2, v2)


@pytest.mark.parametrize("D,N,M,ins_on", [(512, 37, 7, True), (256, 300, 5, True), (512, 9, 3, False), (512, 1000, 32, True)])
def test_wide_full_rollout_step_matches_streamed_and_separate_insert(ops, monkeypatch, D, N, M, ins_on):
    """in_dim 256 / 512 on both networks: mappo_rollout_step takes wide_rollout_full_kernel (W1' staged whole, the insert's row copies
    riding on the forward's loads).  Against the streamed form (MAPPO_WIDE_FULL_STEP=0: mappo_insert_mpe + wide_rollout_step_kernel):
    buffer slots bit for bit, values / log-probs to fp32 summation order, actions equal but for CDF-boundary flips.  Sizes: ragged
    last tile, waves with two tiles (32 000 rows = 2 000 tiles on 1 024 waves per network), no insert."""
    A = 5
    R = N * M
    g = torch.Generator(device="cuda").manual_seed(D + N)
    if N % 2:                                              # the env's output as a strided view (thread stride M D + 1: rows 4-byte aligned only)
        obs = (torch.randn(N, M * D + 1, device="cuda", generator=g) * 1.5 + 0.3)[:, :M * D].view(N, M, D)
    else:
        obs = torch.randn(N, M, D, device="cuda", generator=g) * 1.5 + 0.3
    rew = torch.randn(N, 1, device="cuda", generator=g).expand(N, M)
    dones = torch.rand(N, M, device="cuda", generator=g) > 0.5
    da, dc = ops.net_desc(D, A), ops.net_desc(D, 1)
    pa = torch.randn(ops.net_param_count(da), device="cuda", generator=g) * 0.1
    pc = torch.randn(ops.net_param_count(dc), device="cuda", generator=g) * 0.1
    out = []
    for full in ("0", "1"):
        monkeypatch.setenv("MAPPO_WIDE_FULL_STEP", full)
        od, sd, rd, md = (torch.full(s_, float("nan"), device="cuda") for s_ in ((N, M, D), (N, M, D), (N, M, 1), (N, M, 1)))
        act, lp, v = torch.empty(R, device="cuda"), torch.empty(R, device="cuda"), torch.empty(R, device="cuda")
        ins = dict(obs_dst=od, share_dst=sd, rewards=(rew, rew.stride(0), rew.stride(1)), dones=(dones, dones.stride(0), dones.stride(1)),
                   rew_dst=rd, mask_dst=md, centralized=False) if ins_on else None
        ops.rollout_step(pa, da, pc, dc, (obs, obs.stride(0), obs.stride(1)), (obs, obs.stride(0), obs.stride(1)), M, R,
                         None, False, 4321, 11, None, act, lp, v, ins)
        torch.cuda.synchronize()
        out.append((od, sd, rd, md, act, lp, v))
    a0, a1 = out
    if ins_on:
        for x0, x1 in zip(a0[:4], a1[:4]):
            np.testing.assert_array_equal(x0.cpu().numpy(), x1.cpu().numpy())
        np.testing.assert_array_equal(a1[0].cpu().numpy(), obs.cpu().numpy())
    same = (a0[4] == a1[4]).cpu().numpy()
    assert same.mean() >= 0.995, same.mean()
    np.testing.assert_allclose(a1[5].cpu().numpy()[same], a0[5].cpu().numpy()[same], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(a1[6].cpu().numpy(), a0[6].cpu().numpy(), rtol=2e-5, atol=2e-6)


def test_reduce_clip_adam_matches_slab_reduce_then_clip_adam(ops):
    """mappo_reduce_clip_adam (2 launches) == mappo_slab_reduce + mappo_clip_adam (4 launches): the reduced gradient bit
    for bit, norms / parameters / moments to fp32 rounding of the norm (its double partial sums associate differently)."""
    g = torch.Generator(device="cuda").manual_seed(3)
    Pa, Pc, n_slabs = 6144, 8448, 37
    P = Pa + Pc
    slabs = torch.randn(n_slabs, P, device="cuda", generator=g) * 0.05
    hyper = torch.tensor([[7e-4, 0.9, 0.999, 1e-5, 0.0, 10.0, 1.0, 1.0], [5e-4, 0.9, 0.999, 1e-5, 0.0, 0.5, 1.0, 1.0]],
                         dtype=torch.float32).cuda()
    p0 = torch.randn(P, device="cuda", generator=g)
    out = []
    for fused in (False, True):
        params = p0.clone(); m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
        step = torch.zeros(2, dtype=torch.int32, device="cuda"); norms = torch.zeros(2, device="cuda")
        acc = torch.zeros(2, dtype=torch.float64, device="cuda"); grad = torch.empty(P, device="cuda")
        ws = ops.optim_workspace(P, params.device)
        for it in range(3):
            if fused:
                ops.reduce_clip_adam(slabs, n_slabs, P, params, grad, m, v, [0, Pa, P], hyper, step, norms, ws, norm_acc=acc)
            else:
                ops.slab_reduce(slabs, n_slabs, P, P, grad)
                ops.clip_adam(params, grad, m, v, [0, Pa, P], hyper, step, norms, ws, norm_acc=acc)
        out.append([t.cpu().numpy() for t in (grad, params, m, v, norms, acc, step)])
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][6], out[1][6])
    for a_, b_ in zip(out[0][1:6], out[1][1:6]):
        np.testing.assert_allclose(a_, b_, rtol=2e-6, atol=1e-7)


def test_rollout_step_values_only_and_copy_batch(ops):
    """actions == logp == NULL: critic + insert only (the bootstrap value of compute()); mappo_copy_batch == the copies."""
    N, M, D = 40, 3, 18
    R = N * M
    g = torch.Generator(device="cuda").manual_seed(5)
    blk = torch.randn(N, M * D + 1, device="cuda", generator=g)
    obs = blk[:, :M * D].view(N, M, D)
    rew = blk[:, M * D:].view(N, 1, 1).expand(N, M, 1)[..., 0]
    dones = torch.rand(N, M, device="cuda", generator=g) > 0.5
    da, dc = ops.net_desc(D, 5), ops.net_desc(M * D, 1)
    pa = torch.randn(ops.net_param_count(da), device="cuda", generator=g) * 0.2
    pc = torch.randn(ops.net_param_count(dc), device="cuda", generator=g) * 0.2
    od, sd, rd, md = (torch.empty(s, device="cuda") for s in ((N, M, D), (N, M, M * D), (N, M, 1), (N, M, 1)))
    ops.insert_mpe(obs, rew, dones, od, sd, rd, md, True)
    v0 = torch.empty(R, 1, device="cuda")
    ops.mlp_forward(pc, dc, sd.view(R, M * D), None, R, v0)
    od1, sd1, rd1, md1 = (torch.full(s, float("nan"), device="cuda") for s in ((N, M, D), (N, M, M * D), (N, M, 1), (N, M, 1)))
    v1 = torch.empty(R, device="cuda")
    ins = dict(obs_dst=od1, share_dst=sd1, rewards=(rew, rew.stride(0), rew.stride(1)), dones=(dones, dones.stride(0), dones.stride(1)),
               rew_dst=rd1, mask_dst=md1, centralized=True)
    ops.rollout_step(pa, da, pc, dc, (obs, obs.stride(0), obs.stride(1)), (obs, obs.stride(0), 0), M, R, None, False, 0, 0, None, None, None, v1, ins)
    for a_, b_ in ((od, od1), (sd, sd1), (rd, rd1), (md, md1)):
        np.testing.assert_array_equal(a_.cpu().numpy(), b_.cpu().numpy())
    np.testing.assert_allclose(v1.cpu().numpy(), v0.view(R).cpu().numpy(), rtol=2e-5, atol=2e-6)
    srcs = [torch.randn(n, device="cuda", generator=g) for n in (7, 1024, 4099, 12)]
    dsts = [torch.zeros_like(t) for t in srcs]
    ops.copy_batch(list(zip(dsts, srcs)))
    for a_, b_ in zip(srcs, dsts):
        np.testing.assert_array_equal(a_.cpu().numpy(), b_.cpu().numpy())


@pytest.mark.parametrize("Da,Dc,A,LN,relu,fnorm,R", [(1, 1, 1, 1, True, True, 5), (64, 64, 32, 2, False, False, 130), (33, 7, 9, 0, True, True, 47),
                                                      (18, 54, 5, 1, False, True, 3072), (20, 60, 17, 2, True, False, 16),
                                                      # wide inputs (both networks 65..512): wide_rollout_step_kernel, the two networks by
                                                      # workgroup role; widths that are not multiples of 4, ragged last tile, > 8 tiles
                                                      (512, 512, 5, 1, True, True, 4100), (130, 70, 9, 0, False, True, 37), (176, 322, 18, 2, True, False, 300)])
def test_rollout_step_shapes_vs_forward_kernels(ops, Da, Dc, A, LN, relu, fnorm, R):
    """The 16x16x4 step kernel against mlp_forward / actor_act over network shapes (odd / maximal widths, partial tiles,
    tanh, no feature norm): values and deterministic log-probs to fp32 rounding, argmax actions equal unless two logits tie
    to rounding."""
    g = torch.Generator(device="cuda").manual_seed(11)
    da = ops.net_desc(Da, A, layer_N=LN, use_relu=relu, use_feature_norm=fnorm)
    dc = ops.net_desc(Dc, 1, layer_N=LN, use_relu=relu, use_feature_norm=fnorm)
    pa = torch.randn(ops.net_param_count(da), device="cuda", generator=g) * 0.3
    pc = torch.randn(ops.net_param_count(dc), device="cuda", generator=g) * 0.3
    obs = torch.randn(R, Da, device="cuda", generator=g)
    sh = torch.randn(R, Dc, device="cuda", generator=g)
    avail = (torch.rand(R, A, device="cuda", generator=g) > 0.3).float()
    avail[:, 0] = 1.0
    act0, lp0, v0 = torch.empty(R, device="cuda"), torch.empty(R, device="cuda"), torch.empty(R, 1, device="cuda")
    ops.actor_act(pa, da, obs, avail, R, True, 1, 0, act0, lp0)
    ops.mlp_forward(pc, dc, sh, None, R, v0)
    act1, lp1, v1 = torch.empty(R, device="cuda"), torch.empty(R, device="cuda"), torch.empty(R, device="cuda")
    ops.rollout_step(pa, da, pc, dc, (obs, 0, 0), (sh, 0, 0), 0, R, avail, True, 1, 0, None, act1, lp1, v1, None)
    np.testing.assert_allclose(v1.cpu().numpy(), v0.view(R).cpu().numpy(), rtol=3e-5, atol=3e-6)
    same = (act1 == act0).cpu().numpy()
    assert same.mean() >= 0.98, same.mean()
    np.testing.assert_allclose(lp1.cpu().numpy()[same], lp0.cpu().numpy()[same], rtol=3e-5, atol=3e-6)


@pytest.mark.parametrize("T,N,Ma,L,nmb,E", [(25, 16, 3, 10, 1, 3), (20, 6, 3, 10, 2, 2), (400, 8, 5, 10, 4, 2), (7, 5, 2, 3, 1, 1)])
def test_recurrent_rows_kernel(ops, T, N, Ma, L, nmb, E):
    """mappo_recurrent_rows (all epochs, one launch) == SharedReplayBuffer.recurrent_rows' index arithmetic
    (shared_buffer.py:385-494: chunks of the (n, m, t) order, time-major stacking), bit for bit, incl. T % L != 0."""
    R = N * Ma
    chunks = (T * R) // L
    g = torch.Generator(device="cuda").manual_seed(T + L)
    perm = torch.rand(E, chunks, device="cuda", generator=g).argsort(dim=1)
    rows, h0 = ops.recurrent_rows(perm, L, T, R, nmb)
    for e in range(E):
        want = O.recurrent_rows(T, R, nmb, L, perm[e].cpu().numpy())          # the oracle's restatement, pinned by golden/generators.npz
        assert len(want) == nmb
        for k, (ref_rows, ref_h0) in enumerate(want):
            np.testing.assert_array_equal(rows[e, k].cpu().numpy(), ref_rows.astype(np.int32))
            np.testing.assert_array_equal(h0[e, k].cpu().numpy(), ref_h0.astype(np.int32))


def test_dual_update_statistics_are_deterministic(ops):
    """Repeated launches of the actor+critic update on the same inputs (row gather, BASELINE config-2 size) give bit-identical
    gradients AND loss statistics, equal to the two single-network launches (regression: the waves' loss sums used to be read
    back after another wave could reuse their LDS slot, so a workgroup's value-loss partial came out short now and then)."""
    a = O.default_args()
    cfg = ops.ppo_cfg(a)
    B, NR = 76800, 77800
    torch.manual_seed(0)
    da, dc = ops.net_desc(18, 5), ops.net_desc(54, 1)
    Pa, Pc = ops.net_param_count(da), ops.net_param_count(dc)
    col_c = ((Pa + 255) // 256) * 256
    P = col_c + ((Pc + 255) // 256) * 256
    pa, pc = torch.randn(Pa, device="cuda") * 0.1, torch.randn(Pc, device="cuda") * 0.1
    obs, sobs = torch.randn(NR, 18, device="cuda"), torch.randn(NR, 54, device="cuda")
    ret, active = torch.randn(NR, device="cuda"), (torch.rand(NR, device="cuda") > 0.1).float()
    rows = torch.randperm(NR, device="cuda")[:B].to(torch.int32).contiguous()
    mom = torch.zeros(4, dtype=torch.float64, device="cuda")
    ops.minibatch_moments(ret, active, rows, B, mom)
    av = (torch.rand(NR, 5, device="cuda") > 0.2).float()
    av[:, 0] = 1
    act, olp = torch.zeros(NR, device="cuda"), -torch.rand(NR, device="cuda") - 1
    adv, vold, vn = torch.randn(NR, device="cuda"), torch.randn(NR, device="cuda"), torch.tensor([0., 1., 1.], device="cuda")
    nd, ns = ops.dual_update_slabs(da, dc, B), ops.mlp_backward_slabs(B)

    def run(dual):
        slabs = torch.zeros(max(nd, ns), P, device="cuda")
        pda, pdc = ops.update_partials("cuda"), ops.update_partials("cuda")
        if dual:
            ops.actor_critic_update(pa, da, obs, pc, dc, sobs, rows, B, av, act, olp, adv, active, vold, ret, vn, mom, cfg, slabs, P, 0, col_c, pda, pdc)
        else:
            ops.actor_update(pa, da, obs, rows, B, av, act, olp, adv, active, mom, cfg, slabs, P, 0, pda)
            ops.critic_update(pc, dc, sobs, rows, B, vold, ret, active, vn, mom, cfg, slabs, P, col_c, pdc)
        stats = torch.zeros(6, dtype=torch.float64, device="cuda")
        n = nd if dual else ns
        ops.update_stats(pda, n, pdc, n, mom, cfg, stats)
        return slabs.double().sum(0).cpu().numpy(), stats.cpu().numpy()

    g0, s0 = run(True)
    gs, ss = run(False)
    np.testing.assert_allclose(s0, ss, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(g0, gs, rtol=0, atol=1e-6 * np.abs(gs).max())
    for _ in range(25):
        g, s = run(True)
        assert np.array_equal(g, g0) and np.array_equal(s, s0)
    for _ in range(5):
        g, s = run(False)
        assert np.array_equal(g, gs) and np.array_equal(s, ss)


def test_wide_critic_layer0_statistics_deterministic_and_vs_oracle(ops):
    """mlp_update16x_kernel<R, 0, 2> — a wide-input critic (in_dim 65..512) with layer_N = 0 owns NO MFMA accumulators, so the
    accumulator-chunk reduction of its epilogue has zero trips: the barrier that orders the waves' loss partial sums before thread
    0 reads them must not depend on that loop (regression for a race that shortened a workgroup's value-loss partial now and
    then).  A buffer on which all eight waves of every workgroup have tiles; 30 launches bit-identical, and the value loss equal
    to the reference expression (r_mappo.py:52-89) evaluated by the oracle network on the CPU."""
    B, D = 40000, 130
    a = O.default_args(layer_N=0)
    cfg = ops.ppo_cfg(a)
    critic = O.CriticRef(a, D); _randomize(critic, 6)
    dc = ops.net_desc(D, 1, 0, True, True)
    pc, lc, Pc = _flat_from_module(ops, critic, dc, "v_out")
    P = ((Pc + 255) // 256) * 256
    rng = np.random.default_rng(5)
    f = np.float32
    sobs = rng.standard_normal((B, D)).astype(f)
    ret = (rng.standard_normal(B) * 3).astype(f)
    active = (rng.random(B) > 0.25).astype(f)
    with torch.no_grad():
        v_now = critic(torch.from_numpy(sobs), None, None)[0].numpy().reshape(-1)
    v_old = (v_now + rng.standard_normal(B) * 0.25).astype(f)
    vn = O.ValueNormRef(); vn.update(ret.reshape(-1, 1))
    g = dict(sobs=dev(sobs), ret=dev(ret), active=dev(active), vold=dev(v_old), vn=dev(vn.state()))
    mom = torch.zeros(4, dtype=torch.float64, device="cuda")
    ops.minibatch_moments(g["ret"], g["active"], None, B, mom)
    ns = ops.mlp_backward_slabs(B)

    def run():
        slabs = torch.zeros(ns, P, device="cuda")
        part = ops.update_partials("cuda")
        ops.critic_update(pc, dc, g["sobs"], None, B, g["vold"], g["ret"], g["active"], g["vn"], mom, cfg, slabs, P, 0, part)
        stats = torch.zeros(6, dtype=torch.float64, device="cuda")
        ops.update_stats(None, ns, part, ns, mom, cfg, stats)
        return slabs.double().sum(0).cpu().numpy(), stats.cpu().numpy()

    g0, s0 = run()
    for _ in range(30):
        gi, si = run()
        assert np.array_equal(gi, g0) and np.array_equal(si, s0)
    t = torch.from_numpy
    vals = critic(t(sobs), None, None)[0]
    act_t = t(active).view(-1, 1)
    tgt = vn.normalize(t(ret).view(-1, 1))
    vo = t(v_old).view(-1, 1)
    vclip = vo + (vals - vo).clamp(-a.clip_param, a.clip_param)
    l = torch.max(O.huber_ref(tgt - vals, a.huber_delta), O.huber_ref(tgt - vclip, a.huber_delta))
    vl = (l * act_t).sum() / act_t.sum()
    (vl * a.value_loss_coef).backward()
    close(s0[0], vl.item(), 1e-5, 1e-7, "value loss vs oracle")
    for key, off, shape in lc:
        close_rel_max(g0[off: off + int(np.prod(shape))].reshape(shape), dict(critic.named_parameters())[key].grad.numpy(), 1e-4, f"critic {key}")
