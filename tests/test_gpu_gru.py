"""GPU parity tests of the recurrent (GRU) path against the reference's golden vectors and the oracle:
single-step get_actions / get_values, L-step masked sequences (rnn.py:30-77), the recurrent ppo_update with
per-parameter gradients, and R_MAPPO.train with recurrent_generator's chunking (T % L == 0 and != 0)."""
import numpy as np
import pytest
import torch

from conftest import golden, sub
from oracle import mappo_oracle as O
from test_gpu_e2e import M, make_args, load_policy, set_vn, fill_buffer, close, TUPLE, BUF_NAMES   # noqa: F401

pytestmark = pytest.mark.gpu


def close_rel_max(a, b, tol, msg=""):
    a = a.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)
    assert err <= tol, f"{msg}: max err / max|ref| = {err:.3e} > {tol}"


@pytest.mark.parametrize("case", [2, 3])
def test_gru_forward_golden(M, case):
    g = golden("forward")
    d = sub(g, f"c{case}")
    relu, rec, D, S, A, B, H = [int(x) for x in d["spec"]]
    assert rec
    a = make_args(M, use_ReLU=bool(relu), use_recurrent_policy=True)
    pol = M.R_MAPPOPolicy(a, [D], [S], M.Discrete(A))
    pol.actor.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sub(g, f"c{case}/actor").items()})
    pol.critic.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sub(g, f"c{case}/critic").items()})
    for tag in ("avail", "noavail"):
        av = d["avail"] if tag == "avail" else None
        v, act, lp, ra, rc = pol.get_actions(d["share_obs"], d["obs"], d["rnn_a"], d["rnn_c"], d["masks"], av, deterministic=True)
        np.testing.assert_array_equal(act.cpu().numpy(), d[f"{tag}/actions"])
        close(lp, d[f"{tag}/logp"], 1e-5, 2e-6); close(v, d[f"{tag}/values"], 1e-5, 2e-6)
        close(ra, d[f"{tag}/rnn_a"], 1e-5, 2e-6); close(rc, d[f"{tag}/rnn_c"], 1e-5, 2e-6)
        close(pol.get_values(d["share_obs"], d["rnn_c"], d["masks"]), d[f"{tag}/get_values"], 1e-5, 2e-6)
        ev, elp, eent = pol.evaluate_actions(d["share_obs"], d["obs"], d["rnn_a"], d["rnn_c"], d[f"{tag}/actions"].astype(np.float32),
                                             d["masks"], av, d[f"{tag}/active"])
        close(ev, d[f"{tag}/eval_values"], 1e-5, 2e-6); close(elp, d[f"{tag}/eval_logp"], 1e-5, 2e-6)
        close(eent, d[f"{tag}/eval_entropy"], 1e-5, 1e-6)
    # L-step chunk with zero masks inside (the reference's segment loop == per-step h*mask)
    ev, elp, eent = pol.evaluate_actions(d["seq/share_obs"], d["seq/obs"], d["seq/h0a"], d["seq/h0c"], d["seq/actions"], d["seq/masks"],
                                         d["seq/avail"], d["seq/active"])
    close(ev, d["seq/values"], 1e-5, 3e-6); close(elp, d["seq/logp"], 1e-5, 3e-6); close(eent, d["seq/entropy"], 1e-5, 1e-6)


def test_ppo_update_recurrent_golden(M):
    """golden ppo_update c9: recurrent_generator sample (L=10 chunks of T=20 x N=2 x M=3), ReLU, H=64."""
    g = golden("ppo_update")
    c = 9
    d = sub(g, f"c{c}")
    T, N, Ma, D, S, A, H = [int(x) for x in d["dims"]]
    fl = dict(zip([str(x) for x in d["flag_names"]], [bool(x) for x in d["flags"]]))
    assert fl["use_recurrent_policy"] and H == 64
    hy = d["hyper"]
    a = make_args(M, episode_length=T, n_rollout_threads=N, lr=float(hy[5]), critic_lr=float(hy[6]), use_recurrent_policy=True,
                  data_chunk_length=int(hy[9]))
    pol = load_policy(M, a, g, f"c{c}", D, S, A)
    tr = M.R_MAPPO(a, pol)
    set_vn(tr, d["vn0"])
    sample = tuple(d[f"sample/{nm}"] for nm in TUPLE)
    out = tr.ppo_update(sample)
    close(np.array(out, dtype=np.float64), d["r0/stats"], 2e-5, 1e-7, "stats")
    # raw (pre-clip == post-clip here: max_grad_norm 10) gradients, parameter by parameter
    for tag, net, seg in (("actor", pol.actor, 0), ("critic", pol.critic, 1)):
        lo = pol.seg_bounds[seg]
        for key, off, shape in net.layout:
            n = int(np.prod(shape))
            close_rel_max(pol.flat_grad[lo + off: lo + off + n].view(shape), d[f"r0/{tag}_grad/{key}"], 2e-4, f"{tag} grad {key}")
        ref_sd = sub(g, f"c{c}/r0/{tag}")
        for k, v in net.state_dict().items():
            close(v, ref_sd[k], 1e-5, 4e-6, f"{tag} {k}")
    close(tr.value_normalizer.state, d["r0/vn"], 2e-6, 1e-9)


def test_train_recurrent_golden(M):
    """golden train c1: R_MAPPO.train with recurrent_generator (T=20, L=10, num_mini_batch=1, 2 epochs, CPU permutation)."""
    g = golden("train")
    d = sub(g, "c1")
    T, N, Ma, D, S, A, H, nmb, rec, epochs, L = [int(x) for x in d["dims"]]
    assert rec
    a = make_args(M, episode_length=T, n_rollout_threads=N, lr=7e-4, critic_lr=7e-4, ppo_epoch=epochs, num_mini_batch=nmb,
                  use_recurrent_policy=True, data_chunk_length=L, perm_device="cpu")
    pol = load_policy(M, a, g, "c1", D, S, A)
    tr = M.R_MAPPO(a, pol)
    buf = M.SharedReplayBuffer(a, Ma, [D], [S], M.Discrete(A))
    fill_buffer(buf, d)
    torch.manual_seed(3000 + 1)
    info = tr.train(buf)
    ref = dict(zip([str(k) for k in d["info_keys"]], d["info"]))
    for k, v in info.items():
        close(v, ref[k], 1e-4, 1e-7, k)
    for tag, net in (("actor1", pol.actor), ("critic1", pol.critic)):
        ref_sd = sub(g, f"c1/{tag}")
        for k, v in net.state_dict().items():
            close(v, ref_sd[k], 1e-4, 6e-6, f"{tag} {k}")
    close(tr.value_normalizer.state, d["vn1"], 2e-6, 1e-9)


def test_recurrent_train_straddling_chunks_vs_oracle(M):
    """T % L != 0 (MPE rmappo: T=25, L=10): chunks straddle two series and the tail is dropped;
    same buffer + weights through the oracle's recurrent train."""
    T, N, Ma, D, A, L = 25, 4, 3, 18, 5, 10
    a = make_args(M, episode_length=T, n_rollout_threads=N, lr=7e-4, critic_lr=7e-4, ppo_epoch=2, num_mini_batch=2,
                  use_recurrent_policy=True, data_chunk_length=L, perm_device="cpu")
    torch.manual_seed(5)
    pol = M.R_MAPPOPolicy(a, [D], [D * Ma], M.Discrete(A))
    tr = M.R_MAPPO(a, pol)
    buf = M.SharedReplayBuffer(a, Ma, [D], [D * Ma], M.Discrete(A))
    rng = np.random.default_rng(3)
    f = np.float32
    for n in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "rewards"):
        arr = getattr(buf, n); arr.copy_(torch.from_numpy(rng.standard_normal(tuple(arr.shape)).astype(f)))
    buf.value_preds.copy_(torch.from_numpy((rng.standard_normal(tuple(buf.value_preds.shape)) * 0.3).astype(f)))
    buf.returns.copy_(torch.from_numpy((rng.standard_normal(tuple(buf.returns.shape)) * 2).astype(f)))
    buf.actions.copy_(torch.from_numpy(rng.integers(0, A, tuple(buf.actions.shape)).astype(f)))
    buf.action_log_probs.copy_(torch.from_numpy((-np.abs(rng.standard_normal(tuple(buf.actions.shape))) - 1).astype(f)))
    buf.masks.copy_(torch.from_numpy((rng.random(tuple(buf.masks.shape)) > 0.15).astype(f)))
    buf.active_masks.copy_(torch.from_numpy((rng.random(tuple(buf.masks.shape)) > 0.2).astype(f)))
    oa = O.default_args(episode_length=T, n_rollout_threads=N, lr=7e-4, critic_lr=7e-4, ppo_epoch=2, num_mini_batch=2,
                        use_recurrent_policy=True, data_chunk_length=L)
    opol = O.PolicyRef(oa, D, D * Ma, A)
    opol.actor.load_state_dict({k: v.cpu() for k, v in pol.actor.state_dict().items()})
    opol.critic.load_state_dict({k: v.cpu() for k, v in pol.critic.state_dict().items()})
    ob = O.BufferRef(oa, Ma, D, D * Ma, A)
    for n in BUF_NAMES:
        getattr(ob, n)[...] = getattr(buf, n).cpu().numpy()
    ovn = O.ValueNormRef()
    torch.manual_seed(11)
    perms = [torch.randperm((T * N * Ma) // L).numpy() for _ in range(2)]
    oinfo = O.train_ref(oa, opol, ovn, ob, perms=perms)
    torch.manual_seed(11)
    info = tr.train(buf)
    for k in oinfo:
        close(info[k], oinfo[k], 1e-4, 1e-6, k)
    for k, v in pol.actor.state_dict().items():
        close(v, opol.actor.state_dict()[k].numpy(), 1e-4, 6e-6, k)
    for k, v in pol.critic.state_dict().items():
        close(v, opol.critic.state_dict()[k].numpy(), 1e-4, 6e-6, k)


@pytest.mark.parametrize("runner_kind", ["mpe", "smac"])
def test_recurrent_runner_iteration_vs_oracle(M, runner_kind):
    """rmappo end to end on the GPU: rollout with the GRU act/value kernels through MPERunner / SMACRunner (availability
    masks, agent deaths, env terminations), then bootstrap + GAE + recurrent train against the oracle on the same
    buffer and weights.  Rollout outputs are re-derived step by step by the oracle from the stored rnn states."""
    from mappo_amd.runner.shared.smac_runner import SMACRunner
    from mappo_amd.envs.synthetic import SyntheticSMACEnv
    T, N, Ma, L = 20, 6, 3, 10
    if runner_kind == "mpe":
        D, S, A = 18, 54, 5
        env = M.SyntheticMPEEnv(N, Ma, D, A, T, seed=2)
        R = M.MPERunner
    else:
        D, S, A = 30, 48, 9
        env = SyntheticSMACEnv(N, Ma, D, S, A, p_death=0.05, p_term=0.1, seed=2)
        R = SMACRunner
    a = make_args(M, episode_length=T, n_rollout_threads=N, ppo_epoch=2, num_mini_batch=1, lr=7e-4, critic_lr=7e-4, seed=1,
                  env_name="MPE", use_recurrent_policy=True, algorithm_name="rmappo", data_chunk_length=L, perm_device="cpu",
                  use_hip_graph=False)
    torch.manual_seed(1)
    runner = R(dict(all_args=a, envs=env, eval_envs=None, num_agents=Ma, device=torch.device("cuda"), run_dir=None))
    oa = O.default_args(episode_length=T, n_rollout_threads=N, ppo_epoch=2, lr=7e-4, critic_lr=7e-4, use_recurrent_policy=True,
                        data_chunk_length=L)
    opol = O.PolicyRef(oa, D, S, A)
    opol.actor.load_state_dict({k: v.cpu() for k, v in runner.policy.actor.state_dict().items()})
    opol.critic.load_state_dict({k: v.cpu() for k, v in runner.policy.critic.state_dict().items()})
    runner.warmup()
    for step in range(T):
        out = runner.collect(step)
        if runner_kind == "mpe":
            values, actions, logp, rs, rc, actions_env = out
            obs, rewards, dones, infos = env.step(actions_env)
            runner.insert((obs, rewards, dones, infos, values, actions, logp, rs, rc))
        else:
            values, actions, logp, rs, rc = out
            obs, share_obs, rewards, dones, infos, avail = env.step(actions)
            runner.insert((obs, share_obs, rewards, dones, infos, avail, values, actions, logp, rs, rc))
    b = runner.buffer
    Rr = N * Ma
    t_ = lambda x: x.cpu()
    with torch.no_grad():
        for step in (0, 7, T - 1):
            av = t_(b.available_actions[step]).reshape(Rr, A) if runner_kind == "smac" else None
            feats, ra = opol.actor.features(t_(b.obs[step]).reshape(Rr, D), t_(b.rnn_states[step]).reshape(Rr, 1, 64), t_(b.masks[step]).reshape(Rr, 1))
            z = opol.actor.act.logits(feats, av)
            lp, _, _ = opol.actor.act.logp_entropy(z, t_(b.actions[step]).reshape(Rr, 1))
            v, rcr = opol.critic(t_(b.share_obs[step]).reshape(Rr, S), t_(b.rnn_states_critic[step]).reshape(Rr, 1, 64), t_(b.masks[step]).reshape(Rr, 1))
            close(b.action_log_probs[step].reshape(Rr, 1), lp.numpy(), 1e-5, 3e-6, f"logp step {step}")
            close(b.value_preds[step].reshape(Rr, 1), v.numpy(), 1e-5, 3e-6, f"value step {step}")
            keep = t_(b.masks[step + 1]).reshape(Rr, 1, 1)                     # insert zeroes the states of finished envs
            close(b.rnn_states[step + 1].reshape(Rr, 1, 64), (ra * keep).numpy(), 1e-5, 3e-6, f"rnn_a step {step}")
            close(b.rnn_states_critic[step + 1].reshape(Rr, 1, 64), (rcr * keep).numpy(), 1e-5, 3e-6, f"rnn_c step {step}")
    if runner_kind == "smac":
        assert float(b.active_masks.min()) == 0.0 and float(b.masks.min()) == 0.0       # deaths and terminations occurred
        picked = torch.gather(b.available_actions[:T], -1, b.actions.long())
        assert float(picked.min()) == 1.0                                              # only available actions were taken
    ob = O.BufferRef(oa, Ma, D, S, A)
    for n in BUF_NAMES:
        if n != "returns":
            getattr(ob, n)[...] = getattr(b, n).cpu().numpy()
    ovn = O.ValueNormRef()
    with torch.no_grad():
        nv, _ = opol.critic(torch.from_numpy(np.concatenate(ob.share_obs[-1])), torch.from_numpy(np.concatenate(ob.rnn_states_critic[-1])),
                            torch.from_numpy(np.concatenate(ob.masks[-1])))
    ob.compute_returns(np.array(np.split(nv.numpy(), N)), ovn)
    runner.compute()
    close(b.returns[:T], ob.returns[:T], 1e-5, 3e-6)
    torch.manual_seed(21)
    perms = [torch.randperm((T * N * Ma) // L).numpy() for _ in range(2)]
    oinfo = O.train_ref(oa, opol, ovn, ob, perms=perms)
    torch.manual_seed(21)
    info = runner.train()
    for k in oinfo:
        close(info[k], oinfo[k], 1e-4, 1e-6, k)
    for k, vv in runner.policy.actor.state_dict().items():
        close(vv, opol.actor.state_dict()[k].numpy(), 1e-4, 6e-6, k)
    for k, vv in runner.policy.critic.state_dict().items():
        close(vv, opol.critic.state_dict()[k].numpy(), 1e-4, 6e-6, k)


@pytest.mark.parametrize("rec", [False, True])
def test_wide_observation_train_vs_oracle(M, rec):
    """SMAC MMM2 shapes (BASELINE configs[3]: 10 agents, obs 176, share_obs 322, 18 actions), MLP and GRU policies:
    R_MAPPO.train through the K-chunked layer-1 kernels + wide_l1_backward against the oracle."""
    T, N, Ma, D, S, A, L = 10, 3, 10, 176, 322, 18, 10
    a = make_args(M, episode_length=T, n_rollout_threads=N, lr=7e-4, critic_lr=7e-4, ppo_epoch=2, num_mini_batch=1,
                  use_recurrent_policy=rec, data_chunk_length=L, perm_device="cpu", gain=1.0)
    torch.manual_seed(9)
    pol = M.R_MAPPOPolicy(a, [D], [S], M.Discrete(A))
    tr = M.R_MAPPO(a, pol)
    buf = M.SharedReplayBuffer(a, Ma, [D], [S], M.Discrete(A))
    rng = np.random.default_rng(4)
    f = np.float32
    for n in ("share_obs", "obs", "rnn_states", "rnn_states_critic"):
        arr = getattr(buf, n); arr.copy_(torch.from_numpy(rng.standard_normal(tuple(arr.shape)).astype(f)))
    buf.value_preds.copy_(torch.from_numpy((rng.standard_normal(tuple(buf.value_preds.shape)) * 0.3).astype(f)))
    buf.returns.copy_(torch.from_numpy((rng.standard_normal(tuple(buf.returns.shape)) * 2).astype(f)))
    buf.actions.copy_(torch.from_numpy(rng.integers(0, A, tuple(buf.actions.shape)).astype(f)))
    buf.action_log_probs.copy_(torch.from_numpy((-np.abs(rng.standard_normal(tuple(buf.actions.shape))) - 2).astype(f)))
    buf.masks.copy_(torch.from_numpy((rng.random(tuple(buf.masks.shape)) > 0.15).astype(f)))
    buf.active_masks.copy_(torch.from_numpy((rng.random(tuple(buf.masks.shape)) > 0.2).astype(f)))
    av = (rng.random(tuple(buf.available_actions.shape)) > 0.3).astype(f)
    np.put_along_axis(av[:T], buf.actions.cpu().numpy().astype(np.int64), 1.0, axis=-1)
    buf.available_actions.copy_(torch.from_numpy(av))
    oa = O.default_args(episode_length=T, n_rollout_threads=N, lr=7e-4, critic_lr=7e-4, ppo_epoch=2, num_mini_batch=1,
                        use_recurrent_policy=rec, data_chunk_length=L, gain=1.0)
    opol = O.PolicyRef(oa, D, S, A)
    opol.actor.load_state_dict({k: v.cpu() for k, v in pol.actor.state_dict().items()})
    opol.critic.load_state_dict({k: v.cpu() for k, v in pol.critic.state_dict().items()})
    ob = O.BufferRef(oa, Ma, D, S, A)
    for n in BUF_NAMES:
        getattr(ob, n)[...] = getattr(buf, n).cpu().numpy()
    ovn = O.ValueNormRef()
    torch.manual_seed(13)
    n_perm = (T * N * Ma) // L if rec else T * N * Ma
    perms = [torch.randperm(n_perm).numpy() for _ in range(2)]
    oinfo = O.train_ref(oa, opol, ovn, ob, perms=perms)
    torch.manual_seed(13)
    info = tr.train(buf)
    for k in oinfo:
        close(info[k], oinfo[k], 1e-4, 1e-6, k)
    for k, v in pol.actor.state_dict().items():
        close(v, opol.actor.state_dict()[k].numpy(), 1e-4, 6e-6, k)
    for k, v in pol.critic.state_dict().items():
        close(v, opol.critic.state_dict()[k].numpy(), 1e-4, 6e-6, k)


@pytest.mark.parametrize("runner_kind", ["mpe", "smac"])
def test_recurrent_graph_replay_matches_eager(M, runner_kind):
    """rmappo iterations replayed as hipGraphs (episode graph + train graph with the chunk permutation drawn on the device)
    == the same iterations launched eagerly: same seeds, same device random streams, identical parameters afterwards."""
    from mappo_amd.runner.shared.smac_runner import SMACRunner
    from mappo_amd.envs.synthetic import SyntheticSMACEnv
    T, N, Ma, L = 20, 6, 3, 10

    def run(use_graph):
        torch.manual_seed(5)
        if runner_kind == "mpe":
            env, R, D, S, A = M.SyntheticMPEEnv(N, Ma, 18, 5, T, seed=2), M.MPERunner, 18, 54, 5
        else:
            env, R = SyntheticSMACEnv(N, Ma, 30, 48, 9, p_death=0.05, p_term=0.1, seed=2), SMACRunner
        a = make_args(M, episode_length=T, n_rollout_threads=N, ppo_epoch=2, num_mini_batch=1, lr=7e-4, critic_lr=7e-4, seed=1,
                      env_name="MPE", use_recurrent_policy=True, algorithm_name="rmappo", data_chunk_length=L, use_hip_graph=use_graph)
        runner = R(dict(all_args=a, envs=env, eval_envs=None, num_agents=Ma, device=torch.device("cuda"), run_dir=None))
        runner.warmup()
        infos = [runner.run_episode()[0] for _ in range(4)]          # eager, capture + replay, replay, replay
        torch.cuda.synchronize()
        if use_graph:
            assert isinstance(runner._rollout_graph, torch.cuda.CUDAGraph)
            assert any(isinstance(g, torch.cuda.CUDAGraph) for g in runner.trainer._graphs.values())
        return runner.policy.flat_params.clone(), infos

    p_eager, i_eager = run(False)
    p_graph, i_graph = run(True)
    np.testing.assert_array_equal(p_graph.cpu().numpy(), p_eager.cpu().numpy())
    for a, b in zip(i_graph, i_eager):
        for k in a:
            assert a[k] == b[k], k


@pytest.mark.parametrize("recurrent", [True, False])
def test_insert_smac_fused_matches_slot_writes(M, recurrent):
    """mappo_insert_smac (one launch) == smac_runner.py:129-151's mask arithmetic + the slot writes of
    SharedReplayBuffer.insert (shared_buffer.py:74-112), bit for bit, incl. env terminations, agent deaths and bad transitions."""
    T, N, Ma, D, S, A, H = 4, 29, 3, 30, 48, 9, 64
    a = make_args(M, episode_length=T, n_rollout_threads=N, use_recurrent_policy=recurrent, algorithm_name="rmappo" if recurrent else "mappo")
    b1, b2 = (M.SharedReplayBuffer(a, Ma, [D], [S], M.Discrete(A)) for _ in range(2))
    g = torch.Generator(device="cuda").manual_seed(3)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    for step in range(3):
        obs, share, avail = rnd(N, Ma, D), rnd(N, Ma, S), (torch.rand(N, Ma, A, device="cuda", generator=g) < 0.7).float()
        rew = rnd(N, 1, 1).expand(N, Ma, 1)
        dones = torch.rand(N, Ma, device="cuda", generator=g) < 0.4
        dones[::5] = True                                                      # terminated envs
        bad = torch.rand(N, Ma, device="cuda", generator=g) < 0.2
        ha, hc = rnd(N * Ma, 1, H), rnd(N * Ma, 1, H)
        assert b1.insert_smac_fused(share, obs, rew, dones, bad, avail, *((ha, hc) if recurrent else ()))
        dones_env = dones.all(dim=1)
        keep_env = (~dones_env).float()
        masks = keep_env.view(N, 1, 1).expand(N, Ma, 1)
        active = torch.where(dones_env.view(N, 1), torch.ones((), device="cuda"), (~dones).float()).view(N, Ma, 1)
        k = keep_env.view(N, 1, 1, 1)
        b2.insert_env(share, obs, rew, masks, ha.view(N, Ma, 1, H) * k if recurrent else None, hc.view(N, Ma, 1, H) * k if recurrent else None,
                      (~bad).float().view(N, Ma, 1), active, avail)
    assert b1.step == b2.step
    for name in ("share_obs", "obs", "rewards", "masks", "bad_masks", "active_masks", "available_actions", "rnn_states", "rnn_states_critic"):
        np.testing.assert_array_equal(getattr(b1, name).cpu().numpy(), getattr(b2, name).cpu().numpy(), err_msg=name)


def test_recurrent_rows_epochs_are_permutations(M):
    """recurrent_rows_epochs: every epoch's minibatches together visit each chunk exactly once (a permutation of the chunks
    per epoch, different between epochs), and the rows of a chunk are L consecutive positions of the (n, m, t) order."""
    T, N, Ma, L, nmb, E = 25, 8, 3, 10, 2, 4
    a = make_args(M, episode_length=T, n_rollout_threads=N, use_recurrent_policy=True, algorithm_name="rmappo", data_chunk_length=L,
                  num_mini_batch=nmb)
    b = M.SharedReplayBuffer(a, Ma, [18], [54], M.Discrete(5))
    R = N * Ma
    chunks = (T * R) // L
    out = b.recurrent_rows_epochs(E, nmb, L)
    assert len(out) == E and all(len(o) == nmb for o in out)
    firsts = []
    for ep in out:
        h0 = torch.cat([h for _, h in ep]).cpu().numpy().astype(np.int64)
        q0 = (h0 % R) * T + h0 // R                                  # buffer row t*R + r  ->  flat position r*T + t
        assert sorted(q0.tolist()) == [c * L for c in range(chunks)]             # chunks % nmb == 0 here: all of them, once
        firsts.append(q0[:16].tolist())
        for rows, h in ep:
            r = rows.cpu().numpy().astype(np.int64).reshape(L, -1)
            q = (r % R) * T + r // R
            np.testing.assert_array_equal(q, q[0][None, :] + np.arange(L)[:, None])
            np.testing.assert_array_equal(r[0], h.cpu().numpy())
    assert len({tuple(f) for f in firsts}) == E                      # epochs differ


def test_gru16_blocked_and_feature_major_paths_agree(M):
    """The 16-sequence-tile training kernels (csrc/gru_train16.hip) take the trunk features either BLOCKED per (t, 16 sequences)
    tile (narrow inputs: mappo_mlp_features_seq, d x left blocked for mappo_trunk_backward_seq) or feature-major [64][B] (wide
    inputs: mappo_mlp_features, d x to dxT for mappo_trunk_backward).  Both forms, on the same critic and rows (Nc = 37: a ragged
    last tile; gathered rows), must give the same loss partials, head / rnn.norm / GRU gradient columns and d x."""
    from mappo_amd import ops
    T, N, Ma, D, S, A, L = 20, 37, 1, 30, 48, 9, 10
    a = make_args(M, episode_length=T, n_rollout_threads=N, use_recurrent_policy=True, data_chunk_length=L)
    torch.manual_seed(2)
    pol = M.R_MAPPOPolicy(a, [D], [S], M.Discrete(A))
    tr = M.R_MAPPO(a, pol)
    net = pol.critic
    H = 64
    Nc = 37
    B = L * Nc
    g = torch.Generator(device="cuda").manual_seed(6)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    n_buf = B + 50
    x = rnd(n_buf, S)
    rows = torch.randperm(n_buf, device="cuda", generator=g)[:B].to(torch.int32).contiguous()
    h0, h0_rows = rnd(n_buf, H), torch.randperm(n_buf, device="cuda", generator=g)[:Nc].to(torch.int32).contiguous()
    masks = (torch.rand(n_buf, device="cuda", generator=g) > 0.2).float()
    active = (torch.rand(n_buf, device="cuda", generator=g) > 0.2).float()
    v_old, ret = rnd(n_buf) * 0.3, rnd(n_buf) * 2
    vn = torch.tensor([0.1, 1.3, 1.0], device="cuda")
    mom = torch.zeros(4, dtype=torch.float64, device="cuda")
    ops.minibatch_moments(ret, active, rows, B, mom)
    P = pol.n_flat
    n_sl = max(ops.gru16_slabs(L, Nc), ops.mlp_backward_slabs(B))
    col0 = pol.seg_bounds[1]
    outs = []
    for blocked in (True, False):
        scratch = torch.zeros(ops.gru16_scratch_floats(L, Nc), device="cuda")
        comp = ops.gru16_blocked_floats(L, Nc)
        slabs = torch.zeros(n_sl, P, device="cuda")
        part = torch.zeros(1024, dtype=torch.float64, device="cuda")
        if blocked:
            feat = torch.empty(comp, device="cuda")
            ops.mlp_features_seq(net.flat, net.desc, x, rows, L, Nc, feat)
            dxT = None
        else:
            feat = torch.empty(H, B, device="cuda")
            ops.mlp_features(net.flat, net.desc, x, rows, B, feat)
            dxT = torch.zeros(H, B, device="cuda")
        ops.gru16_forward_loss(net.flat, net.desc, feat, blocked, h0, h0_rows, masks, rows, L, Nc, 2, None, None, None, None, active, v_old,
                               ret, vn, mom, tr._cfg, scratch, slabs, P, col0, part)
        ops.gru16_backward(net.flat, net.desc, masks, rows, L, Nc, scratch, dxT)
        ops.gru16_wgrad(net.desc, feat, blocked, scratch, L, Nc, slabs, P, col0)
        if blocked:
            ops.trunk_backward_seq(net.flat, net.desc, x, rows, L, Nc, scratch[5 * comp:6 * comp], slabs, P, col0)
            # blocked d x -> feature-major for the comparison: [L][n_ct][4 b][16 q... lane][4 i] -> feature 16 b + 4 q + i, sequence 16 j + n
            n_ct = (Nc + 15) // 16
            d = scratch[5 * comp:6 * comp].view(L, n_ct, 4, 4, 16, 4)            # t, j, b, q, n, i
            dxT = d.permute(2, 3, 5, 0, 1, 4).reshape(H, L, n_ct * 16)[:, :, :Nc].reshape(H, B)
        else:
            ops.trunk_backward(net.flat, net.desc, x, rows, B, dxT, slabs, P, col0)
        outs.append((dxT.cpu().numpy(), slabs.double().sum(0).cpu().numpy(), part.cpu().numpy()))
    (dx_b, g_b, p_b), (dx_f, g_f, p_f) = outs
    np.testing.assert_allclose(p_b, p_f, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(dx_b, dx_f, rtol=0, atol=2e-6 * np.abs(dx_f).max())
    np.testing.assert_allclose(g_b, g_f, rtol=0, atol=1e-5 * np.abs(g_f).max())
    assert np.abs(g_f).max() > 0 and np.abs(dx_f).max() > 0


@pytest.mark.parametrize("Do,Ds,A,R", [(30, 48, 9, 3 * 37 + 5), (176, 322, 18, 640), (130, 70, 5, 37)])
def test_recurrent_step_dual_matches_separate_kernels(M, Do, Ds, A, R):
    """The one-launch rollout step of both networks (mappo_recurrent_step_dual: trunks + GRU step + heads; narrow inputs with the
    trunk in registers, wide inputs with the split-K trunk handing over through LDS) == the per-network launches
    (mappo_mlp_features + mappo_gru_forward head modes 2 / 1): same actions, and log-probs / values / next states equal to fp32
    rounding (the fused kernels contract a few multiply-adds differently: 1-3 ulp, measured 7e-7 absolute at most).
    Narrow inputs (SMAC 3m shapes) and wide ones (MMM2 shapes; widths that are not multiples of 4)."""
    from mappo_amd import recurrent
    a = make_args(M, use_recurrent_policy=True, algorithm_name="rmappo")
    pol = M.R_MAPPOPolicy(a, [Do], [Ds], M.Discrete(A))
    assert recurrent.can_step_dual(pol.actor, pol.critic)
    H = 64
    g = torch.Generator(device="cuda").manual_seed(4)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    obs, cent, ha, hc = rnd(R, Do), rnd(R, Ds), rnd(R, 1, H), rnd(R, 1, H)
    masks = (torch.rand(R, 1, device="cuda", generator=g) > 0.3).float()
    avail = (torch.rand(R, A, device="cuda", generator=g) < 0.7).float()
    avail[:, 0] = 1.0
    act1, lp1, v1 = torch.empty(R, device="cuda"), torch.empty(R, device="cuda"), torch.empty(R, 1, device="cuda")
    na1 = recurrent.actor_step(pol.actor, obs, ha, masks, avail, False, act1, lp1, counter=7)
    nc1 = recurrent.critic_forward(pol.critic, cent, hc, masks, v1)
    act2, lp2, v2 = torch.empty(R, device="cuda"), torch.empty(R, device="cuda"), torch.empty(R, device="cuda")
    na2, nc2 = recurrent.step_dual(pol.actor, pol.critic, obs, cent, ha, hc, masks, avail, False, act2, lp2, v2, 7)
    np.testing.assert_array_equal(act2.cpu().numpy(), act1.cpu().numpy())
    for x, y in ((lp1, lp2), (v1.view(R), v2), (na1, na2), (nc1, nc2)):
        np.testing.assert_allclose(y.cpu().numpy(), x.cpu().numpy(), rtol=2e-6, atol=2e-6)
    picked = torch.gather(avail, 1, act2.long().view(R, 1))
    assert float(picked.min()) == 1.0


@pytest.mark.parametrize("recurrent", [False, True])
def test_update_actor_toggle_across_graph_replays(M, recurrent):
    """R_MAPPO.train(update_actor=...) replayed from hipGraphs keyed by the flag: the host-side switch (Adam skips the actor,
    r_mappo.py:143-148 with grad None) is applied eagerly before every launch sequence, so a replay never inherits the mode of
    the previous call: the actor's parameters move exactly in the update_actor=True calls, the critic's in all of them."""
    T, N, Ma = 10, 8, 3
    a = make_args(M, episode_length=T, n_rollout_threads=N, ppo_epoch=2, num_mini_batch=1, lr=7e-4, critic_lr=7e-4,
                  use_recurrent_policy=recurrent, algorithm_name="rmappo" if recurrent else "mappo", data_chunk_length=5)
    env = M.SyntheticMPEEnv(N, Ma, 18, 5, T, seed=3)
    runner = M.MPERunner(dict(all_args=a, envs=env, eval_envs=None, num_agents=Ma, device=torch.device("cuda"), run_dir=None))
    runner.warmup()
    pol, tr, buf = runner.policy, runner.trainer, runner.buffer
    lo, mid, hi = pol.seg_bounds
    runner.rollout()
    torch.cuda.synchronize()
    for flag in (True, True, True, False, False, False, True, False, True):
        before = pol.flat_params.clone()
        tr.train(buf, update_actor=flag)
        torch.cuda.synchronize()
        moved_a = bool((pol.flat_params[lo:mid] != before[lo:mid]).any())
        moved_c = bool((pol.flat_params[mid:hi] != before[mid:hi]).any())
        assert moved_a == flag and moved_c, (flag, moved_a, moved_c)
    assert sum(isinstance(g, torch.cuda.CUDAGraph) for g in tr._graphs.values()) == 2


def test_smac_runner_eval_loop(M):
    """SMACRunner.eval (smac_runner.py:160-214): deterministic act with availability masks until `eval_episodes` env
    terminations; the win rate it reports equals the env's own count, chosen actions are available ones."""
    from mappo_amd.runner.shared.smac_runner import SMACRunner
    from mappo_amd.envs.synthetic import SyntheticSMACEnv
    N, Ma, D, S, A = 5, 3, 30, 48, 9

    class HostInfoEnv(SyntheticSMACEnv):
        """the synthetic env with SMAC's host-side infos: [[{'won': ...}] * agents] per env"""
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.ended, self.won, self.picked_ok, self._avail = 0, 0, True, None

        def reset(self):
            out = super().reset()
            self._avail = out[2]
            return out

        def step(self, actions):
            act = torch.as_tensor(actions).to(self.device).long().view(self.N, self.M, 1)
            self.picked_ok &= bool(torch.gather(self._avail, -1, act).min() == 1.0)
            obs, share, rew, dones, bad, avail = super().step(actions)
            self._avail = avail
            done_env = dones.all(dim=1).cpu().numpy()
            infos = []
            for i in range(self.N):
                won = bool(done_env[i]) and (self.ended + i) % 2 == 0
                infos.append([{"won": won, "bad_transition": False} for _ in range(self.M)])
                if done_env[i]:
                    self.won += int(won)
            self.ended += int(done_env.sum())
            return obs, share, rew, dones, infos, avail

    a = make_args(M, episode_length=8, n_rollout_threads=N, n_eval_rollout_threads=N, use_recurrent_policy=True, algorithm_name="rmappo",
                  env_name="StarCraft2", use_eval=True, eval_episodes=7)
    env = SyntheticSMACEnv(N, Ma, D, S, A, seed=2)
    eval_env = HostInfoEnv(N, Ma, D, S, A, p_death=0.05, p_term=0.15, seed=3)
    runner = SMACRunner(dict(all_args=a, envs=env, eval_envs=eval_env, num_agents=Ma, device=torch.device("cuda"), run_dir=None))
    rate = runner.eval(0)
    assert eval_env.ended >= 7 and eval_env.picked_ok
    assert rate == eval_env.won / eval_env.ended


def test_config3_full_size_train_vs_oracle_on_active_subset(M):
    """BASELINE configs[2] at FULL size (T=400, N=256 rollout threads, 3 agents, obs 30 / state 48 / 9 actions, GRU, chunks of 10:
    307 200 rows, 30 720 chunks per update).  The oracle cannot run that on the CPU in test time, so the buffer is built with
    active_masks == 0 everywhere except four rollout threads: with the active-mask denominators (r_mappo.py:84,130-134; the
    nan-masked advantage statistics :174-182) and ValueNorm off, inactive rows contribute exactly nothing, and R_MAPPO.train on
    the full buffer must equal the oracle's train on the 4-thread sub-buffer — losses, entropy, gradient norms, updated weights
    (`ratio` is an unmasked mean: not compared).  Every chunk still runs through the full-size kernels."""
    T, N, Ma, D, S, A, L = 400, 256, 3, 30, 48, 9, 10
    keep = [5, 77, 130, 201]
    common = dict(episode_length=T, lr=5e-4, critic_lr=5e-4, ppo_epoch=2, num_mini_batch=1, use_recurrent_policy=True,
                  data_chunk_length=L, use_valuenorm=False)
    a = make_args(M, n_rollout_threads=N, perm_device="cpu", algorithm_name="rmappo", **common)
    torch.manual_seed(13)
    pol = M.R_MAPPOPolicy(a, [D], [S], M.Discrete(A))
    tr = M.R_MAPPO(a, pol)
    buf = M.SharedReplayBuffer(a, Ma, [D], [S], M.Discrete(A))
    g = torch.Generator(device="cuda").manual_seed(17)
    rnd = lambda shape: torch.randn(tuple(shape), device="cuda", generator=g)
    for n in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "rewards"):
        getattr(buf, n).copy_(rnd(getattr(buf, n).shape))
    buf.value_preds.copy_(rnd(buf.value_preds.shape) * 0.3)
    buf.returns.copy_(rnd(buf.returns.shape) * 2)
    buf.actions.copy_(torch.randint(0, A, tuple(buf.actions.shape), device="cuda", generator=g).float())
    buf.action_log_probs.copy_(-rnd(buf.actions.shape).abs() - 1)
    buf.masks.copy_((torch.rand(tuple(buf.masks.shape), device="cuda", generator=g) > 0.05).float())
    buf.available_actions.fill_(1.0)                            # (all actions available: the stored actions were drawn uniformly)
    act = torch.zeros(tuple(buf.active_masks.shape), device="cuda")
    act[:, keep] = (torch.rand(act[:, keep].shape, device="cuda", generator=g) > 0.2).float()
    buf.active_masks.copy_(act)
    oa = O.default_args(n_rollout_threads=len(keep), **common)
    opol = O.PolicyRef(oa, D, S, A)
    opol.actor.load_state_dict({k: v.cpu() for k, v in pol.actor.state_dict().items()})
    opol.critic.load_state_dict({k: v.cpu() for k, v in pol.critic.state_dict().items()})
    ob = O.BufferRef(oa, Ma, D, S, A)
    for n in BUF_NAMES:
        getattr(ob, n)[...] = getattr(buf, n)[:, keep].cpu().numpy()
    torch.manual_seed(3)
    perms = [torch.randperm((T * len(keep) * Ma) // L).numpy() for _ in range(2)]
    oinfo = O.train_ref(oa, opol, None, ob, perms=perms)
    torch.manual_seed(5)
    info = tr.train(buf)
    for k in oinfo:
        if k != "ratio":
            close(info[k], oinfo[k], 2e-4, 1e-6, k)
    for net, onet in ((pol.actor, opol.actor), (pol.critic, opol.critic)):
        for k, v in net.state_dict().items():
            close(v, onet.state_dict()[k].numpy(), 2e-4, 1e-5, k)


def _random_fill(buf, A, g, active=None):
    """Device-side random contents for a full-size buffer (the fill is not what is tested)."""
    rnd = lambda shape: torch.randn(tuple(shape), device="cuda", generator=g)
    for n in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "rewards"):
        getattr(buf, n).copy_(rnd(getattr(buf, n).shape))
    buf.value_preds.copy_(rnd(buf.value_preds.shape) * 0.3)
    buf.returns.copy_(rnd(buf.returns.shape) * 2)
    buf.actions.copy_(torch.randint(0, A, tuple(buf.actions.shape), device="cuda", generator=g).float())
    buf.action_log_probs.copy_(-rnd(buf.actions.shape).abs() - 1)
    buf.masks.copy_((torch.rand(tuple(buf.masks.shape), device="cuda", generator=g) > 0.05).float())
    buf.available_actions.fill_(1.0)                            # (all actions available: the stored actions were drawn uniformly)
    if active is None:
        active = (torch.rand(tuple(buf.active_masks.shape), device="cuda", generator=g) > 0.2).float()
    buf.active_masks.copy_(active)


def test_config5_shape_train_vs_oracle_multi_tile(M):
    """BASELINE configs[4] SHAPE (64 agents, obs = state = 512, MLP policy) at T=40, N=20: 51 200 rows = 3 200 sixteen-row tiles,
    more than the 2 048 waves of a launch, so R_MAPPO.train runs the wide-input update path in its steady state — the in-place
    register refill of wide_l1_fwd16_kernel, the multi-tile loops of mlp_update16x_kernel and wide_l1_bwd16_kernel — and is
    compared with the oracle's train on the SAME buffer (mlp.py:18-55, r_mappo.py:91-219): losses, gradient norms, weights."""
    T, N, Ma, D, S, A = 40, 20, 64, 512, 512, 5
    common = dict(episode_length=T, n_rollout_threads=N, lr=5e-4, critic_lr=5e-4, ppo_epoch=2, num_mini_batch=1)
    a = make_args(M, perm_device="cpu", **common)
    torch.manual_seed(21)
    pol = M.R_MAPPOPolicy(a, [D], [S], M.Discrete(A))
    tr = M.R_MAPPO(a, pol)
    buf = M.SharedReplayBuffer(a, Ma, [D], [S], M.Discrete(A))
    _random_fill(buf, A, torch.Generator(device="cuda").manual_seed(23))
    oa = O.default_args(**common)
    opol = O.PolicyRef(oa, D, S, A)
    opol.actor.load_state_dict({k: v.cpu() for k, v in pol.actor.state_dict().items()})
    opol.critic.load_state_dict({k: v.cpu() for k, v in pol.critic.state_dict().items()})
    ob = O.BufferRef(oa, Ma, D, S, A)
    for n in BUF_NAMES:
        getattr(ob, n)[...] = getattr(buf, n).cpu().numpy()
    ovn = O.ValueNormRef()
    perms = [np.arange(T * N * Ma) for _ in range(2)]           # num_mini_batch == 1: the permutation only reorders sums
    oinfo = O.train_ref(oa, opol, ovn, ob, perms=perms)
    info = tr.train(buf)
    for k in oinfo:
        close(info[k], oinfo[k], 1e-4, 1e-6, k)
    for net, onet in ((pol.actor, opol.actor), (pol.critic, opol.critic)):
        for k, v in net.state_dict().items():
            close(v, onet.state_dict()[k].numpy(), 1e-4, 6e-6, k)
    close(tr.value_normalizer.state, ovn.state(), 1e-5, 1e-7, "ValueNorm state")


def test_config4_per_gpu_size_train_vs_oracle_on_active_subset(M):
    """BASELINE configs[3] at its PER-GPU size of an 8-GPU run (T=400, N=64 rollout threads, 10 agents, obs 176 / state 322 /
    18 actions, GRU, chunks of 10, num_mini_batch=2: 256 000 rows, 12 800 chunks per minibatch): wide-input trunk features,
    GRU training kernels and the wide trunk backward at the size where their tile loops iterate.  Same device as
    test_config3_full_size_train_vs_oracle_on_active_subset: active_masks is zero except on four rollout threads, ValueNorm is
    off, so train() on the full buffer must equal the oracle's train on the 4-thread sub-buffer.  With two minibatches the
    chunk permutation matters: the full run's permutations put the kept threads' chunks into the SAME minibatch the oracle's
    permutation of the sub-buffer puts them in (shared_buffer.py:385-494), padded with inactive chunks in random order."""
    T, N, Ma, D, S, A, L, nmb, E = 400, 64, 10, 176, 322, 18, 10, 2, 2
    keep = [3, 17, 40, 63]
    common = dict(episode_length=T, lr=5e-4, critic_lr=5e-4, ppo_epoch=E, num_mini_batch=nmb, use_recurrent_policy=True,
                  data_chunk_length=L, use_valuenorm=False)
    a = make_args(M, n_rollout_threads=N, perm_device="cpu", algorithm_name="rmappo", **common)
    torch.manual_seed(31)
    pol = M.R_MAPPOPolicy(a, [D], [S], M.Discrete(A))
    tr = M.R_MAPPO(a, pol)
    buf = M.SharedReplayBuffer(a, Ma, [D], [S], M.Discrete(A))
    g = torch.Generator(device="cuda").manual_seed(37)
    act = torch.zeros(tuple(buf.active_masks.shape), device="cuda")
    act[:, keep] = (torch.rand(act[:, keep].shape, device="cuda", generator=g) > 0.2).float()
    _random_fill(buf, A, g, active=act)
    oa = O.default_args(n_rollout_threads=len(keep), **common)
    opol = O.PolicyRef(oa, D, S, A)
    opol.actor.load_state_dict({k: v.cpu() for k, v in pol.actor.state_dict().items()})
    opol.critic.load_state_dict({k: v.cpu() for k, v in pol.critic.state_dict().items()})
    ob = O.BufferRef(oa, Ma, D, S, A)
    for n in BUF_NAMES:
        getattr(ob, n)[...] = getattr(buf, n)[:, keep].cpu().numpy()
    # chunk c of a buffer with R series = series c // (T/L), time block c % (T/L); series = thread * Ma + agent
    per = T // L
    n_sub, n_full = len(keep) * Ma * per, N * Ma * per
    rng = np.random.default_rng(41)
    sub_perms, full_perms = [], []
    for e in range(E):
        sp = rng.permutation(n_sub)
        k_idx, rest = sp // (Ma * per), sp % (Ma * per)
        mapped = np.asarray(keep)[k_idx] * (Ma * per) + rest                     # the same chunks, indexed in the full buffer
        inactive = rng.permutation(np.setdiff1d(np.arange(n_full), mapped))
        hs, hf = n_sub // nmb, n_full // nmb
        fp = []
        for k in range(nmb):
            part = np.concatenate([mapped[k * hs:(k + 1) * hs], inactive[k * (hf - hs):(k + 1) * (hf - hs)]])
            fp.append(rng.permutation(part))                                     # order inside a minibatch only reorders sums
        sub_perms.append(sp); full_perms.append(np.concatenate(fp))
    oinfo = O.train_ref(oa, opol, None, ob, perms=sub_perms)
    it = iter(full_perms)
    buf._randperm = lambda n: torch.from_numpy(next(it)).to(buf.device)          # the permutation stream of recurrent_rows
    info = tr.train(buf)
    for k in oinfo:
        if k != "ratio":
            close(info[k], oinfo[k], 2e-4, 1e-6, k)
    for net, onet in ((pol.actor, opol.actor), (pol.critic, opol.critic)):
        for k, v in net.state_dict().items():
            close(v, onet.state_dict()[k].numpy(), 2e-4, 1e-5, k)


@pytest.mark.parametrize("Do,Ds,A,N,Ma", [(30, 48, 9, 37, 3), (176, 322, 18, 9, 10)])
def test_fused_insert_and_recurrent_step_matches_separate_launches(M, Do, Ds, A, N, Ma):
    """mappo_recurrent_rollout_step (SMAC insert of the pending env output + get_actions / get_values on it, one launch) ==
    SharedReplayBuffer.insert_smac_fused followed by the rollout step on the slot it filled (smac_runner.py:110-151): the slot
    arrays bit for bit (obs, share_obs, available_actions, rewards, masks, bad_masks, active_masks, masked rnn states), the same
    actions, log-probs / values / next states to fp32 rounding.  Narrow (trunks in registers) and wide (split-K trunks) inputs."""
    from mappo_amd import recurrent
    T = 6
    a = make_args(M, episode_length=T, n_rollout_threads=N, use_recurrent_policy=True, algorithm_name="rmappo")
    torch.manual_seed(5)
    pol = M.R_MAPPOPolicy(a, [Do], [Ds], M.Discrete(A))
    R = N * Ma
    assert pol.can_fuse_recurrent_step(R)
    g = torch.Generator(device="cuda").manual_seed(8)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    obs, share = rnd(N, Ma, Do), rnd(N, Ma, Ds)
    avail = (torch.rand(N, Ma, A, device="cuda", generator=g) < 0.7).float()
    avail[:, :, 0] = 1.0
    rew = rnd(N).view(N, 1, 1).expand(N, Ma, 1)
    dones = torch.rand(N, Ma, device="cuda", generator=g) < 0.4
    dones[::3] = True                                              # some envs with ALL agents done: masks 0, states reset
    bad = torch.rand(N, Ma, device="cuda", generator=g) < 0.2
    ha, hc = rnd(R, 1, 64), rnd(R, 1, 64)
    step = 3
    names = ("obs", "share_obs", "available_actions", "rewards", "masks", "bad_masks", "active_masks", "rnn_states", "rnn_states_critic",
             "actions", "action_log_probs", "value_preds")
    res = []
    for fused in (False, True):
        buf = M.SharedReplayBuffer(a, Ma, [Do], [Ds], M.Discrete(A))
        for nme in names:
            getattr(buf, nme).fill_(float("nan"))
        buf.step = step - 1
        if fused:
            out = pol.collect_step_fused_recurrent(buf, step, (obs, share, rew, dones, bad, avail, ha, hc))
            assert out is not None
            _, na, nc = out
        else:
            assert buf.insert_smac_fused(share, obs, rew, dones, bad, avail, ha, hc)
            _, na, nc = pol.collect_into(buf, step, use_available_actions=True)
        assert buf.step == step
        res.append(dict({nme: getattr(buf, nme).clone() for nme in names}, na=na.clone(), nc=nc.clone()))
    sep, fus = res
    for nme in names[:9]:
        lo = step - 1 if nme == "rewards" else step
        np.testing.assert_array_equal(fus[nme][lo].cpu().numpy(), sep[nme][lo].cpu().numpy(), err_msg=nme)
        other = torch.ones(fus[nme].shape[0], dtype=torch.bool)
        other[lo] = False
        assert torch.isnan(fus[nme][other]).all(), nme             # nothing but the slot was written
    np.testing.assert_array_equal(fus["actions"][step].cpu().numpy(), sep["actions"][step].cpu().numpy())
    for nme in ("action_log_probs", "value_preds"):
        np.testing.assert_allclose(fus[nme][step].cpu().numpy(), sep[nme][step].cpu().numpy(), rtol=2e-6, atol=2e-6, err_msg=nme)
    np.testing.assert_allclose(fus["na"].cpu().numpy(), sep["na"].cpu().numpy(), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(fus["nc"].cpu().numpy(), sep["nc"].cpu().numpy(), rtol=2e-6, atol=2e-6)
    picked = torch.gather(avail.view(R, A), 1, fus["actions"][step].view(R, 1).long())
    assert float(picked.min()) == 1.0
