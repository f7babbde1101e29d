"""GPU parity tests of the host-side mirror (SharedReplayBuffer / R_MAPPOPolicy / R_MAPPO / MPERunner on the
HIP kernels) against the reference's own outputs (tests/golden/*.npz) and the oracle.

Tolerances as in tests/test_oracle_golden.py: 1e-5 relative on losses / returns / forward outputs; post-Adam
parameters carry an absolute floor of 3e-6 (lr = 7e-4: where |g| ~ eps the first Adam steps amplify an fp32
re-association of g to ~0.3 % of lr — the oracle shows the same spread against the reference)."""
import os

import numpy as np
import pytest
import torch

from conftest import golden, sub, ROOT
from oracle import mappo_oracle as O

pytestmark = pytest.mark.gpu

TUPLE = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "returns",
         "masks", "active_masks", "old_action_log_probs", "adv_targ", "available_actions")
BUF_NAMES = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "value_preds", "returns", "available_actions",
             "actions", "action_log_probs", "rewards", "masks", "bad_masks", "active_masks")


def close(a, b, rtol=1e-5, atol=1e-6, msg=""):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a.astype(np.float64), b.astype(np.float64), rtol=rtol, atol=atol, err_msg=msg)


@pytest.fixture(scope="module")
def M(gpu_device):
    import mappo_amd
    from mappo_amd.config import get_config
    from mappo_amd.utils.shared_buffer import SharedReplayBuffer
    from mappo_amd.utils.util import Discrete
    from mappo_amd.algorithms.r_mappo.r_mappo import R_MAPPO
    from mappo_amd.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    from mappo_amd.runner.shared.mpe_runner import MPERunner
    from mappo_amd.envs.synthetic import SyntheticMPEEnv

    class NS:
        pass
    ns = NS()
    ns.get_config, ns.SharedReplayBuffer, ns.Discrete, ns.R_MAPPO, ns.R_MAPPOPolicy = get_config, SharedReplayBuffer, Discrete, R_MAPPO, R_MAPPOPolicy
    ns.MPERunner, ns.SyntheticMPEEnv = MPERunner, SyntheticMPEEnv
    return ns


def make_args(M, **kw):
    a = M.get_config().parse_known_args([])[0]
    a.use_recurrent_policy = False
    a.use_naive_recurrent_policy = False
    for k, v in kw.items():
        assert hasattr(a, k), k
        setattr(a, k, v)
    return a


def fill_buffer(buf, d, prefix="buf/"):
    for n in BUF_NAMES:
        getattr(buf, n).copy_(torch.from_numpy(np.ascontiguousarray(d[prefix + n])))


def test_insert_after_update_slots(M):
    g = golden("insert")
    T, N, Ma = 4, 2, 3
    a = make_args(M, episode_length=T, n_rollout_threads=N, hidden_size=8)
    buf = M.SharedReplayBuffer(a, Ma, [6], [18], M.Discrete(5))
    for s in range(int(g["n_inserts"])):
        d = sub(g, f"in{s}")
        buf.insert(d["share_obs"], d["obs"], d["rnn_a"], d["rnn_c"], d["actions"], d["logp"], d["values"], d["rewards"],
                   d["masks"], d["bad"], d["active"], d["avail"])
        assert buf.step == int(g[f"step_after{s}"])
        if s == T - 1:
            for n in BUF_NAMES:
                np.testing.assert_array_equal(getattr(buf, n).cpu().numpy(), g["full/" + n])
            buf.after_update()
            for n in BUF_NAMES:
                np.testing.assert_array_equal(getattr(buf, n).cpu().numpy(), g["after_update/" + n])
    for n in BUF_NAMES:
        np.testing.assert_array_equal(getattr(buf, n).cpu().numpy(), g["final/" + n])


def test_generators_bit_exact(M):
    """Same torch seed + perm_device='cpu' => the reference's permutation and every yielded array, bit for bit."""
    g = golden("generators")
    for c in range(int(g["n_cases"])):
        d = sub(g, f"c{c}")
        kind, T, N, Ma, nmb, L, seed = [int(x) for x in d["spec"]]
        a = make_args(M, episode_length=T, n_rollout_threads=N, hidden_size=int(d["buf/rnn_states"].shape[-1]), perm_device="cpu")
        buf = M.SharedReplayBuffer(a, Ma, [int(d["buf/obs"].shape[-1])], [int(d["buf/share_obs"].shape[-1])],
                                   M.Discrete(int(d["buf/available_actions"].shape[-1])))
        fill_buffer(buf, d)
        torch.manual_seed(seed)
        if kind == 0:
            gen = buf.feed_forward_generator(d["adv"], nmb)
        elif kind == 1:
            gen = buf.recurrent_generator(d["adv"], nmb, L)
        else:
            gen = buf.naive_recurrent_generator(d["adv"], nmb)
        batches = list(gen)
        assert len(batches) == int(d["n_batches"])
        for bi, sample in enumerate(batches):
            for nm, arr in zip(TUPLE, sample):
                np.testing.assert_array_equal(arr.cpu().numpy(), d[f"b{bi}/{nm}"], err_msg=f"case {c} batch {bi} {nm}")


def test_compute_returns_through_buffer(M):
    g = golden("gae")
    for c in range(int(g["n_cases"])):
        d = sub(g, f"c{c}")
        use_gae, ptl, use_vn = [bool(x) for x in d["flags"]]
        T, N, Ma = d["rewards"].shape[:3]
        a = make_args(M, episode_length=T, n_rollout_threads=N, use_gae=use_gae, use_proper_time_limits=ptl, use_valuenorm=use_vn)
        buf = M.SharedReplayBuffer(a, Ma, [18], [54], M.Discrete(5))
        for n in ("rewards", "value_preds", "masks", "bad_masks"):
            getattr(buf, n).copy_(torch.from_numpy(d[n]))
        vn = None
        if use_vn:
            from mappo_amd.utils.valuenorm import ValueNorm
            vn = ValueNorm(1)
            vn.load_state_dict({"running_mean": torch.tensor([d["vn_state"][0]]), "running_mean_sq": torch.tensor([d["vn_state"][1]]),
                                "debiasing_term": torch.tensor(d["vn_state"][2])})
        buf.compute_returns(d["next_value"], vn)
        close(buf.returns, d["returns"], 1e-5, 2e-6, f"case {c}")


def load_policy(M, a, g, prefix, D, S, A):
    pol = M.R_MAPPOPolicy(a, [D], [S], M.Discrete(A))
    pol.actor.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sub(g, f"{prefix}/actor0").items()})
    pol.critic.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sub(g, f"{prefix}/critic0").items()})
    return pol


def set_vn(tr, state):
    tr.value_normalizer.load_state_dict({"running_mean": torch.tensor([state[0]]), "running_mean_sq": torch.tensor([state[1]]),
                                         "debiasing_term": torch.tensor(state[2])})


def test_policy_state_dict_keys_and_param_views(M):
    """Checkpoint compatibility: reference key names (incl. the unused fc_h) and views aliasing the flat buffer."""
    g = golden("ppo_update")
    a = make_args(M)
    pol = M.R_MAPPOPolicy(a, [18], [54], M.Discrete(5))
    assert set(pol.actor.state_dict().keys()) == set(sub(g, "c0/actor0").keys())
    assert set(pol.critic.state_dict().keys()) == set(sub(g, "c0/critic0").keys())
    assert sum(p.numel() for p in pol.actor.parameters()) == 10281 and sum(p.numel() for p in pol.critic.parameters()) == 12397
    w = pol.actor.base.mlp.fc1[0].weight
    w.data.fill_(3.0)
    off = dict((k, o) for k, o, _ in pol.actor.layout)["base.mlp.fc1.0.weight"]
    assert float(pol.flat_params[off]) == 3.0                       # the Parameter is a view of the flat buffer


def test_reference_seeded_init_matches(M):
    """Same torch.manual_seed => same initial weights as the reference (RNG consumption order kept).  The draws
    are identical; orthogonal_'s QR runs in the host's LAPACK, whose last-bit rounding differs between the build
    container (where the fixture was made) and the GPU box's CPU (observed: 3.5e-6 on 0.4 % of a 64x64 Q), hence
    an absolute 2e-5 instead of bit equality — still far below the 1e-1 scale of the weights."""
    g = golden("train")
    d = sub(g, "c2")
    T, N, Ma, D, S, A, H, nmb, rec, epochs, L = [int(x) for x in d["dims"]]
    torch.manual_seed(700 + 2)                                     # seed used by generate_golden.gen_train for c2
    a = make_args(M, lr=7e-4, critic_lr=7e-4)
    pol = M.R_MAPPOPolicy(a, [D], [S], M.Discrete(A))
    for k, v in pol.actor.state_dict().items():
        close(v, d[f"actor0/{k}"], 1e-5, 2e-5, k)
    for k, v in pol.critic.state_dict().items():
        close(v, d[f"critic0/{k}"], 1e-5, 2e-5, k)


@pytest.mark.parametrize("unfused", [False, True])
def test_ppo_update_golden_variants(M, unfused):
    """R_MAPPO.ppo_update on the reference's own sample tuples: the 6 returned statistics, post-step
    parameters, Adam moments and ValueNorm state (all MLP cases of the fixture that the kernels are tiled for)."""
    g = golden("ppo_update")
    done = 0
    for c in range(int(g["n_cases"])):
        d = sub(g, f"c{c}")
        T, N, Ma, D, S, A, H = [int(x) for x in d["dims"]]
        fl = dict(zip([str(x) for x in d["flag_names"]], [bool(x) for x in d["flags"]]))
        if fl["use_recurrent_policy"] or H != 64:
            continue
        hy = d["hyper"]
        a = make_args(M, episode_length=T, n_rollout_threads=N, hidden_size=H, clip_param=float(hy[0]), entropy_coef=float(hy[1]),
                      value_loss_coef=float(hy[2]), huber_delta=float(hy[3]), max_grad_norm=float(hy[4]), lr=float(hy[5]),
                      critic_lr=float(hy[6]), opti_eps=float(hy[7]), weight_decay=float(hy[8]), unfused_update=unfused,
                      **{k: v for k, v in fl.items() if k not in ("update_actor", "two_steps", "use_recurrent_policy")})
        pol = load_policy(M, a, g, f"c{c}", D, S, A)
        tr = M.R_MAPPO(a, pol)
        if a.use_valuenorm:
            set_vn(tr, d["vn0"])
        sample = tuple(d[f"sample/{nm}"] for nm in TUPLE)
        for rep in range(2 if fl["two_steps"] else 1):
            out = tr.ppo_update(sample, fl["update_actor"])
            # north_star: 1e-5 relative on losses.  Measured over the 12 variants (profiles/r03/tolerance_budget_stat_errors.json):
            # at most 1.0e-7 relative on any of the six statistics — checked at 2e-6, five times inside the target
            _record_stat_errors("ppo_update" + ("_unfused" if unfused else ""), out, d[f"r{rep}/stats"])
            close(np.array(out, dtype=np.float64), d[f"r{rep}/stats"], 2e-6, 1e-8, f"case {c} stats")
            for tag, net, opt, seg in (("actor", pol.actor, pol.actor_optimizer, 0), ("critic", pol.critic, pol.critic_optimizer, 1)):
                ref_sd = sub(g, f"c{c}/r{rep}/{tag}")
                for k, v in net.state_dict().items():
                    close(v, ref_sd[k], 1e-5, 3e-6, f"case {c} {tag} {k}")
                adam = sub(g, f"c{c}/r{rep}/{tag}_adam")
                ea = dict((k, v) for k, v in net.named_parameters())
                lo = pol.seg_bounds[seg]
                for key, off, shape in net.layout:
                    n = int(np.prod(shape))
                    if f"{key}/exp_avg" in adam:
                        close(pol.exp_avg[lo + off: lo + off + n].view(shape), adam[f"{key}/exp_avg"], 1e-4, 1e-8, f"{key} exp_avg")
                        close(pol.exp_avg_sq[lo + off: lo + off + n].view(shape), adam[f"{key}/exp_avg_sq"], 2e-4, 1e-12, f"{key} exp_avg_sq")
                        assert int(pol.opt_step[seg]) == int(adam[f"{key}/step"])
            if a.use_valuenorm:
                close(tr.value_normalizer.state, d[f"r{rep}/vn"], 2e-6, 1e-9)
        done += 1
    assert done >= 2


STAT_NAMES = ("value_loss", "critic_grad_norm", "policy_loss", "dist_entropy", "actor_grad_norm", "ratio")
_STAT_ERR = {}


def _record_stat_errors(tag, got, ref):
    """Max relative error per statistic over the fixture's cases; dumped to gpurun_out/stat_errors.json (committed under
    profiles/ as the measured budget behind the tolerances of these tests)."""
    import json
    got, ref = np.asarray(got, dtype=np.float64).reshape(-1), np.asarray(ref, dtype=np.float64).reshape(-1)
    e = _STAT_ERR.setdefault(tag, {})
    for n, gv, rv in zip(STAT_NAMES, got, ref):
        rel, ab = abs(gv - rv) / max(abs(rv), 1e-30), abs(gv - rv)
        cur = e.get(n, dict(max_rel=0.0, max_abs=0.0, ref_at_max_rel=0.0))
        if rel > cur["max_rel"]:
            cur["max_rel"], cur["ref_at_max_rel"] = rel, rv
        cur["max_abs"] = max(cur["max_abs"], ab)
        e[n] = cur
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "stat_errors.json"), "w") as f:
            json.dump(_STAT_ERR, f, indent=1, sort_keys=True)


def test_ppo_update_small_hidden_rejected(M):
    a = make_args(M, hidden_size=16)
    with pytest.raises(NotImplementedError):
        M.R_MAPPOPolicy(a, [18], [54], M.Discrete(5))


@pytest.mark.parametrize("case,unfused", [(0, False), (2, False), (0, True)])
def test_train_golden_end_to_end(M, case, unfused):
    """R_MAPPO.train on the reference's buffer: train_info and final parameters after ppo_epoch x num_mini_batch
    updates.  case 0: num_mini_batch=2 with the reference's CPU permutation stream (perm_device='cpu');
    case 2: num_mini_batch=1, default in-place streaming (no gather) — must equal the permuted reference run."""
    g = golden("train")
    d = sub(g, f"c{case}")
    T, N, Ma, D, S, A, H, nmb, rec, epochs, L = [int(x) for x in d["dims"]]
    a = make_args(M, episode_length=T, n_rollout_threads=N, lr=7e-4, critic_lr=7e-4, ppo_epoch=epochs, num_mini_batch=nmb,
                  perm_device="cpu", unfused_update=unfused)
    pol = load_policy(M, a, g, f"c{case}", D, S, A)
    tr = M.R_MAPPO(a, pol)
    buf = M.SharedReplayBuffer(a, Ma, [D], [S], M.Discrete(A))
    fill_buffer(buf, d)
    torch.manual_seed(3000 + case)
    info = tr.train(buf)
    ref = dict(zip([str(k) for k in d["info_keys"]], d["info"]))
    _record_stat_errors(f"train_c{case}" + ("_unfused" if unfused else ""), [info[k] for k in STAT_NAMES], [ref[k] for k in STAT_NAMES])
    for k, v in info.items():
        # averages over ppo_epoch x num_mini_batch updates: 8.5e-8 relative measured at most (same file); checked at 2e-6
        close(v, ref[k], 2e-6, 1e-8, k)
    for tag, net in (("actor1", pol.actor), ("critic1", pol.critic)):
        ref_sd = sub(g, f"c{case}/{tag}")
        for k, v in net.state_dict().items():
            close(v, ref_sd[k], 1e-4, 5e-6, f"{tag} {k}")
    close(tr.value_normalizer.state, d["vn1"], 2e-6, 1e-9)
    if nmb == 1:
        # the gather path (exact_minibatch_order) gives the same result as in-place streaming
        a2 = make_args(M, episode_length=T, n_rollout_threads=N, lr=7e-4, critic_lr=7e-4, ppo_epoch=epochs, num_mini_batch=nmb,
                       perm_device="cpu", exact_minibatch_order=True)
        pol2 = load_policy(M, a2, g, f"c{case}", D, S, A)
        tr2 = M.R_MAPPO(a2, pol2)
        torch.manual_seed(3000 + case)
        info2 = tr2.train(buf)
        for k in info:
            close(info2[k], info[k], 2e-5, 1e-7, k)
        close(pol2.flat_params, pol.flat_params, 1e-4, 5e-6)


def _oracle_twin(runner, oa, D, S, A):
    """Oracle policy / ValueNorm with the runner's current weights, Adam moments and normaliser state."""
    opol = O.PolicyRef(oa, D, S, A)
    pol = runner.policy
    for net, onet, opt, seg in ((pol.actor, opol.actor, opol.actor_optimizer, 0), (pol.critic, opol.critic, opol.critic_optimizer, 1)):
        onet.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
        step = int(pol.opt_step[seg])
        if step > 0:
            lo = pol.seg_bounds[seg]
            named = dict(onet.named_parameters())
            for key, off, shape in net.layout:
                n = int(np.prod(shape))
                opt.state[named[key]] = dict(step=torch.tensor(float(step)),
                                             exp_avg=pol.exp_avg[lo + off: lo + off + n].view(shape).cpu().clone(),
                                             exp_avg_sq=pol.exp_avg_sq[lo + off: lo + off + n].view(shape).cpu().clone())
    ovn = O.ValueNormRef()
    ovn.load_state(runner.trainer.value_normalizer.state.cpu().numpy())
    return opol, ovn


@pytest.mark.parametrize("n_warm,use_graph", [(0, True), (2, True), (2, False)])
def test_runner_iteration_vs_oracle(M, n_warm, use_graph):
    """Whole iteration at BASELINE config-1 shape (N=8, T=25, M=3): our MPERunner collects with its own sampled
    actions; the oracle then recomputes get_actions' log-probs / values on the same observations, bootstrap + GAE
    and the full train() from the same weights / Adam state / buffer — covers collect_into, insert_env, compute,
    train.  n_warm=2 with graphs: the compared rollout and train() are hipGraph REPLAYS (eager -> capture -> replay)."""
    T, N, Ma, D, A = 25, 8, 3, 18, 5
    a = make_args(M, episode_length=T, n_rollout_threads=N, ppo_epoch=3, lr=7e-4, critic_lr=7e-4, seed=1, env_name="MPE",
                  use_hip_graph=use_graph)
    torch.manual_seed(1)
    env = M.SyntheticMPEEnv(N, Ma, D, A, T, seed=1)
    runner = M.MPERunner(dict(all_args=a, envs=env, eval_envs=None, num_agents=Ma, device=torch.device("cuda"), run_dir=None))
    oa = O.default_args(episode_length=T, n_rollout_threads=N, ppo_epoch=3, lr=7e-4, critic_lr=7e-4)
    runner.warmup()
    prev_obs = None
    for _ in range(n_warm):
        runner.run_episode()
        cur = runner.buffer.obs[1:].clone()
        assert prev_obs is None or not torch.equal(cur, prev_obs)           # every (replayed) episode sees fresh data
        prev_obs = cur
    if use_graph and n_warm >= 2:
        assert isinstance(runner._rollout_graph, torch.cuda.CUDAGraph)
        assert any(isinstance(g, torch.cuda.CUDAGraph) for g in runner.trainer._graphs.values())
    opol, ovn = _oracle_twin(runner, oa, D, D * Ma, A)
    acts_before = runner.buffer.actions.clone()
    runner.rollout()                                                          # T x (collect, env.step, insert) + compute
    b = runner.buffer
    if n_warm:
        assert not torch.equal(b.actions, acts_before)                        # sampling stream advanced
    with torch.no_grad():
        obs_f = b.obs[:T].reshape(-1, D).cpu()
        act_f = b.actions.reshape(-1, 1).cpu()
        lp, _, _ = opol.actor.evaluate_actions(obs_f, None, act_f, None)
        v, _ = opol.critic(b.share_obs[:T].reshape(-1, D * Ma).cpu(), None, None)
    close(b.action_log_probs.reshape(-1, 1), lp.numpy(), 1e-5, 2e-6)
    close(b.value_preds[:T].reshape(-1, 1), v.numpy(), 1e-5, 2e-6)
    assert float(b.masks[T].min()) == 0.0 and float(b.masks[1:T].min()) == 1.0        # done on the T-th step only
    np.testing.assert_array_equal(b.share_obs[3, :, 0].cpu().numpy(), b.obs[3].reshape(N, -1).cpu().numpy())
    acts = b.actions.cpu().numpy()
    assert acts.min() >= 0 and acts.max() <= A - 1 and len(np.unique(acts)) > 1
    ob = O.BufferRef(oa, Ma, D, D * Ma, A)
    for n in BUF_NAMES:
        if n != "returns":
            getattr(ob, n)[...] = getattr(b, n).cpu().numpy()
    with torch.no_grad():
        nv, _ = opol.critic(torch.from_numpy(np.concatenate(ob.share_obs[-1])), None, None)
    ob.compute_returns(np.array(np.split(nv.numpy(), N)), ovn)
    close(b.returns[:T], ob.returns[:T], 1e-5, 3e-6)
    oinfo = O.train_ref(oa, opol, ovn, ob)
    last_obs = b.obs[-1].clone()
    info = runner.train()
    for k in oinfo:
        close(info[k], oinfo[k], 1e-4, 1e-6, k)
    for k, vv in runner.policy.actor.state_dict().items():
        close(vv, opol.actor.state_dict()[k].numpy(), 1e-4, 5e-6, k)
    for k, vv in runner.policy.critic.state_dict().items():
        close(vv, opol.critic.state_dict()[k].numpy(), 1e-4, 5e-6, k)
    close(runner.trainer.value_normalizer.state, ovn.state(), 2e-6, 1e-9)
    np.testing.assert_array_equal(b.obs[0].cpu().numpy(), last_obs.cpu().numpy())     # after_update: slot T -> slot 0


def test_checkpoint_roundtrip(M, tmp_path):
    a = make_args(M, episode_length=5, n_rollout_threads=4, env_name="MPE")
    env = M.SyntheticMPEEnv(4, 3, 18, 5, 5, seed=3)
    r1 = M.MPERunner(dict(all_args=a, envs=env, eval_envs=None, num_agents=3, device=torch.device("cuda"), run_dir=tmp_path))
    r1.warmup(); r1.run_episode()
    r1.save()
    a.model_dir = str(tmp_path / "models")
    r2 = M.MPERunner(dict(all_args=a, envs=env, eval_envs=None, num_agents=3, device=torch.device("cuda"), run_dir=tmp_path))
    np.testing.assert_array_equal(r1.policy.flat_params.cpu().numpy(), r2.policy.flat_params.cpu().numpy())
    np.testing.assert_array_equal(r1.trainer.value_normalizer.state.cpu().numpy(), r2.trainer.value_normalizer.state.cpu().numpy())


def test_cpu_device_refused(M):
    from mappo_amd._lib import MappoHipError
    a = make_args(M)
    with pytest.raises(MappoHipError):
        M.R_MAPPOPolicy(a, [18], [54], M.Discrete(5), device=torch.device("cpu"))


def test_concurrent_update_launch_matches_sequential(M):
    """BASELINE config-2 batch (76 800 samples): the split-grid two-stream launch of the actor / critic update kernels
    gives the same update as launching them one after the other (only the slab partition of the sums differs)."""
    T, N, Ma, D, A = 25, 1024, 3, 18, 5
    res = []
    for split in (True, False):
        a = make_args(M, episode_length=T, n_rollout_threads=N, ppo_epoch=2, lr=7e-4, critic_lr=7e-4, use_hip_graph=False,
                      concurrent_update=split)
        torch.manual_seed(3)
        pol = M.R_MAPPOPolicy(a, [D], [D * Ma], M.Discrete(A))
        tr = M.R_MAPPO(a, pol)
        if split:
            assert tr._split_grid(256, D, D * Ma, T * N * Ma)[0] > 0
        buf = M.SharedReplayBuffer(a, Ma, [D], [D * Ma], M.Discrete(A))
        g = torch.Generator(device="cuda").manual_seed(7)
        for n in ("share_obs", "obs", "rewards"):
            getattr(buf, n).copy_(torch.randn(getattr(buf, n).shape, device="cuda", generator=g))
        buf.value_preds.copy_(torch.randn(buf.value_preds.shape, device="cuda", generator=g) * 0.3)
        buf.returns.copy_(torch.randn(buf.returns.shape, device="cuda", generator=g) * 2)
        buf.actions.copy_(torch.randint(0, A, buf.actions.shape, device="cuda", generator=g).float())
        buf.action_log_probs.copy_(-torch.rand(buf.actions.shape, device="cuda", generator=g) - 1.2)
        buf.active_masks.copy_((torch.rand(buf.masks.shape, device="cuda", generator=g) > 0.2).float())
        info = tr.train(buf)
        res.append((info, pol.flat_params.clone(), pol.flat_grad.clone()))
    for k in res[0][0]:
        close(res[0][0][k], res[1][0][k], 1e-5, 1e-7, k)
    close(res[0][2], res[1][2], 1e-4, 1e-7, "last gradient")
    close(res[0][1], res[1][1], 1e-5, 2e-6, "parameters")


def test_data_parallel_path_matches_single_process(M):
    """The data-parallel trainer (moments first, per-epoch captured kernel segments, eager all-reduce + clip_adam, rollout
    graph with thread-local capture) on ONE rank with a real NCCL/RCCL process group must reproduce the single-process
    run: same weights after several iterations (the all-reduce over one rank is the identity), replays included."""
    import torch.distributed as dist
    from mappo_amd.distributed import DataParallel
    T, N, Ma, D, A = 25, 64, 3, 18, 5

    def run(dp):
        a = make_args(M, episode_length=T, n_rollout_threads=N, ppo_epoch=4, lr=7e-4, critic_lr=7e-4, seed=1, env_name="MPE")
        torch.manual_seed(1)
        env = M.SyntheticMPEEnv(N, Ma, D, A, T, seed=1)
        r = M.MPERunner(dict(all_args=a, envs=env, eval_envs=None, num_agents=Ma, device=torch.device("cuda"), run_dir=None, dist_group=dp))
        r.warmup()
        infos = [r.run_episode(i, 5)[0] for i in range(4)]           # eager, capture, replay, replay
        torch.cuda.synchronize()
        return infos, r.trainer.policy.flat_params.clone()

    infos0, p0 = run(None)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29571", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        infos1, p1 = run(DataParallel())
    finally:
        if created:
            dist.destroy_process_group()
    for i0, i1 in zip(infos0, infos1):
        for k in i0:
            np.testing.assert_allclose(i1[k], i0[k], rtol=2e-5, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(p1.cpu().numpy(), p0.cpu().numpy(), rtol=0, atol=2e-6)
