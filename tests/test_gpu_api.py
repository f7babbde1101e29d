"""GPU tests of the drop-in boundary: the reference's import paths (`mappo_amd.install_as_onpolicy`), a
NumPy-returning vec-env driven through those paths (train iteration + eval loop), `lr_decay`, and the naive-recurrent
training path — each against the oracle.  Tolerances as in tests/test_gpu_e2e.py."""
import numpy as np
import pytest
import torch

from oracle import mappo_oracle as O

pytestmark = pytest.mark.gpu

BUF_NAMES = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "value_preds", "returns", "available_actions",
             "actions", "action_log_probs", "rewards", "masks", "bad_masks", "active_masks")


def close(a, b, rtol=1e-5, atol=1e-6, msg=""):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a.astype(np.float64), b.astype(np.float64), rtol=rtol, atol=atol, err_msg=msg)


class _Space:
    """duck-typed spaces as the reference reads them (class name + .n / .shape, utils/util.py:31-51)"""


class Discrete(_Space):
    def __init__(self, n):
        self.n = n


class Box(_Space):
    def __init__(self, dim):
        self.shape = (dim,)


class NumpyMPEEnv:
    """A vec-env with the reference's DummyVecEnv contract and NOTHING of this framework's own (no graph_safe /
    accepts_device_actions attributes): NumPy in, NumPy out (env_wrappers.py:257-272, envs/mpe/environment.py:117-148).
    The dynamics depend on the actions, so a runner that fed the wrong actions would be found out."""

    def __init__(self, N, M, D, A, T, seed=0):
        self.N, self.M, self.D, self.A, self.T = N, M, D, A, T
        self.rng = np.random.default_rng(seed)
        self.observation_space = [Box(D)] * M
        self.share_observation_space = [Box(D * M)] * M
        self.action_space = [Discrete(A)] * M
        self.t = 0
        self.received = []

    def reset(self):
        self.t = 0
        self.obs = self.rng.standard_normal((self.N, self.M, self.D)).astype(np.float32)
        return self.obs.copy()

    def step(self, actions_env):
        assert isinstance(actions_env, np.ndarray), f"the reference's vec-envs receive NumPy, got {type(actions_env)}"
        assert actions_env.shape == (self.N, self.M, self.A)
        np.testing.assert_array_equal(actions_env.sum(-1), 1.0)                     # one-hot (mpe_runner.py:119)
        self.received.append(actions_env.copy())
        a = actions_env.argmax(-1).astype(np.float32)
        self.t += 1
        self.obs = (0.5 * self.obs + 0.1 * a[..., None] + 0.5 * self.rng.standard_normal(self.obs.shape)).astype(np.float32)
        rew = np.repeat(-np.abs(self.obs).mean(axis=(1, 2), keepdims=True), self.M, axis=1).astype(np.float32) + 0.05 * a[..., None]
        dones = np.full((self.N, self.M), self.t % self.T == 0)
        infos = [[{"individual_reward": float(rew[n, m, 0])} for m in range(self.M)] for n in range(self.N)]
        return self.obs.copy(), rew.astype(np.float32), dones, infos


def _args(get_config, **kw):
    a = get_config().parse_known_args([])[0]
    a.use_recurrent_policy = False
    a.use_naive_recurrent_policy = False
    for k, v in kw.items():
        assert hasattr(a, k), k
        setattr(a, k, v)
    return a


def _twin(runner, oa, D, S, A):
    opol = O.PolicyRef(oa, D, S, A)
    opol.actor.load_state_dict({k: v.cpu() for k, v in runner.policy.actor.state_dict().items()})
    opol.critic.load_state_dict({k: v.cpu() for k, v in runner.policy.critic.state_dict().items()})
    return opol, O.ValueNormRef()


def test_numpy_env_iteration_and_eval_through_onpolicy_import_paths(gpu_device):
    """install_as_onpolicy(), then everything through the REFERENCE's import paths: one training iteration and the eval loop
    of MPERunner under a NumPy vec-env (the drop-in claim of SURVEY 8b), checked against the oracle on the same data."""
    import mappo_amd
    mappo_amd.install_as_onpolicy(force=True)
    from onpolicy.config import get_config
    from onpolicy.runner.shared.mpe_runner import MPERunner
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    import mappo_amd.runner.shared.mpe_runner as ours
    assert MPERunner is ours.MPERunner

    T, N, Ma, D, A = 25, 8, 3, 18, 5
    a = _args(get_config, episode_length=T, n_rollout_threads=N, ppo_epoch=3, lr=7e-4, critic_lr=7e-4, seed=1, env_name="MPE",
              n_eval_rollout_threads=4, use_eval=True)
    torch.manual_seed(1)
    env, eval_env = NumpyMPEEnv(N, Ma, D, A, T, seed=1), NumpyMPEEnv(4, Ma, D, A, T, seed=2)
    runner = MPERunner(dict(all_args=a, envs=env, eval_envs=eval_env, num_agents=Ma, device=torch.device("cuda"), run_dir=None))
    assert isinstance(runner.buffer, SharedReplayBuffer) and isinstance(runner.trainer, R_MAPPO) and isinstance(runner.policy, R_MAPPOPolicy)
    assert not runner._use_graph                                               # a host env is never captured
    oa = O.default_args(episode_length=T, n_rollout_threads=N, ppo_epoch=3, lr=7e-4, critic_lr=7e-4)
    runner.warmup()
    opol, ovn = _twin(runner, oa, D, D * Ma, A)
    runner.rollout()
    b = runner.buffer
    assert len(env.received) == T
    # the env got exactly the actions the buffer recorded
    for t in range(T):
        np.testing.assert_array_equal(env.received[t].argmax(-1), b.actions[t].view(N, Ma).cpu().numpy().astype(np.int64))
    with torch.no_grad():
        lp, _, _ = opol.actor.evaluate_actions(b.obs[:T].reshape(-1, D).cpu(), None, b.actions.reshape(-1, 1).cpu(), None)
        v, _ = opol.critic(b.share_obs[:T].reshape(-1, D * Ma).cpu(), None, None)
    close(b.action_log_probs.reshape(-1, 1), lp.numpy(), 1e-5, 2e-6)
    close(b.value_preds[:T].reshape(-1, 1), v.numpy(), 1e-5, 2e-6)
    ob = O.BufferRef(oa, Ma, D, D * Ma, A)
    for n in BUF_NAMES:
        if n != "returns":
            getattr(ob, n)[...] = getattr(b, n).cpu().numpy()
    with torch.no_grad():
        nv, _ = opol.critic(torch.from_numpy(np.concatenate(ob.share_obs[-1])), None, None)
    ob.compute_returns(np.array(np.split(nv.numpy(), N)), ovn)
    close(b.returns[:T], ob.returns[:T], 1e-5, 3e-6)
    oinfo = O.train_ref(oa, opol, ovn, ob)
    info = runner.train()
    for k in oinfo:
        close(info[k], oinfo[k], 1e-4, 1e-6, k)
    for k, vv in runner.policy.actor.state_dict().items():
        close(vv, opol.actor.state_dict()[k].numpy(), 1e-4, 5e-6, k)

    # ---- eval loop (mpe_runner.py:141-183): deterministic act on the eval envs ----
    opol.actor.load_state_dict({k: v.cpu() for k, v in runner.policy.actor.state_dict().items()})     # same weights bit for bit
    logged = {}
    runner.log_env = lambda env_infos, total: logged.update(env_infos)
    twin = NumpyMPEEnv(4, Ma, D, A, T, seed=2)                                  # same seed: the oracle replays the eval episode
    obs = twin.reset()
    total = np.zeros((4, Ma, 1), np.float64)
    for _ in range(T):
        with torch.no_grad():
            acts, _, _ = opol.actor(torch.from_numpy(obs.reshape(-1, D)), None, None, None, deterministic=True)
        obs, rew, dones, _ = twin.step(np.eye(A, dtype=np.float32)[acts.view(4, Ma).numpy()])
        total += rew
    runner.eval(0)
    assert len(eval_env.received) == T
    for t in range(T):
        np.testing.assert_array_equal(eval_env.received[t], twin.received[t], err_msg=f"eval step {t}: deterministic actions")
    close(logged["eval_average_episode_rewards"][0], float(total.mean()), 1e-5, 1e-6)


def test_lr_decay_matches_reference_schedule(gpu_device):
    """R_MAPPOPolicy.lr_decay (rMAPPOPolicy.py:39-46, utils/util.py:17-21): lr = lr0 - lr0 * episode / episodes for both
    optimizers, visible to the device-resident Adam (opt_hyper) — an update after the decay equals the oracle's with that lr."""
    from mappo_amd.config import get_config
    from mappo_amd.utils.util import Discrete as D_
    from mappo_amd.utils.shared_buffer import SharedReplayBuffer
    from mappo_amd.algorithms.r_mappo.r_mappo import R_MAPPO
    from mappo_amd.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    T, N, Ma, D, A = 10, 4, 3, 18, 5
    lr0, clr0, ep, eps = 7e-4, 5e-4, 3, 10
    a = _args(get_config, episode_length=T, n_rollout_threads=N, ppo_epoch=2, lr=lr0, critic_lr=clr0, use_hip_graph=False)
    torch.manual_seed(2)
    pol = R_MAPPOPolicy(a, [D], [D * Ma], D_(A))
    tr = R_MAPPO(a, pol)
    pol.lr_decay(ep, eps)
    want_a, want_c = lr0 - lr0 * (ep / float(eps)), clr0 - clr0 * (ep / float(eps))
    assert pol.actor_optimizer.param_groups[0]["lr"] == pytest.approx(want_a, rel=1e-12)
    assert pol.critic_optimizer.param_groups[0]["lr"] == pytest.approx(want_c, rel=1e-12)
    close(pol.opt_hyper[:, 0], np.array([want_a, want_c], np.float32), 1e-7, 0)
    buf = SharedReplayBuffer(a, Ma, [D], [D * Ma], D_(A))
    rng = np.random.default_rng(5)
    f = np.float32
    for n in ("share_obs", "obs", "rewards"):
        arr = getattr(buf, n); arr.copy_(torch.from_numpy(rng.standard_normal(tuple(arr.shape)).astype(f)))
    buf.value_preds.copy_(torch.from_numpy((rng.standard_normal(tuple(buf.value_preds.shape)) * 0.3).astype(f)))
    buf.returns.copy_(torch.from_numpy((rng.standard_normal(tuple(buf.returns.shape)) * 2).astype(f)))
    buf.actions.copy_(torch.from_numpy(rng.integers(0, A, tuple(buf.actions.shape)).astype(f)))
    buf.action_log_probs.copy_(torch.from_numpy((-np.abs(rng.standard_normal(tuple(buf.actions.shape))) - 1).astype(f)))
    oa = O.default_args(episode_length=T, n_rollout_threads=N, ppo_epoch=2, lr=want_a, critic_lr=want_c)
    opol = O.PolicyRef(oa, D, D * Ma, A)
    opol.actor.load_state_dict({k: v.cpu() for k, v in pol.actor.state_dict().items()})
    opol.critic.load_state_dict({k: v.cpu() for k, v in pol.critic.state_dict().items()})
    ob = O.BufferRef(oa, Ma, D, D * Ma, A)
    for n in BUF_NAMES:
        getattr(ob, n)[...] = getattr(buf, n).cpu().numpy()
    ovn = O.ValueNormRef()
    oinfo = O.train_ref(oa, opol, ovn, ob)
    info = tr.train(buf)
    for k in oinfo:
        close(info[k], oinfo[k], 1e-4, 1e-6, k)
    for k, v in pol.actor.state_dict().items():
        close(v, opol.actor.state_dict()[k].numpy(), 1e-4, 5e-6, k)
    for k, v in pol.critic.state_dict().items():
        close(v, opol.critic.state_dict()[k].numpy(), 1e-4, 5e-6, k)


@pytest.mark.parametrize("use_graph", [False, True])
def test_naive_recurrent_train_vs_oracle(gpu_device, use_graph):
    """use_naive_recurrent_policy: whole-episode sequences per (thread, agent) (shared_buffer.py:283-383) through train(),
    eager and — with the device permutation stream — the captured hipGraph replay, against the oracle's naive-recurrent
    train on the same buffer / weights.  The CPU permutation stream pins the minibatch membership for the eager run; the
    replayed run (device permutations) is checked for the permutation-independent case num_mini_batch == 1."""
    from mappo_amd.config import get_config
    from mappo_amd.utils.util import Discrete as D_
    from mappo_amd.utils.shared_buffer import SharedReplayBuffer
    from mappo_amd.algorithms.r_mappo.r_mappo import R_MAPPO
    from mappo_amd.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    T, N, Ma, D, A = 12, 4, 3, 18, 5
    nmb = 1 if use_graph else 2
    a = _args(get_config, episode_length=T, n_rollout_threads=N, lr=7e-4, critic_lr=7e-4, ppo_epoch=2, num_mini_batch=nmb,
              use_naive_recurrent_policy=True, perm_device="cuda" if use_graph else "cpu", use_hip_graph=use_graph)
    torch.manual_seed(5)
    pol = R_MAPPOPolicy(a, [D], [D * Ma], D_(A))
    tr = R_MAPPO(a, pol)
    buf = SharedReplayBuffer(a, Ma, [D], [D * Ma], D_(A))
    rng = np.random.default_rng(3)
    f = np.float32

    def fill():
        for n in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "rewards"):
            arr = getattr(buf, n); arr.copy_(torch.from_numpy(rng.standard_normal(tuple(arr.shape)).astype(f)))
        buf.value_preds.copy_(torch.from_numpy((rng.standard_normal(tuple(buf.value_preds.shape)) * 0.3).astype(f)))
        buf.returns.copy_(torch.from_numpy((rng.standard_normal(tuple(buf.returns.shape)) * 2).astype(f)))
        buf.actions.copy_(torch.from_numpy(rng.integers(0, A, tuple(buf.actions.shape)).astype(f)))
        buf.action_log_probs.copy_(torch.from_numpy((-np.abs(rng.standard_normal(tuple(buf.actions.shape))) - 1).astype(f)))
        buf.masks.copy_(torch.from_numpy((rng.random(tuple(buf.masks.shape)) > 0.15).astype(f)))
        buf.active_masks.copy_(torch.from_numpy((rng.random(tuple(buf.masks.shape)) > 0.2).astype(f)))

    oa = O.default_args(episode_length=T, n_rollout_threads=N, lr=7e-4, critic_lr=7e-4, ppo_epoch=2, num_mini_batch=nmb,
                        use_naive_recurrent_policy=True)
    opol = O.PolicyRef(oa, D, D * Ma, A)
    opol.actor.load_state_dict({k: v.cpu() for k, v in pol.actor.state_dict().items()})
    opol.critic.load_state_dict({k: v.cpu() for k, v in pol.critic.state_dict().items()})
    ovn = O.ValueNormRef()
    for it in range(3 if use_graph else 1):                     # eager, capture + replay, replay
        fill()
        ob = O.BufferRef(oa, Ma, D, D * Ma, A)
        for n in BUF_NAMES:
            getattr(ob, n)[...] = getattr(buf, n).cpu().numpy()
        torch.manual_seed(11 + it)
        perms = [torch.randperm(N * Ma).numpy() for _ in range(2)]
        oinfo = O.train_ref(oa, opol, ovn, ob, perms=perms)
        torch.manual_seed(11 + it)
        info = tr.train(buf)
        for k in oinfo:
            close(info[k], oinfo[k], 1e-4, 1e-6, f"iteration {it}: {k}")
        for k, v in pol.actor.state_dict().items():
            close(v, opol.actor.state_dict()[k].numpy(), 1e-4, 8e-6, f"iteration {it}: {k}")
        for k, v in pol.critic.state_dict().items():
            close(v, opol.critic.state_dict()[k].numpy(), 1e-4, 8e-6, f"iteration {it}: {k}")
    if use_graph:
        assert any(isinstance(g, torch.cuda.CUDAGraph) for g in tr._graphs.values())


class PoolEnv:
    """Replays pre-drawn observations / rewards (independent of the actions): once as a CPU vec-env (NumPy in and out), once as
    a device-resident env (tensors in and out).  Same data, so the two rollouts must fill the buffer identically."""

    def __init__(self, obs, rew, T, A, device=None):
        self.T, self.t = T, 0
        self.N, self.M, self.D = obs.shape[1:]
        self.on_device = device is not None
        if self.on_device:
            self.accepts_device_actions = True
            self.obs, self.rew = torch.from_numpy(obs).to(device), torch.from_numpy(rew).to(device)
        else:
            self.obs, self.rew = obs, rew
        self.observation_space = [Box(self.D)] * self.M
        self.share_observation_space = [Box(self.D * self.M)] * self.M
        self.action_space = [Discrete(A)] * self.M
        self.received = []

    def reset(self):
        self.t = 0
        return self.obs[0]

    def step(self, actions_env):
        assert torch.is_tensor(actions_env) == self.on_device
        self.received.append(actions_env.cpu().numpy().copy() if self.on_device else actions_env.copy())
        self.t += 1
        dones = np.full((self.N, self.M), self.t % self.T == 0)
        if self.on_device:
            dones = torch.from_numpy(dones).to(self.obs.device)
        return self.obs[self.t], self.rew[self.t], dones, None


@pytest.mark.parametrize("fuse", [True, False])
def test_host_env_staging_equals_device_env(gpu_device, fuse):
    """a NumPy vec-env through the pinned double-buffered staging (mappo_amd/utils/host_staging.py) fills the
    buffer exactly as a device-resident env with the same data does, step by step, and trains to the same statistics —
    for the one-launch rollout step and for the separate insert / collect launches."""
    from mappo_amd.config import get_config
    from mappo_amd.runner.shared.mpe_runner import MPERunner
    T, N, Ma, D, A = 25, 16, 3, 18, 5
    rng = np.random.default_rng(9)
    obs = rng.standard_normal((2 * T + 2, N, Ma, D)).astype(np.float32)
    rew = np.repeat(rng.standard_normal((2 * T + 2, N, 1, 1)).astype(np.float32), Ma, axis=2)
    out = []
    for device in (None, torch.device("cuda:0")):
        a = _args(get_config, episode_length=T, n_rollout_threads=N, ppo_epoch=2, lr=7e-4, critic_lr=7e-4, seed=1, env_name="MPE",
                  use_hip_graph=False)
        a.fuse_rollout_step = fuse
        torch.manual_seed(1)
        env = PoolEnv(obs, rew, T, A, device)
        r = MPERunner(dict(all_args=a, envs=env, eval_envs=None, num_agents=Ma, device=torch.device("cuda:0"), run_dir=None))
        assert (r._staging is not None) == (device is None)
        r.warmup()
        infos = [r.run_episode()[0], r.run_episode()[0]]            # two episodes: the staging slots wrap around many times
        b = r.buffer
        out.append(({n: getattr(b, n).cpu().numpy().copy() for n in BUF_NAMES}, infos, np.stack(env.received),
                    r.policy.flat_params.cpu().numpy().copy()))
    (b0, i0, a0, p0), (b1, i1, a1, p1) = out
    np.testing.assert_array_equal(a0, a1, err_msg="actions handed to the envs")
    for n in BUF_NAMES:
        np.testing.assert_array_equal(b0[n], b1[n], err_msg=n)
    for x, y in zip(i0, i1):
        for k in x:
            assert x[k] == y[k], k
    np.testing.assert_array_equal(p0, p1)


@pytest.mark.gpu
def test_synthetic_smac_env_fused_step():
    """The one-launch synthetic SMAC env (bench utility): shapes, value ranges, episode structure and determinism."""
    from mappo_amd.envs.synthetic import SyntheticSMACEnv
    env = SyntheticSMACEnv(512, num_agents=5, obs_dim=30, share_dim=48, n_actions=9, p_death=0.05, p_term=0.1, seed=3)
    env.reset()
    died = term = 0
    acc = []
    for _ in range(20):
        dead_before = env.dead.clone()
        obs, share, rew, dones, bad, avail = env.step(None)
        assert obs.shape == (512, 5, 30) and share.shape == (512, 5, 48) and avail.shape == (512, 5, 9)
        assert rew.shape == (512, 5, 1) and dones.shape == (512, 5) and dones.dtype == torch.bool and not bad.any()
        assert torch.isfinite(obs).all() and torch.isfinite(share).all()
        assert (avail[:, :, 0] == 1).all() and ((avail == 0) | (avail == 1)).all()
        assert (rew[:, 0] == rew[:, 4]).all()                               # shared reward
        all_done = dones.all(dim=1)
        assert (dones | ~dead_before).all()                                 # a dead agent reports done
        assert (env.dead <= dones).all()
        term += int((all_done & ~env.dead.any(dim=1)).sum())                # terminated envs restart alive
        died += int((env.dead & ~dead_before).sum())
        acc.append(torch.cat([obs.flatten(), share.flatten()]).clone())
    x = torch.cat(acc)
    assert abs(float(x.mean())) < 5e-3 and abs(float(x.std()) - 1.0) < 5e-3
    assert 0.6 < float(avail[:, :, 1:].mean()) < 0.8
    assert 0.05 * 512 * 20 < term < 0.2 * 512 * 20 and died > 0
    env2 = SyntheticSMACEnv(512, num_agents=5, obs_dim=30, share_dim=48, n_actions=9, p_death=0.05, p_term=0.1, seed=3)
    env2.reset()
    o2 = None
    for _ in range(20):
        o2 = env2.step(None)[0]
    assert torch.equal(o2, obs)


@pytest.mark.gpu
def test_synthetic_smac_env_pooled_equals_stepwise():
    """pool_steps = P (P steps per launch, handed out as views) produces exactly the stream of the one-launch-per-step env:
    observations, availability, rewards, dones and the carried `dead` state, over more than two pools."""
    from mappo_amd.envs.synthetic import SyntheticSMACEnv
    kw = dict(num_agents=4, obs_dim=30, share_dim=48, n_actions=9, p_death=0.05, p_term=0.1, seed=5)
    a, b = SyntheticSMACEnv(37, **kw), SyntheticSMACEnv(37, pool_steps=7, **kw)
    a.reset(); b.reset()
    for t in range(17):
        xa, xb = a.step(None), b.step(None)
        for i, (u, v) in enumerate(zip(xa, xb)):
            assert torch.equal(u, v), (t, i)
        if (t + 1) % 7 == 0:
            assert torch.equal(a.dead, b.dead), t          # (inside a pool the pooled env's state already sits at the pool's end)
