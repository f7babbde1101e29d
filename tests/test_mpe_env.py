"""MPE simple_spread.  PARITY UNPINNED: the reference's env modules need seaborn / gym, which are not
installed, and it holds no fixtures for them (see oracle/mpe_oracle.py).  CPU tests pin the oracle's restatement with
properties the dynamics must have; the GPU tests compare the HIP kernel with the oracle step by step (float64 physics on
both sides; tolerance 1e-6 = the fp32 cast of the outputs) and run a full training iteration on the device env."""
import numpy as np
import pytest

from oracle import mpe_oracle as R


def _state(N, M, L, seed):
    rng = np.random.default_rng(seed)
    return rng.uniform(-1, 1, (N, M, 2)), rng.standard_normal((N, M, 2)) * 0.3, 0.8 * rng.uniform(-1, 1, (N, L, 2))


def _onehot(idx, A=5):
    return np.eye(A)[idx]


def test_oracle_collision_force_is_action_reaction_and_momentum_damps():
    """Pairwise forces cancel (core.py:305-308: equal masses), so with no action the total momentum only damps:
    sum v' = 0.75 sum v; far-apart agents feel no force; overlapping agents are pushed apart along their separation."""
    pos, vel, _ = _state(1, 3, 3, 0)
    pos[0, 1] = pos[0, 0] + np.array([0.2, 0.05])                       # agents 0 and 1 overlap (0.206 < 0.3)
    f = R.collision_forces(pos[0])
    np.testing.assert_allclose(f.sum(0), 0.0, atol=1e-12)
    assert np.dot(f[0], pos[0, 0] - pos[0, 1]) > 0 and np.dot(f[1], pos[0, 1] - pos[0, 0]) > 0
    p2, v2 = R.world_step(pos[0], vel[0], np.zeros((3, 5)))
    np.testing.assert_allclose(v2.sum(0), 0.75 * vel[0].sum(0), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(p2, pos[0] + 0.1 * v2, rtol=0, atol=1e-15)
    far = np.array([[0.0, 0.0], [5.0, 0.0], [0.0, 5.0]])
    np.testing.assert_allclose(R.collision_forces(far), 0.0, atol=1e-300)


def test_oracle_action_decoding_and_reward_counts_self_collision():
    """u = [a1 - a2, a3 - a4] * 5 (environment.py:223-235); an agent is always 'in collision' with itself (simple_spread.py:80-83
    loops over all agents), so the shared reward of M well-separated agents on top of the landmarks is -M."""
    np.testing.assert_array_equal(R.action_force(_onehot(np.array([0, 1, 2, 3, 4]))), [[0, 0], [5, 0], [-5, 0], [0, 5], [0, -5]])
    lpos = np.array([[0.0, 0.0], [2.0, 0.0], [0.0, 2.0]])
    assert R.reward(lpos.copy(), lpos) == pytest.approx(-3.0)
    assert R.observation(lpos, np.zeros((3, 2)), lpos).shape == (3, 18)


def test_oracle_permutation_and_translation_symmetry():
    pos, vel, lpos = _state(1, 3, 3, 1)
    acts = _onehot(np.array([2, 4, 1]))
    perm = np.array([2, 0, 1])
    p1, v1 = R.world_step(pos[0], vel[0], acts)
    p2, v2 = R.world_step(pos[0][perm], vel[0][perm], acts[perm])
    np.testing.assert_allclose(p2, p1[perm], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(R.reward(p2, lpos[0]), R.reward(p1, lpos[0]), rtol=1e-13)
    shift = np.array([3.0, -2.0])
    o1 = R.observation(p1, v1, lpos[0])
    o2 = R.observation(p1 + shift, v1, lpos[0] + shift)
    np.testing.assert_allclose(np.delete(o2, [2, 3], axis=1), np.delete(o1, [2, 3], axis=1), rtol=0, atol=1e-12)   # all but own position


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["onehot", "index"])
def test_kernel_matches_oracle_step_by_step(gpu_device, mode):
    """60 steps (two time-limit resets inside) of 64 environments from explicit states, random actions: observations,
    shared rewards and dones of mappo_mpe_spread_step against the oracle.  Crowded starts make agents collide."""
    import torch
    from mappo_amd.envs.mpe_spread import SimpleSpreadVecEnv
    N, M, L, T = 64, 3, 3, 25
    pos, vel, lpos = _state(N, M, L, 7)
    pos[: N // 2] *= 0.25                                              # half of the environments start crowded
    env = SimpleSpreadVecEnv(N, M, L, T, seed=5)
    env.reset()
    env.set_state(pos, vel, lpos)
    ref = R.SimpleSpreadRef(pos, vel, lpos, T)
    rng = np.random.default_rng(11)
    n_coll = 0
    for step in range(60):
        idx = rng.integers(0, 5, (N, M))
        a = torch.from_numpy(_onehot(idx).astype(np.float32)).cuda() if mode == "onehot" else torch.from_numpy(idx.astype(np.float32)).cuda().view(N, M, 1)
        obs, rew, dones, _ = env.step(a)
        n_coll += int((np.linalg.norm(ref.pos[:, 0] - ref.pos[:, 1], axis=-1) < 0.3).sum())
        d_env = dones.cpu().numpy()

        def reset_states(n):                                           # the oracle adopts the kernel's own reset draw
            return env.agent_pos[n].cpu().numpy(), env.agent_vel[n].cpu().numpy(), env.landmark_pos[n].cpu().numpy()
        o_ref, r_ref, d_ref = ref.step(_onehot(idx), reset_states)
        np.testing.assert_array_equal(d_env, d_ref, err_msg=f"step {step}")
        assert d_env.all() == ((step + 1) % T == 0)
        np.testing.assert_allclose(rew.cpu().numpy(), r_ref, rtol=1e-6, atol=1e-6, err_msg=f"rewards, step {step}")
        np.testing.assert_allclose(obs.cpu().numpy(), o_ref, rtol=1e-6, atol=1e-6, err_msg=f"obs, step {step}")
        if d_env.all():
            ap, lp = env.agent_pos.cpu().numpy(), env.landmark_pos.cpu().numpy()
            assert np.abs(ap).max() <= 1.0 and np.abs(lp).max() <= 0.8 and float(env.agent_vel.abs().max()) == 0.0
            assert len(np.unique(ap.round(6))) > N                     # fresh draws, different per environment
    assert n_coll > 50, "the test is meant to exercise the collision force"


@pytest.mark.gpu
def test_reset_draws_are_uniform_and_reproducible(gpu_device):
    import torch
    from mappo_amd.envs.mpe_spread import SimpleSpreadVecEnv
    e1, e2, e3 = (SimpleSpreadVecEnv(4096, 3, 3, 25, seed=s) for s in (3, 3, 4))
    o1, o2, o3 = e1.reset().clone(), e2.reset().clone(), e3.reset().clone()
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)
    p, l = e1.agent_pos.cpu().numpy(), e1.landmark_pos.cpu().numpy()
    assert abs(p.mean()) < 0.03 and abs(p.var() - 1 / 3) < 0.02 and abs(l.var() - 0.64 / 3) < 0.02
    assert float(o1[..., :2].abs().max()) == 0.0                       # velocities start at rest
    o4 = e1.reset()
    assert not torch.equal(o4, o1)                                     # next episode, new draw


@pytest.mark.gpu
@pytest.mark.parametrize("use_graph", [False, True])
def test_training_iteration_on_the_device_env(gpu_device, use_graph):
    """MPERunner on SimpleSpreadVecEnv (real MPE dynamics, index actions, one-launch rollout step, optional hipGraph replay):
    what the buffer recorded is consistent with the env's rules, and the losses match the oracle's train() on that buffer."""
    import torch
    from mappo_amd.config import get_config
    from mappo_amd.envs.mpe_spread import SimpleSpreadVecEnv
    from mappo_amd.runner.shared.mpe_runner import MPERunner
    from oracle import mappo_oracle as O
    T, N, M, D, A = 25, 32, 3, 18, 5
    a = get_config().parse_known_args([])[0]
    a.use_recurrent_policy = a.use_naive_recurrent_policy = False
    a.episode_length, a.n_rollout_threads, a.ppo_epoch, a.lr, a.critic_lr, a.seed, a.env_name = T, N, 3, 7e-4, 7e-4, 1, "MPE"
    a.use_hip_graph = use_graph
    torch.manual_seed(1)
    env = SimpleSpreadVecEnv(N, M, 3, T, seed=1)
    r = MPERunner(dict(all_args=a, envs=env, eval_envs=None, num_agents=M, device=torch.device("cuda"), run_dir=None))
    r.warmup()
    for _ in range(3 if use_graph else 1):
        r.rollout()
        b = r.buffer
        obs = b.obs.cpu().numpy()
        # consecutive observations obey the integrator: pos' = pos + 0.1 vel' (inside an episode)
        np.testing.assert_allclose(obs[2:T, ..., 2:4], obs[1:T - 1, ..., 2:4] + 0.1 * obs[2:T, ..., 0:2], rtol=0, atol=2e-6)
        rew = b.rewards.cpu().numpy()
        assert (rew <= -M + 1e-6).all() and np.abs(rew[:, :, 0] - rew[:, :, 1]).max() == 0.0      # shared, and at most -M
        assert float(b.masks[T].max()) == 0.0 and float(b.masks[1:T].min()) == 1.0
        info = r.train()
    if use_graph:
        assert isinstance(r._rollout_graph, torch.cuda.CUDAGraph)
    assert np.isfinite(list(info.values())).all() and info["dist_entropy"] > 1.0
