"""CPU restatement (NumPy float64) of MPE `simple_spread` as the reference steps it — TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

PARITY UNPINNED.  The reference's own modules cannot be imported in the build container without stand-ins
(`onpolicy/envs/mpe/core.py:2` imports seaborn, `onpolicy/envs/mpe/environment.py:1` imports gym; neither is installed), and
the reference holds no test or fixture for this path.  The functions below restate the cited lines; the GPU kernel
(mappo_amd/csrc/mpe_env.hip) is checked against them and against properties of the dynamics (action / reaction of the
collision force, damping of the total momentum, agent-permutation equivariance, translation invariance of the relative
observations) in tests/test_mpe_env.py.

    world.step            onpolicy/envs/mpe/core.py:207-229   (action force, environment force, integrate)
    collision force       onpolicy/envs/mpe/core.py:283-322   (softmax penetration, contact_force 1e2, contact_margin 1e-3)
    integrate_state       onpolicy/envs/mpe/core.py:264-275   (damping 0.25, dt 0.1)
    reset_world           onpolicy/envs/mpe/scenarios/simple_spread.py:32-47
    reward / observation  onpolicy/envs/mpe/scenarios/simple_spread.py:73-103
    action decoding, shared reward, done   onpolicy/envs/mpe/environment.py:117-148,179-185,200-245
    reset on done         onpolicy/envs/env_wrappers.py:146-152,676-682"""
import numpy as np

DT, DAMPING, CONTACT_FORCE, CONTACT_MARGIN, AGENT_SIZE, SENSITIVITY = 0.1, 0.25, 1e2, 1e-3, 0.15, 5.0


def action_force(actions_env):
    """actions_env [M, 5] (one-hot or probabilities) -> u [M, 2] * sensitivity (environment.py:223-235), mass 1, no noise."""
    a = np.asarray(actions_env, np.float64)
    u = np.stack([a[:, 1] - a[:, 2], a[:, 3] - a[:, 4]], axis=1)
    return u * SENSITIVITY


def collision_forces(pos):
    """Sum of the pairwise soft-collision forces on every agent (core.py:238-262,283-322); landmarks do not collide."""
    M = pos.shape[0]
    f = np.zeros((M, 2))
    for a in range(M):
        for b in range(a + 1, M):
            delta = pos[a] - pos[b]
            dist = np.sqrt(np.sum(np.square(delta)))
            dist_min = 2 * AGENT_SIZE
            k = CONTACT_MARGIN
            penetration = np.logaddexp(0, -(dist - dist_min) / k) * k
            force = CONTACT_FORCE * delta / dist * penetration
            f[a] = force + f[a]
            f[b] = -force + f[b]
    return f


def world_step(pos, vel, actions_env):
    """One World.step for ONE environment: returns the new (pos, vel) [M, 2]."""
    f = action_force(actions_env) + collision_forces(pos)
    vel = vel * (1 - DAMPING)
    vel = vel + f * DT
    pos = pos + vel * DT
    return pos, vel


def reward(pos, lpos):
    """Shared reward (environment.py:139-143): sum over agents of -(sum over landmarks of the closest agent's distance) -
    (number of agents within collision distance, the agent itself included: simple_spread.py:80-83 loops over all agents)."""
    M = pos.shape[0]
    base = 0.0
    for l in range(lpos.shape[0]):
        base -= min(np.sqrt(np.sum(np.square(pos[a] - lpos[l]))) for a in range(M))
    total = 0.0
    for i in range(M):
        r = base
        for a in range(M):
            if np.sqrt(np.sum(np.square(pos[a] - pos[i]))) < 2 * AGENT_SIZE:
                r -= 1
        total += r
    return total


def observation(pos, vel, lpos):
    """[M, 4 + 2 L + 4 (M - 1)]: own velocity, own position, landmarks and other agents relative to it, the others' (silent)
    communication (simple_spread.py:86-103)."""
    M = pos.shape[0]
    rows = []
    for i in range(M):
        ent = [lpos[l] - pos[i] for l in range(lpos.shape[0])]
        oth = [pos[j] - pos[i] for j in range(M) if j != i]
        comm = [np.zeros(2) for j in range(M) if j != i]
        rows.append(np.concatenate([vel[i], pos[i]] + ent + oth + comm))
    return np.stack(rows)


class SimpleSpreadRef:
    """N independent environments stepped one after the other (the reference's DummyVecEnv loop), explicit initial states."""

    def __init__(self, pos, vel, lpos, episode_length=25, tstep=0):
        self.pos, self.vel, self.lpos = [np.array(x, np.float64) for x in (pos, vel, lpos)]
        self.T, self.t = episode_length, np.full(self.pos.shape[0], tstep, np.int64)

    def obs(self):
        return np.stack([observation(self.pos[n], self.vel[n], self.lpos[n]) for n in range(self.pos.shape[0])])

    def step(self, actions_env, reset_states=None):
        """actions_env [N, M, 5].  reset_states(n) -> (pos, vel, lpos) for an environment whose episode ends (the reference
        draws them from NumPy's global generator).  Returns obs [N, M, OD], rewards [N, M, 1], dones [N, M]."""
        N, M = self.pos.shape[:2]
        rew = np.zeros((N, M, 1))
        dones = np.zeros((N, M), bool)
        for n in range(N):
            self.pos[n], self.vel[n] = world_step(self.pos[n], self.vel[n], actions_env[n])
            rew[n, :, 0] = reward(self.pos[n], self.lpos[n])
            self.t[n] += 1
            if self.t[n] >= self.T:
                dones[n] = True
                if reset_states is not None:
                    self.pos[n], self.vel[n], self.lpos[n] = reset_states(n)
                self.t[n] = 0
        return self.obs(), rew, dones
