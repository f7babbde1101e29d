// mpe_env.hip — GPU-vectorised MPE `simple_spread`: N environments x M agents x L landmarks stepped by ONE
// kernel launch, one lane per environment.  Reference: onpolicy/envs/mpe/core.py:207-322 (World.step: action forces,
// pairwise soft-collision force, damping + integration), onpolicy/envs/mpe/scenarios/simple_spread.py:32-103 (reset, reward,
// observation), onpolicy/envs/mpe/environment.py:117-256 (action decoding, shared reward, time-limit done) and the reset-on-
// done of the vec-env wrappers (envs/env_wrappers.py:146-152,676-682).
//
// The reference computes in float64 NumPy; so does this kernel (an environment is ~200 flops per step: the launch is bound
// by its 20 + 72 M bytes of state / output per environment, not by the fp64 rate), so trajectories agree with the oracle to
// rounding of the transcendental functions and the fp32 cast of the outputs.  State (positions, velocities, landmarks, step
// and episode counters) is device resident; resets draw from a counter-based Philox stream keyed by (seed, episode, env), so
// the step is a pure device op and an episode of rollout steps + env steps can be captured into one hipGraph.
#include "mlp_core.h"

#define MPE_MAX_M 8
#define MPE_MAX_L 8

struct MpeArgs {
  double *apos, *avel, *lpos;      // [N][M][2], [N][M][2], [N][L][2]
  int32_t *tstep;                  // [N] steps since the last reset
  int64_t *episode;                // [N] resets so far (Philox counter)
  const float *actions;            // mode 0: one-hot / probabilities [N][M][5] (actions_env of the reference) | mode 1: index [N][M]
  float *obs, *rewards;            // [N][M][4 + 2 L + 4 (M - 1)], [N][M]
  uint8_t *dones;                  // [N][M] bool bytes
  int N, M, L, T, mode;
  uint64_t seed;
};

__device__ __forceinline__ double mpe_uniform(uint64_t seed, uint64_t ctr, uint64_t idx) {       // U(-1, 1)
  const uint32_t hi = philox_u32(seed, ctr, 2 * idx), lo = philox_u32(seed, ctr, 2 * idx + 1);
  const double u = ((double)hi * 4294967296.0 + (double)lo + 0.5) * (1.0 / 18446744073709551616.0);
  return 2.0 * u - 1.0;
}

// scenario.reset_world (simple_spread.py:32-47): agents U(-1,1)^2 at rest, landmarks 0.8 U(-1,1)^2
__device__ __forceinline__ void mpe_reset_env(const MpeArgs &p, int n, double (&ap)[MPE_MAX_M][2], double (&av)[MPE_MAX_M][2],
                                              double (&lp)[MPE_MAX_L][2], int64_t ep) {
  const uint64_t base = (uint64_t)n * (2 * (MPE_MAX_M + MPE_MAX_L));
  for (int i = 0; i < p.M; ++i) {
    ap[i][0] = mpe_uniform(p.seed, (uint64_t)ep, base + 2 * i);
    ap[i][1] = mpe_uniform(p.seed, (uint64_t)ep, base + 2 * i + 1);
    av[i][0] = av[i][1] = 0.0;
  }
  for (int l = 0; l < p.L; ++l) {
    lp[l][0] = 0.8 * mpe_uniform(p.seed, (uint64_t)ep, base + 2 * MPE_MAX_M + 2 * l);
    lp[l][1] = 0.8 * mpe_uniform(p.seed, (uint64_t)ep, base + 2 * MPE_MAX_M + 2 * l + 1);
  }
}

// scenario.observation (simple_spread.py:86-103): [vel, pos, landmarks - pos, others - pos, comm of the others (zeros: silent)]
__device__ __forceinline__ void mpe_write_obs(const MpeArgs &p, int n, const double (&ap)[MPE_MAX_M][2], const double (&av)[MPE_MAX_M][2],
                                              const double (&lp)[MPE_MAX_L][2]) {
  const int OD = 4 + 2 * p.L + 4 * (p.M - 1);
  for (int i = 0; i < p.M; ++i) {
    float *o = p.obs + ((size_t)n * p.M + i) * OD;
    o[0] = (float)av[i][0]; o[1] = (float)av[i][1]; o[2] = (float)ap[i][0]; o[3] = (float)ap[i][1];
    int k = 4;
    for (int l = 0; l < p.L; ++l) { o[k++] = (float)(lp[l][0] - ap[i][0]); o[k++] = (float)(lp[l][1] - ap[i][1]); }
    for (int j = 0; j < p.M; ++j)
      if (j != i) { o[k++] = (float)(ap[j][0] - ap[i][0]); o[k++] = (float)(ap[j][1] - ap[i][1]); }
    for (int j = 0; j < 2 * (p.M - 1); ++j) o[k++] = 0.f;
  }
}

__device__ __forceinline__ void mpe_load(const MpeArgs &p, int n, double (&ap)[MPE_MAX_M][2], double (&av)[MPE_MAX_M][2],
                                         double (&lp)[MPE_MAX_L][2]) {
  for (int i = 0; i < p.M; ++i) {
    ap[i][0] = p.apos[((size_t)n * p.M + i) * 2]; ap[i][1] = p.apos[((size_t)n * p.M + i) * 2 + 1];
    av[i][0] = p.avel[((size_t)n * p.M + i) * 2]; av[i][1] = p.avel[((size_t)n * p.M + i) * 2 + 1];
  }
  for (int l = 0; l < p.L; ++l) { lp[l][0] = p.lpos[((size_t)n * p.L + l) * 2]; lp[l][1] = p.lpos[((size_t)n * p.L + l) * 2 + 1]; }
}

__device__ __forceinline__ void mpe_store(const MpeArgs &p, int n, const double (&ap)[MPE_MAX_M][2], const double (&av)[MPE_MAX_M][2],
                                          const double (&lp)[MPE_MAX_L][2], bool landmarks) {
  for (int i = 0; i < p.M; ++i) {
    p.apos[((size_t)n * p.M + i) * 2] = ap[i][0]; p.apos[((size_t)n * p.M + i) * 2 + 1] = ap[i][1];
    p.avel[((size_t)n * p.M + i) * 2] = av[i][0]; p.avel[((size_t)n * p.M + i) * 2 + 1] = av[i][1];
  }
  if (landmarks)
    for (int l = 0; l < p.L; ++l) { p.lpos[((size_t)n * p.L + l) * 2] = lp[l][0]; p.lpos[((size_t)n * p.L + l) * 2 + 1] = lp[l][1]; }
}

__global__ __launch_bounds__(256) void mpe_spread_reset_kernel(MpeArgs p) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= p.N) return;
  double ap[MPE_MAX_M][2], av[MPE_MAX_M][2], lp[MPE_MAX_L][2];
  const int64_t ep = p.episode[n] + 1;
  mpe_reset_env(p, n, ap, av, lp, ep);
  p.episode[n] = ep;
  p.tstep[n] = 0;
  mpe_store(p, n, ap, av, lp, true);
  mpe_write_obs(p, n, ap, av, lp);
}

__global__ __launch_bounds__(256) void mpe_spread_step_kernel(MpeArgs p) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= p.N) return;
  const int M = p.M, L = p.L;
  double ap[MPE_MAX_M][2], av[MPE_MAX_M][2], lp[MPE_MAX_L][2], f[MPE_MAX_M][2];
  mpe_load(p, n, ap, av, lp);
  // ---- action -> force (environment.py:200-245: u = [a1 - a2, a3 - a4] * sensitivity 5; core.py:227-236: mass 1, no noise) ----
  for (int i = 0; i < M; ++i) {
    double u0 = 0.0, u1 = 0.0;
    if (p.mode == 0) {
      const float *a = p.actions + ((size_t)n * M + i) * 5;
      u0 = (double)a[1] - (double)a[2]; u1 = (double)a[3] - (double)a[4];
    } else {
      const int a = (int)p.actions[(size_t)n * M + i];
      u0 = a == 1 ? 1.0 : (a == 2 ? -1.0 : 0.0);                    // the one-hot of index a through the line above
      u1 = a == 3 ? 1.0 : (a == 4 ? -1.0 : 0.0);
    }
    f[i][0] = 5.0 * u0; f[i][1] = 5.0 * u1;
  }
  // ---- pairwise soft-collision forces between agents (core.py:238-262,283-322; landmarks do not collide) ----
  for (int a = 0; a < M; ++a)
    for (int b = a + 1; b < M; ++b) {
      const double dx = ap[a][0] - ap[b][0], dy = ap[a][1] - ap[b][1];
      const double dist = sqrt(dx * dx + dy * dy);
      const double k = 1e-3, x = -(dist - 0.3) / k;                       // dist_min = 0.15 + 0.15, contact_margin 1e-3
      const double pen = (x > 0.0 ? x + log1p(exp(-x)) : log1p(exp(x))) * k;     // np.logaddexp(0, x) * k
      const double fx = 1e2 * dx / dist * pen, fy = 1e2 * dy / dist * pen;       // contact_force 1e2
      f[a][0] = fx + f[a][0]; f[a][1] = fy + f[a][1];
      f[b][0] = -fx + f[b][0]; f[b][1] = -fy + f[b][1];
    }
  // ---- integrate (core.py:264-275: damping 0.25, dt 0.1, no max_speed) ----
  for (int i = 0; i < M; ++i) {
    av[i][0] = av[i][0] * (1.0 - 0.25); av[i][1] = av[i][1] * (1.0 - 0.25);
    av[i][0] += f[i][0] * 0.1; av[i][1] += f[i][1] * 0.1;
    ap[i][0] += av[i][0] * 0.1; ap[i][1] += av[i][1] * 0.1;
  }
  // ---- reward (simple_spread.py:73-84, shared: environment.py:139-143) ----
  double base = 0.0;
  for (int l = 0; l < L; ++l) {
    double dmin = 1e300;
    for (int a = 0; a < M; ++a) {
      const double dx = ap[a][0] - lp[l][0], dy = ap[a][1] - lp[l][1];
      dmin = fmin(dmin, sqrt(dx * dx + dy * dy));
    }
    base -= dmin;
  }
  double total = 0.0;
  for (int i = 0; i < M; ++i) {
    double r = base;
    for (int a = 0; a < M; ++a) {                                   // includes a == i (distance 0 < 0.3), as the reference does
      const double dx = ap[a][0] - ap[i][0], dy = ap[a][1] - ap[i][1];
      if (sqrt(dx * dx + dy * dy) < 0.3) r -= 1.0;
    }
    total += r;
  }
  const int t = p.tstep[n] + 1;
  const bool done = t >= p.T;                                       // environment.py:179-185
  for (int i = 0; i < M; ++i) {
    p.rewards[(size_t)n * M + i] = (float)total;
    p.dones[(size_t)n * M + i] = done ? 1 : 0;
  }
  if (done) {                                                       // vec-env wrappers: the returned obs are the reset obs
    const int64_t ep = p.episode[n] + 1;
    mpe_reset_env(p, n, ap, av, lp, ep);
    p.episode[n] = ep;
    p.tstep[n] = 0;
  } else {
    p.tstep[n] = t;
  }
  mpe_store(p, n, ap, av, lp, done);
  mpe_write_obs(p, n, ap, av, lp);
}

static int mpe_check(int N, int M, int L, const char *who) {
  MAPPO_REQUIRE(N > 0 && M >= 1 && M <= MPE_MAX_M && L >= 1 && L <= MPE_MAX_L, "%s: N=%d M=%d L=%d unsupported (M, L <= %d)", who, N, M, L,
                MPE_MAX_M);
  return MAPPO_OK;
}

extern "C" int mappo_mpe_spread_reset(double *agent_pos, double *agent_vel, double *landmark_pos, int32_t *tstep, int64_t *episode,
                                      float *obs, int32_t N, int32_t M, int32_t L, uint64_t seed, mappo_stream_t stream) {
  if (int rc = mpe_check(N, M, L, "mpe_spread_reset")) return rc;
  MAPPO_REQUIRE(agent_pos && agent_vel && landmark_pos && tstep && episode && obs, "mpe_spread_reset: null pointer");
  MpeArgs p = {};
  p.apos = agent_pos; p.avel = agent_vel; p.lpos = landmark_pos; p.tstep = tstep; p.episode = episode; p.obs = obs;
  p.N = N; p.M = M; p.L = L; p.seed = seed;
  hipLaunchKernelGGL(mpe_spread_reset_kernel, dim3((N + 255) / 256), dim3(256), 0, as_stream(stream), p);
  MAPPO_CHECK_LAUNCH("mpe_spread_reset");
  return MAPPO_OK;
}

extern "C" int mappo_mpe_spread_step(double *agent_pos, double *agent_vel, double *landmark_pos, int32_t *tstep, int64_t *episode,
                                     const float *actions, int32_t action_mode, float *obs, float *rewards, uint8_t *dones, int32_t N,
                                     int32_t M, int32_t L, int32_t episode_length, uint64_t seed, mappo_stream_t stream) {
  if (int rc = mpe_check(N, M, L, "mpe_spread_step")) return rc;
  MAPPO_REQUIRE(agent_pos && agent_vel && landmark_pos && tstep && episode && actions && obs && rewards && dones,
                "mpe_spread_step: null pointer");
  MAPPO_REQUIRE(action_mode == 0 || action_mode == 1, "mpe_spread_step: action_mode %d", action_mode);
  MpeArgs p = {};
  p.apos = agent_pos; p.avel = agent_vel; p.lpos = landmark_pos; p.tstep = tstep; p.episode = episode; p.actions = actions;
  p.obs = obs; p.rewards = rewards; p.dones = dones; p.N = N; p.M = M; p.L = L; p.T = episode_length; p.mode = action_mode; p.seed = seed;
  hipLaunchKernelGGL(mpe_spread_step_kernel, dim3((N + 255) / 256), dim3(256), 0, as_stream(stream), p);
  MAPPO_CHECK_LAUNCH("mpe_spread_step");
  return MAPPO_OK;
}
