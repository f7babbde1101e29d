// synth_env.hip — the synthetic SMAC-shaped vec-env of the benchmarks (mappo_amd/envs/synthetic.py: SyntheticSMACEnv) as ONE
// launch per step instead of ~12 elementwise launches (which were 20 % of a config-3 / config-4 iteration).  Bench / test
// infrastructure, not part of the reference's hot path: the reference's environments are CPU StarCraft II processes
// (onpolicy/envs/starcraft2/StarCraft2_Env.py); this env only reproduces the SHAPES and the episode structure (agents that
// die, episodes that end) the runner and buffer paths react to.
//   obs [N][M][D], share_obs [N][M][S] ~ N(0, 1);  avail [N][M][A] in {0, 1} (action 0 always available, the others with
//   probability 0.7);  rewards [N] ~ N(0, 1) (shared by the agents of an env);  dead [N][M] state: dies with probability
//   p_death per step;  an env terminates with probability p_term per step (all agents done, restarts alive);
//   dones = dead | terminated.
// Counter-based Philox stream keyed by (seed, counter_dev[0]): a pure device op, capturable into the rollout hipGraph.  The
// launch advances the counter ITSELF: every workgroup reads it first and takes a ticket (counter_dev[1..33]) when it is done; the
// workgroup that draws the last ticket stores counter + 1 and clears the tickets — no second launch per step for a `ctr += 1`
// (that elementwise launch was 4.5 us of every rollout step).
#include "mlp_core.h"

struct SynthArgs {
  float *obs, *share, *avail, *rewards;
  uint8_t *dead, *dones;
  uint64_t *counter_dev;      // [34]: counter, group-of-groups tickets, 32 group tickets
  int N, M, D, S, A;
  float p_death, p_term;
  uint64_t seed;
};

__device__ __forceinline__ float synth_uniform(uint64_t seed, uint64_t ctr, uint64_t idx) {       // (0, 1]
  return ((float)(philox_u32(seed, ctr, idx) >> 8) + 1.0f) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float synth_normal(uint64_t seed, uint64_t ctr, uint64_t idx) {        // Box-Muller
  const float u1 = synth_uniform(seed, ctr, 2 * idx), u2 = synth_uniform(seed, ctr, 2 * idx + 1);
  return sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2);
}

__global__ __launch_bounds__(256) void synth_smac_step_kernel(SynthArgs p) {
  const uint64_t ctr = *p.counter_dev;
  const int W = p.D + p.S + p.A;
  const int64_t total = (int64_t)p.N * p.M * W;
  // index spaces of the stream: [0, total) per-element draws | then per-agent death draws | per-env termination | per-env reward
  const uint64_t base_agent = (uint64_t)total, base_env = base_agent + (uint64_t)p.N * p.M;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t ag = e / W;
    const int k = (int)(e - ag * W);
    if (k < p.D) p.obs[ag * p.D + k] = synth_normal(p.seed, ctr, (uint64_t)e);
    else if (k < p.D + p.S) p.share[ag * p.S + (k - p.D)] = synth_normal(p.seed, ctr, (uint64_t)e);
    else {
      const int a = k - p.D - p.S;
      p.avail[ag * p.A + a] = (a == 0 || synth_uniform(p.seed, ctr, 2 * (uint64_t)e) <= 0.7f) ? 1.f : 0.f;
    }
    if (k == 0) {
      const int64_t env = ag / p.M;
      const bool term = synth_uniform(p.seed, ctr, 2 * (base_env + (uint64_t)env)) <= p.p_term;
      const bool dead = p.dead[ag] != 0 || synth_uniform(p.seed, ctr, 2 * (base_agent + (uint64_t)ag)) <= p.p_death;
      p.dones[ag] = (dead || term) ? 1 : 0;
      p.dead[ag] = (dead && !term) ? 1 : 0;                    // a terminated env restarts with every agent alive
      if (ag - env * p.M == 0) p.rewards[env] = synth_normal(p.seed, ctr, base_env + (uint64_t)p.N + (uint64_t)env);
    }
  }
  // advance the stream: the last workgroup to finish (every workgroup has read the counter by then)
  // (two levels, 32 groups: a single ticket counter serialises one atomic per workgroup — 12 us for 4 096 of them)
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long *tk = reinterpret_cast<unsigned long long *>(p.counter_dev);
    const unsigned n_groups = gridDim.x < 32u ? gridDim.x : 32u, g = blockIdx.x % n_groups;
    const unsigned in_group = (gridDim.x - g + n_groups - 1u) / n_groups;       // workgroups b with b % n_groups == g
    if (atomicAdd(tk + 2 + g, 1ull) == (unsigned long long)in_group - 1ull) {
      tk[2 + g] = 0ull;
      if (atomicAdd(tk + 1, 1ull) == (unsigned long long)n_groups - 1ull) {
        tk[1] = 0ull;
        tk[0] = ctr + 1ull;
      }
    }
  }
}

// P consecutive steps in ONE launch (the runner's episode: like SyntheticMPEEnv, which draws an episode's block at its first step):
// outputs are [P][...] pools the env hands out step by step as views, so that a rollout step has no env launch at all.  The state
// part (agents die, envs terminate and restart) is sequential in the step: the thread that owns an agent's first element walks
// the P steps for it; everything else is independent per (step, element).  Step p uses Philox counter ctr + p.
__global__ __launch_bounds__(256) void synth_smac_pool_kernel(SynthArgs p, int P) {
  const uint64_t ctr0 = *p.counter_dev;
  const int W = p.D + p.S + p.A;
  const int64_t NM = (int64_t)p.N * p.M, total = NM * W;
  const uint64_t base_agent = (uint64_t)total, base_env = base_agent + (uint64_t)NM;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total * P; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t st = e / total, r = e - st * total;
    const uint64_t ctr = ctr0 + (uint64_t)st;
    const int64_t ag = r / W;
    const int k = (int)(r - ag * W);
    if (k < p.D) p.obs[(st * NM + ag) * p.D + k] = synth_normal(p.seed, ctr, (uint64_t)r);
    else if (k < p.D + p.S) p.share[(st * NM + ag) * p.S + (k - p.D)] = synth_normal(p.seed, ctr, (uint64_t)r);
    else {
      const int a = k - p.D - p.S;
      p.avail[(st * NM + ag) * p.A + a] = (a == 0 || synth_uniform(p.seed, ctr, 2 * (uint64_t)r) <= 0.7f) ? 1.f : 0.f;
    }
    if (k == 0 && st == 0) {                                     // this agent's state over the P steps
      const int64_t env = ag / p.M;
      bool dead = p.dead[ag] != 0;
      for (int s2 = 0; s2 < P; ++s2) {
        const uint64_t c2 = ctr0 + (uint64_t)s2;
        const bool term = synth_uniform(p.seed, c2, 2 * (base_env + (uint64_t)env)) <= p.p_term;
        dead = dead || synth_uniform(p.seed, c2, 2 * (base_agent + (uint64_t)ag)) <= p.p_death;
        p.dones[(int64_t)s2 * NM + ag] = (dead || term) ? 1 : 0;
        dead = dead && !term;
        if (ag - env * p.M == 0) p.rewards[(int64_t)s2 * p.N + env] = synth_normal(p.seed, c2, base_env + (uint64_t)p.N + (uint64_t)env);
      }
      p.dead[ag] = dead ? 1 : 0;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long *tk = reinterpret_cast<unsigned long long *>(p.counter_dev);
    const unsigned n_groups = gridDim.x < 32u ? gridDim.x : 32u, g = blockIdx.x % n_groups;
    const unsigned in_group = (gridDim.x - g + n_groups - 1u) / n_groups;
    if (atomicAdd(tk + 2 + g, 1ull) == (unsigned long long)in_group - 1ull) {
      tk[2 + g] = 0ull;
      if (atomicAdd(tk + 1, 1ull) == (unsigned long long)n_groups - 1ull) {
        tk[1] = 0ull;
        tk[0] = ctr0 + (unsigned long long)P;
      }
    }
  }
}

extern "C" int mappo_synth_smac_pool(float *obs, float *share_obs, float *avail, float *rewards, uint8_t *dead, uint8_t *dones,
                                     int32_t P, int32_t N, int32_t M, int32_t D, int32_t S, int32_t A, float p_death, float p_term,
                                     uint64_t seed, uint64_t *counter_dev, mappo_stream_t stream) {
  MAPPO_REQUIRE(obs && share_obs && avail && rewards && dead && dones && counter_dev, "synth_smac_pool: null argument");
  MAPPO_REQUIRE(P > 0 && N > 0 && M > 0 && D > 0 && S > 0 && A > 0, "synth_smac_pool: bad shape");
  SynthArgs p = {obs, share_obs, avail, rewards, dead, dones, counter_dev, N, M, D, S, A, p_death, p_term, seed};
  const int64_t total = (int64_t)P * N * M * (D + S + A);
  const int64_t blocks = (total + 255) / 256;
  hipLaunchKernelGGL(synth_smac_pool_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, p, (int)P);
  MAPPO_CHECK_LAUNCH("synth_smac_pool");
  return MAPPO_OK;
}

extern "C" int mappo_synth_smac_step(float *obs, float *share_obs, float *avail, float *rewards, uint8_t *dead, uint8_t *dones,
                                     int32_t N, int32_t M, int32_t D, int32_t S, int32_t A, float p_death, float p_term,
                                     uint64_t seed, uint64_t *counter_dev, mappo_stream_t stream) {
  MAPPO_REQUIRE(obs && share_obs && avail && rewards && dead && dones && counter_dev, "synth_smac_step: null argument");
  MAPPO_REQUIRE(N > 0 && M > 0 && D > 0 && S > 0 && A > 0, "synth_smac_step: bad shape");
  SynthArgs p = {obs, share_obs, avail, rewards, dead, dones, counter_dev, N, M, D, S, A, p_death, p_term, seed};
  const int64_t total = (int64_t)N * M * (D + S + A);
  const int64_t blocks = (total + 255) / 256;
  hipLaunchKernelGGL(synth_smac_step_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, p);
  MAPPO_CHECK_LAUNCH("synth_smac_step");
  return MAPPO_OK;
}
