// insert.hip — K1: the per-step slot writes of the MPE rollout (onpolicy/runner/shared/mpe_runner.py:125-139 +
// SharedReplayBuffer.insert, shared_buffer.py:96-112) in ONE launch.  The policy kernels already wrote actions,
// log-probs and values into slot `step` (R_MAPPOPolicy.collect_into); what remains are the environment's outputs:
//   obs[step+1]        <- obs                                   [N][M][D]
//   share_obs[step+1]  <- all agents' obs of the thread, repeated per agent (use_centralized_V, mpe_runner.py:133-135)
//                         or obs itself
//   rewards[step]      <- rewards                               [N][M]
//   masks[step+1]      <- 1 - done                              [N][M]
// Pure HBM traffic: 4*(M*D + M*S + 2M) bytes per rollout thread, one thread per share_obs element (the largest output).
#include "insert_core.h"

__global__ __launch_bounds__(256) void insert_mpe_kernel(InsertArgs p) { insert_mpe_body(p, blockIdx.x, gridDim.x); }

extern "C" int mappo_insert_mpe(const float *obs, int64_t obs_stride_n, int64_t obs_stride_m, const float *rewards,
                                int64_t rew_stride_n, int64_t rew_stride_m, const uint8_t *dones, int64_t done_stride_n,
                                int64_t done_stride_m, float *obs_dst, float *share_dst, float *rew_dst, float *mask_dst,
                                int32_t N, int32_t M, int32_t D, int32_t centralized, mappo_stream_t stream) {
  MAPPO_REQUIRE(obs && rewards && dones && obs_dst && share_dst && rew_dst && mask_dst && N > 0 && M > 0 && D > 0,
                "insert_mpe: bad arguments");
  InsertArgs a;
  a.obs = obs; a.obs_sn = obs_stride_n; a.obs_sm = obs_stride_m; a.rew = rewards; a.rew_sn = rew_stride_n; a.rew_sm = rew_stride_m;
  a.done = dones; a.done_sn = done_stride_n; a.done_sm = done_stride_m; a.obs_dst = obs_dst; a.share_dst = share_dst;
  a.rew_dst = rew_dst; a.mask_dst = mask_dst; a.N = N; a.M = M; a.D = D; a.centralized = centralized;
  const int64_t total = (int64_t)N * M * (centralized ? M * D : D);
  int64_t nb = (total + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(insert_mpe_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), a);
  MAPPO_CHECK_LAUNCH("insert_mpe");
  return MAPPO_OK;
}
