// insert_core.h — the MPE rollout insert (see insert.hip) as a device function, shared with the fused rollout-step kernel.
#pragma once
#include "common.h"

struct InsertArgs {
  const float *obs;  int64_t obs_sn, obs_sm;       // element (n, m, d) at obs[n*obs_sn + m*obs_sm + d]
  const float *rew;  int64_t rew_sn, rew_sm;       // element (n, m)    at rew[n*rew_sn + m*rew_sm]   (0 strides broadcast)
  const uint8_t *done; int64_t done_sn, done_sm;   // bool bytes
  float *obs_dst, *share_dst, *rew_dst, *mask_dst; // contiguous slots
  int N, M, D, centralized;
};

// workgroup `bid` of `nb` cooperating 256-thread workgroups
__device__ __forceinline__ void insert_mpe_body(const InsertArgs &p, int bid, int nb) {
  const int S = p.centralized ? p.M * p.D : p.D;
  const int64_t total = (int64_t)p.N * p.M * S;
  for (int64_t e = (int64_t)bid * blockDim.x + threadIdx.x; e < total; e += (int64_t)nb * blockDim.x) {
    const int64_t nm = e / S;
    const int j = (int)(e - nm * S);
    const int n = (int)(nm / p.M), m = (int)(nm - (int64_t)n * p.M);
    const int ms = p.centralized ? j / p.D : m, d = p.centralized ? j - ms * p.D : j;      // source agent / feature
    const float v = p.obs[n * p.obs_sn + ms * p.obs_sm + d];
    p.share_dst[e] = v;
    if (!p.centralized || ms == m) p.obs_dst[nm * p.D + d] = v;                            // each obs element exactly once
    if (j == 0) {
      p.rew_dst[nm] = p.rew[n * p.rew_sn + m * p.rew_sm];
      p.mask_dst[nm] = p.done[n * p.done_sn + m * p.done_sm] ? 0.f : 1.f;
    }
  }
}
