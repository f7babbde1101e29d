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

// ---- the SMAC rollout insert (insert.hip: mappo_insert_smac) as a device function, shared with the fused recurrent rollout step ----
struct SmacInsert {
  const float *obs, *share, *avail;              // contiguous [N*M][D | S | A]
  const float *rew; int64_t rew_sn, rew_sm;
  const uint8_t *done; int64_t done_sn, done_sm;
  const uint8_t *bad;                            // contiguous [N*M] bool bytes or NULL (no bad transitions)
  const float *h_a, *h_c;                        // contiguous [N*M][H] or NULL
  float *obs_dst, *share_dst, *avail_dst, *rew_dst, *mask_dst, *bad_dst, *active_dst, *ha_dst, *hc_dst;
  int N, M, D, S, A, H;
};
__device__ __forceinline__ bool env_done(const SmacInsert &p, int n) {
  bool all = true;
  for (int m = 0; m < p.M; ++m) all = all && p.done[n * p.done_sn + m * p.done_sm] != 0;
  return all;
}
// workgroup `bid` of `nb` cooperating workgroups
__device__ __forceinline__ void insert_smac_body(const SmacInsert &p, int bid, int nb) {
  const int64_t tid = (int64_t)bid * blockDim.x + threadIdx.x, nthr = (int64_t)nb * blockDim.x;
  const int64_t R = (int64_t)p.N * p.M;
  for (int64_t e = tid; e < R * p.D; e += nthr) p.obs_dst[e] = p.obs[e];
  for (int64_t e = tid; e < R * p.S; e += nthr) p.share_dst[e] = p.share[e];
  if (p.avail)
    for (int64_t e = tid; e < R * p.A; e += nthr) p.avail_dst[e] = p.avail[e];
  for (int64_t e = tid; e < R; e += nthr) {
    const int n = (int)(e / p.M), m = (int)(e - (int64_t)n * p.M);
    const bool de = env_done(p, n), d = p.done[n * p.done_sn + m * p.done_sm] != 0;
    p.rew_dst[e] = p.rew[n * p.rew_sn + m * p.rew_sm];
    p.mask_dst[e] = de ? 0.f : 1.f;
    p.active_dst[e] = de ? 1.f : (d ? 0.f : 1.f);
    p.bad_dst[e] = (p.bad && p.bad[e]) ? 0.f : 1.f;
  }
  if (p.h_a) {
    const int h4 = p.H >> 2;
    for (int64_t e = tid; e < R * h4; e += nthr) {
      const int n = (int)((e / h4) / p.M);
      const float keep = env_done(p, n) ? 0.f : 1.f;
      float4 a = reinterpret_cast<const float4 *>(p.h_a)[e], c = reinterpret_cast<const float4 *>(p.h_c)[e];
      a.x *= keep; a.y *= keep; a.z *= keep; a.w *= keep;
      c.x *= keep; c.y *= keep; c.z *= keep; c.w *= keep;
      reinterpret_cast<float4 *>(p.ha_dst)[e] = a;
      reinterpret_cast<float4 *>(p.hc_dst)[e] = c;
    }
  }
}

