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

// The same insert for recurrent policies: additionally rnn_states / rnn_states_critic of slot step+1 = the states the
// networks just returned, zeroed where the episode ended (mpe_runner.py:126-128) — one launch instead of the mask cast,
// two products and two slot copies.
struct RnnInsert {
  const float *h_a, *h_c;       // [N*M][H] contiguous
  float *dst_a, *dst_c;
  int H;                        // recurrent_N * hidden_size, multiple of 4
};
__global__ __launch_bounds__(256) void insert_mpe_rnn_kernel(InsertArgs p, RnnInsert r) {
  insert_mpe_body(p, blockIdx.x, gridDim.x);
  const int h4 = r.H >> 2;
  const int64_t total = (int64_t)p.N * p.M * h4;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t nm = e / h4;
    const int n = (int)(nm / p.M), m = (int)(nm - (int64_t)n * p.M);
    const float keep = p.done[n * p.done_sn + m * p.done_sm] ? 0.f : 1.f;
    float4 a = reinterpret_cast<const float4 *>(r.h_a)[e], c = reinterpret_cast<const float4 *>(r.h_c)[e];
    a.x *= keep; a.y *= keep; a.z *= keep; a.w *= keep;
    c.x *= keep; c.y *= keep; c.z *= keep; c.w *= keep;
    reinterpret_cast<float4 *>(r.dst_a)[e] = a;
    reinterpret_cast<float4 *>(r.dst_c)[e] = c;
  }
}

extern "C" int mappo_insert_mpe_rnn(const float *obs, int64_t obs_stride_n, int64_t obs_stride_m, const float *rewards,
                                    int64_t rew_stride_n, int64_t rew_stride_m, const uint8_t *dones, int64_t done_stride_n,
                                    int64_t done_stride_m, float *obs_dst, float *share_dst, float *rew_dst, float *mask_dst,
                                    int32_t N, int32_t M, int32_t D, int32_t centralized, const float *rnn_states,
                                    const float *rnn_states_critic, float *rnn_dst, float *rnn_critic_dst, int32_t H,
                                    mappo_stream_t stream) {
  MAPPO_REQUIRE(obs && rewards && dones && obs_dst && share_dst && rew_dst && mask_dst && N > 0 && M > 0 && D > 0,
                "insert_mpe_rnn: bad arguments");
  MAPPO_REQUIRE(rnn_states && rnn_states_critic && rnn_dst && rnn_critic_dst && H > 0 && (H & 3) == 0, "insert_mpe_rnn: bad state arguments");
  MAPPO_REQUIRE(((((uintptr_t)rnn_states) | ((uintptr_t)rnn_states_critic) | ((uintptr_t)rnn_dst) | ((uintptr_t)rnn_critic_dst)) & 15) == 0,
                "insert_mpe_rnn: state arrays must be 16-byte aligned");
  InsertArgs a;
  a.obs = obs; a.obs_sn = obs_stride_n; a.obs_sm = obs_stride_m; a.rew = rewards; a.rew_sn = rew_stride_n; a.rew_sm = rew_stride_m;
  a.done = dones; a.done_sn = done_stride_n; a.done_sm = done_stride_m; a.obs_dst = obs_dst; a.share_dst = share_dst;
  a.rew_dst = rew_dst; a.mask_dst = mask_dst; a.N = N; a.M = M; a.D = D; a.centralized = centralized;
  RnnInsert r;
  r.h_a = rnn_states; r.h_c = rnn_states_critic; r.dst_a = rnn_dst; r.dst_c = rnn_critic_dst; r.H = H;
  const int64_t t1 = (int64_t)N * M * (centralized ? M * D : D), t2 = (int64_t)N * M * (H >> 2);
  int64_t nb = ((t1 > t2 ? t1 : t2) + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(insert_mpe_rnn_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), a, r);
  MAPPO_CHECK_LAUNCH("insert_mpe_rnn");
  return MAPPO_OK;
}

// The SMAC rollout insert (smac_runner.py:129-151 + shared_buffer.py:74-112) in one launch.  With dones_env[n] = all_m dones[n][m]:
//   masks = 1 - dones_env (per env, every agent) ; active_masks = dones_env ? 1 : 1 - dones ; bad_masks = 1 - bad_transition ;
//   rnn_states / rnn_states_critic = states * (1 - dones_env) ; obs, share_obs, available_actions, rewards copied to their slots.
__global__ __launch_bounds__(256) void insert_smac_kernel(SmacInsert p) { insert_smac_body(p, blockIdx.x, gridDim.x); }

extern "C" int mappo_insert_smac(const float *obs, const float *share_obs, const float *avail, const float *rewards, int64_t rew_stride_n,
                                 int64_t rew_stride_m, const uint8_t *dones, int64_t done_stride_n, int64_t done_stride_m,
                                 const uint8_t *bad_transition, const float *rnn_states, const float *rnn_states_critic, float *obs_dst,
                                 float *share_dst, float *avail_dst, float *rew_dst, float *mask_dst, float *bad_mask_dst,
                                 float *active_mask_dst, float *rnn_dst, float *rnn_critic_dst, int32_t N, int32_t M, int32_t D,
                                 int32_t S, int32_t A, int32_t H, mappo_stream_t stream) {
  MAPPO_REQUIRE(obs && share_obs && rewards && dones && obs_dst && share_dst && rew_dst && mask_dst && bad_mask_dst && active_mask_dst &&
                N > 0 && M > 0 && D > 0 && S > 0, "insert_smac: bad arguments");
  MAPPO_REQUIRE(!avail || (avail_dst && A > 0), "insert_smac: available_actions slot");
  if (rnn_states) {
    MAPPO_REQUIRE(rnn_states_critic && rnn_dst && rnn_critic_dst && H > 0 && (H & 3) == 0, "insert_smac: bad state arguments");
    MAPPO_REQUIRE(((((uintptr_t)rnn_states) | ((uintptr_t)rnn_states_critic) | ((uintptr_t)rnn_dst) | ((uintptr_t)rnn_critic_dst)) & 15) == 0,
                  "insert_smac: state arrays must be 16-byte aligned");
  }
  SmacInsert a;
  a.obs = obs; a.share = share_obs; a.avail = avail; a.rew = rewards; a.rew_sn = rew_stride_n; a.rew_sm = rew_stride_m;
  a.done = dones; a.done_sn = done_stride_n; a.done_sm = done_stride_m; a.bad = bad_transition; a.h_a = rnn_states; a.h_c = rnn_states_critic;
  a.obs_dst = obs_dst; a.share_dst = share_dst; a.avail_dst = avail_dst; a.rew_dst = rew_dst; a.mask_dst = mask_dst; a.bad_dst = bad_mask_dst;
  a.active_dst = active_mask_dst; a.ha_dst = rnn_dst; a.hc_dst = rnn_critic_dst; a.N = N; a.M = M; a.D = D; a.S = S; a.A = A; a.H = H;
  int64_t most = (int64_t)N * M * (D > S ? D : S);
  int64_t nb = (most + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(insert_smac_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), a);
  MAPPO_CHECK_LAUNCH("insert_smac");
  return MAPPO_OK;
}

// recurrent_generator's index arithmetic (shared_buffer.py:385-494) for ALL ppo epochs of a train() call in one launch.
// perm[e] is a permutation of the data chunks; minibatch k of epoch e takes chunks c = perm[e][k*mbs + j] and stacks them
// time-major: flat position q = c*L + l of the reference's (n, m, t) order is buffer row (q % T)*R + q / T.
struct RecRows {
  const int64_t *perm;          // [E][chunks]
  int32_t *rows, *h0;           // [E][nmb][L*mbs], [E][nmb][mbs]
  int64_t chunks;
  int E, L, T, R, nmb, mbs;
};
__global__ __launch_bounds__(256) void recurrent_rows_kernel(RecRows p) {
  const int64_t per_e = (int64_t)p.nmb * p.mbs, total = (int64_t)p.E * per_e * p.L;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = i % p.mbs;
    const int l = (int)((i / p.mbs) % p.L);
    const int64_t ek = i / ((int64_t)p.mbs * p.L);                 // e * nmb + k
    const int64_t e = ek / p.nmb, k = ek - e * p.nmb;
    const int64_t c = p.perm[e * p.chunks + k * p.mbs + j];
    const int64_t q = c * p.L + l;
    const int32_t row = (int32_t)((q % p.T) * p.R + q / p.T);
    p.rows[i] = row;                                               // [(e*nmb + k)][l*mbs + j]
    if (l == 0) p.h0[ek * p.mbs + j] = row;
  }
}

extern "C" int mappo_recurrent_rows(const int64_t *perm, int32_t n_epochs, int64_t data_chunks, int32_t L, int32_t T, int32_t R,
                                    int32_t num_mini_batch, int32_t *rows, int32_t *h0_rows, mappo_stream_t stream) {
  MAPPO_REQUIRE(perm && rows && h0_rows && n_epochs > 0 && data_chunks > 0 && L > 0 && T > 0 && R > 0 && num_mini_batch > 0,
                "recurrent_rows: bad arguments");
  MAPPO_REQUIRE(data_chunks / num_mini_batch > 0, "recurrent_rows: fewer chunks than minibatches");
  MAPPO_REQUIRE((int64_t)T * R < ((int64_t)1 << 31), "recurrent_rows: buffer rows exceed int32");
  RecRows a;
  a.perm = perm; a.rows = rows; a.h0 = h0_rows; a.chunks = data_chunks; a.E = n_epochs; a.L = L; a.T = T; a.R = R;
  a.nmb = num_mini_batch; a.mbs = (int)(data_chunks / num_mini_batch);
  const int64_t total = (int64_t)n_epochs * a.nmb * a.mbs * L;
  int64_t nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(recurrent_rows_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), a);
  MAPPO_CHECK_LAUNCH("recurrent_rows");
  return MAPPO_OK;
}

// K1 (after_update, shared_buffer.py:114-131): up to 16 independent device copies in ONE launch (the reference copies
// slot T of eight arrays back to slot 0; as separate copies each is a launch of a few microseconds).
#define COPY_MAX 16
struct CopyBatch {
  float *dst[COPY_MAX];
  const float *src[COPY_MAX];
  int64_t n[COPY_MAX];          // floats
  int count;
};
__global__ __launch_bounds__(256) void copy_batch_kernel(CopyBatch c) {
  for (int j = 0; j < c.count; ++j) {
    const float *__restrict__ s = c.src[j];
    float *__restrict__ d = c.dst[j];
    const int64_t n = c.n[j];
    if (((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0) {
      const int64_t n4 = n >> 2;
      for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        reinterpret_cast<float4 *>(d)[i] = reinterpret_cast<const float4 *>(s)[i];
      for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = s[i];
    } else {
      for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = s[i];
    }
  }
}

extern "C" int mappo_copy_batch(int32_t count, float *const *dst, const float *const *src, const int64_t *n_floats,
                                mappo_stream_t stream) {
  MAPPO_REQUIRE(count >= 1 && count <= COPY_MAX && dst && src && n_floats, "copy_batch: bad arguments (count=%d)", count);
  CopyBatch c;
  c.count = count;
  int64_t total = 0;
  for (int j = 0; j < count; ++j) {
    MAPPO_REQUIRE(dst[j] && src[j] && n_floats[j] >= 0, "copy_batch: entry %d", j);
    c.dst[j] = dst[j]; c.src[j] = src[j]; c.n[j] = n_floats[j];
    total += n_floats[j];
  }
  int64_t nb = (total / 4 + 255) / 256;
  if (nb < 1) nb = 1;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(copy_batch_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), c);
  MAPPO_CHECK_LAUNCH("copy_batch");
  return MAPPO_OK;
}
