// stats.hip — K3 (advantage build + normalisation, r_mappo.py:174-182), K12 (ValueNorm update,
// valuenorm.py:37-54) and the minibatch denominators the fused loss needs (r_mappo.py:84,130-134).
// All reductions: per-thread double accumulation -> wave shuffle -> per-block partial -> one-block final
// pass, i.e. deterministic (no atomics) and far inside the 1e-5 budget for fp32 statistics.
#include "common.h"

#define STAT_BLOCK 256
#define STAT_MAX_BLOCKS 1024

static inline int stat_blocks(int64_t n) {
  int64_t b = (n + STAT_BLOCK * 4 - 1) / (STAT_BLOCK * 4);
  if (b < 1) b = 1;
  if (b > STAT_MAX_BLOCKS) b = STAT_MAX_BLOCKS;
  return (int)b;
}

// out[0..NV) = sum of the per-block partials; out[NV] = `extra` when extra >= 0 (the minibatch size B)
template <int NV>
__global__ __launch_bounds__(STAT_BLOCK) void final_reduce_kernel(const double *__restrict__ partials, int nblk,
                                                                 double *__restrict__ out, double extra) {
  __shared__ double smem[16 * NV];
  double v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = 0.0;
  for (int b = threadIdx.x; b < nblk; b += blockDim.x) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] += partials[(size_t)b * NV + i];
  }
  block_sum<NV>(v, smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) out[i] = v[i];
    if (extra >= 0.0) out[NV] = extra;
  }
}

// ---- advantages -------------------------------------------------------------------------------------
__global__ __launch_bounds__(STAT_BLOCK) void adv_moments_kernel(const float *__restrict__ returns,
                                                                const float *__restrict__ value_preds,
                                                                const float *__restrict__ active,
                                                                const float *__restrict__ vn_state,
                                                                float *__restrict__ adv, double *__restrict__ partials,
                                                                int64_t n) {
  __shared__ double smem[16 * 3];
  const VnStats vn = vn_stats(vn_state);
  double v[3] = {0.0, 0.0, 0.0};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float a = returns[i] - (value_preds[i] * vn.sd + vn.mean);   // r_mappo.py:174-177
    adv[i] = a;
    if (active[i] != 0.f) {                                             // :178-181 (nanmean / nanstd)
      v[0] += (double)a;
      v[1] += (double)a * (double)a;
      v[2] += 1.0;
    }
  }
  block_sum<3>(v, smem);
  if (threadIdx.x == 0) {
    partials[blockIdx.x * 3 + 0] = v[0];
    partials[blockIdx.x * 3 + 1] = v[1];
    partials[blockIdx.x * 3 + 2] = v[2];
  }
}

__global__ __launch_bounds__(STAT_BLOCK) void adv_normalize_kernel(float *__restrict__ adv,
                                                                  const double *__restrict__ moments, int64_t n) {
  const double cnt = moments[2] > 0.0 ? moments[2] : 1.0;
  const double mean_d = moments[0] / cnt;
  double var_d = moments[1] / cnt - mean_d * mean_d;
  if (var_d < 0.0) var_d = 0.0;
  const float mean = (float)mean_d;
  const float denom = (float)sqrt(var_d) + 1e-5f;                       // r_mappo.py:182
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    adv[i] = (adv[i] - mean) / denom;
}

extern "C" int64_t mappo_adv_workspace_bytes(int64_t n) { return (int64_t)STAT_MAX_BLOCKS * 3 * sizeof(double); }

extern "C" int mappo_adv_moments(const float *returns, const float *value_preds, const float *active_masks,
                                 const float *vn_state, float *adv, double *moments, void *workspace, int64_t n,
                                 mappo_stream_t stream) {
  MAPPO_REQUIRE(n > 0 && returns && value_preds && active_masks && adv && moments && workspace,
                "adv_moments: bad arguments (n=%lld)", (long long)n);
  const int nblk = stat_blocks(n);
  hipLaunchKernelGGL(adv_moments_kernel, dim3(nblk), dim3(STAT_BLOCK), 0, as_stream(stream), returns, value_preds,
                     active_masks, vn_state, adv, (double *)workspace, n);
  hipLaunchKernelGGL(final_reduce_kernel<3>, dim3(1), dim3(STAT_BLOCK), 0, as_stream(stream),
                     (const double *)workspace, nblk, moments, -1.0);
  MAPPO_CHECK_LAUNCH("adv_moments");
  return MAPPO_OK;
}

extern "C" int mappo_adv_normalize(float *adv, const double *moments, int64_t n, mappo_stream_t stream) {
  MAPPO_REQUIRE(n > 0 && adv && moments, "adv_normalize: bad arguments");
  hipLaunchKernelGGL(adv_normalize_kernel, dim3(stat_blocks(n)), dim3(STAT_BLOCK), 0, as_stream(stream), adv,
                     moments, n);
  MAPPO_CHECK_LAUNCH("adv_normalize");
  return MAPPO_OK;
}

// ---- minibatch moments + ValueNorm --------------------------------------------------------------------
__global__ __launch_bounds__(STAT_BLOCK) void minibatch_moments_kernel(const float *__restrict__ returns,
                                                                      const float *__restrict__ active,
                                                                      const int32_t *__restrict__ rows, int64_t B,
                                                                      double *__restrict__ partials) {
  __shared__ double smem[16 * 3];
  double v[3] = {0.0, 0.0, 0.0};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += stride) {
    const int64_t row = rows ? (int64_t)rows[i] : i;
    const float r = returns[row];
    v[0] += (double)r;
    v[1] += (double)r * (double)r;
    v[2] += (double)active[row];
  }
  block_sum<3>(v, smem);
  if (threadIdx.x == 0) {
    partials[blockIdx.x * 3 + 0] = v[0];
    partials[blockIdx.x * 3 + 1] = v[1];
    partials[blockIdx.x * 3 + 2] = v[2];
  }
}

extern "C" int64_t mappo_moments_workspace_bytes(int64_t B) { return (int64_t)STAT_MAX_BLOCKS * 3 * sizeof(double); }

extern "C" int mappo_minibatch_moments(const float *returns, const float *active_masks, const int32_t *rows, int64_t B,
                                       double *mb_moments, void *workspace, mappo_stream_t stream) {
  MAPPO_REQUIRE(B > 0 && returns && active_masks && mb_moments && workspace, "minibatch_moments: bad arguments");
  const int nblk = stat_blocks(B);
  hipLaunchKernelGGL(minibatch_moments_kernel, dim3(nblk), dim3(STAT_BLOCK), 0, as_stream(stream), returns,
                     active_masks, rows, B, (double *)workspace);
  hipLaunchKernelGGL(final_reduce_kernel<3>, dim3(1), dim3(STAT_BLOCK), 0, as_stream(stream),
                     (const double *)workspace, nblk, mb_moments, (double)B);
  MAPPO_CHECK_LAUNCH("minibatch_moments");
  return MAPPO_OK;
}

// valuenorm.py:37-54: running <- running*w + batch*(1-w); each product/sum rounded to fp32 like torch.
__global__ void valuenorm_update_kernel(float *vn_state, const double *mb_moments, float w, float omw) {
  const double B = mb_moments[3] > 0.0 ? mb_moments[3] : 1.0;
  const float bm = (float)(mb_moments[0] / B);
  const float bsq = (float)(mb_moments[1] / B);
  vn_state[0] = __fadd_rn(__fmul_rn(vn_state[0], w), __fmul_rn(bm, omw));
  vn_state[1] = __fadd_rn(__fmul_rn(vn_state[1], w), __fmul_rn(bsq, omw));
  vn_state[2] = __fadd_rn(__fmul_rn(vn_state[2], w), omw);
}

extern "C" int mappo_valuenorm_update(float *vn_state, const double *mb_moments, double beta, mappo_stream_t stream) {
  MAPPO_REQUIRE(vn_state && mb_moments, "valuenorm_update: null pointer");
  // (1 - beta) is formed in double like the Python expression `1.0 - weight` (valuenorm.py:52-54) and only
  // then rounded to fp32, the precision torch multiplies the fp32 tensors with.
  hipLaunchKernelGGL(valuenorm_update_kernel, dim3(1), dim3(1), 0, as_stream(stream), vn_state, mb_moments,
                     (float)beta, (float)(1.0 - beta));
  MAPPO_CHECK_LAUNCH("valuenorm_update");
  return MAPPO_OK;
}

__global__ void valuenorm_update_n_kernel(float *vn_state, const double *mb_moments, float w, float omw, int n, float *states_out) {
  const double B = mb_moments[3] > 0.0 ? mb_moments[3] : 1.0;
  const float bm = (float)(mb_moments[0] / B);
  const float bsq = (float)(mb_moments[1] / B);
  float s0 = vn_state[0], s1 = vn_state[1], s2 = vn_state[2];
  for (int e = 0; e < n; ++e) {
    s0 = __fadd_rn(__fmul_rn(s0, w), __fmul_rn(bm, omw));
    s1 = __fadd_rn(__fmul_rn(s1, w), __fmul_rn(bsq, omw));
    s2 = __fadd_rn(__fmul_rn(s2, w), omw);
    states_out[3 * e + 0] = s0; states_out[3 * e + 1] = s1; states_out[3 * e + 2] = s2;
  }
  vn_state[0] = s0; vn_state[1] = s1; vn_state[2] = s2;
}

extern "C" int mappo_valuenorm_update_n(float *vn_state, const double *mb_moments, double beta, int32_t n, float *states_out,
                                        mappo_stream_t stream) {
  MAPPO_REQUIRE(vn_state && mb_moments && states_out && n >= 1, "valuenorm_update_n: bad arguments");
  hipLaunchKernelGGL(valuenorm_update_n_kernel, dim3(1), dim3(1), 0, as_stream(stream), vn_state, mb_moments, (float)beta,
                     (float)(1.0 - beta), (int)n, states_out);
  MAPPO_CHECK_LAUNCH("valuenorm_update_n");
  return MAPPO_OK;
}
