// ppo_loss.hip — K5 (+K6): the fused PPO minibatch loss, forward and backward in one pass.
//   actor objective  : L_pi - entropy_coef * H      (r_mappo.py:124-141, act.py:154-160, distributions.py:14-28,64-68)
//   critic objective : value_loss_coef * L_V        (r_mappo.py:52-89,151-155; utils/util.py:23-29)
// Per sample the kernel reads A logits (+A availability flags), action, old log-prob, advantage, active flag,
// value, old value, return and writes A d(logits) and 1 d(value): 4*(3A+8) B (4*(2A+8) without avail).
//
// Mapping: one lane per sample (the softmax is a per-lane loop over A <= 32, no cross-lane traffic), 256
// samples per workgroup.  The [256][A] logits tile and the d(logits) tile are staged through LDS so that
// global memory only sees contiguous 16-B-per-lane traffic; the LDS tile also holds the masked logits
// between the passes (max, log-sum-exp, entropy, gradient), so no per-lane register array is indexed at
// run time.  Per-sample scalars (advantage, old log-prob, ...) are one value per lane and go straight to
// registers.  Statistics are reduced in double: lane -> wave shuffle -> block partial -> one-block final.
#include "common.h"
#include <float.h>

// exp / log on the hardware transcendental units (v_exp_f32 / v_log_f32, 1 ulp) as in the fused update kernels (mlp_upd16.h): the
// libm forms are ~20-25 instructions each, 2 A + 2 of them per sample — at configs[4] size (6.5 M samples) 45 % of this kernel's time
// was their arithmetic, on a kernel whose roofline is the HBM stream.  Arguments are differences from the running max / old
// log-probs (bounded), results feed sums and products of the same 1e-5 tolerance class.
__device__ __forceinline__ float pl_exp(float x) { return __builtin_amdgcn_exp2f(1.44269504088896341f * x); }
__device__ __forceinline__ float pl_log(float x) { return 0.693147180559945309f * __builtin_amdgcn_logf(x); }

#define PL_BLOCK 256
#define PL_MAX_BLOCKS 65535

struct PlArgs {
  const float *logits, *values;
  const int32_t *rows;
  const float *avail, *actions, *old_logp, *adv, *active, *v_old, *returns, *vn_state;
  const double *mb_moments;
  float *dlogits, *dvalues;
  double *partials;
  mappo_ppo_cfg cfg;
  int64_t B;
  int A;
};

__global__ __launch_bounds__(PL_BLOCK) void ppo_loss_kernel(PlArgs p) {
  extern __shared__ __align__(16) float tile[];   // [PL_BLOCK * A] logits / d(logits) | [PL_BLOCK * A] availability flags
  __shared__ double smem[16 * 4];
  const int A = p.A;
  const int tid = threadIdx.x;
  const VnStats vn = vn_stats(p.cfg.use_valuenorm ? p.vn_state : nullptr);
  // denominators come from mb_moments (global over all ranks in a data-parallel run), not from the local B
  const double sum_active = p.mb_moments[2];
  const float inv_B = (float)(1.0 / (p.mb_moments[3] > 0.0 ? p.mb_moments[3] : 1.0));
  const float inv_act = (float)(1.0 / (sum_active > 0.0 ? sum_active : 1.0));
  const float scale_pi = p.cfg.use_policy_active_masks ? inv_act : inv_B;
  const float scale_v = p.cfg.use_value_active_masks ? inv_act : inv_B;
  const float clip = p.cfg.clip_param;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};   // sum w*min(s1,s2), sum w*H, sum w_v*l, sum ratio

  const int64_t n_tiles = (p.B + PL_BLOCK - 1) / PL_BLOCK;
  for (int64_t tb = blockIdx.x; tb < n_tiles; tb += gridDim.x) {
    const int64_t base = tb * PL_BLOCK;
    const int n_here = (int)min((int64_t)PL_BLOCK, p.B - base);
    const int n_el = n_here * A;
    // ---- stage the logits tile and (whole-buffer minibatches) the availability tile: both are contiguous [n_here][A]
    // blocks, fetched as 16-byte-per-lane streams; read per lane as av[a] the flags cost A dword loads with a 4 A-byte lane
    // stride.  The per-sample scalars are issued before the barrier, so one memory latency covers the whole tile. ----
    float *atile = tile + PL_BLOCK * A;
    const bool av_lds = p.avail != nullptr && p.rows == nullptr;
    {
      const float *src = p.logits + base * A;
      const float *asrc = av_lds ? p.avail + base * A : src;
      if ((((uintptr_t)src) & 15) == 0 && (((uintptr_t)asrc) & 15) == 0) {
        const int n4 = n_el >> 2;
        for (int i = tid; i < n4; i += PL_BLOCK) {
          const float4 zv = reinterpret_cast<const float4 *>(src)[i];
          float4 avv = zv;
          if (av_lds) avv = reinterpret_cast<const float4 *>(asrc)[i];
          reinterpret_cast<float4 *>(tile)[i] = zv;
          if (av_lds) reinterpret_cast<float4 *>(atile)[i] = avv;
        }
        for (int i = (n4 << 2) + tid; i < n_el; i += PL_BLOCK) { tile[i] = src[i]; if (av_lds) atile[i] = asrc[i]; }
      } else {
        for (int i = tid; i < n_el; i += PL_BLOCK) { tile[i] = src[i]; if (av_lds) atile[i] = asrc[i]; }
      }
    }
    const bool mine = tid < n_here;
    const int64_t i = base + (mine ? tid : 0);
    const int64_t row = p.rows ? (int64_t)p.rows[i] : i;
    const float f_act = p.actions[row], old_lp = p.old_logp[row], adv = p.adv[row], active = p.active[row], v = p.values[i],
                vo = p.v_old[row], ret = p.returns[row];
    __syncthreads();
    if (mine) {
      float *z = tile + tid * A;
      const int act = (int)f_act;
      // ---- pass 1: availability mask (distributions.py:66-67) + max ----
      uint32_t dead = 0u;
      float zmax = -FLT_MAX;
      const float *av = p.avail ? (av_lds ? atile + tid * A : p.avail + row * A) : nullptr;
      for (int a = 0; a < A; ++a) {
        float za = z[a];
        if (av && av[a] == 0.f) { za = -1e10f; dead |= (1u << a); z[a] = za; }
        zmax = fmaxf(zmax, za);
      }
      // ---- pass 2: e_a = exp(z_a - max), kept in registers when A <= 8 (one exp per action instead of three) ----
      float ereg[8];
      const bool small = A <= 8;
      float se = 0.f;
#pragma unroll
      for (int a = 0; a < 8; ++a) { ereg[a] = 0.f; if (small && a < A) { ereg[a] = pl_exp(z[a] - zmax); se += ereg[a]; } }
      if (!small) for (int a = 0; a < A; ++a) se += pl_exp(z[a] - zmax);
      const float log_se = pl_log(se), inv_se = 1.0f / se;
      // ---- pass 3: entropy  H = -sum p * max(logp, finfo.min) ----
      float H = 0.f;
      if (small) {
#pragma unroll
        for (int a = 0; a < 8; ++a) if (a < A) { const float lp = (z[a] - zmax) - log_se; H -= ereg[a] * inv_se * fmaxf(lp, -FLT_MAX); }
      } else {
        for (int a = 0; a < A; ++a) { const float lp = (z[a] - zmax) - log_se; H -= pl_exp(lp) * fmaxf(lp, -FLT_MAX); }
      }
      // ---- policy surrogate (r_mappo.py:124-134) ----
      const float logp = (z[act] - zmax) - log_se;
      const float ratio = pl_exp(logp - old_lp);
      const float s1 = ratio * adv;
      const float s2 = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip) * adv;
      const float w_pi = p.cfg.use_policy_active_masks ? active : 1.f;
      const float dlogp = (s1 <= s2) ? -(w_pi * scale_pi) * adv * ratio : 0.f;
      const float ce = p.cfg.entropy_coef * w_pi * scale_pi;
      // ---- pass 4: d(objective)/d logits, written back into the tile ----
      if (small) {
#pragma unroll
        for (int a = 0; a < 8; ++a) if (a < A) {
          const float lp = (z[a] - zmax) - log_se, pa = ereg[a] * inv_se;
          float g = dlogp * ((a == act ? 1.f : 0.f) - pa) + ce * pa * (lp + H);
          if (dead & (1u << a)) g = 0.f;
          z[a] = g;
        }
      } else {
        for (int a = 0; a < A; ++a) {
          const float lp = (z[a] - zmax) - log_se, pa = pl_exp(lp);
          float g = dlogp * ((a == act ? 1.f : 0.f) - pa) + ce * pa * (lp + H);
          if (dead & (1u << a)) g = 0.f;          // overwritten logits get no gradient
          z[a] = g;
        }
      }
      // ---- value loss (r_mappo.py:62-87) ----
      const float tgt = p.cfg.use_valuenorm ? (ret - vn.mean) / vn.sd : ret;
      const float dvc = fminf(fmaxf(v - vo, -clip), clip);
      const float e_o = tgt - v, e_c = tgt - (vo + dvc);
      float l_o, l_c, g_o, g_c;
      if (p.cfg.use_huber_loss) {
        const float dl = p.cfg.huber_delta;
        const bool so = fabsf(e_o) <= dl, sc = fabsf(e_c) <= dl;
        l_o = so ? e_o * e_o * 0.5f : dl * (fabsf(e_o) - dl * 0.5f);
        l_c = sc ? e_c * e_c * 0.5f : dl * (fabsf(e_c) - dl * 0.5f);
        g_o = so ? e_o : copysignf(dl, e_o);
        g_c = sc ? e_c : copysignf(dl, e_c);
      } else {
        l_o = e_o * e_o * 0.5f; l_c = e_c * e_c * 0.5f; g_o = e_o; g_c = e_c;
      }
      float l, dv;
      if (p.cfg.use_clipped_value_loss) {
        const float inside = (fabsf(v - vo) <= clip) ? 1.f : 0.f;
        const float d_o = -g_o, d_c = -g_c * inside;
        l = fmaxf(l_o, l_c);
        dv = (l_o > l_c) ? d_o : ((l_c > l_o) ? d_c : 0.5f * (d_o + d_c));   // torch.max splits ties evenly
      } else {
        l = l_o; dv = -g_o;
      }
      const float w_v = p.cfg.use_value_active_masks ? active : 1.f;
      p.dvalues[i] = dv * (w_v * scale_v) * p.cfg.value_loss_coef;
      acc[0] += (double)(w_pi * fminf(s1, s2));
      acc[1] += (double)(w_pi * H);
      acc[2] += (double)(w_v * l);
      acc[3] += (double)ratio;
    }
    __syncthreads();
    // ---- write the d(logits) tile ----
    {
      float *dst = p.dlogits + base * A;
      if ((((uintptr_t)dst) & 15) == 0) {
        const int n4 = n_el >> 2;
        for (int i = tid; i < n4; i += PL_BLOCK) reinterpret_cast<float4 *>(dst)[i] = reinterpret_cast<const float4 *>(tile)[i];
        for (int i = (n4 << 2) + tid; i < n_el; i += PL_BLOCK) dst[i] = tile[i];
      } else {
        for (int i = tid; i < n_el; i += PL_BLOCK) dst[i] = tile[i];
      }
    }
    __syncthreads();
  }
  block_sum<4>(acc, smem);
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) p.partials[(size_t)blockIdx.x * 4 + k] = acc[k];
  }
}

__global__ __launch_bounds__(PL_BLOCK) void ppo_stats_kernel(const double *__restrict__ partials, int nblk,
                                                            const double *__restrict__ mb_moments, int64_t B,
                                                            int use_policy_active, int use_value_active,
                                                            double *__restrict__ stats) {
  __shared__ double smem[16 * 4];
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < nblk; b += blockDim.x) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += partials[(size_t)b * 4 + k];
  }
  block_sum<4>(v, smem);
  if (threadIdx.x == 0) {
    // local numerators over GLOBAL denominators: summing stats[0..3] over data-parallel ranks gives the
    // single-process value
    const double sa = mb_moments[2] > 0.0 ? mb_moments[2] : 1.0;
    const double Bg = mb_moments[3] > 0.0 ? mb_moments[3] : 1.0;
    const double den_pi = use_policy_active ? sa : Bg;
    const double den_v = use_value_active ? sa : Bg;
    stats[0] = v[2] / den_v;        // value_loss
    stats[1] = -v[0] / den_pi;      // policy_loss
    stats[2] = v[1] / den_pi;       // dist_entropy
    stats[3] = v[3] / Bg;           // imp_weights.mean()
    stats[4] = mb_moments[2];
    stats[5] = mb_moments[3];
  }
}

static inline int pl_blocks(int64_t B) {
  int64_t t = (B + PL_BLOCK - 1) / PL_BLOCK;
  if (t > 2048) t = 2048;            // ~8 workgroups per CU, grid-stride over the rest
  return (int)(t < 1 ? 1 : t);
}

extern "C" int64_t mappo_ppo_loss_workspace_bytes(int64_t B) { return (int64_t)2048 * 4 * sizeof(double); }

extern "C" int mappo_ppo_loss_fwd_bwd(const float *logits, const float *values, const int32_t *rows, const float *avail,
                                      const float *actions, const float *old_logp, const float *adv, const float *active,
                                      const float *v_old, const float *returns, const float *vn_state,
                                      const double *mb_moments, float *dlogits, float *dvalues, double *stats,
                                      void *workspace, const mappo_ppo_cfg *cfg, int64_t B, int32_t A,
                                      mappo_stream_t stream) {
  MAPPO_REQUIRE(B > 0 && A >= 1 && A <= MAPPO_MAX_ACTIONS, "ppo_loss: B=%lld A=%d unsupported", (long long)B, A);
  MAPPO_REQUIRE(logits && values && actions && old_logp && adv && active && v_old && returns && mb_moments && dlogits &&
                    dvalues && stats && workspace && cfg,
                "ppo_loss: null pointer");
  MAPPO_REQUIRE(!cfg->use_valuenorm || vn_state, "ppo_loss: use_valuenorm needs vn_state");
  PlArgs p;
  p.logits = logits; p.values = values; p.rows = rows; p.avail = avail; p.actions = actions; p.old_logp = old_logp;
  p.adv = adv; p.active = active; p.v_old = v_old; p.returns = returns; p.vn_state = vn_state; p.mb_moments = mb_moments;
  p.dlogits = dlogits; p.dvalues = dvalues; p.partials = (double *)workspace; p.cfg = *cfg; p.B = B; p.A = A;
  const int nblk = pl_blocks(B);
  // logits / d(logits) tile, + the availability tile only when the kernel stages it (avail given, rows streamed in place):
  // same predicate as `av_lds` in the kernel — without it the second tile only halved the occupancy
  const bool av_lds = avail != nullptr && rows == nullptr;
  const size_t lds = (size_t)(av_lds ? 2 : 1) * PL_BLOCK * A * sizeof(float);      // <= 64 KiB at A = MAPPO_MAX_ACTIONS
  static const hipError_t attr_rc = hipFuncSetAttribute((const void *)ppo_loss_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        (int)(2 * PL_BLOCK * MAPPO_MAX_ACTIONS * sizeof(float)));
  if (attr_rc != hipSuccess) { mappo_set_error("ppo_loss: hipFuncSetAttribute: %s", hipGetErrorString(attr_rc)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  PROF_LAUNCH(MAPPO_PROF_PPO_LOSS, ppo_loss_kernel, dim3(nblk), dim3(PL_BLOCK), lds, as_stream(stream), p);
  hipLaunchKernelGGL(ppo_stats_kernel, dim3(1), dim3(PL_BLOCK), 0, as_stream(stream), (const double *)workspace, nblk,
                     mb_moments, B, cfg->use_policy_active_masks, cfg->use_value_active_masks, stats);
  MAPPO_CHECK_LAUNCH("ppo_loss_fwd_bwd");
  return MAPPO_OK;
}
