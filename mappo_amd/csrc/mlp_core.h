// mlp_core.h — device code shared by mlp.hip (MLP trunk/head kernels) and gru.hip (recurrent layer kernels):
// the MFMA accumulator-layout helpers, the flat parameter offsets, the counter-based RNG and the per-sample loss heads.
#pragma once
#include "common.h"
#include <float.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define HID 64
#define TS 32          // samples per wave tile
#define TP 33          // tile row stride (floats)
#define WP 65          // hidden-weight row stride (floats)
#define HP 33          // head-weight row stride
#define MAXD 64
#define LN_EPS 1e-5f

struct NetOff {
  int fn_w, fn_b, w1, b1, ln1_w, ln1_b;
  int w2[MAPPO_MAX_LAYER_N], b2[MAPPO_MAX_LAYER_N], ln2_w[MAPPO_MAX_LAYER_N], ln2_b[MAPPO_MAX_LAYER_N];
  int gru_wih, gru_whh, gru_bih, gru_bhh, rn_w, rn_b;
  int wh, bh, total;
};

__host__ __device__ inline NetOff net_offsets(const mappo_net_desc &d) {
  NetOff o;
  int p = 0;
  const int D = d.in_dim, H = d.hidden;
  o.fn_w = o.fn_b = -1;
  if (d.use_feature_norm) { o.fn_w = p; p += D; o.fn_b = p; p += D; }
  o.w1 = p; p += H * D; o.b1 = p; p += H; o.ln1_w = p; p += H; o.ln1_b = p; p += H;
  for (int l = 0; l < MAPPO_MAX_LAYER_N; ++l) {
    o.w2[l] = o.b2[l] = o.ln2_w[l] = o.ln2_b[l] = -1;
    if (l < d.layer_N) { o.w2[l] = p; p += H * H; o.b2[l] = p; p += H; o.ln2_w[l] = p; p += H; o.ln2_b[l] = p; p += H; }
  }
  o.gru_wih = o.gru_whh = o.gru_bih = o.gru_bhh = o.rn_w = o.rn_b = -1;
  if (d.recurrent) {
    o.gru_wih = p; p += 3 * H * H; o.gru_whh = p; p += 3 * H * H; o.gru_bih = p; p += 3 * H; o.gru_bhh = p; p += 3 * H;
    o.rn_w = p; p += H; o.rn_b = p; p += H;
  }
  o.wh = p; p += d.out_dim * H; o.bh = p; p += d.out_dim;
  o.total = p;
  return o;
}


__host__ __device__ inline int al4(int p) { return (p + 3) & ~3; }

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_sync() {
  // The tiles are private to one wavefront and the LDS executes a wave's DS instructions in order; what has to
  // be prevented is the COMPILER moving a tile read above the tile write that produced it.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// v(lane) + v(lane ^ 32): one v_permlane32_swap (gfx950) instead of a ds_bpermute round trip through the LDS crossbar.
// With vdst = src = v the swap leaves {lo, lo} in one result and {hi, hi} in the other, so their sum is lo + hi in every lane.
__device__ __forceinline__ float xhalf_sum(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// v(lane) + v(lane ^ 16)
__device__ __forceinline__ float xrow_sum(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// fire-and-forget LDS float add (ds_add_f32): used where exactly one wave adds into a location per phase, so the
// result does not depend on arrival order
__device__ __forceinline__ void lds_add(float *p, float v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// row (feature within a 32-row MFMA tile) held by accumulator register `reg` of lane-half `half`
#define ROWMAP(reg, half) (((reg) & 3) + 8 * ((reg) >> 2) + 4 * (half))

template <bool RELU>
__device__ __forceinline__ float act_fwd(float z) { return RELU ? fmaxf(z, 0.f) : tanhf(z); }

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// 4 consecutive features (reg&3 = 0..3) of a per-feature vector in LDS, as one 16-byte read
__device__ __forceinline__ float4 vec4_of(const float *sV, int t, int q, int half) {
  return *reinterpret_cast<const float4 *>(sV + 32 * t + 8 * q + 4 * half);
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 (counter-based RNG for action sampling)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t philox_u32(uint64_t seed, uint64_t counter, uint64_t index) {
  uint32_t c0 = (uint32_t)index, c1 = (uint32_t)(index >> 32), c2 = (uint32_t)counter, c3 = (uint32_t)(counter >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c0;
}

// ------------------------------------------------------------------------------------------------
// per-sample PPO loss heads (r_mappo.py:52-89,124-141; act.py:154-160; distributions.py:64-68) — the arithmetic
// of ppo_loss.hip, shared by the fused update kernels of mlp.hip and gru.hip.  One lane = one sample.
// ------------------------------------------------------------------------------------------------
struct LossScales { float scale_pi, scale_v, vn_mean, vn_sd; };

__device__ __forceinline__ LossScales loss_scales(const mappo_ppo_cfg &cfg, const double *mb_moments, const float *vn_state) {
  LossScales ls;
  const double sa = mb_moments[2], Bg = mb_moments[3];
  const float inv_act = (float)(1.0 / (sa > 0.0 ? sa : 1.0)), inv_B = (float)(1.0 / (Bg > 0.0 ? Bg : 1.0));
  ls.scale_pi = cfg.use_policy_active_masks ? inv_act : inv_B;
  ls.scale_v = cfg.use_value_active_masks ? inv_act : inv_B;
  const VnStats vn = vn_stats(cfg.use_valuenorm ? vn_state : nullptr);
  ls.vn_mean = vn.mean; ls.vn_sd = vn.sd;
  return ls;
}

// Register form of the actor loss for A <= 8 (Discrete spaces of up to 8 actions — MPE's 5): z[0..A) logits in, the
// gradient d(actor objective)/d logits out in g; fully unrolled, same expressions and evaluation order as the general
// path of actor_loss_lane.  lacc[0..2] += w*min(s1,s2), w*H, ratio.
__device__ __forceinline__ void actor_loss_regs(float (&z)[8], float (&g)[8], int A, uint32_t dead, int act, float old_lp, float adv,
                                                float active, const mappo_ppo_cfg &cfg, float scale_pi, double (&lacc)[4]) {
  const float clip8 = cfg.clip_param;
  float e[8];
  float zmax8 = -FLT_MAX;
#pragma unroll
  for (int a = 0; a < 8; ++a)
    if (a < A) {
      if (dead & (1u << a)) z[a] = -1e10f;
      zmax8 = fmaxf(zmax8, z[a]);
    }
  float se8 = 0.f;
#pragma unroll
  for (int a = 0; a < 8; ++a) { e[a] = 0.f; if (a < A) { e[a] = expf(z[a] - zmax8); se8 += e[a]; } }
  const float log_se8 = logf(se8), inv_se8 = 1.0f / se8;
  float H8 = 0.f, z_act = z[0];
#pragma unroll
  for (int a = 0; a < 8; ++a)
    if (a < A) {
      const float l_ = (z[a] - zmax8) - log_se8;
      H8 -= (e[a] * inv_se8) * fmaxf(l_, -FLT_MAX);
      if (a == act) z_act = z[a];
    }
  const float logp8 = (z_act - zmax8) - log_se8;
  const float ratio8 = expf(logp8 - old_lp);
  const float s18 = ratio8 * adv, s28 = fminf(fmaxf(ratio8, 1.f - clip8), 1.f + clip8) * adv;
  const float w8 = cfg.use_policy_active_masks ? active : 1.f;
  const float dlogp8 = (s18 <= s28) ? -(w8 * scale_pi) * adv * ratio8 : 0.f;
  const float ce8 = cfg.entropy_coef * w8 * scale_pi;
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    g[a] = 0.f;
    if (a < A) {
      const float l_ = (z[a] - zmax8) - log_se8;
      const float pa = e[a] * inv_se8;
      float gg = dlogp8 * ((a == act ? 1.f : 0.f) - pa) + ce8 * pa * (l_ + H8);
      if (dead & (1u << a)) gg = 0.f;
      g[a] = gg;
    }
  }
  lacc[0] += (double)(w8 * fminf(s18, s28));
  lacc[1] += (double)(w8 * H8);
  lacc[2] += (double)ratio8;
}

// zl[0..A): logits of this sample (row of a [s][TP] LDS tile, columns 16..31 free when A <= 16); on return zl holds
// d(actor objective)/d logits.  lacc[0..2] += w*min(s1,s2), w*H, ratio.
__device__ __forceinline__ void actor_loss_lane(float *zl, int A, uint32_t dead, int act, float old_lp, float adv, float active,
                                                const mappo_ppo_cfg &cfg, float scale_pi, double (&lacc)[4]) {
  if (A <= 8) {
    float z[8], g[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) z[a] = zl[a < A ? a : 0];
    actor_loss_regs(z, g, A, dead, act, old_lp, adv, active, cfg, scale_pi, lacc);
#pragma unroll
    for (int a = 0; a < 8; ++a) if (a < A) zl[a] = g[a];
    return;
  }
  const float clip = cfg.clip_param;
  float zmax = -FLT_MAX;
  for (int a = 0; a < A; ++a) {
    float za = zl[a];
    if (dead & (1u << a)) { za = -1e10f; zl[a] = za; }
    zmax = fmaxf(zmax, za);
  }
  // e_a = exp(z_a - max) is computed once and parked behind the logits
  const bool park = A <= 16;
  float se = 0.f;
  for (int a = 0; a < A; ++a) { const float e = expf(zl[a] - zmax); if (park) zl[16 + a] = e; se += e; }
  const float log_se = logf(se), inv_se = 1.0f / se;
  float Hent = 0.f;
  for (int a = 0; a < A; ++a) {
    const float l_ = (zl[a] - zmax) - log_se;
    const float pa = (park ? zl[16 + a] : expf(zl[a] - zmax)) * inv_se;
    Hent -= pa * fmaxf(l_, -FLT_MAX);
  }
  const float logp = (zl[act] - zmax) - log_se;
  const float ratio = expf(logp - old_lp);
  const float s1 = ratio * adv, s2 = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip) * adv;
  const float w_pi = cfg.use_policy_active_masks ? active : 1.f;
  const float dlogp = (s1 <= s2) ? -(w_pi * scale_pi) * adv * ratio : 0.f;
  const float ce = cfg.entropy_coef * w_pi * scale_pi;
  for (int a = 0; a < A; ++a) {
    const float l_ = (zl[a] - zmax) - log_se;
    const float pa = (park ? zl[16 + a] : expf(zl[a] - zmax)) * inv_se;
    float g = dlogp * ((a == act ? 1.f : 0.f) - pa) + ce * pa * (l_ + Hent);
    if (dead & (1u << a)) g = 0.f;
    zl[a] = g;
  }
  lacc[0] += (double)(w_pi * fminf(s1, s2));
  lacc[1] += (double)(w_pi * Hent);
  lacc[2] += (double)ratio;
}

// returns d(value_loss_coef * value loss)/d v ; lacc[0] += w_v * l
__device__ __forceinline__ float critic_loss_lane(float v, float vo, float ret, float active, const mappo_ppo_cfg &cfg,
                                                  const LossScales &ls, double (&lacc)[4]) {
  const float clip = cfg.clip_param;
  const float tgt = cfg.use_valuenorm ? (ret - ls.vn_mean) / ls.vn_sd : ret;
  const float dvc = fminf(fmaxf(v - vo, -clip), clip);
  const float e_o = tgt - v, e_c = tgt - (vo + dvc);
  float l_o, l_c, g_o, g_c;
  if (cfg.use_huber_loss) {
    const float dl = cfg.huber_delta;
    const bool so = fabsf(e_o) <= dl, sc = fabsf(e_c) <= dl;
    l_o = so ? e_o * e_o * 0.5f : dl * (fabsf(e_o) - dl * 0.5f);
    l_c = sc ? e_c * e_c * 0.5f : dl * (fabsf(e_c) - dl * 0.5f);
    g_o = so ? e_o : copysignf(dl, e_o);
    g_c = sc ? e_c : copysignf(dl, e_c);
  } else {
    l_o = e_o * e_o * 0.5f; l_c = e_c * e_c * 0.5f; g_o = e_o; g_c = e_c;
  }
  float l, dv;
  if (cfg.use_clipped_value_loss) {
    const float inside = (fabsf(v - vo) <= clip) ? 1.f : 0.f;
    const float d_o = -g_o, d_c = -g_c * inside;
    l = fmaxf(l_o, l_c);
    dv = (l_o > l_c) ? d_o : ((l_c > l_o) ? d_c : 0.5f * (d_o + d_c));   // torch.max splits ties evenly
  } else {
    l = l_o; dv = -g_o;
  }
  const float w_v = cfg.use_value_active_masks ? active : 1.f;
  lacc[0] += (double)(w_v * l);
  return dv * (w_v * ls.scale_v) * cfg.value_loss_coef;
}

// categorical epilogue of get_actions (distributions.py:14-28,64-68): mask, argmax | inverse-CDF sample, log-prob
// bit a set <=> available_actions[a] == 0 (A <= 32).  Batches of eight unconditional clamped loads: read one per trip inside
// the softmax loop, every action cost a memory round trip at the very end of a rollout step.
__device__ __forceinline__ uint32_t avail_dead_mask(const float *__restrict__ av, int A) {
  uint32_t dead = 0u;
  for (int a0 = 0; a0 < A; a0 += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = av[min(a0 + j, A - 1)];
#pragma unroll
    for (int j = 0; j < 8; ++j) dead |= ((a0 + j < A && v[j] == 0.f) ? 1u : 0u) << ((a0 + j) & 31);
  }
  return dead;
}

__device__ __forceinline__ void categorical_act_mask(float *zl, int A, uint32_t dead, bool deterministic, uint64_t seed,
                                                     uint64_t ctr, uint64_t index, float &action, float &logp);

__device__ __forceinline__ void categorical_act_lane(float *zl, int A, const float *av, bool deterministic, uint64_t seed,
                                                     uint64_t ctr, uint64_t index, float &action, float &logp) {
  categorical_act_mask(zl, A, av ? avail_dead_mask(av, A) : 0u, deterministic, seed, ctr, index, action, logp);
}

// exp / log here run on the hardware transcendental units (v_exp_f32 / v_log_f32, 1 ulp): this epilogue is the serial tail of every
// rollout step (one lane per row walks the actions), and the libm forms are ~20-25 instructions per call, 2 A + 1 of them
__device__ __forceinline__ float hw_exp(float x) { return __builtin_amdgcn_exp2f(1.44269504088896341f * x); }
__device__ __forceinline__ float hw_log(float x) { return 0.693147180559945309f * __builtin_amdgcn_logf(x); }
__device__ __forceinline__ void categorical_act_mask(float *zl, int A, uint32_t dead, bool deterministic, uint64_t seed,
                                                     uint64_t ctr, uint64_t index, float &action, float &logp) {
  float zmax = -FLT_MAX;
  for (int a = 0; a < A; ++a) {
    float za = zl[a];
    if ((dead >> a) & 1u) { za = -1e10f; zl[a] = za; }
    zmax = fmaxf(zmax, za);
  }
  float se = 0.f;
  for (int a = 0; a < A; ++a) se += hw_exp(zl[a] - zmax);
  const float lse = zmax + hw_log(se);
  int chosen = 0;
  if (deterministic) {
    float best = -FLT_MAX;                         // probs.argmax: first maximum
    for (int a = 0; a < A; ++a) { if (zl[a] > best) { best = zl[a]; chosen = a; } }
  } else {
    const float u = (float)(philox_u32(seed, ctr, index) >> 8) * (1.0f / 16777216.0f);
    float c = 0.f;
    bool found = false;
    for (int a = 0; a < A; ++a) {
      const float pa = hw_exp(zl[a] - lse);
      c += pa;
      if (!found && pa > 0.f) chosen = a;        // fallback: last action with support
      if (!found && u < c) { chosen = a; found = true; }
    }
  }
  action = (float)chosen;
  logp = zl[chosen] - lse;
}

// head output (accumulator layout) -> tZ[s][a]
__device__ __forceinline__ void head_to_tile(float *tZ, const f32x16 &z, int A, int l31, int half) {
  // rows a >= A go to column 32 of the row (the tile's padding column): an address select instead of 16 exec-mask regions
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int a = ROWMAP(r, half);
    tZ[l31 * TP + (a < A ? a : TP - 1)] = z[r];
  }
}
