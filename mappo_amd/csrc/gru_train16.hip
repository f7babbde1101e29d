// gru_train16.hip — the TRAINING pass of the recurrent layer (onpolicy/algorithms/utils/rnn.py:25-79 inside
// r_mappo.py:91-164's evaluate_actions + loss + backward) on v_mfma_f32_16x16x4_f32, one wavefront per 16-SEQUENCE tile.
//
// Round-2 state (gru.hip: gru_gi / gru_fwd_train2 / gru_head_bwd / gru_cell_bwd2 / gru_dx / gru_wgrad, 32x32x2 tiles, feature-major
// scratch): six launches per network whose recurrences ran at 0.12-0.14 of the fp32 MFMA rate — every per-step value went
// through HBM as 4-byte accesses 128 B apart, the input-side products had their own launches and scratch, and the two waves of a
// tile met at a barrier every step.  Here:
//
//   gru16_fwd_kernel    per step t of a tile: gi = W_ih x_t AND gh = W_hh (h mask) in ONE accumulator set (the input products are
//                       independent of the recurrence: they fill the dependent chain's bubbles), gates, h_t; then — still in
//                       registers — rnn.norm, the head, the PPO / value loss and their backward, so that what leaves the wave is
//                       d h_t (without the recurrent term) and the gate values the cell backward needs.  gi and h_t never
//                       reach HBM; head / rnn.norm gradients and the loss sums are reduced per workgroup.
//   gru16_bwd_kernel    reverse time: d gates from (d h_t + carry); carry = (W_hh^T d gh + d h z) mask AND d x_t = W_ih^T d gi
//                       from the same registers (again independent work beside the dependent chain); d gates / d x overwrite the
//                       forward's slots in place.
//   gru16_wgrad_kernel  dW_ih, dW_hh, db_ih, db_hh = sum over rows of (d gates)^T (x | h mask): row-tile GEMMs, 4 roles per tile.
//
// Layouts.  16x16x4 accumulator layout: lane (n = lane & 15, q = lane >> 4) holds features 16 b + 4 q + i (b, i = 0..3) of
// sequence n; the reduction order over k is free, so k-step (b, i) takes k = 16 b + 4 q + i and a lane's own registers are
// the B operand (mlp_upd16.h).  Weights sit in LDS in FRAGMENT order: block (bo, b) of a row-major [G][K] matrix is 256 floats,
// lane (n, q) owning M[16 bo + n][16 b + 4 q + 0..3] at float offset 4 * lane — the A operand of four MFMAs as ONE 16-byte read
// at base + 16 * lane bytes: every 16-lane group of a ds_read_b128 covers all 64 banks exactly once (a row-major copy with
// stride 68 puts lanes (11, q) and (12, q - 1) of a group on the same bank quad: one conflict cycle in five).  Scratch is
// BLOCKED the same way: [component][t][tile][b][lane][4] — a wave instruction moves one contiguous KiB.
#include <stdlib.h>
#define MLP_TU_GRU16
#include "mlp_impl.h"

#define G16_THREADS 512
#define G16_WAVES (G16_THREADS / WAVE)
#define G16_NG 192
#define G16_COMPS 6
#define C_HM 0      // h_{t-1} * mask_t                           (backward: read by the weight-gradient kernel)
#define C_R 1       // r          -> d pre_r  (backward, in place)
#define C_Z 2       // z          -> d pre_z
#define C_N 3       // n          -> d pre_n  (= d gi_n)
#define C_GHN 4     // W_hn h + b_hn -> d gh_n (= d pre_n * r)
#define C_DH 5      // d h_t without the recurrent term (forward) -> d x_t (backward, blocked x only)

__device__ __forceinline__ float sigmoid16(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x)); }
__device__ __forceinline__ float tanh16(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x)); }

struct Gru16Args {
  const float *params;
  NetOff off;
  const float *x;             // trunk features: blocked [L][n_ct][4][256] (x_blocked) | feature-major [64][L * Nc]
  int x_blocked;
  const float *h0;            // [.][64] row-major initial states
  const int32_t *h0_rows;     // [Nc] or NULL (identity)
  const float *masks;         // buffer order, indexed by rows[t * Nc + c] (NULL rows: identity)
  const int32_t *rows;
  int L, Nc, A;
  float *scratch;             // [6][L][n_ct][4][256]
  const float *avail, *actions, *old_logp, *adv, *active, *v_old, *returns, *vn_state;
  const double *mb_moments;
  mappo_ppo_cfg cfg;
  float *dxT;                 // backward: feature-major d x [64][L * Nc], or NULL: blocked, in scratch component C_DH
  float *slabs;
  int64_t slab_stride, slab_col0;
  double *partials;           // [grid][4]
};

// ---- staging -----------------------------------------------------------------------------------------------------------------
// dst block (bo, b) <- M[16 bo + n][16 b + 4 q + i], M row-major [G][K] (rows >= g_valid read as zero); one 16-byte global
// load -> one 16-byte LDS store; all loads of a batch are issued before the first store
template <int NB>
__device__ __forceinline__ void stage_frag(float *dst, const float *__restrict__ src, int G, int K, int g_valid) {
  const int k4 = K >> 2, n4 = G * k4, kb = K >> 4, nthr = blockDim.x;
  for (int e0 = threadIdx.x; e0 < n4; e0 += NB * nthr) {
    float4 v[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int e = min(e0 + j * nthr, n4 - 1), g = e / k4;
      v[j] = reinterpret_cast<const float4 *>(src)[min(g, g_valid - 1) * k4 + (e - g * k4)];
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int e = e0 + j * nthr;
      if (e < n4) {
        const int g = e / k4, kq = e - g * k4;
        const float4 t = g < g_valid ? v[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(dst + (((g >> 4) * kb + (kq >> 2)) * 64 + (kq & 3) * 16 + (g & 15)) * 4) = t;
      }
    }
  }
}
// transposed copy: dst block (bo, b) <- M[16 b + 4 q + i][16 bo + n]  (out block over K, k-steps over G)
template <int NB>
__device__ __forceinline__ void stage_frag_T(float *dst, const float *__restrict__ src, int G, int K) {
  const int k4 = K >> 2, n4 = G * k4, gb = G >> 4, nthr = blockDim.x;
  for (int e0 = threadIdx.x; e0 < n4; e0 += NB * nthr) {
    float4 v[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) v[j] = reinterpret_cast<const float4 *>(src)[min(e0 + j * nthr, n4 - 1)];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int e = e0 + j * nthr;
      if (e < n4) {
        const int g = e / k4, kq = e - g * k4;
        const float vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
        float *d = dst + (g >> 4) * 256 + ((g >> 2) & 3) * 64 + (g & 3);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = 4 * kq + r;
          d[((k >> 4) * gb) * 256 + (k & 15) * 4] = vv[r];
        }
      }
    }
  }
}

// acc[bo] += M[16 (bo0 + bo) + n][16 b + 4 q + i] * v[b][i] over all 16 k-steps: FOUR output blocks at a time, so that four
// independent accumulators sit between two MFMAs of one chain (dependent latency 40 cycles against 32 of issue)
template <int KB>
__device__ __forceinline__ void frag_mma4(f32x4 (&acc)[4], const float *W, int bo0, const f32x4 (&v)[4], int lane) {
  const float *base = W + bo0 * KB * 256 + lane * 4;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    f32x4 a[4];
#pragma unroll
    for (int bo = 0; bo < 4; ++bo) a[bo] = ld4(base + (bo * KB + b) * 256);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int bo = 0; bo < 4; ++bo) acc[bo] = mfma16(a[bo][i], v[b][i], acc[bo]);
  }
}

// actor objective of one sample, NBH head blocks: lane (n, q) holds z[bo][i] = logit of action 16 bo + 4 q + i (A <= 16 NBH).
// On return z = d(actor objective)/d logits.  Expressions of actor_loss_quad (mlp_upd16.h) / actor_loss_regs (mlp_core.h).
template <int NBH>
__device__ __forceinline__ void actor_loss_q16(f32x4 (&z)[NBH], int A, int q, uint32_t dead, int act, float old_lp, float adv, float active,
                                               bool count, const mappo_ppo_cfg &cfg, float scale_pi, float (&lacc)[3]) {
  const float clip = cfg.clip_param;
  float zm = -FLT_MAX;
  bool valid[NBH][4];
#pragma unroll
  for (int bo = 0; bo < NBH; ++bo)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      valid[bo][i] = 16 * bo + 4 * q + i < A;
      if ((dead >> (4 * bo + i)) & 1u) z[bo][i] = -1e10f;
      if (valid[bo][i]) zm = fmaxf(zm, z[bo][i]);
    }
  const float zmax = quad_max16(zm);
  float e[NBH][4], se = 0.f;
#pragma unroll
  for (int bo = 0; bo < NBH; ++bo)
#pragma unroll
    for (int i = 0; i < 4; ++i) { e[bo][i] = valid[bo][i] ? expf(z[bo][i] - zmax) : 0.f; se += e[bo][i]; }
  se = quad_sum16(se);
  const float log_se = logf(se), inv_se = 1.0f / se;
  float hp = 0.f, za = 0.f;
#pragma unroll
  for (int bo = 0; bo < NBH; ++bo)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float l_ = (z[bo][i] - zmax) - log_se;
      if (valid[bo][i]) hp += (e[bo][i] * inv_se) * fmaxf(l_, -FLT_MAX);
      if (valid[bo][i] && 16 * bo + 4 * q + i == act) za = z[bo][i];
    }
  const float H = -quad_sum16(hp);
  const float z_act = quad_sum16(za);                     // one lane / register of the sample is non-zero: exact
  const float logp = (z_act - zmax) - log_se;
  const float ratio = expf(logp - old_lp);
  const float s1 = ratio * adv, s2 = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip) * adv;
  const float w = cfg.use_policy_active_masks ? active : 1.f;
  const float dlogp = (s1 <= s2) ? -(w * scale_pi) * adv * ratio : 0.f;
  const float ce = cfg.entropy_coef * w * scale_pi;
#pragma unroll
  for (int bo = 0; bo < NBH; ++bo)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float l_ = (z[bo][i] - zmax) - log_se;
      const float pa = e[bo][i] * inv_se;
      float g = dlogp * ((16 * bo + 4 * q + i == act ? 1.f : 0.f) - pa) + ce * pa * (l_ + H);
      if (!valid[bo][i] || ((dead >> (4 * bo + i)) & 1u) || !count) g = 0.f;
      z[bo][i] = g;
    }
  if (count && q == 0) {
    lacc[0] += w * fminf(s1, s2);
    lacc[1] += w * H;
    lacc[2] += ratio;
  }
}

// ---- LDS maps ------------------------------------------------------------------------------------------------------------------
template <int HEAD, int NBH>
struct F16Lds {
  static constexpr int DLS = NBH == 1 ? 20 : 36;                 // row stride of the d(logits) tile
  static constexpr int WIH = 0, WHH = WIH + G16_NG * HID, BIAS = WHH + G16_NG * HID;     // b_r + b_r', b_z + b_z', b_in, b_hn
  static constexpr int RN_G = BIAS + 4 * HID, RN_B = RN_G + HID;
  static constexpr int WH = RN_B + HID;                          // actor: head weights, fragment order [NBH][4][256] | critic: [64]
  static constexpr int BH = WH + (HEAD == 1 ? NBH * 4 * 256 : HID);
  static constexpr int TILES = BH + 32;
  static constexpr int UY = 0, UDL = UY + 16 * RS16, WAVE_STRIDE = UDL + (HEAD == 1 ? 16 * DLS : 0);
  static constexpr int TOTAL = TILES + G16_WAVES * WAVE_STRIDE;   // at the full 8 waves; a launch with nw waves asks for total(nw)
  __host__ __device__ static constexpr int total(int nw) { return TILES + nw * WAVE_STRIDE; }
  static_assert(TOTAL * 4 <= 159 * 1024, "gru16 forward: LDS");
  static_assert(G16_WAVES * NBH * 4 * 256 <= 2 * G16_NG * HID, "epilogue overlays the GRU weights");
};


// ---- the per-row tail: rnn.norm, head, loss and back to d h_t --------------------------------------------------------------------
// Shared by gru16_fwd_kernel (inside its time loop) and gru16_head_kernel (the small-batch pass over all (t, tile) row tiles).
template <int HEAD, int NBH>
struct Tail16Acc {                                                // one wave's running sums
  f32x4 gWh[HEAD == 1 ? NBH : 1][HEAD == 1 ? 4 : 1];
  float gBh[HEAD == 1 ? NBH : 1];
  float gWc, gNw, gNb;                                            // critic head product / rnn.norm gradients: lane = feature
  float lacc[3];
  __device__ __forceinline__ void clear() {
    gWc = gNw = gNb = 0.f;
    lacc[0] = lacc[1] = lacc[2] = 0.f;
#pragma unroll
    for (int bo = 0; bo < (HEAD == 1 ? NBH : 1); ++bo) {
      gBh[bo] = 0.f;
#pragma unroll
      for (int bk = 0; bk < (HEAD == 1 ? 4 : 1); ++bk) gWh[bo][bk] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
};
// A row's loss inputs: requested early, used late.  NOTHING here may be computed from a loaded value: an instruction on a value still
// in flight puts the wait for it — and, the counter being in order, for every load issued before it — right where the loads are
// issued (the availability mask was built here at first: s_waitcnt vmcnt(0) at the top of every step).  Row indices stay the raw
// 32-bit values for the same reason (the caller widens them a step later).
template <int NBH>
struct Tail16In { float f0, f1, f2, f3; float av[NBH][4]; };
struct Tail16Lds { const float *rn_g, *rn_b, *wh, *bh; float *Uy, *Udl; };

template <int HEAD, int NBH>
__device__ __forceinline__ Tail16In<NBH> tail16_load(const Gru16Args &p, int64_t brow, int q, int A) {
  Tail16In<NBH> r;
  r.f3 = 0.f;
  if constexpr (HEAD == 1) {
    r.f0 = p.actions[brow]; r.f1 = p.old_logp[brow]; r.f2 = p.adv[brow]; r.f3 = p.active[brow];
    // unconditional (a branch around these loads ends in copies of the loaded registers at its join — a wait again): without
    // available_actions the four dwords are re-reads of active[brow], never looked at
    const float *av = p.avail ? p.avail + brow * A : p.active + brow;
    const int amax = p.avail ? A - 1 : 0;
#pragma unroll
    for (int bo = 0; bo < NBH; ++bo)
#pragma unroll
      for (int i = 0; i < 4; ++i) r.av[bo][i] = av[min(16 * bo + 4 * q + i, amax)];
  } else {
    r.f0 = p.v_old[brow]; r.f1 = p.returns[brow]; r.f2 = p.active[brow];
  }
  return r;
}
template <int NBH>
__device__ __forceinline__ uint32_t tail16_dead(const Gru16Args &p, const Tail16In<NBH> &in, int q, int A) {
  uint32_t dead = 0u;
  if (p.avail) {
#pragma unroll
    for (int bo = 0; bo < NBH; ++bo)
#pragma unroll
      for (int i = 0; i < 4; ++i) dead |= ((16 * bo + 4 * q + i < A && in.av[bo][i] == 0.f) ? 1u : 0u) << (4 * bo + i);
  }
  return dead;
}

// h: the wave's tile of h_t (accumulator layout); d <- d h_t without the recurrent term
template <int HEAD, int NBH>
__device__ __forceinline__ void tail16_step(const Gru16Args &p, const Tail16Lds &L, const LossScales &ls, const Tail16In<NBH> &in, const f32x4 (&h)[4],
                                            bool ok, int A, int lane, Tail16Acc<HEAD, NBH> &acc, f32x4 (&d)[4]) {
  constexpr int DLS = NBH == 1 ? 20 : 36;
  const int n = lane & 15, q = lane >> 4;
  float *Uy = L.Uy, *Udl = L.Udl;
  // ---- y = rnn.norm(h_t) (rnn.py:79) ----
  f32x4 xh[4], y[4];
  float mean, rstd;
  {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 4; ++b) s += h[b];
    mean = quad_sum16((s[0] + s[1]) + (s[2] + s[3])) * (1.f / HID);
    const f32x4 mean4 = {mean, mean, mean, mean};
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 4; ++b) { xh[b] = h[b] - mean4; v += xh[b] * xh[b]; }
    rstd = 1.0f / sqrtf(quad_sum16((v[0] + v[1]) + (v[2] + v[3])) * (1.f / HID) + LN_EPS);
    const f32x4 rstd4 = {rstd, rstd, rstd, rstd};
#pragma unroll
    for (int b = 0; b < 4; ++b) { xh[b] *= rstd4; y[b] = xh[b] * ld4(L.rn_g + 16 * b + 4 * q) + ld4(L.rn_b + 16 * b + 4 * q); }
  }
  f32x4 dy[4];
  if constexpr (HEAD == 1) {
#pragma unroll
    for (int b = 0; b < 4; ++b) st4(Uy + n * RS16 + 16 * b + 4 * q, y[b]);
    f32x4 zl[NBH];
#pragma unroll
    for (int bo = 0; bo < NBH; ++bo) zl[bo] = ld4(L.bh + 16 * bo + 4 * q);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      f32x4 a[NBH];
#pragma unroll
      for (int bo = 0; bo < NBH; ++bo) a[bo] = ld4(L.wh + ((bo * 4 + b) * 64 + lane) * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int bo = 0; bo < NBH; ++bo) zl[bo] = mfma16(a[bo][i], y[b][i], zl[bo]);
    }
    actor_loss_q16<NBH>(zl, A, q, tail16_dead<NBH>(p, in, q, A), (int)in.f0, in.f1, in.f2, in.f3, ok, p.cfg, ls.scale_pi, acc.lacc);
#pragma unroll
    for (int bo = 0; bo < NBH; ++bo) st4(Udl + n * DLS + 16 * bo + 4 * q, zl[bo]);
    wave_lds_sync();
    dw_accum16<NBH, 4>(acc.gWh, acc.gBh, Udl, DLS, Uy, RS16, n, q);
    // d y = Wh^T dl: out block bk, k-step (bo, i) takes action 16 bo + 4 q + i; A operand Wh[16 bo + 4 q + i][16 bk + n] sits in
    // fragment (bo, bk) at lane' = (4 q + i, n >> 2), element n & 3
#pragma unroll
    for (int bk = 0; bk < 4; ++bk) dy[bk] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *wt = L.wh + ((n >> 2) * 16 + 4 * q) * 4 + (n & 3);
#pragma unroll
    for (int bo = 0; bo < NBH; ++bo)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float a[4];
#pragma unroll
        for (int bk = 0; bk < 4; ++bk) a[bk] = wt[(bo * 4 + bk) * 256 + 4 * i];
#pragma unroll
        for (int bk = 0; bk < 4; ++bk) dy[bk] = mfma16(a[bk], zl[bo][i], dy[bk]);
      }
    wave_lds_sync();
  } else {
    f32x4 wv[4];
    float sacc = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      wv[b] = ld4(L.wh + 16 * b + 4 * q);
#pragma unroll
      for (int i = 0; i < 4; ++i) sacc += wv[b][i] * y[b][i];
    }
    const float v = quad_sum16(sacc) + L.bh[0];
    float dv = critic_loss16(v, in.f0, in.f1, in.f2, p.cfg, ls, ok && q == 0, acc.lacc);
    dv = ok ? dv : 0.f;
    if (q == 0) acc.gBh[0] += dv;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      st4(Uy + n * RS16 + 16 * b + 4 * q, y[b] * f32x4{dv, dv, dv, dv});
      dy[b] = wv[b] * f32x4{dv, dv, dv, dv};
    }
    wave_lds_sync();
    acc.gWc += col_sum16(Uy, RS16, lane);
    wave_lds_sync();
  }
  // rnn.norm backward: d gamma = sum dy o xhat, d beta = sum dy (column sums through the wave's tile), then d h
#pragma unroll
  for (int b = 0; b < 4; ++b) st4(Uy + n * RS16 + 16 * b + 4 * q, dy[b] * xh[b]);
  wave_lds_sync();
  acc.gNw += col_sum16(Uy, RS16, lane);
  wave_lds_sync();
#pragma unroll
  for (int b = 0; b < 4; ++b) st4(Uy + n * RS16 + 16 * b + 4 * q, dy[b]);
  wave_lds_sync();
  acc.gNb += col_sum16(Uy, RS16, lane);
  wave_lds_sync();
  {
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 4; ++b) { dy[b] *= ld4(L.rn_g + 16 * b + 4 * q); s1 += dy[b]; s2 += dy[b] * xh[b]; }
    const float m1 = quad_sum16((s1[0] + s1[1]) + (s1[2] + s1[3])) * (1.f / HID);
    const float m2 = quad_sum16((s2[0] + s2[1]) + (s2[2] + s2[3])) * (1.f / HID);
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) d[b][i] = rstd * (dy[b][i] - m1 - xh[b][i] * m2);
  }
}

// workgroup reduction of the waves' sums -> slab row `bid` and loss partials row `bid`.  red: [nw][NBH * 4][256] floats (actor),
// vec: [nw][8][64]; the caller has made both areas free (a __syncthreads() after the last use of what they overlay).
template <int HEAD, int NBH>
__device__ __forceinline__ void tail16_epilogue(const Gru16Args &p, const Tail16Acc<HEAD, NBH> &acc, float *red, float *vec, int wave, int nw,
                                                int lane, int bid) {
  const NetOff &o = p.off;
  const int A = p.A, nthr = blockDim.x;
  if constexpr (HEAD == 1) {
#pragma unroll
    for (int bo = 0; bo < NBH; ++bo)
#pragma unroll
      for (int bk = 0; bk < 4; ++bk) st4(red + ((wave * NBH * 4 + bo * 4 + bk) * 64 + lane) * 4, acc.gWh[bo][bk]);
  }
  vec[(wave * 8 + 0) * 64 + lane] = acc.gNw;
  vec[(wave * 8 + 1) * 64 + lane] = acc.gNb;
  if constexpr (HEAD == 1) {
#pragma unroll
    for (int bo = 0; bo < NBH; ++bo) vec[(wave * 8 + 2 + bo) * 64 + lane] = quad_sum16(acc.gBh[bo]);
  } else {
    vec[(wave * 8 + 2) * 64 + lane] = acc.gWc;
    const float sdv = wave_sum_f(acc.gBh[0]);
    if (lane == 0) vec[(wave * 8 + 3) * 64] = sdv;
  }
  {
    const float l0 = wave_sum_f(acc.lacc[0]), l1 = wave_sum_f(acc.lacc[1]), l2 = wave_sum_f(acc.lacc[2]);
    if (lane == 0) { vec[(wave * 8 + 4) * 64 + 0] = l0; vec[(wave * 8 + 4) * 64 + 1] = l1; vec[(wave * 8 + 4) * 64 + 2] = l2; }
  }
  __syncthreads();
  float *slab = p.slabs + (size_t)bid * p.slab_stride + p.slab_col0;
  if constexpr (HEAD == 1) {
    for (int e = threadIdx.x; e < NBH * 4 * 256; e += nthr) {
      float s = 0.f;
      for (int w = 0; w < nw; ++w) s += red[w * NBH * 4 * 256 + e];
      const int blk = e >> 8, ln = (e >> 2) & 63, i = e & 3;
      const int a = 16 * (blk >> 2) + 4 * (ln >> 4) + i, k = 16 * (blk & 3) + (ln & 15);
      if (a < A) slab[o.wh + a * HID + k] = s;
    }
  }
  for (int e = threadIdx.x; e < 4 * 64; e += nthr) {
    const int which = e >> 6, k = e & 63;                        // 0: rn_w, 1: rn_b, 2: head vector 0, 3: head vector 1
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += vec[(w * 8 + which) * 64 + k];
    if (which == 0) slab[o.rn_w + k] = s;
    else if (which == 1) slab[o.rn_b + k] = s;
    else if (HEAD == 1) {
      const int a = 16 * (which - 2) + k;
      if (k < 16 && which - 2 < NBH && a < A) slab[o.bh + a] = s;
    } else if (which == 2) slab[o.wh + k] = s;
    else if (k == 0) slab[o.bh] = s;
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double v = 0.0;
      if (k < 3) for (int w = 0; w < nw; ++w) v += (double)vec[(w * 8 + 4) * 64 + k];
      p.partials[(size_t)bid * 4 + k] = v;
    }
  }
}

// ================================================================================================================================
// forward + head + loss + head backward
// ================================================================================================================================
template <int HEAD, int NBH, bool XBLK>
__device__ __forceinline__ void gru16_fwd_body(const Gru16Args &p, float *lds, const int bid, const int nb) {
  typedef F16Lds<HEAD, NBH> M;
  const NetOff &o = p.off;
  const int lane = threadIdx.x & (WAVE - 1), n = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
  const int nw = blockDim.x / WAVE;                               // 1..8 waves: launches with few tiles spread them over more CUs
  const int A = p.A;
  // ---- staging ----
  stage_frag<6>(lds + M::WIH, p.params + o.gru_wih, G16_NG, HID, G16_NG);
  stage_frag<6>(lds + M::WHH, p.params + o.gru_whh, G16_NG, HID, G16_NG);
  for (int e = threadIdx.x; e < 6 * HID + 32; e += blockDim.x) {
    if (e < 4 * HID) {
      const int gate = e >> 6, f = e & 63;
      float v;
      if (gate < 2) v = p.params[o.gru_bih + e] + p.params[o.gru_bhh + e];
      else if (gate == 2) v = p.params[o.gru_bih + 2 * HID + f];
      else v = p.params[o.gru_bhh + 2 * HID + f];
      lds[M::BIAS + e] = v;
    } else if (e < 5 * HID) lds[M::RN_G + (e - 4 * HID)] = p.params[o.rn_w + (e - 4 * HID)];
    else if (e < 6 * HID) lds[M::RN_B + (e - 5 * HID)] = p.params[o.rn_b + (e - 5 * HID)];
    else {
      const int a = e - 6 * HID;
      lds[M::BH + a] = a < A ? p.params[o.bh + a] : 0.f;
    }
  }
  if constexpr (HEAD == 1) stage_frag<2>(lds + M::WH, p.params + o.wh, 16 * NBH, HID, A);
  else { if (threadIdx.x < HID) lds[M::WH + threadIdx.x] = p.params[o.wh + threadIdx.x]; }
  __syncthreads();
  float *Uy = lds + M::TILES + wave * M::WAVE_STRIDE + M::UY;
  float *Udl = lds + M::TILES + wave * M::WAVE_STRIDE + M::UDL;
  LossScales ls = loss_scales(p.cfg, p.mb_moments, p.vn_state);
  ls.scale_pi = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ls.scale_pi)));
  ls.scale_v = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ls.scale_v)));
  ls.vn_mean = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ls.vn_mean)));
  ls.vn_sd = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ls.vn_sd)));
  Tail16Acc<HEAD, NBH> acc;
  acc.clear();
  const Tail16Lds TL = {lds + M::RN_G, lds + M::RN_B, lds + M::WH, lds + M::BH, Uy, Udl};
  const int n_ct = (p.Nc + 15) >> 4;
  const int64_t B = (int64_t)p.L * p.Nc;
  const int64_t CS = (int64_t)p.L * n_ct * 1024;                  // floats per scratch component
  for (int tile = bid * nw + wave; tile < n_ct; tile += nb * nw) {
    const int c = tile * 16 + n;
    const bool ok = c < p.Nc;
    const int cc = ok ? c : 0;
    const int64_t hrow = p.h0_rows ? (int64_t)p.h0_rows[cc] : (int64_t)cc;
    f32x4 h[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) h[b] = ld4(p.h0 + hrow * HID + 16 * b + 4 * q);
    // the reset mask gates the first MFMA of a step, behind a two-deep dependent load (rows -> masks): the row index runs two steps
    // ahead and the mask one step ahead, so a lone wave of a SIMD (small batches) does not wait for either
    auto row_of = [&](int t) -> int {                           // raw: see Tail16In
      const int64_t col = (int64_t)min(t, p.L - 1) * p.Nc + cc;
      int r = (int)col;
      if (p.rows) r = p.rows[col];
      return r;
    };
    int r0 = row_of(0), r1 = row_of(1);
    float mk = p.masks[r0];
    for (int t = 0; t < p.L; ++t) {
      const int64_t col = (int64_t)t * p.Nc + cc;
      const int r2 = row_of(t + 2);
      const float mk1 = p.masks[r1];
      const int64_t brow = r0;
      // ---- this step's inputs: everything is requested before the first use ----
      f32x4 x[4];
      if constexpr (XBLK) {
        const float *xb = p.x + ((int64_t)(t * n_ct + tile) * 4) * 256 + lane * 4;
#pragma unroll
        for (int b = 0; b < 4; ++b) x[b] = ld4(xb + b * 256);
      } else {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) x[b][i] = p.x[(int64_t)(16 * b + 4 * q + i) * B + col];
      }
      const Tail16In<NBH> in = tail16_load<HEAD, NBH>(p, brow, q, A);
      const float mk0 = ok ? mk : 0.f;
      f32x4 hm[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) hm[b] = h[b] * f32x4{mk0, mk0, mk0, mk0};
      // ---- gates: gi + gh, 384 MFMAs in four-accumulator groups ----
      f32x4 ar[4], az[4], ain[4], ahn[4];
#pragma unroll
      for (int bo = 0; bo < 4; ++bo) {
        ar[bo] = ld4(lds + M::BIAS + 16 * bo + 4 * q);
        az[bo] = ld4(lds + M::BIAS + HID + 16 * bo + 4 * q);
        ain[bo] = ld4(lds + M::BIAS + 2 * HID + 16 * bo + 4 * q);
        ahn[bo] = ld4(lds + M::BIAS + 3 * HID + 16 * bo + 4 * q);
      }
      frag_mma4<4>(ar, lds + M::WHH, 0, hm, lane);
      frag_mma4<4>(az, lds + M::WHH, 4, hm, lane);
      frag_mma4<4>(ahn, lds + M::WHH, 8, hm, lane);
      frag_mma4<4>(ar, lds + M::WIH, 0, x, lane);
      frag_mma4<4>(az, lds + M::WIH, 4, x, lane);
      frag_mma4<4>(ain, lds + M::WIH, 8, x, lane);
      float *sb = p.scratch + ((int64_t)(t * n_ct + tile) * 4) * 256 + lane * 4;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        f32x4 r, z, nn;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          r[i] = sigmoid16(ar[b][i]);
          z[i] = sigmoid16(az[b][i]);
          nn[i] = tanh16(ain[b][i] + r[i] * ahn[b][i]);
          h[b][i] = (1.f - z[i]) * nn[i] + z[i] * hm[b][i];
        }
        st4(sb + C_HM * CS + b * 256, hm[b]);
        st4(sb + C_R * CS + b * 256, r);
        st4(sb + C_Z * CS + b * 256, z);
        st4(sb + C_N * CS + b * 256, nn);
        st4(sb + C_GHN * CS + b * 256, ahn[b]);
      }
      // ---- y = rnn.norm(h_t) (rnn.py:79), head, loss, and back to d h_t ----
      f32x4 d[4];
      tail16_step<HEAD, NBH>(p, TL, ls, in, h, ok, A, lane, acc, d);
#pragma unroll
      for (int b = 0; b < 4; ++b) st4(sb + C_DH * CS + b * 256, d[b]);
      r0 = r1; r1 = r2; mk = mk1;
    }
  }
  // ---- workgroup reduction -> slab row `bid`, loss partials ----
  __syncthreads();                                               // every wave is done with the weights: their area is free
  tail16_epilogue<HEAD, NBH>(p, acc, lds, lds + G16_WAVES * NBH * 4 * 256, wave, nw, lane, bid);
}

template <int HEAD, int NBH, bool XBLK>
__global__ __launch_bounds__(G16_THREADS, 2) void gru16_fwd_kernel(Gru16Args a) {
  extern __shared__ __align__(16) float lds[];
  gru16_fwd_body<HEAD, NBH, XBLK>(a, lds, blockIdx.x, gridDim.x);
}

// ================================================================================================================================
// backward recurrence + d x
// ================================================================================================================================
template <bool DXBLK, bool PRE>
__device__ __forceinline__ void gru16_bwd_body(const Gru16Args &p, float *lds, const int bid, const int nb) {
  const NetOff &o = p.off;
  const int lane = threadIdx.x & (WAVE - 1), n = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
  float *WHT = lds, *WIT = lds + G16_NG * HID;                    // W_hh^T, W_ih^T: out blocks over k (4), k-steps over the 192 gate rows (12)
  stage_frag_T<6>(WHT, p.params + o.gru_whh, G16_NG, HID);
  stage_frag_T<6>(WIT, p.params + o.gru_wih, G16_NG, HID);
  __syncthreads();
  const int nw = blockDim.x / WAVE;
  const int n_ct = (p.Nc + 15) >> 4;
  const int64_t B = (int64_t)p.L * p.Nc;
  const int64_t CS = (int64_t)p.L * n_ct * 1024;
  for (int tile = bid * nw + wave; tile < n_ct; tile += nb * nw) {
    const int c = tile * 16 + n;
    const bool ok = c < p.Nc;
    const int cc = ok ? c : 0;
    f32x4 carry[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) carry[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // PRE (one wave per SIMD: no partner to cover the latency, and 512 registers to spend): step t - 1's six vectors are requested
    // at the top of step t and land under its 384 MFMAs
    f32x4 pf[PRE ? 6 : 1][4];
    auto load6 = [&](f32x4 (&d)[PRE ? 6 : 1][4], const float *s0) {
      if constexpr (PRE) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          d[0][b] = ld4(s0 + C_DH * CS + b * 256); d[1][b] = ld4(s0 + C_Z * CS + b * 256); d[2][b] = ld4(s0 + C_N * CS + b * 256);
          d[3][b] = ld4(s0 + C_R * CS + b * 256); d[4][b] = ld4(s0 + C_GHN * CS + b * 256); d[5][b] = ld4(s0 + C_HM * CS + b * 256);
        }
      }
    };
    load6(pf, p.scratch + ((int64_t)((p.L - 1) * n_ct + tile) * 4) * 256 + lane * 4);
    // reset mask of a step: behind a two-deep gather (rows -> masks).  The row index runs two steps ahead and the mask one, both
    // untouched until the step that uses them (raw 32-bit index: see Tail16In) — read in place, the chain was a memory round trip
    // in front of every step's 24 loads.
    auto row_of = [&](int t) -> int {
      const int64_t col = (int64_t)max(t, 0) * p.Nc + cc;
      int r = (int)col;
      if (p.rows) r = p.rows[col];
      return r;
    };
    int r1 = row_of(p.L - 2);
    float mkc = p.masks[row_of(p.L - 1)];
    for (int t = p.L - 1; t >= 0; --t) {
      // (compiler fence: without it hipcc hoists the NEXT step's 24 loads above this step's MFMA phase — 96 more live registers,
      // 124 of them spilled; the partner wave of the SIMD covers the load latency instead)
      asm volatile("" ::: "memory");
      const int64_t col = (int64_t)t * p.Nc + cc;
      float *sb = p.scratch + ((int64_t)(t * n_ct + tile) * 4) * 256 + lane * 4;
      f32x4 dh[4], hm[4], gr[4], gz[4], gn[4], ghn[4];
      if constexpr (PRE) {
#pragma unroll
        for (int b = 0; b < 4; ++b) { dh[b] = pf[0][b]; gz[b] = pf[1][b]; gn[b] = pf[2][b]; gr[b] = pf[3][b]; ghn[b] = pf[4][b]; hm[b] = pf[5][b]; }
        if (t > 0) load6(pf, sb - (int64_t)n_ct * 1024);
      } else {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          dh[b] = ld4(sb + C_DH * CS + b * 256);
          gz[b] = ld4(sb + C_Z * CS + b * 256);
          gn[b] = ld4(sb + C_N * CS + b * 256);
          gr[b] = ld4(sb + C_R * CS + b * 256);
          ghn[b] = ld4(sb + C_GHN * CS + b * 256);
          hm[b] = ld4(sb + C_HM * CS + b * 256);
        }
      }
      const int r2 = row_of(t - 2);
      const float mk1 = p.masks[r1];
      const float mk = ok ? mkc : 0.f;
      // d gates, IN PLACE (gr <- d pre_r, hm <- d pre_z, gn <- d pre_n = d gi_n, ghn <- d gh_n; dh <- d h_t + carry): the register
      // file holds exactly the six loaded vectors.  Dead sequences: d h = 0 and carry = 0, so every product below is 0.
#pragma unroll
      for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float dhh = (ok ? dh[b][i] : 0.f) + carry[b][i], zz = gz[b][i], nn = gn[b][i], rr = gr[b][i];
          dh[b][i] = dhh;
          const float dn_pre = dhh * (1.f - zz) * (1.f - nn * nn);
          gn[b][i] = dn_pre;
          gr[b][i] = dn_pre * ghn[b][i] * rr * (1.f - rr);
          ghn[b][i] = dn_pre * rr;
          hm[b][i] = dhh * (hm[b][i] - nn) * zz * (1.f - zz);
        }
        st4(sb + C_R * CS + b * 256, gr[b]);
        st4(sb + C_Z * CS + b * 256, hm[b]);
        st4(sb + C_N * CS + b * 256, gn[b]);
        st4(sb + C_GHN * CS + b * 256, ghn[b]);
      }
      // carry: W_hh^T [d_r, d_z, d_hn]; d x: W_ih^T [d_r, d_z, d_n] — two independent four-accumulator sets
      f32x4 dhm[4], dx[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) { dhm[b] = f32x4{0.f, 0.f, 0.f, 0.f}; dx[b] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      const float *bh = WHT + lane * 4, *bi = WIT + lane * 4;
#pragma unroll
      for (int gate = 0; gate < 3; ++gate) {
        const f32x4 (&vh)[4] = gate == 0 ? gr : (gate == 1 ? hm : ghn);
        const f32x4 (&vi)[4] = gate == 0 ? gr : (gate == 1 ? hm : gn);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          f32x4 ah[4], ai[4];
#pragma unroll
          for (int bo = 0; bo < 4; ++bo) {
            ah[bo] = ld4(bh + (bo * 12 + gate * 4 + b) * 256);
            ai[bo] = ld4(bi + (bo * 12 + gate * 4 + b) * 256);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int bo = 0; bo < 4; ++bo) dhm[bo] = mfma16(ah[bo][i], vh[b][i], dhm[bo]);
#pragma unroll
            for (int bo = 0; bo < 4; ++bo) dx[bo] = mfma16(ai[bo][i], vi[b][i], dx[bo]);
          }
        }
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if constexpr (DXBLK) st4(sb + C_DH * CS + b * 256, dx[b]);
        else if (ok) {
#pragma unroll
          for (int i = 0; i < 4; ++i) p.dxT[(int64_t)(16 * b + 4 * q + i) * B + col] = dx[b][i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) carry[b][i] = (dhm[b][i] + dh[b][i] * gz[b][i]) * mk;
      }
      r1 = r2; mkc = mk1;
    }
  }
}

template <bool DXBLK>
__global__ __launch_bounds__(G16_THREADS, 2) void gru16_bwd_kernel(Gru16Args a) {
  extern __shared__ __align__(16) float lds[];
  gru16_bwd_body<DXBLK, false>(a, lds, blockIdx.x, gridDim.x);
}
// at most four waves per workgroup (small batches, seq_waves()): one wave per SIMD, so the whole register file is the wave's
template <bool DXBLK>
__global__ __launch_bounds__(4 * WAVE, 1) void gru16_bwd4_kernel(Gru16Args a) {
  extern __shared__ __align__(16) float lds[];
  gru16_bwd_body<DXBLK, true>(a, lds, blockIdx.x, gridDim.x);
}

// ================================================================================================================================
// small batches: the same pass when the tile count cannot fill the chip with one wave per tile
// ================================================================================================================================
// With n_ct <= ~4 tiles per CU the kernels above run one wave per SIMD, each alone with the whole 444-MFMA + gate + head chain of
// its tile's step (config-2 rmappo: 29k cycles per step, MFMA pipe 50 % idle, half of the SIMDs empty).  Here a tile's step is
// split over the FOUR waves of a workgroup by output-feature block (wave j: features 16 j .. 16 j + 15 of every gate — 96 MFMAs,
// a quarter of the gate arithmetic), the weights of a wave (2 x 48 rows x 64) stay in REGISTERS as A operands (no LDS staging,
// no LDS weight traffic), h_t / the d-gates are exchanged through a double-buffered LDS tile with ONE barrier per step, and
// the per-row tail (rnn.norm, head, loss, their backward) — which has no time dependence — leaves the recurrence and runs as a
// parallel pass over all (t, tile) row tiles (gru16_head_kernel).  Scratch layout and contents are those of the kernels above
// (during the forward, component C_DH carries h_t from gru16s_fwd_kernel to gru16_head_kernel, which overwrites it with d h_t).
template <bool XBLK>
__global__ __launch_bounds__(4 * WAVE) void gru16s_fwd_kernel(Gru16Args p) {
  __shared__ __align__(16) float ex[2][4 * 256];                  // h_t, blocked [b][lane][4]
  const NetOff &o = p.off;
  const int lane = threadIdx.x & (WAVE - 1), n = lane & 15, q = lane >> 4;
  const int j = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
  const int n_ct = (p.Nc + 15) >> 4;
  const int64_t B = (int64_t)p.L * p.Nc;
  const int64_t CS = (int64_t)p.L * n_ct * 1024;
  // A operands: lane (m = n, k quarter q) of out block j, k-step (b, i): W[gate * 64 + 16 j + n][16 b + 4 q + i]
  f32x4 wh[3][4], wi[3][4];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      wh[g][b] = ld4(p.params + o.gru_whh + (g * HID + 16 * j + n) * HID + 16 * b + 4 * q);
      wi[g][b] = ld4(p.params + o.gru_wih + (g * HID + 16 * j + n) * HID + 16 * b + 4 * q);
    }
  const int fo = 16 * j + 4 * q;                                  // the lane's four features
  const f32x4 b_r = ld4(p.params + o.gru_bih + fo) + ld4(p.params + o.gru_bhh + fo);
  const f32x4 b_z = ld4(p.params + o.gru_bih + HID + fo) + ld4(p.params + o.gru_bhh + HID + fo);
  const f32x4 b_in = ld4(p.params + o.gru_bih + 2 * HID + fo), b_hn = ld4(p.params + o.gru_bhh + 2 * HID + fo);
  for (int tile = blockIdx.x; tile < n_ct; tile += gridDim.x) {   // (the weights stay in registers across a workgroup's tiles)
  const int c = tile * 16 + n;
  const bool ok = c < p.Nc;
  const int cc = ok ? c : 0;
  const int64_t hrow = p.h0_rows ? (int64_t)p.h0_rows[cc] : (int64_t)cc;
  __syncthreads();                                                // the previous tile's last reads of ex[] are done
  st4(&ex[0][j * 256 + lane * 4], ld4(p.h0 + hrow * HID + fo));
  auto row_of = [&](int t) -> int {                             // raw: see Tail16In
    const int64_t col = (int64_t)min(t, p.L - 1) * p.Nc + cc;
    int r = (int)col;
    if (p.rows) r = p.rows[col];
    return r;
  };
  auto load_x = [&](f32x4 (&x)[4], int t) {
    const int tt = min(t, p.L - 1);
    if constexpr (XBLK) {
      const float *xb = p.x + ((int64_t)(tt * n_ct + tile) * 4) * 256 + lane * 4;
#pragma unroll
      for (int b = 0; b < 4; ++b) x[b] = ld4(xb + b * 256);
    } else {
      const int64_t col = (int64_t)tt * p.Nc + cc;
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) x[b][i] = p.x[(int64_t)(16 * b + 4 * q + i) * B + col];
    }
  };
  int r1 = row_of(1);
  float mk = p.masks[row_of(0)];
  f32x4 x[4];
  load_x(x, 0);
  for (int t = 0; t < p.L; ++t) {
    const int r2 = row_of(t + 2);
    const float mk1 = p.masks[r1];
    // input-side products first: they do not wait for the other waves' h_{t-1}
    f32x4 ar = b_r, az = b_z, ain = b_in, ahn = b_hn;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ar = mfma16(wi[0][b][i], x[b][i], ar);
        az = mfma16(wi[1][b][i], x[b][i], az);
        ain = mfma16(wi[2][b][i], x[b][i], ain);
      }
    load_x(x, t + 1);                                             // lands under the recurrent half
    __syncthreads();                                              // ex[t & 1] = h_{t-1} is complete
    const float *e = ex[t & 1];
    const float mk0 = ok ? mk : 0.f;
    const f32x4 mk4 = {mk0, mk0, mk0, mk0};
    f32x4 hm[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) hm[b] = ld4(e + b * 256 + lane * 4) * mk4;
    const f32x4 hmj = ld4(e + j * 256 + lane * 4) * mk4;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ar = mfma16(wh[0][b][i], hm[b][i], ar);
        az = mfma16(wh[1][b][i], hm[b][i], az);
        ahn = mfma16(wh[2][b][i], hm[b][i], ahn);
      }
    f32x4 r, z, nn, hn;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      r[i] = sigmoid16(ar[i]);
      z[i] = sigmoid16(az[i]);
      nn[i] = tanh16(ain[i] + r[i] * ahn[i]);
      hn[i] = (1.f - z[i]) * nn[i] + z[i] * hmj[i];
    }
    st4(&ex[(t + 1) & 1][j * 256 + lane * 4], hn);                // read after the next step's barrier
    float *sb = p.scratch + ((int64_t)(t * n_ct + tile) * 4) * 256 + j * 256 + lane * 4;
    st4(sb + C_HM * CS, hmj);
    st4(sb + C_R * CS, r);
    st4(sb + C_Z * CS, z);
    st4(sb + C_N * CS, nn);
    st4(sb + C_GHN * CS, ahn);
    st4(sb + C_DH * CS, hn);                                      // h_t for gru16_head_kernel
    r1 = r2; mk = mk1;
  }
  }
}

template <int HEAD, int NBH>
struct H16Lds {
  static constexpr int DLS = NBH == 1 ? 20 : 36;
  static constexpr int NW = 4;
  static constexpr int RN_G = 0, RN_B = RN_G + HID, WH = RN_B + HID, BH = WH + (HEAD == 1 ? NBH * 4 * 256 : HID), TILES = BH + 32;
  static constexpr int WAVE_STRIDE = 16 * RS16 + (HEAD == 1 ? 16 * DLS : 0);
  static constexpr int RED = TILES;                              // the epilogue's sums overlay the waves' tiles (after a barrier)
  static constexpr int RED_N = HEAD == 1 ? NW * NBH * 4 * 256 : 0;
  static constexpr int VEC = TILES + (RED_N > NW * WAVE_STRIDE ? RED_N : NW * WAVE_STRIDE), TOTAL = VEC + NW * 8 * 64;
  static_assert(TOTAL * 4 <= 64 * 1024, "gru16 head: static LDS");
};

// rnn.norm + head + loss + their backward over ALL (t, tile) row tiles: h_t (scratch C_DH) -> d h_t in place
template <int HEAD, int NBH>
__global__ __launch_bounds__(4 * WAVE) void gru16_head_kernel(Gru16Args p) {
  typedef H16Lds<HEAD, NBH> M;
  __shared__ __align__(16) float lds[M::TOTAL];
  const NetOff &o = p.off;
  const int lane = threadIdx.x & (WAVE - 1), n = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
  const int A = p.A;
  for (int e = threadIdx.x; e < 2 * HID + 32; e += blockDim.x) {
    if (e < HID) lds[M::RN_G + e] = p.params[o.rn_w + e];
    else if (e < 2 * HID) lds[M::RN_B + (e - HID)] = p.params[o.rn_b + (e - HID)];
    else {
      const int a = e - 2 * HID;
      lds[M::BH + a] = a < A ? p.params[o.bh + a] : 0.f;
    }
  }
  if constexpr (HEAD == 1) stage_frag<2>(lds + M::WH, p.params + o.wh, 16 * NBH, HID, A);
  else { if (threadIdx.x < HID) lds[M::WH + threadIdx.x] = p.params[o.wh + threadIdx.x]; }
  __syncthreads();
  LossScales ls = loss_scales(p.cfg, p.mb_moments, p.vn_state);
  ls.scale_pi = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ls.scale_pi)));
  ls.scale_v = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ls.scale_v)));
  ls.vn_mean = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ls.vn_mean)));
  ls.vn_sd = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ls.vn_sd)));
  Tail16Acc<HEAD, NBH> acc;
  acc.clear();
  float *Uy = lds + M::TILES + wave * M::WAVE_STRIDE;
  const Tail16Lds TL = {lds + M::RN_G, lds + M::RN_B, lds + M::WH, lds + M::BH, Uy, Uy + 16 * RS16};
  const int n_ct = (p.Nc + 15) >> 4, n_rt = p.L * n_ct;
  const int64_t CS = (int64_t)p.L * n_ct * 1024;
  const int stride = gridDim.x * M::NW;
  auto brow_of = [&](int rt) -> int {                             // (clamped: the tile after a wave's last is a repeat, discarded; raw: see Tail16In)
    const int r = min(rt, n_rt - 1), t = r / n_ct, c = (r - t * n_ct) * 16 + n;
    const int64_t col = (int64_t)t * p.Nc + (c < p.Nc ? c : 0);
    int b = (int)col;
    if (p.rows) b = p.rows[col];
    return b;
  };
  auto load_tile = [&](f32x4 (&h)[4], Tail16In<NBH> &in, int rt, int64_t brow) {
    const float *sb = p.scratch + C_DH * CS + (int64_t)min(rt, n_rt - 1) * 1024 + lane * 4;
#pragma unroll
    for (int b = 0; b < 4; ++b) h[b] = ld4(sb + b * 256);
    in = tail16_load<HEAD, NBH>(p, brow, q, A);
  };
  // a row's inputs sit behind a two-deep gather (rows -> loss inputs): row indices run two tiles ahead, the inputs one
  int rt = blockIdx.x * M::NW + wave;
  f32x4 h[4], hn[4];
  Tail16In<NBH> in, inn;
  int brow1 = brow_of(rt + stride);
  load_tile(h, in, rt, brow_of(rt));
  for (; rt < n_rt; rt += stride) {
    const int brow2 = brow_of(rt + 2 * stride);
    load_tile(hn, inn, rt + stride, brow1);
    const int t = rt / n_ct, tile = rt - t * n_ct;
    const bool ok = tile * 16 + n < p.Nc;
    float *sb = p.scratch + C_DH * CS + (int64_t)rt * 1024 + lane * 4;
    f32x4 d[4];
    tail16_step<HEAD, NBH>(p, TL, ls, in, h, ok, A, lane, acc, d);
#pragma unroll
    for (int b = 0; b < 4; ++b) { st4(sb + b * 256, d[b]); h[b] = hn[b]; }
    in = inn; brow1 = brow2;
  }
  __syncthreads();                                               // every wave is done with its tiles: the sums overlay them
  tail16_epilogue<HEAD, NBH>(p, acc, lds + M::RED, lds + M::VEC, wave, M::NW, lane, blockIdx.x);
}

// reverse time, wave j = out block j of the carry AND of d x; the four d-gate tiles of a step go through LDS
template <bool DXBLK>
__global__ __launch_bounds__(4 * WAVE) void gru16s_bwd_kernel(Gru16Args p) {
  __shared__ __align__(16) float ex[2][4][4 * 256];               // [buffer][d_r, d_z, d_n, d_hn][b][lane][4]
  const NetOff &o = p.off;
  const int lane = threadIdx.x & (WAVE - 1), n = lane & 15, q = lane >> 4;
  const int j = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
  const int n_ct = (p.Nc + 15) >> 4;
  const int64_t B = (int64_t)p.L * p.Nc;
  const int64_t CS = (int64_t)p.L * n_ct * 1024;
  // A operands of W^T: lane (m = n, q) of out block j, k-step (gate, b, i): W[gate * 64 + 16 b + 4 q + i][16 j + n]
  f32x4 wh[3][4], wi[3][4];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wh[g][b][i] = p.params[o.gru_whh + (g * HID + 16 * b + 4 * q + i) * HID + 16 * j + n];
        wi[g][b][i] = p.params[o.gru_wih + (g * HID + 16 * b + 4 * q + i) * HID + 16 * j + n];
      }
  for (int tile = blockIdx.x; tile < n_ct; tile += gridDim.x) {   // (a tile's first two steps write both buffers only after
  const int c = tile * 16 + n;                                    // barriers every wave reaches once done with the previous tile)
  const bool ok = c < p.Nc;
  const int cc = ok ? c : 0;
  f32x4 carry = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  struct Step { f32x4 dh, z, nn, r, ghn, hm; float mk; };
  auto row_of = [&](int t) -> int {                               // raw 32-bit row index, used a step after it is loaded (see Tail16In)
    const int64_t col = (int64_t)max(t, 0) * p.Nc + cc;
    int r = (int)col;
    if (p.rows) r = p.rows[col];
    return r;
  };
  auto load_step = [&](Step &s, int t, int row) {
    const int tt = max(t, 0);
    const float *sb = p.scratch + ((int64_t)(tt * n_ct + tile) * 4) * 256 + j * 256 + lane * 4;
    s.dh = ld4(sb + C_DH * CS); s.z = ld4(sb + C_Z * CS); s.nn = ld4(sb + C_N * CS);
    s.r = ld4(sb + C_R * CS); s.ghn = ld4(sb + C_GHN * CS); s.hm = ld4(sb + C_HM * CS);
    s.mk = p.masks[row];
  };
  Step cur, nxt;
  load_step(cur, p.L - 1, row_of(p.L - 1));
  int r1 = row_of(p.L - 2);
  for (int t = p.L - 1; t >= 0; --t) {
    const int r2 = row_of(t - 2);                                 // rows two steps ahead, everything else of a step one ahead
    load_step(nxt, t - 1, r1);                                    // own block only (6 KiB per wave)
    float *sb = p.scratch + ((int64_t)(t * n_ct + tile) * 4) * 256 + j * 256 + lane * 4;
    f32x4 dhh, d_r, d_z, d_n, d_hn;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dhh[i] = (ok ? cur.dh[i] : 0.f) + carry[i];
      const float zz = cur.z[i], nn = cur.nn[i], rr = cur.r[i];
      d_n[i] = dhh[i] * (1.f - zz) * (1.f - nn * nn);
      d_r[i] = d_n[i] * cur.ghn[i] * rr * (1.f - rr);
      d_hn[i] = d_n[i] * rr;
      d_z[i] = dhh[i] * (cur.hm[i] - nn) * zz * (1.f - zz);
    }
    float *e = &ex[t & 1][0][0];
    st4(e + 0 * 1024 + j * 256 + lane * 4, d_r);
    st4(e + 1 * 1024 + j * 256 + lane * 4, d_z);
    st4(e + 2 * 1024 + j * 256 + lane * 4, d_n);
    st4(e + 3 * 1024 + j * 256 + lane * 4, d_hn);
    st4(sb + C_R * CS, d_r);
    st4(sb + C_Z * CS, d_z);
    st4(sb + C_N * CS, d_n);
    st4(sb + C_GHN * CS, d_hn);
    __syncthreads();                                              // (the buffer is rewritten two steps later: one barrier per step)
    // carry block j = W_hh^T [d_r, d_z, d_hn], d x block j = W_ih^T [d_r, d_z, d_n]: two accumulators each (even / odd k-steps)
    f32x4 ca[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, xa[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const f32x4 v = ld4(e + g * 1024 + b * 256 + lane * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ca[i & 1] = mfma16(wh[g][b][i], v[i], ca[i & 1]);
          xa[i & 1] = mfma16(wi[g][b][i], v[i], xa[i & 1]);
        }
      }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const f32x4 vn = ld4(e + 2 * 1024 + b * 256 + lane * 4), vh = ld4(e + 3 * 1024 + b * 256 + lane * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xa[i & 1] = mfma16(wi[2][b][i], vn[i], xa[i & 1]);
        ca[i & 1] = mfma16(wh[2][b][i], vh[i], ca[i & 1]);
      }
    }
    const f32x4 dx = xa[0] + xa[1];
    if constexpr (DXBLK) st4(sb + C_DH * CS, dx);
    else if (ok) {
      const int64_t col = (int64_t)t * p.Nc + cc;
#pragma unroll
      for (int i = 0; i < 4; ++i) p.dxT[(int64_t)(16 * j + 4 * q + i) * B + col] = dx[i];
    }
    const float mk = ok ? cur.mk : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) carry[i] = (ca[0][i] + ca[1][i] + dhh[i] * cur.z[i]) * mk;
    cur = nxt; r1 = r2;
  }
  }
}

// ================================================================================================================================
// weight gradients: workgroup = 2 sets of 4 roles; role = (matrix ih | hh) x (gate rows 0..95 | 96..191); a set walks row tiles
// ================================================================================================================================
struct Gru16WgArgs {
  NetOff off;
  const float *x;
  int x_blocked;
  const float *scratch;
  int L, Nc;
  float *slabs;
  int64_t slab_stride, slab_col0;
};
#define WG16_AS 100      // row stride of the d-gate tile [16][96 (+4)]: 100 = 4 mod 8, transposed reads conflict-free (as RS16)

template <bool XBLK>
__global__ __launch_bounds__(G16_THREADS, 2) void gru16_wgrad_kernel(Gru16WgArgs p) {
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & (WAVE - 1), n = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
  const int set = wave >> 2, role = wave & 3, mat = role >> 1, gh = role & 1;
  float *Ua = lds + wave * (16 * WG16_AS + 16 * RS16), *Ub = Ua + 16 * WG16_AS;
  const int n_ct = (p.Nc + 15) >> 4;
  const int64_t B = (int64_t)p.L * p.Nc;
  const int64_t CS = (int64_t)p.L * n_ct * 1024;
  const int n_rt = p.L * n_ct;
  f32x4 acc[6][4];
  float gb[6];
#pragma unroll
  for (int bf = 0; bf < 6; ++bf) {
    gb[bf] = 0.f;
#pragma unroll
    for (int bk = 0; bk < 4; ++bk) acc[bf][bk] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // gate-row block g = 6 gh + bf (0..11): component of its values and block inside the component
  const float *srcA[6];
#pragma unroll
  for (int bf = 0; bf < 6; ++bf) {
    const int g = 6 * gh + bf;
    const int comp = g < 4 ? C_R : (g < 8 ? C_Z : (mat ? C_GHN : C_N));
    srcA[bf] = p.scratch + comp * CS + (g & 3) * 256 + lane * 4;
  }
  const float *srcB = mat ? p.scratch + C_HM * CS + lane * 4 : p.x + lane * 4;
  const int stride = gridDim.x * 2;
  int rt = blockIdx.x * 2 + set;
  f32x4 va[6], vb[4];
  auto load_tile = [&](int r) {
    const int rr = min(r, n_rt - 1);
    const int64_t off = (int64_t)rr * 1024;
#pragma unroll
    for (int bf = 0; bf < 6; ++bf) va[bf] = ld4(srcA[bf] + off);
    if (mat || XBLK) {
#pragma unroll
      for (int bk = 0; bk < 4; ++bk) vb[bk] = ld4(srcB + off + bk * 256);
    } else {
      const int t = rr / n_ct, c = min((rr - t * n_ct) * 16 + n, p.Nc - 1);
      const int64_t col = (int64_t)t * p.Nc + c;
#pragma unroll
      for (int bk = 0; bk < 4; ++bk)
#pragma unroll
        for (int i = 0; i < 4; ++i) vb[bk][i] = p.x[(int64_t)(16 * bk + 4 * q + i) * B + col];
    }
  };
  load_tile(rt);
  for (; rt < n_rt; rt += stride) {
#pragma unroll
    for (int bf = 0; bf < 6; ++bf) st4(Ua + n * WG16_AS + 16 * bf + 4 * q, va[bf]);
#pragma unroll
    for (int bk = 0; bk < 4; ++bk) st4(Ub + n * RS16 + 16 * bk + 4 * q, vb[bk]);
    load_tile(rt + stride);                                     // next tile: in flight under this tile's products
    wave_lds_sync();
    dw_accum16<6, 4>(acc, gb, Ua, WG16_AS, Ub, RS16, n, q);
    wave_lds_sync();
  }
  // ---- set 1 hands its sums to set 0 through LDS; set 0 writes the workgroup's slab row ----
  __syncthreads();
  float *red = lds + role * (26 * 256);                          // [24 accumulator blocks][256] | 6 bias-partial rows of 64
  if (set == 1) {
#pragma unroll
    for (int bf = 0; bf < 6; ++bf) {
#pragma unroll
      for (int bk = 0; bk < 4; ++bk) st4(red + ((bf * 4 + bk) * 64 + lane) * 4, acc[bf][bk]);
      red[24 * 256 + bf * 64 + lane] = gb[bf];
    }
  }
  __syncthreads();
  if (set == 1) return;
  float *slab = p.slabs + (size_t)blockIdx.x * p.slab_stride + p.slab_col0;
  const int woff = mat ? p.off.gru_whh : p.off.gru_wih, boff = mat ? p.off.gru_bhh : p.off.gru_bih;
#pragma unroll
  for (int bf = 0; bf < 6; ++bf) {
#pragma unroll
    for (int bk = 0; bk < 4; ++bk) {
      const f32x4 o2 = ld4(red + ((bf * 4 + bk) * 64 + lane) * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) slab[woff + (96 * gh + 16 * bf + 4 * q + i) * HID + 16 * bk + n] = acc[bf][bk][i] + o2[i];
    }
    const float bsum = quad_sum16(gb[bf] + red[24 * 256 + bf * 64 + lane]);
    if (q == 0) slab[boff + 96 * gh + 16 * bf + n] = bsum;
  }
}

// ================================================================================================================================
// trunk features of the sequence-tiled minibatch (in_dim <= 64, layer_N <= 1), blocked output: mlp.py:18-55 up to the trunk's last
// LayerNorm.  The rollout-sized features kernel (mlp_fwd16.h, weights in REGISTERS: 256 of them, one wave per SIMD) walks ~10
// tiles per wave at training sizes, each paying its gather and its dependent chain alone: 183 us at BASELINE configs[2].  Here the
// weights sit in LDS in fragment order (shared by 8 waves, 2 per SIMD) and the next tile's rows are fetched under the current
// tile's products.
// ================================================================================================================================
struct Feat16Args {
  const float *params;
  NetOff off;
  const float *x;
  const int32_t *rows;
  int L, Nc, D, fnorm;
  float *out;                 // blocked [L][n_ct][4][256]
};
template <int LN>
struct T16Lds {
  static constexpr int W1 = 0, W2 = W1 + HID * HID, B1 = W2 + (LN > 0 ? HID * HID : 0), B2 = B1 + HID;
  static constexpr int FN_G = B2 + HID, FN_B = FN_G + HID, G1 = FN_B + HID, T1 = G1 + HID, G2 = T1 + HID, T2 = G2 + HID, TOTAL = T2 + HID;
};

// KB1 = ceil(in_dim / 16) is a template parameter (the layer-1 loop has no branches), a row is fetched as KB1 16-byte loads that
// never leave it (ld4_row_raw: clamped start, lanes shifted and zero-filled at the row end by ld4_row_fix — only the last block can
// need it), and everything a tile needs from the argument block is copied to locals first.  The first form (sixteen clamped dword
// loads with 64-bit addresses per lane, a branch per block, 15-17 spilled registers) reloaded spills between the two halves of the
// next tile's fetch: two exposed memory round trips per tile (118 us per network at BASELINE configs[2] against ~50 of work).
template <bool RELU, int LN, int KB1>
__global__ __launch_bounds__(G16_THREADS, 2) void gru16_features_kernel(Feat16Args p) {
  extern __shared__ __align__(16) float lds[];
  typedef T16Lds<LN> M;
  const NetOff &o = p.off;
  const int lane = threadIdx.x & (WAVE - 1), n = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)), nw = blockDim.x / WAVE;
  const int D = p.D;
  constexpr int kb1 = KB1;
  // ---- staging: W1 [64][D] -> fragment blocks (bo, b < kb1), columns >= D zero; W2; vectors (feature-norm vectors zero beyond D) ----
  for (int e = threadIdx.x; e < HID * kb1 * 4; e += blockDim.x) {
    const int f = e / (kb1 * 4), kq = e - f * (kb1 * 4);
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int k = 4 * kq + i; const float w = p.params[o.w1 + f * D + min(k, D - 1)]; v[i] = k < D ? w : 0.f; }
    *reinterpret_cast<float4 *>(lds + M::W1 + (((f >> 4) * kb1 + (kq >> 2)) * 64 + (kq & 3) * 16 + (f & 15)) * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
  if constexpr (LN > 0) stage_frag<2>(lds + M::W2, p.params + o.w2[0], HID, HID, HID);
  for (int e = threadIdx.x; e < 8 * HID; e += blockDim.x) {
    const int which = e >> 6, k = e & 63;
    float v = 0.f;
    if (which == 0) v = p.params[o.b1 + k];
    else if (which == 1) { if (LN > 0) v = p.params[o.b2[0] + k]; }
    else if (which == 2) { if (p.fnorm && k < D) v = p.params[o.fn_w + k]; }
    else if (which == 3) { if (p.fnorm && k < D) v = p.params[o.fn_b + k]; }
    else if (which == 4) v = p.params[o.ln1_w + k];
    else if (which == 5) v = p.params[o.ln1_b + k];
    else if (which == 6) { if (LN > 0) v = p.params[o.ln2_w[0] + k]; }
    else { if (LN > 0) v = p.params[o.ln2_b[0] + k]; }
    lds[M::B1 + e] = v;                                          // B1, B2, FN_G, FN_B, G1, T1, G2, T2 are consecutive
  }
  __syncthreads();
  const int Nc = p.Nc, n_ct = (Nc + 15) >> 4;
  const int64_t n_tiles = (int64_t)p.L * n_ct, stride = (int64_t)gridDim.x * nw;
  const float *const xbase = p.x;
  const int32_t *const rows = p.rows;
  float *const out = p.out;
  const bool fnorm = p.fnorm != 0, al4 = (D & 3) == 0;
  const float inv_D = 1.0f / (float)D;
  int n_pad = 0;                                                  // slots of this lane beyond D (they hold 0)
#pragma unroll
  for (int b = 0; b < KB1; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i) n_pad += (16 * b + 4 * q + i >= D) ? 1 : 0;
  auto fetch = [&](int64_t tl, f32x4 (&xv)[KB1]) {
    const int64_t t = tl / n_ct;
    const int c = (int)(tl - t * n_ct) * 16 + n;
    const int64_t i = t * Nc + (c < Nc ? c : 0);
    int r = (int)i;
    if (rows) r = rows[i];
    const float *src = xbase + (int64_t)r * D;
    int ql = q;
    asm volatile("" : "+v"(ql));                                  // (offsets recomputed per call, not kept as address pairs)
#pragma unroll
    for (int b = 0; b < KB1; ++b) xv[b] = ld4_row_raw(src, 16 * b + 4 * ql, D);
  };
  int64_t tile = (int64_t)blockIdx.x * nw + wave;
  if (tile >= n_tiles) return;
  f32x4 xn[KB1];
  fetch(tile, xn);
  for (; tile < n_tiles; tile += stride) {
    f32x4 x[KB1];
#pragma unroll
    for (int b = 0; b < KB1; ++b) x[b] = xn[b];
    x[KB1 - 1] = ld4_row_fix(x[KB1 - 1], 16 * (KB1 - 1) + 4 * q, D, al4);      // (blocks before the last are inside the row)
    fetch(min(tile + stride, n_tiles - 1), xn);
    if (fnorm) {
      float s = 0.f;
#pragma unroll
      for (int b = 0; b < KB1; ++b) s += (x[b][0] + x[b][1]) + (x[b][2] + x[b][3]);
      const float mean = quad_sum16(s) * inv_D;
      float v = 0.f;
#pragma unroll
      for (int b = 0; b < KB1; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float cdev = x[b][i] - mean; x[b][i] = cdev; v += cdev * cdev; }
      v -= (float)n_pad * mean * mean;                            // the padded slots contributed (0 - mean)^2 each
      const float rstd = 1.0f / sqrtf(fmaxf(quad_sum16(v), 0.f) * inv_D + LN_EPS);
#pragma unroll
      for (int b = 0; b < KB1; ++b) {
        const f32x4 g = ld4(lds + M::FN_G + 16 * b + 4 * q), t = ld4(lds + M::FN_B + 16 * b + 4 * q);
        x[b] = x[b] * f32x4{rstd, rstd, rstd, rstd} * g + t;     // gamma = beta = 0 beyond D: padded slots are exactly 0
      }
    }
    f32x4 h[4];
#pragma unroll
    for (int bo = 0; bo < 4; ++bo) h[bo] = ld4(lds + M::B1 + 16 * bo + 4 * q);
    {
      const float *base = lds + M::W1 + lane * 4;
#pragma unroll
      for (int b = 0; b < KB1; ++b) {
        f32x4 a[4];
#pragma unroll
        for (int bo = 0; bo < 4; ++bo) a[bo] = ld4(base + (bo * kb1 + b) * 256);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int bo = 0; bo < 4; ++bo) h[bo] = mfma16(a[bo][i], x[b][i], h[bo]);
      }
    }
    act_ln16<RELU>(h, lds + M::G1, lds + M::T1, q);
    if constexpr (LN > 0) {
      f32x4 h2[4];
#pragma unroll
      for (int bo = 0; bo < 4; ++bo) h2[bo] = ld4(lds + M::B2 + 16 * bo + 4 * q);
      frag_mma4<4>(h2, lds + M::W2, 0, h, lane);
      act_ln16<RELU>(h2, lds + M::G2, lds + M::T2, q);
#pragma unroll
      for (int b = 0; b < 4; ++b) h[b] = h2[b];
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) st4(out + (tile * 4 + b) * 256 + lane * 4, h[b]);
  }
}

// ================================================================================================================================
// host
// ================================================================================================================================
#ifndef NUM_CU
#define NUM_CU 256
#endif
static int check_rec16(const mappo_net_desc *d, const char *who) {
  MAPPO_REQUIRE(d && d->recurrent, "%s: needs a recurrent network descriptor", who);
  MAPPO_REQUIRE(d->hidden == HID, "%s: hidden_size %d unsupported", who, d->hidden);
  MAPPO_REQUIRE(d->out_dim >= 1 && d->out_dim <= MAPPO_MAX_ACTIONS, "%s: out_dim %d", who, d->out_dim);
  return MAPPO_OK;
}
// Waves per workgroup of the sequence kernels.  Every workgroup stages its own ~100 KB of weights, and only one fits a CU, so few
// fat workgroups amortise the staging while thin ones spread a small tile count over more CUs.  Measured (train() of configs
// 2-rmappo / 3 / 4, ms; scripts/phase_split.py with MAPPO_GRU16_WAVES): 8 waves 5.86 / 23.0 / 12.9, 4 waves 4.95 / 24.9 / 12.8,
// 2 waves 6.47 / 32.8 / 15.5, 1 wave 9.6 / 50.9 / 21.1 — so: 4 waves up to 256 x 4 tiles, growing to 8 with the tile count.
static int seq_waves(int Nc) {
  const int n_ct = (Nc + 15) / 16;
  int w = (n_ct + NUM_CU - 1) / NUM_CU;
  w = w < 4 ? (n_ct < 4 ? n_ct : 4) : w;
  if (const char *e = getenv("MAPPO_GRU16_WAVES")) w = atoi(e);      // diagnostic override (scripts/phase_split.py A/B)
  return w < 1 ? 1 : (w > G16_WAVES ? G16_WAVES : w);
}
// Tile count up to which a tile's step is split over four waves (gru16s_* + gru16_head_kernel).  Measured train() ms, unsplit /
// split (scripts/phase_split_n.py with MAPPO_GRU16_SPLIT_TILES[_BWD] = 0 | 4096): config-2 rmappo (384 tiles) 4.80 / 4.69; config 4
// at 64 threads (1600 tiles) 12.28 / 12.09; config 3 (1920 tiles) 22.88 / 23.56 — with >= 4 tiles per CU the unsplit kernels already
// keep two waves on every SIMD and both forms are bound by the same MFMA + gate arithmetic, the split one paying four x-tile reads
// and a barrier per step on top.  MAPPO_GRU16_SPLIT_TILES / _BWD: A/B overrides.
static bool seq_split(int Nc, bool bwd) {
  int thr = 1024;
  if (const char *e = getenv(bwd ? "MAPPO_GRU16_SPLIT_TILES_BWD" : "MAPPO_GRU16_SPLIT_TILES")) thr = atoi(e);
  return (Nc + 15) / 16 <= thr;
}
static int split_grid(int Nc) {                                   // two workgroups of the split kernels fit a CU (registers)
  const int n_ct = (Nc + 15) / 16;
  int cap = 2 * NUM_CU;
  if (const char *e = getenv("MAPPO_GRU16_SPLIT_GRID")) cap = atoi(e);
  return n_ct < cap ? n_ct : cap;
}
static int head_grid(int L, int Nc) {
  const int n_rt = L * ((Nc + 15) / 16), want = (n_rt + 3) / 4;
  return want < NUM_CU ? want : NUM_CU;
}
static int seq_grid(int Nc) {
  const int n_ct = (Nc + 15) / 16, nw = seq_waves(Nc), want = (n_ct + nw - 1) / nw;
  return want < NUM_CU ? want : NUM_CU;
}
static int wg_grid(int L, int Nc) {
  const int n_rt = L * ((Nc + 15) / 16), want = (n_rt + 1) / 2;
  return want < NUM_CU ? want : NUM_CU;
}

extern "C" int64_t mappo_gru16_scratch_floats(int32_t L, int32_t Nc) { return (int64_t)G16_COMPS * L * ((Nc + 15) / 16) * 1024; }
extern "C" int64_t mappo_gru16_blocked_floats(int32_t L, int32_t Nc) { return (int64_t)L * ((Nc + 15) / 16) * 1024; }
extern "C" int32_t mappo_gru16_slabs(int32_t L, int32_t Nc) {
  const int a = seq_grid(Nc), b = wg_grid(L, Nc);                 // (head_grid(L, Nc) <= wg_grid(L, Nc))
  return a > b ? a : b;
}

int mlp_features_blocked_wide_(const float *params, const mappo_net_desc *desc, const float *x, const int32_t *rows, int64_t B,
                               float *out_blocked, mappo_stream_t stream);      // mlp.hip
template <bool RELU, int LN>
static int feat16_launch(const Feat16Args &a, dim3 grid, dim3 block, hipStream_t st) {
  const size_t lb = (size_t)T16Lds<LN>::TOTAL * sizeof(float);
  switch ((a.D + 15) / 16) {
    case 1: hipLaunchKernelGGL((gru16_features_kernel<RELU, LN, 1>), grid, block, lb, st, a); break;
    case 2: hipLaunchKernelGGL((gru16_features_kernel<RELU, LN, 2>), grid, block, lb, st, a); break;
    case 3: hipLaunchKernelGGL((gru16_features_kernel<RELU, LN, 3>), grid, block, lb, st, a); break;
    default: hipLaunchKernelGGL((gru16_features_kernel<RELU, LN, 4>), grid, block, lb, st, a); break;
  }
  return MAPPO_OK;
}

extern "C" int mappo_mlp_features_seq(const float *params, const mappo_net_desc *desc, const float *x, const int32_t *rows,
                                      int32_t L, int32_t Nc, float *out_blocked, mappo_stream_t stream) {
  MAPPO_REQUIRE(desc && desc->hidden == HID, "mlp_features_seq: hidden_size unsupported");
  MAPPO_REQUIRE(params && x && out_blocked && L > 0 && Nc > 0, "mlp_features_seq: bad arguments");
  if (desc->in_dim > MAXD)       // wide inputs (Nc % 16 == 0): the one-launch wide forward writes the same blocked form (mlp.hip)
    return mlp_features_blocked_wide_(params, desc, x, rows, (int64_t)L * Nc, out_blocked, stream);
  MAPPO_REQUIRE(desc->in_dim >= 4 && desc->layer_N >= 0 && desc->layer_N <= 1,
                "mlp_features_seq: in_dim %d < 4 / layer_N %d take mappo_mlp_features (feature-major)", desc->in_dim, desc->layer_N);
  MAPPO_CLEAR_STICKY();
  Feat16Args a = {};
  a.params = params; a.off = net_offsets(*desc); a.x = x; a.rows = rows; a.L = L; a.Nc = Nc; a.D = desc->in_dim;
  a.fnorm = desc->use_feature_norm; a.out = out_blocked;
  const int64_t n_tiles = (int64_t)L * ((Nc + 15) / 16);
  const int nw = n_tiles >= 8 * NUM_CU ? 8 : (n_tiles >= 4 * NUM_CU ? 4 : (n_tiles >= 4 ? 4 : 1));
  int64_t nb = (n_tiles + nw - 1) / nw;
  int64_t cap = NUM_CU;                                          // (two / four workgroups per CU fit now — measured no better: 21.9 / 22.0 / 22.1 ms
  if (const char *e = getenv("MAPPO_FEAT16_GRID")) cap = atoll(e);      //  of config-3 train at 256 / 512 / 1024; A/B override)
  if (nb > cap) nb = cap;
  const dim3 grid((unsigned)nb), block(WAVE * nw);
  const bool relu = desc->use_relu != 0;
  if (desc->layer_N == 0) { if (relu) feat16_launch<true, 0>(a, grid, block, as_stream(stream)); else feat16_launch<false, 0>(a, grid, block, as_stream(stream)); }
  else { if (relu) feat16_launch<true, 1>(a, grid, block, as_stream(stream)); else feat16_launch<false, 1>(a, grid, block, as_stream(stream)); }
  MAPPO_CHECK_LAUNCH("mlp_features_seq");
  return MAPPO_OK;
}

template <int HEAD, int NBH, bool XBLK>
static int fwd16_launch(const Gru16Args &a, dim3 grid, hipStream_t st) {
  typedef F16Lds<HEAD, NBH> M;
  static const hipError_t e_ = hipFuncSetAttribute((const void *)gru16_fwd_kernel<HEAD, NBH, XBLK>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
  if (e_ != hipSuccess) { mappo_set_error("gru16_forward_loss: hipFuncSetAttribute: %s", hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  const int nw = seq_waves(a.Nc);
  hipLaunchKernelGGL((gru16_fwd_kernel<HEAD, NBH, XBLK>), grid, dim3(WAVE * nw), (size_t)M::total(nw) * sizeof(float), st, a);
  return MAPPO_OK;
}

extern "C" int mappo_gru16_forward_loss(const float *params, const mappo_net_desc *desc, const float *x, int32_t x_blocked, const float *h0,
                                        const int32_t *h0_rows, const float *masks, const int32_t *rows, int32_t L, int32_t Nc, int32_t head,
                                        const float *avail, const float *actions, const float *old_logp, const float *adv, const float *active,
                                        const float *v_old, const float *returns, const float *vn_state, const double *mb_moments,
                                        const mappo_ppo_cfg *cfg, float *scratch, float *slabs, int64_t slab_stride, int64_t slab_col0,
                                        double *partials, mappo_stream_t stream) {
  if (int rc = check_rec16(desc, "gru16_forward_loss")) return rc;
  MAPPO_REQUIRE(params && x && h0 && masks && scratch && slabs && partials && mb_moments && cfg && active && L > 0 && Nc > 0,
                "gru16_forward_loss: bad arguments");
  MAPPO_REQUIRE(head == 1 || head == 2, "gru16_forward_loss: head %d", head);
  MAPPO_REQUIRE(head == 1 ? (actions && old_logp && adv) : (v_old && returns && desc->out_dim == 1), "gru16_forward_loss: loss inputs");
  MAPPO_REQUIRE(!cfg->use_valuenorm || head == 1 || vn_state, "gru16_forward_loss: use_valuenorm needs vn_state");
  MAPPO_CLEAR_STICKY();
  Gru16Args a = {};
  a.params = params; a.off = net_offsets(*desc); a.x = x; a.x_blocked = x_blocked; a.h0 = h0; a.h0_rows = h0_rows; a.masks = masks; a.rows = rows;
  a.L = L; a.Nc = Nc; a.A = desc->out_dim; a.scratch = scratch; a.avail = avail; a.actions = actions; a.old_logp = old_logp; a.adv = adv;
  a.active = active; a.v_old = v_old; a.returns = returns; a.vn_state = vn_state; a.mb_moments = mb_moments; a.cfg = *cfg;
  a.slabs = slabs; a.slab_stride = slab_stride; a.slab_col0 = slab_col0; a.partials = partials;
  MAPPO_REQUIRE(slab_col0 >= 0 && slab_col0 + a.off.total <= slab_stride, "gru16_forward_loss: slab column range");
  const dim3 grid((unsigned)seq_grid(Nc));
  const hipStream_t st = as_stream(stream);
  int rc;
  if (seq_split(Nc, false)) {                                     // recurrence split over four waves per tile, then the row-parallel tail
    const dim3 g1((unsigned)split_grid(Nc)), g2((unsigned)head_grid(L, Nc)), blk(4 * WAVE);
    if (x_blocked) hipLaunchKernelGGL(gru16s_fwd_kernel<true>, g1, blk, 0, st, a);
    else hipLaunchKernelGGL(gru16s_fwd_kernel<false>, g1, blk, 0, st, a);
    if (head == 2) hipLaunchKernelGGL((gru16_head_kernel<2, 1>), g2, blk, 0, st, a);
    else if (a.A <= 16) hipLaunchKernelGGL((gru16_head_kernel<1, 1>), g2, blk, 0, st, a);
    else hipLaunchKernelGGL((gru16_head_kernel<1, 2>), g2, blk, 0, st, a);
    MAPPO_CHECK_LAUNCH("gru16_forward_loss");
    return MAPPO_OK;
  }
  if (head == 2) rc = x_blocked ? fwd16_launch<2, 1, true>(a, grid, st) : fwd16_launch<2, 1, false>(a, grid, st);
  else if (a.A <= 16) rc = x_blocked ? fwd16_launch<1, 1, true>(a, grid, st) : fwd16_launch<1, 1, false>(a, grid, st);
  else rc = x_blocked ? fwd16_launch<1, 2, true>(a, grid, st) : fwd16_launch<1, 2, false>(a, grid, st);
  if (rc) return rc;
  MAPPO_CHECK_LAUNCH("gru16_forward_loss");
  return MAPPO_OK;
}

template <bool DXBLK, bool PRE>
static int bwd16_launch(const Gru16Args &a, dim3 grid, int nw, size_t lds_bytes, hipStream_t st) {
  void (*const fn)(Gru16Args) = PRE ? gru16_bwd4_kernel<DXBLK> : gru16_bwd_kernel<DXBLK>;
  static const hipError_t e_ = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
  if (e_ != hipSuccess) { mappo_set_error("gru16_backward: hipFuncSetAttribute: %s", hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  hipLaunchKernelGGL(fn, grid, dim3(WAVE * nw), lds_bytes, st, a);
  return MAPPO_OK;
}

extern "C" int mappo_gru16_backward(const float *params, const mappo_net_desc *desc, const float *masks, const int32_t *rows, int32_t L,
                                    int32_t Nc, float *scratch, float *dxT, mappo_stream_t stream) {
  if (int rc = check_rec16(desc, "gru16_backward")) return rc;
  MAPPO_REQUIRE(params && masks && scratch && L > 0 && Nc > 0, "gru16_backward: bad arguments");
  MAPPO_CLEAR_STICKY();
  Gru16Args a = {};
  a.params = params; a.off = net_offsets(*desc); a.masks = masks; a.rows = rows; a.L = L; a.Nc = Nc; a.scratch = scratch; a.dxT = dxT;
  const dim3 grid((unsigned)seq_grid(Nc));
  const size_t lds_bytes = (size_t)2 * G16_NG * HID * sizeof(float);
  const hipStream_t st = as_stream(stream);
  if (seq_split(Nc, true)) {
    const dim3 g1((unsigned)split_grid(Nc)), blk(4 * WAVE);
    if (dxT) hipLaunchKernelGGL(gru16s_bwd_kernel<false>, g1, blk, 0, st, a);
    else hipLaunchKernelGGL(gru16s_bwd_kernel<true>, g1, blk, 0, st, a);
    MAPPO_CHECK_LAUNCH("gru16_backward");
    return MAPPO_OK;
  }
  const int nw = seq_waves(Nc);
  const int rc = nw <= 4 ? (dxT ? bwd16_launch<false, true>(a, grid, nw, lds_bytes, st) : bwd16_launch<true, true>(a, grid, nw, lds_bytes, st))
                         : (dxT ? bwd16_launch<false, false>(a, grid, nw, lds_bytes, st) : bwd16_launch<true, false>(a, grid, nw, lds_bytes, st));
  if (rc) return rc;
  MAPPO_CHECK_LAUNCH("gru16_backward");
  return MAPPO_OK;
}

extern "C" int mappo_gru16_wgrad(const mappo_net_desc *desc, const float *x, int32_t x_blocked, const float *scratch, int32_t L, int32_t Nc,
                                 float *slabs, int64_t slab_stride, int64_t slab_col0, mappo_stream_t stream) {
  if (int rc = check_rec16(desc, "gru16_wgrad")) return rc;
  MAPPO_REQUIRE(x && scratch && slabs && L > 0 && Nc > 0, "gru16_wgrad: bad arguments");
  MAPPO_CLEAR_STICKY();
  Gru16WgArgs a = {};
  a.off = net_offsets(*desc); a.x = x; a.x_blocked = x_blocked; a.scratch = scratch; a.L = L; a.Nc = Nc;
  a.slabs = slabs; a.slab_stride = slab_stride; a.slab_col0 = slab_col0;
  MAPPO_REQUIRE(slab_col0 >= 0 && slab_col0 + a.off.total <= slab_stride, "gru16_wgrad: slab column range");
  const dim3 grid((unsigned)wg_grid(L, Nc));
  // per-wave operand tiles; the epilogue's hand-over buffer (4 roles x 26 blocks of 256 floats) overlays them
  const size_t tiles = (size_t)G16_WAVES * (16 * WG16_AS + 16 * RS16), epi = (size_t)4 * 26 * 256;
  const size_t lds_bytes = (tiles > epi ? tiles : epi) * sizeof(float);
  if (x_blocked) {
    static const hipError_t e_ = hipFuncSetAttribute((const void *)gru16_wgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    if (e_ != hipSuccess) { mappo_set_error("gru16_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
    hipLaunchKernelGGL(gru16_wgrad_kernel<true>, grid, dim3(G16_THREADS), lds_bytes, as_stream(stream), a);
  } else {
    static const hipError_t e_ = hipFuncSetAttribute((const void *)gru16_wgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    if (e_ != hipSuccess) { mappo_set_error("gru16_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
    hipLaunchKernelGGL(gru16_wgrad_kernel<false>, grid, dim3(G16_THREADS), lds_bytes, as_stream(stream), a);
  }
  MAPPO_CHECK_LAUNCH("gru16_wgrad");
  return MAPPO_OK;
}
