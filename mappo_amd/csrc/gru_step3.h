// gru_step3.h — one rollout / evaluation step of the recurrent layer for a 16-row tile per 4-wave workgroup, shared by gru.hip
// (trunk features from HBM, or the narrow-input trunk fused in) and by the wide-input fused step (mlp_wide16.h: the split-K trunk
// of the same workgroup hands its output over through LDS).  rnn.py:25-29 (single step), r_actor_critic.py:43-70,146-165.
#pragma once
#include "mlp_core.h"
#include "mlp_trunk16r.h"

#define GS 193              // LDS row stride of the GRU weights (k-major: sW[k*GS + g], g < 192)
#define NG 192

// ---- LDS maps -----------------------------------------------------------------------------------------
struct GruLds {
  int wih, whh, wh, bih, bhh, nw, nb, bh, tiles, wave_stride, total;
};
__host__ __device__ inline GruLds gru_lds(int n_waves, int wave_rows, bool with_ih = true) {
  GruLds m;
  int p = 0;
  m.wih = p; if (with_ih) p = al4(p + HID * GS);          // not resident when the input products run in their own kernels
  m.whh = p; p = al4(p + HID * GS);
  m.wh = p; p = al4(p + HID * HP);
  m.bih = p; p += NG; m.bhh = p; p += NG;
  m.nw = p; p += HID; m.nb = p; p += HID;
  m.bh = p; p += 32;
  m.tiles = p;
  m.wave_stride = al4(wave_rows * TP);
  p += n_waves * m.wave_stride;
  m.total = p;
  return m;
}
__host__ __device__ inline int al4(int p);


// Gate nonlinearities on the hardware transcendental units (v_exp_f32 / v_rcp_f32, 1 ulp each): ~5 instructions instead of
// the ~30 (sigmoid) / ~50 (tanh) of the libm forms — per step and lane the recurrences evaluate 32 + 16 of them back to
// back.  Absolute error ~1e-7, against the 1e-5 parity bar (tests/test_gpu_gru.py compares every use with the oracle).
#ifndef GRU_LIBM_GATES
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x)); }
#else
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return tanhf(x); }
#endif


struct GruFwdArgs {
  const float *params;
  NetOff off;
  GruLds map;
  const float *xT;            // [64][B] trunk features, B = L*Nc, column t*Nc + c
  const float *h0;            // [.][64] row-major initial states
  const int32_t *h0_rows;     // [Nc] or NULL (identity)
  const float *masks;         // buffer-order masks, indexed by rows[t*Nc + c] (NULL rows = identity)
  const int32_t *rows;
  int L, Nc, A, head_mode;    // head_mode 0: none, 1: out[B][A], 2: sample (actions/logp [B])
  int tile_waves;             // waves of a workgroup that own tiles; any further waves only help staging the weights
  float *h_last;              // [Nc][64] row-major or NULL
  float *out;
  const float *avail;         // [B][A] minibatch order, or NULL
  float *actions, *logp;
  int deterministic;
  uint64_t seed, counter;
  const uint64_t *counter_dev;
  // fused rollout step (gru_step3f_*): the trunk runs in the same launch on the rows x_rows [Nc][in_dim]
  const float *x_rows;
  mappo_net_desc desc;
  // fused insert + step (mappo_recurrent_rollout_step): the row mask is derived from the env's `dones` of the step before
  // (mask = 0 where ALL agents of the row's env report done, smac_runner.py:132-138) instead of read from the buffer slot the
  // insert role of the same launch is writing
  const uint8_t *dones;
  int done_M;
  int64_t done_sn, done_sm;
};


// ---- rollout step, third form: one 16-row tile per 4-wave workgroup, the HIDDEN UNITS split over the waves -------------------
// The two-waves-per-32-rows step above spends its time in the prologue (96 KB of GRU weights through LDS per workgroup) and in
// 192 32x32x2 MFMAs per wave on one wave per SIMD: 19 us for a step whose arithmetic is tiny.  Here wave w of a workgroup owns
// hidden units [16 w, 16 w + 16) of ONE 16-row tile on v_mfma_f32_16x16x4_f32:
//   * its A operands are the 3 x 16 rows of W_ih and W_hh that produce those units' r / z / n gates, read STRAIGHT from global
//     memory as 16-byte loads (24 per lane; every weight is used by exactly one wave: no LDS staging, no staging barrier);
//   * the B operands are the tile's trunk features (feature-major xT: 16 dwords per lane) and h * mask (row-major: four 16-byte
//     loads per lane); k-step (b, i) takes k = 16 b + 4 q + i, so for b = w the h operand IS the lane's own units' previous state;
//   * 96 MFMAs per wave, gates and the state update per lane (4 units x 1 row), the new state goes out as 16-byte stores;
//   * LayerNorm over the 64 units: two 4-wave exchanges of 16 floats (exact two-pass form); the head is split over k the same
//     way (each wave multiplies its 16 normalised units), partial logits meet in LDS, wave 0 masks / samples / writes.
// All loads of a workgroup are issued before anything waits: one memory latency per step.
struct Step3Shared {
  float sS[4][16], sV[4][16];
  float4 sZ[3][2][64];                                        // partial logits of waves 1..3
  float tZ[16][36];                                           // logits [sample][action] for the categorical epilogue
};
typedef float g4_t __attribute__((ext_vector_type(4)));
typedef float g4u_t __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ g4_t g4_load(const float *p) { const g4u_t v = *reinterpret_cast<const g4u_t *>(p); g4_t r; r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; r[3] = v[3]; return r; }
__device__ __forceinline__ g4_t mfma16g(float a, float b, g4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float quad_sum16g(float v) { return xhalf_sum(xrow_sum(v)); }     // lanes n, n + 16, n + 32, n + 48

// TR: 0 = trunk features from xT | 1 = fused, ReLU trunk | 2 = fused, tanh trunk;  TLN: the trunk's layer_N (fused only).
// Fused: every wave of the workgroup runs the (narrow-input) trunk of the tile itself with the weights in registers
// (mlp_trunk16r.h) — the trunk's output in the accumulator layout IS the B operand of the W_ih products, so there is nothing to
// exchange — and the separate features launch (a ~10 us floor per step) disappears.
// this wave's operands of a step: loaded up front (all requests before the first wait), then any number of tiles
template <int TLN>
struct Step3W {
  g4_t wi[3][4], wh[3][4];
  Trunk16R<TLN> tw;
  g4_t bir, biz, bin, bhr, bhz, bhn, nw4, nb4, hw[2];
};
template <int TR, int TLN>
__device__ __forceinline__ void gru_step3_load(Step3W<TLN> &W, const GruFwdArgs &p, const int w, const int n, const int q) {
  const int A = p.A;
  g4_t (&wi)[3][4] = W.wi, (&wh)[3][4] = W.wh;
  {
    const float *ri = p.params + p.off.gru_wih + (size_t)(16 * w + n) * HID + 4 * q;
    const float *rh = p.params + p.off.gru_whh + (size_t)(16 * w + n) * HID + 4 * q;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int b = 0; b < 4; ++b) { wi[g][b] = g4_load(ri + g * HID * HID + 16 * b); wh[g][b] = g4_load(rh + g * HID * HID + 16 * b); }
  }
  if constexpr (TR == 1 || TR == 2) trunk16r_load<TLN>(W.tw, p.params, p.off, p.desc, n, q);
  const int u0 = 16 * w + 4 * q;                                // this lane's four hidden units
  W.bir = g4_load(p.params + p.off.gru_bih + u0); W.biz = g4_load(p.params + p.off.gru_bih + HID + u0); W.bin = g4_load(p.params + p.off.gru_bih + 2 * HID + u0);
  W.bhr = g4_load(p.params + p.off.gru_bhh + u0); W.bhz = g4_load(p.params + p.off.gru_bhh + HID + u0); W.bhn = g4_load(p.params + p.off.gru_bhh + 2 * HID + u0);
  W.nw4 = g4_load(p.params + p.off.rn_w + u0); W.nb4 = g4_load(p.params + p.off.rn_b + u0);
  // head rows a = 16 bo + n (zero beyond A), columns = this lane's units
#pragma unroll
  for (int bo = 0; bo < 2; ++bo) {
    const int a = 16 * bo + n;
    const g4_t t = g4_load(p.params + p.off.wh + (size_t)min(a, A - 1) * HID + u0);
    W.hw[bo] = a < A ? t : g4_t{0.f, 0.f, 0.f, 0.f};
  }
}

// TR 3: the tile's trunk features arrive through LDS, xs[b][lane] = features 16 b + 4 q .. + 3 of row n (written by the wave that
// ran the trunk of a wide-input network in the same workgroup, behind a barrier: mlp_wide16.h, wide_recurrent_step_dual_kernel)
template <int HM, int TR, int TLN>
__device__ __forceinline__ void gru_step3_tiles(const Step3W<TLN> &W, const GruFwdArgs &p, Step3Shared &sh, const int bid, const int nb,
                                                const float4 *xs = nullptr) {
  const int lane = threadIdx.x & (WAVE - 1), w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)), n = lane & 15, q = lane >> 4;
  const int n_tiles = (p.Nc + 15) / 16;
  const int64_t B = p.Nc;                                       // L == 1: column = sequence
  const int A = p.A;
  const int u0 = 16 * w + 4 * q;
  const g4_t (&wi)[3][4] = W.wi, (&wh)[3][4] = W.wh;
  const Trunk16R<TLN> &tw = W.tw;
  const g4_t bir = W.bir, biz = W.biz, bin = W.bin, bhr = W.bhr, bhz = W.bhz, bhn = W.bhn, nw4 = W.nw4, nb4 = W.nb4;
  const g4_t hw[2] = {W.hw[0], W.hw[1]};
  for (int tile = bid; tile < n_tiles; tile += nb) {
    const int c = tile * 16 + n;
    const bool ok = c < p.Nc;
    const int cc = ok ? c : 0;
    const int64_t hrow = p.h0_rows ? (int64_t)p.h0_rows[cc] : (int64_t)cc;
    float mk;
    if (p.dones) {
      const int env = cc / p.done_M;
      bool all = true;
      for (int m = 0; m < p.done_M; ++m) all = all && p.dones[env * p.done_sn + m * p.done_sm] != 0;
      mk = (ok && !all) ? 1.f : 0.f;
    } else {
      mk = ok ? p.masks[p.rows ? (int64_t)p.rows[cc] : (int64_t)cc] : 0.f;
    }
    uint32_t dead = 0u;
    if (HM == 2 && p.avail && w == 0 && q == 0) dead = avail_dead_mask(p.avail + (int64_t)cc * A, A);    // the sampling lanes
    g4_t x[4], hm[4];
    if constexpr (TR == 3) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float4 t = xs[b * 64 + lane];
        x[b][0] = t.x; x[b][1] = t.y; x[b][2] = t.z; x[b][3] = t.w;
        hm[b] = g4_load(p.h0 + hrow * HID + 16 * b + 4 * q);
      }
    } else if constexpr (TR != 0) {
      const int D = p.desc.in_dim;
      const float *xr = p.x_rows + (int64_t)cc * D;
      f32x4 xin[4];
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) xin[b][i] = xr[min(16 * b + 4 * q + i, D - 1)];
#pragma unroll
      for (int b = 0; b < 4; ++b) hm[b] = g4_load(p.h0 + hrow * HID + 16 * b + 4 * q);
      trunk16r_apply<TR == 1, TLN>(tw, xin, x, D, ok, p.desc.use_feature_norm != 0, q);
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[b][i] = p.xT[(int64_t)(16 * b + 4 * q + i) * B + cc];
        hm[b] = g4_load(p.h0 + hrow * HID + 16 * b + 4 * q);
      }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { x[b][i] = ok ? x[b][i] : 0.f; hm[b][i] *= mk; }
    }
    g4_t ar = bir + bhr, az = biz + bhz, ain = bin, ahn = bhn;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ar = mfma16g(wi[0][b][i], x[b][i], ar);
        az = mfma16g(wi[1][b][i], x[b][i], az);
        ain = mfma16g(wi[2][b][i], x[b][i], ain);
        ar = mfma16g(wh[0][b][i], hm[b][i], ar);
        az = mfma16g(wh[1][b][i], hm[b][i], az);
        ahn = mfma16g(wh[2][b][i], hm[b][i], ahn);
      }
    // the lane's own units' previous state = the h operand of k-block b == w
    const g4_t hprev = w == 0 ? hm[0] : (w == 1 ? hm[1] : (w == 2 ? hm[2] : hm[3]));
    g4_t h;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float gr = sigmoidf_(ar[i]), gz = sigmoidf_(az[i]);
      const float gn = tanhf_(ain[i] + gr * ahn[i]);
      h[i] = (1.f - gz) * gn + gz * hprev[i];
    }
    if (p.h_last && ok) {
      g4u_t o; o[0] = h[0]; o[1] = h[1]; o[2] = h[2]; o[3] = h[3];
      *reinterpret_cast<g4u_t *>(p.h_last + (int64_t)c * HID + u0) = o;
    }
    // ---- LayerNorm over the 64 units of a row (rnn.py:22,79): exact two-pass, two exchanges ----
    const float ps = quad_sum16g((h[0] + h[1]) + (h[2] + h[3]));
    if (q == 0) sh.sS[w][n] = ps;
    __syncthreads();
    const float mean = ((sh.sS[0][n] + sh.sS[1][n]) + (sh.sS[2][n] + sh.sS[3][n])) * (1.f / HID);
    g4_t d = h - g4_t{mean, mean, mean, mean};
    const float pv = quad_sum16g((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
    if (q == 0) sh.sV[w][n] = pv;
    __syncthreads();
    const float rstd = 1.0f / sqrtf(((sh.sV[0][n] + sh.sV[1][n]) + (sh.sV[2][n] + sh.sV[3][n])) * (1.f / HID) + LN_EPS);
    const g4_t y = d * g4_t{rstd, rstd, rstd, rstd} * nw4 + nb4;
    // ---- head, split over k: this wave's 16 units (k-step i <-> unit 16 w + 4 q + i) ----
    g4_t z[2] = {g4_t{0.f, 0.f, 0.f, 0.f}, g4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      z[0] = mfma16g(hw[0][i], y[i], z[0]);
      if (A > 16) z[1] = mfma16g(hw[1][i], y[i], z[1]);
    }
    if (w > 0) {
      sh.sZ[w - 1][0][lane] = make_float4(z[0][0], z[0][1], z[0][2], z[0][3]);
      sh.sZ[w - 1][1][lane] = make_float4(z[1][0], z[1][1], z[1][2], z[1][3]);
    }
    __syncthreads();
    if (w == 0) {
      // lane (n, q) holds logits of actions 16 bo + 4 q + i for row n
#pragma unroll
      for (int bo = 0; bo < 2; ++bo) {
#pragma unroll
        for (int sw = 0; sw < 3; ++sw) {
          const float4 t = sh.sZ[sw][bo][lane];
          z[bo][0] += t.x; z[bo][1] += t.y; z[bo][2] += t.z; z[bo][3] += t.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int a = 16 * bo + 4 * q + i;
          if (a < A) {
            const float v = z[bo][i] + p.params[p.off.bh + a];
            if (HM == 1) { if (ok) p.out[(int64_t)c * A + a] = v; }
            else sh.tZ[n][a] = v;
          }
        }
      }
      if (HM == 2) {
        wave_lds_sync();
        if (q == 0 && ok) {
          const uint64_t ctr = p.counter + (p.counter_dev ? *p.counter_dev : 0ull);
          float action, logp;
          categorical_act_mask(&sh.tZ[n][0], A, dead, p.deterministic != 0, p.seed, ctr, (uint64_t)c, action, logp);
          p.actions[c] = action;
          p.logp[c] = logp;
        }
      }
    }
    __syncthreads();                                            // the exchange buffers are rewritten by the next tile
  }
}

template <int HM, int TR = 0, int TLN = 0>
__device__ __forceinline__ void gru_step3_body(const GruFwdArgs &p, Step3Shared &sh, const int bid, const int nb) {
  const int lane = threadIdx.x & (WAVE - 1), w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)), n = lane & 15, q = lane >> 4;
  if (bid >= (p.Nc + 15) / 16) return;
  Step3W<TLN> W;
  gru_step3_load<TR, TLN>(W, p, w, n, q);
  gru_step3_tiles<HM, TR, TLN>(W, p, sh, bid, nb);
}

