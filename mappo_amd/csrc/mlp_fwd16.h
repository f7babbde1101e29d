// mlp_fwd16.h — rollout forward on v_mfma_f32_16x16x4_f32: one wave per 16 samples, the whole trunk in registers.
//
// The rollout step runs the networks on a few thousand rows (3 072 at BASELINE config 2): one tile per wave, so what
// counts is the LENGTH of the dependent chain, not throughput.  With the 32x32x2 tiles of mlp_forward_kernel a layer is
// 32 k-steps, each an LDS round trip (weights and the previous layer's xhat tile) before its two MFMAs.  Here:
//   * 16x16x4 MFMAs: a layer is 16 k-groups of four independent accumulators (the 4 x 16 output features);
//   * in the 16x16 accumulator layout lane (j = lane & 15, q = lane >> 4) holds features 16 b + 4 q + r (b, r = 0..3) of
//     sample j.  The B operand of a k-group needs ONE feature per q for sample j — and the reduction order over k is
//     free, so k-group (b, r) takes k = 16 b + 4 q + r: exactly the value the lane already holds.  The activations never
//     leave the registers: no xhat tiles, no LDS synchronisation inside the trunk; only the weights are read from LDS
//     (A operand, row k of the k-major copy);
//   * LayerNorm: 16 values per lane, the 4 lanes of a sample combine with two permlane swaps;
//   * input rows: lane (j, q) loads features k = 4 g + q of its sample, which is the B operand of layer 1 as it is.
// Same arithmetic as mlp_forward_kernel up to the order of the fp32 sums.
#pragma once

#include "mlp_trunk16r.h"

// act + LayerNorm(64) of acc[b][r] (feature 16 b + 4 q + r) -> h = xhat * gamma + beta (the next layer's B operands)
template <bool RELU>
__device__ __forceinline__ void act_ln16(f32x4 (&acc)[4], const float *sG, const float *sBt, int q) {
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc[b][r] = act_fwd<RELU>(acc[b][r]); s += acc[b][r]; }
  const float mean = quad_sum16(s) * (1.f / HID);
  float v = 0.f;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float c = acc[b][r] - mean; v += c * c; }
  const float rstd = 1.0f / sqrtf(quad_sum16(v) * (1.f / HID) + LN_EPS);
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const float4 g = *reinterpret_cast<const float4 *>(sG + 16 * b + 4 * q), t = *reinterpret_cast<const float4 *>(sBt + 16 * b + 4 * q);
    acc[b][0] = (acc[b][0] - mean) * rstd * g.x + t.x;
    acc[b][1] = (acc[b][1] - mean) * rstd * g.y + t.y;
    acc[b][2] = (acc[b][2] - mean) * rstd * g.z + t.z;
    acc[b][3] = (acc[b][3] - mean) * rstd * g.w + t.w;
  }
}

// out[bo] (features 16 bo + .., bo < NB) = bias + W . h, h in registers (feature 16 b + 4 q + r); sW k-major, row stride ws
template <int NB>
__device__ __forceinline__ void layer16(f32x4 (&out)[NB], const f32x4 (&h)[4], const float *sW, int ws, const float *sB, int j, int q) {
#pragma unroll
  for (int bo = 0; bo < NB; ++bo) {
    const float4 bv = *reinterpret_cast<const float4 *>(sB + 16 * bo + 4 * q);
    out[bo][0] = bv.x; out[bo][1] = bv.y; out[bo][2] = bv.z; out[bo][3] = bv.w;
  }
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float *row = sW + (16 * b + 4 * q + r) * ws + j;
#pragma unroll
      for (int bo = 0; bo < NB; ++bo) out[bo] = mfma16(row[16 * bo], h[b][r], out[bo]);
    }
}

// Everything after layer 1's pre-activation h (bias included): act + LayerNorm, hidden layers, head / sampling / trunk output.
template <bool RELU, int LN, int MODE>
__device__ __forceinline__ void forward16_tail(const FwdArgs &p, float *lds, const LdsMap &m, f32x4 (&h)[4], const int64_t i, const bool ok,
                                               const int j, const int q, float *tZ) {
  const int A = p.desc.out_dim;
  act_ln16<RELU>(h, lds + m.ln1_w, lds + m.ln1_b, q);
#pragma unroll
  for (int l = 0; l < LN; ++l) {
    f32x4 h2[4];
    layer16<4>(h2, h, lds + m.w2[l], WP, lds + m.b2[l], j, q);
    act_ln16<RELU>(h2, lds + m.ln2_w[l], lds + m.ln2_b[l], q);
#pragma unroll
    for (int b = 0; b < 4; ++b) h[b] = h2[b];
  }
  // ---- head ----
  if (MODE == 4) {
    // trunk output handed to the other waves of the workgroup through LDS (wide_recurrent_step_dual_kernel): xs[b][lane]
    float4 *xs = reinterpret_cast<float4 *>(tZ);
#pragma unroll
    for (int b = 0; b < 4; ++b) xs[b * 64 + q * 16 + j] = make_float4(h[b][0], h[b][1], h[b][2], h[b][3]);
  } else if (MODE == 2) {
    if (p.out_blocked) {                                            // one contiguous KiB per store instruction (B % 16 == 0)
      float *ob = p.out + ((i >> 4) * 4) * 256 + (q * 16 + j) * 4;
#pragma unroll
      for (int b = 0; b < 4; ++b) *reinterpret_cast<float4 *>(ob + b * 256) = make_float4(h[b][0], h[b][1], h[b][2], h[b][3]);
    } else if (ok) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) p.out[(int64_t)(16 * b + 4 * q + r) * p.B + i] = h[b][r];      // 16 samples x 4 B per segment
    }
  } else if (MODE == 0) {
    if (A == 1) {
      f32x4 z[1];
      layer16<1>(z, h, lds + m.wh, HP, lds + m.bh, j, q);
      if (ok && q == 0) p.out[i] = z[0][0];                                    // out_dim 1: row a = 0 sits in (q = 0, r = 0)
    } else {                                                                   // logits [B][A] (mlp_forward on an actor)
      f32x4 z[2];
      layer16<2>(z, h, lds + m.wh, HP, lds + m.bh, j, q);
      if (ok) {
#pragma unroll
        for (int bo = 0; bo < 2; ++bo)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int a = 16 * bo + 4 * q + r;
            if (a < A) p.out[i * A + a] = z[bo][r];
          }
      }
    }
  } else {
    f32x4 z[2];
    layer16<2>(z, h, lds + m.wh, HP, lds + m.bh, j, q);
#pragma unroll
    for (int bo = 0; bo < 2; ++bo)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = 16 * bo + 4 * q + r;
        if (a < A) tZ[j * TP + a] = z[bo][r];
      }
    wave_lds_sync();
    if (ok && q == 0) {
      const uint64_t ctr = p.counter + (p.counter_dev ? *p.counter_dev : 0ull);
      float action, logp;
      categorical_act_lane(tZ + j * TP, A, p.avail ? p.avail + i * A : nullptr, p.deterministic != 0, p.seed, ctr, (uint64_t)i, action, logp);
      p.actions[i] = action;
      p.logp[i] = logp;
    }
    wave_lds_sync();
  }
}

// ---- the same forward with the weights in REGISTERS -------------------------------------------------------------------------
// A rollout-sized launch is one tile per wave: staging the weights through LDS (global -> registers -> LDS -> barrier -> scalar
// operand reads) is then most of the launch.  Here every wave loads its A operands straight from global memory as 16-byte loads
// (W1, W2.., head: <= 200 registers; the waves of a workgroup fetch the same lines, so L1 / L2 serve all but the first), along
// with the per-feature vectors it needs as 4-wide registers, ALL issued before anything waits: one memory latency, no barrier,
// no LDS except the logits tile of the sampling epilogue.  k-step (b, i) takes input / hidden feature 16 b + 4 q + i in every
// layer.  (Rows of W1 are read up to 15 floats past in_dim — into the next row or the bias that follows W1 in the flat
// parameter vector: finite values that meet zero inputs.)
template <bool RELU, int LN, int MODE>
__device__ __forceinline__ void forward16r_body(const FwdArgs &p, float *lds, const int bid, const int nb) {
  const int n_waves = blockDim.x / WAVE;
  const NetOff &o = p.off;
  const int lane = threadIdx.x & (WAVE - 1), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)), j = lane & 15, q = lane >> 4;
  const int D = p.desc.in_dim, A = p.desc.out_dim;
  const int64_t n_tiles = (p.B + 15) / 16;
  const bool fnorm = p.desc.use_feature_norm != 0;
  float *tZ = lds + wave * 16 * TP;                              // [16][TP] logits of this wave's samples (MODE 1)
  int64_t tile = (int64_t)bid * n_waves + wave;
  if (tile >= n_tiles) return;
  const float *P = p.params;
  // ---- the first tile's rows: requested BEFORE the weights (the counter is in order: what is asked for first arrives first, and
  // the rows are what the first arithmetic — the feature norm — needs; behind the ~80 weight loads and the selects on the head's
  // rows they were a second memory round trip after the weights') ----
  auto load_x = [&](int64_t tl, f32x4 (&xv)[4]) {
    const int64_t i = tl * 16 + j;
    const int64_t row = i < p.B ? (p.rows ? (int64_t)p.rows[i] : i) : 0;
    const int64_t off = p.x_M ? (row / p.x_M) * p.x_sn + (row % p.x_M) * p.x_sm : row * D;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) xv[b][r] = p.x[off + min(16 * b + 4 * q + r, D - 1)];
  };
  f32x4 xn[4];
  load_x(tile, xn);
  // ---- weights and vectors of this lane (issued before the first wait) ----
  Trunk16R<LN> tw;
  trunk16r_load<LN>(tw, P, o, p.desc, j, q);
  constexpr int NBH = MODE == 1 ? 2 : (MODE == 0 ? 1 : 0);       // head blocks of 16 outputs (critic: row 0 only)
  f32x4 wh[NBH > 0 ? NBH : 1][4];
  f32x4 bhv[NBH > 0 ? NBH : 1];
  if constexpr (NBH > 0) {
#pragma unroll
    for (int bo = 0; bo < NBH; ++bo) {
      const int a = 16 * bo + j;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        // rows beyond A: a repeat of the last row (clamped index) — their logits are never stored, and a select here would be
        // an instruction on a value in flight (see trunk16r_load)
        wh[bo][b] = ld4ua(P + o.wh + (size_t)min(a, A - 1) * HID + 16 * b + 4 * q);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) bhv[bo][r] = P[o.bh + min(16 * bo + 4 * q + r, A - 1)];
    }
  }
  for (; tile < n_tiles; tile += (int64_t)nb * n_waves) {
    const int64_t i = tile * 16 + j;
    const bool ok = i < p.B;
    f32x4 x[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) x[b] = xn[b];
    if (tile + (int64_t)nb * n_waves < n_tiles) load_x(tile + (int64_t)nb * n_waves, xn);      // (a wave with a second tile: its rows under this tile)
    f32x4 h[4];
    trunk16r_apply<RELU, LN>(tw, x, h, D, ok, fnorm, q);
    // ---- head ----
    if constexpr (MODE == 2) {
      if (ok) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) p.out[(int64_t)(16 * b + 4 * q + r) * p.B + i] = h[b][r];
      }
    } else {
      f32x4 z[NBH > 0 ? NBH : 1];
#pragma unroll
      for (int bo = 0; bo < NBH; ++bo) {
        z[bo] = bhv[bo];
        if (bo == 0 || A > 16) {
#pragma unroll
          for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) z[bo] = mfma16(wh[bo][b][r], h[b][r], z[bo]);
        }
      }
      if constexpr (MODE == 0) {
        if (A == 1) { if (ok && q == 0) p.out[i] = z[0][0]; }
        else if (ok) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { const int a = 4 * q + r; if (a < A) p.out[i * A + a] = z[0][r]; }     // (out_dim <= 16 through this path)
        }
      } else {
#pragma unroll
        for (int bo = 0; bo < NBH; ++bo)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const int a = 16 * bo + 4 * q + r; if (a < A) tZ[j * TP + a] = z[bo][r]; }
        wave_lds_sync();
        if (ok && q == 0) {
          const uint64_t ctr = p.counter + (p.counter_dev ? *p.counter_dev : 0ull);
          float action, logp;
          categorical_act_lane(tZ + j * TP, A, p.avail ? p.avail + i * A : nullptr, p.deterministic != 0, p.seed, ctr, (uint64_t)i, action, logp);
          p.actions[i] = action;
          p.logp[i] = logp;
        }
        wave_lds_sync();
      }
    }
  }
}

// workgroup `bid` of `nb`; MODE 0: out[i] = value (critic, out_dim 1) | MODE 1: sample / argmax + log-prob (actor)
// | MODE 2: the trunk's output (LayerNorm of the last hidden layer) feature-major, out[64][B] (recurrent networks)
template <bool RELU, int LN, int MODE>
__device__ __forceinline__ void forward16_body(const FwdArgs &p, float *lds, const int bid, const int nb) {
  const int n_waves = blockDim.x / WAVE;
  const NetOff &o = p.off;
  const LdsMap &m = p.map;
  const int lane = threadIdx.x & (WAVE - 1), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)), j = lane & 15, q = lane >> 4;
  const int D = p.desc.in_dim, Dp = (D + 1) & ~1, A = p.desc.out_dim;
  const int NG = (D + 3) >> 2;                                   // k-groups of the input layer
  const int64_t n_tiles = (p.B + 15) / 16;
  const bool fnorm = p.desc.use_feature_norm != 0;
  const float inv_D = 1.0f / (float)D;
  float *tZ = lds + m.tiles + wave * m.wave_stride;              // [16][TP] logits of this wave's samples (MODE 1)
  bool staged = false;
  for (int64_t tile = (int64_t)bid * n_waves + wave; tile < n_tiles || !staged; tile += (int64_t)nb * n_waves) {
    const int64_t i = tile * 16 + j;
    const bool ok = tile < n_tiles && i < p.B;
    const int64_t row = ok ? (p.rows ? (int64_t)p.rows[i] : i) : 0;
    const int64_t off = p.x_M ? (row / p.x_M) * p.x_sn + (row % p.x_M) * p.x_sm : row * D;
    // ---- input rows: features 4 g + q of sample j (unconditional clamped loads, all in flight under the weight staging) ----
    float xin[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) xin[g] = p.x[off + min(4 * g + q, D - 1)];
    if (!staged) {                                               // first pass: stage the weights under the row loads
      stage_all_weights_1shot<LN>(lds, m, p.params, o, p.desc);
      __syncthreads();
      staged = true;
      if (tile >= n_tiles) break;                                // a wave without a tile only helped staging
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) xin[g] = (ok && 4 * g + q < D) ? xin[g] : 0.f;
    if (fnorm) {
      // slots with 4 g + q >= D hold 0: they add (0 - mean)^2 to the centred sum (taken out again below), and their
      // gamma / beta entries are staged as 0, so no lane predicate is needed after the zeroing above
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) s += xin[g];
      const float mean = quad_sum16(s) * inv_D;
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) { const float c = xin[g] - mean; v += c * c; }
      const int n_empty = 16 - max(0, min(16, (D - q + 3) >> 2));            // slots of this lane beyond D
      v -= (float)n_empty * mean * mean;
      const float rstd = 1.0f / sqrtf(fmaxf(quad_sum16(v), 0.f) * inv_D + LN_EPS);
#pragma unroll
      for (int g = 0; g < 16; ++g) xin[g] = (xin[g] - mean) * rstd * lds[m.fn_w + 4 * g + q] + lds[m.fn_b + 4 * g + q];
    }
    // ---- layer 1: k-group g = input features 4 g + q; groups come in chunks of four (one wave-uniform branch per chunk;
    // a chunk's surplus groups multiply zeros: xin = 0 beyond D, weight row clamped) ----
    f32x4 h[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float4 bv = *reinterpret_cast<const float4 *>(lds + m.b1 + 16 * b + 4 * q);
      h[b][0] = bv.x; h[b][1] = bv.y; h[b][2] = bv.z; h[b][3] = bv.w;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (4 * c < NG) {
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
          const int g = 4 * c + gg;
          const float *rowp = lds + m.w1 + min(4 * g + q, Dp - 1) * WP + j;
#pragma unroll
          for (int b = 0; b < 4; ++b) h[b] = mfma16(rowp[16 * b], xin[g], h[b]);
        }
      }
    }
    forward16_tail<RELU, LN, MODE>(p, lds, m, h, i, ok, j, q, tZ);
  }
}
