// mlp_trunk16r.h — the MLP trunk (feature norm, fc1, fc_h.., each with activation + LayerNorm; mlp.py:18-55) of ONE 16-sample tile
// on v_mfma_f32_16x16x4_f32 with the weights in REGISTERS: shared by the rollout forward kernels (mlp_fwd16.h) and the fused
// recurrent rollout step (gru.hip).  Lane (j = lane & 15, q = lane >> 4) holds features 16 b + 4 q + r of sample j in every
// layer; k-step (b, r) takes input / hidden feature 16 b + 4 q + r.  A operands are 16-byte loads straight from the flat
// parameter vector (rows of W1 are read up to 15 floats past in_dim — into the next row or the bias that follows W1: finite
// values that meet zero inputs).
#pragma once

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ float quad_sum16(float v) { return xhalf_sum(xrow_sum(v)); }     // lanes j, j+16, j+32, j+48

typedef float f32x4_ua __attribute__((ext_vector_type(4), aligned(4)));

// Elements k .. k + 3 of a row of D floats (D >= 4) as one 16-byte load that never leaves the row: the load starts at
// min(k, D - 4) (ld4_row_raw) and the lanes are shifted down by the difference, zeros beyond the row (ld4_row_fix — separate,
// so that the load can stay in flight).  Rows need 4-byte alignment only (global_load_dwordx4 takes any dword address); with
// D % 4 == 0 and k % 4 == 0 the shift is 0 or >= 4 (all zero).
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ f32x4 ld4_row_raw(const float *row, int k, int D) {
  const f32x4_u l = *reinterpret_cast<const f32x4_u *>(row + min(k, D - 4));
  f32x4 r; r[0] = l[0]; r[1] = l[1]; r[2] = l[2]; r[3] = l[3];
  return r;
}
__device__ __forceinline__ f32x4 ld4_row_fix(const f32x4 l, int k, int D, bool al4) {
  // bit masks, not selects: nested selects on a per-lane shift come out as divergent branches
  const int s = k - min(k, D - 4);
  const uint32_t l0 = __float_as_uint(l[0]), l1 = __float_as_uint(l[1]), l2 = __float_as_uint(l[2]), l3 = __float_as_uint(l[3]);
  const uint32_t m0 = s == 0 ? ~0u : 0u;
  f32x4 r;
  if (al4) {
    r[0] = __uint_as_float(l0 & m0); r[1] = __uint_as_float(l1 & m0); r[2] = __uint_as_float(l2 & m0); r[3] = __uint_as_float(l3 & m0);
  } else {
    const uint32_t m1 = s == 1 ? ~0u : 0u, m2 = s == 2 ? ~0u : 0u, m3 = s == 3 ? ~0u : 0u;
    r[0] = __uint_as_float((l0 & m0) | (l1 & m1) | (l2 & m2) | (l3 & m3));
    r[1] = __uint_as_float((l1 & m0) | (l2 & m1) | (l3 & m2));
    r[2] = __uint_as_float((l2 & m0) | (l3 & m1));
    r[3] = __uint_as_float(l3 & m0);
  }
  return r;
}
__device__ __forceinline__ f32x4 ld4_row(const float *row, int k, int D, bool al4) { return ld4_row_fix(ld4_row_raw(row, k, D), k, D, al4); }

__device__ __forceinline__ f32x4 ld4ua(const float *p) { const f32x4_ua v = *reinterpret_cast<const f32x4_ua *>(p); f32x4 r; r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; r[3] = v[3]; return r; }

// act + LayerNorm(64) with the affine in registers (g, t: features 16 b + 4 q + r of gamma / beta)
template <bool RELU>
__device__ __forceinline__ void act_ln16r(f32x4 (&acc)[4], const f32x4 (&g)[4], const f32x4 (&t)[4]) {
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
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[b][r] = (acc[b][r] - mean) * rstd * g[b][r] + t[b][r];
}


template <int LN>
struct Trunk16R {
  f32x4 w1[4][4];                                                // [bo][b]: W1[16 bo + j][16 b + 4 q ..]
  f32x4 g0[4], t0[4];                                            // feature-norm affine of the inputs (zero beyond in_dim)
  f32x4 b1v[4], g1[4], t1[4];
  f32x4 w2[LN > 0 ? LN : 1][4][4], b2v[LN > 0 ? LN : 1][4], g2[LN > 0 ? LN : 1][4], t2[LN > 0 ? LN : 1][4];
};

// all loads of the trunk's weights and vectors for lane (j, q) (nothing waits here)
template <int LN>
__device__ __forceinline__ void trunk16r_load(Trunk16R<LN> &w, const float *P, const NetOff &o, const mappo_net_desc &d, int j, int q) {
  const int D = d.in_dim, NB1 = (D + 15) >> 4;
  const bool fnorm = d.use_feature_norm != 0;
#pragma unroll
  for (int b = 0; b < 4; ++b)
    if (b < NB1) {
#pragma unroll
      for (int bo = 0; bo < 4; ++bo) w.w1[bo][b] = ld4ua(P + o.w1 + (size_t)(16 * bo + j) * D + 16 * b + 4 * q);
    }
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 16 * b + 4 * q + r;
      const int ofw = fnorm ? o.fn_w : o.w1, ofb = fnorm ? o.fn_b : o.w1;      // (no feature norm: fn_w / fn_b are -1 — read something valid)
      // raw (clamped index): slots beyond in_dim are zeroed where the affine is applied (trunk16r_apply) — a select HERE is an
      // instruction on a value in flight: the wait for it, and for every load issued before it, lands in front of the tile loop
      w.g0[b][r] = P[ofw + min(k, D - 1)];
      w.t0[b][r] = P[ofb + min(k, D - 1)];
    }
#pragma unroll
  for (int b = 0; b < 4; ++b) { w.b1v[b] = ld4ua(P + o.b1 + 16 * b + 4 * q); w.g1[b] = ld4ua(P + o.ln1_w + 16 * b + 4 * q); w.t1[b] = ld4ua(P + o.ln1_b + 16 * b + 4 * q); }
#pragma unroll
  for (int l = 0; l < LN; ++l)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
      for (int bo = 0; bo < 4; ++bo) w.w2[l][bo][b] = ld4ua(P + o.w2[l] + (size_t)(16 * bo + j) * HID + 16 * b + 4 * q);
      w.b2v[l][b] = ld4ua(P + o.b2[l] + 16 * b + 4 * q); w.g2[l][b] = ld4ua(P + o.ln2_w[l] + 16 * b + 4 * q); w.t2[l][b] = ld4ua(P + o.ln2_b[l] + 16 * b + 4 * q);
    }
}

// x: the lane's raw inputs (feature 16 b + 4 q + r of its sample; any value beyond in_dim / for a missing sample) -> h: the
// trunk's output (LayerNorm of the last hidden layer, affine included)
template <bool RELU, int LN>
__device__ __forceinline__ void trunk16r_apply(const Trunk16R<LN> &w, f32x4 (&x)[4], f32x4 (&h)[4], int D, bool ok, bool fnorm, int q) {
  const int NB1 = (D + 15) >> 4;
  const float inv_D = 1.0f / (float)D;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) x[b][r] = (ok && 16 * b + 4 * q + r < D) ? x[b][r] : 0.f;
  if (fnorm) {
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) s += (x[b][0] + x[b][1]) + (x[b][2] + x[b][3]);
    const float mean = quad_sum16(s) * inv_D;
    float v = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float c = (16 * b + 4 * q + r < D) ? x[b][r] - mean : 0.f; x[b][r] = c; v += c * c; }
    const float rstd = 1.0f / sqrtf(quad_sum16(v) * inv_D + LN_EPS);
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) x[b][r] = (16 * b + 4 * q + r < D) ? x[b][r] * rstd * w.g0[b][r] + w.t0[b][r] : 0.f;
  }
#pragma unroll
  for (int bo = 0; bo < 4; ++bo) h[bo] = w.b1v[bo];
#pragma unroll
  for (int b = 0; b < 4; ++b)
    if (b < NB1) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int bo = 0; bo < 4; ++bo) h[bo] = mfma16(w.w1[bo][b][r], x[b][r], h[bo]);
    }
  act_ln16r<RELU>(h, w.g1, w.t1);
#pragma unroll
  for (int l = 0; l < LN; ++l) {
    f32x4 h2[4];
#pragma unroll
    for (int bo = 0; bo < 4; ++bo) h2[bo] = w.b2v[l][bo];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int bo = 0; bo < 4; ++bo) h2[bo] = mfma16(w.w2[l][bo][b][r], h[b][r], h2[bo]);
    act_ln16r<RELU>(h2, w.g2[l], w.t2[l]);
#pragma unroll
    for (int b = 0; b < 4; ++b) h[b] = h2[b];
  }
}
