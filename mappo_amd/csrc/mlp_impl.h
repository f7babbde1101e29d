// mlp.hip — K7 / K6 / K8 (+K5 fused): the shared actor/critic MLP (onpolicy/algorithms/utils/mlp.py:6-55) with
// its head (distributions.py:55-68 logits, r_actor_critic.py:136-165 v_out) on the fp32 matrix cores.
//
//   trunk:  x -> LN_D -> Linear(D,64) -> act -> LN_64 -> [Linear(64,64) -> act -> LN_64] x layer_N -> head
//
// Formulation (MI355X-first, not a GEMM-library call chain): everything is computed TRANSPOSED,
// Y^T[f][s] = W[f][:] . X^T[:][s], with v_mfma_f32_32x32x2_f32.  A wavefront owns a tile of 32 samples; in the
// MFMA accumulator layout a lane then holds ONE sample (column) and 32 of its 64 features (rows, the other
// 32 sit in lane^32), so bias, activation and LayerNorm are per-lane register loops plus one cross-half
// exchange — no LDS transposes, no atomics.  Weights sit in LDS for the lifetime of the workgroup (k-major,
// row stride 65/33 so that both the forward A-operand read W[f][k] (lanes <-> f) and the backward read
// W^T (lanes <-> k) are bank-conflict free); each wave keeps its activations in private LDS tiles
// [feature][sample] (row stride 33), which serve as B operand of the next layer (lanes <-> sample) and,
// read transposed (lanes <-> feature, k <-> sample), as both operands of the weight-gradient products.
// A tile stores the NORMALISED value xhat = (a - mean) * rstd of its LayerNorm; the affine (gamma, beta) is
// applied when the tile is read as an operand (two broadcast LDS reads + one FMA per MFMA pair), so the
// backward pass finds xhat in the tile and only mean/rstd/sign-mask (3 registers) survive from the forward.
//
// Kernels
//   mlp_forward_kernel  MODE 0: out = head(trunk(x[rows]))           (evaluate_actions / get_values)
//                       MODE 1: + availability mask, sample|argmax, log-prob  (get_actions / act)
//   mlp_update_kernel   forward, head gradient, backward and weight-gradient accumulation in ONE launch:
//                       HEAD 0: head gradient supplied by the caller (mappo_mlp_backward)
//                       HEAD 1: actor  — PPO clipped surrogate + entropy computed in the kernel from the logits
//                       HEAD 2: critic — clipped Huber|MSE value loss computed in the kernel from the values
//                       (r_mappo.py:52-89,124-141: logits / values / their gradients never touch HBM).
//   The forward is recomputed per tile instead of reading saved activations back from HBM, dW accumulators
//   stay in registers across the persistent tile loop, the next tile's rows are prefetched into registers
//   under the current tile's MFMA chains, and every workgroup writes ONE partial-gradient slab, summed by
//   mappo_slab_reduce (deterministic, no float atomics).
//
// Wide observations (in_dim > 64, up to 512: BASELINE configs 4-5) use the XW == 2 instantiations: layer 1 is
// K-chunked (the workgroup streams 64-column chunks of W1 through LDS, each wave re-normalises its rows chunk by
// chunk), and its weight / feature-norm gradients come from wide_l1_bwd_kernel, which owns one K-chunk per
// workgroup and sweeps the row tiles (dz1 goes through a feature-major HBM scratch).
// Limits of this build: hidden == 64, out_dim <= 32, layer_N <= 2, in_dim <= 512.
#include "mlp_core.h"
#include <stdlib.h>

#ifdef MLP_TU_MAIN
extern "C" int64_t mappo_net_param_count(const mappo_net_desc *desc) {
  if (!desc) return -1;
  return net_offsets(*desc).total;
}
#endif

// ------------------------------------------------------------------------------------------------
// LDS carve-up (floats); every region starts on a 16-byte boundary.  Dp = in_dim rounded up to even.
// Per wave: tX [Dp rows] | tH [(layer_N+1) x 64 rows] | tZ [32 rows: head output / head gradient as [s][a]].
// ------------------------------------------------------------------------------------------------
struct LdsMap {
  int w1, w2[MAPPO_MAX_LAYER_N], wh;
  int fn_w, fn_b, b1, ln1_w, ln1_b, b2[MAPPO_MAX_LAYER_N], ln2_w[MAPPO_MAX_LAYER_N], ln2_b[MAPPO_MAX_LAYER_N], bh;
  int scratch;       // n_waves x 192 floats: exchange buffers / epilogue scratch of the update kernels
  int tiles, x_rows, wave_stride, total;
  int fn_size;       // floats reserved per feature-norm vector (64, or in_dim rounded up to 64 for wide inputs)
};


__host__ __device__ inline LdsMap lds_map(const mappo_net_desc &d, int n_waves) {
  LdsMap m;
  int p = 0;
  const bool xw = d.in_dim > MAXD;
  const int Dp = xw ? MAXD : ((d.in_dim + 1) & ~1);          // wide inputs: one 64-column chunk of W1 / of the rows at a time
  m.fn_size = xw ? ((d.in_dim + 63) / 64) * 64 : MAXD;
  m.w1 = p; p = al4(p + Dp * WP);
  for (int l = 0; l < MAPPO_MAX_LAYER_N; ++l) { m.w2[l] = p; if (l < d.layer_N) p = al4(p + HID * WP); }
  m.wh = p; p = al4(p + HID * HP);
  m.fn_w = p; p += m.fn_size; m.fn_b = p; p += m.fn_size;
  m.b1 = p; p += HID; m.ln1_w = p; p += HID; m.ln1_b = p; p += HID;
  for (int l = 0; l < MAPPO_MAX_LAYER_N; ++l) {
    m.b2[l] = p; m.ln2_w[l] = p; m.ln2_b[l] = p;
    if (l < d.layer_N) { m.b2[l] = p; p += HID; m.ln2_w[l] = p; p += HID; m.ln2_b[l] = p; p += HID; }
  }
  m.bh = p; p += 32;
  m.scratch = p; p += n_waves * 192;           // update kernels: exchange buffers + epilogue scratch
  m.tiles = p;
  m.x_rows = Dp;
  m.wave_stride = al4((Dp + (d.layer_N + 1) * HID + TS) * TP);
  p += n_waves * m.wave_stride;
  m.total = p;
  return m;
}

// The part of the map the register-resident tail (mlp_fwd16.h: forward16_tail after stage_tail_1shot) reads — no W1 chunk area, no
// feature-norm vectors, no per-wave tiles: what a kernel that keeps W1' elsewhere in LDS puts behind it (wide_features16_resident_kernel).
__host__ __device__ inline LdsMap lds_map_tail(const mappo_net_desc &d) {
  LdsMap m = lds_map(d, 1);
  int p = 0;
  m.w1 = 0; m.fn_w = 0; m.fn_b = 0; m.fn_size = 0;
  for (int l = 0; l < MAPPO_MAX_LAYER_N; ++l) { m.w2[l] = p; if (l < d.layer_N) p = al4(p + HID * WP); }
  m.wh = p; p = al4(p + HID * HP);
  m.b1 = p; p += HID; m.ln1_w = p; p += HID; m.ln1_b = p; p += HID;
  for (int l = 0; l < MAPPO_MAX_LAYER_N; ++l) {
    m.b2[l] = p; m.ln2_w[l] = p; m.ln2_b[l] = p;
    if (l < d.layer_N) { m.b2[l] = p; p += HID; m.ln2_w[l] = p; p += HID; m.ln2_b[l] = p; p += HID; }
  }
  m.bh = p; p += 32;
  m.scratch = p;
  m.tiles = p;
  m.wave_stride = 0;
  m.total = p;
  return m;
}

// Workgroup-cooperative staging of one weight matrix: global W[f][k] (row-major, K columns) -> LDS dst[k*stride + f],
// rows k in [K, Kpad) zeroed.  Loads are UNCONDITIONAL (clamped index) and all issued before the first LDS write, so a
// thread pays one memory latency for its whole share (a predicated load is waited for individually by hipcc).
__device__ __forceinline__ void stage_weight_T(float *dst, const float *__restrict__ src, int F, int K, int Kpad, int stride) {
  const int total = F * K;
  const int nthr = blockDim.x, tid = threadIdx.x;
  if ((((uintptr_t)src) & 15) == 0 && (total & 3) == 0) {
    const int n4 = total >> 2;
    for (int i0 = 0; i0 < n4; i0 += 8 * nthr) {
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = reinterpret_cast<const float4 *>(src)[min(i0 + j * nthr + tid, n4 - 1)];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = i0 + j * nthr + tid;
        if (i < n4) {
          const int e = i << 2;
          int f = e / K, k = e - f * K;
          const float vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            dst[k * stride + f] = vv[c];
            if (++k == K) { k = 0; ++f; }
          }
        }
      }
    }
  } else {
    for (int e0 = 0; e0 < total; e0 += 8 * nthr) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[min(e0 + j * nthr + tid, total - 1)];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = e0 + j * nthr + tid;
        if (e < total) { const int f = e / K; dst[(e - f * K) * stride + f] = v[j]; }
      }
    }
  }
  for (int e = tid; e < F * (Kpad - K); e += nthr) {
    const int f = e % F, k = K + e / F;
    dst[k * stride + f] = 0.f;
  }
}

// All per-feature vectors of the network in ONE pass: element e of the concatenated LDS vector area
// [fn_w 64 | fn_b 64 | b1 ln1_w ln1_b | (b2 ln2_w ln2_b) x LN | bh 32] maps to a global offset or a fill value.
template <int LN>
__device__ __forceinline__ void stage_vectors(float *lds, const LdsMap &m, const float *__restrict__ params, const NetOff &o,
                                              const mappo_net_desc &d) {
  const int D = d.in_dim, A = d.out_dim;
  const int FN = m.fn_size;
  const int n_total = 2 * FN + 3 * HID * (1 + LN) + 32;
  const int nthr = blockDim.x, tid = threadIdx.x;
  for (int e0 = 0; e0 < n_total; e0 += 4 * nthr) {
    float v[4]; int dsti[4]; bool wr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = e0 + j * nthr + tid;
      int src = -1; float fill = 0.f; int dst = m.fn_w;
      wr[j] = e < n_total;
      if (e < FN) { dst = m.fn_w + e; if (d.use_feature_norm) { if (e < D) src = o.fn_w + e; } else fill = e < D ? 1.f : 0.f; }
      else if (e < 2 * FN) { const int i = e - FN; dst = m.fn_b + i; if (d.use_feature_norm && i < D) src = o.fn_b + i; }
      else if (e < 2 * FN + 3 * HID) { const int i = e - 2 * FN; dst = m.b1 + i; src = o.b1 + i; }
      else if (e < 2 * FN + 3 * HID * (1 + LN)) {
        const int i = e - 2 * FN - 3 * HID, l = i / (3 * HID), r = i - l * 3 * HID;
        dst = (l == 0 ? m.b2[0] : m.b2[LN > 1 ? 1 : 0]) + r;
        src = (l == 0 ? o.b2[0] : o.b2[LN > 1 ? 1 : 0]) + r;
      } else { const int i = e - 2 * FN - 3 * HID * (1 + LN); dst = m.bh + i; if (i < A) src = o.bh + i; }
      const float ld = params[src >= 0 ? src : 0];        // unconditional load, selected below
      v[j] = src >= 0 ? ld : fill;
      dsti[j] = dst;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) if (wr[j]) lds[dsti[j]] = v[j];
  }
}

template <int LN>
__device__ __forceinline__ void stage_all_weights(float *lds, const LdsMap &m, const float *__restrict__ params,
                                                  const NetOff &o, const mappo_net_desc &d) {
  const int D = d.in_dim, Dp = (D + 1) & ~1, A = d.out_dim;
  stage_vectors<LN>(lds, m, params, o, d);
  if (D <= MAXD) stage_weight_T(lds + m.w1, params + o.w1, HID, D, Dp, WP);      // wide inputs stream W1 chunk by chunk
#pragma unroll
  for (int l = 0; l < LN; ++l) stage_weight_T(lds + m.w2[l], params + o.w2[l], HID, HID, HID, WP);
  // head: dst[k*HP + a] = Wh[a][k]; columns a >= A are zero
  {
    const int nthr = blockDim.x, tid = threadIdx.x, total = HID * 32, real = A * HID;
    for (int e0 = 0; e0 < total; e0 += 8 * nthr) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = params[o.wh + min(e0 + j * nthr + tid, real - 1)];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = e0 + j * nthr + tid;
        if (e < total) { const int a = e >> 6, k = e & 63; lds[m.wh + k * HP + a] = (e < real) ? v[j] : 0.f; }
      }
    }
  }
}

// One-shot variant for in_dim <= 64 and >= 256 threads: EVERY global load of the staging (vectors, W1, W2.., head) is
// issued before the first LDS store, so a thread pays one memory latency for the whole network instead of one per
// matrix (four to five dependent L2 / Infinity-Cache round trips otherwise — a large part of a rollout-sized launch).
// Falls back to stage_all_weights when the float4 views are not available (odd in_dim, unaligned params, small block).
template <int LN>
__device__ __forceinline__ void stage_all_weights_1shot(float *lds, const LdsMap &m, const float *__restrict__ params,
                                                        const NetOff &o, const mappo_net_desc &d) {
  const int D = d.in_dim, Dp = (D + 1) & ~1, A = d.out_dim;
  const int nthr = blockDim.x, tid = threadIdx.x;
  const int FN = m.fn_size;
  const int n_vec = 2 * FN + 3 * HID * (1 + LN) + 32;
  if (nthr < 256 || (D & 1) || D > MAXD || ((((uintptr_t)params) & 15) != 0) || n_vec > 4 * nthr) {
    stage_all_weights<LN>(lds, m, params, o, d);
    return;
  }
  // ---- loads ----
  const float4 *g1 = reinterpret_cast<const float4 *>(params + o.w1), *gh = reinterpret_cast<const float4 *>(params + o.wh);
  const int n4_1 = 16 * D, n4_h = 16 * A;
  float4 w1v[4], w2v[LN > 0 ? LN : 1][4], whv[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) w1v[j] = g1[min(j * nthr + tid, n4_1 - 1)];
#pragma unroll
  for (int l = 0; l < LN; ++l) {
    const float4 *g2 = reinterpret_cast<const float4 *>(params + o.w2[l]);
#pragma unroll
    for (int j = 0; j < 4; ++j) w2v[l][j] = g2[min(j * nthr + tid, 1023)];
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) whv[j] = gh[min(j * nthr + tid, n4_h - 1)];
  float vv[4]; int vdst[4]; bool vwr[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = j * nthr + tid;
    int src = -1; float fill = 0.f; int dst = m.fn_w;
    vwr[j] = e < n_vec;
    if (e < FN) { dst = m.fn_w + e; if (d.use_feature_norm) { if (e < D) src = o.fn_w + e; } else fill = e < D ? 1.f : 0.f; }
    else if (e < 2 * FN) { const int i = e - FN; dst = m.fn_b + i; if (d.use_feature_norm && i < D) src = o.fn_b + i; }
    else if (e < 2 * FN + 3 * HID) { const int i = e - 2 * FN; dst = m.b1 + i; src = o.b1 + i; }
    else if (e < 2 * FN + 3 * HID * (1 + LN)) {
      const int i = e - 2 * FN - 3 * HID, l = i / (3 * HID), r = i - l * 3 * HID;
      dst = (l == 0 ? m.b2[0] : m.b2[LN > 1 ? 1 : 0]) + r;
      src = (l == 0 ? o.b2[0] : o.b2[LN > 1 ? 1 : 0]) + r;
    } else { const int i = e - 2 * FN - 3 * HID * (1 + LN); dst = m.bh + i; if (i < A) src = o.bh + i; }
    const float ld = params[src >= 0 ? src : 0];
    vv[j] = src >= 0 ? ld : fill;
    vdst[j] = dst;
  }
  // ---- stores ----
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = j * nthr + tid;
    if (i < n4_1) {
      const int e = i << 2;
      int f = e / D, k = e - f * D;
      const float t[4] = {w1v[j].x, w1v[j].y, w1v[j].z, w1v[j].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        lds[m.w1 + k * WP + f] = t[c];
        if (++k == D) { k = 0; ++f; }
      }
    }
  }
  for (int e = tid; e < HID * (Dp - D); e += nthr) lds[m.w1 + (D + e / HID) * WP + (e % HID)] = 0.f;
#pragma unroll
  for (int l = 0; l < LN; ++l)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = j * nthr + tid;
      if (i < 1024) {
        const int e = i << 2, f = e >> 6, k = e & 63;
        float *q = lds + m.w2[l] + k * WP + f;
        q[0] = w2v[l][j].x; q[WP] = w2v[l][j].y; q[2 * WP] = w2v[l][j].z; q[3 * WP] = w2v[l][j].w;
      }
    }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = j * nthr + tid;
    if (i < n4_h) {
      const int e = i << 2, a = e >> 6, k = e & 63;
      float *q = lds + m.wh + k * HP + a;
      q[0] = whv[j].x; q[HP] = whv[j].y; q[2 * HP] = whv[j].z; q[3 * HP] = whv[j].w;
    }
  }
  for (int e = tid; e < HID * (32 - A); e += nthr) { const int k = e / (32 - A), a = A + e - k * (32 - A); lds[m.wh + k * HP + a] = 0.f; }
#pragma unroll
  for (int j = 0; j < 4; ++j) if (vwr[j]) lds[vdst[j]] = vv[j];
}

// ------------------------------------------------------------------------------------------------
// input rows.  Lane (s = lane & 31, half = lane >> 5) fetches features k = 2j + half of sample s of the NEXT tile
// into registers (NV = ceil(D/2) <= 16 | 32); at the top of the tile the LayerNorm over the D input features
// (mlp.py:45,51-52) is a per-lane register loop plus one cross-half exchange, and xhat0 goes to tX[k][s] in the
// layout the B operand of layer 1 reads.  (A wave load touches 32 rows, one word each; the rows of a tile are
// re-touched by the next j while still in L1.)
// ------------------------------------------------------------------------------------------------
template <bool WIDE>
struct RowPrefetch {
  float v[WIDE ? TS : TS / 2];
  int my_row;         // source row of sample lane&31 of the prefetched tile (both halves hold it)
  int n_valid;
  bool flat;          // v holds the float4 chunks 4*(lane + 64 j) of the tile's contiguous [32][D] block (see below)
};

// Two fetch modes.  Gather (any row list, partial tiles, odd D): lane (s, half) loads x[row_s][2j + half] — every wave
// load touches 32 rows, one word each.  Flat (rows == nullptr, full tile, even D, 16-B aligned x): the tile is one
// contiguous block of 32*D floats, fetched as fully coalesced float4s (8x fewer cache lines touched per instruction);
// commit_rows then routes it through an LDS staging area to reach the lane <-> sample layout.
template <bool WIDE>
__device__ __forceinline__ void prefetch_rows(RowPrefetch<WIDE> &pf, const float *__restrict__ x,
                                              const int32_t *__restrict__ rows, int64_t base, int64_t B, int D, int lane,
                                              int64_t x_sn = 0, int64_t x_sm = 0, int x_M = 0) {
  const int s = lane & 31, half = lane >> 5;
  pf.n_valid = (int)max((int64_t)0, min((int64_t)TS, B - base));
  pf.my_row = 0;
  const bool ok = s < pf.n_valid;
  if (ok) pf.my_row = rows ? rows[base + s] : (int)(base + s);
  constexpr int NV = WIDE ? TS : TS / 2;
  pf.flat = rows == nullptr && x_M == 0 && pf.n_valid == TS && (D & 1) == 0 && (((uintptr_t)x) & 15) == 0;
  if (pf.flat) {
    const float4 *src4 = reinterpret_cast<const float4 *>(x + base * D);
    const int n4 = TS * D / 4;
#pragma unroll
    for (int j = 0; j < NV / 4; ++j) {
      const float4 q = src4[min(lane + 64 * j, n4 - 1)];
      pf.v[4 * j + 0] = q.x; pf.v[4 * j + 1] = q.y; pf.v[4 * j + 2] = q.z; pf.v[4 * j + 3] = q.w;
    }
    return;
  }
  const int64_t row_off = x_M ? (int64_t)(pf.my_row / x_M) * x_sn + (int64_t)(pf.my_row % x_M) * x_sm : (int64_t)pf.my_row * D;
  const float *src = x + row_off + half;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    pf.v[j] = 0.f;
#ifndef EXP_NO_PREFETCH
    if (ok && 2 * j + half < D) pf.v[j] = src[2 * j];
#endif
  }
}

// tX[k][s] <- xhat0 (feature LayerNorm, affine applied on read) or the raw input when feature norm is off, for ALL
// k < 2*NV: rows k >= D receive a finite filler (they spill into the activation tiles, dead at this point); nothing
// reads them with a non-zero weight (W1's padding rows are zero, gradient columns k >= D are dropped).  Writing them
// unconditionally keeps 32 loop-invariant lane predicates out of the tile loop (hipcc hoists each into an SGPR pair
// and then spills them).
// tF: >= 32*(D+2) floats of wave-private LDS that are dead at this point (the activation tiles), staging of the flat mode.
// magic = 2^32 / D + 1 (flat mode: e / D == umulhi(e, magic) for the small e used here).
template <bool WIDE>
__device__ __forceinline__ void commit_rows(float *tX, float *tF, const RowPrefetch<WIDE> &pf, int D, uint32_t magic, int lane,
                                            bool feature_norm) {
  const int s = lane & 31, half = lane >> 5;
  constexpr int NV = WIDE ? TS : TS / 2;
  float v[NV];
  if (pf.flat) {
    // row stride D when D = 2 (mod 4) (lanes (s, half) then read 64 distinct banks), D + 2 when D = 0 (mod 4)
    const bool pad = (D & 3) == 0;
    const int stride = pad ? D + 2 : D;
#pragma unroll
    for (int j = 0; j < NV / 4; ++j) {
      const int e4 = 4 * (lane + 64 * j);
      if (e4 < TS * D) {
        if (!pad) {
          *reinterpret_cast<float4 *>(tF + e4) = make_float4(pf.v[4 * j], pf.v[4 * j + 1], pf.v[4 * j + 2], pf.v[4 * j + 3]);
        } else {
          const int r = (int)__umulhi((uint32_t)e4, magic), k = e4 - r * D;
          float2 *q = reinterpret_cast<float2 *>(tF + r * stride + k);
          q[0] = make_float2(pf.v[4 * j], pf.v[4 * j + 1]);
          q[1] = make_float2(pf.v[4 * j + 2], pf.v[4 * j + 3]);
        }
      }
    }
    wave_lds_sync();
    const float *rowp = tF + s * stride + half;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      v[j] = 0.f;
      if (2 * j < D) v[j] = rowp[2 * j];           // D is even here: the bound is wave-uniform (scalar branch, no lane mask)
    }
  } else {
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = pf.v[j];   // slots beyond D hold 0
  }
  float mean = 0.f, rstd = 1.f;
  if (feature_norm) {
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) sum += v[j];
    mean = xhalf_sum(sum) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) { const float c = v[j] - mean; q += c * c; }
    // the NV - ceil((D - half)/2) empty slots of this lane each added (0 - mean)^2: take them out again
    const int n_empty = NV - ((D - half + 1) >> 1);
    q -= (float)n_empty * mean * mean;
    rstd = 1.0f / sqrtf(fmaxf(xhalf_sum(q), 0.f) / (float)D + LN_EPS);
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) tX[(2 * j + half) * TP + s] = (v[j] - mean) * rstd;
}

// ---- wide inputs (in_dim > 64) ----
// 64 columns [c0, c0+kc) of W1[64][D] -> sW[kk*WP + f]; rows kk in [kc, 64) zeroed.  Workgroup-cooperative, batched loads.
__device__ __forceinline__ void stage_w1_chunk(float *dst, const float *__restrict__ w1, int D, int c0, int kc) {
  const int nthr = blockDim.x, tid = threadIdx.x, total = HID * kc;
  for (int e0 = 0; e0 < total; e0 += 8 * nthr) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = min(e0 + j * nthr + tid, total - 1);
      const int f = e / kc, kk = e - f * kc;
      v[j] = w1[f * D + c0 + kk];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = e0 + j * nthr + tid;
      if (e < total) { const int f = e / kc; dst[(e - f * kc) * WP + f] = v[j]; }
    }
  }
  for (int e = tid; e < HID * (MAXD - kc); e += nthr) dst[(kc + e / HID) * WP + (e % HID)] = 0.f;
}

// LayerNorm statistics of a full input row (two passes over the row, which stays in L1/L2 between them)
__device__ __forceinline__ void wide_row_stats(const float *__restrict__ xr, int D, bool ok, int half, bool feature_norm, float &mean,
                                               float &rstd) {
  mean = 0.f; rstd = 1.f;
  if (!feature_norm) return;
  // unconditional loads (a lane without a row reads row 0: finite values nobody uses) and eight of them in flight per trip:
  // as a predicated one-load-per-trip loop the two passes cost ~160 memory round trips each
  float s0 = 0.f;
#pragma unroll 8
  for (int k = half; k < D; k += 2) s0 += xr[k];
  mean = xhalf_sum(s0) / (float)D;
  float q = 0.f;
#pragma unroll 8
  for (int k = half; k < D; k += 2) { const float c = xr[k] - mean; q += c * c; }
  rstd = 1.0f / sqrtf(xhalf_sum(q) / (float)D + LN_EPS);
}

// tX[kk][s] <- xhat0 of columns [c0, c0+64) of this lane's row
__device__ __forceinline__ void wide_commit_chunk(float *tX, const float *__restrict__ xr, int D, int c0, bool ok, float mean, float rstd,
                                                  int l31, int half) {
  float v[TS];
#pragma unroll
  for (int j = 0; j < TS; ++j) v[j] = xr[min(c0 + 2 * j + half, D - 1)];          // unconditional, clamped: all 32 in flight
#pragma unroll
  for (int j = 0; j < TS; ++j) {
    const int k = c0 + 2 * j + half;
    tX[(2 * j + half) * TP + l31] = (ok && k < D) ? (v[j] - mean) * rstd : 0.f;   // padding columns / rows contribute 0
  }
}

// acc (2 tiles of 32 features) <- bias
__device__ __forceinline__ void init_bias(f32x16 (&acc)[2], const float *sB, int half) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b = vec4_of(sB, t, q, half);
      acc[t][4 * q + 0] = b.x; acc[t][4 * q + 1] = b.y; acc[t][4 * q + 2] = b.z; acc[t][4 * q + 3] = b.w;
    }
}

// act + LayerNorm(64) statistics in the accumulator layout.  On return acc holds a = act(z).
template <bool RELU>
__device__ __forceinline__ void act_ln_stats(f32x16 (&acc)[2], float &mean, float &rstd) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[t][r] = act_fwd<RELU>(acc[t][r]); s += acc[t][r]; }
  mean = xhalf_sum(s) * (1.f / HID);
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float c = acc[t][r] - mean; q += c * c; }
  rstd = 1.0f / sqrtf(xhalf_sum(q) * (1.f / HID) + LN_EPS);
}

// tile[f][s] <- xhat = (a - mean) * rstd   (the LayerNorm affine is applied by whoever reads the tile)
__device__ __forceinline__ void xhat_to_tile(float *tile, const f32x16 (&a)[2], float mean, float rstd, int l31, int half) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[(32 * t + ROWMAP(r, half)) * TP + l31] = (a[t][r] - mean) * rstd;
}

// acc[t] += W-tile . (tin * gamma + beta)   (forward layer; weights k-major in LDS, K = 2*ksteps); operands of
// step kk+1 are fetched from LDS before the MFMAs of step kk issue
__device__ __forceinline__ void layer_mfma(f32x16 (&acc)[2], const float *sW, const float *tin, const float *sG,
                                           const float *sBt, int ksteps, int l31, int half) {
  // unrolled so that hipcc issues the LDS reads of several k-steps ahead of the MFMA chain that consumes them
#pragma unroll 8
  for (int kk = 0; kk < ksteps; ++kk) {
    const int k = 2 * kk + half;
    const float b = tin[k * TP + l31] * sG[k] + sBt[k];
    const float a0 = sW[k * WP + l31], a1 = sW[k * WP + 32 + l31];
    acc[0] = mfma(a0, b, acc[0]);
    acc[1] = mfma(a1, b, acc[1]);
  }
}

template <int LN>
struct TileStats {
  float mean[LN + 1], rstd[LN + 1];
  uint32_t pos[LN + 1];   // bit (16*t + r): post-activation value > 0 (exact ReLU gate for the backward pass)
};

__device__ __forceinline__ uint32_t positive_mask(const f32x16 (&a)[2]) {
  uint32_t mk = 0u;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) mk |= (a[t][r] > 0.f ? 1u : 0u) << (16 * t + r);
  return mk;
}

// LayerNorm affine parameters (LDS offsets) of the tile that feeds hidden layer l / the head
template <int LN>
__device__ __forceinline__ int ln_w_of(const LdsMap &m, int l) { return l == 0 ? m.ln1_w : m.ln2_w[l - 1]; }
template <int LN>
__device__ __forceinline__ int ln_b_of(const LdsMap &m, int l) { return l == 0 ? m.ln1_b : m.ln2_b[l - 1]; }

// forward of one 32-sample tile: tX (xhat0) -> tH[0..LN] (xhat of every LayerNorm); statistics kept for backward
template <bool RELU, int LN>
__device__ __forceinline__ void tile_forward_rest(const float *lds, const LdsMap &m, float *tH, f32x16 (&acc)[2], int l31, int half,
                                                  TileStats<LN> &st);

template <bool RELU, int LN>
__device__ __forceinline__ void tile_forward(const float *lds, const LdsMap &m, float *tX, float *tH, int D, int l31, int half,
                                             TileStats<LN> &st) {
  const int Dp = (D + 1) & ~1;
  f32x16 acc[2];
  init_bias(acc, lds + m.b1, half);
  layer_mfma(acc, lds + m.w1, tX, lds + m.fn_w, lds + m.fn_b, Dp / 2, l31, half);
  tile_forward_rest<RELU, LN>(lds, m, tH, acc, l31, half, st);
}

// wide inputs: layer 1 accumulated over 64-column chunks; every wave of the workgroup must call this the same
// number of times (block barriers around the shared W1 chunk)
template <bool RELU, int LN>
__device__ __forceinline__ void tile_forward_wide(float *lds, const LdsMap &m, const float *__restrict__ w1, const float *__restrict__ xr,
                                                  bool ok, float mean0, float rstd0, float *tX, float *tH, int D, int l31, int half,
                                                  TileStats<LN> &st) {
  f32x16 acc[2];
  init_bias(acc, lds + m.b1, half);
  for (int c0 = 0; c0 < D; c0 += MAXD) {
    const int kc = min(MAXD, D - c0);
    __syncthreads();                                   // the previous chunk of W1 is no longer being read
    stage_w1_chunk(lds + m.w1, w1, D, c0, kc);
    wide_commit_chunk(tX, xr, D, c0, ok, mean0, rstd0, l31, half);
    __syncthreads();
    layer_mfma(acc, lds + m.w1, tX, lds + m.fn_w + c0, lds + m.fn_b + c0, (kc + 1) / 2, l31, half);
  }
  tile_forward_rest<RELU, LN>(lds, m, tH, acc, l31, half, st);
}

template <bool RELU, int LN>
__device__ __forceinline__ void tile_forward_rest(const float *lds, const LdsMap &m, float *tH, f32x16 (&acc)[2], int l31, int half,
                                                  TileStats<LN> &st) {
  act_ln_stats<RELU>(acc, st.mean[0], st.rstd[0]);
  st.pos[0] = positive_mask(acc);
  xhat_to_tile(tH, acc, st.mean[0], st.rstd[0], l31, half);
  wave_lds_sync();
#pragma unroll
  for (int l = 0; l < LN; ++l) {
    init_bias(acc, lds + m.b2[l], half);
    layer_mfma(acc, lds + m.w2[l], tH + l * HID * TP, lds + ln_w_of<LN>(m, l), lds + ln_b_of<LN>(m, l), HID / 2, l31, half);
    act_ln_stats<RELU>(acc, st.mean[l + 1], st.rstd[l + 1]);
    st.pos[l + 1] = positive_mask(acc);
    xhat_to_tile(tH + (l + 1) * HID * TP, acc, st.mean[l + 1], st.rstd[l + 1], l31, half);
    wave_lds_sync();
  }
}

// head: out^T[a][s] (a < 32) = Wh . h_last + bh, accumulator layout
__device__ __forceinline__ f32x16 head_forward(const float *lds, const LdsMap &m, const float *tLast, const float *sG,
                                               const float *sBt, int l31, int half) {
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 b = *reinterpret_cast<const float4 *>(lds + m.bh + 8 * q + 4 * half);
    acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
  }
  const float *sW = lds + m.wh;
#pragma unroll 16
  for (int kk = 0; kk < HID / 2; ++kk) {
    const int k = 2 * kk + half;
    acc = mfma(sW[k * HP + l31], tLast[k * TP + l31] * sG[k] + sBt[k], acc);
  }
  return acc;
}


// ------------------------------------------------------------------------------------------------
// forward kernel.  MODE 0: out[B][A] = head output.   MODE 1: sample/argmax + log-prob (get_actions).
// ------------------------------------------------------------------------------------------------
struct FwdArgs {
  const float *params, *x;
  const int32_t *rows;
  const float *avail;
  float *out, *actions, *logp;
  mappo_net_desc desc;
  NetOff off;
  LdsMap map;
  int64_t B;
  int deterministic;
  uint64_t seed, counter;
  const uint64_t *counter_dev;   // optional device word added to `counter` (lets a captured hipGraph draw fresh numbers)
  // optional strided source rows (x_M > 0): sample i = (n, m) = (i / x_M, i % x_M) starts at x[n * x_sn + m * x_sm] — the
  // env's output read in place (fused rollout step); x_M == 0: contiguous rows x[i * in_dim]
  int64_t x_sn, x_sm;
  int x_M;
  // MODE 2 through forward16_tail (wide inputs) only: write the trunk output BLOCKED per 16-row tile, out[((i >> 4) * 4 + b) * 256 +
  // 4 * lane + r] = feature 16 b + 4 q + r of row i (B a multiple of 16) — the layout of the recurrent training kernels (gru_train16.hip)
  int out_blocked;
};

// XW: 0 = in_dim <= 32, 1 = in_dim <= 64 (rows prefetched into registers), 2 = in_dim > 64 (K-chunked layer 1)
// workgroup `bid` of `nb` workgroups that share the B rows (blockIdx / gridDim of a plain forward launch)
template <bool RELU, int LN, int MODE, int XW>
__device__ __forceinline__ void forward_body(const FwdArgs &p, float *lds, const int bid, const int nb) {
  constexpr bool WIDE = XW >= 1, XWIDE = XW == 2;
  const int n_waves = blockDim.x / WAVE;
  const NetOff &o = p.off;
  const LdsMap &m = p.map;
  const int lane = threadIdx.x & (WAVE - 1), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)), l31 = lane & 31, half = lane >> 5;
  const int D = p.desc.in_dim, Dp = (D + 1) & ~1, A = p.desc.out_dim;
  const uint32_t magic = (uint32_t)(0x100000000ull / (uint32_t)D) + 1u;
  const int64_t n_tiles = (p.B + TS - 1) / TS;
  const int64_t tile_stride = (int64_t)nb * n_waves;
  const int64_t n_btiles = (n_tiles + n_waves - 1) / n_waves;    // the tile loop is uniform over the workgroup's waves
  RowPrefetch<WIDE> pf;
  if (!XWIDE) prefetch_rows(pf, p.x, p.rows, ((int64_t)bid * n_waves + wave) * TS, p.B, D, lane, p.x_sn, p.x_sm, p.x_M);   // under the staging
  stage_all_weights<LN>(lds, m, p.params, o, p.desc);
  __syncthreads();
  float *tX = lds + m.tiles + wave * m.wave_stride;
  float *tH = tX + m.x_rows * TP;
  float *tZ = tH + (LN + 1) * HID * TP;
  for (int64_t tb = bid; tb < n_btiles; tb += nb) {
    const int64_t tile = tb * n_waves + wave;
    const int64_t base = tile * TS;
    int n_valid;
    TileStats<LN> st;
    if (!XWIDE) {
      n_valid = pf.n_valid;
      commit_rows(tX, tH + ((4 - ((m.x_rows * TP) & 3)) & 3), pf, D, magic, lane, p.desc.use_feature_norm != 0);   // 16-B aligned staging
      wave_lds_sync();
      prefetch_rows(pf, p.x, p.rows, (tile + tile_stride) * TS, p.B, D, lane, p.x_sn, p.x_sm, p.x_M);
      tile_forward<RELU, LN>(lds, m, tX, tH, D, l31, half, st);
    } else {
      n_valid = (int)max((int64_t)0, min((int64_t)TS, p.B - base));
      const bool ok = l31 < n_valid;
      const int64_t row = ok ? (p.rows ? (int64_t)p.rows[base + l31] : base + l31) : 0;
      const float *xr = p.x + row * D;
      float mean0, rstd0;
      wide_row_stats(xr, D, ok, half, p.desc.use_feature_norm != 0, mean0, rstd0);
      tile_forward_wide<RELU, LN>(lds, m, p.params + o.w1, xr, ok, mean0, rstd0, tX, tH, D, l31, half, st);
    }
    if (MODE == 2) {
      // trunk features (LayerNorm output of the last layer, affine applied) feature-major: out[f][B], the input
      // layout of the GRU kernels (gru.hip); a register's 32 lanes write one 128-B segment
      const float *tL = tH + LN * HID * TP, *sG = lds + ln_w_of<LN>(m, LN), *sBt = lds + ln_b_of<LN>(m, LN);
      if (l31 < n_valid) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int f = 32 * t + ROWMAP(r, half);
            p.out[(int64_t)f * p.B + base + l31] = tL[f * TP + l31] * sG[f] + sBt[f];
          }
      }
      wave_lds_sync();
      continue;
    }
    const f32x16 z = head_forward(lds, m, tH + LN * HID * TP, lds + ln_w_of<LN>(m, LN), lds + ln_b_of<LN>(m, LN), l31, half);
    head_to_tile(tZ, z, A, l31, half);
    wave_lds_sync();
    if (MODE == 0) {
      for (int e = lane; e < n_valid * A; e += WAVE) {
        const int s = e / A, a = e - s * A;
        p.out[base * A + e] = tZ[s * TP + a];
      }
    } else {
      if (lane < n_valid) {
        const int64_t i = base + lane;
        const uint64_t ctr = p.counter + (p.counter_dev ? *p.counter_dev : 0ull);
        float action, logp;
        categorical_act_lane(tZ + lane * TP, A, p.avail ? p.avail + i * A : nullptr, p.deterministic != 0, p.seed, ctr, (uint64_t)i,
                             action, logp);
        p.actions[i] = action;
        p.logp[i] = logp;
      }
    }
    wave_lds_sync();
  }
}

template <bool RELU, int LN, int MODE, int XW>
__global__ __launch_bounds__(256, 1) void mlp_forward_kernel(FwdArgs p) {
  extern __shared__ __align__(16) float lds[];
  forward_body<RELU, LN, MODE, XW>(p, lds, blockIdx.x, gridDim.x);
}

// Fused rollout step (K7 + K8 + K1 in ONE launch): workgroups [0, nA) run the actor's get_actions, [nA, nA + nC) the
// critic's get_values, the rest copy the env output the rows come from into the buffer slots (insert_core.h).  All
// three read only their sources and write disjoint outputs, so there is nothing to order inside the launch.
#include "insert_core.h"
struct StepArgs {
  FwdArgs a, c;
  InsertArgs ins;
  int nA, nC, nI;
};
#include "mlp_fwd16.h"
template <bool RELU, int LN>
__global__ __launch_bounds__(256, 1) void rollout_step_kernel(StepArgs s) {
  extern __shared__ __align__(16) float lds[];
  const int bid = blockIdx.x;
  if (bid < s.nA) forward16r_body<RELU, LN, 1>(s.a, lds, bid, s.nA);
  else if (bid < s.nA + s.nC) forward16r_body<RELU, LN, 0>(s.c, lds, bid - s.nA, s.nC);
  else insert_mpe_body(s.ins, bid - s.nA - s.nC, s.nI);
}

// trunk features of a recurrent network (mappo_mlp_features, in_dim <= 64) on the same register-resident 16x16x4 path
template <bool RELU, int LN>
__global__ __launch_bounds__(256, 1) void features16_kernel(FwdArgs a) {
  extern __shared__ __align__(16) float lds[];
  forward16r_body<RELU, LN, 2>(a, lds, blockIdx.x, gridDim.x);
}
template <bool RELU, int LN>
__global__ __launch_bounds__(256, 1) void features16_dual_kernel(FwdArgs a, FwdArgs c, int nA) {
  extern __shared__ __align__(16) float lds[];
  if ((int)blockIdx.x < nA) forward16r_body<RELU, LN, 2>(a, lds, blockIdx.x, nA);
  else forward16r_body<RELU, LN, 2>(c, lds, blockIdx.x - nA, gridDim.x - nA);
}

// ------------------------------------------------------------------------------------------------
// diagnostic build only (-DMLP_STAMPS, scripts/stamps.py): per-phase cycle shares of the update kernel.
// In the product build STAMP() expands to nothing and no stamp executes.
// ------------------------------------------------------------------------------------------------
#ifdef MLP_STAMPS
#define N_STAMPS 24
static unsigned long long *g_stamp_host = nullptr;          // [gridDim.x][N_STAMPS], set by mappo_debug_set_stamps (main TU)
#define STAMP_DECL unsigned long long st_acc_[N_STAMPS] = {}; unsigned long long st_prev_ = __builtin_readcyclecounter();
#define STAMP(i)                                                          \
  do {                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                    \
    const unsigned long long now_ = __builtin_readcyclecounter();         \
    st_acc_[i] += now_ - st_prev_;                                        \
    st_prev_ = now_;                                                      \
    __builtin_amdgcn_sched_barrier(0);                                    \
  } while (0)
#define STAMP_FLUSH()                                                                     \
  do {                                                                                    \
    if (p.stamps && threadIdx.x == 0)                                                     \
      for (int i_ = 0; i_ < N_STAMPS; ++i_) p.stamps[blockIdx.x * N_STAMPS + i_] = st_acc_[i_]; \
  } while (0)
#ifdef MLP_TU_MAIN
extern "C" int mappo_debug_set_stamps(unsigned long long *buf) { g_stamp_host = buf; return 0; }
#endif
#else
#define STAMP_DECL
#define STAMP(i) do { } while (0)
#define STAMP_FLUSH() do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// update kernel: forward + head gradient (external | PPO actor loss | value loss) + backward
// ------------------------------------------------------------------------------------------------
struct UpdArgs {
  const float *params, *x;
  const int32_t *rows;
  float *slabs;
  int64_t slab_stride, slab_col0;
  mappo_net_desc desc;
  NetOff off;
  LdsMap map;
  int64_t B;
  int n_blocks;              // grid size = slab rows written (0: one workgroup per CU, capped by the tile count)
  int n_regions;             // LDS regions of P floats used for the end-of-kernel reduction (2 when they fit)
  int p_red;                 // end of the flat parameter range this launch reduces (trunk only for HEAD 3)
  int red_base;              // start of that range (b1 for wide inputs: W1 / feature-norm grads come from wide_l1_bwd_kernel)
  float *wide_ws;            // wide inputs: [64][B] dz1 (feature-major) | mean0[B] | rstd0[B]
  const float *dHT;          // HEAD 3: gradient w.r.t. the trunk output, feature-major [64][B]
  int seq_nc;                // HEAD 3 (wide inputs: seq_nc a multiple of 16, so the tiles of the flat order ARE the sequence tiles): > 0 = the B rows are a time-major [L][seq_nc] minibatch tiled per (t, 16 sequences) and dHT is BLOCKED per tile (gru_train16.hip)
  // HEAD 0
  const float *dout;
  // HEAD 1 / 2 (buffer-order arrays, indexed by rows)
  const float *avail, *actions, *old_logp, *adv, *active, *v_old, *returns, *vn_state;
  const double *mb_moments;
  double *partials;          // [gridDim.x][4]
  mappo_ppo_cfg cfg;
  unsigned long long *stamps;   // diagnostic build (-DMLP_STAMPS) only
};

// sum over the 32 samples of row `f` (= lane) of a [64][TP] tile
__device__ __forceinline__ float tile_row_sum(const float *tile, int lane) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll 4
  for (int j = 0; j < TS; j += 2) { s0 += tile[lane * TP + j]; s1 += tile[lane * TP + j + 1]; }
  return s0 + s1;
}

__device__ __forceinline__ void regs_to_tile(float *tile, const f32x16 (&v)[2], int l31, int half) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[(32 * t + ROWMAP(r, half)) * TP + l31] = v[t][r];
}

// LayerNorm + activation backward in the accumulator layout.
//   in : dH = d/d(h) with h = xhat*gamma + beta the LayerNorm output; `tile` holds xhat
//   out: dH <- d/d(z) (pre-activation), also written over `tile` (each lane rewrites exactly the words it read)
// AFFINE = false (every LayerNorm that feeds a weight matrix of this kernel): the LayerNorm weight/bias gradients
// are NOT accumulated here — they follow from the raw products G = dz_next . xhat^T the dW MFMAs accumulate anyway
// (d gamma[k] = sum_f W_next[f][k] G[f][k], d beta[k] = sum_f W_next[f][k] db_next[f]; see the epilogue).
// AFFINE = true (HEAD 3: the gradient arrives at the trunk output, no weight matrix behind it): gG/gB (lane =
// feature) += sum_s dy*xhat, sum_s dy through two transposed row sums.
template <bool RELU, bool AFFINE>
__device__ __forceinline__ void ln_act_backward(f32x16 (&dH)[2], float *tile, float mean, float rstd, uint32_t pos,
                                                const float *sG, float &gG, float &gB, int lane, int l31, int half) {
  f32x16 xh[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) xh[t][r] = tile[(32 * t + ROWMAP(r, half)) * TP + l31];
  if (AFFINE) {
    wave_lds_sync();
    regs_to_tile(tile, dH, l31, half);
    wave_lds_sync();
    gB += tile_row_sum(tile, lane);
    wave_lds_sync();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) tile[(32 * t + ROWMAP(r, half)) * TP + l31] = dH[t][r] * xh[t][r];
    wave_lds_sync();
    gG += tile_row_sum(tile, lane);
    wave_lds_sync();
  }
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 g4 = vec4_of(sG, t, q, half);
      const float gq[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int r = 4 * q + c;
        const float dxh = dH[t][r] * gq[c];
        dH[t][r] = dxh;
        m1 += dxh;
        m2 += dxh * xh[t][r];
      }
    }
  m1 = xhalf_sum(m1) * (1.f / HID);
  m2 = xhalf_sum(m2) * (1.f / HID);
  const float inv_rstd = 1.0f / rstd;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float da = rstd * (dH[t][r] - m1 - xh[t][r] * m2);
      if (RELU) {
        dH[t][r] = ((pos >> (16 * t + r)) & 1u) ? da : 0.f;          // exact gate saved by the forward
      } else {
        const float a = xh[t][r] * inv_rstd + mean;                  // tanh output recovered from xhat
        dH[t][r] = da * (1.f - a * a);
      }
    }
  regs_to_tile(tile, dH, l31, half);
  wave_lds_sync();
}

// per-sample inputs of the in-kernel loss heads, prefetched one tile ahead (lanes 0..31 hold one sample each)
struct LossPrefetch {
  float f0, f1, f2, f3;     // actor: action, old_logp, adv, active   | critic: v_old, ret, active, -
  uint32_t dead;            // actor: bit a set <=> available_actions[a] == 0
};

template <int HEAD>
__device__ __forceinline__ void prefetch_loss(LossPrefetch &lp, const UpdArgs &p, int64_t row, int n_valid, int lane, int A) {
  lp.f0 = lp.f1 = lp.f2 = lp.f3 = 0.f;
  lp.dead = 0u;
  if (HEAD == 0 || HEAD == 3 || lane >= n_valid) return;        // lanes 0..31 carry the per-sample loss inputs
  if (HEAD == 1) {
    lp.f0 = p.actions[row]; lp.f1 = p.old_logp[row]; lp.f2 = p.adv[row]; lp.f3 = p.active[row];
    if (p.avail) {
      const float *av = p.avail + row * A;
      for (int a = 0; a < A; ++a) lp.dead |= (av[a] == 0.f ? 1u : 0u) << a;
    }
  } else {
    lp.f0 = p.v_old[row]; lp.f1 = p.returns[row]; lp.f2 = p.active[row];
  }
}

// Epilogue of the update kernel, per wave and in registers: raw products -> gradient partials.
// With h_in = xhat_in*gamma + beta feeding z = W h_in + b,  G[f][k] = sum_s dz[f][s] xhat_in[k][s],  db[f] = sum_s dz[f][s]:
//   dW[f][k] = gamma[k] G[f][k] + beta[k] db[f]      d gamma[k] = sum_f W[f][k] G[f][k]      d beta[k] = sum_f W[f][k] db[f]
// (linear in G and db, so applying them to each wave's partial sums commutes with the reductions that follow).
// g[ti][tj]: accumulator tiles, rows f = 32 ti + ROWMAP(r, half), columns k = 32 tj + l31.  dbv: db, lane = f.
// sW: the consumer's weights in LDS, k-major (sW[k*wstride + f]).  On return g holds dW, dgam/dbet (lane = k) the affine grads.
template <int NTI>
__device__ __forceinline__ void raw_to_grad(f32x16 (&g)[NTI][2], float dbv, float *scr, const float *sW, int wstride, const float *sG,
                                            const float *sBt, int K, bool two_k_tiles, int lane, int l31, int half, float &dgam,
                                            float &dbet) {
  scr[lane] = dbv;
  wave_lds_sync();
  float dg[2] = {0.f, 0.f}, dt[2] = {0.f, 0.f};
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    if (tj == 1 && !two_k_tiles) break;
    const int k = 32 * tj + l31;
    const bool valid = k < K;
    const int kc = valid ? k : 0;
    const float gam = valid ? sG[kc] : 0.f, bet = valid ? sBt[kc] : 0.f;
#pragma unroll
    for (int ti = 0; ti < NTI; ++ti) {
      float w[16], d[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) { const int f = 32 * ti + ROWMAP(r, half); w[r] = sW[kc * wstride + f]; d[r] = scr[f]; }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float wv = valid ? w[r] : 0.f;
        dg[tj] += wv * g[ti][tj][r];
        dt[tj] += wv * d[r];
        g[ti][tj][r] = gam * g[ti][tj][r] + bet * d[r];
      }
    }
  }
  const float a0 = xhalf_sum(dg[0]), a1 = xhalf_sum(dg[1]), b0 = xhalf_sum(dt[0]), b1 = xhalf_sum(dt[1]);
  dgam = half ? a1 : a0;
  dbet = half ? b1 : b0;
  wave_lds_sync();
}

#ifdef EXP_WAVES8          // experiment (scripts/exp_waves.py): 8 waves per workgroup = 2 per SIMD, 256 registers each
#define UPD_THREADS 512
#else
#define UPD_THREADS 256
#endif
template <bool RELU, int LN, int HEAD, int XW>
__global__ __launch_bounds__(UPD_THREADS, 1) void mlp_update_kernel(UpdArgs p) {
  extern __shared__ __align__(16) float lds[];
  __shared__ double red_smem[16 * 4];
  constexpr bool WIDE = XW >= 1, XWIDE = XW == 2;
  const int n_waves = blockDim.x / WAVE;
  const NetOff &o = p.off;
  const LdsMap &m = p.map;
  const int lane = threadIdx.x & (WAVE - 1), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)), l31 = lane & 31, half = lane >> 5;
  const int D = p.desc.in_dim, Dp = (D + 1) & ~1, A = p.desc.out_dim;
  const uint32_t magic = (uint32_t)(0x100000000ull / (uint32_t)D) + 1u;
  constexpr bool wide = WIDE;        // second 32-wide tile over the input features in use
  const int64_t n_tiles = (p.B + TS - 1) / TS;
  const int64_t tile_stride = (int64_t)gridDim.x * n_waves;
  const int64_t n_btiles = (n_tiles + n_waves - 1) / n_waves;    // uniform tile loop (block barriers in the wide path)
  RowPrefetch<WIDE> pf;
  LossPrefetch lp;
  STAMP_DECL
  if (!XWIDE) {
    prefetch_rows(pf, p.x, p.rows, ((int64_t)blockIdx.x * n_waves + wave) * TS, p.B, D, lane);
    prefetch_loss<HEAD>(lp, p, pf.my_row, pf.n_valid, lane, A);
  }
  stage_all_weights<LN>(lds, m, p.params, o, p.desc);
  __syncthreads();
  STAMP(0);   // staging
  float *tX = lds + m.tiles + wave * m.wave_stride;
  float *tH = tX + m.x_rows * TP;
  float *tZ = tH + (LN + 1) * HID * TP;

  // loss constants (HEAD 1/2): denominators are GLOBAL (mb_moments), see ppo_loss.hip
  LossScales ls = {0.f, 0.f, 0.f, 1.f};
  if (HEAD == 1 || HEAD == 2) ls = loss_scales(p.cfg, p.mb_moments, p.vn_state);
  double lacc[4] = {0.0, 0.0, 0.0, 0.0};   // actor: sum w*min(s1,s2), sum w*H, sum ratio | critic: sum w_v*l

  // ---- gradient accumulators (registers, live across the tile loop) ----
  f32x16 gWh[1][2], gW2[LN > 0 ? LN : 1][2][2], gW1[2][2];
  // raw products: gW*[f][k] = sum_s dz[f][s] * xhat_in[k][s] (LayerNorm affine of the input NOT applied), gB = sum_s dz.
  // The epilogue turns them into weight, LayerNorm-affine and feature-norm gradients.
  float gBh = 0.f;
  float gB[LN + 1];
  float gLnW = 0.f, gLnB = 0.f;          // HEAD 3 only: affine of the last LayerNorm (the gradient arrives behind it)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) gWh[0][i][r] = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        gW1[i][j][r] = 0.f;
#pragma unroll
        for (int l = 0; l < LN; ++l) gW2[l][i][j][r] = 0.f;
      }
  }
#pragma unroll
  for (int l = 0; l <= LN; ++l) gB[l] = 0.f;

  for (int64_t tb = blockIdx.x; tb < n_btiles; tb += gridDim.x) {
    const int64_t tile = tb * n_waves + wave;
    const int64_t base = tile * TS;
    int n_valid;
    LossPrefetch cur;
    TileStats<LN> st;
    float mean0 = 0.f, rstd0 = 1.f;
    if (!XWIDE) {
      n_valid = pf.n_valid;
      cur = lp;
      commit_rows(tX, tH + ((4 - ((m.x_rows * TP) & 3)) & 3), pf, D, magic, lane, p.desc.use_feature_norm != 0);   // 16-B aligned staging
      wave_lds_sync();
      prefetch_rows(pf, p.x, p.rows, (tile + tile_stride) * TS, p.B, D, lane);   // next tile, hidden under the MFMAs below
      prefetch_loss<HEAD>(lp, p, pf.my_row, pf.n_valid, lane, A);
      STAMP(1);   // commit (+ feature norm) + prefetch issue
      tile_forward<RELU, LN>(lds, m, tX, tH, D, l31, half, st);
    } else {
      n_valid = (int)max((int64_t)0, min((int64_t)TS, p.B - base));
      const bool ok = l31 < n_valid;
      const int64_t row = ok ? (p.rows ? (int64_t)p.rows[base + l31] : base + l31) : 0;
      prefetch_loss<HEAD>(cur, p, row, n_valid, lane, A);
      const float *xr = p.x + row * D;
      wide_row_stats(xr, D, ok, half, p.desc.use_feature_norm != 0, mean0, rstd0);
      STAMP(1);
      tile_forward_wide<RELU, LN>(lds, m, p.params + o.w1, xr, ok, mean0, rstd0, tX, tH, D, l31, half, st);
    }
    float *tLast = tH + LN * HID * TP;
    STAMP(2);   // trunk forward

    // ---- head gradient into tZ[s][a] ----
    if (HEAD == 3) {
      // nothing: the gradient arrives at the trunk output (loaded below)
    } else if (HEAD == 0) {
      for (int e = lane; e < TS * A; e += WAVE) {
        const int s = e / A, a = e - s * A;
        tZ[s * TP + a] = (s < n_valid) ? p.dout[base * A + e] : 0.f;
      }
    } else {
      const f32x16 z = head_forward(lds, m, tLast, lds + ln_w_of<LN>(m, LN), lds + ln_b_of<LN>(m, LN), l31, half);
      if (HEAD == 1) {
        head_to_tile(tZ, z, A, l31, half);
        wave_lds_sync();
        if (lane < TS) {
          float *zl = tZ + lane * TP;
          if (lane < n_valid) {
            actor_loss_lane(zl, A, cur.dead, (int)cur.f0, cur.f1, cur.f2, cur.f3, p.cfg, ls.scale_pi, lacc);
          } else {
            for (int a = 0; a < A; ++a) zl[a] = 0.f;
          }
        }
      } else {
        // value loss (r_mappo.py:62-87); the value of sample s is register 0 of lane s (half 0)
        if (lane < TS) {
          float dvv = 0.f;
          if (lane < n_valid) dvv = critic_loss_lane(z[0], cur.f0, cur.f1, cur.f2, p.cfg, ls, lacc);
          tZ[lane * TP] = dvv;
        }
      }
    }
    wave_lds_sync();
    STAMP(3);   // head forward + loss

    // ---- (A) raw head products:  gWh[a][f] += sum_s dz[s][a] * xhat_last[f][s] ----
    if (HEAD != 3) {
      float bsum = 0.f;
#pragma unroll 2
      for (int ss = 0; ss < TS / 2; ++ss) {
        const int s = 2 * ss + half;
        const float av = (l31 < A) ? tZ[s * TP + l31] : 0.f;
        bsum += av;
        gWh[0][0] = mfma(av, tLast[l31 * TP + s], gWh[0][0]);
        gWh[0][1] = mfma(av, tLast[(32 + l31) * TP + s], gWh[0][1]);
      }
      gBh += xhalf_sum(bsum);
    }
    // ---- (B) d h_last = Wh^T . dz ----
    f32x16 dH[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) dH[t][r] = 0.f;
    if (HEAD == 3) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (l31 < n_valid) dH[t][r] = p.dHT[(int64_t)(32 * t + ROWMAP(r, half)) * p.B + base + l31];
    } else {
      const float *sW = lds + m.wh;
      for (int kk = 0; kk < (A + 1) / 2; ++kk) {
        const int a = 2 * kk + half;
        const float b = (a < A) ? tZ[l31 * TP + a] : 0.f;
        dH[0] = mfma(sW[l31 * HP + a], b, dH[0]);
        dH[1] = mfma(sW[(32 + l31) * HP + a], b, dH[1]);
      }
    }
    STAMP(4);   // head grads (A), (B)
    // ---- hidden layers, last to first ----
#pragma unroll
    for (int l = LN; l >= 1; --l) {
      float *tCur = tH + l * HID * TP;          // xhat of this layer's LayerNorm -> scratch -> dz
      float *tPrev = tH + (l - 1) * HID * TP;   // xhat of the layer's input
      if (HEAD == 3 && l == LN)
        ln_act_backward<RELU, true>(dH, tCur, st.mean[l], st.rstd[l], st.pos[l], lds + m.ln2_w[l - 1], gLnW, gLnB, lane, l31, half);
      else
        ln_act_backward<RELU, false>(dH, tCur, st.mean[l], st.rstd[l], st.pos[l], lds + m.ln2_w[l - 1], gLnW, gLnB, lane, l31, half);
      gB[l] += tile_row_sum(tCur, lane);
      STAMP(5);   // LN + act backward (hidden)
      // gW2[f_out][k_in] += sum_s dz[f_out][s] * xhat_prev[k_in][s]
      {
#pragma unroll 2
        for (int ss = 0; ss < TS / 2; ++ss) {
          const int s = 2 * ss + half;
          const float a0 = tCur[l31 * TP + s], a1 = tCur[(32 + l31) * TP + s];
          const float b0 = tPrev[l31 * TP + s], b1 = tPrev[(32 + l31) * TP + s];
          gW2[l - 1][0][0] = mfma(a0, b0, gW2[l - 1][0][0]);
          gW2[l - 1][0][1] = mfma(a0, b1, gW2[l - 1][0][1]);
          gW2[l - 1][1][0] = mfma(a1, b0, gW2[l - 1][1][0]);
          gW2[l - 1][1][1] = mfma(a1, b1, gW2[l - 1][1][1]);
        }
      }
      STAMP(6);   // dW2
      // d h_prev = W2^T . dz
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dH[t][r] = 0.f;
      {
        const float *sW = lds + m.w2[l - 1];
#pragma unroll 2
        for (int kk = 0; kk < HID / 2; ++kk) {
          const int fo = 2 * kk + half;
          const float b = tCur[fo * TP + l31];
          dH[0] = mfma(sW[l31 * WP + fo], b, dH[0]);
          dH[1] = mfma(sW[(32 + l31) * WP + fo], b, dH[1]);
        }
      }
      wave_lds_sync();
      STAMP(7);   // dH (hidden)
    }
    // ---- layer 1 ----
    {
      float *tCur = tH;
      if (HEAD == 3 && LN == 0)
        ln_act_backward<RELU, true>(dH, tCur, st.mean[0], st.rstd[0], st.pos[0], lds + m.ln1_w, gLnW, gLnB, lane, l31, half);
      else
        ln_act_backward<RELU, false>(dH, tCur, st.mean[0], st.rstd[0], st.pos[0], lds + m.ln1_w, gLnW, gLnB, lane, l31, half);
      gB[0] += tile_row_sum(tCur, lane);
      STAMP(8);   // LN + act backward (layer 1)
      if (XWIDE) {
        // wide inputs: dz1 (feature-major) and the row statistics go to HBM; wide_l1_bwd_kernel turns them into
        // dW1 and the feature-norm gradients (64 x in_dim accumulators do not fit one wave's registers)
        float *dz1T = p.wide_ws, *stats = p.wide_ws + (int64_t)HID * p.B;
        if (l31 < n_valid) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) dz1T[(int64_t)(32 * t + ROWMAP(r, half)) * p.B + base + l31] = dH[t][r];
          if (half == 0) { stats[base + l31] = mean0; stats[p.B + base + l31] = rstd0; }
        }
        wave_lds_sync();
        continue;
      }
      // gW1[f_out][k] += sum_s dz1[f_out][s] * xhat0[k][s]   (raw input rows when feature norm is off).  No dX pass:
      // the feature-norm gradients follow from gW1 and W1 in the epilogue.
      {
        // rows k >= D of tX hold finite fillers (commit_rows); the columns they produce are never reduced
        const int k0 = l31, k1 = 32 + l31;
#pragma unroll 2
        for (int ss = 0; ss < TS / 2; ++ss) {
          const int s = 2 * ss + half;
          const float a0 = tCur[l31 * TP + s], a1 = tCur[(32 + l31) * TP + s];
          const float b0 = tX[k0 * TP + s];
          gW1[0][0] = mfma(a0, b0, gW1[0][0]);
          gW1[1][0] = mfma(a1, b0, gW1[1][0]);
          if (wide) {
            const float b1 = tX[k1 * TP + s];
            gW1[0][1] = mfma(a0, b1, gW1[0][1]);
            gW1[1][1] = mfma(a1, b1, gW1[1][1]);
          }
        }
      }
      STAMP(9);   // dW1
      wave_lds_sync();
      STAMP(10);
    }
  }

  // ---- loss statistics of this workgroup ----
  if (HEAD == 1 || HEAD == 2) {
    block_sum<4>(lacc, red_smem);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double *q = p.partials + (size_t)blockIdx.x * 4 + k;
        *q = p.cfg.accumulate_partials ? *q + lacc[k] : lacc[k];
      }
    }
  }

  // ---- raw products -> gradient partials (per wave, registers; see raw_to_grad) ----
  const bool fnorm = p.desc.use_feature_norm != 0;
  float vLnW[LN + 1], vLnB[LN + 1], vFnW = 0.f, vFnB = 0.f;      // lane = k
  {
    float *scr = lds + m.scratch + wave * HID;
#pragma unroll
    for (int j = 0; j <= LN; ++j) { vLnW[j] = 0.f; vLnB[j] = 0.f; }
    if (HEAD != 3) {
      raw_to_grad<1>(gWh, (lane < TS && l31 < A) ? gBh : 0.f, scr, lds + m.wh, HP, lds + ln_w_of<LN>(m, LN), lds + ln_b_of<LN>(m, LN),
                     HID, true, lane, l31, half, vLnW[LN], vLnB[LN]);
    } else {
      vLnW[LN] = gLnW; vLnB[LN] = gLnB;
    }
#pragma unroll
    for (int l = LN - 1; l >= 0; --l)
      raw_to_grad<2>(gW2[l], gB[l + 1], scr, lds + m.w2[l], WP, lds + ln_w_of<LN>(m, l), lds + ln_b_of<LN>(m, l), HID, true, lane, l31,
                     half, vLnW[l], vLnB[l]);
    if (!XWIDE && fnorm) raw_to_grad<2>(gW1, gB[0], scr, lds + m.w1, WP, lds + m.fn_w, lds + m.fn_b, D, wide, lane, l31, half, vFnW, vFnB);
  }
  STAMP(11);

  // ---- reduce the waves' accumulators through LDS (two regions, waves pair up) and write the slab ----
  __syncthreads();
  const int rb = p.red_base;                         // first flat parameter this launch reduces (0, or b1 for wide inputs)
  const int P = p.p_red - rb;
  float *red0 = lds + m.tiles - rb;                  // indexed by absolute flat offsets >= rb                     // n_regions * P floats fit in the tile area (checked on the host)
  const int n_reg = p.n_regions;
  for (int round = 0; round < (n_waves + n_reg - 1) / n_reg; ++round) {
    if (wave / n_reg == round) {
      float *red = red0 + (wave % n_reg) * P;
      const bool first = (round == 0);
      // one accumulator tile: 16 old values are read, then 16 sums written (reads never wait on the writes)
      auto red_tile = [&](const f32x16 &acc, int idx0, int ld, bool valid) {
        if (!valid) return;
        float *q = red + idx0;
        if (first) {                                     // first wave of a region: plain stores, nothing to read
#pragma unroll
          for (int r = 0; r < 16; ++r) q[((r & 3) + 8 * (r >> 2)) * ld] = acc[r];
        } else {
          float old[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) old[r] = q[((r & 3) + 8 * (r >> 2)) * ld];
#pragma unroll
          for (int r = 0; r < 16; ++r) q[((r & 3) + 8 * (r >> 2)) * ld] = old[r] + acc[r];
        }
      };
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          const int col = 32 * tj + l31, row0 = 32 * ti + 4 * half;
          if (!XWIDE) red_tile(gW1[ti][tj], o.w1 + row0 * D + col, D, col < D);
#pragma unroll
          for (int l = 0; l < LN; ++l) red_tile(gW2[l][ti][tj], o.w2[l] + row0 * HID + col, HID, true);
        }
      // head: rows a = ROWMAP(r, half) < A only
      if (HEAD != 3)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { const int a = ROWMAP(r, half); old[r] = (!first && a < A) ? red[o.wh + a * HID + 32 * tj + l31] : 0.f; }
#pragma unroll
        for (int r = 0; r < 16; ++r) { const int a = ROWMAP(r, half); if (a < A) red[o.wh + a * HID + 32 * tj + l31] = old[r] + gWh[0][tj][r]; }
      }
      {
        constexpr int NVAL = 3 * (LN + 1) + 3;
        float vals[NVAL]; int idx[NVAL]; bool ok[NVAL];
        int n = 0;
        vals[n] = gB[0]; idx[n] = o.b1 + lane; ok[n++] = true;
        vals[n] = vLnW[0]; idx[n] = o.ln1_w + lane; ok[n++] = true;
        vals[n] = vLnB[0]; idx[n] = o.ln1_b + lane; ok[n++] = true;
#pragma unroll
        for (int l = 0; l < LN; ++l) {
          vals[n] = gB[l + 1]; idx[n] = o.b2[l] + lane; ok[n++] = true;
          vals[n] = vLnW[l + 1]; idx[n] = o.ln2_w[l] + lane; ok[n++] = true;
          vals[n] = vLnB[l + 1]; idx[n] = o.ln2_b[l] + lane; ok[n++] = true;
        }
        vals[n] = gBh; idx[n] = (HEAD != 3) ? o.bh + l31 : 0; ok[n++] = (HEAD != 3 && half == 0 && l31 < A);
        const bool fn = !XWIDE && fnorm && lane < D;
        vals[n] = vFnW; idx[n] = o.fn_w + lane; ok[n++] = fn;
        vals[n] = vFnB; idx[n] = o.fn_b + lane; ok[n++] = fn;
        float old[NVAL];
#pragma unroll
        for (int i = 0; i < NVAL; ++i) old[i] = (!first && ok[i]) ? red[idx[i]] : 0.f;
#pragma unroll
        for (int i = 0; i < NVAL; ++i) if (ok[i]) red[idx[i]] = old[i] + vals[i];
      }
    }
    __syncthreads();
  }
  STAMP(12);    // block reduction through LDS
  float *slab = p.slabs + (size_t)blockIdx.x * p.slab_stride + p.slab_col0 + rb;
  const float *redv = red0 + rb;
  if (n_reg > 1) {
    for (int e = threadIdx.x; e < P; e += blockDim.x) slab[e] = redv[e] + redv[P + e];
  } else {
    for (int e = threadIdx.x; e < P; e += blockDim.x) slab[e] = redv[e];
  }
  STAMP(13);    // slab write
  STAMP_FLUSH();
}

#include "mlp_upd2.h"
#include "mlp_upd16.h"
#include "mlp_wide16_args.h"
#if defined(MLP_TU_WIDE) || defined(MLP_TU_WIDE_FWD) || defined(MLP_TU_WIDE_SK)
#include "mlp_wide16.h"
#endif

// ---- launchers of the wide-input kernels (mlp_wide16.h), compiled in their own translation unit (mlp_wide.hip) ----
int wide16_launch_l1_fwd(const Wide16Args &w, dim3 grid, hipStream_t st);
int wide16_launch_forward(int mode, bool relu, int ln, bool small, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st,
                          const Wide16Args &w, const FwdArgs &a, const char *who);
int wide16_launch_l1_bwd(const WideBwd16Args &w, dim3 grid, hipStream_t st);
int wide16_launch_features_dual(bool relu, int ln, bool small, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const Wide16Args &wa,
                                const FwdArgs &a, const Wide16Args &wc, const FwdArgs &c, int nA);
template <bool R>
int wide16_launch_forward_r(int mode, int ln, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const Wide16Args &w, const FwdArgs &a,
                            const char *who);
template <bool R>
int wide16_launch_features_dual_r(int ln, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const Wide16Args &wa, const FwdArgs &a,
                                  const Wide16Args &wc, const FwdArgs &c, int nA);
template <bool R>
int wide16_launch_rollout_step_r(int ln, dim3 grid, size_t lds_bytes, hipStream_t st, const Wide16Args &wa, const FwdArgs &a,
                                 const Wide16Args &wc, const FwdArgs &c, int nA);
template <bool R>
int wide16_launch_features_resident_r(int ln, dim3 grid, hipStream_t st, const Wide16Args &w, const FwdArgs &a);
template <bool R>
int wide16_launch_rollout_full_r(int ln, dim3 grid, size_t lds_bytes, hipStream_t st, const Wide16Args &wa, const FwdArgs &a,
                                 const Wide16Args &wc, const FwdArgs &c, int nA, const InsertArgs *ins);
// split-K variants (one tile per 4-wave workgroup): step-sized batches
int wide16_launch_forward_sk(int mode, bool relu, int ln, dim3 grid, size_t lds_bytes, hipStream_t st, const Wide16Args &w, const FwdArgs &a,
                             const char *who);
int wide16_launch_features_sk_dual(bool relu, int ln, dim3 grid, size_t lds_bytes, hipStream_t st, const Wide16Args &wa, const FwdArgs &a,
                                   const Wide16Args &wc, const FwdArgs &c, int nA);
// trunks + GRU step + heads of both networks in one launch (wide_recurrent_step_dual_kernel); one tile per workgroup
struct WideStepIO {
  const SmacInsert *ins;        // or NULL: no fused insert (then `masks` is read; with it the row mask comes from ins->done)
  const float *actor_h0, *critic_h0, *masks, *avail;
  float *actor_h_last, *critic_h_last, *actions, *logp, *values;
  int Nc, deterministic;
  uint64_t seed, counter;
  const uint64_t *counter_dev;
};
int wide16_launch_recurrent_step_dual(bool relu, int ln, size_t lds_bytes, hipStream_t st, const Wide16Args &wa, const FwdArgs &a,
                                      const Wide16Args &wc, const FwdArgs &c, int nt16, const WideStepIO &io);
#define WIDE_SK_MAX_TILES 256          // per network: above, the streamed kernels fill the chip

#ifdef MLP_TU_WIDE
template <int NCH>
static int wide16_l1_fwd_one(const Wide16Args &w, dim3 grid, size_t lds_bytes, hipStream_t st) {
  static const hipError_t e_ = hipFuncSetAttribute((const void *)wide_l1_fwd16_kernel<NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(159 * 1024));
  if (e_ != hipSuccess) { mappo_set_error("wide_l1_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  hipLaunchKernelGGL((wide_l1_fwd16_kernel<NCH>), grid, dim3(512), lds_bytes, st, w);
  return MAPPO_OK;
}
int wide16_launch_l1_fwd(const Wide16Args &w, dim3 grid, hipStream_t st) {
  const int nch = (w.D + 63) / 64;                                                    // exact: the row-end chunk is static in the kernel
  const size_t lds_bytes = sizeof(float) * ((size_t)HID * 64 * nch + HID);            // W1' whole (fragment order) + folded bias
  switch (nch) {
    case 2: return wide16_l1_fwd_one<2>(w, grid, lds_bytes, st);
    case 3: return wide16_l1_fwd_one<3>(w, grid, lds_bytes, st);
    case 4: return wide16_l1_fwd_one<4>(w, grid, lds_bytes, st);
    case 5: return wide16_l1_fwd_one<5>(w, grid, lds_bytes, st);
    case 6: return wide16_l1_fwd_one<6>(w, grid, lds_bytes, st);
    case 7: return wide16_l1_fwd_one<7>(w, grid, lds_bytes, st);
    default: return wide16_l1_fwd_one<8>(w, grid, lds_bytes, st);
  }
}

#endif

#ifdef MLP_TU_WIDE_FWD
// compiled once per activation (mlp_wide_fwd_r{0,1}.hip: MLP_WIDE_RELU); 8-wave workgroups only — batches of up to 256 tiles take
// the split-K kernels, and above that the 8-wave groups have enough tiles
template <bool R, int L, int MODE, int NCH>
static int wide16_forward_one(dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const Wide16Args &w, const FwdArgs &a, const char *who) {
  static const hipError_t e_ = hipFuncSetAttribute((const void *)wide_forward16_kernel<R, L, MODE, 8, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(96 * 1024));
  if (e_ != hipSuccess) { mappo_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  const int pid = (MODE == 1) ? MAPPO_PROF_ACT : MAPPO_PROF_MLP_FWD;
  PROF_LAUNCH(pid, (wide_forward16_kernel<R, L, MODE, 8, NCH>), grid, block, lds_bytes, st, w, a);
  return MAPPO_OK;
}
template <bool R, int L, int MODE>
static int wide16_forward_nch(dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const Wide16Args &w, const FwdArgs &a, const char *who) {
  return w.D <= 256 ? wide16_forward_one<R, L, MODE, 4>(grid, block, lds_bytes, st, w, a, who)       // row block registers for 4 chunks instead of 8
                    : wide16_forward_one<R, L, MODE, 8>(grid, block, lds_bytes, st, w, a, who);
}
template <bool R, int L>
static int wide16_forward_mode(int mode, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const Wide16Args &w, const FwdArgs &a, const char *who) {
  if (mode == 0) return wide16_forward_nch<R, L, 0>(grid, block, lds_bytes, st, w, a, who);
  if (mode == 1) return wide16_forward_nch<R, L, 1>(grid, block, lds_bytes, st, w, a, who);
  return wide16_forward_nch<R, L, 2>(grid, block, lds_bytes, st, w, a, who);
}
template <bool R>
int wide16_launch_forward_r(int mode, int ln, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const Wide16Args &w, const FwdArgs &a,
                            const char *who) {
  if (ln == 0) return wide16_forward_mode<R, 0>(mode, grid, block, lds_bytes, st, w, a, who);
  if (ln == 1) return wide16_forward_mode<R, 1>(mode, grid, block, lds_bytes, st, w, a, who);
  return wide16_forward_mode<R, 2>(mode, grid, block, lds_bytes, st, w, a, who);
}
template int wide16_launch_forward_r<MLP_WIDE_RELU>(int, int, dim3, dim3, size_t, hipStream_t, const Wide16Args &, const FwdArgs &, const char *);

template <bool R, int L>
static int wide16_features_dual_one(dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const WideDualArgs &d) {
  static const hipError_t e_ = hipFuncSetAttribute((const void *)wide_features16_dual_kernel<R, L, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(96 * 1024));
  if (e_ != hipSuccess) { mappo_set_error("mlp_features_dual: hipFuncSetAttribute: %s", hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  PROF_LAUNCH(MAPPO_PROF_MLP_FWD, (wide_features16_dual_kernel<R, L, 8>), grid, block, lds_bytes, st, d);
  return MAPPO_OK;
}
template <bool R>
int wide16_launch_features_dual_r(int ln, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const Wide16Args &wa, const FwdArgs &a,
                                  const Wide16Args &wc, const FwdArgs &c, int nA) {
  WideDualArgs d;
  d.wa = wa; d.wc = wc; d.a = a; d.c = c; d.nA = nA;
  if (ln == 0) return wide16_features_dual_one<R, 0>(grid, block, lds_bytes, st, d);
  if (ln == 1) return wide16_features_dual_one<R, 1>(grid, block, lds_bytes, st, d);
  return wide16_features_dual_one<R, 2>(grid, block, lds_bytes, st, d);
}
template int wide16_launch_features_dual_r<MLP_WIDE_RELU>(int, dim3, dim3, size_t, hipStream_t, const Wide16Args &, const FwdArgs &,
                                                          const Wide16Args &, const FwdArgs &, int);

template <bool R, int L, int NCH>
static int wide16_rollout_step_one(dim3 grid, size_t lds_bytes, hipStream_t st, const WideStepArgs &s) {
  static const hipError_t e_ = hipFuncSetAttribute((const void *)wide_rollout_step_kernel<R, L, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(96 * 1024));
  if (e_ != hipSuccess) { mappo_set_error("rollout_step: hipFuncSetAttribute: %s", hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  PROF_LAUNCH(MAPPO_PROF_ACT, (wide_rollout_step_kernel<R, L, NCH>), grid, dim3(512), lds_bytes, st, s);
  return MAPPO_OK;
}
template <bool R>
int wide16_launch_rollout_step_r(int ln, dim3 grid, size_t lds_bytes, hipStream_t st, const Wide16Args &wa, const FwdArgs &a,
                                 const Wide16Args &wc, const FwdArgs &c, int nA) {
  WideStepArgs s;
  s.wa = wa; s.wc = wc; s.a = a; s.c = c; s.nA = nA;
  const bool big = wa.D > 256 || wc.D > 256;                      // row-block registers for 8 chunks instead of 4
  if (ln == 0) return big ? wide16_rollout_step_one<R, 0, 8>(grid, lds_bytes, st, s) : wide16_rollout_step_one<R, 0, 4>(grid, lds_bytes, st, s);
  if (ln == 1) return big ? wide16_rollout_step_one<R, 1, 8>(grid, lds_bytes, st, s) : wide16_rollout_step_one<R, 1, 4>(grid, lds_bytes, st, s);
  return big ? wide16_rollout_step_one<R, 2, 8>(grid, lds_bytes, st, s) : wide16_rollout_step_one<R, 2, 4>(grid, lds_bytes, st, s);
}
template int wide16_launch_rollout_step_r<MLP_WIDE_RELU>(int, dim3, size_t, hipStream_t, const Wide16Args &, const FwdArgs &, const Wide16Args &,
                                                         const FwdArgs &, int);

template <bool R, int L, int NCH>
static int wide16_features_resident_one(dim3 grid, hipStream_t st, const Wide16Args &w, const FwdArgs &a) {
  static const hipError_t e_ = hipFuncSetAttribute((const void *)wide_features16_resident_kernel<R, L, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(159 * 1024));
  if (e_ != hipSuccess) { mappo_set_error("mlp_features: hipFuncSetAttribute: %s", hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  const size_t lds_bytes = sizeof(float) * ((size_t)HID * 64 * NCH + HID + a.map.tiles);
  if (lds_bytes > 159 * 1024) { mappo_set_error("mlp_features: needs %zu B of LDS", lds_bytes); return MAPPO_EINVAL; }
  PROF_LAUNCH(MAPPO_PROF_MLP_FWD, (wide_features16_resident_kernel<R, L, NCH>), grid, dim3(512), lds_bytes, st, w, a);
  return MAPPO_OK;
}
template <bool R, int L>
static int wide16_features_resident_nch(dim3 grid, hipStream_t st, const Wide16Args &w, const FwdArgs &a) {
  switch ((w.D + 63) / 64) {
    case 2: return wide16_features_resident_one<R, L, 2>(grid, st, w, a);
    case 3: return wide16_features_resident_one<R, L, 3>(grid, st, w, a);
    case 4: return wide16_features_resident_one<R, L, 4>(grid, st, w, a);
    case 5: return wide16_features_resident_one<R, L, 5>(grid, st, w, a);
    case 6: return wide16_features_resident_one<R, L, 6>(grid, st, w, a);
    case 7: return wide16_features_resident_one<R, L, 7>(grid, st, w, a);
    default: return wide16_features_resident_one<R, L, 8>(grid, st, w, a);
  }
}
// layer_N <= 1 (the caller checks); grid: one workgroup per 8 tiles, at most one per CU
template <bool R>
int wide16_launch_features_resident_r(int ln, dim3 grid, hipStream_t st, const Wide16Args &w, const FwdArgs &a) {
  return ln == 0 ? wide16_features_resident_nch<R, 0>(grid, st, w, a) : wide16_features_resident_nch<R, 1>(grid, st, w, a);
}
template int wide16_launch_features_resident_r<MLP_WIDE_RELU>(int, dim3, hipStream_t, const Wide16Args &, const FwdArgs &);

template <bool R, int L, int NCH>
static int wide16_rollout_full_one(dim3 grid, size_t lds_bytes, hipStream_t st, const WideFullArgs &s) {
  static const hipError_t e_ = hipFuncSetAttribute((const void *)wide_rollout_full_kernel<R, L, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(159 * 1024));
  if (e_ != hipSuccess) { mappo_set_error("rollout_step: hipFuncSetAttribute: %s", hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  PROF_LAUNCH(MAPPO_PROF_ACT, (wide_rollout_full_kernel<R, L, NCH>), grid, dim3(512), lds_bytes, st, s);
  return MAPPO_OK;
}
// in_dim of BOTH networks == 64 NCH, NCH 4 or 8 (the caller checks); ins: rewards / masks of the fused insert or NULL
template <bool R>
int wide16_launch_rollout_full_r(int ln, dim3 grid, size_t lds_bytes, hipStream_t st, const Wide16Args &wa, const FwdArgs &a,
                                 const Wide16Args &wc, const FwdArgs &c, int nA, const InsertArgs *ins) {
  WideFullArgs s = {};
  s.wa = wa; s.wc = wc; s.a = a; s.c = c; s.nA = nA;
  if (ins) { s.ins = *ins; s.has_ins = 1; }
  const bool big = wa.D == 512;
  if (ln == 0) return big ? wide16_rollout_full_one<R, 0, 8>(grid, lds_bytes, st, s) : wide16_rollout_full_one<R, 0, 4>(grid, lds_bytes, st, s);
  if (ln == 1) return big ? wide16_rollout_full_one<R, 1, 8>(grid, lds_bytes, st, s) : wide16_rollout_full_one<R, 1, 4>(grid, lds_bytes, st, s);
  return big ? wide16_rollout_full_one<R, 2, 8>(grid, lds_bytes, st, s) : wide16_rollout_full_one<R, 2, 4>(grid, lds_bytes, st, s);
}
template int wide16_launch_rollout_full_r<MLP_WIDE_RELU>(int, dim3, size_t, hipStream_t, const Wide16Args &, const FwdArgs &, const Wide16Args &,
                                                         const FwdArgs &, int, const InsertArgs *);
#endif

#ifdef MLP_TU_WIDE_SK
template <bool R, int L, int MODE>
static int wide16_forward_sk_one(dim3 grid, size_t lds_bytes, hipStream_t st, const Wide16Args &w, const FwdArgs &a, const char *who) {
  static const hipError_t e_ = hipFuncSetAttribute((const void *)wide_forward16_sk_kernel<R, L, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(96 * 1024));
  if (e_ != hipSuccess) { mappo_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  const int pid = (MODE == 1) ? MAPPO_PROF_ACT : MAPPO_PROF_MLP_FWD;
  PROF_LAUNCH(pid, (wide_forward16_sk_kernel<R, L, MODE>), grid, dim3(256), lds_bytes, st, w, a);
  return MAPPO_OK;
}
template <bool R, int L>
static int wide16_forward_sk_mode(int mode, dim3 grid, size_t lds_bytes, hipStream_t st, const Wide16Args &w, const FwdArgs &a, const char *who) {
  if (mode == 0) return wide16_forward_sk_one<R, L, 0>(grid, lds_bytes, st, w, a, who);
  if (mode == 1) return wide16_forward_sk_one<R, L, 1>(grid, lds_bytes, st, w, a, who);
  return wide16_forward_sk_one<R, L, 2>(grid, lds_bytes, st, w, a, who);
}
int wide16_launch_forward_sk(int mode, bool relu, int ln, dim3 grid, size_t lds_bytes, hipStream_t st, const Wide16Args &w, const FwdArgs &a,
                             const char *who) {
  if (ln == 0) return relu ? wide16_forward_sk_mode<true, 0>(mode, grid, lds_bytes, st, w, a, who) : wide16_forward_sk_mode<false, 0>(mode, grid, lds_bytes, st, w, a, who);
  if (ln == 1) return relu ? wide16_forward_sk_mode<true, 1>(mode, grid, lds_bytes, st, w, a, who) : wide16_forward_sk_mode<false, 1>(mode, grid, lds_bytes, st, w, a, who);
  return relu ? wide16_forward_sk_mode<true, 2>(mode, grid, lds_bytes, st, w, a, who) : wide16_forward_sk_mode<false, 2>(mode, grid, lds_bytes, st, w, a, who);
}
template <bool R, int L>
static int wide16_features_sk_dual_one(dim3 grid, size_t lds_bytes, hipStream_t st, const WideDualArgs &d) {
  static const hipError_t e_ = hipFuncSetAttribute((const void *)wide_features16_sk_dual_kernel<R, L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(96 * 1024));
  if (e_ != hipSuccess) { mappo_set_error("mlp_features_dual: hipFuncSetAttribute: %s", hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  PROF_LAUNCH(MAPPO_PROF_MLP_FWD, (wide_features16_sk_dual_kernel<R, L>), grid, dim3(256), lds_bytes, st, d);
  return MAPPO_OK;
}
int wide16_launch_features_sk_dual(bool relu, int ln, dim3 grid, size_t lds_bytes, hipStream_t st, const Wide16Args &wa, const FwdArgs &a,
                                   const Wide16Args &wc, const FwdArgs &c, int nA) {
  WideDualArgs d;
  d.wa = wa; d.wc = wc; d.a = a; d.c = c; d.nA = nA;
  if (ln == 0) return relu ? wide16_features_sk_dual_one<true, 0>(grid, lds_bytes, st, d) : wide16_features_sk_dual_one<false, 0>(grid, lds_bytes, st, d);
  if (ln == 1) return relu ? wide16_features_sk_dual_one<true, 1>(grid, lds_bytes, st, d) : wide16_features_sk_dual_one<false, 1>(grid, lds_bytes, st, d);
  return relu ? wide16_features_sk_dual_one<true, 2>(grid, lds_bytes, st, d) : wide16_features_sk_dual_one<false, 2>(grid, lds_bytes, st, d);
}
template <bool R, int L>
static int wide16_recurrent_step_one(dim3 grid, size_t lds_bytes, hipStream_t st, const WideRecDualArgs &r) {
  static const hipError_t e_ = hipFuncSetAttribute((const void *)wide_recurrent_step_dual_kernel<R, L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(96 * 1024));
  if (e_ != hipSuccess) { mappo_set_error("recurrent_step_dual: hipFuncSetAttribute: %s", hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  PROF_LAUNCH(MAPPO_PROF_ACT, (wide_recurrent_step_dual_kernel<R, L>), grid, dim3(256), lds_bytes, st, r);
  return MAPPO_OK;
}
int wide16_launch_recurrent_step_dual(bool relu, int ln, size_t lds_bytes, hipStream_t st, const Wide16Args &wa, const FwdArgs &a,
                                      const Wide16Args &wc, const FwdArgs &c, int nt16, const WideStepIO &io) {
  WideRecDualArgs r = {};
  r.d.wa = wa; r.d.wc = wc; r.d.a = a; r.d.c = c; r.d.nA = nt16;
  GruFwdArgs &ga = r.ga, &gc = r.gc;
  ga.params = a.params; ga.off = a.off; ga.desc = a.desc; ga.h0 = io.actor_h0; ga.masks = io.masks; ga.L = 1; ga.Nc = io.Nc; ga.A = a.desc.out_dim;
  ga.head_mode = 2; ga.h_last = io.actor_h_last; ga.avail = io.avail; ga.actions = io.actions; ga.logp = io.logp;
  ga.deterministic = io.deterministic; ga.seed = io.seed; ga.counter = io.counter; ga.counter_dev = io.counter_dev;
  gc.params = c.params; gc.off = c.off; gc.desc = c.desc; gc.h0 = io.critic_h0; gc.masks = io.masks; gc.L = 1; gc.Nc = io.Nc; gc.A = 1;
  gc.head_mode = 1; gc.h_last = io.critic_h_last; gc.out = io.values;
  r.nI = 0;
  if (io.ins) {
    r.ins = *io.ins;
    ga.dones = gc.dones = io.ins->done; ga.done_M = gc.done_M = io.ins->M; ga.done_sn = gc.done_sn = io.ins->done_sn; ga.done_sm = gc.done_sm = io.ins->done_sm;
    const int64_t most = (int64_t)io.Nc * (io.ins->D > io.ins->S ? io.ins->D : io.ins->S);
    const int64_t ni = (most + 2047) / 2048;                     // ~8 elements per thread
    r.nI = (int)(ni < 1 ? 1 : (ni > 64 ? 64 : ni));
  }
  const dim3 grid((unsigned)(2 * nt16 + r.nI));
  if (ln == 0) return relu ? wide16_recurrent_step_one<true, 0>(grid, lds_bytes, st, r) : wide16_recurrent_step_one<false, 0>(grid, lds_bytes, st, r);
  if (ln == 1) return relu ? wide16_recurrent_step_one<true, 1>(grid, lds_bytes, st, r) : wide16_recurrent_step_one<false, 1>(grid, lds_bytes, st, r);
  return relu ? wide16_recurrent_step_one<true, 2>(grid, lds_bytes, st, r) : wide16_recurrent_step_one<false, 2>(grid, lds_bytes, st, r);
}

#endif

#ifdef MLP_TU_WIDE
int wide16_launch_forward(int mode, bool relu, int ln, bool small, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st,
                          const Wide16Args &w, const FwdArgs &a, const char *who) {
  (void)small;
  return relu ? wide16_launch_forward_r<true>(mode, ln, grid, block, lds_bytes, st, w, a, who)
              : wide16_launch_forward_r<false>(mode, ln, grid, block, lds_bytes, st, w, a, who);
}
int wide16_launch_features_dual(bool relu, int ln, bool small, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const Wide16Args &wa,
                                const FwdArgs &a, const Wide16Args &wc, const FwdArgs &c, int nA) {
  (void)small;
  return relu ? wide16_launch_features_dual_r<true>(ln, grid, block, lds_bytes, st, wa, a, wc, c, nA)
              : wide16_launch_features_dual_r<false>(ln, grid, block, lds_bytes, st, wa, a, wc, c, nA);
}

template <bool FN, bool GATHER>
static void wide16_l1_bwd_one(const WideBwd16Args &w, dim3 grid, hipStream_t st) {
  if ((w.D & 63) == 0) hipLaunchKernelGGL((wide_l1_bwd16_kernel<FN, GATHER, true>), grid, dim3(512), 0, st, w);      // whole chunks: branch-free tile loop
  else hipLaunchKernelGGL((wide_l1_bwd16_kernel<FN, GATHER, false>), grid, dim3(512), 0, st, w);
}
int wide16_launch_l1_bwd(const WideBwd16Args &w, dim3 grid, hipStream_t st) {
  const bool fn = w.fn_w >= 0;
  if (w.rows) { if (fn) wide16_l1_bwd_one<true, true>(w, grid, st); else wide16_l1_bwd_one<false, true>(w, grid, st); }
  else { if (fn) wide16_l1_bwd_one<true, false>(w, grid, st); else wide16_l1_bwd_one<false, false>(w, grid, st); }
  return MAPPO_OK;
}
#endif

#define LDS_LIMIT (160 * 1024)
#define LDS_STATIC 1024                      // static __shared__ of the kernels (reduction scratch), rounded up
#define LDS_DYN_MAX (LDS_LIMIT - LDS_STATIC) // what hipFuncAttributeMaxDynamicSharedMemorySize may be raised to
#define NUM_CU 256
#define UPD16_LDS_MAX (LDS_LIMIT - 256)          // the 16-sample-tile update kernels have no static __shared__

// One launch of mlp_update_kernel<RELU, LN, HEAD, *>.  The 72 instantiations of that kernel are spread over the
// translation units mlp_upd_r{0,1}_l{0,1,2}.hip (one (RELU, LN) pair each, compiled in parallel); mlp.hip holds the
// host entry points and every other kernel.
template <bool R, int L, int HEAD>
int upd_inst(int xw, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const UpdArgs &a, const char *who);
// pair kernel (mlp_upd2.h), in_dim <= 64: translation units mlp_upd2_r{0,1}_l{0,1,2}.hip
template <bool R, int L, int HEAD>
int upd2_inst(bool wide, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const UpdArgs &a, const char *who);
// actor + critic in one launch (mlp_update2_dual_kernel): translation units mlp_upd2d_r{0,1}_l{0,1,2}.hip
template <bool R, int L>
int upd2d_inst(bool wide_a, bool wide_c, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const DualArgs &d);
// one wave per 16-sample tile (mlp_upd16.h), in_dim <= 64, layer_N <= 1: translation units mlp_upd16_r{0,1}_l{0,1}.hip
template <bool R, int L>
int upd16_inst(int head, bool wide, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const Upd16Args &a);
template <bool R, int L>
int upd16d_inst(bool wide_a, bool wide_c, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const Dual16Args &d);
// the same network from the layer-1 pre-activations on (in_dim 65..512; layer 1 in mlp_wide16.h / wide_l1_bwd_kernel)
template <bool R, int L>
int upd16x_inst(int head, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const Upd16Args &a);

#ifdef MLP_TU_UPD
template <bool R, int L, int HEAD, int W>
static int upd_launch(dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const UpdArgs &a, const char *who) {
  static const hipError_t e_ = hipFuncSetAttribute((const void *)mlp_update_kernel<R, L, HEAD, W>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)LDS_DYN_MAX);
  if (e_ != hipSuccess) { mappo_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  PROF_LAUNCH(MAPPO_PROF_MLP_BWD, (mlp_update_kernel<R, L, HEAD, W>), grid, block, lds_bytes, st, a);
  return MAPPO_OK;
}
template <bool R, int L, int HEAD>
int upd_inst(int xw, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const UpdArgs &a, const char *who) {
  if (xw != 2) { mappo_set_error("%s: in_dim <= 64 is served by the pair kernel", who); return MAPPO_EINVAL; }
  return upd_launch<R, L, HEAD, 2>(grid, block, lds_bytes, st, a, who);
}
template int upd_inst<MLP_UPD_RELU, MLP_UPD_LN, 0>(int, dim3, dim3, size_t, hipStream_t, const UpdArgs &, const char *);
template int upd_inst<MLP_UPD_RELU, MLP_UPD_LN, 1>(int, dim3, dim3, size_t, hipStream_t, const UpdArgs &, const char *);
template int upd_inst<MLP_UPD_RELU, MLP_UPD_LN, 2>(int, dim3, dim3, size_t, hipStream_t, const UpdArgs &, const char *);
template int upd_inst<MLP_UPD_RELU, MLP_UPD_LN, 3>(int, dim3, dim3, size_t, hipStream_t, const UpdArgs &, const char *);
#endif

#ifdef MLP_TU_UPD2
template <bool R, int L, int HEAD, bool W>
static int upd2_launch(dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const UpdArgs &a, const char *who) {
  static const hipError_t e_ = hipFuncSetAttribute((const void *)mlp_update2_kernel<R, L, HEAD, W>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)LDS_DYN_MAX);
  if (e_ != hipSuccess) { mappo_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  PROF_LAUNCH(MAPPO_PROF_MLP_BWD, (mlp_update2_kernel<R, L, HEAD, W>), grid, block, lds_bytes, st, a);
  return MAPPO_OK;
}
template <bool R, int L, int HEAD>
int upd2_inst(bool wide, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const UpdArgs &a, const char *who) {
  return wide ? upd2_launch<R, L, HEAD, true>(grid, block, lds_bytes, st, a, who) : upd2_launch<R, L, HEAD, false>(grid, block, lds_bytes, st, a, who);
}
template int upd2_inst<MLP_UPD_RELU, MLP_UPD_LN, 0>(bool, dim3, dim3, size_t, hipStream_t, const UpdArgs &, const char *);
template int upd2_inst<MLP_UPD_RELU, MLP_UPD_LN, 1>(bool, dim3, dim3, size_t, hipStream_t, const UpdArgs &, const char *);
template int upd2_inst<MLP_UPD_RELU, MLP_UPD_LN, 2>(bool, dim3, dim3, size_t, hipStream_t, const UpdArgs &, const char *);
template int upd2_inst<MLP_UPD_RELU, MLP_UPD_LN, 3>(bool, dim3, dim3, size_t, hipStream_t, const UpdArgs &, const char *);
#endif

#ifdef MLP_TU_UPD2D
template <bool R, int L, bool WA, bool WC>
static int upd2d_launch(dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const DualArgs &d) {
  static const hipError_t e_ = hipFuncSetAttribute((const void *)mlp_update2_dual_kernel<R, L, WA, WC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)LDS_DYN_MAX);
  if (e_ != hipSuccess) { mappo_set_error("actor_critic_update: hipFuncSetAttribute: %s", hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  PROF_LAUNCH(MAPPO_PROF_MLP_BWD, (mlp_update2_dual_kernel<R, L, WA, WC>), grid, block, lds_bytes, st, d);
  return MAPPO_OK;
}
template <bool R, int L>
int upd2d_inst(bool wa, bool wc, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const DualArgs &d) {
  if (wa) return wc ? upd2d_launch<R, L, true, true>(grid, block, lds_bytes, st, d) : upd2d_launch<R, L, true, false>(grid, block, lds_bytes, st, d);
  return wc ? upd2d_launch<R, L, false, true>(grid, block, lds_bytes, st, d) : upd2d_launch<R, L, false, false>(grid, block, lds_bytes, st, d);
}
template int upd2d_inst<MLP_UPD_RELU, MLP_UPD_LN>(bool, bool, dim3, dim3, size_t, hipStream_t, const DualArgs &);
#endif

#ifdef MLP_TU_UPD16
template <bool R, int L, int HEAD, bool W>
static int upd16_launch(dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const Upd16Args &a) {
  static const hipError_t e_ = hipFuncSetAttribute((const void *)mlp_update16_kernel<R, L, HEAD, W>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)UPD16_LDS_MAX);
  if (e_ != hipSuccess) { mappo_set_error("update16: hipFuncSetAttribute: %s", hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  PROF_LAUNCH(MAPPO_PROF_MLP_BWD, (mlp_update16_kernel<R, L, HEAD, W>), grid, block, lds_bytes, st, a);
  return MAPPO_OK;
}
template <bool R, int L>
int upd16_inst(int head, bool wide, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const Upd16Args &a) {
  if (head == 1) return wide ? upd16_launch<R, L, 1, true>(grid, block, lds_bytes, st, a) : upd16_launch<R, L, 1, false>(grid, block, lds_bytes, st, a);
  if (head == 3) return wide ? upd16_launch<R, L, 3, true>(grid, block, lds_bytes, st, a) : upd16_launch<R, L, 3, false>(grid, block, lds_bytes, st, a);
  return wide ? upd16_launch<R, L, 2, true>(grid, block, lds_bytes, st, a) : upd16_launch<R, L, 2, false>(grid, block, lds_bytes, st, a);
}
template int upd16_inst<MLP_UPD_RELU, MLP_UPD_LN>(int, bool, dim3, dim3, size_t, hipStream_t, const Upd16Args &);

template <bool R, int L, bool WA, bool WC>
static int upd16d_launch(dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const Dual16Args &d) {
  static const hipError_t e_ = hipFuncSetAttribute((const void *)mlp_update16_dual_kernel<R, L, WA, WC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)UPD16_LDS_MAX);
  if (e_ != hipSuccess) { mappo_set_error("actor_critic_update: hipFuncSetAttribute: %s", hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  PROF_LAUNCH(MAPPO_PROF_MLP_BWD, (mlp_update16_dual_kernel<R, L, WA, WC>), grid, block, lds_bytes, st, d);
  return MAPPO_OK;
}
template <bool R, int L>
int upd16d_inst(bool wa, bool wc, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const Dual16Args &d) {
  if (wa) return wc ? upd16d_launch<R, L, true, true>(grid, block, lds_bytes, st, d) : upd16d_launch<R, L, true, false>(grid, block, lds_bytes, st, d);
  return wc ? upd16d_launch<R, L, false, true>(grid, block, lds_bytes, st, d) : upd16d_launch<R, L, false, false>(grid, block, lds_bytes, st, d);
}
template int upd16d_inst<MLP_UPD_RELU, MLP_UPD_LN>(bool, bool, dim3, dim3, size_t, hipStream_t, const Dual16Args &);

template <bool R, int L, int HEAD>
static int upd16x_launch(dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const Upd16Args &a) {
  static const hipError_t e_ = hipFuncSetAttribute((const void *)mlp_update16x_kernel<R, L, HEAD>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)UPD16_LDS_MAX);
  if (e_ != hipSuccess) { mappo_set_error("update16x: hipFuncSetAttribute: %s", hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  PROF_LAUNCH(MAPPO_PROF_MLP_BWD, (mlp_update16x_kernel<R, L, HEAD>), grid, block, lds_bytes, st, a);
  return MAPPO_OK;
}
template <bool R, int L>
int upd16x_inst(int head, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const Upd16Args &a) {
  if (head == 3) return upd16x_launch<R, L, 3>(grid, block, lds_bytes, st, a);
  return head == 1 ? upd16x_launch<R, L, 1>(grid, block, lds_bytes, st, a) : upd16x_launch<R, L, 2>(grid, block, lds_bytes, st, a);
}
template int upd16x_inst<MLP_UPD_RELU, MLP_UPD_LN>(int, dim3, dim3, size_t, hipStream_t, const Upd16Args &);
#endif

#if defined(MLP_TU_MAIN) || defined(MLP_TU_STEP)
// LDS map, layer-1 arguments and launch shape of the one-launch wide forward (mlp_wide16.h) for the network in `a` (a.off / a.map set)
static int wide_forward_prepare(FwdArgs &a, Wide16Args &w, size_t &lb, dim3 &grid, dim3 &block, bool &small, const char *who) {
  a.map.wave_stride = 16 * TP;                                   // the tail only needs the [16][TP] logits tile of a wave
  a.map.total = a.map.tiles + 8 * a.map.wave_stride;
  lb = (size_t)a.map.total * sizeof(float);
  MAPPO_REQUIRE(lb + sizeof(float) * (2 * HID * RS16 + HID) <= LDS_DYN_MAX, "%s: needs %zu B of LDS", who, lb);
  w = Wide16Args{};
  w.params = a.params; w.x = a.x; w.rows = a.rows; w.B = a.B; w.D = a.desc.in_dim; w.w1 = a.off.w1; w.b1 = a.off.b1;
  w.fn_w = a.desc.use_feature_norm ? a.off.fn_w : -1; w.fn_b = a.desc.use_feature_norm ? a.off.fn_b : -1;
  // 8 tiles (one per wave) share the weight stream; at most one workgroup per CU
  const int64_t n_tiles16 = (a.B + 15) / 16;
  small = false;                                               // (4-wave groups went away with the split-K kernels for <= 256 tiles)
  const int64_t n_groups = (n_tiles16 + 7) / 8;
  grid = dim3((unsigned)(n_groups < NUM_CU ? n_groups : NUM_CU));
  block = dim3(512);
  return MAPPO_OK;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int check_desc_common(const mappo_net_desc *d, const char *who) {
  MAPPO_REQUIRE(d, "%s: null desc", who);
  MAPPO_REQUIRE(d->hidden == HID, "%s: hidden_size %d unsupported (kernels are tiled for %d)", who, d->hidden, HID);
  MAPPO_REQUIRE(d->in_dim >= 1 && d->in_dim <= MAPPO_MAX_IN_DIM, "%s: in_dim %d outside [1,%d]", who, d->in_dim, MAPPO_MAX_IN_DIM);
  MAPPO_REQUIRE(d->out_dim >= 1 && d->out_dim <= MAPPO_MAX_ACTIONS, "%s: out_dim %d outside [1,%d]", who, d->out_dim,
                MAPPO_MAX_ACTIONS);
  MAPPO_REQUIRE(d->layer_N >= 0 && d->layer_N <= MAPPO_MAX_LAYER_N, "%s: layer_N %d outside [0,%d]", who, d->layer_N,
                MAPPO_MAX_LAYER_N);
  return MAPPO_OK;
}
static int check_desc(const mappo_net_desc *d, const char *who) {
  if (int rc = check_desc_common(d, who)) return rc;
  MAPPO_REQUIRE(!d->recurrent, "%s: recurrent networks go through mlp_features / gru_* / trunk_backward", who);
  return MAPPO_OK;
}
static int check_desc_trunk(const mappo_net_desc *d, const char *who) { return check_desc_common(d, who); }


static int fit_waves(const mappo_net_desc &d, int want) {
  int nw = want;
  while (nw > 1 && (size_t)lds_map(d, nw).total * sizeof(float) > LDS_DYN_MAX) nw >>= 1;
  return nw;
}

#endif

int launch_features16(const FwdArgs &a_in, hipStream_t st);      // defined in the step translation unit

#ifdef MLP_TU_STEP
template <bool R, int L>
static int features16_launch(dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const FwdArgs &a) {
  static const hipError_t e_ = hipFuncSetAttribute((const void *)features16_kernel<R, L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_DYN_MAX);
  if (e_ != hipSuccess) { mappo_set_error("mlp_features: hipFuncSetAttribute: %s", hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  hipLaunchKernelGGL((features16_kernel<R, L>), grid, block, lds_bytes, st, a);
  return MAPPO_OK;
}

int launch_features16(const FwdArgs &a_in, hipStream_t st) {
  MAPPO_CLEAR_STICKY();
  FwdArgs a = a_in;
  const int64_t n_tiles = (a.B + 15) / 16;
  const int nw = fit_waves(a.desc, n_tiles >= 4 ? 4 : (n_tiles >= 2 ? 2 : 1));
  a.off = net_offsets(a.desc); a.map = lds_map(a.desc, nw);
  const size_t lds_bytes = (size_t)a.map.total * sizeof(float);
  MAPPO_REQUIRE(lds_bytes <= LDS_DYN_MAX, "mlp_features: needs %zu B of LDS", lds_bytes);
  int64_t nb = (n_tiles + nw - 1) / nw;
  if (nb > 2 * NUM_CU) nb = 2 * NUM_CU;                      // every workgroup stages the weights once, then walks its tiles
  dim3 grid((unsigned)nb), block(WAVE * nw);
  const bool relu = a.desc.use_relu != 0;
  int rc;
  switch (a.desc.layer_N) {
    case 0: rc = relu ? features16_launch<true, 0>(grid, block, lds_bytes, st, a) : features16_launch<false, 0>(grid, block, lds_bytes, st, a); break;
    case 1: rc = relu ? features16_launch<true, 1>(grid, block, lds_bytes, st, a) : features16_launch<false, 1>(grid, block, lds_bytes, st, a); break;
    default: rc = relu ? features16_launch<true, 2>(grid, block, lds_bytes, st, a) : features16_launch<false, 2>(grid, block, lds_bytes, st, a); break;
  }
  if (rc) return rc;
  MAPPO_CHECK_LAUNCH("mlp_features");
  return MAPPO_OK;
}

template <bool R, int L>
static int features16_dual_launch(dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const FwdArgs &a, const FwdArgs &c, int nA) {
  static const hipError_t e_ = hipFuncSetAttribute((const void *)features16_dual_kernel<R, L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_DYN_MAX);
  if (e_ != hipSuccess) { mappo_set_error("mlp_features_dual: hipFuncSetAttribute: %s", hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  hipLaunchKernelGGL((features16_dual_kernel<R, L>), grid, block, lds_bytes, st, a, c, nA);
  return MAPPO_OK;
}

extern "C" int mappo_mlp_features_dual(const float *params_a, const mappo_net_desc *desc_a, const float *x_a, float *featT_a,
                                       const float *params_c, const mappo_net_desc *desc_c, const float *x_c, float *featT_c,
                                       int64_t B, mappo_stream_t stream) {
  if (int rc = check_desc_trunk(desc_a, "mlp_features_dual")) return rc;
  if (int rc = check_desc_trunk(desc_c, "mlp_features_dual")) return rc;
  MAPPO_REQUIRE((desc_a->in_dim <= MAXD) == (desc_c->in_dim <= MAXD) && desc_a->in_dim <= 512 && desc_c->in_dim <= 512,
                "mlp_features_dual: both networks narrow (in_dim <= %d) or both wide (<= 512)", MAXD);
  MAPPO_REQUIRE(desc_a->layer_N == desc_c->layer_N && desc_a->use_relu == desc_c->use_relu,
                "mlp_features_dual: the networks must share layer_N and the activation");
  MAPPO_REQUIRE(params_a && x_a && featT_a && params_c && x_c && featT_c && B > 0, "mlp_features_dual: bad arguments");
  MAPPO_CLEAR_STICKY();
  if (desc_a->in_dim > MAXD) {
    // wide inputs: the one-launch wide forward (mlp_wide16.h) of both networks by workgroup role
    FwdArgs a = {}, c = {};
    a.params = params_a; a.x = x_a; a.out = featT_a; a.desc = *desc_a; a.B = B; a.off = net_offsets(a.desc); a.map = lds_map(a.desc, 8);
    c.params = params_c; c.x = x_c; c.out = featT_c; c.desc = *desc_c; c.B = B; c.off = net_offsets(c.desc); c.map = lds_map(c.desc, 8);
    Wide16Args wa, wc;
    size_t lba, lbc;
    dim3 ga, gc, ba, bc;
    bool sa, sc;
    if (int rcp = wide_forward_prepare(a, wa, lba, ga, ba, sa, "mlp_features_dual")) return rcp;
    if (int rcp = wide_forward_prepare(c, wc, lbc, gc, bc, sc, "mlp_features_dual")) return rcp;
    const int64_t nt16 = (B + 15) / 16;
    if (nt16 <= WIDE_SK_MAX_TILES && !getenv("MAPPO_WIDE_NO_SK")) {
      if (int rcw = wide16_launch_features_sk_dual(desc_a->use_relu != 0, desc_a->layer_N, dim3((unsigned)(2 * nt16)), lba > lbc ? lba : lbc,
                                                   as_stream(stream), wa, a, wc, c, (int)nt16))
        return rcw;
    } else if (int rcw = wide16_launch_features_dual(desc_a->use_relu != 0, desc_a->layer_N, sa, dim3(ga.x + gc.x), ba, lba > lbc ? lba : lbc,
                                                     as_stream(stream), wa, a, wc, c, (int)ga.x))
      return rcw;
    MAPPO_CHECK_LAUNCH("mlp_features_dual");
    return MAPPO_OK;
  }
  const int64_t n_tiles = (B + 15) / 16;
  const int want = n_tiles >= 4 ? 4 : (n_tiles >= 2 ? 2 : 1);
  int nw = fit_waves(*desc_a, want);
  const int nwc = fit_waves(*desc_c, want);
  nw = nw < nwc ? nw : nwc;
  FwdArgs a = {}, c = {};
  a.params = params_a; a.x = x_a; a.out = featT_a; a.desc = *desc_a; a.B = B; a.off = net_offsets(a.desc); a.map = lds_map(a.desc, nw);
  c.params = params_c; c.x = x_c; c.out = featT_c; c.desc = *desc_c; c.B = B; c.off = net_offsets(c.desc); c.map = lds_map(c.desc, nw);
  const int totA = a.map.total, totC = c.map.total;
  const size_t lds_bytes = (size_t)(totA > totC ? totA : totC) * sizeof(float);
  MAPPO_REQUIRE(lds_bytes <= LDS_DYN_MAX, "mlp_features_dual: needs %zu B of LDS", lds_bytes);
  int64_t nb = (n_tiles + nw - 1) / nw;
  if (nb > NUM_CU) nb = NUM_CU;
  dim3 grid((unsigned)(2 * nb)), block(WAVE * nw);
  const bool relu = desc_a->use_relu != 0;
  int rc;
  switch (desc_a->layer_N) {
    case 0: rc = relu ? features16_dual_launch<true, 0>(grid, block, lds_bytes, as_stream(stream), a, c, (int)nb) : features16_dual_launch<false, 0>(grid, block, lds_bytes, as_stream(stream), a, c, (int)nb); break;
    case 1: rc = relu ? features16_dual_launch<true, 1>(grid, block, lds_bytes, as_stream(stream), a, c, (int)nb) : features16_dual_launch<false, 1>(grid, block, lds_bytes, as_stream(stream), a, c, (int)nb); break;
    default: rc = relu ? features16_dual_launch<true, 2>(grid, block, lds_bytes, as_stream(stream), a, c, (int)nb) : features16_dual_launch<false, 2>(grid, block, lds_bytes, as_stream(stream), a, c, (int)nb); break;
  }
  if (rc) return rc;
  MAPPO_CHECK_LAUNCH("mlp_features_dual");
  return MAPPO_OK;
}

// wide-input branch of mappo_recurrent_step_dual (gru.hip): both trunks (split-K), GRU steps and heads in one launch
#define WIDE_REC_STEP_MAX_TILES 1024
int mappo_recurrent_step_dual_wide_(const float *actor_params, const mappo_net_desc *actor_desc, const float *obs, const float *actor_h0,
                                    float *actor_h_last, const float *critic_params, const mappo_net_desc *critic_desc, const float *share_obs,
                                    const float *critic_h0, float *critic_h_last, const float *masks, int32_t Nc, const float *avail,
                                    int32_t deterministic, uint64_t seed, uint64_t counter, const uint64_t *counter_dev, float *actions,
                                    float *logp, float *values, const SmacInsert *ins, mappo_stream_t stream) {
  if (int rc = check_desc_trunk(actor_desc, "recurrent_step_dual")) return rc;
  if (int rc = check_desc_trunk(critic_desc, "recurrent_step_dual")) return rc;
  MAPPO_REQUIRE(actor_desc->in_dim > MAXD && critic_desc->in_dim > MAXD && actor_desc->in_dim <= 512 && critic_desc->in_dim <= 512,
                "recurrent_step_dual: both networks wide (in_dim %d..512)", MAXD + 1);
  const int64_t nt16 = ((int64_t)Nc + 15) / 16;
  MAPPO_REQUIRE(nt16 <= WIDE_REC_STEP_MAX_TILES, "recurrent_step_dual: %d rows exceed the one-launch step (mlp_features_dual + gru_step_dual)", Nc);
  MAPPO_CLEAR_STICKY();
  FwdArgs a = {}, c = {};
  a.params = actor_params; a.x = obs; a.desc = *actor_desc; a.B = Nc; a.off = net_offsets(a.desc); a.map = lds_map(a.desc, 8);
  c.params = critic_params; c.x = share_obs; c.desc = *critic_desc; c.B = Nc; c.off = net_offsets(c.desc); c.map = lds_map(c.desc, 8);
  Wide16Args wa, wc;
  size_t lba, lbc;
  dim3 ga, gc, ba, bc;
  bool sa, sc;
  if (int rcp = wide_forward_prepare(a, wa, lba, ga, ba, sa, "recurrent_step_dual")) return rcp;
  if (int rcp = wide_forward_prepare(c, wc, lbc, gc, bc, sc, "recurrent_step_dual")) return rcp;
  WideStepIO io = {};
  io.ins = ins;
  io.actor_h0 = actor_h0; io.critic_h0 = critic_h0; io.masks = masks; io.avail = avail; io.actor_h_last = actor_h_last;
  io.critic_h_last = critic_h_last; io.actions = actions; io.logp = logp; io.values = values; io.Nc = Nc; io.deterministic = deterministic;
  io.seed = seed; io.counter = counter; io.counter_dev = counter_dev;
  if (int rcw = wide16_launch_recurrent_step_dual(actor_desc->use_relu != 0, actor_desc->layer_N, lba > lbc ? lba : lbc, as_stream(stream), wa, a,
                                                  wc, c, (int)nt16, io))
    return rcw;
  MAPPO_CHECK_LAUNCH("recurrent_step_dual");
  return MAPPO_OK;
}

// ---- fused rollout step (rollout_step_kernel): translation unit mlp_step.hip --------------------------------------
template <bool R, int L>
static int step_launch(dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const StepArgs &a) {
  static const hipError_t e_ = hipFuncSetAttribute((const void *)rollout_step_kernel<R, L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_DYN_MAX);
  if (e_ != hipSuccess) { mappo_set_error("rollout_step: hipFuncSetAttribute: %s", hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  PROF_LAUNCH(MAPPO_PROF_ACT, (rollout_step_kernel<R, L>), grid, block, lds_bytes, st, a);
  return MAPPO_OK;
}

extern "C" int mappo_rollout_step(const float *actor_params, const mappo_net_desc *actor_desc, const float *critic_params,
                                  const mappo_net_desc *critic_desc, const float *obs, int64_t obs_stride_n, int64_t obs_stride_m,
                                  const float *share_obs, int64_t share_stride_n, int64_t share_stride_m, int32_t M, int64_t B,
                                  const float *avail, int32_t deterministic, uint64_t seed, uint64_t counter,
                                  const uint64_t *counter_dev, float *actions, float *logp, float *values, float *obs_dst,
                                  float *share_dst, const float *rewards, int64_t rew_stride_n, int64_t rew_stride_m,
                                  const uint8_t *dones, int64_t done_stride_n, int64_t done_stride_m, float *rew_dst,
                                  float *mask_dst, int32_t centralized, mappo_stream_t stream) {
  if (int rc = check_desc(actor_desc, "rollout_step")) return rc;
  if (int rc = check_desc(critic_desc, "rollout_step")) return rc;
  MAPPO_REQUIRE((actor_desc->in_dim <= MAXD) == (critic_desc->in_dim <= MAXD) && actor_desc->in_dim <= 512 && critic_desc->in_dim <= 512,
                "rollout_step: both networks narrow (in_dim <= %d) or both wide (<= 512)", MAXD);
  MAPPO_REQUIRE(actor_desc->layer_N == critic_desc->layer_N && actor_desc->use_relu == critic_desc->use_relu,
                "rollout_step: actor and critic must share layer_N and the activation");
  MAPPO_REQUIRE(critic_desc->out_dim == 1, "rollout_step: critic out_dim must be 1");
  MAPPO_REQUIRE(actor_params && critic_params && obs && share_obs && values && B > 0 && M >= 0 && (!actions == !logp),
                "rollout_step: bad arguments");                  // actions == logp == NULL: critic (+ insert) only
  MAPPO_REQUIRE(M > 0 || !obs_dst, "rollout_step: the fused insert needs the (thread, agent) row layout (M > 0)");
  MAPPO_REQUIRE(!obs_dst || (share_dst && rewards && dones && rew_dst && mask_dst && B % M == 0), "rollout_step: incomplete insert arguments");
  MAPPO_CLEAR_STICKY();
  if (actor_desc->in_dim > MAXD) {
    // Wide inputs: the insert is its own (HBM-bound: it moves the rows it copies once in, twice out) launch, the two networks share
    // one (wide_rollout_step_kernel: every CU busy for one chunk-latency chain instead of half the chip for two).
    // in_dim 256 / 512 on both networks and at most two tiles per wave: W1' staged whole, and the insert's row copies ride on the
    // forward's loads (wide_rollout_full_kernel); MAPPO_WIDE_FULL_STEP=0: the streamed form (A/B)
    const int64_t nt16_ = (B + 15) / 16;
    const bool full_step = actor_desc->in_dim == critic_desc->in_dim && (actor_desc->in_dim == 256 || actor_desc->in_dim == 512) &&
                           nt16_ <= 2 * 8 * (NUM_CU / 2) && !(getenv("MAPPO_WIDE_FULL_STEP") && atoi(getenv("MAPPO_WIDE_FULL_STEP")) == 0);
    const bool fuse_ins = full_step && obs_dst && actions && !centralized && obs_stride_m == actor_desc->in_dim &&
                          share_stride_m == critic_desc->in_dim;
    if (obs_dst && !fuse_ins)
      if (int rci = mappo_insert_mpe(obs, obs_stride_n, obs_stride_m, rewards, rew_stride_n, rew_stride_m, dones, done_stride_n, done_stride_m,
                                     obs_dst, share_dst, rew_dst, mask_dst, (int32_t)(B / M), M, actor_desc->in_dim, centralized, stream))
        return rci;
    FwdArgs a = {}, c = {};
    a.params = actor_params; a.x = obs; a.avail = avail; a.actions = actions; a.logp = logp; a.desc = *actor_desc; a.B = B;
    a.deterministic = deterministic; a.seed = seed; a.counter = counter; a.counter_dev = counter_dev;
    a.off = net_offsets(a.desc); a.map = lds_map(a.desc, 8);
    c.params = critic_params; c.x = share_obs; c.out = values; c.desc = *critic_desc; c.B = B; c.off = net_offsets(c.desc); c.map = lds_map(c.desc, 8);
    Wide16Args wa, wc;
    size_t lba, lbc;
    dim3 ga, gc, ba, bc;
    bool sa, sc;
    if (int rcp = wide_forward_prepare(a, wa, lba, ga, ba, sa, "rollout_step")) return rcp;
    if (int rcp = wide_forward_prepare(c, wc, lbc, gc, bc, sc, "rollout_step")) return rcp;
    wa.x_M = M; wa.x_sn = obs_stride_n; wa.x_sm = obs_stride_m;
    wc.x_M = M; wc.x_sn = share_stride_n; wc.x_sm = share_stride_m;
    // each network: one workgroup per 8 tiles, at most half of the chip's CUs
    const int64_t n_groups = ((B + 15) / 16 + 7) / 8;
    const int nb = (int)(n_groups < NUM_CU / 2 ? n_groups : NUM_CU / 2);
    const int nA = actions ? nb : 0;
    const dim3 grid((unsigned)(nA + nb));
    const size_t lb = lba > lbc ? lba : lbc;
    if (full_step) {
      const size_t lw = sizeof(float) * ((size_t)HID * actor_desc->in_dim + HID);      // W1' whole + folded bias; the tail's map reuses the space
      const size_t lf = lw > lb ? lw : lb;
      MAPPO_REQUIRE(lf <= 159 * 1024, "rollout_step: needs %zu B of LDS", lf);
      InsertArgs ins = {};
      if (fuse_ins) {
        wa.copy_dst = obs_dst; wc.copy_dst = share_dst;
        ins.rew = rewards; ins.rew_sn = rew_stride_n; ins.rew_sm = rew_stride_m; ins.done = dones; ins.done_sn = done_stride_n;
        ins.done_sm = done_stride_m; ins.rew_dst = rew_dst; ins.mask_dst = mask_dst; ins.N = (int)(B / M); ins.M = M;
      }
      const int rcf = actor_desc->use_relu
          ? wide16_launch_rollout_full_r<true>(actor_desc->layer_N, grid, lf, as_stream(stream), wa, a, wc, c, nA, fuse_ins ? &ins : nullptr)
          : wide16_launch_rollout_full_r<false>(actor_desc->layer_N, grid, lf, as_stream(stream), wa, a, wc, c, nA, fuse_ins ? &ins : nullptr);
      if (rcf) return rcf;
      MAPPO_CHECK_LAUNCH("rollout_step");
      return MAPPO_OK;
    }
    const int rcw = actor_desc->use_relu ? wide16_launch_rollout_step_r<true>(actor_desc->layer_N, grid, lb, as_stream(stream), wa, a, wc, c, nA)
                                         : wide16_launch_rollout_step_r<false>(actor_desc->layer_N, grid, lb, as_stream(stream), wa, a, wc, c, nA);
    if (rcw) return rcw;
    MAPPO_CHECK_LAUNCH("rollout_step");
    return MAPPO_OK;
  }
  const int64_t n_tiles = (B + 15) / 16;                 // forward16_body: 16 samples per wave
  const int want = n_tiles >= 4 ? 4 : (n_tiles >= 2 ? 2 : 1);
  int nw = fit_waves(*actor_desc, want);
  const int nwc = fit_waves(*critic_desc, want);
  nw = nw < nwc ? nw : nwc;
  StepArgs s = {};
  s.a.params = actor_params; s.a.x = obs; s.a.avail = avail; s.a.actions = actions; s.a.logp = logp; s.a.desc = *actor_desc; s.a.B = B;
  s.a.deterministic = deterministic; s.a.seed = seed; s.a.counter = counter; s.a.counter_dev = counter_dev;
  s.a.x_sn = obs_stride_n; s.a.x_sm = obs_stride_m; s.a.x_M = M;
  s.a.off = net_offsets(s.a.desc); s.a.map = lds_map(s.a.desc, nw);
  s.c.params = critic_params; s.c.x = share_obs; s.c.out = values; s.c.desc = *critic_desc; s.c.B = B;
  s.c.x_sn = share_stride_n; s.c.x_sm = share_stride_m; s.c.x_M = M;
  s.c.off = net_offsets(s.c.desc); s.c.map = lds_map(s.c.desc, nw);
  const int totA = s.a.map.total, totC = s.c.map.total;
  const size_t lds_bytes = (size_t)(totA > totC ? totA : totC) * sizeof(float);
  MAPPO_REQUIRE(lds_bytes <= LDS_DYN_MAX, "rollout_step: needs %zu B of LDS", lds_bytes);
  int64_t nb = (n_tiles + nw - 1) / nw;
  if (nb > NUM_CU / 2) nb = NUM_CU / 2;
  s.nA = actions ? (int)nb : 0; s.nC = (int)nb; s.nI = 0;
  if (obs_dst) {
    InsertArgs &i = s.ins;
    i.obs = obs; i.obs_sn = obs_stride_n; i.obs_sm = obs_stride_m; i.rew = rewards; i.rew_sn = rew_stride_n; i.rew_sm = rew_stride_m;
    i.done = dones; i.done_sn = done_stride_n; i.done_sm = done_stride_m; i.obs_dst = obs_dst; i.share_dst = share_dst;
    i.rew_dst = rew_dst; i.mask_dst = mask_dst; i.N = (int)(B / M); i.M = M; i.D = actor_desc->in_dim; i.centralized = centralized;
    const int64_t total = B * (centralized ? (int64_t)M * i.D : i.D);
    int64_t ni = (total + 2047) / 2048;                  // ~8 elements per thread
    s.nI = (int)(ni > NUM_CU ? NUM_CU : ni);              // (64 insert workgroups became the long pole of the launch beyond ~2 000 threads)
  }
  dim3 grid((unsigned)(s.nA + s.nC + s.nI)), block(WAVE * nw);
  const bool relu = actor_desc->use_relu != 0;
  int rc;
  switch (actor_desc->layer_N) {
    case 0: rc = relu ? step_launch<true, 0>(grid, block, lds_bytes, as_stream(stream), s) : step_launch<false, 0>(grid, block, lds_bytes, as_stream(stream), s); break;
    case 1: rc = relu ? step_launch<true, 1>(grid, block, lds_bytes, as_stream(stream), s) : step_launch<false, 1>(grid, block, lds_bytes, as_stream(stream), s); break;
    default: rc = relu ? step_launch<true, 2>(grid, block, lds_bytes, as_stream(stream), s) : step_launch<false, 2>(grid, block, lds_bytes, as_stream(stream), s); break;
  }
  if (rc) return rc;
  MAPPO_CHECK_LAUNCH("rollout_step");
  return MAPPO_OK;
}
#endif

#ifdef MLP_TU_MAIN
template <int MODE>
static int launch_forward(const FwdArgs &a_in, hipStream_t st, const char *who) {
  MAPPO_CLEAR_STICKY();
  const int64_t n_tiles = (a_in.B + TS - 1) / TS;
  const int LN = a_in.desc.layer_N;
  // all waves of a workgroup stage the weights together, so 4 waves per workgroup even for rollout-sized batches
  const int nw = fit_waves(a_in.desc, n_tiles >= 4 ? 4 : (n_tiles >= 2 ? 2 : 1));
  FwdArgs a = a_in;
  a.off = net_offsets(a.desc);
  a.map = lds_map(a.desc, nw);
  const size_t lds_bytes = (size_t)a.map.total * sizeof(float);
  MAPPO_REQUIRE(lds_bytes <= LDS_DYN_MAX, "%s: needs %zu B of LDS", who, lds_bytes);
  int64_t nb = (n_tiles + nw - 1) / nw;
  if (nb > NUM_CU) nb = NUM_CU;
  dim3 grid((unsigned)nb), block(WAVE * nw);
  {
    if (a.desc.in_dim > MAXD && a.desc.in_dim <= 512 && a.x_M == 0) {
      // wide inputs: layer 1 from registers + double-buffered W1 chunks, the rest of the network on the same tile (mlp_wide16.h)
      Wide16Args w;
      size_t lb;
      dim3 g2, b2;
      bool small;
      if (int rcp = wide_forward_prepare(a, w, lb, g2, b2, small, who)) return rcp;
      const int64_t nt16 = (a.B + 15) / 16;
      int64_t sk_max = WIDE_SK_MAX_TILES;
      if (const char *e = getenv("MAPPO_WIDE_SK_TILES")) sk_max = atoll(e);      // diagnostic override
      if (nt16 <= sk_max && !getenv("MAPPO_WIDE_NO_SK")) {          // step-sized batch: one tile per 4-wave workgroup, split-K
        const int64_t gsk = nt16 < NUM_CU ? nt16 : NUM_CU;          // (512 registers per wave: one workgroup per CU; more tiles are walked)
        if (int rcw = wide16_launch_forward_sk(MODE, a.desc.use_relu != 0, LN, dim3((unsigned)gsk), lb, st, w, a, who)) return rcw;
        MAPPO_CHECK_LAUNCH(who);
        return MAPPO_OK;
      }
      // trunk features of a training-sized batch: W1' resident (wide_features16_resident_kernel); MAPPO_WIDE_RESIDENT=0: streamed (A/B)
      FwdArgs ar = a;
      ar.map = lds_map_tail(a.desc);
      const size_t lres = sizeof(float) * ((size_t)HID * 64 * ((a.desc.in_dim + 63) / 64) + HID + ar.map.tiles);
      if (MODE == 2 && LN <= 1 && nt16 >= 2 * 8 * NUM_CU && lres <= 159 * 1024 &&
          !(getenv("MAPPO_WIDE_RESIDENT") && atoi(getenv("MAPPO_WIDE_RESIDENT")) == 0)) {
        const int rcr = a.desc.use_relu ? wide16_launch_features_resident_r<true>(LN, g2, st, w, ar) : wide16_launch_features_resident_r<false>(LN, g2, st, w, ar);
        if (rcr) return rcr;
        MAPPO_CHECK_LAUNCH(who);
        return MAPPO_OK;
      }
      if (int rcw = wide16_launch_forward(MODE, a.desc.use_relu != 0, LN, small, g2, b2, lb, st, w, a, who)) return rcw;
      MAPPO_CHECK_LAUNCH(who);
      return MAPPO_OK;
    }
  }
#define FWD2(R, L, W)                                                                                          \
  do {                                                                                                         \
    static const hipError_t e_ = hipFuncSetAttribute((const void *)mlp_forward_kernel<R, L, MODE, W>,                      \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_DYN_MAX);       \
    if (e_ != hipSuccess) { mappo_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; } \
    PROF_LAUNCH(prof_id, (mlp_forward_kernel<R, L, MODE, W>), grid, block, lds_bytes, st, a);                   \
  } while (0)
#define FWD_W(R, L) do { if (xw == 2) FWD2(R, L, 2); else if (xw == 1) FWD2(R, L, 1); else FWD2(R, L, 0); } while (0)
  const bool relu = a.desc.use_relu != 0;
  const int xw = a.desc.in_dim > MAXD ? 2 : (a.desc.in_dim > 32 ? 1 : 0);
  const int prof_id = (MODE == 1) ? MAPPO_PROF_ACT : MAPPO_PROF_MLP_FWD;
  if (LN == 0) { if (relu) FWD_W(true, 0); else FWD_W(false, 0); }
  else if (LN == 1) { if (relu) FWD_W(true, 1); else FWD_W(false, 1); }
  else { if (relu) FWD_W(true, 2); else FWD_W(false, 2); }
#undef FWD_W
#undef FWD2
  MAPPO_CHECK_LAUNCH(who);
  return MAPPO_OK;
}

extern "C" int mappo_mlp_forward(const float *params, const mappo_net_desc *desc, const float *x, const int32_t *rows,
                                 int64_t B, float *out, mappo_stream_t stream) {
  if (int rc = check_desc(desc, "mlp_forward")) return rc;
  MAPPO_REQUIRE(params && x && out && B > 0, "mlp_forward: bad arguments");
  FwdArgs a = {};
  a.params = params; a.x = x; a.rows = rows; a.out = out; a.desc = *desc; a.B = B;
  return launch_forward<0>(a, as_stream(stream), "mlp_forward");
}

static int check_desc_trunk(const mappo_net_desc *d, const char *who);

extern "C" int mappo_mlp_features(const float *params, const mappo_net_desc *desc, const float *x, const int32_t *rows,
                                  int64_t B, float *featT, mappo_stream_t stream) {
  if (int rc = check_desc_trunk(desc, "mlp_features")) return rc;
  MAPPO_REQUIRE(params && x && featT && B > 0, "mlp_features: bad arguments");
  FwdArgs a = {};
  a.params = params; a.x = x; a.rows = rows; a.out = featT; a.desc = *desc; a.B = B;
  if (desc->in_dim <= MAXD) return launch_features16(a, as_stream(stream));
  return launch_forward<2>(a, as_stream(stream), "mlp_features");
}

// wide-input branch of mappo_mlp_features_seq (gru_train16.hip): the one-launch wide forward with the blocked output form
int mlp_features_blocked_wide_(const float *params, const mappo_net_desc *desc, const float *x, const int32_t *rows, int64_t B,
                               float *out_blocked, mappo_stream_t stream) {
  if (int rc = check_desc_trunk(desc, "mlp_features_seq")) return rc;
  MAPPO_REQUIRE(desc->in_dim > MAXD && desc->in_dim <= 512 && (B & 15) == 0, "mlp_features_seq: wide inputs need in_dim <= 512 and Nc %% 16 == 0");
  FwdArgs a = {};
  a.params = params; a.x = x; a.rows = rows; a.out = out_blocked; a.desc = *desc; a.B = B; a.out_blocked = 1;
  return launch_forward<2>(a, as_stream(stream), "mlp_features_seq");
}

extern "C" int mappo_actor_act(const float *params, const mappo_net_desc *desc, const float *obs, const float *avail,
                               int64_t B, int32_t deterministic, uint64_t seed, uint64_t counter,
                               const uint64_t *counter_dev, float *actions, float *logp, mappo_stream_t stream) {
  if (int rc = check_desc(desc, "actor_act")) return rc;
  MAPPO_REQUIRE(params && obs && actions && logp && B > 0, "actor_act: bad arguments");
  FwdArgs a = {};
  a.params = params; a.x = obs; a.rows = nullptr; a.avail = avail; a.actions = actions; a.logp = logp; a.desc = *desc;
  a.B = B; a.deterministic = deterministic; a.seed = seed; a.counter = counter; a.counter_dev = counter_dev;
  return launch_forward<1>(a, as_stream(stream), "actor_act");
}

// ---- one wave per 16-sample tile (mlp_upd16.h) ---------------------------------------------------------------------
#define UPD16_WAVES (UPD16_THREADS / WAVE)
static bool upd16_eligible(const mappo_net_desc &d, bool actor) {
  return d.in_dim <= MAXD && d.layer_N <= 1 && !d.recurrent && (actor ? d.out_dim <= 16 : d.out_dim == 1);
}
// MFMA instructions per 16-sample tile (+ a flat allowance for the VALU phases): the share of the chip a network gets
static int upd16_tile_cost(const mappo_net_desc &d, bool actor) {
  const int C = (d.in_dim + 3) >> 2, nbk = d.in_dim > 32 ? 4 : 2;
  int c = 4 * C + 16 * nbk + 40;
  if (d.layer_N > 0) c += 3 * 64;
  if (actor) c += 16 + 16 + 4 * ((d.out_dim + 3) >> 2);
  return c;
}
// trunk backward (gradient arriving at the trunk output; recurrent networks): same kernels, no head
static bool upd16_trunk_eligible(const mappo_net_desc &d) {
  return d.in_dim <= 512 && d.layer_N <= 1;
}
static size_t upd16_trunk_lds_floats(const mappo_net_desc &d) {
  if (d.in_dim > MAXD) return d.layer_N > 0 ? L16<1, 3, true, true>::TOTAL : L16<0, 3, true, true>::TOTAL;
  const bool w = d.in_dim > 32;
  if (d.layer_N > 0) return w ? L16<1, 3, true>::TOTAL : L16<1, 3, false>::TOTAL;
  return w ? L16<0, 3, true>::TOTAL : L16<0, 3, false>::TOTAL;
}
static size_t upd16_lds_floats(const mappo_net_desc &d, bool actor) {
  const bool w = d.in_dim > 32;
  if (actor) {
    if (d.layer_N > 0) return w ? L16<1, 1, true>::TOTAL : L16<1, 1, false>::TOTAL;
    return w ? L16<0, 1, true>::TOTAL : L16<0, 1, false>::TOTAL;
  }
  if (d.layer_N > 0) return w ? L16<1, 2, true>::TOTAL : L16<1, 2, false>::TOTAL;
  return w ? L16<0, 2, true>::TOTAL : L16<0, 2, false>::TOTAL;
}
static int prep16(Upd16Args &a, bool actor, const char *who) {
  UpdArgs &u = a.u;
  u.off = net_offsets(u.desc);
  MAPPO_REQUIRE(u.slab_col0 >= 0 && u.slab_col0 + u.off.total <= u.slab_stride, "%s: slab column range", who);
  MAPPO_REQUIRE(upd16_lds_floats(u.desc, actor) * sizeof(float) <= UPD16_LDS_MAX, "%s: needs %zu B of LDS", who,
                upd16_lds_floats(u.desc, actor) * sizeof(float));
  a.zero_row0 = a.zero_row1 = 0; a.zero_col0 = 0; a.zero_cols = 0; a.zero_partials = nullptr;
  return MAPPO_OK;
}
// ---- wide inputs through the 16-sample-tile kernels ----
static bool upd16x_eligible(const mappo_net_desc &d, bool actor) {
  return d.in_dim > MAXD && d.in_dim <= 512 && d.layer_N <= 1 && !d.recurrent && (actor ? d.out_dim <= 16 : d.out_dim == 1);
}
static int64_t wide_z1_offset(int64_t B) { return wide16_z1_offset(B); }     // workspace layout: mlp_wide16.h (>= the [64][B] | mean0 | rstd0 of the round-1 kernels)

// The layout a producer leaves in a wide workspace is a pure function of (network descriptor, producer): mappo_wide_layout.
// The caller passes it on to mappo_wide_l1_backward — no host-side state, nothing keyed by pointers.
static int wide_layout_of(const mappo_net_desc &d, int producer) {
  if (producer == MAPPO_PRODUCER_TRUNK_BACKWARD) return upd16_trunk_eligible(d) ? MAPPO_WIDE_LAYOUT_BLOCKED : MAPPO_WIDE_LAYOUT_FEATURE_MAJOR;
  if (producer == MAPPO_PRODUCER_ACTOR_UPDATE || producer == MAPPO_PRODUCER_CRITIC_UPDATE)
    return upd16x_eligible(d, producer == MAPPO_PRODUCER_ACTOR_UPDATE) ? MAPPO_WIDE_LAYOUT_BLOCKED : MAPPO_WIDE_LAYOUT_FEATURE_MAJOR;
  return MAPPO_WIDE_LAYOUT_FEATURE_MAJOR;                        // mappo_mlp_backward (external gradient): K-chunked kernel
}
extern "C" int32_t mappo_wide_layout(const mappo_net_desc *desc, int32_t producer) {
  if (!desc || producer < MAPPO_PRODUCER_MLP_BACKWARD || producer > MAPPO_PRODUCER_TRUNK_BACKWARD) {
    mappo_set_error("wide_layout: bad arguments");
    return MAPPO_EINVAL;
  }
  return wide_layout_of(*desc, producer);
}
static size_t upd16x_lds_floats(const mappo_net_desc &d, bool actor) {
  if (actor) return d.layer_N > 0 ? L16<1, 1, true, true>::TOTAL : L16<0, 1, true, true>::TOTAL;
  return d.layer_N > 0 ? L16<1, 2, true, true>::TOTAL : L16<0, 2, true, true>::TOTAL;
}
static int prep16x(Upd16Args &a, bool actor, const char *who) {
  UpdArgs &u = a.u;
  u.off = net_offsets(u.desc);
  MAPPO_REQUIRE(u.slab_col0 >= 0 && u.slab_col0 + u.off.total <= u.slab_stride, "%s: slab column range", who);
  MAPPO_REQUIRE(upd16x_lds_floats(u.desc, actor) * sizeof(float) <= UPD16_LDS_MAX, "%s: needs %zu B of LDS", who,
                upd16x_lds_floats(u.desc, actor) * sizeof(float));
  a.zero_row0 = a.zero_row1 = 0; a.zero_col0 = 0; a.zero_cols = 0; a.zero_partials = nullptr;
  return MAPPO_OK;
}
// z1 = b1' + W1' xhat0 and the row statistics, one launch (mlp_wide16.h)
static int launch_wide_l1_fwd(const float *params, const mappo_net_desc &d, const NetOff &o, const float *x, const int32_t *rows, int64_t B,
                              float *z1, float *mean0, float *rstd0, hipStream_t st, const char *who) {
  Wide16Args w = {};
  w.params = params; w.x = x; w.rows = rows; w.z1 = z1; w.mean0 = mean0; w.rstd0 = rstd0; w.B = B; w.D = d.in_dim;
  w.w1 = o.w1; w.b1 = o.b1; w.fn_w = d.use_feature_norm ? o.fn_w : -1; w.fn_b = d.use_feature_norm ? o.fn_b : -1;
  const int64_t n_groups = ((B + 15) / 16 + 7) / 8;
  dim3 grid((unsigned)(n_groups < NUM_CU ? n_groups : NUM_CU));
  if (int rcl = wide16_launch_l1_fwd(w, grid, st)) return rcl;
  MAPPO_CHECK_LAUNCH(who);
  return MAPPO_OK;
}

// workgroups of the dual launch: the chip's 256 CUs split by the networks' tile costs (few tiles: one tile per wave)
static void upd16_split(const mappo_net_desc &da, const mappo_net_desc &dc, int64_t B, int &nA, int &nC) {
  const int64_t n_tiles = (B + 15) / 16;
  const int64_t want = (n_tiles + UPD16_WAVES - 1) / UPD16_WAVES;
  const int ca = upd16_tile_cost(da, true), cc = upd16_tile_cost(dc, false);
  int a = (int)((int64_t)NUM_CU * ca / (ca + cc));
  if (const char *e = getenv("MAPPO_UPD16_NA")) a = atoi(e);   // diagnostic override of the actor's share (scripts/time_dual.py)
  a = a < 64 ? 64 : (a > NUM_CU - 64 ? NUM_CU - 64 : a);
  int c = NUM_CU - a;
  nA = (int)(want < a ? want : a);
  nC = (int)(want < c ? want : c);
}

extern "C" int32_t mappo_mlp_backward_slabs(int64_t B) {
  // number of slabs an update/backward launch writes: one per workgroup, at most one workgroup per CU
  int64_t n_tiles = (B + TS - 1) / TS;
  return (int32_t)(n_tiles < NUM_CU ? n_tiles : NUM_CU);
}

template <int HEAD>
static int launch_update(UpdArgs &a, hipStream_t st, const char *who) {
  MAPPO_CLEAR_STICKY();
  const mappo_net_desc &d = a.desc;
  a.off = net_offsets(d);
  MAPPO_REQUIRE(a.slab_col0 >= 0 && a.slab_col0 + a.off.total <= a.slab_stride, "%s: slab column range", who);
  const int LN = d.layer_N;
  const bool relu = d.use_relu != 0;
  a.p_red = (HEAD == 3 && d.recurrent) ? a.off.gru_wih : a.off.total;
  int nb = mappo_mlp_backward_slabs(a.B);          // every slab the caller sized for is written: grid == that count
  if (a.n_blocks > 0) {                            // caller-chosen grid (actor and critic side by side on disjoint CUs)
    MAPPO_REQUIRE(a.n_blocks <= NUM_CU, "%s: n_blocks %d > %d", who, a.n_blocks, NUM_CU);
    nb = a.n_blocks < nb ? a.n_blocks : nb;
  }
#ifdef MLP_STAMPS
  a.stamps = g_stamp_host;
#endif
  int rc;
  const bool trunk16 = HEAD == 3 && upd16_trunk_eligible(d);
  if (((HEAD == 1 || HEAD == 2) && upd16_eligible(d, HEAD == 1)) || (trunk16 && d.in_dim <= MAXD)) {
    // one wave per 16-sample tile (mlp_upd16.h)
    Upd16Args a16 = {};
    a16.u = a;
    if (HEAD == 3) {
      a16.u.off = a.off;
      a16.zero_row0 = a16.zero_row1 = 0; a16.zero_col0 = 0; a16.zero_cols = 0; a16.zero_partials = nullptr;
      MAPPO_REQUIRE(upd16_trunk_lds_floats(d) * sizeof(float) <= UPD16_LDS_MAX, "%s: needs %zu B of LDS", who, upd16_trunk_lds_floats(d) * sizeof(float));
    } else if (int rc16 = prep16(a16, HEAD == 1, who)) return rc16;
    const size_t lds_bytes = (HEAD == 3 ? upd16_trunk_lds_floats(d) : upd16_lds_floats(d, HEAD == 1)) * sizeof(float);
    dim3 grid((unsigned)nb), block(WAVE * UPD16_WAVES);
    const bool wide = d.in_dim > 32;
    if (LN == 0) rc = relu ? upd16_inst<true, 0>(HEAD, wide, grid, block, lds_bytes, st, a16) : upd16_inst<false, 0>(HEAD, wide, grid, block, lds_bytes, st, a16);
    else rc = relu ? upd16_inst<true, 1>(HEAD, wide, grid, block, lds_bytes, st, a16) : upd16_inst<false, 1>(HEAD, wide, grid, block, lds_bytes, st, a16);
  } else if (((HEAD == 1 || HEAD == 2) && upd16x_eligible(d, HEAD == 1)) || (trunk16 && d.in_dim > MAXD)) {
    // wide inputs: layer-1 forward as its own kernel (mlp_wide16.h), then the 16-sample-tile update kernel from z1 on; the
    // caller's mappo_wide_l1_backward turns dz1 + the row statistics into the W1 / feature-norm gradients
    MAPPO_REQUIRE(a.wide_ws, "%s: in_dim %d needs the wide workspace (mappo_wide_workspace_floats)", who, d.in_dim);
    if (int rcw = launch_wide_l1_fwd(a.params, d, a.off, a.x, a.rows, a.B, a.wide_ws + wide_z1_offset(a.B), a.wide_ws + 64 * wide16_bp(a.B),
                                     a.wide_ws + 65 * wide16_bp(a.B), st, who))
      return rcw;
    Upd16Args a16 = {};
    a16.u = a;
    if (HEAD == 3) {
      a16.zero_row0 = a16.zero_row1 = 0; a16.zero_col0 = 0; a16.zero_cols = 0; a16.zero_partials = nullptr;
      MAPPO_REQUIRE(upd16_trunk_lds_floats(d) * sizeof(float) <= UPD16_LDS_MAX, "%s: needs %zu B of LDS", who, upd16_trunk_lds_floats(d) * sizeof(float));
    } else if (int rc16 = prep16x(a16, HEAD == 1, who)) return rc16;
    const size_t lds_bytes = (HEAD == 3 ? upd16_trunk_lds_floats(d) : upd16x_lds_floats(d, HEAD == 1)) * sizeof(float);
    dim3 grid((unsigned)nb), block(WAVE * UPD16_WAVES);
    if (LN == 0) rc = relu ? upd16x_inst<true, 0>(HEAD, grid, block, lds_bytes, st, a16) : upd16x_inst<false, 0>(HEAD, grid, block, lds_bytes, st, a16);
    else rc = relu ? upd16x_inst<true, 1>(HEAD, grid, block, lds_bytes, st, a16) : upd16x_inst<false, 1>(HEAD, grid, block, lds_bytes, st, a16);
  } else if (d.in_dim <= MAXD) {
    // pair kernel (mlp_upd2.h): n_pairs tiles in flight per workgroup, two waves each
    const int np = fit_waves(d, 4);
    a.map = lds_map(d, np);
    const size_t lds_bytes = (size_t)a.map.total * sizeof(float);
    MAPPO_REQUIRE(lds_bytes <= LDS_DYN_MAX, "%s: needs %zu B of LDS", who, lds_bytes);
    a.red_base = 0;
    const int tile_area = np * a.map.wave_stride, vec_floats = 2 * np * (3 * (LN + 1) + 3) * 64;
    a.n_regions = (np > 1 && 2 * a.p_red + vec_floats <= tile_area) ? 2 : 1;
    MAPPO_REQUIRE(a.n_regions * a.p_red + vec_floats <= tile_area, "%s: reduction buffer too small", who);
    dim3 grid((unsigned)nb), block(2 * WAVE * np);
    const bool wide = d.in_dim > 32;
    if (LN == 0) rc = relu ? upd2_inst<true, 0, HEAD>(wide, grid, block, lds_bytes, st, a, who) : upd2_inst<false, 0, HEAD>(wide, grid, block, lds_bytes, st, a, who);
    else if (LN == 1) rc = relu ? upd2_inst<true, 1, HEAD>(wide, grid, block, lds_bytes, st, a, who) : upd2_inst<false, 1, HEAD>(wide, grid, block, lds_bytes, st, a, who);
    else rc = relu ? upd2_inst<true, 2, HEAD>(wide, grid, block, lds_bytes, st, a, who) : upd2_inst<false, 2, HEAD>(wide, grid, block, lds_bytes, st, a, who);
  } else {
    // wide inputs: one wave per tile, layer 1 K-chunked; W1 / feature-norm gradients come from wide_l1_bwd_kernel
    const int nw = fit_waves(d, UPD_THREADS / WAVE);
    a.map = lds_map(d, nw);
    const size_t lds_bytes = (size_t)a.map.total * sizeof(float);
    MAPPO_REQUIRE(lds_bytes <= LDS_DYN_MAX, "%s: needs %zu B of LDS", who, lds_bytes);
    a.red_base = a.off.b1;
    MAPPO_REQUIRE(a.wide_ws, "%s: in_dim %d needs the wide workspace (mappo_wide_workspace_floats)", who, d.in_dim);
    const int p_span = a.p_red - a.red_base;
    a.n_regions = (nw > 1 && nw * a.map.wave_stride >= 2 * p_span) ? 2 : 1;
    MAPPO_REQUIRE(nw * a.map.wave_stride >= a.n_regions * p_span, "%s: reduction buffer too small", who);
    dim3 grid((unsigned)nb), block(WAVE * nw);
    if (LN == 0) rc = relu ? upd_inst<true, 0, HEAD>(2, grid, block, lds_bytes, st, a, who) : upd_inst<false, 0, HEAD>(2, grid, block, lds_bytes, st, a, who);
    else if (LN == 1) rc = relu ? upd_inst<true, 1, HEAD>(2, grid, block, lds_bytes, st, a, who) : upd_inst<false, 1, HEAD>(2, grid, block, lds_bytes, st, a, who);
    else rc = relu ? upd_inst<true, 2, HEAD>(2, grid, block, lds_bytes, st, a, who) : upd_inst<false, 2, HEAD>(2, grid, block, lds_bytes, st, a, who);
  }
  if (rc) return rc;
  MAPPO_CHECK_LAUNCH(who);
  return MAPPO_OK;
}

extern "C" int mappo_mlp_backward(const float *params, const mappo_net_desc *desc, const float *x, const int32_t *rows,
                                  int64_t B, const float *dout, float *slabs, int64_t slab_stride, int64_t slab_col0,
                                  float *wide_ws, mappo_stream_t stream) {
  if (int rc = check_desc(desc, "mlp_backward")) return rc;
  MAPPO_REQUIRE(params && x && dout && slabs && B > 0, "mlp_backward: bad arguments");
  UpdArgs a = {};
  a.params = params; a.x = x; a.rows = rows; a.dout = dout; a.slabs = slabs; a.slab_stride = slab_stride;
  a.slab_col0 = slab_col0; a.desc = *desc; a.B = B; a.wide_ws = wide_ws;
  return launch_update<0>(a, as_stream(stream), "mlp_backward");
}

extern "C" int mappo_trunk_backward(const float *params, const mappo_net_desc *desc, const float *x, const int32_t *rows,
                                    int64_t B, const float *dHT, float *slabs, int64_t slab_stride, int64_t slab_col0,
                                    float *wide_ws, mappo_stream_t stream) {
  if (int rc = check_desc_trunk(desc, "trunk_backward")) return rc;
  MAPPO_REQUIRE(params && x && dHT && slabs && B > 0, "trunk_backward: bad arguments");
  UpdArgs a = {};
  a.params = params; a.x = x; a.rows = rows; a.dHT = dHT; a.slabs = slabs; a.slab_stride = slab_stride;
  a.slab_col0 = slab_col0; a.desc = *desc; a.B = B; a.wide_ws = wide_ws;
  return launch_update<3>(a, as_stream(stream), "trunk_backward");
}

// the same for the sequence-tiled minibatch of the recurrent training pass: d(trunk output) arrives BLOCKED per (t, 16 sequences)
// tile (the d x component gru16_bwd_kernel leaves in its scratch); in_dim <= 64
extern "C" int mappo_trunk_backward_seq(const float *params, const mappo_net_desc *desc, const float *x, const int32_t *rows,
                                        int32_t L, int32_t Nc, const float *dx_blocked, float *slabs, int64_t slab_stride,
                                        int64_t slab_col0, float *wide_ws, mappo_stream_t stream) {
  if (int rc = check_desc_trunk(desc, "trunk_backward_seq")) return rc;
  MAPPO_REQUIRE(params && x && dx_blocked && slabs && L > 0 && Nc > 0, "trunk_backward_seq: bad arguments");
  MAPPO_REQUIRE(desc->in_dim <= 512 && desc->layer_N <= 1, "trunk_backward_seq: in_dim %d / layer_N %d take mappo_trunk_backward", desc->in_dim, desc->layer_N);
  MAPPO_REQUIRE(desc->in_dim <= MAXD || (Nc & 15) == 0, "trunk_backward_seq: wide inputs need Nc %% 16 == 0 (Nc = %d)", Nc);
  UpdArgs a = {};
  a.params = params; a.x = x; a.rows = rows; a.dHT = dx_blocked; a.slabs = slabs; a.slab_stride = slab_stride;
  a.slab_col0 = slab_col0; a.desc = *desc; a.B = (int64_t)L * Nc; a.seq_nc = Nc; a.wide_ws = wide_ws;
  return launch_update<3>(a, as_stream(stream), "trunk_backward_seq");
}

extern "C" int64_t mappo_update_partials_bytes(void) { return (int64_t)NUM_CU * 4 * sizeof(double); }

extern "C" int mappo_actor_update(const float *params, const mappo_net_desc *desc, const float *obs, const int32_t *rows,
                                  int64_t B, const float *avail, const float *actions, const float *old_logp,
                                  const float *adv, const float *active, const double *mb_moments,
                                  const mappo_ppo_cfg *cfg, float *slabs, int64_t slab_stride, int64_t slab_col0,
                                  double *partials, float *wide_ws, int32_t n_blocks, mappo_stream_t stream) {
  if (int rc = check_desc(desc, "actor_update")) return rc;
  MAPPO_REQUIRE(params && obs && actions && old_logp && adv && active && mb_moments && cfg && slabs && partials && B > 0,
                "actor_update: bad arguments");
  UpdArgs a = {};
  a.params = params; a.x = obs; a.rows = rows; a.slabs = slabs; a.slab_stride = slab_stride; a.slab_col0 = slab_col0;
  a.desc = *desc; a.B = B; a.avail = avail; a.actions = actions; a.old_logp = old_logp; a.adv = adv; a.active = active;
  a.mb_moments = mb_moments; a.partials = partials; a.cfg = *cfg; a.wide_ws = wide_ws; a.n_blocks = n_blocks;
  return launch_update<1>(a, as_stream(stream), "actor_update");
}

extern "C" int mappo_critic_update(const float *params, const mappo_net_desc *desc, const float *share_obs,
                                   const int32_t *rows, int64_t B, const float *v_old, const float *returns,
                                   const float *active, const float *vn_state, const double *mb_moments,
                                   const mappo_ppo_cfg *cfg, float *slabs, int64_t slab_stride, int64_t slab_col0,
                                   double *partials, float *wide_ws, int32_t n_blocks, mappo_stream_t stream) {
  if (int rc = check_desc(desc, "critic_update")) return rc;
  MAPPO_REQUIRE(desc->out_dim == 1, "critic_update: out_dim must be 1");
  MAPPO_REQUIRE(params && share_obs && v_old && returns && active && mb_moments && cfg && slabs && partials && B > 0,
                "critic_update: bad arguments");
  MAPPO_REQUIRE(!cfg->use_valuenorm || vn_state, "critic_update: use_valuenorm needs vn_state");
  UpdArgs a = {};
  a.params = params; a.x = share_obs; a.rows = rows; a.slabs = slabs; a.slab_stride = slab_stride; a.slab_col0 = slab_col0;
  a.desc = *desc; a.B = B; a.v_old = v_old; a.returns = returns; a.active = active; a.vn_state = vn_state;
  a.mb_moments = mb_moments; a.partials = partials; a.cfg = *cfg; a.wide_ws = wide_ws; a.n_blocks = n_blocks;
  return launch_update<2>(a, as_stream(stream), "critic_update");
}

// ---- actor + critic update in one launch --------------------------------------------------------------------------
static int prep_pair(UpdArgs &a, int np, const char *who) {
  a.off = net_offsets(a.desc);
  MAPPO_REQUIRE(a.slab_col0 >= 0 && a.slab_col0 + a.off.total <= a.slab_stride, "%s: slab column range", who);
  a.map = lds_map(a.desc, np);
  a.p_red = a.off.total;
  a.red_base = 0;
  const int LN = a.desc.layer_N;
  const int tile_area = np * a.map.wave_stride, vec_floats = 2 * np * (3 * (LN + 1) + 3) * 64;
  a.n_regions = (np > 1 && 2 * a.p_red + vec_floats <= tile_area) ? 2 : 1;
  MAPPO_REQUIRE(a.n_regions * a.p_red + vec_floats <= tile_area, "%s: reduction buffer too small", who);
  return MAPPO_OK;
}

extern "C" int32_t mappo_dual_update_slabs(const mappo_net_desc *actor_desc, const mappo_net_desc *critic_desc, int64_t B) {
  // slab rows (= loss-partial rows) the caller provides PER NETWORK for mappo_actor_critic_update; every one of them is written
  if (actor_desc && critic_desc && upd16_eligible(*actor_desc, true) && upd16_eligible(*critic_desc, false)) {
    int nA, nC;
    upd16_split(*actor_desc, *critic_desc, B, nA, nC);
    return nA > nC ? nA : nC;
  }
  int64_t n_tiles = (B + TS - 1) / TS;              // pair kernel: half the CUs each
  return (int32_t)(n_tiles < NUM_CU / 2 ? n_tiles : NUM_CU / 2);
}

extern "C" int mappo_actor_critic_update(const float *actor_params, const mappo_net_desc *actor_desc, const float *obs,
                                         const float *critic_params, const mappo_net_desc *critic_desc, const float *share_obs,
                                         const int32_t *rows, int64_t B, const float *avail, const float *actions,
                                         const float *old_logp, const float *adv, const float *active, const float *v_old,
                                         const float *returns, const float *vn_state, const double *mb_moments,
                                         const mappo_ppo_cfg *cfg, float *slabs, int64_t slab_stride, int64_t actor_col0,
                                         int64_t critic_col0, double *actor_partials, double *critic_partials,
                                         mappo_stream_t stream) {
  if (int rc = check_desc(actor_desc, "actor_critic_update")) return rc;
  if (int rc = check_desc(critic_desc, "actor_critic_update")) return rc;
  MAPPO_REQUIRE(actor_desc->in_dim <= MAXD && critic_desc->in_dim <= MAXD, "actor_critic_update: in_dim > %d takes the separate launches", MAXD);
  MAPPO_REQUIRE(actor_desc->layer_N == critic_desc->layer_N && actor_desc->use_relu == critic_desc->use_relu,
                "actor_critic_update: actor and critic must share layer_N and the activation");
  MAPPO_REQUIRE(critic_desc->out_dim == 1, "actor_critic_update: critic out_dim must be 1");
  MAPPO_REQUIRE(actor_params && critic_params && obs && share_obs && actions && old_logp && adv && active && v_old && returns && mb_moments &&
                    cfg && slabs && actor_partials && critic_partials && B > 0, "actor_critic_update: bad arguments");
  MAPPO_REQUIRE(!cfg->use_valuenorm || vn_state, "actor_critic_update: use_valuenorm needs vn_state");
  MAPPO_CLEAR_STICKY();
  if (upd16_eligible(*actor_desc, true) && upd16_eligible(*critic_desc, false)) {
    Dual16Args d = {};
    UpdArgs &a = d.a.u, &c = d.c.u;
    a.params = actor_params; a.x = obs; a.rows = rows; a.slabs = slabs; a.slab_stride = slab_stride; a.slab_col0 = actor_col0;
    a.desc = *actor_desc; a.B = B; a.avail = avail; a.actions = actions; a.old_logp = old_logp; a.adv = adv; a.active = active;
    a.mb_moments = mb_moments; a.partials = actor_partials; a.cfg = *cfg;
    c.params = critic_params; c.x = share_obs; c.rows = rows; c.slabs = slabs; c.slab_stride = slab_stride; c.slab_col0 = critic_col0;
    c.desc = *critic_desc; c.B = B; c.v_old = v_old; c.returns = returns; c.active = active; c.vn_state = vn_state;
    c.mb_moments = mb_moments; c.partials = critic_partials; c.cfg = *cfg;
    if (int rc = prep16(d.a, true, "actor_critic_update")) return rc;
    if (int rc = prep16(d.c, false, "actor_critic_update")) return rc;
    upd16_split(a.desc, c.desc, B, d.nA, d.nC);
    // the network with fewer workgroups: its missing slab / partial rows are zero-filled by the other one's workgroups
    if (d.nA < d.nC) { d.c.zero_row0 = d.nA; d.c.zero_row1 = d.nC; d.c.zero_col0 = actor_col0; d.c.zero_cols = a.off.total; d.c.zero_partials = actor_partials; }
    if (d.nC < d.nA) { d.a.zero_row0 = d.nC; d.a.zero_row1 = d.nA; d.a.zero_col0 = critic_col0; d.a.zero_cols = c.off.total; d.a.zero_partials = critic_partials; }
    const size_t la = upd16_lds_floats(a.desc, true), lc = upd16_lds_floats(c.desc, false);
    const size_t lds_bytes = (la > lc ? la : lc) * sizeof(float);
    dim3 grid((unsigned)(d.nA + d.nC)), block(WAVE * UPD16_WAVES);
    const bool wa = a.desc.in_dim > 32, wc = c.desc.in_dim > 32, relu = a.desc.use_relu != 0;
    int rc;
    if (a.desc.layer_N == 0) rc = relu ? upd16d_inst<true, 0>(wa, wc, grid, block, lds_bytes, as_stream(stream), d) : upd16d_inst<false, 0>(wa, wc, grid, block, lds_bytes, as_stream(stream), d);
    else rc = relu ? upd16d_inst<true, 1>(wa, wc, grid, block, lds_bytes, as_stream(stream), d) : upd16d_inst<false, 1>(wa, wc, grid, block, lds_bytes, as_stream(stream), d);
    if (rc) return rc;
    MAPPO_CHECK_LAUNCH("actor_critic_update");
    return MAPPO_OK;
  }
  DualArgs d = {};
  UpdArgs &a = d.a, &c = d.c;
  a.params = actor_params; a.x = obs; a.rows = rows; a.slabs = slabs; a.slab_stride = slab_stride; a.slab_col0 = actor_col0;
  a.desc = *actor_desc; a.B = B; a.avail = avail; a.actions = actions; a.old_logp = old_logp; a.adv = adv; a.active = active;
  a.mb_moments = mb_moments; a.partials = actor_partials; a.cfg = *cfg;
  c.params = critic_params; c.x = share_obs; c.rows = rows; c.slabs = slabs; c.slab_stride = slab_stride; c.slab_col0 = critic_col0;
  c.desc = *critic_desc; c.B = B; c.v_old = v_old; c.returns = returns; c.active = active; c.vn_state = vn_state;
  c.mb_moments = mb_moments; c.partials = critic_partials; c.cfg = *cfg;
  int np = fit_waves(a.desc, 4);
  const int npc = fit_waves(c.desc, 4);
  np = np < npc ? np : npc;
  if (int rc = prep_pair(a, np, "actor_critic_update")) return rc;
  if (int rc = prep_pair(c, np, "actor_critic_update")) return rc;
  const int ta = a.map.total, tc = c.map.total;
  const size_t lds_bytes = (size_t)(ta > tc ? ta : tc) * sizeof(float);
  MAPPO_REQUIRE(lds_bytes <= LDS_DYN_MAX, "actor_critic_update: needs %zu B of LDS", lds_bytes);
  d.nA = d.nC = mappo_dual_update_slabs(nullptr, nullptr, B);
#ifdef MLP_STAMPS
  a.stamps = c.stamps = nullptr;
#endif
  dim3 grid((unsigned)(d.nA + d.nC)), block(2 * WAVE * np);
  const bool wa = a.desc.in_dim > 32, wc = c.desc.in_dim > 32, relu = a.desc.use_relu != 0;
  int rc;
  switch (a.desc.layer_N) {
    case 0: rc = relu ? upd2d_inst<true, 0>(wa, wc, grid, block, lds_bytes, as_stream(stream), d) : upd2d_inst<false, 0>(wa, wc, grid, block, lds_bytes, as_stream(stream), d); break;
    case 1: rc = relu ? upd2d_inst<true, 1>(wa, wc, grid, block, lds_bytes, as_stream(stream), d) : upd2d_inst<false, 1>(wa, wc, grid, block, lds_bytes, as_stream(stream), d); break;
    default: rc = relu ? upd2d_inst<true, 2>(wa, wc, grid, block, lds_bytes, as_stream(stream), d) : upd2d_inst<false, 2>(wa, wc, grid, block, lds_bytes, as_stream(stream), d); break;
  }
  if (rc) return rc;
  MAPPO_CHECK_LAUNCH("actor_critic_update");
  return MAPPO_OK;
}

// ------------------------------------------------------------------------------------------------
// wide_l1_bwd_kernel: layer-1 weight gradient and feature-norm gradients for in_dim > 64.
//   dW1[f][k]  = sum_s dz1[f][s] * xn[k][s],   xn = xhat0 * gamma0 + beta0
//   dxn[k][s]  = sum_f W1[f][k] * dz1[f][s] ;  dgamma0[k] = sum_s dxn * xhat0 ;  dbeta0[k] = sum_s dxn
// A workgroup owns ONE 64-column chunk of W1 (blockIdx.y) and one share of the row tiles (blockIdx.x); its 4 waves
// walk row tiles, keep the chunk's 64x64 dW1 block in registers, and the per-sample-lane partial sums of the
// feature-norm gradients are reduced across lanes once at the end.  One slab row per blockIdx.x.
// ------------------------------------------------------------------------------------------------
struct WideArgs {
  const float *params, *x;
  const int32_t *rows;
  const float *wide_ws;
  float *slabs;
  int64_t slab_stride, slab_col0;
  NetOff off;
  int64_t B;
  int D, use_feature_norm;
};

__global__ __launch_bounds__(256, 1) void wide_l1_bwd_kernel(WideArgs p) {
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & (WAVE - 1), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)), l31 = lane & 31, half = lane >> 5;
  const int D = p.D, c0 = blockIdx.y * MAXD, kc = min(MAXD, D - c0);
  float *sW = lds;                                   // [64 kk][WP]  W1 chunk, k-major
  float *sG = sW + MAXD * WP, *sBt = sG + MAXD;      // gamma0 / beta0 of the chunk
  float *tD = sBt + MAXD + wave * (2 * HID * TP);    // [64 f][TP]   dz1 tile
  float *tXc = tD + HID * TP;                        // [64 kk][TP]  xhat0 tile of the chunk
  stage_w1_chunk(sW, p.params + p.off.w1, D, c0, kc);
  for (int e = threadIdx.x; e < MAXD; e += blockDim.x) {
    const bool in = e < kc;
    sG[e] = in ? (p.use_feature_norm ? p.params[p.off.fn_w + c0 + e] : 1.f) : 0.f;
    sBt[e] = (in && p.use_feature_norm) ? p.params[p.off.fn_b + c0 + e] : 0.f;
  }
  __syncthreads();
  const float *dz1T = p.wide_ws, *stats = p.wide_ws + (int64_t)HID * p.B;
  f32x16 gW[2][2], accB[2], accG[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { accB[i][r] = 0.f; accG[i][r] = 0.f; }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) gW[i][j][r] = 0.f;
  }
  const int64_t n_tiles = (p.B + TS - 1) / TS;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t base = tile * TS;
    const int n_valid = (int)min((int64_t)TS, p.B - base);
    const bool ok = l31 < n_valid;
    // dz1 tile [f][s] (feature-major source: 128-B segments)
    {
      float v[HID / 2];                                // lane (s = l31, half) takes features 2 i + half: all 32 loads in flight
      const int64_t col = base + min(l31, n_valid - 1);
#pragma unroll
      for (int i = 0; i < HID / 2; ++i) v[i] = dz1T[(int64_t)(2 * i + half) * p.B + col];
#pragma unroll
      for (int i = 0; i < HID / 2; ++i) tD[(2 * i + half) * TP + l31] = ok ? v[i] : 0.f;
    }
    const int64_t row = ok ? (p.rows ? (int64_t)p.rows[base + l31] : base + l31) : 0;
    const float mean0 = ok ? stats[base + l31] : 0.f, rstd0 = ok ? stats[p.B + base + l31] : 1.f;
    wide_commit_chunk(tXc, p.x + row * D, D, c0, ok, mean0, rstd0, l31, half);
    wave_lds_sync();
    // dW1 chunk
    {
      const float g0 = sG[l31], b0 = sBt[l31], g1 = sG[32 + l31], b1 = sBt[32 + l31];
#pragma unroll 2
      for (int ss = 0; ss < TS / 2; ++ss) {
        const int s = 2 * ss + half;
        const float a0 = tD[l31 * TP + s], a1 = tD[(32 + l31) * TP + s];
        // padding samples carry dz1 = 0, padding columns carry gamma = beta = 0
        const float x0 = tXc[l31 * TP + s] * g0 + b0, x1 = tXc[(32 + l31) * TP + s] * g1 + b1;
        gW[0][0] = mfma(a0, x0, gW[0][0]);
        gW[0][1] = mfma(a0, x1, gW[0][1]);
        gW[1][0] = mfma(a1, x0, gW[1][0]);
        gW[1][1] = mfma(a1, x1, gW[1][1]);
      }
    }
    if (p.use_feature_norm) {
      f32x16 dX[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dX[t][r] = 0.f;
#pragma unroll 4
      for (int ff = 0; ff < HID / 2; ++ff) {
        const int f = 2 * ff + half;
        const float b = tD[f * TP + l31];
        dX[0] = mfma(sW[l31 * WP + f], b, dX[0]);
        dX[1] = mfma(sW[(32 + l31) * WP + f], b, dX[1]);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          accB[t][r] += dX[t][r];
          accG[t][r] += dX[t][r] * tXc[(32 * t + ROWMAP(r, half)) * TP + l31];
        }
    }
    wave_lds_sync();
  }
  // ---- cross-lane (sample) reduction of the feature-norm partial sums, through this wave's tiles ----
  float gFnB = 0.f, gFnW = 0.f;
  if (p.use_feature_norm) {
    wave_lds_sync();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        tD[(32 * t + ROWMAP(r, half)) * TP + l31] = accB[t][r];
        tXc[(32 * t + ROWMAP(r, half)) * TP + l31] = accG[t][r];
      }
    wave_lds_sync();
    for (int j = 0; j < TS; ++j) { gFnB += tD[lane * TP + j]; gFnW += tXc[lane * TP + j]; }
  }
  // ---- reduce the 4 waves through LDS (reusing wave 0's tiles), write this workgroup's slab columns ----
  __syncthreads();
  float *red = sBt + MAXD;                           // >= 64*64 + 128 floats available (4 waves x 2 tiles)
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          float old[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) old[r] = (w == 0) ? 0.f : red[(32 * ti + ROWMAP(r, half)) * MAXD + 32 * tj + l31];
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(32 * ti + ROWMAP(r, half)) * MAXD + 32 * tj + l31] = old[r] + gW[ti][tj][r];
        }
      const float ob = (w == 0) ? 0.f : red[HID * MAXD + lane], og = (w == 0) ? 0.f : red[HID * MAXD + MAXD + lane];
      red[HID * MAXD + lane] = ob + gFnB;
      red[HID * MAXD + MAXD + lane] = og + gFnW;
    }
    __syncthreads();
  }
  float *slab = p.slabs + (size_t)blockIdx.x * p.slab_stride + p.slab_col0;
  for (int e = threadIdx.x; e < HID * kc; e += blockDim.x) {
    const int f = e / kc, kk = e - f * kc;
    slab[p.off.w1 + f * D + c0 + kk] = red[f * MAXD + kk];
  }
  if (p.use_feature_norm)
    for (int e = threadIdx.x; e < kc; e += blockDim.x) {
      slab[p.off.fn_b + c0 + e] = red[HID * MAXD + e];
      slab[p.off.fn_w + c0 + e] = red[HID * MAXD + MAXD + e];
    }
}

extern "C" int64_t mappo_wide_workspace_floats(int64_t B) { return wide_z1_offset(B) + (int64_t)HID * B; }     // + z1 [B][64] (mlp_wide16.h)

extern "C" int32_t mappo_wide_l1_slabs(int64_t B) {
  // slab rows mappo_wide_l1_backward may write (never more than the update launch's own mappo_mlp_backward_slabs(B))
  return mappo_mlp_backward_slabs(B);
}

extern "C" int mappo_wide_l1_backward(const float *params, const mappo_net_desc *desc, const float *x, const int32_t *rows,
                                      int64_t B, const float *wide_ws, float *slabs, int64_t slab_stride, int64_t slab_col0,
                                      int32_t layout, mappo_stream_t stream) {
  if (int rc = check_desc_trunk(desc, "wide_l1_backward")) return rc;
  MAPPO_REQUIRE(desc->in_dim > MAXD, "wide_l1_backward: in_dim %d is handled inside the update kernels", desc->in_dim);
  MAPPO_REQUIRE(params && x && wide_ws && slabs && B > 0, "wide_l1_backward: bad arguments");
  MAPPO_REQUIRE(layout == MAPPO_WIDE_LAYOUT_BLOCKED || layout == MAPPO_WIDE_LAYOUT_FEATURE_MAJOR,
                "wide_l1_backward: layout %d (pass mappo_wide_layout(desc, producer) of the launch that filled the workspace)", (int)layout);
  MAPPO_REQUIRE(layout != MAPPO_WIDE_LAYOUT_BLOCKED || desc->in_dim <= 512, "wide_l1_backward: the blocked layout exists for in_dim <= 512 only");
  {
    if (layout == MAPPO_WIDE_LAYOUT_BLOCKED) {
      // 16x16x4 kernel (mlp_wide16.h): raw products, every input element read once
      const NetOff o = net_offsets(*desc);
      MAPPO_REQUIRE(slab_col0 >= 0 && slab_col0 + o.total <= slab_stride, "wide_l1_backward: slab column range");
      WideBwd16Args w = {};
      w.params = params; w.x = x; w.rows = rows; w.wide_ws = wide_ws; w.slabs = slabs; w.slab_stride = slab_stride; w.slab_col0 = slab_col0;
      w.B = B; w.D = desc->in_dim; w.w1 = o.w1; w.fn_w = desc->use_feature_norm ? o.fn_w : -1; w.fn_b = desc->use_feature_norm ? o.fn_b : -1;
      const int nch = (desc->in_dim + 63) / 64;
      w.nca = nch <= 2 ? 2 : (nch <= 4 ? 4 : 8);
      const int rows_max = mappo_mlp_backward_slabs(B);
      w.groups = 8 / w.nca;
      if (rows_max < w.groups) w.groups = 1;
      const int gx = rows_max / w.groups;
      (void)wide16_launch_l1_bwd(w, dim3((unsigned)gx), as_stream(stream));
      MAPPO_CHECK_LAUNCH("wide_l1_backward");
      return MAPPO_OK;
    }
  }
  MAPPO_CLEAR_STICKY();
  WideArgs a = {};
  a.params = params; a.x = x; a.rows = rows; a.wide_ws = wide_ws; a.slabs = slabs; a.slab_stride = slab_stride; a.slab_col0 = slab_col0;
  a.off = net_offsets(*desc); a.B = B; a.D = desc->in_dim; a.use_feature_norm = desc->use_feature_norm;
  MAPPO_REQUIRE(slab_col0 >= 0 && slab_col0 + a.off.total <= slab_stride, "wide_l1_backward: slab column range");
  const size_t lds_bytes = (size_t)(MAXD * WP + 2 * MAXD + 4 * 2 * HID * TP) * sizeof(float);
  static const hipError_t e_ = hipFuncSetAttribute((const void *)wide_l1_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_DYN_MAX);
  if (e_ != hipSuccess) { mappo_set_error("wide_l1_backward: hipFuncSetAttribute: %s", hipGetErrorString(e_)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  dim3 grid((unsigned)mappo_wide_l1_slabs(B), (unsigned)((desc->in_dim + MAXD - 1) / MAXD));
  hipLaunchKernelGGL(wide_l1_bwd_kernel, grid, dim3(256), lds_bytes, as_stream(stream), a);
  MAPPO_CHECK_LAUNCH("wide_l1_backward");
  return MAPPO_OK;
}

// statistics of one fused update from the two kernels' per-workgroup partial sums (same layout as
// mappo_ppo_loss_fwd_bwd's `stats`)
__global__ __launch_bounds__(256) void update_stats_kernel(const double *__restrict__ pa, const double *__restrict__ pc, int na,
                                                          int nc, const double *__restrict__ mb_moments, int use_policy_active,
                                                          int use_value_active, double *__restrict__ stats,
                                                          double *__restrict__ acc) {
  __shared__ double smem[16 * 4];
  double v[4] = {0.0, 0.0, 0.0, 0.0};    // sum w*min, sum w*H, sum ratio, sum w_v*l
  for (int b = threadIdx.x; b < na; b += blockDim.x) { v[0] += pa[b * 4 + 0]; v[1] += pa[b * 4 + 1]; v[2] += pa[b * 4 + 2]; }
  for (int b = threadIdx.x; b < nc; b += blockDim.x) v[3] += pc[b * 4 + 0];
  block_sum<4>(v, smem);
  if (threadIdx.x == 0) {
    const double sa = mb_moments[2] > 0.0 ? mb_moments[2] : 1.0;
    const double Bg = mb_moments[3] > 0.0 ? mb_moments[3] : 1.0;
    const double den_pi = use_policy_active ? sa : Bg, den_v = use_value_active ? sa : Bg;
    stats[0] = v[3] / den_v;
    stats[1] = -v[0] / den_pi;
    stats[2] = v[1] / den_pi;
    stats[3] = v[2] / Bg;
    stats[4] = mb_moments[2];
    stats[5] = mb_moments[3];
    if (acc) { acc[0] += stats[0]; acc[1] += stats[1]; acc[2] += stats[2]; acc[3] += stats[3]; }   // train_info sums (r_mappo.py:207-212)
  }
}

extern "C" int mappo_update_stats(const double *actor_partials, int32_t n_actor, const double *critic_partials,
                                  int32_t n_critic, const double *mb_moments, const mappo_ppo_cfg *cfg, double *stats,
                                  double *acc, mappo_stream_t stream) {
  MAPPO_REQUIRE(critic_partials && mb_moments && cfg && stats && n_critic > 0 && n_actor >= 0, "update_stats: bad arguments");
  hipLaunchKernelGGL(update_stats_kernel, dim3(1), dim3(256), 0, as_stream(stream), actor_partials, critic_partials,
                     actor_partials ? (int)n_actor : 0, (int)n_critic, mb_moments, cfg->use_policy_active_masks,
                     cfg->use_value_active_masks, stats, acc);
  MAPPO_CHECK_LAUNCH("update_stats");
  return MAPPO_OK;
}

// ------------------------------------------------------------------------------------------------
// self test of the documented v_mfma_f32_32x32x2_f32 lane maps (tests/test_gpu_kernels.py)
// ------------------------------------------------------------------------------------------------
__global__ void selftest_mfma_kernel(const float *A, const float *Bm, float *Dm) {
  const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = mfma(A[l31 * 2 + half], Bm[half * 32 + l31], acc);     // A[i=l31][k=half], B[k=half][j=l31]
#pragma unroll
  for (int r = 0; r < 16; ++r) Dm[ROWMAP(r, half) * 32 + l31] = acc[r];
}

extern "C" int mappo_selftest_mfma(const float *A, const float *Bm, float *D, mappo_stream_t stream) {
  MAPPO_REQUIRE(A && Bm && D, "selftest_mfma: null pointer");
  hipLaunchKernelGGL(selftest_mfma_kernel, dim3(1), dim3(WAVE), 0, as_stream(stream), A, Bm, D);
  MAPPO_CHECK_LAUNCH("selftest_mfma");
  return MAPPO_OK;
}

#endif  // MLP_TU_MAIN
