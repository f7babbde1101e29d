// gru.hip — K9: the recurrent layer of R_Actor / R_Critic (onpolicy/algorithms/utils/rnn.py:7-80):
//   h_t = GRU(x_t, h_{t-1} * mask_t)   (torch nn.GRU, 1 layer, gate order r,z,n; rnn.py:13,27,67)
//   y_t = LayerNorm(h_t)               (rnn.py:22,79)          -> head (distributions.py:55-68 | v_out)
// The reference's mask-segmented sequence run (rnn.py:30-77) equals multiplying h by mask_t before every step
// (SURVEY.md §3.4), which is what these kernels do.
//
// Same transposed fp32-MFMA formulation as mlp.hip: a wavefront owns 32 sequences (chunks) and walks their L steps;
// a lane holds one sequence and 32 of the 64 hidden features, so the gate nonlinearities, the state update, the
// LayerNorm and the whole cell backward are per-lane register code.  W_ih / W_hh (2 x 192x64 fp32 = 98 KB) stay in
// LDS for the workgroup's lifetime (k-major, row stride 193: conflict-free both for W.x and for W^T.dg).
// All per-step scratch between the kernels is FEATURE-MAJOR ([feature][row], row = t*Nc + c), so accumulator-layout
// registers are loaded/stored as coalesced 128-B segments and MFMA B operands can be read straight from HBM/L2.
//
//   gru_fwd_kernel        x_T[64][B], h0 -> h_t for t < L; optional per-step head output / action sampling (rollout,
//                         get_values, evaluate); with scratch stores hm, r, z, n, gh_n, h' per step
//   gru_gi_kernel / gru_dx_kernel   the input-side products W_ih x + b_ih and W_ih^T dgi over all rows (no time dependence)
//   gru_fwd_train2_kernel the training forward recurrence on precomputed input gates, two waves per 32 sequences
//   gru_head_bwd_kernel   LayerNorm + head + in-kernel PPO loss (actor | critic) and their backward over all rows:
//                         d h'_t without the recurrent term, head / rnn.norm gradient slabs, loss partial sums
//   gru_cell_bwd2_kernel  reverse time, two waves per 32 sequences: cell backward, carry = (W_hh^T dgh + dh*z) * mask;
//                         writes dgi_T, dghn_T
//   gru_wgrad_kernel      dW_ih, dW_hh, db_ih, db_hh = row-tile GEMMs over (dgi, x) and (dgh, hm) -> slabs
#include <stdlib.h>
#include "mlp_core.h"
#include "mlp_trunk16r.h"
#include "gru_step3.h"

#define N_SCR 6             // scratch components
#define SCR_HM 0
#define SCR_R 1
#define SCR_Z 2
#define SCR_N 3
#define SCR_GHN 4
#define SCR_HS 5

// stage W[g][k] (row-major [192][64]) -> dst[k*GS + g]; batched unconditional 16-byte loads
__device__ __forceinline__ void stage_gru_weight(float *dst, const float *__restrict__ src) {
  const int nthr = blockDim.x, tid = threadIdx.x, n4 = NG * HID / 4;
  for (int i0 = 0; i0 < n4; i0 += 8 * nthr) {
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = reinterpret_cast<const float4 *>(src)[min(i0 + j * nthr + tid, n4 - 1)];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = i0 + j * nthr + tid;
      if (i < n4) {
        const int g = i >> 4, k = (i & 15) << 2;
        dst[(k + 0) * GS + g] = v[j].x; dst[(k + 1) * GS + g] = v[j].y;
        dst[(k + 2) * GS + g] = v[j].z; dst[(k + 3) * GS + g] = v[j].w;
      }
    }
  }
}

__device__ __forceinline__ void stage_gru_all(float *lds, const GruLds &m, const float *__restrict__ params, const NetOff &o, int A,
                                              bool need_ih, bool need_head) {
  if (need_ih) stage_gru_weight(lds + m.wih, params + o.gru_wih);
  stage_gru_weight(lds + m.whh, params + o.gru_whh);
  const int nthr = blockDim.x, tid = threadIdx.x;
  for (int e = tid; e < NG; e += nthr) { lds[m.bih + e] = params[o.gru_bih + e]; lds[m.bhh + e] = params[o.gru_bhh + e]; }
  for (int e = tid; e < HID; e += nthr) { lds[m.nw + e] = params[o.rn_w + e]; lds[m.nb + e] = params[o.rn_b + e]; }
  if (need_head) {
    for (int e = tid; e < 32; e += nthr) lds[m.bh + e] = e < A ? params[o.bh + e] : 0.f;
    for (int e = tid; e < HID * 32; e += nthr) {
      const int a = e >> 6, k = e & 63;
      lds[m.wh + k * HP + a] = (a < A) ? params[o.wh + a * HID + k] : 0.f;
    }
  }
}

// The same for the rollout step kernels (256 threads, both matrices + head), with ONE memory latency: every global load is issued
// before the first LDS store.  stage_gru_all walks W_ih, W_hh, the bias / norm vectors and the head one after the other — about
// fifteen dependent L2 round trips, half of a step-sized launch (a step is one tile per wave: nothing amortises the staging).
__device__ __forceinline__ void stage_gru_step_1shot(float *lds, const GruLds &m, const float *__restrict__ params, const NetOff &o, int A) {
  const int tid = threadIdx.x;                                 // blockDim.x == 256
  constexpr int n4 = NG * HID / 4, J = n4 / 256;               // 12 float4 per thread and matrix
  float4 wi[J], wh[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    wi[j] = reinterpret_cast<const float4 *>(params + o.gru_wih)[j * 256 + tid];
    wh[j] = reinterpret_cast<const float4 *>(params + o.gru_whh)[j * 256 + tid];
  }
  // vectors: bih [192] | bhh [192] | rn_w [64] | rn_b [64] | bh [32] = 544 entries, 3 per thread
  float vv[3]; int vd[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int e = j * 256 + tid;
    int src = -1, dst = -1;
    if (e < NG) { src = o.gru_bih + e; dst = m.bih + e; }
    else if (e < 2 * NG) { src = o.gru_bhh + e - NG; dst = m.bhh + e - NG; }
    else if (e < 2 * NG + HID) { src = o.rn_w + e - 2 * NG; dst = m.nw + e - 2 * NG; }
    else if (e < 2 * NG + 2 * HID) { src = o.rn_b + e - 2 * NG - HID; dst = m.nb + e - 2 * NG - HID; }
    else if (e < 2 * NG + 2 * HID + 32) { const int i = e - 2 * NG - 2 * HID; dst = m.bh + i; if (i < A) src = o.bh + i; }
    const float ld = params[src >= 0 ? src : 0];
    vv[j] = src >= 0 ? ld : 0.f; vd[j] = dst;
  }
  // head Wh [A][64]: 16 A float4 (A <= 32: two per thread)
  const int n4h = 16 * A;
  float4 hv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) hv[j] = reinterpret_cast<const float4 *>(params + o.wh)[min(j * 256 + tid, n4h - 1)];
  // ---- stores ----
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int i = j * 256 + tid, g = i >> 4, k = (i & 15) << 2;
    float *di = lds + m.wih + k * GS + g, *dh = lds + m.whh + k * GS + g;
    di[0] = wi[j].x; di[GS] = wi[j].y; di[2 * GS] = wi[j].z; di[3 * GS] = wi[j].w;
    dh[0] = wh[j].x; dh[GS] = wh[j].y; dh[2 * GS] = wh[j].z; dh[3 * GS] = wh[j].w;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) if (vd[j] >= 0) lds[vd[j]] = vv[j];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = j * 256 + tid;
    if (i < n4h) {
      const int a = i >> 4, k = (i & 15) << 2;
      float *d = lds + m.wh + k * HP + a;
      d[0] = hv[j].x; d[HP] = hv[j].y; d[2 * HP] = hv[j].z; d[3 * HP] = hv[j].w;
    }
  }
  for (int e = tid; e < HID * (32 - A); e += 256) {            // head columns a >= A are zero
    const int k = e / (32 - A), a = A + e - k * (32 - A);
    lds[m.wh + k * HP + a] = 0.f;
  }
}

// row-major state [row][64] <-> accumulator layout (lane = sequence, registers = 32 of its 64 features)
__device__ __forceinline__ void load_state_rowmajor(f32x16 (&h)[2], const float *__restrict__ src, int64_t row, bool ok, int half) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 v = *reinterpret_cast<const float4 *>(src + (ok ? row : 0) * HID + 32 * t + 8 * q + 4 * half);
      if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      h[t][4 * q + 0] = v.x; h[t][4 * q + 1] = v.y; h[t][4 * q + 2] = v.z; h[t][4 * q + 3] = v.w;
    }
}
__device__ __forceinline__ void store_state_rowmajor(float *__restrict__ dst, int64_t row, const f32x16 (&h)[2], bool ok, int half) {
  if (!ok) return;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4 *>(dst + row * HID + 32 * t + 8 * q + 4 * half) =
          make_float4(h[t][4 * q + 0], h[t][4 * q + 1], h[t][4 * q + 2], h[t][4 * q + 3]);
}
// feature-major [64][ld] <-> accumulator layout: for a fixed register the 32 lanes of a half touch 128 contiguous bytes
__device__ __forceinline__ void load_fm(f32x16 (&v)[2], const float *__restrict__ src, int64_t ld, int64_t col, bool ok, int half) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) v[t][r] = src[(int64_t)(32 * t + ROWMAP(r, half)) * ld + (ok ? col : 0)];   // unconditional: all 32 in flight
  if (!ok) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) v[t][r] = 0.f;
  }
}
__device__ __forceinline__ void store_fm(float *__restrict__ dst, int64_t ld, int64_t col, const f32x16 (&v)[2], bool ok, int half) {
  if (!ok) return;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(int64_t)(32 * t + ROWMAP(r, half)) * ld + col] = v[t][r];
}
__device__ __forceinline__ void regs_to_tile64(float *tile, const f32x16 (&v)[2], int l31, int half) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[(32 * t + ROWMAP(r, half)) * TP + l31] = v[t][r];
}

// LayerNorm(64) statistics of an accumulator-layout vector
__device__ __forceinline__ void ln_stats(const f32x16 (&v)[2], float &mean, float &rstd) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += v[t][r];
  mean = xhalf_sum(s) * (1.f / HID);
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float c = v[t][r] - mean; q += c * c; }
  rstd = 1.0f / sqrtf(xhalf_sum(q) * (1.f / HID) + LN_EPS);
}

// head on a tile of normalised states (affine applied on read): z^T[a][s]
__device__ __forceinline__ f32x16 gru_head(const float *lds, const GruLds &m, const float *tN, int l31, int half) {
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 b = *reinterpret_cast<const float4 *>(lds + m.bh + 8 * q + 4 * half);
    acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
  }
  const float *sW = lds + m.wh, *sG = lds + m.nw, *sB = lds + m.nb;
#pragma unroll 16
  for (int kk = 0; kk < HID / 2; ++kk) {
    const int k = 2 * kk + half;
    acc = mfma(sW[k * HP + l31], tN[k * TP + l31] * sG[k] + sB[k], acc);
  }
  return acc;
}

// ---- one GRU cell step for a tile: gates from x (B operand straight from feature-major HBM) and hm (LDS tile) ----
struct CellOut { f32x16 r[2], z[2], n[2], ghn[2]; };

// PRE_GI: gi = W_ih x + b_ih comes precomputed (mappo_gru_input_gates: feature-major giT[192][ldx]); the cell then only
// runs the W_hh products — half the MFMA work of the sequential kernel, and W_ih need not be resident in LDS.
template <bool PRE_GI>
__device__ __forceinline__ void gru_cell(CellOut &c, const float *lds, const GruLds &m, const float *__restrict__ xT, int64_t ldx,
                                         int64_t col, bool ok, const float *tHm, int l31, int half) {
  f32x16 arz[4], ain[2], ahn[2];
  if (PRE_GI) {
    const int64_t cc = ok ? col : 0;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b = *reinterpret_cast<const float4 *>(lds + m.bhh + 32 * t + 8 * q + 4 * half);
        const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) arz[t][4 * q + e] = xT[(int64_t)(32 * t + 8 * q + 4 * half + e) * ldx + cc] + bb[e];
      }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b = *reinterpret_cast<const float4 *>(lds + m.bhh + 128 + 32 * t + 8 * q + 4 * half);
        const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ain[t][4 * q + e] = xT[(int64_t)(128 + 32 * t + 8 * q + 4 * half + e) * ldx + cc];
          ahn[t][4 * q + e] = bb[e];
        }
      }
    const float *sH = lds + m.whh;
#pragma unroll 4
    for (int kk = 0; kk < HID / 2; ++kk) {
      const int k = 2 * kk + half;
      const float bh = tHm[k * TP + l31];
#pragma unroll
      for (int t = 0; t < 4; ++t) arz[t] = mfma(sH[k * GS + 32 * t + l31], bh, arz[t]);
#pragma unroll
      for (int t = 0; t < 2; ++t) ahn[t] = mfma(sH[k * GS + 128 + 32 * t + l31], bh, ahn[t]);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        c.r[t][r] = sigmoidf_(arz[t][r]);
        c.z[t][r] = sigmoidf_(arz[2 + t][r]);
        c.ghn[t][r] = ahn[t][r];
        c.n[t][r] = tanhf_(ain[t][r] + c.r[t][r] * ahn[t][r]);
      }
    return;
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 a = *reinterpret_cast<const float4 *>(lds + m.bih + 32 * t + 8 * q + 4 * half);
      const float4 b = *reinterpret_cast<const float4 *>(lds + m.bhh + 32 * t + 8 * q + 4 * half);
      arz[t][4 * q + 0] = a.x + b.x; arz[t][4 * q + 1] = a.y + b.y; arz[t][4 * q + 2] = a.z + b.z; arz[t][4 * q + 3] = a.w + b.w;
    }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 a = *reinterpret_cast<const float4 *>(lds + m.bih + 128 + 32 * t + 8 * q + 4 * half);
      const float4 b = *reinterpret_cast<const float4 *>(lds + m.bhh + 128 + 32 * t + 8 * q + 4 * half);
      ain[t][4 * q + 0] = a.x; ain[t][4 * q + 1] = a.y; ain[t][4 * q + 2] = a.z; ain[t][4 * q + 3] = a.w;
      ahn[t][4 * q + 0] = b.x; ahn[t][4 * q + 1] = b.y; ahn[t][4 * q + 2] = b.z; ahn[t][4 * q + 3] = b.w;
    }
  const float *sI = lds + m.wih, *sH = lds + m.whh;
  // B operand of W_ih.x: lane (s, khalf) needs x[k = 2kk+khalf][s] = one coalesced 128-B segment per half -> registers
  float bx[HID / 2];
#pragma unroll
  for (int kk = 0; kk < HID / 2; ++kk) bx[kk] = xT[(int64_t)(2 * kk + half) * ldx + (ok ? col : 0)];      // unconditional: all in flight
  if (!ok) {
#pragma unroll
    for (int kk = 0; kk < HID / 2; ++kk) bx[kk] = 0.f;
  }
#pragma unroll 4
  for (int kk = 0; kk < HID / 2; ++kk) {
    const int k = 2 * kk + half;
    const float bxx = bx[kk], bh = tHm[k * TP + l31];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      arz[t] = mfma(sI[k * GS + 32 * t + l31], bxx, arz[t]);
      arz[t] = mfma(sH[k * GS + 32 * t + l31], bh, arz[t]);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      ain[t] = mfma(sI[k * GS + 128 + 32 * t + l31], bxx, ain[t]);
      ahn[t] = mfma(sH[k * GS + 128 + 32 * t + l31], bh, ahn[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      c.r[t][r] = sigmoidf_(arz[t][r]);
      c.z[t][r] = sigmoidf_(arz[2 + t][r]);
      c.ghn[t][r] = ahn[t][r];
      c.n[t][r] = tanhf_(ain[t][r] + c.r[t][r] * ahn[t][r]);
    }
}

// ---- forward kernel -----------------------------------------------------------------------------------
// (GruFwdArgs: gru_step3.h)
// PRE_GI: input gates precomputed (training); HM: head mode 0 none | 1 out[B][A] | 2 sample.  Compile-time so that each use
// (rollout step with sampling, evaluation with head output, training with scratch stores) is a lean instantiation — as one
// kernel with runtime switches the register allocator spilled ~200 registers.
template <bool PRE_GI, int HM>
__global__ __launch_bounds__(256, 1) void gru_fwd_kernel(GruFwdArgs p) {
  extern __shared__ __align__(16) float lds[];
  const GruLds &m = p.map;
  const int n_waves = p.tile_waves;
  const int lane = threadIdx.x & (WAVE - 1), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)), l31 = lane & 31, half = lane >> 5;
  stage_gru_all(lds, m, p.params, p.off, p.A, !PRE_GI, HM != 0);
  __syncthreads();
  if (wave >= n_waves) return;                           // staging helper (no workgroup barrier below)
  float *tHm = lds + m.tiles + wave * m.wave_stride;     // [64][TP] masked previous state (B operand)
  float *tN = tHm + HID * TP;                            // [64][TP] normalised state (head input)   (head modes only)
  float *tZ = tN + HID * TP;                             // [32][TP] head output [s][a]               (head modes only)
  const int64_t B = (int64_t)p.L * p.Nc;
  const int n_tiles = (p.Nc + TS - 1) / TS;
  for (int tile = blockIdx.x * n_waves + wave; tile < n_tiles; tile += gridDim.x * n_waves) {
    const int c = tile * TS + l31;
    const bool ok = c < p.Nc;
    const int n_valid = min(TS, p.Nc - tile * TS);
    f32x16 h[2];
    load_state_rowmajor(h, p.h0, ok ? (p.h0_rows ? (int64_t)p.h0_rows[c] : (int64_t)c) : 0, ok, half);
    for (int t = 0; t < p.L; ++t) {
      const int64_t col = (int64_t)t * p.Nc + c;
      float mk = 0.f;
      if (ok) mk = p.masks[p.rows ? (int64_t)p.rows[col] : col];
      f32x16 hm[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) hm[tt][r] = h[tt][r] * mk;
      regs_to_tile64(tHm, hm, l31, half);
      wave_lds_sync();
      CellOut co;
      gru_cell<PRE_GI>(co, lds, m, PRE_GI ? p.giT : p.xT, B, col, ok, tHm, l31, half);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[tt][r] = (1.f - co.z[tt][r]) * co.n[tt][r] + co.z[tt][r] * hm[tt][r];
      if (p.scratch) {
        const int64_t comp = (int64_t)p.L * HID * p.Nc;
        float *base = p.scratch + (int64_t)t * HID * p.Nc;
        store_fm(base + SCR_HM * comp, p.Nc, c, hm, ok, half);
        store_fm(base + SCR_R * comp, p.Nc, c, co.r, ok, half);
        store_fm(base + SCR_Z * comp, p.Nc, c, co.z, ok, half);
        store_fm(base + SCR_N * comp, p.Nc, c, co.n, ok, half);
        store_fm(base + SCR_GHN * comp, p.Nc, c, co.ghn, ok, half);
        store_fm(base + SCR_HS * comp, p.Nc, c, h, ok, half);
      }
      if (HM != 0) {
        float mean, rstd;
        ln_stats(h, mean, rstd);
        f32x16 xh[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 16; ++r) xh[tt][r] = (h[tt][r] - mean) * rstd;
        regs_to_tile64(tN, xh, l31, half);
        wave_lds_sync();
        const f32x16 z = gru_head(lds, m, tN, l31, half);
        head_to_tile(tZ, z, p.A, l31, half);
        wave_lds_sync();
        const int64_t row0 = (int64_t)t * p.Nc + tile * TS;
        if (HM == 1) {
          for (int e = lane; e < n_valid * p.A; e += WAVE) {
            const int s = e / p.A, a = e - s * p.A;
            p.out[row0 * p.A + e] = tZ[s * TP + a];
          }
        } else if (lane < n_valid) {
          const int64_t i = row0 + lane;
          const uint64_t ctr = p.counter + (p.counter_dev ? *p.counter_dev : 0ull);
          float action, logp;
          categorical_act_lane(tZ + lane * TP, p.A, p.avail ? p.avail + i * p.A : nullptr, p.deterministic != 0, p.seed, ctr,
                               (uint64_t)i, action, logp);
          p.actions[i] = action;
          p.logp[i] = logp;
        }
      }
      wave_lds_sync();
    }
    if (p.h_last) store_state_rowmajor(p.h_last, c, h, ok, half);
  }
}

// ---- backward kernel ----------------------------------------------------------------------------------
struct GruBwdArgs {
  const float *params;
  NetOff off;
  GruLds map;
  const float *scratch;       // from the forward: [6][L][64][Nc]
  const float *masks;
  const int32_t *rows;
  int L, Nc, A, head;         // head 1: actor loss, 2: critic loss
  // loss inputs (buffer order, indexed through rows)
  const float *avail, *actions, *old_logp, *adv, *active, *v_old, *returns, *vn_state;
  const double *mb_moments;
  mappo_ppo_cfg cfg;
  // outputs
  float *dxT;                 // [64][B]
  float *dgiT;                // [192][B]  (r, z, n parts of d gi; the r, z parts of d gh are identical)
  float *dghnT;               // [64][B]   n part of d gh
  float *slabs;               // [gridDim.x][slab_stride]; this kernel writes head + rnn.norm columns at slab_col0 + offsets
  int64_t slab_stride, slab_col0;
  double *partials;           // [gridDim.x][4]
};

// ---- the row-local part of one backward step: y = LayerNorm(h'_t) -> head -> PPO loss -> d h'_t (without the recurrent term) ----
// One wave, 32 rows (hsT: feature-major h' of step t, column c); accumulates the head / rnn.norm gradients and loss sums.
template <int HEAD>
__device__ __forceinline__ void gru_head_backward(f32x16 (&dh)[2], const GruBwdArgs &p, const float *lds, const GruLds &m, float *tN, float *tZ,
                                                  const LossScales &ls, const float *__restrict__ hsT, int c, bool ok, int64_t brow, int lane,
                                                  int l31, int half, f32x16 (&gWh)[2], float &gBh, float &gNw, float &gNb, double (&lacc)[4]) {
  const int A = p.A;
  // ---- y = LayerNorm(h'_t) -> head -> loss gradient at the head ----
  f32x16 hs[2];
  load_fm(hs, hsT, p.Nc, c, ok, half);
  float mean, rstd;
  ln_stats(hs, mean, rstd);
  f32x16 xh[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 16; ++r) xh[tt][r] = (hs[tt][r] - mean) * rstd;
  regs_to_tile64(tN, xh, l31, half);
  wave_lds_sync();
  const f32x16 z = gru_head(lds, m, tN, l31, half);
  if (HEAD == 1) {
    head_to_tile(tZ, z, A, l31, half);
    wave_lds_sync();
    if (lane < TS) {
      float *zl = tZ + lane * TP;
      if (ok) {
        const uint32_t dead = p.avail ? avail_dead_mask(p.avail + brow * A, A) : 0u;
        actor_loss_lane(zl, A, dead, (int)p.actions[brow], p.old_logp[brow], p.adv[brow], p.active[brow], p.cfg, ls.scale_pi, lacc);
      } else {
        for (int a = 0; a < A; ++a) zl[a] = 0.f;
      }
    }
  } else if (lane < TS) {
    float dvv = 0.f;
    if (ok) dvv = critic_loss_lane(z[0], p.v_old[brow], p.returns[brow], p.active[brow], p.cfg, ls, lacc);
    tZ[lane * TP] = dvv;
  }
  wave_lds_sync();
  // ---- head weight / bias gradients and d y = Wh^T dz ----
  f32x16 dH[2];
  {
    const float *sG = lds + m.nw, *sBt = lds + m.nb;
    const float g0 = sG[l31], c0 = sBt[l31], g1 = sG[32 + l31], c1 = sBt[32 + l31];
    float bsum = 0.f;
#pragma unroll 4
    for (int ss = 0; ss < TS / 2; ++ss) {
      const int s = 2 * ss + half;
      const float av = (l31 < A) ? tZ[s * TP + l31] : 0.f;
      bsum += av;
      gWh[0] = mfma(av, tN[l31 * TP + s] * g0 + c0, gWh[0]);
      gWh[1] = mfma(av, tN[(32 + l31) * TP + s] * g1 + c1, gWh[1]);
    }
    gBh += xhalf_sum(bsum);
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dH[tt][r] = 0.f;
    const float *sW = lds + m.wh;
    for (int kk = 0; kk < (A + 1) / 2; ++kk) {
      const int a = 2 * kk + half;
      const float b = (a < A) ? tZ[l31 * TP + a] : 0.f;
      dH[0] = mfma(sW[l31 * HP + a], b, dH[0]);
      dH[1] = mfma(sW[(32 + l31) * HP + a], b, dH[1]);
    }
  }
  wave_lds_sync();
  // ---- LayerNorm backward (rnn.norm): row sums through the tN tile, then d h' ----
  regs_to_tile64(tN, dH, l31, half);
  wave_lds_sync();
  { float s0 = 0.f; for (int j = 0; j < TS; ++j) s0 += tN[lane * TP + j]; gNb += s0; }
  wave_lds_sync();
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = 32 * tt + ROWMAP(r, half);
      tN[f * TP + l31] = dH[tt][r] * xh[tt][r];
      const float dxh = dH[tt][r] * lds[m.nw + f];
      dH[tt][r] = dxh;
      m1 += dxh; m2 += dxh * xh[tt][r];
    }
  wave_lds_sync();
  { float s0 = 0.f; for (int j = 0; j < TS; ++j) s0 += tN[lane * TP + j]; gNw += s0; }
  wave_lds_sync();
  m1 = xhalf_sum(m1) * (1.f / HID);
  m2 = xhalf_sum(m2) * (1.f / HID);
  // LN backward: d h'_t, the recurrent term is added by the caller
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dh[tt][r] = rstd * (dH[tt][r] - m1 - xh[tt][r] * m2);
}

// ---- training path, split three ways -------------------------------------------------------------------
// The sequential kernels above give one wave 32 sequences and all 64 hidden features: at BASELINE config 2 that is 240
// lone waves, each paying every load latency, MFMA chain and store burst of a step back to back.  Training therefore runs
//   gru_head_bwd_kernel   the row-local half of the backward (LayerNorm, head, PPO loss, LayerNorm backward): no
//                         recurrence in it, so it runs over all L x Nc rows at once and leaves d h'_t (without the
//                         recurrent term) where the forward stored h'_t;
//   gru_fwd_train2_kernel / gru_cell_bwd2_kernel
//                         the recurrences proper, a workgroup of TWO waves per 32 sequences: wave w owns hidden features
//                         [32 w, 32 w + 32) — its gate rows, its half of the state, half of every load and store, half of
//                         the MFMAs.  The waves meet once per step, at the LDS tile that holds the step's B operand
//                         (h_{t-1} * mask, or the step's d gates); the tile is double-buffered, so one barrier per step.
// Accumulation orders are those of the one-wave kernels: same values bit for bit (up to the order of the slab sums).

// Workgroup barrier for LDS traffic only.  __syncthreads() also drains the vector-memory queue (s_waitcnt vmcnt(0)): inside the
// step loops that would wait for the operands prefetched for the NEXT step and for this step's stores — a memory latency per step.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct SeqLds { int whh, bhh, tiles, tile_stride, total; };
__host__ __device__ inline SeqLds seq2_lds(int tile_rows, int n_buf) {
  SeqLds m;
  int p = 0;
  m.whh = p; p = al4(p + HID * GS);
  m.bhh = p; p += NG;
  m.tiles = p;
  m.tile_stride = al4(tile_rows * TP);
  p += n_buf * m.tile_stride;
  m.total = p;
  return m;
}

// 16 registers of one wave's feature half: feature 32 w + ROWMAP(r, half) of sequence column `col`
__device__ __forceinline__ void load_fm1(f32x16 &v, const float *__restrict__ src, int64_t ld, int64_t col, int w, int half) {
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = src[(int64_t)(32 * w + ROWMAP(r, half)) * ld + col];
}
__device__ __forceinline__ void store_fm1(float *__restrict__ dst, int64_t ld, int64_t col, const f32x16 &v, bool ok, int w, int half) {
  if (!ok) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) dst[(int64_t)(32 * w + ROWMAP(r, half)) * ld + col] = v[r];
}
__device__ __forceinline__ void regs_to_tile1(float *tile, const f32x16 &v, int w, int l31, int half) {
#pragma unroll
  for (int r = 0; r < 16; ++r) tile[(32 * w + ROWMAP(r, half)) * TP + l31] = v[r];
}

// forward recurrence of the training pass (input gates precomputed, all per-step gate values stored for the backward)
__global__ __launch_bounds__(128, 1) void gru_fwd_train2_kernel(GruFwdArgs p, SeqLds m) {
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & (WAVE - 1), w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)), l31 = lane & 31, half = lane >> 5;
  stage_gru_weight(lds + m.whh, p.params + p.off.gru_whh);
  for (int e = threadIdx.x; e < NG; e += blockDim.x) lds[m.bhh + e] = p.params[p.off.gru_bhh + e];
  __syncthreads();
  const int64_t B = (int64_t)p.L * p.Nc;
  const int64_t comp = (int64_t)p.L * HID * p.Nc;
  const int n_tiles = (p.Nc + TS - 1) / TS;
  f32x16 b_r, b_z, b_n;                                  // b_hh of this wave's gate rows
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int f = 32 * w + ROWMAP(r, half);
    b_r[r] = lds[m.bhh + f]; b_z[r] = lds[m.bhh + HID + f]; b_n[r] = lds[m.bhh + 2 * HID + f];
  }
  const float *sH = lds + m.whh + 32 * w + l31;          // A operand: W_hh[g = gate*64 + 32 w + l31][k] at sH[k*GS + gate*64]
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int c = tile * TS + l31;
    const bool ok = c < p.Nc;
    const int cc = ok ? c : 0;
    f32x16 h;
    {
      const int64_t row = p.h0_rows ? (int64_t)p.h0_rows[cc] : (int64_t)cc;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(p.h0 + row * HID + 32 * w + 8 * q + 4 * half);
        h[4 * q + 0] = v.x; h[4 * q + 1] = v.y; h[4 * q + 2] = v.z; h[4 * q + 3] = v.w;
      }
    }
    f32x16 gi_r, gi_z, gi_n;
    load_fm1(gi_r, p.giT, B, cc, w, half);
    load_fm1(gi_z, p.giT + (int64_t)HID * B, B, cc, w, half);
    load_fm1(gi_n, p.giT + (int64_t)2 * HID * B, B, cc, w, half);
    // masks[rows[.]] is a chain of two dependent loads: the row index runs two steps ahead and the mask one, so that no step
    // waits on a load it has just issued (such a wait would also drain the previous step's stores: vmcnt counts in order)
    // (the index stays a 32-bit register until it is used: a widening right after the load would wait for it)
    float mk_next = p.masks[p.rows ? (int64_t)p.rows[cc] : (int64_t)cc];
    int r_next = min(1, p.L - 1) * p.Nc + cc;
    if (p.rows) r_next = p.rows[r_next];
    for (int t = 0; t < p.L; ++t) {
      const float mk = ok ? mk_next : 0.f;
      float *tH = lds + m.tiles + (t & 1) * m.tile_stride;      // [64][TP] h_{t-1} * mask
      f32x16 hm;
#pragma unroll
      for (int r = 0; r < 16; ++r) hm[r] = h[r] * mk;
      regs_to_tile1(tH, hm, w, l31, half);
      // next step's input gates: in flight under this step's products (the last step re-reads its own)
      const int64_t ncol = (int64_t)min(t + 1, p.L - 1) * p.Nc + cc;
      mk_next = p.masks[(int64_t)r_next];
      r_next = min(t + 2, p.L - 1) * p.Nc + cc;
      if (p.rows) r_next = p.rows[r_next];
      f32x16 nx_r, nx_z, nx_n;
      load_fm1(nx_r, p.giT, B, ncol, w, half);
      load_fm1(nx_z, p.giT + (int64_t)HID * B, B, ncol, w, half);
      load_fm1(nx_n, p.giT + (int64_t)2 * HID * B, B, ncol, w, half);
      lds_barrier();
      f32x16 ar, az, ahn = b_n;
#pragma unroll
      for (int r = 0; r < 16; ++r) { ar[r] = gi_r[r] + b_r[r]; az[r] = gi_z[r] + b_z[r]; }
#pragma unroll 4
      for (int kk = 0; kk < HID / 2; ++kk) {
        const int k = 2 * kk + half;
        const float bh = tH[k * TP + l31];
        ar = mfma(sH[k * GS], bh, ar);
        az = mfma(sH[k * GS + HID], bh, az);
        ahn = mfma(sH[k * GS + 2 * HID], bh, ahn);
      }
      f32x16 gr, gz, gn;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        gr[r] = sigmoidf_(ar[r]);
        gz[r] = sigmoidf_(az[r]);
        gn[r] = tanhf_(gi_n[r] + gr[r] * ahn[r]);
        h[r] = (1.f - gz[r]) * gn[r] + gz[r] * hm[r];
      }
      float *base = p.scratch + (int64_t)t * HID * p.Nc;
      store_fm1(base + SCR_HM * comp, p.Nc, c, hm, ok, w, half);
      store_fm1(base + SCR_R * comp, p.Nc, c, gr, ok, w, half);
      store_fm1(base + SCR_Z * comp, p.Nc, c, gz, ok, w, half);
      store_fm1(base + SCR_N * comp, p.Nc, c, gn, ok, w, half);
      store_fm1(base + SCR_GHN * comp, p.Nc, c, ahn, ok, w, half);
      store_fm1(base + SCR_HS * comp, p.Nc, c, h, ok, w, half);
      gi_r = nx_r; gi_z = nx_z; gi_n = nx_n;
    }
    if (p.h_last && ok) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4 *>(p.h_last + (int64_t)c * HID + 32 * w + 8 * q + 4 * half) =
            make_float4(h[4 * q + 0], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
    }
    __syncthreads();                                     // the next tile's first step may reuse the buffer read last
  }
}

// One rollout / evaluation step (L = 1) with head, two waves per 32 rows: wave w computes the gates of hidden features
// [32 w, 32 w + 32) from x (B operand straight from feature-major HBM) and h * mask (LDS tile written by both waves) — 192 of
// the step's 384 MFMAs, in six independent chains.  LayerNorm(64) joins the two halves' (mean, M2) (Chan); the head is split
// over k (each wave its own 32 normalised features), wave 0 adds the partner's partial logits and samples / writes the output.
#define STEP2_PAIR_FLOATS (HID * TP + 3 * TS * TP + 128)     // tHm (later the partial-logit exchange) | tN x 2 | tZ | LN stats
template <int HM>
__device__ __forceinline__ void gru_step2_body(const GruFwdArgs &p, float *lds, const int bid, const int nb) {
  const GruLds &m = p.map;
  const int lane = threadIdx.x & (WAVE - 1), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)), l31 = lane & 31, half = lane >> 5;
  const int pair = wv >> 1, w = wv & 1;
  if (blockDim.x == 256 && ((((uintptr_t)p.params) | (uintptr_t)(4 * p.off.gru_wih) | (uintptr_t)(4 * p.off.gru_whh) | (uintptr_t)(4 * p.off.wh)) & 15) == 0)
    stage_gru_step_1shot(lds, m, p.params, p.off, p.A);
  else
    stage_gru_all(lds, m, p.params, p.off, p.A, true, true);
  __syncthreads();
  float *pb = lds + m.tiles + pair * STEP2_PAIR_FLOATS;
  float *tHm = pb, *tX = pb;                              // [64][TP] h * mask ; after the products: [16][64] partial logits of wave 1
  float *tN = pb + HID * TP + w * TS * TP;                // [32][TP] this wave's normalised features (B operand of the head)
  float *tZ = pb + HID * TP + 2 * TS * TP;                // [32][TP] logits [s][a]
  float *st = tZ + TS * TP;                               // [2][2][32] LayerNorm partial statistics
  const int64_t B = p.Nc;                                 // L == 1: column = sequence
  const int n_tiles = (p.Nc + TS - 1) / TS;
  const float *sI = lds + m.wih + 32 * w + l31, *sH = lds + m.whh + 32 * w + l31;
  for (int t0 = bid * 2; t0 < n_tiles; t0 += nb * 2) {                      // both pairs make every trip (workgroup barriers inside)
    const int tile = t0 + pair;
    const int c = tile * TS + l31;
    const bool ok = tile < n_tiles && c < p.Nc;
    const int cc = ok ? c : 0;
    const int n_valid = tile < n_tiles ? min(TS, p.Nc - tile * TS) : 0;
    const int64_t hrow = p.h0_rows ? (int64_t)p.h0_rows[cc] : (int64_t)cc;
    const float mk = ok ? p.masks[p.rows ? (int64_t)p.rows[cc] : (int64_t)cc] : 0.f;
    uint32_t dead = 0u;                                   // the sampling lane's availability mask, in flight under the products
    if (HM == 2 && p.avail) {
      const int64_t arow = n_valid > 0 ? (int64_t)tile * TS + min(lane, n_valid - 1) : 0;       // a pair without a tile reads row 0
      dead = avail_dead_mask(p.avail + arow * p.A, p.A);
    }
    float bx[HID / 2];
#pragma unroll
    for (int kk = 0; kk < HID / 2; ++kk) bx[kk] = p.xT[(int64_t)(2 * kk + half) * B + cc];
    f32x16 hm;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4 *>(p.h0 + hrow * HID + 32 * w + 8 * q + 4 * half);
      hm[4 * q + 0] = v.x * mk; hm[4 * q + 1] = v.y * mk; hm[4 * q + 2] = v.z * mk; hm[4 * q + 3] = v.w * mk;
    }
    regs_to_tile1(tHm, hm, w, l31, half);
    f32x16 ar, az, ain, ahn;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = 32 * w + ROWMAP(r, half);
      ar[r] = lds[m.bih + f] + lds[m.bhh + f];
      az[r] = lds[m.bih + HID + f] + lds[m.bhh + HID + f];
      ain[r] = lds[m.bih + 2 * HID + f];
      ahn[r] = lds[m.bhh + 2 * HID + f];
    }
    lds_barrier();
#pragma unroll 4
    for (int kk = 0; kk < HID / 2; ++kk) {
      const int k = 2 * kk + half;
      const float bxx = ok ? bx[kk] : 0.f, bh = tHm[k * TP + l31];
      ar = mfma(sI[k * GS], bxx, ar);
      az = mfma(sI[k * GS + HID], bxx, az);
      ain = mfma(sI[k * GS + 2 * HID], bxx, ain);
      ar = mfma(sH[k * GS], bh, ar);
      az = mfma(sH[k * GS + HID], bh, az);
      ahn = mfma(sH[k * GS + 2 * HID], bh, ahn);
    }
    f32x16 h;
    float s1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float gr = sigmoidf_(ar[r]), gz = sigmoidf_(az[r]);
      const float gn = tanhf_(ain[r] + gr * ahn[r]);
      h[r] = (1.f - gz) * gn + gz * hm[r];
      s1 += h[r];
    }
    if (p.h_last && ok) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4 *>(p.h_last + (int64_t)c * HID + 32 * w + 8 * q + 4 * half) =
            make_float4(h[4 * q + 0], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
    }
    // ---- LayerNorm(64): this wave's (mean, M2) over its 32 features, joined with the partner's ----
    const float mean_w = xhalf_sum(s1) * (1.f / 32.f);
    float q2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float d = h[r] - mean_w; q2 += d * d; }
    q2 = xhalf_sum(q2);
    if (half == 0) { st[w * 64 + l31] = mean_w; st[w * 64 + 32 + l31] = q2; }
    lds_barrier();
    const float mean_o = st[(1 - w) * 64 + l31], q2_o = st[(1 - w) * 64 + 32 + l31];
    const float mean = 0.5f * (mean_w + mean_o), dm = mean_w - mean_o;
    const float rstd = 1.0f / sqrtf((q2 + q2_o + dm * dm * 16.f) * (1.f / HID) + LN_EPS);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int fl = ROWMAP(r, half);
      tN[fl * TP + l31] = (h[r] - mean) * rstd * lds[m.nw + 32 * w + fl] + lds[m.nb + 32 * w + fl];
    }
    wave_lds_sync();
    // ---- head, split over k: partial logits of this wave's 32 features ----
    f32x16 z;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b = *reinterpret_cast<const float4 *>(lds + m.bh + 8 * q + 4 * half);
      z[4 * q + 0] = w == 0 ? b.x : 0.f; z[4 * q + 1] = w == 0 ? b.y : 0.f; z[4 * q + 2] = w == 0 ? b.z : 0.f; z[4 * q + 3] = w == 0 ? b.w : 0.f;
    }
    {
      const float *sW = lds + m.wh + 32 * w * HP;
#pragma unroll 8
      for (int kk = 0; kk < TS / 2; ++kk) {
        const int kl = 2 * kk + half;
        z = mfma(sW[kl * HP + l31], tN[kl * TP + l31], z);
      }
    }
    if (w == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) tX[r * WAVE + lane] = z[r];
    }
    lds_barrier();
    if (w == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] += tX[r * WAVE + lane];
      head_to_tile(tZ, z, p.A, l31, half);
      wave_lds_sync();
      const int64_t row0 = (int64_t)tile * TS;
      if (HM == 1) {
        for (int e = lane; e < n_valid * p.A; e += WAVE) {
          const int sidx = e / p.A, a = e - sidx * p.A;
          p.out[row0 * p.A + e] = tZ[sidx * TP + a];
        }
      } else if (lane < n_valid) {
        const int64_t i = row0 + lane;
        const uint64_t ctr = p.counter + (p.counter_dev ? *p.counter_dev : 0ull);
        float action, logp;
        categorical_act_mask(tZ + lane * TP, p.A, dead, p.deterministic != 0, p.seed, ctr, (uint64_t)i, action, logp);
        p.actions[i] = action;
        p.logp[i] = logp;
      }
    }
    lds_barrier();                                        // the tiles are rewritten by the next trip
  }
}
template <int HM>
__global__ __launch_bounds__(256, 1) void gru_step3_kernel(GruFwdArgs p) {
  __shared__ Step3Shared sh;
  gru_step3_body<HM>(p, sh, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256, 1) void gru_step3_dual_kernel(GruFwdArgs a, GruFwdArgs c, int nA) {
  __shared__ Step3Shared sh;
  if ((int)blockIdx.x < nA) gru_step3_body<2>(a, sh, blockIdx.x, nA);
  else gru_step3_body<1>(c, sh, blockIdx.x - nA, gridDim.x - nA);
}

// trunk + GRU step + head of a recurrent actor AND critic (narrow inputs) in ONE launch
template <int TR, int TLN>
__global__ __launch_bounds__(256, 1) void gru_step3f_dual_kernel(GruFwdArgs a, GruFwdArgs c, int nA) {
  __shared__ Step3Shared sh;
  if ((int)blockIdx.x < nA) gru_step3_body<2, TR, TLN>(a, sh, blockIdx.x, nA);
  else gru_step3_body<1, TR, TLN>(c, sh, blockIdx.x - nA, gridDim.x - nA);
}

template <int HM>
__global__ __launch_bounds__(256, 1) void gru_step2_kernel(GruFwdArgs p) {
  extern __shared__ __align__(16) float lds[];
  gru_step2_body<HM>(p, lds, blockIdx.x, gridDim.x);
}
// the rollout step of BOTH networks in one launch: workgroups [0, nA) sample the actor's actions, the rest write the critic's
// values — one queue, no fork / join between the two networks' kernels (their latency was half of a rollout step)
__global__ __launch_bounds__(256, 1) void gru_step2_dual_kernel(GruFwdArgs a, GruFwdArgs c, int nA) {
  extern __shared__ __align__(16) float lds[];
  if ((int)blockIdx.x < nA) gru_step2_body<2>(a, lds, blockIdx.x, nA);
  else gru_step2_body<1>(c, lds, blockIdx.x - nA, gridDim.x - nA);
}

// row-local half of the backward over all L x Nc rows; d h' replaces h' in the forward's scratch (component SCR_HS)
#define HEAD_BWD_WAVES 4
template <int HEAD>
__global__ __launch_bounds__(WAVE * HEAD_BWD_WAVES, 1) void gru_head_bwd_kernel(GruBwdArgs p, float *dhT) {
  extern __shared__ __align__(16) float lds[];
  __shared__ double red_smem[16 * 4];
  const GruLds &m = p.map;
  const NetOff &o = p.off;
  const int n_waves = blockDim.x / WAVE;
  const int lane = threadIdx.x & (WAVE - 1), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)), l31 = lane & 31, half = lane >> 5;
  const int A = p.A;
  {
    const int nthr = blockDim.x, tid = threadIdx.x;
    for (int e = tid; e < HID; e += nthr) { lds[m.nw + e] = p.params[o.rn_w + e]; lds[m.nb + e] = p.params[o.rn_b + e]; }
    for (int e = tid; e < 32; e += nthr) lds[m.bh + e] = e < A ? p.params[o.bh + e] : 0.f;
    for (int e = tid; e < HID * 32; e += nthr) {
      const int a = e >> 6, k = e & 63;
      lds[m.wh + k * HP + a] = (a < A) ? p.params[o.wh + a * HID + k] : 0.f;
    }
  }
  __syncthreads();
  float *tN = lds + m.tiles + wave * m.wave_stride;      // [64][TP]
  float *tZ = tN + HID * TP;                             // [32][TP]
  const int64_t comp = (int64_t)p.L * HID * p.Nc;
  const LossScales ls = loss_scales(p.cfg, p.mb_moments, p.vn_state);
  double lacc[4] = {0.0, 0.0, 0.0, 0.0};
  f32x16 gWh[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) gWh[i][r] = 0.f;
  float gBh = 0.f, gNw = 0.f, gNb = 0.f;
  const int n_ct = (p.Nc + TS - 1) / TS, n_tiles = p.L * n_ct;
  for (int tile = blockIdx.x * n_waves + wave; tile < n_tiles; tile += gridDim.x * n_waves) {
    const int t = tile / n_ct, c = (tile - t * n_ct) * TS + l31;
    const bool ok = c < p.Nc;
    const int64_t col = (int64_t)t * p.Nc + c;
    const int64_t brow = ok ? (p.rows ? (int64_t)p.rows[col] : col) : 0;
    const int64_t hs_off = SCR_HS * comp + (int64_t)t * HID * p.Nc;
    f32x16 dh[2];
    gru_head_backward<HEAD>(dh, p, lds, m, tN, tZ, ls, p.scratch + hs_off, c, ok, brow, lane, l31, half, gWh, gBh, gNw, gNb, lacc);
    store_fm(dhT + hs_off, p.Nc, c, dh, ok, half);
  }
  // ---- loss partial sums; the waves' head / rnn.norm gradients meet in LDS, wave 0 writes the workgroup's slab ----
  block_sum<4>(lacc, red_smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) p.partials[(size_t)blockIdx.x * 4 + k] = lacc[k];
  }
  __syncthreads();
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int r = 0; r < 16; ++r) tN[(16 * tj + r) * WAVE + lane] = gWh[tj][r];
  tZ[lane] = gBh; tZ[WAVE + lane] = gNw; tZ[2 * WAVE + lane] = gNb;
  __syncthreads();
  if (wave != 0) return;
  for (int ww = 1; ww < n_waves; ++ww) {
    const float *oN = lds + m.tiles + ww * m.wave_stride, *oZ = oN + HID * TP;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) gWh[tj][r] += oN[(16 * tj + r) * WAVE + lane];
    gBh += oZ[lane]; gNw += oZ[WAVE + lane]; gNb += oZ[2 * WAVE + lane];
  }
  float *slab = p.slabs + (size_t)blockIdx.x * p.slab_stride + p.slab_col0;
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int a = ROWMAP(r, half);
      if (a < A) slab[o.wh + a * HID + 32 * tj + l31] = gWh[tj][r];
    }
  if (half == 0 && l31 < A) slab[o.bh + l31] = gBh;
  slab[o.rn_w + lane] = gNw;
  slab[o.rn_b + lane] = gNb;
}

// backward recurrence: d h'_t (from gru_head_bwd_kernel) + carry -> d gates -> carry = (W_hh^T d gh + d h' z) * mask
__global__ __launch_bounds__(128, 1) void gru_cell_bwd2_kernel(GruBwdArgs p, SeqLds m) {
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & (WAVE - 1), w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)), l31 = lane & 31, half = lane >> 5;
  stage_gru_weight(lds + m.whh, p.params + p.off.gru_whh);
  __syncthreads();
  const int64_t B = (int64_t)p.L * p.Nc;
  const int64_t comp = (int64_t)p.L * HID * p.Nc;
  const int n_tiles = (p.Nc + TS - 1) / TS;
  const float *sH = lds + m.whh + (32 * w + l31) * GS;   // A operand of W_hh^T: W_hh[g][k = 32 w + l31] at sH[g]
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int c = tile * TS + l31;
    const bool ok = c < p.Nc;
    const int cc = ok ? c : 0;
    f32x16 carry;
#pragma unroll
    for (int r = 0; r < 16; ++r) carry[r] = 0.f;
    f32x16 dh, hm, gr, gz, gn, ghn;
    {
      const float *sb = p.scratch + (int64_t)(p.L - 1) * HID * p.Nc;
      load_fm1(dh, sb + SCR_HS * comp, p.Nc, cc, w, half);
      load_fm1(hm, sb + SCR_HM * comp, p.Nc, cc, w, half);
      load_fm1(gr, sb + SCR_R * comp, p.Nc, cc, w, half);
      load_fm1(gz, sb + SCR_Z * comp, p.Nc, cc, w, half);
      load_fm1(gn, sb + SCR_N * comp, p.Nc, cc, w, half);
      load_fm1(ghn, sb + SCR_GHN * comp, p.Nc, cc, w, half);
    }
    // masks[rows[.]]: row index two steps ahead, mask one step ahead (see gru_fwd_train2_kernel)
    float mk_next;
    int r_next;
    {
      const int64_t c0 = (int64_t)(p.L - 1) * p.Nc + cc;
      mk_next = p.masks[p.rows ? (int64_t)p.rows[c0] : c0];
      r_next = max(p.L - 2, 0) * p.Nc + cc;
      if (p.rows) r_next = p.rows[r_next];
    }
    for (int t = p.L - 1; t >= 0; --t) {
      const int64_t col = (int64_t)t * p.Nc + cc;
      const float mk = ok ? mk_next : 0.f;
      mk_next = p.masks[(int64_t)r_next];
      r_next = max(t - 2, 0) * p.Nc + cc;
      if (p.rows) r_next = p.rows[r_next];
      float *tG = lds + m.tiles;                                // [192][TP] d gh of the step (r, z, n rows), single-buffered
      // the previous step's values: in flight under this step's work (step 0 re-reads its own)
      f32x16 n_dh, n_hm, n_r, n_z, n_n, n_ghn;
      {
        const float *sb = p.scratch + (int64_t)max(t - 1, 0) * HID * p.Nc;
        load_fm1(n_dh, sb + SCR_HS * comp, p.Nc, cc, w, half);
        load_fm1(n_hm, sb + SCR_HM * comp, p.Nc, cc, w, half);
        load_fm1(n_r, sb + SCR_R * comp, p.Nc, cc, w, half);
        load_fm1(n_z, sb + SCR_Z * comp, p.Nc, cc, w, half);
        load_fm1(n_n, sb + SCR_N * comp, p.Nc, cc, w, half);
        load_fm1(n_ghn, sb + SCR_GHN * comp, p.Nc, cc, w, half);
      }
      f32x16 d_r, d_z, d_n, d_hn;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float dhh = (ok ? dh[r] : 0.f) + carry[r], zz = gz[r], nn = gn[r], rr = gr[r];
        dh[r] = dhh;
        const float dn_pre = dhh * (1.f - zz) * (1.f - nn * nn);
        d_n[r] = dn_pre;
        d_hn[r] = dn_pre * rr;
        d_r[r] = dn_pre * ghn[r] * rr * (1.f - rr);
        d_z[r] = dhh * (hm[r] - nn) * zz * (1.f - zz);
      }
      regs_to_tile1(tG, d_r, w, l31, half);
      regs_to_tile1(tG + HID * TP, d_z, w, l31, half);
      regs_to_tile1(tG + 2 * HID * TP, d_hn, w, l31, half);
      store_fm1(p.dgiT, B, col, d_r, ok, w, half);
      store_fm1(p.dgiT + (int64_t)HID * B, B, col, d_z, ok, w, half);
      store_fm1(p.dgiT + (int64_t)2 * HID * B, B, col, d_n, ok, w, half);
      store_fm1(p.dghnT, B, col, d_hn, ok, w, half);
      lds_barrier();
      f32x16 dhm, dhm2;                                  // two independent MFMA chains
#pragma unroll
      for (int r = 0; r < 16; ++r) { dhm[r] = 0.f; dhm2[r] = 0.f; }
#pragma unroll 4
      for (int gg = 0; gg < NG / 2; gg += 2) {
        const int g = 2 * gg + half;
        dhm = mfma(sH[g], tG[g * TP + l31], dhm);
        dhm2 = mfma(sH[g + 2], tG[(g + 2) * TP + l31], dhm2);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) carry[r] = (dhm[r] + dhm2[r] + dh[r] * gz[r]) * mk;
      dh = n_dh; hm = n_hm; gr = n_r; gz = n_z; gn = n_n; ghn = n_ghn;
      lds_barrier();                                     // both waves are done reading the tile
    }
  }
}

// ---- weight-gradient kernel ---------------------------------------------------------------------------
// Workgroup = 4 waves = 4 roles over the same row tiles: role = (matrix ih|hh) x (gate rows 0..95 | 96..191).
// dW[g][k] = sum_rows dG[g][row] * in[k][row]:  A operand = dG tile read transposed (lanes <-> g), B = input tile.
struct GruWgArgs {
  NetOff off;
  const float *xT;            // [64][B]
  const float *scratch;       // hm at component SCR_HM: [L][64][Nc]
  const float *dgiT, *dghnT;
  int L, Nc;
  float *slabs;
  int64_t slab_stride, slab_col0;
};

__global__ __launch_bounds__(256, 1) void gru_wgrad_kernel(GruWgArgs p) {
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & (WAVE - 1), role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)), l31 = lane & 31, half = lane >> 5;
  const int mat = role >> 1, ghalf = role & 1;              // mat 0: W_ih (x), 1: W_hh (hm)
  float *tA = lds + role * ((96 + HID) * TP);                // [96][TP] dG rows of this role
  float *tB = tA + 96 * TP;                                  // [64][TP] input tile
  const int64_t B = (int64_t)p.L * p.Nc;
  f32x16 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bacc0 = 0.f, bacc1 = 0.f;                            // bias grads: rows lane and 64 + (lane & 31)
  const int ct = (p.Nc + TS - 1) / TS;
  const int n_tiles = p.L * ct;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int t = tile / ct, c0 = (tile - t * ct) * TS;
    const int nv = min(TS, p.Nc - c0);
    const int64_t col0 = (int64_t)t * p.Nc + c0;
    // stage: 96 gate rows and 64 input rows, 32 columns each; a wave instruction covers two rows (2 x 128 B).  Loads are
    // unconditional (clamped column) and issued 16 at a time before their LDS stores: a predicated load would be waited
    // for on its own, which made this kernel latency-bound (80 serial global loads per tile).
    {
      const int s = lane & 31, sc = min(s, nv - 1), r2 = lane >> 5;
#pragma unroll
      for (int b0 = 0; b0 < 48; b0 += 16) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int gl = 2 * (b0 + j) + r2, g = 96 * ghalf + gl;
          const float *src = (mat == 1 && g >= 128) ? p.dghnT + (int64_t)(g - 128) * B : p.dgiT + (int64_t)g * B;
          v[j] = src[col0 + sc];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) tA[(2 * (b0 + j) + r2) * TP + s] = (s < nv) ? v[j] : 0.f;
      }
      const float *inb = (mat == 0) ? p.xT : p.scratch + (int64_t)SCR_HM * p.L * HID * p.Nc + (int64_t)t * HID * p.Nc;
      const int64_t ldi = (mat == 0) ? B : p.Nc;
      const int64_t ci = (mat == 0) ? col0 : c0;
#pragma unroll
      for (int b0 = 0; b0 < 32; b0 += 16) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = inb[(int64_t)(2 * (b0 + j) + r2) * ldi + ci + sc];
#pragma unroll
        for (int j = 0; j < 16; ++j) tB[(2 * (b0 + j) + r2) * TP + s] = (s < nv) ? v[j] : 0.f;
      }
    }
    wave_lds_sync();
#pragma unroll 2
    for (int ss = 0; ss < TS / 2; ++ss) {
      const int s = 2 * ss + half;
      const float b0 = tB[l31 * TP + s], b1 = tB[(32 + l31) * TP + s];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float a = tA[(32 * i + l31) * TP + s];
        acc[i][0] = mfma(a, b0, acc[i][0]);
        acc[i][1] = mfma(a, b1, acc[i][1]);
      }
    }
    { float s0 = 0.f; for (int j = 0; j < TS; ++j) s0 += tA[lane * TP + j]; bacc0 += s0; }
    if (lane < 32) { float s1 = 0.f; for (int j = 0; j < TS; ++j) s1 += tA[(64 + lane) * TP + j]; bacc1 += s1; }
    wave_lds_sync();
  }
  float *slab = p.slabs + (size_t)blockIdx.x * p.slab_stride + p.slab_col0;
  const int woff = (mat == 0) ? p.off.gru_wih : p.off.gru_whh, boff = (mat == 0) ? p.off.gru_bih : p.off.gru_bhh;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int g = 96 * ghalf + 32 * i + ROWMAP(r, half), k = 32 * j + l31;
        slab[woff + g * HID + k] = acc[i][j][r];
      }
  slab[boff + 96 * ghalf + lane] = bacc0;
  if (lane < 32) slab[boff + 96 * ghalf + 64 + lane] = bacc1;
}

// ---- input-side products, out of the sequential kernels --------------------------------------------------------------
// gi = W_ih x + b_ih has no dependence on time, and neither has d x = W_ih^T d gi: both run here as plain row-tile
// products over all B = L * Nc rows (every CU busy), so that the sequential forward / backward kernels carry only the
// W_hh half of the MFMA work and keep 49 KB of LDS free (two workgroups — actor's and critic's — fit a CU side by side).
struct GruInArgs {
  const float *params;
  NetOff off;
  const float *inT;           // gates: xT [64][B]      | backward: dgiT [192][B]
  float *outT;                // gates: giT [192][B]    | backward: dxT  [64][B]
  int64_t B;
};

__global__ __launch_bounds__(256, 1) void gru_gi_kernel(GruInArgs p) {
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & (WAVE - 1), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)), l31 = lane & 31, half = lane >> 5;
  const int n_waves = blockDim.x / WAVE;
  float *sI = lds, *sB = lds + al4(HID * GS);
  stage_gru_weight(sI, p.params + p.off.gru_wih);
  for (int e = threadIdx.x; e < NG; e += blockDim.x) sB[e] = p.params[p.off.gru_bih + e];
  __syncthreads();
  const int64_t n_tiles = (p.B + TS - 1) / TS;
  for (int64_t tile = (int64_t)blockIdx.x * n_waves + wave; tile < n_tiles; tile += (int64_t)gridDim.x * n_waves) {
    const int64_t col = tile * TS + l31;
    const bool ok = col < p.B;
    const int64_t cc = ok ? col : 0;
    float bx[HID / 2];
#pragma unroll
    for (int kk = 0; kk < HID / 2; ++kk) bx[kk] = p.inT[(int64_t)(2 * kk + half) * p.B + cc];
    f32x16 acc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b = *reinterpret_cast<const float4 *>(sB + 32 * t + 8 * q + 4 * half);
        acc[t][4 * q + 0] = b.x; acc[t][4 * q + 1] = b.y; acc[t][4 * q + 2] = b.z; acc[t][4 * q + 3] = b.w;
      }
#pragma unroll 4
    for (int kk = 0; kk < HID / 2; ++kk) {
      const int k = 2 * kk + half;
#pragma unroll
      for (int t = 0; t < 6; ++t) acc[t] = mfma(sI[k * GS + 32 * t + l31], bx[kk], acc[t]);
    }
    if (ok) {
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) p.outT[(int64_t)(32 * t + ROWMAP(r, half)) * p.B + col] = acc[t][r];
    }
  }
}

__global__ __launch_bounds__(256, 1) void gru_dx_kernel(GruInArgs p) {
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & (WAVE - 1), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)), l31 = lane & 31, half = lane >> 5;
  const int n_waves = blockDim.x / WAVE;
  float *sI = lds;
  stage_gru_weight(sI, p.params + p.off.gru_wih);
  __syncthreads();
  const int64_t n_tiles = (p.B + TS - 1) / TS;
  for (int64_t tile = (int64_t)blockIdx.x * n_waves + wave; tile < n_tiles; tile += (int64_t)gridDim.x * n_waves) {
    const int64_t col = tile * TS + l31;
    const bool ok = col < p.B;
    const int64_t cc = ok ? col : 0;
    f32x16 dx[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dx[tt][r] = 0.f;
    for (int g0 = 0; g0 < NG / 2; g0 += 32) {                 // 3 batches of 32 gate rows per half: loads first, then the MFMAs
      float b[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) b[j] = p.inT[(int64_t)(2 * (g0 + j) + half) * p.B + cc];
#pragma unroll 4
      for (int j = 0; j < 32; ++j) {
        const int g = 2 * (g0 + j) + half;
        dx[0] = mfma(sI[l31 * GS + g], b[j], dx[0]);
        dx[1] = mfma(sI[(32 + l31) * GS + g], b[j], dx[1]);
      }
    }
    if (ok) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) p.outT[(int64_t)(32 * tt + ROWMAP(r, half)) * p.B + col] = dx[tt][r];
    }
  }
}

// ---- host ---------------------------------------------------------------------------------------------
#define LDS_LIMIT (160 * 1024)
#define LDS_DYN_MAX (LDS_LIMIT - 1024)
#define NUM_CU 256

static int check_rec(const mappo_net_desc *d, const char *who) {
  MAPPO_REQUIRE(d && d->recurrent, "%s: needs a recurrent network descriptor", who);
  MAPPO_REQUIRE(d->hidden == HID, "%s: hidden_size %d unsupported", who, d->hidden);
  MAPPO_REQUIRE(d->out_dim >= 1 && d->out_dim <= MAPPO_MAX_ACTIONS, "%s: out_dim %d", who, d->out_dim);
  return MAPPO_OK;
}

template <typename K>
static int raise_lds(K kernel, const char *who) {
  hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_DYN_MAX);
  if (e != hipSuccess) { mappo_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e)); (void)hipGetLastError(); return MAPPO_ELAUNCH; }
  return MAPPO_OK;
}

extern "C" int64_t mappo_gru_scratch_floats(int32_t L, int32_t Nc) { return (int64_t)N_SCR * L * HID * Nc; }

extern "C" int mappo_gru_forward(const float *params, const mappo_net_desc *desc, const float *xT, const float *giT, const float *h0,
                                 const int32_t *h0_rows, const float *masks, const int32_t *rows, int32_t L, int32_t Nc,
                                 float *h_last, float *scratch, int32_t head_mode, float *out, const float *avail,
                                 int32_t deterministic, uint64_t seed, uint64_t counter, const uint64_t *counter_dev,
                                 float *actions, float *logp, mappo_stream_t stream) {
  if (int rc = check_rec(desc, "gru_forward")) return rc;
  MAPPO_REQUIRE(params && (xT || giT) && h0 && masks && L > 0 && Nc > 0, "gru_forward: bad arguments");
  MAPPO_REQUIRE(head_mode >= 0 && head_mode <= 2, "gru_forward: head_mode %d", head_mode);
  MAPPO_REQUIRE(head_mode != 1 || out, "gru_forward: out required");
  MAPPO_REQUIRE(head_mode != 2 || (actions && logp), "gru_forward: actions/logp required");
  MAPPO_CLEAR_STICKY();
  GruFwdArgs a = {};
  a.params = params; a.off = net_offsets(*desc); a.xT = xT; a.giT = giT; a.h0 = h0; a.h0_rows = h0_rows; a.masks = masks; a.rows = rows;
  a.L = L; a.Nc = Nc; a.A = desc->out_dim; a.head_mode = head_mode; a.h_last = h_last; a.scratch = scratch; a.out = out;
  a.avail = avail; a.actions = actions; a.logp = logp; a.deterministic = deterministic; a.seed = seed; a.counter = counter;
  a.counter_dev = counter_dev;
  const int n_tiles = (Nc + TS - 1) / TS;
  if (giT && head_mode == 0 && scratch) {                 // training pass: two waves per 32 sequences
    const SeqLds sm = seq2_lds(HID, 2);
    const size_t bytes = (size_t)sm.total * sizeof(float);
    static const int lds_rc_attr2 = raise_lds(gru_fwd_train2_kernel, "gru_forward");
    if (lds_rc_attr2) return lds_rc_attr2;
    hipLaunchKernelGGL(gru_fwd_train2_kernel, dim3(n_tiles < 4 * NUM_CU ? n_tiles : 4 * NUM_CU), dim3(2 * WAVE), bytes, as_stream(stream), a, sm);
    MAPPO_CHECK_LAUNCH("gru_forward");
    return MAPPO_OK;
  }
  if (!giT && !scratch && L == 1 && head_mode != 0 && !(getenv("MAPPO_GRU_STEP3") && getenv("MAPPO_GRU_STEP3")[0] == '0')) {
    // rollout / get_values step: one 16-row tile per 4-wave workgroup, hidden units split over the waves
    const int nt16 = (Nc + 15) / 16;
    const int g3 = nt16 < 2 * NUM_CU ? nt16 : 2 * NUM_CU;
    if (head_mode == 1) hipLaunchKernelGGL(gru_step3_kernel<1>, dim3(g3), dim3(4 * WAVE), 0, as_stream(stream), a);
    else hipLaunchKernelGGL(gru_step3_kernel<2>, dim3(g3), dim3(4 * WAVE), 0, as_stream(stream), a);
    MAPPO_CHECK_LAUNCH("gru_forward");
    return MAPPO_OK;
  }
  if (!giT && !scratch && L == 1 && head_mode != 0) {       // (MAPPO_GRU_STEP3=0) two waves per 32 rows
    a.map = gru_lds(0, 0, true);
    const size_t bytes = (size_t)(a.map.total + 2 * STEP2_PAIR_FLOATS) * sizeof(float);
    MAPPO_REQUIRE(bytes <= LDS_DYN_MAX, "gru_forward: needs %zu B of LDS", bytes);
    int nb2 = (n_tiles + 1) / 2;
    if (nb2 > NUM_CU) nb2 = NUM_CU;
    if (head_mode == 1) {
      static const int lds_rc_attr = raise_lds(gru_step2_kernel<1>, "gru_forward");
      if (lds_rc_attr) return lds_rc_attr;
      hipLaunchKernelGGL(gru_step2_kernel<1>, dim3(nb2), dim3(4 * WAVE), bytes, as_stream(stream), a);
    } else {
      static const int lds_rc_attr = raise_lds(gru_step2_kernel<2>, "gru_forward");
      if (lds_rc_attr) return lds_rc_attr;
      hipLaunchKernelGGL(gru_step2_kernel<2>, dim3(nb2), dim3(4 * WAVE), bytes, as_stream(stream), a);
    }
    MAPPO_CHECK_LAUNCH("gru_forward");
    return MAPPO_OK;
  }
  int nw = head_mode ? 2 : 4;
  while (nw > 1 && n_tiles < nw) nw >>= 1;
  const int wave_rows = head_mode ? (HID + HID + TS) : HID;
  a.map = gru_lds(nw, wave_rows, giT == nullptr);
  const size_t lds_bytes = (size_t)a.map.total * sizeof(float);
  MAPPO_REQUIRE(lds_bytes <= LDS_DYN_MAX, "gru_forward: needs %zu B of LDS", lds_bytes);
  int nb = (n_tiles + nw - 1) / nw;
  if (nb > NUM_CU) nb = NUM_CU;
  a.tile_waves = nw;
  // a rollout step (L = 1) is mostly the staging of 98 KB of GRU weights: four waves stage, `nw` of them (LDS budget) own tiles
  const int launch_waves = 4;
#define GRU_FWD(PRE, HM_)                                                                                         \
  do {                                                                                                            \
    static const int lds_rc_attr = raise_lds(gru_fwd_kernel<PRE, HM_>, "gru_forward");                                                                                     \
    if (lds_rc_attr) return lds_rc_attr;       \
    hipLaunchKernelGGL((gru_fwd_kernel<PRE, HM_>), dim3(nb), dim3(WAVE * launch_waves), lds_bytes, as_stream(stream), a); \
  } while (0)
  if (giT) { if (head_mode == 0) GRU_FWD(true, 0); else if (head_mode == 1) GRU_FWD(true, 1); else GRU_FWD(true, 2); }
  else { if (head_mode == 0) GRU_FWD(false, 0); else if (head_mode == 1) GRU_FWD(false, 1); else GRU_FWD(false, 2); }
#undef GRU_FWD
  MAPPO_CHECK_LAUNCH("gru_forward");
  return MAPPO_OK;
}

static int launch_gru_in(bool gates, const float *params, const mappo_net_desc *desc, const float *inT, int64_t B, float *outT,
                         hipStream_t st, const char *who);

extern "C" int mappo_gru_step_dual(const float *actor_params, const mappo_net_desc *actor_desc, const float *actor_featT,
                                   const float *actor_h0, float *actor_h_last, const float *critic_params,
                                   const mappo_net_desc *critic_desc, const float *critic_featT, const float *critic_h0,
                                   float *critic_h_last, const float *masks, int32_t Nc, const float *avail, int32_t deterministic,
                                   uint64_t seed, uint64_t counter, const uint64_t *counter_dev, float *actions, float *logp,
                                   float *values, mappo_stream_t stream) {
  if (int rc = check_rec(actor_desc, "gru_step_dual")) return rc;
  if (int rc = check_rec(critic_desc, "gru_step_dual")) return rc;
  MAPPO_REQUIRE(critic_desc->out_dim == 1, "gru_step_dual: critic out_dim must be 1");
  MAPPO_REQUIRE(actor_params && actor_featT && actor_h0 && critic_params && critic_featT && critic_h0 && masks && actions && logp &&
                values && Nc > 0, "gru_step_dual: bad arguments");
  MAPPO_CLEAR_STICKY();
  GruFwdArgs a = {}, c = {};
  a.params = actor_params; a.off = net_offsets(*actor_desc); a.xT = actor_featT; a.h0 = actor_h0; a.masks = masks; a.L = 1; a.Nc = Nc;
  a.A = actor_desc->out_dim; a.head_mode = 2; a.h_last = actor_h_last; a.avail = avail; a.actions = actions; a.logp = logp;
  a.deterministic = deterministic; a.seed = seed; a.counter = counter; a.counter_dev = counter_dev;
  c.params = critic_params; c.off = net_offsets(*critic_desc); c.xT = critic_featT; c.h0 = critic_h0; c.masks = masks; c.L = 1; c.Nc = Nc;
  c.A = 1; c.head_mode = 1; c.h_last = critic_h_last; c.out = values;
  if (!(getenv("MAPPO_GRU_STEP3") && getenv("MAPPO_GRU_STEP3")[0] == '0')) {
    const int nt16 = (Nc + 15) / 16;
    const int g3 = nt16 < NUM_CU ? nt16 : NUM_CU;
    hipLaunchKernelGGL(gru_step3_dual_kernel, dim3(2 * g3), dim3(4 * WAVE), 0, as_stream(stream), a, c, g3);
    MAPPO_CHECK_LAUNCH("gru_step_dual");
    return MAPPO_OK;
  }
  a.map = c.map = gru_lds(0, 0, true);
  const size_t bytes = (size_t)(a.map.total + 2 * STEP2_PAIR_FLOATS) * sizeof(float);
  MAPPO_REQUIRE(bytes <= LDS_DYN_MAX, "gru_step_dual: needs %zu B of LDS", bytes);
  const int n_tiles = (Nc + TS - 1) / TS;
  int nb = (n_tiles + 1) / 2;
  if (nb > NUM_CU / 2) nb = NUM_CU / 2;
  static const int lds_rc_attr = raise_lds(gru_step2_dual_kernel, "gru_step_dual");
  if (lds_rc_attr) return lds_rc_attr;
  hipLaunchKernelGGL(gru_step2_dual_kernel, dim3(2 * nb), dim3(4 * WAVE), bytes, as_stream(stream), a, c, nb);
  MAPPO_CHECK_LAUNCH("gru_step_dual");
  return MAPPO_OK;
}

int mappo_recurrent_step_dual_wide_(const float *actor_params, const mappo_net_desc *actor_desc, const float *obs, const float *actor_h0,
                                    float *actor_h_last, const float *critic_params, const mappo_net_desc *critic_desc, const float *share_obs,
                                    const float *critic_h0, float *critic_h_last, const float *masks, int32_t Nc, const float *avail,
                                    int32_t deterministic, uint64_t seed, uint64_t counter, const uint64_t *counter_dev, float *actions,
                                    float *logp, float *values, mappo_stream_t stream);      // mlp.hip
// One rollout step of a recurrent actor and critic, trunk included: obs / share_obs rows -> actions, log-probs,
// values, next states (r_actor_critic.py:43-70,146-165 for both networks on the same rows).
extern "C" int mappo_recurrent_step_dual(const float *actor_params, const mappo_net_desc *actor_desc, const float *obs,
                                         const float *actor_h0, float *actor_h_last, const float *critic_params,
                                         const mappo_net_desc *critic_desc, const float *share_obs, const float *critic_h0,
                                         float *critic_h_last, const float *masks, int32_t Nc, const float *avail, int32_t deterministic,
                                         uint64_t seed, uint64_t counter, const uint64_t *counter_dev, float *actions, float *logp,
                                         float *values, mappo_stream_t stream) {
  if (int rc = check_rec(actor_desc, "recurrent_step_dual")) return rc;
  if (int rc = check_rec(critic_desc, "recurrent_step_dual")) return rc;
  MAPPO_REQUIRE(critic_desc->out_dim == 1, "recurrent_step_dual: critic out_dim must be 1");
  MAPPO_REQUIRE(actor_params && obs && actor_h0 && critic_params && share_obs && critic_h0 && masks && actions && logp && values && Nc > 0,
                "recurrent_step_dual: bad arguments");
  MAPPO_REQUIRE(actor_desc->layer_N == critic_desc->layer_N && actor_desc->use_relu == critic_desc->use_relu,
                "recurrent_step_dual: the networks must share layer_N and the activation");
  if (actor_desc->in_dim > 64 && critic_desc->in_dim > 64)       // wide inputs: split-K trunks + GRU step in one launch (mlp_wide16.h)
    return mappo_recurrent_step_dual_wide_(actor_params, actor_desc, obs, actor_h0, actor_h_last, critic_params, critic_desc, share_obs, critic_h0,
                                           critic_h_last, masks, Nc, avail, deterministic, seed, counter, counter_dev, actions, logp, values, stream);
  MAPPO_REQUIRE(actor_desc->in_dim <= 64 && critic_desc->in_dim <= 64, "recurrent_step_dual: both networks narrow (in_dim <= 64) or both wide (65..512)");
  MAPPO_REQUIRE(actor_desc->layer_N <= 1, "recurrent_step_dual: narrow inputs: layer_N <= 1 (two hidden layers do not fit the register file)");
  MAPPO_CLEAR_STICKY();
  GruFwdArgs a = {}, c = {};
  a.params = actor_params; a.off = net_offsets(*actor_desc); a.x_rows = obs; a.desc = *actor_desc; a.h0 = actor_h0; a.masks = masks; a.L = 1; a.Nc = Nc;
  a.A = actor_desc->out_dim; a.head_mode = 2; a.h_last = actor_h_last; a.avail = avail; a.actions = actions; a.logp = logp;
  a.deterministic = deterministic; a.seed = seed; a.counter = counter; a.counter_dev = counter_dev;
  c.params = critic_params; c.off = net_offsets(*critic_desc); c.x_rows = share_obs; c.desc = *critic_desc; c.h0 = critic_h0; c.masks = masks; c.L = 1; c.Nc = Nc;
  c.A = 1; c.head_mode = 1; c.h_last = critic_h_last; c.out = values;
  const int nt16 = (Nc + 15) / 16;
  const int g3 = nt16 < NUM_CU ? nt16 : NUM_CU;
  const dim3 grid(2 * g3), block(4 * WAVE);
  hipStream_t st = as_stream(stream);
  const bool relu = actor_desc->use_relu != 0;
  switch (actor_desc->layer_N) {
    case 0: if (relu) hipLaunchKernelGGL((gru_step3f_dual_kernel<1, 0>), grid, block, 0, st, a, c, g3); else hipLaunchKernelGGL((gru_step3f_dual_kernel<2, 0>), grid, block, 0, st, a, c, g3); break;
    default: if (relu) hipLaunchKernelGGL((gru_step3f_dual_kernel<1, 1>), grid, block, 0, st, a, c, g3); else hipLaunchKernelGGL((gru_step3f_dual_kernel<2, 1>), grid, block, 0, st, a, c, g3); break;
  }
  MAPPO_CHECK_LAUNCH("recurrent_step_dual");
  return MAPPO_OK;
}

extern "C" int32_t mappo_gru_backward_slabs(int32_t Nc) {
  const int n_tiles = (Nc + TS - 1) / TS;
  return n_tiles < NUM_CU ? n_tiles : NUM_CU;
}

extern "C" int mappo_gru_backward(const float *params, const mappo_net_desc *desc, const float *scratch, const float *masks,
                                  const int32_t *rows, int32_t L, int32_t Nc, int32_t head, const float *avail,
                                  const float *actions, const float *old_logp, const float *adv, const float *active,
                                  const float *v_old, const float *returns, const float *vn_state, const double *mb_moments,
                                  const mappo_ppo_cfg *cfg, float *dxT, float *dgiT, float *dghnT, float *slabs,
                                  int64_t slab_stride, int64_t slab_col0, double *partials, mappo_stream_t stream) {
  if (int rc = check_rec(desc, "gru_backward")) return rc;
  MAPPO_REQUIRE(params && scratch && masks && active && mb_moments && cfg && dgiT && dghnT && slabs && partials && L > 0 && Nc > 0,
                "gru_backward: bad arguments");              // dxT == NULL: d x deferred to mappo_gru_input_backward
  MAPPO_REQUIRE(head == 1 || head == 2, "gru_backward: head %d", head);
  MAPPO_REQUIRE(head != 1 || (actions && old_logp && adv), "gru_backward: actor loss inputs");
  MAPPO_REQUIRE(head != 2 || (v_old && returns && (!cfg->use_valuenorm || vn_state)), "gru_backward: critic loss inputs");
  MAPPO_CLEAR_STICKY();
  GruBwdArgs a = {};
  a.params = params; a.off = net_offsets(*desc); a.scratch = scratch; a.masks = masks; a.rows = rows; a.L = L; a.Nc = Nc;
  a.A = desc->out_dim; a.head = head; a.avail = avail; a.actions = actions; a.old_logp = old_logp; a.adv = adv; a.active = active;
  a.v_old = v_old; a.returns = returns; a.vn_state = vn_state; a.mb_moments = mb_moments; a.cfg = *cfg; a.dxT = dxT; a.dgiT = dgiT;
  a.dghnT = dghnT; a.slabs = slabs; a.slab_stride = slab_stride; a.slab_col0 = slab_col0; a.partials = partials;
  MAPPO_REQUIRE(slab_col0 >= 0 && slab_col0 + a.off.total <= slab_stride, "gru_backward: slab column range");
  const int nb = mappo_gru_backward_slabs(Nc);
  {
    // the row-local half over all L x Nc rows (d h' replaces h' in the forward's scratch), then the recurrence
    const int n_ct = (Nc + TS - 1) / TS;
    const int64_t n_tiles = (int64_t)L * n_ct;
    int nw = (int)((n_tiles + nb - 1) / nb);
    if (nw > HEAD_BWD_WAVES) nw = HEAD_BWD_WAVES;
    GruLds hm = {};
    {
      int q = 0;
      hm.wh = q; q = al4(q + HID * HP);
      hm.nw = q; q += HID; hm.nb = q; q += HID;
      hm.bh = q; q += 32;
      hm.tiles = q; hm.wave_stride = al4((HID + TS) * TP);
      q += nw * hm.wave_stride;
      hm.total = q;
    }
    a.map = hm;
    const size_t hbytes = (size_t)hm.total * sizeof(float);
    MAPPO_REQUIRE(hbytes <= LDS_DYN_MAX, "gru_backward: needs %zu B of LDS", hbytes);
    float *dhT = const_cast<float *>(scratch);
    if (head == 1) {
      static const int lds_rc_attr = raise_lds(gru_head_bwd_kernel<1>, "gru_backward");
      if (lds_rc_attr) return lds_rc_attr;
      hipLaunchKernelGGL(gru_head_bwd_kernel<1>, dim3(nb), dim3(WAVE * nw), hbytes, as_stream(stream), a, dhT);
    } else {
      static const int lds_rc_attr = raise_lds(gru_head_bwd_kernel<2>, "gru_backward");
      if (lds_rc_attr) return lds_rc_attr;
      hipLaunchKernelGGL(gru_head_bwd_kernel<2>, dim3(nb), dim3(WAVE * nw), hbytes, as_stream(stream), a, dhT);
    }
    MAPPO_CHECK_LAUNCH("gru_backward (head)");
    const SeqLds sm = seq2_lds(NG, 1);    // 75 KB: the actor's and the critic's workgroups share a CU
    const size_t sbytes = (size_t)sm.total * sizeof(float);
    static const int lds_rc_attr2 = raise_lds(gru_cell_bwd2_kernel, "gru_backward");
    if (lds_rc_attr2) return lds_rc_attr2;
    hipLaunchKernelGGL(gru_cell_bwd2_kernel, dim3(n_ct < 4 * NUM_CU ? n_ct : 4 * NUM_CU), dim3(2 * WAVE), sbytes, as_stream(stream), a, sm);
    MAPPO_CHECK_LAUNCH("gru_backward (cell)");
  }
  if (dxT) return launch_gru_in(false, params, desc, dgiT, (int64_t)L * Nc, dxT, as_stream(stream), "gru_backward");   // d x = W_ih^T d gi
  return MAPPO_OK;
}

extern "C" int32_t mappo_gru_wgrad_slabs(int32_t L, int32_t Nc) {
  const int n_tiles = L * ((Nc + TS - 1) / TS);
  return n_tiles < NUM_CU ? n_tiles : NUM_CU;
}

extern "C" int mappo_gru_wgrad(const mappo_net_desc *desc, const float *xT, const float *scratch, const float *dgiT,
                               const float *dghnT, int32_t L, int32_t Nc, float *slabs, int64_t slab_stride, int64_t slab_col0,
                               mappo_stream_t stream) {
  if (int rc = check_rec(desc, "gru_wgrad")) return rc;
  MAPPO_REQUIRE(xT && scratch && dgiT && dghnT && slabs && L > 0 && Nc > 0, "gru_wgrad: bad arguments");
  MAPPO_CLEAR_STICKY();
  GruWgArgs a = {};
  a.off = net_offsets(*desc); a.xT = xT; a.scratch = scratch; a.dgiT = dgiT; a.dghnT = dghnT; a.L = L; a.Nc = Nc;
  a.slabs = slabs; a.slab_stride = slab_stride; a.slab_col0 = slab_col0;
  MAPPO_REQUIRE(slab_col0 >= 0 && slab_col0 + a.off.total <= slab_stride, "gru_wgrad: slab column range");
  const size_t lds_bytes = (size_t)4 * (96 + HID) * TP * sizeof(float);
  static const int lds_rc_attr = raise_lds(gru_wgrad_kernel, "gru_wgrad");
  if (lds_rc_attr) return lds_rc_attr;
  const int nb = mappo_gru_wgrad_slabs(L, Nc);
  hipLaunchKernelGGL(gru_wgrad_kernel, dim3(nb), dim3(256), lds_bytes, as_stream(stream), a);
  MAPPO_CHECK_LAUNCH("gru_wgrad");
  return MAPPO_OK;
}

static int launch_gru_in(bool gates, const float *params, const mappo_net_desc *desc, const float *inT, int64_t B, float *outT,
                         hipStream_t st, const char *who) {
  if (int rc = check_rec(desc, who)) return rc;
  MAPPO_REQUIRE(params && inT && outT && B > 0, "%s: bad arguments", who);
  MAPPO_CLEAR_STICKY();
  GruInArgs a = {};
  a.params = params; a.off = net_offsets(*desc); a.inT = inT; a.outT = outT; a.B = B;
  const size_t lds_bytes = (size_t)(al4(HID * GS) + NG) * sizeof(float);
  const int64_t n_tiles = (B + TS - 1) / TS;
  int64_t nb = (n_tiles + 3) / 4;
  if (nb > NUM_CU) nb = NUM_CU;
  if (gates) {
    static const int lds_rc_attr = raise_lds(gru_gi_kernel, who);
    if (lds_rc_attr) return lds_rc_attr;
    hipLaunchKernelGGL(gru_gi_kernel, dim3((unsigned)nb), dim3(256), lds_bytes, st, a);
  } else {
    static const int lds_rc_attr = raise_lds(gru_dx_kernel, who);
    if (lds_rc_attr) return lds_rc_attr;
    hipLaunchKernelGGL(gru_dx_kernel, dim3((unsigned)nb), dim3(256), lds_bytes, st, a);
  }
  MAPPO_CHECK_LAUNCH(who);
  return MAPPO_OK;
}

extern "C" int mappo_gru_input_gates(const float *params, const mappo_net_desc *desc, const float *xT, int64_t B, float *giT,
                                     mappo_stream_t stream) {
  return launch_gru_in(true, params, desc, xT, B, giT, as_stream(stream), "gru_input_gates");
}

extern "C" int mappo_gru_input_backward(const float *params, const mappo_net_desc *desc, const float *dgiT, int64_t B, float *dxT,
                                        mappo_stream_t stream) {
  return launch_gru_in(false, params, desc, dgiT, B, dxT, as_stream(stream), "gru_input_backward");
}
