// gru.hip — K9: the recurrent layer of R_Actor / R_Critic (onpolicy/algorithms/utils/rnn.py:7-80) OUTSIDE the training pass:
//   h_t = GRU(x_t, h_{t-1} * mask_t)   (torch nn.GRU, 1 layer, gate order r,z,n; rnn.py:13,27,67)
//   y_t = LayerNorm(h_t)               (rnn.py:22,79)          -> head (distributions.py:55-68 | v_out)
// The reference's mask-segmented sequence run (rnn.py:30-77) equals multiplying h by mask_t before every step
// (SURVEY.md §3.4), which is what these kernels do.
//
//   gru_step3*_kernel     one rollout / get_values step (L = 1): 16-row tile per 4-wave workgroup, hidden units split over the
//                         waves (gru_step3.h); dual forms serve the actor AND the critic, with the narrow-input trunks fused in
//                         (gru_step3f_dual_kernel; wide inputs: wide_recurrent_step_dual_kernel, mlp_wide16.h)
//   gru_fwd_kernel        L-step sequences from trunk features x_T [64][B] (evaluate_actions / get_values on stacked chunks):
//                         a wavefront owns 32 sequences on 32x32x2 tiles, W_ih / W_hh k-major in LDS (row stride 193)
// The TRAINING pass (forward recurrence + head + loss, backward recurrence, weight gradients) is gru_train16.hip.
#include <stdlib.h>
#include "mlp_core.h"
#include "mlp_trunk16r.h"
#include "gru_step3.h"
#include "insert_core.h"

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

__device__ __forceinline__ void gru_cell(CellOut &c, const float *lds, const GruLds &m, const float *__restrict__ xT, int64_t ldx,
                                         int64_t col, bool ok, const float *tHm, int l31, int half) {
  f32x16 arz[4], ain[2], ahn[2];
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
// L-step sequences outside the training pass (evaluate_actions / get_values on stacked chunks; the training pass is
// gru_train16.hip, a rollout step gru_step3.h).  HM: head mode 0 none | 1 out[B][A] | 2 sample — compile-time, so that each use is
// a lean instantiation (as one kernel with runtime switches the register allocator spilled ~200 registers).
template <int HM>
__global__ __launch_bounds__(256, 1) void gru_fwd_kernel(GruFwdArgs p) {
  extern __shared__ __align__(16) float lds[];
  const GruLds &m = p.map;
  const int n_waves = p.tile_waves;
  const int lane = threadIdx.x & (WAVE - 1), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)), l31 = lane & 31, half = lane >> 5;
  stage_gru_all(lds, m, p.params, p.off, p.A, true, HM != 0);
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
      gru_cell(co, lds, m, p.xT, B, col, ok, tHm, l31, half);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[tt][r] = (1.f - co.z[tt][r]) * co.n[tt][r] + co.z[tt][r] * hm[tt][r];
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

// the same with the SMAC insert of the env output the rows are read from as a third workgroup role (mappo_recurrent_rollout_step)
template <int TR, int TLN>
__global__ __launch_bounds__(256, 1) void gru_step3f_dual_ins_kernel(GruFwdArgs a, GruFwdArgs c, int nA, SmacInsert ins, int nI) {
  __shared__ Step3Shared sh;
  if ((int)blockIdx.x < nA) gru_step3_body<2, TR, TLN>(a, sh, blockIdx.x, nA);
  else if ((int)blockIdx.x < 2 * nA) gru_step3_body<1, TR, TLN>(c, sh, blockIdx.x - nA, nA);
  else insert_smac_body(ins, (int)blockIdx.x - 2 * nA, nI);
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

extern "C" int mappo_gru_forward(const float *params, const mappo_net_desc *desc, const float *xT, const float *h0,
                                 const int32_t *h0_rows, const float *masks, const int32_t *rows, int32_t L, int32_t Nc,
                                 float *h_last, int32_t head_mode, float *out, const float *avail,
                                 int32_t deterministic, uint64_t seed, uint64_t counter, const uint64_t *counter_dev,
                                 float *actions, float *logp, mappo_stream_t stream) {
  if (int rc = check_rec(desc, "gru_forward")) return rc;
  MAPPO_REQUIRE(params && xT && h0 && masks && L > 0 && Nc > 0, "gru_forward: bad arguments");
  MAPPO_REQUIRE(head_mode >= 0 && head_mode <= 2, "gru_forward: head_mode %d", head_mode);
  MAPPO_REQUIRE(head_mode != 1 || out, "gru_forward: out required");
  MAPPO_REQUIRE(head_mode != 2 || (actions && logp), "gru_forward: actions/logp required");
  MAPPO_CLEAR_STICKY();
  GruFwdArgs a = {};
  a.params = params; a.off = net_offsets(*desc); a.xT = xT; a.h0 = h0; a.h0_rows = h0_rows; a.masks = masks; a.rows = rows;
  a.L = L; a.Nc = Nc; a.A = desc->out_dim; a.head_mode = head_mode; a.h_last = h_last; a.out = out;
  a.avail = avail; a.actions = actions; a.logp = logp; a.deterministic = deterministic; a.seed = seed; a.counter = counter;
  a.counter_dev = counter_dev;
  const int n_tiles = (Nc + TS - 1) / TS;
  if (L == 1 && head_mode != 0) {
    // rollout / get_values step: one 16-row tile per 4-wave workgroup, hidden units split over the waves
    const int nt16 = (Nc + 15) / 16;
    const int g3 = nt16 < 2 * NUM_CU ? nt16 : 2 * NUM_CU;
    if (head_mode == 1) hipLaunchKernelGGL(gru_step3_kernel<1>, dim3(g3), dim3(4 * WAVE), 0, as_stream(stream), a);
    else hipLaunchKernelGGL(gru_step3_kernel<2>, dim3(g3), dim3(4 * WAVE), 0, as_stream(stream), a);
    MAPPO_CHECK_LAUNCH("gru_forward");
    return MAPPO_OK;
  }
  // L-step sequences (evaluate_actions / get_values on stacked chunks outside the training pass, which is gru_train16.hip)
  int nw = head_mode ? 2 : 4;
  while (nw > 1 && n_tiles < nw) nw >>= 1;
  const int wave_rows = head_mode ? (HID + HID + TS) : HID;
  a.map = gru_lds(nw, wave_rows, true);
  const size_t lds_bytes = (size_t)a.map.total * sizeof(float);
  MAPPO_REQUIRE(lds_bytes <= LDS_DYN_MAX, "gru_forward: needs %zu B of LDS", lds_bytes);
  int nb = (n_tiles + nw - 1) / nw;
  if (nb > NUM_CU) nb = NUM_CU;
  a.tile_waves = nw;
  const int launch_waves = 4;                              // four waves stage the 98 KB of GRU weights, `nw` of them (LDS budget) own tiles
#define GRU_FWD(HM_)                                                                                              \
  do {                                                                                                            \
    static const int lds_rc_attr = raise_lds(gru_fwd_kernel<HM_>, "gru_forward");                          \
    if (lds_rc_attr) return lds_rc_attr;                                                                          \
    hipLaunchKernelGGL((gru_fwd_kernel<HM_>), dim3(nb), dim3(WAVE * launch_waves), lds_bytes, as_stream(stream), a); \
  } while (0)
  if (head_mode == 0) GRU_FWD(0); else if (head_mode == 1) GRU_FWD(1); else GRU_FWD(2);
#undef GRU_FWD
  MAPPO_CHECK_LAUNCH("gru_forward");
  return MAPPO_OK;
}

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
  const int nt16 = (Nc + 15) / 16;
  const int g3 = nt16 < NUM_CU ? nt16 : NUM_CU;
  hipLaunchKernelGGL(gru_step3_dual_kernel, dim3(2 * g3), dim3(4 * WAVE), 0, as_stream(stream), a, c, g3);
  MAPPO_CHECK_LAUNCH("gru_step_dual");
  return MAPPO_OK;
}

int mappo_recurrent_step_dual_wide_(const float *actor_params, const mappo_net_desc *actor_desc, const float *obs, const float *actor_h0,
                                    float *actor_h_last, const float *critic_params, const mappo_net_desc *critic_desc, const float *share_obs,
                                    const float *critic_h0, float *critic_h_last, const float *masks, int32_t Nc, const float *avail,
                                    int32_t deterministic, uint64_t seed, uint64_t counter, const uint64_t *counter_dev, float *actions,
                                    float *logp, float *values, const SmacInsert *ins, mappo_stream_t stream);      // mlp.hip
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
                                           critic_h_last, masks, Nc, avail, deterministic, seed, counter, counter_dev, actions, logp, values, nullptr,
                                           stream);
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

// The SMAC rollout step in ONE launch (smac_runner.py:110-151 across two consecutive steps): the insert of the env output of
// step k - 1 into buffer slot k (mappo_insert_smac: masks / active_masks / bad_masks, rnn states x (1 - env done), slot copies) AND
// get_actions / get_values of step k, which read that env output and the states the step before returned IN PLACE (the row mask is
// derived from `dones` exactly as the insert derives masks[k]).  The insert was a launch of its own, 5 us of a 30-45 us step.
extern "C" int mappo_recurrent_rollout_step(const float *actor_params, const mappo_net_desc *actor_desc, const float *critic_params,
                                            const mappo_net_desc *critic_desc, const float *obs, const float *share_obs, const float *avail,
                                            const float *rewards, int64_t rew_stride_n, int64_t rew_stride_m, const uint8_t *dones,
                                            int64_t done_stride_n, int64_t done_stride_m, const uint8_t *bad_transition, const float *actor_h,
                                            const float *critic_h, float *actor_h_next, float *critic_h_next, int32_t N, int32_t M,
                                            int32_t deterministic, uint64_t seed, uint64_t counter, const uint64_t *counter_dev, float *actions,
                                            float *logp, float *values, const mappo_smac_slot *dst, mappo_stream_t stream) {
  if (int rc = check_rec(actor_desc, "recurrent_rollout_step")) return rc;
  if (int rc = check_rec(critic_desc, "recurrent_rollout_step")) return rc;
  MAPPO_REQUIRE(critic_desc->out_dim == 1, "recurrent_rollout_step: critic out_dim must be 1");
  MAPPO_REQUIRE(actor_params && critic_params && obs && share_obs && rewards && dones && actor_h && critic_h && actor_h_next && critic_h_next &&
                    actions && logp && values && dst && N > 0 && M > 0, "recurrent_rollout_step: bad arguments");
  MAPPO_REQUIRE(dst->obs && dst->share_obs && dst->rewards && dst->masks && dst->bad_masks && dst->active_masks && dst->rnn_states &&
                    dst->rnn_states_critic && (!avail || dst->available_actions), "recurrent_rollout_step: incomplete destination slot");
  MAPPO_REQUIRE(actor_desc->layer_N == critic_desc->layer_N && actor_desc->use_relu == critic_desc->use_relu,
                "recurrent_rollout_step: the networks must share layer_N and the activation");
  MAPPO_REQUIRE(((((uintptr_t)actor_h) | ((uintptr_t)critic_h) | ((uintptr_t)dst->rnn_states) | ((uintptr_t)dst->rnn_states_critic)) & 15) == 0,
                "recurrent_rollout_step: state arrays must be 16-byte aligned");
  const int Nc = N * M;
  SmacInsert ins;
  ins.obs = obs; ins.share = share_obs; ins.avail = avail; ins.rew = rewards; ins.rew_sn = rew_stride_n; ins.rew_sm = rew_stride_m;
  ins.done = dones; ins.done_sn = done_stride_n; ins.done_sm = done_stride_m; ins.bad = bad_transition; ins.h_a = actor_h; ins.h_c = critic_h;
  ins.obs_dst = dst->obs; ins.share_dst = dst->share_obs; ins.avail_dst = dst->available_actions; ins.rew_dst = dst->rewards;
  ins.mask_dst = dst->masks; ins.bad_dst = dst->bad_masks; ins.active_dst = dst->active_masks; ins.ha_dst = dst->rnn_states;
  ins.hc_dst = dst->rnn_states_critic; ins.N = N; ins.M = M; ins.D = actor_desc->in_dim; ins.S = critic_desc->in_dim;
  ins.A = actor_desc->out_dim; ins.H = HID;
  if (actor_desc->in_dim > 64 && critic_desc->in_dim > 64)
    return mappo_recurrent_step_dual_wide_(actor_params, actor_desc, obs, actor_h, actor_h_next, critic_params, critic_desc, share_obs, critic_h,
                                           critic_h_next, nullptr, Nc, avail, deterministic, seed, counter, counter_dev, actions, logp, values, &ins,
                                           stream);
  MAPPO_REQUIRE(actor_desc->in_dim <= 64 && critic_desc->in_dim <= 64 && actor_desc->layer_N <= 1,
                "recurrent_rollout_step: both networks narrow (in_dim <= 64, layer_N <= 1) or both wide (65..512)");
  MAPPO_CLEAR_STICKY();
  GruFwdArgs a = {}, c = {};
  a.params = actor_params; a.off = net_offsets(*actor_desc); a.x_rows = obs; a.desc = *actor_desc; a.h0 = actor_h; a.L = 1; a.Nc = Nc;
  a.A = actor_desc->out_dim; a.head_mode = 2; a.h_last = actor_h_next; a.avail = avail; a.actions = actions; a.logp = logp;
  a.deterministic = deterministic; a.seed = seed; a.counter = counter; a.counter_dev = counter_dev;
  a.dones = dones; a.done_M = M; a.done_sn = done_stride_n; a.done_sm = done_stride_m;
  c.params = critic_params; c.off = net_offsets(*critic_desc); c.x_rows = share_obs; c.desc = *critic_desc; c.h0 = critic_h; c.L = 1; c.Nc = Nc;
  c.A = 1; c.head_mode = 1; c.h_last = critic_h_next; c.out = values;
  c.dones = dones; c.done_M = M; c.done_sn = done_stride_n; c.done_sm = done_stride_m;
  const int nt16 = (Nc + 15) / 16;
  const int g3 = nt16 < NUM_CU ? nt16 : NUM_CU;
  const int64_t most = (int64_t)Nc * (ins.D > ins.S ? ins.D : ins.S);
  int nI = (int)((most + 2047) / 2048);                         // ~8 elements per thread
  nI = nI < 1 ? 1 : (nI > 64 ? 64 : nI);
  const dim3 grid(2 * g3 + nI), block(4 * WAVE);
  hipStream_t st = as_stream(stream);
  const bool relu = actor_desc->use_relu != 0;
  switch (actor_desc->layer_N) {
    case 0: if (relu) hipLaunchKernelGGL((gru_step3f_dual_ins_kernel<1, 0>), grid, block, 0, st, a, c, g3, ins, nI); else hipLaunchKernelGGL((gru_step3f_dual_ins_kernel<2, 0>), grid, block, 0, st, a, c, g3, ins, nI); break;
    default: if (relu) hipLaunchKernelGGL((gru_step3f_dual_ins_kernel<1, 1>), grid, block, 0, st, a, c, g3, ins, nI); else hipLaunchKernelGGL((gru_step3f_dual_ins_kernel<2, 1>), grid, block, 0, st, a, c, g3, ins, nI); break;
  }
  MAPPO_CHECK_LAUNCH("recurrent_rollout_step");
  return MAPPO_OK;
}
