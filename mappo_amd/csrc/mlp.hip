// mlp.hip — K7 / K6 / K8: the shared actor/critic MLP (onpolicy/algorithms/utils/mlp.py:6-55) with its
// head (distributions.py:55-68 logits, r_actor_critic.py:136-165 v_out) on the fp32 matrix cores.
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
// backward pass finds xhat in the tile and only mean/rstd (2 registers) survive from the forward.
//
// mappo_mlp_backward recomputes the forward per tile instead of reading saved activations back from HBM
// (saving 64*4*(2+layer_N) B/sample each way at the cost of ~1/3 more MFMA work), keeps dW accumulators in
// registers across its persistent tile loop, and writes ONE partial-gradient slab per workgroup; the slabs
// are summed by mappo_slab_reduce (deterministic, no float atomics).
//
// Limits of this build: hidden == 64, out_dim <= 32, layer_N <= 2, in_dim <= 64 (the K-chunked layer-1 path
// for wider observations is a separate kernel).
#include "common.h"

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

extern "C" int64_t mappo_net_param_count(const mappo_net_desc *desc) {
  if (!desc) return -1;
  return net_offsets(*desc).total;
}

// ------------------------------------------------------------------------------------------------
// LDS carve-up (floats).  Dp = in_dim rounded up to even.
// ------------------------------------------------------------------------------------------------
struct LdsMap {
  int w1, w2[MAPPO_MAX_LAYER_N], wh;          // weights
  int fn_w, fn_b, b1, ln1_w, ln1_b, b2[MAPPO_MAX_LAYER_N], ln2_w[MAPPO_MAX_LAYER_N], ln2_b[MAPPO_MAX_LAYER_N], bh;
  int tiles;                                   // start of the per-wave tile area
  int tiles_per_wave, wave_stride, total;
};

__host__ __device__ inline LdsMap lds_map(const mappo_net_desc &d, int n_waves, int tiles_per_wave) {
  LdsMap m;
  int p = 0;
  const int Dp = (d.in_dim + 1) & ~1;
  m.w1 = p; p += Dp * WP;
  for (int l = 0; l < MAPPO_MAX_LAYER_N; ++l) { m.w2[l] = p; if (l < d.layer_N) p += HID * WP; }
  m.wh = p; p += HID * HP;
  m.fn_w = p; p += MAXD; m.fn_b = p; p += MAXD;
  m.b1 = p; p += HID; m.ln1_w = p; p += HID; m.ln1_b = p; p += HID;
  for (int l = 0; l < MAPPO_MAX_LAYER_N; ++l) {
    m.b2[l] = p; m.ln2_w[l] = p; m.ln2_b[l] = p;
    if (l < d.layer_N) { m.b2[l] = p; p += HID; m.ln2_w[l] = p; p += HID; m.ln2_b[l] = p; p += HID; }
  }
  m.bh = p; p += 32;
  p = (p + 3) & ~3;
  m.tiles = p;
  m.tiles_per_wave = tiles_per_wave;
  m.wave_stride = tiles_per_wave * HID * TP;
  p += n_waves * m.wave_stride;
  m.total = p;
  return m;
}

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

__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32, WAVE); }

// row (feature within a 32-row MFMA tile) held by accumulator register `reg` of lane-half `half`
#define ROWMAP(reg, half) (((reg) & 3) + 8 * ((reg) >> 2) + 4 * (half))

template <bool RELU>
__device__ __forceinline__ float act_fwd(float z) { return RELU ? fmaxf(z, 0.f) : tanhf(z); }
template <bool RELU>
__device__ __forceinline__ float act_bwd(float a, float da) { return RELU ? (a > 0.f ? da : 0.f) : da * (1.f - a * a); }

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Workgroup-cooperative staging of the weights: global W[f][k] (row-major, K columns) -> LDS dst[k*stride + f].
__device__ __forceinline__ void stage_weight_T(float *dst, const float *__restrict__ src, int F, int K, int Kpad, int stride) {
  for (int e = threadIdx.x; e < F * K; e += blockDim.x) {
    const int f = e / K, k = e - f * K;
    dst[k * stride + f] = src[e];
  }
  for (int e = threadIdx.x; e < F * (Kpad - K); e += blockDim.x) {   // zero the padded k rows
    const int f = e % F, k = K + e / F;
    dst[k * stride + f] = 0.f;
  }
}
__device__ __forceinline__ void stage_vec(float *dst, const float *__restrict__ src, int n, int npad, float fill) {
  for (int e = threadIdx.x; e < npad; e += blockDim.x) dst[e] = (src != nullptr && e < n) ? src[e] : fill;
}

template <int LN>
__device__ __forceinline__ void stage_all_weights(float *lds, const LdsMap &m, const float *__restrict__ params,
                                                  const NetOff &o, const mappo_net_desc &d) {
  const int D = d.in_dim, Dp = (D + 1) & ~1, A = d.out_dim;
  stage_weight_T(lds + m.w1, params + o.w1, HID, D, Dp, WP);
#pragma unroll
  for (int l = 0; l < LN; ++l) stage_weight_T(lds + m.w2[l], params + o.w2[l], HID, HID, HID, WP);
  // head: dst[k*HP + a] = Wh[a][k]; columns a >= A are zero
  for (int e = threadIdx.x; e < HID * 32; e += blockDim.x) {
    const int k = e >> 5, a = e & 31;
    lds[m.wh + k * HP + a] = (a < A) ? params[o.wh + a * HID + k] : 0.f;
  }
  if (d.use_feature_norm) {
    stage_vec(lds + m.fn_w, params + o.fn_w, D, MAXD, 0.f);
    stage_vec(lds + m.fn_b, params + o.fn_b, D, MAXD, 0.f);
  } else {
    for (int e = threadIdx.x; e < MAXD; e += blockDim.x) { lds[m.fn_w + e] = e < D ? 1.f : 0.f; lds[m.fn_b + e] = 0.f; }
  }
  stage_vec(lds + m.b1, params + o.b1, HID, HID, 0.f);
  stage_vec(lds + m.ln1_w, params + o.ln1_w, HID, HID, 0.f);
  stage_vec(lds + m.ln1_b, params + o.ln1_b, HID, HID, 0.f);
#pragma unroll
  for (int l = 0; l < LN; ++l) {
    stage_vec(lds + m.b2[l], params + o.b2[l], HID, HID, 0.f);
    stage_vec(lds + m.ln2_w[l], params + o.ln2_w[l], HID, HID, 0.f);
    stage_vec(lds + m.ln2_b[l], params + o.ln2_b[l], HID, HID, 0.f);
  }
  stage_vec(lds + m.bh, params + o.bh, A, 32, 0.f);
}

// Gather a tile of 32 input rows into tX[k][s] (raw values), zero for samples >= n_valid and for k in [D, Dp).
__device__ __forceinline__ void gather_tile(float *tX, const float *__restrict__ x, const int32_t *__restrict__ rows,
                                            int64_t base, int n_valid, int D, int Dp, int lane) {
  // lane s (< 32) learns the source row of sample s; rows are then broadcast with shuffles
  int64_t my_row = 0;
  if (lane < TS && lane < n_valid) my_row = rows ? (int64_t)rows[base + lane] : base + lane;
  const int per = (D <= 32) ? 2 : 1;            // samples fetched per wave-instruction
  const int kl = (per == 2) ? (lane & 31) : lane;
  const int sub = (per == 2) ? (lane >> 5) : 0;
#pragma unroll 8
  for (int s0 = 0; s0 < TS; s0 += per) {
    const int s = s0 + sub;
    const int64_t row = __shfl(my_row, s, WAVE);
    float v = 0.f;
    if (kl < D && s < n_valid) v = x[row * D + kl];
    if (kl < Dp) tX[kl * TP + s] = v;
  }
}

// LayerNorm over the D input features of each sample, in place: tX <- xhat0 (the affine is applied on read).
__device__ __forceinline__ void feature_norm_tile(float *tX, int D, int Dp, int l31, int half, bool enabled) {
  if (!enabled) return;
  float s = 0.f;
  for (int k = half; k < D; k += 2) s += tX[k * TP + l31];
  const float mean = xhalf_sum(s) / (float)D;
  float q = 0.f;
  for (int k = half; k < D; k += 2) { const float c = tX[k * TP + l31] - mean; q += c * c; }
  const float rstd = 1.0f / sqrtf(xhalf_sum(q) / (float)D + LN_EPS);
  for (int k = half; k < D; k += 2) tX[k * TP + l31] = (tX[k * TP + l31] - mean) * rstd;
}

// acc (2 tiles of 32 features) <- bias
__device__ __forceinline__ void init_bias(f32x16 (&acc)[2], const float *sB, int half) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = sB[32 * t + ROWMAP(r, half)];
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

// acc[t] += W-tile . (tin * gamma + beta)   (forward layer; weights k-major in LDS, K = 2*ksteps)
__device__ __forceinline__ void layer_mfma(f32x16 (&acc)[2], const float *sW, const float *tin, const float *sG,
                                           const float *sBt, int ksteps, int l31, int half) {
#pragma unroll 2
  for (int kk = 0; kk < ksteps; ++kk) {
    const int k = 2 * kk + half;
    const float b = tin[k * TP + l31] * sG[k] + sBt[k];
    const float a0 = sW[k * WP + l31];
    const float a1 = sW[k * WP + 32 + l31];
    acc[0] = mfma(a0, b, acc[0]);
    acc[1] = mfma(a1, b, acc[1]);
  }
}

// ------------------------------------------------------------------------------------------------
// forward of one 32-sample tile.  SAVE keeps the post-activation values and LN statistics for backward.
// Tiles: tX (xhat0), tH[0..layer_N] (layer outputs h_1 .. h_{layer_N+1}).
// ------------------------------------------------------------------------------------------------
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

template <bool RELU, int LN>
__device__ __forceinline__ void tile_forward(const float *lds, const LdsMap &m, float *tX, float *tH, int D, int l31, int half,
                                             TileStats<LN> &st) {
  const int Dp = (D + 1) & ~1;
  f32x16 acc[2];
  // ---- layer 1 (input = xhat0 with the feature-norm affine applied on read) ----
  init_bias(acc, lds + m.b1, half);
  layer_mfma(acc, lds + m.w1, tX, lds + m.fn_w, lds + m.fn_b, Dp / 2, l31, half);
  act_ln_stats<RELU>(acc, st.mean[0], st.rstd[0]);
  st.pos[0] = positive_mask(acc);
  xhat_to_tile(tH, acc, st.mean[0], st.rstd[0], l31, half);
  wave_lds_sync();
  // ---- hidden layers ----
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
  for (int r = 0; r < 16; ++r) acc[r] = lds[m.bh + ROWMAP(r, half)];
  const float *sW = lds + m.wh;
#pragma unroll 4
  for (int kk = 0; kk < HID / 2; ++kk) {
    const int k = 2 * kk + half;
    acc = mfma(sW[k * HP + l31], tLast[k * TP + l31] * sG[k] + sBt[k], acc);
  }
  return acc;
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
};

template <bool RELU, int LN, int MODE>
__global__ __launch_bounds__(256, 1) void mlp_forward_kernel(FwdArgs p) {
  extern __shared__ __align__(16) float lds[];
  const int n_waves = blockDim.x / WAVE;
  const NetOff &o = p.off;
  const LdsMap &m = p.map;
  stage_all_weights<LN>(lds, m, p.params, o, p.desc);
  __syncthreads();
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE, l31 = lane & 31, half = lane >> 5;
  const int D = p.desc.in_dim, Dp = (D + 1) & ~1, A = p.desc.out_dim;
  float *tX = lds + m.tiles + wave * m.wave_stride;
  float *tH = tX + HID * TP;
  const int64_t n_tiles = (p.B + TS - 1) / TS;
  for (int64_t tile = (int64_t)blockIdx.x * n_waves + wave; tile < n_tiles; tile += (int64_t)gridDim.x * n_waves) {
    const int64_t base = tile * TS;
    const int n_valid = (int)min((int64_t)TS, p.B - base);
    gather_tile(tX, p.x, p.rows, base, n_valid, D, Dp, lane);
    wave_lds_sync();
    feature_norm_tile(tX, D, Dp, l31, half, p.desc.use_feature_norm != 0);
    wave_lds_sync();
    TileStats<LN> st;
    tile_forward<RELU, LN>(lds, m, tX, tH, D, l31, half, st);
    const f32x16 z = head_forward(lds, m, tH + LN * HID * TP, lds + ln_w_of<LN>(m, LN), lds + ln_b_of<LN>(m, LN), l31, half);
    // stage the head output as [s][a] (row stride TP) in tX, which is free now
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int a = ROWMAP(r, half);
      if (a < A) tX[l31 * TP + a] = z[r];
    }
    wave_lds_sync();
    if (MODE == 0) {
      for (int e = lane; e < n_valid * A; e += WAVE) {
        const int s = e / A, a = e - s * A;
        p.out[base * A + e] = tX[s * TP + a];
      }
    } else {
      if (lane < n_valid) {
        float *zl = tX + lane * TP;
        const int64_t i = base + lane;
        const float *av = p.avail ? p.avail + i * A : nullptr;
        float zmax = -3.4e38f;
        for (int a = 0; a < A; ++a) {
          float za = zl[a];
          if (av && av[a] == 0.f) { za = -1e10f; zl[a] = za; }
          zmax = fmaxf(zmax, za);
        }
        float se = 0.f;
        for (int a = 0; a < A; ++a) se += expf(zl[a] - zmax);
        const float lse = zmax + logf(se);
        int chosen = 0;
        if (p.deterministic) {
          float best = -3.4e38f;                       // probs.argmax: first maximum
          for (int a = 0; a < A; ++a) { if (zl[a] > best) { best = zl[a]; chosen = a; } }
        } else {
          const float u = (float)(philox_u32(p.seed, p.counter, (uint64_t)i) >> 8) * (1.0f / 16777216.0f);
          float c = 0.f;
          bool found = false;
          for (int a = 0; a < A; ++a) {
            const float pa = expf(zl[a] - lse);
            c += pa;
            if (!found && pa > 0.f) chosen = a;        // fallback: last action with support
            if (!found && u < c) { chosen = a; found = true; }
          }
        }
        p.actions[i] = (float)chosen;
        p.logp[i] = zl[chosen] - lse;
      }
    }
    wave_lds_sync();
  }
}

// ------------------------------------------------------------------------------------------------
// backward kernel
// ------------------------------------------------------------------------------------------------
struct BwdArgs {
  const float *params, *x;
  const int32_t *rows;
  const float *dout;
  float *slabs;
  int64_t slab_stride, slab_col0;
  mappo_net_desc desc;
  NetOff off;
  LdsMap map;
  int64_t B;
};

// sum over the 32 samples of row `f` (= lane) of a [64][TP] tile
__device__ __forceinline__ float tile_row_sum(const float *tile, int lane) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
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
//   in : dH = d/d(h) with h = xhat*gamma + beta the LayerNorm output; `tile` holds xhat (it is consumed: the
//        tile is reused as scratch for the row sums and finally receives dz)
//   out: dH <- d/d(z) (pre-activation), also written to `tile`;  gG/gB (lane = feature) += LN weight/bias grads
template <bool RELU>
__device__ __forceinline__ void ln_act_backward(f32x16 (&dH)[2], float *tile, float mean, float rstd, uint32_t pos,
                                                const float *sG, float &gG, float &gB, int lane, int l31, int half) {
  f32x16 xh[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) xh[t][r] = tile[(32 * t + ROWMAP(r, half)) * TP + l31];
  wave_lds_sync();
  // dbeta[f] = sum_s dy
  regs_to_tile(tile, dH, l31, half);
  wave_lds_sync();
  gB += tile_row_sum(tile, lane);
  wave_lds_sync();
  // dgamma[f] = sum_s dy * xhat ;  dxhat = dy * gamma
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = 32 * t + ROWMAP(r, half);
      tile[f * TP + l31] = dH[t][r] * xh[t][r];
      const float dxh = dH[t][r] * sG[f];
      dH[t][r] = dxh;
      m1 += dxh;
      m2 += dxh * xh[t][r];
    }
  wave_lds_sync();
  gG += tile_row_sum(tile, lane);
  wave_lds_sync();
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

template <bool RELU, int LN>
__global__ __launch_bounds__(256, 1) void mlp_backward_kernel(BwdArgs p) {
  extern __shared__ __align__(16) float lds[];
  const int n_waves = blockDim.x / WAVE;
  const NetOff &o = p.off;
  const LdsMap &m = p.map;
  stage_all_weights<LN>(lds, m, p.params, o, p.desc);
  __syncthreads();
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE, l31 = lane & 31, half = lane >> 5;
  const int D = p.desc.in_dim, Dp = (D + 1) & ~1, A = p.desc.out_dim;
  const bool wide = D > 32;          // second 32-wide tile over the input features in use
  float *tX = lds + m.tiles + wave * m.wave_stride;
  float *tH = tX + HID * TP;

  // ---- gradient accumulators (registers, live across the tile loop) ----
  f32x16 gWh[2], gW2[LN > 0 ? LN : 1][2][2], gW1[2][2];
  float gBh = 0.f, gFnW = 0.f, gFnB = 0.f;
  float gB[LN + 1], gLnW[LN + 1], gLnB[LN + 1];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) gWh[i][r] = 0.f;
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
  for (int l = 0; l <= LN; ++l) { gB[l] = 0.f; gLnW[l] = 0.f; gLnB[l] = 0.f; }

  const int64_t n_tiles = (p.B + TS - 1) / TS;
  for (int64_t tile = (int64_t)blockIdx.x * n_waves + wave; tile < n_tiles; tile += (int64_t)gridDim.x * n_waves) {
    const int64_t base = tile * TS;
    const int n_valid = (int)min((int64_t)TS, p.B - base);
    gather_tile(tX, p.x, p.rows, base, n_valid, D, Dp, lane);
    wave_lds_sync();
    feature_norm_tile(tX, D, Dp, l31, half, p.desc.use_feature_norm != 0);
    wave_lds_sync();
    TileStats<LN> st;
    tile_forward<RELU, LN>(lds, m, tX, tH, D, l31, half, st);
    float *tLast = tH + LN * HID * TP;

    // ---- (A) head weight / bias gradients:  dWh[a][f] += sum_s dout[s][a] * h_last[f][s] ----
    {
      const float *sG = lds + ln_w_of<LN>(m, LN), *sBt = lds + ln_b_of<LN>(m, LN);
      const float g0 = sG[l31], c0 = sBt[l31], g1 = sG[32 + l31], c1 = sBt[32 + l31];
      float bsum = 0.f;
#pragma unroll 2
      for (int ss = 0; ss < TS / 2; ++ss) {
        const int s = 2 * ss + half;
        float av = 0.f;
        if (l31 < A && s < n_valid) av = p.dout[(base + s) * A + l31];
        bsum += av;
        gWh[0] = mfma(av, tLast[l31 * TP + s] * g0 + c0, gWh[0]);
        gWh[1] = mfma(av, tLast[(32 + l31) * TP + s] * g1 + c1, gWh[1]);
      }
      gBh += xhalf_sum(bsum);
    }
    // ---- (B) d h_last = Wh^T . dout ----
    f32x16 dH[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) dH[t][r] = 0.f;
    {
      const float *sW = lds + m.wh;
      for (int kk = 0; kk < (A + 1) / 2; ++kk) {
        const int a = 2 * kk + half;
        float b = 0.f;
        if (a < A && l31 < n_valid) b = p.dout[(base + l31) * A + a];
        dH[0] = mfma(sW[l31 * HP + a], b, dH[0]);
        dH[1] = mfma(sW[(32 + l31) * HP + a], b, dH[1]);
      }
    }
    // ---- hidden layers, last to first ----
#pragma unroll
    for (int l = LN; l >= 1; --l) {
      float *tCur = tH + l * HID * TP;          // xhat of this layer's LayerNorm -> scratch -> dz
      float *tPrev = tH + (l - 1) * HID * TP;   // xhat of the layer's input
      ln_act_backward<RELU>(dH, tCur, st.mean[l], st.rstd[l], st.pos[l], lds + m.ln2_w[l - 1], gLnW[l], gLnB[l], lane, l31, half);
      gB[l] += tile_row_sum(tCur, lane);
      // dW2[f_out][k_in] += sum_s dz[f_out][s] * h_prev[k_in][s]
      {
        const float *sG = lds + ln_w_of<LN>(m, l - 1), *sBt = lds + ln_b_of<LN>(m, l - 1);
        const float g0 = sG[l31], c0 = sBt[l31], g1 = sG[32 + l31], c1 = sBt[32 + l31];
#pragma unroll 2
        for (int ss = 0; ss < TS / 2; ++ss) {
          const int s = 2 * ss + half;
          const float a0 = tCur[l31 * TP + s], a1 = tCur[(32 + l31) * TP + s];
          const float b0 = tPrev[l31 * TP + s] * g0 + c0, b1 = tPrev[(32 + l31) * TP + s] * g1 + c1;
          gW2[l - 1][0][0] = mfma(a0, b0, gW2[l - 1][0][0]);
          gW2[l - 1][0][1] = mfma(a0, b1, gW2[l - 1][0][1]);
          gW2[l - 1][1][0] = mfma(a1, b0, gW2[l - 1][1][0]);
          gW2[l - 1][1][1] = mfma(a1, b1, gW2[l - 1][1][1]);
        }
      }
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
    }
    // ---- layer 1 ----
    {
      float *tCur = tH;
      ln_act_backward<RELU>(dH, tCur, st.mean[0], st.rstd[0], st.pos[0], lds + m.ln1_w, gLnW[0], gLnB[0], lane, l31, half);
      gB[0] += tile_row_sum(tCur, lane);
      // dW1[f_out][k] += sum_s dz1[f_out][s] * xn[k][s],  xn = xhat0 * gamma0 + beta0
      {
        const int k0 = l31, k1 = 32 + l31;
        const bool v0 = k0 < Dp, v1 = k1 < Dp;
        const float g0 = v0 ? lds[m.fn_w + k0] : 0.f, c0 = v0 ? lds[m.fn_b + k0] : 0.f;
        const float g1 = v1 ? lds[m.fn_w + k1] : 0.f, c1 = v1 ? lds[m.fn_b + k1] : 0.f;
#pragma unroll 2
        for (int ss = 0; ss < TS / 2; ++ss) {
          const int s = 2 * ss + half;
          const float a0 = tCur[l31 * TP + s], a1 = tCur[(32 + l31) * TP + s];
          const float b0 = v0 ? tX[k0 * TP + s] * g0 + c0 : 0.f;
          gW1[0][0] = mfma(a0, b0, gW1[0][0]);
          gW1[1][0] = mfma(a1, b0, gW1[1][0]);
          if (wide) {
            const float b1 = v1 ? tX[k1 * TP + s] * g1 + c1 : 0.f;
            gW1[0][1] = mfma(a0, b1, gW1[0][1]);
            gW1[1][1] = mfma(a1, b1, gW1[1][1]);
          }
        }
      }
      // feature-norm gradients: dxn = W1^T . dz1 ; dgamma0[k] = sum_s dxn*xhat0 ; dbeta0[k] = sum_s dxn
      if (p.desc.use_feature_norm) {
        f32x16 dX[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) dX[t][r] = 0.f;
        const float *sW = lds + m.w1;
        const int k0 = l31, k1 = 32 + l31;
#pragma unroll 2
        for (int kk = 0; kk < HID / 2; ++kk) {
          const int fo = 2 * kk + half;
          const float b = tCur[fo * TP + l31];
          const float a0 = (k0 < Dp) ? sW[k0 * WP + fo] : 0.f;
          dX[0] = mfma(a0, b, dX[0]);
          if (wide) {
            const float a1 = (k1 < Dp) ? sW[k1 * WP + fo] : 0.f;
            dX[1] = mfma(a1, b, dX[1]);
          }
        }
        wave_lds_sync();
        regs_to_tile(tCur, dX, l31, half);     // dxn tile [k][s]
        wave_lds_sync();
        if (lane < D) {
          float sb = 0.f, sg = 0.f;
#pragma unroll 8
          for (int j = 0; j < TS; ++j) {
            const float dx = tCur[lane * TP + j];
            sb += dx;
            sg += dx * tX[lane * TP + j];
          }
          gFnB += sb;
          gFnW += sg;
        }
      }
      wave_lds_sync();
    }
  }

  // ---- reduce the waves' accumulators through LDS and write this workgroup's slab ----
  __syncthreads();
  float *red = lds + m.tiles;                      // >= P floats (checked on the host)
  const int P = o.total;
  for (int w = 0; w < n_waves; ++w) {
    if (wave == w) {
      const bool first = (w == 0);
#define RED(idx, val) do { const int i_ = (idx); if (first) red[i_] = (val); else red[i_] += (val); } while (0)
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * ti + ROWMAP(r, half);
#pragma unroll
          for (int tj = 0; tj < 2; ++tj) {
            const int col = 32 * tj + l31;
            if (col < D) RED(o.w1 + row * D + col, gW1[ti][tj][r]);
#pragma unroll
            for (int l = 0; l < LN; ++l) RED(o.w2[l] + row * HID + col, gW2[l][ti][tj][r]);
          }
        }
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int a = ROWMAP(r, half);
          if (a < A) RED(o.wh + a * HID + 32 * tj + l31, gWh[tj][r]);
        }
      RED(o.b1 + lane, gB[0]); RED(o.ln1_w + lane, gLnW[0]); RED(o.ln1_b + lane, gLnB[0]);
#pragma unroll
      for (int l = 0; l < LN; ++l) {
        RED(o.b2[l] + lane, gB[l + 1]); RED(o.ln2_w[l] + lane, gLnW[l + 1]); RED(o.ln2_b[l] + lane, gLnB[l + 1]);
      }
      if (half == 0 && l31 < A) RED(o.bh + l31, gBh);
      if (p.desc.use_feature_norm && lane < D) { RED(o.fn_w + lane, gFnW); RED(o.fn_b + lane, gFnB); }
#undef RED
    }
    __syncthreads();
  }
  float *slab = p.slabs + (size_t)blockIdx.x * p.slab_stride + p.slab_col0;
  for (int e = threadIdx.x; e < P; e += blockDim.x) slab[e] = red[e];
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int check_desc(const mappo_net_desc *d, const char *who) {
  MAPPO_REQUIRE(d, "%s: null desc", who);
  MAPPO_REQUIRE(d->hidden == HID, "%s: hidden_size %d unsupported (kernels are tiled for %d)", who, d->hidden, HID);
  MAPPO_REQUIRE(d->in_dim >= 1 && d->in_dim <= MAXD, "%s: in_dim %d outside [1,%d] (wide-input path not built)", who,
                d->in_dim, MAXD);
  MAPPO_REQUIRE(d->out_dim >= 1 && d->out_dim <= MAPPO_MAX_ACTIONS, "%s: out_dim %d outside [1,%d]", who, d->out_dim,
                MAPPO_MAX_ACTIONS);
  MAPPO_REQUIRE(d->layer_N >= 0 && d->layer_N <= MAPPO_MAX_LAYER_N, "%s: layer_N %d outside [0,%d]", who, d->layer_N,
                MAPPO_MAX_LAYER_N);
  MAPPO_REQUIRE(!d->recurrent, "%s: recurrent networks go through the GRU entry points", who);
  return MAPPO_OK;
}

#define LDS_LIMIT (160 * 1024)
#define NUM_CU 256

static int pick_waves(const mappo_net_desc &d, int tiles_per_wave, int64_t n_tiles) {
  int nw = 4;
  while (nw > 1 && (size_t)lds_map(d, nw, tiles_per_wave).total * sizeof(float) > LDS_LIMIT) nw >>= 1;
  // few tiles (rollout-sized batches): one wave per workgroup spreads them over more CUs
  while (nw > 1 && n_tiles < (int64_t)NUM_CU * nw) nw >>= 1;
  return nw;
}

template <int MODE>
static int launch_forward(const FwdArgs &a_in, hipStream_t st, const char *who) {
  const int64_t n_tiles = (a_in.B + TS - 1) / TS;
  const int LN = a_in.desc.layer_N;
  const int nw = pick_waves(a_in.desc, LN + 2, n_tiles);
  FwdArgs a = a_in;
  a.off = net_offsets(a.desc);
  a.map = lds_map(a.desc, nw, LN + 2);
  const size_t lds_bytes = (size_t)a.map.total * sizeof(float);
  MAPPO_REQUIRE(lds_bytes <= LDS_LIMIT, "%s: needs %zu B of LDS", who, lds_bytes);
  int64_t nb = (n_tiles + nw - 1) / nw;
  if (nb > NUM_CU) nb = NUM_CU;
  dim3 grid((unsigned)nb), block(WAVE * nw);
#define FWD(R, L)                                                                                              \
  do {                                                                                                         \
    static size_t attr_set = 0;                                                                                \
    if (attr_set < lds_bytes) {                                                                                \
      (void)hipFuncSetAttribute((const void *)mlp_forward_kernel<R, L, MODE>,                                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);                   \
      attr_set = LDS_LIMIT;                                                                                    \
    }                                                                                                          \
    hipLaunchKernelGGL((mlp_forward_kernel<R, L, MODE>), grid, block, lds_bytes, st, a);                        \
  } while (0)
  const bool relu = a.desc.use_relu != 0;
  const int prof_id = (MODE == 1) ? MAPPO_PROF_ACT : MAPPO_PROF_MLP_FWD;
  PROF_BEGIN(prof_id, st);
  if (LN == 0) { if (relu) FWD(true, 0); else FWD(false, 0); }
  else if (LN == 1) { if (relu) FWD(true, 1); else FWD(false, 1); }
  else { if (relu) FWD(true, 2); else FWD(false, 2); }
  PROF_END(prof_id, st);
#undef FWD
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

extern "C" int mappo_actor_act(const float *params, const mappo_net_desc *desc, const float *obs, const float *avail,
                               int64_t B, int32_t deterministic, uint64_t seed, uint64_t counter, float *actions,
                               float *logp, mappo_stream_t stream) {
  if (int rc = check_desc(desc, "actor_act")) return rc;
  MAPPO_REQUIRE(params && obs && actions && logp && B > 0, "actor_act: bad arguments");
  FwdArgs a = {};
  a.params = params; a.x = obs; a.rows = nullptr; a.avail = avail; a.actions = actions; a.logp = logp; a.desc = *desc;
  a.B = B; a.deterministic = deterministic; a.seed = seed; a.counter = counter;
  return launch_forward<1>(a, as_stream(stream), "actor_act");
}

static int bwd_waves(const mappo_net_desc &d) {
  int nw = 4;
  while (nw > 1 && (size_t)lds_map(d, nw, d.layer_N + 2).total * sizeof(float) > LDS_LIMIT) nw >>= 1;
  return nw;
}

extern "C" int32_t mappo_mlp_backward_slabs(int64_t B) {
  // upper bound used to size the slab buffer: one slab per workgroup, at most one workgroup per CU
  int64_t n_tiles = (B + TS - 1) / TS;
  return (int32_t)(n_tiles < NUM_CU ? n_tiles : NUM_CU);
}

extern "C" int mappo_mlp_backward(const float *params, const mappo_net_desc *desc, const float *x, const int32_t *rows,
                                  int64_t B, const float *dout, float *slabs, int64_t slab_stride, int64_t slab_col0,
                                  mappo_stream_t stream) {
  if (int rc = check_desc(desc, "mlp_backward")) return rc;
  MAPPO_REQUIRE(params && x && dout && slabs && B > 0, "mlp_backward: bad arguments");
  const NetOff o = net_offsets(*desc);
  MAPPO_REQUIRE(slab_col0 >= 0 && slab_col0 + o.total <= slab_stride, "mlp_backward: slab column range");
  const int LN = desc->layer_N;
  const int nw = bwd_waves(*desc);
  const LdsMap m = lds_map(*desc, nw, LN + 2);
  const size_t lds_bytes = (size_t)m.total * sizeof(float);
  MAPPO_REQUIRE(lds_bytes <= LDS_LIMIT, "mlp_backward: needs %zu B of LDS", lds_bytes);
  MAPPO_REQUIRE(nw * m.wave_stride >= o.total, "mlp_backward: reduction buffer smaller than the parameter count");
  // every slab the caller sized for (mappo_mlp_backward_slabs) must be written: grid == that count
  const int nb = mappo_mlp_backward_slabs(B);
  BwdArgs a;
  a.params = params; a.x = x; a.rows = rows; a.dout = dout; a.slabs = slabs; a.slab_stride = slab_stride;
  a.slab_col0 = slab_col0; a.desc = *desc; a.B = B; a.off = o; a.map = m;
  dim3 grid((unsigned)nb), block(WAVE * nw);
  hipStream_t st = as_stream(stream);
#define BWD(R, L)                                                                                          \
  do {                                                                                                     \
    static size_t attr_set = 0;                                                                            \
    if (attr_set < lds_bytes) {                                                                            \
      (void)hipFuncSetAttribute((const void *)mlp_backward_kernel<R, L>,                                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);               \
      attr_set = LDS_LIMIT;                                                                                \
    }                                                                                                      \
    hipLaunchKernelGGL((mlp_backward_kernel<R, L>), grid, block, lds_bytes, st, a);                         \
  } while (0)
  const bool relu = desc->use_relu != 0;
  PROF_BEGIN(MAPPO_PROF_MLP_BWD, st);
  if (LN == 0) { if (relu) BWD(true, 0); else BWD(false, 0); }
  else if (LN == 1) { if (relu) BWD(true, 1); else BWD(false, 1); }
  else { if (relu) BWD(true, 2); else BWD(false, 2); }
  PROF_END(MAPPO_PROF_MLP_BWD, st);
#undef BWD
  MAPPO_CHECK_LAUNCH("mlp_backward");
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
