// mlp_wide16.h — layer 1 of the MLP trunk for wide inputs (in_dim 65..512: SMAC 25m observations / states, the 512-wide
// stress config), as its own kernel on v_mfma_f32_16x16x4_f32:
//     z1[s][f] = b1'[f] + sum_k W1'[f][k] xhat0[s][k],   xhat0 = LayerNorm(x[s]) without affine (folded: W1' = W1 gamma0, b1' = b1 + W1 beta0)
// (mlp.py:45,51-55: feature norm + fc1 up to the pre-activation) plus the row statistics (mean0, rstd0) the weight-gradient
// kernel needs.  The rest of the network runs from z1 (mlp_upd16.h with XL1, 64 values per sample instead of in_dim).
//
// Why a separate kernel: with in_dim = 512 layer 1 is 88 % of the forward MACs but W1 (128 KB) does not fit LDS next to
// anything else and its gradient (64 x 512 accumulators) does not fit a wave.  Round 1 K-chunked it inside the one-wave-per-
// tile update kernel: 7 % of the fp32 MFMA peak (profiles/r02/c_configs345_kernel_stats_before_wide16.txt).  Here:
//   * a wave owns a 16-sample tile and keeps its WHOLE input row block in registers (16 NCH values per lane: lane (n, q)
//     holds columns 64 c + 16 q + j of sample n), read ONCE from HBM as 16-byte loads; the LayerNorm statistics are the exact
//     two-pass form on those registers; the registers are the B operands as they are (k-step (c, j) <-> column 64 c + 16 q + j);
//   * the 8 waves of a workgroup share each 64-column chunk of W1' through a double-buffered LDS tile [64][68]: chunk c + 1 is
//     fetched (global -> registers) under the 64 MFMAs of chunk c and stored before the single barrier per chunk;
//   * output z1 [B][64] row-major (the accumulator layout gives each lane 4 x 16 contiguous bytes).
#pragma once

struct Wide16Args {
  const float *params, *x;
  const int32_t *rows;
  float *z1;                 // [B][64]
  float *mean0, *rstd0;      // [B] each (may be NULL: rollout forward)
  int64_t B;
  int D, w1, b1, fn_w, fn_b; // offsets into params (fn_* < 0: no feature norm)
};


// Elements k .. k + 3 of a row of D floats (D >= 4) as one 16-byte load that never leaves the row: the load starts at
// min(k, D - 4) and the lanes are shifted down by the difference, zeros beyond the row.  Rows need 4-byte alignment only
// (global_load_dwordx4 takes any dword address); with D % 4 == 0 and k % 4 == 0 the shift is 0 or >= 4 (all zero).
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ f32x4 ld4_row(const float *row, int k, int D, bool al4) {
  const int kk = min(k, D - 4), s = k - kk;
  const f32x4_u l = *reinterpret_cast<const f32x4_u *>(row + kk);
  f32x4 r;
  if (al4) {
    const bool in = s == 0;
    r[0] = in ? l[0] : 0.f; r[1] = in ? l[1] : 0.f; r[2] = in ? l[2] : 0.f; r[3] = in ? l[3] : 0.f;
  } else {
    r[0] = s == 0 ? l[0] : (s == 1 ? l[1] : (s == 2 ? l[2] : (s == 3 ? l[3] : 0.f)));
    r[1] = s == 0 ? l[1] : (s == 1 ? l[2] : (s == 2 ? l[3] : 0.f));
    r[2] = s == 0 ? l[2] : (s == 1 ? l[3] : 0.f);
    r[3] = s == 0 ? l[3] : 0.f;
  }
  return r;
}

// b1'[f] = b1[f] + sum_k W1[f][k] beta0[k]   (64 rows x 8 threads, all 512 threads)
__device__ __forceinline__ void wide16_fold_bias(const Wide16Args &p, float *sB) {
  const int tid = threadIdx.x, f = tid >> 3, part = tid & 7;
  float acc = 0.f;
  if (p.fn_b >= 0) {
    const float *w = p.params + p.w1 + (size_t)f * p.D, *bt = p.params + p.fn_b;
    for (int k0 = part; k0 < p.D; k0 += 64) {
      float wv[8], bv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int k = min(k0 + 8 * j, p.D - 1); wv[j] = w[k]; bv[j] = bt[k]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) if (k0 + 8 * j < p.D) acc += wv[j] * bv[j];
    }
  }
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) acc += __shfl_xor(acc, off, WAVE);
  if (part == 0) sB[f] = p.params[p.b1 + f] + acc;
}

// The tile loop of layer 1: for every 16-sample tile of this wave calls tail(acc, i, ok, mean, rstd) with acc = z1 of sample i
// (accumulator layout: lane (n, q) holds features 16 b + 4 q + r).  sW: [2][64 * RS16] chunk buffers, sB: [64] folded bias.
template <int NCH, class Tail>
__device__ __forceinline__ void wide16_layer1(const Wide16Args &p, float (*sW)[HID * RS16], float *sB, Tail &&tail) {
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = p.D;
  const bool fnorm = p.fn_w >= 0;
  const bool al4 = (D & 3) == 0;                             // every 4-column group is inside or outside a row as a whole
  const float inv_D = 1.0f / (float)D;
  wide16_fold_bias(p, sB);

  // this thread's share of a weight chunk: row wf, columns 8 wp .. 8 wp + 7 (64 rows x 8 threads)
  const int wf = tid >> 3, wp = tid & 7;
  auto fetch_chunk = [&](int c, float (&w)[8]) {
    const float *wrow = p.params + p.w1 + (size_t)wf * D;
    int wpl = wp;
    asm volatile("" : "+v"(wrow), "+v"(wpl));                 // per-call addresses (hoisted out of the tile loop they cost 60 VGPRs)
    const int k0 = 64 * c + 8 * wpl;
    const f32x4 a = ld4_row(wrow, k0, D, al4), b = ld4_row(wrow, k0 + 4, D, al4);
    w[0] = a[0]; w[1] = a[1]; w[2] = a[2]; w[3] = a[3]; w[4] = b[0]; w[5] = b[1]; w[6] = b[2]; w[7] = b[3];
    if (fnorm) {
      const float *g = p.params + p.fn_w;
      asm volatile("" : "+s"(g));
      const f32x4 ga = ld4_row(g, k0, D, al4), gb = ld4_row(g, k0 + 4, D, al4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { w[j] *= ga[j]; w[4 + j] *= gb[j]; }
    }
  };
  auto store_chunk = [&](int buf, const float (&w)[8]) {
    float *dst = &sW[buf][wf * RS16 + 8 * wp];
    *reinterpret_cast<float4 *>(dst) = make_float4(w[0], w[1], w[2], w[3]);
    *reinterpret_cast<float4 *>(dst + 4) = make_float4(w[4], w[5], w[6], w[7]);
  };

  const int64_t n_tiles = (p.B + 15) / 16;
  const int64_t n_groups = (n_tiles + 7) / 8;                 // 8 tiles (one per wave) share the weight stream
  const int n_chunks = (D + 63) / 64;
  for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const int64_t tile = grp * 8 + wave;
    const int64_t i = tile * 16 + n;
    const bool ok = i < p.B;
    const int64_t row = ok ? (p.rows ? (int64_t)p.rows[i] : i) : 0;
    const float *xr = p.x + row * D;
    int ql = q;
    asm volatile("" : "+v"(ql));                               // column offsets / masks per tile, not 32 hoisted address pairs
    // ---- the row block into registers: column 64 c + 16 q + j ----
    float xv[NCH][16];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        const f32x4 t = ld4_row(xr, 64 * c + 16 * ql + 4 * j4, D, al4);
        xv[c][4 * j4] = t[0]; xv[c][4 * j4 + 1] = t[1]; xv[c][4 * j4 + 2] = t[2]; xv[c][4 * j4 + 3] = t[3];
      }
    }
    float w_next[8];
    fetch_chunk(0, w_next);
    // ---- LayerNorm statistics over the D inputs (exact two-pass on the registers) ----
    float mean = 0.f, rstd = 1.f;
    if (fnorm) {
      float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 16; ++j) s4[j & 3] += xv[c][j];
      mean = quad_sum16((s4[0] + s4[1]) + (s4[2] + s4[3])) * inv_D;
      float v4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float d = (64 * c + 16 * ql + j < D) ? xv[c][j] - mean : 0.f;
          xv[c][j] = d;
          v4[j & 3] += d * d;
        }
      rstd = 1.0f / sqrtf(quad_sum16((v4[0] + v4[1]) + (v4[2] + v4[3])) * inv_D + LN_EPS);
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 16; ++j) xv[c][j] *= rstd;
    }
    // ---- z1 = b1' + W1' xhat0, chunk by chunk ----
    __syncthreads();                                           // sB ready (first group) / previous group's last chunk consumed
    store_chunk(0, w_next);
    f32x4 acc[4];
#pragma unroll
    for (int bo = 0; bo < 4; ++bo) acc[bo] = ld4(sB + 16 * bo + 4 * q);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c < n_chunks) {
        if (c + 1 < n_chunks) fetch_chunk(c + 1, w_next);
        const float *Wc = &sW[c & 1][n * RS16 + 16 * q];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          f32x4 a[4];
#pragma unroll
          for (int bo = 0; bo < 4; ++bo) a[bo] = ld4(Wc + 16 * bo * RS16 + 4 * jj);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int bo = 0; bo < 4; ++bo) acc[bo] = mfma16(a[bo][t], xv[c][4 * jj + t], acc[bo]);
        }
        if (c + 1 < n_chunks) store_chunk((c + 1) & 1, w_next);
        __syncthreads();
      }
    }
    tail(acc, i, ok, mean, rstd);
  }
}

template <int NCH>
__global__ __launch_bounds__(512, 2) void wide_l1_fwd16_kernel(Wide16Args p) {
  __shared__ __align__(16) float sW[2][HID * RS16];       // W1' chunk [f][k], double buffered
  __shared__ __align__(16) float sB[HID];
  const int q = (threadIdx.x & 63) >> 4;
  wide16_layer1<NCH>(p, sW, sB, [&](f32x4 (&acc)[4], int64_t i, bool ok, float mean, float rstd) __attribute__((always_inline)) {
    if (ok) {
      if (q == 0 && p.mean0) { p.mean0[i] = mean; p.rstd0[i] = rstd; }
#pragma unroll
      for (int b = 0; b < 4; ++b) st4(p.z1 + i * HID + 16 * b + 4 * q, acc[b]);
    }
  });
}

// The whole forward of a wide-input network in one launch (rollout: get_actions / get_values / trunk features): layer 1 as above,
// then the register-resident 16x16x4 tail of the narrow kernels (mlp_fwd16.h) on the same tile.
template <bool RELU, int LN, int MODE>
__global__ __launch_bounds__(512, 2) void wide_forward16_kernel(Wide16Args w, FwdArgs p) {
  extern __shared__ __align__(16) float lds[];
  __shared__ __align__(16) float sW[2][HID * RS16];
  __shared__ __align__(16) float sB[HID];
  stage_all_weights<LN>(lds, p.map, p.params, p.off, p.desc);      // everything but W1 (in_dim > 64: streamed in chunks)
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float *tZ = lds + p.map.tiles + wave * p.map.wave_stride;
  wide16_layer1<8>(w, sW, sB, [&](f32x4 (&acc)[4], int64_t i, bool ok, float, float) __attribute__((always_inline)) {
    forward16_tail<RELU, LN, MODE>(p, lds, p.map, acc, i, ok, j, q, tZ);
  });
}

// ------------------------------------------------------------------------------------------------------------------------
// wide_l1_bwd16_kernel — weight gradient of layer 1 and the feature-norm gradients for in_dim 65..512 from dz1 (feature-major
// [64][B]) and the row statistics the forward left:
//     G[f][k] = sum_s dz1[f][s] xhat0[k][s]   (RAW product: xhat0 without the affine),   db[f] = sum_s dz1[f][s]
//     dW1 = gamma0[k] G + beta0[k] db[f],   dgamma0[k] = sum_f W1[f][k] G[f][k],   dbeta0[k] = sum_f W1[f][k] db[f]
// (the raw-product identities of mlp_impl.h: no dX = W1^T dz1 pass — half the MFMA work of the round-1 kernel, which also
// re-read dz1 once per 64-column chunk).  A workgroup walks 16-sample tiles; wave w owns the 64-column chunk w % NCA of W1's
// gradient (64 accumulator registers) for the tiles of its tile group w / NCA:
//   * A operand = dz1^T: lane (m, q) needs dz1[16 bf + m][4 q .. 4 q + 3] — ONE 16-byte load from the feature-major array;
//   * B operand = xhat0^T of the chunk: lane (n, q) reads x[row 4 q + j][chunk + 16 bk + n] (64-byte segments; every input
//     element is read exactly once by exactly one wave) and normalises it with the sample's (mean0, rstd0);
//   * 64 MFMAs per tile and wave; the next tile's operands are fetched under them.
// The transform above runs per wave in registers at the end (it is linear, so it commutes with the slab reduction); one slab
// row per (workgroup, tile group).
// ------------------------------------------------------------------------------------------------------------------------
struct WideBwd16Args {
  const float *params, *x;
  const int32_t *rows;
  const float *wide_ws;      // dz1 [64][B] | mean0 [B] | rstd0 [B]
  float *slabs;
  int64_t slab_stride, slab_col0, B;
  int D, w1, fn_w, fn_b;     // fn_* < 0: no feature norm
  int nca, groups;           // chunk owners per tile group (power of two >= number of chunks), tile groups (nca * groups <= 8)
};

struct WideBwdOps {
  f32x4 a[4];                // dz1[16 bf + m][4 q + j]
  float b[4][4];             // x[row 4 q + j][chunk + 16 bk + n]  (b[j][bk])
  f32x4 mean, rstd;          // of samples 4 q + j
};

__device__ __forceinline__ void wide_bwd_fetch(WideBwdOps &o, const WideBwd16Args &p, int64_t tile, int c0, int n, int q, bool al) {
  const int64_t base = tile * 16, s0 = base + 4 * q;
  const float *dz = p.wide_ws, *st = p.wide_ws + (int64_t)HID * p.B;
  const bool full = base + 16 <= p.B;
  if (full && al) {
#pragma unroll
    for (int bf = 0; bf < 4; ++bf) o.a[bf] = ld4(dz + (int64_t)(16 * bf + n) * p.B + s0);
    o.mean = ld4(st + s0);
    o.rstd = ld4(st + p.B + s0);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t s = min(s0 + j, p.B - 1);
      const bool ok = s0 + j < p.B;
#pragma unroll
      for (int bf = 0; bf < 4; ++bf) { const float v = dz[(int64_t)(16 * bf + n) * p.B + s]; o.a[bf][j] = ok ? v : 0.f; }
      o.mean[j] = st[s]; o.rstd[j] = st[p.B + s];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t s = min(s0 + j, p.B - 1);
    const int64_t row = p.rows ? (int64_t)p.rows[s] : s;
    const float *xr = p.x + row * p.D;
#pragma unroll
    for (int bk = 0; bk < 4; ++bk) o.b[j][bk] = xr[min(c0 + 16 * bk + n, p.D - 1)];
  }
}

template <int UNUSED>            // (a template only so that the header may be included by every translation unit)
__global__ __launch_bounds__(512, 2) void wide_l1_bwd16_kernel(WideBwd16Args p) {
  __shared__ float sDb[8][HID];
  const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int chunk = wave % p.nca, tg = wave / p.nca;
  const int c0 = 64 * chunk;
  const bool active = tg < p.groups && c0 < p.D;
  const bool fnorm = p.fn_w >= 0;
  const bool al = (p.B & 3) == 0 && (((uintptr_t)p.wide_ws) & 15) == 0;
  f32x4 G[4][4];
  float db[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int bf = 0; bf < 4; ++bf)
#pragma unroll
    for (int bk = 0; bk < 4; ++bk) { G[bf][bk][0] = 0.f; G[bf][bk][1] = 0.f; G[bf][bk][2] = 0.f; G[bf][bk][3] = 0.f; }
  const int64_t n_tiles = (p.B + 15) / 16;
  const int64_t stride = (int64_t)gridDim.x * p.groups;
  if (active) {
    int64_t tile = (int64_t)blockIdx.x * p.groups + tg;
    WideBwdOps cur, nxt;
    if (tile < n_tiles) wide_bwd_fetch(cur, p, tile, c0, n, q, al);
    for (; tile < n_tiles; tile += stride) {
      if (tile + stride < n_tiles) wide_bwd_fetch(nxt, p, tile + stride, c0, n, q, al);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float b[4];
#pragma unroll
        for (int bk = 0; bk < 4; ++bk) {
          const float raw = (c0 + 16 * bk + n < p.D) ? cur.b[j][bk] : 0.f;
          b[bk] = fnorm ? ((c0 + 16 * bk + n < p.D) ? (raw - cur.mean[j]) * cur.rstd[j] : 0.f) : raw;
        }
#pragma unroll
        for (int bf = 0; bf < 4; ++bf) {
          db[bf] += cur.a[bf][j];
#pragma unroll
          for (int bk = 0; bk < 4; ++bk) G[bf][bk] = mfma16(cur.a[bf][j], b[bk], G[bf][bk]);
        }
      }
      cur = nxt;
    }
  }
  // ---- raw products -> gradients, per wave (db: lane (m, q) holds the sum over its samples of dz1[16 bf + m]) ----
#pragma unroll
  for (int bf = 0; bf < 4; ++bf) db[bf] = quad_sum16(db[bf]);
  if (q == 0) {
#pragma unroll
    for (int bf = 0; bf < 4; ++bf) sDb[wave][16 * bf + n] = db[bf];
  }
  wave_lds_sync();
  if (!active) return;
  float *slab = p.slabs + (size_t)((int64_t)blockIdx.x * p.groups + tg) * p.slab_stride + p.slab_col0;
  float dgam[4] = {0.f, 0.f, 0.f, 0.f}, dbet[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int bk = 0; bk < 4; ++bk) {
    const int k = c0 + 16 * bk + n;
    const bool kv = k < p.D;
    const int kc = kv ? k : p.D - 1;
    const float gam = fnorm ? p.params[p.fn_w + kc] : 1.f, bet = fnorm ? p.params[p.fn_b + kc] : 0.f;
#pragma unroll
    for (int bf = 0; bf < 4; ++bf) {
      const f32x4 dbv = ld4(&sDb[wave][16 * bf + 4 * q]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = 16 * bf + 4 * q + i;
        const float g = G[bf][bk][i];
        if (fnorm) {
          const float w = p.params[p.w1 + f * p.D + kc];
          dgam[bk] += w * g; dbet[bk] += w * dbv[i];
        }
        if (kv) slab[p.w1 + f * p.D + k] = gam * g + bet * dbv[i];
      }
    }
    if (fnorm) {
      const float dg = quad_sum16(dgam[bk]), dt = quad_sum16(dbet[bk]);
      if (q == 0 && kv) { slab[p.fn_w + k] = dg; slab[p.fn_b + k] = dt; }
    }
  }
}
