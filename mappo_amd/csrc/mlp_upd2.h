// mlp_upd2.h — the update kernel for in_dim <= 64 ("pair" formulation): forward + head gradient (external | PPO actor
// loss | value loss | trunk gradient in) + backward, like mlp_update_kernel (mlp_impl.h), but with TWO wavefronts per
// 32-sample tile.
//
// Why: with one wave per tile the 64x64 gradient accumulators need ~450 registers, which limits a CU to 4 waves (one
// per SIMD) — and a lone wave cannot hide its own LDS / MFMA latencies (scripts/exp_waves.py: 2 waves per SIMD run the
// per-tile work ~1.7x faster).  Here the two waves of a pair split every layer's 64 OUTPUT features 32 / 32:
//   wave `fh` of a pair owns rows [32 fh, 32 fh + 32) of every activation, of every dz and of every weight gradient.
// Each wave then carries half the accumulators (<= 256 registers), a workgroup holds 8 waves = 2 per SIMD, and the
// LDS footprint per tile is unchanged (the pair shares the tile's xhat / dz tiles).  What the split costs:
//   * LayerNorm statistics span both halves: each wave reduces its 32 features (mean, M2), the halves meet through a
//     64-float exchange buffer and combine with Chan's formula (forward), resp. add their partial sums (backward);
//   * a PAIR barrier (pair_sync: an LDS counter, not s_barrier) wherever one wave consumes rows its partner produced
//     (11 per tile for layer_N = 1).  Pairs are not synchronised with each other inside the tile loop, so the two
//     tiles that share a SIMD drift apart and fill each other's MFMA / VALU / LDS latencies.
//   * the head forward + per-sample loss (32 lanes of work) is done by wave 0 of the pair while wave 1 waits.
// Accumulated quantities are the RAW products (see raw_to_grad in mlp_impl.h); the epilogue is per wave.
#pragma once

#define XS 65            // row stride (floats) of the flat-commit staging area: lanes (s16, q) hit 64 distinct banks

// Barrier between the TWO waves of a pair (gfx950 has only the workgroup-wide s_barrier, and that would keep all 8
// waves of the workgroup in the same phase — the co-resident waves of a SIMD then want the MFMA pipe, or the VALU, at
// the same time).  Arrive = one LDS atomic increment, wait = poll until both arrivals of this epoch are in.  Both waves
// are resident for the life of the workgroup, so the wait cannot deadlock; a wave's LDS operations execute in issue
// order, so everything it wrote before arriving is visible to the partner once the partner sees the count.
struct PairSync {
  unsigned *cnt; unsigned epoch;
#ifdef MLP_STAMPS
  unsigned long long cyc;     // diagnostic build: cycles spent inside pair_sync
#endif
};
__device__ __forceinline__ void pair_sync(PairSync &ps, int lane) {
  // No fence: a workgroup-scope release would also wait for the global prefetch loads in flight (vmcnt), which is
  // exactly what must stay asynchronous.  Only the compiler has to keep the LDS accesses on their side of the barrier.
  asm volatile("" ::: "memory");
#ifdef MLP_STAMPS
  const unsigned long long t0_ = __builtin_readcyclecounter();
#endif
  ps.epoch += 2u;
  if (lane == 0) (void)__hip_atomic_fetch_add(ps.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(ps.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < ps.epoch) {
#ifdef EXP_SYNC_SLEEP
    __builtin_amdgcn_s_sleep(EXP_SYNC_SLEEP);
#endif
  }
#ifdef MLP_STAMPS
  ps.cyc += __builtin_readcyclecounter() - t0_;
#endif
  asm volatile("" ::: "memory");
}

template <bool WIDE>
struct HalfPrefetch {
  float v[WIDE ? 16 : 8];
  int n_valid;           // valid samples of the whole 32-sample tile
  bool flat;             // v = float4 chunks (lane + 64 j) of the contiguous [16][D] block of this wave's 16 samples
};

// Wave fh fetches samples [16 fh, 16 fh + 16) of the tile; lane (s16 = lane & 15, q = lane >> 4) ends up with features
// k = NV q + j of sample 16 fh + s16.  Flat mode as in prefetch_rows (coalesced float4s of one contiguous block).
template <bool WIDE>
__device__ __forceinline__ void prefetch_half(HalfPrefetch<WIDE> &pf, const float *__restrict__ x, const int32_t *__restrict__ rows,
                                              int64_t base, int64_t B, int D, int lane, int fh) {
  constexpr int NV = WIDE ? 16 : 8;
  const int s16 = lane & 15, q = lane >> 4, s = 16 * fh + s16;
  pf.n_valid = (int)max((int64_t)0, min((int64_t)TS, B - base));
  pf.flat = rows == nullptr && pf.n_valid == TS && (((uintptr_t)x) & 15) == 0;
  if (pf.flat) {
    const float4 *src4 = reinterpret_cast<const float4 *>(x + (base + 16 * fh) * D);
    const int n4 = 4 * D;                                  // float4s in the 16 x D block
#pragma unroll
    for (int j = 0; j < NV / 4; ++j) {
      const float4 t = src4[min(lane + 64 * j, n4 - 1)];
      pf.v[4 * j + 0] = t.x; pf.v[4 * j + 1] = t.y; pf.v[4 * j + 2] = t.z; pf.v[4 * j + 3] = t.w;
    }
    return;
  }
  const bool ok = s < pf.n_valid;
  const int64_t row = ok ? (rows ? (int64_t)rows[base + s] : base + s) : 0;
  const float *src = x + row * D + NV * q;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    pf.v[j] = 0.f;
    if (ok && NV * q + j < D) pf.v[j] = src[j];
  }
}

__device__ __forceinline__ float quad_sum(float v) { return xhalf_sum(xrow_sum(v)); }   // over the 4 lanes (q) of a sample

// tX[k][16 fh + s16] <- xhat0 (or the raw input).  tF: this wave's 16*XS floats of dead staging (flat mode); `dummy`: a
// dead LDS word that absorbs the stores of lanes with nothing to write (address select instead of a branch per
// element: hipcc otherwise wraps every store in its own exec-mask region and waits for every load on its own).
template <bool WIDE>
__device__ __forceinline__ void commit_half(float *tX, float *tF, float *dummy, const HalfPrefetch<WIDE> &pf, int D, int Dp, float inv_D,
                                            uint32_t magic, int lane, int fh, bool feature_norm) {
  constexpr int NV = WIDE ? 16 : 8;
  int ln = lane;
  asm volatile("" : "+v"(ln));        // opaque per tile: keeps hipcc from hoisting the lane predicates into SGPR pairs
  const int s16 = ln & 15, q = ln >> 4, s = 16 * fh + s16;
  float v[NV];
  if (pf.flat) {
#pragma unroll
    for (int j = 0; j < NV / 4; ++j) {
      const bool in = (ln + 64 * j) < 4 * D;                 // whole float4 chunks: 16*D is a multiple of 4
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int e = 4 * (ln + 64 * j) + c;
        const int r = (int)__umulhi((uint32_t)e, magic);
        float *dst = in ? tF + r * XS + (e - r * D) : dummy;
        *dst = pf.v[4 * j + c];
      }
    }
    wave_lds_sync();
    float t[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) t[j] = tF[s16 * XS + min(NV * q + j, D - 1)];      // all loads first, selected below
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = (NV * q + j < D) ? t[j] : 0.f;
  } else {
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = pf.v[j];             // slots beyond D hold 0
  }
  float mean = 0.f, rstd = 1.f;
  if (feature_norm) {
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) sum += v[j];
    mean = quad_sum(sum) * inv_D;
    float qq = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) { const float c = (NV * q + j < D) ? v[j] - mean : 0.f; qq += c * c; }
    rstd = 1.0f / sqrtf(quad_sum(qq) * inv_D + LN_EPS);
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int k = NV * q + j;
    float *dst = (k < Dp) ? tX + k * TP + s : dummy;
    *dst = (k < D) ? (v[j] - mean) * rstd : 0.f;
  }
}

// Pointer form: A(kk) = pa[kk * sa], B(kk) = pb[kk * sb] (LDS).  The operands run two k-steps ahead
// of their MFMA; the window may read up to 5 steps past n — rows that exist in LDS (the next matrix / tile) and are
// never fed to an MFMA.
__device__ __forceinline__ void mfma_chain_p(f32x16 &acc, int n, const float *pa, int sa, const float *pb, int sb) {
#if !defined(EXP_MFMA_NOPIPE)
  // (a second, interleaved accumulator — a dependent MFMA waits ~100 cycles for its predecessor — measured -2 % on the
  // critic kernel but cost the register-tighter actor kernel spills; with the two in one launch the actor is the bound)
  float a[6], b[6];
  a[0] = pa[0]; b[0] = pb[0]; a[1] = pa[sa]; b[1] = pb[sb];
  for (int k0 = 0; k0 < n; k0 += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j + 2] = pa[(j + 2) * sa]; b[j + 2] = pb[(j + 2) * sb];
      if (k0 + j < n) acc = mfma(a[j], b[j], acc);
    }
    a[0] = a[4]; b[0] = b[4]; a[1] = a[5]; b[1] = b[5];
    pa += 4 * sa; pb += 4 * sb;
  }
#else
#pragma unroll 8
  for (int kk = 0; kk < n; ++kk) acc = mfma(pa[kk * sa], pb[kk * sb], acc);
#endif
}

// Folded weights (pair kernel): the LayerNorm affine of a layer's input is folded into the staged copy,
//   W'[f][k] = W[f][k] * gamma[k],   b'[f] = b[f] + sum_k W[f][k] * beta[k]     (fold_affine below),
// so the forward operand is the tile's xhat itself and  W'^T dz = d xhat  directly in the backward.
// acc (32 features of this wave) += W'[row0 + i][k] . xhat[k][s];  sWr = sW + row0
__device__ __forceinline__ void layer_mfma1(f32x16 &acc, const float *sWr, const float *tin, int ksteps, int l31, int half) {
  mfma_chain_p(acc, ksteps, sWr + half * WP + l31, 2 * WP, tin + half * TP + l31, 2 * TP);
}

// head output (accumulator layout) from the folded head weights
__device__ __forceinline__ f32x16 head_forward1(const float *lds, const LdsMap &m, const float *tLast, int l31, int half) {
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 b = *reinterpret_cast<const float4 *>(lds + m.bh + 8 * q + 4 * half);
    acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
  }
  const float *sW = lds + m.wh;
  mfma_chain_p(acc, HID / 2, sW + half * HP + l31, 2 * HP, tLast + half * TP + l31, 2 * TP);
  return acc;
}

// In-place fold of the staged weights / biases (all waves of the workgroup; contains workgroup barriers).
template <int LN, int HEAD>
__device__ __forceinline__ void fold_affine(float *lds, const LdsMap &m, int D, int Dp, int wave, int n_waves, int lane) {
  // (1) biases, one job per wave at a time (lane = output feature): b' = b + W beta
  const int n_jobs = LN + 1 + (HEAD != 3 ? 1 : 0);
  for (int job = wave; job < n_jobs; job += n_waves) {
    if (job == 0) {
      float acc = lds[m.b1 + lane];
      for (int k = 0; k < D; ++k) acc += lds[m.w1 + k * WP + lane] * lds[m.fn_b + k];
      lds[m.b1 + lane] = acc;
    } else if (job <= LN) {
      const int l = job - 1;
      float acc = lds[m.b2[l] + lane];
      const float *sW = lds + m.w2[l], *sB = lds + ln_b_of<LN>(m, l);
#pragma unroll 8
      for (int k = 0; k < HID; ++k) acc += sW[k * WP + lane] * sB[k];
      lds[m.b2[l] + lane] = acc;
    } else if (lane < 32) {
      float acc = lds[m.bh + lane];
      const float *sW = lds + m.wh, *sB = lds + ln_b_of<LN>(m, LN);
#pragma unroll 8
      for (int k = 0; k < HID; ++k) acc += sW[k * HP + lane] * sB[k];
      lds[m.bh + lane] = acc;
    }
  }
  __syncthreads();
  // (2) weights: row k of every k-major matrix times gamma_in[k]
  const int nthr = blockDim.x, tid = threadIdx.x;
  for (int e = tid; e < Dp * HID; e += nthr) { const int k = e >> 6, f = e & 63; lds[m.w1 + k * WP + f] *= lds[m.fn_w + k]; }
#pragma unroll
  for (int l = 0; l < LN; ++l)
    for (int e = tid; e < HID * HID; e += nthr) { const int k = e >> 6, f = e & 63; lds[m.w2[l] + k * WP + f] *= lds[ln_w_of<LN>(m, l) + k]; }
  if (HEAD != 3)
    for (int e = tid; e < HID * 32; e += nthr) { const int k = e >> 5, a = e & 31; lds[m.wh + k * HP + a] *= lds[ln_w_of<LN>(m, LN) + k]; }
  __syncthreads();
}

__device__ __forceinline__ void init_bias1(f32x16 &acc, const float *sB, int fh, int half) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 b = vec4_of(sB, fh, q, half);
    acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
  }
}

template <int LN>
struct TileStats1 {
  float mean[LN + 1], rstd[LN + 1];
  uint32_t pos[LN + 1];      // bit r: post-activation value of accumulator register r > 0
};

// act + LayerNorm over 64 features of which this wave holds 32: local (mean, M2), exchange, Chan combine; xhat -> tile.
// Contains one pair barrier (exchange); the caller places the barrier that publishes the tile.
template <bool RELU>
__device__ __forceinline__ void act_ln_to_tile1(f32x16 &acc, float *tile, float *xch, PairSync &ps, int fh, int lane, int l31, int half,
                                                float &mean, float &rstd, uint32_t &pos) {
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = act_fwd<RELU>(acc[r]); s += acc[r]; }
  const float m_loc = xhalf_sum(s) * (1.f / 32.f);
  float q = 0.f;
  uint32_t mk = 0u;
#pragma unroll
  for (int r = 0; r < 16; ++r) { const float c = acc[r] - m_loc; q += c * c; mk |= (acc[r] > 0.f ? 1u : 0u) << r; }
  const float M2_loc = xhalf_sum(q);
  pos = mk;
  if (half == 0) { xch[fh * 64 + l31] = m_loc; xch[fh * 64 + 32 + l31] = M2_loc; }
  pair_sync(ps, lane);
  const float m_o = xch[(1 - fh) * 64 + l31], M2_o = xch[(1 - fh) * 64 + 32 + l31];
  const float d = m_loc - m_o;
  mean = 0.5f * (m_loc + m_o);
  rstd = 1.0f / sqrtf((M2_loc + M2_o + 16.f * d * d) * (1.f / HID) + LN_EPS);      // Chan: n_a n_b / (n_a + n_b) = 16
#pragma unroll
  for (int r = 0; r < 16; ++r) tile[(32 * fh + ROWMAP(r, half)) * TP + l31] = (acc[r] - mean) * rstd;
}

// sum over the 32 samples of row (row0 + l31) of a tile; every lane of the wave returns the sum of ITS l31 row
__device__ __forceinline__ float half_row_sum(const float *tile, int row0, int l31, int half) {
  const float *rp = tile + (row0 + l31) * TP + 16 * half;
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int j = 0; j < 16; j += 2) { s0 += rp[j]; s1 += rp[j + 1]; }
  return xhalf_sum(s0 + s1);
}

// LayerNorm + activation backward for the 32 features of this wave (see ln_act_backward).  One pair barrier
// (exchange of the partial sums); the caller places the barrier that publishes dz.
template <bool RELU, bool AFFINE>
__device__ __forceinline__ void ln_act_backward1(f32x16 &dH, float *tile, float *xch, PairSync &ps, int fh, float mean, float rstd,
                                                 uint32_t pos, const float *sG, float &gG, float &gB, int lane, int l31, int half) {
  const int row0 = 32 * fh;
  float xh[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) xh[r] = tile[(row0 + ROWMAP(r, half)) * TP + l31];
  if (AFFINE) {            // own rows only: wave-private scratch use of the tile
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[(row0 + ROWMAP(r, half)) * TP + l31] = dH[r];
    wave_lds_sync();
    gB += half_row_sum(tile, row0, l31, half);
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[(row0 + ROWMAP(r, half)) * TP + l31] = dH[r] * xh[r];
    wave_lds_sync();
    gG += half_row_sum(tile, row0, l31, half);
    wave_lds_sync();
  }
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 g4 = vec4_of(sG, fh, q, half);
    const float gq[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int r = 4 * q + c;
      const float dxh = AFFINE ? dH[r] * gq[c] : dH[r];      // folded weights: W'^T dz is d xhat already
      dH[r] = dxh;
      m1 += dxh;
      m2 += dxh * xh[r];
    }
  }
  m1 = xhalf_sum(m1);
  m2 = xhalf_sum(m2);
  if (half == 0) { xch[fh * 64 + l31] = m1; xch[fh * 64 + 32 + l31] = m2; }
  pair_sync(ps, lane);
  m1 = (m1 + xch[(1 - fh) * 64 + l31]) * (1.f / HID);
  m2 = (m2 + xch[(1 - fh) * 64 + 32 + l31]) * (1.f / HID);
  const float inv_rstd = 1.0f / rstd;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float da = rstd * (dH[r] - m1 - xh[r] * m2);
    if (RELU) {
      dH[r] = ((pos >> r) & 1u) ? da : 0.f;
    } else {
      const float a = xh[r] * inv_rstd + mean;
      dH[r] = da * (1.f - a * a);
    }
    tile[(row0 + ROWMAP(r, half)) * TP + l31] = dH[r];
  }
}

// Epilogue helpers.  The LDS copy of the weights is folded with gamma, so the RAW consumer weights come from global
// memory (row-major [n_rows][ldw]); raw_w_load issues those loads for all tiles of a product up front — the caller
// loads every product's weights before transforming the first one, so the epilogue pays one global latency, not five.
// Tiles: rows f = frow0 + ROWMAP(r, half), columns k = kcol0 + 32 tj + l31.
template <int NTJ>
__device__ __forceinline__ void raw_w_load(float (&w)[NTJ][16], const float *__restrict__ gW, int ldw, int n_rows, int frow0, int K,
                                           int kcol0, int n_tj, int l31, int half) {
#pragma unroll
  for (int tj = 0; tj < NTJ; ++tj) {
    const int k = kcol0 + 32 * tj + l31;
    const int kc = (k < K) ? k : 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = frow0 + ROWMAP(r, half);
      const float t = gW[min(f, n_rows - 1) * ldw + kc];
      w[tj][r] = (tj < n_tj && k < K && f < n_rows) ? t : 0.f;
    }
  }
}

// raw products -> gradient partials (see raw_to_grad in mlp_impl.h).  dbv: db of row frow0 + l31 (any half).
// dg / dt (per tile): this wave's partial d gamma / d beta of column k, valid in every lane of that l31.
template <int NTJ>
__device__ __forceinline__ void raw_to_grad1(f32x16 (&g)[NTJ], const float (&w)[NTJ][16], float dbv, float *scr, const float *sG,
                                             const float *sBt, int K, int kcol0, int n_tj, int lane, int l31, int half,
                                             float (&dg)[NTJ], float (&dt)[NTJ]) {
  if (lane < 32) scr[lane] = dbv;
  wave_lds_sync();
#pragma unroll
  for (int tj = 0; tj < NTJ; ++tj) {
    dg[tj] = 0.f; dt[tj] = 0.f;
    if (tj >= n_tj) continue;
    const int k = kcol0 + 32 * tj + l31;
    const bool valid = k < K;
    const int kc = valid ? k : 0;
    const float gam = valid ? sG[kc] : 0.f, bet = valid ? sBt[kc] : 0.f;
    float d[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r] = scr[ROWMAP(r, half)];
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      a0 += w[tj][r] * g[tj][r];
      a1 += w[tj][r] * d[r];
      g[tj][r] = gam * g[tj][r] + bet * d[r];
    }
    dg[tj] = xhalf_sum(a0);
    dt[tj] = xhalf_sum(a1);
  }
  wave_lds_sync();
}

// vector ids of the per-wave vector partials (epilogue): b1 ln1_w ln1_b | (b2 ln2_w ln2_b) x LN | bh | fn_w fn_b
template <int LN> struct VecIds { static constexpr int BH = 3 * (LN + 1), FNW = BH + 1, FNB = BH + 2, N = BH + 3; };

// Workgroup `bid` of the `nb` workgroups that share this network's B rows (blockIdx / gridDim of a plain launch; a
// sub-range of the grid in the dual launch below).  red_smem: 64 doubles, pair_cnt: 4 words of static LDS.
template <bool RELU, int LN, int HEAD, bool WIDE>
__device__ __forceinline__ void update2_body(const UpdArgs &p, float *lds, double *red_smem, unsigned *pair_cnt, const int bid,
                                             const int nb) {
  const NetOff &o = p.off;
  const LdsMap &m = p.map;
  const int n_pairs = blockDim.x / (2 * WAVE);
  const int lane = threadIdx.x & (WAVE - 1), l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));   // wave-uniform: tile indices and bases stay scalar
  const int pair = wave >> 1, fh = wave & 1, row0 = 32 * fh;     // partners sit on different SIMDs; a SIMD hosts two TILES
  if (threadIdx.x < 4) pair_cnt[threadIdx.x] = 0u;              // published by the barrier after the weight staging
  PairSync ps = {};
  ps.cnt = pair_cnt + pair;
  const int D = p.desc.in_dim, Dp = (D + 1) & ~1, A = p.desc.out_dim;
  const uint32_t magic = (uint32_t)(0x100000000ull / (uint32_t)D) + 1u;
  const float inv_D = 1.0f / (float)D;
  const bool fnorm = p.desc.use_feature_norm != 0;
  // Pairs are independent workers (their barriers are pair-local): worker w = pair * gridDim + block takes tiles
  // w, w + W, w + 2W, ...  A tile count that is not a multiple of W then leaves its remainder spread over ALL CUs
  // (first the pairs 0, then the pairs 1, ...) instead of a few workgroups running one more full round.
  const int64_t n_tiles = (p.B + TS - 1) / TS;
  const int64_t tile_stride = (int64_t)nb * n_pairs;
  const int64_t tile0 = (int64_t)pair * nb + bid;
  HalfPrefetch<WIDE> pf;
  LossPrefetch lp;
  STAMP_DECL
  {
    const int64_t base0 = tile0 * TS;
    prefetch_half(pf, p.x, p.rows, base0, p.B, D, lane, fh);
    const int nv = pf.n_valid;
    const int64_t row = (lane < nv) ? (p.rows ? (int64_t)p.rows[base0 + lane] : base0 + lane) : 0;
    if (fh == 0) prefetch_loss<HEAD>(lp, p, row, nv, lane, A); else { lp.f0 = lp.f1 = lp.f2 = lp.f3 = 0.f; lp.dead = 0u; }
  }
  stage_all_weights_1shot<LN>(lds, m, p.params, o, p.desc);
  __syncthreads();
#ifndef EXP_NOFOLDPASS
  fold_affine<LN, HEAD>(lds, m, D, Dp, wave, blockDim.x / WAVE, lane);
#endif
  STAMP(0);   // staging
  float *tX = lds + m.tiles + pair * m.wave_stride;
  float *tH = tX + m.x_rows * TP;
  float *tZ = tH + (LN + 1) * HID * TP;
  float *tF = tH + fh * 16 * XS;                         // flat-commit staging inside the (dead) first activation tile
  float *xch = lds + m.scratch + pair * 128;             // [fh][2][32] exchange buffer of the pair

  LossScales ls = {0.f, 0.f, 0.f, 1.f};
  if (HEAD == 1 || HEAD == 2) ls = loss_scales(p.cfg, p.mb_moments, p.vn_state);
  double lacc[4] = {0.0, 0.0, 0.0, 0.0};

  // ---- raw-product accumulators of this wave's 32 output rows ----
  f32x16 gWh[1], gW2[LN > 0 ? LN : 1][2], gW1[2];
  float gBh = 0.f, gB[LN + 1], gLnW = 0.f, gLnB = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    gWh[0][r] = 0.f; gW1[0][r] = 0.f; gW1[1][r] = 0.f;
#pragma unroll
    for (int l = 0; l < LN; ++l) { gW2[l][0][r] = 0.f; gW2[l][1][r] = 0.f; }
  }
#pragma unroll
  for (int l = 0; l <= LN; ++l) gB[l] = 0.f;

  for (int64_t tile = tile0; tile < n_tiles; tile += tile_stride) {
    const int64_t base = tile * TS;
    const int n_valid = pf.n_valid;
    const LossPrefetch cur = lp;
    TileStats1<LN> st;
    commit_half(tX, tF, tH + 2 * 16 * XS, pf, D, Dp, inv_D, magic, lane, fh, fnorm);
    STAMP(16);  // commit
    {
      const int64_t nbase = (tile + tile_stride) * TS;
      prefetch_half(pf, p.x, p.rows, nbase, p.B, D, lane, fh);                   // next tile, hidden under the MFMAs below
      const int nv = pf.n_valid;
      const int64_t row = (lane < nv) ? (p.rows ? (int64_t)p.rows[nbase + lane] : nbase + lane) : 0;
      if (fh == 0) prefetch_loss<HEAD>(lp, p, row, nv, lane, A);
    }
    STAMP(17);  // prefetch issue
    pair_sync(ps, lane);                                     // both halves of tX written
    STAMP(1);   // sync after commit
    // ---- trunk forward (this wave: features row0..row0+31 of every layer) ----
    {
      f32x16 acc;
      init_bias1(acc, lds + m.b1, fh, half);
      layer_mfma1(acc, lds + m.w1 + row0, tX, Dp / 2, l31, half);
      act_ln_to_tile1<RELU>(acc, tH, xch, ps, fh, lane, l31, half, st.mean[0], st.rstd[0], st.pos[0]);
      pair_sync(ps, lane);
#pragma unroll
      for (int l = 0; l < LN; ++l) {
        init_bias1(acc, lds + m.b2[l], fh, half);
        layer_mfma1(acc, lds + m.w2[l] + row0, tH + l * HID * TP, HID / 2, l31, half);
        act_ln_to_tile1<RELU>(acc, tH + (l + 1) * HID * TP, xch, ps, fh, lane, l31, half, st.mean[l + 1], st.rstd[l + 1], st.pos[l + 1]);
        pair_sync(ps, lane);
      }
    }
    float *tLast = tH + LN * HID * TP;
    STAMP(2);   // trunk forward

    // ---- head gradient into tZ[s][a] (wave 0 of the pair; its partner waits at the barrier) ----
    if (HEAD != 3) {
      if (fh == 0) {
        if (HEAD == 0) {
          for (int e = lane; e < TS * A; e += WAVE) {
            const int s = e / A, a = e - s * A;
            tZ[s * TP + a] = (s < n_valid) ? p.dout[base * A + e] : 0.f;
          }
        } else if (HEAD == 2) {
          // critic (out_dim 1): the head is one 64-term dot product per sample — on the VALU (each half sums 32 features),
          // not a 32-step MFMA chain for one useful output row
          const float *sW = lds + m.wh;
          float a0 = 0.f, a1 = 0.f;
#pragma unroll
          for (int f = 0; f < 32; f += 2) {
            a0 += sW[(32 * half + f) * HP] * tLast[(32 * half + f) * TP + l31];
            a1 += sW[(32 * half + f + 1) * HP] * tLast[(32 * half + f + 1) * TP + l31];
          }
          const float v = xhalf_sum(a0 + a1) + lds[m.bh];
          if (lane < TS) {
            float dvv = 0.f;
            if (lane < n_valid) dvv = critic_loss_lane(v, cur.f0, cur.f1, cur.f2, p.cfg, ls, lacc);
            tZ[lane * TP] = dvv;
          }
        } else {
          const f32x16 z = head_forward1(lds, m, tLast, l31, half);
          STAMP(18);  // head forward (MFMA chain)
          if (HEAD == 1) {
            head_to_tile(tZ, z, A, l31, half);
            wave_lds_sync();
            STAMP(19);  // logits -> tile
            if (lane < TS) {
              float *zl = tZ + lane * TP;
              if (lane < n_valid) {
                actor_loss_lane(zl, A, cur.dead, (int)cur.f0, cur.f1, cur.f2, cur.f3, p.cfg, ls.scale_pi, lacc);
              } else {
                for (int a = 0; a < A; ++a) zl[a] = 0.f;
              }
            }
            STAMP(20);  // per-sample loss
          } else {
            if (lane < TS) {
              float dvv = 0.f;
              if (lane < n_valid) dvv = critic_loss_lane(z[0], cur.f0, cur.f1, cur.f2, p.cfg, ls, lacc);
              tZ[lane * TP] = dvv;
            }
          }
        }
      }
      pair_sync(ps, lane);
    }
    STAMP(3);   // head forward + loss

    // ---- (A) raw head products gWh[a][f] (columns f = row0 + l31), (B) d h_last rows row0.. ----
    f32x16 dH;
#pragma unroll
    for (int r = 0; r < 16; ++r) dH[r] = 0.f;
    if (HEAD == 2) {
      // out_dim 1 on the VALU: G[f] = sum_s dv[s] xhat[f][s] (lane = feature row0 + l31, each half 16 samples),
      // db = sum_s dv[s], d h[f][s] = Wh'[f] dv[s]
      const float *rp = tLast + (row0 + l31) * TP + 16 * half;
      float g0 = 0.f, g1 = 0.f, b0 = 0.f;
#pragma unroll
      for (int jj = 0; jj < 16; jj += 2) {
        const float d0 = tZ[(16 * half + jj) * TP], d1 = tZ[(16 * half + jj + 1) * TP];
        g0 += d0 * rp[jj]; g1 += d1 * rp[jj + 1];
        b0 += d0 + d1;
      }
      const float gsum = xhalf_sum(g0 + g1), bsum = xhalf_sum(b0);
      if (half == 0) gWh[0][0] += gsum;                  // accumulator element (a = 0, f = row0 + l31): lane (l31, half 0), register 0
      if (fh == 0) gBh += bsum;
      const float dvs = tZ[l31 * TP];
      const float *sW = lds + m.wh;
#pragma unroll
      for (int r = 0; r < 16; ++r) dH[r] = sW[(row0 + ROWMAP(r, half)) * HP] * dvs;
    } else if (HEAD != 3) {
      float bsum = 0.f;
#pragma unroll 2
      for (int ss = 0; ss < TS / 2; ++ss) {
        const int s = 2 * ss + half;
        const float av = (l31 < A) ? tZ[s * TP + l31] : 0.f;
        bsum += av;
        gWh[0] = mfma(av, tLast[(row0 + l31) * TP + s], gWh[0]);
      }
      if (fh == 0) gBh += xhalf_sum(bsum);
      const float *sW = lds + m.wh;
      for (int kk = 0; kk < (A + 1) / 2; ++kk) {
        const int a = 2 * kk + half;
        const float b = (a < A) ? tZ[l31 * TP + a] : 0.f;
        dH = mfma(sW[(row0 + l31) * HP + a], b, dH);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (l31 < n_valid) dH[r] = p.dHT[(int64_t)(row0 + ROWMAP(r, half)) * p.B + base + l31];
    }
    STAMP(4);   // head grads (A), (B)
    // ---- hidden layers, last to first ----
#pragma unroll
    for (int l = LN; l >= 1; --l) {
      float *tCur = tH + l * HID * TP;          // xhat of this layer's LayerNorm -> dz
      float *tPrev = tH + (l - 1) * HID * TP;   // xhat of the layer's input
      if (HEAD == 3 && l == LN)
        ln_act_backward1<RELU, true>(dH, tCur, xch, ps, fh, st.mean[l], st.rstd[l], st.pos[l], lds + m.ln2_w[l - 1], gLnW, gLnB, lane, l31, half);
      else
        ln_act_backward1<RELU, false>(dH, tCur, xch, ps, fh, st.mean[l], st.rstd[l], st.pos[l], lds + m.ln2_w[l - 1], gLnW, gLnB, lane, l31, half);
      pair_sync(ps, lane);                          // all 64 rows of dz published
      gB[l] += half_row_sum(tCur, row0, l31, half);
      STAMP(5);   // LN + act backward (hidden)
#pragma unroll 2
      for (int ss = 0; ss < TS / 2; ++ss) {
        const int s = 2 * ss + half;
        const float a = tCur[(row0 + l31) * TP + s];
        gW2[l - 1][0] = mfma(a, tPrev[l31 * TP + s], gW2[l - 1][0]);
        gW2[l - 1][1] = mfma(a, tPrev[(32 + l31) * TP + s], gW2[l - 1][1]);
      }
      STAMP(6);   // dW2
#pragma unroll
      for (int r = 0; r < 16; ++r) dH[r] = 0.f;
      {
        const float *sW = lds + m.w2[l - 1] + (row0 + l31) * WP;
        mfma_chain_p(dH, HID / 2, sW + half, 2, tCur + half * TP + l31, 2 * TP);
      }
      STAMP(7);   // dH (hidden)
    }
    // ---- layer 1 ----
    {
      float *tCur = tH;
      if (HEAD == 3 && LN == 0)
        ln_act_backward1<RELU, true>(dH, tCur, xch, ps, fh, st.mean[0], st.rstd[0], st.pos[0], lds + m.ln1_w, gLnW, gLnB, lane, l31, half);
      else
        ln_act_backward1<RELU, false>(dH, tCur, xch, ps, fh, st.mean[0], st.rstd[0], st.pos[0], lds + m.ln1_w, gLnW, gLnB, lane, l31, half);
      wave_lds_sync();                          // gW1 / db read only this wave's own dz rows
      gB[0] += half_row_sum(tCur, row0, l31, half);
      STAMP(8);   // LN + act backward (layer 1)
      // rows k >= Dp of the tX region belong to the activation tiles: finite values whose columns are never reduced
#pragma unroll 2
      for (int ss = 0; ss < TS / 2; ++ss) {
        const int s = 2 * ss + half;
        const float a = tCur[(row0 + l31) * TP + s];
        gW1[0] = mfma(a, tX[l31 * TP + s], gW1[0]);
        if (WIDE) gW1[1] = mfma(a, tX[(32 + l31) * TP + s], gW1[1]);
      }
      STAMP(9);   // dW1
    }
    pair_sync(ps, lane);                            // tiles free for the next commit
    STAMP(10);
  }

  // ---- loss statistics of this workgroup ----
  if (HEAD == 1 || HEAD == 2) {
    block_sum<4>(lacc, red_smem);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double *q = p.partials + (size_t)bid * 4 + k;
        *q = p.cfg.accumulate_partials ? *q + lacc[k] : lacc[k];
      }
    }
  }

  // ---- raw products -> gradient partials (per wave, registers) ----
  // Per-wave partials of the per-feature vectors: ve0[i] = entry l31, ve1[i] = entry 32 + l31 of vector id i.
  typedef VecIds<LN> V;
  float ve0[V::N], ve1[V::N];
#pragma unroll
  for (int i = 0; i < V::N; ++i) { ve0[i] = 0.f; ve1[i] = 0.f; }
  auto put_rows = [&](int i, float val) { ve0[i] = (fh == 0) ? val : 0.f; ve1[i] = (fh == 1) ? val : 0.f; };   // entry row0 + l31
  {
    float *scr = lds + m.scratch + n_pairs * 128 + wave * 32;
    // raw consumer weights of every product of this wave, all global loads in flight together
    float wH[1][16], w2[LN > 0 ? LN : 1][2][16], w1[2][16];
    if (HEAD != 3) raw_w_load<1>(wH, p.params + o.wh, HID, A, 0, HID, row0, 1, l31, half);
#pragma unroll
    for (int l = 0; l < LN; ++l) raw_w_load<2>(w2[l], p.params + o.w2[l], HID, HID, row0, HID, 0, 2, l31, half);
    if (fnorm) raw_w_load<2>(w1, p.params + o.w1, D, HID, row0, D, 0, WIDE ? 2 : 1, l31, half);
    put_rows(0, gB[0]);
#pragma unroll
    for (int l = 0; l < LN; ++l) put_rows(3 + 3 * l, gB[l + 1]);
    // last LayerNorm: consumer = head (Wh rows a, columns k = row0 + l31), or accumulated directly (HEAD 3)
    {
      constexpr int iw = (LN == 0) ? 1 : 4 + 3 * (LN - 1), ib = iw + 1;
      if (HEAD != 3) {
        // db_h[a] lives in lanes l31 = a of wave fh = 0 only; its partner needs it too: through the exchange buffer
        if (fh == 0 && lane < 32) xch[lane] = (l31 < A) ? gBh : 0.f;
        __syncthreads();
        const float dbh = xch[l31];
        float dg[1], dt[1];
        raw_to_grad1<1>(gWh, wH, dbh, scr, lds + ln_w_of<LN>(m, LN), lds + ln_b_of<LN>(m, LN), HID, row0, 1, lane, l31, half, dg, dt);
        put_rows(iw, dg[0]);
        put_rows(ib, dt[0]);
        ve0[V::BH] = (fh == 0) ? dbh : 0.f;
      } else {
        put_rows(iw, gLnW);
        put_rows(ib, gLnB);
      }
    }
    // LayerNorm l (tile tH[l]) feeds hidden layer l: products gW2[l] (rows row0.., all 64 columns)
#pragma unroll
    for (int l = LN - 1; l >= 0; --l) {
      const int iw = (l == 0) ? 1 : 4 + 3 * (l - 1), ib = iw + 1;
      float dg[2], dt[2];
      raw_to_grad1<2>(gW2[l], w2[l], gB[l + 1], scr, lds + ln_w_of<LN>(m, l), lds + ln_b_of<LN>(m, l), HID, 0, 2, lane, l31, half, dg, dt);
      ve0[iw] = dg[0]; ve1[iw] = dg[1]; ve0[ib] = dt[0]; ve1[ib] = dt[1];
    }
    if (fnorm) {
      float dg[2], dt[2];
      raw_to_grad1<2>(gW1, w1, gB[0], scr, lds + m.fn_w, lds + m.fn_b, D, 0, WIDE ? 2 : 1, lane, l31, half, dg, dt);
      ve0[V::FNW] = dg[0]; ve1[V::FNW] = dg[1]; ve0[V::FNB] = dt[0]; ve1[V::FNB] = dt[1];
    }
  }
  STAMP(11);

  // ---- reduce through LDS.  Weight tiles: pairs take turns on n_regions copies of the flat parameter range (the two
  // waves of a pair own disjoint rows).  Vectors: every wave fills its own [V::N][64] slot (zero where it has no entry).
  __syncthreads();
  const int P = p.p_red;
  float *red0 = lds + m.tiles;
  const int n_reg = p.n_regions;
  {
    float *vslot = red0 + n_reg * P + wave * (V::N * 64);
#pragma unroll
    for (int i = 0; i < V::N; ++i) vslot[i * 64 + lane] = half ? ve1[i] : ve0[i];
  }
  for (int round = 0; round < (n_pairs + n_reg - 1) / n_reg; ++round) {
    if (pair / n_reg == round) {
      float *red = red0 + (pair % n_reg) * P;
      const bool first = (round == 0);
      auto red_tile = [&](const f32x16 &acc, int idx0, int ld, bool valid) {
        if (!valid) return;
        float *q = red + idx0;
        if (first) {
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
      for (int tj = 0; tj < 2; ++tj) {
        const int col = 32 * tj + l31, rowq = row0 + 4 * half;
        if (tj == 0 || WIDE) red_tile(gW1[tj], o.w1 + rowq * D + col, D, col < D);
#pragma unroll
        for (int l = 0; l < LN; ++l) red_tile(gW2[l][tj], o.w2[l] + rowq * HID + col, HID, true);
      }
      if (HEAD != 3) {                                   // head: rows a = ROWMAP(r, half) < A, columns row0 + l31
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { const int a = ROWMAP(r, half); old[r] = (!first && a < A) ? red[o.wh + a * HID + row0 + l31] : 0.f; }
#pragma unroll
        for (int r = 0; r < 16; ++r) { const int a = ROWMAP(r, half); if (a < A) red[o.wh + a * HID + row0 + l31] = old[r] + gWh[0][r]; }
      }
    }
    __syncthreads();
  }
  // vectors: sum of the waves' slots -> region 0 (and zero in the other regions)
  {
    const int n_waves = blockDim.x / WAVE;
    const float *vs = red0 + n_reg * P;
    for (int t = threadIdx.x; t < V::N * 64; t += blockDim.x) {
      const int i = t >> 6, k = t & 63;
      float s = 0.f;
      for (int w = 0; w < n_waves; ++w) s += vs[w * (V::N * 64) + t];
      int off = -1;
      if (i == 0) off = o.b1 + k;
      else if (i == 1) off = o.ln1_w + k;
      else if (i == 2) off = o.ln1_b + k;
      else if (i < V::BH) {
        const int l = (i - 3) / 3, c = (i - 3) - 3 * l;
        const int b2 = (l == 0) ? o.b2[0] : o.b2[LN > 1 ? 1 : 0], w2 = (l == 0) ? o.ln2_w[0] : o.ln2_w[LN > 1 ? 1 : 0],
                  bb = (l == 0) ? o.ln2_b[0] : o.ln2_b[LN > 1 ? 1 : 0];
        off = (c == 0 ? b2 : (c == 1 ? w2 : bb)) + k;
      } else if (i == V::BH) { if (HEAD != 3 && k < A) off = o.bh + k; }
      else if (fnorm && k < D) off = (i == V::FNW ? o.fn_w : o.fn_b) + k;
      if (off >= 0) {
        red0[off] = s;
        if (n_reg > 1) red0[P + off] = 0.f;
      }
    }
  }
  __syncthreads();
  STAMP(12);    // block reduction through LDS
  float *slab = p.slabs + (size_t)bid * p.slab_stride + p.slab_col0;
  if (n_reg > 1) {
    for (int e = threadIdx.x; e < P; e += blockDim.x) slab[e] = red0[e] + red0[P + e];
  } else {
    for (int e = threadIdx.x; e < P; e += blockDim.x) slab[e] = red0[e];
  }
  STAMP(13);    // slab write
#ifdef MLP_STAMPS
  st_acc_[14] = ps.cyc;      // (informational: contained in the phases above)
#endif
  STAMP_FLUSH();
}

template <bool RELU, int LN, int HEAD, bool WIDE>
__global__ __launch_bounds__(512, 1) void mlp_update2_kernel(UpdArgs p) {
  extern __shared__ __align__(16) float lds[];
  __shared__ double red_smem[16 * 4];
  __shared__ unsigned pair_cnt[4];
  update2_body<RELU, LN, HEAD, WIDE>(p, lds, red_smem, pair_cnt, blockIdx.x, gridDim.x);
}

// Actor AND critic update in ONE launch: workgroups [0, nA) run the actor's update (HEAD 1), [nA, nA + nC) the critic's
// (HEAD 2).  Each network alone leaves the chip with a ragged tail (2 400 tiles on 1 024 pair-workers: 2.3 rounds of
// work take 3 rounds of time, twice per PPO update, plus two staging / epilogue phases); side by side on half the CUs
// each they take 5 rounds and one staging / epilogue phase.  The two bodies share nothing (own slab columns, own
// loss partials), so nothing needs ordering inside the launch.
struct DualArgs {
  UpdArgs a, c;
  int nA, nC;
};
template <bool RELU, int LN, bool WIDE_A, bool WIDE_C>
__global__ __launch_bounds__(512, 1) void mlp_update2_dual_kernel(DualArgs d) {
  extern __shared__ __align__(16) float lds[];
  __shared__ double red_smem[16 * 4];
  __shared__ unsigned pair_cnt[4];
  const int bid = blockIdx.x;
  if (bid < d.nA) update2_body<RELU, LN, 1, WIDE_A>(d.a, lds, red_smem, pair_cnt, bid, d.nA);
  else update2_body<RELU, LN, 2, WIDE_C>(d.c, lds, red_smem, pair_cnt, bid - d.nA, d.nC);
}
