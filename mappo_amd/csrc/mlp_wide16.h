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
#include "insert_core.h"
#include "mlp_wide16_args.h"

// experiment hooks (scripts/exp_build.py): -DWIDE_EXP_NOMFMA replaces the matrix instruction by one add (what is left is the
// non-MFMA time), -DWIDE_EXP_ROW0 makes every tile read rows 0..15 (no HBM traffic)
#ifdef WIDE_EXP_NOMFMA
#define WIDE_MFMA(a, b, c) ((c) + (a) * (b))
#else
#define WIDE_MFMA(a, b, c) mfma16(a, b, c)
#endif

// b1'[f] = b1[f] + sum_k W1[f][k] beta0[k]   (64 rows x TPR threads: the whole workgroup)
template <int TPR>
__device__ __forceinline__ void wide16_fold_bias(const Wide16Args &p, float *sB) {
  const int tid = threadIdx.x, f = tid / TPR, part = tid % TPR;
  float acc = 0.f;
  if (p.fn_b >= 0) {
    const float *w = p.params + p.w1 + (size_t)f * p.D, *bt = p.params + p.fn_b;
    for (int k0 = part; k0 < p.D; k0 += 8 * TPR) {
      float wv[8], bv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int k = min(k0 + TPR * j, p.D - 1); wv[j] = w[k]; bv[j] = bt[k]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) if (k0 + TPR * j < p.D) acc += wv[j] * bv[j];
    }
  }
#pragma unroll
  for (int off = 1; off < TPR; off <<= 1) acc += __shfl_xor(acc, off, WAVE);
  if (part == 0) sB[f] = p.params[p.b1 + f] + acc;
}

__device__ __forceinline__ f32x4 ld4u(const float *ptr) {             // 16-byte load from a 4-byte aligned address
  const f32x4_u l = *reinterpret_cast<const f32x4_u *>(ptr);
  f32x4 r; r[0] = l[0]; r[1] = l[1]; r[2] = l[2]; r[3] = l[3];
  return r;
}

// The tile loop of layer 1 for a workgroup of NW waves (8 or 4), W1' streamed through LDS in 64-column chunks (rollout-sized
// batches: a workgroup sees one or two tiles per wave, too few to amortise staging all of W1): for every 16-sample tile of this
// wave calls tail(acc, i, ok, mean, rstd) with acc = z1 of sample i (accumulator layout: lane (n, q) holds features
// 16 b + 4 q + r).  sW: [2][64 * RS16] chunk buffers, sB: [64] folded bias.  pre() runs once, after the first loads are issued
// (the caller's own staging overlaps their latency).
//   * Latency is what matters here (a launch is a handful of chunks long): the chunk fetches run TWO chunks ahead of the MFMAs
//     (two register sets, the LDS tile stays double buffered), the folded bias b1' = b1 + W1 beta0 is accumulated by the
//     staging threads from the chunks they load anyway (no separate pass over W1), and the row block is read before anything
//     else.
//   * Per-element work is kept off the main path as in the training kernel below: 1/std on the 16 accumulators, packed
//     statistics, row-end clamping / masking only for the last chunk.
template <int NCH, int NW, class Pre, class Tail>
__device__ __forceinline__ void wide16_layer1(const Wide16Args &p, float (*sW)[HID * RS16], float *sB, const int bid, const int nb,
                                              Pre &&pre, Tail &&tail) {      // workgroup `bid` of the `nb` that share this network
  constexpr int TPR = NW, CW = 64 / NW, CV = CW / 4;          // threads per weight row, floats (vectors) per thread and chunk
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = p.D;
  const bool fnorm = p.fn_w >= 0;
  const bool al4 = (D & 3) == 0;                             // every 4-column group is inside or outside a row as a whole
  const float inv_D = 1.0f / (float)D;
  const int64_t n_tiles = (p.B + 15) / 16;
  const int64_t n_groups = (n_tiles + NW - 1) / NW;           // NW tiles (one per wave) share the weight stream
  const int c_last = (D + 63) / 64 - 1;                       // the chunk that holds the row end (>= 1: in_dim > 64)

  // this thread's share of a weight chunk: row wf, columns CW wp .. CW wp + CW - 1
  const int wf = tid / TPR, wp = tid % TPR;
  float bacc = 0.f;                                           // this thread's part of (W1 beta0)[wf]
  auto fetch_chunk = [&](int c, f32x4 (&w)[CV], bool with_bias) {
    int wfl = wf, wpl = wp;
    asm volatile("" : "+v"(wfl), "+v"(wpl));                  // per-call addresses (hoisted out of the tile loop they cost 60 VGPRs)
    const float *wrow = p.params + p.w1 + (size_t)wfl * D, *g = p.params + p.fn_w, *bt = p.params + p.fn_b;
    const int k0 = 64 * c + CW * wpl;
    f32x4 gm[CV], be[CV];
    if (c < c_last) {
#pragma unroll
      for (int j4 = 0; j4 < CV; ++j4) w[j4] = ld4u(wrow + k0 + 4 * j4);
      if (fnorm) {
#pragma unroll
        for (int j4 = 0; j4 < CV; ++j4) gm[j4] = ld4u(g + k0 + 4 * j4);
        if (with_bias) {
#pragma unroll
          for (int j4 = 0; j4 < CV; ++j4) be[j4] = ld4u(bt + k0 + 4 * j4);
        }
      }
    } else {
#pragma unroll
      for (int j4 = 0; j4 < CV; ++j4) w[j4] = ld4_row(wrow, k0 + 4 * j4, D, al4);
      if (fnorm) {
#pragma unroll
        for (int j4 = 0; j4 < CV; ++j4) gm[j4] = ld4_row(g, k0 + 4 * j4, D, al4);
        if (with_bias) {
#pragma unroll
          for (int j4 = 0; j4 < CV; ++j4) be[j4] = ld4_row(bt, k0 + 4 * j4, D, al4);
        }
      }
    }
    if (fnorm) {
      if (with_bias) {
#pragma unroll
        for (int j4 = 0; j4 < CV; ++j4) { const f32x4 t = w[j4] * be[j4]; bacc += (t[0] + t[1]) + (t[2] + t[3]); }
      }
#pragma unroll
      for (int j4 = 0; j4 < CV; ++j4) w[j4] *= gm[j4];
    }
  };
  auto store_chunk = [&](int buf, const f32x4 (&w)[CV]) {
    float *dst = &sW[buf][wf * RS16 + CW * wp];
#pragma unroll
    for (int j4 = 0; j4 < CV; ++j4) st4(dst + 4 * j4, w[j4]);
  };
  auto row_ptr = [&](int64_t grp) {
    const int64_t i = (grp * NW + wave) * 16 + n;
    const int64_t row = i < p.B ? (p.rows ? (int64_t)p.rows[i] : i) : 0;
    return p.x + (p.x_M ? (row / p.x_M) * p.x_sn + (row % p.x_M) * p.x_sm : row * D);
  };
  f32x4 xq[NCH][4];                                           // the row block: xq[c][j4][t] = column 64 c + 16 j4 + 4 q + t (a row's 64 B per load)
  auto load_rows = [&](const float *xr, int c) {              // raw (ld4_row_fix is applied to the last chunk when the tile starts)
    int ql = q;
    asm volatile("" : "+v"(ql));                              // (offsets recomputed per call, not kept as address pairs)
    if (c < c_last) {
      const float *xc = xr + 4 * ql;
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) xq[c][j4] = ld4u(xc + 64 * c + 16 * j4);
    } else if (c == c_last) {
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) xq[c][j4] = ld4_row_raw(xr, 64 * c + 16 * j4 + 4 * ql, D);
    }
  };

  int64_t grp = bid;
  const bool any = grp < n_groups;                            // (uniform; the launch never has more workgroups than groups)
  if (any) {
    const float *xr = row_ptr(grp);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) xq[c][j4] = f32x4{0.f, 0.f, 0.f, 0.f};     // chunks beyond the row stay zero
      load_rows(xr, c);
    }
  }
  f32x4 wA[CV], wB[CV];                                       // chunks c (even) / c (odd) on their way to LDS
  bool first = true;
  if (any) { fetch_chunk(0, wA, true); fetch_chunk(1, wB, true); }
  pre();
  if (!any) return;
  for (;;) {
    const int64_t i = (grp * NW + wave) * 16 + n;
    const bool ok = i < p.B;
    const int64_t next = grp + nb;
    const bool has_next = next < n_groups;
    const float *xr_next = row_ptr(has_next ? next : grp);
    int ql = q;
    asm volatile("" : "+v"(ql));
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (c == c_last) {
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) xq[c][j4] = ld4_row_fix(xq[c][j4], 64 * c + 16 * j4 + 4 * ql, D, al4);
      }
    // ---- LayerNorm statistics over the D inputs (exact two-pass on the registers); the inputs become x - mean ----
    float mean = 0.f, rstd = 1.f;
    if (fnorm) {
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) s4 += xq[c][j4];
      mean = quad_sum16((s4[0] + s4[1]) + (s4[2] + s4[3])) * inv_D;
      const f32x4 mean4 = {mean, mean, mean, mean};
      f32x4 v4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (c < c_last) {
#pragma unroll
          for (int j4 = 0; j4 < 4; ++j4) { const f32x4 d = xq[c][j4] - mean4; xq[c][j4] = d; v4 += d * d; }
        } else if (c == c_last) {
#pragma unroll
          for (int j4 = 0; j4 < 4; ++j4) {
            f32x4 d = xq[c][j4] - mean4;
#pragma unroll
            for (int t = 0; t < 4; ++t) d[t] = (64 * c + 16 * j4 + 4 * ql + t < D) ? d[t] : 0.f;
            xq[c][j4] = d; v4 += d * d;
          }
        }
      }
      rstd = 1.0f / sqrtf(quad_sum16((v4[0] + v4[1]) + (v4[2] + v4[3])) * inv_D + LN_EPS);
    }
    // ---- z1 = b1' + rstd W1' (x - mean), chunk by chunk ----
    store_chunk(0, wA);                                        // (buffer 0 is free: the barrier after the previous tile's last chunk)
    f32x4 acc[4];
#pragma unroll
    for (int bo = 0; bo < 4; ++bo) acc[bo] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c <= c_last) {
        if (c + 2 <= c_last) { if (c & 1) fetch_chunk(c + 2, wB, first); else fetch_chunk(c + 2, wA, first); }
        const float *Wc = &sW[c & 1][n * RS16 + 4 * q];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          f32x4 a[4];
#pragma unroll
          for (int bo = 0; bo < 4; ++bo) a[bo] = ld4(Wc + 16 * bo * RS16 + 16 * jj);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int bo = 0; bo < 4; ++bo) acc[bo] = WIDE_MFMA(a[bo][t], xq[c][jj][t], acc[bo]);
        }
        if (has_next) load_rows(xr_next, c);                   // these 16 registers are free: the next tile's columns
        if (c + 1 <= c_last) {
          if (c & 1) store_chunk(0, wA); else store_chunk(1, wB);          // chunk c + 1 (set (c + 1) & 1) into buffer (c + 1) & 1
          if (first && c + 1 == c_last) {                      // every chunk has been fetched: the folded bias of row wf
            float t = bacc;
#pragma unroll
            for (int off = 1; off < TPR; off <<= 1) t += __shfl_xor(t, off, WAVE);
            if (wp == 0) sB[wf] = p.params[p.b1 + wf] + (p.fn_b >= 0 ? t : 0.f);
          }
        }
        __syncthreads();
      }
    }
    const f32x4 rstd4 = {rstd, rstd, rstd, rstd};
#pragma unroll
    for (int bo = 0; bo < 4; ++bo) acc[bo] = acc[bo] * rstd4 + ld4(sB + 16 * bo + 4 * q);
    if (has_next) { fetch_chunk(0, wA, false); fetch_chunk(1, wB, false); }       // re-prime the stream under the tail
    first = false;
    tail(acc, i, ok, mean, rstd);
    if (!has_next) break;
    grp = next;
  }
}

// wide_l1_fwd16_kernel — training-sized batches: W1' (gamma folded in) is staged ONCE per workgroup into LDS, whole
// (64 x 64 nch floats in fragment order: 128 KB at in_dim 512), so the tile loop has no barrier and no weight traffic at all: the 8 waves walk
// their tiles independently (one wave's row loads and statistics run under the other waves' MFMAs).  Streaming the chunks
// through a double buffer (wide16_layer1, kept for the rollout kernel where a workgroup sees too few tiles to amortise the
// staging) left every chunk waiting for an L2 round trip behind a workgroup barrier: 4 us per chunk against 2 us of MFMA.
// NCH is EXACT (in_dim in (64 (NCH - 1), 64 NCH]): the chunk that holds the row end is static, so the tile loop is ONE basic block —
// no per-chunk branches on the row length and no branch around the refill (the last tile reloads itself): hipcc then issues a
// chunk's LDS reads under the MFMAs of the chunk before (with the branches every 32 MFMAs started behind an exposed LDS round
// trip: 1280 -> 990 us at 1.6 M x 512).  The row end inside the last chunk is loads clamped into the row plus a fix-up / mask at
// the top of the tile, skipped (one uniform branch, outside the MFMA loop) when in_dim == 64 NCH.
// tail(acc, i, ok, mean, rstd): acc = z1 of sample i, bias included (accumulator layout).  pre(): the caller's own staging, before
// the barrier that publishes W1'.  LDS: [W1' 64 x 64 NCH | b1' 64 | caller's].
template <int NCH, class Pre, class Tail>
__device__ __forceinline__ void wide_l1_resident_body(const Wide16Args &p, float *lds, Pre &&pre, Tail &&tail) {
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = p.D;
  const bool fnorm = p.fn_w >= 0;
  const bool al4 = (D & 3) == 0;
  const bool full = D == 64 * NCH;
  const float inv_D = 1.0f / (float)D;
  constexpr int c_last = NCH - 1;
  // W1' in LDS in FRAGMENT order: the 16 x 16 block (bo, kb) — rows 16 bo + n, columns 16 kb + 4 q + t — is 256 consecutive
  // floats with lane (n, q)'s four values at float offset 4 * lane, so an A operand read is base + 16 * lane bytes and every
  // 16-lane group of the ds_read_b128 covers the 64 banks once.  (The row-major copy with stride 64 nch + 4 of round 2 put lanes
  // (11, q) and (12, q - 1) of a group on one bank quad: SQ_LDS_BANK_CONFLICT was 49 % of SQ_LDS_IDX_ACTIVE — one conflict cycle
  // per MFMA, profiles/r02/d_wide_l1_kernels_sq_pmc.txt.)
  constexpr int KB = 4 * NCH;                                   // 16-column blocks per row
  float *sW = lds, *sB = lds + HID * 16 * KB;
  // ---- stage W1' = W1 gamma0 (zeros beyond the row) and the folded bias ----
  {
    constexpr int nv = 4 * KB;                                  // 16-byte groups per row
    for (int e = tid; e < HID * nv; e += blockDim.x) {
      const int f = e / nv, g4 = e - f * nv, k = 4 * g4;
      f32x4 w = ld4_row(p.params + p.w1 + (size_t)f * D, k, D, al4);
      if (fnorm) w *= ld4_row(p.params + p.fn_w, k, D, al4);
      st4(sW + (((f >> 4) * KB + (g4 >> 2)) * 64 + (g4 & 3) * 16 + (f & 15)) * 4, w);
    }
    wide16_fold_bias<8>(p, sB);
    pre();
  }
  __syncthreads();

  const int64_t n_tiles = (p.B + 15) / 16, stride = (int64_t)gridDim.x * 8;
  auto row_ptr = [&](int64_t tile) {
    const int64_t i = tile * 16 + n;
#ifdef WIDE_EXP_ROW0
    const int64_t row = n + 0 * i;
#else
    const int64_t row = i < p.B ? (p.rows ? (int64_t)p.rows[i] : i) : 0;
#endif
    return p.x + row * D;
  };
  f32x4 xq[NCH][4];                                           // the row block: xq[c][j4][t] = column 64 c + 16 j4 + 4 q + t (a row's 64 B per load)
  auto load_rows = [&](const float *xr, int c) {              // raw (ld4_row_fix is applied to the last chunk when the tile starts)
    int ql = q;
    asm volatile("" : "+v"(ql));                              // (offsets recomputed per call, not kept as address pairs)
    if (c < c_last) {
      const float *xc = xr + 4 * ql;
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) xq[c][j4] = ld4u(xc + 64 * c + 16 * j4);
    } else {
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) xq[c][j4] = ld4_row_raw(xr, 64 * c + 16 * j4 + 4 * ql, D);
    }
  };
  int64_t tile = (int64_t)blockIdx.x * 8 + wave;
  if (tile >= n_tiles) return;
  {
    const float *xr = row_ptr(tile);
#pragma unroll
    for (int c = 0; c < NCH; ++c) load_rows(xr, c);
  }
  for (;;) {
    const int64_t i = tile * 16 + n;
    const bool ok = i < p.B;
    const int64_t next = tile + stride;
    const bool has_next = next < n_tiles;
    const float *xr_next = row_ptr(has_next ? next : tile);
    int ql = q;
    asm volatile("" : "+v"(ql));
    if (!full) {
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) xq[c_last][j4] = ld4_row_fix(xq[c_last][j4], 64 * c_last + 16 * j4 + 4 * ql, D, al4);
    }
    // ---- LayerNorm statistics over the D inputs (exact two-pass on the registers); the inputs become x - mean ----
    float mean = 0.f, rstd = 1.f;
    if (fnorm) {
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) s4 += xq[c][j4];
      mean = quad_sum16((s4[0] + s4[1]) + (s4[2] + s4[3])) * inv_D;
      const f32x4 mean4 = {mean, mean, mean, mean};
      f32x4 v4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < c_last; ++c)
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) { const f32x4 d = xq[c][j4] - mean4; xq[c][j4] = d; v4 += d * d; }
      if (full) {
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) { const f32x4 d = xq[c_last][j4] - mean4; xq[c_last][j4] = d; v4 += d * d; }
      } else {                                                  // columns beyond the row stay out of the sum and meet the MFMAs as zeros
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
          f32x4 d = xq[c_last][j4] - mean4;
#pragma unroll
          for (int t = 0; t < 4; ++t) d[t] = (64 * c_last + 16 * j4 + 4 * ql + t < D) ? d[t] : 0.f;
          xq[c_last][j4] = d; v4 += d * d;
        }
      }
      rstd = 1.0f / sqrtf(quad_sum16((v4[0] + v4[1]) + (v4[2] + v4[3])) * inv_D + LN_EPS);
    }
    // ---- z1 = b1' + rstd W1' (x - mean) ----
    f32x4 acc[4];
#pragma unroll
    for (int bo = 0; bo < 4; ++bo) acc[bo] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float *Wc = sW + (4 * c * 64 + lane) * 4;          // block (bo, kb = 4 c + jj) at (bo KB + kb) * 256
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        f32x4 a[4];
#pragma unroll
        for (int bo = 0; bo < 4; ++bo) a[bo] = ld4(Wc + (bo * KB + jj) * 256);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int bo = 0; bo < 4; ++bo) acc[bo] = WIDE_MFMA(a[bo][t], xq[c][jj][t], acc[bo]);
      }
      load_rows(xr_next, c);                                   // these 16 registers are free: the next tile's columns (the last tile reloads itself)
    }
    const f32x4 rstd4 = {rstd, rstd, rstd, rstd};
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = acc[b] * rstd4 + ld4(sB + 16 * b + 4 * q);
    tail(acc, i, ok, mean, rstd);
    if (!has_next) break;
    tile = next;
  }
}

template <int NCH>
__global__ __launch_bounds__(512, 2) void wide_l1_fwd16_kernel(Wide16Args p) {
  extern __shared__ __align__(16) float lds[];
  const int q = (threadIdx.x & 63) >> 4;
  wide_l1_resident_body<NCH>(p, lds, []() {},
    [&](f32x4 (&acc)[4], int64_t i, bool ok, float mean, float rstd) __attribute__((always_inline)) {
      if (ok) {
        if (q == 0 && p.mean0) { p.mean0[i] = mean; p.rstd0[i] = rstd; }
#pragma unroll
        for (int b = 0; b < 4; ++b) st4(p.z1 + i * HID + 16 * b + 4 * q, acc[b]);
      }
    });
}

// Everything the register-resident tail (mlp_fwd16.h) reads from LDS — the per-feature vectors from b1 on, W2.., the head —
// staged with ONE memory latency: every global load is issued before the first LDS store (stage_all_weights walks the
// matrices one after the other: four to five dependent L2 round trips, most of a step-sized launch).  W1 and the feature-norm
// vectors are not staged: the wide kernels read them from global memory.  256 or 512 threads.
template <int LN>
__device__ __forceinline__ void stage_tail_1shot(float *lds, const LdsMap &m, const float *__restrict__ params, const NetOff &o,
                                                 const mappo_net_desc &d) {
  const int A = d.out_dim, nthr = blockDim.x, tid = threadIdx.x;
  constexpr int NVEC = 3 * HID * (1 + LN) + 32, JV = (NVEC + 255) / 256;
  float vv[JV]; int vd[JV];
#pragma unroll
  for (int j = 0; j < JV; ++j) {
    const int e = j * nthr + tid;
    int src = -1, dst = -1;
    if (e < 3 * HID) { dst = m.b1 + e; src = o.b1 + e; }
    else if (e < 3 * HID * (1 + LN)) {
      const int i = e - 3 * HID, l = i / (3 * HID), r = i - l * 3 * HID;
      dst = (l == 0 ? m.b2[0] : m.b2[LN > 1 ? 1 : 0]) + r;
      src = (l == 0 ? o.b2[0] : o.b2[LN > 1 ? 1 : 0]) + r;
    } else if (e < NVEC) { const int i = e - 3 * HID * (1 + LN); dst = m.bh + i; if (i < A) src = o.bh + i; }
    const float ld = params[src >= 0 ? src : 0];
    vv[j] = src >= 0 ? ld : 0.f; vd[j] = dst;
  }
  f32x4 w2v[LN > 0 ? LN : 1][4], whv[2];
#pragma unroll
  for (int l = 0; l < LN; ++l)
#pragma unroll
    for (int j = 0; j < 4; ++j) w2v[l][j] = ld4u(params + o.w2[l] + 4 * min(j * nthr + tid, 1023));
  const int n4_h = 16 * A;
#pragma unroll
  for (int j = 0; j < 2; ++j) whv[j] = ld4u(params + o.wh + 4 * min(j * nthr + tid, n4_h - 1));
  // ---- stores ----
#pragma unroll
  for (int j = 0; j < JV; ++j) if (vd[j] >= 0) lds[vd[j]] = vv[j];
#pragma unroll
  for (int l = 0; l < LN; ++l)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = j * nthr + tid;
      if (i < 1024) {
        const int f = i >> 4, k = (i & 15) << 2;                 // element 4 i = W2[f][k .. k + 3] -> dst[k * WP + f]
#pragma unroll
        for (int c = 0; c < 4; ++c) lds[m.w2[l] + (k + c) * WP + f] = w2v[l][j][c];
      }
    }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = j * nthr + tid;
    if (i < n4_h) {
      const int a = i >> 4, k = (i & 15) << 2;                   // Wh[a][k .. k + 3] -> dst[k * HP + a]
#pragma unroll
      for (int c = 0; c < 4; ++c) lds[m.wh + (k + c) * HP + a] = whv[j][c];
    }
  }
  for (int e = tid; e < HID * (32 - A); e += nthr) {             // columns a >= A of the head are zero
    const int k = e / (32 - A), a = A + e - k * (32 - A);
    lds[m.wh + k * HP + a] = 0.f;
  }
}

// The whole forward of a wide-input network in one launch (rollout: get_actions / get_values / trunk features): layer 1 as above,
// then the register-resident 16x16x4 tail of the narrow kernels (mlp_fwd16.h) on the same tile.
template <bool RELU, int LN, int MODE, int NW, int NCH>
__device__ __forceinline__ void wide_forward16_body(const Wide16Args &w, const FwdArgs &p, float *lds, float (*sW)[HID * RS16], float *sB,
                                                    const int bid, const int nb) {
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float *tZ = lds + p.map.tiles + wave * p.map.wave_stride;
  wide16_layer1<NCH, NW>(w, sW, sB, bid, nb,
    [&]() __attribute__((always_inline)) { stage_tail_1shot<LN>(lds, p.map, p.params, p.off, p.desc); },    // everything but W1 (streamed in chunks)
    [&](f32x4 (&acc)[4], int64_t i, bool ok, float, float) __attribute__((always_inline)) {
      forward16_tail<RELU, LN, MODE>(p, lds, p.map, acc, i, ok, j, q, tZ);
    });
}

template <bool RELU, int LN, int MODE, int NW, int NCH>
__global__ __launch_bounds__(64 * NW, 2) void wide_forward16_kernel(Wide16Args w, FwdArgs p) {
  extern __shared__ __align__(16) float lds[];
  __shared__ __align__(16) float sW[2][HID * RS16];
  __shared__ __align__(16) float sB[HID];
  wide_forward16_body<RELU, LN, MODE, NW, NCH>(w, p, lds, sW, sB, blockIdx.x, gridDim.x);
}

// get_actions of the actor AND get_values of the critic in one launch (mpe_runner.py:95-109 for MLP policies with wide inputs):
// workgroups [0, nA) the actor (sampling), [nA, nA + nC) the critic.  As two launches — even on two streams inside the captured
// episode — each network's 8-tile groups occupied half of the chip's CUs for the length of its chunk-latency chain, one after
// the other (45 + 49 us per step at BASELINE configs[4]).
struct WideStepArgs { Wide16Args wa, wc; FwdArgs a, c; int nA; };
template <bool RELU, int LN, int NCH>
__global__ __launch_bounds__(512, 2) void wide_rollout_step_kernel(WideStepArgs s) {
  extern __shared__ __align__(16) float lds[];
  __shared__ __align__(16) float sW[2][HID * RS16];
  __shared__ __align__(16) float sB[HID];
  if ((int)blockIdx.x < s.nA) wide_forward16_body<RELU, LN, 1, 8, NCH>(s.wa, s.a, lds, sW, sB, blockIdx.x, s.nA);
  else wide_forward16_body<RELU, LN, 0, 8, NCH>(s.wc, s.c, lds, sW, sB, (int)blockIdx.x - s.nA, (int)gridDim.x - s.nA);
}

// The same step when in_dim is 256 or 512 and a wave sees at most two tiles (BASELINE configs[4]: 1 024 tiles per network on 128
// workgroups each): W1' is staged WHOLE, in fragment order, behind ONE memory latency (the streamed form above pays a chunk
// fetch + barrier eight times: 48 us per step), the tile's rows — requested before the staging — feed the branch-free MFMA loop of
// wide_l1_fwd16_kernel<NCH, true>, and z1 waits in registers while the workgroup swaps W1' for the tail's weights (the two do not
// fit the LDS together).  The rows a network loads are also stored to its buffer slot (copy_dst): the rollout insert's obs /
// share_obs copies — 34 us of their own at configs[4], an element-wise kernel moving each row once in and twice out — ride on
// loads the forward does anyway; rewards and masks are a few elements per workgroup at the end.
struct WideFullArgs { Wide16Args wa, wc; FwdArgs a, c; int nA; InsertArgs ins; int has_ins; };

template <bool RELU, int LN, int MODE, int NCH>
__device__ __forceinline__ void wide_full_body(const Wide16Args &w, const FwdArgs &p, float *lds, const int bid, const int nb) {
  constexpr int D = 64 * NCH, KB = 4 * NCH;
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool fnorm = w.fn_w >= 0;
  const float inv_D = 1.0f / (float)D;
  float *sW = lds, *sB = lds + HID * D;
  const int64_t n_tiles = (w.B + 15) / 16, stride = (int64_t)nb * 8;
  const int64_t tile0 = (int64_t)bid * 8 + wave;
  const bool has[2] = {tile0 < n_tiles, tile0 + stride < n_tiles};
  auto src_of = [&](int64_t tile) {
    const int64_t i = tile * 16 + n, row = i < w.B ? i : 0;
    return w.x + (w.x_M ? (row / w.x_M) * w.x_sn + (row % w.x_M) * w.x_sm : row * D);
  };
  f32x4 xq[NCH][4];
  auto load_rows = [&](const float *xr, int c) {
    int ql = q;
    asm volatile("" : "+v"(ql));
    const float *xc = xr + 4 * ql + 64 * c;
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) xq[c][j4] = ld4u(xc + 16 * j4);
  };
  {
    const float *xr = src_of(has[0] ? tile0 : 0);             // requested before the staging: in flight under it
#pragma unroll
    for (int c = 0; c < NCH; ++c) load_rows(xr, c);
  }
  {
    constexpr int nv = 4 * KB;
    for (int e = tid; e < HID * nv; e += blockDim.x) {
      const int f = e / nv, g4 = e - f * nv, k = 4 * g4;
      f32x4 wv = ld4u(w.params + w.w1 + (size_t)f * D + k);
      if (fnorm) wv *= ld4u(w.params + w.fn_w + k);
      st4(sW + (((f >> 4) * KB + (g4 >> 2)) * 64 + (g4 & 3) * 16 + (f & 15)) * 4, wv);
    }
    wide16_fold_bias<8>(w, sB);
  }
  __syncthreads();
  f32x4 z[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int b = 0; b < 4; ++b) z[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (has[t]) {                                              // (wave-uniform; the block below is straight-line code)
      const int64_t tile = tile0 + t * stride, i = tile * 16 + n;
      const bool ok = i < w.B;
      if (w.copy_dst && ok) {
        float *dst = w.copy_dst + i * D + 4 * q;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int j4 = 0; j4 < 4; ++j4) st4(dst + 64 * c + 16 * j4, xq[c][j4]);
      }
      float rstd = 1.f;
      if (fnorm) {
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int j4 = 0; j4 < 4; ++j4) s4 += xq[c][j4];
        const float mean = quad_sum16((s4[0] + s4[1]) + (s4[2] + s4[3])) * inv_D;
        const f32x4 mean4 = {mean, mean, mean, mean};
        f32x4 v4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int j4 = 0; j4 < 4; ++j4) { const f32x4 d = xq[c][j4] - mean4; xq[c][j4] = d; v4 += d * d; }
        rstd = 1.0f / sqrtf(quad_sum16((v4[0] + v4[1]) + (v4[2] + v4[3])) * inv_D + LN_EPS);
      }
      const float *xr_next = src_of(has[1] ? tile0 + stride : tile0);
      f32x4 acc[4];
#pragma unroll
      for (int bo = 0; bo < 4; ++bo) acc[bo] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const float *Wc = sW + (4 * c * 64 + lane) * 4;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          f32x4 a[4];
#pragma unroll
          for (int bo = 0; bo < 4; ++bo) a[bo] = ld4(Wc + (bo * KB + jj) * 256);
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int bo = 0; bo < 4; ++bo) acc[bo] = WIDE_MFMA(a[bo][r], xq[c][jj][r], acc[bo]);
        }
        if (t == 0) load_rows(xr_next, c);                     // the second tile's columns (a repeat of the first if there is none)
      }
      const f32x4 rstd4 = {rstd, rstd, rstd, rstd};
#pragma unroll
      for (int b = 0; b < 4; ++b) z[t][b] = acc[b] * rstd4 + ld4(sB + 16 * b + 4 * q);
    }
  }
  __syncthreads();                                             // every wave is done with W1': the tail's weights take its place
  stage_tail_1shot<LN>(lds, p.map, p.params, p.off, p.desc);
  __syncthreads();
  float *tZ = lds + p.map.tiles + wave * p.map.wave_stride;
#pragma unroll
  for (int t = 0; t < 2; ++t)
    if (has[t]) {
      const int64_t i = (tile0 + t * stride) * 16 + n;
      forward16_tail<RELU, LN, MODE>(p, lds, p.map, z[t], i, i < w.B, n, q, tZ);
    }
}

template <bool RELU, int LN, int NCH>
__global__ __launch_bounds__(512, 2) void wide_rollout_full_kernel(WideFullArgs s) {
  extern __shared__ __align__(16) float lds[];
  if ((int)blockIdx.x < s.nA) wide_full_body<RELU, LN, 1, NCH>(s.wa, s.a, lds, blockIdx.x, s.nA);
  else wide_full_body<RELU, LN, 0, NCH>(s.wc, s.c, lds, (int)blockIdx.x - s.nA, (int)gridDim.x - s.nA);
  if (s.has_ins) {                                             // rewards / masks of the insert (the row copies rode on the loads above)
    const InsertArgs &p = s.ins;
    const int64_t R = (int64_t)p.N * p.M;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < R; e += (int64_t)gridDim.x * blockDim.x) {
      const int nn = (int)(e / p.M), m = (int)(e - (int64_t)nn * p.M);
      p.rew_dst[e] = p.rew[nn * p.rew_sn + m * p.rew_sm];
      p.mask_dst[e] = p.done[nn * p.done_sn + m * p.done_sm] ? 0.f : 1.f;
    }
  }
}

// Trunk features of a TRAINING-sized batch (recurrent networks: mappo_mlp_features / _seq, 128 000 rows per minibatch at BASELINE
// configs[3]): W1' resident as in wide_l1_fwd16_kernel — a workgroup walks enough tiles to amortise staging it whole — and the rest of
// the trunk from the accumulators of the same tile.  The streamed rollout form (wide_forward16_kernel) pays a chunk fetch + workgroup
// barrier per 64 columns and tile group: 0.16 of the fp32 MFMA peak at in_dim 322.  LDS: [W1' | b1' | the tail's map without its tiles].
template <bool RELU, int LN, int NCH>
__global__ __launch_bounds__(512, 2) void wide_features16_resident_kernel(Wide16Args w, FwdArgs p) {
  extern __shared__ __align__(16) float lds[];
  float *lt = lds + HID * 64 * NCH + HID;
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  wide_l1_resident_body<NCH>(w, lds,
    [&]() __attribute__((always_inline)) { stage_tail_1shot<LN>(lt, p.map, p.params, p.off, p.desc); },
    [&](f32x4 (&acc)[4], int64_t i, bool ok, float, float) __attribute__((always_inline)) {
      forward16_tail<RELU, LN, 2>(p, lt, p.map, acc, i, ok, j, q, nullptr);
    });
}

// Trunk features of a recurrent actor AND critic with wide inputs in one launch (rollout step: the two networks read different
// rows of the same step; at 40 tiles each neither fills the chip): workgroups [0, nA) actor, [nA, gridDim.x) critic.
struct WideDualArgs { Wide16Args wa, wc; FwdArgs a, c; int nA; };
template <bool RELU, int LN, int NW>
__global__ __launch_bounds__(64 * NW, 2) void wide_features16_dual_kernel(WideDualArgs d) {
  extern __shared__ __align__(16) float lds[];
  __shared__ __align__(16) float sW[2][HID * RS16];
  __shared__ __align__(16) float sB[HID];
  if ((int)blockIdx.x < d.nA) wide_forward16_body<RELU, LN, 2, NW, 8>(d.wa, d.a, lds, sW, sB, blockIdx.x, d.nA);
  else wide_forward16_body<RELU, LN, 2, NW, 8>(d.wc, d.c, lds, sW, sB, (int)blockIdx.x - d.nA, (int)gridDim.x - d.nA);
}

// ---- split-K forward for step-sized batches (a rollout step of configs[3] is 40 tiles per network) --------------------------
// With that few tiles the streamed kernel above is one latency chain per workgroup: 3..8 chunks x (L2 fetch, LDS, barrier, 64
// MFMAs on ONE wave per SIMD).  Here a workgroup of 4 waves owns ONE tile and splits the input columns: wave w takes the
// 64-column chunks w, w + 4 of the tile's rows AND of W1 — its A operands come straight from global memory (16-byte loads, no
// LDS staging, no chunk barriers: each weight is used by exactly one wave), everything is requested up front (ONE memory
// latency), then <= 128 MFMAs per wave, and the four partial accumulators meet in LDS.  The B operand is the full LayerNorm
// output  gamma0 xhat0 + beta0  (the affine applied to the registers), so the weights are used raw and no folded bias is needed.
// The LayerNorm statistics over the whole row are two small exchanges (exact two-pass form).  Wave 0 then runs the
// register-resident tail.  A workgroup walking several tiles keeps its A operands.
struct SkShared {
  float sS[4][16], sV[4][16];                                 // per-wave partial sums / centred squares of the 16 samples
  float4 sAcc[3][4][64];                                      // partial accumulators of waves 1..3
};

struct SkNoHook { __device__ __forceinline__ void operator()() const {} };
// after_loads(): called once, when the first tile's weight and row loads are out — loads the CALLER needs later (the GRU step's
// weights in the fused recurrent step) go behind them: the counter is in order, so what is requested first is what the first MFMA
// waits for
template <bool RELU, int LN, int MODE, class Hook = SkNoHook>
__device__ __forceinline__ void wide_forward16_sk_body(const Wide16Args &w, const FwdArgs &p, float *lds, SkShared &sh, const int bid, const int nb,
                                                       float *xshare = nullptr, Hook &&after_loads = Hook()) {      // MODE 4: the trunk output goes to xshare[4][64] float4
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = w.D;
  const bool fnorm = w.fn_w >= 0;
  const bool al4 = (D & 3) == 0;
  const float inv_D = 1.0f / (float)D;
  const int c_last = (D + 63) / 64 - 1;
  const int64_t n_tiles = (w.B + 15) / 16;
  if (bid >= n_tiles) return;                                 // (uniform; never more workgroups than tiles)
  // ---- the first tile's rows, then this wave's columns (chunks wave, wave + 4): their W1 rows, gamma0, beta0 ----
  // Everything is requested RAW — loads clamped into the row / to the last chunk, no instruction on a value in flight — and in the
  // order it is needed: the counter is in order, and a fix-up or a select right behind the loads (as this prologue had them: a
  // branch per chunk with zeros on its other side) puts the wait for ALL of them in front of the row loads.  The row-end shift /
  // zero-fill of the weights (ld4_row_fix) happens once, after the statistics exchanges of the first tile.
  f32x4 xraw[2][4];
  auto load_rows = [&](int64_t tl) {
    const int64_t i = tl * 16 + n;
    const int64_t row = i < w.B ? (w.rows ? (int64_t)w.rows[i] : i) : 0;
    const float *xr = w.x + row * D;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = min(wave + 4 * j, c_last);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) xraw[j][jj] = ld4_row_raw(xr, 64 * c + 16 * jj + 4 * q, D);
    }
  };
  load_rows(bid);
  f32x4 A[2][4][4];                                           // [chunk j][bo][jj]: W1[16 bo + n][64 c + 16 jj + 4 q .. + 3]
  f32x4 gam[2][4], bet[2][4];
  const float *gsrc = fnorm ? w.params + w.fn_w : w.params + w.w1, *bsrc = fnorm ? w.params + w.fn_b : w.params + w.w1;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = min(wave + 4 * j, c_last);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int k = 64 * c + 16 * jj + 4 * q;
#pragma unroll
      for (int bo = 0; bo < 4; ++bo) A[j][bo][jj] = ld4_row_raw(w.params + w.w1 + (size_t)(16 * bo + n) * D, k, D);
      gam[j][jj] = ld4_row_raw(gsrc, k, D);
      bet[j][jj] = ld4_row_raw(bsrc, k, D);
    }
  }
  const int jq = lane & 15;
  float *tZ = lds + p.map.tiles;                               // wave 0's logits tile
  for (int64_t tile = bid; tile < n_tiles; tile += nb) {
    const int64_t i = tile * 16 + n;
    const bool ok = i < w.B;
    f32x4 xq[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = wave + 4 * j;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        xq[j][jj] = (c <= c_last) ? ld4_row_fix(xraw[j][jj], 64 * c + 16 * jj + 4 * q, D, al4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (tile + nb < n_tiles) load_rows(tile + nb);             // (a workgroup with a second tile: its rows under this one)
    if (tile == bid) {     // first tile: the tail's weights (everything but W1) go to LDS behind the loads above — one memory latency in all
      after_loads();
      stage_tail_1shot<LN>(lds, p.map, p.params, p.off, p.desc);
    }
    if (fnorm) {
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) s4 += xq[j][jj];
      const float ps = quad_sum16((s4[0] + s4[1]) + (s4[2] + s4[3]));
      if (q == 0) sh.sS[wave][n] = ps;
      __syncthreads();
      const float mean = ((sh.sS[0][n] + sh.sS[1][n]) + (sh.sS[2][n] + sh.sS[3][n])) * inv_D;
      const f32x4 mean4 = {mean, mean, mean, mean};
      f32x4 v4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = wave + 4 * j;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          f32x4 d = xq[j][jj] - mean4;
          if (c >= c_last) {
#pragma unroll
            for (int t = 0; t < 4; ++t) d[t] = (64 * c + 16 * jj + 4 * q + t < D) ? d[t] : 0.f;
          }
          xq[j][jj] = d; v4 += d * d;
        }
      }
      const float pv = quad_sum16((v4[0] + v4[1]) + (v4[2] + v4[3]));
      if (q == 0) sh.sV[wave][n] = pv;
      __syncthreads();
      const float rstd = 1.0f / sqrtf(((sh.sV[0][n] + sh.sV[1][n]) + (sh.sV[2][n] + sh.sV[3][n])) * inv_D + LN_EPS);
      const f32x4 rstd4 = {rstd, rstd, rstd, rstd};
      if (tile == bid) {                                        // (once: the raw feature-norm vectors -> shifted / zero-filled at the row end)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = wave + 4 * j;
          if (c >= c_last) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const int k = 64 * c + 16 * jj + 4 * q;
              gam[j][jj] = c == c_last ? ld4_row_fix(gam[j][jj], k, D, al4) : f32x4{0.f, 0.f, 0.f, 0.f};
              bet[j][jj] = c == c_last ? ld4_row_fix(bet[j][jj], k, D, al4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) xq[j][jj] = xq[j][jj] * rstd4 * gam[j][jj] + bet[j][jj];
    }
    if (tile == bid) {                                          // (once: the raw W1 columns of the row-end chunk; chunks beyond it are skipped below)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (wave + 4 * j == c_last) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int bo = 0; bo < 4; ++bo) A[j][bo][jj] = ld4_row_fix(A[j][bo][jj], 64 * c_last + 16 * jj + 4 * q, D, al4);
        }
      }
    }
    f32x4 acc[4];
#pragma unroll
    for (int bo = 0; bo < 4; ++bo) acc[bo] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (wave + 4 * j <= c_last) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int bo = 0; bo < 4; ++bo) acc[bo] = mfma16(A[j][bo][jj][t], xq[j][jj][t], acc[bo]);
      }
    }
    if (wave > 0) {
#pragma unroll
      for (int bo = 0; bo < 4; ++bo) sh.sAcc[wave - 1][bo][lane] = make_float4(acc[bo][0], acc[bo][1], acc[bo][2], acc[bo][3]);
    }
    __syncthreads();                                           // partial accumulators (and, first tile, the staged weights) are in LDS
    if (wave == 0) {
#pragma unroll
      for (int bo = 0; bo < 4; ++bo) {
        const f32x4 b1 = ld4u(w.params + w.b1 + 16 * bo + 4 * q);
#pragma unroll
        for (int sw = 0; sw < 3; ++sw) {
          const float4 t = sh.sAcc[sw][bo][lane];
          acc[bo][0] += t.x; acc[bo][1] += t.y; acc[bo][2] += t.z; acc[bo][3] += t.w;
        }
        acc[bo] += b1;
      }
      forward16_tail<RELU, LN, MODE>(p, lds, p.map, acc, i, ok, jq, q, MODE == 4 ? xshare : tZ);
    }
    __syncthreads();                                           // sAcc / sS / sV free for the next tile
  }
}

template <bool RELU, int LN, int MODE>
__global__ __launch_bounds__(256, 1) void wide_forward16_sk_kernel(Wide16Args w, FwdArgs p) {
  extern __shared__ __align__(16) float lds[];
  __shared__ SkShared sh;
  wide_forward16_sk_body<RELU, LN, MODE>(w, p, lds, sh, blockIdx.x, gridDim.x);
}

template <bool RELU, int LN>
__global__ __launch_bounds__(256, 1) void wide_features16_sk_dual_kernel(WideDualArgs d) {
  extern __shared__ __align__(16) float lds[];
  __shared__ SkShared sh;
  if ((int)blockIdx.x < d.nA) wide_forward16_sk_body<RELU, LN, 2>(d.wa, d.a, lds, sh, blockIdx.x, d.nA);
  else wide_forward16_sk_body<RELU, LN, 2>(d.wc, d.c, lds, sh, (int)blockIdx.x - d.nA, (int)gridDim.x - d.nA);
}

#ifdef MLP_TU_WIDE_SK
// ---- one rollout step of a recurrent actor AND critic with wide inputs in ONE launch (r_actor_critic.py:43-70,146-165;
// smac_runner.py:110-127) ----
// Round 2 ran such a step as two launches (mappo_mlp_features_dual: split-K trunks -> featT in HBM; mappo_gru_step_dual: GRU cell +
// rnn.norm + heads), 21 + 14 us at BASELINE configs[3] where the arithmetic is a few microseconds: two launch latencies, two
// weight-fetch latencies and a feature round trip through HBM.  Here the 4-wave workgroup that runs a tile's split-K trunk goes
// straight on to the tile's GRU step: the GRU / head operands of every wave are requested BEFORE the trunk starts (they land
// under it), wave 0's trunk output crosses to the other waves through 4 KB of LDS, and gru_step3_tiles finishes the row.
#include "gru_step3.h"
struct WideRecDualArgs {
  WideDualArgs d;
  GruFwdArgs ga, gc;
  SmacInsert ins;             // nI > 0: workgroups [2 nA, 2 nA + nI) perform the SMAC insert of the env output the rows are read from
  int nI;
};
template <bool RELU, int LN>
__global__ __launch_bounds__(256, 1) void wide_recurrent_step_dual_kernel(WideRecDualArgs r) {
  extern __shared__ __align__(16) float lds[];
  __shared__ SkShared sh;
  __shared__ Step3Shared s3;
  __shared__ float4 sX[4 * 64];
  const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool actor = (int)blockIdx.x < r.d.nA;                    // one tile per workgroup: grid = 2 x tiles (+ insert workgroups)
  const int bid = actor ? (int)blockIdx.x : (int)blockIdx.x - r.d.nA, nb = r.d.nA;
  if ((int)blockIdx.x >= 2 * r.d.nA) { insert_smac_body(r.ins, (int)blockIdx.x - 2 * r.d.nA, r.nI); return; }
  Step3W<0> W;
  // (the GRU step's 96 weight registers are requested BEHIND the trunk's own weights and rows: asked for first, they were what the
  // trunk's first MFMA waited for)
  if (actor) {
    wide_forward16_sk_body<RELU, LN, 4>(r.d.wa, r.d.a, lds, sh, bid, nb, reinterpret_cast<float *>(sX),
                                        [&]() __attribute__((always_inline)) { gru_step3_load<3, 0>(W, r.ga, wv, n, q); });
    gru_step3_tiles<2, 3, 0>(W, r.ga, s3, bid, nb, sX);
  } else {
    wide_forward16_sk_body<RELU, LN, 4>(r.d.wc, r.d.c, lds, sh, bid, nb, reinterpret_cast<float *>(sX),
                                        [&]() __attribute__((always_inline)) { gru_step3_load<3, 0>(W, r.gc, wv, n, q); });
    gru_step3_tiles<1, 3, 0>(W, r.gc, s3, bid, nb, sX);
  }
}
#undef GS
#undef NG
#endif

// ------------------------------------------------------------------------------------------------------------------------
// wide_l1_bwd16_kernel — weight gradient of layer 1 and the feature-norm gradients for in_dim 65..512 from dz1 and the row
// statistics the forward left (workspace layout below):
//     G[f][k] = sum_s dz1[f][s] xhat0[k][s]   (RAW product: xhat0 without the affine),   db[f] = sum_s dz1[f][s]
//     dW1 = gamma0[k] G + beta0[k] db[f],   dgamma0[k] = sum_f W1[f][k] G[f][k],   dbeta0[k] = sum_f W1[f][k] db[f]
// (the raw-product identities of mlp_impl.h: no dX = W1^T dz1 pass — half the MFMA work of the round-1 kernel, which also
// re-read dz1 once per 64-column chunk).  A workgroup walks 16-sample tiles; wave w owns the 64-column chunk w % NCA of W1's
// gradient (64 accumulator registers) for the tiles of its tile group w / NCA:
//   * A operand = dz1^T: lane (m, q) needs dz1[16 bf + m][4 q .. 4 q + 3] — ONE 16-byte load; dz1 is stored BLOCKED,
//     [tile][64 features][16 samples], so that such a load instruction covers 1 KB of contiguous memory (feature-major [64][B]
//     made every lane touch its own cache line: 1.46 -> 1.18 ms at B = 1.6 M);
//   * B operand = xhat0^T of the chunk: lane (n, q) reads x[row 4 q + j][chunk + 4 n .. 4 n + 3] as ONE 16-byte load (the
//     accumulator block bk holds column chunk + 4 n + bk: a row's 64 columns are 256 contiguous bytes over the 16 lanes; every
//     input element is read exactly once by exactly one wave) and normalises it with the sample's (mean0, rstd0);
//   * 64 MFMAs per tile and wave; the next tile's operands are fetched under them.  Like the forward kernel this one is bound by
//     instruction issue (fp32 MFMA shares the vector ALU), so the tile loop is specialised: full tiles only (the one partial
//     tile of a batch is a separate masked step), row gather or not, row-end chunk or not — addresses advance by pointer
//     increments, the per-element work is two packed instructions per sample.
// The transform above runs per wave in registers at the end (it is linear, so it commutes with the slab reduction); one slab
// row per (workgroup, tile group).
//
// Workspace of the 16x16x4 wide path (floats; Bp = B rounded up to 16):
//     [0, 64 Bp)  dz1 blocked [tile][64][16]  |  [64 Bp, 65 Bp) mean0  |  [65 Bp, 66 Bp) rstd0  |  [66 Bp, 66 Bp + 64 B)  z1 [B][64]
// ------------------------------------------------------------------------------------------------------------------------
struct WideBwdOps {
  f32x4 a[4];                // dz1[16 bf + m][4 q + j]
  f32x4 b[4];                // x[row 4 q + j][chunk + 4 n + bk]  (b[j][bk]; raw load, see ld4_row_raw)
  f32x4 mean, rstd;          // of samples 4 q + j
};

// One kernel instance has ONE copy of the tile loop (several specialised copies under uniform branches made the compiler keep
// the 64 accumulators in different registers per copy and shuffle / spill them at the joins): feature norm and row gather are
// template parameters chosen by the host; the row-end chunk and the one partial tile of a batch are small uniform branches
// around the per-element fix-ups only.
// FULL (in_dim a multiple of 64: no row-end chunk): the loop over a wave's FULL tiles is ONE basic block — three operand sets in
// rotation, every fetch unconditional (tile indices clamped to the batch, a repeat's products scaled by 0), the chunk-0 wave's bias
// sums by a 0/1 factor instead of a branch — and the batch's one partial tile runs after it.  With the conditional fetches of the
// general loop hipcc's wait-count insertion loses track of the rotation at the branch joins and puts s_waitcnt vmcnt(0) in front
// of a set's products: each compute then also waits for the prefetch issued just before it (58 % of the wave cycles waiting,
// profiles/r03/d_wide_l1_kernels_sq_pmc.txt); in the single block it counts the outstanding loads exactly (vmcnt(20)).
template <bool FNORM, bool GATHER, bool FULL>
__global__ __launch_bounds__(512, 2) void wide_l1_bwd16_kernel(WideBwd16Args p) {
  __shared__ float sDb[8][HID];
  const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int chunk = wave % p.nca, tg = wave / p.nca;
  const int c0 = 64 * chunk;
  const bool active = tg < p.groups && c0 < p.D;
  const bool fnorm = FNORM;
  const bool edge = c0 + 64 > p.D;                             // this wave's chunk holds the row end: clamped loads, masked columns
  const bool al4 = (p.D & 3) == 0;
  const int kcol = c0 + 4 * n, kcl = min(kcol, p.D - 4);       // this lane's 4 columns; where its 16-byte load starts
  const int64_t Bp = wide16_bp(p.B);
  const float *dz = p.wide_ws, *st_mean = p.wide_ws + 64 * Bp, *st_rstd = p.wide_ws + 65 * Bp;
  f32x4 G[4][4];
  float db[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int bf = 0; bf < 4; ++bf)
#pragma unroll
    for (int bk = 0; bk < 4; ++bk) G[bf][bk] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t n_full = p.B >> 4;                             // full tiles; tile n_full is the partial one (if any)
  const int64_t n_tiles = (p.B + 15) >> 4;
  const int64_t stride = (int64_t)gridDim.x * p.groups;
  const int64_t tile0 = (int64_t)blockIdx.x * p.groups + tg;

  auto fetch_rows = [&](int (&ridx)[4], int64_t tile) __attribute__((always_inline)) {
    int64_t s0 = tile * 16 + 4 * q;
    if (tile == n_full) {                                      // partial tile: stay inside rows[]
#pragma unroll
      for (int j = 0; j < 4; ++j) ridx[j] = p.rows[min(s0 + j, p.B - 1)];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) ridx[j] = p.rows[s0 + j];
    }
  };
  auto fetch = [&](WideBwdOps &o, int64_t tile, const int (&ridx)[4]) __attribute__((always_inline)) {
    const float *dzt = dz + tile * 1024 + n * 16 + 4 * q;      // (padded to whole tiles: always in bounds)
#pragma unroll
    for (int bf = 0; bf < 4; ++bf) o.a[bf] = ld4(dzt + 256 * bf);
    o.mean = ld4(st_mean + tile * 16 + 4 * q);
    o.rstd = ld4(st_rstd + tile * 16 + 4 * q);
    if (GATHER) {
#pragma unroll
      for (int j = 0; j < 4; ++j) o.b[j] = ld4u(p.x + (int64_t)ridx[j] * p.D + kcl);
    } else if (tile == n_full) {
#pragma unroll
      for (int j = 0; j < 4; ++j) o.b[j] = ld4u(p.x + min(tile * 16 + 4 * q + j, p.B - 1) * p.D + kcl);
    } else {
      const float *xr = p.x + (tile * 16 + 4 * q) * p.D + kcl;
#pragma unroll
      for (int j = 0; j < 4; ++j) o.b[j] = ld4u(xr + (int64_t)j * p.D);
    }
  };
  // one tile: 64 MFMAs; 2 packed instructions per sample for (x - mean) rstd; the bias gradient (the same for every chunk) is
  // accumulated by the chunk-0 wave alone
  auto compute = [&](WideBwdOps &o, int64_t tile) __attribute__((always_inline)) {
    if (tile == n_full) {                                      // partial tile: samples beyond the batch contribute nothing
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool v = tile * 16 + 4 * q + j < p.B;
#pragma unroll
        for (int bf = 0; bf < 4; ++bf) o.a[bf][j] = v ? o.a[bf][j] : 0.f;
        o.mean[j] = v ? o.mean[j] : 0.f; o.rstd[j] = v ? o.rstd[j] : 0.f;
      }
    }
    if (edge) {                                                // zeros beyond the row
#pragma unroll
      for (int j = 0; j < 4; ++j) o.b[j] = ld4_row_fix(o.b[j], kcol, p.D, al4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 xh = o.b[j];
      if (fnorm) {
        const f32x4 m4 = {o.mean[j], o.mean[j], o.mean[j], o.mean[j]}, r4 = {o.rstd[j], o.rstd[j], o.rstd[j], o.rstd[j]};
        xh = (xh - m4) * r4;
        if (edge) {
#pragma unroll
          for (int bk = 0; bk < 4; ++bk) xh[bk] = (kcol + bk < p.D) ? xh[bk] : 0.f;
        }
      }
#pragma unroll
      for (int bf = 0; bf < 4; ++bf)
#pragma unroll
        for (int bk = 0; bk < 4; ++bk) G[bf][bk] = WIDE_MFMA(o.a[bf][j], xh[bk], G[bf][bk]);
    }
    if (chunk == 0) {
#pragma unroll
      for (int bf = 0; bf < 4; ++bf) db[bf] += (o.a[bf][0] + o.a[bf][1]) + (o.a[bf][2] + o.a[bf][3]);
    }
  };
  if (FULL && active) {
    const float dbw = chunk == 0 ? 1.f : 0.f;
    const int64_t cnt = tile0 < n_full ? (n_full - 1 - tile0) / stride + 1 : 0;      // this wave's full tiles
    const int64_t t_hi = n_full - 1;
    auto fetch_f = [&](WideBwdOps &o, int64_t tile, const int (&ridx)[4]) __attribute__((always_inline)) {
      const float *dzt = dz + tile * 1024 + n * 16 + 4 * q;
#pragma unroll
      for (int bf = 0; bf < 4; ++bf) o.a[bf] = ld4(dzt + 256 * bf);
      o.mean = ld4(st_mean + tile * 16 + 4 * q);
      o.rstd = ld4(st_rstd + tile * 16 + 4 * q);
      if (GATHER) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o.b[j] = ld4u(p.x + (int64_t)ridx[j] * p.D + kcol);
      } else {
#ifdef WIDE_EXP_ROW0
        const float *xr = p.x + (0 * tile * 16 + 4 * q) * p.D + kcol;
#else
        const float *xr = p.x + (tile * 16 + 4 * q) * p.D + kcol;
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) o.b[j] = ld4u(xr + (int64_t)j * p.D);
      }
    };
    auto rows_f = [&](int (&ridx)[4], int64_t tile) __attribute__((always_inline)) {
      if (GATHER) {
        const int64_t s0 = tile * 16 + 4 * q;
#pragma unroll
        for (int j = 0; j < 4; ++j) ridx[j] = p.rows[s0 + j];
      }
    };
    auto compute_f = [&](WideBwdOps &o, float valid) __attribute__((always_inline)) {
      const f32x4 v4 = {valid, valid, valid, valid};
#pragma unroll
      for (int bf = 0; bf < 4; ++bf) o.a[bf] *= v4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 xh = o.b[j];
        if (fnorm) {
          const f32x4 m4 = {o.mean[j], o.mean[j], o.mean[j], o.mean[j]}, r4 = {o.rstd[j], o.rstd[j], o.rstd[j], o.rstd[j]};
          xh = (xh - m4) * r4;
        }
#pragma unroll
        for (int bf = 0; bf < 4; ++bf)
#pragma unroll
          for (int bk = 0; bk < 4; ++bk) G[bf][bk] = WIDE_MFMA(o.a[bf][j], xh[bk], G[bf][bk]);
      }
#pragma unroll
      for (int bf = 0; bf < 4; ++bf) db[bf] += dbw * ((o.a[bf][0] + o.a[bf][1]) + (o.a[bf][2] + o.a[bf][3]));
    };
    if (cnt > 0) {
      auto tl = [&](int64_t k) { return min(tile0 + k * stride, t_hi); };
      WideBwdOps o0, o1, o2;
      int r0[4] = {0, 0, 0, 0}, r1[4] = {0, 0, 0, 0}, r2[4] = {0, 0, 0, 0};
      rows_f(r0, tl(0)); rows_f(r1, tl(1)); rows_f(r2, tl(2));
      fetch_f(o0, tl(0), r0);
      fetch_f(o1, tl(1), r1);
      // (sched_barrier: hipcc's scheduler otherwise sinks a set's loads down to their first use two tiles later — fewer live
      // registers, and every product behind an exposed memory round trip)
      for (int64_t k = 0; k < cnt; k += 3) {
        fetch_f(o2, tl(k + 2), r2); rows_f(r0, tl(k + 3));
        __builtin_amdgcn_sched_barrier(0);
        compute_f(o0, 1.f);
        __builtin_amdgcn_sched_barrier(0);
        fetch_f(o0, tl(k + 3), r0); rows_f(r1, tl(k + 4));
        __builtin_amdgcn_sched_barrier(0);
        compute_f(o1, k + 1 < cnt ? 1.f : 0.f);
        __builtin_amdgcn_sched_barrier(0);
        fetch_f(o1, tl(k + 4), r1); rows_f(r2, tl(k + 5));
        __builtin_amdgcn_sched_barrier(0);
        compute_f(o2, k + 2 < cnt ? 1.f : 0.f);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the partial tile (at most one per batch, owned by one tile group)
    if (n_tiles > n_full && n_full >= tile0 && (n_full - tile0) % stride == 0) {
      WideBwdOps o;
      int r[4] = {0, 0, 0, 0};
      if (GATHER) fetch_rows(r, n_full);
      fetch(o, n_full, r);
      compute(o, n_full);
    }
  } else if (active) {
    // three operand sets: the loads of a tile are issued two tiles (~2 x 1.2 us of MFMA) before its products; with a row gather
    // the indices are fetched one tile further ahead still, so that the x loads never wait for them
    int64_t tile = tile0;
    WideBwdOps o0, o1, o2;
    int r0[4] = {0, 0, 0, 0}, r1[4] = {0, 0, 0, 0}, r2[4] = {0, 0, 0, 0};
    if (GATHER) {
      if (tile < n_tiles) fetch_rows(r0, tile);
      if (tile + stride < n_tiles) fetch_rows(r1, tile + stride);
      if (tile + 2 * stride < n_tiles) fetch_rows(r2, tile + 2 * stride);
    }
    if (tile < n_tiles) fetch(o0, tile, r0);
    if (tile + stride < n_tiles) fetch(o1, tile + stride, r1);
    while (tile < n_tiles) {
      if (tile + 2 * stride < n_tiles) fetch(o2, tile + 2 * stride, r2);
      if (GATHER && tile + 3 * stride < n_tiles) fetch_rows(r0, tile + 3 * stride);
      compute(o0, tile);
      tile += stride;
      if (tile >= n_tiles) break;
      if (tile + 2 * stride < n_tiles) fetch(o0, tile + 2 * stride, r0);
      if (GATHER && tile + 3 * stride < n_tiles) fetch_rows(r1, tile + 3 * stride);
      compute(o1, tile);
      tile += stride;
      if (tile >= n_tiles) break;
      if (tile + 2 * stride < n_tiles) fetch(o1, tile + 2 * stride, r1);
      if (GATHER && tile + 3 * stride < n_tiles) fetch_rows(r2, tile + 3 * stride);
      compute(o2, tile);
      tile += stride;
    }
  }
  // ---- bias gradient of the tile group: lane (m, q) of the chunk-0 wave holds the sum over its samples of dz1[16 bf + m] ----
  if (active && chunk == 0) {
#pragma unroll
    for (int bf = 0; bf < 4; ++bf) {
      const float d = quad_sum16(db[bf]);
      if (q == 0) sDb[tg][16 * bf + n] = d;
    }
  }
  __syncthreads();
  if (!active) return;
  // ---- raw products -> gradients, per wave ----
  float *slab = p.slabs + (size_t)((int64_t)blockIdx.x * p.groups + tg) * p.slab_stride + p.slab_col0;
  float dgam[4] = {0.f, 0.f, 0.f, 0.f}, dbet[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int bk = 0; bk < 4; ++bk) {
    const int k = c0 + 4 * n + bk;
    const bool kv = k < p.D;
    const int kc = kv ? k : p.D - 1;
    const float gam = fnorm ? p.params[p.fn_w + kc] : 1.f, bet = fnorm ? p.params[p.fn_b + kc] : 0.f;
#pragma unroll
    for (int bf = 0; bf < 4; ++bf) {
      const f32x4 dbq = ld4(&sDb[tg][16 * bf + 4 * q]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = 16 * bf + 4 * q + i;
        const float g = G[bf][bk][i];
        if (fnorm) {
          const float w = p.params[p.w1 + f * p.D + kc];
          dgam[bk] += w * g; dbet[bk] += w * dbq[i];
        }
        if (kv) slab[p.w1 + f * p.D + k] = gam * g + bet * dbq[i];
      }
    }
    if (fnorm) {
      const float dg = quad_sum16(dgam[bk]), dt = quad_sum16(dbet[bk]);
      if (q == 0 && kv) { slab[p.fn_w + k] = dg; slab[p.fn_b + k] = dt; }
    }
  }
}
