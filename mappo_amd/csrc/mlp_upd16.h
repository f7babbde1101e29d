// mlp_upd16.h — the PPO update kernel for in_dim <= 64, layer_N <= 1, out_dim <= 16: forward + in-kernel loss (PPO actor
// objective | clipped value loss) + backward + weight-gradient accumulation, ONE WAVEFRONT PER 16-SAMPLE TILE on
// v_mfma_f32_16x16x4_f32 (r_mappo.py:91-164 with evaluate_actions, cal_value_loss and both backward passes inside).
//
// Why not the pair kernel (mlp_upd2.h): there two waves share a 32-sample tile with the FEATURES split between them, so
// every LayerNorm needs an exchange, every layer a pair barrier (11 per tile), the per-sample loss runs on one wave while
// its partner waits, and both MFMA operands of every product come from LDS.  Measured 0.33 of the fp32 MFMA peak with the
// waves parked ~40 % of their cycles.  Here a wave owns ALL 64 features of its 16 samples:
//   * 16x16x4 accumulator layout: lane (n = lane & 15, q = lane >> 4) holds features 16 b + 4 q + i (b, i = 0..3) of sample
//     n.  The B operand of a k-step needs one feature per q for sample n and the reduction order over k is free, so
//     k-step (b, i) takes k = 16 b + 4 q + i — the value the lane already holds.  Forward activations, LayerNorm (16
//     values per lane + two permlane swaps), the loss and the backward-data products  d xhat = W'^T dz  (same trick, the
//     lane's dz values are the B operand) never leave the registers: NO cross-wave synchronisation inside the tile loop,
//     and only ONE operand per MFMA (the weights) is read from LDS — a 16-byte read feeds four MFMAs.
//   * the weight-gradient products  G[f][k] += sum_s dz[f][s] xhat[k][s]  contract over SAMPLES, so both operands must
//     be transposed (lane <-> feature): the wave writes xhat / dz once to private [sample][feature] LDS tiles (row
//     stride 68: the transposed reads of lanes (n, q) hit 32 distinct banks) and reads them back k <-> sample.
//   * bias gradients are column sums of the dz tile (lane = feature, 16 reads), so they cost one register per layer.
// The accumulators of a wave are the RAW products of the whole network (critic, in_dim 54: 64 + 64 registers); the
// LayerNorm-affine transform (see raw_to_grad in mlp_impl.h) is linear in them and is applied ONCE per workgroup, after
// the waves' accumulators have been summed in LDS, by all threads.
#pragma once
#include <utility>
#include <type_traits>

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <class F, int... I>
__device__ __forceinline__ void static_for_impl16(F &f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for16(F &&f) { static_for_impl16(f, std::make_integer_sequence<int, N>{}); }

#define UPD16_THREADS 512
#define RS16 68          // row stride (floats) of the 64-column LDS arrays
#define HS16 80          // row stride of the actor head weights [action][k]: rows 1 apart -> banks 16 apart
#define DLS16 20         // row stride of the d(logits) tile [sample][action]

// LDS carve-up (floats), a compile-time function of the instantiation so that every LDS address is
// "lane pattern (one VGPR) + immediate".  W1' is staged as [f][q * CP + t] (input feature k = q * C + t, C = ceil(D / 4)).
template <int LN, int HEAD, bool WIDE, bool XL1 = false>
struct L16 {
  // XL1: layer 1 runs in its own kernels (mlp_wide16.h forward, wide_l1_bwd_kernel weight gradient): no W1 copy, no xhat0 tile
  // The three matrices are staged in FRAGMENT order: the 16 x 16 block (bo, b) is 256 consecutive floats, lane (n, q) owning
  // M[16 bo + n][4 consecutive k of its k-step group] at float offset 4 * lane, so the A operands of four MFMAs are ONE 16-byte
  // read at base + 16 * lane bytes and each 16-lane group of the ds_read_b128 touches every bank exactly once.  (Round 2 kept
  // row-major copies with strides 36 / 68: lanes (11, q) and (12, q - 1) of a group then share a bank quad — one conflict cycle
  // in five, SQ_LDS_BANK_CONFLICT = 31 % of SQ_LDS_IDX_ACTIVE, profiles/r02/dual_update_sq_pmc.txt.)
  static constexpr int CP = WIDE ? 16 : 8, NT = CP / 4, NBK = WIDE ? 4 : 2, XST = 16 * NBK + 4;
  static constexpr int W1 = 0;                                    // W1' block (bo, tc): rows 16 bo + n, columns q CP + 4 tc + i
  static constexpr int W2 = W1 + (XL1 ? 0 : HID * 4 * CP);        // W2' block (bo, b): rows 16 bo + n, columns 16 b + 4 q + i (forward A operand)
  static constexpr int W2T = W2 + (LN > 0 ? HID * HID : 0);       // W2'^T, the same with rows = k, columns = f (backward-data A operand)
  static constexpr int WH = W2T + (LN > 0 ? HID * HID : 0);       // actor: Wh' [action][k] (16 rows, stride HS16) | critic: Wh' [k]
  __host__ __device__ static constexpr int w1_at(int f, int r) { return ((f >> 4) * NT + ((r % CP) >> 2)) * 256 + ((r / CP) * 16 + (f & 15)) * 4 + (r & 3); }
  static constexpr int B1 = WH + (HEAD == 1 ? 16 * HS16 : HID), B2 = B1 + HID, BH = B2 + HID;      // folded biases
  static constexpr int FN_W = BH + 16, FN_B = FN_W + HID, G1 = FN_B + HID, T1 = G1 + HID, G2 = T1 + HID, T2 = G2 + HID;   // raw affine vectors
  static constexpr int TILES = T2 + HID;
  // per-wave tiles: xhat0 [16][XST] | xhat1 [16][68] (layer_N == 1) | scratch tile [16][68] | dlogits [16][20] (actor)
  static constexpr int UX = 0, UH = UX + (XL1 ? 0 : 16 * XST), UT = UH + (LN > 0 ? 16 * RS16 : 0), UDL = UT + 16 * RS16;
  static constexpr int WAVE_STRIDE = UDL + (HEAD == 1 ? 16 * DLS16 : 0);
  static constexpr int N_WAVES = UPD16_THREADS / WAVE;
  // epilogue (overlaid on the tile area): [1024 scratch | chunk buffer N_WAVES x CH x 256 | flat gradient (<= PMAX)]
  static constexpr int DMAX = XL1 ? 0 : (WIDE ? 64 : 32);
  static constexpr int PMAX = 2 * DMAX + HID * DMAX + 3 * HID + (LN > 0 ? HID * HID + 3 * HID : 0) + (HEAD == 1 ? 16 * HID + 16 : HID + 1);
  static constexpr int EPI4 = 1024 + N_WAVES * 4 * 256 + PMAX;
  static constexpr int TILE_AREA = N_WAVES * WAVE_STRIDE > EPI4 ? N_WAVES * WAVE_STRIDE : EPI4;    // small tile sets: the epilogue's need
  static constexpr int TOTAL = TILES + TILE_AREA;
  static constexpr int CH = (1024 + N_WAVES * 8 * 256 + PMAX <= TILE_AREA) ? 8 : 4;       // accumulators per reduction chunk
  static_assert(1024 + N_WAVES * CH * 256 + PMAX <= TILE_AREA, "epilogue buffers do not fit the tile area");
};

struct Upd16Args {
  UpdArgs u;             // params, x, rows, slabs, desc, off, B, loss inputs, partials, cfg (LdsMap / wide fields unused)
  // rows [zero_row0, zero_row1) of ANOTHER network's slab columns / loss partials that no workgroup of that network writes
  // (dual launch with unequal shares): zero-filled by this network's workgroups of the same row index
  int zero_row0, zero_row1;
  int64_t zero_col0;
  int zero_cols;
  double *zero_partials;
};

__device__ __forceinline__ float xhalf_max(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xrow_max(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float quad_max16(float v) { return xhalf_max(xrow_max(v)); }

__device__ __forceinline__ f32x4 ld4(const float *p) {
  const float4 t = *reinterpret_cast<const float4 *>(p);
  f32x4 r; r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
  return r;
}
__device__ __forceinline__ void st4(float *p, const f32x4 &v) { *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]); }

// act + LayerNorm statistics over the 64 features of sample n (16 per lane, 4 lanes); a <- xhat.  Four independent partial
// sums per reduction: a lone dependent chain of 16 adds costs the wave 16 instruction latencies.
template <bool RELU>
__device__ __forceinline__ void act_ln_fwd16(f32x4 (&a)[4], float &mean, float &rstd, uint32_t &pos) {
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < 4; ++b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a[b][i] = act_fwd<RELU>(a[b][i]);
    s += a[b];                                                   // (4-wide: packed fp32 adds)
  }
  mean = quad_sum16((s[0] + s[1]) + (s[2] + s[3])) * (1.f / HID);
  const f32x4 mean4 = {mean, mean, mean, mean};
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < 4; ++b) { a[b] -= mean4; v += a[b] * a[b]; }
  pos = 0u;                                                      // (the ReLU gate is re-derived from xhat in the backward, see ln_act_bwd16)
  rstd = __builtin_amdgcn_rsqf(quad_sum16((v[0] + v[1]) + (v[2] + v[3])) * (1.f / HID) + LN_EPS);      // v_rsq_f32 (1 ulp)
  const f32x4 rstd4 = {rstd, rstd, rstd, rstd};
#pragma unroll
  for (int b = 0; b < 4; ++b) a[b] *= rstd4;
}

// LayerNorm (no affine: folded into the consumer's weights) + activation backward: d = d/d xhat in, d/d z out.
// ReLU gate: the forward computed xhat = (a - mean) rstd with a = max(z, 0), so z <= 0  <=>  a == 0  <=>  xhat == (0 - mean) rstd
// bit for bit (the same two roundings); the gate is one compare against that threshold instead of a saved bit mask (which
// cost two instructions per element in the forward).  An element with 0 < a < ulp(mean) / 2 reads as gated — a gradient
// term of relative weight < 1e-7 on a measure-zero set.
template <bool RELU>
__device__ __forceinline__ void ln_act_bwd16(f32x4 (&d)[4], const f32x4 (&xh)[4], float mean, float rstd, uint32_t pos) {
  (void)pos;
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < 4; ++b) { s1 += d[b]; s2 += d[b] * xh[b]; }
  const float m1 = quad_sum16((s1[0] + s1[1]) + (s1[2] + s1[3])) * (1.f / HID);
  const float m2 = quad_sum16((s2[0] + s2[1]) + (s2[2] + s2[3])) * (1.f / HID);
  const float inv_rstd = __builtin_amdgcn_rcpf(rstd);
  const float c0 = -m1 * rstd, c1 = -m2 * rstd;                  // da = rstd (d - m1 - xhat m2) as two fmas per element
  const f32x4 c04 = {c0, c0, c0, c0}, c14 = {c1, c1, c1, c1}, rstd4 = {rstd, rstd, rstd, rstd};
  const float thr = (0.f - mean) * rstd;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const f32x4 da = xh[b] * c14 + (d[b] * rstd4 + c04);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (RELU) {
        d[b][i] = xh[b][i] > thr ? da[i] : 0.f;
      } else {
        const float a = xh[b][i] * inv_rstd + mean;
        d[b][i] = da[i] * (1.f - a * a);
      }
    }
  }
}

// out[bo] += W'[16 bo + n][16 b + 4 q + i] * h[b][i]  (W' in fragment order, see L16): hidden -> hidden forward, and with
// the transposed copy the backward-data product.  One 16-byte operand read feeds four MFMAs.  (fp32 MFMAs execute on the
// vector ALU — SQ_VALU_MFMA_COEXEC_CYCLES reads 0 for this kernel — and the waves wait on LDS for 2 % of their cycles, so
// there is nothing to gain from software-pipelining the operand reads; the plain form needs the fewest registers.)
__host__ __device__ constexpr int frag64_at(int row, int col) {   // element (row, col) of a 64 x 64 matrix in fragment order
  return ((row >> 4) * 4 + (col >> 4)) * 256 + (((col >> 2) & 3) * 16 + (row & 15)) * 4 + (col & 3);
}
__device__ __forceinline__ void hidden_fwd16(f32x4 (&out)[4], const f32x4 (&h)[4], const float *sW, int n, int q) {
  const float *base = sW + (q * 16 + n) * 4;                     // fragment order: block (bo, b) at (4 bo + b) * 256, lane * 4 inside
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    f32x4 a[4];
#pragma unroll
    for (int bo = 0; bo < 4; ++bo) a[bo] = ld4(base + (4 * bo + b) * 256);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int bo = 0; bo < 4; ++bo) out[bo] = mfma16(a[bo][i], h[b][i], out[bo]);
  }
}

// out[bo] (input features 16 bo + ..) = sum_f W'[f][16 bo + n] * dz[f], f = 16 b + 4 q + i: backward-data of a hidden layer.
// sWT = W'^T in fragment order, so the four k-steps of a block are one 16-byte read, as in the forward.
__device__ __forceinline__ void hidden_bwd16(f32x4 (&out)[4], const f32x4 (&dz)[4], const float *sWT, int n, int q) {
#pragma unroll
  for (int bo = 0; bo < 4; ++bo) { out[bo][0] = 0.f; out[bo][1] = 0.f; out[bo][2] = 0.f; out[bo][3] = 0.f; }
  hidden_fwd16(out, dz, sWT, n, q);
}

// G[bf][bk] (rows f = 16 bf + 4 q + i, columns k = 16 bk + n) += sum_s UA[s][16 bf + ..] * UB[s][16 bk + ..].
// gb[bf] += the lane's A operands (dz[16 bf + n] of samples 4 q + j): per-lane partial sums of the bias gradient (reduced
// over q once, in the epilogue) — no separate pass over the dz tile.
template <int NBF, int NBK>
__device__ __forceinline__ void dw_accum16(f32x4 (&G)[NBF][NBK], float (&gb)[NBF], const float *UA, int sa, const float *UB, int sb, int n, int q) {
  const float *pa = UA + 4 * q * sa + n, *pb = UB + 4 * q * sb + n;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float a[NBF], b[NBK];
#pragma unroll
    for (int bf = 0; bf < NBF; ++bf) a[bf] = pa[j * sa + 16 * bf];
#pragma unroll
    for (int bk = 0; bk < NBK; ++bk) b[bk] = pb[j * sb + 16 * bk];
#pragma unroll
    for (int bf = 0; bf < NBF; ++bf) {
      gb[bf] += a[bf];
#pragma unroll
      for (int bk = 0; bk < NBK; ++bk) G[bf][bk] = mfma16(a[bf], b[bk], G[bf][bk]);
    }
  }
}

// sum over the 16 samples of column `col` of a [16][stride] tile
__device__ __forceinline__ float col_sum16(const float *U, int stride, int col) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int s = 0; s < 16; s += 2) { s0 += U[s * stride + col]; s1 += U[(s + 1) * stride + col]; }
  return s0 + s1;
}

template <bool WIDE>
struct Prefetch16 {
  float xv[WIDE ? 16 : 8];   // flat: float4 chunks (lane + 64 j) of the tile's contiguous [16][D] block | gather: features q C + t of sample n
  float f0, f1, f2, f3;      // actor: action, old_logp, adv, active | critic: v_old, ret, active, -
  uint32_t dead;             // actor: bit i set <=> available_actions[4 i + q] == 0
  int n_valid;
  bool flat;
};

// Input rows of the NEXT tile into registers (`tile` is wave-uniform).  Whole-buffer minibatches (rows == NULL, full tile,
// 16-byte aligned x): the tile is ONE contiguous 16 x D block, fetched as fully coalesced 16-byte buffer loads — the
// descriptor's bounds check returns 0 beyond the block, so there is no per-lane clamp — and redistributed through the
// wave's xhat0 tile at the commit.  Gathered rows / ragged last tile: lane (n, q) fetches its own features.
template <int HEAD, bool WIDE>
__device__ __forceinline__ void prefetch16(Prefetch16<WIDE> &pf, const UpdArgs &p, int64_t tile, int64_t n_tiles, int D, int C, int A,
                                           int lane, int n, int q) {
  constexpr int NV = WIDE ? 16 : 8;
  int64_t base = tile * 16;
  int nv = (tile < n_tiles) ? (int)min((int64_t)16, p.B - base) : 0;
  if (HEAD == 3 && p.seq_nc > 0 && tile < n_tiles) {               // (t, 16 sequences) tiles of a time-major [L][seq_nc] minibatch
    const int n_ct = (p.seq_nc + 15) >> 4;
    const int64_t t = tile / n_ct;
    const int j0 = (int)(tile - t * n_ct) * 16;
    base = t * p.seq_nc + j0;
    nv = min(16, p.seq_nc - j0);
  }
  pf.n_valid = nv;
  pf.flat = p.rows == nullptr && nv == 16 && (((uintptr_t)(p.x + base * D)) & 15) == 0;
  pf.f0 = pf.f1 = pf.f2 = pf.f3 = 0.f;
  pf.dead = 0u;
  if (nv == 0) return;                                           // no such tile (wave-uniform)
  if (pf.flat) {
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + base * D), 0, 16 * D * 4, 0x00020000);
#pragma unroll
    for (int j = 0; j < NV / 4; ++j) {
      const f32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rx, lane * 16, 1024 * j, 0);
      pf.xv[4 * j + 0] = t[0]; pf.xv[4 * j + 1] = t[1]; pf.xv[4 * j + 2] = t[2]; pf.xv[4 * j + 3] = t[3];
    }
    if constexpr (HEAD == 1) {
      pf.f0 = (p.actions + base)[n]; pf.f1 = (p.old_logp + base)[n]; pf.f2 = (p.adv + base)[n]; pf.f3 = (p.active + base)[n];
      if (p.avail) {
        // availability of action 4 i + q: one bounds-checked dword per i (0 = unavailable | beyond A: masked below)
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)(p.avail + base * A), 0, 16 * A * 4, 0x00020000);
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b32(ra, (n * A + q) * 4, 16 * i, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) pf.dead |= ((4 * i + q < A && v[i] == 0.f) ? 1u : 0u) << i;
      }
    } else if constexpr (HEAD == 2) {
      pf.f0 = (p.v_old + base)[n]; pf.f1 = (p.returns + base)[n]; pf.f2 = (p.active + base)[n];
    }
    return;
  }
  const bool ok = n < nv;
  const int64_t row = ok ? (p.rows ? (int64_t)p.rows[base + n] : base + n) : 0;
  // (q made opaque per call: otherwise hipcc hoists the clamped 64-bit element offsets out of the tile loop — 32 registers)
  int qo = q;
  asm volatile("" : "+v"(qo));
  const float *src = p.x + row * D;
#pragma unroll
  for (int t = 0; t < NV; ++t) pf.xv[t] = src[min(qo * C + t, D - 1)];      // unconditional clamped loads, masked at the commit
  if constexpr (HEAD == 1) {
    pf.f0 = p.actions[row]; pf.f1 = p.old_logp[row]; pf.f2 = p.adv[row]; pf.f3 = p.active[row];
    if (p.avail) {
      const float *av = p.avail + row * A;
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = av[min(4 * i + qo, A - 1)];
#pragma unroll
      for (int i = 0; i < 4; ++i) pf.dead |= ((4 * i + q < A && v[i] == 0.f) ? 1u : 0u) << i;
    }
  } else if constexpr (HEAD == 2) {
    pf.f0 = p.v_old[row]; pf.f1 = p.returns[row]; pf.f2 = p.active[row];
  }
}

// XL1: the tile's input is z1 [B][64] (pre-activation of layer 1, bias included; minibatch order) from wide_l1_fwd16_kernel:
// lane (n, q) fetches z1[16 b + 4 q .. + 3] of sample n as four bounds-checked 16-byte loads (rows beyond B read as 0).
template <int HEAD>
__device__ __forceinline__ void prefetch16x(Prefetch16<true> &pf, const UpdArgs &p, const float *z1, int64_t tile, int64_t n_tiles, int A,
                                            int n, int q) {
  const int64_t base = tile * 16;
  const int nv = (tile < n_tiles) ? (int)min((int64_t)16, p.B - base) : 0;
  pf.n_valid = nv;
  pf.flat = true;
  pf.f0 = pf.f1 = pf.f2 = pf.f3 = 0.f;
  pf.dead = 0u;
  if (nv == 0) return;
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void *)(z1 + base * HID), 0, nv * HID * 4, 0x00020000);
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const f32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rz, (n * HID + 4 * q) * 4, 64 * b, 0);
    pf.xv[4 * b + 0] = t[0]; pf.xv[4 * b + 1] = t[1]; pf.xv[4 * b + 2] = t[2]; pf.xv[4 * b + 3] = t[3];
  }
  const bool ok = n < nv;
  const int64_t row = ok ? (p.rows ? (int64_t)p.rows[base + n] : base + n) : 0;
  if constexpr (HEAD == 1) {
    pf.f0 = p.actions[row]; pf.f1 = p.old_logp[row]; pf.f2 = p.adv[row]; pf.f3 = p.active[row];
    if (p.avail) {
      const float *av = p.avail + row * A;
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = av[min(4 * i + q, A - 1)];
#pragma unroll
      for (int i = 0; i < 4; ++i) pf.dead |= ((4 * i + q < A && v[i] == 0.f) ? 1u : 0u) << i;
    }
  } else if constexpr (HEAD == 2) {
    pf.f0 = p.v_old[row]; pf.f1 = p.returns[row]; pf.f2 = p.active[row];
  }
}

// exp / log on the hardware transcendental units (v_exp_f32 / v_log_f32, 1 ulp, as the GRU gate nonlinearities): the libm forms are
// ~20 / ~25 instructions each on a pipe the MFMAs share, five per lane and tile in the actor's loss.  Arguments here are
// z - zmax <= 0 (exp underflows to exactly 0 for masked logits, as libm's), se in [1, A], and a log-ratio of O(1).
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(1.44269504088896341f * x); }
__device__ __forceinline__ float fast_log(float x) { return 0.693147180559945309f * __builtin_amdgcn_logf(x); }

// Actor objective of one sample in the head layout: lane (n, q) holds z[i] = logit of action 4 i + q (A <= 16).  On return z
// holds d(actor objective) / d logits.  Same expressions as actor_loss_regs (mlp_core.h); the sums over actions run over
// the lane's registers first and the 4 lanes of the sample second.
__device__ __forceinline__ void actor_loss_quad(f32x4 &z, int A, int q, uint32_t dead, int act, float old_lp, float adv, float active,
                                                bool count, const mappo_ppo_cfg &cfg, float scale_pi, float (&lacc)[3]) {
  const float clip = cfg.clip_param;
  float zm = -FLT_MAX;
  bool valid[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    valid[i] = 4 * i + q < A;
    if ((dead >> i) & 1u) z[i] = -1e10f;
    if (valid[i]) zm = fmaxf(zm, z[i]);
  }
  const float zmax = quad_max16(zm);
  float e[4], se = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) { e[i] = valid[i] ? fast_exp(z[i] - zmax) : 0.f; se += e[i]; }
  se = quad_sum16(se);
  const float log_se = fast_log(se), inv_se = __builtin_amdgcn_rcpf(se);
  float hp = 0.f, za = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float l_ = (z[i] - zmax) - log_se;
    if (valid[i]) hp += (e[i] * inv_se) * fmaxf(l_, -FLT_MAX);
    if (valid[i] && 4 * i + q == act) za = z[i];
  }
  const float H = -quad_sum16(hp);
  const float z_act = quad_sum16(za);                     // one lane / register of the sample is non-zero: exact
  const float logp = (z_act - zmax) - log_se;
  const float ratio = fast_exp(logp - old_lp);
  const float s1 = ratio * adv, s2 = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip) * adv;
  const float w = cfg.use_policy_active_masks ? active : 1.f;
  const float dlogp = (s1 <= s2) ? -(w * scale_pi) * adv * ratio : 0.f;
  const float ce = cfg.entropy_coef * w * scale_pi;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float l_ = (z[i] - zmax) - log_se;
    const float pa = e[i] * inv_se;
    float g = dlogp * ((4 * i + q == act ? 1.f : 0.f) - pa) + ce * pa * (l_ + H);
    if (!valid[i] || ((dead >> i) & 1u) || !count) g = 0.f;
    z[i] = g;
  }
  if (count && q == 0) {
    lacc[0] += w * fminf(s1, s2);
    lacc[1] += w * H;
    lacc[2] += ratio;
  }
}

// critic_loss_lane with float statistics (a lane sees a handful of samples; the cross-lane sums are taken in double)
__device__ __forceinline__ float critic_loss16(float v, float vo, float ret, float active, const mappo_ppo_cfg &cfg, const LossScales &ls,
                                               bool count, float (&lacc)[3]) {
  double l4[4] = {0.0, 0.0, 0.0, 0.0};
  const float dv = critic_loss_lane(v, vo, ret, active, cfg, ls, l4);
  if (count) lacc[0] += (float)l4[0];
  return dv;
}

// Epilogue transform of one weight matrix with a LayerNorm affine on its input (raw_to_grad, mlp_impl.h), on the
// workgroup's summed raw products in R:  G at R[wo + f K + k], db at R[bo + f] (f < F <= 8 NJ, k < K <= 64).
//   R[wo..] <- gam[k] G + bet[k] db[f];  R[go + k] <- sum_f W[f][k] G[f][k];  R[to + k] <- sum_f W[f][k] db[f]
// Thread (k = tid & 63, part = tid >> 6) owns rows f = part + 8 j; w[j] = W[f][k] (raw, preloaded from global memory).
template <int NJ>
__device__ __forceinline__ void affine_epilogue16(float *R, int wo, int bo, int F, int K, const float *gam, const float *bet,
                                                  const float (&w)[NJ], int go, int to, float *scr) {
  const int k = threadIdx.x & 63, part = threadIdx.x >> 6;
  float sg = 0.f, sb = 0.f;
  if (k < K) {
    const float gk = gam[k], bk = bet[k];
    float g[NJ], d[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int f = min(part + 8 * j, F - 1); g[j] = R[wo + f * K + k]; d[j] = R[bo + f]; }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int f = part + 8 * j;
      if (f < F) {
        sg += w[j] * g[j]; sb += w[j] * d[j];
        R[wo + f * K + k] = gk * g[j] + bk * d[j];
      }
    }
  }
  scr[part * 64 + k] = sg;
  scr[512 + part * 64 + k] = sb;
  __syncthreads();
  if (threadIdx.x < 64 && k < K) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) { a += scr[pp * 64 + k]; b += scr[512 + pp * 64 + k]; }
    R[go + k] = a; R[to + k] = b;
  }
  __syncthreads();
}

// Workgroup `bid` of the `nb` workgroups that share this network's rows.  512 threads (8 waves, 2 per SIMD).
template <bool RELU, int LN, int HEAD, bool WIDE, bool XL1 = false>
__device__ __forceinline__ void update16_body(const Upd16Args &P, float *lds, const int bid, const int nb) {
  static_assert(LN <= 1 && (HEAD == 1 || HEAD == 2 || HEAD == 3), "update16: layer_N <= 1; heads: actor loss, critic loss, trunk (gradient in)");
  static_assert(!XL1 || WIDE, "XL1 uses the 16-register prefetch block");
  constexpr int NV = WIDE ? 16 : 8, NBK = WIDE ? 4 : 2;
  const UpdArgs &p = P.u;
  typedef L16<LN, HEAD, WIDE, XL1> M;
  // XL1 workspace (mappo_wide_workspace_floats): dz1 [64][B] feature-major | mean0 [B] | rstd0 [B] | z1 [B][64]
  const float *z1 = XL1 ? p.wide_ws + 66 * ((p.B + 15) & ~(int64_t)15) : nullptr;      // wide16_z1_offset (mlp_wide16.h)
  const NetOff &o = p.off;
  const int lane = threadIdx.x & (WAVE - 1), n = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)), n_waves = blockDim.x / WAVE;
  const int D = p.desc.in_dim, A = p.desc.out_dim, C = (D + 3) >> 2;
  const bool fnorm = p.desc.use_feature_norm != 0;
  const float inv_D = 1.0f / (float)D;
  // sequence tiling + blocked d(trunk output) (recurrent training, gru_train16.hip); wide inputs (XL1) are only given multiples of
  // 16 sequences, for which the flat 16-row tiles already are the sequence tiles
  const bool seq = HEAD == 3 && p.seq_nc > 0;
  const int64_t n_tiles = seq ? (p.B / p.seq_nc) * ((p.seq_nc + 15) >> 4) : (p.B + 15) / 16;
  const int64_t tile_stride = (int64_t)nb * n_waves;
  const int64_t tile0 = (int64_t)wave * nb + bid;          // remainder of the last round spreads over all CUs
  Prefetch16<WIDE> pf;
  if constexpr (XL1) prefetch16x<HEAD>(pf, p, z1, tile0, n_tiles, A, n, q);
  else prefetch16<HEAD, WIDE>(pf, p, tile0, n_tiles, D, C, A, lane, n, q);
  STAMP_DECL

  // ---- stage the network: W' = W * gamma_in (columns), b' = b + W beta_in.  Thread (f8 = tid / 8, part = tid % 8) owns the
  // columns k = part + 8 j of row f8 of every matrix: ALL global loads are issued before the first LDS store (one memory
  // latency for the whole network), the scaled copy is stored and the bias dot products are reduced over the 8 lanes of a row.
  {
    static_assert(UPD16_THREADS == 512, "staging assumes 64 rows x 8 threads");
    const float *__restrict__ g = p.params;
    const int tid = threadIdx.x, f8 = tid >> 3, part = tid & 7;
    const int vid = tid >> 6, ve = tid & 63;                      // vector staging: 8 vectors x 64 entries
    float w1r[M::CP / 2], w2r[8], whr[8], vecv, b1raw, b2raw = 0.f, bhraw = 0.f;
    // W1: slot r = part + 8 j of row f8 <-> (chunk r / CP, t = r % CP) <-> input feature k = chunk * C + t
    if constexpr (!XL1) {
#pragma unroll
      for (int j = 0; j < M::CP / 2; ++j) {
        const int r = part + 8 * j, k = (r / M::CP) * C + (r % M::CP);
        w1r[j] = g[o.w1 + f8 * D + min(k, D - 1)];
      }
    }
    if constexpr (LN > 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) w2r[j] = g[o.w2[0] + f8 * HID + part + 8 * j];
    }
    if constexpr (HEAD == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) whr[j] = g[o.wh + min(f8, A - 1) * HID + part + 8 * j];
      bhraw = g[o.bh + min(f8, A - 1)];
    } else if constexpr (HEAD == 2) {
      whr[0] = g[o.wh + ve];
      bhraw = g[o.bh];
    }
    b1raw = g[o.b1 + f8];
    if constexpr (LN > 0) b2raw = g[o.b2[0] + f8];
    {
      // vectors: 0 fn_w | 1 fn_b | 2 g1 | 3 t1 | 4 g2 | 5 t2 (6, 7 idle)
      int src = -1; float fill = 0.f;
      if (vid == 0) { if (fnorm && !XL1) { if (ve < D) src = o.fn_w + ve; } else fill = 1.f; }
      else if (vid == 1) { if (fnorm && !XL1 && ve < D) src = o.fn_b + ve; }
      else if (vid == 2) src = o.ln1_w + ve;
      else if (vid == 3) src = o.ln1_b + ve;
      else if (vid == 4) { if (LN > 0) src = o.ln2_w[0] + ve; }
      else if (vid == 5) { if (LN > 0) src = o.ln2_b[0] + ve; }
      const float ld = g[src >= 0 ? src : 0];
      vecv = src >= 0 ? ld : fill;
    }
    // zero-fill this wave's xhat0 tile (its padding columns feed gradient columns that are never stored)
    if constexpr (!XL1) {
      float *Ux0 = lds + M::TILES + wave * M::WAVE_STRIDE + M::UX;
      for (int e = lane; e < 16 * M::XST / 4; e += WAVE) *reinterpret_cast<float4 *>(Ux0 + 4 * e) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (vid < 6) lds[M::FN_W + vid * HID + ve] = vecv;             // FN_W, FN_B, G1, T1, G2, T2 are consecutive
    __syncthreads();
    const int gl = LN > 0 ? M::G2 : M::G1, tl = LN > 0 ? M::T2 : M::T1;       // LayerNorm feeding the head
    float pb1 = 0.f, pb2 = 0.f, pbh = 0.f;
    if constexpr (!XL1) {
#pragma unroll
      for (int j = 0; j < M::CP / 2; ++j) {
        const int r = part + 8 * j, t = r % M::CP, k = (r / M::CP) * C + t;
        const bool ok = t < C && k < D;                          // padding slots of the staged copy are zero
        const int kc = min(k, 63);
        const float w = ok ? w1r[j] : 0.f;
        lds[M::W1 + M::w1_at(f8, r)] = w * lds[M::FN_W + kc];
        pb1 += w * lds[M::FN_B + kc];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = part + 8 * j;
      if constexpr (LN > 0) {
        const float wsc = w2r[j] * lds[M::G1 + k];
        lds[M::W2 + frag64_at(f8, k)] = wsc;
        lds[M::W2T + frag64_at(k, f8)] = wsc;
        pb2 += w2r[j] * lds[M::T1 + k];
      }
      if constexpr (HEAD == 1) {
        if (f8 < 16) {
          const float w = f8 < A ? whr[j] : 0.f;
          lds[M::WH + f8 * HS16 + k] = w * lds[gl + k];
          pbh += w * lds[tl + k];
        }
      }
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
      pb1 += __shfl_xor(pb1, off, WAVE);
      if constexpr (LN > 0) pb2 += __shfl_xor(pb2, off, WAVE);
      if constexpr (HEAD == 1) pbh += __shfl_xor(pbh, off, WAVE);
    }
    if (part == 0) {
      lds[M::B1 + f8] = b1raw + pb1;
      if constexpr (LN > 0) lds[M::B2 + f8] = b2raw + pb2;
      if constexpr (HEAD == 1) { if (f8 < 16) lds[M::BH + f8] = f8 < A ? bhraw + pbh : 0.f; }
    }
    if constexpr (HEAD == 2) {
      if (tid < 64) {
        lds[M::WH + tid] = whr[0] * lds[gl + tid];
        const float sum = wave_sum_f(whr[0] * lds[tl + tid]);
        if (tid == 0) lds[M::BH] = bhraw + sum;
      }
    }
    __syncthreads();
  }

  STAMP(0);   // staging + fold
  float *Ux = lds + M::TILES + wave * M::WAVE_STRIDE + M::UX;
  float *Uh = lds + M::TILES + wave * M::WAVE_STRIDE + M::UH;
  float *Ut = lds + M::TILES + wave * M::WAVE_STRIDE + M::UT;
  float *Udl = lds + M::TILES + wave * M::WAVE_STRIDE + M::UDL;
  const int xs = M::XST;

  LossScales ls = {};
  if constexpr (HEAD != 3) ls = loss_scales(p.cfg, p.mb_moments, p.vn_state);
  // (wave-uniform values computed on the vector ALU: move them to scalar registers, the vector file is what is scarce here)
  ls.scale_pi = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ls.scale_pi)));
  ls.scale_v = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ls.scale_v)));
  ls.vn_mean = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ls.vn_mean)));
  ls.vn_sd = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ls.vn_sd)));
  float lacc[3] = {0.f, 0.f, 0.f};

  // ---- raw-product accumulators of this wave ----
  f32x4 gW1[XL1 ? 1 : 4][XL1 ? 1 : NBK], gW2[LN > 0 ? 4 : 1][LN > 0 ? 4 : 1], gWh[1][HEAD == 1 ? 4 : 1];
  f32x4 gB1x[XL1 ? 4 : 1];                                      // XL1: per-lane sums of dz1 (feature 16 b + 4 q + i of the lane's samples)
  // bias-gradient partials: gB1/gB2[bf] = sum over this lane's samples of dz[16 bf + n] (reduced over q in the epilogue);
  // gBh: actor = the same for d logits (action n), critic = per-lane sum of dv; gWc: critic raw head product, lane = k
  float gB1[4] = {0.f, 0.f, 0.f, 0.f}, gB2[4] = {0.f, 0.f, 0.f, 0.f}, gBh[1] = {0.f}, gWc = 0.f;
  // HEAD 3 (trunk only: the gradient arrives at the trunk output, feature-major dHT [64][B]): the last LayerNorm keeps its
  // affine (out = gamma xhat + beta), its gradients are column sums of the tile's dH o xhat and dH
  float gLg_f = 0.f, gLb_f = 0.f;                                // lane = feature (column sums of the wave's scratch tile, like gWc)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int bf = 0; bf < 4; ++bf) {
      if constexpr (!XL1) {
#pragma unroll
        for (int bk = 0; bk < NBK; ++bk) gW1[bf][bk][i] = 0.f;
      } else {
        gB1x[bf][i] = 0.f;
      }
      if constexpr (LN > 0) {
#pragma unroll
        for (int bk = 0; bk < 4; ++bk) gW2[bf][bk][i] = 0.f;
      }
    }
#pragma unroll
    for (int bk = 0; bk < (HEAD == 1 ? 4 : 1); ++bk) gWh[0][bk][i] = 0.f;
  }
  // lane constants of the xhat0 commit: this lane's chunk holds input features [q C, q C + nvl)
  const int nvl = max(0, min(C, D - q * C));
  const float n_empty = (float)(NV - nvl);
  const float *rawp = Ux + n * D + q * C;                       // flat commit: own features inside the linear [16][D] block
  float *x0p = Ux + n * xs + q * NV;                            // xhat0 [sample][q NV + t]: chunk-major columns (see the epilogue)

  // (with 128 weight-gradient accumulators — inputs 33..64 wide — there is no room for a second set: the loads then go out at
  // the top of their own tile)
  constexpr bool DH_AHEAD = HEAD == 3 && (XL1 || !WIDE);
  f32x4 dh_next[DH_AHEAD ? 4 : 1];
  if constexpr (DH_AHEAD) {
    if (seq) {
      const float *db = p.dHT + min(tile0, n_tiles - 1) * 1024 + lane * 4;
#pragma unroll
      for (int b = 0; b < 4; ++b) dh_next[b] = ld4(db + b * 256);
    } else {
      const float *dcol = p.dHT + min(tile0 * 16 + n, p.B - 1);
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) dh_next[b][i] = dcol[(int64_t)(16 * b + 4 * q + i) * p.B];
    }
  }
  for (int64_t tile = tile0; tile < n_tiles; tile += tile_stride) {
    const int n_valid = pf.n_valid;
    const bool live = n < n_valid;
    const float c0 = pf.f0, c1 = pf.f1, c2 = pf.f2, c3 = pf.f3;
    const uint32_t cdead = pf.dead;
    f32x4 xh[4];
    float mean1, rstd1, mean2 = 0.f, rstd2 = 1.f;
    uint32_t pos1, pos2 = 0u;
    if constexpr (XL1) {
      // layer 1 ran in wide_l1_fwd16_kernel: the prefetched registers ARE the pre-activations (bias included)
#pragma unroll
      for (int b = 0; b < 4; ++b) { xh[b][0] = pf.xv[4 * b]; xh[b][1] = pf.xv[4 * b + 1]; xh[b][2] = pf.xv[4 * b + 2]; xh[b][3] = pf.xv[4 * b + 3]; }
      STAMP(1);
    } else {
    // ---- xhat0: LayerNorm over the D input features (mlp.py:45,51-52); lane holds k = q C + t ----
    float x0[NV];
    {
      if (pf.flat) {
        // the contiguous block lands in the tile as it is (linear [16][D]; lanes beyond the block hold the zeros of the
        // bounds-checked loads and land inside the tile, too), then every lane reads its own features back
#pragma unroll
        for (int j = 0; j < NV / 4; ++j)
          *reinterpret_cast<float4 *>(Ux + 4 * (lane + 64 * j)) = make_float4(pf.xv[4 * j], pf.xv[4 * j + 1], pf.xv[4 * j + 2], pf.xv[4 * j + 3]);
        wave_lds_sync();
#pragma unroll
        for (int t = 0; t < NV; ++t) x0[t] = rawp[t];
        wave_lds_sync();
      } else {
#pragma unroll
        for (int t = 0; t < NV; ++t) x0[t] = pf.xv[t];
      }
      // slots beyond the lane's features hold 0: they add (0 - mean)^2 to the centred sum (taken out again below) and meet
      // zero weights in layer 1, so one select per slot is all the masking there is
      float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < NV; ++t) { x0[t] = (t < nvl) ? x0[t] : 0.f; s4[t & 3] += x0[t]; }
      if (fnorm) {
        const float mean = quad_sum16((s4[0] + s4[1]) + (s4[2] + s4[3])) * inv_D;
        float v4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NV; ++t) { x0[t] -= mean; v4[t & 3] += x0[t] * x0[t]; }
        const float var = quad_sum16(((v4[0] + v4[1]) + (v4[2] + v4[3])) - n_empty * mean * mean);
        const float rstd = __builtin_amdgcn_rsqf(fmaxf(var, 0.f) * inv_D + LN_EPS);
#pragma unroll
        for (int t = 0; t < NV; ++t) x0[t] *= rstd;
      }
      // column q NV + t of the tile <-> input feature q C + t: chunks do not overlap, so the lane stores all its slots as
      // 16-byte writes (slots beyond its features land in columns whose gradient columns are never stored)
#pragma unroll
      for (int j = 0; j < NV / 4; ++j)
        *reinterpret_cast<float4 *>(x0p + 4 * j) = make_float4(x0[4 * j], x0[4 * j + 1], x0[4 * j + 2], x0[4 * j + 3]);
    }

    STAMP(1);   // xhat0
    // ---- layer 1 ----
#pragma unroll
    for (int bo = 0; bo < 4; ++bo) xh[bo] = ld4(lds + M::B1 + 16 * bo + 4 * q);
#pragma unroll
    for (int tc = 0; tc < NV / 4; ++tc) {
      if (4 * tc < C) {
        f32x4 a[4];
#pragma unroll
        for (int bo = 0; bo < 4; ++bo) a[bo] = ld4(lds + M::W1 + ((bo * M::NT + tc) * 64 + lane) * 4);
        // (all four k-steps of a started group: slots beyond C meet zero weights — up to 12 idle MFMAs per tile instead of
        // a uniform branch per k-step, each of which ended a basic block in front of the next group's LDS reads)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int bo = 0; bo < 4; ++bo) xh[bo] = mfma16(a[bo][i], x0[4 * tc + i], xh[bo]);
      }
    }
    }
    STAMP(2);   // layer 1 MFMA
    // next tile, under this tile's MFMAs (issued after the loss instead, the prefetch registers live through the
    // register-hungry backward pass only — but that is where the pressure peaks: 106 spilled registers, measured)
    if constexpr (XL1) prefetch16x<HEAD>(pf, p, z1, tile + tile_stride, n_tiles, A, n, q);
    else prefetch16<HEAD, WIDE>(pf, p, tile + tile_stride, n_tiles, D, C, A, lane, n, q);
    f32x4 dh[HEAD == 3 ? 4 : 1];
    if constexpr (HEAD == 3) {
      // d(trunk output): this tile's values were fetched a tile ago; the next tile's 16 dwords per lane (64-byte segments)
      // go out now, with the next tile's rows
      if constexpr (DH_AHEAD) {
#pragma unroll
        for (int b = 0; b < 4; ++b) dh[b] = dh_next[b];
        if (seq) {
          const float *db = p.dHT + min(tile + tile_stride, n_tiles - 1) * 1024 + lane * 4;
#pragma unroll
          for (int b = 0; b < 4; ++b) dh_next[b] = ld4(db + b * 256);
        } else {
          const float *dcol = p.dHT + min((tile + tile_stride) * 16 + n, p.B - 1);
#pragma unroll
          for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) dh_next[b][i] = dcol[(int64_t)(16 * b + 4 * q + i) * p.B];
        }
      } else if (seq) {
        const float *db = p.dHT + tile * 1024 + lane * 4;
#pragma unroll
        for (int b = 0; b < 4; ++b) dh[b] = ld4(db + b * 256);
      } else {
        const float *dcol = p.dHT + min(tile * 16 + n, p.B - 1);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) dh[b][i] = dcol[(int64_t)(16 * b + 4 * q + i) * p.B];
      }
    }
    act_ln_fwd16<RELU>(xh, mean1, rstd1, pos1);
    STAMP(3);   // prefetch issue + act/LN 1
    // ---- hidden layer ----
    if constexpr (LN > 0) {
#pragma unroll
      for (int b = 0; b < 4; ++b) st4(Uh + n * RS16 + 16 * b + 4 * q, xh[b]);
      f32x4 h2[4];
#pragma unroll
      for (int bo = 0; bo < 4; ++bo) h2[bo] = ld4(lds + M::B2 + 16 * bo + 4 * q);
      hidden_fwd16(h2, xh, lds + M::W2, n, q);
      STAMP(4);   // layer 2 MFMA
      act_ln_fwd16<RELU>(h2, mean2, rstd2, pos2);
#pragma unroll
      for (int b = 0; b < 4; ++b) xh[b] = h2[b];
    }
    STAMP(5);   // act/LN 2
    // xh = xhat of the last LayerNorm
    f32x4 dx[4];
    if constexpr (HEAD == 3) {
      const float *gam = lds + (LN > 0 ? M::G2 : M::G1);       // raw gamma of the LayerNorm that ends the trunk
#pragma unroll
      for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dh[b][i] = live ? dh[b][i] : 0.f;
        st4(Ut + n * RS16 + 16 * b + 4 * q, dh[b] * xh[b]);
      }
      wave_lds_sync();
      gLg_f += col_sum16(Ut, RS16, lane);
      wave_lds_sync();
#pragma unroll
      for (int b = 0; b < 4; ++b) st4(Ut + n * RS16 + 16 * b + 4 * q, dh[b]);
      wave_lds_sync();
      gLb_f += col_sum16(Ut, RS16, lane);
      wave_lds_sync();
#pragma unroll
      for (int b = 0; b < 4; ++b) dx[b] = dh[b] * ld4(gam + 16 * b + 4 * q);
    } else if constexpr (HEAD == 2) {
      // ---- critic head (out_dim 1) on the VALU ----
      f32x4 wv[4];
      float acc = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        wv[b] = ld4(lds + M::WH + 16 * b + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc += wv[b][i] * xh[b][i];
      }
      const float v = quad_sum16(acc) + lds[M::BH];
      float dv = critic_loss16(v, c0, c1, c2, p.cfg, ls, live && q == 0, lacc);
      dv = live ? dv : 0.f;
      if (q == 0) gBh[0] += dv;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        f32x4 t;
#pragma unroll
        for (int i = 0; i < 4; ++i) { t[i] = dv * xh[b][i]; dx[b][i] = wv[b][i] * dv; }
        st4(Ut + n * RS16 + 16 * b + 4 * q, t);
      }
      wave_lds_sync();
      gWc += col_sum16(Ut, RS16, lane);                       // raw head product G[k], lane = k
      wave_lds_sync();
    } else {
      // ---- actor head: logits of action 4 i + q in z[i] (head row m = 4 q + i <-> action 4 (m & 3) + (m >> 2)) ----
#pragma unroll
      for (int b = 0; b < 4; ++b) st4(Ut + n * RS16 + 16 * b + 4 * q, xh[b]);
      f32x4 z;
#pragma unroll
      for (int i = 0; i < 4; ++i) z[i] = lds[M::BH + min(4 * i + q, 15)];
      {
        const float *wr = lds + M::WH + (4 * (n & 3) + (n >> 2)) * HS16 + 4 * q;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const f32x4 a = ld4(wr + 16 * b);
#pragma unroll
          for (int i = 0; i < 4; ++i) z = mfma16(a[i], xh[b][i], z);
        }
      }
      actor_loss_quad(z, A, q, cdead, (int)c0, c1, c2, c3, live, p.cfg, ls.scale_pi, lacc);
#pragma unroll
      for (int i = 0; i < 4; ++i) Udl[n * DLS16 + 4 * i + q] = z[i];
      wave_lds_sync();
      // raw head products Gh[a][k] += sum_s dl[a][s] xhat[k][s]; db_h = column sums of the dlogits tile
      dw_accum16<1, HEAD == 1 ? 4 : 1>(gWh, gBh, Udl, DLS16, Ut, RS16, n, q);
      // d xhat = Wh'^T dl: k-step i takes action 4 i + q
#pragma unroll
      for (int bo = 0; bo < 4; ++bo) { dx[bo][0] = 0.f; dx[bo][1] = 0.f; dx[bo][2] = 0.f; dx[bo][3] = 0.f; }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (4 * i < A) {
          const float *row = lds + M::WH + (4 * i + q) * HS16 + n;
          float a[4];
#pragma unroll
          for (int bo = 0; bo < 4; ++bo) a[bo] = row[16 * bo];
#pragma unroll
          for (int bo = 0; bo < 4; ++bo) dx[bo] = mfma16(a[bo], z[i], dx[bo]);
        }
      }
      wave_lds_sync();
    }
    STAMP(6);   // head + loss (+ head products)
    // ---- backward ----
    if constexpr (LN > 0) {
      ln_act_bwd16<RELU>(dx, xh, mean2, rstd2, pos2);          // dx = dz2
#pragma unroll
      for (int b = 0; b < 4; ++b) st4(Ut + n * RS16 + 16 * b + 4 * q, dx[b]);
      wave_lds_sync();
      STAMP(7);   // LN2 backward + tile write
      dw_accum16<4, 4>(gW2, gB2, Ut, RS16, Uh, RS16, n, q);
      STAMP(8);   // dW2
      f32x4 d1[4];
      hidden_bwd16(d1, dx, lds + M::W2T, n, q);
#pragma unroll
      for (int b = 0; b < 4; ++b) { xh[b] = ld4(Uh + n * RS16 + 16 * b + 4 * q); dx[b] = d1[b]; }
      wave_lds_sync();
    }
    STAMP(9);   // d xhat1
    ln_act_bwd16<RELU>(dx, xh, mean1, rstd1, pos1);            // dx = dz1
    if constexpr (XL1) {
      // dz1 goes to HBM (blocked feature-major per tile) for wide_l1_bwd16_kernel (W1 / feature-norm gradients: 64 x in_dim accumulators do not
      // fit a wave); the bias gradient accumulates per lane
      float *dz1b = p.wide_ws + tile * 1024 + n;               // blocked [tile][64 features][16 samples] (mlp_wide16.h)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (live) dz1b[(16 * b + 4 * q + i) * 16] = dx[b][i];
          gB1x[b][i] += dx[b][i];
        }
      STAMP(10);
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b) st4(Ut + n * RS16 + 16 * b + 4 * q, dx[b]);
      wave_lds_sync();
      STAMP(10);  // LN1 backward + tile write
      dw_accum16<4, NBK>(gW1, gB1, Ut, RS16, Ux, xs, n, q);
      wave_lds_sync();
    }
    STAMP(11);  // dW1
  }

  STAMP(12);  // loop exit
  // ---- epilogue.  Raw consumer weights of the transform first: their global loads fly under the reduction ----
  constexpr int NJH = HEAD == 1 ? 2 : 1;
  float ewh[NJH], ew2[LN > 0 ? 8 : 1], ew1[8];
  {
    const int k = threadIdx.x & 63, part = threadIdx.x >> 6;
    if constexpr (HEAD != 3) {
#pragma unroll
      for (int j = 0; j < NJH; ++j) ewh[j] = p.params[o.wh + min(part + 8 * j, A - 1) * HID + k];
    }
    if constexpr (LN > 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) ew2[j] = p.params[o.w2[0] + (part + 8 * j) * HID + k];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) ew1[j] = (fnorm && !XL1) ? p.params[o.w1 + (part + 8 * j) * D + min(k, D - 1)] : 0.f;
  }
  float gB1x_f = 0.f;
  if constexpr (XL1) {
#pragma unroll
    for (int b = 0; b < 4; ++b) st4(Ut + n * RS16 + 16 * b + 4 * q, gB1x[b]);
    wave_lds_sync();
    gB1x_f = col_sum16(Ut, RS16, lane);                         // lane = feature
    wave_lds_sync();
  }
  __syncthreads();                                              // every wave is done with its tiles: the tile area is free
  STAMP(13);  // transform loads + first barrier (slowest wave of the workgroup)
  // loss statistics: wave sums (float: a lane holds a handful of samples), combined in double by thread 0 at the very end
  float *scr = lds + M::TILES;
  {
    const float l0 = wave_sum_f(lacc[0]), l1 = wave_sum_f(lacc[1]), l2 = wave_sum_f(lacc[2]);
    if (lane == 0) { scr[960 + wave * 4 + 0] = l0; scr[960 + wave * 4 + 1] = l1; scr[960 + wave * 4 + 2] = l2; }
  }
  // bias partials -> lane = feature: sum over the 4 lanes of a sample column, lane (n, q) keeps block bf = q (feature 16 q + n)
  float gB1f, gB2f = 0.f, gBhf;
  {
    float t1[4], t2[4];
#pragma unroll
    for (int bf = 0; bf < 4; ++bf) { t1[bf] = quad_sum16(gB1[bf]); t2[bf] = LN > 0 ? quad_sum16(gB2[bf]) : 0.f; }
    gB1f = XL1 ? gB1x_f : (q == 0 ? t1[0] : (q == 1 ? t1[1] : (q == 2 ? t1[2] : t1[3])));
    gB2f = q == 0 ? t2[0] : (q == 1 ? t2[1] : (q == 2 ? t2[2] : t2[3]));
    gBhf = HEAD == 1 ? quad_sum16(gBh[0]) : (HEAD == 2 ? wave_sum_f(gBh[0]) : gLg_f);   // actor: lane n < 16 = action | critic: sum of dv in lane 0 | trunk: d gamma
    if constexpr (HEAD == 3) gWc = gLb_f;                         // trunk: d beta rides in the critic's head-product slot
  }
  double pold[4] = {0.0, 0.0, 0.0, 0.0};
  if (HEAD != 3 && threadIdx.x == 0 && p.cfg.accumulate_partials) {
#pragma unroll
    for (int k = 0; k < 4; ++k) pold[k] = p.partials[(size_t)bid * 4 + k];      // in flight under the reduction
  }
  // ---- sum the waves' raw products (deterministic: fixed order, no atomics).  The accumulators (f32x4 per lane) go through
  // a [wave][CH][lane] buffer CH at a time, one 16-byte store each; wave w then sums accumulator 8 c + w of chunk c over the
  // 8 source waves (16-byte reads) and scatters the four sums to the flat parameter layout in R0.
  const int Pn = HEAD == 3 ? (o.gru_wih >= 0 ? o.gru_wih : o.wh) : o.total;      // trunk only: the parameters in front of the GRU / head
  constexpr int CH = M::CH;                                     // accumulators per chunk (8 | 4)
  float *cb = scr + 1024;                                       // chunk buffer: n_waves x CH x 256 floats
  const int rb = XL1 ? o.b1 : 0;                                // XL1: W1 / feature-norm gradients are not this kernel's
  float *R0 = cb + M::N_WAVES * CH * 256 - rb;                  // indexed by absolute flat offsets >= rb
  constexpr int N1 = XL1 ? 0 : 4 * NBK, N2 = LN > 0 ? 16 : 0, NH = HEAD == 1 ? 4 : 0, NACC = N1 + N2 + NH;
  auto acc_of = [&](auto idc) -> f32x4 & {
    constexpr int id = decltype(idc)::value;
    if constexpr (id < N1) return gW1[id / NBK][id % NBK];
    else if constexpr (id < N1 + N2) return gW2[(id - N1) / 4][(id - N1) % 4];
    else return gWh[0][id - N1 - N2];
  };
  static_for16<(NACC + CH - 1) / CH>([&](auto cc) {
    constexpr int c0 = decltype(cc)::value * CH;
    static_for16<CH>([&](auto jc) {
      constexpr int id = c0 + decltype(jc)::value;
      if constexpr (id < NACC) st4(cb + ((wave * CH + (id - c0)) * 64 + lane) * 4, acc_of(std::integral_constant<int, id>{}));
    });
    __syncthreads();
    static_for16<CH>([&](auto jc) {
      constexpr int id = c0 + decltype(jc)::value;
      if constexpr (id < NACC) {
        if (wave == id - c0) {
          f32x4 t[M::N_WAVES];
#pragma unroll
          for (int sw = 0; sw < M::N_WAVES; ++sw) t[sw] = ld4(cb + ((sw * CH + (id - c0)) * 64 + lane) * 4);
          f32x4 v = t[0];
#pragma unroll
          for (int sw = 1; sw < M::N_WAVES; ++sw) { v[0] += t[sw][0]; v[1] += t[sw][1]; v[2] += t[sw][2]; v[3] += t[sw][3]; }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if constexpr (id < N1) {
              const int f = 16 * (id / NBK) + 4 * q + i, kc = 16 * (id % NBK) + n;      // tile column q' NV + t
              const int tq = kc / NV, tt = kc % NV, k = tq * C + tt;
              if (tt < C && k < D) R0[o.w1 + f * D + k] = v[i];
            } else if constexpr (id < N1 + N2) {
              const int f = 16 * ((id - N1) / 4) + 4 * q + i, k = 16 * ((id - N1) % 4) + n;
              R0[o.w2[0] + f * HID + k] = v[i];
            } else {
              const int a = 4 * q + i, k = 16 * (id - N1 - N2) + n;
              if (a < A) R0[o.wh + a * HID + k] = v[i];
            }
          }
        }
      }
    });
    __syncthreads();
  });
  STAMP(18);  // accumulator chunks
  float lsum[M::N_WAVES * 4];
  // per-feature vectors: {gB1, gB2, gBh, gWc} of every wave, wave w < 4 sums component w
  {
    f32x4 sv; sv[0] = gB1f; sv[1] = gB2f; sv[2] = gBhf; sv[3] = gWc;
    st4(cb + (wave * 64 + lane) * 4, sv);
    __syncthreads();
    // the waves' loss sums into thread 0's registers HERE: behind a barrier that every instantiation executes after the
    // scr[960..] writes (the accumulator-chunk loop above has ZERO trips when the kernel owns no MFMA accumulators — wide
    // critic, layer_N = 0 — so its barriers cannot be relied on), and in front of the barrier below, after which
    // affine_epilogue16 reuses scr[0, 1024) as scratch (read any later, the eighth wave's scratch writes raced with this
    // read: the value-loss statistic of a workgroup came out short now and then — the gradients never go through this slot)
    if (threadIdx.x == 0) {
#pragma unroll
      for (int e = 0; e < M::N_WAVES * 4; ++e) lsum[e] = scr[960 + e];
    }
    if (wave < 4) {
      float v = 0.f;
#pragma unroll
      for (int sw = 0; sw < UPD16_THREADS / WAVE; ++sw) v += cb[(sw * 64 + lane) * 4 + wave];
      if (wave == 0) R0[o.b1 + lane] = v;
      if (wave == 1) { if constexpr (LN > 0) R0[o.b2[0] + lane] = v; }
      if (wave == 2) {
        if constexpr (HEAD == 3) R0[(LN > 0 ? o.ln2_w[0] : o.ln1_w) + lane] = v;
        else if (HEAD == 1 ? lane < A : lane == 0) R0[o.bh + lane] = v;
      }
      if (wave == 3) {
        if constexpr (HEAD == 2) R0[o.wh + lane] = v;
        if constexpr (HEAD == 3) R0[(LN > 0 ? o.ln2_b[0] : o.ln1_b) + lane] = v;
      }
    }
    __syncthreads();
  }
  STAMP(14);  // cross-wave reduction
  // ---- raw products -> gradients (LayerNorm affines of the consumers' inputs), all threads ----
  {
    const int gl = LN > 0 ? M::G2 : M::G1, tl = LN > 0 ? M::T2 : M::T1;
    const int ogl = LN > 0 ? o.ln2_w[0] : o.ln1_w, otl = LN > 0 ? o.ln2_b[0] : o.ln1_b;
    if constexpr (HEAD != 3) affine_epilogue16<NJH>(R0, o.wh, o.bh, A, HID, lds + gl, lds + tl, ewh, ogl, otl, scr);
    if constexpr (LN > 0) affine_epilogue16<8>(R0, o.w2[0], o.b2[0], HID, HID, lds + M::G1, lds + M::T1, ew2, o.ln1_w, o.ln1_b, scr);
    if (fnorm && !XL1) affine_epilogue16<8>(R0, o.w1, o.b1, HID, D, lds + M::FN_W, lds + M::FN_B, ew1, o.fn_w, o.fn_b, scr);
  }
  STAMP(15);  // raw -> gradient transform
  float *slab = p.slabs + (size_t)bid * p.slab_stride + p.slab_col0;
  for (int e = rb + threadIdx.x; e < Pn; e += blockDim.x) slab[e] = R0[e];
  if (HEAD != 3 && threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double v = pold[k];
      if (k < 3) for (int w = 0; w < M::N_WAVES; ++w) v += (double)lsum[w * 4 + k];
      p.partials[(size_t)bid * 4 + k] = v;
    }
  }
  STAMP(16);  // slab write + loss partials
  STAMP_FLUSH();
  // ---- rows of the other network that none of its workgroups writes (dual launch, unequal shares) ----
  if (bid >= P.zero_row0 && bid < P.zero_row1) {
    float *zs = p.slabs + (size_t)bid * p.slab_stride + P.zero_col0;
    for (int e = threadIdx.x; e < P.zero_cols; e += blockDim.x) zs[e] = 0.f;
    if (threadIdx.x < 4 && P.zero_partials && !p.cfg.accumulate_partials) P.zero_partials[(size_t)bid * 4 + threadIdx.x] = 0.0;
  }
}

template <bool RELU, int LN, int HEAD, bool WIDE>
__global__ __launch_bounds__(512, 2) void mlp_update16_kernel(Upd16Args a) {
  extern __shared__ __align__(16) float lds[];
  update16_body<RELU, LN, HEAD, WIDE>(a, lds, blockIdx.x, gridDim.x);
}

// Wide inputs (in_dim 65..512): the network from z1 on; layer 1 forward = wide_l1_fwd16_kernel, its weight gradient = wide_l1_bwd_kernel
template <bool RELU, int LN, int HEAD>
__global__ __launch_bounds__(512, 2) void mlp_update16x_kernel(Upd16Args a) {
  extern __shared__ __align__(16) float lds[];
  update16_body<RELU, LN, HEAD, true, true>(a, lds, blockIdx.x, gridDim.x);
}

// Actor AND critic in one launch: workgroups [0, nA) the actor's update, [nA, nA + nC) the critic's; the shares follow the
// networks' MFMA counts per tile, so both halves finish together.
struct Dual16Args {
  Upd16Args a, c;
  int nA, nC;
};
template <bool RELU, int LN, bool WIDE_A, bool WIDE_C>
__global__ __launch_bounds__(512, 2) void mlp_update16_dual_kernel(Dual16Args d) {
  extern __shared__ __align__(16) float lds[];
  const int bid = blockIdx.x;
  if (bid < d.nA) update16_body<RELU, LN, 1, WIDE_A>(d.a, lds, bid, d.nA);
  else update16_body<RELU, LN, 2, WIDE_C>(d.c, lds, bid - d.nA, d.nC);
}
