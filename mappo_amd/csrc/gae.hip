// gae.hip — K2: compute_returns (onpolicy/utils/shared_buffer.py:168-224) as a segmented affine scan.
//
// Every (thread, agent) series is the reverse-time recurrence  g_t = d_t + c_t * g_{t+1}:
//   GAE        : d_t = [r_t + gamma*dn(v_{t+1})*m_{t+1} - dn(v_t)] * bad_{t+1},  c_t = gamma*lambda*m_{t+1}*bad_{t+1},
//                returns_t = g_t + dn(v_t),  g_T = 0                                   (:175-192, :206-220)
//   discounted : d_t = r_t*bad_{t+1} + (1-bad_{t+1})*dn(v_t),  c_t = gamma*m_{t+1}*bad_{t+1},
//                returns_t = g_t,  g_T = next_value                                    (:194-204, :221-224)
// (bad_{t+1} == 1 unless use_proper_time_limits).  Affine maps compose: (c1,d1)o(c2,d2) = (c1*c2, d1+c1*d2).
//
// Layout [T(+1)][R]: lane <-> series (64 consecutive series = one 256-B line per wave load), wavefront <->
// time segment.  A block is 64 series x S segments; each wave reduces its segment to one affine map
// (keeping c_t, d_t, dn(v_t) in registers), the maps of later segments are folded through LDS to get the
// wave's carry-in, and the wave then replays its segment from registers.  HBM traffic is the algorithmic
// 16 B (20 B with bad_masks) per agent-step, read once and written once.
#include "common.h"

#define GAE_LMAX 16   // steps of one segment held in registers (three to four loaded values per step, all requested up front)
#define GAE_SMAX 16   // segments (waves) per block

template <bool USE_GAE, bool PTL>
__global__ __launch_bounds__(64 * GAE_SMAX) void gae_scan_kernel(
    const float *__restrict__ rewards, float *__restrict__ value_preds, const float *__restrict__ next_value,
    const float *__restrict__ masks, const float *__restrict__ bad_masks, float *__restrict__ returns,
    const float *__restrict__ vn_state, int T, int R, float gamma, float gamlam, int Lseg) {
  __shared__ float sC[GAE_SMAX][WAVE], sD[GAE_SMAX][WAVE], sCarry[WAVE];
  const int lane = threadIdx.x & (WAVE - 1), seg = threadIdx.x >> 6, S = blockDim.x >> 6;
  const int r = blockIdx.x * WAVE + lane;
  const bool valid = r < R;
  const VnStats vn = vn_stats(vn_state);
  const int chunk = S * Lseg;  // steps covered per pass of the block (>= T in the common case)

  if (seg == 0) sCarry[lane] = (USE_GAE || !valid) ? 0.f : next_value[r];
  if (seg == S - 1 && valid) {
    if (USE_GAE) value_preds[(size_t)T * R + r] = next_value[r];   // shared_buffer.py:176,207
    else returns[(size_t)T * R + r] = next_value[r];               // shared_buffer.py:194,222
  }
  __syncthreads();

  const int rc = valid ? r : R - 1;                              // (lanes beyond R read the last series: nothing of theirs is stored)
  const float nv_last = next_value[rc];
  for (int base = ((T - 1) / chunk) * chunk; base >= 0; base -= chunk) {
    const int t0 = base + seg * Lseg;
    const int len = max(0, min(T, t0 + Lseg) - t0);
    // ---- every load of the segment first: unconditional, time index clamped into the episode, nothing computed from a loaded
    // value until all are out (read inside the composition loop each step was two dependent memory round trips: 2 x 25 of them in
    // series per wave at T = 400) ----
    float c[GAE_LMAX], d[GAE_LMAX], vd[GAE_LMAX], bm[PTL ? GAE_LMAX : 1];
#pragma unroll
    for (int i = 0; i < GAE_LMAX; ++i) {
      const size_t o = (size_t)min(t0 + i, T - 1) * R + rc;
      d[i] = rewards[o];
      c[i] = masks[o + R];
      vd[i] = value_preds[o];
      if (PTL) bm[i] = bad_masks[o + R];
    }
    // value of the step after the segment (t0 + len < T: from the buffer; == T: next_value — slot T of value_preds is being
    // written by this very launch)
    const float vend_raw = value_preds[(size_t)min(t0 + len, T - 1) * R + rc];
    const float vend = ((t0 + len >= T) ? nv_last : vend_raw) * vn.sd + vn.mean;
    float C = 1.f, D = 0.f;
#pragma unroll
    for (int i = GAE_LMAX - 1; i >= 0; --i) {
      const float rw = d[i], m1 = c[i], b1 = PTL ? bm[i] : 1.f;
      const float v_t = vd[i] * vn.sd + vn.mean;
      const float v_n = (i + 1 < GAE_LMAX && i + 1 < len) ? vd[(i + 1 < GAE_LMAX) ? i + 1 : i] : vend;      // (vd[i + 1] is already denormalised: the loop runs downwards)
      c[i] = 1.f; d[i] = 0.f; vd[i] = v_t;
      if (i < len && valid) {
        float ci, di;
        if (USE_GAE) {
          const float delta = rw + gamma * v_n * m1 - v_t;
          ci = gamlam * m1;
          di = delta;
          if (PTL) { ci *= b1; di *= b1; }
        } else {
          ci = gamma * m1;
          di = rw;
          if (PTL) { ci *= b1; di = rw * b1 + (1.f - b1) * v_t; }
        }
        c[i] = ci; d[i] = di;
        D = di + ci * D;   // compose step t in front of the steps after it
        C = ci * C;
      }
    }
    sC[seg][lane] = C;
    sD[seg][lane] = D;
    __syncthreads();
    // carry-in of this segment = g at its end = later segments folded onto the chunk's terminal value
    float g = sCarry[lane];
    for (int s2 = S - 1; s2 > seg; --s2) g = sD[s2][lane] + sC[s2][lane] * g;
#pragma unroll
    for (int i = GAE_LMAX - 1; i >= 0; --i) {
      if (i < len && valid) {
        g = d[i] + c[i] * g;
        returns[(size_t)(t0 + i) * R + r] = USE_GAE ? g + vd[i] : g;
      }
    }
    __syncthreads();
    if (seg == 0) sCarry[lane] = g;   // g at t = base: terminal value of the next (earlier) chunk
    __syncthreads();
  }
}

extern "C" int mappo_gae_scan(const float *rewards, float *value_preds, const float *next_value,
                              const float *masks, const float *bad_masks, float *returns,
                              const float *vn_state, int32_t T, int32_t R, float gamma, float gae_lambda,
                              int32_t use_gae, int32_t use_proper_time_limits, mappo_stream_t stream) {
  MAPPO_REQUIRE(T > 0 && R > 0, "gae_scan: T=%d R=%d", T, R);
  MAPPO_REQUIRE(rewards && value_preds && next_value && masks && returns, "gae_scan: null pointer");
  MAPPO_REQUIRE(!use_proper_time_limits || bad_masks, "gae_scan: bad_masks required with proper time limits");
  int S = (T + 7) / 8;
  if (S > GAE_SMAX) S = GAE_SMAX;
  if (S < 1) S = 1;
  int Lseg = (T + S - 1) / S;
  if (Lseg > GAE_LMAX) Lseg = GAE_LMAX;   // long episodes: the block walks several chunks of S * GAE_LMAX steps
  const float gamlam = (float)((double)gamma * (double)gae_lambda);
  dim3 grid((R + WAVE - 1) / WAVE), block(WAVE * S);
  hipStream_t st = as_stream(stream);
#define LAUNCH(G, P)                                                                                          \
  PROF_LAUNCH(MAPPO_PROF_GAE, (gae_scan_kernel<G, P>), grid, block, 0, st, rewards, value_preds, next_value, masks, \
              bad_masks, returns, vn_state, (int)T, (int)R, gamma, gamlam, Lseg)
  if (use_gae) { if (use_proper_time_limits) LAUNCH(true, true); else LAUNCH(true, false); }
  else         { if (use_proper_time_limits) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
  MAPPO_CHECK_LAUNCH("gae_scan");
  return MAPPO_OK;
}
