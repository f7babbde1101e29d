// mlp_wide16_args.h — argument structs and the workspace layout of the wide-input kernels (mlp_wide16.h), for the host code of
// every translation unit; the kernels themselves are only compiled in mlp_wide.hip.
#pragma once

struct Wide16Args {
  const float *params, *x;
  const int32_t *rows;
  float *z1;                 // [B][64]
  float *mean0, *rstd0;      // [B] each (may be NULL: rollout forward)
  int64_t B;
  int D, w1, b1, fn_w, fn_b; // offsets into params (fn_* < 0: no feature norm)
  // streamed rollout forward only (wide16_layer1): x_M > 0 = strided source rows, sample i = (n, m) = (i / x_M, i % x_M) starts at
  // x[n * x_sn + m * x_sm] — the env's output read in place (mappo_rollout_step); x_M == 0: contiguous rows x[i * D]
  int x_M;
  int64_t x_sn, x_sm;
  // wide_rollout_full_kernel only: the rows this network reads are also written, as they are, to copy_dst [B][D] (the rollout insert's
  // obs / share_obs slot copy, riding on the forward's loads) — or NULL
  float *copy_dst;
};


// workspace offsets (layout: see wide_l1_bwd16_kernel in mlp_wide16.h)
__host__ __device__ __forceinline__ int64_t wide16_bp(int64_t B) { return (B + 15) & ~(int64_t)15; }
__host__ __device__ __forceinline__ int64_t wide16_z1_offset(int64_t B) { return 66 * wide16_bp(B); }

struct WideBwd16Args {
  const float *params, *x;
  const int32_t *rows;
  const float *wide_ws;      // see above
  float *slabs;
  int64_t slab_stride, slab_col0, B;
  int D, w1, fn_w, fn_b;     // fn_* < 0: no feature norm
  int nca, groups;           // chunk owners per tile group (power of two >= number of chunks), tile groups (nca * groups <= 8)
};

