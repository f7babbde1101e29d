// optim.hip — K10 + K11: gradient slab reduction, global-norm clipping and Adam on ONE flat buffer.
//   clip : torch.nn.utils.clip_grad_norm_ (r_mappo.py:143-146,157-160): norm = ||g||_2 over the segment,
//          coef = min(1, max_norm / (norm + 1e-6)), the PRE-clip norm is what train_info logs.
//   Adam : torch.optim.Adam defaults + eps=opti_eps (rMAPPOPolicy.py:31-37):
//          m += (g-m)(1-b1); v = v*b2 + g*g*(1-b2); p -= (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
// The flat buffer holds `n_seg` segments (actor | critic), each a multiple of 256 floats so that a
// 256-thread block never straddles two segments.  Step counters and hyper-parameters are device resident
// (lr_decay just rewrites one float), so the whole update is capturable in a hipGraph.
#include "common.h"

#define OPT_BLOCK 256
#define OPT_MAX_SEG 4

struct SegBounds {
  int64_t b[OPT_MAX_SEG + 1];
  int n;
};

__device__ __forceinline__ int seg_of(const SegBounds &sb, int64_t i) {
  int s = 0;
#pragma unroll
  for (int k = 1; k < OPT_MAX_SEG; ++k)
    if (k < sb.n && i >= sb.b[k]) s = k;
  return s;
}

// grad[p] = sum over slabs (per-workgroup partial gradients written by the update kernels); coalesced in p.  A workgroup owns
// 128 parameters and splits the slab rows in four: quarter q (two waves) sums rows [q n / 4, (q + 1) n / 4) with 16 independent
// loads in flight per thread, the quarters meet in LDS in a fixed order (deterministic).  One thread per parameter walked the
// 256 rows as 16 dependent batches — a latency chain of 6.5 us per PPO epoch for a pass that moves 7 MB.
#define SLAB_BLOCK 128
#define SLAB_Q 4
__device__ __forceinline__ float slab_quarter_sum(const float *__restrict__ slabs, int n_slabs, int64_t stride, int64_t p, int quarter) {
  const int s_hi = (int)(((int64_t)(quarter + 1) * n_slabs) / SLAB_Q);
  int s = (int)(((int64_t)quarter * n_slabs) / SLAB_Q);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (; s + 16 <= s_hi; s += 16) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = slabs[(size_t)(s + j) * stride + p];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j & 3] += v[j];          // fixed association: deterministic
  }
  for (; s < s_hi; ++s) acc[0] += slabs[(size_t)s * stride + p];
  return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}
// the workgroup's 128 sums: valid in the threads of quarter 0 (threadIdx.x < 128), 0 elsewhere
__device__ __forceinline__ float slab_block_sum(const float *__restrict__ slabs, int n_slabs, int64_t stride, int64_t P, int64_t &p_out) {
  __shared__ float sq[SLAB_Q][SLAB_BLOCK];
  const int pi = threadIdx.x & (SLAB_BLOCK - 1), quarter = threadIdx.x / SLAB_BLOCK;
  const int64_t p = (int64_t)blockIdx.x * SLAB_BLOCK + pi;
  p_out = p;
  sq[quarter][pi] = p < P ? slab_quarter_sum(slabs, n_slabs, stride, p, quarter) : 0.f;
  __syncthreads();
  return quarter == 0 ? (sq[0][pi] + sq[1][pi]) + (sq[2][pi] + sq[3][pi]) : 0.f;
}

__global__ __launch_bounds__(SLAB_BLOCK * SLAB_Q) void slab_reduce_kernel(const float *__restrict__ slabs, int n_slabs,
                                                                         int64_t stride, int64_t P, float *__restrict__ grad) {
  int64_t p;
  const float g = slab_block_sum(slabs, n_slabs, stride, P, p);
  if (threadIdx.x < SLAB_BLOCK && p < P) grad[p] = g;
}

// ---- two-launch variant of slab_reduce + clip_adam (the per-update tail of the PPO loop is launch-bound: every
// kernel here runs a few microseconds, so the launches themselves are what it costs) --------------------------------
// (1) slab_reduce that also leaves the squared-norm partial of its 128 gradient entries;
// (2) norm finalisation + Adam in one kernel (see norm_adam_kernel).
__global__ __launch_bounds__(SLAB_BLOCK * SLAB_Q) void slab_reduce_sq_kernel(const float *__restrict__ slabs, int n_slabs, int64_t stride,
                                                                            int64_t P, float *__restrict__ grad, double *__restrict__ partials,
                                                                            const float *__restrict__ hyper, int32_t *step, int n_seg) {
  __shared__ double smem[16];
  if (blockIdx.x == 0 && threadIdx.x < n_seg && hyper[threadIdx.x * 8 + 7] != 0.f) step[threadIdx.x] += 1;   // Adam step of enabled segments
  int64_t p;
  const float g = slab_block_sum(slabs, n_slabs, stride, P, p);         // (0 in the threads of quarters 1..3)
  if (threadIdx.x < SLAB_BLOCK && p < P) grad[p] = g;
  double v[1] = {(double)g * (double)g};
  block_sum<1>(v, smem);
  if (threadIdx.x == 0) partials[blockIdx.x] = v[0];
}

// squared-norm partials per 128 gradient entries (+ the Adam step increment of the enabled segments): the first launch of
// mappo_clip_adam when the gradient was reduced elsewhere (data-parallel all-reduce, recurrent path)
__global__ __launch_bounds__(SLAB_BLOCK) void sqnorm128_kernel(const float *__restrict__ grad, int64_t P, double *__restrict__ partials,
                                                              const float *__restrict__ hyper, int32_t *step, int n_seg) {
  __shared__ double smem[16];
  if (blockIdx.x == 0 && threadIdx.x < n_seg && hyper[threadIdx.x * 8 + 7] != 0.f) step[threadIdx.x] += 1;
  const int64_t p = (int64_t)blockIdx.x * SLAB_BLOCK + threadIdx.x;
  double v[1] = {0.0};
  if (p < P) { const double g = (double)grad[p]; v[0] = g * g; }
  block_sum<1>(v, smem);
  if (threadIdx.x == 0) partials[blockIdx.x] = v[0];
}

// One 256-thread workgroup per 256 parameters (never straddles a segment).  Every workgroup finalises the norm of ITS
// segment itself (a few dozen partials + two pow()s: redundant across workgroups, but in parallel and without a
// separate launch), then applies Adam to its slice.  The step counters were advanced by slab_reduce_sq_kernel's first
// workgroup (a kernel boundary orders that increment before every read here); workgroup 0 also publishes the norms.
__global__ __launch_bounds__(OPT_BLOCK) void norm_adam_kernel(const double *__restrict__ partials, SegBounds sb,
                                                             const float *__restrict__ hyper, const int32_t *__restrict__ step,
                                                             float *__restrict__ grad_norms, double *__restrict__ norm_acc,
                                                             float *__restrict__ params, const float *__restrict__ grad,
                                                             float *__restrict__ m, float *__restrict__ v) {
  __shared__ double smem[16];
  __shared__ float ws[4];
  const int64_t i = (int64_t)blockIdx.x * OPT_BLOCK + threadIdx.x;
  const int s = seg_of(sb, (int64_t)blockIdx.x * OPT_BLOCK);
  const float gi = grad[i], pi = params[i], mi0 = m[i], vi0 = v[i];      // in flight under the norm reduction
  const int n_seg_pub = (blockIdx.x == 0) ? sb.n : 1;                   // workgroup 0 walks all segments to publish their norms
  for (int k = 0; k < n_seg_pub; ++k) {
    const int sk = (blockIdx.x == 0) ? (k == 0 ? s : (k <= s ? k - 1 : k)) : s;      // own segment first
    const int b0 = (int)(sb.b[sk] / SLAB_BLOCK), b1 = (int)(sb.b[sk + 1] / SLAB_BLOCK);
    double acc[1] = {0.0};
    for (int b = b0 + threadIdx.x; b < b1; b += blockDim.x) acc[0] += partials[b];
    block_sum<1>(acc, smem);
    if (threadIdx.x == 0) {
      const float *h = hyper + sk * 8;
      const float norm = (float)sqrt(acc[0]);
      if (blockIdx.x == 0) {
        grad_norms[sk] = norm;
        if (norm_acc) norm_acc[sk] += (double)norm;
      }
      if (k == 0) {
        float coef = 1.f;
        if (h[6] != 0.f) coef = fminf(h[5] / (norm + 1e-6f), 1.f);
        const int t = step[sk];
        const double bc1 = 1.0 - pow((double)h[1], (double)t);
        const double bc2 = 1.0 - pow((double)h[2], (double)t);
        ws[0] = coef;
        ws[1] = (float)((double)h[0] / (bc1 > 0.0 ? bc1 : 1.0));
        ws[2] = (float)sqrt(bc2 > 0.0 ? bc2 : 1.0);
        ws[3] = h[7] != 0.f ? 1.f : 0.f;
      }
    }
    __syncthreads();
  }
  if (ws[3] == 0.f) return;
  const float *h = hyper + s * 8;
  const float b1 = h[1], b2 = h[2], eps = h[3], wd = h[4];
  float g = gi * ws[0];
  float pp = pi;
  if (wd != 0.f) g = g + wd * pp;
  float mi = mi0, vi = vi0;
  mi = mi + (g - mi) * (1.f - b1);
  vi = vi * b2 + g * g * (1.f - b2);
  const float denom = sqrtf(vi) / ws[2] + eps;
  pp = pp - ws[1] * (mi / denom);
  params[i] = pp; m[i] = mi; v[i] = vi;
}

extern "C" int64_t mappo_optim_workspace_bytes(int64_t P) {
  const int64_t nblk = (P + SLAB_BLOCK - 1) / SLAB_BLOCK;       // mappo_reduce_clip_adam keeps one partial per 128 entries
  return nblk * (int64_t)sizeof(double) + OPT_MAX_SEG * 4 * (int64_t)sizeof(float) + 64;
}

extern "C" int mappo_slab_reduce(const float *slabs, int32_t n_slabs, int64_t slab_stride, int64_t P, float *grad,
                                 mappo_stream_t stream) {
  MAPPO_REQUIRE(slabs && grad && n_slabs > 0 && P > 0 && slab_stride >= P, "slab_reduce: bad arguments");
  const int nblk = (int)((P + SLAB_BLOCK - 1) / SLAB_BLOCK);
  PROF_LAUNCH(MAPPO_PROF_SLAB_REDUCE, slab_reduce_kernel, dim3(nblk), dim3(SLAB_BLOCK * SLAB_Q), 0, as_stream(stream), slabs,
              (int)n_slabs, slab_stride, P, grad);
  MAPPO_CHECK_LAUNCH("slab_reduce");
  return MAPPO_OK;
}

extern "C" int mappo_clip_adam(float *params, const float *grad, float *exp_avg, float *exp_avg_sq,
                               const int64_t *seg_bounds, int32_t n_seg, const float *opt_hyper, int32_t *opt_step,
                               float *grad_norms, double *norm_acc, void *workspace, mappo_stream_t stream) {
  MAPPO_REQUIRE(params && grad && exp_avg && exp_avg_sq && seg_bounds && opt_hyper && opt_step && grad_norms && workspace,
                "clip_adam: null pointer");
  MAPPO_REQUIRE(n_seg >= 1 && n_seg <= OPT_MAX_SEG, "clip_adam: n_seg=%d", n_seg);
  SegBounds sb;
  sb.n = n_seg;
  for (int s = 0; s <= n_seg; ++s) {
    MAPPO_REQUIRE(seg_bounds[s] % OPT_BLOCK == 0 && (s == 0 || seg_bounds[s] > seg_bounds[s - 1]),
                  "clip_adam: segment bounds must be increasing multiples of %d", OPT_BLOCK);
    sb.b[s] = seg_bounds[s];
  }
  MAPPO_REQUIRE(seg_bounds[0] == 0, "clip_adam: seg_bounds[0] must be 0");
  const int64_t P = seg_bounds[n_seg];
  double *partials = (double *)workspace;
  hipStream_t st = as_stream(stream);
  // two launches: squared-norm partials (+ step increment), then norm finalisation + Adam in one kernel
  PROF_BEGIN(MAPPO_PROF_ADAM, st);
  hipLaunchKernelGGL(sqnorm128_kernel, dim3((unsigned)(P / SLAB_BLOCK)), dim3(SLAB_BLOCK), 0, st, grad, P, partials, opt_hyper, opt_step,
                     (int)n_seg);
  hipLaunchKernelGGL(norm_adam_kernel, dim3((unsigned)(P / OPT_BLOCK)), dim3(OPT_BLOCK), 0, st, (const double *)partials, sb, opt_hyper,
                     (const int32_t *)opt_step, grad_norms, norm_acc, params, grad, exp_avg, exp_avg_sq);
  PROF_END(MAPPO_PROF_ADAM, st);
  MAPPO_CHECK_LAUNCH("clip_adam");
  return MAPPO_OK;
}

extern "C" int mappo_reduce_clip_adam(const float *slabs, int32_t n_slabs, int64_t slab_stride, float *params, float *grad,
                                      float *exp_avg, float *exp_avg_sq, const int64_t *seg_bounds, int32_t n_seg,
                                      const float *opt_hyper, int32_t *opt_step, float *grad_norms, double *norm_acc,
                                      void *workspace, mappo_stream_t stream) {
  MAPPO_REQUIRE(slabs && params && grad && exp_avg && exp_avg_sq && seg_bounds && opt_hyper && opt_step && grad_norms && workspace,
                "reduce_clip_adam: null pointer");
  MAPPO_REQUIRE(n_seg >= 1 && n_seg <= OPT_MAX_SEG && n_slabs > 0, "reduce_clip_adam: n_seg=%d n_slabs=%d", n_seg, n_slabs);
  SegBounds sb;
  sb.n = n_seg;
  for (int s = 0; s <= n_seg; ++s) {
    MAPPO_REQUIRE(seg_bounds[s] % OPT_BLOCK == 0 && (s == 0 || seg_bounds[s] > seg_bounds[s - 1]),
                  "reduce_clip_adam: segment bounds must be increasing multiples of %d", OPT_BLOCK);
    sb.b[s] = seg_bounds[s];
  }
  MAPPO_REQUIRE(seg_bounds[0] == 0, "reduce_clip_adam: seg_bounds[0] must be 0");
  const int64_t P = seg_bounds[n_seg];
  MAPPO_REQUIRE(slab_stride >= P, "reduce_clip_adam: slab_stride < P");
  const int nblk = (int)(P / SLAB_BLOCK);             // P is a multiple of 256: mappo_optim_workspace_bytes(P) covers P/128 doubles
  double *partials = (double *)workspace;
  hipStream_t st = as_stream(stream);
  PROF_LAUNCH(MAPPO_PROF_SLAB_REDUCE, slab_reduce_sq_kernel, dim3(nblk), dim3(SLAB_BLOCK * SLAB_Q), 0, st, slabs, (int)n_slabs, slab_stride, P,
              grad, partials, opt_hyper, opt_step, (int)n_seg);
  PROF_LAUNCH(MAPPO_PROF_ADAM, norm_adam_kernel, dim3((unsigned)(P / OPT_BLOCK)), dim3(OPT_BLOCK), 0, st, (const double *)partials, sb,
              opt_hyper, (const int32_t *)opt_step, grad_norms, norm_acc, params, (const float *)grad, exp_avg, exp_avg_sq);
  MAPPO_CHECK_LAUNCH("reduce_clip_adam");
  return MAPPO_OK;
}
