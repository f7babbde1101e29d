// common.h — shared helpers for the gfx950 kernels of libmappo_hip.so (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/mappo_hip.h"

#define WAVE 64

void mappo_set_error(const char *fmt, ...);

#define MAPPO_REQUIRE(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      mappo_set_error(__VA_ARGS__);         \
      return MAPPO_EINVAL;                  \
    }                                       \
  } while (0)

#define MAPPO_CLEAR_STICKY() (void)hipGetLastError()

#define MAPPO_CHECK_LAUNCH(name)                                              \
  do {                                                                        \
    hipError_t e_ = hipGetLastError();                                        \
    if (e_ != hipSuccess) {                                                   \
      mappo_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return MAPPO_ELAUNCH;                                                   \
    }                                                                         \
  } while (0)

static inline hipStream_t as_stream(mappo_stream_t s) { return (hipStream_t)s; }

// ---- one-shot HIP-event bracket around the dominant kernel of an entry point (mappo_profile_arm) ----------
struct ProfSlot { hipEvent_t start, stop; };
extern ProfSlot g_prof[MAPPO_PROF_COUNT];
// Launch `kernel`; when the hook `id` is armed, its two events are attached to THIS dispatch (start = kernel
// begin, stop = kernel end: the same timestamps rocprofv3's kernel trace reports) and the hook disarms.
#define PROF_LAUNCH(id, kernel, grid, block, lds, st, ...)                                                   \
  do {                                                                                                        \
    if (g_prof[id].start && g_prof[id].stop) {                                                                \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, st, g_prof[id].start, g_prof[id].stop, 0, __VA_ARGS__); \
      g_prof[id].start = g_prof[id].stop = nullptr;                                                           \
    } else {                                                                                                  \
      hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                          \
    }                                                                                                         \
  } while (0)
#define PROF_BEGIN(id, st) do { if (g_prof[id].start) (void)hipEventRecord(g_prof[id].start, st); } while (0)
#define PROF_END(id, st)                                                     \
  do {                                                                       \
    if (g_prof[id].stop) {                                                   \
      (void)hipEventRecord(g_prof[id].stop, st);                             \
      g_prof[id].start = g_prof[id].stop = nullptr;                          \
    }                                                                        \
  } while (0)

// ---- ValueNorm statistics from the 3-float state (valuenorm.py:31-35), fp32 like torch ------------
struct VnStats {
  float mean, sd;  // sd = sqrt(var)
};
__device__ __forceinline__ VnStats vn_stats(const float *vn_state) {
  VnStats s;
  if (vn_state == nullptr) {
    s.mean = 0.f;
    s.sd = 1.f;
    return s;
  }
  float d = fmaxf(vn_state[2], 1e-5f);
  float mean = vn_state[0] / d;
  float msq = vn_state[1] / d;
  float var = fmaxf(msq - mean * mean, 1e-2f);
  s.mean = mean;
  s.sd = sqrtf(var);
  return s;
}

// ---- wave / block reductions (double), deterministic ------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
  return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
  return v;
}

// Sum NV doubles across a block of up to 1024 threads; result valid in thread 0.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *smem /*[16*NV]*/) {
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE, nw = (blockDim.x + WAVE - 1) / WAVE;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) smem[wid * NV + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double s = 0.0;
      for (int w = 0; w < nw; ++w) s += smem[w * NV + i];
      v[i] = s;
    }
  }
}
