// Shared device helpers for libdosx (gfx950 / CDNA4 only: wave64, fp32 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dosx.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DOSX_WAVE 64

void dosx_set_error(const char* fmt, ...);

#define DOSX_CHECK_ARG(cond, ...)   \
  do {                              \
    if (!(cond)) {                  \
      dosx_set_error(__VA_ARGS__);  \
      return -22;                   \
    }                               \
  } while (0)

#define DOSX_LAUNCH_CHECK()                                                    \
  do {                                                                         \
    hipError_t e_ = hipGetLastError();                                         \
    if (e_ != hipSuccess) {                                                    \
      dosx_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return -5;                                                               \
    }                                                                          \
  } while (0)

static inline hipStream_t to_stream(dosx_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ int dosx_map_row(const DosxRowMap& rm, int r) {
  int t = (r / rm.d) * rm.m + (r % rm.d) * rm.c + rm.off;
  return rm.idx ? rm.idx[t] : t;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float f4get(const float4& v, int i) {
  return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}

// Row statistics of a row held as `per` float4 per lane of one wave (whole row over 64 lanes).
// Two-pass (mean, then centred second moment) like torch's LayerNorm; eps = 1e-5.
#define DOSX_LN_EPS 1e-5f
