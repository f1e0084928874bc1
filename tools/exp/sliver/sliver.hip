// Experiment (round 4): does a workgroup with a TINY footprint - 4 waves, <= 64 VGPRs, a few KB of LDS - get a CU slot
// promptly while a weight-gradient group (2 workgroups of 8 waves / 107 VGPRs / 75 KB of LDS per CU) occupies the chip?
// The kernels of the dgrad chain (8 waves, 90-250 VGPRs, 60-150 KB) do not: they wait for an EMPTY CU.
// Two dummy kernels doing the same MFMA work per workgroup: `sliver` (256 threads, small) and `fat` (512 threads, 96 KB of
// LDS, a register block of > 128 VGPRs).  tools/exp/sliver_probe.py times them alone and under a weight-gradient group.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PRIO, bool VALU>
__global__ __launch_bounds__(256) void sliver_kernel(float* __restrict__ buf, int iters) {
  if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
  extern __shared__ float sm[];
  const int tid = threadIdx.x;
  sm[tid] = (float)tid;
  __syncthreads();
  f32x4 acc[12];                                   // 48 accumulator registers: a realistic small GEMM micro-tile (<= 64 VGPRs)
#pragma unroll
  for (int j = 0; j < 12; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = sm[(tid * 7) & 255], b = 1.0001f;
  for (int i = 0; i < iters; ++i) {
    if (VALU) {                                    // the same number of instructions on the vector ALU instead of the matrix pipe
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        acc[j][0] = fmaf(a, b, acc[j][0]); acc[j][1] = fmaf(a, b, acc[j][1]);
        acc[j][2] = fmaf(a, b, acc[j][2]); acc[j][3] = fmaf(a, b, acc[j][3]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 12; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    }
    a += 1e-6f;
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 12; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  buf[(size_t)blockIdx.x * 256 + tid] = s;
}

__global__ __launch_bounds__(512, 2) void fat_kernel(float* __restrict__ buf, int iters) {
  extern __shared__ float sm[];
  const int tid = threadIdx.x;
  sm[tid] = (float)tid;
  __syncthreads();
  f32x4 acc[36];                                   // 144 accumulator registers: > 128 VGPRs per wave
#pragma unroll
  for (int j = 0; j < 36; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = sm[(tid * 7) & 511], b = 1.0001f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 36; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    a += 1e-6f;
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 36; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  buf[(size_t)blockIdx.x * 512 + tid] = s;
}

extern "C" int sliver_launch(float* buf, int nwg, int iters, int lds_bytes, void* stream) {
  hipLaunchKernelGGL((sliver_kernel<0, false>), dim3(nwg), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, buf, iters);
  return (int)hipGetLastError();
}
extern "C" int sliver_prio_launch(float* buf, int nwg, int iters, int lds_bytes, void* stream) {
  hipLaunchKernelGGL((sliver_kernel<3, false>), dim3(nwg), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, buf, iters);
  return (int)hipGetLastError();
}
extern "C" int sliver_valu_launch(float* buf, int nwg, int iters, int lds_bytes, void* stream) {
  hipLaunchKernelGGL((sliver_kernel<0, true>), dim3(nwg), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, buf, iters);
  return (int)hipGetLastError();
}
extern "C" int sliver_valu_prio_launch(float* buf, int nwg, int iters, int lds_bytes, void* stream) {
  hipLaunchKernelGGL((sliver_kernel<3, true>), dim3(nwg), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, buf, iters);
  return (int)hipGetLastError();
}
extern "C" int fat_launch(float* buf, int nwg, int iters, int lds_bytes, void* stream) {
  static bool set = false;
  if (!set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
  hipLaunchKernelGGL(fat_kernel, dim3(nwg), dim3(512), (size_t)lds_bytes, (hipStream_t)stream, buf, iters);
  return (int)hipGetLastError();
}
