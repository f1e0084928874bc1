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

// ---- a vector-ALU GEMM with a sliver footprint: C[M,N] = A[M,K] . W[K,N] (+ R[M,N]) -------------------------------------------
// 256 threads = 16 x 16, a 64 x 64 output tile, 4 x 4 per thread, k-chunks of 16 through 8 KB of LDS (A chunk stored k-major),
// the next chunk prefetched into registers; packed fp32 FMAs (the compiler emits v_pk_fma_f32 for the float2 arithmetic).
// <= 64 VGPRs, so that one such workgroup fits next to two resident weight-gradient workgroups on a CU and computes on the
// vector ALU while their matrix waves own the MFMA pipe.
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int SG_T = 64, SG_K = 16;
__global__ __launch_bounds__(256) void sliver_gemm_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                          const float* __restrict__ R, int ldr, float* __restrict__ C, int ldc,
                                                          int M, int N, int K, int prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);
  __shared__ __align__(16) float As[SG_K][SG_T + 4];
  __shared__ __align__(16) float Bs[SG_K][SG_T + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.x * SG_T, n0 = blockIdx.y * SG_T;
  // staging: A chunk = 64 rows x 16 k: thread -> (row tid / 4, k (tid % 4) * 4 ..+3); W chunk = 16 k x 64 n: thread -> (k tid / 16, n (tid % 16) * 4)
  const int ar = tid >> 2, ak = (tid & 3) * 4;
  const int wk = tid >> 4, wn = (tid & 15) * 4;
  const float* ap = A + (size_t)min(m0 + ar, M - 1) * lda + ak;
  const float* wp = W + (size_t)wk * ldw + min(n0 + wn, N - 4);
  float4 ra = *reinterpret_cast<const float4*>(ap), rw = *reinterpret_cast<const float4*>(wp);
  f32x2 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { acc[i][0] = f32x2{0.f, 0.f}; acc[i][1] = f32x2{0.f, 0.f}; }
  for (int k0 = 0; k0 < K; k0 += SG_K) {
    __syncthreads();
    As[ak + 0][ar] = ra.x; As[ak + 1][ar] = ra.y; As[ak + 2][ar] = ra.z; As[ak + 3][ar] = ra.w;
    *reinterpret_cast<float4*>(&Bs[wk][wn]) = rw;
    __syncthreads();
    if (k0 + SG_K < K) {
      ra = *reinterpret_cast<const float4*>(ap + k0 + SG_K);
      rw = *reinterpret_cast<const float4*>(wp + (size_t)(k0 + SG_K) * ldw);
    }
#pragma unroll
    for (int k = 0; k < SG_K; ++k) {
      const float4 a = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
      const float4 b = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
      const f32x2 b01 = {b.x, b.y}, b23 = {b.z, b.w};
      acc[0][0] += f32x2{a.x, a.x} * b01; acc[0][1] += f32x2{a.x, a.x} * b23;
      acc[1][0] += f32x2{a.y, a.y} * b01; acc[1][1] += f32x2{a.y, a.y} * b23;
      acc[2][0] += f32x2{a.z, a.z} * b01; acc[2][1] += f32x2{a.z, a.z} * b23;
      acc[3][0] += f32x2{a.w, a.w} * b01; acc[3][1] += f32x2{a.w, a.w} * b23;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = m0 + ty * 4 + i, c = n0 + tx * 4;
    if (r < M && c < N) {
      float4 o = make_float4(acc[i][0][0], acc[i][0][1], acc[i][1][0], acc[i][1][1]);
      if (R) { const float4 q = *reinterpret_cast<const float4*>(R + (size_t)r * ldr + c); o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w; }
      *reinterpret_cast<float4*>(C + (size_t)r * ldc + c) = o;
    }
  }
}
extern "C" int sliver_gemm_launch(const float* A, int lda, const float* W, int ldw, const float* R, int ldr, float* C, int ldc,
                                  int M, int N, int K, int prio, void* stream) {
  if ((K % SG_K) || (N & 3) || N < 4) return -22;
  hipLaunchKernelGGL(sliver_gemm_kernel, dim3((M + SG_T - 1) / SG_T, (N + SG_T - 1) / SG_T), dim3(256), 0, (hipStream_t)stream,
                     A, lda, W, ldw, R, ldr, C, ldc, M, N, K, prio);
  return (int)hipGetLastError();
}
