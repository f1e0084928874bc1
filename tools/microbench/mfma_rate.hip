// Microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 on gfx950 against the number of independent accumulator chains,
// waves per SIMD and interleaved LDS reads.  Prints s_memtime ticks per MFMA and the wall-clock TFLOP/s of the whole chip.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS, int LDS>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* ticks, int iters) {
  __shared__ float sm[64 * 68];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 64 * 68; i += blockDim.x) sm[i] = (float)i * 1e-6f;
  __syncthreads();
  f32x16 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float a = lane * 0.001f, b = lane * 0.002f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / CHAINS; ++u) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if (LDS) { a = sm[((it + u) & 31) * 68 + (lane & 31) + c]; b = sm[(32 + ((it + u) & 31)) * 68 + (lane & 31) + c]; }
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int CHAINS, int LDS>
void run(const char* name, int threads, int blocks) {
  float* out; unsigned long long* ticks;
  hipMalloc(&out, sizeof(float) * threads * blocks);
  hipMalloc(&ticks, 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CHAINS, LDS>), dim3(blocks), dim3(threads), 0, 0, out, ticks, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<CHAINS, LDS>), dim3(blocks), dim3(threads), 0, 0, out, ticks, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  const double mf = (double)iters * 16;                       // MFMAs per wave
  const double flops = mf * 4096.0 * (threads / 64) * blocks;
  printf("%-44s chains=%d lds=%d threads=%4d blocks=%4d: %7.1f ticks/MFMA(wave)  %8.3f ms  %7.1f TF/s  (%5.1f %% of 157.3)  => %.0f ticks/us\n",
         name, CHAINS, LDS, threads, blocks, (double)t / mf, ms, flops / ms / 1e9, 100 * flops / ms / 1e9 / 157.3, (double)t / (ms * 1e3));
  hipFree(out); hipFree(ticks);
}

int main() {
  run<1, 0>("1 wave/SIMD, 1 chain", 256, 256);
  run<2, 0>("1 wave/SIMD, 2 chains", 256, 256);
  run<4, 0>("1 wave/SIMD, 4 chains", 256, 256);
  run<2, 0>("2 waves/SIMD, 2 chains", 512, 256);
  run<4, 0>("2 waves/SIMD, 4 chains", 512, 256);
  run<2, 0>("4 waves/SIMD (2 WG/CU), 2 chains", 512, 512);
  run<2, 1>("1 wave/SIMD, 2 chains + 2 ds_read_b32 each", 256, 256);
  run<4, 1>("1 wave/SIMD, 4 chains + 2 ds_read_b32 each", 256, 256);
  run<2, 1>("2 waves/SIMD, 2 chains + 2 ds_read_b32 each", 512, 256);
  return 0;
}
