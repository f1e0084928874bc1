// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 issue rate (s_memtime ticks per instruction and TFLOP/s)
// with 1 and 2 waves per SIMD on every CU, independent accumulators.
//   hipcc -O3 --offload-arch=gfx950 -w -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void mfma_kernel(float* out, int iters, unsigned long long* ticks) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int NACC>
void run(int threads, float* out, unsigned long long* ticks) {
  const int iters = 2000, wgs = 256 * 4;
  mfma_kernel<NACC><<<wgs, threads>>>(out, 10, ticks);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  mfma_kernel<NACC><<<wgs, threads>>>(out, iters, ticks);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  unsigned long long h;
  hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
  const double n_per_wave = (double)iters * 8 * NACC;
  const double flops = n_per_wave * 4096.0 * (threads / 64) * wgs;
  printf("accumulators/wave=%d waves/WG=%d: %.1f ticks per MFMA per wave, %.1f TFLOP/s chip-wide, kernel %.2f ms\n", NACC,
         threads / 64, (double)h / n_per_wave, flops / (ms * 1e-3) / 1e12, ms);
}

int main() {
  float* out;
  unsigned long long* ticks;
  hipMalloc(&out, 4);
  hipMalloc(&ticks, 8);
  run<1>(256, out, ticks);
  run<2>(256, out, ticks);
  run<4>(256, out, ticks);
  run<4>(512, out, ticks);
  run<1>(64, out, ticks);
  return 0;
}
