// Micro-benchmark: rate at which a workgroup can stream the k-chunks of a row-major [N][K] fp32 matrix
// (the W operand of gemm.hip: 8 lanes x 16 B per row, rows K*4 bytes apart) versus the same bytes laid
// out chunk-contiguously ("packed": [K/32][N][32]).  All workgroups read the same matrix.
//   hipcc -O3 --offload-arch=gfx950 -w -o tilebw tilebw.hip && ./tilebw
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int NW4, int PACKED>
__global__ void tile_kernel(const float* __restrict__ w, int N, int K, int reps, float* sink) {
  const int st = threadIdx.x;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int rep = 0; rep < reps; ++rep) {
    for (int k0 = 0; k0 < K; k0 += 32) {
      float4 v[NW4];
#pragma unroll
      for (int i = 0; i < NW4; ++i) {
        const int n = (st >> 3) + 32 * i, kq = (st & 7) * 4;
        const float* p = PACKED ? w + ((size_t)(k0 / 32) * N + n) * 32 + kq : w + (size_t)n * K + k0 + kq;
        v[i] = *reinterpret_cast<const float4*>(p);
      }
#pragma unroll
      for (int i = 0; i < NW4; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

template <int NW4, int PACKED>
void run(const float* w, int K, int wgs, float* sink) {
  const int N = 32 * NW4, reps = 64;
  tile_kernel<NW4, PACKED><<<wgs, 256>>>(w, N, K, 2, sink);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  tile_kernel<NW4, PACKED><<<wgs, 256>>>(w, N, K, reps, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)reps * N * K * 4.0;
  printf("N=%3d K=%4d %-8s wgs=%4d : %8.1f us  %6.1f GB/s per WG  %8.1f GB/s total\n", N, K, PACKED ? "packed" : "rowmajor", wgs,
         ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes * wgs / (ms * 1e-3) / 1e9);
}

int main() {
  float *w, *sink;
  hipMalloc(&w, 64 << 20);
  hipMemset(w, 0, 64 << 20);
  hipMalloc(&sink, 4);
  for (int wgs : {16, 256}) {
    for (int K : {128, 256, 384, 512, 416, 768}) {
      run<4, 0>(w, K, wgs, sink);
      run<4, 1>(w, K, wgs, sink);
      run<8, 0>(w, K, wgs, sink);
      run<8, 1>(w, K, wgs, sink);
    }
  }
  return 0;
}
