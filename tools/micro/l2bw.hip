// Micro-benchmark: how many bytes per clock can ONE CU pull from an L2-resident buffer with
// global_load_dwordx4, as a function of waves per workgroup and independent loads in flight per lane?
// (motivates the tile shapes of gemm.hip: see DESIGN.md "per-CU load rate").
//   hipcc -O3 --offload-arch=gfx950 -o l2bw l2bw.hip && ./l2bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int NLOAD>
__global__ void stream_kernel(const float4* __restrict__ buf, size_t n4, int iters, float* sink, unsigned long long* clk) {
  const int tid = threadIdx.x, nth = blockDim.x;
  // every workgroup walks the same buffer (like the W operand of a GEMM), from a different offset
  size_t pos = ((size_t)blockIdx.x * 977 * nth + tid) % n4;
  float4 acc = make_float4(0, 0, 0, 0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    float4 v[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      v[i] = buf[pos];
      pos += nth;
      if (pos >= n4) pos -= n4;
    }
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
  if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

template <int NLOAD>
void run(const float4* buf, size_t n4, int wgs, int threads, float* sink, unsigned long long* clk, const char* what) {
  const int iters = 200;
  stream_kernel<NLOAD><<<wgs, threads>>>(buf, n4, 10, sink, clk);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  stream_kernel<NLOAD><<<wgs, threads>>>(buf, n4, iters, sink, clk);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(wgs);
  hipMemcpy(h.data(), clk, sizeof(unsigned long long) * wgs, hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto c : h) avg += (double)c;
  avg /= wgs;
  const double bytes_per_wg = (double)iters * NLOAD * threads * 16.0;
  printf("%-10s wgs=%4d threads=%4d loads/lane=%2d : %7.1f us  %8.1f GB/s total  %6.1f GB/s per WG  (%.1f B per 100MHz-tick per WG)\n",
         what, wgs, threads, NLOAD, ms * 1e3, bytes_per_wg * wgs / (ms * 1e-3) / 1e9, bytes_per_wg / (ms * 1e-3) / 1e9,
         bytes_per_wg / avg);
}

int main() {
  float* sink;
  unsigned long long* clk;
  hipMalloc(&sink, 4);
  hipMalloc(&clk, sizeof(unsigned long long) * 4096);
  for (size_t kb : {256, 16384, 1048576}) {      // L2-resident W-like, MALL-resident, HBM
    const size_t n4 = kb * 1024 / 16;
    float4* buf;
    hipMalloc(&buf, n4 * 16);
    hipMemset(buf, 0, n4 * 16);
    printf("---- buffer %zu KB\n", kb);
    for (int wgs : {1, 16, 256, 1024}) {
      for (int threads : {256, 512, 1024}) {
        run<4>(buf, n4, wgs, threads, sink, clk, "ld4x4");
        run<8>(buf, n4, wgs, threads, sink, clk, "ld4x8");
        run<16>(buf, n4, wgs, threads, sink, clk, "ld4x16");
      }
    }
    hipFree(buf);
  }
  return 0;
}
