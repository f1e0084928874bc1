// 16 x 16 x 4 fp32 MFMA jobs of the crystal-aligned attention tiles (ffn.hip: the attention half inside the feed-forward
// launches; attention_aligned.hip: the stand-alone kernels).
#pragma once
#include "common.h"

// One 16 x 16 job of the attention tiles' small products on v_mfma_f32_16x16x4_f32, n <= MAXS steps of 16 along k: ALL fragments of
// the job are requested before the first MFMA (the trip counts are run-time values - a rolled loop would expose one LDS round trip
// per step: 500-660 clk per step measured against 4 x 32 of MFMA issue).  Ap / Bp: this lane's fragment base.
//   mma_kk: both operands k-contiguous (one ds_read_b128 per step each):   A[l15][k], B[l15][k]
//   mma_kn: B stored [k][n] (four ds_read_b32 per step):                    A[l15][k], B[k][l15]
template <int MAXS>
__device__ __forceinline__ f32x4 mma_kk(const float* __restrict__ Ap, const float* __restrict__ Bp, const int n,
                                        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}) {
  float4 av[MAXS], bv[MAXS];
#pragma unroll
  for (int s = 0; s < MAXS; ++s)
    if (s < n) { av[s] = ld4(Ap + 16 * s); bv[s] = ld4(Bp + 16 * s); }
#pragma unroll
  for (int s = 0; s < MAXS; ++s)
    if (s < n) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s].x, bv[s].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s].y, bv[s].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s].z, bv[s].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s].w, bv[s].w, acc, 0, 0, 0);
    }
  return acc;
}
template <int MAXS>
__device__ __forceinline__ f32x4 mma_kn(const float* __restrict__ Ap, const float* __restrict__ Bp, const int ldb, const int n) {
  float4 av[MAXS];
  float bv[MAXS][4];
#pragma unroll
  for (int s = 0; s < MAXS; ++s)
    if (s < n) {
      av[s] = ld4(Ap + 16 * s);
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[s][j] = Bp[(16 * s + j) * ldb];
    }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < MAXS; ++s)
    if (s < n) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s].x, bv[s][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s].y, bv[s][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s].z, bv[s][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s].w, bv[s][3], acc, 0, 0, 0);
    }
  return acc;
}

