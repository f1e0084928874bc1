// Linear -> LayerNorm -> PReLU -> Linear (+ residual) in ONE launch for small row counts: the NodeModel MLP of a
// message-passing layer (DOSTransformer_phonon.py:200-212 / DOSTransformer.py:178-190: node_mlp_2 on cat[x, agg]).
//
//     z    = [a0 | a1] . W1^T + b1                 [M,NH]
//     xhat = LN_noaffine(z), rstd                  (saved: the backward and the weight gradients read them)
//     out  = prelu(xhat*gamma + beta) . W2^T + b2 (+ res)          [M,NO]
//
// A batch has a few hundred nodes (cfg2: 450), so each of the two dosx_gemm launches this replaces was one partial
// round of 16-row workgroups whose duration is the fixed part of a launch (~9.7 us each + the gap between them); here a
// workgroup keeps its 16 x NH intermediate tile in LDS between the two products and all 8 waves multiply (16x16x4 MFMA),
// each pulling the weight fragments of its own columns straight from L2 (see mlp_ln_fwd_kernel).
// The backward kernel is the mirror image: da = dy . W2 -> PReLU / LayerNorm backward on the LDS tile (dz written out for
// the W1 weight gradient, dgamma | dbeta | dalpha partial sums per workgroup) -> dcat = dz . W1.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int MR = 16;           // rows per workgroup
constexpr int NCG = 2;           // float4 column groups per lane in the row phases (NH <= 512)
constexpr int WQ = 64;           // k extent of one weight chunk
constexpr int WLD_NK = WQ + 4;   // padded row of an [n][k] chunk in LDS
constexpr int WLD_KN = 32 + 4;   // padded row of a [k][n] chunk in LDS
constexpr int WP_FLOATS = 32 * WLD_NK > WQ * WLD_KN ? 32 * WLD_NK : WQ * WLD_KN;     // one wave's private chunk buffer

// At 16 rows per workgroup a weight element is used by exactly ONE wave (the one that owns its output column), so the
// weights need no workgroup-wide staging: every wave streams the chunks of its own NC columns global -> registers
// (coalesced: whole 256-byte / 128-byte row pieces) -> its PRIVATE LDS buffer -> MFMA fragments.  LDS operations of one
// wave execute in order, so the hand-over needs no s_barrier, only a compiler fence (wave_barrier); loads run one chunk
// ahead in registers.  All 8 waves multiply; the only workgroup barriers are the three phase boundaries.
//   version 1 (ffn.hip's recipe: 4 staging + 4 matrix waves, one barrier per chunk): 14 us forward at M = 424;
//   version 2 (B fragments straight from global memory, 16 cache lines per load instruction): 17 us;
//   the two dosx_gemm launches this kernel replaces: 9.7 + 9.6 us + the gap between them.

// Both streams keep D chunks in flight in registers (a ring: the slot of chunk q takes chunk q + D as soon as its data
// has been handed to the LDS) and separate the FIRST D requests from the loop, so that a kernel can put them in front of
// whatever precedes the product - round 5: with one chunk ahead every chunk paid most of an L2 round trip (~2 k clk x 8
// chunks = most of the kernel's 14 us at M = 424); the weights depend on nothing, so at hidden 128 ALL of a wave's W1
// chunks are requested before the input tile is and all of its W2 chunks before the LayerNorm row phase.

// acc[t] += As[16][kdim] . W[col_base + 16 t + (0..15)][0..kdim)^T,  W row-major [n][ldw]  (nn.Linear weight)
template <int NC, int D, bool STATIC>
struct NkStream {
  static constexpr int NR = NC / 4;                // float4 per lane per chunk (4 rows of 16 lanes per instruction)
  float4 r[D][NR];
  const float* src;
  int ldw;
  __device__ __forceinline__ void issue(float4 (&x)[NR], int q) {
#pragma unroll
    for (int i = 0; i < NR; ++i) x[i] = ld4(src + (size_t)(4 * i) * ldw + q * WQ);
    __builtin_amdgcn_sched_barrier(0);             // (chunks are consumed in request order: keep the scheduler from mixing them)
  }
  __device__ __forceinline__ void prefetch(const float* w, int ldw_, int col_base, int kdim, int lane) {
    ldw = ldw_;
    src = w + (size_t)(col_base + (lane >> 4)) * ldw + (lane & 15) * 4;
    const int nq = kdim / WQ;
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (d < nq) issue(r[d], d);
  }
  __device__ __forceinline__ void run(f32x4 (&acc)[NC / 16], int kdim, const float* As, int lda, float* Wp, int lane) {
    const int l15 = lane & 15, g4 = lane >> 4;
    const int nq = kdim / WQ;
    auto round = [&](int q0) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int q = q0 + d;
        if (q >= nq) break;
#pragma unroll
        for (int i = 0; i < NR; ++i) st4(Wp + (g4 + 4 * i) * WLD_NK + l15 * 4, r[d][i]);
        if (q + D < nq) issue(r[d], q + D);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kk = 0; kk < WQ; kk += 16) {
          const float4 av = ld4(As + l15 * lda + q * WQ + kk + 4 * g4);
#pragma unroll
          for (int t = 0; t < NC / 16; ++t) {
            const float4 b = ld4(Wp + (16 * t + l15) * WLD_NK + kk + 4 * g4);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b.x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b.y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b.z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b.w, acc[t], 0, 0, 0);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    };
    if constexpr (STATIC) {
#pragma unroll
      for (int q0 = 0; q0 < nq; q0 += D) round(q0);
    } else {
#pragma unroll 1
      for (int q0 = 0; q0 < nq; q0 += D) round(q0);
    }
  }
};

// acc[t] += As[16][kdim] . W[0..kdim)[col_base + 16 t + (0..15)],  W row-major [k][ldw]  (a stored weight read as its
// transpose: the dgrad products).  32 columns per wave.
template <int D, bool STATIC>
struct KnStream {
  static constexpr int NR = 8;                     // 8 k rows of 8 lanes per instruction, 64 rows per chunk
  float4 r[D][NR];
  const float* src;
  int ldw;
  __device__ __forceinline__ void issue(float4 (&x)[NR], int q) {
#pragma unroll
    for (int i = 0; i < NR; ++i) x[i] = ld4(src + (size_t)(q * WQ + 8 * i) * ldw);
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void prefetch(const float* w, int ldw_, int col_base, int kdim, int lane) {
    ldw = ldw_;
    src = w + (size_t)(lane >> 3) * ldw + col_base + (lane & 7) * 4;
    const int nq = kdim / WQ;
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (d < nq) issue(r[d], d);
  }
  __device__ __forceinline__ void run(f32x4 (&acc)[2], int kdim, const float* As, int lda, float* Wp, int lane) {
    const int l15 = lane & 15, g4 = lane >> 4, kr = lane >> 3, n4 = (lane & 7) * 4;
    const int nq = kdim / WQ;
    auto round = [&](int q0) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int q = q0 + d;
        if (q >= nq) break;
#pragma unroll
        for (int i = 0; i < NR; ++i) st4(Wp + (kr + 8 * i) * WLD_KN + n4, r[d][i]);
        if (q + D < nq) issue(r[d], q + D);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kk = 0; kk < WQ; kk += 16) {
          const float4 av = ld4(As + l15 * lda + q * WQ + kk + 4 * g4);
          const float* bp = Wp + (kk + 4 * g4) * WLD_KN + l15;
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bp[0], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bp[16], acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bp[WLD_KN], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bp[WLD_KN + 16], acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bp[2 * WLD_KN], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bp[2 * WLD_KN + 16], acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bp[3 * WLD_KN], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bp[3 * WLD_KN + 16], acc[1], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
      }
    };
    if constexpr (STATIC) {
#pragma unroll
      for (int q0 = 0; q0 < nq; q0 += D) round(q0);
    } else {
#pragma unroll 1
      for (int q0 = 0; q0 < nq; q0 += D) round(q0);
    }
  }
};

// HC = hidden / 64 when (K, NH, NO) = (2, 2, 1) x hidden - the NodeModel - so that every chunk loop unrolls; 0: run-time shapes
template <int HC>
__global__ __launch_bounds__(512) void mlp_ln_fwd_kernel(const DosxMlpLn a) {
  DOSX_SET_MAIN_PRIO();
  extern __shared__ __align__(16) float sm[];
  constexpr int D = HC == 4 ? 3 : HC ? 4 : 2;
  const int K = HC ? 128 * HC : a.K, NH = HC ? 128 * HC : a.NH, NO = HC ? 64 * HC : a.NO, M = a.M;
  const int LDX = K + 4, LDT = NH + 4, LDC = NO + 4;
  float* Xs = sm;                                  // [16][LDX]  input tile (A operand of the first product); later the C tile
  float* T = Xs + MR * LDX;                        // [16][LDT]  z -> prelu(LN(z)) (A operand of the second product)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  float* Wp = T + MR * LDT + wave * WP_FLOATS;     // this wave's private weight-chunk buffer
  const int m0 = blockIdx.x * MR;
  NkStream<32, D, HC != 0> s1;
  NkStream<16, D, HC != 0> s2;
  // [a0 | a1] tile (row tid/32, 4-float groups tid%32 + 32 j): requested FIRST - loads return in order, and the first
  // product waits for these only, with the weight chunks behind them still in flight
  float4 xin[4];
  {
    const int rr = min(m0 + (tid >> 5), M - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = (tid & 31) * 4 + 128 * j;
      if (c < K) xin[j] = ld4(c < a.k0 ? a.a0 + (size_t)rr * a.lda0 + c : a.a1 + (size_t)rr * a.lda1 + (c - a.k0));
    }
  }
  s1.prefetch(a.w1, K, wave * 32 < NH ? wave * 32 : 0, K, lane);          // (the first column block's chunks, before anything else)

  // epilogue operands of this wave's 2 rows, fetched at kernel start
  const int c0 = lane * 4;
  const bool con = c0 < NO;
  float4 rres[2], bias2 = f4zero();
  if (con) bias2 = ld4(a.b2 + c0);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = min(m0 + wave * 2 + i, M - 1);
    // (an unconditional load - of the bias when there is no residual - so that no branch joins in front of the products
    //  and waits for everything requested so far)
    rres[i] = ld4(a.res ? a.res + (size_t)r * a.ldres + (con ? c0 : 0) : a.b2 + (con ? c0 : 0));
  }
  const bool has_res = a.res != nullptr;
  // operands of the row phase (rows 2*wave, 2*wave+1; columns lane*4 + 256 j)
  float4 gam[NCG], bet[NCG];
  bool on[NCG];
#pragma unroll
  for (int j = 0; j < NCG; ++j) {
    const int c = lane * 4 + 256 * j;
    on[j] = c < NH;
    gam[j] = ld4(a.gamma + (on[j] ? c : 0));
    bet[j] = ld4(a.beta + (on[j] ? c : 0));
  }
  const float alpha = *a.alpha;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = (tid & 31) * 4 + 128 * j;
    if (c < K) st4(Xs + (tid >> 5) * LDX + c, xin[j]);
  }
  __syncthreads();                                         // Xs visible
  // ---- first product: 256-column blocks (this wave: columns wave*32 .. +31 of the block = two 16-column tiles) ----
  for (int blk = 0; blk * 256 < NH; ++blk) {
    const int cb = blk * 256 + wave * 32;
    const bool cok = cb < NH;                              // (wave-uniform: NH is a multiple of 128)
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const float b1a = a.b1[(cok ? cb : 0) + l15], b1b = a.b1[(cok ? cb : 0) + l15 + 16];
    if (blk) s1.prefetch(a.w1, K, cok ? cb : 0, K, lane);
    s1.run(acc, K, Xs, LDX, Wp, lane);
    if ((blk + 1) * 256 >= NH) s2.prefetch(a.w2, NH, wave * 16 < NO ? wave * 16 : 0, NH, lane);   // in flight under the row phase
    if (cok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        T[(4 * g4 + r) * LDT + cb + l15] = acc[0][r] + b1a;
        T[(4 * g4 + r) * LDT + cb + l15 + 16] = acc[1][r] + b1b;
      }
    }
  }
  __syncthreads();                                         // z tile complete
  // ---- row phase: LayerNorm statistics (two-pass, like torch) -> xhat, rstd out; prelu(xhat*gamma+beta) -> T ----
  {
    const float invN = 1.f / (float)NH;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lr = wave * 2 + i, r = m0 + lr;
      float4 z[NCG];
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NCG; ++j) {
        z[j] = f4zero();
        if (on[j]) {
          z[j] = ld4(T + lr * LDT + lane * 4 + 256 * j);
          s += (z[j].x + z[j].y) + (z[j].z + z[j].w);
        }
      }
      const float mean = wave_sum(s) * invN;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < NCG; ++j) {
        if (on[j]) {
          z[j] = make_float4(z[j].x - mean, z[j].y - mean, z[j].z - mean, z[j].w - mean);
          q += (z[j].x * z[j].x + z[j].y * z[j].y) + (z[j].z * z[j].z + z[j].w * z[j].w);
        }
      }
      const float rstd = rsqrtf(wave_sum(q) * invN + DOSX_LN_EPS);
#pragma unroll
      for (int j = 0; j < NCG; ++j) {
        if (!on[j]) continue;
        const float4 xh = make_float4(z[j].x * rstd, z[j].y * rstd, z[j].z * rstd, z[j].w * rstd);
        if (r < M) st4(a.xhat + (size_t)r * NH + lane * 4 + 256 * j, xh);
        float4 y = make_float4(xh.x * gam[j].x + bet[j].x, xh.y * gam[j].y + bet[j].y, xh.z * gam[j].z + bet[j].z,
                               xh.w * gam[j].w + bet[j].w);
        y.x = y.x >= 0.f ? y.x : alpha * y.x; y.y = y.y >= 0.f ? y.y : alpha * y.y;
        y.z = y.z >= 0.f ? y.z : alpha * y.z; y.w = y.w >= 0.f ? y.w : alpha * y.w;
        st4(T + lr * LDT + lane * 4 + 256 * j, y);
      }
      if (lane == 0 && r < M) a.rstd[r] = rstd;
    }
  }
  __syncthreads();                                         // activated tile complete
  // ---- second product: 128-column blocks (this wave: one 16-column tile), A operand = T; C tile -> Xs ----
  float* Cs = Xs;
  for (int blk = 0; blk * 128 < NO; ++blk) {
    const int oc = blk * 128 + wave * 16;
    const bool ook = oc < NO;
    f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
    if (blk) s2.prefetch(a.w2, NH, ook ? oc : 0, NH, lane);
    s2.run(acc, NH, T, LDT, Wp, lane);
    if (ook) {
#pragma unroll
      for (int r = 0; r < 4; ++r) Cs[(4 * g4 + r) * LDC + oc + l15] = acc[0][r];
    }
  }
  // (third product: its first weight chunks are requested here, behind the second product's last ones)
  NkStream<32, D, HC != 0> s3;
  const bool third = a.w3 != nullptr;
  if (third) s3.prefetch(a.w3, a.ldw3, wave * 32, NO, lane);
  __syncthreads();
  // ---- row epilogue (8 waves x 2 rows): out = C + b2 (+ res) ----
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lr = wave * 2 + i, r = m0 + lr;
    if (con) {
      const float4 v = ld4(Cs + lr * LDC + c0);
      const float4 rr = has_res ? rres[i] : f4zero();
      const float4 o = make_float4(v.x + bias2.x + rr.x, v.y + bias2.y + rr.y, v.z + bias2.z + rr.z, v.w + bias2.w + rr.w);
      if (r < M) st4(a.out + (size_t)r * a.ldo + c0, o);
      if (third) st4(Cs + lr * LDC + c0, o);         // the finished rows: A operand of the third product
    }
  }
  if (!third) return;
  __syncthreads();
  // ---- third product (DosxMlpLn.w3): pq[:, b n3 + n] = out . w3[n, b NO + k]; 256-column blocks, two 16-column tiles per wave,
  //      k over NO; C straight to HBM ----
  for (int b = 0; b < a.nb3; ++b)
    for (int blk = 0; blk * 256 < a.n3; ++blk) {
      const int cb = blk * 256 + wave * 32;
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      if (b | blk) s3.prefetch(a.w3 + b * NO, a.ldw3, cb, NO, lane);
      s3.run(acc, NO, Cs, LDC, Wp, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + 4 * g4 + r;
        if (row < M) {
          a.pq[(size_t)row * a.ldpq + b * a.n3 + cb + l15] = acc[0][r];
          a.pq[(size_t)row * a.ldpq + b * a.n3 + cb + l15 + 16] = acc[1][r];
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Backward:  da = dy . W2 ;  dy' = da o prelu'(xhat*gamma+beta) ;  dz = LN_bwd(dy' * gamma)  (written out) ;
//            dcat = dz . W1 ;  partials[wg] = [ sum dy'*xhat (NH) | sum dy' (NH) | pad | sum_{y<0} da*y ]
// Both weight matrices are read as stored (k-major for these products: W2 [NO][NH], W1 [NH][K]).
template <int HC>
__global__ __launch_bounds__(512) void mlp_ln_bwd_kernel(const DosxMlpLnBwd a) {
  DOSX_SET_MAIN_PRIO();
  extern __shared__ __align__(16) float sm[];
  constexpr int D = HC ? 4 : 2;
  const int K = HC ? 128 * HC : a.K, NH = HC ? 128 * HC : a.NH, NO = HC ? 64 * HC : a.NO, M = a.M;
  const int LDY = NO + 4, LDT = NH + 4;
  float* Ys = sm;                                  // [16][LDY]  dy tile
  float* T = Ys + MR * LDY;                        // [16][LDT]  da -> dz
  float* Ps = T + MR * LDT;                        // [8][2][NH] column sums of the 8 waves + 8 dalpha terms
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  float* Wp = Ps + 16 * NH + 8 + wave * WP_FLOATS; // this wave's private weight-chunk buffer
  const int m0 = blockIdx.x * MR;
  KnStream<D, HC != 0> s1, s2;
  float4 yin[2];                                   // dy tile (row tid/32, 4-float groups tid%32 + 32 j; NO <= 256): requested first
  {
    const int rr = min(m0 + (tid >> 5), M - 1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = (tid & 31) * 4 + 128 * j;
      if (c < NO) yin[j] = ld4(a.dy + (size_t)rr * a.lddy + c);
    }
  }
  s1.prefetch(a.w2, NH, wave * 32 < NH ? wave * 32 : 0, NO, lane);        // (the first column block's chunks, before anything else)

  // operands of the row phase (rows 2*wave, 2*wave+1; columns lane*4 + 256 j), in flight under the first product
  float4 gam[NCG], bet[NCG], xh[2][NCG];
  float rs[2];
  bool on[NCG];
#pragma unroll
  for (int j = 0; j < NCG; ++j) {
    const int c = lane * 4 + 256 * j;
    on[j] = c < NH;
    gam[j] = ld4(a.gamma + (on[j] ? c : 0));
    bet[j] = ld4(a.beta + (on[j] ? c : 0));
  }
  const float alpha = *a.alpha;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = min(m0 + wave * 2 + i, M - 1);
    rs[i] = a.rstd[r];
#pragma unroll
    for (int j = 0; j < NCG; ++j) xh[i][j] = ld4(a.xhat + (size_t)r * NH + (on[j] ? lane * 4 + 256 * j : 0));
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = (tid & 31) * 4 + 128 * j;
    if (c < NO) st4(Ys + (tid >> 5) * LDY + c, yin[j]);
  }
  __syncthreads();                                         // Ys visible
  // ---- first product: da tile, 256-column blocks (this wave: two 16-column tiles), k over NO ----
  for (int blk = 0; blk * 256 < NH; ++blk) {
    const int cb = blk * 256 + wave * 32;
    const bool cok = cb < NH;
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (blk) s1.prefetch(a.w2, NH, cok ? cb : 0, NO, lane);
    s1.run(acc, NO, Ys, LDY, Wp, lane);
    if ((blk + 1) * 256 >= NH) s2.prefetch(a.w1, K, wave * 32 < K ? wave * 32 : 0, NH, lane);     // in flight under the row phase
    if (cok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        T[(4 * g4 + r) * LDT + cb + l15] = acc[0][r];
        T[(4 * g4 + r) * LDT + cb + l15 + 16] = acc[1][r];
      }
    }
  }
  __syncthreads();                                         // da tile complete
  // ---- row phase: PReLU backward, LayerNorm backward over the row; column sums for dgamma / dbeta / dalpha ----
  {
    const float invN = 1.f / (float)NH;
    float4 pg[NCG], pb[NCG];
    float pal = 0.f;
#pragma unroll
    for (int j = 0; j < NCG; ++j) { pg[j] = f4zero(); pb[j] = f4zero(); }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lr = wave * 2 + i, r = m0 + lr;
      const bool rvalid = r < M;                   // (wave-uniform)
      float4 dxh[NCG];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < NCG; ++j) {
        dxh[j] = f4zero();
        if (!on[j] || !rvalid) continue;
        float4 dy = ld4(T + lr * LDT + lane * 4 + 256 * j);
        const float4 x = xh[i][j], gm = gam[j], bt = bet[j];
        const float y0 = x.x * gm.x + bt.x, y1 = x.y * gm.y + bt.y, y2 = x.z * gm.z + bt.z, y3 = x.w * gm.w + bt.w;
        if (y0 < 0.f) { pal += dy.x * y0; dy.x *= alpha; }
        if (y1 < 0.f) { pal += dy.y * y1; dy.y *= alpha; }
        if (y2 < 0.f) { pal += dy.z * y2; dy.z *= alpha; }
        if (y3 < 0.f) { pal += dy.w * y3; dy.w *= alpha; }
        pg[j].x += dy.x * x.x; pg[j].y += dy.y * x.y; pg[j].z += dy.z * x.z; pg[j].w += dy.w * x.w;
        pb[j] = f4add(pb[j], dy);
        dxh[j] = make_float4(dy.x * gm.x, dy.y * gm.y, dy.z * gm.z, dy.w * gm.w);
        s1 += dxh[j].x + dxh[j].y + dxh[j].z + dxh[j].w;
        s2 += dxh[j].x * x.x + dxh[j].y * x.y + dxh[j].z * x.z + dxh[j].w * x.w;
      }
      const float m1 = wave_sum(s1) * invN, m2 = wave_sum(s2) * invN;
#pragma unroll
      for (int j = 0; j < NCG; ++j) {
        if (!on[j]) continue;
        float4 o = f4zero();                       // rows beyond M feed zeros to the second product
        if (rvalid) {
          const float4 x = xh[i][j];
          o = make_float4(rs[i] * (dxh[j].x - m1 - x.x * m2), rs[i] * (dxh[j].y - m1 - x.y * m2),
                          rs[i] * (dxh[j].z - m1 - x.z * m2), rs[i] * (dxh[j].w - m1 - x.w * m2));
          st4(a.dz + (size_t)r * NH + lane * 4 + 256 * j, o);
        }
        st4(T + lr * LDT + lane * 4 + 256 * j, o);
      }
    }
#pragma unroll
    for (int j = 0; j < NCG; ++j) {
      if (!on[j]) continue;
      st4(Ps + (wave * 2 + 0) * NH + lane * 4 + 256 * j, pg[j]);
      st4(Ps + (wave * 2 + 1) * NH + lane * 4 + 256 * j, pb[j]);
    }
    const float sal = wave_sum(pal);
    if (lane == 0) Ps[16 * NH + wave] = sal;
  }
  __syncthreads();                                         // dz tile + column sums complete
  {
    float* prow = a.partials + (size_t)blockIdx.x * a.partial_ld;
    for (int cc = tid; cc < 2 * NH; cc += 512) {
      const int which = cc >= NH ? 1 : 0, col = cc - which * NH;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += Ps[(w * 2 + which) * NH + col];
      prow[which * NH + col] = s;
    }
    if (tid == 0) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += Ps[16 * NH + w];
      prow[a.partial_ld - 1] = s;
    }
  }
  // ---- second product: dcat = dz . W1, 256-column blocks (two 16-column tiles per wave), k over NH; C straight to HBM ----
  for (int blk = 0; blk * 256 < K; ++blk) {
    const int cb = blk * 256 + wave * 32;
    const bool cok = cb < K;
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (blk) s2.prefetch(a.w1, K, cok ? cb : 0, NH, lane);
    s2.run(acc, NH, T, LDT, Wp, lane);
    if (cok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + 4 * g4 + r;
        if (row < M) {
          float o0 = acc[0][r], o1 = acc[1][r];
          if (a.add_dy) {          // + dy on the first NO columns: the residual connection around the block (Ys still holds the dy tile)
            if (cb + l15 < NO) o0 += Ys[(4 * g4 + r) * LDY + cb + l15];
            if (cb + l15 + 16 < NO) o1 += Ys[(4 * g4 + r) * LDY + cb + l15 + 16];
          }
          a.dcat[(size_t)row * a.lddcat + cb + l15] = o0;
          a.dcat[(size_t)row * a.lddcat + cb + l15 + 16] = o1;
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------
// COLUMN-SPLIT form of the two kernels above (round 6).  A batch of the benchmark configuration has ~450 node
// rows = 29 tiles of 16 rows: the kernels above run 29 workgroups on a 256-CU chip, each pulling ALL of the block's weights
// (384 KB at hidden 128) through one CU and issuing all of its MFMAs (5 us of matrix pipe per workgroup at 16 rows).  Here a
// tile is shared by NS = hidden / 16 workgroups of 4 waves: workgroup (tile, j) owns 32 columns of the first product and 16
// columns of the second (1 / NS of the weights, 1 / NS of the MFMAs: 0.6 us), its 4 waves split the k range and add their partial
// tiles through LDS in wave order.  The LayerNorm in the middle needs whole rows, and the second product the whole activated tile
// as its A operand, so the NS workgroups EXCHANGE their slices of the intermediate in-launch: publish with sc1 stores, drain, one
// ticket per workgroup on the tile's counter, then - unlike the last-arriver reductions of gemm.hip - EVERY workgroup waits for
// the counter to reach NS (one lane polls with agent-scope loads, bounded) and reads the whole 16 x NH tile back with sc1 loads;
// row statistics and activations are recomputed by all NS workgroups (16 rows x NH elements: nothing next to a launch).
// The siblings of a tile are consecutive workgroups of one grid of <= 512 small workgroups (25 KB of LDS, 4 waves: several fit a
// CU, also next to the weight-gradient stream's workgroups), so they are dispatched together; the poll is bounded all the same -
// a sibling that never arrives costs a wrong tile, not a hung GPU.  Every weight / input fragment is requested at kernel start,
// straight from global memory into the MFMA operand registers (no LDS staging: a fragment is used by exactly one wave).
// Counters: NS tickets per exchange + NS exit tickets; the workgroup that draws the last exit ticket stores zero.
typedef int v2i32 __attribute__((ext_vector_type(2)));
typedef int v4i32_ __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void cs_publish_done_and_wait(int* cnt, const int target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's published stores have completed at the coherent level
  __syncthreads();
  if (threadIdx.x == 0) {
    dosx_ticket(cnt);
    int it = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++it < (1 << 21)) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    // (compiler only: the read-back below stays below)
}
__device__ __forceinline__ void cs_exit(int* cnt, const int last_ticket) {
  if (threadIdx.x == 0) {
    const int tk = dosx_ticket(cnt);
    if (tk == last_ticket) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
#define CS_MFMA4(ACC, A4, B4)                                              \
  do {                                                                     \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A4).x, (B4).x, ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A4).y, (B4).y, ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A4).z, (B4).z, ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A4).w, (B4).w, ACC, 0, 0, 0); \
  } while (0)

// H = hidden: (K, NH, NO) = (2H, 2H, H), k0 a multiple of H / 2.  grid = tiles * NS, 256 threads.
template <int H>
__global__ __launch_bounds__(256) void mlp_ln_cs_fwd_kernel(const DosxMlpLn a) {
  DOSX_SET_MAIN_PRIO();
  constexpr int K = 2 * H, NH = 2 * H, NO = H, NS = H / 16;
  constexpr int KW = K / 4, S1 = KW / 16;           // k range of one wave (both products reduce over 2H), its 16-wide steps
  constexpr int NG = NH / 64;                       // float4 column groups per lane in the row phase (16 lanes per row)
  constexpr int LDT = NH + 4, LDR = 36;
  constexpr int KW3 = NO / 4, S3 = KW3 / 16 > 0 ? KW3 / 16 : 1;   // third product: reduces over H (hidden 64: one 16-wide step, waves 0-3 .. see below)
  __shared__ __align__(16) float T[16 * LDT];
  __shared__ __align__(16) float Rd[4 * 16 * LDR * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  const int row = tid >> 4, q = tid & 15;           // row-phase / epilogue coordinates: 16 lanes per tile row
  const int tile = (int)blockIdx.x / NS, j = (int)blockIdx.x - tile * NS;
  const int m0 = tile * 16, M = a.M;
  int* cnt = a.cs_cnt + tile;
  // ---- operands that depend on nothing: requested first, in the order they are used ----
  const int kbase = wave * KW;
  const int rowA = min(m0 + l15, M - 1);
  const float* ap = kbase < a.k0 ? a.a0 + (size_t)rowA * a.lda0 + kbase : a.a1 + (size_t)rowA * a.lda1 + (kbase - a.k0);
  float4 av[S1], bw1[2][S1], bw2[S1];
#pragma unroll
  for (int s = 0; s < S1; ++s) av[s] = ld4(ap + 16 * s + 4 * g4);
  {
    const float* w1p = a.w1 + (size_t)(32 * j + l15) * K + kbase + 4 * g4;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < S1; ++s) bw1[t][s] = ld4(w1p + (size_t)(16 * t) * K + 16 * s);
  }
  const float2 b1v = *reinterpret_cast<const float2*>(a.b1 + 32 * j + 2 * q);
  float4 gam[NG], bet[NG];
#pragma unroll
  for (int i = 0; i < NG; ++i) { gam[i] = ld4(a.gamma + 4 * q + 64 * i); bet[i] = ld4(a.beta + 4 * q + 64 * i); }
  const float alpha = *a.alpha;
  {
    const float* w2p = a.w2 + (size_t)(16 * j + l15) * NH + kbase + 4 * g4;
#pragma unroll
    for (int s = 0; s < S1; ++s) bw2[s] = ld4(w2p + 16 * s);
  }
  const int grow = m0 + row;
  const bool rvalid = grow < M;
  const float b2v = a.b2[16 * j + q];
  const float resv = (a.res && rvalid) ? a.res[(size_t)grow * a.ldres + 16 * j + q] : 0.f;
  // ---- first product: this workgroup's 32 columns, this wave's k range; partial tiles -> LDS ----
  {
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < S1; ++s) {
      CS_MFMA4(acc[0], av[s], bw1[0][s]);
      CS_MFMA4(acc[1], av[s], bw1[1][s]);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) Rd[(wave * 16 + 4 * g4 + r) * LDR + 16 * t + l15] = acc[t][r];
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rZ = __builtin_amdgcn_make_buffer_rsrc((void*)a.cs_buf, 0, 0x7fffffff, 0x00020000);
  {
    float2 z = b1v;
#pragma unroll
    for (int w = 0; w < 4; ++w) {                   // (wave order: fixed)
      const float2 p = *reinterpret_cast<const float2*>(Rd + (w * 16 + row) * LDR + 2 * q);
      z.x += p.x; z.y += p.y;
    }
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i32, z), rZ, (uint32_t)(((size_t)(m0 + row) * NH + 32 * j + 2 * q) * 4), 0, 16);   // sc1
  }
  cs_publish_done_and_wait(cnt, NS);
  // ---- row phase on the WHOLE tile (every sibling): LayerNorm statistics (two-pass, like torch), xhat / rstd out (rows dealt
  //      over the siblings), prelu(xhat * gamma + beta) -> T ----
  {
    float4 z[NG];
    float sm_ = 0.f;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      z[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rZ, (uint32_t)(((size_t)(m0 + row) * NH + 4 * q + 64 * i) * 4), 0, 16));   // sc1
      sm_ += (z[i].x + z[i].y) + (z[i].z + z[i].w);
    }
    const float invN = 1.f / (float)NH;
    const float mean = row16_sum(sm_) * invN;
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      z[i] = make_float4(z[i].x - mean, z[i].y - mean, z[i].z - mean, z[i].w - mean);
      qq += (z[i].x * z[i].x + z[i].y * z[i].y) + (z[i].z * z[i].z + z[i].w * z[i].w);
    }
    const float rstd = rsqrtf(row16_sum(qq) * invN + DOSX_LN_EPS);
    const bool mine = rvalid && (row & (NS - 1)) == j;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const float4 xh = make_float4(z[i].x * rstd, z[i].y * rstd, z[i].z * rstd, z[i].w * rstd);
      if (mine) st4(a.xhat + (size_t)grow * NH + 4 * q + 64 * i, xh);
      float4 y = make_float4(xh.x * gam[i].x + bet[i].x, xh.y * gam[i].y + bet[i].y, xh.z * gam[i].z + bet[i].z, xh.w * gam[i].w + bet[i].w);
      y.x = y.x >= 0.f ? y.x : alpha * y.x; y.y = y.y >= 0.f ? y.y : alpha * y.y;
      y.z = y.z >= 0.f ? y.z : alpha * y.z; y.w = y.w >= 0.f ? y.w : alpha * y.w;
      st4(T + row * LDT + 4 * q + 64 * i, y);
    }
    if (mine && q == 0) a.rstd[grow] = rstd;
  }
  __syncthreads();
  // ---- second product: this workgroup's 16 output columns ----
  {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S1; ++s) {
      const float4 t4 = ld4(T + l15 * LDT + kbase + 16 * s + 4 * g4);
      CS_MFMA4(acc, t4, bw2[s]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Rd[(wave * 16 + 4 * g4 + r) * LDR + l15] = acc[r];
  }
  __syncthreads();
  float o = b2v;
#pragma unroll
  for (int w = 0; w < 4; ++w) o += Rd[(w * 16 + row) * LDR + q];
  o += resv;
  const bool third = a.w3 != nullptr;
  if (!third) {
    if (rvalid) a.out[(size_t)grow * a.ldo + 16 * j + q] = o;
    cs_exit(cnt, 2 * NS - 1);
    return;
  }
  // ---- third product (DosxMlpLn.w3: the next layer's node products): the finished rows are exchanged the same way - `out` itself
  //      is the medium (sc1 stores, sc1 read-back of the 16 x NO tile) - then 4H / NS = 64 columns of pq per workgroup ----
  const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, 0x7fffffff, 0x00020000);
  // (its weight fragments: 64 columns c = 64 j + 16 t + l15 of pq <-> block b = c / n3, row n = c % n3 of w3; this wave reduces
  //  over k in [wave * H / 4, +H / 4))
  float4 bw3[4][S3];
  {
    constexpr bool full = KW3 >= 16;                // hidden 64: H / 4 = 16 -> one step per wave; (hidden 32 would need 8-wide steps: not instantiated)
    static_assert(full, "hidden >= 64");
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = 64 * j + 16 * t + l15, b = c / a.n3, n = c - b * a.n3;
      const float* wp = a.w3 + (size_t)n * a.ldw3 + b * NO + wave * KW3 + 4 * g4;
#pragma unroll
      for (int s = 0; s < S3; ++s) bw3[t][s] = ld4(wp + 16 * s);
    }
  }
  if (rvalid) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, o), rO, (uint32_t)(((size_t)grow * a.ldo + 16 * j + q) * 4), 0, 16);   // sc1
  cs_publish_done_and_wait(cnt, 2 * NS);
  {
    constexpr int LDO = NO + 4, OG = NO / 64;       // out tile [16][NO + 4] in T; float4 groups per lane
    const int rr = min(grow, M - 1);
#pragma unroll
    for (int i = 0; i < OG; ++i) {
      const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rO, (uint32_t)(((size_t)rr * a.ldo + 4 * q + 64 * i) * 4), 0, 16));   // sc1
      st4(T + row * LDO + 4 * q + 64 * i, v);
    }
    __syncthreads();
    f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < S3; ++s) {
      const float4 t4 = ld4(T + l15 * LDO + wave * KW3 + 16 * s + 4 * g4);
#pragma unroll
      for (int t = 0; t < 4; ++t) CS_MFMA4(acc[t], t4, bw3[t][s]);
    }
    constexpr int LDR3 = 68;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) Rd[(wave * 16 + 4 * g4 + r) * LDR3 + 16 * t + l15] = acc[t][r];
    __syncthreads();
    float4 p = f4zero();
#pragma unroll
    for (int w = 0; w < 4; ++w) p = f4add(p, ld4(Rd + (w * 16 + row) * LDR3 + 4 * q));
    if (rvalid) st4(a.pq + (size_t)grow * a.ldpq + 64 * j + 4 * q, p);
  }
  cs_exit(cnt, 3 * NS - 1);
}

// The node ENCODER (Linear(Fa, H) -> PReLU -> Linear(H, H): DOSTransformer_phonon.py:129,141) + the first message-passing
// layer's node products pq = x0 . [Wa | Wb]^T, column-split like the NodeModel kernel above: three launch-bound N-row launches
// (two dosx_gemm + dosx_gemm_pair) as one, two in-launch exchanges (z: the saved pre-activation; x0: the output - both are outputs
// anyway and serve as the exchange media).  Fa is any width (phonon: 118 - rows of 472 bytes, so the first product's fragments are
// 8-byte loads with the tail zeroed); the k range of the first product is Fa rounded up to 64, split over the 4 waves.
template <int H, int KP>                            // KP: Fa rounded up to a multiple of 64
__global__ __launch_bounds__(256) void enc_cs_fwd_kernel(const DosxEncCs a) {
  DOSX_SET_MAIN_PRIO();
  constexpr int NS = H / 16, KW1 = KP / 4, S1 = KW1 / 16, KW2 = H / 4, S2 = KW2 / 16 > 0 ? KW2 / 16 : 1;
  constexpr int LDT = H + 4, LDR = 20, OG = H / 64, LDR3 = 68;
  static_assert(KW2 >= 16, "hidden >= 64");
  __shared__ __align__(16) float T[16 * LDT];
  __shared__ __align__(16) float Rd[4 * 16 * LDR3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  const int row = tid >> 4, q = tid & 15;
  const int tile = (int)blockIdx.x / NS, j = (int)blockIdx.x - tile * NS;
  const int m0 = tile * 16, M = a.M, Fa = a.Fa;
  int* cnt = a.cs_cnt + tile;
  const int rowA = min(m0 + l15, M - 1), grow = m0 + row, rc = min(grow, M - 1);
  const bool rvalid = grow < M;
  // ---- first product's fragments: x rows and W0 rows, both [*, Fa] with 8-byte aligned rows; k beyond Fa reads as zero ----
  float4 av[S1], bw0[S1];
  {
    const float* xp = a.x + (size_t)rowA * a.ldx;
    const float* wp = a.w0 + (size_t)(16 * j + l15) * a.ldw0;
#pragma unroll
    for (int s = 0; s < S1; ++s) {
      const int k = wave * KW1 + 16 * s + 4 * g4;
      float2 x0 = make_float2(0.f, 0.f), x1 = x0, w0_ = x0, w1_ = x0;
      if (k + 1 < Fa) { x0 = *reinterpret_cast<const float2*>(xp + k); w0_ = *reinterpret_cast<const float2*>(wp + k); }
      else if (k < Fa) { x0.x = xp[k]; w0_.x = wp[k]; }
      if (k + 3 < Fa) { x1 = *reinterpret_cast<const float2*>(xp + k + 2); w1_ = *reinterpret_cast<const float2*>(wp + k + 2); }
      else if (k + 2 < Fa) { x1.x = xp[k + 2]; w1_.x = wp[k + 2]; }
      av[s] = make_float4(x0.x, x0.y, x1.x, x1.y);
      bw0[s] = make_float4(w0_.x, w0_.y, w1_.x, w1_.y);
    }
  }
  const float b0v = a.b0[16 * j + q], b2v = a.b2[16 * j + q], alpha = *a.alpha;
  float4 bw2[S2];
  {
    const float* w2p = a.w2 + (size_t)(16 * j + l15) * H + wave * KW2 + 4 * g4;
#pragma unroll
    for (int s = 0; s < S2; ++s) bw2[s] = ld4(w2p + 16 * s);
  }
  float4 bw3[4][S2];
  {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = 64 * j + 16 * t + l15, b = c / a.n3, n = c - b * a.n3;
      const float* wp = a.w3 + (size_t)n * a.ldw3 + b * H + wave * KW2 + 4 * g4;
#pragma unroll
      for (int s = 0; s < S2; ++s) bw3[t][s] = ld4(wp + 16 * s);
    }
  }
  // ---- z slice (16 columns) ----
  {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S1; ++s) CS_MFMA4(acc, av[s], bw0[s]);
#pragma unroll
    for (int r = 0; r < 4; ++r) Rd[(wave * 16 + 4 * g4 + r) * LDR + l15] = acc[r];
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rZ = __builtin_amdgcn_make_buffer_rsrc((void*)a.z, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, 0x7fffffff, 0x00020000);
  {
    float z = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) z += Rd[(w * 16 + row) * LDR + q];
    z += b0v;
    if (rvalid) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, z), rZ, (uint32_t)(((size_t)grow * H + 16 * j + q) * 4), 0, 16);   // sc1
  }
  cs_publish_done_and_wait(cnt, NS);
  // ---- PReLU(z tile) -> T ----
#pragma unroll
  for (int i = 0; i < OG; ++i) {
    float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rZ, (uint32_t)(((size_t)rc * H + 4 * q + 64 * i) * 4), 0, 16));   // sc1
    v.x = v.x >= 0.f ? v.x : alpha * v.x; v.y = v.y >= 0.f ? v.y : alpha * v.y;
    v.z = v.z >= 0.f ? v.z : alpha * v.z; v.w = v.w >= 0.f ? v.w : alpha * v.w;
    st4(T + row * LDT + 4 * q + 64 * i, v);
  }
  __syncthreads();
  // ---- x0 slice (16 columns) ----
  {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S2; ++s) {
      const float4 t4 = ld4(T + l15 * LDT + wave * KW2 + 16 * s + 4 * g4);
      CS_MFMA4(acc, t4, bw2[s]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Rd[(wave * 16 + 4 * g4 + r) * LDR + l15] = acc[r];
  }
  __syncthreads();
  {
    float o = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) o += Rd[(w * 16 + row) * LDR + q];
    o += b2v;
    if (rvalid) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, o), rO, (uint32_t)(((size_t)grow * a.ldo + 16 * j + q) * 4), 0, 16);   // sc1
  }
  cs_publish_done_and_wait(cnt, 2 * NS);
  // ---- pq: 64 columns of x0 . [Wa | Wb]^T ----
  {
#pragma unroll
    for (int i = 0; i < OG; ++i)
      st4(T + row * LDT + 4 * q + 64 * i,
          __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rO, (uint32_t)(((size_t)rc * a.ldo + 4 * q + 64 * i) * 4), 0, 16)));   // sc1
    __syncthreads();
    f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < S2; ++s) {
      const float4 t4 = ld4(T + l15 * LDT + wave * KW2 + 16 * s + 4 * g4);
#pragma unroll
      for (int t = 0; t < 4; ++t) CS_MFMA4(acc[t], t4, bw3[t][s]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) Rd[(wave * 16 + 4 * g4 + r) * LDR3 + 16 * t + l15] = acc[t][r];
    __syncthreads();
    float4 p = f4zero();
#pragma unroll
    for (int w = 0; w < 4; ++w) p = f4add(p, ld4(Rd + (w * 16 + row) * LDR3 + 4 * q));
    if (rvalid) st4(a.pq + (size_t)grow * a.ldpq + 64 * j + 4 * q, p);
  }
  cs_exit(cnt, 3 * NS - 1);
}

// Backward, column-split the same way: da slice (32 columns) -> exchange -> PReLU / LayerNorm backward of the whole tile by every
// sibling (dz rows dealt over the siblings; column sums of this workgroup's 32 columns; dalpha by sibling 0) -> 32 columns of dcat.
// PRE: what PRODUCES dy, in the same launch (DosxMlpLnBwd.pre) - the N-row kernel that used to run in front of this one:
//   1  the node side of the LATER layer's factored input gradient (dosx_node_grad: source-node sums of dz, dx = res + res2 +
//      aggS Wa + aggD Wb): the tile's 16 nodes are dealt over the siblings for the gather (16 / NS nodes each, their rows summed
//      in CSR order; at hidden 128 two waves per node, halves added in order), the sums are exchanged (aggs itself is the medium),
//      every sibling multiplies its 16 columns of dx, dx is exchanged (dy itself is the medium) - three exchanges per launch;
//   2  the to_dense_batch / key-LayerNorm backward + the pooled decoder gradient (dosx_dense_normalize_pool_bwd): row-local, every
//      sibling computes the tile's 16 rows, the rows are written by the sibling they are dealt to - no extra exchange.
template <int H, int PRE>
__global__ __launch_bounds__(256) void mlp_ln_cs_bwd_kernel(const DosxMlpLnBwd a) {
  DOSX_SET_MAIN_PRIO();
  constexpr int K = 2 * H, NH = 2 * H, NO = H, NS = H / 16;
  constexpr int KW1 = NO / 4, S1 = KW1 / 16;        // first product reduces over NO = H
  constexpr int KW2 = NH / 4, S2 = KW2 / 16;        // second over NH = 2H
  constexpr int NG = NH / 64, OG = NO / 64;
  constexpr int LDT = NH + 4, LDR = 36, LDY = NO + 4;
  constexpr int NPW = 16 / NS, PARTS = 4 / NPW;     // PRE 1: nodes per workgroup, waves per node
  constexpr int SP = H / 16;                        // PRE 1: 16-wide steps of one wave's k range (H of the 4H)
  static_assert(S1 >= 1, "hidden >= 64");
  __shared__ __align__(16) float T[16 * LDT];
  __shared__ __align__(16) float Rd[4 * 16 * LDR];
  __shared__ __align__(16) float Sg[2 * 16 * 32];
  __shared__ __align__(16) float Ys[PRE ? 16 * LDY : 4];
  __shared__ __align__(16) float Pg[PRE == 1 ? 4 * NH : 4];
  __shared__ float Pal[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  const int row = tid >> 4, q = tid & 15;
  const int tile = (int)blockIdx.x / NS, j = (int)blockIdx.x - tile * NS;
  const int m0 = tile * 16, M = a.M;
  int* cnt = a.cs_cnt + tile;
  const int rowA = min(m0 + l15, M - 1);
  const int grow = m0 + row;
  const bool rvalid = grow < M;
  const int rc = min(grow, M - 1);
  // ---- operands, requested first: (PRE 0: dy fragments,) W2 fragments ([k][n] as stored: one dword per MFMA), the row-phase rows ----
  float4 av[S1];
  float bq1[2][S1][4];
  {
    if constexpr (PRE == 0) {
      const float* dyp = a.dy + (size_t)rowA * a.lddy + wave * KW1 + 4 * g4;
#pragma unroll
      for (int s = 0; s < S1; ++s) av[s] = ld4(dyp + 16 * s);
    }
    const float* wp = a.w2 + (size_t)(wave * KW1 + 4 * g4) * NH + 32 * j + l15;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < S1; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) bq1[t][s][i] = wp[(size_t)(16 * s + i) * NH + 16 * t];
  }
  // ---- PRE 1, part 1: this wave's share of a node's source segment (bounds -> edge ids -> rows, 12 in flight), the weight
  //      fragments of the dx product, the aggD fragments (waves 2, 3) ----
  float pw[PRE == 1 ? SP : 1][4];
  float4 pa[PRE == 1 ? SP : 1];
  if constexpr (PRE == 1) {
    const int sgp = wave >> 1, kl0 = (wave & 1) * H;         // this wave multiplies rows [kl0, kl0 + H) of Wa (waves 0, 1) / Wb (2, 3)
    {
      const float* wp = a.pre_w + (size_t)(kl0 + 4 * g4) * a.pre_ldw + sgp * H + 16 * j + l15;
#pragma unroll
      for (int s = 0; s < SP; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) pw[s][i] = wp[(size_t)(16 * s + i) * a.pre_ldw];
    }
    if (sgp == 1) {
#pragma unroll
      for (int s = 0; s < SP; ++s) pa[s] = ld4(a.pre_aggd + (size_t)rowA * NH + kl0 + 16 * s + 4 * g4);
    }
    const int ns = wave % NPW, part = wave / NPW;
    const int n = m0 + j + NS * ns;
    const bool nv = n < M;
    const int beg = nv ? a.pre_rowptr_src[n] : 0, end = nv ? a.pre_rowptr_src[n + 1] : 0;
    const int per = (end - beg + PARTS - 1) / PARTS;
    const int pb = beg + part * per, pe = min(end, pb + per);
    const int c = lane * 4, cc = c < NH ? c : 0;
    float4 acc = f4zero();
    constexpr int U = 12;
    for (int j0 = pb; j0 < pe; j0 += 64) {
      const int je = min(pe, j0 + 64);
      const int myid = a.pre_perm_src[min(j0 + lane, je - 1)];
      for (int u0 = j0; u0 < je; u0 += U) {
        float4 m[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int ei = __builtin_amdgcn_readlane(myid, min(u0 - j0 + u, 63));
          m[u] = ld4(a.pre_dz + (size_t)ei * NH + cc);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (u0 + u < je) acc = f4add(acc, m[u]);
      }
    }
    if (c < NH) st4(Pg + (ns * PARTS + part) * NH + c, acc);
  }
  float4 gam[NG], bet[NG], xh[NG];
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    gam[i] = ld4(a.gamma + 4 * q + 64 * i); bet[i] = ld4(a.beta + 4 * q + 64 * i);
    xh[i] = ld4(a.xhat + (size_t)rc * NH + 4 * q + 64 * i);
  }
  const float rs = a.rstd[rc];
  const float alpha = *a.alpha;
  float bq2[2][S2][4];
  {
    const float* wp = a.w1 + (size_t)(wave * KW2 + 4 * g4) * K + 32 * j + l15;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < S2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) bq2[t][s][i] = wp[(size_t)(16 * s + i) * K + 16 * t];
  }
  float2 dyres = make_float2(0.f, 0.f);
  if constexpr (PRE == 0) {
    if (a.add_dy && 32 * j + 2 * q < NO) dyres = *reinterpret_cast<const float2*>(a.dy + (size_t)rc * a.lddy + 32 * j + 2 * q);
  }
  int xphase = 0;                                   // exchanges done so far
  if constexpr (PRE == 1) {
    // ---- part 2: the source sums -> aggs (write-through: the siblings read them back; the weight-gradient job reads them later) ----
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)a.pre_aggs, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)a.pre_dy, 0, 0x7fffffff, 0x00020000);
    __syncthreads();
    if (tid < NPW * (NH / 4)) {
      const int ns = tid / (NH / 4), c4 = tid - ns * (NH / 4);
      float4 v = ld4(Pg + (ns * PARTS) * NH + 4 * c4);
      if constexpr (PARTS == 2) v = f4add(v, ld4(Pg + (ns * PARTS + 1) * NH + 4 * c4));
      const int n = m0 + j + NS * ns;
      if (n < M) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i32_, v), rS, (uint32_t)(((size_t)n * NH + 4 * c4) * 4), 0, 16);   // sc1
    }
    cs_publish_done_and_wait(cnt, NS * ++xphase);
    // ---- part 3: 16 columns of dx = res + res2 + aggS Wa + aggD Wb (this wave: H of the 4H reduction) ----
    const int sgp = wave >> 1, kl0 = (wave & 1) * H;
    if (sgp == 0) {
#pragma unroll
      for (int s = 0; s < SP; ++s)
        pa[s] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rS, (uint32_t)(((size_t)rowA * NH + kl0 + 16 * s + 4 * g4) * 4), 0, 16));   // sc1
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < SP; ++s) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[s].x, pw[s][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[s].y, pw[s][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[s].z, pw[s][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[s].w, pw[s][3], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Rd[(wave * 16 + 4 * g4 + r) * LDR + l15] = acc[r];
    __syncthreads();
    {
      float o = (a.pre_res ? a.pre_res[(size_t)rc * a.pre_ldres + 16 * j + q] : 0.f) + (a.pre_res2 ? a.pre_res2[(size_t)rc * a.pre_ldres2 + 16 * j + q] : 0.f);
#pragma unroll
      for (int w = 0; w < 4; ++w) o += Rd[(w * 16 + row) * LDR + q];
      if (rvalid) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, o), rY, (uint32_t)(((size_t)grow * a.lddy + 16 * j + q) * 4), 0, 16);   // sc1
    }
    cs_publish_done_and_wait(cnt, NS * ++xphase);
#pragma unroll
    for (int i = 0; i < OG; ++i)
      st4(Ys + row * LDY + 4 * q + 64 * i,
          __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rY, (uint32_t)(((size_t)rc * a.lddy + 4 * q + 64 * i) * 4), 0, 16)));   // sc1
    __syncthreads();
  }
  if constexpr (PRE == 2) {
    // ---- dy rows = key-LayerNorm backward of the dense key gradient + the pooled decoder gradient (row-local; all 16 rows) ----
    const int dr = a.pre_dense_row[rc];
    const int gph = a.pre_node_graph[rc];
    const float* add = gph < a.pre_num_graphs ? a.pre_dpool + (size_t)gph * a.pre_ld_dpool : nullptr;
    const bool ghost = dr == a.pre_ghost_row;
    float4 gk[OG], xk[OG], ad[OG];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < OG; ++i) {
      const int c = 4 * q + 64 * i;
      gk[i] = ghost ? f4zero() : ld4(a.pre_dkv + (size_t)dr * NO + c);
      xk[i] = ghost ? f4zero() : ld4(a.pre_kvhat + (size_t)dr * NO + c);
      ad[i] = add ? ld4(add + c) : f4zero();
      s1 += (gk[i].x + gk[i].y) + (gk[i].z + gk[i].w);
      s2 += (gk[i].x * xk[i].x + gk[i].y * xk[i].y) + (gk[i].z * xk[i].z + gk[i].w * xk[i].w);
    }
    const float rsn = ghost ? 0.f : a.pre_rstd_nodes[rc];
    const float m1 = row16_sum(s1) * (1.f / (float)NO), m2 = row16_sum(s2) * (1.f / (float)NO);
    const bool mine = rvalid && (row & (NS - 1)) == j;
#pragma unroll
    for (int i = 0; i < OG; ++i) {
      const int c = 4 * q + 64 * i;
      const float4 o = make_float4(ad[i].x + rsn * (gk[i].x - m1 - xk[i].x * m2), ad[i].y + rsn * (gk[i].y - m1 - xk[i].y * m2),
                                   ad[i].z + rsn * (gk[i].z - m1 - xk[i].z * m2), ad[i].w + rsn * (gk[i].w - m1 - xk[i].w * m2));
      st4(Ys + row * LDY + c, o);
      if (mine) st4(a.pre_dy + (size_t)grow * a.lddy + c, o);
    }
    __syncthreads();
  }
  if constexpr (PRE != 0) {
#pragma unroll
    for (int s = 0; s < S1; ++s) av[s] = ld4(Ys + l15 * LDY + wave * KW1 + 16 * s + 4 * g4);
    if (a.add_dy && 32 * j + 2 * q < NO) dyres = *reinterpret_cast<const float2*>(Ys + row * LDY + 32 * j + 2 * q);
  }
  // ---- first product ----
  {
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < S1; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s].x, bq1[t][s][0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s].y, bq1[t][s][1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s].z, bq1[t][s][2], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s].w, bq1[t][s][3], acc[t], 0, 0, 0);
      }
    if constexpr (PRE == 1) __syncthreads();        // (Rd: the dx product's partial tiles have been read)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) Rd[(wave * 16 + 4 * g4 + r) * LDR + 16 * t + l15] = acc[t][r];
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rZ = __builtin_amdgcn_make_buffer_rsrc((void*)a.cs_buf, 0, 0x7fffffff, 0x00020000);
  {
    float2 z = make_float2(0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float2 p = *reinterpret_cast<const float2*>(Rd + (w * 16 + row) * LDR + 2 * q);
      z.x += p.x; z.y += p.y;
    }
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i32, z), rZ, (uint32_t)(((size_t)(m0 + row) * NH + 32 * j + 2 * q) * 4), 0, 16);   // sc1
  }
  cs_publish_done_and_wait(cnt, NS * ++xphase);
  // ---- row phase on the whole tile: PReLU backward, LayerNorm backward; dz out (rows dealt over the siblings) and -> T ----
  {
    const float invN = 1.f / (float)NH;
    float4 dxh[NG];
    float s1 = 0.f, s2 = 0.f, pal = 0.f;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      float4 dy = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rZ, (uint32_t)(((size_t)(m0 + row) * NH + 4 * q + 64 * i) * 4), 0, 16));   // sc1
      if (!rvalid) dy = f4zero();
      const float4 x = xh[i], gm = gam[i], bt = bet[i];
      const float y0 = x.x * gm.x + bt.x, y1 = x.y * gm.y + bt.y, y2 = x.z * gm.z + bt.z, y3 = x.w * gm.w + bt.w;
      if (y0 < 0.f) { pal += dy.x * y0; dy.x *= alpha; }
      if (y1 < 0.f) { pal += dy.y * y1; dy.y *= alpha; }
      if (y2 < 0.f) { pal += dy.z * y2; dy.z *= alpha; }
      if (y3 < 0.f) { pal += dy.w * y3; dy.w *= alpha; }
      if (2 * i + (q >> 3) == j) {                  // this column group lies in the workgroup's own 32 columns: column-sum operands
        st4(Sg + row * 32 + 4 * (q & 7), make_float4(dy.x * x.x, dy.y * x.y, dy.z * x.z, dy.w * x.w));
        st4(Sg + 16 * 32 + row * 32 + 4 * (q & 7), dy);
      }
      dxh[i] = make_float4(dy.x * gm.x, dy.y * gm.y, dy.z * gm.z, dy.w * gm.w);
      s1 += (dxh[i].x + dxh[i].y) + (dxh[i].z + dxh[i].w);
      s2 += (dxh[i].x * x.x + dxh[i].y * x.y) + (dxh[i].z * x.z + dxh[i].w * x.w);
    }
    const float m1 = row16_sum(s1) * invN, m2 = row16_sum(s2) * invN;
    const bool mine = rvalid && (row & (NS - 1)) == j;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const float4 x = xh[i];
      float4 o = f4zero();                          // rows beyond M feed zeros to the second product
      if (rvalid) o = make_float4(rs * (dxh[i].x - m1 - x.x * m2), rs * (dxh[i].y - m1 - x.y * m2),
                                  rs * (dxh[i].z - m1 - x.z * m2), rs * (dxh[i].w - m1 - x.w * m2));
      if (mine) st4(a.dz + (size_t)grow * NH + 4 * q + 64 * i, o);
      st4(T + row * LDT + 4 * q + 64 * i, o);
    }
    const float sal = wave_sum(pal);
    if (lane == 0) Pal[wave] = sal;
  }
  __syncthreads();
  {
    float* prow = a.partials + (size_t)tile * a.partial_ld;
    if (tid < 64) {
      const int which = tid >> 5, c = tid & 31;
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += Sg[which * 16 * 32 + r * 32 + c];
      prow[which * NH + 32 * j + c] = s;
    }
    if (tid == 64 && j == 0) prow[a.partial_ld - 1] = (Pal[0] + Pal[1]) + (Pal[2] + Pal[3]);
  }
  // ---- second product: 32 columns of dcat ----
  {
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < S2; ++s) {
      const float4 t4 = ld4(T + l15 * LDT + wave * KW2 + 16 * s + 4 * g4);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(t4.x, bq2[t][s][0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(t4.y, bq2[t][s][1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(t4.z, bq2[t][s][2], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(t4.w, bq2[t][s][3], acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) Rd[(wave * 16 + 4 * g4 + r) * LDR + 16 * t + l15] = acc[t][r];
  }
  __syncthreads();
  {
    float2 o = make_float2(0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float2 p = *reinterpret_cast<const float2*>(Rd + (w * 16 + row) * LDR + 2 * q);
      o.x += p.x; o.y += p.y;
    }
    o.x += dyres.x; o.y += dyres.y;                 // (the residual path LAST: dcat with add_dy == dcat without + dy, to the bit)
    if (rvalid) *reinterpret_cast<float2*>(a.dcat + (size_t)grow * a.lddcat + 32 * j + 2 * q) = o;
  }
  cs_exit(cnt, NS * (xphase + 1) - 1);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// hidden / 64 for the NodeModel's shapes (K, NH, NO) = (2, 2, 1) x hidden, hidden in {64, 128, 256}; 0: anything else
inline int node_shape(int K, int NH, int NO) { return (K == NH && NH == 2 * NO && (NO == 64 || NO == 128 || NO == 256)) ? NO / 64 : 0; }

}  // namespace

extern "C" int dosx_mlp_ln_supported(int K, int NH, int NO) {
  return K % 128 == 0 && K >= 128 && K <= 512 && NH % 128 == 0 && NH >= 128 && NH <= 512 && NO % 64 == 0 && NO >= 64 &&
         NO <= 256 && (NO <= 128 || NO % 128 == 0) && NO <= K;
}

// column-split form: the NodeModel shapes at hidden 64 / 128
extern "C" int dosx_mlp_ln_cs_supported(int K, int NH, int NO) {
  const int hc = node_shape(K, NH, NO);
  return hc == 1 || hc == 2;
}
extern "C" int dosx_mlp_ln_cs_tiles(int M) { return M <= 0 ? 0 : ceil_div(M, MR); }
extern "C" int64_t dosx_mlp_ln_cs_scratch_floats(int M, int NH) { return M <= 0 ? 0 : (int64_t)ceil_div(M, MR) * MR * NH; }

extern "C" int dosx_enc_cs_supported(int Fa, int H) { return (H == 64 || H == 128) && Fa >= 2 && Fa <= 256 && (Fa & 1) == 0; }

extern "C" int dosx_enc_cs_fwd(const DosxEncCs* ap, dosx_stream_t stream) {
  DOSX_CHECK_ARG(ap != nullptr, "dosx_enc_cs_fwd: null descriptor");
  const DosxEncCs& a = *ap;
  if (a.M <= 0) return 0;
  DOSX_CHECK_ARG(dosx_enc_cs_supported(a.Fa, a.H), "dosx_enc_cs_fwd: Fa=%d H=%d unsupported (even Fa <= 256, hidden 64 / 128)", a.Fa, a.H);
  DOSX_CHECK_ARG(a.x && a.w0 && a.b0 && a.alpha && a.w2 && a.b2 && a.z && a.out && a.w3 && a.pq && a.cs_cnt, "dosx_enc_cs_fwd: null operand");
  DOSX_CHECK_ARG((a.ldx & 1) == 0 && (a.ldw0 & 1) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 7) == 0 && (reinterpret_cast<uintptr_t>(a.w0) & 7) == 0 &&
                     aligned16(a.w2) && aligned16(a.w3) && aligned16(a.z) && aligned16(a.out) && aligned16(a.pq) && (a.ldo & 3) == 0 && a.ldo >= a.H &&
                     (a.ldw3 & 3) == 0 && (a.ldpq & 3) == 0 && a.nb3 * a.n3 == 4 * a.H && a.ldw3 >= a.nb3 * a.H,
                 "dosx_enc_cs_fwd: x / w0 rows 8-byte aligned (even leading dimensions), the rest 16-byte aligned; nb3 * n3 == 4 * H");
  const dim3 grid(ceil_div(a.M, MR) * (a.H / 16));
  hipStream_t st = to_stream(stream);
  const int kp = (a.Fa + 63) / 64 * 64;
#define DOSX_ENC(HH, KK) hipLaunchKernelGGL((enc_cs_fwd_kernel<HH, KK>), grid, dim3(256), 0, st, a)
  if (a.H == 64) { if (kp == 64) DOSX_ENC(64, 64); else if (kp == 128) DOSX_ENC(64, 128); else if (kp == 192) DOSX_ENC(64, 192); else DOSX_ENC(64, 256); }
  else { if (kp == 64) DOSX_ENC(128, 64); else if (kp == 128) DOSX_ENC(128, 128); else if (kp == 192) DOSX_ENC(128, 192); else DOSX_ENC(128, 256); }
#undef DOSX_ENC
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_mlp_ln_fwd(const DosxMlpLn* ap, dosx_stream_t stream) {
  DOSX_CHECK_ARG(ap != nullptr, "dosx_mlp_ln_fwd: null descriptor");
  const DosxMlpLn& a = *ap;
  if (a.M <= 0) return 0;
  DOSX_CHECK_ARG(dosx_mlp_ln_supported(a.K, a.NH, a.NO), "dosx_mlp_ln_fwd: K=%d NH=%d NO=%d unsupported", a.K, a.NH, a.NO);
  DOSX_CHECK_ARG(a.a0 && a.w1 && a.b1 && a.gamma && a.beta && a.alpha && a.w2 && a.b2 && a.xhat && a.rstd && a.out,
                 "dosx_mlp_ln_fwd: null operand");
  DOSX_CHECK_ARG(a.k0 > 0 && a.k0 <= a.K && (a.k0 & 3) == 0 && (a.k0 == a.K || a.a1), "dosx_mlp_ln_fwd: bad input split k0=%d", a.k0);
  DOSX_CHECK_ARG((a.lda0 & 3) == 0 && (a.lda1 & 3) == 0 && (a.ldo & 3) == 0 && (a.ldres & 3) == 0 && aligned16(a.a0) &&
                 aligned16(a.a1) && aligned16(a.out) && aligned16(a.res) && aligned16(a.xhat),
                 "dosx_mlp_ln_fwd: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
  const long long span = (const char*)a.w1 > (const char*)a.w2 ? (const char*)a.w1 - (const char*)a.w2 : (const char*)a.w2 - (const char*)a.w1;
  DOSX_CHECK_ARG(span + (long long)4 * a.NH * (a.K + a.NO) < 0x7fffffffLL, "dosx_mlp_ln_fwd: the two weight matrices are more than 2 GiB apart");
  if (a.w3 && !a.cs_buf)
    DOSX_CHECK_ARG(a.pq && a.nb3 >= 1 && a.n3 >= 256 && a.n3 % 256 == 0 && (a.ldw3 & 3) == 0 && a.ldw3 >= a.nb3 * a.NO && aligned16(a.w3) &&
                       a.ldpq >= a.nb3 * a.n3,
                   "dosx_mlp_ln_fwd: third product needs pq, n3 a multiple of 256, ldw3 >= nb3 * NO (multiple of 4), ldpq >= nb3 * n3");
  if (a.cs_buf) {         // column-split form
    DOSX_CHECK_ARG(dosx_mlp_ln_cs_supported(a.K, a.NH, a.NO) && a.cs_cnt, "dosx_mlp_ln_fwd: column-split form needs the NodeModel shape at hidden 64 / 128 and counters");
    DOSX_CHECK_ARG(a.k0 % (a.K / 4) == 0 && a.ldo >= a.NO && aligned16(a.cs_buf) && aligned16(a.w1) && aligned16(a.w2) && aligned16(a.gamma) &&
                   aligned16(a.beta) && (reinterpret_cast<uintptr_t>(a.b1) & 7) == 0,
                   "dosx_mlp_ln_fwd: column-split form needs k0 a multiple of K / 4 and aligned weights");
    if (a.w3) DOSX_CHECK_ARG(a.pq && a.nb3 >= 1 && a.nb3 * a.n3 == 4 * a.NO && (a.ldw3 & 3) == 0 && a.ldw3 >= a.nb3 * a.NO && aligned16(a.w3) && (a.ldpq & 3) == 0 && aligned16(a.pq), "dosx_mlp_ln_fwd: column-split third product needs nb3 * n3 == 4 * NO");
    const int ns = a.NO / 16;
    const dim3 grid(ceil_div(a.M, MR) * ns);
    if (a.NO == 64) hipLaunchKernelGGL(mlp_ln_cs_fwd_kernel<64>, grid, dim3(256), 0, to_stream(stream), a);
    else hipLaunchKernelGGL(mlp_ln_cs_fwd_kernel<128>, grid, dim3(256), 0, to_stream(stream), a);
    DOSX_LAUNCH_CHECK();
    return 0;
  }
  const size_t smem = sizeof(float) * ((size_t)MR * (a.K + 4) + (size_t)MR * (a.NH + 4) + 8 * (size_t)WP_FLOATS);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_ln_fwd_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_ln_fwd_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_ln_fwd_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_ln_fwd_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const dim3 grid(ceil_div(a.M, MR));
  switch (node_shape(a.K, a.NH, a.NO)) {
    case 1: hipLaunchKernelGGL(mlp_ln_fwd_kernel<1>, grid, dim3(512), smem, to_stream(stream), a); break;
    case 2: hipLaunchKernelGGL(mlp_ln_fwd_kernel<2>, grid, dim3(512), smem, to_stream(stream), a); break;
    case 4: hipLaunchKernelGGL(mlp_ln_fwd_kernel<4>, grid, dim3(512), smem, to_stream(stream), a); break;
    default: hipLaunchKernelGGL(mlp_ln_fwd_kernel<0>, grid, dim3(512), smem, to_stream(stream), a);
  }
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_mlp_ln_bwd_partial_rows(int M) { return M <= 0 ? 0 : ceil_div(M, MR); }

extern "C" int dosx_mlp_ln_bwd(const DosxMlpLnBwd* ap, dosx_stream_t stream) {
  DOSX_CHECK_ARG(ap != nullptr, "dosx_mlp_ln_bwd: null descriptor");
  const DosxMlpLnBwd& a = *ap;
  if (a.M <= 0) return 0;
  DOSX_CHECK_ARG(dosx_mlp_ln_supported(a.K, a.NH, a.NO), "dosx_mlp_ln_bwd: K=%d NH=%d NO=%d unsupported", a.K, a.NH, a.NO);
  DOSX_CHECK_ARG(a.dy && a.xhat && a.rstd && a.w1 && a.w2 && a.gamma && a.beta && a.alpha && a.dz && a.dcat && a.partials,
                 "dosx_mlp_ln_bwd: null operand");
  DOSX_CHECK_ARG((a.lddy & 3) == 0 && aligned16(a.dy) && aligned16(a.xhat) && aligned16(a.dz),
                 "dosx_mlp_ln_bwd: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
  DOSX_CHECK_ARG(a.partial_ld >= 2 * a.NH + 1, "dosx_mlp_ln_bwd: partial_ld %d < 2*NH+1", a.partial_ld);
  const long long span = (const char*)a.w1 > (const char*)a.w2 ? (const char*)a.w1 - (const char*)a.w2 : (const char*)a.w2 - (const char*)a.w1;
  DOSX_CHECK_ARG(span + (long long)4 * a.NH * (a.K + a.NO) < 0x7fffffffLL, "dosx_mlp_ln_bwd: the two weight matrices are more than 2 GiB apart");
  if (a.cs_buf) {         // column-split form
    DOSX_CHECK_ARG(dosx_mlp_ln_cs_supported(a.K, a.NH, a.NO) && a.cs_cnt, "dosx_mlp_ln_bwd: column-split form needs the NodeModel shape at hidden 64 / 128 and counters");
    DOSX_CHECK_ARG((a.lddcat & 1) == 0 && (reinterpret_cast<uintptr_t>(a.dcat) & 7) == 0 && aligned16(a.cs_buf) && aligned16(a.gamma) && aligned16(a.beta),
                   "dosx_mlp_ln_bwd: column-split form needs an 8-byte aligned dcat with an even leading dimension");
    DOSX_CHECK_ARG(a.pre >= 0 && a.pre <= 2, "dosx_mlp_ln_bwd: pre %d", a.pre);
    if (a.pre == 1)
      DOSX_CHECK_ARG(a.pre_dz && a.pre_rowptr_src && a.pre_perm_src && a.pre_aggd && a.pre_w && a.pre_aggs && a.pre_dy == a.dy && a.pre_ldw >= 2 * a.NO &&
                         aligned16(a.pre_dz) && aligned16(a.pre_aggd) && aligned16(a.pre_aggs) && aligned16(a.dy) && a.lddy >= a.NO,
                     "dosx_mlp_ln_bwd: pre = 1 (node side of the factored input gradient) needs dz / rowptr_src / perm_src / aggd / w / aggs and pre_dy == dy");
    if (a.pre == 2)
      DOSX_CHECK_ARG(a.pre_dkv && a.pre_kvhat && a.pre_rstd_nodes && a.pre_dense_row && a.pre_dpool && a.pre_node_graph && a.pre_num_graphs > 0 &&
                         (a.pre_ld_dpool & 3) == 0 && a.pre_dy == a.dy && aligned16(a.pre_dkv) && aligned16(a.pre_kvhat) && aligned16(a.pre_dpool) && aligned16(a.dy),
                     "dosx_mlp_ln_bwd: pre = 2 (dense-key backward + pooled gradient) needs dkv / kvhat / rstd_nodes / dense_row / dpool / node_graph and pre_dy == dy");
    const int ns = a.NO / 16;
    const dim3 grid(ceil_div(a.M, MR) * ns);
    hipStream_t st = to_stream(stream);
#define DOSX_CSB(HH)                                                                                           \
  do {                                                                                                         \
    if (a.pre == 1) hipLaunchKernelGGL((mlp_ln_cs_bwd_kernel<HH, 1>), grid, dim3(256), 0, st, a);              \
    else if (a.pre == 2) hipLaunchKernelGGL((mlp_ln_cs_bwd_kernel<HH, 2>), grid, dim3(256), 0, st, a);         \
    else hipLaunchKernelGGL((mlp_ln_cs_bwd_kernel<HH, 0>), grid, dim3(256), 0, st, a);                         \
  } while (0)
    if (a.NO == 64) DOSX_CSB(64);
    else DOSX_CSB(128);
#undef DOSX_CSB
    DOSX_LAUNCH_CHECK();
    return 0;
  }
  DOSX_CHECK_ARG(a.pre == 0, "dosx_mlp_ln_bwd: pre needs the column-split form (cs_buf)");
  const size_t smem = sizeof(float) * ((size_t)MR * (a.NO + 4) + (size_t)MR * (a.NH + 4) + 16 * (size_t)a.NH + 8 + 8 * (size_t)WP_FLOATS);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_ln_bwd_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_ln_bwd_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_ln_bwd_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_ln_bwd_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const dim3 grid(ceil_div(a.M, MR));
  switch (node_shape(a.K, a.NH, a.NO)) {
    case 1: hipLaunchKernelGGL(mlp_ln_bwd_kernel<1>, grid, dim3(512), smem, to_stream(stream), a); break;
    case 2: hipLaunchKernelGGL(mlp_ln_bwd_kernel<2>, grid, dim3(512), smem, to_stream(stream), a); break;
    case 4: hipLaunchKernelGGL(mlp_ln_bwd_kernel<4>, grid, dim3(512), smem, to_stream(stream), a); break;
    default: hipLaunchKernelGGL(mlp_ln_bwd_kernel<0>, grid, dim3(512), smem, to_stream(stream), a);
  }
  DOSX_LAUNCH_CHECK();
  return 0;
}
