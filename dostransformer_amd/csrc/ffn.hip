// Fused position-wise feed-forward half of the reference encoder layer (layers/transformer.py:141-148):
//
//     out = x + fc2( relu( fc1( LN1(x) ) ) ),        h = relu(fc1(LN1(x))) is also written out (saved for backward)
//
// One workgroup owns 32 rows and runs BOTH GEMMs: the 32 x 4H intermediate tile stays in LDS as the A operand of
// fc2, so the layer costs one launch, one prologue and one row epilogue instead of two of each (the k-loops of the
// two GEMMs at this workload's sizes are ~12 us together; the fixed part of a launch is ~5 us).  Same wave
// specialisation as gemm_kernel: waves 0-3 multiply, waves 4-7 stream the 128x32 weight chunks of fc1 (column block
// by column block) and then of fc2 through two LDS stage buffers, one barrier per chunk, loads two chunks deep;
// they also copy the finished h tile to HBM while the matrix waves are already in fc2.
// Needs H % 32 == 0 and H <= 128 (tile: 32 x 516 floats); larger H uses the two-GEMM path.
#include <stdlib.h>

#include "common.h"

typedef int v4i32 __attribute__((ext_vector_type(4)));

namespace {

constexpr int FBK = 32;          // k-chunk
constexpr int FLDW = FBK + 4;    // 36: padded rows of a staged weight chunk
constexpr int FBN = 128;         // columns per chunk / per column block

// HALF: the workgroup owns 16 rows and multiplies with the 16x16x4 MFMA (two 16-column tiles per wave instead of one
// 32-column tile): twice the workgroups, half the MFMA time each, and two of them fit the LDS of one CU.
template <bool HALF>
__global__ __launch_bounds__(512) void ffn_fwd_kernel(const DosxFfn a) {
  extern __shared__ __align__(16) float sm[];
  constexpr int R = HALF ? 16 : 32;                // rows per workgroup
  constexpr int ER = R / 8;                        // epilogue rows per wave
  const int H = a.H, H4 = 4 * a.H, M = a.M;
  const int LDX = H + 4, LDT = H4 + 4;
  float* Xs = sm;                                  // [R][LDX]  LN1(x) tile (A operand of fc1)
  float* T = Xs + R * LDX;                         // [R][LDT]  relu(fc1) tile (A operand of fc2)
  float* ST = T + R * LDT;                         // 2 stage buffers [128][36]; later the C tile [R][H+4]
  constexpr int STG = FBN * FLDW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int l15 = lane & 15, g4 = lane >> 4;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int m0 = blockIdx.x * R;
  const int nk1 = H / FBK, nb1 = H4 / FBN, n1 = nb1 * nk1, n2 = H4 / FBK, nch = n1 + n2;

  // epilogue operands of this wave's 4 rows, fetched at kernel start (all 8 waves)
  const int c0 = lane * 4;
  const bool con = c0 < H;
  float4 xres[ER], bias2 = f4zero();
  if (con) bias2 = ld4(a.b2 + c0);
#pragma unroll
  for (int i = 0; i < ER; ++i) {
    const int r = min(m0 + wave * ER + i, M - 1);
    xres[i] = ld4(a.x + (size_t)r * a.ldx + (con ? c0 : 0));
  }

  if (wave_u >= 4) {
    // =============================== staging waves ===============================================
    const int st = tid - 256, jr = st >> 3, kq = (st & 7) * 4;
    const float* wlo = a.w1 < a.w2 ? a.w1 : a.w2;
    const uint32_t d1 = (uint32_t)((const char*)a.w1 - (const char*)wlo), d2 = (uint32_t)((const char*)a.w2 - (const char*)wlo);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)wlo, 0, 0x7fffffff, 0x00020000);
    uint32_t v1[4], v2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v1[i] = d1 + (uint32_t)(((jr + 32 * i) * H + kq) * 4);                       // fc1 rows: (cb*128 + j), j = jr + 32 i
      v2[i] = d2 + (uint32_t)((min(jr + 32 * i, H - 1) * H4 + kq) * 4);            // fc2 rows: j < H (clamped)
    }
    float4 r0[4], r1[4];
    auto issue = [&](float4(&r)[4], int c) {
      const int cu = __builtin_amdgcn_readfirstlane(c);
      const bool p1 = cu < n1;
      const int so = p1 ? ((cu / nk1) * FBN * H + (cu % nk1) * FBK) * 4 : (cu - n1) * FBK * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        r[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rW, p1 ? v1[i] : v2[i], so, 0));
    };
    auto store = [&](float* buf, const float4(&r)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) st4(buf + (jr + 32 * i) * FLDW + kq, r[i]);
    };
    auto copy_h = [&]() {        // the finished relu(fc1) tile -> HBM (rows m0.., H4 columns), float4 per lane
      const int per_row = H4 / 4;
      for (int i = st; i < R * per_row; i += 256) {
        const int r = i / per_row, c = (i % per_row) * 4;
        if (m0 + r < M) st4(a.h + (size_t)(m0 + r) * a.ldh + c, ld4(T + r * LDT + c));
      }
    };
    issue(r0, 0);
    issue(r1, 1);
    store(ST, r0);
    issue(r0, 2);
    __syncthreads();                               // (matrix waves: Xs written) chunk 0 visible
    for (int c = 0; c < nch; c += 2) {
      if (c + 1 < nch) {
        store(ST + STG, r1);
        if (c + 3 < nch) issue(r1, c + 3);
      }
      __syncthreads();
      if (c + 1 == n1) { __syncthreads(); copy_h(); }        // phase boundary: T complete behind this extra barrier
      if (c + 1 >= nch) break;
      if (c + 2 < nch) {
        store(ST, r0);
        if (c + 4 < nch) issue(r0, c + 4);
      }
      __syncthreads();
      if (c + 2 == n1) { __syncthreads(); copy_h(); }
    }
  } else {
    // =============================== matrix waves ================================================
    {   // LN1(x) tile -> Xs  (row r = tid/8, 4-float groups tid%8 + 8 i)
      const int r = tid >> 3, rr = min(m0 + r, M - 1);
      const float mean = a.stats[2 * (size_t)rr], rstd = a.stats[2 * (size_t)rr + 1];
      for (int c = (tid & 7) * 4; c < H && r < R; c += 32) {
        const float4 v = ld4(a.x + (size_t)rr * a.ldx + c), g = ld4(a.gamma + c), b = ld4(a.beta + c);
        st4(Xs + r * LDX + c, make_float4((v.x - mean) * rstd * g.x + b.x, (v.y - mean) * rstd * g.y + b.y,
                                          (v.z - mean) * rstd * g.z + b.z, (v.w - mean) * rstd * g.w + b.w));
      }
    }
    __syncthreads();
    int c = 0;
    if constexpr (HALF) {
      f32x4 acc0, acc1;
      // ---- fc1: nb1 column blocks of 128 (this wave: two 16-column tiles), each nk1 chunks ----
      for (int cb = 0; cb < nb1; ++cb) {
        acc0 = f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = acc0;
        for (int kc = 0; kc < nk1; ++kc, ++c) {
          const float* Ws = ST + (c & 1) * STG;
#pragma unroll
          for (int kk = 0; kk < FBK; kk += 16) {
            const float4 av = ld4(Xs + l15 * LDX + kc * FBK + kk + 4 * g4);
            const float4 b0 = ld4(Ws + (wave * 32 + l15) * FLDW + kk + 4 * g4);
            const float4 b1 = ld4(Ws + (wave * 32 + 16 + l15) * FLDW + kk + 4 * g4);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b0.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b1.w, acc1, 0, 0, 0);
          }
          __syncthreads();
        }
        const int col = cb * FBN + wave * 32 + l15;
        const float b1a = a.b1[col], b1b = a.b1[col + 16];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          T[(4 * g4 + r) * LDT + col] = fmaxf(acc0[r] + b1a, 0.f);
          T[(4 * g4 + r) * LDT + col + 16] = fmaxf(acc1[r] + b1b, 0.f);
        }
      }
      __syncthreads();                             // T complete (the staging waves copy it out from here on)
      acc0 = f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = acc0;
      for (int kc = 0; kc < n2; ++kc, ++c) {
        const float* Ws = ST + (c & 1) * STG;
#pragma unroll
        for (int kk = 0; kk < FBK; kk += 16) {
          const float4 av = ld4(T + l15 * LDT + kc * FBK + kk + 4 * g4);
          const float4 b0 = ld4(Ws + (wave * 32 + l15) * FLDW + kk + 4 * g4);
          const float4 b1 = ld4(Ws + (wave * 32 + 16 + l15) * FLDW + kk + 4 * g4);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b0.x, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b1.x, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b0.y, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b1.y, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b0.z, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b1.z, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b0.w, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b1.w, acc1, 0, 0, 0);
        }
        __syncthreads();
      }
      float* Cs = ST;                              // C tile (the stage buffers are dead: the last chunk ended with a barrier)
      const int col = wave * 32 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Cs[(4 * g4 + r) * (FBN + 4) + col] = acc0[r];
        Cs[(4 * g4 + r) * (FBN + 4) + col + 16] = acc1[r];
      }
    } else {
    f32x16 acc;
    // ---- fc1: nb1 column blocks of 128, each nk1 chunks ----
    for (int cb = 0; cb < nb1; ++cb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      for (int kc = 0; kc < nk1; ++kc, ++c) {
        const float* Ws = ST + (c & 1) * STG;
#pragma unroll
        for (int kk = 0; kk < FBK; kk += 8) {
          const float4 av = ld4(Xs + l31 * LDX + kc * FBK + kk + 4 * hh);
          const float4 bv = ld4(Ws + (wave * 32 + l31) * FLDW + kk + 4 * hh);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
        }
        __syncthreads();
      }
      const int col = cb * FBN + wave * 32 + l31;
      const float b1 = a.b1[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * hh) * LDT + col] = fmaxf(acc[r] + b1, 0.f);
    }
    __syncthreads();                               // T complete (the staging waves copy it out from here on)
    // ---- fc2: one 128-column block, n2 chunks, A operand = T ----
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kc = 0; kc < n2; ++kc, ++c) {
      const float* Ws = ST + (c & 1) * STG;
#pragma unroll
      for (int kk = 0; kk < FBK; kk += 8) {
        const float4 av = ld4(T + l31 * LDT + kc * FBK + kk + 4 * hh);
        const float4 bv = ld4(Ws + (wave * 32 + l31) * FLDW + kk + 4 * hh);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
      }
      __syncthreads();
    }
    // C tile (the stage buffers are dead: the last chunk ended with a barrier)
    float* Cs = ST;
    const int col = wave * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) Cs[((r & 3) + 8 * (r >> 2) + 4 * hh) * (FBN + 4) + col] = acc[r];
    }   // !HALF
  }
  __syncthreads();
  // ---- row epilogue (8 waves x ER rows): out = C + b2 + x ----
  {
    const float* Cs = ST;
#pragma unroll
    for (int i = 0; i < ER; ++i) {
      const int lr = wave * ER + i, r = m0 + lr;
      if (!(con && r < M)) continue;
      const float4 v = ld4(Cs + lr * (FBN + 4) + c0);
      st4(a.out + (size_t)r * a.ldo + c0, make_float4(v.x + bias2.x + xres[i].x, v.y + bias2.y + xres[i].y,
                                                       v.z + bias2.z + xres[i].z, v.w + bias2.w + xres[i].w));
    }
  }
}

}  // namespace

extern "C" int dosx_ffn_supported(int H) { return (H % 32) == 0 && H >= 32 && H <= 128; }

extern "C" int dosx_ffn_fwd(const DosxFfn* ap, dosx_stream_t stream) {
  DOSX_CHECK_ARG(ap != nullptr, "dosx_ffn_fwd: null descriptor");
  const DosxFfn& a = *ap;
  if (a.M <= 0) return 0;
  DOSX_CHECK_ARG(dosx_ffn_supported(a.H), "dosx_ffn_fwd: H=%d unsupported (multiple of 32, <= 128)", a.H);
  DOSX_CHECK_ARG(a.x && a.stats && a.gamma && a.beta && a.w1 && a.b1 && a.w2 && a.b2 && a.h && a.out, "dosx_ffn_fwd: null operand");
  DOSX_CHECK_ARG((a.ldx & 3) == 0 && (a.ldh & 3) == 0 && (a.ldo & 3) == 0, "dosx_ffn_fwd: leading dimensions must be multiples of 4");
  const long long span = (const char*)a.w1 > (const char*)a.w2 ? (const char*)a.w1 - (const char*)a.w2 : (const char*)a.w2 - (const char*)a.w1;
  DOSX_CHECK_ARG(span + (long long)16 * a.H * a.H < 0x7fffffffLL, "dosx_ffn_fwd: fc1 / fc2 weights more than 2 GiB apart");
  const int H = a.H, H4 = 4 * H;
  static int half_max = -1;
  if (half_max < 0) { const char* e = getenv("DOSX_FFN_HALF_MAX"); half_max = e ? atoi(e) : 128; }
  const bool half = ceil_div(a.M, 32) <= half_max;       // 16-row workgroups while the 32-row grid is one partial round
  const int R = half ? 16 : 32;
  const size_t smem = sizeof(float) * ((size_t)R * (H + 4) + (size_t)R * (H4 + 4) + 2 * (size_t)FBN * FLDW);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  if (half) hipLaunchKernelGGL(ffn_fwd_kernel<true>, dim3(ceil_div(a.M, 16)), dim3(512), smem, to_stream(stream), a);
  else hipLaunchKernelGGL(ffn_fwd_kernel<false>, dim3(ceil_div(a.M, 32)), dim3(512), smem, to_stream(stream), a);
  DOSX_LAUNCH_CHECK();
  return 0;
}
