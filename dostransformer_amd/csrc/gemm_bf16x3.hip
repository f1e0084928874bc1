// Split-bf16 GEMM (round 6): C[M,N] = A[M,K] . W^T (+ bias) with fp32 operands and an fp32 result, computed on
// the bf16 matrix pipe - 16x the fp32-MFMA rate of gfx950 - from a three-way split of every operand element
//     x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)      (3 x 8 significant bits)
// and the six products whose weight is >= 2^-16 of the leading one
//     x.w ~ hi.hi + hi.mid + mid.hi + mid.mid + hi.lo + lo.hi
// accumulated in the MFMA's fp32 accumulators (the dropped terms are <= 2^-24 |x||w| per product: the size of ONE fp32 rounding
// of that product).  NOT the product path: the training / inference programs run the exact-fp32 kernels of gemm.hip; this kernel
// is measured next to them (bench.py `secondary.split_bf16`, tools/bench_kernels.py --what bf16x3) and checked against float64 at
// the SAME tolerance as dosx_gemm (tests/test_gpu_gemm.py).
//
// The split happens ON THE FLY in the staging path (global fp32 -> registers -> 3 x v_cvt_pk_bf16_f32 + 2 subtractions per pair
// -> three bf16 planes in LDS): the kernel takes the same fp32 buffers as dosx_gemm, HBM / L2 traffic is that of the fp32 kernel
// (4 bytes per element, not 6), and the vector ALU - idle next to the matrix pipe - pays for it (~3.5 lane-ops per element).
// Workgroup = 8 waves, wave-specialised like gemm.hip: 4 matrix waves (tile 128 x 128, wave: 64 x 64 = 2 x 2
// v_mfma_f32_32x32x16_bf16 tiles; per 16 k-columns 6 + 6 fragments by ds_read_b128 and 24 MFMAs = 768 cycles) + 4 staging waves
// (loads, split, LDS writes); 32 k-columns per stage, two stages.  LDS rows are padded to 40 bf16 (80 bytes): a quarter wave's
// 16 b128 reads then cover all 64 banks exactly once.  (First form, all four waves doing both jobs at 16 columns per stage:
// 25 % MFMA-busy - every wave serialised reads -> MFMAs -> conversions -> barrier; tools/exp/r6_pmc_bx.sh.)
// Weight layouts: 0 = [N][K] (nn.Linear: forward), 1 = [K][N] (the same matrix read k-major: input gradients) - transposed while
// it is written to LDS.
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

constexpr int BX_BN = 128, BX_BK = 16, BX_LDK = BX_BK + 8;

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// three bf16 planes of four consecutive fp32 values; pairwise, so that every conversion is one v_cvt_pk_bf16_f32 and the
// bf16 -> fp32 widening is a shift / a mask of the packed pair
struct Split3 { uint2 p[3]; };
__device__ __forceinline__ void split2(const float x0, const float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
  const float a0 = x0 - __builtin_bit_cast(float, h << 16), a1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
  m = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a0, a1}, bf16x2));
  const float b0 = a0 - __builtin_bit_cast(float, m << 16), b1 = a1 - __builtin_bit_cast(float, m & 0xffff0000u);
  l = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{b0, b1}, bf16x2));
}
__device__ __forceinline__ Split3 split4(const float4 v) {
  Split3 s;
  split2(v.x, v.y, s.p[0].x, s.p[1].x, s.p[2].x);
  split2(v.z, v.w, s.p[0].y, s.p[1].y, s.p[2].y);
  return s;
}

// WL = 0: W[n][k] (ldw >= K);  WL = 1: W[k][n] (ldw >= N)
// BM = 256: 512 threads = 8 waves as 4 (rows) x 2 (columns), each a 64 x 64 quadrant of the 256 x 128 tile, one workgroup per CU
// (108 KB of LDS); BM = 128: 4 waves, 128 x 128 tile, two workgroups per CU - one's tile prologue / C write under the other's
// products, for the short-K shapes.  Every wave stages AND multiplies.
template <int WL, int BM>
__global__ __launch_bounds__(2 * BM) void gemm_bf16x3_kernel(const float* __restrict__ A, const int lda, const float* __restrict__ W,
                                                          const int ldw, const float* __restrict__ bias, float* __restrict__ C,
                                                          const int ldc, const int M, const int N, const int K, const int act,
                                                          const float* __restrict__ res, const int ldres,
                                                          const float* __restrict__ mask, const int ldmask) {
  constexpr int BK = BX_BK, LDK = BX_LDK, NT = 2 * BM;
  constexpr int PA = BM * LDK, PW = BX_BN * LDK, STAGE = 3 * (PA + PW);      // one bf16 plane of the A / W tile; a stage = 3 + 3 planes
  constexpr int F4R = BK / 4;                       // float4 per tile row (4)
  constexpr int NA = BM * F4R / NT, NW = BX_BN * F4R / NT;              // float4 per thread: A 2, W 1 (BM 256) / 2 (BM 128)
  extern __shared__ __align__(16) unsigned char smraw[];
  __bf16* sm = reinterpret_cast<__bf16*>(smraw);
  const int tid = threadIdx.x, lane = tid & 63, r31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware tile order: consecutive workgroups land on different XCDs (round robin over 8); the column tiles of one row tile
  // share the A rows, so one XCD takes a contiguous run of tiles: tile index = (wg % 8) * per + wg / 8
  const int ntn = N / BX_BN, ntm = (M + BM - 1) / BM, ntiles = ntn * ntm;
  const int wg = blockIdx.x;
  const int per = (ntiles + 7) >> 3;
  const int t = (wg & 7) * per + (wg >> 3);
  if (t >= ntiles) return;                          // (grid is rounded up to a multiple of 8)
  const int tm = t / ntn, tn = t - tm * ntn;
  const int m0 = tm * BM, n0 = tn * BX_BN;
  const int nk = K / BK;

  // staged operands: TWO register sets - a stage is requested two iterations before its tile is multiplied
  float4 ra[2][NA], rw[2][NW];
  auto load = [&](float4(&qa)[NA], float4(&qw)[NW], const int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + NT * i, row = e / F4R, kq = (e % F4R) * 4;
      qa[i] = ld4(A + (size_t)min(m0 + row, M - 1) * lda + k0 + kq);
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int e = tid + NT * i;
      if constexpr (WL == 0) {
        const int row = e / F4R, kq = (e % F4R) * 4;
        qw[i] = ld4(W + (size_t)(n0 + row) * ldw + k0 + kq);
      } else {        // [K][N]: lanes along n (coalesced dwords), this thread's four consecutive k of column n = e % 128
        const int n = e & 127, kq = (e >> 7) * 4;
        const float* wp = W + (size_t)(k0 + kq) * ldw + n0 + n;
        qw[i] = make_float4(wp[0], wp[ldw], wp[2 * (size_t)ldw], wp[3 * (size_t)ldw]);
      }
    }
  };
  auto store_a = [&](const float4& q, const int i, const int buf) {
    const int e = tid + NT * i, row = e / F4R, kq = (e % F4R) * 4;
    const Split3 sp = split4(q);
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(sm + buf * STAGE + p * PA + row * LDK + kq) = sp.p[p];
  };
  auto store_w = [&](const float4& q, const int i, const int buf) {
    const int e = tid + NT * i;
    const int row = WL == 0 ? e / F4R : (e & 127), kq = WL == 0 ? (e % F4R) * 4 : (e >> 7) * 4;
    const Split3 sp = split4(q);
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(sm + buf * STAGE + 3 * PA + p * PW + row * LDK + kq) = sp.p[p];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const __bf16* fa = sm + (64 * wr + r31) * LDK + 8 * hh;                 // this lane's A fragment of plane 0, row tile 0
  const __bf16* fb = sm + 3 * PA + (64 * wc + r31) * LDK + 8 * hh;

  // one iteration: multiply LDS buffer `cur`; meanwhile convert the register set that holds the NEXT stage into the other buffer
  // (read in the previous iteration: released by its barrier) - the conversions sit BETWEEN the four quadrants' MFMA groups in
  // program order, so that a wave's vector-ALU work runs in the shadow of its own matrix instructions
  auto step = [&](const int cur, const bool more, const float4(&qa)[NA], const float4(&qw)[NW]) {
    const int bo = cur * STAGE;
    bf16x8 a[2][3], b[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        a[i][p] = *reinterpret_cast<const bf16x8*>(fa + bo + p * PA + 32 * i * LDK);
        b[i][p] = *reinterpret_cast<const bf16x8*>(fb + bo + p * PW + 32 * i * LDK);
      }
    auto quad = [&](const int i, const int j) {
      // small terms first: they meet an accumulator that has not yet taken this k-step's leading product
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
    };
    quad(0, 0);
    if (more) store_a(qa[0], 0, cur ^ 1);
    quad(0, 1);
    if (more) store_a(qa[1], 1, cur ^ 1);
    quad(1, 0);
    if (more) {
#pragma unroll
      for (int i = 0; i < NW; ++i) store_w(qw[i], i, cur ^ 1);
    }
    quad(1, 1);
  };

  load(ra[0], rw[0], 0);
  load(ra[1], rw[1], BK);
#pragma unroll
  for (int i = 0; i < NA; ++i) store_a(ra[0][i], i, 0);
#pragma unroll
  for (int i = 0; i < NW; ++i) store_w(rw[0][i], i, 0);
  __syncthreads();
  // iteration kt multiplies LDS buffer kt & 1; register set (kt + 1) & 1 holds stage kt + 1 (requested one iteration ago), set
  // kt & 1 is free (its stage is in LDS) and takes stage kt + 2.  Two iterations per trip so that the sets are compile-time, and
  // NO branch inside the trip (nk is even: K % 32 == 0; the last two iterations are peeled): with conditional loads / stores in
  // the body the compiler's s_waitcnt placement merges the paths conservatively and waits for the loads it has just issued
  // (vmcnt(2..0) where vmcnt(5..3) is meant) - the look-ahead is gone and every iteration pays the full load latency
  for (int kt = 0; kt < nk - 2; kt += 2) {
    load(ra[0], rw[0], (kt + 2) * BK);
    step(0, true, ra[1], rw[1]);
    __syncthreads();
    load(ra[1], rw[1], (kt + 3) * BK);
    step(1, true, ra[0], rw[0]);
    __syncthreads();
  }
  step(0, true, ra[1], rw[1]);
  __syncthreads();
  step(1, false, ra[0], rw[0]);
  // C: lane = column, 16 registers = rows (r & 3) + 8 (r >> 2) + 4 hh of the 32 x 32 tile.  The plain form (bias only) has its own
  // store loop: three per-element conditions in front of 64 stores per lane cost the K = 256 shapes 5 %.
  const bool plain = !act && !res && !mask;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + 64 * wc + 32 * j + r31;
    const float bv = bias ? bias[n] : 0.f;
    if (plain) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + 64 * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (m < M) C[(size_t)m * ldc + n] = acc[i][j][r] + bv;
        }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + 64 * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (m < M) {
            float v = acc[i][j][r] + bv;
            if (act) v = fmaxf(v, 0.f);
            if (res) v += res[(size_t)m * ldres + n];
            if (mask) v = mask[(size_t)m * ldmask + n] > 0.f ? v : 0.f;
            C[(size_t)m * ldc + n] = v;
          }
        }
    }
  }
}

template <int WL, int BM>
int launch_bf16x3(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M, int N, int K,
                  int act, const float* res, int ldres, const float* mask, int ldmask, hipStream_t s) {
  constexpr size_t smem = 2 * (size_t)(3 * (BM + BX_BN) * BX_LDK) * sizeof(__bf16);
  static_assert(smem <= 160 * 1024, "gemm_bf16x3_kernel: LDS");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x3_kernel<WL, BM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const int ntiles = (N / BX_BN) * ceil_div(M, BM);
  hipLaunchKernelGGL((gemm_bf16x3_kernel<WL, BM>), dim3(8 * ceil_div(ntiles, 8)), dim3(2 * BM), smem, s, A, lda, W, ldw, bias, C, ldc, M, N, K, act, res, ldres, mask, ldmask);
  DOSX_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int dosx_gemm_bf16x3_supported(int M, int N, int K) { return M > 0 && N > 0 && (N % 128) == 0 && K > 0 && (K % 32) == 0; }

// C[M,N] = epi( A[M,K] . op(W) + bias[N] );  w_layout 0: W is [N][K], 1: W is [K][N].  fp32 in, fp32 out, split-bf16 arithmetic
// (see top).  epi: relu when act == 1, then + res[M,N] when given, then zero where mask[M,N] <= 0 when given (the ReLU-mask
// epilogue of an input gradient: dosx_gemm's EPI_RELU_MASK) - what the feed-forward half of an encoder layer needs.
extern "C" int dosx_gemm_bf16x3(const float* A, int lda, const float* W, int ldw, int w_layout, const float* bias, float* C, int ldc,
                                int M, int N, int K, int act, const float* res, int ldres, const float* mask, int ldmask,
                                dosx_stream_t stream) {
  if (M <= 0) return 0;
  DOSX_CHECK_ARG(A && W && C, "dosx_gemm_bf16x3: null operand");
  DOSX_CHECK_ARG(dosx_gemm_bf16x3_supported(M, N, K), "dosx_gemm_bf16x3: N=%d must be a multiple of 128 and K=%d of 32", N, K);
  DOSX_CHECK_ARG(w_layout == 0 || w_layout == 1, "dosx_gemm_bf16x3: w_layout %d", w_layout);
  DOSX_CHECK_ARG((act == 0 || act == 1) && (!res || ldres >= N) && (!mask || ldmask >= N), "dosx_gemm_bf16x3: act 0 / 1, ldres / ldmask >= N");
  DOSX_CHECK_ARG((lda & 3) == 0 && (ldw & 3) == 0 && lda >= K && ldw >= (w_layout ? N : K) && ldc >= N &&
                     (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0,
                 "dosx_gemm_bf16x3: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
  hipStream_t s = to_stream(stream);
  // tile height: 128 rows (two workgroups per CU) while a problem's k range is short - the tile prologue and the C write of one
  // workgroup then run under the other's products; 256 rows (half the W traffic per flop) for long k ranges
  static int force = -1;
  if (force < 0) { const char* e = getenv("DOSX_BF16X3_BM"); force = e ? atoi(e) : 0; }
  const bool tall = force ? force == 256 : K > 512;
  if (tall) return w_layout ? launch_bf16x3<1, 256>(A, lda, W, ldw, bias, C, ldc, M, N, K, act, res, ldres, mask, ldmask, s) : launch_bf16x3<0, 256>(A, lda, W, ldw, bias, C, ldc, M, N, K, act, res, ldres, mask, ldmask, s);
  return w_layout ? launch_bf16x3<1, 128>(A, lda, W, ldw, bias, C, ldc, M, N, K, act, res, ldres, mask, ldmask, s) : launch_bf16x3<0, 128>(A, lda, W, ldw, bias, C, ldc, M, N, K, act, res, ldres, mask, ldmask, s);
}
