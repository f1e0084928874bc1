// fp32 MFMA GEMM family for the DOSTransformer hot path (gfx950).
//
//   dosx_gemm   : C[M,N] = epilogue( prologue(A)[M,K] . B )       (nn.Linear fwd / dgrad)
//   dosx_wgrad  : slab[s][N,K] = dY[ms:me]^T . prologue(A)[ms:me]  (nn.Linear wgrad, split over M)
//   dosx_reduce_partials : deterministic sum of the split slabs
//
// One workgroup = 4 waves (one per SIMD); v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma
// chain) so results track an fp32 torch reference to rounding.  A is gathered / concatenated /
// normalised on the fly while it is staged into LDS, so torch.cat([x[row], x[col], e]) and the
// LayerNorm/PReLU between the two Linear layers of every MLP never exist in HBM.
#include <stdlib.h>

#include "common.h"

// Diagnostic build only (-DDOSX_STAMPS, never shipped): wave 0 of a few workgroups records
// s_memtime at phase boundaries (see tools/stamp_gemm.py).
#ifdef DOSX_STAMPS
extern "C" { __device__ unsigned long long dosx_stamp_buf[64 * 64]; }
#define STAMP(slot)                                                                              \
  do {                                                                                           \
    if ((threadIdx.x == 0) && (blockIdx.y == 0) && (blockIdx.x % 37 == 0) && (blockIdx.x / 37) < 64 && (slot) < 64) \
      dosx_stamp_buf[(blockIdx.x / 37) * 64 + (slot)] = __builtin_amdgcn_s_memtime();            \
  } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

namespace {

constexpr int BM = 32;
constexpr int BK = 32;
constexpr int LDA = BK + 4;  // 36 floats: 16-B aligned rows, conflict-free ds_read_b128

struct GemmLaunch {
  DosxGemm g;
  int vecA;
  int vecW;
  int dma;   // W tile by LDS-DMA (needs K % 32 == 0: no k tail to zero-fill)
  int rt;    // 32-row blocks per workgroup (1 or 2)
};

__device__ __forceinline__ float prelu_f(float v, float a) { return v >= 0.f ? v : a * v; }

// ---- A-operand staging: this thread owns row `arow` of the tile and 4 consecutive k ----------
struct AState {
  const float* rp[3];
  int w0, w01;
  float mean, rstd, alpha;
  bool ok;
};

// `gm` must be a VALID row (callers clamp it); `ok` says whether the row's data is used or zeroed.
// Loads are issued unconditionally from valid addresses and masked afterwards: a branch around a
// global load makes hipcc wait vmcnt(0) per load and serialises the L2 round trips.
__device__ __forceinline__ void a_state_init(AState& st, const DosxSeg* segs, int nseg, int pro,
                                              const float* pro_stats, const float* pro_alpha, int gm,
                                              bool ok) {
  st.ok = ok;
  st.rp[0] = segs[0].p + (size_t)dosx_map_row(segs[0].map, gm) * (size_t)segs[0].ld;
  st.rp[1] = st.rp[2] = st.rp[0];
  if (nseg > 1) st.rp[1] = segs[1].p + (size_t)dosx_map_row(segs[1].map, gm) * (size_t)segs[1].ld;
  if (nseg > 2) st.rp[2] = segs[2].p + (size_t)dosx_map_row(segs[2].map, gm) * (size_t)segs[2].ld;
  st.w0 = segs[0].width;
  st.w01 = nseg > 1 ? st.w0 + segs[1].width : 0x7fffffff;
  if (nseg == 1) st.w0 = 0x7fffffff;
  st.mean = 0.f;
  st.rstd = 0.f;
  st.alpha = 0.f;
  if (pro == DOSX_PRO_ROWLN) {
    st.mean = pro_stats[2 * (size_t)gm];
    st.rstd = pro_stats[2 * (size_t)gm + 1];
  }
  if (pro == DOSX_PRO_PRELU || pro == DOSX_PRO_LN_PRELU) st.alpha = *pro_alpha;
}

template <int PRO>
__device__ __forceinline__ float a_xform1(const AState& st, float v, int k, const float* gamma, const float* beta) {
  if (PRO == DOSX_PRO_PRELU) return prelu_f(v, st.alpha);
  if (PRO == DOSX_PRO_LN_PRELU) return prelu_f(v * gamma[k] + beta[k], st.alpha);
  if (PRO == DOSX_PRO_ROWLN) return (v - st.mean) * st.rstd * gamma[k] + beta[k];
  return v;
}

// PRO and VEC are compile-time so that the staging code is straight-line.  Staging is split in two:
// a_issue() only ISSUES the global loads (unconditionally, from clamped = always valid addresses);
// a_finish() applies the prologue transform and the out-of-range mask and runs one k-chunk later,
// right before the LDS store — so the loads stay in flight across the MFMA block of the current
// chunk (a select placed right after a load makes hipcc wait for it before the MFMAs).
struct ARaw {
  float4 v, g, b;
};

template <int PRO, int VEC>
__device__ __forceinline__ ARaw a_issue(const AState& st, int k, int K, const float* gamma, const float* beta) {
  ARaw r;
  r.v = f4zero(); r.g = f4zero(); r.b = f4zero();
  if (VEC) {
    const int kc = (k < K) ? k : 0;
    const float* p0 = st.rp[0] + kc;
    const float* p1 = st.rp[1] + (kc - st.w0);
    const float* p2 = st.rp[2] + (kc - st.w01);
    const float* p = (kc < st.w0) ? p0 : ((kc < st.w01) ? p1 : p2);
    r.v = ld4(p);
    if (PRO == DOSX_PRO_LN_PRELU || PRO == DOSX_PRO_ROWLN) {
      r.g = ld4(gamma + kc);
      r.b = ld4(beta + kc);
    }
  } else {
    if (!st.ok || k >= K) return r;
    float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int kk = k + i;
      if (kk < K) {
        float x;
        if (kk < st.w0) x = st.rp[0][kk];
        else if (kk < st.w01) x = st.rp[1][kk - st.w0];
        else x = st.rp[2][kk - st.w01];
        t[i] = a_xform1<PRO>(st, x, kk, gamma, beta);
      }
    }
    r.v = make_float4(t[0], t[1], t[2], t[3]);
  }
  return r;
}

template <int PRO, int VEC>
__device__ __forceinline__ float4 a_finish(const AState& st, const ARaw& r, int k, int K) {
  float4 v = r.v;
  if (!VEC) return v;
  if (PRO == DOSX_PRO_PRELU) {
    v.x = prelu_f(v.x, st.alpha); v.y = prelu_f(v.y, st.alpha);
    v.z = prelu_f(v.z, st.alpha); v.w = prelu_f(v.w, st.alpha);
  } else if (PRO == DOSX_PRO_LN_PRELU) {
    v.x = prelu_f(v.x * r.g.x + r.b.x, st.alpha); v.y = prelu_f(v.y * r.g.y + r.b.y, st.alpha);
    v.z = prelu_f(v.z * r.g.z + r.b.z, st.alpha); v.w = prelu_f(v.w * r.g.w + r.b.w, st.alpha);
  } else if (PRO == DOSX_PRO_ROWLN) {
    v.x = (v.x - st.mean) * st.rstd * r.g.x + r.b.x; v.y = (v.y - st.mean) * st.rstd * r.g.y + r.b.y;
    v.z = (v.z - st.mean) * st.rstd * r.g.z + r.b.z; v.w = (v.w - st.mean) * st.rstd * r.g.w + r.b.w;
  }
  if (!(st.ok && k < K)) v = f4zero();
  return v;
}

// ---------------------------------------------------------------------------------------------
// C = prologue(A) . B  with fused row-wise epilogues.   NTW = 32-wide MFMA tiles per wave,
// BN = 128*NTW columns per workgroup.  WL: 0 = W[N,K] (k-contiguous), 1 = W[K,N] (n-contiguous).
// ---------------------------------------------------------------------------------------------
// DMA = 1: the W tile (the bulk of the staged bytes) goes global -> LDS directly
// (__builtin_amdgcn_global_load_lds, 16 B per lane: no VGPR staging, no ds_write, no masks), double
// buffered so the DMA of chunk k+1 flies under the MFMAs of chunk k.  The destination of one
// wave-instruction is lane-linear (1 KiB), so the W[N,K] tile is stored with UNPADDED 128-B rows and
// the 16-B chunk index XOR-swizzled with (row>>1)&7 on the SOURCE address (and on the fragment read)
// to keep ds_read_b128 conflict-free; the W[K,N] tile is read with ds_read_b32 along n and needs none.
// RT = 32-row blocks per workgroup (BM = 32*RT).  The per-CU global->LDS rate (~10-25 B/clk measured,
// W comes from L2) - not the MFMA pipe - bounds a 32-row tile: every k-chunk re-streams the whole
// [BN x 32] W slice for only 32 rows (18 B per MFMA-clk at BN=256).  RT = 2 halves the bytes per flop.
template <int RT, int NTW, int WL, int PRO, int VEC, int EPI, int DMA>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmLaunch L) {
  const DosxGemm& g = L.g;
  constexpr int BMR = BM * RT;
  constexpr int BN = 128 * NTW;
  constexpr int LDWT = DMA ? ((WL == 0) ? BK : BN) : ((WL == 0) ? (BK + 4) : (BN + 4));
  constexpr int WROWS = (WL == 0) ? BN : BK;
  constexpr int LDC = BN + 4;
  constexpr int STAGE = BMR * LDA + WROWS * LDWT;     // floats of one staging buffer (A tile + W tile)
  constexpr int CTILE = BM * LDC;
  constexpr int MAINF = DMA ? 2 * STAGE : (STAGE > CTILE ? STAGE : CTILE);
  static_assert(!DMA || 2 * STAGE >= CTILE, "C tile must fit in the staging buffers");
  constexpr int CG = (BN + 255) / 256;   // float4 column groups per lane in the row-wise epilogue
  constexpr int NW4 = BN / 32;           // float4 W loads per thread per k-chunk

  extern __shared__ __align__(16) float smem[];
  float* As = smem;
  float* Ws = smem + BMR * LDA;
  float* Cs = smem;
  float* Ps = smem + CTILE;              // [4][2][BN] + 4 (behind the C tile; staging memory is dead by then)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int m0 = blockIdx.x * BMR, n0 = blockIdx.y * BN;
  const int M = g.M, N = g.N, K = g.K;

  STAMP(0);
  // ---- staging setup ----
  const int arow = tid >> 3, akq = (tid & 7) * 4;
  AState ast[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
    a_state_init(ast[r], g.a, g.nseg, PRO, g.pro_stats, g.pro_alpha, min(m0 + arow + 32 * r, M - 1),
                 (m0 + arow + 32 * r) < M);
  auto issueA = [&](ARaw(&ar)[RT], int k0) {
#pragma unroll
    for (int r = 0; r < RT; ++r) ar[r] = a_issue<PRO, VEC>(ast[r], k0 + akq, K, g.pro_gamma, g.pro_beta);
  };
  auto storeA = [&](float* Ad, const ARaw(&ar)[RT], int k0) {
#pragma unroll
    for (int r = 0; r < RT; ++r)
      st4(&Ad[(arow + 32 * r) * LDA + akq], a_finish<PRO, VEC>(ast[r], ar[r], k0 + akq, K));
  };

  auto loadW = [&](int k0, float4(&wr)[NW4]) {
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
      float4 v = f4zero();
      if (WL == 0) {
        const int n = n0 + (tid >> 3) + 32 * i, k = k0 + (tid & 7) * 4;
        if (VEC) {   // unconditional load from a clamped (valid) address; masked in storeW
          v = ld4(g.w + (size_t)min(n, N - 1) * g.ldw + (k < K ? k : 0));
        } else if (n < N && k < K) {
          const float* p = g.w + (size_t)n * g.ldw + k;
          v.x = p[0];
          if (k + 1 < K) v.y = p[1];
          if (k + 2 < K) v.z = p[2];
          if (k + 3 < K) v.w = p[3];
        }
      } else {
        const int lin = tid + 256 * i;
        const int r = lin / (BN / 4), c4 = (lin % (BN / 4)) * 4;
        const int k = k0 + r, n = n0 + c4;
        if (VEC) {
          v = ld4(g.w + (size_t)min(k, K - 1) * g.ldw + (n < N ? n : 0));
        } else if (k < K && n < N) {
          const float* p = g.w + (size_t)k * g.ldw + n;
          v.x = p[0];
          if (n + 1 < N) v.y = p[1];
          if (n + 2 < N) v.z = p[2];
          if (n + 3 < N) v.w = p[3];
        }
      }
      wr[i] = v;
    }
  };
  auto storeW = [&](float* Wd, int k0, const float4(&wr)[NW4], int i0, int i1) {
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
      if (i < i0 || i >= i1) continue;
      float4 v = wr[i];
      if (WL == 0) {
        if (VEC && !((n0 + (tid >> 3) + 32 * i) < N && (k0 + (tid & 7) * 4) < K)) v = f4zero();
        st4(&Wd[((tid >> 3) + 32 * i) * LDWT + (tid & 7) * 4], v);
      } else {
        const int lin = tid + 256 * i;
        if (VEC && !((k0 + lin / (BN / 4)) < K && (n0 + (lin % (BN / 4)) * 4) < N)) v = f4zero();
        st4(&Wd[(lin / (BN / 4)) * LDWT + (lin % (BN / 4)) * 4], v);
      }
    }
  };

  f32x16 acc[RT][NTW];
#pragma unroll
  for (int rr = 0; rr < RT; ++rr)
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rr][t][r] = 0.f;

  // wave-uniform: how many of this wave's 32-column tiles intersect [0, N)
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  int tiles_on = (N - n0 - wave_u * NTW * 32 + 31) / 32;
  tiles_on = tiles_on < 0 ? 0 : (tiles_on > NTW ? NTW : tiles_on);

  const int nk = (K + BK - 1) / BK;

  // One k-chunk of MFMAs from a staging buffer.  Pipelined variant: ALL fragment reads of the chunk are
  // issued first, then `between()` (the LDS stores of the NEXT chunk into the other buffer), then the
  // MFMA chain - so the stores drain through the LDS pipe underneath the matrix work.
  // B-operand addressing (see the DMA note above)
  auto wfrag4 = [&](const float* Wsb, int col, int kq /* k offset, multiple of 4 */) -> float4 {
    if (DMA) return ld4(&Wsb[col * BK + ((((kq >> 2) ^ ((col >> 1) & 7))) << 2)]);
    return ld4(&Wsb[col * LDWT + kq]);
  };

  // One k-chunk of MFMAs from a staging buffer.
  auto compute = [&](const float* Asb, const float* Wsb) {
    if (tiles_on == NTW) {
      // fast path (every tile of this wave is inside N): straight-line, so hipcc hoists the ds_reads
#pragma unroll
      for (int kk = 0; kk < BK; kk += 8) {
        float4 a[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) a[r] = ld4(&Asb[(32 * r + l31) * LDA + kk + 4 * hh]);
        float b[NTW][4];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          if (WL == 0) {
            const float4 v = wfrag4(Wsb, (wave * NTW + t) * 32 + l31, kk + 4 * hh);
            b[t][0] = v.x; b[t][1] = v.y; b[t][2] = v.z; b[t][3] = v.w;
          } else {
            const float* bp = &Wsb[(kk + 4 * hh) * LDWT + (wave * NTW + t) * 32 + l31];
            b[t][0] = bp[0]; b[t][1] = bp[LDWT]; b[t][2] = bp[2 * LDWT]; b[t][3] = bp[3 * LDWT];
          }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int r = 0; r < RT; ++r) {
            const float av = c == 0 ? a[r].x : c == 1 ? a[r].y : c == 2 ? a[r].z : a[r].w;
#pragma unroll
            for (int t = 0; t < NTW; ++t)
              acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[t][c], acc[r][t], 0, 0, 0);
          }
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < BK; kk += 8) {
        float4 a[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) a[r] = ld4(&Asb[(32 * r + l31) * LDA + kk + 4 * hh]);
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          if (t >= tiles_on) continue;
          const int col = (wave * NTW + t) * 32 + l31;
          float b[4];
          if (WL == 0) {
            const float4 v = wfrag4(Wsb, col, kk + 4 * hh);
            b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
          } else {
            const float* bp = &Wsb[(kk + 4 * hh) * LDWT + col];
            b[0] = bp[0]; b[1] = bp[LDWT]; b[2] = bp[2 * LDWT]; b[3] = bp[3 * LDWT];
          }
#pragma unroll
          for (int r = 0; r < RT; ++r) {
            acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].x, b[0], acc[r][t], 0, 0, 0);
            acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].y, b[1], acc[r][t], 0, 0, 0);
            acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].z, b[2], acc[r][t], 0, 0, 0);
            acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].w, b[3], acc[r][t], 0, 0, 0);
          }
        }
      }
    }
  };

  STAMP(1);
  if constexpr (DMA) {
    // W tile of chunk k0 -> LDS buffer Wd, asynchronously.  Thread/lane mapping is the register path's
    // (row = tid/8 + 32 i, 16-B chunk = tid%8 | float4 index = tid + 256 i), which makes every
    // wave-instruction's 64 x 16 B land contiguously at a wave-uniform LDS base.
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    auto dmaW = [&](float* Wd, int k0) {
#pragma unroll
      for (int i = 0; i < NW4; ++i) {
        const float* src;
        float* dst;
        if (WL == 0) {
          const int row = (tid >> 3) + 32 * i;                    // tile-local output column
          const int c = (tid & 7) ^ ((row >> 1) & 7);             // swizzled source chunk
          src = g.w + (size_t)min(n0 + row, N - 1) * g.ldw + k0 + c * 4;
          dst = Wd + (8 * wave_s + 32 * i) * BK;
        } else {
          const int lin = tid + 256 * i;
          const int r = lin / (BN / 4), c4 = (lin % (BN / 4)) * 4;
          const int n = n0 + c4;
          src = g.w + (size_t)min(k0 + r, K - 1) * g.ldw + (n < N ? n : 0);
          dst = Wd + (64 * wave_s + 256 * i) * 4;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    };
    ARaw araw[RT];
    issueA(araw, 0);
    dmaW(smem + BMR * LDA, 0);
    storeA(smem, araw, 0);
    __syncthreads();        // (emits vmcnt(0) while a DMA is in flight: the tile has landed)
    for (int kt = 0; kt < nk; ++kt) {
      float* Acur = smem + (kt & 1) * STAGE;
      float* Anxt = smem + ((kt & 1) ^ 1) * STAGE;
      if (kt + 1 < nk) {
        issueA(araw, (kt + 1) * BK);
        dmaW(Anxt + BMR * LDA, (kt + 1) * BK);
      }
      STAMP(3 + 3 * kt);
      compute(Acur, Acur + BMR * LDA);
      STAMP(4 + 3 * kt);
      if (kt + 1 < nk) storeA(Anxt, araw, (kt + 1) * BK);
      __syncthreads();
    }
  } else {
    ARaw araw[RT];
    issueA(araw, 0);
    float4 wreg[NW4];
    loadW(0, wreg);
    for (int kt = 0; kt < nk; ++kt) {
      storeA(As, araw, kt * BK);
      storeW(Ws, kt * BK, wreg, 0, NW4);
      STAMP(2 + 3 * kt);
      __syncthreads();
      STAMP(3 + 3 * kt);
      if (kt + 1 < nk) {
        issueA(araw, (kt + 1) * BK);
        loadW((kt + 1) * BK, wreg);
      }
      compute(As, Ws);
      STAMP(4 + 3 * kt);
      __syncthreads();
    }
  }
  STAMP(55);

  // ---- epilogue operand prefetch -----------------------------------------------------------------
  // Every global operand of the row-wise epilogue (bias / gamma / beta vectors, the rows of `aux` and
  // `res`, the per-row statistics) is loaded HERE, unconditionally and from always-valid addresses
  // (absent operands alias `out`, whose values are then ignored), before the accumulator round trip
  // through LDS.  The row loop below then touches no global memory except its stores; a load inside
  // that loop costs one exposed L2/HBM round trip per row (8 per wave).
  const int ncols = min(BN, N - n0);
  const float invN = 1.f / (float)N;
  constexpr int epi = EPI;     // compile-time: only this epilogue's code exists in the kernel
  const bool is_prelu_ln = (epi == DOSX_EPI_PRELU_LN_BWD), is_rowln = (epi == DOSX_EPI_ROWLN_BWD);
  const bool aux_first = (epi != DOSX_EPI_BIAS_ACT && epi != DOSX_EPI_LN);
  const float* dummy = g.out;
  const float* p1 = aux_first ? (g.aux ? g.aux : dummy) : (g.res ? g.res : dummy);   // per-row operand 1
  const int ld1 = aux_first ? (g.aux ? g.ldaux : 0) : (g.res ? g.ldr : 0);
  const float* p2 = (is_rowln && g.res) ? g.res : dummy;                                // per-row operand 2
  const int ld2 = (is_rowln && g.res) ? g.ldr : 0;
  const float* statp = g.aux_stats ? g.aux_stats : dummy;
  float4 pg[CG], pb[CG];   // column partial sums (dgamma, dbeta) over all row blocks of this workgroup
  float pal = 0.f;         // dalpha partial
#pragma unroll
  for (int j = 0; j < CG; ++j) { pg[j] = f4zero(); pb[j] = f4zero(); }
  float e_alpha = 0.f;
  if (epi == DOSX_EPI_PRELU_LN_BWD || epi == DOSX_EPI_PRELU_BWD) e_alpha = *g.epi_alpha;
  bool on[CG];
  int gcol[CG];
  float4 biasv[CG], gamv[CG], betv[CG];
#pragma unroll
  for (int j = 0; j < CG; ++j) {
    const int c = lane * 4 + 256 * j;
    on[j] = c < ncols;
    gcol[j] = n0 + (on[j] ? c : 0);
    biasv[j] = ld4((g.bias ? g.bias : dummy) + gcol[j]);
    gamv[j] = ld4((g.epi_gamma ? g.epi_gamma : dummy) + gcol[j]);
    betv[j] = ld4((g.epi_beta ? g.epi_beta : dummy) + gcol[j]);
    if (!g.bias) biasv[j] = f4zero();
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {        // one 32-row block at a time through the LDS C tile
  const int mb = m0 + 32 * rt;
  float4 pv1[8][CG], pv2[8][CG];
  float st0[8], st1[8];
  size_t orow_[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int rc = min(mb + wave * 8 + i, M - 1);
    const size_t r1 = (size_t)((!aux_first && g.res) ? dosx_map_row(g.res_map, rc) : rc) * ld1;
    const size_t r2 = (size_t)((is_rowln && g.res) ? dosx_map_row(g.res_map, rc) : rc) * ld2;
    orow_[i] = (size_t)(epi == DOSX_EPI_BIAS_ACT ? dosx_map_row(g.out_map, rc) : rc) * g.ldo;
#pragma unroll
    for (int j = 0; j < CG; ++j) {
      pv1[i][j] = ld4(p1 + r1 + gcol[j]);
      pv2[i][j] = ld4(p2 + r2 + gcol[j]);
    }
    st0[i] = statp[is_rowln ? 2 * (size_t)rc : (size_t)rc];
    st1[i] = statp[is_rowln ? 2 * (size_t)rc + 1 : (size_t)rc];
  }

  // ---- accumulators -> LDS C tile (aliases the staging buffers; loop ended with a barrier) ----
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int col = (wave * NTW + t) * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
      Cs[row * LDC + col] = acc[rt][t][r];
    }
  }
  STAMP(56);
  __syncthreads();
  STAMP(57);

  // ---- row-wise epilogue: wave w owns rows 8w..8w+7, lanes sweep the columns as float4 --------
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int lr = wave * 8 + i, r = mb + lr;
    const bool rvalid = r < M;              // wave-uniform
    float4 v[CG];
#pragma unroll
    for (int j = 0; j < CG; ++j) v[j] = on[j] ? ld4(&Cs[lr * LDC + lane * 4 + 256 * j]) : f4zero();
    float* const orow = g.out + orow_[i];
    if (epi == DOSX_EPI_BIAS_ACT) {
      float s1 = 0.f;
#pragma unroll
      for (int j = 0; j < CG; ++j) {
        if (!on[j]) continue;
        v[j] = f4add(v[j], biasv[j]);
        if (g.act == 1) {
          v[j].x = fmaxf(v[j].x, 0.f); v[j].y = fmaxf(v[j].y, 0.f);
          v[j].z = fmaxf(v[j].z, 0.f); v[j].w = fmaxf(v[j].w, 0.f);
        } else if (g.act == 2) {
          const float sl = g.act_slope;
          v[j].x = v[j].x >= 0.f ? v[j].x : sl * v[j].x; v[j].y = v[j].y >= 0.f ? v[j].y : sl * v[j].y;
          v[j].z = v[j].z >= 0.f ? v[j].z : sl * v[j].z; v[j].w = v[j].w >= 0.f ? v[j].w : sl * v[j].w;
        }
        if (g.res) v[j] = f4add(v[j], pv1[i][j]);
        if (rvalid) st4(orow + gcol[j], v[j]);
        s1 += v[j].x + v[j].y + v[j].z + v[j].w;
      }
      if (g.stats_out) {
        const float mean = wave_sum(s1) * invN;
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < CG; ++j) {
          if (!on[j]) continue;
          const float a = v[j].x - mean, b = v[j].y - mean, c2 = v[j].z - mean, d = v[j].w - mean;
          s2 += a * a + b * b + c2 * c2 + d * d;
        }
        const float var = wave_sum(s2) * invN;
        if (lane == 0 && rvalid) {
          g.stats_out[2 * (size_t)r] = mean;
          g.stats_out[2 * (size_t)r + 1] = rsqrtf(var + DOSX_LN_EPS);
        }
      }
    } else if (epi == DOSX_EPI_LN) {
      float s1 = 0.f;
#pragma unroll
      for (int j = 0; j < CG; ++j) {
        if (!on[j]) continue;
        v[j] = f4add(v[j], biasv[j]);
        s1 += v[j].x + v[j].y + v[j].z + v[j].w;
      }
      const float mean = wave_sum(s1) * invN;
      float s2 = 0.f;
#pragma unroll
      for (int j = 0; j < CG; ++j) {
        if (!on[j]) continue;
        v[j].x -= mean; v[j].y -= mean; v[j].z -= mean; v[j].w -= mean;
        s2 += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
      }
      const float rstd = rsqrtf(wave_sum(s2) * invN + DOSX_LN_EPS);
#pragma unroll
      for (int j = 0; j < CG; ++j) {
        if (!on[j] || !rvalid) continue;
        st4(orow + gcol[j], make_float4(v[j].x * rstd, v[j].y * rstd, v[j].z * rstd, v[j].w * rstd));
      }
      if (lane == 0 && rvalid) g.aux_out[r] = rstd;
    } else if (epi == DOSX_EPI_RELU_MASK) {
#pragma unroll
      for (int j = 0; j < CG; ++j) {
        if (!on[j] || !rvalid) continue;
        const float4 h = pv1[i][j];
        st4(orow + gcol[j], make_float4(h.x > 0.f ? v[j].x : 0.f, h.y > 0.f ? v[j].y : 0.f,
                                        h.z > 0.f ? v[j].z : 0.f, h.w > 0.f ? v[j].w : 0.f));
      }
    } else if (epi == DOSX_EPI_PRELU_BWD) {
#pragma unroll
      for (int j = 0; j < CG; ++j) {
        if (!on[j] || !rvalid) continue;
        const float4 z = pv1[i][j];
        float4 o;
        o.x = z.x >= 0.f ? v[j].x : e_alpha * v[j].x; if (z.x < 0.f) pal += v[j].x * z.x;
        o.y = z.y >= 0.f ? v[j].y : e_alpha * v[j].y; if (z.y < 0.f) pal += v[j].y * z.y;
        o.z = z.z >= 0.f ? v[j].z : e_alpha * v[j].z; if (z.z < 0.f) pal += v[j].z * z.z;
        o.w = z.w >= 0.f ? v[j].w : e_alpha * v[j].w; if (z.w < 0.f) pal += v[j].w * z.w;
        st4(orow + gcol[j], o);
      }
    } else {  // DOSX_EPI_PRELU_LN_BWD or DOSX_EPI_ROWLN_BWD : LayerNorm backward over the full row
      const float mean = is_prelu_ln ? 0.f : st0[i];
      const float rstd = is_prelu_ln ? st0[i] : st1[i];
      float4 xh[CG], dxh[CG];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < CG; ++j) {
        xh[j] = f4zero(); dxh[j] = f4zero();
        if (!on[j] || !rvalid) continue;
        float4 a = pv1[i][j];
        const float4 gm = gamv[j];
        float4 dy = v[j];
        if (is_prelu_ln) {
          const float4 bt = betv[j];
          const float y0 = a.x * gm.x + bt.x, y1 = a.y * gm.y + bt.y, y2 = a.z * gm.z + bt.z,
                      y3 = a.w * gm.w + bt.w;
          if (y0 < 0.f) { pal += dy.x * y0; dy.x *= e_alpha; }
          if (y1 < 0.f) { pal += dy.y * y1; dy.y *= e_alpha; }
          if (y2 < 0.f) { pal += dy.z * y2; dy.z *= e_alpha; }
          if (y3 < 0.f) { pal += dy.w * y3; dy.w *= e_alpha; }
        } else {
          a.x = (a.x - mean) * rstd; a.y = (a.y - mean) * rstd;
          a.z = (a.z - mean) * rstd; a.w = (a.w - mean) * rstd;
        }
        xh[j] = a;
        pg[j].x += dy.x * a.x; pg[j].y += dy.y * a.y; pg[j].z += dy.z * a.z; pg[j].w += dy.w * a.w;
        pb[j] = f4add(pb[j], dy);
        dxh[j] = make_float4(dy.x * gm.x, dy.y * gm.y, dy.z * gm.z, dy.w * gm.w);
        s1 += dxh[j].x + dxh[j].y + dxh[j].z + dxh[j].w;
        s2 += dxh[j].x * a.x + dxh[j].y * a.y + dxh[j].z * a.z + dxh[j].w * a.w;
      }
      const float m1 = wave_sum(s1) * invN, m2 = wave_sum(s2) * invN;
#pragma unroll
      for (int j = 0; j < CG; ++j) {
        if (!on[j] || !rvalid) continue;
        float4 o = make_float4(rstd * (dxh[j].x - m1 - xh[j].x * m2), rstd * (dxh[j].y - m1 - xh[j].y * m2),
                               rstd * (dxh[j].z - m1 - xh[j].z * m2), rstd * (dxh[j].w - m1 - xh[j].w * m2));
        if (is_rowln && g.res) o = f4add(o, pv2[i][j]);
        st4(orow + gcol[j], o);
      }
    }
  }

  if (rt + 1 < RT) __syncthreads();     // the C tile is rewritten by the next row block
  }   // rt
  STAMP(58);
  // ---- per-workgroup partial sums for the parameter gradients of the fused LN / PReLU ----------
  if (g.partials && (epi == DOSX_EPI_PRELU_LN_BWD || epi == DOSX_EPI_ROWLN_BWD || epi == DOSX_EPI_PRELU_BWD)) {
    float* prow = g.partials + (size_t)(blockIdx.x * gridDim.y + blockIdx.y) * g.partial_ld;
    if (epi != DOSX_EPI_PRELU_BWD) {
#pragma unroll
      for (int j = 0; j < CG; ++j) {
        const int c = lane * 4 + 256 * j;
        if (c < BN) {
          st4(&Ps[(wave * 2 + 0) * BN + c], pg[j]);
          st4(&Ps[(wave * 2 + 1) * BN + c], pb[j]);
        }
      }
    }
    if (epi != DOSX_EPI_ROWLN_BWD) {
      const float s = wave_sum(pal);
      if (lane == 0) Ps[8 * BN + wave] = s;
    }
    __syncthreads();
    if (epi != DOSX_EPI_PRELU_BWD) {
      for (int c = tid; c < 2 * BN; c += 256) {
        const int which = c / BN, col = c % BN;
        if (col < ncols) {
          const float s = Ps[(0 * 2 + which) * BN + col] + Ps[(1 * 2 + which) * BN + col] +
                          Ps[(2 * 2 + which) * BN + col] + Ps[(3 * 2 + which) * BN + col];
          prow[which * N + n0 + col] = s;
        }
      }
    }
    if (epi != DOSX_EPI_ROWLN_BWD && tid == 0)
      prow[g.partial_ld - 1] = Ps[8 * BN] + Ps[8 * BN + 1] + Ps[8 * BN + 2] + Ps[8 * BN + 3];
  }
}

template <int RT, int NTW, int WL, int DMA>
constexpr size_t gemm_smem_bytes() {
  constexpr int BN = 128 * NTW;
  constexpr int LDWT = DMA ? ((WL == 0) ? BK : BN) : ((WL == 0) ? (BK + 4) : (BN + 4));
  constexpr int WROWS = (WL == 0) ? BN : BK;
  constexpr int STAGE = BM * RT * LDA + WROWS * LDWT;
  constexpr int CTILE = BM * (BN + 4);
  constexpr int MAINF = DMA ? 2 * STAGE : (STAGE > CTILE ? STAGE : CTILE);
  constexpr int EPIF = CTILE + 8 * BN + 4;
  return (size_t)(MAINF > EPIF ? MAINF : EPIF) * sizeof(float);
}

template <int RT, int NTW, int WL, int PRO, int VEC, int EPI, int DMA>
int launch_gemm3(const GemmLaunch& L, hipStream_t s) {
  constexpr int BN = 128 * NTW;
  dim3 grid(ceil_div(L.g.M, BM * RT), ceil_div(L.g.N, BN));
  constexpr size_t smem = gemm_smem_bytes<RT, NTW, WL, DMA>();
  static_assert(smem <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<RT, NTW, WL, PRO, VEC, EPI, DMA>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_kernel<RT, NTW, WL, PRO, VEC, EPI, DMA>), grid, dim3(256), smem, s, L);
  DOSX_LAUNCH_CHECK();
  return 0;
}

template <int NTW, int WL, int PRO, int VEC, int EPI, int DMA>
int launch_gemm2(const GemmLaunch& L, hipStream_t s) {
  if (L.rt == 2) return launch_gemm3<2, NTW, WL, PRO, VEC, EPI, DMA>(L, s);
  return launch_gemm3<1, NTW, WL, PRO, VEC, EPI, DMA>(L, s);
}

// The LDS-DMA staging variant is compiled only with -DDOSX_ENABLE_DMA (experiments): measured on
// MI355X it loses to the register-staged path at every shape of this workload (2.78 vs 2.68 ms per
// cfg2 step; 73 vs 97 TF/s at M=262144,N=256,K=384) because its two staging buffers halve the
// number of co-resident workgroups, which is what actually hides the load latency here.
template <int NTW, int WL, int PRO, int VEC, int EPI>
int launch_gemm(const GemmLaunch& L, hipStream_t s) {
#ifdef DOSX_ENABLE_DMA
  if (VEC && L.dma) return launch_gemm2<NTW, WL, PRO, VEC, EPI, VEC ? 1 : 0>(L, s);
#endif
  return launch_gemm2<NTW, WL, PRO, VEC, EPI, 0>(L, s);
}

// Only the (layout, prologue, epilogue) combinations the forward / backward programs use exist.
template <int NTW>
int dispatch_gemm(const GemmLaunch& L, hipStream_t s) {
  const int vec = L.vecA && L.vecW;
  const int pro = L.g.pro, epi = L.g.epi;
  if (!vec) {
    // generic (unaligned) staging: the raw 118 / 41 / 4 / 2 wide input features of the three encoders
    if (pro != DOSX_PRO_NONE || epi != DOSX_EPI_BIAS_ACT) {
      dosx_set_error("dosx_gemm: prologue %d / epilogue %d need 4-float aligned operands", pro, epi);
      return -22;
    }
    return L.g.w_layout == 0 ? launch_gemm<NTW, 0, DOSX_PRO_NONE, 0, DOSX_EPI_BIAS_ACT>(L, s)
                             : launch_gemm<NTW, 1, DOSX_PRO_NONE, 0, DOSX_EPI_BIAS_ACT>(L, s);
  }
  if (L.g.w_layout == 0) {
    if (epi == DOSX_EPI_LN && pro == DOSX_PRO_NONE) return launch_gemm<NTW, 0, DOSX_PRO_NONE, 1, DOSX_EPI_LN>(L, s);
    if (epi == DOSX_EPI_BIAS_ACT) {
      switch (pro) {
        case DOSX_PRO_NONE: return launch_gemm<NTW, 0, DOSX_PRO_NONE, 1, DOSX_EPI_BIAS_ACT>(L, s);
        case DOSX_PRO_PRELU: return launch_gemm<NTW, 0, DOSX_PRO_PRELU, 1, DOSX_EPI_BIAS_ACT>(L, s);
        case DOSX_PRO_LN_PRELU: return launch_gemm<NTW, 0, DOSX_PRO_LN_PRELU, 1, DOSX_EPI_BIAS_ACT>(L, s);
        case DOSX_PRO_ROWLN: return launch_gemm<NTW, 0, DOSX_PRO_ROWLN, 1, DOSX_EPI_BIAS_ACT>(L, s);
      }
    }
  } else if (pro == DOSX_PRO_NONE) {
    switch (epi) {
      case DOSX_EPI_BIAS_ACT: return launch_gemm<NTW, 1, DOSX_PRO_NONE, 1, DOSX_EPI_BIAS_ACT>(L, s);
      case DOSX_EPI_PRELU_LN_BWD: return launch_gemm<NTW, 1, DOSX_PRO_NONE, 1, DOSX_EPI_PRELU_LN_BWD>(L, s);
      case DOSX_EPI_RELU_MASK: return launch_gemm<NTW, 1, DOSX_PRO_NONE, 1, DOSX_EPI_RELU_MASK>(L, s);
      case DOSX_EPI_ROWLN_BWD: return launch_gemm<NTW, 1, DOSX_PRO_NONE, 1, DOSX_EPI_ROWLN_BWD>(L, s);
      case DOSX_EPI_PRELU_BWD: return launch_gemm<NTW, 1, DOSX_PRO_NONE, 1, DOSX_EPI_PRELU_BWD>(L, s);
    }
  }
  dosx_set_error("dosx_gemm: unsupported combination w_layout=%d prologue=%d epilogue=%d", L.g.w_layout, pro, epi);
  return -22;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int seg_vec_ok(const DosxSeg* segs, int nseg) {
  for (int i = 0; i < nseg; ++i)
    if ((segs[i].ld & 3) || (segs[i].width & 3) || !aligned16(segs[i].p)) return 0;
  return 1;
}

int gemm_bn(int M, int N, int epi) {
  const bool full_row = (epi == DOSX_EPI_LN || epi == DOSX_EPI_PRELU_LN_BWD || epi == DOSX_EPI_ROWLN_BWD);
  if (full_row) return N <= 128 ? 128 : (N <= 256 ? 256 : 512);
  static int forced = -1;
  if (forced < 0) {
    const char* e = getenv("DOSX_GEMM_BN");
    forced = e ? atoi(e) : 0;
  }
  if (forced == 128 || forced == 256 || forced == 512) return N >= forced ? forced : 128;
  (void)M;
  return 128;
}

}  // namespace

// 32-row blocks per workgroup.  64-row tiles halve the W bytes streamed per flop (the binding
// resource, see gemm_kernel) but also halve the number of workgroups: use them when the grid still
// fills the 256 CUs, or when the 32-row grid would run a nearly empty second round.
inline int gemm_rt(int M, int N, int epi) {
  if (M <= BM) return 1;
  static int forced = -1;
  if (forced < 0) { const char* e = getenv("DOSX_GEMM_RT"); forced = e ? atoi(e) : 0; }
  if (forced == 1 || forced == 2) return forced;
  const int ntiles = ceil_div(N, gemm_bn(M, N, epi));
  const int wg1 = ceil_div(M, BM) * ntiles, wg2 = ceil_div(M, 2 * BM) * ntiles;
  if (wg2 >= 192) return 2;
  if (wg1 > 256 && wg1 <= 400 && wg2 >= 128) return 2;
  return 1;
}

extern "C" int dosx_gemm_partial_rows(int M, int N, int epi) {
  // one partial row per workgroup; row-wise epilogues run as one N tile (N <= 512), the
  // element-wise PRELU_BWD epilogue tiles N by 128.
  const int rows = ceil_div(M, BM * gemm_rt(M, N, epi));
  if (epi == DOSX_EPI_PRELU_BWD) return rows * ceil_div(N, 128);
  return rows;
}

extern "C" int dosx_gemm(const DosxGemm* gp, dosx_stream_t stream) {
  DOSX_CHECK_ARG(gp != nullptr, "dosx_gemm: null descriptor");
  const DosxGemm& g = *gp;
  if (g.M <= 0 || g.N <= 0) return 0;
  DOSX_CHECK_ARG(g.K > 0, "dosx_gemm: K=%d", g.K);
  DOSX_CHECK_ARG(g.nseg >= 1 && g.nseg <= 3, "dosx_gemm: nseg=%d", g.nseg);
  int ksum = 0;
  for (int i = 0; i < g.nseg; ++i) {
    DOSX_CHECK_ARG(g.a[i].p && g.a[i].width > 0 && g.a[i].map.d > 0, "dosx_gemm: bad segment %d", i);
    ksum += g.a[i].width;
  }
  DOSX_CHECK_ARG(ksum == g.K, "dosx_gemm: segment widths sum to %d, K=%d", ksum, g.K);
  DOSX_CHECK_ARG((g.N & 3) == 0 && (g.ldo & 3) == 0 && aligned16(g.out), "dosx_gemm: N/ldo/out must be 4-float aligned");
  DOSX_CHECK_ARG(g.w && g.out, "dosx_gemm: null w/out");
  const bool full_row = (g.epi == DOSX_EPI_LN || g.epi == DOSX_EPI_PRELU_LN_BWD || g.epi == DOSX_EPI_ROWLN_BWD);
  DOSX_CHECK_ARG(!full_row || g.N <= 512, "dosx_gemm: row-wise epilogue needs N <= 512 (hidden <= 256), got %d", g.N);
  DOSX_CHECK_ARG(!g.stats_out || g.N <= 128 * 4, "dosx_gemm: stats_out needs N <= 512");
  DOSX_CHECK_ARG(g.out_map.d > 0, "dosx_gemm: out_map.d must be > 0");
  if (g.res) DOSX_CHECK_ARG(g.res_map.d > 0 && (g.ldr & 3) == 0 && aligned16(g.res), "dosx_gemm: bad residual");
  if (g.pro == DOSX_PRO_LN_PRELU || g.pro == DOSX_PRO_ROWLN)
    DOSX_CHECK_ARG(g.pro_gamma && g.pro_beta, "dosx_gemm: prologue needs gamma/beta");
  if (g.pro == DOSX_PRO_ROWLN) DOSX_CHECK_ARG(g.pro_stats, "dosx_gemm: ROWLN prologue needs stats");
  if (g.pro == DOSX_PRO_PRELU || g.pro == DOSX_PRO_LN_PRELU) DOSX_CHECK_ARG(g.pro_alpha, "dosx_gemm: prologue needs alpha");
  if (g.epi == DOSX_EPI_LN) DOSX_CHECK_ARG(g.aux_out, "dosx_gemm: EPI_LN needs aux_out (rstd)");
  if (g.epi == DOSX_EPI_PRELU_LN_BWD)
    DOSX_CHECK_ARG(g.aux && g.aux_stats && g.epi_gamma && g.epi_beta && g.epi_alpha && (g.ldaux & 3) == 0,
                   "dosx_gemm: PRELU_LN_BWD needs aux/aux_stats/gamma/beta/alpha");
  if (g.epi == DOSX_EPI_ROWLN_BWD)
    DOSX_CHECK_ARG(g.aux && g.aux_stats && g.epi_gamma && (g.ldaux & 3) == 0, "dosx_gemm: ROWLN_BWD needs aux/aux_stats/gamma");
  if (g.epi == DOSX_EPI_RELU_MASK) DOSX_CHECK_ARG(g.aux && (g.ldaux & 3) == 0, "dosx_gemm: RELU_MASK needs aux");
  if (g.epi == DOSX_EPI_PRELU_BWD) DOSX_CHECK_ARG(g.aux && g.epi_alpha && (g.ldaux & 3) == 0, "dosx_gemm: PRELU_BWD needs aux/alpha");

  GemmLaunch L;
  L.g = g;
  L.vecA = seg_vec_ok(g.a, g.nseg) && (g.K & 3) == 0;
  if (g.pro == DOSX_PRO_LN_PRELU || g.pro == DOSX_PRO_ROWLN)
    L.vecA = L.vecA && aligned16(g.pro_gamma) && aligned16(g.pro_beta);
  L.vecW = ((g.ldw & 3) == 0) && aligned16(g.w) && (g.w_layout == 0 ? (g.K & 3) == 0 : (g.N & 3) == 0);
  {
    static int dma_on = -1;
    if (dma_on < 0) { const char* e = getenv("DOSX_GEMM_DMA"); dma_on = (e && e[0] == '1') ? 1 : 0; }
    L.dma = (dma_on && L.vecA && L.vecW && (g.K % BK) == 0) ? 1 : 0;
  }
  L.rt = gemm_rt(g.M, g.N, g.epi);
  int bn = gemm_bn(g.M, g.N, g.epi);
  if (g.stats_out && bn < g.N) bn = g.N <= 256 ? 256 : 512;
  hipStream_t s = to_stream(stream);
  if (bn == 128) return dispatch_gemm<1>(L, s);
  if (bn == 256) return dispatch_gemm<2>(L, s);
  return dispatch_gemm<4>(L, s);
}

// =============================================================================================
// weight gradient: slab[z][n][k] = sum_{m in split z} dY[m][n] * A'[m][k]      (64 x 64 tile / WG)
// =============================================================================================
namespace {

constexpr int WT = 64;        // tile edge (n and k)
constexpr int LDT = WT + 4;   // 68

struct WgradLaunch {
  DosxWgrad g;
  int vecA;
  int vecY;
};

template <int PRO, int VEC>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradLaunch L) {
  const DosxWgrad& g = L.g;
  __shared__ __align__(16) float Ys[BM * LDT];
  __shared__ __align__(16) float Xs[BM * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int k0 = blockIdx.x * WT, n0 = blockIdx.y * WT, z = blockIdx.z;
  const int M = g.M, N = g.N, K = g.K;
  const int chunk = ((M + g.nsplit - 1) / g.nsplit + BM - 1) / BM * BM;
  const int ms = z * chunk, me = min(M, ms + chunk);
  const int wn = wave >> 1, wk = wave & 1;
  const int r = tid >> 3, c4 = (tid & 7) * 4;   // staging: row r, cols c4 and c4+32

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float bsum = 0.f;
  const bool do_bias = (g.slab_bias != nullptr) && (blockIdx.x == 0);

  // staging registers of the NEXT chunk: raw loads are issued before the MFMA block of the current
  // chunk and finished (prologue transform + masks) right before their LDS store
  float4 y0, y1;
  ARaw xr0, xr1;
  AState st;
  bool row_ok = false;
  auto issue = [&](int m) {
    row_ok = (m + r) < me;
    const int gm = min(m + r, me - 1);          // me > ms >= 0 here: always a valid row
    y0 = y1 = f4zero();
    const float* yp = g.dy.p + (size_t)dosx_map_row(g.dy.map, gm) * g.dy.ld;
#pragma unroll
    for (int hseg = 0; hseg < 2; ++hseg) {
      const int n = n0 + c4 + 32 * hseg;
      float4 v = f4zero();
      if (VEC) {
        v = ld4(yp + (n < N ? n : 0));
      } else if (row_ok && n < N) {
        v.x = yp[n];
        if (n + 1 < N) v.y = yp[n + 1];
        if (n + 2 < N) v.z = yp[n + 2];
        if (n + 3 < N) v.w = yp[n + 3];
      }
      if (hseg == 0) y0 = v; else y1 = v;
    }
    a_state_init(st, g.a, g.nseg, PRO, g.pro_stats, g.pro_alpha, gm, row_ok);
    xr0 = a_issue<PRO, VEC>(st, k0 + c4, K, g.pro_gamma, g.pro_beta);
    xr1 = a_issue<PRO, VEC>(st, k0 + c4 + 32, K, g.pro_gamma, g.pro_beta);
  };
  auto store = [&]() {
    float4 a0 = y0, a1 = y1;
    if (VEC) {
      if (!(row_ok && (n0 + c4) < N)) a0 = f4zero();
      if (!(row_ok && (n0 + c4 + 32) < N)) a1 = f4zero();
    }
    st4(&Ys[r * LDT + c4], a0);
    st4(&Ys[r * LDT + c4 + 32], a1);
    st4(&Xs[r * LDT + c4], a_finish<PRO, VEC>(st, xr0, k0 + c4, K));
    st4(&Xs[r * LDT + c4 + 32], a_finish<PRO, VEC>(st, xr1, k0 + c4 + 32, K));
  };

  if (ms < me) issue(ms);
  for (int m = ms; m < me; m += BM) {
    store();
    __syncthreads();
    if (m + BM < me) issue(m + BM);
    if (do_bias && tid < WT) {
#pragma unroll 8
      for (int rr = 0; rr < BM; ++rr) bsum += Ys[rr * LDT + tid];
    }
#pragma unroll
    for (int mm = 0; mm < BM; mm += 2) {
      const float a = Ys[(mm + hh) * LDT + wn * 32 + l31];
      const float b = Xs[(mm + hh) * LDT + wk * 32 + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }

  float* slab = g.slab + (size_t)z * N * K;
  const int kcol = k0 + wk * 32 + l31;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int n = n0 + wn * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
    if (n < N && kcol < K) slab[(size_t)n * K + kcol] = acc[i];
  }
  if (do_bias && tid < WT && n0 + tid < N) g.slab_bias[(size_t)z * N + n0 + tid] = bsum;
}

// Jobs travel as a kernel argument (no device-side table, no host->device copy, graph-capturable).
// Blocks are mapped to (job, 1024-element slice) through the prefix array `first_block`.
constexpr int RP_MAX_JOBS = 64;
constexpr int RP_SLICE = 256;       // elements per block: 64 float4 lanes x 4 waves that split the slabs
struct ReduceLaunch {
  DosxReduceJob job[RP_MAX_JOBS];
  int first_block[RP_MAX_JOBS + 1];
  int n;
};

__global__ __launch_bounds__(256) void reduce_partials_kernel(const ReduceLaunch L) {
  __shared__ float4 red[4][64];
  // binary search: largest j with first_block[j] <= blockIdx.x
  int lo = 0, hi = L.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (L.first_block[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const DosxReduceJob j = L.job[lo];
  const int q = threadIdx.x & 63, sl = threadIdx.x >> 6;      // wave sl sums slabs sl, sl+4, sl+8, ...
  const int i = ((int)blockIdx.x - L.first_block[lo]) * RP_SLICE + q * 4;
  const bool vec = ((j.stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(j.src) & 15) == 0) && ((j.count & 3) == 0);
  float4 s0 = f4zero(), s1 = f4zero();
  if (i < j.count) {
    if (vec) {
      const float* p = j.src + i;
      int k = sl;
      for (; k + 4 < j.nsplit; k += 8) {       // two independent chains keep two loads in flight
        s0 = f4add(s0, ld4(p + (size_t)k * j.stride));
        s1 = f4add(s1, ld4(p + (size_t)(k + 4) * j.stride));
      }
      if (k < j.nsplit) s0 = f4add(s0, ld4(p + (size_t)k * j.stride));
    } else {
      const int e1 = (i + 1 < j.count) ? 1 : 0, e2 = (i + 2 < j.count) ? 2 : 0, e3 = (i + 3 < j.count) ? 3 : 0;
      for (int k = sl; k < j.nsplit; k += 4) {
        const float* p = j.src + (size_t)k * j.stride + i;
        s0 = f4add(s0, make_float4(p[0], p[e1], p[e2], p[e3]));
      }
    }
  }
  red[sl][q] = f4add(s0, s1);
  __syncthreads();
  if (sl == 0 && i < j.count) {
    float4 t = f4add(f4add(red[0][q], red[1][q]), f4add(red[2][q], red[3][q]));
    float* d = j.dst + i;
    const int n = min(4, j.count - i);
    const float tv[4] = {t.x, t.y, t.z, t.w};
    for (int e = 0; e < n; ++e) d[e] = tv[e] + (j.accumulate ? d[e] : 0.f);
  }
}

}  // namespace

extern "C" int dosx_wgrad_splits(int M, int N, int K) {
  if (M <= 0) return 1;
  const int tiles = ceil_div(N, WT) * ceil_div(K, WT);
  int s = ceil_div(512, tiles);
  const int cap = ceil_div(M, 128);
  if (s > cap) s = cap;
  if (s > 16) s = 16;
  if (s < 1) s = 1;
  return s;
}

extern "C" int dosx_wgrad(const DosxWgrad* gp, dosx_stream_t stream) {
  DOSX_CHECK_ARG(gp != nullptr, "dosx_wgrad: null descriptor");
  const DosxWgrad& g = *gp;
  DOSX_CHECK_ARG(g.N > 0 && g.K > 0 && g.nsplit >= 1, "dosx_wgrad: bad dims N=%d K=%d nsplit=%d", g.N, g.K, g.nsplit);
  DOSX_CHECK_ARG(g.nseg >= 1 && g.nseg <= 3 && g.slab && g.dy.p, "dosx_wgrad: bad operands");
  int ksum = 0;
  for (int i = 0; i < g.nseg; ++i) {
    DOSX_CHECK_ARG(g.a[i].p && g.a[i].width > 0 && g.a[i].map.d > 0, "dosx_wgrad: bad segment %d", i);
    ksum += g.a[i].width;
  }
  DOSX_CHECK_ARG(ksum == g.K, "dosx_wgrad: segment widths sum to %d, K=%d", ksum, g.K);
  DOSX_CHECK_ARG(g.dy.map.d > 0, "dosx_wgrad: dy.map.d must be > 0");
  WgradLaunch L;
  L.g = g;
  L.vecA = seg_vec_ok(g.a, g.nseg) && (g.K & 3) == 0;
  if (g.pro == DOSX_PRO_LN_PRELU || g.pro == DOSX_PRO_ROWLN)
    L.vecA = L.vecA && aligned16(g.pro_gamma) && aligned16(g.pro_beta);
  L.vecY = ((g.dy.ld & 3) == 0) && aligned16(g.dy.p) && (g.N & 3) == 0;
  dim3 grid(ceil_div(g.K, WT), ceil_div(g.N, WT), g.nsplit);
  hipStream_t st = to_stream(stream);
  const int vec = L.vecA && L.vecY;
  if (!vec) {
    DOSX_CHECK_ARG(g.pro == DOSX_PRO_NONE, "dosx_wgrad: prologue %d needs 4-float aligned operands", g.pro);
    hipLaunchKernelGGL((wgrad_kernel<DOSX_PRO_NONE, 0>), grid, dim3(256), 0, st, L);
  } else {
    switch (g.pro) {
      case DOSX_PRO_NONE: hipLaunchKernelGGL((wgrad_kernel<DOSX_PRO_NONE, 1>), grid, dim3(256), 0, st, L); break;
      case DOSX_PRO_PRELU: hipLaunchKernelGGL((wgrad_kernel<DOSX_PRO_PRELU, 1>), grid, dim3(256), 0, st, L); break;
      case DOSX_PRO_LN_PRELU: hipLaunchKernelGGL((wgrad_kernel<DOSX_PRO_LN_PRELU, 1>), grid, dim3(256), 0, st, L); break;
      case DOSX_PRO_ROWLN: hipLaunchKernelGGL((wgrad_kernel<DOSX_PRO_ROWLN, 1>), grid, dim3(256), 0, st, L); break;
      default: DOSX_CHECK_ARG(false, "dosx_wgrad: bad prologue %d", g.pro);
    }
  }
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_reduce_partials(const DosxReduceJob* jobs_host, int n_jobs, dosx_stream_t stream) {
  if (n_jobs <= 0) return 0;
  DOSX_CHECK_ARG(jobs_host != nullptr, "dosx_reduce_partials: null job table");
  for (int done = 0; done < n_jobs; done += RP_MAX_JOBS) {
    ReduceLaunch L;
    L.n = n_jobs - done < RP_MAX_JOBS ? n_jobs - done : RP_MAX_JOBS;
    int blocks = 0;
    for (int i = 0; i < L.n; ++i) {
      const DosxReduceJob& j = jobs_host[done + i];
      DOSX_CHECK_ARG(j.src && j.dst && j.count > 0 && j.nsplit > 0, "dosx_reduce_partials: bad job %d", done + i);
      L.job[i] = j;
      L.first_block[i] = blocks;
      blocks += ceil_div(j.count, RP_SLICE);
    }
    L.first_block[L.n] = blocks;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(blocks), dim3(256), 0, to_stream(stream), L);
    DOSX_LAUNCH_CHECK();
  }
  return 0;
}

#ifdef DOSX_STAMPS
extern "C" int dosx_debug_read_stamps(unsigned long long* host64x64) {
  return (int)hipMemcpyFromSymbol(host64x64, HIP_SYMBOL(dosx_stamp_buf), sizeof(unsigned long long) * 64 * 64);
}
#endif
