// fp32 MFMA GEMM family for the DOSTransformer hot path (gfx950).
//
//   dosx_gemm   : C[M,N] = epilogue( prologue(A)[M,K] . B )       (nn.Linear fwd / dgrad)
//   dosx_wgrad  : slab[s][N,K] = dY[ms:me]^T . prologue(A)[ms:me]  (nn.Linear wgrad, split over M)
//   dosx_reduce_partials : deterministic sum of the split slabs
//
// One workgroup = 8 waves: 4 matrix waves (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain, so results
// track an fp32 torch reference to rounding) + 4 staging waves (global -> registers -> LDS).  A is gathered /
// concatenated / normalised on the fly while it is staged, so torch.cat([x[row], x[col], e]) and the
// LayerNorm/PReLU between the two Linear layers of every MLP never exist in HBM.
#include <stdlib.h>

#include "common.h"

typedef int v4i32 __attribute__((ext_vector_type(4)));

// Diagnostic build only (-DDOSX_STAMPS, never shipped): wave 0 of a few workgroups records
// s_memtime at phase boundaries (see tools/stamp_gemm.py).
#ifdef DOSX_STAMPS
extern "C" { __device__ unsigned long long dosx_stamp_buf[64 * 64]; }
#define STAMP(slot)                                                                              \
  do {                                                                                           \
    if ((threadIdx.x == 0) && (blockIdx.x % 37 == 0) && (blockIdx.x / 37) < 64 && (slot) < 64)   \
      dosx_stamp_buf[(blockIdx.x / 37) * 64 + (slot)] = __builtin_amdgcn_s_memtime();            \
  } while (0)
// staging wave 0 (thread 256) of workgroup 0 -> row 32 of the stamp buffer
#define STAMP_S(slot)                                                                            \
  do {                                                                                           \
    if ((threadIdx.x == 256) && (blockIdx.x == 0) && (slot) < 64)                                \
      dosx_stamp_buf[32 * 64 + (slot)] = __builtin_amdgcn_s_memtime();                           \
  } while (0)
// wgrad_kernel: workgroup (0,0,0), matrix wave 0 -> row 0, staging wave 0 -> row 32
#define WSTAMP(slot)                                                                             \
  do {                                                                                           \
    if (threadIdx.x == 0 && blockIdx.x == 0 && (slot) < 64)                                      \
      dosx_stamp_buf[(slot)] = __builtin_amdgcn_s_memtime();                                     \
  } while (0)
#define WSTAMP_S(slot)                                                                           \
  do {                                                                                           \
    if (threadIdx.x == 256 && blockIdx.x == 0 && (slot) < 64)                                    \
      dosx_stamp_buf[32 * 64 + (slot)] = __builtin_amdgcn_s_memtime();                           \
  } while (0)
#else
#define STAMP(slot) do { } while (0)
#define STAMP_S(slot) do { } while (0)
#define WSTAMP(slot) do { } while (0)
#define WSTAMP_S(slot) do { } while (0)
#endif

namespace {

constexpr int BM = 32;
constexpr int BK = 32;
constexpr int LDA = BK + 4;  // 36 floats: 16-B aligned rows, conflict-free ds_read_b128
constexpr int GEMM_KMAX = 1024;   // largest K of a GEMM with a LayerNorm prologue (its gamma/beta live in LDS)

struct GemmLaunch {
  DosxGemm g;
  int vecA;
  int vecW;
  int rt;    // 32-row blocks per workgroup (1 or 2)
  int m_base = 0;     // first row of this launch (tail launch of a split call, see gemm_tail_split)
  int part_base = 0;  // first partial-row block of this launch
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
  float4 v;
};

template <int PRO, int VEC>
__device__ __forceinline__ ARaw a_issue(const AState& st, int k, int K, const float* gamma, const float* beta) {
  ARaw r;
  r.v = f4zero();
  if (VEC) {
    const int kc = (k < K) ? k : 0;
    const float* p0 = st.rp[0] + kc;
    const float* p1 = st.rp[1] + (kc - st.w0);
    const float* p2 = st.rp[2] + (kc - st.w01);
    const float* p = (kc < st.w0) ? p0 : ((kc < st.w01) ? p1 : p2);
    r.v = ld4(p);
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

// g / b: the 4 gamma / beta values of the prologue LayerNorm at columns k..k+3 (ignored otherwise).
// MASK = 0: no zero-fill (the caller knows k < K; rows beyond M are clamped duplicates whose results are
// never stored).  Every VALU instruction of a staging wave has to squeeze in between the matrix wave's
// back-to-back MFMAs on the same SIMD (measured: ~30 clk per VALU op), so the interior path has none.
template <int PRO, int VEC, int MASK = 1>
__device__ __forceinline__ float4 a_finish(const AState& st, const ARaw& r, int k, int K, float4 g, float4 b) {
  float4 v = r.v;
  if (!VEC) return v;
  if (PRO == DOSX_PRO_PRELU) {
    v.x = prelu_f(v.x, st.alpha); v.y = prelu_f(v.y, st.alpha);
    v.z = prelu_f(v.z, st.alpha); v.w = prelu_f(v.w, st.alpha);
  } else if (PRO == DOSX_PRO_LN_PRELU) {
    v.x = prelu_f(v.x * g.x + b.x, st.alpha); v.y = prelu_f(v.y * g.y + b.y, st.alpha);
    v.z = prelu_f(v.z * g.z + b.z, st.alpha); v.w = prelu_f(v.w * g.w + b.w, st.alpha);
  } else if (PRO == DOSX_PRO_ROWLN) {
    v.x = (v.x - st.mean) * st.rstd * g.x + b.x; v.y = (v.y - st.mean) * st.rstd * g.y + b.y;
    v.z = (v.z - st.mean) * st.rstd * g.z + b.z; v.w = (v.w - st.mean) * st.rstd * g.w + b.w;
  }
  if (MASK && !(st.ok && k < K)) v = f4zero();
  return v;
}

// ---------------------------------------------------------------------------------------------
// C = prologue(A) . B  with fused row-wise epilogues.   NTW = 32-wide MFMA tiles per wave,
// BN = 128*NTW columns per workgroup.  WL: 0 = W[N,K] (k-contiguous), 1 = W[K,N] (n-contiguous).
// RT = 32-row blocks per workgroup (BM = 32*RT): RT = 2 halves the W bytes streamed per flop.
//
// WAVE SPECIALISATION.  A workgroup is 8 waves = 2 per SIMD: waves 0-3 ("matrix waves") only read MFMA
// fragments from LDS and issue v_mfma; waves 4-7 ("staging waves") only move the next k-chunk
// global -> registers -> (prologue transform) -> LDS.  Two LDS stage buffers, ONE barrier per chunk.
// Measured with s_memtime stamps on the previous 4-wave kernel (every wave staged, then multiplied):
// at this workload's sizes (<= 1-2 workgroups per CU) 30-60 % of every k-chunk was the exposed
// global-load wait + ds_write phase, because one wave per SIMD cannot overlap its own staging with its
// own MFMAs.  With a staging wave next to each matrix wave the matrix pipe only stalls on the barrier.
// The epilogue uses all 8 waves (4 rows each per 32-row block).
// ---------------------------------------------------------------------------------------------
// (128-column tiles with a light epilogue are held to 128 VGPRs = 4 waves per SIMD: two workgroups per CU)
// Tiles of NTW >= DOSX_HOIST_MAX_NTW x 128 columns fetch the row operands of their epilogue AFTER the k-loop: with 512-column
// tiles (eDOS, 2H = 512) the hoisted rows pushed the staging waves past 256 VGPRs - 13-76 spilled registers, reloaded with
// scratch loads INSIDE the staging loop, where they queue behind the chunk loads (vmcnt is in-order).  Round 3: the edge
// dgrad GEMM with the PReLU/LayerNorm-backward epilogue 196 -> 184 us in the eDOS step, no spills left in the backward tiles.
#ifndef DOSX_HOIST_MAX_NTW
#define DOSX_HOIST_MAX_NTW 4
#endif
template <int RTP, int NTW, int WL, int PRO, int VEC, int EPI>
__device__ __forceinline__ void gemm_body(const GemmLaunch& L, const int bid) {     // bid: this workgroup's index in L's grid
  DOSX_SET_MAIN_PRIO();
  const DosxGemm& g = L.g;
  // RTP = 0: HALF tile.  The workgroup owns 16 rows and multiplies with the 16x16x4 MFMA (same flop rate, half the
  // rows per instruction): kernels with M of a few hundred rows get twice the workgroups, each with half the MFMA
  // time per k-chunk.  Staging, LDS layout and register sets are those of RT = 1 (the staging waves still fill a
  // 32-row A tile; its upper 16 rows are the next workgroup's rows, loaded from valid addresses and never read).
  constexpr bool HALF = (RTP == 0 || RTP == 3);
  constexpr int HT = RTP == 3 ? 3 : 1;      // 16-row sub-tiles of a HALF-family workgroup (RTP = 3: 48 rows, for row
                                            // counts where 32 rows per CU are too few and 64 leave CUs idle: M = 9000)
  constexpr int RT = HALF ? (HT + 1) / 2 : RTP;   // 32-row blocks of the staged A tile / register sets of the staging waves
  constexpr int RTE = HALF ? 1 : RT;        // row blocks of the epilogue (the HALF family runs it as one block)
  constexpr int BMR = HALF ? 16 * HT : BM * RT;  // rows of C this workgroup owns
  constexpr int BMS = BM * RT;              // rows of the staged A tile
  constexpr int BN = 128 * NTW;
  constexpr int LDWT = (WL == 0) ? (BK + 4) : (BN + 4);
  constexpr int WROWS = (WL == 0) ? BN : BK;
  constexpr int LDC = BN + 4;
  constexpr int STAGE = BMS * LDA + WROWS * LDWT;     // floats of one staging buffer (A tile + W tile)
  // TB: three stage buffers instead of two (256-column HALF-family tiles: they run one workgroup per CU anyway).  The
  // staging waves then run TWO chunks ahead, so at the barrier that ends chunk c the data of chunk c+1 is already
  // visible and the matrix waves fetch its first fragments underneath the last MFMAs of chunk c - without that, every
  // chunk starts with ~1000 clk of exposed ds_read latency behind the barrier (stamps: 4100 clk per chunk for 3072 clk
  // of MFMAs, also with the global loads removed).
  constexpr bool TB = HALF && NTW == 2;
  constexpr int NB = TB ? 3 : 2;
  constexpr int CTILE = BM * LDC;
  constexpr int CG = (BN + 255) / 256;   // float4 column groups per lane in the row-wise epilogue
  constexpr int NW4 = BN / 32;           // float4 W loads per staging thread per k-chunk
  constexpr int ER = HALF ? 2 * HT : 4;  // epilogue rows per wave per row block (8 waves)
  constexpr bool PROLN = VEC && (PRO == DOSX_PRO_LN_PRELU || PRO == DOSX_PRO_ROWLN);

  extern __shared__ __align__(16) float smem[];
  float* Cs = smem;
  float* Ps = smem + RT * CTILE;         // [8][2][BN] + 8 (behind the C tiles; staging memory is dead by then)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // 1-D grid.  Workgroups go to the 8 XCDs round-robin by linear id: within a group of 8 consecutive row
  // blocks the column blocks are enumerated next, so all column blocks of one row block run on the same XCD
  // (id % 8 == row block % 8) and its A rows cross the fabric once, not once per column block.
  const int gx = (g.M - L.m_base + BMR - 1) / BMR, gy = (g.N + BN - 1) / BN;
  int bx = bid, by = 0;
  if (gy > 1) {                                            // (one column tile: the mapping is the identity - skip its four run-time
    const int lin = bid, grp = lin / (8 * gy), rem = lin % (8 * gy);     //  integer divisions, ~40 instructions each)
    const int rows_in_grp = min(8, gx - grp * 8);          // last group may be short
    bx = grp * 8 + rem % rows_in_grp;
    by = rem / rows_in_grp;
  }
  // EPI_SEGSUM: the row tiles are NODE-ALIGNED and of variable height (<= BMR): workgroup t owns the rows
  // [seg_tile[t], seg_tile[t+1]) = the whole destination segments of the nodes [seg_tile[T+1+t], seg_tile[T+2+t]).
  // Every bound below that says M means "end of this workgroup's rows".
  constexpr bool SEGT = (EPI == DOSX_EPI_SEGSUM || EPI == DOSX_EPI_PRELU_LN_BWD_SEG);   // node-aligned row tiles + segment sums
  int m0_ = L.m_base + bx * BMR, mend_ = g.M, nlo_ = 0, nhi_ = 0;
  if constexpr (SEGT) {
    bx = bid;
    by = 0;
    m0_ = g.seg_tile[bx];
    mend_ = g.seg_tile[bx + 1];
    nlo_ = g.seg_tile[g.seg_ntiles + 1 + bx];
    nhi_ = g.seg_tile[g.seg_ntiles + 2 + bx];
    if (m0_ >= mend_) {                                   // no rows: at most nodes without a row of their own (ghost /
      const int w4 = g.N >> 2;                            // padding nodes behind the last real tile): aggregate = 0
      for (int i = (int)threadIdx.x; i < (nhi_ - nlo_) * w4; i += 512)
        st4(g.seg_agg + (size_t)(nlo_ + i / w4) * g.N + (i % w4) * 4, f4zero());
      if (EPI == DOSX_EPI_PRELU_LN_BWD_SEG && g.partials != nullptr)      // one partial row per tile slot: an empty one adds zeros
        for (int i = (int)threadIdx.x; i < g.partial_ld; i += 512) g.partials[(size_t)bx * g.partial_ld + i] = 0.f;
      return;
    }
  }
  const int m0 = m0_, n0 = by * BN;
  const int M = mend_, N = g.N, K = g.K;
  const int nk = (K + BK - 1) / BK;

  f32x16 acc[RT][NTW];
  f32x4 acch[HT][2 * NTW];  // HALF: 16x16 accumulators of this wave's HT x 2*NTW tiles

  STAMP(0);
  // ---- epilogue operand prefetch (all 8 waves; wave w owns rows 4w..4w+3 of each 32-row block) -----
  // Every global operand of the row-wise epilogue (bias / gamma / beta vectors, the rows of `aux` and
  // `res`, the per-row statistics) is loaded HERE, at kernel start, from always-valid (clamped)
  // addresses: the loads fly under the whole k-loop and the row loop at the end touches no global
  // memory except its stores (a load inside that loop costs one exposed L2/HBM round trip per row;
  // issued after the k-loop they still cost one round trip per kernel: ~1.2k clk of a 30k-clk kernel).
  const int ncols = min(BN, N - n0);
  const float invN = 1.f / (float)N;
  constexpr int epi = EPI;     // compile-time: only this epilogue's code exists in the kernel
  constexpr bool is_prelu_ln = (epi == DOSX_EPI_PRELU_LN_BWD || epi == DOSX_EPI_PRELU_LN_BWD_SEG), is_rowln = (epi == DOSX_EPI_ROWLN_BWD);
  constexpr bool aux_first = (epi != DOSX_EPI_BIAS_ACT && epi != DOSX_EPI_LN && epi != DOSX_EPI_SEGSUM);   // operand 1 is `aux`
  constexpr bool use_stats = is_prelu_ln || is_rowln;
  const bool has1 = aux_first ? (g.aux != nullptr)
                              : (g.res != nullptr && (epi == DOSX_EPI_BIAS_ACT || (epi == DOSX_EPI_SEGSUM && g.out != nullptr)));
  const bool has2 = is_rowln && g.res != nullptr;
  const float* p1 = aux_first ? g.aux : g.res;
  const int ld1 = aux_first ? g.ldaux : g.ldr;
  float4 pg[CG], pb[CG];   // column partial sums (dgamma, dbeta) over all row blocks of this workgroup
  float pal = 0.f;         // dalpha partial
#pragma unroll
  for (int j = 0; j < CG; ++j) { pg[j] = f4zero(); pb[j] = f4zero(); }
  float e_alpha = 0.f;
  if (is_prelu_ln || epi == DOSX_EPI_PRELU_BWD) e_alpha = *g.epi_alpha;
  bool on[CG];
  int gcol[CG];
  float4 biasv[CG], gamv[CG], betv[CG];
  // LN epilogue: quarter-wave layout (16 lanes per row, KQ float4 per lane at columns 4*q16 + 64*k)
  constexpr int KQ = BN / 64;
  float4 biasq[KQ];
  if constexpr (EPI == DOSX_EPI_LN && NTW <= 2) {
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
      const int c = (lane & 15) * 4 + 64 * k;
      biasq[k] = (g.bias && c < ncols) ? ld4(g.bias + n0 + c) : f4zero();
    }
  }
#pragma unroll
  for (int j = 0; j < CG; ++j) {
    const int c = lane * 4 + 256 * j;
    on[j] = c < ncols;
    gcol[j] = n0 + (on[j] ? c : 0);
    biasv[j] = f4zero(); gamv[j] = f4zero(); betv[j] = f4zero();
    if (g.bias) biasv[j] = ld4(g.bias + gcol[j]);
    if (use_stats) {
      gamv[j] = ld4(g.epi_gamma + gcol[j]);
      if (is_prelu_ln) betv[j] = ld4(g.epi_beta + gcol[j]);
    }
  }
  float4 pv1[RTE][ER][CG], pv2[RTE][ER][CG];
  float st0[RTE][ER], st1[RTE][ER];
  size_t orow_[RTE][ER];
  // The per-row operands are hoisted above the k-loop only for the backward epilogues (they always have
  // them).  The plain bias/activation epilogue keeps its registers for occupancy instead: at <= 128
  // VGPRs two 8-wave workgroups share a CU, which matters more for its (larger) grids; its optional
  // residual rows are fetched after the k-loop; so do the 512-column tiles (DOSX_HOIST_MAX_NTW).
  constexpr bool HOIST = (epi != DOSX_EPI_BIAS_ACT) && (NTW < DOSX_HOIST_MAX_NTW);
  auto prefetch_rows = [&]() {
#pragma unroll
  for (int rt = 0; rt < RTE; ++rt) {
    const int mb = m0 + 32 * rt;
#pragma unroll
    for (int i = 0; i < ER; ++i) {
      const int rc = min(mb + wave * ER + i, M - 1);
      orow_[rt][i] = (size_t)(epi == DOSX_EPI_BIAS_ACT ? dosx_map_row(g.out_map, rc) : rc) * g.ldo;
      st0[rt][i] = 0.f; st1[rt][i] = 0.f;
#pragma unroll
      for (int j = 0; j < CG; ++j) { pv1[rt][i][j] = f4zero(); pv2[rt][i][j] = f4zero(); }
    }
    if (has1) {          // wave-uniform, outside every loop: the loads of all rows are issued back to back
      // res_col0 (plain epilogue only): the residual covers the columns [res_col0, N) and its own column 0 is the
      // output's column res_col0 - the edge-state gradient rides on the last H columns of the [E,3H] concat gradient
      const int rc0 = aux_first ? 0 : g.res_col0;
#pragma unroll
      for (int i = 0; i < ER; ++i) {
        const int rc = min(mb + wave * ER + i, M - 1);
        const size_t r1 = (size_t)(aux_first ? rc : dosx_map_row(g.res_map, rc)) * ld1;
#pragma unroll
        for (int j = 0; j < CG; ++j) {
          const bool in = gcol[j] >= rc0;                       // (load unconditionally from a valid address, then select)
          const float4 t = ld4(p1 + r1 + (in ? gcol[j] - rc0 : 0));
          pv1[rt][i][j] = in ? t : f4zero();
        }
      }
    }
    if (has2) {
#pragma unroll
      for (int i = 0; i < ER; ++i) {
        const int rc = min(mb + wave * ER + i, M - 1);
        const size_t r2 = (size_t)dosx_map_row(g.res_map, rc) * g.ldr;
#pragma unroll
        for (int j = 0; j < CG; ++j) pv2[rt][i][j] = ld4(g.res + r2 + gcol[j]);
      }
    }
    if (use_stats) {
#pragma unroll
      for (int i = 0; i < ER; ++i) {
        const int rc = min(mb + wave * ER + i, M - 1);
        st0[rt][i] = g.aux_stats[is_rowln ? 2 * (size_t)rc : (size_t)rc];
        st1[rt][i] = g.aux_stats[is_rowln ? 2 * (size_t)rc + 1 : (size_t)rc];
      }
    }
  }
  };
  if constexpr (HOIST) prefetch_rows();
  // EPI_LN with gathered row addends (DosxGemm.add_p / add_q: the node products of the factored EdgeModel Linear): the two rows
  // of every output row this lane normalises are requested here, above the k-loop (L2-resident [nodes, N] tensors; K is H, so
  // the loop is short - behind it the gathers would be an exposed index -> row round-trip pair), in the layout of the epilogue
  // that consumes them: quarter wave per row (NTW <= 2), or whole wave per row fetched after the loop (512-column tiles).
  constexpr int LNP = (ER + 3) / 4;                 // quarter-wave passes over a wave's ER rows
  float4 adq[(EPI == DOSX_EPI_LN && NTW <= 2) ? RTE : 1][(EPI == DOSX_EPI_LN && NTW <= 2) ? LNP : 1][2][KQ];
  const bool has_add = (EPI == DOSX_EPI_LN) && g.add_p != nullptr;
  if constexpr (EPI == DOSX_EPI_LN && NTW <= 2) {
    if (has_add) {
      int ip[RTE][LNP], iq[RTE][LNP];
#pragma unroll
      for (int rt = 0; rt < RTE; ++rt)
#pragma unroll
        for (int p = 0; p < LNP; ++p) {
          const int li = 4 * p + (lane >> 4);
          const int rc = min(m0 + 32 * rt + wave * ER + (li < ER ? li : 0), M - 1);
          ip[rt][p] = g.add_ip[rc];
          iq[rt][p] = g.add_iq[rc];
        }
#pragma unroll
      for (int rt = 0; rt < RTE; ++rt)
#pragma unroll
        for (int p = 0; p < LNP; ++p)
#pragma unroll
          for (int k = 0; k < KQ; ++k) {
            const int c = (lane & 15) * 4 + 64 * k;
            const int cc = n0 + (c < ncols ? c : 0);
            adq[rt][p][0][k] = ld4(g.add_p + (size_t)ip[rt][p] * g.ld_add + cc);
            adq[rt][p][1][k] = ld4(g.add_q + (size_t)iq[rt][p] * g.ld_add + cc);
          }
    }
  }

  if (wave_u >= 4) {
    // =============================== staging waves ===============================================
    const int st = tid - 256;                      // 0..255
    const int arow = st >> 3, akq = (st & 7) * 4;
    AState ast[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
      a_state_init(ast[r], g.a, g.nseg, PRO, g.pro_stats, g.pro_alpha, min(m0 + arow + 32 * r, M - 1),
                   (m0 + arow + 32 * r) < M);
    auto issueA = [&](ARaw(&ar)[RT], int k0) {
#pragma unroll
      for (int r = 0; r < RT; ++r) ar[r] = a_issue<PRO, VEC>(ast[r], k0 + akq, K, g.pro_gamma, g.pro_beta);
    };
    const float* Gs = smem + NB * STAGE;            // [KMAX] gamma, [KMAX] beta of the prologue LayerNorm
    const float* Bs = Gs + GEMM_KMAX;
    auto storeA = [&](float* Ad, const ARaw(&ar)[RT], int k0) {
      const int k = k0 + akq;
      float4 gv = f4zero(), bv = f4zero();
      if constexpr (PROLN) {                         // (K % 4 == 0 here: k < K means k+3 < K)
        gv = ld4(Gs + (k < K ? k : 0));
        bv = ld4(Bs + (k < K ? k : 0));
      }
#pragma unroll
      for (int r = 0; r < RT; ++r)
        st4(&Ad[(arow + 32 * r) * LDA + akq], a_finish<PRO, VEC, 1>(ast[r], ar[r], k, K, gv, bv));
    };
    auto loadW = [&](int k0, float4(&wr)[NW4]) {
#pragma unroll
      for (int i = 0; i < NW4; ++i) {
        float4 v = f4zero();
        if (WL == 0) {
          const int n = n0 + (st >> 3) + 32 * i, k = k0 + (st & 7) * 4;
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
          const int lin = st + 256 * i;
          const int r = lin / (BN / 4), c4 = (lin % (BN / 4)) * 4;
          const int k = k0 + r, n = n0 + c4;
          int kr = min(k, K - 1), wofs = 0;          // DosxGemm.w_seg_off: every K-segment multiplies its own block of W
          if (g.w_seg_off != 0) {
            const int w0_ = g.a[0].width, w01_ = g.nseg > 2 ? w0_ + g.a[1].width : 0x7fffffff;
            const int sg = kr < w0_ ? 0 : (kr < w01_ ? 1 : 2);
            kr -= sg == 0 ? 0 : (sg == 1 ? w0_ : w01_);
            wofs = sg * g.w_seg_off;
          }
          if (VEC) {
            v = ld4(g.w + (size_t)kr * g.ldw + wofs + (n < N ? n : 0));
          } else if (k < K && n < N) {
            const float* p = g.w + (size_t)kr * g.ldw + wofs + n;
            v.x = p[0];
            if (n + 1 < N) v.y = p[1];
            if (n + 2 < N) v.z = p[2];
            if (n + 3 < N) v.w = p[3];
          }
        }
        wr[i] = v;
      }
    };
    auto storeW = [&](float* Wd, int k0, const float4(&wr)[NW4]) {
#pragma unroll
      for (int i = 0; i < NW4; ++i) {
        float4 v = wr[i];
        if (WL == 0) {
          if (VEC && !((n0 + (st >> 3) + 32 * i) < N && (k0 + (st & 7) * 4) < K)) v = f4zero();
          st4(&Wd[((st >> 3) + 32 * i) * LDWT + (st & 7) * 4], v);
        } else {
          const int lin = st + 256 * i;
          if (VEC && !((k0 + lin / (BN / 4)) < K && (n0 + (lin % (BN / 4)) * 4) < N)) v = f4zero();
          st4(&Wd[(lin / (BN / 4)) * LDWT + (lin % (BN / 4)) * 4], v);
        }
      }
    };

    // Register pipeline, two chunks deep: chunk c travels in register set c & 1 and is stored to LDS
    // buffer c & 1 one iteration before the matrix waves read it, so every global load has two chunk
    // periods to land.  `issue(set_a, set_w, k0)` starts the loads of a chunk, `store(buf, set_a,
    // set_w, k0)` finishes it (prologue transform, LDS store).
    ARaw ar0[RT], ar1[RT];
    float4 wr0[NW4], wr1[NW4];
    auto pipeline = [&](auto&& issue, auto&& store) {
      STAMP_S(0);
      issue(ar0, wr0, 0);
      if (nk > 1) issue(ar1, wr1, BK);
      if constexpr (PROLN) {
        float* Gw = smem + NB * STAGE;
        for (int k = st * 4; k < K; k += 1024) {
          st4(Gw + k, ld4(g.pro_gamma + k));
          st4(Gw + GEMM_KMAX + k, ld4(g.pro_beta + k));
        }
        __syncthreads();                                 // gamma / beta visible to every staging wave
      }
      store(smem, ar0, wr0, 0);
      if (nk > 2) issue(ar0, wr0, 2 * BK);
      if constexpr (TB) {
        // chunk c travels in register set c & 1 and is stored to buffer c % 3 TWO iterations before it is read
        if (nk > 1) {
          store(smem + STAGE, ar1, wr1, BK);
          if (nk > 3) issue(ar1, wr1, 3 * BK);
        }
        __syncthreads();                                 // chunks 0 and 1 are visible
        int b2 = 2;                                      // buffer of chunk kt + 2
        for (int kt = 0; kt < nk; kt += 2) {
          if (kt + 2 < nk) {                             // buffer (kt+2) % 3 was last read in iteration kt - 1
            store(smem + b2 * STAGE, ar0, wr0, (kt + 2) * BK);
            if (kt + 4 < nk) issue(ar0, wr0, (kt + 4) * BK);
          }
          b2 = b2 == 2 ? 0 : b2 + 1;
          __syncthreads();
          if (kt + 1 >= nk) break;
          if (kt + 3 < nk) {
            store(smem + b2 * STAGE, ar1, wr1, (kt + 3) * BK);
            if (kt + 5 < nk) issue(ar1, wr1, (kt + 5) * BK);
          }
          b2 = b2 == 2 ? 0 : b2 + 1;
          __syncthreads();
        }
      } else {
      STAMP_S(1);
      __syncthreads();                                   // chunk 0 is visible
      for (int kt = 0; kt < nk; kt += 2) {
        STAMP_S(2 + 3 * kt);
        if (kt + 1 < nk) {                               // chunk kt+1: set 1 -> buffer 1 (last read a barrier ago)
          store(smem + STAGE, ar1, wr1, (kt + 1) * BK);
          STAMP_S(3 + 3 * kt);
          if (kt + 3 < nk) issue(ar1, wr1, (kt + 3) * BK);
        }
        STAMP_S(4 + 3 * kt);
        __syncthreads();
        if (kt + 1 >= nk) break;
        STAMP_S(5 + 3 * kt);
        if (kt + 2 < nk) {                               // chunk kt+2: set 0 -> buffer 0
          store(smem, ar0, wr0, (kt + 2) * BK);
          STAMP_S(6 + 3 * kt);
          if (kt + 4 < nk) issue(ar0, wr0, (kt + 4) * BK);
        }
        STAMP_S(7 + 3 * kt);
        __syncthreads();
      }
      }
    };

    // ---- interior fast path: buffer addressing, no per-chunk vector ALU ---------------------------
    // Every VALU instruction of a staging wave has to squeeze in between the back-to-back MFMAs of the
    // matrix wave on the same SIMD (measured ~30 clk per VALU op, stamps in DESIGN.md).  So the per-lane
    // byte offsets of every load slot are computed ONCE; a chunk advances a scalar offset only
    // (buffer_load_dwordx4 v, voffset, rsrc, soffset) and its stores are plain ds_write_b128.
    // Needs: aligned operands, K % 32 == 0, segment widths % 32 == 0 (a chunk never straddles two
    // segments), lane offsets < 2^31 (checked per wave; a wave that fails uses the pointer path, both
    // produce the same LDS image).
    bool fast = VEC && (K % BK) == 0;
    uint32_t voffA[RT][3], voffW[NW4];
    if (VEC) {
      const int w0 = g.a[0].width, w1 = g.nseg > 1 ? g.a[1].width : 0;
      if (g.nseg > 1 && ((w0 | w1) & (BK - 1))) fast = false;
      bool fits = true;
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        const int gm = min(m0 + arow + 32 * r, M - 1);
#pragma unroll
        for (int sgi = 0; sgi < 3; ++sgi) {
          voffA[r][sgi] = 0;
          if (sgi < g.nseg) {
            const size_t off = ((size_t)dosx_map_row(g.a[sgi].map, gm) * (size_t)g.a[sgi].ld + akq) * 4;
            fits = fits && off < 0x7fffffffu;
            voffA[r][sgi] = (uint32_t)off;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NW4; ++i) {
        size_t off;
        if (WL == 0) {
          off = ((size_t)min(n0 + (st >> 3) + 32 * i, N - 1) * g.ldw + (st & 7) * 4) * 4;
        } else {
          const int lin = st + 256 * i, c4 = (lin % (BN / 4)) * 4;
          off = ((size_t)(lin / (BN / 4)) * g.ldw + ((n0 + c4) < N ? (n0 + c4) : 0)) * 4;
        }
        fits = fits && off < 0x7fffffffu;
        voffW[i] = (uint32_t)off;
      }
      if (WL == 1 && (size_t)K * g.ldw * 4 >= 0x7fffffffu) fits = false;
      fast = fast && __all(fits);
    }
    if (fast) {
      const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)g.w, 0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rA0 = __builtin_amdgcn_make_buffer_rsrc((void*)g.a[0].p, 0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rA1 =
          __builtin_amdgcn_make_buffer_rsrc((void*)g.a[g.nseg > 1 ? 1 : 0].p, 0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rA2 =
          __builtin_amdgcn_make_buffer_rsrc((void*)g.a[g.nseg > 2 ? 2 : 0].p, 0, 0x7fffffff, 0x00020000);
      const int e0 = g.nseg > 1 ? g.a[0].width : 0x7fffffff;                       // end of segment 0
      const int e1 = g.nseg > 2 ? e0 + g.a[1].width : 0x7fffffff;                  // end of segment 1
      auto issue = [&](ARaw(&ar)[RT], float4(&wr)[NW4], int k0) {
        const int k0u = __builtin_amdgcn_readfirstlane(k0);
        const int sgi = k0u < e0 ? 0 : (k0u < e1 ? 1 : 2);                          // wave-uniform
        const int soffA = __builtin_amdgcn_readfirstlane((k0u - (sgi == 0 ? 0 : (sgi == 1 ? e0 : e1))) * 4);
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          v4i32 v;
          if (sgi == 0) v = __builtin_amdgcn_raw_buffer_load_b128(rA0, voffA[r][0], soffA, 0);
          else if (sgi == 1) v = __builtin_amdgcn_raw_buffer_load_b128(rA1, voffA[r][1], soffA, 0);
          else v = __builtin_amdgcn_raw_buffer_load_b128(rA2, voffA[r][2], soffA, 0);
          ar[r].v = __builtin_bit_cast(float4, v);
        }
        int kw = k0u, wso = 0;
        if (WL == 1 && g.w_seg_off != 0) {           // every K-segment multiplies its own block of W (rows restart at 0)
          kw = k0u - (sgi == 0 ? 0 : (sgi == 1 ? e0 : e1));
          wso = sgi * g.w_seg_off;
        }
        const int soffW = __builtin_amdgcn_readfirstlane((WL == 0) ? k0u * 4 : (kw * g.ldw + wso) * 4);   // (keeps it in an SGPR: no waterfall loop)
#pragma unroll
        for (int i = 0; i < NW4; ++i)
          wr[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rW, voffW[i], soffW, 0));
      };
      auto store = [&](float* buf, const ARaw(&ar)[RT], const float4(&wr)[NW4], int k0) {
        const int k = k0 + akq;
        float4 gv = f4zero(), bv = f4zero();
        if constexpr (PROLN) {
          gv = ld4(Gs + k);
          bv = ld4(Bs + k);
        }
#pragma unroll
        for (int r = 0; r < RT; ++r)
          st4(&buf[(arow + 32 * r) * LDA + akq], a_finish<PRO, VEC, 0>(ast[r], ar[r], k, K, gv, bv));
        float* Wd = buf + BMS * LDA;
#pragma unroll
        for (int i = 0; i < NW4; ++i) {
          if (WL == 0) st4(&Wd[((st >> 3) + 32 * i) * LDWT + (st & 7) * 4], wr[i]);
          else st4(&Wd[((st + 256 * i) / (BN / 4)) * LDWT + ((st + 256 * i) % (BN / 4)) * 4], wr[i]);
        }
      };
      pipeline(issue, store);
    } else {
      auto issue = [&](ARaw(&ar)[RT], float4(&wr)[NW4], int k0) {
        issueA(ar, k0);
        loadW(k0, wr);
      };
      auto store = [&](float* buf, const ARaw(&ar)[RT], const float4(&wr)[NW4], int k0) {
        storeA(buf, ar, k0);
        storeW(buf + BMS * LDA, k0, wr);
      };
      pipeline(issue, store);
    }
  } else {
    // =============================== matrix waves ================================================
#pragma unroll
    for (int rr = 0; rr < RT; ++rr)
#pragma unroll
      for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rr][t][r] = 0.f;
    f32x16 accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
#pragma unroll
    for (int h = 0; h < HT; ++h)
#pragma unroll
      for (int t = 0; t < 2 * NTW; ++t) acch[h][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (PROLN) __syncthreads();
    __syncthreads();
    STAMP(1);
    if constexpr (HALF) {
      // 16x16x4 MFMA: lane l holds A[row l&15][k = 4*(l>>4) + j] and B[k][col l&15] for the j-th of the 4 MFMAs
      // of a float4; one float4 per lane covers 16 k values, a 32-wide chunk is two such steps.
      const int l15 = lane & 15, g4 = lane >> 4;
      struct Frag { float4 a[HT]; float b[2 * NTW][4]; };
      auto fetch = [&](Frag& f, const float* Asb, int kk) {
        const float* Wsb = Asb + BMS * LDA;
#pragma unroll
        for (int h = 0; h < HT; ++h) f.a[h] = ld4(&Asb[(16 * h + l15) * LDA + kk + 4 * g4]);
#pragma unroll
        for (int t = 0; t < 2 * NTW; ++t) {
          if (WL == 0) {
            const float4 v = ld4(&Wsb[((wave * 2 * NTW + t) * 16 + l15) * LDWT + kk + 4 * g4]);
            f.b[t][0] = v.x; f.b[t][1] = v.y; f.b[t][2] = v.z; f.b[t][3] = v.w;
          } else {
            const float* bp = &Wsb[(kk + 4 * g4) * LDWT + (wave * 2 * NTW + t) * 16 + l15];
            f.b[t][0] = bp[0]; f.b[t][1] = bp[LDWT]; f.b[t][2] = bp[2 * LDWT]; f.b[t][3] = bp[3 * LDWT];
          }
        }
      };
      auto mma = [&](const Frag& f) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int h = 0; h < HT; ++h) {
            const float av = c == 0 ? f.a[h].x : c == 1 ? f.a[h].y : c == 2 ? f.a[h].z : f.a[h].w;
#pragma unroll
            for (int t = 0; t < 2 * NTW; ++t)
              acch[h][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, f.b[t][c], acch[h][t], 0, 0, 0);
          }
      };
      if constexpr (TB) {
        Frag f0, f1;
        int cur = 0;
        fetch(f0, smem, 0);
        for (int kt = 0; kt < nk; ++kt) {
          const float* Asb = smem + cur * STAGE;
          cur = cur == 2 ? 0 : cur + 1;
          fetch(f1, Asb, 16);
          mma(f0);
          fetch(f0, smem + cur * STAGE, 0);      // chunk kt+1: stored before the previous barrier (stale after the last chunk, unused)
          mma(f1);
          __syncthreads();
        }
      } else {
        for (int kt = 0; kt < nk; ++kt) {
          const float* Asb = smem + (kt & 1) * STAGE;
          Frag f;
#pragma unroll
          for (int kk = 0; kk < BK; kk += 16) {
            fetch(f, Asb, kk);
            mma(f);
          }
          __syncthreads();
        }
      }
    } else {
    // One k-chunk of MFMAs.  Straight-line and unconditional: W rows / columns beyond N are zero-filled
    // in LDS, so a ragged last column block just multiplies zeros.  (A wave-uniform "skip my
    // out-of-range tiles" branch here made hipcc keep the accumulators in VGPRs across the k-loop and
    // copy them to/from AGPRs around the MFMA block: 2 x 16*RT*NTW moves per chunk.)
    for (int kt = 0; kt < nk; ++kt) {
      const float* Asb = smem + (kt & 1) * STAGE;
      const float* Wsb = Asb + BMS * LDA;
      STAMP(3 + 3 * kt);
#pragma unroll
      for (int kk = 0; kk < BK; kk += 8) {
        float4 a[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) a[r] = ld4(&Asb[(32 * r + l31) * LDA + kk + 4 * hh]);
        float b[NTW][4];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          if (WL == 0) {
            const float4 v = ld4(&Wsb[((wave * NTW + t) * 32 + l31) * LDWT + kk + 4 * hh]);
            b[t][0] = v.x; b[t][1] = v.y; b[t][2] = v.z; b[t][3] = v.w;
          } else {
            const float* bp = &Wsb[(kk + 4 * hh) * LDWT + (wave * NTW + t) * 32 + l31];
            b[t][0] = bp[0]; b[t][1] = bp[LDWT]; b[t][2] = bp[2 * LDWT]; b[t][3] = bp[3 * LDWT];
          }
        }
        if constexpr (RT * NTW == 1) {
          // a single tile per wave: alternate two accumulators, or the 16 MFMAs of a chunk form one dependent
          // chain and issue every ~87 clk instead of every 64
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0].x, b[0][0], acc[0][0], 0, 0, 0);
          accb = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0].y, b[0][1], accb, 0, 0, 0);
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0].z, b[0][2], acc[0][0], 0, 0, 0);
          accb = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0].w, b[0][3], accb, 0, 0, 0);
        } else {
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
      }
      STAMP(4 + 3 * kt);
      __syncthreads();
    }
    if constexpr (RT * NTW == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][0][r] += accb[r];
    }
    }   // !HALF
  }
  STAMP(55);

  if constexpr (!HOIST) prefetch_rows();
  // ---- accumulators -> LDS C tiles (alias the staging buffers; the k-loop ended with a barrier) ----
  if constexpr (HALF) {
    if (wave_u < 4) {                 // 16x16 C fragment: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
      for (int h = 0; h < HT; ++h)
#pragma unroll
        for (int t = 0; t < 2 * NTW; ++t) {
          const int col = (wave * 2 * NTW + t) * 16 + (lane & 15);
#pragma unroll
          for (int r = 0; r < 4; ++r) Cs[(16 * h + 4 * (lane >> 4) + r) * LDC + col] = acch[h][t][r];
        }
    }
  } else if (wave_u < 4) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const int col = (wave * NTW + t) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
          Cs[rt * CTILE + row * LDC + col] = acc[rt][t][r];
        }
      }
  }
  STAMP(56);
  __syncthreads();
  STAMP(57);

  if constexpr (epi == DOSX_EPI_LN && NTW <= 2) {          // (512-column tiles keep the full-wave version: 8 float4 per lane cost more than they save)
    // ---- LayerNorm epilogue, one QUARTER wave per row: four rows of a wave are reduced at once with 4-step DPP
    // sums (a 64-lane sum per row - DPP + 4 v_readlane + scalar adds, twice per row, rows one after the other - took
    // ~830 clk per row: 5000 clk of a 54000-clk kernel at 6 rows per wave)
    const int q16 = lane & 15, qd = lane >> 4;
#pragma unroll
    for (int rt = 0; rt < RTE; ++rt)
#pragma unroll
      for (int p = 0; p < (ER + 3) / 4; ++p) {
        const int li = 4 * p + qd;                       // this quarter's row among the wave's ER rows
        const int lr = wave * ER + (li < ER ? li : 0);
        const int r = m0 + 32 * rt + lr;
        const bool rv = li < ER && r < M;
        float4 v[KQ];
        float s1 = 0.f;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
          const int c = q16 * 4 + 64 * k;
          v[k] = f4zero();
          if (c < ncols) {
            v[k] = f4add(ld4(&Cs[rt * CTILE + lr * LDC + c]), biasq[k]);
            if (has_add) v[k] = f4add(v[k], f4add(adq[rt][p][0][k], adq[rt][p][1][k]));
            s1 += v[k].x + v[k].y + v[k].z + v[k].w;
          }
        }
        const float mean = row16_sum(s1) * invN;
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
          if (q16 * 4 + 64 * k < ncols) {
            v[k].x -= mean; v[k].y -= mean; v[k].z -= mean; v[k].w -= mean;
            s2 += v[k].x * v[k].x + v[k].y * v[k].y + v[k].z * v[k].z + v[k].w * v[k].w;
          }
        }
        const float rstd = rsqrtf(row16_sum(s2) * invN + DOSX_LN_EPS);
        if (rv) {
          float* const orow = g.out + (size_t)r * g.ldo + n0;
#pragma unroll
          for (int k = 0; k < KQ; ++k) {
            const int c = q16 * 4 + 64 * k;
            if (c < ncols) st4(orow + c, make_float4(v[k].x * rstd, v[k].y * rstd, v[k].z * rstd, v[k].w * rstd));
          }
          if (q16 == 0) g.aux_out[r] = rstd;
        }
      }
  } else {
#pragma unroll
  for (int rt = 0; rt < RTE; ++rt) {
  const int mb = m0 + 32 * rt;
  // ---- row-wise epilogue: lanes sweep the columns as float4 ------------------------------------
#pragma unroll
  for (int i = 0; i < ER; ++i) {
    const int lr = wave * ER + i, r = mb + lr;
    const bool rvalid = r < M;              // wave-uniform
    float4 v[CG];
#pragma unroll
    for (int j = 0; j < CG; ++j) v[j] = on[j] ? ld4(&Cs[rt * CTILE + lr * LDC + lane * 4 + 256 * j]) : f4zero();
    float* const orow = g.out + orow_[rt][i];
    if (epi == DOSX_EPI_BIAS_ACT) {
      float s1 = 0.f;
#pragma unroll
      for (int j = 0; j < CG; ++j) {
        if (!on[j]) continue;
        v[j] = f4add(v[j], biasv[j]);
        if (g.res_pre) v[j] = f4add(v[j], pv1[rt][i][j]);       // pre-activation row term (zeros when absent)
        if (g.act == 1) {
          v[j].x = fmaxf(v[j].x, 0.f); v[j].y = fmaxf(v[j].y, 0.f);
          v[j].z = fmaxf(v[j].z, 0.f); v[j].w = fmaxf(v[j].w, 0.f);
        } else if (g.act == 2) {
          const float sl = g.act_slope;
          v[j].x = v[j].x >= 0.f ? v[j].x : sl * v[j].x; v[j].y = v[j].y >= 0.f ? v[j].y : sl * v[j].y;
          v[j].z = v[j].z >= 0.f ? v[j].z : sl * v[j].z; v[j].w = v[j].w >= 0.f ? v[j].w : sl * v[j].w;
        }
        if (!g.res_pre) v[j] = f4add(v[j], pv1[rt][i][j]);   // residual (zeros when absent)
        if (rvalid) st4(orow + gcol[j], v[j]);
        s1 += v[j].x + v[j].y + v[j].z + v[j].w;
      }
      if (g.stats_out || g.norm_out) {
        const float mean = wave_sum(s1) * invN;
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < CG; ++j) {
          if (!on[j]) continue;
          const float a = v[j].x - mean, b = v[j].y - mean, c2 = v[j].z - mean, d = v[j].w - mean;
          s2 += a * a + b * b + c2 * c2 + d * d;
        }
        const float var = wave_sum(s2) * invN;
        const float rstd = rsqrtf(var + DOSX_LN_EPS);
        if (g.stats_out && lane == 0 && rvalid) {
          g.stats_out[2 * (size_t)r] = mean;
          g.stats_out[2 * (size_t)r + 1] = rstd;
        }
        if (g.norm_out && rvalid) {          // the normalised rows next to the plain ones (same mapped row, same ld)
          float* const nrow = g.norm_out + orow_[rt][i];
#pragma unroll
          for (int j = 0; j < CG; ++j)
            if (on[j]) st4(nrow + gcol[j], make_float4((v[j].x - mean) * rstd, (v[j].y - mean) * rstd, (v[j].z - mean) * rstd,
                                                       (v[j].w - mean) * rstd));
          if (lane == 0) g.norm_rstd[dosx_map_row(g.out_map, r)] = rstd;
        }
      }
    } else if (epi == DOSX_EPI_LN) {
      float s1 = 0.f;
      if (has_add) {                   // (512-column tiles: the gathered addends of this row, requested together)
        const int rc = min(r, M - 1);
        const int ip = g.add_ip[rc], iq = g.add_iq[rc];
        float4 ap[CG], aq[CG];
#pragma unroll
        for (int j = 0; j < CG; ++j) {
          ap[j] = ld4(g.add_p + (size_t)ip * g.ld_add + gcol[j]);
          aq[j] = ld4(g.add_q + (size_t)iq * g.ld_add + gcol[j]);
        }
#pragma unroll
        for (int j = 0; j < CG; ++j) v[j] = f4add(v[j], f4add(ap[j], aq[j]));
      }
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
    } else if (epi == DOSX_EPI_SEGSUM) {
      // edge residual e' = e + msg (DOSTransformer_phonon.py:84); msg itself (acc + bias) never goes to HBM: its only other
      // consumer is the segment sum below, which reads it from the LDS tile
      if (g.out != nullptr) {
#pragma unroll
        for (int j = 0; j < CG; ++j) {
          if (!on[j] || !rvalid) continue;
          st4(orow + gcol[j], f4add(f4add(v[j], biasv[j]), pv1[rt][i][j]));
        }
      }
    } else if (epi == DOSX_EPI_RELU_MASK) {
#pragma unroll
      for (int j = 0; j < CG; ++j) {
        if (!on[j] || !rvalid) continue;
        const float4 h = pv1[rt][i][j];
        st4(orow + gcol[j], make_float4(h.x > 0.f ? v[j].x : 0.f, h.y > 0.f ? v[j].y : 0.f,
                                        h.z > 0.f ? v[j].z : 0.f, h.w > 0.f ? v[j].w : 0.f));
      }
    } else if (epi == DOSX_EPI_PRELU_BWD) {
#pragma unroll
      for (int j = 0; j < CG; ++j) {
        if (!on[j] || !rvalid) continue;
        const float4 z = pv1[rt][i][j];
        float4 o;
        o.x = z.x >= 0.f ? v[j].x : e_alpha * v[j].x; if (z.x < 0.f) pal += v[j].x * z.x;
        o.y = z.y >= 0.f ? v[j].y : e_alpha * v[j].y; if (z.y < 0.f) pal += v[j].y * z.y;
        o.z = z.z >= 0.f ? v[j].z : e_alpha * v[j].z; if (z.z < 0.f) pal += v[j].z * z.z;
        o.w = z.w >= 0.f ? v[j].w : e_alpha * v[j].w; if (z.w < 0.f) pal += v[j].w * z.w;
        st4(orow + gcol[j], o);
      }
    } else {  // DOSX_EPI_PRELU_LN_BWD or DOSX_EPI_ROWLN_BWD : LayerNorm backward over the full row
      const float mean = is_prelu_ln ? 0.f : st0[rt][i];
      const float rstd = is_prelu_ln ? st0[rt][i] : st1[rt][i];
      float4 xh[CG], dxh[CG];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < CG; ++j) {
        xh[j] = f4zero(); dxh[j] = f4zero();
        if (!on[j] || !rvalid) continue;
        float4 a = pv1[rt][i][j];
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
        if (is_rowln) o = f4add(o, pv2[rt][i][j]);
        st4(orow + gcol[j], o);
        if constexpr (epi == DOSX_EPI_PRELU_LN_BWD_SEG) st4(&Cs[rt * CTILE + lr * LDC + lane * 4 + 256 * j], o);   // dz row for the node sums below
      }
    }
  }

  }   // rt
  }   // epilogues other than LN
  if constexpr (epi == DOSX_EPI_PRELU_LN_BWD_SEG) __syncthreads();       // the C tile now holds dz: every wave's rows are in place
  if constexpr (SEGT) {
    // ---- segment sums: agg[n] = scale[n] * sum_{e in seg(n)} (acc[e] + bias)  (scatter_mean / scatter_sum by `col`,
    // DOSTransformer_phonon.py:209 / DOSTransformer.py:187), one wave per node, rows in order (the summation order of the
    // stand-alone segment_reduce kernel's row loop, so the tiling never changes a bit).  Rows of a node outside this tile
    // can only be ghost rows of the first padding node: clipped.
    // An OVER-FULL node (more than BMR incoming edges) is cut into chunks of BMR rows, one tile each (batch.seg_tiles_host):
    // chunk info != 0 says this tile's first node, nlo_, is such a node and which chunk this is.  Its rows here (clipped
    // to the tile) give a chunk sum, published write-through to seg_part[tile]; a ticket on the counter of the node's FIRST
    // tile tells the last arriving tile, which adds the chunk sums in chunk order (a fixed order: deterministic) and writes
    // the aggregate.  One wave does all of it, so the hand-off needs no workgroup barrier.  Whole nodes follow from nlo_ + 1.
    const int pinfo = g.seg_tile[2 * (g.seg_ntiles + 1) + bx];
    const int nfirst = pinfo ? nlo_ + 1 : nlo_;
    if (pinfo && wave == 7 && g.seg_part != nullptr) {
      const int n = nlo_, ci = pinfo >> 16, nc = pinfo & 0xffff, t0 = bx - ci;
      const int re = min(g.seg_rowptr[n + 1], M) - m0;                  // (rows 0 .. re of the tile: the chunk starts the tile)
      const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)g.seg_part, 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int j = 0; j < CG; ++j) {
        if (!on[j]) continue;
        float4 t = f4zero();
        for (int r = 0; r < re; ++r) t = f4add(t, f4add(ld4(&Cs[r * LDC + lane * 4 + 256 * j]), biasv[j]));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i32, t), rP, (uint32_t)(((size_t)bx * N + gcol[j]) * 4), 0, 16);   // sc1
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int tk = 0;
      if (lane == 0) tk = dosx_ticket(g.seg_cnt + t0);
      tk = __builtin_amdgcn_readfirstlane(tk);
      if (tk == nc - 1) {                                                 // every chunk of the node has been published
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float sc = g.seg_scale ? g.seg_scale[n] : 1.f;
#pragma unroll
        for (int j = 0; j < CG; ++j) {
          if (!on[j]) continue;
          float4 t = f4zero();
          for (int c = 0; c < nc; ++c) {
            const float4 p = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                rP, (uint32_t)(((size_t)(t0 + c) * N + gcol[j]) * 4), 0, 16));                            // sc1
            t = c == 0 ? p : f4add(t, p);
          }
          st4(g.seg_agg + (size_t)n * N + gcol[j], make_float4(t.x * sc, t.y * sc, t.z * sc, t.w * sc));
        }
        if (lane == 0) __hip_atomic_store(g.seg_cnt + t0, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    for (int n = nfirst + wave; n < nhi_; n += 8) {
      const int rb = max(g.seg_rowptr[n], m0) - m0, re = min(g.seg_rowptr[n + 1], M) - m0;
      const float sc = g.seg_scale ? g.seg_scale[n] : 1.f;
#pragma unroll
      for (int j = 0; j < CG; ++j) {
        if (!on[j]) continue;
        float4 t = f4zero();
        for (int r = rb; r < re; ++r) t = f4add(t, f4add(ld4(&Cs[r * LDC + lane * 4 + 256 * j]), biasv[j]));
        st4(g.seg_agg + (size_t)n * N + gcol[j], make_float4(t.x * sc, t.y * sc, t.z * sc, t.w * sc));
      }
    }
  }
  STAMP(58);
  // ---- per-workgroup partial sums for the parameter gradients of the fused LN / PReLU ----------
  if (g.partials && (is_prelu_ln || epi == DOSX_EPI_ROWLN_BWD || epi == DOSX_EPI_PRELU_BWD)) {
    float* prow = g.partials + (size_t)((L.part_base + bx) * gy + by) * g.partial_ld;
    __syncthreads();                    // (the C tile rows of other waves are still being read above)
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
      if (lane == 0) Ps[16 * BN + wave] = s;
    }
    __syncthreads();
    if (epi != DOSX_EPI_PRELU_BWD) {
      for (int c = tid; c < 2 * BN; c += 512) {
        const int which = c / BN, col = c % BN;
        if (col < ncols) {
          float s = 0.f;
#pragma unroll
          for (int w = 0; w < 8; ++w) s += Ps[(w * 2 + which) * BN + col];
          prow[which * N + n0 + col] = s;
        }
      }
    }
    if (epi != DOSX_EPI_ROWLN_BWD && tid == 0) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += Ps[16 * BN + w];
      prow[g.partial_ld - 1] = s;
    }
  }
}

template <int RTP, int NTW, int WL, int PRO, int VEC, int EPI>
__global__ __launch_bounds__(512, (NTW == 1 && (EPI == DOSX_EPI_BIAS_ACT || EPI == DOSX_EPI_LN || EPI == DOSX_EPI_RELU_MASK)) ? 4 : 2)
void gemm_kernel(const GemmLaunch L) {
  gemm_body<RTP, NTW, WL, PRO, VEC, EPI>(L, (int)blockIdx.x);
}

// TWO independent problems of the same tile configuration in ONE grid (dosx_gemm_pair: the two output heads `fc` and
// `fc_prompt`, DOSTransformer_phonon.py:93-109 - same N, disjoint output rows): workgroups [0, nblk1) run the first, the rest
// the second.  Each head alone is one partial round of workgroups (102 of 32 rows at the benchmark shape); together they are
// still one round, so the pair costs one kernel's latency instead of two + a launch.  Plain epilogue, W[N,K], aligned operands.
template <int RTP, int NTW>
__global__ __launch_bounds__(512, NTW == 1 ? 4 : 2)
void gemm_pair_kernel(const GemmLaunch L1, const GemmLaunch L2, const int nblk1) {
  const bool second = (int)blockIdx.x >= nblk1;
  gemm_body<RTP, NTW, 0, DOSX_PRO_NONE, 1, DOSX_EPI_BIAS_ACT>(second ? L2 : L1, second ? (int)blockIdx.x - nblk1 : (int)blockIdx.x);
}

// ONE problem, TWO tile heights in one grid (gemm_tail_split): workgroups [0, nblk1) own the rows of the full rounds as
// RTA-high tiles (L1: rows [0, L1.g.M)), the rest the remaining rows as RTB-high tiles (L2: rows [L2.m_base, M)).  The small
// tiles are dispatched last and share their CUs with the last large ones instead of forming a nearly empty round of their own.
template <int RTA, int RTB, int NTW, int WL, int PRO, int VEC, int EPI>
__global__ __launch_bounds__(512, (NTW == 1 && (EPI == DOSX_EPI_BIAS_ACT || EPI == DOSX_EPI_LN || EPI == DOSX_EPI_RELU_MASK)) ? 4 : 2)
void gemm_mixed_kernel(const GemmLaunch L1, const GemmLaunch L2, const int nblk1) {
  if ((int)blockIdx.x < nblk1) gemm_body<RTA, NTW, WL, PRO, VEC, EPI>(L1, (int)blockIdx.x);
  else gemm_body<RTB, NTW, WL, PRO, VEC, EPI>(L2, (int)blockIdx.x - nblk1);
}

template <int RTP, int NTW, int WL, int PROLN>
constexpr size_t gemm_smem_bytes() {
  constexpr int RT = RTP == 0 ? 1 : (RTP == 3 ? 2 : RTP);
  constexpr int BN = 128 * NTW;
  constexpr int LDWT = (WL == 0) ? (BK + 4) : (BN + 4);
  constexpr int WROWS = (WL == 0) ? BN : BK;
  constexpr int STAGE = BM * RT * LDA + WROWS * LDWT;
  constexpr int CTILE = BM * (BN + 4);
  constexpr int NB = ((RTP == 0 || RTP == 3) && NTW == 2) ? 3 : 2;      // (TB in gemm_kernel)
  constexpr int MAINF = NB * STAGE + (PROLN ? 2 * GEMM_KMAX : 0);
  constexpr int EPIF = RT * CTILE + 16 * BN + 8;
  return (size_t)(MAINF > EPIF ? MAINF : EPIF) * sizeof(float);
}

template <int RT, int NTW, int WL, int PRO, int VEC, int EPI>
int launch_gemm3(const GemmLaunch& L, hipStream_t s) {
  constexpr int BN = 128 * NTW;
  dim3 grid(ceil_div(L.g.M - L.m_base, RT == 0 ? 16 : (RT == 3 ? 48 : BM * RT)) * ceil_div(L.g.N, BN));
  if (EPI == DOSX_EPI_SEGSUM || EPI == DOSX_EPI_PRELU_LN_BWD_SEG) grid = dim3(L.g.seg_ntiles);          // one workgroup per node-aligned row tile
  constexpr size_t smem = gemm_smem_bytes<RT, NTW, WL, (VEC && (PRO == DOSX_PRO_LN_PRELU || PRO == DOSX_PRO_ROWLN)) ? 1 : 0>();
  if constexpr (smem > 160 * 1024) {     // (512-column tile + LayerNorm prologue: no caller has this shape)
    dosx_set_error("dosx_gemm: tile %dx%d with prologue %d exceeds the 160 KB LDS", BM * RT, BN, PRO);
    return -22;
  } else {
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<RT, NTW, WL, PRO, VEC, EPI>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<RT, NTW, WL, PRO, VEC, EPI>), grid, dim3(512), smem, s, L);
    DOSX_LAUNCH_CHECK();
    return 0;
  }
}

template <int NTW, int WL, int PRO, int VEC, int EPI>
int launch_gemm(const GemmLaunch& L, hipStream_t s) {
  if constexpr (NTW < 4) {              // (two 64-row stage buffers of a 512-column tile exceed the LDS)
    if (L.rt == 2) return launch_gemm3<2, NTW, WL, PRO, VEC, EPI>(L, s);
    if (L.rt == 3) return launch_gemm3<3, NTW, WL, PRO, VEC, EPI>(L, s);
  }
  if (L.rt == 0) return launch_gemm3<0, NTW, WL, PRO, VEC, EPI>(L, s);
  return launch_gemm3<1, NTW, WL, PRO, VEC, EPI>(L, s);
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
    if (epi == DOSX_EPI_SEGSUM) {
      if constexpr (NTW <= 2) {
        if (pro == DOSX_PRO_LN_PRELU) return launch_gemm3<3, NTW, 0, DOSX_PRO_LN_PRELU, 1, DOSX_EPI_SEGSUM>(L, s);
      }
      dosx_set_error("dosx_gemm: EPI_SEGSUM exists for the LN_PRELU prologue and N <= 256 only");
      return -22;
    }
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
      case DOSX_EPI_PRELU_LN_BWD_SEG:
        if constexpr (NTW <= 2) return launch_gemm3<3, NTW, 1, DOSX_PRO_NONE, 1, DOSX_EPI_PRELU_LN_BWD_SEG>(L, s);
        break;
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
  const bool full_row = (epi == DOSX_EPI_LN || epi == DOSX_EPI_PRELU_LN_BWD || epi == DOSX_EPI_ROWLN_BWD || epi == DOSX_EPI_PRELU_LN_BWD_SEG);
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
  static int forced = -2, half_max = -1, ht3 = 1;
  if (forced == -2) {
    const char* e = getenv("DOSX_GEMM_RT");
    forced = e ? atoi(e) : -1;
    const char* h = getenv("DOSX_GEMM_HALF_MAX");
    half_max = h ? atoi(h) : 128;
    const char* t = getenv("DOSX_GEMM_HT3");
    ht3 = t ? atoi(t) : 1;
  }
  if (M <= 16) return 0;
  if (forced >= 0 && forced <= 3 && !(forced >= 2 && gemm_bn(M, N, epi) == 512)) return M <= BM && forced >= 2 ? 1 : forced;
  const int ntiles = ceil_div(N, gemm_bn(M, N, epi));
  const int wg1 = ceil_div(M, BM) * ntiles, wg2 = ceil_div(M, 2 * BM) * ntiles;
  // HALF (16-row) tiles: a kernel this small is one partial round of workgroups whichever way it is cut, so its
  // duration is one workgroup's latency - halve that
  if (wg1 <= half_max) return 0;
  if (M <= BM) return 1;
  if (gemm_bn(M, N, epi) == 512) return 1;
  // one column tile, more 32-row blocks than CUs: 48-row workgroups (three 16-row sub-tiles) if those fit one round -
  // the busiest CU then holds 48 rows instead of 64 (M = 9000: 188 workgroups instead of 282 / 141)
  if (ht3 && ntiles == 1 && wg1 > 256 && ceil_div(M, 48) <= 256) return 3;
  if (wg2 >= 192) return 2;
  if (wg1 > 256 && wg1 <= 400 && wg2 >= 128) return 2;
  return 1;
}

// TAIL SPLIT (round 4).  A grid of W workgroups on 256 CUs runs ceil(W / 256) rounds: 804 tiles of 64 x 128 (the Electron-DOS
// feed-forward GEMMs, M = 25728) take 4 rounds where 3.14 would do - measured 137.0 us against 99.5 us for M = 24576 (768 tiles:
// 3 full rounds), i.e. 37 us for the last 4.7 % of the rows (profiles/r04_ab_gemm_tiles.log).  Such a call becomes TWO launches:
// the rows of the full rounds with the large tile, then the remaining rows as their own small problem under the normal tile
// policy (16- / 32- / 48-row tiles: one short round).  Returns the first row of the tail (0 = no split).
// RESULT of the first form - TWO launches, the tail behind the full rounds - (profiles/r04_ab_gemm_split.log): a tile of a quarter of the rows is NOT a quarter of the time - its k-loop is the same
// 32 chunks of barrier + weight-tile staging - so the 1152-row tail takes 26 us instead of 37 (fc2 forward 137.7 -> 126.5 us,
// fc1 input gradient 130.8 -> 120.4), the M = 12864 calls lose (65.9 -> 74.0 us) and the Electron-DOS step is 1.8 % SLOWER
// (7.54 -> 7.67 ms, four interleaved pairs), configs[4]'s shard 3.7 %.  SECOND form (shipped): the same split in ONE launch -
// gemm_mixed_kernel, the tail's 32- / 48-row tiles at the end of the grid, no launch boundary in between: the same kernel
// times (fc2 forward 138.9 -> 126.5 us, fc1 input gradient 129.9 -> 119.5), and the steps no longer lose: Electron-DOS
// 7.40 ms either way (its forward tails already run as a concurrent chain), configs[4]'s shard 7.28 -> 7.23 ms
// (profiles/r04_ab_gemm_mixed.log).  DOSX_GEMM_SPLIT=0 switches it off.
inline int gemm_tail_split(int M, int N, int epi) {
  static int on = -1, max_tail = 160;
  if (on < 0) {
    const char* e = getenv("DOSX_GEMM_SPLIT");
    on = e ? atoi(e) : 1;          // (the one-launch mixed-height form; the two-launch form lost 2-4 % per step)
    const char* t = getenv("DOSX_GEMM_SPLIT_MAXTAIL");
    if (t) max_tail = atoi(t);
  }
  // (the mixed-height kernel exists for the plain / ReLU-mask epilogues on 128-column tiles and for the LayerNorm-backward
  //  epilogue on one 256-column tile: the feed-forward GEMMs of a hidden-256 model and the plain large GEMMs around them)
  if (!on) return 0;
  const int bn_ = gemm_bn(M, N, epi);
  if (bn_ == 512 && (epi == DOSX_EPI_LN || epi == DOSX_EPI_PRELU_LN_BWD) && gemm_rt(M, N, epi) == 1) {
    // Round 5: the 512-column row-epilogue GEMMs of the hidden-256 message passing (E = 17880 edge rows: 559 tiles of 32 rows, one
    // workgroup per CU = 2.18 rounds run as 3 - 64.9 us against 49.8 at E = 16384, 81.1 / 61.3 with the PReLU-LayerNorm-backward
    // epilogue; the 32-crystal shard: 42.8 against 23.9 us, tools/exp/r5_quant512.py): the rows of the full rounds as 32-row tiles,
    // the rest as 16-row tiles (16 x 16 x 4 MFMA) behind them in the same grid
    const int full = M / (256 * BM), tail_wg = ceil_div(M - full * 256 * BM, BM);
    if (full < 1 || tail_wg == 0 || tail_wg > max_tail) return 0;
    return full * 256 * BM;
  }
  if (!((bn_ == 128 && (epi == DOSX_EPI_BIAS_ACT || epi == DOSX_EPI_RELU_MASK)) || (bn_ == 256 && epi == DOSX_EPI_ROWLN_BWD))) return 0;
  if (gemm_rt(M, N, epi) != 2) return 0;                   // the large-problem regime only (64-row tiles)
  const int gy = ceil_div(N, gemm_bn(M, N, epi));
  if (256 % gy) return 0;
  const int rows_round = 256 / gy * 2 * BM;                // rows of one full round of 64-row tiles
  const int full = M / rows_round;
  const int tail_wg = ceil_div(M - full * rows_round, 2 * BM) * gy;
  if (full < 1 || tail_wg == 0 || tail_wg > max_tail) return 0;
  return full * rows_round;
}

// tile height (in gemm_kernel's RT code) of the tail rows of a split call: 48-row tiles where the normal policy picks them
// (LayerNorm-backward epilogue only), 32-row tiles otherwise
inline int gemm_tail_rt(int Mt, int N, int epi) {
  if (epi == DOSX_EPI_LN || epi == DOSX_EPI_PRELU_LN_BWD) return 0;       // (the 512-column split: 16-row tiles)
  return (epi == DOSX_EPI_ROWLN_BWD && gemm_rt(Mt, N, epi) == 3) ? 3 : 1;
}
// rows per workgroup of the HEAD part of a split call (64-row tiles; 32-row tiles for the 512-column epilogues)
inline int gemm_head_rows(int epi) { return (epi == DOSX_EPI_LN || epi == DOSX_EPI_PRELU_LN_BWD) ? BM : 2 * BM; }
inline int gemm_tail_rows(int rt) { return rt == 0 ? 16 : (rt == 3 ? 48 : BM); }

inline int gemm_rows_per_wg(int M, int N, int epi) {
  const int rt = gemm_rt(M, N, epi);
  return rt == 0 ? 16 : (rt == 3 ? 48 : BM * rt);
}

// tile / vector-path selection of one GEMM: fills L, returns the column tile (128 / 256 / 512)
static int gemm_plan(const DosxGemm& g, GemmLaunch& L) {
  L.g = g;
  L.vecA = seg_vec_ok(g.a, g.nseg) && (g.K & 3) == 0;
  if (g.pro == DOSX_PRO_LN_PRELU || g.pro == DOSX_PRO_ROWLN)
    L.vecA = L.vecA && aligned16(g.pro_gamma) && aligned16(g.pro_beta);
  L.vecW = ((g.ldw & 3) == 0) && aligned16(g.w) && (g.w_layout == 0 ? (g.K & 3) == 0 : (g.N & 3) == 0);
  L.rt = gemm_rt(g.M, g.N, g.epi);
  int bn = gemm_bn(g.M, g.N, g.epi);
  if ((g.stats_out || g.norm_out) && bn < g.N) bn = g.N <= 256 ? 256 : 512;
  if (bn == 512 && L.rt >= 2) L.rt = 1;
  if (g.epi == DOSX_EPI_SEGSUM || g.epi == DOSX_EPI_PRELU_LN_BWD_SEG) {          // node-aligned 48-row tiles, one column tile
    L.rt = 3;
    bn = g.N <= 128 ? 128 : 256;
  }
  return bn;
}

// ---- vector-ALU "sliver" GEMM (round 4) ----------------------------------------------------------------------------------
// C[M,N] = A[M,K] . W[K,N] (+ R[M,N]) for the SMALL plain dgrad GEMMs of the backward pass (w_layout 1, no prologue, no
// bias / activation; up to DOSX_SLIVER_MAX_GF GF - an experiment, OFF by default): 256 threads = 16 x 16, a 64 x 64 output tile, 4 x 4 per thread, k-chunks of 16
// through 8.5 KB of LDS (A chunk stored k-major), the next chunk prefetched into registers, PACKED fp32 FMAs on the vector ALU
// (v_pk_fma_f32: the same 157 TF/s peak as the fp32 MFMA), raised wave priority.  Why: these kernels run while a
// weight-gradient group owns the chip - two workgroups of 8 waves / 107 VGPRs / 75 KB of LDS per CU, their matrix waves
// issuing MFMAs back to back.  An MFMA kernel of the chain then waits for an EMPTY CU (it needs > 100 KB of LDS or > 128
// VGPRs), and even a tiny co-resident MFMA workgroup gets ~1/15 of the matrix pipe (tools/exp/sliver_probe.py: a 10-us MFMA
// probe takes 150-180 us behind a group whatever its footprint and priority).  This workgroup - 4 waves, 56 VGPRs, 8.5 KB -
// FITS next to the two resident weight-gradient workgroups and computes on the pipe they leave idle: the same probe on the
// vector ALU with s_setprio 3 takes 31-34 us; a 6528 x 128 x 128 dgrad 32 us behind a group against 170-190 us for the MFMA
// kernel, the 1554 x 512 x 512 node-MLP dgrad 105 against 200-215 us (profiles/r04_sliver_probe.log).
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int SG_T = 64, SG_K = 16;
__global__ __launch_bounds__(256) void sliver_gemm_kernel(const DosxGemm g) {
  __builtin_amdgcn_s_setprio(3);
  __shared__ __align__(16) float As[SG_K][SG_T + 4];
  __shared__ __align__(16) float Bs[SG_K][SG_T + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int M = g.M, N = g.N, K = g.K;
  const int m0 = blockIdx.x * SG_T, n0 = blockIdx.y * SG_T;
  // staging: A chunk = 64 rows x 16 k (thread -> row tid / 4, k (tid % 4) * 4 ..+3), W chunk = 16 k x 64 n (thread -> k tid / 16, n (tid % 16) * 4)
  const int ar = tid >> 2, ak = (tid & 3) * 4;
  const int wk = tid >> 4, wn = (tid & 15) * 4;
  const float* ap = g.a[0].p + (size_t)dosx_map_row(g.a[0].map, min(m0 + ar, M - 1)) * (size_t)g.a[0].ld + ak;
  const float* wp = g.w + (size_t)wk * g.ldw + min(n0 + wn, N - 4);
  float4 ra = ld4(ap), rw = ld4(wp);
  f32x2 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { acc[i][0] = f32x2{0.f, 0.f}; acc[i][1] = f32x2{0.f, 0.f}; }
  for (int k0 = 0; k0 < K; k0 += SG_K) {
    __syncthreads();
    As[ak + 0][ar] = ra.x; As[ak + 1][ar] = ra.y; As[ak + 2][ar] = ra.z; As[ak + 3][ar] = ra.w;
    st4(&Bs[wk][wn], rw);
    __syncthreads();
    if (k0 + SG_K < K) {
      ra = ld4(ap + k0 + SG_K);
      rw = ld4(wp + (size_t)(k0 + SG_K) * g.ldw);
    }
#pragma unroll
    for (int k = 0; k < SG_K; ++k) {
      const float4 a = ld4(&As[k][ty * 4]);
      const float4 b = ld4(&Bs[k][tx * 4]);
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
      if (g.res) o = f4add(o, ld4(g.res + (size_t)r * g.ldr + c));
      st4(g.out + (size_t)r * g.ldo + c, o);
    }
  }
}

// which calls take the sliver kernel (host side): plain dgrad GEMMs small enough that latency under a weight-gradient group,
// not throughput, is what they cost.  DOSX_SLIVER_MAX_GF / dosx_set_sliver_max_gf: the flop limit in GF; DEFAULT 0 = never.
// Measured in the step (tools/exp/ab_sliver*.sh, tools/exp/sliver_sites.sh): the routed kernels do get faster
// (Electron-DOS: the head dgrads 237 -> 101 us, the node-MLP dgrad 99 -> 65 us) but the kernels BEHIND them then wait longer
// for the same weight-gradient group - the step moves by -0.2 .. -0.4 % (Electron-DOS) and by +-1 % box-to-box noise (Phonon-DOS).
static double g_sliver_max_flop = -1.0;
static bool sliver_ok(const DosxGemm& g) {
  if (g_sliver_max_flop < 0.0) {
    const char* e = getenv("DOSX_SLIVER_MAX_GF");
    g_sliver_max_flop = (e ? atof(e) : 0.0) * 1e9;
  }
  const double max_flop = g_sliver_max_flop;
  if (max_flop <= 0.0 || 2.0 * g.M * (double)g.N * g.K > max_flop) return false;
  const bool ident_out = g.out_map.d >= (1 << 30) && g.out_map.idx == nullptr && g.out_map.c == 1 && g.out_map.off == 0;
  const bool ident_res = !g.res || (g.res_map.d >= (1 << 30) && g.res_map.idx == nullptr && g.res_map.c == 1 && g.res_map.off == 0 &&
                                    g.res_col0 == 0 && (g.ldr & 3) == 0 && aligned16(g.res));
  return g.w_layout == 1 && g.pro == DOSX_PRO_NONE && g.epi == DOSX_EPI_BIAS_ACT && g.act == 0 && !g.bias &&
         g.nseg == 1 && !g.stats_out && !g.norm_out && !g.aux_out && g.out && ident_out && ident_res && (g.K % SG_K) == 0 &&
         (g.N & 3) == 0 && g.N >= 4 && g.M >= 1 && (g.a[0].ld & 3) == 0 && aligned16(g.a[0].p) && (g.ldw & 3) == 0 &&
         aligned16(g.w) && (g.ldo & 3) == 0 && aligned16(g.out);
}

extern "C" int dosx_set_sliver_max_gf(double gf) {      // experiments / tests: route plain dgrad GEMMs of up to `gf` GF to sliver_gemm_kernel
  g_sliver_max_flop = gf > 0.0 ? gf * 1e9 : 0.0;
  return 0;
}

// The device symbol dosx_gemm would launch for this descriptor, as rocprofv3 prints it ("gemm_kernel<RT, NTW, WL, PRO,
// VEC, EPI>"): lets a profiler-side tool (bench.py's roofline, tools/pmc_traffic.py) tie a call site to its kernel.
extern "C" int dosx_gemm_kernel_name(const DosxGemm* gp, char* buf, int n) {
  DOSX_CHECK_ARG(gp && buf && n > 0, "dosx_gemm_kernel_name: bad args");
  if (sliver_ok(*gp)) {
    snprintf(buf, (size_t)n, "sliver_gemm_kernel");
    return 0;
  }
  GemmLaunch L;
  const int bn = gemm_plan(*gp, L);
  const int vec = L.vecA && L.vecW;
  snprintf(buf, (size_t)n, "gemm_kernel<%d, %d, %d, %d, %d, %d>", L.rt, bn / 128, gp->w_layout, vec ? gp->pro : 0, vec,
           vec ? gp->epi : 0);
  return 0;
}

extern "C" int dosx_gemm_partial_rows(int M, int N, int epi) {
  // one partial row per workgroup; row-wise epilogues run as one N tile (N <= 512), the
  // element-wise PRELU_BWD epilogue tiles N by 128.
  const int mA = gemm_tail_split(M, N, epi);
  const int rows = mA ? mA / gemm_head_rows(epi) + ceil_div(M - mA, gemm_tail_rows(gemm_tail_rt(M - mA, N, epi)))
                      : ceil_div(M, gemm_rows_per_wg(M, N, epi));
  if (epi == DOSX_EPI_PRELU_BWD) return rows * ceil_div(N, 128);
  return rows;
}

template <int RTB, int NTW, int WL, int PRO, int EPI, int RTA = 2>
static int launch_gemm_mixed3(const GemmLaunch& A, const GemmLaunch& T, hipStream_t s) {
  constexpr int BN = 128 * NTW, ROWS_A = BM * RTA, ROWS_B = RTB == 0 ? 16 : (RTB == 3 ? 48 : BM * RTB);
  const int n1 = ceil_div(A.g.M, ROWS_A) * ceil_div(A.g.N, BN), n2 = ceil_div(T.g.M - T.m_base, ROWS_B) * ceil_div(T.g.N, BN);
  constexpr int PROLN = (PRO == DOSX_PRO_LN_PRELU || PRO == DOSX_PRO_ROWLN) ? 1 : 0;
  constexpr size_t sa = gemm_smem_bytes<RTA, NTW, WL, PROLN>(), sb = gemm_smem_bytes<RTB, NTW, WL, PROLN>();
  constexpr size_t smem = sa > sb ? sa : sb;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mixed_kernel<RTA, RTB, NTW, WL, PRO, 1, EPI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_mixed_kernel<RTA, RTB, NTW, WL, PRO, 1, EPI>), dim3(n1 + n2), dim3(512), smem, s, A, T, n1);
  DOSX_LAUNCH_CHECK();
  return 0;
}

// -1: no mixed-height kernel for this combination (the caller launches the plain grid)
static int launch_gemm_mixed(int bn, const GemmLaunch& A, const GemmLaunch& T, hipStream_t s) {
  const DosxGemm& g = A.g;
  if (bn == 128 && T.rt == 1) {
    if (g.epi == DOSX_EPI_BIAS_ACT && g.w_layout == 0 && g.pro == DOSX_PRO_NONE)
      return launch_gemm_mixed3<1, 1, 0, DOSX_PRO_NONE, DOSX_EPI_BIAS_ACT>(A, T, s);
    if (g.epi == DOSX_EPI_BIAS_ACT && g.w_layout == 0 && g.pro == DOSX_PRO_ROWLN)
      return launch_gemm_mixed3<1, 1, 0, DOSX_PRO_ROWLN, DOSX_EPI_BIAS_ACT>(A, T, s);
    if (g.epi == DOSX_EPI_BIAS_ACT && g.w_layout == 1 && g.pro == DOSX_PRO_NONE)
      return launch_gemm_mixed3<1, 1, 1, DOSX_PRO_NONE, DOSX_EPI_BIAS_ACT>(A, T, s);
    if (g.epi == DOSX_EPI_RELU_MASK && g.w_layout == 1 && g.pro == DOSX_PRO_NONE)
      return launch_gemm_mixed3<1, 1, 1, DOSX_PRO_NONE, DOSX_EPI_RELU_MASK>(A, T, s);
  }
  if (bn == 256 && g.epi == DOSX_EPI_ROWLN_BWD && g.w_layout == 1 && g.pro == DOSX_PRO_NONE) {
    if (T.rt == 1) return launch_gemm_mixed3<1, 2, 1, DOSX_PRO_NONE, DOSX_EPI_ROWLN_BWD>(A, T, s);
    if (T.rt == 3) return launch_gemm_mixed3<3, 2, 1, DOSX_PRO_NONE, DOSX_EPI_ROWLN_BWD>(A, T, s);
  }
  if (bn == 512 && T.rt == 0 && A.rt == 1 && g.pro == DOSX_PRO_NONE) {           // 32-row tiles + a tail of 16-row tiles
    if (g.epi == DOSX_EPI_LN && g.w_layout == 0) return launch_gemm_mixed3<0, 4, 0, DOSX_PRO_NONE, DOSX_EPI_LN, 1>(A, T, s);
    if (g.epi == DOSX_EPI_PRELU_LN_BWD && g.w_layout == 1) return launch_gemm_mixed3<0, 4, 1, DOSX_PRO_NONE, DOSX_EPI_PRELU_LN_BWD, 1>(A, T, s);
  }
  return -1;
}

static int gemm_validate(const DosxGemm& g) {
  DOSX_CHECK_ARG(g.K > 0, "dosx_gemm: K=%d", g.K);
  DOSX_CHECK_ARG(g.nseg >= 1 && g.nseg <= 3, "dosx_gemm: nseg=%d", g.nseg);
  int ksum = 0;
  for (int i = 0; i < g.nseg; ++i) {
    DOSX_CHECK_ARG(g.a[i].p && g.a[i].width > 0 && g.a[i].map.d > 0, "dosx_gemm: bad segment %d", i);
    ksum += g.a[i].width;
  }
  DOSX_CHECK_ARG(ksum == g.K, "dosx_gemm: segment widths sum to %d, K=%d", ksum, g.K);
  DOSX_CHECK_ARG((g.N & 3) == 0 && (g.ldo & 3) == 0 && aligned16(g.out), "dosx_gemm: N/ldo/out must be 4-float aligned");
  DOSX_CHECK_ARG(g.w && (g.out || g.epi == DOSX_EPI_SEGSUM), "dosx_gemm: null w/out");
  const bool full_row = (g.epi == DOSX_EPI_LN || g.epi == DOSX_EPI_PRELU_LN_BWD || g.epi == DOSX_EPI_ROWLN_BWD || g.epi == DOSX_EPI_PRELU_LN_BWD_SEG);
  DOSX_CHECK_ARG(!full_row || g.N <= 512, "dosx_gemm: row-wise epilogue needs N <= 512 (hidden <= 256), got %d", g.N);
  DOSX_CHECK_ARG(!g.stats_out || g.N <= 128 * 4, "dosx_gemm: stats_out needs N <= 512");
  DOSX_CHECK_ARG(!g.norm_out || (g.N <= 128 * 4 && g.norm_rstd && g.epi == DOSX_EPI_BIAS_ACT && aligned16(g.norm_out)),
                 "dosx_gemm: norm_out needs the plain epilogue, N <= 512 and norm_rstd");
  DOSX_CHECK_ARG(g.out_map.d > 0, "dosx_gemm: out_map.d must be > 0");
  if (g.res) DOSX_CHECK_ARG(g.res_map.d > 0 && (g.ldr & 3) == 0 && aligned16(g.res), "dosx_gemm: bad residual");
  DOSX_CHECK_ARG(!g.res_pre || (g.res && g.epi == DOSX_EPI_BIAS_ACT && g.res_col0 == 0), "dosx_gemm: res_pre needs a residual and the plain epilogue");
  DOSX_CHECK_ARG(g.res_col0 >= 0 && (g.res_col0 & 3) == 0 && (g.res_col0 == 0 || (g.res && g.epi == DOSX_EPI_BIAS_ACT)),
                 "dosx_gemm: res_col0=%d needs a residual, the plain epilogue and a multiple of 4", g.res_col0);
  if (g.pro == DOSX_PRO_LN_PRELU || g.pro == DOSX_PRO_ROWLN)
    DOSX_CHECK_ARG(g.pro_gamma && g.pro_beta, "dosx_gemm: prologue needs gamma/beta");
  if (g.pro == DOSX_PRO_ROWLN) DOSX_CHECK_ARG(g.pro_stats, "dosx_gemm: ROWLN prologue needs stats");
  if (g.pro == DOSX_PRO_LN_PRELU || g.pro == DOSX_PRO_ROWLN)
    DOSX_CHECK_ARG(g.K <= GEMM_KMAX, "dosx_gemm: LayerNorm prologue needs K <= %d, got %d", GEMM_KMAX, g.K);
  if (g.pro == DOSX_PRO_PRELU || g.pro == DOSX_PRO_LN_PRELU) DOSX_CHECK_ARG(g.pro_alpha, "dosx_gemm: prologue needs alpha");
  if (g.epi == DOSX_EPI_LN) DOSX_CHECK_ARG(g.aux_out, "dosx_gemm: EPI_LN needs aux_out (rstd)");
  if (g.epi == DOSX_EPI_PRELU_LN_BWD || g.epi == DOSX_EPI_PRELU_LN_BWD_SEG)
    DOSX_CHECK_ARG(g.aux && g.aux_stats && g.epi_gamma && g.epi_beta && g.epi_alpha && (g.ldaux & 3) == 0,
                   "dosx_gemm: PRELU_LN_BWD needs aux/aux_stats/gamma/beta/alpha");
  if (g.epi == DOSX_EPI_ROWLN_BWD)
    DOSX_CHECK_ARG(g.aux && g.aux_stats && g.epi_gamma && (g.ldaux & 3) == 0, "dosx_gemm: ROWLN_BWD needs aux/aux_stats/gamma");
  if (g.epi == DOSX_EPI_RELU_MASK) DOSX_CHECK_ARG(g.aux && (g.ldaux & 3) == 0, "dosx_gemm: RELU_MASK needs aux");
  if (g.epi == DOSX_EPI_PRELU_BWD) DOSX_CHECK_ARG(g.aux && g.epi_alpha && (g.ldaux & 3) == 0, "dosx_gemm: PRELU_BWD needs aux/alpha");
  if (g.epi == DOSX_EPI_PRELU_LN_BWD_SEG)
    DOSX_CHECK_ARG(g.seg_tile && g.seg_ntiles > 0 && g.seg_rowptr && g.seg_agg && g.N <= 256 && g.out && g.w_layout == 1 &&
                       g.pro == DOSX_PRO_NONE && !g.bias && g.seg_part && g.seg_cnt && (!g.partials || g.partial_ld >= 2 * g.N + 1),
                   "dosx_gemm: EPI_PRELU_LN_BWD_SEG needs seg_tile / seg_rowptr / seg_agg / seg_part / seg_cnt, N <= 256, W[K,N], "
                   "no prologue / bias");
  if (g.add_p || g.add_q)
    DOSX_CHECK_ARG(g.epi == DOSX_EPI_LN && g.add_p && g.add_q && g.add_ip && g.add_iq && g.ld_add >= g.N && (g.ld_add & 3) == 0 &&
                       aligned16(g.add_p) && aligned16(g.add_q),
                   "dosx_gemm: add_p / add_q need EPI_LN, both tensors with their row indices, and 4-float aligned rows");
  if (g.w_seg_off != 0) {
    DOSX_CHECK_ARG(g.w_layout == 1 && g.nseg > 1 && (g.w_seg_off & 3) == 0, "dosx_gemm: w_seg_off needs w_layout 1, several segments and a multiple of 4");
    for (int i = 0; i < g.nseg; ++i) DOSX_CHECK_ARG((g.a[i].width & 31) == 0, "dosx_gemm: w_seg_off needs segment widths that are multiples of 32");
  }
  if (g.epi == DOSX_EPI_SEGSUM)
    DOSX_CHECK_ARG(g.seg_tile && g.seg_ntiles > 0 && g.seg_rowptr && g.seg_agg && g.N <= 256 && (!g.out || g.res) &&
                       g.out_map.d >= (1 << 30) && g.out_map.idx == nullptr && g.out_map.c == 1 && g.out_map.off == 0 &&
                       g.seg_part && g.seg_cnt,
                   "dosx_gemm: EPI_SEGSUM needs seg_tile / seg_rowptr / seg_agg / seg_part / seg_cnt, N <= 256, an identity "
                   "out_map and res with out");

  return 0;
}

extern "C" int dosx_gemm(const DosxGemm* gp, dosx_stream_t stream) {
  DOSX_CHECK_ARG(gp != nullptr, "dosx_gemm: null descriptor");
  const DosxGemm& g = *gp;
  if (g.M <= 0 || g.N <= 0) return 0;
  if (const int rc = gemm_validate(g)) return rc;

  hipStream_t s = to_stream(stream);
  if (sliver_ok(g)) {
    hipLaunchKernelGGL(sliver_gemm_kernel, dim3(ceil_div(g.M, SG_T), ceil_div(g.N, SG_T)), dim3(256), 0, s, g);
    DOSX_LAUNCH_CHECK();
    return 0;
  }
  GemmLaunch L;
  const int bn = gemm_plan(g, L);
  auto go = [&](const GemmLaunch& X) {
    if (bn == 128) return dispatch_gemm<1>(X, s);
    if (bn == 256) return dispatch_gemm<2>(X, s);
    return dispatch_gemm<4>(X, s);
  };
  const int mA = ((L.rt == 2 || (L.rt == 1 && bn == 512)) && L.vecA && L.vecW && !g.stats_out && !g.norm_out) ? gemm_tail_split(g.M, g.N, g.epi) : 0;
  // dosx_gemm_partial_rows() knows (M, N, epi) only: a call that reduces partial rows must take the split it reports
  DOSX_CHECK_ARG(g.partials == nullptr || mA == gemm_tail_split(g.M, g.N, g.epi),
                 "dosx_gemm: this descriptor (alignment / stats_out / norm_out) cannot take the tail split dosx_gemm_partial_rows "
                 "reports for M=%d N=%d epilogue %d", g.M, g.N, g.epi);
  if (mA > 0) {
    GemmLaunch A = L;                        // the full rounds: rows [0, mA) as 64-row tiles (32-row: the 512-column epilogues)
    A.g.M = mA;
    GemmLaunch T = L;                        // the tail: rows [mA, M) as 32- / 48- (16-)row tiles, same grid
    T.m_base = mA;
    T.part_base = mA / gemm_head_rows(g.epi);
    T.rt = gemm_tail_rt(g.M - mA, g.N, g.epi);
    const int rc = launch_gemm_mixed(bn, A, T, s);
    if (rc != -1) return rc;
    DOSX_CHECK_ARG(g.partials == nullptr, "dosx_gemm: split policy and kernel set disagree (epilogue %d)", g.epi);
  }
  return go(L);
}

template <int RT, int NTW>
static int launch_gemm_pair(const GemmLaunch& A, const GemmLaunch& B, hipStream_t s) {
  constexpr int BN = 128 * NTW, ROWS = RT == 0 ? 16 : (RT == 3 ? 48 : BM * RT);
  const int n1 = ceil_div(A.g.M, ROWS) * ceil_div(A.g.N, BN), n2 = ceil_div(B.g.M, ROWS) * ceil_div(B.g.N, BN);
  constexpr size_t smem = gemm_smem_bytes<RT, NTW, 0, 0>();
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pair_kernel<RT, NTW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_pair_kernel<RT, NTW>), dim3(n1 + n2), dim3(512), smem, s, A, B, n1);
  DOSX_LAUNCH_CHECK();
  return 0;
}

// Two GEMMs in one launch when they share a tile configuration (see gemm_pair_kernel), otherwise one after the other.
extern "C" int dosx_gemm_pair(const DosxGemm* ap, const DosxGemm* bp, dosx_stream_t stream) {
  DOSX_CHECK_ARG(ap != nullptr && bp != nullptr, "dosx_gemm_pair: null descriptor");
  const DosxGemm &a = *ap, &b = *bp;
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("DOSX_GEMM_PAIR");
    on = e ? atoi(e) : 1;
  }
  bool one = on && a.M > 0 && b.M > 0 && a.N == b.N && a.N > 0 && a.w_layout == 0 && b.w_layout == 0 && a.pro == DOSX_PRO_NONE &&
             b.pro == DOSX_PRO_NONE && a.epi == DOSX_EPI_BIAS_ACT && b.epi == DOSX_EPI_BIAS_ACT && !sliver_ok(a) && !sliver_ok(b);
  GemmLaunch A, B;
  int bn = 0;
  if (one) {
    if (const int rc = gemm_validate(a)) return rc;
    if (const int rc = gemm_validate(b)) return rc;
    bn = gemm_plan(a, A);
    one = gemm_plan(b, B) == bn && bn <= 256 && A.vecA && A.vecW && B.vecA && B.vecW;
  }
  if (!one && on && a.M > 0 && b.M > 0 && a.N == b.N && a.N == 128 && a.w_layout == 1 && b.w_layout == 1 && a.pro == DOSX_PRO_NONE &&
      b.pro == DOSX_PRO_NONE && a.epi == DOSX_EPI_PRELU_BWD && b.epi == DOSX_EPI_PRELU_BWD && !sliver_ok(a) && !sliver_ok(b)) {
    // The two encoders' backward (node rows: 16-row tiles; edge rows: 48-row tiles) at the tail of the step: each problem at
    // ITS tile height in one grid (gemm_mixed_kernel with two independent descriptors; partial rows numbered per problem)
    if (const int rc = gemm_validate(a)) return rc;
    if (const int rc = gemm_validate(b)) return rc;
    if (gemm_plan(a, A) == 128 && gemm_plan(b, B) == 128 && A.vecA && A.vecW && B.vecA && B.vecW && A.rt == 0 && B.rt == 3 &&
        gemm_tail_split(a.M, a.N, a.epi) == 0 && gemm_tail_split(b.M, b.N, b.epi) == 0) {
      constexpr size_t sa = gemm_smem_bytes<0, 1, 1, 0>(), sb = gemm_smem_bytes<3, 1, 1, 0>();
      constexpr size_t smem = sa > sb ? sa : sb;
      static bool attr_set = false;
      if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mixed_kernel<0, 3, 1, 1, DOSX_PRO_NONE, 1, DOSX_EPI_PRELU_BWD>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
      }
      const int n1 = ceil_div(a.M, 16), n2 = ceil_div(b.M, 48);
      hipLaunchKernelGGL((gemm_mixed_kernel<0, 3, 1, 1, DOSX_PRO_NONE, 1, DOSX_EPI_PRELU_BWD>), dim3(n1 + n2), dim3(512), smem,
                         to_stream(stream), A, B, n1);
      DOSX_LAUNCH_CHECK();
      return 0;
    }
  }
  if (!one) {
    if (const int rc = dosx_gemm(ap, stream)) return rc;
    return dosx_gemm(bp, stream);
  }
  // the tile height of ONE problem with all the rows: the pair shares the CUs like one grid
  const int rt = gemm_rt(a.M + b.M, a.N, a.epi);
  A.rt = B.rt = rt;
  hipStream_t s = to_stream(stream);
  if (bn == 128) {
    switch (rt) {
      case 0: return launch_gemm_pair<0, 1>(A, B, s);
      case 1: return launch_gemm_pair<1, 1>(A, B, s);
      case 2: return launch_gemm_pair<2, 1>(A, B, s);
      default: return launch_gemm_pair<3, 1>(A, B, s);
    }
  }
  switch (rt) {
    case 0: return launch_gemm_pair<0, 2>(A, B, s);
    case 1: return launch_gemm_pair<1, 2>(A, B, s);
    case 2: return launch_gemm_pair<2, 2>(A, B, s);
    default: return launch_gemm_pair<3, 2>(A, B, s);
  }
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
  int variant;     // grouped launches: 0..3 = fast vectorised path with prologue NONE / PRELU / LN_PRELU / ROWLN, -1 = not groupable
  int nt;          // 64-row sub-tiles of a workgroup's tile (1 or 2)
};

// Wave-specialised like gemm_kernel: waves 0-3 multiply (one 32x32 sub-tile each), waves 4-7 stage the
// next 32-row chunk of dY and A' (two LDS stage buffers, one barrier per chunk, loads two chunks deep).
// FAST = 1: every row map on the path is affine (optionally followed by an index gather on A) and the
// tile lies inside one K-segment, so all lane offsets are fixed and a chunk advances scalar offsets
// only (buffer loads; rows beyond M read as zero through the buffer bounds: no masks, no vector ALU
// beyond the prologue transform).  FAST = 0: generic pointer path (any map, ragged everything).
// Tile of one workgroup: TN = 64 * NT rows of dW (columns of dY) x 64 columns of dW (columns of A).  NT = 2 (round 3):
// every matrix wave multiplies TWO 32x32 sub-tiles that share their A-operand fragment, i.e. 32 MFMAs per 32-row chunk and
// wave behind one barrier and 48 LDS fragment reads instead of 16 behind 32 - a lone workgroup's chunk period was ~1650 clk
// for 1024 clk of MFMAs with the 64x64 tile (tools/stamp_wgrad.py) - and the ~12 k clk of fixed cost per workgroup
// (prologue, publish, ticket) are spent once per 128x64 tile.  NT = 1 for N < 128 and the generic-staging variant.
template <int NT> constexpr int wstg() { return BM * (64 * NT + 4) + BM * LDT; }   // floats of one stage buffer: Ys | Xs
constexpr int WSTG = wstg<1>();
constexpr int WSTG_MAX = wstg<2>();

// ---- finished mode (DosxWgrad.dst != NULL): in-launch reduction over the M-splits by the last arriver of a tile ----
// Protocol (cdna_hip_programming.md, in-launch split-K reduction, write-through form): every workgroup stores its TN x 64
// partial tile to its private slot of the scratch slab with 16-byte sc1 (write-through) stores, every wave drains its
// stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, ONE lane draws a ticket with a relaxed agent-scope
// fetch_add on the tile's counter; the workgroup that draws nsplit-1 reads ALL nsplit partial tiles back with sc1 loads
// (they bypass this CU's L1; every load of handed-off bytes is such a load) and adds them in split order 0, 1, 2, ...:
// a fixed order whoever arrives last, so results are bitwise reproducible.  Scratch layout is tile-major
// ([split][tile][TN][64]): every partial tile is one contiguous, 16-byte aligned block whatever N and K are.
template <int NT>
__device__ __forceinline__ void wgrad_finish(const DosxWgrad& g, float* __restrict__ Sm, const float4 (&bs)[2 * NT],
                                             const bool do_bias, const int z, const int bx, const int by, const int ntk) {
  constexpr int TN = 64 * NT, LDY = TN + 4, WTILE = TN * WT, NH = 2 * NT;      // NH float4 per lane cover the tile
  constexpr int W_FLAG = 3 * wstg<NT>() - 4;             // LDS word that broadcasts the ticket
  const int tid = threadIdx.x;
  const int N = g.N, K = g.K, ns = g.nsplit;
  const int n0 = by * TN, k0 = bx * WT;
  const int ntn = (N + TN - 1) / TN;
  const int tile = by * ntk + bx;
  const int ntiles = ntk * ntn;
  const int nbp = (N + WT - 1) / WT * WT;                 // row stride of the bias scratch
  float* Br = Sm + TN * LDT;                              // staged dY column sums: [32][LDY] behind the tile
  if (do_bias && tid >= 256) {
    const int st = tid - 256, r = st >> 3, c4 = (st & 7) * 4;
#pragma unroll
    for (int j = 0; j < 2 * NT; ++j) st4(&Br[r * LDY + c4 + 32 * j], bs[j]);
  }
  __syncthreads();                                        // tile (matrix waves) and bias rows (staging waves) are in LDS
  WSTAMP(61);
  float bsum = 0.f;
  if (do_bias && tid < TN) {
#pragma unroll 8
    for (int rr = 0; rr < BM; ++rr) bsum += Br[rr * LDY + tid];
  }
  // lane -> NH float4 of the tile: element index e = tid + 512 h (row e >> 4, columns 4 * (e & 15) ..)
  auto write_dst = [&](const float4 (&v)[NH], const float bfin) {
    const size_t ldd = g.ldd > 0 ? (size_t)g.ldd : (size_t)K;
    const bool vec_ok = ((K & 3) == 0) && ((ldd & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.dst) & 15) == 0);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int e = tid + 512 * h, n = n0 + (e >> 4), k = k0 + (e & 15) * 4;
      if (n >= N || k >= K) continue;
      float* d = g.dst + (size_t)n * ldd + k;
      if (vec_ok) {                                       // (K % 4 == 0: k < K means k + 3 < K)
        st4(d, g.accumulate ? f4add(ld4(d), v[h]) : v[h]);
      } else {
        const float tv[4] = {v[h].x, v[h].y, v[h].z, v[h].w};
        for (int c = 0; c < 4 && k + c < K; ++c) d[c] = tv[c] + (g.accumulate ? d[c] : 0.f);
      }
    }
    if (do_bias && tid < TN && n0 + tid < N) g.dst_bias[n0 + tid] = bfin + (g.accumulate ? g.dst_bias[n0 + tid] : 0.f);
  };
  float4 v[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const int e = tid + 512 * h;
    v[h] = ld4(&Sm[(e >> 4) * LDT + (e & 15) * 4]);
  }
  if (ns == 1) {                                          // no split: this workgroup's tile IS the result
    write_dst(v, bsum);
    return;
  }
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)g.slab, 0, 0x7fffffff, 0x00020000);
  const uint32_t zstride = (uint32_t)ntiles * WTILE * 4;  // bytes between the slots of consecutive splits (host checks the total < 2^31)
  const uint32_t voff = ((uint32_t)tile * WTILE + (uint32_t)tid * 4) * 4;
#pragma unroll
  for (int h = 0; h < NH; ++h)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i32, v[h]), rS, voff + h * 512 * 16, (uint32_t)z * zstride, 16);   // aux 16 = sc1
  if (do_bias && tid < TN && n0 + tid < N)
    __hip_atomic_store(g.slab_bias + (size_t)z * nbp + n0 + tid, bsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // (sc1 store)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // EVERY storing wave drains its write-through stores
  __syncthreads();
  WSTAMP(62);
  int* flag = reinterpret_cast<int*>(Sm + W_FLAG);
  if (tid == 0) *flag = dosx_ticket(g.counters + tile);
  __syncthreads();
  WSTAMP(63);
  if (*flag != ns - 1) return;                            // not the last arriver of this tile
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // (no instruction: keeps the loads below the ticket)
  // ---- last arriver: sum the nsplit partial tiles in split order (sc1 loads, all issued before the first add) ----
  constexpr int ZB = 16 / NT;                             // loads in flight per lane and float4 slot
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    float4 s = f4zero();
    for (int zb = 0; zb < ns; zb += ZB) {
      float4 t[ZB];
#pragma unroll
      for (int q = 0; q < ZB; ++q) {
        const int zz = min(zb + q, ns - 1);               // (clamped duplicates are loaded, never added)
        t[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rS, voff + h * 512 * 16, (uint32_t)zz * zstride, 16));
      }
#pragma unroll
      for (int q = 0; q < ZB; ++q)
        if (zb + q < ns) s = (zb + q == 0) ? t[q] : f4add(s, t[q]);
    }
    v[h] = s;
  }
  float bfin = 0.f;
  if (do_bias && tid < TN && n0 + tid < N) {
    for (int zz = 0; zz < ns; ++zz) {
      const float b = __hip_atomic_load(g.slab_bias + (size_t)zz * nbp + n0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bfin = zz == 0 ? b : bfin + b;
    }
  }
  write_dst(v, bfin);
  if (tid == 0) __hip_atomic_store(g.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
}

template <int PRO, int VEC, int FAST, int NT = 1>
__device__ __forceinline__ void wgrad_body(const WgradLaunch& L, const int bid, float* __restrict__ Sm) {
  const DosxWgrad& g = L.g;
  constexpr int TN = 64 * NT, LDY = TN + 4, NY = 2 * NT;   // NY float4 of dY per staging lane and chunk
  constexpr int STG = wstg<NT>();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // 1-D grid, the M-split index varies fastest: workgroups are dealt to the 8 XCDs round-robin by linear id,
  // so (with 8 | nsplit) all the (n, k) tiles of one M-split land on the SAME XCD and share its L2: every
  // dY / A row of the split crosses the fabric once instead of once per tile that uses it
  // (split counts and K / 64 are powers of two at every shape of the BASELINE models: shifts instead of five run-time integer
  //  divisions of ~40 instructions each in the fixed part of every workgroup)
  const int ntk = (g.K + WT - 1) / WT;
  const UDiv dns(g.nsplit), dntk(ntk);
  const int tile = dns.div(bid), z = bid - tile * g.nsplit;
  const int by = dntk.div(tile), bx = tile - by * ntk;
  const int k0 = bx * WT, n0 = by * TN;
  const int M = g.M, N = g.N, K = g.K;
  const int chunk = (dns.div(M + g.nsplit - 1) + BM - 1) / BM * BM;
  const int ms = z * chunk, me = min(M, ms + chunk);
  const int nch = ms < me ? (me - ms + BM - 1) / BM : 0;
  const bool do_bias = ((g.dst != nullptr ? g.dst_bias : g.slab_bias) != nullptr) && (bx == 0);

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  float4 bs[NY];                                          // staging lanes: column sums of their dY values
#pragma unroll
  for (int j = 0; j < NY; ++j) bs[j] = f4zero();
  WSTAMP(0);

  if (wave_u >= 4) {
    // =============================== staging waves ===============================================
    const int st = tid - 256;
    const int r = st >> 3, c4 = (st & 7) * 4;             // row r of the chunk, columns c4 + 32 j
    float4 gq0 = f4zero(), gq1 = f4zero(), bq0 = f4zero(), bq1 = f4zero();
    if (VEC && (PRO == DOSX_PRO_LN_PRELU || PRO == DOSX_PRO_ROWLN)) {
      const int ka = (k0 + c4) < K ? (k0 + c4) : 0, kb = (k0 + c4 + 32) < K ? (k0 + c4 + 32) : 0;
      gq0 = ld4(g.pro_gamma + ka); bq0 = ld4(g.pro_beta + ka);
      gq1 = ld4(g.pro_gamma + kb); bq1 = ld4(g.pro_beta + kb);
    }
    struct Set {
      float4 y[NY];
      ARaw x0, x1;
      AState st;        // generic path: row pointers / statistics of this lane's row
      bool row_ok;
      int idx;          // fast path: gathered row index of this lane's row, fetched two chunks ahead
      float mean, rstd;
    };
    // THREE stage buffers, the staging waves TWO chunks ahead of the matrix waves in LDS (chunk c+2 is stored while chunk c
    // is multiplied): at the barrier that ends chunk c the data of chunk c+1 has been visible for a whole chunk, so the
    // matrix waves fetch its fragments UNDER the MFMAs of chunk c.  Chunk c travels in register set c % NSET and LDS
    // buffer c % 3.  Every `issue` is UNCONDITIONAL (chunks past the end of the split read valid / bounds-zeroed rows that
    // are never stored): a conditionally issued batch of loads makes the number of loads in flight path-dependent, and
    // hipcc then protects every later use with s_waitcnt vmcnt(0) - i.e. each store waits for the loads issued LATER as
    // well and the pipeline runs one deep.
    constexpr int NSET = 2;       // (4 sets, 115 VGPRs: measured no faster - the chunk period of a lone workgroup is LDS
                                  //  fragment reads + barrier, not loads)
    Set sets[NSET];
    auto pipeline = [&](auto&& issue, auto&& store) {
      WSTAMP_S(0);
#pragma unroll
      for (int i = 0; i < NSET; ++i) issue(sets[i], ms + i * BM);
      if (nch > 0) store(Sm, sets[0]);
      issue(sets[0], ms + NSET * BM);
      if (nch > 1) store(Sm + STG, sets[1]);
      issue(sets[1], ms + (NSET + 1) * BM);
      WSTAMP_S(1);
      __syncthreads();                                    // chunks 0 and 1 are visible
      int b2 = 2;                                         // buffer of chunk c + 2
      // (the staging waves always run an EVEN number of chunk periods - the matrix waves add a barrier when nch is odd:
      //  a `break` between the two halves made the number of loads in flight path-dependent, hipcc then waited for the
      //  gathered-row index with vmcnt(2) instead of vmcnt(7), i.e. for every load of the other register set too, and the
      //  pipeline ran one chunk deep)
      for (int c = 0; c < nch; c += NSET) {
#pragma unroll
        for (int u = 0; u < NSET; ++u) {                  // iteration c + u: chunk c+u+2 -> LDS, chunk c+u+2+NSET -> its set
          Set& q = sets[(u + 2) % NSET];
          WSTAMP_S(2 + 3 * (c + u));
          if (c + u + 2 < nch) store(Sm + b2 * STG, q);   // buffer (c+u+2) % 3 was last read during iteration c+u-1
          WSTAMP_S(3 + 3 * (c + u));
          issue(q, ms + (c + u + 2 + NSET) * BM);
          b2 = b2 == 2 ? 0 : b2 + 1;
          WSTAMP_S(4 + 3 * (c + u));
          __syncthreads();
        }
      }
    };
    if constexpr (FAST) {
      // the K-segment this tile lives in (host guarantees it does not straddle two)
      int sgi = 0, kbase = 0;
      if (g.nseg > 1 && k0 >= g.a[0].width) { sgi = 1; kbase = g.a[0].width; }
      if (g.nseg > 2 && k0 >= g.a[0].width + g.a[1].width) { sgi = 2; kbase = g.a[0].width + g.a[1].width; }
      const DosxSeg sa = g.a[sgi];
      const int kw = sa.width - (k0 - kbase);                         // columns of this tile inside K
      const int cy = g.dy.map.c, oy = g.dy.map.off, ca = sa.map.c, oa = sa.map.off;
      const bool gather = sa.map.idx != nullptr;
      // Row maps on this path are affine (row = r*c + off) or BLOCKED div/mod maps (row = (r/d)*m + (r%d)*c + off with
      // 32 | d, checked by the host): a chunk starts at a multiple of 32 rows, so the 32 rows of a chunk never wrap and
      // the map splits into a per-lane part (r_local*c + off) and a per-chunk SCALAR part chunk_row0(R) - the heads'
      // weight gradients (dY rows of one prediction branch, energies / graph rows broadcast over the other axis).
      const bool by_blk = g.dy.map.d < (1 << 30), ba_blk = sa.map.d < (1 << 30);
      auto row0 = [](const DosxRowMap& mp, bool blk, int R) { return blk ? (R / mp.d) * mp.m + (R % mp.d) * mp.c : R * mp.c; };
      auto maxrow = [](const DosxRowMap& mp, bool blk, int Mr) {        // largest row a valid r < Mr maps to
        if (!blk) return (Mr - 1) * mp.c + mp.off;
        if (mp.m == 0) return (min(Mr, mp.d) - 1) * mp.c + mp.off;       // pure mod map: rows repeat
        return ((Mr - 1) / mp.d) * mp.m + ((Mr - 1) % mp.d) * mp.c + mp.off;
      };
      // bounds: rows >= M read as zero (dY) -> they add nothing to the sums; no masks anywhere
      const uint32_t ybytes = (uint32_t)(((size_t)maxrow(g.dy.map, by_blk, M) + 1) * (size_t)g.dy.ld * 4);
      const uint32_t abytes = gather ? 0x7fffffffu : (uint32_t)(((size_t)maxrow(sa.map, ba_blk, M) + 1) * (size_t)sa.ld * 4);
      const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)g.dy.p, 0, ybytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)sa.p, 0, abytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(gather ? sa.map.idx : (const int*)g.dy.p), 0, (uint32_t)(((size_t)maxrow(sa.map, ba_blk, M) + 1) * 4), 0x00020000);
      const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(PRO == DOSX_PRO_ROWLN ? g.pro_stats : g.dy.p), 0, (uint32_t)((size_t)M * 8), 0x00020000);
      const int kx0 = c4 < kw ? c4 : 0, kx1 = (c4 + 32) < kw ? (c4 + 32) : 0;
      // per-lane parts (row r of the chunk), the chunk's first row comes in as a scalar offset
      uint32_t vY[NY];
#pragma unroll
      for (int j = 0; j < NY; ++j) {
        const int ny = (n0 + c4 + 32 * j) < N ? (n0 + c4 + 32 * j) : 0;
        vY[j] = (uint32_t)((((size_t)r * cy + oy) * g.dy.ld + ny) * 4);
      }
      const uint32_t colA0 = (uint32_t)((k0 - kbase + kx0) * 4), colA1 = (uint32_t)((k0 - kbase + kx1) * 4);
      const uint32_t vA0 = (uint32_t)(((size_t)r * ca + oa) * sa.ld * 4) + colA0;   // (!gather)
      const uint32_t vA1 = (uint32_t)(((size_t)r * ca + oa) * sa.ld * 4) + colA1;
      const uint32_t vI = (uint32_t)(((size_t)r * ca + oa) * 4);
      const uint32_t vS = (uint32_t)((ms + r) * 8);
      const int ldy4 = g.dy.ld * 4, lda4 = sa.ld * 4;
      const float alpha = (PRO == DOSX_PRO_PRELU || PRO == DOSX_PRO_LN_PRELU) ? *g.pro_alpha : 0.f;
      // (every load below is unconditional: a load inside a run-time branch makes hipcc place s_waitcnt vmcnt(0)
      //  at the join, i.e. right behind the issue - measured 900-1850 clk per chunk in the staging waves)
      auto load_idx = [&](Set& q, int m) {
        const int mu = __builtin_amdgcn_readfirstlane(m);
        q.idx = __builtin_amdgcn_raw_buffer_load_b32(rI, vI, row0(sa.map, ba_blk, mu) * 4, 0);      // (!gather: dummy, unused)
      };
#pragma unroll
      for (int i = 0; i < NSET; ++i) load_idx(sets[i], ms + i * BM);
      auto issue = [&](Set& q, int m) {
        const int mu = __builtin_amdgcn_readfirstlane(m);
        const int cidx = __builtin_amdgcn_readfirstlane((m - ms) / BM);
        const int soY = __builtin_amdgcn_readfirstlane(row0(g.dy.map, by_blk, mu) * ldy4);
#pragma unroll
        for (int j = 0; j < NY; ++j)
          q.y[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rY, vY[j], soY, 0));
        const uint32_t rowb = (uint32_t)q.idx * (uint32_t)lda4;
        const uint32_t o0 = gather ? rowb + colA0 : vA0, o1 = gather ? rowb + colA1 : vA1;
        const int soA = __builtin_amdgcn_readfirstlane(gather ? 0 : row0(sa.map, ba_blk, mu) * lda4);
        q.x0.v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rA, o0, soA, 0));
        q.x1.v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rA, o1, soA, 0));
        load_idx(q, m + NSET * BM);                 // rows beyond M read index 0 through the buffer bounds
        if (PRO == DOSX_PRO_ROWLN) {
          q.mean = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rS, vS, cidx * (BM * 8), 0));
          q.rstd = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rS, vS + 4, cidx * (BM * 8), 0));
        }
      };
      auto store = [&](float* buf, Set& q) {
        AState t;
        t.ok = true; t.mean = q.mean; t.rstd = q.rstd; t.alpha = alpha;
#pragma unroll
        for (int j = 0; j < NY; ++j) st4(&buf[r * LDY + c4 + 32 * j], q.y[j]);
        st4(&buf[BM * LDY + r * LDT + c4], a_finish<PRO, VEC, 0>(t, q.x0, 0, K, gq0, bq0));
        st4(&buf[BM * LDY + r * LDT + c4 + 32], a_finish<PRO, VEC, 0>(t, q.x1, 0, K, gq1, bq1));
        if (do_bias) {
#pragma unroll
          for (int j = 0; j < NY; ++j) bs[j] = f4add(bs[j], q.y[j]);
        }
      };
#pragma unroll
      for (int i = 0; i < NSET; ++i) sets[i].mean = sets[i].rstd = 0.f;
      pipeline(issue, store);
    } else {
      auto issue = [&](Set& q, int m) {
        q.row_ok = (m + r) < me;
        const int gm = max(min(m + r, M - 1), 0);   // always a valid row (chunks past the split are loaded, never stored)
        const float* yp = g.dy.p + (size_t)dosx_map_row(g.dy.map, gm) * g.dy.ld;
#pragma unroll
        for (int j = 0; j < NY; ++j) {
          const int n = n0 + c4 + 32 * j;
          float4 v = f4zero();
          if (VEC) {
            v = ld4(yp + (n < N ? n : 0));
          } else if (q.row_ok && n < N) {
            v.x = yp[n];
            if (n + 1 < N) v.y = yp[n + 1];
            if (n + 2 < N) v.z = yp[n + 2];
            if (n + 3 < N) v.w = yp[n + 3];
          }
          q.y[j] = v;
        }
        a_state_init(q.st, g.a, g.nseg, PRO, g.pro_stats, g.pro_alpha, gm, q.row_ok);
        q.x0 = a_issue<PRO, VEC>(q.st, k0 + c4, K, g.pro_gamma, g.pro_beta);
        q.x1 = a_issue<PRO, VEC>(q.st, k0 + c4 + 32, K, g.pro_gamma, g.pro_beta);
      };
      auto store = [&](float* buf, Set& q) {
#pragma unroll
        for (int j = 0; j < NY; ++j) {
          float4 a0 = q.y[j];
          if (VEC && !(q.row_ok && (n0 + c4 + 32 * j) < N)) a0 = f4zero();
          st4(&buf[r * LDY + c4 + 32 * j], a0);
          if (do_bias) bs[j] = f4add(bs[j], a0);
        }
        st4(&buf[BM * LDY + r * LDT + c4], a_finish<PRO, VEC, 1>(q.st, q.x0, k0 + c4, K, gq0, bq0));
        st4(&buf[BM * LDY + r * LDT + c4 + 32], a_finish<PRO, VEC, 1>(q.st, q.x1, k0 + c4 + 32, K, gq1, bq1));
      };
      pipeline(issue, store);
    }
  } else {
    // =============================== matrix waves ================================================
    // wave -> (wn, wk): NT 32-row sub-tiles of dW rows [(wn*NT + t)*32, +32) x the 32 dW columns [wk*32, +32)
    const int wn = wave >> 1, wk = wave & 1;
    // NT = 1: a second accumulator (even / odd row pairs) - MFMAs chained through ONE accumulator issue every ~87 clk
    // instead of every 64 (dependent-issue latency of the 16-pass instruction).  NT = 2: the two sub-tiles ARE two chains.
    constexpr int NA2 = NT == 1 ? 1 : 0;
    f32x16 acc2[1];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[0][i] = 0.f;
    __syncthreads();
    WSTAMP(1);
    // Fragments travel in HALF chunks (16 rows = 8 MFMA steps per sub-tile): the reads of the next half are issued right
    // before the MFMAs of the current one, so every LDS round trip hides under matrix work - across the barrier too: the
    // first half of chunk c+1 is fetched under the second half of chunk c (its buffer has been complete since the previous
    // barrier).
    struct Frag { float a[NT][BM / 4], b[BM / 4]; };
    auto fetch = [&](Frag& f, int buf, int half) {
      const float* Ys = Sm + buf * STG;
      const float* Xs = Ys + BM * LDY;
#pragma unroll
      for (int i = 0; i < BM / 4; ++i) {
#pragma unroll
        for (int t = 0; t < NT; ++t) f.a[t][i] = Ys[(BM / 2 * half + 2 * i + hh) * LDY + (wn * NT + t) * 32 + l31];
        f.b[i] = Xs[(BM / 2 * half + 2 * i + hh) * LDT + wk * 32 + l31];
      }
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
      for (int i = 0; i < BM / 4; i += 2) {
        if constexpr (NA2) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[0][i], f.b[i], acc[0], 0, 0, 0);
          acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[0][i + 1], f.b[i + 1], acc2[0], 0, 0, 0);
        } else {
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t)
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[t][i + u], f.b[i + u], acc[t], 0, 0, 0);
        }
      }
    };
    Frag f0, f1;
    int cur = 0;                                            // buffer of chunk c
    if (nch > 0) fetch(f0, 0, 0);
    for (int c = 0; c < nch; ++c) {
      WSTAMP(2 + 2 * c);
      const int nxt = cur == 2 ? 0 : cur + 1;
      fetch(f1, cur, 1);
      mma(f0);
      if (c + 1 < nch) fetch(f0, nxt, 0);
      mma(f1);
      cur = nxt;
      WSTAMP(3 + 2 * c);
      __syncthreads();
    }
    if (nch & 1) __syncthreads();                           // (the staging waves' loop runs an even number of periods)
    WSTAMP(60);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if constexpr (NA2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] += acc2[0][i];
      }
      if (g.dst != nullptr) {                                // finished mode: the tile goes to LDS first (below)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          Sm[((wn * NT + t) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh) * LDT + wk * 32 + l31] = acc[t][i];
      } else {
        float* slab = g.slab + (size_t)z * N * K;
        const int kcol = k0 + wk * 32 + l31;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int n = n0 + (wn * NT + t) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          if (n < N && kcol < K) slab[(size_t)n * K + kcol] = acc[t][i];
        }
      }
    }
  }
  if (g.dst != nullptr) {
    wgrad_finish<NT>(g, Sm, bs, do_bias, z, bx, by, ntk);
    return;
  }
  if (do_bias) {          // workgroup-uniform: column sums of the staged dY rows -> slab_bias[z][n0 .. n0+TN-1]
    float* Br = Sm;       // [32][LDY]   (the stage buffers are dead: the loop ended with a barrier)
    if (wave_u >= 4) {
      const int st = tid - 256, r = st >> 3, c4 = (st & 7) * 4;
#pragma unroll
      for (int j = 0; j < NY; ++j) st4(&Br[r * LDY + c4 + 32 * j], bs[j]);
    }
    __syncthreads();
    if (tid < TN && n0 + tid < N) {
      float sum = 0.f;
#pragma unroll 8
      for (int rr = 0; rr < BM; ++rr) sum += Br[rr * LDY + tid];
      g.slab_bias[(size_t)z * N + n0 + tid] = sum;
    }
  }
}

template <int PRO, int VEC, int FAST, int NT>
__global__ __launch_bounds__(512) void wgrad_kernel(const WgradLaunch L) {
  __shared__ __align__(16) float Sm[3 * wstg<NT>()];
  wgrad_body<PRO, VEC, FAST, NT>(L, (int)blockIdx.x, Sm);
}

// Grouped launch: ONE grid over several weight-gradient jobs (the jobs travel as a kernel argument like the slab
// reduction's).  Interleaved one by one with the dgrad chain, each job is a kernel of 256-384 workgroups that shares the
// matrix pipes with whatever the main stream runs; issued together they are one saturating grid with no tails.
constexpr int WG_MAX_JOBS = 8;
constexpr int WG_MAX_RJOBS = 40;
struct WgradGroup {
  WgradLaunch job[WG_MAX_JOBS];
  int first_block[WG_MAX_JOBS + 1];
  int n;
  // row-partial reductions riding in the same grid (blocks >= first_block[n]): dosx_grad_flush
  DosxReduceJob rjob[WG_MAX_RJOBS];
  int rfirst[WG_MAX_RJOBS + 1];
  int nr;
  int block_off;     // this launch covers the blocks [block_off, block_off + gridDim.x) of the group (dosx_grad_flush: rounds)
};
static_assert(sizeof(WgradGroup) <= 4064, "WgradGroup must fit the kernel argument segment");

__device__ __forceinline__ void reduce_body(const DosxReduceJob& j, int slice, int t256, float4 (*red)[64]);

// (two workgroups per CU: 4 waves per SIMD at <= 128 VGPRs, 2 x 77 KB of LDS.  Three per CU - 80 VGPRs - measured no faster
//  with the 64x64 tile, and spilled.)
#ifndef DOSX_WGRAD_OCC
#define DOSX_WGRAD_OCC 4
#endif
__global__ __launch_bounds__(512, DOSX_WGRAD_OCC) void wgrad_grouped_kernel(const WgradGroup G_arg) {
  __shared__ __align__(16) float Sm[3 * WSTG_MAX];
  // the job table is read where it lies, in the kernel-argument segment (scalar loads through a constant-address-space
  // pointer): indexing the by-value parameter with a run-time index made hipcc copy the whole 4 KB table to scratch
  const WgradGroup& G = *(const WgradGroup*)__builtin_amdgcn_kernarg_segment_ptr();
  const int bx = (int)blockIdx.x + G.block_off;
  if (bx >= G.first_block[G.n]) {
    // ---- a reduction block: 2 slices of 256 elements (one per half of the workgroup) ----
    const int rb = bx - G.first_block[G.n];
    int lo = 0, hi = G.nr - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (G.rfirst[mid] <= rb) lo = mid; else hi = mid - 1;
    }
    const int half = (int)threadIdx.x >> 8;
    reduce_body(G.rjob[lo], (rb - G.rfirst[lo]) * 2 + half, (int)threadIdx.x & 255,
                reinterpret_cast<float4(*)[64]>(Sm + half * 4 * 64 * 4));
    return;
  }
  int lo = 0, hi = G.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (G.first_block[mid] <= bx) lo = mid; else hi = mid - 1;
  }
  const WgradLaunch& L = G.job[lo];
  const int bid = bx - G.first_block[lo];
  switch (L.variant + 8 * (L.nt - 1)) {             // workgroup-uniform
    case 0: wgrad_body<DOSX_PRO_NONE, 1, 1, 1>(L, bid, Sm); break;
    case 1: wgrad_body<DOSX_PRO_PRELU, 1, 1, 1>(L, bid, Sm); break;
    case 2: wgrad_body<DOSX_PRO_LN_PRELU, 1, 1, 1>(L, bid, Sm); break;
    case 3: wgrad_body<DOSX_PRO_ROWLN, 1, 1, 1>(L, bid, Sm); break;
    case 4: wgrad_body<DOSX_PRO_NONE, 0, 0, 1>(L, bid, Sm); break;      // unaligned operands (K = 118 atom features)
    case 8: wgrad_body<DOSX_PRO_NONE, 1, 1, 2>(L, bid, Sm); break;      // 128 x 64 tiles
    case 9: wgrad_body<DOSX_PRO_PRELU, 1, 1, 2>(L, bid, Sm); break;
    case 10: wgrad_body<DOSX_PRO_LN_PRELU, 1, 1, 2>(L, bid, Sm); break;
    case 11: wgrad_body<DOSX_PRO_ROWLN, 1, 1, 2>(L, bid, Sm); break;
    default: break;
  }
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

// One 256-element slice of one job, by 256 threads (t256 = thread within the slice's group; wave sl of the group sums
// the partial rows sl, sl+4, sl+8, ...; `red`: [4][64] float4 of LDS).  NOTE: contains a barrier - both halves of a
// 512-thread block (wgrad_grouped_kernel) must call it.
__device__ __forceinline__ void reduce_body(const DosxReduceJob& j, int slice, int t256, float4 (*red)[64]) {
  const int q = t256 & 63, sl = t256 >> 6;
  const int i = slice * RP_SLICE + q * 4;
  const bool vec = ((j.stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(j.src) & 15) == 0) && ((j.count & 3) == 0);
  float4 s0 = f4zero(), s1 = f4zero();
  if (i < j.count) {
    if (vec) {
      const float* p = j.src + i;
      int k = sl;
      for (; k + 12 < j.nsplit; k += 16) {     // four loads in flight per lane (16 slabs: one batch per wave)
        const float4 v0 = ld4(p + (size_t)k * j.stride), v1 = ld4(p + (size_t)(k + 4) * j.stride);
        const float4 v2 = ld4(p + (size_t)(k + 8) * j.stride), v3 = ld4(p + (size_t)(k + 12) * j.stride);
        s0 = f4add(s0, f4add(v0, v2));
        s1 = f4add(s1, f4add(v1, v3));
      }
      for (; k + 4 < j.nsplit; k += 8) {
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

__global__ __launch_bounds__(256) void reduce_partials_kernel(const ReduceLaunch L) {
  __shared__ float4 red[4][64];
  // binary search: largest j with first_block[j] <= blockIdx.x
  int lo = 0, hi = L.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (L.first_block[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  reduce_body(L.job[lo], (int)blockIdx.x - L.first_block[lo], (int)threadIdx.x, red);
}

}  // namespace

namespace {
// rows of dW per workgroup tile / 64: 128 x 64 tiles for the long jobs (M >= 16384 rows, N >= 128).  Measured (round 3):
// eDOS H = 256 (M = 17880 / 25728) 8.29-8.38 -> 8.16-8.20 ms per step, the grouped launches 418 -> 398 us; the Phonon-DOS
// step (M <= 9344: 35 chunks per workgroup at 8 splits) is 0.5 % SLOWER with them - half as many workgroups of twice the
// length leave more of the 512 slots idle at the end of a group than the 25 % fewer L2 bytes per flop give back.
int wgrad_nt(int M, int N) {
  static int force = -1;
  if (force < 0) {
    const char* e = getenv("DOSX_WGRAD_NT");       // 1 / 2: every job that can (tests, tools/exp/ab_nt*.sh)
    force = e ? atoi(e) : 0;
  }
  if (N < 2 * WT || force == 1) return 1;
  return (force == 2 || M >= 16384) ? 2 : 1;
}
}  // namespace

extern "C" int dosx_wgrad_splits(int M, int N, int K) {
  if (M <= 0) return 1;
  const int tiles = ceil_div(N, WT * wgrad_nt(M, N)) * ceil_div(K, WT);
  int s = ceil_div(512, tiles);
  const int cap = ceil_div(M, 128);
  if (s > cap) s = cap;
  static int max_split = 0;
  if (max_split == 0) {
    const char* e = getenv("DOSX_WGRAD_MAXSPLIT");
    // 8 (round 3, finished mode): half the partial tiles the last arriver has to read back, workgroups twice as long
    // against ~12k clk of fixed prologue + publish cost each: GNN-layer-pair group alone 76 -> 69 us, encoder-stack group
    // 54 -> 47 us, step 1.277 -> 1.271 ms (16 was the optimum of round 2's slab + reduce_partials scheme)
    max_split = e ? atoi(e) : 8;
    if (max_split < 1) max_split = 8;
  }
  // The cap scales with M (round 4): 8 is the optimum where it was tuned - the BASELINE shapes, M <= 25728 rows, whose jobs
  // run in groups that fill the chip together - but a LONG job (M > 32768) with few tiles is a grid of tiles x 8 <= 128-192
  // workgroups on 256 CUs whatever its length (M = 262144, N = 512, K = 128: 17.5 % of the MFMA peak).  Every workgroup keeps
  // at least 4096 rows (128 chunks of MFMAs against its ~12 k clk of fixed cost), up to the 64 partial tiles a counter /
  // scratch slot set is laid out for; the last arriver of a tile then sums up to 64 partials, 8 loads in flight at a time.
  int cap_m = M / 4096;
  if (cap_m > 64) cap_m = 64;
  if (s > (max_split > cap_m ? max_split : cap_m)) s = max_split > cap_m ? max_split : cap_m;
  // ... but no workgroup lives longer than max_chunks 32-row chunks: a weight-gradient workgroup cannot be pre-empted, and a
  // kernel of the dgrad chain that arrives while long-lived ones hold the CUs waits for them (eDOS, M = 25728 rows at 8
  // splits: 100 chunks = ~70 us per workgroup; the two 15-us head dgrad GEMMs behind the self encoder took 228 us each)
  static int max_chunks = -1;
  if (max_chunks < 0) {
    const char* e = getenv("DOSX_WGRAD_MAXCHUNKS");
    max_chunks = e ? atoi(e) : 0;
  }
  if (max_chunks > 0) {
    const int need = ceil_div(M, BM * max_chunks);
    if (s < need) s = need < 64 ? need : 64;
  }
  if (s < 1) s = 1;
  return s;
}

extern "C" int dosx_wgrad_tiles(int N, int K) { return ceil_div(N, WT) * ceil_div(K, WT); }

extern "C" int64_t dosx_wgrad_scratch_floats(int N, int K, int nsplit) {
  // (M is not known here: room for either tile height - N padded to 128 rows covers both)
  const int nt = N >= 2 * WT ? 2 : 1;
  return nsplit <= 1 ? 0 : (int64_t)nsplit * ceil_div(N, WT * nt) * ceil_div(K, WT) * (WT * nt * WT);
}

namespace {

// validation + launch parameters of one job; `fast` / `vec` select the kernel variant
int wgrad_prepare(const DosxWgrad& g, WgradLaunch& L, int& vec, bool& fast, int& blocks) {
  DOSX_CHECK_ARG(g.N > 0 && g.K > 0 && g.nsplit >= 1, "dosx_wgrad: bad dims N=%d K=%d nsplit=%d", g.N, g.K, g.nsplit);
  DOSX_CHECK_ARG(g.nseg >= 1 && g.nseg <= 3 && (g.slab || (g.dst && g.nsplit == 1)) && g.dy.p, "dosx_wgrad: bad operands");
  if (g.dst != nullptr) {             // finished mode
    DOSX_CHECK_ARG(g.counters != nullptr || g.nsplit == 1, "dosx_wgrad: finished mode needs tile counters");
    DOSX_CHECK_ARG(!g.dst_bias || g.slab_bias || g.nsplit == 1, "dosx_wgrad: dst_bias needs the slab_bias scratch");
    DOSX_CHECK_ARG(!g.slab_bias || g.dst_bias, "dosx_wgrad: finished mode with slab_bias needs dst_bias");
    DOSX_CHECK_ARG(g.ldd == 0 || g.ldd >= g.K, "dosx_wgrad: ldd=%d < K=%d", g.ldd, g.K);
    DOSX_CHECK_ARG(dosx_wgrad_scratch_floats(g.N, g.K, g.nsplit) * 4 < 0x7fffffffll, "dosx_wgrad: scratch slab beyond 2 GiB");
    DOSX_CHECK_ARG(g.nsplit <= 64, "dosx_wgrad: nsplit=%d", g.nsplit);
  }
  int ksum = 0;
  for (int i = 0; i < g.nseg; ++i) {
    DOSX_CHECK_ARG(g.a[i].p && g.a[i].width > 0 && g.a[i].map.d > 0, "dosx_wgrad: bad segment %d", i);
    ksum += g.a[i].width;
  }
  DOSX_CHECK_ARG(ksum == g.K, "dosx_wgrad: segment widths sum to %d, K=%d", ksum, g.K);
  DOSX_CHECK_ARG(g.dy.map.d > 0, "dosx_wgrad: dy.map.d must be > 0");
  L.g = g;
  L.vecA = seg_vec_ok(g.a, g.nseg) && (g.K & 3) == 0;
  if (g.pro == DOSX_PRO_LN_PRELU || g.pro == DOSX_PRO_ROWLN)
    L.vecA = L.vecA && aligned16(g.pro_gamma) && aligned16(g.pro_beta);
  L.vecY = ((g.dy.ld & 3) == 0) && aligned16(g.dy.p) && (g.N & 3) == 0;
  vec = L.vecA && L.vecY;
  // fast (buffer-addressed) staging: affine row maps (+ optional gather on A), K tiles inside one segment,
  // every byte offset below 2^31
  // fast (buffer-addressed) staging: every row map affine, or a BLOCKED div/mod map (32 | d, d | M, blocks in row order or
  // a pure mod map; dY's map must send rows >= M past its last valid row: they read as zero through the buffer bounds);
  // an index gather only behind an affine map; K tiles inside one segment; every byte offset below 2^31
  auto maxrow = [](const DosxRowMap& m, int M) -> long long {
    if (m.d >= (1 << 30)) return (long long)(M - 1) * m.c + m.off;
    if (m.m == 0) return (long long)((M < m.d ? M : m.d) - 1) * m.c + m.off;
    return (long long)((M - 1) / m.d) * m.m + (long long)((M - 1) % m.d) * m.c + m.off;
  };
  auto fastmap = [&](const DosxRowMap& m, bool need_monotone) {
    if (m.c < 0 || m.off < 0 || m.m < 0) return false;
    if (m.d >= (1 << 30)) return m.c >= 1 || !need_monotone;
    if ((m.d % BM) != 0 || (g.M % m.d) != 0 || m.idx != nullptr) return false;
    const bool monotone = (long long)m.m >= (long long)(m.d - 1) * m.c + 1;
    return need_monotone ? (monotone && m.c >= 1) : (monotone || m.m == 0);
  };
  fast = vec && g.M > 0 && fastmap(g.dy.map, true) && g.dy.map.idx == nullptr &&
         (unsigned long long)(maxrow(g.dy.map, g.M) + 1) * (unsigned long long)g.dy.ld * 4 < 0x7fffffffull;
  for (int i = 0; fast && i < g.nseg; ++i) {
    const DosxSeg& sg = g.a[i];
    fast = fastmap(sg.map, false) && (g.nseg == 1 || (sg.width % WT) == 0) && (size_t)sg.ld * 4 < 0x7fffffffull &&
           (unsigned long long)(maxrow(sg.map, g.M) + 1) * (unsigned long long)(sg.map.idx ? 4 : (size_t)sg.ld * 4) < 0x7fffffffull;
  }
  if (!vec) DOSX_CHECK_ARG(g.pro == DOSX_PRO_NONE, "dosx_wgrad: prologue %d needs 4-float aligned operands", g.pro);
  DOSX_CHECK_ARG(g.pro >= DOSX_PRO_NONE && g.pro <= DOSX_PRO_ROWLN, "dosx_wgrad: bad prologue %d", g.pro);
  L.variant = (vec && fast) ? (g.pro == DOSX_PRO_NONE ? 0 : g.pro == DOSX_PRO_PRELU ? 1 : g.pro == DOSX_PRO_LN_PRELU ? 2 : 3)
                            : (!vec ? 4 : -1);       // (the aligned generic-staging variant in the grouped kernel too made
                                                      //  hipcc copy the whole job table to scratch, 3.8 KB per lane: those jobs -
                                                      //  the two heads' div/mod row maps - keep their own launches)
  L.nt = (L.variant >= 0 && L.variant <= 3) ? wgrad_nt(g.M, g.N) : 1;
  blocks = ceil_div(g.K, WT) * ceil_div(g.N, WT * L.nt) * g.nsplit;
  return 0;
}

int wgrad_launch_one(const WgradLaunch& L, int vec, bool fast, int blocks, hipStream_t st) {
  const DosxWgrad& g = L.g;
  dim3 grid(blocks);
  if (!vec) {
    hipLaunchKernelGGL((wgrad_kernel<DOSX_PRO_NONE, 0, 0, 1>), grid, dim3(512), 0, st, L);
  } else if (fast && L.nt == 2) {
    switch (g.pro) {
      case DOSX_PRO_NONE: hipLaunchKernelGGL((wgrad_kernel<DOSX_PRO_NONE, 1, 1, 2>), grid, dim3(512), 0, st, L); break;
      case DOSX_PRO_PRELU: hipLaunchKernelGGL((wgrad_kernel<DOSX_PRO_PRELU, 1, 1, 2>), grid, dim3(512), 0, st, L); break;
      case DOSX_PRO_LN_PRELU: hipLaunchKernelGGL((wgrad_kernel<DOSX_PRO_LN_PRELU, 1, 1, 2>), grid, dim3(512), 0, st, L); break;
      default: hipLaunchKernelGGL((wgrad_kernel<DOSX_PRO_ROWLN, 1, 1, 2>), grid, dim3(512), 0, st, L); break;
    }
  } else if (fast) {
    switch (g.pro) {
      case DOSX_PRO_NONE: hipLaunchKernelGGL((wgrad_kernel<DOSX_PRO_NONE, 1, 1, 1>), grid, dim3(512), 0, st, L); break;
      case DOSX_PRO_PRELU: hipLaunchKernelGGL((wgrad_kernel<DOSX_PRO_PRELU, 1, 1, 1>), grid, dim3(512), 0, st, L); break;
      case DOSX_PRO_LN_PRELU: hipLaunchKernelGGL((wgrad_kernel<DOSX_PRO_LN_PRELU, 1, 1, 1>), grid, dim3(512), 0, st, L); break;
      default: hipLaunchKernelGGL((wgrad_kernel<DOSX_PRO_ROWLN, 1, 1, 1>), grid, dim3(512), 0, st, L); break;
    }
  } else {
    switch (g.pro) {
      case DOSX_PRO_NONE: hipLaunchKernelGGL((wgrad_kernel<DOSX_PRO_NONE, 1, 0, 1>), grid, dim3(512), 0, st, L); break;
      case DOSX_PRO_PRELU: hipLaunchKernelGGL((wgrad_kernel<DOSX_PRO_PRELU, 1, 0, 1>), grid, dim3(512), 0, st, L); break;
      case DOSX_PRO_LN_PRELU: hipLaunchKernelGGL((wgrad_kernel<DOSX_PRO_LN_PRELU, 1, 0, 1>), grid, dim3(512), 0, st, L); break;
      default: hipLaunchKernelGGL((wgrad_kernel<DOSX_PRO_ROWLN, 1, 0, 1>), grid, dim3(512), 0, st, L); break;
    }
  }
  DOSX_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int dosx_wgrad(const DosxWgrad* gp, dosx_stream_t stream) {
  DOSX_CHECK_ARG(gp != nullptr, "dosx_wgrad: null descriptor");
  WgradLaunch L;
  int vec = 0, blocks = 0;
  bool fast = false;
  if (int rc = wgrad_prepare(*gp, L, vec, fast, blocks)) return rc;
  return wgrad_launch_one(L, vec, fast, blocks, to_stream(stream));
}

extern "C" int dosx_grad_flush(const DosxWgrad* jobs, int n_jobs, const DosxReduceJob* rjobs, int n_rjobs,
                               dosx_stream_t stream) {
  if (n_jobs <= 0 && n_rjobs <= 0) return 0;
  DOSX_CHECK_ARG(n_jobs <= 0 || jobs != nullptr, "dosx_grad_flush: null job table");
  DOSX_CHECK_ARG(n_rjobs <= 0 || rjobs != nullptr, "dosx_grad_flush: null reduction table");
  hipStream_t st = to_stream(stream);
#ifdef DOSX_STAMPS
  // diagnostic (stamps) build only, never the shipped library: leave the weight-gradient tiles out of the grid to time the rest
  static int skip_w = -1;
  if (skip_w < 0) skip_w = getenv("DOSX_DEBUG_SKIP_WGRAD") ? 1 : 0;
  if (skip_w) n_jobs = 0;
#endif
  WgradGroup G;
  G.n = 0;
  G.nr = 0;
  int blocks_total = 0, rblocks = 0, rdone = 0;
  auto flush = [&]() -> int {
    // reduction jobs still waiting ride along (up to the table's capacity)
    while (rdone < n_rjobs && G.nr < WG_MAX_RJOBS) {
      const DosxReduceJob& j = rjobs[rdone];
      DOSX_CHECK_ARG(j.src && j.dst && j.count > 0 && j.nsplit > 0, "dosx_grad_flush: bad reduction job %d", rdone);
      G.rjob[G.nr] = j;
      G.rfirst[G.nr] = rblocks;
      rblocks += ceil_div(ceil_div(j.count, RP_SLICE), 2);        // two 256-element slices per 512-thread block
      ++G.nr;
      ++rdone;
    }
    if (G.n == 0 && G.nr == 0) return 0;
    G.first_block[G.n] = blocks_total;
    G.rfirst[G.nr] = rblocks;
    // DOSX_WGRAD_ROUND (round 4, default 0 = one launch per group): the group as successive launches of at most r workgroups
    // ("auto": 512 for groups of up to 1024 workgroups, else 1024).  A kernel of the dgrad chain that arrives while a
    // weight-gradient grid is being dispatched waits until a CU is EMPTY (its workgroups need > 100 KB of LDS / > 128 VGPRs; a
    // freed half-CU slot is refilled at once from the weight-gradient grid's backlog); at a launch boundary the backlog is
    // empty and CUs drain completely.  Measured (tools/exp/ab_round*.sh, profiles/r04_ab_wgrad_round.log, interleaved): "auto"
    // takes 0.3-0.6 % off the Phonon-DOS step, 0.4-0.6 % off the Electron-DOS step, 0.9 % off its T4 / 32-crystal shard -
    // and costs the weight-gradient kernels themselves 23 % (387 vs 314 us of kernel time per cfg2 step: every boundary is a
    // tail of partly idle CUs), i.e. the chain gains slightly more than the groups lose.  Not worth a dominant kernel at
    // 0.26 instead of 0.32 of its roofline: off by default.
    static int round = -2;
    if (round == -2) {
      const char* e = getenv("DOSX_WGRAD_ROUND");
      round = !e ? 0 : (e[0] == 'a' ? -1 : atoi(e));
    }
    const int total = blocks_total + rblocks;
    int per = total;
    if (round > 0) per = round;
    else if (round < 0) per = total <= 1024 ? 512 : 1024;
    for (int off = 0; off < total; off += per) {
      G.block_off = off;
      hipLaunchKernelGGL(wgrad_grouped_kernel, dim3(total - off < per ? total - off : per), dim3(512), 0, st, G);
      DOSX_LAUNCH_CHECK();
    }
    G.n = 0;
    G.nr = 0;
    blocks_total = 0;
    rblocks = 0;
    return 0;
  };
  for (int i = 0; i < n_jobs; ++i) {
    WgradLaunch L;
    int vec = 0, blocks = 0;
    bool fast = false;
    if (int rc = wgrad_prepare(jobs[i], L, vec, fast, blocks)) return rc;
    if (L.variant < 0) {                      // (non-affine operands: its own launch, as dosx_wgrad would)
      if (int rc = wgrad_launch_one(L, vec, fast, blocks, st)) return rc;
      continue;
    }
    G.job[G.n] = L;
    G.first_block[G.n] = blocks_total;
    blocks_total += blocks;
    if (++G.n == WG_MAX_JOBS)
      if (int rc = flush()) return rc;
  }
  if (int rc = flush()) return rc;
  while (rdone < n_rjobs)                     // more reductions than one table holds
    if (int rc = flush()) return rc;
  return 0;
}

extern "C" int dosx_wgrad_grouped(const DosxWgrad* jobs, int n_jobs, dosx_stream_t stream) {
  return dosx_grad_flush(jobs, n_jobs, nullptr, 0, stream);
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
