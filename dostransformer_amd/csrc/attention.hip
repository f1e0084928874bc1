// Pre-norm energy-vs-atom attention block of the reference encoder layer (gfx950, fp32 MFMA).
//
//   x1 = x + softmax_fp32( LN0(x) K^T / sqrt(H) ) K ,   K = V = kvhat * gamma0 + beta0
//
// (layers/transformer.py:131-138 + layers/multihead_attention.py:68-74: no q/k/v/out projection,
//  no mask, no head split; zero-padded atoms are real keys with value beta0 — SURVEY.md §0.2-0.3.)
// (More than 320 keys: attention_general.hip, same contract.)
// One workgroup owns a 32-query tile of one crystal; the key set of the crystal (Nk <= 320: atoms <= Nmax,
// or the 51/201 energy bins for self attention) is streamed twice through LDS in 32-wide chunks by four
// staging waves while four matrix waves run the row phases and the MFMAs; the whole score row of a
// query lives in LDS, so the softmax is exact (no online rescaling).
#include <stdlib.h>

#include "common.h"

// Diagnostic build only (-DDOSX_STAMPS): wave 0 of workgroup (0,0) records s_memtime at phase boundaries.
#ifdef DOSX_STAMPS
extern "C" { __device__ unsigned long long dosx_attn_stamp_buf[64]; }
#define ASTAMP(slot)                                                                     \
  do {                                                                                   \
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0)                          \
      dosx_attn_stamp_buf[(slot)] = __builtin_amdgcn_s_memtime();                        \
  } while (0)
#define ASTAMP_S(slot)                                                                   \
  do {                                                                                   \
    if (threadIdx.x == 256 && blockIdx.x == 0 && blockIdx.y == 0 && (slot) < 32)         \
      dosx_attn_stamp_buf[32 + (slot)] = __builtin_amdgcn_s_memtime();                   \
  } while (0)
#else
#define ASTAMP(slot) do { } while (0)
#define ASTAMP_S(slot) do { } while (0)
#endif

namespace {

constexpr int QT = 32;           // query rows per workgroup
constexpr int KC = 32;           // k-chunk (features) for QK^T, key-chunk for PV
constexpr int LDK = KC + 4;      // 36
constexpr int MAX_KT = 3;        // key tiles (32 keys) per wave in QK^T  -> Nk <= 4*3*32 = 384
constexpr int MAX_CT = 2;        // output column tiles per wave in PV    -> H  <= 4*2*32 = 256

struct Geo {
  int HP;    // H rounded up to 32
  int LDH;   // HP + 4
  int NKP;   // Nk rounded up to 32
  int LDS_;  // NKP + 4
};
__host__ __device__ inline Geo make_geo(int H, int Nk) {
  Geo g;
  g.HP = (H + 31) / 32 * 32;
  g.LDH = g.HP + 4;
  g.NKP = (Nk + 31) / 32 * 32;
  g.LDS_ = g.NKP + 4;
  return g;
}

// floats of one streamed-chunk buffer: a K chunk is [NKP][36], a V chunk is [32][LDH]
__host__ __device__ inline int chunk_buf_floats(int NKP, int LDH) { return NKP * LDK > KC * LDH ? NKP * LDK : KC * LDH; }

__device__ __forceinline__ void store_out_tile(const f32x16 (&acc)[MAX_CT], float* Os, int LDH, int HP, int tid) {
  const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int nct = HP / 32;
#pragma unroll
  for (int t = 0; t < MAX_CT; ++t) {
    const int ct = wave + 4 * t;
    if (ct >= nct) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
      Os[row * LDH + ct * 32 + l31] = acc[t][r];
    }
  }
}

// ---- row phases --------------------------------------------------------------------------------
// LayerNorm of the queries, softmax, residual + statistics, dS and the LayerNorm backward all work on
// whole rows.  Layout: a wave owns 8 rows of the 32-row tile and processes them in 2 passes of 4 rows,
// one QUARTER WAVE (16 lanes = one DPP row) per row: lane q of a row owns columns 4q + 64k (k < KCB)
// and keys q + 16 jj (jj < NJ).  A reduction is then 4 DPP instructions for 4 rows at once
// (row16_sum); the first version gave each row a full wave and paid a 64-lane reduction (DPP +
// v_readlane) per row: 16 of them per phase were ~5k clk of the 33k-clk forward kernel (stamps).
constexpr int RP = 2;      // passes (4 rows each) per wave
constexpr int KCB = 4;     // 64-column blocks per row (H <= 256)

__device__ __forceinline__ int row_of(int wave, int p, int lane) { return wave * 8 + p * 4 + (lane >> 4); }

// ================================== streamed keys: wave-specialised kernels ========================
// When the key set of a crystal does not fit the LDS next to the query tile (eDOS: 201 bins x 256 features),
// the keys are STREAMED twice (once for Q.K^T, once for P.V) in 32-wide chunks.  Same recipe as gemm_kernel:
// 8 waves = 4 matrix waves (row phases + MFMA) + 4 staging waves that only move the next chunk
// global -> registers -> LDS (two chunk buffers, one barrier per chunk, loads two chunks deep).
// The staging path carries NO arithmetic: the key affine is folded out of the products,
//     (x̂q γ+β) . (k̂ γ+β)  =  ((x̂q γ+β) ∘ γ) . k̂  + const(q)        (the constant cancels in the softmax)
//     P . (k̂ γ+β)          =  (P . k̂) ∘ γ + β                        (rows of P sum to 1)
//     dO . (k̂ γ+β)^T       =  (dO ∘ γ) . k̂^T + const(q)               (cancels in dS = P ∘ (dP - Σ P dP))
//     dS . (k̂ γ+β)         =  (dS . k̂) ∘ γ                           (rows of dS sum to 0)
// so both operands are the raw normalised keys k̂ (`kvhat`), fetched with buffer loads whose bounds return
// zeros for the rows beyond Nk.  (Exact in real arithmetic; fp32 rounding differs from the resident-key
// kernels at the 1e-7 level, tests/test_gpu_attention.py::test_attention_fwd_bwd covers both.)
constexpr int SKR = 10;    // 32-key row blocks per crystal (Nk <= 320): float4 per staging lane per K chunk
constexpr int SVC = 8;     // 32-column blocks per row (H <= 256):      float4 per staging lane per V chunk

struct StreamGeo {
  int nkc;     // K chunks  = HP / 32 (feature chunks of the Q.K^T-shaped product)
  int nvc;     // V chunks  = NKP / 32 (key chunks of the P.V-shaped product)
  int CHB;     // floats of one chunk buffer
};

// The staging waves' whole job.  `sync()` is the workgroup barrier; the matrix waves execute the mirror-image
// sequence in stream_matrix_*().  Barrier schedule:  [K prologue] B | nkc x (B) | [V prologue] B | B | nvc x (B)
__device__ __forceinline__ void stream_stage(const DosxAttn& a, const Geo& g, const StreamGeo& sg, float* CH, int bk, int st) {
  const int H = a.H, Nk = a.Nk;
  const __amdgpu_buffer_rsrc_t rK =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.kvhat, 0, (uint32_t)((size_t)Nk * a.Bk * H * 4), 0x00020000);
  const int jr = st >> 3, q4 = (st & 7) * 4;
  // ---- K chunks: Ks[j][0..31] = kvhat[j*Bk+bk][32c .. 32c+31], j = jr + 32 i
  {
    const int nrb = g.NKP / 32;
    uint32_t voff[SKR];
#pragma unroll
    for (int i = 0; i < SKR; ++i) voff[i] = (uint32_t)((((size_t)(jr + 32 * i) * a.Bk + bk) * H + q4) * 4);
    float4 r0[SKR], r1[SKR];
    auto issue = [&](float4(&r)[SKR], int c) {
      const int so = __builtin_amdgcn_readfirstlane(c) * (KC * 4);
#pragma unroll
      for (int i = 0; i < SKR; ++i)
        if (i < nrb) r[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rK, voff[i], so, 0));
    };
    auto store = [&](float* buf, const float4(&r)[SKR]) {
#pragma unroll
      for (int i = 0; i < SKR; ++i)
        if (i < nrb) st4(buf + (jr + 32 * i) * LDK + q4, r[i]);
    };
    issue(r0, 0);
    if (sg.nkc > 1) issue(r1, 1);
    store(CH, r0);
    if (sg.nkc > 2) issue(r0, 2);
    __syncthreads();
    for (int c = 0; c < sg.nkc; c += 2) {
      if (c + 1 < sg.nkc) {
        store(CH + sg.CHB, r1);
        if (c + 3 < sg.nkc) issue(r1, c + 3);
      }
      __syncthreads();
      if (c + 1 >= sg.nkc) break;
      if (c + 2 < sg.nkc) {
        store(CH, r0);
        if (c + 4 < sg.nkc) issue(r0, c + 4);
      }
      __syncthreads();
    }
  }
  // ---- V chunks: Vs[jj][0..HP) = kvhat[(32c+jj)*Bk+bk][0..HP)   (both chunk buffers are free again)
  {
    const int ncb = g.HP / 32;
    uint32_t voff[SVC];
#pragma unroll
    for (int i = 0; i < SVC; ++i) {
      const int c = q4 + 32 * i;
      voff[i] = (uint32_t)((((size_t)jr * a.Bk + bk) * H + (c < H ? c : 0)) * 4);
    }
    const int step = KC * a.Bk * H * 4;                    // bytes between consecutive 32-key chunks
    float4 r0[SVC], r1[SVC];
    auto issue = [&](float4(&r)[SVC], int c) {
      const int so = __builtin_amdgcn_readfirstlane(c) * step;
#pragma unroll
      for (int i = 0; i < SVC; ++i)
        if (i < ncb) r[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rK, voff[i], so, 0));
    };
    auto store = [&](float* buf, const float4(&r)[SVC]) {
#pragma unroll
      for (int i = 0; i < SVC; ++i)
        if (i < ncb) st4(buf + jr * g.LDH + q4 + 32 * i, r[i]);
    };
    issue(r0, 0);
    if (sg.nvc > 1) issue(r1, 1);
    store(CH, r0);
    if (sg.nvc > 2) issue(r0, 2);
    __syncthreads();            // (matrix waves: scores stored)
    __syncthreads();            // (matrix waves: softmax / dS done)
    for (int c = 0; c < sg.nvc; c += 2) {
      if (c + 1 < sg.nvc) {
        store(CH + sg.CHB, r1);
        if (c + 3 < sg.nvc) issue(r1, c + 3);
      }
      __syncthreads();
      if (c + 1 >= sg.nvc) break;
      if (c + 2 < sg.nvc) {
        store(CH, r0);
        if (c + 4 < sg.nvc) issue(r0, c + 4);
      }
      __syncthreads();
    }
  }
}

// ---- one or two key tiles (Nk <= 64): RESIDENT keys, Q.K^T split over the feature dimension -------------------------------
// With a single 32-key tile the streamed Q.K^T runs on ONE matrix wave (two tiles: on two) behind one barrier per 32-feature
// chunk (H = 128: four chunks of 16 MFMAs each, ~2.4 us of a ~9.5 us kernel).  The whole key set of the crystal is only HP/32
// chunks of 4.6 / 9.2 KB, so the staging waves fetch ALL of them at once (one barrier); matrix wave w multiplies key tile
// w % tiles with the feature chunks w / tiles, + 4 / tiles, ... and leaves its partial 32 x 32 score tile in LDS; the softmax
// / dS phase adds the 4 / tiles partials of a key while it reads the scores.
constexpr int KCH = QT * LDK;          // floats of one partial score tile [32][36] (= one 32-key K chunk)
constexpr int MAX_KCH = 8;             // H <= 256

__host__ __device__ inline int resident_k_floats(const Geo& g) { return (g.HP / 32) * g.NKP * LDK; }

__device__ __forceinline__ void stage_resident(const DosxAttn& a, const Geo& g, const StreamGeo& sg, float* CH, int bk, int st) {
  const int H = a.H, Nk = a.Nk;
  const __amdgpu_buffer_rsrc_t rK =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.kvhat, 0, (uint32_t)((size_t)Nk * a.Bk * H * 4), 0x00020000);
  const int jr = st >> 3, q4 = (st & 7) * 4;
  const int nkc = g.HP / 32, nrb = g.NKP / 32, kcf = g.NKP * LDK;
  float4 rk[MAX_KCH][2], rv[SVC][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t < nrb) {
      const uint32_t vk = (uint32_t)((((size_t)(jr + 32 * t) * a.Bk + bk) * H + q4) * 4);   // key row (rows >= Nk: zeros by the bounds)
#pragma unroll
      for (int c = 0; c < MAX_KCH; ++c)
        if (c < nkc) rk[c][t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rK, vk, c * (KC * 4), 0));
#pragma unroll
      for (int i = 0; i < SVC; ++i) {  // the same rows again as the V chunks [32 keys][HP] (served by L2)
        const int c = q4 + 32 * i;
        if (i < nkc)
          rv[i][t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
              rK, (uint32_t)((((size_t)(jr + 32 * t) * a.Bk + bk) * H + (c < H ? c : 0)) * 4), 0, 0));
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
    if (t < nrb) {
#pragma unroll
      for (int c = 0; c < MAX_KCH; ++c)
        if (c < nkc) st4(CH + c * kcf + (jr + 32 * t) * LDK + q4, rk[c][t]);
    }
  __syncthreads();                     // keys visible (matrix waves: Qs written)
  __syncthreads();                     // partial scores stored: the key chunks are dead
#pragma unroll
  for (int t = 0; t < 2; ++t)
    if (t < nrb) {
#pragma unroll
      for (int i = 0; i < SVC; ++i)
        if (i < nkc) st4(CH + t * sg.CHB + jr * g.LDH + q4 + 32 * i, rv[i][t]);
    }
  __syncthreads();                     // softmax done, V visible
  for (int c = 0; c < nrb; ++c) __syncthreads();       // one per V chunk (stream_pv)
}

// matrix wave w: Sp[w][32][36] = Qs[:, its feature chunks] . K[key tile w % tiles]^T   (partial over its share of the features)
__device__ __forceinline__ void qk_resident(const float* As, int LDH, const float* CH, int nkc, int NKP, float* Sp, int tid) {
  const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int nkt = NKP >> 5, jt = wave % nkt, step = 4 / nkt, kcf = NKP * LDK;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int c = wave / nkt; c < nkc; c += step) {
    const float* Kc = CH + c * kcf + jt * 32 * LDK;
#pragma unroll
    for (int kk = 0; kk < KC; kk += 8) {
      const float4 av = ld4(As + l31 * LDH + c * KC + kk + 4 * hh);
      const float4 b = ld4(Kc + l31 * LDK + kk + 4 * hh);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b.w, acc, 0, 0, 0);
    }
  }
  float* mine = Sp + wave * KCH;
#pragma unroll
  for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * hh) * LDK + l31] = acc[r];
}

// the score of (query row of `prow`, key j): sum of the partial tiles of the waves that own key tile j / 32
__device__ __forceinline__ float resident_score(const float* prow, int j, int NKP) {
  const int jl = j & 31;
  if (NKP <= 32) return (prow[jl] + prow[KCH + jl]) + (prow[2 * KCH + jl] + prow[3 * KCH + jl]);
  const int jt = j >> 5;
  return prow[jt * KCH + jl] + prow[(jt + 2) * KCH + jl];
}

// matrix waves: S[32][NKP] = A[32][HP] . kvhat^T, one barrier per streamed feature chunk
__device__ __forceinline__ void stream_qk(f32x16 (&acc)[MAX_KT], const float* As, int LDH, const float* CH, const StreamGeo& sg,
                                          int NKP, int tid) {
  const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int nkt = NKP / 32;
#pragma unroll
  for (int t = 0; t < MAX_KT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int c = 0; c < sg.nkc; ++c) {
    const float* Kc = CH + (c & 1) * sg.CHB;
#pragma unroll
    for (int kk = 0; kk < KC; kk += 8) {
      const float4 av = ld4(As + l31 * LDH + c * KC + kk + 4 * hh);
#pragma unroll
      for (int t = 0; t < MAX_KT; ++t) {
        const int jt = wave + 4 * t;
        if (jt >= nkt) continue;
        const float4 b = ld4(Kc + (jt * 32 + l31) * LDK + kk + 4 * hh);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b.w, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
  }
}

// matrix waves: O[32][HP] = P[32][NKP] . kvhat, one barrier per streamed key chunk
__device__ __forceinline__ void stream_pv(f32x16 (&acc)[MAX_CT], const float* Ps, int LDS_, const float* CH, const StreamGeo& sg,
                                          int HP, int LDH, int tid) {
  const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int nct = HP / 32;
#pragma unroll
  for (int t = 0; t < MAX_CT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int c = 0; c < sg.nvc; ++c) {
    const float* Vc = CH + (c & 1) * sg.CHB;
#pragma unroll
    for (int kk = 0; kk < KC; kk += 8) {
      const float4 av = ld4(Ps + l31 * LDS_ + c * KC + kk + 4 * hh);
#pragma unroll
      for (int t = 0; t < MAX_CT; ++t) {
        const int ct = wave + 4 * t;
        if (ct >= nct) continue;
        const float* bp = Vc + (kk + 4 * hh) * LDH + ct * 32 + l31;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bp[0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bp[LDH], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bp[2 * LDH], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bp[3 * LDH], acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
  }
}

__device__ __forceinline__ void store_scores1(const f32x16 (&acc)[MAX_KT], float* Ss, int LDS_, int NKP, int tid) {
  const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int nkt = NKP / 32;
#pragma unroll
  for (int t = 0; t < MAX_KT; ++t) {
    const int jt = wave + 4 * t;
    if (jt >= nkt) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) Ss[((r & 3) + 8 * (r >> 2) + 4 * hh) * LDS_ + jt * 32 + l31] = acc[t][r];
  }
}

template <int NJ, bool RESIDENT>
__global__ __launch_bounds__(512) void attn_fwd_stream_kernel(const DosxAttn a) {
  DOSX_SET_MAIN_PRIO();
  extern __shared__ __align__(16) float sm[];
  const Geo g = make_geo(a.H, a.Nk);
  StreamGeo sg;
  sg.nkc = g.HP / 32; sg.nvc = g.NKP / 32; sg.CHB = chunk_buf_floats(g.NKP, g.LDH);
  float* Qs = sm;                                   // [32][LDH]   (later: output tile)
  float* Ss = Qs + QT * g.LDH;                      // [32][LDS_]  scores -> P
  float* CH = Ss + QT * g.LDS_;                     // two chunk buffers
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q16 = lane & 15;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int s0 = blockIdx.x * QT, bq = blockIdx.y, bk = bq % a.Bk;
  // RESIDENT (Nk <= 64 when it fits the LDS): keys resident, Q.K^T split over the features (see qk_resident)
  float* Sp = CH + max(2 * sg.CHB, resident_k_floats(g));      // [4][32][36] partial score tiles (RESIDENT only)
  if (wave_u >= 4) {
    if constexpr (RESIDENT) stage_resident(a, g, sg, CH, bk, tid - 256);
    else stream_stage(a, g, sg, CH, bk, tid - 256);
    return;                                         // (the epilogue barrier below counts live waves only)
  }
  const int H = a.H, Sq = a.Sq, Nk = a.Nk;
  const bool raw_q = (a.flags & DOSX_ATTN_RAW_Q) != 0;
  const float invH = 1.f / (float)H;

  // ---- query rows: global -> registers (kept for the residual) -> LayerNorm -> (x gamma) -> LDS ----
  float4 xr[RP][KCB], g0[KCB], b0[KCB];
#pragma unroll
  for (int k = 0; k < KCB; ++k) {
    const int c = q16 * 4 + 64 * k, cc = c < H ? c : 0;
    g0[k] = ld4(a.gamma0 + cc);
    b0[k] = ld4(a.beta0 + cc);
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      const int s = min(s0 + row_of(wave, p, lane), Sq - 1);
      xr[p][k] = ld4(a.x + ((size_t)s * a.q_stride_s + (size_t)bq * a.q_stride_b) * H + cc);
    }
  }
  {
    float mean[RP], rstd[RP];
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < KCB; ++k)
        if (q16 * 4 + 64 * k < H) t += (xr[p][k].x + xr[p][k].y) + (xr[p][k].z + xr[p][k].w);
      mean[p] = t;
    }
#pragma unroll
    for (int p = 0; p < RP; ++p) mean[p] = row16_sum(mean[p]) * invH;
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < KCB; ++k) {
        if (q16 * 4 + 64 * k < H) {
          const float p0 = xr[p][k].x - mean[p], p1 = xr[p][k].y - mean[p], p2 = xr[p][k].z - mean[p],
                      p3 = xr[p][k].w - mean[p];
          t += (p0 * p0 + p1 * p1) + (p2 * p2 + p3 * p3);
        }
      }
      rstd[p] = t;
    }
#pragma unroll
    for (int p = 0; p < RP; ++p) rstd[p] = rsqrtf(row16_sum(rstd[p]) * invH + DOSX_LN_EPS);
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      const int lr = row_of(wave, p, lane), s = s0 + lr;
#pragma unroll
      for (int k = 0; k < KCB; ++k) {
        const int c = q16 * 4 + 64 * k;
        if (c >= g.HP) continue;
        const float4 v = xr[p][k];
        float4 o = v;
        if (!raw_q)
          o = make_float4((v.x - mean[p]) * rstd[p] * g0[k].x + b0[k].x, (v.y - mean[p]) * rstd[p] * g0[k].y + b0[k].y,
                          (v.z - mean[p]) * rstd[p] * g0[k].z + b0[k].z, (v.w - mean[p]) * rstd[p] * g0[k].w + b0[k].w);
        o = make_float4(o.x * g0[k].x, o.y * g0[k].y, o.z * g0[k].z, o.w * g0[k].w);      // key gamma folded into Q
        if (!(c < H && s < Sq)) o = f4zero();
        st4(Qs + lr * g.LDH + c, o);
      }
      if (!raw_q && a.qstats && q16 == 0 && s < Sq) {
        a.qstats[2 * ((size_t)s * a.Bq + bq)] = mean[p];
        a.qstats[2 * ((size_t)s * a.Bq + bq) + 1] = rstd[p];
      }
    }
  }
  __syncthreads();

  if constexpr (RESIDENT) {
    qk_resident(Qs, g.LDH, CH, sg.nkc, g.NKP, Sp, tid);
  } else {
    f32x16 sacc[MAX_KT];
    stream_qk(sacc, Qs, g.LDH, CH, sg, g.NKP, tid);
    store_scores1(sacc, Ss, g.LDS_, g.NKP, tid);
  }
  __syncthreads();

  float psum[RP];
#pragma unroll
  for (int p = 0; p < RP; ++p) psum[p] = 1.f;
  // ---- exact fp32 softmax over the Nk keys (padded atoms included, like the reference; key_ptr: the crystal's own keys only) ----
  {
    const float scale = rsqrtf((float)H);
    const int bk_ = bq % a.Bk;
    const int nk = a.key_ptr ? min(a.key_ptr[bk_ + 1] - a.key_ptr[bk_], Nk) : Nk;
    float v[RP][NJ], mx[RP], sum[RP];
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      const float* row = Ss + row_of(wave, p, lane) * g.LDS_;
      const float* prow = Sp + row_of(wave, p, lane) * LDK;
      mx[p] = -INFINITY;
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const int j = q16 + 16 * jj;
        float sc_ = 0.f;
        if constexpr (RESIDENT) { if (j < Nk) sc_ = resident_score(prow, j, g.NKP); }
        else sc_ = row[j];
        const float t = j < nk ? sc_ * scale : -INFINITY;          // (nk: the crystal's own key count with DosxAttn.key_ptr, else Nk)
        v[p][jj] = t;
        mx[p] = fmaxf(mx[p], t);
      }
    }
#pragma unroll
    for (int p = 0; p < RP; ++p) mx[p] = row16_max(mx[p]);
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      sum[p] = 0.f;
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const float e = (q16 + 16 * jj) < nk ? expf(v[p][jj] - mx[p]) : 0.f;
        v[p][jj] = e;
        sum[p] += e;
      }
    }
#pragma unroll
    for (int p = 0; p < RP; ++p) sum[p] = 1.f / row16_sum(sum[p]);
    // attention dropout (multihead_attention.py:70: F.dropout on the softmax output): the saved probabilities stay the
    // un-dropped P (the backward needs them), the P.V product takes P' = P o M, M in {0, 1/(1-p)} read from drop_mask
    float mk[RP][NJ];
#pragma unroll
    for (int p = 0; p < RP; ++p)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) mk[p][jj] = 1.f;
    if (a.drop_mask) {
#pragma unroll
      for (int p = 0; p < RP; ++p) {
        const int s = min(s0 + row_of(wave, p, lane), Sq - 1);
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
          const int j = q16 + 16 * jj;
          mk[p][jj] = a.drop_mask[((size_t)bq * Sq + s) * Nk + (j < Nk ? j : 0)];
        }
      }
    }
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      const int lr = row_of(wave, p, lane), s = s0 + lr;
      float* row = Ss + lr * g.LDS_;
      float t = 0.f;
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const int j = q16 + 16 * jj;
        if (j >= g.NKP) continue;
        const float pr = v[p][jj] * sum[p];           // 0 beyond Nk
        const float prm = pr * mk[p][jj];
        row[j] = prm;
        t += prm;
        if (j < Nk && s < Sq) a.probs[((size_t)bq * Sq + s) * Nk + j] = pr;
      }
      if (NJ * 16 < g.NKP) {
        for (int j = NJ * 16 + q16; j < g.NKP; j += 16) row[j] = 0.f;
      }
      if (a.drop_mask) psum[p] = row16_sum(t);        // rows of P' no longer sum to 1: P'.(k̂ gamma + beta) = (P'.k̂) gamma + beta sum(P')
    }
  }
  __syncthreads();

  f32x16 oacc[MAX_CT];
  stream_pv(oacc, Ss, g.LDS_, CH, sg, g.HP, g.LDH, tid);
  store_out_tile(oacc, Qs, g.LDH, g.HP, tid);
  __syncthreads();                                  // (matrix waves only: the staging waves have exited)

  // ---- epilogue: (P.k̂) gamma + beta + residual, statistics of the output rows (feeds LN1) ----
  {
    const bool no_res = (a.flags & DOSX_ATTN_NO_RESIDUAL) != 0;
    float4 o[RP][KCB];
    float mean[RP], var[RP];
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      const int lr = row_of(wave, p, lane), s = s0 + lr;
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < KCB; ++k) {
        const int c = q16 * 4 + 64 * k;
        float4 v = f4zero();
        if (c < H) {
          const float4 r = ld4(Qs + lr * g.LDH + c);
          v = make_float4(r.x * g0[k].x + b0[k].x * psum[p], r.y * g0[k].y + b0[k].y * psum[p],
                          r.z * g0[k].z + b0[k].z * psum[p], r.w * g0[k].w + b0[k].w * psum[p]);
          if (!no_res) v = f4add(v, xr[p][k]);
          if (s < Sq) st4(a.out + ((size_t)s * a.Bq + bq) * H + c, v);
          t += (v.x + v.y) + (v.z + v.w);
        }
        o[p][k] = v;
      }
      mean[p] = t;
    }
    if (a.out_stats) {
#pragma unroll
      for (int p = 0; p < RP; ++p) mean[p] = row16_sum(mean[p]) * invH;
#pragma unroll
      for (int p = 0; p < RP; ++p) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < KCB; ++k) {
          if (q16 * 4 + 64 * k < H) {
            const float p0 = o[p][k].x - mean[p], p1 = o[p][k].y - mean[p], p2 = o[p][k].z - mean[p],
                        p3 = o[p][k].w - mean[p];
            t += (p0 * p0 + p1 * p1) + (p2 * p2 + p3 * p3);
          }
        }
        var[p] = t;
      }
#pragma unroll
      for (int p = 0; p < RP; ++p) var[p] = row16_sum(var[p]) * invH;
#pragma unroll
      for (int p = 0; p < RP; ++p) {
        const int s = s0 + row_of(wave, p, lane);
        if (q16 == 0 && s < Sq) {
          a.out_stats[2 * ((size_t)s * a.Bq + bq)] = mean[p];
          a.out_stats[2 * ((size_t)s * a.Bq + bq) + 1] = rsqrtf(var[p] + DOSX_LN_EPS);
        }
      }
      if (a.ln1_out) {      // the next LayerNorm on the rows still in registers (DosxAttn.ln1_*)
#pragma unroll
        for (int k = 0; k < KCB; ++k) {
          const int c = q16 * 4 + 64 * k;
          if (c >= H) continue;
          const float4 g1 = ld4(a.ln1_gamma + c), b1 = ld4(a.ln1_beta + c);
#pragma unroll
          for (int p = 0; p < RP; ++p) {
            const int s = s0 + row_of(wave, p, lane);
            const float m = mean[p], rs = rsqrtf(var[p] + DOSX_LN_EPS);
            if (s < Sq)
              st4(a.ln1_out + ((size_t)s * a.Bq + bq) * H + c,
                  make_float4((o[p][k].x - m) * rs * g1.x + b1.x, (o[p][k].y - m) * rs * g1.y + b1.y,
                              (o[p][k].z - m) * rs * g1.z + b1.z, (o[p][k].w - m) * rs * g1.w + b1.w));
          }
        }
      }
    }
  }
}

// PKV: the kernel ALSO produces this tile's share of dK + dV ("partial key gradient"), while P, dS, dO and the
// normalised query rows are in LDS anyway:    part[bq][tile][j][:] = sum_{q in tile} P[q,j] dO[q,:] + dS[q,j] Q[q,:]
// (Q = LN0(x) gamma0 + beta0).  A small reduction kernel (attn_dkv_reduce_kernel) sums the 2-4 partials of every key row
// and applies the key-side chain rule.  That replaces attn_bwd_dkv_kernel for NKP <= 64 (cfg2: cross attention over
// <= 12 atoms, self attention over 51 bins): that kernel re-streamed every dO / x row and the dS round trip through HBM
// behind one barrier per 16 queries - 27 us per launch for 0.08 GF - where this costs 32-64 MFMAs per wave.
constexpr int DKR = 16;          // key rows per reduction group = partial-sum rows per crystal: ceil(Nk / 16)
template <bool SC1>
__device__ __forceinline__ void dkv_reduce_group(const DosxAttn& a, const int nqt, const int grp, const int ngroups,
                                                 const int bk, float (*Pp)[2 * 256], const int tid);

template <int NJ, bool PKV, bool RESIDENT>
__global__ __launch_bounds__(512) void attn_bwd_dq_stream_kernel(const DosxAttn a) {
  DOSX_SET_MAIN_PRIO();
  extern __shared__ __align__(16) float sm[];
  const Geo g = make_geo(a.H, a.Nk);
  StreamGeo sg;
  sg.nkc = g.HP / 32; sg.nvc = g.NKP / 32; sg.CHB = chunk_buf_floats(g.NKP, g.LDH);
  float* Ds = sm;                                   // [32][LDH]  dOut*gamma tile, later dS.k̂ tile
  float* Ss = Ds + QT * g.LDH;                      // [32][LDS_] dP -> dS
  float* CH = Ss + QT * g.LDS_;                     // two chunk buffers; later [16][2][HP] column partial sums
  float* Pp = CH;
  float* Ps2 = CH + max(max(2 * sg.CHB, 32 * g.HP), RESIDENT ? resident_k_floats(g) : 0);     // PKV: [32][LDS_] P tile
  float* dOr = Ps2 + QT * g.LDS_;                   // PKV: [32][LDH] raw dO rows
  float* Ql = Ds;                                   // PKV: [32][LDH] LN0(x) gamma0 + beta0 - takes over Ds once dP is done
  float* Sp = PKV ? dOr + QT * g.LDH + 64 : Ps2;    // RESIDENT: [4][32][36] partial dP tiles, behind everything else
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q16 = lane & 15;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int s0 = blockIdx.x * QT, bq = blockIdx.y, bk = bq % a.Bk;
  if (wave_u >= 4) {
    if constexpr (RESIDENT) stage_resident(a, g, sg, CH, bk, tid - 256);
    else stream_stage(a, g, sg, CH, bk, tid - 256);
    return;
  }
  const int H = a.H, Sq = a.Sq, Nk = a.Nk;
  const bool raw_q = (a.flags & DOSX_ATTN_RAW_Q) != 0, no_res = (a.flags & DOSX_ATTN_NO_RESIDUAL) != 0;
  const float invH = 1.f / (float)H;

  float4 go[RP][KCB], xr[RP][KCB], g0[KCB];
  float mean[RP], rstd[RP], pr[RP][NJ];
#pragma unroll
  for (int k = 0; k < KCB; ++k) g0[k] = ld4(a.gamma0 + ((q16 * 4 + 64 * k) < H ? (q16 * 4 + 64 * k) : 0));
#pragma unroll
  for (int p = 0; p < RP; ++p) {
    const int s = min(s0 + row_of(wave, p, lane), Sq - 1);
    const size_t orow = (size_t)s * a.Bq + bq;
#pragma unroll
    for (int k = 0; k < KCB; ++k) {
      const int c = q16 * 4 + 64 * k, cc = c < H ? c : 0;
      go[p][k] = ld4(a.dout + orow * H + cc);
      xr[p][k] = ld4(a.x + ((size_t)s * a.q_stride_s + (size_t)bq * a.q_stride_b) * H + cc);
    }
    mean[p] = raw_q ? 0.f : a.qstats[2 * orow];
    rstd[p] = raw_q ? 1.f : a.qstats[2 * orow + 1];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const int j = q16 + 16 * jj;
      pr[p][jj] = a.probs[((size_t)bq * Sq + s) * Nk + (j < Nk ? j : 0)];
    }
  }
  // attention dropout: dP = M o dP' with dP' = dO.V^T.  Without a mask the row constant dO.beta0 of dP' cancels in
  // dS = P o (dP - sum P dP) and is never formed; with a mask it does not (sum_j P_j M_j != 1): cq = dO . beta0
  float mk[RP][NJ], cq[RP];
#pragma unroll
  for (int p = 0; p < RP; ++p) {
    cq[p] = 0.f;
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) mk[p][jj] = 1.f;
  }
  if (a.drop_mask) {
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      const int s = min(s0 + row_of(wave, p, lane), Sq - 1);
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const int j = q16 + 16 * jj;
        mk[p][jj] = a.drop_mask[((size_t)bq * Sq + s) * Nk + (j < Nk ? j : 0)];
      }
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < KCB; ++k) {
        const int c = q16 * 4 + 64 * k;
        if (c < H) {
          const float4 b0 = ld4(a.beta0 + c);
          t += (go[p][k].x * b0.x + go[p][k].y * b0.y) + (go[p][k].z * b0.z + go[p][k].w * b0.w);
        }
      }
      cq[p] = row16_sum(t);
    }
  }
#pragma unroll
  for (int p = 0; p < RP; ++p) {
    const int lr = row_of(wave, p, lane);
#pragma unroll
    for (int k = 0; k < KCB; ++k) {
      const int c = q16 * 4 + 64 * k;
      if (c >= g.HP) continue;
      const bool rv = c < H && (s0 + lr) < Sq;
      float4 d = make_float4(go[p][k].x * g0[k].x, go[p][k].y * g0[k].y, go[p][k].z * g0[k].z, go[p][k].w * g0[k].w);
      if (!rv) d = f4zero();
      st4(Ds + lr * g.LDH + c, d);
      if constexpr (PKV) st4(dOr + lr * g.LDH + c, rv ? go[p][k] : f4zero());
    }
  }
  __syncthreads();

  // dP (up to a row constant) = (dO gamma) . k̂^T
  if constexpr (RESIDENT) {
    qk_resident(Ds, g.LDH, CH, sg.nkc, g.NKP, Sp, tid);
  } else {
    f32x16 sacc[MAX_KT];
    stream_qk(sacc, Ds, g.LDH, CH, sg, g.NKP, tid);
    store_scores1(sacc, Ss, g.LDS_, g.NKP, tid);
  }
  __syncthreads();

  // dS = P * (dP - rowsum(P*dP)) * scale
  {
    const float scale = rsqrtf((float)H);
    float dp[RP][NJ], dot[RP];
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      const float* row = Ss + row_of(wave, p, lane) * g.LDS_;
      const float* prow = Sp + row_of(wave, p, lane) * LDK;
      dot[p] = 0.f;
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const int j = q16 + 16 * jj;
        float t = 0.f;
        if constexpr (RESIDENT) { if (j < Nk) t = resident_score(prow, j, g.NKP); }
        else t = j < Nk ? row[j] : 0.f;
        if (a.drop_mask) t = (t + cq[p]) * mk[p][jj];
        if (j < Nk) dot[p] += pr[p][jj] * t;
        dp[p][jj] = t;
      }
    }
#pragma unroll
    for (int p = 0; p < RP; ++p) dot[p] = row16_sum(dot[p]);
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      const int lr = row_of(wave, p, lane), s = s0 + lr;
      float* row = Ss + lr * g.LDS_;
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const int j = q16 + 16 * jj;
        if (j >= g.NKP) continue;
        const float ds = (j < Nk && s < Sq) ? pr[p][jj] * (dp[p][jj] - dot[p]) * scale : 0.f;
        row[j] = ds;
        if constexpr (PKV) Ps2[lr * g.LDS_ + j] = (j < Nk && s < Sq) ? pr[p][jj] * mk[p][jj] : 0.f;   // dV takes P' = P o M
        else if (j < Nk && s < Sq) a.dscores[((size_t)bq * Sq + s) * Nk + j] = ds;
      }
      if (NJ * 16 < g.NKP) {
        for (int j = NJ * 16 + q16; j < g.NKP; j += 16) {
          row[j] = 0.f;
          if constexpr (PKV) Ps2[lr * g.LDS_ + j] = 0.f;
        }
      }
    }
  }
  __syncthreads();

  f32x16 oacc[MAX_CT];
  stream_pv(oacc, Ss, g.LDS_, CH, sg, g.HP, g.LDH, tid);
  store_out_tile(oacc, Ds, g.LDH, g.HP, tid);
  __syncthreads();                                  // (matrix waves only from here on)

  // dq_ln = (dS.k̂) gamma ; LN0 backward on the query rows + residual; partial dgamma0 / dbeta0 (query side)
  {
    float4 pg[KCB], pb[KCB], d[RP][KCB], xh[RP][KCB];
    float s1[RP], s2[RP];
#pragma unroll
    for (int k = 0; k < KCB; ++k) { pg[k] = f4zero(); pb[k] = f4zero(); }
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      const int lr = row_of(wave, p, lane);
      s1[p] = 0.f; s2[p] = 0.f;
#pragma unroll
      for (int k = 0; k < KCB; ++k) {
        const int c = q16 * 4 + 64 * k;
        const bool rv = c < H && (s0 + lr) < Sq;
        float4 dd = rv ? ld4(Ds + lr * g.LDH + c) : f4zero();
        dd = make_float4(dd.x * g0[k].x, dd.y * g0[k].y, dd.z * g0[k].z, dd.w * g0[k].w);
        const float4 xv = xr[p][k];
        float4 h = make_float4((xv.x - mean[p]) * rstd[p], (xv.y - mean[p]) * rstd[p], (xv.z - mean[p]) * rstd[p],
                               (xv.w - mean[p]) * rstd[p]);
        if (!rv) h = f4zero();
        d[p][k] = dd; xh[p][k] = h;
        pg[k].x += dd.x * h.x; pg[k].y += dd.y * h.y; pg[k].z += dd.z * h.z; pg[k].w += dd.w * h.w;
        pb[k] = f4add(pb[k], dd);
        const float4 dh = make_float4(dd.x * g0[k].x, dd.y * g0[k].y, dd.z * g0[k].z, dd.w * g0[k].w);
        s1[p] += (dh.x + dh.y) + (dh.z + dh.w);
        s2[p] += (dh.x * h.x + dh.y * h.y) + (dh.z * h.z + dh.w * h.w);
      }
    }
    if (!raw_q) {
#pragma unroll
      for (int p = 0; p < RP; ++p) { s1[p] = row16_sum(s1[p]) * invH; s2[p] = row16_sum(s2[p]) * invH; }
    }
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      const int s = s0 + row_of(wave, p, lane);
#pragma unroll
      for (int k = 0; k < KCB; ++k) {
        const int c = q16 * 4 + 64 * k;
        if (!(c < H && s < Sq)) continue;
        const float4 dd = d[p][k], h = xh[p][k];
        float4 o;
        if (raw_q) o = dd;
        else
          o = make_float4(rstd[p] * (dd.x * g0[k].x - s1[p] - h.x * s2[p]), rstd[p] * (dd.y * g0[k].y - s1[p] - h.y * s2[p]),
                          rstd[p] * (dd.z * g0[k].z - s1[p] - h.z * s2[p]), rstd[p] * (dd.w * g0[k].w - s1[p] - h.w * s2[p]));
        if (!no_res) o = f4add(o, go[p][k]);
        st4(a.dx + ((size_t)s * a.Bq + bq) * H + c, o);
      }
    }
    const int slot = wave * 4 + (lane >> 4);
#pragma unroll
    for (int k = 0; k < KCB; ++k) {
      const int c = q16 * 4 + 64 * k;
      if (c >= g.HP) continue;
      if (raw_q) { pg[k] = f4zero(); pb[k] = f4zero(); }
      st4(Pp + slot * 2 * g.HP + c, pg[k]);
      st4(Pp + slot * 2 * g.HP + g.HP + c, pb[k]);
    }
  }
  __syncthreads();
  float* prow = a.partials_q + ((size_t)bq * gridDim.x + blockIdx.x) * 2 * H;
  for (int c = tid; c < 2 * H; c += 256) {
    const int which = c / H, col = c % H;
    const int o = which * g.HP + col;
    float t = 0.f;
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) t += Pp[sl * 2 * g.HP + o];
    prow[c] = t;
  }

  if constexpr (PKV) {          // dq is done and stored; each quarter wave is past its last read of its own Ds rows
                                // (the dS.k̂ tile): they become Q = LN0(x) gamma0 + beta0 for the key-gradient product
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      const int lr = row_of(wave, p, lane);
#pragma unroll
      for (int k = 0; k < KCB; ++k) {
        const int c = q16 * 4 + 64 * k;
        if (c >= g.HP) continue;
        const bool rv = c < H && (s0 + lr) < Sq;
        const float4 b0 = ld4(a.beta0 + (c < H ? c : 0));
        const float4 xv = xr[p][k];
        float4 q = xv;
        if (!raw_q)
          q = make_float4((xv.x - mean[p]) * rstd[p] * g0[k].x + b0.x, (xv.y - mean[p]) * rstd[p] * g0[k].y + b0.y,
                          (xv.z - mean[p]) * rstd[p] * g0[k].z + b0.z, (xv.w - mean[p]) * rstd[p] * g0[k].w + b0.w);
        st4(Ql + lr * g.LDH + c, rv ? q : f4zero());
      }
    }
  }
  if constexpr (PKV) __syncthreads();
  if constexpr (PKV) {
    // ---- this tile's share of dK + dV: [NKP keys] x [32 queries] . [32 queries] x [HP].  LAST in the kernel: placed
    // before the dS.k̂ product its global stores sat in front of that loop's barriers (s_waitcnt vmcnt(0)) ----
    constexpr int NKT = NJ == 1 ? 1 : 2;            // 32-key tiles (NJ = 1: Nk <= 16; NJ = 4: Nk <= 64)
    const int l31 = lane & 31, hh = lane >> 5, nct = g.HP / 32;
    f32x16 dacc[NKT][MAX_CT], dacc2[NKT][MAX_CT];     // P^T dO and dS^T Q in separate chains (a dependent MFMA chain
#pragma unroll                                       //  issues every ~87 clk instead of every 64)
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int t = 0; t < MAX_CT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dacc[kt][t][r] = 0.f; dacc2[kt][t][r] = 0.f; }
#pragma unroll
    for (int mm = 0; mm < QT; mm += 2) {          // fully unrolled: the fragment reads of later steps issue under the MFMAs
      float pa[NKT], sa[NKT];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {            // (NKT = 2 with NKP = 32: the second tile reads past the row, into the
        pa[kt] = Ps2[(mm + hh) * g.LDS_ + kt * 32 + l31];   //  next row / buffer - in bounds of the LDS, rows j >= Nk are
        sa[kt] = Ss[(mm + hh) * g.LDS_ + kt * 32 + l31];    //  never stored)
      }
#pragma unroll
      for (int t = 0; t < MAX_CT; ++t) {
        const int ct = wave + 4 * t;
        if (ct >= nct) continue;
        const float b1 = dOr[(mm + hh) * g.LDH + ct * 32 + l31];
        const float b2 = Ql[(mm + hh) * g.LDH + ct * 32 + l31];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          dacc[kt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[kt], b1, dacc[kt][t], 0, 0, 0);
          dacc2[kt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[kt], b2, dacc2[kt][t], 0, 0, 0);
        }
      }
    }
    float* part = a.dkv_part + ((size_t)bq * gridDim.x + blockIdx.x) * (size_t)Nk * H;
    const bool fused = a.dkv_cnt != nullptr;          // the last arriving workgroup of a key crystal finishes dK + dV itself
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int t = 0; t < MAX_CT; ++t) {
        const int ct = wave + 4 * t, col = ct * 32 + l31;
        if (ct >= nct) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (j < Nk && col < H) {
            const float v = dacc[kt][t][r] + dacc2[kt][t][r];
            if (fused) __hip_atomic_store(part + (size_t)j * H + col, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1
            else part[(size_t)j * H + col] = v;
          }
        }
      }
    if (fused) {
      // In-launch reduction of the partial key gradients (same protocol as DosxWgrad's finished mode: write-through
      // publish, every wave drains, barrier, one ticket per workgroup on the key crystal's counter).  The workgroup that
      // draws the last ticket of crystal bk - rep * tiles arrive - runs the reduction groups the stand-alone
      // attn_dkv_reduce_kernel would run, in the same order with the same arithmetic: bitwise the two-launch result.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      int* flag = reinterpret_cast<int*>(sm);
      const int arrivers = (a.Bq / a.Bk) * (int)gridDim.x;
      if (tid == 0) *flag = dosx_ticket(a.dkv_cnt + bk);
      __syncthreads();
      const bool last = *flag == arrivers - 1;
      __syncthreads();                                  // (the flag word is about to be overwritten by the reduction's LDS rows)
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int ngroups = (Nk + DKR - 1) / DKR;
        float (*PpR)[2 * 256] = reinterpret_cast<float (*)[2 * 256]>(sm);
        for (int grp = 0; grp < ngroups; ++grp) {
          dkv_reduce_group<true>(a, (int)gridDim.x, grp, ngroups, bk, PpR, tid);
          __syncthreads();
        }
        if (tid == 0) __hip_atomic_store(a.dkv_cnt + bk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }

}

// ================================== backward: dk + dv ============================================
// One workgroup per (KG x 32-key group, crystal):  dK[keys,H] = sum_q P^T dO + dS^T LN0(x)  over every query
// row of every query batch entry that maps to this crystal (bq = bk + i*Bk); then dkvhat += dK * gamma0 and the
// key-side partial sums of dgamma0 / dbeta0.  Wave-specialised like the other kernels: 4 staging waves stream
// 16-query chunks (dO rows, the query rows with LayerNorm + affine applied, the P and dS columns of this key
// group) into two LDS stage buffers, 4 matrix waves accumulate; KG = 2 key tiles per workgroup halve the
// number of times every dO / x row is re-read.  All 8 waves run the row epilogue.
constexpr int DQC = 16;    // queries per streamed chunk

template <int KG>
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(const DosxAttn a) {
  extern __shared__ __align__(16) float sm[];
  const Geo g = make_geo(a.H, a.Nk);
  constexpr int LDP = 32 * KG + 4;
  const int STG = 2 * DQC * g.LDH + 2 * DQC * LDP;        // floats of one stage buffer: dOs | Qs | Pc | Sc
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5, q16 = lane & 15;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int j0 = blockIdx.x * 32 * KG, bk = blockIdx.y;
  const int H = a.H, Sq = a.Sq, Nk = a.Nk;
  const int nct = g.HP / 32;
  const bool raw_q = (a.flags & DOSX_ATTN_RAW_Q) != 0;
  const int nchunks = (Sq + DQC - 1) / DQC, nit = (a.Bq / a.Bk) * nchunks;

  // epilogue operands, fetched up front by every wave: this wave's 4*KG key rows (quarter wave per row)
  float4 kh[KG][KCB], dk0[KG][KCB], g0[KCB];
#pragma unroll
  for (int k = 0; k < KCB; ++k) g0[k] = ld4(a.gamma0 + ((q16 * 4 + 64 * k) < H ? (q16 * 4 + 64 * k) : 0));
#pragma unroll
  for (int p = 0; p < KG; ++p) {
    const int j = min(j0 + wave * 4 * KG + p * 4 + (lane >> 4), Nk - 1);
    const size_t krow = ((size_t)j * a.Bk + bk) * H;
#pragma unroll
    for (int k = 0; k < KCB; ++k) {
      const int c = q16 * 4 + 64 * k, cc = c < H ? c : 0;
      kh[p][k] = ld4(a.kvhat + krow + cc);
      dk0[p][k] = a.dkv_accumulate ? ld4(a.dkvhat + krow + cc) : f4zero();
    }
  }

  f32x16 acc[KG][MAX_CT];
  ASTAMP(0);
  if (wave_u >= 4) {
    // =============================== staging waves ===============================================
    const int st = tid - 256, row = st >> 4, cq = (st & 15) * 4;
    const uint32_t xrows = (uint32_t)((size_t)(Sq - 1) * a.q_stride_s + (size_t)(a.Bq - 1) * a.q_stride_b + 1);
    const __amdgpu_buffer_rsrc_t rO =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.dout, 0, (uint32_t)((size_t)Sq * a.Bq * H * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, xrows * (uint32_t)(H * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rP =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.probs, 0, (uint32_t)((size_t)a.Bq * Sq * Nk * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rS =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.dscores, 0, (uint32_t)((size_t)a.Bq * Sq * Nk * 4), 0x00020000);
    const bool has_mask = a.drop_mask != nullptr;       // attention dropout: dV = (P o M)^T dO
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(has_mask ? a.drop_mask : a.probs), 0, (uint32_t)((size_t)a.Bq * Sq * Nk * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(raw_q ? (const float*)a.dout : a.qstats), 0, (uint32_t)((size_t)Sq * a.Bq * 8), 0x00020000);
    uint32_t vO[KCB], vX[KCB], vP[2 * KG];
    float4 gq[KCB], bq4[KCB];
#pragma unroll
    for (int k = 0; k < KCB; ++k) {
      const int c = cq + 64 * k, cc = c < H ? c : 0;
      vO[k] = (uint32_t)(((size_t)row * a.Bq * H + cc) * 4);
      vX[k] = (uint32_t)(((size_t)row * a.q_stride_s * H + cc) * 4);
      gq[k] = ld4(a.gamma0 + cc);
      bq4[k] = ld4(a.beta0 + cc);
    }
#pragma unroll
    for (int u = 0; u < 2 * KG; ++u) vP[u] = (uint32_t)(((size_t)row * Nk + min(j0 + (st & 15) + 16 * u, Nk - 1)) * 4);
    const uint32_t vT = (uint32_t)((size_t)row * a.Bq * 8);
    struct Set {
      float4 d[KCB], x[KCB];
      float pp[2 * KG], ss[2 * KG], mm[2 * KG], mean, rstd;
      bool rok;
    };
    Set s0, s1;
    auto issue = [&](Set& q, int it) {
      const int itu = __builtin_amdgcn_readfirstlane(it);
      const int bq = bk + (itu / nchunks) * a.Bk, sb = (itu % nchunks) * DQC;
      q.rok = (sb + row) < Sq;
      const int soO = (sb * a.Bq + bq) * H * 4;
      const int soX = (int)(((size_t)sb * a.q_stride_s + (size_t)bq * a.q_stride_b) * H * 4);
      const int soP = (bq * Sq + sb) * Nk * 4;
      const int soT = (sb * a.Bq + bq) * 8;
#pragma unroll
      for (int k = 0; k < KCB; ++k) {
        if (k < (g.HP + 63) / 64) {
          q.d[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rO, vO[k], soO, 0));
          q.x[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rX, vX[k], soX, 0));
        }
      }
#pragma unroll
      for (int u = 0; u < 2 * KG; ++u) {
        q.pp[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rP, vP[u], soP, 0));
        q.ss[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rS, vP[u], soP, 0));
        q.mm[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rM, vP[u], soP, 0));
      }
      q.mean = 0.f; q.rstd = 1.f;
      if (!raw_q) {
        q.mean = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rT, vT, soT, 0));
        q.rstd = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rT, vT + 4, soT, 0));
      }
    };
    auto store = [&](float* buf, const Set& q) {
      float* dOs = buf;
      float* Qs = buf + DQC * g.LDH;
      float* Pc = Qs + DQC * g.LDH;
      float* Sc = Pc + DQC * LDP;
#pragma unroll
      for (int k = 0; k < KCB; ++k) {
        const int c = cq + 64 * k;
        if (c >= g.HP) continue;
        float4 d = q.d[k], x = q.x[k];
        if (!raw_q)
          x = make_float4((x.x - q.mean) * q.rstd * gq[k].x + bq4[k].x, (x.y - q.mean) * q.rstd * gq[k].y + bq4[k].y,
                          (x.z - q.mean) * q.rstd * gq[k].z + bq4[k].z, (x.w - q.mean) * q.rstd * gq[k].w + bq4[k].w);
        if (!(q.rok && c < H)) { d = f4zero(); x = f4zero(); }
        st4(dOs + row * g.LDH + c, d);
        st4(Qs + row * g.LDH + c, x);
      }
#pragma unroll
      for (int u = 0; u < 2 * KG; ++u) {
        const int jl = (st & 15) + 16 * u;
        const bool ok = q.rok && (j0 + jl) < Nk;
        Pc[row * LDP + jl] = ok ? (has_mask ? q.pp[u] * q.mm[u] : q.pp[u]) : 0.f;
        Sc[row * LDP + jl] = ok ? q.ss[u] : 0.f;
      }
    };
    ASTAMP_S(0);
    if (nit > 0) issue(s0, 0);
    if (nit > 1) issue(s1, 1);
    if (nit > 0) store(sm, s0);
    if (nit > 2) issue(s0, 2);
    ASTAMP_S(1);
    __syncthreads();
    for (int c = 0; c < nit; c += 2) {
      ASTAMP_S(2 + c);
      if (c + 1 < nit) {
        store(sm + STG, s1);
        if (c + 3 < nit) issue(s1, c + 3);
      }
      ASTAMP_S(3 + c);
      __syncthreads();
      if (c + 1 >= nit) break;
      if (c + 2 < nit) {
        store(sm, s0);
        if (c + 4 < nit) issue(s0, c + 4);
      }
      __syncthreads();
    }
  } else {
    // =============================== matrix waves ================================================
#pragma unroll
    for (int kt = 0; kt < KG; ++kt)
#pragma unroll
      for (int t = 0; t < MAX_CT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kt][t][r] = 0.f;
    __syncthreads();
    ASTAMP(1);
    for (int c = 0; c < nit; ++c) {
      ASTAMP(2 + c);
      const float* dOs = sm + (c & 1) * STG;
      const float* Qs = dOs + DQC * g.LDH;
      const float* Pc = Qs + DQC * g.LDH;
      const float* Sc = Pc + DQC * LDP;
#pragma unroll
      for (int mm = 0; mm < DQC; mm += 2) {
        float pa[KG], sa[KG];
#pragma unroll
        for (int kt = 0; kt < KG; ++kt) {
          pa[kt] = Pc[(mm + hh) * LDP + kt * 32 + l31];
          sa[kt] = Sc[(mm + hh) * LDP + kt * 32 + l31];
        }
#pragma unroll
        for (int t = 0; t < MAX_CT; ++t) {
          const int ct = wave + 4 * t;
          if (ct >= nct) continue;
          const float b1 = dOs[(mm + hh) * g.LDH + ct * 32 + l31];
          const float b2 = Qs[(mm + hh) * g.LDH + ct * 32 + l31];
#pragma unroll
          for (int kt = 0; kt < KG; ++kt) {
            acc[kt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[kt], b1, acc[kt][t], 0, 0, 0);
            acc[kt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[kt], b2, acc[kt][t], 0, 0, 0);
          }
        }
      }
      __syncthreads();
    }
  }

  ASTAMP(30);
  // ---- result tile R[32*KG][LDH] (aliases the stage buffers) + 32 partial-sum slots behind it ----
  float* R = sm;
  float* Pp = sm + 32 * KG * g.LDH;                  // [32 slots][2][HP]
  if (wave_u < 4) {
#pragma unroll
    for (int kt = 0; kt < KG; ++kt)
#pragma unroll
      for (int t = 0; t < MAX_CT; ++t) {
        const int ct = wave + 4 * t;
        if (ct >= nct) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          R[(kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * g.LDH + ct * 32 + l31] = acc[kt][t][r];
      }
  }
  __syncthreads();
  {
    float4 pg[KCB], pb[KCB];
#pragma unroll
    for (int k = 0; k < KCB; ++k) { pg[k] = f4zero(); pb[k] = f4zero(); }
#pragma unroll
    for (int p = 0; p < KG; ++p) {
      const int lr = wave * 4 * KG + p * 4 + (lane >> 4), j = j0 + lr;
      if (j >= Nk) continue;
      const size_t krow = ((size_t)j * a.Bk + bk) * H;
#pragma unroll
      for (int k = 0; k < KCB; ++k) {
        const int c = q16 * 4 + 64 * k;
        if (c >= H) continue;
        const float4 d = ld4(R + lr * g.LDH + c), xh = kh[p][k];
        pg[k].x += d.x * xh.x; pg[k].y += d.y * xh.y; pg[k].z += d.z * xh.z; pg[k].w += d.w * xh.w;
        pb[k] = f4add(pb[k], d);
        st4(a.dkvhat + krow + c, make_float4(d.x * g0[k].x + dk0[p][k].x, d.y * g0[k].y + dk0[p][k].y,
                                             d.z * g0[k].z + dk0[p][k].z, d.w * g0[k].w + dk0[p][k].w));
      }
    }
    const int slot = wave * 4 + (lane >> 4);
#pragma unroll
    for (int k = 0; k < KCB; ++k) {
      const int c = q16 * 4 + 64 * k;
      if (c >= g.HP) continue;
      st4(Pp + slot * 2 * g.HP + c, pg[k]);
      st4(Pp + slot * 2 * g.HP + g.HP + c, pb[k]);
    }
  }
  __syncthreads();
  // partial rows are indexed by 32-key tile (the caller sizes them so): this group's sums go to its first tile,
  // its other tiles get zeros
  const int nkt = (Nk + 31) / 32;
  for (int kt = 0; kt < KG; ++kt) {
    const int tile = blockIdx.x * KG + kt;
    if (tile >= nkt) break;
    float* prow = a.partials_kv + ((size_t)bk * nkt + tile) * 2 * H;
    for (int c = tid; c < 2 * H; c += 512) {
      float t = 0.f;
      if (kt == 0) {
        const int o = (c / H) * g.HP + (c % H);
#pragma unroll
        for (int sl = 0; sl < 32; ++sl) t += Pp[sl * 2 * g.HP + o];
      }
      prow[c] = t;
    }
  }
}

// Sum of the partial key gradients + the key-side chain rule (see attn_bwd_dq_stream_kernel<.., PKV>):
//   d[j]       = sum_{i < Bq/Bk} sum_{tile} part[bk + i*Bk][tile][j][:]          (fixed order: deterministic)
//   dkvhat[j] (+)= d[j] * gamma0 ;  partial dgamma0 += d[j] * k̂[j] ;  partial dbeta0 += d[j]
// One workgroup per (16 key rows, crystal), one quarter wave per key row, the partials of a row fetched 4 at a time.
// (First version: one workgroup per crystal walking its rows 16 at a time - 7.3 us avg at cfg2, 67 us for the 37 MB of
// partials of an eDOS cross-attention layer on 64 workgroups.)
// One group of DKR key rows of crystal bk, by 256 threads (tid256); Pp: [16][512] floats of LDS; ngroups = ceil(Nk / DKR).
// SC1: the partials were published by OTHER workgroups of this launch (fused form: the last arriving dq workgroup of a
// crystal runs this) - every load of them is then an sc1 load (bypasses this CU's L1).  Contains a barrier.
template <bool SC1>
__device__ __forceinline__ void dkv_reduce_group(const DosxAttn& a, const int nqt, const int grp, const int ngroups,
                                                 const int bk, float (*Pp)[2 * 256], const int tid) {
  const int lane = tid & 63, wave = tid >> 6, q16 = lane & 15, slot = wave * 4 + (lane >> 4);
  const int H = a.H, Nk = a.Nk, rep = a.Bq / a.Bk;
  const int j = grp * DKR + slot;
  const bool jv = j < Nk;
  const int jc = jv ? j : 0;
  const size_t krow = ((size_t)jc * a.Bk + bk) * H;
  float4 g0[KCB], d[KCB], kh[KCB], d0[KCB];
#pragma unroll
  for (int k = 0; k < KCB; ++k) {
    const int c = q16 * 4 + 64 * k, cc = c < H ? c : 0;
    g0[k] = ld4(a.gamma0 + cc);
    kh[k] = ld4(a.kvhat + krow + cc);
    d0[k] = a.dkv_accumulate ? ld4(a.dkvhat + krow + cc) : f4zero();
    d[k] = f4zero();
  }
  const int np = rep * nqt;                          // partials of this key row, in (i, tile) order
  const size_t pstride = (size_t)Nk * H;             // between consecutive tiles of one query batch entry
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)a.dkv_part, 0, 0x7fffffff, 0x00020000);
  for (int p0 = 0; p0 < np; p0 += 4) {
    float4 v[4][KCB];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pi = min(p0 + u, np - 1), i = pi / nqt, t = pi % nqt;
      const size_t off = ((size_t)(bk + i * a.Bk) * nqt + t) * pstride + (size_t)jc * H;
#pragma unroll
      for (int k = 0; k < KCB; ++k) {
        const int c = q16 * 4 + 64 * k;
        if constexpr (SC1)
          v[u][k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rP, (uint32_t)((off + (c < H ? c : 0)) * 4), 0, 16));
        else
          v[u][k] = ld4(a.dkv_part + off + (c < H ? c : 0));
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (p0 + u < np) {
#pragma unroll
        for (int k = 0; k < KCB; ++k) d[k] = f4add(d[k], v[u][k]);
      }
  }
#pragma unroll
  for (int k = 0; k < KCB; ++k) {
    const int c = q16 * 4 + 64 * k;
    float4 pg = f4zero(), pb = f4zero();
    if (jv && c < H) {
      pg = make_float4(d[k].x * kh[k].x, d[k].y * kh[k].y, d[k].z * kh[k].z, d[k].w * kh[k].w);
      pb = d[k];
      st4(a.dkvhat + krow + c, make_float4(d[k].x * g0[k].x + d0[k].x, d[k].y * g0[k].y + d0[k].y,
                                           d[k].z * g0[k].z + d0[k].z, d[k].w * g0[k].w + d0[k].w));
    }
    if (c < 256) {
      st4(&Pp[slot][c], pg);
      st4(&Pp[slot][256 + c], pb);
    }
  }
  __syncthreads();
  float* prow = a.partials_kv + ((size_t)bk * ngroups + grp) * 2 * H;
  for (int c = tid; c < 2 * H; c += 256) {
    const int o = (c / H) * 256 + (c % H);
    float t = 0.f;
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) t += Pp[sl][o];
    prow[c] = t;
  }
}

__global__ __launch_bounds__(256) void attn_dkv_reduce_kernel(const DosxAttn a, int nqt) {
  __shared__ float Pp[16][2 * 256];                 // per quarter-wave slot: [dgamma | dbeta] (H <= 256)
  dkv_reduce_group<false>(a, nqt, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y, Pp, (int)threadIdx.x);
}

size_t fwd_smem(const Geo& g, bool resident = false) {      // (resident: all key chunks at once + the four partial score tiles)
  const int ch = 2 * chunk_buf_floats(g.NKP, g.LDH);
  return sizeof(float) * (size_t)(QT * g.LDH + QT * g.LDS_ + (resident ? max(ch, resident_k_floats(g)) + 4 * KCH : ch));
}
inline bool fwd_resident(const DosxAttn& a) { return a.Nk <= 64 && fwd_smem(make_geo(a.H, a.Nk), true) <= 160 * 1024; }
size_t dq_smem(const Geo& g, bool pkv = false, bool resident = false) {    // (the [16][2][HP] column partial sums reuse the chunk buffers)
  size_t fl = (size_t)QT * g.LDH + (size_t)QT * g.LDS_ +
              (size_t)max(max(2 * chunk_buf_floats(g.NKP, g.LDH), 32 * g.HP), resident ? resident_k_floats(g) : 0);
  if (pkv) fl += (size_t)QT * g.LDS_ + (size_t)QT * g.LDH + 64;      // P tile, raw dO rows (the query rows take over Ds)
  if (resident) fl += 4 * KCH;                                         // partial dP tiles of the resident-key path
  return sizeof(float) * fl;
}
// the partial-dK/dV path needs its tiles in LDS; the resident-key path on top of it only where both still fit
inline bool pkv_fits(int H, int Nk) { return Nk <= 64 && dq_smem(make_geo(H, Nk), true, false) <= 160 * 1024; }
inline bool dq_resident(int H, int Nk, bool pkv) { return Nk <= 64 && dq_smem(make_geo(H, Nk), pkv, true) <= 160 * 1024; }
inline bool pkv_ok(const DosxAttn& a) { return a.dkv_part != nullptr && pkv_fits(a.H, a.Nk); }
size_t dkv_smem(const Geo& g, int kg) {
  const size_t stage = 2 * (size_t)(2 * DQC * g.LDH + 2 * DQC * (32 * kg + 4));
  const size_t epi = (size_t)32 * kg * g.LDH + 32 * 2 * g.HP;
  return sizeof(float) * (stage > epi ? stage : epi);
}

inline int attn_nj3() {               // DOSX_ATTN_NJ3=0: the 33-48-key shapes on the 64-entry row phases again (A/B)
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("DOSX_ATTN_NJ3");
    v = (e && atoi(e) == 0) ? 4 : 3;
  }
  return v;
}

constexpr int MAX_FUSED_NK = 320;      // the score row of a query lives in LDS

int check_attn(const DosxAttn& a, const char* who) {
  DOSX_CHECK_ARG(a.H > 0 && (a.H & 3) == 0 && a.H <= 32 * 4 * MAX_CT, "%s: H=%d unsupported (multiple of 4, <= 256)", who, a.H);
  DOSX_CHECK_ARG(a.Nk > 0, "%s: Nk=%d", who, a.Nk);       // (more than MAX_FUSED_NK keys: attention_general.hip)
  DOSX_CHECK_ARG(a.Sq > 0 && a.Bq > 0 && a.Bk > 0 && a.Bq % a.Bk == 0, "%s: bad Sq/Bq/Bk = %d/%d/%d", who, a.Sq, a.Bq, a.Bk);
  DOSX_CHECK_ARG(a.x && a.kvhat && a.gamma0 && a.beta0 && a.probs, "%s: null operand", who);
  DOSX_CHECK_ARG(a.qstats || (a.flags & DOSX_ATTN_RAW_Q), "%s: qstats required unless RAW_Q", who);
  return 0;
}

}  // namespace

namespace dosx_detail {            // attention_general.hip: the same contract for any number of keys
int attn_general_fwd(const DosxAttn& a, hipStream_t st);
int attn_general_bwd(const DosxAttn& a, hipStream_t st);
// attention_aligned.hip: <= 64 keys on crystal-aligned query tiles (1: launched, 0: not this shape, < 0: error)
int attn_aligned_fwd(const DosxAttn& a, hipStream_t st);
int attn_aligned_bwd(const DosxAttn& a, hipStream_t st);
}  // namespace dosx_detail

extern "C" int dosx_attention_pkv_supported(int Nk, int H) {
  return Nk > 0 && H > 0 && H <= 256 && pkv_fits(H, Nk);
}

extern "C" int dosx_attention_fwd(const DosxAttn* ap, dosx_stream_t stream) {
  DOSX_CHECK_ARG(ap, "dosx_attention_fwd: null");
  const DosxAttn& a = *ap;
  if (int rc = check_attn(a, "dosx_attention_fwd")) return rc;
  DOSX_CHECK_ARG(a.out, "dosx_attention_fwd: null out");
  DOSX_CHECK_ARG(!a.ln1_out || (a.out_stats && a.ln1_gamma && a.ln1_beta && a.Nk <= MAX_FUSED_NK && !(a.flags & DOSX_ATTN_NO_RESIDUAL)),
                 "dosx_attention_fwd: ln1_out needs out_stats, ln1_gamma / ln1_beta, <= %d keys and the residual", MAX_FUSED_NK);
  if (const int rc = dosx_detail::attn_aligned_fwd(a, to_stream(stream))) return rc < 0 ? rc : 0;
  DOSX_CHECK_ARG(!a.key_ptr || a.Nk <= MAX_FUSED_NK, "dosx_attention_fwd: key_ptr (per-crystal key counts) needs <= %d keys", MAX_FUSED_NK);
  if (a.Nk > MAX_FUSED_NK) return dosx_detail::attn_general_fwd(a, to_stream(stream));
  const Geo g = make_geo(a.H, a.Nk);
  const bool res = fwd_resident(a);
  const size_t smem = fwd_smem(g, res);
  DOSX_CHECK_ARG(smem <= 160 * 1024, "dosx_attention_fwd: LDS need %zu > 160 KiB", smem);
  const dim3 grid(ceil_div(a.Sq, QT), a.Bq);
  // (NJ = keys per lane of the row phases, in units of 16.  3: 33-48 keys - the 41-key Electron-DOS cross attention - runs
  //  the softmax phases on 48 instead of 64 entries per row; the MFMA tiles stay 32 keys wide)
  const int nj = a.Nk <= 16 ? 1 : (a.Nk <= 48 ? attn_nj3() : (a.Nk <= 64 ? 4 : (a.Nk <= 208 ? 13 : 20)));
#define DOSX_FWDS(NJ_, RES_)                                                                                \
  do {                                                                                                      \
    static bool attr_set = false;                                                                           \
    if (!attr_set) {                                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_stream_kernel<NJ_, RES_>),          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                    \
      attr_set = true;                                                                                      \
    }                                                                                                       \
    hipLaunchKernelGGL((attn_fwd_stream_kernel<NJ_, RES_>), grid, dim3(512), smem, to_stream(stream), a);   \
  } while (0)
  if (nj == 1) { if (res) DOSX_FWDS(1, true); else DOSX_FWDS(1, false); }
  else if (nj == 3) { if (res) DOSX_FWDS(3, true); else DOSX_FWDS(3, false); }
  else if (nj == 4) { if (res) DOSX_FWDS(4, true); else DOSX_FWDS(4, false); }
  else if (nj == 13) DOSX_FWDS(13, false); else DOSX_FWDS(20, false);
#undef DOSX_FWDS
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_attention_bwd(const DosxAttn* ap, dosx_stream_t stream) {
  DOSX_CHECK_ARG(ap, "dosx_attention_bwd: null");
  const DosxAttn& a = *ap;
  if (int rc = check_attn(a, "dosx_attention_bwd")) return rc;
  DOSX_CHECK_ARG(a.dout && a.dx && (a.dscores || pkv_ok(a)) && a.dkvhat && a.partials_q && a.partials_kv,
                 "dosx_attention_bwd: null operand");
  if (const int rc = dosx_detail::attn_aligned_bwd(a, to_stream(stream))) return rc < 0 ? rc : 0;
  DOSX_CHECK_ARG(!a.key_ptr, "dosx_attention_bwd: key_ptr (per-crystal key counts) is a forward-only mode");
  if (a.Nk > MAX_FUSED_NK) return dosx_detail::attn_general_bwd(a, to_stream(stream));
  const Geo g = make_geo(a.H, a.Nk);
  const int kg = a.Nk > 32 ? 2 : 1;
  const bool pkv = pkv_ok(a);
  const bool res = dq_resident(a.H, a.Nk, pkv);
  const bool fused = pkv && a.dkv_cnt != nullptr;     // dK + dV finished inside the dq launch by the last arriver of a crystal
  DOSX_CHECK_ARG(!fused || !(a.flags & (DOSX_ATTN_BWD_DKV_HALF | DOSX_ATTN_BWD_DQ_HALF)),
                 "dosx_attention_bwd: dkv_cnt (one-launch backward) excludes the DKV_HALF / DQ_HALF flags");
  DOSX_CHECK_ARG(!fused || (size_t)a.Bq * ceil_div(a.Sq, QT) * a.Nk * a.H * 4 < 0x7fffffffull, "dosx_attention_bwd: dkv_part beyond 2 GiB");
  size_t s1 = dq_smem(g, pkv, res);
  const size_t s2 = dkv_smem(g, kg);
  if (fused && s1 < sizeof(float) * 16 * 512 + 64) s1 = sizeof(float) * 16 * 512 + 64;
  DOSX_CHECK_ARG(s1 <= 160 * 1024 && s2 <= 160 * 1024, "dosx_attention_bwd: LDS need %zu/%zu > 160 KiB", s1, s2);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const dim3 grid(ceil_div(a.Sq, QT), a.Bq);
  if (!(a.flags & DOSX_ATTN_BWD_DKV_HALF)) {
    const int nj = a.Nk <= 16 ? 1 : (a.Nk <= 48 ? attn_nj3() : (a.Nk <= 64 ? 4 : (a.Nk <= 208 ? 13 : 20)));
#define DOSX_DQS(NJ_, PKV_, RES_)                                                                           \
  do {                                                                                                      \
    static bool attr_dq = false;                                                                            \
    if (!attr_dq) {                                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_stream_kernel<NJ_, PKV_, RES_>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                    \
      attr_dq = true;                                                                                       \
    }                                                                                                       \
    hipLaunchKernelGGL((attn_bwd_dq_stream_kernel<NJ_, PKV_, RES_>), grid, dim3(512), s1, to_stream(stream), a); \
  } while (0)
#define DOSX_DQS2(NJ_, PKV_) do { if (res) DOSX_DQS(NJ_, PKV_, true); else DOSX_DQS(NJ_, PKV_, false); } while (0)
    if (pkv) { if (nj == 1) DOSX_DQS2(1, true); else if (nj == 3) DOSX_DQS2(3, true); else DOSX_DQS2(4, true); }
    else if (nj == 1) DOSX_DQS2(1, false); else if (nj == 3) DOSX_DQS2(3, false); else if (nj == 4) DOSX_DQS2(4, false); else if (nj == 13) DOSX_DQS(13, false, false);
    else DOSX_DQS(20, false, false);
#undef DOSX_DQS2
#undef DOSX_DQS
    DOSX_LAUNCH_CHECK();
  }
  if (fused) {
    // nothing else to launch
  } else if (!(a.flags & DOSX_ATTN_BWD_DQ_HALF) && pkv) {
    hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3(ceil_div(a.Nk, DKR), a.Bk), dim3(256), 0, to_stream(stream), a,
                       ceil_div(a.Sq, QT));
    DOSX_LAUNCH_CHECK();
  } else if (!(a.flags & DOSX_ATTN_BWD_DQ_HALF)) {
    if (kg == 2) hipLaunchKernelGGL((attn_bwd_dkv_kernel<2>), dim3(ceil_div(a.Nk, 64), a.Bk), dim3(512), s2, to_stream(stream), a);
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<1>), dim3(ceil_div(a.Nk, 32), a.Bk), dim3(512), s2, to_stream(stream), a);
    DOSX_LAUNCH_CHECK();
  }
  return 0;
}

#ifdef DOSX_STAMPS
extern "C" int dosx_debug_read_attn_stamps(unsigned long long* host64) {
  return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(dosx_attn_stamp_buf), sizeof(unsigned long long) * 64);
}
#endif
