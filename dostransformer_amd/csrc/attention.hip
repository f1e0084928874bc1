// Pre-norm energy-vs-atom attention block of the reference encoder layer (gfx950, fp32 MFMA).
//
//   x1 = x + softmax_fp32( LN0(x) K^T / sqrt(H) ) K ,   K = V = kvhat * gamma0 + beta0
//
// (layers/transformer.py:131-138 + layers/multihead_attention.py:68-74: no q/k/v/out projection,
//  no mask, no head split; zero-padded atoms are real keys with value beta0 — SURVEY.md §0.2-0.3.)
// One workgroup (4 waves) owns a 32-query tile of one crystal; the whole key set of a crystal
// (Nk <= 320: atoms <= Nmax, or the 51/201 energy bins for self attention) lives in LDS, so the
// softmax is exact (no online rescaling).  QK^T and PV run on v_mfma_f32_32x32x2_f32.
#include "common.h"

// Diagnostic build only (-DDOSX_STAMPS): wave 0 of workgroup (0,0) records s_memtime at phase boundaries.
#ifdef DOSX_STAMPS
extern "C" { __device__ unsigned long long dosx_attn_stamp_buf[64]; }
#define ASTAMP(slot)                                                                     \
  do {                                                                                   \
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0)                          \
      dosx_attn_stamp_buf[(slot)] = __builtin_amdgcn_s_memtime();                        \
  } while (0)
#else
#define ASTAMP(slot) do { } while (0)
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

// Load 32 rows x H (float4) of a row-strided matrix into LDS [32][LDH], zero padded.
// row pointer for tile row i: base + rowoff(i) (nullptr -> zeros).  Optional affine LN on load:
//   v = (v - mean)*rstd*gamma + beta   (stats == nullptr: plain copy)
template <class RowPtr>
__device__ __forceinline__ void load_rows(float* dst, int LDH, int HP, int H, RowPtr rowptr, int tid) {
  const int row = tid >> 3;
  const float* p = rowptr(row);
  const bool ok = p != nullptr;
  const float* q = ok ? p : rowptr(0);          // tile row 0 is always a valid row
  for (int c = (tid & 7) * 4; c < HP; c += 32) {
    float4 v = ld4(q + (c < H ? c : 0));        // unconditional load, masked afterwards (no branch)
    if (!(ok && c < H)) v = f4zero();
    st4(dst + row * LDH + c, v);
  }
}

// Streamed (non-resident) key tiles.  A chunk is fetched into registers one chunk ahead (the loads fly
// under the MFMAs of the current chunk) and written, with the key affine applied, into the OTHER of two
// LDS chunk buffers: one barrier per chunk.
constexpr int MAXKR = 10;   // 32-key row blocks per crystal (Nk <= 320)
constexpr int MAXVC = 8;    // 32-column blocks per row (H <= 256)

// K chunk for the NT products: Ks[j][0..31] = kvhat[(j*Bk+bk)][kc..kc+31]*gamma+beta, j < Nk else 0
struct KChunk { float4 h[MAXKR], g, b; };
__device__ __forceinline__ void k_chunk_load(KChunk& r, const float* __restrict__ kvhat, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, int Nk, int NKP, int Bk, int bk, int H,
                                             int kc, int tid) {
  const int k = kc + (tid & 7) * 4, kk = k < H ? k : 0;
  r.g = ld4(gamma + kk);
  r.b = ld4(beta + kk);
#pragma unroll
  for (int i = 0; i < MAXKR; ++i) {
    const int j = (tid >> 3) + 32 * i;
    if (j < NKP) r.h[i] = ld4(kvhat + ((size_t)min(j, Nk - 1) * Bk + bk) * H + kk);
  }
}
__device__ __forceinline__ void k_chunk_store(float* Ks, const KChunk& r, int Nk, int NKP, int H, int kc, int tid) {
  const int kq = (tid & 7) * 4, k = kc + kq;
#pragma unroll
  for (int i = 0; i < MAXKR; ++i) {
    const int j = (tid >> 3) + 32 * i;
    if (j >= NKP) break;
    const float4 h = r.h[i];
    float4 v = make_float4(h.x * r.g.x + r.b.x, h.y * r.g.y + r.b.y, h.z * r.g.z + r.b.z, h.w * r.g.w + r.b.w);
    if (!(j < Nk && k < H)) v = f4zero();
    st4(Ks + j * LDK + kq, v);
  }
}

// V chunk for the NN products: Vs[jj][0..HP) = K rows j0..j0+31 (affine), zero beyond Nk / H
struct VChunk { float4 h[MAXVC]; };
__device__ __forceinline__ void v_chunk_load(VChunk& r, const float* __restrict__ kvhat, int Nk, int Bk, int bk, int H,
                                             int HP, int j0, int tid) {
  const int j = j0 + (tid >> 3);
  const float* row = kvhat + ((size_t)min(j, Nk - 1) * Bk + bk) * H;
#pragma unroll
  for (int i = 0; i < MAXVC; ++i) {
    const int c = (tid & 7) * 4 + 32 * i;
    if (c < HP) r.h[i] = ld4(row + (c < H ? c : 0));
  }
}
__device__ __forceinline__ void v_chunk_store(float* Vs, const VChunk& r, const float4 (&g)[MAXVC], const float4 (&b)[MAXVC],
                                              int Nk, int H, int HP, int LDH, int j0, int tid) {
  const int jj = tid >> 3, j = j0 + jj;
#pragma unroll
  for (int i = 0; i < MAXVC; ++i) {
    const int c = (tid & 7) * 4 + 32 * i;
    if (c >= HP) break;
    const float4 h = r.h[i];
    float4 v = make_float4(h.x * g[i].x + b[i].x, h.y * g[i].y + b[i].y, h.z * g[i].z + b[i].z, h.w * g[i].w + b[i].w);
    if (!(j < Nk && c < H)) v = f4zero();
    st4(Vs + jj * LDH + c, v);
  }
}

// Whole key set of one crystal into LDS: Ks[j][0..HP) = kvhat[(j*Bk+bk)]*gamma+beta (zero beyond Nk / H).
// Used by the key-resident kernels: ONE global-load phase, then QK^T and PV both read this tile.
__device__ __forceinline__ void stage_k_full(float* Ks, const float* __restrict__ kvhat, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, int Nk, int NKP, int Bk, int bk, int H, int HP,
                                             int LDH, int tid) {
  // items: (32-key block jb, 32-column block cb); this thread owns key jb*32 + tid/8, columns cb*32 + (tid%8)*4.
  // Loads are issued in batches of 8 before any is used (a load-use-store loop exposes one global round
  // trip per item: ~0.7 us each, 4..8 of them in front of the first MFMA).
  const int ncb = HP / 32, nit = (NKP / 32) * ncb;
  const int jr = tid >> 3, cq = (tid & 7) * 4;
  for (int i0 = 0; i0 < nit; i0 += 8) {
    float4 h[8], gq[8], bq[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int it = min(i0 + u, nit - 1);
      const int j = (it / ncb) * 32 + jr, c = (it % ncb) * 32 + cq, cc = c < H ? c : 0;
      h[u] = ld4(kvhat + ((size_t)min(j, Nk - 1) * Bk + bk) * H + cc);
      gq[u] = ld4(gamma + cc);
      bq[u] = ld4(beta + cc);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int it = i0 + u;
      if (it >= nit) break;
      const int j = (it / ncb) * 32 + jr, c = (it % ncb) * 32 + cq;
      float4 v = make_float4(h[u].x * gq[u].x + bq[u].x, h[u].y * gq[u].y + bq[u].y, h[u].z * gq[u].z + bq[u].z,
                             h[u].w * gq[u].w + bq[u].w);
      if (!(j < Nk && c < H)) v = f4zero();
      st4(Ks + j * LDH + c, v);
    }
  }
}

// How the QK^T-shaped products are split over the 4 waves.  With >= 3 key tiles every wave owns whole
// 32-key tiles (jt = wave + 4t) and the full feature range.  With 1 or 2 key tiles (the 12-atom /
// 51-bin cases of the phonon configs) that would leave 3 or 2 waves idle for 16 serial MFMA steps, so
// the FEATURE range is split instead: ks = 4 (or 2) partial score tiles, summed when the scores are read.
struct QkSplit {
  int ks;       // partial tiles
  int kpart;    // this wave's partial
  int jt0;      // this wave's first key tile
  int jstep;    // key-tile stride (4 when ks == 1: tiles wave, wave+4, ...; else no second tile)
};
__host__ __device__ inline int qk_ks(int NKP, int HP, bool kres) {
  if (!kres) return 1;
  const int nkt = NKP / 32;
  if (nkt == 1 && HP % 32 == 0) return 4;
  if (nkt == 2 && HP % 16 == 0) return 2;
  return 1;
}
__device__ __forceinline__ QkSplit make_split(int NKP, int HP, bool kres, int wave) {
  QkSplit q;
  q.ks = qk_ks(NKP, HP, kres);
  q.kpart = q.ks == 4 ? wave : (q.ks == 2 ? (wave >> 1) : 0);
  q.jt0 = q.ks == 4 ? 0 : (q.ks == 2 ? (wave & 1) : wave);
  q.jstep = q.ks == 1 ? 4 : 1024;
  return q;
}

// S[32][NKP] (+)= A[32][H] . K^T : A tile resident in LDS (As, stride LDH), K streamed in k-chunks
// (or resident: KRES).  Partial products of wave-split `q` stay in acc; store_scores() writes them to
// the partial tile q.kpart.
template <bool KRES>
__device__ __forceinline__ void qk_product(f32x16 (&acc)[MAX_KT], const float* As, int LDH, float* Ks,
                                           const float* kvhat, const float* gamma, const float* beta, int Nk, int NKP,
                                           int Bk, int bk, int H, int HP, int tid, const QkSplit q) {
  const int lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int nkt = NKP / 32;
#pragma unroll
  for (int t = 0; t < MAX_KT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int kbeg = q.kpart * (HP / q.ks), kend = kbeg + HP / q.ks;
  if (KRES) {
    // software-pipelined: the fragments of step k+8 are read from LDS while the MFMAs of step k run
    // (the loop bounds are run-time values, hipcc does not pipeline it by itself: ~200 clk of exposed
    // ds_read latency per 8-12 MFMAs otherwise)
    const float* ap = As + l31 * LDH + 4 * hh;
    const float* bp[MAX_KT];
    bool on[MAX_KT];
#pragma unroll
    for (int t = 0; t < MAX_KT; ++t) {
      const int jt = q.jt0 + q.jstep * t;
      on[t] = jt < nkt;
      bp[t] = Ks + ((on[t] ? jt : 0) * 32 + l31) * LDH + 4 * hh;
    }
    float4 an = ld4(ap + kbeg), bn[MAX_KT];
#pragma unroll
    for (int t = 0; t < MAX_KT; ++t) bn[t] = on[t] ? ld4(bp[t] + kbeg) : f4zero();
    for (int k = kbeg; k < kend; k += 8) {
      const float4 a = an;
      float4 b[MAX_KT];
#pragma unroll
      for (int t = 0; t < MAX_KT; ++t) b[t] = bn[t];
      const int kn = (k + 8 < kend) ? k + 8 : k;
      an = ld4(ap + kn);
#pragma unroll
      for (int t = 0; t < MAX_KT; ++t)
        if (on[t]) bn[t] = ld4(bp[t] + kn);
#pragma unroll
      for (int t = 0; t < MAX_KT; ++t) {
        if (!on[t]) continue;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[t].x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[t].y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[t].z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[t].w, acc[t], 0, 0, 0);
      }
    }
    return;
  }
  // streamed K: two chunk buffers (Ks, Ks + CHB), register prefetch one chunk ahead, one barrier per chunk
  const int CHB = chunk_buf_floats(NKP, LDH);
  KChunk kr;
  k_chunk_load(kr, kvhat, gamma, beta, Nk, NKP, Bk, bk, H, 0, tid);
  k_chunk_store(Ks, kr, Nk, NKP, H, 0, tid);
  __syncthreads();
  for (int kc = 0, ib = 0; kc < HP; kc += KC, ib ^= 1) {
    const float* Kc = Ks + ib * CHB;
    const bool more = kc + KC < HP;
    if (more) k_chunk_load(kr, kvhat, gamma, beta, Nk, NKP, Bk, bk, H, kc + KC, tid);
#pragma unroll
    for (int kk = 0; kk < KC; kk += 8) {
      const float4 a = ld4(As + l31 * LDH + kc + kk + 4 * hh);
#pragma unroll
      for (int t = 0; t < MAX_KT; ++t) {
        const int jt = q.jt0 + q.jstep * t;
        if (jt >= nkt) continue;
        const float4 b = ld4(Kc + (jt * 32 + l31) * LDK + kk + 4 * hh);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[t], 0, 0, 0);
      }
    }
    if (more) k_chunk_store(Ks + (ib ^ 1) * CHB, kr, Nk, NKP, H, kc + KC, tid);
    __syncthreads();
  }
}

// store the (partial) QK^T accumulators into partial tile q.kpart of Ss ([ks][32][LDS_])
__device__ __forceinline__ void store_scores(const f32x16 (&acc)[MAX_KT], float* Ss, int LDS_, int NKP, int tid,
                                             const QkSplit q) {
  const int lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int nkt = NKP / 32;
  float* Sp = Ss + q.kpart * QT * LDS_;
#pragma unroll
  for (int t = 0; t < MAX_KT; ++t) {
    const int jt = q.jt0 + q.jstep * t;
    if (jt >= nkt) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
      Sp[row * LDS_ + jt * 32 + l31] = acc[t][r];
    }
  }
}

// O[32][HP] = P[32][NKP] . V : P resident in LDS (Ps, stride LDS_), V streamed in 32-key chunks.
template <bool KRES>
__device__ __forceinline__ void pv_product(f32x16 (&acc)[MAX_CT], const float* Ps, int LDS_, float* Vs,
                                           const float* kvhat, const float* gamma, const float* beta, int Nk, int NKP,
                                           int Bk, int bk, int H, int HP, int LDH, int tid) {
  const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int nct = HP / 32;
#pragma unroll
  for (int t = 0; t < MAX_CT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  if (KRES) {
    // resident V (= K) tile: software-pipelined like qk_product
    const float* ap = Ps + l31 * LDS_ + 4 * hh;
    const float* vp[MAX_CT];
    bool on[MAX_CT];
#pragma unroll
    for (int t = 0; t < MAX_CT; ++t) {
      const int ct = wave + 4 * t;
      on[t] = ct < nct;
      vp[t] = Vs + (4 * hh) * LDH + (on[t] ? ct : 0) * 32 + l31;
    }
    float4 an = ld4(ap);
    float bn[MAX_CT][4];
#pragma unroll
    for (int t = 0; t < MAX_CT; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c) bn[t][c] = on[t] ? vp[t][c * LDH] : 0.f;
    for (int j = 0; j < NKP; j += 8) {
      const float4 a = an;
      float b[MAX_CT][4];
#pragma unroll
      for (int t = 0; t < MAX_CT; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) b[t][c] = bn[t][c];
      const int jn = (j + 8 < NKP) ? j + 8 : j;
      an = ld4(ap + jn);
#pragma unroll
      for (int t = 0; t < MAX_CT; ++t)
        if (on[t]) {
#pragma unroll
          for (int c = 0; c < 4; ++c) bn[t][c] = vp[t][(jn + c) * LDH];
        }
#pragma unroll
      for (int t = 0; t < MAX_CT; ++t) {
        if (!on[t]) continue;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[t][0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[t][1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[t][2], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[t][3], acc[t], 0, 0, 0);
      }
    }
    return;
  }
  const int CHB = chunk_buf_floats(NKP, LDH);
  float4 gq[MAXVC], bq[MAXVC];
#pragma unroll
  for (int i = 0; i < MAXVC; ++i) {
    const int c = (tid & 7) * 4 + 32 * i, cc = c < H ? c : 0;
    gq[i] = ld4(gamma + cc);
    bq[i] = ld4(beta + cc);
  }
  VChunk vr;
  v_chunk_load(vr, kvhat, Nk, Bk, bk, H, HP, 0, tid);
  v_chunk_store(Vs, vr, gq, bq, Nk, H, HP, LDH, 0, tid);
  __syncthreads();
  for (int j0 = 0, ib = 0; j0 < NKP; j0 += KC, ib ^= 1) {
    const float* Vc = Vs + ib * CHB;
    const bool more = j0 + KC < NKP;
    if (more) v_chunk_load(vr, kvhat, Nk, Bk, bk, H, HP, j0 + KC, tid);
#pragma unroll
    for (int kk = 0; kk < KC; kk += 8) {
      const float4 a = ld4(Ps + l31 * LDS_ + j0 + kk + 4 * hh);
#pragma unroll
      for (int t = 0; t < MAX_CT; ++t) {
        const int ct = wave + 4 * t;
        if (ct >= nct) continue;
        const float* bp = Vc + (kk + 4 * hh) * LDH + ct * 32 + l31;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bp[0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bp[LDH], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bp[2 * LDH], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bp[3 * LDH], acc[t], 0, 0, 0);
      }
    }
    if (more) v_chunk_store(Vs + (ib ^ 1) * CHB, vr, gq, bq, Nk, H, HP, LDH, j0 + KC, tid);
    __syncthreads();
  }
}

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

// ================================== forward =====================================================
// NJ = ceil(Nk / 16): keys per lane in the softmax (compile time: 1 / 4 / 20 cover Nk <= 16 / 64 / 320).
template <bool KRES, int NJ>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const DosxAttn a) {
  extern __shared__ __align__(16) float sm[];
  const Geo g = make_geo(a.H, a.Nk);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q16 = lane & 15;
  const QkSplit q = make_split(g.NKP, g.HP, KRES, wave);
  float* Qs = sm;                                   // [32][LDH]   (later: output tile)
  float* Ss = Qs + QT * g.LDH;                      // [ks][32][LDS_] partial scores; tile 0 becomes P
  float* KV = Ss + q.ks * QT * g.LDS_;              // max(NKP*36, 32*LDH) or the whole key tile
  const int s0 = blockIdx.x * QT, bq = blockIdx.y, bk = bq % a.Bk;
  const int H = a.H, Sq = a.Sq, Nk = a.Nk;
  const bool raw_q = (a.flags & DOSX_ATTN_RAW_Q) != 0;
  const float invH = 1.f / (float)H;
  ASTAMP(0);

  // ---- query rows: global -> registers (kept for the residual) -> LayerNorm -> LDS ----
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
  if (KRES) stage_k_full(KV, a.kvhat, a.gamma0, a.beta0, Nk, g.NKP, a.Bk, bk, H, g.HP, g.LDH, tid);
  ASTAMP(1);
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
        if (!(c < H && s < Sq)) o = f4zero();
        st4(Qs + lr * g.LDH + c, o);
      }
      if (!raw_q && a.qstats && q16 == 0 && s < Sq) {
        a.qstats[2 * ((size_t)s * a.Bq + bq)] = mean[p];
        a.qstats[2 * ((size_t)s * a.Bq + bq) + 1] = rstd[p];
      }
    }
  }
  ASTAMP(2);
  __syncthreads();
  ASTAMP(3);

  f32x16 sacc[MAX_KT];
  qk_product<KRES>(sacc, Qs, g.LDH, KV, a.kvhat, a.gamma0, a.beta0, Nk, g.NKP, a.Bk, bk, H, g.HP, tid, q);
  store_scores(sacc, Ss, g.LDS_, g.NKP, tid, q);
  ASTAMP(4);
  __syncthreads();
  ASTAMP(5);

  // ---- exact fp32 softmax over the Nk keys (padded atoms included, like the reference) ----
  {
    const float scale = rsqrtf((float)H);
    const int pstride = QT * g.LDS_;
    float v[RP][NJ], mx[RP], sum[RP];
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      const float* row = Ss + row_of(wave, p, lane) * g.LDS_;
      mx[p] = -INFINITY;
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const int j = q16 + 16 * jj;
        float t = -INFINITY;
        if (j < Nk) {
          t = row[j];
          if (q.ks > 1) t += row[pstride + j];
          if (q.ks > 2) t += row[2 * pstride + j] + row[3 * pstride + j];
          t *= scale;
        }
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
        const float e = (q16 + 16 * jj) < Nk ? expf(v[p][jj] - mx[p]) : 0.f;
        v[p][jj] = e;
        sum[p] += e;
      }
    }
#pragma unroll
    for (int p = 0; p < RP; ++p) sum[p] = 1.f / row16_sum(sum[p]);
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      const int lr = row_of(wave, p, lane), s = s0 + lr;
      float* row = Ss + lr * g.LDS_;
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const int j = q16 + 16 * jj;
        if (j >= g.NKP) continue;
        const float pr = v[p][jj] * sum[p];           // 0 beyond Nk
        row[j] = pr;
        if (j < Nk && s < Sq) a.probs[((size_t)bq * Sq + s) * Nk + j] = pr;
      }
      if (NJ * 16 < g.NKP) {                          // (NJ*16 >= Nk always; zero the rest of the padded tile)
        for (int j = NJ * 16 + q16; j < g.NKP; j += 16) row[j] = 0.f;
      }
    }
  }
  ASTAMP(6);
  __syncthreads();
  ASTAMP(7);

  f32x16 oacc[MAX_CT];
  pv_product<KRES>(oacc, Ss, g.LDS_, KV, a.kvhat, a.gamma0, a.beta0, Nk, g.NKP, a.Bk, bk, H, g.HP, g.LDH, tid);
  store_out_tile(oacc, Qs, g.LDH, g.HP, tid);
  ASTAMP(8);
  __syncthreads();
  ASTAMP(9);

  // ---- epilogue: residual add from the kept query rows, statistics of the output rows (feeds LN1) ----
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
          v = ld4(Qs + lr * g.LDH + c);
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
    }
  }
  ASTAMP(10);
}

// ================================== backward: dq / dx ============================================
template <bool KRES, int NJ>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const DosxAttn a) {
  extern __shared__ __align__(16) float sm[];
  const Geo g = make_geo(a.H, a.Nk);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q16 = lane & 15;
  const QkSplit q = make_split(g.NKP, g.HP, KRES, wave);
  float* Ds = sm;                                   // [32][LDH]  dOut tile, later dq tile
  float* Ss = Ds + QT * g.LDH;                      // [ks][32][LDS_] partial dP; tile 0 becomes dS
  float* KV = Ss + q.ks * QT * g.LDS_;              // chunk staging, or the whole key tile (KRES)
  // [16][2][HP] partial column sums: behind the resident key tile, or (streamed keys) ON the chunk buffers,
  // which are dead after the dS.K product
  float* Pp = KRES ? KV + g.NKP * g.LDH : KV;
  const int s0 = blockIdx.x * QT, bq = blockIdx.y, bk = bq % a.Bk;
  const int H = a.H, Sq = a.Sq, Nk = a.Nk;
  const bool raw_q = (a.flags & DOSX_ATTN_RAW_Q) != 0, no_res = (a.flags & DOSX_ATTN_NO_RESIDUAL) != 0;
  const float invH = 1.f / (float)H;

  // ---- this wave's rows of dOut (kept for the residual) and of x, their statistics, their P row ----
  float4 go[RP][KCB], xr[RP][KCB], g0[KCB];
  float mean[RP], rstd[RP], pr[RP][NJ];
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
#pragma unroll
  for (int k = 0; k < KCB; ++k) g0[k] = ld4(a.gamma0 + ((q16 * 4 + 64 * k) < H ? (q16 * 4 + 64 * k) : 0));
  if (KRES) stage_k_full(KV, a.kvhat, a.gamma0, a.beta0, Nk, g.NKP, a.Bk, bk, H, g.HP, g.LDH, tid);
#pragma unroll
  for (int p = 0; p < RP; ++p) {
    const int lr = row_of(wave, p, lane);
#pragma unroll
    for (int k = 0; k < KCB; ++k) {
      const int c = q16 * 4 + 64 * k;
      if (c >= g.HP) continue;
      float4 d = go[p][k];
      if (!(c < H && (s0 + lr) < Sq)) d = f4zero();
      st4(Ds + lr * g.LDH + c, d);
    }
  }
  __syncthreads();

  // dP = dO . V^T
  f32x16 sacc[MAX_KT];
  qk_product<KRES>(sacc, Ds, g.LDH, KV, a.kvhat, a.gamma0, a.beta0, Nk, g.NKP, a.Bk, bk, H, g.HP, tid, q);
  store_scores(sacc, Ss, g.LDS_, g.NKP, tid, q);
  __syncthreads();

  // dS = P * (dP - rowsum(P*dP)) * scale
  {
    const float scale = rsqrtf((float)H);
    const int pstride = QT * g.LDS_;
    float dp[RP][NJ], dot[RP];
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      const float* row = Ss + row_of(wave, p, lane) * g.LDS_;
      dot[p] = 0.f;
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const int j = q16 + 16 * jj;
        float t = 0.f;
        if (j < Nk) {
          t = row[j];
          if (q.ks > 1) t += row[pstride + j];
          if (q.ks > 2) t += row[2 * pstride + j] + row[3 * pstride + j];
          dot[p] += pr[p][jj] * t;
        }
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
        if (j < Nk && s < Sq) a.dscores[((size_t)bq * Sq + s) * Nk + j] = ds;
      }
      if (NJ * 16 < g.NKP) {
        for (int j = NJ * 16 + q16; j < g.NKP; j += 16) row[j] = 0.f;
      }
    }
  }
  __syncthreads();

  // dq_ln = dS . K
  f32x16 oacc[MAX_CT];
  pv_product<KRES>(oacc, Ss, g.LDS_, KV, a.kvhat, a.gamma0, a.beta0, Nk, g.NKP, a.Bk, bk, H, g.HP, g.LDH, tid);
  store_out_tile(oacc, Ds, g.LDH, g.HP, tid);
  __syncthreads();

  // LN0 backward on the query rows + residual;  partial dgamma0 / dbeta0 (query side)
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
        const float4 dd = rv ? ld4(Ds + lr * g.LDH + c) : f4zero();
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
    // column partial sums: one slot per quarter wave (16 slots), summed in a fixed order below
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
}

// ================================== backward: dk + dv ============================================
// One workgroup per (32-key tile, crystal): dkv_ln[keys,H] = sum_q P^T dO + dS^T LN0(x) over every
// query row of every query batch entry that maps to this crystal (bq = bk + i*Bk).
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const DosxAttn a) {
  extern __shared__ __align__(16) float sm[];
  const Geo g = make_geo(a.H, a.Nk);
  float* dOs = sm;                                  // [32 q][LDH]
  float* Qs = dOs + QT * g.LDH;                     // [32 q][LDH]  (later: result tile)
  float* Pc = Qs + QT * g.LDH;                      // [32 q][36]   P   chunk (cols = keys of this tile)
  float* Sc = Pc + QT * LDK;                        // [32 q][36]   dS  chunk
  float* Pp = Sc + QT * LDK;                        // [4][2][HP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int j0 = blockIdx.x * 32, bk = blockIdx.y;
  const int H = a.H, Sq = a.Sq, Nk = a.Nk;
  const int nct = g.HP / 32;
  const bool raw_q = (a.flags & DOSX_ATTN_RAW_Q) != 0;
  constexpr int MAXC = 4 * MAX_CT;                  // column groups of 32 per row (H <= 256)

  f32x16 acc[MAX_CT];
#pragma unroll
  for (int t = 0; t < MAX_CT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // staging registers of the NEXT (query batch entry, 32-query chunk): loads are issued one iteration
  // ahead (before the MFMA block) from clamped addresses and masked / normalised when stored to LDS
  const int ri = tid >> 3, cg = tid & 7;
  float4 rdo[MAXC], rx[MAXC], gq[MAXC], bt[MAXC];
  float rp[4], rs[4], rmean = 0.f, rrstd = 1.f;
  bool rok = false;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int col = cg * 4 + 32 * c, cc = col < H ? col : 0;
    gq[c] = ld4(a.gamma0 + cc);
    bt[c] = ld4(a.beta0 + cc);
  }
  const int nchunks = (Sq + QT - 1) / QT, nit = (a.Bq / a.Bk) * nchunks;
  auto issue = [&](int it) {
    const int bq = bk + (it / nchunks) * a.Bk, s = (it % nchunks) * QT + ri;
    rok = s < Sq;
    const int sc = rok ? s : Sq - 1;
    const float* dop = a.dout + ((size_t)sc * a.Bq + bq) * H;
    const float* xp = a.x + ((size_t)sc * a.q_stride_s + (size_t)bq * a.q_stride_b) * H;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c < nct) {
        const int col = cg * 4 + 32 * c, cc = col < H ? col : 0;
        rdo[c] = ld4(dop + cc);
        rx[c] = ld4(xp + cc);
      }
    }
    const size_t prow = ((size_t)bq * Sq + sc) * Nk;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int jc = min(j0 + cg + 8 * jj, Nk - 1);
      rp[jj] = a.probs[prow + jc];
      rs[jj] = a.dscores[prow + jc];
    }
    if (!raw_q) {
      rmean = a.qstats[2 * ((size_t)sc * a.Bq + bq)];
      rrstd = a.qstats[2 * ((size_t)sc * a.Bq + bq) + 1];
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c < nct) {
        const int col = cg * 4 + 32 * c;
        const bool valid = rok && col < H;
        float4 d = rdo[c], q = rx[c];
        if (!raw_q)
          q = make_float4((q.x - rmean) * rrstd * gq[c].x + bt[c].x, (q.y - rmean) * rrstd * gq[c].y + bt[c].y,
                          (q.z - rmean) * rrstd * gq[c].z + bt[c].z, (q.w - rmean) * rrstd * gq[c].w + bt[c].w);
        if (!valid) { d = f4zero(); q = f4zero(); }
        st4(dOs + ri * g.LDH + col, d);
        st4(Qs + ri * g.LDH + col, q);
      }
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const bool valid = rok && (j0 + cg + 8 * jj) < Nk;
      Pc[ri * LDK + cg + 8 * jj] = valid ? rp[jj] : 0.f;
      Sc[ri * LDK + cg + 8 * jj] = valid ? rs[jj] : 0.f;
    }
  };

  issue(0);
  for (int it = 0; it < nit; ++it) {
    store();
    __syncthreads();
    if (it + 1 < nit) issue(it + 1);
#pragma unroll
    for (int mm = 0; mm < QT; mm += 2) {
      const float pa = Pc[(mm + hh) * LDK + l31];
      const float sa = Sc[(mm + hh) * LDK + l31];
#pragma unroll
      for (int t = 0; t < MAX_CT; ++t) {
        const int ct = wave + 4 * t;
        if (ct >= nct) continue;
        const float b1 = dOs[(mm + hh) * g.LDH + ct * 32 + l31];
        const float b2 = Qs[(mm + hh) * g.LDH + ct * 32 + l31];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa, b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa, b2, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  store_out_tile(acc, Qs, g.LDH, g.HP, tid);
  for (int c = lane; c < 2 * g.HP; c += 64) Pp[wave * 2 * g.HP + c] = 0.f;
  __syncthreads();
  for (int i = 0; i < 8; ++i) {
    const int lr = wave * 8 + i, j = j0 + lr;
    if (j >= Nk) break;
    const size_t krow = ((size_t)j * a.Bk + bk) * H;
    for (int c = lane * 4; c < H; c += 256) {
      const float4 d = ld4(Qs + lr * g.LDH + c), xh = ld4(a.kvhat + krow + c), gm = ld4(a.gamma0 + c);
      float* pgm = Pp + wave * 2 * g.HP + c;
      pgm[0] += d.x * xh.x; pgm[1] += d.y * xh.y; pgm[2] += d.z * xh.z; pgm[3] += d.w * xh.w;
      float* pbt = pgm + g.HP;
      pbt[0] += d.x; pbt[1] += d.y; pbt[2] += d.z; pbt[3] += d.w;
      float4 o = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
      if (a.dkv_accumulate) o = f4add(o, ld4(a.dkvhat + krow + c));
      st4(a.dkvhat + krow + c, o);
    }
  }
  __syncthreads();
  float* prow = a.partials_kv + ((size_t)bk * gridDim.x + blockIdx.x) * 2 * H;
  for (int c = tid; c < 2 * H; c += 256) {
    const int which = c / H, col = c % H;
    const int o = which * g.HP + col;
    prow[c] = Pp[o] + Pp[2 * g.HP + o] + Pp[4 * g.HP + o] + Pp[6 * g.HP + o];
  }
}

size_t fwd_smem(const Geo& g, bool kres) {
  return sizeof(float) * (size_t)(QT * g.LDH + qk_ks(g.NKP, g.HP, kres) * QT * g.LDS_ +
                                  (kres ? g.NKP * g.LDH : 2 * chunk_buf_floats(g.NKP, g.LDH)));
}
size_t dq_smem(const Geo& g, bool kres) {
  const size_t kv = kres ? (size_t)g.NKP * g.LDH + 32 * g.HP : (size_t)max(2 * chunk_buf_floats(g.NKP, g.LDH), 32 * g.HP);
  return sizeof(float) * ((size_t)QT * g.LDH + (size_t)qk_ks(g.NKP, g.HP, kres) * QT * g.LDS_ + kv);
}
constexpr size_t KRES_LDS_LIMIT = 144 * 1024;   // keep the whole key tile of a crystal in LDS when it fits
size_t dkv_smem(const Geo& g) { return sizeof(float) * (size_t)(2 * QT * g.LDH + 2 * QT * LDK + 8 * g.HP); }

int check_attn(const DosxAttn& a, const char* who) {
  DOSX_CHECK_ARG(a.H > 0 && (a.H & 3) == 0 && a.H <= 32 * 4 * MAX_CT, "%s: H=%d unsupported (multiple of 4, <= 256)", who, a.H);
  DOSX_CHECK_ARG(a.Nk > 0 && a.Nk <= 320, "%s: Nk=%d unsupported (1..320 keys per crystal)", who, a.Nk);
  DOSX_CHECK_ARG(a.Sq > 0 && a.Bq > 0 && a.Bk > 0 && a.Bq % a.Bk == 0, "%s: bad Sq/Bq/Bk = %d/%d/%d", who, a.Sq, a.Bq, a.Bk);
  DOSX_CHECK_ARG(a.x && a.kvhat && a.gamma0 && a.beta0 && a.probs, "%s: null operand", who);
  DOSX_CHECK_ARG(a.qstats || (a.flags & DOSX_ATTN_RAW_Q), "%s: qstats required unless RAW_Q", who);
  return 0;
}

}  // namespace

extern "C" int dosx_attention_fwd(const DosxAttn* ap, dosx_stream_t stream) {
  DOSX_CHECK_ARG(ap, "dosx_attention_fwd: null");
  const DosxAttn& a = *ap;
  if (int rc = check_attn(a, "dosx_attention_fwd")) return rc;
  DOSX_CHECK_ARG(a.out, "dosx_attention_fwd: null out");
  const Geo g = make_geo(a.H, a.Nk);
  const bool kres = fwd_smem(g, true) <= KRES_LDS_LIMIT;
  const size_t smem = fwd_smem(g, kres);
  DOSX_CHECK_ARG(smem <= 160 * 1024, "dosx_attention_fwd: LDS need %zu > 160 KiB", smem);
  const dim3 grid(ceil_div(a.Sq, QT), a.Bq);
  const int nj = a.Nk <= 16 ? 1 : (a.Nk <= 64 ? 4 : (a.Nk <= 208 ? 13 : 20));
#define DOSX_FWD(KR, NJ_)                                                                                   \
  do {                                                                                                      \
    static bool attr_set = false;                                                                           \
    if (!attr_set) {                                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<KR, NJ_>),                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                    \
      attr_set = true;                                                                                      \
    }                                                                                                       \
    hipLaunchKernelGGL((attn_fwd_kernel<KR, NJ_>), grid, dim3(256), smem, to_stream(stream), a);            \
  } while (0)
  if (kres) { if (nj == 1) DOSX_FWD(true, 1); else if (nj == 4) DOSX_FWD(true, 4); else if (nj == 13) DOSX_FWD(true, 13); else DOSX_FWD(true, 20); }
  else { if (nj == 1) DOSX_FWD(false, 1); else if (nj == 4) DOSX_FWD(false, 4); else if (nj == 13) DOSX_FWD(false, 13); else DOSX_FWD(false, 20); }
#undef DOSX_FWD
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_attention_bwd(const DosxAttn* ap, dosx_stream_t stream) {
  DOSX_CHECK_ARG(ap, "dosx_attention_bwd: null");
  const DosxAttn& a = *ap;
  if (int rc = check_attn(a, "dosx_attention_bwd")) return rc;
  DOSX_CHECK_ARG(a.dout && a.dx && a.dscores && a.dkvhat && a.partials_q && a.partials_kv, "dosx_attention_bwd: null operand");
  const Geo g = make_geo(a.H, a.Nk);
  const bool kres = dq_smem(g, true) <= KRES_LDS_LIMIT;
  const size_t s1 = dq_smem(g, kres), s2 = dkv_smem(g);
  DOSX_CHECK_ARG(s1 <= 160 * 1024 && s2 <= 160 * 1024, "dosx_attention_bwd: LDS need %zu/%zu > 160 KiB", s1, s2);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const dim3 grid(ceil_div(a.Sq, QT), a.Bq);
  if (!(a.flags & DOSX_ATTN_BWD_SKIP_DQ)) {
    const int nj = a.Nk <= 16 ? 1 : (a.Nk <= 64 ? 4 : (a.Nk <= 208 ? 13 : 20));
#define DOSX_DQ(KR, NJ_)                                                                                    \
  do {                                                                                                      \
    static bool attr_dq = false;                                                                            \
    if (!attr_dq) {                                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<KR, NJ_>),                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                    \
      attr_dq = true;                                                                                       \
    }                                                                                                       \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<KR, NJ_>), grid, dim3(256), s1, to_stream(stream), a);           \
  } while (0)
    if (kres) { if (nj == 1) DOSX_DQ(true, 1); else if (nj == 4) DOSX_DQ(true, 4); else if (nj == 13) DOSX_DQ(true, 13); else DOSX_DQ(true, 20); }
    else { if (nj == 1) DOSX_DQ(false, 1); else if (nj == 4) DOSX_DQ(false, 4); else if (nj == 13) DOSX_DQ(false, 13); else DOSX_DQ(false, 20); }
#undef DOSX_DQ
    DOSX_LAUNCH_CHECK();
  }
  if (!(a.flags & DOSX_ATTN_BWD_SKIP_DKV)) {
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(ceil_div(a.Nk, 32), a.Bk), dim3(256), s2, to_stream(stream), a);
    DOSX_LAUNCH_CHECK();
  }
  return 0;
}

#ifdef DOSX_STAMPS
extern "C" int dosx_debug_read_attn_stamps(unsigned long long* host64) {
  return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(dosx_attn_stamp_buf), sizeof(unsigned long long) * 64);
}
#endif
