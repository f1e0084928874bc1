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

// K chunk for the NT products: Ks[j][0..31] = kvhat[(j*Bk+bk)][kc..kc+31]*gamma+beta, j < Nk else 0
__device__ __forceinline__ void stage_k_chunk(float* Ks, const float* __restrict__ kvhat, const float* __restrict__ gamma,
                                              const float* __restrict__ beta, int Nk, int NKP, int Bk, int bk, int H,
                                              int kc, int tid) {
  const int kq = (tid & 7) * 4, k = kc + kq;
  const int kk = k < H ? k : 0;
  const float4 g = ld4(gamma + kk), b = ld4(beta + kk);
  for (int j = tid >> 3; j < NKP; j += 32) {
    const float4 h = ld4(kvhat + ((size_t)min(j, Nk - 1) * Bk + bk) * H + kk);
    float4 v = make_float4(h.x * g.x + b.x, h.y * g.y + b.y, h.z * g.z + b.z, h.w * g.w + b.w);
    if (!(j < Nk && k < H)) v = f4zero();
    st4(Ks + j * LDK + kq, v);
  }
}

// V chunk for the NN products: Vs[jj][0..HP) = K rows j0..j0+31 (affine), zero beyond Nk / H
__device__ __forceinline__ void stage_v_chunk(float* Vs, const float* __restrict__ kvhat, const float* __restrict__ gamma,
                                              const float* __restrict__ beta, int Nk, int Bk, int bk, int H, int HP,
                                              int LDH, int j0, int tid) {
  const int jj = tid >> 3, j = j0 + jj;
  const float* row = kvhat + ((size_t)min(j, Nk - 1) * Bk + bk) * H;
  for (int c = (tid & 7) * 4; c < HP; c += 32) {
    const int cc = c < H ? c : 0;
    const float4 h = ld4(row + cc), g = ld4(gamma + cc), b = ld4(beta + cc);
    float4 v = make_float4(h.x * g.x + b.x, h.y * g.y + b.y, h.z * g.z + b.z, h.w * g.w + b.w);
    if (!(j < Nk && c < H)) v = f4zero();
    st4(Vs + jj * LDH + c, v);
  }
}

// Whole key set of one crystal into LDS: Ks[j][0..HP) = kvhat[(j*Bk+bk)]*gamma+beta (zero beyond Nk / H).
// Used by the key-resident kernels: ONE global-load phase, then QK^T and PV both read this tile.
__device__ __forceinline__ void stage_k_full(float* Ks, const float* __restrict__ kvhat, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, int Nk, int NKP, int Bk, int bk, int H, int HP,
                                             int LDH, int tid) {
  for (int c = (tid & 7) * 4; c < HP; c += 32) {
    const int cc = c < H ? c : 0;
    const float4 g = ld4(gamma + cc), b = ld4(beta + cc);
    for (int j = tid >> 3; j < NKP; j += 32) {
      const float4 h = ld4(kvhat + ((size_t)min(j, Nk - 1) * Bk + bk) * H + cc);
      float4 v = make_float4(h.x * g.x + b.x, h.y * g.y + b.y, h.z * g.z + b.z, h.w * g.w + b.w);
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
    for (int k = kbeg; k < kend; k += 8) {
      const float4 a = ld4(As + l31 * LDH + k + 4 * hh);
#pragma unroll
      for (int t = 0; t < MAX_KT; ++t) {
        const int jt = q.jt0 + q.jstep * t;
        if (jt >= nkt) continue;
        const float4 b = ld4(Ks + (jt * 32 + l31) * LDH + k + 4 * hh);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[t], 0, 0, 0);
      }
    }
    return;
  }
  for (int kc = 0; kc < HP; kc += KC) {
    stage_k_chunk(Ks, kvhat, gamma, beta, Nk, NKP, Bk, bk, H, kc, tid);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KC; kk += 8) {
      const float4 a = ld4(As + l31 * LDH + kc + kk + 4 * hh);
#pragma unroll
      for (int t = 0; t < MAX_KT; ++t) {
        const int jt = q.jt0 + q.jstep * t;
        if (jt >= nkt) continue;
        const float4 b = ld4(Ks + (jt * 32 + l31) * LDK + kk + 4 * hh);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[t], 0, 0, 0);
      }
    }
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
  for (int j0 = 0; j0 < NKP; j0 += KC) {
    if (!KRES) {
      stage_v_chunk(Vs, kvhat, gamma, beta, Nk, Bk, bk, H, HP, LDH, j0, tid);
      __syncthreads();
    }
#pragma unroll
    for (int kk = 0; kk < KC; kk += 8) {
      const float4 a = ld4(Ps + l31 * LDS_ + j0 + kk + 4 * hh);
#pragma unroll
      for (int t = 0; t < MAX_CT; ++t) {
        const int ct = wave + 4 * t;
        if (ct >= nct) continue;
        const float* bp = Vs + ((KRES ? j0 : 0) + kk + 4 * hh) * LDH + ct * 32 + l31;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bp[0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bp[LDH], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bp[2 * LDH], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bp[3 * LDH], acc[t], 0, 0, 0);
      }
    }
    if (!KRES) __syncthreads();
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

constexpr int RW = 8;      // query rows per wave in the row phases (32-row tile / 4 waves)
constexpr int MAXJ = 5;    // keys per lane in the softmax phases (Nk <= 320)

// ================================== forward =====================================================
// Row phases (LayerNorm of the queries, softmax, residual + statistics) keep a wave's 8 rows in
// REGISTERS (lane l owns columns 4l..4l+3, H <= 256) and run the 8 reductions as independent chains:
// a row-at-a-time loop through LDS cost ~8 exposed latency chains per phase and a global round trip
// per row for the residual (the raw query rows are simply kept from the first load).
template <bool KRES>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const DosxAttn a) {
  extern __shared__ __align__(16) float sm[];
  const Geo g = make_geo(a.H, a.Nk);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const QkSplit q = make_split(g.NKP, g.HP, KRES, wave);
  float* Qs = sm;                                   // [32][LDH]   (later: output tile)
  float* Ss = Qs + QT * g.LDH;                      // [ks][32][LDS_] partial scores; tile 0 becomes P
  float* KV = Ss + q.ks * QT * g.LDS_;              // max(NKP*36, 32*LDH) or the whole key tile
  const int s0 = blockIdx.x * QT, bq = blockIdx.y, bk = bq % a.Bk;
  const int H = a.H, Sq = a.Sq, Nk = a.Nk;
  const bool raw_q = (a.flags & DOSX_ATTN_RAW_Q) != 0;
  const int c0 = lane * 4;
  const bool con = c0 < H;                          // this lane owns 4 real columns
  const float invH = 1.f / (float)H;

  // ---- query rows: global -> registers (kept for the residual) -> LayerNorm -> LDS ----
  float4 xr[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int s = min(s0 + wave * RW + i, Sq - 1);
    xr[i] = ld4(a.x + ((size_t)s * a.q_stride_s + (size_t)bq * a.q_stride_b) * H + (con ? c0 : 0));
  }
  const float4 g0 = ld4(a.gamma0 + (con ? c0 : 0)), b0 = ld4(a.beta0 + (con ? c0 : 0));
  if (KRES) stage_k_full(KV, a.kvhat, a.gamma0, a.beta0, Nk, g.NKP, a.Bk, bk, H, g.HP, g.LDH, tid);
  {
    float mean[RW], rstd[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const float4 v = xr[i];
      mean[i] = con ? (v.x + v.y) + (v.z + v.w) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) mean[i] = wave_sum(mean[i]) * invH;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const float4 v = xr[i];
      const float p0 = v.x - mean[i], p1 = v.y - mean[i], p2 = v.z - mean[i], p3 = v.w - mean[i];
      rstd[i] = con ? (p0 * p0 + p1 * p1) + (p2 * p2 + p3 * p3) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) rstd[i] = rsqrtf(wave_sum(rstd[i]) * invH + DOSX_LN_EPS);
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int lr = wave * RW + i, s = s0 + lr;
      const float4 v = xr[i];
      float4 o = v;
      if (!raw_q)
        o = make_float4((v.x - mean[i]) * rstd[i] * g0.x + b0.x, (v.y - mean[i]) * rstd[i] * g0.y + b0.y,
                        (v.z - mean[i]) * rstd[i] * g0.z + b0.z, (v.w - mean[i]) * rstd[i] * g0.w + b0.w);
      if (!(con && s < Sq)) o = f4zero();
      if (c0 < g.HP) st4(Qs + lr * g.LDH + c0, o);
      if (!raw_q && a.qstats && lane == 0 && s < Sq) {
        a.qstats[2 * ((size_t)s * a.Bq + bq)] = mean[i];
        a.qstats[2 * ((size_t)s * a.Bq + bq) + 1] = rstd[i];
      }
    }
  }
  __syncthreads();

  f32x16 sacc[MAX_KT];
  qk_product<KRES>(sacc, Qs, g.LDH, KV, a.kvhat, a.gamma0, a.beta0, Nk, g.NKP, a.Bk, bk, H, g.HP, tid, q);
  store_scores(sacc, Ss, g.LDS_, g.NKP, tid, q);
  __syncthreads();

  // ---- exact fp32 softmax over the Nk keys (padded atoms included, like the reference) ----
  {
    const float scale = rsqrtf((float)H);
    const int pstride = QT * g.LDS_;
    float v[RW][MAXJ], mx[RW], sum[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const float* row = Ss + (wave * RW + i) * g.LDS_;
      mx[i] = -INFINITY;
#pragma unroll
      for (int jj = 0; jj < MAXJ; ++jj) {
        const int j = lane + 64 * jj;
        float t = -INFINITY;
        if (j < Nk) {
          t = row[j];
          if (q.ks > 1) t += row[pstride + j];
          if (q.ks > 2) t += row[2 * pstride + j] + row[3 * pstride + j];
          t *= scale;
        }
        v[i][jj] = t;
        mx[i] = fmaxf(mx[i], t);
      }
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) mx[i] = wave_max(mx[i]);
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      sum[i] = 0.f;
#pragma unroll
      for (int jj = 0; jj < MAXJ; ++jj) {
        const float e = (lane + 64 * jj) < Nk ? expf(v[i][jj] - mx[i]) : 0.f;
        v[i][jj] = e;
        sum[i] += e;
      }
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) sum[i] = 1.f / wave_sum(sum[i]);
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int lr = wave * RW + i, s = s0 + lr;
      float* row = Ss + lr * g.LDS_;
#pragma unroll
      for (int jj = 0; jj < MAXJ; ++jj) {
        const int j = lane + 64 * jj;
        if (j >= g.NKP) continue;
        const float pr = v[i][jj] * sum[i];           // 0 beyond Nk
        row[j] = pr;
        if (j < Nk && s < Sq) a.probs[((size_t)bq * Sq + s) * Nk + j] = pr;
      }
    }
  }
  __syncthreads();

  f32x16 oacc[MAX_CT];
  pv_product<KRES>(oacc, Ss, g.LDS_, KV, a.kvhat, a.gamma0, a.beta0, Nk, g.NKP, a.Bk, bk, H, g.HP, g.LDH, tid);
  store_out_tile(oacc, Qs, g.LDH, g.HP, tid);
  __syncthreads();

  // ---- epilogue: residual add from the kept query rows, statistics of the output rows (feeds LN1) ----
  {
    const bool no_res = (a.flags & DOSX_ATTN_NO_RESIDUAL) != 0;
    float4 o[RW];
    float mean[RW], var[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int lr = wave * RW + i, s = s0 + lr;
      float4 v = con ? ld4(Qs + lr * g.LDH + c0) : f4zero();
      if (!no_res) v = f4add(v, xr[i]);
      o[i] = v;
      if (con && s < Sq) st4(a.out + ((size_t)s * a.Bq + bq) * H + c0, v);
      mean[i] = con ? (v.x + v.y) + (v.z + v.w) : 0.f;
    }
    if (a.out_stats) {
#pragma unroll
      for (int i = 0; i < RW; ++i) mean[i] = wave_sum(mean[i]) * invH;
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        const float p0 = o[i].x - mean[i], p1 = o[i].y - mean[i], p2 = o[i].z - mean[i], p3 = o[i].w - mean[i];
        var[i] = con ? (p0 * p0 + p1 * p1) + (p2 * p2 + p3 * p3) : 0.f;
      }
#pragma unroll
      for (int i = 0; i < RW; ++i) var[i] = wave_sum(var[i]) * invH;
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        const int s = s0 + wave * RW + i;
        if (lane == 0 && s < Sq) {
          a.out_stats[2 * ((size_t)s * a.Bq + bq)] = mean[i];
          a.out_stats[2 * ((size_t)s * a.Bq + bq) + 1] = rsqrtf(var[i] + DOSX_LN_EPS);
        }
      }
    }
  }
}

// ================================== backward: dq / dx ============================================
template <bool KRES>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const DosxAttn a) {
  extern __shared__ __align__(16) float sm[];
  const Geo g = make_geo(a.H, a.Nk);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const QkSplit q = make_split(g.NKP, g.HP, KRES, wave);
  float* Ds = sm;                                   // [32][LDH]  dOut tile, later dq tile
  float* Ss = Ds + QT * g.LDH;                      // [ks][32][LDS_] partial dP; tile 0 becomes dS
  float* KV = Ss + q.ks * QT * g.LDS_;              // chunk staging, or the whole key tile (KRES)
  float* Pp = KV + (KRES ? g.NKP * g.LDH : max(g.NKP * LDK, KC * g.LDH));    // [4][2][HP] partial column sums
  const int s0 = blockIdx.x * QT, bq = blockIdx.y, bk = bq % a.Bk;
  const int H = a.H, Sq = a.Sq, Nk = a.Nk;
  const bool raw_q = (a.flags & DOSX_ATTN_RAW_Q) != 0, no_res = (a.flags & DOSX_ATTN_NO_RESIDUAL) != 0;
  const int c0 = lane * 4;
  const bool con = c0 < H;
  const float invH = 1.f / (float)H;

  // ---- this wave's 8 rows of dOut (kept for the residual) and of x, their statistics, their P row ----
  float4 go[RW], xr[RW];
  float mean[RW], rstd[RW], pr[RW][MAXJ];
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int s = min(s0 + wave * RW + i, Sq - 1);
    const size_t orow = (size_t)s * a.Bq + bq;
    go[i] = ld4(a.dout + orow * H + (con ? c0 : 0));
    xr[i] = ld4(a.x + ((size_t)s * a.q_stride_s + (size_t)bq * a.q_stride_b) * H + (con ? c0 : 0));
    mean[i] = raw_q ? 0.f : a.qstats[2 * orow];
    rstd[i] = raw_q ? 1.f : a.qstats[2 * orow + 1];
#pragma unroll
    for (int jj = 0; jj < MAXJ; ++jj) {
      const int j = lane + 64 * jj;
      pr[i][jj] = a.probs[((size_t)bq * Sq + s) * Nk + (j < Nk ? j : 0)];
    }
  }
  const float4 g0 = ld4(a.gamma0 + (con ? c0 : 0));
  if (KRES) stage_k_full(KV, a.kvhat, a.gamma0, a.beta0, Nk, g.NKP, a.Bk, bk, H, g.HP, g.LDH, tid);
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int lr = wave * RW + i;
    float4 d = go[i];
    if (!(con && (s0 + lr) < Sq)) d = f4zero();
    if (c0 < g.HP) st4(Ds + lr * g.LDH + c0, d);
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
    float dp[RW][MAXJ], dot[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const float* row = Ss + (wave * RW + i) * g.LDS_;
      dot[i] = 0.f;
#pragma unroll
      for (int jj = 0; jj < MAXJ; ++jj) {
        const int j = lane + 64 * jj;
        float t = 0.f;
        if (j < Nk) {
          t = row[j];
          if (q.ks > 1) t += row[pstride + j];
          if (q.ks > 2) t += row[2 * pstride + j] + row[3 * pstride + j];
          dot[i] += pr[i][jj] * t;
        }
        dp[i][jj] = t;
      }
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) dot[i] = wave_sum(dot[i]);
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int lr = wave * RW + i, s = s0 + lr;
      float* row = Ss + lr * g.LDS_;
#pragma unroll
      for (int jj = 0; jj < MAXJ; ++jj) {
        const int j = lane + 64 * jj;
        if (j >= g.NKP) continue;
        const float ds = (j < Nk && s < Sq) ? pr[i][jj] * (dp[i][jj] - dot[i]) * scale : 0.f;
        row[j] = ds;
        if (j < Nk && s < Sq) a.dscores[((size_t)bq * Sq + s) * Nk + j] = ds;
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
    float4 pg = f4zero(), pb = f4zero();
    float4 d[RW], xh[RW];
    float s1[RW], s2[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int lr = wave * RW + i;
      const bool rv = con && (s0 + lr) < Sq;
      d[i] = rv ? ld4(Ds + lr * g.LDH + c0) : f4zero();
      const float4 xv = xr[i];
      xh[i] = make_float4((xv.x - mean[i]) * rstd[i], (xv.y - mean[i]) * rstd[i], (xv.z - mean[i]) * rstd[i],
                          (xv.w - mean[i]) * rstd[i]);
      if (!rv) xh[i] = f4zero();
      pg.x += d[i].x * xh[i].x; pg.y += d[i].y * xh[i].y; pg.z += d[i].z * xh[i].z; pg.w += d[i].w * xh[i].w;
      pb = f4add(pb, d[i]);
      const float4 dh = make_float4(d[i].x * g0.x, d[i].y * g0.y, d[i].z * g0.z, d[i].w * g0.w);
      s1[i] = (dh.x + dh.y) + (dh.z + dh.w);
      s2[i] = (dh.x * xh[i].x + dh.y * xh[i].y) + (dh.z * xh[i].z + dh.w * xh[i].w);
    }
    if (!raw_q) {
#pragma unroll
      for (int i = 0; i < RW; ++i) { s1[i] = wave_sum(s1[i]) * invH; s2[i] = wave_sum(s2[i]) * invH; }
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int s = s0 + wave * RW + i;
      if (!(con && s < Sq)) continue;
      float4 o;
      if (raw_q) o = d[i];
      else
        o = make_float4(rstd[i] * (d[i].x * g0.x - s1[i] - xh[i].x * s2[i]), rstd[i] * (d[i].y * g0.y - s1[i] - xh[i].y * s2[i]),
                        rstd[i] * (d[i].z * g0.z - s1[i] - xh[i].z * s2[i]), rstd[i] * (d[i].w * g0.w - s1[i] - xh[i].w * s2[i]));
      if (!no_res) o = f4add(o, go[i]);
      st4(a.dx + ((size_t)s * a.Bq + bq) * H + c0, o);
    }
    if (raw_q) { pg = f4zero(); pb = f4zero(); }
    if (c0 < g.HP) {
      st4(Pp + wave * 2 * g.HP + c0, pg);
      st4(Pp + wave * 2 * g.HP + g.HP + c0, pb);
    }
  }
  __syncthreads();
  float* prow = a.partials_q + ((size_t)bq * gridDim.x + blockIdx.x) * 2 * H;
  for (int c = tid; c < 2 * H; c += 256) {
    const int which = c / H, col = c % H;
    const int o = which * g.HP + col;
    prow[c] = Pp[o] + Pp[2 * g.HP + o] + Pp[4 * g.HP + o] + Pp[6 * g.HP + o];
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
                                  (kres ? g.NKP * g.LDH : max(g.NKP * LDK, KC * g.LDH)));
}
size_t dq_smem(const Geo& g, bool kres) {
  return sizeof(float) * (size_t)(QT * g.LDH + qk_ks(g.NKP, g.HP, kres) * QT * g.LDS_ +
                                  (kres ? g.NKP * g.LDH : max(g.NKP * LDK, KC * g.LDH)) + 8 * g.HP);
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
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const dim3 grid(ceil_div(a.Sq, QT), a.Bq);
  if (kres) hipLaunchKernelGGL((attn_fwd_kernel<true>), grid, dim3(256), smem, to_stream(stream), a);
  else hipLaunchKernelGGL((attn_fwd_kernel<false>), grid, dim3(256), smem, to_stream(stream), a);
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
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const dim3 grid(ceil_div(a.Sq, QT), a.Bq);
  if (!(a.flags & DOSX_ATTN_BWD_SKIP_DQ)) {
    if (kres) hipLaunchKernelGGL((attn_bwd_dq_kernel<true>), grid, dim3(256), s1, to_stream(stream), a);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<false>), grid, dim3(256), s1, to_stream(stream), a);
    DOSX_LAUNCH_CHECK();
  }
  if (!(a.flags & DOSX_ATTN_BWD_SKIP_DKV)) {
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(ceil_div(a.Nk, 32), a.Bk), dim3(256), s2, to_stream(stream), a);
    DOSX_LAUNCH_CHECK();
  }
  return 0;
}
