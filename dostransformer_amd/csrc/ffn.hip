// Fused position-wise feed-forward half of the reference encoder layer (layers/transformer.py:141-148):
//
//     out = x + fc2( relu( fc1( LN1(x) ) ) ),        h = relu(fc1(LN1(x))) is also written out (saved for backward)
//
// One workgroup owns 32 rows and runs BOTH GEMMs: the 32 x 4H intermediate tile stays in LDS as the A operand of
// fc2, so the layer costs one launch, one prologue and one row epilogue instead of two of each (the k-loops of the
// two GEMMs at this workload's sizes are ~12 us together; the fixed part of a launch is ~5 us).  Same wave
// specialisation as gemm_kernel: waves 0-3 multiply, waves 4-7 stream the 128x32 weight chunks of fc1 (column block
// by column block) and then of fc2 through two LDS stage buffers, one barrier per chunk, loads two chunks deep;
// the matrix waves write h to HBM straight from the accumulators (round 1: the staging waves copied the finished tile out in one burst).
// Needs H % 32 == 0 and H <= 128 (tile: 32 x 516 floats); larger H uses the two-GEMM path.
#include <stdlib.h>

#include "common.h"
#include "mma16.h"

typedef int v4i32 __attribute__((ext_vector_type(4)));

#ifdef DOSX_STAMPS
__device__ unsigned long long dosx_ffn_stamp_buf[64];
extern "C" int dosx_debug_read_ffn_stamps(unsigned long long* host64) {
  return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(dosx_ffn_stamp_buf), sizeof(unsigned long long) * 64);
}
#define FSTAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x == 0) dosx_ffn_stamp_buf[(slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define FSTAMP_S(slot) do { if (threadIdx.x == 256 && blockIdx.x == 0) dosx_ffn_stamp_buf[32 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FSTAMP(slot) do { } while (0)
#define FSTAMP_S(slot) do { } while (0)
#endif

namespace {

// k-chunk width KB (32, or 64 when H % 64 == 0: half as many barriers - a 32-wide chunk is only 16 MFMAs = 1024 clk
// per wave and each barrier costs ~450 clk of ds_read round trip and skew, tools/stamp_ffn.py) is a template argument
constexpr int FBN = 128;         // columns per chunk / per column block

// One query row of the fused attention prologue (ffn_fwd_kernel<.., ATT = true>), by one QUARTER wave: lane q16 owns the
// float4 columns 4 q16 + 64 k (k < 2: H <= 128).  Writes probs / qstats / x1 / st1 of row m0 + lr and the LN1-normalised row
// into the Xs tile.
__device__ __forceinline__ void ffn_att_row(const DosxFfn& a, float* __restrict__ Xs, const int LDX, const int lr, const int m0,
                                            const int lane) {
  const int H = a.H, M = a.M;
  const int q16 = lane & 15;
  const float scale = rsqrtf((float)H), invH = 1.f / (float)H;
  const int Nk = a.att_Nk, Bq = a.att_Bq, Bk = a.att_Bk, Sq = a.att_Sq;
  float4 g0[2], b0[2], g1[2], bb1[2];
  bool on[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c = q16 * 4 + 64 * k, cc = c < H ? c : 0;
    on[k] = c < H;
    g0[k] = ld4(a.att_gamma0 + cc); b0[k] = ld4(a.att_beta0 + cc);
    g1[k] = ld4(a.gamma + cc); bb1[k] = ld4(a.beta + cc);
  }
  {
    const int r = m0 + lr, rc = min(r, M - 1);
    const bool rv = r < M;
    const int s = rc / Bq, bq = rc - s * Bq, bk = bq % Bk;
    const int nk = a.att_key_ptr ? min(a.att_key_ptr[bk + 1] - a.att_key_ptr[bk], Nk) : Nk;      // keys this crystal attends over (DosxFfn.att_key_ptr)
    const float* xrow = a.x + ((size_t)s * a.att_qs + (size_t)bq * a.att_qb) * a.ldx;
    float4 xr[2];
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      xr[k] = on[k] ? ld4(xrow + q16 * 4 + 64 * k) : f4zero();
      t += (xr[k].x + xr[k].y) + (xr[k].z + xr[k].w);
    }
    // the crystal's key rows (pre-normalised; row j of crystal bk at (j * Bk + bk)): all requested before the first use
    float4 kv[16][2];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float* kr = a.att_kvhat + ((size_t)(j < Nk ? j : 0) * Bk + bk) * H;
#pragma unroll
      for (int k = 0; k < 2; ++k) kv[j][k] = on[k] ? ld4(kr + q16 * 4 + 64 * k) : f4zero();
    }
    const float mean = row16_sum(t) * invH;
    t = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (!on[k]) continue;
      const float p0 = xr[k].x - mean, p1 = xr[k].y - mean, p2 = xr[k].z - mean, p3 = xr[k].w - mean;
      t += (p0 * p0 + p1 * p1) + (p2 * p2 + p3 * p3);
    }
    const float rstd = rsqrtf(row16_sum(t) * invH + DOSX_LN_EPS);
    float4 qg[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float4 v = xr[k];
      float4 q = make_float4((v.x - mean) * rstd * g0[k].x + b0[k].x, (v.y - mean) * rstd * g0[k].y + b0[k].y,
                             (v.z - mean) * rstd * g0[k].z + b0[k].z, (v.w - mean) * rstd * g0[k].w + b0[k].w);
      q = make_float4(q.x * g0[k].x, q.y * g0[k].y, q.z * g0[k].z, q.w * g0[k].w);      // key gamma folded into Q
      qg[k] = on[k] ? q : f4zero();
    }
    float sc[16], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float d = 0.f;
#pragma unroll
      for (int k = 0; k < 2; ++k)
        d += (qg[k].x * kv[j][k].x + qg[k].y * kv[j][k].y) + (qg[k].z * kv[j][k].z + qg[k].w * kv[j][k].w);
      sc[j] = row16_sum(d) * scale;
      if (j < nk) mx = fmaxf(mx, sc[j]);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float e = j < nk ? expf(sc[j] - mx) : 0.f;
      sc[j] = e;
      sum += e;
    }
    const float inv = 1.f / sum;
    const size_t prow = ((size_t)bq * Sq + s) * Nk;
    float psum = 0.f;
    float4 o[2] = {f4zero(), f4zero()};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float pr = sc[j] * inv;                                   // 0 beyond Nk
      const float mk = (a.att_mask && j < Nk) ? a.att_mask[prow + j] : 1.f;
      const float pm = pr * mk;
      psum += pm;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        o[k].x += pm * kv[j][k].x; o[k].y += pm * kv[j][k].y; o[k].z += pm * kv[j][k].z; o[k].w += pm * kv[j][k].w;
      }
      if (rv && q16 == j && j < Nk) a.att_probs[prow + j] = pr;       // the un-dropped P (the backward reads it)
    }
    if (!a.att_mask) psum = 1.f;
    float4 x1[2];
    t = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      x1[k] = f4zero();
      if (!on[k]) continue;
      x1[k] = make_float4(o[k].x * g0[k].x + b0[k].x * psum + xr[k].x, o[k].y * g0[k].y + b0[k].y * psum + xr[k].y,
                          o[k].z * g0[k].z + b0[k].z * psum + xr[k].z, o[k].w * g0[k].w + b0[k].w * psum + xr[k].w);
      if (rv) st4(a.att_x1 + (size_t)r * a.att_ldx1 + q16 * 4 + 64 * k, x1[k]);
      t += (x1[k].x + x1[k].y) + (x1[k].z + x1[k].w);
    }
    const float mean1 = row16_sum(t) * invH;
    t = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (!on[k]) continue;
      const float p0 = x1[k].x - mean1, p1 = x1[k].y - mean1, p2 = x1[k].z - mean1, p3 = x1[k].w - mean1;
      t += (p0 * p0 + p1 * p1) + (p2 * p2 + p3 * p3);
    }
    const float rstd1 = rsqrtf(row16_sum(t) * invH + DOSX_LN_EPS);
    if (rv && q16 == 0) {
      a.att_qstats[2 * (size_t)r] = mean; a.att_qstats[2 * (size_t)r + 1] = rstd;
      a.att_st1[2 * (size_t)r] = mean1;   a.att_st1[2 * (size_t)r + 1] = rstd1;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (!on[k]) continue;
      const float4 v = x1[k];
      st4(Xs + lr * LDX + q16 * 4 + 64 * k,
          make_float4((v.x - mean1) * rstd1 * g1[k].x + bb1[k].x, (v.y - mean1) * rstd1 * g1[k].y + bb1[k].y,
                      (v.z - mean1) * rstd1 * g1[k].z + bb1[k].z, (v.w - mean1) * rstd1 * g1[k].w + bb1[k].w));
    }
  }
}

// The attention half of a CRYSTAL-ALIGNED tile (ATT = 2) on the MFMA (round 5): the R query rows of the tile share one key set
// (Ks [NkP][H + 4] in LDS, rows >= Nk zeroed), so scores and P.K are two small products on v_mfma_f32_16x16x4_f32 instead of one
// key per iteration on the vector ALU (round 4's form: 0.4 us per key and pass, the 51-key self attention 47 us per layer):
//   A: one quarter wave per row - LayerNorm-0, the key gamma folded into q (see ffn_att_row) -> Qs [R][H + 4]
//   B: S = Qs . Ks^T (scaled) -> Sc [R][68]          (job = 16 rows x 16 keys, K = H: H / 4 MFMAs; jobs dealt over the 8 waves)
//   C: one quarter wave per row - exact fp32 softmax over the Nk scores, P written out, P o mask -> Sc (zeros beyond Nk)
//   D: O = Sc . Ks -> Qs                             (job = 16 rows x 16 columns, K = NkP: NkP / 4 MFMAs)
//   E: one quarter wave per row - x1 = O o g0 + b0 sum(P) + x, both statistics, LN1(x1) -> Xs
// All 8 waves run it (R = 32: four rows per wave, one pass; the staging waves have their first weight chunks in flight); four
// workgroup barriers inside, the caller adds the one that frees the key / score region.
template <int R>
__device__ __forceinline__ void ffn_att_tile(const DosxFfn& a, float* __restrict__ Xs, const int LDX, float* __restrict__ Ks,
                                             float* __restrict__ Sc, float* __restrict__ Qs, const int al_s0, const int al_bq,
                                             const int tid) {
  const int H = a.H, LDK = a.H + 4;
  const int lane = tid & 63, wave = tid >> 6, q16 = lane & 15, l15 = lane & 15, g4 = lane >> 4;
  const int Nk = a.att_Nk, NkP = (Nk + 15) & ~15, Sq = a.att_Sq;
  const float scale = rsqrtf((float)H), invH = 1.f / (float)H;
  // this quarter wave's row (R = 16: waves 0-3 only)
  const int lr = wave * 4 + g4;
  const bool rowok = lr < R;
  const int s = min(al_s0 + (rowok ? lr : 0), Sq - 1);
  const bool rv = rowok && (al_s0 + lr) < Sq;
  const int r = s * a.att_Bq + al_bq;                  // global row (valid memory also for the clamped duplicates)
  float4 g0[2], b0[2], xr[2], g1[2], bb1[2];         // (g1 / bb1: LayerNorm-1's affine for phase E, requested with everything else)
  bool on[2];
  float mean = 0.f, rstd = 0.f;
  {
    const float* xrow = a.x + ((size_t)s * a.att_qs + (size_t)al_bq * a.att_qb) * a.ldx;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = q16 * 4 + 64 * k, cc = c < H ? c : 0;
      on[k] = c < H;
      g0[k] = ld4(a.att_gamma0 + cc); b0[k] = ld4(a.att_beta0 + cc);
      g1[k] = ld4(a.gamma + cc); bb1[k] = ld4(a.beta + cc);
      xr[k] = on[k] ? ld4(xrow + c) : f4zero();
    }
    {   // the crystal's key rows -> Ks (all 8 waves; rows beyond Nk zero): requested right behind the row operands above
      const int bk = al_bq % a.att_Bk, h4 = H >> 2;
      const UDiv dh4(h4);
      float4 kr[4];                                  // NkP * h4 <= 64 * 32 = 4 float4 per thread
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = tid + 512 * i, j = dh4.div(e), c = (e - j * h4) * 4;
        kr[i] = (e < NkP * h4 && j < Nk) ? ld4(a.att_kvhat + ((size_t)j * a.att_Bk + bk) * H + c) : f4zero();
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = tid + 512 * i, j = dh4.div(e), c = (e - j * h4) * 4;
        if (e < NkP * h4) st4(Ks + j * LDK + c, kr[i]);
      }
    }
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) t += (xr[k].x + xr[k].y) + (xr[k].z + xr[k].w);
    mean = row16_sum(t) * invH;
    t = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (!on[k]) continue;
      const float p0 = xr[k].x - mean, p1 = xr[k].y - mean, p2 = xr[k].z - mean, p3 = xr[k].w - mean;
      t += (p0 * p0 + p1 * p1) + (p2 * p2 + p3 * p3);
    }
    rstd = rsqrtf(row16_sum(t) * invH + DOSX_LN_EPS);
    if (rowok) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (!on[k]) continue;
        const float4 v = xr[k];
        float4 q = make_float4((v.x - mean) * rstd * g0[k].x + b0[k].x, (v.y - mean) * rstd * g0[k].y + b0[k].y,
                               (v.z - mean) * rstd * g0[k].z + b0[k].z, (v.w - mean) * rstd * g0[k].w + b0[k].w);
        st4(Qs + lr * LDK + q16 * 4 + 64 * k, make_float4(q.x * g0[k].x, q.y * g0[k].y, q.z * g0[k].z, q.w * g0[k].w));   // key gamma folded into Q
      }
    }
  }
  FSTAMP(10);
  __syncthreads();
  FSTAMP(11);
  // ---- B: scores ----
  // (fewer than 8 jobs - <= 16 keys: 2 - would leave six waves idle behind a chain of H / 4 dependent MFMAs: the K range is
  //  split KS ways, every wave leaves a partial tile in Sp [KS][R][68] and phase C adds them in a fixed order)
  const int nctS = NkP >> 4, njobsS = (R / 16) * nctS;
  int KS = njobsS >= 8 ? 1 : (njobsS >= 4 ? 2 : 4);
  while (KS > 1 && ((H % (16 * KS)) != 0 || LDK + (KS - 1) * 68 > 4 * H + 4)) KS >>= 1;   // (whole 16-wide MFMA steps; the partial tiles fit the T region)
  const int klen = H / KS;
  const UDiv dKS(KS);
  {
    for (int unit = wave; unit < njobsS * KS; unit += 8) {
      const int job = dKS.div(unit), kq = unit - job * KS;
      const int rt = job >= nctS ? job / nctS : 0, ct = job - rt * nctS;
      const f32x4 acc = mma_kk<8>(Qs + (16 * rt + l15) * LDK + kq * klen + 4 * g4, Ks + (16 * ct + l15) * LDK + kq * klen + 4 * g4, klen >> 4);
      float* Sp = kq == 0 ? Sc : Qs + R * LDK + (kq - 1) * R * 68;      // (partials 1 .. KS - 1 behind the Q tile, in the T region)
#pragma unroll
      for (int i = 0; i < 4; ++i) Sp[(16 * rt + 4 * g4 + i) * 68 + 16 * ct + l15] = acc[i] * scale;
    }
  }
  FSTAMP(12);
  __syncthreads();
  // ---- C: softmax of this quarter wave's row (over the crystal's own nk <= Nk keys with DosxFfn.att_key_ptr) ----
  float psum = 1.f;
  const int bk_ = al_bq % a.att_Bk;
  const int nk = a.att_key_ptr ? min(a.att_key_ptr[bk_ + 1] - a.att_key_ptr[bk_], Nk) : Nk;
  if (rowok) {
    float* Sr = Sc + lr * 68;
    float v[4], mx = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = q16 + 16 * jj;
      v[jj] = -INFINITY;
      if (j < nk) {
        v[jj] = Sr[j];
        for (int kq = 1; kq < KS; ++kq) v[jj] += Qs[R * LDK + (kq - 1) * R * 68 + lr * 68 + j];
      }
      mx = fmaxf(mx, v[jj]);
    }
    mx = row16_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const float e = (q16 + 16 * jj) < nk ? expf(v[jj] - mx) : 0.f;
      v[jj] = e;
      sum += e;
    }
    const float inv = 1.f / row16_sum(sum);
    const size_t prow = ((size_t)al_bq * Sq + s) * Nk;
    float ps = 0.f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = q16 + 16 * jj;
      if (j >= NkP) continue;
      float pm = 0.f;
      if (j < Nk) {
        const float pr = v[jj] * inv;
        pm = a.att_mask ? pr * a.att_mask[prow + j] : pr;
        if (rv) a.att_probs[prow + j] = pr;            // the un-dropped P (the backward reads it)
      }
      Sr[j] = pm;                                      // (zeros beyond Nk: the padded keys of the second product)
      ps += pm;
    }
    psum = a.att_mask ? row16_sum(ps) : 1.f;
  }
  FSTAMP(13);
  __syncthreads();
  // ---- D: O = P . K ----
  {
    const int nct = H >> 4, njobs = (R / 16) * nct;
    const UDiv dnct(nct);
    for (int job = wave; job < njobs; job += 8) {
      const int rt = dnct.div(job), ct = job - rt * nct;
      const f32x4 acc = mma_kn<4>(Sc + (16 * rt + l15) * 68 + 4 * g4, Ks + (4 * g4) * LDK + 16 * ct + l15, LDK, NkP >> 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) Qs[(16 * rt + 4 * g4 + i) * LDK + 16 * ct + l15] = acc[i];
    }
  }
  FSTAMP(14);
  __syncthreads();
  FSTAMP(15);
  // ---- E: residual, statistics, LN1 -> Xs ----
  if (rowok) {
    float4 x1[2];
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = q16 * 4 + 64 * k;
      x1[k] = f4zero();
      if (!on[k]) continue;
      const float4 o = ld4(Qs + lr * LDK + c);
      x1[k] = make_float4(o.x * g0[k].x + b0[k].x * psum + xr[k].x, o.y * g0[k].y + b0[k].y * psum + xr[k].y,
                          o.z * g0[k].z + b0[k].z * psum + xr[k].z, o.w * g0[k].w + b0[k].w * psum + xr[k].w);
      if (rv) st4(a.att_x1 + (size_t)r * a.att_ldx1 + c, x1[k]);
      t += (x1[k].x + x1[k].y) + (x1[k].z + x1[k].w);
    }
    const float mean1 = row16_sum(t) * invH;
    t = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (!on[k]) continue;
      const float p0 = x1[k].x - mean1, p1 = x1[k].y - mean1, p2 = x1[k].z - mean1, p3 = x1[k].w - mean1;
      t += (p0 * p0 + p1 * p1) + (p2 * p2 + p3 * p3);
    }
    const float rstd1 = rsqrtf(row16_sum(t) * invH + DOSX_LN_EPS);
    if (rv && q16 == 0) {
      a.att_qstats[2 * (size_t)r] = mean; a.att_qstats[2 * (size_t)r + 1] = rstd;
      a.att_st1[2 * (size_t)r] = mean1;   a.att_st1[2 * (size_t)r + 1] = rstd1;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (!on[k]) continue;
      const float4 vv = x1[k];
      st4(Xs + lr * LDX + q16 * 4 + 64 * k,
          make_float4((vv.x - mean1) * rstd1 * g1[k].x + bb1[k].x, (vv.y - mean1) * rstd1 * g1[k].y + bb1[k].y,
                      (vv.z - mean1) * rstd1 * g1[k].z + bb1[k].z, (vv.w - mean1) * rstd1 * g1[k].w + bb1[k].w));
    }
  }
}

// ATT (round 4): the attention half of the layer runs in the prologue, for key sets of <= 16 rows per crystal (DosxFfn.att_*):
// per query row - one QUARTER WAVE per row, like the row phases of attention.hip - LayerNorm-0, the <= 16 scores against
// the crystal's pre-normalised key rows (read straight from L2: 16 x 512 B per row, the key set of a crystal is shared by
// the Sq rows that name it), the exact fp32 softmax, P.K, the residual, the LayerNorm-1 statistics - and the normalised
// row goes into the Xs tile where the plain kernel puts LN1(x).  No MFMA: 2 x 16 x H multiply-adds per row are ~1 % of the
// feed-forward half's.  The key affine is folded out exactly as in attn_fwd_stream_kernel: (q.(khat g + b)) = (q o g).khat
// + q.b, the second term is the same for every key of the row (padded keys included: khat = 0) and cancels in the softmax;
// P.(khat g + b) = (P.khat) o g + b sum(P).
// HALF: the workgroup owns 16 rows and multiplies with the 16x16x4 MFMA (two 16-column tiles per wave instead of one
// 32-column tile): twice the workgroups, half the MFMA time each, and two of them fit the LDS of one CU.
// ATT = 2 (round 4: tile layout, one key per iteration on the vector ALU - slower than two launches; round 5: the two products
// on the MFMA, ffn_att_tile): CRYSTAL-ALIGNED tiles - a workgroup owns R consecutive query rows s of ONE query batch entry bq
// (row r = s * Bq + bq, grid = Bq x ceil(Sq / R)), so the tile's rows share one key set: the crystal's <= 64 pre-normalised key
// rows are copied to LDS once (into the stage-buffer region, which the weight chunks take over after the prologue) - the 51-key
// self attention and the 32-row launches, where the per-row global key fetch of ATT = 1 does not pay, take this form while the
// grid stays one round of workgroups.
template <bool HALF, int KB, int ATT>
__device__ __forceinline__ void ffn_fwd_body(const DosxFfn& a, float* __restrict__ sm) {
  constexpr int FBK = KB, FLDW = KB + 4;           // chunk width, padded row of a staged weight chunk
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
  // tile row lr -> global row (-1: beyond the data).  ATT = 2: rows s0 .. s0 + R - 1 of query batch entry al_bq
  int al_bq = 0, al_s0 = 0;
  if constexpr (ATT == 2) {
    const int tpc = (a.att_Sq + R - 1) / R;
    al_bq = (int)blockIdx.x / tpc;
    al_s0 = ((int)blockIdx.x % tpc) * R;
  }
  auto grow = [&](const int lr) -> int {
    if constexpr (ATT == 2) {
      const int s = al_s0 + lr;
      return s < a.att_Sq ? s * a.att_Bq + al_bq : -1;
    } else {
      const int r = m0 + lr;
      return r < M ? r : -1;
    }
  };
  auto growc = [&](const int lr) -> int {          // clamped to a valid row of the tile (duplicates are never stored)
    const int r = grow(lr);
    return r >= 0 ? r : grow(0);
  };
  FSTAMP(0);
  const int nk1 = H / FBK, nb1 = H4 / FBN, n1 = nb1 * nk1, n2 = H4 / FBK, nch = n1 + n2;

  // epilogue operands of this wave's 4 rows, fetched at kernel start (all 8 waves)
  const int c0 = lane * 4;
  const bool con = c0 < H;
  float4 xres[ER], bias2 = f4zero();
  if (con) bias2 = ld4(a.b2 + c0);
  if constexpr (ATT == 0) {
#pragma unroll
    for (int i = 0; i < ER; ++i) {
      const int r = growc(wave * ER + i);
      xres[i] = ld4(a.x + (size_t)r * a.ldx + (con ? c0 : 0));
    }
  }
  if constexpr (ATT == 2) {
    // the crystal's key rows -> LDS (all 8 waves), in the stage-buffer region: [Nk][H + 4], the score rows [R][68] behind them
    // (the crystal's key rows go to LDS inside ffn_att_tile, behind the loads of its row phase A: one round trip, not two)
  }
  // ATT = 2: keys Ks [NkP][H + 4] and scores Sc [R][68] in the stage-buffer region, Q / O tile [R][H + 4] in the (still unused) T region
  float* const attKs = ST;
  float* const attSc = ST + ((a.att_Nk + 15) & ~15) * (H + 4);

  if (wave_u >= 4) {
    // =============================== staging waves ===============================================
    constexpr int TPR = FBK / 4, RPP = 256 / TPR, NP = FBN / RPP;      // threads per chunk row, rows per pass, passes
    const int st = tid - 256, jr = st / TPR, kq = (st % TPR) * 4;
    const float* wlo = a.w1 < a.w2 ? a.w1 : a.w2;
    const uint32_t d1 = (uint32_t)((const char*)a.w1 - (const char*)wlo), d2 = (uint32_t)((const char*)a.w2 - (const char*)wlo);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)wlo, 0, 0x7fffffff, 0x00020000);
    uint32_t v1[NP], v2[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      v1[i] = d1 + (uint32_t)(((jr + RPP * i) * H + kq) * 4);                      // fc1 rows: (cb*128 + j), j = jr + RPP i
      v2[i] = d2 + (uint32_t)((min(jr + RPP * i, H - 1) * H4 + kq) * 4);           // fc2 rows: j < H (clamped)
    }
    float4 r0[NP], r1[NP];
    // chunks are issued strictly in order 0, 1, 2, ...: the scalar offset of the next one is carried along (cu / nk1
    // and cu % nk1 by a run-time nk1 are ~40 emulated-division instructions, and every vector-ALU instruction of a
    // staging wave waits ~30 clk for an issue slot between the matrix wave's MFMAs: `issue` took up to 2200 clk)
    int nxt_c = 0, nxt_kc = 0, nxt_so1 = 0;
    auto issue = [&](float4(&r)[NP], int c) {
      (void)c;
      const int cu = __builtin_amdgcn_readfirstlane(nxt_c);
      const bool p1 = cu < n1;
      const int so = __builtin_amdgcn_readfirstlane(p1 ? nxt_so1 : (cu - n1) * FBK * 4);
#pragma unroll
      for (int i = 0; i < NP; ++i)
        r[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rW, p1 ? v1[i] : v2[i], so, 0));
      ++nxt_c;
      if (++nxt_kc == nk1) { nxt_kc = 0; nxt_so1 += (FBN * H - (nk1 - 1) * FBK) * 4; } else nxt_so1 += FBK * 4;
    };
    auto store = [&](float* buf, const float4(&r)[NP]) {
#pragma unroll
      for (int i = 0; i < NP; ++i) st4(buf + (jr + RPP * i) * FLDW + kq, r[i]);
    };
    issue(r0, 0);
    issue(r1, 1);
    if constexpr (ATT == 2) {
      ffn_att_tile<R>(a, Xs, LDX, attKs, attSc, T, al_s0, al_bq, tid);
      __syncthreads();                             // the attention prologue is done with the keys / scores in the stage buffers
    }
    store(ST, r0);
    issue(r0, 2);
    __syncthreads();                               // (matrix waves: Xs written) chunk 0 visible
    for (int c = 0; c < nch; c += 2) {
      if (c + 1 < nch) {
        store(ST + STG, r1);
        if (c + 3 < nch) issue(r1, c + 3);
      }
      __syncthreads();
      if (c + 1 == n1) __syncthreads();                      // phase boundary: T complete behind this extra barrier
      if (c + 1 >= nch) break;
      if (c + 2 < nch) {
        store(ST, r0);
        if (c + 4 < nch) issue(r0, c + 4);
      }
      __syncthreads();
      if (c + 2 == n1) __syncthreads();
    }
  } else {
    // =============================== matrix waves ================================================
    // fc1 bias of this lane's columns in every 128-column block, fetched up front: loaded at the end of a block it
    // cost one exposed global round trip (~1600 clk) per block - 4 per kernel (stamps, tools/stamp_ffn.py)
    float b1r[4][2];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const int col = min(cb, nb1 - 1) * FBN + wave * 32 + (HALF ? l15 : l31);
      b1r[cb][0] = a.b1[col];
      b1r[cb][1] = HALF ? a.b1[col + 16] : 0.f;
    }
    if constexpr (ATT == 2) {
      ffn_att_tile<R>(a, Xs, LDX, attKs, attSc, T, al_s0, al_bq, tid);
      __syncthreads();                             // keys / scores dead: the staging waves may store the first weight chunk
    } else if constexpr (ATT == 1) {
      // 16 rows per pass over the 4 matrix waves.  (All 8 waves in one pass - the staging waves taking rows 16-31 while their
      // first weight chunks are in flight - was built and spills: the 16 x 2 float4 key registers next to the staged chunks.)
#pragma unroll
      for (int p = 0; p < R / 16; ++p) ffn_att_row(a, Xs, LDX, p * 16 + wave * 4 + (lane >> 4), m0, lane);
    } else {   // LN1(x) tile -> Xs  (row r = tid/8, 4-float groups tid%8 + 8 i)
      const int r = tid >> 3, rr = growc(min(r, R - 1));
      const float mean = a.stats[2 * (size_t)rr], rstd = a.stats[2 * (size_t)rr + 1];
      for (int c = (tid & 7) * 4; c < H && r < R; c += 32) {
        const float4 v = ld4(a.x + (size_t)rr * a.ldx + c), g = ld4(a.gamma + c), b = ld4(a.beta + c);
        st4(Xs + r * LDX + c, make_float4((v.x - mean) * rstd * g.x + b.x, (v.y - mean) * rstd * g.y + b.y,
                                          (v.z - mean) * rstd * g.z + b.z, (v.w - mean) * rstd * g.w + b.w));
      }
    }
    __syncthreads();
    FSTAMP(1);
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
        const float b1a = cb == 0 ? b1r[0][0] : cb == 1 ? b1r[1][0] : cb == 2 ? b1r[2][0] : b1r[3][0];
        const float b1b = cb == 0 ? b1r[0][1] : cb == 1 ? b1r[1][1] : cb == 2 ? b1r[2][1] : b1r[3][1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float h0 = fmaxf(acc0[r] + b1a, 0.f), h1 = fmaxf(acc1[r] + b1b, 0.f);
          T[(4 * g4 + r) * LDT + col] = h0;
          T[(4 * g4 + r) * LDT + col + 16] = h1;
          const int gr = grow(4 * g4 + r);
          if (gr >= 0) {                           // h -> HBM straight from the accumulators (64-byte row segments)
            a.h[(size_t)gr * a.ldh + col] = h0;
            a.h[(size_t)gr * a.ldh + col + 16] = h1;
          }
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
      const float b1 = cb == 0 ? b1r[0][0] : cb == 1 ? b1r[1][0] : cb == 2 ? b1r[2][0] : b1r[3][0];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // h goes to the LDS tile (A operand of fc2) AND straight to HBM from the accumulators (one 128-byte row segment
        // per half wave): the staging waves used to copy the finished 64 KB tile out in one burst at the phase boundary
        // and fell a tile behind with the fc2 weight chunks - 3.2 of 26.2 us at M = 6528
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
        const float hvv = fmaxf(acc[r] + b1, 0.f);
        T[row * LDT + col] = hvv;
        const int gr = grow(row);
        if (gr >= 0) a.h[(size_t)gr * a.ldh + col] = hvv;
      }
    }
    FSTAMP(2);
    __syncthreads();                               // T complete (the staging waves copy it out from here on)
    FSTAMP(3);
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
    FSTAMP(4);
    // C tile (the stage buffers are dead: the last chunk ended with a barrier)
    float* Cs = ST;
    const int col = wave * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) Cs[((r & 3) + 8 * (r >> 2) + 4 * hh) * (FBN + 4) + col] = acc[r];
    }   // !HALF
  }
  __syncthreads();
  FSTAMP(5);
  if constexpr (ATT != 0) {    // the residual rows are the x1 rows the prologue wrote (this workgroup's own stores: many barriers ago)
#pragma unroll
    for (int i = 0; i < ER; ++i) {
      const int r = growc(wave * ER + i);
      xres[i] = ld4(a.att_x1 + (size_t)r * a.att_ldx1 + (con ? c0 : 0));
    }
  }
  // ---- row epilogue (8 waves x ER rows): out = C + b2 + x, optionally followed by the encoder's final LayerNorm ----
  {
    const float* Cs = ST;
    const bool fin = a.fin_gamma != nullptr;
    const bool fdot = fin && a.fin_dos != nullptr;       // + the H -> 1 output layer on the normalised rows (model head)
    float4 fg = f4zero(), fb = f4zero(), fw = f4zero();
    if (fin && con) { fg = ld4(a.fin_gamma + c0); fb = ld4(a.fin_beta + c0); }
    if (fdot && con) fw = ld4(a.fin_w + c0);
    const float fbias = fdot ? a.fin_b[0] : 0.f;
    const float invH = 1.f / (float)H;
#pragma unroll
    for (int i = 0; i < ER; ++i) {
      const int lr = wave * ER + i, r = grow(lr);
      const bool ok = con && r >= 0;               // (wave-uniform)
      float4 o = f4zero();
      if (ok) {
        const float4 v = ld4(Cs + lr * (FBN + 4) + c0);
        o = make_float4(v.x + bias2.x + xres[i].x, v.y + bias2.y + xres[i].y, v.z + bias2.z + xres[i].z,
                        v.w + bias2.w + xres[i].w);
      }
      if (!fin) {
        if (ok) st4(a.out + (size_t)r * a.ldo + c0, o);
        continue;
      }
      const float mean = wave_sum(o.x + o.y + o.z + o.w) * invH;        // lanes beyond H / rows beyond M hold zeros
      const float4 d = ok ? make_float4(o.x - mean, o.y - mean, o.z - mean, o.w - mean) : f4zero();
      const float rstd = rsqrtf(wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * invH + DOSX_LN_EPS);
      float dot = 0.f;
      if (ok) {
        const float4 xh = make_float4(d.x * rstd, d.y * rstd, d.z * rstd, d.w * rstd);
        const float4 y = make_float4(xh.x * fg.x + fb.x, xh.y * fg.y + fb.y, xh.z * fg.z + fb.z, xh.w * fg.w + fb.w);
        st4(a.fin_xhat + (size_t)r * H + c0, xh);
        if (a.out) st4(a.out + (size_t)r * a.ldo + c0, y);
        if (lane == 0) a.fin_rstd[r] = rstd;
        dot = y.x * fw.x + y.y * fw.y + y.z * fw.z + y.w * fw.w;
      }
      if (fdot) {                                    // dos[bq][s] of row r = s * Bq + bq  (what dosx_ln_rowdot writes)
        dot = wave_sum(dot);
        if (lane == 0 && r >= 0) a.fin_dos[(size_t)(r % a.fin_Bq) * a.fin_S + (r / a.fin_Bq)] = dot + fbias;
      }
    }
  }
  FSTAMP(6);
}

template <bool HALF, int KB, int ATT>
__global__ __launch_bounds__(512) void ffn_fwd_kernel(const DosxFfn a) {
  DOSX_SET_MAIN_PRIO();
  extern __shared__ __align__(16) float sm[];
  ffn_fwd_body<HALF, KB, ATT>(a, sm);
}

// Round 6: the layers of ONE encoder stack in one launch.  Every transformer layer attends over the ORIGINAL keys
// (layers/transformer.py:72-73), so with the attention half inside the feed-forward launch a layer is ROW-LOCAL per tile: the
// rows a workgroup writes as layer t's output are exactly the rows it reads as layer t + 1's input (same tiles in every layer:
// same Sq / Bq / tile height).  The workgroup therefore runs the layers back to back - its own stores drained (vmcnt(0)) and a
// workgroup barrier in between; the rows were never read before in this launch, so no stale line can sit in the CU's L1 - and the
// stack costs one launch instead of T.  All layers of a call share the template instance, grid and LDS size.
// (two layers per launch - the reference's default depth, `--transformer 2`; deeper stacks go pair by pair.  The two bodies are two
//  inlined copies with compile-time descriptor offsets: a loop over a descriptor array made the compiler hold the descriptor in
//  vector registers - 256 VGPRs and scratch where the single-layer kernel needs 220.)
constexpr int FFN_MULTI_MAX = 2;
struct FfnMulti {
  DosxFfn d[FFN_MULTI_MAX];
  int n;
};
template <bool HALF, int KB, int ATT>
__global__ __launch_bounds__(512) void ffn_fwd_multi_kernel(const FfnMulti A) {
  DOSX_SET_MAIN_PRIO();
  extern __shared__ __align__(16) float sm[];
  ffn_fwd_body<HALF, KB, ATT>(A.d[0], sm);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ffn_fwd_body<HALF, KB, ATT>(A.d[1], sm);
}


// ------------------------------------------------------------------------------------------------------------------
// Backward of the same half layer, one launch:
//     dh  = (dy . W2) o [h > 0]                              [M,4H]   (written out: the fc1 weight gradient needs it)
//     dx  = dy + LN1_bwd( dh . W1 )                          [M,H]
//     partials[wg] = [ sum_rows dyl*xhat | sum_rows dyl ]    (dgamma | dbeta of LN1; dyl = dh . W1)
// Same structure as the forward: the R x 4H tile of dh stays in LDS as the A operand of the second GEMM.  Both weight
// matrices are read as stored (k-major: W2 [H][4H], W1 [4H][H]), so a staged chunk is 32 k-rows x 128 columns.
// ------------------------------------------------------------------------------------------------------------------
// The attention half's BACKWARD for a crystal-aligned tile, behind the feed-forward half's (ffn_bwd_kernel<.., ATT = 2>,
// round 5): what dosx_attention_bwd's one-launch form (attention.hip: attn_bwd_dq_stream_kernel<NJ, PKV = true> + the
// in-launch key-gradient reduction) computes, on the tile's R rows whose gradient dO = dL/dx1 the row epilogue has just left
// in LDS (Os) - the tile's rows share one key set, so the four products are small 16x16x4 MFMA jobs dealt over the 8 waves:
//   a: quarter wave per row - Ds = dO o g0                          e: quarter wave per row - dq_ln = (dS.K) o g0, LayerNorm-0
//   b: dP = Ds . Ks^T -> Sc (K split over idle waves)                  backward + residual -> dxin; query-side dg0 / db0 slots;
//   c: quarter wave per row - dS = P o (dP' - sum P dP') scale,        Ql = LN0(x) g0 + b0
//      dP' = (dP + dO.b0) o M with a dropout mask; P' = P o M       f: dK^ share = P'^T . dO + dS^T . Ql  [NkP, H] -> the tile's slot
//   d: dq = dS . Ks -> Ds                                              of dkv_part (write-through), ticket on the key crystal's
//                                                                      counter, the last arriver reduces (ffn_dkv_reduce)
// Same saved tensors, scratch layout (dkv_part [Bq][tiles][Nk][H], partials_q [Bq][tiles][2H], partials_kv [Bk][groups][2H]) and
// summation orders of the partials as the stand-alone kernel with 32-query tiles, so the two forms are interchangeable per layer.
struct AttBwdSm {
  float *Os, *Ds, *Ql, *Sp, *Ks, *Sc, *Ss, *Ps2, *Pp;
};

// ALL key rows of crystal bk at once (<= 64: two rows per quarter-wave slot, their loads requested together - the last arriving
// tile runs this alone at the end of its launch, so it is one round trip, not one per group): sum of the partial key gradients in
// (query batch entry, tile) order, key-side chain rule, dkvhat (+)=, and per group of 16 key rows the [dg0 | db0] partial row,
// slots added in row order - the arithmetic and orders of attention.hip's dkv_reduce_group.  Pp: [64][256] floats of LDS.
// Contains a barrier.
__device__ __forceinline__ void ffn_dkv_reduce(const DosxFfnBwd& a, const int nqt, const int bk, float* __restrict__ Pp, const int tid) {
  const int lane = tid & 63, q16 = lane & 15, slot = tid >> 4;                 // 32 slots
  const int H = a.H, Nk = a.att_Nk, rep = a.att_Bq / a.att_Bk, ngroups = (Nk + 15) / 16;
  const int np = rep * nqt;
  const UDiv dnqt(nqt);
  const size_t pstride = (size_t)Nk * H;
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)a.att_dkv_part, 0, 0x7fffffff, 0x00020000);
  float4 g0[2], d[2][2], kh[2][2], d0[2][2];
  bool jv[2];
  size_t krow[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) g0[k] = ld4(a.att_gamma0 + ((q16 * 4 + 64 * k) < H ? (q16 * 4 + 64 * k) : 0));
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int j = slot + 32 * p;
    jv[p] = j < Nk;
    krow[p] = ((size_t)(jv[p] ? j : 0) * a.att_Bk + bk) * H;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = q16 * 4 + 64 * k, cc = c < H ? c : 0;
      kh[p][k] = ld4(a.att_kvhat + krow[p] + cc);
      d0[p][k] = a.att_dkv_accumulate ? ld4(a.att_dkvhat + krow[p] + cc) : f4zero();
      d[p][k] = f4zero();
    }
  }
  for (int p0 = 0; p0 < np; p0 += 4) {
    float4 v[2][4][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pi = min(p0 + u, np - 1), i = dnqt.div(pi), t = pi - i * nqt;
        const size_t off = ((size_t)(bk + i * a.att_Bk) * nqt + t) * pstride + (size_t)(jv[p] ? slot + 32 * p : 0) * H;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int c = q16 * 4 + 64 * k;
          v[p][u][k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rP, (uint32_t)((off + (c < H ? c : 0)) * 4), 0, 16));   // sc1
        }
      }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (p0 + u < np) {
#pragma unroll
          for (int k = 0; k < 2; ++k) d[p][k] = f4add(d[p][k], v[p][u][k]);
        }
  }
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = q16 * 4 + 64 * k;
      float4 pg = f4zero(), pb = f4zero();
      if (jv[p] && c < H) {
        pg = make_float4(d[p][k].x * kh[p][k].x, d[p][k].y * kh[p][k].y, d[p][k].z * kh[p][k].z, d[p][k].w * kh[p][k].w);
        pb = d[p][k];
        st4(a.att_dkvhat + krow[p] + c, make_float4(d[p][k].x * g0[k].x + d0[p][k].x, d[p][k].y * g0[k].y + d0[p][k].y,
                                                    d[p][k].z * g0[k].z + d0[p][k].z, d[p][k].w * g0[k].w + d0[p][k].w));
      }
      st4(Pp + (slot + 32 * p) * 256 + c, pg);
      st4(Pp + (slot + 32 * p) * 256 + 128 + c, pb);
    }
  __syncthreads();
  const UDiv dH(H), d2H(2 * H);
  for (int o = tid; o < ngroups * 2 * H; o += 512) {
    const int grp = d2H.div(o), c = o - grp * 2 * H;
    const int hi = dH.div(c), col = hi * 128 + (c - hi * H);
    float t = 0.f;
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) t += Pp[(grp * 16 + sl) * 256 + col];
    a.att_partials_kv[((size_t)bk * ngroups + grp) * 2 * H + c] = t;
  }
}

// the global operands of the attention epilogue's row phases + this thread's share of the crystal's key rows: requested by
// ffn_bwd_kernel right behind its row epilogue, so the round trip runs under the column-sum block in between
struct AttBwdRegs {
  float4 g0[2], b0[2], xr[2], kr[4];
  float mean, rstd, pr[4], mk[4];
};
template <int R>
__device__ __forceinline__ void ffn_att_bwd_prefetch(const DosxFfnBwd& a, AttBwdRegs& q, const int al_s0, const int al_bq, const int tid) {
  const int H = a.H, lane = tid & 63, wave = tid >> 6, q16 = lane & 15, g4 = lane >> 4;
  const int Nk = a.att_Nk, NkP = (Nk + 15) & ~15, Sq = a.att_Sq, bk = al_bq % a.att_Bk;
  const int lr = wave * 4 + g4;
  const int s = min(al_s0 + (lr < R ? lr : 0), Sq - 1);
  const size_t orow = (size_t)s * a.att_Bq + al_bq;
  const float* xrow = a.att_x + ((size_t)s * a.att_qs + (size_t)al_bq * a.att_qb) * a.att_ldxin;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c = q16 * 4 + 64 * k, cc = c < H ? c : 0;
    q.g0[k] = ld4(a.att_gamma0 + cc); q.b0[k] = ld4(a.att_beta0 + cc);
    q.xr[k] = c < H ? ld4(xrow + c) : f4zero();
  }
  q.mean = a.att_qstats[2 * orow]; q.rstd = a.att_qstats[2 * orow + 1];
  const size_t prow_ = ((size_t)al_bq * Sq + s) * Nk;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int j = q16 + 16 * jj, jc = j < Nk ? j : 0;
    q.pr[jj] = a.att_probs[prow_ + jc];
    q.mk[jj] = a.att_mask ? a.att_mask[prow_ + jc] : 1.f;
  }
  const int h4 = H >> 2;
  const UDiv dh4(h4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + 512 * i, j = dh4.div(e), c = (e - j * h4) * 4;
    q.kr[i] = (e < NkP * h4 && j < Nk) ? ld4(a.att_kvhat + ((size_t)j * a.att_Bk + bk) * H + c) : f4zero();
  }
}

template <int R>
__device__ __forceinline__ void ffn_att_bwd_tile(const DosxFfnBwd& a, float* __restrict__ sm, const AttBwdSm& L, const AttBwdRegs& q,
                                                 const int al_s0, const int al_bq, const int tile, const int nqt, const int tid) {
  const int H = a.H, LDK = a.H + 4;
  const int lane = tid & 63, wave = tid >> 6, q16 = lane & 15, l15 = lane & 15, g4 = lane >> 4;
  const int Nk = a.att_Nk, NkP = (Nk + 15) & ~15, Sq = a.att_Sq, bk = al_bq % a.att_Bk;
  const float scale = rsqrtf((float)H), invH = 1.f / (float)H;
  const int lr = wave * 4 + g4;                        // this quarter wave's row (R = 16: waves 0-3 only)
  const bool rowok = lr < R;
  const int s = min(al_s0 + (rowok ? lr : 0), Sq - 1);
  const bool rv = rowok && (al_s0 + lr) < Sq;
  const size_t orow = (size_t)s * a.att_Bq + al_bq;    // global row (valid memory also for the clamped duplicates)
  float4 g0[2], b0[2], xr[2], go[2];
  bool on[2];
  float pr[4], mk[4];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    on[k] = (q16 * 4 + 64 * k) < H;
    g0[k] = q.g0[k]; b0[k] = q.b0[k]; xr[k] = q.xr[k];
  }
  const float mean = q.mean, rstd = q.rstd;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) { pr[jj] = q.pr[jj]; mk[jj] = q.mk[jj]; }
  {   // the crystal's key rows -> Ks
    const int h4 = H >> 2;
    const UDiv dh4(h4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 512 * i, j = dh4.div(e), c = (e - j * h4) * 4;
      if (e < NkP * h4) st4(L.Ks + j * LDK + c, q.kr[i]);
    }
  }
  FSTAMP(17);
  // ---- a: dO rows (left in Os by the row epilogue; zeros beyond the data) -> Ds = dO o g0;  cq = dO . b0 (dropout only) ----
  float cq = 0.f;
  if (rowok) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      go[k] = f4zero();
      if (!on[k]) continue;
      go[k] = ld4(L.Os + lr * LDK + q16 * 4 + 64 * k);
      st4(L.Ds + lr * LDK + q16 * 4 + 64 * k, make_float4(go[k].x * g0[k].x, go[k].y * g0[k].y, go[k].z * g0[k].z, go[k].w * g0[k].w));
      t += (go[k].x * b0[k].x + go[k].y * b0[k].y) + (go[k].z * b0[k].z + go[k].w * b0[k].w);
    }
    if (a.att_mask) cq = row16_sum(t);
  }
  FSTAMP(18);
  __syncthreads();
  FSTAMP(19);
  // ---- b: dP (up to a row constant) = Ds . Ks^T ----
  const int nctS = NkP >> 4, njobsS = (R / 16) * nctS;
  int KS = njobsS >= 8 ? 1 : (njobsS >= 4 ? 2 : 4);
  while (KS > 1 && (H % (16 * KS)) != 0) KS >>= 1;
  const int klen = H / KS;
  const UDiv dKS(KS);
  for (int unit = wave; unit < njobsS * KS; unit += 8) {
    const int job = dKS.div(unit), kq = unit - job * KS;
    const int rt = job >= nctS ? job / nctS : 0, ct = job - rt * nctS;
    const f32x4 acc = mma_kk<8>(L.Ds + (16 * rt + l15) * LDK + kq * klen + 4 * g4, L.Ks + (16 * ct + l15) * LDK + kq * klen + 4 * g4, klen >> 4);
    float* Sq_ = kq == 0 ? L.Sc : L.Sp + (kq - 1) * R * 68;
#pragma unroll
    for (int i = 0; i < 4; ++i) Sq_[(16 * rt + 4 * g4 + i) * 68 + 16 * ct + l15] = acc[i];
  }
  FSTAMP(20);
  __syncthreads();
  // ---- c: dS = P o (dP' - sum_j P dP') scale -> Ss;  P' = P o M -> Ps2 (zeros beyond Nk / the data) ----
  if (rowok) {
    float dp[4], dot = 0.f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = q16 + 16 * jj;
      float t = 0.f;
      if (j < Nk) {
        t = L.Sc[lr * 68 + j];
        for (int kq = 1; kq < KS; ++kq) t += L.Sp[(kq - 1) * R * 68 + lr * 68 + j];
      }
      if (a.att_mask) t = (t + cq) * mk[jj];
      if (j < Nk) dot += pr[jj] * t;
      dp[jj] = t;
    }
    dot = row16_sum(dot);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = q16 + 16 * jj;
      if (j >= NkP) continue;
      const bool v = j < Nk && rv;
      L.Ss[lr * 68 + j] = v ? pr[jj] * (dp[jj] - dot) * scale : 0.f;
      L.Ps2[lr * 68 + j] = v ? pr[jj] * mk[jj] : 0.f;
    }
  }
  FSTAMP(21);
  __syncthreads();
  // ---- d: dq = dS . Ks -> Ds ----
  {
    const int nct = H >> 4, njobs = (R / 16) * nct;
    const UDiv dnct(nct);
    for (int job = wave; job < njobs; job += 8) {
      const int rt = dnct.div(job), ct = job - rt * nct;
      const f32x4 acc = mma_kn<4>(L.Ss + (16 * rt + l15) * 68 + 4 * g4, L.Ks + (4 * g4) * LDK + 16 * ct + l15, LDK, NkP >> 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) L.Ds[(16 * rt + 4 * g4 + i) * LDK + 16 * ct + l15] = acc[i];
    }
  }
  FSTAMP(22);
  __syncthreads();
  FSTAMP(23);
  // ---- e: LayerNorm-0 backward on the query rows + residual -> dxin; query-side dg0 / db0; Ql = LN0(x) g0 + b0 ----
  {
    float4 pg[2] = {f4zero(), f4zero()}, pb[2] = {f4zero(), f4zero()};
    if (rowok) {
      float4 d[2], xh[2];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        d[k] = f4zero(); xh[k] = f4zero();
        if (!(on[k] && rv)) continue;
        float4 dd = ld4(L.Ds + lr * LDK + q16 * 4 + 64 * k);
        dd = make_float4(dd.x * g0[k].x, dd.y * g0[k].y, dd.z * g0[k].z, dd.w * g0[k].w);
        const float4 xv = xr[k];
        const float4 h = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
        d[k] = dd; xh[k] = h;
        pg[k] = make_float4(dd.x * h.x, dd.y * h.y, dd.z * h.z, dd.w * h.w);
        pb[k] = dd;
        const float4 dh = make_float4(dd.x * g0[k].x, dd.y * g0[k].y, dd.z * g0[k].z, dd.w * g0[k].w);
        s1 += (dh.x + dh.y) + (dh.z + dh.w);
        s2 += (dh.x * h.x + dh.y * h.y) + (dh.z * h.z + dh.w * h.w);
      }
      s1 = row16_sum(s1) * invH; s2 = row16_sum(s2) * invH;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (!on[k]) continue;
        const float4 dd = d[k], h = xh[k];
        if (rv)
          st4(a.att_dxin + orow * a.att_lddxin + q16 * 4 + 64 * k,
              make_float4(rstd * (dd.x * g0[k].x - s1 - h.x * s2) + go[k].x, rstd * (dd.y * g0[k].y - s1 - h.y * s2) + go[k].y,
                          rstd * (dd.z * g0[k].z - s1 - h.z * s2) + go[k].z, rstd * (dd.w * g0[k].w - s1 - h.w * s2) + go[k].w));
        st4(L.Ql + lr * LDK + q16 * 4 + 64 * k,
            rv ? make_float4(h.x * g0[k].x + b0[k].x, h.y * g0[k].y + b0[k].y, h.z * g0[k].z + b0[k].z, h.w * g0[k].w + b0[k].w)
               : f4zero());
      }
    }
    // quarter-wave slots [32][2][128] (R = 16: the upper sixteen hold zeros)
    const int slot = wave * 4 + g4;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = q16 * 4 + 64 * k;
      st4(L.Pp + slot * 256 + c, pg[k]);
      st4(L.Pp + slot * 256 + 128 + c, pb[k]);
    }
  }
  __syncthreads();
  {
    float* prow = a.att_partials_q + ((size_t)al_bq * nqt + tile) * 2 * H;
    for (int c = tid; c < 2 * H; c += 512) {
      const int hi = c >= H ? 1 : 0, o = hi * 128 + (c - hi * H);
      float t = 0.f;
#pragma unroll
      for (int sl = 0; sl < 32; ++sl) t += L.Pp[sl * 256 + o];
      prow[c] = t;
    }
  }
  FSTAMP(24);
  // ---- f: this tile's share of dK + dV: [NkP keys] x [R queries] . [R queries] x [H] -> its slot of dkv_part (write-through) ----
  {
    float* part = a.att_dkv_part + ((size_t)al_bq * nqt + tile) * (size_t)Nk * H;
    const int nct = H >> 4, njobs = nctS * nct;
    const UDiv dnct(nct);
    for (int job = wave; job < njobs; job += 8) {
      const int jt = dnct.div(job), ct = job - jt * nct;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < R; kk += 16) {
        const float* pa = L.Ps2 + (kk + 4 * g4) * 68 + 16 * jt + l15;
        const float* sa = L.Ss + (kk + 4 * g4) * 68 + 16 * jt + l15;
        const float* b1 = L.Os + (kk + 4 * g4) * LDK + 16 * ct + l15;
        const float* b2 = L.Ql + (kk + 4 * g4) * LDK + 16 * ct + l15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[i * 68], b1[i * LDK], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[i * 68], b2[i * LDK], acc2, 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * jt + 4 * g4 + i, col = 16 * ct + l15;
        if (j < Nk) __hip_atomic_store(part + (size_t)j * H + col, acc[i] + acc2[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1
      }
    }
  }
  // ---- publish / ticket: the last arriving tile of key crystal bk finishes its key gradient ----
  FSTAMP(25);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  FSTAMP(26);
  int* flag = reinterpret_cast<int*>(sm);
  const int arrivers = (a.att_Bq / a.att_Bk) * nqt;
  if (tid == 0) *flag = dosx_ticket(a.att_dkv_cnt + bk);
  __syncthreads();
  const bool last = *flag == arrivers - 1;
  __syncthreads();                                  // (the flag word is about to be overwritten by the reduction's LDS rows)
  FSTAMP(27);
  if (last) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    ffn_dkv_reduce(a, nqt, bk, sm, tid);
    if (tid == 0) __hip_atomic_store(a.att_dkv_cnt + bk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    FSTAMP(28);
  }
}

constexpr int BLDW = FBN + 4;    // 132: padded rows of a k-major weight chunk

#ifndef DOSX_FFN_BWD_OCC
#define DOSX_FFN_BWD_OCC 2      // waves per SIMD the register budget allows (2 = one workgroup per CU, 208 VGPRs)
#endif
// ATT = 2 (round 5): CRYSTAL-ALIGNED tiles like ffn_fwd_kernel<.., 2> (a workgroup = R consecutive query rows s of ONE query
// batch entry, row r = s * Bq + bq, grid = Bq x ceil(Sq / R)), and the attention half's backward behind the row epilogue
// (ffn_att_bwd_tile): dx1 never reaches HBM, the layer's backward is one launch.
template <bool HALF, int KB, int ATT = 0>
__global__ __launch_bounds__(512, DOSX_FFN_BWD_OCC) void ffn_bwd_kernel(const DosxFfnBwd a) {
  DOSX_SET_MAIN_PRIO();
  extern __shared__ __align__(16) float sm[];
  constexpr int FBK = KB;
  constexpr int R = HALF ? 16 : 32;
  constexpr int ER = R / 8;
  constexpr int NV = HALF ? 8 : 16;                // C elements per lane per 128-column block
  const int H = a.H, H4 = 4 * a.H, M = a.M;
  const int LDX = H + 4, LDT = H4 + 4;
  float* Ys = sm;                                  // [R][LDX]  dy tile (A operand of the fc2 dgrad)
  float* T = Ys + R * LDX;                         // [R][LDT]  dh tile (A operand of the fc1 dgrad)
  float* ST = T + R * LDT;                         // 2 stage buffers [32][132]; later C tile [R][132] + Ps [8][2][128]
  constexpr int STG = FBK * BLDW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int l15 = lane & 15, g4 = lane >> 4;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int m0 = blockIdx.x * R;
  FSTAMP(0);
  const int nk1 = H / FBK, nb1 = H4 / FBN, n1 = nb1 * nk1, n2 = H4 / FBK, nch = n1 + n2;
  // tile row lr -> global row (-1: beyond the data).  ATT = 2: rows s0 .. s0 + R - 1 of query batch entry al_bq
  int al_bq = 0, al_s0 = 0, al_tile = 0, al_tpc = 1;
  if constexpr (ATT == 2) {
    al_tpc = (a.att_Sq + R - 1) / R;
    al_bq = (int)blockIdx.x / al_tpc;
    al_tile = (int)blockIdx.x % al_tpc;
    al_s0 = al_tile * R;
  }
  auto grow = [&](const int lr) -> int {
    if constexpr (ATT == 2) {
      const int s = al_s0 + lr;
      return s < a.att_Sq ? s * a.att_Bq + al_bq : -1;
    } else {
      const int r = m0 + lr;
      return r < M ? r : -1;
    }
  };
  auto growc = [&](const int lr) -> int {          // clamped to a valid row of the tile (duplicates are never stored)
    const int r = grow(lr);
    return r >= 0 ? r : grow(0);
  };

  // epilogue operands of this wave's ER rows, fetched at kernel start (all 8 waves)
  const int c0 = lane * 4;
  const bool con = c0 < H;
  float4 dyr[ER], xr[ER], gam = f4zero();
  float mean[ER], rstd[ER];
  if (con) gam = ld4(a.gamma + c0);
#pragma unroll
  for (int i = 0; i < ER; ++i) {
    const int r = growc(wave * ER + i);
    dyr[i] = f4zero();
    if (a.dy) dyr[i] = ld4(a.dy + (size_t)r * a.lddy + (con ? c0 : 0));
    xr[i] = ld4(a.x + (size_t)r * a.ldx + (con ? c0 : 0));
    mean[i] = a.stats[2 * (size_t)r];
    rstd[i] = a.stats[2 * (size_t)r + 1];
  }
  // The encoder's FINAL LayerNorm (layers/transformer.py:76-77) sits right behind the last layer's feed-forward half: with
  // fin_gamma set, `dy` is the gradient w.r.t. that LayerNorm's output and its backward runs here, on the rows each wave
  // holds for the epilogue anyway (one wave per row) - no stand-alone ln_bwd launch in front of this kernel.  The result
  // goes to fin_dy (the fc2 weight gradient reads it) and straight into the LDS tile of the first product.
  const bool fin = a.fin_gamma != nullptr;
  const bool fdot = fin && a.fin_ddos != nullptr;   // ... and in front of it the H -> 1 output layer (dy rows = ddos[r] * w)
  float4 pgf = f4zero(), pbf = f4zero(), pwf = f4zero();
  float pdb = 0.f;
  auto fin_rows = [&]() {
  {
    float4 gf = f4zero(), bf = f4zero(), wf = f4zero();
    if (con) gf = ld4(a.fin_gamma + c0);
    if (fdot && con) { bf = ld4(a.fin_beta + c0); wf = ld4(a.fin_w + c0); }
    const float invH = 1.f / (float)a.H;
#pragma unroll
    for (int i = 0; i < ER; ++i) {
      const int lr = wave * ER + i, r = grow(lr), rc = growc(lr);
      const bool ok = con && r >= 0;
      const float4 fx = ld4(a.fin_xhat + (size_t)rc * a.H + (con ? c0 : 0));
      const float frs = a.fin_rstd[rc];
      float4 d = dyr[i], dh = f4zero();
      if (fdot) {
        // row r = s * Bq + bq of the [S, Bq] row space; ddos is [Bq, S]
        const float dd = a.fin_ddos[(size_t)(rc % a.fin_Bq) * a.fin_S + (rc / a.fin_Bq)];
        d = make_float4(dd * wf.x, dd * wf.y, dd * wf.z, dd * wf.w);
        if (ok) {
          pwf.x += dd * (fx.x * gf.x + bf.x); pwf.y += dd * (fx.y * gf.y + bf.y);
          pwf.z += dd * (fx.z * gf.z + bf.z); pwf.w += dd * (fx.w * gf.w + bf.w);
        }
        if (lane == 0 && r >= 0) pdb += dd;
      }
      float s1 = 0.f, s2 = 0.f;
      if (con) {
        if (ok) {
          pgf.x += d.x * fx.x; pgf.y += d.y * fx.y; pgf.z += d.z * fx.z; pgf.w += d.w * fx.w;
          pbf = f4add(pbf, d);
        }
        dh = make_float4(d.x * gf.x, d.y * gf.y, d.z * gf.z, d.w * gf.w);
        s1 = (dh.x + dh.y) + (dh.z + dh.w);
        s2 = (dh.x * fx.x + dh.y * fx.y) + (dh.z * fx.z + dh.w * fx.w);
      }
      const float m1 = wave_sum(s1) * invH, m2 = wave_sum(s2) * invH;
      d = make_float4(frs * (dh.x - m1 - fx.x * m2), frs * (dh.y - m1 - fx.y * m2), frs * (dh.z - m1 - fx.z * m2),
                      frs * (dh.w - m1 - fx.w * m2));
      dyr[i] = d;
      if (con) {
        st4(sm + lr * (a.H + 4) + c0, d);                    // Ys[lr][c0]
        if (ok) st4(a.fin_dy + (size_t)r * a.lddy + c0, d);
      }
    }
  }
  };

  if (wave_u >= 4) {
    // =============================== staging waves ===============================================
    const int st = tid - 256;
    const float* wlo = a.w1 < a.w2 ? a.w1 : a.w2;
    const uint32_t d1 = (uint32_t)((const char*)a.w1 - (const char*)wlo), d2 = (uint32_t)((const char*)a.w2 - (const char*)wlo);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)wlo, 0, 0x7fffffff, 0x00020000);
    constexpr int NP = FBK * 32 / 256;             // float4 per staging thread per chunk
    uint32_t v1[NP], v2[NP], lds[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int lin = st + 256 * i, r = lin >> 5, c4 = (lin & 31) * 4;
      v1[i] = d2 + (uint32_t)((r * H4 + c4) * 4);                         // phase 1: W2 rows k, columns cb*128 + c4
      v2[i] = d1 + (uint32_t)((r * H + min(c4, H - 4)) * 4);              // phase 2: W1 rows k, columns c4 (< H, clamped)
      lds[i] = (uint32_t)(r * BLDW + c4);
    }
    float4 r0[NP], r1[NP];
    // (chunks are issued strictly in order: the scalar offset is carried along instead of dividing by a run-time nk1,
    //  see ffn_fwd_kernel)
    int nxt_c = 0, nxt_kc = 0, nxt_so1 = 0;
    auto issue = [&](float4(&r)[NP], int c) {
      (void)c;
      const int cu = __builtin_amdgcn_readfirstlane(nxt_c);
      const bool p1 = cu < n1;
      const int so = __builtin_amdgcn_readfirstlane(p1 ? nxt_so1 : (cu - n1) * FBK * H * 4);
#pragma unroll
      for (int i = 0; i < NP; ++i)
        r[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rW, p1 ? v1[i] : v2[i], so, 0));
      ++nxt_c;
      if (++nxt_kc == nk1) { nxt_kc = 0; nxt_so1 += (FBN - (nk1 - 1) * FBK * H4) * 4; } else nxt_so1 += FBK * H4 * 4;
    };
    auto store = [&](float* buf, const float4(&r)[NP]) {
#pragma unroll
      for (int i = 0; i < NP; ++i) st4(buf + lds[i], r[i]);
    };
    issue(r0, 0);
    issue(r1, 1);
    if (fin) fin_rows();                           // (this wave's rows of the final-LayerNorm backward, weight loads in flight)
    store(ST, r0);
    issue(r0, 2);
    __syncthreads();                               // (matrix waves: Ys written) chunk 0 visible
    for (int c = 0; c < nch; c += 2) {
      if (c + 1 < nch) {
        store(ST + STG, r1);
        if (c + 3 < nch) issue(r1, c + 3);
      }
      __syncthreads();
      if (c + 1 == n1) __syncthreads();
      if (c + 1 >= nch) break;
      if (c + 2 < nch) {
        store(ST, r0);
        if (c + 4 < nch) issue(r0, c + 4);
      }
      __syncthreads();
      if (c + 2 == n1) __syncthreads();
    }
  } else {
    // =============================== matrix waves ================================================
    if (fin) fin_rows();
    if (!fin) {   // dy tile -> Ys
      const int r = tid >> 3, rr = growc(min(r, R - 1));
      for (int c = (tid & 7) * 4; c < H && r < R; c += 32) st4(Ys + r * LDX + c, ld4(a.dy + (size_t)rr * a.lddy + c));
    }
    __syncthreads();
    FSTAMP(1);
    int c = 0;
    // C-fragment coordinates of this lane inside a 128-column block: rows crow(v), columns ccol(v)
    auto crow = [&](int v) { return HALF ? 4 * g4 + (v & 3) : (v & 3) + 8 * (v >> 2) + 4 * hh; };
    auto ccol = [&](int v) { return HALF ? wave * 32 + l15 + 16 * (v >> 2) : wave * 32 + l31; };
    float acc[NV];
    // ---- phase 1: dh tile, nb1 column blocks of 128, each nk1 chunks ----
    for (int cb = 0; cb < nb1; ++cb) {
      float hv[NV];                                // relu mask operand, in flight under the k-loop of the block
#pragma unroll
      for (int v = 0; v < NV; ++v)
        hv[v] = a.h[(size_t)growc(crow(v)) * a.ldh + cb * FBN + ccol(v)];
      if constexpr (HALF) {
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        for (int kc = 0; kc < nk1; ++kc, ++c) {
          const float* Ws = ST + (c & 1) * STG;
#pragma unroll
          for (int kk = 0; kk < FBK; kk += 16) {
            const float4 av = ld4(Ys + l15 * LDX + kc * FBK + kk + 4 * g4);
            const float* bp = Ws + (kk + 4 * g4) * BLDW + wave * 32 + l15;
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bp[0], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bp[16], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bp[BLDW], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bp[BLDW + 16], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bp[2 * BLDW], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bp[2 * BLDW + 16], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bp[3 * BLDW], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bp[3 * BLDW + 16], a1, 0, 0, 0);
          }
          __syncthreads();
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) { acc[v] = a0[v]; acc[4 + v] = a1[v]; }
      } else {
        f32x16 a0;
#pragma unroll
        for (int v = 0; v < 16; ++v) a0[v] = 0.f;
        for (int kc = 0; kc < nk1; ++kc, ++c) {
          const float* Ws = ST + (c & 1) * STG;
#pragma unroll
          for (int kk = 0; kk < FBK; kk += 8) {
            const float4 av = ld4(Ys + l31 * LDX + kc * FBK + kk + 4 * hh);
            const float* bp = Ws + (kk + 4 * hh) * BLDW + wave * 32 + l31;
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bp[0], a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bp[BLDW], a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bp[2 * BLDW], a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bp[3 * BLDW], a0, 0, 0, 0);
          }
          __syncthreads();
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = a0[v];
      }
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const float d = hv[v] > 0.f ? acc[v] : 0.f;
        T[crow(v) * LDT + cb * FBN + ccol(v)] = d;
        const int gr = grow(crow(v));
        if (gr >= 0) a.dh[(size_t)gr * a.lddh + cb * FBN + ccol(v)] = d;     // (see ffn_fwd_kernel)
      }
    }
    FSTAMP(2);
    __syncthreads();                               // T complete (the staging waves copy it out from here on)
    FSTAMP(3);
    // ---- phase 2: dyl = dh . W1, one 128-column block, n2 chunks, A operand = T ----
    if constexpr (HALF) {
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
      for (int kc = 0; kc < n2; ++kc, ++c) {
        const float* Ws = ST + (c & 1) * STG;
#pragma unroll
        for (int kk = 0; kk < FBK; kk += 16) {
          const float4 av = ld4(T + l15 * LDT + kc * FBK + kk + 4 * g4);
          const float* bp = Ws + (kk + 4 * g4) * BLDW + wave * 32 + l15;
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bp[0], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bp[16], a1, 0, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bp[BLDW], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bp[BLDW + 16], a1, 0, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bp[2 * BLDW], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bp[2 * BLDW + 16], a1, 0, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bp[3 * BLDW], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bp[3 * BLDW + 16], a1, 0, 0, 0);
        }
        __syncthreads();
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) { acc[v] = a0[v]; acc[4 + v] = a1[v]; }
    } else {
      f32x16 a0;
#pragma unroll
      for (int v = 0; v < 16; ++v) a0[v] = 0.f;
      for (int kc = 0; kc < n2; ++kc, ++c) {
        const float* Ws = ST + (c & 1) * STG;
#pragma unroll
        for (int kk = 0; kk < FBK; kk += 8) {
          const float4 av = ld4(T + l31 * LDT + kc * FBK + kk + 4 * hh);
          const float* bp = Ws + (kk + 4 * hh) * BLDW + wave * 32 + l31;
          a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bp[0], a0, 0, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bp[BLDW], a0, 0, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bp[2 * BLDW], a0, 0, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bp[3 * BLDW], a0, 0, 0, 0);
        }
        __syncthreads();
      }
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[v] = a0[v];
    }
    FSTAMP(4);
    float* Cs = ST;                                // C tile (the stage buffers are dead: the last chunk ended with a barrier)
#pragma unroll
    for (int v = 0; v < NV; ++v) Cs[crow(v) * BLDW + ccol(v)] = acc[v];
  }
  __syncthreads();
  FSTAMP(5);
  // ---- row epilogue (8 waves x ER rows): LayerNorm backward over the row + residual; column sums for dgamma / dbeta ----
  float4 pg = f4zero(), pb = f4zero();
  {
    const float* Cs = ST;
    const float invH = 1.f / (float)H;
#pragma unroll
    for (int i = 0; i < ER; ++i) {
      const int lr = wave * ER + i, r = grow(lr);
      const bool ok = con && r >= 0;               // (wave-uniform)
      float4 xh = f4zero(), dxh = f4zero();
      float s1 = 0.f, s2 = 0.f;
      if (ok) {
        const float4 dyl = ld4(Cs + lr * BLDW + c0);
        xh = make_float4((xr[i].x - mean[i]) * rstd[i], (xr[i].y - mean[i]) * rstd[i], (xr[i].z - mean[i]) * rstd[i],
                         (xr[i].w - mean[i]) * rstd[i]);
        pg.x += dyl.x * xh.x; pg.y += dyl.y * xh.y; pg.z += dyl.z * xh.z; pg.w += dyl.w * xh.w;
        pb = f4add(pb, dyl);
        dxh = make_float4(dyl.x * gam.x, dyl.y * gam.y, dyl.z * gam.z, dyl.w * gam.w);
        s1 = dxh.x + dxh.y + dxh.z + dxh.w;
        s2 = dxh.x * xh.x + dxh.y * xh.y + dxh.z * xh.z + dxh.w * xh.w;
      }
      const float m1 = wave_sum(s1) * invH, m2 = wave_sum(s2) * invH;
      float4 o = f4zero();
      if (ok)
        o = make_float4(rstd[i] * (dxh.x - m1 - xh.x * m2) + dyr[i].x, rstd[i] * (dxh.y - m1 - xh.y * m2) + dyr[i].y,
                        rstd[i] * (dxh.z - m1 - xh.z * m2) + dyr[i].z, rstd[i] * (dxh.w - m1 - xh.w * m2) + dyr[i].w);
      if constexpr (ATT == 2) {                    // dO = dL/dx1 stays in LDS for the attention half's backward (zeros beyond the data)
        if (con) st4(sm + lr * LDX + c0, o);
      } else if (ok) {
        st4(a.dx + (size_t)r * a.lddx + c0, o);
      }
    }
  }
  AttBwdRegs attq;
  if constexpr (ATT == 2) ffn_att_bwd_prefetch<R>(a, attq, al_s0, al_bq, tid);
  {
    // column sums of the 8 waves: [8][npv][128] (+ 8 scalars) in the dh tile's LDS (dead since the second product)
    float* Ps = T;
    const int npv = fdot ? 5 : (fin ? 4 : 2);      // dgamma | dbeta of LN1 (| dgamma | dbeta of the final LayerNorm (| dw))
    if (fin) __syncthreads();                      // (4-5 groups can reach past T into the C tile other waves still read)
    if (con) {
      st4(Ps + (wave * npv + 0) * FBN + c0, pg);
      st4(Ps + (wave * npv + 1) * FBN + c0, pb);
      if (fin) {
        st4(Ps + (wave * npv + 2) * FBN + c0, pgf);
        st4(Ps + (wave * npv + 3) * FBN + c0, pbf);
      }
      if (fdot) st4(Ps + (wave * npv + 4) * FBN + c0, pwf);
    }
    if (fdot && lane == 0) Ps[8 * 5 * FBN + wave] = pdb;
    __syncthreads();
    float* prow = a.partials + (size_t)blockIdx.x * a.partial_ld;
    const UDiv dHp(H);
    for (int c = tid; c < npv * H; c += 512) {
      const int which = dHp.div(c), col = c - which * H;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += Ps[(w * npv + which) * FBN + col];
      prow[which * H + col] = s;
    }
    if (fdot && tid == 0) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += Ps[8 * 5 * FBN + w];
      prow[5 * H] = s;
    }
  }
  if constexpr (ATT == 2) {
    __syncthreads();                               // the column sums are out of T / the C tile: everything but Os (= Ys) is free
    AttBwdSm L;
    const int LDK = H + 4, NkP = (a.att_Nk + 15) & ~15;
    L.Os = sm;
    L.Ds = sm + R * LDK;
    L.Ql = L.Ds + R * LDK;
    L.Sc = L.Ql + R * LDK;
    L.Ss = L.Sc + R * 68;
    L.Ps2 = L.Ss + R * 68;
    L.Sp = L.Ps2 + R * 68;                         // [3][R][68] partial dP tiles; afterwards the 32 quarter-wave slots Pp [32][256]
    L.Ks = L.Sp + 3 * R * 68;
    L.Pp = L.Sp;
    (void)NkP;
    FSTAMP(16);
    ffn_att_bwd_tile<R>(a, sm, L, attq, al_s0, al_bq, al_tile, al_tpc, tid);
    FSTAMP(29);
  }
}

}  // namespace

// floats of LDS the attention epilogue of ffn_bwd_kernel<.., 2> lays out (R rows per workgroup)
static size_t ffn_att_bwd_floats(int H, int Nk, int R) {
  const size_t LDK = (size_t)H + 4, NkP = (size_t)((Nk + 15) & ~15);
  const size_t tiles = 3 * R * LDK + 6 * (size_t)R * 68 + NkP * LDK;
  const size_t pp = 3 * R * LDK + 3 * (size_t)R * 68 + 32 * 256;          // (the query-side slots alias Sp | Ks)
  const size_t red = 64 * 256;                                              // key-gradient reduction: [64 key rows][dg0 | db0]
  size_t m = tiles > pp ? tiles : pp;
  return m > red ? m : red;
}

static int ffn_bwd_att_rows(int Sq, int Bq);

static int ffn_chunk(int H) {
  static int forced = -1;
  if (forced < 0) { const char* e = getenv("DOSX_FFN_KB"); forced = e ? atoi(e) : 0; }
  if (forced == 32) return 32;
  return (H % 64) == 0 ? 64 : 32;
}

extern "C" int dosx_ffn_supported(int H) { return (H % 32) == 0 && H >= 32 && H <= 128; }
extern "C" int dosx_ffn_att_supported(int H, int Nk) { return dosx_ffn_supported(H) && Nk >= 1 && Nk <= 16; }
// ... with crystal-aligned tiles (DosxFfn.att_aligned): <= 64 keys whose rows fit the stage-buffer region next to 32 score rows
extern "C" int dosx_ffn_att_aligned_supported(int H, int Nk) {
  if (!dosx_ffn_supported(H) || Nk < 1 || Nk > 64) return 0;
  const int kb = ffn_chunk(H);          // (what the launch uses - not a second copy of the policy)
  return (size_t)((Nk + 15) & ~15) * (H + 4) + 32 * 68 <= 2 * (size_t)128 * (kb + 4);
}

// validation + launch plan of one forward descriptor (shared by dosx_ffn_fwd and dosx_ffn_fwd_multi)
struct FfnPlan { bool half; int kb, mode, grid; size_t smem; };
static int ffn_fwd_prepare(const DosxFfn& a, FfnPlan& pl) {
  DOSX_CHECK_ARG(dosx_ffn_supported(a.H), "dosx_ffn_fwd: H=%d unsupported (multiple of 32, <= 128)", a.H);
  const bool att = a.att_kvhat != nullptr;
  DOSX_CHECK_ARG(a.x && (a.stats || att) && a.gamma && a.beta && a.w1 && a.b1 && a.w2 && a.b2 && a.h && (a.out || a.fin_dos), "dosx_ffn_fwd: null operand");
  if (att)
    DOSX_CHECK_ARG((a.att_aligned ? dosx_ffn_att_aligned_supported(a.H, a.att_Nk) : dosx_ffn_att_supported(a.H, a.att_Nk)) && a.att_gamma0 && a.att_beta0 && a.att_probs && a.att_qstats && a.att_x1 &&
                       a.att_st1 && a.att_Bk > 0 && a.att_Bq > 0 && a.att_Bq % a.att_Bk == 0 && a.att_Sq > 0 &&
                       a.att_Sq * a.att_Bq == a.M && (a.att_ldx1 & 3) == 0 && a.att_ldx1 >= a.H && a.att_qs >= 0 && a.att_qb >= 0,
                   "dosx_ffn_fwd: fused attention needs <= 16 keys, gamma0 / beta0 / probs / qstats / x1 / st1 and Sq * Bq == M");
  if (a.fin_dos) DOSX_CHECK_ARG(a.fin_gamma && a.fin_w && a.fin_b && a.fin_S > 0 && a.fin_Bq > 0 && a.fin_S * a.fin_Bq == a.M,
                                "dosx_ffn_fwd: output-layer epilogue needs the final LayerNorm, w, b and S * Bq == M");
  DOSX_CHECK_ARG((a.ldx & 3) == 0 && (a.ldh & 3) == 0 && (a.ldo & 3) == 0, "dosx_ffn_fwd: leading dimensions must be multiples of 4");
  if (a.fin_gamma) DOSX_CHECK_ARG(a.fin_beta && a.fin_xhat && a.fin_rstd, "dosx_ffn_fwd: final LayerNorm needs beta / xhat / rstd");
  const long long span = (const char*)a.w1 > (const char*)a.w2 ? (const char*)a.w1 - (const char*)a.w2 : (const char*)a.w2 - (const char*)a.w1;
  DOSX_CHECK_ARG(span + (long long)16 * a.H * a.H < 0x7fffffffLL, "dosx_ffn_fwd: fc1 / fc2 weights more than 2 GiB apart");
  const int H = a.H, H4 = 4 * H;
  static int half_max = -1;
  if (half_max < 0) { const char* e = getenv("DOSX_FFN_HALF_MAX"); half_max = e ? atoi(e) : 128; }
  const bool aligned = att && a.att_aligned != 0;        // crystal-aligned tiles (ATT = 2): grid = Bq x ceil(Sq / R)
  pl.half = aligned ? ffn_bwd_att_rows(a.att_Sq, a.att_Bq) == 16 : ceil_div(a.M, 32) <= half_max;   // 16-row workgroups while the 32-row grid is one partial round
  const int R = pl.half ? 16 : 32;
  pl.kb = ffn_chunk(H);
  pl.mode = aligned ? 2 : (att ? 1 : 0);
  pl.smem = sizeof(float) * ((size_t)R * (H + 4) + (size_t)R * (H4 + 4) + 2 * (size_t)FBN * (pl.kb + 4));
  if (aligned)
    DOSX_CHECK_ARG(a.att_Nk <= 64 && ((size_t)((a.att_Nk + 15) & ~15) * (H + 4) + (size_t)R * 68) <= 2 * (size_t)FBN * (pl.kb + 4),
                   "dosx_ffn_fwd: crystal-aligned attention needs <= 64 keys that fit the stage buffers (Nk=%d, H=%d)", a.att_Nk, H);
  pl.grid = aligned ? a.att_Bq * ceil_div(a.att_Sq, R) : ceil_div(a.M, R);
  static bool attr_set = false;
  if (!attr_set) {
#define DOSX_FFN_ATTR(HALF_, KB_, ATT_) \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fwd_kernel<HALF_, KB_, ATT_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
#define DOSX_FFN_ATTR2(HALF_, KB_, ATT_) \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fwd_multi_kernel<HALF_, KB_, ATT_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    DOSX_FFN_ATTR(false, 32, 0); DOSX_FFN_ATTR(true, 32, 0); DOSX_FFN_ATTR(false, 64, 0); DOSX_FFN_ATTR(true, 64, 0);
    DOSX_FFN_ATTR(false, 32, 1); DOSX_FFN_ATTR(true, 32, 1); DOSX_FFN_ATTR(false, 64, 1); DOSX_FFN_ATTR(true, 64, 1);
    DOSX_FFN_ATTR(false, 32, 2); DOSX_FFN_ATTR(true, 32, 2); DOSX_FFN_ATTR(false, 64, 2); DOSX_FFN_ATTR(true, 64, 2);
    DOSX_FFN_ATTR2(false, 32, 1); DOSX_FFN_ATTR2(true, 32, 1); DOSX_FFN_ATTR2(false, 64, 1); DOSX_FFN_ATTR2(true, 64, 1);
    DOSX_FFN_ATTR2(false, 32, 2); DOSX_FFN_ATTR2(true, 32, 2); DOSX_FFN_ATTR2(false, 64, 2); DOSX_FFN_ATTR2(true, 64, 2);
#undef DOSX_FFN_ATTR
#undef DOSX_FFN_ATTR2
    attr_set = true;
  }
  return 0;
}

#define DOSX_FFN_DISPATCH(KERNEL, ARG)                                                                                    \
  do {                                                                                                                     \
    const dim3 grid_(pl.grid);                                                                                             \
    hipStream_t st_ = to_stream(stream);                                                                                   \
    if (pl.half && pl.kb == 64) { if (pl.mode == 2) hipLaunchKernelGGL((KERNEL<true, 64, 2>), grid_, dim3(512), pl.smem, st_, ARG); else if (pl.mode == 1) hipLaunchKernelGGL((KERNEL<true, 64, 1>), grid_, dim3(512), pl.smem, st_, ARG); else hipLaunchKernelGGL((KERNEL<true, 64, 0>), grid_, dim3(512), pl.smem, st_, ARG); } \
    else if (pl.half) { if (pl.mode == 2) hipLaunchKernelGGL((KERNEL<true, 32, 2>), grid_, dim3(512), pl.smem, st_, ARG); else if (pl.mode == 1) hipLaunchKernelGGL((KERNEL<true, 32, 1>), grid_, dim3(512), pl.smem, st_, ARG); else hipLaunchKernelGGL((KERNEL<true, 32, 0>), grid_, dim3(512), pl.smem, st_, ARG); } \
    else if (pl.kb == 64) { if (pl.mode == 2) hipLaunchKernelGGL((KERNEL<false, 64, 2>), grid_, dim3(512), pl.smem, st_, ARG); else if (pl.mode == 1) hipLaunchKernelGGL((KERNEL<false, 64, 1>), grid_, dim3(512), pl.smem, st_, ARG); else hipLaunchKernelGGL((KERNEL<false, 64, 0>), grid_, dim3(512), pl.smem, st_, ARG); } \
    else { if (pl.mode == 2) hipLaunchKernelGGL((KERNEL<false, 32, 2>), grid_, dim3(512), pl.smem, st_, ARG); else if (pl.mode == 1) hipLaunchKernelGGL((KERNEL<false, 32, 1>), grid_, dim3(512), pl.smem, st_, ARG); else hipLaunchKernelGGL((KERNEL<false, 32, 0>), grid_, dim3(512), pl.smem, st_, ARG); } \
  } while (0)

extern "C" int dosx_ffn_fwd(const DosxFfn* ap, dosx_stream_t stream) {
  DOSX_CHECK_ARG(ap != nullptr, "dosx_ffn_fwd: null descriptor");
  const DosxFfn& a = *ap;
  if (a.M <= 0) return 0;
  FfnPlan pl;
  if (int rc = ffn_fwd_prepare(a, pl)) return rc;
  DOSX_FFN_DISPATCH(ffn_fwd_kernel, a);
  DOSX_LAUNCH_CHECK();
  return 0;
}

// The n <= 4 layers of ONE encoder stack in one launch (see ffn_fwd_multi_kernel): descs[l + 1] reads descs[l]'s output rows as its
// input rows (dense: row (s, bq) at s * Bq + bq), every layer carries the attention half (att_* set), all share the launch plan.
extern "C" int dosx_ffn_fwd_multi(const DosxFfn* descs, int n, dosx_stream_t stream) {
  DOSX_CHECK_ARG(descs != nullptr && n >= 1 && n <= FFN_MULTI_MAX, "dosx_ffn_fwd_multi: 1 .. %d layers per launch", FFN_MULTI_MAX);
  if (descs[0].M <= 0) return 0;
  if (n == 1) return dosx_ffn_fwd(descs, stream);
  FfnMulti A;
  A.n = n;
  FfnPlan pl;
  for (int l = 0; l < n; ++l) {
    FfnPlan p2;
    if (int rc = ffn_fwd_prepare(descs[l], p2)) return rc;
    if (l == 0) pl = p2;
    DOSX_CHECK_ARG(p2.half == pl.half && p2.kb == pl.kb && p2.mode == pl.mode && p2.grid == pl.grid && p2.smem == pl.smem && descs[l].M == descs[0].M,
                   "dosx_ffn_fwd_multi: layer %d has another launch plan than layer 0", l);
    DOSX_CHECK_ARG(p2.mode != 0, "dosx_ffn_fwd_multi: every layer must carry its attention half (row-local layers only)");
    if (l > 0) {
      const DosxFfn& p = descs[l - 1];
      const DosxFfn& c = descs[l];
      DOSX_CHECK_ARG(p.out != nullptr && c.x == p.out && c.ldx == p.ldo && c.att_qs == c.att_Bq && c.att_qb == 1,
                     "dosx_ffn_fwd_multi: layer %d must read layer %d's output rows (dense query rows)", l, l - 1);
    }
    A.d[l] = descs[l];
  }
  {
    const dim3 grid_(pl.grid);
    hipStream_t st_ = to_stream(stream);
#define DOSX_FFN_GO2(HALF_, KB_) \
    do { if (pl.mode == 2) hipLaunchKernelGGL((ffn_fwd_multi_kernel<HALF_, KB_, 2>), grid_, dim3(512), pl.smem, st_, A); \
         else hipLaunchKernelGGL((ffn_fwd_multi_kernel<HALF_, KB_, 1>), grid_, dim3(512), pl.smem, st_, A); } while (0)
    if (pl.half && pl.kb == 64) DOSX_FFN_GO2(true, 64);
    else if (pl.half) DOSX_FFN_GO2(true, 32);
    else if (pl.kb == 64) DOSX_FFN_GO2(false, 64);
    else DOSX_FFN_GO2(false, 32);
#undef DOSX_FFN_GO2
  }
  DOSX_LAUNCH_CHECK();
  return 0;
}

static int ffn_bwd_half(int M) {
  static int half_max = -1;
  if (half_max < 0) { const char* e = getenv("DOSX_FFN_HALF_MAX"); half_max = e ? atoi(e) : 128; }
  return ceil_div(M, 32) <= half_max;
}

extern "C" int dosx_ffn_bwd_partial_rows(int M) { return M <= 0 ? 0 : ceil_div(M, ffn_bwd_half(M) ? 16 : 32); }

// rows per workgroup of the crystal-aligned backward launch (DosxFfnBwd.att_*): 16 while the 32-row grid is at most half a round
static int ffn_bwd_att_rows(int Sq, int Bq) {
  static int half_max = -1;
  if (half_max < 0) { const char* e = getenv("DOSX_FFN_HALF_MAX"); half_max = e ? atoi(e) : 128; }
  return Bq * ceil_div(Sq, 32) <= half_max ? 16 : 32;
}
// whether dosx_ffn_bwd takes the att_* fields for this shape: hidden 64 / 128 (64-wide weight chunks), <= 64 keys, and the
// attention tiles fit the launch's LDS;  *_partial_rows: workgroups = partial rows of such a launch
extern "C" int dosx_ffn_att_bwd_supported(int H, int Nk, int Sq, int Bq) {
  if (!dosx_ffn_supported(H) || ffn_chunk(H) != 64 || Nk < 1 || Nk > 64 || Sq < 1 || Bq < 1) return 0;
  const int R = ffn_bwd_att_rows(Sq, Bq);
  const size_t ffn = (size_t)R * (H + 4) + (size_t)R * (4 * H + 4) + 2 * (size_t)64 * (FBN + 4);
  const size_t att = ffn_att_bwd_floats(H, Nk, R);
  return (ffn > att ? ffn : att) * sizeof(float) <= 160 * 1024;
}
extern "C" int dosx_ffn_att_bwd_partial_rows(int Sq, int Bq) { return Bq * ceil_div(Sq, ffn_bwd_att_rows(Sq, Bq)); }
// rows per workgroup of a crystal-aligned launch, forward and backward (the host's grid-size policy asks the library)
extern "C" int dosx_ffn_att_aligned_rows(int Sq, int Bq) { return (Sq < 1 || Bq < 1) ? 0 : ffn_bwd_att_rows(Sq, Bq); }

extern "C" int dosx_ffn_bwd(const DosxFfnBwd* ap, dosx_stream_t stream) {
  DOSX_CHECK_ARG(ap != nullptr, "dosx_ffn_bwd: null descriptor");
  const DosxFfnBwd& a = *ap;
  if (a.M <= 0) return 0;
  DOSX_CHECK_ARG(dosx_ffn_supported(a.H), "dosx_ffn_bwd: H=%d unsupported (multiple of 32, <= 128)", a.H);
  DOSX_CHECK_ARG((a.dy || a.fin_ddos) && a.h && a.x && a.stats && a.gamma && a.w1 && a.w2 && a.dh && (a.dx || a.att_kvhat) && a.partials, "dosx_ffn_bwd: null operand");
  DOSX_CHECK_ARG((a.lddy & 3) == 0 && (a.ldh & 3) == 0 && (a.ldx & 3) == 0 && (a.lddh & 3) == 0 && (a.lddx & 3) == 0,
                 "dosx_ffn_bwd: leading dimensions must be multiples of 4");
  const int npv = a.fin_gamma ? (a.fin_ddos ? 5 : 4) : 2;
  DOSX_CHECK_ARG(a.partial_ld >= npv * a.H + (a.fin_ddos ? 1 : 0), "dosx_ffn_bwd: partial_ld %d too small for %d column groups", a.partial_ld, npv);
  if (a.fin_gamma) DOSX_CHECK_ARG(a.fin_xhat && a.fin_rstd && a.fin_dy, "dosx_ffn_bwd: final LayerNorm backward needs xhat / rstd / fin_dy");
  if (a.fin_ddos) DOSX_CHECK_ARG(a.fin_gamma && a.fin_beta && a.fin_w && a.fin_S > 0 && a.fin_Bq > 0 && a.fin_S * a.fin_Bq == a.M,
                                 "dosx_ffn_bwd: output-layer backward needs gamma / beta / w and S * Bq == M");
  const long long span = (const char*)a.w1 > (const char*)a.w2 ? (const char*)a.w1 - (const char*)a.w2 : (const char*)a.w2 - (const char*)a.w1;
  DOSX_CHECK_ARG(span + (long long)16 * a.H * a.H < 0x7fffffffLL, "dosx_ffn_bwd: fc1 / fc2 weights more than 2 GiB apart");
  const int H = a.H, H4 = 4 * H;
  const bool att = a.att_kvhat != nullptr;
  if (att) {
    DOSX_CHECK_ARG(dosx_ffn_att_bwd_supported(H, a.att_Nk, a.att_Sq, a.att_Bq) && ffn_chunk(H) == 64, "dosx_ffn_bwd: fused attention backward unsupported for H=%d Nk=%d", H, a.att_Nk);
    DOSX_CHECK_ARG(a.att_x && a.att_gamma0 && a.att_beta0 && a.att_probs && a.att_qstats && a.att_dxin && a.att_partials_q && a.att_partials_kv &&
                       a.att_dkv_part && a.att_dkv_cnt && a.att_dkvhat && a.att_Bk > 0 && a.att_Bq % a.att_Bk == 0 && a.att_Sq * a.att_Bq == a.M &&
                       (a.att_ldxin & 3) == 0 && (a.att_lddxin & 3) == 0 && a.att_qs >= 0 && a.att_qb >= 0,
                   "dosx_ffn_bwd: fused attention backward needs x / gamma0 / beta0 / probs / qstats / dxin / partials / dkv scratch + counters and Sq * Bq == M");
  }
  const bool half = att ? ffn_bwd_att_rows(a.att_Sq, a.att_Bq) == 16 : ffn_bwd_half(a.M);
  const int R = half ? 16 : 32;
  const int kb = ffn_chunk(H);
  size_t fl = (size_t)R * (H + 4) + (size_t)R * (H4 + 4) + 2 * (size_t)kb * (FBN + 4);
  if (att && ffn_att_bwd_floats(H, a.att_Nk, R) > fl) fl = ffn_att_bwd_floats(H, a.att_Nk, R);
  const size_t smem = sizeof(float) * fl;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_bwd_kernel<false, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_bwd_kernel<true, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_bwd_kernel<false, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_bwd_kernel<true, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_bwd_kernel<false, 64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_bwd_kernel<true, 64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const dim3 grid(att ? a.att_Bq * ceil_div(a.att_Sq, R) : ceil_div(a.M, R));
  if (att && half) hipLaunchKernelGGL((ffn_bwd_kernel<true, 64, 2>), grid, dim3(512), smem, to_stream(stream), a);
  else if (att) hipLaunchKernelGGL((ffn_bwd_kernel<false, 64, 2>), grid, dim3(512), smem, to_stream(stream), a);
  else if (half && kb == 64) hipLaunchKernelGGL((ffn_bwd_kernel<true, 64>), grid, dim3(512), smem, to_stream(stream), a);
  else if (half) hipLaunchKernelGGL((ffn_bwd_kernel<true, 32>), grid, dim3(512), smem, to_stream(stream), a);
  else if (kb == 64) hipLaunchKernelGGL((ffn_bwd_kernel<false, 64>), grid, dim3(512), smem, to_stream(stream), a);
  else hipLaunchKernelGGL((ffn_bwd_kernel<false, 32>), grid, dim3(512), smem, to_stream(stream), a);
  DOSX_LAUNCH_CHECK();
  return 0;
}
