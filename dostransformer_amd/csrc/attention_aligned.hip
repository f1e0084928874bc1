// Attention over the <= 64 keys of ONE crystal on crystal-aligned query tiles, stand-alone (round 5): the kernels
// dosx_attention_fwd / dosx_attention_bwd run for key sets of up to 64 rows when the layer's feed-forward half is NOT in the
// same launch (hidden 256: the Electron-DOS cross attention over a crystal's <= 64 atoms, layers/multihead_attention.py:49-76
// inside layers/transformer.py:131-139) - the tile arithmetic of ffn.hip's ffn_att_tile / ffn_att_bwd_tile for any hidden size
// up to 256, with a workgroup that STAYS on its crystal:
//   grid = Bq x wpc workgroups; workgroup (bq, w) owns the 32-row query tiles w, w + wpc, ... of query batch entry bq;
//   the crystal's key rows (pre-normalised, [NkP][H + 4], rows beyond Nk zero) are copied to LDS ONCE per workgroup;
//   per tile: quarter-wave row phases (LayerNorm-0 / softmax / residual) between 16 x 16 x 4 fp32 MFMA jobs dealt over the 8
//   waves; the next tile's rows are requested while the current one is multiplied.
// The kernels they replace for these shapes (attention.hip: attn_fwd_stream_kernel / attn_bwd_dq_stream_kernel, one 32-query
// tile per workgroup, keys re-staged per tile, 4 matrix waves) ran at 9-13 % of the fp32 MFMA peak in the Electron-DOS step.
// Backward: a workgroup keeps its share of dK^ = P'^T.dO + dS^T.Q in registers ACROSS its tiles (one [Nk, H] share per
// workgroup instead of one per tile), publishes it to its slot of dkv_part, takes a ticket on the key crystal's counter, and the
// last arriver sums the shares in (query batch entry, workgroup) order, applies the key-side chain rule and writes dkvhat and the
// key-side [dgamma | dbeta] partial rows - DosxAttn's one-launch contract (include/dosx.h: dkv_part + dkv_cnt).  The query-side
// [dgamma | dbeta] partial rows: one per WORKGROUP in row (bq * nqt + w), the rows of the tiles it also owned are zeroed.
// Deterministic: fixed tile -> workgroup assignment, fixed summation orders.
#include <stdlib.h>

#include "common.h"
#include "mma16.h"

#ifdef DOSX_STAMPS
__device__ unsigned long long dosx_al_stamp_buf[64];
extern "C" int dosx_debug_read_attn_aligned_stamps(unsigned long long* host64) {
  return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(dosx_al_stamp_buf), sizeof(unsigned long long) * 64);
}
// workgroup 0, wave 0: slot + 16 * (tile ordinal, first two tiles); slot 0 = kernel start, 15 = end (ordinal 3)
#define ASTAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x == 0) dosx_al_stamp_buf[(slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ASTAMP(slot) do { } while (0)
#endif

namespace {

constexpr int R = 32;          // query rows per tile (one quarter wave per row)

struct AlGeo {
  int wpc, KS;
};

// The two product shapes of a tile as PIPELINED job streams (stamps of the first version, tools/stamp_attn_aligned.py: a wave's jobs
// ran load-all-fragments -> wait -> MFMAs one after the other - 5.0-5.3 k clk per phase for 3.1 k clk of MFMA issue per SIMD):
// the fragments of the NEXT (half-)job are requested before the MFMAs of the current one, two register sets.
//
// kk: C[r][j] = sum_k A[r][k] B[j][k] (scores, dP): both operands k-contiguous rows of pitch LDK.  Unit (job, kq) = the 16 x 16
// tile job = (rt, ct) over the k-blocks [kq HPU, (kq + 1) HPU) of 64, HPU = NG / KS; partial tile kq of Sc gets `scale` x its sum.
// Units are dealt over the 8 waves; a wave walks its units k-block by k-block.
template <int NG, int LDK>
__device__ __forceinline__ void al_kk_phase(const float* __restrict__ As, const float* __restrict__ Ks, float* __restrict__ Sc,
                                            const int nctS, const int njobsS, const int KS, const float scale, const int wave,
                                            const int l15, const int g4) {
  // (KS and HPU = NG / KS are powers of two, a tile has two row blocks: shifts and one compare - run-time integer divisions
  //  cost ~40 instructions each, six per half-job were most of this phase's 5.5 k clk)
  const int ksl = KS == 4 ? 2 : (KS == 2 ? 1 : 0), hpl = (NG == 4 ? 2 : (NG == 2 ? 1 : 0)) - ksl;
  const int HPU = 1 << hpl, nunits = njobsS << ksl;
  // unit = wave + 8 i: waves w and w + 4 share a SIMD, so a partial last round (12 units: waves 0-3) still gives every SIMD the
  // same number of units (dealing the extra units to the even waves instead was measured: slower, 54.1 -> 56.9 us backward)
  const int nmine = wave < nunits ? (nunits - wave + 7) >> 3 : 0;
  const int nh = nmine << hpl;
  if (nh == 0) return;
  auto unit_of = [&](const int i) { return wave + 8 * i; };
  auto req = [&](const int hj, float4(&fa)[4], float4(&fb)[4]) {
    const int u = unit_of(hj >> hpl), kbi = hj & (HPU - 1);
    const int job = u >> ksl, kq = u & (KS - 1), rt = job >= nctS ? 1 : 0, ct = job - rt * nctS;
    const int k0 = 64 * ((kq << hpl) + kbi) + 4 * g4;
    const float* Ap = As + (16 * rt + l15) * LDK + k0;
    const float* Bp = Ks + (16 * ct + l15) * LDK + k0;
#pragma unroll
    for (int st = 0; st < 4; ++st) { fa[st] = ld4(Ap + 16 * st); fb[st] = ld4(Bp + 16 * st); }
  };
  // TWO accumulator chains per wave (even / odd steps, added at the end of the unit): a wave that is alone on its SIMD - the
  // waves with the extra unit of a partial round - cannot issue DEPENDENT MFMAs back to back
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
  auto mul = [&](const int hj, const float4(&fa)[4], const float4(&fb)[4]) {
#pragma unroll
    for (int st = 0; st < 4; st += 2) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st].x, fb[st].x, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st + 1].x, fb[st + 1].x, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st].y, fb[st].y, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st + 1].y, fb[st + 1].y, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st].z, fb[st].z, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st + 1].z, fb[st + 1].z, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st].w, fb[st].w, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st + 1].w, fb[st + 1].w, acc2, 0, 0, 0);
    }
    const int ui = hj >> hpl;
    if ((hj & (HPU - 1)) == HPU - 1) {               // the unit's last k-block: its partial tile
      const int u = unit_of(ui), job = u >> ksl, kq = u & (KS - 1), rt = job >= nctS ? 1 : 0, ct = job - rt * nctS;
      float* Sp = Sc + kq * R * 68;
#pragma unroll
      for (int i = 0; i < 4; ++i) Sp[(16 * rt + 4 * g4 + i) * 68 + 16 * ct + l15] = (acc[i] + acc2[i]) * scale;
      acc = f32x4{0.f, 0.f, 0.f, 0.f};
      acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  float4 a0[4], b0[4], a1[4], b1[4];
  req(0, a0, b0);
  for (int hj = 0; hj < nh; hj += 2) {
    if (hj + 1 < nh) req(hj + 1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mul(hj, a0, b0);
    if (hj + 1 >= nh) break;
    if (hj + 2 < nh) req(hj + 2, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    mul(hj + 1, a1, b1);
  }
}

// kn: C[r][c] = sum_j A[r][j] B[j][c] (P.K, dS.K): A = [R][68] rows (j-contiguous), B = the key rows ([j][LDK], four ds_read_b32 per
// step), NkP / 16 steps of 16 keys per job; the 2 x H / 16 jobs are dealt over the 8 waves, output rows of pitch LDK.
template <int NG, int LDK, bool PIPE = true>
__device__ __forceinline__ void al_kn_phase(const float* __restrict__ Ss, const float* __restrict__ Ks, float* __restrict__ Os,
                                            const int nsteps, const int wave, const int l15, const int g4) {
  constexpr int nct = 4 * NG, njobs = (R / 16) * nct, NJ = njobs / 8;       // NJ = NG jobs per wave
  auto req = [&](const int i, float4(&fa)[4], float(&fb)[4][4]) {
    const int job = wave + 8 * i, rt = job / nct, ct = job - rt * nct;
    const float* Ap = Ss + (16 * rt + l15) * 68 + 4 * g4;
    const float* Bp = Ks + (4 * g4) * LDK + 16 * ct + l15;
#pragma unroll
    for (int st = 0; st < 4; ++st)
      if (st < nsteps) {
        fa[st] = ld4(Ap + 16 * st);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[st][j] = Bp[(16 * st + j) * LDK];
      }
  };
  auto mul = [&](const int i, const float4(&fa)[4], const float(&fb)[4][4]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < 4; ++st)
      if (st < nsteps) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st].x, fb[st][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st].y, fb[st][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st].z, fb[st][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st].w, fb[st][3], acc, 0, 0, 0);
      }
    const int job = wave + 8 * i, rt = job / nct, ct = job - rt * nct;
#pragma unroll
    for (int r = 0; r < 4; ++r) Os[(16 * rt + 4 * g4 + r) * LDK + 16 * ct + l15] = acc[r];
  };
  float4 a0[4], a1[4];
  float b0[4][4], b1[4][4];
  if constexpr (!PIPE) {               // (the backward kernel at hidden 256 has no registers for the second set)
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      req(i, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mul(i, a0, b0);
    }
    return;
  }
  req(0, a0, b0);
#pragma unroll
  for (int i = 0; i < NJ; i += 2) {
    if (i + 1 < NJ) req(i + 1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mul(i, a0, b0);
    if (i + 1 >= NJ) break;
    if (i + 2 < NJ) req(i + 2, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    mul(i + 1, a1, b1);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
template <int NG>
__global__ __launch_bounds__(512) void attn_al_fwd_kernel(const DosxAttn a, const AlGeo geo) {
  DOSX_SET_MAIN_PRIO();
  extern __shared__ __align__(16) float sm[];
  constexpr int H = 64 * NG, LDK = H + 4;        // (hidden = 64 NG exactly: every tile offset is an immediate)
  const int Nk = a.Nk, NkP = (Nk + 15) & ~15, Sq = a.Sq;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q16 = lane & 15, l15 = q16, g4 = lane >> 4;
  const int wpc = geo.wpc, KS = geo.KS;
  const int bq = blockIdx.x / wpc, w = blockIdx.x - bq * wpc, bk = bq % a.Bk;
  const int nk = a.key_ptr ? min(a.key_ptr[bk + 1] - a.key_ptr[bk], Nk) : Nk;      // keys this crystal attends over
  const int nqt = (Sq + R - 1) / R;
  float* Ks = sm;                      // [NkP][LDK]
  float* Qs = Ks + NkP * LDK;          // [R][LDK]   q o g0 -> O
  float* Sc = Qs + R * LDK;            // [KS][R][68] score partials; [0]: P o mask
  const float scale = rsqrtf((float)H), invH = 1.f / (float)H;
  const int lr = wave * 4 + g4;        // this quarter wave's row of the tile
  float4 g0[NG], b0[NG];
  bool on[NG];
#pragma unroll
  for (int k = 0; k < NG; ++k) {
    const int c = q16 * 4 + 64 * k;
    on[k] = c < H;
    g0[k] = ld4(a.gamma0 + (on[k] ? c : 0));
    b0[k] = ld4(a.beta0 + (on[k] ? c : 0));
  }
  auto load_x = [&](const int t, float4(&x)[NG]) {
    const int s = min(t * R + lr, Sq - 1);
    const float* xrow = a.x + ((size_t)s * a.q_stride_s + (size_t)bq * a.q_stride_b) * H;
#pragma unroll
    for (int k = 0; k < NG; ++k) x[k] = on[k] ? ld4(xrow + q16 * 4 + 64 * k) : f4zero();
  };
  ASTAMP(0);
  float4 xr[NG], xn[NG];
  load_x(w, xr);
  {   // the crystal's key rows -> Ks (rows beyond Nk zero), behind the first tile's rows
    const int h4 = H >> 2;
    float4 kr[2 * NG];                 // NkP * h4 <= 64 * 16 NG = 2 NG float4 per thread
#pragma unroll
    for (int i = 0; i < 2 * NG; ++i) {
      const int e = tid + 512 * i, j = e / h4, c = (e - j * h4) * 4;
      kr[i] = (e < NkP * h4 && j < Nk) ? ld4(a.kvhat + ((size_t)j * a.Bk + bk) * H + c) : f4zero();
    }
#pragma unroll
    for (int i = 0; i < 2 * NG; ++i) {
      const int e = tid + 512 * i, j = e / h4, c = (e - j * h4) * 4;
      if (e < NkP * h4) st4(Ks + j * LDK + c, kr[i]);
    }
  }
  const int nctS = NkP >> 4, njobsS = (R / 16) * nctS, klen = H / KS;
  for (int t = w; t < nqt; t += wpc) {
    const int s = min(t * R + lr, Sq - 1);
    const bool rv = (t * R + lr) < Sq;
    const size_t r = (size_t)s * a.Bq + bq;          // global row (valid memory also for the clamped duplicates)
    // ---- A: LayerNorm-0, the key gamma folded into q -> Qs ----
    float mean, rstd;
    {
      float u = 0.f;
#pragma unroll
      for (int k = 0; k < NG; ++k) u += (xr[k].x + xr[k].y) + (xr[k].z + xr[k].w);
      mean = row16_sum(u) * invH;
      u = 0.f;
#pragma unroll
      for (int k = 0; k < NG; ++k) {
        if (!on[k]) continue;
        const float p0 = xr[k].x - mean, p1 = xr[k].y - mean, p2 = xr[k].z - mean, p3 = xr[k].w - mean;
        u += (p0 * p0 + p1 * p1) + (p2 * p2 + p3 * p3);
      }
      rstd = rsqrtf(row16_sum(u) * invH + DOSX_LN_EPS);
#pragma unroll
      for (int k = 0; k < NG; ++k) {
        if (!on[k]) continue;
        const float4 v = xr[k];
        const float4 q = make_float4((v.x - mean) * rstd * g0[k].x + b0[k].x, (v.y - mean) * rstd * g0[k].y + b0[k].y,
                                     (v.z - mean) * rstd * g0[k].z + b0[k].z, (v.w - mean) * rstd * g0[k].w + b0[k].w);
        st4(Qs + lr * LDK + q16 * 4 + 64 * k, make_float4(q.x * g0[k].x, q.y * g0[k].y, q.z * g0[k].z, q.w * g0[k].w));
      }
    }
    if (t + wpc < nqt) load_x(t + wpc, xn);           // the next tile's rows: in flight under this tile's products
    const int so = t == w ? 0 : (t == w + wpc ? 16 : 48);     // (stamps of the first two tiles)
    ASTAMP(so + 1);
    __syncthreads();
    ASTAMP(so + 2);
    // ---- B: S = Qs . Ks^T (scaled); the K range split KS ways, partial tiles added in a fixed order by phase C ----
    al_kk_phase<NG, LDK>(Qs, Ks, Sc, nctS, njobsS, KS, scale, wave, l15, g4);
    ASTAMP(so + 3);
    __syncthreads();
    ASTAMP(so + 4);
    // ---- C: exact fp32 softmax of this quarter wave's row; P written out; P o mask -> Sc ----
    float psum = 1.f;
    {
      float* Sr = Sc + lr * 68;
      float v[4], mx = -INFINITY;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = q16 + 16 * jj;
        v[jj] = -INFINITY;
        if (j < nk) {                                  // (nk: the crystal's own key count with DosxAttn.key_ptr, else Nk)
          v[jj] = Sr[j];
          for (int kq = 1; kq < KS; ++kq) v[jj] += Sr[kq * R * 68 + j];
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
      const size_t prow = ((size_t)bq * Sq + s) * Nk;
      float ps = 0.f;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = q16 + 16 * jj;
        if (j >= NkP) continue;
        float pm = 0.f;
        if (j < Nk) {
          const float pr = v[jj] * inv;
          pm = a.drop_mask ? pr * a.drop_mask[prow + j] : pr;
          if (rv) a.probs[prow + j] = pr;              // the un-dropped P (the backward reads it)
        }
        Sr[j] = pm;                                    // (zeros beyond Nk: the padded keys of the second product)
        ps += pm;
      }
      psum = a.drop_mask ? row16_sum(ps) : 1.f;
    }
    ASTAMP(so + 5);
    __syncthreads();
    ASTAMP(so + 6);
    // ---- D: O = (P o mask) . Ks -> Qs ----
    al_kn_phase<NG, LDK>(Sc, Ks, Qs, NkP >> 4, wave, l15, g4);
    ASTAMP(so + 7);
    __syncthreads();
    ASTAMP(so + 8);
    // ---- E: x1 = O o g0 + b0 sum(P o mask) + x; both rows of statistics ----
    {
      float4 x1[NG];
      float u = 0.f;
#pragma unroll
      for (int k = 0; k < NG; ++k) {
        const int c = q16 * 4 + 64 * k;
        x1[k] = f4zero();
        if (!on[k]) continue;
        const float4 o = ld4(Qs + lr * LDK + c);
        x1[k] = make_float4(o.x * g0[k].x + b0[k].x * psum + xr[k].x, o.y * g0[k].y + b0[k].y * psum + xr[k].y,
                            o.z * g0[k].z + b0[k].z * psum + xr[k].z, o.w * g0[k].w + b0[k].w * psum + xr[k].w);
        if (rv) st4(a.out + r * H + c, x1[k]);
        u += (x1[k].x + x1[k].y) + (x1[k].z + x1[k].w);
      }
      if (a.out_stats) {
        const float mean1 = row16_sum(u) * invH;
        u = 0.f;
#pragma unroll
        for (int k = 0; k < NG; ++k) {
          if (!on[k]) continue;
          const float p0 = x1[k].x - mean1, p1 = x1[k].y - mean1, p2 = x1[k].z - mean1, p3 = x1[k].w - mean1;
          u += (p0 * p0 + p1 * p1) + (p2 * p2 + p3 * p3);
        }
        const float rstd1 = rsqrtf(row16_sum(u) * invH + DOSX_LN_EPS);
        if (rv && q16 == 0) { a.out_stats[2 * r] = mean1; a.out_stats[2 * r + 1] = rstd1; }
        if (a.ln1_out && rv) {         // the next LayerNorm on the row still in registers (DosxAttn.ln1_*)
#pragma unroll
          for (int k = 0; k < NG; ++k) {
            const int c = q16 * 4 + 64 * k;
            const float4 g1 = ld4(a.ln1_gamma + c), b1 = ld4(a.ln1_beta + c);
            st4(a.ln1_out + r * H + c,
                make_float4((x1[k].x - mean1) * rstd1 * g1.x + b1.x, (x1[k].y - mean1) * rstd1 * g1.y + b1.y,
                            (x1[k].z - mean1) * rstd1 * g1.z + b1.z, (x1[k].w - mean1) * rstd1 * g1.w + b1.w));
          }
        }
      }
      if (rv && q16 == 0) { a.qstats[2 * r] = mean; a.qstats[2 * r + 1] = rstd; }
    }
    // (no barrier: phase A of the next tile writes only this quarter wave's own row of Qs, which phase E has just read; the
    //  other waves read Qs / Sc again only behind the next tile's first barrier)
    ASTAMP(so + 9);
#pragma unroll
    for (int k = 0; k < NG; ++k) xr[k] = xn[k];
  }
  ASTAMP(63);
}

// ------------------------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------------------------
// Sum of the np = rep * wpc shares of key crystal bk's gradient (share (i, w) in slot ((bk + i Bk) nqt + w) of dkv_part), all
// <= 64 key rows at once (two per quarter-wave slot), key-side chain rule, dkvhat (+)=, and per group of 16 key rows the
// [dgamma | dbeta] partial row (slots added in row order: the layout dosx_attention_bwd's dkv_part path documents).
// Pp: [64][2 * 64 NG] floats of LDS.  Contains a barrier.
template <int NG>
__device__ __forceinline__ void al_dkv_reduce(const DosxAttn& a, const int nqt, const int wpc, const int bk, float* __restrict__ Pp,
                                              const int tid) {
  constexpr int HP = 64 * NG, U = NG <= 2 ? 4 : 2;
  const int lane = tid & 63, q16 = lane & 15, slot = tid >> 4;                 // 32 slots
  constexpr int H = 64 * NG;
  const int Nk = a.Nk, rep = a.Bq / a.Bk, ngroups = (Nk + 15) / 16;
  const int np = rep * wpc;
  const size_t pstride = (size_t)Nk * H;
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)a.dkv_part, 0, 0x7fffffff, 0x00020000);
  float4 g0[NG], d[2][NG], kh[2][NG], d0[2][NG];
  bool jv[2];
  size_t krow[2];
#pragma unroll
  for (int k = 0; k < NG; ++k) g0[k] = ld4(a.gamma0 + ((q16 * 4 + 64 * k) < H ? (q16 * 4 + 64 * k) : 0));
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int j = slot + 32 * p;
    jv[p] = j < Nk;
    krow[p] = ((size_t)(jv[p] ? j : 0) * a.Bk + bk) * H;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int c = q16 * 4 + 64 * k, cc = c < H ? c : 0;
      kh[p][k] = ld4(a.kvhat + krow[p] + cc);
      d0[p][k] = a.dkv_accumulate ? ld4(a.dkvhat + krow[p] + cc) : f4zero();
      d[p][k] = f4zero();
    }
  }
  for (int p0 = 0; p0 < np; p0 += U) {
    float4 v[2][U][NG];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pi = min(p0 + u, np - 1), i = pi / wpc, ww = pi - i * wpc;
        const size_t off = ((size_t)(bk + i * a.Bk) * nqt + ww) * pstride + (size_t)(jv[p] ? slot + 32 * p : 0) * H;
#pragma unroll
        for (int k = 0; k < NG; ++k) {
          const int c = q16 * 4 + 64 * k;
          v[p][u][k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rP, (uint32_t)((off + (c < H ? c : 0)) * 4), 0, 16));   // sc1
        }
      }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (p0 + u < np) {
#pragma unroll
          for (int k = 0; k < NG; ++k) d[p][k] = f4add(d[p][k], v[p][u][k]);
        }
  }
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int c = q16 * 4 + 64 * k;
      float4 pg = f4zero(), pb = f4zero();
      if (jv[p] && c < H) {
        pg = make_float4(d[p][k].x * kh[p][k].x, d[p][k].y * kh[p][k].y, d[p][k].z * kh[p][k].z, d[p][k].w * kh[p][k].w);
        pb = d[p][k];
        st4(a.dkvhat + krow[p] + c, make_float4(d[p][k].x * g0[k].x + d0[p][k].x, d[p][k].y * g0[k].y + d0[p][k].y,
                                                d[p][k].z * g0[k].z + d0[p][k].z, d[p][k].w * g0[k].w + d0[p][k].w));
      }
      st4(Pp + (slot + 32 * p) * 2 * HP + c, pg);
      st4(Pp + (slot + 32 * p) * 2 * HP + HP + c, pb);
    }
  __syncthreads();
  for (int o = tid; o < ngroups * 2 * H; o += 512) {
    const int grp = o / (2 * H), c = o - grp * 2 * H;
    const int col = (c / H) * HP + (c % H);
    float t = 0.f;
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) t += Pp[(grp * 16 + sl) * 2 * HP + col];
    a.partials_kv[((size_t)bk * ngroups + grp) * 2 * H + c] = t;
  }
}

template <int NG>
__global__ __launch_bounds__(512) void attn_al_bwd_kernel(const DosxAttn a, const AlGeo geo) {
  DOSX_SET_MAIN_PRIO();
  extern __shared__ __align__(16) float sm[];
  constexpr int HP = 64 * NG, NJW = 2 * NG;            // <= 4 x 4 NG key-tile x column-tile jobs of the share, 8 waves
  constexpr int H = 64 * NG, LDK = H + 4;        // (hidden = 64 NG exactly: every tile offset is an immediate)
  const int Nk = a.Nk, NkP = (Nk + 15) & ~15, Sq = a.Sq;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q16 = lane & 15, l15 = q16, g4 = lane >> 4;
  const int wpc = geo.wpc, KS = geo.KS;
  const int bq = blockIdx.x / wpc, w = blockIdx.x - bq * wpc, bk = bq % a.Bk;
  const int nqt = (Sq + R - 1) / R;
  float* Ks = sm;                      // [NkP][LDK]
  float* Os = Ks + NkP * LDK;          // [R][LDK]   dO rows
  float* Ds = Os + R * LDK;            // [R][LDK]   dO o g0 -> dq -> Ql = LN0(x) g0 + b0
  float* Sc = Ds + R * LDK;            // [2][R][68] dP partials -> [0]: dS (Ss), [1]: P o mask (Ps2)
  float* Ss = Sc;
  float* Ps2 = Sc + R * 68;
  const float scale = rsqrtf((float)H), invH = 1.f / (float)H;
  const int lr = wave * 4 + g4;
  float4 g0[NG], b0[NG];
  bool on[NG];
#pragma unroll
  for (int k = 0; k < NG; ++k) {
    const int c = q16 * 4 + 64 * k;
    on[k] = c < H;
    g0[k] = ld4(a.gamma0 + (on[k] ? c : 0));
    b0[k] = ld4(a.beta0 + (on[k] ? c : 0));
  }
  auto load_go = [&](const int t, float4(&go)[NG]) {
    const int s = min(t * R + lr, Sq - 1);
    const size_t orow = (size_t)s * a.Bq + bq;
#pragma unroll
    for (int k = 0; k < NG; ++k) go[k] = on[k] ? ld4(a.dout + orow * H + q16 * 4 + 64 * k) : f4zero();
  };
  ASTAMP(0);
  float4 ngo[NG];                      // the NEXT tile's dO row (requested one tile ahead)
  load_go(w, ngo);
  {   // the crystal's key rows -> Ks
    const int h4 = H >> 2;
    float4 kr[2 * NG];
#pragma unroll
    for (int i = 0; i < 2 * NG; ++i) {
      const int e = tid + 512 * i, j = e / h4, c = (e - j * h4) * 4;
      kr[i] = (e < NkP * h4 && j < Nk) ? ld4(a.kvhat + ((size_t)j * a.Bk + bk) * H + c) : f4zero();
    }
#pragma unroll
    for (int i = 0; i < 2 * NG; ++i) {
      const int e = tid + 512 * i, j = e / h4, c = (e - j * h4) * 4;
      if (e < NkP * h4) st4(Ks + j * LDK + c, kr[i]);
    }
  }
  const int nctS = NkP >> 4, njobsS = (R / 16) * nctS, klen = H / KS;
  constexpr int nct = H >> 4;
  f32x4 accK[NJW];
#pragma unroll
  for (int i = 0; i < NJW; ++i) accK[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Phase f / the publication: job i of this wave = key tile jt0 + JT(i), column tile ct0 + CT(i) with compile-time JT, CT, so
  // that every fragment address is one of FOUR lane bases + an immediate (run-time job coordinates made the compiler keep
  // ~250 precomputed addresses alive across the tile loop: 128 spilled registers at hidden 256)
  const int jt0 = nct >= 8 ? 0 : (wave >> 2), ct0 = nct >= 8 ? wave : (wave & 3);
  const float* pPs = Ps2 + (4 * g4) * 68 + 16 * jt0 + l15;
  const float* pSs = Ss + (4 * g4) * 68 + 16 * jt0 + l15;
  const float* pOs = Os + (4 * g4) * LDK + 16 * ct0 + l15;
  const float* pQl = Ds + (4 * g4) * LDK + 16 * ct0 + l15;
  float4 pg[NG], pb[NG];
#pragma unroll
  for (int k = 0; k < NG; ++k) { pg[k] = f4zero(); pb[k] = f4zero(); }

  for (int t = w; t < nqt; t += wpc) {
    const int s = min(t * R + lr, Sq - 1);
    const bool rv = (t * R + lr) < Sq;
    const size_t orow = (size_t)s * a.Bq + bq;
    // this row's softmax weights + dropout multipliers (phase c): requested now
    float pr[4], mk[4];
    {
      const size_t prow = ((size_t)bq * Sq + s) * Nk;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = q16 + 16 * jj, jc = j < Nk ? j : 0;
        pr[jj] = a.probs[prow + jc];
        mk[jj] = a.drop_mask ? a.drop_mask[prow + jc] : 1.f;
      }
    }
    // ---- a: dO -> Os (zeros beyond the data);  Ds = dO o g0;  cq = dO . b0 (dropout only) ----
    float cq = 0.f;
    {
      float u = 0.f;
#pragma unroll
      for (int k = 0; k < NG; ++k) {
        if (!on[k]) continue;
        const float4 go = rv ? ngo[k] : f4zero();
        st4(Os + lr * LDK + q16 * 4 + 64 * k, go);
        st4(Ds + lr * LDK + q16 * 4 + 64 * k, make_float4(go.x * g0[k].x, go.y * g0[k].y, go.z * g0[k].z, go.w * g0[k].w));
        u += (go.x * b0[k].x + go.y * b0[k].y) + (go.z * b0[k].z + go.w * b0[k].w);
      }
      if (a.drop_mask) cq = row16_sum(u);
    }
    if (t + wpc < nqt) load_go(t + wpc, ngo);
    const int so = t == w ? 0 : (t == w + wpc ? 16 : 48);
    ASTAMP(so + 1);
    __syncthreads();
    ASTAMP(so + 2);
    // ---- b: dP (up to a row constant) = Ds . Ks^T ----
    al_kk_phase<NG, LDK>(Ds, Ks, Sc, nctS, njobsS, KS, 1.f, wave, l15, g4);
    ASTAMP(so + 3);
    __syncthreads();
    ASTAMP(so + 4);
    // ---- c: dS = P o (dP' - sum_j P dP') scale -> Ss;  P' = P o M -> Ps2 (zeros beyond Nk / the data); in place, row-wise ----
    {
      float dp[4], dot = 0.f;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = q16 + 16 * jj;
        float u = 0.f;
        if (j < Nk) {
          u = Sc[lr * 68 + j];
          if (KS > 1) u += Sc[R * 68 + lr * 68 + j];
        }
        if (a.drop_mask) u = (u + cq) * mk[jj];
        if (j < Nk) dot += pr[jj] * u;
        dp[jj] = u;
      }
      dot = row16_sum(dot);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = q16 + 16 * jj;
        if (j >= NkP) continue;
        const bool v = j < Nk && rv;
        Ss[lr * 68 + j] = v ? pr[jj] * (dp[jj] - dot) * scale : 0.f;
        Ps2[lr * 68 + j] = v ? pr[jj] * mk[jj] : 0.f;
      }
    }
    // this row's query + LayerNorm-0 statistics (phase e): in flight under phase d
    float4 xr[NG];
    {
      const float* xrow = a.x + ((size_t)s * a.q_stride_s + (size_t)bq * a.q_stride_b) * H;
#pragma unroll
      for (int k = 0; k < NG; ++k) xr[k] = on[k] ? ld4(xrow + q16 * 4 + 64 * k) : f4zero();
    }
    const float mean = a.qstats[2 * orow], rstd = a.qstats[2 * orow + 1];
    ASTAMP(so + 5);
    __syncthreads();
    ASTAMP(so + 6);
    // ---- d: dq = dS . Ks -> Ds ----
    al_kn_phase<NG, LDK, (NG < 4)>(Ss, Ks, Ds, NkP >> 4, wave, l15, g4);
    ASTAMP(so + 7);
    __syncthreads();
    ASTAMP(so + 8);
    // ---- e: LayerNorm-0 backward on the query rows + residual -> dx; query-side dg0 / db0 (kept per row slot across the
    //         tiles); Ql = LN0(x) g0 + b0 over this quarter wave's own row of Ds ----
    {
      float4 d[NG], xh[NG];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < NG; ++k) {
        d[k] = f4zero(); xh[k] = f4zero();
        if (!(on[k] && rv)) continue;
        float4 dd = ld4(Ds + lr * LDK + q16 * 4 + 64 * k);
        dd = make_float4(dd.x * g0[k].x, dd.y * g0[k].y, dd.z * g0[k].z, dd.w * g0[k].w);
        const float4 xv = xr[k];
        const float4 h = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
        d[k] = dd; xh[k] = h;
        pg[k] = f4add(pg[k], make_float4(dd.x * h.x, dd.y * h.y, dd.z * h.z, dd.w * h.w));
        pb[k] = f4add(pb[k], dd);
        const float4 dh = make_float4(dd.x * g0[k].x, dd.y * g0[k].y, dd.z * g0[k].z, dd.w * g0[k].w);
        s1 += (dh.x + dh.y) + (dh.z + dh.w);
        s2 += (dh.x * h.x + dh.y * h.y) + (dh.z * h.z + dh.w * h.w);
      }
      s1 = row16_sum(s1) * invH; s2 = row16_sum(s2) * invH;
#pragma unroll
      for (int k = 0; k < NG; ++k) {
        if (!on[k]) continue;
        const float4 dd = d[k], h = xh[k];
        if (rv) {
          const float4 go = ld4(Os + lr * LDK + q16 * 4 + 64 * k);      // (this row's dO: the residual path)
          st4(a.dx + orow * H + q16 * 4 + 64 * k,
              make_float4(rstd * (dd.x * g0[k].x - s1 - h.x * s2) + go.x, rstd * (dd.y * g0[k].y - s1 - h.y * s2) + go.y,
                          rstd * (dd.z * g0[k].z - s1 - h.z * s2) + go.z, rstd * (dd.w * g0[k].w - s1 - h.w * s2) + go.w));
        }
        st4(Ds + lr * LDK + q16 * 4 + 64 * k,
            rv ? make_float4(h.x * g0[k].x + b0[k].x, h.y * g0[k].y + b0[k].y, h.z * g0[k].z + b0[k].z, h.w * g0[k].w + b0[k].w)
               : f4zero());
      }
    }
    ASTAMP(so + 9);
    __syncthreads();
    ASTAMP(so + 10);
    // ---- f: this workgroup's share of dK + dV += P'^T . dO + dS^T . Ql   ([NkP keys] x [R queries] . [R queries] x [H]) ----
#pragma unroll
    for (int i = 0; i < NJW; ++i) {
      // job i of this wave: key tile jt, column tile ct - jt a compile-time value (or one of two), ct = the wave's own column
      // tile(s), so that every fragment address is ONE lane base + an immediate (run-time job coordinates made the compiler
      // keep ~250 precomputed addresses alive across the tile loop: 128 spilled registers at hidden 256)
      const int JT = nct >= 8 ? i / (nct / 8) : 2 * i, CT = nct >= 8 ? 8 * (i % (nct / 8)) : 0;
      if (jt0 + JT >= nctS) continue;
      float pa[R / 4], sa[R / 4], b1[R / 4], b2[R / 4];
#pragma unroll
      for (int kk = 0; kk < R; kk += 16)
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
          const int o = (kk >> 2) + i2;
          pa[o] = pPs[(kk + i2) * 68 + 16 * JT];
          sa[o] = pSs[(kk + i2) * 68 + 16 * JT];
          b1[o] = pOs[(kk + i2) * LDK + 16 * CT];
          b2[o] = pQl[(kk + i2) * LDK + 16 * CT];
        }
      __builtin_amdgcn_sched_barrier(0);               // (all 32 fragments of the job requested before its first MFMA)
#pragma unroll
      for (int o = 0; o < R / 4; ++o) {
        accK[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[o], b1[o], accK[i], 0, 0, 0);
        accK[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[o], b2[o], accK[i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);               // (one job's 32 fragments at a time: the unrolled jobs must not be merged)
    }
    ASTAMP(so + 11);
    __syncthreads();                   // (the next tile's phase a overwrites Os / Ds)
    ASTAMP(so + 12);
  }
  ASTAMP(60);
  // ---- publish this workgroup's share (write-through) ----
  {
    float* part = a.dkv_part + ((size_t)bq * nqt + w) * (size_t)Nk * H;
#pragma unroll
    for (int i = 0; i < NJW; ++i) {
      const int jt = jt0 + (nct >= 8 ? i / (nct / 8) : 2 * i), ct = ct0 + (nct >= 8 ? 8 * (i % (nct / 8)) : 0);
      if (jt >= nctS) continue;
#pragma unroll
      for (int i2 = 0; i2 < 4; ++i2) {
        const int j = 16 * jt + 4 * g4 + i2, col = 16 * ct + l15;
        if (j < Nk) __hip_atomic_store(part + (size_t)j * H + col, accK[i][i2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1
      }
    }
  }
  // ---- the query-side [dgamma | dbeta] row of this workgroup: 32 row slots -> column sums; zero rows for its other tiles ----
  {
    float* Pp = sm;                    // [32][2][HP]  (everything else in LDS is dead: the loop ended with a barrier)
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int c = q16 * 4 + 64 * k;
      st4(Pp + lr * 2 * HP + c, pg[k]);
      st4(Pp + lr * 2 * HP + HP + c, pb[k]);
    }
    __syncthreads();
    float* prow = a.partials_q + ((size_t)bq * nqt + w) * 2 * H;
    for (int c = tid; c < 2 * H; c += 512) {
      const int o = (c / H) * HP + (c % H);
      float u = 0.f;
#pragma unroll
      for (int sl = 0; sl < 32; ++sl) u += Pp[sl * 2 * HP + o];
      prow[c] = u;
    }
    for (int t2 = w + wpc; t2 < nqt; t2 += wpc) {
      float* z = a.partials_q + ((size_t)bq * nqt + t2) * 2 * H;
      for (int c = tid; c < 2 * H; c += 512) z[c] = 0.f;
    }
  }
  // ---- ticket: the last arriving workgroup of key crystal bk finishes its key gradient ----
  ASTAMP(61);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* flag = reinterpret_cast<int*>(sm);
  const int arrivers = (a.Bq / a.Bk) * wpc;
  if (tid == 0) *flag = dosx_ticket(a.dkv_cnt + bk);
  __syncthreads();
  const bool last = *flag == arrivers - 1;
  __syncthreads();                     // (the flag word is about to be overwritten by the reduction's LDS rows)
  if (last) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    al_dkv_reduce<NG>(a, nqt, wpc, bk, sm, tid);
    if (tid == 0) __hip_atomic_store(a.dkv_cnt + bk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  ASTAMP(63);
}

int g_al_mode = -1;
inline int al_mode() {
  // DOSX_ATTN_ALIGNED: 0 = never, 1 = hidden > 128 only, 2 = every shape these kernels take (default: faster than attention.hip's
  // at every BASELINE shape with <= 64 keys, profiles/r05_kernel_microbench.log `attn`)
  if (g_al_mode < 0) {
    const char* e = getenv("DOSX_ATTN_ALIGNED");
    g_al_mode = e ? atoi(e) : 2;
  }
  return g_al_mode;
}

inline bool al_shape_ok(const DosxAttn& a) {
  const int m = al_mode();
  return m > 0 && a.flags == 0 && a.Nk >= 1 && a.Nk <= 64 && (a.H == 64 || a.H == 128 || a.H == 256) && (m > 1 || a.H > 128) &&
         a.qstats != nullptr;
}

inline AlGeo al_geo(const DosxAttn& a, const bool bwd) {
  const int nqt = ceil_div(a.Sq, R), NkP = (a.Nk + 15) & ~15;
  AlGeo g;
  g.wpc = 256 / a.Bq;
  if (g.wpc < 1) g.wpc = 1;
  if (g.wpc > nqt) g.wpc = nqt;
  const int njobsS = (R / 16) * (NkP >> 4);
  // K split of the score / dP jobs over otherwise idle waves, in whole k-blocks of 64 (KS divides NG = H / 64)
  int KS = njobsS >= 8 ? 1 : (njobsS >= 4 ? 2 : 4);
  if (bwd && KS > 2) KS = 2;                               // (two partial tiles of LDS in the backward kernel)
  const int ng = a.H / 64;
  while (KS > ng) KS >>= 1;
  g.KS = KS;
  return g;
}

inline size_t al_fwd_smem(const DosxAttn& a, const AlGeo& g) {
  const int NkP = (a.Nk + 15) & ~15, LDK = a.H + 4;
  return sizeof(float) * ((size_t)NkP * LDK + (size_t)R * LDK + (size_t)g.KS * R * 68);
}

inline size_t al_bwd_smem(const DosxAttn& a) {
  const int NkP = (a.Nk + 15) & ~15, LDK = a.H + 4, HP = 64 * ceil_div(a.H, 64);
  const size_t mainf = (size_t)NkP * LDK + 2 * (size_t)R * LDK + 2 * (size_t)R * 68, red = (size_t)64 * 2 * HP;
  return sizeof(float) * (mainf > red ? mainf : red);
}

}  // namespace

namespace dosx_detail {

// 1: launched; 0: not this shape (the caller runs attention.hip's kernels); < 0: error
int attn_aligned_fwd(const DosxAttn& a, hipStream_t st) {
  if (!al_shape_ok(a)) return 0;
  const AlGeo g = al_geo(a, false);
  const size_t smem = al_fwd_smem(a, g);
  if (smem > 160 * 1024) return 0;
  const dim3 grid(a.Bq * g.wpc);
#define DOSX_ALF(NG_)                                                                                                    \
  do {                                                                                                                   \
    static bool attr_set = false;                                                                                        \
    if (!attr_set) {                                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_al_fwd_kernel<NG_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr_set = true;                                                                                                   \
    }                                                                                                                    \
    hipLaunchKernelGGL((attn_al_fwd_kernel<NG_>), grid, dim3(512), smem, st, a, g);                                      \
  } while (0)
  switch (ceil_div(a.H, 64)) {
    case 1: DOSX_ALF(1); break;
    case 2: DOSX_ALF(2); break;
    default: DOSX_ALF(4);
  }
#undef DOSX_ALF
  DOSX_LAUNCH_CHECK();
  return 1;
}

int attn_aligned_bwd(const DosxAttn& a, hipStream_t st) {
  if (!al_shape_ok(a) || !a.dkv_part || !a.dkv_cnt) return 0;
  const AlGeo g = al_geo(a, true);
  if (g.KS > 2) return 0;
  const size_t smem = al_bwd_smem(a);
  if (smem > 160 * 1024 || (size_t)a.Bq * ceil_div(a.Sq, R) * a.Nk * a.H * 4 >= 0x7fffffffull) return 0;
  const dim3 grid(a.Bq * g.wpc);
#define DOSX_ALB(NG_)                                                                                                    \
  do {                                                                                                                   \
    static bool attr_set = false;                                                                                        \
    if (!attr_set) {                                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_al_bwd_kernel<NG_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr_set = true;                                                                                                   \
    }                                                                                                                    \
    hipLaunchKernelGGL((attn_al_bwd_kernel<NG_>), grid, dim3(512), smem, st, a, g);                                      \
  } while (0)
  switch (ceil_div(a.H, 64)) {
    case 1: DOSX_ALB(1); break;
    case 2: DOSX_ALB(2); break;
    default: DOSX_ALB(4);
  }
#undef DOSX_ALB
  DOSX_LAUNCH_CHECK();
  return 1;
}

}  // namespace dosx_detail

extern "C" int dosx_attention_aligned_mode(int mode) {
  const int prev = al_mode();
  if (mode >= 0) g_al_mode = mode > 2 ? 2 : mode;
  return prev;
}
