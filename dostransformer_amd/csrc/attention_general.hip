// dosx_attention_fwd / _bwd for MORE than 320 keys per crystal (layers/multihead_attention.py:62-74 and the attention half
// of layers/transformer.py:131-138 have no limit on the number of keys; a crystal of more than 320 atoms is a key set of
// that size, DOSTransformer_phonon.py:86-88).  The MFMA kernels of attention.hip keep the score row of a query in LDS, which
// is what limits them; these three kernels implement the SAME contract (include/dosx.h: DosxAttn - strided query rows, the
// RAW_Q / NO_RESIDUAL / DKV_HALF / DQ_HALF flags, qstats / out_stats, dropout multiplier mask, accumulate flag, partial-sum
// rows) with one wave per row, fp32 FMA chains in index order (deterministic) and no scratch beyond what the contract already
// hands over (probs, dscores).  A correctness path for rare shapes, not a hot path: every key row is re-read (and its affine
// re-applied) per query row from L2.  H <= 256: one float4 of the row per lane.
#include "common.h"

namespace {

constexpr int TR = 32;                   // rows per workgroup (8 waves x 4 rows): one partial-sum row per workgroup

__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4fma(float s, float4 a, float4 c) {
  return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}
__device__ __forceinline__ float f4dot(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 affine(float4 v, float4 g, float4 b) {
  return make_float4(fmaf(v.x, g.x, b.x), fmaf(v.y, g.y, b.y), fmaf(v.z, g.z, b.z), fmaf(v.w, g.w, b.w));
}
__device__ __forceinline__ float lane_bcast(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}

// the query row (s, bq) as the kernels use it: q = LN0(x) gamma0 + beta0, or x itself (RAW_Q); xhat for the LN0 backward
struct QRow { float4 x, xhat, q; float mean, rstd; };
__device__ __forceinline__ QRow load_q(const DosxAttn& a, int s, int bq, int c, bool on, float4 g, float4 b, bool use_saved_stats) {
  QRow r;
  const int H = a.H;
  r.x = on ? ld4(a.x + ((size_t)s * a.q_stride_s + (size_t)bq * a.q_stride_b) * H + c) : f4zero();
  r.mean = 0.f; r.rstd = 1.f; r.xhat = r.x; r.q = r.x;
  if (a.flags & DOSX_ATTN_RAW_Q) return r;
  if (use_saved_stats) {
    r.mean = a.qstats[2 * ((size_t)s * a.Bq + bq)];
    r.rstd = a.qstats[2 * ((size_t)s * a.Bq + bq) + 1];
  } else {
    r.mean = wave_sum(r.x.x + r.x.y + r.x.z + r.x.w) / (float)H;
    const float4 d = on ? make_float4(r.x.x - r.mean, r.x.y - r.mean, r.x.z - r.mean, r.x.w - r.mean) : f4zero();
    r.rstd = rsqrtf(wave_sum(f4dot(d, d)) / (float)H + DOSX_LN_EPS);
  }
  r.xhat = on ? make_float4((r.x.x - r.mean) * r.rstd, (r.x.y - r.mean) * r.rstd, (r.x.z - r.mean) * r.rstd, (r.x.w - r.mean) * r.rstd)
              : f4zero();
  r.q = on ? affine(r.xhat, g, b) : f4zero();
  return r;
}

// ---------------------------------------------------------------- forward: one wave per query row
__global__ __launch_bounds__(256) void attn_general_fwd_kernel(const DosxAttn a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = (int)blockIdx.x * 4 + wave, bq = (int)blockIdx.y, bk = bq % a.Bk;
  if (s >= a.Sq) return;
  const int H = a.H, Nk = a.Nk, c = lane * 4;
  const bool on = c < H;
  const float scale = 1.f / sqrtf((float)H);
  const float4 g = on ? ld4(a.gamma0 + c) : f4zero(), b = on ? ld4(a.beta0 + c) : f4zero();
  const QRow qr = load_q(a, s, bq, c, on, g, b, false);
  const size_t row = (size_t)s * a.Bq + bq;
  if (a.qstats && lane == 0) { a.qstats[2 * row] = qr.mean; a.qstats[2 * row + 1] = qr.rstd; }
  float* pr = a.probs + ((size_t)bq * a.Sq + s) * Nk;
  const float* mk = a.drop_mask ? a.drop_mask + ((size_t)bq * a.Sq + s) * Nk : nullptr;
  // scores: lane (j % 64) owns key j (it writes and re-reads pr[j] itself)
  float mymax = -INFINITY;
  for (int j = 0; j < Nk; ++j) {
    const float4 kh = on ? ld4(a.kvhat + ((size_t)j * a.Bk + bk) * H + c) : f4zero();
    const float t = wave_sum(on ? f4dot(qr.q, affine(kh, g, b)) : 0.f) * scale;
    if (lane == (j & 63)) { pr[j] = t; mymax = fmaxf(mymax, t); }
  }
  const float m = wave_max(mymax);
  float mysum = 0.f;
  for (int j = lane; j < Nk; j += 64) { const float e = expf(pr[j] - m); pr[j] = e; mysum += e; }
  const float inv = 1.f / wave_sum(mysum);
  for (int j = lane; j < Nk; j += 64) pr[j] *= inv;
  // out = sum_j (P o M)[j] K[j]  (+ x)
  float4 acc = f4zero();
  for (int j0 = 0; j0 < Nk; j0 += 64) {
    const int jl = j0 + lane;
    const float w = jl < Nk ? (mk ? pr[jl] * mk[jl] : pr[jl]) : 0.f;
    const int nj = min(64, Nk - j0);
    for (int jj = 0; jj < nj; ++jj) {
      const float wj = lane_bcast(w, jj);
      const float4 kh = on ? ld4(a.kvhat + ((size_t)(j0 + jj) * a.Bk + bk) * H + c) : f4zero();
      acc = f4fma(wj, on ? affine(kh, g, b) : f4zero(), acc);
    }
  }
  if (!(a.flags & DOSX_ATTN_NO_RESIDUAL)) acc = f4add(acc, qr.x);
  if (on) st4(a.out + row * H + c, acc);
  if (a.out_stats) {
    const float mean = wave_sum(on ? acc.x + acc.y + acc.z + acc.w : 0.f) / (float)H;
    const float4 d = on ? make_float4(acc.x - mean, acc.y - mean, acc.z - mean, acc.w - mean) : f4zero();
    const float var = wave_sum(f4dot(d, d)) / (float)H;
    if (lane == 0) { a.out_stats[2 * row] = mean; a.out_stats[2 * row + 1] = rsqrtf(var + DOSX_LN_EPS); }
  }
}

// ---------------------------------------------------------------- backward, query side: dS, dq, LN0 backward, dx
// One workgroup per 32 query rows of one batch entry (8 waves x 4 rows), ONE partial row [dgamma0 | dbeta0] per workgroup.
__global__ __launch_bounds__(512) void attn_general_dq_kernel(const DosxAttn a) {
  __shared__ float4 red[8][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bq = (int)blockIdx.y, bk = bq % a.Bk;
  const int H = a.H, Nk = a.Nk, c = lane * 4;
  const bool on = c < H;
  const float scale = 1.f / sqrtf((float)H);
  const float4 g = on ? ld4(a.gamma0 + c) : f4zero(), b = on ? ld4(a.beta0 + c) : f4zero();
  float4 pg = f4zero(), pb = f4zero();
  for (int i = 0; i < 4; ++i) {
    const int s = (int)blockIdx.x * TR + wave * 4 + i;
    if (s >= a.Sq) break;
    const QRow qr = load_q(a, s, bq, c, on, g, b, true);
    const size_t row = (size_t)s * a.Bq + bq;
    const float4 dout = on ? ld4(a.dout + row * H + c) : f4zero();
    const float* pr = a.probs + ((size_t)bq * a.Sq + s) * Nk;
    const float* mk = a.drop_mask ? a.drop_mask + ((size_t)bq * a.Sq + s) * Nk : nullptr;
    float* ds = a.dscores + ((size_t)bq * a.Sq + s) * Nk;
    // dP o M (owner lane keeps it in dscores), t = sum_j (dP o M)[j] P[j]
    float myt = 0.f;
    for (int j = 0; j < Nk; ++j) {
      const float4 kh = on ? ld4(a.kvhat + ((size_t)j * a.Bk + bk) * H + c) : f4zero();
      const float dp = wave_sum(on ? f4dot(dout, affine(kh, g, b)) : 0.f);
      if (lane == (j & 63)) {
        const float dpm = mk ? dp * mk[j] : dp;
        ds[j] = dpm;
        myt += dpm * pr[j];
      }
    }
    const float t = wave_sum(myt);
    for (int j = lane; j < Nk; j += 64) ds[j] = scale * pr[j] * (ds[j] - t);       // dS (multihead_attention.py:68-70 backwards)
    // dq = sum_j dS[j] K[j]
    float4 dq = f4zero();
    for (int j0 = 0; j0 < Nk; j0 += 64) {
      const int jl = j0 + lane;
      const float w = jl < Nk ? ds[jl] : 0.f;
      const int nj = min(64, Nk - j0);
      for (int jj = 0; jj < nj; ++jj) {
        const float wj = lane_bcast(w, jj);
        const float4 kh = on ? ld4(a.kvhat + ((size_t)(j0 + jj) * a.Bk + bk) * H + c) : f4zero();
        dq = f4fma(wj, on ? affine(kh, g, b) : f4zero(), dq);
      }
    }
    float4 dx = dq;
    if (!(a.flags & DOSX_ATTN_RAW_Q)) {                    // LayerNorm 0 backward on the query row
      pg = f4add(pg, f4mul(dq, qr.xhat));
      pb = f4add(pb, dq);
      const float4 dxh = f4mul(dq, g);
      const float m1 = wave_sum(on ? dxh.x + dxh.y + dxh.z + dxh.w : 0.f) / (float)H;
      const float m2 = wave_sum(on ? f4dot(dxh, qr.xhat) : 0.f) / (float)H;
      dx = make_float4(qr.rstd * (dxh.x - m1 - qr.xhat.x * m2), qr.rstd * (dxh.y - m1 - qr.xhat.y * m2),
                       qr.rstd * (dxh.z - m1 - qr.xhat.z * m2), qr.rstd * (dxh.w - m1 - qr.xhat.w * m2));
    }
    if (!(a.flags & DOSX_ATTN_NO_RESIDUAL)) dx = f4add(dx, dout);
    if (on) st4(a.dx + row * H + c, dx);
  }
  red[wave][0][lane] = pg;
  red[wave][1][lane] = pb;
  __syncthreads();
  if (wave == 0 && on) {
    float4 sg = red[0][0][lane], sb = red[0][1][lane];
    for (int w = 1; w < 8; ++w) { sg = f4add(sg, red[w][0][lane]); sb = f4add(sb, red[w][1][lane]); }
    float* p = a.partials_q + ((size_t)bq * gridDim.x + blockIdx.x) * 2 * H;
    st4(p + c, sg);
    st4(p + H + c, sb);
  }
}

// ---------------------------------------------------------------- backward, key side: dK + dV, dkvhat, key partials
// One workgroup per 32 keys of one crystal (8 waves x 4 keys), one partial row per workgroup.
__global__ __launch_bounds__(512) void attn_general_dkv_kernel(const DosxAttn a) {
  __shared__ float4 red[8][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bk = (int)blockIdx.y;
  const int H = a.H, Nk = a.Nk, c = lane * 4;
  const bool on = c < H;
  const float4 g = on ? ld4(a.gamma0 + c) : f4zero(), b = on ? ld4(a.beta0 + c) : f4zero();
  float4 pg = f4zero(), pb = f4zero();
  for (int i = 0; i < 4; ++i) {
    const int j = (int)blockIdx.x * TR + wave * 4 + i;
    if (j >= Nk) break;
    float4 dk = f4zero();
    for (int bq = bk; bq < a.Bq; bq += a.Bk)
      for (int s = 0; s < a.Sq; ++s) {
        const size_t pi = ((size_t)bq * a.Sq + s) * Nk + j;
        const float pm = a.drop_mask ? a.probs[pi] * a.drop_mask[pi] : a.probs[pi];
        const float dsj = a.dscores[pi];
        const QRow qr = load_q(a, s, bq, c, on, g, b, true);
        const float4 dout = on ? ld4(a.dout + ((size_t)s * a.Bq + bq) * H + c) : f4zero();
        dk = f4fma(pm, dout, f4fma(dsj, qr.q, dk));        // dV + dK: the keys ARE the values
      }
    const size_t kr = ((size_t)j * a.Bk + bk) * H + c;
    if (on) {
      const float4 kh = ld4(a.kvhat + kr);
      pg = f4add(pg, f4mul(dk, kh));
      pb = f4add(pb, dk);
      const float4 d = f4mul(dk, g);
      st4(a.dkvhat + kr, a.dkv_accumulate ? f4add(ld4(a.dkvhat + kr), d) : d);
    }
  }
  red[wave][0][lane] = pg;
  red[wave][1][lane] = pb;
  __syncthreads();
  if (wave == 0 && on) {
    float4 sg = red[0][0][lane], sb = red[0][1][lane];
    for (int w = 1; w < 8; ++w) { sg = f4add(sg, red[w][0][lane]); sb = f4add(sb, red[w][1][lane]); }
    float* p = a.partials_kv + ((size_t)bk * gridDim.x + blockIdx.x) * 2 * H;
    st4(p + c, sg);
    st4(p + H + c, sb);
  }
}

}  // namespace

namespace dosx_detail {

int attn_general_fwd(const DosxAttn& a, hipStream_t st) {
  hipLaunchKernelGGL(attn_general_fwd_kernel, dim3(ceil_div(a.Sq, 4), a.Bq), dim3(256), 0, st, a);
  DOSX_LAUNCH_CHECK();
  return 0;
}

int attn_general_bwd(const DosxAttn& a, hipStream_t st) {
  DOSX_CHECK_ARG(a.dscores, "dosx_attention_bwd: Nk=%d > 320 needs the dscores scratch [Bq, Sq, Nk]", a.Nk);
  DOSX_CHECK_ARG(!a.dkv_cnt, "dosx_attention_bwd: dkv_cnt (one-launch backward) is for Nk <= 64");
  if (!(a.flags & DOSX_ATTN_BWD_DKV_HALF)) {
    hipLaunchKernelGGL(attn_general_dq_kernel, dim3(ceil_div(a.Sq, TR), a.Bq), dim3(512), 0, st, a);
    DOSX_LAUNCH_CHECK();
  }
  if (!(a.flags & DOSX_ATTN_BWD_DQ_HALF)) {
    hipLaunchKernelGGL(attn_general_dkv_kernel, dim3(ceil_div(a.Nk, TR), a.Bk), dim3(512), 0, st, a);
    DOSX_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace dosx_detail
