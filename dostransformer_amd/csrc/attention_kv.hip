// Attention with K != V (layers/multihead_attention.py:49-76 accepts any key / value pair of equal shape; the encoder's
// embed dropout draws different masks for keys and values, transformer.py:61-68).  No reference CALL SITE does that - every
// one passes the same tensor (DOSTransformer_phonon.py:88,97,99) - so the MFMA kernels of attention.hip are built for K = V
// and this file supplies the general case as plain building blocks around them: the softmax weights still come from
// dosx_attention_fwd (run on the keys), and
//     out   = (P o M) . V                       dosx_attn_pv      (also dQ = dS . K)
//     dV    = (P o M)^T . dOut                  dosx_attn_tv      (also dK = dS^T . Q)
//     dPd   = dOut . V^T                        dosx_attn_dp
//     dS    = scale * P o (dP - rowsum(dP o P)),  dP = dPd o M     dosx_softmax_bwd
// One wave per output row, fp32 FMA chains in index order (deterministic); correctness path, not a hot path.
// Layouts as in dosx_attention_*: rows [S*Bq, H] with row (s, bq) at s*Bq + bq, keys / values [Nk*Bk, H] with row
// (j, bk) at j*Bk + bk, query batch entry bq reads crystal bq % Bk, weights [Bq, Sq, Nk].
#include "common.h"

namespace {

// out[(s,bq)] = sum_j A[bq,s,j] * (mask ? mask[bq,s,j] : 1) * V[(j, bq % Bk)]
__global__ __launch_bounds__(256) void attn_pv_kernel(const float* __restrict__ A, const float* __restrict__ mask,
                                                      const float* __restrict__ V, float* __restrict__ out, int Sq, int Bq,
                                                      int Nk, int Bk, int H) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)Sq * Bq) return;
  const int s = (int)(row / Bq), bq = (int)(row % Bq), bk = bq % Bk;
  const float* a = A + ((size_t)bq * Sq + s) * Nk;
  const float* m = mask ? mask + ((size_t)bq * Sq + s) * Nk : nullptr;
  for (int c = lane * 4; c < H; c += 256) {
    float4 acc = f4zero();
    for (int j = 0; j < Nk; ++j) {
      const float w = m ? a[j] * m[j] : a[j];
      const float4 v = ld4(V + ((size_t)j * Bk + bk) * H + c);
      acc = make_float4(fmaf(w, v.x, acc.x), fmaf(w, v.y, acc.y), fmaf(w, v.z, acc.z), fmaf(w, v.w, acc.w));
    }
    st4(out + (size_t)row * H + c, acc);
  }
}

// out[(j,bk)] (+)= sum_{i < Bq/Bk} sum_s A[bk + i*Bk, s, j] * (mask ...) * X[(s, bk + i*Bk)]
__global__ __launch_bounds__(256) void attn_tv_kernel(const float* __restrict__ A, const float* __restrict__ mask,
                                                      const float* __restrict__ X, float* __restrict__ out, int Sq, int Bq,
                                                      int Nk, int Bk, int H, int accumulate) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)Nk * Bk) return;
  const int j = (int)(row / Bk), bk = (int)(row % Bk);
  for (int c = lane * 4; c < H; c += 256) {
    float4 acc = accumulate ? ld4(out + (size_t)row * H + c) : f4zero();
    for (int bq = bk; bq < Bq; bq += Bk)
      for (int s = 0; s < Sq; ++s) {
        const size_t ai = ((size_t)bq * Sq + s) * Nk + j;
        const float w = mask ? A[ai] * mask[ai] : A[ai];
        const float4 x = ld4(X + ((size_t)s * Bq + bq) * H + c);
        acc = make_float4(fmaf(w, x.x, acc.x), fmaf(w, x.y, acc.y), fmaf(w, x.z, acc.z), fmaf(w, x.w, acc.w));
      }
    st4(out + (size_t)row * H + c, acc);
  }
}

// dP[bq,s,j] = X[(s,bq)] . V[(j, bq % Bk)]
__global__ __launch_bounds__(256) void attn_dp_kernel(const float* __restrict__ X, const float* __restrict__ V,
                                                      float* __restrict__ dP, int Sq, int Bq, int Nk, int Bk, int H) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)Sq * Bq) return;
  const int s = (int)(row / Bq), bq = (int)(row % Bq), bk = bq % Bk;
  float* o = dP + ((size_t)bq * Sq + s) * Nk;
  for (int j = 0; j < Nk; ++j) {
    float t = 0.f;
    for (int c = lane * 4; c < H; c += 256) {
      const float4 x = ld4(X + (size_t)row * H + c), v = ld4(V + ((size_t)j * Bk + bk) * H + c);
      t += x.x * v.x + x.y * v.y + x.z * v.z + x.w * v.w;
    }
    t = wave_sum(t);
    if (lane == 0) o[j] = t;
  }
}

// dS[r,:] = scale * P[r,:] o (dP[r,:] - sum_j dP[r,j] P[r,j]),  dP = dPd o mask     (softmax backward, multihead_attention.py:68-70)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ P, const float* __restrict__ mask,
                                                          const float* __restrict__ dPd, float* __restrict__ dS, long long rows,
                                                          int Nk, float scale) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const size_t o = (size_t)r * Nk;
  float t = 0.f;
  for (int j = lane; j < Nk; j += 64) t += (mask ? dPd[o + j] * mask[o + j] : dPd[o + j]) * P[o + j];
  t = wave_sum(t);
  for (int j = lane; j < Nk; j += 64) dS[o + j] = scale * P[o + j] * ((mask ? dPd[o + j] * mask[o + j] : dPd[o + j]) - t);
}

// P[r,:] = softmax_fp32(scale * S[r,:])   (multihead_attention.py:68-70 for ANY number of keys: dosx_attention_fwd keeps the
// score row of a query in LDS, Nk <= 320; with S from dosx_attn_dp this pair is the general form).  One wave per row.
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ S, float* __restrict__ P, long long rows, int Nk,
                                                          float scale) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const size_t o = (size_t)r * Nk;
  float m = -INFINITY;
  for (int j = lane; j < Nk; j += 64) m = fmaxf(m, scale * S[o + j]);
  m = wave_max(m);
  float t = 0.f;
  for (int j = lane; j < Nk; j += 64) {
    const float e = expf(scale * S[o + j] - m);
    P[o + j] = e;                                  // (the same lane re-reads it below)
    t += e;
  }
  t = wave_sum(t);
  const float inv = 1.f / t;
  for (int j = lane; j < Nk; j += 64) P[o + j] *= inv;
}

int check_kv(int Sq, int Bq, int Nk, int Bk, int H, const char* who) {
  DOSX_CHECK_ARG(Sq > 0 && Bq > 0 && Nk > 0 && Bk > 0 && Bq % Bk == 0 && H > 0 && (H & 3) == 0,
                 "%s: bad dims Sq=%d Bq=%d Nk=%d Bk=%d H=%d (H %% 4 == 0, Bq %% Bk == 0)", who, Sq, Bq, Nk, Bk, H);
  return 0;
}

}  // namespace

extern "C" int dosx_attn_pv(const float* A, const float* mask, const float* V, float* out, int Sq, int Bq, int Nk, int Bk,
                            int H, dosx_stream_t stream) {
  if (int rc = check_kv(Sq, Bq, Nk, Bk, H, "dosx_attn_pv")) return rc;
  DOSX_CHECK_ARG(A && V && out, "dosx_attn_pv: null operand");
  hipLaunchKernelGGL(attn_pv_kernel, dim3((unsigned)(((long long)Sq * Bq + 3) / 4)), dim3(256), 0, to_stream(stream), A, mask, V,
                     out, Sq, Bq, Nk, Bk, H);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_attn_tv(const float* A, const float* mask, const float* X, float* out, int Sq, int Bq, int Nk, int Bk,
                            int H, int accumulate, dosx_stream_t stream) {
  if (int rc = check_kv(Sq, Bq, Nk, Bk, H, "dosx_attn_tv")) return rc;
  DOSX_CHECK_ARG(A && X && out, "dosx_attn_tv: null operand");
  hipLaunchKernelGGL(attn_tv_kernel, dim3((unsigned)(((long long)Nk * Bk + 3) / 4)), dim3(256), 0, to_stream(stream), A, mask, X,
                     out, Sq, Bq, Nk, Bk, H, accumulate);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_attn_dp(const float* X, const float* V, float* dP, int Sq, int Bq, int Nk, int Bk, int H,
                            dosx_stream_t stream) {
  if (int rc = check_kv(Sq, Bq, Nk, Bk, H, "dosx_attn_dp")) return rc;
  DOSX_CHECK_ARG(X && V && dP, "dosx_attn_dp: null operand");
  hipLaunchKernelGGL(attn_dp_kernel, dim3((unsigned)(((long long)Sq * Bq + 3) / 4)), dim3(256), 0, to_stream(stream), X, V, dP,
                     Sq, Bq, Nk, Bk, H);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_softmax_bwd(const float* P, const float* mask, const float* dPd, float* dS, long long rows, int Nk,
                                float scale, dosx_stream_t stream) {
  if (rows <= 0) return 0;
  DOSX_CHECK_ARG(P && dPd && dS && Nk > 0, "dosx_softmax_bwd: bad args");
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, to_stream(stream), P, mask, dPd, dS,
                     rows, Nk, scale);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_softmax_fwd(const float* S, float* P, long long rows, int Nk, float scale, dosx_stream_t stream) {
  if (rows <= 0) return 0;
  DOSX_CHECK_ARG(S && P && Nk > 0 && S != P, "dosx_softmax_fwd: bad args");
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, to_stream(stream), S, P, rows, Nk, scale);
  DOSX_LAUNCH_CHECK();
  return 0;
}
