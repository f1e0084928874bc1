// Small fused launches of round 6: (1) the backward of the two output heads, (2) the phonon edge encoder.
//
// (1) The backward of the two output heads in ONE launch - the three launch-latency-bound kernels between the self
// encoder's and the first encoder's backward (DOSTransformer_phonon.py:93-109 differentiated):
//
//     dpre[(s, bq)] = ( ddosin[(s, bq)] + rownorm_bwd(dkvs, kvs, rstd)[(s, bq)] ) * leaky_relu'(dosin[(s, bq)])      all S * 2B rows
//     dE1[(s, b)]   = dpre[(s, b)] . Wg[:, :H]  +  dpre[(s, B + b)] . Ws[:, :H]                                       S * B rows
//
// dkvs: the key-side gradient of the self attention (its keys are the NORMALISED head outputs kvs, rstd: DosxGemm.norm_out),
// ddosin: the query-side gradient, dosin = leaky_relu(fc / fc_prompt pre-activation), rows (s, bq) at s * 2B + bq with bq < B the
// global branch (fc: Wg = fc.weight [H, ldwg]) and bq >= B the system branch (fc_prompt: Ws [H, ldws]); E1 is the first H input
// columns of both heads.  What dosx_rownorm_bwd_act + two dosx_gemm launches (w_layout 1, row-mapped A, the second accumulating
// into the first) compute: dpre is written (the heads' weight gradients and the per-crystal row sums read it), dE1 is the sum of
// the two products in the order (global branch k = 0 .. H-1, then system branch) - one k-ordered MFMA chain per output element.
//
// One workgroup (4 waves) = 16 dE1 rows x 64 output columns: phase 0 computes the 32 dpre rows behind its 16 output rows (16
// lanes per row, two passes; column slice 0 writes them out) into an LDS tile [16][2H]; phase 1: every wave multiplies one
// 16-column tile over the whole K = 2H with its B fragments straight from global memory (the weights as stored: [k][n], one dword
// per MFMA, 64 in flight) - no LDS staging of weights, no k-split, no barrier in the product.
#include "common.h"

namespace {

template <int H>
__global__ __launch_bounds__(256) void heads_bwd_kernel(const DosxHeadsBwd a) {
  DOSX_SET_MAIN_PRIO();
  constexpr int K = 2 * H, LDA = K + 4, NG = H / 64 > 0 ? H / 64 : 1, NSL = H / 64;      // NSL column slices of 64
  constexpr int KCH = 256;                          // k range whose B fragments are in registers at once (64 dwords per lane)
  __shared__ __align__(16) float As[16 * LDA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  const int row = tid >> 4, q = tid & 15;
  const int tile = (int)blockIdx.x / NSL, sl = (int)blockIdx.x - tile * NSL;
  const int B = a.B, M = a.S * B;
  const int m0 = tile * 16;
  const int col0 = sl * 64 + wave * 16;             // this wave's 16 output columns
  // ---- B fragments of the first k range, requested before anything else ----
  float bq[KCH / 4];
  auto issue = [&](const int k0) {
#pragma unroll
    for (int s = 0; s < KCH / 16; ++s) {
      const int kk = k0 + 16 * s;                   // (compile-time after unrolling; H is a multiple of 16: a step never straddles the two heads)
      const float* w = kk < H ? a.wg + (size_t)(kk + 4 * g4) * a.ldwg : a.ws + (size_t)(kk - H + 4 * g4) * a.ldws;
      const int ld = kk < H ? a.ldwg : a.ldws;
#pragma unroll
      for (int i = 0; i < 4; ++i) bq[s * 4 + i] = kk < K ? w[(size_t)i * ld + col0 + l15] : 0.f;
    }
  };
  issue(0);
  // ---- phase 0: the 32 dpre rows of this tile (pass p = branch p), 16 lanes per row ----
  {
    const int gr = min(m0 + row, M - 1), s = gr / B, b = gr - s * B;
    const bool rv = m0 + row < M;
#pragma unroll 1
    for (int p = 0; p < 2; ++p) {
      const size_t r = (size_t)s * 2 * B + (size_t)p * B + b;
      float4 g[NG], xh[NG], d[NG], y[NG];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const int c = 4 * q + 64 * i;
        g[i] = ld4(a.dkvs + r * H + c); xh[i] = ld4(a.kvs + r * H + c);
        d[i] = ld4(a.ddosin + r * H + c); y[i] = ld4(a.dosin + r * H + c);
        s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
      }
      const float rs = a.rstd[r];
      const float m1 = row16_sum(s1) * (1.f / (float)H), m2 = row16_sum(s2) * (1.f / (float)H);
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const int c = 4 * q + 64 * i;
        const float4 t = make_float4(d[i].x + rs * (g[i].x - m1 - xh[i].x * m2), d[i].y + rs * (g[i].y - m1 - xh[i].y * m2),
                                     d[i].z + rs * (g[i].z - m1 - xh[i].z * m2), d[i].w + rs * (g[i].w - m1 - xh[i].w * m2));
        const float4 o = make_float4(y[i].x > 0.f ? t.x : a.slope * t.x, y[i].y > 0.f ? t.y : a.slope * t.y,
                                     y[i].z > 0.f ? t.z : a.slope * t.z, y[i].w > 0.f ? t.w : a.slope * t.w);
        st4(As + row * LDA + p * H + c, o);
        if (rv && sl == 0) st4(a.dpre + r * H + c, o);
      }
    }
  }
  __syncthreads();
  // ---- phase 1: dE1 tile = As [16][2H] . [Wg' ; Ws'] (this wave: 16 columns, the whole K) ----
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k0 = 0; k0 < K; k0 += KCH) {
    if (k0) issue(k0);
#pragma unroll
    for (int s = 0; s < KCH / 16; ++s) {
      if (k0 + 16 * s < K) {
        const float4 av = ld4(As + l15 * LDA + k0 + 16 * s + 4 * g4);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bq[s * 4 + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bq[s * 4 + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bq[s * 4 + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bq[s * 4 + 3], acc, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gr = m0 + 4 * g4 + r;
    if (gr < M) a.de1[(size_t)gr * a.ldde1 + col0 + l15] = acc[r];
  }
}

// The phonon EDGE ENCODER in one launch (round 6): SH(l <= 1)(v) * smooth_cutoff(|v| / r_max) features (DOSTransformer_phonon.py:74-77),
// the K = 4 Linear on them, PReLU, the second Linear (GN_encoder.edge_encoder, :129,142) - what dosx_edge_embed_sh1 followed by a
// dosx_gemm with the PReLU prologue compute.  16 edge rows x 64 output columns per workgroup: phase 0 makes the 16 pre-activation
// rows (the same k-ordered fma chain as edge_embed_kernel; column slice 0 writes attr and z - both are saved for the backward),
// PReLU into an LDS tile; phase 1: one 16-column tile per wave over K = H, B fragments (W2 [n][k]: float4 along k) from global.
__device__ __forceinline__ float smooth_cutoff_e(float x) {
  const float u = 2.f * (x - 1.f);
  float y = (1.f - cospif(u)) * 0.5f;
  if (u > 0.f) y = 0.f;
  if (u < -1.f) y = 1.f;
  return y;
}

template <int H>
__global__ __launch_bounds__(256) void edge_enc_fwd_kernel(const DosxEdgeEnc a) {
  DOSX_SET_MAIN_PRIO();
  constexpr int LDA = H + 4, NG = H / 64, NSL = H / 64, SK = H / 16;
  __shared__ __align__(16) float As[16 * LDA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  const int row = tid >> 4, q = tid & 15;
  const int tile = (int)blockIdx.x / NSL, sl = (int)blockIdx.x - tile * NSL;
  const int E = a.E, m0 = tile * 16;
  const int col0 = sl * 64 + wave * 16;
  float4 bw[SK];
  {
    const float* wp = a.w2 + (size_t)(col0 + l15) * H + 4 * g4;
#pragma unroll
    for (int s = 0; s < SK; ++s) bw[s] = ld4(wp + 16 * s);
  }
  const float b2v = a.b2[col0 + l15];
  {
    const int e = min(m0 + row, E - 1);
    const bool rv = m0 + row < E;
    const float x = a.vec[3 * (size_t)e], y = a.vec[3 * (size_t)e + 1], zz = a.vec[3 * (size_t)e + 2];
    const float alpha = *a.alpha;
    const float len = sqrtf(x * x + y * y + zz * zz);
    const float inv = 1.f / fmaxf(len, 1e-12f);
    const float cut = smooth_cutoff_e(len * a.inv_rmax);
    const float s3 = 1.7320508075688772f * cut;
    const float4 f = make_float4(cut, s3 * x * inv, s3 * y * inv, s3 * zz * inv);
    if (rv && q == 0 && sl == 0) st4(a.attr + 4 * (size_t)e, f);
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int c = 4 * q + 64 * i;
      float o[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float4 w = ld4(a.w0 + (size_t)(c + jj) * 4);
        float t = f.x * w.x;
        t = fmaf(f.y, w.y, t);
        t = fmaf(f.z, w.z, t);
        t = fmaf(f.w, w.w, t);
        o[jj] = t;
      }
      const float4 b = ld4(a.b0 + c);
      const float4 z4 = make_float4(o[0] + b.x, o[1] + b.y, o[2] + b.z, o[3] + b.w);
      if (rv && sl == 0) st4(a.z + (size_t)e * H + c, z4);
      st4(As + row * LDA + c, make_float4(z4.x >= 0.f ? z4.x : alpha * z4.x, z4.y >= 0.f ? z4.y : alpha * z4.y,
                                          z4.z >= 0.f ? z4.z : alpha * z4.z, z4.w >= 0.f ? z4.w : alpha * z4.w));
    }
  }
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < SK; ++s) {
    const float4 av = ld4(As + l15 * LDA + 16 * s + 4 * g4);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bw[s].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bw[s].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bw[s].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bw[s].w, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = m0 + 4 * g4 + r;
    if (e < E) a.out[(size_t)e * a.ldo + col0 + l15] = acc[r] + b2v;
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int dosx_edge_enc_supported(int H) { return H == 64 || H == 128; }

extern "C" int dosx_edge_enc_fwd(const DosxEdgeEnc* ap, dosx_stream_t stream) {
  DOSX_CHECK_ARG(ap != nullptr, "dosx_edge_enc_fwd: null descriptor");
  const DosxEdgeEnc& a = *ap;
  if (a.E <= 0) return 0;
  DOSX_CHECK_ARG(dosx_edge_enc_supported(a.H), "dosx_edge_enc_fwd: hidden %d unsupported (64, 128)", a.H);
  DOSX_CHECK_ARG(a.vec && a.w0 && a.b0 && a.alpha && a.w2 && a.b2 && a.attr && a.z && a.out, "dosx_edge_enc_fwd: null operand");
  DOSX_CHECK_ARG(aligned16(a.w0) && aligned16(a.b0) && aligned16(a.w2) && aligned16(a.attr) && aligned16(a.z) && a.ldo >= a.H,
                 "dosx_edge_enc_fwd: w0 / b0 / w2 / attr / z must be 16-byte aligned, ldo >= H");
  const dim3 grid(ceil_div(a.E, 16) * (a.H / 64));
  hipStream_t st = to_stream(stream);
  if (a.H == 64) hipLaunchKernelGGL(edge_enc_fwd_kernel<64>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(edge_enc_fwd_kernel<128>, grid, dim3(256), 0, st, a);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_heads_bwd_supported(int H) { return H == 64 || H == 128 || H == 256; }

extern "C" int dosx_heads_bwd(const DosxHeadsBwd* ap, dosx_stream_t stream) {
  DOSX_CHECK_ARG(ap != nullptr, "dosx_heads_bwd: null descriptor");
  const DosxHeadsBwd& a = *ap;
  if (a.S <= 0 || a.B <= 0) return 0;
  DOSX_CHECK_ARG(dosx_heads_bwd_supported(a.H), "dosx_heads_bwd: hidden %d unsupported (64, 128, 256)", a.H);
  DOSX_CHECK_ARG(a.dkvs && a.kvs && a.rstd && a.ddosin && a.dosin && a.dpre && a.wg && a.ws && a.de1, "dosx_heads_bwd: null operand");
  DOSX_CHECK_ARG(aligned16(a.dkvs) && aligned16(a.kvs) && aligned16(a.ddosin) && aligned16(a.dosin) && aligned16(a.dpre) && a.ldwg >= a.H &&
                     a.ldws >= a.H && a.ldde1 >= a.H, "dosx_heads_bwd: row operands must be 16-byte aligned, ldwg / ldws / ldde1 >= H");
  const dim3 grid(ceil_div(a.S * a.B, 16) * (a.H / 64));
  hipStream_t st = to_stream(stream);
  if (a.H == 64) hipLaunchKernelGGL(heads_bwd_kernel<64>, grid, dim3(256), 0, st, a);
  else if (a.H == 128) hipLaunchKernelGGL(heads_bwd_kernel<128>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(heads_bwd_kernel<256>, grid, dim3(256), 0, st, a);
  DOSX_LAUNCH_CHECK();
  return 0;
}
