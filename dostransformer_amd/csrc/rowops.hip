// Row-wise kernels: LayerNorm (+backward), the fused final-LN + out_layer dot product, the
// callers' losses and the flat AdamW step (gfx950).  One wave per row, float4 lanes.
#include <math.h>

#include "common.h"

namespace {

// y = xhat*gamma + beta ; saves xhat and rstd
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        float* __restrict__ xhat, float* __restrict__ rstd_out,
                                                        int M, int H) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M) return;
  const float* row = x + (size_t)r * H;
  float s1 = 0.f;
  for (int c = lane * 4; c < H; c += 256) {
    const float4 v = ld4(row + c);
    s1 += v.x + v.y + v.z + v.w;
  }
  const float mean = wave_sum(s1) / (float)H;
  float s2 = 0.f;
  for (int c = lane * 4; c < H; c += 256) {
    const float4 v = ld4(row + c);
    const float a = v.x - mean, b = v.y - mean, c2 = v.z - mean, d = v.w - mean;
    s2 += a * a + b * b + c2 * c2 + d * d;
  }
  const float rstd = rsqrtf(wave_sum(s2) / (float)H + DOSX_LN_EPS);
  for (int c = lane * 4; c < H; c += 256) {
    const float4 v = ld4(row + c), g = ld4(gamma + c), b = ld4(beta + c);
    const float4 h = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
    if (xhat) st4(xhat + (size_t)r * H + c, h);
    st4(y + (size_t)r * H + c, make_float4(h.x * g.x + b.x, h.y * g.y + b.y, h.z * g.z + b.z, h.w * g.w + b.w));
  }
  if (lane == 0 && rstd_out) rstd_out[r] = rstd;
}

// LN-backward row kernels.  NV = number of H-wide partial vectors per workgroup (2: dgamma,dbeta ; 3: + dw),
// plus an optional trailing scalar (db).  Workgroup = 4 waves x 8 rows = 32 rows; partial row layout
// [v0(H) | v1(H) | (v2(H)) | (scalar)].  One QUARTER WAVE per row, 2 passes of 4 rows per wave, everything in
// registers: a reduction over 4 rows is 4 DPP instructions (row16_sum), and the column partial sums are
// register accumulators written once (the first version walked its 8 rows one after the other with in-loop
// global loads, two 64-lane reductions and LDS read-modify-writes per row: 12-15 us at 3-6k rows).
template <bool ROWDOT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy,     // [M,H]   (!ROWDOT)
                                                     const float* __restrict__ ddos,   // [Bq,S]  (ROWDOT)
                                                     const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ w, float* __restrict__ dx,
                                                     float* __restrict__ partials, int M, int H, int S, int Bq) {
  extern __shared__ __align__(16) float sm[];   // [16 slots][NV*H] + [16]
  constexpr int NV = ROWDOT ? 3 : 2;
  constexpr int KB = 4;                          // 64-column blocks per row (H <= 256)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q16 = lane & 15;
  const int pld = NV * H + (ROWDOT ? 1 : 0);
  const float invH = 1.f / (float)H;
  float4 g[KB], bt[KB], ww[KB], pg[KB], pb[KB], pw[KB];
#pragma unroll
  for (int k = 0; k < KB; ++k) {
    const int c = q16 * 4 + 64 * k, cc = c < H ? c : 0;
    g[k] = ld4(gamma + cc);
    bt[k] = ROWDOT ? ld4(beta + cc) : f4zero();
    ww[k] = ROWDOT ? ld4(w + cc) : f4zero();
    pg[k] = f4zero(); pb[k] = f4zero(); pw[k] = f4zero();
  }
  float4 xh[2][KB], d[2][KB];
  float rs[2], dyr[2], s1[2], s2[2];
  float db = 0.f;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int r = blockIdx.x * 32 + wave * 8 + p * 4 + (lane >> 4);
    const bool rv = r < M;
    const int rc = rv ? r : M - 1;
    rs[p] = rstd[rc];
    dyr[p] = 0.f;
    if (ROWDOT) {
      dyr[p] = rv ? ddos[(size_t)(rc % Bq) * S + (rc / Bq)] : 0.f;
      if (q16 == 0) db += dyr[p];
    }
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int c = q16 * 4 + 64 * k, cc = c < H ? c : 0;
      xh[p][k] = ld4(xhat + (size_t)rc * H + cc);
      if (ROWDOT) d[p][k] = make_float4(dyr[p] * ww[k].x, dyr[p] * ww[k].y, dyr[p] * ww[k].z, dyr[p] * ww[k].w);
      else d[p][k] = ld4(dy + (size_t)rc * H + cc);
      if (!(rv && c < H)) { d[p][k] = f4zero(); xh[p][k] = f4zero(); }
    }
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    s1[p] = 0.f; s2[p] = 0.f;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const float4 dd = d[p][k], h = xh[p][k];
      if (ROWDOT) {
        pw[k].x += dyr[p] * (h.x * g[k].x + bt[k].x); pw[k].y += dyr[p] * (h.y * g[k].y + bt[k].y);
        pw[k].z += dyr[p] * (h.z * g[k].z + bt[k].z); pw[k].w += dyr[p] * (h.w * g[k].w + bt[k].w);
      }
      pg[k].x += dd.x * h.x; pg[k].y += dd.y * h.y; pg[k].z += dd.z * h.z; pg[k].w += dd.w * h.w;
      pb[k] = f4add(pb[k], dd);
      const float4 dh = make_float4(dd.x * g[k].x, dd.y * g[k].y, dd.z * g[k].z, dd.w * g[k].w);
      s1[p] += (dh.x + dh.y) + (dh.z + dh.w);
      s2[p] += (dh.x * h.x + dh.y * h.y) + (dh.z * h.z + dh.w * h.w);
    }
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) { s1[p] = row16_sum(s1[p]) * invH; s2[p] = row16_sum(s2[p]) * invH; }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int r = blockIdx.x * 32 + wave * 8 + p * 4 + (lane >> 4);
    if (r >= M) continue;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int c = q16 * 4 + 64 * k;
      if (c >= H) continue;
      const float4 dd = d[p][k], h = xh[p][k];
      st4(dx + (size_t)r * H + c,
          make_float4(rs[p] * (dd.x * g[k].x - s1[p] - h.x * s2[p]), rs[p] * (dd.y * g[k].y - s1[p] - h.y * s2[p]),
                      rs[p] * (dd.z * g[k].z - s1[p] - h.z * s2[p]), rs[p] * (dd.w * g[k].w - s1[p] - h.w * s2[p])));
    }
  }
  // column partial sums: one slot per quarter wave (16), summed in a fixed order
  const int slot = wave * 4 + (lane >> 4);
  float* my = sm + (size_t)slot * (NV * H);
#pragma unroll
  for (int k = 0; k < KB; ++k) {
    const int c = q16 * 4 + 64 * k;
    if (c >= H) continue;
    st4(my + c, pg[k]);
    st4(my + H + c, pb[k]);
    if (ROWDOT) st4(my + 2 * H + c, pw[k]);
  }
  if (ROWDOT && q16 == 0) sm[16 * NV * H + slot] = db;
  __syncthreads();
  float* prow = partials + (size_t)blockIdx.x * pld;
  for (int c = threadIdx.x; c < NV * H; c += 256) {
    float t = 0.f;
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) t += sm[sl * NV * H + c];
    prow[c] = t;
  }
  if (ROWDOT && threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) t += sm[16 * NV * H + sl];
    prow[NV * H] = t;
  }
}

// ---- rows wider than 256 floats (hidden > 256; the LayerNorm of the GNN blocks spans 2 x hidden) --------------------
// The same three backward passes for ANY row width up to 1024: one wave per row, a lane owns the float4 columns
// lane*4 + 256 k (k < 4), a workgroup (4 waves) walks its 32 rows 8 per wave and leaves ONE partial row like the
// register-resident kernel above: the column sums of a wave are register accumulators, combined over the 4 waves in a
// fixed order through LDS.  A correctness path: these widths are outside every BASELINE configuration.
//   MODE 0: dx = LN_bwd(dy)                                         partial row [dgamma(W) | dbeta(W)]
//   MODE 1: dy[r] = ddos[r % Bq][r / Bq] * w (ln_rowdot backward)   partial row [dgamma | dbeta | dw | db]
//   MODE 2: dy is the gradient BEHIND the PReLU that follows the LayerNorm (DOSTransformer_phonon.py:193,204
//           `LayerNorm -> PReLU`): y = xhat*gamma+beta, dy *= (y < 0 ? alpha : 1), dalpha += dy*y where y < 0 - what the
//           EPI_PRELU_LN_BWD epilogue of dosx_gemm does for rows of up to 512 floats.  partial row [dgamma | dbeta | pad(3) | dalpha]
constexpr int LNW_K = 4;
template <int MODE, int KN = LNW_K>
__global__ __launch_bounds__(256) void ln_bwd_wide_kernel(const float* __restrict__ dy, const float* __restrict__ ddos,
                                                          const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ w, const float* __restrict__ alpha,
                                                          float* __restrict__ dx, float* __restrict__ partials, int M, int W,
                                                          int S, int Bq, const int* __restrict__ dyidx = nullptr,
                                                          const float* __restrict__ dyscale = nullptr) {
  // dyidx / dyscale (MODE 2): row r reads dy[dyidx[r]] * dyscale[dyidx[r]] - the gradient behind the PReLU is one row per
  // destination NODE (the last message-passing layer, whose second Linear runs on the aggregated rows)
  extern __shared__ __align__(16) float sm[];   // [4 waves][NV*W] + [4]
  constexpr int NV = MODE == 1 ? 3 : 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pld = MODE == 1 ? 3 * W + 1 : (MODE == 2 ? 2 * W + 4 : 2 * W);
  const float invW = 1.f / (float)W;
  const float al = MODE == 2 ? *alpha : 0.f;
  float4 g[KN], bt[KN], ww[KN], pg[KN], pb[KN], pw[KN];
#pragma unroll
  for (int k = 0; k < KN; ++k) {
    const int c = lane * 4 + 256 * k, cc = c < W ? c : 0;
    g[k] = ld4(gamma + cc);
    bt[k] = MODE != 0 ? ld4(beta + cc) : f4zero();
    ww[k] = MODE == 1 ? ld4(w + cc) : f4zero();
    pg[k] = f4zero(); pb[k] = f4zero(); pw[k] = f4zero();
  }
  float psc = 0.f;                              // db (MODE 1) / dalpha (MODE 2)
  int gi[8];                                    // gathered rows: indices and scales of this wave's 8 rows, fetched up front
  float gs[8];
  if (MODE == 2 && dyidx) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = min(blockIdx.x * 32 + wave * 8 + i, M - 1);
      gi[i] = dyidx[r];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) gs[i] = dyscale ? dyscale[gi[i]] : 1.f;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = blockIdx.x * 32 + wave * 8 + i;
    if (r >= M) break;                          // (wave-uniform)
    const float rs = rstd[r];
    float dyr = 0.f;
    if (MODE == 1) {
      dyr = ddos[(size_t)(r % Bq) * S + (r / Bq)];
      if (lane == 0) psc += dyr;
    }
    float4 xh[KN], d[KN];
    float s1 = 0.f, s2 = 0.f;
    const size_t dr = (MODE == 2 && dyidx) ? (size_t)gi[i] : (size_t)r;
    const float dsc = (MODE == 2 && dyidx) ? gs[i] : 1.f;
#pragma unroll
    for (int k = 0; k < KN; ++k) {
      const int c = lane * 4 + 256 * k;
      xh[k] = f4zero(); d[k] = f4zero();
      if (c >= W) continue;
      xh[k] = ld4(xhat + (size_t)r * W + c);
      if (MODE == 1) d[k] = make_float4(dyr * ww[k].x, dyr * ww[k].y, dyr * ww[k].z, dyr * ww[k].w);
      else {
        d[k] = ld4(dy + dr * W + c);
        if (MODE == 2 && dyidx) d[k] = make_float4(d[k].x * dsc, d[k].y * dsc, d[k].z * dsc, d[k].w * dsc);
      }
      const float4 h = xh[k];
      if (MODE == 2) {
        const float y0 = h.x * g[k].x + bt[k].x, y1 = h.y * g[k].y + bt[k].y, y2 = h.z * g[k].z + bt[k].z,
                    y3 = h.w * g[k].w + bt[k].w;
        if (y0 < 0.f) { psc += d[k].x * y0; d[k].x *= al; }
        if (y1 < 0.f) { psc += d[k].y * y1; d[k].y *= al; }
        if (y2 < 0.f) { psc += d[k].z * y2; d[k].z *= al; }
        if (y3 < 0.f) { psc += d[k].w * y3; d[k].w *= al; }
      }
      const float4 dd = d[k];
      if (MODE == 1) {
        pw[k].x += dyr * (h.x * g[k].x + bt[k].x); pw[k].y += dyr * (h.y * g[k].y + bt[k].y);
        pw[k].z += dyr * (h.z * g[k].z + bt[k].z); pw[k].w += dyr * (h.w * g[k].w + bt[k].w);
      }
      pg[k].x += dd.x * h.x; pg[k].y += dd.y * h.y; pg[k].z += dd.z * h.z; pg[k].w += dd.w * h.w;
      pb[k] = f4add(pb[k], dd);
      const float4 dh = make_float4(dd.x * g[k].x, dd.y * g[k].y, dd.z * g[k].z, dd.w * g[k].w);
      s1 += (dh.x + dh.y) + (dh.z + dh.w);
      s2 += (dh.x * h.x + dh.y * h.y) + (dh.z * h.z + dh.w * h.w);
    }
    s1 = wave_sum(s1) * invW;
    s2 = wave_sum(s2) * invW;
#pragma unroll
    for (int k = 0; k < KN; ++k) {
      const int c = lane * 4 + 256 * k;
      if (c >= W) continue;
      const float4 dd = d[k], h = xh[k];
      st4(dx + (size_t)r * W + c,
          make_float4(rs * (dd.x * g[k].x - s1 - h.x * s2), rs * (dd.y * g[k].y - s1 - h.y * s2),
                      rs * (dd.z * g[k].z - s1 - h.z * s2), rs * (dd.w * g[k].w - s1 - h.w * s2)));
    }
  }
  float* my = sm + (size_t)wave * (NV * W);
#pragma unroll
  for (int k = 0; k < KN; ++k) {
    const int c = lane * 4 + 256 * k;
    if (c >= W) continue;
    st4(my + c, pg[k]);
    st4(my + W + c, pb[k]);
    if (MODE == 1) st4(my + 2 * W + c, pw[k]);
  }
  if (MODE != 0) {
    const float t = wave_sum(psc);
    if (lane == 0) sm[4 * NV * W + wave] = t;
  }
  __syncthreads();
  float* prow = partials + (size_t)blockIdx.x * pld;
  for (int c = threadIdx.x; c < NV * W; c += 256)
    prow[c] = ((sm[c] + sm[NV * W + c]) + sm[2 * NV * W + c]) + sm[3 * NV * W + c];
  if (MODE != 0 && threadIdx.x == 0)
    prow[pld - 1] = ((sm[4 * NV * W] + sm[4 * NV * W + 1]) + sm[4 * NV * W + 2]) + sm[4 * NV * W + 3];
}

// MODE 2 of the kernel above for rows of up to 512 floats, built to RUN NEXT TO a weight-gradient group: those workgroups hold
// a CU's LDS (2 x 75 KB) and 448 of the 512 vector registers of every SIMD lane for hundreds of microseconds, and a kernel
// that does not fit into what is left - one wave of <= 64 registers per SIMD, < 10 KB of LDS - waits for an EMPTY CU (the
// general kernel, 98 registers + 16 KB: 97 us in the Electron-DOS step for 19 us of work).  Memory-bound row kernels and the
// matrix waves want different pipes, so co-residency is not zero-sum here.  64 registers (gamma / beta re-read per row from
// L1 instead of held), 4 KB of LDS (the four waves add their column sums one after the other into ONE row).
__global__ __launch_bounds__(256, 8) void ln_prelu_bwd_lean_kernel(const float* __restrict__ dy, const int* __restrict__ dyidx,
                                                                   const float* __restrict__ dyscale,
                                                                   const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   const float* __restrict__ alpha, float* __restrict__ dx,
                                                                   float* __restrict__ partials, int M, int W) {
  extern __shared__ __align__(16) float sm[];   // [2 * W] + [4]
  constexpr int KN = 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pld = 2 * W + 4;
  const float invW = 1.f / (float)W;
  const float al = *alpha;
  float4 pg[KN], pb[KN];
#pragma unroll
  for (int k = 0; k < KN; ++k) { pg[k] = f4zero(); pb[k] = f4zero(); }
  float psc = 0.f;
  const int r0 = blockIdx.x * 32 + wave * 8;
  for (int i = 0; i < 8; ++i) {
    const int r = r0 + i;
    if (r >= M) break;                          // (wave-uniform)
    const size_t dr = dyidx ? (size_t)dyidx[r] : (size_t)r;
    const float dsc = dyscale ? dyscale[dr] : 1.f;
    const float rs = rstd[r];
    float4 xh[KN], d[KN];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < KN; ++k) {
      const int c = lane * 4 + 256 * k;
      xh[k] = f4zero(); d[k] = f4zero();
      if (c >= W) continue;
      xh[k] = ld4(xhat + (size_t)r * W + c);
      d[k] = ld4(dy + dr * W + c);
      const float4 g = ld4(gamma + c), bt = ld4(beta + c), h = xh[k];
      d[k] = make_float4(d[k].x * dsc, d[k].y * dsc, d[k].z * dsc, d[k].w * dsc);
      const float y0 = h.x * g.x + bt.x, y1 = h.y * g.y + bt.y, y2 = h.z * g.z + bt.z, y3 = h.w * g.w + bt.w;
      if (y0 < 0.f) { psc += d[k].x * y0; d[k].x *= al; }
      if (y1 < 0.f) { psc += d[k].y * y1; d[k].y *= al; }
      if (y2 < 0.f) { psc += d[k].z * y2; d[k].z *= al; }
      if (y3 < 0.f) { psc += d[k].w * y3; d[k].w *= al; }
      const float4 dd = d[k];
      pg[k].x += dd.x * h.x; pg[k].y += dd.y * h.y; pg[k].z += dd.z * h.z; pg[k].w += dd.w * h.w;
      pb[k] = f4add(pb[k], dd);
      d[k] = make_float4(dd.x * g.x, dd.y * g.y, dd.z * g.z, dd.w * g.w);      // dy * gamma
      s1 += (d[k].x + d[k].y) + (d[k].z + d[k].w);
      s2 += (d[k].x * h.x + d[k].y * h.y) + (d[k].z * h.z + d[k].w * h.w);
    }
    s1 = wave_sum(s1) * invW;
    s2 = wave_sum(s2) * invW;
#pragma unroll
    for (int k = 0; k < KN; ++k) {
      const int c = lane * 4 + 256 * k;
      if (c >= W) continue;
      const float4 dh = d[k], h = xh[k];
      st4(dx + (size_t)r * W + c, make_float4(rs * (dh.x - s1 - h.x * s2), rs * (dh.y - s1 - h.y * s2),
                                              rs * (dh.z - s1 - h.z * s2), rs * (dh.w - s1 - h.w * s2)));
    }
  }
  // column sums: wave 0 stores, waves 1..3 add in order (fixed summation order), then all write the partial row
  const float pt = wave_sum(psc);
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int k = 0; k < KN; ++k) {
        const int c = lane * 4 + 256 * k;
        if (c >= W) continue;
        if (w == 0) { st4(sm + c, pg[k]); st4(sm + W + c, pb[k]); }
        else { st4(sm + c, f4add(ld4(sm + c), pg[k])); st4(sm + W + c, f4add(ld4(sm + W + c), pb[k])); }
      }
      if (lane == 0) sm[2 * W] = (w == 0 ? 0.f : sm[2 * W]) + pt;
    }
    __syncthreads();
  }
  float* prow = partials + (size_t)blockIdx.x * pld;
  for (int c = threadIdx.x; c < 2 * W; c += 256) prow[c] = sm[c];
  if (threadIdx.x == 0) prow[pld - 1] = sm[2 * W];
}

// y[r] = (LN(x[r])*gamma+beta) . w + b   ->  dos[(r % Bq)*S + r / Bq]
__global__ __launch_bounds__(256) void ln_rowdot_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ xhat,
                                                        float* __restrict__ rstd_out, float* __restrict__ dos, int S,
                                                        int Bq, int H) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= S * Bq) return;
  const float* row = x + (size_t)r * H;
  float s1 = 0.f;
  for (int c = lane * 4; c < H; c += 256) {
    const float4 v = ld4(row + c);
    s1 += v.x + v.y + v.z + v.w;
  }
  const float mean = wave_sum(s1) / (float)H;
  float s2 = 0.f;
  for (int c = lane * 4; c < H; c += 256) {
    const float4 v = ld4(row + c);
    const float a = v.x - mean, bb = v.y - mean, c2 = v.z - mean, d = v.w - mean;
    s2 += a * a + bb * bb + c2 * c2 + d * d;
  }
  const float rstd = rsqrtf(wave_sum(s2) / (float)H + DOSX_LN_EPS);
  float dot = 0.f;
  for (int c = lane * 4; c < H; c += 256) {
    const float4 v = ld4(row + c), g = ld4(gamma + c), bt = ld4(beta + c), ww = ld4(w + c);
    const float4 h = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
    st4(xhat + (size_t)r * H + c, h);
    dot += (h.x * g.x + bt.x) * ww.x + (h.y * g.y + bt.y) * ww.y + (h.z * g.z + bt.z) * ww.z + (h.w * g.w + bt.w) * ww.w;
  }
  dot = wave_sum(dot);
  if (lane == 0) {
    rstd_out[r] = rstd;
    dos[(size_t)(r % Bq) * S + (r / Bq)] = dot + b[0];
  }
}

// plain y[r] = x[r] . w + b  (x already activated)
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ dos, int S,
                                                     int Bq, int H) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= S * Bq) return;
  float dot = 0.f;
  for (int c = lane * 4; c < H; c += 256) {
    const float4 v = ld4(x + (size_t)r * H + c), ww = ld4(w + c);
    dot += v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
  }
  dot = wave_sum(dot);
  if (lane == 0) dos[(size_t)(r % Bq) * S + (r / Bq)] = dot + b[0];
}

// dx[r] = ddos * w ; partial rows [dw(H) | db]
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float* __restrict__ ddos, const float* __restrict__ x,
                                                         const float* __restrict__ w, float* __restrict__ dx,
                                                         float* __restrict__ partials, int S, int Bq, int H) {
  extern __shared__ __align__(16) float sm[];   // [4][H+1]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int M = S * Bq;
  float* my = sm + (size_t)wave * (H + 1);
  for (int c = lane; c < H + 1; c += 64) my[c] = 0.f;
  float db = 0.f;
  for (int i = 0; i < 8; ++i) {
    const int r = blockIdx.x * 32 + wave * 8 + i;
    if (r >= M) break;
    const float dyr = ddos[(size_t)(r % Bq) * S + (r / Bq)];
    db += dyr;
    for (int c = lane * 4; c < H; c += 256) {
      const float4 v = ld4(x + (size_t)r * H + c), ww = ld4(w + c);
      my[c] += dyr * v.x; my[c + 1] += dyr * v.y; my[c + 2] += dyr * v.z; my[c + 3] += dyr * v.w;
      st4(dx + (size_t)r * H + c, make_float4(dyr * ww.x, dyr * ww.y, dyr * ww.z, dyr * ww.w));
    }
  }
  if (lane == 0) my[H] = db;
  __syncthreads();
  float* prow = partials + (size_t)blockIdx.x * (H + 1);
  for (int c = threadIdx.x; c < H + 1; c += 256)
    prow[c] = sm[c] + sm[(H + 1) + c] + sm[2 * (H + 1) + c] + sm[3 * (H + 1) + c];
}

// ---- losses -------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sse2_kernel(const float* __restrict__ pg, const float* __restrict__ ps,
                                                    const float* __restrict__ y, float* __restrict__ sse, int count) {
  __shared__ float red[2][16];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < count; i += 1024) {
    const float t = y[i], d0 = pg[i] - t, d1 = ps[i] - t;
    a += d0 * d0;
    b += d1 * d1;
  }
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = a; red[1][wave] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s0 = 0.f, s1 = 0.f;
    for (int i = 0; i < 16; ++i) { s0 += red[0][i]; s1 += red[1][i]; }
    sse[0] = s0;
    sse[1] = s1;
  }
}

__global__ void loss_phonon_bwd_kernel(const float* __restrict__ pg, const float* __restrict__ ps,
                                       const float* __restrict__ y, const float* __restrict__ sse, float beta,
                                       float inv_count, float* __restrict__ dpg, float* __restrict__ dps,
                                       float* __restrict__ loss, int count) {
  const float r0 = sqrtf(sse[0] * inv_count), r1 = sqrtf(sse[1] * inv_count);
  const float k0 = inv_count / r0, k1 = beta * inv_count / r1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && loss) loss[0] = r0 + beta * r1;
  if (i < count) {
    const float t = y[i];
    dpg[i] = (pg[i] - t) * k0;
    dps[i] = (ps[i] - t) * k1;
  }
}

// Single-process form of the two kernels above in ONE launch (one workgroup: the gradient needs the global SSE pair
// first; B*S is a few thousand elements).  Same summation order as sse2_kernel -> bitwise the two-phase result.
__global__ __launch_bounds__(1024) void loss_phonon_fused_kernel(const float* __restrict__ pg, const float* __restrict__ ps,
                                                                 const float* __restrict__ y, float* __restrict__ sse,
                                                                 float beta, float inv_count, float* __restrict__ dpg,
                                                                 float* __restrict__ dps, float* __restrict__ loss, int count) {
  __shared__ float red[2][16];
  __shared__ float tot[2];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < count; i += 1024) {
    const float t = y[i], d0 = pg[i] - t, d1 = ps[i] - t;
    a += d0 * d0;
    b += d1 * d1;
  }
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = a; red[1][wave] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s0 = 0.f, s1 = 0.f;
    for (int i = 0; i < 16; ++i) { s0 += red[0][i]; s1 += red[1][i]; }
    tot[0] = s0; tot[1] = s1;
    if (sse) { sse[0] = s0; sse[1] = s1; }
  }
  __syncthreads();
  const float r0 = sqrtf(tot[0] * inv_count), r1 = sqrtf(tot[1] * inv_count);
  const float k0 = inv_count / r0, k1 = beta * inv_count / r1;
  if (threadIdx.x == 0 && loss) loss[0] = r0 + beta * r1;
  for (int i = threadIdx.x; i < count; i += 1024) {
    const float t = y[i];
    dpg[i] = (pg[i] - t) * k0;
    dps[i] = (ps[i] - t) * k1;
  }
}

// one wave per crystal
__global__ __launch_bounds__(256) void loss_edos_kernel(const float* __restrict__ pg, const float* __restrict__ ps,
                                                        const float* __restrict__ y_ft, float beta, int B, int S,
                                                        float inv_bglobal, float* __restrict__ dpg,
                                                        float* __restrict__ dps, float* __restrict__ loss_partial) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const size_t o = (size_t)b * S;
  float a = 0.f, c = 0.f;
  for (int s = lane; s < S; s += 64) {
    const float t = fmaxf(y_ft[o + s], 0.f);   // torch.where(y < 0, 0, y)
    const float d0 = t - pg[o + s], d1 = t - ps[o + s];
    a += d0 * d0;
    c += d1 * d1;
  }
  a = wave_sum(a);
  c = wave_sum(c);
  const float r0 = sqrtf(a / (float)S), r1 = sqrtf(c / (float)S);
  const float k0 = inv_bglobal / ((float)S * r0), k1 = beta * inv_bglobal / ((float)S * r1);
  for (int s = lane; s < S; s += 64) {
    const float t = fmaxf(y_ft[o + s], 0.f);
    dpg[o + s] = (pg[o + s] - t) * k0;
    dps[o + s] = (ps[o + s] - t) * k1;
  }
  if (lane == 0) loss_partial[b] = (r0 + beta * r1) * inv_bglobal;
}

// dst[0] = sum(src[0..n))  (one block; deterministic order)
__global__ __launch_bounds__(256) void sum_kernel(const float* __restrict__ src, int n, float* __restrict__ dst) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += src[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) dst[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, size_t n, float decay, float beta1, float beta2, float eps,
                             float step_size, float inv_bc2s, float gscale) {
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 pp = ld4(p + 4 * i), gg = ld4(g + 4 * i), mm = ld4(m + 4 * i), vv = ld4(v + 4 * i);
    float* P = &pp.x; float* G = &gg.x; float* Mm = &mm.x; float* V = &vv.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gr = G[j] * gscale;
      float pj = P[j] * decay;
      const float mj = Mm[j] + (gr - Mm[j]) * (1.f - beta1);
      const float vj = V[j] * beta2 + (1.f - beta2) * gr * gr;
      const float denom = sqrtf(vj) * inv_bc2s + eps;
      pj -= step_size * (mj / denom);
      P[j] = pj; Mm[j] = mj; V[j] = vj;
    }
    st4(p + 4 * i, pp); st4(m + 4 * i, mm); st4(v + 4 * i, vv);
  }
  // tail (n not a multiple of 4)
  const size_t t0 = n4 << 2;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid < n - t0) {
    const size_t i = t0 + tid;
    const float gr = g[i] * gscale;
    float pj = p[i] * decay;
    const float mj = m[i] + (gr - m[i]) * (1.f - beta1);
    const float vj = v[i] * beta2 + (1.f - beta2) * gr * gr;
    pj -= step_size * (mj / (sqrtf(vj) * inv_bc2s + eps));
    p[i] = pj; m[i] = mj; v[i] = vj;
  }
}

// Philox4x32-10 (Salmon et al., SC'11): counter-based, so element i's draw depends only on (seed, stream_id, i)
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ mask, long long n, float p, float keep_scale,
                                                           const unsigned long long* __restrict__ seed_dev,
                                                           unsigned long long stream_id) {
  const unsigned long long seed = *seed_dev;
  const long long nblk = (n + 3) / 4;
  for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += (long long)gridDim.x * blockDim.x) {
    uint32_t c[4] = {(uint32_t)b, (uint32_t)((unsigned long long)b >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long i = 4 * b + e;
      if (i < n) mask[i] = ((float)(c[e] >> 8) * (1.f / 16777216.f) >= p) ? keep_scale : 0.f;
    }
  }
}

// Batched device-to-device copy: the jobs travel as a kernel argument (like the slab reduction's); blocks are mapped to
// (job, 4 KiB slice) through the prefix array.  One launch moves every field + index array of a batch into its shape
// bucket's static buffers (train._Slot.load: 12 hipMemcpyAsync of ~4.3 us each before).
constexpr int CP_MAX_JOBS = 24;
constexpr int CP_SLICE = 1024;      // dwords per block (256 lanes x uint4)
struct CopyLaunch {
  DosxCopyJob job[CP_MAX_JOBS];
  int first_block[CP_MAX_JOBS + 1];
  int n;
};

__global__ __launch_bounds__(256) void copy_many_kernel(const CopyLaunch L) {
  int lo = 0, hi = L.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (L.first_block[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const DosxCopyJob j = L.job[lo];
  const long long i = ((long long)((int)blockIdx.x - L.first_block[lo]) * CP_SLICE) + threadIdx.x * 4;
  const uint32_t* s = reinterpret_cast<const uint32_t*>(j.src);
  uint32_t* d = reinterpret_cast<uint32_t*>(j.dst);
  const bool vec = ((reinterpret_cast<uintptr_t>(j.src) | reinterpret_cast<uintptr_t>(j.dst)) & 15) == 0;
  if (vec && i + 4 <= j.dwords) {
    *reinterpret_cast<uint4*>(d + i) = *reinterpret_cast<const uint4*>(s + i);
  } else {
    for (int e = 0; e < 4; ++e)
      if (i + e < j.dwords) d[i + e] = s[i + e];
  }
}

}  // namespace

#define CHECK_H4(H) DOSX_CHECK_ARG((H) > 0 && ((H) & 3) == 0, "%s: H=%d must be a multiple of 4", __func__, (H))

extern "C" int dosx_layernorm(const float* x, const float* gamma, const float* beta, float* y, float* xhat, float* rstd,
                              int M, int H, dosx_stream_t stream) {
  if (M <= 0) return 0;
  CHECK_H4(H);
  DOSX_CHECK_ARG(x && gamma && beta && y, "dosx_layernorm: bad args");
  hipLaunchKernelGGL(layernorm_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, to_stream(stream), x, gamma, beta, y, xhat, rstd,
                     M, H);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dx,
                                  float* partials, int M, int H, dosx_stream_t stream) {
  if (M <= 0) return 0;
  CHECK_H4(H);
  DOSX_CHECK_ARG(dy && xhat && rstd && gamma && dx && partials, "dosx_layernorm_bwd: bad args");
  if (H > 256) {                     // wide rows (hidden > 256): the one-wave-per-row kernel
    DOSX_CHECK_ARG(H <= 256 * LNW_K, "dosx_layernorm_bwd: H=%d > %d", H, 256 * LNW_K);
    hipLaunchKernelGGL((ln_bwd_wide_kernel<0>), dim3(ceil_div(M, 32)), dim3(256), (4 * (size_t)(2 * H) + 4) * sizeof(float),
                       to_stream(stream), dy, nullptr, xhat, rstd, gamma, nullptr, nullptr, nullptr, dx, partials, M, H, 0, 1);
    DOSX_LAUNCH_CHECK();
    return 0;
  }
  const size_t smem = (16 * (size_t)(2 * H) + 16) * sizeof(float);
  hipLaunchKernelGGL((ln_bwd_kernel<false>), dim3(ceil_div(M, 32)), dim3(256), smem, to_stream(stream), dy, nullptr, xhat,
                     rstd, gamma, nullptr, nullptr, dx, partials, M, H, 0, 1);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_ln_rowdot(const float* x, const float* gamma, const float* beta, const float* w, const float* b,
                              float* xhat, float* rstd, float* dos, int S, int Bq, int H, dosx_stream_t stream) {
  if (S * Bq <= 0) return 0;
  CHECK_H4(H);
  DOSX_CHECK_ARG(x && gamma && beta && w && b && xhat && rstd && dos, "dosx_ln_rowdot: bad args");
  hipLaunchKernelGGL(ln_rowdot_kernel, dim3(ceil_div(S * Bq, 4)), dim3(256), 0, to_stream(stream), x, gamma, beta, w, b,
                     xhat, rstd, dos, S, Bq, H);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_ln_rowdot_bwd(const float* ddos, const float* xhat, const float* rstd, const float* gamma,
                                  const float* beta, const float* w, float* dx, float* partials, int S, int Bq, int H,
                                  dosx_stream_t stream) {
  const int M = S * Bq;
  if (M <= 0) return 0;
  CHECK_H4(H);
  DOSX_CHECK_ARG(ddos && xhat && rstd && gamma && beta && w && dx && partials, "dosx_ln_rowdot_bwd: bad args");
  if (H > 256) {
    DOSX_CHECK_ARG(H <= 256 * LNW_K, "dosx_ln_rowdot_bwd: H=%d > %d", H, 256 * LNW_K);
    hipLaunchKernelGGL((ln_bwd_wide_kernel<1>), dim3(ceil_div(M, 32)), dim3(256), (4 * (size_t)(3 * H) + 4) * sizeof(float),
                       to_stream(stream), nullptr, ddos, xhat, rstd, gamma, beta, w, nullptr, dx, partials, M, H, S, Bq);
    DOSX_LAUNCH_CHECK();
    return 0;
  }
  const size_t smem = (16 * (size_t)(3 * H) + 16) * sizeof(float);
  hipLaunchKernelGGL((ln_bwd_kernel<true>), dim3(ceil_div(M, 32)), dim3(256), smem, to_stream(stream), nullptr, ddos, xhat,
                     rstd, gamma, beta, w, dx, partials, M, H, S, Bq);
  DOSX_LAUNCH_CHECK();
  return 0;
}

static bool ln_lean_on() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("DOSX_LN_LEAN");
    on = e ? atoi(e) : 1;
  }
  return on != 0;
}

extern "C" int dosx_ln_prelu_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma,
                                 const float* beta, const float* alpha, float* dz, float* partials, int M, int W,
                                 dosx_stream_t stream) {
  if (M <= 0) return 0;
  CHECK_H4(W);
  DOSX_CHECK_ARG(dy && xhat && rstd && gamma && beta && alpha && dz && partials, "dosx_ln_prelu_bwd: bad args");
  DOSX_CHECK_ARG(W <= 256 * LNW_K, "dosx_ln_prelu_bwd: row width %d > %d", W, 256 * LNW_K);
  if (W <= 512 && ln_lean_on())
    hipLaunchKernelGGL(ln_prelu_bwd_lean_kernel, dim3(ceil_div(M, 32)), dim3(256), ((size_t)(2 * W) + 4) * sizeof(float),
                       to_stream(stream), dy, nullptr, nullptr, xhat, rstd, gamma, beta, alpha, dz, partials, M, W);
  else
    hipLaunchKernelGGL((ln_bwd_wide_kernel<2>), dim3(ceil_div(M, 32)), dim3(256), (4 * (size_t)(2 * W) + 4) * sizeof(float),
                       to_stream(stream), dy, nullptr, xhat, rstd, gamma, beta, nullptr, alpha, dz, partials, M, W, 0, 1);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_ln_prelu_bwd_gather(const float* dy, const int32_t* idx, const float* scale, const float* xhat,
                                        const float* rstd, const float* gamma, const float* beta, const float* alpha, float* dz,
                                        float* partials, int M, int W, dosx_stream_t stream) {
  if (M <= 0) return 0;
  CHECK_H4(W);
  DOSX_CHECK_ARG(dy && idx && xhat && rstd && gamma && beta && alpha && dz && partials, "dosx_ln_prelu_bwd_gather: bad args");
  DOSX_CHECK_ARG(W <= 256 * LNW_K, "dosx_ln_prelu_bwd_gather: row width %d > %d", W, 256 * LNW_K);
  if (W <= 512 && ln_lean_on())
    hipLaunchKernelGGL(ln_prelu_bwd_lean_kernel, dim3(ceil_div(M, 32)), dim3(256), ((size_t)(2 * W) + 4) * sizeof(float),
                       to_stream(stream), dy, idx, scale, xhat, rstd, gamma, beta, alpha, dz, partials, M, W);
  else if (W <= 512)        // (two float4 column groups per lane: half the registers of the general form)
    hipLaunchKernelGGL((ln_bwd_wide_kernel<2, 2>), dim3(ceil_div(M, 32)), dim3(256), (4 * (size_t)(2 * W) + 4) * sizeof(float),
                       to_stream(stream), dy, nullptr, xhat, rstd, gamma, beta, nullptr, alpha, dz, partials, M, W, 0, 1, idx, scale);
  else
    hipLaunchKernelGGL((ln_bwd_wide_kernel<2>), dim3(ceil_div(M, 32)), dim3(256), (4 * (size_t)(2 * W) + 4) * sizeof(float),
                       to_stream(stream), dy, nullptr, xhat, rstd, gamma, beta, nullptr, alpha, dz, partials, M, W, 0, 1, idx, scale);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_rowdot(const float* x, const float* w, const float* b, float* dos, int S, int Bq, int H,
                           dosx_stream_t stream) {
  if (S * Bq <= 0) return 0;
  CHECK_H4(H);
  DOSX_CHECK_ARG(x && w && b && dos, "dosx_rowdot: bad args");
  hipLaunchKernelGGL(rowdot_kernel, dim3(ceil_div(S * Bq, 4)), dim3(256), 0, to_stream(stream), x, w, b, dos, S, Bq, H);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_rowdot_bwd(const float* ddos, const float* x, const float* w, float* dx, float* partials, int S,
                               int Bq, int H, dosx_stream_t stream) {
  const int M = S * Bq;
  if (M <= 0) return 0;
  CHECK_H4(H);
  DOSX_CHECK_ARG(ddos && x && w && dx && partials, "dosx_rowdot_bwd: bad args");
  const size_t smem = 4 * (size_t)(H + 1) * sizeof(float);
  hipLaunchKernelGGL(rowdot_bwd_kernel, dim3(ceil_div(M, 32)), dim3(256), smem, to_stream(stream), ddos, x, w, dx, partials,
                     S, Bq, H);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_sse2(const float* pg, const float* ps, const float* y, float* sse, int count, dosx_stream_t stream) {
  DOSX_CHECK_ARG(pg && ps && y && sse && count > 0, "dosx_sse2: bad args");
  hipLaunchKernelGGL(sse2_kernel, dim3(1), dim3(1024), 0, to_stream(stream), pg, ps, y, sse, count);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_loss_phonon_bwd(const float* pg, const float* ps, const float* y, const float* sse, float beta,
                                    double count_global, float* dpg, float* dps, float* loss, int count,
                                    dosx_stream_t stream) {
  DOSX_CHECK_ARG(pg && ps && y && sse && dpg && dps && count > 0 && count_global > 0, "dosx_loss_phonon_bwd: bad args");
  hipLaunchKernelGGL(loss_phonon_bwd_kernel, dim3(ceil_div(count, 256)), dim3(256), 0, to_stream(stream), pg, ps, y, sse,
                     beta, (float)(1.0 / count_global), dpg, dps, loss, count);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_loss_phonon(const float* pg, const float* ps, const float* y, float* sse, float beta, float* dpg,
                                float* dps, float* loss, int count, dosx_stream_t stream) {
  DOSX_CHECK_ARG(pg && ps && y && dpg && dps && count > 0, "dosx_loss_phonon: bad args");
  hipLaunchKernelGGL(loss_phonon_fused_kernel, dim3(1), dim3(1024), 0, to_stream(stream), pg, ps, y, sse, beta,
                     (float)(1.0 / (double)count), dpg, dps, loss, count);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_loss_edos(const float* pg, const float* ps, const float* y_ft, float beta, int B, int S, int B_global,
                              float* dpg, float* dps, float* loss_partial, dosx_stream_t stream) {
  DOSX_CHECK_ARG(pg && ps && y_ft && dpg && dps && loss_partial && B > 0 && S > 0 && B_global > 0, "dosx_loss_edos: bad args");
  hipLaunchKernelGGL(loss_edos_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, to_stream(stream), pg, ps, y_ft, beta, B, S,
                     1.f / (float)B_global, dpg, dps, loss_partial);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_sum(const float* src, int n, float* dst, dosx_stream_t stream) {
  DOSX_CHECK_ARG(src && dst && n > 0, "dosx_sum: bad args");
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, to_stream(stream), src, n, dst);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                          float eps, float weight_decay, int step, float grad_scale, dosx_stream_t stream) {
  if (n <= 0) return 0;
  DOSX_CHECK_ARG(p && g && m && v && step >= 1, "dosx_adamw: bad args");
  DOSX_CHECK_ARG(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                   reinterpret_cast<uintptr_t>(v)) & 15) == 0, "dosx_adamw: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_bc2s = (float)(1.0 / sqrt(bc2));
  const float decay = 1.f - lr * weight_decay;
  size_t g1 = ((size_t)n / 4 + 255) / 256;
  if (g1 > 4096) g1 = 4096;
  if (g1 < 1) g1 = 1;
  hipLaunchKernelGGL(adamw_kernel, dim3((int)g1), dim3(256), 0, to_stream(stream), p, g, m, v, (size_t)n, decay, beta1,
                     beta2, eps, step_size, inv_bc2s, grad_scale);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_copy_many(const DosxCopyJob* jobs_host, int n_jobs, dosx_stream_t stream) {
  if (n_jobs <= 0) return 0;
  DOSX_CHECK_ARG(jobs_host != nullptr, "dosx_copy_many: null job table");
  for (int done = 0; done < n_jobs; done += CP_MAX_JOBS) {
    CopyLaunch L;
    L.n = 0;
    int blocks = 0;
    for (int i = done; i < n_jobs && L.n < CP_MAX_JOBS; ++i) {
      const DosxCopyJob& j = jobs_host[i];
      DOSX_CHECK_ARG(j.dwords >= 0 && (j.dwords == 0 || (j.src && j.dst)), "dosx_copy_many: bad job %d", i);
      DOSX_CHECK_ARG(((reinterpret_cast<uintptr_t>(j.src) | reinterpret_cast<uintptr_t>(j.dst)) & 3) == 0,
                     "dosx_copy_many: job %d is not 4-byte aligned", i);
      if (j.dwords == 0) continue;
      L.job[L.n] = j;
      L.first_block[L.n] = blocks;
      blocks += (int)((j.dwords + CP_SLICE - 1) / CP_SLICE);
      ++L.n;
    }
    L.first_block[L.n] = blocks;
    if (blocks > 0) {
      hipLaunchKernelGGL(copy_many_kernel, dim3(blocks), dim3(256), 0, to_stream(stream), L);
      DOSX_LAUNCH_CHECK();
    }
  }
  return 0;
}

extern "C" int dosx_dropout_mask(float* mask, int64_t n, float p, const unsigned long long* seed_dev, long long stream_id,
                                 dosx_stream_t stream) {
  if (n <= 0) return 0;
  DOSX_CHECK_ARG(mask && seed_dev && p >= 0.f && p < 1.f, "dosx_dropout_mask: bad args (0 <= p < 1)");
  long long blocks = ((n + 3) / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, to_stream(stream), mask, (long long)n, p,
                     1.f / (1.f - p), seed_dev, (unsigned long long)stream_id);
  DOSX_LAUNCH_CHECK();
  return 0;
}
