// The EdgeModel of one message-passing layer as ONE launch per direction (round 5):
//
//   reference (DOSTransformer_phonon.py:186-197,209 / :81-84):
//       z   = Linear(3H,2H)(cat[x[row], x[col], e]) ; act = PReLU(LayerNorm(z)) ; msg = Linear(2H,H)(act)
//       agg = scatter_mean/sum(msg, col) ; e' = e + msg
//
// with the first Linear FACTORED (cat[x_s, x_d, e] W1^T = (x Wa^T)[s] + (x Wb^T)[d] + e Wc^T, the two node products are an
// N-row dosx_gemm_pair in front of this kernel), so both products of the kernel have E rows and K = H / K = 2H:
//
//   forward  (edge_fwd_kernel):  z = e Wc^T + b1 + P[src] + Q[dst] -> LayerNorm (xhat, rstd saved) -> PReLU -> . W3^T + b3
//                                -> e' = e + msg, agg[n] = scale[n] * sum over the node's destination segment
//
// One workgroup = one NODE-ALIGNED row tile of the batch's tile table (<= 48 edges = whole destination segments, the table of
// DosxGemm EPI_SEGSUM: include/dosx.h), three 16-row sub-tiles on the 16x16x4 fp32 MFMA.  The 48 x 2H intermediate never leaves
// the LDS; the e rows are read once (A operand of the first product AND the residual of the last epilogue).  Same wave
// specialisation as gemm.hip: waves 0-3 multiply, waves 4-7 stream the weight chunks (first Wc in 32-wide k-chunks, then W3 in
// 64-wide ones) through two LDS stage buffers, one barrier per chunk, loads two chunks deep; the chunk sequence runs straight
// through the phase boundary, so the first W3 chunks are in LDS / in flight while all 8 waves run the LayerNorm row phase.
#include <stdlib.h>

#include "common.h"

typedef int v4i32 __attribute__((ext_vector_type(4)));

// Diagnostic build only (-DDOSX_STAMPS, never shipped): s_memtime at the phase boundaries of one workgroup in the middle of the
// grid - slots [0, 32): matrix wave 0, [32, 64): staging wave 0 (tools/stamp_edge.py)
#ifdef DOSX_STAMPS
__device__ unsigned long long dosx_edge_stamp_buf[3 * 64];
extern "C" int dosx_debug_read_edge_stamps(unsigned long long* host192) {
  return (int)hipMemcpyFromSymbol(host192, HIP_SYMBOL(dosx_edge_stamp_buf), sizeof(unsigned long long) * 192);
}
#define ESTAMP(k, slot) do { if ((threadIdx.x == 0 || threadIdx.x == 256) && blockIdx.x == ((k) == 2 ? 5 : 37)) dosx_edge_stamp_buf[(k) * 64 + (threadIdx.x ? 32 : 0) + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ESTAMP(k, slot) do { } while (0)
#endif

namespace {

constexpr int ER_ = 48;          // rows of a tile (batch.SEG_TILE_ROWS)

template <int NH>
struct EdgeCfg {
  static constexpr int H = NH / 2;
  static constexpr int LDE = H + 4, LDT = NH + 4;
  static constexpr int KC1 = 32, NC1 = H / KC1;           // first product: k-chunks of Wc [NH rows][KC1]
  static constexpr int KC3 = 64, NC3 = NH / KC3;          // second product: k-chunks of W3 [H rows][KC3]
  static constexpr int LW1 = KC1 + 4, LW3 = KC3 + 4;
  static constexpr int STG = (NH * LW1 > H * LW3) ? NH * LW1 : H * LW3;
  static constexpr int CT1 = NH / 64, CT3 = H / 64;       // 16-column tiles per matrix wave
  static constexpr int NV1 = NH / 32, NV3 = H / 16;       // float4 per staging thread and chunk
  static constexpr int NV = NV1 > NV3 ? NV1 : NV3;
  static constexpr int KQ = NH / 64;                      // float4 per lane of a quarter-wave row
  static constexpr int LDC = H + 4;
  static constexpr int SMEM = ER_ * LDE + ER_ * LDT + 2 * STG + 3 * NH;   // floats
  static_assert(ER_ * LDC <= 2 * STG, "C tile aliases the stage buffers");
};

template <int NH>
__global__ __launch_bounds__(512) void edge_fwd_kernel(const DosxEdgeMlp a) {
  DOSX_SET_MAIN_PRIO();
  ESTAMP(0, 0);
  using C = EdgeCfg<NH>;
  constexpr int H = C::H, LDE = C::LDE, LDT = C::LDT, STG = C::STG, KQ = C::KQ, LDC = C::LDC;
  constexpr int NC1 = C::NC1, NC3 = C::NC3, NCH = NC1 + NC3;
  static_assert(NC1 % 2 == 0, "the phase boundary falls on an even chunk");
  extern __shared__ __align__(16) float sm[];
  float* Es = sm;                          // [48][LDE]  e rows: A of the first product, residual of the last epilogue
  float* T = Es + ER_ * LDE;               // [48][LDT]  z -> prelu(LN(z)): A of the second product
  float* ST = T + ER_ * LDT;               // 2 stage buffers; at the end the C tile [48][LDC]
  float* Gs = ST + 2 * STG;                // b1 | gamma | beta  [3][NH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int bx = blockIdx.x, T_ = a.seg_ntiles;
  const int m0 = a.seg_tile[bx], mend = a.seg_tile[bx + 1];
  const int nlo = a.seg_tile[T_ + 1 + bx], nhi = a.seg_tile[T_ + 2 + bx];
  if (m0 >= mend) {                        // no rows: at most nodes without a row of their own -> aggregate 0
    constexpr int w4 = H / 4;
    for (int i = tid; i < (nhi - nlo) * w4; i += 512) st4(a.seg_agg + (size_t)(nlo + i / w4) * H + (i % w4) * 4, f4zero());
    return;
  }
  const int rows = mend - m0;

  // staging waves: the first two weight chunks are requested BEFORE the prologue below (they depend on nothing of the tile)
  const int st = tid - 256;
  const float* wlo = a.w1 < a.w3 ? a.w1 : a.w3;
  const uint32_t d1 = (uint32_t)((const char*)a.w1 - (const char*)wlo), d3 = (uint32_t)((const char*)a.w3 - (const char*)wlo);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)wlo, 0, 0x7fffffff, 0x00020000);
  uint32_t v1[C::NV1], v3[C::NV3];
#pragma unroll
  for (int i = 0; i < C::NV1; ++i) v1[i] = d1 + (uint32_t)((((st >> 3) + 32 * i) * a.ldw1 + (st & 7) * 4) * 4);
#pragma unroll
  for (int i = 0; i < C::NV3; ++i) v3[i] = d3 + (uint32_t)((((st >> 4) + 16 * i) * NH + (st & 15) * 4) * 4);
  float4 r0[C::NV], r1[C::NV];
  auto issue = [&](float4(&r)[C::NV], int c) {
    const int cu = __builtin_amdgcn_readfirstlane(c);
    if (cu < NC1) {
      const int so = cu * C::KC1 * 4;
#pragma unroll
      for (int i = 0; i < C::NV1; ++i) r[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rW, v1[i], so, 0));
    } else {
      const int so = (cu - NC1) * C::KC3 * 4;
#pragma unroll
      for (int i = 0; i < C::NV3; ++i) r[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rW, v3[i], so, 0));
    }
  };
  if (wave_u >= 4) {
    issue(r0, 0);
    issue(r1, 1);
  }

  // ---- prologue, all 8 waves: the e tile, the LayerNorm constants, and the gathered node rows of this wave's 6 rows ----
  for (int i = tid; i < ER_ * (H / 4); i += 512) {
    const int r = i / (H / 4), c = (i % (H / 4)) * 4;
    st4(Es + r * LDE + c, ld4(a.e + (size_t)(m0 + min(r, rows - 1)) * a.lde + c));
  }
  for (int i = tid; i < NH / 4; i += 512) {
    st4(Gs + i * 4, ld4(a.b1 + i * 4));
    st4(Gs + NH + i * 4, ld4(a.gamma + i * 4));
    st4(Gs + 2 * NH + i * 4, ld4(a.beta + i * 4));
  }
  // quarter wave per row: pass p, quarter qd -> row wave * 6 + 4 p + qd (p = 1: qd < 2)
  float4 adp[2][KQ], adq[2][KQ];
  {
    int ip[2], iq[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int li = 4 * p + g4;
      const int rc = m0 + min(wave * 6 + (li < 6 ? li : 0), rows - 1);
      ip[p] = a.src[rc];
      iq[p] = a.dst[rc];
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int k = 0; k < KQ; ++k) {
        adp[p][k] = ld4(a.pq + (size_t)ip[p] * a.ldpq + l15 * 4 + 64 * k);
        adq[p][k] = ld4(a.pq + (size_t)iq[p] * a.ldpq + NH + l15 * 4 + 64 * k);
      }
  }
  const float alpha = *a.alpha;
  float4 b3v = f4zero();
  if (lane * 4 < H) b3v = ld4(a.b3 + lane * 4);
  // the segment bounds / scale of the first node this wave sums at the end: requested here (at the end they would be one more
  // exposed global round trip in front of the last stores)
  const int pinfo = a.seg_tile[2 * (T_ + 1) + bx];
  const int nfirst = pinfo ? nlo + 1 : nlo;
  const int nw0 = min(nfirst + wave, max(nhi - 1, 0));
  const int seg_b0 = a.seg_rowptr[nw0], seg_e0 = a.seg_rowptr[nw0 + 1];
  const float seg_s0 = a.seg_scale ? a.seg_scale[nw0] : 1.f;

  // the LayerNorm row phase at the phase boundary (all 8 waves; T holds z without bias / node rows)
  auto ln_rows = [&]() {
    constexpr float invN = 1.f / (float)NH;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int li = 4 * p + g4;
      const int lr = wave * 6 + (li < 6 ? li : 0);
      const bool rv = li < 6 && lr < rows;
      float4 v[KQ];
      float s1 = 0.f;
#pragma unroll
      for (int k = 0; k < KQ; ++k) {
        const int c = l15 * 4 + 64 * k;
        v[k] = f4add(f4add(ld4(T + lr * LDT + c), ld4(Gs + c)), f4add(adp[p][k], adq[p][k]));
        s1 += (v[k].x + v[k].y) + (v[k].z + v[k].w);
      }
      const float mean = row16_sum(s1) * invN;
      float s2 = 0.f;
#pragma unroll
      for (int k = 0; k < KQ; ++k) {
        v[k].x -= mean; v[k].y -= mean; v[k].z -= mean; v[k].w -= mean;
        s2 += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
      }
      const float rstd = rsqrtf(row16_sum(s2) * invN + DOSX_LN_EPS);
      if (li < 6) {                        // (rows beyond the tile are clamped duplicates: normalised like any row, never stored)
        float* const xrow = a.xhat + (size_t)(m0 + lr) * NH;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
          const int c = l15 * 4 + 64 * k;
          const float4 xh = make_float4(v[k].x * rstd, v[k].y * rstd, v[k].z * rstd, v[k].w * rstd);
          if (rv) st4(xrow + c, xh);
          const float4 g = ld4(Gs + NH + c), b = ld4(Gs + 2 * NH + c);
          float4 y = make_float4(xh.x * g.x + b.x, xh.y * g.y + b.y, xh.z * g.z + b.z, xh.w * g.w + b.w);
          y.x = y.x >= 0.f ? y.x : alpha * y.x; y.y = y.y >= 0.f ? y.y : alpha * y.y;
          y.z = y.z >= 0.f ? y.z : alpha * y.z; y.w = y.w >= 0.f ? y.w : alpha * y.w;
          st4(T + lr * LDT + c, y);
        }
        if (rv && l15 == 0) a.rstd[m0 + lr] = rstd;
      }
    }
  };

  if (wave_u >= 4) {
    // =============================== staging waves ===============================================
    auto store = [&](float* buf, const float4(&r)[C::NV], int c) {
      if (c < NC1) {
#pragma unroll
        for (int i = 0; i < C::NV1; ++i) st4(buf + ((st >> 3) + 32 * i) * C::LW1 + (st & 7) * 4, r[i]);
      } else {
#pragma unroll
        for (int i = 0; i < C::NV3; ++i) st4(buf + ((st >> 4) + 16 * i) * C::LW3 + (st & 15) * 4, r[i]);
      }
    };
    store(ST, r0, 0);
    issue(r0, 2);
    __syncthreads();                               // prologue tiles + chunk 0 visible
#pragma unroll
    for (int c = 0; c < NCH; c += 2) {
      store(ST + STG, r1, c + 1);                  // (NCH is even: chunk c + 1 always exists)
      if (c + 3 < NCH) issue(r1, c + 3);
      __syncthreads();                             // end of chunk c
      if (c + 2 < NCH) {
        store(ST, r0, c + 2);
        if (c + 4 < NCH) issue(r0, c + 4);
      }
      __syncthreads();                             // end of chunk c + 1
      if (c + 2 == NC1) {                          // phase boundary: z tile complete behind the first barrier
        __syncthreads();
        ln_rows();
        __syncthreads();
      }
    }
  } else {
    // =============================== matrix waves ================================================
    f32x4 acc1[3][C::CT1];
#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
      for (int t = 0; t < C::CT1; ++t) acc1[h][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    ESTAMP(0, 1);
    for (int c = 0; c < NC1; ++c) {
      const float* Ws = ST + (c & 1) * STG;
#pragma unroll
      for (int kk = 0; kk < C::KC1; kk += 16) {
        float4 av[3], bv[C::CT1];
#pragma unroll
        for (int h = 0; h < 3; ++h) av[h] = ld4(Es + (16 * h + l15) * LDE + c * C::KC1 + kk + 4 * g4);
#pragma unroll
        for (int t = 0; t < C::CT1; ++t) bv[t] = ld4(Ws + ((wave * C::CT1 + t) * 16 + l15) * C::LW1 + kk + 4 * g4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int h = 0; h < 3; ++h)
#pragma unroll
            for (int t = 0; t < C::CT1; ++t)
              acc1[h][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(av[h], j), f4get(bv[t], j), acc1[h][t], 0, 0, 0);
      }
      __syncthreads();
      ESTAMP(0, 20 + c);
    }
    // z tile -> T (C fragment: column lane & 15, rows 4 * (lane >> 4) + r)
#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
      for (int t = 0; t < C::CT1; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(16 * h + 4 * g4 + r) * LDT + (wave * C::CT1 + t) * 16 + l15] = acc1[h][t][r];
    __syncthreads();
    ESTAMP(0, 2);
    ln_rows();
    ESTAMP(0, 3);
    __syncthreads();
    ESTAMP(0, 4);
    f32x4 acc3[3][C::CT3];
#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
      for (int t = 0; t < C::CT3; ++t) acc3[h][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < NC3; ++c) {
      const float* Ws = ST + ((NC1 + c) & 1) * STG;
#pragma unroll
      for (int kk = 0; kk < C::KC3; kk += 16) {
        float4 av[3], bv[C::CT3];
#pragma unroll
        for (int h = 0; h < 3; ++h) av[h] = ld4(T + (16 * h + l15) * LDT + c * C::KC3 + kk + 4 * g4);
#pragma unroll
        for (int t = 0; t < C::CT3; ++t) bv[t] = ld4(Ws + ((wave * C::CT3 + t) * 16 + l15) * C::LW3 + kk + 4 * g4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int h = 0; h < 3; ++h)
#pragma unroll
            for (int t = 0; t < C::CT3; ++t)
              acc3[h][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(av[h], j), f4get(bv[t], j), acc3[h][t], 0, 0, 0);
      }
      __syncthreads();
      ESTAMP(0, 10 + c);
    }
    // msg tile -> C (aliases the stage buffers: the k-loop ended with a barrier)
#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
      for (int t = 0; t < C::CT3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) ST[(16 * h + 4 * g4 + r) * LDC + (wave * C::CT3 + t) * 16 + l15] = acc3[h][t][r];
  }
  __syncthreads();
  ESTAMP(0, 7);
  const float* Cs = ST;
  const bool con = lane * 4 < H;
  // ---- e' = e + msg (DOSTransformer_phonon.py:84): 6 rows per wave, lanes sweep the columns ----
  if (a.e_out != nullptr && con) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int lr = wave * 6 + i;
      if (lr < rows)
        st4(a.e_out + (size_t)(m0 + lr) * a.ldeo + lane * 4,
            f4add(f4add(ld4(Cs + lr * LDC + lane * 4), b3v), ld4(Es + lr * LDE + lane * 4)));
    }
  }
  // ---- segment sums: agg[n] = scale[n] * sum_{e in seg(n)} (msg[e] + b3), rows in order; over-full nodes as in gemm.hip's
  // EPI_SEGSUM (chunk sum published write-through, ticket on the node's first tile, the last arriver adds the chunk sums
  // in chunk order) ----
  if (pinfo && wave == 7) {
    const int n = nlo, ci = pinfo >> 16, nc = pinfo & 0xffff, t0 = bx - ci;
    const int re = min(a.seg_rowptr[n + 1], mend) - m0;
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)a.seg_part, 0, 0x7fffffff, 0x00020000);
    if (con) {
      float4 t = f4zero();
      for (int r = 0; r < re; ++r) t = f4add(t, f4add(ld4(Cs + r * LDC + lane * 4), b3v));
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i32, t), rP, (uint32_t)(((size_t)bx * H + lane * 4) * 4), 0, 16);   // sc1
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int tk = 0;
    if (lane == 0) tk = dosx_ticket(a.seg_cnt + t0);
    tk = __builtin_amdgcn_readfirstlane(tk);
    if (tk == nc - 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const float sc = a.seg_scale ? a.seg_scale[n] : 1.f;
      if (con) {
        float4 t = f4zero();
        for (int c = 0; c < nc; ++c) {
          const float4 p = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
              rP, (uint32_t)(((size_t)(t0 + c) * H + lane * 4) * 4), 0, 16));                                    // sc1
          t = c == 0 ? p : f4add(t, p);
        }
        st4(a.seg_agg + (size_t)n * H + lane * 4, make_float4(t.x * sc, t.y * sc, t.z * sc, t.w * sc));
      }
      if (lane == 0) __hip_atomic_store(a.seg_cnt + t0, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  for (int n = nfirst + wave; n < nhi; n += 8) {
    const bool first = n == nfirst + wave;
    const int rb = max(first ? seg_b0 : a.seg_rowptr[n], m0) - m0, re = min(first ? seg_e0 : a.seg_rowptr[n + 1], mend) - m0;
    const float sc = first ? seg_s0 : (a.seg_scale ? a.seg_scale[n] : 1.f);
    if (con) {
      float4 t = f4zero();
      for (int r = rb; r < re; r += 8) {            // 8 rows requested at once, added in row order
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ld4(Cs + min(r + u, re - 1) * LDC + lane * 4);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (r + u < re) t = f4add(t, f4add(v[u], b3v));
      }
      st4(a.seg_agg + (size_t)n * H + lane * 4, make_float4(t.x * sc, t.y * sc, t.z * sc, t.w * sc));
    }
  }
  ESTAMP(0, 8);
}

// ------------------------------------------------------------------------------------------------------------------
// Backward of the same block, one launch per layer:
//     dmsg[r] = de_next[r] + scale[dst[r]] * dagg[dst[r]]       (gradient of msg: the edge residual's + the aggregate's; written out:
//                                                              the second Linear's weight gradient reads it)
//     da = dmsg . W3 ; dy' = da o prelu'(xhat*gamma+beta) ; dz = LayerNorm_bwd(dy' o gamma)      [E,2H] written (weight gradients)
//     aggD[n] = sum_{r in seg(n)} dz[r]     (destination-node sums: the factored first Linear's dWb and the dx term aggD . Wb)
//     de[r] = dz[r] . Wc + de_next[r]       (gradient of e_l: the only E-row block of the first Linear's input gradient left)
//     partials[tile] = [ sum dy'*xhat (2H) | sum dy' (2H) | pad | sum_{y<0} da*y ]    (layout of DOSX_EPI_PRELU_LN_BWD)
// What dosx_edge_grad_combine + dosx_gemm(EPI_PRELU_LN_BWD_SEG) + dosx_gemm(dz Wc + res) compute.  Same tile / wave layout as
// the forward kernel; both weight matrices are read as stored (n-contiguous for these products: W3 [H][2H], Wc [2H][H]).
template <int NH>
struct EdgeBwdCfg {
  static constexpr int H = NH / 2;
  static constexpr int LDD = H + 4, LDT = NH + 4;
  static constexpr int KC1 = 32, NC1 = H / KC1;           // da = dmsg . W3: k-chunks [KC1 rows][NH]
  static constexpr int KC3 = 64, NC3 = NH / KC3;          // de = dz . Wc:   k-chunks [KC3 rows][H]
  static constexpr int LW1 = NH + 4, LW3 = H + 4;
  static constexpr int STG = (KC1 * LW1 > KC3 * LW3) ? KC1 * LW1 : KC3 * LW3;
  static constexpr int CT1 = NH / 64, CT3 = H / 64;
  static constexpr int NV1 = NH / 32, NV3 = H / 16;
  static constexpr int NV = NV1 > NV3 ? NV1 : NV3;
  static constexpr int LDC = H + 4;
  static constexpr int SMEM = ER_ * LDD + ER_ * LDT + 2 * STG + 2 * NH;   // floats
  static_assert(ER_ * LDC <= 2 * STG, "C tile aliases the stage buffers");
  static_assert(16 * NH + 8 <= ER_ * LDD, "the column sums alias the dmsg tile");
};

template <int NH>
__global__ __launch_bounds__(512) void edge_bwd_kernel(const DosxEdgeMlpBwd a) {
  DOSX_SET_MAIN_PRIO();
  ESTAMP(1, 0);
  using C = EdgeBwdCfg<NH>;
  constexpr int H = C::H, LDD = C::LDD, LDT = C::LDT, STG = C::STG, LDC = C::LDC;
  constexpr int NC1 = C::NC1, NC3 = C::NC3, NCH = NC1 + NC3;
  static_assert(NC1 % 2 == 0, "the phase boundary falls on an even chunk");
  extern __shared__ __align__(16) float sm[];
  float* Ds = sm;                          // [48][LDD]  dmsg rows: A of the first product; afterwards the column sums Ps
  float* T = Ds + ER_ * LDD;               // [48][LDT]  da -> dz: A of the second product
  float* ST = T + ER_ * LDT;               // 2 stage buffers; at the end the C tile [48][LDC]
  float* Gs = ST + 2 * STG;                // gamma | beta  [2][NH]
  float* Ps = Ds;                          // [8][2][NH] + 8
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int bx = blockIdx.x, T_ = a.seg_ntiles;
  const int m0 = a.seg_tile[bx], mend = a.seg_tile[bx + 1];
  const int nlo = a.seg_tile[T_ + 1 + bx], nhi = a.seg_tile[T_ + 2 + bx];
  if (m0 >= mend) {
    constexpr int w4 = NH / 4;
    for (int i = tid; i < (nhi - nlo) * w4; i += 512) st4(a.seg_agg + (size_t)(nlo + i / w4) * NH + (i % w4) * 4, f4zero());
    if (a.partials != nullptr)
      for (int i = tid; i < a.partial_ld; i += 512) a.partials[(size_t)bx * a.partial_ld + i] = 0.f;
    return;
  }
  const int rows = mend - m0;

  // staging waves: the first two weight chunks are requested BEFORE the prologue below
  const int st = tid - 256;
  const float* wlo = a.w1 < a.w3 ? a.w1 : a.w3;
  const uint32_t d1 = (uint32_t)((const char*)a.w3 - (const char*)wlo), d3 = (uint32_t)((const char*)a.w1 - (const char*)wlo);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)wlo, 0, 0x7fffffff, 0x00020000);
  uint32_t v1[C::NV1], v3[C::NV3];
#pragma unroll
  for (int i = 0; i < C::NV1; ++i) {
    const int lin = st + 256 * i, r = lin / (NH / 4), c4 = (lin % (NH / 4)) * 4;
    v1[i] = d1 + (uint32_t)((r * NH + c4) * 4);
  }
#pragma unroll
  for (int i = 0; i < C::NV3; ++i) {
    const int lin = st + 256 * i, r = lin / (H / 4), c4 = (lin % (H / 4)) * 4;
    v3[i] = d3 + (uint32_t)((r * a.ldw1 + c4) * 4);
  }
  float4 r0[C::NV], r1[C::NV];
  auto issue = [&](float4(&r)[C::NV], int c) {
    if (c < NC1) {
      const int so = c * C::KC1 * NH * 4;
#pragma unroll
      for (int i = 0; i < C::NV1; ++i) r[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rW, v1[i], so, 0));
    } else {
      const int so = __builtin_amdgcn_readfirstlane((c - NC1) * C::KC3 * a.ldw1 * 4);
#pragma unroll
      for (int i = 0; i < C::NV3; ++i) r[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rW, v3[i], so, 0));
    }
  };
  if (wave_u >= 4) {
    issue(r0, 0);
    issue(r1, 1);
  }

  // ---- prologue, all 8 waves: the dmsg tile (written out too), gamma / beta, the xhat rows of this wave's 6 rows ----
  for (int i = tid; i < ER_ * (H / 4); i += 512) {
    const int r = i / (H / 4), c = (i % (H / 4)) * 4;
    const int e = m0 + min(r, rows - 1);
    const int d = a.dst[e];
    const float sc = a.seg_scale ? a.seg_scale[d] : 1.f;
    const float4 g = ld4(a.dagg + (size_t)d * a.lddagg + c);
    float4 v = make_float4(sc * g.x, sc * g.y, sc * g.z, sc * g.w);
    if (a.de_next != nullptr) v = f4add(v, ld4(a.de_next + (size_t)e * a.ldden + c));
    st4(Ds + r * LDD + c, v);
    if (r < rows) st4(a.dmsg + (size_t)e * H + c, v);
  }
  for (int i = tid; i < NH / 4; i += 512) {
    st4(Gs + i * 4, ld4(a.gamma + i * 4));
    st4(Gs + NH + i * 4, ld4(a.beta + i * 4));
  }
  const bool on = lane * 4 < NH;           // row phase: lanes sweep the 2H columns as float4
  const bool con = lane * 4 < H;           // last epilogue: the H columns
  float4 xh[6], dn[6];                     // dn: the de_next rows the last epilogue adds (in flight under both products)
  float rs[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int e = m0 + min(wave * 6 + i, rows - 1);
    xh[i] = ld4(a.xhat + (size_t)e * NH + (on ? lane * 4 : 0));
    rs[i] = a.rstd[e];
    dn[i] = a.de_next != nullptr ? ld4(a.de_next + (size_t)e * a.ldden + (con ? lane * 4 : 0)) : f4zero();
  }
  const float alpha = *a.alpha;

  // PReLU + LayerNorm backward over this wave's 6 rows of the da tile (all 8 waves), column sums into Ps
  auto row_phase = [&]() {
    constexpr float invN = 1.f / (float)NH;
    const float4 gm = on ? ld4(Gs + lane * 4) : f4zero(), bt = on ? ld4(Gs + NH + lane * 4) : f4zero();
    float4 pg = f4zero(), pb = f4zero();
    float pal = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int lr = wave * 6 + i;
      const bool rvalid = lr < rows;               // (wave-uniform)
      float4 dxh = f4zero();
      const float4 x = xh[i];
      float s1 = 0.f, s2 = 0.f;
      if (on && rvalid) {
        float4 dy = ld4(T + lr * LDT + lane * 4);
        const float y0 = x.x * gm.x + bt.x, y1 = x.y * gm.y + bt.y, y2 = x.z * gm.z + bt.z, y3 = x.w * gm.w + bt.w;
        if (y0 < 0.f) { pal += dy.x * y0; dy.x *= alpha; }
        if (y1 < 0.f) { pal += dy.y * y1; dy.y *= alpha; }
        if (y2 < 0.f) { pal += dy.z * y2; dy.z *= alpha; }
        if (y3 < 0.f) { pal += dy.w * y3; dy.w *= alpha; }
        pg.x += dy.x * x.x; pg.y += dy.y * x.y; pg.z += dy.z * x.z; pg.w += dy.w * x.w;
        pb = f4add(pb, dy);
        dxh = make_float4(dy.x * gm.x, dy.y * gm.y, dy.z * gm.z, dy.w * gm.w);
        s1 = (dxh.x + dxh.y) + (dxh.z + dxh.w);
        s2 = (dxh.x * x.x + dxh.y * x.y) + (dxh.z * x.z + dxh.w * x.w);
      }
      const float m1 = wave_sum(s1) * invN, m2 = wave_sum(s2) * invN;
      if (on) {
        float4 o = f4zero();                       // rows beyond the tile feed zeros to the second product
        if (rvalid) {
          o = make_float4(rs[i] * (dxh.x - m1 - x.x * m2), rs[i] * (dxh.y - m1 - x.y * m2),
                          rs[i] * (dxh.z - m1 - x.z * m2), rs[i] * (dxh.w - m1 - x.w * m2));
          st4(a.dz + (size_t)(m0 + lr) * NH + lane * 4, o);
        }
        st4(T + lr * LDT + lane * 4, o);
      }
    }
    if (on) {
      st4(Ps + (wave * 2 + 0) * NH + lane * 4, pg);
      st4(Ps + (wave * 2 + 1) * NH + lane * 4, pb);
    }
    const float sal = wave_sum(pal);
    if (lane == 0) Ps[16 * NH + wave] = sal;
  };

  if (wave_u >= 4) {
    // =============================== staging waves ===============================================
    auto store = [&](float* buf, const float4(&r)[C::NV], int c) {
      if (c < NC1) {
#pragma unroll
        for (int i = 0; i < C::NV1; ++i) {
          const int lin = st + 256 * i;
          st4(buf + (lin / (NH / 4)) * C::LW1 + (lin % (NH / 4)) * 4, r[i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < C::NV3; ++i) {
          const int lin = st + 256 * i;
          st4(buf + (lin / (H / 4)) * C::LW3 + (lin % (H / 4)) * 4, r[i]);
        }
      }
    };
    store(ST, r0, 0);
    issue(r0, 2);
    __syncthreads();                               // prologue tiles + chunk 0 visible
#pragma unroll
    for (int c = 0; c < NCH; c += 2) {
      store(ST + STG, r1, c + 1);
      if (c + 3 < NCH) issue(r1, c + 3);
      __syncthreads();                             // end of chunk c
      if (c + 2 < NCH) {
        store(ST, r0, c + 2);
        if (c + 4 < NCH) issue(r0, c + 4);
      }
      __syncthreads();                             // end of chunk c + 1
      if (c + 2 == NC1) {                          // phase boundary
        __syncthreads();                           // da tile complete
        row_phase();
        __syncthreads();                           // dz tile + column sums complete
        // the staging waves have slack here (the next two weight chunks are stored / in flight): they finish the column
        // sums and the destination-node sums while the matrix waves start the second product
        if (a.partials != nullptr) {
          float* prow = a.partials + (size_t)bx * a.partial_ld;
          for (int cc = st; cc < 2 * NH; cc += 256) {
            const int which = cc >= NH ? 1 : 0, col = cc - which * NH;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += Ps[(w * 2 + which) * NH + col];
            prow[which * NH + col] = s;
          }
          if (st == 0) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += Ps[16 * NH + w];
            prow[a.partial_ld - 1] = s;
          }
        }
        // aggD[n] = sum of the node's dz rows (rows in order; over-full nodes: chunk sum + ticket, as in the forward kernel)
        const int pinfo = a.seg_tile[2 * (T_ + 1) + bx];
        const int nfirst = pinfo ? nlo + 1 : nlo;
        if (pinfo && wave == 7) {
          const int n = nlo, ci = pinfo >> 16, nc = pinfo & 0xffff, t0 = bx - ci;
          const int re = min(a.seg_rowptr[n + 1], mend) - m0;
          const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)a.seg_part, 0, 0x7fffffff, 0x00020000);
          if (on) {
            float4 t = f4zero();
            for (int r = 0; r < re; ++r) t = f4add(t, ld4(T + r * LDT + lane * 4));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i32, t), rP, (uint32_t)(((size_t)bx * NH + lane * 4) * 4), 0, 16);   // sc1
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          int tk = 0;
          if (lane == 0) tk = dosx_ticket(a.seg_cnt + t0);
          tk = __builtin_amdgcn_readfirstlane(tk);
          if (tk == nc - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (on) {
              float4 t = f4zero();
              for (int cI = 0; cI < nc; ++cI) {
                const float4 p = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                    rP, (uint32_t)(((size_t)(t0 + cI) * NH + lane * 4) * 4), 0, 16));                              // sc1
                t = cI == 0 ? p : f4add(t, p);
              }
              st4(a.seg_agg + (size_t)n * NH + lane * 4, t);
            }
            if (lane == 0) __hip_atomic_store(a.seg_cnt + t0, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        for (int n = nfirst + (wave - 4); n < nhi; n += 4) {
          const int rb = max(a.seg_rowptr[n], m0) - m0, re = min(a.seg_rowptr[n + 1], mend) - m0;
          if (on) {
            float4 t = f4zero();
            for (int r = rb; r < re; r += 8) {        // 8 rows requested at once, added in row order
              float4 v[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) v[u] = ld4(T + min(r + u, re - 1) * LDT + lane * 4);
#pragma unroll
              for (int u = 0; u < 8; ++u)
                if (r + u < re) t = f4add(t, v[u]);
            }
            st4(a.seg_agg + (size_t)n * NH + lane * 4, t);
          }
        }
      }
    }
  } else {
    // =============================== matrix waves ================================================
    f32x4 acc1[3][C::CT1];
#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
      for (int t = 0; t < C::CT1; ++t) acc1[h][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    ESTAMP(1, 1);
    for (int c = 0; c < NC1; ++c) {
      const float* Ws = ST + (c & 1) * STG;
#pragma unroll
      for (int kk = 0; kk < C::KC1; kk += 16) {
        float4 av[3];
        float bv[C::CT1][4];
#pragma unroll
        for (int h = 0; h < 3; ++h) av[h] = ld4(Ds + (16 * h + l15) * LDD + c * C::KC1 + kk + 4 * g4);
#pragma unroll
        for (int t = 0; t < C::CT1; ++t) {
          const float* bp = Ws + (kk + 4 * g4) * C::LW1 + (wave * C::CT1 + t) * 16 + l15;
          bv[t][0] = bp[0]; bv[t][1] = bp[C::LW1]; bv[t][2] = bp[2 * C::LW1]; bv[t][3] = bp[3 * C::LW1];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int h = 0; h < 3; ++h)
#pragma unroll
            for (int t = 0; t < C::CT1; ++t)
              acc1[h][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(av[h], j), bv[t][j], acc1[h][t], 0, 0, 0);
      }
      __syncthreads();
      ESTAMP(1, 20 + c);
    }
#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
      for (int t = 0; t < C::CT1; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(16 * h + 4 * g4 + r) * LDT + (wave * C::CT1 + t) * 16 + l15] = acc1[h][t][r];
    __syncthreads();
    ESTAMP(1, 2);
    row_phase();
    ESTAMP(1, 3);
    __syncthreads();
    ESTAMP(1, 4);
    f32x4 acc3[3][C::CT3];
#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
      for (int t = 0; t < C::CT3; ++t) acc3[h][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < NC3; ++c) {
      const float* Ws = ST + ((NC1 + c) & 1) * STG;
#pragma unroll
      for (int kk = 0; kk < C::KC3; kk += 16) {
        float4 av[3];
        float bv[C::CT3][4];
#pragma unroll
        for (int h = 0; h < 3; ++h) av[h] = ld4(T + (16 * h + l15) * LDT + c * C::KC3 + kk + 4 * g4);
#pragma unroll
        for (int t = 0; t < C::CT3; ++t) {
          const float* bp = Ws + (kk + 4 * g4) * C::LW3 + (wave * C::CT3 + t) * 16 + l15;
          bv[t][0] = bp[0]; bv[t][1] = bp[C::LW3]; bv[t][2] = bp[2 * C::LW3]; bv[t][3] = bp[3 * C::LW3];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int h = 0; h < 3; ++h)
#pragma unroll
            for (int t = 0; t < C::CT3; ++t)
              acc3[h][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(av[h], j), bv[t][j], acc3[h][t], 0, 0, 0);
      }
      __syncthreads();
      ESTAMP(1, 10 + c);
    }
#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
      for (int t = 0; t < C::CT3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) ST[(16 * h + 4 * g4 + r) * LDC + (wave * C::CT3 + t) * 16 + l15] = acc3[h][t][r];
  }
  __syncthreads();
  ESTAMP(1, 7);
  // ---- de = dz . Wc + de_next: 6 rows per wave ----
  if (con) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int lr = wave * 6 + i;
      if (lr < rows) {
        st4(a.de + (size_t)(m0 + lr) * a.ldde + lane * 4, f4add(ld4(ST + lr * LDC + lane * 4), dn[i]));
      }
    }
  }
  ESTAMP(1, 8);
}

// ------------------------------------------------------------------------------------------------------------------
// The NODE side of the factored input gradient in one launch (behind edge_bwd_kernel):
//     aggS[n] = sum_{e: src(e) = n} dz[e]                       (source-node sums, CSR by source; written: dWa reads them)
//     dx[n]   = res[n] (+ res2[n]) + aggS[n] . Wa + aggD[n] . Wb        Wa = W1[:, :H], Wb = W1[:, H:2H]  ([2H, 3H], row stride ldw)
// What dosx_segment_reduce_perm + dosx_gemm (two K-segments, w_seg_off) compute.  A few hundred rows, every launch one latency
// chain: one workgroup = 8 nodes, ONE PER WAVE, so the source sums are three dependent round trips (segment bounds -> the
// segment's edge ids, one per lane -> up to 24 rows in flight); the whole K = 4H sits in LDS as A tile (rows 8-15 of the
// 16-row MFMA tile are don't-cares); the 8 waves own the 16-column tiles of the output and read THEIR columns of the weights
// straight from L2 into MFMA B fragments, k-chunks of 128 requested before the source sums are gathered (hidden <= 128: all of
// them; 256: two chunks ahead) - no weight staging through LDS, no barrier in the product, no split-K reduction.
constexpr int NG_NODES = 8;
template <int H>
__global__ __launch_bounds__(512) void node_grad_kernel(const DosxNodeGrad a) {
  DOSX_SET_MAIN_PRIO();
  constexpr int NH = 2 * H, K = 2 * NH, LDA = K + 4, KC = 128, NCK = K / KC;
  constexpr int NT = H / 16;                       // 16-column tiles of the output
  constexpr int CTW = NT >= 8 ? NT / 8 : 1;        // tiles per wave
  constexpr int NCB = NH / 256 ? NH / 256 : 1;     // 256-column blocks of a 2H-wide row (lanes sweep them as float4)
  constexpr int NBUF = (NCK * CTW <= 4) ? NCK : 2; // weight chunks in registers at once
  constexpr int NG_U = H >= 256 ? 12 : 24;         // rows of a source segment in flight (Phonon-DOS graphs: 20 out-edges per atom, Electron-DOS: 12)
  extern __shared__ __align__(16) float sm[];
  float* As = sm;                                  // [16][LDA]  aggS | aggD rows of this workgroup's nodes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  const int n0 = blockIdx.x * NG_NODES, N = a.N;
  ESTAMP(2, 0);
  const bool mm = wave * CTW < NT;                 // this wave multiplies (hidden 64: four tiles, waves 0-3)
  const int col0 = (mm ? wave * CTW : 0) * 16 + l15;
  float bq[NBUF][CTW][KC / 4];
  auto issue = [&](int buf, int c) {               // chunk c: k in [c KC, c KC + KC); the S rows multiply Wa, the D rows Wb
    const int k0 = c * KC, sg = k0 >= NH ? 1 : 0;
    const float* wp = a.w + (size_t)(k0 - sg * NH + 4 * g4) * a.ldw + sg * H + col0;
#pragma unroll
    for (int s = 0; s < KC / 16; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < CTW; ++t) bq[buf][t][s * 4 + j] = wp[(size_t)(16 * s + j) * a.ldw + 16 * t];
  };
  if (mm) {
#pragma unroll
    for (int c = 0; c < NBUF; ++c) issue(c, c);
  }
  // ---- source sums of this wave's node: bounds -> edge ids (lane j holds the id of the segment's j-th edge) -> rows, in CSR
  // order, NG_U in flight ----
  {
    const int r = wave, n = n0 + r;
    const bool nv = n < N;
    const int beg = nv ? a.rowptr_src[n] : 0, end = nv ? a.rowptr_src[n + 1] : 0;
    float4 aggdv[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int c = cb * 256 + lane * 4;
      aggdv[cb] = (nv && c < NH) ? ld4(a.aggd + (size_t)n * NH + c) : f4zero();
    }
    float4 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[cb] = f4zero();
    for (int j0 = beg; j0 < end; j0 += 64) {
      const int je = min(end, j0 + 64);
      const int myid = a.perm_src[min(j0 + lane, je - 1)];
      for (int u0 = j0; u0 < je; u0 += NG_U) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const int c = cb * 256 + lane * 4;
          const int cc = c < NH ? c : 0;
          float4 m[NG_U];
#pragma unroll
          for (int u = 0; u < NG_U; ++u) {
            const int ei = __builtin_amdgcn_readlane(myid, min(u0 - j0 + u, 63));      // (wave-uniform index: u0, j0 are)
            m[u] = ld4(a.dz + (size_t)ei * NH + cc);
          }
#pragma unroll
          for (int u = 0; u < NG_U; ++u)
            if (u0 + u < je) acc[cb] = f4add(acc[cb], m[u]);
        }
      }
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int c = cb * 256 + lane * 4;
      if (c < NH) {
        st4(As + r * LDA + c, acc[cb]);
        if (nv) st4(a.aggs + (size_t)n * NH + c, acc[cb]);
        st4(As + r * LDA + NH + c, aggdv[cb]);
      }
    }
  }
  __syncthreads();
  ESTAMP(2, 1);
  if (!mm) return;
  float rsd[CTW][4];                                // the residual addends of this lane's outputs, in flight under the product
#pragma unroll
  for (int t = 0; t < CTW; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = min(n0 + 4 * (g4 & (NG_NODES / 4 - 1)) + r, N - 1), col = col0 + 16 * t;
      rsd[t][r] = (a.res ? a.res[(size_t)n * a.ldres + col] : 0.f) + (a.res2 ? a.res2[(size_t)n * a.ldres2 + col] : 0.f);
    }
  f32x4 acc[CTW];
#pragma unroll
  for (int t = 0; t < CTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NCK; ++c) {
    if (NBUF < NCK && c >= 1 && c + 1 < NCK) issue((c + 1) % NBUF, c + 1);      // (two buffers: chunk c + 1 into the one chunk c - 1 left)
#pragma unroll
    for (int s = 0; s < KC / 16; ++s) {
      const float4 av = ld4(As + l15 * LDA + c * KC + 16 * s + 4 * g4);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < CTW; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(av, j), bq[c % NBUF][t][s * 4 + j], acc[t], 0, 0, 0);
    }
  }
  ESTAMP(2, 2);
  if (g4 < NG_NODES / 4) {
#pragma unroll
    for (int t = 0; t < CTW; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 4 * g4 + r, col = col0 + 16 * t;
        if (n < N) a.dx[(size_t)n * a.lddx + col] = acc[t][r] + rsd[t][r];
      }
  }
  ESTAMP(2, 3);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int NH>
int launch_edge_fwd(const DosxEdgeMlp& a, hipStream_t s) {
  constexpr size_t smem = sizeof(float) * (size_t)EdgeCfg<NH>::SMEM;
  static_assert(smem <= 160 * 1024, "edge_fwd_kernel: LDS");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_fwd_kernel<NH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((edge_fwd_kernel<NH>), dim3(a.seg_ntiles), dim3(512), smem, s, a);
  DOSX_LAUNCH_CHECK();
  return 0;
}

template <int NH>
int launch_edge_bwd(const DosxEdgeMlpBwd& a, hipStream_t s) {
  constexpr size_t smem = sizeof(float) * (size_t)EdgeBwdCfg<NH>::SMEM;
  static_assert(smem <= 160 * 1024, "edge_bwd_kernel: LDS");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_bwd_kernel<NH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((edge_bwd_kernel<NH>), dim3(a.seg_ntiles), dim3(512), smem, s, a);
  DOSX_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int dosx_edge_mlp_supported(int H) { return H == 64 || H == 128; }

extern "C" int dosx_edge_mlp_fwd(const DosxEdgeMlp* ap, dosx_stream_t stream) {
  DOSX_CHECK_ARG(ap != nullptr, "dosx_edge_mlp_fwd: null descriptor");
  const DosxEdgeMlp& a = *ap;
  if (a.E <= 0) return 0;
  DOSX_CHECK_ARG(dosx_edge_mlp_supported(a.H), "dosx_edge_mlp_fwd: hidden %d unsupported (64 or 128)", a.H);
  DOSX_CHECK_ARG(a.e && a.pq && a.src && a.dst && a.w1 && a.b1 && a.gamma && a.beta && a.alpha && a.w3 && a.b3 && a.xhat && a.rstd,
                 "dosx_edge_mlp_fwd: null operand");
  DOSX_CHECK_ARG(a.seg_tile && a.seg_ntiles > 0 && a.seg_rowptr && a.seg_agg && a.seg_part && a.seg_cnt,
                 "dosx_edge_mlp_fwd: needs seg_tile / seg_rowptr / seg_agg / seg_part / seg_cnt");
  DOSX_CHECK_ARG((a.lde & 3) == 0 && (a.ldpq & 3) == 0 && (a.ldw1 & 3) == 0 && (a.ldeo & 3) == 0 && a.ldpq >= 4 * a.H && aligned16(a.e) &&
                     aligned16(a.pq) && aligned16(a.w1) && aligned16(a.w3) && aligned16(a.b1) && aligned16(a.b3) && aligned16(a.gamma) &&
                     aligned16(a.beta) && aligned16(a.xhat) && aligned16(a.e_out) && aligned16(a.seg_agg),
                 "dosx_edge_mlp_fwd: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
  const long long span = (const char*)a.w1 > (const char*)a.w3 ? (const char*)a.w1 - (const char*)a.w3 : (const char*)a.w3 - (const char*)a.w1;
  DOSX_CHECK_ARG(span + (long long)4 * 2 * a.H * (a.ldw1 + 2 * a.H) < 0x7fffffffLL, "dosx_edge_mlp_fwd: the two weight matrices are more than 2 GiB apart");
  hipStream_t s = to_stream(stream);
  return a.H == 128 ? launch_edge_fwd<256>(a, s) : launch_edge_fwd<128>(a, s);
}

extern "C" int dosx_edge_mlp_bwd(const DosxEdgeMlpBwd* ap, dosx_stream_t stream) {
  DOSX_CHECK_ARG(ap != nullptr, "dosx_edge_mlp_bwd: null descriptor");
  const DosxEdgeMlpBwd& a = *ap;
  if (a.E <= 0) return 0;
  DOSX_CHECK_ARG(dosx_edge_mlp_supported(a.H), "dosx_edge_mlp_bwd: hidden %d unsupported (64 or 128)", a.H);
  DOSX_CHECK_ARG(a.dagg && a.dst && a.xhat && a.rstd && a.w1 && a.w3 && a.gamma && a.beta && a.alpha && a.dmsg && a.dz && a.de,
                 "dosx_edge_mlp_bwd: null operand");
  DOSX_CHECK_ARG(a.seg_tile && a.seg_ntiles > 0 && a.seg_rowptr && a.seg_agg && a.seg_part && a.seg_cnt,
                 "dosx_edge_mlp_bwd: needs seg_tile / seg_rowptr / seg_agg / seg_part / seg_cnt");
  DOSX_CHECK_ARG(!a.partials || a.partial_ld >= 4 * a.H + 1, "dosx_edge_mlp_bwd: partial_ld %d < 4H+1", a.partial_ld);
  DOSX_CHECK_ARG((a.lddagg & 3) == 0 && (a.ldden & 3) == 0 && (a.ldw1 & 3) == 0 && (a.ldde & 3) == 0 && aligned16(a.dagg) &&
                     aligned16(a.de_next) && aligned16(a.w1) && aligned16(a.w3) && aligned16(a.gamma) && aligned16(a.beta) &&
                     aligned16(a.xhat) && aligned16(a.dmsg) && aligned16(a.dz) && aligned16(a.de) && aligned16(a.seg_agg),
                 "dosx_edge_mlp_bwd: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
  const long long span = (const char*)a.w1 > (const char*)a.w3 ? (const char*)a.w1 - (const char*)a.w3 : (const char*)a.w3 - (const char*)a.w1;
  DOSX_CHECK_ARG(span + (long long)4 * 2 * a.H * (a.ldw1 + 2 * a.H) < 0x7fffffffLL, "dosx_edge_mlp_bwd: the two weight matrices are more than 2 GiB apart");
  hipStream_t s = to_stream(stream);
  return a.H == 128 ? launch_edge_bwd<256>(a, s) : launch_edge_bwd<128>(a, s);
}

extern "C" int dosx_node_grad(const DosxNodeGrad* ap, dosx_stream_t stream) {
  DOSX_CHECK_ARG(ap != nullptr, "dosx_node_grad: null descriptor");
  const DosxNodeGrad& a = *ap;
  if (a.N <= 0) return 0;
  DOSX_CHECK_ARG(a.H == 64 || a.H == 128 || a.H == 256, "dosx_node_grad: hidden %d unsupported (64, 128, 256)", a.H);
  DOSX_CHECK_ARG(a.dz && a.rowptr_src && a.perm_src && a.aggd && a.w && a.aggs && a.dx, "dosx_node_grad: null operand");
  DOSX_CHECK_ARG(aligned16(a.dz) && aligned16(a.aggd) && aligned16(a.aggs) && a.ldw >= 2 * a.H, "dosx_node_grad: dz / aggd / aggs must be 16-byte aligned");
  hipStream_t s = to_stream(stream);
  const dim3 grid(ceil_div(a.N, NG_NODES));
#define DOSX_NG(HH)                                                                                                       \
  do {                                                                                                                    \
    constexpr size_t smem = sizeof(float) * 16 * (4 * HH + 4);                                                            \
    static bool attr_set = false;                                                                                         \
    if (!attr_set) {                                                                                                      \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&node_grad_kernel<HH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
      attr_set = true;                                                                                                    \
    }                                                                                                                     \
    hipLaunchKernelGGL((node_grad_kernel<HH>), grid, dim3(512), smem, s, a);                                              \
  } while (0)
  if (a.H == 64) DOSX_NG(64);
  else if (a.H == 128) DOSX_NG(128);
  else DOSX_NG(256);
#undef DOSX_NG
  DOSX_LAUNCH_CHECK();
  return 0;
}
