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
    auto store = [&](float* buf, const float4(&r)[C::NV], int c) {
      if (c < NC1) {
#pragma unroll
        for (int i = 0; i < C::NV1; ++i) st4(buf + ((st >> 3) + 32 * i) * C::LW1 + (st & 7) * 4, r[i]);
      } else {
#pragma unroll
        for (int i = 0; i < C::NV3; ++i) st4(buf + ((st >> 4) + 16 * i) * C::LW3 + (st & 15) * 4, r[i]);
      }
    };
    issue(r0, 0);
    issue(r1, 1);
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
    }
    // z tile -> T (C fragment: column lane & 15, rows 4 * (lane >> 4) + r)
#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
      for (int t = 0; t < C::CT1; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(16 * h + 4 * g4 + r) * LDT + (wave * C::CT1 + t) * 16 + l15] = acc1[h][t][r];
    __syncthreads();
    ln_rows();
    __syncthreads();
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
  const int pinfo = a.seg_tile[2 * (T_ + 1) + bx];
  const int nfirst = pinfo ? nlo + 1 : nlo;
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
    const int rb = max(a.seg_rowptr[n], m0) - m0, re = min(a.seg_rowptr[n + 1], mend) - m0;
    const float sc = a.seg_scale ? a.seg_scale[n] : 1.f;
    if (con) {
      float4 t = f4zero();
      for (int r = rb; r < re; ++r) t = f4add(t, f4add(ld4(Cs + r * LDC + lane * 4), b3v));
      st4(a.seg_agg + (size_t)n * H + lane * 4, make_float4(t.x * sc, t.y * sc, t.z * sc, t.w * sc));
    }
  }
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
