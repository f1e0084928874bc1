// Graph-structured, HBM-bound kernels of the message-passing stack (gfx950).
//
// All of them are atomic-free: edges are stored sorted by destination, so the edge->node
// aggregation (torch_scatter.scatter_mean/sum in the reference) is a contiguous CSR segment
// reduction, and the gather-backward scatter-adds go through the per-source inverse index.
// Rows are H floats; `lpr` = min(64, H/4) lanes own one row as float4, so a wave streams
// 64/lpr rows per step with fully coalesced 16-B accesses.
#include "common.h"

namespace {

__device__ __forceinline__ float smooth_cutoff_f(float x) {
  // e3nn gate_points_2101.smooth_cutoff: u = 2(x-1); (1 - cos(pi u))/2 ; 0 if u > 0 ; 1 if u < -1
  const float u = 2.f * (x - 1.f);
  float y = (1.f - cospif(u)) * 0.5f;
  if (u > 0.f) y = 0.f;
  if (u < -1.f) y = 1.f;
  return y;
}

__global__ void edge_feat_kernel(const float* __restrict__ vec, float* __restrict__ out, int E, float inv_rmax) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const float x = vec[3 * (size_t)e], y = vec[3 * (size_t)e + 1], z = vec[3 * (size_t)e + 2];
  const float len = sqrtf(x * x + y * y + z * z);
  const float inv = 1.f / fmaxf(len, 1e-12f);
  const float c = smooth_cutoff_f(len * inv_rmax);
  const float s3 = 1.7320508075688772f * c;
  st4(out + 4 * (size_t)e, make_float4(c, s3 * x * inv, s3 * y * inv, s3 * z * inv));
}

// Phonon edge embedding, first layer: edge_attr = SH(l<=1)(v) * smooth_cutoff(|v|/r_max)  (DOSTransformer_phonon.py:74-77)
// AND z = edge_attr . W0^T + b0 (the K = 4 Linear of GN_encoder.edge_encoder, :129,142) in one pass: a K = 4 "GEMM" is 4
// fma per output - as a dosx_gemm it was a 9.6 us launch behind a 4.4 us feature kernel.  Same k-ordered fma chain as the
// MFMA path (k = 0..3, then + bias).  One thread per (edge, 4 output columns).
__global__ void edge_embed_kernel(const float* __restrict__ vec, const float* __restrict__ w0, const float* __restrict__ b0,
                                  float* __restrict__ attr, float* __restrict__ z, int E, int H, float inv_rmax) {
  const int h4 = H >> 2;
  const size_t total = (size_t)E * h4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(i / h4), c = (int)(i % h4) * 4;
    const float x = vec[3 * (size_t)e], y = vec[3 * (size_t)e + 1], zz = vec[3 * (size_t)e + 2];
    const float len = sqrtf(x * x + y * y + zz * zz);
    const float inv = 1.f / fmaxf(len, 1e-12f);
    const float cut = smooth_cutoff_f(len * inv_rmax);
    const float s3 = 1.7320508075688772f * cut;
    const float4 f = make_float4(cut, s3 * x * inv, s3 * y * inv, s3 * zz * inv);
    if (c == 0) st4(attr + 4 * (size_t)e, f);
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 w = ld4(w0 + (size_t)(c + j) * 4);
      float t = f.x * w.x;
      t = fmaf(f.y, w.y, t);
      t = fmaf(f.z, w.z, t);
      t = fmaf(f.w, w.w, t);
      o[j] = t;
    }
    const float4 b = ld4(b0 + c);
    st4(z + (size_t)e * H + c, make_float4(o[0] + b.x, o[1] + b.y, o[2] + b.z, o[3] + b.w));
  }
}

// Sub-row layout helper: lane -> (sub-row slot, float4 column)
struct RowLanes {
  int lpr;     // lanes per row
  int rps;     // rows per wave step = 64 / lpr
  int slot;    // which row of the step this lane serves
  int c4;      // first column (floats) owned by this lane
};
__device__ __forceinline__ RowLanes row_lanes(int H, int lane) {
  RowLanes r;
  r.lpr = min(64, H >> 2);
  r.rps = 64 / r.lpr;
  r.slot = lane / r.lpr;
  r.c4 = (lane % r.lpr) * 4;
  return r;
}
// sum the per-slot partial vectors of a wave so that every lane holds the total for its column
__device__ __forceinline__ float4 slots_sum(float4 v, int lpr) {
  for (int o = lpr; o < 64; o <<= 1) {
    v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64);
    v.z += __shfl_xor(v.z, o, 64); v.w += __shfl_xor(v.w, o, 64);
  }
  return v;
}

// agg[n] = scale[n] * sum_{e in seg(n)} msg[e] ; e_out[e] = e_in[e] + msg[e].  One wave per node, two nodes per
// workgroup; every lane slot keeps up to SR_U rows of the segment in flight at once (msg and e_in), so a node of <= SR_U *
// rows-per-step edges (24 at H = 128) costs two dependent round trips (segment bounds, rows) instead of one per group of 4.
constexpr int SR_U = 12;

__global__ __launch_bounds__(128) void segment_reduce_kernel(const float* __restrict__ msg,
                                                             const int* __restrict__ rowptr,
                                                             const float* __restrict__ scale,
                                                             float* __restrict__ agg,
                                                             const float* e_in, float* e_out, int N,
                                                             int H) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 2 + (threadIdx.x >> 6);
  if (n >= N) return;
  const RowLanes rl = row_lanes(H, lane);
  const int beg = rowptr[n], end = rowptr[n + 1];
  const float sc = scale ? scale[n] : 1.f;
  for (int cb = 0; cb < H; cb += 256) {           // column blocks (H > 256 only loops)
    const int c = cb + rl.c4;
    const bool cok = c < H;
    const int cc = cok ? c : 0;
    float4 acc = f4zero();
    for (int e0 = beg + rl.slot; e0 < end; e0 += SR_U * rl.rps) {
      float4 m[SR_U], a[SR_U];
#pragma unroll
      for (int u = 0; u < SR_U; ++u) {
        const size_t o = (size_t)min(e0 + u * rl.rps, end - 1) * H + cc;
        m[u] = ld4(msg + o);
        if (e_out) a[u] = ld4(e_in + o);
      }
#pragma unroll
      for (int u = 0; u < SR_U; ++u) {
        const int e = e0 + u * rl.rps;
        if (e < end) {
          acc = f4add(acc, m[u]);
          if (e_out && cok) st4(e_out + (size_t)e * H + c, f4add(a[u], m[u]));
        }
      }
    }
    acc = slots_sum(acc, rl.lpr);
    if (cok && rl.slot == 0) st4(agg + (size_t)n * H + c, make_float4(acc.x * sc, acc.y * sc, acc.z * sc, acc.w * sc));
  }
}

// agg[n] = sum_{j in seg(n)} msg[perm[j]]: the segment sums over a permuted row order (rowptr_src / perm_src: the edges that
// leave node n).  One wave per node, two per workgroup, up to SR_U rows per lane slot in flight (their indices first).
__global__ __launch_bounds__(128) void segment_reduce_perm_kernel(const float* __restrict__ msg, const int* __restrict__ rowptr,
                                                                  const int* __restrict__ perm, float* __restrict__ agg, int N,
                                                                  int H) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 2 + (threadIdx.x >> 6);
  if (n >= N) return;
  const RowLanes rl = row_lanes(H, lane);
  const int beg = rowptr[n], end = rowptr[n + 1];
  for (int cb = 0; cb < H; cb += 256) {
    const int c = cb + rl.c4;
    const bool cok = c < H;
    const int cc = cok ? c : 0;
    float4 acc = f4zero();
    for (int j0 = beg + rl.slot; j0 < end; j0 += SR_U * rl.rps) {
      int ei[SR_U];
#pragma unroll
      for (int u = 0; u < SR_U; ++u) ei[u] = perm[min(j0 + u * rl.rps, end - 1)];
      float4 m[SR_U];
#pragma unroll
      for (int u = 0; u < SR_U; ++u) m[u] = ld4(msg + (size_t)ei[u] * H + cc);
#pragma unroll
      for (int u = 0; u < SR_U; ++u)
        if (j0 + u * rl.rps < end) acc = f4add(acc, m[u]);
    }
    acc = slots_sum(acc, rl.lpr);
    if (cok && rl.slot == 0) st4(agg + (size_t)n * H + c, acc);
  }
}

// S[n] = scale[n] * sum_{e in seg(n)} prelu(xhat[e] * gamma + beta)  (rows of W floats; the segments are contiguous row ranges:
// edges sorted by destination) and R[n] = c_n * bias (Hout floats), c_n = the number of rows of the segment (scale == null:
// scatter_sum) or [segment not empty] (scatter_mean).  With these, scale * sum_e (act_e W^T + b) = S[n] W^T + R[n]: the second
// Linear of the LAST message-passing layer (whose per-edge output nobody reads, DOSTransformer_phonon.py:84 - the edge update
// of the last layer is dead) runs on N aggregated rows instead of E.  One wave per (node, 256-column block), 8 rows in flight,
// rows added in segment order (fixed).
constexpr int AS_U = 8;
__global__ __launch_bounds__(256) void act_segment_sum_kernel(const float* __restrict__ xhat, const int* __restrict__ rowptr,
                                                              const float* __restrict__ scale, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ alpha,
                                                              const float* __restrict__ bias, float* __restrict__ S,
                                                              float* __restrict__ R, int N, int W, int Hout) {
  const int lane = threadIdx.x & 63;
  const int nblk = (W + 255) >> 8;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n = wid / nblk, cb = (wid - n * nblk) << 8;
  if (n >= N) return;
  const int c = cb + lane * 4;
  const bool cok = c < W;
  const int cc = cok ? c : 0;
  const int beg = rowptr[n], end = rowptr[n + 1];
  const float4 g = ld4(gamma + cc), b = ld4(beta + cc);
  const float al = *alpha;
  float4 acc = f4zero();
  for (int e0 = beg; e0 < end; e0 += AS_U) {
    float4 v[AS_U];
#pragma unroll
    for (int u = 0; u < AS_U; ++u) v[u] = ld4(xhat + (size_t)min(e0 + u, end - 1) * W + cc);
#pragma unroll
    for (int u = 0; u < AS_U; ++u) {
      if (e0 + u >= end) break;
      float4 y = make_float4(v[u].x * g.x + b.x, v[u].y * g.y + b.y, v[u].z * g.z + b.z, v[u].w * g.w + b.w);
      y.x = y.x < 0.f ? al * y.x : y.x; y.y = y.y < 0.f ? al * y.y : y.y;
      y.z = y.z < 0.f ? al * y.z : y.z; y.w = y.w < 0.f ? al * y.w : y.w;
      acc = f4add(acc, y);
    }
  }
  const float sc = scale ? scale[n] : 1.f;
  if (cok) st4(S + (size_t)n * W + c, make_float4(acc.x * sc, acc.y * sc, acc.z * sc, acc.w * sc));
  if (cb == 0 && R) {
    const float cn = scale ? (end > beg ? 1.f : 0.f) : (float)(end - beg);
    for (int h = lane * 4; h < Hout; h += 256) {
      const float4 bb = ld4(bias + h);
      st4(R + (size_t)n * Hout + h, make_float4(bb.x * cn, bb.y * cn, bb.z * cn, bb.w * cn));
    }
  }
}

// out[n] = c_n * in[n] (c_n as above): the bias gradient of that Linear is the column sum of these rows
__global__ void seg_count_scale_kernel(const float* __restrict__ in, int ld_in, const int* __restrict__ rowptr, int mean,
                                       float* __restrict__ out, int N, int H) {
  const int h4 = H >> 2;
  const size_t total = (size_t)N * h4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / h4), c = (int)(i % h4) * 4;
    const int deg = rowptr[n + 1] - rowptr[n];
    const float cn = mean ? (deg > 0 ? 1.f : 0.f) : (float)deg;
    const float4 v = ld4(in + (size_t)n * ld_in + c);
    st4(out + (size_t)n * H + c, make_float4(v.x * cn, v.y * cn, v.z * cn, v.w * cn));
  }
}

// dmsg[e] = de_new[e] + scale[dst[e]] * dagg[dst[e]]   (element-wise over E*H/4 float4; de_new rows ld_de floats apart:
// it is the e-block of the previous layer's [E,3H] concat gradient)
__global__ void edge_grad_combine_kernel(const float* __restrict__ de_new, int ld_de, const float* __restrict__ dagg,
                                         int ld_dagg, const int* __restrict__ dst,
                                         const float* __restrict__ scale, float* __restrict__ dmsg, int E,
                                         int H) {
  const int h4 = H >> 2;
  const size_t total = (size_t)E * h4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(i / h4), c = (int)(i % h4) * 4;
    const int d = dst[e];
    const float s = scale ? scale[d] : 1.f;
    float4 v = ld4(dagg + (size_t)d * ld_dagg + c);
    v = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
    if (de_new) v = f4add(v, ld4(de_new + (size_t)e * ld_de + c));
    st4(dmsg + (size_t)e * H + c, v);
  }
}

// Backward of the two gathers x[row], x[col] (+ node residuals).  One wave per node, H/4 lanes per row (a wave covers
// 64/lpr rows per step).  Everything a node needs is fetched in THREE dependent round trips instead of one per group of
// rows: (1) the four segment bounds, (2) up to GB_U rows per lane of the destination segment (contiguous rows of dcat,
// columns [H,2H)) TOGETHER with the source segment's edge ids, (3) the source rows (columns [0,H)) those ids name.  The
// previous version walked each segment 4 rows at a time (a chain of ~10 exposed L2 round trips per node: 20-34 us for
// 450 nodes); segments longer than GB_U * rows-per-step simply loop.  Optional de_out / de_new: the edge residual
// gradient (legacy form; the training programs now carry it on dcat's e-block instead, see DosxGemm.res_col0).
constexpr int GB_U = 12;

__global__ __launch_bounds__(128) void gather_bwd_kernel(const float* __restrict__ dcat,
                                                         const float* __restrict__ dnode, int ld_dnode,
                                                         const float* __restrict__ dx_res,
                                                         const int* __restrict__ rowptr_dst,
                                                         const int* __restrict__ rowptr_src,
                                                         const int* __restrict__ perm_src,
                                                         const float* __restrict__ de_new,
                                                         float* __restrict__ dx, float* __restrict__ de_out,
                                                         int N, int H) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 2 + (threadIdx.x >> 6);
  if (n >= N) return;
  const RowLanes rl = row_lanes(H, lane);
  const size_t ldc = 3 * (size_t)H;
  const int db = rowptr_dst[n], de = rowptr_dst[n + 1];
  const int sb = rowptr_src[n], se = rowptr_src[n + 1];
  for (int cb = 0; cb < H; cb += 256) {
    const int c = cb + rl.c4;
    const bool cok = c < H;
    const int cc = cok ? c : 0;
    float4 acc = f4zero();
    float4 r0 = f4zero(), r1 = f4zero();
    if (rl.slot == 0) {                          // node rows: issued up front, consumed at the very end
      if (dx_res) r0 = ld4(dx_res + (size_t)n * H + cc);
      if (dnode) r1 = ld4(dnode + (size_t)n * ld_dnode + cc);
    }
    int e0 = db + rl.slot, j0 = sb + rl.slot;
    const int estep = GB_U * rl.rps;
    while (e0 < de || j0 < se) {                 // wave-uniform trip count is max over the slots: lanes mask themselves
      int ei[GB_U];
      float4 v1[GB_U];
#pragma unroll
      for (int u = 0; u < GB_U; ++u) ei[u] = perm_src[min(max(j0 + u * rl.rps, 0), max(se - 1, 0))];
#pragma unroll
      for (int u = 0; u < GB_U; ++u) {
        const int e = min(e0 + u * rl.rps, max(de - 1, 0));
        v1[u] = ld4(dcat + (size_t)e * ldc + H + cc);
      }
      if (de_out) {                              // legacy edge-residual part (not on the training path any more)
#pragma unroll
        for (int u = 0; u < GB_U; ++u) {
          const int e = e0 + u * rl.rps;
          if (e < de && cok) {
            float4 v = ld4(dcat + (size_t)e * ldc + 2 * H + c);
            if (de_new) v = f4add(v, ld4(de_new + (size_t)e * H + c));
            st4(de_out + (size_t)e * H + c, v);
          }
        }
      }
      float4 v0[GB_U];
#pragma unroll
      for (int u = 0; u < GB_U; ++u) v0[u] = ld4(dcat + (size_t)ei[u] * ldc + cc);
#pragma unroll
      for (int u = 0; u < GB_U; ++u)
        if (e0 + u * rl.rps < de) acc = f4add(acc, v1[u]);
#pragma unroll
      for (int u = 0; u < GB_U; ++u)
        if (j0 + u * rl.rps < se) acc = f4add(acc, v0[u]);
      e0 += estep;
      j0 += estep;
    }
    acc = slots_sum(acc, rl.lpr);
    if (cok && rl.slot == 0) st4(dx + (size_t)n * H + c, f4add(acc, f4add(r0, r1)));
  }
}

__global__ __launch_bounds__(256) void graph_pool_kernel(const float* __restrict__ x, const int* __restrict__ ptr,
                                                         float* __restrict__ out, int ld_out, int B, int H) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const RowLanes rl = row_lanes(H, lane);
  const int beg = ptr[b], end = ptr[b + 1];
  for (int cb = 0; cb < H; cb += 256) {
    const int c = cb + rl.c4;
    float4 acc = f4zero();
    if (c < H)
      for (int n = beg + rl.slot; n < end; n += rl.rps) acc = f4add(acc, ld4(x + (size_t)n * H + c));
    acc = slots_sum(acc, rl.lpr);
    if (c < H && rl.slot == 0) st4(out + (size_t)b * ld_out + c, acc);
  }
}

__global__ void graph_pool_bwd_kernel(const float* __restrict__ dpool, int ld, const int* __restrict__ node_graph,
                                      float* __restrict__ dx, int N, int H, int accumulate, int num_graphs) {
  const int h4 = H >> 2;
  const size_t total = (size_t)N * h4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / h4), c = (int)(i % h4) * 4;
    const int gph = node_graph[n];
    const bool ghost = num_graphs > 0 && gph >= num_graphs;
    float4 v = ld4(dpool + (size_t)(ghost ? 0 : gph) * ld + c);
    if (ghost) v = f4zero();
    if (accumulate) v = f4add(v, ld4(dx + (size_t)n * H + c));
    st4(dx + (size_t)n * H + c, v);
  }
}

// wave per node: LayerNorm statistics of the node row, normalised row scattered to its dense slot
__global__ __launch_bounds__(256) void dense_normalize_kernel(const float* __restrict__ x,
                                                              const int* __restrict__ dense_row,
                                                              float* __restrict__ kvhat,
                                                              float* __restrict__ rstd_nodes, int N, int H) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float* row = x + (size_t)n * H;
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
  float* o = kvhat + (size_t)dense_row[n] * H;
  for (int c = lane * 4; c < H; c += 256) {
    const float4 v = ld4(row + c);
    st4(o + c, make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd));
  }
  if (lane == 0) rstd_nodes[n] = rstd;
}

// One wave per dense slot (pos, b): the node's normalised row, or zeros for a padded slot / the spare ghost row.
__global__ __launch_bounds__(256) void dense_normalize_slots_kernel(const float* __restrict__ x,
                                                                    const int* __restrict__ graph_ptr,
                                                                    float* __restrict__ kvhat,
                                                                    float* __restrict__ rstd_nodes, int B, int n_max, int H) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r > n_max * B) return;
  float* o = kvhat + (size_t)r * H;
  int n = -1;
  if (r < n_max * B) {
    const int pos = r / B, b = r % B;
    const int beg = graph_ptr[b], end = graph_ptr[b + 1];
    if (beg + pos < end) n = beg + pos;
  }
  if (n < 0) {
    for (int c = lane * 4; c < H; c += 256) st4(o + c, f4zero());
    return;
  }
  const float* row = x + (size_t)n * H;
  float s1 = 0.f;
  for (int c = lane * 4; c < H; c += 256) {
    const float4 v = ld4(row + c);
    s1 += v.x + v.y + v.z + v.w;
  }
  const float mean = wave_sum(s1) / (float)H;
  float s2 = 0.f;
  for (int c = lane * 4; c < H; c += 256) {
    const float4 v = ld4(row + c);
    const float a = v.x - mean, b2 = v.y - mean, c2 = v.z - mean, d = v.w - mean;
    s2 += a * a + b2 * b2 + c2 * c2 + d * d;
  }
  const float rstd = rsqrtf(wave_sum(s2) / (float)H + DOSX_LN_EPS);
  for (int c = lane * 4; c < H; c += 256) {
    const float4 v = ld4(row + c);
    st4(o + c, make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd));
  }
  if (lane == 0) rstd_nodes[n] = rstd;
}

// to_dense_batch alone (DOSTransformer_phonon.py:86-87), without the key normalisation: slot (pos, b) = the node's row or
// zeros.  The keys of the unfused attention path (hidden > 256: functional.encoder_kv_fwd normalises them itself).
__global__ __launch_bounds__(256) void dense_slots_kernel(const float* __restrict__ x, const int* __restrict__ graph_ptr,
                                                          float* __restrict__ dense, int B, int n_max, int H) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_max * B) return;
  const int pos = r / B, b = r % B;
  const int beg = graph_ptr[b], end = graph_ptr[b + 1];
  const float* row = beg + pos < end ? x + (size_t)(beg + pos) * H : nullptr;
  for (int c = lane * 4; c < H; c += 256) st4(dense + (size_t)r * H + c, row ? ld4(row + c) : f4zero());
}

// ... and its backward: dx[n] (+)= ddense[dense_row[n]]; ghost / padding nodes (dense_row == ghost_row) get nothing
__global__ __launch_bounds__(256) void dense_slots_bwd_kernel(const float* __restrict__ ddense, const int* __restrict__ dense_row,
                                                              float* __restrict__ dx, int N, int H, int accumulate,
                                                              int ghost_row) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int dr = dense_row[n];
  for (int c = lane * 4; c < H; c += 256) {
    float4 o = accumulate ? ld4(dx + (size_t)n * H + c) : f4zero();
    if (dr != ghost_row) o = f4add(o, ld4(ddense + (size_t)dr * H + c));
    st4(dx + (size_t)n * H + c, o);
  }
}

// generic "no-affine LN backward" for one row: dx = rstd * (g - mean(g) - xhat * mean(g*xhat))
__device__ __forceinline__ void rownorm_bwd_row(const float* __restrict__ g, const float* __restrict__ xh,
                                                float rstd, float* __restrict__ dx, int H, int lane,
                                                int accumulate, const float* __restrict__ addend = nullptr) {
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane * 4; c < H; c += 256) {
    const float4 a = ld4(g + c), b = ld4(xh + c);
    s1 += a.x + a.y + a.z + a.w;
    s2 += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
  }
  const float m1 = wave_sum(s1) / (float)H, m2 = wave_sum(s2) / (float)H;
  for (int c = lane * 4; c < H; c += 256) {
    const float4 a = ld4(g + c), b = ld4(xh + c);
    float4 o = make_float4(rstd * (a.x - m1 - b.x * m2), rstd * (a.y - m1 - b.y * m2),
                           rstd * (a.z - m1 - b.z * m2), rstd * (a.w - m1 - b.w * m2));
    if (accumulate) o = f4add(o, ld4(dx + c));
    if (addend) o = f4add(ld4(addend + c), o);
    st4(dx + c, o);
  }
}

// dpool / node_graph (optional): the sum-pooling backward of the decoder input rides along - dx[n] += dpool[node_graph[n]]
// (scatter_sum(x, batch) backward, DOSTransformer_phonon.py:178-181), ghost nodes (graph id >= num_graphs) get nothing
__global__ __launch_bounds__(256) void dense_normalize_bwd_kernel(const float* __restrict__ dkvhat,
                                                                  const float* __restrict__ kvhat,
                                                                  const float* __restrict__ rstd_nodes,
                                                                  const int* __restrict__ dense_row,
                                                                  float* __restrict__ dx, int N, int H,
                                                                  int accumulate, int ghost_row,
                                                                  const float* __restrict__ dpool, int ld_dpool,
                                                                  const int* __restrict__ node_graph, int num_graphs) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int dr = dense_row[n];
  const float* add = nullptr;
  if (dpool) {
    const int gph = node_graph[n];
    if (gph < num_graphs) add = dpool + (size_t)gph * ld_dpool;
  }
  if (dr == ghost_row) {            // ghost / padding node: zero gradient, rstd_nodes[n] was never written
    if (!accumulate || add)
      for (int c = lane * 4; c < H; c += 256) {
        float4 o = accumulate ? ld4(dx + (size_t)n * H + c) : f4zero();
        if (add) o = f4add(ld4(add + c), o);
        st4(dx + (size_t)n * H + c, o);
      }
    return;
  }
  const size_t d = (size_t)dr * H;
  rownorm_bwd_row(dkvhat + d, kvhat + d, rstd_nodes[n], dx + (size_t)n * H, H, lane, accumulate, add);
}

__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ x, float* __restrict__ xhat,
                                                      float* __restrict__ rstd_out, int M, int H) {
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
    const float4 v = ld4(row + c);
    st4(xhat + (size_t)r * H + c,
        make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd));
  }
  if (lane == 0) rstd_out[r] = rstd;
}

// out[r] = (res ? res[r] : 0) + a[r] o (mask ? mask[r] : 1), and optionally the LayerNorm statistics (mean, rstd) of the
// out row: the dropout -> add-residual steps of the encoder layer (layers/transformer.py:137-138,145-148) for the
// relu / res dropout path, where they cannot ride in the fused attention / feed-forward epilogues.  One wave per row.
// xhat[e] = rownorm( z[e] + p[src[e]] + q[dst[e]] ): the LayerNorm input of the EdgeModel when its first Linear is FACTORED
// (functional.mlp_ln_fwd): Linear(cat[x[row], x[col], e]) = x[row] Wa^T + x[col] Wb^T + e Wc^T + b with the two node terms
// computed once per NODE (p = x Wa^T, q = x Wb^T: N rows) and gathered here, the edge term z = e Wc^T + b from an E-row GEMM
// on a third of the columns.  One wave per edge row, rows of up to 1024 floats kept in registers (W / 256 float4 per lane).
__global__ __launch_bounds__(256) void gather_add_rownorm_kernel(const float* __restrict__ z, const float* __restrict__ p, int ldp,
                                                                 const float* __restrict__ q, int ldq,
                                                                 const int* __restrict__ src, const int* __restrict__ dst,
                                                                 float* __restrict__ xhat, float* __restrict__ rstd_out, int E, int W) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= E) return;
  const float* pr = p + (size_t)src[r] * ldp;
  const float* qr = q + (size_t)dst[r] * ldq;
  const float* zr = z + (size_t)r * W;
  float4 v[4];
  float s1 = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = lane * 4 + 256 * k;
    v[k] = f4zero();
    if (c < W) {
      v[k] = f4add(ld4(zr + c), f4add(ld4(pr + c), ld4(qr + c)));
      s1 += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
  }
  const float mean = wave_sum(s1) / (float)W;
  float s2 = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (lane * 4 + 256 * k < W) {
      const float a = v[k].x - mean, b = v[k].y - mean, c2 = v[k].z - mean, d = v[k].w - mean;
      s2 += a * a + b * b + c2 * c2 + d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(s2) / (float)W + DOSX_LN_EPS);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = lane * 4 + 256 * k;
    if (c < W)
      st4(xhat + (size_t)r * W + c, make_float4((v[k].x - mean) * rstd, (v[k].y - mean) * rstd, (v[k].z - mean) * rstd,
                                                (v[k].w - mean) * rstd));
  }
  if (lane == 0) rstd_out[r] = rstd;
}

__global__ __launch_bounds__(256) void mask_residual_kernel(const float* __restrict__ a, int lda, const float* __restrict__ mask,
                                                            const float* __restrict__ res, int ldr, float* __restrict__ out,
                                                            int ldo, float* __restrict__ stats, int M, int H) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M) return;
  float s1 = 0.f;
  for (int c = lane * 4; c < H; c += 256) {
    float4 v = ld4(a + (size_t)r * lda + c);
    if (mask) {
      const float4 m = ld4(mask + (size_t)r * H + c);
      v = make_float4(v.x * m.x, v.y * m.y, v.z * m.z, v.w * m.w);
    }
    if (res) v = f4add(v, ld4(res + (size_t)r * ldr + c));
    st4(out + (size_t)r * ldo + c, v);
    s1 += v.x + v.y + v.z + v.w;
  }
  if (stats == nullptr) return;
  const float mean = wave_sum(s1) / (float)H;
  float s2 = 0.f;
  for (int c = lane * 4; c < H; c += 256) {                 // (this lane re-reads what it has just written)
    const float4 v = ld4(out + (size_t)r * ldo + c);
    const float x0 = v.x - mean, x1 = v.y - mean, x2 = v.z - mean, x3 = v.w - mean;
    s2 += x0 * x0 + x1 * x1 + x2 * x2 + x3 * x3;
  }
  const float rstd = rsqrtf(wave_sum(s2) / (float)H + DOSX_LN_EPS);
  if (lane == 0) {
    stats[2 * (size_t)r] = mean;
    stats[2 * (size_t)r + 1] = rstd;
  }
}

__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const float* __restrict__ dxhat,
                                                          const float* __restrict__ xhat,
                                                          const float* __restrict__ rstd, float* __restrict__ dx,
                                                          int M, int H, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M) return;
  rownorm_bwd_row(dxhat + (size_t)r * H, xhat + (size_t)r * H, rstd[r], dx + (size_t)r * H, H, lane, accumulate);
}

// out = (dx_in + rownorm_bwd(dxhat, xhat, rstd)) * act'(y): key-side LN backward + the LeakyReLU backward behind it
__global__ __launch_bounds__(256) void rownorm_bwd_act_kernel(const float* __restrict__ dxhat,
                                                              const float* __restrict__ xhat,
                                                              const float* __restrict__ rstd,
                                                              const float* __restrict__ dx_in,
                                                              const float* __restrict__ y, float slope,
                                                              float* __restrict__ out, int M, int H) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M) return;
  const size_t o = (size_t)r * H;
  const float* g = dxhat + o;
  const float* xh = xhat + o;
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane * 4; c < H; c += 256) {
    const float4 a = ld4(g + c), b = ld4(xh + c);
    s1 += a.x + a.y + a.z + a.w;
    s2 += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
  }
  const float m1 = wave_sum(s1) / (float)H, m2 = wave_sum(s2) / (float)H, rs = rstd[r];
  for (int c = lane * 4; c < H; c += 256) {
    const float4 a = ld4(g + c), b = ld4(xh + c), d = ld4(dx_in + o + c), v = ld4(y + o + c);
    const float4 t = make_float4(d.x + rs * (a.x - m1 - b.x * m2), d.y + rs * (a.y - m1 - b.y * m2),
                                 d.z + rs * (a.z - m1 - b.z * m2), d.w + rs * (a.w - m1 - b.w * m2));
    st4(out + o + c, make_float4(v.x > 0.f ? t.x : slope * t.x, v.y > 0.f ? t.y : slope * t.y,
                                 v.z > 0.f ? t.z : slope * t.z, v.w > 0.f ? t.w : slope * t.w));
  }
}

__global__ void fill_kernel(float* p, float v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void embed_rows_kernel(const float* __restrict__ table, const int* __restrict__ idx,
                                  float* __restrict__ out, int rows, int width) {
  const size_t total = (size_t)rows * width;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / width), c = (int)(i % width);
    out[i] = table[(size_t)idx[r] * width + c];
  }
}

// one block per table row; threads own columns; rows scanned in order -> deterministic
__global__ void embed_rows_bwd_kernel(const float* __restrict__ dout, int ld, const int* __restrict__ idx,
                                      float* __restrict__ dtable, int rows, int width) {
  const int t = blockIdx.x;
  for (int c = threadIdx.x; c < width; c += blockDim.x) {
    float s = 0.f;
    int r = 0;
    // unconditional loads, 8 in flight (a load under `if (idx[r] == t)` made this a chain of `rows` exposed round
    // trips: 9 us for 64 rows); same summation order
    for (; r + 8 <= rows; r += 8) {
      float v[8];
      int k[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { k[u] = idx[r + u]; v[u] = dout[(size_t)(r + u) * ld + c]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) if (k[u] == t) s += v[u];
    }
    for (; r < rows; ++r)
      if (idx[r] == t) s += dout[(size_t)r * ld + c];
    dtable[(size_t)t * width + c] = s;
  }
}

// dst[i, 0:W] (+)= sum_j src[(i*stride_out + j*stride_red), 0:W]   (rows of ld_src / ld_dst floats)
__global__ void reduce_rows_kernel(const float* __restrict__ src, int ld_src, float* dst, int ld_dst, int n_out,
                                   int n_red, int stride_out, int stride_red, int W, int accumulate) {
  const int w4 = W >> 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out * w4) return;
  const int o = i / w4, c = (i % w4) * 4;
  // 8 independent loads in flight per lane (a one-load-per-iteration loop is a chain of n_red exposed
  // L2 round trips: 15-18 us for the 64..128-row sums of the heads' backward); fixed summation order
  float4 acc = f4zero();
  const float* base = src + (size_t)o * stride_out * ld_src + c;
  const size_t step = (size_t)stride_red * ld_src;
  int j = 0;
  for (; j + 8 <= n_red; j += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ld4(base + (size_t)(j + u) * step);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = f4add(acc, v[u]);
  }
  for (; j < n_red; ++j) acc = f4add(acc, ld4(base + (size_t)j * step));
  if (accumulate) acc = f4add(acc, ld4(dst + (size_t)o * ld_dst + c));
  st4(dst + (size_t)o * ld_dst + c, acc);
}

// out = dy * (y > 0 ? 1 : slope)    (LeakyReLU / ReLU backward from the saved OUTPUT)
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float slope, float* out,
                               size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 d = ld4(dy + 4 * i), v = ld4(y + 4 * i);
    st4(out + 4 * i, make_float4(v.x > 0.f ? d.x : slope * d.x, v.y > 0.f ? d.y : slope * d.y,
                                 v.z > 0.f ? d.z : slope * d.z, v.w > 0.f ? d.w : slope * d.w));
  }
}

inline int grid_1d(size_t total, int block) {
  size_t g = (total + block - 1) / block;
  if (g > 2048 * 4) g = 2048 * 4;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

// (row kernels: H / 4 lanes per row below 256 columns - a divisor of 64 - and column blocks of 256 with a guarded tail above)
#define CHECK_H(H) DOSX_CHECK_ARG((H) > 0 && ((H) & 3) == 0 && ((H) >= 256 || (256 % (H)) == 0), \
                                  "%s: H=%d must be a power-of-two multiple of 4 (or any multiple of 4 from 256)", __func__, (H))

extern "C" int dosx_edge_feat_sh1(const float* edge_vec, float* edge_attr, int E, float r_max, dosx_stream_t stream) {
  if (E <= 0) return 0;
  DOSX_CHECK_ARG(edge_vec && edge_attr && r_max > 0.f, "dosx_edge_feat_sh1: bad args");
  hipLaunchKernelGGL(edge_feat_kernel, dim3(ceil_div(E, 256)), dim3(256), 0, to_stream(stream), edge_vec, edge_attr, E,
                     1.f / r_max);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_edge_embed_sh1(const float* edge_vec, const float* w0, const float* b0, float* edge_attr, float* z, int E,
                                   int H, float r_max, dosx_stream_t stream) {
  if (E <= 0) return 0;
  DOSX_CHECK_ARG(edge_vec && w0 && b0 && edge_attr && z && r_max > 0.f && H > 0 && (H & 3) == 0, "dosx_edge_embed_sh1: bad args");
  hipLaunchKernelGGL(edge_embed_kernel, dim3(grid_1d((size_t)E * (H / 4), 256)), dim3(256), 0, to_stream(stream), edge_vec,
                     w0, b0, edge_attr, z, E, H, 1.f / r_max);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_segment_reduce(const float* msg, const int32_t* rowptr, const float* scale, float* agg,
                                   const float* e_in, float* e_out, int N, int E, int H, dosx_stream_t stream) {
  (void)E;
  if (N <= 0) return 0;
  CHECK_H(H);
  DOSX_CHECK_ARG(msg && rowptr && agg && (!e_out || e_in), "dosx_segment_reduce: bad args");
  hipLaunchKernelGGL(segment_reduce_kernel, dim3(ceil_div(N, 2)), dim3(128), 0, to_stream(stream), msg, rowptr, scale,
                     agg, e_in, e_out, N, H);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_segment_reduce_perm(const float* msg, const int32_t* rowptr, const int32_t* perm, float* agg, int N, int E,
                                        int H, dosx_stream_t stream) {
  (void)E;
  if (N <= 0) return 0;
  CHECK_H(H);
  DOSX_CHECK_ARG(msg && rowptr && perm && agg, "dosx_segment_reduce_perm: bad args");
  hipLaunchKernelGGL(segment_reduce_perm_kernel, dim3(ceil_div(N, 2)), dim3(128), 0, to_stream(stream), msg, rowptr, perm, agg, N, H);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_act_segment_sum(const float* xhat, const int32_t* rowptr, const float* scale, const float* gamma,
                                    const float* beta, const float* alpha, const float* bias, float* S, float* R, int N, int W,
                                    int Hout, dosx_stream_t stream) {
  if (N <= 0) return 0;
  DOSX_CHECK_ARG(xhat && rowptr && gamma && beta && alpha && S && (!R || bias), "dosx_act_segment_sum: bad args");
  DOSX_CHECK_ARG(W > 0 && (W & 3) == 0 && (Hout & 3) == 0, "dosx_act_segment_sum: widths %d / %d must be multiples of 4", W, Hout);
  const int nblk = (W + 255) >> 8;
  hipLaunchKernelGGL(act_segment_sum_kernel, dim3(ceil_div(N * nblk, 4)), dim3(256), 0, to_stream(stream), xhat, rowptr, scale,
                     gamma, beta, alpha, bias, S, R, N, W, Hout);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_seg_count_scale(const float* in, int ld_in, const int32_t* rowptr, int mean, float* out, int N, int H,
                                    dosx_stream_t stream) {
  if (N <= 0) return 0;
  DOSX_CHECK_ARG(in && rowptr && out && (H & 3) == 0 && (ld_in & 3) == 0, "dosx_seg_count_scale: bad args");
  const size_t total = (size_t)N * (H >> 2);
  hipLaunchKernelGGL(seg_count_scale_kernel, dim3((unsigned)std::min<size_t>(ceil_div(total, (size_t)256), 2048)), dim3(256), 0,
                     to_stream(stream), in, ld_in, rowptr, mean, out, N, H);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_edge_grad_combine(const float* de_new, int ld_de_new, const float* dagg, int ld_dagg, const int32_t* dst,
                                      const float* scale, float* dmsg, int E, int H, dosx_stream_t stream) {
  if (E <= 0) return 0;
  CHECK_H(H);
  DOSX_CHECK_ARG(dagg && dst && dmsg && (ld_dagg & 3) == 0 && (!de_new || (ld_de_new >= H && (ld_de_new & 3) == 0)),
                 "dosx_edge_grad_combine: bad args");
  hipLaunchKernelGGL(edge_grad_combine_kernel, dim3(grid_1d((size_t)E * (H / 4), 256)), dim3(256), 0,
                     to_stream(stream), de_new, ld_de_new, dagg, ld_dagg, dst, scale, dmsg, E, H);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_gather_bwd(const float* dcat, const float* dnode, int ld_dnode, const float* dx_res,
                               const int32_t* rowptr_dst, const int32_t* rowptr_src, const int32_t* perm_src,
                               const float* de_new, float* dx, float* de_out, int N, int E, int H,
                               dosx_stream_t stream) {
  (void)E;
  if (N <= 0) return 0;
  CHECK_H(H);
  DOSX_CHECK_ARG(dcat && rowptr_dst && rowptr_src && perm_src && dx && (ld_dnode & 3) == 0, "dosx_gather_bwd: bad args");
  hipLaunchKernelGGL(gather_bwd_kernel, dim3(ceil_div(N, 2)), dim3(128), 0, to_stream(stream), dcat, dnode, ld_dnode,
                     dx_res, rowptr_dst, rowptr_src, perm_src, de_new, dx, de_out, N, H);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_graph_pool(const float* x, const int32_t* graph_ptr, float* out, int ld_out, int B, int H,
                               dosx_stream_t stream) {
  if (B <= 0) return 0;
  CHECK_H(H);
  DOSX_CHECK_ARG(x && graph_ptr && out && (ld_out & 3) == 0, "dosx_graph_pool: bad args");
  hipLaunchKernelGGL(graph_pool_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, to_stream(stream), x, graph_ptr, out, ld_out,
                     B, H);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_graph_pool_bwd(const float* dpool, int ld_dpool, const int32_t* node_graph, float* dx, int N, int H,
                                   int accumulate, int num_graphs, dosx_stream_t stream) {
  if (N <= 0) return 0;
  CHECK_H(H);
  DOSX_CHECK_ARG(dpool && node_graph && dx && (ld_dpool & 3) == 0, "dosx_graph_pool_bwd: bad args");
  hipLaunchKernelGGL(graph_pool_bwd_kernel, dim3(grid_1d((size_t)N * (H / 4), 256)), dim3(256), 0, to_stream(stream),
                     dpool, ld_dpool, node_graph, dx, N, H, accumulate, num_graphs);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_dense_normalize(const float* x, const int32_t* dense_row, float* kvhat, float* rstd_nodes, int N,
                                    int H, int dense_rows, dosx_stream_t stream) {
  CHECK_H(H);
  DOSX_CHECK_ARG(kvhat && dense_rows >= 0, "dosx_dense_normalize: bad args");
  if (dense_rows > 0) {
    const size_t n = (size_t)dense_rows * H;      // (own fill kernel: 2.5 us where the runtime's memset kernel takes 5.5)
    hipLaunchKernelGGL(fill_kernel, dim3(grid_1d(n, 256)), dim3(256), 0, to_stream(stream), kvhat, 0.f, n);
    DOSX_LAUNCH_CHECK();
  }
  if (N <= 0) return 0;
  DOSX_CHECK_ARG(x && dense_row && rstd_nodes, "dosx_dense_normalize: bad args");
  hipLaunchKernelGGL(dense_normalize_kernel, dim3(ceil_div(N, 4)), dim3(256), 0, to_stream(stream), x, dense_row, kvhat,
                     rstd_nodes, N, H);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_dense_normalize_slots(const float* x, const int32_t* graph_ptr, float* kvhat, float* rstd_nodes, int B,
                                          int n_max, int H, dosx_stream_t stream) {
  CHECK_H(H);
  DOSX_CHECK_ARG(x && graph_ptr && kvhat && rstd_nodes && B > 0 && n_max >= 0, "dosx_dense_normalize_slots: bad args");
  hipLaunchKernelGGL(dense_normalize_slots_kernel, dim3(ceil_div(n_max * B + 1, 4)), dim3(256), 0, to_stream(stream), x,
                     graph_ptr, kvhat, rstd_nodes, B, n_max, H);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_dense_slots(const float* x, const int32_t* graph_ptr, float* dense, int B, int n_max, int H,
                                dosx_stream_t stream) {
  if (B * n_max <= 0) return 0;
  CHECK_H(H);
  DOSX_CHECK_ARG(x && graph_ptr && dense, "dosx_dense_slots: bad args");
  hipLaunchKernelGGL(dense_slots_kernel, dim3(ceil_div(n_max * B, 4)), dim3(256), 0, to_stream(stream), x, graph_ptr, dense, B,
                     n_max, H);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_dense_slots_bwd(const float* ddense, const int32_t* dense_row, float* dx, int N, int H, int accumulate,
                                    int ghost_row, dosx_stream_t stream) {
  if (N <= 0) return 0;
  CHECK_H(H);
  DOSX_CHECK_ARG(ddense && dense_row && dx, "dosx_dense_slots_bwd: bad args");
  hipLaunchKernelGGL(dense_slots_bwd_kernel, dim3(ceil_div(N, 4)), dim3(256), 0, to_stream(stream), ddense, dense_row, dx, N,
                     H, accumulate, ghost_row);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_dense_normalize_bwd(const float* dkvhat, const float* kvhat, const float* rstd_nodes,
                                        const int32_t* dense_row, float* dx, int N, int H, int accumulate, int ghost_row,
                                        dosx_stream_t stream) {
  if (N <= 0) return 0;
  CHECK_H(H);
  DOSX_CHECK_ARG(dkvhat && kvhat && rstd_nodes && dense_row && dx, "dosx_dense_normalize_bwd: bad args");
  hipLaunchKernelGGL(dense_normalize_bwd_kernel, dim3(ceil_div(N, 4)), dim3(256), 0, to_stream(stream), dkvhat, kvhat,
                     rstd_nodes, dense_row, dx, N, H, accumulate, ghost_row, (const float*)nullptr, 0, (const int*)nullptr, 0);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_dense_normalize_pool_bwd(const float* dkvhat, const float* kvhat, const float* rstd_nodes,
                                             const int32_t* dense_row, const float* dpool, int ld_dpool,
                                             const int32_t* node_graph, int num_graphs, float* dx, int N, int H,
                                             int accumulate, int ghost_row, dosx_stream_t stream) {
  if (N <= 0) return 0;
  CHECK_H(H);
  DOSX_CHECK_ARG(dkvhat && kvhat && rstd_nodes && dense_row && dx && dpool && node_graph && num_graphs > 0 && (ld_dpool & 3) == 0,
                 "dosx_dense_normalize_pool_bwd: bad args");
  hipLaunchKernelGGL(dense_normalize_bwd_kernel, dim3(ceil_div(N, 4)), dim3(256), 0, to_stream(stream), dkvhat, kvhat,
                     rstd_nodes, dense_row, dx, N, H, accumulate, ghost_row, dpool, ld_dpool, node_graph, num_graphs);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_gather_add_rownorm(const float* z, const float* p, int ldp, const float* q, int ldq, const int32_t* src,
                                       const int32_t* dst, float* xhat, float* rstd, int E, int W, dosx_stream_t stream) {
  if (E <= 0) return 0;
  DOSX_CHECK_ARG(z && p && q && src && dst && xhat && rstd && W > 0 && (W & 3) == 0 && W <= 1024 && (ldp & 3) == 0 && (ldq & 3) == 0,
                 "dosx_gather_add_rownorm: bad args (W=%d: multiple of 4, <= 1024)", W);
  hipLaunchKernelGGL(gather_add_rownorm_kernel, dim3(ceil_div(E, 4)), dim3(256), 0, to_stream(stream), z, p, ldp, q, ldq, src, dst,
                     xhat, rstd, E, W);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_rownorm(const float* x, float* xhat, float* rstd, int M, int H, dosx_stream_t stream) {
  if (M <= 0) return 0;
  CHECK_H(H);
  DOSX_CHECK_ARG(x && xhat && rstd, "dosx_rownorm: bad args");
  hipLaunchKernelGGL(rownorm_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, to_stream(stream), x, xhat, rstd, M, H);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_mask_residual(const float* a, int lda, const float* mask, const float* res, int ldr, float* out, int ldo,
                                  float* stats, int M, int H, dosx_stream_t stream) {
  if (M <= 0) return 0;
  CHECK_H(H);
  DOSX_CHECK_ARG(a && out && (lda & 3) == 0 && (ldo & 3) == 0 && (!res || (ldr & 3) == 0), "dosx_mask_residual: bad args");
  hipLaunchKernelGGL(mask_residual_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, to_stream(stream), a, lda, mask, res, ldr, out,
                     ldo, stats, M, H);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_rownorm_bwd_act(const float* dxhat, const float* xhat, const float* rstd, const float* dx_in,
                                    const float* y, float slope, float* out, int M, int H, dosx_stream_t stream) {
  if (M <= 0) return 0;
  CHECK_H(H);
  DOSX_CHECK_ARG(dxhat && xhat && rstd && dx_in && y && out, "dosx_rownorm_bwd_act: bad args");
  hipLaunchKernelGGL(rownorm_bwd_act_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, to_stream(stream), dxhat, xhat, rstd,
                     dx_in, y, slope, out, M, H);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_rownorm_bwd(const float* dxhat, const float* xhat, const float* rstd, float* dx, int M, int H,
                                int accumulate, dosx_stream_t stream) {
  if (M <= 0) return 0;
  CHECK_H(H);
  DOSX_CHECK_ARG(dxhat && xhat && rstd && dx, "dosx_rownorm_bwd: bad args");
  hipLaunchKernelGGL(rownorm_bwd_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, to_stream(stream), dxhat, xhat, rstd, dx, M,
                     H, accumulate);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_fill(float* p, float value, int64_t n, dosx_stream_t stream) {
  if (n <= 0) return 0;
  DOSX_CHECK_ARG(p, "dosx_fill: null");
  hipLaunchKernelGGL(fill_kernel, dim3(grid_1d((size_t)n, 256)), dim3(256), 0, to_stream(stream), p, value, (size_t)n);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_embed_rows(const float* table, const int32_t* idx, float* out, int rows, int width,
                               dosx_stream_t stream) {
  if (rows <= 0) return 0;
  DOSX_CHECK_ARG(table && idx && out && width > 0, "dosx_embed_rows: bad args");
  hipLaunchKernelGGL(embed_rows_kernel, dim3(grid_1d((size_t)rows * width, 256)), dim3(256), 0, to_stream(stream), table,
                     idx, out, rows, width);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_embed_rows_bwd(const float* dout, int ld_dout, const int32_t* idx, float* dtable, int rows,
                                   int table_rows, int width, dosx_stream_t stream) {
  if (table_rows <= 0) return 0;
  DOSX_CHECK_ARG(dout && idx && dtable && width > 0, "dosx_embed_rows_bwd: bad args");
  hipLaunchKernelGGL(embed_rows_bwd_kernel, dim3(table_rows), dim3(128), 0, to_stream(stream), dout, ld_dout, idx, dtable,
                     rows, width);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_reduce_rows(const float* src, int ld_src, float* dst, int ld_dst, int n_out, int n_red,
                                int stride_out, int stride_red, int width, int accumulate, dosx_stream_t stream) {
  if (n_out <= 0) return 0;
  DOSX_CHECK_ARG(src && dst && (width & 3) == 0 && (ld_src & 3) == 0 && (ld_dst & 3) == 0 && width > 0,
                 "dosx_reduce_rows: bad args");
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(ceil_div(n_out * (width / 4), 256)), dim3(256), 0, to_stream(stream), src,
                     ld_src, dst, ld_dst, n_out, n_red, stride_out, stride_red, width, accumulate);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_act_bwd(const float* dy, const float* y, float slope, float* out, int64_t n, dosx_stream_t stream) {
  if (n <= 0) return 0;
  DOSX_CHECK_ARG(dy && y && out && (n & 3) == 0, "dosx_act_bwd: bad args (n must be a multiple of 4)");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_1d((size_t)n / 4, 256)), dim3(256), 0, to_stream(stream), dy, y, slope, out,
                     (size_t)n / 4);
  DOSX_LAUNCH_CHECK();
  return 0;
}
