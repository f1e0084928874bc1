// Device-side graph metadata ("CSR build") for a batch that arrives as PyG-style index tensors
// (counterpart of what torch_geometric's collate + to_dense_batch + torch_scatter derive on the
// fly: `DOSTransformer_phonon.py:48-56,86,209`).  Everything the kernels of this library index with is derived
// from `edge_index [2,E]` (int64, any order) and `batch [N]` (int64, non-decreasing) ON THE DEVICE, stream-ordered,
// with no host round trip:
//     edges stably sorted by destination (so every aggregation is a contiguous, atomic-free, bit-reproducible
//     segment sum), the permutation that sort applied, CSR row pointers by destination and by source, the
//     source-sorted inverse index for the gather backward, per-graph node ranges, the dense (pos, graph) slot of
//     every node, 1/in-degree, and the largest crystal (device scalar).
// The two stable sorts are rocPRIM LSD radix sorts (a plain library primitive); the rest are the small kernels below.
#include <string.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"
#include <stdlib.h>

namespace {

__global__ void csr_keys_kernel(const long long* __restrict__ edge_index, int E, int* __restrict__ kdst, int* __restrict__ iota) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  kdst[e] = (int)edge_index[(size_t)E + e];
  iota[e] = e;
}

// after the sort by destination: src in the new order, the 64-bit permutation for the caller's edge tensors
__global__ void csr_permute_kernel(const long long* __restrict__ edge_index, const int* __restrict__ perm, int E,
                                   int* __restrict__ src, long long* __restrict__ edge_perm, int* __restrict__ iota) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int p = perm[e];
  src[e] = (int)edge_index[p];
  if (edge_perm) edge_perm[e] = p;
  iota[e] = e;
}

__device__ __forceinline__ int lower_bound_i32(const int* __restrict__ a, int n, int key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ int lower_bound_i64(const long long* __restrict__ a, int n, long long key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// row pointers from the sorted key arrays (binary search per node: no atomics, no scan)
__global__ void csr_rowptr_kernel(const int* __restrict__ dst_sorted, const int* __restrict__ src_sorted, int E, int N,
                                  int* __restrict__ rowptr_dst, int* __restrict__ rowptr_src) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n > N) return;
  rowptr_dst[n] = lower_bound_i32(dst_sorted, E, n);
  rowptr_src[n] = lower_bound_i32(src_sorted, E, n);
}

__global__ void csr_graph_ptr_kernel(const long long* __restrict__ batch, int N, int B, int* __restrict__ graph_ptr) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > B) return;
  graph_ptr[b] = lower_bound_i64(batch, N, (long long)b);
}

__global__ void csr_nodes_kernel(const long long* __restrict__ batch, const int* __restrict__ graph_ptr,
                                 const int* __restrict__ rowptr_dst, int N, int B, int* __restrict__ node_graph,
                                 int* __restrict__ dense_row, float* __restrict__ inv_deg) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int b = (int)batch[n];
  node_graph[n] = b;
  dense_row[n] = (n - graph_ptr[b]) * B + b;
  const int deg = rowptr_dst[n + 1] - rowptr_dst[n];
  inv_deg[n] = 1.f / (float)(deg > 1 ? deg : 1);
}

__global__ void csr_nmax_kernel(const int* __restrict__ graph_ptr, int B, int* __restrict__ n_max) {
  __shared__ int red[256];
  int m = 0;
  for (int b = threadIdx.x; b < B; b += 256) m = max(m, graph_ptr[b + 1] - graph_ptr[b]);
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) n_max[0] = red[0];
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t sort_temp_bytes(int E) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const int*)nullptr, (int*)nullptr, (const int*)nullptr, (int*)nullptr,
                                  (size_t)(E > 0 ? E : 1), 0, 32, (hipStream_t)0);
  return bytes;
}

}  // namespace

extern "C" int dosx_csr_workspace_bytes(int E, size_t* bytes) {
  DOSX_CHECK_ARG(bytes != nullptr && E >= 0, "dosx_csr_workspace_bytes: bad args");
  // [kdst | iota | keys_out] (3 x E int32) + the radix sort's own temporary storage
  *bytes = 3 * align256((size_t)(E > 0 ? E : 1) * 4) + align256(sort_temp_bytes(E));
  return 0;
}

extern "C" int dosx_csr_build(const long long* edge_index, const long long* batch, int N, int E, int B, int* src, int* dst,
                              long long* edge_perm, int* rowptr_dst, int* perm_src, int* rowptr_src, int* graph_ptr,
                              int* node_graph, int* dense_row, float* inv_deg, int* n_max, void* workspace, size_t ws_bytes,
                              dosx_stream_t stream) {
  DOSX_CHECK_ARG(N >= 0 && E >= 0 && B >= 0, "dosx_csr_build: bad sizes N=%d E=%d B=%d", N, E, B);
  DOSX_CHECK_ARG(src && dst && rowptr_dst && perm_src && rowptr_src && graph_ptr && node_graph && dense_row && inv_deg,
                 "dosx_csr_build: null output");
  DOSX_CHECK_ARG((E == 0 || edge_index) && (N == 0 || batch), "dosx_csr_build: null input");
  size_t need = 0;
  dosx_csr_workspace_bytes(E, &need);
  DOSX_CHECK_ARG(workspace && ws_bytes >= need, "dosx_csr_build: workspace %zu < %zu bytes", ws_bytes, need);
  hipStream_t s = to_stream(stream);
  char* w = static_cast<char*>(workspace);
  const size_t seg = align256((size_t)(E > 0 ? E : 1) * 4);
  int* kdst = reinterpret_cast<int*>(w);
  int* iota = reinterpret_cast<int*>(w + seg);
  int* keys_out = reinterpret_cast<int*>(w + 2 * seg);
  void* tmp = w + 3 * seg;
  size_t tmp_bytes = ws_bytes - 3 * seg;
  if (E > 0) {
    const int g = ceil_div(E, 256);
    hipLaunchKernelGGL(csr_keys_kernel, dim3(g), dim3(256), 0, s, edge_index, E, kdst, iota);
    // stable sort by destination: dst <- sorted keys, perm_src (scratch) <- original position of every sorted edge
    hipError_t e1 = rocprim::radix_sort_pairs(tmp, tmp_bytes, kdst, dst, iota, perm_src, (size_t)E, 0, 32, s);
    DOSX_CHECK_ARG(e1 == hipSuccess, "dosx_csr_build: radix sort failed: %s", hipGetErrorString(e1));
    hipLaunchKernelGGL(csr_permute_kernel, dim3(g), dim3(256), 0, s, edge_index, perm_src, E, src, edge_perm, iota);
    // stable sort of the dst-ordered edge ids by source: perm_src = ids (dst-sorted numbering) ordered by src
    hipError_t e2 = rocprim::radix_sort_pairs(tmp, tmp_bytes, src, keys_out, iota, perm_src, (size_t)E, 0, 32, s);
    DOSX_CHECK_ARG(e2 == hipSuccess, "dosx_csr_build: radix sort failed: %s", hipGetErrorString(e2));
  }
  hipLaunchKernelGGL(csr_rowptr_kernel, dim3(ceil_div(N + 1, 256)), dim3(256), 0, s, dst, keys_out, E, N, rowptr_dst, rowptr_src);
  hipLaunchKernelGGL(csr_graph_ptr_kernel, dim3(ceil_div(B + 1, 256)), dim3(256), 0, s, batch, N, B, graph_ptr);
  if (N > 0)
    hipLaunchKernelGGL(csr_nodes_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, s, batch, graph_ptr, rowptr_dst, N, B, node_graph,
                       dense_row, inv_deg);
  if (n_max) hipLaunchKernelGGL(csr_nmax_kernel, dim3(1), dim3(256), 0, s, graph_ptr, B, n_max);
  DOSX_LAUNCH_CHECK();
  return 0;
}

// =============================================================================================================
// Collate on the device ("cache per-crystal CSR, collate by offset-add on GPU").
// The whole dataset is resident: per crystal its edges are already sorted by destination and its local CSR is cached
// (loader.DeviceDataset).  A batch = the selected crystals' segments copied next to each other with the node / edge
// offsets of the batch added: because node ids of crystal b are all smaller than those of crystal b+1, the
// concatenation IS the batch's destination-sorted edge list, its CSR and its source-sorted inverse index.
// =============================================================================================================
namespace {

__device__ __forceinline__ int seg_of(const int* __restrict__ ptr, int B, int i) {     // largest b with ptr[b] <= i
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (ptr[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ void collate_nodes_kernel(const int* __restrict__ sel, const int* __restrict__ node_ptr_all,
                                     const int* __restrict__ out_node_ptr, const int* __restrict__ out_edge_ptr, int B, int N, int E,
                                     const int* __restrict__ rowptr_dst_all, const int* __restrict__ rowptr_src_all,
                                     const float* __restrict__ inv_deg_all, long long* __restrict__ batch,
                                     int* __restrict__ node_graph, int* __restrict__ dense_row, float* __restrict__ inv_deg,
                                     int* __restrict__ rowptr_dst, int* __restrict__ rowptr_src, int* __restrict__ node_row) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n > N) return;
  if (n == N) {
    rowptr_dst[N] = E;
    rowptr_src[N] = E;
    return;
  }
  const int b = seg_of(out_node_ptr, B, n), c = sel[b];
  const int l = n - out_node_ptr[b], s = node_ptr_all[c] + l;
  batch[n] = b;
  node_graph[n] = b;
  dense_row[n] = l * B + b;
  inv_deg[n] = inv_deg_all[s];
  rowptr_dst[n] = rowptr_dst_all[s + c] + out_edge_ptr[b];       // the cached row pointers hold n_c + 1 entries per crystal
  rowptr_src[n] = rowptr_src_all[s + c] + out_edge_ptr[b];
  node_row[n] = s;
}

__global__ void collate_edges_kernel(const int* __restrict__ sel, const int* __restrict__ edge_ptr_all,
                                     const int* __restrict__ out_node_ptr, const int* __restrict__ out_edge_ptr, int B, int E,
                                     const int* __restrict__ src_all, const int* __restrict__ dst_all,
                                     const int* __restrict__ perm_src_all, int* __restrict__ src, int* __restrict__ dst,
                                     int* __restrict__ perm_src, long long* __restrict__ edge_index, int* __restrict__ edge_row) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int b = seg_of(out_edge_ptr, B, e), c = sel[b];
  const int l = e - out_edge_ptr[b], s = edge_ptr_all[c] + l;
  const int no = out_node_ptr[b];
  const int sv = src_all[s] + no, dv = dst_all[s] + no;
  src[e] = sv;
  dst[e] = dv;
  perm_src[e] = perm_src_all[s] + out_edge_ptr[b];
  edge_index[e] = sv;
  edge_index[(size_t)E + e] = dv;
  edge_row[e] = s;
}

}  // namespace

extern "C" int dosx_collate(const int* sel, const int* node_ptr_all, const int* edge_ptr_all, const int* out_node_ptr,
                            const int* out_edge_ptr, int B, int N, int E, const int* src_all, const int* dst_all,
                            const int* perm_src_all, const int* rowptr_dst_all, const int* rowptr_src_all,
                            const float* inv_deg_all, long long* batch, long long* edge_index, int* src, int* dst, int* perm_src,
                            int* rowptr_dst, int* rowptr_src, int* node_graph, int* dense_row, float* inv_deg, int* node_row,
                            int* edge_row, dosx_stream_t stream) {
  DOSX_CHECK_ARG(B > 0 && N >= 0 && E >= 0, "dosx_collate: bad sizes B=%d N=%d E=%d", B, N, E);
  DOSX_CHECK_ARG(sel && node_ptr_all && edge_ptr_all && out_node_ptr && out_edge_ptr, "dosx_collate: null index input");
  DOSX_CHECK_ARG(batch && src && dst && perm_src && rowptr_dst && rowptr_src && node_graph && dense_row && inv_deg && node_row &&
                     (E == 0 || (edge_index && edge_row)),
                 "dosx_collate: null output");
  hipStream_t s = to_stream(stream);
  hipLaunchKernelGGL(collate_nodes_kernel, dim3(ceil_div(N + 1, 256)), dim3(256), 0, s, sel, node_ptr_all, out_node_ptr, out_edge_ptr,
                     B, N, E, rowptr_dst_all, rowptr_src_all, inv_deg_all, batch, node_graph, dense_row, inv_deg, rowptr_dst,
                     rowptr_src, node_row);
  if (E > 0)
    hipLaunchKernelGGL(collate_edges_kernel, dim3(ceil_div(E, 256)), dim3(256), 0, s, sel, edge_ptr_all, out_node_ptr, out_edge_ptr, B,
                       E, src_all, dst_all, perm_src_all, src, dst, perm_src, edge_index, edge_row);
  DOSX_LAUNCH_CHECK();
  return 0;
}


// =============================================================================================================
// Collate STRAIGHT INTO the static buffers of a shape bucket (train.Trainer.step_dataset): what dosx_collate + the row
// gathers + batch.pad_batch + the slot copy did in ~40 launches, in three.  Real nodes / edges as in dosx_collate; the
// ghost tail reproduces batch.pad_batch exactly: ghost nodes carry zero features, belong to no crystal (node_graph = B,
// dense slot = the spare row n_max*B), ghost edges are self loops on the first ghost node with zero features.
// =============================================================================================================
namespace {

__global__ void collate_pad_nodes_kernel(const DosxCollate d) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n > d.N_pad) return;
  if (n <= d.B) d.graph_ptr[n] = d.out_node_ptr[n];     // (B <= N < N_pad: these threads exist)
  if (n >= d.N) {                                        // ghost tail (+ the closing row-pointer entries)
    const int rp = n == d.N ? d.E : d.E_pad;
    d.rowptr_dst[n] = rp;
    d.rowptr_src[n] = rp;
    if (n < d.N_pad) {
      d.node_graph[n] = d.B;
      d.dense_row[n] = d.n_max * d.B;
      d.inv_deg[n] = n == d.N ? 1.f / (float)max(d.E_pad - d.E, 1) : 1.f;
      d.node_row[n] = -1;
    }
    return;
  }
  const int b = seg_of(d.out_node_ptr, d.B, n), c = d.sel[b];
  const int l = n - d.out_node_ptr[b], s = d.node_ptr_all[c] + l;
  d.node_graph[n] = b;
  d.dense_row[n] = l * d.B + b;
  d.inv_deg[n] = d.inv_deg_all[s];
  d.rowptr_dst[n] = d.rowptr_dst_all[s + c] + d.out_edge_ptr[b];
  d.rowptr_src[n] = d.rowptr_src_all[s + c] + d.out_edge_ptr[b];
  d.node_row[n] = s;
}

__global__ void collate_pad_edges_kernel(const DosxCollate d) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= d.E_pad) return;
  if (e >= d.E) {
    d.src[e] = d.N;
    d.dst[e] = d.N;
    d.perm_src[e] = e;
    d.edge_row[e] = -1;
    return;
  }
  const int b = seg_of(d.out_edge_ptr, d.B, e), c = d.sel[b];
  const int l = e - d.out_edge_ptr[b], s = d.edge_ptr_all[c] + l;
  const int no = d.out_node_ptr[b];
  d.src[e] = d.src_all[s] + no;
  d.dst[e] = d.dst_all[s] + no;
  d.perm_src[e] = d.perm_src_all[s] + d.out_edge_ptr[b];
  d.edge_row[e] = s;
}

// node-aligned row tiles of the message GEMM (DosxGemm EPI_SEGSUM): the selected crystals' precomputed local tilings with
// the batch offsets added, then the ghost tiles (batch.pad_seg_tiles is the host twin of this kernel)
__global__ void collate_pad_tiles_kernel(const DosxCollate d) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t > d.T) return;
  const int real = d.out_tile_ptr[d.B];
  int eb, nb, pi = 0;
  if (t < real) {
    const int b = seg_of(d.out_tile_ptr, d.B, t), c = d.sel[b], l = t - d.out_tile_ptr[b];
    eb = d.out_edge_ptr[b] + d.tile_e_all[d.tile_off_all[c] + l];
    nb = d.out_node_ptr[b] + d.tile_n_all[d.tile_off_all[c] + l];
    pi = d.tile_p_all[d.tile_off_all[c] + l];
  } else {
    const int k = t - real;
    eb = min(d.E + k * d.tile_rows, d.E_pad);
    nb = k == 0 ? d.N : d.N_pad;
  }
  d.seg_tile[t] = eb;
  d.seg_tile[d.T + 1 + t] = nb;
  d.seg_tile[2 * (d.T + 1) + t] = pi;
}

// The same table by the HOST's rule - greedy over the whole batch, tiles may span crystal boundaries (batch.seg_tiles_host is the
// twin, tile for tile) - in one workgroup (round 6).  The crystal-aligned table above ends every crystal with a partial tile:
// 256 tiles for 64 seven-atom crystals where the greedy packing makes 212, i.e. 20 % more workgroups (and weight streams) for the
// EdgeModel launches of every step that collates on the device.  A greedy packing is a chain - each tile starts where the previous
// one ends - so: (1) every node k computes, as if a tile started at it, where that tile (or, for a node of more than `rows`
// incoming edges, its run of chunk tiles) ends: nxt[k]; (2) the starts actually reached from node 0 are marked by pointer
// doubling (round r marks what is 2^r .. 2^(r+1) - 1 tiles away); (3) a prefix sum over the marked nodes' tile counts places them.
// LDS: four int arrays of N + 1 entries; N <= GT_MAXN, larger batches keep the crystal-aligned table.
constexpr int GT_MAXN = 8192, GT_THREADS = 1024;
__global__ __launch_bounds__(GT_THREADS) void collate_greedy_tiles_kernel(const DosxCollate d) {
  extern __shared__ int gsm[];
  const int N = d.N, E = d.E, rows = d.tile_rows, tid = threadIdx.x;
  int* rp = gsm;                 // [N + 1] destination row pointers of the real nodes
  int* nxt = rp + (N + 1);       // [N + 1] start of the tile behind the tile(s) that start at k; later the jump table
  int* jm2 = nxt + (N + 1);      // [N + 1] second jump buffer
  int* mark = jm2 + (N + 1);     // [N + 1] 1: a tile starts at node k; later the exclusive prefix of the tile counts
  __shared__ int wsum[GT_THREADS / 64 + 1];
  for (int k = tid; k <= N; k += GT_THREADS) rp[k] = d.rowptr_dst[k];
  __syncthreads();
  auto last_le = [&](const int lim) {            // largest j in [0, N] with rp[j] <= lim (rp is non-decreasing, rp[0] = 0 <= lim)
    int lo = 0, hi = N;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (rp[mid] <= lim) lo = mid; else hi = mid - 1;
    }
    return lo;
  };
  auto ntiles = [&](const int k) {               // tiles that start at node k: 1, or its chunk tiles
    const int deg = rp[k + 1] - rp[k];
    return deg > rows ? (deg + rows - 1) / rows : 1;
  };
  for (int k = tid; k < N; k += GT_THREADS) {
    const int deg = rp[k + 1] - rp[k];
    int j;
    if (deg > rows) {
      const int nfull = deg / rows, r = deg - nfull * rows;
      if (r) {                                   // the remainder's tile goes on with whole nodes
        j = min(max(last_le(rp[k] + nfull * rows + rows), k + 1), N);
      } else {
        // full last chunk: when the tile behind it would hold no rows (isolated nodes in front of the next over-full node, or
        // of the end), the chunk tile takes those nodes (seg_tiles_host: a tile without rows breaks seg_tile_bound's pairing)
        j = k + 1;
        if (j < N && rp[j + 1] - rp[j] <= rows) {
          const int j2 = min(max(last_le(rp[j] + rows), j + 1), N);
          if (rp[j2] == rp[j]) j = j2;
        }
      }
    } else {
      j = min(max(last_le(rp[k] + rows), k + 1), N);
    }
    nxt[k] = j;
    mark[k] = k == 0 ? 1 : 0;
  }
  if (tid == 0) { nxt[N] = N; mark[N] = 0; }
  __syncthreads();
  int* ja = nxt;
  int* jb = jm2;
  for (int span = 1; span < N; span <<= 1) {     // after the round: every start fewer than 2 * span tiles from node 0 is marked
    for (int k = tid; k <= N; k += GT_THREADS) jb[k] = ja[ja[k]];
    __syncthreads();
    for (int k = tid; k < N; k += GT_THREADS)
      if (mark[k] && ja[k] < N) mark[ja[k]] = 1;           // (several writers, one value)
    __syncthreads();
    int* t_ = ja; ja = jb; jb = t_;
  }
  // exclusive prefix of the marked nodes' tile counts: a contiguous run of nodes per thread, wave scan, wave totals
  const int per = (N + GT_THREADS - 1) / GT_THREADS, k0 = tid * per, k1 = min(k0 + per, N);
  int mine = 0;
  for (int k = k0; k < k1; ++k) mine += mark[k] ? ntiles(k) : 0;
  int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if ((tid & 63) >= o) incl += v;
  }
  if ((tid & 63) == 63) wsum[tid >> 6] = incl;
  __syncthreads();
  if (tid == 0) {
    int a = 0;
    for (int w = 0; w < GT_THREADS / 64; ++w) { const int v = wsum[w]; wsum[w] = a; a += v; }
    wsum[GT_THREADS / 64] = a;
  }
  __syncthreads();
  const int real = wsum[GT_THREADS / 64];
  int base = wsum[tid >> 6] + incl - mine;
  int* seb = d.seg_tile;
  int* snb = d.seg_tile + (d.T + 1);
  int* spi = d.seg_tile + 2 * (d.T + 1);
  for (int k = k0; k < k1; ++k) {
    if (!mark[k]) continue;
    const int deg = rp[k + 1] - rp[k];
    if (deg > rows) {
      const int nc = (deg + rows - 1) / rows;
      for (int i = 0; i < nc && base + i <= d.T; ++i) {
        seb[base + i] = rp[k] + i * rows;
        snb[base + i] = k;
        spi[base + i] = (i << 16) | nc;
      }
      base += nc;
    } else {
      if (base <= d.T) { seb[base] = rp[k]; snb[base] = k; spi[base] = 0; }
      ++base;
    }
  }
  // the boundary behind the last real tile, then the ghost edges in `rows`-row tiles (the first owns every ghost node), then
  // empty slots: batch.pad_seg_tiles
  for (int t = real + tid; t <= d.T; t += GT_THREADS) {
    const int kk = t - real;
    seb[t] = min(E + kk * rows, d.E_pad);
    snb[t] = kk == 0 ? N : d.N_pad;
    spi[t] = 0;
  }
}

// feature rows: x [N_pad,Fa], edge features [E_pad,Fe], per-crystal targets [B,S] / globals [B,n_glob] / system [B]
__global__ void collate_pad_gather_kernel(const DosxCollate d) {
  const size_t nx = (size_t)d.N_pad * d.Fa, ne = (size_t)d.E_pad * d.Fe, nt = (size_t)d.B * d.S, ng = (size_t)d.B * d.n_glob;
  const size_t total = nx + ne + nt + ng + d.B;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    if (i < nx) {
      const int n = (int)(i / d.Fa), c = (int)(i % d.Fa), r = d.node_row[n];
      d.x[i] = r >= 0 ? d.x_all[(size_t)r * d.Fa + c] : 0.f;
    } else if (i < nx + ne) {
      const size_t k = i - nx;
      const int e = (int)(k / d.Fe), c = (int)(k % d.Fe), r = d.edge_row[e];
      d.edge_feat[k] = r >= 0 ? d.edge_feat_all[(size_t)r * d.Fe + c] : 0.f;
    } else if (i < nx + ne + nt) {
      const size_t k = i - nx - ne;
      d.target[k] = d.target_all[(size_t)d.sel[k / d.S] * d.S + k % d.S];
    } else if (i < nx + ne + nt + ng) {
      const size_t k = i - nx - ne - nt;
      d.glob[k] = d.glob_all[(size_t)d.sel[k / d.n_glob] * d.n_glob + k % d.n_glob];
    } else {
      const int b = (int)(i - nx - ne - nt - ng);
      d.system[b] = d.system_all[d.sel[b]];
    }
  }
}

}  // namespace

extern "C" int dosx_collate_padded(const DosxCollate* dp, dosx_stream_t stream) {
  DOSX_CHECK_ARG(dp != nullptr, "dosx_collate_padded: null descriptor");
  const DosxCollate& d = *dp;
  DOSX_CHECK_ARG(d.B > 0 && d.N >= d.B && d.E >= 0 && d.N_pad > d.N && d.E_pad >= d.E && d.n_max > 0,
                 "dosx_collate_padded: bad sizes B=%d N=%d E=%d N_pad=%d E_pad=%d n_max=%d (needs >= 1 ghost node)", d.B, d.N, d.E,
                 d.N_pad, d.E_pad, d.n_max);
  DOSX_CHECK_ARG(d.Fa > 0 && d.Fe > 0 && d.S >= 0 && d.n_glob >= 0, "dosx_collate_padded: bad widths");
  DOSX_CHECK_ARG(d.sel && d.out_node_ptr && d.out_edge_ptr && d.node_ptr_all && d.edge_ptr_all && d.src_all && d.dst_all &&
                     d.perm_src_all && d.rowptr_dst_all && d.rowptr_src_all && d.inv_deg_all && d.x_all && d.edge_feat_all &&
                     d.system_all && (d.S == 0 || d.target_all) && (d.n_glob == 0 || d.glob_all),
                 "dosx_collate_padded: null input");
  DOSX_CHECK_ARG(d.x && d.edge_feat && d.system && (d.S == 0 || d.target) && (d.n_glob == 0 || d.glob) && d.src && d.dst &&
                     d.perm_src && d.rowptr_dst && d.rowptr_src && d.graph_ptr && d.node_graph && d.dense_row && d.inv_deg &&
                     d.node_row && d.edge_row,
                 "dosx_collate_padded: null output");
  hipStream_t s = to_stream(stream);
  hipLaunchKernelGGL(collate_pad_nodes_kernel, dim3(ceil_div(d.N_pad + 1, 256)), dim3(256), 0, s, d);
  if (d.E_pad > 0) hipLaunchKernelGGL(collate_pad_edges_kernel, dim3(ceil_div(d.E_pad, 256)), dim3(256), 0, s, d);
  if (d.seg_tile) {
    DOSX_CHECK_ARG(d.T > 0 && d.tile_rows > 0 && d.out_tile_ptr && d.tile_off_all && d.tile_e_all && d.tile_n_all && d.tile_p_all,
                   "dosx_collate_padded: seg_tile needs T, tile_rows and the per-crystal tile tables");
    static int greedy = -1;
    if (greedy < 0) { const char* e = getenv("DOSX_COLLATE_GREEDY_TILES"); greedy = e ? atoi(e) : 1; }
    if (greedy && d.N <= GT_MAXN) {
      const size_t smem = sizeof(int) * 4 * (size_t)(d.N + 1);
      static bool attr_set = false;
      if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&collate_greedy_tiles_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 4 * (GT_MAXN + 1));
        attr_set = true;
      }
      hipLaunchKernelGGL(collate_greedy_tiles_kernel, dim3(1), dim3(GT_THREADS), smem, s, d);      // (behind collate_pad_nodes_kernel: rowptr_dst)
    } else {
      hipLaunchKernelGGL(collate_pad_tiles_kernel, dim3(ceil_div(d.T + 1, 256)), dim3(256), 0, s, d);
    }
  }
  const size_t total = (size_t)d.N_pad * d.Fa + (size_t)d.E_pad * d.Fe + (size_t)d.B * (d.S + d.n_glob + 1);
  size_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(collate_pad_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, s, d);
  DOSX_LAUNCH_CHECK();
  return 0;
}
