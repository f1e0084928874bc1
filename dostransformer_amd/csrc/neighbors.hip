// Periodic neighbour list for a set of crystals: the graph-building step of the phonon data path,
// `utils.py:249-303` (build_data), whose edges come from ASE's `neighbor_list("ijS", a=structure, cutoff=r_max,
// self_interaction=True)` (`utils.py:267`; ASE is not in the tree and not pinned — its documented behaviour is
// restated here): every ordered triple (i, j, S) of atoms i, j of the same crystal and an integer lattice shift S
// with  | pos[j] - pos[i] + S·cell | < cutoff ;  (i, i, 0) only when self_interaction is set; images (i, i, S≠0)
// always.  `edge_vec` is that difference vector, summed in the reference's order (`utils.py:271-273`).
//
// Work decomposition: one thread per ordered atom pair (i, j) of a crystal (Σ n_c² pairs).  The shifts a pair can
// have are bounded per axis by  |f_k + S_k| < cutoff·|g_k|  (f = fractional coordinates of pos[j]-pos[i], g_k the
// k-th column of cell⁻¹: the fractional coordinate of any vector d is d·g_k ≤ |d||g_k|), so each thread walks its own
// minimal integer box and tests the exact distance — positions need not be wrapped into the cell.  Two passes
// (count, then fill at the exclusive prefix sum the caller computes), so the output order is deterministic:
// by crystal, then i, then j, then shift lexicographically — i.e. grouped by source atom like ASE's.
#include "common.h"

// membership and edge_vec are compared bit-for-bit with the oracle: no fused multiply-adds anywhere in this file
// (plain operators below, NOT __dmul_rn / __dadd_rn: the HIP header versions of those are compiled with contraction
// allowed and fuse after inlining)
#pragma clang fp contract(off)

namespace {

struct NlGeom {
  double L[9];      // lattice rows
  double G[9];      // inverse (columns g_k = G[.][k])
  double R[3];      // cutoff * |g_k|
};

__device__ __forceinline__ void nl_geometry(const double* __restrict__ cell, double cutoff, NlGeom& q) {
#pragma unroll
  for (int k = 0; k < 9; ++k) q.L[k] = cell[k];
  const double* L = q.L;
  const double c00 = L[4] * L[8] - L[5] * L[7], c01 = L[5] * L[6] - L[3] * L[8], c02 = L[3] * L[7] - L[4] * L[6];
  const double det = L[0] * c00 + L[1] * c01 + L[2] * c02;
  const double id = 1.0 / det;
  q.G[0] = c00 * id; q.G[1] = (L[2] * L[7] - L[1] * L[8]) * id; q.G[2] = (L[1] * L[5] - L[2] * L[4]) * id;
  q.G[3] = c01 * id; q.G[4] = (L[0] * L[8] - L[2] * L[6]) * id; q.G[5] = (L[2] * L[3] - L[0] * L[5]) * id;
  q.G[6] = c02 * id; q.G[7] = (L[1] * L[6] - L[0] * L[7]) * id; q.G[8] = (L[0] * L[4] - L[1] * L[3]) * id;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    q.R[k] = cutoff * sqrt(q.G[k] * q.G[k] + q.G[3 + k] * q.G[3 + k] + q.G[6 + k] * q.G[6 + k]);
}

__device__ __forceinline__ int nl_crystal_of(const long long* __restrict__ pair_ptr, int C, long long p) {
  int lo = 0, hi = C;                                   // last c with pair_ptr[c] <= p
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (pair_ptr[mid] <= p) lo = mid; else hi = mid;
  }
  return lo;
}

// d = (pos_j - pos_i) + ((s0*L0 + s1*L1) + s2*L2), no contraction: the reference's (and the oracle's) rounding
__device__ __forceinline__ void nl_vec(const double* dp, const double* L, int s0, int s1, int s2, double* d) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double sh = ((double)s0 * L[a] + (double)s1 * L[3 + a]) + (double)s2 * L[6 + a];
    d[a] = dp[a] + sh;
  }
}

template <bool FILL>
__global__ void neighbor_pairs_kernel(const double* __restrict__ pos, const double* __restrict__ cell,
                                      const int* __restrict__ atom_ptr, const long long* __restrict__ pair_ptr, int C,
                                      long long n_pairs, double cutoff, int self_interaction, int pbc_mask,
                                      int* __restrict__ pair_count, const long long* __restrict__ pair_off,
                                      int* __restrict__ crystal, int* __restrict__ src, int* __restrict__ dst,
                                      int* __restrict__ shift, double* __restrict__ edge_vec) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  const int c = nl_crystal_of(pair_ptr, C, p);
  const int a0 = atom_ptr[c], n = atom_ptr[c + 1] - a0;
  const int lp = (int)(p - pair_ptr[c]);
  const int i = lp / n, j = lp - i * n;
  NlGeom q;
  nl_geometry(cell + (size_t)c * 9, cutoff, q);
  double dp[3], f[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) dp[a] = pos[(size_t)(a0 + j) * 3 + a] - pos[(size_t)(a0 + i) * 3 + a];
  int lo[3], hi[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    f[k] = dp[0] * q.G[k] + dp[1] * q.G[3 + k] + dp[2] * q.G[6 + k];
    const bool periodic = (pbc_mask >> k) & 1;
    // one extra shift either side: the box bound is evaluated in floating point, the membership test below is exact
    lo[k] = periodic ? (int)ceil(-f[k] - q.R[k]) - 1 : 0;
    hi[k] = periodic ? (int)floor(-f[k] + q.R[k]) + 1 : 0;
  }
  const double rc2 = cutoff * cutoff;
  long long out = FILL ? pair_off[p] : 0;
  int cnt = 0;
  for (int s0 = lo[0]; s0 <= hi[0]; ++s0)
    for (int s1 = lo[1]; s1 <= hi[1]; ++s1)
      for (int s2 = lo[2]; s2 <= hi[2]; ++s2) {
        if (i == j && s0 == 0 && s1 == 0 && s2 == 0 && !self_interaction) continue;
        double d[3];
        nl_vec(dp, q.L, s0, s1, s2, d);
        const double r2 = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
        if (!(r2 < rc2)) continue;
        if (FILL) {
          crystal[out] = c; src[out] = i; dst[out] = j;
          shift[out * 3 + 0] = s0; shift[out * 3 + 1] = s1; shift[out * 3 + 2] = s2;
          edge_vec[out * 3 + 0] = d[0]; edge_vec[out * 3 + 1] = d[1]; edge_vec[out * 3 + 2] = d[2];
          ++out;
        }
        ++cnt;
      }
  if (!FILL) pair_count[p] = cnt;
}

int nl_check(const char* who, const void* pos, const void* cell, const void* atom_ptr, const void* pair_ptr, int C,
             long long n_pairs, double cutoff) {
  DOSX_CHECK_ARG(C > 0 && n_pairs >= 0, "%s: bad sizes C=%d n_pairs=%lld", who, C, n_pairs);
  DOSX_CHECK_ARG(n_pairs < (1ll << 31) * 256, "%s: too many atom pairs (%lld)", who, n_pairs);
  DOSX_CHECK_ARG(pos && cell && atom_ptr && pair_ptr, "%s: null input", who);
  DOSX_CHECK_ARG(cutoff > 0.0, "%s: cutoff must be positive (%g)", who, cutoff);
  return 0;
}

}  // namespace

extern "C" int dosx_neighbor_count(const double* pos, const double* cell, const int* atom_ptr, const long long* pair_ptr,
                                   int C, long long n_pairs, double cutoff, int self_interaction, int pbc_mask,
                                   int* pair_count, dosx_stream_t stream) {
  if (int rc = nl_check("dosx_neighbor_count", pos, cell, atom_ptr, pair_ptr, C, n_pairs, cutoff)) return rc;
  DOSX_CHECK_ARG(pair_count || n_pairs == 0, "dosx_neighbor_count: null output");
  if (n_pairs == 0) return 0;
  const int threads = 256;
  neighbor_pairs_kernel<false><<<(unsigned)((n_pairs + threads - 1) / threads), threads, 0, to_stream(stream)>>>(
      pos, cell, atom_ptr, pair_ptr, C, n_pairs, cutoff, self_interaction, pbc_mask, pair_count, nullptr, nullptr, nullptr,
      nullptr, nullptr, nullptr);
  DOSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dosx_neighbor_fill(const double* pos, const double* cell, const int* atom_ptr, const long long* pair_ptr,
                                  int C, long long n_pairs, double cutoff, int self_interaction, int pbc_mask,
                                  const long long* pair_off, int* crystal, int* src, int* dst, int* shift,
                                  double* edge_vec, dosx_stream_t stream) {
  if (int rc = nl_check("dosx_neighbor_fill", pos, cell, atom_ptr, pair_ptr, C, n_pairs, cutoff)) return rc;
  if (n_pairs == 0) return 0;
  DOSX_CHECK_ARG(pair_off && crystal && src && dst && shift && edge_vec, "dosx_neighbor_fill: null output / offsets");
  const int threads = 256;
  neighbor_pairs_kernel<true><<<(unsigned)((n_pairs + threads - 1) / threads), threads, 0, to_stream(stream)>>>(
      pos, cell, atom_ptr, pair_ptr, C, n_pairs, cutoff, self_interaction, pbc_mask, nullptr, pair_off, crystal, src, dst,
      shift, edge_vec);
  DOSX_LAUNCH_CHECK();
  return 0;
}
