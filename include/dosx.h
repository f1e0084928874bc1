/* dosx.h — C ABI of libdosx.so: the MI355X (gfx950) hot path of DOSTransformer.
 *
 * The reference (HeewoongNoh/DOSTransformer) is pure Python and has NO FFI / plugin
 * boundary of its own; its hot path is a chain of implicit PyTorch /
 * torch_scatter / PyG kernels.  This header is the boundary the build introduces:
 * every entry point replaces a group of those implicit launches, and the Python
 * modules in dostransformer_amd/ (same class names / ctor signatures / state_dict
 * keys as the reference's embedder_phDOS, embedder_eDOS and layers packages) bind
 * them through ctypes (see INTEGRATION.md for the stub).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory unless noted;
 *    all floating point data is fp32 row-major, all index data int32;
 *  - nothing is allocated, freed or retained by the library: the caller passes
 *    outputs and scratch ("partials") buffers;
 *  - every call is asynchronous on the given hipStream_t, re-entrant and
 *    graph-capturable (no hidden synchronisation, no host<->device copies);
 *  - return value 0 on success, negative on error; dosx_last_error() gives a
 *    thread-local message.
 */
#ifndef DOSX_H
#define DOSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dosx_stream_t; /* hipStream_t */

/* Row indirection used for gathered operands and remapped outputs:
 *   t = (r / d) * m + (r % d) * c + off ;  row = idx ? idx[t] : t
 * identity: d = 1<<30, m = 0, c = 1, off = 0.  r % B: d=B,m=0,c=1.  r / B: d=B,m=1,c=0. */
typedef struct DosxRowMap {
  int32_t d, m, c, off;
  const int32_t* idx;
} DosxRowMap;

/* One K-segment of a (virtually concatenated, row-gathered) matrix operand.
 * Replaces torch.cat([x[row], x[col], edge_attr], 1) (DOSTransformer_phonon.py:166,194). */
typedef struct DosxSeg {
  const float* p;
  int32_t ld;    /* row stride in floats */
  int32_t width; /* columns taken from this segment */
  DosxRowMap map;
} DosxSeg;

/* Prologue applied to the A operand while it is staged into LDS. */
enum {
  DOSX_PRO_NONE = 0,
  DOSX_PRO_PRELU = 1,    /* a = z >= 0 ? z : alpha*z            (nn.PReLU, DOSTransformer_phonon.py:129) */
  DOSX_PRO_LN_PRELU = 2, /* a = prelu(xhat*gamma + beta)          (LayerNorm+PReLU of Edge/NodeModel, :193,204) */
  DOSX_PRO_ROWLN = 3     /* a = (x-mean[r])*rstd[r]*gamma + beta  (pre-norm LN, layers/transformer.py:141-142) */
};

/* Epilogues of dosx_gemm (all but PLAIN/BIAS_ACT need the whole row in one tile: N <= 512). */
enum {
  DOSX_EPI_BIAS_ACT = 0,     /* out = act(acc + bias) [+ res]; act: 0 none, 1 relu, 2 leaky(slope)   */
  DOSX_EPI_LN = 1,           /* out = xhat = LN_noaffine(acc + bias); aux_out = rstd[M]              */
  DOSX_EPI_PRELU_LN_BWD = 2, /* acc = dL/d prelu(y), y = xhat*g+b  ->  out = dL/dz (pre-LN)          */
  DOSX_EPI_RELU_MASK = 3,    /* out = acc * (aux > 0)                                                 */
  DOSX_EPI_ROWLN_BWD = 4,    /* acc = dL/d LN(x) -> out = res + dL/dx ; x = aux, stats = aux_stats    */
  DOSX_EPI_PRELU_BWD = 5,    /* out = acc * (aux >= 0 ? 1 : alpha); partial dalpha                    */
  DOSX_EPI_SEGSUM = 6,       /* message GEMM of a GNN layer with the aggregation in its epilogue (a5 + a6):
                                msg = acc + bias stays in LDS;  seg_agg[n] = seg_scale[n] * sum_{e in seg(n)} msg[e];
                                out = msg + res (the edge residual e' = e + msg) unless out is NULL.  Row tiles are
                                node-aligned: workgroup t owns rows [seg_tile[t], seg_tile[t+1]) (<= 48) = the whole
                                destination segments of the nodes [seg_tile[T+1+t], seg_tile[T+2+t]), T = seg_ntiles    */
  DOSX_EPI_PRELU_LN_BWD_SEG = 7 /* EPI_PRELU_LN_BWD on the node-aligned row tiles of EPI_SEGSUM (same seg_* fields, w_layout 1,
                                N <= 256): out = dL/dz as before AND seg_agg[n] = seg_scale[n] * sum_{e in seg(n)} out[e] - the
                                destination-node sums of dz that the FACTORED first Linear of the EdgeModel needs for its
                                weight and input gradients (DOSTransformer_phonon.py:190-197: cat[x[row], x[col], e] W1^T =
                                (x Wa^T)[row] + (x Wb^T)[col] + e Wc^T, so dWb = (sum_{e: col(e)=n} dz_e)^T x).  One partial
                                row per TILE (seg_ntiles rows, empty tiles write zeros).                                  */
};

/* C[M,N] = epilogue( prologue(A)[M,K] * B ),  fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * w_layout 0: B = W^T with W[N,K] row-major  (nn.Linear forward,  y = x W^T + b)
 * w_layout 1: B = W   with W[K,N] row-major  (nn.Linear backward, dx = dy W)         */
typedef struct DosxGemm {
  int32_t M, N, K;
  int32_t nseg;
  DosxSeg a[3];
  int32_t pro;
  const float* pro_gamma;
  const float* pro_beta;
  const float* pro_alpha; /* device scalar */
  const float* pro_stats; /* [M,2] mean,rstd (ROWLN) */
  const float* w;
  int32_t ldw;
  int32_t w_layout;
  int32_t epi;
  int32_t act;
  float act_slope;
  const float* bias;
  float* out;
  int32_t ldo;
  DosxRowMap out_map;
  const float* res;
  int32_t ldr;
  DosxRowMap res_map;
  float* stats_out;   /* optional [M,2]: mean, rstd of the final output rows (needs N <= tile) */
  float* aux_out;     /* EPI_LN: rstd [M] */
  const float* aux;   /* backward epilogues: xhat / h / x / z, [M, ldaux] */
  int32_t ldaux;
  const float* aux_stats; /* rstd [M] (PRELU_LN_BWD) or [M,2] mean,rstd (ROWLN_BWD) */
  const float* epi_gamma;
  const float* epi_beta;
  const float* epi_alpha;
  float* partials;    /* per-workgroup partial sums: row wg = [dgamma(N) | dbeta(N) | dalpha] */
  int32_t partial_ld;
  /* EPI_SEGSUM only */
  const int32_t* seg_tile;   /* [3][seg_ntiles + 1]: row (edge) boundaries, node boundaries, chunk info of the node-aligned tiles.
                                chunk info != 0 (= chunk << 16 | chunks): the tile's first node has more than 48 incoming edges
                                and the tile owns that chunk of its rows (all but the last chunk are full tiles without
                                other nodes); the chunk sums of a node are published to seg_part and added in chunk order by
                                the last arriving tile (ticket on seg_cnt) - deterministic, no second launch */
  int32_t seg_ntiles;
  const int32_t* seg_rowptr; /* [nodes + 1] CSR by destination */
  const float* seg_scale;    /* [nodes] 1/max(in-degree,1) for scatter_mean, NULL for scatter_sum */
  float* seg_agg;            /* [nodes, N] */
  float* seg_part;           /* [seg_ntiles, N] scratch for the chunk sums of over-full nodes.  REQUIRED (like seg_cnt): the table lives in
                                device memory, so the host cannot know whether a tile carries chunk info; a call without it is
                                refused (-22) instead of silently dropping an over-full node's aggregate */
  int32_t* seg_cnt;          /* [seg_ntiles] arrival counters, zero before the launch, zero again after it (like DosxWgrad.counters) */
  int32_t res_col0;   /* EPI_BIAS_ACT: the residual is added to the output columns [res_col0, N) only, res column c
                         to output column res_col0 + c (0 = all columns).  The backward of the edge residual
                         e += e' (DOSTransformer_phonon.py:84) rides on the e-block of the [E,3H] concat gradient. */
  float* norm_out;    /* EPI_BIAS_ACT, optional (N <= 512): ALSO the LayerNorm-normalised output rows without affine,
                         (y - mean) * rstd, eps 1e-5, at the same (mapped) rows and leading dimension as `out` - the
                         stale keys of the self-attention encoder are the normalised head outputs (layers/transformer.py:
                         131-134 on DOSTransformer_phonon.py:97's input), so the heads' GEMMs write them and no
                         dosx_rownorm launch follows */
  float* norm_rstd;   /* with norm_out: rstd per OUTPUT row [rows of out] */
  int32_t res_pre;    /* EPI_BIAS_ACT: 1 = the residual rows are added BEFORE the activation, out = act(A W + bias + res[res_map(r)]).
                         The K-segments of an output head that are constant along the energy axis - cat[x, graph(, prompt)]
                         (DOSTransformer_phonon.py:93-95,105-109) repeats the pooled crystal vector for every energy - are
                         multiplied once per crystal and enter the per-energy GEMM as such a row-mapped pre-activation term. */
  /* EPI_LN, optional: two GATHERED row addends in front of the LayerNorm statistics (round 5),
   *     xhat[r] = LN_noaffine( acc[r] + bias + add_p[add_ip[r]] + add_q[add_iq[r]] ),   rows of add_p / add_q: N floats, stride ld_add.
   * The EdgeModel's first Linear on cat[x[row], x[col], e] (DOSTransformer_phonon.py:190-197) FACTORED: the two node products
   * P = x Wa^T, Q = x Wb^T are N-row GEMMs (dosx_gemm_pair), this call multiplies the E edge rows by Wc only (K = H instead of
   * 3H) and gathers P[row(e)] + Q[col(e)] from the L2-resident [nodes, 4H] product in its epilogue. */
  const float* add_p;
  const int32_t* add_ip;
  const float* add_q;
  const int32_t* add_iq;
  int32_t ld_add;
  /* w_layout 1 with nseg > 1, optional: K-segment s of A multiplies its OWN block of W - rows restart at 0 and the block is
   * shifted by s * w_seg_off floats:  B(k, n) = w[(k - k0_s) * ldw + s * w_seg_off + n]  (0 = one [K,N] matrix as usual).
   * The node part of the factored EdgeModel input gradient, dx = [S | D] . [Wa ; Wb] with Wa = W1[:, :H], Wb = W1[:, H:2H] two
   * column blocks of ONE [2H,3H] matrix: segments S, D of width 2H, ldw = 3H, w_seg_off = H.  Needs aligned operands and
   * segment widths that are multiples of 32. */
  int32_t w_seg_off;
} DosxGemm;

/* exact number of workgroup rows dosx_gemm writes into `partials` for this epilogue: ceil(M/32) for
 * the row-wise epilogues, ceil(M/32)*ceil(N/128) for the element-wise PRELU_BWD epilogue. */
/* Experiment switch (round 4, default 0 = off; also DOSX_SLIVER_MAX_GF in the environment): plain dgrad GEMMs (w_layout 1, no
 * prologue / bias / activation, one segment, identity out / residual maps) of up to `gf` GFLOP run on a vector-ALU kernel whose
 * workgroups (4 waves, 56 VGPRs, 8.5 KB of LDS) fit next to two resident weight-gradient workgroups. */
int dosx_set_sliver_max_gf(double gf);
int dosx_gemm_partial_rows(int M, int N, int epi);
int dosx_gemm(const DosxGemm* g, dosx_stream_t stream);
/* Two independent GEMMs in ONE launch when they share a tile configuration (same N, W[N,K], no prologue, the plain epilogue,
 * 4-float aligned operands), one after the other otherwise - the two output heads `fc` / `fc_prompt`
 * (DOSTransformer_phonon.py:93-95,105-109; DOSTransformer.py:64-66,76-80) write disjoint rows of one tensor and are each one
 * partial round of workgroups: together still one round.  Tile height chosen for the rows of both.
 * Round 5: also two EPI_PRELU_BWD problems (W[K,N], N = 128) of different heights - the node and the edge encoder's backward at
 * the tail of a step (DOSTransformer_phonon.py:129-130,141-142 differentiated): each at ITS tile height in one grid (16-row tiles
 * for the few hundred node rows, 48-row tiles for the edge rows), partial rows numbered per problem as dosx_gemm would. */
int dosx_gemm_pair(const DosxGemm* a, const DosxGemm* b, dosx_stream_t stream);
/* diagnostic: the device symbol dosx_gemm launches for this descriptor, as a profiler prints it
 * ("gemm_kernel<RT, NTW, WL, PRO, VEC, EPI>"), written to the HOST buffer buf[n]. */
int dosx_gemm_kernel_name(const DosxGemm* g, char* buf, int n);

/* dW of y = A W^T + b (nn.Linear weight gradient), split over the M rows into `nsplit` partial sums
 *   part[s][n][k] = sum_{m in split s} dY[m][n] * prologue(A)[m][k],   bias part[s][n] = sum_m dY[m][n].
 * `nsplit` must come from dosx_wgrad_splits().  Two modes:
 *   dst == NULL (slab mode): the partial sums are left in slab[nsplit][N][K] / slab_bias[nsplit][N] for
 *     dosx_reduce_partials;
 *   dst != NULL (finished mode): the kernel itself finishes dW: every workgroup publishes its 64x64 partial tile
 *     (write-through stores), draws a ticket on its tile's counter, and the LAST arriver of a tile adds the nsplit partial
 *     tiles in split order (a fixed order whoever arrives last: bitwise reproducible, no float atomics) and writes
 *     dst[N][K] (row stride K; `accumulate`: dst +=) and dst_bias[N].  No second launch, no slab re-read through HBM.
 *     slab is then private scratch of dosx_wgrad_scratch_floats(N, K, nsplit) floats (tile-major; may be NULL when
 *     nsplit == 1), slab_bias of nsplit * 64*ceil(N/64) floats, `counters` one int32 per 64x64 tile
 *     (dosx_wgrad_tiles(N, K)), ZERO before the first launch that uses them and zero again when the launch is done
 *     (the last arriver resets its counter), so a caller allocates them zeroed once. */
typedef struct DosxWgrad {
  int32_t M, N, K;
  DosxSeg dy;        /* width = N */
  int32_t nseg;
  DosxSeg a[3];
  int32_t pro;
  const float* pro_gamma;
  const float* pro_beta;
  const float* pro_alpha;
  const float* pro_stats;
  float* slab;       /* slab mode: [nsplit, N, K]; finished mode: scratch (see above) */
  float* slab_bias;  /* slab mode: [nsplit, N] or NULL; finished mode: scratch or NULL */
  int32_t nsplit;
  int32_t accumulate; /* finished mode: dst (+)= */
  float* dst;        /* finished mode: [N, K] */
  float* dst_bias;   /* finished mode: [N] or NULL (needs slab_bias) */
  int32_t* counters; /* finished mode: [dosx_wgrad_tiles(N, K)] */
  int32_t ldd;       /* finished mode: row stride of dst in floats (0 = K: contiguous).  A column block of a wider gradient -
                        the three K-segments of the EdgeModel's first Linear are separate jobs (round 4, see
                        dosx_segment_reduce_perm) - is written in place. */
} DosxWgrad;
int dosx_wgrad_splits(int M, int N, int K);
int dosx_wgrad_tiles(int N, int K);
int64_t dosx_wgrad_scratch_floats(int N, int K, int nsplit);
int dosx_wgrad(const DosxWgrad* g, dosx_stream_t stream);
/* The same for `n_jobs` independent jobs in as few launches as possible (one grid over up to 8 jobs at a time; jobs
 * with non-affine operands get their own launch).  Results are identical to n_jobs dosx_wgrad calls. */
int dosx_wgrad_grouped(const DosxWgrad* jobs, int n_jobs, dosx_stream_t stream);

/* Batched deterministic reduction of partial slabs: dst[i] (+)= sum_s src[s*stride + i]. */
typedef struct DosxReduceJob {
  const float* src;
  float* dst;
  int32_t nsplit;
  int32_t stride;
  int32_t count;
  int32_t accumulate; /* 0: overwrite dst, 1: add to dst */
} DosxReduceJob;
/* jobs: HOST array of n_jobs entries (copied into kernel arguments, <= 64 jobs per launch: no device
 * table, no host->device copy, safe under graph capture).  Jobs of one call must have distinct dst. */
int dosx_reduce_partials(const DosxReduceJob* jobs_host, int n_jobs, dosx_stream_t stream);
/* One flush point of a backward pass in ONE launch: the weight-gradient jobs (dosx_wgrad_grouped) and the row-partial
 * reductions of the fused LayerNorm / PReLU / bias epilogues (dosx_reduce_partials; they depend on earlier kernels only,
 * not on the weight-gradient jobs) share a grid: the reduction blocks are appended behind the weight-gradient tiles.
 * Up to 8 weight-gradient + 40 reduction jobs per launch (kernel-argument table); more jobs -> more launches. */
int dosx_grad_flush(const DosxWgrad* jobs, int n_jobs, const DosxReduceJob* rjobs_host, int n_rjobs, dosx_stream_t stream);

/* a1: edge_attr[E,4] = smooth_cutoff(|v|/r_max) * [1, sqrt3 * v/max(|v|,1e-12)]
 * (DOSTransformer_phonon.py:74-77; e3nn semantics restated in oracle/dos_oracle.py). */
int dosx_edge_feat_sh1(const float* edge_vec, float* edge_attr, int E, float r_max, dosx_stream_t stream);
/* the same + the first Linear of GN_encoder.edge_encoder on it (DOSTransformer_phonon.py:129,142; K = 4) in one launch:
 *   edge_attr[E,4] as above ;  z[E,H] = edge_attr . w0^T + b0     (w0 [H,4] row-major, b0 [H]) */
int dosx_edge_embed_sh1(const float* edge_vec, const float* w0, const float* b0, float* edge_attr, float* z, int E, int H,
                        float r_max, dosx_stream_t stream);

/* a5 + a6: CSR segment reduction over destination-sorted edges (replaces torch_scatter
 * scatter_mean / scatter_sum by `col`, DOSTransformer_phonon.py:209 / DOSTransformer.py:187)
 *   agg[n]   = scale[n] * sum_{e in [rowptr[n], rowptr[n+1])} msg[e]       (scale NULL -> 1)
 *   e_out[e] = e_in[e] + msg[e]   (edge residual, DOSTransformer_phonon.py:84; skipped if e_out NULL) */
int dosx_segment_reduce(const float* msg, const int32_t* rowptr, const float* scale, float* agg,
                        const float* e_in, float* e_out, int N, int E, int H, dosx_stream_t stream);
/* The same segment sums over a PERMUTED row order: agg[n] = sum_{j in [rowptr[n], rowptr[n+1])} msg[perm[j]]  (no scale, no
 * residual).  With rowptr_src / perm_src: the sum of a per-edge tensor over the edges that LEAVE node n.  Round 4 uses the pair
 * (this, dosx_segment_reduce) to factor the weight gradient of `Linear(cat[x[row], x[col], e])` (DOSTransformer_phonon.py:
 * 190-197): sum_e dz_e (x) x[row(e)] = sum_n (sum_{e: row(e) = n} dz_e) (x) x_n - the node blocks of that gradient become
 * N-row jobs instead of E-row ones (20 x fewer rows at 20 edges per node). */
int dosx_segment_reduce_perm(const float* msg, const int32_t* rowptr, const int32_t* perm, float* agg, int N, int E, int H,
                             dosx_stream_t stream);

/* The LAST message-passing layer's second Linear on aggregated rows (round 4).  Its per-edge output feeds only the aggregation
 * (`edge_attr += out` of the last layer is dead, DOSTransformer_phonon.py:84,209 / DOSTransformer.py:59,187), and the Linear
 * commutes with the segment sum:  scale_n * sum_{e -> n} (act_e W^T + b) = S[n] W^T + R[n]  with
 *   S[n] = scale[n] * sum_{e in [rowptr[n], rowptr[n+1])} PReLU(xhat[e] * gamma + beta)     [N, W]   (W = 2 * hidden)
 *   R[n] = c_n * bias, c_n = segment length (scale == NULL: scatter_sum) or [segment not empty] (scatter_mean)   [N, Hout]
 * so an N-row dosx_gemm (res = R) replaces the E-row one; backward: dosx_gemm on N rows, dosx_ln_prelu_bwd_gather for the
 * LayerNorm -> PReLU backward of the edge rows, an N-row weight-gradient job on (dagg, S), and the bias gradient as the column
 * sum of dosx_seg_count_scale's rows  out[n] = c_n * in[n]. */
int dosx_act_segment_sum(const float* xhat, const int32_t* rowptr, const float* scale, const float* gamma, const float* beta,
                         const float* alpha, const float* bias, float* S, float* R, int N, int W, int Hout, dosx_stream_t stream);
int dosx_seg_count_scale(const float* in, int ld_in, const int32_t* rowptr, int mean, float* out, int N, int H,
                         dosx_stream_t stream);

/* backward of the aggregation + residual:  dmsg[e] = (de_new ? de_new[e] : 0) + scale[dst[e]] * dagg[dst[e]]
 * dagg has row stride ld_dagg (it is a column block of the node-MLP input gradient), de_new row stride ld_de_new
 * (it is the e-block of the next layer's [E,3H] concat gradient, or a plain [E,H] tensor). */
int dosx_edge_grad_combine(const float* de_new, int ld_de_new, const float* dagg, int ld_dagg, const int32_t* dst,
                           const float* scale, float* dmsg, int E, int H, dosx_stream_t stream);

/* backward of the two gathers x[row], x[col] (the scatter-adds of SURVEY.md §2.2) fused with
 * the node / edge residual gradients, atomic-free:
 *   dx[n]      = dx_res[n] + dnode[n, 0:H]
 *              + sum_{e in dst-seg(n)} dcat[e, H:2H] + sum_{j in src-seg(n)} dcat[perm_src[j], 0:H]
 *   de_out[e]  = (de_new ? de_new[e] : 0) + dcat[e, 2H:3H]          (skipped if de_out NULL) */
int dosx_gather_bwd(const float* dcat, const float* dnode, int ld_dnode, const float* dx_res,
                    const int32_t* rowptr_dst, const int32_t* rowptr_src, const int32_t* perm_src,
                    const float* de_new, float* dx, float* de_out, int N, int E, int H,
                    dosx_stream_t stream);

/* a7: node->graph sum pooling over contiguous node ranges (scatter_sum(x, batch),
 * DOSTransformer_phonon.py:180) and its backward (broadcast add). */
int dosx_graph_pool(const float* x, const int32_t* graph_ptr, float* out, int ld_out, int B, int H,
                    dosx_stream_t stream);
/* num_graphs > 0: nodes with node_graph[n] >= num_graphs (ghost / padding nodes) receive a zero pooled gradient and no
 * row of dpool is read for them (otherwise dpool needs a zero row at index num_graphs). */
int dosx_graph_pool_bwd(const float* dpool, int ld_dpool, const int32_t* node_graph, float* dx, int N, int H,
                        int accumulate, int num_graphs, dosx_stream_t stream);

/* a8 (+ the LN statistics of layer_norms[0] on keys): to_dense_batch + row normalisation.
 *   kvhat[dense_row[n]] = (x[n]-mean)*rstd ; rstd_nodes[n] ; all other rows of kvhat = 0
 * (to_dense_batch DOSTransformer_phonon.py:86-87; padded rows stay exact zeros so that
 *  LayerNorm gives beta on them).  kvhat must be zero-filled by the caller
 *  (dosx_fill) or pass zero_rows = total dense rows to let the kernel do it. */
int dosx_dense_normalize(const float* x, const int32_t* dense_row, float* kvhat, float* rstd_nodes,
                         int N, int H, int dense_rows, dosx_stream_t stream);
/* The same in ONE launch, driven by the dense slots instead of the nodes (no separate zero fill): slot (pos, b) holds
 * node graph_ptr[b] + pos if pos < atoms(b), else zeros; kvhat has n_max*B + 1 rows (the last one = the zero row ghost /
 * padding nodes point at).  rstd_nodes is written for the nodes of the B graphs only. */
int dosx_dense_normalize_slots(const float* x, const int32_t* graph_ptr, float* kvhat, float* rstd_nodes, int B,
                               int n_max, int H, dosx_stream_t stream);
/* backward: dx[n] (+)= rstd[n] * (g - mean(g) - xhat*mean(g*xhat)),  g = dkvhat[dense_row[n]];
 * nodes whose dense_row equals `ghost_row` (>= 0) get dx (+)= 0 without reading rstd_nodes (ghost / padding nodes). */
int dosx_dense_normalize_bwd(const float* dkvhat, const float* kvhat, const float* rstd_nodes,
                             const int32_t* dense_row, float* dx, int N, int H, int accumulate, int ghost_row,
                             dosx_stream_t stream);
/* ... with the backward of the decoder's sum pooling (scatter_sum(x, batch), DOSTransformer_phonon.py:178-181) in the same
 * launch: dx[n] (+)= [the above] + dpool[node_graph[n]]  (dpool [B, >= H] with row stride ld_dpool; nodes whose graph id is
 * >= num_graphs - ghost nodes - get no pooled term).  What dosx_dense_normalize_bwd followed by dosx_graph_pool_bwd do. */
int dosx_dense_normalize_pool_bwd(const float* dkvhat, const float* kvhat, const float* rstd_nodes,
                                  const int32_t* dense_row, const float* dpool, int ld_dpool, const int32_t* node_graph,
                                  int num_graphs, float* dx, int N, int H, int accumulate, int ghost_row,
                                  dosx_stream_t stream);

/* to_dense_batch alone (DOSTransformer_phonon.py:86-87; torch_geometric.utils.to_dense_batch with the mask dropped): dense
 * [n_max*B, H], slot (pos, b) at row pos*B + b = node graph_ptr[b] + pos, or zeros.  The keys of the UNFUSED attention
 * path (hidden > 256), which applies layer_norms[0] to them itself; and its backward: dx[n] (+)= ddense[dense_row[n]],
 * nothing for nodes whose dense_row is ghost_row (padding nodes). */
int dosx_dense_slots(const float* x, const int32_t* graph_ptr, float* dense, int B, int n_max, int H, dosx_stream_t stream);
int dosx_dense_slots_bwd(const float* ddense, const int32_t* dense_row, float* dx, int N, int H, int accumulate,
                         int ghost_row, dosx_stream_t stream);

/* xhat[e] = (v - mean(v)) * rstd(v),  v = z[e] + p[src[e]] + q[dst[e]]   (rows of W <= 1024 floats; rstd [E] saved):
 * the LayerNorm input of the EdgeModel (DOSTransformer_phonon.py:193-195) when its first Linear is FACTORED -
 * Linear(cat[x[row], x[col], e]) = (x Wa^T)[row] + (x Wb^T)[col] + e Wc^T + b: p and q are N-row products (row strides
 * ldp / ldq), z the E-row product on a third of the columns.  Used for large edge sets (functional._factor_edge). */
int dosx_gather_add_rownorm(const float* z, const float* p, int ldp, const float* q, int ldq, const int32_t* src,
                            const int32_t* dst, float* xhat, float* rstd, int E, int W, dosx_stream_t stream);

/* Row LayerNorm without affine (key/value side of self attention) and with affine. */
int dosx_rownorm(const float* x, float* xhat, float* rstd, int M, int H, dosx_stream_t stream);
int dosx_rownorm_bwd(const float* dxhat, const float* xhat, const float* rstd, float* dx, int M, int H,
                     int accumulate, dosx_stream_t stream);
/* out[r] = (res ? res[r] : 0) + a[r] o (mask ? mask[r] : 1)  (+ the LayerNorm statistics [M,2] = (mean, rstd) of the out rows
 * when `stats` is given): the "dropout -> add residual" steps of the encoder layer (layers/transformer.py:137-138,145-148)
 * for relu / res dropout > 0, with the Bernoulli draw as an explicit multiplier (dosx_dropout_mask); mask is [M,H] dense. */
int dosx_mask_residual(const float* a, int lda, const float* mask, const float* res, int ldr, float* out, int ldo,
                       float* stats, int M, int H, dosx_stream_t stream);
/* The same followed by the backward of the (Leaky)ReLU that produced the normalised rows' input y
 * (DOSTransformer_phonon.py:103,106 `F.leaky_relu(self.fc(...))` feeding the self-attention keys):
 *     out = (dx_in + rownorm_bwd(dxhat, xhat, rstd)) * (y > 0 ? 1 : slope)         one launch instead of two */
int dosx_rownorm_bwd_act(const float* dxhat, const float* xhat, const float* rstd, const float* dx_in, const float* y,
                         float slope, float* out, int M, int H, dosx_stream_t stream);
/* y = LN(x)*gamma+beta (layers/transformer.py:76-77); saves xhat, rstd. */
int dosx_layernorm(const float* x, const float* gamma, const float* beta, float* y, float* xhat, float* rstd,
                   int M, int H, dosx_stream_t stream);
/* Backward of `LayerNorm -> PReLU` on rows of W <= 1024 floats (the Edge / Node MLPs, DOSTransformer_phonon.py:193,204, when
 * 2 * hidden > 512 - narrower rows run this in the EPI_PRELU_LN_BWD epilogue of dosx_gemm): dy [M,W] is the gradient behind
 * the PReLU, xhat / rstd what the forward saved; dz [M,W] = gradient in front of the LayerNorm; partials: ceil(M/32) rows
 * of [dgamma(W) | dbeta(W) | pad(3) | dalpha]  (row stride 2W + 4). */
int dosx_ln_prelu_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, const float* beta,
                      const float* alpha, float* dz, float* partials, int M, int W, dosx_stream_t stream);
/* The same with the gradient rows gathered and scaled: row r reads dy[idx[r]] * (scale ? scale[idx[r]] : 1) - dy holds one row per
 * destination node (see dosx_act_segment_sum). */
int dosx_ln_prelu_bwd_gather(const float* dy, const int32_t* idx, const float* scale, const float* xhat, const float* rstd,
                             const float* gamma, const float* beta, const float* alpha, float* dz, float* partials, int M, int W,
                             dosx_stream_t stream);
/* dx = LNbwd(dy); partials: ceil(M/32) rows of [dgamma(H) | dbeta(H)]   (H <= 1024) */
int dosx_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dx,
                       float* partials, int M, int H, dosx_stream_t stream);

/* a9 + the attention half of a10: one pre-norm attention block
 *   x1 = x + softmax_fp32( LN0(x) K^T * H^-1/2 ) K ,  K = kvhat*gamma0 + beta0   (K == V)
 * (layers/transformer.py:131-138, layers/multihead_attention.py:68-74: no projections, no
 *  mask, no heads).  Query row (s, bq) is at x + (s*q_stride_s + bq*q_stride_b)*H; output
 *  row (s,bq) at s*Bq + bq; key row (j, bk) at j*Bk + bk with bk = bq % Bk.
 *  H <= 256, H % 4 == 0; ANY number of keys: up to 320 per crystal the MFMA kernels (score row of a query in LDS), more
 *  through one-wave-per-row kernels with the same results contract (csrc/attention_general.hip; they need `dscores`). */
/* BWD_*_HALF: the two-kernel form of dosx_attention_bwd (dq half: needs dout,P -> dx,dS ; dkv half: needs dS -> dkvhat) issued
 * as TWO calls - one with DOSX_ATTN_BWD_DQ_HALF, one with DOSX_ATTN_BWD_DKV_HALF, e.g. the dkv half on a second stream (only the
 * final key-gradient consumers need it).  A caller that sets one flag owes the other call: together they are the whole backward. */
enum { DOSX_ATTN_RAW_Q = 1, DOSX_ATTN_NO_RESIDUAL = 2, DOSX_ATTN_BWD_DKV_HALF = 4, DOSX_ATTN_BWD_DQ_HALF = 8 };
typedef struct DosxAttn {
  int32_t Sq, Bq, Nk, Bk, H;
  int32_t q_stride_s, q_stride_b;
  int32_t flags;       /* DOSX_ATTN_RAW_Q: queries are used as given (no LN0, gamma0/beta0 still applied to
                          kvhat); DOSX_ATTN_NO_RESIDUAL: out = softmax(..)K without "+ x".  Both set =
                          the bare MultiheadAttention.forward (layers/multihead_attention.py:49-76). */
  const float* x;      /* queries / residual */
  const float* kvhat;  /* [Nk*Bk, H] normalised keys (no affine) */
  const float* gamma0;
  const float* beta0;
  float* out;          /* [Sq*Bq, H] */
  float* probs;        /* [Bq, Sq, Nk] saved softmax */
  float* qstats;       /* [Sq*Bq, 2] mean, rstd of LN0 on the query rows */
  float* out_stats;    /* optional [Sq*Bq, 2] mean, rstd of the OUTPUT rows (for the following LN1) */
  /* backward only */
  const float* dout;   /* [Sq*Bq, H] */
  float* dx;           /* [Sq*Bq, H] = dout + LN0bwd(dq) */
  float* dscores;      /* [Bq, Sq, Nk] scratch: dS */
  float* dkvhat;       /* [Nk*Bk, H] (+)= */
  int32_t dkv_accumulate;
  float* partials_q;   /* [Bq * ceil(Sq/32)] rows of [dgamma(H) | dbeta(H)] */
  float* partials_kv;  /* [Bk * ceil(Nk/32)] rows of [dgamma(H) | dbeta(H)]; [Bk * ceil(Nk/16)] rows on the dkv_part path */
  const float* drop_mask; /* optional [Bq, Sq, Nk]: attention dropout (F.dropout on the softmax output,
                          layers/multihead_attention.py:70) as an explicit multiplier M in {0, 1/(1-p)} (dosx_dropout_mask);
                          `probs` stays the un-dropped softmax.  NULL = no dropout (eval mode / p = 0) */
  float* dkv_part;     /* optional scratch [Bq * ceil(Sq/32), Nk, H]: with it (and Nk <= 64) the dq kernel also leaves each
                          query tile's share of dK + dV there and the dkv half is a small reduction over those partials
                          (no dscores round trip: dscores may then be NULL); without it the streamed dkv kernel runs */
  int32_t* dkv_cnt;    /* optional, with dkv_part: [Bk] arrival counters, zero before the launch and zero again after it.  The
                          backward is then ONE launch: the last query-tile workgroup of a key crystal to arrive (ticket)
                          sums the partial key gradients of that crystal and writes dkvhat / partials_kv itself, in the
                          reduction kernel's order (bitwise the two-launch result).  Excludes the DKV_HALF / DQ_HALF flags */
  /* forward only, optional (round 5; needs out_stats, <= 320 keys): the layer's NEXT LayerNorm applied to the output rows while
   * they are in registers - ln1_out[r] = (out[r] - mean) * rstd * ln1_gamma + ln1_beta (layers/transformer.py:141: the input of
   * fc1) - so that the fc1 GEMM of a layer whose feed-forward half is not fused reads a plain A operand instead of normalising
   * it in its prologue for every one of its column tiles (hidden 256, M = 25728: 163 -> 139 us) */
  const float* ln1_gamma;
  const float* ln1_beta;
  float* ln1_out;      /* [Sq*Bq, H] */
  /* round 6 (forward only): per-crystal key counts.  key_ptr [Bk + 1] (NULL: every crystal attends over all Nk
   * key rows - padded rows included, like the reference's unmasked to_dense_batch): crystal bk attends over its FIRST
   * key_ptr[bk + 1] - key_ptr[bk] key rows only (the batch's graph_ptr: its own atoms).  A batch of B crystals then gives what B
   * batch-size-1 forwards give - the reference evaluates at batch_size = 1 (main_eDOS.py:55-56, utils.py:61-143), where
   * Nmax = the crystal's own atom count - in one pass.  probs beyond a crystal's count are written as zeros.  Nk <= 320. */
  const int32_t* key_ptr;
} DosxAttn;
/* 1 if dosx_attention_bwd takes the partial-dKV path for this key count / width when dkv_part is given (else it needs
 * dscores and runs the streamed dkv kernel) */
int dosx_attention_pkv_supported(int Nk, int H);
/* Which shapes dosx_attention_fwd / dosx_attention_bwd (one-launch form: dkv_part + dkv_cnt) route to the crystal-aligned
 * kernels (csrc/attention_aligned.hip: flags 0, Nk <= 64, H in {64, 128, 256}): 0 = none (attention.hip's kernels), 1 = H > 128
 * only, 2 = all of them (the default; also from the environment variable DOSX_ATTN_ALIGNED).  Sets the mode unless `mode` < 0;
 * returns the previous one. */
int dosx_attention_aligned_mode(int mode);
int dosx_attention_fwd(const DosxAttn* a, dosx_stream_t stream);
int dosx_attention_bwd(const DosxAttn* a, dosx_stream_t stream);

/* Attention with K != V (`multihead_attention.py:49-76` takes any key / value pair; embed dropout, `transformer.py:61-68`,
 * draws different masks for keys and values).  No reference call site does that, so the MFMA kernels above are built for
 * K = V; the general case is composed from them and these building blocks (csrc/attention_kv.hip: one wave per output row,
 * deterministic, not a hot path).  A / mask / dP: [Bq, Sq, Nk]; X, out rows (s, bq) at s*Bq + bq; V rows (j, bk) at j*Bk + bk:
 *   dosx_attn_pv:     out[(s,bq)] = sum_j (A o mask)[bq,s,j] V[(j, bq % Bk)]                 (P.V, and dQ = dS.K)
 *   dosx_attn_tv:     out[(j,bk)] (+)= sum_{bq = bk mod Bk} sum_s (A o mask)[bq,s,j] X[(s,bq)]   (dV = P^T dOut, dK = dS^T Q)
 *   dosx_attn_dp:     dP[bq,s,j] = X[(s,bq)] . V[(j, bq % Bk)]
 *   dosx_softmax_bwd: dS = scale * P o (dPd o mask - rowsum(dPd o mask o P))                   (rows = Bq*Sq)
 *   dosx_softmax_fwd: P = softmax_fp32(scale * S) row by row - with S from dosx_attn_dp(Q, K) the attention weights for ANY
 *                     number of keys and any H % 4 == 0 (dosx_attention_fwd: H <= 256) */
int dosx_attn_pv(const float* A, const float* mask, const float* V, float* out, int Sq, int Bq, int Nk, int Bk, int H,
                 dosx_stream_t stream);
int dosx_attn_tv(const float* A, const float* mask, const float* X, float* out, int Sq, int Bq, int Nk, int Bk, int H,
                 int accumulate, dosx_stream_t stream);
int dosx_attn_dp(const float* X, const float* V, float* dP, int Sq, int Bq, int Nk, int Bk, int H, dosx_stream_t stream);
int dosx_softmax_bwd(const float* P, const float* mask, const float* dPd, float* dS, long long rows, int Nk, float scale,
                     dosx_stream_t stream);
int dosx_softmax_fwd(const float* S, float* P, long long rows, int Nk, float scale, dosx_stream_t stream);

/* out_layer (nn.Linear(H,1), DOSTransformer_phonon.py:101,115) fused with the encoder's final
 * LayerNorm (layers/transformer.py:76-77) and the squeeze/transposed store:
 *   y[r] = (xhat[r]*gamma+beta) . w + b ;  dos[(r % Bq) , r / Bq] = y[r]   (dos is [Bq, S]) */
int dosx_ln_rowdot(const float* x, const float* gamma, const float* beta, const float* w, const float* b,
                   float* xhat, float* rstd, float* dos, int S, int Bq, int H, dosx_stream_t stream);
/* ddos [Bq,S] -> dx [S*Bq,H]; partials: ceil(M/32) rows of [dgamma(H) | dbeta(H) | dw(H) | db] */
int dosx_ln_rowdot_bwd(const float* ddos, const float* xhat, const float* rstd, const float* gamma,
                       const float* beta, const float* w, float* dx, float* partials, int S, int Bq, int H,
                       dosx_stream_t stream);
/* plain y[r] = act(x[r]) . w + b for the GNN-only variants' last layer (graphnetwork_phonon.py:26,70) */
int dosx_rowdot(const float* x, const float* w, const float* b, float* dos, int S, int Bq, int H,
                dosx_stream_t stream);
int dosx_rowdot_bwd(const float* ddos, const float* x, const float* w, float* dx, float* partials, int S,
                    int Bq, int H, dosx_stream_t stream);

/* a14: losses of the callers, forward + gradient in one pass.
 * phonon (main_phDOS.py:109-114): loss = sqrt(mean_all (pg-y)^2) + beta*sqrt(mean_all (ps-y)^2)
 *   two-phase so that data-parallel ranks can all-reduce the two SSE scalars in between:
 *   dosx_sse2 writes sse[0..1]; dosx_loss_phonon_bwd consumes (possibly all-reduced) sse and
 *   `count` = global number of elements.
 * eDOS (main_eDOS.py:111-123): loss = mean_b rmse_b(global) + beta * mean_b rmse_b(system),
 *   target clamped at 0; `B_global` = number of crystals in the un-sharded batch. */
int dosx_sse2(const float* pg, const float* ps, const float* y, float* sse, int count, dosx_stream_t stream);
int dosx_loss_phonon_bwd(const float* pg, const float* ps, const float* y, const float* sse, float beta,
                         double count_global, float* dpg, float* dps, float* loss, int count,
                         dosx_stream_t stream);
/* single-process form: dosx_sse2 + dosx_loss_phonon_bwd in one launch (count = all B*S elements; sse may be NULL) */
int dosx_loss_phonon(const float* pg, const float* ps, const float* y, float* sse, float beta, float* dpg, float* dps,
                     float* loss, int count, dosx_stream_t stream);
int dosx_loss_edos(const float* pg, const float* ps, const float* y_ft, float beta, int B, int S, int B_global,
                   float* dpg, float* dps, float* loss_partial, dosx_stream_t stream);

/* dst[0] = sum_i src[i]: the mean over crystals of the eDOS loss (main_eDOS.py:117,121) from the
 * per-crystal shares written by dosx_loss_edos. */
int dosx_sum(const float* src, int n, float* dst, dosx_stream_t stream);

/* torch.optim.AdamW step on a flat buffer (main_eDOS.py:93,127): decoupled weight decay.
 * `step` is the 1-based step count. */
int dosx_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
               float eps, float weight_decay, int step, float grad_scale, dosx_stream_t stream);

/* Fused feed-forward half of the encoder layer (layers/transformer.py:141-148), forward:
 *     h   = relu( LN1(x) . W1^T + b1 )      [M,4H]   (written out: the backward needs it)
 *     out = x + h . W2^T + b2               [M,H]
 * LN1(x) = (x - mean)*rstd*gamma + beta with the per-row (mean, rstd) pairs in `stats` (produced by the attention
 * kernel's epilogue).  One launch instead of two GEMMs; dosx_ffn_supported(H) says whether this H has the fused kernel
 * (H % 32 == 0, H <= 128), otherwise the caller issues the two dosx_gemm calls. */
typedef struct DosxFfn {
  int32_t M, H;
  const float* x; int32_t ldx;
  const float* stats;
  const float* gamma; const float* beta;
  const float* w1; const float* b1;       /* fc1.weight [4H,H], fc1.bias [4H] */
  const float* w2; const float* b2;       /* fc2.weight [H,4H], fc2.bias [H]  */
  float* h; int32_t ldh;
  float* out; int32_t ldo;
  /* optional: the encoder's final LayerNorm (layers/transformer.py:76-77) applied to the result of the LAST layer in
   * the same launch: with fin_gamma != NULL, `out` receives LN(x + ffn)*gamma + beta, fin_xhat [M,H] (row stride H) the
   * normalised rows and fin_rstd [M] their 1/sigma (what dosx_layernorm would have produced from the un-normalised sum) */
  const float* fin_gamma; const float* fin_beta;
  float* fin_xhat; float* fin_rstd;
  /* optional, with the final LayerNorm: the H -> 1 output layer of the model head on the normalised rows
   * (DOSTransformer_phonon.py:116-117), dos[r % Bq][r / Bq] = LN(..)[r] . fin_w + fin_b[0] for the [S, Bq] row space
   * (fin_dos [Bq, S]; what dosx_ln_rowdot computes as a launch of its own); `out` may then be NULL */
  const float* fin_w; const float* fin_b; float* fin_dos;
  int32_t fin_S, fin_Bq;
  /* optional (round 4): the ATTENTION half of the layer in front of the feed-forward half, for key sets of at most 16 rows
   * per crystal (the prompt-guided cross attention over the atoms of a crystal, layers/transformer.py:131-138 +
   * multihead_attention.py:68-74) - one launch per encoder layer instead of dosx_attention_fwd + dosx_ffn_fwd.  With
   * att_kvhat != NULL:  `x` is the layer INPUT (the query rows: row r = s*att_Bq + bq at x + (s*att_qs + bq*att_qb)*ldx),
   * `stats` is ignored, and per row   q = LN0(x)*g0+b0,  P = softmax_fp32(q . (khat_j*g0+b0) / sqrt(H)) over the att_Nk keys
   * khat_j = att_kvhat[j*att_Bk + bq % att_Bk] (pre-normalised keys, dosx_dense_normalize_slots),  x1 = x + (P o mask) . K
   * is what the feed-forward half then reads.  Outputs, in the formats dosx_attention_fwd writes (its backward reads them):
   * att_probs [Bq,Sq,Nk] (un-dropped P), att_qstats [M,2] (mean, rstd of the query rows), att_x1 [M,H] (row stride
   * att_ldx1), att_st1 [M,2] (LayerNorm-1 statistics of x1).  att_mask: NULL or the dropout multipliers [Bq,Sq,Nk]. */
  const float* att_kvhat; const float* att_gamma0; const float* att_beta0; const float* att_mask;
  float* att_probs; float* att_qstats; float* att_x1; float* att_st1;
  int32_t att_Nk, att_Bk, att_Bq, att_Sq, att_qs, att_qb, att_ldx1;
  /* att_aligned != 0: CRYSTAL-ALIGNED tiles - a workgroup owns consecutive query rows s of ONE query batch entry (grid =
   * att_Bq x ceil(att_Sq / rows per workgroup)), the crystal's key rows are staged in LDS once per workgroup: up to 64 keys
   * (dosx_ffn_att_aligned_supported), e.g. the 51-key self attention over the energy bins; same outputs.  The caller chooses
   * it while that grid is about one round of workgroups: Sq is padded to the tile height per crystal. */
  int32_t att_aligned;
  const int32_t* att_key_ptr;   /* as DosxAttn.key_ptr: per-crystal key counts of the fused attention half (forward only) */
} DosxFfn;
int dosx_ffn_supported(int H);
int dosx_ffn_att_supported(int H, int Nk);   /* whether dosx_ffn_fwd takes the att_* fields for this shape (H, Nk <= 16) */
int dosx_ffn_att_aligned_supported(int H, int Nk);   /* ... with att_aligned (Nk <= 64 and the key rows fit the stage buffers) */
int dosx_ffn_fwd(const DosxFfn* a, dosx_stream_t stream);
/* n <= 2 consecutive layers of ONE encoder stack (layers/transformer.py:70-79) in one launch (round 6): with the attention half inside the
 * launch a layer is row-local per tile - every layer attends over the ORIGINAL keys (transformer.py:72-73) - so a workgroup runs the
 * layers back to back on its own rows.  descs[l + 1].x must be descs[l].out (dense query rows), every descriptor carries att_*, all
 * share one launch plan (same M, H, tile form); same outputs as n dosx_ffn_fwd calls, bitwise. */
int dosx_ffn_fwd_multi(const DosxFfn* descs, int n, dosx_stream_t stream);

/* Backward of the same half layer in one launch (what autograd derives from layers/transformer.py:141-148):
 *     dh = (dy . W2) o [h > 0]            [M,4H]  (written out: the fc1 weight gradient reads it)
 *     dx = dy + LN1_bwd(dh . W1)          [M,H]   (dy also flows through the residual connection)
 *     partials[r] = [ dgamma | dbeta ] of LN1 summed over the rows of workgroup r (dosx_ffn_bwd_partial_rows(M) rows,
 *                   reduced by dosx_reduce_partials like every other slab)
 * Same H support as the forward (dosx_ffn_supported).  The weight / bias gradients of fc1 and fc2 stay dosx_wgrad calls. */
typedef struct DosxFfnBwd {
  int32_t M, H;
  const float* dy; int32_t lddy;
  const float* h; int32_t ldh;            /* relu(fc1(LN1 x)) saved by the forward */
  const float* x; int32_t ldx;            /* input of the half layer (pre-LN1) */
  const float* stats;                     /* (mean, rstd) per row of x */
  const float* gamma;                     /* LN1 weight */
  const float* w1; const float* w2;       /* fc1.weight [4H,H], fc2.weight [H,4H] */
  float* dh; int32_t lddh;
  float* dx; int32_t lddx;
  float* partials; int32_t partial_ld;
  /* optional: the backward of the encoder's final LayerNorm (layers/transformer.py:76-77) in front of the LAST layer's
   * half, same launch.  With fin_gamma != NULL, `dy` is the gradient w.r.t. that LayerNorm's OUTPUT; the kernel computes
   * dy' = LN_bwd(dy) from fin_xhat [M,H] (row stride H) / fin_rstd [M] (what the forward's fin_* outputs hold), writes
   * it to fin_dy [M,H] (row stride lddy: the fc2 weight gradient reads it), continues with dy', and appends
   * [ dgamma | dbeta ] of that LayerNorm to every partial row (columns [2H,4H): partial_ld >= 4H). */
  const float* fin_gamma; const float* fin_xhat; const float* fin_rstd;
  float* fin_dy;
  /* ... and, in front of that LayerNorm, the H -> 1 output layer of the model head (DOSTransformer_phonon.py:116-117
   * `out_layer(LN(h))`): with fin_ddos != NULL (then `dy` may be NULL) the rows are  dy[r] = ddos[r % Bq][r / Bq] * w  for
   * the [S, Bq] row space of the encoder and ddos [Bq, S]; the partial rows get [ dw (H) | db ] behind the LayerNorm's two
   * groups (columns [4H, 5H] : partial_ld >= 5H + 1).  What dosx_ln_rowdot_bwd computes as a launch of its own. */
  const float* fin_ddos; const float* fin_w; const float* fin_beta;
  int32_t fin_S, fin_Bq;
  /* optional (round 5; att_kvhat != NULL): the ATTENTION half's backward of the same encoder layer behind the feed-forward
   * half's, same launch (layers/transformer.py:131-138 + layers/multihead_attention.py:68-74 differentiated) - what
   * dosx_attention_bwd's one-launch form computes from `dx` as its `dout`, which then never reaches HBM (`dx` may be NULL).
   * Crystal-aligned tiles: a workgroup owns consecutive query rows s of ONE query batch entry (row r = s * att_Bq + bq; grid =
   * att_Bq x ceil(att_Sq / rows per workgroup) = dosx_ffn_att_bwd_partial_rows workgroups and partial rows).
   *   att_x [rows, att_ldxin]: the LAYER input (row (s, bq) at s * att_qs + bq * att_qb); att_qstats / att_probs / att_mask: saved
   *   by the forward; att_kvhat [Nk * Bk, H]: the pre-normalised keys; att_dxin [M, att_lddxin]: gradient w.r.t. the layer input;
   *   att_partials_q [Bq * tiles, 2H], att_partials_kv [Bk * ceil(Nk / 16), 2H], att_dkv_part [Bq * tiles * Nk, H] scratch,
   *   att_dkv_cnt [Bk] arrival counters (zero before and after), att_dkvhat (+)= the key gradient (att_dkv_accumulate)
   * - the fields of DosxAttn with the same names, same layouts (32-query tiles), same summation orders. */
  const float* att_x; int32_t att_ldxin;
  const float* att_kvhat; const float* att_gamma0; const float* att_beta0;
  const float* att_probs; const float* att_qstats; const float* att_mask;
  float* att_dxin; int32_t att_lddxin;
  float* att_partials_q; float* att_partials_kv; float* att_dkv_part; int32_t* att_dkv_cnt;
  float* att_dkvhat; int32_t att_dkv_accumulate;
  int32_t att_Nk, att_Bk, att_Bq, att_Sq, att_qs, att_qb;
} DosxFfnBwd;
int dosx_ffn_bwd_partial_rows(int M);
int dosx_ffn_att_bwd_supported(int H, int Nk, int Sq, int Bq);   /* whether dosx_ffn_bwd takes the att_* fields for this shape */
int dosx_ffn_att_bwd_partial_rows(int Sq, int Bq);               /* workgroups = partial rows of such a launch */
int dosx_ffn_att_aligned_rows(int Sq, int Bq);                   /* rows per workgroup (16 / 32) of a crystal-aligned launch, fwd and bwd */
int dosx_ffn_bwd(const DosxFfnBwd* a, dosx_stream_t stream);

/* The phonon EDGE ENCODER in one launch (round 6; csrc/heads.hip): attr = SH(l <= 1)(vec) * smooth_cutoff(|vec| / r_max)
 * (DOSTransformer_phonon.py:74-77; [E,4]), z = attr . w0^T + b0 ([E,H]: the saved pre-activation), out = prelu(z) . w2^T + b2
 * (GN_encoder.edge_encoder, :129,142).  What dosx_edge_embed_sh1 + dosx_gemm (PReLU prologue) compute; hidden 64 / 128. */
typedef struct DosxEdgeEnc {
  int32_t E, H;
  const float* vec;                         /* [E,3] */
  float inv_rmax;
  const float* w0; const float* b0;         /* [H,4], [H] */
  const float* alpha;
  const float* w2; const float* b2;         /* [H,H], [H] */
  float* attr; float* z;                    /* [E,4], [E,H] OUT */
  float* out; int32_t ldo;                  /* [E,H] OUT */
} DosxEdgeEnc;
int dosx_edge_enc_supported(int H);
int dosx_edge_enc_fwd(const DosxEdgeEnc* a, dosx_stream_t stream);

/* The backward of the two output heads (DOSTransformer_phonon.py:93-109, DOSTransformer.py:67-83: `fc`, `fc_prompt`, F.leaky_relu)
 * between the self encoder's and the first encoder's backward, in ONE launch (round 6; csrc/heads.hip):
 *     dpre[(s, bq)] = ( ddosin + rownorm_bwd(dkvs, kvs, rstd) )[(s, bq)] * leaky_relu'(dosin[(s, bq)])       rows (s, bq) at s * 2B + bq
 *     de1[(s, b)]   = dpre[(s, b)] . wg[:, :H] + dpre[(s, B + b)] . ws[:, :H]                                rows (s, b) at s * B + b
 * wg = fc.weight [H, ldwg], ws = fc_prompt.weight [H, ldws] (nn.Linear layout: only their first H columns - the E1 inputs - are
 * read).  What dosx_rownorm_bwd_act followed by two dosx_gemm calls (w_layout 1, row-mapped A) compute; hidden 64 / 128 / 256. */
typedef struct DosxHeadsBwd {
  int32_t S, B, H;
  const float* dkvs; const float* kvs; const float* rstd;     /* [S*2B, H], [S*2B, H], [S*2B] */
  const float* ddosin; const float* dosin;                    /* [S*2B, H] */
  float slope;                                                /* leaky_relu negative slope (0.01) */
  float* dpre;                                                /* [S*2B, H] OUT */
  const float* wg; int32_t ldwg; const float* ws; int32_t ldws;
  float* de1; int32_t ldde1;                                  /* [S*B, H] OUT */
} DosxHeadsBwd;
int dosx_heads_bwd_supported(int H);
int dosx_heads_bwd(const DosxHeadsBwd* a, dosx_stream_t stream);

/* Split-bf16 GEMM (round 6, csrc/gemm_bf16x3.hip) - a MEASUREMENT next to dosx_gemm, not part of any training / inference program:
 *     C[M,N] = epi( A[M,K] . op(W) + bias[N] )      w_layout 0: W is [N][K] (nn.Linear), 1: W is [K][N]
 *     epi: relu (act = 1), then + res[M,N] (optional), then zero where mask[M,N] <= 0 (optional: dosx_gemm's EPI_RELU_MASK)
 * fp32 operands and result; every element is split into three bf16 terms on the fly and the six leading products run on the
 * bf16 matrix pipe with fp32 accumulation (error of the size of one fp32 rounding per product; tests hold it to dosx_gemm's
 * tolerance against float64).  N % 128 == 0, K % 32 == 0, 16-byte aligned operands, leading dimensions multiples of 4.
 * bench.py reports it as `secondary.split_bf16` for the two largest Electron-DOS GEMM shapes (no reference counterpart: the
 * reference computes in fp32 / fp64, main_phDOS.py:15-16). */
int dosx_gemm_bf16x3_supported(int M, int N, int K);
int dosx_gemm_bf16x3(const float* A, int lda, const float* W, int ldw, int w_layout, const float* bias, float* C, int ldc,
                     int M, int N, int K, int act, const float* res, int ldres, const float* mask, int ldmask,
                     dosx_stream_t stream);

/* Linear -> LayerNorm -> PReLU -> Linear (+ residual) in ONE launch for small row counts: the NodeModel MLP of a GNN layer
 * (DOSTransformer_phonon.py:200-212 `node_mlp_2(cat[x, agg])`, DOSTransformer.py:178-190).
 *     z    = [a0 | a1] . W1^T + b1                  [M,NH]     (a0: k0 columns, a1: K - k0 columns, plain row-major)
 *     xhat = (z - mean) * rstd   (eps 1e-5)         written out with rstd: the backward and dosx_wgrad read them
 *     out  = prelu(xhat*gamma + beta) . W2^T + b2 (+ res)      [M,NO]
 * What two dosx_gemm calls (EPI_LN, then PRO_LN_PRELU) compute, for the shapes dosx_mlp_ln_supported accepts
 * (K, NH multiples of 128 up to 512; NO a multiple of 64 up to 256, NO <= K); 16 rows per workgroup, so meant for
 * M of a few thousand rows at most - the caller keeps the two-GEMM path for long inputs. */
typedef struct DosxMlpLn {
  int32_t M, K, NH, NO, k0;
  const float* a0; int32_t lda0;
  const float* a1; int32_t lda1;          /* may be NULL when k0 == K */
  const float* w1; const float* b1;       /* [NH,K], [NH]  (nn.Linear layout) */
  const float* gamma; const float* beta;  /* LayerNorm(NH) */
  const float* alpha;                     /* PReLU (1 parameter) */
  const float* w2; const float* b2;       /* [NO,NH], [NO] */
  const float* res; int32_t ldres;        /* optional residual added to the output (NULL: none) */
  float* xhat; float* rstd;               /* [M,NH] (row stride NH), [M] */
  float* out; int32_t ldo;
  /* optional third product on the finished output rows (round 5: the NEXT message-passing layer's node products of the
   * factored EdgeModel Linear, DosxGemm.add_p / add_q - the dosx_gemm_pair launch between two layers):
   *     pq[r, b * n3 + n] = sum_k out[r, k] * w3[n * ldw3 + b * NO + k]      b < nb3, n < n3 (a multiple of 256)
   * NULL w3: none */
  const float* w3; int32_t ldw3, n3, nb3;
  float* pq; int32_t ldpq;
  /* round 6 - COLUMN-SPLIT form (cs_buf != NULL; shapes of dosx_mlp_ln_cs_supported: the NodeModel (K, NH, NO) = (2, 2, 1) x
   * hidden, hidden 64 / 128): a 16-row tile is shared by hidden / 16 workgroups of 4 waves, each owning 32 columns of the first
   * product and 16 of the second (1 / 8 of the weights and of the MFMAs at hidden 128) - 232 workgroups instead of 29 for the 450
   * node rows of the benchmark batch; the siblings exchange the pre-LayerNorm tile (and, with w3, the output tile) IN the launch:
   * publish with write-through stores, one ticket each on the tile's counter, every sibling waits for all tickets and reads the
   * tile back (csrc/mlp2.hip).  cs_buf: dosx_mlp_ln_cs_scratch_floats(M, NH) floats of scratch; cs_cnt: dosx_mlp_ln_cs_tiles(M)
   * arrival counters, zero before and after the launch.  Same outputs as the one-workgroup-per-tile form to fp32 rounding
   * (k-split sums added in wave order: bitwise reproducible run to run). */
  float* cs_buf; int32_t* cs_cnt;
} DosxMlpLn;
int dosx_mlp_ln_supported(int K, int NH, int NO);
int dosx_mlp_ln_cs_supported(int K, int NH, int NO);      /* whether the column-split form exists for this block shape */
int dosx_mlp_ln_cs_tiles(int M);                          /* counters it needs (= 16-row tiles) */
int64_t dosx_mlp_ln_cs_scratch_floats(int M, int NH);     /* floats of cs_buf */
int dosx_mlp_ln_fwd(const DosxMlpLn* a, dosx_stream_t stream);

/* The node ENCODER + the first message-passing layer's node products in ONE column-split launch (round 6; csrc/mlp2.hip):
 *     z   = x . W0^T + b0            [M,H]  written (the saved pre-activation: the backward reads it)        x [M,Fa], W0 [H,Fa]
 *     out = prelu(z) . W2^T + b2     [M,H]  (DOSTransformer_phonon.py:129,141: Linear -> PReLU -> Linear)
 *     pq[r, b * n3 + n] = sum_k out[r, k] * w3[n * ldw3 + b * H + k]      b < nb3, nb3 * n3 == 4H (DosxMlpLn.w3)
 * What two dosx_gemm calls and dosx_gemm_pair compute; hidden 64 / 128, even Fa <= 256; cs_cnt: dosx_mlp_ln_cs_tiles(M) counters. */
typedef struct DosxEncCs {
  int32_t M, Fa, H;
  const float* x; int32_t ldx;
  const float* w0; int32_t ldw0; const float* b0;
  const float* alpha;
  const float* w2; const float* b2;
  float* z; float* out; int32_t ldo;
  const float* w3; int32_t ldw3, n3, nb3;
  float* pq; int32_t ldpq;
  int32_t* cs_cnt;
} DosxEncCs;
int dosx_enc_cs_supported(int Fa, int H);
int dosx_enc_cs_fwd(const DosxEncCs* a, dosx_stream_t stream);

/* Backward of the same block in one launch:
 *     da = dy . W2 ; dy' = da o prelu'(xhat*gamma+beta) ; dz = LayerNorm_bwd(dy' o gamma)  [M,NH] (written: the W1 weight
 *     gradient reads it) ; dcat = dz . W1  [M,K]
 *     partials[r] = [ dgamma (NH) | dbeta (NH) | pad | dalpha ] summed over the rows of workgroup r — the layout of
 *     DOSX_EPI_PRELU_LN_BWD (dalpha in column partial_ld - 1), dosx_mlp_ln_bwd_partial_rows(M) rows.
 * The weight / bias gradients of the two Linear layers stay dosx_wgrad calls. */
typedef struct DosxMlpLnBwd {
  int32_t M, K, NH, NO;
  const float* dy; int32_t lddy;
  const float* xhat; const float* rstd;   /* saved by the forward */
  const float* w1; const float* w2;
  const float* gamma; const float* beta; const float* alpha;
  float* dz;                              /* [M,NH], row stride NH */
  float* dcat; int32_t lddcat;            /* [M,K] */
  float* partials; int32_t partial_ld;
  int32_t add_dy;                         /* 1: dcat[:, :NO] += dy - the block's residual connection out = res + MLP(cat[res, .])
                                             (NodeModel, DOSTransformer_phonon.py:204-212 + :83) differentiated in the same launch */
  float* cs_buf; int32_t* cs_cnt;         /* column-split form (see DosxMlpLn): the siblings exchange the `da` tile; same sizes */
  /* round 6, column-split form only: what PRODUCES dy, in the same launch - the N-row launch that used to run in front of it.
   *   pre = 1: the node side of the LATER message-passing layer's factored input gradient (DosxNodeGrad: pre_dz [E,2H] that
   *            layer's dz, pre_rowptr_src / pre_perm_src the CSR by source, pre_aggd [M,2H], pre_w [2H, >= 2H] row stride pre_ldw,
   *            pre_res / pre_res2 optional [M,H] addends, pre_aggs [M,2H] OUT): dy = pre_res + pre_res2 + aggs Wa + aggd Wb;
   *   pre = 2: dosx_dense_normalize_pool_bwd (pre_dkv / pre_kvhat dense [*,H], pre_rstd_nodes [M], pre_dense_row [M], pre_dpool
   *            [graphs, pre_ld_dpool], pre_node_graph [M], pre_num_graphs, pre_ghost_row; no accumulation): dy = its dx.
   * dy is then an OUTPUT (pre_dy = the same pointer, writable; [M,H], row stride lddy): the weight-gradient jobs read it. */
  int32_t pre;
  float* pre_dy;
  const float* pre_dz; const int32_t* pre_rowptr_src; const int32_t* pre_perm_src; const float* pre_aggd;
  const float* pre_w; int32_t pre_ldw;
  const float* pre_res; int32_t pre_ldres; const float* pre_res2; int32_t pre_ldres2;
  float* pre_aggs;
  const float* pre_dkv; const float* pre_kvhat; const float* pre_rstd_nodes; const int32_t* pre_dense_row;
  const float* pre_dpool; int32_t pre_ld_dpool; const int32_t* pre_node_graph; int32_t pre_num_graphs, pre_ghost_row;
} DosxMlpLnBwd;
int dosx_mlp_ln_bwd_partial_rows(int M);
int dosx_mlp_ln_bwd(const DosxMlpLnBwd* a, dosx_stream_t stream);

/* The EdgeModel of one message-passing layer + the aggregation behind it in ONE launch (round 5; SURVEY.md §7 step 4;
 * DOSTransformer_phonon.py:186-197,209 and the edge residual :84), first Linear FACTORED (see DosxGemm.add_p):
 *     z    = e . Wc^T + b1 + pq[src[r], 0:2H] + pq[dst[r], 2H:4H]          Wc = first Linear's weight block [2H, H], row stride ldw1
 *     xhat = (z - mean) * rstd  (eps 1e-5; written with rstd: the backward and dosx_wgrad read them)
 *     msg  = prelu(xhat * gamma + beta) . W3^T + b3                        W3 [H, 2H]
 *     e_out = e + msg (skipped when NULL) ;  seg_agg[n] = seg_scale[n] * sum_{r in seg(n)} msg[r]
 * What dosx_gemm (EPI_LN + add_p / add_q) followed by dosx_gemm (PRO_LN_PRELU, EPI_SEGSUM) compute, for hidden 64 / 128: one
 * workgroup per node-aligned row tile of the batch's tile table (seg_* exactly as in DosxGemm), the [48, 2H] intermediate in LDS.
 * pq [nodes, >= 4H] holds the two node products P | Q (dosx_gemm_pair on the N node rows). */
typedef struct DosxEdgeMlp {
  int32_t E, H;
  const float* e; int32_t lde;
  const float* pq; int32_t ldpq;
  const int32_t* src; const int32_t* dst;
  const float* w1; int32_t ldw1; const float* b1;
  const float* gamma; const float* beta; const float* alpha;
  const float* w3; const float* b3;
  float* xhat; float* rstd;
  float* e_out; int32_t ldeo;
  const int32_t* seg_tile; int32_t seg_ntiles;
  const int32_t* seg_rowptr; const float* seg_scale; float* seg_agg; float* seg_part; int32_t* seg_cnt;
} DosxEdgeMlp;
/* Backward of the same block in one launch (what dosx_edge_grad_combine + dosx_gemm EPI_PRELU_LN_BWD_SEG + the E-row input-
 * gradient GEMM compute):
 *     dmsg[r] = de_next[r] + seg_scale[dst[r]] * dagg[dst[r]]     (written: W3's weight gradient reads it; de_next NULL: the last layer)
 *     dz = LayerNorm_bwd(PReLU_bwd(dmsg . W3))  [E,2H] written ;  seg_agg[n] = sum_{r in seg(n)} dz[r]  [nodes, 2H]  (unscaled)
 *     de[r] = dz[r] . Wc + de_next[r]                              (gradient of the layer's edge state)
 *     partials[tile] = [ dgamma (2H) | dbeta (2H) | pad | dalpha ]  - seg_ntiles rows, the layout of DOSX_EPI_PRELU_LN_BWD. */
typedef struct DosxEdgeMlpBwd {
  int32_t E, H;
  const float* dagg; int32_t lddagg;
  const float* de_next; int32_t ldden;
  const int32_t* dst;
  const float* xhat; const float* rstd;
  const float* w3;
  const float* w1; int32_t ldw1;
  const float* gamma; const float* beta; const float* alpha;
  float* dmsg;
  float* dz;
  float* de; int32_t ldde;
  float* partials; int32_t partial_ld;
  const int32_t* seg_tile; int32_t seg_ntiles;
  const int32_t* seg_rowptr; const float* seg_scale; float* seg_agg; float* seg_part; int32_t* seg_cnt;
} DosxEdgeMlpBwd;
int dosx_edge_mlp_bwd(const DosxEdgeMlpBwd* a, dosx_stream_t stream);
/* The NODE side of the factored EdgeModel input gradient in one launch (what dosx_segment_reduce_perm + a two-segment dosx_gemm
 * with w_seg_off compute; hidden 64 / 128 / 256):
 *     aggs[n] = sum_{e: src(e) = n} dz[e]   (rows perm_src[rowptr_src[n] .. rowptr_src[n+1]) of dz [E,2H]; written: the weight
 *                                            gradient of the source block reads it)
 *     dx[n]   = res[n] + res2[n] + aggs[n] . w[:, :H] + aggd[n] . w[:, H:2H]        w: the first Linear's weight [2H, 3H], row stride ldw
 * (DOSTransformer_phonon.py:166,190-197 differentiated w.r.t. x[row] / x[col]; res / res2 optional [N,H] addends). */
typedef struct DosxNodeGrad {
  int32_t N, H;
  const float* dz;
  const int32_t* rowptr_src; const int32_t* perm_src;
  const float* aggd;
  const float* w; int32_t ldw;
  const float* res; int32_t ldres;
  const float* res2; int32_t ldres2;
  float* aggs;
  float* dx; int32_t lddx;
} DosxNodeGrad;
int dosx_node_grad(const DosxNodeGrad* a, dosx_stream_t stream);
int dosx_edge_mlp_supported(int H);
int dosx_edge_mlp_fwd(const DosxEdgeMlp* a, dosx_stream_t stream);

/* Graph metadata ("CSR build") on the device, stream-ordered, no host round trip — counterpart of what PyG's
 * collate / to_dense_batch / torch_scatter derive per call from `edge_index` and `batch`
 * (DOSTransformer_phonon.py:48-56,86,209).
 *   edge_index [2,E] int64 (any order), batch [N] int64 non-decreasing, B graphs.
 *   dst/src [E]      : the edges STABLY sorted by destination (aggregation = contiguous segment sum);
 *   edge_perm [E]    : int64, position of sorted edge e in the caller's edge arrays (may be NULL);
 *   rowptr_dst/src [N+1], perm_src [E] : CSR by destination / by source (ids in the sorted numbering);
 *   graph_ptr [B+1], node_graph [N], dense_row [N] = pos*B + graph, inv_deg [N] = 1/max(in-degree,1);
 *   n_max [1]        : device scalar, atoms of the largest crystal (may be NULL).
 * workspace: dosx_csr_workspace_bytes(E) bytes of device scratch owned by the caller. */
int dosx_csr_workspace_bytes(int E, size_t* bytes);
int dosx_csr_build(const long long* edge_index, const long long* batch, int N, int E, int B, int32_t* src, int32_t* dst,
                   long long* edge_perm, int32_t* rowptr_dst, int32_t* perm_src, int32_t* rowptr_src,
                   int32_t* graph_ptr, int32_t* node_graph, int32_t* dense_row, float* inv_deg, int32_t* n_max,
                   void* workspace, size_t ws_bytes, dosx_stream_t stream);

/* Collate on the device: the dataset is resident with per-crystal destination-sorted edges and cached
 * local CSR (`*_all` arrays, crystal c owns nodes [node_ptr_all[c], node_ptr_all[c+1]) and edges likewise; its cached
 * row pointers have n_c + 1 entries starting at node_ptr_all[c] + c).  `sel [B]` picks the crystals of the batch,
 * `out_node_ptr / out_edge_ptr [B+1]` are the batch's prefix sums (N, E = their last entries).  Outputs: everything
 * dosx_csr_build would give for the concatenated batch (already destination-sorted), plus `batch` / `edge_index` in
 * the reference's int64 schema and `node_row [N]`, `edge_row [E]` = source rows for gathering the feature tensors. */
int dosx_collate(const int32_t* sel, const int32_t* node_ptr_all, const int32_t* edge_ptr_all, const int32_t* out_node_ptr,
                 const int32_t* out_edge_ptr, int B, int N, int E, const int32_t* src_all, const int32_t* dst_all,
                 const int32_t* perm_src_all, const int32_t* rowptr_dst_all, const int32_t* rowptr_src_all,
                 const float* inv_deg_all, long long* batch, long long* edge_index, int32_t* src, int32_t* dst,
                 int32_t* perm_src, int32_t* rowptr_dst, int32_t* rowptr_src, int32_t* node_graph, int32_t* dense_row,
                 float* inv_deg, int32_t* node_row, int32_t* edge_row, dosx_stream_t stream);

/* Collate straight into the STATIC buffers of a shape bucket, ghost padding included (train.Trainer.step_dataset): the
 * selected crystals' segments + feature rows as dosx_collate / the row gathers would give them, followed by the ghost tail
 * batch.pad_batch appends (ghost nodes: zero features, node_graph = B, dense slot n_max*B; ghost edges: self loops on the
 * first ghost node, zero features; rowptr[N] = E, rowptr[N+1..N_pad] = E_pad).  All pointers device memory; feature data
 * fp32.  target: [B,S] rows (phdos / y_ft), glob: [B,n_glob] (n_glob = 0: none), system: int32 [B].
 * node_row / edge_row: scratch [N_pad] / [E_pad].  Needs N_pad > N (at least one ghost node). */
typedef struct DosxCollate {
  int32_t B, N, E, N_pad, E_pad, n_max, Fa, Fe, S, n_glob;
  const int32_t* sel;
  const int32_t* out_node_ptr;
  const int32_t* out_edge_ptr;
  const int32_t* node_ptr_all;
  const int32_t* edge_ptr_all;
  const int32_t* src_all;
  const int32_t* dst_all;
  const int32_t* perm_src_all;
  const int32_t* rowptr_dst_all;
  const int32_t* rowptr_src_all;
  const float* inv_deg_all;
  const float* x_all;
  const float* edge_feat_all;
  const float* target_all;
  const float* glob_all;
  const int32_t* system_all;
  float* x;
  float* edge_feat;
  float* target;
  float* glob;
  int32_t* system;
  int32_t* src;
  int32_t* dst;
  int32_t* perm_src;
  int32_t* rowptr_dst;
  int32_t* rowptr_src;
  int32_t* graph_ptr;
  int32_t* node_graph;
  int32_t* dense_row;
  float* inv_deg;
  int32_t* node_row;
  int32_t* edge_row;
  /* optional: node-aligned row tiles of the message GEMM (DosxGemm EPI_SEGSUM), seg_tile [3][T+1] (NULL = skip).  Crystal c
   * owns the tiles [tile_off_all[c], tile_off_all[c+1]) of tile_e_all / tile_n_all / tile_p_all (crystal-local START edge /
   * node and the chunk info of each tile); out_tile_ptr [B+1] = prefix sums of the selected crystals' tile counts; ghost
   * edges follow in tile_rows-row tiles, the first of which owns every ghost node; the remaining slots are empty. */
  int32_t T, tile_rows;
  const int32_t* out_tile_ptr;
  const int32_t* tile_off_all;
  const int32_t* tile_e_all;
  const int32_t* tile_n_all;
  int32_t* seg_tile;
  const int32_t* tile_p_all;
} DosxCollate;
int dosx_collate_padded(const DosxCollate* d, dosx_stream_t stream);

/* Periodic neighbour list of C crystals at once (replaces ASE's `neighbor_list("ijS", a, cutoff=r_max,
 * self_interaction=True)` + the edge_vec arithmetic of `utils.py:267-273` in build_data).  `pos [N][3]` Cartesian,
 * `cell [C][3][3]` lattice vectors as rows, crystal c owns atoms [atom_ptr[c], atom_ptr[c+1]) and the ordered atom
 * pairs [pair_ptr[c], pair_ptr[c+1]) (pair_ptr = prefix sums of n_c^2, n_pairs = its last entry).  An edge is a triple
 * (i, j, S) with |pos[j]-pos[i]+S·cell| < cutoff; (i, i, 0) only if `self_interaction`; bit k of `pbc_mask` = axis k is
 * periodic.  Two passes: `_count` writes the number of edges of every pair, the caller turns it into exclusive offsets
 * `pair_off` (E = their total), `_fill` writes per edge: crystal, src = i and dst = j (crystal-local), shift [E][3],
 * edge_vec [E][3] = pos[dst]-pos[src]+shift·cell.  Order: crystal, i, j, shift (lexicographic) — deterministic. */
int dosx_neighbor_count(const double* pos, const double* cell, const int32_t* atom_ptr, const long long* pair_ptr, int C,
                        long long n_pairs, double cutoff, int self_interaction, int pbc_mask, int32_t* pair_count,
                        dosx_stream_t stream);
int dosx_neighbor_fill(const double* pos, const double* cell, const int32_t* atom_ptr, const long long* pair_ptr, int C,
                       long long n_pairs, double cutoff, int self_interaction, int pbc_mask, const long long* pair_off,
                       int32_t* crystal, int32_t* src, int32_t* dst, int32_t* shift, double* edge_vec,
                       dosx_stream_t stream);

/* Replay of a recorded launch list (the host side of train.Trainer(replay=True), see csrc/replay.cpp): `n` calls are
 * issued in order.  A call names its callee by `op` (from dosx_replay_op: every `int dosx_*` entry point of this header,
 * plus "hipEventRecord" / "hipStreamWaitEvent" for stream fork/join) and carries the callee's integer-class arguments in
 * declaration order (pointers, integers, by-pointer descriptors; at most 19) and its float/double arguments in
 * declaration order (at most 6).  Each op is dispatched through a typed thunk generated from this header, i.e. an
 * ordinary C call.  Returns the first non-zero return code (its index in *failed_index) or 0. */
enum { DOSX_OP_HIP_BASE = 1000 };
typedef struct DosxCall {
  int32_t op;
  int32_t nint;
  int32_t nflt;
  int32_t reserved;
  int64_t iarg[19];
  double farg[6];
} DosxCall;
/* op index of an entry point by name (-1 if unknown); optionally its integer-class / floating-point argument counts */
int dosx_replay_op(const char* name, int* n_int, int* n_float);
int dosx_replay(const DosxCall* calls, int n, int* failed_index);
/* Measurement twin: replays the list with a HIP event pair around every library entry, recorded on the stream that entry
 * launches on; ms_out[n] (HOST array) receives the elapsed milliseconds of every entry.  Ends with a device
 * synchronisation - diagnostic use only (bench.py's per-kernel roofline figures). */
int dosx_replay_timed(const DosxCall* calls, int n, float* ms_out, int* failed_index);

/* misc */
/* Dropout multiplier: mask[i] = u_i >= p ? 1/(1-p) : 0 with u_i uniform in [0,1) from Philox4x32-10 (counter = (i/4, stream_id),
 * key = *seed_dev, word i%4 of the block; the top 24 bits make u).  The seed is read FROM DEVICE MEMORY so a recorded /
 * graph-captured program draws fresh masks on every replay: the caller bumps *seed_dev between steps.  stream_id
 * separates the masks of different layers / sites drawn under one seed. */
int dosx_dropout_mask(float* mask, int64_t n, float p, const unsigned long long* seed_dev, long long stream_id,
                      dosx_stream_t stream);
/* Batched device-to-device copy in ONE launch: dst[0:dwords] = src[0:dwords] (32-bit words, 4-byte aligned, any mix of
 * fp32 / int32 buffers).  jobs: HOST array, copied into kernel arguments (no device table, graph-capturable).
 * train._Slot.load moves a batch's fields and CSR arrays into the static buffers of its shape bucket with it. */
typedef struct DosxCopyJob {
  const void* src;
  void* dst;
  int64_t dwords;
} DosxCopyJob;
int dosx_copy_many(const DosxCopyJob* jobs_host, int n_jobs, dosx_stream_t stream);
int dosx_fill(float* p, float value, int64_t n, dosx_stream_t stream);
/* out[r] = table[idx[r]] rows (prompt_token[g.system], DOSTransformer_phonon.py:105) and its backward
 * (deterministic: one workgroup per table row scans idx). */
int dosx_embed_rows(const float* table, const int32_t* idx, float* out, int rows, int width, dosx_stream_t stream);
int dosx_embed_rows_bwd(const float* dout, int ld_dout, const int32_t* idx, float* dtable, int rows,
                        int table_rows, int width, dosx_stream_t stream);
/* dst[i, 0:width] (+)= sum_{j < n_red} src[(i*stride_out + j*stride_red), 0:width]
 * Row reductions of the [S,B,H] layout: over the batch (gradient of the energy-embedding broadcast,
 * DOSTransformer_phonon.py:143: n_out=S, n_red=B, stride_out=B, stride_red=1) or over the energy bins
 * (gradient of graph.expand(51,...), :91: n_out=B, n_red=S, stride_out=1, stride_red=B). */
int dosx_reduce_rows(const float* src, int ld_src, float* dst, int ld_dst, int n_out, int n_red, int stride_out,
                     int stride_red, int width, int accumulate, dosx_stream_t stream);
/* out = dy * (y > 0 ? 1 : slope): backward of F.leaky_relu (DOSTransformer_phonon.py:95) from its output. */
int dosx_act_bwd(const float* dy, const float* y, float slope, float* out, int64_t n, dosx_stream_t stream);

const char* dosx_last_error(void);
int dosx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DOSX_H */
