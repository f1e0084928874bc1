"""ctypes binding of ``csrc/libdosx.so`` (C ABI declared in ``include/dosx.h``).

There is NO fallback: if the HIP library is missing or fails to load, every op raises
:class:`DosxUnavailable` — the product path never routes through a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DOSX_LIB") or os.path.join(_HERE, "csrc", "libdosx.so")   # DOSX_LIB: diagnostic builds only

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int32)
BIG = 1 << 30


class DosxUnavailable(RuntimeError):
    pass


class DosxError(RuntimeError):
    pass


class RowMap(C.Structure):
    _fields_ = [("d", C.c_int32), ("m", C.c_int32), ("c", C.c_int32), ("off", C.c_int32), ("idx", C.c_void_p)]


class Seg(C.Structure):
    _fields_ = [("p", C.c_void_p), ("ld", C.c_int32), ("width", C.c_int32), ("map", RowMap)]


class Gemm(C.Structure):
    _fields_ = [
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("nseg", C.c_int32),
        ("a", Seg * 3),
        ("pro", C.c_int32),
        ("pro_gamma", C.c_void_p), ("pro_beta", C.c_void_p), ("pro_alpha", C.c_void_p), ("pro_stats", C.c_void_p),
        ("w", C.c_void_p), ("ldw", C.c_int32), ("w_layout", C.c_int32),
        ("epi", C.c_int32), ("act", C.c_int32), ("act_slope", C.c_float),
        ("bias", C.c_void_p),
        ("out", C.c_void_p), ("ldo", C.c_int32), ("out_map", RowMap),
        ("res", C.c_void_p), ("ldr", C.c_int32), ("res_map", RowMap),
        ("stats_out", C.c_void_p), ("aux_out", C.c_void_p),
        ("aux", C.c_void_p), ("ldaux", C.c_int32),
        ("aux_stats", C.c_void_p),
        ("epi_gamma", C.c_void_p), ("epi_beta", C.c_void_p), ("epi_alpha", C.c_void_p),
        ("partials", C.c_void_p), ("partial_ld", C.c_int32),
        ("seg_tile", C.c_void_p), ("seg_ntiles", C.c_int32), ("seg_rowptr", C.c_void_p), ("seg_scale", C.c_void_p),
        ("seg_agg", C.c_void_p), ("seg_part", C.c_void_p), ("seg_cnt", C.c_void_p), ("res_col0", C.c_int32),
        ("norm_out", C.c_void_p), ("norm_rstd", C.c_void_p), ("res_pre", C.c_int32),
        ("add_p", C.c_void_p), ("add_ip", C.c_void_p), ("add_q", C.c_void_p), ("add_iq", C.c_void_p), ("ld_add", C.c_int32),
        ("w_seg_off", C.c_int32),
    ]


class Wgrad(C.Structure):
    _fields_ = [
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("dy", Seg),
        ("nseg", C.c_int32),
        ("a", Seg * 3),
        ("pro", C.c_int32),
        ("pro_gamma", C.c_void_p), ("pro_beta", C.c_void_p), ("pro_alpha", C.c_void_p), ("pro_stats", C.c_void_p),
        ("slab", C.c_void_p), ("slab_bias", C.c_void_p),
        ("nsplit", C.c_int32), ("accumulate", C.c_int32),
        ("dst", C.c_void_p), ("dst_bias", C.c_void_p), ("counters", C.c_void_p), ("ldd", C.c_int32),
    ]


class ReduceJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("nsplit", C.c_int32), ("stride", C.c_int32),
                ("count", C.c_int32), ("accumulate", C.c_int32)]


class Attn(C.Structure):
    _fields_ = [
        ("Sq", C.c_int32), ("Bq", C.c_int32), ("Nk", C.c_int32), ("Bk", C.c_int32), ("H", C.c_int32),
        ("q_stride_s", C.c_int32), ("q_stride_b", C.c_int32), ("flags", C.c_int32),
        ("x", C.c_void_p), ("kvhat", C.c_void_p), ("gamma0", C.c_void_p), ("beta0", C.c_void_p),
        ("out", C.c_void_p), ("probs", C.c_void_p), ("qstats", C.c_void_p), ("out_stats", C.c_void_p),
        ("dout", C.c_void_p), ("dx", C.c_void_p), ("dscores", C.c_void_p), ("dkvhat", C.c_void_p),
        ("dkv_accumulate", C.c_int32),
        ("partials_q", C.c_void_p), ("partials_kv", C.c_void_p), ("drop_mask", C.c_void_p), ("dkv_part", C.c_void_p), ("dkv_cnt", C.c_void_p),
        ("ln1_gamma", C.c_void_p), ("ln1_beta", C.c_void_p), ("ln1_out", C.c_void_p),
        ("key_ptr", C.c_void_p),
    ]


class FfnBwd(C.Structure):
    _fields_ = [
        ("M", C.c_int32), ("H", C.c_int32),
        ("dy", C.c_void_p), ("lddy", C.c_int32),
        ("h", C.c_void_p), ("ldh", C.c_int32),
        ("x", C.c_void_p), ("ldx", C.c_int32),
        ("stats", C.c_void_p),
        ("gamma", C.c_void_p),
        ("w1", C.c_void_p), ("w2", C.c_void_p),
        ("dh", C.c_void_p), ("lddh", C.c_int32),
        ("dx", C.c_void_p), ("lddx", C.c_int32),
        ("partials", C.c_void_p), ("partial_ld", C.c_int32),
        ("fin_gamma", C.c_void_p), ("fin_xhat", C.c_void_p), ("fin_rstd", C.c_void_p),
        ("fin_dy", C.c_void_p),
        ("fin_ddos", C.c_void_p), ("fin_w", C.c_void_p), ("fin_beta", C.c_void_p),
        ("fin_S", C.c_int32), ("fin_Bq", C.c_int32),
        ("att_x", C.c_void_p), ("att_ldxin", C.c_int32),
        ("att_kvhat", C.c_void_p), ("att_gamma0", C.c_void_p), ("att_beta0", C.c_void_p),
        ("att_probs", C.c_void_p), ("att_qstats", C.c_void_p), ("att_mask", C.c_void_p),
        ("att_dxin", C.c_void_p), ("att_lddxin", C.c_int32),
        ("att_partials_q", C.c_void_p), ("att_partials_kv", C.c_void_p), ("att_dkv_part", C.c_void_p), ("att_dkv_cnt", C.c_void_p),
        ("att_dkvhat", C.c_void_p), ("att_dkv_accumulate", C.c_int32),
        ("att_Nk", C.c_int32), ("att_Bk", C.c_int32), ("att_Bq", C.c_int32), ("att_Sq", C.c_int32), ("att_qs", C.c_int32), ("att_qb", C.c_int32),
    ]


class Ffn(C.Structure):
    _fields_ = [
        ("M", C.c_int32), ("H", C.c_int32),
        ("x", C.c_void_p), ("ldx", C.c_int32),
        ("stats", C.c_void_p),
        ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("w1", C.c_void_p), ("b1", C.c_void_p),
        ("w2", C.c_void_p), ("b2", C.c_void_p),
        ("h", C.c_void_p), ("ldh", C.c_int32),
        ("out", C.c_void_p), ("ldo", C.c_int32),
        ("fin_gamma", C.c_void_p), ("fin_beta", C.c_void_p),
        ("fin_xhat", C.c_void_p), ("fin_rstd", C.c_void_p),
        ("fin_w", C.c_void_p), ("fin_b", C.c_void_p), ("fin_dos", C.c_void_p),
        ("fin_S", C.c_int32), ("fin_Bq", C.c_int32),
        ("att_kvhat", C.c_void_p), ("att_gamma0", C.c_void_p), ("att_beta0", C.c_void_p), ("att_mask", C.c_void_p),
        ("att_probs", C.c_void_p), ("att_qstats", C.c_void_p), ("att_x1", C.c_void_p), ("att_st1", C.c_void_p),
        ("att_Nk", C.c_int32), ("att_Bk", C.c_int32), ("att_Bq", C.c_int32), ("att_Sq", C.c_int32),
        ("att_qs", C.c_int32), ("att_qb", C.c_int32), ("att_ldx1", C.c_int32), ("att_aligned", C.c_int32),
        ("att_key_ptr", C.c_void_p),
    ]


class MlpLn(C.Structure):
    _fields_ = [
        ("M", C.c_int32), ("K", C.c_int32), ("NH", C.c_int32), ("NO", C.c_int32), ("k0", C.c_int32),
        ("a0", C.c_void_p), ("lda0", C.c_int32),
        ("a1", C.c_void_p), ("lda1", C.c_int32),
        ("w1", C.c_void_p), ("b1", C.c_void_p),
        ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("alpha", C.c_void_p),
        ("w2", C.c_void_p), ("b2", C.c_void_p),
        ("res", C.c_void_p), ("ldres", C.c_int32),
        ("xhat", C.c_void_p), ("rstd", C.c_void_p),
        ("out", C.c_void_p), ("ldo", C.c_int32),
        ("w3", C.c_void_p), ("ldw3", C.c_int32), ("n3", C.c_int32), ("nb3", C.c_int32),
        ("pq", C.c_void_p), ("ldpq", C.c_int32),
        ("cs_buf", C.c_void_p), ("cs_cnt", C.c_void_p),
    ]


class MlpLnBwd(C.Structure):
    _fields_ = [
        ("M", C.c_int32), ("K", C.c_int32), ("NH", C.c_int32), ("NO", C.c_int32),
        ("dy", C.c_void_p), ("lddy", C.c_int32),
        ("xhat", C.c_void_p), ("rstd", C.c_void_p),
        ("w1", C.c_void_p), ("w2", C.c_void_p),
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("alpha", C.c_void_p),
        ("dz", C.c_void_p),
        ("dcat", C.c_void_p), ("lddcat", C.c_int32),
        ("partials", C.c_void_p), ("partial_ld", C.c_int32), ("add_dy", C.c_int32),
        ("cs_buf", C.c_void_p), ("cs_cnt", C.c_void_p),
        ("pre", C.c_int32), ("pre_dy", C.c_void_p),
        ("pre_dz", C.c_void_p), ("pre_rowptr_src", C.c_void_p), ("pre_perm_src", C.c_void_p), ("pre_aggd", C.c_void_p),
        ("pre_w", C.c_void_p), ("pre_ldw", C.c_int32),
        ("pre_res", C.c_void_p), ("pre_ldres", C.c_int32), ("pre_res2", C.c_void_p), ("pre_ldres2", C.c_int32),
        ("pre_aggs", C.c_void_p),
        ("pre_dkv", C.c_void_p), ("pre_kvhat", C.c_void_p), ("pre_rstd_nodes", C.c_void_p), ("pre_dense_row", C.c_void_p),
        ("pre_dpool", C.c_void_p), ("pre_ld_dpool", C.c_int32), ("pre_node_graph", C.c_void_p), ("pre_num_graphs", C.c_int32),
        ("pre_ghost_row", C.c_int32),
    ]


class EncCs(C.Structure):
    _fields_ = [
        ("M", C.c_int32), ("Fa", C.c_int32), ("H", C.c_int32),
        ("x", C.c_void_p), ("ldx", C.c_int32),
        ("w0", C.c_void_p), ("ldw0", C.c_int32), ("b0", C.c_void_p),
        ("alpha", C.c_void_p),
        ("w2", C.c_void_p), ("b2", C.c_void_p),
        ("z", C.c_void_p), ("out", C.c_void_p), ("ldo", C.c_int32),
        ("w3", C.c_void_p), ("ldw3", C.c_int32), ("n3", C.c_int32), ("nb3", C.c_int32),
        ("pq", C.c_void_p), ("ldpq", C.c_int32),
        ("cs_cnt", C.c_void_p),
    ]


class EdgeEnc(C.Structure):
    _fields_ = [
        ("E", C.c_int32), ("H", C.c_int32),
        ("vec", C.c_void_p), ("inv_rmax", C.c_float),
        ("w0", C.c_void_p), ("b0", C.c_void_p), ("alpha", C.c_void_p),
        ("w2", C.c_void_p), ("b2", C.c_void_p),
        ("attr", C.c_void_p), ("z", C.c_void_p),
        ("out", C.c_void_p), ("ldo", C.c_int32),
    ]


class HeadsBwd(C.Structure):
    _fields_ = [
        ("S", C.c_int32), ("B", C.c_int32), ("H", C.c_int32),
        ("dkvs", C.c_void_p), ("kvs", C.c_void_p), ("rstd", C.c_void_p),
        ("ddosin", C.c_void_p), ("dosin", C.c_void_p),
        ("slope", C.c_float),
        ("dpre", C.c_void_p),
        ("wg", C.c_void_p), ("ldwg", C.c_int32), ("ws", C.c_void_p), ("ldws", C.c_int32),
        ("de1", C.c_void_p), ("ldde1", C.c_int32),
    ]


class EdgeMlp(C.Structure):
    _fields_ = [
        ("E", C.c_int32), ("H", C.c_int32),
        ("e", C.c_void_p), ("lde", C.c_int32),
        ("pq", C.c_void_p), ("ldpq", C.c_int32),
        ("src", C.c_void_p), ("dst", C.c_void_p),
        ("w1", C.c_void_p), ("ldw1", C.c_int32), ("b1", C.c_void_p),
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("alpha", C.c_void_p),
        ("w3", C.c_void_p), ("b3", C.c_void_p),
        ("xhat", C.c_void_p), ("rstd", C.c_void_p),
        ("e_out", C.c_void_p), ("ldeo", C.c_int32),
        ("seg_tile", C.c_void_p), ("seg_ntiles", C.c_int32),
        ("seg_rowptr", C.c_void_p), ("seg_scale", C.c_void_p), ("seg_agg", C.c_void_p), ("seg_part", C.c_void_p), ("seg_cnt", C.c_void_p),
    ]


class EdgeMlpBwd(C.Structure):
    _fields_ = [
        ("E", C.c_int32), ("H", C.c_int32),
        ("dagg", C.c_void_p), ("lddagg", C.c_int32),
        ("de_next", C.c_void_p), ("ldden", C.c_int32),
        ("dst", C.c_void_p),
        ("xhat", C.c_void_p), ("rstd", C.c_void_p),
        ("w3", C.c_void_p),
        ("w1", C.c_void_p), ("ldw1", C.c_int32),
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("alpha", C.c_void_p),
        ("dmsg", C.c_void_p), ("dz", C.c_void_p),
        ("de", C.c_void_p), ("ldde", C.c_int32),
        ("partials", C.c_void_p), ("partial_ld", C.c_int32),
        ("seg_tile", C.c_void_p), ("seg_ntiles", C.c_int32),
        ("seg_rowptr", C.c_void_p), ("seg_scale", C.c_void_p), ("seg_agg", C.c_void_p), ("seg_part", C.c_void_p), ("seg_cnt", C.c_void_p),
    ]


class NodeGrad(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("H", C.c_int32),
        ("dz", C.c_void_p),
        ("rowptr_src", C.c_void_p), ("perm_src", C.c_void_p),
        ("aggd", C.c_void_p),
        ("w", C.c_void_p), ("ldw", C.c_int32),
        ("res", C.c_void_p), ("ldres", C.c_int32),
        ("res2", C.c_void_p), ("ldres2", C.c_int32),
        ("aggs", C.c_void_p),
        ("dx", C.c_void_p), ("lddx", C.c_int32),
    ]


class CopyJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("dwords", C.c_int64)]


class Collate(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("B", "N", "E", "N_pad", "E_pad", "n_max", "Fa", "Fe", "S", "n_glob")] + \
               [(k, C.c_void_p) for k in ("sel", "out_node_ptr", "out_edge_ptr", "node_ptr_all", "edge_ptr_all", "src_all", "dst_all",
                                          "perm_src_all", "rowptr_dst_all", "rowptr_src_all", "inv_deg_all", "x_all",
                                          "edge_feat_all", "target_all", "glob_all", "system_all", "x", "edge_feat", "target",
                                          "glob", "system", "src", "dst", "perm_src", "rowptr_dst", "rowptr_src", "graph_ptr",
                                          "node_graph", "dense_row", "inv_deg", "node_row", "edge_row")] + \
               [("T", C.c_int32), ("tile_rows", C.c_int32)] + \
               [(k, C.c_void_p) for k in ("out_tile_ptr", "tile_off_all", "tile_e_all", "tile_n_all", "seg_tile", "tile_p_all")]


class Call(C.Structure):
    _fields_ = [("op", C.c_int32), ("nint", C.c_int32), ("nflt", C.c_int32), ("reserved", C.c_int32),
                ("iarg", C.c_int64 * 19), ("farg", C.c_double * 6)]


# name -> argtypes  (restype is int unless listed in _RESTYPES)
_P, _I, _F, _L, _D = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_double
_SIGS = {
    "dosx_gemm_partial_rows": [_I, _I, _I],
    "dosx_set_sliver_max_gf": [_D],
    "dosx_gemm": [C.POINTER(Gemm), _P],
    "dosx_gemm_pair": [C.POINTER(Gemm), C.POINTER(Gemm), _P],
    "dosx_gemm_kernel_name": [C.POINTER(Gemm), C.c_char_p, _I],
    "dosx_wgrad_splits": [_I, _I, _I],
    "dosx_wgrad_tiles": [_I, _I],
    "dosx_wgrad_scratch_floats": [_I, _I, _I],
    "dosx_wgrad": [C.POINTER(Wgrad), _P],
    "dosx_grad_flush": [C.POINTER(Wgrad), _I, C.POINTER(ReduceJob), _I, _P],
    "dosx_wgrad_grouped": [C.POINTER(Wgrad), _I, _P],
    "dosx_reduce_partials": [C.POINTER(ReduceJob), _I, _P],
    "dosx_edge_feat_sh1": [_P, _P, _I, _F, _P],
    "dosx_edge_embed_sh1": [_P, _P, _P, _P, _P, _I, _I, _F, _P],
    "dosx_segment_reduce": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "dosx_segment_reduce_perm": [_P, _P, _P, _P, _I, _I, _I, _P],
    "dosx_edge_grad_combine": [_P, _I, _P, _I, _P, _P, _P, _I, _I, _P],
    "dosx_gather_bwd": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "dosx_graph_pool": [_P, _P, _P, _I, _I, _I, _P],
    "dosx_graph_pool_bwd": [_P, _I, _P, _P, _I, _I, _I, _I, _P],
    "dosx_dense_normalize": [_P, _P, _P, _P, _I, _I, _I, _P],
    "dosx_dense_normalize_slots": [_P, _P, _P, _P, _I, _I, _I, _P],
    "dosx_dense_normalize_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dosx_dense_normalize_pool_bwd": [_P, _P, _P, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P],
    "dosx_dense_slots": [_P, _P, _P, _I, _I, _I, _P],
    "dosx_dense_slots_bwd": [_P, _P, _P, _I, _I, _I, _I, _P],
    "dosx_ln_prelu_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "dosx_ln_prelu_bwd_gather": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "dosx_act_segment_sum": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "dosx_seg_count_scale": [_P, _I, _P, _I, _P, _I, _I, _P],
    "dosx_gather_add_rownorm": [_P, _P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _P],
    "dosx_rownorm": [_P, _P, _P, _I, _I, _P],
    "dosx_rownorm_bwd": [_P, _P, _P, _P, _I, _I, _I, _P],
    "dosx_mask_residual": [_P, _I, _P, _P, _I, _P, _I, _P, _I, _I, _P],
    "dosx_rownorm_bwd_act": [_P, _P, _P, _P, _P, _F, _P, _I, _I, _P],
    "dosx_layernorm": [_P, _P, _P, _P, _P, _P, _I, _I, _P],
    "dosx_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _P],
    "dosx_attention_pkv_supported": [_I, _I],
    "dosx_attention_aligned_mode": [_I],
    "dosx_attention_fwd": [C.POINTER(Attn), _P],
    "dosx_attention_bwd": [C.POINTER(Attn), _P],
    "dosx_attn_pv": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dosx_attn_tv": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "dosx_attn_dp": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dosx_softmax_bwd": [_P, _P, _P, _P, _L, _I, _F, _P],
    "dosx_softmax_fwd": [_P, _P, _L, _I, _F, _P],
    "dosx_ln_rowdot": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "dosx_ln_rowdot_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "dosx_rowdot": [_P, _P, _P, _P, _I, _I, _I, _P],
    "dosx_rowdot_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "dosx_sse2": [_P, _P, _P, _P, _I, _P],
    "dosx_loss_phonon_bwd": [_P, _P, _P, _P, _F, _D, _P, _P, _P, _I, _P],
    "dosx_loss_phonon": [_P, _P, _P, _P, _F, _P, _P, _P, _I, _P],
    "dosx_loss_edos": [_P, _P, _P, _F, _I, _I, _I, _P, _P, _P, _P],
    "dosx_sum": [_P, _I, _P, _P],
    "dosx_adamw": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P],
    "dosx_ffn_supported": [_I],
    "dosx_ffn_att_supported": [_I, _I],
    "dosx_ffn_att_aligned_supported": [_I, _I],
    "dosx_ffn_fwd": [C.POINTER(Ffn), _P],
    "dosx_ffn_fwd_multi": [C.POINTER(Ffn), _I, _P],
    "dosx_ffn_bwd_partial_rows": [_I],
    "dosx_ffn_att_bwd_supported": [_I, _I, _I, _I],
    "dosx_ffn_att_bwd_partial_rows": [_I, _I],
    "dosx_ffn_att_aligned_rows": [_I, _I],
    "dosx_ffn_bwd": [C.POINTER(FfnBwd), _P],
    "dosx_mlp_ln_supported": [_I, _I, _I],
    "dosx_mlp_ln_cs_supported": [_I, _I, _I],
    "dosx_mlp_ln_cs_tiles": [_I],
    "dosx_mlp_ln_cs_scratch_floats": [_I, _I],
    "dosx_mlp_ln_fwd": [C.POINTER(MlpLn), _P],
    "dosx_mlp_ln_bwd_partial_rows": [_I],
    "dosx_mlp_ln_bwd": [C.POINTER(MlpLnBwd), _P],
    "dosx_enc_cs_supported": [_I, _I],
    "dosx_enc_cs_fwd": [C.POINTER(EncCs), _P],
    "dosx_edge_enc_supported": [_I],
    "dosx_edge_enc_fwd": [C.POINTER(EdgeEnc), _P],
    "dosx_heads_bwd_supported": [_I],
    "dosx_heads_bwd": [C.POINTER(HeadsBwd), _P],
    "dosx_gemm_bf16x3_supported": [_I, _I, _I],
    "dosx_gemm_bf16x3": [_P, _I, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P],
    "dosx_edge_mlp_supported": [_I],
    "dosx_edge_mlp_fwd": [C.POINTER(EdgeMlp), _P],
    "dosx_edge_mlp_bwd": [C.POINTER(EdgeMlpBwd), _P],
    "dosx_node_grad": [C.POINTER(NodeGrad), _P],
    "dosx_csr_workspace_bytes": [_I, C.POINTER(C.c_size_t)],
    "dosx_csr_build": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P],
    "dosx_collate": [_P] * 5 + [_I] * 3 + [_P] * 18 + [_P],
    "dosx_collate_padded": [C.POINTER(Collate), _P],
    "dosx_neighbor_count": [_P, _P, _P, _P, _I, _L, _D, _I, _I, _P, _P],
    "dosx_neighbor_fill": [_P, _P, _P, _P, _I, _L, _D, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "dosx_replay_op": [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "dosx_replay": [C.POINTER(Call), _I, C.POINTER(C.c_int)],
    "dosx_replay_timed": [C.POINTER(Call), _I, C.POINTER(C.c_float), C.POINTER(C.c_int)],
    "dosx_dropout_mask": [_P, _L, _F, _P, _L, _P],
    "dosx_copy_many": [C.POINTER(CopyJob), _I, _P],
    "dosx_fill": [_P, _F, _L, _P],
    "dosx_embed_rows": [_P, _P, _P, _I, _I, _P],
    "dosx_embed_rows_bwd": [_P, _I, _P, _P, _I, _I, _I, _P],
    "dosx_reduce_rows": [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "dosx_act_bwd": [_P, _P, _F, _P, _L, _P],
    "dosx_last_error": [],
    "dosx_version": [],
}
_RESTYPES = {"dosx_last_error": C.c_char_p, "dosx_wgrad_scratch_floats": C.c_int64, "dosx_mlp_ln_cs_scratch_floats": C.c_int64}
EXPORTS = tuple(_SIGS)

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libdosx.so once; raise DosxUnavailable (never fall back) if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DosxUnavailable(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C dostransformer_amd/csrc` (needs hipcc, --offload-arch=gfx950). "
            f"dostransformer_amd has no CPU fallback.")
    try:
        # torch bundles its own HIP runtime (torch/lib/libamdhip64.so); it must be the one already in the
        # process when libdosx.so is resolved, otherwise two runtimes coexist and ours sees no device.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise DosxUnavailable(f"cannot load {LIB_PATH}: {e}") from e
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the sources that decide which kernels a step launches and what they do: csrc/,
    include/dosx.h and the launch-sequencing Python.  profiles/r*_pmc_traffic.json carries the hash it was measured on;
    bench.py refuses the file when it differs (the box has no .git, so a content hash stands in for the commit)."""
    import glob
    import hashlib
    root = os.path.dirname(_HERE)
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")) +
                   glob.glob(os.path.join(_HERE, "csrc", "*.cpp")))
    files += [os.path.join(root, "include", "dosx.h")] + [os.path.join(_HERE, f) for f in ("functional.py", "ops.py", "train.py")]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, root).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().dosx_last_error()
        raise DosxError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
