"""ctypes binding of libspr_hip.so (the C ABI declared in include/spr.h).

The library is the product: there is no CPU or PyTorch fallback behind these
calls.  If the shared object is missing or a symbol is absent, importing /
calling fails loudly.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPR_HIP_LIB: experiment builds only (scripts/abl: ablation variants of single kernels)
LIB_PATH = os.environ.get("SPR_HIP_LIB") or os.path.join(_HERE, "libspr_hip.so")

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_sz = ctypes.c_size_t
_l = ctypes.c_long

# name -> (restype, argtypes); mirrors include/spr.h exactly
SIGNATURES = {
    "spr_version": (_i, []),
    "spr_last_error": (ctypes.c_char_p, []),
    "spr_grid_subsample_workspace_bytes": (_sz, [_i, _i]),
    "spr_grid_subsample": (_i, [_vp, _vp, _i, _i, _f, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "spr_voxel_downsample_workspace_bytes": (_sz, [_i]),
    "spr_voxel_downsample": (_i, [_vp, _i, ctypes.c_double, _vp, _vp, _vp, _sz, _vp]),
    "spr_radius_neighbors_workspace_bytes": (_sz, [_i, _i, _i]),
    "spr_radius_neighbors": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _f, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "spr_radius_table_bytes": (_sz, [_i, _i]),
    "spr_radius_table_build_workspace_bytes": (_sz, [_i, _i]),
    "spr_radius_table_query_workspace_bytes": (_sz, [_i]),
    "spr_radius_table_slots": (_i, []),
    "spr_radius_table_build": (_i, [_vp, _vp, _i, _i, _f, _vp, _sz, _vp, _sz, _vp]),
    "spr_radius_table_query": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "spr_radius_table_query_a": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "spr_kpconv_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "spr_kpconv_fwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _f, _vp,
                            _i, _vp, _sz, _vp]),
    "spr_instnorm_workspace_bytes": (_sz, [_i, _i, _i]),
    "spr_instnorm": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _f, _vp, _vp, _sz, _vp]),
    "spr_maxpool_gather": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp]),
    "spr_linear_workspace_bytes": (_sz, []),
    "spr_linear": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "spr_linear_r": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    "spr_range_parts": (_i, []),
    "spr_absmax_multi": (_i, [_vp, _i, _i, _vp, _vp]),
    "spr_absmax": (_i, [_vp, _l, _i, _l, _vp, _vp]),
    "spr_kpconv_fwd_r": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _f, _vp,
                              _i, _vp, _i, _vp, _i, _vp, _sz, _vp]),
    "spr_kpconv_plan_bytes": (_sz, [_i]),
    "spr_kpconv_plan": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "spr_kpconv_wplanes_bytes": (_sz, [_i, _i]),
    "spr_kpconv_prep_weights": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _sz, _vp]),
    "spr_kpconv_fwd_p": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _f, _vp,
                              _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "spr_instnorm_r": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _f, _vp, _vp, _i, _vp, _sz, _vp]),
    "spr_block_tail_tile_rows": (_i, [_i, _i, _i]),
    "spr_block_tail_tiles_len": (_sz, [_i, _i, _i]),
    "spr_block_tail_tiles": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "spr_block_tail_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "spr_block_tail": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp, _vp, _i, _vp, _i,
                            _vp, _i, _vp, _i, _vp, _i, _vp, _sz, _vp]),
    "spr_block_tail_n": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp, _vp, _i, _vp, _i,
                              _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _f, _vp, _sz, _vp]),
    "spr_instnorm_stats": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    "spr_maxpool_gather_r": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "spr_maxpool_gather_o": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "spr_cell_order_workspace_bytes": (_sz, [_i]),
    "spr_cell_order": (_i, [_vp, _vp, _i, _i, _f, _vp, _vp, _sz, _vp]),
    "spr_layernorm_range_count": (_i, [_i]),
    "spr_layernorm_r": (_i, [_vp, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "spr_attn_inproj_varlen_fwd_r": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _i, _vp, _i, _vp,
                                          _i, _vp, _vp, _vp, _sz, _vp]),
    "spr_attn_inproj_prepare": (_i, [_vp, _i, _vp, _vp]),
    "spr_set_gemm_mode": (_i, [_i]),
    "spr_layernorm": (_i, [_vp, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "spr_posemb_sine": (_i, [_vp, _i, _i, _f, _f, _vp, _vp]),
    "spr_attn_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "spr_attn_varlen_fwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _i,
                                 _vp, _sz, _vp]),
    "spr_attn_varlen_fwd_lse": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _vp,
                                     _vp, _sz, _vp]),
    "spr_attn_varlen_bwd_lse": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i,
                                     _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "spr_attn_bwd_workspace_bytes": (_sz, [_i, _i]),
    "spr_attn_bwd_workspace_bytes2": (_sz, [_i, _i, _i]),
    "spr_attn_varlen_bwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f,
                                 _vp, _vp, _vp, _vp, _sz, _vp]),
    "spr_attn_inproj_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "spr_attn_inproj_varlen_fwd": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _i, _vp,
                                        _sz, _vp]),
    "spr_set_attn_mode": (_i, [_i]),
    "spr_xenc_prepared_bytes": (_sz, [_i, _i]),
    "spr_xenc_plan_bytes": (_sz, []),
    "spr_xenc_prepare": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _f, _f, _vp, _sz, _vp, _sz, _vp]),
    "spr_xenc_workspace_bytes": (_sz, [_i, _i]),
    "spr_xenc_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "spr_match_workspace_bytes": (_sz, [_vp, _i]),
    "spr_match_dualsoftmax": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "spr_match_dualsoftmax2": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "spr_pose_residuals": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "spr_pose_scores": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp]),
    "spr_weighted_procrustes": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "spr_sinkhorn_workspace_bytes": (_sz, [_vp, _i]),
    "spr_match_sinkhorn": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "spr_sinkhorn_correspondences": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp,
                                          _sz, _vp]),
    "spr_loss_workspace_bytes": (_sz, [_i, _i, _i]),
    "spr_overlap_pool": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "spr_bce_logits_mean": (_i, [_vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "spr_infonce_pair": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _sz, _vp]),
    "spr_transform_l1_pair": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "spr_sum_scaled": (_i, [_vp, _i, _f, _vp, _vp]),
    "spr_gather_rows": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp]),
    "spr_bgemm": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _l, _l, _l, _l, _l, _l, _f, _f, _vp]),
    "spr_tn_product_split_workspace_bytes": (_sz, []),
    "spr_tn_product_split": (_i, [_vp, _vp, _l, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    "spr_tn_product_f64_workspace_bytes": (_sz, [_l, _i, _i]),
    "spr_tn_product_f64": (_i, [_vp, _vp, _l, _i, _i, _vp, _vp, _sz, _vp]),
    "spr_reduce_parts": (_i, [_vp, _i, _l, _f, _vp, _i, _vp]),
    "spr_act_bwd": (_i, [_vp, _vp, _i, _l, _vp, _vp]),
    "spr_colsum_workspace_bytes": (_sz, [_i]),
    "spr_colsum": (_i, [_vp, _l, _i, _vp, _vp, _sz, _vp]),
    "spr_layernorm_bwd_workspace_bytes": (_sz, [_i]),
    "spr_layernorm_bwd": (_i, [_vp, _i, _i, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "spr_instnorm_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "spr_instnorm_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    "spr_scatter_workspace_bytes": (_sz, [_l, _i]),
    "spr_maxpool_bwd": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "spr_scatter_rows_add": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "spr_kpconv_weighted_features": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _f, _vp, _vp, _vp]),
    "spr_kpconv_bwd_dx": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    "spr_kpconv_bwd_dx_r": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _i, _f, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "spr_softmax_rows": (_i, [_vp, _vp, _i, _i, _vp]),
    "spr_softmax_bwd_rows": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "spr_bce_logits_mean_bwd": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "spr_infonce_pair_dlogits": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _sz,
                                      _vp]),
    "spr_wsym_bwd": (_i, [_vp, _i, _vp, _vp]),
    "spr_transform_l1_pair_bwd": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "spr_weighted_procrustes_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "spr_sinkhorn_bwd_workspace_bytes": (_sz, [_vp, _i, _i]),
    "spr_sinkhorn_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "spr_selftest": (_i, [_vp]),
    "spr_prof_enable": (_i, [_i]),
    "spr_prof_read": (_i, [_i, _vp, _vp, _vp]),
}

_lib = None


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc"), "clean"])
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(_HERE, "csrc")])
    return LIB_PATH


def lib():
    """The loaded library with typed entry points.  Raises if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no fallback path)")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().spr_last_error()
        raise RuntimeError(f"{what}: {msg.decode() if msg else 'error'} (status {rc})")
