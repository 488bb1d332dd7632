"""ctypes binding of libls3d.so (the hipcc-built C ABI declared in include/ls3d.h).

The product path has exactly one implementation: the HIP library.  If it is missing or fails to load,
importing any op raises — there is no CPU or PyTorch fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libls3d.so")

OK = 0
ERR_UNSUPPORTED = -3
_ERR = {-1: "LS3D_ERR_ARG", -2: "LS3D_ERR_LAUNCH", -3: "LS3D_ERR_UNSUPPORTED", -4: "LS3D_ERR_WORKSPACE"}

EXPORTS = [
    "ls3d_version", "ls3d_voxelize_dynamic", "ls3d_voxelize_hard_workspace_bytes", "ls3d_voxelize_hard",
    "ls3d_dynamic_scatter_workspace_bytes", "ls3d_dynamic_scatter", "ls3d_dynamic_scatter_backward_workspace_bytes",
    "ls3d_dynamic_scatter_backward", "ls3d_segment_reduce_workspace_bytes", "ls3d_segment_reduce", "ls3d_vfe_mean", "ls3d_vfe_improved_mean",
    "ls3d_vfe_tokens", "ls3d_transvfe", "ls3d_transvfe_workspace_bytes", "ls3d_mha_core", "ls3d_group_max", "ls3d_layernorm", "ls3d_index_build",
    "ls3d_rulebook_subm", "ls3d_rulebook_conv_workspace_bytes", "ls3d_rulebook_conv", "ls3d_rulebook_masks", "ls3d_rulebook_sort_keys", "ls3d_rulebook_parity_keys", "ls3d_point_mlp", "ls3d_segment_local_index", "ls3d_segment_local_index32", "ls3d_gather_gemm", "ls3d_gather_gemm_pack", "ls3d_gather_gemm_packed_floats", "ls3d_gather_gemm_default_nt", "ls3d_spconv_wgrad_workspace_bytes", "ls3d_spconv_wgrad", "ls3d_spconv_pairs_bytes", "ls3d_spconv_pairs", "ls3d_spconv_identity_pairs", "ls3d_spconv_wgrad_on_pairs", 
    "ls3d_tile_keys", "ls3d_tile_plan_bytes", "ls3d_tile_build", "ls3d_tile_plan", "ls3d_tile_plan_workspace_bytes", "ls3d_radix_sort",
    "ls3d_radix_sort_workspace_bytes", "ls3d_tile_conv_packed_bytes", "ls3d_tile_conv_pack", "ls3d_tile_conv_packed_bytes_bf16", "ls3d_tile_conv_pack_bf16", "ls3d_tile_conv", "ls3d_tile_conv_workspace_bytes", "ls3d_tile_conv_counter_bytes", "ls3d_tile_conv_trace_bytes", "ls3d_tile_chain_state_bytes", "ls3d_tile_conv_chain", "ls3d_transvfe_planes_bytes", "ls3d_transvfe_pack_planes",
    "ls3d_voxel_centers", "ls3d_frame_offsets", "ls3d_three_nn", "ls3d_three_interpolate", "ls3d_three_interpolate_grad",
    "ls3d_seg_loss_workspace_bytes", "ls3d_seg_loss_saved_bytes", "ls3d_seg_loss_forward", "ls3d_seg_loss_backward", "ls3d_layer_norm_workspace_bytes", "ls3d_layer_norm_forward", "ls3d_layer_norm_backward", "ls3d_batch_norm_workspace_bytes", "ls3d_batch_norm_stats", "ls3d_batch_norm_finalize", "ls3d_batch_norm_apply", "ls3d_batch_norm_backward_sums", "ls3d_batch_norm_backward_apply", "ls3d_column_sums_workspace_bytes", "ls3d_column_sums", "ls3d_token_attention_forward", "ls3d_token_attention_workspace_bytes", "ls3d_token_attention_backward", "ls3d_devoxelize", "ls3d_devoxelize_grid", "ls3d_devoxelize_grid_workspace_bytes", "ls3d_interpolate_rows", "ls3d_interpolate_rows_backward", "ls3d_interpolate_rows_backward_workspace_bytes", "ls3d_grid_gather", "ls3d_nchw_to_nhwc", "ls3d_complete_concat", "ls3d_sfam", "ls3d_cross_attn", "ls3d_sffm_decoder", "ls3d_sffm_memory", "ls3d_sffm_memory_trace", "ls3d_stamp", "ls3d_points_cp", "ls3d_points_cuv",
    "ls3d_dynamic_point_to_voxel_workspace_bytes", "ls3d_dynamic_point_to_voxel_index", "ls3d_dynamic_point_to_voxel_forward", "ls3d_dynamic_point_to_voxel_backward",
    "ls3d_cyl_voxelize", "ls3d_unique_sorted_workspace_bytes", "ls3d_unique_sorted", "ls3d_dyn_point_features", "ls3d_act_affine", "ls3d_tta_merge",
]


class Grid(ctypes.Structure):
    _fields_ = [("vs", ctypes.c_float * 3), ("lo", ctypes.c_float * 3), ("grid", ctypes.c_int32 * 3)]


class PointsLayout(ctypes.Structure):
    _fields_ = [("stride", ctypes.c_int32), ("xyz_col", ctypes.c_int32), ("batch_col", ctypes.c_int32),
                ("feat_col", ctypes.c_int32), ("n_feat", ctypes.c_int32)]


class Epilogue(ctypes.Structure):
    _fields_ = [("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p), ("res_pre", ctypes.c_void_p),
                ("res_pre_ld", ctypes.c_int32), ("pair", ctypes.c_void_p), ("pair_ld", ctypes.c_int32),
                ("relu", ctypes.c_int32), ("ln_gamma", ctypes.c_void_p), ("ln_beta", ctypes.c_void_p), ("ln_eps", ctypes.c_float)]


class TileChainLayer(ctypes.Structure):
    """ls3d_tile_chain_layer_t: one layer of ls3d_tile_conv_chain"""
    _fields_ = [("in_", ctypes.c_void_p), ("in_ld", ctypes.c_int32), ("w_packed", ctypes.c_void_p), ("cin", ctypes.c_int32), ("cout", ctypes.c_int32),
                ("epi", Epilogue), ("out", ctypes.c_void_p), ("out_ld", ctypes.c_int32)]


class TransVFELayer(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ("wqkv", "bqkv", "wo", "bo", "w1", "b1", "w2", "b2", "n1_gamma", "n1_beta", "n2_gamma",
                                                "n2_beta")] + [("n1_eps", ctypes.c_float), ("n2_eps", ctypes.c_float)]


class TransVFE(ctypes.Structure):
    _fields_ = [("w_embed", ctypes.c_void_p), ("b_embed", ctypes.c_void_p), ("w_compress", ctypes.c_void_p), ("b_compress", ctypes.c_void_p),
                ("layers", ctypes.POINTER(TransVFELayer)), ("num_layers", ctypes.c_int32), ("num_compressed", ctypes.c_int32),
                ("embed", ctypes.c_int32), ("heads", ctypes.c_int32), ("ffn", ctypes.c_int32), ("token_ld", ctypes.c_int32),
                ("planes", ctypes.c_int32), ("flags", ctypes.c_int32)]


class SffmLayer(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ("wq", "bq", "wo", "bo", "w1a", "w1b", "b1", "w2a", "w2b", "b2", "n2_gamma", "n2_beta", "n3_gamma",
                                                "n3_beta")] + [("n2_eps", ctypes.c_float), ("n3_eps", ctypes.c_float)] + \
        [(k, ctypes.c_void_p) for k in ("wq_planes", "wo_planes", "w1a_planes", "w1b_planes", "w2a_planes", "w2b_planes")]


class PointMlpLayer(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p), ("cin", ctypes.c_int32), ("cout", ctypes.c_int32),
                ("relu", ctypes.c_int32)]


class SffmMemoryLayer(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ("wqkv_t", "bqkv", "wo_t", "bo", "n1_gamma", "n1_beta", "wk_t", "bk", "wv_t", "bv")] + [("n1_eps", ctypes.c_float)]


class Sffm(ctypes.Structure):
    _fields_ = [("w_in", ctypes.c_void_p), ("b_in", ctypes.c_void_p), ("layers", ctypes.POINTER(SffmLayer)), ("num_layers", ctypes.c_int32),
                ("d_in", ctypes.c_int32), ("d_model", ctypes.c_int32), ("heads", ctypes.c_int32), ("ffn", ctypes.c_int32),
                ("norm_gamma", ctypes.c_void_p), ("norm_beta", ctypes.c_void_p), ("norm_eps", ctypes.c_float), ("attention", ctypes.c_int32),
                ("w_in_planes", ctypes.c_void_p), ("pt_off", ctypes.c_void_p), ("gemm_products", ctypes.c_int32)]


class LibraryMissing(RuntimeError):
    pass


_lib = None


def _configure(lib):
    lib.ls3d_version.restype = ctypes.c_char_p
    for name in ("ls3d_voxelize_hard_workspace_bytes", "ls3d_dynamic_scatter_workspace_bytes", "ls3d_dynamic_scatter_backward_workspace_bytes", "ls3d_segment_reduce_workspace_bytes",
                 "ls3d_rulebook_conv_workspace_bytes", "ls3d_devoxelize_grid_workspace_bytes", "ls3d_interpolate_rows_backward_workspace_bytes", "ls3d_gather_gemm_packed_floats",
                 "ls3d_spconv_wgrad_workspace_bytes", "ls3d_spconv_pairs_bytes", "ls3d_tile_plan_bytes", "ls3d_column_sums_workspace_bytes", "ls3d_token_attention_workspace_bytes", "ls3d_tile_conv_packed_bytes", "ls3d_tile_conv_packed_bytes_bf16", "ls3d_tile_plan_workspace_bytes", "ls3d_radix_sort_workspace_bytes",
                 "ls3d_tile_conv_workspace_bytes", "ls3d_tile_conv_counter_bytes", "ls3d_tile_conv_trace_bytes", "ls3d_tile_chain_state_bytes", "ls3d_transvfe_planes_bytes", "ls3d_transvfe_workspace_bytes",
                 "ls3d_seg_loss_workspace_bytes", "ls3d_seg_loss_saved_bytes", "ls3d_layer_norm_workspace_bytes", "ls3d_batch_norm_workspace_bytes",
                 "ls3d_dynamic_point_to_voxel_workspace_bytes", "ls3d_unique_sorted_workspace_bytes"):
        getattr(lib, name).restype = ctypes.c_size_t
    for name in EXPORTS:
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        if fn.restype is ctypes.c_int:
            fn.restype = ctypes.c_int
    return lib


def load(path=None):
    """Load (once) and return the HIP library.  Raises LibraryMissing when it has not been built
    (`python -m lidarseg3d_amd.build`)."""
    global _lib
    if _lib is None or path is not None:
        p = path or LIB_PATH
        if not os.path.exists(p):
            raise LibraryMissing(
                "%s not found: build it with `python -m lidarseg3d_amd.build` (hipcc, gfx950). "
                "lidarseg3d_amd has no CPU/PyTorch fallback." % p)
        _lib = _configure(ctypes.CDLL(p))
    return _lib


def use_library_for_testing(path):
    """TEST HOOK: make the Python host layer call another build of the same C ABI (tests/hipsim's host
    emulation of the kernels) so that host logic and kernel indexing can be exercised without a GPU.
    Never called by the package itself."""
    global _lib
    _lib = _configure(ctypes.CDLL(path)) if path else None
    return _lib


def check(rc, what):
    if rc != OK:
        raise RuntimeError("%s failed: %s" % (what, _ERR.get(rc, rc)))
