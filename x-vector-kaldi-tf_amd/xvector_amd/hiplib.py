"""ctypes binding of ``libxvector_hip.so`` (the C ABI declared in ``include/xvector_hip.h``).

There is NO fallback: if the shared library is missing, or no MI355X is visible, every entry point
raises.  torch is imported first so that the library's ``libamdhip64.so.7`` dependency resolves to the
HIP runtime torch already loaded (same SONAME) -- device pointers of torch tensors are then valid in
the kernels, and torch's current stream can be passed straight through.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("XVECTOR_HIP_LIB") or os.path.join(_HERE, "libxvector_hip.so")     # override: kernel experiments
ABI_VERSION = 23

# every symbol include/xvector_hip.h declares (tests check the .so exports all of them)
SYMBOLS = ("xv_version", "xv_last_error", "xv_set_tuning", "xv_pack_weights_f32", "xv_fold_bn_f32", "xv_tdnn_layer_f32",
           "xv_stats_pool_workspace_bytes", "xv_stats_pool_f32", "xv_fc_f32", "xv_fc_splitk_workspace_bytes", "xv_fc_splitk_f32", "xv_chunk_average_f32",
           "xv_packed_weights_bf16x3_bytes", "xv_pack_weights_bf16x3", "xv_pack_weights_bf16x3_many", "xv_split_row_bytes", "xv_split_encode_f32",
           "xv_split_decode_f32", "xv_tdnn_layer_bf16x3", "xv_tdnn_layer_bf16x3_sums", "xv_tdnn_layer_bf16x3_moments", "xv_fc_bf16x3",
           "xv_block_stats_bytes", "xv_tdnn_layer_pool_bf16x3", "xv_stats_pool_blocks_f32", "xv_tdnn_layer_pool_f32",
           "xv_packed_weights_rows_f32_floats", "xv_pack_weights_rows_f32", "xv_tdnn_layer_rows_f32",
           "xv_toom_supported", "xv_packed_weights_toom_f32_floats", "xv_pack_weights_toom_f32", "xv_tdnn_layer_toom_f32", "xv_tdnn_layer_toom_dilated_f32",
           "xv_packed_pair_bf16x3_bytes", "xv_pack_pair_bf16x3", "xv_tdnn_pair_pool_bf16x3",
           "xv_packed_first_bf16x3_bytes", "xv_pack_first_bf16x3", "xv_tdnn_first_bf16x3",
           "xv_packed_weights_f16bf8_bytes", "xv_pack_weights_f16bf8", "xv_split8_encode_f32", "xv_split8_decode_f32",
           "xv_tdnn_layer_f16bf8", "xv_tdnn_layer_pool_f16bf8", "xv_tdnn_first_f16bf8",
           "xv_packed_pair_f16bf8_bytes", "xv_pack_pair_f16bf8", "xv_tdnn_pair_pool_f16bf8",
           # training step
           "xv_chunk_moments_f32", "xv_merge_moments_f32", "xv_rows_affine_f32", "xv_rows_affine_split_f32", "xv_wgrad_workspace_bytes", "xv_wgrad_f32", "xv_wgrad_bf16x3",
           "xv_wgrad_bias_workspace_bytes", "xv_wgrad_bias_bf16x3",
           "xv_col_sums_workspace_bytes", "xv_col_sums_f32", "xv_bn_act_backward_f32", "xv_bn_act_backward_split_f32", "xv_pool_backward_f32",
           "xv_bn_act_backward_parts_f32", "xv_col_sums_merge_f32", "xv_pool_bn_act_backward_f32", "xv_bn_moments_fold_f32", "xv_bn_small_forward_f32", "xv_bn_small_backward_f32",
           "xv_softmax_ce_f32", "xv_adam_f32", "xv_ema_f32", "xv_axpy_f32", "xv_sumsq_workspace_bytes", "xv_sumsq_f32", "xv_dropout_f32", "xv_pack_minibatch_f32", "xv_minibatch_layout",
           "xv_prelu_backward_f32", "xv_l2_normalize_rows_f32", "xv_l2_normalize_backward_f32", "xv_am_margin_f32",
           # feature front-end
           "xv_cmn_sliding_scatter_f32",
           # self-attentive pooling
           "xv_attention_scores_f32", "xv_attention_softmax_f32", "xv_attention_pool_workspace_bytes", "xv_attention_pool_f32",
           "xv_attention_pool_backward_f32", "xv_attention_softmax_backward_f32", "xv_attention_scores_backward_f32")

FMT_F32, FMT_SPLIT, FMT_SPLIT8 = 0, 1, 2
SPLIT_PAD_BEFORE, SPLIT_PAD_AFTER = 8, 264
TUNE_TILE_ROWS = 1
TUNE_FIRST_TILES = 2
TUNE_FP32_GEMM = 3
TUNE_XCD_COLUMNS = 4

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_PRELU = 0, 1, 2, 3

_lib = None


class XvectorHipError(RuntimeError):
    pass


def load():
    """Load the shared library (no GPU needed for loading / symbol checks)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise XvectorHipError("HIP extension not built: %s is missing (run `python -c 'import __graft_entry__ as g; "
                              "g.build()'` or `make -C x-vector-kaldi-tf_amd/csrc`)" % SO_PATH)
    import torch  # noqa: F401  (loads torch's libamdhip64 first; see module docstring)
    lib = ctypes.CDLL(SO_PATH)
    vp, ci, cf, i64, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64, ctypes.c_size_t
    lib.xv_version.restype = ci
    lib.xv_version.argtypes = []
    lib.xv_last_error.restype = ctypes.c_char_p
    lib.xv_last_error.argtypes = []
    lib.xv_set_tuning.restype = ci
    lib.xv_set_tuning.argtypes = [ci, ci]
    lib.xv_pack_weights_f32.restype = ci
    lib.xv_pack_weights_f32.argtypes = [vp, ci, ci, vp, vp]
    lib.xv_fold_bn_f32.restype = ci
    lib.xv_fold_bn_f32.argtypes = [vp, vp, vp, vp, cf, ci, vp, vp, vp]
    lib.xv_tdnn_layer_f32.restype = ci
    lib.xv_tdnn_layer_f32.argtypes = [vp, i64, ci, ci, vp, vp, vp, vp, ci, vp, ci, ci, ci, vp, vp, ci, vp, vp]
    lib.xv_packed_weights_rows_f32_floats.restype = sz
    lib.xv_packed_weights_rows_f32_floats.argtypes = [ci, ci, ci, ci]
    lib.xv_pack_weights_rows_f32.restype = ci
    lib.xv_pack_weights_rows_f32.argtypes = [vp, ci, ci, ci, ci, vp, vp]
    lib.xv_tdnn_layer_rows_f32.restype = ci
    lib.xv_tdnn_layer_rows_f32.argtypes = [vp, i64, ci, ci, vp, vp, vp, vp, ci, vp, ci, ci, vp, vp, ci, vp]
    lib.xv_toom_supported.restype = ci
    lib.xv_toom_supported.argtypes = [ci, ci, ci, ci]
    lib.xv_packed_weights_toom_f32_floats.restype = sz
    lib.xv_packed_weights_toom_f32_floats.argtypes = [ci, ci, ci]
    lib.xv_pack_weights_toom_f32.restype = ci
    lib.xv_pack_weights_toom_f32.argtypes = [vp, ci, ci, ci, vp, vp]
    lib.xv_tdnn_layer_toom_f32.restype = ci
    lib.xv_tdnn_layer_toom_f32.argtypes = [vp, i64, ci, ci, vp, vp, vp, vp, ci, vp, ci, ci, vp, vp, ci, vp]
    lib.xv_tdnn_layer_toom_dilated_f32.restype = ci
    lib.xv_tdnn_layer_toom_dilated_f32.argtypes = [vp, i64, ci, ci, vp, vp, vp, vp, ci, vp, ci, ci, ci, vp, vp, ci, vp]
    lib.xv_stats_pool_workspace_bytes.restype = sz
    lib.xv_stats_pool_workspace_bytes.argtypes = [ci, ci, ci, ci]
    lib.xv_stats_pool_f32.restype = ci
    lib.xv_stats_pool_f32.argtypes = [vp, i64, ci, vp, vp, ci, ci, ci, cf, vp, vp, vp]
    lib.xv_fc_f32.restype = ci
    lib.xv_fc_f32.argtypes = [vp, ci, ci, vp, vp, vp, vp, ci, vp, ci, vp, vp, vp]
    lib.xv_fc_splitk_workspace_bytes.restype = sz
    lib.xv_fc_splitk_workspace_bytes.argtypes = [ci, ci, ci]
    lib.xv_fc_splitk_f32.restype = ci
    lib.xv_fc_splitk_f32.argtypes = [vp, ci, ci, vp, vp, vp, vp, ci, vp, ci, vp, vp, vp, vp]
    lib.xv_packed_weights_bf16x3_bytes.restype = sz
    lib.xv_packed_weights_bf16x3_bytes.argtypes = [ci, ci, ci]
    lib.xv_pack_weights_bf16x3.restype = ci
    lib.xv_pack_weights_bf16x3.argtypes = [vp, ci, ci, ci, vp, vp]
    lib.xv_pack_weights_bf16x3_many.restype = ci
    lib.xv_pack_weights_bf16x3_many.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.xv_split_row_bytes.restype = sz
    lib.xv_split_row_bytes.argtypes = [ci]
    lib.xv_split_encode_f32.restype = ci
    lib.xv_split_encode_f32.argtypes = [vp, i64, ci, ci, vp, vp]
    lib.xv_split_decode_f32.restype = ci
    lib.xv_split_decode_f32.argtypes = [vp, i64, ci, vp, ci, vp]
    lib.xv_tdnn_layer_bf16x3.restype = ci
    lib.xv_tdnn_layer_bf16x3.argtypes = [vp, ci, i64, ci, ci, vp, vp, vp, vp, ci, vp, ci, ci, ci, vp, vp, ci, ci, vp, ci, vp]
    lib.xv_fc_bf16x3.restype = ci
    lib.xv_fc_bf16x3.argtypes = [vp, ci, ci, vp, vp, vp, vp, ci, vp, ci, vp, vp, vp]
    lib.xv_block_stats_bytes.restype = ctypes.c_size_t
    lib.xv_block_stats_bytes.argtypes = [i64, ci]
    lib.xv_tdnn_layer_pool_f32.restype = ci
    lib.xv_tdnn_layer_pool_f32.argtypes = [vp, i64, ci, ci, vp, vp, vp, vp, ci, vp, ci, ci, ci, vp, vp, vp]
    lib.xv_tdnn_layer_pool_bf16x3.restype = ci
    lib.xv_tdnn_layer_pool_bf16x3.argtypes = [vp, ci, i64, ci, ci, vp, vp, vp, vp, ci, vp, ci, ci, ci, vp, vp, vp]
    lib.xv_packed_first_bf16x3_bytes.restype = sz
    lib.xv_packed_first_bf16x3_bytes.argtypes = [ci, ci, ci]
    lib.xv_pack_first_bf16x3.restype = ci
    lib.xv_pack_first_bf16x3.argtypes = [vp, ci, ci, ci, vp, vp]
    lib.xv_tdnn_first_bf16x3.restype = ci
    lib.xv_tdnn_first_bf16x3.argtypes = [vp, i64, ci, ci, vp, vp, vp, vp, ci, vp, ci, ci, ci, vp, vp, vp]
    lib.xv_packed_weights_f16bf8_bytes.restype = sz
    lib.xv_packed_weights_f16bf8_bytes.argtypes = [ci, ci, ci]
    lib.xv_pack_weights_f16bf8.restype = ci
    lib.xv_pack_weights_f16bf8.argtypes = [vp, ci, ci, ci, vp, vp]
    lib.xv_split8_encode_f32.restype = ci
    lib.xv_split8_encode_f32.argtypes = [vp, i64, ci, ci, vp, vp, vp]
    lib.xv_split8_decode_f32.restype = ci
    lib.xv_split8_decode_f32.argtypes = [vp, i64, ci, vp, ci, vp]
    lib.xv_tdnn_layer_f16bf8.restype = ci
    lib.xv_tdnn_layer_f16bf8.argtypes = [vp, i64, ci, vp, vp, vp, vp, ci, vp, ci, ci, ci, vp, vp, ci, ci, vp, vp]
    lib.xv_tdnn_layer_pool_f16bf8.restype = ci
    lib.xv_tdnn_layer_pool_f16bf8.argtypes = [vp, i64, ci, vp, vp, vp, vp, ci, vp, ci, ci, ci, vp, vp, vp]
    lib.xv_tdnn_first_f16bf8.restype = ci
    lib.xv_tdnn_first_f16bf8.argtypes = [vp, i64, ci, ci, vp, vp, vp, vp, ci, vp, ci, ci, ci, vp, vp, vp, vp]
    lib.xv_packed_pair_f16bf8_bytes.restype = sz
    lib.xv_packed_pair_f16bf8_bytes.argtypes = [ci, ci, ci]
    lib.xv_pack_pair_f16bf8.restype = ci
    lib.xv_pack_pair_f16bf8.argtypes = [vp, vp, ci, ci, ci, vp, vp]
    lib.xv_tdnn_pair_pool_f16bf8.restype = ci
    lib.xv_tdnn_pair_pool_f16bf8.argtypes = [vp, i64, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp]
    lib.xv_packed_pair_bf16x3_bytes.restype = sz
    lib.xv_packed_pair_bf16x3_bytes.argtypes = [ci, ci, ci]
    lib.xv_pack_pair_bf16x3.restype = ci
    lib.xv_pack_pair_bf16x3.argtypes = [vp, vp, ci, ci, ci, vp, vp]
    lib.xv_tdnn_pair_pool_bf16x3.restype = ci
    lib.xv_tdnn_pair_pool_bf16x3.argtypes = [vp, i64, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp]
    lib.xv_stats_pool_blocks_f32.restype = ci
    lib.xv_stats_pool_blocks_f32.argtypes = [vp, ci, vp, vp, ci, cf, vp, vp]
    lib.xv_chunk_average_f32.restype = ci
    lib.xv_chunk_average_f32.argtypes = [vp, vp, vp, ci, ci, vp, vp]
    lib.xv_chunk_moments_f32.restype = ci
    lib.xv_chunk_moments_f32.argtypes = [vp, i64, ci, vp, vp, ci, ci, ci, vp, vp, vp]
    lib.xv_merge_moments_f32.restype = ci
    lib.xv_merge_moments_f32.argtypes = [vp, vp, ci, ci, vp, vp, vp]
    lib.xv_rows_affine_f32.restype = ci
    lib.xv_rows_affine_f32.argtypes = [vp, ci, i64, ci, vp, vp, vp, vp, ci, vp]
    lib.xv_rows_affine_split_f32.restype = ci
    lib.xv_rows_affine_split_f32.argtypes = [vp, ci, i64, ci, vp, vp, vp, vp, ci, vp, vp]
    lib.xv_wgrad_workspace_bytes.restype = sz
    lib.xv_wgrad_workspace_bytes.argtypes = [i64, ci, ci, ci]
    lib.xv_wgrad_f32.restype = ci
    lib.xv_wgrad_f32.argtypes = [vp, ci, vp, ci, i64, ci, ci, ci, ci, vp, vp, vp]
    lib.xv_wgrad_bf16x3.restype = ci
    lib.xv_wgrad_bf16x3.argtypes = lib.xv_wgrad_f32.argtypes
    lib.xv_wgrad_bias_workspace_bytes.restype = sz
    lib.xv_wgrad_bias_workspace_bytes.argtypes = [i64, ci, ci, ci]
    lib.xv_wgrad_bias_bf16x3.restype = ci
    lib.xv_wgrad_bias_bf16x3.argtypes = [vp, ci, vp, ci, i64, ci, ci, ci, ci, vp, vp, vp, vp]
    lib.xv_col_sums_workspace_bytes.restype = sz
    lib.xv_col_sums_workspace_bytes.argtypes = [i64, ci]
    lib.xv_col_sums_f32.restype = ci
    lib.xv_col_sums_f32.argtypes = [vp, ci, vp, ci, i64, ci, vp, vp, vp, vp]
    lib.xv_bn_act_backward_f32.restype = ci
    lib.xv_bn_act_backward_f32.argtypes = [vp, vp, ci, i64, ci, vp, vp, vp, vp, vp, cf, cf, ci, cf, vp, vp, vp, vp, vp, vp]
    lib.xv_bn_act_backward_split_f32.restype = ci
    lib.xv_bn_act_backward_split_f32.argtypes = [vp, vp, ci, i64, ci, vp, vp, vp, vp, vp, cf, cf, ci, cf, vp, vp, vp, vp, vp, vp, vp]
    lib.xv_tdnn_layer_bf16x3_sums.restype = ci
    lib.xv_tdnn_layer_bf16x3_sums.argtypes = [vp, ci, i64, ci, ci, vp, vp, vp, vp, ci, vp, ci, ci, ci, vp, vp, ci, vp, ci, vp, vp]
    lib.xv_tdnn_layer_bf16x3_moments.restype = ci
    lib.xv_tdnn_layer_bf16x3_moments.argtypes = [vp, ci, i64, ci, ci, vp, vp, vp, vp, ci, vp, ci, ci, ci, vp, vp, ci, vp, ci, vp, vp]
    lib.xv_bn_moments_fold_f32.restype = ci
    lib.xv_bn_moments_fold_f32.argtypes = [vp, i64, ci, cf, vp, vp, cf, vp, vp, vp, vp, vp]
    lib.xv_bn_small_forward_f32.restype = ci
    lib.xv_bn_small_forward_f32.argtypes = [vp, ci, ci, ci, vp, vp, cf, vp, vp, vp, ci, vp]
    lib.xv_bn_small_backward_f32.restype = ci
    lib.xv_bn_small_backward_f32.argtypes = [vp, vp, ci, ci, ci, vp, vp, vp, cf, ci, cf, vp, vp, vp, vp]
    lib.xv_bn_act_backward_parts_f32.restype = ci
    lib.xv_bn_act_backward_parts_f32.argtypes = [vp, vp, ci, i64, ci, vp, vp, vp, vp, cf, cf, ci, cf, vp, vp, vp, vp, vp, vp, vp]
    lib.xv_col_sums_merge_f32.restype = ci
    lib.xv_col_sums_merge_f32.argtypes = [vp, i64, ci, vp, vp, vp]
    lib.xv_pool_bn_act_backward_f32.restype = ci
    lib.xv_pool_bn_act_backward_f32.argtypes = [vp, vp, ci, ci, vp, vp, ci, i64, vp, vp, vp, vp, vp, vp, cf, cf, ci, cf, vp, vp, vp, vp, vp, vp]
    lib.xv_pool_backward_f32.restype = ci
    lib.xv_pool_backward_f32.argtypes = [vp, ci, ci, vp, vp, ci, i64, vp, vp, vp, vp]
    lib.xv_softmax_ce_f32.restype = ci
    lib.xv_softmax_ce_f32.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp]
    lib.xv_adam_f32.restype = ci
    lib.xv_adam_f32.argtypes = [vp, vp, vp, vp, i64, cf, cf, cf, cf, vp]
    lib.xv_ema_f32.restype = ci
    lib.xv_ema_f32.argtypes = [vp, vp, ci, cf, vp]
    lib.xv_axpy_f32.restype = ci
    lib.xv_axpy_f32.argtypes = [vp, vp, cf, i64, vp]
    lib.xv_sumsq_workspace_bytes.restype = sz
    lib.xv_sumsq_workspace_bytes.argtypes = [i64]
    lib.xv_sumsq_f32.restype = ci
    lib.xv_sumsq_f32.argtypes = [vp, i64, vp, vp, vp]
    lib.xv_dropout_f32.restype = ci
    lib.xv_dropout_f32.argtypes = [vp, ci, i64, ci, ctypes.c_uint64, cf, vp]
    lib.xv_minibatch_layout.restype = ci
    lib.xv_minibatch_layout.argtypes = [ci, ci, ci, i64, vp, vp, vp, vp]
    lib.xv_pack_minibatch_f32.restype = ci
    lib.xv_pack_minibatch_f32.argtypes = [vp, ci, ci, ci, ci, ci, ci, vp, i64, vp]
    lib.xv_prelu_backward_f32.restype = ci
    lib.xv_prelu_backward_f32.argtypes = [vp, vp, ci, i64, ci, vp, vp]
    lib.xv_l2_normalize_rows_f32.restype = ci
    lib.xv_l2_normalize_rows_f32.argtypes = [vp, ci, ci, ci, vp, ci, vp, vp]
    lib.xv_l2_normalize_backward_f32.restype = ci
    lib.xv_l2_normalize_backward_f32.argtypes = [vp, vp, vp, ci, ci, vp, vp]
    lib.xv_am_margin_f32.restype = ci
    lib.xv_am_margin_f32.argtypes = [vp, vp, ci, ci, cf, cf, vp]
    lib.xv_cmn_sliding_scatter_f32.restype = ci
    lib.xv_cmn_sliding_scatter_f32.argtypes = [vp, ci, ci, vp, vp, ci, ci, ci, ci, ci, vp, vp, ci, vp]
    lib.xv_attention_scores_f32.restype = ci
    lib.xv_attention_scores_f32.argtypes = [vp, i64, i64, ci, vp, vp, vp, i64, vp]
    lib.xv_attention_softmax_f32.restype = ci
    lib.xv_attention_softmax_f32.argtypes = [vp, vp, vp, ci, vp, vp]
    lib.xv_attention_pool_workspace_bytes.restype = sz
    lib.xv_attention_pool_workspace_bytes.argtypes = [ci, ci, ci, ci]
    lib.xv_attention_pool_f32.restype = ci
    lib.xv_attention_pool_f32.argtypes = [vp, i64, ci, vp, vp, vp, ci, ci, ci, cf, vp, vp, vp]
    lib.xv_attention_pool_backward_f32.restype = ci
    lib.xv_attention_pool_backward_f32.argtypes = [vp, i64, ci, vp, vp, vp, ci, ci, vp, vp, vp, i64, vp, vp]
    lib.xv_attention_softmax_backward_f32.restype = ci
    lib.xv_attention_softmax_backward_f32.argtypes = [vp, vp, vp, vp, ci, vp, vp]
    lib.xv_attention_scores_backward_f32.restype = ci
    lib.xv_attention_scores_backward_f32.argtypes = [vp, i64, vp, vp, i64, ci, vp, i64, vp]
    if lib.xv_version() != ABI_VERSION:
        raise XvectorHipError("libxvector_hip.so ABI version %d != expected %d" % (lib.xv_version(), ABI_VERSION))
    _lib = lib
    return lib


_GPU_SEEN = []


def require_gpu():
    """Load the library and insist on a visible GPU.  Called by every compute entry point.  (Only the POSITIVE answer is
    remembered: a training step makes ~150 calls, and torch.cuda.is_available() behind each of them was a tenth of the host
    time of a step; without a GPU every call still raises.)"""
    if _GPU_SEEN:
        return _GPU_SEEN[0]
    lib = load()
    import torch
    if not torch.cuda.is_available():
        raise XvectorHipError("no MI355X visible (torch.cuda.is_available() is False): the x-vector hot path "
                              "has no CPU fallback")
    _GPU_SEEN.append(lib)
    return lib


def set_tuning(key, value):
    """Process-wide launch tuning (``TUNE_*`` keys of the header); never changes results."""
    lib = load()
    _check(lib.xv_set_tuning(int(key), int(value)), "xv_set_tuning")


def _check(rc, what):
    if rc != 0:
        raise XvectorHipError("%s failed (%d): %s" % (what, rc, load().xv_last_error().decode()))


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_RAW_STREAM = []


def _stream():
    """The HIP stream torch considers current on the current device (the launches of this package go where torch's own work
    goes).  Through torch's raw-stream accessor when it has one: building a torch.cuda.Stream object per launch was a fifth of
    the host time of a training step."""
    import torch
    if not _RAW_STREAM:
        _RAW_STREAM.append(getattr(torch._C, "_cuda_getCurrentRawStream", None))
    raw = _RAW_STREAM[0]
    if raw is not None:
        return ctypes.c_void_p(raw(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, name):
    import torch
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "%s must be a contiguous cuda float32 tensor" % name
    return t


# ------------------------------------------------------------------------------------------------
# thin wrappers over the ABI, operating on torch-ROCm tensors (device memory + stream plumbing only)
# ------------------------------------------------------------------------------------------------
def pack_weights(w2d):
    """w2d: [Kred, Cout] -> packed [Cout, Kred]."""
    import torch
    lib = require_gpu()
    _f32(w2d, "w")
    kred, cout = w2d.shape
    wp = torch.empty((cout, kred), dtype=torch.float32, device=w2d.device)
    _check(lib.xv_pack_weights_f32(_ptr(w2d), kred, cout, _ptr(wp), _stream()), "xv_pack_weights_f32")
    return wp


class PackedRows(object):
    """Weights of a first layer for the "rows" form (xv_pack_weights_rows_f32): wp[Cout, roundup32(K ldx)] for feature rows of ldx floats."""

    def __init__(self, wp, K, cin, ldx, cout):
        self.wp, self.K, self.cin, self.ldx, self.cout = wp, K, cin, ldx, cout


def pack_weights_rows(w3d, ldx):
    """w3d: TF layout [K, Cin, Cout] (cuda float32), ldx: floats per packed feature row (>= Cin, a multiple of 4) -> PackedRows."""
    import torch
    lib = require_gpu()
    _f32(w3d, "w")
    K, cin, cout = w3d.shape
    n = int(lib.xv_packed_weights_rows_f32_floats(K, cin, int(ldx), cout))
    assert n > 0, "rows form unsupported for K=%d Cin=%d ldx=%d" % (K, cin, ldx)
    wp = torch.empty((cout, n // cout), dtype=torch.float32, device=w3d.device)
    _check(lib.xv_pack_weights_rows_f32(_ptr(w3d), K, cin, int(ldx), cout, _ptr(wp), _stream()), "xv_pack_weights_rows_f32")
    return PackedRows(wp, K, cin, int(ldx), cout)


def tdnn_layer_rows(x, w, bias, scale, shift, act, alpha, row_valid, y, rows=None):
    """xv_tdnn_layer_rows_f32: x[R, ldx] contiguous feature rows -> y[R, Cout] fp32 rows, w a PackedRows."""
    lib = require_gpu()
    _rows2d(x, "x")
    R = x.shape[0] if rows is None else int(rows)
    assert x.shape[1] == w.ldx and x.stride(0) == w.ldx and y.shape[1] == w.cout and y.shape[0] >= R
    if row_valid is not None:
        assert row_valid.is_cuda and row_valid.numel() >= R and row_valid.element_size() == 1
    _check(lib.xv_tdnn_layer_rows_f32(_ptr(x), R, w.cin, w.ldx, _ptr(w.wp), _ptr(bias), _ptr(scale), _ptr(shift), int(act), _ptr(alpha),
                                      w.K, w.cout, _ptr(row_valid), _ptr(y), y.stride(0), _stream()), "xv_tdnn_layer_rows_f32")


class PackedToom(object):
    """Transformed taps of one K in {3, 5, 7} layer for the Toom-Cook F(2, K) kernel (xv_pack_weights_toom_f32): wp[Cout, (K+1) Cin]."""

    def __init__(self, wp, K, cin, cout):
        self.wp, self.K, self.cin, self.cout = wp, K, cin, cout


def toom_supported(K, dilation, cin, cout):
    return bool(load().xv_toom_supported(int(K), int(dilation), int(cin), int(cout)))


def pack_weights_toom(w3d):
    """w3d: TF layout [K, Cin, Cout] (cuda float32) -> PackedToom."""
    import torch
    lib = require_gpu()
    _f32(w3d, "w")
    K, cin, cout = w3d.shape
    n = int(lib.xv_packed_weights_toom_f32_floats(K, cin, cout))
    assert n > 0, "Toom-Cook form unsupported for K=%d Cin=%d Cout=%d" % (K, cin, cout)
    wp = torch.empty((cout, (K + 1) * cin), dtype=torch.float32, device=w3d.device)
    _check(lib.xv_pack_weights_toom_f32(_ptr(w3d), K, cin, cout, _ptr(wp), _stream()), "xv_pack_weights_toom_f32")
    return PackedToom(wp, K, cin, cout)


def tdnn_layer_toom(x, w, bias, scale, shift, act, alpha, row_valid, y, rows=None, dilation=1):
    """xv_tdnn_layer_toom_dilated_f32: x[R, Cin] -> y[R, Cout] fp32 rows, w a PackedToom (chunks on multiples of 2 * dilation rows)."""
    lib = require_gpu()
    _rows2d(x, "x")
    R = x.shape[0] if rows is None else int(rows)
    assert x.shape[1] == w.cin and y.shape[1] == w.cout and y.shape[0] >= R
    if row_valid is not None:
        assert row_valid.is_cuda and row_valid.numel() >= R and row_valid.element_size() == 1
    if int(dilation) == 1:
        _check(lib.xv_tdnn_layer_toom_f32(_ptr(x), R, w.cin, x.stride(0), _ptr(w.wp), _ptr(bias), _ptr(scale), _ptr(shift), int(act),
                                          _ptr(alpha), w.K, w.cout, _ptr(row_valid), _ptr(y), y.stride(0), _stream()),
               "xv_tdnn_layer_toom_f32")
        return
    _check(lib.xv_tdnn_layer_toom_dilated_f32(_ptr(x), R, w.cin, x.stride(0), _ptr(w.wp), _ptr(bias), _ptr(scale), _ptr(shift), int(act),
                                              _ptr(alpha), w.K, int(dilation), w.cout, _ptr(row_valid), _ptr(y), y.stride(0), _stream()),
           "xv_tdnn_layer_toom_dilated_f32")


class Packed3(object):
    """Tiled bf16x3 weights of one layer (xv_pack_weights_bf16x3) + the shape they were packed for."""

    def __init__(self, wt, K, cin, cout):
        self.wt, self.K, self.cin, self.cout = wt, K, cin, cout


def pack_weights_bf16x3(w3d):
    """w3d: [K, Cin, Cout] fp32 (TF layout; FC: [1, In, Out]) -> Packed3."""
    import torch
    lib = require_gpu()
    _f32(w3d, "w")
    K, cin, cout = w3d.shape
    nbytes = int(lib.xv_packed_weights_bf16x3_bytes(K, cin, cout))
    wt = torch.empty(nbytes, dtype=torch.uint8, device=w3d.device)
    _check(lib.xv_pack_weights_bf16x3(_ptr(w3d), K, cin, cout, _ptr(wt), _stream()), "xv_pack_weights_bf16x3")
    return Packed3(wt, K, cin, cout)


class PackPlan(object):
    """The bf16x3 tiles of several layers, forward and (optionally) input-gradient orientation, re-packed by ONE launch
    (xv_pack_weights_bf16x3_many) into one persistent device buffer: what a training step does after every optimizer update.
    ``layers``: [(w3d [K, Cin, Cout] fp32 device tensor -- a VIEW whose storage the optimizer updates in place --, cin_pad, want_bwd)]."""

    def __init__(self, layers):
        import ctypes
        import torch
        lib = require_gpu()
        n = len(layers)
        self.n = n
        dev = layers[0][0].device
        sizes, offs, total = [], [], 0
        for w, cin_pad, want_bwd in layers:
            _f32(w, "w")
            K, cin, cout = w.shape
            assert cin_pad >= cin
            f = int(lib.xv_packed_weights_bf16x3_bytes(K, cin_pad, cout))
            b = int(lib.xv_packed_weights_bf16x3_bytes(K, cout, cin_pad)) if want_bwd else 0
            offs.append((total, total + f))
            total += f + b
            sizes.append((f, b))
        self.buf = torch.empty(total, dtype=torch.uint8, device=dev)
        base = self.buf.data_ptr()
        assert base % 16 == 0
        arr_p, arr_i = ctypes.c_void_p * n, ctypes.c_int32 * n
        self._w = arr_p(*[w.data_ptr() for w, _, _ in layers])
        self._K = arr_i(*[int(w.shape[0]) for w, _, _ in layers])
        self._cin = arr_i(*[int(w.shape[1]) for w, _, _ in layers])
        self._pad = arr_i(*[int(c) for _, c, _ in layers])
        self._cout = arr_i(*[int(w.shape[2]) for w, _, _ in layers])
        self._fwd = arr_p(*[base + o[0] for o in offs])
        self._bwd = arr_p(*[(base + o[1]) if sz[1] else None for o, sz in zip(offs, sizes)])
        self._keep = [w for w, _, _ in layers]
        self.fwd, self.bwd = [], []
        for (w, cin_pad, want_bwd), o, sz in zip(layers, offs, sizes):
            K, cin, cout = w.shape
            self.fwd.append(Packed3(self.buf[o[0]:o[0] + sz[0]], K, cin_pad, cout))
            self.bwd.append(Packed3(self.buf[o[1]:o[1] + sz[1]], K, cout, cin_pad) if want_bwd else None)

    def repack(self):
        lib = require_gpu()
        _check(lib.xv_pack_weights_bf16x3_many(self.n, self._w, self._K, self._cin, self._pad, self._cout, self._fwd, self._bwd,
                                               _stream()), "xv_pack_weights_bf16x3_many")


class SplitBuf(object):
    """Device buffer in a split activation format (``fmt``: FMT_SPLIT = bf16 hi/lo planes, FMT_SPLIT8 = fp16 hi + bf8
    cross bytes; same geometry) with the zero padding rows the layer kernels may read
    (rows [-SPLIT_PAD_BEFORE, rows + SPLIT_PAD_AFTER))."""

    def __init__(self, rows, channels, device, fmt=FMT_SPLIT):
        import torch
        self.rows, self.channels, self.fmt = int(rows), int(channels), int(fmt)
        self.row_bytes = int(load().xv_split_row_bytes(self.channels))
        self.base = torch.zeros((SPLIT_PAD_BEFORE + self.rows + SPLIT_PAD_AFTER) * self.row_bytes, dtype=torch.uint8, device=device)
        self.ptr = self.base.data_ptr() + SPLIT_PAD_BEFORE * self.row_bytes      # row 0

    def view(self, channels, fmt=None):
        """Same storage seen as a buffer of ``channels`` channels per row (<= allocated width), optionally in the other
        split format (all-zero bytes are zeros in both)."""
        v = object.__new__(SplitBuf)
        v.rows, v.channels, v.base = self.rows, int(channels), self.base
        v.fmt = self.fmt if fmt is None else int(fmt)
        v.row_bytes = int(load().xv_split_row_bytes(v.channels))
        assert v.row_bytes <= self.row_bytes
        v.ptr = self.base.data_ptr() + SPLIT_PAD_BEFORE * v.row_bytes
        return v


def split_encode(x, buf, rows=None, status=None):
    """fp32 rows x[R, C] -> buf (SplitBuf, either format; ``status``: int32 device tensor, FMT_SPLIT8 only)."""
    lib = require_gpu()
    _f32(x, "x")
    R = x.shape[0] if rows is None else int(rows)
    assert R <= buf.rows and x.shape[1] == buf.channels
    if buf.fmt == FMT_SPLIT8:
        _check(lib.xv_split8_encode_f32(_ptr(x), R, x.shape[1], x.stride(0), ctypes.c_void_p(buf.ptr), _ptr(status), _stream()),
               "xv_split8_encode_f32")
        return
    _check(lib.xv_split_encode_f32(_ptr(x), R, x.shape[1], x.stride(0), ctypes.c_void_p(buf.ptr), _stream()), "xv_split_encode_f32")


def split_decode(buf, rows, out=None):
    """SplitBuf -> fp32 tensor [rows, C]."""
    import torch
    lib = require_gpu()
    if out is None:
        out = torch.empty((rows, buf.channels), dtype=torch.float32, device=buf.base.device)
    fn = lib.xv_split8_decode_f32 if buf.fmt == FMT_SPLIT8 else lib.xv_split_decode_f32
    _check(fn(ctypes.c_void_p(buf.ptr), int(rows), buf.channels, _ptr(out), out.stride(0), _stream()), "xv_split_decode_f32")
    return out


def tdnn_layer3(x, R, w, bias, scale, shift, act, alpha, dilation, row_valid, y, y_preact=None):
    """bf16x3 layer.  x / y: contiguous fp32 tensors [>=R, C] (XV_FMT_F32) or SplitBuf (XV_FMT_SPLIT); w: Packed3."""
    lib = require_gpu()
    assert isinstance(w, Packed3)
    xs, ys = isinstance(x, SplitBuf), isinstance(y, SplitBuf)
    assert not (xs and x.fmt != FMT_SPLIT) and not (ys and y.fmt != FMT_SPLIT), "bf16x3 layers take the bf16 split format"
    if xs:
        assert x.channels == w.cin and x.rows >= R
        xp, ldx = ctypes.c_void_p(x.ptr), 0
    else:
        _rows2d(x, "x"); assert x.shape[1] == w.cin and x.shape[0] >= R
        xp, ldx = _ptr(x), x.stride(0)
    if y is None:
        yp, ldy = None, 0
    elif ys:
        assert y.channels == w.cout and y.rows >= R
        yp, ldy = ctypes.c_void_p(y.ptr), 0
    else:
        _rows2d(y, "y"); assert y.shape[1] == w.cout and y.shape[0] >= R
        yp, ldy = _ptr(y), y.stride(0)
    ldpre = 0
    if y_preact is not None:
        _f32(y_preact, "y_preact"); assert y_preact.shape[1] == w.cout and y_preact.shape[0] >= R
        ldpre = y_preact.stride(0)
    if row_valid is not None:
        assert row_valid.is_cuda and row_valid.numel() >= R and row_valid.element_size() == 1
    _check(lib.xv_tdnn_layer_bf16x3(xp, FMT_SPLIT if xs else FMT_F32, int(R), w.cin, ldx, _ptr(w.wt), _ptr(bias), _ptr(scale),
                                    _ptr(shift), int(act), _ptr(alpha), w.K, int(dilation), w.cout, _ptr(row_valid), yp,
                                    FMT_SPLIT if ys else FMT_F32, ldy, _ptr(y_preact), ldpre, _stream()), "xv_tdnn_layer_bf16x3")


def col_sums_workspace(rows, c, device):
    """Partial-sum workspace of xv_col_sums_f32 / xv_tdnn_layer_bf16x3_sums: [ceil(rows / 128)][2][c] doubles."""
    return _ws(load().xv_col_sums_workspace_bytes(int(rows), int(c)), device)


def tdnn_layer3_sums(x, R, w, dilation, row_valid, y, sum_r, workspace):
    """xv_tdnn_layer_bf16x3_sums: the plain GEMM y = x * w (no bias / activation: the input-gradient GEMM of the training step),
    fp32 rows out, and per 128-row tile the partial column sums [sum y | sum y * sum_r] in ``workspace`` (col_sums_workspace)."""
    lib = require_gpu()
    assert isinstance(w, Packed3) and supports_sums(w.cout)
    xs = isinstance(x, SplitBuf)
    if xs:
        assert x.fmt == FMT_SPLIT and x.channels == w.cin and x.rows >= R
        xp, ldx = ctypes.c_void_p(x.ptr), 0
    else:
        _rows2d(x, "x"); assert x.shape[1] == w.cin and x.shape[0] >= R
        xp, ldx = _ptr(x), x.stride(0)
    _rows2d(y, "y"); assert y.shape[1] == w.cout and y.shape[0] >= R
    _rows2d(sum_r, "sum_r"); assert sum_r.shape[1] == w.cout and sum_r.shape[0] >= R
    if row_valid is not None:
        assert row_valid.is_cuda and row_valid.numel() >= R and row_valid.element_size() == 1
    _check(lib.xv_tdnn_layer_bf16x3_sums(xp, FMT_SPLIT if xs else FMT_F32, int(R), w.cin, ldx, _ptr(w.wt), None, None, None, ACT_NONE,
                                         None, w.K, int(dilation), w.cout, _ptr(row_valid), _ptr(y), y.stride(0), _ptr(sum_r),
                                         sum_r.stride(0), _ptr(workspace), _stream()), "xv_tdnn_layer_bf16x3_sums")


def supports_sums(cout):
    return cout % 8 == 0


def tdnn_layer3_moments(x, R, w, bias, act, alpha, dilation, row_valid, y, y_preact, workspace):
    """xv_tdnn_layer_bf16x3_moments: the forward layer r = act(conv(x, w) + b) with fp32 rows out (+ the pre-activation), and per
    128-row tile the partial sums [sum r | sum r^2] in ``workspace`` (col_sums_workspace) -- BN's batch moments (bn_moments_fold)."""
    lib = require_gpu()
    assert isinstance(w, Packed3) and supports_sums(w.cout)
    xs = isinstance(x, SplitBuf)
    if xs:
        assert x.fmt == FMT_SPLIT and x.channels == w.cin and x.rows >= R
        xp, ldx = ctypes.c_void_p(x.ptr), 0
    else:
        _rows2d(x, "x"); assert x.shape[1] == w.cin and x.shape[0] >= R
        xp, ldx = _ptr(x), x.stride(0)
    _rows2d(y, "y"); assert y.shape[1] == w.cout and y.shape[0] >= R
    ldpre = 0
    if y_preact is not None:
        _f32(y_preact, "y_preact"); assert y_preact.shape[1] == w.cout and y_preact.shape[0] >= R
        ldpre = y_preact.stride(0)
    if row_valid is not None:
        assert row_valid.is_cuda and row_valid.numel() >= R and row_valid.element_size() == 1
    _check(lib.xv_tdnn_layer_bf16x3_moments(xp, FMT_SPLIT if xs else FMT_F32, int(R), w.cin, ldx, _ptr(w.wt), _ptr(bias), None, None, int(act),
                                            _ptr(alpha), w.K, int(dilation), w.cout, _ptr(row_valid), _ptr(y), y.stride(0), _ptr(y_preact),
                                            ldpre, _ptr(workspace), _stream()), "xv_tdnn_layer_bf16x3_moments")


def bn_moments_fold(workspace, rows, n_frames, gamma, beta, eps, mean, var):
    """(scale, shift) of the BN fold and the batch moments (into mean, var) from the partial sums of tdnn_layer3_moments."""
    import torch
    lib = require_gpu()
    c = gamma.numel()
    scale = torch.empty(c, dtype=torch.float32, device=gamma.device)
    shift = torch.empty(c, dtype=torch.float32, device=gamma.device)
    _check(lib.xv_bn_moments_fold_f32(_ptr(workspace), int(rows), c, float(n_frames), _ptr(_f32(gamma, "gamma")), _ptr(_f32(beta, "beta")),
                                      float(eps), _ptr(_f32(mean, "mean")), _ptr(_f32(var, "var")), _ptr(scale), _ptr(shift), _stream()),
           "xv_bn_moments_fold_f32")
    return scale, shift


POOL_BLOCK_ROWS = 8      # chunks fed to tdnn_layer_pool must start on a multiple of this many rows


def block_stats_floats(rows, cout):
    return int(load().xv_block_stats_bytes(int(rows), int(cout))) // 4


def tdnn_layer_pool(x, R, w, bias, scale, shift, act, alpha, dilation, row_valid, block_stats, K=None):
    """Last frame-level layer with the block-statistics epilogue: block_stats = flat fp32 tensor of
    >= block_stats_floats(R, cout) elements, laid out [ceil(R/8)][2][Cout].  w: Packed3 (bf16x3) or the packed fp32 weights
    of pack_weights() together with ``K`` (exact fp32, x: fp32 rows)."""
    lib = require_gpu()
    _f32(block_stats, "block_stats")
    if row_valid is not None:
        assert row_valid.is_cuda and row_valid.numel() >= R and row_valid.element_size() == 1
    if not isinstance(w, Packed3):
        _rows2d(x, "x"); _f32(w, "wp")
        cin, cout = x.shape[1], w.shape[0]
        assert K is not None and w.shape[1] == int(K) * cin and x.shape[0] >= R
        assert block_stats.numel() >= block_stats_floats(R, cout), "block_stats too small"
        _check(lib.xv_tdnn_layer_pool_f32(_ptr(x), int(R), cin, x.stride(0), _ptr(w), _ptr(bias), _ptr(scale), _ptr(shift), int(act),
                                          _ptr(alpha), int(K), int(dilation), cout, _ptr(row_valid), _ptr(block_stats), _stream()),
               "xv_tdnn_layer_pool_f32")
        return
    if isinstance(x, SplitBuf):
        assert x.channels == w.cin and x.rows >= R and x.fmt == FMT_SPLIT
        xp, ldx, fmt = ctypes.c_void_p(x.ptr), 0, FMT_SPLIT
    else:
        _f32(x, "x"); assert x.shape[1] == w.cin and x.shape[0] >= R
        xp, ldx, fmt = _ptr(x), x.stride(0), FMT_F32
    assert block_stats.numel() >= block_stats_floats(R, w.cout), "block_stats too small"
    _check(lib.xv_tdnn_layer_pool_bf16x3(xp, fmt, int(R), w.cin, ldx, _ptr(w.wt), _ptr(bias), _ptr(scale), _ptr(shift), int(act),
                                         _ptr(alpha), w.K, int(dilation), w.cout, _ptr(row_valid), _ptr(block_stats), _stream()),
           "xv_tdnn_layer_pool_bf16x3")


class PackedFirst(object):
    """Weights of the first frame-level layer in the fragment order of xv_tdnn_first_bf16x3."""

    def __init__(self, wt, K, cin, cout):
        self.wt, self.K, self.cin, self.cout = wt, K, cin, cout


def first_supported(K, cin, cout):
    return int(load().xv_packed_first_bf16x3_bytes(int(K), int(cin), int(cout))) > 0


def pack_first_bf16x3(w3d):
    """w[K, Cin, Cout] (device fp32, TF order) -> PackedFirst."""
    import torch
    lib = require_gpu()
    _f32(w3d, "w"); assert w3d.dim() == 3
    K, cin, cout = (int(v) for v in w3d.shape)
    nbytes = int(lib.xv_packed_first_bf16x3_bytes(K, cin, cout))
    if nbytes == 0:
        raise XvectorHipError("xv_pack_first_bf16x3: unsupported shape K=%d, %d -> %d" % (K, cin, cout))
    wt = torch.empty(nbytes, dtype=torch.uint8, device=w3d.device)
    _check(lib.xv_pack_first_bf16x3(_ptr(w3d.contiguous()), K, cin, cout, _ptr(wt), _stream()), "xv_pack_first_bf16x3")
    return PackedFirst(wt, K, cin, cout)


def tdnn_first(x, R, w, bias, scale, shift, act, alpha, dilation, row_valid, y, status=None):
    """First frame-level layer: x fp32 rows [>=R, ld] (ld % 8 == 0, columns >= Cin zero) -> y (SplitBuf; its format selects
    the entry point, ``status`` as for tdnn_layer8).  w: PackedFirst."""
    lib = require_gpu()
    assert isinstance(w, PackedFirst) and isinstance(y, SplitBuf)
    _rows2d(x, "x"); assert x.shape[0] >= R and x.shape[1] >= w.cin
    assert y.channels == w.cout and y.rows >= R
    if row_valid is not None:
        assert row_valid.is_cuda and row_valid.numel() >= R and row_valid.element_size() == 1
    if y.fmt == FMT_SPLIT8:
        _check(lib.xv_tdnn_first_f16bf8(_ptr(x), int(R), w.cin, x.stride(0), _ptr(w.wt), _ptr(bias), _ptr(scale), _ptr(shift), int(act),
                                        _ptr(alpha), w.K, int(dilation), w.cout, _ptr(row_valid), ctypes.c_void_p(y.ptr), _ptr(status),
                                        _stream()), "xv_tdnn_first_f16bf8")
        return
    _check(lib.xv_tdnn_first_bf16x3(_ptr(x), int(R), w.cin, x.stride(0), _ptr(w.wt), _ptr(bias), _ptr(scale), _ptr(shift), int(act),
                                    _ptr(alpha), w.K, int(dilation), w.cout, _ptr(row_valid), ctypes.c_void_p(y.ptr), _stream()),
           "xv_tdnn_first_bf16x3")


class Packed8(object):
    """Tiled f16bf8 weights of one layer (xv_pack_weights_f16bf8) + the shape they were packed for."""

    def __init__(self, wt, K, cin, cout):
        self.wt, self.K, self.cin, self.cout = wt, K, cin, cout


def f16bf8_supported(K, dilation):
    """Shapes xv_tdnn_layer_f16bf8 takes (the split-input shapes of the bf16x3 kernel)."""
    span = (int(K) - 1) * int(dilation)
    return int(K) in (1, 3, 5, 7) and (int(K) == 1 or 2 <= span <= 8)


def pack_weights_f16bf8(w3d):
    """w3d: [K, Cin, Cout] fp32 (TF layout) -> Packed8."""
    import torch
    lib = require_gpu()
    _f32(w3d, "w")
    K, cin, cout = w3d.shape
    nbytes = int(lib.xv_packed_weights_f16bf8_bytes(K, cin, cout))
    wt = torch.empty(nbytes, dtype=torch.uint8, device=w3d.device)
    _check(lib.xv_pack_weights_f16bf8(_ptr(w3d), K, cin, cout, _ptr(wt), _stream()), "xv_pack_weights_f16bf8")
    return Packed8(wt, K, cin, cout)


def tdnn_layer8(x, R, w, bias, scale, shift, act, alpha, dilation, row_valid, y, status=None):
    """f16bf8 layer.  x: SplitBuf in FMT_SPLIT8; y: contiguous fp32 tensor [>=R, C] or SplitBuf of either format; w: Packed8;
    status: int32 device tensor whose bit 0 is set when a FMT_SPLIT8 output had to be clamped (None: not reported)."""
    lib = require_gpu()
    assert isinstance(w, Packed8) and isinstance(x, SplitBuf) and x.fmt == FMT_SPLIT8
    assert x.channels == w.cin and x.rows >= R
    if isinstance(y, SplitBuf):
        assert y.channels == w.cout and y.rows >= R
        yp, ldy, yfmt = ctypes.c_void_p(y.ptr), 0, y.fmt
    else:
        _rows2d(y, "y"); assert y.shape[1] == w.cout and y.shape[0] >= R
        yp, ldy, yfmt = _ptr(y), y.stride(0), FMT_F32
    if row_valid is not None:
        assert row_valid.is_cuda and row_valid.numel() >= R and row_valid.element_size() == 1
    _check(lib.xv_tdnn_layer_f16bf8(ctypes.c_void_p(x.ptr), int(R), w.cin, _ptr(w.wt), _ptr(bias), _ptr(scale), _ptr(shift), int(act),
                                    _ptr(alpha), w.K, int(dilation), w.cout, _ptr(row_valid), yp, yfmt, ldy, _ptr(status), _stream()),
           "xv_tdnn_layer_f16bf8")


def tdnn_layer_pool8(x, R, w, bias, scale, shift, act, alpha, dilation, row_valid, block_stats):
    """tdnn_layer_pool in the f16bf8 arithmetic (x: SplitBuf in FMT_SPLIT8, w: Packed8)."""
    lib = require_gpu()
    assert isinstance(w, Packed8) and isinstance(x, SplitBuf) and x.fmt == FMT_SPLIT8
    assert x.channels == w.cin and x.rows >= R
    _f32(block_stats, "block_stats")
    assert block_stats.numel() >= block_stats_floats(R, w.cout), "block_stats too small"
    if row_valid is not None:
        assert row_valid.is_cuda and row_valid.numel() >= R and row_valid.element_size() == 1
    _check(lib.xv_tdnn_layer_pool_f16bf8(ctypes.c_void_p(x.ptr), int(R), w.cin, _ptr(w.wt), _ptr(bias), _ptr(scale), _ptr(shift),
                                         int(act), _ptr(alpha), w.K, int(dilation), w.cout, _ptr(row_valid), _ptr(block_stats),
                                         _stream()), "xv_tdnn_layer_pool_f16bf8")


class PackedPair(object):
    """Weights of two consecutive K = 1 layers in the stage order of xv_tdnn_pair_pool_bf16x3."""

    def __init__(self, wt, cin, cmid, cout):
        self.wt, self.cin, self.cmid, self.cout = wt, cin, cmid, cout


def pair_supported(cin, cmid, cout):
    return int(load().xv_packed_pair_bf16x3_bytes(int(cin), int(cmid), int(cout))) > 0


def pack_pair_bf16x3(w1, w2):
    """w1[Cin, Cmid], w2[Cmid, Cout] (device fp32, TF's [in, out] order) -> PackedPair."""
    import torch
    lib = require_gpu()
    _f32(w1, "w1"); _f32(w2, "w2")
    assert w1.dim() == 2 and w2.dim() == 2 and w1.shape[1] == w2.shape[0]
    cin, cmid, cout = int(w1.shape[0]), int(w1.shape[1]), int(w2.shape[1])
    nbytes = int(lib.xv_packed_pair_bf16x3_bytes(cin, cmid, cout))
    if nbytes == 0:
        raise XvectorHipError("xv_pack_pair_bf16x3: unsupported shape %d -> %d -> %d" % (cin, cmid, cout))
    wt = torch.empty(nbytes, dtype=torch.uint8, device=w1.device)
    _check(lib.xv_pack_pair_bf16x3(_ptr(w1.contiguous()), _ptr(w2.contiguous()), cin, cmid, cout, _ptr(wt), _stream()),
           "xv_pack_pair_bf16x3")
    return PackedPair(wt, cin, cmid, cout)


class PackedPair8(object):
    """Weights of two consecutive K = 1 layers in the stage order of xv_tdnn_pair_pool_f16bf8."""

    def __init__(self, wt, cin, cmid, cout):
        self.wt, self.cin, self.cmid, self.cout = wt, cin, cmid, cout


def pair8_supported(cin, cmid, cout):
    return int(load().xv_packed_pair_f16bf8_bytes(int(cin), int(cmid), int(cout))) > 0


def pack_pair_f16bf8(w1, w2):
    """w1[Cin, Cmid], w2[Cmid, Cout] (device fp32, TF's [in, out] order) -> PackedPair8."""
    import torch
    lib = require_gpu()
    _f32(w1, "w1"); _f32(w2, "w2")
    assert w1.dim() == 2 and w2.dim() == 2 and w1.shape[1] == w2.shape[0]
    cin, cmid, cout = int(w1.shape[0]), int(w1.shape[1]), int(w2.shape[1])
    nbytes = int(lib.xv_packed_pair_f16bf8_bytes(cin, cmid, cout))
    if nbytes == 0:
        raise XvectorHipError("xv_pack_pair_f16bf8: unsupported shape %d -> %d -> %d" % (cin, cmid, cout))
    wt = torch.empty(nbytes, dtype=torch.uint8, device=w1.device)
    _check(lib.xv_pack_pair_f16bf8(_ptr(w1.contiguous()), _ptr(w2.contiguous()), cin, cmid, cout, _ptr(wt), _stream()),
           "xv_pack_pair_f16bf8")
    return PackedPair8(wt, cin, cmid, cout)


def tdnn_pair_pool8(x, R, w, p1, p2, act, row_valid, block_stats, status=None):
    """tdnn_pair_pool in the f16bf8 arithmetic: x: SplitBuf in FMT_SPLIT8; w: PackedPair8; status: int32 device tensor (bit 0:
    the intermediate activation was clamped)."""
    lib = require_gpu()
    assert isinstance(x, SplitBuf) and isinstance(w, PackedPair8)
    assert x.channels == w.cin and x.rows >= R and x.fmt == FMT_SPLIT8
    _f32(block_stats, "block_stats"); assert block_stats.numel() >= block_stats_floats(R, w.cout)
    if row_valid is not None:
        assert row_valid.is_cuda and row_valid.numel() >= R and row_valid.element_size() == 1
    _check(lib.xv_tdnn_pair_pool_f16bf8(ctypes.c_void_p(x.ptr), int(R), w.cin, w.cmid, w.cout, _ptr(w.wt), _ptr(p1[0]), _ptr(p1[1]),
                                        _ptr(p1[2]), _ptr(p1[3]), _ptr(p2[0]), _ptr(p2[1]), _ptr(p2[2]), _ptr(p2[3]), int(act),
                                        _ptr(row_valid), _ptr(block_stats), _ptr(status), _stream()), "xv_tdnn_pair_pool_f16bf8")


def tdnn_pair_pool(x, R, w, p1, p2, act, row_valid, block_stats):
    """Two K = 1 layers + pooling block statistics in one launch.  x: SplitBuf; w: PackedPair; p1 / p2: (bias, bn_scale,
    bn_shift, act_alpha) device tensors (None allowed) of the first / second layer."""
    lib = require_gpu()
    assert isinstance(x, SplitBuf) and isinstance(w, PackedPair)
    assert x.channels == w.cin and x.rows >= R and x.fmt == FMT_SPLIT
    _f32(block_stats, "block_stats"); assert block_stats.numel() >= block_stats_floats(R, w.cout)
    if row_valid is not None:
        assert row_valid.is_cuda and row_valid.numel() >= R and row_valid.element_size() == 1
    _check(lib.xv_tdnn_pair_pool_bf16x3(ctypes.c_void_p(x.ptr), int(R), w.cin, w.cmid, w.cout, _ptr(w.wt), _ptr(p1[0]), _ptr(p1[1]),
                                        _ptr(p1[2]), _ptr(p1[3]), _ptr(p2[0]), _ptr(p2[1]), _ptr(p2[2]), _ptr(p2[3]), int(act),
                                        _ptr(row_valid), _ptr(block_stats), _stream()), "xv_tdnn_pair_pool_bf16x3")


def stats_pool_blocks(block_stats, c, row_start, row_len, nchunks, eps, out):
    import torch
    lib = require_gpu()
    _f32(block_stats, "block_stats"); _f32(out, "out")
    assert row_start.dtype == torch.int32 and row_len.dtype == torch.int32 and row_start.is_cuda and row_len.is_cuda
    assert out.shape[1] == 2 * c and out.shape[0] >= nchunks
    _check(lib.xv_stats_pool_blocks_f32(_ptr(block_stats), int(c), _ptr(row_start), _ptr(row_len), int(nchunks), float(eps),
                                        _ptr(out), _stream()), "xv_stats_pool_blocks_f32")


def fold_bn(gamma, beta, mean, var, eps):
    import torch
    lib = require_gpu()
    c = gamma.numel()
    scale = torch.empty(c, dtype=torch.float32, device=gamma.device)
    shift = torch.empty(c, dtype=torch.float32, device=gamma.device)
    _check(lib.xv_fold_bn_f32(_ptr(_f32(gamma, "gamma")), _ptr(_f32(beta, "beta")), _ptr(_f32(mean, "mean")),
                              _ptr(_f32(var, "var")), float(eps), c, _ptr(scale), _ptr(shift), _stream()),
           "xv_fold_bn_f32")
    return scale, shift


def tdnn_layer(x, wp, bias, scale, shift, act, alpha, K, dilation, row_valid, y, y_preact=None, rows=None):
    """x[R,Cin] -> y[R,Cout] (both contiguous 2-D cuda float32; only the first `rows` rows if given)."""
    lib = require_gpu()
    if isinstance(wp, Packed3):
        assert wp.K == K
        R = int(rows) if rows is not None else (x.rows if isinstance(x, SplitBuf) else x.shape[0])
        return tdnn_layer3(x, R, wp, bias, scale, shift, act, alpha, dilation, row_valid, y, y_preact)
    if isinstance(wp, PackedToom):
        assert wp.K == K and y_preact is None
        return tdnn_layer_toom(x, wp, bias, scale, shift, act, alpha, row_valid, y, rows, dilation)
    if isinstance(wp, PackedRows):
        assert wp.K == K and dilation == 1 and y_preact is None
        return tdnn_layer_rows(x, wp, bias, scale, shift, act, alpha, row_valid, y, rows)
    _rows2d(x, "x")
    _f32(wp, "wp")
    R = x.shape[0] if rows is None else int(rows)
    cin = x.shape[1]
    cout = wp.shape[0]
    assert wp.shape[1] == K * cin, "packed weight shape %s does not match K=%d Cin=%d" % (tuple(wp.shape), K, cin)
    out = y if y is not None else y_preact
    assert out.shape[1] == cout and out.shape[0] >= R
    if y is not None and y_preact is not None:
        assert y.shape[1] == y_preact.shape[1]
    if row_valid is not None:
        assert row_valid.is_cuda and row_valid.numel() >= R and row_valid.element_size() == 1
    _check(lib.xv_tdnn_layer_f32(_ptr(x), R, cin, x.stride(0), _ptr(wp), _ptr(bias), _ptr(scale), _ptr(shift), int(act),
                                 _ptr(alpha), int(K), int(dilation), cout, _ptr(row_valid), _ptr(y), out.stride(0),
                                 _ptr(y_preact), _stream()), "xv_tdnn_layer_f32")


def stats_pool_workspace_bytes(c, nchunks, max_len, split_rows):
    return int(load().xv_stats_pool_workspace_bytes(int(c), int(nchunks), int(max_len), int(split_rows)))


def stats_pool(h, row_start, row_len, nchunks, max_len, split_rows, eps, out, workspace=None):
    import torch
    lib = require_gpu()
    _f32(h, "h"); _f32(out, "out")
    assert row_start.dtype == torch.int32 and row_len.dtype == torch.int32 and row_start.is_cuda and row_len.is_cuda
    c = h.shape[1]
    assert out.shape[1] == 2 * c and out.shape[0] >= nchunks
    need = stats_pool_workspace_bytes(c, nchunks, max_len, split_rows)
    if need:
        assert workspace is not None and workspace.numel() * workspace.element_size() >= need, "pool workspace too small"
    _check(lib.xv_stats_pool_f32(_ptr(h), h.stride(0), c, _ptr(row_start), _ptr(row_len), int(nchunks), int(max_len),
                                 int(split_rows), float(eps), _ptr(out), _ptr(workspace), _stream()), "xv_stats_pool_f32")


def fc(x, wp, bias, scale, shift, act, alpha, y, y_preact, rows=None):
    lib = require_gpu()
    _f32(x, "x")
    split = isinstance(wp, Packed3)
    if not split:
        _f32(wp, "wp")
    n = x.shape[0] if rows is None else int(rows)
    out_dim, in_dim = (wp.cout, wp.cin) if split else wp.shape
    assert x.shape[1] == in_dim
    for t in (y, y_preact):
        if t is not None:
            _f32(t, "y"); assert t.shape[1] == out_dim and t.shape[0] >= n
    if split:
        _check(lib.xv_fc_bf16x3(_ptr(x), n, in_dim, _ptr(wp.wt), _ptr(bias), _ptr(scale), _ptr(shift), int(act),
                                _ptr(alpha), out_dim, _ptr(y), _ptr(y_preact), _stream()), "xv_fc_bf16x3")
        return
    _check(lib.xv_fc_f32(_ptr(x), n, in_dim, _ptr(wp), _ptr(bias), _ptr(scale), _ptr(shift), int(act), _ptr(alpha),
                         out_dim, _ptr(y), _ptr(y_preact), _stream()), "xv_fc_f32")


def fc_splitk_supported(nrows, in_dim, out_dim):
    """Is ``[nrows, in_dim] -> out_dim`` skinny enough for the split-K form (xv_fc_splitk_f32)?"""
    return int(load().xv_fc_splitk_workspace_bytes(int(nrows), int(in_dim), int(out_dim))) > 0


def fc_splitk(x, wp, bias, scale, shift, act, alpha, y, y_preact):
    """xv_fc_splitk_f32: the exact-fp32 FC of a skinny problem with its reduction dealt to groups of workgroups (training
    minibatches' segment level).  wp: fp32 packed weights [Out, In] (pack_weights).  NOT used by extraction: the sum order
    differs from xv_fc_f32's, and an utterance's bits must not depend on the batch it is in."""
    lib = require_gpu()
    _f32(x, "x"); _f32(wp, "wp")
    n, in_dim = x.shape
    out_dim = wp.shape[0]
    assert wp.shape[1] == in_dim
    for t in (y, y_preact):
        if t is not None:
            _f32(t, "y"); assert t.shape[1] == out_dim and t.shape[0] >= n
    ws = _ws(lib.xv_fc_splitk_workspace_bytes(n, in_dim, out_dim), x.device)
    _check(lib.xv_fc_splitk_f32(_ptr(x), n, in_dim, _ptr(wp), _ptr(bias), _ptr(scale), _ptr(shift), int(act), _ptr(alpha), out_dim,
                                _ptr(y), _ptr(y_preact), _ptr(ws), _stream()), "xv_fc_splitk_f32")


def chunk_average(e, seg_start, chunk_len, nutts, out):
    import torch
    lib = require_gpu()
    _f32(e, "e"); _f32(out, "out")
    assert seg_start.dtype == torch.int32 and chunk_len.dtype == torch.int32
    dim = e.shape[1]
    assert out.shape[1] == dim and out.shape[0] >= nutts and seg_start.numel() >= nutts + 1
    _check(lib.xv_chunk_average_f32(_ptr(e), _ptr(seg_start), _ptr(chunk_len), int(nutts), dim, _ptr(out), _stream()),
           "xv_chunk_average_f32")


# ------------------------------------------------------------------------------------------------
# training-step wrappers
# ------------------------------------------------------------------------------------------------
def _ws(nbytes, device):
    import torch
    return torch.empty(max(int(nbytes), 8), dtype=torch.uint8, device=device)


def chunk_moments(h, row_start, row_len, nchunks, max_len, out, split_rows=512):
    """out[B, 2C] = per-chunk [mean || biased variance]."""
    lib = require_gpu()
    _f32(h, "h"); _f32(out, "out")
    c = h.shape[1]
    ws = _ws(stats_pool_workspace_bytes(c, nchunks, max_len, split_rows), h.device)
    _check(lib.xv_chunk_moments_f32(_ptr(h), h.stride(0), c, _ptr(row_start), _ptr(row_len), int(nchunks), int(max_len),
                                    int(split_rows), _ptr(out), _ptr(ws), _stream()), "xv_chunk_moments_f32")


def merge_moments(cm, row_len, nchunks, mean, var):
    lib = require_gpu()
    c = cm.shape[1] // 2
    _check(lib.xv_merge_moments_f32(_ptr(_f32(cm, "cm")), _ptr(row_len), int(nchunks), c, _ptr(mean), _ptr(var), _stream()),
           "xv_merge_moments_f32")


def rows_affine(x, scale, shift, row_valid, y, rows=None, y_split=None):
    """y_split: optional SplitBuf (bf16 split format) that receives a second copy of y."""
    lib = require_gpu()
    R = x.shape[0] if rows is None else int(rows)
    if y_split is not None:
        assert y_split.fmt == FMT_SPLIT and y_split.channels == x.shape[1] and y_split.rows >= R
    _check(lib.xv_rows_affine_split_f32(_ptr(_f32(x, "x")), x.stride(0), R, x.shape[1], _ptr(scale), _ptr(shift), _ptr(row_valid),
                                        _ptr(_f32(y, "y")), y.stride(0), ctypes.c_void_p(y_split.ptr) if y_split is not None else None,
                                        _stream()), "xv_rows_affine_split_f32")


def wgrad_takes_bias(precision, x, dz):
    """Whether ``wgrad(..., db=)`` can leave the bias gradient too (xv_wgrad_bias_bf16x3: bf16x3, 32-bit buffer offsets)."""
    return precision == "bf16x3" and x.shape[0] * x.stride(0) * 4 < 2 ** 31 and dz.shape[0] * dz.stride(0) * 4 < 2 ** 31


def wgrad(x, dz, K, dilation, dw, precision="fp32", db=None):
    """dw[K, Cin, Cout] (contiguous) = sum_r x[r + tap shift] (x) dz[r]; precision "fp32" (exact fp32 MFMA) or "bf16x3".
    ``db`` (bf16x3 only, see wgrad_takes_bias): also db[Cout] = sum_r dz[r], out of the rows the kernel streams anyway."""
    lib = require_gpu()
    fn, name = (lib.xv_wgrad_bf16x3, "xv_wgrad_bf16x3") if precision == "bf16x3" else (lib.xv_wgrad_f32, "xv_wgrad_f32")
    _rows2d(x, "x"); _f32(dz, "dz"); _f32(dw, "dw")
    R, cin = x.shape
    cout = dz.shape[1]
    assert dz.shape[0] == R and tuple(dw.shape) == (K, cin, cout)
    if db is not None:
        assert wgrad_takes_bias(precision, x, dz) and _f32(db, "db").numel() == cout
        ws = _ws(lib.xv_wgrad_bias_workspace_bytes(R, cin, cout, K), x.device)
        _check(lib.xv_wgrad_bias_bf16x3(_ptr(x), x.stride(0), _ptr(dz), dz.stride(0), R, cin, cout, int(K), int(dilation), _ptr(dw), _ptr(db),
                                        _ptr(ws), _stream()), "xv_wgrad_bias_bf16x3")
        return
    ws = _ws(lib.xv_wgrad_workspace_bytes(R, cin, cout, K), x.device)
    _check(fn(_ptr(x), x.stride(0), _ptr(dz), dz.stride(0), R, cin, cout, int(K), int(dilation), _ptr(dw), _ptr(ws), _stream()), name)


def col_sums(a, b, sum_a, sum_ab=None):
    lib = require_gpu()
    _f32(a, "a")
    R, c = a.shape
    ws = _ws(lib.xv_col_sums_workspace_bytes(R, c), a.device)
    _check(lib.xv_col_sums_f32(_ptr(a), a.stride(0), _ptr(b), b.stride(0) if b is not None else 0, R, c, _ptr(sum_a), _ptr(sum_ab),
                               _ptr(ws), _stream()), "xv_col_sums_f32")


def bn_act_backward(dh, r, sum_dh, sum_dh_r, mean, var, gamma, eps, n_frames, act, alpha, row_valid, dgamma, dbeta, dz, dz_split=None):
    """dz_split: optional SplitBuf (bf16 split format) that receives a second copy of dz."""
    import torch
    lib = require_gpu()
    R, c = dh.shape
    coef = torch.empty(3 * c, dtype=torch.float32, device=dh.device)
    if dz_split is not None:
        assert dz_split.fmt == FMT_SPLIT and dz_split.channels == c and dz_split.rows >= R
    _check(lib.xv_bn_act_backward_split_f32(_ptr(_f32(dh, "dh")), _ptr(_f32(r, "r")), dh.stride(0), R, c, _ptr(sum_dh), _ptr(sum_dh_r),
                                            _ptr(mean), _ptr(var), _ptr(gamma), float(eps), float(n_frames), int(act), float(alpha),
                                            _ptr(row_valid), _ptr(dgamma), _ptr(dbeta), _ptr(coef), _ptr(_f32(dz, "dz")),
                                            ctypes.c_void_p(dz_split.ptr) if dz_split is not None else None, _stream()),
           "xv_bn_act_backward_split_f32")


def bn_act_backward_parts(dh, r, workspace, mean, var, gamma, eps, n_frames, act, alpha, row_valid, dgamma, dbeta, dz, dz_split=None):
    """bn_act_backward from the partial sums a producer left in ``workspace`` (tdnn_layer3_sums): no col_sums pass."""
    import torch
    lib = require_gpu()
    R, c = dh.shape
    coef = torch.empty(3 * c, dtype=torch.float32, device=dh.device)
    if dz_split is not None:
        assert dz_split.fmt == FMT_SPLIT and dz_split.channels == c and dz_split.rows >= R
    _check(lib.xv_bn_act_backward_parts_f32(_ptr(_f32(dh, "dh")), _ptr(_f32(r, "r")), dh.stride(0), R, c, _ptr(workspace), _ptr(mean),
                                            _ptr(var), _ptr(gamma), float(eps), float(n_frames), int(act), float(alpha), _ptr(row_valid),
                                            _ptr(dgamma), _ptr(dbeta), _ptr(coef), _ptr(_f32(dz, "dz")),
                                            ctypes.c_void_p(dz_split.ptr) if dz_split is not None else None, _stream()),
           "xv_bn_act_backward_parts_f32")


BN_SMALL_MAX_ROWS = 1024


def bn_small_forward(x, gamma, beta, eps, mean, var, y):
    """xv_bn_small_forward_f32: training-mode BN of a matrix of <= 1024 rows in one launch (moments into mean / var, y = BN(x))."""
    lib = require_gpu()
    _rows2d(x, "x"); _rows2d(y, "y")
    n, c = x.shape
    assert n <= BN_SMALL_MAX_ROWS and y.shape == x.shape and gamma.numel() == c
    _check(lib.xv_bn_small_forward_f32(_ptr(x), x.stride(0), n, c, _ptr(_f32(gamma, "gamma")), _ptr(_f32(beta, "beta")), float(eps),
                                       _ptr(_f32(mean, "mean")), _ptr(_f32(var, "var")), _ptr(y), y.stride(0), _stream()),
           "xv_bn_small_forward_f32")


def bn_small_backward(dh, r, mean, var, gamma, eps, act, alpha, dgamma, dbeta, dz):
    """xv_bn_small_backward_f32: the BN + activation backward of a matrix of <= 1024 rows in one launch."""
    lib = require_gpu()
    _rows2d(dh, "dh"); _rows2d(r, "r"); _rows2d(dz, "dz")
    n, c = dh.shape
    assert n <= BN_SMALL_MAX_ROWS and r.shape == dh.shape == dz.shape and dh.stride(0) == r.stride(0) == dz.stride(0)
    _check(lib.xv_bn_small_backward_f32(_ptr(dh), _ptr(r), dh.stride(0), n, c, _ptr(mean), _ptr(var), _ptr(gamma), float(eps), int(act),
                                        float(alpha), _ptr(dgamma), _ptr(dbeta), _ptr(dz), _stream()), "xv_bn_small_backward_f32")


def col_sums_merge(workspace, rows, c, sum_a, sum_ab=None):
    _check(require_gpu().xv_col_sums_merge_f32(_ptr(workspace), int(rows), int(c), _ptr(sum_a), _ptr(sum_ab), _stream()),
           "xv_col_sums_merge_f32")


def pool_bn_act_backward(h, r, row_start, row_len, nchunks, pooled, dpooled, chunk_moments_r, mean, var, gamma, eps, n_frames, act, alpha,
                         dgamma, dbeta, dz, dz_split=None):
    """Backward of [statistics pooling -> BN -> activation] of the last frame-level layer (xv_pool_bn_act_backward_f32)."""
    import torch
    lib = require_gpu()
    R, c = r.shape
    coef = torch.empty(3 * c, dtype=torch.float32, device=r.device)
    assert h.shape == r.shape and h.stride(0) == r.stride(0) == dz.stride(0)
    if dz_split is not None:
        assert dz_split.fmt == FMT_SPLIT and dz_split.channels == c and dz_split.rows >= R
    _check(lib.xv_pool_bn_act_backward_f32(_ptr(_f32(h, "h")), _ptr(_f32(r, "r")), r.stride(0), c, _ptr(row_start), _ptr(row_len),
                                           int(nchunks), R, _ptr(_f32(pooled, "pooled")), _ptr(_f32(dpooled, "dpooled")),
                                           _ptr(_f32(chunk_moments_r, "chunk_moments")), _ptr(mean), _ptr(var), _ptr(gamma), float(eps),
                                           float(n_frames), int(act), float(alpha), _ptr(dgamma), _ptr(dbeta), _ptr(coef),
                                           _ptr(_f32(dz, "dz")), ctypes.c_void_p(dz_split.ptr) if dz_split is not None else None,
                                           _stream()), "xv_pool_bn_act_backward_f32")


def pool_backward(h, row_start, row_len, nchunks, pooled, dpooled, dh):
    lib = require_gpu()
    _check(lib.xv_pool_backward_f32(_ptr(_f32(h, "h")), h.stride(0), h.shape[1], _ptr(row_start), _ptr(row_len), int(nchunks),
                                    h.shape[0], _ptr(_f32(pooled, "pooled")), _ptr(_f32(dpooled, "dpooled")), _ptr(_f32(dh, "dh")),
                                    _stream()), "xv_pool_backward_f32")


def softmax_ce(logits, labels, loss_acc, dlogits=None):
    import torch
    lib = require_gpu()
    B, N = logits.shape
    assert labels.dtype == torch.int32
    ws = torch.empty(2 * B, dtype=torch.float32, device=logits.device)
    _check(lib.xv_softmax_ce_f32(_ptr(_f32(logits, "logits")), _ptr(labels), B, N, _ptr(loss_acc), _ptr(ws), _ptr(dlogits), _stream()),
           "xv_softmax_ce_f32")


def adam(param, grad, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-8):
    lib = require_gpu()
    n = param.numel()
    assert grad.numel() == n and m.numel() == n and v.numel() == n and param.is_contiguous() and grad.is_contiguous()
    _check(lib.xv_adam_f32(_ptr(param), _ptr(grad), _ptr(m), _ptr(v), n, float(lr_t), float(beta1), float(beta2), float(eps), _stream()),
           "xv_adam_f32")


def ema(moving, batch, decay):
    lib = require_gpu()
    _check(lib.xv_ema_f32(_ptr(moving), _ptr(batch), moving.numel(), float(decay), _stream()), "xv_ema_f32")


def axpy(y, x, a):
    lib = require_gpu()
    assert y.numel() == x.numel() and y.is_contiguous() and x.is_contiguous()
    _check(lib.xv_axpy_f32(_ptr(y), _ptr(x), float(a), y.numel(), _stream()), "xv_axpy_f32")


def sumsq(x, out):
    lib = require_gpu()
    assert x.is_contiguous()
    ws = _ws(lib.xv_sumsq_workspace_bytes(x.numel()), x.device)
    _check(lib.xv_sumsq_f32(_ptr(x), x.numel(), _ptr(out), _ptr(ws), _stream()), "xv_sumsq_f32")


def minibatch_layout(B, T, gap, rows, device):
    """(row_start[B] int32, row_len[B] int32, row_valid[rows] uint8) of a minibatch of B chunks of T frames, generated on the device."""
    import torch
    lib = require_gpu()
    rs = torch.empty(B, dtype=torch.int32, device=device)
    rl = torch.empty(B, dtype=torch.int32, device=device)
    rv = torch.empty(rows, dtype=torch.uint8, device=device)
    _check(lib.xv_minibatch_layout(int(B), int(T), int(gap), int(rows), _ptr(rs), _ptr(rl), _ptr(rv), _stream()), "xv_minibatch_layout")
    return rs, rl, rv


def pack_minibatch(src, B, T, F, gap, dst):
    """src: device tensor holding a [B, T, F] minibatch (float16 or float32, contiguous) -> dst[rows, in_dim] float32: the packed
    rows with gaps (chunk b at rows gap + b*(T+gap)), everything else zero."""
    import torch
    lib = require_gpu()
    assert src.is_cuda and src.is_contiguous() and src.dtype in (torch.float16, torch.float32) and src.numel() >= B * T * F
    _f32(dst, "dst"); assert dst.dim() == 2 and dst.is_contiguous()
    _check(lib.xv_pack_minibatch_f32(_ptr(src), 1 if src.dtype == torch.float16 else 0, int(B), int(T), int(F), int(gap), dst.shape[1],
                                     _ptr(dst), dst.shape[0], _stream()), "xv_pack_minibatch_f32")


def dropout(x, seed, keep_prob, rows=None):
    """In-place tf.nn.dropout on x[R, C] with the stateless mask of include/xvector_hip.h (same call = backward)."""
    import torch
    lib = require_gpu()
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1, "x must be a row-major cuda float32 matrix"
    R = x.shape[0] if rows is None else int(rows)
    _check(lib.xv_dropout_f32(_ptr(x), x.stride(0), R, x.shape[1], int(seed) & 0xFFFFFFFFFFFFFFFF, float(keep_prob), _stream()),
           "xv_dropout_f32")


def dropout_mask_reference(seed, R, C, keep_prob):
    """NumPy restatement of the kernel's mask (bool [R, C]); used by tests and to hand the oracle the same mask."""
    import numpy as np
    idx = np.arange(1, R * C + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = np.uint64(int(seed) & 0xFFFFFFFFFFFFFFFF) ^ (idx * np.uint64(0x9E3779B97F4A7C15))
        x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    thr = min(float(np.float32(keep_prob)) * 4294967296.0, 4294967295.0)
    return ((x >> np.uint64(32)).astype(np.int64) < int(np.uint32(thr))).reshape(R, C)


def prelu_backward(dr, z, alpha):
    lib = require_gpu()
    _f32(dr, "dr"); _f32(z, "z"); _f32(alpha, "alpha")
    assert dr.shape == z.shape and dr.stride(0) == z.stride(0) and alpha.numel() == dr.shape[1]
    _check(lib.xv_prelu_backward_f32(_ptr(dr), _ptr(z), dr.stride(0), dr.shape[0], dr.shape[1], _ptr(alpha), _stream()),
           "xv_prelu_backward_f32")


def cmn_sliding_scatter(x, utt_start, utt_len, n_utts, max_len, cmn_window, center, min_window, dst_row, y):
    """Sliding-window CMN of the utterances in x[sum T, F] scattered to rows dst_row[t] of y (see include/xvector_hip.h)."""
    import torch
    lib = require_gpu()
    _f32(x, "x")
    assert y.is_cuda and y.dtype == torch.float32 and y.dim() == 2 and y.stride(1) == 1 and y.shape[1] >= x.shape[1]
    for t in (utt_start, utt_len, dst_row):
        assert t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()
    assert dst_row.numel() >= x.shape[0] and utt_start.numel() >= n_utts and utt_len.numel() >= n_utts
    _check(lib.xv_cmn_sliding_scatter_f32(_ptr(x), x.stride(0), x.shape[1], _ptr(utt_start), _ptr(utt_len), int(n_utts), int(max_len),
                                          int(cmn_window), 1 if center else 0, int(min_window), _ptr(dst_row), _ptr(y), y.stride(0),
                                          _stream()), "xv_cmn_sliding_scatter_f32")


def l2_normalize_rows(x):
    """-> (y = x / ||x|| per row, norms[R])."""
    import torch
    lib = require_gpu()
    _f32(x, "x")
    y = torch.empty_like(x)
    norm = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    _check(lib.xv_l2_normalize_rows_f32(_ptr(x), x.stride(0), x.shape[0], x.shape[1], _ptr(y), y.stride(0), _ptr(norm), _stream()),
           "xv_l2_normalize_rows_f32")
    return y, norm


def l2_normalize_backward(dy, y, norm):
    import torch
    lib = require_gpu()
    _f32(dy, "dy"); _f32(y, "y"); _f32(norm, "norm")
    assert dy.shape == y.shape and norm.numel() == y.shape[0]
    dx = torch.empty_like(y)
    _check(lib.xv_l2_normalize_backward_f32(_ptr(dy), _ptr(y), _ptr(norm), y.shape[0], y.shape[1], _ptr(dx), _stream()),
           "xv_l2_normalize_backward_f32")
    return dx


def am_margin(cosines, labels, scale, margin):
    import torch
    lib = require_gpu()
    _f32(cosines, "cosines")
    assert labels.is_cuda and labels.dtype == torch.int32 and labels.numel() == cosines.shape[0]
    _check(lib.xv_am_margin_f32(_ptr(cosines), _ptr(labels), cosines.shape[0], cosines.shape[1], float(scale), float(margin), _stream()),
           "xv_am_margin_f32")


# ------------------------------------------------------------------------------------------------
# self-attentive statistics pooling (local/tf/models.py:1036-1052)
# ------------------------------------------------------------------------------------------------
def _rows2d(t, name):
    """2-D cuda float32 tensor whose rows are contiguous (a column slice of a wider buffer is fine)."""
    import torch
    assert t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1, "%s must be cuda float32 rows" % name
    return t


def attention_scores(u, v, scores, nonlin=None, rows=None):
    """scores[r] = sum_c v[c]*tanh(u[r,c]); nonlin (optional, same shape as u) receives tanh(u)."""
    lib = require_gpu()
    _rows2d(u, "u"); _f32(v, "v"); _f32(scores, "scores")
    R = u.shape[0] if rows is None else int(rows)
    assert v.numel() == u.shape[1] and scores.numel() >= R
    ldn = 0
    if nonlin is not None:
        _rows2d(nonlin, "nonlin"); assert nonlin.shape[1] == u.shape[1] and nonlin.shape[0] >= R
        ldn = nonlin.stride(0)
    _check(lib.xv_attention_scores_f32(_ptr(u), u.stride(0), R, u.shape[1], _ptr(v), _ptr(scores), _ptr(nonlin), ldn, _stream()),
           "xv_attention_scores_f32")


def attention_softmax(scores, row_start, row_len, nchunks, att):
    import torch
    lib = require_gpu()
    _f32(scores, "scores"); _f32(att, "att")
    assert row_start.dtype == torch.int32 and row_len.dtype == torch.int32 and row_start.is_cuda and row_len.is_cuda
    assert att.numel() >= scores.numel()
    _check(lib.xv_attention_softmax_f32(_ptr(scores), _ptr(row_start), _ptr(row_len), int(nchunks), _ptr(att), _stream()),
           "xv_attention_softmax_f32")


def attention_pool_workspace_bytes(c, nchunks, max_len, split_rows):
    return int(load().xv_attention_pool_workspace_bytes(int(c), int(nchunks), int(max_len), int(split_rows)))


def attention_pool(h, att, row_start, row_len, nchunks, max_len, split_rows, eps, out, workspace=None):
    """out[b] = [sum_t att h | sqrt(sum_t att h^2 - (sum_t att h)^2 + eps)] over the rows of chunk b; h: [R, C] rows (may be a
    column slice of a wider buffer)."""
    import torch
    lib = require_gpu()
    _rows2d(h, "h"); _f32(att, "att"); _f32(out, "out")
    assert row_start.dtype == torch.int32 and row_len.dtype == torch.int32 and row_start.is_cuda and row_len.is_cuda
    c = h.shape[1]
    assert out.shape[1] == 2 * c and out.shape[0] >= nchunks and att.numel() >= h.shape[0]
    need = attention_pool_workspace_bytes(c, nchunks, max_len, split_rows)
    if need:
        assert workspace is not None and workspace.numel() * workspace.element_size() >= need, "attention pool workspace too small"
    _check(lib.xv_attention_pool_f32(_ptr(h), h.stride(0), c, _ptr(att), _ptr(row_start), _ptr(row_len), int(nchunks), int(max_len),
                                     int(split_rows), float(eps), _ptr(out), _ptr(workspace), _stream()), "xv_attention_pool_f32")


def attention_pool_backward(h, att, row_start, row_len, nchunks, max_len, pooled, dpooled, dh, datt):
    """h, dh: [R, C] rows (column slices allowed); pooled, dpooled: [nchunks, 2C]; datt: [R] (only chunk rows are written)."""
    lib = require_gpu()
    _rows2d(h, "h"); _rows2d(dh, "dh"); _f32(att, "att"); _f32(pooled, "pooled"); _f32(dpooled, "dpooled"); _f32(datt, "datt")
    c = h.shape[1]
    assert dh.shape == h.shape and pooled.shape[1] == 2 * c and dpooled.shape == pooled.shape and datt.numel() >= h.shape[0]
    _check(lib.xv_attention_pool_backward_f32(_ptr(h), h.stride(0), c, _ptr(att), _ptr(row_start), _ptr(row_len), int(nchunks),
                                              int(max_len), _ptr(pooled), _ptr(dpooled), _ptr(dh), dh.stride(0), _ptr(datt),
                                              _stream()), "xv_attention_pool_backward_f32")


def attention_softmax_backward(att, datt, row_start, row_len, nchunks, dscores):
    lib = require_gpu()
    _f32(att, "att"); _f32(datt, "datt"); _f32(dscores, "dscores")
    _check(lib.xv_attention_softmax_backward_f32(_ptr(att), _ptr(datt), _ptr(row_start), _ptr(row_len), int(nchunks), _ptr(dscores),
                                                 _stream()), "xv_attention_softmax_backward_f32")


def attention_scores_backward(nonlin, dscores, v, du):
    """du = dscores v (1 - nonlin^2); nonlin <- dscores * nonlin (column sums = dv)."""
    lib = require_gpu()
    _rows2d(nonlin, "nonlin"); _rows2d(du, "du"); _f32(dscores, "dscores"); _f32(v, "v")
    R, c = nonlin.shape
    assert du.shape == nonlin.shape and dscores.numel() >= R and v.numel() == c
    _check(lib.xv_attention_scores_backward_f32(_ptr(nonlin), nonlin.stride(0), _ptr(dscores), _ptr(v), R, c, _ptr(du), du.stride(0),
                                                _stream()), "xv_attention_scores_backward_f32")
