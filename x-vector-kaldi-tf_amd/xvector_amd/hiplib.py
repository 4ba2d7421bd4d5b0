"""ctypes binding of ``libxvector_hip.so`` (the C ABI declared in ``include/xvector_hip.h``).

There is NO fallback: if the shared library is missing, or no MI355X is visible, every entry point
raises.  torch is imported first so that the library's ``libamdhip64.so.7`` dependency resolves to the
HIP runtime torch already loaded (same SONAME) -- device pointers of torch tensors are then valid in
the kernels, and torch's current stream can be passed straight through.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libxvector_hip.so")
ABI_VERSION = 2

# every symbol include/xvector_hip.h declares (tests check the .so exports all of them)
SYMBOLS = ("xv_version", "xv_last_error", "xv_pack_weights_f32", "xv_fold_bn_f32", "xv_tdnn_layer_f32",
           "xv_stats_pool_workspace_bytes", "xv_stats_pool_f32", "xv_fc_f32", "xv_chunk_average_f32",
           "xv_pack_weights_bf16x3", "xv_tdnn_layer_bf16x3", "xv_fc_bf16x3")

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
    lib.xv_pack_weights_f32.restype = ci
    lib.xv_pack_weights_f32.argtypes = [vp, ci, ci, vp, vp]
    lib.xv_fold_bn_f32.restype = ci
    lib.xv_fold_bn_f32.argtypes = [vp, vp, vp, vp, cf, ci, vp, vp, vp]
    lib.xv_tdnn_layer_f32.restype = ci
    lib.xv_tdnn_layer_f32.argtypes = [vp, i64, ci, ci, vp, vp, vp, vp, ci, vp, ci, ci, ci, vp, vp, ci, vp, vp]
    lib.xv_stats_pool_workspace_bytes.restype = sz
    lib.xv_stats_pool_workspace_bytes.argtypes = [ci, ci, ci, ci]
    lib.xv_stats_pool_f32.restype = ci
    lib.xv_stats_pool_f32.argtypes = [vp, i64, ci, vp, vp, ci, ci, ci, cf, vp, vp, vp]
    lib.xv_fc_f32.restype = ci
    lib.xv_fc_f32.argtypes = [vp, ci, ci, vp, vp, vp, vp, ci, vp, ci, vp, vp, vp]
    lib.xv_pack_weights_bf16x3.restype = ci
    lib.xv_pack_weights_bf16x3.argtypes = [vp, ci, ci, vp, vp, vp]
    lib.xv_tdnn_layer_bf16x3.restype = ci
    lib.xv_tdnn_layer_bf16x3.argtypes = [vp, i64, ci, ci, vp, vp, vp, vp, vp, ci, vp, ci, ci, ci, vp, vp, ci, vp, vp]
    lib.xv_fc_bf16x3.restype = ci
    lib.xv_fc_bf16x3.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, ci, vp, ci, vp, vp, vp]
    lib.xv_chunk_average_f32.restype = ci
    lib.xv_chunk_average_f32.argtypes = [vp, vp, vp, ci, ci, vp, vp]
    if lib.xv_version() != ABI_VERSION:
        raise XvectorHipError("libxvector_hip.so ABI version %d != expected %d" % (lib.xv_version(), ABI_VERSION))
    _lib = lib
    return lib


def require_gpu():
    """Load the library and insist on a visible GPU.  Called by every compute entry point."""
    lib = load()
    import torch
    if not torch.cuda.is_available():
        raise XvectorHipError("no MI355X visible (torch.cuda.is_available() is False): the x-vector hot path "
                              "has no CPU fallback")
    return lib


def _check(rc, what):
    if rc != 0:
        raise XvectorHipError("%s failed (%d): %s" % (what, rc, load().xv_last_error().decode()))


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    import torch
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


def pack_weights_bf16x3(w2d):
    """w2d: [Kred, Cout] fp32 -> (hi, lo) bf16 planes [Cout, Kred] (stored as int16 tensors)."""
    import torch
    lib = require_gpu()
    _f32(w2d, "w")
    kred, cout = w2d.shape
    hi = torch.empty((cout, kred), dtype=torch.int16, device=w2d.device)
    lo = torch.empty((cout, kred), dtype=torch.int16, device=w2d.device)
    _check(lib.xv_pack_weights_bf16x3(_ptr(w2d), kred, cout, _ptr(hi), _ptr(lo), _stream()), "xv_pack_weights_bf16x3")
    return hi, lo


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
    _f32(x, "x")
    split = isinstance(wp, tuple)
    w0 = wp[0] if split else _f32(wp, "wp")
    R = x.shape[0] if rows is None else int(rows)
    cin = x.shape[1]
    cout = w0.shape[0]
    assert w0.shape[1] == K * cin, "packed weight shape %s does not match K=%d Cin=%d" % (tuple(w0.shape), K, cin)
    out = y if y is not None else y_preact
    assert out.shape[1] == cout and out.shape[0] >= R
    if y is not None and y_preact is not None:
        assert y.shape[1] == y_preact.shape[1]
    if row_valid is not None:
        assert row_valid.is_cuda and row_valid.numel() >= R and row_valid.element_size() == 1
    if split:
        _check(lib.xv_tdnn_layer_bf16x3(_ptr(x), R, cin, x.stride(0), _ptr(wp[0]), _ptr(wp[1]), _ptr(bias), _ptr(scale),
                                        _ptr(shift), int(act), _ptr(alpha), int(K), int(dilation), cout, _ptr(row_valid),
                                        _ptr(y), out.stride(0), _ptr(y_preact), _stream()), "xv_tdnn_layer_bf16x3")
        return
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
    split = isinstance(wp, tuple)
    w0 = wp[0] if split else _f32(wp, "wp")
    n = x.shape[0] if rows is None else int(rows)
    out_dim, in_dim = w0.shape
    assert x.shape[1] == in_dim
    for t in (y, y_preact):
        if t is not None:
            _f32(t, "y"); assert t.shape[1] == out_dim and t.shape[0] >= n
    if split:
        _check(lib.xv_fc_bf16x3(_ptr(x), n, in_dim, _ptr(wp[0]), _ptr(wp[1]), _ptr(bias), _ptr(scale), _ptr(shift), int(act),
                                _ptr(alpha), out_dim, _ptr(y), _ptr(y_preact), _stream()), "xv_fc_bf16x3")
        return
    _check(lib.xv_fc_f32(_ptr(x), n, in_dim, _ptr(wp), _ptr(bias), _ptr(scale), _ptr(shift), int(act), _ptr(alpha),
                         out_dim, _ptr(y), _ptr(y_preact), _stream()), "xv_fc_f32")


def chunk_average(e, seg_start, chunk_len, nutts, out):
    import torch
    lib = require_gpu()
    _f32(e, "e"); _f32(out, "out")
    assert seg_start.dtype == torch.int32 and chunk_len.dtype == torch.int32
    dim = e.shape[1]
    assert out.shape[1] == dim and out.shape[0] >= nutts and seg_start.numel() >= nutts + 1
    _check(lib.xv_chunk_average_f32(_ptr(e), _ptr(seg_start), _ptr(chunk_len), int(nutts), dim, _ptr(out), _stream()),
           "xv_chunk_average_f32")
