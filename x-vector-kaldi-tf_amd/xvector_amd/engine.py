"""Host side of the extraction hot path: chunk planning, ragged batching and the kernel sequence.

What the reference does per utterance inside ``Model.make_embedding`` (local/tf/models.py:373-423) --
validate the length, cut into chunks, one ``sess.run`` per chunk at batch 1, length-weighted average --
is done here for MANY utterances at once: chunks are packed into one ragged ``[R, C]`` matrix with zero
gap rows between them (see include/xvector_hip.h) and pushed through

    fp32 path    5 x xv_tdnn_layer_f32 -> xv_stats_pool_f32 -> xv_fc_f32 (embed_layer-0; optional embed_layer-1)
                 -> xv_chunk_average_f32
    bf16x3 path  4 x xv_tdnn_layer_bf16x3 (split-format activations) -> xv_tdnn_layer_pool_bf16x3 (last layer, 8-row
    (default)    block statistics instead of the activation) -> xv_stats_pool_blocks_f32 -> xv_fc_bf16x3
                 -> xv_chunk_average_f32

PyTorch is used for device memory, streams and H2D/D2H copies only; every arithmetic step of the path
is one of the HIP kernels behind the C ABI.
"""
import ctypes
import math
import os

import numpy as np

from . import hiplib
from . import topology as tp

_HOST = []


def _host_lib():
    """libxvector_host.so (csrc/xv_host.cpp, built next to local/tf/kaldi_io.py): the native batch packer.  Optional -- without
    it (or with an older build) batches are packed with NumPy."""
    if not _HOST:
        lib = None
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "local", "tf", "libxvector_host.so")
        try:
            if os.environ.get("XVECTOR_NO_HOST_LIB") == "1":
                raise OSError("disabled")
            lib = ctypes.CDLL(path)
            lib.xv_pack_rows_f32.restype = ctypes.c_int
            lib.xv_pack_rows_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]
            lib.xv_raw_row_plan.restype = ctypes.c_int
            lib.xv_raw_row_plan.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                                                                  ctypes.c_void_p, ctypes.c_int64]
        except (OSError, AttributeError):
            lib = None
        _HOST.append(lib)
    return _HOST[0]


def can_submit_raw(mats, feat_dim):
    """Extractor.submit_raw reads the raw matrices in place through the native packer: host library present and every matrix a
    C-contiguous float32 [T, feat_dim] array."""
    if _host_lib() is None:
        return False
    if hasattr(mats, "uniform_cols"):                     # kaldi_io.ArkMats: float32 matrices where the stream was read to
        return mats.uniform_cols() == feat_dim
    return all(m.dtype == np.float32 and m.ndim == 2 and m.shape[1] == feat_dim and m.flags.c_contiguous for m in mats)


def matrix_addresses(mats, feat_dim):
    """uint64 address of row 0 of every utterance matrix, or None when one of them is not a C-contiguous float32
    ``[T, feat_dim]`` array (the native packer reads them in place)."""
    if hasattr(mats, "uniform_cols"):
        return mats.addrs if mats.uniform_cols() == feat_dim else None
    out = np.empty(len(mats), dtype=np.uint64)
    for i, m in enumerate(mats):
        if m.dtype != np.float32 or m.ndim != 2 or m.shape[1] != feat_dim or not m.flags.c_contiguous:
            return None
        out[i] = m.__array_interface__["data"][0]
    return out


# ------------------------------------------------------------------------------------------------
# chunk planning -- local/tf/models.py:377-407
# ------------------------------------------------------------------------------------------------
def plan_chunks(num_rows, min_chunk_size, chunk_size):
    """[(start, length), ...] of the chunks the reference would run for an utterance of ``num_rows``
    frames, or None when it rejects the utterance (zero length / shorter than ``min_chunk_size``,
    models.py:378-387: a warning, nothing is written for the key)."""
    if num_rows == 0 or num_rows < min_chunk_size:
        return None
    this_chunk = chunk_size
    if num_rows < chunk_size:            # models.py:389-392
        this_chunk = num_rows
    elif chunk_size == -1:               # models.py:393-394
        this_chunk = num_rows
    num_chunks = int(math.ceil(num_rows / float(this_chunk)))       # models.py:396
    plan = []
    for i in range(num_chunks):
        length = min(this_chunk, num_rows - i * this_chunk)          # models.py:405 (tail is NOT shifted back)
        if length < min_chunk_size:                                   # models.py:406-407
            continue
        plan.append((i * this_chunk, length))
    return plan


def slot_rows(lengths, gap, align=1):
    """Rows a chunk occupies in a batch: its frames, >= ``gap`` zero rows, padded so the next chunk starts on a multiple
    of ``align`` rows."""
    lengths = np.asarray(lengths, dtype=np.int64)
    return (lengths + gap + align - 1) // align * align


def plan_chunk_table(lengths, min_chunk_size, chunk_size):
    """``plan_chunks`` for many utterances at once, as the flat chunk table the extractor works on: utterances that yield at
    least one chunk, ordered by length (stable), and their chunks in order.  Returns int64 arrays
    ``(order, c_utt, c_start, c_len, seg_start)``: chunk j belongs to utterance ``c_utt[j]`` and covers frames
    ``[c_start[j], c_start[j]+c_len[j])``; the chunks of ``order[k]`` are ``seg_start[k]:seg_start[k+1]``.
    Vectorised (the per-utterance Python loop was the serial prefix of every window); same rules, models.py:377-407."""
    T = np.asarray(lengths, dtype=np.int64)
    if chunk_size != -1 and chunk_size < min_chunk_size:
        # full-size chunks would be skipped too (models.py:406 applies to every chunk): rare enough for the plain loop
        plans = [plan_chunks(int(t), min_chunk_size, chunk_size) for t in T]
        order = np.array(sorted((i for i, p in enumerate(plans) if p), key=lambda i: T[i]), dtype=np.int64)
        c_utt = np.array([u for u in order for _ in plans[u]], dtype=np.int64)
        c_start = np.array([s for u in order for s, _ in plans[u]], dtype=np.int64)
        c_len = np.array([n for u in order for _, n in plans[u]], dtype=np.int64)
        seg = np.zeros(len(order) + 1, np.int64)
        np.cumsum([len(plans[u]) for u in order], out=seg[1:])
        return order, c_utt, c_start, c_len, seg
    valid = np.flatnonzero((T > 0) & (T >= min_chunk_size))
    order = valid[np.argsort(T[valid], kind="stable")]
    To = T[order]
    cs = To if chunk_size == -1 else np.where(To < chunk_size, To, chunk_size)          # models.py:388-394
    nch = -(-To // np.maximum(cs, 1))                                                      # ceil, models.py:396
    last = To - (nch - 1) * cs                                                             # tail chunk: simply shorter
    kept = nch - (last < min_chunk_size)                                                   # models.py:405-407
    seg = np.zeros(len(order) + 1, np.int64)
    np.cumsum(kept, out=seg[1:])
    c_utt = np.repeat(order, kept)
    idx = np.arange(int(seg[-1]), dtype=np.int64) - np.repeat(seg[:-1], kept)             # chunk index within its utterance
    csr = np.repeat(cs, kept)
    c_start = idx * csr
    c_len = np.where(idx == np.repeat(nch - 1, kept), np.repeat(last, kept), csr)
    return order, c_utt, c_start, c_len, seg


class BatchLayout(object):
    """Row layout of one ragged batch: chunk b owns rows [row_start[b], row_start[b]+row_len[b]); at least ``gap`` zero
    rows precede the first chunk and follow every chunk, and every chunk starts on a multiple of ``align`` rows
    (``align=8`` for the fused pooling epilogue, include/xvector_hip.h)."""

    def __init__(self, lengths, gap, align=1):
        lengths = np.asarray(lengths, dtype=np.int64)
        self.gap = int(gap)
        self.align = int(align)
        self.row_len = lengths.astype(np.int32)
        self.lead = (self.gap + self.align - 1) // self.align * self.align
        self.slots = slot_rows(lengths, self.gap, self.align)
        starts = np.empty(len(lengths), dtype=np.int64)
        if len(lengths):
            starts[0] = self.lead
            np.cumsum(self.slots[:-1], out=starts[1:])
            starts[1:] += self.lead
        self.row_start = starts.astype(np.int32)
        self.rows = int(self.lead + self.slots.sum())
        self.nchunks = len(lengths)
        self.max_len = int(lengths.max()) if len(lengths) else 0
        assert self.rows < 2 ** 31

    def row_valid(self, out=None):
        """uint8[rows]: 1 for frames, 0 for gap rows (one vectorised np.repeat)."""
        n = self.nchunks
        vals = np.zeros(2 * n + 1, dtype=np.uint8)
        vals[1::2] = 1
        reps = np.empty(2 * n + 1, dtype=np.int64)
        reps[0] = self.lead
        reps[1::2] = self.row_len
        reps[2::2] = self.slots - self.row_len
        m = np.repeat(vals, reps)
        if out is not None:
            out[:self.rows] = m
            return out[:self.rows]
        return m

    def pack(self, mats, out):
        """Copy the chunk matrices into ``out[rows, >=F]`` and zero the gap rows.  Columns >= F of ``out`` are not
        touched (callers keep them zero).  One np.concatenate instead of a Python loop over chunks."""
        if not mats:
            out[:self.rows] = 0
            return
        F = mats[0].shape[1]
        z = np.zeros((self.gap + self.align, F), dtype=out.dtype)
        parts = [z[:self.lead]] * (2 * len(mats) + 1)
        parts[1::2] = mats
        if self.align > 1:
            parts[2::2] = [z[:g] for g in (self.slots - self.row_len).tolist()]
        elif self.lead != self.gap:
            raise AssertionError
        np.concatenate(parts, axis=0, out=out[:self.rows, :F])


# ------------------------------------------------------------------------------------------------
# device model
# ------------------------------------------------------------------------------------------------
class _ColumnView(object):
    """weights[name][cols] for the per-channel vectors of a layer."""

    def __init__(self, weights, cols):
        self.weights, self.cols = weights, cols

    def __getitem__(self, name):
        return self.weights[name][self.cols]


class _Layer(dict):
    """Parameters of one layer in kernel layout.  ``wp`` (the bf16x3 / fp32 packing of the weights) may be deferred: an
    f16bf8 model runs its frame-level layers from other packings (first-layer, f16bf8, pair kernels) and only the debug
    helpers ever ask for it -- uploading and packing 17 MB for nothing was a third of the model load."""

    def __missing__(self, key):
        if key == "wp" and "_make_wp" in self:
            self["wp"] = dict.pop(self, "_make_wp")()
            return self["wp"]
        raise KeyError(key)


class DeviceModel(object):
    """Weights of one model directory resident in HBM in kernel layout, plus reusable activation
    buffers.  ``weights`` is keyed by the TF variable names (weights.py)."""

    POOL_SPLIT_ROWS = 512

    def __init__(self, weights, topo, device="cuda:0", embedding_index=0, precision="bf16x3", fused_pool=None, pair_kernel=None):
        """precision: "fp32" = exact fp32 MFMA GEMMs; "bf16x3" = split-precision bf16 MFMA GEMMs (fp32-class
        accuracy, ~3e-6 rel-L2 on the x-vector; see include/xvector_hip.h).  fused_pool (default: on; fp32 needs a last layer whose
        width is a multiple of 4; never with attention pooling): the last frame-level layer reduces its output to 8-row block
        statistics in the GEMM epilogue instead of storing it (xv_tdnn_layer_pool_bf16x3 / _f32); batches must then be laid
        out with ``align`` = 8.  pair_kernel (default: on with
        fused_pool when the last two layers have kernel size 1 and a shape xv_tdnn_pair_pool_bf16x3 supports;
        XVECTOR_PAIR_KERNEL=0 turns it off): those two layers and the block statistics run as one launch whose
        intermediate activation stays in registers."""
        import torch
        hiplib.require_gpu()
        assert precision in ("fp32", "fp32tc", "bf16x3", "f16bf8")
        # "fp32tc": the exact-fp32 path with the K = 3 / 5 / 7 layers formed as Toom-Cook F(2, K) over time (xv_tdnn_layer_toom_f32,
        # _dilated_f32: 4 / 6 / 8 transformed products per row PAIR instead of 6 / 10 / 14, exact fp32 products; same accuracy class
        # as the direct form).  Every other layer, pooling and the FCs are the "fp32" kernels.  Row pairs sit on even rows:
        # batches are laid out with chunk starts on even rows (align >= 2; 8 with the fused pooling epilogue).
        self.toom = precision == "fp32tc"
        # "f16bf8": the hidden frame-level layers form a product as one fp16 MFMA + one block-scaled bf8 MFMA instead of
        # three bf16 MFMAs (xv_tdnn_layer_f16bf8; ~1e-5 rel-L2 on the x-vector).  Everything else (first layer, segment FCs,
        # the pair kernel, pooling) is the bf16x3 path.  A topology the f16bf8 kernels do not cover runs as plain bf16x3;
        # a batch whose activations leave the fp16 range raises ``status`` and is repeated by ``fallback()``.
        self.requested_precision = precision
        if self.toom:
            precision = "fp32"
        self.f16bf8 = False
        if precision == "f16bf8":
            precision = "bf16x3"
            ks, ds = list(topo["kernel_sizes"]), list(topo["dilations"])
            w0 = weights["frame_level_info_layer-0/w:0"]
            # (layer 0 stays in bf16x3: on the first-layer kernel where its shape fits -- K * ceil8(Cin) <= 128 --, else on the
            # general kernel + one encoding pass, e.g. 30-dimensional MFCCs with K = 5)
            self.f16bf8 = (not tp.is_attention(topo) and fused_pool in (None, True) and len(ks) >= 3 and (ks[0] - 1) * ds[0] <= 8 and
                           all(hiplib.f16bf8_supported(k, d) for k, d in zip(ks[1:], ds[1:])))
        self.precision = precision
        self._weights = weights if self.f16bf8 else None
        self._fallback = None
        self.attention = tp.is_attention(topo)
        # fused pooling: the last layer's GEMM epilogue reduces 8-row blocks to (mean, M2) instead of storing the layer (bf16x3 /
        # f16bf8 kernels, and the exact-fp32 GEMM when its width allows 16-byte stores)
        can_fuse = not self.attention and (precision == "bf16x3" or int(topo["layer_sizes"][-1]) % 4 == 0)
        self.fused_pool = can_fuse if fused_pool is None else bool(fused_pool)
        assert not (self.fused_pool and precision == "fp32" and int(topo["layer_sizes"][-1]) % 4), "fused fp32 pooling needs Cout % 4 == 0"
        assert not (self.fused_pool and self.attention), "the fused epilogue computes plain statistics, not attention-weighted ones"
        self.align = hiplib.POOL_BLOCK_ROWS if self.fused_pool else (2 if self.toom else 1)
        self.torch = torch
        self.device = torch.device(device)
        self.topo = topo
        self.embedding_index = int(embedding_index)
        self.gap = tp.max_halo(topo)
        self.act = tp.ACT_CODES[topo.get("activation", "relu")]
        self.feat_dim = int(weights["frame_level_info_layer-0/w:0"].shape[1])
        # features are packed with the column count rounded up to a multiple of 4 (23 -> 24, extra columns
        # zero, matching zero weight rows) so that layer 0 takes the 16-byte vector staging path
        self.in_dim = (self.feat_dim + 3) // 4 * 4
        self.layers = []
        self._uploaded = {}
        with torch.cuda.device(self.device):
            for i, (k, d) in enumerate(zip(topo["kernel_sizes"], topo["dilations"])):
                sc = "frame_level_info_layer-%d" % i
                w = weights[sc + "/w:0"]
                assert w.shape[0] == k, "kernel size mismatch in %s" % sc
                if i == 0 and self.in_dim != self.feat_dim:
                    wpad = np.zeros((k, self.in_dim, w.shape[2]), np.float32)
                    wpad[:, :self.feat_dim] = w
                    w = wpad
                # (the last layer's pooling epilogue exists on the direct fp32 kernel only: no Toom-Cook form for it)
                last = i == len(topo["kernel_sizes"]) - 1
                self.layers.append(self._prep(weights, sc, w, k, d, defer_wp=self.f16bf8, toom_ok=not (last and self.fused_pool)))
                if isinstance(self.layers[-1].get("wp"), hiplib.PackedToom) and d > 1:
                    # a dilated Toom-Cook layer pairs the rows (r, r + d) with floor(r / d) even: chunks on multiples of 2 d rows, so
                    # that a chunk's pairing -- its bits -- does not depend on where in a batch it lies (d = 3: 24-row alignment)
                    import math
                    self.align = math.lcm(self.align, 2 * int(d))
            if self.attention:
                # models.py:1036-1046: h = [h1 | h2]; u = h1 . attention/w + attention/b is one more K=1 layer.  On the
                # bf16x3 path the last frame-level layer runs as two launches over the two column halves of its weights so
                # that h1 comes out in the split format (input of the attention GEMM) and h2 as fp32 rows (pooling input)
                A = self.layers[-1]["cout"] // 2
                self.att_dim = A
                aw = weights["attention/w:0"]
                assert aw.shape == (A, A), "attention/w:0 must be [%d, %d]" % (A, A)
                self.att = dict(K=1, dil=1, cin=A, cout=A, bias=self._dev(weights["attention/b:0"]),
                                v=self._dev(weights["attention/v:0"]))
                if precision == "bf16x3":
                    self.att["wp"] = hiplib.pack_weights_bf16x3(self._dev(aw[None, :, :]))
                    sc = "frame_level_info_layer-%d" % (len(self.layers) - 1)
                    k, d = topo["kernel_sizes"][-1], topo["dilations"][-1]
                    self.last_halves = [self._prep(weights, sc, weights[sc + "/w:0"], k, d, cols=slice(a, a + A)) for a in (0, A)]
                else:
                    self.att["wp"] = hiplib.pack_weights(self._dev(aw))
            # layer 0 on the kernel built for it (output-stream bound; see csrc/xv_first.hip) when its shape allows
            self.first = None
            L0 = self.layers[0]
            if precision == "bf16x3" and os.environ.get("XVECTOR_FIRST_KERNEL", "1") != "0" and len(self.layers) > 1 and \
                    self.in_dim % 8 == 0 and hiplib.first_supported(L0["K"], self.in_dim, L0["cout"]) and (L0["K"] - 1) * L0["dil"] <= 8:
                w0 = weights["frame_level_info_layer-0/w:0"]
                wpad = np.zeros((L0["K"], self.in_dim, w0.shape[2]), np.float32)
                wpad[:, :self.feat_dim] = w0
                self.first = hiplib.pack_first_bf16x3(self._dev(wpad))
            self.pair = None
            want_pair = (os.environ.get("XVECTOR_PAIR_KERNEL", "1") != "0") if pair_kernel is None else bool(pair_kernel)
            if want_pair and self.fused_pool and precision == "bf16x3" and len(self.layers) >= 3:
                La, Lb = self.layers[-2], self.layers[-1]
                if La["K"] == 1 and Lb["K"] == 1 and hiplib.pair_supported(La["cin"], La["cout"], Lb["cout"]):
                    n = len(self.layers)
                    wa = self._wdev(weights, "frame_level_info_layer-%d/w:0" % (n - 2))[0]
                    wb = self._wdev(weights, "frame_level_info_layer-%d/w:0" % (n - 1))[0]
                    self.pair = hiplib.pack_pair_bf16x3(wa, wb)
            assert not (pair_kernel and self.pair is None), "pair_kernel=True but the topology / precision does not allow it"
            self.pair8 = None
            if self.f16bf8 and self.pair is not None and os.environ.get("XVECTOR_PAIR8_KERNEL", "1") != "0":
                La, Lb = self.layers[-2], self.layers[-1]
                if hiplib.pair8_supported(La["cin"], La["cout"], Lb["cout"]):
                    n = len(self.layers)
                    self.pair8 = hiplib.pack_pair_f16bf8(self._wdev(weights, "frame_level_info_layer-%d/w:0" % (n - 2))[0],
                                                         self._wdev(weights, "frame_level_info_layer-%d/w:0" % (n - 1))[0])
            if self.f16bf8:
                if self.first is None:
                    self.layers[0]["wp"]                          # (deferred above: this model does use layer 0's bf16x3 packing)
                n = len(self.layers)
                for i in range(1, n - 2 if self.pair is not None else n):
                    sc = "frame_level_info_layer-%d" % i
                    self.layers[i]["wp8"] = hiplib.pack_weights_f16bf8(self._wdev(weights, sc + "/w:0"))
                self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
            self.embed = []
            for j in range(len(topo["embedding_sizes"])):
                sc = "embed_layer-%d" % j
                self.embed.append(self._prep(weights, sc, weights[sc + "/w:0"][None, :, :], 1, 1))
            torch.cuda.synchronize()
        self._uploaded = {}                             # (the packings are what stays; the plain copies go)
        self.pooled_dim = tp.pooled_dim(topo)
        self.embed_dim = self.embed[self.embedding_index]["cout"]
        self._cap_rows = 0
        self._cap_chunks = 0
        self._pool_ws = None
        self._a0 = None

    @property
    def arithmetic(self):
        """Name of the arithmetic this model runs ("f16bf8" falls back to "bf16x3" on a topology its kernels do not cover)."""
        return "f16bf8" if self.f16bf8 else ("fp32tc" if self.toom else self.precision)

    def probe_vectors(self, batch):
        """x-vectors (host float32 [chunks, E]) of a packed host batch ``(x[R, in_dim], row_start, row_len, row_valid, max_len)``
        in this model's arithmetic (load-time accuracy probe)."""
        torch = self.torch
        x, rs, rl, rv, max_len = batch
        with torch.cuda.device(self.device):
            dx, drs, drl, drv = (torch.from_numpy(np.ascontiguousarray(a)).to(self.device) for a in (x, rs, rl, rv))
            out = torch.empty((len(rs), self.embed_dim), dtype=torch.float32, device=self.device)
            if self.f16bf8:
                self.status.zero_()
            self.forward_packed(dx, drs, drl, drv, len(rs), int(max_len), out)
            return out.cpu().numpy()

    def fallback(self):
        """The bf16x3 twin of an f16bf8 model (built on first use): repeats a batch whose activations left the fp16 range."""
        if self._fallback is None:
            assert self.f16bf8
            self._fallback = DeviceModel(self._weights, self.topo, self.device, self.embedding_index, "bf16x3")
        return self._fallback

    def _dev(self, a):
        return self.torch.as_tensor(np.array(a, dtype=np.float32, order="C")).to(self.device)     # copy: sources may be read-only

    def _wdev(self, weights, key):
        """The array ``weights[key]`` on the device, uploaded once per model (several packings read the same weights)."""
        if key not in self._uploaded:
            self._uploaded[key] = self._dev(weights[key])
        return self._uploaded[key]

    def _prep(self, weights, scope, w3d, k, d, cols=slice(None), defer_wp=False, toom_ok=True):
        """Kernel-layout parameters of one layer (``cols``: only these output channels; ``defer_wp``: see _Layer)."""
        w3d = w3d[:, :, cols]
        weights = _ColumnView(weights, cols)
        layer = _Layer(K=k, dil=d, cin=w3d.shape[1], cout=w3d.shape[2])
        if self.precision == "bf16x3":
            make = lambda: hiplib.pack_weights_bf16x3(self._dev(w3d))                 # tiled hi/lo bf16
        elif self.toom and scope == "frame_level_info_layer-0" and d == 1 and w3d.shape[1] == self.in_dim and \
                w3d.shape[2] % 4 == 0 and os.environ.get("XVECTOR_ROWS_FIRST", "1") != "0":
            # layer 0 as a K = 1 GEMM over the overlapping windows of the packed feature rows (xv_tdnn_layer_rows_f32)
            make = lambda: hiplib.pack_weights_rows(self._dev(w3d), self.in_dim)
        elif self.toom and toom_ok and hiplib.toom_supported(k, d, w3d.shape[1], w3d.shape[2]):
            make = lambda: hiplib.pack_weights_toom(self._dev(w3d))                   # transformed taps [Cout, (K+1) Cin]
        else:
            make = lambda: hiplib.pack_weights(self._dev(w3d.reshape(-1, w3d.shape[2])))
        if defer_wp:
            layer["_make_wp"] = make
        else:
            layer["wp"] = make()
        layer["bias"] = self._dev(weights[scope + "/b:0"])
        layer["scale"], layer["shift"] = hiplib.fold_bn(*(self._dev(weights["%s/%s:0" % (scope, n)])
                                                          for n in ("gamma", "beta", "mean", "variance")),
                                                        tp.BN_EPSILON)
        layer["alpha"] = None
        if self.act == tp.ACT_LRELU:
            layer["alpha"] = self._dev(np.array([self.topo.get("lrelu_alpha", 0.2)]))
        elif self.act == tp.ACT_PRELU:
            layer["alpha"] = self._dev(weights[scope + "/prelu/prelu:0"])
        return layer

    # -- buffers ----------------------------------------------------------------------------------
    def reserve(self, rows, nchunks, max_len=None):
        """(Re)allocate activation buffers for batches of up to ``rows`` rows / ``nchunks`` chunks."""
        torch = self.torch
        if rows > self._cap_rows:
            self._cap_rows = int(rows)
            widths = sorted(set(l["cout"] for l in self.layers[:-1]))
            wmax = max(widths) if widths else self.layers[-1]["cout"]
            if self.precision == "bf16x3":
                # hidden activations live in the split-bf16 format (zero padding rows included)
                self._ping = hiplib.SplitBuf(self._cap_rows, wmax, self.device)
                self._pong = hiplib.SplitBuf(self._cap_rows, wmax, self.device)
            else:
                self._ping = torch.empty((self._cap_rows, wmax), dtype=torch.float32, device=self.device)
                self._pong = torch.empty((self._cap_rows, wmax), dtype=torch.float32, device=self.device)
            if self.f16bf8 and self.first is None:
                self._l0_rows = torch.empty((self._cap_rows, self.layers[0]["cout"]), dtype=torch.float32, device=self.device)
            if self.fused_pool:
                self._last = torch.empty(hiplib.block_stats_floats(self._cap_rows, self.layers[-1]["cout"]), dtype=torch.float32,
                                         device=self.device)
            elif self.attention:
                A = self.att_dim
                if self.precision == "bf16x3":
                    self._h1 = hiplib.SplitBuf(self._cap_rows, A, self.device)
                    self._last = torch.empty((self._cap_rows, A), dtype=torch.float32, device=self.device)          # h2
                else:
                    self._last = torch.empty((self._cap_rows, 2 * A), dtype=torch.float32, device=self.device)      # [h1 | h2]
                self._u = torch.empty((self._cap_rows, A), dtype=torch.float32, device=self.device)
                self._scores = torch.empty(self._cap_rows, dtype=torch.float32, device=self.device)
                self._att = torch.zeros(self._cap_rows, dtype=torch.float32, device=self.device)
            else:
                self._last = torch.empty((self._cap_rows, self.layers[-1]["cout"]), dtype=torch.float32, device=self.device)
        if nchunks > self._cap_chunks:
            self._cap_chunks = int(nchunks)
            self._pooled = torch.empty((self._cap_chunks, self.pooled_dim), dtype=torch.float32, device=self.device)
            self._pool_ws = None
        if max_len is not None and not self.fused_pool:
            if self.attention:
                need = hiplib.attention_pool_workspace_bytes(self.att_dim, self._cap_chunks, max_len, self.POOL_SPLIT_ROWS)
            else:
                need = hiplib.stats_pool_workspace_bytes(self.layers[-1]["cout"], self._cap_chunks, max_len, self.POOL_SPLIT_ROWS)
            if need and (self._pool_ws is None or self._pool_ws.numel() * 4 < need):
                self._pool_ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=self.device)

    def _view(self, buf, rows, width):
        # contiguous [rows, width] view on the front of a flat buffer (fp32) / narrower split view
        if isinstance(buf, hiplib.SplitBuf):
            return buf if buf.channels == width else buf.view(width)
        return buf.view(-1)[: rows * width].view(rows, width)

    # -- the kernel sequence ----------------------------------------------------------------------
    def frame_level(self, x, row_start, row_len, row_valid, nchunks, max_len, pooled, events=None, status=None):
        """Frame-level part for ONE ragged batch: x[R,in_dim] (gap rows zero) -> 5 TDNN layers -> statistics
        pooling, written to pooled[nchunks, 2*C_last].  Device tensors only; no allocation when ``reserve`` was
        called with sufficient capacity.  ``events``: optional 3 torch.cuda.Event recorded before the first layer,
        after the last layer and after pooling (bench.py's per-kernel timing).  ``status`` (f16bf8 models): int32 device
        tensor that collects the out-of-range flag of this batch (default: the model's own ``status`` word)."""
        R = x.shape[0]
        self.reserve(R, nchunks, max_len)
        h = x
        bufs = (self._ping, self._pong)
        if events is not None:
            events[0].record()
        if self.f16bf8:
            self._frame_level_f16bf8(x, R, row_valid, self.status if status is None else status)
            if events is not None:
                events[1].record()
            hiplib.stats_pool_blocks(self._last, self.layers[-1]["cout"], row_start, row_len, nchunks, tp.VAR2STD_EPSILON, pooled)
            if events is not None:
                events[2].record()
            return pooled
        for i, L in enumerate(self.layers):
            last = i == len(self.layers) - 1
            if i == 0 and self.first is not None and isinstance(bufs[0], hiplib.SplitBuf):
                y = self._view(bufs[0], R, L["cout"])
                hiplib.tdnn_first(h, R, self.first, L["bias"], L["scale"], L["shift"], self.act, L["alpha"], L["dil"], row_valid, y)
                h = y
                continue
            if self.pair is not None and i == len(self.layers) - 2:
                Lb = self.layers[-1]
                hiplib.tdnn_pair_pool(h, R, self.pair, (L["bias"], L["scale"], L["shift"], L["alpha"]),
                                      (Lb["bias"], Lb["scale"], Lb["shift"], Lb["alpha"]), self.act, row_valid, self._last)
                break
            if last and self.fused_pool:
                hiplib.tdnn_layer_pool(h, R, L["wp"], L["bias"], L["scale"], L["shift"], self.act, L["alpha"], L["dil"], row_valid,
                                       self._last, K=L["K"])
                break
            if last and self.attention:
                h = self._attention_scores(h, R, L, row_valid)
                break
            y = self._view(self._last if last else bufs[i & 1], R, L["cout"])
            hiplib.tdnn_layer(h, L["wp"], L["bias"], L["scale"], L["shift"], self.act, L["alpha"], L["K"], L["dil"],
                              row_valid, y, rows=R)
            h = y
        if events is not None:
            events[1].record()
        if self.fused_pool:
            hiplib.stats_pool_blocks(self._last, self.layers[-1]["cout"], row_start, row_len, nchunks, tp.VAR2STD_EPSILON, pooled)
        elif self.attention:
            hiplib.attention_softmax(self._scores, row_start, row_len, nchunks, self._att)
            hiplib.attention_pool(h, self._att, row_start, row_len, nchunks, max_len, self.POOL_SPLIT_ROWS, tp.VAR2STD_EPSILON,
                                  pooled, self._pool_ws)
        else:
            hiplib.stats_pool(h, row_start, row_len, nchunks, max_len, self.POOL_SPLIT_ROWS, tp.VAR2STD_EPSILON, pooled,
                              self._pool_ws)
        if events is not None:
            events[2].record()
        return pooled

    def _frame_level_f16bf8(self, x, R, row_valid, status, marks=None):
        """first-layer kernel -> split8; hidden layers in the f16bf8 arithmetic; the layer in front of the pair kernel writes
        the format that kernel reads; block statistics end up in ``self._last``.  ``marks``: optional list that receives one
        ``(label, event)`` per launch, the event recorded right after it (bench.py's per-kernel breakdown)."""
        def mark(label):
            if marks is not None:
                ev = self.torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append((label, ev))
        n = len(self.layers)
        bufs = (self._ping, self._pong)
        S8, S3 = hiplib.FMT_SPLIT8, hiplib.FMT_SPLIT
        L = self.layers[0]
        stop = n - 2 if self.pair is not None else n - 1        # layers [1, stop) run on xv_tdnn_layer_f16bf8
        pair_fmt = S8 if self.pair8 is not None else S3         # what the pair kernel in use reads
        h = bufs[0].view(L["cout"], pair_fmt if (self.pair is not None and stop == 1) else S8)
        mark("start")
        if self.first is not None:
            hiplib.tdnn_first(x, R, self.first, L["bias"], L["scale"], L["shift"], self.act, L["alpha"], L["dil"], row_valid, h, status)
            mark("layer 0: tdnn_first_kernel (bf16x3)")
        else:
            # a first layer the dedicated kernel does not take: the general bf16x3 GEMM into fp32 rows, then one encoding pass
            y32 = self._l0_rows[:R]
            hiplib.tdnn_layer(x, L["wp"], L["bias"], L["scale"], L["shift"], self.act, L["alpha"], L["K"], L["dil"], row_valid, y32, rows=R)
            hiplib.split_encode(y32, h, rows=R, status=status if h.fmt == S8 else None)
            mark("layer 0: tdnn_gemm_bf16x3_kernel + split encode")
        for i in range(1, stop):
            L = self.layers[i]
            y = bufs[i & 1].view(L["cout"], pair_fmt if (self.pair is not None and i == stop - 1) else S8)
            hiplib.tdnn_layer8(h, R, L["wp8"], L["bias"], L["scale"], L["shift"], self.act, L["alpha"], L["dil"], row_valid, y, status)
            mark("layer %d: tdnn_gemm_f16bf8 (K=%d, %d -> %d)" % (i, L["K"], L["cin"], L["cout"]))
            h = y
        if self.pair is not None:
            La, Lb = self.layers[-2], self.layers[-1]
            if self.pair8 is not None:
                hiplib.tdnn_pair_pool8(h, R, self.pair8, (La["bias"], La["scale"], La["shift"], La["alpha"]),
                                       (Lb["bias"], Lb["scale"], Lb["shift"], Lb["alpha"]), self.act, row_valid, self._last, status)
            else:
                hiplib.tdnn_pair_pool(h, R, self.pair, (La["bias"], La["scale"], La["shift"], La["alpha"]),
                                      (Lb["bias"], Lb["scale"], Lb["shift"], Lb["alpha"]), self.act, row_valid, self._last)
        else:
            L = self.layers[-1]
            hiplib.tdnn_layer_pool8(h, R, L["wp8"], L["bias"], L["scale"], L["shift"], self.act, L["alpha"], L["dil"], row_valid,
                                    self._last)
        mark("layers %d+%d + pooling statistics: %s" % (n - 2, n - 1, "tdnn_pair_pool_f16bf8_kernel" if self.pair8 is not None else
                                                        "tdnn_pair_pool_kernel (bf16x3)") if self.pair is not None else
             "layer %d + pooling statistics: tdnn_gemm_f16bf8 POOL" % (n - 1))

    def _attention_scores(self, h, R, L, row_valid):
        """Last frame-level layer + attention scores (models.py:1022-1046) for one batch: leaves the per-row scores in
        ``self._scores`` and returns h2[R, A], the half of the layer output that is pooled."""
        A, T = self.att_dim, self.att
        u = self._u[:R]
        if self.precision == "bf16x3":
            L1, L2 = self.last_halves
            hiplib.tdnn_layer(h, L1["wp"], L1["bias"], L1["scale"], L1["shift"], self.act, L1["alpha"], L1["K"], L1["dil"],
                              row_valid, self._h1, rows=R)
            h2 = self._last[:R]
            hiplib.tdnn_layer(h, L2["wp"], L2["bias"], L2["scale"], L2["shift"], self.act, L2["alpha"], L2["K"], L2["dil"],
                              row_valid, h2, rows=R)
            h1 = self._h1
        else:
            y = self._last[:R]
            hiplib.tdnn_layer(h, L["wp"], L["bias"], L["scale"], L["shift"], self.act, L["alpha"], L["K"], L["dil"],
                              row_valid, y, rows=R)
            h1, h2 = y[:, :A], y[:, A:]
        hiplib.tdnn_layer(h1, T["wp"], T["bias"], None, None, tp.ACT_NONE, None, 1, 1, None, u, rows=R)
        hiplib.attention_scores(u, T["v"], self._scores, rows=R)
        return h2

    def segment_level(self, pooled, out):
        """Segment-level part for ANY number of chunks at once (run once per window so the GEMM has enough rows to
        fill the chip): pooled[N, 2*C_last] -> out[N, E]."""
        n = pooled.shape[0]
        E0 = self.embed[0]
        if self.embedding_index == 0:
            # embed_layer-0/scores IS the x-vector (local/tf/models.py:159,414): pre-activation output
            hiplib.fc(pooled, E0["wp"], E0["bias"], None, None, tp.ACT_NONE, None, None, out)
        else:
            if self._a0 is None or self._a0.shape[0] < n:
                self._a0 = self.torch.empty((n, E0["cout"]), dtype=self.torch.float32, device=self.device)
            a0 = self._a0[:n]
            hiplib.fc(pooled, E0["wp"], E0["bias"], E0["scale"], E0["shift"], self.act, E0["alpha"], a0, None)
            E1 = self.embed[1]
            hiplib.fc(a0, E1["wp"], E1["bias"], None, None, tp.ACT_NONE, None, None, out)
        return out

    def forward_packed(self, x, row_start, row_len, row_valid, nchunks, max_len, out):
        """frame_level + segment_level for one batch (tests / smoke; the extractor runs the FC per window)."""
        self.reserve(x.shape[0], nchunks, max_len)
        pooled = self.frame_level(x, row_start, row_len, row_valid, nchunks, max_len, self._pooled[:nchunks])
        return self.segment_level(pooled, out)

    def intermediates_packed(self, x, row_valid):
        """Debug/test helper: per-layer outputs as fp32 [R, Cout] for a packed batch (allocates)."""
        torch = self.torch
        R = x.shape[0]
        outs = []
        h = x
        for i, L in enumerate(self.layers):
            last = i == len(self.layers) - 1
            if self.precision == "bf16x3" and not last:
                y = hiplib.SplitBuf(R, L["cout"], self.device)
            else:
                y = torch.empty((R, L["cout"]), dtype=torch.float32, device=self.device)
            hiplib.tdnn_layer(h, L["wp"], L["bias"], L["scale"], L["shift"], self.act, L["alpha"], L["K"], L["dil"],
                              row_valid, y, rows=R)
            outs.append(hiplib.split_decode(y, R) if isinstance(y, hiplib.SplitBuf) else y)
            h = y
        return outs


# ------------------------------------------------------------------------------------------------
# which arithmetic may this checkpoint use?  (load-time accuracy probe)
# ------------------------------------------------------------------------------------------------
# The reference computes in IEEE fp32 (local/tf/models.py:54-76); f16bf8 and bf16x3 are faster arithmetics that reproduce it to
# ~1e-5 / ~5e-6 relative L2 on networks whose weights and BatchNorm statistics look like a trained TDNN's.  Nothing forces a
# checkpoint to look like that, so the arithmetic is chosen PER MODEL: a fixed synthetic batch (MFCC-like: AR(1) over the frames,
# decaying per-coefficient scale, mean-normalised; 8 chunks x 256 frames) goes through the loaded weights in the candidate
# arithmetic and in the next more exact one, and the candidate is kept only when their x-vectors agree to PROBE_LIMIT_*.
PROBE_LIMIT_F16BF8 = 2e-5        # f16bf8 kept when within this of bf16x3 (typical: 1e-5); the parity bar is 1e-4
PROBE_LIMIT_BF16X3 = 4e-5        # bf16x3 kept when within this of the exact-fp32 kernels (typical: 5e-6)
DATA_PROBE_LIMIT = 3e-5          # run-time check on real utterances (Extractor): per-chunk vectors, shortest chunks of a window
_PROBE_INPUT = {}


def probe_batch(feat_dim, in_dim, gap, align, chunks=8, frames=256, seed=20240917):
    """The fixed probe batch in kernel layout (host arrays): ``(x[R, in_dim], row_start, row_len, row_valid, max_len)``."""
    key = (feat_dim, in_dim, gap, align, chunks, frames, seed)
    if key not in _PROBE_INPUT:
        rng = np.random.default_rng(seed)
        rho = 0.92
        # AR(1) over the frames as one lower-triangular matrix product (no per-frame loop): x_t = sum_j rho^(t-j) e_j
        t = np.arange(frames)
        L = np.tril(rho ** np.maximum(t[:, None] - t[None, :], 0))
        scale = 12.0 / (1.0 + np.arange(feat_dim)) ** 0.7
        mats = []
        for _ in range(chunks):
            m = L @ (rng.standard_normal((frames, feat_dim)) * np.sqrt(1.0 - rho * rho))
            t0 = int(rng.integers(0, max(frames - 10, 1)))
            m[t0:t0 + 10] *= 4.0                     # a burst (non-speech event): what wakes up near-dead channels
            m *= scale
            mats.append((m - m.mean(axis=0, keepdims=True)).astype(np.float32))
        lay = BatchLayout([frames] * chunks, gap, align)
        x = np.zeros((lay.rows, in_dim), np.float32)
        lay.pack(mats, x)
        _PROBE_INPUT[key] = (x, lay.row_start, lay.row_len, lay.row_valid(), lay.max_len)
    return _PROBE_INPUT[key]


def _max_rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if not (np.isfinite(a).all() and np.isfinite(b).all()):
        return float("inf")                       # a NaN on either side must never read as agreement
    num = np.sqrt(((a - b) ** 2).sum(axis=1))
    den = np.sqrt((b ** 2).sum(axis=1))
    with np.errstate(invalid="ignore", divide="ignore"):
        r = np.where(den > 0, num / den, np.where(num > 0, np.inf, 0.0))
    return float(np.nan_to_num(r, nan=np.inf).max()) if len(r) else 0.0


def select_model(weights, topo, device="cuda:0", embedding_index=0, precision="f16bf8", probe=None):
    """``DeviceModel`` in the fastest arithmetic, not faster than ``precision``, that the accuracy probe admits for THESE weights:
    f16bf8 -> (its x-vectors of the probe batch differ from bf16x3's by more than PROBE_LIMIT_F16BF8, or are not finite, or left
    the fp16 range) -> bf16x3 -> (differs from the exact-fp32 kernels by more than PROBE_LIMIT_BF16X3) -> the exact rung.  The
    second step only runs when the first one failed: a model that passes as f16bf8 loads exactly as before plus two small forwards.
    The exact rung is ``fp32tc`` (exact fp32 products, the K = 5 / 7 layers as Toom-Cook F(2, K): more accurate than the direct form
    on all 22 checkpoints of profiles/r05_fp32tc_accuracy_sweep.txt and 1.3x faster; bit-identical to ``fp32`` on a topology the
    Toom-Cook kernel does not cover) -- a demoted model does not fall to the slowest form.  XVECTOR_EXACT_RUNG=fp32 (or a request
    of ``precision="fp32"``) keeps the direct contraction, the literal statement of local/tf/models.py:60.
    ``model.selection`` reports what was measured.  ``probe=False`` (or XVECTOR_ACCURACY_PROBE=0) takes ``precision`` as given."""
    if probe is None:
        probe = os.environ.get("XVECTOR_ACCURACY_PROBE", "1") != "0"
    model = DeviceModel(weights, topo, device, embedding_index, precision)
    sel = dict(requested=precision, selected=model.arithmetic, probed=False)
    model.selection = sel
    if not probe or not model.f16bf8:
        return model
    batch = probe_batch(model.feat_dim, model.in_dim, model.gap, model.align)
    got = model.probe_vectors(batch)
    clamped = int(model.status.item()) != 0
    model.status.zero_()
    twin = model.fallback()
    ref = twin.probe_vectors(batch)
    err = _max_rel_l2(got, ref)
    sel.update(probed=True, f16bf8_vs_bf16x3=err, f16bf8_limit=PROBE_LIMIT_F16BF8, probe_left_fp16_range=clamped,
               probe_frames=int(np.sum(batch[2])))
    if err <= PROBE_LIMIT_F16BF8 and not clamped:
        return model
    rung = "fp32" if os.environ.get("XVECTOR_EXACT_RUNG", "fp32tc") == "fp32" else "fp32tc"
    exact = DeviceModel(weights, topo, device, embedding_index, rung)
    err3 = _max_rel_l2(ref, exact.probe_vectors(batch))
    sel.update(bf16x3_vs_fp32=err3, bf16x3_limit=PROBE_LIMIT_BF16X3, exact_rung=exact.arithmetic)
    chosen = twin if err3 <= PROBE_LIMIT_BF16X3 else exact
    sel["selected"] = chosen.arithmetic
    chosen.selection = sel
    return chosen


class _Rows(object):
    """Utterances ``idx`` of a window: a sequence with the ``lengths`` / ``addrs`` arrays the extractor plans and packs from;
    a matrix is only materialised when somebody indexes it."""

    def __init__(self, mats, idx, lens, addrs):
        self.mats, self.idx = mats, np.asarray(idx, np.int64)
        self.lengths = np.asarray(lens, np.int64)[self.idx]
        self.addrs = None if addrs is None else np.asarray(addrs, np.uint64)[self.idx]

    def __len__(self):
        return len(self.idx)

    def __getitem__(self, i):
        return self.mats[int(self.idx[i])]

    def __iter__(self):
        for i in self.idx.tolist():
            yield self.mats[i]


# ------------------------------------------------------------------------------------------------
# extractor: utterances in, x-vectors out
# ------------------------------------------------------------------------------------------------
_STAGE_CACHE = {}          # (in_dim, NBUF) -> parked pinned staging sets of finished extractors (at most 2 per key)
_STAGE_LOCK = __import__("threading").Lock()


_STAGE_PENDING = {}        # (in_dim, NBUF) -> side thread that is pinning a set for the cache (prewarm_staging)


def _new_stage(torch, in_dim, rows, nchunks, nbuf):
    return [dict(x=torch.zeros((rows, in_dim), dtype=torch.float32).pin_memory(), rv=torch.zeros(rows, dtype=torch.uint8).pin_memory(),
                 meta=torch.zeros((2, nchunks), dtype=torch.int32).pin_memory(), event=None) for _ in range(nbuf)]


def prewarm_staging(in_dim, rows=262144, nchunks=8192):
    """Pin the staging sets of a full-size extractor on a side thread and park them where ``Extractor._staging`` looks first.
    Page-pinning is a host-wide serial resource: 400 MB take 0.27 s in a lone process and 3.7-4.4 s in EACH of eight that start
    together (tools/experiments/startup_contention_probe.py) -- the ranks of a node's job do start together.  Called by the CLI
    worker as soon as the HIP runtime is up, the ~100 MB of a worker are pinned while its weights are read, packed and probed
    instead of in front of its first window."""
    import threading
    import torch
    key = (int(in_dim), Extractor.NBUF)
    with _STAGE_LOCK:
        if key in _STAGE_PENDING or any(st[0]["x"].shape[0] >= rows for st in _STAGE_CACHE.get(key, [])):
            return

        def run():
            try:
                stage = _new_stage(torch, key[0], rows, nchunks, key[1])
                with _STAGE_LOCK:
                    _STAGE_CACHE.setdefault(key, []).append(stage)
            except Exception:          # the extractor pins its own set then, and reports what goes wrong
                pass
        th = threading.Thread(target=run, name="xv-pin-staging", daemon=True)
        _STAGE_PENDING[key] = th
    th.start()


def _park_stage(holder):
    if len(holder) == 2 and holder[1] is not None:
        key, stage = holder
        for st in stage:
            if st.get("event") is not None:
                st["event"].synchronize()
                st["event"] = None
        with _STAGE_LOCK:
            parked = _STAGE_CACHE.setdefault(key, [])
            if len(parked) < 2 and all(p is not stage for p in parked):
                parked.append(stage)
        holder[:] = []


class Extractor(object):
    """Batches the chunk plans of many utterances (length-bucketed), runs them and averages per
    utterance.  Output order == input order, rejected utterances yield ``None`` vectors.

    Host packing and the H2D copies of batch i+1 overlap the kernels of batch i: ``NBUF`` pinned staging sets are
    rotated, copies go through a dedicated HIP stream, and the compute stream waits on the copy's event."""

    NBUF = 3
    PROBE_ROWS = 16384           # run-time accuracy check: this many rows of a window's first batch are repeated on the bf16x3 twin
    PROBE_EVERY = 64             # ... for the first window of an extractor and every 64th after it

    def __init__(self, model, min_chunk_size, chunk_size, max_batch_rows=262144, max_batch_chunks=8192, accuracy_probe=None):
        """accuracy_probe (f16bf8 models; default on, XVECTOR_ACCURACY_PROBE=0 turns it off): the leading (= shortest) chunks of
        the first batch of a window -- the packed rows are already in HBM -- also run through the bf16x3 twin, and ``finish``
        compares the two sets of chunk vectors.  Beyond DATA_PROBE_LIMIT the extractor is DEMOTED: the window is repeated on
        the twin and every later window goes there directly.  The load-time probe (``select_model``) sees the weights on
        synthetic input; this one sees them on the caller's features."""
        self.model = model
        if accuracy_probe is None:
            accuracy_probe = os.environ.get("XVECTOR_ACCURACY_PROBE", "1") != "0"
        self.accuracy_probe = bool(accuracy_probe) and model.f16bf8
        self.demoted = False
        self._windows = 0
        self.min_chunk_size = int(min_chunk_size)
        self.chunk_size = int(chunk_size)
        self.max_batch_rows = int(max_batch_rows)
        self.max_batch_chunks = int(max_batch_chunks)
        self.stats = dict(batches=0, chunks=0, frames=0, rows=0)
        self._stage = None
        self._copy_stream = None
        self._turn = 0
        self._finalizer = None
        self._holder = []                              # [cache key, staging sets]: what the finalizer parks
        self._pin_free = {}
        self._fallback_ex = None                       # bf16x3 twin of an f16bf8 extractor (out-of-range windows)

    def _staging(self, rows, nchunks):
        """NBUF pinned sets: features [rows, in_dim] (padding columns zeroed once), row_valid[rows], meta int32[2, chunks].
        Pinning 3 x 25 MB costs tens of milliseconds, so the sets of a finished extractor are parked process-wide
        (``_STAGE_CACHE``) and the next one -- every ``make_embedding`` call builds its own -- picks them up."""
        torch = self.model.torch
        if self._stage is None or self._stage[0]["x"].shape[0] < rows or self._stage[0]["meta"].shape[1] < nchunks:
            # sized for the largest regular batch up front: re-pinning 3 x 25 MB whenever a window brings a slightly larger
            # batch costs more than the kernels of that batch
            rows = max(rows, 1024) if rows <= 4096 else max(rows, self.max_batch_rows)
            nchunks = max(nchunks, 64) if nchunks <= 64 else max(nchunks, min(self.max_batch_chunks, 8192))
            if self._stage is not None:
                _park_stage(self._holder)
            with _STAGE_LOCK:
                pending = _STAGE_PENDING.pop((self.model.in_dim, self.NBUF), None)
            if pending is not None:
                pending.join()                                      # (prewarm_staging: a set is being pinned for us right now)
            with _STAGE_LOCK:
                parked = _STAGE_CACHE.get((self.model.in_dim, self.NBUF), [])
                hit = next((i for i, st in enumerate(parked) if st[0]["x"].shape[0] >= rows and st[0]["meta"].shape[1] >= nchunks), None)
                self._stage = parked.pop(hit) if hit is not None else None
            if self._stage is None:
                self._stage = _new_stage(torch, self.model.in_dim, rows, nchunks, self.NBUF)
            if self._finalizer is None:
                import weakref
                self._finalizer = weakref.finalize(self, _park_stage, self._holder)
            self._holder[:] = [(self.model.in_dim, self.NBUF), self._stage]
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.model.device)
        return self._stage

    def _reserve_caps(self, bounds):
        """Rows / chunks to size the device buffers for: the largest REGULAR batch from the first window on (as the pinned
        staging sets above) -- growing them when a later window brings a larger batch is a hipMalloc of gigabytes in the
        middle of the stream (50 ms per call when the first read window was a short one)."""
        rows = max(r for _, _, r in bounds)
        nch = max(b1 - b0 for b0, b1, _ in bounds)
        rows = rows if rows <= 4096 else max(rows, self.max_batch_rows)
        nch = nch if nch <= 64 else max(nch, min(self.max_batch_chunks, 8192))
        return rows, nch

    def _pinned(self, kind, shape, dtype):
        """A pinned host tensor of ``shape`` from this extractor's free list (per-window buffers: the chunk table that goes up,
        the x-vectors that come down): one hipHostMalloc per window is a millisecond the pipeline has no use for."""
        torch = self.model.torch
        need = int(np.prod(shape))
        free = self._pin_free.setdefault((kind, dtype), [])
        best = next((i for i, t in enumerate(free) if t.numel() >= need), None)
        if best is None:
            # torch's pinned allocator hands out power-of-two blocks: ask for the whole block the request lands in (the slack that lets
            # the next, slightly larger window reuse it) and not for 5/4 of it, which doubled every block just above a power of two --
            # and pinning is what N workers starting together queue for (prewarm_staging)
            size = torch.empty(0, dtype=dtype).element_size()
            want = 1 << max(int(need * size - 1).bit_length(), 9)
            flat = torch.empty(want // size, dtype=dtype).pin_memory()
        else:
            flat = free.pop(best)
        return flat, flat[:need].view(shape)

    def _unpin(self, kind, flat):
        free = self._pin_free.setdefault((kind, flat.dtype), [])
        if len(free) < 4:
            free.append(flat)

    PACK_THREADS = int(os.environ.get("XVECTOR_PACK_THREADS", "4"))
    ROUND_ROWS = 32768           # rows one round of resident workgroups of the GEMM kernels covers on the 256 CUs (_batch_bounds)

    def _batch_bounds(self, cum, lead):
        """Chunk ranges of the batches of a window: ``cum[j]`` = rows the chunks before j occupy.  The window's rows are dealt
        EVENLY to the fewest batches that respect ``max_batch_rows`` (3.2 batches' worth of rows become 4 batches of 0.8, not
        3 + a 0.2 remnant that fills a fraction of the chip); a single chunk may still exceed the row budget."""
        nch = len(cum) - 1
        budget = max(self.max_batch_rows - lead, 1)
        total = int(cum[-1])
        # Large windows: batch sizes in whole ROUNDS of workgroups.  The wide GEMM (256 rows x 128 of the 256 CUs per column
        # tile) and the pair kernel (128 rows x 256 CUs) both finish 32768 rows per round of resident workgroups, and a partly
        # filled round costs a whole one: 2.67 batches' worth of rows as 3 even batches are 3 x 7.1 -> 24 rounds, as 8 + 7 + 7
        # rounds they are 22.  The rounds the window needs are dealt to the fewest batches as evenly as possible.
        R = self.ROUND_ROWS
        cap = (lead + budget) // R                                   # rounds a batch may hold (8 at 262144 rows)
        if cap >= 2 and total > budget:
            n_b = max(1, -(-(total + lead) // (cap * R - lead)))
            rounds = -(-(total + n_b * lead) // R)
            n_b = max(n_b, -(-rounds // cap))
            per = [rounds // n_b + (1 if k < rounds % n_b else 0) for k in range(n_b)]
            cuts, b0 = [0], 0
            for k in range(n_b):
                room = min(per[k], cap) * R - lead
                b1 = int(np.searchsorted(cum, cum[b0] + room, side="right")) - 1
                b1 = min(max(b1, b0 + 1), nch)
                cuts.append(b1)
                b0 = b1
                if b0 >= nch:
                    break
            while b0 < nch:                                          # (chunks do not split: what the dealt rounds could not take)
                b1 = int(np.searchsorted(cum, cum[b0] + cap * R - lead, side="right")) - 1
                b0 = min(max(b1, b0 + 1), nch)
                cuts.append(b0)
            if all(int(cum[b1] - cum[b0]) <= budget or b1 - b0 == 1 for b0, b1 in zip(cuts[:-1], cuts[1:])):
                return self._split_by_chunks(cum, cuts, lead)
        n_b = max(1, -(-total // budget))
        cuts = None
        for extra in (0, 1):
            # batch k ends at the first chunk boundary at or past k/n_b of the rows
            c = sorted(set([0] + [int(np.searchsorted(cum, total * k / float(n_b + extra), side="left")) for k in range(1, n_b + extra)] + [nch]))
            if all(int(cum[b1] - cum[b0]) <= budget or b1 - b0 == 1 for b0, b1 in zip(c[:-1], c[1:])):
                cuts = c
                break
        if cuts is None:
            # chunks that are large next to the row budget (tests, tiny budgets): greedy fill
            cuts, b0 = [0], 0
            while b0 < nch:
                b1 = int(np.searchsorted(cum, cum[b0] + budget, side="right")) - 1
                b0 = min(max(b1, b0 + 1), nch)
                cuts.append(b0)
        return self._split_by_chunks(cum, cuts, lead)

    def _split_by_chunks(self, cum, cuts, lead):
        bounds = []
        for b0, b1 in zip(cuts[:-1], cuts[1:]):
            while b1 - b0 > self.max_batch_chunks:   # (only with thousands of very short chunks)
                bounds.append((b0, b0 + self.max_batch_chunks, lead + int(cum[b0 + self.max_batch_chunks] - cum[b0])))
                b0 += self.max_batch_chunks
            bounds.append((b0, b1, lead + int(cum[b1] - cum[b0])))
        return bounds

    def extract(self, mats, addrs=None):
        """mats: list of float32 [T, F] arrays.  Returns a list of float32[E] (or None) per input."""
        return self.finish(self.submit(mats, addrs))

    def submit(self, mats, addrs=None):
        """Plan, pack, copy and launch everything for ``mats`` and start the asynchronous D2H copy of the x-vectors; returns a
        handle for ``finish``.  The caller may submit the next window before finishing this one (the host work of window
        i+1 then overlaps the kernels of window i).  ``addrs``: optional ``matrix_addresses(mats, F)`` computed elsewhere
        (e.g. by the reader thread).  ``mats`` may be a lazy sequence with a ``lengths`` array (kaldi_io.ArkMats): with ``addrs``
        the native packer reads the rows in place and no matrix object is ever built."""
        if self.demoted:
            return self._on_twin(lambda ex: ex.submit(mats, addrs))
        torch = self.model.torch
        model = self.model
        dev = model.device
        lens = mats.lengths if hasattr(mats, "lengths") else [m.shape[0] for m in mats]
        # chunk table, utterances ordered by length so that batches are length-homogeneous
        order, c_utt, c_start, c_len, seg_start = plan_chunk_table(lens, self.min_chunk_size, self.chunk_size)
        nch = len(c_utt)
        handle = dict(n=len(lens), order=order, nch=nch)
        if nch == 0:
            return handle
        F = model.feat_dim
        lib = _host_lib()
        if lib is not None and addrs is None:
            addrs = matrix_addresses(mats, F)
        native = lib is not None and addrs is not None
        if native:
            c_src = np.asarray(addrs, dtype=np.uint64)[c_utt] + (c_start * (F * 4)).astype(np.uint64)
            c_len32 = c_len.astype(np.int32)
        else:
            assert mats[int(c_utt[0])].shape[1] == F, "feature dimension does not match the model"
            l_utt, l_start, l_len = c_utt.tolist(), c_start.tolist(), c_len.tolist()
        gap, align = model.gap, model.align
        lead = (gap + align - 1) // align * align
        # batch boundaries: greedy fill up to max_batch_rows / max_batch_chunks (a single chunk may exceed the row budget)
        cum = np.zeros(nch + 1, dtype=np.int64)
        np.cumsum(slot_rows(c_len, gap, align), out=cum[1:])
        bounds = self._batch_bounds(cum, lead)
        stage = self._staging(max(r for _, _, r in bounds), max(b1 - b0 for b0, b1, _ in bounds))
        with torch.cuda.device(dev):
            compute = torch.cuda.current_stream()
            E_all = torch.empty((nch, model.embed_dim), dtype=torch.float32, device=dev)
            P_all = torch.empty((nch, model.pooled_dim), dtype=torch.float32, device=dev)
            model.reserve(*self._reserve_caps(bounds), int(c_len.max()))
            keep = []                                   # device inputs stay referenced until the window is done
            status = torch.zeros(1, dtype=torch.int32, device=dev) if model.f16bf8 else None
            for bi, (b0, b1, _) in enumerate(bounds):
                layout = BatchLayout(c_len[b0:b1], gap, align)
                st = stage[self._turn % self.NBUF]
                self._turn += 1
                if st["event"] is not None:
                    st["event"].synchronize()            # the copy that last read this pinned set has finished
                if native:
                    xs = st["x"].numpy()
                    rc = lib.xv_pack_rows_f32(c_src[b0:b1].ctypes.data, c_len32[b0:b1].ctypes.data, layout.row_start.ctypes.data,
                                              layout.nchunks, F, xs.ctypes.data, xs.shape[1], layout.rows,
                                              st["rv"].numpy().ctypes.data, self.PACK_THREADS)
                    assert rc == 0, "xv_pack_rows_f32 rejected the batch layout"
                else:
                    layout.pack([mats[l_utt[i]][l_start[i]:l_start[i] + l_len[i]] for i in range(b0, b1)], st["x"].numpy())
                    layout.row_valid(st["rv"].numpy())
                meta = st["meta"].numpy()
                meta[0, :layout.nchunks] = layout.row_start
                meta[1, :layout.nchunks] = layout.row_len
                with torch.cuda.stream(self._copy_stream):
                    x = st["x"][:layout.rows].to(dev, non_blocking=True)
                    rv = st["rv"][:layout.rows].to(dev, non_blocking=True)
                    md = st["meta"][:, :layout.nchunks].to(dev, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self._copy_stream)
                st["event"] = ev
                compute.wait_event(ev)
                model.frame_level(x, md[0], md[1], rv, layout.nchunks, layout.max_len, P_all[b0:b1], status=status)
                keep.append((x, rv, md))
                if bi == 0:
                    self._start_probe(handle, x, md, rv, layout)
                self.stats["batches"] += 1
                self.stats["chunks"] += layout.nchunks
                self.stats["frames"] += int(layout.row_len.sum())
                self.stats["rows"] += layout.rows
            model.segment_level(P_all, E_all)
            # (pinned + copy stream: a pageable H2D copy here would block the host until every kernel of the window is done)
            tail_np = np.concatenate([np.asarray(seg_start, dtype=np.int32), c_len.astype(np.int32)])
            tail_flat, tail = self._pinned("tail", tail_np.shape, torch.int32)
            tail.numpy()[:] = tail_np
            with torch.cuda.stream(self._copy_stream):
                tail_d = tail.to(dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._copy_stream)
            compute.wait_event(ev)
            seg, cl = tail_d[:len(order) + 1], tail_d[len(order) + 1:]
            out = torch.empty((len(order), model.embed_dim), dtype=torch.float32, device=dev)
            hiplib.chunk_average(E_all, seg, cl, len(order), out)
            host_flat, host = self._pinned("xvec", (len(order), model.embed_dim), torch.float32)
            host.copy_(out, non_blocking=True)
            self._end_probe(handle, E_all)
            self._watch(handle, status, lambda ex: ex.submit(mats, addrs))
            done = torch.cuda.Event()
            done.record(compute)
        # mats (the native packer read them in place) and the device buffers stay referenced until finish()
        handle.update(host=host, done=done, keep=(keep, E_all, P_all, tail, tail_d, out, mats), pinned=(tail_flat, host_flat))
        return handle

    def _watch(self, handle, status, redo):
        """f16bf8 models: bring the window's out-of-range flag down with its x-vectors; ``finish`` repeats the window on the
        bf16x3 twin of the model (``redo``) when it is set."""
        if status is None:
            return
        flat, flag = self._pinned("flag", (1,), self.model.torch.int32)
        flag.copy_(status, non_blocking=True)
        handle.update(flag=flag, flag_flat=flat, redo=redo, status=status)

    def _twin(self):
        """The extractor of the model's bf16x3 twin (out-of-range windows, windows after a demotion)."""
        if self._fallback_ex is None:
            self._fallback_ex = Extractor(self.model.fallback(), self.min_chunk_size, self.chunk_size, self.max_batch_rows,
                                          self.max_batch_chunks)
        return self._fallback_ex

    def _on_twin(self, submit):
        """A window of a demoted extractor: submitted on the twin (``finish`` follows the handle's ``owner``); its batches
        count in this extractor's statistics."""
        ex = self._twin()
        before = dict(ex.stats)
        handle = submit(ex)
        for k in ("batches", "chunks", "frames", "rows"):
            self.stats[k] += ex.stats[k] - before[k]
        handle["owner"] = ex
        return handle

    PROBE_MIN_LEN = 128          # chunks the probe prefers (a vector pooled over a handful of frames is noisier in EVERY arithmetic)

    def _start_probe(self, handle, x, md, rv, layout):
        """Run-time accuracy check, first half (called right behind the first batch of a window): a run of consecutive chunks of
        the batch -- the shortest ones of at least PROBE_MIN_LEN frames (all shorter: the longest ones), at most PROBE_ROWS rows
        -- through the bf16x3 twin.  The rows are already in HBM: a run of slots of a packed batch, taken from ``lead`` rows
        before its first chunk, is itself a packed batch once the chunk starts are re-based (the rows in front of it are the
        previous chunk's tail: never read by this run's frames, whose halo ends in that chunk's zero gap rows)."""
        self._windows += 1
        if not self.accuracy_probe or (self._windows - 1) % self.PROBE_EVERY:
            return
        n = layout.nchunks
        i0 = int(np.searchsorted(layout.row_len, self.PROBE_MIN_LEN, side="left"))        # (chunks of a batch ascend in length)
        if i0 >= n:
            i0 = max(0, n - 16)
        cum = np.cumsum(layout.slots[i0:])
        m = int(np.searchsorted(cum, self.PROBE_ROWS - layout.lead, side="right"))
        if m == 0:
            if int(cum[0]) > 4 * self.PROBE_ROWS:
                self._windows -= 1                   # a very long chunk: look at the next window instead
                return
            m = 1
        base = int(layout.row_start[i0]) - layout.lead
        rows = layout.lead + int(cum[m - 1])
        torch, twin = self.model.torch, self.model.fallback()
        dev = twin.device
        flat, meta = self._pinned("probe_meta", (2, m), torch.int32)
        mh = meta.numpy()
        mh[0] = layout.row_start[i0:i0 + m] - base
        mh[1] = layout.row_len[i0:i0 + m]
        with torch.cuda.stream(self._copy_stream):
            md2 = meta.to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        torch.cuda.current_stream().wait_event(ev)
        P = torch.empty((m, twin.pooled_dim), dtype=torch.float32, device=dev)
        E = torch.empty((m, twin.embed_dim), dtype=torch.float32, device=dev)
        twin.frame_level(x[base:base + rows], md2[0], md2[1], rv[base:base + rows], m, int(layout.row_len[i0:i0 + m].max()), P)
        twin.segment_level(P, E)
        handle["probe"] = dict(i0=i0, m=m, ref=E, keep=(P, md2, meta), meta_flat=flat)

    def _end_probe(self, handle, E_all):
        """Second half (behind the window's segment-level launch): both sets of chunk vectors start their way to the host."""
        pr = handle.get("probe")
        if pr is None:
            return
        i0, m = pr["i0"], pr["m"]
        flat, host = self._pinned("probe", (2, m, self.model.embed_dim), self.model.torch.float32)
        host[0].copy_(E_all[i0:i0 + m], non_blocking=True)
        host[1].copy_(pr["ref"], non_blocking=True)
        pr.update(host=host, flat=flat)

    def _probe_failed(self, handle):
        """finish(): True when the window's probe found the f16bf8 vectors too far from the twin's -> demotion."""
        pr = handle.pop("probe", None)
        if pr is None or "host" not in pr:
            return False
        got = pr["host"].numpy()
        err = _max_rel_l2(got[0], got[1])
        self._unpin("probe", pr["flat"])
        self._unpin("probe_meta", pr["meta_flat"])
        self.stats["probe_windows"] = self.stats.get("probe_windows", 0) + 1
        self.stats["probe_rel_l2_max"] = max(self.stats.get("probe_rel_l2_max", 0.0), err)
        return err > DATA_PROBE_LIMIT

    def submit_raw(self, mats, vads, cmn_window, center=True, min_window=100, addrs=None):
        """``submit`` for RAW features: sliding-window CMN + VAD frame selection (xv_cmn_sliding_scatter_f32) run on the
        device and write every voiced, normalised frame straight into its row of the packed batch -- the selected features
        never exist on the host.  mats: float32 [T, F] arrays; vads: None or one 1-D array (non-zero = voiced) / None per
        utterance; addrs: optional ``matrix_addresses(mats, F)``.  Returns ``(handle, lengths, vad_dropped)``: the handle for ``finish``; the number of selected frames per
        utterance; a bool mask of the utterances select-voiced-frames drops (VAD length mismatch / no voiced frame)."""
        from .frontend import VadRuns, select_voiced
        if self.demoted:
            box = []

            def go(ex):
                box[:] = ex.submit_raw(mats, vads, cmn_window, center, min_window, addrs)
                return box[0]
            return (self._on_twin(go),) + tuple(box[1:])
        torch = self.model.torch
        model = self.model
        dev = model.device
        n = len(mats)
        F = model.feat_dim
        lib = _host_lib()
        if addrs is None and lib is not None:
            addrs = matrix_addresses(mats, F)
        assert lib is not None and addrs is not None, "submit_raw needs libxvector_host.so and C-contiguous float32 [T, %d] matrices" % F
        T, cand, voiced, _, _ = select_voiced(mats, vads)
        V = np.zeros(n, dtype=np.int64)                       # selected frames per utterance
        vstart = np.zeros(n, dtype=np.int64)                  # offset of the utterance's flags in `voiced`
        if voiced is None:
            V[cand] = T[cand]
        else:
            cs = np.zeros(len(cand), dtype=np.int64)
            np.cumsum(T[cand][:-1], out=cs[1:])
            vstart[cand] = cs
            V[cand] = np.add.reduceat(voiced, cs) if len(cand) else 0
        vad_dropped = np.ones(n, dtype=bool)
        vad_dropped[cand] = False
        if vads is None:
            vad_dropped[:] = False
        else:
            if not isinstance(vads, VadRuns):             # (runs: every utterance has a vector)
                vad_dropped &= np.fromiter((v is not None for v in vads), dtype=bool, count=n)
        order, c_utt, c_start, c_len, seg_start = plan_chunk_table(V, self.min_chunk_size, self.chunk_size)
        nch = len(c_utt)
        handle = dict(n=n, order=order, nch=nch)
        if nch == 0:
            return handle, V, vad_dropped
        gap, align = model.gap, model.align
        lead = (gap + align - 1) // align * align
        cum = np.zeros(nch + 1, dtype=np.int64)
        np.cumsum(slot_rows(c_len, gap, align), out=cum[1:])
        bounds = self._batch_bounds(cum, lead)
        kept = np.ascontiguousarray(np.diff(seg_start), dtype=np.int64)                  # chunks per utterance of `order`
        first_len = np.ascontiguousarray(c_len[seg_start[:-1]], dtype=np.int64)          # = the chunk size the plan used for the utterance
        seg_o = np.ascontiguousarray(seg_start[:-1], dtype=np.int64)
        vstart_o = np.ascontiguousarray(vstart[order], dtype=np.int64)                   # (per utterance of `order`, like the three above)
        if voiced is not None:
            voiced = np.ascontiguousarray(voiced, dtype=np.bool_)
        # utterances (positions in `order`) touched by each batch, and the raw rows they bring
        spans = [(int(np.searchsorted(seg_start, b0, side="right")) - 1, int(np.searchsorted(seg_start, b1, side="left")))
                 for b0, b1, _ in bounds]
        raw_rows = max(int(T[order[lo:hi]].sum()) for lo, hi in spans)
        self._raw_staging(raw_rows, max(hi - lo for lo, hi in spans), F)
        self._staging(max(r for _, _, r in bounds), max(b1 - b0 for b0, b1, _ in bounds))
        with torch.cuda.device(dev):
            compute = torch.cuda.current_stream()
            E_all = torch.empty((nch, model.embed_dim), dtype=torch.float32, device=dev)
            P_all = torch.empty((nch, model.pooled_dim), dtype=torch.float32, device=dev)
            model.reserve(*self._reserve_caps(bounds), int(c_len.max()))
            keep = []
            status = torch.zeros(1, dtype=torch.int32, device=dev) if model.f16bf8 else None
            for bi, ((b0, b1, _), (lo, hi)) in enumerate(zip(bounds, spans)):
                layout = BatchLayout(c_len[b0:b1], gap, align)
                U = order[lo:hi]
                Tb = T[U]
                nU, rows_in = len(U), int(Tb.sum())
                ustart = np.zeros(nU, dtype=np.int64)
                np.cumsum(Tb[:-1], out=ustart[1:])
                st = self._raw_stage[self._turn % self.NBUF]
                sx = self._stage[self._turn % self.NBUF]
                self._turn += 1
                for s_ in (st, sx):
                    if s_["event"] is not None:
                        s_["event"].synchronize()
                src_b, len_b, start_b = addrs[U], Tb.astype(np.int32), ustart.astype(np.int32)      # (kept alive across the call)
                rc = lib.xv_pack_rows_f32(src_b.ctypes.data, len_b.ctypes.data, start_b.ctypes.data, nU, F,
                                          st["raw"].numpy().ctypes.data, F, rows_in, None, self.PACK_THREADS)
                assert rc == 0, "xv_pack_rows_f32 rejected the raw layout"
                # destination row of every raw frame of the batch's utterances (-1: unvoiced / chunk of another batch / dropped tail),
                # straight into the pinned array (native: one pass over the frames)
                rc = lib.xv_raw_row_plan(nU, Tb.ctypes.data, voiced.ctypes.data if voiced is not None else None, vstart_o[lo:hi].ctypes.data,
                                         first_len[lo:hi].ctypes.data, kept[lo:hi].ctypes.data, seg_o[lo:hi].ctypes.data, int(b0), int(b1),
                                         layout.row_start.ctypes.data, st["dst"].numpy().ctypes.data, rows_in)
                assert rc == 0, "xv_raw_row_plan rejected the raw layout"
                um = st["utt"].numpy()
                um[0, :nU] = ustart
                um[1, :nU] = Tb
                layout.row_valid(sx["rv"].numpy())
                meta = sx["meta"].numpy()
                meta[0, :layout.nchunks] = layout.row_start
                meta[1, :layout.nchunks] = layout.row_len
                with torch.cuda.stream(self._copy_stream):
                    raw_d = st["raw"][:rows_in].to(dev, non_blocking=True)
                    dst_d = st["dst"][:rows_in].to(dev, non_blocking=True)
                    utt_d = st["utt"][:, :nU].to(dev, non_blocking=True)
                    rv = sx["rv"][:layout.rows].to(dev, non_blocking=True)
                    md = sx["meta"][:, :layout.nchunks].to(dev, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self._copy_stream)
                st["event"] = sx["event"] = ev
                compute.wait_event(ev)
                x = torch.zeros((layout.rows, model.in_dim), dtype=torch.float32, device=dev)      # gap rows and padding columns
                hiplib.cmn_sliding_scatter(raw_d, utt_d[0], utt_d[1], nU, int(Tb.max()), cmn_window, center, min_window, dst_d, x)
                model.frame_level(x, md[0], md[1], rv, layout.nchunks, layout.max_len, P_all[b0:b1], status=status)
                keep.append((x, rv, md, raw_d, dst_d, utt_d))
                if bi == 0:
                    self._start_probe(handle, x, md, rv, layout)
                self.stats["batches"] += 1
                self.stats["chunks"] += layout.nchunks
                self.stats["frames"] += int(layout.row_len.sum())
                self.stats["rows"] += layout.rows
            model.segment_level(P_all, E_all)
            tail_np = np.concatenate([np.asarray(seg_start, dtype=np.int32), c_len.astype(np.int32)])
            tail_flat, tail = self._pinned("tail", tail_np.shape, torch.int32)
            tail.numpy()[:] = tail_np
            with torch.cuda.stream(self._copy_stream):
                tail_d = tail.to(dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._copy_stream)
            compute.wait_event(ev)
            seg, cl = tail_d[:len(order) + 1], tail_d[len(order) + 1:]
            out = torch.empty((len(order), model.embed_dim), dtype=torch.float32, device=dev)
            hiplib.chunk_average(E_all, seg, cl, len(order), out)
            host_flat, host = self._pinned("xvec", (len(order), model.embed_dim), torch.float32)
            host.copy_(out, non_blocking=True)
            self._end_probe(handle, E_all)
            self._watch(handle, status, lambda ex: ex.submit_raw(mats, vads, cmn_window, center, min_window, addrs)[0])
            done = torch.cuda.Event()
            done.record(compute)
        handle.update(host=host, done=done, keep=(keep, E_all, P_all, tail, tail_d, out, mats), pinned=(tail_flat, host_flat))
        return handle, V, vad_dropped

    def _raw_staging(self, rows, nutts, feat_dim):
        """NBUF pinned sets for submit_raw: raw features [rows, F], destination rows int32[rows], (start, length) int32[2, utts]."""
        torch = self.model.torch
        cur = getattr(self, "_raw_stage", None)
        if cur is None or cur[0]["raw"].shape[0] < rows or cur[0]["utt"].shape[1] < nutts or cur[0]["raw"].shape[1] != feat_dim:
            rows = max(rows, 2 * self.max_batch_rows) if rows > 4096 else max(rows, 1024)
            nutts = max(nutts, min(self.max_batch_chunks, 8192)) if nutts > 64 else max(nutts, 64)
            self._raw_stage = [dict(raw=torch.zeros((rows, feat_dim), dtype=torch.float32).pin_memory(),
                                    dst=torch.zeros(rows, dtype=torch.int32).pin_memory(),
                                    utt=torch.zeros((2, nutts), dtype=torch.int32).pin_memory(), event=None) for _ in range(self.NBUF)]
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.model.device)
        return self._raw_stage

    def finish(self, handle, as_array=False):
        """Wait for a submitted window and return its x-vectors in input order: a list with None for rejected utterances,
        or with ``as_array`` the pair (float32 [n, E] array, bool [n] mask of the utterances that produced a vector)."""
        owner = handle.get("owner")
        if owner is not None and owner is not self:
            return owner.finish(handle, as_array)             # a window of a demoted extractor: it ran on the twin
        n = handle["n"]
        if handle["nch"]:
            handle["done"].synchronize()
            if handle.get("flag") is not None:
                clamped = int(handle["flag"][0]) != 0
                self._unpin("flag", handle.pop("flag_flat"))
                handle["flag"] = None
                if self._probe_failed(handle) and not self.demoted:
                    self.demoted = True
                    self.stats["demoted_at_window"] = self._windows
                if clamped or self.demoted:
                    # some activation of the window left the fp16 range of the f16bf8 arithmetic, or an accuracy probe failed
                    # (this window's, or an earlier one's while this window was already in flight): the whole window again on
                    # the bf16x3 twin -- its results replace these
                    redo = handle.pop("redo")
                    pinned = handle.pop("pinned", None)
                    if pinned is not None:
                        self._unpin("tail", pinned[0])
                        self._unpin("xvec", pinned[1])
                    handle["keep"] = None
                    self.stats["fallback_windows"] = self.stats.get("fallback_windows", 0) + 1
                    return self._twin().finish(redo(self._twin()), as_array)
                handle.pop("redo", None)
            host_out = handle["host"].numpy()
            handle["keep"] = None
        if as_array:
            full = np.zeros((n, self.model.embed_dim), dtype=np.float32)
            valid = np.zeros(n, dtype=bool)
            if handle["nch"]:
                full[handle["order"]] = host_out
                valid[handle["order"]] = True
                pinned = handle.pop("pinned", None)          # the vectors were copied out: the pinned buffers go round again
                if pinned is not None:
                    self._unpin("tail", pinned[0])
                    self._unpin("xvec", pinned[1])
            return full, valid
        results = [None] * n
        if handle["nch"]:
            for j, u in enumerate(handle["order"].tolist()):
                results[u] = host_out[j]
        return results
