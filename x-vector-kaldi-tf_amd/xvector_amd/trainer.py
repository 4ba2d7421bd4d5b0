"""Training / evaluation step on the MI355X.

Host side of what ``Model.train_one_iteration`` / ``Model.eval`` run per minibatch in the reference
(local/tf/models.py:243-289 / 325-343): forward in train or eval phase, softmax cross-entropy loss + accuracy,
backward, Adam, batch-norm moving-average updates.  Every arithmetic step is a kernel behind the C ABI
(include/xvector_hip.h, "training step" section): the forward and input-gradient GEMMs are the exact-fp32 inference
kernel (dgrad = same kernel, taps flipped + Cin/Cout swapped), plus wgrad, reductions, BN/pooling backward, softmax-CE,
Adam.  PyTorch holds the device buffers and does layout shuffles (flip/permute) only.

Covers every model class of topology.py: ``ModelWithoutDropout`` (the one the recipe trains, run_xvector.sh:90), its
Tdnn / PReLU / LRelu variants, the L2-loss terms of the ``ModelL2Loss*`` classes, and class ``Model``'s dropout
(tf.nn.dropout after BN of every frame-level / embedding layer but the last of each group, models.py:70-72,92-94) with a
stateless counter-based mask (xv_dropout_f32) instead of TF's random stream.
"""
import math
import os

import numpy as np

from . import hiplib
from . import topology as tp
from .engine import BatchLayout

ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8          # tf.train.AdamOptimizer defaults (models.py:112)
BN_DECAY = 0.95                                        # batch_norm_wrapper(h, decay=0.95, ...)  models.py:65


class Trainer(object):
    def __init__(self, weights, topo, device="cuda:0", adam=None, precision="fp32"):
        """precision: arithmetic of the forward, input-gradient and weight-gradient GEMMs -- "fp32" (exact fp32 MFMA) or "bf16x3"
        (split bf16 MFMA with fp32 accumulate, fp32 rows in/out; ~1e-5 relative on every activation/gradient).  Reductions,
        BN, loss and Adam are fp32/fp64-accumulated either way."""
        import torch
        hiplib.require_gpu()
        assert precision in ("fp32", "bf16x3")
        self.attention = tp.is_attention(topo)              # self-attentive pooling (models.py:1036-1050)
        self.precision = precision
        self.skinny_fc = os.environ.get("XVECTOR_TRAIN_SPLITK_FC", "1") != "0"
        # BN-backward column sums from their producers (the input-gradient GEMM's epilogue; the pooling gradient's per-chunk form)
        # instead of a pass over the gradient: XVECTOR_TRAIN_FUSED_SUMS=0 restores the separate col_sums launches (A/B, tests)
        self.fused_sums = os.environ.get("XVECTOR_TRAIN_FUSED_SUMS", "1") != "0"
        self.torch = torch
        self.device = torch.device(device)
        self.topo = topo
        self.act = tp.ACT_CODES[topo.get("activation", "relu")]
        self.prelu = self.act == tp.ACT_PRELU
        self.has_dropout = bool(topo.get("dropout", False))
        self.alpha = float(topo.get("lrelu_alpha", 0.2)) if self.act == tp.ACT_LRELU else 0.0
        self.alpha_t = torch.tensor([self.alpha], dtype=torch.float32, device=self.device) if self.act == tp.ACT_LRELU else None
        self.gap = tp.max_halo(topo)
        self.P = {k: torch.as_tensor(np.array(v, dtype=np.float32, order="C")).to(self.device) for k, v in weights.items()}
        self.feat_dim = int(self.P["frame_level_info_layer-0/w:0"].shape[1])
        self.in_dim = (self.feat_dim + 3) // 4 * 4
        self.num_classes = int(self.P["output/w:0"].shape[1])
        self.frame_scopes = ["frame_level_info_layer-%d" % i for i in range(len(topo["layer_sizes"]))]
        self.embed_scopes = ["embed_layer-%d" % j for j in range(len(topo["embedding_sizes"]))]
        self.trainable = []
        for sc in self.frame_scopes + self.embed_scopes:
            self.trainable += ["%s/%s:0" % (sc, n) for n in ("w", "b", "gamma", "beta")]
            if self.prelu:
                self.trainable.append(sc + "/prelu/prelu:0")
        if self.attention:
            self.trainable += ["attention/w:0", "attention/b:0", "attention/v:0"]
        self.trainable += ["output/w:0", "output/b:0"]
        self.t = int(adam["t"]) if adam else 0
        # every trainable tensor (and its Adam slots) is a view into ONE flat buffer, so the optimizer is one launch per step
        # and the data-parallel all-reduce needs no staging copy; segments start on 64-element boundaries (kernel alignment)
        offs, total = {}, 0
        for n in self.trainable:
            offs[n] = total
            total += (self.P[n].numel() + 63) // 64 * 64
        self._offs, self._flat_n = offs, total
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.flat_g = torch.zeros_like(self.flat_p)                 # gradients land here directly (views in self.G)
        # BN moving statistics and the batch statistics of a step: two flat buffers with the same layout -> one EMA launch
        stat_names = [sc + sfx for sc in self.frame_scopes + self.embed_scopes for sfx in ("/mean:0", "/variance:0")]
        so, stotal = {}, 0
        for n in stat_names:
            so[n] = stotal
            stotal += (self.P[n].numel() + 63) // 64 * 64
        self.flat_moving = torch.zeros(stotal, dtype=torch.float32, device=self.device)
        self.flat_batch = torch.zeros_like(self.flat_moving)
        self.B = {}
        for n in stat_names:
            k = self.P[n].numel()
            self.flat_moving[so[n]:so[n] + k].copy_(self.P[n])
            self.P[n] = self.flat_moving[so[n]:so[n] + k]
            self.B[n] = self.flat_batch[so[n]:so[n] + k]
        self.m, self.v, self.G = {}, {}, {}
        for n in self.trainable:
            k, shape = self.P[n].numel(), self.P[n].shape
            seg = slice(offs[n], offs[n] + k)
            self.flat_p[seg].copy_(self.P[n].reshape(-1))
            self.P[n] = self.flat_p[seg].view(shape)
            self.m[n], self.v[n] = self.flat_m[seg].view(shape), self.flat_v[seg].view(shape)
            self.G[n] = self.flat_g[seg].view(shape)
            if adam and n in adam["m"]:
                self.m[n].copy_(torch.as_tensor(np.array(adam["m"][n], np.float32)).to(self.device).view(shape))
            if adam and n in adam["v"]:
                self.v[n].copy_(torch.as_tensor(np.array(adam["v"][n], np.float32)).to(self.device).view(shape))
        self._packed = None
        self._plan = None
        self._layouts = {}
        self._one_start = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._one_len = {}
        self._side_busy = False
        self.two_streams = os.environ.get("XVECTOR_TRAIN_STREAMS", "2") != "1"
        self.wgrad_after = os.environ.get("XVECTOR_TRAIN_WGRAD_AFTER", "1") != "0"
        self.fused_bias = os.environ.get("XVECTOR_TRAIN_FUSED_BIAS", "1") != "0"
        self._splits = {}                                          # split-format copies for the K = 1 layers' GEMMs (bf16x3)
        self.split_k1 = os.environ.get("XVECTOR_TRAIN_SPLIT_K1", "1") != "0"
        self._side = None                                          # second stream: weight gradients beside the input-gradient GEMMs
        self.l2_beta = float(topo.get("l2_beta", 0.0))
        self.l2_terms = (("embed_layer-0", 0.1), ("embed_layer-1", 1.0), ("output", 1.0))       # models.py:811-832
        head = topo.get("head") or {}
        self.am = head if head.get("type") == "am_softmax" else None         # build-defined additive-margin head

    def _alpha(self, scope):
        """act_alpha argument of the layer kernels: per-channel vector (PReLU), 1-element tensor (LReLU) or None."""
        return self.P[scope + "/prelu/prelu:0"] if self.prelu else self.alpha_t

    @staticmethod
    def dropout_seed(seed, step, layer):
        """64-bit mask seed of one dropout site: (random_seed, optimizer step, site index)."""
        return ((int(seed) & 0xFFFFFFFF) << 32) ^ ((int(step) & 0xFFFFFF) << 8) ^ (int(layer) & 0xFF)

    def _dropout_sites(self):
        """[(kind, index)] of the tf.nn.dropout sites: after every frame-level / embedding layer but the last of its group."""
        return [("frame", i) for i in range(len(self.frame_scopes) - 1)] + [("embed", j) for j in range(len(self.embed_scopes) - 1)]

    def _l2_launch(self):
        """The sums of squares of the penalised tensors, enqueued (None for classes without the L2 term): read by _l2_read."""
        if not self.l2_beta:
            return None
        torch = self.torch
        names = [sc + s for sc, _ in self.l2_terms for s in ("/w:0", "/b:0")]
        out = torch.empty(len(names), dtype=torch.float32, device=self.device)
        for i, n in enumerate(names):
            hiplib.sumsq(self.P[n], out[i:i + 1])
        return out

    def _l2_read(self, out):
        if out is None:
            return 0.0
        v = out.cpu().numpy().astype(np.float64)
        return self._l2_from_host(v)

    def _l2_value(self):
        """beta * sum coef * tf.nn.l2_loss(t) over the penalised tensors (0 for classes without the L2 term)."""
        return self._l2_read(self._l2_launch())

    # -- weights in kernel layout (re-packed after every optimizer step) -----------------------------------------
    def _w3(self, scope, k):
        w = self.P[scope + "/w:0"]
        if w.dim() == 2:
            w = w.unsqueeze(0)
        if scope == self.frame_scopes[0] and self.in_dim != self.feat_dim:
            pad = self.torch.zeros((w.shape[0], self.in_dim - self.feat_dim, w.shape[2]), dtype=w.dtype, device=w.device)
            w = self.torch.cat([w, pad], dim=1)
        return w.contiguous()

    def _pack(self):
        if self._packed is not None:
            return self._packed
        scopes = self.frame_scopes + self.embed_scopes + ["output"] + (["attention"] if self.attention else [])
        pk = {}
        plan_scopes = []
        for sc in scopes:
            w = self.P[sc + "/w:0"]
            K, cin, cout = (1,) + tuple(w.shape) if w.dim() == 2 else tuple(w.shape)
            cin_pad = self.in_dim if sc == self.frame_scopes[0] else cin
            if self.precision == "bf16x3" and cin_pad % 4 == 0 and cout % 4 == 0:
                plan_scopes.append((sc, w.view(K, cin, cout), cin_pad))
                continue
            w = self._w3(sc, None)                                           # [K, Cin, Cout]
            # dgrad: dx[r,c] = sum_{k,o} dz[r - (k-(K-1)/2)d, o] w[k,c,o]  == the forward kernel on w'[k',o,c] = w[K-1-k',c,o]
            wt = w.flip(0).permute(0, 2, 1).contiguous()                     # [K, Cout, Cin]
            pk[sc] = hiplib.pack_weights(w.reshape(K * cin_pad, cout))
            pk[sc + "/T"] = hiplib.pack_weights(wt.reshape(K * cout, cin_pad))
        if plan_scopes:
            # the parameters are views into ONE flat buffer that Adam updates in place: the plan (pointers, destinations) is built
            # once, a step re-packs every layer in both orientations with a single launch (xv_pack_weights_bf16x3_many)
            if self._plan is None:
                self._plan = hiplib.PackPlan([(w, cin_pad, True) for _, w, cin_pad in plan_scopes])
            self._plan.repack()
            for i, (sc, _, _) in enumerate(plan_scopes):
                pk[sc], pk[sc + "/T"] = self._plan.fwd[i], self._plan.bwd[i]
        self._packed = pk
        return pk

    def _split_for(self, role, rows, channels):
        """A bf16 split-format buffer (hiplib.SplitBuf) of ``channels`` channels for >= ``rows`` rows, one per role, kept across steps.
        The K = 1 layers of a bf16x3 step read their GEMM input from it -- written by the kernel that produces the fp32 rows anyway
        (xv_rows_affine_split_f32 / xv_bn_act_backward_split_f32) -- and so take the DMA-fed GEMM (K as a template constant) instead of
        the one that splits fp32 rows while staging them: at K = 1 that one stages a new A tile in EVERY stage.  Rows past ``rows`` hold
        an earlier minibatch's values: only tile rows past R read them, and those are not written."""
        buf = self._splits.get((role, channels))
        if buf is None or buf.rows < rows:
            buf = self._splits[(role, channels)] = hiplib.SplitBuf(max(rows, 28672), channels, self.device)
        return buf

    def _wants_split(self, K, channels, other):
        """A split-format copy of a ``channels``-wide operand for the K = 1 GEMM that reads it?  Only when that GEMM's weights are
        tiled for the DMA-fed kernel (_pack: both of its dimensions multiples of 4; ``other`` is the one that is not ``channels``) --
        a layer whose weights stay fp32-packed runs on the fp32 rows."""
        return (self.split_k1 and self.precision == "bf16x3" and K == 1 and channels % 32 == 0 and other % 4 == 0 and
                not self.prelu)

    def _stage_in(self, x):
        """A contiguous float16 / float32 / int32 host array -> a device tensor of the same dtype, through one of two alternating
        pinned buffers per dtype (the copy is asynchronous -- a pageable ``.to(device)`` in the middle of a step makes the host
        wait for everything enqueued before it).  The buffer of step i is written again by step i + 2: an event recorded behind
        each copy is waited for before the host overwrites the buffer, so a caller that keeps more than two steps in flight
        (step_async) waits here instead of corrupting a minibatch whose copy has not run yet."""
        torch = self.torch
        dt = {np.dtype(np.float16): torch.float16, np.dtype(np.float32): torch.float32, np.dtype(np.int32): torch.int32}[x.dtype]
        slots = self.__dict__.setdefault("_in_stage", {})
        turns = self.__dict__.setdefault("_in_turn", {})
        turn = turns[dt] = turns.get(dt, 0) ^ 1
        pin, copied = slots.get((dt, turn), (None, None))
        if pin is None or pin.numel() < x.size:
            pin, copied = torch.empty(max(x.size, 1) * 5 // 4 + 64, dtype=dt).pin_memory(), torch.cuda.Event()
            slots[(dt, turn)] = (pin, copied)
        else:
            copied.synchronize()                        # (returns at once when the copy of two steps ago has run, the normal case)
        host = pin[:x.size]
        host.numpy()[:] = x.reshape(-1)
        dev = host.to(self.device, non_blocking=True)
        copied.record()
        return dev

    def _layout(self, B, T):
        key = (B, T)
        if key not in self._layouts:
            torch = self.torch
            lay = BatchLayout([T] * B, self.gap)
            # The three index arrays of a minibatch of B equal chunks are arithmetic: they are generated ON the device.  (Copies
            # of pageable host arrays are host synchronisations; with the loss read back one step late the stream is NOT empty at
            # the head of a step, and a run that draws its length per minibatch -- 201 lengths -- met a new one, and with it a
            # full stop of the host, in most of its first few hundred steps.)
            assert lay.align == 1 and lay.lead == self.gap and lay.rows == self.gap + B * (T + self.gap)
            rs, rl, rv = hiplib.minibatch_layout(B, T, self.gap, lay.rows, self.device)
            if B not in self._one_len:
                self._one_len[B] = torch.full((1,), B, dtype=torch.int32, device=self.device)
            self._layouts[key] = dict(lay=lay, rs=rs, rl=rl, rv=rv, one_start=self._one_start, one_len=self._one_len[B])
        return self._layouts[key]

    def _bn_scopes_stats(self, r, scope, L, rows_per_chunk, nchunks, train, valid, frame_level, split_out=None, moments_ws=None):
        """BN (train: batch statistics + moving-average update; eval: moving statistics) applied to r -> h.
        moments_ws: the partial sums (r, r^2) the layer's GEMM left behind -- moments and fold in one launch, no pass over r."""
        torch = self.torch
        C = r.shape[1]
        if train and moments_ws is not None:
            mean, var = self.B[scope + "/mean:0"], self.B[scope + "/variance:0"]
            scale, shift = hiplib.bn_moments_fold(moments_ws, r.shape[0], float(rows_per_chunk * nchunks), self.P[scope + "/gamma:0"],
                                                  self.P[scope + "/beta:0"], tp.BN_EPSILON, mean, var)
            self._last_chunk_moments = None
            h = torch.empty_like(r)
            hiplib.rows_affine(r, scale, shift, valid, h, y_split=split_out)
            return h, mean, var
        if train and not frame_level and self.fused_sums and r.shape[0] <= hiplib.BN_SMALL_MAX_ROWS and valid is None and split_out is None:
            # the segment level (64 rows): moments, fold and affine in ONE launch instead of four dependent ones
            mean, var = self.B[scope + "/mean:0"], self.B[scope + "/variance:0"]
            h = torch.empty_like(r)
            hiplib.bn_small_forward(r, self.P[scope + "/gamma:0"], self.P[scope + "/beta:0"], tp.BN_EPSILON, mean, var, h)
            return h, mean, var
        if train:
            cm = torch.empty((nchunks, 2 * C), dtype=torch.float32, device=self.device)
            rs, rl = (L["rs"], L["rl"]) if frame_level else (L["one_start"], L["one_len"])
            hiplib.chunk_moments(r, rs, rl, nchunks, rows_per_chunk, cm)
            mean, var = self.B[scope + "/mean:0"], self.B[scope + "/variance:0"]      # moving averages: one EMA at step end
            hiplib.merge_moments(cm, rl, nchunks, mean, var)
            self._last_chunk_moments = cm                  # (the last frame-level layer's feed the fused pooling backward)
        else:
            mean, var = self.P[scope + "/mean:0"], self.P[scope + "/variance:0"]
            self._last_chunk_moments = None
        scale, shift = hiplib.fold_bn(self.P[scope + "/gamma:0"], self.P[scope + "/beta:0"], mean, var, tp.BN_EPSILON)
        h = torch.empty_like(r)
        hiplib.rows_affine(r, scale, shift, valid, h, y_split=split_out)
        return h, mean, var

    # -- forward ---------------------------------------------------------------------------------------------------
    def _forward(self, x, labels, train, want_grad, keep_prob=1.0, seed=0):
        torch = self.torch
        x = np.asarray(x)
        B, T, F = x.shape
        assert F == self.feat_dim, "feature dimension %d does not match the model (%d)" % (F, self.feat_dim)
        labels = np.asarray(labels)
        # the reference one-hot encodes on the host and raises IndexError for a label outside the output layer
        # (create_one_hot_output_matrix, models.py:164-169); the loss kernel indexes logits[b, label] and must never see one
        if labels.shape != (B,) or (B and (labels.min() < 0 or labels.max() >= self.num_classes)):
            raise IndexError("labels must be %d integers in [0, %d): got shape %s, range [%s, %s] (egs built with another "
                             "--num-targets?)" % (B, self.num_classes, labels.shape, labels.min() if labels.size else "-",
                                                  labels.max() if labels.size else "-"))
        L = self._layout(B, T)
        lay = L["lay"]
        lab = self._stage_in(np.ascontiguousarray(labels, dtype=np.int32))
        # the minibatch goes up as it is (float16 from the egs loader, examples_io.py:165,176) through a pinned staging buffer;
        # conversion to fp32 and the scatter into rows with gaps happen on the device (xv_pack_minibatch_f32)
        if x.dtype not in (np.float16, np.float32):
            x = x.astype(np.float32)
        raw = self._stage_in(np.ascontiguousarray(x))
        X = torch.empty((lay.rows, self.in_dim), dtype=torch.float32, device=self.device)
        hiplib.pack_minibatch(raw, B, T, F, self.gap, X)
        pk = self._pack()
        drop = train and self.has_dropout and keep_prob < 1.0
        sites = self._dropout_sites()
        S = dict(L=L, B=B, T=T, R=lay.rows, X=X, r=[], z=[], h=[X], hs=None, mean=[], var=[], keep=float(keep_prob) if drop else 1.0,
                 seeds={site: self.dropout_seed(seed, self.t, n) for n, site in enumerate(sites)})
        for i, sc in enumerate(self.frame_scopes):
            K, d = self.topo["kernel_sizes"][i], self.topo["dilations"][i]
            C = self.topo["layer_sizes"][i]
            r = torch.empty((lay.rows, C), dtype=torch.float32, device=self.device)
            z = torch.empty_like(r) if (self.prelu and want_grad) else None            # PReLU backward needs the pre-activation
            # BN's batch moments from the GEMM's own epilogue (per-tile sums of r and r^2 in double) for every layer but the last,
            # whose per-chunk moments feed the fused pooling backward
            moments_ws = None
            if train and self.fused_sums and self.precision == "bf16x3" and i + 1 < len(self.frame_scopes) and hiplib.supports_sums(C):
                moments_ws = hiplib.col_sums_workspace(lay.rows, C, self.device)
                hiplib.tdnn_layer3_moments(S["hs"] if S["hs"] is not None else S["h"][-1], lay.rows, pk[sc], self.P[sc + "/b:0"], self.act,
                                           self._alpha(sc), d, L["rv"], r, z, moments_ws)
            else:
                hiplib.tdnn_layer(S["hs"] if S["hs"] is not None else S["h"][-1], pk[sc], self.P[sc + "/b:0"], None, None, self.act,
                                  self._alpha(sc), K, d, L["rv"], r, z, rows=lay.rows)
            # the next layer's input once more in the split format when that layer is context-free (and nothing rewrites h after BN)
            nxt_split = None
            if i + 1 < len(self.frame_scopes) and self._wants_split(self.topo["kernel_sizes"][i + 1], C, self.topo["layer_sizes"][i + 1]) and \
                    not (drop and ("frame", i) in S["seeds"]):
                nxt_split = self._split_for("h%d" % (i & 1), lay.rows, C)
            h, mean, var = self._bn_scopes_stats(r, sc, L, T, B, train, L["rv"], True, split_out=nxt_split, moments_ws=moments_ws)
            S["hs"] = nxt_split
            if drop and ("frame", i) in S["seeds"]:
                hiplib.dropout(h, S["seeds"][("frame", i)], S["keep"])
            S["r"].append(r); S["z"].append(z); S["h"].append(h); S["mean"].append(mean); S["var"].append(var)
            S["cm_last"] = self._last_chunk_moments
        Cl = self.topo["layer_sizes"][-1]
        if self.attention:
            # h = [h1 | h2]: u = h1.W + b (one more K=1 GEMM), scores = v.tanh(u), softmax over the frames of each chunk,
            # weighted mean / std of h2
            A = Cl // 2
            hl = S["h"][-1]
            u = torch.empty((lay.rows, A), dtype=torch.float32, device=self.device)
            hiplib.tdnn_layer(hl[:, :A], pk["attention"], self.P["attention/b:0"], None, None, tp.ACT_NONE, None, 1, 1, None, u)
            scores = torch.empty(lay.rows, dtype=torch.float32, device=self.device)
            S["nl"] = torch.empty_like(u) if want_grad else None
            hiplib.attention_scores(u, self.P["attention/v:0"], scores, S["nl"])
            S["att"] = torch.zeros(lay.rows, dtype=torch.float32, device=self.device)
            hiplib.attention_softmax(scores, L["rs"], L["rl"], B, S["att"])
            pooled = torch.empty((B, 2 * A), dtype=torch.float32, device=self.device)
            hiplib.attention_pool(hl[:, A:], S["att"], L["rs"], L["rl"], B, T, 512, tp.VAR2STD_EPSILON, pooled,
                                  hiplib._ws(hiplib.attention_pool_workspace_bytes(A, B, T, 512), self.device))
        else:
            pooled = torch.empty((B, 2 * Cl), dtype=torch.float32, device=self.device)
            hiplib.stats_pool(S["h"][-1], L["rs"], L["rl"], B, T, 512, tp.VAR2STD_EPSILON, pooled,
                              hiplib._ws(hiplib.stats_pool_workspace_bytes(Cl, B, T, 512), self.device))
        S["pooled"] = pooled
        S["e_in"], S["e_r"], S["e_z"], S["e_mean"], S["e_var"] = [pooled], [], [], [], []
        for j, sc in enumerate(self.embed_scopes):
            C = self.topo["embedding_sizes"][j]
            r = torch.empty((B, C), dtype=torch.float32, device=self.device)
            z = torch.empty_like(r) if (self.prelu and want_grad) else None
            e_in = S["e_in"][-1]
            if self.skinny_fc and hiplib.fc_splitk_supported(B, e_in.shape[1], C):
                # 64 rows x 3072 -> 512: four output tiles would each walk 96 slabs one after the other (168 us); split-K, exact fp32
                key = sc + "/fc32"
                if key not in pk:
                    pk[key] = hiplib.pack_weights(self.P[sc + "/w:0"])
                hiplib.fc_splitk(e_in, pk[key], self.P[sc + "/b:0"], None, None, self.act, self._alpha(sc), r, z)
            else:
                hiplib.fc(e_in, pk[sc], self.P[sc + "/b:0"], None, None, self.act, self._alpha(sc), r, z)
            a, mean, var = self._bn_scopes_stats(r, sc, L, B, 1, train, None, False)
            if drop and ("embed", j) in S["seeds"]:
                hiplib.dropout(a, S["seeds"][("embed", j)], S["keep"])
            S["e_r"].append(r); S["e_z"].append(z); S["e_in"].append(a); S["e_mean"].append(mean); S["e_var"].append(var)
        logits = torch.empty((B, self.num_classes), dtype=torch.float32, device=self.device)
        if self.am:
            # cosines of the L2-normalised features and class vectors (columns of output/w), then margin + scale in place
            S["xh"], S["xnorm"] = hiplib.l2_normalize_rows(S["e_in"][-1])
            wt = self.P["output/w:0"].t().contiguous()                                   # [classes, E]: class vectors as rows
            S["wh_t"], S["wnorm"] = hiplib.l2_normalize_rows(wt)
            # the FC takes its weights as [Out, In] rows (xv_pack_weights_f32 is that transposition of [In, Out]): the normalised
            # class vectors ARE that matrix -- no transpose-and-pack-back pair of launches
            hiplib.fc(S["xh"], S["wh_t"], None, None, None, tp.ACT_NONE, None, None, logits)
            hiplib.am_margin(logits, lab, self.am["scale"], self.am["margin"])
        else:
            hiplib.fc(S["e_in"][-1], pk["output"], self.P["output/b:0"], None, None, tp.ACT_NONE, None, None, logits)
        loss_acc = torch.empty(2, dtype=torch.float32, device=self.device)
        dlogits = torch.empty_like(logits) if want_grad else None
        hiplib.softmax_ce(logits, lab, loss_acc, dlogits)
        S["dlogits"] = dlogits
        S["loss_acc"] = loss_acc
        return S

    def eval_batch(self, x, labels):
        """(loss, accuracy) of one minibatch in the eval phase (moving BN statistics)."""
        la = self._forward(x, labels, train=False, want_grad=False)["loss_acc"].cpu().numpy()
        return float(la[0]) + self._l2_value(), float(la[1])

    # -- backward + Adam -------------------------------------------------------------------------------------------
    def _dense_backward(self, scope, x_in, dz, K, dil, grads, need_dx, valid, dx_out=None, dz_split=None, sums=None):
        """dW, db (and dx) of  z = conv(x_in, W) + b  given dz.  dx_out: optional [R, Cin] rows that receive dx.
        sums = (r_below, workspace): the input-gradient GEMM also leaves the partial column sums of (dx, dx * r_below) in the
        workspace (xv_tdnn_layer_bf16x3_sums) -- what the BN backward of the layer below starts from."""
        torch = self.torch
        pk = self._pack()
        R, cin = x_in.shape
        cout = dz.shape[1]
        gw, db = self.G[scope + "/w:0"], self.G[scope + "/b:0"]

        def weight_side():
            # (bf16x3: the bias gradient comes out of the weight-gradient kernel, which streams dz anyway -- xv_wgrad_bias_bf16x3; the
            # separate pass over dz was 4 % of a step.  XVECTOR_TRAIN_FUSED_BIAS=0: xv_col_sums_f32 as before)
            fused = self.fused_bias and hiplib.wgrad_takes_bias(self.precision, x_in, dz)
            if scope == self.frame_scopes[0] and self.in_dim != self.feat_dim:
                dw = torch.empty((K, cin, cout), dtype=torch.float32, device=self.device)
                hiplib.wgrad(x_in, dz, K, dil, dw, self.precision, db=db if fused else None)
                gw.copy_(dw[:, :self.feat_dim, :])                          # drop the padding column
            else:
                hiplib.wgrad(x_in, dz, K, dil, gw.view(K, cin, cout), self.precision, db=db if fused else None)
            if not fused:
                hiplib.col_sums(dz, None, db)

        grads[scope + "/w:0"] = gw
        grads[scope + "/b:0"] = db
        # dW / db and dx only share their inputs: the weight side goes to a second stream, so that its workgroups fill the
        # last, partly empty round of the input-gradient GEMM (a minibatch is 1.2 rounds of 128-row tiles) and vice versa;
        # whoever reads the gradients next (a bucket's all-reduce, Adam) waits for that stream (_join_side)
        if need_dx and self.two_streams:
            main = torch.cuda.current_stream(self.device)
            side = self._side_stream()
            ready = None
            if self.wgrad_after:
                ready = torch.cuda.Event()
                ready.record(main)                                     # dz (and x_in) are final here
            else:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    weight_side()
        else:
            weight_side()
        if not need_dx:
            return None
        dx = dx_out if dx_out is not None else torch.empty((R, cin), dtype=torch.float32, device=self.device)
        if sums is not None:
            hiplib.tdnn_layer3_sums(dz_split if dz_split is not None else dz, R, pk[scope + "/T"], dil, valid, dx, sums[0], sums[1])
        else:
            hiplib.tdnn_layer(dz_split if dz_split is not None else dz, pk[scope + "/T"], None, None, None, tp.ACT_NONE, None, K, dil, valid,
                              dx, rows=R)
        if need_dx and self.two_streams:
            if self.wgrad_after:
                # the weight side is queued BEHIND the input-gradient GEMM it runs beside (it waits for dz, not for that GEMM): when
                # both are ready the hardware takes the critical path's workgroups first
                side.wait_event(ready)
                with torch.cuda.stream(side):
                    weight_side()
            dz.record_stream(side)
            x_in.record_stream(side)
            self._side_busy = True
        return dx

    def _side_stream(self):
        """The second stream (weight gradients, bias sums, the moving-average update).  (Stream priorities were tried: the device
        offers 0 and -1 only, so the side stream cannot go below the default, and the step on a -1 stream is 2 % SLOWER --
        profiles/r06_train_stream_ab.txt.)"""
        if self._side is None:
            self._side = self.torch.cuda.Stream(device=self.device)
        return self._side

    def _join_side(self):
        if self._side_busy:
            self.torch.cuda.current_stream(self.device).wait_stream(self._side)
            self._side_busy = False

    def _bn_backward(self, scope, dh, r, z, mean, var, n_frames, valid, grads, split_out=None, sums_ws=None, pool=None):
        """dz = dL/d(pre-activation) and dgamma / dbeta, given dh = dL/d(BN output).  The two column sums it starts from come from
        (a) ``pool`` = (h, row_start, row_len, nchunks, pooled, dpooled, chunk moments of r): the last frame-level layer, whose dh
        is the pooling's gradient -- sums from per-chunk numbers, dh never materialised (dh is None); (b) ``sums_ws``: partial sums
        the input-gradient GEMM that produced dh left behind; (c) a pass over dh and r (col_sums)."""
        torch = self.torch
        C = r.shape[1]
        dgamma, dbeta = self.G[scope + "/gamma:0"], self.G[scope + "/beta:0"]
        dz = torch.empty_like(r)
        act = tp.ACT_NONE if self.prelu else self.act
        if pool is not None:
            h, rs, rl, nchunks, pooled, dpooled, cm = pool
            hiplib.pool_bn_act_backward(h, r, rs, rl, nchunks, pooled, dpooled, cm, mean, var, self.P[scope + "/gamma:0"], tp.BN_EPSILON,
                                        n_frames, act, self.alpha, dgamma, dbeta, dz, dz_split=split_out)
        elif self.fused_sums and valid is None and split_out is None and r.shape[0] <= hiplib.BN_SMALL_MAX_ROWS and float(r.shape[0]) == n_frames:
            hiplib.bn_small_backward(dh, r, mean, var, self.P[scope + "/gamma:0"], tp.BN_EPSILON, act, self.alpha, dgamma, dbeta, dz)
        elif sums_ws is not None:
            hiplib.bn_act_backward_parts(dh, r, sums_ws, mean, var, self.P[scope + "/gamma:0"], tp.BN_EPSILON, n_frames, act, self.alpha,
                                         valid, dgamma, dbeta, dz, dz_split=split_out)
        else:
            s1 = torch.empty(C, dtype=torch.float32, device=self.device)
            s2 = torch.empty_like(s1)
            hiplib.col_sums(dh, r, s1, s2)
            hiplib.bn_act_backward(dh, r, s1, s2, mean, var, self.P[scope + "/gamma:0"], tp.BN_EPSILON, n_frames, act, self.alpha, valid,
                                   dgamma, dbeta, dz, dz_split=split_out)
        if self.prelu:
            # dz holds dL/d(act output); z the pre-activation: -> dz = dL/dz, z = dr*min(z,0) whose column sums are dalpha
            hiplib.prelu_backward(dz, z, self.P[scope + "/prelu/prelu:0"])
            dalpha = self.G[scope + "/prelu/prelu:0"]
            hiplib.col_sums(z, None, dalpha)
            grads[scope + "/prelu/prelu:0"] = dalpha
        grads[scope + "/gamma:0"] = dgamma
        grads[scope + "/beta:0"] = dbeta
        return dz

    def _ready_ranges(self):
        """[(start, end, frame layer after whose backward the range is final | None)]: contiguous element ranges of
        ``flat_g`` in the order the backward pass completes them -- the segment-level tail (embedding layers, attention,
        output; final before the frame-level backward starts), then frame layers 4+3, 2, 1+0.  These are the buckets of the
        data-parallel all-reduce; together they cover every trainable tensor."""
        def span(names):
            a = min(self._offs[n] for n in names)
            b = max(self._offs[n] + (self.P[n].numel() + 63) // 64 * 64 for n in names)
            return a, b
        of = lambda scopes: [n for n in self.trainable if n.split("/")[0] in scopes]      # noqa: E731
        nf = len(self.frame_scopes)
        cuts = sorted({0, max(nf - 3, 0), max(nf - 2, 0), nf})
        out = [span(of(set(self.embed_scopes) | {"attention", "output"})) + (None,)]
        for lo, hi in reversed(list(zip(cuts[:-1], cuts[1:]))):
            out.append(span(of(set(self.frame_scopes[lo:hi]))) + (lo,))
        return out

    def _l2_grad(self, scope, grads):
        """g += beta*coef*t for the penalised tensors of ``scope`` (models.py:811-842), right after they were produced."""
        if self.l2_beta:
            self._join_side()                                   # (the weight gradient it adds to may still be on its way)
            for sc, coef in self.l2_terms:
                if sc == scope:
                    for suffix in ("/w:0", "/b:0"):
                        if self.am and sc == "output" and suffix == "/b:0":
                            # the AM head has no bias: its gradient segment is zero from the start and never re-zeroed (gradients()),
                            # so an L2 term added here would ACCUMULATE over the steps.  The bias is not part of that head's model.
                            continue
                        hiplib.axpy(grads[sc + suffix], self.P[sc + suffix], self.l2_beta * coef)

    def gradients(self, x, labels, dropout_proportion=0.0, seed=0, on_bucket=None, defer_loss=False):
        """Forward (train phase, updates the moving statistics) + backward.  Returns (loss, acc, {name: grad tensor});
        ``on_bucket(i)`` (optional) is called as soon as range i of ``_ready_ranges()`` is final;
        the tensors are views into ``self.flat_g`` and are overwritten by the next call.
        dropout_proportion > 0 is honoured by the classes with dropout sites (topology "dropout": True) and ignored by
        the others, as in the reference where only class Model wires the keep-prob placeholder into the graph."""
        torch = self.torch
        S = self._forward(x, labels, train=True, want_grad=True, keep_prob=1.0 - float(dropout_proportion), seed=seed)
        # moving <- 0.95*moving + 0.05*batch, all scopes at once -- beside the backward pass, not in front of it (every launch of the
        # segment level is a ~5 us link of one dependent chain); the stream is joined in front of the optimizer update
        if self.two_streams:
            main = torch.cuda.current_stream(self.device)
            self._side_stream().wait_stream(main)
            with torch.cuda.stream(self._side):
                hiplib.ema(self.flat_moving, self.flat_batch, BN_DECAY)
            self._side_busy = True
        else:
            hiplib.ema(self.flat_moving, self.flat_batch, BN_DECAY)
        L, B, T = S["L"], S["B"], S["T"]
        grads = {}
        if self.am:
            # logits = scale*(cos - margin*onehot): dL/dcos = scale*dlogits; cos = xh @ wh  ->  dxh = g wh^T, dwh = xh^T g;
            # then back through the two L2 normalisations.  output/b is not part of this head: zero gradient.
            g = S["dlogits"]
            hiplib.axpy(g, g, float(self.am["scale"]) - 1.0)
            E = S["xh"].shape[1]
            dwh = torch.empty((1, E, self.num_classes), dtype=torch.float32, device=self.device)
            # (output/b is not part of this head: its gradient segment of flat_g is zero from the start and nothing ever writes it)
            hiplib.wgrad(S["xh"], g, 1, 1, dwh, self.precision)
            dxh = torch.empty_like(S["xh"])
            wh = S["wh_t"].t().contiguous()                                  # [E, classes] = the packed form of wh_t as GEMM weights
            hiplib.tdnn_layer(g, wh, None, None, None, tp.ACT_NONE, None, 1, 1, None, dxh)
            d = hiplib.l2_normalize_backward(dxh, S["xh"], S["xnorm"])
            dwt = hiplib.l2_normalize_backward(dwh[0].t().contiguous(), S["wh_t"], S["wnorm"])       # [classes, E]
            self.G["output/w:0"].copy_(dwt.t())
            grads["output/w:0"], grads["output/b:0"] = self.G["output/w:0"], self.G["output/b:0"]
        else:
            d = self._dense_backward("output", S["e_in"][-1], S["dlogits"], 1, 1, grads, True, None)
        self._l2_grad("output", grads)
        for j in reversed(range(len(self.embed_scopes))):
            sc = self.embed_scopes[j]
            if S["keep"] < 1.0 and ("embed", j) in S["seeds"]:
                hiplib.dropout(d, S["seeds"][("embed", j)], S["keep"])
            dz = self._bn_backward(sc, d, S["e_r"][j], S["e_z"][j], S["e_mean"][j], S["e_var"][j], float(B), None, grads)
            d = self._dense_backward(sc, S["e_in"][j], dz, 1, 1, grads, True, None)
            self._l2_grad(sc, grads)
        pool = None
        if self.attention:
            hl = S["h"][-1]
            A = hl.shape[1] // 2
            dh = torch.zeros_like(hl)                                   # [dh1 | dh2]; gap rows stay zero
            datt = torch.zeros(hl.shape[0], dtype=torch.float32, device=self.device)
            dscores = torch.zeros_like(datt)
            hiplib.attention_pool_backward(hl[:, A:], S["att"], L["rs"], L["rl"], B, T, S["pooled"], d, dh[:, A:], datt)
            hiplib.attention_softmax_backward(S["att"], datt, L["rs"], L["rl"], B, dscores)
            du = torch.empty_like(S["nl"])
            hiplib.attention_scores_backward(S["nl"], dscores, self.P["attention/v:0"], du)     # S["nl"] <- dscores * tanh(u)
            hiplib.col_sums(S["nl"], None, self.G["attention/v:0"])
            grads["attention/v:0"] = self.G["attention/v:0"]
            self._dense_backward("attention", hl[:, :A], du, 1, 1, grads, True, None, dx_out=dh[:, :A])
        elif self.fused_sums and S.get("cm_last") is not None and S["h"][-1].shape[1] % 4 == 0:
            # the pooling's gradient is consumed where it is formed (xv_pool_bn_act_backward_f32): no dh, no pass for its sums
            dh, pool = None, (S["h"][-1], L["rs"], L["rl"], B, S["pooled"], d, S["cm_last"])
        else:
            dh = torch.empty_like(S["h"][-1])
            hiplib.pool_backward(S["h"][-1], L["rs"], L["rl"], B, S["pooled"], d, dh)
        fire_after = {after: k for k, (_, _, after) in enumerate(self._ready_ranges())}     # frame layer -> bucket final after it
        sums_ws = None
        if on_bucket is not None:
            self._join_side()
            on_bucket(fire_after[None])                                    # segment-level tail (embed / attention / output)
        for i in reversed(range(len(self.frame_scopes))):
            sc = self.frame_scopes[i]
            if S["keep"] < 1.0 and ("frame", i) in S["seeds"]:
                hiplib.dropout(dh, S["seeds"][("frame", i)], S["keep"])
            Ki = self.topo["kernel_sizes"][i]
            dzs = self._split_for("dz", S["R"], S["r"][i].shape[1]) if (i > 0 and self._wants_split(Ki, S["r"][i].shape[1], S["h"][i].shape[1])) else None
            dz = self._bn_backward(sc, dh, S["r"][i], S["z"][i], S["mean"][i], S["var"][i], float(B * T), L["rv"], grads, split_out=dzs,
                                   sums_ws=sums_ws, pool=pool)
            pool = sums_ws = sums = None
            # the input-gradient GEMM leaves the column sums the layer below starts its BN backward from -- unless dropout rewrites
            # that gradient in between, or the GEMM is not the bf16x3 one
            if i > 0 and self.fused_sums and self.precision == "bf16x3" and hiplib.supports_sums(S["r"][i - 1].shape[1]) and \
                    not (S["keep"] < 1.0 and ("frame", i - 1) in S["seeds"]):
                sums_ws = hiplib.col_sums_workspace(S["R"], S["r"][i - 1].shape[1], self.device)
                sums = (S["r"][i - 1], sums_ws)
            dh = self._dense_backward(sc, S["h"][i], dz, Ki, self.topo["dilations"][i], grads, i > 0, L["rv"], dz_split=dzs, sums=sums)
            if on_bucket is not None and i in fire_after:
                self._join_side()
                on_bucket(fire_after[i])
        self._join_side()
        if defer_loss:
            # (step(): loss, accuracy and the L2 penalty of the CURRENT weights are computed here, in stream order in front of
            # the optimizer update, but read back only after that update has been enqueued -- the read is the step's one host
            # synchronisation, and everything launched before it runs while the host waits)
            return (S["loss_acc"], self._l2_launch()), None, grads
        la = S["loss_acc"].cpu().numpy()
        return float(la[0]) + self._l2_value(), float(la[1]), grads

    def step(self, x, labels, learning_rate, dropout_proportion=0.0, seed=0):
        """One optimizer step on a minibatch x[B,T,F] (float16/32), labels[B].  Returns (loss, accuracy): ``step_async`` + the
        read-back, i.e. the host waits for the whole step before it can stage the next minibatch."""
        return self.step_async(x, labels, learning_rate, dropout_proportion, seed).result()

    def step_async(self, x, labels, learning_rate, dropout_proportion=0.0, seed=0):
        """The same step with its ONE host synchronisation handed to the caller: everything is enqueued, (loss, accuracy) travel to a
        pinned slot behind the optimizer update, and ``.result()`` of the returned handle waits for them.  A loop that asks for step
        i's result AFTER it has enqueued step i + 1 (Model.train_one_iteration, bench.py) keeps the GPU busy across the step
        boundary: with the read-back at the end of every step the GPU idles ~0.25 ms per step while the host stages the next
        minibatch (kernel trace: 280 us between the optimizer update and the next step's first GEMM).  At most two steps may be
        outstanding: the input staging buffers and the result slots alternate.


        Data parallelism (one process per GPU): the gradients are averaged over the ranks by bucketed, asynchronous RCCL
        all-reduces of ranges of the flat gradient buffer, each issued the moment the backward pass has finished its range
        (segment-level tail 7.5 MB, frame layers 4+3 4.2 MB, layer 2 7.3 MB, layers 1+0 5.4 MB for the default topology), so
        that all but the last bucket travel over xGMI while the remaining layers are still in their backward pass; Adam
        waits for all of them.  The reference's own multi-job scheme never exchanges anything (its model averaging is a
        stub, ze_utils.py:164-183), so this is build-defined.  BN statistics stay per replica."""
        import torch.distributed as dist
        works = []
        if dist.is_initialized():
            ranges = self._ready_ranges()

            def on_bucket(i):
                a, b, _ = ranges[i]
                works.append(dist.all_reduce(self.flat_g[a:b], op=dist.ReduceOp.SUM, async_op=True))
        else:
            on_bucket = None
        pending, _, grads = self.gradients(x, labels, dropout_proportion, seed, on_bucket, defer_loss=True)   # grads: views into flat_g
        for w in works:
            w.wait()
        if works and dist.get_world_size() > 1:
            hiplib.axpy(self.flat_g, self.flat_g, 1.0 / dist.get_world_size() - 1.0)       # flat_g /= world
        self.t += 1
        lr_t = learning_rate * math.sqrt(1.0 - ADAM_B2 ** self.t) / (1.0 - ADAM_B1 ** self.t)
        hiplib.adam(self.flat_p, self.flat_g, self.flat_m, self.flat_v, lr_t, ADAM_B1, ADAM_B2, ADAM_EPS)
        self._packed = None
        torch = self.torch
        slots = self.__dict__.setdefault("_result_slots", [None, None])
        turn = self.__dict__["_result_turn"] = self.__dict__.get("_result_turn", 0) ^ 1
        n_l2 = 0 if pending[1] is None else pending[1].numel()
        # the slot of step i is reused by step i + 2: the handle that still points at it takes its values out first (a handle awaited
        # late returns ITS step's loss, not a later one's)
        owners = self.__dict__.setdefault("_result_owner", [None, None])
        if owners[turn] is not None:
            owners[turn].result()
        if slots[turn] is None or slots[turn][0].numel() < 2 + n_l2:
            slots[turn] = (torch.empty(2 + n_l2, dtype=torch.float32).pin_memory(), torch.cuda.Event())
        host, done = slots[turn]
        host[:2].copy_(pending[0].view(-1)[:2], non_blocking=True)
        if n_l2:
            host[2:2 + n_l2].copy_(pending[1].view(-1), non_blocking=True)
        done.record()
        owners[turn] = _PendingStep(self, host, done, n_l2)
        return owners[turn]

    def _l2_from_host(self, v):
        """beta * sum coef * (sumsq(w) + sumsq(b)) / 2 from the [w, b] sums of squares per penalised scope; the AM head has no bias, so
        output/b is no part of its penalty (value here, gradient in _l2_grad)."""
        v = np.asarray(v, dtype=np.float64)
        return float(self.l2_beta * sum(coef * 0.5 * (v[2 * i] + (0.0 if (self.am and sc == "output") else v[2 * i + 1]))
                                        for i, (sc, coef) in enumerate(self.l2_terms)))

    def export(self):
        """-> (weights {tf name: float32 ndarray}, adam {"t", "m", "v"}) for the model directory."""
        w = {k: v.cpu().numpy() for k, v in self.P.items()}
        adam = dict(t=self.t, m={k: v.cpu().numpy() for k, v in self.m.items()}, v={k: v.cpu().numpy() for k, v in self.v.items()})
        return w, adam


class _PendingStep(object):
    """Handle of ``Trainer.step_async``: ``result()`` -> (loss, accuracy) once the step's values have reached the host."""

    def __init__(self, trainer, host, done, n_l2):
        self._t, self._host, self._done, self._n_l2 = trainer, host, done, n_l2
        self._value = None

    def result(self):
        if self._value is None:
            self._done.synchronize()
            la = self._host.numpy()
            l2 = self._t._l2_from_host(la[2:2 + self._n_l2]) if self._n_l2 else 0.0
            self._value = (float(la[0]) + l2, float(la[1]))
            self._host = self._done = None              # the pinned slot belongs to a later step from now on
        return self._value


# ------------------------------------------------------------------------------------------------
# which arithmetic may this checkpoint TRAIN in?  (first-minibatch gradient probe)
# ------------------------------------------------------------------------------------------------
# The bf16x3 step (forward / input-gradient / weight-gradient GEMMs on the split-precision MFMA path) is half the time of the
# exact-fp32 step and reproduces its gradients to ~1e-4 .. 1e-3 on networks that look like a TDNN in training.  Nothing forces a
# checkpoint to look like that (cf. engine.select_model for extraction), so the choice is made PER RUN on real data: the first
# minibatch's gradients are computed in both arithmetics and bf16x3 is kept only when every weight / gamma gradient agrees with the
# fp32 one to TRAIN_PROBE_LIMIT relative L2 and the losses to 1e-3 (bias / beta gradients are plain sums of signed terms over all
# frames -- ill-conditioned in ANY arithmetic, see tests/test_gpu_training.py -- and are not part of the verdict).
TRAIN_PROBE_LIMIT = 2e-2


def select_trainer(weights, topo, device, adam, x, labels, logger=None):
    """(Trainer, verdict dict): the bf16x3 trainer when the first minibatch ``(x, labels)`` admits it, else the fp32 one.  The probe
    leaves no trace in either trainer (moving statistics restored; no optimizer step).  In a data-parallel group every rank
    probes its own minibatch and the group takes bf16x3 only if every rank would."""
    import torch
    fast = Trainer(weights, topo, device, adam, precision="bf16x3")
    exact = Trainer(weights, topo, device, adam, precision="fp32")
    keep = [t.flat_moving.clone() for t in (fast, exact)]
    l3, _, g3 = fast.gradients(x, labels)
    l32, _, g32 = exact.gradients(x, labels)
    worst, worst_name = 0.0, ""
    for name, ref in g32.items():
        if name.endswith("/b:0") or name.endswith("/beta:0"):
            continue
        den = float(ref.double().norm().item())
        num = float((g3[name].double() - ref.double()).norm().item())
        rel = num / den if den > 0 else (0.0 if num == 0 else float("inf"))
        if not np.isfinite(rel):
            rel = float("inf")
        if rel > worst:
            worst, worst_name = rel, name
    for t, m in zip((fast, exact), keep):
        t.flat_moving.copy_(m)
    ok = bool(np.isfinite(l3) and np.isfinite(l32) and abs(l3 - l32) <= 1e-3 * max(1.0, abs(l32)) and worst <= TRAIN_PROBE_LIMIT)
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=fast.device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item())
    except ImportError:
        pass
    verdict = dict(selected="bf16x3" if ok else "fp32", worst_gradient_rel_l2=worst, worst_tensor=worst_name, limit=TRAIN_PROBE_LIMIT,
                   loss_bf16x3=float(l3), loss_fp32=float(l32))
    if logger is not None:
        logger.info("Training arithmetic: %s (first-minibatch gradient probe: bf16x3 vs fp32 %.2e on %s, limit %.0e; loss %.6f vs %.6f)" % (
            verdict["selected"], worst, worst_name, TRAIN_PROBE_LIMIT, l3, l32))
    chosen = fast if ok else exact
    del fast, exact, g3, g32, keep                    # the rejected trainer (weights, Adam slots, flat buffers) is freed here, not at some later collection
    return chosen, verdict
