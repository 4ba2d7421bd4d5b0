"""Feature front-end on the MI355X: sliding-window CMN + VAD frame selection.

The reference's extraction script pipes every utterance through two Kaldi binaries before extract_embedding.py sees it
(local/tf/extract_xvectors.sh:68):

    apply-cmvn-sliding --norm-vars=false --center=true --cmn-window=300 scp:feats.scp ark:- |
    select-voiced-frames ark:- scp,s,cs:vad.scp ark:- |

``FrontEnd.apply`` does both for a window of utterances with ONE kernel launch (xv_cmn_sliding_scatter_f32), so the
extractor can read raw MFCC + VAD tables directly and the recipe needs no external processes on the feature side.
``cmn_window <= 0`` switches the normalisation off (frame selection only, a host-side row gather).
Utterance-level rules follow select-voiced-frames: a VAD vector whose length differs from the feature matrix, or one with
no voiced frame, drops the utterance (result ``None``; the binary warns and writes nothing for the key).
"""
import numpy as np

from . import hiplib


class VadRuns(object):
    """The VAD vectors of consecutive utterances as a few RUNS -- (values, offsets) pairs that cover many utterances each -- instead
    of one small array per utterance: what the reader of a vad.scp / VAD ark hands over when the keys arrive in the order of the
    features (extract_xvectors.sh reads them as scp,s,cs).  Behaves as the sequence of per-utterance vectors it stands for; the
    frame selection takes the concatenation directly (50 k generator steps and 50 k tiny arrays per job otherwise)."""

    def __init__(self):
        self._vals, self._offs, self._n = [], [], 0
        self._flat = None

    def add_run(self, vals, offs):
        """vals: the decisions of m utterances back to back; offs: their m + 1 offsets into vals (offs[0] == 0)."""
        if len(offs) > 1:
            self._vals.append(np.asarray(vals).reshape(-1))
            self._offs.append(np.asarray(offs, np.int64))
            self._n += len(offs) - 1
            self._flat = None

    def add_one(self, v):
        v = np.asarray(v).reshape(-1)
        self.add_run(v, np.array([0, v.shape[0]], np.int64))

    def __len__(self):
        return self._n

    def flat(self):
        """(values of all utterances concatenated, their n + 1 offsets)."""
        if self._flat is None:
            if not self._vals:
                self._flat = (np.zeros(0, np.float32), np.zeros(1, np.int64))
            elif len(self._vals) == 1:
                self._flat = (self._vals[0], self._offs[0])
            else:
                base = np.cumsum([0] + [int(o[-1]) for o in self._offs[:-1]])
                offs = np.concatenate([self._offs[0][:1]] + [o[1:] + b for o, b in zip(self._offs, base)])
                self._flat = (np.concatenate(self._vals), offs)
        return self._flat

    def __iter__(self):
        vals, offs = self.flat()
        o = offs.tolist()
        return (vals[o[i]:o[i + 1]] for i in range(self._n))

    def __getitem__(self, i):
        vals, offs = self.flat()
        return vals[int(offs[i]):int(offs[i + 1])]


def select_voiced(mats, vads):
    """The utterance-level rules of select-voiced-frames for a window, vectorised.  Returns
    ``(T, cand, voiced, empty, dropped)``: frame counts of all utterances; indices of the utterances that go on (T > 0, VAD of
    the right length with at least one voiced frame, or no VAD at all); the voiced flags of their frames concatenated in that
    order (None = every frame voiced); indices of utterances that have neither frames nor a VAD (they stay empty matrices);
    the number of utterances dropped because of their VAD (length mismatch / nothing voiced)."""
    n = len(mats)
    T = np.asarray(mats.lengths, np.int64) if hasattr(mats, "lengths") else np.fromiter((m.shape[0] for m in mats), dtype=np.int64, count=n)
    if vads is None:
        return T, np.flatnonzero(T > 0), None, np.flatnonzero(T == 0).tolist(), 0
    if isinstance(vads, VadRuns):
        # every utterance has a vector (possibly empty), and they lie back to back already
        assert len(vads) == n
        vals, offs = vads.flat()
        vl = np.diff(offs)
        cand = np.flatnonzero((vl == T) & (T > 0))                            # a length mismatch drops the key
        if len(cand) == n:
            voiced = vals != 0
        elif len(cand):
            voiced = vals[np.repeat((vl == T) & (T > 0), vl)] != 0
        else:
            voiced = np.zeros(0, bool)
        starts = np.zeros(len(cand), dtype=np.int64)
        np.cumsum(T[cand][:-1], out=starts[1:])
        counts = np.add.reduceat(voiced, starts) if len(cand) else np.zeros(0, np.int64)
        has = counts > 0
        if not has.all():
            voiced = voiced[np.repeat(has, T[cand])]
            cand = cand[has]
        return T, cand, voiced, [], n - len(cand)
    flat = [None if v is None else np.asarray(v).reshape(-1) for v in vads]
    vl = np.fromiter((-1 if v is None else v.shape[0] for v in flat), dtype=np.int64, count=n)
    empty = np.flatnonzero((vl < 0) & (T == 0)).tolist()
    cand = np.flatnonzero(((vl == T) | (vl < 0)) & (T > 0))                  # a length mismatch drops the key
    if len(cand):
        # ONE comparison over the concatenated decisions (not one small array operation per utterance); an utterance without a VAD
        # contributes a slice of ones
        ones = None
        parts = []
        for i in cand.tolist():
            v = flat[i]
            if v is None:
                if ones is None:
                    ones = np.ones(int(T[cand].max()), np.float32)
                v = ones[:int(T[i])]
            parts.append(v)
        voiced = np.concatenate(parts) != 0
    else:
        voiced = np.zeros(0, bool)
    starts = np.zeros(len(cand), dtype=np.int64)
    np.cumsum(T[cand][:-1], out=starts[1:])
    counts = np.add.reduceat(voiced, starts) if len(cand) else np.zeros(0, np.int64)
    has = counts > 0                                                          # ... and so does a VAD without a voiced frame
    if not has.all():                                                         # (the usual window drops nobody: no pass over the frames)
        voiced = voiced[np.repeat(has, T[cand])]
        cand = cand[has]
    dropped = int(np.count_nonzero(vl >= 0)) - int(np.count_nonzero(vl[cand] >= 0))
    return T, cand, voiced, empty, dropped


class FrontEnd(object):
    def __init__(self, device="cuda:0", cmn_window=300, center=True, min_window=100):
        import torch
        hiplib.require_gpu()
        self.torch = torch
        self.device = torch.device(device)
        self.cmn_window = int(cmn_window)
        self.center = bool(center)
        self.min_window = int(min_window)
        self.stats = dict(utterances=0, frames_in=0, frames_out=0, dropped=0)
        # own stream: the result goes back to the host, so nothing on the compute stream depends on it -- and on the compute
        # stream the D2H copy would queue behind the extraction kernels of the previous window (measured: the whole pipeline
        # serialised)
        self._stream = torch.cuda.Stream(device=self.device)

    def apply(self, mats, vads=None):
        """mats: list of float32 [T, F]; vads: None (every frame voiced) or a list of 1-D arrays (non-zero = voiced).
        Returns a list of float32 [T_voiced, F] arrays, ``None`` for dropped utterances."""
        torch = self.torch
        n = len(mats)
        out = [None] * n
        if n == 0:
            return out
        # everything per frame is done on concatenated arrays (one NumPy call per window, not per utterance)
        T, cand, voiced, empty, dropped = select_voiced(mats, vads)
        for i in empty:
            out[i] = np.zeros((0, mats[i].shape[1]), np.float32)
        self.stats["dropped"] += dropped
        if len(cand) == 0:
            return out
        keep = cand.tolist()
        F = mats[keep[0]].shape[1]
        lens = T[cand]
        starts = np.zeros(len(keep), dtype=np.int64)
        np.cumsum(lens[:-1], out=starts[1:])
        total_in = int(lens.sum())
        assert total_in < 2 ** 31
        if voiced is None:
            counts = lens
            dst = np.arange(total_in, dtype=np.int32)
        else:
            counts = np.add.reduceat(voiced, starts).astype(np.int64)
            dst = np.where(voiced, np.cumsum(voiced, dtype=np.int64) - 1, -1).astype(np.int32)
        ostarts = np.zeros(len(keep), dtype=np.int64)
        np.cumsum(counts[:-1], out=ostarts[1:])
        total_out = int(counts.sum())
        raw = np.concatenate([np.asarray(mats[i], dtype=np.float32) for i in keep], axis=0)
        if self.cmn_window <= 0:                               # selection only: no arithmetic, no device round trip
            host = raw if voiced is None else raw[voiced]
        else:
            with torch.cuda.device(self.device), torch.cuda.stream(self._stream):
                x = torch.from_numpy(raw).to(self.device)
                y = torch.empty((max(total_out, 1), F), dtype=torch.float32, device=self.device)
                hiplib.cmn_sliding_scatter(x, torch.from_numpy(starts.astype(np.int32)).to(self.device),
                                           torch.from_numpy(lens.astype(np.int32)).to(self.device), len(keep), int(lens.max()),
                                           self.cmn_window, self.center, self.min_window, torch.from_numpy(dst).to(self.device), y)
                host = y[:total_out].cpu().numpy()
        for i, o0, c in zip(keep, ostarts.tolist(), counts.tolist()):
            out[i] = host[o0:o0 + c]
        self.stats["utterances"] += len(keep)
        self.stats["frames_in"] += total_in
        self.stats["frames_out"] += total_out
        return out
