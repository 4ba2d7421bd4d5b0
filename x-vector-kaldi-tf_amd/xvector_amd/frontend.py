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

    def apply(self, mats, vads=None):
        """mats: list of float32 [T, F]; vads: None (every frame voiced) or a list of 1-D arrays (non-zero = voiced).
        Returns a list of float32 [T_voiced, F] arrays, ``None`` for dropped utterances."""
        torch = self.torch
        n = len(mats)
        out = [None] * n
        keep, sel = [], []
        for i, m in enumerate(mats):
            T = m.shape[0]
            if vads is None or vads[i] is None:
                idx = np.arange(T, dtype=np.int64)
            else:
                v = np.asarray(vads[i]).reshape(-1)
                if v.shape[0] != T or not np.any(v != 0):          # select-voiced-frames: mismatch / nothing voiced
                    self.stats["dropped"] += 1
                    continue
                idx = np.flatnonzero(v != 0)
            if T == 0:
                out[i] = np.zeros((0, m.shape[1]), np.float32)
                continue
            keep.append(i)
            sel.append(idx)
        if not keep:
            return out
        if self.cmn_window <= 0:                               # selection only: no arithmetic, no device round trip
            for i, idx in zip(keep, sel):
                out[i] = np.ascontiguousarray(mats[i][idx], dtype=np.float32)
                self.stats["frames_in"] += mats[i].shape[0]
                self.stats["frames_out"] += len(idx)
            self.stats["utterances"] += len(keep)
            return out
        F = mats[keep[0]].shape[1]
        lens = np.array([mats[i].shape[0] for i in keep], dtype=np.int64)
        starts = np.zeros(len(keep), dtype=np.int64)
        np.cumsum(lens[:-1], out=starts[1:])
        total_in = int(lens.sum())
        assert total_in < 2 ** 31
        counts = np.array([len(s) for s in sel], dtype=np.int64)
        ostarts = np.zeros(len(keep), dtype=np.int64)
        np.cumsum(counts[:-1], out=ostarts[1:])
        total_out = int(counts.sum())
        dst = np.full(total_in, -1, dtype=np.int32)
        for s0, o0, idx in zip(starts, ostarts, sel):
            dst[s0 + idx] = np.arange(o0, o0 + len(idx), dtype=np.int32)
        raw = np.concatenate([np.ascontiguousarray(mats[i], dtype=np.float32) for i in keep], axis=0)
        with torch.cuda.device(self.device):
            x = torch.from_numpy(raw).to(self.device)
            y = torch.empty((max(total_out, 1), F), dtype=torch.float32, device=self.device)
            hiplib.cmn_sliding_scatter(x, torch.from_numpy(starts.astype(np.int32)).to(self.device),
                                       torch.from_numpy(lens.astype(np.int32)).to(self.device), len(keep), int(lens.max()),
                                       self.cmn_window, self.center, self.min_window, torch.from_numpy(dst).to(self.device), y)
            host = y[:total_out].cpu().numpy()
        for i, o0, c in zip(keep, ostarts, counts):
            out[i] = host[o0:o0 + c]
        self.stats["utterances"] += len(keep)
        self.stats["frames_in"] += total_in
        self.stats["frames_out"] += total_out
        return out
