#!/usr/bin/env python
"""Speaker-level x-vectors -- what ``ivector-mean ark:spk2utt scp:xvector.scp ark,scp:spk_xvector.ark,spk_xvector.scp
ark,t:num_utts.ark`` does at the end of the reference's extraction script (local/tf/extract_xvectors.sh:97-103), without the
Kaldi binary: for every speaker of spk2utt the plain mean of its utterances' vectors (utterances missing from the table are
skipped with a warning, speakers left with none are skipped), written as a Kaldi float-vector table plus the text table of
utterance counts.  The averaging itself runs through the extraction path's own kernel (xv_chunk_average_f32 with unit
weights: float32 accumulate in spk2utt order, then one divide -- Kaldi's Vector<float> AddVec / Scale order).
"""
from __future__ import print_function

import argparse
import logging
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(_HERE)))

import kaldi_io  # noqa: E402

logger = logging.getLogger('speaker_mean')
logger.addHandler(logging.StreamHandler())
logger.setLevel(logging.INFO)


def read_spk2utt(path):
    with open(path, 'rt') as fid:
        for line in fid:
            parts = line.split()
            if parts:
                yield parts[0], parts[1:]


def speaker_means(spk2utt, vectors, device="cuda:0"):
    """spk2utt: iterable of (speaker, [utt, ...]); vectors: {utt: float32[dim]}.  -> (speakers, means[n, dim], counts)."""
    import torch
    from xvector_amd import hiplib
    speakers, counts, rows, seg = [], [], [], [0]
    for spk, utts in spk2utt:
        have = [u for u in utts if u in vectors]
        for u in utts:
            if u not in vectors:
                logger.warning("No iVector present in input for utterance %s" % u)
        if not have:
            logger.warning("Not producing output for speaker %s since no utterances had iVectors" % spk)
            continue
        speakers.append(spk)
        counts.append(len(have))
        rows.extend(have)
        seg.append(len(rows))
    if not speakers:
        return [], np.zeros((0, 0), np.float32), []
    E = torch.from_numpy(np.stack([np.asarray(vectors[u], np.float32) for u in rows])).to(device)
    out = torch.empty((len(speakers), E.shape[1]), dtype=torch.float32, device=device)
    hiplib.chunk_average(E, torch.tensor(seg, dtype=torch.int32, device=device),
                         torch.ones(len(rows), dtype=torch.int32, device=device), len(speakers), out)
    return speakers, out.cpu().numpy(), counts


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("spk2utt")
    ap.add_argument("xvector_scp", help="scp of utterance-level x-vectors")
    ap.add_argument("out_ark")
    ap.add_argument("out_scp")
    ap.add_argument("num_utts", help="text table <speaker> <number of utterances averaged>")
    args = ap.parse_args(argv)
    vectors = dict(kaldi_io.read_vec_flt_scp(args.xvector_scp))
    speakers, means, counts = speaker_means(read_spk2utt(args.spk2utt), vectors)
    with kaldi_io.TableWriter(args.out_ark, args.out_scp) as out:
        kaldi_io.write_vec_flt_batch(out, speakers, list(means))
    with open(args.num_utts, 'wt') as fid:
        for spk, c in zip(speakers, counts):
            fid.write("%s %d\n" % (spk, c))
    logger.info("Computed mean of %d speakers (%d with no utterances), consisting of %d utterances." %
                (len(speakers), 0, sum(counts)))


if __name__ == "__main__":
    main()
