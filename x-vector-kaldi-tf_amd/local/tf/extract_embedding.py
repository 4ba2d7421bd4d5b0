#!/usr/bin/env python
"""x-vector extraction worker -- MI355X twin of the reference's ``local/tf/extract_embedding.py``.

Same command line (extract_embedding.py:50-70 of the reference):

    extract_embedding.py --use-gpu {yes,no} --min-chunk-size N --chunk-size N \
        --feature-rspecifier RSPEC --vector-wspecifier WSPEC --model-dir DIR

and the same file protocol: ``ark,scp:A,S`` outputs are written to ``A.tmp.ark`` / ``S.tmp.scp`` and
renamed at the end with the scp text patched (reference lines 94-108, 134-148); the call returns early
when both outputs already exist (126-128); any exception prints a traceback and exits 1 (153-163).
A wspecifier that is a bare ``ark,scp:A,S`` (no ``| copy-vector`` pipe) is written natively by
``kaldi_io.TableWriter`` so no Kaldi binary is needed.  ``--use-gpu`` is accepted and ignored: the
forward pass always runs on the MI355X (see models.py).
"""
from __future__ import print_function

import argparse
import logging
import os
import sys
import traceback

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import kaldi_io  # noqa: E402
from models import Model  # noqa: E402

logger = logging.getLogger('extract_embedding')
logger.setLevel(logging.INFO)
_handler = logging.StreamHandler()
_handler.setLevel(logging.INFO)
_handler.setFormatter(logging.Formatter("%(asctime)s [%(pathname)s:%(lineno)s - %(funcName)s - %(levelname)s ] %(message)s"))
logger.addHandler(_handler)


def get_args(argv=None):
    parser = argparse.ArgumentParser(
        description="Extract x-vectors from Kaldi features with the MI355X-native extractor.",
        formatter_class=argparse.ArgumentDefaultsHelpFormatter, conflict_handler='resolve')
    parser.add_argument("--use-gpu", type=str, dest='use_gpu', choices=["yes", "no"], default="no",
                        help="Accepted for compatibility; the extractor always runs on the GPU.")
    parser.add_argument("--min-chunk-size", type=int, dest='min_chunk_size', default=100,
                        help="Minimum chunk-size allowed when extracting xvectors.")
    parser.add_argument("--chunk-size", type=int, dest='chunk_size', default=-1,
                        help="If set, extracts xvectors from specified chunk-size, and averages.  "
                             "If not set, extracts an xvector from all available features.")
    parser.add_argument("--feature-rspecifier", type=str, dest='feature_rspecifier', required=True,
                        help="Kaldi rspecifier of the input features.")
    parser.add_argument("--vector-wspecifier", type=str, dest='vector_wspecifier', required=True,
                        help="Kaldi wspecifier for the x-vectors (ark, ark pipe, or ark,scp:A,S).")
    parser.add_argument("--model-dir", type=str, dest='model_dir', required=True,
                        help="Model directory (model.meta + weights + done).")
    return process_args(parser.parse_args(argv))


def process_args(args):
    args.model_dir = args.model_dir.strip()
    if args.model_dir == '' or not os.path.exists(os.path.join(args.model_dir, 'model.meta')):
        raise Exception("This scripts expects the input model was exist in '{0}' directory.".format(args.model_dir))
    return args


def process_wspecifier(wspecifier):
    """-> (temporary wspecifier, final ark, final scp); (wspecifier, None, None) when there is nothing to rename."""
    parts = wspecifier.split()
    head = ''.join(p + ' ' for p in parts[:-1])
    last = parts[-1]
    if last.startswith('ark,scp:'):
        ark, scp = last[8:].split(',')
        return head + 'ark,scp:%s.tmp.ark,%s.tmp.scp' % (ark, scp), ark, scp
    if last.startswith('scp,ark:'):
        scp, ark = last[8:].split(',')
        return head + 'scp,ark:%s.tmp.scp,%s.tmp.ark' % (scp, ark), ark, scp
    return wspecifier, None, None


def _open_output(wspecifier, ark, scp):
    if ark is not None and not wspecifier.lstrip().startswith('|'):
        return kaldi_io.TableWriter(ark + '.tmp.ark', scp + '.tmp.scp')
    return kaldi_io.open_or_fd(wspecifier, 'wb')


def eval_dnn(args):
    use_gpu = args.use_gpu == 'yes'
    wspecifier, ark, scp = process_wspecifier(args.vector_wspecifier)
    if ark is not None and os.path.exists(ark) and scp is not None and os.path.exists(scp):
        logger.info('Both output ark and scp files exist. Return from this call.')
        return
    model = Model()
    with kaldi_io.open_or_fd(args.feature_rspecifier) as input_fid:
        with _open_output(wspecifier, ark, scp) as output_fid:
            model.make_embedding(input_fid, output_fid, args.model_dir, args.min_chunk_size, args.chunk_size,
                                 use_gpu, logger)
    if ark is not None:
        os.rename(ark + '.tmp.ark', ark)
    if scp is not None:
        with open(scp + '.tmp.scp', 'rt') as fid_in:
            text = fid_in.read().replace(ark + '.tmp.ark', ark)
        if text and text[-1] != '\n':
            text += '\n'
        with open(scp + '.tmp', 'wt') as fid_out:
            fid_out.write(text)
        os.rename(scp + '.tmp', scp)
        os.remove(scp + '.tmp.scp')


def main(argv=None):
    try:
        eval_dnn(get_args(argv))
    except BaseException as e:
        if not isinstance(e, KeyboardInterrupt):
            traceback.print_exc()
        sys.exit(1)


if __name__ == "__main__":
    main()
