#!/usr/bin/env python
"""Diagnostic pass over one egs archive: average loss / accuracy of a stored model in the eval phase (moving batch-norm
statistics) -- the job ``train_dnn.py`` schedules on its validation and train-subset archives, whose log the accuracy
report reads back (reference: local/tf/eval_dnn.py:30-86, ze_utils.py:498-499).

    eval_dnn.py --tar-file egs/valid_diagnostic_egs.1.tar --input-dir exp/xvector_nnet/model_12 --log-file LOG [--use-gpu yes|no]

The forward pass always runs on the MI355X; ``--use-gpu`` is only parsed.
"""
from __future__ import print_function

import logging
import os
import sys
import traceback

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import train_dnn_one_iteration as _cli  # noqa: E402  (shares the flag-table parser and the archive checks)
from examples_io import TarFileDataLoader  # noqa: E402
from models import Model  # noqa: E402

logger = logging.getLogger('eval_dnn')
logger.setLevel(logging.INFO)

_FLAGS = (
    ("--use-gpu", "use_gpu", str, "yes", False, ("yes", "no"), "Parsed for compatibility."),
    ("--tar-file", "tar_file", str, None, True, None, "egs archive; <archive>.npy holds the labels."),
    ("--input-dir", "input_dir", str, None, True, None, "Model directory to evaluate."),
    ("--log-file", "log_file", str, None, True, None, "Where the evaluation log goes (kept apart from stdout)."),
)


def get_args(argv=None):
    args = _cli.parse_flags(_FLAGS, "Loss and accuracy of a model on an egs archive (MI355X).", argv)
    _cli.check_model_dir(args.input_dir)
    _cli.check_archive(args.tar_file)
    args.input_dir = args.input_dir.strip()
    sink = logging.StreamHandler(open(args.log_file, 'wt'))
    sink.setLevel(logging.INFO)
    sink.setFormatter(_cli.LOG_FORMAT)
    logger.addHandler(sink)
    logger.info('Starting DNN evaluation (eval_dnn.py)')
    return args


def eval_dnn(args):
    Model().eval(TarFileDataLoader(args.tar_file, queue_size=16), args.input_dir, args.use_gpu == 'yes', logger)


def main(argv=None):
    args = get_args(argv)          # outside the try, as in the reference: --help / usage errors exit through argparse
    try:
        eval_dnn(args)
    except KeyboardInterrupt:
        sys.exit(1)
    except BaseException:
        traceback.print_exc()
        sys.exit(1)


if __name__ == "__main__":
    main()
