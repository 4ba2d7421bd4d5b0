#!/usr/bin/env python
"""One training iteration on one egs archive -- MI355X twin of the reference's ``local/tf/train_dnn_one_iteration.py``.

``train_dnn.py`` launches this once per job and iteration (train_dnn.py:258-300) with the model of the previous iteration
(``--input-dir``), the archive (``--tar-file``), the per-iteration learning rate and dropout proportion (ze_utils.py
schedules) and the directory for the new model (``--output-dir``); the log it prints is parsed back by
``ze_utils.get_successful_models``.  Same flags as the reference (train_dnn_one_iteration.py:41-134); the ones its model
code never reads (momentum, max-param-change, l2-regularize-factor, scale, verbose, use-gpu, sequential-loading) are
accepted and ignored here too.  Both input modes are provided: the egs tar (``--tar-file``, preferred when it exists,
train_dnn.py:264-267) and the ranges/scp mode (``--ranges-file`` + ``--scp-file`` [+ ``--shuffle``]).
"""
from __future__ import print_function

import argparse
import logging
import os
import pprint
import sys
import traceback

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import numpy as np  # noqa: E402

import models  # noqa: E402
from examples_io import RangesDataLoader, TarFileDataLoader  # noqa: E402

logger = logging.getLogger('train_dnn_one_iteration')
logger.setLevel(logging.INFO)
_handler = logging.StreamHandler(sys.stdout)
_handler.setLevel(logging.INFO)
LOG_FORMAT = logging.Formatter("%(asctime)s [%(pathname)s:%(lineno)s - %(funcName)s - %(levelname)s ] %(message)s")
_handler.setFormatter(LOG_FORMAT)
logger.addHandler(_handler)

# (flag, dest, type, default, required, choices, help) -- the reference's command line, train_dnn_one_iteration.py:41-134
_FLAGS = (
    ("--use-gpu", "use_gpu", str, "yes", False, ("yes", "no", "wait"), "Accepted for compatibility; training always runs on the GPU."),
    ("--momentum", "momentum", float, 0.0, False, None, "Accepted for compatibility (unused by the reference's model code)."),
    ("--shuffle", "shuffle", bool, False, False, None, "Permute the minibatches of the ranges/scp mode (any non-empty value = true, as in the reference)."),
    ("--max-param-change", "max_param_change", float, 2.0, False, None, "Accepted for compatibility (unused)."),
    ("--l2-regularize-factor", "l2_regularize_factor", float, 1.0, False, None, "Accepted for compatibility (unused)."),
    ("--ranges-file", "ranges_file", str, "", False, None, "Kaldi ranges file of this archive (used when no --tar-file is given)."),
    ("--scp-file", "scp_file", str, "", False, None, "Feature scp restricted to the utterances of the ranges file."),
    ("--sequential-loading", "sequential_loading", str, "true", False, ("true", "false"), "Accepted for compatibility."),
    ("--random-seed", "random_seed", int, 0, False, None, "Seed of the dropout masks (and of NumPy, as in the reference)."),
    ("--print-interval", "print_interval", int, 10, False, None, "The interval for log printing."),
    ("--verbose", "verbose", int, 0, False, None, "Accepted for compatibility (unused)."),
    ("--feature-dim", "feature_dim", int, None, True, None, "Feature dimension; checked against the model."),
    ("--minibatch-size", "minibatch_size", int, None, True, None, "Minibatch size of the archive."),
    ("--minibatch-count", "minibatch_count", int, None, True, None, "Number of minibatches in the archive."),
    ("--learning-rate", "learning_rate", float, -1.0, False, None, "Learning rate of this iteration (Adam)."),
    ("--scale", "scale", float, 1.0, False, None, "Accepted for compatibility (unused)."),
    ("--dropout-proportion", "dropout_proportion", float, 0.0, False, None, "Dropout proportion of this iteration."),
    ("--tar-file", "tar_file", str, "", False, None, "egs archive (minibatch_<i>.npy members) with its .npy label file."),
    ("--input-dir", "input_dir", str, None, True, None, "Model directory to start from."),
    ("--output-dir", "output_dir", str, None, True, None, "Directory the new model is written to."),
)


def parse_flags(table, description, argv=None):
    """argparse from a (flag, dest, type, default, required, choices, help) table; echoes the command line like the
    reference's scripts do."""
    parser = argparse.ArgumentParser(description=description, formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                                     conflict_handler='resolve')
    for flag, dest, typ, default, required, choices, text in table:
        kw = dict(dest=dest, type=typ, help=text)
        if required:
            kw["required"] = True
        else:
            kw["default"] = default
        if choices:
            kw["choices"] = list(choices)
        parser.add_argument(flag, **kw)
    print(' '.join(sys.argv))
    return parser.parse_args(argv)


def check_model_dir(path):
    path = (path or "").strip()
    if not path or not os.path.exists(os.path.join(path, 'model.meta')):
        raise Exception("no model found in '%s' (model.meta is missing)" % path)


def check_archive(tar_file):
    if not tar_file or not os.path.exists(tar_file):
        raise Exception("egs archive '%s' does not exist" % tar_file)
    if not os.path.exists(tar_file.replace('.tar', '.npy')):
        raise Exception("egs archive '%s' has no label file next to it (.npy)" % tar_file)


def get_args(argv=None):
    args = parse_flags(_FLAGS, "One DNN training iteration on the MI355X.", argv)
    check_model_dir(args.input_dir)
    args.input_dir = args.input_dir.strip()
    if args.tar_file:
        check_archive(args.tar_file)
    else:
        for what, path in (("ranges", args.ranges_file), ("scp", args.scp_file)):
            if not path or not os.path.exists(path):
                raise Exception("the %s file '%s' does not exist (and no --tar-file was given)" % (what, path))
    if not 0.0 <= args.dropout_proportion <= 1.0:
        raise Exception("--dropout-proportion must lie in [0, 1]")
    return args


def train(args):
    logger.info("Arguments for the experiment\n{0}".format(pprint.pformat(vars(args))))
    if args.random_seed != 0:
        np.random.seed(args.random_seed)
    if args.tar_file:
        data_loader = TarFileDataLoader(args.tar_file, logger=None, queue_size=16)
    else:
        data_loader = RangesDataLoader(args.ranges_file, args.scp_file, args.minibatch_count, args.minibatch_size,
                                       args.feature_dim, shuffle=args.shuffle)
    models.Model().train_one_iteration(data_loader, args, logger)      # the model class comes from the model directory


def main(argv=None):
    args = get_args(argv)          # outside the try, as in the reference: --help / usage errors exit through argparse
    try:
        train(args)
    except BaseException as e:
        if not isinstance(e, KeyboardInterrupt):
            traceback.print_exc()
        sys.exit(1)


if __name__ == "__main__":
    main()
