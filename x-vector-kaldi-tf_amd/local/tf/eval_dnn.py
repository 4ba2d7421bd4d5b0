#!/usr/bin/env python
"""Loss / accuracy of a model on an egs archive -- MI355X twin of the reference's ``local/tf/eval_dnn.py`` (the diagnostic
job ``train_dnn.py`` runs on the validation and train-subset archives; its log is parsed by ze_utils.py:498-499).
Same flags (eval_dnn.py:39-53): ``--use-gpu`` (ignored), ``--tar-file``, ``--input-dir``, ``--log-file``."""
from __future__ import print_function

import argparse
import logging
import os
import sys
import traceback

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

from examples_io import TarFileDataLoader  # noqa: E402
from models import Model  # noqa: E402

logger = logging.getLogger('eval_dnn')
logger.setLevel(logging.INFO)
_FORMAT = logging.Formatter("%(asctime)s [%(pathname)s:%(lineno)s - %(funcName)s - %(levelname)s ] %(message)s")


def get_args(argv=None):
    parser = argparse.ArgumentParser(description="Evaluate a trained model on an egs archive (MI355X).",
                                     formatter_class=argparse.ArgumentDefaultsHelpFormatter, conflict_handler='resolve')
    parser.add_argument("--use-gpu", type=str, dest='use_gpu', choices=["yes", "no"], default="yes",
                        help="Accepted for compatibility; evaluation always runs on the GPU.")
    parser.add_argument("--tar-file", type=str, dest='tar_file', required=True, help="egs archive with its .npy label file.")
    parser.add_argument("--input-dir", type=str, dest='input_dir', required=True, help="Model directory.")
    parser.add_argument("--log-file", type=str, dest='log_file', required=True, help="File the evaluation log is written to.")
    print(' '.join(sys.argv))
    args = process_args(parser.parse_args(argv))
    handler = logging.StreamHandler(open(args.log_file, 'wt'))
    handler.setLevel(logging.INFO)
    handler.setFormatter(_FORMAT)
    logger.addHandler(handler)
    logger.info('Starting DNN evaluation (eval_dnn.py)')
    return args


def process_args(args):
    args.input_dir = args.input_dir.strip()
    if not args.input_dir or not os.path.exists(os.path.join(args.input_dir, 'model.meta')):
        raise Exception("This scripts expects the input model was exist in '{0}' directory.".format(args.input_dir))
    if not args.tar_file or not os.path.exists(args.tar_file):
        raise Exception("The specified tar file '{0}' not exist.".format(args.tar_file))
    if not os.path.exists(args.tar_file.replace('.tar', '.npy')):
        raise Exception("There is no corresponding npy label file for tar file '{0}'.".format(args.tar_file))
    return args


def eval_dnn(args):
    data_loader = TarFileDataLoader(args.tar_file, logger=None, queue_size=16)
    Model().eval(data_loader, args.input_dir, args.use_gpu == 'yes', logger)


def main(argv=None):
    try:
        eval_dnn(get_args(argv))
    except BaseException as e:
        if not isinstance(e, KeyboardInterrupt):
            traceback.print_exc()
        sys.exit(1)


if __name__ == "__main__":
    main()
