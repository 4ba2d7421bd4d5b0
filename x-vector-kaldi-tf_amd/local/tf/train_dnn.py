#!/usr/bin/env python
"""Training driver on one MI355X node -- the twin of the reference's ``local/tf/train_dnn.py`` with the batch queue
replaced by one process group.

The reference runs an iteration by submitting ``num_jobs`` copies of ``train_dnn_one_iteration.py`` to a queue, each on its
own egs archive, keeping the model of the best job (its model averaging is a stub, ze_utils.py:164-183) and submitting two
``eval_dnn.py`` diagnostics (train_dnn.py:224-460).  Here the jobs of an iteration are the ranks of ONE ``torchrun`` group
(one process per GPU, RCCL over xGMI):

    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 local/tf/train_dnn.py --tf-model-class ModelWithoutDropout \\
        --dir exp/xvector_nnet --egs-dir exp/xvector_nnet/egs --num-targets 7185 --minibatch-size 64 --num-epochs 6

* rank j plays job j+1 of the iteration: it reads archive ``((num_archives_processed + j) % num_archives) + 1``
  (train_dnn.py:246-249; ``egs.<n>.tar`` when it exists, else the ranges/scp pair, train_dnn.py:258-267);
* the ranks train SYNCHRONOUSLY: every optimizer step all-reduces the flat gradient buffer (bucketed, overlapped with the backward pass: Trainer.step), so all
  ranks hold the same weights and rank 0 writes ``model_<iter+1>`` -- no job selection, no averaging pass.  An iteration
  takes as many steps as its shortest archive has minibatches;
* learning-rate and dropout schedules, iteration count, ``model_0`` creation, ``model_name.txt``, clean-up rule,
  ``model_final`` link and ``accuracy.report`` follow train_dnn.py:463-593; the per-job logs ``log/train.<iter>.<job>.log``
  and the diagnostics ``log/compute_prob_{valid,train_subset}.<iter>.log`` carry the reference's log lines (and the
  ``# Accounting: time=`` trailer its report reads), so its own ``ze_utils.generate_report`` parses them too.

Same flags as the reference (train_dnn.py:29-186).  ``--num-jobs-initial/--num-jobs-final`` are the queue's degree of
parallelism there; here the number of jobs per iteration is WORLD_SIZE (1 without torchrun) and the two flags are only
checked.  Queue options (``--cmd``), momentum / max-param-change / shrinkage and the model-combination flags are accepted and
unused, as they are by the reference's TensorFlow path.
"""
from __future__ import print_function

import argparse
import logging
import os
import pprint
import sys
import time
import traceback

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import models  # noqa: E402
import ze_utils as utils  # noqa: E402
from examples_io import RangesDataLoader, TarFileDataLoader  # noqa: E402
from train_dnn_one_iteration import LOG_FORMAT  # noqa: E402
from xvector_amd import dist as xdist  # noqa: E402

logger = logging.getLogger('train_dnn')
logger.setLevel(logging.INFO)
_handler = logging.StreamHandler(sys.stdout)
_handler.setFormatter(LOG_FORMAT)
logger.addHandler(_handler)

_TRUE_FALSE = ("true", "false")


def _flag_parser():
    p = argparse.ArgumentParser(description="Train an x-vector DNN on the GPUs of one MI355X node (launch with torchrun for "
                                            "more than one GPU).", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    a = p.add_argument
    a("--tf-model-class", dest="tf_model_class", required=True, help="Class name in models.py.")
    a("--dir", required=True, help="Experiment directory: model_<iter>/, log/, model_name.txt, accuracy.report.")
    a("--egs-dir", dest="egs_dir", required=True, help="egs directory (info/, temp/, egs.<n>.tar, *_egs.1.tar).")
    a("--num-targets", dest="num_targets", type=int, required=True, help="Number of speakers (output classes).")
    a("--minibatch-size", dest="minibatch_size", type=int, required=True, help="Chunks per minibatch of the archives.")
    a("--num-epochs", dest="num_epochs", type=float, default=6.0)
    a("--initial-effective-lrate", dest="initial_effective_lrate", type=float, default=0.0003)
    a("--final-effective-lrate", dest="final_effective_lrate", type=float, default=0.00003)
    a("--num-jobs-initial", dest="num_jobs_initial", type=int, default=1)
    a("--num-jobs-final", dest="num_jobs_final", type=int, default=8)
    a("--random-seed", dest="random_seed", type=int, default=0)
    a("--dropout-schedule", dest="dropout_schedule", default=None, help="e.g. '0,0@0.20,0.1@0.50,0' (ze_utils.py:365-425).")
    a("--preserve-model-interval", dest="preserve_model_interval", type=int, default=10)
    a("--cleanup", type=lambda v: {"true": True, "false": False}[v], choices=[True, False], default=True, metavar="{true,false}")
    a("--stage", type=int, default=-4, help="Resume: iterations below this number are not run again.")
    a("--print-interval", dest="print_interval", type=int, default=10)
    # accepted for command-line compatibility; unused here as in the reference's TensorFlow path
    a("--use-gpu", dest="use_gpu", choices=["yes", "no", "wait"], default="yes")
    a("--momentum", type=float, default=0.0)
    a("--targets-scp", dest="targets_scp", default=None)
    a("--do-final-combination", dest="do_final_combination", choices=_TRUE_FALSE, default="false")
    a("--max-objective-evaluations", dest="max_objective_evaluations", type=int, default=30)
    a("--max-param-change", dest="max_param_change", type=float, default=2.0)
    a("--proportional-shrink", dest="proportional_shrink", type=float, default=0.0)
    a("--cmd", dest="command", default="queue.pl")
    a("--max-models-combine", dest="max_models_combine", type=int, default=20)
    return p


def get_args(argv=None):
    args = _flag_parser().parse_args(argv)
    if args.do_final_combination == "true":
        raise Exception("--do-final-combination true: combining models is not implemented (nor is it in the reference, "
                        "train_dnn.py:568-571)")
    if not hasattr(models, args.tf_model_class):
        raise Exception("models.py has no class '%s'" % args.tf_model_class)
    return args


class _Capped(object):
    """A data loader limited to its first ``count`` minibatches (ranks of one iteration must take the same number of
    optimizer steps: each step is a collective)."""

    def __init__(self, loader, count):
        self.loader, self.count = loader, int(count)

    def pop(self, timeout=30):
        return self.loader.pop(timeout)


def _file_logger(name, path):
    log = logging.getLogger(name)
    log.setLevel(logging.INFO)
    log.propagate = False
    for h in list(log.handlers):
        log.removeHandler(h)
        h.close()
    sink = logging.FileHandler(path, mode="wt")
    sink.setFormatter(LOG_FORMAT)
    log.addHandler(sink)
    return log, sink


def _open_archive(egs_dir, index, minibatch_count, args, feat_dim):
    tar = os.path.join(egs_dir, "egs.%d.tar" % index)
    if os.path.exists(tar):                                           # train_dnn.py:264-267
        return TarFileDataLoader(tar, logger=None, queue_size=16)
    return RangesDataLoader(os.path.join(egs_dir, "temp", "ranges.%d" % index), os.path.join(egs_dir, "temp", "feats.scp.%d" % index),
                            minibatch_count, args.minibatch_size, feat_dim, shuffle=True)


def _diagnostics(exp_dir, egs_dir, _iter, rank, world):
    """compute_prob_valid / compute_prob_train_subset of model_<_iter> (train_dnn.py:429-460), spread over the ranks."""
    jobs = [("compute_prob_valid", "valid_egs.1.tar"), ("compute_prob_train_subset", "train_subset_egs.1.tar")]
    for n, (stem, tar_name) in enumerate(jobs):
        tar = os.path.join(egs_dir, tar_name)
        if n % world != rank or not os.path.exists(tar) or not utils.is_correct_model_dir(os.path.join(exp_dir, "model_%d" % _iter)):
            continue                                                  # (a resumed run may find model_<_iter> cleaned up already)
        log, sink = _file_logger("%s.%d" % (stem, _iter), os.path.join(exp_dir, "log", "%s.%d.log" % (stem, _iter)))
        loader = TarFileDataLoader(tar, queue_size=16)
        try:
            models.Model().eval(loader, os.path.join(exp_dir, "model_%d" % _iter), True, log)
        finally:
            loader.close()
            log.removeHandler(sink)
            sink.close()


def train(args):
    import torch.distributed as dist
    rank, world = xdist.init_process_group()
    barrier = dist.barrier if dist.is_initialized() else (lambda: None)
    exp_dir, egs_dir = args.dir, args.egs_dir
    if rank == 0:
        logger.info("Arguments for the experiment\n{0}".format(pprint.pformat(vars(args))))
        os.makedirs(os.path.join(exp_dir, "log"), exist_ok=True)
    num_archives, feat_dim, archive_counts = utils.verify_egs_dir(egs_dir)
    if world > num_archives:
        raise Exception("%d ranks cannot each take a different archive: the egs directory has %d" % (world, num_archives))
    if rank == 0 and (args.num_jobs_initial != world or args.num_jobs_final != world):
        logger.info("Every iteration runs %d synchronous job(s) (WORLD_SIZE); --num-jobs-initial/--num-jobs-final (%d/%d) "
                    "describe the reference's queue and are not used." % (world, args.num_jobs_initial, args.num_jobs_final))

    if args.stage <= -1 and rank == 0:                                 # train_dnn.py:488-500
        if not os.path.exists(os.path.join(exp_dir, "model_0", "done")):
            logger.info("Preparing the initial network.")
            getattr(models, args.tf_model_class)().build_model(args.num_targets, feat_dim, os.path.join(exp_dir, "model_0"), logger=logger)
            with open(os.path.join(exp_dir, "model_name.txt"), "wt") as fid:
                fid.write(args.tf_model_class)
        else:
            logger.info("The initial network exist from before.")
    barrier()

    num_archives_to_process = int(args.num_epochs * num_archives)
    num_iters = max(1, num_archives_to_process // world)              # train_dnn.py:506 with jobs_initial == jobs_final == world
    if rank == 0:
        logger.info("Training will run for {0} epochs = {1} iterations".format(args.num_epochs, num_iters))
    processed = 0
    for _iter in range(num_iters):
        next_dir = os.path.join(exp_dir, "model_%d" % (_iter + 1))
        if args.stage <= _iter:
            lrate = utils.get_learning_rate(_iter, world, num_iters, processed, num_archives_to_process,
                                            args.initial_effective_lrate, args.final_effective_lrate)
            dropout = utils.get_dropout_edit_string(args.dropout_schedule, float(processed) / num_archives_to_process)
            if rank == 0:
                logger.info("Iter: {0}/{1}    Epoch: {2:0.2f}/{3:0.1f} ({4:0.1f}% complete)    lr: {5:0.6f}    ".format(
                    _iter, num_iters - 1, processed * args.num_epochs / num_archives_to_process, args.num_epochs,
                    processed * 100.0 / num_archives_to_process, lrate))
            _diagnostics(exp_dir, egs_dir, _iter, rank, world)
            if utils.is_correct_model_dir(next_dir):                   # train_dnn.py:342-345
                if rank == 0:
                    logger.info("The output model %s was exist and so I do not continue this iteration." % next_dir)
            else:
                index = (processed + rank) % num_archives + 1          # train_dnn.py:246-249
                steps = min(archive_counts[(processed + j) % num_archives + 1] for j in range(world))
                loader = _open_archive(egs_dir, index, archive_counts[index], args, feat_dim)
                log, sink = _file_logger("train.%d.%d" % (_iter, rank + 1), os.path.join(exp_dir, "log", "train.%d.%d.log" % (_iter, rank + 1)))
                job = argparse.Namespace(learning_rate=lrate, print_interval=args.print_interval, dropout_proportion=dropout or 0.0,
                                         input_dir=os.path.join(exp_dir, "model_%d" % _iter), output_dir=next_dir,
                                         random_seed=_iter + args.random_seed, save_model=(rank == 0))
                t0 = time.time()
                try:
                    models.Model().train_one_iteration(_Capped(loader, steps), job, log)
                finally:
                    if hasattr(loader, "close"):
                        loader.close()
                    sink.stream.write("# Accounting: time=%d threads=1\n" % int(round(time.time() - t0)))
                    log.removeHandler(sink)
                    sink.close()
                barrier()
                if rank == 0 and not utils.is_correct_model_dir(next_dir):
                    raise Exception("Could not find a complete model in {0} at the end of iteration {1}".format(next_dir, _iter))
            if args.cleanup and rank == 0:
                utils.remove_model(exp_dir, _iter - 2, None, args.preserve_model_interval)      # train_dnn.py:565-567
            barrier()
        processed += world

    if rank == 0:
        if args.stage <= num_iters:
            utils.force_symlink("model_%d" % num_iters, os.path.join(exp_dir, "model_final"))     # train_dnn.py:583
        if args.cleanup:
            logger.info("Cleaning up the experiment directory {0}".format(exp_dir))
            for _iter in range(num_iters):
                utils.remove_model(exp_dir, _iter, None, args.preserve_model_interval)
        report = utils.generate_report(exp_dir)[0]
        with open(os.path.join(exp_dir, "accuracy.report"), "wt") as fid:
            fid.write(report)
    barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


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
