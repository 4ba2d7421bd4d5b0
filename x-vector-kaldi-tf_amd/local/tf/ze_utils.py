"""The slice of the reference's ``local/tf/ze_utils.py`` that the training hot path consumes.

``train_dnn.py`` computes two per-iteration scalars with these helpers and hands them to
``Model.train_one_iteration`` through ``args``: the learning rate (ze_utils.py:111-120, called at train_dnn.py:524)
and the dropout proportion (ze_utils.py:310-443, called at train_dnn.py:545).  Plus the model-directory predicate
(ze_utils.py:561-567), the egs-directory reader (ze_utils.py:56-73), the model-directory housekeeping of the training
loop (ze_utils.py:186-194,250-257) and an accuracy report in the reference's table layout (ze_utils.py:531-558) for the
in-process driver ``train_dnn.py``.  The queue.pl command lines, background-command threads and the model-averaging stub
of that file are control plane and out of scope.  Schedules pinned against values recorded from the reference itself:
tests/golden/schedules.npz.
"""
import datetime
import glob
import logging
import math
import os
import re
import shutil

from xvector_amd.weights import is_correct_model_dir          # noqa: F401  (same predicate, same name)

logger = logging.getLogger(__name__)
logger.addHandler(logging.NullHandler())


def get_learning_rate(_iter, num_jobs, num_iters, num_archives_processed, num_archives_to_process,
                      initial_effective_lrate, final_effective_lrate):
    """Exponential decay of the EFFECTIVE rate in the fraction of archives processed, pinned to the final value on the
    last iteration, times the number of parallel jobs (ze_utils.py:111-120)."""
    if _iter + 1 < num_iters:
        log_ratio = math.log(final_effective_lrate / initial_effective_lrate)
        rate = initial_effective_lrate * math.exp(num_archives_processed * log_ratio / num_archives_to_process)
    else:
        rate = final_effective_lrate
    return num_jobs * rate


def _parse_dropout_string(dropout_str):
    """'p_start, p@f, ..., p_end' -> [(data_fraction, proportion)] in DESCENDING fraction order (ze_utils.py:365-424).
    A middle item without '@' sits at fraction 0.5; fractions must not decrease and must stay <= 1."""
    items = dropout_str.strip().split(',')
    try:
        if len(items) < 2:
            raise ValueError("a dropout schedule needs at least a start and an end value")
        knots = [(0, float(items[0]))]
        for item in items[1:-1]:
            fields = item.split('@')
            if len(fields) == 1:
                fraction, proportion = 0.5, float(fields[0])
            elif len(fields) == 2:
                proportion, fraction = float(fields[0]), float(fields[1])
            else:
                raise ValueError("more than one '@' in '%s'" % item)
            if fraction < knots[-1][0] or fraction > 1.0:
                logger.error("Failed while parsing value %s in dropout-schedule. dropout-schedule must be in "
                             "increasing order of data fractions.", item)
                raise ValueError(item)
            knots.append((fraction, proportion))
        knots.append((1.0, float(items[-1])))
    except Exception:
        logger.error("Unable to parse dropout proportion string %s. See help for option --trainer.dropout-schedule.",
                     dropout_str)
        raise
    knots.reverse()
    for fraction, proportion in knots:
        assert 0.0 <= fraction <= 1.0 and 0.0 <= proportion <= 1.0
    return knots


def _get_component_dropout(dropout_schedule, data_fraction):
    """Piecewise-linear interpolation on a descending schedule (ze_utils.py:310-362), including its tie rules: the
    segment is [first knot with fraction <= data_fraction, the knot listed just before it)."""
    if data_fraction == 0:
        assert dropout_schedule[-1][0] == 0
        return dropout_schedule[-1][1]
    lower = None
    for index, (fraction, proportion) in enumerate(dropout_schedule):
        if fraction <= data_fraction:
            lower = index
            break
    if lower is None:
        raise RuntimeError("Could not find data_fraction in dropout schedule corresponding to data_fraction {0}.\n"
                           "Maybe something wrong with the parsed dropout schedule {1}.".format(data_fraction, dropout_schedule))
    lo_fraction, lo_value = dropout_schedule[lower]
    if lower == 0:
        assert lo_fraction == 1 and data_fraction == 1
        return lo_value
    hi_fraction, hi_value = dropout_schedule[lower - 1]
    if hi_fraction == lo_fraction:
        assert data_fraction == lo_fraction
        return lo_value
    assert lo_fraction <= data_fraction < hi_fraction
    return (data_fraction - lo_fraction) * (hi_value - lo_value) / (hi_fraction - lo_fraction) + lo_value


def get_dropout_edit_string(dropout_schedule, data_fraction):
    """Dropout proportion for the fraction of data seen so far, or None without a schedule (ze_utils.py:427-443)."""
    if dropout_schedule is None:
        return None
    return _get_component_dropout(_parse_dropout_string(dropout_schedule), data_fraction)


# ------------------------------------------------------------------------------------------------
# training-loop housekeeping (used by the in-process driver train_dnn.py)
# ------------------------------------------------------------------------------------------------
def verify_egs_dir(egs_dir):
    """-> [num_archives, feat_dim, {archive index: minibatch count}] from ``info/feat_dim``, ``info/num_archives`` and
    ``temp/archive_minibatch_count`` ("<archive> <count>" per line), the three files get_egs.sh leaves behind and
    ze_utils.py:56-73 reads."""
    def first_int(rel):
        with open(os.path.join(egs_dir, rel), "rt") as fid:
            return int(fid.readline())
    try:
        feat_dim, num_archives = first_int("info/feat_dim"), first_int("info/num_archives")
        counts = {}
        with open(os.path.join(egs_dir, "temp", "archive_minibatch_count"), "rt") as fid:
            for fields in (line.split() for line in fid):
                if fields:
                    counts[int(fields[0])] = int(fields[1])
    except (IOError, ValueError):
        logger.error("The egs dir %s has missing or malformed files." % egs_dir)
        raise
    return [num_archives, feat_dim, counts]


def remove_model(nnet_dir, _iter, models_to_combine=None, preserve_model_interval=100):
    """Delete ``model_<_iter>`` unless it is a preserved multiple or needed for the final combination (ze_utils.py:186-194)."""
    keep = _iter % preserve_model_interval == 0 or (models_to_combine is not None and _iter in models_to_combine)
    path = os.path.join(nnet_dir, "model_%d" % _iter)
    if not keep and os.path.isdir(path):
        shutil.rmtree(path)


def force_symlink(target, link_name):
    """ln -sf (ze_utils.py:250-257)."""
    if os.path.islink(link_name) or os.path.exists(link_name):
        os.remove(link_name)
    os.symlink(target, link_name)


_PROB_LINE = re.compile(r"Overall average loss is ([0-9.eE+-]+) over \d+ segments\. Also, the overall average accuracy is ([0-9.eE+-]+)\.")
_TIME_LINE = re.compile(r"# Accounting: time=([0-9]+) ")


def _per_iteration(exp_dir, stem, regex, reduce):
    out = {}
    for path in glob.glob(os.path.join(exp_dir, "log", stem + ".*.log")):
        m = re.match(re.escape(stem) + r"\.(\d+)(?:\.\d+)?\.log$", os.path.basename(path))
        if not m:
            continue
        with open(path, "rt") as fid:
            hits = regex.findall(fid.read())
        if hits:
            it = int(m.group(1))
            out[it] = reduce(out[it], hits[-1]) if it in out else hits[-1]
    return out


def generate_report(exp_dir, key="accuracy"):
    """-> [report text, {iter: seconds}, [(iter, train_loss, train_acc, valid_loss, valid_acc)]] from the compute_prob_* and
    train.* logs of ``exp_dir``; same columns as the reference's accuracy.report (ze_utils.py:531-558)."""
    train = _per_iteration(exp_dir, "compute_prob_train_subset", _PROB_LINE, lambda a, b: b)
    valid = _per_iteration(exp_dir, "compute_prob_valid", _PROB_LINE, lambda a, b: b)
    times = {it: float(sec) for it, sec in _per_iteration(exp_dir, "train", _TIME_LINE, lambda a, b: max(a, b, key=float)).items()}
    rows = ["%Iter\tduration\ttrain_loss\tvalid_loss\tdifference\ttrain_acc\tvalid_acc\tdifference"]
    data = []
    for it in sorted(set(train) & set(valid)):
        tl, ta, vl, va = float(train[it][0]), float(train[it][1]), float(valid[it][0]), float(valid[it][1])
        data.append((it, tl, ta, vl, va))
        if it in times:
            rows.append("%d\t%s\t%g\t%g\t%g\t%g\t%g\t%g" % (it, times[it], tl, vl, vl - tl, ta, va, ta - va))
    rows.append("Total training time is %s\n" % datetime.timedelta(seconds=sum(times.values())))
    return ["\n".join(rows), times, data]
