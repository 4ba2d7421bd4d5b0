"""The slice of the reference's ``local/tf/ze_utils.py`` that the training hot path consumes.

``train_dnn.py`` computes two per-iteration scalars with these helpers and hands them to
``Model.train_one_iteration`` through ``args``: the learning rate (ze_utils.py:111-120, called at train_dnn.py:524)
and the dropout proportion (ze_utils.py:310-443, called at train_dnn.py:545).  Plus the model-directory predicate
(ze_utils.py:561-567).  Everything else in that file (queue.pl command lines, log parsing, model averaging) is control
plane and out of scope.  Pinned against values recorded from the reference itself: tests/golden/schedules.npz.
"""
import logging
import math

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
