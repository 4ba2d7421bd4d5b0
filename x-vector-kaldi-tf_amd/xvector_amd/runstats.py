"""Running averages behind the log lines of ``Model.train_one_iteration`` / ``Model.eval``.

The TEXT of those lines is a contract: the reference's ``ze_utils.get_successful_models`` and ``parse_prob_logs``
(local/tf/ze_utils.py:126-127, 498-499) pick the per-job objective and the diagnostic loss / accuracy out of the log
files with regular expressions, and ``generate_report`` builds ``accuracy.report`` from them.  The wording and the
averaging rule therefore follow local/tf/models.py:285-303, 343-354 -- sums are divided by the number of minibatch
INDICES covered (planned count / width of the reporting interval), whether or not every index produced a step -- while
the bookkeeping itself is this one object instead of a dozen loose counters.
"""
import time


class _Sums(object):
    """What is summed per optimizer / eval step."""

    __slots__ = ("segments", "frames", "loss", "accuracy")

    def __init__(self):
        self.segments = self.frames = 0
        self.loss = self.accuracy = 0.0

    def add(self, segments, frames, loss, accuracy):
        self.segments += segments
        self.frames += frames
        self.loss += loss
        self.accuracy += accuracy


class Meter(object):
    """``planned``: minibatch indices the pass will visit (``data_loader.count``); ``report_every``: width of the
    progress interval in indices (``args.print_interval``; None = no progress lines, as in ``eval``)."""

    def __init__(self, planned, report_every=None):
        self.planned = int(planned)
        self.report_every = int(report_every) if report_every else None
        self.total = _Sums()
        self.window = _Sums()
        self.window_first = 0                      # first index the open interval covers
        self.wait = dict(disk=0.0, gpu=0.0)        # seconds inside the open interval
        self.started = time.time()
        self._stepped_last = False

    # -- feeding -----------------------------------------------------------------------------------
    def waited(self, what, seconds):
        self.wait[what] += seconds

    def stepped(self, index, segments, frames, loss, accuracy):
        self.total.add(segments, frames, loss, accuracy)
        self.window.add(segments, frames, loss, accuracy)
        self._stepped_last = True

    def skipped(self, index):
        self._stepped_last = False

    def interval_due(self, index):
        """Would ``index`` close a progress interval if it produced a step?  (A caller that feeds steps one late asks this to
        know when it has to catch up before ``interval_line``.)"""
        return bool(self.report_every) and (index + 1) % self.report_every == 0

    # -- lines -------------------------------------------------------------------------------------
    def interval_line(self, index):
        """Progress line when ``index`` closes an interval, else None.  An index without a step never closes one (the
        interval then runs on to the next multiple, models.py:253,277)."""
        if not self.report_every or not self._stepped_last or (index + 1) % self.report_every:
            return None
        width = index + 1 - self.window_first
        w = self.window
        line = ("Average training loss for minibatches %d-%d is %.4f over %d segments. Also, the "
                "average training accuracy for these minibatches is %.4f and the average "
                "objective function for these minibatches is %.4f. Average DISK waiting: %.1f "
                "secs and average GPU waiting: %.1f secs for each minibatch." %
                (self.window_first + 1, index + 1, w.loss / width, w.segments, w.accuracy / width, -w.loss / width,
                 self.wait["disk"] / width, self.wait["gpu"] / width))
        self.window = _Sums()
        self.window_first = index + 1
        self.wait = dict(disk=0.0, gpu=0.0)
        return line

    def _processed_line(self):
        t, n = self.total, self.planned
        return ("Processed %d segments of average size %d into %d minibatches. Avg minibatch size was %d." %
                (t.segments, t.frames / n, n, t.segments / n))

    def training_summary(self):
        t, n = self.total, self.planned
        return [self._processed_line(),
                "Overall average training loss is %.4f over %d segments. Also, the overall "
                "average training accuracy is %.4f." % (t.loss / n, t.segments, t.accuracy / n),
                "Overall average objective function is %.4f over %d segments." % (-t.loss / n, t.segments)]

    def eval_summary(self):
        t, n = self.total, self.planned
        return [self._processed_line(),
                "Overall average loss is %.4f over %d segments. Also, the overall "
                "average accuracy is %.4f." % (t.loss / n, t.segments, t.accuracy / n)]

    def elapsed_line(self):
        return ("Elapsed time for processing whole training minibatches is %.2f minutes." %
                ((time.time() - self.started) / 60.0))
