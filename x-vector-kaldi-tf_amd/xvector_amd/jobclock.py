"""Wall-clock marks of one extraction job, from the birth of the process: what a user replacing the reference's
``extract_xvectors.sh`` job (local/tf/extract_xvectors.sh:72-89) actually waits for -- interpreter start, imports, process
group, model load, the first window, the drain, the exchange, the write -- as opposed to the resident kernel rate."""
import os
import time

_MARKS = []


def process_birth():
    """time.time() at which this process was created (from /proc: start time in clock ticks after boot), or None."""
    try:
        with open("/proc/self/stat", "rb") as f:
            fields = f.read().rsplit(b") ", 1)[1].split()
        ticks = float(fields[19])                       # field 22 of stat(5): starttime
        with open("/proc/uptime", "rb") as f:
            up = float(f.read().split()[0])
        return time.time() - (up - ticks / os.sysconf("SC_CLK_TCK"))
    except Exception:
        return None


_NOTES = []


def note(name, seconds):
    """A duration measured elsewhere (a side thread), reported next to the marks."""
    _NOTES.append((name, float(seconds)))


_HOOKS = {}


def on(name, fn):
    """``fn()`` runs (once) when ``name`` is marked -- or at once if it already was.  The process group's side thread waits for
    "first window launched" this way (dist.init_process_group_async)."""
    if any(n == name for n, _ in _MARKS):
        fn()
    else:
        _HOOKS.setdefault(name, []).append(fn)


def mark(name):
    _MARKS.append((name, time.time()))
    for fn in _HOOKS.pop(name, []):
        fn()


def once(name):
    if all(n != name for n, _ in _MARKS):
        mark(name)


def report(reset=True):
    """[(name, seconds since the previous mark)] starting at the birth of the process, plus ("total", ...)."""
    t_prev = process_birth()
    if t_prev is None and _MARKS:
        t_prev = _MARKS[0][1]
    t0, out = t_prev, []
    for name, t in _MARKS:
        out.append((name, t - t_prev))
        t_prev = t
    if _MARKS:
        out.append(("total", _MARKS[-1][1] - t0))
    if reset:
        del _MARKS[:]
    return out


def line(reset=True):
    notes = "".join("; [%s %.3f s]" % kv for kv in _NOTES)
    if reset:
        del _NOTES[:]
    return "Job wall clock: " + ", ".join("%s %.3f s" % kv for kv in report(reset)) + notes
