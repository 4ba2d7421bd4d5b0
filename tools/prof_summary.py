"""Summarise rocprofv3 rocpd SQLite outputs (ROCm 7.2 default format) into small text files for profiles/.

  python tools/prof_summary.py stats <results.db>            -> per-kernel calls / total / avg / %
  python tools/prof_summary.py pmc   <results.db> [filter]   -> per-kernel, per-counter mean and sum
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 90 else name[:87] + "..."


def stats(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("%-92s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for n, c, t, a, p in rows:
        print("%-92s %8d %14.1f %12.2f %7.2f" % (short(n), c, t, a, p))


def pmc(path, flt=None):
    cur = sqlite3.connect(path).cursor()
    q = ("select kernel_name, counter_name, count(*), avg(value), sum(value), avg(duration) from counters_collection "
         "group by kernel_name, counter_name order by sum(duration) desc")
    print("%-70s %-28s %7s %16s %18s %12s" % ("kernel", "counter", "calls", "mean/launch", "sum", "avg_dur_ns"))
    for n, c, k, a, s, d in cur.execute(q):
        if flt and flt not in n:
            continue
        print("%-70s %-28s %7d %16.1f %18.1f %12.0f" % (short(n)[:70], c, k, a, s, d))


def dispatches(path, flt, limit=40):
    """Per-dispatch counter values (one line per dispatch, counters as columns), for telling layers apart."""
    cur = sqlite3.connect(path).cursor()
    rows = {}
    names = []
    for did, kn, grid, cn, val, dur in cur.execute(
            "select dispatch_id, kernel_name, grid_size, counter_name, value, duration from counters_collection order by dispatch_id"):
        if flt not in kn:
            continue
        rows.setdefault(did, dict(grid=grid, dur=dur))[cn] = val
        if cn not in names:
            names.append(cn)
    print("%8s %10s %10s " % ("dispatch", "grid", "dur_us") + " ".join("%22s" % n[-22:] for n in names))
    for i, (did, r) in enumerate(sorted(rows.items())):
        if i >= limit:
            break
        print("%8d %10d %10.1f " % (did, r["grid"], r["dur"] / 1e3) + " ".join("%22.0f" % r.get(n, float("nan")) for n in names))


if __name__ == "__main__":
    if sys.argv[1] == "dispatches":
        dispatches(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 40)
        sys.exit(0)
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
