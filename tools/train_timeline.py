"""Timeline of ONE steady-state training step from a rocprofv3 kernel trace: per stream (queue) the kernels in start order with
their durations and the idle gap in front of each, the busy time per stream and the span of the step.
  rocprofv3 --kernel-trace -d gpurun_out/tt -- python bench.py --mode train --train-precision bf16x3 --steps 12 --warmup 3
  python tools/train_timeline.py $(find gpurun_out/tt -name "*.db" | head -1)"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = next(t for t in tabs if t.startswith("kernels") or t == "kernels")
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kd)]
sys.stderr.write("table %s cols %s\n" % (kd, cols))
name_col = "name" if "name" in cols else "kernel_name"
q_col = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(cur.execute("select %s, start, end, %s from %s order by start" % (name_col, q_col or "0", kd)))
short = lambda n: re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", n)))[:60]
# a step = from one adam_kernel's end to the next one's end; take the 6th
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
lo, hi = adam[5] + 1, adam[6] + 1
step = rows[lo:hi]
t0 = step[0][1]
print("step of %d launches, span %.1f us" % (len(step), (step[-1][2] - t0) / 1e3))
byq = {}
for n, s, e, q in step:
    byq.setdefault(q, []).append((n, s, e))
for q, ks in byq.items():
    busy = sum(e - s for _, s, e in ks)
    print("== queue %s: %d launches, busy %.1f us" % (q, len(ks), busy / 1e3))
    prev = None
    for n, s, e in ks:
        gap = (s - prev) / 1e3 if prev is not None else 0.0
        print("  +%8.1f  gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, short(n)))
        prev = e
