"""GPU timeline of the ark -> ark path (development aid): run under  rocprofv3 --kernel-trace --memory-copy-trace  and pass the
.db here: kernel-busy time against the span from the first to the last kernel of the LAST make_embedding call, and the
largest idle gaps with the kernels on either side.
    python tools/e2e_timeline.py run <n_utts>      (the workload: 3 make_embedding calls on an in-memory ark)
    python tools/e2e_timeline.py report <results.db>"""
import io, logging, os, sqlite3, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "x-vector-kaldi-tf_amd"), os.path.join(ROOT, "x-vector-kaldi-tf_amd", "local", "tf")):
    sys.path.insert(0, p)


def run(n):
    import kaldi_io, models
    from xvector_amd import synthetic, topology as tp
    topo = tp.get("ModelWithoutDropout"); w = synthetic.trained_like(topo, 23, seed=1)
    d = tempfile.mkdtemp()
    models.Model.save_model(dict(weights=w, topology=topo, model_class="Model", num_classes=64, feat_dim=23), d, None)
    bio = io.BytesIO()
    for k, m in synthetic.make_utterances(n, 200, 400, 23, 1234):
        kaldi_io.write_mat(bio, m, key=k)
    raw = bio.getvalue()
    log = logging.getLogger("e2e"); log.setLevel(logging.ERROR)
    for rep in range(3):
        out = io.BytesIO(); t0 = time.time()
        models.Model().make_embedding(io.BytesIO(raw), out, d, 25, 10000, False, log); dt = time.time() - t0
        print("pass %d: %d utts in %.3f s -> %.0f utt/s" % (rep, n, dt, n / dt))


def report(path):
    cur = sqlite3.connect(path).cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kt = next(t for t in tables if t == "kernels" or t.startswith("kernels"))
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kt)]
    print("table", kt, cols)
    rows = list(cur.execute("select name, start, end from %s order by start" % kt))
    t00 = rows[0][1]
    print("whole trace: gaps > 2 ms:")
    for i in range(1, len(rows)):
        if rows[i][1] - rows[i - 1][2] > 2e6:
            print("   +%8.1f ms: idle %.1f ms before %s" % ((rows[i - 1][2] - t00) / 1e6, (rows[i][1] - rows[i - 1][2]) / 1e6, rows[i][0].split("(")[0][:50]))
    # the last call = after the last gap longer than 60 ms (python between two calls: building the output, the next model)
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][1] - rows[i - 1][2] > float(os.environ.get("CALL_GAP_MS", "60")) * 1e6:
            cut = i
    rows = rows[cut:]
    span = rows[-1][2] - rows[0][1]
    busy, last_end, gaps = 0, rows[0][1], []
    for n, s, e in rows:
        if s > last_end:
            gaps.append((s - last_end, last_end - rows[0][1], n))
        busy += max(0, e - max(s, last_end))
        last_end = max(last_end, e)
    print("last call: %d kernels, span %.1f ms, busy %.1f ms (%.1f %%)" % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span))
    tot = {}
    for n, s, e in rows:
        tot[n.split("(")[0][:60]] = tot.get(n.split("(")[0][:60], 0) + e - s
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:8]:
        print("   %-62s %8.1f ms" % (k, v / 1e6))
    print("idle: %d gaps, total %.1f ms; > 100 us: %d gaps, %.1f ms" % (len(gaps), sum(g[0] for g in gaps) / 1e6, sum(1 for g in gaps if g[0] > 1e5), sum(g[0] for g in gaps if g[0] > 1e5) / 1e6))
    for g, at, n in sorted(gaps, reverse=True)[:12]:
        print("   gap %7.1f us at +%6.1f ms before %s" % (g / 1e3, at / 1e6, n.split("(")[0][:50]))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 50000)
    else:
        report(sys.argv[2])
