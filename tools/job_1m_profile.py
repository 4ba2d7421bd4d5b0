"""Host-side pieces of the 1 M-utterance job (BASELINE configs[3]) timed alone, on the box they will run on:
the scp shard cut of one of 8 ranks (kaldi_io.ScpText: native line index, own range decoded) against the per-line Python split it
replaced, and rank 0's write of 1 M x-vectors into ark + scp on tmpfs by the native record writer (xv_vec_records_write_fd, from
the gathered [emitted? | x-vector] blocks as they lie) against the NumPy / Python serialisation (XVECTOR_NATIVE_WRITER=0).
    python tools/job_1m_profile.py [n_utterances]"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "x-vector-kaldi-tf_amd"), os.path.join(ROOT, "x-vector-kaldi-tf_amd", "local", "tf")]
import extract_embedding as ee  # noqa: E402
import kaldi_io  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
work = tempfile.mkdtemp(prefix="xv_1m_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    p = os.path.join(work, "feats.scp")
    with open(p, "w") as f:
        f.write("".join("spk%05d-utt%07d /data/feats/raw_mfcc.%d.ark:%d\n" % (i % 9973, i, i % 80, 17 + i * 27653) for i in range(n)))
    for rep in range(2):
        t = time.perf_counter()
        mine, _, sk = ee._scp_shard("scp:" + p, 3, 8)
        t_new = time.perf_counter() - t
        t = time.perf_counter()
        with open(p, "rb") as fid:
            lines = [ln for ln in fid.read().decode().splitlines(True) if ln.strip()]
        cuts = [len(lines) * r // 8 for r in range(9)]
        old_keys = [[ln.split(None, 1)[0] for ln in lines[cuts[r]:cuts[r + 1]]] for r in range(8)]
        t_old = time.perf_counter() - t
        assert mine == lines[cuts[3]:cuts[4]] and sk[3] == old_keys[3]
        print("scp shard of rank 3 of 8, %d lines: ScpText %.3f s, per-line Python split %.3f s" % (n, t_new, t_old))
    t = time.perf_counter()
    _, v, _ = ee._scp_shard("scp:" + p, 3, 8, "scp:" + p)
    print("  with a parallel vad.scp: %.3f s" % (time.perf_counter() - t))
    block = np.random.default_rng(0).standard_normal((n // 8, 513)).astype(np.float32)
    emitted = np.ones(n // 8, bool)
    emitted[::5000] = False
    for mode in ("1", "0", "1"):
        os.environ["XVECTOR_NATIVE_WRITER"] = mode
        a, s = os.path.join(work, "o.ark"), os.path.join(work, "o.scp")
        t = time.perf_counter()
        with kaldi_io.TableWriter(a, s, scp_ark_name=a) as out:
            for r in range(8):
                kaldi_io.write_vec_flt_batch(out, sk[r], block[:, 1:], emitted)
        dt = time.perf_counter() - t
        print("write of %d records (%.2f GB ark + scp) by %s: %.3f s" % (int(emitted.sum()) * 8, os.path.getsize(a) / 1e9,
                                                                          "the native writer" if mode == "1" else "NumPy / Python", dt))
        os.remove(a), os.remove(s)
finally:
    shutil.rmtree(work, ignore_errors=True)
