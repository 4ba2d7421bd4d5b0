"""Randomised cross-check of the in-place Kaldi readers (arena reader of ark streams incl. CompressedMatrix records;
scp tables: subsets, several arks, repeated and reversed entries) against the record-by-record reader.  CPU only.

    python tools/fuzz_readers.py ark 0 150      # seeds 0..149 of the ark-stream fuzzer
    python tools/fuzz_readers.py scp 0 120
    python tools/fuzz_readers.py mixed 0 120    # every record type in one stream; arena reader and mapped walk
"""
import io
import os
import struct
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ark_main(argv):
    sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/x-vector-kaldi-tf_amd")
    from local.tf import kaldi_io
    from fixture_inputs import encode_cm_record
    bad = 0
    for seed in range(int(argv[0]), int(argv[1])):
        rng = np.random.default_rng(seed)
        bio, want = io.BytesIO(), []
        n = int(rng.integers(1, 300))
        f = int(rng.choice([1, 5, 20, 23, 24, 30, 40, 80]))
        for i in range(n):
            t = int(rng.choice([1, 2, 7, 8, 9, 16, 33, 100, 257, 1200], p=[.05,.05,.05,.05,.05,.1,.15,.3,.15,.05]))
            if rng.random() < 0.05: f = int(rng.choice([13, 23, 40]))
            m = (rng.standard_normal((t, f)) * float(rng.choice([1e-3, 1, 50, 1e4])) + float(rng.choice([0, 3, -100]))).astype(np.float32)
            kind = rng.random()
            if kind < 0.1: m[:, rng.integers(0, f)] = 2.5            # a constant column
            if kind < 0.03: m[:] = 0
            key = "k%d_%d" % (seed, i)
            r = rng.random()
            if r < 0.1: kaldi_io.write_mat(bio, m, key=key)
            elif r < 0.15:
                two = rng.random() < 0.5
                u = rng.integers(0, 65536 if two else 256, size=m.shape).astype("<u2" if two else np.uint8)
                bio.write(key.encode() + (b" \0BCM2 " if two else b" \0BCM3 ") + struct.pack("<ffii", -3.0, 7.5, t, f) + u.tobytes())
            else: bio.write(encode_cm_record(key, m))
            want.append(key)
        raw = bio.getvalue()
        ref = dict(kaldi_io.read_mat_ark(io.BytesIO(raw)))
        asz = int(rng.choice([1 << 14, 1 << 16, 1 << 18, 1 << 20]))
        lim = None if rng.random() < 0.5 else int(rng.choice([1, 7, 64, 4096]))
        pool = [kaldi_io.ArkArena(asz) for _ in range(4)]
        free, again = list(pool), {}
        try:
            for keys, addr, rows, cols, holder in kaldi_io.scan_mat_ark_windows(io.BytesIO(raw), free.pop, lim, free.append):
                am = kaldi_io.ArkMats(); am.add(addr, rows, cols, holder)
                for j, k in enumerate(keys): again[k] = np.array(am[j])
                if isinstance(holder, kaldi_io.ArkArena): free.append(holder)
            ok = list(again) == want and all(np.array_equal(again[k], ref[k], equal_nan=True) for k in want) and len(free) == 4
        except Exception as e:
            ok = False; print("seed", seed, "exc", repr(e))
        if not ok:
            bad += 1; print("seed", seed, "BAD n", n, "asz", asz, "lim", lim, len(again), len(want), len(free))
    print("done bad", bad)

def scp_main(argv):
    sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/x-vector-kaldi-tf_amd")
    from local.tf import kaldi_io
    from fixture_inputs import encode_cm_record
    bad = 0
    td = tempfile.mkdtemp()
    for seed in range(int(argv[0]), int(argv[1])):
        rng = np.random.default_rng(1000 + seed)
        n = int(rng.integers(1, 400)); f = int(rng.choice([13, 23, 40]))
        mats = [rng.standard_normal((int(rng.integers(1, 80)), f)).astype(np.float32) for _ in range(n)]
        narks = int(rng.integers(1, 4))
        lines = []
        cm = rng.random() < 0.4
        for a in range(narks):
            ark = os.path.join(td, "f%d_%d.ark" % (seed, a))
            with open(ark, "wb") as fh:
                for i in range(a, n, narks):
                    key = "utt%05d" % i
                    at = fh.tell()
                    if cm and i % 5: fh.write(encode_cm_record(key, mats[i]))
                    else: kaldi_io.write_mat(fh, mats[i], key=key)
                    lines.append((i, "%s %s:%d" % (key, ark, at + len(key) + 1)))
        lines.sort()
        ref = {}
        for i, l in lines:
            k, rx = l.split()
            ref[k] = kaldi_io.read_mat(rx)
        dens = float(rng.choice([1.0, 0.95, 0.6, 0.3, 0.1]))
        sel = [l for _, l in lines if rng.random() < dens] or [lines[0][1]]
        if rng.random() < 0.15: sel = sel[::-1]
        if rng.random() < 0.15:
            j = rng.integers(0, len(sel)); sel.append(sel[j])     # a repeated entry
        p = os.path.join(td, "s%d.scp" % seed)
        open(p, "wt").write("\n".join(sel) + "\n")
        asz = int(rng.choice([1 << 13, 1 << 16, 1 << 20]))
        lim = None if rng.random() < 0.5 else int(rng.choice([1, 5, 100]))
        out = []
        try:
            taken, released, held = [], [], []          # every arena taken is either the holder of one item or handed to release

            def take():
                taken.append(kaldi_io.ArkArena(asz))
                return taken[-1]
            for keys, addr, rows, cols, holder in kaldi_io.MatScp(p).windows(take, lim, released.append):
                am = kaldi_io.ArkMats(); am.add(addr, rows, cols, holder)
                out += [(k, np.array(am[j])) for j, k in enumerate(keys)]
                if isinstance(holder, kaldi_io.ArkArena): held.append(holder)
            ok = [k for k, _ in out] == [l.split()[0] for l in sel] and all(np.array_equal(a, ref[k]) for k, a in out)
            ok = ok and sorted(map(id, taken)) == sorted(map(id, held + released))
            blk = [(k, v[int(o[j]):int(o[j + 1])]) for ks, v, o in kaldi_io.MatScp(p).blocks() for j, k in enumerate(ks)]
            ok = ok and [k for k, _ in blk] == [l.split()[0] for l in sel] and all(np.array_equal(a, ref[k]) for k, a in blk)
        except Exception as e:
            import traceback; traceback.print_exc()
            ok = False
        if not ok:
            bad += 1; print("seed", seed, "BAD", n, narks, cm, dens, asz, lim, len(out), len(sel))
    print("done bad", bad)

def mixed_main(argv):
    """Streams that mix every record type the reader knows (float, double, text and compressed matrices, empty ones), through the arena
    reader and through the mapped walk with its copy-free fallback (kaldi_io.MemStream); arena accounting."""
    sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/x-vector-kaldi-tf_amd")
    from local.tf import kaldi_io
    from fixture_inputs import encode_cm_record
    bad = 0
    for seed in range(int(argv[0]), int(argv[1])):
        rng = np.random.default_rng(7000 + seed)
        bio, want = io.BytesIO(), []
        n = int(rng.integers(1, 200)); f = int(rng.choice([5, 23, 40]))
        for i in range(n):
            t = int(rng.choice([0, 1, 3, 9, 50, 300, 1500], p=[.03,.07,.1,.1,.3,.3,.1]))
            m = (rng.standard_normal((t, f)) * 3).astype(np.float32)
            key = "k%d_%d" % (seed, i)
            r = rng.random()
            if r < 0.45 or t == 0 and r < 0.9: kaldi_io.write_mat(bio, m, key=key)
            elif r < 0.6: kaldi_io.write_mat(bio, m.astype(np.float64), key=key)
            elif r < 0.7 and t > 0:
                bio.write((key + "  [\n" + "\n".join(" ".join("%.6g" % v for v in row) for row in m) + " ]\n").encode())
            elif t > 0: bio.write(encode_cm_record(key, m))
            else: kaldi_io.write_mat(bio, m, key=key)
            want.append(key)
        raw = bio.getvalue()
        ref = dict(kaldi_io.read_mat_ark(io.BytesIO(raw)))
        assert list(ref) == want
        asz = int(rng.choice([1 << 13, 1 << 16, 1 << 20]))
        lim = None if rng.random() < 0.5 else int(rng.choice([1, 64, 4096]))
        for src in ("bytesio", "mapped"):
            pool = [kaldi_io.ArkArena(asz) for _ in range(4)]; free, got = list(pool), {}
            try:
                if src == "bytesio":
                    it = kaldi_io.scan_mat_ark_windows(io.BytesIO(raw), free.pop, lim, free.append)
                else:
                    it = kaldi_io.scan_mat_ark_mapped(kaldi_io.map_stream(io.BytesIO(raw)), asz, lim,
                                                      fallback=lambda rest: kaldi_io.scan_mat_ark_windows(kaldi_io.MemStream(rest), free.pop, None, free.append))
                for keys, addr, rows, cols, holder in it:
                    am = kaldi_io.ArkMats(); am.add(addr, rows, cols, holder)
                    for j, k in enumerate(keys): got[k] = np.array(am[j])
                    if isinstance(holder, kaldi_io.ArkArena): free.append(holder)
                ok = list(got) == want and all(got[k].shape == ref[k].shape and np.array_equal(got[k].astype(np.float32), ref[k].astype(np.float32)) for k in want) and len(free) == 4
            except Exception as e:
                import traceback; traceback.print_exc(); ok = False
            if not ok:
                bad += 1; print("seed", seed, src, "BAD", n, f, asz, lim, len(got), len(want), len(free))
    print("done bad", bad)



if __name__ == "__main__":
    {"ark": ark_main, "scp": scp_main, "mixed": mixed_main}[sys.argv[1]](sys.argv[2:4])
