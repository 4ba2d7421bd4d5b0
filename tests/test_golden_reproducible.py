"""The committed fixtures in tests/golden/ are what tests/golden/make_golden.py produces from the reference's own Python
TODAY: when /root/reference is mounted (the build container; never the GPU box) the generator is re-run into a scratch
directory and every array of every fixture must come out bit for bit.  This is what "pinned against the reference" means
for the rows DESIGN.md section 5 lists as reference-pinned (control flow of make_embedding, ark framing, schedules, the egs
loader) -- and, since round 6, for the arithmetic: forward_refgraph.npz / train_refgraph.npz are what the reference's own graphs and
training loop compute when executed under tests/golden/numpy_tf1.py.  Byte fixtures must come out bit for bit; the float64 arrays of the
two refgraph files within 1e-12 (a BLAS summation order may differ between machines), everything else in them exactly."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.mark.skipif(not os.path.isdir("/root/reference/local/tf"), reason="the reference tree is only mounted in the build container")
def test_committed_fixtures_reproduce_bit_for_bit(tmp_path):
    env = dict(os.environ, XV_GOLDEN_OUT=str(tmp_path), PYTHONDONTWRITEBYTECODE="1")
    run = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden.py")], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=1500)
    assert run.returncode == 0, run.stdout.decode()[-3000:]
    names = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    assert names == sorted(f for f in os.listdir(str(tmp_path)) if f.endswith(".npz")) and len(names) == 6
    for name in names:
        with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as want, np.load(str(tmp_path / name), allow_pickle=False) as got:
            assert sorted(want.files) == sorted(got.files), name
            for k in want.files:
                a, b = want[k], got[k]
                assert a.dtype == b.dtype and a.shape == b.shape, (name, k)
                if name.endswith("_refgraph.npz") and a.dtype == np.float64 and not k.endswith("_init_stats"):
                    assert np.linalg.norm(a - b) <= 1e-12 * max(np.linalg.norm(a), 1e-300), (name, k)
                else:
                    assert a.tobytes() == b.tobytes(), (name, k)
