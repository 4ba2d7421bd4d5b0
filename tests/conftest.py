"""pytest configuration: `gpu` marker, import paths, shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "x-vector-kaldi-tf_amd")
TWIN = os.path.join(PKG, "local", "tf")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, PKG, TWIN, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def default_weights():
    """(topology, trained-like weights) for the default topology, seed as in the golden fixtures."""
    from xvector_amd import synthetic, topology
    topo = topology.get("ModelWithoutDropout")
    return topo, synthetic.trained_like(topo, 23, seed=2024)


# Collection order (the driver runs `pytest -m gpu -x`): the kernel-by-kernel and whole-network parity tests against the oracle
# come FIRST, files that start subprocesses (benches, rehearsals over several ranks, CLIs) LAST -- a hiccup in a bench contract or a
# rendezvous must not leave every parity row of SURVEY section 8 "untested".  Within a class the order of collection is kept.
_FIRST = ("test_gpu_kernels.py", "test_gpu_toom.py", "test_gpu_f16bf8.py", "test_gpu_forward.py", "test_gpu_fuzz.py", "test_gpu_hostile.py",
          "test_gpu_frontend.py", "test_gpu_training.py", "test_gpu_config3.py", "test_gpu_trained_checkpoint.py")
_LAST = ("test_gpu_two_ranks.py", "test_gpu_eight_ranks.py", "test_gpu_bench_contract.py")


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.basename(str(item.fspath))
        if name in _FIRST:
            return _FIRST.index(name)
        if name in _LAST:
            return 1000 + _LAST.index(name)
        return 500
    items.sort(key=rank)                      # (stable: ties keep their collection order)
