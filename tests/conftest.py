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
