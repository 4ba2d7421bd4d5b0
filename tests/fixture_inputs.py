"""Seeded inputs shared by tests/golden/make_golden.py (which records the reference's outputs for them)
and the tests (which regenerate the same inputs and compare against the recorded outputs)."""
import numpy as np

CONTROL_LENGTHS = [0, 24, 25, 99, 100, 200, 1000, 10000, 10001, 10024, 10025, 20024, 25000, 30010]
CONTROL_SETTINGS = [(25, 10000), (100, -1), (25, -1), (100, 10000), (25, 300)]     # (min_chunk, chunk)
CONTROL_SEED, CONTROL_FEAT = 4321, 5

FWD_SEED, FWD_T = 2024, [25, 200, 400, 1000]


def control_inputs():
    rng = np.random.default_rng(CONTROL_SEED)
    return [("key%02d-T%d" % (i, T), (rng.standard_normal((T, CONTROL_FEAT)) * 3.0).astype(np.float32))
            for i, T in enumerate(CONTROL_LENGTHS)]
