"""Soak run of the random-topology parity tests over many more seeds than the test suite takes (development aid; run on the GPU
box from the repo root): every arithmetic, the first-layer / pair kernels, fp32 with the fused pooling epilogue."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "x-vector-kaldi-tf_amd")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle import oracle
import test_gpu_fuzz as tf
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (40, 64)
bad = 0
for name in ("test_random_topologies_match_the_oracle", "test_random_topologies_through_the_first_layer_and_pair_kernels",
             "test_random_topologies_in_the_f16bf8_arithmetic"):
    fn = getattr(tf, name)
    for seed in range(lo, hi):
        try:
            fn(oracle, seed)
        except AssertionError as e:
            bad += 1
            print(name, seed, "FAIL", str(e)[:200])
    print(name, "seeds %d..%d done" % (lo, hi - 1))
print("failures:", bad)
