"""Soak run of the random-topology parity tests over many more seeds than the test suite takes (development aid; run on the GPU
box from the repo root): every arithmetic, the first-layer / pair kernels, fp32 with the fused pooling epilogue."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "x-vector-kaldi-tf_amd")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle import oracle
import test_gpu_fuzz as tf
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (40, 64)
bad = 0
only_grad = len(sys.argv) > 3 and sys.argv[3] == "grad"          # (third argument "grad": the training-step part only)
if len(sys.argv) > 3 and sys.argv[3] == "r4":                    # (third argument "r4": the kernel-level fuzzers of the round-4 kernels only)
    for name in ("test_random_shapes_through_the_16x16_f16bf8_kernel", "test_random_shapes_through_the_dma_fed_fp32_gemm"):
        for seed in range(lo, hi):
            try:
                getattr(tf, name)(oracle, seed)
            except AssertionError as e:
                bad += 1
                print(name, seed, "FAIL", str(e)[:300])
        print(name, "seeds %d..%d done" % (lo, hi - 1))
    print("failures:", bad)
    sys.exit(1 if bad else 0)
for name in () if only_grad else ("test_random_topologies_match_the_oracle", "test_random_topologies_through_the_first_layer_and_pair_kernels",
             "test_random_topologies_in_the_f16bf8_arithmetic"):
    fn = getattr(tf, name)
    for seed in range(lo, hi):
        try:
            fn(oracle, seed)
        except AssertionError as e:
            bad += 1
            print(name, seed, "FAIL", str(e)[:200])
    print(name, "seeds %d..%d done" % (lo, hi - 1))
for seed in range(lo, hi):                       # the training step against the float64 autograd oracle (takes the seed only)
    try:
        tf.test_random_topologies_gradients_match_autograd(seed)
    except AssertionError as e:
        # a case whose oracle has a pre-activation within float32 rounding of the activation's kink is not a failure of either side
        # (oracle/train_ref.py: kink_margin): reported, not counted
        import re
        m = re.search(r"kink margin ([0-9.e+-]+)", str(e))
        near = m is not None and float(m.group(1)) < 3e-6
        bad += 0 if near else 1
        print("test_random_topologies_gradients_match_autograd", seed, "NEAR-KINK" if near else "FAIL", str(e)[:400])
print("test_random_topologies_gradients_match_autograd seeds %d..%d done" % (lo, hi - 1))


class _Env(object):                              # (the monkeypatch fixture's setenv, for the test below)
    def setenv(self, k, v):
        os.environ[k] = v


for seed in range(lo, hi):
    try:
        tf.test_random_topologies_bf16x3_step_with_reductions_from_their_producers(seed, _Env())
    except AssertionError as e:
        bad += 1
        print("test_random_topologies_bf16x3_step_with_reductions_from_their_producers", seed, "FAIL", str(e)[:400])
os.environ.pop("XVECTOR_TRAIN_FUSED_SUMS", None)
print("test_random_topologies_bf16x3_step_with_reductions_from_their_producers seeds %d..%d done" % (lo, hi - 1))
if not only_grad:
    for seed in range(lo, hi):
        try:
            tf.test_random_topologies_in_the_fp32tc_arithmetic(oracle, seed)
        except AssertionError as e:
            bad += 1
            print("test_random_topologies_in_the_fp32tc_arithmetic", seed, "FAIL", str(e)[:300])
    print("test_random_topologies_in_the_fp32tc_arithmetic seeds %d..%d done" % (lo, hi - 1))
print("failures:", bad)
