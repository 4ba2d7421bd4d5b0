"""Timing-only variants of tdnn_pair_pool_f16bf8_kernel (WRONG results on purpose) -> build/variants/libxv_p8_<name>.so, run through
XVECTOR_HIP_LIB by tools/pair8_bench.py: what the exchange of partial tiles through LDS, the pooling arithmetic, the stage barriers
and the weight DMA cost (the LDS budget of a stage: 512 cycles of fragment reads + ~256 of DMA writes + ~256 of exchange writes +
~128 of pooling reads against 1024 MFMA cycles).
  nokeep   the half a wave finishes itself is not written to / read back from LDS (pooled from a register instead)
  noexch   neither half is written or read back
  nopool   no pooling arithmetic, exchange reads or statistics stores
  nobar    no stage barriers
  nodma    no weight / frames DMA in the loops
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "x-vector-kaldi-tf_amd", "csrc", "xv_pair8.hip")
OUT = os.path.join(ROOT, "build", "variants")
src = open(SRC).read()


def rep(s, old, new, count=None):
    assert old in s, old
    return s.replace(old, new) if count is None else s.replace(old, new, count)


def variant(name):
    s = src
    if name in ("nokeep", "noexch"):
        s = rep(s, "                    keep_mine[r * 64] = hf ? y[1][r] : y[0][r];\n", "")
        s = rep(s, "        pa[i] = pool_keep[off];\n", "        pa[i] = prm[1] + (float)off;\n")
        if name == "noexch":
            s = rep(s, "                    red_mine[r * 64] = hf ? y[0][r] : y[1][r];\n", "                    { float t0 = y[0][r], t1 = y[1][r]; asm volatile(\"\" :: \"v\"(t0), \"v\"(t1)); }\n")
            s = rep(s, "        pb[i] = pool_red[off];\n", "        pb[i] = prm[2] + (float)off;\n")
    elif name == "nopool":
        s = rep(s, "        if constexpr (count >= 1) pool_read(std::integral_constant<int, first>{}, I0{});\n        if constexpr (count == 2) pool_read(std::integral_constant<int, first + 1>{}, I1{});\n", "")
        s = rep(s, "        if constexpr (count >= 1) pool_row(std::integral_constant<int, first>{}, I0{}, ct - 1);\n        if constexpr (count == 2) pool_row(std::integral_constant<int, first + 1>{}, I1{}, ct - 1);\n", "")
        s = s.replace("pin_a(std::integral_constant<int, 4 + 2 * pool_count(q, 0)>{})", "pin_a(std::integral_constant<int, 4>{})")
        s = s.replace("pin_a(std::integral_constant<int, 4 + 2 * pool_count(q, 1)>{})", "pin_a(std::integral_constant<int, 4>{})")
        s = s.replace("pin_a(std::integral_constant<int, 6 + 2 * pool_count(q, 2)>{})", "pin_a(std::integral_constant<int, 6>{})")
        s = s.replace("pin_a(std::integral_constant<int, 2 + 2 * pool_count(q, 3)>{})", "pin_a(std::integral_constant<int, 2>{})")
    elif name == "nobar":
        s = rep(s, '        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");\n        __builtin_amdgcn_s_barrier();\n', '        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");\n')
    elif name == "nodma":
        s = rep(s, "        XV_BLDS16(wrs, lds + fill + wave * 4096, wvoff, wsoff, k * 1024);\n", "        if (wleft < 0) XV_BLDS16(wrs, lds + fill + wave * 4096, wvoff, wsoff, k * 1024);\n")
    else:
        raise SystemExit("unknown variant " + name)
    return s


for name in sys.argv[1:]:
    path = os.path.join(OUT, "xv_pair8_%s.hip" % name)
    open(path, "w").write(variant(name))
    obj = os.path.join(OUT, "xv_pair8_%s.o" % name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.dirname(SRC), "-Wno-unused-function", "-Wno-unused-variable", "-c", "-o", obj, path])
    objs = [os.path.join(ROOT, "build", "obj", f) for f in os.listdir(os.path.join(ROOT, "build", "obj")) if f.endswith(".o") and f != "xv_pair8.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, "libxv_p8_%s.so" % name), obj] + objs)
    print("built", name)
